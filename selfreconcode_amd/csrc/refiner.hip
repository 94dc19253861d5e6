// Device-driven masked Newton ray / surface refiner (utils/FindSurfacePs.py:114-163; the
// `sr_trace_newton` of SURVEY.md 8(b)).  The set of unfinished rays lives in device memory as a dense, compacted queue:
// every step ends with a wave-ballot compaction (one atomicAdd per wave on the next step's counter), so a ray leaves the
// M dimension of the layer GEMMs the moment it passes the convergence test -- as the reference's `initTmpPs[unfinished]`
// does -- but without the boolean gather / host synchronisation the reference pays per step: the live count never
// visits the host, the layer chains (sr_mlp_chain) and the kernels below read it from the `live` array.
//
// Phases of one call (P rays, T = times):
//   phase 0      : forward chain -> mid(CHECK): rays that already pass are retired, the rest compacted
//   phase 1..T   : forward chain -> mid(STEP): LBS + Jacobian, convergence test of the current points,
//                  cotangents of the residual -> reverse chain -> finish: Newton update of the failing rays,
//                  retirement of the passing ones, compaction
//   phase T+1    : forward chain -> mid(FINAL): test of the last update, everything retired
// = 1 + 2 + 4 T + 2 launches, whatever the number of live rays.  Every kernel that puts a ray into a queue (init, mid(CHECK),
// finish) also writes the ray's first-layer input rows of both networks at its slot (embed_rows below), so the next forward
// chain follows it directly; sr_refine_embed is the same arithmetic as a launch of its own.  Results land at the rays' ORIGINAL
// indices, so the order in which the waves claim queue slots does not matter (a row's arithmetic does not depend on its position).
#include <algorithm>
#include "lbs_device.h"

namespace {
constexpr float kRad2Deg = 180.0f / 3.14159265358979323846f;

__device__ __forceinline__ int wave_compact_slot(bool keep, int32_t* counter) {
  const unsigned long long m = __ballot(keep);
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == 0 && m) base = atomicAdd(counter, (int)__popcll(m));
  base = __shfl(base, 0, 64);
  return base + (int)__popcll(m & ((1ull << lane) - 1ull));
}

// One element of a ray's first-layer input rows, columns [0, ld_a0) = SDF network [x | PE_L(x)], [ld_a0, ld_a0 + ld_a0d) =
// deformer [x | PE_L(x) | code[frame]]; same arithmetic as pe_embed_kernel (model/Embedder.py:9-41).
__device__ __forceinline__ void embed_store(const sr_refine_args& g, int64_t p, int c, float x0, float x1, float x2, int frame) {
  const bool def = c >= g.ld_a0;
  if (def) c -= g.ld_a0;
  const int L = def ? g.L_def : g.L_sdf;
  const float* w = def ? g.w_def : g.w_sdf;
  float v = 0.f;
  if (c < 3) {
    v = c == 0 ? x0 : (c == 1 ? x1 : x2);
  } else if (c < 3 + 6 * L) {
    const int k = (c - 3) / 6, r = (c - 3) % 6, comp = r % 3;
    const float a = (comp == 0 ? x0 : (comp == 1 ? x1 : x2)) * (float)(1 << k);
    const float wk = w[2 * k + (r >= 3)];
    v = r < 3 ? wk * sinf(a) : wk * cosf(a);
  } else if (def && c < 3 + 6 * L + g.E) {
    v = g.conds[(int64_t)frame * g.ld_conds + (c - 3 - 6 * L)];
  }
  (def ? g.a0d + p * g.ld_a0d : g.a0 + p * g.ld_a0)[c] = v;
}

__global__ __launch_bounds__(256) void refine_init_kernel(sr_refine_args g) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < g.P) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { g.x[0][i * 3 + c] = g.p0[i * 3 + c]; g.v[0][i * 3 + c] = g.rays[i * 3 + c]; }
    g.frame[0][i] = (int32_t)g.batch_inds[i];
    g.orig[0][i] = (int32_t)i;
    g.unit[i * 4] = 1.f; g.unit[i * 4 + 1] = 0.f; g.unit[i * 4 + 2] = 0.f; g.unit[i * 4 + 3] = 0.f;
  }
  if (i <= g.times + 2) g.live[i] = i == 0 ? g.P : 0;
  if (g.a0 && g.a0d) {                                  // phase 0's first-layer inputs: every ray, at its own index
    const int wtot = (int)(g.ld_a0 + g.ld_a0d);
    const int64_t total = (int64_t)g.P * wtot;
    for (int64_t e = i; e < total; e += (int64_t)gridDim.x * blockDim.x) {
      const int64_t p = e / wtot;
      embed_store(g, p, (int)(e % wtot), g.p0[p * 3], g.p0[p * 3 + 1], g.p0[p * 3 + 2], (int32_t)g.batch_inds[p]);
    }
  }
}

// First-layer inputs of both networks for the live rays of `phase` (the stand-alone form of what init / mid(CHECK) / finish do
// for the rays they enqueue).
__global__ __launch_bounds__(256) void refine_embed_kernel(sr_refine_args g, int phase) {
  const int M = g.live[phase], cur = phase & 1;
  const int wtot = (int)(g.ld_a0 + g.ld_a0d);
  const int64_t total = (int64_t)M * wtot;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / wtot;
    embed_store(g, p, (int)(i % wtot), g.x[cur][p * 3], g.x[cur][p * 3 + 1], g.x[cur][p * 3 + 2], g.frame[cur][p]);
  }
}

// The enqueueing kernels work on kEnq rays per 256-thread workgroup: one ray per lane of the first quarter-wave for the per-ray
// part, then ALL 256 threads write the kEnq x (ld_a0 + ld_a0d) input elements of the rays that stay in the queue.  (One ray
// per thread would leave a 6k-ray call with 24 workgroups for ~1.3 M sin / cos evaluations.)
constexpr int kEnq = 16;
struct EnqRow { float x0, x1, x2; int frame, slot; };
__device__ __forceinline__ void embed_rows(const sr_refine_args& g, EnqRow* rows, bool keep, int slot, float x0, float x1, float x2, int frame) {
  if (threadIdx.x < kEnq) rows[threadIdx.x] = EnqRow{x0, x1, x2, frame, keep ? slot : -1};
  __syncthreads();
  if (g.a0 && g.a0d) {
    const int wtot = (int)(g.ld_a0 + g.ld_a0d);
    for (int i = threadIdx.x; i < kEnq * wtot; i += blockDim.x) {
      const EnqRow r = rows[i / wtot];
      if (r.slot >= 0) embed_store(g, r.slot, i % wtot, r.x0, r.x1, r.x2, r.frame);
    }
  }
  __syncthreads();
}

struct Residual { float s, ex, ey, ez; bool ok; };
// convergence test (FindSurfacePs.py:115-126 / :153-161) and d s / d y of the ray term s = |(y - c) x v| / |y - c|
__device__ __forceinline__ Residual residual_of(const sr_refine_args& g, float f, const float (&y)[3], float vx, float vy, float vz) {
  Residual r;
  const float dx = y[0] - g.cam[0], dy = y[1] - g.cam[1], dz = y[2] - g.cam[2];
  const float ux = dy * vz - dz * vy, uy = dz * vx - dx * vz, uz = dx * vy - dy * vx;   // (d - c) x v
  const float un = sqrtf(ux * ux + uy * uy + uz * uz), dn = sqrtf(dx * dx + dy * dy + dz * dz);
  r.s = un / dn;
  r.ok = (fabsf(f) < g.dthreshold) && (asinf(r.s) * kRad2Deg < g.athreshold);
  const float iu = un > 0.f ? 1.f / un : 0.f;
  const float hx = ux * iu, hy = uy * iu, hz = uz * iu;
  r.ex = (vy * hz - vz * hy) / dn - un * dx / (dn * dn * dn);
  r.ey = (vz * hx - vx * hz) / dn - un * dy / (dn * dn * dn);
  r.ez = (vx * hy - vy * hx) / dn - un * dz / (dn * dn * dn);
  return r;
}

// mode 0 CHECK : test, retire the passing rays, compact the others into the next queue (positions unchanged)
// mode 1 STEP  : test + LBS Jacobian; conv flag, cotangent t = J_lbs^T ds/dy of the deformation offset, s
// mode 2 FINAL : test, retire everything
template <int MODE>
__global__ __launch_bounds__(256) void refine_mid_kernel(sr_refine_args g, int phase) {
  __shared__ float sA[8 * srlbs::NJ * 12];
  __shared__ float sT[8 * 3];
  __shared__ EnqRow rows[kEnq];
  constexpr int RPB = MODE == 0 ? kEnq : 256;             // rays per workgroup and pass (CHECK also writes the kept rays' input rows)
  const bool stage = g.nframes <= 8;
  if (stage) {
    for (int i = threadIdx.x; i < g.nframes * srlbs::NJ * 12; i += blockDim.x) sA[i] = g.A[i];
    for (int i = threadIdx.x; i < g.nframes * 3; i += blockDim.x) sT[i] = g.trans[i];
    __syncthreads();
  }
  const int M = g.live[phase], cur = phase & 1, nxt = cur ^ 1;
  for (int64_t base = (int64_t)blockIdx.x * RPB; base < M; base += (int64_t)gridDim.x * RPB) {
    const int64_t i = base + threadIdx.x;
    const bool valid = (int)threadIdx.x < RPB && i < M;
    bool ok = false;
    float px = 0.f, py = 0.f, pz = 0.f, vx = 0.f, vy = 0.f, vz = 0.f;
    int frame = 0;
    if (valid) {
      px = g.x[cur][i * 3]; py = g.x[cur][i * 3 + 1]; pz = g.x[cur][i * 3 + 2];
      vx = g.v[cur][i * 3]; vy = g.v[cur][i * 3 + 1]; vz = g.v[cur][i * 3 + 2];
      frame = g.frame[cur][i];
      const float f = g.sdf_out[i * g.ld_sdf];
      const float* off = g.def_out + i * g.ld_def;
      const float qx = px + off[0], qy = py + off[1], qz = pz + off[2];     // MLPTranslator: p + offset (Deformer.py:72)
      const float* Af = stage ? sA + frame * srlbs::NJ * 12 : g.A + (int64_t)frame * srlbs::NJ * 12;
      const float* tf = stage ? sT + frame * 3 : g.trans + frame * 3;
      float y[3], J[9];
      srlbs::lbs_point<MODE == 1>(qx, qy, qz, qx, qy, qz, g.vol, g.D, g.H, g.W, g.bmin, g.bmax, Af, tf, y, J);
      const Residual r = residual_of(g, f, y, vx, vy, vz);
      ok = r.ok;
      if (MODE == 1) {
        g.conv[i] = ok ? 1 : 0;
        float* t = g.t + i * 4;
        t[0] = J[0] * r.ex + J[3] * r.ey + J[6] * r.ez;
        t[1] = J[1] * r.ex + J[4] * r.ey + J[7] * r.ez;
        t[2] = J[2] * r.ex + J[5] * r.ey + J[8] * r.ez;
        t[3] = 0.f;
        g.s[i] = r.s;
      }
    }
    if (MODE != 1) {
      const bool retire = valid && (ok || MODE == 2);
      if (retire) {
        const int o = g.orig[cur][i];
        g.p_out[o * 3] = px; g.p_out[o * 3 + 1] = py; g.p_out[o * 3 + 2] = pz;
        g.conv_out[o] = ok ? 1 : 0;
      }
      if (MODE == 0) {
        const bool keep = valid && !ok;
        const int slot = wave_compact_slot(keep, g.live + phase + 1);
        if (keep) {
          g.x[nxt][slot * 3] = px; g.x[nxt][slot * 3 + 1] = py; g.x[nxt][slot * 3 + 2] = pz;
          g.v[nxt][slot * 3] = vx; g.v[nxt][slot * 3 + 1] = vy; g.v[nxt][slot * 3 + 2] = vz;
          g.frame[nxt][slot] = frame;
          g.orig[nxt][slot] = g.orig[cur][i];
        }
        embed_rows(g, rows, keep, slot, px, py, pz, frame);
      }
    }
  }
}

// d/dx of one network's first-layer input row: cotangent row g0 (+ optional second row added on the first n2 columns)
__device__ __forceinline__ void pe_pullback(const float (&x)[3], int L, const float* __restrict__ w, const float* __restrict__ g0,
                                            const float* __restrict__ g2, int n2, float (&out)[3]) {
  auto at = [&](int c) { return g0[c] + ((g2 && c < n2) ? g2[c] : 0.f); };
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float acc = at(c);
    for (int k = 0; k < L; ++k) {
      const float f = (float)(1 << k);
      float sn, cs;
      sincosf(x[c] * f, &sn, &cs);
      const int is = 3 + 6 * k + c, ic = is + 3;
      acc += f * (w[2 * k] * cs * at(is) - w[2 * k + 1] * sn * at(ic));
    }
    out[c] = acc;
  }
}

// End of a step: g = w1 sign(f) grad f + w2 (t + J_off^T t),  p <- p - L g / |g|^2 for the rays that failed this step's
// test (FindSurfacePs.py:146-151); the passing ones are retired at their current position; compaction.
__global__ __launch_bounds__(256) void refine_finish_kernel(sr_refine_args g, int phase) {
  __shared__ EnqRow rows[kEnq];
  const int M = g.live[phase], cur = phase & 1, nxt = cur ^ 1;
  for (int64_t base = (int64_t)blockIdx.x * kEnq; base < M; base += (int64_t)gridDim.x * kEnq) {
    const int64_t i = base + threadIdx.x;
    const bool valid = (int)threadIdx.x < kEnq && i < M;
    bool keep = false;
    int frame = 0;
    float x[3] = {0.f, 0.f, 0.f};
    if (valid) {
#pragma unroll
      for (int c = 0; c < 3; ++c) x[c] = g.x[cur][i * 3 + c];
      if (g.conv[i]) {
        const int o = g.orig[cur][i];
        g.p_out[o * 3] = x[0]; g.p_out[o * 3 + 1] = x[1]; g.p_out[o * 3 + 2] = x[2];
        g.conv_out[o] = 1;
      } else {
        keep = true;
        float gf[3], goff[3];
        pe_pullback(x, g.L_sdf, g.w_sdf, g.a0bar + i * g.ld_a0bar, g.skipbar ? g.skipbar + i * g.ld_skipbar : nullptr, g.n_skip, gf);
        pe_pullback(x, g.L_def, g.w_def, g.a0dbar + i * g.ld_a0dbar, nullptr, 0, goff);
        const float f = g.sdf_out[i * g.ld_sdf];
        const float sg = f > 0.f ? 1.f : (f < 0.f ? -1.f : 0.f);
        const float* t = g.t + i * 4;
        float gv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) gv[c] = g.w1 * sg * gf[c] + g.w2 * (t[c] + goff[c]);
        const float Lr = g.w1 * fabsf(f) + g.w2 * g.s[i];
        const float st = -Lr / (gv[0] * gv[0] + gv[1] * gv[1] + gv[2] * gv[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) x[c] += st * gv[c];
      }
    }
    const int slot = wave_compact_slot(keep, g.live + phase + 1);
    if (keep) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { g.x[nxt][slot * 3 + c] = x[c]; g.v[nxt][slot * 3 + c] = g.v[cur][i * 3 + c]; }
      frame = g.frame[cur][i];
      g.frame[nxt][slot] = frame;
      g.orig[nxt][slot] = g.orig[cur][i];
    }
    embed_rows(g, rows, keep, slot, x[0], x[1], x[2], frame);
  }
}

int check_args(const sr_refine_args* a) {
  if (!a || a->P < 0 || a->times < 0 || a->nframes <= 0 || a->L_sdf < 0 || a->L_sdf > 16 || a->L_def < 0 || a->L_def > 16 || a->E < 0) return SR_EINVAL;
  if (a->P == 0) return SR_OK;
  if (!a->live || !a->x[0] || !a->x[1] || !a->v[0] || !a->v[1] || !a->frame[0] || !a->frame[1] || !a->orig[0] || !a->orig[1]) return SR_EINVAL;
  if (!a->p_out || !a->conv_out || !a->cam || !a->A || !a->trans || !a->vol) return SR_EINVAL;
  if (a->ld_a0 < 3 + 6 * a->L_sdf || a->ld_a0d < 3 + 6 * a->L_def + a->E) return SR_EINVAL;
  return SR_OK;
}
}  // namespace

extern "C" {
int sr_refine_init(const sr_refine_args* a, void* stream) {
  const int rc = check_args(a);
  if (rc != SR_OK) return rc;
  if (!a->p0 || !a->rays || !a->batch_inds || !a->unit) { if (a->P > 0) return SR_EINVAL; }
  if ((a->a0 != nullptr) != (a->a0d != nullptr)) return SR_EINVAL;
  if (a->a0 && (!a->w_sdf || !a->w_def || (a->E > 0 && !a->conds))) return SR_EINVAL;
  const int64_t n = a->P > a->times + 3 ? a->P : a->times + 3;
  int64_t blocks = sr_cdiv(n, 256);                                  // one thread per ray (and per `live` entry) ...
  if (a->a0) blocks = std::max<int64_t>(blocks, sr_stream_grid((int64_t)a->P * (a->ld_a0 + a->ld_a0d), 256));   // ... striding the input elements
  hipLaunchKernelGGL(refine_init_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *a);
  return sr_launch_status();
}

int sr_refine_embed(const sr_refine_args* a, int32_t phase, void* stream) {
  const int rc = check_args(a);
  if (rc != SR_OK || a->P == 0) return rc;
  if (phase < 0 || phase > a->times + 1 || !a->a0 || !a->a0d || !a->w_sdf || !a->w_def || (a->E > 0 && !a->conds)) return SR_EINVAL;
  hipLaunchKernelGGL(refine_embed_kernel, dim3(sr_stream_grid((int64_t)a->P * (a->ld_a0 + a->ld_a0d), 256)), dim3(256), 0, (hipStream_t)stream, *a, phase);
  return sr_launch_status();
}

int sr_refine_mid(const sr_refine_args* a, int32_t phase, int32_t mode, void* stream) {
  const int rc = check_args(a);
  if (rc != SR_OK || a->P == 0) return rc;
  if (phase < 0 || phase > a->times + 1 || mode < 0 || mode > 2 || !a->sdf_out || !a->def_out) return SR_EINVAL;
  if (mode == 1 && (!a->conv || !a->t || !a->s)) return SR_EINVAL;
  if (mode == 0 && ((a->a0 != nullptr) != (a->a0d != nullptr) || (a->a0 && (!a->w_sdf || !a->w_def || (a->E > 0 && !a->conds))))) return SR_EINVAL;
  const dim3 grid(sr_stream_grid(a->P, 256)), block(256);
  if (mode == 0) hipLaunchKernelGGL(refine_mid_kernel<0>, dim3(sr_stream_grid((int64_t)a->P * (256 / kEnq), 256)), block, 0, (hipStream_t)stream, *a, phase);
  else if (mode == 1) hipLaunchKernelGGL(refine_mid_kernel<1>, grid, block, 0, (hipStream_t)stream, *a, phase);
  else hipLaunchKernelGGL(refine_mid_kernel<2>, grid, block, 0, (hipStream_t)stream, *a, phase);
  return sr_launch_status();
}

int sr_refine_finish(const sr_refine_args* a, int32_t phase, void* stream) {
  const int rc = check_args(a);
  if (rc != SR_OK || a->P == 0) return rc;
  if (phase < 1 || phase > a->times || !a->conv || !a->t || !a->s || !a->a0bar || !a->a0dbar || !a->sdf_out || !a->w_sdf || !a->w_def) return SR_EINVAL;
  if (a->ld_a0bar < 3 + 6 * a->L_sdf || a->ld_a0dbar < 3 + 6 * a->L_def || (a->skipbar && a->n_skip > a->ld_skipbar)) return SR_EINVAL;
  if ((a->a0 != nullptr) != (a->a0d != nullptr) || (a->E > 0 && a->a0 && !a->conds)) return SR_EINVAL;
  hipLaunchKernelGGL(refine_finish_kernel, dim3(sr_stream_grid((int64_t)a->P * (256 / kEnq), 256)), dim3(256), 0, (hipStream_t)stream, *a, phase);
  return sr_launch_status();
}
}
