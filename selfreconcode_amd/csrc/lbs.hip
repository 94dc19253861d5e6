// Fused linear-blend skinning: per point, sample the 24-channel skinning
// weight volume (trilinear, border, align_corners=False -- the K3 semantics), blend the 24 posed
// joint transforms and apply them, optionally with the analytic 3x3 Jacobian dy/dp and its
// closed-form inverse (the "fused 3x3-inverse + LBS" of the north star: replaces
// GridSamplerMine.forward + per-frame matmul loop of model/Deformer.py:207-233, the three reverse
// passes of utils/utils.py:106-120 and FastMinv for the LBS part of the chain).
//
// Volume layout: channel-last [D,H,W,24] (the host keeps the reference's [1,24,D,H,W] tensor in
// channels_last_3d format), so a corner is one contiguous 96-byte run read as 6 float4.
// Gather-bound: 8 x 96 B per point from a 181 MB volume that sits in the 256 MB Infinity Cache.
#include "sr_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int NJ = 24;

struct Axis {
  float u;     // clipped unnormalised coordinate
  float du;    // d u / d p (0 on the border, K4 rule)
  int i0;
  float w0, w1;
};

__device__ __forceinline__ Axis make_axis(float p, float bmin, float bmax, int S) {
  Axis a;
  const float n = 2.f * (p - bmin) / (bmax - bmin) - 1.f;   // Deformer.py:207
  float t = (n + 1.f) * (float)S;
  t = (float)(((double)t - 1.0) / 2.0);                     // GridSamplerMineKernel.cu:210-212
  float mult = 1.f;
  if (t <= 0.f) { t = 0.f; mult = 0.f; }
  else if (t >= (float)(S - 1)) { t = (float)(S - 1); mult = 0.f; }
  if (!isfinite(t)) t = -100.f;
  a.u = t;
  a.du = mult * (float)S / (bmax - bmin);
  a.i0 = (int)floorf(t);
  a.w0 = (float)(a.i0 + 1) - t;
  a.w1 = t - (float)a.i0;
  return a;
}

template <bool WITH_JAC>
__global__ __launch_bounds__(256) void lbs_fwd_kernel(sr_lbs_args g) {
  __shared__ float sA[8 * NJ * 12];   // up to 8 frames of posed transforms staged per workgroup
  __shared__ float sT[8 * 3];
  const bool stage = g.nframes <= 8;
  if (stage) {
    for (int i = threadIdx.x; i < g.nframes * NJ * 12; i += blockDim.x) sA[i] = g.A[i];
    for (int i = threadIdx.x; i < g.nframes * 3; i += blockDim.x) sT[i] = g.trans[i];
    __syncthreads();
  }
  const int64_t sH = (int64_t)g.W * NJ, sD = (int64_t)g.H * g.W * NJ;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < g.P; idx += (int64_t)gridDim.x * blockDim.x) {
    const float px = g.p[idx * 3], py = g.p[idx * 3 + 1], pz = g.p[idx * 3 + 2];
    const float qx = g.tp ? g.tp[idx * 3] : px, qy = g.tp ? g.tp[idx * 3 + 1] : py, qz = g.tp ? g.tp[idx * 3 + 2] : pz;
    const Axis ax = make_axis(qx, g.bmin[0], g.bmax[0], g.W);
    const Axis ay = make_axis(qy, g.bmin[1], g.bmax[1], g.H);
    const Axis az = make_axis(qz, g.bmin[2], g.bmax[2], g.D);
    // sampled weights and their derivatives wrt the unnormalised coordinates
    float w[NJ], wx[NJ], wy[NJ], wz[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { w[j] = 0.f; wx[j] = 0.f; wy[j] = 0.f; wz[j] = 0.f; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
      const int x = ax.i0 + dx, y = ay.i0 + dy, z = az.i0 + dz;
      if (x < 0 || x >= g.W || y < 0 || y >= g.H || z < 0 || z >= g.D) continue;
      const float cx = dx ? ax.w1 : ax.w0, cy = dy ? ay.w1 : ay.w0, cz = dz ? az.w1 : az.w0;
      const float wk = cx * cy * cz;
      const float gx = (dx ? 1.f : -1.f) * cy * cz, gy = (dy ? 1.f : -1.f) * cx * cz, gz = (dz ? 1.f : -1.f) * cx * cy;
      const f32x4* src = reinterpret_cast<const f32x4*>(g.vol + z * sD + y * sH + (int64_t)x * NJ);
#pragma unroll
      for (int v = 0; v < NJ / 4; ++v) {
        const f32x4 c4 = src[v];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          w[4 * v + e] += c4[e] * wk;
          if (WITH_JAC) { wx[4 * v + e] += c4[e] * gx; wy[4 * v + e] += c4[e] * gy; wz[4 * v + e] += c4[e] * gz; }
        }
      }
    }
    const int frame = g.batch_inds ? (int)g.batch_inds[idx] : (int)(idx / g.points_per_frame);
    const float* Af = stage ? sA + frame * NJ * 12 : g.A + (int64_t)frame * NJ * 12;
    const float* tf = stage ? sT + frame * 3 : g.trans + frame * 3;
    float T[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) T[e] = 0.f;
    float Jx[3] = {0.f, 0.f, 0.f}, Jy[3] = {0.f, 0.f, 0.f}, Jz[3] = {0.f, 0.f, 0.f};  // columns of sum_j (A_j [p;1]) d w_j / d u
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const float* a = Af + j * 12;
#pragma unroll
      for (int e = 0; e < 12; ++e) T[e] += w[j] * a[e];
      if (WITH_JAC) {
        const float v0 = a[0] * px + a[1] * py + a[2] * pz + a[3];
        const float v1 = a[4] * px + a[5] * py + a[6] * pz + a[7];
        const float v2 = a[8] * px + a[9] * py + a[10] * pz + a[11];
        Jx[0] += v0 * wx[j]; Jx[1] += v1 * wx[j]; Jx[2] += v2 * wx[j];
        Jy[0] += v0 * wy[j]; Jy[1] += v1 * wy[j]; Jy[2] += v2 * wy[j];
        Jz[0] += v0 * wz[j]; Jz[1] += v1 * wz[j]; Jz[2] += v2 * wz[j];
      }
    }
    g.y[idx * 3 + 0] = T[0] * px + T[1] * py + T[2] * pz + T[3] + tf[0];
    g.y[idx * 3 + 1] = T[4] * px + T[5] * py + T[6] * pz + T[7] + tf[1];
    g.y[idx * 3 + 2] = T[8] * px + T[9] * py + T[10] * pz + T[11] + tf[2];
    if (WITH_JAC) {
      // dy_r/dp_c = T[r][c] + (sum_j v_j[r] dw_j/du_c) * du_c/dp_c   (weights looked up at tp == p)
      float J[9];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        J[r * 3 + 0] = T[r * 4 + 0] + Jx[r] * ax.du;
        J[r * 3 + 1] = T[r * 4 + 1] + Jy[r] * ay.du;
        J[r * 3 + 2] = T[r * 4 + 2] + Jz[r] * az.du;
      }
#pragma unroll
      for (int e = 0; e < 9; ++e) g.jac[idx * 9 + e] = J[e];
    }
  }
}

// Backward of y = LBS(p) for a cotangent ybar [P,3]:  pbar = J^T ybar (analytic Jacobian incl. the sampler term),
// Abar[frame][j] += w_j ybar (x) [p;1],  transbar[frame] += ybar.  Per-workgroup partial sums of Abar / transbar
// live in LDS and are flushed with one atomicAdd per entry (Guideline 12: reduce first, then one atomic per block).
__global__ __launch_bounds__(256) void lbs_bwd_kernel(sr_lbs_args g, const float* __restrict__ ybar, float* __restrict__ pbar,
                                                       float* __restrict__ Abar, float* __restrict__ tbar) {
  extern __shared__ float sacc[];                 // nframes * (24*12 + 3)
  const int per = NJ * 12 + 3;
  for (int i = threadIdx.x; i < g.nframes * per; i += blockDim.x) sacc[i] = 0.f;
  __syncthreads();
  const int64_t sH = (int64_t)g.W * NJ, sD = (int64_t)g.H * g.W * NJ;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < g.P; base += (int64_t)gridDim.x * blockDim.x) {
    const int64_t idx = base + threadIdx.x;
    if (idx < g.P) {
      const float px = g.p[idx * 3], py = g.p[idx * 3 + 1], pz = g.p[idx * 3 + 2];
      const float bx = ybar[idx * 3], by = ybar[idx * 3 + 1], bz = ybar[idx * 3 + 2];
      const Axis ax = make_axis(px, g.bmin[0], g.bmax[0], g.W);
      const Axis ay = make_axis(py, g.bmin[1], g.bmax[1], g.H);
      const Axis az = make_axis(pz, g.bmin[2], g.bmax[2], g.D);
      const int frame = g.batch_inds ? (int)g.batch_inds[idx] : (int)(idx / g.points_per_frame);
      const float* Af = g.A + (int64_t)frame * NJ * 12;
      float* acc = sacc + frame * per;
      float w[NJ], s[NJ];                           // weights and s_j = ybar . (A_j [p;1])
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float* a = Af + j * 12;
        w[j] = 0.f;
        s[j] = bx * (a[0] * px + a[1] * py + a[2] * pz + a[3]) + by * (a[4] * px + a[5] * py + a[6] * pz + a[7]) +
               bz * (a[8] * px + a[9] * py + a[10] * pz + a[11]);
      }
      float gu = 0.f, gv = 0.f, gw = 0.f;           // sum_j s_j dw_j/du_{x,y,z}
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
        const int x = ax.i0 + dx, y = ay.i0 + dy, z = az.i0 + dz;
        if (x < 0 || x >= g.W || y < 0 || y >= g.H || z < 0 || z >= g.D) continue;
        const float cx = dx ? ax.w1 : ax.w0, cy = dy ? ay.w1 : ay.w0, cz = dz ? az.w1 : az.w0;
        const float wk = cx * cy * cz;
        const f32x4* src = reinterpret_cast<const f32x4*>(g.vol + z * sD + y * sH + (int64_t)x * NJ);
        float dot = 0.f;
#pragma unroll
        for (int v = 0; v < NJ / 4; ++v) {
          const f32x4 c4 = src[v];
#pragma unroll
          for (int e = 0; e < 4; ++e) { w[4 * v + e] += c4[e] * wk; dot += c4[e] * s[4 * v + e]; }
        }
        gu += dot * (dx ? 1.f : -1.f) * cy * cz;
        gv += dot * (dy ? 1.f : -1.f) * cx * cz;
        gw += dot * (dz ? 1.f : -1.f) * cx * cy;
      }
      float T[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const float* a = Af + j * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) T[r * 3 + c] += w[j] * a[r * 4 + c];
        if (Abar) {
          const float wj = w[j];
          float* o = acc + j * 12;
          atomicAdd(o + 0, wj * bx * px); atomicAdd(o + 1, wj * bx * py); atomicAdd(o + 2, wj * bx * pz); atomicAdd(o + 3, wj * bx);
          atomicAdd(o + 4, wj * by * px); atomicAdd(o + 5, wj * by * py); atomicAdd(o + 6, wj * by * pz); atomicAdd(o + 7, wj * by);
          atomicAdd(o + 8, wj * bz * px); atomicAdd(o + 9, wj * bz * py); atomicAdd(o + 10, wj * bz * pz); atomicAdd(o + 11, wj * bz);
        }
      }
      if (tbar) { atomicAdd(acc + NJ * 12, bx); atomicAdd(acc + NJ * 12 + 1, by); atomicAdd(acc + NJ * 12 + 2, bz); }
      if (pbar) {
        pbar[idx * 3 + 0] = T[0] * bx + T[3] * by + T[6] * bz + gu * ax.du;
        pbar[idx * 3 + 1] = T[1] * bx + T[4] * by + T[7] * bz + gv * ay.du;
        pbar[idx * 3 + 2] = T[2] * bx + T[5] * by + T[8] * bz + gw * az.du;
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < g.nframes * per; i += blockDim.x) {
    const int f = i / per, e = i % per;
    const float v = sacc[i];
    if (v == 0.f) continue;
    if (e < NJ * 12) { if (Abar) atomicAdd(Abar + (int64_t)f * NJ * 12 + e, v); }
    else if (tbar) atomicAdd(tbar + f * 3 + (e - NJ * 12), v);
  }
}
}  // namespace

extern "C" int sr_lbs_fwd(const sr_lbs_args* a, void* stream) {
  if (!a || a->P < 0 || a->nframes <= 0 || a->D <= 0 || a->H <= 0 || a->W <= 0) return SR_EINVAL;
  if (a->P == 0) return SR_OK;
  if (!a->p || !a->A || !a->trans || !a->vol || !a->y || ((uintptr_t)a->vol & 15)) return SR_EINVAL;
  if (!a->batch_inds && a->points_per_frame <= 0) return SR_EINVAL;
  if (a->jac && a->tp) return SR_EINVAL;   // the analytic Jacobian assumes weights are looked up at p itself
  const int grid = sr_stream_grid(a->P, 256);
  if (a->jac) hipLaunchKernelGGL(lbs_fwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  else hipLaunchKernelGGL(lbs_fwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, *a);
  return sr_launch_status();
}

// Reverse of sr_lbs_fwd (weights looked up at p itself).  Abar [nframes,24,12] and transbar [nframes,3] must be
// zero-filled by the caller (they are accumulated); any of pbar / Abar / transbar may be NULL.
extern "C" int sr_lbs_bwd(const sr_lbs_args* a, const float* ybar, float* pbar, float* Abar, float* transbar, void* stream) {
  if (!a || a->P < 0 || a->nframes <= 0 || a->nframes > 32 || a->D <= 0 || a->H <= 0 || a->W <= 0) return SR_EINVAL;
  if (a->P == 0) return SR_OK;
  if (!a->p || !a->A || !a->vol || !ybar || a->tp || ((uintptr_t)a->vol & 15)) return SR_EINVAL;
  if (!a->batch_inds && a->points_per_frame <= 0) return SR_EINVAL;
  int grid = sr_stream_grid(a->P, 256);
  if (grid > 512) grid = 512;                      // fewer, fatter workgroups: fewer global atomics for Abar
  const size_t lds = (size_t)a->nframes * (NJ * 12 + 3) * sizeof(float);
  hipLaunchKernelGGL(lbs_bwd_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, *a, ybar, pbar, Abar, transbar);
  return sr_launch_status();
}
