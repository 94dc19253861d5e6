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
#include <type_traits>

#include "lbs_device.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
using srlbs::NJ; using srlbs::Axis; using srlbs::make_axis;

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
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < g.P; idx += (int64_t)gridDim.x * blockDim.x) {
    const float px = g.p[idx * 3], py = g.p[idx * 3 + 1], pz = g.p[idx * 3 + 2];
    const float qx = g.tp ? g.tp[idx * 3] : px, qy = g.tp ? g.tp[idx * 3 + 1] : py, qz = g.tp ? g.tp[idx * 3 + 2] : pz;
    const int frame = g.batch_inds ? (int)g.batch_inds[idx] : (int)(idx / g.points_per_frame);
    const float* Af = stage ? sA + frame * NJ * 12 : g.A + (int64_t)frame * NJ * 12;
    const float* tf = stage ? sT + frame * 3 : g.trans + frame * 3;
    float y[3], J[9];
    srlbs::lbs_point<WITH_JAC>(px, py, pz, qx, qy, qz, g.vol, g.D, g.H, g.W, g.bmin, g.bmax, Af, tf, y, J);
    g.y[idx * 3 + 0] = y[0]; g.y[idx * 3 + 1] = y[1]; g.y[idx * 3 + 2] = y[2];
    if (WITH_JAC) {
#pragma unroll
      for (int e = 0; e < 9; ++e) g.jac[idx * 9 + e] = J[e];
    }
  }
}

// Sum of v over the 64 lanes of a fully active wave: four DPP steps fold each row of 16 lanes, the four row totals are
// read back as scalars.  Every lane returns the same value.
__device__ __forceinline__ float wave_sum(float v) {
  auto dpp = [](float x, auto ctrl) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xf, 0xf, true));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});    // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});    // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
  const int b = __builtin_bit_cast(int, v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)) +
         __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
}

// Deterministic per-frame accumulation (both backward kernels).  A lane's contribution belongs to the frame of its point; the lanes of
// a wave are reduced frame by frame (one pass per distinct frame among the 64 lanes: one pass when the points come sorted by
// frame, as every caller of the training step passes them; at most `nframes` passes for any order) with the DPP butterfly, and lane
// 0 adds the wave's total into the WAVE'S OWN accumulator row in LDS -- plain read-modify-writes in program order, no atomics.
// The four rows of a workgroup are folded in a fixed order into a per-workgroup partial in global memory, and a second launch sums
// the partials of all workgroups in double precision, again in a fixed order.  The result is bit-reproducible run to run and
// independent of the schedule; its rounding error is that of a 64-lane butterfly plus one double-precision sum.
struct FramePass {
  unsigned long long todo;
  __device__ __forceinline__ FramePass() : todo(~0ull) {}
  // -> true while a frame is pending; sets f (wave-uniform) and mine (this lane belongs to it)
  __device__ __forceinline__ bool next(int frame, int& f, bool& mine) {
    if (!todo) return false;
    const int src = __builtin_amdgcn_readfirstlane((int)__ffsll((long long)todo) - 1);
    f = __builtin_amdgcn_readlane(frame, src);
    mine = frame == f;
    todo &= ~__ballot(mine);
    return true;
  }
};

// 16 outputs per workgroup, 16 lanes per output: lane l sums the partial rows l, l + 16, ... in double, then the 16 sums are folded
// in a fixed order (a fixed function of nblocks alone: bit-reproducible).  One thread per output (the first version) was 512
// dependent loads in a row: 65 us for 3.5 MB.
__global__ __launch_bounds__(256) void lbs_reduce_partials(const float* __restrict__ partials, int nblocks, int nframes,
                                                            float* __restrict__ Abar, float* __restrict__ tbar) {
  const int per = NJ * 12 + 3, n = nframes * per;
  const int o = threadIdx.x & 15, l = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + o;
  double s = 0.0;
  if (i < n)
    for (int b = l; b < nblocks; b += 16) s += (double)partials[(int64_t)b * n + i];
  __shared__ double fold[16][17];
  fold[l][o] = s;
  __syncthreads();
  if (l == 0 && i < n) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += fold[k][o];
    const int f = i / per, e = i % per;
    if (e < NJ * 12) { if (Abar) Abar[(int64_t)f * NJ * 12 + e] = (float)t; }
    else if (tbar) tbar[f * 3 + (e - NJ * 12)] = (float)t;
  }
}

// Backward of y = LBS(p) for a cotangent ybar [P,3]:  pbar = J^T ybar (analytic Jacobian incl. the sampler term),
// Abar[frame][j] = sum w_j ybar (x) [p;1],  transbar[frame] = sum ybar  (deterministic, see FramePass above).
__global__ __launch_bounds__(256) void lbs_bwd_kernel(sr_lbs_args g, const float* __restrict__ ybar, float* __restrict__ pbar,
                                                       bool want_A, bool want_t, float* __restrict__ partials) {
  extern __shared__ float sacc[];                 // (waves of the block) x nframes x (24*12 + 3)
  const int per = NJ * 12 + 3;
  const int nwaves = blockDim.x >> 6;
  for (int i = threadIdx.x; i < nwaves * g.nframes * per; i += blockDim.x) sacc[i] = 0.f;
  __syncthreads();
  float* wacc = sacc + (threadIdx.x >> 6) * g.nframes * per;      // this wave's accumulator rows
  const bool lane0 = (threadIdx.x & 63) == 0;
  const int64_t sH = (int64_t)g.W * NJ, sD = (int64_t)g.H * g.W * NJ;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < g.P; base += (int64_t)gridDim.x * blockDim.x) {
    const bool valid = base + threadIdx.x < g.P;                  // lanes past the end run along on the last point with a zero cotangent:
    const int64_t idx = valid ? base + threadIdx.x : g.P - 1;     // the wave-wide reductions below need all 64 lanes
    {
      const float px = g.p[idx * 3], py = g.p[idx * 3 + 1], pz = g.p[idx * 3 + 2];
      const float bx = valid ? ybar[idx * 3] : 0.f, by = valid ? ybar[idx * 3 + 1] : 0.f, bz = valid ? ybar[idx * 3 + 2] : 0.f;
      const Axis ax = make_axis(px, g.bmin[0], g.bmax[0], g.W);
      const Axis ay = make_axis(py, g.bmin[1], g.bmax[1], g.H);
      const Axis az = make_axis(pz, g.bmin[2], g.bmax[2], g.D);
      const int frame = g.batch_inds ? (int)g.batch_inds[idx] : (int)(idx / g.points_per_frame);
      const float* Af = g.A + (int64_t)frame * NJ * 12;
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
        if (want_A) {
          const float wj = w[j];
          const float c12[12] = {wj * bx * px, wj * bx * py, wj * bx * pz, wj * bx, wj * by * px, wj * by * py, wj * by * pz, wj * by,
                                 wj * bz * px, wj * bz * py, wj * bz * pz, wj * bz};
          FramePass fp; int f; bool mine;
          while (fp.next(frame, f, mine)) {
            float* o = wacc + f * per + j * 12;
#pragma unroll
            for (int e = 0; e < 12; ++e) {
              const float v = wave_sum(mine ? c12[e] : 0.f);
              if (lane0) o[e] += v;
            }
          }
        }
      }
      if (want_t) {
        FramePass fp; int f; bool mine;
        while (fp.next(frame, f, mine)) {
          const float sx = wave_sum(mine ? bx : 0.f), sy = wave_sum(mine ? by : 0.f), sz = wave_sum(mine ? bz : 0.f);
          if (lane0) { float* o = wacc + f * per + NJ * 12; o[0] += sx; o[1] += sy; o[2] += sz; }
        }
      }
      if (pbar && valid) {
        pbar[idx * 3 + 0] = T[0] * bx + T[3] * by + T[6] * bz + gu * ax.du;
        pbar[idx * 3 + 1] = T[1] * bx + T[4] * by + T[7] * bz + gv * ay.du;
        pbar[idx * 3 + 2] = T[2] * bx + T[5] * by + T[8] * bz + gw * az.du;
      }
    }
  }
  __syncthreads();
  const int n = g.nframes * per;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {               // the wave rows, in a fixed order
    float t = sacc[i];
    for (int wv = 1; wv < nwaves; ++wv) t += sacc[wv * n + i];
    partials[(int64_t)blockIdx.x * n + i] = t;
  }
}

// Backward of (y, J) = sr_lbs_fwd with its analytic Jacobian, for cotangents ybar [P,3] (nullable) and Jbar [P,3,3]:
//   y = sum_j w_j v_j + t,  J = sum_j w_j A_j[:, :3] + sum_j v_j (x) grad w_j,   v_j = A_j [q;1],  w_j = trilinear sample at q.
// With  e_j = <ybar, v_j> + <Jbar, A_j[:, :3]>  (cotangent of w_j)  and  gam_j = Jbar^T v_j  (cotangent of grad w_j):
//   qbar  = T3^T ybar + sum_j A_j3^T (Jbar grad w_j) + sum_j e_j grad w_j + sum_j H_j gam_j      (H_j: mixed second derivatives
//           of the trilinear interpolant, zero on the diagonal)
//   Abar_j = (w_j ybar + Jbar grad w_j) (x) [q;1] + w_j [Jbar | 0],   tbar = ybar.
// Two sweeps over the 8 corners (the second one hits the cache): the first rebuilds w_j and grad w_j, the joint loop between
// them emits Abar and replaces (w, grad w) by (e, gam) in the same registers, the second contracts the corners with (e, gam).
__global__ __launch_bounds__(256) void lbs_jac_bwd_kernel(sr_lbs_args g, const float* __restrict__ ybar, const float* __restrict__ Jbar,
                                                           float* __restrict__ pbar, bool want_A, bool want_t, float* __restrict__ partials) {
  extern __shared__ float sacc[];                 // (waves of the block) x nframes x (24*12 + 3)
  const int per = NJ * 12 + 3;
  const int nwaves = blockDim.x >> 6;
  for (int i = threadIdx.x; i < nwaves * g.nframes * per; i += blockDim.x) sacc[i] = 0.f;
  __syncthreads();
  float* wacc = sacc + (threadIdx.x >> 6) * g.nframes * per;
  const bool lane0 = (threadIdx.x & 63) == 0;
  const int64_t sH = (int64_t)g.W * NJ, sD = (int64_t)g.H * g.W * NJ;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < g.P; base += (int64_t)gridDim.x * blockDim.x) {
    const bool valid = base + threadIdx.x < g.P;                  // (see lbs_bwd_kernel)
    const int64_t idx = valid ? base + threadIdx.x : g.P - 1;
    {
      const float q[3] = {g.p[idx * 3], g.p[idx * 3 + 1], g.p[idx * 3 + 2]};
      float yb[3] = {0.f, 0.f, 0.f}, Jb[9];
      if (ybar && valid) { yb[0] = ybar[idx * 3]; yb[1] = ybar[idx * 3 + 1]; yb[2] = ybar[idx * 3 + 2]; }
#pragma unroll
      for (int e = 0; e < 9; ++e) Jb[e] = valid ? Jbar[idx * 9 + e] : 0.f;
      const Axis ax = make_axis(q[0], g.bmin[0], g.bmax[0], g.W);
      const Axis ay = make_axis(q[1], g.bmin[1], g.bmax[1], g.H);
      const Axis az = make_axis(q[2], g.bmin[2], g.bmax[2], g.D);
      const int frame = g.batch_inds ? (int)g.batch_inds[idx] : (int)(idx / g.points_per_frame);
      const float* Af = g.A + (int64_t)frame * NJ * 12;
      // The joints are processed in two halves of 12 (every sum over j is additive): half the coefficient registers, and each half
      // reads only its own 48 bytes of a corner's 96-byte run.
      float T[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, m[3] = {0.f, 0.f, 0.f};
      float gu = 0.f, gv = 0.f, gw_ = 0.f;             // sum_j e_j grad_u w_j
      float hxy_y = 0.f, hxy_x = 0.f, hxz_z = 0.f, hxz_x = 0.f, hyz_z = 0.f, hyz_y = 0.f;
      constexpr int HJ = NJ / 2;
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        const int j0 = half * HJ;
        // sweep 1: w_j and grad_u w_j
        float c0[HJ], c1[HJ], c2[HJ], c3[HJ];
#pragma unroll
        for (int j = 0; j < HJ; ++j) { c0[j] = 0.f; c1[j] = 0.f; c2[j] = 0.f; c3[j] = 0.f; }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
          const int x = ax.i0 + dx, y = ay.i0 + dy, z = az.i0 + dz;
          if (x < 0 || x >= g.W || y < 0 || y >= g.H || z < 0 || z >= g.D) continue;
          const float cx = dx ? ax.w1 : ax.w0, cy = dy ? ay.w1 : ay.w0, cz = dz ? az.w1 : az.w0;
          const float wk = cx * cy * cz, gx = (dx ? 1.f : -1.f) * cy * cz, gy = (dy ? 1.f : -1.f) * cx * cz, gz = (dz ? 1.f : -1.f) * cx * cy;
          const f32x4* src = reinterpret_cast<const f32x4*>(g.vol + z * sD + y * sH + (int64_t)x * NJ + j0);
#pragma unroll
          for (int v = 0; v < HJ / 4; ++v) {
            const f32x4 c4 = src[v];
#pragma unroll
            for (int e = 0; e < 4; ++e) { c0[4 * v + e] += c4[e] * wk; c1[4 * v + e] += c4[e] * gx; c2[4 * v + e] += c4[e] * gy; c3[4 * v + e] += c4[e] * gz; }
          }
        }
        // joint loop: T3, m = sum_j A_j3^T (Jbar grad w_j), Abar; then (w, grad w) -> (e, gam)
#pragma unroll
        for (int j = 0; j < HJ; ++j) {
          const float* a = Af + (j0 + j) * 12;
          const float wj = c0[j];
          const float gw[3] = {c1[j] * ax.du, c2[j] * ay.du, c3[j] * az.du};        // grad_q w_j
          float u[3], v3[3];
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            v3[r] = a[r * 4] * q[0] + a[r * 4 + 1] * q[1] + a[r * 4 + 2] * q[2] + a[r * 4 + 3];
            const float jg = Jb[r * 3] * gw[0] + Jb[r * 3 + 1] * gw[1] + Jb[r * 3 + 2] * gw[2];   // (Jbar grad w_j)_r
            u[r] = wj * yb[r] + jg;
#pragma unroll
            for (int c = 0; c < 3; ++c) { T[r * 3 + c] += wj * a[r * 4 + c]; m[c] += a[r * 4 + c] * jg; }
          }
          if (want_A) {
            float c12[12];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
              for (int c = 0; c < 3; ++c) c12[r * 4 + c] = u[r] * q[c] + wj * Jb[r * 3 + c];
              c12[r * 4 + 3] = u[r];
            }
            FramePass fp; int f; bool mine;
            while (fp.next(frame, f, mine)) {
              float* o = wacc + f * per + (j0 + j) * 12;
#pragma unroll
              for (int e = 0; e < 12; ++e) {
                const float v = wave_sum(mine ? c12[e] : 0.f);
                if (lane0) o[e] += v;
              }
            }
          }
          float ej = yb[0] * v3[0] + yb[1] * v3[1] + yb[2] * v3[2];
#pragma unroll
          for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) ej += Jb[r * 3 + c] * a[r * 4 + c];
          c0[j] = ej;
          c1[j] = Jb[0] * v3[0] + Jb[3] * v3[1] + Jb[6] * v3[2];                    // gam_j = Jbar^T v_j
          c2[j] = Jb[1] * v3[0] + Jb[4] * v3[1] + Jb[7] * v3[2];
          c3[j] = Jb[2] * v3[0] + Jb[5] * v3[1] + Jb[8] * v3[2];
        }
        if (pbar) {
          // sweep 2: contract the corners with (e, gam)
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
            const int x = ax.i0 + dx, y = ay.i0 + dy, z = az.i0 + dz;
            if (x < 0 || x >= g.W || y < 0 || y >= g.H || z < 0 || z >= g.D) continue;
            const float cx = dx ? ax.w1 : ax.w0, cy = dy ? ay.w1 : ay.w0, cz = dz ? az.w1 : az.w0;
            const float sx = dx ? 1.f : -1.f, sy = dy ? 1.f : -1.f, sz = dz ? 1.f : -1.f;
            const f32x4* src = reinterpret_cast<const f32x4*>(g.vol + z * sD + y * sH + (int64_t)x * NJ + j0);
            float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
#pragma unroll
            for (int v = 0; v < HJ / 4; ++v) {
              const f32x4 c4 = src[v];
#pragma unroll
              for (int e = 0; e < 4; ++e) { d0 += c4[e] * c0[4 * v + e]; d1 += c4[e] * c1[4 * v + e]; d2 += c4[e] * c2[4 * v + e]; d3 += c4[e] * c3[4 * v + e]; }
            }
            gu += d0 * sx * cy * cz; gv += d0 * cx * sy * cz; gw_ += d0 * cx * cy * sz;
            const float mxy = sx * sy * cz, mxz = sx * cy * sz, myz = cx * sy * sz;     // mixed second derivatives of the corner weight
            hxy_y += mxy * d2; hxy_x += mxy * d1; hxz_z += mxz * d3; hxz_x += mxz * d1; hyz_z += myz * d3; hyz_y += myz * d2;
          }
        }
      }
      if (want_t && ybar) {
        FramePass fp; int f; bool mine;
        while (fp.next(frame, f, mine)) {
          const float sx = wave_sum(mine ? yb[0] : 0.f), sy = wave_sum(mine ? yb[1] : 0.f), sz = wave_sum(mine ? yb[2] : 0.f);
          if (lane0) { float* o = wacc + f * per + NJ * 12; o[0] += sx; o[1] += sy; o[2] += sz; }
        }
      }
      if (pbar && valid) {
        pbar[idx * 3 + 0] = T[0] * yb[0] + T[3] * yb[1] + T[6] * yb[2] + m[0] + ax.du * (gu + ay.du * hxy_y + az.du * hxz_z);
        pbar[idx * 3 + 1] = T[1] * yb[0] + T[4] * yb[1] + T[7] * yb[2] + m[1] + ay.du * (gv + ax.du * hxy_x + az.du * hyz_z);
        pbar[idx * 3 + 2] = T[2] * yb[0] + T[5] * yb[1] + T[8] * yb[2] + m[2] + az.du * (gw_ + ax.du * hxz_x + ay.du * hyz_y);
      }
    }
  }
  __syncthreads();
  const int n = g.nframes * per;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {               // the wave rows, in a fixed order
    float t = sacc[i];
    for (int wv = 1; wv < nwaves; ++wv) t += sacc[wv * n + i];
    partials[(int64_t)blockIdx.x * n + i] = t;
  }
}
}  // namespace

static int lbs_bwd_grid(int64_t P) {
  int grid = sr_stream_grid(P, 256);
  return grid > 512 ? 512 : grid;                  // fewer, fatter workgroups: fewer partial rows to fold
}

// floats of the `partials` workspace of sr_lbs_bwd / sr_lbs_jac_bwd (one row of nframes x 291 sums per workgroup)
extern "C" int64_t sr_lbs_bwd_workspace_floats(int64_t P, int32_t nframes) {
  if (P < 0 || nframes <= 0) return SR_EINVAL;
  return (int64_t)(P > 0 ? lbs_bwd_grid(P) : 0) * nframes * (NJ * 12 + 3);
}

template <class K, class... Args>
static int lbs_bwd_launch(K kernel, const sr_lbs_args* a, float* Abar, float* transbar, float* partials, void* stream, Args... args) {
  const int grid = lbs_bwd_grid(a->P);
  // One accumulator row of nframes x 291 floats per wave of the block (1164 B per frame and wave).  Four waves per block up to 14
  // frames (<= 64 KB: no attribute call, and room for other streams' GEMM workgroups on the CU); beyond that -- no shipped
  // configuration: batches are 3 / 2 / 1 frames -- one wave per block (37 KB at the 32-frame limit).
  const int threads = (size_t)4 * a->nframes * (NJ * 12 + 3) * sizeof(float) <= 64 * 1024 ? 256 : 64;
  const size_t lds = (size_t)(threads / 64) * a->nframes * (NJ * 12 + 3) * sizeof(float);
  const bool want = Abar || transbar;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), lds, (hipStream_t)stream, *a, args..., Abar != nullptr, transbar != nullptr, partials);
  if (want)
    hipLaunchKernelGGL(lbs_reduce_partials, dim3(sr_cdiv(a->nframes * (NJ * 12 + 3), 16)), dim3(256), 0, (hipStream_t)stream, partials, grid, a->nframes,
                       Abar, transbar);
  return sr_launch_status();
}

extern "C" int sr_lbs_jac_bwd(const sr_lbs_args* a, const float* ybar, const float* Jbar, float* pbar, float* Abar, float* transbar, float* partials,
                              void* stream) {
  if (!a || a->P < 0 || a->nframes <= 0 || a->nframes > 32 || a->D <= 0 || a->H <= 0 || a->W <= 0) return SR_EINVAL;
  if (a->P == 0) {                                   // no points: the sums are zero
    if (Abar && hipMemsetAsync(Abar, 0, sizeof(float) * a->nframes * NJ * 12, (hipStream_t)stream) != hipSuccess) return SR_ELAUNCH;
    if (transbar && hipMemsetAsync(transbar, 0, sizeof(float) * a->nframes * 3, (hipStream_t)stream) != hipSuccess) return SR_ELAUNCH;
    return SR_OK;
  }
  if (!a->p || !a->A || !a->vol || !Jbar || a->tp || ((uintptr_t)a->vol & 15) || !partials) return SR_EINVAL;
  if (!a->batch_inds && a->points_per_frame <= 0) return SR_EINVAL;
  return lbs_bwd_launch(lbs_jac_bwd_kernel, a, Abar, transbar, partials, stream, ybar, Jbar, pbar);
}

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

// Reverse of sr_lbs_fwd (weights looked up at p itself).  Abar [nframes,24,12] and transbar [nframes,3] are WRITTEN (no zero fill
// needed); any of pbar / Abar / transbar may be NULL.  `partials`: sr_lbs_bwd_workspace_floats(P, nframes) floats of scratch.
extern "C" int sr_lbs_bwd(const sr_lbs_args* a, const float* ybar, float* pbar, float* Abar, float* transbar, float* partials, void* stream) {
  if (!a || a->P < 0 || a->nframes <= 0 || a->nframes > 32 || a->D <= 0 || a->H <= 0 || a->W <= 0) return SR_EINVAL;
  if (a->P == 0) {
    if (Abar && hipMemsetAsync(Abar, 0, sizeof(float) * a->nframes * NJ * 12, (hipStream_t)stream) != hipSuccess) return SR_ELAUNCH;
    if (transbar && hipMemsetAsync(transbar, 0, sizeof(float) * a->nframes * 3, (hipStream_t)stream) != hipSuccess) return SR_ELAUNCH;
    return SR_OK;
  }
  if (!a->p || !a->A || !a->vol || !ybar || a->tp || ((uintptr_t)a->vol & 15) || !partials) return SR_EINVAL;
  if (!a->batch_inds && a->points_per_frame <= 0) return SR_EINVAL;
  return lbs_bwd_launch(lbs_bwd_kernel, a, Abar, transbar, partials, stream, ybar, pbar);
}

// ------------------------------------------------------------------------------------------------
// SMPL kinematic chain (model/Deformer.py:175-203 / posedSkeleton :144-165): axis-angle -> rotation
// (half-angle quaternion, +1e-8 inside the norm, smpl_pytorch/util.py:35-78), G_i = G_parent [R_i | J_i - J_parent],
// A_i = G_i * init_pose_i.  The reference (and a straight torch restatement) issues ~80 tiny launches per call and
// ~160 more in backward; it is called ~8 times per iteration.  Here: one launch forward, one backward, one thread
// per frame.  The rotation derivative comes from forward-mode dual numbers (value + 3 partials), so the backward
// follows the exact same arithmetic as the forward.
namespace {
struct Dual {
  float v, d[3];
};
__device__ __forceinline__ Dual mk(float v) { return Dual{v, {0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) {
  return Dual{a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  const float iv = 1.f / b.v, q = a.v * iv;
  return Dual{q, {(a.d[0] - q * b.d[0]) * iv, (a.d[1] - q * b.d[1]) * iv, (a.d[2] - q * b.d[2]) * iv}};
}
__device__ __forceinline__ Dual dsqrt(Dual a) { const float s = sqrtf(a.v), h = 0.5f / s; return Dual{s, {a.d[0] * h, a.d[1] * h, a.d[2] * h}}; }
__device__ __forceinline__ Dual dsin(Dual a) { const float s = sinf(a.v), c = cosf(a.v); return Dual{s, {a.d[0] * c, a.d[1] * c, a.d[2] * c}}; }
__device__ __forceinline__ Dual dcos(Dual a) { const float s = sinf(a.v), c = cosf(a.v); return Dual{c, {-a.d[0] * s, -a.d[1] * s, -a.d[2] * s}}; }

template <typename S>
struct Ops;
template <>
struct Ops<float> {
  static __device__ __forceinline__ float c(float v) { return v; }
  static __device__ __forceinline__ float sq(float a) { return sqrtf(a); }
  static __device__ __forceinline__ float sn(float a) { return sinf(a); }
  static __device__ __forceinline__ float cs(float a) { return cosf(a); }
};
template <>
struct Ops<Dual> {
  static __device__ __forceinline__ Dual c(float v) { return mk(v); }
  static __device__ __forceinline__ Dual sq(Dual a) { return dsqrt(a); }
  static __device__ __forceinline__ Dual sn(Dual a) { return dsin(a); }
  static __device__ __forceinline__ Dual cs(Dual a) { return dcos(a); }
};

template <typename S>
__device__ __forceinline__ void rodrigues(S tx, S ty, S tz, S (&R)[9]) {
  using O = Ops<S>;
  const S e = O::c(1e-8f);
  const S ax = tx + e, ay = ty + e, az = tz + e;
  const S angle = O::sq(ax * ax + ay * ay + az * az);
  const S nx = tx / angle, ny = ty / angle, nz = tz / angle;
  const S half = angle * O::c(0.5f);
  const S qw0 = O::cs(half), sh = O::sn(half);
  const S qx0 = sh * nx, qy0 = sh * ny, qz0 = sh * nz;
  const S qn = O::sq(qw0 * qw0 + qx0 * qx0 + qy0 * qy0 + qz0 * qz0);
  const S w = qw0 / qn, x = qx0 / qn, y = qy0 / qn, z = qz0 / qn;
  const S w2 = w * w, x2 = x * x, y2 = y * y, z2 = z * z;
  const S wx = w * x, wy = w * y, wz = w * z, xy = x * y, xz = x * z, yz = y * z;
  const S two = O::c(2.f);
  R[0] = w2 + x2 - y2 - z2; R[1] = two * xy - two * wz; R[2] = two * wy + two * xz;
  R[3] = two * wz + two * xy; R[4] = w2 - x2 + y2 - z2; R[5] = two * yz - two * wx;
  R[6] = two * xz - two * wy; R[7] = two * wx + two * yz; R[8] = w2 - x2 - y2 + z2;
}

struct ChainConst {
  float rel[24 * 3];      // J_i - J_parent (J_0 for the root)
  float P[24 * 12];       // init_pose (inverse rest chain), rows 0..2 of each 4x4
  int parent[24];
};

// Per-pose working arrays (24 joints x 3x4) live in LDS, element-major: element e of thread t at sm[e * CHAIN_T + t].  (As
// per-thread arrays indexed by the runtime parent table they were scratch memory, and the two kernels, one dependent chain of
// global round trips per lane, took 77 / 166 us for three poses.)
constexpr int CHAIN_T = 32;    // poses per workgroup
struct LdsMat {                // view of one thread's [24*12] array
  float* base;
  __device__ __forceinline__ float& operator()(int joint, int e) const { return base[(joint * 12 + e) * CHAIN_T]; }
};

// G (3x4 per joint, row-major [R|t]) for one frame
__device__ __forceinline__ void chain_forward(const float* __restrict__ pose, const ChainConst& cc, const LdsMat& g) {
  for (int i = 0; i < 24; ++i) {
    float R[9];
    rodrigues<float>(pose[i * 3], pose[i * 3 + 1], pose[i * 3 + 2], R);
    const float* rel = cc.rel + i * 3;
    if (i == 0) {
      for (int r = 0; r < 3; ++r) { g(0, r * 4) = R[r * 3]; g(0, r * 4 + 1) = R[r * 3 + 1]; g(0, r * 4 + 2) = R[r * 3 + 2]; g(0, r * 4 + 3) = rel[r]; }
    } else {
      const int pa = cc.parent[i];
      for (int r = 0; r < 3; ++r) {
        const float p0 = g(pa, r * 4), p1 = g(pa, r * 4 + 1), p2 = g(pa, r * 4 + 2), p3 = g(pa, r * 4 + 3);
        for (int c = 0; c < 3; ++c) g(i, r * 4 + c) = p0 * R[c] + p1 * R[3 + c] + p2 * R[6 + c];
        g(i, r * 4 + 3) = p0 * rel[0] + p1 * rel[1] + p2 * rel[2] + p3;
      }
    }
  }
}

__global__ __launch_bounds__(CHAIN_T) void chain_fwd_kernel(const float* __restrict__ poses, int B, ChainConst cc, float* __restrict__ G,
                                                             float* __restrict__ A) {
  extern __shared__ float chain_sm[];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const LdsMat g{chain_sm + threadIdx.x};
  chain_forward(poses + (int64_t)b * 72, cc, g);
  for (int i = 0; i < 24; ++i) {
    const float* p = cc.P + i * 12;
    float* go = G + ((int64_t)b * 24 + i) * 16;
    float* ao = A + ((int64_t)b * 24 + i) * 16;
    for (int r = 0; r < 3; ++r) {
      const float g0 = g(i, r * 4), g1 = g(i, r * 4 + 1), g2 = g(i, r * 4 + 2), g3 = g(i, r * 4 + 3);
      go[r * 4] = g0; go[r * 4 + 1] = g1; go[r * 4 + 2] = g2; go[r * 4 + 3] = g3;
      for (int c = 0; c < 4; ++c) ao[r * 4 + c] = g0 * p[c] + g1 * p[4 + c] + g2 * p[8 + c] + (c == 3 ? g3 : 0.f);
    }
    go[12] = go[13] = go[14] = 0.f; go[15] = 1.f;
    ao[12] = ao[13] = ao[14] = 0.f; ao[15] = 1.f;
  }
}

// posebar[b,i,:] from the cotangents Abar, Gbar ([B,24,4,4], either nullable)
__global__ __launch_bounds__(CHAIN_T) void chain_bwd_kernel(const float* __restrict__ poses, int B, ChainConst cc, const float* __restrict__ Abar,
                                                             const float* __restrict__ Gbar, float* __restrict__ posebar) {
  extern __shared__ float chain_sm[];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const LdsMat g{chain_sm + threadIdx.x}, gb{chain_sm + 24 * 12 * CHAIN_T + threadIdx.x};
  const float* pose = poses + (int64_t)b * 72;
  chain_forward(pose, cc, g);
  for (int i = 0; i < 24; ++i) {
    const float* p = cc.P + i * 12;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 4; ++c) {
        float s = Gbar ? Gbar[((int64_t)b * 24 + i) * 16 + r * 4 + c] : 0.f;
        if (Abar) {
          const float* ab = Abar + ((int64_t)b * 24 + i) * 16 + r * 4;
          // A[r][k] = sum_c G[r][c] P[c][k]  (+ G[r][3] for k == 3): Gbar[r][c] += sum_k Abar[r][k] P[c][k]
          if (c < 3) s += ab[0] * p[c * 4] + ab[1] * p[c * 4 + 1] + ab[2] * p[c * 4 + 2] + ab[3] * p[c * 4 + 3];
          else s += ab[3];
        }
        gb(i, r * 4 + c) = s;
      }
  }
  for (int i = 23; i >= 0; --i) {
    float gbi[12];
    for (int e = 0; e < 12; ++e) gbi[e] = gb(i, e);
    float Rb[9];
    if (i == 0) {
      for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rb[r * 3 + c] = gbi[r * 4 + c];
    } else {
      const int pa = cc.parent[i];
      float R[9];
      rodrigues<float>(pose[i * 3], pose[i * 3 + 1], pose[i * 3 + 2], R);
      const float* rel = cc.rel + i * 3;
      // G_i = G_p L_i: Gbar_p += Gbar_i L_i^T (L = [R | rel; 0 1]);  Rbar = G_p[:, :3]^T Gbar_i[:, :3]
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c)
          gb(pa, r * 4 + c) += gbi[r * 4] * R[c * 3] + gbi[r * 4 + 1] * R[c * 3 + 1] + gbi[r * 4 + 2] * R[c * 3 + 2] + gbi[r * 4 + 3] * rel[c];
        gb(pa, r * 4 + 3) += gbi[r * 4 + 3];
      }
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rb[r * 3 + c] = g(pa, 0 * 4 + r) * gbi[0 * 4 + c] + g(pa, 1 * 4 + r) * gbi[1 * 4 + c] + g(pa, 2 * 4 + r) * gbi[2 * 4 + c];
    }
    Dual R[9];
    Dual tx{pose[i * 3], {1.f, 0.f, 0.f}}, ty{pose[i * 3 + 1], {0.f, 1.f, 0.f}}, tz{pose[i * 3 + 2], {0.f, 0.f, 1.f}};
    rodrigues<Dual>(tx, ty, tz, R);
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    for (int e = 0; e < 9; ++e) { o0 += Rb[e] * R[e].d[0]; o1 += Rb[e] * R[e].d[1]; o2 += Rb[e] * R[e].d[2]; }
    posebar[(int64_t)b * 72 + i * 3] = o0; posebar[(int64_t)b * 72 + i * 3 + 1] = o1; posebar[(int64_t)b * 72 + i * 3 + 2] = o2;
  }
}

int fill_chain_const(const float* host_Js, const int32_t* host_parents, const float* host_init_pose, ChainConst& cc) {
  for (int i = 0; i < 24; ++i) {
    const int pa = host_parents[i];
    if (i > 0 && (pa < 0 || pa >= i)) return SR_EINVAL;
    cc.parent[i] = pa;
    for (int c = 0; c < 3; ++c) cc.rel[i * 3 + c] = host_Js[i * 3 + c] - (i > 0 ? host_Js[pa * 3 + c] : 0.f);
    for (int e = 0; e < 12; ++e) cc.P[i * 12 + e] = host_init_pose[i * 16 + e];
  }
  return SR_OK;
}
}  // namespace

extern "C" int sr_lbs_chain_fwd(const float* poses, int32_t B, const float* host_Js, const int32_t* host_parents,
                                const float* host_init_pose, float* G, float* A, void* stream) {
  if (B < 0 || !host_Js || !host_parents || !host_init_pose) return SR_EINVAL;
  if (B == 0) return SR_OK;
  if (!poses || !G || !A) return SR_EINVAL;
  ChainConst cc;
  if (fill_chain_const(host_Js, host_parents, host_init_pose, cc) != SR_OK) return SR_EINVAL;
  hipLaunchKernelGGL(chain_fwd_kernel, dim3((B + CHAIN_T - 1) / CHAIN_T), dim3(CHAIN_T), 24 * 12 * CHAIN_T * sizeof(float), (hipStream_t)stream, poses, B, cc, G, A);
  return sr_launch_status();
}

extern "C" int sr_lbs_chain_bwd(const float* poses, int32_t B, const float* host_Js, const int32_t* host_parents,
                                const float* host_init_pose, const float* Abar, const float* Gbar, float* posebar, void* stream) {
  if (B < 0 || !host_Js || !host_parents || !host_init_pose) return SR_EINVAL;
  if (B == 0) return SR_OK;
  if (!poses || !posebar || (!Abar && !Gbar)) return SR_EINVAL;
  ChainConst cc;
  if (fill_chain_const(host_Js, host_parents, host_init_pose, cc) != SR_OK) return SR_EINVAL;
  hipLaunchKernelGGL(chain_bwd_kernel, dim3((B + CHAIN_T - 1) / CHAIN_T), dim3(CHAIN_T), 2 * 24 * 12 * CHAIN_T * sizeof(float), (hipStream_t)stream, poses, B, cc, Abar, Gbar,
                     posebar);
  return sr_launch_status();
}
