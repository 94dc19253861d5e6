// Per-point core of the fused linear-blend skinning (shared by lbs.hip and refiner.hip): sample the 24-channel skinning
// weight volume (trilinear, border, align_corners=False -- the K3 semantics of MCAcc/cuda/GridSamplerMineKernel.cu:162-328),
// blend the 24 posed joint transforms and apply them (model/Deformer.py:207-233), optionally with the analytic dy/dp.
#pragma once
#include "sr_common.h"

typedef float sr_f32x4 __attribute__((ext_vector_type(4)));

namespace srlbs {
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

// y = LBS(p) with the weights looked up at q (q == p except for the [tps, ps] call form); Af: the frame's 24 x 12 posed
// transforms, tf: its translation.  J (row major 3x3) = dy/dp for q == p.
template <bool WITH_JAC>
__device__ __forceinline__ void lbs_point(float px, float py, float pz, float qx, float qy, float qz, const float* __restrict__ vol, int D, int H,
                                          int W, const float* bmin, const float* bmax, const float* __restrict__ Af,
                                          const float* __restrict__ tf, float (&y)[3], float (&J)[9]) {
  const int64_t sH = (int64_t)W * NJ, sD = (int64_t)H * W * NJ;
  const Axis ax = make_axis(qx, bmin[0], bmax[0], W);
  const Axis ay = make_axis(qy, bmin[1], bmax[1], H);
  const Axis az = make_axis(qz, bmin[2], bmax[2], D);
  // sampled weights and their derivatives wrt the unnormalised coordinates
  float w[NJ], wx[NJ], wy[NJ], wz[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { w[j] = 0.f; wx[j] = 0.f; wy[j] = 0.f; wz[j] = 0.f; }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
    const int x = ax.i0 + dx, yy = ay.i0 + dy, z = az.i0 + dz;
    if (x < 0 || x >= W || yy < 0 || yy >= H || z < 0 || z >= D) continue;
    const float cx = dx ? ax.w1 : ax.w0, cy = dy ? ay.w1 : ay.w0, cz = dz ? az.w1 : az.w0;
    const float wk = cx * cy * cz;
    const float gx = (dx ? 1.f : -1.f) * cy * cz, gy = (dy ? 1.f : -1.f) * cx * cz, gz = (dz ? 1.f : -1.f) * cx * cy;
    const sr_f32x4* src = reinterpret_cast<const sr_f32x4*>(vol + z * sD + yy * sH + (int64_t)x * NJ);
#pragma unroll
    for (int v = 0; v < NJ / 4; ++v) {
      const sr_f32x4 c4 = src[v];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        w[4 * v + e] += c4[e] * wk;
        if (WITH_JAC) { wx[4 * v + e] += c4[e] * gx; wy[4 * v + e] += c4[e] * gy; wz[4 * v + e] += c4[e] * gz; }
      }
    }
  }
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
  y[0] = T[0] * px + T[1] * py + T[2] * pz + T[3] + tf[0];
  y[1] = T[4] * px + T[5] * py + T[6] * pz + T[7] + tf[1];
  y[2] = T[8] * px + T[9] * py + T[10] * pz + T[11] + tf[2];
  if (WITH_JAC) {
    // dy_r/dp_c = T[r][c] + (sum_j v_j[r] dw_j/du_c) * du_c/dp_c   (weights looked up at tp == p)
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      J[r * 3 + 0] = T[r * 4 + 0] + Jx[r] * ax.du;
      J[r * 3 + 1] = T[r * 4 + 1] + Jy[r] * ay.du;
      J[r * 3 + 2] = T[r * 4 + 2] + Jz[r] * az.du;
    }
  }
}
}  // namespace srlbs
