// Per-ray / per-vertex "tails" of one optimisation step, fused.
//
// Between the MLP evaluations the reference's step is a few hundred elementwise torch ops on [P,3] / [P,3,3]
// tensors (P <= 6144 rays, a few 10k template vertices): camera projection, per-pixel rays, cardinal rays through
// the inverse deformation Jacobian, colour / normal / eikonal / deformation-regulariser / mask-IoU reductions and
// the 3x3 normal equations of the implicit differentiation.  Each op is a launch of a few microseconds of device
// time and ~10-20 us of host issue time, and autograd doubles them in the backward.  Here every such block is ONE
// kernel for the value and ONE for the gradient.  All of them are latency-bound (KBs to a few MB of traffic), so
// the design rule is: coalesced row access, everything of a row in registers, deterministic reductions (fixed
// order: per-thread -> wave shuffle -> per-wave LDS slots -> per-block partial -> ordered final sum; no float
// atomics), and no workspace that must be zeroed by another launch.
//
// Reference semantics (file:line cited at each entry point in include/selfrecon_hip.h):
//   model/CameraMine.py:44-70,129-170,171-262   utils/utils.py:48-52,132-169   model/network.py:543-639,647-697,702-814
#include "sr_common.h"

namespace {
constexpr int kBlk = 256;
constexpr int kRedMaxFrames = SR_STEP_MAX_FRAMES;     // per-frame (numerator, denominator) pairs kept in registers
constexpr int kNV = 2 * kRedMaxFrames;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

// Sum NV per-thread values over the workgroup (<= 1024 threads); every thread returns with the totals in v.
template <int NV>
__device__ __forceinline__ void block_sum(float (&v)[NV], float* smem /* [NV * 16] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = wave_sum(v[i]);
    if (lane == 0) smem[i * 16 + wave] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += smem[i * 16 + w];
    v[i] = s;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------- 3x3 helpers
struct M3 { float m[9]; };
__device__ __forceinline__ M3 load9(const float* p) {
  M3 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.m[i] = p[i];
  return r;
}
// adjugate / det inverse with the reference's singularity rule (FastMinv/Matrix3x3InvKernels.cu: |det| < 1e-4 -> zeros, false)
__device__ __forceinline__ bool inv3(const M3& a, M3& o) {
  const float* m = a.m;
  const float c00 = m[4] * m[8] - m[5] * m[7], c01 = -m[3] * m[8] + m[5] * m[6], c02 = m[3] * m[7] - m[4] * m[6];
  const float c10 = -m[1] * m[8] + m[2] * m[7], c11 = m[0] * m[8] - m[2] * m[6], c12 = -m[0] * m[7] + m[1] * m[6];
  const float c20 = m[1] * m[5] - m[2] * m[4], c21 = -m[0] * m[5] + m[2] * m[3], c22 = m[0] * m[4] - m[1] * m[3];
  const float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  if (fabs((double)det) < 0.0001) {
#pragma unroll
    for (int i = 0; i < 9; ++i) o.m[i] = 0.f;
    return false;
  }
  o.m[0] = c00 / det; o.m[1] = c10 / det; o.m[2] = c20 / det;
  o.m[3] = c01 / det; o.m[4] = c11 / det; o.m[5] = c21 / det;
  o.m[6] = c02 / det; o.m[7] = c12 / det; o.m[8] = c22 / det;
  return true;
}
__device__ __forceinline__ void matvec(const M3& a, const float (&v)[3], float (&o)[3]) {      // o = A v
#pragma unroll
  for (int i = 0; i < 3; ++i) o[i] = a.m[i * 3] * v[0] + a.m[i * 3 + 1] * v[1] + a.m[i * 3 + 2] * v[2];
}
__device__ __forceinline__ void matTvec(const M3& a, const float (&v)[3], float (&o)[3]) {     // o = A^T v
#pragma unroll
  for (int j = 0; j < 3; ++j) o[j] = a.m[j] * v[0] + a.m[3 + j] * v[1] + a.m[6 + j] * v[2];
}
__device__ __forceinline__ float norm3(const float (&v)[3]) { return sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
// cotangent of v for n = v / |v| given the cotangent of n (zero vector -> zero, as torch's norm backward)
__device__ __forceinline__ void normalize_bwd(const float (&n)[3], float len, const float (&gn)[3], float (&gv)[3]) {
  const float d = n[0] * gn[0] + n[1] * gn[1] + n[2] * gn[2];
  const float inv = len > 0.f ? 1.f / len : 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) gv[i] = (gn[i] - n[i] * d) * inv;
}

// ---------------------------------------------------------------------------------------------- camera
struct Cam {
  float R[9], T[3], f[2], c[2], ax, ay, bx, by, hw, hh;
};
__device__ __forceinline__ Cam load_cam(const sr_camera& k) {
  Cam c;
#pragma unroll
  for (int i = 0; i < 9; ++i) c.R[i] = k.R[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) c.T[i] = k.T ? k.T[i] : 0.f;
  c.f[0] = k.f[0]; c.f[1] = k.f[1]; c.c[0] = k.c[0]; c.c[1] = k.c[1];
  c.hw = k.W * 0.5f; c.hh = k.H * 0.5f;
  c.ax = c.f[0] / c.hw; c.ay = c.f[1] / c.hh;
  c.bx = k.one_minus_inv_w - c.c[0] / c.hw; c.by = k.one_minus_inv_h - c.c[1] / c.hh;
  return c;
}

__global__ __launch_bounds__(kBlk) void project_ndc_fwd_kernel(const float* __restrict__ ps, int64_t n, sr_camera k,
                                                                float* __restrict__ xy, float* __restrict__ z) {
  const Cam c = load_cam(k);
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const float p0 = ps[i * 3], p1 = ps[i * 3 + 1], p2 = ps[i * 3 + 2];
    const float q0 = p0 * c.R[0] + p1 * c.R[3] + p2 * c.R[6] + c.T[0];
    const float q1 = p0 * c.R[1] + p1 * c.R[4] + p2 * c.R[7] + c.T[1];
    const float q2 = p0 * c.R[2] + p1 * c.R[5] + p2 * c.R[8] + c.T[2];
    xy[i * 2] = c.ax * q0 / q2 + c.bx;
    xy[i * 2 + 1] = c.ay * q1 / q2 + c.by;
    z[i] = q2;
  }
}

// gps (optional) and the 16 parameter-gradient partials of this workgroup: gR[9] | gT[3] | gf[2] | gc[2]
__global__ __launch_bounds__(kBlk) void project_ndc_bwd_kernel(const float* __restrict__ ps, int64_t n, sr_camera k,
                                                                const float* __restrict__ gxy, const float* __restrict__ gz,
                                                                float* __restrict__ gps, float* __restrict__ partial) {
  __shared__ float smem[16 * 16];
  const Cam c = load_cam(k);
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const float p[3] = {ps[i * 3], ps[i * 3 + 1], ps[i * 3 + 2]};
    const float q0 = p[0] * c.R[0] + p[1] * c.R[3] + p[2] * c.R[6] + c.T[0];
    const float q1 = p[0] * c.R[1] + p[1] * c.R[4] + p[2] * c.R[7] + c.T[1];
    const float q2 = p[0] * c.R[2] + p[1] * c.R[5] + p[2] * c.R[8] + c.T[2];
    const float gx = gxy ? gxy[i * 2] : 0.f, gy = gxy ? gxy[i * 2 + 1] : 0.f;
    const float iz = 1.f / q2;
    float gq[3];
    gq[0] = gx * c.ax * iz;
    gq[1] = gy * c.ay * iz;
    gq[2] = (gz ? gz[i] : 0.f) - (gx * c.ax * q0 + gy * c.ay * q1) * iz * iz;
    if (gps) {
#pragma unroll
      for (int a = 0; a < 3; ++a) gps[i * 3 + a] = c.R[a * 3] * gq[0] + c.R[a * 3 + 1] * gq[1] + c.R[a * 3 + 2] * gq[2];
    }
    if (partial) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) acc[a * 3 + b] += p[a] * gq[b];
      acc[9] += gq[0]; acc[10] += gq[1]; acc[11] += gq[2];
      acc[12] += gx * (q0 * iz) / c.hw; acc[13] += gy * (q1 * iz) / c.hh;
      acc[14] -= gx / c.hw; acc[15] -= gy / c.hh;
    }
  }
  if (partial) {
    block_sum<16>(acc, smem);
    if (threadIdx.x < 16) partial[blockIdx.x * 16 + threadIdx.x] = acc[threadIdx.x];
  }
}

__global__ __launch_bounds__(kBlk) void view_rays_fwd_kernel(const float* __restrict__ px, int64_t n, sr_camera k, float* __restrict__ rays) {
  const Cam c = load_cam(k);
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const float u = px[i * 3], w = px[i * 3 + 1], h = px[i * 3 + 2];
    float raw[3] = {-u / c.f[0] + h * c.c[0] / c.f[0], -w / c.f[1] + h * c.c[1] / c.f[1], h};
    const float len = norm3(raw);
    const float r[3] = {raw[0] / len, raw[1] / len, raw[2] / len};
#pragma unroll
    for (int a = 0; a < 3; ++a) rays[i * 3 + a] = r[0] * c.R[a * 3] + r[1] * c.R[a * 3 + 1] + r[2] * c.R[a * 3 + 2];
  }
}

// parameter gradients only (pixels are constants): gR[9] | - | gf[2] | gc[2] in the 16-slot layout of project_ndc_bwd
__global__ __launch_bounds__(kBlk) void view_rays_bwd_kernel(const float* __restrict__ px, int64_t n, sr_camera k,
                                                              const float* __restrict__ grays, float* __restrict__ partial) {
  __shared__ float smem[16 * 16];
  const Cam c = load_cam(k);
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const float u = px[i * 3], w = px[i * 3 + 1], h = px[i * 3 + 2];
    float raw[3] = {-u / c.f[0] + h * c.c[0] / c.f[0], -w / c.f[1] + h * c.c[1] / c.f[1], h};
    const float len = norm3(raw);
    const float r[3] = {raw[0] / len, raw[1] / len, raw[2] / len};
    const float go[3] = {grays[i * 3], grays[i * 3 + 1], grays[i * 3 + 2]};
    float gr[3], graw[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) gr[j] = go[0] * c.R[j] + go[1] * c.R[3 + j] + go[2] * c.R[6 + j];
    normalize_bwd(r, len, gr, graw);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[a * 3 + b] += go[a] * r[b];
    acc[12] -= graw[0] * raw[0] / c.f[0]; acc[13] -= graw[1] * raw[1] / c.f[1];
    acc[14] += graw[0] * h / c.f[0]; acc[15] += graw[1] * h / c.f[1];
  }
  block_sum<16>(acc, smem);
  if (threadIdx.x < 16) partial[blockIdx.x * 16 + threadIdx.x] = acc[threadIdx.x];
}

// out[j] (+)= sum_b partial[b][j], b ascending (deterministic)
__global__ __launch_bounds__(64) void sum_partials_kernel(const float* __restrict__ partial, int nblocks, int nv, float* __restrict__ out, int accumulate) {
  const int j = threadIdx.x;
  if (j >= nv) return;
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += partial[b * nv + j];
  out[j] = accumulate ? out[j] + s : s;
}

// ---------------------------------------------------------------------------------------------- cardinal rays / normals
__global__ __launch_bounds__(kBlk) void cardinal_rays_fwd_kernel(const float* __restrict__ J, const float* __restrict__ v, int64_t n,
                                                                  float* __restrict__ out, uint8_t* __restrict__ ok) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const M3 j = load9(J + i * 9);
    M3 a;
    const bool good = inv3(j, a);
    const float r[3] = {v[i * 3], v[i * 3 + 1], v[i * 3 + 2]};
    float u[3];
    if (good) matvec(a, r, u); else { u[0] = r[0]; u[1] = r[1]; u[2] = r[2]; }
    const float len = norm3(u);
#pragma unroll
    for (int k = 0; k < 3; ++k) out[i * 3 + k] = u[k] / len;
    ok[i] = good ? 1 : 0;
  }
}

__global__ __launch_bounds__(kBlk) void cardinal_rays_bwd_kernel(const float* __restrict__ J, const float* __restrict__ v, int64_t n,
                                                                  const float* __restrict__ gout, float* __restrict__ gJ, float* __restrict__ gv) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const M3 j = load9(J + i * 9);
    M3 a;
    const bool good = inv3(j, a);
    float w[3] = {0.f, 0.f, 0.f}, u[3] = {0.f, 0.f, 0.f};
    if (good) {
      const float r[3] = {v[i * 3], v[i * 3 + 1], v[i * 3 + 2]};
      matvec(a, r, u);
      const float len = norm3(u);
      const float c[3] = {u[0] / len, u[1] / len, u[2] / len};
      const float gc[3] = {gout[i * 3], gout[i * 3 + 1], gout[i * 3 + 2]};
      float gu[3];
      normalize_bwd(c, len, gc, gu);
      matTvec(a, gu, w);                       // = cotangent of v; gJ = -(A^T gu)(A v)^T
    }
    if (gJ) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int q = 0; q < 3; ++q) gJ[i * 9 + p * 3 + q] = -w[p] * u[q];
    }
    if (gv) {
#pragma unroll
      for (int p = 0; p < 3; ++p) gv[i * 3 + p] = w[p];
    }
  }
}

// unit deformed normal: normalize(J^-T n), J n for singular J (utils/utils.py:132-153, 'test' phase: no gradient)
__device__ __forceinline__ void deformed_normal(const M3& j, const float (&onx)[3], float (&nx)[3]) {
  M3 a;
  if (inv3(j, a)) matTvec(a, onx, nx); else matvec(j, onx, nx);
  const float len = norm3(nx);
  nx[0] /= len; nx[1] /= len; nx[2] /= len;
}
__global__ __launch_bounds__(kBlk) void deformed_normals_kernel(const float* __restrict__ J, const float* __restrict__ onx, int64_t n,
                                                                 float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const M3 j = load9(J + i * 9);
    const float o[3] = {onx[i * 3], onx[i * 3 + 1], onx[i * 3 + 2]};
    float nx[3];
    deformed_normal(j, o, nx);
    out[i * 3] = nx[0]; out[i * 3 + 1] = nx[1]; out[i * 3 + 2] = nx[2];
  }
}

// ---------------------------------------------------------------------------------------------- reductions to a loss
// Every loss below is  (1/N) sum_f  num_f / max(den_f, 1)   [mode 0: scatter-mean over frames, then mean]
//                 or   num_0 / den_0                          [mode 1: plain mean, den = row count]
//                 or   (1/N) sum_f (1 - num_f / den_f)        [mode 2: mask IoU]
// out[0] = loss, out[1 .. 1+2F) = the (num_f, den_f) totals (the backward needs den_f / both).
__device__ __forceinline__ void finish_loss(const float (&tot)[kNV], int mode, int N, float* out) {
  float loss = 0.f;
  if (mode == 1) {
    loss = tot[0] / tot[kRedMaxFrames];
  } else {
    for (int f = 0; f < N; ++f) {
      const float num = tot[f], den = tot[kRedMaxFrames + f];
      loss += mode == 0 ? num / fmaxf(den, 1.f) : 1.f - num / den;
    }
    loss /= (float)N;
  }
  out[0] = loss;
  for (int i = 0; i < kNV; ++i) out[1 + i] = tot[i];
}
template <class Row>
__device__ __forceinline__ void reduce_rows(int64_t n, int mode, int N, float* __restrict__ partial, float* __restrict__ out, Row row) {
  __shared__ float smem[kNV * 16];
  float acc[kNV];
#pragma unroll
  for (int i = 0; i < kNV; ++i) acc[i] = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int f = 0;
    float num = 0.f, den = 0.f;
    row(i, f, num, den);
#pragma unroll
    for (int k = 0; k < kRedMaxFrames; ++k) {
      acc[k] += k == f ? num : 0.f;
      acc[kRedMaxFrames + k] += k == f ? den : 0.f;
    }
  }
  block_sum<kNV>(acc, smem);
  if (gridDim.x == 1) {
    if (threadIdx.x == 0) finish_loss(acc, mode, N, out);
  } else if (threadIdx.x < kNV) {
    partial[blockIdx.x * kNV + threadIdx.x] = acc[threadIdx.x];
  }
}
__global__ __launch_bounds__(64) void finish_loss_kernel(const float* __restrict__ partial, int nblocks, int mode, int N, float* __restrict__ out) {
  __shared__ float tot_s[kNV];
  if (threadIdx.x < kNV) {
    float s = 0.f;
    for (int b = 0; b < nblocks; ++b) s += partial[b * kNV + threadIdx.x];
    tot_s[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot[kNV];
    for (int i = 0; i < kNV; ++i) tot[i] = tot_s[i];
    finish_loss(tot, mode, N, out);
  }
}

// colour term (model/network.py:611-618): |gt[b,r,c,:] - colour|_1 per ray
__global__ __launch_bounds__(1024) void color_loss_fwd_kernel(sr_ray_pixels px, const float* __restrict__ colors, const float* __restrict__ gt,
                                                               float* __restrict__ partial, float* __restrict__ out) {
  reduce_rows(px.P, 0, px.N, partial, out, [&](int64_t i, int& f, float& num, float& den) {
    f = (int)px.b[i];
    if (f < 0) return;                    // a masked row (frame index -1: a ray the refiner did not accept): no term, no count
    const float* g = gt + (((int64_t)f * px.H + px.r[i]) * px.W + px.c[i]) * 3;
    num = fabsf(g[0] - colors[i * 3]) + fabsf(g[1] - colors[i * 3 + 1]) + fabsf(g[2] - colors[i * 3 + 2]);
    den = 1.f;
  });
}
__global__ __launch_bounds__(kBlk) void color_loss_bwd_kernel(sr_ray_pixels px, const float* __restrict__ colors, const float* __restrict__ gt,
                                                               const float* __restrict__ saved, const float* __restrict__ gloss,
                                                               float* __restrict__ gcolors) {
  const float g0 = gloss[0] / (float)px.N;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < px.P; i += (int64_t)gridDim.x * kBlk) {
    const int f = (int)px.b[i];
    if (f < 0) {                          // masked row: exact zeros
      gcolors[i * 3] = 0.f; gcolors[i * 3 + 1] = 0.f; gcolors[i * 3 + 2] = 0.f;
      continue;
    }
    const float* g = gt + (((int64_t)f * px.H + px.r[i]) * px.W + px.c[i]) * 3;
    const float s = g0 / fmaxf(saved[1 + kRedMaxFrames + f], 1.f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float d = g[k] - colors[i * 3 + k];
      gcolors[i * 3 + k] = d > 0.f ? -s : (d < 0.f ? s : 0.f);
    }
  }
}

// normal term (model/network.py:620-639): ground-truth camera-space normal -> world (R diag(-1,1,-1)) -> canonical (J^T), compared
// with the unit SDF gradient; per-ray weight clamp(-v . n_deformed, 0, 1)^2 (detached) when `weighted`
struct NormalRow {
  bool valid;
  float gtn[3], nx[3], nlen, e[3], elen, w;
};
__device__ __forceinline__ NormalRow normal_row(const sr_ray_pixels& px, int64_t i, const float* __restrict__ nx_raw, const float* __restrict__ J,
                                                 const float* __restrict__ gtimg, const float* __restrict__ R, const float* __restrict__ rays,
                                                 int weighted) {
  NormalRow o;
  const int f = (int)px.b[i];
  if (f < 0) {                            // masked row (frame index -1): takes no part, receives exact zeros
    o.valid = false; o.nlen = 1.f; o.elen = 0.f; o.w = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { o.gtn[k] = 0.f; o.nx[k] = 0.f; o.e[k] = 0.f; }
    return o;
  }
  const float* g = gtimg + (((int64_t)f * px.H + px.r[i]) * px.W + px.c[i]) * 3;
  const float gc[3] = {-g[0], g[1], -g[2]};                       // the flip (network.py:626)
  float gw[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) gw[a] = R[a * 3] * gc[0] + R[a * 3 + 1] * gc[1] + R[a * 3 + 2] * gc[2];
  const float glen = norm3(gw);
  o.valid = glen > 0.0001f;
  if (o.valid) {
    const float d = fmaxf(glen, 1e-12f);
    gw[0] /= d; gw[1] /= d; gw[2] /= d;
  }
  o.gtn[0] = gw[0]; o.gtn[1] = gw[1]; o.gtn[2] = gw[2];
  const M3 j = load9(J + i * 9);
  float t[3];
  matTvec(j, gw, t);
  const float raw[3] = {nx_raw[i * 3], nx_raw[i * 3 + 1], nx_raw[i * 3 + 2]};
  o.nlen = norm3(raw);
#pragma unroll
  for (int k = 0; k < 3; ++k) { o.nx[k] = raw[k] / o.nlen; o.e[k] = t[k] - o.nx[k]; }
  o.elen = norm3(o.e);
  o.w = 1.f;
  if (weighted) {
    float cn[3];
    deformed_normal(j, raw, cn);
    const float d = -(rays[i * 3] * cn[0] + rays[i * 3 + 1] * cn[1] + rays[i * 3 + 2] * cn[2]);
    const float cl = fminf(fmaxf(d, 0.f), 1.f);
    o.w = cl * cl;
  }
  return o;
}
__global__ __launch_bounds__(1024) void normal_loss_fwd_kernel(sr_ray_pixels px, const float* __restrict__ nx_raw, const float* __restrict__ J,
                                                                const float* __restrict__ gtimg, const float* __restrict__ R,
                                                                const float* __restrict__ rays, int weighted, float* __restrict__ partial,
                                                                float* __restrict__ out) {
  reduce_rows(px.P, 0, px.N, partial, out, [&](int64_t i, int& f, float& num, float& den) {
    f = (int)px.b[i];
    const NormalRow r = normal_row(px, i, nx_raw, J, gtimg, R, rays, weighted);
    num = r.valid ? r.elen * r.w : 0.f;
    den = r.valid ? 1.f : 0.f;
  });
}
__global__ __launch_bounds__(kBlk) void normal_loss_bwd_kernel(sr_ray_pixels px, const float* __restrict__ nx_raw, const float* __restrict__ J,
                                                                const float* __restrict__ gtimg, const float* __restrict__ R,
                                                                const float* __restrict__ rays, int weighted, const float* __restrict__ saved,
                                                                const float* __restrict__ gloss, float* __restrict__ gnx_raw, float* __restrict__ gJ) {
  const float g0 = gloss[0] / (float)px.N;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < px.P; i += (int64_t)gridDim.x * kBlk) {
    const NormalRow r = normal_row(px, i, nx_raw, J, gtimg, R, rays, weighted);
    float ge[3] = {0.f, 0.f, 0.f};
    if (r.valid && r.elen > 0.f) {
      const float s = g0 * r.w / fmaxf(saved[1 + kRedMaxFrames + (int)px.b[i]], 1.f) / r.elen;
      ge[0] = s * r.e[0]; ge[1] = s * r.e[1]; ge[2] = s * r.e[2];
    }
    const float gn[3] = {-ge[0], -ge[1], -ge[2]};
    float graw[3];
    normalize_bwd(r.nx, r.nlen, gn, graw);
    gnx_raw[i * 3] = graw[0]; gnx_raw[i * 3 + 1] = graw[1]; gnx_raw[i * 3 + 2] = graw[2];
    if (gJ) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) gJ[i * 9 + a * 3 + b] = r.gtn[a] * ge[b];
    }
  }
}

// eikonal term (model/network.py:547-549): mean (|g| - 1)^2
__global__ __launch_bounds__(1024) void eikonal_fwd_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partial, float* __restrict__ out) {
  reduce_rows(n, 1, 1, partial, out, [&](int64_t i, int& f, float& num, float& den) {
    const float v[3] = {g[i * 3], g[i * 3 + 1], g[i * 3 + 2]};
    const float d = norm3(v) - 1.f;
    num = d * d; den = 1.f;
  });
}
__global__ __launch_bounds__(kBlk) void eikonal_bwd_kernel(const float* __restrict__ g, int64_t n, const float* __restrict__ gloss, float* __restrict__ gg) {
  const float s = gloss[0] * 2.f / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const float v[3] = {g[i * 3], g[i * 3 + 1], g[i * 3 + 2]};
    const float len = norm3(v);
    const float k = len > 0.f ? s * (len - 1.f) / len : 0.f;
    gg[i * 3] = k * v[0]; gg[i * 3 + 1] = k * v[1]; gg[i * 3 + 2] = k * v[2];
  }
}

// deformation regulariser (model/network.py:565-582, utils/utils.py:48-52): x = sum_k log(s_k)^2 of the singular values of the offset
// Jacobian, Geman-McClure 2 (x/c^2) / (x/c^2 + 4), mean
__global__ __launch_bounds__(1024) void def_regu_fwd_kernel(const float* __restrict__ S, int64_t n, float c, float* __restrict__ partial,
                                                             float* __restrict__ out) {
  reduce_rows(n, 1, 1, partial, out, [&](int64_t i, int& f, float& num, float& den) {
    const float l0 = logf(S[i * 3]), l1 = logf(S[i * 3 + 1]), l2 = logf(S[i * 3 + 2]);
    const float x = l0 * l0 + l1 * l1 + l2 * l2;
    const float y = x / (c * c);
    num = 2.f * y / (y + 4.f); den = 1.f;
  });
}
// gJ = U diag(gS) V^T with gS_k = gloss/n * dGM/dx * 2 log(s_k)/s_k
__global__ __launch_bounds__(kBlk) void def_regu_bwd_kernel(const float* __restrict__ U, const float* __restrict__ S, const float* __restrict__ V, int64_t n,
                                                             float c, const float* __restrict__ gloss, float* __restrict__ gJ) {
  const float g0 = gloss[0] / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const float s[3] = {S[i * 3], S[i * 3 + 1], S[i * 3 + 2]};
    const float l[3] = {logf(s[0]), logf(s[1]), logf(s[2])};
    const float x = l[0] * l[0] + l[1] * l[1] + l[2] * l[2];
    const float y = x / (c * c);
    const float dgm = 8.f / ((y + 4.f) * (y + 4.f)) / (c * c);
    const M3 u = load9(U + i * 9), v = load9(V + i * 9);
    float gs[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) gs[k] = g0 * dgm * 2.f * l[k] / s[k];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        gJ[i * 9 + a * 3 + b] = u.m[a * 3] * gs[0] * v.m[b * 3] + u.m[a * 3 + 1] * gs[1] * v.m[b * 3 + 1] + u.m[a * 3 + 2] * gs[2] * v.m[b * 3 + 2];
  }
}
// plain dS -> dJ of the singular values (ops.SingularValues3x3 backward)
__global__ __launch_bounds__(kBlk) void svd_bwd_kernel(const float* __restrict__ U, const float* __restrict__ V, const float* __restrict__ gS, int64_t n,
                                                        float* __restrict__ gJ) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const M3 u = load9(U + i * 9), v = load9(V + i * 9);
    const float gs[3] = {gS[i * 3], gS[i * 3 + 1], gS[i * 3 + 2]};
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        gJ[i * 9 + a * 3 + b] = u.m[a * 3] * gs[0] * v.m[b * 3] + u.m[a * 3 + 1] * gs[1] * v.m[b * 3 + 1] + u.m[a * 3 + 2] * gs[2] * v.m[b * 3 + 2];
  }
}

// mask IoU (model/network.py:652-654): per frame 1 - sum(m g) / sum |m + g - m g|
__global__ __launch_bounds__(1024) void mask_iou_fwd_kernel(const float* __restrict__ m, const float* __restrict__ g, int N, int64_t hw,
                                                             float* __restrict__ partial, float* __restrict__ out) {
  reduce_rows((int64_t)N * hw, 2, N, partial, out, [&](int64_t i, int& f, float& num, float& den) {
    f = (int)(i / hw);
    const float a = m[i], b = g[i];
    num = a * b; den = fabsf(a + b - a * b);
  });
}
__global__ __launch_bounds__(kBlk) void mask_iou_bwd_kernel(const float* __restrict__ m, const float* __restrict__ g, int N, int64_t hw,
                                                             const float* __restrict__ saved, const float* __restrict__ gloss, float* __restrict__ gm) {
  const float g0 = gloss[0] / (float)N;
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < (int64_t)N * hw; i += (int64_t)gridDim.x * kBlk) {
    const int f = (int)(i / hw);
    const float I = saved[1 + f], Un = saved[1 + kRedMaxFrames + f];
    const float a = m[i], b = g[i];
    const float u = a + b - a * b;
    const float su = u > 0.f ? 1.f : (u < 0.f ? -1.f : 0.f);
    // d(1 - I/U)/da = -(b U - I su (1 - b)) / U^2
    gm[i] = -g0 * (b * Un - I * su * (1.f - b)) / (Un * Un);
  }
}

// ---------------------------------------------------------------------------------------------- implicit differentiation
// model/network.py:702-771: b = [grad f ; [v]x J] (4x3), rhs = grad_l^T (b^T b)^-1 b^T (1x4); outputs -rhs[0], rhs[1:4], rhs[1:4] (-[v]x)
__global__ __launch_bounds__(kBlk) void implicit_solve_kernel(const float* __restrict__ gf, const float* __restrict__ J, const float* __restrict__ v,
                                                               const float* __restrict__ gl, int64_t n, float* __restrict__ cot_f,
                                                               float* __restrict__ rhs_tail, float* __restrict__ temp, uint8_t* __restrict__ ok) {
  for (int64_t i = (int64_t)blockIdx.x * kBlk + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlk) {
    const M3 j = load9(J + i * 9);
    const float vv[3] = {v[i * 3], v[i * 3 + 1], v[i * 3 + 2]};
    float b[4][3];
    b[0][0] = gf[i * 3]; b[0][1] = gf[i * 3 + 1]; b[0][2] = gf[i * 3 + 2];
#pragma unroll
    for (int q = 0; q < 3; ++q) {                 // [v]x J: row0 = -v2 J1 + v1 J2, row1 = v2 J0 - v0 J2, row2 = -v1 J0 + v0 J1
      b[1][q] = -vv[2] * j.m[3 + q] + vv[1] * j.m[6 + q];
      b[2][q] = vv[2] * j.m[q] - vv[0] * j.m[6 + q];
      b[3][q] = -vv[1] * j.m[q] + vv[0] * j.m[3 + q];
    }
    M3 btb, inv;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int q = 0; q < 3; ++q) btb.m[p * 3 + q] = b[0][p] * b[0][q] + b[1][p] * b[1][q] + b[2][p] * b[2][q] + b[3][p] * b[3][q];
    const bool good = inv3(btb, inv);
    const float l[3] = {gl[i * 3], gl[i * 3 + 1], gl[i * 3 + 2]};
    float y[3];                                   // grad_l^T inv
    matTvec(inv, l, y);
    float rhs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) rhs[r] = y[0] * b[r][0] + y[1] * b[r][1] + y[2] * b[r][2];
    cot_f[i] = -rhs[0];
    rhs_tail[i * 3] = rhs[1]; rhs_tail[i * 3 + 1] = rhs[2]; rhs_tail[i * 3 + 2] = rhs[3];
    // rhs[1:4] (-[v]x):  -[v]x = [[0, v2, -v1], [-v2, 0, v0], [v1, -v0, 0]]
    temp[i * 3] = -rhs[2] * vv[2] + rhs[3] * vv[1];
    temp[i * 3 + 1] = rhs[1] * vv[2] - rhs[3] * vv[0];
    temp[i * 3 + 2] = -rhs[1] * vv[1] + rhs[2] * vv[0];
    ok[i] = good ? 1 : 0;
  }
}

inline int red_blocks(int64_t n) {                // one workgroup of 1024 up to 32k rows (finish in-kernel), more for image-sized inputs
  if (n <= 32768) return 1;
  int64_t b = sr_cdiv(n, 1024 * 8);
  return (int)(b > 256 ? 256 : b);
}
inline bool bad_px(const sr_ray_pixels* px) {
  return !px || px->P < 0 || px->N < 1 || px->N > SR_STEP_MAX_FRAMES || px->H < 1 || px->W < 1 || (px->P > 0 && (!px->b || !px->r || !px->c));
}
}  // namespace

extern "C" {
int sr_step_reduce_blocks(int64_t rows) { return rows < 0 ? SR_EINVAL : red_blocks(rows); }
int sr_step_param_blocks(int64_t rows) { return rows < 0 ? SR_EINVAL : sr_stream_grid(rows, kBlk) > 128 ? 128 : sr_stream_grid(rows, kBlk); }

int sr_cam_project_ndc_fwd(const float* ps, int64_t n, const sr_camera* cam, float* xy, float* z, void* stream) {
  if (n < 0 || !cam || !cam->R || !cam->f || !cam->c) return SR_EINVAL;
  if (n == 0) return SR_OK;
  if (!ps || !xy || !z) return SR_EINVAL;
  hipLaunchKernelGGL(project_ndc_fwd_kernel, dim3(sr_stream_grid(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, ps, n, *cam, xy, z);
  return sr_launch_status();
}
int sr_cam_project_ndc_bwd(const float* ps, int64_t n, const sr_camera* cam, const float* gxy, const float* gz, float* gps, float* partial,
                           float* gparams, void* stream) {
  if (n < 0 || !cam || !cam->R || !cam->f || !cam->c || (partial && !gparams)) return SR_EINVAL;
  if (n == 0) {
    if (gparams) return hipMemsetAsync(gparams, 0, 16 * sizeof(float), (hipStream_t)stream) == hipSuccess ? SR_OK : SR_ELAUNCH;
    return SR_OK;
  }
  if (!ps || (!gxy && !gz)) return SR_EINVAL;
  const int blocks = partial ? sr_step_param_blocks(n) : sr_stream_grid(n, kBlk);
  hipLaunchKernelGGL(project_ndc_bwd_kernel, dim3(blocks), dim3(kBlk), 0, (hipStream_t)stream, ps, n, *cam, gxy, gz, gps, partial);
  if (partial) hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, blocks, 16, gparams, 0);
  return sr_launch_status();
}
int sr_cam_view_rays_fwd(const float* pixels, int64_t n, const sr_camera* cam, float* rays, void* stream) {
  if (n < 0 || !cam || !cam->R || !cam->f || !cam->c) return SR_EINVAL;
  if (n == 0) return SR_OK;
  if (!pixels || !rays) return SR_EINVAL;
  hipLaunchKernelGGL(view_rays_fwd_kernel, dim3(sr_stream_grid(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, pixels, n, *cam, rays);
  return sr_launch_status();
}
int sr_cam_view_rays_bwd(const float* pixels, int64_t n, const sr_camera* cam, const float* grays, float* partial, float* gparams, void* stream) {
  if (n < 0 || !cam || !cam->R || !cam->f || !cam->c || !partial || !gparams) return SR_EINVAL;
  if (n == 0) return hipMemsetAsync(gparams, 0, 16 * sizeof(float), (hipStream_t)stream) == hipSuccess ? SR_OK : SR_ELAUNCH;
  if (!pixels || !grays) return SR_EINVAL;
  const int blocks = sr_step_param_blocks(n);
  hipLaunchKernelGGL(view_rays_bwd_kernel, dim3(blocks), dim3(kBlk), 0, (hipStream_t)stream, pixels, n, *cam, grays, partial);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, blocks, 16, gparams, 0);
  return sr_launch_status();
}

int sr_cardinal_rays_fwd(const float* J, const float* v, int64_t n, float* out, uint8_t* ok, void* stream) {
  if (n < 0) return SR_EINVAL;
  if (n == 0) return SR_OK;
  if (!J || !v || !out || !ok) return SR_EINVAL;
  hipLaunchKernelGGL(cardinal_rays_fwd_kernel, dim3(sr_stream_grid(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, J, v, n, out, ok);
  return sr_launch_status();
}
int sr_cardinal_rays_bwd(const float* J, const float* v, int64_t n, const float* gout, float* gJ, float* gv, void* stream) {
  if (n < 0) return SR_EINVAL;
  if (n == 0) return SR_OK;
  if (!J || !v || !gout || (!gJ && !gv)) return SR_EINVAL;
  hipLaunchKernelGGL(cardinal_rays_bwd_kernel, dim3(sr_stream_grid(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, J, v, n, gout, gJ, gv);
  return sr_launch_status();
}
int sr_deformed_normals(const float* J, const float* onx, int64_t n, float* out, void* stream) {
  if (n < 0) return SR_EINVAL;
  if (n == 0) return SR_OK;
  if (!J || !onx || !out) return SR_EINVAL;
  hipLaunchKernelGGL(deformed_normals_kernel, dim3(sr_stream_grid(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, J, onx, n, out);
  return sr_launch_status();
}

int sr_color_loss_fwd(const sr_ray_pixels* px, const float* colors, const float* gt, float* partial, float* out, void* stream) {
  if (bad_px(px) || !out || px->P < 1 || !colors || !gt) return SR_EINVAL;
  const int blocks = red_blocks(px->P);
  if (blocks > 1 && !partial) return SR_EINVAL;
  hipLaunchKernelGGL(color_loss_fwd_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, *px, colors, gt, partial, out);
  if (blocks > 1) hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, blocks, 0, px->N, out);
  return sr_launch_status();
}
int sr_color_loss_bwd(const sr_ray_pixels* px, const float* colors, const float* gt, const float* saved, const float* gloss, float* gcolors,
                      void* stream) {
  if (bad_px(px) || px->P < 1 || !colors || !gt || !saved || !gloss || !gcolors) return SR_EINVAL;
  hipLaunchKernelGGL(color_loss_bwd_kernel, dim3(sr_stream_grid(px->P, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, *px, colors, gt, saved, gloss, gcolors);
  return sr_launch_status();
}
int sr_normal_loss_fwd(const sr_ray_pixels* px, const float* nx_raw, const float* J, const float* gt_normals, const float* R, const float* rays,
                       int weighted, float* partial, float* out, void* stream) {
  if (bad_px(px) || px->P < 1 || !nx_raw || !J || !gt_normals || !R || (weighted && !rays) || !out) return SR_EINVAL;
  const int blocks = red_blocks(px->P);
  if (blocks > 1 && !partial) return SR_EINVAL;
  hipLaunchKernelGGL(normal_loss_fwd_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, *px, nx_raw, J, gt_normals, R, rays, weighted, partial, out);
  if (blocks > 1) hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, blocks, 0, px->N, out);
  return sr_launch_status();
}
int sr_normal_loss_bwd(const sr_ray_pixels* px, const float* nx_raw, const float* J, const float* gt_normals, const float* R, const float* rays,
                       int weighted, const float* saved, const float* gloss, float* gnx_raw, float* gJ, void* stream) {
  if (bad_px(px) || px->P < 1 || !nx_raw || !J || !gt_normals || !R || (weighted && !rays) || !saved || !gloss || !gnx_raw) return SR_EINVAL;
  hipLaunchKernelGGL(normal_loss_bwd_kernel, dim3(sr_stream_grid(px->P, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, *px, nx_raw, J, gt_normals, R, rays,
                     weighted, saved, gloss, gnx_raw, gJ);
  return sr_launch_status();
}
int sr_eikonal_loss_fwd(const float* g, int64_t n, float* partial, float* out, void* stream) {
  if (n < 1 || !g || !out) return SR_EINVAL;
  const int blocks = red_blocks(n);
  if (blocks > 1 && !partial) return SR_EINVAL;
  hipLaunchKernelGGL(eikonal_fwd_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, g, n, partial, out);
  if (blocks > 1) hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, blocks, 1, 1, out);
  return sr_launch_status();
}
int sr_eikonal_loss_bwd(const float* g, int64_t n, const float* gloss, float* gg, void* stream) {
  if (n < 1 || !g || !gloss || !gg) return SR_EINVAL;
  hipLaunchKernelGGL(eikonal_bwd_kernel, dim3(sr_stream_grid(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, g, n, gloss, gg);
  return sr_launch_status();
}
int sr_def_regu_loss_fwd(const float* S, int64_t n, float c, float* partial, float* out, void* stream) {
  if (n < 1 || !S || !out || !(c > 0.f)) return SR_EINVAL;
  const int blocks = red_blocks(n);
  if (blocks > 1 && !partial) return SR_EINVAL;
  hipLaunchKernelGGL(def_regu_fwd_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, S, n, c, partial, out);
  if (blocks > 1) hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, blocks, 1, 1, out);
  return sr_launch_status();
}
int sr_def_regu_loss_bwd(const float* U, const float* S, const float* V, int64_t n, float c, const float* gloss, float* gJ, void* stream) {
  if (n < 1 || !U || !S || !V || !gloss || !gJ || !(c > 0.f)) return SR_EINVAL;
  hipLaunchKernelGGL(def_regu_bwd_kernel, dim3(sr_stream_grid(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, U, S, V, n, c, gloss, gJ);
  return sr_launch_status();
}
int sr_svd3x3_bwd(const float* U, const float* V, const float* gS, int64_t n, float* gJ, void* stream) {
  if (n < 0) return SR_EINVAL;
  if (n == 0) return SR_OK;
  if (!U || !V || !gS || !gJ) return SR_EINVAL;
  hipLaunchKernelGGL(svd_bwd_kernel, dim3(sr_stream_grid(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, U, V, gS, n, gJ);
  return sr_launch_status();
}
int sr_mask_iou_loss_fwd(const float* masks, const float* gt, int N, int64_t hw, float* partial, float* out, void* stream) {
  if (N < 1 || N > SR_STEP_MAX_FRAMES || hw < 1 || !masks || !gt || !out) return SR_EINVAL;
  const int blocks = red_blocks((int64_t)N * hw);
  if (blocks > 1 && !partial) return SR_EINVAL;
  hipLaunchKernelGGL(mask_iou_fwd_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, masks, gt, N, hw, partial, out);
  if (blocks > 1) hipLaunchKernelGGL(finish_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, partial, blocks, 2, N, out);
  return sr_launch_status();
}
int sr_mask_iou_loss_bwd(const float* masks, const float* gt, int N, int64_t hw, const float* saved, const float* gloss, float* gmasks, void* stream) {
  if (N < 1 || N > SR_STEP_MAX_FRAMES || hw < 1 || !masks || !gt || !saved || !gloss || !gmasks) return SR_EINVAL;
  hipLaunchKernelGGL(mask_iou_bwd_kernel, dim3(sr_stream_grid((int64_t)N * hw, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, masks, gt, N, hw, saved, gloss,
                     gmasks);
  return sr_launch_status();
}
int sr_implicit_solve(const float* grad_f, const float* J, const float* v, const float* grad_l, int64_t n, float* cot_f, float* rhs_tail, float* temp,
                      uint8_t* ok, void* stream) {
  if (n < 0) return SR_EINVAL;
  if (n == 0) return SR_OK;
  if (!grad_f || !J || !v || !grad_l || !cot_f || !rhs_tail || !temp || !ok) return SR_EINVAL;
  hipLaunchKernelGGL(implicit_solve_kernel, dim3(sr_stream_grid(n, kBlk)), dim3(kBlk), 0, (hipStream_t)stream, grad_f, J, v, grad_l, n, cot_f, rhs_tail,
                     temp, ok);
  return sr_launch_status();
}
}
