// 3-D trilinear grid sampler with forward, backward and double-backward.
// Semantics follow MCAcc/cuda/GridSamplerMineKernel.cu:162-328 / 333-570 / 575-914: border padding,
// align_corners=False, border gradient rule of :44-60 (zero when the unnormalised coordinate is
// <= 0 or >= S-1), arbitrary strides.  The maths is re-derived here (see DESIGN.md "grid sampler"):
//   out_c   = sum_k w_k(u) I_c[k]                           u = clipped unnormalised coords
//   GI_c[k] = w_k gO_c ,  GG_a = s_a sum_c gO_c sum_k d_a w_k I_c[k]        s_a = (S_a/2) mult_a
//   double backward of (A . GI + B . GG):  dI_c[k] = gO_c tmp_k,  tmp_k = sum_a B_a s_a d_a w_k
//                                          dgO_c   = sum_k (A_c[k] w_k + I_c[k] tmp_k)
//                                          dg_b    = s_b sum_c gO_c sum_k (A_c[k] d_b w_k
//                                                      + I_c[k] sum_{a!=b} B_a s_a d_a d_b w_k)
// One thread owns one sample point and walks the channels; with the channel-last volume layout
// the host keeps for the skinning weights (stride[1]==1) the 8 corner reads of a point are 8
// contiguous 96-byte runs instead of 192 cache lines.  Gather-bound: HBM/L2 roofline.
#include "sr_common.h"
#include <limits.h>

namespace {

struct Corner {
  int x0, y0, z0;       // floor corner
  bool inb[8];          // in-bounds flag per corner, k = dz*4 + dy*2 + dx
};

template <typename T>
__device__ __forceinline__ T unnormalize(T g, int64_t S) {
  // ((g + 1) * S - 1) / 2 ; the reference's double literals promote only the "-1" and "/2"
  // steps (GridSamplerMineKernel.cu:210-212), which rounds identically to this.
  T t = (g + T(1)) * T(S);
  return T((double(t) - 1.0) / 2.0);
}

template <typename T>
__device__ __forceinline__ T clip_fwd(T t, int64_t S) {  // forward rule (:33-35): NaN -> 0
  T lo = (t > T(0)) ? t : T(0);
  T hi = T(S - 1);
  return lo < hi ? lo : hi;
}

template <typename T>
__device__ __forceinline__ T clip_grad(T t, int64_t S, T* mult) {  // backward rule (:44-60)
  if (t <= T(0)) { *mult = T(0); return T(0); }
  T hi = T(S - 1);
  if (t >= hi) { *mult = T(0); return hi; }
  *mult = T(1);
  return t;
}

template <typename T>
__device__ __forceinline__ T safe_int_range(T x) {  // :122-129
  if (x > T(INT_MAX - 1) || x < T(INT_MIN) || !isfinite((double)x)) return T(-100.0);
  return x;
}

template <typename T>
struct Weights {
  T wx[2], wy[2], wz[2];  // index 0: weight of the floor corner (x1 - x), 1: (x - x0)
};

template <typename T>
__device__ __forceinline__ void setup(T ix, T iy, T iz, int64_t W, int64_t H, int64_t D, Corner& c, Weights<T>& w) {
  c.x0 = (int)floor((double)ix);
  c.y0 = (int)floor((double)iy);
  c.z0 = (int)floor((double)iz);
  w.wx[0] = T(c.x0 + 1) - ix; w.wx[1] = ix - T(c.x0);
  w.wy[0] = T(c.y0 + 1) - iy; w.wy[1] = iy - T(c.y0);
  w.wz[0] = T(c.z0 + 1) - iz; w.wz[1] = iz - T(c.z0);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int x = c.x0 + (k & 1), y = c.y0 + ((k >> 1) & 1), z = c.z0 + (k >> 2);
    c.inb[k] = x >= 0 && x < W && y >= 0 && y < H && z >= 0 && z < D;
  }
}

struct PointIdx {
  int64_t n, d, h, w;
};
__device__ __forceinline__ PointIdx split(int64_t idx, const sr_tensor5& g) {
  PointIdx p;
  if ((g.size[1] | g.size[2]) == 1 && idx <= 0xffffffffll && g.size[3] <= 0xffffffffll) {   // the usual [N,1,1,P,3] grid: one u32 division
    const uint32_t n = (uint32_t)idx / (uint32_t)g.size[3];
    p.n = n; p.d = 0; p.h = 0; p.w = (uint32_t)idx - n * (uint32_t)g.size[3];
    return p;
  }
  p.w = idx % g.size[3];
  p.h = (idx / g.size[3]) % g.size[2];
  p.d = (idx / (g.size[2] * g.size[3])) % g.size[1];
  p.n = idx / (g.size[1] * g.size[2] * g.size[3]);
  return p;
}

template <typename T>
__global__ __launch_bounds__(256) void gs_fwd_kernel(int64_t total, const T* __restrict__ input, sr_tensor5 in_d,
                                                      const T* __restrict__ grid, sr_tensor5 g_d, T* __restrict__ out,
                                                      sr_tensor5 o_d) {
  const int64_t C = in_d.size[1], D = in_d.size[2], H = in_d.size[3], W = in_d.size[4];
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const PointIdx p = split(idx, g_d);
    const T* gp = grid + p.n * g_d.stride[0] + p.d * g_d.stride[1] + p.h * g_d.stride[2] + p.w * g_d.stride[3];
    T ix = safe_int_range(clip_fwd(unnormalize(gp[0], W), W));
    T iy = safe_int_range(clip_fwd(unnormalize(gp[g_d.stride[4]], H), H));
    T iz = safe_int_range(clip_fwd(unnormalize(gp[2 * g_d.stride[4]], D), D));
    Corner c; Weights<T> w;
    setup(ix, iy, iz, W, H, D, c, w);
    int64_t off[8]; T wk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      off[k] = (int64_t)(c.z0 + (k >> 2)) * in_d.stride[2] + (int64_t)(c.y0 + ((k >> 1) & 1)) * in_d.stride[3] +
               (int64_t)(c.x0 + (k & 1)) * in_d.stride[4];
      wk[k] = w.wx[k & 1] * w.wy[(k >> 1) & 1] * w.wz[k >> 2];
    }
    const T* ip = input + p.n * in_d.stride[0];
    T* op = out + p.n * o_d.stride[0] + p.d * o_d.stride[2] + p.h * o_d.stride[3] + p.w * o_d.stride[4];
    for (int64_t ch = 0; ch < C; ++ch, ip += in_d.stride[1], op += o_d.stride[1]) {
      T acc = T(0);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (c.inb[k]) acc += ip[off[k]] * wk[k];
      *op = acc;
    }
  }
}

template <typename T>
__device__ __forceinline__ void atomic_add(T* p, T v) { atomicAdd(p, v); }
// fp16 accumulation (the reference's gpuAtomicAdd on at::Half): CAS on the aligned 32-bit word that holds the element
template <>
__device__ __forceinline__ void atomic_add<_Float16>(_Float16* p, _Float16 v) {
  unsigned int* word = reinterpret_cast<unsigned int*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)3);
  const bool hi = (reinterpret_cast<uintptr_t>(p) & 2) != 0;
  unsigned int old = *word, assumed;
  do {
    assumed = old;
    const unsigned short bits = hi ? (unsigned short)(assumed >> 16) : (unsigned short)(assumed & 0xFFFFu);
    const _Float16 sum = __builtin_bit_cast(_Float16, bits) + v;
    const unsigned int sb = __builtin_bit_cast(unsigned short, sum);
    const unsigned int repl = hi ? ((assumed & 0x0000FFFFu) | (sb << 16)) : ((assumed & 0xFFFF0000u) | sb);
    old = atomicCAS(word, assumed, repl);
  } while (old != assumed);
}

template <typename T>
__global__ __launch_bounds__(256) void gs_bwd_kernel(int64_t total, const T* __restrict__ input, sr_tensor5 in_d,
                                                      const T* __restrict__ grid, sr_tensor5 g_d,
                                                      const T* __restrict__ gout, sr_tensor5 go_d, T* __restrict__ gin,
                                                      sr_tensor5 gi_d, T* __restrict__ ggrid) {
  const int64_t C = in_d.size[1], D = in_d.size[2], H = in_d.size[3], W = in_d.size[4];
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const PointIdx p = split(idx, g_d);
    const T* gp = grid + p.n * g_d.stride[0] + p.d * g_d.stride[1] + p.h * g_d.stride[2] + p.w * g_d.stride[3];
    T mx, my, mz;
    T ix = safe_int_range(clip_grad(unnormalize(gp[0], W), W, &mx));
    T iy = safe_int_range(clip_grad(unnormalize(gp[g_d.stride[4]], H), H, &my));
    T iz = safe_int_range(clip_grad(unnormalize(gp[2 * g_d.stride[4]], D), D, &mz));
    Corner c; Weights<T> w;
    setup(ix, iy, iz, W, H, D, c, w);
    int64_t off[8], goff[8]; T wk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t z = c.z0 + (k >> 2), y = c.y0 + ((k >> 1) & 1), x = c.x0 + (k & 1);
      off[k] = z * in_d.stride[2] + y * in_d.stride[3] + x * in_d.stride[4];
      goff[k] = z * gi_d.stride[2] + y * gi_d.stride[3] + x * gi_d.stride[4];
      wk[k] = w.wx[k & 1] * w.wy[(k >> 1) & 1] * w.wz[k >> 2];
    }
    const T* ip = input + p.n * in_d.stride[0];
    const T* gop = gout + p.n * go_d.stride[0] + p.d * go_d.stride[2] + p.h * go_d.stride[3] + p.w * go_d.stride[4];
    T* gip = gin ? gin + p.n * gi_d.stride[0] : nullptr;
    T gix = T(0), giy = T(0), giz = T(0);
    for (int64_t ch = 0; ch < C; ++ch, ip += in_d.stride[1], gop += go_d.stride[1]) {
      const T go = *gop;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (!c.inb[k]) continue;
        if (gip) atomic_add(gip + goff[k], wk[k] * go);
        const T v = ip[off[k]];
        const T sx = (k & 1) ? T(1) : T(-1), sy = ((k >> 1) & 1) ? T(1) : T(-1), sz = (k >> 2) ? T(1) : T(-1);
        gix += sx * (v * w.wy[(k >> 1) & 1] * w.wz[k >> 2] * go);
        giy += sy * (v * w.wx[k & 1] * w.wz[k >> 2] * go);
        giz += sz * (v * w.wx[k & 1] * w.wy[(k >> 1) & 1] * go);
      }
      if (gip) gip += gi_d.stride[1];
    }
    gix = T(double(gix * T(W)) / 2.0);
    giy = T(double(giy * T(H)) / 2.0);
    giz = T(double(giz * T(D)) / 2.0);
    T* gg = ggrid + idx * 3;
    gg[0] = mx * gix; gg[1] = my * giy; gg[2] = mz * giz;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gs_dbwd_kernel(int64_t total, const T* __restrict__ gOi, sr_tensor5 goi_d,
                                                       const T* __restrict__ gOg, sr_tensor5 gog_d,
                                                       const T* __restrict__ input, sr_tensor5 in_d,
                                                       const T* __restrict__ grid, sr_tensor5 g_d,
                                                       const T* __restrict__ gout, sr_tensor5 go_d, T* __restrict__ gin,
                                                       sr_tensor5 gi_d, T* __restrict__ ggrid, T* __restrict__ ggout) {
  const int64_t C = in_d.size[1], D = in_d.size[2], H = in_d.size[3], W = in_d.size[4];
  const int64_t npts = g_d.size[1] * g_d.size[2] * g_d.size[3];
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const PointIdx p = split(idx, g_d);
    const T* gp = grid + p.n * g_d.stride[0] + p.d * g_d.stride[1] + p.h * g_d.stride[2] + p.w * g_d.stride[3];
    T mx, my, mz;
    T ix = safe_int_range(clip_grad(unnormalize(gp[0], W), W, &mx));
    T iy = safe_int_range(clip_grad(unnormalize(gp[g_d.stride[4]], H), H, &my));
    T iz = safe_int_range(clip_grad(unnormalize(gp[2 * g_d.stride[4]], D), D, &mz));
    Corner c; Weights<T> w;
    setup(ix, iy, iz, W, H, D, c, w);
    const T* bp = gOg + p.n * gog_d.stride[0] + p.d * gog_d.stride[1] + p.h * gog_d.stride[2] + p.w * gog_d.stride[3];
    const T Bx = bp[0], By = bp[gog_d.stride[4]], Bz = bp[2 * gog_d.stride[4]];
    const T sX = T(0.5) * T(W) * mx, sY = T(0.5) * T(H) * my, sZ = T(0.5) * T(D) * mz;
    const T sXY = sX * sY, sXZ = sX * sZ, sYZ = sY * sZ;
    int64_t off[8], goff[8], aoff[8]; T wk[8], tmp[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
      const int64_t z = c.z0 + dz, y = c.y0 + dy, x = c.x0 + dx;
      off[k] = z * in_d.stride[2] + y * in_d.stride[3] + x * in_d.stride[4];
      goff[k] = z * gi_d.stride[2] + y * gi_d.stride[3] + x * gi_d.stride[4];
      aoff[k] = z * goi_d.stride[2] + y * goi_d.stride[3] + x * goi_d.stride[4];
      wk[k] = w.wx[dx] * w.wy[dy] * w.wz[dz];
      const T sx = dx ? T(1) : T(-1), sy = dy ? T(1) : T(-1), sz = dz ? T(1) : T(-1);
      tmp[k] = sx * (Bx * sX * w.wy[dy] * w.wz[dz]) + sy * (By * sY * w.wx[dx] * w.wz[dz]) + sz * (Bz * sZ * w.wx[dx] * w.wy[dy]);
    }
    const T* ip = input + p.n * in_d.stride[0];
    const T* ap = gOi ? gOi + p.n * goi_d.stride[0] : nullptr;
    const T* gop = gout + p.n * go_d.stride[0] + p.d * go_d.stride[2] + p.h * go_d.stride[3] + p.w * go_d.stride[4];
    T* gip = gin ? gin + p.n * gi_d.stride[0] : nullptr;
    // grad_grad_output is dense [N,C,Do,Ho,Wo]
    T* ggo = ggout + p.n * C * npts + (idx - p.n * npts);
    T gix = T(0), giy = T(0), giz = T(0);
    for (int64_t ch = 0; ch < C; ++ch, ip += in_d.stride[1], gop += go_d.stride[1], ggo += npts) {
      const T go = *gop;
      T acc = T(0);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (!c.inb[k]) continue;
        const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
        const T sx = dx ? T(1) : T(-1), sy = dy ? T(1) : T(-1), sz = dz ? T(1) : T(-1);
        if (gip) atomic_add(gip + goff[k], tmp[k] * go);
        if (ap) {
          const T a = ap[aoff[k]];
          gix += sx * (a * w.wy[dy] * w.wz[dz] * go * sX);
          giy += sy * (a * w.wx[dx] * w.wz[dz] * go * sY);
          giz += sz * (a * w.wx[dx] * w.wy[dy] * go * sZ);
          acc += a * wk[k];
        }
        const T v = ip[off[k]];
        // mixed second partials: d_x d_y w_k = sx sy wz, ...
        gix += v * (By * (sx * sy * w.wz[dz]) * sXY + Bz * (sx * sz * w.wy[dy]) * sXZ) * go;
        giy += v * (Bx * (sx * sy * w.wz[dz]) * sXY + Bz * (sy * sz * w.wx[dx]) * sYZ) * go;
        giz += v * (Bx * (sx * sz * w.wy[dy]) * sXZ + By * (sy * sz * w.wx[dx]) * sYZ) * go;
        acc += v * tmp[k];
      }
      *ggo = acc;
      if (gip) gip += gi_d.stride[1];
      if (ap) ap += goi_d.stride[1];
    }
    T* gg = ggrid + idx * 3;
    gg[0] = gix; gg[1] = giy; gg[2] = giz;
  }
}


// ---- channel-last fast path (fp32, input stride[1] == 1, C % 4 == 0): a corner's channels are one contiguous run, read
// as float4.  This is the layout the host keeps for the skinning-weight volume.  The loop is corner-major -- a corner's
// 16*CV-byte run is consumed in one go while its cache lines are hot, with 4*CV channel accumulators in registers -- and
// passes over the channels in slabs of 4*CV.  Per channel the corners are still summed in the order k = 0..7, so the
// forward value is bit-identical to the generic kernel.
typedef float gs_f32x4 __attribute__((ext_vector_type(4)));

template <int CV>
__global__ __launch_bounds__(256) void gs_fwd_cl_kernel(int64_t total, const float* __restrict__ input, sr_tensor5 in_d,
                                                         const float* __restrict__ grid, sr_tensor5 g_d, float* __restrict__ out,
                                                         sr_tensor5 o_d) {
  const int64_t C = in_d.size[1], D = in_d.size[2], H = in_d.size[3], W = in_d.size[4];
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const PointIdx p = split(idx, g_d);
    const float* gp = grid + p.n * g_d.stride[0] + p.d * g_d.stride[1] + p.h * g_d.stride[2] + p.w * g_d.stride[3];
    float ix = safe_int_range(clip_fwd(unnormalize(gp[0], W), W));
    float iy = safe_int_range(clip_fwd(unnormalize(gp[g_d.stride[4]], H), H));
    float iz = safe_int_range(clip_fwd(unnormalize(gp[2 * g_d.stride[4]], D), D));
    Corner c; Weights<float> w;
    setup(ix, iy, iz, W, H, D, c, w);
    const float* ip = input + p.n * in_d.stride[0];
    float* op = out + p.n * o_d.stride[0] + p.d * o_d.stride[2] + p.h * o_d.stride[3] + p.w * o_d.stride[4];
    for (int64_t ch = 0; ch < C; ch += 4 * CV) {
      gs_f32x4 acc[CV];
#pragma unroll
      for (int v = 0; v < CV; ++v) acc[v] = gs_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (!c.inb[k]) continue;
        const float wk = w.wx[k & 1] * w.wy[(k >> 1) & 1] * w.wz[k >> 2];
        const gs_f32x4* src = reinterpret_cast<const gs_f32x4*>(ip + (int64_t)(c.z0 + (k >> 2)) * in_d.stride[2] +
                                                                (int64_t)(c.y0 + ((k >> 1) & 1)) * in_d.stride[3] +
                                                                (int64_t)(c.x0 + (k & 1)) * in_d.stride[4] + ch);
#pragma unroll
        for (int v = 0; v < CV; ++v) acc[v] += src[v] * wk;
      }
#pragma unroll
      for (int v = 0; v < CV; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) op[(ch + 4 * v + e) * o_d.stride[1]] = acc[v][e];
    }
  }
}

// grad_grid only (the volume is a buffer): the gin == NULL case of gs_bwd_kernel.  Per corner the channel dot product
// d_k = sum_c I_c[k] gO_c is formed first, then GG_a = sum_k d_a w_k * d_k.
template <int CV>
__global__ __launch_bounds__(256) void gs_bwd_cl_kernel(int64_t total, const float* __restrict__ input, sr_tensor5 in_d,
                                                         const float* __restrict__ grid, sr_tensor5 g_d, const float* __restrict__ gout,
                                                         sr_tensor5 go_d, float* __restrict__ ggrid) {
  const int64_t C = in_d.size[1], D = in_d.size[2], H = in_d.size[3], W = in_d.size[4];
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
    const PointIdx p = split(idx, g_d);
    const float* gp = grid + p.n * g_d.stride[0] + p.d * g_d.stride[1] + p.h * g_d.stride[2] + p.w * g_d.stride[3];
    float mx, my, mz;
    float ix = safe_int_range(clip_grad(unnormalize(gp[0], W), W, &mx));
    float iy = safe_int_range(clip_grad(unnormalize(gp[g_d.stride[4]], H), H, &my));
    float iz = safe_int_range(clip_grad(unnormalize(gp[2 * g_d.stride[4]], D), D, &mz));
    Corner c; Weights<float> w;
    setup(ix, iy, iz, W, H, D, c, w);
    const float* ip = input + p.n * in_d.stride[0];
    const float* gop = gout + p.n * go_d.stride[0] + p.d * go_d.stride[2] + p.h * go_d.stride[3] + p.w * go_d.stride[4];
    float gix = 0.f, giy = 0.f, giz = 0.f;
    for (int64_t ch = 0; ch < C; ch += 4 * CV) {
      float go[4 * CV];
#pragma unroll
      for (int e = 0; e < 4 * CV; ++e) go[e] = gop[(ch + e) * go_d.stride[1]];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (!c.inb[k]) continue;
        const gs_f32x4* src = reinterpret_cast<const gs_f32x4*>(ip + (int64_t)(c.z0 + (k >> 2)) * in_d.stride[2] +
                                                                (int64_t)(c.y0 + ((k >> 1) & 1)) * in_d.stride[3] +
                                                                (int64_t)(c.x0 + (k & 1)) * in_d.stride[4] + ch);
        float dot = 0.f;
#pragma unroll
        for (int v = 0; v < CV; ++v) {
          const gs_f32x4 c4 = src[v];
#pragma unroll
          for (int e = 0; e < 4; ++e) dot += c4[e] * go[4 * v + e];
        }
        const float sx = (k & 1) ? 1.f : -1.f, sy = ((k >> 1) & 1) ? 1.f : -1.f, sz = (k >> 2) ? 1.f : -1.f;
        gix += sx * (w.wy[(k >> 1) & 1] * w.wz[k >> 2] * dot);
        giy += sy * (w.wx[k & 1] * w.wz[k >> 2] * dot);
        giz += sz * (w.wx[k & 1] * w.wy[(k >> 1) & 1] * dot);
      }
    }
    gix = (float)((double)(gix * (float)W) / 2.0);
    giy = (float)((double)(giy * (float)H) / 2.0);
    giz = (float)((double)(giz * (float)D) / 2.0);
    float* gg = ggrid + idx * 3;
    gg[0] = mx * gix; gg[1] = my * giy; gg[2] = mz * giz;
  }
}

inline int channel_slab(int64_t C) { return C % 24 == 0 ? 6 : C % 16 == 0 ? 4 : C % 8 == 0 ? 2 : 1; }

inline bool channel_last_ok(const void* input, const sr_tensor5& in_d) {
  return in_d.stride[1] == 1 && (in_d.size[1] & 3) == 0 && ((uintptr_t)input & 15) == 0 && (in_d.stride[0] & 3) == 0 &&
         (in_d.stride[2] & 3) == 0 && (in_d.stride[3] & 3) == 0 && (in_d.stride[4] & 3) == 0;
}

bool valid_desc(const sr_tensor5& in_d, const sr_tensor5& g_d) {
  if (in_d.size[0] != g_d.size[0] || g_d.size[4] != 3) return false;
  for (int i = 0; i < 5; ++i)
    if (in_d.size[i] < 0 || g_d.size[i] < 0) return false;
  for (int i = 2; i < 5; ++i)
    if (in_d.size[i] <= 0) return false;
  return true;
}

template <typename T>
int gs_fwd(const T* input, sr_tensor5 in_d, const T* grid, sr_tensor5 g_d, T* out, sr_tensor5 o_d, void* stream) {
  if (!valid_desc(in_d, g_d)) return SR_EINVAL;
  const int64_t total = g_d.size[0] * g_d.size[1] * g_d.size[2] * g_d.size[3];
  if (total == 0 || in_d.size[1] == 0) return SR_OK;
  if (!input || !grid || !out) return SR_EINVAL;
  if constexpr (sizeof(T) == 4) {
    if (channel_last_ok(input, in_d)) {
#define SR_GS_FWD_CL(CV)                                                                                                          \
  hipLaunchKernelGGL(gs_fwd_cl_kernel<CV>, dim3(sr_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, total, (const float*)input, \
                     in_d, (const float*)grid, g_d, (float*)out, o_d)
      switch (channel_slab(in_d.size[1])) {
        case 6: SR_GS_FWD_CL(6); break;
        case 4: SR_GS_FWD_CL(4); break;
        case 2: SR_GS_FWD_CL(2); break;
        default: SR_GS_FWD_CL(1); break;
      }
#undef SR_GS_FWD_CL
      return sr_launch_status();
    }
  }
  hipLaunchKernelGGL(gs_fwd_kernel<T>, dim3(sr_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, total, input, in_d, grid, g_d, out, o_d);
  return sr_launch_status();
}
template <typename T>
int gs_bwd(const T* input, sr_tensor5 in_d, const T* grid, sr_tensor5 g_d, const T* gout, sr_tensor5 go_d, T* gin,
           sr_tensor5 gi_d, T* ggrid, void* stream) {
  if (!valid_desc(in_d, g_d)) return SR_EINVAL;
  const int64_t total = g_d.size[0] * g_d.size[1] * g_d.size[2] * g_d.size[3];
  if (total == 0) return SR_OK;
  if (!input || !grid || !gout || !ggrid) return SR_EINVAL;
  if constexpr (sizeof(T) == 4) {
    if (!gin && channel_last_ok(input, in_d)) {
#define SR_GS_BWD_CL(CV)                                                                                                          \
  hipLaunchKernelGGL(gs_bwd_cl_kernel<CV>, dim3(sr_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, total, (const float*)input, \
                     in_d, (const float*)grid, g_d, (const float*)gout, go_d, (float*)ggrid)
      switch (channel_slab(in_d.size[1])) {
        case 6: SR_GS_BWD_CL(6); break;
        case 4: SR_GS_BWD_CL(4); break;
        case 2: SR_GS_BWD_CL(2); break;
        default: SR_GS_BWD_CL(1); break;
      }
#undef SR_GS_BWD_CL
      return sr_launch_status();
    }
  }
  hipLaunchKernelGGL(gs_bwd_kernel<T>, dim3(sr_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, total, input, in_d, grid, g_d, gout, go_d, gin, gi_d, ggrid);
  return sr_launch_status();
}
template <typename T>
int gs_dbwd(const T* gOi, sr_tensor5 goi_d, const T* gOg, sr_tensor5 gog_d, const T* input, sr_tensor5 in_d, const T* grid,
            sr_tensor5 g_d, const T* gout, sr_tensor5 go_d, T* gin, sr_tensor5 gi_d, T* ggrid, T* ggout, void* stream) {
  if (!valid_desc(in_d, g_d)) return SR_EINVAL;
  const int64_t total = g_d.size[0] * g_d.size[1] * g_d.size[2] * g_d.size[3];
  if (total == 0) return SR_OK;
  if (!gOg || !input || !grid || !gout || !ggrid || !ggout) return SR_EINVAL;
  hipLaunchKernelGGL(gs_dbwd_kernel<T>, dim3(sr_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, total, gOi, goi_d, gOg, gog_d, input, in_d, grid, g_d, gout, go_d, gin, gi_d, ggrid, ggout);
  return sr_launch_status();
}
}  // namespace

extern "C" {
// fp16 (the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF, GridSamplerMineKernel.cu:931,963,1001): the same templates
// on the native _Float16 type -- every +, -, * rounds to half exactly as at::Half arithmetic does (float op, one rounding).
int sr_gridsample3d_fwd_f16(const void* i, sr_tensor5 id, const void* g, sr_tensor5 gd, void* o, sr_tensor5 od, void* s) {
  return gs_fwd<_Float16>((const _Float16*)i, id, (const _Float16*)g, gd, (_Float16*)o, od, s);
}
int sr_gridsample3d_bwd_f16(const void* i, sr_tensor5 id, const void* g, sr_tensor5 gd, const void* go, sr_tensor5 god, void* gi, sr_tensor5 gid,
                            void* gg, void* s) {
  return gs_bwd<_Float16>((const _Float16*)i, id, (const _Float16*)g, gd, (const _Float16*)go, god, (_Float16*)gi, gid, (_Float16*)gg, s);
}
int sr_gridsample3d_dbwd_f16(const void* a, sr_tensor5 ad, const void* b, sr_tensor5 bd, const void* i, sr_tensor5 id, const void* g, sr_tensor5 gd,
                             const void* go, sr_tensor5 god, void* gi, sr_tensor5 gid, void* gg, void* ggo, void* s) {
  return gs_dbwd<_Float16>((const _Float16*)a, ad, (const _Float16*)b, bd, (const _Float16*)i, id, (const _Float16*)g, gd, (const _Float16*)go, god,
                           (_Float16*)gi, gid, (_Float16*)gg, (_Float16*)ggo, s);
}
int sr_gridsample3d_fwd_f32(const float* i, sr_tensor5 id, const float* g, sr_tensor5 gd, float* o, sr_tensor5 od, void* s) { return gs_fwd<float>(i, id, g, gd, o, od, s); }
int sr_gridsample3d_fwd_f64(const double* i, sr_tensor5 id, const double* g, sr_tensor5 gd, double* o, sr_tensor5 od, void* s) { return gs_fwd<double>(i, id, g, gd, o, od, s); }
int sr_gridsample3d_bwd_f32(const float* i, sr_tensor5 id, const float* g, sr_tensor5 gd, const float* go, sr_tensor5 god, float* gi, sr_tensor5 gid, float* gg, void* s) { return gs_bwd<float>(i, id, g, gd, go, god, gi, gid, gg, s); }
int sr_gridsample3d_bwd_f64(const double* i, sr_tensor5 id, const double* g, sr_tensor5 gd, const double* go, sr_tensor5 god, double* gi, sr_tensor5 gid, double* gg, void* s) { return gs_bwd<double>(i, id, g, gd, go, god, gi, gid, gg, s); }
int sr_gridsample3d_dbwd_f32(const float* a, sr_tensor5 ad, const float* b, sr_tensor5 bd, const float* i, sr_tensor5 id, const float* g, sr_tensor5 gd, const float* go, sr_tensor5 god, float* gi, sr_tensor5 gid, float* gg, float* ggo, void* s) { return gs_dbwd<float>(a, ad, b, bd, i, id, g, gd, go, god, gi, gid, gg, ggo, s); }
int sr_gridsample3d_dbwd_f64(const double* a, sr_tensor5 ad, const double* b, sr_tensor5 bd, const double* i, sr_tensor5 id, const double* g, sr_tensor5 gd, const double* go, sr_tensor5 god, double* gi, sr_tensor5 gid, double* gg, double* ggo, void* s) { return gs_dbwd<double>(a, ad, b, bd, i, id, g, gd, go, god, gi, gid, gg, ggo, s); }
}
