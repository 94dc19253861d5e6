// interp2x_boundary3d: 2x trilinear upsampling of a [B,C,d,h,w] volume to
// [B,C,2d-1,2h-1,2w-1] fused with the "parents disagree in sign" flag, and its adjoint.  Semantics follow
// MCAcc/cuda/interp2x_boundary3d_kernel.cu:11-151 / 155-239: a fine voxel is the mean of its 1/2/4/8 coarse parents
// (float sum in the reference's parent order, division in double), boundary = the parents' (v > balance) flags are
// not all equal.  One thread per output voxel; rows along x are contiguous, so reads and writes are coalesced
// dword streams (4 B in per coarse voxel, 5 B out per fine voxel: HBM bound).
#include "sr_common.h"

namespace {
// (x, y, z, bc) of a flat voxel index, advanced through the grid-stride loop by carry arithmetic: the 64-bit divisions are
// paid once per thread, not once per voxel.
struct Walker {
  int x, y, z; int64_t bc;
  int dx, dy, dz; int64_t dbc;
  int W, H, D;
  __device__ Walker(int64_t start, int64_t stride, int W_, int H_, int D_) : W(W_), H(H_), D(D_) {
    x = (int)(start % W); y = (int)((start / W) % H); z = (int)((start / ((int64_t)W * H)) % D); bc = start / ((int64_t)W * H * D);
    dx = (int)(stride % W); dy = (int)((stride / W) % H); dz = (int)((stride / ((int64_t)W * H)) % D); dbc = stride / ((int64_t)W * H * D);
  }
  __device__ __forceinline__ void advance() {
    x += dx; if (x >= W) { x -= W; ++y; }
    y += dy; if (y >= H) { y -= H; ++z; }
    z += dz; if (z >= D) { z -= D; ++bc; }
    bc += dbc;
  }
};

template <typename T>
__global__ __launch_bounds__(256) void interp2x_fwd_kernel(const T* __restrict__ in, int64_t BC, int d, int h, int w, float balance,
                                                            T* __restrict__ out, uint8_t* __restrict__ bnd) {
  const int D = 2 * d - 1, H = 2 * h - 1, W = 2 * w - 1;
  const int64_t total = BC * D * H * W;
  const int64_t start = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  Walker p(start, stride, W, H, D);
  for (int64_t i = start; i < total; i += stride, p.advance()) {
    const int x = p.x, y = p.y, z = p.z;
    const T* src = in + p.bc * d * h * w;
    const bool ox = x & 1, oy = y & 1, oz = z & 1;
    const int x0 = ox ? (x - 1) / 2 : x / 2, x1 = ox ? (x + 1) / 2 : x / 2;
    const int y0 = oy ? (y - 1) / 2 : y / 2, y1 = oy ? (y + 1) / 2 : y / 2;
    const int z0 = oz ? (z - 1) / 2 : z / 2, z1 = oz ? (z + 1) / 2 : z / 2;
    auto at = [&](int zz, int yy, int xx) { return src[((int64_t)zz * h + yy) * w + xx]; };
    T v[8];
    int n = 1;
    v[0] = at(z0, y0, x0);
    if (ox && oy && oz) {           // z outer, y, x inner (:119-126)
      n = 8;
      v[0] = at(z0, y0, x0); v[1] = at(z0, y0, x1); v[2] = at(z0, y1, x0); v[3] = at(z0, y1, x1);
      v[4] = at(z1, y0, x0); v[5] = at(z1, y0, x1); v[6] = at(z1, y1, x0); v[7] = at(z1, y1, x1);
    } else if (ox && oy) {          // skip_z: y outer, x inner (:71-74)
      n = 4; v[1] = at(z0, y0, x1); v[2] = at(z0, y1, x0); v[3] = at(z0, y1, x1);
    } else if (oy && oz) {          // skip_x: y outer, z inner (:86-89)
      n = 4; v[1] = at(z1, y0, x0); v[2] = at(z0, y1, x0); v[3] = at(z1, y1, x0);
    } else if (ox && oz) {          // skip_y: x outer, z inner (:102-105)
      n = 4; v[1] = at(z1, y0, x0); v[2] = at(z0, y0, x1); v[3] = at(z1, y0, x1);
    } else if (ox) { n = 2; v[1] = at(z0, y0, x1); }
    else if (oy) { n = 2; v[1] = at(z0, y1, x0); }
    else if (oz) { n = 2; v[1] = at(z1, y0, x0); }
    T s = v[0];
    bool disagree = false;
    const bool f0 = v[0] > (T)balance;
    for (int k = 1; k < n; ++k) { s += v[k]; disagree |= ((v[k] > (T)balance) != f0); }
    out[i] = s * (n == 1 ? T(1) : n == 2 ? T(0.5) : n == 4 ? T(0.25) : T(0.125));   // == (T)((double)s / n): n is a power of two
    bnd[i] = disagree ? 1 : 0;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void interp2x_bwd_kernel(const T* __restrict__ go, int64_t BC, int d, int h, int w, T* __restrict__ gi) {
  const int D = 2 * d - 1, H = 2 * h - 1, W = 2 * w - 1;
  const int64_t total = BC * d * h * w;
  const int64_t start = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (int64_t)gridDim.x * blockDim.x;
  Walker p(start, stride, w, h, d);
  for (int64_t i = start; i < total; i += stride, p.advance()) {
    const int x = p.x, y = p.y, z = p.z;
    const T* src = go + p.bc * D * H * W;
    auto at = [&](int zz, int yy, int xx) { return src[((int64_t)zz * H + yy) * W + xx]; };
    const bool xm = x > 0, xp = x < w - 1, ym = y > 0, yp = y < h - 1, zm = z > 0, zp = z < d - 1;
    const int X = 2 * x, Y = 2 * y, Z = 2 * z;
    T g = at(Z, Y, X);
    // (T)((double)g + (double)v / div): v / div is exact (div a power of two), one rounding from the double sum
    auto add = [&](bool ok, int zz, int yy, int xx, double div) { if (ok) g = (T)((double)g + (double)at(zz, yy, xx) * (1.0 / div)); };
    // the reference's accumulation order (:180-236): 6 edges, 12 faces (xy, xz, yz), 8 corners
    add(xm, Z, Y, X - 1, 2.0); add(xp, Z, Y, X + 1, 2.0); add(ym, Z, Y - 1, X, 2.0); add(yp, Z, Y + 1, X, 2.0);
    add(zm, Z - 1, Y, X, 2.0); add(zp, Z + 1, Y, X, 2.0);
    add(xm && ym, Z, Y - 1, X - 1, 4.0); add(xp && ym, Z, Y - 1, X + 1, 4.0); add(xm && yp, Z, Y + 1, X - 1, 4.0); add(xp && yp, Z, Y + 1, X + 1, 4.0);
    add(xm && zm, Z - 1, Y, X - 1, 4.0); add(xp && zm, Z - 1, Y, X + 1, 4.0); add(xm && zp, Z + 1, Y, X - 1, 4.0); add(xp && zp, Z + 1, Y, X + 1, 4.0);
    add(ym && zm, Z - 1, Y - 1, X, 4.0); add(yp && zm, Z - 1, Y + 1, X, 4.0); add(ym && zp, Z + 1, Y - 1, X, 4.0); add(yp && zp, Z + 1, Y + 1, X, 4.0);
    add(xm && ym && zm, Z - 1, Y - 1, X - 1, 8.0); add(xp && ym && zm, Z - 1, Y - 1, X + 1, 8.0);
    add(xm && yp && zm, Z - 1, Y + 1, X - 1, 8.0); add(xp && yp && zm, Z - 1, Y + 1, X + 1, 8.0);
    add(xm && ym && zp, Z + 1, Y - 1, X - 1, 8.0); add(xp && ym && zp, Z + 1, Y - 1, X + 1, 8.0);
    add(xm && yp && zp, Z + 1, Y + 1, X - 1, 8.0); add(xp && yp && zp, Z + 1, Y + 1, X + 1, 8.0);
    gi[i] = g;
  }
}

template <typename T>
int fwd(const T* in, int64_t BC, int d, int h, int w, float bal, T* out, uint8_t* bnd, void* st) {
  if (BC < 0 || d <= 0 || h <= 0 || w <= 0) return SR_EINVAL;
  if (BC == 0) return SR_OK;
  if (!in || !out || !bnd) return SR_EINVAL;
  const int64_t total = BC * (2 * d - 1) * (2 * h - 1) * (2 * w - 1);
  hipLaunchKernelGGL(interp2x_fwd_kernel<T>, dim3(sr_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)st, in, BC, d, h, w, bal, out, bnd);
  return sr_launch_status();
}
template <typename T>
int bwd(const T* go, int64_t BC, int d, int h, int w, T* gi, void* st) {
  if (BC < 0 || d <= 0 || h <= 0 || w <= 0) return SR_EINVAL;
  if (BC == 0) return SR_OK;
  if (!go || !gi) return SR_EINVAL;
  hipLaunchKernelGGL(interp2x_bwd_kernel<T>, dim3(sr_stream_grid(BC * d * h * w, 256)), dim3(256), 0, (hipStream_t)st, go, BC, d, h, w, gi);
  return sr_launch_status();
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// Seg3dLossless candidate selection (MCAcc/seg3d_lossless.py:296-312 of the reference: 3x3x3 box filter of the boundary
// flags > 0, minus the voxels evaluated at earlier levels, then nonzero): one pass over the two byte volumes, the surviving
// voxel indices are compacted by wave ballots (one atomicAdd per wave).  Output order is whatever the waves claim; the
// caller's queries and scatter do not depend on it (it may sort the list for a reproducible order).
namespace {
__global__ __launch_bounds__(256) void seg3d_candidates_kernel(const uint8_t* __restrict__ bnd, const uint8_t* __restrict__ done, int D, int H, int W,
                                                                int64_t* __restrict__ out, unsigned long long* __restrict__ count) {
  const int64_t total = (int64_t)D * H * W;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < total; base += stride) {
    const int64_t i = base + threadIdx.x;
    bool keep = false;
    if (i < total && !done[i]) {
      const int x = (int)(i % W), y = (int)((i / W) % H), z = (int)(i / ((int64_t)W * H));
      for (int dz = -1; dz <= 1 && !keep; ++dz) {
        const int zz = z + dz;
        if (zz < 0 || zz >= D) continue;
        for (int dy = -1; dy <= 1 && !keep; ++dy) {
          const int yy = y + dy;
          if (yy < 0 || yy >= H) continue;
          const uint8_t* row = bnd + ((int64_t)zz * H + yy) * W;
          keep = (x > 0 && row[x - 1]) || row[x] || (x + 1 < W && row[x + 1]);
        }
      }
    }
    const unsigned long long m = __ballot(keep);
    const int lane = threadIdx.x & 63;
    unsigned long long slot0 = 0;
    if (lane == 0 && m) slot0 = atomicAdd(count, (unsigned long long)__popcll(m));
    slot0 = __shfl(slot0, 0, 64);
    if (keep) out[slot0 + __popcll(m & ((1ull << lane) - 1ull))] = i;
  }
}
}  // namespace

extern "C" {
int sr_seg3d_candidates(const uint8_t* is_boundary, const uint8_t* done, int32_t D, int32_t H, int32_t W, int64_t* out_index, uint64_t* count_dev,
                        void* stream) {
  if (D <= 0 || H <= 0 || W <= 0 || !is_boundary || !done || !out_index || !count_dev) return SR_EINVAL;
  if (hipMemsetAsync(count_dev, 0, 8, (hipStream_t)stream) != hipSuccess) return SR_ELAUNCH;
  hipLaunchKernelGGL(seg3d_candidates_kernel, dim3(sr_stream_grid((int64_t)D * H * W, 256)), dim3(256), 0, (hipStream_t)stream, is_boundary, done, D, H, W,
                     out_index, (unsigned long long*)count_dev);
  return sr_launch_status();
}
int sr_interp2x3d_fwd_f32(const float* in, int64_t BC, int32_t d, int32_t h, int32_t w, float balance, float* out, uint8_t* is_boundary, void* s) { return fwd<float>(in, BC, d, h, w, balance, out, is_boundary, s); }
int sr_interp2x3d_fwd_f64(const double* in, int64_t BC, int32_t d, int32_t h, int32_t w, float balance, double* out, uint8_t* is_boundary, void* s) { return fwd<double>(in, BC, d, h, w, balance, out, is_boundary, s); }
int sr_interp2x3d_bwd_f32(const float* go, int64_t BC, int32_t d, int32_t h, int32_t w, float* gi, void* s) { return bwd<float>(go, BC, d, h, w, gi, s); }
int sr_interp2x3d_bwd_f64(const double* go, int64_t BC, int32_t d, int32_t h, int32_t w, double* gi, void* s) { return bwd<double>(go, BC, d, h, w, gi, s); }
}
