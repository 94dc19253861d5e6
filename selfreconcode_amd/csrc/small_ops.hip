// Small per-element kernels around the MLPs.
//  * sr_svd3x3: batched 3x3 SVD (one thread per matrix, cyclic Jacobi on J^T J) -- removes the
//    device -> host -> device round trip of model/network.py:576 (`torch.svd(Jacobs.cpu())`) from the
//    deformation regulariser (pulled forward because a 60k-matrix CPU SVD
//    would dominate the iteration).  Singular values descending, like torch.svd.
//  * sr_splat_fwd / sr_splat_bwd: order-independent soft point-splat silhouette
//    (stand-in for pytorch3d PointsRasterizer + AlphaCompositor with unit features: the composite
//    1 - prod_k (1 - a_k), a_k = 1 - d_k^2 / r^2, is commutative, so it is accumulated as a sum of
//    logs with atomics instead of a per-pixel z-sorted top-50 list).
#include "sr_common.h"

namespace {
__device__ __forceinline__ void jacobi_rotate(float (&S)[3][3], float (&V)[3][3], int p, int q) {
  if (fabsf(S[p][q]) < 1e-30f) return;
  const float theta = (S[q][q] - S[p][p]) / (2.0f * S[p][q]);
  const float t = (theta >= 0.f ? 1.f : -1.f) / (fabsf(theta) + sqrtf(theta * theta + 1.0f));
  const float c = 1.0f / sqrtf(t * t + 1.0f), s = t * c;
  const int r = 3 - p - q;
  const float spp = S[p][p], sqq = S[q][q], spq = S[p][q], srp = S[r][p], srq = S[r][q];
  S[p][p] = spp - t * spq; S[q][q] = sqq + t * spq; S[p][q] = S[q][p] = 0.f;
  S[r][p] = S[p][r] = c * srp - s * srq;
  S[r][q] = S[q][r] = s * srp + c * srq;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float vp = V[k][p], vq = V[k][q];
    V[k][p] = c * vp - s * vq;
    V[k][q] = s * vp + c * vq;
  }
}

__global__ __launch_bounds__(256) void svd3_kernel(const float* __restrict__ A, int64_t n, float* __restrict__ U, float* __restrict__ Sg,
                                                    float* __restrict__ Vo) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float a[3][3], S[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) a[r][c] = A[i * 9 + r * 3 + c];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) S[r][c] = a[0][r] * a[0][c] + a[1][r] * a[1][c] + a[2][r] * a[2][c];
    for (int sweep = 0; sweep < 6; ++sweep) {
      jacobi_rotate(S, V, 0, 1);
      jacobi_rotate(S, V, 0, 2);
      jacobi_rotate(S, V, 1, 2);
    }
    float ev[3] = {S[0][0], S[1][1], S[2][2]};
    int o[3] = {0, 1, 2};
#define SR_SWAP(x, y) if (ev[o[x]] < ev[o[y]]) { int tmp = o[x]; o[x] = o[y]; o[y] = tmp; }
    SR_SWAP(0, 1) SR_SWAP(0, 2) SR_SWAP(1, 2)
#undef SR_SWAP
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = o[k];
      const float sv = sqrtf(fmaxf(ev[c], 0.f));
      Sg[i * 3 + k] = sv;
      // u_k = A v_k / s_k
      float u0 = a[0][0] * V[0][c] + a[0][1] * V[1][c] + a[0][2] * V[2][c];
      float u1 = a[1][0] * V[0][c] + a[1][1] * V[1][c] + a[1][2] * V[2][c];
      float u2 = a[2][0] * V[0][c] + a[2][1] * V[1][c] + a[2][2] * V[2][c];
      const float inv = sv > 1e-20f ? 1.0f / sv : 0.f;
      U[i * 9 + 0 * 3 + k] = u0 * inv; U[i * 9 + 1 * 3 + k] = u1 * inv; U[i * 9 + 2 * 3 + k] = u2 * inv;
      Vo[i * 9 + 0 * 3 + k] = V[0][c]; Vo[i * 9 + 1 * 3 + k] = V[1][c]; Vo[i * 9 + 2 * 3 + k] = V[2][c];
    }
  }
}

// ---- soft point-splat silhouette ----------------------------------------------------------------
// pix: [N, V, 2] pixel coordinates (x = column, y = row) of the projected points, vis: [N,V] u8 (in front of
// the camera); logT: [N,H,W] zero-filled accumulator of sum_k log(1 - a_k).
__global__ __launch_bounds__(256) void splat_fwd_kernel(const float* __restrict__ pix, const uint8_t* __restrict__ vis, int64_t npts,
                                                         int64_t pts_per_img, int H, int W, float r, float* __restrict__ logT) {
  const float r2 = r * r;
  const int ir = (int)ceilf(r);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npts; i += (int64_t)gridDim.x * blockDim.x) {
    if (vis && !vis[i]) continue;
    const float px = pix[i * 2], py = pix[i * 2 + 1];
    if (!(px > -r - 1 && px < W + r && py > -r - 1 && py < H + r)) continue;
    const int64_t img = i / pts_per_img;
    const int cx = (int)floorf(px + 0.5f), cy = (int)floorf(py + 0.5f);
    for (int y = cy - ir; y <= cy + ir; ++y) {
      if (y < 0 || y >= H) continue;
      for (int x = cx - ir; x <= cx + ir; ++x) {
        if (x < 0 || x >= W) continue;
        const float dx = (float)x - px, dy = (float)y - py;
        const float d2 = dx * dx + dy * dy;
        if (d2 >= r2) continue;
        const float a = fminf(1.0f - d2 / r2, 0.9999f);
        atomicAdd(logT + (img * H + y) * W + x, logf(1.0f - a));
      }
    }
  }
}

// gpix[i] = sum over covered pixels of gmask * dmask/da * da/dpix,  dmask/da_k = T / (1 - a_k)
__global__ __launch_bounds__(256) void splat_bwd_kernel(const float* __restrict__ pix, const uint8_t* __restrict__ vis, int64_t npts,
                                                         int64_t pts_per_img, int H, int W, float r, const float* __restrict__ logT,
                                                         const float* __restrict__ gmask, float* __restrict__ gpix) {
  const float r2 = r * r;
  const int ir = (int)ceilf(r);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npts; i += (int64_t)gridDim.x * blockDim.x) {
    float gx = 0.f, gy = 0.f;
    const float px = pix[i * 2], py = pix[i * 2 + 1];
    if ((!vis || vis[i]) && (px > -r - 1 && px < W + r && py > -r - 1 && py < H + r)) {
      const int64_t img = i / pts_per_img;
      const int cx = (int)floorf(px + 0.5f), cy = (int)floorf(py + 0.5f);
      for (int y = cy - ir; y <= cy + ir; ++y) {
        if (y < 0 || y >= H) continue;
        for (int x = cx - ir; x <= cx + ir; ++x) {
          if (x < 0 || x >= W) continue;
          const float dx = (float)x - px, dy = (float)y - py;
          const float d2 = dx * dx + dy * dy;
          if (d2 >= r2) continue;
          const float a = 1.0f - d2 / r2;
          if (a >= 0.9999f) continue;                       // clamped in the forward: zero derivative
          const int64_t o = (img * H + y) * W + x;
          const float w = gmask[o] * expf(logT[o]) / (1.0f - a);
          // a = 1 - ((x-px)^2 + (y-py)^2)/r2  ->  da/dpx = 2 (x - px) / r2
          gx += w * 2.0f * dx / r2;
          gy += w * 2.0f * dy / r2;
        }
      }
    }
    gpix[i * 2] = gx;
    gpix[i * 2 + 1] = gy;
  }
}
}  // namespace

extern "C" {
int sr_svd3x3(const float* A, int64_t n, float* U, float* S, float* V, void* stream) {
  if (n < 0) return SR_EINVAL;
  if (n == 0) return SR_OK;
  if (!A || !U || !S || !V) return SR_EINVAL;
  hipLaunchKernelGGL(svd3_kernel, dim3(sr_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, A, n, U, S, V);
  return sr_launch_status();
}
int sr_splat_fwd(const float* pix, const uint8_t* vis, int64_t nimg, int64_t pts_per_img, int32_t H, int32_t W, float radius_px,
                 float* logT, void* stream) {
  if (nimg < 0 || pts_per_img < 0 || H <= 0 || W <= 0 || !(radius_px > 0.f)) return SR_EINVAL;
  const int64_t n = nimg * pts_per_img;
  if (n == 0) return SR_OK;
  if (!pix || !logT) return SR_EINVAL;
  hipLaunchKernelGGL(splat_fwd_kernel, dim3(sr_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, pix, vis, n, pts_per_img, H, W, radius_px, logT);
  return sr_launch_status();
}
int sr_splat_bwd(const float* pix, const uint8_t* vis, int64_t nimg, int64_t pts_per_img, int32_t H, int32_t W, float radius_px,
                 const float* logT, const float* gmask, float* gpix, void* stream) {
  if (nimg < 0 || pts_per_img < 0 || H <= 0 || W <= 0 || !(radius_px > 0.f)) return SR_EINVAL;
  const int64_t n = nimg * pts_per_img;
  if (n == 0) return SR_OK;
  if (!pix || !logT || !gmask || !gpix) return SR_EINVAL;
  hipLaunchKernelGGL(splat_bwd_kernel, dim3(sr_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, pix, vis, n, pts_per_img, H, W, radius_px, logT, gmask, gpix);
  return sr_launch_status();
}
}

// ------------------------------------------------------------------------------------------------
// Hard mesh rasteriser: nearest triangle per pixel centre + perspective-correct
// barycentrics -- what the reference takes from pytorch3d's MeshRasterizer(faces_per_pixel=1, blur_radius=0,
// perspective_correct=True, cull_backfaces=False) at model/network.py:492 to seed its rays (FindSurfacePs).
// Pass 1: one thread per (image, face): bounding box, inside test at pixel centres, 64-bit atomicMin of
// (depth bits << 32 | face).  Pass 2: one thread per pixel: barycentrics of the winning face.
namespace {
struct Tri { float x0, y0, z0, x1, y1, z1, x2, y2, z2; };

__device__ __forceinline__ bool load_tri(const float* __restrict__ pix, const float* __restrict__ z, const int64_t* __restrict__ faces,
                                         int64_t img, int64_t V, int64_t f, Tri& t) {
  const int64_t a = faces[f * 3], b = faces[f * 3 + 1], c = faces[f * 3 + 2];
  if (a < 0 || b < 0 || c < 0) return false;
  const int64_t o = img * V;
  t.x0 = pix[(o + a) * 2]; t.y0 = pix[(o + a) * 2 + 1]; t.z0 = z[o + a];
  t.x1 = pix[(o + b) * 2]; t.y1 = pix[(o + b) * 2 + 1]; t.z1 = z[o + b];
  t.x2 = pix[(o + c) * 2]; t.y2 = pix[(o + c) * 2 + 1]; t.z2 = z[o + c];
  return t.z0 > 0.f && t.z1 > 0.f && t.z2 > 0.f;
}

// barycentrics of pixel centre (px,py); returns false when outside (or degenerate)
__device__ __forceinline__ bool bary_of(const Tri& t, float px, float py, float& b0, float& b1, float& b2, float& depth) {
  const float area = (t.x1 - t.x0) * (t.y2 - t.y0) - (t.x2 - t.x0) * (t.y1 - t.y0);
  if (fabsf(area) < 1e-12f) return false;
  const float w0 = ((t.x1 - px) * (t.y2 - py) - (t.x2 - px) * (t.y1 - py)) / area;
  const float w1 = ((t.x2 - px) * (t.y0 - py) - (t.x0 - px) * (t.y2 - py)) / area;
  const float w2 = 1.0f - w0 - w1;
  if (w0 < 0.f || w1 < 0.f || w2 < 0.f) return false;
  const float i0 = w0 / t.z0, i1 = w1 / t.z1, i2 = w2 / t.z2;     // perspective correction
  const float s = i0 + i1 + i2;
  b0 = i0 / s; b1 = i1 / s; b2 = i2 / s;
  depth = 1.0f / s;
  return true;
}

__device__ __forceinline__ void raster_pixel(const Tri& t, int x, int y, int64_t img, int64_t f, int H, int W, unsigned long long* __restrict__ zbuf) {
  float b0, b1, b2, d;
  if (!bary_of(t, (float)x, (float)y, b0, b1, b2, d)) return;
  const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(unsigned int)f;
  unsigned long long* slot = zbuf + (img * H + y) * W + x;
  if (key < __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slot, key);   // most fragments lose: skip the RMW
}

struct Box { int xmin, xmax, ymin, ymax; };
__device__ __forceinline__ Box tri_box(const Tri& t, int H, int W) {
  Box b;
  b.xmin = max(0, (int)ceilf(fminf(t.x0, fminf(t.x1, t.x2)))); b.xmax = min(W - 1, (int)floorf(fmaxf(t.x0, fmaxf(t.x1, t.x2))));
  b.ymin = max(0, (int)ceilf(fminf(t.y0, fminf(t.y1, t.y2)))); b.ymax = min(H - 1, (int)floorf(fmaxf(t.y0, fmaxf(t.y1, t.y2))));
  return b;
}

constexpr int RASTER_SMALL = 32;   // pixel tests a single lane does itself; larger boxes go to the wave-per-triangle pass

// Pass 1: one thread per (image, face).  Almost every triangle of a remeshed template covers 0-2 pixel centres, but a
// few large ones (grazing angles, coarse remesh levels) would keep one lane looping over thousands of pixels while the
// other 63 wait: those are queued (their packed id in `big_list`, count in `big_count`) for pass 1b.
__global__ __launch_bounds__(256) void raster_pass1(const float* __restrict__ pix, const float* __restrict__ z, const int64_t* __restrict__ faces,
                                                     int64_t nimg, int64_t V, int64_t F, int H, int W, unsigned long long* __restrict__ zbuf,
                                                     int64_t* __restrict__ big_list, unsigned long long* __restrict__ big_count, int64_t big_cap) {
  const int64_t total = nimg * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t img = i / F, f = i % F;
    Tri t;
    if (!load_tri(pix, z, faces, img, V, f, t)) continue;
    const Box b = tri_box(t, H, W);
    if (b.xmax < b.xmin || b.ymax < b.ymin) continue;
    if (b.xmax - b.xmin > 256 || b.ymax - b.ymin > 256) continue;     // guard against a degenerate projection covering the image
    if ((b.xmax - b.xmin + 1) * (b.ymax - b.ymin + 1) > RASTER_SMALL) {
      const unsigned long long slot = atomicAdd(big_count, 1ull);
      if ((int64_t)slot < big_cap) { big_list[slot] = i; continue; }
    }
    for (int y = b.ymin; y <= b.ymax; ++y)
      for (int x = b.xmin; x <= b.xmax; ++x) raster_pixel(t, x, y, img, f, H, W, zbuf);
  }
}

// Pass 1b: one wave per queued triangle, lanes stride over the pixels of its box.
__global__ __launch_bounds__(256) void raster_pass1b(const float* __restrict__ pix, const float* __restrict__ z, const int64_t* __restrict__ faces,
                                                      int64_t V, int64_t F, int H, int W, unsigned long long* __restrict__ zbuf,
                                                      const int64_t* __restrict__ big_list, const unsigned long long* __restrict__ big_count, int64_t big_cap) {
  const int64_t n = min((int64_t)*big_count, big_cap);
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t k = wave; k < n; k += nwaves) {
    const int64_t i = big_list[k];
    const int64_t img = i / F, f = i % F;
    Tri t;
    if (!load_tri(pix, z, faces, img, V, f, t)) continue;
    const Box b = tri_box(t, H, W);
    const int bw = b.xmax - b.xmin + 1, npix = bw * (b.ymax - b.ymin + 1);
    for (int p = lane; p < npix; p += 64) raster_pixel(t, b.xmin + p % bw, b.ymin + p / bw, img, f, H, W, zbuf);
  }
}

__global__ __launch_bounds__(256) void raster_pass2(const float* __restrict__ pix, const float* __restrict__ z, const int64_t* __restrict__ faces,
                                                     int64_t nimg, int64_t V, int64_t F, int H, int W, const unsigned long long* __restrict__ zbuf,
                                                     int64_t* __restrict__ pix_to_face, float* __restrict__ bary, float* __restrict__ zout) {
  const int64_t total = nimg * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long key = zbuf[i];
    int64_t out = -1;
    float b0 = -1.f, b1 = -1.f, b2 = -1.f, d = -1.f;
    if (key != 0xFFFFFFFFFFFFFFFFull) {
      const int64_t img = i / ((int64_t)H * W), f = (int64_t)(key & 0xFFFFFFFFull);
      const int y = (int)((i / W) % H), x = (int)(i % W);
      Tri t;
      if (load_tri(pix, z, faces, img, V, f, t) && bary_of(t, (float)x, (float)y, b0, b1, b2, d)) out = img * F + f;   // packed index, as pytorch3d
    }
    pix_to_face[i] = out;
    bary[i * 3] = b0; bary[i * 3 + 1] = b1; bary[i * 3 + 2] = b2;
    if (zout) zout[i] = d;
  }
}
}  // namespace

extern "C" int sr_raster_mesh(const float* pix, const float* z, const int64_t* faces, int64_t nimg, int64_t V, int64_t F, int32_t H, int32_t W,
                              void* zbuf_u64, int64_t* pix_to_face, float* bary, float* zout, void* stream) {
  if (nimg < 0 || V < 0 || F < 0 || H <= 0 || W <= 0) return SR_EINVAL;
  if (nimg == 0) return SR_OK;
  if (!zbuf_u64 || !pix_to_face || !bary || (F > 0 && (!pix || !z || !faces))) return SR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(zbuf_u64, 0xFF, (size_t)nimg * H * W * 8, st) != hipSuccess) return SR_ELAUNCH;
  if (F > 0) {
    // scratch for the large-triangle queue lives in the outputs pass 2 overwrites: ids in pix_to_face, the counter in bary[0..1]
    unsigned long long* big_count = (unsigned long long*)bary;
    const int64_t big_cap = nimg * H * W;
    if (hipMemsetAsync(big_count, 0, 8, st) != hipSuccess) return SR_ELAUNCH;
    hipLaunchKernelGGL(raster_pass1, dim3(sr_stream_grid(nimg * F, 256)), dim3(256), 0, st, pix, z, faces, nimg, V, F, H, W,
                       (unsigned long long*)zbuf_u64, pix_to_face, big_count, big_cap);
    hipLaunchKernelGGL(raster_pass1b, dim3(512), dim3(256), 0, st, pix, z, faces, V, F, H, W, (unsigned long long*)zbuf_u64, pix_to_face,
                       big_count, big_cap);
  }
  hipLaunchKernelGGL(raster_pass2, dim3(sr_stream_grid(nimg * H * W, 256)), dim3(256), 0, st, pix, z, faces, nimg, V, F, H, W,
                     (const unsigned long long*)zbuf_u64, pix_to_face, bary, zout);
  return sr_launch_status();
}
