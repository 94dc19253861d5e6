// The two rasterisation steps either side of the ray refiner, with the semantics of the
// pytorch3d 0.4.0 calls the reference makes (third-party code that is not in the reference repository; restated in
// oracle/raster_oracle.py, parity unpinned):
//
//  * sr_points_silhouette_{fwd,bwd}: PointsRasterizer(radius, points_per_pixel = K) + AlphaCompositor with ONE all-ones
//    feature (model/network.py:178-190,495-497 through PointsRendererWithFrags, model/CameraMine.py:285-305):
//        mask[pixel] = sum_k a_k prod_{j<k} (1 - a_j)   over the K points NEAREST IN Z that cover the pixel,
//        a = 1 - dist2 / r^2,  dist2 < r^2 (NDC units),  points with z < 0 skipped.
//    With unit features the composite is 1 - prod_k (1 - a_k): commutative, so pixels covered by <= K points are
//    accumulated as sum_k log(1 - a_k) with atomics, in one pass over the points and with no per-pixel lists.  The sum is kept
//    in FIXED POINT and shares one 64-bit word with the pair count (one integer atomic per pair instead of a float and an
//    integer one): integer addition is associative, so the result does not depend on the order in which the atomics land --
//    the image, and everything the training step derives from it, is bit-reproducible run to run.  Only the
//    pixels that more than K points cover (silhouette rims at grazing angles; none to a few hundred per image) go
//    through the selection: their (z, point) keys are gathered into buckets, a wave per pixel finds the K-th smallest
//    key by a 64-step binary search on the key bits, and the pixel is re-composited from the keys up to that threshold.
//    The per-pixel threshold is what the backward pass reads to decide whether a (point, pixel) pair took part.
//
//  * sr_rasterize_meshes: MeshRasterizer(blur_radius 0, faces_per_pixel 1, perspective_correct, no clipping of the
//    barycentrics, no culling) (model/network.py:877-892, 492): per pixel the nearest face whose perspective-corrected
//    barycentrics at the pixel centre are all > 0, arithmetic as CheckPixelInsideFace of rasterize_meshes.cu.
//
// Both work in pytorch3d's NDC frame: +x left, +y up, pixel (row, col) centred at (1 - (2 col + 1)/W, 1 - (2 row + 1)/H).
#include "sr_common.h"

namespace {
constexpr float kEps = 1e-8f;        // kEpsilon of pytorch3d's geometry_utils.cuh
constexpr float kAlphaMax = 1.0f - 1e-6f;   // a point exactly on a pixel centre would make log(1 - a) infinite

__device__ __forceinline__ float pix_to_ndc(int i, int S) { return 1.0f - (2.0f * (float)i + 1.0f) / (float)S; }

struct PixBox { int c0, c1, r0, r1; };
// pixels whose centres can lie within `radius` (NDC) of the point
__device__ __forceinline__ PixBox point_box(float px, float py, float radius, int H, int W) {
  PixBox b;
  // col = ((1 - x) W - 1) / 2; one extra pixel each side absorbs the rounding of this inverse map
  const float cc = ((1.0f - px) * (float)W - 1.0f) * 0.5f, rc = ((1.0f - py) * (float)H - 1.0f) * 0.5f;
  const float rx = radius * (float)W * 0.5f, ry = radius * (float)H * 0.5f;
  b.c0 = max(0, (int)floorf(cc - rx) - 1); b.c1 = min(W - 1, (int)ceilf(cc + rx) + 1);
  b.r0 = max(0, (int)floorf(rc - ry) - 1); b.r1 = min(H - 1, (int)ceilf(rc + ry) + 1);
  return b;
}

__device__ __forceinline__ unsigned long long point_key(float z, int64_t packed_idx) {
  return ((unsigned long long)__float_as_uint(z) << 32) | (unsigned long long)(uint32_t)packed_idx;   // z >= 0: bits are monotonic
}

__device__ __forceinline__ bool point_ok(float px, float py, float z) {
  return z >= 0.f && fabsf(px) < 4.f && fabsf(py) < 4.f;       // behind the camera / NaN / absurdly far off screen
}

// Packed accumulator of a pixel: low kCountBits = number of covering points, the bits above = sum of round(log(1 - a) * 2^frac)
// (two's complement).  frac is chosen from K so that the sum field holds K terms of log(1 - a) >= log(1e-6) = -13.8 exactly; pixels
// covered by more than K points may wrap it -- they are re-composited from their keys by ps_select.
constexpr int kCountBits = 22;
__host__ __device__ inline int ps_frac_bits(int K) {
  int ib = 1;                                   // integer bits (incl. sign) for |sum| <= 14 K
  while ((1ll << (ib - 1)) <= 14ll * K) ++ib;
  const int f = 64 - kCountBits - ib;
  return f > 36 ? 36 : f;
}
__device__ __forceinline__ long long ps_quantise(float d2, float r2, float scale) {
  return __float2ll_rn(__logf(1.0f - fminf(1.0f - d2 / r2, kAlphaMax)) * scale);        // <= 0
}

// Pass 1: every (point, covered pixel) pair: ONE 64-bit integer atomic (count + fixed-point log(1 - a)).
__global__ __launch_bounds__(256) void ps_accumulate(const float* __restrict__ xy, const float* __restrict__ z, int64_t npts, int64_t V,
                                                      int H, int W, float radius, float scale, unsigned long long* __restrict__ acc) {
  const float r2 = radius * radius;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npts; i += (int64_t)gridDim.x * blockDim.x) {
    const float px = xy[i * 2], py = xy[i * 2 + 1];
    if (!point_ok(px, py, z[i])) continue;
    const int64_t img = i / V;
    const PixBox b = point_box(px, py, radius, H, W);
    for (int r = b.r0; r <= b.r1; ++r) {
      const float dy = pix_to_ndc(r, H) - py;
      for (int c = b.c0; c <= b.c1; ++c) {
        const float dx = pix_to_ndc(c, W) - px;
        const float d2 = dx * dx + dy * dy;
        if (!(d2 < r2)) continue;
        const int64_t o = (img * H + r) * W + c;
        atomicAdd(acc + o, (unsigned long long)(ps_quantise(d2, r2, scale) * (1ll << kCountBits) + 1ll));
      }
    }
  }
}

// Pass 2: per pixel: finish the common case, queue the pixels that more than K points cover.
// over[0] = number of queued pixels, over[1] = number of keys their buckets hold.
__global__ __launch_bounds__(256) void ps_resolve(int64_t npix, int K, float inv_scale, const unsigned long long* __restrict__ acc,
                                                   uint32_t* __restrict__ count, float* __restrict__ logT,
                                                   float* __restrict__ mask, unsigned long long* __restrict__ thresh,
                                                   int32_t* __restrict__ slot_of, unsigned long long* __restrict__ over,
                                                   int64_t* __restrict__ over_pix, int64_t* __restrict__ over_off) {
  for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < npix; o += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long packed = acc[o];
    const uint32_t n = (uint32_t)(packed & ((1ull << kCountBits) - 1));          // (2^22 or more points per image would let it wrap: refused by the launcher)
    const float lt = (float)((long long)packed >> kCountBits) * inv_scale;       // arithmetic shift: the signed fixed-point sum
    count[o] = n;
    logT[o] = lt;
    thresh[o] = ~0ull;
    slot_of[o] = -1;
    mask[o] = 1.0f - __expf(lt);
    if (n > (uint32_t)K) {
      const unsigned long long slot = atomicAdd(over, 1ull);
      over_pix[slot] = o;
      over_off[slot] = (int64_t)atomicAdd(over + 1, (unsigned long long)n);
      slot_of[o] = (int32_t)slot;
    }
  }
}

// Pass 3: gather the keys of the queued pixels (fill[] counts the keys written so far per slot).
__global__ __launch_bounds__(256) void ps_gather(const float* __restrict__ xy, const float* __restrict__ z, int64_t npts, int64_t V, int H, int W,
                                                  float radius, const int32_t* __restrict__ slot_of, const unsigned long long* __restrict__ over,
                                                  const int64_t* __restrict__ over_off, uint32_t* __restrict__ fill,
                                                  unsigned long long* __restrict__ bucket) {
  if (over[0] == 0) return;
  const float r2 = radius * radius;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npts; i += (int64_t)gridDim.x * blockDim.x) {
    const float px = xy[i * 2], py = xy[i * 2 + 1], pz = z[i];
    if (!point_ok(px, py, pz)) continue;
    const int64_t img = i / V;
    const PixBox b = point_box(px, py, radius, H, W);
    for (int r = b.r0; r <= b.r1; ++r) {
      const float dy = pix_to_ndc(r, H) - py;
      for (int c = b.c0; c <= b.c1; ++c) {
        const int32_t slot = slot_of[(img * H + r) * W + c];
        if (slot < 0) continue;
        const float dx = pix_to_ndc(c, W) - px;
        if (!(dx * dx + dy * dy < r2)) continue;
        bucket[over_off[slot] + atomicAdd(fill + slot, 1u)] = point_key(pz, i);
      }
    }
  }
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

// Pass 4: one wave per queued pixel: K-th smallest key (binary search on the 64 key bits), then the composite of the
// keys up to it (integer sums: deterministic whatever order the bucket was filled in).
__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

__global__ __launch_bounds__(256) void ps_select(const float* __restrict__ xy, int H, int W, float radius, int K, float scale, const uint32_t* __restrict__ count,
                                                  const unsigned long long* __restrict__ over, const int64_t* __restrict__ over_pix,
                                                  const int64_t* __restrict__ over_off, const unsigned long long* __restrict__ bucket,
                                                  float* __restrict__ mask, float* __restrict__ logT, unsigned long long* __restrict__ thresh) {
  const int64_t nover = (int64_t)over[0];
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  const float r2 = radius * radius;
  for (int64_t s = wave; s < nover; s += nwaves) {
    const int64_t o = over_pix[s];
    const uint32_t n = count[o];
    const unsigned long long* keys = bucket + over_off[s];
    unsigned long long prefix = 0;           // the K-th smallest key, built from the top bit down
    uint32_t want = (uint32_t)K;             // rank still to be found among the keys that match the prefix so far
    for (int bit = 63; bit >= 0; --bit) {
      const unsigned long long hi_mask = bit == 63 ? 0ull : (~0ull << (bit + 1));
      uint32_t zeros = 0;
      for (uint32_t j = lane; j < n; j += 64) {
        const unsigned long long k = keys[j];
        zeros += ((k & hi_mask) == (prefix & hi_mask) && !((k >> bit) & 1ull)) ? 1u : 0u;
      }
      zeros = wave_sum_u32(zeros);
      if (want > zeros) { want -= zeros; prefix |= 1ull << bit; }
    }
    const int c = (int)(o % W), r = (int)((o / W) % H);
    const float xf = pix_to_ndc(c, W), yf = pix_to_ndc(r, H);
    long long qsum = 0;                      // the same fixed-point terms as pass 1: the bucket order (claimed by atomics) does not matter
    for (uint32_t j = lane; j < n; j += 64) {
      const unsigned long long k = keys[j];
      if (k > prefix) continue;
      const int64_t i = (int64_t)(uint32_t)k;
      const float dx = xf - xy[i * 2], dy = yf - xy[i * 2 + 1];
      qsum += ps_quantise(dx * dx + dy * dy, r2, scale);
    }
    qsum = wave_sum_i64(qsum);
    const float acc = (float)qsum / scale;
    if (lane == 0) { thresh[o] = prefix; logT[o] = acc; mask[o] = 1.0f - __expf(acc); }
  }
}

// Backward: d mask / d xy_k = (prod_{j != k} (1 - a_j)) d a_k / d xy_k  over the pairs that took part.
__global__ __launch_bounds__(256) void ps_backward(const float* __restrict__ xy, const float* __restrict__ z, int64_t npts, int64_t V, int H, int W,
                                                    float radius, const float* __restrict__ logT, const unsigned long long* __restrict__ thresh,
                                                    const float* __restrict__ gmask, float* __restrict__ gxy) {
  const float r2 = radius * radius;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npts; i += (int64_t)gridDim.x * blockDim.x) {
    float gx = 0.f, gy = 0.f;
    const float px = xy[i * 2], py = xy[i * 2 + 1], pz = z[i];
    if (point_ok(px, py, pz)) {
      const int64_t img = i / V;
      const unsigned long long key = point_key(pz, i);
      const PixBox b = point_box(px, py, radius, H, W);
      for (int r = b.r0; r <= b.r1; ++r) {
        const float dy = pix_to_ndc(r, H) - py;
        for (int c = b.c0; c <= b.c1; ++c) {
          const float dx = pix_to_ndc(c, W) - px;
          const float d2 = dx * dx + dy * dy;
          if (!(d2 < r2)) continue;
          const int64_t o = (img * H + r) * W + c;
          if (key > thresh[o]) continue;                      // not among the K nearest of this pixel
          const float a = 1.0f - d2 / r2;
          if (a >= kAlphaMax) continue;                       // clamped in the forward
          const float w = gmask[o] * __expf(logT[o]) / (1.0f - a);
          gx += w * 2.0f * dx / r2;                           // a = 1 - ((xf-px)^2 + (yf-py)^2)/r2  ->  da/dpx = 2 (xf - px) / r2
          gy += w * 2.0f * dy / r2;
        }
      }
    }
    gxy[i * 2] = gx;
    gxy[i * 2 + 1] = gy;
  }
}

struct PsLayout { int64_t npix, cap; size_t acc, count, logT, thresh, slot_of, over, over_pix, over_off, fill, bucket, total; };
PsLayout ps_layout(int64_t nimg, int64_t V, int32_t H, int32_t W, float radius) {
  PsLayout L;
  L.npix = nimg * H * W;
  const int64_t bx = (int64_t)ceilf(radius * W * 0.5f) * 2 + 5, by = (int64_t)ceilf(radius * H * 0.5f) * 2 + 5;
  L.cap = nimg * V * bx * by;                 // every pair a point can form: the buckets can never overflow
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
  L.thresh = take((size_t)L.npix * 8); L.logT = take((size_t)L.npix * 4); L.count = take((size_t)L.npix * 4); L.acc = take((size_t)L.npix * 8);
  L.slot_of = take((size_t)L.npix * 4); L.over = take(16); L.over_pix = take((size_t)L.npix * 8); L.over_off = take((size_t)L.npix * 8);
  L.fill = take((size_t)L.npix * 4); L.bucket = take((size_t)L.cap * 8);
  L.total = o;
  return L;
}
}  // namespace

extern "C" {
int64_t sr_points_silhouette_workspace_bytes(int64_t nimg, int64_t pts_per_img, int32_t H, int32_t W, float radius) {
  if (nimg < 0 || pts_per_img < 0 || H <= 0 || W <= 0 || !(radius > 0.f)) return SR_EINVAL;
  return (int64_t)ps_layout(nimg, pts_per_img, H, W, radius).total;
}

int sr_points_silhouette_fwd(const float* xy_ndc, const float* z, int64_t nimg, int64_t pts_per_img, int32_t H, int32_t W, float radius,
                             int32_t K, float* mask, void* workspace, void* stream) {
  if (nimg < 0 || pts_per_img < 0 || H <= 0 || W <= 0 || !(radius > 0.f) || K <= 0 || K > (1 << 16) || nimg * pts_per_img >= ((int64_t)1 << 32)) return SR_EINVAL;
  if (pts_per_img >= ((int64_t)1 << kCountBits)) return SR_EINVAL;     // the per-pixel pair count shares a word with the fixed-point sum
  if (nimg == 0) return SR_OK;
  if (!mask || !workspace || ((uintptr_t)workspace & 255) || (pts_per_img > 0 && (!xy_ndc || !z))) return SR_EINVAL;
  const PsLayout L = ps_layout(nimg, pts_per_img, H, W, radius);
  char* ws = (char*)workspace;
  hipStream_t st = (hipStream_t)stream;
  // zero: packed accumulators, over counters ... fill: one memset over the contiguous head, buckets untouched
  if (hipMemsetAsync(ws + L.logT, 0, L.bucket - L.logT, st) != hipSuccess) return SR_ELAUNCH;
  const float scale = (float)(1ll << ps_frac_bits(K));
  unsigned long long* acc = (unsigned long long*)(ws + L.acc);
  const int64_t npts = nimg * pts_per_img;
  uint32_t* count = (uint32_t*)(ws + L.count); float* logT = (float*)(ws + L.logT);
  unsigned long long* thresh = (unsigned long long*)(ws + L.thresh); int32_t* slot_of = (int32_t*)(ws + L.slot_of);
  unsigned long long* over = (unsigned long long*)(ws + L.over); int64_t* over_pix = (int64_t*)(ws + L.over_pix);
  int64_t* over_off = (int64_t*)(ws + L.over_off); uint32_t* fill = (uint32_t*)(ws + L.fill);
  unsigned long long* bucket = (unsigned long long*)(ws + L.bucket);
  if (npts > 0)
    hipLaunchKernelGGL(ps_accumulate, dim3(sr_stream_grid(npts, 256)), dim3(256), 0, st, xy_ndc, z, npts, pts_per_img, H, W, radius, scale, acc);
  hipLaunchKernelGGL(ps_resolve, dim3(sr_stream_grid(L.npix, 256)), dim3(256), 0, st, L.npix, K, 1.0f / scale, acc, count, logT, mask, thresh, slot_of, over, over_pix, over_off);
  if (npts > 0) {
    hipLaunchKernelGGL(ps_gather, dim3(sr_stream_grid(npts, 256)), dim3(256), 0, st, xy_ndc, z, npts, pts_per_img, H, W, radius, slot_of, over, over_off, fill, bucket);
    hipLaunchKernelGGL(ps_select, dim3(512), dim3(256), 0, st, xy_ndc, H, W, radius, K, scale, count, over, over_pix, over_off, bucket, mask, logT, thresh);
  }
  return sr_launch_status();
}

int sr_points_silhouette_bwd(const float* xy_ndc, const float* z, int64_t nimg, int64_t pts_per_img, int32_t H, int32_t W, float radius,
                             const void* workspace, const float* gmask, float* gxy, void* stream) {
  if (nimg < 0 || pts_per_img < 0 || H <= 0 || W <= 0 || !(radius > 0.f)) return SR_EINVAL;
  const int64_t npts = nimg * pts_per_img;
  if (npts == 0) return SR_OK;
  if (!xy_ndc || !z || !workspace || !gmask || !gxy) return SR_EINVAL;
  const PsLayout L = ps_layout(nimg, pts_per_img, H, W, radius);
  const char* ws = (const char*)workspace;
  hipLaunchKernelGGL(ps_backward, dim3(sr_stream_grid(npts, 256)), dim3(256), 0, (hipStream_t)stream, xy_ndc, z, npts, pts_per_img, H, W, radius,
                     (const float*)(ws + L.logT), (const unsigned long long*)(ws + L.thresh), gmask, gxy);
  return sr_launch_status();
}
}

// ------------------------------------------------------------------------------------------------
// Mesh rasteriser.  Pass 1: one thread per (image, face) tests the pixel centres of its bounding box and keeps
// the nearest hit with a 64-bit atomicMin of (depth bits << 32 | face); large boxes go to a wave-per-face pass.
// Pass 2: one thread per pixel recomputes the barycentrics of the winner.
namespace {
struct Tri { float x0, y0, z0, x1, y1, z1, x2, y2, z2; };

__device__ __forceinline__ bool load_tri(const float* __restrict__ xy, const float* __restrict__ z, const int64_t* __restrict__ faces,
                                         int64_t img, int64_t V, int64_t f, Tri& t) {
  const int64_t a = faces[f * 3], b = faces[f * 3 + 1], c = faces[f * 3 + 2];
  if (a < 0 || b < 0 || c < 0) return false;            // marching-cubes border faces (MCGpu: owner cell outside the volume)
  const int64_t o = img * V;
  t.x0 = xy[(o + a) * 2]; t.y0 = xy[(o + a) * 2 + 1]; t.z0 = z[o + a];
  t.x1 = xy[(o + b) * 2]; t.y1 = xy[(o + b) * 2 + 1]; t.z1 = z[o + b];
  t.x2 = xy[(o + c) * 2]; t.y2 = xy[(o + c) * 2 + 1]; t.z2 = z[o + c];
  if (fmaxf(t.z0, fmaxf(t.z1, t.z2)) < 0.f) return false;                                      // face behind the camera
  const float area = (t.x0 - t.x1) * (t.y2 - t.y1) - (t.y0 - t.y1) * (t.x2 - t.x1);            // EdgeFunctionForward(v0, v1, v2)
  return !(area <= kEps && area >= -kEps) && area == area;
}

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

// CheckPixelInsideFace (blur 0, perspective correct, unclipped): barycentrics + depth at pixel centre (xf, yf)
__device__ __forceinline__ bool face_hit(const Tri& t, float xf, float yf, float& b0, float& b1, float& b2, float& pz) {
  const float den = edge_fn(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1) + kEps;
  const float w0 = edge_fn(xf, yf, t.x1, t.y1, t.x2, t.y2) / den;
  const float w1 = edge_fn(xf, yf, t.x2, t.y2, t.x0, t.y0) / den;
  const float w2 = edge_fn(xf, yf, t.x0, t.y0, t.x1, t.y1) / den;
  const float t0 = w0 * t.z1 * t.z2, t1 = t.z0 * w1 * t.z2, t2 = t.z0 * t.z1 * w2;
  const float dn = fmaxf(t0 + t1 + t2, kEps);
  b0 = t0 / dn; b1 = t1 / dn; b2 = t2 / dn;
  pz = b0 * t.z0 + b1 * t.z1 + b2 * t.z2;
  return b0 > 0.f && b1 > 0.f && b2 > 0.f && pz >= 0.f;
}

struct Box { int c0, c1, r0, r1; };
__device__ __forceinline__ Box tri_box(const Tri& t, int H, int W) {
  // pixel centres inside the NDC bounding box (CheckPointOutsideBoundingBox with blur 0); NDC decreases with the index
  const float xmin = fminf(t.x0, fminf(t.x1, t.x2)), xmax = fmaxf(t.x0, fmaxf(t.x1, t.x2));
  const float ymin = fminf(t.y0, fminf(t.y1, t.y2)), ymax = fmaxf(t.y0, fmaxf(t.y1, t.y2));
  Box b;
  // centre of column c lies in [xmin, xmax]  <=>  c in [col(xmax), col(xmin)], col(x) = ((1 - x) W - 1) / 2; the 1e-3 pixel of slack
  // covers the rounding of this inverse map (raster_pixel repeats the exact NDC test), and keeps the box tight: most faces of a
  // remeshed template cover 0-2 pixel centres and must stay on the one-lane path
  const float eps = 1e-3f;
  b.c0 = max(0, (int)ceilf(((1.0f - xmax) * (float)W - 1.0f) * 0.5f - eps)); b.c1 = min(W - 1, (int)floorf(((1.0f - xmin) * (float)W - 1.0f) * 0.5f + eps));
  b.r0 = max(0, (int)ceilf(((1.0f - ymax) * (float)H - 1.0f) * 0.5f - eps)); b.r1 = min(H - 1, (int)floorf(((1.0f - ymin) * (float)H - 1.0f) * 0.5f + eps));
  return b;
}

__device__ __forceinline__ void raster_pixel(const Tri& t, int c, int r, int64_t img, int64_t f, int H, int W, unsigned long long* __restrict__ zbuf) {
  const float xf = pix_to_ndc(c, W), yf = pix_to_ndc(r, H);
  if (xf > fmaxf(t.x0, fmaxf(t.x1, t.x2)) || xf < fminf(t.x0, fminf(t.x1, t.x2)) || yf > fmaxf(t.y0, fmaxf(t.y1, t.y2)) ||
      yf < fminf(t.y0, fminf(t.y1, t.y2)))
    return;
  float b0, b1, b2, pz;
  if (!face_hit(t, xf, yf, b0, b1, b2, pz)) return;
  const unsigned long long key = ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned long long)(unsigned int)f;
  unsigned long long* slot = zbuf + (img * H + r) * W + c;
  if (key < __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(slot, key);   // most fragments lose: skip the RMW
}

constexpr int RASTER_SMALL = 32;   // pixel tests a single lane does itself; larger boxes go to the wave-per-face pass

__global__ __launch_bounds__(256) void rm_pass1(const float* __restrict__ xy, const float* __restrict__ z, const int64_t* __restrict__ faces,
                                                 int64_t nimg, int64_t V, int64_t F, int H, int W, unsigned long long* __restrict__ zbuf,
                                                 int64_t* __restrict__ big_list, unsigned long long* __restrict__ big_count, int64_t big_cap) {
  const int64_t total = nimg * F;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t img = i / F, f = i % F;
    Tri t;
    if (!load_tri(xy, z, faces, img, V, f, t)) continue;
    const Box b = tri_box(t, H, W);
    if (b.c1 < b.c0 || b.r1 < b.r0) continue;
    if (!(fabsf(t.x0) < 8.f && fabsf(t.x1) < 8.f && fabsf(t.x2) < 8.f && fabsf(t.y0) < 8.f && fabsf(t.y1) < 8.f && fabsf(t.y2) < 8.f)) continue;   // NaN / wildly off-screen vertex
    if ((b.c1 - b.c0 + 1) * (b.r1 - b.r0 + 1) > RASTER_SMALL) {
      const unsigned long long slot = atomicAdd(big_count, 1ull);
      if ((int64_t)slot < big_cap) { big_list[slot] = i; continue; }
    }
    for (int r = b.r0; r <= b.r1; ++r)
      for (int c = b.c0; c <= b.c1; ++c) raster_pixel(t, c, r, img, f, H, W, zbuf);
  }
}

__global__ __launch_bounds__(256) void rm_pass1b(const float* __restrict__ xy, const float* __restrict__ z, const int64_t* __restrict__ faces,
                                                  int64_t V, int64_t F, int H, int W, unsigned long long* __restrict__ zbuf,
                                                  const int64_t* __restrict__ big_list, const unsigned long long* __restrict__ big_count, int64_t big_cap) {
  const int64_t n = min((int64_t)*big_count, big_cap);
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t k = wave; k < n; k += nwaves) {
    const int64_t i = big_list[k];
    const int64_t img = i / F, f = i % F;
    Tri t;
    if (!load_tri(xy, z, faces, img, V, f, t)) continue;
    const Box b = tri_box(t, H, W);
    const int bw = b.c1 - b.c0 + 1, npix = bw * (b.r1 - b.r0 + 1);
    if (fminf(t.z0, fminf(t.z1, t.z2)) <= 0.f) {
      // a vertex behind the camera (unclipped, as pytorch3d 0.4.0): the pixels that pass face_hit are NOT the interior of the projected
      // triangle then, so every pixel centre of the box is tested
      for (int p = lane; p < npix; p += 64) raster_pixel(t, b.c0 + p % bw, b.r0 + p / bw, img, f, H, W, zbuf);
      continue;
    }
    // All vertices in front: a pixel centre passes only strictly inside the projected triangle.  A lane takes a LINE of the box -- a row if
    // the box is taller than wide, a column otherwise -- and walks the pixel centres between the two edge crossings of that line (+- one
    // pixel of slack; raster_pixel repeats the exact test), so a long thin face -- the stretched faces of a template whose neighbouring
    // vertices follow different bones have boxes of 10^3..10^5 pixel centres and cover a few hundred -- costs its lines + its area, not
    // its box.
    const bool by_rows = (b.r1 - b.r0) >= (b.c1 - b.c0);
    const int l0 = by_rows ? b.r0 : b.c0, l1 = by_rows ? b.r1 : b.c1;          // the lines
    const int m0 = by_rows ? b.c0 : b.r0, m1 = by_rows ? b.c1 : b.r1;          // the range along a line
    const int nl = by_rows ? H : W, nm = by_rows ? W : H;
    // u = the coordinate that is constant on a line, v = the one that runs along it
    const float eu[3] = {by_rows ? t.y0 : t.x0, by_rows ? t.y1 : t.x1, by_rows ? t.y2 : t.x2};
    const float ev[3] = {by_rows ? t.x0 : t.y0, by_rows ? t.x1 : t.y1, by_rows ? t.x2 : t.y2};
    for (int l = l0 + lane; l <= l1; l += 64) {
      const float uf = pix_to_ndc(l, nl);
      float vlo = 3.0e38f, vhi = -3.0e38f;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const float ua = eu[e], va = ev[e], ub = eu[(e + 1) % 3], vb = ev[(e + 1) % 3];
        if ((uf - ua) * (uf - ub) > 0.f) continue;                   // the line misses this edge
        if (ua == ub) { vlo = fminf(vlo, fminf(va, vb)); vhi = fmaxf(vhi, fmaxf(va, vb)); continue; }
        const float v = va + (uf - ua) * ((vb - va) / (ub - ua));
        vlo = fminf(vlo, v); vhi = fmaxf(vhi, v);
      }
      if (vlo > vhi) continue;
      // index of an NDC coordinate v along the line: ((1 - v) n - 1) / 2, decreasing in v
      const int i0 = max(m0, (int)floorf(((1.0f - vhi) * (float)nm - 1.0f) * 0.5f) - 1);
      const int i1 = min(m1, (int)ceilf(((1.0f - vlo) * (float)nm - 1.0f) * 0.5f) + 1);
      for (int i2 = i0; i2 <= i1; ++i2) raster_pixel(t, by_rows ? i2 : l, by_rows ? l : i2, img, f, H, W, zbuf);
    }
  }
}

__global__ __launch_bounds__(256) void rm_pass2(const float* __restrict__ xy, const float* __restrict__ z, const int64_t* __restrict__ faces,
                                                 int64_t nimg, int64_t V, int64_t F, int H, int W, const unsigned long long* __restrict__ zbuf,
                                                 int64_t* __restrict__ pix_to_face, float* __restrict__ bary, float* __restrict__ zout) {
  const int64_t total = nimg * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const unsigned long long key = zbuf[i];
    int64_t out = -1;
    float b0 = -1.f, b1 = -1.f, b2 = -1.f, d = -1.f;
    if (key != 0xFFFFFFFFFFFFFFFFull) {
      const int64_t img = i / ((int64_t)H * W), f = (int64_t)(key & 0xFFFFFFFFull);
      const int r = (int)((i / W) % H), c = (int)(i % W);
      Tri t;
      if (load_tri(xy, z, faces, img, V, f, t) && face_hit(t, pix_to_ndc(c, W), pix_to_ndc(r, H), b0, b1, b2, d)) out = img * F + f;   // packed index, as pytorch3d
    }
    pix_to_face[i] = out;
    bary[i * 3] = b0; bary[i * 3 + 1] = b1; bary[i * 3 + 2] = b2;
    if (zout) zout[i] = d;
  }
}
}  // namespace

extern "C" int sr_rasterize_meshes(const float* xy_ndc, const float* z, const int64_t* faces, int64_t nimg, int64_t V, int64_t F, int32_t H,
                                   int32_t W, void* zbuf_u64, int64_t* pix_to_face, float* bary, float* zout, void* stream) {
  if (nimg < 0 || V < 0 || F < 0 || H <= 0 || W <= 0 || F >= ((int64_t)1 << 32)) return SR_EINVAL;
  if (nimg == 0) return SR_OK;
  if (!zbuf_u64 || !pix_to_face || !bary || (F > 0 && (!xy_ndc || !z || !faces))) return SR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(zbuf_u64, 0xFF, (size_t)nimg * H * W * 8, st) != hipSuccess) return SR_ELAUNCH;
  if (F > 0) {
    // scratch for the large-face queue lives in the outputs pass 2 overwrites: ids in pix_to_face, the counter in bary[0..1]
    unsigned long long* big_count = (unsigned long long*)bary;
    const int64_t big_cap = nimg * H * W;
    if (hipMemsetAsync(big_count, 0, 8, st) != hipSuccess) return SR_ELAUNCH;
    hipLaunchKernelGGL(rm_pass1, dim3(sr_stream_grid(nimg * F, 256)), dim3(256), 0, st, xy_ndc, z, faces, nimg, V, F, H, W,
                       (unsigned long long*)zbuf_u64, pix_to_face, big_count, big_cap);
    hipLaunchKernelGGL(rm_pass1b, dim3(512), dim3(256), 0, st, xy_ndc, z, faces, V, F, H, W, (unsigned long long*)zbuf_u64, pix_to_face,
                       big_count, big_cap);
  }
  hipLaunchKernelGGL(rm_pass2, dim3(sr_stream_grid(nimg * H * W, 256)), dim3(256), 0, st, xy_ndc, z, faces, nimg, V, F, H, W,
                     (const unsigned long long*)zbuf_u64, pix_to_face, bary, zout);
  return sr_launch_status();
}
