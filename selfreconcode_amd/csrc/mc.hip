// Deterministic marching cubes.  Same geometry as the reference's
// MCGpu (MCGpu/CudaKernels.cu:304-523): 256-case table, strict `value < iso` classification,
// vertices owned by the cell whose corner-0 edges (cube edges 0, 3, 8 = the x, y, z lattice edges
// at the cell origin) are crossed, cells with i = NX-1 / j = NY-1 / k = NZ-1 skipped, offsets
// t = (float)((double)(iso - v1) / (double)(float)(v2 - v1)) (0.5 when equal), faces with reversed
// winding in int64.  What changes is the ORDER: the reference hands out vertex ids and face slots
// with atomicAdd (its output order differs run to run); here ids come from an
// exclusive scan, so vertices are sorted by lattice-edge key (cell*3 + dir) and faces by
// (cell, triangle) -- bit-reproducible, and equal to the reference after canonicalisation.
//
// Traffic: classify reads the volume once (4 B/voxel, rows along k are contiguous so a wave reads
// coalesced 256-byte runs) and writes two u32 counters per voxel; the reference rewrites a
// 12 B/voxel edge table on every call (K8).  HBM-bound.
#include "sr_common.h"
#include "mc_tables.h"

namespace {
__constant__ uint64_t dTri[256];
__constant__ uint8_t dTriCount[256];
bool g_tables_loaded[16] = {false};

struct Dim { int NX, NY, NZ; };

__device__ __forceinline__ int cube_case(const float* __restrict__ s, int64_t c, int NY, int NZ, float iso, float (&v)[8]) {
  const int64_t sx = (int64_t)NY * NZ, sy = NZ;
  v[0] = s[c]; v[1] = s[c + sx]; v[2] = s[c + sx + sy]; v[3] = s[c + sy];
  v[4] = s[c + 1]; v[5] = s[c + sx + 1]; v[6] = s[c + sx + sy + 1]; v[7] = s[c + sy + 1];
  int idx = 0;
#pragma unroll
  for (int b = 0; b < 8; ++b) idx |= (v[b] < iso) ? (1 << b) : 0;
  return idx;
}

// crossing mask of the 3 owned edges (bit0: x edge v0v1, bit1: y edge v0v3, bit2: z edge v0v4)
__device__ __forceinline__ int owned_mask(float v0, float vx, float vy, float vz, float iso) {
  const bool b0 = v0 < iso;
  return ((b0 != (vx < iso)) ? 1 : 0) | ((b0 != (vy < iso)) ? 2 : 0) | ((b0 != (vz < iso)) ? 4 : 0);
}

__global__ __launch_bounds__(256) void mc_classify_kernel(const float* __restrict__ sdf, Dim d, float iso, uint32_t* __restrict__ vcnt,
                                                           uint32_t* __restrict__ tcnt) {
  const int64_t total = (int64_t)d.NX * d.NY * d.NZ;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(c % d.NZ), j = (int)((c / d.NZ) % d.NY), i = (int)(c / ((int64_t)d.NZ * d.NY));
    uint32_t nv = 0, nt = 0;
    if (i < d.NX - 1 && j < d.NY - 1 && k < d.NZ - 1) {
      float v[8];
      const int idx = cube_case(sdf, c, d.NY, d.NZ, iso, v);
      nt = dTriCount[idx];
      nv = __popc(owned_mask(v[0], v[1], v[3], v[4], iso));
    }
    vcnt[c] = nv;
    tcnt[c] = nt;
  }
}

// ---- exclusive scan of u32 (three small kernels: per-block sums, scan of sums, down-sweep) ----
constexpr int SCAN_ITEMS = 2048;   // elements per workgroup (256 threads x 8)

__global__ __launch_bounds__(256) void scan_block_sums(const uint32_t* __restrict__ in, int64_t n, uint32_t* __restrict__ sums) {
  __shared__ uint32_t red[256];
  const int64_t base = (int64_t)blockIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
  for (int e = 0; e < 8; ++e) {
    const int64_t i = base + threadIdx.x * 8 + e;
    if (i < n) s += in[i];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(1024) void scan_sums_serial(uint32_t* __restrict__ sums, int nblocks, uint32_t* __restrict__ total) {
  // single workgroup: chunked inclusive scan over the (few thousand) block sums
  __shared__ uint32_t buf[1024];
  __shared__ uint32_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < nblocks ? sums[i] : 0;
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const uint32_t t = threadIdx.x >= off ? buf[threadIdx.x - off] : 0;
      __syncthreads();
      buf[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblocks) sums[i] = carry + buf[threadIdx.x] - v;   // exclusive
    __syncthreads();
    if (threadIdx.x == 1023) carry += buf[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(256) void scan_downsweep(uint32_t* __restrict__ data, int64_t n, const uint32_t* __restrict__ sums) {
  __shared__ uint32_t tsum[256];
  const int64_t base = (int64_t)blockIdx.x * SCAN_ITEMS + threadIdx.x * 8;
  uint32_t v[8], s = 0;
  for (int e = 0; e < 8; ++e) { v[e] = (base + e < n) ? data[base + e] : 0; s += v[e]; }
  tsum[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const uint32_t t = threadIdx.x >= off ? tsum[threadIdx.x - off] : 0;
    __syncthreads();
    tsum[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = sums[blockIdx.x] + tsum[threadIdx.x] - s;
  for (int e = 0; e < 8; ++e) {
    if (base + e < n) data[base + e] = run;
    run += v[e];
  }
}

__device__ __forceinline__ float edge_offset(float v1, float v2, float iso) {   // CudaKernels.cu:304-313
  const double delta = (double)(v2 - v1);
  if (delta == 0.0) return 0.5f;
  return (float)((double)(iso - v1) / delta);
}

// cube edge -> (di, dj, dk, dir) of the lattice edge it lies on
__constant__ int8_t dEdgeBase[12][4] = {{0, 0, 0, 0}, {1, 0, 0, 1}, {0, 1, 0, 0}, {0, 0, 0, 1}, {0, 0, 1, 0}, {1, 0, 1, 1},
                                        {0, 1, 1, 0}, {0, 0, 1, 1}, {0, 0, 0, 2}, {1, 0, 0, 2}, {1, 1, 0, 2}, {0, 1, 0, 2}};

__global__ __launch_bounds__(256) void mc_emit_kernel(const float* __restrict__ sdf, Dim d, float iso, const uint32_t* __restrict__ voff,
                                                       const uint32_t* __restrict__ toff, float sx, float sy, float sz, float ox,
                                                       float oy, float oz, float* __restrict__ verts, int64_t* __restrict__ faces) {
  const int64_t total = (int64_t)d.NX * d.NY * d.NZ;
  const int64_t strideX = (int64_t)d.NY * d.NZ, strideY = d.NZ;
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(c % d.NZ), j = (int)((c / d.NZ) % d.NY), i = (int)(c / ((int64_t)d.NZ * d.NY));
    if (!(i < d.NX - 1 && j < d.NY - 1 && k < d.NZ - 1)) continue;
    float v[8];
    const int idx = cube_case(sdf, c, d.NY, d.NZ, iso, v);
    const int own = owned_mask(v[0], v[1], v[3], v[4], iso);
    if (own) {
#pragma clang fp contract(off)   // keep (a*b)+c unfused: vertex coordinates are then bit-equal to the C oracle
      uint32_t vid = voff[c];
      const float fX = (float)i, fY = (float)j, fZ = (float)k;
      if (own & 1) {   // cube edge 0: v0 -> v1, direction +x
        const float t = edge_offset(v[0], v[1], iso);
        verts[vid * 3 + 0] = (fX + (0.0f + t * 1.0f)) * sx + ox;
        verts[vid * 3 + 1] = (fY + (0.0f + t * 0.0f)) * sy + oy;
        verts[vid * 3 + 2] = (fZ + (0.0f + t * 0.0f)) * sz + oz;
        ++vid;
      }
      if (own & 2) {   // cube edge 3: v3 -> v0, direction -y, starting at (0,1,0)
        const float t = edge_offset(v[3], v[0], iso);
        verts[vid * 3 + 0] = (fX + (0.0f + t * 0.0f)) * sx + ox;
        verts[vid * 3 + 1] = (fY + (1.0f + t * -1.0f)) * sy + oy;
        verts[vid * 3 + 2] = (fZ + (0.0f + t * 0.0f)) * sz + oz;
        ++vid;
      }
      if (own & 4) {   // cube edge 8: v0 -> v4, direction +z
        const float t = edge_offset(v[0], v[4], iso);
        verts[vid * 3 + 0] = (fX + (0.0f + t * 0.0f)) * sx + ox;
        verts[vid * 3 + 1] = (fY + (0.0f + t * 0.0f)) * sy + oy;
        verts[vid * 3 + 2] = (fZ + (0.0f + t * 1.0f)) * sz + oz;
      }
    }
    const int nt = dTriCount[idx];
    if (nt == 0) continue;
    const uint64_t word = dTri[idx];
    int64_t* f = faces + (int64_t)toff[c] * 3;
    for (int t = 0; t < nt; ++t) {
#pragma unroll
      for (int corner = 0; corner < 3; ++corner) {
        const int e = (int)((word >> (4 * (3 * t + corner))) & 0xF);
        const int bi = i + dEdgeBase[e][0], bj = j + dEdgeBase[e][1], bk = k + dEdgeBase[e][2], dir = dEdgeBase[e][3];
        int64_t id = -1;
        if (bi < d.NX - 1 && bj < d.NY - 1 && bk < d.NZ - 1) {   // the owner cell exists
          const int64_t oc = (int64_t)bi * strideX + (int64_t)bj * strideY + bk;
          const int m = owned_mask(sdf[oc], sdf[oc + strideX], sdf[oc + strideY], sdf[oc + 1], iso);
          if (m & (1 << dir)) id = (int64_t)voff[oc] + __popc(m & ((1 << dir) - 1));
        }
        f[t * 3 + (2 - corner)] = id;   // reversed winding, CudaKernels.cu:503
      }
    }
  }
}

int ensure_tables() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return SR_EINVAL;
  if (!g_tables_loaded[dev]) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(dTri), kMcTriWords, sizeof(kMcTriWords)) != hipSuccess) return SR_ELAUNCH;
    if (hipMemcpyToSymbol(HIP_SYMBOL(dTriCount), kMcTriCount, sizeof(kMcTriCount)) != hipSuccess) return SR_ELAUNCH;
    g_tables_loaded[dev] = true;
  }
  return SR_OK;
}

void exclusive_scan(uint32_t* data, int64_t n, uint32_t* sums, uint32_t* total_dev, hipStream_t st) {
  const int nblocks = (int)sr_cdiv(n, SCAN_ITEMS);
  hipLaunchKernelGGL(scan_block_sums, dim3(nblocks), dim3(256), 0, st, data, n, sums);
  hipLaunchKernelGGL(scan_sums_serial, dim3(1), dim3(1024), 0, st, sums, nblocks, total_dev);
  hipLaunchKernelGGL(scan_downsweep, dim3(nblocks), dim3(256), 0, st, data, n, sums);
}
}  // namespace

extern "C" {

int64_t sr_mc_workspace_bytes(int32_t nx, int32_t ny, int32_t nz) {
  if (nx <= 0 || ny <= 0 || nz <= 0) return SR_EINVAL;
  const int64_t n = (int64_t)nx * ny * nz;
  const int64_t nblocks = sr_cdiv(n, SCAN_ITEMS);
  return (2 * n + 2 * nblocks + 4) * (int64_t)sizeof(uint32_t);
}

// Pass 1: classify + scan.  counts_dev[0] = #vertices, counts_dev[1] = #faces (device, 2 x u32).
int sr_mc_count(const float* sdf, int32_t nx, int32_t ny, int32_t nz, float iso, void* workspace, uint32_t* counts_dev, void* stream) {
  if (nx <= 0 || ny <= 0 || nz <= 0) return SR_EINVAL;
  if (!sdf || !workspace || !counts_dev) return SR_EINVAL;
  const int rc = ensure_tables();
  if (rc != SR_OK) return rc;
  const int64_t n = (int64_t)nx * ny * nz;
  const int64_t nblocks = sr_cdiv(n, SCAN_ITEMS);
  uint32_t* vcnt = (uint32_t*)workspace;
  uint32_t* tcnt = vcnt + n;
  uint32_t* vsum = tcnt + n;
  uint32_t* tsum = vsum + nblocks;
  hipStream_t st = (hipStream_t)stream;
  Dim d{nx, ny, nz};
  hipLaunchKernelGGL(mc_classify_kernel, dim3(sr_stream_grid(n, 256)), dim3(256), 0, st, sdf, d, iso, vcnt, tcnt);
  exclusive_scan(vcnt, n, vsum, counts_dev, st);
  exclusive_scan(tcnt, n, tsum, counts_dev + 1, st);
  return sr_launch_status();
}

// Pass 2: emit into exactly-sized outputs (verts [V,3] f32 already scaled v*step+min, faces [F,3] i64).
int sr_mc_emit(const float* sdf, int32_t nx, int32_t ny, int32_t nz, float iso, const void* workspace, float xstep, float ystep,
               float zstep, float xmin, float ymin, float zmin, float* verts, int64_t* faces, void* stream) {
  if (nx <= 0 || ny <= 0 || nz <= 0 || !sdf || !workspace) return SR_EINVAL;
  const int64_t n = (int64_t)nx * ny * nz;
  const uint32_t* voff = (const uint32_t*)workspace;
  const uint32_t* toff = voff + n;
  Dim d{nx, ny, nz};
  hipLaunchKernelGGL(mc_emit_kernel, dim3(sr_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, sdf, d, iso, voff, toff, xstep,
                     ystep, zstep, xmin, ymin, zmin, verts, faces);
  return sr_launch_status();
}
}
