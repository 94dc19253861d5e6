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
// coalesced 256-byte runs) and writes ONE byte per voxel (owned-edge mask | triangle count << 3)
// plus two u32 sums per 64-voxel group (a wave = a group; sums come from ballots).  Only the group
// sums are scanned (n/64 elements).  Emit walks the scanned group offsets (8 B per 64 voxels),
// skips empty groups wave-uniformly and touches the volume again only for surface cells; ids of
// neighbour-owned vertices are group offset + popcount over the group's 64 code bytes.  The
// reference rewrites a 12 B/voxel edge table on every call (K8).  HBM-bound: ~5.3 B/voxel.
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

// lane i <- lane i+1 across the whole wave (DPP wave_shl:1); lane 63 takes lane 0 of `next`
__device__ __forceinline__ float from_next_lane(float x, float next) {
  const int last = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, next));
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(last, __builtin_bit_cast(int, x), 0x130, 0xf, 0xf, false));
}

// Classification of a 64-cell group is kept as six 64-bit planes (bit l = cell l): planes 0-2 the owned-edge crossing
// mask (x, y, z lattice edge at the cell origin), planes 3-5 the triangle count (0..5).
// A wave owns MC_CHUNK consecutive groups: the (i, j, k) split of the flat index costs one pair of divisions per chunk
// and is carried forward by addition; the next group's k-plane values are loaded one step ahead, and the k+1 plane of a
// cell is the k plane of the next lane (one DPP move), so a cell costs 4 coalesced loads instead of 8.  Corner signs are
// wave-wide compare masks, so the owned-edge planes are scalar XORs and a group without a sign change never touches
// the case table.
constexpr int MC_CHUNK = 16;
constexpr int MC_BATCH = 4;      // groups whose loads are in flight together (MC_CHUNK % MC_BATCH == 0)
constexpr int MC_PLANES = 6;

__global__ __launch_bounds__(256) void mc_classify_kernel(const float* __restrict__ sdf, Dim d, float iso, uint64_t* __restrict__ planes,
                                                           uint32_t* __restrict__ vs64, uint32_t* __restrict__ ts64) {
  const int64_t total = (int64_t)d.NX * d.NY * d.NZ;
  const int64_t G = (total + 63) >> 6;
  const int64_t sx = (int64_t)d.NY * d.NZ, sy = d.NZ;
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0) { vs64[G] = 0; ts64[G] = 0; }   // scan sentinel: off[G] = total
  // XCD-aware order (gridDim.x is a multiple of 8, block b runs on XCD b % 8): each XCD sweeps its own contiguous eighth of
  // the volume, so the rows j+1 and planes i+1 a chunk shares with its neighbours are found in that XCD's L2
  const int64_t nchunks = (G + MC_CHUNK - 1) / MC_CHUNK;
  const int64_t per_xcd = (nchunks + 7) / 8;
  const int xcd = blockIdx.x & 7;
  const int64_t lwave = (int64_t)(blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6), lwaves = (int64_t)(gridDim.x >> 3) * (blockDim.x >> 6);
  const int64_t chunk_end = (xcd + 1) * per_xcd < nchunks ? (xcd + 1) * per_xcd : nchunks;
  for (int64_t chunk = xcd * per_xcd + lwave; chunk < chunk_end; chunk += lwaves) {
    const int64_t g0 = chunk * MC_CHUNK;
    int64_t c = (g0 << 6) + lane;
    const int64_t row = c / d.NZ;
    int k = (int)(c - row * d.NZ), i = (int)(row / d.NY);
    int j = (int)(row - (int64_t)i * d.NY);
    const int64_t gend = (g0 + MC_CHUNK < G) ? g0 + MC_CHUNK : G;
    // MC_BATCH groups are in flight at a time: the 4 x MC_BATCH row loads of the next groups are issued before the first of them is
    // used (one group ahead -- 4 loads per wave in flight -- left the kernel at 1 TB/s, bound by memory latency, not bandwidth).
    // q[b] = (k, j, i, interior-row flag) and a[b][0..3] = the four k-plane values of group g + b; slot MC_BATCH of a batch is slot
    // 0 of the next one (and lane 63's k+1 neighbour of the batch's last group).
    int qk[MC_BATCH + 1], qj[MC_BATCH + 1], qi[MC_BATCH + 1];
    bool qok[MC_BATCH + 1];
    float a[MC_BATCH + 1][4];
    qk[0] = k; qj[0] = j; qi[0] = i;
    qok[0] = i < d.NX - 1 && j < d.NY - 1;       // c >= total implies i >= NX
#pragma unroll
    for (int e = 0; e < 4; ++e) a[0][e] = 0.f;
    if (qok[0]) { a[0][0] = sdf[c]; a[0][1] = sdf[c + sx]; a[0][2] = sdf[c + sx + sy]; a[0][3] = sdf[c + sy]; }
    for (int64_t g = g0; g < gend; g += MC_BATCH, c += 64 * MC_BATCH) {
#pragma unroll
      for (int b = 1; b <= MC_BATCH; ++b) {
        int nk = qk[b - 1] + 64, nj = qj[b - 1], ni = qi[b - 1];
        while (nk >= d.NZ) { nk -= d.NZ; if (++nj == d.NY) { nj = 0; ++ni; } }
        qk[b] = nk; qj[b] = nj; qi[b] = ni;
        qok[b] = ni < d.NX - 1 && nj < d.NY - 1 && g + b <= gend;     // (groups past the chunk's last "next" group are never used)
        const int64_t cb = c + 64 * b;
#pragma unroll
        for (int e = 0; e < 4; ++e) a[b][e] = 0.f;
        if (qok[b]) { a[b][0] = sdf[cb]; a[b][1] = sdf[cb + sx]; a[b][2] = sdf[cb + sx + sy]; a[b][3] = sdf[cb + sy]; }
      }
#pragma unroll
      for (int b = 0; b < MC_BATCH; ++b) {
        if (g + b >= gend) break;                                      // (wave-uniform)
        const float a0 = a[b][0], a1 = a[b][1], a2 = a[b][2], a3 = a[b][3];
        const float b0 = from_next_lane(a0, a[b + 1][0]), b1 = from_next_lane(a1, a[b + 1][1]), b2 = from_next_lane(a2, a[b + 1][2]),
                    b3 = from_next_lane(a3, a[b + 1][3]);
        const uint64_t I = __ballot(qok[b] && qk[b] < d.NZ - 1);
        const uint64_t s0 = __ballot(a0 < iso), s1 = __ballot(a1 < iso), s2 = __ballot(a2 < iso), s3 = __ballot(a3 < iso);
        const uint64_t s4 = __ballot(b0 < iso), s5 = __ballot(b1 < iso), s6 = __ballot(b2 < iso), s7 = __ballot(b3 < iso);
        const uint64_t mixed = ((s0 ^ s1) | (s0 ^ s2) | (s0 ^ s3) | (s0 ^ s4) | (s0 ^ s5) | (s0 ^ s6) | (s0 ^ s7)) & I;
        uint64_t m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0;
        if (mixed) {
          m0 = (s0 ^ s1) & I; m1 = (s0 ^ s3) & I; m2 = (s0 ^ s4) & I;
          const int idx = (a0 < iso ? 1 : 0) | (a1 < iso ? 2 : 0) | (a2 < iso ? 4 : 0) | (a3 < iso ? 8 : 0) | (b0 < iso ? 16 : 0) |
                          (b1 < iso ? 32 : 0) | (b2 < iso ? 64 : 0) | (b3 < iso ? 128 : 0);
          const uint32_t nt = dTriCount[idx];
          m3 = __ballot(nt & 1) & I; m4 = __ballot(nt & 2) & I; m5 = __ballot(nt & 4) & I;
        }
        if (lane < MC_PLANES) {
          const uint64_t mine = lane == 0 ? m0 : lane == 1 ? m1 : lane == 2 ? m2 : lane == 3 ? m3 : lane == 4 ? m4 : m5;
          planes[(g + b) * MC_PLANES + lane] = mine;
        }
        if (lane == 0) {
          vs64[g + b] = __popcll(m0) + __popcll(m1) + __popcll(m2);
          ts64[g + b] = __popcll(m3) + 2 * __popcll(m4) + 4 * __popcll(m5);
        }
      }
      qk[0] = qk[MC_BATCH]; qj[0] = qj[MC_BATCH]; qi[0] = qi[MC_BATCH]; qok[0] = qok[MC_BATCH];
#pragma unroll
      for (int e = 0; e < 4; ++e) a[0][e] = a[MC_BATCH][e];
    }
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

// id of the vertex on lattice edge (cell oc, direction dir), or -1 when that edge is not crossed / not owned
__device__ __forceinline__ int64_t vertex_id(const uint64_t* __restrict__ planes, const uint32_t* __restrict__ voff64, int64_t oc, int dir) {
  const int64_t g = oc >> 6;
  const int l = (int)(oc & 63);
  const uint64_t* P = planes + g * MC_PLANES;
  const uint64_t p0 = P[0], p1 = P[1], p2 = P[2];
  const uint64_t mine = dir == 0 ? p0 : dir == 1 ? p1 : p2;
  if (!((mine >> l) & 1)) return -1;
  const uint64_t lt = (1ull << l) - 1;
  int cnt = __popcll(p0 & lt) + __popcll(p1 & lt) + __popcll(p2 & lt);
  if (dir > 0) cnt += (int)((p0 >> l) & 1);
  if (dir > 1) cnt += (int)((p1 >> l) & 1);
  return (int64_t)voff64[g] + cnt;
}

// Emit, step 1.  One lane walks one group: the scanned offsets say whether the group holds any surface cell, the planes
// say which.  Every output slot is tagged with its producer -- vertex slot v gets key cell*4 + dir (two u32 words in
// verts[v]), face slot f gets key cell*8 + triangle (faces[f][0]) -- so that steps 2 and 3 run one thread per OUTPUT
// with all lanes busy.  The keys live in the output buffers themselves; no extra workspace.
__global__ __launch_bounds__(256) void mc_tag_kernel(Dim d, const uint64_t* __restrict__ planes, const uint32_t* __restrict__ voff64,
                                                      const uint32_t* __restrict__ toff64, uint32_t* __restrict__ vkeys,
                                                      int64_t* __restrict__ faces) {
  const int64_t total = (int64_t)d.NX * d.NY * d.NZ;
  const int64_t G = (total + 63) >> 6;
  for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < G; g += (int64_t)gridDim.x * blockDim.x) {
    uint32_t vid = voff64[g], tid = toff64[g];
    if (voff64[g + 1] == vid && toff64[g + 1] == tid) continue;
    const uint64_t* P = planes + g * MC_PLANES;
    const uint64_t p0 = P[0], p1 = P[1], p2 = P[2], p3 = P[3], p4 = P[4], p5 = P[5];
    uint64_t todo = p0 | p1 | p2 | p3 | p4 | p5;
    while (todo) {
      const int l = __builtin_ctzll(todo);
      todo &= todo - 1;
      const uint64_t c = ((uint64_t)g << 6) + l;
      if ((p0 >> l) & 1) { vkeys[vid * 3] = (uint32_t)(c * 4 + 0); vkeys[vid * 3 + 1] = (uint32_t)((c * 4 + 0) >> 32); ++vid; }
      if ((p1 >> l) & 1) { vkeys[vid * 3] = (uint32_t)(c * 4 + 1); vkeys[vid * 3 + 1] = (uint32_t)((c * 4 + 1) >> 32); ++vid; }
      if ((p2 >> l) & 1) { vkeys[vid * 3] = (uint32_t)(c * 4 + 2); vkeys[vid * 3 + 1] = (uint32_t)((c * 4 + 2) >> 32); ++vid; }
      const int nt = (int)(((p3 >> l) & 1) | (((p4 >> l) & 1) << 1) | (((p5 >> l) & 1) << 2));
      for (int t = 0; t < nt; ++t) faces[(int64_t)(tid + t) * 3] = (int64_t)(c * 8 + t);
      tid += nt;
    }
  }
}

// Emit, step 2: one thread per vertex.
__global__ __launch_bounds__(256) void mc_vertex_kernel(const float* __restrict__ sdf, Dim d, float iso, const uint32_t* __restrict__ voff64,
                                                         float sx, float sy, float sz, float ox, float oy, float oz, float* __restrict__ verts) {
  const int64_t total = (int64_t)d.NX * d.NY * d.NZ;
  const int64_t V = voff64[(total + 63) >> 6];
  const int64_t strideX = (int64_t)d.NY * d.NZ, strideY = d.NZ;
  const uint32_t* vkeys = reinterpret_cast<const uint32_t*>(verts);
  for (int64_t vid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; vid < V; vid += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = (uint64_t)vkeys[vid * 3] | ((uint64_t)vkeys[vid * 3 + 1] << 32);
    const int64_t c = (int64_t)(key >> 2);
    const int dir = (int)(key & 3);
    const int64_t row = c / d.NZ;
    const int k = (int)(c - row * d.NZ), i = (int)(row / d.NY);
    const int j = (int)(row - (int64_t)i * d.NY);
    const float v0 = sdf[c];
    const float vn = sdf[c + (dir == 0 ? strideX : dir == 1 ? strideY : 1)];
    // Lattice coordinate of the crossing as CudaKernels.cu:363-365 writes it, fX + (offset + t * dir): the products are exact
    // (dir is 0 or +-1), so they are folded here.  The scaling v*step+min of d_scale_vertices (:517-519) is ONE fused
    // multiply-add, as nvcc's default -fmad=true compiles it: bit-equal to the reference's kernels built that way
    // (oracle/_ref/libmc_ref_fma.so, tests/test_mc_reference_pin.py).
    float x = (float)i, y = (float)j, z = (float)k;
    if (dir == 0) {          // cube edge 0: v0 -> v1, direction +x
      x = x + (0.0f + edge_offset(v0, vn, iso));
    } else if (dir == 1) {   // cube edge 3: v3 -> v0, direction -y, starting at (0,1,0)
      y = y + (1.0f + -edge_offset(vn, v0, iso));
    } else {                 // cube edge 8: v0 -> v4, direction +z
      z = z + (0.0f + edge_offset(v0, vn, iso));
    }
    x = __builtin_fmaf(x, sx, ox); y = __builtin_fmaf(y, sy, oy); z = __builtin_fmaf(z, sz, oz);
    verts[vid * 3 + 0] = x; verts[vid * 3 + 1] = y; verts[vid * 3 + 2] = z;
  }
}

// Emit, step 3: one thread per triangle.
__global__ __launch_bounds__(256) void mc_face_kernel(const float* __restrict__ sdf, Dim d, float iso, const uint64_t* __restrict__ planes,
                                                       const uint32_t* __restrict__ voff64, const uint32_t* __restrict__ toff64,
                                                       int64_t* __restrict__ faces) {
  const int64_t total = (int64_t)d.NX * d.NY * d.NZ;
  const int64_t F = toff64[(total + 63) >> 6];
  const int64_t strideX = (int64_t)d.NY * d.NZ, strideY = d.NZ;
  for (int64_t fid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; fid < F; fid += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t key = (uint64_t)faces[fid * 3];
    const int64_t c = (int64_t)(key >> 3);
    const int t = (int)(key & 7);
    float v[8];
    const int idx = cube_case(sdf, c, d.NY, d.NZ, iso, v);
    const uint64_t tri = dTri[idx];
    int64_t out[3];
#pragma unroll
    for (int corner = 0; corner < 3; ++corner) {
      const int e = (int)((tri >> (4 * (3 * t + corner))) & 0xF);
      const int64_t oc = c + dEdgeBase[e][0] * strideX + dEdgeBase[e][1] * strideY + dEdgeBase[e][2];
      // boundary cells have empty planes, so an edge whose owner cell does not exist resolves to -1 as in the reference
      out[2 - corner] = vertex_id(planes, voff64, oc, dEdgeBase[e][3]);   // reversed winding, CudaKernels.cu:503
    }
    faces[fid * 3 + 0] = out[0]; faces[fid * 3 + 1] = out[1]; faces[fid * 3 + 2] = out[2];
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

// workspace: planes u64[6 G] | vs64 u32[G+1] | ts64 u32[G+1] | block sums u32[2*nblocks]   (G = ceil(n / 64) groups)
struct McLayout { int64_t G, nblocks; };
static McLayout mc_layout(int32_t nx, int32_t ny, int32_t nz) {
  McLayout L;
  const int64_t n = (int64_t)nx * ny * nz;
  L.G = (n + 63) >> 6;
  L.nblocks = sr_cdiv(L.G + 1, SCAN_ITEMS);
  return L;
}

int64_t sr_mc_workspace_bytes(int32_t nx, int32_t ny, int32_t nz) {
  if (nx <= 0 || ny <= 0 || nz <= 0) return SR_EINVAL;
  const McLayout L = mc_layout(nx, ny, nz);
  return L.G * MC_PLANES * (int64_t)sizeof(uint64_t) + (2 * (L.G + 1) + 2 * L.nblocks + 4) * (int64_t)sizeof(uint32_t);
}

// Pass 1: classify + scan.  counts_dev[0] = #vertices, counts_dev[1] = #faces (device, 2 x u32).
int sr_mc_count(const float* sdf, int32_t nx, int32_t ny, int32_t nz, float iso, void* workspace, uint32_t* counts_dev, void* stream) {
  if (nx <= 0 || ny <= 0 || nz <= 0) return SR_EINVAL;
  if (!sdf || !workspace || !counts_dev || ((uintptr_t)workspace & 7)) return SR_EINVAL;
  const int rc = ensure_tables();
  if (rc != SR_OK) return rc;
  const McLayout L = mc_layout(nx, ny, nz);
  uint64_t* planes = (uint64_t*)workspace;
  uint32_t* vs64 = (uint32_t*)(planes + L.G * MC_PLANES);
  uint32_t* ts64 = vs64 + (L.G + 1);
  uint32_t* vsum = ts64 + (L.G + 1);
  uint32_t* tsum = vsum + L.nblocks;
  hipStream_t st = (hipStream_t)stream;
  Dim d{nx, ny, nz};
  hipLaunchKernelGGL(mc_classify_kernel, dim3((sr_stream_grid(sr_cdiv(L.G, MC_CHUNK) * 64, 256) + 7) & ~7), dim3(256), 0, st, sdf, d, iso, planes, vs64, ts64);
  exclusive_scan(vs64, L.G + 1, vsum, counts_dev, st);
  exclusive_scan(ts64, L.G + 1, tsum, counts_dev + 1, st);
  return sr_launch_status();
}

// Pass 2: emit into exactly-sized outputs (verts [V,3] f32 already scaled v*step+min, faces [F,3] i64).
int sr_mc_emit(const float* sdf, int32_t nx, int32_t ny, int32_t nz, float iso, const void* workspace, float xstep, float ystep,
               float zstep, float xmin, float ymin, float zmin, float* verts, int64_t* faces, void* stream) {
  if (nx <= 0 || ny <= 0 || nz <= 0 || !sdf || !workspace || !verts || !faces) return SR_EINVAL;
  const McLayout L = mc_layout(nx, ny, nz);
  const uint64_t* planes = (const uint64_t*)workspace;
  const uint32_t* voff64 = (const uint32_t*)(planes + L.G * MC_PLANES);
  const uint32_t* toff64 = voff64 + (L.G + 1);
  hipStream_t st = (hipStream_t)stream;
  Dim d{nx, ny, nz};
  hipLaunchKernelGGL(mc_tag_kernel, dim3(sr_stream_grid(L.G, 256)), dim3(256), 0, st, d, planes, voff64, toff64, (uint32_t*)verts, faces);
  // V and F live on the device (off[G]); the per-output kernels bound their grid-stride loops by them
  const int out_grid = sr_stream_grid(L.G * 8 < (int64_t)1 << 22 ? L.G * 8 : (int64_t)1 << 22, 256);
  hipLaunchKernelGGL(mc_vertex_kernel, dim3(out_grid), dim3(256), 0, st, sdf, d, iso, voff64, xstep, ystep, zstep, xmin, ymin, zmin, verts);
  hipLaunchKernelGGL(mc_face_kernel, dim3(out_grid), dim3(256), 0, st, sdf, d, iso, planes, voff64, toff64, faces);
  return sr_launch_status();
}
}
