// Layer kernels of the three MLPs (SDF a2/a3, deformer a4, render a10): exact-fp32 MFMA GEMMs
// (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD = the 157.3 TFLOP/s fp32 roof) with the layer's
// elementwise work fused into the epilogue.
//
// Row layout ("tangent-interleaved rows"): a sample owns `group` = 1, 2 or 4 consecutive rows
// -- its primal activation followed by up to 3 forward-mode tangents.  The 32x32 MFMA
// accumulator gives every lane 4 consecutive rows of one column (rows 4q..4q+3 of a quad), so
// a sample's primal and tangents sit in ONE lane: activation derivatives (softplus'/softplus''
// or the ReLU mask) are applied in registers, with no cross-lane traffic and no extra pass
// over HBM.  Forward, backward-data and the second-order terms all run through the same NT
// kernel (backward-data uses the pre-transposed weight); the weight gradient is a split-R TN
// kernel with a deterministic slab reduction.
//
// Tiling: 128x128x32 per 256-thread workgroup (2x2 waves of 64x64 = 2x2 MFMA blocks, 64
// accumulator VGPRs), operands staged through LDS in [row][32+4] images -- a 144-byte row pitch
// makes the ds_read_b128 fragment reads bank-conflict free -- double buffered in LDS plus one
// register stage: tile t+2 is in flight from global memory while tile t+1 sits in the other LDS
// buffer and tile t feeds the MFMAs; the one barrier per tile sits in the middle of the MFMA stream.
// Operand tiles of the K % 32 == 0 layers are fetched with buffer loads (descriptor base + constant
// lane offset + scalar step offset): the tile loop holds MFMAs and memory instructions only.
#include "sr_common.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// 16-byte load through a buffer descriptor: `buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen` -- the address is descriptor base (4
// SGPRs, fixed per tile) + a 32-bit per-lane byte offset that never changes + a 32-bit SCALAR byte offset that advances with the step:
// no vector arithmetic at all in front of a tile-loop load (the flat form needs a 64-bit per-lane add per load and step).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sr_make_rsrc(const void* base, int64_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes < 0xFFFFFFFFll ? bytes : 0xFFFFFFFFll), 0x00020000);
}
__device__ __forceinline__ f32x4 sr_buffer_load16(__amdgpu_buffer_rsrc_t rsrc, uint32_t lane_off, uint32_t scalar_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off, (int)scalar_off, 0));
}

namespace {

constexpr int BK = 32;        // K per tile step
constexpr int LDSP = BK + 4;  // LDS row pitch in floats (144 B: 16-B aligned, conflict-free b128 reads)

// Softplus(beta=100, threshold=20) on the hardware exp2/log2 units (v_exp_f32 / v_log_f32, ~1 ulp).  log1p is
// kept accurate for small e = exp(100 z) by its series; elsewhere 1+e rounds with < 6e-8 absolute error in the
// logarithm, i.e. < 6e-10 in the activation -- three orders below the parity tolerance.  The libm versions cost
// ~17% of a K=512 layer in the epilogue; these cost ~3%.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float softplus100(float z) {
  const float t = z * 100.0f;
  const float e = fast_exp(fminf(t, 20.0f));
  const float l = e < 1e-3f ? e * (1.0f - e * (0.5f - e * 0.33333334f)) : fast_log(1.0f + e);
  return t > 20.0f ? z : l * 0.01f;          // selects, not branches; * 0.01f instead of an IEEE division (1 ulp)
}
__device__ __forceinline__ float dsoftplus100(float z) {  // torch: z*beta > threshold ? 1 : e/(e+1)
  const float t = z * 100.0f;
  const float e = fast_exp(fminf(t, 20.0f));
  return t > 20.0f ? 1.0f : e * __builtin_amdgcn_rcpf(e + 1.0f);
}
// derivative factors of an activation from its STORED value a (= aux_scale * sigma(z)), x = 100 a / aux_scale:
// sigma' = 1 - e^{-x}, sigma''/sigma' = 100 e^{-x} (exact identities for softplus(beta = 100))
__device__ __forceinline__ void softplus100_from_stored(float x, float& d, float& c2) {
  const float em = fast_exp(-x);
  d = x < 1e-3f ? x * (1.0f - x * (0.5f - x * 0.16666667f)) : 1.0f - em;
  c2 = 100.0f * em;
}

template <int WM, int WN, int TM, int TN>
struct Cfg {
  static constexpr int kWaves = WM * WN;
  static constexpr int kThreads = kWaves * 64;
  static constexpr int BM = WM * TM * 32;
  static constexpr int BN = WN * TN * 32;
  static constexpr int kALoads = BM * (BK / 4) / kThreads;  // float4 per thread per tile
  static constexpr int kBLoads = (BN * (BK / 4) + kThreads - 1) / kThreads;
  static constexpr int kOperandFloats = 2 * (BM + BN) * LDSP;            // two LDS buffers of A and B tiles
  static constexpr int kStageFloats = kWaves * TM * 32 * (TN * 32 + 4);  // epilogue staging image, one per wave
  static constexpr int kLdsFloats = kOperandFloats > kStageFloats ? kOperandFloats : kStageFloats;
};

// Operand tiles move global -> registers -> LDS as one float4 per thread and slot.  Everything is branch-free so the tile loop is
// ONE basic block the scheduler can interleave with the MFMA stream: rows outside [0,R) are clamped to a valid row (their
// products land in accumulator rows / columns the epilogue never stores), and the K tail is zeroed with selects.
template <int ROWS, int NLOADS, int THREADS>
struct TileLoader {
  const float* p[NLOADS];   // row base + kq*4 of each slot (row clamped)
  __amdgpu_buffer_rsrc_t rsrc;   // (KTAIL = false) descriptor of the rows from the tile's first one on; off[j] = this thread's byte offset
  uint32_t off[NLOADS];
  int kq4;                  // this thread's k offset inside a tile (same for all slots: THREADS % 8 == 0)
  int K, kmax;              // logical K and the last float4 start that stays inside the padded row
  __device__ __forceinline__ TileLoader(const float* __restrict__ P, int64_t ld, int R, int K_, int r0) : K(K_) {
    kq4 = (threadIdx.x & 7) * 4;
    kmax = ((K_ + 3) & ~3) - 4;
    const int rb = r0 < R ? r0 : R - 1;
    rsrc = sr_make_rsrc(P + (int64_t)rb * ld, (int64_t)(R - rb) * ld * 4);
#pragma unroll
    for (int j = 0; j < NLOADS; ++j) {
      int row = (threadIdx.x + j * THREADS) >> 3;
      if (row >= ROWS) row = ROWS - 1;
      int gr = r0 + row;
      if (gr >= R) gr = R - 1;
      p[j] = P + (int64_t)gr * ld;
      off[j] = (uint32_t)(((int64_t)(gr - rb) * ld + kq4) * 4);      // < ROWS * ld * 4 bytes
    }
  }
  // raw loads; the K tail is masked when the registers are written to LDS (a select right after the load would make the
  // wave wait for the data in the same iteration)
  __device__ __forceinline__ void load(int k0, f32x4 (&reg)[NLOADS]) const {
    const int gk = k0 + kq4;
    const int gkc = gk < kmax ? gk : kmax;
#pragma unroll
    for (int j = 0; j < NLOADS; ++j) reg[j] = *reinterpret_cast<const f32x4*>(p[j] + gkc);
  }
  // K % 32 == 0 (no float4 reaches past K): buffer loads, the step's k offset is the scalar offset.  A prefetch past the last tile
  // re-reads the last one.
  __device__ __forceinline__ void load_full(int k0, int klast, f32x4 (&reg)[NLOADS]) const {
    const uint32_t kc = (uint32_t)(k0 < klast ? k0 : klast) * 4u;
#pragma unroll
    for (int j = 0; j < NLOADS; ++j) reg[j] = sr_buffer_load16(rsrc, off[j], kc);
  }
  // KTAIL = false: K is a multiple of BK (every 512-wide layer), no float4 reaches past K and the four selects per slot -- 32 VALU
  // operations per thread and step, next to 64 MFMAs -- are not compiled in.
  template <bool KTAIL = true>
  __device__ __forceinline__ void store(float* __restrict__ lds, int k0, const f32x4 (&reg)[NLOADS]) const {
    const int nvalid = K - (k0 + kq4);      // floats of this float4 inside [0,K): >= 4 keeps all, <= 0 keeps none
#pragma unroll
    for (int j = 0; j < NLOADS; ++j) {
      const int idx = threadIdx.x + j * THREADS;
      const int row = idx >> 3;
      f32x4 v = reg[j];
      if (KTAIL) { v.x = nvalid > 0 ? v.x : 0.f; v.y = nvalid > 1 ? v.y : 0.f; v.z = nvalid > 2 ? v.z : 0.f; v.w = nvalid > 3 ? v.w : 0.f; }
      if (ROWS * 8 % THREADS == 0 || row < ROWS) *reinterpret_cast<f32x4*>(lds + row * LDSP + kq4) = v;
    }
  }
};

// Epilogue for one wave: every lane holds, per 32x32 block, 4 quads of 4 consecutive rows (one column).
// G rows form one sample (primal + G-1 tangents); all per-sample maths is in-register.
template <int WM, int WN, int TM, int TN, int G>
__device__ __forceinline__ void epilogue(const sr_gemm_args& g, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                         int li, int kh, float* __restrict__ stage) {
  constexpr int SP = TN * 32 + 4;      // pitch of the per-wave staging image in LDS
  const int ncols = g.N + g.naux_fwd;  // columns of C this launch produces
  const float inv_aux_scale = 1.0f / g.aux_scale;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + (wn * TN + b) * 32 + li;
      if (col >= ncols) continue;
      const float bias = (g.bias && col < g.N) ? g.bias[col] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r0 = m0 + (wm * TM + a) * 32 + 8 * q + 4 * kh;  // rows r0..r0+3 live in this lane
        if (r0 >= g.M) continue;
        float v[4] = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
        float o[4];
        if (g.mode == SR_EPI_FWD) {
          if (col >= g.N) {  // skip-concat filler: C[:, N + j] = aux[:, j] * out_scale
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (r0 + j < g.M) ? g.aux[(int64_t)(r0 + j) * g.ldaux + (col - g.N)] * g.out_scale : 0.f;
          } else {
#pragma unroll
            for (int s = 0; s < 4; s += G) {  // one sample: primal row s, tangents s+1..s+G-1
              const float z = v[s] + bias;
              float d;
              if (g.act == SR_ACT_SOFTPLUS100) { o[s] = softplus100(z) * g.out_scale; d = (G > 1) ? dsoftplus100(z) : 0.f; }
              else if (g.act == SR_ACT_RELU) { o[s] = fmaxf(z, 0.f) * g.out_scale; d = z > 0.f ? 1.f : 0.f; }
              else { o[s] = z * g.out_scale; d = 1.f; }
#pragma unroll
              for (int tI = 1; tI < G; ++tI) o[s + tI] = d * v[s + tI] * g.out_scale;
            }
          }
        } else {  // SR_EPI_BWD: acc = cotangent of the STORED activations; emit cotangent of pre-activations
          if (col >= g.nact_bwd) {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = v[j] * g.out_scale;
          } else {
            float sv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) sv[j] = (r0 + j < g.M) ? g.aux[(int64_t)(r0 + j) * g.ldaux + col] : 0.f;
#pragma unroll
            for (int s = 0; s < 4; s += G) {
              float d, c2;
              if (g.act == SR_ACT_SOFTPLUS100) {
                // stored = aux_scale * softplus(z);  1 + e^{100 z} = e^{100 a}  =>  sigma' = 1 - e^{-100 a},
                // sigma''/sigma' = 100 e^{-100 a}  (both exact identities, no division by sigma')
                softplus100_from_stored(100.0f * (sv[s] * inv_aux_scale), d, c2);
              } else if (g.act == SR_ACT_RELU) { d = sv[s] > 0.f ? 1.f : 0.f; c2 = 0.f; }
              else { d = 1.f; c2 = 0.f; }
              float cross = 0.f;
#pragma unroll
              for (int tI = 1; tI < G; ++tI) {
                cross += sv[s + tI] * v[s + tI];           // stored tangent activation x its cotangent
                o[s + tI] = d * g.aux_scale * v[s + tI];
              }
              o[s] = d * g.aux_scale * v[s] + c2 * cross;
            }
          }
        }
        // results go to the wave's LDS image; rows leave as 16-byte stores below (the MFMA layout would give
        // 64 dword stores per lane -- store-issue bound, and 2-3x slower on the K=39/K=167 first layers)
        float* sp = stage + (a * 32 + 8 * q + 4 * kh) * SP + b * 32 + li;
#pragma unroll
        for (int j = 0; j < 4; ++j) sp[j * SP] = o[j];
      }
    }
  }
  __syncthreads();
  constexpr int QPR = TN * 8;          // float4 per row of the wave tile
  constexpr int RPI = 64 / QPR;        // rows covered by one wave-wide store
  const int lane = kh * 32 + li;
  const int qc = lane % QPR, ro = lane / QPR;
  const int col0 = n0 + wn * TN * 32 + qc * 4;
#pragma unroll
  for (int i = 0; i < TM * 32 / RPI; ++i) {
    const int lr = i * RPI + ro;
    const int row = m0 + wm * TM * 32 + lr;
    if (row >= g.M || col0 >= ncols) continue;
    const f32x4 v = *reinterpret_cast<const f32x4*>(stage + lr * SP + qc * 4);
    float* dst = g.C + (int64_t)row * g.ldc + col0;
    if (col0 + 3 < ncols) {
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {
      dst[0] = v.x;
      if (col0 + 1 < ncols) dst[1] = v.y;
      if (col0 + 2 < ncols) dst[2] = v.z;
    }
  }
}

// The same epilogue for a tile that lies completely inside the matrix and inside the activated columns -- almost every tile of
// a launch.  Mode, activation and group size are template parameters and nothing is bounds-checked: the generic version
// above spends ~3.5k instructions and ~340 branches per wave on a 128x128 tile (10-18 % of a K = 512 layer); this one is a
// straight line.
template <int WM, int WN, int TM, int TN, int G, int MODE, int ACT>
__device__ __forceinline__ void epilogue_interior(const sr_gemm_args& g, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int li,
                                                  int kh, float* __restrict__ stage) {
  constexpr int SP = TN * 32 + 4;
  const float os = g.out_scale, as = g.aux_scale, inv_as = 1.0f / g.aux_scale;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + (wn * TN + b) * 32 + li;
      const float bias = (MODE == SR_EPI_FWD && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r0 = m0 + (wm * TM + a) * 32 + 8 * q + 4 * kh;
        const float v[4] = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
        float o[4];
        if constexpr (MODE == SR_EPI_FWD) {
#pragma unroll
          for (int s = 0; s < 4; s += G) {
            const float z = v[s] + bias;
            float d = 1.f;
            if constexpr (ACT == SR_ACT_SOFTPLUS100) { o[s] = softplus100(z) * os; if (G > 1) d = dsoftplus100(z); }
            else if constexpr (ACT == SR_ACT_RELU) { o[s] = fmaxf(z, 0.f) * os; d = z > 0.f ? 1.f : 0.f; }
            else { o[s] = z * os; }
#pragma unroll
            for (int tI = 1; tI < G; ++tI) o[s + tI] = d * v[s + tI] * os;
          }
        } else {
          float sv[4];
          if constexpr (ACT != SR_ACT_NONE || G > 1) {
            const float* ap = g.aux + (int64_t)r0 * g.ldaux + col;
#pragma unroll
            for (int j = 0; j < 4; ++j) sv[j] = ap[(int64_t)j * g.ldaux];
          }
#pragma unroll
          for (int s = 0; s < 4; s += G) {
            float d = 1.f, c2 = 0.f;
            if constexpr (ACT == SR_ACT_SOFTPLUS100) softplus100_from_stored(100.0f * (sv[s] * inv_as), d, c2);
            else if constexpr (ACT == SR_ACT_RELU) d = sv[s] > 0.f ? 1.f : 0.f;
            float cross = 0.f;
#pragma unroll
            for (int tI = 1; tI < G; ++tI) {
              cross += sv[s + tI] * v[s + tI];
              o[s + tI] = d * as * v[s + tI];
            }
            o[s] = d * as * v[s] + c2 * cross;
          }
        }
        float* sp = stage + (a * 32 + 8 * q + 4 * kh) * SP + b * 32 + li;
#pragma unroll
        for (int j = 0; j < 4; ++j) sp[j * SP] = o[j];
      }
    }
  }
  __syncthreads();
  constexpr int QPR = TN * 8, RPI = 64 / QPR;
  const int lane = kh * 32 + li;
  const int qc = lane % QPR, ro = lane / QPR;
  float* dst = g.C + (int64_t)(m0 + wm * TM * 32 + ro) * g.ldc + n0 + wn * TN * 32 + qc * 4;
#pragma unroll
  for (int i = 0; i < TM * 32 / RPI; ++i)
    *reinterpret_cast<f32x4*>(dst + (int64_t)i * RPI * g.ldc) = *reinterpret_cast<const f32x4*>(stage + (i * RPI + ro) * SP + qc * 4);
}

template <int WM, int WN, int TM, int TN, int G, int MODE>
__device__ __forceinline__ void epilogue_interior_act(const sr_gemm_args& g, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int li,
                                                      int kh, float* __restrict__ stage) {
  if (g.act == SR_ACT_SOFTPLUS100) epilogue_interior<WM, WN, TM, TN, G, MODE, SR_ACT_SOFTPLUS100>(g, acc, m0, n0, wm, wn, li, kh, stage);
  else if (g.act == SR_ACT_RELU) epilogue_interior<WM, WN, TM, TN, G, MODE, SR_ACT_RELU>(g, acc, m0, n0, wm, wn, li, kh, stage);
  else epilogue_interior<WM, WN, TM, TN, G, MODE, SR_ACT_NONE>(g, acc, m0, n0, wm, wn, li, kh, stage);
}

// ------------------------------------------------------------------------------------------------
// C = epilogue(A[M,K] * B[N,K]^T)
// One output tile (workgroup-wide).  `wg` is the linear tile index of this launch / layer; `smem` the workgroup's LDS
// (Cfg::kLdsFloats floats).  Called once per workgroup by gemm_nt_kernel and by mlp_layer_pair_kernel (row count from device memory).
// (`probe(i)`: diagnostics hook of tools/nt_lab.hip -- cycle stamps at the end of the prologue (0), of the tile loop (1) and of the
// epilogue (2); the product kernels pass the empty default, which compiles to nothing.)
struct NoProbe { __device__ __forceinline__ void operator()(int) const {} };
template <int WM, int WN, int TM, int TN, bool KTAIL = true, class Probe = NoProbe>
__device__ __forceinline__ void gemm_nt_tile(const sr_gemm_args& g, int wg, float* __restrict__ smem, Probe probe = Probe()) {
  using C_ = Cfg<WM, WN, TM, TN>;
  auto As = [&](int buf) -> float* { return smem + buf * (C_::BM * LDSP); };
  auto Bs = [&](int buf) -> float* { return smem + 2 * C_::BM * LDSP + buf * (C_::BN * LDSP); };

  // XCD-aware tile order: consecutive workgroups land on different XCDs (block b -> XCD b%8), so give
  // each XCD a contiguous run of M-tiles sharing the same weight panel in its private L2.
  const int tiles_n = (g.N + g.naux_fwd + C_::BN - 1) / C_::BN;
  const int tiles_m = (g.M + C_::BM - 1) / C_::BM;
  const int nwg = tiles_m * tiles_n;
  {
    const int q = nwg / 8, r = nwg % 8, xcd = wg % 8, loc = wg / 8;
    wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
  }
  const int tn = wg % tiles_n, tm = wg / tiles_n;
  const int m0 = tm * C_::BM, n0 = tn * C_::BN;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, kh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // Software-pipelined tile loop with one barrier per tile placed in the MIDDLE of the MFMA stream: the wave holds half a
  // tile of fragments in registers when it meets the barrier.  Loads past the last tile are clamped re-reads, never used.
  const TileLoader<C_::BM, C_::kALoads, C_::kThreads> la(g.A, g.lda, g.M, g.K, m0);
  const TileLoader<C_::BN, C_::kBLoads, C_::kThreads> lb(g.B, g.ldb, g.N, g.K, n0);
  const int nk = (g.K + BK - 1) / BK;
  const int a_off = (wm * TM * 32 + li) * LDSP + kh * 4, b_off = (wn * TN * 32 + li) * LDSP + kh * 4;
  f32x4 fa0[2][TM], fb0[2][TN], fa1[2][TM], fb1[2][TN];   // fragments of kk = 0,1 and kk = 2,3
  auto read_frags = [&](const float* abuf, const float* bbuf, int kk0, f32x4 (&fa)[2][TM], f32x4 (&fb)[2][TN]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[h][a] = *reinterpret_cast<const f32x4*>(abuf + a_off + a * 32 * LDSP + (kk0 + h) * 8);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[h][b] = *reinterpret_cast<const f32x4*>(bbuf + b_off + b * 32 * LDSP + (kk0 + h) * 8);
    }
  };
  auto mfma_kk = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN]) {   // one kk = 8 k-values = 4 x TM x TN MFMAs
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][e], fb[b][e], acc[a][b], 0, 0, 0);
  };
  constexpr int kMem = C_::kALoads + C_::kBLoads;          // float4 slots per thread and tile

  // Two register stages: tile t+2 is requested from global memory at the top of step t and written to LDS at the top of step
  // t+1, so a load has a whole step to land.  Per step the wave issues, in this order and each slotted between two MFMAs
  // (sched_group_barrier): the refill loads, the LDS stores of tile t+1, the kk=2,3 fragment reads of tile t; then the one
  // barrier of the step, the kk=0,1 fragment reads of tile t+1 and the second half of the MFMAs.  Measured with s_memtime
  // stamps (profiles/r01_summary.md): with the memory operations in their own phase the wave spent 1000-1500 clk per
  // 4096-clk step with an idle MFMA pipe.
  f32x4 ra0[C_::kALoads], rb0[C_::kBLoads], ra1[C_::kALoads], rb1[C_::kBLoads];
  const int klast = (nk - 1) * BK;
  auto load_tile = [&](int k0, f32x4 (&ar)[C_::kALoads], f32x4 (&br)[C_::kBLoads]) {
    if constexpr (KTAIL) { la.load(k0, ar); lb.load(k0, br); }
    else { la.load_full(k0, klast, ar); lb.load_full(k0, klast, br); }
  };
  load_tile(0, ra0, rb0);
  load_tile(BK, ra1, rb1);
  la.template store<KTAIL>(As(0), 0, ra0); lb.template store<KTAIL>(Bs(0), 0, rb0);
  __syncthreads();
  read_frags(As(0), Bs(0), 0, fa0, fb0);
  probe(0);
  auto step = [&](int t, f32x4 (&ain)[C_::kALoads], f32x4 (&bin)[C_::kBLoads], const f32x4 (&aout)[C_::kALoads],
                  const f32x4 (&bout)[C_::kBLoads]) {
    const int cur = t & 1;
    load_tile((t + 2) * BK, ain, bin);
    la.template store<KTAIL>(As(cur ^ 1), (t + 1) * BK, aout); lb.template store<KTAIL>(Bs(cur ^ 1), (t + 1) * BK, bout);
    read_frags(As(cur), Bs(cur), 2, fa1, fb1);
    mfma_kk(fa0[0], fb0[0]);
    mfma_kk(fa0[1], fb0[1]);
#pragma unroll
    for (int i = 0; i < kMem; ++i) {        // global loads first (one per MFMA), then the LDS stores, then the fragment reads
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < kMem; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < 2 * (TM + TN); ++i) {
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    read_frags(As(cur ^ 1), Bs(cur ^ 1), 0, fa0, fb0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_kk(fa1[0], fb1[0]);
    mfma_kk(fa1[1], fb1[1]);
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int t = 0; t < nk; t += 2) {
    step(t, ra0, rb0, ra1, rb1);
    if (t + 1 < nk) step(t + 1, ra1, rb1, ra0, rb0);
  }
  __syncthreads();   // the epilogue reuses the operand buffers
  probe(1);

  // ---------------------------------------------------------------- epilogue
  float* stage = smem + wave * (TM * 32 * (TN * 32 + 4));   // the operand buffers are free after the last barrier
  // (block-uniform) tiles completely inside the matrix and inside the activated columns take the straight-line epilogue
  const bool interior = m0 + C_::BM <= g.M && n0 + C_::BN <= (g.mode == SR_EPI_FWD ? g.N : (g.nact_bwd < g.N ? g.nact_bwd : g.N));
  if (interior) {
    if (g.mode == SR_EPI_FWD) {
      switch (g.group) {
        case 1: epilogue_interior_act<WM, WN, TM, TN, 1, SR_EPI_FWD>(g, acc, m0, n0, wm, wn, li, kh, stage); break;
        case 2: epilogue_interior_act<WM, WN, TM, TN, 2, SR_EPI_FWD>(g, acc, m0, n0, wm, wn, li, kh, stage); break;
        default: epilogue_interior_act<WM, WN, TM, TN, 4, SR_EPI_FWD>(g, acc, m0, n0, wm, wn, li, kh, stage); break;
      }
    } else {
      switch (g.group) {
        case 1: epilogue_interior_act<WM, WN, TM, TN, 1, SR_EPI_BWD>(g, acc, m0, n0, wm, wn, li, kh, stage); break;
        case 2: epilogue_interior_act<WM, WN, TM, TN, 2, SR_EPI_BWD>(g, acc, m0, n0, wm, wn, li, kh, stage); break;
        default: epilogue_interior_act<WM, WN, TM, TN, 4, SR_EPI_BWD>(g, acc, m0, n0, wm, wn, li, kh, stage); break;
      }
    }
  } else {
    switch (g.group) {
      case 1: epilogue<WM, WN, TM, TN, 1>(g, acc, m0, n0, wm, wn, li, kh, stage); break;
      case 2: epilogue<WM, WN, TM, TN, 2>(g, acc, m0, n0, wm, wn, li, kh, stage); break;
      default: epilogue<WM, WN, TM, TN, 4>(g, acc, m0, n0, wm, wn, li, kh, stage); break;
    }
  }
  probe(2);
}

template <int WM, int WN, int TM, int TN, bool KTAIL>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(2))) void gemm_nt_kernel(sr_gemm_args g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // (Tried in round 3: delaying the second workgroup of every CU by 0.1 - 0.6 of a tile at the start of a launch, so that the two
  // co-resident workgroups do not run their prologues / epilogues in phase -- no effect, 119.0 +- 0.4 TFLOP/s at every delay.  Round 6's
  // cycle stamps (tools/nt_lab.hip, profiles/r06_nt_lab.md) show why: the two workgroups of a CU ALTERNATE by themselves -- the younger
  // one is starved in its prologue until the older one's tile loop ends -- so a CU completes a tile per (tile loop alone), and what that
  // loop costs beyond its MFMAs is the issue time of its other instructions: hence the buffer loads of TileLoader::load_full.)
  gemm_nt_tile<WM, WN, TM, TN, KTAIL>(g, blockIdx.x, smem);
}

// (Tried in round 4: the same tiles walked by a RESIDENT grid of 512 workgroups with a stride loop -- 135.4 against 135.3 TFLOP/s on
// 262144 x 512 x 512: retiring a workgroup and placing a fresh one is not where the time goes.)

using ChainCfg = Cfg<2, 2, 1, 1>;      // 64x64 tiles: the refiner's few thousand rows give ~100 row panels
// (Rounds 2-5 also carried a one-launch form of a layer chain -- a resident grid walking the tiles of every layer with a device-wide
// barrier between layers.  It lost every measurement: ~35 us per barrier against ~13 us per dependent launch, and a cooperative grid
// cannot get residency next to the 128x128 tiles of the stream it runs beside.  Removed in round 6.)

// ------------------------------------------------------------------------------------------------
// dW[N,K] = sum_r Z[r,N]^T A[r,K]   (split over r into `splits` slabs, reduced below)
constexpr int TBR = 32;          // rows (reduction) per tile step
// Tile of dW: ZW columns of Z (rows of dW) x XW columns of A (columns of dW); 4 waves of 64 x 64 each.  128 x 128 is the general
// shape; 256 x 64 serves the first layers (K = 39: on the square tile 70 % of the A columns were clamped re-reads).
template <int ZW, int XW>
struct TnCfg {
  static constexpr int ZLD = ZW + 4, XLD = XW + 4;                       // LDS pitches of the [TBR][width] images
  static constexpr int kZ = TBR * ZLD, kX = TBR * XLD;
  static constexpr int kLdsFloats = 2 * (kZ + kX);                        // both images double buffered (dynamic LDS)
  static constexpr int TLZ = TBR * (ZW / 4) / 256, TLX = TBR * (XW / 4) / 256;   // float4 loads per thread and step
  static constexpr int WN = XW / 64;                                      // waves along the A columns
};

// One workgroup of a weight-gradient launch: tile and row slab from (`bid`, `nblocks`) -- the workgroup's index and the workgroup count of
// ITS problem, which is the whole grid for gemm_tn_kernel and a slice of it for gemm_tn_group_kernel.
template <int ZW, int XW>
__device__ __forceinline__ void gemm_tn_body(const sr_gemm_tn_args& g, int rows_per_split, int bid, int nblocks, float* __restrict__ tn_smem) {
  using C = TnCfg<ZW, XW>;
  auto Zs = [&](int buf) -> float* { return tn_smem + buf * C::kZ; };
  auto Xs = [&](int buf) -> float* { return tn_smem + 2 * C::kZ + buf * C::kX; };
  const int tiles_k = (g.K + XW - 1) / XW;
  const int tiles_n = (g.N + ZW - 1) / ZW;
  // XCD-aware order: workgroup b runs on XCD b % 8 with a private L2.  The tiles of one row-slice (split) all read the
  // same Z and A rows, so give each XCD whole splits: virtual id = xcd * (grid/8) + b / 8.  (Before: the 4 k-tiles /
  // 4 n-tiles of a slice sat on different XCDs and every operand row was streamed from HBM 4 times.)
  int vb = bid;
  if ((nblocks & 7) == 0) vb = (bid & 7) * (nblocks >> 3) + (bid >> 3);
  const int tile = vb % (tiles_k * tiles_n), split = vb / (tiles_k * tiles_n);
  const int n0 = (tile / tiles_k) * ZW, k0 = (tile % tiles_k) * XW;
  const int r_begin = split * rows_per_split;
  const int r_end = min(g.R, r_begin + rows_per_split);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN, li = lane & 31, kh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // A step moves a [TBR rows][ZW cols] tile of Z and a [TBR][XW] tile of A: TLZ / TLX float4 per thread.  Like the NT
  // loop it is branch-free (one basic block): row addresses are clamped into the slice, rows past its end are zeroed with
  // selects when the registers go to LDS (they are the reduction dimension and must contribute nothing), and columns past
  // N / K are clamped re-reads whose products land in rows / columns of dW that are never stored.
  constexpr int TLZ = C::TLZ, TLX = C::TLX;
  constexpr int ZT = ZW / 4, XT = XW / 4;                                   // threads per image row
  const int cqz = (threadIdx.x % ZT) * 4, lrz = threadIdx.x / ZT;          // this thread's column quad and first row inside a tile
  const int cqx = (threadIdx.x % XT) * 4, lrx = threadIdx.x / XT;
  const int zmax = ((g.N + 3) & ~3) - 4, xmax = ((g.K + 3) & ~3) - 4;
  const float* zcol = g.Z + (n0 + cqz < zmax ? n0 + cqz : zmax);
  const float* xcol = g.A + (k0 + cqx < xmax ? k0 + cqx : xmax);
  f32x4 rz[TLZ], rx[TLX];
  // Two forms of the step, chosen per workgroup (block-uniform):
  //   TAIL = true   the slab's row count is not a multiple of TBR (the last slab of a launch): per-lane row clamp on the loads, per-lane
  //                 zeroing of the rows past the end when the registers go to LDS;
  //   TAIL = false  every step is a full tile: a thread's load address is its base pointer + (uniform row offset) * pitch -- no per-lane
  //                 multiply, clamp or select.  The prefetches past the slab's last tile (never used) re-read that last tile.
  // The ISA of the single form had 192 VALU operations per step next to its 64 MFMAs (16 quarter-rate v_mul_lo_u32, 32 selects, 26
  // bias adds in every workgroup although only the k0 == 0 column of tiles folds a bias gradient): the kernel ran at 118 TFLOP/s where
  // the NT tile code (12 VALU operations per step) reaches 135.
  // (round 6) a full-tile load is a buffer load: descriptor = the slab's rows, the thread's place inside a tile a 32-bit byte offset that
  // never changes, the tile's first row a SCALAR byte offset -- no per-lane 64-bit add in front of every load (the step had 16 of them).
  const __amdgpu_buffer_rsrc_t zrsrc = sr_make_rsrc(g.Z + (int64_t)r_begin * g.ldz, (int64_t)(g.R - r_begin) * g.ldz * 4);
  const __amdgpu_buffer_rsrc_t xrsrc = sr_make_rsrc(g.A + (int64_t)r_begin * g.lda, (int64_t)(g.R - r_begin) * g.lda * 4);
  uint32_t zbyte[TLZ], xbyte[TLX];
#pragma unroll
  for (int j = 0; j < TLZ; ++j) zbyte[j] = (uint32_t)(((int64_t)(lrz + j * (256 / ZT)) * g.ldz + (zcol - g.Z)) * 4);
#pragma unroll
  for (int j = 0; j < TLX; ++j) xbyte[j] = (uint32_t)(((int64_t)(lrx + j * (256 / XT)) * g.lda + (xcol - g.A)) * 4);
  const int last_full = r_end - TBR;                  // start row of the slab's last full tile (TAIL = false)
  auto load = [&](int r0, auto tail) {
    if constexpr (decltype(tail)::value) {
#pragma unroll
      for (int j = 0; j < TLZ; ++j) {
        int gr = r0 + lrz + j * (256 / ZT);
        gr = gr < r_end ? gr : r_end - 1;
        rz[j] = *reinterpret_cast<const f32x4*>(zcol + (int64_t)gr * g.ldz);
      }
#pragma unroll
      for (int j = 0; j < TLX; ++j) {
        int gr = r0 + lrx + j * (256 / XT);
        gr = gr < r_end ? gr : r_end - 1;
        rx[j] = *reinterpret_cast<const f32x4*>(xcol + (int64_t)gr * g.lda);
      }
    } else {
      const int rc = r0 < last_full ? r0 : last_full;                       // (uniform)
      const uint32_t zo = (uint32_t)((int64_t)(rc - r_begin) * g.ldz * 4), xo = (uint32_t)((int64_t)(rc - r_begin) * g.lda * 4);   // (uniform)
#pragma unroll
      for (int j = 0; j < TLZ; ++j) rz[j] = sr_buffer_load16(zrsrc, zbyte[j], zo);
#pragma unroll
      for (int j = 0; j < TLX; ++j) rx[j] = sr_buffer_load16(xrsrc, xbyte[j], xo);
    }
  };
  auto store = [&](int buf, int r0, auto tail) {
    float* zs = Zs(buf);
    float* xs = Xs(buf);
#pragma unroll
    for (int j = 0; j < TLZ; ++j) {
      const int row = lrz + j * (256 / ZT);
      f32x4 z = rz[j];
      if constexpr (decltype(tail)::value) {
        const bool ok = r0 + row < r_end;
        z.x = ok ? z.x : 0.f; z.y = ok ? z.y : 0.f; z.z = ok ? z.z : 0.f; z.w = ok ? z.w : 0.f;
      }
      *reinterpret_cast<f32x4*>(zs + row * C::ZLD + cqz) = z;
    }
#pragma unroll
    for (int j = 0; j < TLX; ++j) {
      const int row = lrx + j * (256 / XT);
      f32x4 x = rx[j];
      if constexpr (decltype(tail)::value) {
        const bool ok = r0 + row < r_end;
        x.x = ok ? x.x : 0.f; x.y = ok ? x.y : 0.f; x.z = ok ? x.z : 0.f; x.w = ok ? x.w : 0.f;
      }
      *reinterpret_cast<f32x4*>(xs + row * C::XLD + cqx) = x;
    }
  };

  // fragments of 8 k-steps (16 rows).  The two 32-wide blocks of a wave's 64 columns are INTERLEAVED -- block a holds the columns
  // 2 i + a (i = 0..31) -- so that one 8-byte LDS read gives a lane the operands of both blocks for a k-step: 32 ds_read_b64 per step
  // where the contiguous blocks (columns i and 32 + i) took 64 ds_read_b32, next to the step's 64 MFMAs.  (Lanes 0..31 of a read cover 64
  // consecutive dwords: conflict-free.)  Only the column index of the final stores knows about the interleave.
  constexpr int HS = TBR / 4;   // k-steps per half tile
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  float z0a[HS], z1a[HS], x0a[HS], x1a[HS], z0b[HS], z1b[HS], x0b[HS], x1b[HS];
  const int zoff = kh * C::ZLD + wm * 64 + 2 * li, xoff = kh * C::XLD + wn * 64 + 2 * li;
  auto read_half = [&](int buf, int half, float (&z0)[HS], float (&z1)[HS], float (&x0)[HS], float (&x1)[HS]) {
    const float* zb = Zs(buf) + zoff + half * HS * 2 * C::ZLD;
    const float* xb = Xs(buf) + xoff + half * HS * 2 * C::XLD;
#pragma unroll
    for (int e = 0; e < HS; ++e) {
      const f32x2 z = *reinterpret_cast<const f32x2*>(zb + 2 * e * C::ZLD), x = *reinterpret_cast<const f32x2*>(xb + 2 * e * C::XLD);
      z0[e] = z.x; z1[e] = z.y; x0[e] = x.x; x1[e] = x.y;
    }
  };
  auto mfma_half = [&](const float (&z0)[HS], const float (&z1)[HS], const float (&x0)[HS], const float (&x1)[HS]) {
#pragma unroll
    for (int e = 0; e < HS; ++e) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(z0[e], x0[e], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(z0[e], x1[e], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(z1[e], x0[e], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(z1[e], x1[e], acc[1][1], 0, 0, 0);
    }
  };

  // bias gradient rides along for free: the Z fragments a lane already holds are column values of rows 2e + kh, so the
  // column sums over the primal rows (row % group == 0; tiles start at multiples of 32) are plain adds on registers --
  // every k-step for group 1, lanes kh == 0 for group 2, and additionally only even k-steps for group 4.
  const bool do_bias = g.db_partial != nullptr && k0 == 0;
  const float wodd = g.group == 4 ? 0.f : 1.f;
  float bsum0 = 0.f, bsum1 = 0.f;
  auto bias_half = [&](const float (&z0)[HS], const float (&z1)[HS]) {
#pragma unroll
    for (int e = 0; e < HS; ++e) {
      if (e & 1) { bsum0 += wodd * z0[e]; bsum1 += wodd * z1[e]; }
      else { bsum0 += z0[e]; bsum1 += z1[e]; }
    }
  };
  const int nsteps = (r_end - r_begin + TBR - 1) / TBR;
  auto run = [&](auto tail, auto bias) {
    load(r_begin, tail);
    store(0, r_begin, tail);
    load(r_begin + TBR, tail);
    __syncthreads();
    read_half(0, 0, z0a, z1a, x0a, x1a);
    // Step t, as in the NT kernel: tile t+1 goes registers -> LDS, the register stage is refilled with tile t+2, the second
    // half's fragments are read -- every memory instruction slotted between two MFMAs of the first half -- then the one
    // barrier, the first-half fragments of tile t+1, and the second half of the MFMAs.
    for (int t = 0; t < nsteps; ++t) {
      const int cur = t & 1;
      store(cur ^ 1, r_begin + (t + 1) * TBR, tail);
      load(r_begin + (t + 2) * TBR, tail);
      read_half(cur, 1, z0b, z1b, x0b, x1b);
      mfma_half(z0a, z1a, x0a, x1a);
#pragma unroll
      for (int i = 0; i < TLZ + TLX; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < TLZ + TLX; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < HS; ++i) {                 // the half's 2 HS 8-byte fragment reads leave the compiler as HS ds_read2_b64
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (decltype(bias)::value) bias_half(z0a, z1a);
      __builtin_amdgcn_sched_barrier(0);
      __syncthreads();
      read_half(cur ^ 1, 0, z0a, z1a, x0a, x1a);
      __builtin_amdgcn_sched_barrier(0);
      mfma_half(z0b, z1b, x0b, x1b);
      if constexpr (decltype(bias)::value) bias_half(z0b, z1b);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (nsteps > 0) {   // (block-uniform)
    const bool tail = ((r_end - r_begin) % TBR) != 0;
    if (tail) {
      if (do_bias) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{});
    } else {
      if (do_bias) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{});
    }
  }
  __syncthreads();
  if (do_bias) {   // (block-uniform) waves wn == 0 hold all ZW columns; fold the two row parities
    float* red = Zs(0);
    if (wn == 0) {
      const bool use = kh == 0 || g.group == 1;
      red[kh * ZW + wm * 64 + 2 * li] = use ? bsum0 : 0.f;          // (interleaved blocks: column 2 li + a)
      red[kh * ZW + wm * 64 + 2 * li + 1] = use ? bsum1 : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < ZW && n0 + threadIdx.x < g.N) g.db_partial[(int64_t)split * g.N + n0 + threadIdx.x] = red[threadIdx.x] + red[threadIdx.x + ZW];
  }
  float* out = g.partial + (int64_t)split * g.N * g.lddw;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int col = k0 + wn * 64 + 2 * li + b;                    // interleaved blocks (see read_half)
      if (col >= g.K) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = n0 + wm * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * kh) + a;
        if (row < g.N) out[(int64_t)row * g.lddw + col] = acc[a][b][r];
      }
    }
}

template <int ZW, int XW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_tn_kernel(sr_gemm_tn_args g, int rows_per_split) {
  extern __shared__ __attribute__((aligned(16))) float tn_smem[];
  gemm_tn_body<ZW, XW>(g, rows_per_split, blockIdx.x, gridDim.x, tn_smem);
}

// Several independent weight gradients of the 128 x 128 tile shape in ONE launch (sr_mlp_gemm_tn_group): the small reverse sweeps of the
// ray branch and the implicit-gradient pass (1-5k rows) produce one weight gradient per layer, each a launch of < 100 workgroups that
// ends before the next has started -- 26 launches per iteration at 27 TFLOP/s.  Here the workgroups of all of them share a grid
// (`first[i]` = first workgroup of problem i, a multiple of 8 so that the XCD-aware tile order of the body still sees block b on XCD
// b % 8); every workgroup runs the same tile body on the same slab as the single launch would: bit-identical partials.
struct TnGroup {
  int n;
  int first[SR_TN_GROUP_MAX + 1];
  int blocks[SR_TN_GROUP_MAX];            // workgroups problem i really has (first[i + 1] - first[i] rounded it up to a multiple of 8)
  int rows_per_split[SR_TN_GROUP_MAX];
  sr_gemm_tn_args p[SR_TN_GROUP_MAX];
};
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gemm_tn_group_kernel(TnGroup G) {
  extern __shared__ __attribute__((aligned(16))) float tn_smem[];
  int i = 0;
  while (i + 1 < G.n && (int)blockIdx.x >= G.first[i + 1]) ++i;      // (uniform)
  const int bid = blockIdx.x - G.first[i];
  if (bid >= G.blocks[i]) return;
  gemm_tn_body<128, 128>(G.p[i], G.rows_per_split[i], bid, G.blocks[i], tn_smem);
}

// 256 x 64 tiles when the A operand is at most 64 columns wide and dW has at least 256 rows
static inline bool tn_narrow(int N, int64_t K) { return K <= 64 && N >= 256; }

// dW = (accumulate ? dW : 0) + sum_s partial[s]; padding columns [K, lddw) are written as 0.  The same launch folds the
// bias-gradient partials (indices past N * lddw): db = (accumulate ? db : 0) + sum_s db_partial[s].
// Slabs are added in slab order (deterministic).  Round 4: four elements per thread as 16-byte loads, and the loads of eight slabs are
// issued before the first of them is added -- the plain loop (`s += partial[p * total + i]`, trip count unknown to the compiler) waited
// for every slab's load in turn: 22 us for 16-32 slabs of a 512 x 512 gradient, i.e. memory latency x slabs, ninety times per iteration
// on the weight-gradient stream.  Same order of additions per element, same bits.
__device__ __forceinline__ void slab_reduce_body(const float* __restrict__ partial, float* __restrict__ dW, int N,
                                                 int K, int64_t lddw, int splits, int accumulate,
                                                 const float* __restrict__ db_partial, float* __restrict__ db, int bid, int nblocks) {
  const int64_t total = (int64_t)N * lddw;
  const int64_t quads = total >> 2;                      // lddw % 4 == 0 (checked by the caller)
  const int64_t all = quads + (db ? N : 0);
  const f32x4* __restrict__ p4 = reinterpret_cast<const f32x4*>(partial);
  f32x4* __restrict__ d4 = reinterpret_cast<f32x4*>(dW);
  for (int64_t i = (int64_t)bid * blockDim.x + threadIdx.x; i < all; i += (int64_t)nblocks * blockDim.x) {
    if (i < quads) {
      const int col = (int)((i << 2) % lddw);
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      if (col < K) {
        const f32x4* src = p4 + i;
        int p = 0;
        for (; p + 8 <= splits; p += 8) {
          f32x4 v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)(p + j) * quads];
#pragma unroll
          for (int j = 0; j < 8; ++j) s += v[j];
        }
        for (; p < splits; ++p) s += src[(int64_t)p * quads];
        if (col + 1 >= K) s.y = 0.f;                     // columns past K hold whatever the workspace held: the tile kernel never writes them
        if (col + 2 >= K) s.z = 0.f;
        if (col + 3 >= K) s.w = 0.f;
      }
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (accumulate) o = d4[i];
      d4[i] = o + s;
    } else {
      const int n = (int)(i - quads);
      float s = 0.f;
      for (int p = 0; p < splits; ++p) s += db_partial[(int64_t)p * N + n];
      db[n] = (accumulate ? db[n] : 0.f) + s;
    }
  }
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, int N,
                                                           int K, int64_t lddw, int splits, int accumulate,
                                                           const float* __restrict__ db_partial, float* __restrict__ db) {
  slab_reduce_body(partial, dW, N, K, lddw, splits, accumulate, db_partial, db, blockIdx.x, gridDim.x);
}

// the slab reductions of a grouped weight-gradient launch, one launch too (same element order as the single reductions)
__global__ __launch_bounds__(256) void slab_reduce_group_kernel(TnGroup G) {
  int i = 0;
  while (i + 1 < G.n && (int)blockIdx.x >= G.first[i + 1]) ++i;
  const int bid = blockIdx.x - G.first[i];
  if (bid >= G.blocks[i]) return;
  const sr_gemm_tn_args& g = G.p[i];
  slab_reduce_body(g.partial, g.dW, g.N, g.K, g.lddw, g.R > 0 ? g.splits : 0, g.accumulate, g.db_partial, g.db, bid, G.blocks[i]);
}

// out[n] = sum over primal rows (r % group == 0) of Z[r][n]; one workgroup per 64 columns x row-slice,
// finished by atomics into a zeroed `out` only when more than one slice exists.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ Z, int64_t ldz, int R, int N, int group,
                                                      float* __restrict__ out, int rows_per_block) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;
  const int r_begin = blockIdx.y * rows_per_block;
  const int r_end = min(R, r_begin + rows_per_block);
  float s = 0.f;
  if (c < N)
    for (int r = r_begin + sub * group; r < r_end; r += 4 * group) s += Z[(int64_t)r * ldz + c];
  red[sub][threadIdx.x & 63] = s;
  __syncthreads();
  if (sub == 0 && c < N) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace

extern "C" {
int sr_mlp_gemm_nt(const sr_gemm_args* a, void* stream) {
  if (!a || a->M < 0 || a->N <= 0 || a->K <= 0) return SR_EINVAL;
  if (a->M == 0) return SR_OK;
  if (!a->A || !a->B || !a->C) return SR_EINVAL;
  if (a->group != 1 && a->group != 2 && a->group != 4) return SR_EINVAL;
  if ((a->lda & 3) || (a->ldb & 3) || (a->ldc & 3) || ((uintptr_t)a->A & 15) || ((uintptr_t)a->B & 15) || ((uintptr_t)a->C & 15)) return SR_EINVAL;
  if (a->mode != SR_EPI_FWD && a->mode != SR_EPI_BWD) return SR_EINVAL;
  if ((a->naux_fwd > 0 || a->mode == SR_EPI_BWD) && a->act != SR_ACT_NONE && !a->aux && a->mode == SR_EPI_BWD) return SR_EINVAL;
  if (a->mode == SR_EPI_FWD && a->naux_fwd > 0 && !a->aux) return SR_EINVAL;
  if (a->M == 0) return SR_OK;
  if (a->M % a->group) return SR_EINVAL;
  const int ncols = a->N + (a->mode == SR_EPI_FWD ? a->naux_fwd : 0);
  sr_gemm_args g = *a;
  if (g.mode != SR_EPI_FWD) g.naux_fwd = 0;
  // Tile choice by a cost model, see below.
  auto cost = [&](int bm, int bn, double eff) {
    const int64_t wgs = sr_cdiv(g.M, bm) * sr_cdiv(ncols, bn);
    return (double)sr_cdiv(wgs, 256) * bm * bn / eff;
  };
#define SR_NT_LAUNCH(WM, WN, TM, TN)                                                                                        \
  do {                                                                                                                      \
    using C_ = Cfg<WM, WN, TM, TN>;                                                                                         \
    const int nwg = (int)(sr_cdiv(g.M, C_::BM) * sr_cdiv(ncols, C_::BN));                                                   \
    if (g.K % BK)                                                                                                           \
      hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, true>), dim3(nwg), dim3(C_::kThreads), C_::kLdsFloats * sizeof(float), \
                         (hipStream_t)stream, g);                                                                           \
    else                                                                                                                    \
      hipLaunchKernelGGL((gemm_nt_kernel<WM, WN, TM, TN, false>), dim3(nwg), dim3(C_::kThreads), C_::kLdsFloats * sizeof(float), \
                         (hipStream_t)stream, g);                                                                           \
  } while (0)
  if (ncols <= 32) {
    // narrow outputs (the 3-wide deformer / render heads, the sdf-only last layer): 256x32 tiles for the template-sized batches,
    // 64x32 / 32x32 for the refiner's few thousand rows (6k rows are only 24 tiles of 256 rows on 256 CUs)
    const double c[3] = {cost(32, 32, 0.6), cost(64, 32, 0.8), cost(256, 32, 1.0)};
    if (c[0] < c[1] && c[0] < c[2]) SR_NT_LAUNCH(1, 1, 1, 1);
    else if (c[1] < c[2]) SR_NT_LAUNCH(2, 1, 1, 1);
    else SR_NT_LAUNCH(4, 1, 2, 1);
  } else {
    // The workgroups resident on a CU share its four MFMA pipes, so a CU's time is (tiles it receives) x (tile work):
    // cost = ceil(workgroups / 256) * bm * bn / eff, eff = measured large-M rate of the configuration relative to 128x128.
    // (M = 6144, N = 512: 64x64 gives 3 tiles/CU = 12.3k, 64x128 2 tiles/CU = 16.4k, 128x128 1 tile on 192 CUs = 16.4k;
    // measured 31.5 / 39.9 / 45 us.)
    // (256x128 and 128x256 tiles with 8 waves were measured in rounds 3-4 and lose: one workgroup per CU.)
    const double c[3] = {cost(64, 64, 0.92), cost(64, 128, 0.94), cost(128, 128, 1.0)};
    int pick = 1;
    for (int i = 1; i < 3; ++i) if (c[i] < c[pick - 1]) pick = i + 1;
    switch (pick) {
      case 1: SR_NT_LAUNCH(2, 2, 1, 1); break;
      case 2: SR_NT_LAUNCH(2, 2, 1, 2); break;
      default: SR_NT_LAUNCH(2, 2, 2, 2); break;
    }
  }
#undef SR_NT_LAUNCH
  return sr_launch_status();
}

// The same layer pair as ONE ordinary launch: grid sized for the capacity row count, workgroups past the tiles of the LIVE row count
// (device memory) return at once.  No barrier, no residency requirement; consecutive layers are ordered by the stream.
__global__ __launch_bounds__(ChainCfg::kThreads) __attribute__((amdgpu_waves_per_eu(2))) void mlp_layer_pair_kernel(sr_gemm_args g0, sr_gemm_args g1, int nprob,
                                                                                                                     const int32_t* __restrict__ m_dev, int m_mul) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int M = *m_dev * m_mul;
  if (M <= 0) return;
  const int rows = (M + ChainCfg::BM - 1) / ChainCfg::BM;
  const int t0 = rows * ((g0.N + (g0.mode == SR_EPI_FWD ? g0.naux_fwd : 0) + ChainCfg::BN - 1) / ChainCfg::BN);
  const int t1 = nprob > 1 ? rows * ((g1.N + (g1.mode == SR_EPI_FWD ? g1.naux_fwd : 0) + ChainCfg::BN - 1) / ChainCfg::BN) : 0;
  const int t = blockIdx.x;
  if (t >= t0 + t1) return;
  // (block-uniform: the 512-wide layers take the tile code without the K-tail selects)
  if (t < t0) { g0.M = M; if (g0.K % BK) gemm_nt_tile<2, 2, 1, 1, true>(g0, t, smem); else gemm_nt_tile<2, 2, 1, 1, false>(g0, t, smem); }
  else { g1.M = M; if (g1.K % BK) gemm_nt_tile<2, 2, 1, 1, true>(g1, t - t0, smem); else gemm_nt_tile<2, 2, 1, 1, false>(g1, t - t0, smem); }
}

int sr_mlp_chain(const sr_chain_args* a, void* stream) {
  if (!a || a->nlayers < 1 || a->nlayers > SR_CHAIN_MAX_LAYERS || !a->m_dev || a->m_mul < 1 || a->m_cap <= 0) return SR_EINVAL;
  for (int l = 0; l < a->nlayers; ++l) {
    if (a->nprob[l] < 1 || a->nprob[l] > 2) return SR_EINVAL;
    for (int p = 0; p < a->nprob[l]; ++p) {
      const sr_gemm_args& g = a->g[l][p];
      if (!g.A || !g.B || !g.C || g.N <= 0 || g.K <= 0 || g.group != a->m_mul) return SR_EINVAL;
      if ((g.lda & 3) || (g.ldb & 3) || (g.ldc & 3) || ((uintptr_t)g.A & 15) || ((uintptr_t)g.B & 15) || ((uintptr_t)g.C & 15)) return SR_EINVAL;
      if (g.mode != SR_EPI_FWD && g.mode != SR_EPI_BWD) return SR_EINVAL;
      if (g.mode == SR_EPI_BWD && g.naux_fwd != 0) return SR_EINVAL;
      if ((g.mode == SR_EPI_BWD && g.act != SR_ACT_NONE && !g.aux) || (g.mode == SR_EPI_FWD && g.naux_fwd > 0 && !g.aux)) return SR_EINVAL;
    }
  }
  // one launch per layer (pair): grids sized for the capacity row count, workgroups past the tiles of the live count return at once;
  // consecutive layers are ordered by the stream (~13 us per dependent launch at 2k rows, the layer's own duration)
  const int64_t rows = sr_cdiv((int64_t)a->m_cap * a->m_mul, ChainCfg::BM);
  for (int l = 0; l < a->nlayers; ++l) {
    int64_t tiles = 0;
    for (int p = 0; p < a->nprob[l]; ++p) {
      const sr_gemm_args& g = a->g[l][p];
      tiles += rows * sr_cdiv(g.N + (g.mode == SR_EPI_FWD ? g.naux_fwd : 0), ChainCfg::BN);
    }
    hipLaunchKernelGGL(mlp_layer_pair_kernel, dim3((unsigned)tiles), dim3(ChainCfg::kThreads), ChainCfg::kLdsFloats * sizeof(float), (hipStream_t)stream,
                       a->g[l][0], a->g[l][a->nprob[l] > 1 ? 1 : 0], a->nprob[l], a->m_dev, a->m_mul);
  }
  return sr_launch_status();
}

int64_t sr_mlp_gemm_tn_workspace_floats(int32_t R, int32_t N, int64_t lddw, int32_t* splits_out) {
  const int tiles = tn_narrow(N, lddw) ? (int)sr_cdiv(N, 256) : (int)(sr_cdiv(N, 128) * sr_cdiv(lddw, 128));
  const int target = 512;
  int splits = (int)sr_cdiv(target, tiles);           // two workgroups per CU are resident (LDS): one full wave of the chip; every
                                                      // further slab costs a 64 KB partial tile written and read again
  const int max_splits = (int)sr_cdiv(R, 384);       // at least 384 rows (12 steps) per split
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  if (splits_out) *splits_out = splits;
  return (int64_t)splits * N * lddw;
}

int sr_mlp_gemm_tn(const sr_gemm_tn_args* a, void* stream) {
  if (!a || !a->dW || !a->partial || a->R < 0 || a->N <= 0 || a->K <= 0 || a->splits < 1) return SR_EINVAL;
  if (a->R > 0 && (!a->Z || !a->A)) return SR_EINVAL;
  if ((a->ldz & 3) || (a->lda & 3) || ((uintptr_t)a->Z & 15) || ((uintptr_t)a->A & 15) || a->lddw < a->K) return SR_EINVAL;
  if ((a->lddw & 3) || ((uintptr_t)a->dW & 15) || ((uintptr_t)a->partial & 15)) return SR_EINVAL;      // the slab reduction moves float4
  if ((a->db != nullptr) != (a->db_partial != nullptr) || (a->db && a->group < 1)) return SR_EINVAL;
  const bool narrow = tn_narrow(a->N, a->lddw);        // (the same rule as the workspace query: it sized `splits`)
  const int tiles = narrow ? (int)sr_cdiv(a->N, 256) : (int)(sr_cdiv(a->N, 128) * sr_cdiv(a->K, 128));
  int rows_per_split = (int)sr_cdiv(a->R, a->splits);
  rows_per_split = (int)(sr_cdiv(rows_per_split, TBR) * TBR);
  if (a->R > 0) {
    using Narrow = TnCfg<256, 64>;
    using Square = TnCfg<128, 128>;
    if (narrow)
      hipLaunchKernelGGL((gemm_tn_kernel<256, 64>), dim3(tiles * a->splits), dim3(256), Narrow::kLdsFloats * sizeof(float), (hipStream_t)stream, *a,
                         rows_per_split);
    else
      hipLaunchKernelGGL((gemm_tn_kernel<128, 128>), dim3(tiles * a->splits), dim3(256), Square::kLdsFloats * sizeof(float), (hipStream_t)stream, *a,
                         rows_per_split);
  }
  const int64_t total = (int64_t)a->N * a->lddw / 4 + (a->db ? a->N : 0);
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(sr_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, a->partial, a->dW,
                     a->N, a->K, a->lddw, a->R > 0 ? a->splits : 0, a->accumulate, a->db_partial, a->db);
  return sr_launch_status();
}

static int tn_check(const sr_gemm_tn_args* a) {
  if (!a || !a->dW || !a->partial || a->R < 0 || a->N <= 0 || a->K <= 0 || a->splits < 1) return SR_EINVAL;
  if (a->R > 0 && (!a->Z || !a->A)) return SR_EINVAL;
  if ((a->ldz & 3) || (a->lda & 3) || ((uintptr_t)a->Z & 15) || ((uintptr_t)a->A & 15) || a->lddw < a->K) return SR_EINVAL;
  if ((a->lddw & 3) || ((uintptr_t)a->dW & 15) || ((uintptr_t)a->partial & 15)) return SR_EINVAL;
  if ((a->db != nullptr) != (a->db_partial != nullptr) || (a->db && a->group < 1)) return SR_EINVAL;
  return SR_OK;
}

int sr_mlp_gemm_tn_group(const sr_gemm_tn_group_args* a, void* stream) {
  if (!a || a->n < 1 || a->n > SR_TN_GROUP_MAX) return SR_EINVAL;
  TnGroup G, Rd;
  G.n = Rd.n = a->n;
  int nb = 0, nr = 0;
  bool any_rows = false;
  for (int i = 0; i < a->n; ++i) {
    const sr_gemm_tn_args& p = a->p[i];
    const int rc = tn_check(&p);
    if (rc != SR_OK) return rc;
    if (tn_narrow(p.N, p.lddw)) return SR_EINVAL;            // (the 256 x 64 shape has its own LDS footprint: single launches)
    const int tiles = (int)(sr_cdiv(p.N, 128) * sr_cdiv(p.K, 128));
    int rows_per_split = (int)sr_cdiv(p.R, p.splits);
    rows_per_split = (int)(sr_cdiv(rows_per_split, TBR) * TBR);
    G.p[i] = Rd.p[i] = p;
    G.rows_per_split[i] = Rd.rows_per_split[i] = rows_per_split;
    G.first[i] = nb; G.blocks[i] = p.R > 0 ? tiles * p.splits : 0;
    nb += (int)(sr_cdiv(G.blocks[i], 8) * 8);
    const int64_t total = (int64_t)p.N * p.lddw / 4 + (p.db ? p.N : 0);
    Rd.first[i] = nr; Rd.blocks[i] = sr_stream_grid(total, 256);
    nr += Rd.blocks[i];
    any_rows = any_rows || p.R > 0;
  }
  G.first[a->n] = nb; Rd.first[a->n] = nr;
  using Square = TnCfg<128, 128>;
  if (any_rows && nb > 0)
    hipLaunchKernelGGL(gemm_tn_group_kernel, dim3(nb), dim3(256), Square::kLdsFloats * sizeof(float), (hipStream_t)stream, G);
  hipLaunchKernelGGL(slab_reduce_group_kernel, dim3(nr), dim3(256), 0, (hipStream_t)stream, Rd);
  return sr_launch_status();
}

int sr_colsum_rows(const float* Z, int64_t ldz, int32_t R, int32_t N, int32_t group, float* out, void* stream) {
  if (!out || R < 0 || N <= 0 || group < 1) return SR_EINVAL;
  if (R == 0) return SR_OK;
  if (!Z) return SR_EINVAL;
  int slices = (int)sr_cdiv(R, 4096);
  if (slices > 256) slices = 256;
  int rows_per_block = (int)sr_cdiv(R, slices);
  rows_per_block = (int)(sr_cdiv(rows_per_block, 4 * group) * 4 * group);
  slices = (int)sr_cdiv(R, rows_per_block);
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)sr_cdiv(N, 64), slices), dim3(256), 0, (hipStream_t)stream, Z, ldz, R, N, group, out, rows_per_block);
  return sr_launch_status();
}
}

// ------------------------------------------------------------------------------------------------
// Positional encoding + feature concatenation -> first-layer input rows (a1).
// out[row, :] = [x(3) | w_k sin(2^k x), w_k cos(2^k x) (k<L) | extra(E) | 0 pad] following
// model/Embedder.py:9-41 (band order, 3-wide blocks) and utils/utils.py:40-46 (weights come in
// pairs).  With group == 4 the 3 rows after each primal row receive d/dx_t of that embedding
// (t = 0,1,2) -- the seed tangents of the forward-mode Jacobian -- and zeros under `extra`.
namespace {
__global__ __launch_bounds__(256) void pe_embed_kernel(const float* __restrict__ x, int64_t P, int L, const float* __restrict__ w,
                                                        const float* __restrict__ extra, int64_t ldextra, int E,
                                                        const int64_t* __restrict__ extra_index, int group,
                                                        float* __restrict__ out, int64_t ldo) {
  const int width = 3 + 6 * L + E;
  const int64_t total = P * ldo;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / ldo;
    const int c = (int)(i % ldo);
    float v = 0.f, t3[3] = {0.f, 0.f, 0.f};
    if (c < 3) {
      v = x[p * 3 + c];
      t3[c] = 1.f;
    } else if (c < 3 + 6 * L) {
      const int k = (c - 3) / 6, r = (c - 3) % 6, comp = r % 3;
      const float f = (float)(1 << k);
      const float a = x[p * 3 + comp] * f;
      const float wk = w[2 * k + (r >= 3)];
      if (r < 3) { v = wk * sinf(a); t3[comp] = wk * f * cosf(a); }
      else { v = wk * cosf(a); t3[comp] = -wk * f * sinf(a); }
    } else if (c < width) {
      const int64_t src = extra_index ? extra_index[p] : p;
      v = extra[src * ldextra + (c - 3 - 6 * L)];
    }
    out[(p * group) * ldo + c] = v;
    if (group == 4) {
      out[(p * 4 + 1) * ldo + c] = t3[0];
      out[(p * 4 + 2) * ldo + c] = t3[1];
      out[(p * 4 + 3) * ldo + c] = t3[2];
    }
  }
}
}  // namespace

extern "C" int sr_pe_embed(const float* x, int64_t P, int32_t L, const float* band_weights, const float* extra, int64_t ldextra,
                           int32_t E, const int64_t* extra_index, int32_t group, float* out, int64_t ldo, void* stream) {
  if (P < 0 || L < 0 || L > 16 || E < 0 || (group != 1 && group != 4) || ldo < 3 + 6 * L + E) return SR_EINVAL;
  if (P == 0) return SR_OK;
  if (!x || !out || (L > 0 && !band_weights) || (E > 0 && !extra)) return SR_EINVAL;
  hipLaunchKernelGGL(pe_embed_kernel, dim3(sr_stream_grid(P * ldo, 256)), dim3(256), 0, (hipStream_t)stream, x, P, L, band_weights,
                     extra, ldextra, E, extra_index, group, out, ldo);
  return sr_launch_status();
}

// Reverse of sr_pe_embed with respect to x: xbar[p,c] = A0bar[primal, c] + sum_k 2^k (w cos . g_sin - w sin . g_cos)
//   (+ for group 4: tangent row t == c only, since the encoding is separable per coordinate:
//      A0bar[t, sin block] * (-w 4^k sin) + A0bar[t, cos block] * (-w 4^k cos) ).
namespace {
__global__ __launch_bounds__(256) void pe_embed_bwd_kernel(const float* __restrict__ x, int64_t P, int L, const float* __restrict__ w,
                                                            int group, const float* __restrict__ gA0, int64_t ldg, float* __restrict__ xbar) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P * 3; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / 3;
    const int c = (int)(i % 3);
    const float xv = x[i];
    const float* g0 = gA0 + (p * group) * ldg;
    const float* gt = group == 4 ? gA0 + (p * 4 + 1 + c) * ldg : nullptr;
    float acc = g0[c];
    for (int k = 0; k < L; ++k) {
      const float f = (float)(1 << k);
      float sn, cs;
      sincosf(xv * f, &sn, &cs);
      const float ws = w[2 * k], wc = w[2 * k + 1];
      const int is = 3 + 6 * k + c, ic = is + 3;
      acc += f * (ws * cs * g0[is] - wc * sn * g0[ic]);
      if (gt) acc -= f * f * (ws * sn * gt[is] + wc * cs * gt[ic]);
    }
    xbar[i] = acc;
  }
}
}  // namespace

extern "C" int sr_pe_embed_bwd(const float* x, int64_t P, int32_t L, const float* band_weights, int32_t group, const float* gA0,
                               int64_t ldg, float* xbar, void* stream) {
  if (P < 0 || L < 0 || L > 16 || (group != 1 && group != 4) || ldg < 3 + 6 * L) return SR_EINVAL;
  if (P == 0) return SR_OK;
  if (!x || !gA0 || !xbar || (L > 0 && !band_weights)) return SR_EINVAL;
  hipLaunchKernelGGL(pe_embed_bwd_kernel, dim3(sr_stream_grid(P * 3, 256)), dim3(256), 0, (hipStream_t)stream, x, P, L, band_weights,
                     group, gA0, ldg, xbar);
  return sr_launch_status();
}
