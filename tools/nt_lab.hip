// Lab bench of the NT layer-GEMM tile code (csrc/mlp_gemm.hip): where does a 128x128 tile's time go, and what do launch-shape
// experiments (staggered co-resident workgroups, a resident grid, one workgroup per CU) buy?  Includes the product source, so the tile
// body measured here IS the product's.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gpurun_out/nt_lab tools/nt_lab.hip && gpurun_out/nt_lab <M> <N> <K> <act> <variant> ...
// Every workgroup leaves cycle stamps (start, prologue end, loop end, epilogue end) and its hardware placement (HW_ID, XCC_ID), dumped
// as CSV for tools/nt_lab_report.py.
#include "../selfreconcode_amd/csrc/mlp_gemm.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
#include <map>
#include <array>
#include <algorithm>

namespace {
struct LabArgs {
  unsigned long long* stamps;   // [tiles][8]: t_start, t_pro, t_loop, t_end, hw_id, xcc_id, block, iteration
  int stagger_mode;             // 0 none, 1 parity of HW_ID.wave_id, 2 parity of HW_ID.tg_id
  int stagger_cycles;
  int first_wave;               // workgroups of the first dispatch wave (only they are delayed)
  int persistent;               // != 0: grid-stride over the tiles
};

// PRIO: wave priority by phase -- 0 none; 1 prologue + epilogue at priority 3, tile loop at 0; 2 prologue only; 3 epilogue only
template <int PRIO>
struct StampProbe {
  unsigned long long* slot;
  __device__ __forceinline__ void operator()(int i) const {
    if (PRIO == 1 || PRIO == 2) { if (i == 0) __builtin_amdgcn_s_setprio(0); }
    if (PRIO == 1 || PRIO == 3) { if (i == 1) __builtin_amdgcn_s_setprio(3); }
    if (PRIO == 3) { if (i == 2) __builtin_amdgcn_s_setprio(0); }
    if (slot && threadIdx.x == 0) slot[1 + i] = (unsigned long long)clock64();
  }
};

#define SR_GETREG(id) __builtin_amdgcn_s_getreg((31 << 11) | (id))

template <bool KTAIL, int PRIO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void lab_kernel(sr_gemm_args g, LabArgs lab) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  using C_ = Cfg<2, 2, 2, 2>;
  if (PRIO == 1 || PRIO == 2) __builtin_amdgcn_s_setprio(3);
  const unsigned hw = SR_GETREG(4), xcc = SR_GETREG(20);
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  if (lab.stagger_mode && (int)blockIdx.x < lab.first_wave) {
    const unsigned par = lab.stagger_mode == 1 ? (hw & 1u) : ((hw >> 16) & 1u);
    if (par)
      while (clock64() - t0 < lab.stagger_cycles) __builtin_amdgcn_s_sleep(32);
  }
  const int tiles = ((g.M + C_::BM - 1) / C_::BM) * ((g.N + g.naux_fwd + C_::BN - 1) / C_::BN);
  int it = 0;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x, ++it) {
    unsigned long long* slot = lab.stamps ? lab.stamps + (size_t)t * 8 : nullptr;
    if (slot && threadIdx.x == 0) {
      slot[0] = (unsigned long long)(it == 0 ? t0 : clock64());
      slot[4] = hw; slot[5] = xcc; slot[6] = blockIdx.x;
    }
    if (PRIO == 1 || PRIO == 2) { if (it > 0) __builtin_amdgcn_s_setprio(3); }
    gemm_nt_tile<2, 2, 2, 2, KTAIL>(g, t, smem, StampProbe<PRIO>{slot});
    if (slot && threadIdx.x == 0) slot[7] = (unsigned long long)(wall_clock64() - w0);     // 100 MHz ticks of this workgroup so far
    if (!lab.persistent) break;
    __syncthreads();
  }
}

// ---- what does each piece of a tile step cost?  A copy of gemm_nt_tile's loop (128x128, K % 32 == 0, plain store epilogue) whose pieces
// can be compiled out (results are then wrong; only the cycle count per tile is read):
//   V & 1: no s_barrier   V & 2: no global loads   V & 4: no LDS stores   V & 8: no sched_group_barrier interleave   V & 16: no fragment reads
template <int V>
__device__ __forceinline__ void lab_tile(const sr_gemm_args& g, int wg, float* __restrict__ smem, unsigned long long* slot) {
  constexpr int WM = 2, WN = 2, TM = 2, TN = 2;
  using C_ = Cfg<WM, WN, TM, TN>;
  auto As = [&](int buf) -> float* { return smem + buf * (C_::BM * LDSP); };
  auto Bs = [&](int buf) -> float* { return smem + 2 * C_::BM * LDSP + buf * (C_::BN * LDSP); };
  const int tiles_n = g.N / C_::BN;
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
  const TileLoader<C_::BM, C_::kALoads, C_::kThreads> la(g.A, g.lda, g.M, g.K, m0);
  const TileLoader<C_::BN, C_::kBLoads, C_::kThreads> lb(g.B, g.ldb, g.N, g.K, n0);
  const int nk = g.K / BK;
  const int a_off = (wm * TM * 32 + li) * LDSP + kh * 4, b_off = (wn * TN * 32 + li) * LDSP + kh * 4;
  f32x4 fa0[2][TM], fb0[2][TN], fa1[2][TM], fb1[2][TN];
  auto read_frags = [&](const float* abuf, const float* bbuf, int kk0, f32x4 (&fa)[2][TM], f32x4 (&fb)[2][TN]) {
    if (V & 16) return;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int a = 0; a < TM; ++a) fa[h][a] = *reinterpret_cast<const f32x4*>(abuf + a_off + a * 32 * LDSP + (kk0 + h) * 8);
#pragma unroll
      for (int b = 0; b < TN; ++b) fb[h][b] = *reinterpret_cast<const f32x4*>(bbuf + b_off + b * 32 * LDSP + (kk0 + h) * 8);
    }
  };
  auto mfma_kk = [&](const f32x4 (&fa)[TM], const f32x4 (&fb)[TN]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][e], fb[b][e], acc[a][b], 0, 0, 0);
  };
  constexpr int kMem = C_::kALoads + C_::kBLoads;
  f32x4 ra0[C_::kALoads], rb0[C_::kBLoads], ra1[C_::kALoads], rb1[C_::kBLoads];
  la.load(0, ra0); lb.load(0, rb0);
  la.load(BK, ra1); lb.load(BK, rb1);
  la.template store<false>(As(0), 0, ra0); lb.template store<false>(Bs(0), 0, rb0);
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int a = 0; a < TM; ++a) { fa0[h][a] = *reinterpret_cast<const f32x4*>(As(0) + a_off + a * 32 * LDSP + h * 8); fa1[h][a] = fa0[h][a]; }
#pragma unroll
    for (int b = 0; b < TN; ++b) { fb0[h][b] = *reinterpret_cast<const f32x4*>(Bs(0) + b_off + b * 32 * LDSP + h * 8); fb1[h][b] = fb0[h][b]; }
  }
  if (slot && threadIdx.x == 0) slot[1] = (unsigned long long)clock64();
  auto step = [&](int t, f32x4 (&ain)[C_::kALoads], f32x4 (&bin)[C_::kBLoads], const f32x4 (&aout)[C_::kALoads],
                  const f32x4 (&bout)[C_::kBLoads]) {
    const int cur = t & 1;
    if (!(V & 2)) { la.load((t + 2) * BK, ain); lb.load((t + 2) * BK, bin); }
    if (!(V & 4)) { la.template store<false>(As(cur ^ 1), (t + 1) * BK, aout); lb.template store<false>(Bs(cur ^ 1), (t + 1) * BK, bout); }
    read_frags(As(cur), Bs(cur), 2, fa1, fb1);
    mfma_kk(fa0[0], fb0[0]);
    mfma_kk(fa0[1], fb0[1]);
    if (!(V & 8)) {
      if (!(V & 2)) {
#pragma unroll
        for (int i = 0; i < kMem; ++i) { __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
      }
      if (!(V & 4)) {
#pragma unroll
        for (int i = 0; i < kMem; ++i) { __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
      }
      if (!(V & 16)) {
#pragma unroll
        for (int i = 0; i < 2 * (TM + TN); ++i) { __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (!(V & 1)) __syncthreads();
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
  __syncthreads();
  if (slot && threadIdx.x == 0) slot[2] = (unsigned long long)clock64();
  // keep everything alive: fold the stage registers into the result, store the MFMA layout directly
  float extra = 0.f;
#pragma unroll
  for (int j = 0; j < C_::kALoads; ++j) extra += ra0[j].x + ra1[j].x + rb0[j].x + rb1[j].x;
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh, col = n0 + (wn * TN + b) * 32 + li;
        g.C[(int64_t)row * g.ldc + col] = acc[a][b][r] + (extra == 12345.f ? 1.f : 0.f);
      }
  if (slot && threadIdx.x == 0) slot[3] = (unsigned long long)clock64();
}

template <int V>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void lab_variant_kernel(sr_gemm_args g, LabArgs lab) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned hw = SR_GETREG(4), xcc = SR_GETREG(20);
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  const int t = blockIdx.x;
  unsigned long long* slot = lab.stamps ? lab.stamps + (size_t)t * 8 : nullptr;
  if (slot && threadIdx.x == 0) { slot[0] = (unsigned long long)t0; slot[4] = hw; slot[5] = xcc; slot[6] = blockIdx.x; }
  lab_tile<V>(g, t, smem, slot);
  if (slot && threadIdx.x == 0) slot[7] = (unsigned long long)(wall_clock64() - w0);
}

// ---- v2 tile loop: operands go global -> LDS directly (global_load_lds_dwordx4, no staging registers, no ds_write), K in steps of 16
// through FOUR 16 KB LDS stages ([128 rows][16 floats] per operand, 16-byte chunks XOR-swizzled with (row >> 2) & 3 on the SOURCE side so
// that the ds_read_b128 fragment reads are conflict-free), stage s+3 in flight while stage s feeds the MFMAs.
#define SR_WAITCNT_VM(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | ((((n) >> 4) & 3) << 14))
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

template <int EPI>
__device__ __forceinline__ void lab_tile_v2(const sr_gemm_args& g, int wg, float* __restrict__ smem, unsigned long long* slot) {
  constexpr int BM = 128, BN = 128, BK2 = 16, STAGE_BYTES = 16384, OPER_BYTES = 8192;
  const int tiles_n = g.N / BN;
  const int tn = wg % tiles_n, tm = wg / tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, kh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  // source pointers of this lane's four loads per stage: A rows (wave*2 + j)*16 + lane/4, B likewise; chunk = (lane % 4) ^ ((row >> 2) & 3)
  const float* src[4];
  unsigned dst[4];
  const unsigned lds0 = (unsigned)(uintptr_t)smem;      // LDS byte address of the dynamic segment
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r0 = (wave * 2 + j) * 16, row = r0 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
    int ga = m0 + row; ga = ga < g.M ? ga : g.M - 1;
    int gb = n0 + row; gb = gb < g.N ? gb : g.N - 1;
    src[j] = g.A + (int64_t)ga * g.lda + 4 * c;
    src[2 + j] = g.B + (int64_t)gb * g.ldb + 4 * c;
    dst[j] = lds0 + r0 * 64;
    dst[2 + j] = lds0 + OPER_BYTES + r0 * 64;
  }
  const int nk = g.K / BK2;
  auto issue = [&](int stage, int buf) {                // (stage past the end: re-read the last one, never used)
    const int k0 = (stage < nk ? stage : nk - 1) * BK2;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(src[j] + k0, __builtin_amdgcn_readfirstlane(dst[j] + buf * STAGE_BYTES));
  };
  // fragment addresses: row = w*64 + a*32 + li, chunk c = 2 kk + kh -> physical chunk c ^ ((li >> 2) & 3)
  const int sw = (li >> 2) & 3;
  const char* lbase = reinterpret_cast<const char*>(smem);
  const int offA[2] = {(wm * 64 + li) * 64 + ((kh ^ sw) << 4), (wm * 64 + li) * 64 + (((2 + kh) ^ sw) << 4)};
  const int offB[2] = {OPER_BYTES + (wn * 64 + li) * 64 + ((kh ^ sw) << 4), OPER_BYTES + (wn * 64 + li) * 64 + (((2 + kh) ^ sw) << 4)};
  f32x4 fa[2][2][2], fb[2][2][2];                       // [set][kk][block]
  auto read_one = [&](int set, int buf, int i) {        // i = 0..7: (operand, kk, block)
    const int kk = (i >> 1) & 1, blk = i & 1;
    if (i < 4) fa[set][kk][blk] = *reinterpret_cast<const f32x4*>(lbase + buf * STAGE_BYTES + offA[kk] + blk * 2048);
    else fb[set][kk][blk] = *reinterpret_cast<const f32x4*>(lbase + buf * STAGE_BYTES + offB[kk] + blk * 2048);
  };
  auto mfma4 = [&](int set, int q) {                    // q = 0..7: (kk, e) -> the four accumulators
    const int kk = q >> 2, e = q & 3;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][kk][a][e], fb[set][kk][b][e], acc[a][b], 0, 0, 0);
  };
  issue(0, 0); issue(1, 1); issue(2, 2);
  SR_WAITCNT_VM(8);
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int i = 0; i < 8; ++i) read_one(0, 0, i);
  if (slot && threadIdx.x == 0) slot[1] = (unsigned long long)clock64();
  auto step = [&](int s, auto SET, auto BUF) {
    constexpr int set = decltype(SET)::value, buf = decltype(BUF)::value;
    // stage s + 3 -> buffer (buf + 3) & 3 (= the buffer stage s - 1 was read from: every wave passed the barrier of step s - 1 after those reads)
    const int k0 = (s + 3 < nk ? s + 3 : nk - 1) * BK2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mfma4(set, j);
      glds16(src[j] + k0, __builtin_amdgcn_readfirstlane(dst[j] + ((buf + 3) & 3) * STAGE_BYTES));
      __builtin_amdgcn_sched_barrier(0);
    }
    SR_WAITCNT_VM(8);                                   // this wave's part of stage s + 1 has landed (stages s + 2, s + 3 may be in flight)
    __builtin_amdgcn_s_barrier();                       // ... and everybody's
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i & 1) mfma4(set, 4 + (i >> 1));
      read_one(set ^ 1, (buf + 1) & 3, i);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  for (int s = 0; s < nk; s += 4) {
    step(s, I0{}, I0{});
    step(s + 1, I1{}, I1{});
    step(s + 2, I0{}, I2{});
    step(s + 3, I1{}, I3{});
  }
  SR_WAITCNT_VM(0);
  __builtin_amdgcn_s_barrier();
  if (slot && threadIdx.x == 0) slot[2] = (unsigned long long)clock64();
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + (wm * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh, col = n0 + (wn * 2 + b) * 32 + li;
        if (row < g.M) g.C[(int64_t)row * g.ldc + col] = acc[a][b][r];
      }
  if (slot && threadIdx.x == 0) slot[3] = (unsigned long long)clock64();
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void lab_v2_kernel(sr_gemm_args g, LabArgs lab) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const unsigned hw = SR_GETREG(4), xcc = SR_GETREG(20);
  const long long t0 = clock64();
  const long long w0 = wall_clock64();
  const int t = blockIdx.x;
  unsigned long long* slot = lab.stamps ? lab.stamps + (size_t)t * 8 : nullptr;
  if (slot && threadIdx.x == 0) { slot[0] = (unsigned long long)t0; slot[4] = hw; slot[5] = xcc; slot[6] = blockIdx.x; }
  lab_tile_v2<0>(g, t, smem, slot);
  if (slot && threadIdx.x == 0) slot[7] = (unsigned long long)(wall_clock64() - w0);
}
}  // namespace

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: nt_lab M N K act [stagger_mode stagger_frac persistent_grid lds_bytes reps dump.csv|- prio]\n"); return 2; }
  const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), act = atoi(argv[4]);
  const int smode = argc > 5 ? atoi(argv[5]) : 0;
  const double sfrac = argc > 6 ? atof(argv[6]) : 0.5;
  const int pgrid = argc > 7 ? atoi(argv[7]) : 0;
  using C_ = Cfg<2, 2, 2, 2>;
  size_t lds = argc > 8 && atoi(argv[8]) > 0 ? (size_t)atoi(argv[8]) : C_::kLdsFloats * sizeof(float);
  const int reps = argc > 9 ? atoi(argv[9]) : 10;
  const char* dump = argc > 10 && strcmp(argv[10], "-") ? argv[10] : nullptr;
  const int prio = argc > 11 ? atoi(argv[11]) : 0;
  const int data = argc > 12 ? atoi(argv[12]) : 0;      // 0 uniform +-0.1, 1 zeros, 2 every row of A is row 0 (lda = 0: all loads hit cache), 3 gaussian-ish
  const int ldk = (K + 3) & ~3;
  float *A, *B, *C, *bias;
  CK(hipMalloc(&A, (size_t)M * ldk * 4)); CK(hipMalloc(&B, (size_t)N * ldk * 4)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&bias, N * 4));
  {
    std::vector<float> h((size_t)M * ldk);
    unsigned s = 12345u;
    for (auto& v : h) {
      s = s * 1664525u + 1013904223u; v = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.2f;
      if (data == 1) v = 0.f;
      if (data == 3) { float a = 0.f; for (int j = 0; j < 4; ++j) { s = s * 1664525u + 1013904223u; a += (s >> 8) * (1.0f / 16777216.0f) - 0.5f; } v = a * 1.7f; }
    }
    CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, h.data() + 777, (size_t)N * ldk * 4, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, N * 4));
  }
  sr_gemm_args g; memset(&g, 0, sizeof g);
  g.A = A; g.lda = ldk; g.B = B; g.ldb = ldk; g.C = C; g.ldc = N; g.M = M; g.N = N; g.K = K; g.bias = bias; g.group = 1; g.act = act;
  g.mode = SR_EPI_FWD; g.out_scale = 1.f; g.aux_scale = 1.f;
  if (data == 2) g.lda = 0;
  const int tiles = (int)(sr_cdiv(M, C_::BM) * sr_cdiv(N, C_::BN));
  unsigned long long* stamps = nullptr;
  CK(hipMalloc(&stamps, (size_t)tiles * 64)); CK(hipMemset(stamps, 0, (size_t)tiles * 64));
  LabArgs lab; lab.stamps = nullptr; lab.stagger_mode = smode; lab.first_wave = 512; lab.persistent = pgrid > 0;
  const int nk = (K + BK - 1) / BK;
  lab.stagger_cycles = (int)(sfrac * 2.0 * nk * 64 * 64);     // a tile at the shared rate: 2 x nk steps x 64 MFMAs x 64 cycles
  const int grid = pgrid > 0 ? pgrid : tiles;
  void (*kern)(sr_gemm_args, LabArgs) = nullptr;
#define PICK(P) case P: kern = (K % BK) ? lab_kernel<true, P> : lab_kernel<false, P>; break;
  const int variant = argc > 13 ? atoi(argv[13]) : -1;   // >= 0: lab_tile<variant>
#define VPICK(P) case P: kern = lab_variant_kernel<P>; break;
  if (variant == 100) { kern = lab_v2_kernel; lds = 65536; }
  else if (variant >= 0) {
    switch (variant) { VPICK(0) VPICK(1) VPICK(2) VPICK(4) VPICK(6) VPICK(8) VPICK(16) VPICK(7) VPICK(23) default: fprintf(stderr, "variant not compiled\n"); return 2; }
  } else
  switch (prio) { PICK(0) PICK(1) PICK(2) PICK(3) default: fprintf(stderr, "prio 0..3\n"); return 2; }
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  auto launch = [&]() { hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, g, lab); };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("M=%d N=%d K=%d act=%d stagger=%d/%.2f grid=%d lds=%zu prio=%d data=%d variant=%d : %9.1f us  %7.1f TF/s\n", M, N, K, act, smode, sfrac, grid, lds, prio, data, variant,
         ms * 1e3, 2.0 * M * N * K / ms / 1e9);
  {
    std::vector<float> hc((size_t)M * N);
    CK(hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long sum = 0; for (size_t i = 0; i < hc.size(); ++i) { unsigned u; memcpy(&u, &hc[i], 4); sum = sum * 1099511628211ull + u; }
    printf("    C digest %016llx  C[0]=%g C[last]=%g\n", sum, hc[0], hc.back());
  }
  if (dump) {
    lab.stamps = stamps;
    launch(); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)tiles * 8);
    CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    {   // cycles per tile and CU: (last end - first start) of a CU's workgroups / their number, median over the CUs (clock-independent)
      std::vector<std::pair<unsigned long long, std::array<unsigned long long, 3>>> v;
      std::map<unsigned long long, std::array<unsigned long long, 3>> cu;
      for (int t = 0; t < tiles; ++t) {
        const unsigned long long key = (h[t * 8 + 5] << 32) | (h[t * 8 + 4] & 0xFF00ull);
        auto it = cu.find(key);
        if (it == cu.end()) cu[key] = {h[t * 8], h[t * 8 + 3], 1ull};
        else { it->second[0] = std::min(it->second[0], h[t * 8]); it->second[1] = std::max(it->second[1], h[t * 8 + 3]); it->second[2]++; }
      }
      std::vector<double> per;
      for (auto& kv : cu) per.push_back((double)(kv.second[1] - kv.second[0]) / kv.second[2]);
      std::sort(per.begin(), per.end());
      double loop = 0; for (int t = 0; t < tiles; ++t) loop += (double)(h[t * 8 + 2] - h[t * 8 + 1]); loop /= tiles;
      printf("    cycles per tile and CU: median %.0f (over %zu CUs); mean tile-loop duration %.0f; ideal %d\n", per[per.size() / 2], per.size(), loop, nk * 64 * 64);
    }
    FILE* f = fopen(dump, "w");
    fprintf(f, "tile,t_start,t_pro,t_loop,t_end,hw_id,xcc,block,wall100mhz\n");
    for (int t = 0; t < tiles; ++t)
      fprintf(f, "%d,%llu,%llu,%llu,%llu,%llu,%llu,%llu,%llu\n", t, h[t * 8], h[t * 8 + 1], h[t * 8 + 2], h[t * 8 + 3], h[t * 8 + 4], h[t * 8 + 5], h[t * 8 + 6], h[t * 8 + 7]);
    fclose(f);
  }
  return 0;
}
