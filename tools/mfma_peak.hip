// Sustained MFMA rate of this box: what the fp32 (v_mfma_f32_32x32x2_f32) and bf16 (v_mfma_f32_32x32x16_bf16) pipes deliver when a
// kernel does nothing but issue independent MFMAs from registers -- the practical ceiling the layer-GEMM kernels are measured against
// (DESIGN.md 3.1), run for a few seconds so that the power-managed shader clock settles (tools/mfma_peak.sh samples rocm-smi beside it).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/_bin/mfma_peak ;  tools/_bin/mfma_peak [seconds per test]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void k_f32(float* out, int iters, float a, float b) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i + r);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k_bf16(float* out, int iters, float av, float bv) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i + r);
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(av + e); b[e] = (__bf16)(bv - e); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

// the gfx90a-generation bf16 instruction (64-bit operands, k = 8 per issue): the one that does NOT disturb neighbouring kernels
typedef short short4v __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k_bf16_1k(float* out, int iters, float av, float bv) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i + r);
  const short4v a = {(short)(0x3c00 + (int)av), 0x3c10, 0x3c20, 0x3c30}, b = {0x3d00, (short)(0x3d10 + (int)bv), 0x3d20, 0x3d30};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 123.456f) out[0] = s;
}

template <class K>
static void run(const char* name, K kernel, double flop_per_mfma, int nacc, int wg_per_cu, double seconds) {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = cus * wg_per_cu;
  int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {                                      // calibrate, then the long run
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 /*waves*/ * (double)iters * 8 * nacc * flop_per_mfma;
    if (rep == 1) printf("%-44s %2d WG/CU x %d acc: %8.1f TFLOP/s  (%.2f s, %d CUs)\n", name, wg_per_cu, nacc, flop / (ms * 1e-3) / 1e12, ms * 1e-3, cus);
    else iters = (int)(iters * (seconds * 1e3 / ms));
  }
  hipFree(out);
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 2.0;
  run("fp32  v_mfma_f32_32x32x2_f32", k_f32<4>, 32.0 * 32 * 2 * 2, 4, 1, secs);
  run("fp32  v_mfma_f32_32x32x2_f32", k_f32<4>, 32.0 * 32 * 2 * 2, 4, 2, secs);
  run("bf16  v_mfma_f32_32x32x16_bf16", k_bf16<4>, 32.0 * 32 * 16 * 2, 4, 1, secs);
  run("bf16  v_mfma_f32_32x32x16_bf16", k_bf16<4>, 32.0 * 32 * 16 * 2, 4, 2, secs);
  run("bf16  v_mfma_f32_32x32x8_bf16_1k", k_bf16_1k<4>, 32.0 * 32 * 8 * 2, 4, 1, secs);
  run("bf16  v_mfma_f32_32x32x8_bf16_1k", k_bf16_1k<4>, 32.0 * 32 * 8 * 2, 4, 2, secs);
  return 0;
}
