/* TEST INFRASTRUCTURE ONLY -- a host stand-in for the few CUDA names MCGpu/CudaKernels.cu of the reference uses, so
 * that the reference's own marching-cubes kernels (d_mc_get_mesh_on_gpu, d_conver_ijkd_to_pindex, d_set_int,
 * d_scale_vertices, class MCGpu) compile with g++ and run sequentially on the CPU: one "thread" after the other in
 * (block, thread) order, atomics as plain read-modify-write.  Kernel launches `k<<<g,b>>>(...)` are rewritten to
 * SR_LAUNCH(k, g, b, ...) by the build recipe (oracle/Makefile) in a scratch copy under oracle/_ref/.  Never shipped. */
#ifndef SR_REF_CUDA_SHIM_H
#define SR_REF_CUDA_SHIM_H
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#define __global__
#define __host__
#define __device__
#define CUDA_VERSION 11000

struct sr_dim3 { int x, y, z; };
static sr_dim3 blockIdx, blockDim, gridDim, threadIdx;

template <class F> static inline void sr_launch_seq(int grid, int block, F body) {
  gridDim.x = grid; blockDim.x = block;
  for (int b = 0; b < grid; ++b)
    for (int t = 0; t < block; ++t) { blockIdx.x = b; threadIdx.x = t; body(); }
}
#define SR_LAUNCH(kernel, grid, block, ...) sr_launch_seq((grid), (block), [&]() { kernel(__VA_ARGS__); })

static inline int atomicAdd(int* p, int v) { int old = *p; *p = old + v; return old; }
static inline int atomicExch(int* p, int v) { int old = *p; *p = v; return old; }
static inline int atomicMax(int* p, int v) { int old = *p; if (v > old) *p = v; return old; }

typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
static inline const char* cudaGetErrorString(cudaError_t) { return "host shim"; }
static inline cudaError_t cudaPeekAtLastError() { return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline cudaError_t cudaSetDevice(int) { return cudaSuccess; }
template <class T> static inline cudaError_t cudaMalloc(T** p, size_t n) { *p = (T*)malloc(n ? n : 1); return cudaSuccess; }
static inline cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return cudaSuccess; }
static inline cudaError_t cudaMemset(void* d, int v, size_t n) { memset(d, v, n); return cudaSuccess; }

typedef int cublasStatus_t;
enum { CUBLAS_STATUS_SUCCESS, CUBLAS_STATUS_NOT_INITIALIZED, CUBLAS_STATUS_ALLOC_FAILED, CUBLAS_STATUS_INVALID_VALUE,
       CUBLAS_STATUS_ARCH_MISMATCH, CUBLAS_STATUS_MAPPING_ERROR, CUBLAS_STATUS_EXECUTION_FAILED, CUBLAS_STATUS_INTERNAL_ERROR,
       CUBLAS_STATUS_NOT_SUPPORTED, CUBLAS_STATUS_LICENSE_ERROR };
#endif
