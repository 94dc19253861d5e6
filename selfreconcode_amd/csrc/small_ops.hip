// Small per-element kernels around the MLPs.
//  * sr_svd3x3: batched 3x3 SVD (one thread per matrix, cyclic Jacobi on J^T J) -- removes the
//    device -> host -> device round trip of model/network.py:576 (`torch.svd(Jacobs.cpu())`) from the
//    deformation regulariser (pulled forward because a 60k-matrix CPU SVD
//    would dominate the iteration).  Singular values descending, like torch.svd.
//  (the point-silhouette renderer and the mesh rasteriser live in raster.hip)
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

}  // namespace

extern "C" {
int sr_svd3x3(const float* A, int64_t n, float* U, float* S, float* V, void* stream) {
  if (n < 0) return SR_EINVAL;
  if (n == 0) return SR_OK;
  if (!A || !U || !S || !V) return SR_EINVAL;
  hipLaunchKernelGGL(svd3_kernel, dim3(sr_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, A, n, U, S, V);
  return sr_launch_status();
}
}
