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

// ------------------------------------------------------------------------------------------------
// Effective-weight packing and its backward for ALL layers of a network in one launch each.
// pack : W[n, :] = g[n] v[n, :] / |v[n, :]| (weight_norm dim 0, model/network.py:65-66; plain copy when g is null), padded to the
//        row pitch; WT = W^T padded (feeds the backward-data GEMM); norms[n] = |v[n, :]|.  The reference recomputes the weight norm
//        in a forward pre-hook at every module call; torch ops here took ~5 launches per layer per optimizer step.
// unpack (backward): gv[n, :] = (g/|v|) (dW[n, :] - v[n, :] <dW[n, :], v[n, :]> / |v|^2),  gg[n] = <dW[n, :], v[n, :]> / |v|
//        (aten::_weight_norm_interface_backward), or gw = dW[:, :K] for a plain layer; optionally added to existing gradients.
// One wave per weight row.
namespace {
__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

__global__ __launch_bounds__(256) void pack_weights_kernel(sr_pack_table t) {
  const sr_pack_layer L = t.layer[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = gridDim.x * (blockDim.x >> 6);
  for (int n = wave; n < L.ldwt; n += nwaves) {          // rows [N, ldwt) only zero the pad columns of WT
    if (n < L.N) {
      const float* v = L.v + (int64_t)n * L.K;
      float scale = 1.f;
      if (L.g) {
        float ss = 0.f;
        for (int k = lane; k < L.K; k += 64) ss += v[k] * v[k];
        const float nrm = sqrtf(wave_sum64(ss));
        if (lane == 0) L.norms[n] = nrm;
        scale = L.g[n] / nrm;
      }
      for (int k = lane; k < L.ldw; k += 64) {
        const float w = k < L.K ? v[k] * scale : 0.f;
        L.W[(int64_t)n * L.ldw + k] = w;
        if (k < L.K) L.WT[(int64_t)k * L.ldwt + n] = w;
      }
    } else {
      for (int k = lane; k < L.K; k += 64) L.WT[(int64_t)k * L.ldwt + n] = 0.f;
    }
  }
}

__global__ __launch_bounds__(256) void unpack_grads_kernel(sr_unpack_table t) {
  const sr_unpack_layer L = t.layer[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = gridDim.x * (blockDim.x >> 6);
  if (L.gb && blockIdx.x == 0)                                   // bias gradient of the layer, accumulated in place
    for (int n = threadIdx.x; n < L.N; n += blockDim.x) L.gb[n] += L.db[n];
  for (int n = wave; n < L.N; n += nwaves) {
    const float* d = L.dW + (int64_t)n * L.lddw;
    float* gv = L.gv + (int64_t)n * L.K;
    if (L.g) {
      const float* v = L.v + (int64_t)n * L.K;
      float dot = 0.f;
      for (int k = lane; k < L.K; k += 64) dot += d[k] * v[k];
      dot = wave_sum64(dot);
      const float nrm = L.norms[n], inv = 1.f / nrm;
      const float a = L.g[n] * inv, b = dot * inv * inv;
      for (int k = lane; k < L.K; k += 64) {
        const float val = a * (d[k] - v[k] * b);
        gv[k] = L.accumulate ? gv[k] + val : val;
      }
      if (lane == 0) L.gg[n] = (L.accumulate ? L.gg[n] : 0.f) + dot * inv;
    } else {
      for (int k = lane; k < L.K; k += 64) gv[k] = L.accumulate ? gv[k] + d[k] : d[k];
    }
  }
}

// dst [rows * group, ldd]: row r*group + g = (g == 0 ? a : b)[r, :n] zero padded to `width` columns (a NULL source gives a zero row).
__global__ __launch_bounds__(256) void rows_pad_kernel(const float* __restrict__ a, int64_t lda, int na, const float* __restrict__ b, int64_t ldb, int nb,
                                                       int64_t rows, int group, float* __restrict__ dst, int64_t ldd, int width) {
  const int64_t total = rows * group * (int64_t)width;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / width;
    const int col = (int)(i - row * width);
    const int64_t r = row / group;
    const int g = (int)(row - r * group);
    const float* src = g == 0 ? a : b;
    const int n = g == 0 ? na : nb;
    const int64_t ld = g == 0 ? lda : ldb;
    dst[row * ldd + col] = (src && col < n) ? src[r * ld + col] : 0.f;
  }
}

// ---- cross-stream ordering through device memory (see the header: sr_stream_flag_set / _wait) ----
__global__ void stream_stamp_kernel(unsigned long long* out) {
  if (threadIdx.x == 0) *out = wall_clock64();
}
__global__ void stream_flag_set_kernel(unsigned* flag, unsigned value) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void stream_flag_wait_kernel(const unsigned* flag, unsigned value, unsigned* timed_out, unsigned long long max_ticks) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();                       // 100 MHz constant counter
  while ((int)(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - value) < 0) {
    __builtin_amdgcn_s_sleep(64);
    if (wall_clock64() - t0 > max_ticks) { if (timed_out) atomicAdd(timed_out, 1u); break; }   // never hang the queue: the caller checks the counter
  }
}
// Adam step of up to SR_ADAM_MAX_TENSORS parameter tensors in ONE launch (torch.optim.Adam without weight decay / amsgrad, the
// optimizer of train.py:139: m <- m + (1 - b1)(g - m);  v <- b2 v + (1 - b2) g g;  p <- p - lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)).
// torch's multi-tensor implementation issues ~10 launches with 40 us of host time between them at the end of every iteration.
__global__ __launch_bounds__(256) void adam_step_kernel(sr_adam_table t) {
  const sr_adam_tensor T = t.tensor[blockIdx.y];
  const float b2 = t.beta2, eps = t.eps, w1 = t.one_minus_beta1, w2 = t.one_minus_beta2;
  const float step_size = T.lr / T.bias1, inv_sqrt_bias2 = T.inv_sqrt_bias2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < T.numel; i += (int64_t)gridDim.x * blockDim.x) {
    const float g = T.g[i];
    float m = T.m[i], v = T.v[i];
    m = m + (g - m) * w1;
    v = v * b2 + w2 * g * g;
    T.m[i] = m; T.v[i] = v;
    const float denom = sqrtf(v) * inv_sqrt_bias2 + eps;
    T.p[i] = T.p[i] - step_size * (m / denom);
  }
}

// out[f, c] = sum over the rows r with index[r] == f of X[r, c]   (n frames <= 32; the backward of gathering a per-frame code into
// every row of a batch: model/Deformer.py:61,75 `conds[batch_inds]`).  torch's index_add does this with float atomics -- a result
// that changes from run to run.  Here every thread owns one (row group, column) lane and adds its rows in program order into its own
// LDS cell per frame; the four row groups of a workgroup, then the row slices of the grid, are folded in a fixed order (the last
// fold in double precision): bit-reproducible.
constexpr int kFsCols = 64, kFsGroups = 4;
__global__ __launch_bounds__(kFsCols * kFsGroups) void rows_frame_sum_kernel(const float* __restrict__ X, int64_t ldx, int64_t P, int E,
                                                                              const int64_t* __restrict__ index, int n, int64_t rows_per_slice,
                                                                              float* __restrict__ partial) {
  extern __shared__ float fs_acc[];                        // [groups][n][cols]
  const int c = threadIdx.x % kFsCols, rg = threadIdx.x / kFsCols;
  const int col = blockIdx.x * kFsCols + c;
  float* mine = fs_acc + (int64_t)rg * n * kFsCols + c;
  for (int f = 0; f < n; ++f) mine[f * kFsCols] = 0.f;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_slice, r1 = min(P, r0 + rows_per_slice);
  if (col < E)
    for (int64_t r = r0 + rg; r < r1; r += kFsGroups) {
      const int f = (int)index[r];
      if (f >= 0 && f < n) mine[f * kFsCols] += X[r * ldx + col];
    }
  __syncthreads();
  if (rg == 0 && col < E)
    for (int f = 0; f < n; ++f) {
      const float* a = fs_acc + f * kFsCols + c;
      partial[((int64_t)blockIdx.y * n + f) * E + col] = ((a[0] + a[(int64_t)n * kFsCols]) + a[(int64_t)2 * n * kFsCols]) + a[(int64_t)3 * n * kFsCols];
    }
}
__global__ __launch_bounds__(256) void rows_frame_sum_finish(const float* __restrict__ partial, int slices, int64_t ne, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ne; i += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int b = 0; b < slices; ++b) s += (double)partial[(int64_t)b * ne + i];
    out[i] = (float)s;
  }
}
static int fs_slices(int64_t P) {
  int64_t s = sr_cdiv(P, 512);
  return (int)(s > 256 ? 256 : (s < 1 ? 1 : s));
}
}  // namespace

extern "C" {
int sr_rows_pad(const float* a, int64_t lda, int32_t na, const float* b, int64_t ldb, int32_t nb, int64_t rows, int32_t group, float* dst, int64_t ldd,
                int32_t width, void* stream) {
  if (rows < 0 || (group != 1 && group != 2) || width < 1 || ldd < width || na < 0 || nb < 0 || na > width || nb > width) return SR_EINVAL;
  if (rows == 0) return SR_OK;
  if (!dst || (a && lda < na) || (b && ldb < nb)) return SR_EINVAL;
  hipLaunchKernelGGL(rows_pad_kernel, dim3(sr_stream_grid(rows * group * (int64_t)width, 256)), dim3(256), 0, (hipStream_t)stream, a, lda, na, b, ldb, nb,
                     rows, group, dst, ldd, width);
  return sr_launch_status();
}
int64_t sr_rows_frame_sum_workspace_floats(int64_t P, int32_t E, int32_t n) {
  if (P < 0 || E <= 0 || n <= 0) return SR_EINVAL;
  return (int64_t)fs_slices(P) * n * E;
}
int sr_rows_frame_sum(const float* X, int64_t ldx, int64_t P, int32_t E, const int64_t* index, int32_t n, float* partial, float* out, void* stream) {
  if (P < 0 || E <= 0 || n <= 0 || n > 32 || ldx < E || !out) return SR_EINVAL;
  if (P == 0) return hipMemsetAsync(out, 0, sizeof(float) * n * E, (hipStream_t)stream) == hipSuccess ? SR_OK : SR_ELAUNCH;
  if (!X || !index || !partial) return SR_EINVAL;
  const int slices = fs_slices(P);
  const int64_t rows_per_slice = sr_cdiv(P, slices);
  hipLaunchKernelGGL(rows_frame_sum_kernel, dim3((unsigned)sr_cdiv(E, kFsCols), slices), dim3(kFsCols * kFsGroups), sizeof(float) * kFsGroups * n * kFsCols,
                     (hipStream_t)stream, X, ldx, P, E, index, n, rows_per_slice, partial);
  hipLaunchKernelGGL(rows_frame_sum_finish, dim3(sr_stream_grid((int64_t)n * E, 256)), dim3(256), 0, (hipStream_t)stream, partial, slices, (int64_t)n * E, out);
  return sr_launch_status();
}
int sr_stream_stamp(uint64_t* out, void* stream) {
  if (!out) return SR_EINVAL;
  hipLaunchKernelGGL(stream_stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out);
  return sr_launch_status();
}
int sr_stream_flag_set(uint32_t* flag, uint32_t value, void* stream) {
  if (!flag) return SR_EINVAL;
  hipLaunchKernelGGL(stream_flag_set_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, value);
  return sr_launch_status();
}
int sr_stream_flag_wait(const uint32_t* flag, uint32_t value, uint32_t* timed_out, int32_t timeout_ms, void* stream) {
  if (!flag || timeout_ms <= 0) return SR_EINVAL;
  hipLaunchKernelGGL(stream_flag_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, value, timed_out, (unsigned long long)timeout_ms * 100000ull);
  return sr_launch_status();
}
int sr_adam_step(const sr_adam_table* t, void* stream) {
  if (!t || t->ntensors < 1 || t->ntensors > SR_ADAM_MAX_TENSORS || !(t->beta1 >= 0.f && t->beta1 < 1.f) || !(t->beta2 >= 0.f && t->beta2 < 1.f)) return SR_EINVAL;
  for (int k = 0; k < t->ntensors; ++k) {
    const sr_adam_tensor& T = t->tensor[k];
    if (!T.p || !T.g || !T.m || !T.v || T.numel <= 0 || !(T.bias1 > 0.f) || !(T.inv_sqrt_bias2 > 0.f)) return SR_EINVAL;
  }
  hipLaunchKernelGGL(adam_step_kernel, dim3(64, t->ntensors), dim3(256), 0, (hipStream_t)stream, *t);
  return sr_launch_status();
}
int sr_pack_weights(const sr_pack_table* t, void* stream) {
  if (!t || t->nlayers < 1 || t->nlayers > SR_PACK_MAX_LAYERS) return SR_EINVAL;
  for (int l = 0; l < t->nlayers; ++l) {
    const sr_pack_layer& L = t->layer[l];
    if (!L.v || !L.W || !L.WT || L.N <= 0 || L.K <= 0 || L.ldw < L.K || L.ldwt < L.N || (L.g && !L.norms)) return SR_EINVAL;
  }
  hipLaunchKernelGGL(pack_weights_kernel, dim3(32, t->nlayers), dim3(256), 0, (hipStream_t)stream, *t);
  return sr_launch_status();
}
int sr_unpack_grads(const sr_unpack_table* t, void* stream) {
  if (!t || t->nlayers < 1 || t->nlayers > SR_PACK_MAX_LAYERS) return SR_EINVAL;
  for (int l = 0; l < t->nlayers; ++l) {
    const sr_unpack_layer& L = t->layer[l];
    if (!L.dW || !L.gv || L.N <= 0 || L.K <= 0 || L.lddw < L.K || (L.g && (!L.v || !L.norms || !L.gg)) || (L.gb && !L.db)) return SR_EINVAL;
  }
  hipLaunchKernelGGL(unpack_grads_kernel, dim3(32, t->nlayers), dim3(256), 0, (hipStream_t)stream, *t);
  return sr_launch_status();
}
}
