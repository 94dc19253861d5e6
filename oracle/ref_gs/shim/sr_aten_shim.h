/* TEST INFRASTRUCTURE ONLY -- host stand-ins for the few ATen names the KERNELS of the reference's grid sampler use
 * (/root/reference/MCAcc/cuda/GridSamplerMineKernel.cu:1-914: TensorInfo, the two GridSampler enums, CUDA_KERNEL_LOOP, atomicAdd), so that
 * grid_sampler_3d_kernel / _backward_kernel / _backward_backward_kernel compile with g++ and run sequentially on the CPU.  The ATen
 * launcher functions behind them (:918-1022: at::empty / AT_DISPATCH / <<<...>>>) are not part of the scratch copy the Makefile makes;
 * oracle/ref_gs/gs_ref_api.cpp calls the kernels directly.  Never shipped. */
#ifndef SR_ATEN_SHIM_H
#define SR_ATEN_SHIM_H
#include <math.h>
#include <climits>
#include <cstdint>
#include <algorithm>
#include "../../ref_mc/shim/cuda.h"
using std::min;
using std::max;
#define __launch_bounds__(n)
#define __forceinline__ inline
#define CUDA_KERNEL_LOOP(i, n) for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += blockDim.x * gridDim.x)
static inline float atomicAdd(float* p, float v) { float old = *p; *p = old + v; return old; }
static inline double atomicAdd(double* p, double v) { double old = *p; *p = old + v; return old; }
namespace at {
namespace cuda { namespace detail {
template <typename T, typename IndexType>
struct TensorInfo {
  T* data;
  IndexType sizes[8];
  IndexType strides[8];
  int dims;
};
}}
namespace native { namespace detail {
enum class GridSamplerInterpolation { Bilinear, Nearest };
enum class GridSamplerPadding { Zeros, Border, Reflection };
}}
}
#endif
