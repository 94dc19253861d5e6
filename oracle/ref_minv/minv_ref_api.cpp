/* TEST INFRASTRUCTURE ONLY -- C entry points around the reference's OWN 3x3-inverse kernels (cu3x3MInv / cu3x3MInv_backward and their
 * four launchers, /root/reference/FastMinv/Matrix3x3InvKernels.cu:22-141), compiled for the host through ../ref_mc/shim/cuda.h
 * (threads run one after the other).  The launch lines `k<T><<<blocks,threads>>>(...)` are rewritten to SR_LAUNCH(...) by
 * oracle/Makefile in a scratch copy under oracle/_ref/ (git-ignored; no reference source enters the repository).  Never shipped. */
#include <cmath>
using std::fabs;
#include SR_REF_MINV_KERNELS      /* oracle/_ref/minv_ref_kernels.cpp */
#include <cstdint>

extern "C" void minv_ref_fwd_f32(const float* ms, float* invs, uint8_t* checks, int n) { M3x3Inv_float(ms, invs, reinterpret_cast<bool*>(checks), n); }
extern "C" void minv_ref_fwd_f64(const double* ms, double* invs, uint8_t* checks, int n) { M3x3Inv_double(ms, invs, reinterpret_cast<bool*>(checks), n); }
extern "C" void minv_ref_bwd_f32(const float* g, const float* invs, float* outs, int n) { M3x3Inv_backward_float(g, invs, outs, n); }
extern "C" void minv_ref_bwd_f64(const double* g, const double* invs, double* outs, int n) { M3x3Inv_backward_double(g, invs, outs, n); }
