/* TEST INFRASTRUCTURE ONLY -- C entry points around the reference's OWN grid-sampler kernels (grid_sampler_3d_kernel,
 * grid_sampler_3d_backward_kernel, grid_sampler_3d_backward_backward_kernel: /root/reference/MCAcc/cuda/GridSamplerMineKernel.cu:162-914),
 * compiled for the host through shim/sr_aten_shim.h; one "thread" walks all output points (CUDA_KERNEL_LOOP), atomics are plain adds, so the
 * accumulation order of grad_input is the point order.  SR_REF_GS_KERNELS is the scratch copy of lines 1-917 of that file made by
 * oracle/Makefile under oracle/_ref/ (git-ignored; no reference source enters the repository); it leaves `namespace at { namespace native {`
 * open, the wrappers below live inside it.  Tensors are described as the kernels see them: data pointer + 5 sizes + 5 strides (elements).
 * Never shipped. */
#include SR_REF_GS_KERNELS

template <class T>
static TensorInfo<T, int> info5(T* data, const int64_t* sizes, const int64_t* strides) {
  TensorInfo<T, int> t;
  t.data = data; t.dims = 5;
  for (int i = 0; i < 5; ++i) { t.sizes[i] = (int)sizes[i]; t.strides[i] = (int)strides[i]; }
  return t;
}
template <class F> static void one_thread(F body) { sr_launch_seq(1, 1, body); }

template <class T>
static void gs_fwd(const T* input, const int64_t* isz, const int64_t* ist, const T* grid, const int64_t* gsz, const int64_t* gst,
                   T* out, const int64_t* osz, const int64_t* ost, int interp, int pad) {
  const int count = (int)(isz[0] * gsz[1] * gsz[2] * gsz[3]);
  one_thread([&]() {
    grid_sampler_3d_kernel<T>(count, info5(const_cast<T*>(input), isz, ist), info5(const_cast<T*>(grid), gsz, gst), info5(out, osz, ost),
                              static_cast<GridSamplerInterpolation>(interp), static_cast<GridSamplerPadding>(pad));
  });
}
template <class T>
static void gs_bwd(const T* gout, const int64_t* gosz, const int64_t* gost, const T* input, const int64_t* isz, const int64_t* ist,
                   const T* grid, const int64_t* gsz, const int64_t* gst, T* ginput, const int64_t* gist, T* ggrid, const int64_t* ggst,
                   int interp, int pad) {
  const int count = (int)(isz[0] * gsz[1] * gsz[2] * gsz[3]);
  one_thread([&]() {
    grid_sampler_3d_backward_kernel<T>(count, info5(const_cast<T*>(gout), gosz, gost), info5(const_cast<T*>(input), isz, ist),
                                       info5(const_cast<T*>(grid), gsz, gst), info5(ginput, isz, gist), info5(ggrid, gsz, ggst),
                                       static_cast<GridSamplerInterpolation>(interp), static_cast<GridSamplerPadding>(pad));
  });
}
template <class T>
static void gs_dbwd(const T* ggi, const int64_t* ggist, const T* ggg, const int64_t* gggst, const T* gout, const int64_t* gosz, const int64_t* gost,
                    const T* input, const int64_t* isz, const int64_t* ist, const T* grid, const int64_t* gsz, const int64_t* gst,
                    T* ginput, const int64_t* gist, T* ggrid, const int64_t* ggst, T* ggout, const int64_t* ggost, int interp, int pad) {
  const int count = (int)(isz[0] * gsz[1] * gsz[2] * gsz[3]);
  one_thread([&]() {
    grid_sampler_3d_backward_backward_kernel<T>(count, info5(const_cast<T*>(ggi), isz, ggist), info5(const_cast<T*>(ggg), gsz, gggst),
                                                info5(const_cast<T*>(gout), gosz, gost), info5(const_cast<T*>(input), isz, ist),
                                                info5(const_cast<T*>(grid), gsz, gst), info5(ginput, isz, gist), info5(ggrid, gsz, ggst),
                                                info5(ggout, gosz, ggost), static_cast<GridSamplerInterpolation>(interp),
                                                static_cast<GridSamplerPadding>(pad));
  });
}
}}  // namespace at::native (opened by the scratch copy)

#define SR_GS_API(SUF, T)                                                                                                                        \
  extern "C" void gs_ref_fwd_##SUF(const T* input, const int64_t* isz, const int64_t* ist, const T* grid, const int64_t* gsz, const int64_t* gst, \
                                   T* out, const int64_t* osz, const int64_t* ost, int interp, int pad) {                                        \
    at::native::gs_fwd<T>(input, isz, ist, grid, gsz, gst, out, osz, ost, interp, pad);                                                           \
  }                                                                                                                                              \
  extern "C" void gs_ref_bwd_##SUF(const T* gout, const int64_t* gosz, const int64_t* gost, const T* input, const int64_t* isz,                   \
                                   const int64_t* ist, const T* grid, const int64_t* gsz, const int64_t* gst, T* ginput, const int64_t* gist,    \
                                   T* ggrid, const int64_t* ggst, int interp, int pad) {                                                          \
    at::native::gs_bwd<T>(gout, gosz, gost, input, isz, ist, grid, gsz, gst, ginput, gist, ggrid, ggst, interp, pad);                             \
  }                                                                                                                                              \
  extern "C" void gs_ref_dbwd_##SUF(const T* ggi, const int64_t* ggist, const T* ggg, const int64_t* gggst, const T* gout, const int64_t* gosz,   \
                                    const int64_t* gost, const T* input, const int64_t* isz, const int64_t* ist, const T* grid,                   \
                                    const int64_t* gsz, const int64_t* gst, T* ginput, const int64_t* gist, T* ggrid, const int64_t* ggst,        \
                                    T* ggout, const int64_t* ggost, int interp, int pad) {                                                        \
    at::native::gs_dbwd<T>(ggi, ggist, ggg, gggst, gout, gosz, gost, input, isz, ist, grid, gsz, gst, ginput, gist, ggrid, ggst, ggout, ggost,    \
                           interp, pad);                                                                                                          \
  }
SR_GS_API(f32, float)
SR_GS_API(f64, double)
