/* selfrecon_hip.h -- flat C ABI of libselfrecon_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary of the SelfRecon per-frame SDF-optimisation hot path: every entry
 * point replaces one pybind/CUDA surface of the reference (cited per function, paths
 * relative to the reference repository).  Conventions, all functions:
 *   - return 0 on success, a negative SR_E* code otherwise; nothing is allocated inside;
 *     the caller owns every buffer; all pointers are DEVICE pointers unless named host_*;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are
 *     asynchronous and re-entrant per stream; no global mutable state except explicit
 *     workspaces passed by the caller;
 *   - tensors are dense row-major unless strides are passed explicitly (in ELEMENTS).
 * The Python host (selfreconcode_amd/) maps non-zero codes to the reference's own error
 * convention (exception; empty list for mc_gpu -- MCGpu/MCGpu.cpp:41-48).
 */
#ifndef SELFRECON_HIP_H_
#define SELFRECON_HIP_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR_OK 0
#define SR_EINVAL (-1)   /* bad argument (null pointer, non-positive size, unsupported mode) */
#define SR_ELAUNCH (-2)  /* hipGetLastError() after launch != hipSuccess */
#define SR_ENOSPC (-3)   /* caller-provided workspace / output capacity too small */

int sr_abi_version(void);            /* bumps when a signature below changes */
const char* sr_build_arch(void);     /* "gfx950" */
const char* sr_build_digest(void);   /* sha256 of the sources the library was built from (selfreconcode_amd/build.py) */

/* ---------------------------------------------------------------- FastMinv (a7)
 * Replaces FastMinv/M3x3Inv.cpp:12-38 Fast3x3Minv -> Matrix3x3InvKernels.cu:22-61 and
 * M3x3Inv.cpp:40-58 Fast3x3Minv_backward -> Matrix3x3InvKernels.cu:64-104.
 * ms/invs/grads/outs: [n,3,3]; checks: [n] bytes (0/1).  Singular rule: |det| < 1e-4
 * -> inverse = 0, check = 0.  backward: out = -(C^T G C^T), C = saved inverse. */
int sr_minv3x3_fwd_f32(const float* ms, float* invs, uint8_t* checks, int64_t n, void* stream);
int sr_minv3x3_fwd_f64(const double* ms, double* invs, uint8_t* checks, int64_t n, void* stream);
int sr_minv3x3_bwd_f32(const float* grads, const float* invs, float* outs, int64_t n, void* stream);
int sr_minv3x3_bwd_f64(const double* grads, const double* invs, double* outs, int64_t n, void* stream);

/* ---------------------------------------------------------------- GridSamplerMine (a6)
 * Replaces MCAcc/cuda/GridSamplerMine.cpp:73-104 (forward / backward / dbackward) ->
 * GridSamplerMineKernel.cu:162-328, 333-570, 575-914.  Trilinear, border padding,
 * align_corners=False only (the only mode the reference accepts, GridSamplerMine.cpp:59-64).
 * A tensor is described by its 5 sizes / strides in elements: input [N,C,D,H,W];
 * grid [N,Do,Ho,Wo,3]; output-like tensors [N,C,Do,Ho,Wo].  grad_grid is dense [N,Do,Ho,Wo,3]
 * (the reference assumes the same, Kernel.cu:538-545).  grad_input may be NULL (skip it: the
 * skinning-weight volume is a buffer) -- when given it must be ZERO-FILLED by the caller and has
 * the input's sizes with its own strides. */
typedef struct {
  int64_t size[5];
  int64_t stride[5];
} sr_tensor5;

int sr_gridsample3d_fwd_f32(const float* input, sr_tensor5 in_d, const float* grid, sr_tensor5 grid_d,
                            float* output, sr_tensor5 out_d, void* stream);
int sr_gridsample3d_fwd_f64(const double* input, sr_tensor5 in_d, const double* grid, sr_tensor5 grid_d,
                            double* output, sr_tensor5 out_d, void* stream);
int sr_gridsample3d_bwd_f32(const float* input, sr_tensor5 in_d, const float* grid, sr_tensor5 grid_d,
                            const float* grad_output, sr_tensor5 gout_d,
                            float* grad_input /*nullable*/, sr_tensor5 gin_d, float* grad_grid, void* stream);
int sr_gridsample3d_bwd_f64(const double* input, sr_tensor5 in_d, const double* grid, sr_tensor5 grid_d,
                            const double* grad_output, sr_tensor5 gout_d,
                            double* grad_input /*nullable*/, sr_tensor5 gin_d, double* grad_grid, void* stream);
/* fp16 (IEEE binary16 storage, `void*` here; the reference dispatches AT_DISPATCH_FLOATING_TYPES_AND_HALF at
 * GridSamplerMineKernel.cu:931,963,1001): same kernels on the native half type, every arithmetic step rounded to half as
 * at::Half arithmetic is; grad_input accumulates with a 16-bit CAS add. */
int sr_gridsample3d_fwd_f16(const void* input, sr_tensor5 in_d, const void* grid, sr_tensor5 grid_d, void* output, sr_tensor5 out_d,
                            void* stream);
int sr_gridsample3d_bwd_f16(const void* input, sr_tensor5 in_d, const void* grid, sr_tensor5 grid_d, const void* grad_output,
                            sr_tensor5 gout_d, void* grad_input /*nullable*/, sr_tensor5 gin_d, void* grad_grid, void* stream);
int sr_gridsample3d_dbwd_f16(const void* gO_input /*nullable*/, sr_tensor5 goi_d, const void* gO_grid, sr_tensor5 gog_d,
                             const void* input, sr_tensor5 in_d, const void* grid, sr_tensor5 grid_d, const void* grad_output,
                             sr_tensor5 gout_d, void* grad_input /*nullable*/, sr_tensor5 gin_d, void* grad_grid,
                             void* grad_grad_output, void* stream);
/* double backward: cotangents (gO_input [like input] nullable = zeros, gO_grid [N,Do,Ho,Wo,3]
 * with strides) of the backward's two outputs -> grad_input (nullable, zero-filled by caller),
 * grad_grid (dense), grad_grad_output (dense [N,C,Do,Ho,Wo]). */
int sr_gridsample3d_dbwd_f32(const float* gO_input /*nullable*/, sr_tensor5 goi_d, const float* gO_grid, sr_tensor5 gog_d,
                             const float* input, sr_tensor5 in_d, const float* grid, sr_tensor5 grid_d,
                             const float* grad_output, sr_tensor5 gout_d,
                             float* grad_input /*nullable*/, sr_tensor5 gin_d, float* grad_grid, float* grad_grad_output,
                             void* stream);
int sr_gridsample3d_dbwd_f64(const double* gO_input /*nullable*/, sr_tensor5 goi_d, const double* gO_grid, sr_tensor5 gog_d,
                             const double* input, sr_tensor5 in_d, const double* grid, sr_tensor5 grid_d,
                             const double* grad_output, sr_tensor5 gout_d,
                             double* grad_input /*nullable*/, sr_tensor5 gin_d, double* grad_grid, double* grad_grad_output,
                             void* stream);


/* ---------------------------------------------------------------- MLP layer kernels (a2, a3, a4, a10)
 * Replace the per-layer nn.Linear (cuBLAS) + activation + torch.cat launches of
 * model/network.py:83-95 (ImplicitNetwork.forward), model/Deformer.py:66-71 (MLPTranslator),
 * model/RenderNet.py:80-88, and their autograd backward / double-backward passes
 * (network.py:102-114, utils/utils.py:106-120).  Exact fp32 on the MFMA pipe
 * (v_mfma_f32_32x32x2_f32); rows are "tangent-interleaved": each sample owns `group` (1, 2 or 4)
 * consecutive rows = primal + forward-mode tangents.
 *
 * sr_mlp_gemm_nt:  C[M, N(+naux_fwd)] = epilogue( A[M,K] * B[N,K]^T )
 *   SR_EPI_FWD : primal rows  c = act(acc + bias) * out_scale
 *                tangent rows c = act'(primal pre-activation) * acc * out_scale
 *                columns [N, N+naux_fwd) = aux[:, 0:naux_fwd] * out_scale       (skip concat, network.py:88-89)
 *   SR_EPI_BWD : acc is the cotangent of the STORED activations `aux` (= aux_scale * act(z), same row layout)
 *                columns < nact_bwd: primal  c = act' * aux_scale * acc + (act''/act') * sum_t aux_t * acc_t
 *                                    tangent c = act' * aux_scale * acc
 *                columns >= nact_bwd: c = acc * out_scale                       (cotangent of the skip filler)
 * Requirements: lda, ldb multiples of 4 floats and A, B 16-byte aligned; M % group == 0. */
#define SR_ACT_NONE 0
#define SR_ACT_SOFTPLUS100 1   /* torch.nn.Softplus(beta=100, threshold=20), network.py:70 */
#define SR_ACT_RELU 2
#define SR_EPI_FWD 0
#define SR_EPI_BWD 1
typedef struct {
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  int32_t M, N, K;
  const float* bias;      /* [N] or NULL; primal rows only */
  int32_t group, act, mode;
  float out_scale;
  const float* aux; int64_t ldaux;
  int32_t naux_fwd;       /* FWD: number of filler columns appended after N */
  int32_t nact_bwd;       /* BWD: leading columns that go through the activation derivative */
  float aux_scale;        /* BWD: scale the stored activations carry */
} sr_gemm_args;
int sr_mlp_gemm_nt(const sr_gemm_args* host_args, void* stream);

/* Layer chain: up to SR_CHAIN_MAX_LAYERS consecutive layer GEMMs of one or two independent networks (layer l of both side
 * by side in one grid) on a row count read from DEVICE memory: rows = *m_dev * m_mul <= m_cap * m_mul (every g[l][p].M is
 * ignored; g[l][p].group must equal m_mul).  g[l+1][p].A may be g[l][p].C.  One launch per layer, grids sized for m_cap,
 * workgroups beyond the live tiles return at once; no host synchronisation.
 * This is the MLP half of the reference's utils/FindSurfacePs.py:129-162 loop body on whatever number of rays is still
 * unfinished, without the host ever learning that number. */
#define SR_CHAIN_MAX_LAYERS 10
typedef struct {
  int32_t nlayers;
  int32_t nprob[SR_CHAIN_MAX_LAYERS];
  sr_gemm_args g[SR_CHAIN_MAX_LAYERS][2];
  const int32_t* m_dev; int32_t m_mul;
  int32_t m_cap;                   /* upper bound of *m_dev (sizes the grids) */
} sr_chain_args;
int sr_mlp_chain(const sr_chain_args* host_args, void* stream);

/* Weight gradient: dW[N, lddw] (+)= sum_r Z[r, 0:N]^T A[r, 0:K], all rows (primal and tangent).
 * Split over r into `splits` slabs in `partial` (>= splits*N*lddw floats, see _workspace_floats),
 * reduced in a fixed order (deterministic).  accumulate != 0 adds into dW.  lddw % 4 == 0, dW and partial 16-byte aligned
 * (the reduction moves float4); columns [K, lddw) of dW are written as 0. */
typedef struct {
  const float* Z; int64_t ldz;
  const float* A; int64_t lda;
  float* dW; int64_t lddw;
  float* partial;
  int32_t R, N, K, splits, accumulate;
  /* optional fused bias gradient: db[n] = sum over primal rows (r % group == 0) of Z[r][n];
   * db_partial: workspace of splits*N floats.  Both NULL to skip. */
  float* db; float* db_partial; int32_t group;
} sr_gemm_tn_args;
int64_t sr_mlp_gemm_tn_workspace_floats(int32_t R, int32_t N, int64_t lddw, int32_t* host_splits_out);
int sr_mlp_gemm_tn(const sr_gemm_tn_args* host_args, void* stream);
/* Up to SR_TN_GROUP_MAX independent weight gradients of the general tile shape (not the K <= 64 & N >= 256 first-layer shape) in ONE
 * launch + one launch for their slab reductions: the reverse sweep of a network on a few thousand rows (the ray branch, the
 * implicit-gradient pass: network.py:599-639, 702-814) leaves one weight gradient per layer, each too small to fill the machine.
 * Every problem is computed exactly as sr_mlp_gemm_tn would compute it alone (same tiles, same slabs, same reduction order: bit-identical);
 * no two problems of a call may share dW / db / partial. */
#define SR_TN_GROUP_MAX 12
typedef struct {
  int32_t n;
  sr_gemm_tn_args p[SR_TN_GROUP_MAX];
} sr_gemm_tn_group_args;
int sr_mlp_gemm_tn_group(const sr_gemm_tn_group_args* host_args, void* stream);
/* Bias gradient: out[n] += sum_{r % group == 0} Z[r][n]  (out must be zero-filled or hold the running sum). */
int sr_colsum_rows(const float* Z, int64_t ldz, int32_t R, int32_t N, int32_t group, float* out, void* stream);

/* Positional encoding (a1) fused with the first-layer input assembly: replaces the 13 elementwise
 * launches + torch.cat of model/Embedder.py:34-41 and the conds[batch_inds] concat of
 * model/Deformer.py:58-61.  out[p*group, :] = [x | w sin/cos bands | extra[extra_index[p] or p] | 0];
 * group == 4 also writes the three d/dx_t seed-tangent rows.  band_weights: 2*L floats (device). */
int sr_pe_embed(const float* x, int64_t P, int32_t L, const float* band_weights, const float* extra, int64_t ldextra,
                int32_t E, const int64_t* extra_index /*nullable*/, int32_t group, float* out, int64_t ldo, void* stream);
/* Reverse of sr_pe_embed w.r.t. x: gA0 = cotangent of the embed rows (same row layout, pitch ldg) -> xbar [P,3].
 * Covers the second-derivative term of the group-4 seed-tangent rows. */
int sr_pe_embed_bwd(const float* x, int64_t P, int32_t L, const float* band_weights, int32_t group, const float* gA0,
                    int64_t ldg, float* xbar, void* stream);

/* ---------------------------------------------------------------- fused LBS (a5, a8/a9 LBS part)
 * Replaces, for the no-autograd callers (ray refiner, inference), LBSkinner.forward's
 * GridSamplerMine.forward + per-frame blend loop (model/Deformer.py:207-233) and, when `jac` is
 * given, the three reverse passes of utils/utils.py:106-120 for the LBS factor of dd/dp.
 * vol: skinning weights, channel-last [D,H,W,24] fp32, 16-byte aligned.  A: posed joint transforms
 * (already times init_pose^-1) as [nframes,24,3,4] row-major.  Frame of point i = batch_inds[i], or
 * i / points_per_frame when batch_inds == NULL.  tp (nullable): points used for the weight lookup when
 * they differ from p (Deformer.py:168-171); jac requires tp == NULL. */
typedef struct {
  const float* p; const float* tp; int64_t P;
  const float* A; const float* trans; int32_t nframes;
  const int64_t* batch_inds; int64_t points_per_frame;
  const float* vol; int32_t D, H, W;
  float bmin[3], bmax[3];
  float* y;        /* [P,3] */
  float* jac;      /* [P,3,3] dy/dp, nullable */
} sr_lbs_args;
int sr_lbs_fwd(const sr_lbs_args* host_args, void* stream);
/* Reverse sweep of sr_lbs_fwd for a cotangent ybar [P,3] (replaces the autograd backward of the K3/K4 sampler +
 * per-frame blend, model/Deformer.py:207-233): pbar = (dy/dp)^T ybar, Abar [nframes,24,12] = sum w_j ybar (x) [p;1],
 * transbar [nframes,3] = sum ybar.  Abar / transbar are WRITTEN (no zero fill); each output is nullable.  The per-frame sums are
 * deterministic (segmented wave reductions, per-workgroup partials in `partials`, folded in double precision in a fixed order):
 * bit-reproducible run to run.  `partials`: sr_lbs_bwd_workspace_floats(P, nframes) floats of scratch. */
int64_t sr_lbs_bwd_workspace_floats(int64_t P, int32_t nframes);
int sr_lbs_bwd(const sr_lbs_args* host_args, const float* ybar, float* pbar, float* Abar, float* transbar, float* partials, void* stream);
/* Reverse sweep of sr_lbs_fwd WITH its Jacobian output: cotangents ybar [P,3] (nullable) and Jbar [P,3,3] of (y, jac) ->
 * pbar (includes the mixed second derivatives of the trilinear sampler), Abar, transbar (written as in sr_lbs_bwd; same scratch).
 * Backward of the value + Jacobian form of the deformer that replaces compute_Jacobian's three reverse passes with
 * create_graph (utils/utils.py:106-120) in the colour / normal branch and in propagateTmpPsGrad. */
int sr_lbs_jac_bwd(const sr_lbs_args* host_args, const float* ybar, const float* Jbar, float* pbar, float* Abar, float* transbar, float* partials,
                   void* stream);

/* SMPL kinematic chain of LBSkinner.forward / posedSkeleton (model/Deformer.py:144-203, smpl_pytorch/util.py:35-78):
 * poses [B,24,3] axis-angle (device) -> G [B,24,4,4] posed chain and A = G * init_pose [B,24,4,4].
 * Js [24,3], parents [24] (parents[i] < i), init_pose [24,4,4] are HOST arrays (passed by value to the kernel).
 * _bwd: posebar [B,24,3] from the cotangents Abar and/or Gbar (nullable). */
int sr_lbs_chain_fwd(const float* poses, int32_t B, const float* host_Js, const int32_t* host_parents, const float* host_init_pose,
                     float* G, float* A, void* stream);
int sr_lbs_chain_bwd(const float* poses, int32_t B, const float* host_Js, const int32_t* host_parents, const float* host_init_pose,
                     const float* Abar, const float* Gbar, float* posebar, void* stream);

/* ---------------------------------------------------------------- ray/surface refiner step (a12)
 * One iteration body of utils/FindSurfacePs.py::OptimizeSurfacePs (:115-126 check, :135-151 step)
 * for M live rays.  sdf4: output rows of the sdf-only SDF MLP, `group` rows per ray (row 0 = f,
 * rows 1..3 = df/dp_t), pitch ld_sdf.  off4: output rows of the deformation MLP (row 0 = offset,
 * rows 1..3 = d offset / d p_t), pitch ld_off.  y / jlbs: LBS output d(p) and dLBS/dq from sr_lbs_fwd.
 * converged[i] = |f| < dthreshold && asin(|(d-c) x v| / |d-c|) * 180/pi < athreshold.
 * group == 4 and p_out != NULL: p_out = converged ? p : p - L g / |g|^2.  group == 1: check only. */
typedef struct {
  int64_t M; int32_t group;
  const float* sdf4; int64_t ld_sdf;
  const float* off4; int64_t ld_off;
  const float* y; const float* jlbs;
  const float* rays; const float* cam;   /* [M,3], [3] (device) */
  const float* p; float* p_out;          /* [M,3] */
  uint8_t* converged;                    /* [M] */
  float dthreshold, athreshold, w1, w2;
} sr_newton_args;
int sr_newton_update(const sr_newton_args* host_args, void* stream);

/* Reverse-mode form of the same refiner step (utils/FindSurfacePs.py:135-151): `sr_newton_prepare` tests convergence and
 * emits t = J_lbs^T (d sin-angle / d y) as the cotangent rows of the deformation offset ([M, ld_t], columns >= 3 zeroed;
 * t_out == NULL: convergence test only); after the two reverse sweeps `sr_newton_apply` forms
 * g = w1 sign(f) grad_f + w2 (t + grad_off) and steps the unconverged points. */
typedef struct {
  int64_t M;
  const float* sdf; int64_t ld_sdf;      /* f = sdf[i*ld_sdf] */
  const float* y; const float* jlbs;     /* LBS output [M,3] and dLBS/dq [M,3,3] */
  const float* rays; const float* cam;
  uint8_t* converged;                    /* [M] written by prepare, read by apply */
  float* t_out; int64_t ld_t; float* s_out;
  const float* grad_f; const float* grad_off;   /* [M,3] each (apply) */
  const float* p; float* p_out;          /* (apply) */
  float dthreshold, athreshold, w1, w2;
} sr_newton2_args;
int sr_newton_prepare(const sr_newton2_args* host_args, void* stream);
int sr_newton_apply(const sr_newton2_args* host_args, void* stream);

/* ---------------------------------------------------------------- device-driven refiner (a12; the `sr_trace_newton` of SURVEY 8(b))
 * The whole of utils/FindSurfacePs.py::OptimizeSurfacePs (:114-163) for P rays as a fixed sequence of launches whose row
 * count lives in device memory: `live[k]` = unfinished rays entering phase k (live[0] = P).  The queue of unfinished rays
 * (x, v, frame, orig: two copies, phase parity selects) is compacted by wave ballots at the end of every phase, so finished
 * rays leave the layer GEMMs immediately and the host never learns a count.  After sr_refine_init the caller issues per phase
 *   sr_mlp_chain(forward, m_dev = live + k) -> sr_refine_mid
 *   [-> sr_mlp_chain(reverse) -> sr_refine_finish   for the update phases 1..times]
 * The kernels that put rays into a queue (_init: phase 0; _mid mode 0: phase 1; _finish of phase k: phase k + 1) also write those
 * rays' first-layer input rows a0 / a0d when both pointers are given; sr_refine_embed(phase) writes the same rows for the queue
 * of `phase` as a launch of its own (same arithmetic, same bits; needed only by a caller that passes a0 = a0d = NULL to the others).
 * phase 0 (mid mode 0): initial test; phases 1..times (mode 1): test of the current points + Newton update of the failing
 * ones; phase times+1 (mode 2): test of the last update.  p_out / conv_out are indexed by the rays' original position.
 * The layer chains read a0 / a0d (first-layer inputs) and leave f in sdf_out[:,0], the deformation
 * offset in def_out[:,0:3]; the reverse chains start from `unit` rows (1,0,0,0) (SDF) and `t` rows (deformer) and leave
 * the input cotangents in a0bar (+ skipbar: the skip-concat part, n_skip columns) and a0dbar. */
typedef struct {
  int32_t P, times;
  const float* p0; const float* rays; const int64_t* batch_inds;   /* [P,3], [P,3], [P] inputs (read by _init) */
  const float* cam;                                                /* [3] camera centre (device) */
  int32_t L_sdf; const float* w_sdf; int32_t L_def; const float* w_def;   /* PE bands + per-band weights (2L floats each) */
  const float* conds; int64_t ld_conds; int32_t E;                 /* per-frame deformation codes [nframes, E] */
  const float* A; const float* trans; int32_t nframes;             /* posed transforms [nframes,24,12], translations [nframes,3] */
  const float* vol; int32_t D, H, W; float bmin[3], bmax[3];       /* channel-last skinning-weight volume + its box */
  float dthreshold, athreshold, w1, w2;
  int32_t* live;                                                   /* [times + 3] */
  float* x[2]; float* v[2]; int32_t* frame[2]; int32_t* orig[2];   /* the queue: [P,3], [P,3], [P], [P] each */
  float* unit;                                                     /* [P,4] rows (1,0,0,0), written by _init */
  float* a0; int64_t ld_a0; float* a0d; int64_t ld_a0d;
  const float* sdf_out; int64_t ld_sdf; const float* def_out; int64_t ld_def;
  uint8_t* conv; float* t; float* s;                               /* [P], [P,4], [P] */
  const float* a0bar; int64_t ld_a0bar; const float* skipbar; int64_t ld_skipbar; int32_t n_skip;
  const float* a0dbar; int64_t ld_a0dbar;
  float* p_out; uint8_t* conv_out;                                 /* [P,3], [P] */
} sr_refine_args;
int sr_refine_init(const sr_refine_args* host_args, void* stream);
int sr_refine_embed(const sr_refine_args* host_args, int32_t phase, void* stream);
int sr_refine_mid(const sr_refine_args* host_args, int32_t phase, int32_t mode, void* stream);
int sr_refine_finish(const sr_refine_args* host_args, int32_t phase, void* stream);


/* ---------------------------------------------------------------- MCGpu (a17)
 * Replaces MCGpu/MCGpu.cpp:20-56 mc_gpu -> MCGpu::init/MC/scaleVertices (CudaKernels.cu:524-639) and
 * kernels K6-K9.  Two calls because the output sizes are data dependent (the reference also copies
 * its two counters to the host between K6 and K7, CudaKernels.cu:628):
 *   sr_mc_count : classify + exclusive scans into `workspace` (sr_mc_workspace_bytes);
 *                 counts_dev[0] = #vertices, counts_dev[1] = #faces (2 x uint32, device memory)
 *   sr_mc_emit  : verts [V,3] f32 = lattice position * step + min, faces [F,3] int64 (reversed winding;
 *                 -1 where the owning cell lies outside the grid, as the reference).
 * sdf: [nx,ny,nz] fp32 dense, index i*ny*nz + j*nz + k.  Output ORDER is deterministic: vertices by
 * lattice-edge key (cell*3+dir), faces by (cell, triangle) -- the reference's atomicAdd order is not
 * reproducible even by itself; compare after canonicalisation.  No growing singleton, no
 * 5%-of-cells scratch guess (CudaKernels.cu:590-592): the caller sizes the outputs exactly.  `workspace` must be 8-byte
 * aligned (it holds 64-bit classification planes); verts / faces double as scratch inside sr_mc_emit before they are written. */
int64_t sr_mc_workspace_bytes(int32_t nx, int32_t ny, int32_t nz);
int sr_mc_count(const float* sdf, int32_t nx, int32_t ny, int32_t nz, float iso, void* workspace, uint32_t* counts_dev, void* stream);
int sr_mc_emit(const float* sdf, int32_t nx, int32_t ny, int32_t nz, float iso, const void* workspace, float xstep, float ystep,
               float zstep, float xmin, float ymin, float zmin, float* verts, int64_t* faces, void* stream);

/* ---------------------------------------------------------------- small per-element ops
 * sr_svd3x3: batched 3x3 SVD on device, A = U diag(S) V^T, S descending -- replaces
 *   `torch.svd(Jacobs.cpu())` of the deformation regulariser (model/network.py:576).  A,U,V [n,3,3], S [n,3].
 */
int sr_svd3x3(const float* A, int64_t n, float* U, float* S, float* V, void* stream);

/* Effective weights of up to SR_PACK_MAX_LAYERS linear layers in one launch (weight_norm dim 0 as model/network.py:65-66 and
 * model/RenderNet.py:46-47: W = g v/|v|; g NULL = plain layer): W [N, ldw] zero padded, WT [K, ldwt] = W^T zero padded,
 * norms [N] = |v_n|.  _unpack is the backward: dW [N, lddw] -> gv [N, K] (and gg [N] for weight-normed layers), i.e.
 * aten::_weight_norm_interface_backward; accumulate != 0 adds to gv / gg. */
/* dst [rows * group, ldd]: row r * group + g = (g == 0 ? a : b)[r, :n] zero padded to `width` columns; group 1 (a only) or 2 (the
 * (primal, tangent) row interleave of the group-2 GEMMs); a NULL source gives zero rows. */
int sr_rows_pad(const float* a, int64_t lda, int32_t na, const float* b, int64_t ldb, int32_t nb, int64_t rows, int32_t group, float* dst, int64_t ldd,
                int32_t width, void* stream);
/* out [n,E] = per-frame sums of the rows of X [P, ldx >= E]: out[f] = sum_{index[r] == f} X[r, :E]  (n <= 32) -- the backward of the
 * per-frame code gather conds[batch_inds] (model/Deformer.py:61,75), deterministic (fixed-order folds; torch's index_add uses float
 * atomics).  `partial`: sr_rows_frame_sum_workspace_floats(P, E, n) floats of scratch. */
int64_t sr_rows_frame_sum_workspace_floats(int64_t P, int32_t E, int32_t n);
int sr_rows_frame_sum(const float* X, int64_t ldx, int64_t P, int32_t E, const int64_t* index, int32_t n, float* partial, float* out, void* stream);

#define SR_PACK_MAX_LAYERS 16
typedef struct { const float* v; const float* g; float* W; float* WT; float* norms; int32_t N, K; int64_t ldw, ldwt; } sr_pack_layer;
typedef struct { int32_t nlayers; sr_pack_layer layer[SR_PACK_MAX_LAYERS]; } sr_pack_table;
typedef struct { const float* dW; int64_t lddw; const float* v; const float* g; const float* norms; float* gv; float* gg;
                 const float* db; float* gb;   /* optional: bias gradient gb[n] += db[n] (n < N) in the same launch */
                 int32_t N, K, accumulate, pad_; } sr_unpack_layer;
typedef struct { int32_t nlayers; sr_unpack_layer layer[SR_PACK_MAX_LAYERS]; } sr_unpack_table;
int sr_pack_weights(const sr_pack_table* host_table, void* stream);
int sr_unpack_grads(const sr_unpack_table* host_table, void* stream);

/* Adam step (torch.optim.Adam, weight_decay 0, amsgrad off -- the optimizer of train.py:139) of up to SR_ADAM_MAX_TENSORS contiguous
 * float32 tensors in one launch.  Per tensor: p, g (gradient), m (exp_avg), v (exp_avg_sq), numel, lr, bias1 = 1 - beta1^step,
 * inv_sqrt_bias2 = 1 / sqrt(1 - beta2^step) (host-side scalars of the tensor's step count). */
#define SR_ADAM_MAX_TENSORS 64
typedef struct { float* p; const float* g; float* m; float* v; int64_t numel; float lr, bias1, inv_sqrt_bias2, pad_; } sr_adam_tensor;
typedef struct { int32_t ntensors; float beta1, beta2, eps;
                 float one_minus_beta1, one_minus_beta2;   /* formed in double on the host: 1 - 0.999f is off by 1.3e-5 relative in float */
                 int32_t pad_[2]; sr_adam_tensor tensor[SR_ADAM_MAX_TENSORS]; } sr_adam_table;
int sr_adam_step(const sr_adam_table* host_table, void* stream);

/* Stream diagnostics (no reference counterpart: the reference runs one stream).
 * sr_stream_flag_set stores `value` to *flag when its stream reaches it; sr_stream_flag_wait holds its stream (one sleeping wave)
 * until *flag - value >= 0 in wrapping 32-bit arithmetic or until timeout_ms has passed, in which case *timed_out (optional) is
 * incremented and the stream proceeds.  An ordering of two streams that does not go through the runtime's events: used by
 * tools/stream_latency2.py to tell a property of hipStreamWaitEvent from a property of the schedule (the 10+ ms the ray selection
 * seemed to wait for the main stream turned out to be the host running a whole iteration ahead of the GPU, with either primitive). */
/* Diagnostics: writes the device's constant 100 MHz counter to *out when the stream reaches this point (timestamps that are
 * comparable across streams, which the runtime's event timestamps are not: tools/host_profile.py). */
int sr_stream_stamp(uint64_t* out, void* stream);
int sr_stream_flag_set(uint32_t* flag, uint32_t value, void* stream);
int sr_stream_flag_wait(const uint32_t* flag, uint32_t value, uint32_t* timed_out, int32_t timeout_ms, void* stream);

/* ---------------------------------------------------------------- interp2x_boundary3d (K10/K11)
 * Replaces MCAcc/cuda/interp2x_boundary3d.cpp:forward/backward -> interp2x_boundary3d_kernel.cu:11-151, 155-239
 * (compiled but never enabled in the reference: every Seg3dLossless is built with use_cuda_impl=False).
 * in [BC, d,h,w] -> out [BC, 2d-1,2h-1,2w-1] = mean of the 1/2/4/8 coarse parents; is_boundary (u8) = parents'
 * (v > balance) flags disagree.  _bwd: grad_out [BC, 2d-1,2h-1,2w-1] -> grad_in [BC, d,h,w] (adjoint stencil). */
int sr_interp2x3d_fwd_f32(const float* in, int64_t BC, int32_t d, int32_t h, int32_t w, float balance, float* out, uint8_t* is_boundary, void* stream);
int sr_interp2x3d_fwd_f64(const double* in, int64_t BC, int32_t d, int32_t h, int32_t w, float balance, double* out, uint8_t* is_boundary, void* stream);
int sr_interp2x3d_bwd_f32(const float* grad_out, int64_t BC, int32_t d, int32_t h, int32_t w, float* grad_in, void* stream);
int sr_interp2x3d_bwd_f64(const double* grad_out, int64_t BC, int32_t d, int32_t h, int32_t w, double* grad_in, void* stream);
/* Candidate voxels of one Seg3dLossless level (MCAcc/seg3d_lossless.py:296-312: smooth_conv3x3(is_boundary) > 0, minus the
 * voxels evaluated at earlier levels, nonzero) in one pass: out_index receives the flat indices z*H*W + y*W + x (capacity
 * D*H*W entries, unordered), *count_dev their number (uint64, device).  is_boundary / done: [D,H,W] bytes (0 / non-0). */
int sr_seg3d_candidates(const uint8_t* is_boundary, const uint8_t* done, int32_t D, int32_t H, int32_t W, int64_t* out_index, uint64_t* count_dev,
                        void* stream);

/* ---------------------------------------------------------------- rasterisation either side of the refiner
 * The reference calls pytorch3d 0.4.0 here (third-party CUDA, not in its repository; restated in oracle/raster_oracle.py,
 * parity unpinned).  Both entry points work in pytorch3d's NDC frame: +x left, +y up, pixel (row, col) centred at
 * (1 - (2 col + 1)/W, 1 - (2 row + 1)/H); xy_ndc [nimg, V, 2], z [nimg, V] = view-space depth.
 *
 * sr_points_silhouette_*: replaces PointsRasterizer(radius, points_per_pixel = K) + AlphaCompositor with one all-ones
 *   feature as called at model/network.py:178-190,495-497 through PointsRendererWithFrags (model/CameraMine.py:285-305):
 *   mask[img,row,col] = sum_k a_k prod_{j<k}(1 - a_j) over the K covering points nearest in z, a = 1 - dist2/radius^2
 *   (dist2 < radius^2, NDC units; z < 0 skipped).  _bwd returns d(sum gmask * mask)/d xy_ndc (what pytorch3d's
 *   rasterize_points backward propagates through `dists`).  `workspace` (256-byte aligned, _workspace_bytes) carries the
 *   per-pixel transmittance and selection thresholds from _fwd to _bwd.
 * sr_rasterize_meshes: replaces MeshRasterizer(blur_radius 0, faces_per_pixel 1, perspective_correct, no barycentric
 *   clipping, no culling) of model/network.py:877-892,492: faces [F,3] int64 shared by all images (rows containing -1 are
 *   skipped: MCGpu border faces) -> pix_to_face [nimg,H,W] int64 (packed img*F + f, -1 = background), bary [nimg,H,W,3]
 *   (-1 on background), zout [nimg,H,W] nullable.  zbuf_u64: scratch of nimg*H*W 8-byte words; pix_to_face and the
 *   (8-byte aligned) head of bary hold the large-face queue until the last pass overwrites them. */
int64_t sr_points_silhouette_workspace_bytes(int64_t nimg, int64_t pts_per_img, int32_t H, int32_t W, float radius);
int sr_points_silhouette_fwd(const float* xy_ndc, const float* z, int64_t nimg, int64_t pts_per_img, int32_t H, int32_t W, float radius,
                             int32_t K, float* mask, void* workspace, void* stream);
int sr_points_silhouette_bwd(const float* xy_ndc, const float* z, int64_t nimg, int64_t pts_per_img, int32_t H, int32_t W, float radius,
                             const void* workspace, const float* gmask, float* gxy, void* stream);
int sr_rasterize_meshes(const float* xy_ndc, const float* z, const int64_t* faces, int64_t nimg, int64_t V, int64_t F, int32_t H, int32_t W,
                        void* zbuf_u64, int64_t* pix_to_face, float* bary, float* zout, void* stream);

/* ---------------------------------------------------------------- fused per-ray / per-vertex tails of one step (csrc/step_ops.hip)
 * Each pair replaces a block of elementwise torch ops of the reference's training step (and their autograd mirror) with one
 * launch for the value and one for the gradient.  Reductions are deterministic (fixed summation order, no float atomics).
 *
 * Camera (model/CameraMine.py:44-70,129-170,171-262).  All pointers are DEVICE pointers (the intrinsics are learnable
 * parameters, config.conf:10-15): R [3,3] row-major with p_cam = p R + T, T [3] (NULL = 0), f [2], c [2]; W, H in pixels;
 * one_minus_inv_w/h = (float)(1 - 1/W), (float)(1 - 1/H) computed on the host as the reference's Python scalars are.
 *   sr_cam_project_ndc_fwd: ps [n,3] -> xy [n,2] in pytorch3d's NDC frame (x = f0/(W/2) X/Z + 1 - 1/W - c0/(W/2)), z [n] = view depth.
 *   sr_cam_project_ndc_bwd: gxy / gz (either may be NULL) -> gps [n,3] (NULL: skipped) and, when `partial`
 *     ([sr_step_param_blocks(n), 16] scratch) is given, gparams [16] = gR[9] | gT[3] | gf[2] | gc[2].
 *   sr_cam_view_rays_fwd: pixels [n,3] = (col, row, 1) -> unit world rays normalize([(c0 h - u)/f0, (c1 h - w)/f1, h]) R^T.
 *   sr_cam_view_rays_bwd: grays -> gparams [16] in the same layout (gT slots untouched = 0). */
#define SR_STEP_MAX_FRAMES 8
typedef struct { const float* R; const float* T; const float* f; const float* c; float W, H, one_minus_inv_w, one_minus_inv_h; } sr_camera;
int sr_step_reduce_blocks(int64_t rows);      /* workgroups (= rows of `partial`, SR_STEP_LOSS_SLOTS - 1 floats each) a loss reduction over `rows` uses */
int sr_step_param_blocks(int64_t rows);       /* rows of `partial` (16 floats each) of the camera backward kernels */
int sr_cam_project_ndc_fwd(const float* ps, int64_t n, const sr_camera* cam, float* xy, float* z, void* stream);
int sr_cam_project_ndc_bwd(const float* ps, int64_t n, const sr_camera* cam, const float* gxy, const float* gz, float* gps, float* partial,
                           float* gparams, void* stream);
int sr_cam_view_rays_fwd(const float* pixels, int64_t n, const sr_camera* cam, float* rays, void* stream);
int sr_cam_view_rays_bwd(const float* pixels, int64_t n, const sr_camera* cam, const float* grays, float* partial, float* gparams, void* stream);

/* Cardinal rays and deformed normals (utils/utils.py:132-169) from the deformation Jacobian J [n,3,3] (J[i][j] = d d_i / d p_j):
 *   sr_cardinal_rays_fwd: out = normalize(J^-1 v); rows whose |det J| < 1e-4 (FastMinv's rule) keep normalize(v) and ok = 0.
 *   sr_cardinal_rays_bwd: cotangent of out -> gJ [n,9] = -(J^-T gu)(J^-1 v)^T and gv [n,3] = J^-T gu (either may be NULL; zero on ok = 0 rows,
 *     where the reference substitutes rays.detach()).
 *   sr_deformed_normals: out = normalize(J^-T n) (J n on singular rows); no gradient ('test' phase of compute_deformed_normals). */
int sr_cardinal_rays_fwd(const float* J, const float* v, int64_t n, float* out, uint8_t* ok, void* stream);
int sr_cardinal_rays_bwd(const float* J, const float* v, int64_t n, const float* gout, float* gJ, float* gv, void* stream);
int sr_deformed_normals(const float* J, const float* onx, int64_t n, float* out, void* stream);

/* Loss reductions.  out [SR_STEP_LOSS_SLOTS]: out[0] = the loss, out[1 + f] / out[1 + SR_STEP_MAX_FRAMES + f] = the per-frame
 * numerator / denominator sums the backward reads back as `saved`.  partial: [sr_step_reduce_blocks(rows), SR_STEP_LOSS_SLOTS - 1]
 * scratch, may be NULL when that is 1.  gloss: device scalar (cotangent of the loss).
 *   colour (model/network.py:611-618): rays (b, r, c) [P] into gt [N,H,W,3]; mean over frames of the per-frame mean of |gt - colour|_1.
 *     A row with b[i] < 0 is MASKED in the colour and the normal reduction: no term, no count, exact-zero gradients (the reference
 *     drops the rays its refiner rejected by a boolean gather, network.py:599-606 -- a host round trip; the mask keeps the shape static).
 *   normal (model/network.py:620-639): gt normal image [N,H,W,3] -> flip diag(-1,1,-1) -> world (R [3,3]) -> unit (valid when
 *     |.| > 1e-4) -> canonical (J^T), against normalize(nx_raw); weighted != 0 multiplies by clamp(-rays . n_def, 0, 1)^2 with
 *     n_def = sr_deformed_normals(J, nx_raw) (detached); masked scatter-mean over frames.  _bwd: gnx_raw [P,3], gJ [P,9] (NULL: skipped).
 *   eikonal (model/network.py:547-549): mean (|g| - 1)^2 over g [n,3].
 *   def_regu (model/network.py:565-582 + utils/utils.py:48-52): S [n,3] singular values -> mean GM(sum_k log(s_k)^2, c); _bwd goes
 *     straight to gJ [n,9] = U diag(gS) V^T with the U, V of sr_svd3x3.   sr_svd3x3_bwd: the plain gS -> gJ.
 *   mask IoU (model/network.py:652-654): masks, gt [N, hw]; mean over frames of 1 - sum(m g) / sum |m + g - m g|. */
#define SR_STEP_LOSS_SLOTS (1 + 2 * SR_STEP_MAX_FRAMES)
typedef struct { const int64_t* b; const int64_t* r; const int64_t* c; int64_t P; int32_t N, H, W; } sr_ray_pixels;
int sr_color_loss_fwd(const sr_ray_pixels* px, const float* colors, const float* gt, float* partial, float* out, void* stream);
int sr_color_loss_bwd(const sr_ray_pixels* px, const float* colors, const float* gt, const float* saved, const float* gloss, float* gcolors,
                      void* stream);
int sr_normal_loss_fwd(const sr_ray_pixels* px, const float* nx_raw, const float* J, const float* gt_normals, const float* R, const float* rays,
                       int weighted, float* partial, float* out, void* stream);
int sr_normal_loss_bwd(const sr_ray_pixels* px, const float* nx_raw, const float* J, const float* gt_normals, const float* R, const float* rays,
                       int weighted, const float* saved, const float* gloss, float* gnx_raw, float* gJ, void* stream);
int sr_eikonal_loss_fwd(const float* g, int64_t n, float* partial, float* out, void* stream);
int sr_eikonal_loss_bwd(const float* g, int64_t n, const float* gloss, float* gg, void* stream);
int sr_def_regu_loss_fwd(const float* S, int64_t n, float c, float* partial, float* out, void* stream);
int sr_def_regu_loss_bwd(const float* U, const float* S, const float* V, int64_t n, float c, const float* gloss, float* gJ, void* stream);
int sr_svd3x3_bwd(const float* U, const float* V, const float* gS, int64_t n, float* gJ, void* stream);
int sr_mask_iou_loss_fwd(const float* masks, const float* gt, int N, int64_t hw, float* partial, float* out, void* stream);
int sr_mask_iou_loss_bwd(const float* masks, const float* gt, int N, int64_t hw, const float* saved, const float* gloss, float* gmasks, void* stream);

/* Normal equations of the implicit differentiation (model/network.py:702-771): b = [grad_f ; [v]x J] (4x3),
 * rhs = grad_l^T (b^T b)^-1 b^T with FastMinv's singularity rule (ok = 0, zeros) -> cot_f [n] = -rhs[0], rhs_tail [n,3] = rhs[1:4],
 * temp [n,3] = rhs[1:4] (-[v]x). */
int sr_implicit_solve(const float* grad_f, const float* J, const float* v, const float* grad_l, int64_t n, float* cot_f, float* rhs_tail, float* temp,
                      uint8_t* ok, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SELFRECON_HIP_H_ */
