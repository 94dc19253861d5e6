"""ctypes binding of libselfrecon_hip.so -- the only door from Python to the HIP kernels.

There is NO fallback: if the shared library is missing or lacks a symbol the import of any
operator module fails loudly (a product path that silently ran on CPU/eager PyTorch would
void every parity and performance claim).
"""
import ctypes
import os
import torch

from .build import LIB

_i64, _vp, _int = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int


class SrTensor5(ctypes.Structure):
    _fields_ = [("size", _i64 * 5), ("stride", _i64 * 5)]


class SrGemmArgs(ctypes.Structure):
    _fields_ = [("A", _vp), ("lda", _i64), ("B", _vp), ("ldb", _i64), ("C", _vp), ("ldc", _i64),
                ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("bias", _vp),
                ("group", ctypes.c_int32), ("act", ctypes.c_int32), ("mode", ctypes.c_int32), ("out_scale", ctypes.c_float),
                ("aux", _vp), ("ldaux", _i64), ("naux_fwd", ctypes.c_int32), ("nact_bwd", ctypes.c_int32),
                ("aux_scale", ctypes.c_float)]


class SrGemmTnArgs(ctypes.Structure):
    _fields_ = [("Z", _vp), ("ldz", _i64), ("A", _vp), ("lda", _i64), ("dW", _vp), ("lddw", _i64), ("partial", _vp),
                ("R", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("splits", ctypes.c_int32),
                ("accumulate", ctypes.c_int32), ("db", _vp), ("db_partial", _vp), ("group", ctypes.c_int32)]


SR_TN_GROUP_MAX = 12


class SrGemmTnGroupArgs(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int32), ("p", SrGemmTnArgs * SR_TN_GROUP_MAX)]


class SrLbsArgs(ctypes.Structure):
    _fields_ = [("p", _vp), ("tp", _vp), ("P", _i64), ("A", _vp), ("trans", _vp), ("nframes", ctypes.c_int32),
                ("batch_inds", _vp), ("points_per_frame", _i64), ("vol", _vp), ("D", ctypes.c_int32), ("H", ctypes.c_int32),
                ("W", ctypes.c_int32), ("bmin", ctypes.c_float * 3), ("bmax", ctypes.c_float * 3), ("y", _vp), ("jac", _vp)]


class SrNewtonArgs(ctypes.Structure):
    _fields_ = [("M", _i64), ("group", ctypes.c_int32), ("sdf4", _vp), ("ld_sdf", _i64), ("off4", _vp), ("ld_off", _i64),
                ("y", _vp), ("jlbs", _vp), ("rays", _vp), ("cam", _vp), ("p", _vp), ("p_out", _vp), ("converged", _vp),
                ("dthreshold", ctypes.c_float), ("athreshold", ctypes.c_float), ("w1", ctypes.c_float), ("w2", ctypes.c_float)]


class SrNewton2Args(ctypes.Structure):
    _fields_ = [("M", _i64), ("sdf", _vp), ("ld_sdf", _i64), ("y", _vp), ("jlbs", _vp), ("rays", _vp), ("cam", _vp), ("converged", _vp),
                ("t_out", _vp), ("ld_t", _i64), ("s_out", _vp), ("grad_f", _vp), ("grad_off", _vp), ("p", _vp), ("p_out", _vp),
                ("dthreshold", ctypes.c_float), ("athreshold", ctypes.c_float), ("w1", ctypes.c_float), ("w2", ctypes.c_float)]


SR_CHAIN_MAX_LAYERS = 10


class SrChainArgs(ctypes.Structure):
    _fields_ = [("nlayers", ctypes.c_int32), ("nprob", ctypes.c_int32 * SR_CHAIN_MAX_LAYERS), ("g", (SrGemmArgs * 2) * SR_CHAIN_MAX_LAYERS),
                ("m_dev", _vp), ("m_mul", ctypes.c_int32), ("m_cap", ctypes.c_int32)]


class SrRefineArgs(ctypes.Structure):
    _f, _i32 = ctypes.c_float, ctypes.c_int32
    _fields_ = [("P", _i32), ("times", _i32), ("p0", _vp), ("rays", _vp), ("batch_inds", _vp), ("cam", _vp),
                ("L_sdf", _i32), ("w_sdf", _vp), ("L_def", _i32), ("w_def", _vp), ("conds", _vp), ("ld_conds", _i64), ("E", _i32),
                ("A", _vp), ("trans", _vp), ("nframes", _i32), ("vol", _vp), ("D", _i32), ("H", _i32), ("W", _i32),
                ("bmin", _f * 3), ("bmax", _f * 3), ("dthreshold", _f), ("athreshold", _f), ("w1", _f), ("w2", _f),
                ("live", _vp), ("x", _vp * 2), ("v", _vp * 2), ("frame", _vp * 2), ("orig", _vp * 2), ("unit", _vp),
                ("a0", _vp), ("ld_a0", _i64), ("a0d", _vp), ("ld_a0d", _i64), ("sdf_out", _vp), ("ld_sdf", _i64), ("def_out", _vp), ("ld_def", _i64),
                ("conv", _vp), ("t", _vp), ("s", _vp), ("a0bar", _vp), ("ld_a0bar", _i64), ("skipbar", _vp), ("ld_skipbar", _i64), ("n_skip", _i32),
                ("a0dbar", _vp), ("ld_a0dbar", _i64), ("p_out", _vp), ("conv_out", _vp)]


SR_PACK_MAX_LAYERS = 16


class SrPackLayer(ctypes.Structure):
    _fields_ = [("v", _vp), ("g", _vp), ("W", _vp), ("WT", _vp), ("norms", _vp), ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("ldw", _i64), ("ldwt", _i64)]


class SrPackTable(ctypes.Structure):
    _fields_ = [("nlayers", ctypes.c_int32), ("layer", SrPackLayer * SR_PACK_MAX_LAYERS)]


class SrUnpackLayer(ctypes.Structure):
    _fields_ = [("dW", _vp), ("lddw", _i64), ("v", _vp), ("g", _vp), ("norms", _vp), ("gv", _vp), ("gg", _vp), ("db", _vp), ("gb", _vp),
                ("N", ctypes.c_int32), ("K", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("pad_", ctypes.c_int32)]


SR_ADAM_MAX_TENSORS = 64


class SrAdamTensor(ctypes.Structure):
    _fields_ = [("p", _vp), ("g", _vp), ("m", _vp), ("v", _vp), ("numel", _i64), ("lr", ctypes.c_float), ("bias1", ctypes.c_float),
                ("inv_sqrt_bias2", ctypes.c_float), ("pad_", ctypes.c_float)]


class SrAdamTable(ctypes.Structure):
    _fields_ = [("ntensors", ctypes.c_int32), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("one_minus_beta1", ctypes.c_float), ("one_minus_beta2", ctypes.c_float), ("pad_", ctypes.c_int32 * 2),
                ("tensor", SrAdamTensor * SR_ADAM_MAX_TENSORS)]


class SrUnpackTable(ctypes.Structure):
    _fields_ = [("nlayers", ctypes.c_int32), ("layer", SrUnpackLayer * SR_PACK_MAX_LAYERS)]


SR_STEP_MAX_FRAMES = 8
SR_STEP_LOSS_SLOTS = 1 + 2 * SR_STEP_MAX_FRAMES


class SrCamera(ctypes.Structure):
    _fields_ = [("R", _vp), ("T", _vp), ("f", _vp), ("c", _vp), ("W", ctypes.c_float), ("H", ctypes.c_float),
                ("one_minus_inv_w", ctypes.c_float), ("one_minus_inv_h", ctypes.c_float)]


class SrRayPixels(ctypes.Structure):
    _fields_ = [("b", _vp), ("r", _vp), ("c", _vp), ("P", _i64), ("N", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32)]


class SrError(RuntimeError):
    pass


_CODES = {-1: "SR_EINVAL (bad argument)", -2: "SR_ELAUNCH (kernel launch failed)", -3: "SR_ENOSPC (workspace too small)"}

if "SELFRECON_HIP_LIB" in os.environ:
    LIB = os.environ["SELFRECON_HIP_LIB"]         # tuning hook: point at an experimental build of the same ABI
else:
    from .build import is_stale, build_lib
    if is_stale():                                # sources changed since the .so was linked (or no .so yet): kernels and the ctypes
        try:                                      # structs below must come from the same tree -- rebuild (one process builds under an
            build_lib(verbose=False)              # flock, the others wait; atomic publish) or fail loudly
        except RuntimeError as e:
            raise ImportError(str(e)) from e
if not os.path.isfile(LIB):
    raise ImportError(f"{LIB} not found: run `python -m selfreconcode_amd.build` (hipcc --offload-arch=gfx950). "
                      "There is no CPU fallback for the HIP hot path.")
_lib = ctypes.CDLL(LIB)

# name -> argtypes; mirrors include/selfrecon_hip.h one to one (tests/test_abi.py checks the set)
_T5 = SrTensor5
SIGNATURES = {
    "sr_minv3x3_fwd_f32": [_vp, _vp, _vp, _i64, _vp],
    "sr_minv3x3_fwd_f64": [_vp, _vp, _vp, _i64, _vp],
    "sr_minv3x3_bwd_f32": [_vp, _vp, _vp, _i64, _vp],
    "sr_minv3x3_bwd_f64": [_vp, _vp, _vp, _i64, _vp],
    "sr_gridsample3d_fwd_f16": [_vp, _T5, _vp, _T5, _vp, _T5, _vp],
    "sr_gridsample3d_bwd_f16": [_vp, _T5, _vp, _T5, _vp, _T5, _vp, _T5, _vp, _vp],
    "sr_gridsample3d_dbwd_f16": [_vp, _T5, _vp, _T5, _vp, _T5, _vp, _T5, _vp, _T5, _vp, _T5, _vp, _vp, _vp],
    "sr_gridsample3d_fwd_f32": [_vp, _T5, _vp, _T5, _vp, _T5, _vp],
    "sr_gridsample3d_fwd_f64": [_vp, _T5, _vp, _T5, _vp, _T5, _vp],
    "sr_gridsample3d_bwd_f32": [_vp, _T5, _vp, _T5, _vp, _T5, _vp, _T5, _vp, _vp],
    "sr_gridsample3d_bwd_f64": [_vp, _T5, _vp, _T5, _vp, _T5, _vp, _T5, _vp, _vp],
    "sr_gridsample3d_dbwd_f32": [_vp, _T5, _vp, _T5, _vp, _T5, _vp, _T5, _vp, _T5, _vp, _T5, _vp, _vp, _vp],
    "sr_gridsample3d_dbwd_f64": [_vp, _T5, _vp, _T5, _vp, _T5, _vp, _T5, _vp, _T5, _vp, _T5, _vp, _vp, _vp],
    "sr_mlp_gemm_nt": [_vp, _vp],
    "sr_mlp_chain": [_vp, _vp],
    "sr_refine_init": [_vp, _vp],
    "sr_refine_embed": [_vp, ctypes.c_int32, _vp],
    "sr_refine_mid": [_vp, ctypes.c_int32, ctypes.c_int32, _vp],
    "sr_refine_finish": [_vp, ctypes.c_int32, _vp],
    "sr_mlp_gemm_tn_workspace_floats": [ctypes.c_int32, ctypes.c_int32, _i64, _vp],
    "sr_mlp_gemm_tn": [_vp, _vp],
    "sr_mlp_gemm_tn_group": [_vp, _vp],
    "sr_colsum_rows": [_vp, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _vp],
    "sr_lbs_chain_fwd": [_vp, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp, _vp],
    "sr_lbs_chain_bwd": [_vp, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sr_lbs_fwd": [_vp, _vp],
    "sr_lbs_bwd_workspace_floats": [_i64, ctypes.c_int32],
    "sr_lbs_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sr_lbs_jac_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sr_newton_update": [_vp, _vp],
    "sr_newton_prepare": [_vp, _vp],
    "sr_newton_apply": [_vp, _vp],
    "sr_mc_workspace_bytes": [ctypes.c_int32, ctypes.c_int32, ctypes.c_int32],
    "sr_mc_count": [_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, _vp, _vp, _vp],
    "sr_mc_emit": [_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, _vp] + [ctypes.c_float] * 6 + [_vp, _vp, _vp],
    "sr_rows_pad": [_vp, _i64, ctypes.c_int32, _vp, _i64, ctypes.c_int32, _i64, ctypes.c_int32, _vp, _i64, ctypes.c_int32, _vp],
    "sr_rows_frame_sum_workspace_floats": [_i64, ctypes.c_int32, ctypes.c_int32],
    "sr_rows_frame_sum": [_vp, _i64, _i64, ctypes.c_int32, _vp, ctypes.c_int32, _vp, _vp, _vp],
    "sr_adam_step": [_vp, _vp],
    "sr_stream_stamp": [_vp, _vp],
    "sr_stream_flag_set": [_vp, ctypes.c_uint32, _vp],
    "sr_stream_flag_wait": [_vp, ctypes.c_uint32, _vp, ctypes.c_int32, _vp],
    "sr_pack_weights": [_vp, _vp],
    "sr_unpack_grads": [_vp, _vp],
    "sr_svd3x3": [_vp, _i64, _vp, _vp, _vp, _vp],
    "sr_step_reduce_blocks": [_i64],
    "sr_step_param_blocks": [_i64],
    "sr_cam_project_ndc_fwd": [_vp, _i64, _vp, _vp, _vp, _vp],
    "sr_cam_project_ndc_bwd": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sr_cam_view_rays_fwd": [_vp, _i64, _vp, _vp, _vp],
    "sr_cam_view_rays_bwd": [_vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "sr_cardinal_rays_fwd": [_vp, _vp, _i64, _vp, _vp, _vp],
    "sr_cardinal_rays_bwd": [_vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "sr_deformed_normals": [_vp, _vp, _i64, _vp, _vp],
    "sr_color_loss_fwd": [_vp, _vp, _vp, _vp, _vp, _vp],
    "sr_color_loss_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sr_normal_loss_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, _vp, _vp, _vp],
    "sr_normal_loss_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp],
    "sr_eikonal_loss_fwd": [_vp, _i64, _vp, _vp, _vp],
    "sr_eikonal_loss_bwd": [_vp, _i64, _vp, _vp, _vp],
    "sr_def_regu_loss_fwd": [_vp, _i64, ctypes.c_float, _vp, _vp, _vp],
    "sr_def_regu_loss_bwd": [_vp, _vp, _vp, _i64, ctypes.c_float, _vp, _vp, _vp],
    "sr_svd3x3_bwd": [_vp, _vp, _vp, _i64, _vp, _vp],
    "sr_mask_iou_loss_fwd": [_vp, _vp, ctypes.c_int32, _i64, _vp, _vp, _vp],
    "sr_mask_iou_loss_bwd": [_vp, _vp, ctypes.c_int32, _i64, _vp, _vp, _vp, _vp],
    "sr_implicit_solve": [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp],
    "sr_points_silhouette_workspace_bytes": [_i64, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_float],
    "sr_points_silhouette_fwd": [_vp, _vp, _i64, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, ctypes.c_int32, _vp, _vp, _vp],
    "sr_points_silhouette_bwd": [_vp, _vp, _i64, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, _vp, _vp, _vp, _vp],
    "sr_interp2x3d_fwd_f32": [_vp, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, _vp, _vp, _vp],
    "sr_interp2x3d_fwd_f64": [_vp, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, _vp, _vp, _vp],
    "sr_interp2x3d_bwd_f32": [_vp, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _vp],
    "sr_interp2x3d_bwd_f64": [_vp, _i64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _vp],
    "sr_seg3d_candidates": [_vp, _vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp],
    "sr_rasterize_meshes": [_vp, _vp, _vp, _i64, _i64, _i64, ctypes.c_int32, ctypes.c_int32, _vp, _vp, _vp, _vp, _vp],
    "sr_pe_embed_bwd": [_vp, _i64, ctypes.c_int32, _vp, ctypes.c_int32, _vp, _i64, _vp, _vp],
    "sr_pe_embed": [_vp, _i64, ctypes.c_int32, _vp, _vp, _i64, ctypes.c_int32, _vp, ctypes.c_int32, _vp, _i64, _vp],
}
_RESTYPE = {"sr_rows_frame_sum_workspace_floats": _i64, "sr_lbs_bwd_workspace_floats": _i64, "sr_mlp_gemm_tn_workspace_floats": _i64, "sr_mc_workspace_bytes": _i64, "sr_points_silhouette_workspace_bytes": _i64}

_fn = {}
for _name, _args in SIGNATURES.items():
    try:
        _f = getattr(_lib, _name)
    except AttributeError as e:  # pragma: no cover
        raise ImportError(f"{LIB} does not export {_name}; rebuild it") from e
    _f.argtypes = _args
    _f.restype = _RESTYPE.get(_name, _int)
    _fn[_name] = _f
_lib.sr_abi_version.restype = _int
_lib.sr_build_arch.restype = ctypes.c_char_p
_lib.sr_build_digest.restype = ctypes.c_char_p


def abi_version():
    return _lib.sr_abi_version()


def build_arch():
    return _lib.sr_build_arch().decode()


def build_digest():
    return _lib.sr_build_digest().decode()


def raw(name):
    return _fn[name]


def call(name, *args):
    rc = _fn[name](*args)
    if rc != 0:
        raise SrError(f"{name} failed: {_CODES.get(rc, rc)}")


def ptr(t):
    return 0 if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_of(t):
    """hipStream_t of torch's CURRENT stream on the tensor's device (kernels are stream-ordered
    with the surrounding torch ops; the reference's FastMinv/MCGpu used the legacy default stream).
    (The raw-handle query is ~0.3 us; `torch.cuda.current_stream(dev).cuda_stream` builds a Stream object per call, ~5 us -- 120 of them per
    iteration, which shows at one frame per rank where the step is bound by the host.)"""
    if _raw_stream is not None and t.device.index is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


class _NoContext:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_CONTEXT = _NoContext()


def on_device(dev):
    """`with on_device(t.device):` = `with torch.cuda.device(t.device):` that costs nothing when that device is already current (the
    one-process-per-GPU case: ~180 context entries per iteration at ~4 us each otherwise)."""
    if dev.index is not None and dev.index == torch.cuda.current_device():
        return _NO_CONTEXT
    return torch.cuda.device(dev)


def desc5(t):
    d = SrTensor5()
    for i in range(5):
        d.size[i] = t.shape[i]
        d.stride[i] = t.stride(i)
    return d


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("selfreconcode_amd: HIP operator called with a non-GPU tensor "
                               "(there is deliberately no CPU fallback)")
