"""Drop-in for the reference's `FastMinv` pybind module (FastMinv/M3x3Inv.cpp:12-63).

    Fast3x3Minv(ms[N,3,3] f32/f64, GPU, contiguous) -> [invs[N,3,3], checks[N] bool]
    Fast3x3Minv_backward(grads, invs) -> outs[N,3,3]

Same argument checks as the reference's AT_ASSERTM macros (:4-6,15,42-44) raised as
RuntimeError; fresh output tensors; runs on torch's current stream.
"""
import torch
from .. import _lib

_SUFFIX = {torch.float32: "f32", torch.float64: "f64"}


def _check(x, name):
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")


def Fast3x3Minv(ms):
    _check(ms, "ms")
    if ms.dtype not in _SUFFIX:
        raise RuntimeError("rs must be a float/double tensor")
    n = ms.size(0)
    invs = torch.empty((n, 3, 3), dtype=ms.dtype, device=ms.device)
    checks = torch.empty((n,), dtype=torch.bool, device=ms.device)
    with _lib.on_device(ms.device):
        _lib.call("sr_minv3x3_fwd_" + _SUFFIX[ms.dtype], _lib.ptr(ms), _lib.ptr(invs), _lib.ptr(checks), n, _lib.stream_of(ms))
    return [invs, checks]


def Fast3x3Minv_backward(grads, invs):
    _check(grads, "grads")
    _check(invs, "invs")
    if grads.dtype not in _SUFFIX:
        raise RuntimeError("grads must be a float/double tensor")
    if invs.dtype not in _SUFFIX:
        raise RuntimeError("invs must be a float/double tensor")
    if invs.dtype != grads.dtype:
        raise RuntimeError("invs must have same type with grads")
    n = invs.size(0)
    outs = torch.empty((n, 3, 3), dtype=invs.dtype, device=invs.device)
    with _lib.on_device(invs.device):
        _lib.call("sr_minv3x3_bwd_" + _SUFFIX[invs.dtype], _lib.ptr(grads), _lib.ptr(invs), _lib.ptr(outs), n, _lib.stream_of(invs))
    return outs
