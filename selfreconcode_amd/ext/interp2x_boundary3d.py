"""Drop-in for the reference's `interp2x_boundary3d` pybind module (MCAcc/cuda/interp2x_boundary3d.cpp:1-36):
    forward(input[B,C,d,h,w], balance_value: float) -> [output[B,C,2d-1,2h-1,2w-1], is_boundary (bool)]
    backward(grad_output) -> grad_input
CUDA + contiguous checks as the reference's TORCH_CHECK; f32 / f64."""
import torch
from .. import _lib

_SUFFIX = {torch.float32: "f32", torch.float64: "f64"}


def _check(x, name):
    if not x.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if x.dtype not in _SUFFIX or x.dim() != 5:
        raise RuntimeError(f"{name} must be a 5-D float/double tensor")


def forward(input, balance_value):
    _check(input, "input")
    B, C, d, h, w = input.shape
    out = torch.empty((B, C, 2 * d - 1, 2 * h - 1, 2 * w - 1), dtype=input.dtype, device=input.device)
    bnd = torch.empty(out.shape, dtype=torch.bool, device=input.device)
    with _lib.on_device(input.device):
        _lib.call("sr_interp2x3d_fwd_" + _SUFFIX[input.dtype], _lib.ptr(input), B * C, d, h, w, float(balance_value), _lib.ptr(out),
                  _lib.ptr(bnd), _lib.stream_of(input))
    return [out, bnd]


def backward(grad_output):
    _check(grad_output, "grad_output")
    B, C, D, H, W = grad_output.shape
    d, h, w = (D + 1) // 2, (H + 1) // 2, (W + 1) // 2
    gi = torch.empty((B, C, d, h, w), dtype=grad_output.dtype, device=grad_output.device)
    with _lib.on_device(grad_output.device):
        _lib.call("sr_interp2x3d_bwd_" + _SUFFIX[grad_output.dtype], _lib.ptr(grad_output), B * C, d, h, w, _lib.ptr(gi), _lib.stream_of(grad_output))
    return gi
