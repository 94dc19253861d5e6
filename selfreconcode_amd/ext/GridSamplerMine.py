"""Drop-in for the reference's `GridSamplerMine` pybind module (MCAcc/cuda/GridSamplerMine.cpp:73-104).

    forward(input[N,C,D,H,W], grid[N,Do,Ho,Wo,3], interp=0, pad=1) -> out[N,C,Do,Ho,Wo]
    backward(input, grid, grad_output, 0, 1) -> (grad_input, grad_grid)
    dbackward(gOut_input, gOut_grid, input, grid, grad_output, 0, 1) -> (grad_input, grad_grid, grad_grad_output)

Checks mirror GridSamplerMine.cpp:24-71 (TORCH_CHECK -> RuntimeError).  Arbitrary strides are
honoured.  Extension over the reference: `want_grad_input=False` skips the zero-fill + atomics into
a volume-sized grad_input (the reference always pays them, GridSamplerMineKernel.cu:948-980, although
the skinning volume is a buffer); the autograd glue in MCAcc/grid_sampler_mine.py uses it.
"""
import torch
from .. import _lib

_SUFFIX = {torch.float16: "f16", torch.float32: "f32", torch.float64: "f64"}      # AT_DISPATCH_FLOATING_TYPES_AND_HALF


def _check(input, grid, interpolation_mode, padding_mode):
    if input is None or grid is None:
        raise RuntimeError("grid_sampler(): expected input and grid to not be undefined")
    if input.device != grid.device:
        raise RuntimeError(f"grid_sampler(): expected input and grid to be on same device, but input is on {input.device} and grid is on {grid.device}")
    if input.dtype != grid.dtype:
        raise RuntimeError(f"grid_sampler(): expected input and grid to have same dtype, but input has {input.dtype} and grid has {grid.dtype}")
    if input.layout != torch.strided or grid.layout != torch.strided:
        raise RuntimeError("grid_sampler(): expected input and grid to have torch.strided layout")
    if input.dim() != 5 or grid.dim() != 5:
        raise RuntimeError(f"grid_sampler(): expected 5D input and grid with same number of dimensions, but got input with sizes {tuple(input.shape)} and grid with sizes {tuple(grid.shape)}")
    if input.size(0) != grid.size(0):
        raise RuntimeError("grid_sampler(): expected grid and input to have same batch size")
    if grid.size(-1) != 3:
        raise RuntimeError("grid_sampler(): expected grid to have size 3 in last dimension")
    if interpolation_mode != 0:
        raise RuntimeError("grid_sampler(): only support Bilinear now")
    if padding_mode != 1:
        raise RuntimeError("grid_sampler(): only support Border Padding now")
    for i in range(2, 5):
        if input.size(i) <= 0:
            raise RuntimeError("grid_sampler(): expected input to have non-empty spatial dimensions")
    if input.dtype not in _SUFFIX:
        raise RuntimeError(f"grid_sampler(): \"grid_sampler_3d_cuda\" not implemented for '{input.dtype}'")
    _lib.require_gpu(input, grid)


def forward(input, grid, interpolation_mode=0, padding_mode=1):
    _check(input, grid, interpolation_mode, padding_mode)
    N, C = input.shape[:2]
    out = torch.empty((N, C) + tuple(grid.shape[1:4]), dtype=input.dtype, device=input.device)
    with _lib.on_device(input.device):
        _lib.call("sr_gridsample3d_fwd_" + _SUFFIX[input.dtype], _lib.ptr(input), _lib.desc5(input), _lib.ptr(grid), _lib.desc5(grid),
                  _lib.ptr(out), _lib.desc5(out), _lib.stream_of(input))
    return out


def backward(input, grid, grad_output, interpolation_mode=0, padding_mode=1, want_grad_input=True):
    _check(input, grid, interpolation_mode, padding_mode)
    grad_input = torch.zeros_like(input, memory_format=torch.contiguous_format) if want_grad_input else None
    grad_grid = torch.empty(tuple(grid.shape), dtype=grid.dtype, device=grid.device)
    gi_desc = _lib.desc5(grad_input) if want_grad_input else _lib.desc5(input)
    with _lib.on_device(input.device):
        _lib.call("sr_gridsample3d_bwd_" + _SUFFIX[input.dtype], _lib.ptr(input), _lib.desc5(input), _lib.ptr(grid), _lib.desc5(grid),
                  _lib.ptr(grad_output), _lib.desc5(grad_output), _lib.ptr(grad_input), gi_desc, _lib.ptr(grad_grid),
                  _lib.stream_of(input))
    return grad_input, grad_grid


def dbackward(grad_output_input, grad_output_grid, input, grid, grad_output, interpolation_mode=0, padding_mode=1,
              want_grad_input=True):
    _check(input, grid, interpolation_mode, padding_mode)
    grad_input = torch.zeros_like(input, memory_format=torch.contiguous_format) if want_grad_input else None
    grad_grid = torch.empty(tuple(grid.shape), dtype=grid.dtype, device=grid.device)
    N, C = input.shape[:2]
    ggo = torch.empty((N, C) + tuple(grid.shape[1:4]), dtype=input.dtype, device=input.device)
    gi_desc = _lib.desc5(grad_input) if want_grad_input else _lib.desc5(input)
    goi_desc = _lib.desc5(grad_output_input) if grad_output_input is not None else _lib.desc5(input)
    with _lib.on_device(input.device):
        _lib.call("sr_gridsample3d_dbwd_" + _SUFFIX[input.dtype], _lib.ptr(grad_output_input), goi_desc,
                  _lib.ptr(grad_output_grid), _lib.desc5(grad_output_grid), _lib.ptr(input), _lib.desc5(input),
                  _lib.ptr(grid), _lib.desc5(grid), _lib.ptr(grad_output), _lib.desc5(grad_output),
                  _lib.ptr(grad_input), gi_desc, _lib.ptr(grad_grid), _lib.ptr(ggo), _lib.stream_of(input))
    return grad_input, grad_grid, ggo
