"""Autograd glue of the 3-D sampler -- mirrors MCAcc/grid_sampler_mine.py:8-65 of the reference
(GridSamplerMine3dFunction / GridSamplerMine3dBackwardFunction: forward -> backward ->
double backward, each one HIP kernel).  grad_input is produced only when the volume
actually requires grad (ctx.needs_input_grad)."""
import torch
from torch.autograd import Function
from ..ext import GridSamplerMine


class GridSamplerMine3dFunction(Function):
    @staticmethod
    def forward(ctx, input, grid, mode='bilinear', padding_mode='border', align_corners=False):
        ctx.save_for_backward(input, grid)
        if align_corners is True:
            raise NotImplementedError
        return GridSamplerMine.forward(input, grid, 0, 1)

    @staticmethod
    def backward(ctx, grad_output):
        input, grid = ctx.saved_tensors
        o0, o1 = GridSamplerMine3dBackwardFunction.apply(input, grid, grad_output)
        return o0, o1, None, None, None


class GridSamplerMine3dBackwardFunction(Function):
    @staticmethod
    def forward(ctx, input, grid, grad_output):
        ctx.save_for_backward(input, grid, grad_output)
        ctx.want_gi = input.requires_grad
        gi, gg = GridSamplerMine.backward(input, grid, grad_output, 0, 1, want_grad_input=ctx.want_gi)
        return gi, gg

    @staticmethod
    def backward(ctx, grad_output_input, grad_output_grid):
        input, grid, grad_output = ctx.saved_tensors
        if grad_output_grid is None:
            grad_output_grid = torch.zeros_like(grid)
        o0, o1, o2 = GridSamplerMine.dbackward(grad_output_input, grad_output_grid.contiguous(), input, grid, grad_output, 0, 1,
                                               want_grad_input=ctx.needs_input_grad[0])
        return o0, o1, o2
