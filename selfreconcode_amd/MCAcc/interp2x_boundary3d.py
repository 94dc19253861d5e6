"""Autograd wrapper of the fused 2x upsample + boundary flag (mirrors MCAcc/interp2x_boundary3d.py:8-33)."""
import torch.nn as nn
from torch.autograd import Function
from ..ext import interp2x_boundary3d


class Interp2xBoundary3dFunction(Function):
    @staticmethod
    def forward(ctx, input, balance_value):
        output, is_boundary = interp2x_boundary3d.forward(input.contiguous(), balance_value)
        ctx.mark_non_differentiable(is_boundary)
        return output, is_boundary

    @staticmethod
    def backward(ctx, grad_output, grad_boundary):
        return interp2x_boundary3d.backward(grad_output.contiguous()), None


class Interp2xBoundary3d(nn.Module):
    def __init__(self, balance_value=0.5):
        super().__init__()
        self.balance_value = balance_value

    def forward(self, input):
        return Interp2xBoundary3dFunction.apply(input, self.balance_value)
