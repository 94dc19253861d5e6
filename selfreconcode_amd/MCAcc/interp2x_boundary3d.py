"""Differentiable 2x upsampling with the sign-disagreement flag, on top of `sr_interp2x3d_{fwd,bwd}` (csrc/interp2x.hip).

Public names follow MCAcc/interp2x_boundary3d.py:8-33 -- `Interp2xBoundary3d(balance_value)(volume) -> (fine, is_boundary)` and the
underlying `Interp2xBoundary3dFunction` -- because `Seg3dLossless(use_cuda_impl=True)` looks them up by name.  The flag output is a
byproduct without a gradient; the adjoint of the interpolation is its own kernel (each coarse voxel gathers its <= 27 fine neighbours),
so nothing is saved for backward."""
import torch
from ..ext import interp2x_boundary3d as _ops


class Interp2xBoundary3dFunction(torch.autograd.Function):
    """(coarse [B,C,d,h,w], balance) -> (fine [B,C,2d-1,2h-1,2w-1], bool flags of the same shape)."""

    @staticmethod
    def forward(ctx, coarse, balance):
        ctx.set_materialize_grads(False)
        fine, flags = _ops.forward(coarse if coarse.is_contiguous() else coarse.contiguous(), float(balance))
        ctx.mark_non_differentiable(flags)
        return fine, flags

    @staticmethod
    def backward(ctx, d_fine, _d_flags):
        if d_fine is None:
            return None, None
        return _ops.backward(d_fine if d_fine.is_contiguous() else d_fine.contiguous()), None


class Interp2xBoundary3d(torch.nn.Module):
    """Module form; `balance_value` is the iso level whose crossing marks a fine voxel as boundary."""

    def __init__(self, balance_value=0.5):
        torch.nn.Module.__init__(self)
        self.balance_value = float(balance_value)

    def extra_repr(self):
        return f"balance_value={self.balance_value}"

    def forward(self, volume):
        fine, flags = Interp2xBoundary3dFunction.apply(volume, self.balance_value)
        return fine, flags
