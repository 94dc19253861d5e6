"""SDF network -- drop-in for model/network.py::ImplicitNetwork / getTmpSdf (lines 14-118).

Same constructor, parameter names (`linK.weight_g / weight_v / bias`, so reference checkpoints
load unchanged), call signature `forward(input[P,3], ratio)` -> [P,1] with the 256-d feature left
in `self.rendcond`, and `.gradient()`.  The arithmetic runs on the fp32-MFMA layer kernels:
positional encoding + concat, nine weight-normed linears, Softplus(beta=100) and the skip
`cat([x, PE])/sqrt(2)` are fused into GEMM prologues/epilogues (see mlp_engine.py).
"""
import numpy as np
import torch
import torch.nn as nn

from .Embedder import embed_rows
from ..mlp_engine import MLPSpec, mlp_apply, pad_cols, pad4, pack_linear, input_grads_only
from ..utils.utils import resolve_band_weights


def effective_weight(lin):
    """W = g * v / ||v|| (weight_norm, dim=0) or the plain weight; the reference recomputes this in a
    forward pre-hook at every call (network.py:65-66)."""
    if hasattr(lin, "weight_g"):
        return torch._weight_norm(lin.weight_v, lin.weight_g, 0)
    return lin.weight


class ImplicitNetwork(nn.Module):
    def __init__(self, feature_vector_size, d_in, d_out, dims, geometric_init=True, bias=1.0, skip_in=(), weight_norm=True,
                 multires=0):
        super().__init__()
        assert d_in == 3 and multires > 0, "the HIP path implements the reference's configuration (xyz input with PE)"
        dims = [d_in] + dims + [d_out + feature_vector_size]
        self.d_out = d_out
        self.multires = multires
        dims[0] = 3 + 6 * multires
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        assert len(set(dims[1:-1])) == 1 and all(0 < s < self.num_layers - 1 for s in self.skip_in) and len(self.skip_in) <= 1
        for l in range(0, self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if l + 1 in self.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)
            if geometric_init:                      # network.py:49-63 (sphere of radius `bias`)
                if l == self.num_layers - 2:
                    torch.nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                    torch.nn.init.constant_(lin.bias, -bias)
                elif l == 0:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
                    torch.nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif l in self.skip_in:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    torch.nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            if weight_norm:
                lin = nn.utils.weight_norm(lin)
            setattr(self, "lin" + str(l), lin)
        self.spec = MLPSpec.sdf(multires, dims[1], self.num_layers - 1, self.skip_in, dims[-1])
        last = self.spec.layers[-1]
        self.spec_sdf_only = MLPSpec(self.spec.layers[:-1] + [type(last)(last.K, d_out, last.act)], self.spec.K0)
        self.rendcond = None

    def packed_weights(self):
        from ..mlp_engine import packed_weights_of
        return packed_weights_of(self, len(self.spec.layers))

    def forward(self, input, ratio=None, sdf_only=False):
        """`sdf_only=True` (extension): evaluate only the d_out distance rows of the last layer -- for callers that never
        read `rendcond` (mask/eikonal terms, refiner, volume queries): skips 256 of its 257 output rows."""
        ratio = ratio if type(ratio) == float or type(ratio) == int or ratio is None else ratio['sdfRatio']
        A0 = embed_rows(input, self.multires, resolve_band_weights(self.multires, ratio))
        Ws, bs = self.packed_weights()
        if sdf_only:
            self.rendcond = None
            return mlp_apply(self.spec_sdf_only, A0, Ws[:-1] + [Ws[-1][:self.d_out]], bs[:-1] + [bs[-1][:self.d_out]])
        x = mlp_apply(self.spec, A0, Ws, bs)
        if x.shape[-1] > self.d_out:
            self.rendcond = x[:, self.d_out:]
            x = x[:, 0:self.d_out]
        else:
            self.rendcond = None
        return x

    def gradient(self, x, y=None):
        x.requires_grad_(True)
        if y is None:
            y = self.forward(x)
        d_output = torch.ones_like(y, requires_grad=False, device=y.device)
        with input_grads_only():
            gradients = torch.autograd.grad(outputs=y, inputs=x, grad_outputs=d_output, create_graph=True, retain_graph=True,
                                            only_inputs=True)[0]
        return gradients.view(-1, 3)


def getTmpSdf(device, multires, bias=0.6, feature_vector_size=256):
    net = ImplicitNetwork(feature_vector_size=feature_vector_size, d_in=3, d_out=1, dims=[512] * 8, geometric_init=True, bias=bias,
                          skip_in=[4], weight_norm=True, multires=multires)
    return net.to(device)
