"""Non-rigid deformation field -- drop-in for model/Deformer.py (CompositeDeformer :10-20,
MLPTranslator :22-76, LBSkinner :86-233)."""
import numpy as np
import torch
import torch.nn as nn

from .Embedder import embed_rows
from ..mlp_engine import MLPSpec, mlp_apply, pad_cols, pad4
from ..utils.utils import resolve_band_weights


class CompositeDeformer(nn.Module):
    def __init__(self, deformers):
        super().__init__()
        self.N = len(deformers)
        self.defs = nn.ModuleList(deformers)

    def forward(self, ps, conds, batch_inds=None, **kwargs):
        assert self.N == len(conds)
        out = ps
        for cond, deformer in zip(conds, self.defs):
            out = deformer(out, cond, batch_inds, **kwargs)
        return out


class MLPTranslator(nn.Module):
    """PE_6(p) (+) per-frame code -> 5 linears (ReLU) -> offset; returns p + offset and keeps
    `self.offset`.  The code columns ride in the fused embed kernel (gathered by batch index)."""

    def __init__(self, feature_vector_size, multires, weight_norm=False):
        super().__init__()
        assert multires > 0
        dims = [3 + 6 * multires + feature_vector_size, 512, 512, 512, 512, 3]
        self.feature_vector_size = feature_vector_size
        self.multires = multires
        self.num_layers = len(dims)
        for l in range(0, self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            if weight_norm:
                print('MLPTranslator:weight norm can influence weight initialization, can not produce small weights as '
                      'initialization. Now do not use weight_norm')
            if l == self.num_layers - 2:            # zero-translation start (Deformer.py:44-46)
                torch.nn.init.normal_(lin.weight, mean=0., std=0.001)
                torch.nn.init.constant_(lin.bias, 0.)
            setattr(self, "lin" + str(l), lin)
        self.spec = MLPSpec.relu_mlp(dims[0], dims[1:])
        self.offset = None

    def packed_weights(self):
        Ws, bs = [], []
        for l, L in enumerate(self.spec.layers):
            lin = getattr(self, "lin" + str(l))
            Ws.append(pad_cols(lin.weight, pad4(L.K)))
            bs.append(lin.bias)
        return Ws, bs

    def forward(self, ps, conds, batch_inds=None, **kwargs):
        ratio = kwargs['ratio']['deformerRatio']
        ws = resolve_band_weights(self.multires, ratio)
        if batch_inds is not None:
            flat, index = ps, batch_inds
        else:                                        # [N, V, 3] with one code per frame
            nb, nv = ps.shape[0], ps.shape[1]
            flat = ps.reshape(-1, 3)
            index = torch.arange(nb, device=ps.device).repeat_interleave(nv)
        A0 = embed_rows(flat, self.multires, ws, extra=conds.reshape(-1, self.feature_vector_size), extra_index=index)
        Ws, bs = self.packed_weights()
        x = mlp_apply(self.spec, A0, Ws, bs)
        if batch_inds is not None:
            self.offset = x
            return ps[..., :3] + x
        self.offset = x.view(ps.shape[0], ps.shape[1], 3)
        return ps[..., :3] + self.offset


def getTranslatorNet(device, conf):
    return MLPTranslator(conf.get_int('condlen'), multires=conf.get_int('multires')).to(device)
