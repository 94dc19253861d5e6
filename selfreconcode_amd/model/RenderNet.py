"""Rendering MLP -- drop-in for model/RenderNet.py::RenderingNetwork_view_norm / getRenderNet (9-95):
cat[p, PE_4(view), n, feat_256] -> 5 weight-normed linears (ReLU) -> tanh, on the fp32-MFMA kernels."""
import torch
import torch.nn as nn

from .Embedder import embed_rows
from .network import effective_weight
from ..mlp_engine import MLPSpec, mlp_apply, pad_cols, pad4, pack_linear, refresh_packs, packed_weights_of
from ..utils.utils import resolve_band_weights


class RenderingNetwork_view_norm(nn.Module):
    def __init__(self, feature_vector_size, mode, d_in, d_out, dims, weight_norm=True, multires_n=0, multires_v=0):
        super().__init__()
        assert mode == 'idr' and multires_n == 0, "the reference's shipped configuration (config.conf:62-69)"
        self.mode = mode
        dims = [d_in + feature_vector_size] + dims + [d_out]
        self.multires_v, self.multires_n = multires_v, multires_n
        if multires_v > 0:
            dims[0] += 6 * multires_v
        self.num_layers = len(dims)
        for l in range(0, self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            if weight_norm:
                lin = nn.utils.weight_norm(lin)
            setattr(self, "lin" + str(l), lin)
        self.spec = MLPSpec.relu_mlp(dims[0], dims[1:])

    def forward(self, points, normals, view_dirs, feature_vectors, ratio):
        ratio = ratio['renderRatio']
        nv = 3 + 6 * self.multires_v
        v = embed_rows(view_dirs, self.multires_v, resolve_band_weights(self.multires_v, ratio))[:, :nv]
        width = points.shape[1] + nv + normals.shape[1] + feature_vectors.shape[1]
        parts = [points, v, normals, feature_vectors]
        if pad4(width) != width:                  # the row pitch the GEMM wants, as a fifth block of the same cat (no separate pad pass)
            parts.append(points.new_zeros((points.shape[0], pad4(width) - width)))
        x = torch.cat(parts, dim=-1)
        Ws, bs = packed_weights_of(self, len(self.spec.layers))      # one launch for all stale layers after an optimizer step
        return torch.tanh(mlp_apply(self.spec, x, Ws, bs))


def getRenderNet(device, conf):
    assert conf.get_string('type') == 'RenderingNetwork_view_norm'
    return RenderingNetwork_view_norm(conf.get_int('condlen'), d_in=9, d_out=3, dims=[512, 512, 512, 512], mode='idr',
                                      weight_norm=True, multires_v=conf.get_int('multires_v'),
                                      multires_n=conf.get_int('multires_n')).to(device)
