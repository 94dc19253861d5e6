"""Autograd faces of the small HIP ops used by the training step."""
import torch
from torch.autograd import Function
from . import _lib


class SingularValues3x3(Function):
    """s[n,3] (descending) of J[n,3,3]; d s_k / d J = u_k v_k^T -- what autograd of torch.svd's S gives,
    without the CPU round trip of model/network.py:576."""

    @staticmethod
    def forward(ctx, J):
        _lib.require_gpu(J)
        A = J.detach().contiguous().float().view(-1, 3, 3)
        n = A.shape[0]
        U = torch.empty_like(A); V = torch.empty_like(A)
        S = torch.empty((n, 3), dtype=torch.float32, device=A.device)
        with torch.cuda.device(A.device):
            _lib.call("sr_svd3x3", _lib.ptr(A), n, _lib.ptr(U), _lib.ptr(S), _lib.ptr(V), _lib.stream_of(A))
        ctx.save_for_backward(U, V)
        return S

    @staticmethod
    def backward(ctx, gS):
        U, V = ctx.saved_tensors
        return (U * gS.unsqueeze(1)) @ V.transpose(1, 2)


def singular_values_3x3(J):
    return SingularValues3x3.apply(J)


class SplatSilhouette(Function):
    """mask[N,H,W] = 1 - prod_k (1 - a_k) over the points splatted within `radius_px` of each pixel."""

    @staticmethod
    def forward(ctx, pix, vis, H, W, radius_px):
        _lib.require_gpu(pix)
        pix = pix.contiguous().float()
        N, V = pix.shape[0], pix.shape[1]
        visb = None if vis is None else vis.contiguous().to(torch.uint8)
        logT = torch.zeros((N, H, W), dtype=torch.float32, device=pix.device)
        with torch.cuda.device(pix.device):
            _lib.call("sr_splat_fwd", _lib.ptr(pix), _lib.ptr(visb), N, V, H, W, float(radius_px), _lib.ptr(logT), _lib.stream_of(pix))
        ctx.save_for_backward(pix, visb, logT)
        ctx.dims = (H, W, float(radius_px))
        return 1.0 - torch.exp(logT)

    @staticmethod
    def backward(ctx, gmask):
        pix, visb, logT = ctx.saved_tensors
        H, W, r = ctx.dims
        gpix = torch.empty_like(pix)
        with torch.cuda.device(pix.device):
            _lib.call("sr_splat_bwd", _lib.ptr(pix), _lib.ptr(visb), pix.shape[0], pix.shape[1], H, W, r, _lib.ptr(logT),
                      _lib.ptr(gmask.contiguous().float()), _lib.ptr(gpix), _lib.stream_of(pix))
        return gpix, None, None, None, None


def splat_silhouette(pix, vis, H, W, radius_px):
    return SplatSilhouette.apply(pix, vis, H, W, radius_px)


class Fragments:
    """pix_to_face [N,H,W,1], bary_coords [N,H,W,1,3], zbuf [N,H,W,1] -- the fields FindSurfacePs reads from pytorch3d's Fragments."""

    def __init__(self, pix_to_face, bary_coords, zbuf):
        self.pix_to_face, self.bary_coords, self.zbuf = pix_to_face, bary_coords, zbuf


def rasterize_mesh(pix, z, faces, H, W):
    """No-grad hard rasterisation of N images of one mesh topology (see sr_raster_mesh)."""
    _lib.require_gpu(pix)
    pix = pix.detach().contiguous().float(); z = z.detach().contiguous().float(); faces = faces.contiguous()
    N, V = pix.shape[0], pix.shape[1]
    dev = pix.device
    zbuf = torch.empty((N, H, W), dtype=torch.int64, device=dev)
    p2f = torch.empty((N, H, W), dtype=torch.int64, device=dev)
    bary = torch.empty((N, H, W, 3), dtype=torch.float32, device=dev)
    zo = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.call("sr_raster_mesh", _lib.ptr(pix), _lib.ptr(z), _lib.ptr(faces), N, V, faces.shape[0], H, W, _lib.ptr(zbuf), _lib.ptr(p2f),
                  _lib.ptr(bary), _lib.ptr(zo), _lib.stream_of(pix))
    return Fragments(p2f.unsqueeze(-1), bary.unsqueeze(3), zo.unsqueeze(-1))
