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
        with _lib.on_device(A.device):
            _lib.call("sr_svd3x3", _lib.ptr(A), n, _lib.ptr(U), _lib.ptr(S), _lib.ptr(V), _lib.stream_of(A))
        ctx.save_for_backward(U, V)
        return S

    @staticmethod
    def backward(ctx, gS):
        U, V = ctx.saved_tensors
        return ((U * gS.unsqueeze(1)).unsqueeze(-1) * V.transpose(1, 2).unsqueeze(-3)).sum(-2)      # 3x3 products: see utils.small_matmul


def singular_values_3x3(J):
    return SingularValues3x3.apply(J)


class PointsSilhouette(Function):
    """mask[N,H,W] of pytorch3d's PointsRasterizer(radius, points_per_pixel=K) + AlphaCompositor with one all-ones feature, as
    the reference renders the deformed template vertices (model/network.py:495-497, model/CameraMine.py:285-305): per pixel
    the K covering points nearest in z, composited front to back with a = 1 - dist2 / radius^2 (NDC units)."""

    @staticmethod
    def forward(ctx, xy_ndc, z, H, W, radius, K):
        _lib.require_gpu(xy_ndc)
        xy = xy_ndc.detach().contiguous().float(); zz = z.detach().contiguous().float()
        N, V = xy.shape[0], xy.shape[1]
        nbytes = _lib.raw("sr_points_silhouette_workspace_bytes")(N, V, H, W, float(radius))
        if nbytes < 0:
            raise _lib.SrError("sr_points_silhouette_workspace_bytes: bad argument")
        ws = torch.empty((max(int(nbytes), 256) + 255,), dtype=torch.uint8, device=xy.device)
        off = (-ws.data_ptr()) % 256
        mask = torch.empty((N, H, W), dtype=torch.float32, device=xy.device)
        with _lib.on_device(xy.device):
            _lib.call("sr_points_silhouette_fwd", _lib.ptr(xy), _lib.ptr(zz), N, V, H, W, float(radius), int(K), _lib.ptr(mask),
                      ws.data_ptr() + off, _lib.stream_of(xy))
        ctx.save_for_backward(xy, zz, ws)
        ctx.dims = (H, W, float(radius), off)
        return mask

    @staticmethod
    def backward(ctx, gmask):
        xy, zz, ws = ctx.saved_tensors
        H, W, radius, off = ctx.dims
        gxy = torch.empty_like(xy)
        with _lib.on_device(xy.device):
            _lib.call("sr_points_silhouette_bwd", _lib.ptr(xy), _lib.ptr(zz), xy.shape[0], xy.shape[1], H, W, radius, ws.data_ptr() + off,
                      _lib.ptr(gmask.contiguous().float()), _lib.ptr(gxy), _lib.stream_of(xy))
        return gxy, None, None, None, None, None


def _require_square(H, W, what):
    # pytorch3d 0.4.0 rescales the NDC range of the longer side of a non-square image; the kernels (pix_to_ndc = 1 - (2 i + 1) / S on
    # both axes) and the reference's own camera class implement the square convention only -- refuse instead of rendering something
    # the reference pipeline would not
    if int(H) != int(W):
        raise RuntimeError(f"{what}: non-square images ({H} x {W}) are not supported (pytorch3d 0.4.0 rescales NDC for them; see DESIGN.md 8)")


def points_silhouette(xy_ndc, z, H, W, radius, points_per_pixel=50):
    _require_square(H, W, "points_silhouette")
    return PointsSilhouette.apply(xy_ndc, z, H, W, radius, points_per_pixel)


class Fragments:
    """pix_to_face [N,H,W,1], bary_coords [N,H,W,1,3], zbuf [N,H,W,1] -- the fields FindSurfacePs reads from pytorch3d's Fragments."""

    def __init__(self, pix_to_face, bary_coords, zbuf):
        self.pix_to_face, self.bary_coords, self.zbuf = pix_to_face, bary_coords, zbuf


def rasterize_meshes(xy_ndc, z, faces, H, W):
    """No-grad hard rasterisation of N images of one mesh topology with the semantics of the reference's MeshRasterizer
    settings (model/network.py:877-892; see sr_rasterize_meshes)."""
    _lib.require_gpu(xy_ndc)
    _require_square(H, W, "rasterize_meshes")
    xy = xy_ndc.detach().contiguous().float(); z = z.detach().contiguous().float(); faces = faces.contiguous()
    N, V = xy.shape[0], xy.shape[1]
    dev = xy.device
    zbuf = torch.empty((N, H, W), dtype=torch.int64, device=dev)
    p2f = torch.empty((N, H, W), dtype=torch.int64, device=dev)
    bary = torch.empty((N, H, W, 3), dtype=torch.float32, device=dev)
    zo = torch.empty((N, H, W), dtype=torch.float32, device=dev)
    with _lib.on_device(dev):
        _lib.call("sr_rasterize_meshes", _lib.ptr(xy), _lib.ptr(z), _lib.ptr(faces), N, V, faces.shape[0], H, W, _lib.ptr(zbuf), _lib.ptr(p2f),
                  _lib.ptr(bary), _lib.ptr(zo), _lib.stream_of(xy))
    return Fragments(p2f.unsqueeze(-1), bary.unsqueeze(3), zo.unsqueeze(-1))
