"""GPU checks of the two in-repo rasterisers that stand where the reference calls pytorch3d (third-party, parity
unpinned -- SURVEY.md 8(c)): each is compared with a plain numpy / torch restatement of its own definition."""
import numpy as np
import pytest
import torch
from oracle import fixtures as fx

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _raster_numpy(pix, z, faces, H, W):
    N = pix.shape[0]
    F = faces.shape[0]
    p2f = -np.ones((N, H, W), np.int64); bary = -np.ones((N, H, W, 3), np.float32); zb = np.full((N, H, W), np.inf, np.float32)
    for n in range(N):
        for f in range(F):
            a, b, c = faces[f]
            if min(a, b, c) < 0:
                continue
            (x0, y0), (x1, y1), (x2, y2) = pix[n, a], pix[n, b], pix[n, c]
            z0, z1, z2 = z[n, a], z[n, b], z[n, c]
            if min(z0, z1, z2) <= 0:
                continue
            area = np.float32((x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0))
            if abs(area) < 1e-12:
                continue
            for y in range(max(0, int(np.ceil(min(y0, y1, y2)))), min(H - 1, int(np.floor(max(y0, y1, y2)))) + 1):
                for x in range(max(0, int(np.ceil(min(x0, x1, x2)))), min(W - 1, int(np.floor(max(x0, x1, x2)))) + 1):
                    px, py = np.float32(x), np.float32(y)
                    w0 = np.float32(((x1 - px) * (y2 - py) - (x2 - px) * (y1 - py)) / area)
                    w1 = np.float32(((x2 - px) * (y0 - py) - (x0 - px) * (y2 - py)) / area)
                    w2 = np.float32(1.0) - w0 - w1
                    if w0 < 0 or w1 < 0 or w2 < 0:
                        continue
                    i0, i1, i2 = w0 / z0, w1 / z1, w2 / z2
                    s = i0 + i1 + i2
                    d = np.float32(1.0) / s
                    if d < zb[n, y, x] or (d == zb[n, y, x] and f < p2f[n, y, x] % F):
                        zb[n, y, x] = d; p2f[n, y, x] = n * F + f; bary[n, y, x] = (i0 / s, i1 / s, i2 / s)
    return p2f, bary, zb


def test_mesh_rasteriser_vs_numpy_and_find_surface_ps():
    from selfreconcode_amd.ops import rasterize_mesh
    from selfreconcode_amd.utils.FindSurfacePs import FindSurfacePs
    N, V, F, H, W = 2, 40, 60, 24, 28
    pix = (fx.det_array((N, V, 2), 1, 1.0) * np.array([16.0, 14.0]) + np.array([14.0, 12.0])).astype(np.float32)
    z = (fx.det_array((N, V), 2, 0.8) + 2.0).astype(np.float32)
    faces = (np.abs(fx.det_array((F, 3), 3, 1000.0)).astype(np.int64)) % V
    faces[5] = -1                                                    # MC border faces carry -1 (MCGpu semantics)
    ref_p2f, ref_bary, ref_z = _raster_numpy(pix, z, faces, H, W)
    fr = rasterize_mesh(torch.from_numpy(pix).to(DEV), torch.from_numpy(z).to(DEV), torch.from_numpy(faces).to(DEV), H, W)
    p2f = fr.pix_to_face[..., 0].cpu().numpy(); bary = fr.bary_coords[:, :, :, 0].cpu().numpy()
    agree = (p2f == ref_p2f)
    assert agree.mean() > 0.995                                       # fp ties at shared edges may pick the neighbour
    hit = agree & (ref_p2f >= 0)
    assert hit.sum() > 100 and np.allclose(bary[hit], ref_bary[hit], atol=2e-5)
    assert (bary[p2f >= 0] >= 0).all() and np.allclose(bary[p2f >= 0].sum(-1), 1.0, atol=1e-5)
    Vc = fx.det_tensor((V, 3), 4, 1.0).to(DEV)
    safe_faces = torch.from_numpy(np.where(faces < 0, 0, faces)).to(DEV)
    b, r, c, p0, finds = FindSurfacePs(Vc, safe_faces, fr)
    sel = (fr.bary_coords[b, r, c, 0] > 0).all(-1)
    exp = (Vc[safe_faces[finds]] * fr.bary_coords[b, r, c, 0].unsqueeze(-1)).sum(1)
    assert sel.all() and torch.allclose(p0, exp, atol=1e-6)


def test_point_splat_silhouette_forward_backward():
    from selfreconcode_amd.ops import splat_silhouette
    N, V, H, W, r = 2, 300, 20, 22, 1.7
    pix = (fx.det_tensor((N, V, 2), 5, 1.0) * torch.tensor([12.0, 11.0]) + torch.tensor([11.0, 10.0]))
    vis = fx.det_tensor((N, V), 6, 1.0) > -0.8

    def dense(p):                                                     # definition: 1 - prod_k (1 - clamp(1 - d^2/r^2))
        ys, xs = torch.meshgrid(torch.arange(H, dtype=p.dtype), torch.arange(W, dtype=p.dtype), indexing='ij')
        d2 = (xs[None, None] - p[:, :, 0, None, None]) ** 2 + (ys[None, None] - p[:, :, 1, None, None]) ** 2     # [N,V,H,W]
        a = torch.where((d2 < r * r) & vis[:, :, None, None], (1 - d2 / (r * r)).clamp(max=0.9999), torch.zeros_like(d2))
        return 1 - torch.prod(1 - a, dim=1)
    pd = pix.double().requires_grad_(True)
    ref = dense(pd)
    go = fx.det_tensor((N, H, W), 7, 1.0)
    (gref,) = torch.autograd.grad(ref, pd, go.double())
    pg = pix.to(DEV).requires_grad_(True)
    m = splat_silhouette(pg, vis.to(DEV), H, W, r)
    torch.testing.assert_close(m.cpu(), ref.float(), rtol=1e-4, atol=2e-5)
    (g,) = torch.autograd.grad(m, pg, go.to(DEV))
    torch.testing.assert_close(g.cpu(), gref.float(), rtol=2e-3, atol=2e-4)
    assert float(m.min()) >= 0 and float(m.max()) <= 1
