"""GPU parity of the two rasterisation kernels (csrc/raster.hip) with the restatement of the pytorch3d 0.4.0 calls the
reference makes (oracle/raster_oracle.py -- third-party arithmetic, parity unpinned: the restatement follows the published
algorithm and the reference's call sites, it could not be run against pytorch3d here)."""
import numpy as np
import pytest
import torch
from oracle import fixtures as fx
from oracle import raster_oracle as ro

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FOCAL, PRINC = torch.tensor([130., 128.]), torch.tensor([31.3, 32.6])
R = torch.tensor([[-1., 0., 0.], [0., 1., 0.], [0., 0., -1.]])
T = torch.tensor([0.02, -0.03, 2.4])


def _camera(H, W):
    from selfreconcode_amd.model.CameraMine import RectifiedPerspectiveCameras
    return RectifiedPerspectiveCameras(FOCAL.view(1, 2), PRINC.view(1, 2), R.view(1, 3, 3), T.view(1, 3), image_size=[(W, H)]).to(DEV)


def test_project_ndc_equals_the_reference_camera_chain():
    cam = _camera(64, 64)
    p = fx.det_tensor((2, 50, 3), 3, 0.5)
    xy, z = cam.project_ndc(p.to(DEV))
    xyo, zo = ro.ndc_projection(p, FOCAL, PRINC, R, T, 64, 64)
    torch.testing.assert_close(xy.cpu(), xyo, rtol=1e-6, atol=1e-6); torch.testing.assert_close(z.cpu(), zo, rtol=1e-6, atol=1e-6)
    pix, _ = cam.project(p.to(DEV))                                    # NDC <-> pixel: col = ((1 - x) W - 1) / 2
    torch.testing.assert_close(((1 - xy) * 64 - 1) / 2, pix, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("K,V,spread", [(50, 600, 0.35), (5, 900, 0.2), (50, 6000, 0.08)])
def test_points_silhouette_forward_backward_vs_pytorch3d_restatement(K, V, spread):
    """(50, 600): nobody exceeds K -> plain 1 - prod(1 - a).  (5, 900) and (50, 6000 points squeezed into a few pixels):
    many pixels are covered by MORE than K points -> the nearest-in-z selection decides value and gradient."""
    from selfreconcode_amd.ops import points_silhouette
    N, H, W, radius = 2, 64, 64, 0.06
    pts = fx.det_tensor((N, V, 3), 11 + V, 1.0) * torch.tensor([spread, spread, 0.3])
    pts[0, :7, 2] = 5.0                                                # a few points behind the camera (z_view < 0)
    po = pts.double().requires_grad_(True)
    mo, idx = ro.render_point_silhouette(po, FOCAL.double(), PRINC.double(), R.double(), T.double(), H, W, radius, K)
    ncover = (idx >= 0).sum(-1)
    if V != 600:
        assert int((ncover == K).sum()) > 20                           # the truncation is exercised
    go = fx.det_tensor((N, H, W), 17, 1.0)
    (gref,) = torch.autograd.grad(mo, po, go.double())
    cam = _camera(H, W)
    pg = pts.to(DEV).requires_grad_(True)
    xy, z = cam.project_ndc(pg)
    m = points_silhouette(xy, z, H, W, radius, K)
    torch.testing.assert_close(m.cpu(), mo.float(), rtol=1e-4, atol=3e-5)
    (g,) = torch.autograd.grad(m, pg, go.to(DEV))
    torch.testing.assert_close(g.cpu(), gref.float(), rtol=2e-3, atol=2e-4 * float(gref.abs().max()))
    assert float(m.min()) >= 0 and float(m.max()) <= 1


def _mesh(V=40, F=70):
    verts = fx.det_tensor((2, V, 3), 1, 1.0) * torch.tensor([0.45, 0.45, 0.25])
    faces = (np.abs(fx.det_array((F, 3), 3, 1000.0)).astype(np.int64)) % V
    faces[5] = -1                                                      # MC border faces carry -1 (MCGpu semantics)
    return verts, faces


def test_mesh_rasteriser_vs_pytorch3d_restatement_and_find_surface_ps():
    from selfreconcode_amd.ops import rasterize_meshes
    from selfreconcode_amd.utils.FindSurfacePs import FindSurfacePs
    H = W = 48
    verts, faces = _mesh()
    cam = _camera(H, W)
    xy, z = cam.project_ndc(verts.to(DEV))
    xyo, zo = ro.ndc_projection(verts, FOCAL, PRINC, R, T, W, H)
    ref_p2f, ref_bary, ref_z = ro.rasterize_meshes(torch.cat([xyo, zo[..., None]], -1).numpy(), faces, H, W)
    fr = rasterize_meshes(xy, z, torch.from_numpy(faces).to(DEV), H, W)
    p2f = fr.pix_to_face.cpu().numpy(); bary = fr.bary_coords.cpu().numpy(); zb = fr.zbuf.cpu().numpy()
    agree = (p2f == ref_p2f)
    assert agree.mean() > 0.997                                        # fp ties at shared edges may pick the neighbour
    hit = (agree & (ref_p2f >= 0))[..., 0]
    assert hit.sum() > 300
    assert np.allclose(bary[..., 0, :][hit], ref_bary[..., 0, :][hit], atol=2e-5) and np.allclose(zb[..., 0][hit], ref_z[..., 0][hit], rtol=1e-5)
    assert (bary[p2f[..., 0] >= 0] > 0).all() and np.allclose(bary[p2f[..., 0] >= 0].sum(-1), 1.0, atol=1e-5)
    Vc = fx.det_tensor((verts.shape[1], 3), 4, 1.0).to(DEV)
    safe_faces = torch.from_numpy(np.where(faces < 0, 0, faces)).to(DEV)
    b, r, c, p0, finds = FindSurfacePs(Vc, safe_faces, fr)
    assert b.numel() == int((p2f >= 0).sum())
    exp = (Vc[safe_faces[finds]] * fr.bary_coords[b, r, c, 0].unsqueeze(-1)).sum(1)
    assert torch.allclose(p0, exp, atol=1e-6)


def test_mesh_rasteriser_long_thin_and_huge_faces():
    """The wave-per-face pass of the mesh rasteriser (boxes of more than 32 pixel centres) walks, per row, only the columns between the two
    edge crossings of that row: long thin faces (the stretched faces of a template whose neighbouring vertices follow different bones),
    faces across the whole image, horizontal / vertical edges exactly on pixel-centre lines, vertices outside the image, and faces with a
    vertex BEHIND the camera (unclipped as in pytorch3d 0.4.0: the passing pixels are not the projected triangle's interior there, so
    those faces keep the test of every pixel centre of the box) -- all against the pytorch3d restatement, pixel for pixel."""
    from selfreconcode_amd.ops import rasterize_meshes
    H = W = 160
    g = np.random.default_rng(7)
    V = 400
    xy = g.uniform(-1.3, 1.3, (1, V, 2)).astype(np.float32)
    z = g.uniform(0.6, 3.0, (1, V)).astype(np.float32)
    faces = []
    for _ in range(120):                                               # thin slivers: two vertices close together, the third far away
        a = g.integers(0, V - 2)
        xy[0, a + 1] = xy[0, a] + g.uniform(-0.02, 0.02, 2)
        faces.append([a, a + 1, g.integers(0, V)])
    faces += [list(g.choice(V, 3, replace=False)) for _ in range(60)]  # arbitrary (mostly huge) faces
    # edges exactly on pixel-centre lines: y = 1 - (2 r + 1) / H and x = 1 - (2 c + 1) / W
    yl, xl = 1.0 - (2 * 37 + 1) / H, 1.0 - (2 * 91 + 1) / W
    xy[0, 0] = [-0.9, yl]; xy[0, 1] = [0.8, yl]; xy[0, 2] = [0.1, yl + 0.31]; faces.append([0, 1, 2])
    xy[0, 3] = [xl, -0.7]; xy[0, 4] = [xl, 0.9]; xy[0, 5] = [xl - 0.4, 0.2]; faces.append([3, 4, 5])
    for k in (10, 11, 12, 13):                                          # a vertex behind the camera
        z[0, k] = -0.7
        faces.append([k, 40 + k, 80 + k])
    faces = np.asarray(faces, np.int64)
    ref_p2f, ref_bary, ref_z = ro.rasterize_meshes(np.concatenate([xy, z[..., None]], -1), faces, H, W)
    fr = rasterize_meshes(torch.from_numpy(xy).to(DEV), torch.from_numpy(z).to(DEV), torch.from_numpy(faces).to(DEV), H, W)
    p2f = fr.pix_to_face.cpu().numpy(); bary = fr.bary_coords.cpu().numpy()
    cover = (ref_p2f >= 0).mean()
    assert cover > 0.5                                                 # the faces do cover a large part of the image
    same = p2f == ref_p2f
    # coverage must be identical; the winning face may differ only where two faces tie in depth to the last bits
    assert np.array_equal(p2f >= 0, ref_p2f >= 0), int(((p2f >= 0) != (ref_p2f >= 0)).sum())
    assert same.mean() > 0.999, same.mean()
    hit = (same & (ref_p2f >= 0))[..., 0]
    assert np.allclose(bary[..., 0, :][hit], ref_bary[..., 0, :][hit], atol=5e-5)
    print("coverage %.3f, identical winners %.5f" % (cover, same.mean()))
    # faces with TWO vertices behind the camera pass pixels OUTSIDE their projected triangle (the signs of the perspective-corrected
    # barycentrics flip): single faces whose passing region the restatement says is 700-1000 pixel centres of a 96 x 96 image
    for seed in (43, 129, 141, 191):
        g = np.random.default_rng(seed)
        xy1 = g.uniform(-0.9, 0.9, (1, 3, 2)).astype(np.float32); z1 = g.uniform(0.6, 3.0, (1, 3)).astype(np.float32)
        z1[0, :2] = -g.uniform(0.2, 2.0, 2).astype(np.float32)
        f1 = np.array([[0, 1, 2]], np.int64)
        r1 = ro.rasterize_meshes_loop(np.concatenate([xy1, z1[..., None]], -1), f1, 96, 96)[0]
        p1 = rasterize_meshes(torch.from_numpy(xy1).to(DEV), torch.from_numpy(z1).to(DEV), torch.from_numpy(f1).to(DEV), 96, 96).pix_to_face.cpu().numpy()
        assert int((r1 >= 0).sum()) > 500 and np.array_equal(p1, r1), (seed, int((r1 >= 0).sum()), int((p1 != r1).sum()))


def test_rasterisers_edge_cases():
    from selfreconcode_amd.ops import rasterize_meshes, points_silhouette
    cam = _camera(32, 32)
    xy = torch.zeros(1, 0, 2, device=DEV); z = torch.zeros(1, 0, device=DEV)
    assert float(points_silhouette(xy, z, 32, 32, 0.05, 50).abs().max()) == 0.0            # empty cloud
    fr = rasterize_meshes(xy, z, torch.zeros(0, 3, dtype=torch.int64, device=DEV), 32, 32)  # empty mesh
    assert int((fr.pix_to_face >= 0).sum()) == 0
    # a point exactly on a pixel centre (a = 1): finite value and gradient
    px = torch.tensor([[[1.0 - (2 * 10 + 1) / 32.0, 1.0 - (2 * 7 + 1) / 32.0]]], device=DEV, requires_grad=True)
    m = points_silhouette(px, torch.ones(1, 1, device=DEV), 32, 32, 0.05, 50)
    assert abs(float(m[0, 7, 10]) - 1.0) < 1e-5
    (g,) = torch.autograd.grad(m.sum(), px)
    assert torch.isfinite(g).all()


def test_mesh_rasteriser_is_reproducible_next_to_the_layer_gemms():
    """The hard mesh rasteriser is a pure function of its inputs (integer keys, order-independent atomicMin): called 150 times on one
    mesh on a high-priority side stream WHILE 196,608-row fp32 layer GEMMs run on the main stream -- the way the ray selection runs
    under the template branch -- it must give the same pix_to_face every time."""
    from selfreconcode_amd import mlp_engine as me
    from selfreconcode_amd.ops import rasterize_meshes
    g = torch.Generator(device="cpu").manual_seed(5)
    H = W = 540
    n = 260                                           # a wavy sheet of n x n vertices: ~135k small faces, 1-4 pixel centres each, two layers in depth
    u = torch.linspace(-0.9, 0.9, n)
    gx, gy = torch.meshgrid(u, u, indexing="ij")
    jit = (torch.rand(n, n, 2, generator=g) - 0.5) * (0.6 * 1.8 / n)
    xy = torch.stack([gx, gy], -1) + jit
    z = 2.0 + 0.3 * torch.sin(5 * gx) * torch.cos(4 * gy)
    ii = torch.arange(n - 1)
    a = (ii[:, None] * n + ii[None, :]).reshape(-1)
    faces = torch.cat([torch.stack([a, a + 1, a + n], 1), torch.stack([a + 1, a + n + 1, a + n], 1)], 0)
    xy2 = torch.cat([xy.reshape(-1, 2), xy.reshape(-1, 2) * 0.8 + 0.03], 0)                 # a second, nearer sheet over the middle
    z2 = torch.cat([z.reshape(-1), z.reshape(-1) - 0.5], 0)
    faces2 = torch.cat([faces, faces + n * n], 0)
    xy_d, z_d, f_d = xy2[None].to(DEV).contiguous(), z2[None].to(DEV).contiguous(), faces2.to(DEV).contiguous()
    M, N, K = 196608, 512, 512
    A = (torch.randn(M, K, device=DEV) * 0.3).contiguous(); B = (torch.randn(N, K, device=DEV) * 0.05).contiguous()
    C = torch.zeros(M, N, device=DEV); bias = torch.zeros(N, device=DEV)
    side = torch.cuda.Stream(priority=-1)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ref = rasterize_meshes(xy_d, z_d, f_d, H, W).pix_to_face.clone()
    torch.cuda.synchronize()
    assert int((ref >= 0).sum()) > 100000
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    for _ in range(150):
        for _ in range(3):
            me._gemm_nt(A, K, B, K, C, N, M, N, K, bias, 1, me.ACT_NONE, me.EPI_FWD)
        with torch.cuda.stream(side):
            bad += (rasterize_meshes(xy_d, z_d, f_d, H, W).pix_to_face != ref).sum()
    torch.cuda.synchronize()
    assert int(bad) == 0, int(bad)
