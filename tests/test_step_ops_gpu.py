"""Fused step tails (csrc/step_ops.hip) against the reference's formulations evaluated with torch in float64:
values and every gradient.  The float64 expressions below restate the cited reference lines; the whole-iteration parity
test (test_iteration_parity_gpu.py) covers the same ops in place against oracle/iteration_oracle.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(a, b, rtol=2e-5, atol=None):
    b = b.to(a.dtype) if a.dtype != b.dtype else b
    a64, b64 = a.double(), b.double()
    assert a64.shape == b64.shape, (a64.shape, b64.shape)
    if a64.numel() == 0:
        return
    atol = (atol if atol is not None else 2e-6 * max(1.0, b64.abs().max().item()))
    assert torch.allclose(a64, b64, rtol=rtol, atol=atol), (a64 - b64).abs().max().item()


def _camera(seed, learn=True):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(4, generator=g); q = q / q.norm()
    w, x, y, z = q.tolist()
    R = torch.tensor([[w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z],
                      [2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x],
                      [2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z]])
    T = torch.tensor([0.05, -0.1, 3.0])
    f = torch.tensor([540., 560.]); c = torch.tensor([270., 262.])
    out = [t.to(DEV).requires_grad_(learn) for t in (R, T, f, c)]
    return out


def _project_ref(ps, R, T, f, c, W, H):
    """model/CameraMine.py:44-70,171-262 (screen-space intrinsics -> NDC), float64."""
    pc = ps.matmul(R) + T.view(1, 3)
    x = (f[0] / (W / 2.)) * pc[..., 0] / pc[..., 2] + (1. - 1. / W - c[0] / (W / 2.))
    y = (f[1] / (H / 2.)) * pc[..., 1] / pc[..., 2] + (1. - 1. / H - c[1] / (H / 2.))
    return torch.stack([x, y], dim=-1), pc[..., 2]


@pytest.mark.parametrize("shape", [(3, 2500, 3), (70001, 3), (0, 3)])
def test_project_ndc_value_and_all_gradients(shape):
    from selfreconcode_amd.step_ops import ProjectNDC
    torch.manual_seed(0)
    R, T, f, c = _camera(1)
    ps = (torch.randn(shape, device=DEV) * 0.4).requires_grad_(True)
    W, H = 540., 520.
    xy, z = ProjectNDC.apply(ps, R, T, f, c, W, H)
    d = [t.detach().double().requires_grad_(True) for t in (ps, R, T, f, c)]
    xy_r, z_r = _project_ref(d[0], d[1], d[2], d[3], d[4], W, H)
    _close(xy, xy_r); _close(z, z_r)
    if ps.numel() == 0:
        return
    gxy, gz = torch.randn_like(xy), torch.randn_like(z)
    grads = torch.autograd.grad([xy, z], [ps, R, T, f, c], [gxy, gz])
    grads_r = torch.autograd.grad([xy_r, z_r], d, [gxy.double(), gz.double()])
    for a, b in zip(grads, grads_r):
        _close(a, b, rtol=2e-4, atol=2e-4 * b.abs().max().item())
    # xy cotangent only (the silhouette path), parameters fixed
    xy2, _ = ProjectNDC.apply(ps, R.detach(), T.detach(), f.detach(), c.detach(), W, H)
    g2, = torch.autograd.grad(xy2, ps, gxy)
    g2r, = torch.autograd.grad(_project_ref(d[0], d[1], d[2], d[3], d[4], W, H)[0], d[0], gxy.double())
    _close(g2, g2r, rtol=2e-4, atol=2e-4 * g2r.abs().max().item())


def test_view_rays_value_and_parameter_gradients():
    from selfreconcode_amd.step_ops import ViewRays
    torch.manual_seed(1)
    R, T, f, c = _camera(2)
    P = 6001
    px = torch.stack([torch.randint(0, 540, (P,)), torch.randint(0, 520, (P,)), torch.ones(P, dtype=torch.long)], dim=-1).float().to(DEV)
    rays = ViewRays.apply(px, R, f, c)
    Rd, fd, cd = [t.detach().double().requires_grad_(True) for t in (R, f, c)]
    ps = px.double()
    raw = torch.stack([-ps[:, 0] / fd[0] + ps[:, 2] * cd[0] / fd[0], -ps[:, 1] / fd[1] + ps[:, 2] * cd[1] / fd[1], ps[:, 2]], dim=1)   # CameraMine.py:129-143
    ref = (raw / raw.norm(dim=1, keepdim=True)).matmul(Rd.t())
    _close(rays, ref)
    g = torch.randn_like(rays)
    got = torch.autograd.grad(rays, [R, f, c], g)
    want = torch.autograd.grad(ref, [Rd, fd, cd], g.double())
    for a, b in zip(got, want):
        _close(a, b, rtol=2e-4, atol=2e-4 * b.abs().max().item())


def _jacobians(P, singular_rows=()):
    J = torch.eye(3, device=DEV).expand(P, 3, 3) + 0.25 * torch.randn(P, 3, 3, device=DEV)
    for r in singular_rows:
        J[r, 2] = J[r, 0] * 2.0
    return J.contiguous()


def test_cardinal_rays_and_deformed_normals_vs_reference_lines():
    from selfreconcode_amd.step_ops import CardinalRays, deformed_normals
    torch.manual_seed(2)
    P = 5000
    J = _jacobians(P, singular_rows=(7, 4999)).requires_grad_(True)
    v = torch.nn.functional.normalize(torch.randn(P, 3, device=DEV), dim=1).requires_grad_(True)
    out, ok = CardinalRays.apply(J, v)
    assert not ok[7] and not ok[4999] and int(ok.sum()) == P - 2
    Jd, vd = J.detach().double().requires_grad_(True), v.detach().double().requires_grad_(True)
    inv = torch.linalg.inv(torch.where(ok[:, None, None], Jd, torch.eye(3, device=DEV, dtype=torch.float64)))
    cr = (inv * vd.unsqueeze(-2)).sum(-1)                                   # utils/utils.py:155-169
    cr = torch.where(ok[:, None], cr, vd.detach())
    ref = cr / cr.norm(dim=1, keepdim=True)
    _close(out, ref, rtol=1e-4, atol=1e-5)
    g = torch.randn_like(out)
    gJ, gv = torch.autograd.grad(out, [J, v], g)
    gJr, gvr = torch.autograd.grad(ref, [Jd, vd], g.double())
    _close(gJ, gJr, rtol=1e-3, atol=1e-4 * gJr.abs().max().item())
    _close(gv, gvr, rtol=1e-3, atol=1e-4 * gvr.abs().max().item())
    onx = torch.randn(P, 3, device=DEV)
    nx = deformed_normals(J.detach(), onx)
    n_ref = (inv.transpose(-2, -1) * onx.double().unsqueeze(-2)).sum(-1)     # utils/utils.py:132-153
    n_ref = torch.where(ok[:, None], n_ref, (Jd.detach() * onx.double().unsqueeze(-2)).sum(-1))
    _close(nx, n_ref / n_ref.norm(dim=1, keepdim=True), rtol=1e-4, atol=1e-5)


def _rays_pixels(P, N, H, W, sort=True):
    b = torch.randint(0, N, (P,), device=DEV)
    if sort:
        b = b.sort().values
    return b, torch.randint(0, H, (P,), device=DEV), torch.randint(0, W, (P,), device=DEV)


def _scatter_mean(vals, index, n):
    s = torch.zeros(n, dtype=vals.dtype, device=vals.device).index_add(0, index, vals)
    c = torch.zeros(n, dtype=vals.dtype, device=vals.device).index_add(0, index, torch.ones_like(vals))
    return s / c.clamp(min=1)


@pytest.mark.parametrize("P,N", [(6144, 3), (1, 1), (40000, 8), (100, 4)])
def test_color_loss(P, N):
    from selfreconcode_amd.step_ops import ColorLoss
    torch.manual_seed(3)
    H, W = 64, 48
    b, r, c = _rays_pixels(P, N, H, W, sort=(P != 100))
    if P == 100:
        b = b.clamp(max=1)                       # frames 2, 3 have no ray: empty bins contribute 0 (torch_scatter semantics)
    gt = torch.rand(N, H, W, 3, device=DEV) * 2 - 1
    col = (torch.rand(P, 3, device=DEV) * 2 - 1).requires_grad_(True)
    loss = ColorLoss.apply(col, gt, b, r, c)
    cd = col.detach().double().requires_grad_(True)
    ref = _scatter_mean((gt.double()[b, r, c, :] - cd).abs().sum(1), b, N).mean()      # model/network.py:611-618
    _close(loss, ref, rtol=1e-5, atol=1e-6)
    g, = torch.autograd.grad(loss * 0.7, col)
    gr, = torch.autograd.grad(ref * 0.7, cd)
    _close(g, gr, rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("weighted", [False, True])
def test_normal_loss(weighted):
    from selfreconcode_amd.step_ops import NormalLoss
    torch.manual_seed(4)
    P, N, H, W = 6000, 3, 40, 56
    b, r, c = _rays_pixels(P, N, H, W)
    gtn = torch.randn(N, H, W, 3, device=DEV)
    gtn[:, ::3] = 0.                                              # background pixels: invalid rows
    J = _jacobians(P, singular_rows=(11,)).requires_grad_(True)
    nx_raw = (torch.randn(P, 3, device=DEV) * 1.3).requires_grad_(True)
    rays = torch.nn.functional.normalize(torch.randn(P, 3, device=DEV), dim=1)
    R = _camera(5, learn=False)[0]
    loss = NormalLoss.apply(nx_raw, J, gtn, R, rays, weighted, b, r, c)
    # model/network.py:620-639 in float64
    Jd, nd = J.detach().double().requires_grad_(True), nx_raw.detach().double().requires_grad_(True)
    nx = nd / nd.norm(dim=1, keepdim=True)
    if weighted:
        det = torch.linalg.det(Jd.detach())
        ok = det.abs() >= 1e-4
        inv = torch.linalg.inv(torch.where(ok[:, None, None], Jd.detach(), torch.eye(3, device=DEV, dtype=torch.float64)))
        cn = torch.where(ok[:, None], (inv.transpose(-2, -1) * nd.detach().unsqueeze(-2)).sum(-1), (Jd.detach() * nd.detach().unsqueeze(-2)).sum(-1))
        cn = cn / cn.norm(dim=1, keepdim=True)
        wts = torch.clamp((-rays.double() * cn).sum(1), max=1., min=0.) ** 2
    else:
        wts = torch.ones(P, device=DEV, dtype=torch.float64)
    flip = torch.tensor([[-1., 0., 0.], [0., 1., 0.], [0., 0., -1.]], device=DEV, dtype=torch.float64)
    g = gtn.double()[b, r, c, :].view(-1, 3) @ (R.double() @ flip).t()
    gn = g.norm(dim=1, keepdim=True)
    valid = (gn > 0.0001)[..., 0]
    g = torch.where(valid[:, None], g / gn.clamp(min=1e-12), g)
    g = (Jd.transpose(-2, -1) * g.unsqueeze(-2)).sum(-1)
    per = (g - nx).norm(2, dim=1) * wts
    ref = _scatter_mean_masked(per, b, valid, N)
    _close(loss, ref, rtol=2e-5, atol=1e-6)
    got = torch.autograd.grad(loss, [nx_raw, J])
    want = torch.autograd.grad(ref, [nd, Jd])
    for a, w in zip(got, want):
        _close(a, w, rtol=1e-3, atol=2e-5 * w.abs().max().item())


def _scatter_mean_masked(per, b, valid, N):
    ssum = torch.zeros(N, dtype=per.dtype, device=per.device).index_add(0, b, torch.where(valid, per, torch.zeros((), dtype=per.dtype, device=per.device)))
    scnt = torch.zeros(N, dtype=per.dtype, device=per.device).index_add(0, b, valid.to(per.dtype))
    return (ssum / scnt.clamp(min=1)).mean()


@pytest.mark.parametrize("n", [1, 11947, 70000])
def test_eikonal_loss(n):
    from selfreconcode_amd.step_ops import EikonalLoss
    torch.manual_seed(5)
    g = (torch.randn(n, 3, device=DEV) * 0.8).requires_grad_(True)
    loss = EikonalLoss.apply(g)
    gd = g.detach().double().requires_grad_(True)
    ref = ((gd.norm(2, dim=-1) - 1) ** 2).mean()                   # model/network.py:547-549
    _close(loss, ref, rtol=1e-5, atol=1e-7)
    a, = torch.autograd.grad(loss * 0.1, g)
    w, = torch.autograd.grad(ref * 0.1, gd)
    _close(a, w, rtol=1e-4, atol=1e-10)


@pytest.mark.parametrize("n", [5, 61440])
def test_def_regu_loss_matches_svd_formulation(n):
    from selfreconcode_amd.step_ops import DefReguLoss
    torch.manual_seed(6)
    J = (torch.eye(3, device=DEV).expand(n, 3, 3) + 0.2 * torch.randn(n, 3, 3, device=DEV)).contiguous().requires_grad_(True)
    c = 0.5
    loss = DefReguLoss.apply(J, c)
    Jd = J.detach().double().requires_grad_(True)
    s = torch.log(torch.linalg.svdvals(Jd))                       # model/network.py:576-579 (torch.svd on the CPU there)
    x = (s * s).sum(1)
    ref = (2. * x / (c * c) / (x / (c * c) + 4)).mean()            # utils/utils.py:48-52, square=True
    _close(loss, ref, rtol=2e-4, atol=1e-7)
    a, = torch.autograd.grad(loss, J)
    w, = torch.autograd.grad(ref, Jd)
    _close(a, w, rtol=5e-3, atol=2e-4 * w.abs().max().item())


@pytest.mark.parametrize("N,H,W", [(3, 540, 540), (1, 33, 17), (8, 64, 64)])
def test_mask_iou_loss(N, H, W):
    from selfreconcode_amd.step_ops import MaskIoULoss
    torch.manual_seed(7)
    m = torch.rand(N, H, W, device=DEV).requires_grad_(True)
    g = (torch.rand(N, H, W, device=DEV) > 0.6).float()
    loss = MaskIoULoss.apply(m, g)
    md, gd = m.detach().double().requires_grad_(True), g.double()
    ref = (1. - (md * gd).view(N, -1).sum(1) / (md + gd - md * gd).abs().view(N, -1).sum(1)).mean()     # model/network.py:652-654
    _close(loss, ref, rtol=2e-5, atol=1e-7)
    a, = torch.autograd.grad(loss * 3., m)
    w, = torch.autograd.grad(ref * 3., md)
    _close(a, w, rtol=1e-3, atol=1e-5 * w.abs().max().item())


def test_implicit_solve_vs_reference_lines():
    from selfreconcode_amd.step_ops import implicit_solve
    torch.manual_seed(8)
    P = 5003
    gf = torch.randn(P, 3, device=DEV); J = _jacobians(P); gl = torch.randn(P, 3, device=DEV)
    v = torch.nn.functional.normalize(torch.randn(P, 3, device=DEV), dim=1)
    gf[3] = 0.; J[3] = 0.                                          # singular normal equations: zeros, ok = False (FastMinv's rule)
    cot_f, tail, temp, ok = implicit_solve(gf, J, v, gl)
    z = torch.zeros(P, device=DEV, dtype=torch.float64)
    vd = v.double()
    vx = torch.stack([z, -vd[:, 2], vd[:, 1], vd[:, 2], z, -vd[:, 0], -vd[:, 1], vd[:, 0], z], dim=1).view(-1, 3, 3)   # model/network.py:757-764
    b = torch.cat([gf.double().view(-1, 1, 3), vx @ J.double()], dim=1)
    btb = b.permute(0, 2, 1) @ b
    det = torch.linalg.det(btb)
    ok_ref = det.abs() >= 1e-4                                     # FastMinv/Matrix3x3InvKernels.cu: |det| < 1e-4 -> zeros, False
    clear = (det.abs() - 1e-4).abs() > 1e-6                        # (rows within float32 rounding of the threshold may flip)
    assert not ok[3] and torch.equal(ok[clear], ok_ref[clear]) and int(ok.sum()) > P // 2
    use = ok & ok_ref
    inv = torch.linalg.inv(torch.where(use[:, None, None], btb, torch.eye(3, device=DEV, dtype=torch.float64)))
    rhs = gl.double().view(-1, 1, 3) @ (inv @ b.permute(0, 2, 1))          # [P,1,4]
    t_ref = (rhs[:, :, 1:] @ (-vx)).view(-1, 3)
    # per-row tolerance: float32 adjugate inverse of a system with condition number k loses ~k eps
    cond = torch.linalg.cond(btb).clamp(max=1e12)
    tol = (4e-6 * cond * rhs.abs().amax(dim=(1, 2)).clamp(min=1e-3))[use]
    assert ((cot_f.double() + rhs[:, 0, 0]).abs()[use] <= tol).all()
    assert ((tail.double() - rhs[:, 0, 1:]).abs().amax(1)[use] <= tol).all()
    assert ((temp.double() - t_ref).abs().amax(1)[use] <= tol).all()
    assert float(cot_f[~ok].abs().max()) == 0.0 and float(temp[~ok].abs().max()) == 0.0


def test_training_step_same_with_and_without_fused_tails():
    """One full iteration with the fused loss tails / cardinal rays / implicit solve against the composite torch formulations
    (same weights, same random draws; the camera ops stay fused on both sides so that both runs select and refine the same
    rays -- they have their own tests above): every loss term, the total and every parameter gradient."""
    from selfreconcode_amd import step_ops, mlp_engine
    from selfreconcode_amd.synthetic import build_synthetic_scene
    res = {}
    for flag in (True, False):
        step_ops.ENABLED = flag
        try:
            net, ds, conf = build_synthetic_scene(device=DEV, frame_num=40, H=96, W=96, resolutions=[(15, 21, 9), (29, 41, 17)],
                                                  lbs_volume_shape=(17, 57, 33), consistent_masks=False)
            net.point_radius = 0.03
            with torch.no_grad():
                net.deformer.defs[0].lin4.weight.mul_(20.0)
            mlp_engine.set_deferred_param_grads(True)
            fids = torch.tensor([5, 17, 30], device=DEV)
            datas = ds.batch(fids)
            ratio = {'sdfRatio': 1., 'deformerRatio': 0.6, 'renderRatio': 1.}
            g = torch.Generator(device=DEV).manual_seed(5)
            rand = {k: torch.rand(100000, device=DEV, generator=g) for k in ('ray_select', 'vert_select', 'vert_select2')}
            rand.update(eik_local=torch.randn(20000, 3, device=DEV, generator=g), eik_global=torch.rand(4000, 3, device=DEV, generator=g),
                        regu_local=torch.randn(20000, 3, device=DEV, generator=g))
            loss = net(datas, 2048, ratio, fids, rand=rand)
            loss.backward()
            net.propagateTmpPsGrad(fids, ratio)
            params = list(ds.learnable_weights()) + [p for p in net.parameters() if p.requires_grad]
            terms = {k: float(v) for k, v in net.info.items() if k.endswith('_loss') and k != 'pc_loss'}
            terms['mask_loss'] = float(net.info['pc_loss']['mask_loss'])
            res[flag] = (float(loss), [None if p.grad is None else p.grad.detach().double().clone() for p in params], terms,
                         tuple(int(x) for x in net.info['rayInfo']))
        finally:
            mlp_engine.set_deferred_param_grads(False)
            step_ops.ENABLED = True
    (la, ga, ta, ra), (lb, gb, tb, rb) = res[True], res[False]
    assert ra == rb and ra[1] > 50, (ra, rb)
    for k in tb:
        assert abs(ta[k] - tb[k]) <= 1e-5 * max(1.0, abs(tb[k])), (k, ta[k], tb[k])
    assert abs(la - lb) <= 1e-5 * max(1.0, abs(lb)), (la, lb)
    n_checked = 0
    for a, b in zip(ga, gb):
        assert (a is None) == (b is None)
        if a is not None:
            assert (a - b).abs().max().item() <= 1e-3 * max(b.abs().max().item(), 1e-7), ((a - b).abs().max().item(), b.abs().max().item())
            n_checked += 1
    assert n_checked > 40
