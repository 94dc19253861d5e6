"""GPU parity of the loss assembly (a14) and of the implicit-gradient propagation (a15) against oracle
restatements of model/network.py:543-639 and :702-814 built from oracle/torch_oracle.py, plus the
equivalence of the deferred-gradient fast path with plain autograd over one full training iteration."""
import numpy as np
import pytest
import torch
from oracle import torch_oracle as orc
from oracle import fixtures as fx

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RATIO = {'sdfRatio': 1.0, 'deformerRatio': 0.62, 'renderRatio': 1.0}


def close(a, b, rtol=1e-3, atol=None):
    b = b.detach().float()
    atol = atol if atol is not None else 3e-4 * max(1e-3, float(b.abs().max()))
    torch.testing.assert_close(a.detach().float().cpu(), b, rtol=rtol, atol=atol)


class _MiniData:
    def __init__(self, F=40):
        self.poses = (fx.det_tensor((F, 24, 3), 1, 0.1)).to(DEV).requires_grad_(True)
        self.trans = (fx.det_tensor((F, 3), 2, 0.05)).to(DEV).requires_grad_(True)
        self.conds = [fx.det_tensor((F, 128), 3, 0.1).to(DEV).requires_grad_(True), fx.det_tensor((F, 256), 4, 0.1).to(DEV).requires_grad_(True)]
        self.frame_num = F

    def get_grad_parameters(self, idxs, device=None):
        return self.poses[idxs], self.trans[idxs], self.conds[0][idxs], self.conds[1][idxs]

    def get_camera_parameters(self, N, device=None):
        R = orc.quat2mat(torch.tensor([[0.02, 0.01, 0.999, 0.03]]))[0].to(DEV)
        return (torch.tensor([[150., 150.]], device=DEV).expand(N, 2), torch.tensor([[64., 64.]], device=DEV).expand(N, 2),
                R[None].expand(N, 3, 3), torch.tensor([[0., 0.1, 2.4]], device=DEV).expand(N, 3), 128, 128)

    def get_batchframe_data(self, name, fids, n):
        data = getattr(self, name)
        starts = (fids - n // 2).clamp(min=0, max=self.frame_num - n)
        return data[starts.view(-1, 1) + torch.arange(0, n, device=fids.device).view(1, n)], fids - starts


def _build():
    from selfreconcode_amd.config import default_config
    from selfreconcode_amd.model.network import getTmpSdf
    from selfreconcode_amd.model.Deformer import MLPTranslator, LBSkinner, CompositeDeformer
    from selfreconcode_amd.model.RenderNet import RenderingNetwork_view_norm
    from selfreconcode_amd.model.optim_network import OptimNetwork
    from selfreconcode_amd.utils import smpl_tmp_Apose, DCTNullSpace
    sdf = getTmpSdf(DEV, 6, 0.6, 256); sdf.load_state_dict(fx.det_params(fx.SDF_SPEC, 101), strict=True)
    tr = MLPTranslator(128, 6).to(DEV); tr.load_state_dict(fx.det_params(fx.DEF_SPEC, 202, last_scale=0.3), strict=True)
    skin = LBSkinner(fx.synthetic_lbs_volume((7, 11, 9)), fx.LBS_BMIN, fx.LBS_BMAX, fx.synthetic_joints(), np.array(fx.SMPL_PARENTS),
                     init_pose=torch.from_numpy(smpl_tmp_Apose(1)), align_corners=False).to(DEV)
    rn = RenderingNetwork_view_norm(256, 'idr', 9, 3, [512] * 4, True, multires_n=0, multires_v=4).to(DEV)
    rn.load_state_dict(fx.det_params(fx.REND_SPEC, 303), strict=True)
    net = OptimNetwork(sdf, CompositeDeformer([tr, skin]).to(DEV), None, None, rn, conf=default_config().get_config('loss_coarse')).to(DEV)
    net.dataset = _MiniData()
    net.dctnull = DCTNullSpace(10, 30).to(DEV)
    return net


def _oracle_params():
    sdf = {k: v.clone().requires_grad_(True) for k, v in fx.det_params(fx.SDF_SPEC, 101).items()}
    trp = {k: v.clone().requires_grad_(True) for k, v in fx.det_params(fx.DEF_SPEC, 202, last_scale=0.3).items()}
    rnd = {k: v.clone().requires_grad_(True) for k, v in fx.det_params(fx.REND_SPEC, 303).items()}
    return sdf, trp, rnd


def _oracle_deformer(net, trp, conds, poses, trans):
    skin = net.deformer.defs[1]
    kw = dict(ws=fx.synthetic_lbs_volume((7, 11, 9)), b_min=torch.tensor(fx.LBS_BMIN), b_max=torch.tensor(fx.LBS_BMAX),
              Js=fx.synthetic_joints(), init_pose=skin.init_pose.cpu())

    def f(p, bi):
        q, _ = orc.translator_forward(trp, p, conds, bi, RATIO)
        return orc.lbs_forward(q, poses, trans, batch_inds=bi, **kw)
    return f


def test_def_regu_and_dct_losses_vs_oracle():
    net = _build()
    sdf_o, trp, _ = _oracle_params()
    N = 2
    fids = torch.tensor([7, 19], device=DEV)
    poses, trans, d_cond, _ = net.dataset.get_grad_parameters(fids)
    pts = fx.det_tensor((25, 3), 5, 0.5)
    noise = fx.det_tensor((25, 3), 6, 1.0)
    loss = net.loss_def_regu(pts.to(DEV), d_cond, N, RATIO, noise.to(DEV))
    g = torch.autograd.grad(loss, [net.deformer.defs[0].lin1.weight, net.dataset.conds[0]])
    # oracle (network.py:565-582)
    co = d_cond.detach().cpu().clone().requires_grad_(True)
    po = torch.cat([pts, pts + noise * 0.01], 0).view(1, -1, 3).expand(N, -1, 3).contiguous().requires_grad_(True)
    do, _ = orc.translator_forward(trp, po, co, None, RATIO)
    Jo = orc.compute_jacobian(po, do, True, True)
    so = torch.log(torch.linalg.svdvals(Jo))
    lo = orc.gm_robust((so * so).sum(1), 0.5, True).mean()
    go = torch.autograd.grad(lo, [trp["lin1.weight"], co])
    close(loss, lo, 1e-4, 1e-7)
    close(g[0], go[0]); close(g[1][fids], go[1])
    # DCT term (network.py:585-593)
    dl = net.loss_dct(fids, N)
    gd = torch.autograd.grad(dl, [net.dataset.poses, net.dataset.trans], allow_unused=True)
    pso = net.dataset.poses.detach().cpu().clone().requires_grad_(True); tso = net.dataset.trans.detach().cpu().clone().requires_grad_(True)
    starts = (fids.cpu() - 15).clamp(min=0, max=40 - 30)
    idx = starts.view(-1, 1) + torch.arange(30).view(1, 30)
    _, newJ = orc.lbs_transforms(pso[idx].reshape(N * 30, 24, 3), fx.synthetic_joints(), net.deformer.defs[1].init_pose.cpu())
    dlo = (orc.dct_null_space(10, 30)[None] @ newJ.reshape(N, 30, 72)).abs().mean()
    close(dl, dlo, 1e-4, 1e-7)
    close(gd[0], torch.autograd.grad(dlo, pso)[0])
    assert gd[1] is None                   # posedSkeleton does not use trans (Deformer.py:144-165)


def test_color_normal_losses_and_propagation_vs_oracle():
    """network.py:599-639 on fixed rays, then propagateTmpPsGrad (:702-814): gradients of every parameter group."""
    from selfreconcode_amd.model.CameraMine import RectifiedPerspectiveCameras
    net = _build()
    sdf_o, trp, rnd = _oracle_params()
    N, P = 2, 24
    fids = torch.tensor([7, 19], device=DEV)
    poses, trans, d_cond, rendcond = net.dataset.get_grad_parameters(fids)
    defconds = [d_cond, [poses, trans]]
    R = orc.quat2mat(torch.tensor([[0.02, 0.01, 0.999, 0.03]]))[0]
    cam = RectifiedPerspectiveCameras(torch.tensor([[150., 150.]]), torch.tensor([[64., 64.]]), R[None], torch.tensor([[0., 0.1, 2.4]]), [(128, 128)]).to(DEV)
    p = fx.det_tensor((P, 3), 8, 0.3) * torch.tensor([0.7, 1.0, 0.3])
    rays = torch.nn.functional.normalize(fx.det_tensor((P, 3), 9, 1.0), dim=1)
    bi = torch.arange(P) % N
    rows, cols = torch.arange(P) % 16, (torch.arange(P) * 3) % 16
    gtC = fx.det_tensor((N, 16, 16, 3), 10, 1.0); gtN = fx.det_tensor((N, 16, 16, 3), 11, 1.0)
    net.TmpPs = p.to(DEV).requires_grad_(True); net.rays = rays.to(DEV); net.batch_inds = bi.to(DEV)
    net.row_inds, net.col_inds = rows.to(DEV), cols.to(DEV)
    total = net.loss_color_normal({'normal': gtN.to(DEV)}, gtC.to(DEV), cam, defconds, rendcond, RATIO, N)
    total.backward()
    from selfreconcode_amd import mlp_engine
    mlp_engine.flush_param_grads()
    # ---- oracle restatement
    co = d_cond.detach().cpu().clone().requires_grad_(True); pso = poses.detach().cpu().clone().requires_grad_(True); tso = trans.detach().cpu().clone().requires_grad_(True)
    dfn = _oracle_deformer(net, trp, co, pso, tso)
    po = p.clone().requires_grad_(True)
    y, feat = orc.sdf_forward(sdf_o, po, RATIO)
    nx = torch.autograd.grad(y, po, torch.ones_like(y), create_graph=True)[0]
    nx = nx / nx.norm(dim=1, keepdim=True)
    ds = dfn(po, bi); J = orc.compute_jacobian(po, ds, True, True)
    Ji, ok = orc.DiffMinv.apply(J); assert ok.all()
    cr = (Ji @ rays.view(-1, 3, 1)).view(-1, 3); cr = cr / cr.norm(dim=1, keepdim=True)
    col = orc.render_forward(rnd, po, nx, cr, feat, RATIO)
    closs = orc.scatter_mean((gtC[bi, rows, cols] - col).abs().sum(1), bi, N).mean()
    y2, _ = orc.sdf_forward(sdf_o, po, RATIO)
    onx = torch.autograd.grad(y2, po, torch.ones_like(y2))[0]
    ds2 = dfn(po, bi); J2 = orc.compute_jacobian(po, ds2, False, False); Ji2, _ = orc.minv3x3(J2)
    cnx = (Ji2.transpose(-2, -1) @ onx.view(-1, 3, 1)).view(-1, 3); cnx = cnx / cnx.norm(dim=1, keepdim=True)
    w = torch.clamp((-rays * cnx.detach()).sum(1), max=1., min=0.) ** 2
    flip = torch.tensor([[-1., 0., 0.], [0., 1., 0.], [0., 0., -1.]])
    gn = ((R @ flip) @ gtN[bi, rows, cols].view(-1, 3, 1)).view(-1, 3)
    gn = gn / gn.norm(dim=1, keepdim=True)
    ds3 = dfn(po, bi); J3 = orc.compute_jacobian(po, ds3, True, True)
    gn = (J3.transpose(-2, -1) @ gn.view(-1, 3, 1)).view(-1, 3)
    nloss = orc.scatter_mean((gn - nx).norm(2, dim=1) * w, bi, N).mean()
    lo = 0.5 * closs + 0.1 * nloss
    lo.backward()
    close(total, lo, 1e-4, 1e-6)
    close(net.TmpPs.grad, po.grad)
    close(net.sdf.lin5.weight_v.grad, sdf_o["lin5.weight_v"].grad); close(net.netRender.lin1.weight_g.grad, rnd["lin1.weight_g"].grad)
    close(net.deformer.defs[0].lin2.weight.grad, trp["lin2.weight"].grad)
    close(net.dataset.poses.grad[fids], pso.grad); close(net.dataset.conds[0].grad[fids], co.grad)
    # ---- propagateTmpPsGrad: oracle restatement of network.py:702-814 (cameras fixed)
    for t in list(net.parameters()) + [net.dataset.poses, net.dataset.trans] + net.dataset.conds:
        t.grad = None
    for d_ in (sdf_o, trp):
        for t in d_.values():
            t.grad = None
    co.grad = pso.grad = tso.grad = None
    net.propagateTmpPsGrad(fids, RATIO)
    glp = po.grad.clone()
    pd = p.clone().requires_grad_(True)
    f = orc.sdf_forward(sdf_o, pd, RATIO)[0]
    gfp = torch.autograd.grad(f, pd, torch.ones_like(f))[0]
    d = dfn(pd, bi); Jd = orc.compute_jacobian(pd, d, False, False)
    vx = torch.zeros(P, 3, 3)
    vx[:, 0, 1], vx[:, 0, 2], vx[:, 1, 0], vx[:, 1, 2], vx[:, 2, 0], vx[:, 2, 1] = -rays[:, 2], rays[:, 1], rays[:, 2], -rays[:, 0], -rays[:, 1], rays[:, 0]
    b = torch.cat([gfp.view(-1, 1, 3), vx @ Jd], 1)
    binv, ok = orc.minv3x3(b.permute(0, 2, 1) @ b)
    rhs = glp.view(-1, 1, 3) @ (binv @ b.permute(0, 2, 1))
    f2 = orc.sdf_forward(sdf_o, p, RATIO)[0]
    d2 = dfn(p, bi)
    torch.autograd.backward([f2, d2], [(-rhs[:, :, 0]).reshape(f2.shape), (rhs[:, :, -3:] @ (-vx)).view(-1, 3)])
    assert int(net.info['invInfo'][1]) == int(ok.sum())
    close(net.sdf.lin2.weight_v.grad, sdf_o["lin2.weight_v"].grad); close(net.sdf.lin8.bias.grad, sdf_o["lin8.bias"].grad)
    close(net.deformer.defs[0].lin0.weight.grad, trp["lin0.weight"].grad)
    # per-frame pose / translation / code gradients are sums over every ray of the frame with heavy cancellation: fp32
    # summation order (wave-level partial sums on the GPU, sequential on the CPU oracle) moves them by a few 1e-4 of the
    # largest entry
    def close_sum(a, b):
        close(a, b, rtol=1e-3, atol=1e-3 * max(1e-3, float(b.detach().abs().max())))
    close_sum(net.dataset.poses.grad[fids], pso.grad); close_sum(net.dataset.trans.grad[fids], tso.grad); close_sum(net.dataset.conds[0].grad[fids], co.grad)


def test_deferred_gradients_equal_plain_autograd_over_a_full_iteration():
    from selfreconcode_amd import mlp_engine
    from selfreconcode_amd.synthetic import build_synthetic_scene
    grads = []
    for deferred in (False, True):
        mlp_engine.set_deferred_param_grads(deferred)
        net, ds, conf = build_synthetic_scene(device=DEV, frame_num=40, H=128, W=128, resolutions=[(15, 21, 9), (29, 41, 17), (57, 81, 33)],
                                              lbs_volume_shape=(17, 57, 33))
        fids = torch.tensor([3, 11, 20], device=DEV)
        torch.manual_seed(5)
        loss = net(ds.batch(fids), 512, {'sdfRatio': 1., 'deformerRatio': 0.5, 'renderRatio': 1.}, fids)
        loss.backward()
        net.propagateTmpPsGrad(fids, {'sdfRatio': 1., 'deformerRatio': 0.5, 'renderRatio': 1.})
        grads.append(([float(loss)] + [p.grad.clone() for p in net.parameters() if p.requires_grad and p.grad is not None], [t.grad.clone() for t in ds.learnable_weights() if t.grad is not None]))
    mlp_engine.set_deferred_param_grads(False)
    assert abs(grads[0][0][0] - grads[1][0][0]) < 1e-5
    assert len(grads[0][0]) == len(grads[1][0]) > 50
    for a, b in zip(grads[0][0][1:] + grads[0][1], grads[1][0][1:] + grads[1][1]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6 * max(1.0, float(b.abs().max())))


@pytest.mark.parametrize("deferred", [True, False])
def test_masked_ray_branch_equals_the_compacted_one(deferred):
    """`OptimNetwork.masked_ray_branch_below` (round 6): the colour / normal terms and the implicit-gradient pass on ALL selected rays with the
    rays the refiner rejected masked out (frame index -1 in the two loss reductions) against the branch on the compacted list of accepted
    rays, which costs the iteration its second host round trip.  Same loss terms, same gradients of every network parameter, per-frame
    tensor and camera parameter, up to the order of the sums; one round trip instead of two."""
    from selfreconcode_amd import mlp_engine, hostsync
    from selfreconcode_amd.synthetic import build_synthetic_scene
    ratio = {'sdfRatio': 1., 'deformerRatio': 0.5, 'renderRatio': 1.}
    res, trips = [], []
    for masked in (False, True):
        mlp_engine.set_deferred_param_grads(deferred)
        try:
            net, ds, conf = build_synthetic_scene(device=DEV, frame_num=40, H=128, W=128, resolutions=[(15, 21, 9), (29, 41, 17), (57, 81, 33)],
                                                  lbs_volume_shape=(17, 57, 33))
            net.masked_ray_branch_below = 1 << 20 if masked else 0
            fids = torch.tensor([3, 11, 20], device=DEV)
            torch.manual_seed(5)
            seen = []
            hostsync.TRACE = seen.append
            try:
                loss = net(ds.batch(fids), 512, ratio, fids)
            finally:
                hostsync.TRACE = None
            trips.append(seen.count('count on the host'))
            loss.backward()
            net.propagateTmpPsGrad(fids, ratio)
            nrays, nconv = int(net.info['rayInfo'][0]), int(net.info['rayInfo'][1])
            assert 0 < nconv < nrays                                    # the mask has something to do
            assert net.TmpPs.shape[0] == (nrays if masked else nconv)
            if masked:
                assert torch.equal(net.TmpPs.grad[~net.ray_valid], torch.zeros_like(net.TmpPs.grad[~net.ray_valid]))      # exact zeros on the rejected rays
            res.append((float(loss), {k: float(v) for k, v in net.info.items() if torch.is_tensor(v) and v.numel() == 1},
                        [p.grad.clone() for p in net.parameters() if p.requires_grad and p.grad is not None], [t.grad.clone() for t in ds.learnable_weights() if t.grad is not None]))
        finally:
            mlp_engine.set_deferred_param_grads(False)
    assert trips == [2, 1], trips
    (la, ia, ga, da), (lb, ib, gb, db) = res
    assert abs(la - lb) < 2e-6 * max(1.0, abs(la)), (la, lb)
    for k in ia:
        assert abs(ia[k] - ib[k]) < 1e-5 * max(1.0, abs(ia[k])), (k, ia[k], ib[k])
    assert len(ga) == len(gb) > 50 and len(da) == len(db) > 3
    for a, b in zip(ga + da, gb + db):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-6 * max(1.0, float(b.abs().max())))


def test_camera_parameters_receive_gradients_when_learnable():
    """opt_camera (config.conf:12-17): focal length / principal point / T / quaternion learnable -> both the loss graph and
    propagateTmpPsGrad's v- and c-terms (network.py:798-813) must reach them."""
    from selfreconcode_amd.synthetic import build_synthetic_scene
    net, ds, conf = build_synthetic_scene(device=DEV, frame_num=40, H=128, W=128, resolutions=[(15, 21, 9), (29, 41, 17), (57, 81, 33)],
                                          lbs_volume_shape=(17, 57, 33))
    keys = ('focal_length', 'princeple_points', 'world2cam_coord_trans', 'cam2world_coord_quat')
    for k in keys:                                   # incl. the quaternion: its graph (normalise -> R) saves tensors, so the silhouette
        ds.camera_params[k].requires_grad_(True)     # projection back-propagated INSIDE forward() must not share it with the rays
    fids = torch.tensor([3, 11, 20], device=DEV)
    r = {'sdfRatio': 1., 'deformerRatio': 0.5, 'renderRatio': 1.}
    loss = net(ds.batch(fids), 512, r, fids)
    loss.backward()
    g0 = {k: ds.camera_params[k].grad.clone() for k in keys}
    net.propagateTmpPsGrad(fids, r)
    for k, g in g0.items():
        g1 = ds.camera_params[k].grad
        assert torch.isfinite(g1).all() and g1.abs().sum() > 0
    assert not torch.equal(ds.camera_params['world2cam_coord_trans'].grad, g0['world2cam_coord_trans'])   # the c-term added something


def test_initialize_tmp_sdf_prefit_reduces_the_manifold_loss():
    """network.py:207-290: a few epochs of the pre-fit pull |f| on the template surface towards 0."""
    from selfreconcode_amd.synthetic import build_synthetic_scene
    net, ds, conf = build_synthetic_scene(device=DEV, frame_num=40, H=64, W=64, resolutions=[(15, 21, 9), (29, 41, 17)], lbs_volume_shape=(9, 29, 17))
    dirs = torch.nn.functional.normalize(fx.det_tensor((4000, 3), 3, 1.0), dim=1).to(DEV)
    net.tmpBodyVs = dirs * torch.tensor([0.35, 0.5, 0.25], device=DEV)            # an ellipsoid "body"
    net.tmpBodyNs = torch.nn.functional.normalize(dirs / torch.tensor([0.35, 0.5, 0.25], device=DEV), dim=1)
    with torch.no_grad():
        before = net.sdf(net.tmpBodyVs, -1).abs().mean()
    torch.manual_seed(11)                                          # the pre-fit draws its sample points from torch's generator
    last = net.initializeTmpSDF(80, None, with_normals=True)
    with torch.no_grad():
        after = net.sdf(net.tmpBodyVs, -1).abs().mean()
    assert torch.isfinite(last[0]) and float(after) < 0.7 * float(before), (float(before), float(after))


def test_fused_adam_matches_torch_adam():
    """selfreconcode_amd.optim.FusedAdam (one launch per step) against torch.optim.Adam on the same gradients: 25 steps on tensors of
    the sizes the training step has (scalars, [3], [64,24,3], [512,512]), a parameter that sometimes has no gradient, two groups with
    different learning rates, a learning-rate change on the way; and a state_dict round trip in both directions."""
    from selfreconcode_amd.optim import FusedAdam
    shapes = [(2,), (3,), (64, 24, 3), (512, 512), (257, 1), (1,)]
    mk = lambda: [fx.det_tensor(s, 50 + i, 0.3).to(DEV).requires_grad_(True) for i, s in enumerate(shapes)]
    pa, pb = mk(), mk()
    oa = torch.optim.Adam([{'params': pa[:3]}, {'params': pa[3:], 'lr': 3e-4}], lr=1e-3)
    ob = FusedAdam([{'params': pb[:3]}, {'params': pb[3:], 'lr': 3e-4}], lr=1e-3)
    for step in range(25):
        if step == 10:
            for o in (oa, ob):
                o.param_groups[0]['lr'] = 2e-4
        if step == 15:                      # state dict round trip: torch -> fused and fused -> torch
            sa, sb = oa.state_dict(), ob.state_dict()
            oa.load_state_dict(sb); ob.load_state_dict(sa)
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 1 and step % 3 == 0:
                a.grad = None; b.grad = None
                continue
            gk = fx.det_tensor(shapes[i], 1000 + 10 * step + i, 1.0).to(DEV) * (10.0 ** ((step % 5) - 3))
            a.grad = gk.clone(); b.grad = gk.clone()
        oa.step(); ob.step()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(b.detach(), a.detach(), rtol=2e-6, atol=1e-7)
    for a, b in zip(pa, pb):
        sa, sb = oa.state[a], ob.state[b]
        assert float(sa['step']) == float(sb['step'])
        torch.testing.assert_close(sb['exp_avg'], sa['exp_avg'], rtol=2e-6, atol=1e-9)
        torch.testing.assert_close(sb['exp_avg_sq'], sa['exp_avg_sq'], rtol=2e-6, atol=1e-12)
