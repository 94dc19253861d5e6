"""GPU parity: LBS skinner (fused kernel and differentiable composition), composite deformer
Jacobians / cardinal rays / deformed normals, FindSurfacePs and the fused ray refiner."""
import pytest
import torch
from oracle import torch_oracle as orc
from oracle import fixtures as fx

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RATIO = {'sdfRatio': 1.0, 'deformerRatio': 0.62, 'renderRatio': 1.0}


def close(a, b, rtol=2e-5, atol=2e-6):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


def _skinner(shape=(7, 11, 9)):
    import numpy as np
    from selfreconcode_amd.model.Deformer import LBSkinner
    from selfreconcode_amd.utils import smpl_tmp_Apose
    vol = fx.synthetic_lbs_volume(shape)
    return LBSkinner(vol, fx.LBS_BMIN, fx.LBS_BMAX, fx.synthetic_joints(), np.array(fx.SMPL_PARENTS),
                     init_pose=torch.from_numpy(smpl_tmp_Apose(1)), align_corners=False).to(DEV)


def _composite(last_scale=None):
    from selfreconcode_amd.model.Deformer import MLPTranslator, CompositeDeformer
    tr = MLPTranslator(128, 6).to(DEV)
    tr.load_state_dict(fx.det_params(fx.DEF_SPEC, 202, last_scale=last_scale), strict=True)
    return CompositeDeformer([tr, _skinner()]).to(DEV)


def test_lbs_golden_fused_and_differentiable(golden):
    g = golden("lbs")
    skin = _skinner()
    close(skin.init_pose, g["init_pose"], atol=1e-6)
    assert skin.ws.shape == (1, 24, 7, 11, 9)
    conds = [g["poses"].to(DEV), g["trans"].to(DEV)]
    close(skin.posedSkeleton(conds), g["newJ"], atol=1e-6)
    with torch.no_grad():                                            # fused kernel
        close(skin(g["p"].to(DEV), conds, g["bi"].to(DEV)), g["y"], atol=2e-6)
        close(skin(g["p"][:48].view(3, 16, 3).to(DEV), conds, None), g["yb"], atol=2e-6)
    p = g["p"].to(DEV).requires_grad_(True)                          # autograd composition
    close(skin(p, conds, g["bi"].to(DEV)), g["y"], atol=2e-6)
    pb = g["p"][:48].view(3, 16, 3).to(DEV).requires_grad_(True)
    close(skin(pb, conds, None), g["yb"], atol=2e-6)


def test_lbs_fused_jacobian_and_pose_gradients(golden):
    g = golden("lbs")
    skin = _skinner()
    poses, trans = g["poses"], g["trans"]
    kw = dict(ws=fx.synthetic_lbs_volume((7, 11, 9)), b_min=torch.tensor(fx.LBS_BMIN), b_max=torch.tensor(fx.LBS_BMAX),
              Js=fx.synthetic_joints(), init_pose=g["init_pose"])
    po = g["p"].clone().requires_grad_(True); pso = poses.clone().requires_grad_(True); to = trans.clone().requires_grad_(True)
    yo = orc.lbs_forward(po, pso, to, batch_inds=g["bi"], **kw)
    Jo = orc.compute_jacobian(po, yo, True, False)
    A = skin.posed_transforms(poses.to(DEV))
    y, J = skin.fused(g["p"].to(DEV), A, trans.to(DEV), g["bi"].to(DEV), with_jac=True)
    close(y, yo, atol=2e-6); close(J, Jo, 1e-4, 1e-5)
    go = fx.det_tensor((50, 3), 77, 1.0)
    ref = torch.autograd.grad((yo * go).sum(), [po, pso, to])
    p = g["p"].to(DEV).requires_grad_(True); ps = poses.to(DEV).requires_grad_(True); t = trans.to(DEV).requires_grad_(True)
    yy = skin(p, [ps, t], g["bi"].to(DEV))
    ours = torch.autograd.grad((yy * go.to(DEV)).sum(), [p, ps, t])
    for a, b in zip(ours, ref):
        close(a, b, 2e-4, 2e-5)


def test_cardinal_rays_and_deformed_normals_golden(golden):
    """utils/utils.py:132-169 through the drop-in modules (second-order graph incl. the sampler's dbackward)."""
    from selfreconcode_amd.utils import compute_cardinal_rays, compute_deformed_normals
    from selfreconcode_amd.model.network import getTmpSdf
    g, gl, gt = golden("cardinal"), golden("lbs"), golden("translator")
    comp = _composite()
    sdf = getTmpSdf(DEV, 6, 0.6, 256)
    sdf.load_state_dict(fx.det_params(fx.SDF_SPEC, 101), strict=True)
    defconds = [gt["conds"].to(DEV).requires_grad_(True), [gl["poses"].to(DEV).requires_grad_(True), gl["trans"].to(DEV)]]
    p = g["p"].to(DEV).requires_grad_(True)
    crays, ds = compute_cardinal_rays(comp, p, g["rays"].to(DEV), defconds, g["bi"].to(DEV), RATIO, 'train')
    close(ds, g["ds"], atol=2e-6); close(crays, g["crays"], 1e-4, 1e-5)
    nx, _ = compute_deformed_normals(sdf, comp, p, defconds, g["bi"].to(DEV), RATIO, 'train')
    close(nx, g["nx"], 1e-4, 1e-5)
    # and the whole second-order graph is differentiable down to weights, codes and poses
    grads = torch.autograd.grad((crays * nx).sum(), [comp.defs[0].lin2.weight, defconds[0], defconds[1][0], sdf.lin5.weight_v, p])
    assert all(torch.isfinite(t).all() and t.abs().sum() > 0 for t in grads)


def test_find_surface_ps_golden(golden):
    from selfreconcode_amd.utils.FindSurfacePs import FindSurfacePs
    g = golden("findsurf")

    class Frag:
        pix_to_face = g["p2f"].to(DEV)
        bary_coords = g["bary"].to(DEV)
    b, r, c, p0, f = FindSurfacePs(g["V"].to(DEV), g["F"].to(DEV), Frag)
    for a, k in ((b, "b"), (r, "r"), (c, "c"), (f, "finds")):
        assert torch.equal(a.cpu(), g[k])
    close(p0, g["p0"], atol=1e-7)


def test_optimize_surface_ps_golden(golden):
    """The fused refiner against the reference's own run (tests/golden/tracer.npz).  |f| < 5e-5 is
    threshold-sensitive in fp32: points must agree to 2e-5 and the converged flags except for rays
    sitting within 10% of a threshold."""
    from selfreconcode_amd.utils.FindSurfacePs import OptimizeSurfacePs
    from selfreconcode_amd.model.network import getTmpSdf
    g, gl, gt = golden("tracer"), golden("lbs"), golden("translator")
    comp = _composite(last_scale=0.05)
    sph = getTmpSdf(DEV, 6, 0.6, 256)
    sph.load_state_dict(fx.sphere_sdf_params(7), strict=True)
    defconds = [gt["conds"].to(DEV), [gl["poses"].to(DEV), gl["trans"].to(DEV)]]
    p_in = g["p0"].to(DEV).clone()
    ps, ok = OptimizeSurfacePs(g["campos"].to(DEV), g["rays"].to(DEV), p_in, g["bi"].to(DEV), sph, RATIO, comp, defconds,
                               dthreshold=5.e-5, athreshold=0.04, w1=3.05, w2=1., times=10)
    assert ps.data_ptr() == p_in.data_ptr() or torch.equal(ps, p_in)          # in-place contract
    close(ps, g["ps"], rtol=0, atol=2e-5)
    agree = (ok.cpu() == g["ok"]).float().mean()
    assert agree > 0.95, agree


def test_translator_forward_mode_jacobian_and_its_reverse():
    """group-4 forward-mode (d, J) == autograd Jacobian of the oracle; gradients of a loss on (d, J) w.r.t. weights,
    per-frame codes and points == the oracle's reverse-over-reverse."""
    from selfreconcode_amd.model.Deformer import MLPTranslator, translator_value_jacobian
    tr = MLPTranslator(128, 6).to(DEV)
    sd = fx.det_params(fx.DEF_SPEC, 11)
    tr.load_state_dict(sd, strict=True)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ps = fx.det_tensor((3, 21, 3), 1, 0.7); conds = fx.det_tensor((3, 128), 2, 0.1)
    cd, cj = fx.det_tensor((3, 21, 3), 3, 1.0), fx.det_tensor((63, 3, 3), 4, 1.0)
    po = ps.clone().requires_grad_(True); co = conds.clone().requires_grad_(True)
    do, _ = orc.translator_forward(sdo, po, co, None, RATIO)
    Jo = orc.compute_jacobian(po, do, True, True)
    ref = torch.autograd.grad((do * cd).sum() + (Jo * cj).sum(), [po, co, sdo["lin0.weight"], sdo["lin2.weight"], sdo["lin4.bias"], sdo["lin1.bias"]])
    pg = ps.to(DEV).requires_grad_(True); cg = conds.to(DEV).requires_grad_(True)
    d, J = translator_value_jacobian(tr, pg, cg, None, RATIO)
    close(d, do, atol=2e-6); close(J, Jo, 1e-4, 1e-5)
    ours = torch.autograd.grad((d * cd.to(DEV)).sum() + (J * cj.to(DEV)).sum(), [pg, cg, tr.lin0.weight, tr.lin2.weight, tr.lin4.bias, tr.lin1.bias])
    for a, b in zip(ours, ref):
        close(a, b, 1e-3, 1e-4 * max(1.0, float(b.abs().max())))
    # ray mode (batch_inds)
    bi = torch.arange(30) % 3
    p2 = fx.det_tensor((30, 3), 5, 0.6)
    d2o, _ = orc.translator_forward(sdo, p2, conds, bi, RATIO)
    d2, J2 = translator_value_jacobian(tr, p2.to(DEV), conds.to(DEV), bi.to(DEV), RATIO)
    close(d2, d2o, atol=2e-6)


def test_frame_batched_translator_equals_the_per_frame_passes():
    """Frame-major batches run as ONE batch with a per-frame first-layer bias (mlp_engine: segmented bias; Deformer.BATCH_FRAMES) or as one
    MLP pass per frame: same offsets, same first-order gradients (points, codes, every layer) and same SECOND-order gradients (the
    group-2 double backward of the engine with a segmented bias), for a vertex count that is not a multiple of any tile size."""
    from selfreconcode_amd.model import Deformer as D
    tr = D.MLPTranslator(128, 6).to(DEV)
    tr.load_state_dict(fx.det_params(fx.DEF_SPEC, 11), strict=True)
    N, V = 3, 333
    ps0 = fx.det_tensor((N, V, 3), 21, 0.7).to(DEV); conds0 = fx.det_tensor((N, 128), 22, 0.1).to(DEV)
    cd, cg = fx.det_tensor((N, V, 3), 23, 1.0).to(DEV), fx.det_tensor((N, V, 3), 24, 1.0).to(DEV)
    params = [tr.lin0.weight, tr.lin0.bias, tr.lin1.weight, tr.lin3.bias, tr.lin4.weight]

    def run(batched):
        old = D.BATCH_FRAMES
        D.BATCH_FRAMES = batched
        try:
            ps = ps0.clone().requires_grad_(True); conds = conds0.clone().requires_grad_(True)
            out = tr(ps, conds, ratio=RATIO)
            first = torch.autograd.grad((out * cd).sum(), [ps, conds] + params, retain_graph=True)
            gp = torch.autograd.grad(out, ps, cd, create_graph=True)[0]                       # d <out, cd> / d ps, kept differentiable
            second = torch.autograd.grad((gp * cg).sum(), [ps, conds] + params, allow_unused=True)
            dJ = D.translator_value_jacobian(tr, ps, conds, None, RATIO)
            jac = torch.autograd.grad((dJ[0] * cd).sum() + (dJ[1].reshape(N, V, 9)[..., :3] * cg).sum(), [ps, conds] + params)
            return out, first, second, dJ, jac
        finally:
            D.BATCH_FRAMES = old
    a, b = run(True), run(False)
    close(a[0], b[0], 1e-5, 1e-6)
    close(a[3][0], b[3][0], 1e-5, 1e-6); close(a[3][1], b[3][1], 1e-5, 1e-6)
    for which in (1, 2, 4):
        for x, y in zip(a[which], b[which]):
            assert (x is None) == (y is None)
            if x is not None:
                close(x, y, 1e-4, 1e-5 * max(1.0, float(y.abs().max())))


def test_kinematic_chain_kernel_vs_oracle():
    """fused chain forward/backward (dual-number Rodrigues) == oracle lbs_transforms + autograd."""
    skin = _skinner()
    poses = fx.det_tensor((5, 24, 3), 31, 0.4)
    poses[0, 3] = 0.0                                             # zero rotation: the +1e-8 path
    po = poses.clone().requires_grad_(True)
    init_pose = skin.init_pose.cpu()
    Ao, newJ = orc.lbs_transforms(po, fx.synthetic_joints(), init_pose)
    ca, cj = fx.det_tensor((5, 24, 4, 4), 32, 1.0), fx.det_tensor((5, 24, 3), 33, 1.0)
    ref = torch.autograd.grad((Ao * ca).sum() + (newJ * cj).sum(), po)[0]
    pg = poses.to(DEV).requires_grad_(True)
    G, A = skin.posed_chain(pg)
    close(A, Ao, atol=2e-6); close(G[:, :, :3, 3], newJ, atol=2e-6)
    ours = torch.autograd.grad((A * ca.to(DEV)).sum() + (G[:, :, :3, 3] * cj.to(DEV)).sum(), pg)[0]
    close(ours, ref, 2e-4, 2e-5)


def test_refiner_forward_and_reverse_mode_agree(golden):
    """the group-4 (forward tangents) and the reverse-sweep formulations of the Newton step give the same refinement."""
    from selfreconcode_amd.utils import FindSurfacePs as F
    from selfreconcode_amd.model.network import getTmpSdf
    g, gl, gt = golden("tracer"), golden("lbs"), golden("translator")
    comp = _composite(last_scale=0.05)
    sph = getTmpSdf(DEV, 6, 0.6, 256)
    sph.load_state_dict(fx.sphere_sdf_params(7), strict=True)
    defconds = [gt["conds"].to(DEV), [gl["poses"].to(DEV), gl["trans"].to(DEV)]]
    outs = []
    for mode in (False, True):
        F.REVERSE_MODE = mode
        ps, ok = F.OptimizeSurfacePs(g["campos"].to(DEV), g["rays"].to(DEV), g["p0"].to(DEV).clone(), g["bi"].to(DEV), sph, RATIO, comp, defconds,
                                     dthreshold=5.e-5, athreshold=0.04, w1=3.05, w2=1., times=10)
        outs.append((ps.cpu(), ok.cpu()))
    F.REVERSE_MODE = True
    close(outs[0][0], outs[1][0], rtol=0, atol=1e-5)
    close(outs[1][0], g["ps"], rtol=0, atol=2e-5)
    assert (outs[0][1] == outs[1][1]).float().mean() > 0.97


@pytest.mark.parametrize("interleaved", [False, True])
def test_forward_mode_deformer_value_jacobian_equals_the_reverse_mode_graph(interleaved):
    """deformer_value_jacobian (group-4 translator pass + fused LBS with analytic Jacobian + sr_lbs_jac_bwd) against the generic
    path it replaces in the colour / normal branch: deformer() + compute_Jacobian with create_graph (three reverse passes through
    the MLP and the differentiable LBS composition, second order through the HIP sampler's double backward).  Value, Jacobian and
    the gradient of a scalar of (d, J) with respect to points, MLP weights, per-frame codes, poses and translations."""
    from selfreconcode_amd.model.Deformer import deformer_value_jacobian
    from selfreconcode_amd.utils.utils import compute_Jacobian
    comp = _composite(last_scale=0.3)
    N, P = 3, 900
    p0 = (fx.det_tensor((P, 3), 31, 1.0) * torch.tensor([0.7, 1.0, 0.35])).to(DEV)
    p0[:5, 0] = 0.95                                                   # a few points outside the skinning-weight box (border rule: zero sampler gradient)
    bi = (torch.arange(P) % N if interleaved else (torch.arange(P) * N) // P).to(DEV)
    wy, wJ = fx.det_tensor((P, 3), 32, 1.0).to(DEV), fx.det_tensor((P, 3, 3), 33, 1.0).to(DEV)
    outs = []
    for fused in (False, True):
        cond = (fx.det_tensor((N, 128), 3, 0.1)).to(DEV).requires_grad_(True)
        poses = fx.det_tensor((N, 24, 3), 1, 0.15).to(DEV).requires_grad_(True)
        trans = fx.det_tensor((N, 3), 2, 0.05).to(DEV).requires_grad_(True)
        p = p0.clone().requires_grad_(True)
        defconds = [cond, [poses, trans]]
        if fused:
            d, J = deformer_value_jacobian(comp, p, defconds, bi, RATIO)
        else:
            d = comp(p, defconds, bi, ratio=RATIO)
            J = compute_Jacobian(p, d, True, True)
        s = (d * wy).sum() + (J * wJ).sum()
        tr = comp.defs[0]
        g = torch.autograd.grad(s, [p, tr.lin0.weight, tr.lin2.weight, tr.lin4.weight, tr.lin4.bias, cond, poses, trans])
        outs.append((d.detach(), J.detach(), g))
    (d0, J0, g0), (d1, J1, g1) = outs
    close(d1, d0, 1e-5, 2e-6); close(J1, J0, 1e-4, 1e-5)
    for name, a, b in zip(("p", "lin0.w", "lin2.w", "lin4.w", "lin4.b", "cond", "poses", "trans"), g1, g0):
        torch.testing.assert_close(a.cpu(), b.cpu(), rtol=2e-3, atol=2e-3 * max(1e-6, float(b.abs().max())), msg=lambda m, name=name: name + ": " + m)


def test_lbs_jacobian_backward_against_float64_autograd():
    """sr_lbs_jac_bwd alone: cotangents on (y, J) of the fused LBS kernel against float64 autograd of the oracle's LBS + its
    autograd Jacobian (incl. the mixed second derivatives of the trilinear weights)."""
    from selfreconcode_amd.model.Deformer import _LBSValueJacobian
    skin = _skinner((9, 13, 11))
    N, P = 2, 400
    q = (fx.det_tensor((P, 3), 41, 1.0) * torch.tensor([0.6, 0.9, 0.3]))
    bi = (torch.arange(P) * N) // P
    poses = fx.det_tensor((N, 24, 3), 1, 0.2); trans = fx.det_tensor((N, 3), 2, 0.05)
    wy, wJ = fx.det_tensor((P, 3), 42, 1.0), fx.det_tensor((P, 3, 3), 43, 1.0)
    qg = q.to(DEV).requires_grad_(True); pg = poses.to(DEV).requires_grad_(True); tg = trans.to(DEV).requires_grad_(True)
    A = skin.posed_transforms(pg)
    y, J = _LBSValueJacobian.apply(skin, qg, A, tg, bi.to(DEV), 0)
    g = torch.autograd.grad((y * wy.to(DEV)).sum() + (J * wJ.to(DEV)).sum(), [qg, pg, tg])
    kw = dict(ws=fx.synthetic_lbs_volume((9, 13, 11)).double(), b_min=torch.tensor(fx.LBS_BMIN).double(), b_max=torch.tensor(fx.LBS_BMAX).double(),
              Js=fx.synthetic_joints().double(), init_pose=skin.init_pose.cpu().double())
    qo = q.double().requires_grad_(True); po = poses.double().requires_grad_(True); to = trans.double().requires_grad_(True)
    yo = orc.lbs_forward(qo, po, to, batch_inds=bi, **kw)
    Jo = orc.compute_jacobian(qo, yo, True, True)
    go = torch.autograd.grad((yo * wy.double()).sum() + (Jo * wJ.double()).sum(), [qo, po, to])
    close(y, yo.float(), 1e-5, 2e-6); close(J, Jo.float(), 1e-4, 1e-5)
    for name, a, b in zip(("q", "poses", "trans"), g, go):
        torch.testing.assert_close(a.cpu(), b.float(), rtol=1e-3, atol=1e-3 * float(b.abs().max()), msg=lambda m, name=name: name + ": " + m)
