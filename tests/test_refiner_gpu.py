"""The device-driven refiner (csrc/refiner.hip + the layer chains of csrc/mlp_gemm.hip): (1) a chain launch equals the layer-by-layer
launches bit for bit, on a row count read from device memory; (2) the compacting refiner returns the same points / flags as the
layer-by-layer host loop on a batch where rays finish at every step, leave no ray behind, and is deterministic;
(3) launch count per call = 3 + 5*times + 3 (+1 init)."""
import ctypes
import numpy as np
import pytest
import torch
from oracle import fixtures as fx

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RATIO = {'sdfRatio': 1.0, 'deformerRatio': 0.62, 'renderRatio': 1.0}


def _nets():
    from selfreconcode_amd.model.network import getTmpSdf
    from selfreconcode_amd.model.Deformer import MLPTranslator, LBSkinner, CompositeDeformer
    from selfreconcode_amd.utils.utils import smpl_tmp_Apose
    sdf = getTmpSdf(DEV, 6, 0.6, 256); sdf.load_state_dict(fx.sphere_sdf_params(7), strict=True)
    tr = MLPTranslator(128, 6).to(DEV); tr.load_state_dict(fx.det_params(fx.DEF_SPEC, 202, last_scale=0.05), strict=True)
    skin = LBSkinner(fx.synthetic_lbs_volume((9, 29, 17)), fx.LBS_BMIN, fx.LBS_BMAX, fx.synthetic_joints(), np.array(fx.SMPL_PARENTS),
                     init_pose=torch.from_numpy(smpl_tmp_Apose(1)), align_corners=False).to(DEV)
    return sdf, CompositeDeformer([tr, skin]).to(DEV)


def _rays(P, N=3, jitter=4e-4):
    """Seeds near the sphere-like zero set along camera rays: a quarter starts on the surface, the rest needs Newton steps."""
    cam = torch.tensor([0., 0.15, 2.4])
    d = torch.nn.functional.normalize(fx.det_tensor((P, 3), 5, 1.0) * torch.tensor([0.22, 0.22, 0.05]) + torch.tensor([0., -0.05, -1.0]), dim=1)
    b = (cam * d).sum(1); c = (cam * cam).sum() - 0.36
    t = -b - torch.sqrt((b * b - c).clamp(min=0))
    p = cam + t[:, None] * d
    return cam, d, p, torch.arange(P) % N, jitter


def test_chain_equals_layerwise_launches_on_a_device_row_count():
    from selfreconcode_amd import _lib, mlp_engine as me
    from selfreconcode_amd.utils import FindSurfacePs as F
    sdf, comp = _nets()
    N = 3
    defconds = [fx.det_tensor((N, 128), 3, 0.1).to(DEV), [fx.det_tensor((N, 24, 3), 1, 0.1).to(DEV), fx.det_tensor((N, 3), 2, 0.05).to(DEV)]]
    with torch.no_grad():
        ev = F._FusedEval(sdf, comp, defconds, RATIO)
        P, M = 3000, 1777                                    # capacity 3072 rows, 1777 of them live
        ws = F._RefinerWorkspace(torch.device(DEV), 3072, ev, 10)
        x = (fx.det_tensor((P, 3), 9, 0.6)).to(DEV)
        bi = (torch.arange(P) % N).to(DEV)
        A0 = ev._embed(x, sdf.multires, ev.w_sdf, None, None, 1)
        A0d = ev._embed(x, ev.tr.multires, ev.w_def, ev.conds, bi, 1)
        ref_s = me.forward(ev.sdf_spec, A0[:M].contiguous(), ev.sdf_W, ev.sdf_b, 1)
        ref_d = me.forward(ev.tr.spec, A0d[:M].contiguous(), ev.def_W, ev.def_b, 1)
        ws.a0[:P].copy_(A0); ws.a0d[:P].copy_(A0d)
        ws.live[0] = M
        fwd = F._forward_chain(ws, ev)
        fwd.m_mul, fwd.m_dev, fwd.m_cap = 1, ws.live.data_ptr(), 3072
        for t in ws.sdf_act + ws.def_act:
            t.fill_(float('nan'))
        _lib.call("sr_mlp_chain", ctypes.byref(fwd), _lib.stream_of(x))
        torch.cuda.synchronize()
        for a, b, L in zip(ws.sdf_act, ref_s, ev.sdf_spec.layers):
            n = L.N + L.nfill                            # (the pad columns of a row are never written by either path)
            assert torch.equal(a[:M, :n], b[:, :n])      # bit-identical: same tile code, same k order
            assert torch.isnan(a[M:]).all()              # rows past the live count untouched
        for a, b, L in zip(ws.def_act, ref_d, ev.tr.spec.layers):
            assert torch.equal(a[:M, :L.N], b[:, :L.N])
        # reverse chain == me.reverse (input gradients)
        ones = ev.unit_cotangent(M)
        tcot = fx.det_tensor((M, 4), 10, 1.0).to(DEV); tcot[:, 3] = 0
        rs, _, _ = me.reverse(ev.sdf_spec, A0[:M].contiguous(), ev.sdf_WT, ref_s, ones, 1, True, False)
        rd, _, _ = me.reverse(ev.tr.spec, A0d[:M].contiguous(), ev.def_WT, ref_d, tcot, 1, True, False)
        ws.unit[:M].copy_(ones); ws.t[:M].copy_(tcot)
        rev = F._reverse_chain(ws, ev)
        rev.m_mul, rev.m_dev, rev.m_cap = 1, ws.live.data_ptr(), 3072
        _lib.call("sr_mlp_chain", ctypes.byref(rev), _lib.stream_of(x))
        torch.cuda.synchronize()
        skip = ws.sdf_zbar[4][:M, 473:512]
        got = ws.a0bar[:M, :39] + skip
        assert torch.equal(got, rs[:, :39]) and torch.equal(ws.a0dbar[:M, :167], rd[:, :167])
        ws.live[0] = 0                                       # no live rows: nothing is touched, nothing hangs
        _lib.call("sr_mlp_chain", ctypes.byref(fwd), _lib.stream_of(x)); torch.cuda.synchronize()


@pytest.mark.parametrize("P,times", [(5000, 10), (777, 3), (64, 10), (130, 2)])
def test_device_driven_refiner_equals_the_layerwise_loop(P, times):
    from selfreconcode_amd.utils import FindSurfacePs as F
    sdf, comp = _nets()
    N = 3
    defconds = [fx.det_tensor((N, 128), 3, 0.1).to(DEV), [fx.det_tensor((N, 24, 3), 1, 0.1).to(DEV), fx.det_tensor((N, 3), 2, 0.05).to(DEV)]]
    cam, _, p0, bi, jitter = _rays(P, N)
    with torch.no_grad():
        # put the seeds ON the zero set of the network (a few exact Newton steps along grad f), then push three quarters of them off it
        x = p0.to(DEV)
        for _ in range(6):
            xg = x.clone().requires_grad_(True)
            with torch.enable_grad():
                f = sdf(xg, 1.0, sdf_only=True)
                g = torch.autograd.grad(f.sum(), xg)[0]
            x = x - f.detach() * g / (g * g).sum(1, keepdim=True)
        push = (fx.det_tensor((P, 3), 6, 1.0) * jitter * (torch.arange(P) % 4 != 0).float()[:, None]).to(DEV) * (1 + (torch.arange(P, device=DEV) % 7)[:, None])
        p0 = (x + push).cpu()
        # the pixel ray of a seed = direction from the camera to its DEFORMED position: the angular term starts at zero
        rays = torch.nn.functional.normalize(comp(p0.to(DEV), defconds, bi.to(DEV), ratio=RATIO).cpu() - cam, dim=1)
    outs = []
    for flag in (False, True, True):
        F.DEVICE_DRIVEN = flag
        p_in = p0.to(DEV).clone()
        ps, ok = F.OptimizeSurfacePs(cam.to(DEV), rays.to(DEV), p_in, bi.to(DEV), sdf, RATIO, comp, defconds, dthreshold=5.e-5, athreshold=0.3,
                                     w1=3.05, w2=1., times=times)
        assert ps.data_ptr() == p_in.data_ptr() or torch.equal(ps, p_in)
        outs.append((ps.cpu().clone(), ok.cpu().clone()))
    F.DEVICE_DRIVEN = True
    (pa, oa), (pb, ob), (pc, oc) = outs
    assert torch.equal(pb, pc) and torch.equal(ob, oc)                       # deterministic although queue slots are claimed by atomics
    # Same arithmetic per ray, but a row's tile is an interior or an edge tile depending on where the compaction put it, and the
    # two epilogue instantiations contract their multiply-adds differently (1 ulp); the residual iteration amplifies that for
    # the rays that zigzag towards a threshold.  So: flags equal up to threshold flips, nearly all points equal to 5e-6, none far.
    assert int((oa != ob).sum()) <= max(3, int(0.015 * P))
    same = oa == ob
    dev_ = (pb[same] - pa[same]).abs().max(dim=1).values
    assert int((dev_ >= 5e-6).sum()) <= max(4, int(0.03 * P)) and float(dev_.max()) < 5e-3, (int((dev_ >= 5e-6).sum()), dev_.max())
    # the queue really shrinks: live counts per phase (device memory, read back here only)
    ws = F._WORKSPACES[(torch.device(DEV), torch.cuda.current_stream(torch.device(DEV)).cuda_stream)]      # the workspace of this (device, stream)
    assert ws.cap >= P
    live = ws.live[:times + 2].cpu().tolist()
    assert live[0] == P and all(a >= b for a, b in zip(live, live[1:]))
    assert 0.15 * P < P - live[1] < 0.6 * P                                   # the on-surface quarter is retired by the initial test
    if times >= 10:
        assert live[times + 1] < 0.5 * P and float(ob.float().mean()) > 0.5, (live, float(ob.float().mean()))   # and most of the rest along the way


def test_refiner_edge_cases():
    """No rays, one ray, zero steps: the fixed launch sequence must cope (grids sized from the ray count, queue compaction on
    partial waves)."""
    from selfreconcode_amd.utils import FindSurfacePs as F
    sdf, comp = _nets()
    N = 3
    defconds = [fx.det_tensor((N, 128), 3, 0.1).to(DEV), [fx.det_tensor((N, 24, 3), 1, 0.1).to(DEV), fx.det_tensor((N, 3), 2, 0.05).to(DEV)]]
    cam = torch.tensor([0., 0.15, 2.4], device=DEV)
    ps, ok = F.OptimizeSurfacePs(cam, torch.zeros(0, 3, device=DEV), torch.zeros(0, 3, device=DEV), torch.zeros(0, dtype=torch.long, device=DEV), sdf, RATIO,
                                 comp, defconds, times=10)
    assert ps.shape == (0, 3) and ok.shape == (0,)
    p0 = torch.tensor([[0.0, 0.0, 0.6]], device=DEV)
    rays = torch.nn.functional.normalize(p0 - cam, dim=1)
    for times in (0, 10):
        outs = []
        for flag in (False, True):
            F.DEVICE_DRIVEN = flag
            ps, ok = F.OptimizeSurfacePs(cam, rays, p0.clone(), torch.zeros(1, dtype=torch.long, device=DEV), sdf, RATIO, comp, defconds,
                                         dthreshold=5e-5, athreshold=0.3, times=times)
            assert ps.shape == (1, 3) and torch.isfinite(ps).all()
            outs.append((ps.cpu(), ok.cpu()))
        F.DEVICE_DRIVEN = True
        assert torch.equal(outs[0][1], outs[1][1]) and torch.allclose(outs[0][0], outs[1][0], atol=5e-6)


def test_enqueueing_kernels_write_the_rows_the_embed_launch_would():
    """init, mid(CHECK) and finish write the first-layer input rows of the rays they put into the next queue; the stand-alone
    sr_refine_embed on the same queue must produce the same bits (padding columns included)."""
    import ctypes
    from selfreconcode_amd import _lib
    from selfreconcode_amd.utils import FindSurfacePs as F
    sdf, comp = _nets()
    N, P, times = 3, 3001, 4
    defconds = [fx.det_tensor((N, 128), 3, 0.1).to(DEV), [fx.det_tensor((N, 24, 3), 1, 0.1).to(DEV), fx.det_tensor((N, 3), 2, 0.05).to(DEV)]]
    cam, rays, p0, bi, _ = _rays(P, N)
    seen = []

    def watch(a, ws, phase, st):
        live = int(ws.live[phase])
        got = (ws.a0[:live].clone(), ws.a0d[:live].clone())
        ws.a0.fill_(float('nan')); ws.a0d.fill_(float('nan'))
        _lib.call("sr_refine_embed", ctypes.byref(a), phase, st)
        for fused, alone in zip(got, (ws.a0[:live], ws.a0d[:live])):
            assert torch.isfinite(alone).all() and torch.equal(fused, alone), phase
        seen.append((phase, live))
    F.ENQUEUE_WATCH = watch
    try:
        F.OptimizeSurfacePs(cam.to(DEV), rays.to(DEV), p0.to(DEV).clone(), bi.to(DEV), sdf, RATIO, comp, defconds, dthreshold=5.e-5, athreshold=0.3,
                            w1=3.05, w2=1., times=times)
    finally:
        F.ENQUEUE_WATCH = None
    assert [p for p, _ in seen] == list(range(times + 2)) and seen[0][1] == P and seen[1][1] > 0
