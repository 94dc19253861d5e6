"""TEST/BENCH INFRASTRUCTURE -- times the CPU oracle (oracle/torch_oracle.py, a restatement of the
reference's own PyTorch path) on a BOUNDED sample of one training iteration and scales linearly to the
full point counts.  Used only by bench.py's `cpu_baseline` leg (kind "port"): a reported baseline, not a
target.  The reference's full train.py cannot run on CPU (CUDA-only extensions + pytorch3d), so this is
a component-sum estimate of one iteration: every MLP / LBS / refiner term of SURVEY.md 3.2 is timed
through the oracle on a sample and multiplied up; rasterisation and remesh are left out (they favour
the CPU number)."""
import os
import time
import torch
from . import torch_oracle as orc
from . import fixtures as fx

RATIO = {'sdfRatio': 1.0, 'deformerRatio': 0.6, 'renderRatio': 1.0}


def _t(fn, reps=1):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def estimate_iteration_seconds(V, P, N, conv_frac=0.5, tracer_iters=6.0, sample=1024, threads=None):
    threads = threads or min(os.cpu_count(), 16)     # small-op oracle: more threads only add sync overhead
    torch.set_num_threads(threads)
    sdf = {k: v.clone().requires_grad_(True) for k, v in fx.sphere_sdf_params(7).items()}
    trp = {k: v.clone().requires_grad_(True) for k, v in fx.det_params(fx.DEF_SPEC, 202, last_scale=0.05).items()}
    rnd = {k: v.clone().requires_grad_(True) for k, v in fx.det_params(fx.REND_SPEC, 303).items()}
    vol = fx.synthetic_lbs_volume((17, 57, 33))
    Js = fx.synthetic_joints()
    import math
    apose = torch.zeros(24, 3); apose[1, 2] = 7 / 180 * math.pi; apose[2, 2] = -7 / 180 * math.pi
    apose[16, 2] = -55 / 180 * math.pi; apose[17, 2] = 55 / 180 * math.pi
    ip = orc.make_init_pose_inverse(apose, Js)
    poses = fx.det_tensor((N, 24, 3), 21, 0.1).requires_grad_(True); trans = fx.det_tensor((N, 3), 22, 0.05).requires_grad_(True)
    conds = fx.det_tensor((N, 128), 13, 0.1).requires_grad_(True)
    kw = dict(ws=vol, b_min=torch.tensor(fx.LBS_BMIN), b_max=torch.tensor(fx.LBS_BMAX), Js=Js, init_pose=ip)
    n = sample
    pts = (fx.det_tensor((n, 3), 1, 0.5) * torch.tensor([0.6, 1.0, 0.3])).requires_grad_(True)
    bi = torch.arange(n) % N

    def deform(p, b):
        q, _ = orc.translator_forward(trp, p, conds, b, RATIO)
        return orc.lbs_forward(q, poses, trans, batch_inds=b, **kw)

    parts = {}
    # step 3+6: deformer on the template fwd+bwd, LBS-only consistency term, |f(TmpVs)|
    def a():
        d = deform(pts, bi)
        l = orc.lbs_forward(pts, poses, trans, batch_inds=bi, **kw)
        ((d - l) ** 2).sum().backward()
    parts['template_deformer'] = _t(a) * (N * V / n)
    def c():
        orc.sdf_forward(sdf, pts, 1.0)[0].abs().mean().backward()
    parts['template_sdf'] = _t(c) * (V / n)
    # step 8: refiner
    nr = min(256, n)
    p0 = pts[:nr].detach().clone(); rays = torch.nn.functional.normalize(fx.det_tensor((nr, 3), 3, 1.0), dim=1)
    def tr():
        with torch.enable_grad():
            orc.optimize_surface_ps(torch.tensor([0., 0., 2.4]), rays, p0.clone(), bi[:nr], lambda p: orc.sdf_forward(sdf, p, RATIO)[0],
                                    lambda p, b: deform(p, b), 5e-5, 0.04, 3.05, 1., 2)
    parts['refiner'] = _t(tr) / 2.0 * tracer_iters * (P / nr)
    # step 9: eikonal
    def e():
        x = pts.detach().clone().requires_grad_(True)
        y, _ = orc.sdf_forward(sdf, x, 1.0)
        g = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True)[0]
        ((g.norm(2, dim=-1) - 1) ** 2).mean().backward()
    parts['eikonal'] = _t(e) * ((P + 4096) * 7 / 6 / n)
    # step 11: deformation regulariser
    def r():
        x = pts.detach().clone().requires_grad_(True)
        d, _ = orc.translator_forward(trp, x, conds, bi, RATIO)
        J = orc.compute_jacobian(x, d, True, True)
        s = torch.log(torch.linalg.svdvals(J))
        orc.gm_robust((s * s).sum(1), 0.5, True).mean().backward()
    parts['def_regu'] = _t(r) * (2 * (P + 4096) * N / n)
    # steps 13+14 (+ propagate): colour and normal branches on the converged rays
    nc = min(256, n)
    def cn():
        x = pts[:nc].detach().clone().requires_grad_(True)
        y, feat = orc.sdf_forward(sdf, x, 1.0)
        nx = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True)[0]
        nx = nx / nx.norm(dim=1, keepdim=True)
        d = deform(x, bi[:nc])
        J = orc.compute_jacobian(x, d, True, True)
        Ji, _ = orc.DiffMinv.apply(J)
        cr = (Ji @ rays[:nc].view(-1, 3, 1)).view(-1, 3)
        cr = cr / cr.norm(dim=1, keepdim=True)
        col = orc.render_forward(rnd, x, nx, cr, feat, RATIO)
        d2 = deform(x, bi[:nc])
        J2 = orc.compute_jacobian(x, d2, True, True)
        gn = (J2.transpose(-2, -1) @ rays[:nc].view(-1, 3, 1)).view(-1, 3)
        (col.abs().sum() + (gn - nx).norm(dim=1).sum()).backward()
        # implicit-gradient propagation: 3 SDF + 2 deformer evaluations with parameter gradients
        orc.sdf_forward(sdf, x.detach(), 1.0)[0].sum().backward()
        deform(x.detach(), bi[:nc]).sum().backward()
    parts['color_normal_propagate'] = _t(cn) * (P * conv_frac / nc)
    total = sum(parts.values())
    return total, parts, threads


def full_iteration_seconds(net, ds, frame_ids, sample_pix, ratio, threads=None):
    """ONE whole iteration of the CPU oracle (oracle/iteration_oracle.py: forward + backward + propagate, its own refiner, the numpy
    rasterisers) AT THE SIZE bench.py times, on the product's current state (weights, template, observations copied to the host), in
    float32 -- measured, not extrapolated.  Returns (seconds, seconds of the rasteriser restatement inside it, threads, info).
    Used only by bench.py's `cpu_baseline` leg (kind "port")."""
    import time
    from . import iteration_oracle as ito
    threads = threads or min(os.cpu_count(), 32)
    torch.set_num_threads(threads)
    cp = lambda sd: {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sd.items() if v.dtype.is_floating_point}
    skin = net.deformer.defs[1]
    sk = dict(ws=skin.ws.detach().cpu().contiguous(), b_min=skin.b_min.cpu().view(3), b_max=skin.b_max.cpu().view(3), Js=skin.Js.cpu(), init_pose=skin.init_pose.cpu())
    leaf = lambda t: t.detach().cpu().clone().requires_grad_(True)
    q = ds.camera_params['cam2world_coord_quat'].detach().cpu().view(1, 4)
    camleaf = lambda t: t.detach().cpu().clone().requires_grad_(t.requires_grad)
    cam = dict(focal=camleaf(ds.camera_params['focal_length']), princ=camleaf(ds.camera_params['princeple_points']), R=orc.quat2mat(q)[0],
               T=camleaf(ds.camera_params['world2cam_coord_trans']), H=ds.H, W=ds.W)
    sc = ito.Scene(cp(net.sdf.state_dict()), cp(dict(net.deformer.defs[0].state_dict())), cp(net.netRender.state_dict()), sk, leaf(ds.poses), leaf(ds.trans),
                   leaf(ds.conds[0]), leaf(ds.conds[1]), cam, net.conf, net.point_radius, float(net.angThred))
    V = net.TmpVs.shape[0]
    TmpVs = net.TmpVs.detach().cpu().clone().requires_grad_(True)
    opt = torch.optim.SGD([TmpVs], lr=0.05, momentum=0.9)
    fo = frame_ids.cpu()
    datas = {k: v.cpu() for k, v in ds.batch(frame_ids).items()}
    g = torch.Generator().manual_seed(0)
    big = ds.H * ds.W * int(fo.numel())
    rand = {'ray_select': torch.rand(big, generator=g), 'vert_select': torch.rand(V, generator=g), 'vert_select2': torch.rand(V, generator=g),
            'eik_local': torch.randn(20000, 3, generator=g), 'eik_global': torch.rand(20000, 3, generator=g), 'regu_local': torch.randn(20000, 3, generator=g)}
    F = ds.frame_num
    bf = lambda f, n: ((f - n // 2).clamp(min=0, max=F - n)).view(-1, 1) + torch.arange(n).view(1, n)
    raster = {'s': 0.0}
    real_mesh, real_pts = ito.ro.rasterize_meshes, ito.ro.rasterize_points

    def timed(fn):
        def wrapper(*a, **k):
            t0 = time.perf_counter()
            out = fn(*a, **k)
            raster['s'] += time.perf_counter() - t0
            return out
        return wrapper
    ito.ro.rasterize_meshes, ito.ro.rasterize_points = timed(real_mesh), timed(real_pts)
    # warm-up (untimed): the thread pool, the BLAS kernels and the allocator see the iteration's kind of work once -- two eikonal steps of
    # the oracle's SDF on 16k points (forward, gradient, double backward; ~1 s) -- so that the ONE timed iteration is not a cold start
    wsd = cp(net.sdf.state_dict())
    for _ in range(2):
        xw = (torch.rand(16384, 3, generator=g) - 0.5).requires_grad_(True)
        yw = orc.sdf_forward(wsd, xw, 1.0)[0]
        gw = torch.autograd.grad(yw, xw, torch.ones_like(yw), create_graph=True)[0]
        ((gw.norm(2, dim=-1) - 1) ** 2).mean().backward()
    del wsd
    try:
        t0 = time.perf_counter()
        tot, info, st = ito.forward(sc, TmpVs, net.Tmpfs.cpu(), opt, datas, sample_pix, ratio, fo, rand, dctnull=net.dctnull.cpu(), batchframe=bf)
        tot.backward()
        if st is not None:
            ito.propagate(sc, st, fo, ratio)
        sec = time.perf_counter() - t0
    finally:
        ito.ro.rasterize_meshes, ito.ro.rasterize_points = real_mesh, real_pts
    return sec, raster['s'], threads, {"rays": int(info['rays']), "rays_converged": int(info['check'].sum()), "template_vertices": int(V)}
