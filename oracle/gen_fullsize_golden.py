"""TEST INFRASTRUCTURE ONLY -- ONE WHOLE training iteration of the reference AT THE SIZE BASELINE.json configs[1] IS QUOTED ON,
run verbatim on CPU and frozen into tests/golden/iteration_full_{coarse,fine}.npz (build container only: needs /root/reference):

    python oracle/gen_fullsize_golden.py [coarse|fine|both] [--time] [--no-golden]

540 x 540 images, the real 65 x 225 x 129 skinning-weight volume, a template of 84 968 vertices / 169 932 faces (coarse stage:
3 frames x 2048 rays, config.conf:28-34) or 173 402 vertices (fine stage: 1 frame x 6144 rays, loss_fine, config.conf:39-48,113).
The harness is the one of oracle/gen_iteration_golden.py (the reference's own OptimNetwork.forward + backward +
propagateTmpPsGrad on the reference's own modules; pytorch3d renderers -> oracle/raster_oracle.py, CUDA extensions / torch_scatter
-> their pinned restatements).  Nothing of MB size is stored -- everything big is a pure function both sides evaluate:
  * template = cube_sphere(n) directions x (0.6 + q / 65536) + det offsets; q (int16) is stored: the radius that puts the
    vertex on the zero set of the SDF, quantised (the 4e-3 offsets on top keep |f| away from 0, as in the miniature fixture);
  * images / normals / poses / codes / network weights = det_tensor / det_params / sphere_sdf_params;
  * the random draws of the iteration = det_tensor / det_normal by call order (torch.rand / torch.randn_like are replaced while
    forward runs), so the product gets the same numbers through `rand=`.
Stored: ray selection, seeds, the refiner's output, every loss term, the total, the template step (strided), dL/dTmpPs, whole
per-frame / camera gradients and -- for every parameter of the three networks -- its L2 norm, two fixed random projections and a
strided slice, and f at the moved template vertices.  The fixture is the reference as it runs, in float32.  (`--f64` evaluates the same
modules in double on the same inputs -- see build().  Tried as the fixture and dropped: the discrete decisions of the iteration --
which (point, pixel) pairs of the silhouette splat exist, which rays the refiner accepts -- are float32 decisions in the reference
and here; a float64 evaluation makes MORE of them differently from the product than the float32 reference does, and a flipped pair
moves the gradient of its vertex by a finite amount.)  `--time` additionally writes profiles/r03_cpu_reference.json: the reference's own modules in float32, as they run, timed on this
container's cores at full size (warm-up 1, then median of 3) -- the `cpu_baseline` of kind "reference" that bench.py reports.
"""
import json
import os
import sys
import time
import types
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import gen_iteration_golden as gi  # noqa: E402   (applies the harness patches on import)
from oracle import fixtures as fx  # noqa: E402
from oracle import torch_oracle as orc  # noqa: E402

ref = gi.ref
OUT = os.path.join(ROOT, "tests", "golden")
RATIO = {'sdfRatio': 1.0, 'deformerRatio': 0.62, 'renderRatio': 1.0}
LOSS_FINE = {'color_weight': 1.0, 'normal_weight': 0.1, 'weighted_normal': True, 'grad_weight': 1., 'offset_weight': 0.,
             'def_regu': {'weight': 0.07, 'c': 0.5}, 'dct_weight': 4., 'sample_pix_num': 6144,
             'pc_weight': {'weight': 10., 'laplacian_weight': -1., 'edge_weight': -10., 'norm_weight': -0.001, 'def_consistent': {'weight': 0.1, 'c': 0.01}}}
LOSS_MEDIUM = {'color_weight': 1.0, 'normal_weight': 0.1, 'weighted_normal': True, 'grad_weight': 1., 'offset_weight': 0.,          # config.conf:84-104
               'def_regu': {'weight': 0.1, 'c': 0.5}, 'dct_weight': 3.,
               'pc_weight': {'weight': 30., 'laplacian_weight': -1., 'edge_weight': -10., 'norm_weight': -0.001, 'def_consistent': {'weight': 0.2, 'c': 0.01}}}
STAGES = {
    "medium": dict(conf=LOSS_MEDIUM, N=2, SP=2048, n_cube=153, radius=0.00465, fids=[9, 27]),      # config.conf:36-42 (batch 2, 289 x 385 x 193 grid -> ~140k vertices)
    "coarse": dict(conf=gi.LOSS_COARSE, N=3, SP=2048, n_cube=119, radius=0.006, fids=[21, 7, 30]),
    "fine": dict(conf=LOSS_FINE, N=1, SP=2048, n_cube=170, radius=0.0041, fids=[13]),       # SP comes from conf.sample_pix_num = 6144 (network.py:520)
    # configs[4]: config_loose.conf at 1080 x 1080.  Against config.conf it changes the schedule, switches the normal loss off
    # (normal_weight = -0.1, :70) and learns the focal length only (opt_camera: princeple_points = false, T = false, :12-14).
    "loose1080": dict(conf=dict(gi.LOSS_COARSE, normal_weight=-0.1), N=3, SP=2048, n_cube=119, radius=0.006, fids=[21, 7, 30], HW=1080, learn_cam=("focal",)),
}
DRAW_SEED0 = 5000
PROJ_SEEDS = (7001, 7002)


def param_digest(p, seed):
    """(L2 norm, <g, r1>, <g, r2>) of a gradient in float64 -- a whole-tensor check that costs three numbers."""
    g = p.detach().double().reshape(-1)
    r = [fx.det_tensor((g.numel(),), s + seed, 1.0, torch.float64) for s in PROJ_SEEDS]
    return np.array([float(g.norm()), float(g @ r[0]), float(g @ r[1])])


def slice_of(t):
    if t.dim() == 2 and t.shape[1] > 1:
        return t[::29, ::7]
    return t.reshape(-1)[::5]


def mask_image(N, H, W):
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    m = (((xs - W / 2.0) / (0.2963 * W)) ** 2 + ((ys - 0.45 * H) / (0.3426 * H)) ** 2 < 1.0).float()
    return m[None].expand(N, H, W).contiguous()


def build(stage, H=None, W=None, lbs_shape=(65, 225, 129), F=40, n_cube=None, dtype=torch.float32):
    """`dtype`: float32 = the reference as it runs (fixture and timing); float64 = the same modules, same float32-representable inputs,
    evaluated in double (torch's default dtype is switched so that every tensor the reference creates on the way is double too)."""
    cfg = STAGES[stage]
    N = cfg["N"]
    H = H or cfg.get("HW", 540); W = W or cfg.get("HW", 540)
    learn_cam = cfg.get("learn_cam", ("focal", "princ", "T"))
    n_cube = n_cube or cfg["n_cube"]
    torch.set_default_dtype(dtype)
    sdf = ref.network.getTmpSdf("cpu", 6, 0.6, 256)
    sdf.load_state_dict(fx.sphere_sdf_params(7), strict=True)
    tr = ref.Deformer.MLPTranslator(128, 6)
    tr.load_state_dict(fx.det_params(fx.DEF_SPEC, 202, last_scale=0.05), strict=True)
    skin = ref.Deformer.LBSkinner(fx.synthetic_lbs_volume(lbs_shape), fx.LBS_BMIN, fx.LBS_BMAX, fx.synthetic_joints(), np.array(orc.SMPL_PARENTS),
                                  init_pose=torch.from_numpy(ref.rutils.smpl_tmp_Apose(1)), align_corners=False)
    comp = ref.Deformer.CompositeDeformer([tr, skin])
    rn = ref.RenderNet.RenderingNetwork_view_norm(256, 'idr', 9, 3, [512, 512, 512, 512], True, multires_n=0, multires_v=4)
    rn.load_state_dict(fx.det_params(fx.REND_SPEC, 303), strict=True)
    for m in (sdf, comp, rn):
        m.to(dtype)

    class Seq:
        frame_num = F
        video_segmented_index = []

        def __init__(self):
            leaf = lambda t: t.to(dtype).clone().requires_grad_(True)
            self.poses = leaf(fx.det_tensor((F, 24, 3), 91, 0.12)); self.trans = leaf(fx.det_tensor((F, 3), 92, 0.04))
            self.conds = [leaf(fx.det_tensor((F, 128), 93, 0.1)), leaf(fx.det_tensor((F, 256), 94, 0.1))]
            cam = lambda name, t: leaf(t) if name in learn_cam else t.to(dtype)
            self.focal = cam("focal", torch.tensor([1.2 * W, 1.2 * W])); self.princ = cam("princ", torch.tensor([W / 2.0, H / 2.0])); self.T = cam("T", torch.tensor([0., 0.1, 2.4]))
            self.R = orc.quat2mat(torch.tensor([[0., 0., 1., 0.]])).to(dtype)

        def get_grad_parameters(self, idxs, device):
            return self.poses[idxs], self.trans[idxs], self.conds[0][idxs], self.conds[1][idxs]

        def get_camera_parameters(self, n, device):
            return self.focal.view(1, 2).expand(n, 2), self.princ.view(1, 2).expand(n, 2), self.R.expand(n, 3, 3), self.T.view(1, 3).expand(n, 3), H, W

        def get_batchframe_data(self, name, fids, batchsize):                     # dataset/dataset.py:128-147, unsegmented video
            data = getattr(self, name)
            starts = (fids - batchsize // 2).clamp(min=0, max=self.frame_num - batchsize)
            return data[starts.view(-1, 1) + torch.arange(0, batchsize).view(1, batchsize)], fids - starts
    ds = Seq()
    dirs, faces = fx.cube_sphere(n_cube)
    with torch.no_grad():
        r = torch.full((dirs.shape[0], 1), 0.6)
        for _ in range(30):
            r = r - torch.cat([sdf(part.to(dtype) * rp, 1.0)[:, 0:1] for part, rp in zip(torch.split(dirs, 20000), torch.split(r, 20000))])
    q = torch.round((r - 0.6) * 65536.).clamp(-32768, 32767).to(torch.int16)
    V0 = template_from_q(dirs, q)                  # float32 arithmetic whatever `dtype` is: the product builds the same template
    net = object.__new__(ref.network.OptimNetwork)
    torch.nn.Module.__init__(net)
    net.conf = gi.DictConf(cfg["conf"])
    net.sdf, net.deformer, net.netRender, net.dataset = sdf, comp, rn, ds
    mr, pr = gi.MaskRender(H, W, faces), gi.PcRender(H, W, cfg["radius"])
    net.maskRender, net.pcRender = mr, pr
    net.engine = None
    net.TmpVs, net.Tmpfs = V0.to(dtype).clone().requires_grad_(True), faces
    net.TmpOptimizer = torch.optim.SGD([net.TmpVs], lr=0.05, momentum=0.9)
    net.forward_time, net.remesh_intersect, net.remesh_time = 1, 30, 0.
    net.next_conf = net.next_train_conf = None
    net.draw, net.enable_mesh_color, net.sdfShrinkRadius = False, True, 0.0
    net.dctnull = ref.rutils.DCTNullSpace(10, 30).to(dtype)
    cam0 = ref.network.RectifiedPerspectiveCameras(*ds.get_camera_parameters(N, 'cpu')[:4], image_size=[(W, H)])
    net.angThred = cam0.angThreshold(0.5)
    fids = torch.tensor(cfg["fids"])
    datas = {'img': fx.det_tensor((N, H, W, 3), 95, 1.0), 'mask': mask_image(N, H, W), 'normal': fx.det_tensor((N, H, W, 3), 96, 1.0)}
    datas['normal'][:, ::5] = 0.
    datas = {k: v.to(dtype) for k, v in datas.items()}
    return net, ds, datas, fids, q, V0, faces


def template_from_q(dirs, q):
    """The template both sides build: float32 products / sums only (IEEE-exact, identical everywhere)."""
    r = 0.6 + q.to(torch.float32) / 65536.
    return dirs.to(torch.float32) * r + fx.det_tensor((dirs.shape[0], 3), 97, 0.004)


class DetDraws:
    """torch.rand / torch.randn_like replaced by det_tensor / det_normal keyed by call order."""

    def __init__(self):
        self.calls = []

    def rand(self, *size, **k):
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        self.calls.append(('rand', shape))
        return (fx.det_tensor(shape, DRAW_SEED0 + len(self.calls) - 1, 0.5) + 0.5).to(torch.get_default_dtype())      # (float32 values)

    def randn_like(self, x, **k):
        self.calls.append(('randn_like', tuple(x.shape)))
        return fx.det_normal(tuple(x.shape), DRAW_SEED0 + len(self.calls) - 1).to(x.dtype)


def draws_for(shapes):
    """The same draws from the product side: shapes[k] = shape of draw k (any length >= the reference's works for the head-sliced
    ones -- but det_tensor is a function of the flat index only for 1-D, so the exact shapes are stored in the fixture)."""
    kinds = ['rand', 'rand', 'randn_like', 'rand', 'rand', 'randn_like']
    names = ['ray_select', 'vert_select', 'eik_local', 'eik_global', 'vert_select2', 'regu_local']
    if len(shapes) == 5:                                   # fewer covered pixels than sample_pix * N: no Bernoulli ray selection (network.py:521)
        kinds, names = kinds[1:], names[1:]
    out = {}
    for k, (kind, name, shape) in enumerate(zip(kinds, names, shapes)):
        shape = tuple(int(s) for s in shape)
        out[name] = (fx.det_tensor(shape, DRAW_SEED0 + k, 0.5) + 0.5) if kind == 'rand' else fx.det_normal(shape, DRAW_SEED0 + k)
    return out


def one_iteration(net, ds, datas, fids, SP, timers=None):
    """forward + backward + propagateTmpPsGrad of the reference with deterministic draws; returns what the fixture stores."""
    draws = DetDraws()
    refined = {}
    real_rand, real_randn_like, real_refiner = torch.rand, torch.randn_like, ref.utils.OptimizeSurfacePs

    def rec_refiner(cam_pos, rays, p0, bi, *a, **k):
        refined.update(cam_pos=cam_pos.clone(), rays=rays.clone(), p0=p0.clone(), bi=bi.clone())
        t0 = time.perf_counter()
        p1, check = real_refiner(cam_pos, rays, p0, bi, *a, **k)
        if timers is not None:
            timers['refiner'] = timers.get('refiner', 0.) + time.perf_counter() - t0
        refined.update(p1=p1.detach().clone(), check=check.clone())
        return p1, check
    torch.rand, torch.randn_like = draws.rand, draws.randn_like
    ref.utils.OptimizeSurfacePs = rec_refiner
    real_float = torch.Tensor.float
    if torch.get_default_dtype() == torch.float64:          # the reference writes `.float()` for "index -> floating point" (network.py:536, ...):
        torch.Tensor.float = lambda self, *a, **k: self.to(torch.float64)      # in the double evaluation that must mean double
    try:
        loss = net(datas, SP, RATIO, fids)
        info = dict(net.info)
        loss.backward()
        g_tmpps = net.TmpPs.grad.clone()
        net.propagateTmpPsGrad(fids, RATIO)
    finally:
        torch.rand, torch.randn_like = real_rand, real_randn_like
        ref.utils.OptimizeSurfacePs = real_refiner
        torch.Tensor.float = real_float
    kinds = [k for k, _ in draws.calls]
    assert kinds in (['rand', 'rand', 'randn_like', 'rand', 'rand', 'randn_like'], ['rand', 'randn_like', 'rand', 'rand', 'randn_like']), draws.calls
    return loss, info, draws, refined, g_tmpps


class TimedRaster:
    """Wraps the two renderer stand-ins so that the third-party rasterisation (restated in numpy) can be reported separately."""

    def __init__(self, inner, timers, key):
        self.inner, self.timers, self.key = inner, timers, key
        self.rasterizer = inner.rasterizer

    def __call__(self, *a):
        t0 = time.perf_counter()
        out = self.inner(*a)
        self.timers[self.key] = self.timers.get(self.key, 0.) + time.perf_counter() - t0
        return out


def run(stage, do_time=False, small=False, dtype=torch.float64, write=True):
    torch.set_num_threads(os.cpu_count())
    cfg = STAGES[stage]
    t0 = time.perf_counter()
    kw = dict(H=48, W=48, lbs_shape=(7, 11, 9), n_cube=10) if small else {}
    net, ds, datas, fids, q, V0, faces = build(stage, dtype=dtype, **kw)
    print(f"[{stage}] scene built in {time.perf_counter() - t0:.1f} s: V = {V0.shape[0]}, F = {faces.shape[0]}", flush=True)
    timers = {}
    net.maskRender, net.pcRender = TimedRaster(net.maskRender, timers, 'raster_mesh'), TimedRaster(net.pcRender, timers, 'raster_points')
    t0 = time.perf_counter()
    loss, info, draws, refined, g_tmpps = one_iteration(net, ds, datas, fids, cfg["SP"], timers)
    wall = time.perf_counter() - t0
    print(f"[{stage}] reference iteration: {wall:.1f} s (rasterisers {timers.get('raster_mesh', 0):.1f} + {timers.get('raster_points', 0):.1f}, refiner {timers.get('refiner', 0):.1f})", flush=True)
    print({k: v for k, v in info.items() if k != 'pc_loss'}, info['pc_loss'], flush=True)
    assert info['rayInfo'][1] > 0.2 * info['rayInfo'][0], info['rayInfo']
    sp, tp, rp = dict(net.sdf.named_parameters()), dict(net.deformer.defs[0].named_parameters()), dict(net.netRender.named_parameters())
    # f at the MOVED template vertices (the argument of the L1 term 60 * mean |f(TmpVs)|, network.py:690-694; its gradient is sign(f) per
    # vertex): stored so that a test can tell a real mismatch from sign flips of |f| ~ 1e-6 entries.  float16 of 1024 f (|f| < 4e-3).
    with torch.no_grad():
        f_moved = torch.cat([net.sdf(part, RATIO)[:, 0] for part in torch.split(net.TmpVs.detach(), 20000)])
    arrs = dict(f_moved_x1024=(f_moved * 1024.).to(torch.float16), pc_weight=np.array(net.conf.get_float('pc_weight.weight')),
                dtype_bits=np.array(64 if dtype == torch.float64 else 32), n_cube=np.array(cfg["n_cube"] if not small else 10), q=q.view(-1), fids=fids, HW=np.array(datas['img'].shape[1:3]), SP=np.array(cfg["SP"]), radius=np.array(cfg["radius"]),
                ang_thr=np.array(net.angThred), frame_num=np.array(ds.frame_num), lbs_shape=np.array(net.deformer.defs[1].ws.shape[2:]),
                draw_shapes=np.array([list(s) + [0] * (2 - len(s)) for _, s in draws.calls]), loss=loss.detach(),
                ray_info=np.array(info['rayInfo']), inv_info=np.array(net.info['invInfo']),
                V_step=(net.TmpVs.detach() - V0)[::23], V_step_digest=param_digest(net.TmpVs.detach() - V0, 11),
                bi=net.batch_inds.to(torch.int16), rows=net.row_inds.to(torch.int16), cols=net.col_inds.to(torch.int16), g_TmpPs=g_tmpps,
                sel_bi=refined['bi'].to(torch.int16), sel_rays=refined['rays'], sel_p0=refined['p0'], sel_p1=refined['p1'], sel_check=refined['check'], cam_pos=refined['cam_pos'],
                **{'L_' + k: np.array(info[k]) for k in ('grad_loss', 'def_loss', 'dct_loss', 'color_loss', 'normal_loss', 'offset_loss', 'pc_loss_sdf') if k in info},
                L_mask_loss=np.array(info['pc_loss']['mask_loss']), L_defconst_loss=np.array(info['pc_loss']['defconst_loss']),
                normal_weight=np.array(net.conf.get_float('normal_weight')), learn_cam=np.array([int(t.requires_grad) for t in (ds.focal, ds.princ, ds.T)]),
                g_poses=ds.poses.grad, g_trans=ds.trans.grad, g_dcond=ds.conds[0].grad,
                **{'g_' + n: t.grad for n, t in (('focal', ds.focal), ('princ', ds.princ), ('T', ds.T)) if t.requires_grad})
    for tag, params in (("sdf", sp), ("tr", tp), ("rn", rp)):
        for k, (name, p) in enumerate(params.items()):
            arrs[f"d_{tag}.{name}"] = param_digest(p.grad, 100 * k)
            arrs[f"s_{tag}.{name}"] = slice_of(p.grad)
    conv = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    conv = {k: (v.astype(np.float32) if (v.dtype == np.float64 and v.size > 8) else v) for k, v in conv.items()}     # values of the double run, stored in single
    name = f"iteration_full_{stage}" + ("_small" if small else "")
    if not small and write:
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **conv)
        print("wrote", name, os.path.getsize(os.path.join(OUT, name + ".npz")), "bytes", flush=True)
    rec = None
    if do_time:
        times = []
        for rep in range(3):
            for p in list(net.parameters()) + [ds.poses, ds.trans, ds.conds[0], ds.conds[1], ds.focal, ds.princ, ds.T]:
                if p.requires_grad:
                    p.grad = None
            tm = {}
            net.maskRender.timers = net.pcRender.timers = tm
            t0 = time.perf_counter()
            one_iteration(net, ds, datas, fids, cfg["SP"], tm)
            w = time.perf_counter() - t0
            times.append((w, tm.get('raster_mesh', 0.) + tm.get('raster_points', 0.), tm.get('refiner', 0.)))
            print(f"[{stage}] timed run {rep}: {w:.1f} s (rasterisers {times[-1][1]:.1f}, refiner {times[-1][2]:.1f})", flush=True)
        times.sort(key=lambda t: t[0] - t[1])
        w, rs, rf = times[1]
        rec = {"stage": stage, "seconds_per_iteration": round(w - rs, 3), "seconds_per_iteration_with_numpy_rasterisers": round(w, 3),
               "refiner_seconds": round(rf, 3), "iterations_per_s": round(1.0 / (w - rs), 5), "template_vertices": int(V0.shape[0]),
               "rays": int(info['rayInfo'][0]), "rays_converged": int(info['rayInfo'][1]), "frames": cfg["N"]}
    return rec


if __name__ == "__main__":
    which = [a for a in sys.argv[1:] if not a.startswith("--")]
    stages = ["coarse", "fine"] if (not which or which[0] == "both") else [which[0]]           # (loose1080: by name)
    small = "--small" in sys.argv
    recs = []
    for st in stages:
        if "--no-golden" not in sys.argv:
            run(st, False, small, torch.float64 if "--f64" in sys.argv else torch.float32)     # the fixture: the reference as it runs (float32)
        if "--time" in sys.argv:
            recs.append(run(st, True, small, torch.float32, write=False))    # the baseline: the reference as it runs, float32
    if "--time" in sys.argv and not small:
        out = {"kind": "reference",
               "what": "the reference's own OptimNetwork.forward + loss.backward() + propagateTmpPsGrad (model/network.py:451-814) on the reference's own modules "
                       "(oracle/gen_fullsize_golden.py harness), full configs[1] size, CPU; the pytorch3d rasterisers (third-party, restated in numpy) are "
                       "excluded from seconds_per_iteration; remesh and the Adam step are not part of the call",
               "where": "build container", "cores": os.cpu_count(), "torch_threads": torch.get_num_threads(), "dtype": "f32", "protocol": "warm-up 1 + median of 3",
               "stages": [r for r in recs if r]}
        with open(os.path.join(ROOT, "profiles", "r03_cpu_reference.json"), "w") as fh:
            json.dump(out, fh, indent=1)
        print(json.dumps(out))
