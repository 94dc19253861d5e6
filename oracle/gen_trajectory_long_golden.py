"""TEST INFRASTRUCTURE ONLY -- K = 64 consecutive training iterations of the REFERENCE with FOUR remeshes and the coarse -> medium
stage switch, run verbatim on CPU and frozen into tests/golden/trajectory_long.npz (build container only: needs /root/reference):
    python oracle/gen_trajectory_long_golden.py

The loop is train.py:147-170 on the miniature sequence of oracle/gen_trajectory_golden.py (same networks, dataset stand-in, renderer
stand-ins, draws keyed by (iteration, call order), Adam lr 1e-4 = the rate config.conf runs its first ten epochs at):
  * coarse stage (loss_coarse, 3 frames per iteration, point radius 0.045, remesh every 12 calls: at k = 6, 18, 30);
  * at k = 24 -- an epoch boundary in train.py:148-152 -- `utils.set_hierarchical_config(conf, 'medium', ...)` (utils/utils.py:237-255):
    2 frames per iteration from now on, a new Seg3dLossless engine on the medium pyramid, `next_conf` / `next_train_conf` pending;
  * the pending configuration is adopted by `update_hierarchical_config` inside forward at the NEXT remesh (network.py:172-205,464):
    k = 30; from there loss_medium, radius 0.035, remesh every 20 calls (k = 50).
`update_hierarchical_config` of the reference builds pytorch3d renderer objects; here its ten lines run with the harness's renderer
stand-ins (same assignments: conf, forward_time = 0, point radius, remesh interval, sdfShrinkRadius = 0, pending configs cleared).

Stored per iteration: every loss term and the total, rayInfo (rays selected, rays the refiner accepted), template vertex count, the stage;
per remesh: iteration, vertex / face count, a strided vertex sample; at the end maskE of `infer` on four frames (network.py:322-324)
and parameter digests.  This is the reference's counterpart of bench.py's lr-1e-4 regime: the fraction of rays its own refiner accepts
while Adam runs at 1e-4, on a sequence both sides can run.
"""
import os
import sys
import types
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import gen_iteration_golden as gi  # noqa: E402
from oracle import gen_fullsize_golden as gf  # noqa: E402
from oracle import gen_trajectory_golden as gt  # noqa: E402
from oracle import fixtures as fx  # noqa: E402
from oracle import torch_oracle as orc  # noqa: E402
from oracle import raster_oracle as ro  # noqa: E402
from oracle import mc as mco  # noqa: E402

ref = gi.ref
OUT = os.path.join(ROOT, "tests", "golden")
K = 64
F, H, W, SP = 36, 64, 64, 300
FIRST_REMESH, SWITCH_AT = 6, 24
STAGE = {"coarse": dict(N=3, radius=0.045, remesh=12, res=[(15, 21, 9), (29, 41, 17), (57, 81, 33)], loss=gi.LOSS_COARSE),
         "medium": dict(N=2, radius=0.035, remesh=20, res=[(19, 25, 13), (37, 49, 25), (73, 97, 49)], loss=gf.LOSS_MEDIUM)}
LR = 1e-4
DRAW_BASE = 19000
EVAL_FRAMES = [2, 11, 19, 30]


def frames_of(k, n):
    return [(7 + 3 * k + 11 * j) % F for j in range(n)]


class Draws(gt.Draws):
    def rand(self, *size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        self.calls.append(('rand', shape))
        return fx.det_tensor(shape, DRAW_BASE + 16 * self.k + len(self.calls) - 1, 0.5) + 0.5

    def randn_like(self, x, **kw):
        self.calls.append(('randn_like', tuple(x.shape)))
        return fx.det_normal(tuple(x.shape), DRAW_BASE + 16 * self.k + len(self.calls) - 1)


def main():
    torch.set_num_threads(os.cpu_count())
    sdf = ref.network.getTmpSdf("cpu", 6, 0.6, 256)
    sdf.load_state_dict(fx.sphere_sdf_params(7), strict=True)
    tr = ref.Deformer.MLPTranslator(128, 6)
    tr.load_state_dict(fx.det_params(fx.DEF_SPEC, 202, last_scale=0.05), strict=True)
    skin = ref.Deformer.LBSkinner(fx.synthetic_lbs_volume((17, 57, 33)), fx.LBS_BMIN, fx.LBS_BMAX, fx.synthetic_joints(), np.array(orc.SMPL_PARENTS),
                                  init_pose=torch.from_numpy(ref.rutils.smpl_tmp_Apose(1)), align_corners=False)
    comp = ref.Deformer.CompositeDeformer([tr, skin])
    rn = ref.RenderNet.RenderingNetwork_view_norm(256, 'idr', 9, 3, [512, 512, 512, 512], True, multires_n=0, multires_v=4)
    rn.load_state_dict(fx.det_params(fx.REND_SPEC, 303), strict=True)

    class Seq:
        frame_num = F
        video_segmented_index = []

        def __init__(self):
            leaf = lambda t: t.clone().requires_grad_(True)
            self.poses = leaf(fx.det_tensor((F, 24, 3), 91, 0.12)); self.trans = leaf(fx.det_tensor((F, 3), 92, 0.04))
            self.conds = [leaf(fx.det_tensor((F, 128), 93, 0.1)), leaf(fx.det_tensor((F, 256), 94, 0.1))]
            self.focal = leaf(torch.tensor([1.2 * W, 1.2 * W])); self.princ = leaf(torch.tensor([W / 2.0, H / 2.0])); self.T = leaf(torch.tensor([0., 0.1, 2.4]))
            self.R = orc.quat2mat(torch.tensor([[0., 0., 1., 0.]]))

        def get_grad_parameters(self, idxs, device):
            return self.poses[idxs], self.trans[idxs], self.conds[0][idxs], self.conds[1][idxs]

        def get_camera_parameters(self, n, device):
            return self.focal.view(1, 2).expand(n, 2), self.princ.view(1, 2).expand(n, 2), self.R.expand(n, 3, 3), self.T.view(1, 3).expand(n, 3), H, W

        def get_batchframe_data(self, name, fids, batchsize):
            data = getattr(self, name)
            starts = (fids - batchsize // 2).clamp(min=0, max=self.frame_num - batchsize)
            return data[starts.view(-1, 1) + torch.arange(0, batchsize).view(1, batchsize)], fids - starts

        def learnable(self):
            return [self.conds[0], self.conds[1], self.focal, self.princ, self.T, self.poses, self.trans]
    ds = Seq()

    def make_engine(stage):
        return ref.MCAcc.Seg3dLossless(query_func=None, b_min=fx.LBS_BMIN, b_max=fx.LBS_BMAX, resolutions=STAGE[stage]["res"], align_corners=False, balance_value=0.0,
                                       device='cpu', visualize=False, debug=False, use_cuda_impl=False, faster=False)

    remeshes = []

    def mc_gpu(sdfs, xs, ys, zs, x0, y0, z0, iso):                       # MCGpu.mc_gpu through the reference's own kernels (host build)
        v, keys, f = mco.reference_marching_cubes(sdfs.numpy(), (float(xs), float(ys), float(zs)), (float(x0), float(y0), float(z0)), float(iso), mode="fma")
        v, keys, f = mco.canonical(v, keys, f)
        remeshes.append((torch.from_numpy(v.copy()), f.shape[0]))
        return [torch.from_numpy(v.copy()), torch.from_numpy(f.copy())]
    ref.network.MCGpu = types.SimpleNamespace(mc_gpu=mc_gpu)

    class _TriMesh:
        def __init__(self, v, f):
            self.n = len(v)

        def vertex_face_indices(self):
            return -np.ones((self.n, 1), np.int64)
    ref.network.om = types.SimpleNamespace(TriMesh=_TriMesh)

    dirs, faces = gi.icosphere(3)
    with torch.no_grad():
        r = torch.full((dirs.shape[0], 1), 0.6)
        for _ in range(30):
            r = r - sdf(dirs * r, 1.0)[:, 0:1]
    q = torch.round((r - 0.6) * 65536.).clamp(-32768, 32767).to(torch.int16)
    V0 = gf.template_from_q(dirs, q)
    net = object.__new__(ref.network.OptimNetwork)
    torch.nn.Module.__init__(net)
    net.conf = gi.DictConf(STAGE["coarse"]["loss"])
    net.sdf, net.deformer, net.netRender, net.dataset = sdf, comp, rn, ds
    net.maskRender, net.pcRender = gt.MaskRender(H, W, faces), gi.PcRender(H, W, STAGE["coarse"]["radius"])
    net.engine = make_engine("coarse")
    net.TmpVs, net.Tmpfs = V0.clone().requires_grad_(True), faces
    net.TmpOptimizer = torch.optim.SGD([net.TmpVs], lr=0.05, momentum=0.9)
    net.remesh_intersect = STAGE["coarse"]["remesh"]
    net.forward_time, net.remesh_time = net.remesh_intersect - FIRST_REMESH, 0.
    net.next_conf = net.next_train_conf = None
    net.draw, net.enable_mesh_color, net.sdfShrinkRadius = False, True, 0.0
    net.dctnull = ref.rutils.DCTNullSpace(10, 30)
    cam0 = ref.network.RectifiedPerspectiveCameras(*ds.get_camera_parameters(3, 'cpu')[:4], image_size=[(W, H)])
    net.angThred = cam0.angThreshold(0.5)
    stage_now = {"name": "coarse"}

    def update_hierarchical_config(device):                             # network.py:172-205 with the harness's renderer stand-ins
        if net.next_conf is not None:
            net.conf = net.next_conf
            net.forward_time = 0
            net.pcRender = gi.PcRender(H, W, net.next_train_conf['radius'])
            net.pcRender.rasterizer.cameras = net.maskRender.rasterizer.cameras          # `cameras=rasterizer.cameras` (network.py:186)
            net.remesh_intersect = net.next_train_conf['remesh']
            net.sdfShrinkRadius = 0.0
            net.next_conf = None
            net.next_train_conf = None
            stage_now["name"] = "medium"
    net.update_hierarchical_config = update_hierarchical_config
    optimizer = torch.optim.Adam([{'params': ds.learnable()}, {'params': [p for p in net.parameters() if p.requires_grad]}], lr=LR)

    real_rand, real_randn_like = torch.rand, torch.randn_like
    out = dict(q=q.view(-1), HW=np.array([H, W]), SP=np.array(SP), K=np.array(K), frame_num=np.array(F), lr=np.array(LR), first_remesh=np.array(FIRST_REMESH),
               switch_at=np.array(SWITCH_AT), ang_thr=np.array(net.angThred), eval_frames=np.array(EVAL_FRAMES), lbs_shape=np.array([17, 57, 33]),
               **{f"{s}_{k}": np.array(v) for s, d in STAGE.items() for k, v in d.items() if k != "loss"})
    names = ('grad_loss', 'def_loss', 'dct_loss', 'color_loss', 'normal_loss', 'offset_loss', 'pc_loss_sdf')
    curve = {n: [] for n in names + ('mask_loss', 'defconst_loss', 'total')}
    ray_counts, draw_shapes, vcount, stage_of, remesh_iters = [], [], [], [], []
    nbatch = STAGE["coarse"]["N"]
    for k in range(K):
        if k == SWITCH_AT:                                              # train.py:148-152 -> utils.set_hierarchical_config (utils/utils.py:237-255)
            nbatch = STAGE["medium"]["N"]
            net.next_conf = gi.DictConf(STAGE["medium"]["loss"])
            net.next_train_conf = {"radius": STAGE["medium"]["radius"], "remesh": STAGE["medium"]["remesh"]}
            net.engine = make_engine("medium")
        fids = torch.tensor(frames_of(k, nbatch))
        draws = Draws(k)
        nrem = len(remeshes)
        torch.rand, torch.randn_like = draws.rand, draws.randn_like
        try:
            optimizer.zero_grad()
            obs = gt.observations(fids)
            loss = net(obs, SP, gt.ratio_of(k), fids)
            loss.backward()
            net.propagateTmpPsGrad(fids, gt.ratio_of(k))
            optimizer.step()
        finally:
            torch.rand, torch.randn_like = real_rand, real_randn_like
        info = net.info
        for n in names:
            curve[n].append(float(info.get(n, float('nan'))) if not (n == 'color_loss' and float(info.get(n, -1.)) < 0) else float('nan'))
        curve['mask_loss'].append(float(info['pc_loss']['mask_loss'])); curve['defconst_loss'].append(float(info['pc_loss']['defconst_loss']))
        curve['total'].append(float(loss))
        ray_counts.append([int(info['rayInfo'][0]), int(info['rayInfo'][1])])
        draw_shapes.append([list(s) + [0] * (2 - len(s)) for _, s in draws.calls] + [[0, 0]] * (6 - len(draws.calls)))
        vcount.append(net.TmpVs.shape[0]); stage_of.append(0 if stage_now["name"] == "coarse" else 1)
        if len(remeshes) > nrem:
            remesh_iters.append(k)
            out[f"remesh{len(remesh_iters) - 1}_V"] = remeshes[-1][0][::7].clone()
        print(k, stage_now["name"], frames_of(k, nbatch), "loss %.6f" % float(loss), info['rayInfo'], "V", net.TmpVs.shape[0], flush=True)
    with torch.no_grad():
        ef = torch.tensor(EVAL_FRAMES)
        poses, trans, dcond, _ = ds.get_grad_parameters(ef, 'cpu')
        defV = comp(net.TmpVs.detach()[None].expand(len(EVAL_FRAMES), -1, 3), [dcond, [poses, trans]], ratio=gt.ratio_of(K))
        xy, z = ro.ndc_projection(defV, ds.focal.detach(), ds.princ.detach(), ds.R[0], ds.T.detach(), W, H)
        p2f, _, _ = ro.rasterize_meshes(torch.cat([xy, z[..., None]], -1).float().numpy(), net.Tmpfs.numpy(), H, W)
        masks = torch.from_numpy((p2f >= 0)[..., 0]).float()
        gtm = gf.mask_image(len(EVAL_FRAMES), H, W)
        n_ = len(EVAL_FRAMES)
        maskE = 1. - (masks * gtm).view(n_, -1).sum(1) / (masks + gtm - masks * gtm).abs().view(n_, -1).sum(1)
    out.update(maskE=maskE, ray_counts=np.array(ray_counts), draw_shapes=np.array(draw_shapes), vcount=np.array(vcount), stage_of=np.array(stage_of),
               remesh_iters=np.array(remesh_iters), remesh_V=np.array([r[0].shape[0] for r in remeshes]), remesh_F=np.array([r[1] for r in remeshes]),
               **{"L_" + n: np.array(v) for n, v in curve.items()})
    for tag, mod in (("sdf", sdf), ("tr", tr), ("rn", rn)):
        for i, (name, p) in enumerate(mod.named_parameters()):
            out[f"d_{tag}.{name}"] = gf.param_digest(p, 100 * i)
    conv = {k_: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k_, v in out.items()}
    np.savez_compressed(os.path.join(OUT, "trajectory_long.npz"), **conv)
    print("wrote trajectory_long.npz", os.path.getsize(os.path.join(OUT, "trajectory_long.npz")), "bytes; remeshes at", remesh_iters, "maskE", maskE.tolist())
    rc = np.array(ray_counts, dtype=np.float64)
    print("converged fraction per 16 iterations:", [round(float(rc[a:a + 16, 1].sum() / rc[a:a + 16, 0].sum()), 3) for a in range(0, K, 16)])


if __name__ == "__main__":
    main()
