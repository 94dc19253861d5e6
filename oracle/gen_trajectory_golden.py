"""TEST INFRASTRUCTURE ONLY -- K = 20 consecutive training iterations of the REFERENCE, run verbatim on CPU, frozen into
tests/golden/trajectory.npz (build container only: needs /root/reference):    python oracle/gen_trajectory_golden.py

The loop is train.py:162-170 -- `optimizer.zero_grad(); loss = optNet(...); loss.backward(); optNet.propagateTmpPsGrad(...);
optimizer.step()` -- with Adam(lr 1e-4) over the dataset's learnable tensors and the three networks (train.py:139), the template's
SGD(momentum 0.9) inside forward (network.py:686-688), the annealing ratio of train.py:158-160, and ONE REMESH in the window
(`forward_time % remesh_intersect == 0` at the call with index 10, network.py:463-478): the reference's own Seg3dLossless (MCAcc/seg3d_lossless.py,
CPU) + the reference's own marching-cubes kernels (oracle/_ref/libmc_ref_fma.so, the host build of MCGpu/CudaKernels.cu) behind
`MCGpu.mc_gpu`, vertices put in lattice-edge order (the reference's order is whatever its atomics produce; the product's is that order).
Harness as oracle/gen_iteration_golden.py (pytorch3d renderers -> oracle/raster_oracle.py, CUDA extensions / torch_scatter -> pinned
restatements).  The random draws of iteration k are det_tensor / det_normal keyed by (k, call order), so the product regenerates them.

Stored per iteration: every loss term and the total, the selected rays (frame, row, column), the refiner's output for them; after the
remesh the new template; at the end `maskE` of `infer` (network.py:322-324: IoU error of the rasterised silhouette) on four frames, the
final template and digests of the final parameters.
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
from oracle import fixtures as fx  # noqa: E402
from oracle import torch_oracle as orc  # noqa: E402
from oracle import raster_oracle as ro  # noqa: E402
from oracle import mc as mco  # noqa: E402

ref = gi.ref
OUT = os.path.join(ROOT, "tests", "golden")
K, REMESH_AT = 20, 10
F, H, W, N, SP = 36, 64, 64, 2, 300
RES = [(15, 21, 9), (29, 41, 17), (57, 81, 33)]
LR = 1e-4
DRAW_BASE = 9000                     # draw c of iteration k: seed DRAW_BASE + 16 k + c
EVAL_FRAMES = [2, 11, 19, 30]


class MaskRender(gi.MaskRender):
    """as gi.MaskRender, but with the topology of the mesh it is handed (the template changes at the remesh)"""

    def __call__(self, meshes):
        self.faces = meshes._faces[0]
        return super().__call__(meshes)


def frames_of(k):
    return [(7 + 3 * k) % F, (21 + 5 * k) % F]


def ratio_of(k):
    return {'sdfRatio': 1., 'deformerRatio': k / 2500. + 0.5, 'renderRatio': 1.}


def observations(fids):
    """Per-frame observations keyed by the GLOBAL frame id (both sides rebuild them): noise colours / normals, a fixed elliptic mask."""
    img = torch.stack([fx.det_tensor((H, W, 3), 9100 + int(f), 1.0) for f in fids])
    nrm = torch.stack([fx.det_tensor((H, W, 3), 9200 + int(f), 1.0) for f in fids])
    nrm[:, ::5] = 0.
    return {'img': img, 'mask': gf.mask_image(len(fids), H, W), 'normal': nrm}


class Draws(gf.DetDraws):
    def __init__(self, k):
        super().__init__()
        self.k = k

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

        def learnable(self):                                            # dataset.learnable_weights(): codes, camera, poses, trans
            return [self.conds[0], self.conds[1], self.focal, self.princ, self.T, self.poses, self.trans]
    ds = Seq()
    engine = ref.MCAcc.Seg3dLossless(query_func=None, b_min=fx.LBS_BMIN, b_max=fx.LBS_BMAX, resolutions=RES, align_corners=False, balance_value=0.0, device='cpu',
                                     visualize=False, debug=False, use_cuda_impl=False, faster=False)

    remeshed = {}

    def mc_gpu(sdfs, xs, ys, zs, x0, y0, z0, iso):                       # MCGpu.mc_gpu (MCGpu/MCGpu.cpp:20-56) through the reference's kernels
        v, keys, f = mco.reference_marching_cubes(sdfs.numpy(), (float(xs), float(ys), float(zs)), (float(x0), float(y0), float(z0)), float(iso), mode="fma")
        v, keys, f = mco.canonical(v, keys, f)
        remeshed['V'], remeshed['F'] = torch.from_numpy(v.copy()), torch.from_numpy(f.copy())
        return [remeshed['V'].clone(), remeshed['F'].clone()]
    ref.network.MCGpu = types.SimpleNamespace(mc_gpu=mc_gpu)

    class _TriMesh:                                                      # openmesh.TriMesh: network.py:472-478 builds vertex->face tables nobody reads
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
    net.conf = gi.DictConf(gi.LOSS_COARSE)
    net.sdf, net.deformer, net.netRender, net.dataset = sdf, comp, rn, ds
    net.maskRender, net.pcRender = MaskRender(H, W, faces), gi.PcRender(H, W, 0.045)
    net.engine = engine
    net.TmpVs, net.Tmpfs = V0.clone().requires_grad_(True), faces
    net.TmpOptimizer = torch.optim.SGD([net.TmpVs], lr=0.05, momentum=0.9)
    net.forward_time, net.remesh_intersect, net.remesh_time = 30 - REMESH_AT, 30, 0.          # one remesh in the window: at the call with index REMESH_AT
    net.next_conf = net.next_train_conf = None
    net.draw, net.enable_mesh_color, net.sdfShrinkRadius = False, True, 0.0
    net.dctnull = ref.rutils.DCTNullSpace(10, 30)
    cam0 = ref.network.RectifiedPerspectiveCameras(*ds.get_camera_parameters(N, 'cpu')[:4], image_size=[(W, H)])
    net.angThred = cam0.angThreshold(0.5)
    optimizer = torch.optim.Adam([{'params': ds.learnable()}, {'params': [p for p in net.parameters() if p.requires_grad]}], lr=LR)

    real_rand, real_randn_like, real_refiner = torch.rand, torch.randn_like, ref.utils.OptimizeSurfacePs
    out = dict(q=q.view(-1), HW=np.array([H, W]), SP=np.array(SP), K=np.array(K), remesh_at=np.array(REMESH_AT), frame_num=np.array(F), lr=np.array(LR),
               radius=np.array(0.045), ang_thr=np.array(net.angThred), res=np.array(RES), eval_frames=np.array(EVAL_FRAMES), lbs_shape=np.array([17, 57, 33]))
    names = ('grad_loss', 'def_loss', 'dct_loss', 'color_loss', 'normal_loss', 'offset_loss', 'pc_loss_sdf')
    curve = {n: [] for n in names + ('mask_loss', 'defconst_loss', 'total')}
    ray_counts, draw_shapes = [], []
    for k in range(K):
        fids = torch.tensor(frames_of(k))
        draws, refined = Draws(k), {}

        def rec_refiner(cam_pos, rays, p0, bi, *a, **kw):
            p1, check = real_refiner(cam_pos, rays, p0, bi, *a, **kw)
            refined.update(p1=p1.detach().clone(), check=check.clone(), bi=bi.clone())
            return p1, check
        torch.rand, torch.randn_like = draws.rand, draws.randn_like
        ref.utils.OptimizeSurfacePs = rec_refiner
        cam_cls = ref.network.RectifiedPerspectiveCameras
        real_view_rays = cam_cls.view_rays
        pix = []

        def rec_view_rays(self, pixels, *a, **kw):                     # forward's call (network.py:536) comes first: the pixels of ALL selected rays
            pix.append(pixels.detach().clone())
            return real_view_rays(self, pixels, *a, **kw)
        cam_cls.view_rays = rec_view_rays
        try:
            optimizer.zero_grad()
            loss = net(observations(fids), SP, ratio_of(k), fids)
            has = net.TmpPs is not None
            rows, cols, bi_conv = (net.row_inds.clone(), net.col_inds.clone(), net.batch_inds.clone()) if has else (torch.zeros(0, dtype=torch.long),) * 3
            loss.backward()
            net.propagateTmpPsGrad(fids, ratio_of(k))
            optimizer.step()
        finally:
            torch.rand, torch.randn_like = real_rand, real_randn_like
            ref.utils.OptimizeSurfacePs = real_refiner
            cam_cls.view_rays = real_view_rays
        info = net.info
        for n in names:
            curve[n].append(float(info.get(n, float('nan'))) if not (n == 'color_loss' and float(info.get(n, -1.)) < 0) else float('nan'))       # no converged ray: no colour / normal term
        curve['mask_loss'].append(float(info['pc_loss']['mask_loss'])); curve['defconst_loss'].append(float(info['pc_loss']['defconst_loss']))
        curve['total'].append(float(loss))
        ray_counts.append(info['rayInfo'])
        draw_shapes.append([list(s) + [0] * (2 - len(s)) for _, s in draws.calls] + [[0, 0]] * (6 - len(draws.calls)))
        out[f"k{k}_p1"], out[f"k{k}_check"], out[f"k{k}_bi"] = refined['p1'], refined['check'], refined['bi'].to(torch.int16)
        assert pix[0].shape[0] == refined['bi'].shape[0]
        out[f"k{k}_rc"] = torch.stack([pix[0][:, 1], pix[0][:, 0]], 1).to(torch.int16)        # (row, column) of every selected ray, same order as k{k}_bi
        out[f"k{k}_conv_rc"] = torch.stack([bi_conv, rows, cols], 1).to(torch.int16)       # the converged rays' pixels (frame, row, column)
        if 'V' in remeshed and f"remesh_V" not in out:
            out["remesh_V"], out["remesh_F"], out["remesh_k"] = remeshed['V'], remeshed['F'].to(torch.int32), np.array(k)
        print(k, frames_of(k), "loss %.6f" % float(loss), info['rayInfo'], "V", net.TmpVs.shape[0], flush=True)
    assert "remesh_V" in out and int(out["remesh_k"]) == REMESH_AT
    # ---- the end state: maskE of `infer` (network.py:306-324) on EVAL_FRAMES, final template, parameter digests
    with torch.no_grad():
        ef = torch.tensor(EVAL_FRAMES)
        poses, trans, dcond, _ = ds.get_grad_parameters(ef, 'cpu')
        defV = comp(net.TmpVs.detach()[None].expand(len(EVAL_FRAMES), -1, 3), [dcond, [poses, trans]], ratio=ratio_of(K))
        xy, z = ro.ndc_projection(defV, ds.focal.detach(), ds.princ.detach(), ds.R[0], ds.T.detach(), W, H)
        p2f, _, _ = ro.rasterize_meshes(torch.cat([xy, z[..., None]], -1).float().numpy(), net.Tmpfs.numpy(), H, W)
        masks = torch.from_numpy((p2f >= 0)[..., 0]).float()
        gt = gf.mask_image(len(EVAL_FRAMES), H, W)
        n_ = len(EVAL_FRAMES)
        maskE = 1. - (masks * gt).view(n_, -1).sum(1) / (masks + gt - masks * gt).abs().view(n_, -1).sum(1)
    out.update(maskE=maskE, final_V=net.TmpVs.detach(), ray_counts=np.array(ray_counts), draw_shapes=np.array(draw_shapes),
               **{"L_" + n: np.array(v) for n, v in curve.items()})
    for tag, mod in (("sdf", sdf), ("tr", tr), ("rn", rn)):
        for i, (name, p) in enumerate(mod.named_parameters()):
            out[f"d_{tag}.{name}"] = gf.param_digest(p, 100 * i)
    out["final_poses"], out["final_trans"], out["final_dcond"] = ds.poses.detach(), ds.trans.detach(), ds.conds[0].detach()
    out["final_cam"] = torch.cat([ds.focal.detach(), ds.princ.detach(), ds.T.detach()])
    conv = {k_: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k_, v in out.items()}
    np.savez_compressed(os.path.join(OUT, "trajectory.npz"), **conv)
    print("wrote trajectory.npz", os.path.getsize(os.path.join(OUT, "trajectory.npz")), "bytes; maskE", maskE.tolist(), "losses", curve['total'])


if __name__ == "__main__":
    main()
