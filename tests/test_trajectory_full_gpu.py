"""The quality pin at the size BASELINE.json configs[1] is quoted on: K = 32 consecutive training iterations of the PRODUCT -- 540 x 540,
3 frames x 2048 rays, the real 65 x 225 x 129 skinning-weight volume, an 85k-vertex template, Adam lr 1e-4 (the rate config.conf runs the
coarse stage at), one remesh on the shipped coarse grid (225 x 321 x 129) at iteration 12 -- against the REFERENCE'S OWN 32 iterations on
the same sequence (tests/golden/trajectory_full.npz, made by oracle/gen_trajectory_full_golden.py from the reference's modules, its own
Seg3dLossless and its own marching-cubes kernels; ~17 minutes of the build container's 8 cores).

The reference's only quantitative quality metric is the mask error 1 - IoU of the rasterised deformed template against the
ground-truth mask (infer.py:172-181, model/network.py:322-324) -- north_star's "matching silhouette IoU after equal iterations".  The
optimisation is chaotic on the scale of float32 rounding in WHICH rays the refiner accepts (tests/test_trajectory_gpu.py), so nothing is
compared ray by ray; asserted, free-running (the product's own refiner, its own remesh, its own rasterisers):
  * the remesh happens at the reference's iteration, vertex and face counts within 1 %;
  * the mask error of EVERY frame of EVERY iteration within 0.02 of the reference's value for that frame and iteration (it moves from
    0.27 to 0.52 across the remesh in this synthetic scene: a template initialised off the SDF's zero set, and an elliptic target mask);
  * the refiner's acceptance rate per block of 8 iterations within 0.10 of the reference's; > 0.9 on the iteration after the remesh on
    both sides; rays selected per iteration within 3 %;
  * maskE of `infer` on four other frames at the end within 0.02; the mean total loss of the last eight iterations within 10 %."""
import numpy as np
import pytest
import torch
from oracle import torch_oracle as orc
from oracle import fixtures as fx

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _draws(k, shapes, base):
    shapes = [tuple(int(x) for x in s if int(x) > 0) for s in shapes if int(s[0]) > 0]
    kinds = ['rand', 'rand', 'randn_like', 'rand', 'rand', 'randn_like']
    names = ['ray_select', 'vert_select', 'eik_local', 'eik_global', 'vert_select2', 'regu_local']
    if len(shapes) == 5:
        kinds, names = kinds[1:], names[1:]
    out = {}
    for c, (kind, name, shape) in enumerate(zip(kinds, names, shapes)):
        shape = (shape[0] + 8192,) + tuple(shape[1:])       # spare rows: free-running, the product's counts differ from the reference's
        out[name] = ((fx.det_tensor(shape, base + 16 * k + c, 0.5) + 0.5) if kind == 'rand' else fx.det_normal(shape, base + 16 * k + c)).to(DEV)
    return out


def test_thirty_two_full_size_iterations_vs_the_references_own_run(golden):
    from selfreconcode_amd import mlp_engine
    from selfreconcode_amd.config import default_config
    from selfreconcode_amd.model.network import getTmpSdf
    from selfreconcode_amd.model.Deformer import MLPTranslator, LBSkinner, CompositeDeformer
    from selfreconcode_amd.model.RenderNet import RenderingNetwork_view_norm
    from selfreconcode_amd.model.optim_network import OptimNetwork
    from selfreconcode_amd.MCAcc import Seg3dLossless
    from selfreconcode_amd.utils import smpl_tmp_Apose, DCTNullSpace
    g = golden("trajectory_full")
    H, W, F, K, SP = int(g["HW"][0]), int(g["HW"][1]), int(g["frame_num"]), int(g["K"]), int(g["SP"])
    REMESH_AT, BASE = int(g["remesh_at"]), int(g["draw_base"])
    assert (H, W, SP, K) == (540, 540, 2048, 32)
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    mask1 = (((xs - W / 2.0) / (0.2963 * W)) ** 2 + ((ys - 0.45 * H) / (0.3426 * H)) ** 2 < 1.0).float().to(DEV)
    obs = {}

    def observations(fids):
        imgs, nrms = [], []
        for f in fids.tolist():
            if f not in obs:
                n = fx.det_tensor((H, W, 3), 9200 + f, 1.0)
                n[::5] = 0.
                obs[f] = (fx.det_tensor((H, W, 3), 9100 + f, 1.0).to(DEV), n.to(DEV))
            imgs.append(obs[f][0]); nrms.append(obs[f][1])
        return {'img': torch.stack(imgs), 'mask': mask1[None].expand(len(imgs), H, W).contiguous(), 'normal': torch.stack(nrms)}

    sdf = getTmpSdf(DEV, 6, 0.6, 256)
    sdf.load_state_dict(fx.sphere_sdf_params(7), strict=True)
    tr = MLPTranslator(128, 6).to(DEV)
    tr.load_state_dict(fx.det_params(fx.DEF_SPEC, 202, last_scale=0.05), strict=True)
    rn = RenderingNetwork_view_norm(256, 'idr', 9, 3, [512, 512, 512, 512], True, multires_n=0, multires_v=4).to(DEV)
    rn.load_state_dict(fx.det_params(fx.REND_SPEC, 303), strict=True)
    skin = LBSkinner(fx.synthetic_lbs_volume(tuple(int(s) for s in g["lbs_shape"])), fx.LBS_BMIN, fx.LBS_BMAX, fx.synthetic_joints(), np.array(fx.SMPL_PARENTS),
                     init_pose=torch.from_numpy(smpl_tmp_Apose(1)), align_corners=False).to(DEV)
    leaf = lambda t: t.to(DEV).clone().requires_grad_(True)

    class Seq:                                                        # the accessors of dataset/dataset.py:76-81,117-147
        frame_num = F
        poses, trans = leaf(fx.det_tensor((F, 24, 3), 91, 0.12)), leaf(fx.det_tensor((F, 3), 92, 0.04))
        conds = [leaf(fx.det_tensor((F, 128), 93, 0.1)), leaf(fx.det_tensor((F, 256), 94, 0.1))]
        camera_params = {'focal_length': leaf(torch.tensor([1.2 * W, 1.2 * W])), 'princeple_points': leaf(torch.tensor([W / 2.0, H / 2.0])),
                         'world2cam_coord_trans': leaf(torch.tensor([0., 0.1, 2.4]))}
        R = orc.quat2mat(torch.tensor([[0., 0., 1., 0.]]))[0].to(DEV)

        def get_grad_parameters(self, idxs, device=None):
            return self.poses[idxs], self.trans[idxs], self.conds[0][idxs], self.conds[1][idxs]

        def get_camera_parameters(self, N, device=None):
            c = self.camera_params
            return (c['focal_length'].view(1, 2).expand(N, 2), c['princeple_points'].view(1, 2).expand(N, 2), self.R.view(1, 3, 3).expand(N, 3, 3),
                    c['world2cam_coord_trans'].view(1, 3).expand(N, 3), H, W)

        def get_batchframe_data(self, name, fids, batchsize):
            data = getattr(self, name)
            starts = (fids - batchsize // 2).clamp(min=0, max=self.frame_num - batchsize)
            return data[starts.view(-1, 1) + torch.arange(0, batchsize, device=fids.device).view(1, batchsize)], fids - starts

        def learnable_weights(self):
            return [self.conds[0], self.conds[1]] + list(self.camera_params.values()) + [self.poses, self.trans]
    ds = Seq()
    res = [tuple(int(x) for x in r) for r in g["res"]]
    assert res[-1] == (225, 321, 129)                                # the shipped coarse grid (train.py:29-37)
    engine = Seg3dLossless(query_func=None, b_min=fx.LBS_BMIN, b_max=fx.LBS_BMAX, resolutions=res, align_corners=False, balance_value=0.0, use_cuda_impl=True).to(DEV)
    net = OptimNetwork(sdf, CompositeDeformer([tr, skin]).to(DEV), engine, None, rn, conf=default_config().get_config('loss_coarse')).to(DEV)
    net.dataset = ds
    net.dctnull = DCTNullSpace(10, 30).to(DEV)
    net.point_radius, net.angThred = float(g["radius"]), float(g["ang_thr"])
    dirs, faces = fx.cube_sphere(int(g["n_cube"]))
    V0 = dirs * (0.6 + g["q"].float().view(-1, 1) / 65536.) + fx.det_tensor((dirs.shape[0], 3), 97, 0.004)
    assert V0.shape[0] == 84968
    net.TmpVs, net.Tmpfs = V0.to(DEV).clone().requires_grad_(True), faces.to(DEV)
    net.TmpOptimizer = torch.optim.SGD([net.TmpVs], lr=0.05, momentum=0.9)
    net.remesh_intersect = 30
    net.forward_time = 30 - REMESH_AT
    opt = torch.optim.Adam([{'params': ds.learnable_weights()}, {'params': [p for p in net.parameters() if p.requires_grad]}], lr=float(g["lr"]))
    mlp_engine.set_deferred_param_grads(True)
    rays, totals, maskE_it, remeshes = [], [], [], []
    try:
        for k in range(K):
            fids = torch.tensor([(7 + 3 * k) % F, (21 + 5 * k) % F, (30 + 7 * k) % F], device=DEV)
            ratio = {'sdfRatio': 1., 'deformerRatio': k / 2500. + 0.5, 'renderRatio': 1.}
            before, dbg = net.TmpVs, {}
            opt.zero_grad(set_to_none=True)
            loss = net(observations(fids), SP, ratio, fids, rand=_draws(k, g["draw_shapes"][k].tolist(), BASE), debug=dbg)
            if net.TmpVs is not before:
                remeshes.append((k, int(net.TmpVs.shape[0]), int(net.Tmpfs.shape[0])))
            loss.backward()
            net.propagateTmpPsGrad(fids, ratio)
            opt.step()
            cover = (dbg['pix_to_face'][..., 0] >= 0).float()         # the silhouette `infer` rasterises (network.py:318-324), for the frames of this batch
            gtm = mask1[None].expand(3, H, W)
            maskE_it.append((1. - (cover * gtm).view(3, -1).sum(1) / (cover + gtm - cover * gtm).abs().view(3, -1).sum(1)).tolist())
            rays.append((int(net.info['rayInfo'][0]), int(net.info['rayInfo'][1])))
            totals.append(float(loss.detach()))
            assert np.isfinite(totals[-1]), k
    finally:
        mlp_engine.set_deferred_param_grads(False)
    rays = np.array(rays, dtype=np.float64); ref_rays = g["ray_counts"].numpy().astype(np.float64)
    maskE_it = np.array(maskE_it); ref_maskE_it = g["maskE_it"].numpy()
    # ---- the remesh
    assert [r[0] for r in remeshes] == [REMESH_AT], remeshes
    Vr, Fr = int(g["remesh_nV"]), int(g["remesh_nF"])
    print("remesh at", remeshes[0][0], "vertices", remeshes[0][1], "reference", Vr, "faces", remeshes[0][2], "reference", Fr)
    assert abs(remeshes[0][1] - Vr) <= 0.01 * Vr and abs(remeshes[0][2] - Fr) <= 0.01 * Fr
    # ---- the quality metric, per frame and iteration
    dE = np.abs(maskE_it - ref_maskE_it)
    for a in range(0, K, 8):
        print("iterations %2d-%2d: maskE product %s reference %s, max |diff| %.4f" % (a, a + 7, np.round(maskE_it[a:a + 8].mean(0), 4).tolist(),
                                                                                  np.round(ref_maskE_it[a:a + 8].mean(0), 4).tolist(), dE[a:a + 8].max()))
    assert dE.max() < 0.02, (float(dE.max()), np.unravel_index(dE.argmax(), dE.shape))
    assert ref_maskE_it[:REMESH_AT].mean() < 0.32 and ref_maskE_it[REMESH_AT:].mean() > 0.42        # (the fixture's own shape: the jump at the remesh is there to be matched)
    # ---- the refiner's acceptance rate at lr 1e-4
    for a in range(0, K, 8):
        mine = rays[a:a + 8, 1].sum() / rays[a:a + 8, 0].sum(); theirs = ref_rays[a:a + 8, 1].sum() / ref_rays[a:a + 8, 0].sum()
        print("iterations %2d-%2d: converged fraction %.3f (reference %.3f), rays per iteration %.0f (reference %.0f)" % (
            a, a + 7, mine, theirs, rays[a:a + 8, 0].mean(), ref_rays[a:a + 8, 0].mean()))
        assert abs(mine - theirs) < 0.10, (a, mine, theirs)
        assert abs(rays[a:a + 8, 0].mean() - ref_rays[a:a + 8, 0].mean()) < 0.03 * ref_rays[a:a + 8, 0].mean()
    assert rays[REMESH_AT, 1] / rays[REMESH_AT, 0] > 0.9 and ref_rays[REMESH_AT, 1] / ref_rays[REMESH_AT, 0] > 0.9
    # ---- the end state
    ef = g["eval_frames"].long().to(DEV)
    gts = {'mask': mask1[None].expand(ef.numel(), H, W).contiguous()}
    net.infer(net.TmpVs.detach(), net.Tmpfs, H, W, {'sdfRatio': 1., 'deformerRatio': K / 2500. + 0.5, 'renderRatio': 1.}, ef, notcolor=True, gts=gts)
    maskE = np.asarray(gts['maskE'])
    print("maskE of infer: product", np.round(maskE, 4).tolist(), "reference", np.round(g["maskE"].numpy(), 4).tolist())
    assert np.abs(maskE - g["maskE"].numpy()).max() < 0.02
    tail, ref_tail = float(np.mean(totals[-8:])), float(g["L_total"][-8:].mean())
    print("mean total loss of the last eight iterations: product %.4f, reference %.4f" % (tail, ref_tail))
    assert abs(tail - ref_tail) < 0.10 * ref_tail
