"""The quality pin at the size BASELINE.json configs[1] is quoted on: K = 32 consecutive training iterations of the PRODUCT -- 540 x 540,
3 frames x 2048 rays, the real 65 x 225 x 129 skinning-weight volume, an 85k-vertex template, Adam lr 1e-4 (the rate config.conf runs the
coarse stage at), one remesh on the shipped coarse grid (225 x 321 x 129) at iteration 12 -- against the REFERENCE'S OWN 32 iterations on
the same sequence (tests/golden/trajectory_full.npz, made by oracle/gen_trajectory_full_golden.py from the reference's modules, its own
Seg3dLossless and its own marching-cubes kernels; ~15 minutes of the build container's 8 cores, + its one-ulp twin up to the remesh).

The reference's only quantitative quality metric is the mask error 1 - IoU of the rasterised deformed template against the
ground-truth mask (infer.py:172-181, model/network.py:322-324) -- north_star's "matching silhouette IoU after equal iterations".
Free-running on both sides (the product's own refiner, its own remesh, its own rasterisers); asserted:
  * the remesh happens at the reference's iteration, vertex and face counts within 1 % (measured 0.13 %);
  * the mask error of EVERY frame of EVERY iteration: within 2e-3 of the reference's before the remesh (measured 8e-4) and within 0.01
    from the remesh on (measured 3.7e-3; a one-ulp twin of the product lands 3.7e-3 from the product, the reference's own one-ulp twin
    `twin_maskE_it` of the fixture lands at a similar distance from the reference) -- half of the 0.02 the round-4 review asked for;
  * the refiner's acceptance rate per block of 8 iterations within 0.03 of the reference's (measured 0.098 / 0.165 / 0.106 / 0.120 against
    0.099 / 0.165 / 0.111 / 0.119), > 0.9 on the iteration after the remesh on both sides, rays selected per iteration within 1 %;
  * maskE of `infer` on four other frames at the end within 0.01; the mean total loss of the last eight iterations within 10 %.

What this fixture taught: its first version started from the one-iteration fixture's template, which sits ON the
initial SDF's zero set with a handful of vertices at |f| < 3e-7.  The sign of f at those vertices -- the gradient of the L1 template
term -- is below what ANY float32 evaluation reproduces; ten of 84 968 came out on the other side, Adam's first steps are sign steps,
and the product ended 0.02-0.028 in maskE from the reference after the remesh (with either side reproducible against its own one-ulp
twin to 0.003-0.006, and a float64 run of the reference 1 % off the float32 reference after one step).  Moving the 1 510 vertices with
|f| < 4e-5 off the zero set by 3e-4 of their radius ON BOTH SIDES removed the coin tosses, and the two runs now track each other to
the twin level through 32 iterations and a remesh."""
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


def _consistent_observation(mask):
    """oracle/gen_trajectory_full_golden.py::consistent_observation, the same torch formulas (here on the GPU)."""
    H, W = mask.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=mask.device), torch.arange(W, dtype=torch.float32, device=mask.device), indexing='ij')
    u, v = xs / W, ys / H
    img = torch.stack([0.6 * torch.sin(6.2831853 * (1.0 * u + 3.0 * v)), 0.6 * torch.sin(6.2831853 * (2.0 * u + 1.0 * v) + 1.0),
                       0.6 * torch.sin(6.2831853 * (3.0 * u + 2.0 * v) + 2.0)], dim=-1)
    img = torch.where(mask[..., None] > 0, img, torch.ones_like(img))
    a, b = (u - 0.5) / 0.32, (v - 0.45) / 0.36
    c = torch.sqrt(torch.clamp(1.0 - a * a - b * b, min=0.04))
    n = torch.stack([a, -b, -c], dim=-1)
    n = n / n.norm(dim=-1, keepdim=True)
    return img, n * mask[..., None]


@pytest.mark.parametrize("stage", ["coarse", "fine", "consistent"])
def test_thirty_two_full_size_iterations_vs_the_references_own_run(golden, stage):
    """`fine`: the stage 189 of the reference's 201 epochs run in -- 1 frame x 6144 rays per iteration, loss_fine, a 173 402-vertex template,
    the remesh on 321 x 417 x 225, Adam at the MultiStepLR rate of epochs 80-129 (tests/golden/trajectory_full_fine.npz).
    `consistent` (round 6): the coarse stage on a scene the optimisation can CONVERGE on -- the ground-truth mask of a frame is the
    silhouette of the initial template scaled by 1.05 under that frame's pose (rendered by the reference's own deformer at fixture time and
    stored), colour / normal targets are smooth functions of the pixel, eight frames cycle through the batch
    (tests/golden/trajectory_full_consistent.npz, `oracle/gen_trajectory_full_golden.py --scene consistent`).  On it the reference's quality
    metric FALLS between the remeshes, and the product's falls with it: same per-frame mask errors, same decrease."""
    scene = "noise"
    if stage == "consistent":
        stage, scene = "coarse", "consistent"
    from selfreconcode_amd import mlp_engine
    from selfreconcode_amd.config import default_config
    from selfreconcode_amd.model.network import getTmpSdf
    from selfreconcode_amd.model.Deformer import MLPTranslator, LBSkinner, CompositeDeformer
    from selfreconcode_amd.model.RenderNet import RenderingNetwork_view_norm
    from selfreconcode_amd.model.optim_network import OptimNetwork
    from selfreconcode_amd.MCAcc import Seg3dLossless
    from selfreconcode_amd.utils import smpl_tmp_Apose, DCTNullSpace
    import os
    g = golden("trajectory_full_consistent" if scene == "consistent" else ("trajectory_full" if stage == "coarse" else "trajectory_full_fine"))
    NF = int(g["frames_per_iteration"]) if "frames_per_iteration" in g else 3
    H, W, F, K, SP = int(g["HW"][0]), int(g["HW"][1]), int(g["frame_num"]), int(g["K"]), int(g["SP"])
    REMESH_AT, BASE = int(g["remesh_at"]), int(g["draw_base"])
    assert (H, W, SP, K) == (540, 540, 2048, 32) and NF == (3 if stage == "coarse" else 1)
    ys, xs = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing='ij')
    mask1 = (((xs - W / 2.0) / (0.2963 * W)) ** 2 + ((ys - 0.45 * H) / (0.3426 * H)) ** 2 < 1.0).float().to(DEV)
    obs = {}
    if scene == "consistent":
        CF = [int(f) for f in g["cons_frames"]]
        cmask = {f: torch.from_numpy(np.unpackbits(g["cons_masks"][i].numpy())[:H * W].reshape(H, W).astype(np.float32)).to(DEV) for i, f in enumerate(CF)}
        frames_of = lambda k: [CF[k % 8], CF[(k + 3) % 8], CF[(k + 5) % 8]][:NF]
    else:
        frames_of = lambda k: [(7 + 3 * k) % F, (21 + 5 * k) % F, (30 + 7 * k) % F][:NF]

    def gt_masks(fids):
        if scene == "consistent":
            return torch.stack([cmask[int(f)] for f in fids.tolist()])
        return mask1[None].expand(fids.numel(), H, W).contiguous()

    def observations(fids):
        if scene == "consistent":
            io = [_consistent_observation(cmask[int(f)]) for f in fids.tolist()]
            return {'img': torch.stack([i for i, _ in io]), 'mask': gt_masks(fids), 'normal': torch.stack([n for _, n in io])}
        imgs, nrms = [], []
        for f in fids.tolist():
            if f not in obs:
                n = fx.det_tensor((H, W, 3), 9200 + f, 1.0)
                n[::5] = 0.
                obs[f] = (fx.det_tensor((H, W, 3), 9100 + f, 1.0).to(DEV), n.to(DEV))
            imgs.append(obs[f][0]); nrms.append(obs[f][1])
        return {'img': torch.stack(imgs), 'mask': mask1[None].expand(len(imgs), H, W).contiguous(), 'normal': torch.stack(nrms)}

    cover_remesh = []

    def run(twin):
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
        assert res[-1] == ((225, 321, 129) if stage == "coarse" else (321, 417, 225))        # the shipped grids (train.py:29-51)
        engine = Seg3dLossless(query_func=None, b_min=fx.LBS_BMIN, b_max=fx.LBS_BMAX, resolutions=res, align_corners=False, balance_value=0.0, use_cuda_impl=True).to(DEV)
        net = OptimNetwork(sdf, CompositeDeformer([tr, skin]).to(DEV), engine, None, rn, conf=default_config().get_config('loss_' + stage)).to(DEV)
        net.dataset = ds
        net.dctnull = DCTNullSpace(10, 30).to(DEV)
        net.point_radius, net.angThred = float(g["radius"]), float(g["ang_thr"])
        dirs, faces = fx.cube_sphere(int(g["n_cube"]))
        V0 = dirs * (0.6 + g["q"].float().view(-1, 1) / 65536.) + fx.det_tensor((dirs.shape[0], 3), 97, 0.004)
        assert V0.shape[0] == (84968 if stage == "coarse" else 173402)
        if "nudge_idx" in g:              # vertices whose |f| under the initial SDF is below what float32 reproduces: moved off the zero set on both sides (see the generator)
            ni = g["nudge_idx"].long()
            V0[ni] = V0[ni] * 1.0003
        Vstart = V0 * (1.0 + 1.2e-7) if twin else V0          # the twin: every template coordinate one float32 ulp away
        net.TmpVs, net.Tmpfs = Vstart.to(DEV).clone().requires_grad_(True), faces.to(DEV)
        net.TmpOptimizer = torch.optim.SGD([net.TmpVs], lr=0.05, momentum=0.9)
        net.remesh_intersect = int(g["remesh_intersect"]) if "remesh_intersect" in g else 30
        net.forward_time = net.remesh_intersect - REMESH_AT
        opt = torch.optim.Adam([{'params': ds.learnable_weights()}, {'params': [p for p in net.parameters() if p.requires_grad]}], lr=float(g["lr"]))
        mlp_engine.set_deferred_param_grads(True)
        rays, totals, maskE_it, remeshes = [], [], [], []
        mlp_engine.flush_param_grads()
        cover_remesh.clear()
        try:
            for k in range(K):
                fids = torch.tensor(frames_of(k), device=DEV)
                ratio = {'sdfRatio': 1., 'deformerRatio': k / 2500. + 0.5, 'renderRatio': 1.}
                before, dbg = net.TmpVs, {}
                opt.zero_grad(set_to_none=True)
                loss = net(observations(fids), SP, ratio, fids, rand=_draws(k, g["draw_shapes"][k].tolist(), BASE), debug=dbg)
                if net.TmpVs is not before:
                    v = net.TmpVs.detach()
                    remeshes.append((k, int(net.TmpVs.shape[0]), int(net.Tmpfs.shape[0]), v.mean(0).cpu().numpy(), (v - v.mean(0)).norm(dim=1).cpu().numpy(),
                                     v.amin(0).cpu().numpy(), v.amax(0).cpu().numpy(), v.cpu().numpy().astype(np.float64)))
                loss.backward()
                net.propagateTmpPsGrad(fids, ratio)
                opt.step()
                cover = (dbg['pix_to_face'][..., 0] >= 0).float()         # the silhouette `infer` rasterises (network.py:318-324), for the frames of this batch
                gtm = gt_masks(fids)
                if k == REMESH_AT:
                    cover_remesh.append(cover.bool().cpu().numpy())
                maskE_it.append((1. - (cover * gtm).view(NF, -1).sum(1) / (cover + gtm - cover * gtm).abs().view(NF, -1).sum(1)).tolist())
                rays.append((int(net.info['rayInfo'][0]), int(net.info['rayInfo'][1])))
                totals.append(float(loss.detach()))
                assert np.isfinite(totals[-1]), k
        finally:
            mlp_engine.set_deferred_param_grads(False)
        ef = g["eval_frames"].long().to(DEV)
        gts = {'mask': gt_masks(ef)}
        net.infer(net.TmpVs.detach(), net.Tmpfs, H, W, {'sdfRatio': 1., 'deformerRatio': K / 2500. + 0.5, 'renderRatio': 1.}, ef, notcolor=True, gts=gts)
        return np.array(rays, dtype=np.float64), np.array(totals), np.array(maskE_it), remeshes, np.asarray(gts['maskE'])

    rays, totals, maskE_it, remeshes, maskE = run(False)
    cover_product = cover_remesh[0].copy()
    rays_t, totals_t, maskE_it_t, remeshes_t, maskE_t = run(True)
    if "cover_at_remesh" in g:          # the reference's rasterised silhouettes of the remesh iteration, pixel by pixel
        ref_cover = np.unpackbits(g["cover_at_remesh"].numpy())[:NF * H * W].reshape(NF, H, W).astype(bool)
        if os.environ.get("SR_TRAJ_DUMP"):
            os.makedirs(os.environ["SR_TRAJ_DUMP"], exist_ok=True)
            np.savez_compressed(os.path.join(os.environ["SR_TRAJ_DUMP"], "covers.npz"), product=np.packbits(cover_product), reference=np.packbits(ref_cover))
        for n in range(NF):
            a, b = cover_product[n], ref_cover[n]
            only_p, only_r = a & ~b, b & ~a
            ys_, xs_ = np.nonzero(b)
            cy, cx = ys_.mean(), xs_.mean()
            def rad(m):
                yy, xx = np.nonzero(m)
                return np.round(np.percentile(np.hypot(yy - cy, xx - cx), [5, 50, 95]), 1).tolist() if yy.size else []
            print("frame %d at the remesh: silhouette pixels product %d reference %d; only product %d (radius pct %s), only reference %d (radius pct %s); reference silhouette radius pct %s; rows of only-product %s" % (
                n, a.sum(), b.sum(), only_p.sum(), rad(only_p), only_r.sum(), rad(only_r), rad(b), np.percentile(np.nonzero(only_p)[0], [5, 50, 95]).tolist() if only_p.any() else []))
    ref_rays = g["ray_counts"].numpy().astype(np.float64)
    ref_maskE_it = g["maskE_it"].numpy()
    # ---- the remesh
    assert [r[0] for r in remeshes] == [REMESH_AT] == [r[0] for r in remeshes_t], (remeshes, remeshes_t)
    Vr, Fr = int(g["remesh_nV"]), int(g["remesh_nF"])
    print("remesh at", remeshes[0][0], "vertices", remeshes[0][1], "(twin", remeshes_t[0][1], ") reference", Vr, "faces", remeshes[0][2], "reference", Fr)
    assert abs(remeshes[0][1] - Vr) <= 0.01 * Vr and abs(remeshes[0][2] - Fr) <= 0.01 * Fr
    rv = g["remesh_V"].numpy().astype(np.float64)                     # every 11th vertex of the reference's remeshed template (lattice-edge order)
    rc = rv.mean(0); rr = np.linalg.norm(rv - rc, axis=1)
    pr = remeshes[0][4]
    q = [5, 25, 50, 75, 95]
    print("remeshed surface: centroid product", np.round(remeshes[0][3], 4).tolist(), "reference", np.round(rc, 4).tolist())
    print("   distance to the centroid, percentiles", q, ": product", np.round(np.percentile(pr, q), 4).tolist(), "reference", np.round(np.percentile(rr, q), 4).tolist())
    print("   bounding box product", np.round(remeshes[0][5], 4).tolist(), np.round(remeshes[0][6], 4).tolist(), "reference", np.round(rv.min(0), 4).tolist(), np.round(rv.max(0), 4).tolist())
    # ---- the quality metric, per frame and iteration
    for k in range(K):
        print("k %2d  maskE product %s twin %s reference %s  rays %d/%d (reference %d/%d)  loss %.4f (reference %.4f)" % (
            k, np.round(maskE_it[k], 4).tolist(), np.round(maskE_it_t[k], 4).tolist(), np.round(ref_maskE_it[k], 4).tolist(),
            rays[k, 1], rays[k, 0], ref_rays[k, 1], ref_rays[k, 0], totals[k], float(g["L_total"][k])))
    dE, dT = np.abs(maskE_it - ref_maskE_it), np.abs(maskE_it - maskE_it_t)
    # Up to the remesh the silhouette is that of the SAME template moved by the mask loss; the remesh replaces the template by the SDF's
    # zero set after REMESH_AT Adam steps.  Bound from the remesh on: 0.01, or 3 x what a one-ulp twin of the product itself shows if
    # that is larger (it is not: 3.7e-3).
    bound_after = max(0.01, 3.0 * float(dT[REMESH_AT:].max()))
    if "twin_maskE_it" in g:
        tw = g["twin_maskE_it"].numpy(); n_ = tw.shape[0]
        print("the reference against ITS one-ulp twin: max |maskE difference| before the remesh %.5f, from it on %.4f (%d iterations); remesh vertices %d / %d" % (
            np.abs(tw[:REMESH_AT] - ref_maskE_it[:REMESH_AT]).max(), np.abs(tw[REMESH_AT:] - ref_maskE_it[REMESH_AT:n_]).max(), n_, int(g["twin_remesh_nV"]), Vr))
    print("max |maskE - reference|: before the remesh %.5f, after %.4f; product against its one-ulp twin: before %.5f, after %.4f -> bound after the remesh %.4f" % (
        dE[:REMESH_AT].max(), dE[REMESH_AT:].max(), dT[:REMESH_AT].max(), dT[REMESH_AT:].max(), bound_after))
    assert dE[:REMESH_AT].max() < 2e-3, float(dE[:REMESH_AT].max())
    assert dE[REMESH_AT:].max() < bound_after, (float(dE[REMESH_AT:].max()), bound_after)
    if stage == "coarse" and scene == "noise":
        assert ref_maskE_it[:REMESH_AT].mean() < 0.32 and ref_maskE_it[REMESH_AT:].mean() > 0.42    # (the fixture's own shape: the jump at the remesh is there to be matched)
    if scene == "consistent":
        # the metric FALLS while the template's SGD step chases a silhouette it can reach -- on both sides, by the same amount: mean over the
        # batch of iterations 2-4 against iterations 9-11 (before the remesh), and the last three iterations against the three after it
        fall = lambda e: (float(e[2:5].mean() - e[REMESH_AT - 3:REMESH_AT].mean()), float(e[REMESH_AT + 1:REMESH_AT + 4].mean() - e[-3:].mean()))
        fp, fr = fall(maskE_it), fall(ref_maskE_it)
        print("decrease of the mean mask error, before the remesh / after it: product %.4f / %.4f, reference %.4f / %.4f" % (fp + fr))
        assert fr[0] > 0.01 and fp[0] > 0.01, (fp, fr)                       # it does fall (the fixture's point) ...
        assert abs(fp[0] - fr[0]) < 0.25 * fr[0] + 1e-3, (fp, fr)              # ... by the same amount
        assert abs(fp[1] - fr[1]) < max(0.25 * abs(fr[1]), 0.01), (fp, fr)
    # ---- the refiner's acceptance rate at lr 1e-4
    for a in range(0, K, 8):
        mine = rays[a:a + 8, 1].sum() / rays[a:a + 8, 0].sum(); theirs = ref_rays[a:a + 8, 1].sum() / ref_rays[a:a + 8, 0].sum()
        print("iterations %2d-%2d: converged fraction %.3f (reference %.3f), rays per iteration %.0f (reference %.0f)" % (
            a, a + 7, mine, theirs, rays[a:a + 8, 0].mean(), ref_rays[a:a + 8, 0].mean()))
        assert abs(mine - theirs) < 0.03, (a, mine, theirs)
        assert abs(rays[a:a + 8, 0].mean() - ref_rays[a:a + 8, 0].mean()) < 0.01 * ref_rays[a:a + 8, 0].mean()
    assert rays[REMESH_AT, 1] / rays[REMESH_AT, 0] > 0.9 and ref_rays[REMESH_AT, 1] / ref_rays[REMESH_AT, 0] > 0.9
    # ---- the end state
    print("maskE of infer: product", np.round(maskE, 4).tolist(), "twin", np.round(maskE_t, 4).tolist(), "reference", np.round(g["maskE"].numpy(), 4).tolist())
    assert np.abs(maskE - g["maskE"].numpy()).max() < max(0.01, 3.0 * float(np.abs(maskE - maskE_t).max()))
    tail, ref_tail = float(np.mean(totals[-8:])), float(g["L_total"][-8:].mean())
    print("mean total loss of the last eight iterations: product %.4f, reference %.4f" % (tail, ref_tail))
    assert abs(tail - ref_tail) < 0.10 * ref_tail
    if scene == "consistent":
        stage = "consistent"
    import json
    rep = {"what": "tests/test_trajectory_full_gpu.py: 32 free-running full-size iterations (540 x 540, %s) against the reference's own run" % (
               "3 x 2048 rays, one remesh on 225 x 321 x 129, Adam lr 1e-4" if stage == "coarse" else
               "the same on the CONSISTENT scene: ground-truth masks = silhouettes of the template scaled by 1.05, smooth colour / normal targets" if stage == "consistent" else
               "fine stage: 1 x 6144 rays, one remesh on 321 x 417 x 225, Adam lr 3.7e-6"),
           "maskE_max_abs_diff_before_remesh": float(dE[:REMESH_AT].max()), "maskE_max_abs_diff_from_remesh_on": float(dE[REMESH_AT:].max()),
           "product_vs_its_one_ulp_twin": {"before": float(dT[:REMESH_AT].max()), "from_remesh_on": float(dT[REMESH_AT:].max())},
           "remesh_vertices": {"product": remeshes[0][1], "product_twin": remeshes_t[0][1], "reference": Vr},
           "converged_fraction_per_8_iterations": {"product": [float(rays[a:a + 8, 1].sum() / rays[a:a + 8, 0].sum()) for a in range(0, K, 8)],
                                                   "reference": [float(ref_rays[a:a + 8, 1].sum() / ref_rays[a:a + 8, 0].sum()) for a in range(0, K, 8)]},
           "maskE_infer_end": {"product": maskE.tolist(), "reference": g["maskE"].tolist()}, "tail_loss": {"product": tail, "reference": ref_tail},
           "maskE_per_iteration": {"product": np.round(maskE_it, 5).tolist(), "reference": np.round(ref_maskE_it, 5).tolist()}}
    d = os.environ.get("SR_PARITY_REPORT_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity")
    try:
        os.makedirs(d, exist_ok=True)
        if scene == "consistent":
            rep["decrease_of_mean_maskE_before_the_remesh_and_after_it"] = {"product": list(fp), "reference": list(fr)}
        with open(os.path.join(d, {"coarse": "quality_trajectory_full.json", "consistent": "quality_trajectory_full_consistent.json"}.get(stage, "quality_trajectory_full_fine.json")), "w") as fh:
            json.dump(rep, fh, indent=1)
    except OSError:
        pass
