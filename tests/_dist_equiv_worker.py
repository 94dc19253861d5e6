"""Worker of tests/test_dist_gpu.py::test_two_ranks_x_one_frame_equal_one_rank_x_two_frames (not a test module itself).

Runs ONE real training iteration of a small scene on the global batch of two frames, either as one rank holding both frames or as two
ranks holding one each (torch.distributed.run; both on device 0 through SR_ALL_RANKS_ON_DEVICE0), and dumps what must agree:
the template after its SGD step and every gradient after the all-reduce.

What makes the two runs comparable (SURVEY.md 8(e) caveats A/B):
  * terms that are per-frame means followed by a batch mean (mask IoU, deformation consistency, colour, normal, DCT) and the
    |f(TmpVs)| term (identical on every rank) are EXACTLY frame-separable: the mean over ranks of the per-rank gradients is the
    gradient of the two-frame batch -- caveat A (template) and the first half of caveat B;
  * the eikonal and deformation-regulariser terms are means over POOLED sample points whose composition depends on the batch a rank
    sees (its own rays + a vertex subset): they are not frame-separable by construction (second half of caveat B; the pooled weights
    n_r R / sum n_r are checked in tests/test_dist_cpu.py) and are switched off here (grad_weight = 0, def_regu.weight = 0);
  * no Bernoulli ray selection (sample_pix covers every silhouette pixel), the vertex-subset draws are injected, and the refiner's
    output is taken from the one-rank run for both (its |f| < 5e-5 acceptance flips on single ulps with the row's tile position).
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--inject", default=None)
    args = ap.parse_args()
    from selfreconcode_amd import dist as srdist, mlp_engine
    from selfreconcode_amd.config import default_config
    from selfreconcode_amd.synthetic import build_synthetic_scene, det_tensor, det_normal
    rank, world, device = srdist.init_from_env("cuda")
    conf = default_config()
    conf['loss_coarse']['grad_weight'] = 0.
    conf['loss_coarse']['def_regu']['weight'] = 0.
    net, ds, conf = build_synthetic_scene(device=device, frame_num=40, H=96, W=96, resolutions=[(15, 21, 9), (29, 41, 17), (57, 81, 33)],
                                          lbs_volume_shape=(17, 57, 33), conf=conf, consistent_masks=False)
    net.point_radius = 0.03
    with torch.no_grad():
        net.deformer.defs[0].lin4.weight.mul_(20.0)
    mlp_engine.set_deferred_param_grads(True)
    params = [p for p in net.parameters() if p.requires_grad]
    bucket = srdist.GradBucket(list(ds.learnable_weights()) + params, early=list(net.netRender.parameters()) + [ds.conds[1]])
    bucket.sync_initial_state()
    glob = torch.tensor([3, 11], device=device)
    fids = srdist.shard_frames(glob, rank, world)
    ratio = {'sdfRatio': 1., 'deformerRatio': 0.62, 'renderRatio': 1.}
    big = 40000
    rand = {'vert_select': det_tensor((big,), 42, 0.5) + 0.5, 'vert_select2': det_tensor((big,), 43, 0.5) + 0.5, 'eik_local': det_normal((big, 3), 44),
            'eik_global': det_tensor((big, 3), 45, 0.5) + 0.5, 'regu_local': det_normal((big, 3), 46)}
    rand = {k: v.to(device) for k, v in rand.items()}
    if args.inject:
        inj = np.load(args.inject)
        mine = torch.from_numpy(inj["frame"]) == int(fids[0])            # this rank's frame: its rays, in the same row-major order
        assert world == 2 and fids.numel() == 1
        rand['refined'] = (torch.from_numpy(inj["p"])[mine], torch.from_numpy(inj["ok"])[mine])
    datas = ds.batch(fids)                                # analytic per-frame silhouettes; the noise images of `batch` are seeded by the FIRST frame of
    for key, seed in (('img', 9000), ('normal', 9500)):   # the call, so give every frame its own observation whatever batch it is part of
        datas[key] = torch.stack([det_tensor((ds.H, ds.W, 3), seed + int(f), 1.0) for f in fids]).to(device)
    dbg = {}
    loss = net(datas, 100000, ratio, fids, rand=rand, debug=dbg)
    loss.backward()
    net.propagateTmpPsGrad(fids, ratio, overlap=bucket)
    bucket.all_reduce_mean()
    torch.cuda.synchronize()
    if rank == 0:
        out = {"TmpVs": net.TmpVs.detach().cpu().numpy(), "nrays": np.array(dbg['check'].numel()), "nconv": np.array(int(dbg['check'].sum()))}
        for tag, mod in (("sdf", net.sdf), ("tr", net.deformer.defs[0]), ("rn", net.netRender)):
            for n, p in mod.named_parameters():
                out[f"g_{tag}.{n}"] = p.grad.cpu().numpy()
        for n, t in (("poses", ds.poses), ("trans", ds.trans), ("dcond", ds.conds[0])) + tuple(ds.camera_params.items()):
            if t.grad is not None:
                out["g_" + n] = t.grad.cpu().numpy()
        if world == 1:
            out["frame"] = glob[dbg['batch_inds']].cpu().numpy(); out["p"] = dbg['initTmpPs'].cpu().numpy(); out["ok"] = dbg['check'].cpu().numpy()
        np.savez(args.out, **out)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
