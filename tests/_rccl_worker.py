"""Worker of tests/test_dist_gpu.py::test_rccl_backend_world_size_one: every collective helper of selfreconcode_amd/dist.py through
the REAL backend ("nccl" = RCCL on ROCm, communicator bound to the device) at world size 1 -- what a one-GPU box can execute of the
path the driver's multi-GPU bench takes -- followed by two real training steps with the collectives active."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    from selfreconcode_amd import dist as srdist
    rank, world, device = srdist.init_from_env("cuda")
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl", (rank, world, dist.get_backend())
    assert srdist.is_distributed(), "SR_DIST_FORCE_INIT=1 must turn the collectives on at world size 1"
    desc = srdist.describe()
    assert desc["world"] == 1 and desc["devices"][0]["device_index"] == device.index and desc["backend"].startswith("rccl")
    # --- the helpers, one by one (values are those of a 1-rank mean: unchanged)
    a = torch.nn.Parameter(torch.arange(12., device=device).view(3, 4)); b = torch.nn.Parameter(torch.ones(5, device=device)); c = torch.nn.Parameter(torch.zeros(2, device=device))
    a.grad = torch.full_like(a, 2.0); b.grad = torch.full_like(b, -1.0)            # c: no gradient on this rank -> zeros travel
    bucket = srdist.GradBucket([a, b, c], early=[b])
    bucket.sync_initial_state(extra=[torch.ones(3, device=device)])
    bucket.start_early()
    bucket.all_reduce_mean()
    torch.cuda.synchronize()
    assert torch.equal(a.grad, torch.full_like(a, 2.0)) and torch.equal(b.grad, torch.full_like(b, -1.0)) and torch.equal(c.grad, torch.zeros_like(c))
    t = torch.full((7, 3), 3.5, device=device)
    srdist.all_reduce_mean_(t)
    assert torch.equal(t, torch.full((7, 3), 3.5, device=device))
    w = srdist.pooled_mean_weight(41, device)
    assert abs(float(w) - 1.0) < 1e-7
    srdist.assert_same_across_ranks(85111, "vertex count")
    full = torch.arange(10007, dtype=torch.float32, device=device) * 0.25 - 3.0        # the remesh's sharded query list: chunk -> all_gather_into_tensor
    lo, hi, per = srdist.chunk_bounds(full.numel(), rank, world)
    assert torch.equal(srdist.all_gather_chunks(full[lo:hi].clone(), full.numel(), per), full)
    gathers = []
    real_gather = srdist.all_gather_chunks
    srdist.all_gather_chunks = lambda mine, n, per: (gathers.append(n), real_gather(mine, n, per))[1]
    # --- the real step with the collectives live (template all-reduce inside forward, count check after the remesh, early + main buffers)
    from selfreconcode_amd import mlp_engine
    from selfreconcode_amd.optim import FusedAdam
    from selfreconcode_amd.synthetic import build_synthetic_scene
    net, ds, conf = build_synthetic_scene(device=device, frame_num=40, H=128, W=128, resolutions=[(15, 21, 9), (29, 41, 17), (57, 81, 33), (113, 161, 65)],
                                          lbs_volume_shape=(17, 57, 33))
    mlp_engine.set_deferred_param_grads(True)
    params = [p for p in net.parameters() if p.requires_grad]
    opt = FusedAdam([{'params': ds.learnable_weights()}, {'params': params}], lr=1e-4)
    bucket = srdist.GradBucket(list(ds.learnable_weights()) + params, early=list(net.netRender.parameters()) + [ds.conds[1]])
    bucket.sync_initial_state()
    ratio = {'sdfRatio': 1., 'deformerRatio': 0.5, 'renderRatio': 1.}
    losses = []
    for it in range(2):
        f = srdist.shard_frames(torch.tensor([3 + it, 11 + it, 20 + it], device=device), rank, world)
        opt.zero_grad(set_to_none=True)
        loss = net(ds.batch(f), 512, ratio, f)
        loss.backward()
        net.propagateTmpPsGrad(f, ratio, overlap=bucket)
        bucket.all_reduce_mean()
        opt.step()
        losses.append(float(loss))
    dist.barrier()
    torch.cuda.synchronize()
    assert all(l == l and abs(l) < 1e6 for l in losses), losses
    assert gathers and max(gathers) >= 4096, gathers                                    # the remesh of the first step went through the all-gather
    print(json.dumps({"ok": True, "rccl": desc, "losses": losses, "gathered_query_lists": gathers}), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
