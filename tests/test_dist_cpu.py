"""world_size-2 gloo tests (CPU) of the frame-parallel data-parallel path: frame sharding, the single flat
gradient all-reduce (incl. parameters that received no gradient on a rank and the dense per-frame tensors),
and the template-vertex gradient all-reduce.  The HIP kernels are not involved -- this checks the collective
logic bench.py runs over RCCL."""
import os
import socket
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from selfreconcode_amd import dist as srdist
    r, w, dev = srdist.init_from_env("cpu")
    assert (r, w) == (rank, world) and srdist.is_distributed()
    torch.manual_seed(0)                                   # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 1))
    unused = torch.nn.Parameter(torch.ones(3))             # receives no gradient anywhere (like rcond)
    frames = torch.arange(8)
    per_frame = torch.nn.Parameter(torch.zeros(8, 4))      # dense per-frame learnable (poses / codes)
    data = torch.randn(8, 5, 6, generator=torch.Generator().manual_seed(1))
    mine = srdist.shard_frames(frames, rank, world)
    assert mine.tolist() == list(range(rank, 8, world))
    loss = sum((net(data[f]) ** 2).mean() + (per_frame[f] - f).pow(2).sum() * 0.1 for f in mine) / len(mine)
    loss.backward()
    params = list(net.parameters()) + [unused, per_frame]
    # two buffers: `per_frame` stands for the gradients that are final early (render network) and is reduced asynchronously
    bucket = srdist.GradBucket(params, early=[per_frame])
    bucket.start_early()
    bucket.all_reduce_mean()
    tv = torch.full((5, 3), float(rank + 1))
    srdist.all_reduce_mean_(tv)
    # replicas are made identical by a broadcast, not by trusting the seeds
    drift = torch.nn.Parameter(torch.full((4,), float(rank)))
    srdist.GradBucket([drift]).sync_initial_state()
    assert drift.detach().tolist() == [0.0] * 4
    # pooled-mean weights (caveat B): ranks holding 1 and 2 points -> 2/3 and 4/3
    w = srdist.pooled_mean_weight(rank + 1, torch.device("cpu"))
    assert abs(float(w) - (rank + 1) * 2.0 / 3.0) < 1e-6
    srdist.assert_same_across_ranks(17, "vertex count")
    try:
        srdist.assert_same_across_ranks(17 + rank, "vertex count")
        raised = False
    except RuntimeError:
        raised = True
    assert raised
    try:
        srdist.all_reduce_mean_(None)
        raised = False
    except RuntimeError:
        raised = True
    assert raised
    # sharded replicated work (strong scaling): rank r of R evaluates chunk r of the remesh's query list and every rank ends up with the
    # same whole vector; the template term sums vertices r::R scaled by R / V, whose rank mean is the mean over all vertices
    assert srdist.shard_world() == (rank, world)
    for n in (11, 2, 1):                                   # ragged: the last chunk shorter, or empty
        full = torch.arange(n, dtype=torch.float32) * 0.5 - 2.0
        lo, hi, per = srdist.chunk_bounds(n, rank, world)
        got = srdist.all_gather_chunks(full[lo:hi].clone(), n, per)
        assert torch.equal(got, full), (n, got)
    f = torch.randn(37, generator=torch.Generator().manual_seed(3))
    part = f[rank::world].abs().sum() * (float(world) / f.numel())
    tot = part.clone(); dist.all_reduce(tot); tot /= world
    assert abs(float(tot) - float(f.abs().mean())) < 1e-6
    srdist.simulate_world((0, 8))
    assert srdist.shard_world() == (0, 8) and srdist.is_simulated()
    srdist.simulate_world(None)
    # by value (numpy): a tensor on a multiprocessing queue travels as a shared-memory handle that dies with this process
    out.put((rank, [p.grad.numpy().copy() for p in params], tv.numpy().copy()))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    res = [(r, [torch.from_numpy(g) for g in gs], torch.from_numpy(tv)) for r, gs, tv in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # reference: single process, full batch, mean of the per-rank means
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 1))
    per_frame = torch.nn.Parameter(torch.zeros(8, 4))
    data = torch.randn(8, 5, 6, generator=torch.Generator().manual_seed(1))
    loss = 0
    for r in range(world):
        mine = list(range(r, 8, world))
        loss = loss + sum((net(data[f]) ** 2).mean() + (per_frame[f] - f).pow(2).sum() * 0.1 for f in mine) / len(mine) / world
    loss.backward()
    ref = [p.grad for p in net.parameters()] + [torch.zeros(3), per_frame.grad]
    for rank, grads, tv in res:
        for a, b in zip(grads, ref):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(tv, torch.full((5, 3), 1.5))
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)                           # bit-identical on both ranks -> replicas stay in lock-step


def _worker4(rank, world, port, out):
    """Four ranks, ragged everything: V % R != 0 for the sharded template term, query lists shorter than the group, frames that do not
    divide over the ranks evenly, a rank with no frame at all, and the per-level count check that guards the all-gather."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from selfreconcode_amd import dist as srdist
    r, w, dev = srdist.init_from_env("cpu")
    assert (r, w) == (rank, world)
    # sharded template term: vertices r::R of V = 4k+1, 4k+2, 4k+3 and V < R
    for V in (41, 42, 43, 3):
        f = torch.randn(V, generator=torch.Generator().manual_seed(V))
        mine = f[rank::world]
        part = (mine.abs().sum() if mine.numel() else f.new_zeros(())) * (float(world) / V)
        tot = part.clone(); dist.all_reduce(tot); tot /= world
        assert abs(float(tot) - float(f.abs().mean())) < 1e-6, (V, float(tot))
    # remesh query chunks: lengths that leave the last ranks a short or EMPTY chunk
    for n in (4096 * 4 + 5, 9, 3, 1):
        full = torch.arange(n, dtype=torch.float32) * 0.25 - 1.0
        lo, hi, per = srdist.chunk_bounds(n, rank, world)
        assert 0 <= lo <= hi <= n and per * world >= n
        got = srdist.all_gather_chunks(full[lo:hi].clone(), n, per)
        assert torch.equal(got, full), n
    # the count check in front of the gather: equal counts pass, a replica that drifted by one query raises on EVERY rank
    srdist.assert_same_across_ranks(12345, "seg3d query count")
    try:
        srdist.assert_same_across_ranks(12345 + (1 if rank == 2 else 0), "seg3d query count")
        raised = False
    except RuntimeError:
        raised = True
    assert raised
    # frames: 6 frames over 4 ranks -> 2, 2, 1, 1; 3 frames -> the last rank has none and still joins every collective with zeros
    assert [srdist.shard_frames(torch.arange(6), r_, world).numel() for r_ in range(world)] == [2, 2, 1, 1]
    mine = srdist.shard_frames(torch.arange(3), rank, world)
    p = torch.nn.Parameter(torch.zeros(3, 2))
    if mine.numel():
        (p[mine] - 1.0).pow(2).sum().backward()
    bucket = srdist.GradBucket([p])
    bucket.all_reduce_mean()
    tv = None if mine.numel() else torch.zeros(5, 3)
    tv = srdist.all_reduce_mean_(torch.full((5, 3), float(rank)) if tv is None else tv)
    out.put((rank, p.grad.numpy().copy(), tv.numpy().copy()))
    dist.destroy_process_group()


def test_four_ranks_ragged_shards_and_the_count_guard():
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker4, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = torch.full((3, 2), -2.0 / 4)                    # d/dp (p - 1)^2 at 0 = -2 on the three owning ranks, mean over 4 ranks
    for rank, g, tv in res:
        torch.testing.assert_close(torch.from_numpy(g), want)
        torch.testing.assert_close(torch.from_numpy(tv), torch.full((5, 3), (0 + 1 + 2 + 0) / 4.0))


def _forced_worker(port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SR_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SR_DIST_FORCE_INIT"):
        os.environ.pop(k, None)
    from selfreconcode_amd import dist as srdist
    assert not srdist.is_distributed()
    assert srdist.pooled_mean_weight(7, torch.device("cpu")) is None            # no group: the pooled-mean weight is not applied at all
    srdist.force_collectives(True, "cpu")                                       # bench.py's strong_scaling_model: every collective live at world size 1
    assert srdist.is_distributed() and dist.get_world_size() == 1
    w = srdist.pooled_mean_weight(7, torch.device("cpu"))
    p = torch.nn.Parameter(torch.arange(6.).view(2, 3)); p.grad = torch.ones(2, 3) * 3
    q = torch.nn.Parameter(torch.zeros(4))                                      # no gradient on this rank
    b = srdist.GradBucket([p, q], early=[q])
    b.start_early(); b.all_reduce_mean()
    tv = torch.full((5, 3), 2.0); srdist.all_reduce_mean_(tv)
    srdist.assert_same_across_ranks(1234, "a count")
    srdist.force_collectives(False)
    ok = (not srdist.is_distributed()) and float(w) == 1.0 and torch.equal(p.grad, torch.ones(2, 3) * 3) and torch.equal(q.grad, torch.zeros(4)) and bool((tv == 2.0).all())
    srdist.force_collectives(True, "cpu")                                       # switching on again reuses the group
    ok = ok and srdist.is_distributed()
    out.put(ok)
    dist.destroy_process_group()


def test_forced_collectives_at_world_size_one():
    """dist.force_collectives (bench.py's `ms_one_rank_of_8_collectives_live_world1`): a one-rank group through the real backend (gloo here,
    RCCL in the bench), every collective of the step runs and is the identity; switching off leaves the group alive."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    p = ctx.Process(target=_forced_worker, args=(_free_port(), out))
    p.start(); p.join(120)
    assert p.exitcode == 0 and out.get(timeout=5) is True
