"""Frame-parallel data parallelism: one process per GPU, every rank optimises its own
frames of the global batch; ONE flat all-reduce of all gradients per optimizer step (RCCL over xGMI
when the backend is "nccl"; "gloo" for the CPU tests).  The reference has no distributed code at all
(single process, single GPU); the semantics preserved here are listed in DESIGN.md "multi-GPU".
"""
import os
import torch
import torch.distributed as dist


PLACEMENT = None       # what affinity.bind did for this rank (init_from_env(bind_cpus=True)); bench.py prints it


def init_from_env(device_type="cuda", bind_cpus=False):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, world, device).
    `bind_cpus`: confine the rank's host threads to a group of cores of its GPU's NUMA node, one group per local rank
    (selfreconcode_amd/affinity.py: the step is ~2 600 launches issued by threads that hand over to each other; at one frame per rank the
    host paces it).  SIDE EFFECT: the mask narrows EVERY thread of the process, existing and future -- torch's intra-op / OpenMP pool (sized
    for the whole machine before the call) and DataLoader workers forked later share the 8 cores + SMT siblings.  Opt-in for that reason:
    a training loop with CPU-side data loading should leave it off, widen it (SR_BIND_CORES) or size its pools to the group
    (`torch.set_num_threads(len(dist.PLACEMENT["cpus"]))` if the record lists them)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type == "cuda":
        slot = local
        if os.environ.get("SR_ALL_RANKS_ON_DEVICE0") == "1":      # functional test of the N>1 path on a 1-GPU box
            local = 0
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
        if bind_cpus:
            from . import affinity
            global PLACEMENT
            PLACEMENT = affinity.bind(local, fallback_slot=slot)
    else:
        device = torch.device("cpu")
    global _FORCED
    force = os.environ.get("SR_DIST_FORCE_INIT") == "1"     # world size 1 through the real backend: every collective of the step runs
    if (world > 1 or force) and not dist.is_initialized():  # over RCCL on a one-GPU box (tests/test_dist_gpu.py, bench.py --rccl-selftest)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = os.environ.get("SR_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        kw = {"device_id": device} if backend == "nccl" else {}     # binds the communicator to this rank's GPU (RCCL)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        _FORCED = force and world == 1
    return rank, world, device


_FORCED = False


def force_collectives(on=True, device=None):
    """Run every collective of the step through the real backend at WORLD SIZE 1 (bench.py: what the gradient buckets, the template
    all-reduce and the small count / weight collectives cost in launches and copies when the wire time is zero).  Initialises a
    one-rank process group on first use (backend `nccl` = RCCL on a GPU, SR_DIST_BACKEND overrides); `on=False` switches the collectives
    off again and leaves the group alive.  No-op inside a real multi-rank group."""
    global _FORCED
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return
    if on and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        cuda = device is not None and torch.device(device).type == "cuda"
        backend = os.environ.get("SR_DIST_BACKEND", "nccl" if cuda else "gloo")
        kw = {"device_id": torch.device(device)} if backend == "nccl" else {}
        dist.init_process_group(backend=backend, rank=0, world_size=1, **kw)
    _FORCED = bool(on)


def is_distributed():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCED)


def describe():
    """What the collectives of this process run on: backend, world size and -- gathered over the ranks -- each rank's device
    (index, PCI bus id).  bench.py prints it so that an N-GPU record proves N ranks on N different devices over RCCL."""
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world": 1, "devices": None}
    backend = dist.get_backend()
    world = dist.get_world_size()
    if backend == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        props = torch.cuda.get_device_properties(dev)
        mine = torch.tensor([dev.index, int(getattr(props, "pci_bus_id", -1)), int(getattr(props, "pci_device_id", -1))], dtype=torch.int64, device=dev)
    else:
        mine = torch.tensor([torch.cuda.current_device() if torch.cuda.is_available() else -1, -1, -1], dtype=torch.int64)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return {"backend": "rccl (torch.distributed nccl)" if backend == "nccl" else backend, "world": world,
            "devices": [{"rank": r, "device_index": int(t[0]), "pci_bus_id": int(t[1]), "pci_device_id": int(t[2])} for r, t in enumerate(out)]}


class GradBucket:
    """All gradients of `tensors` as contiguous fp32 buffers -> one collective per buffer.
    15.2 MB of MLP parameters + the dense per-frame learnables (poses, trans, codes: every row moves
    through Adam's moments each step, so they are reduced densely).

    `early`: the subset whose gradients are final after the outer `loss.backward()` (render network, render codes:
    `propagateTmpPsGrad` does not touch them).  `start_early()` launches their all-reduce asynchronously so that it runs
    under the implicit-gradient pass; `all_reduce_mean()` reduces the rest and waits for both."""

    def __init__(self, tensors, early=()):
        early_ids = self.early_ids = {id(t) for t in early}
        self.tensors = [t for t in tensors if t.requires_grad]
        self.groups = [[t for t in self.tensors if id(t) in early_ids], [t for t in self.tensors if id(t) not in early_ids]]
        self.numel = sum(t.numel() for t in self.tensors)
        self.flat = [None, None]
        self.views = [None, None]
        self._pending = None

    def sync_initial_state(self, extra=()):
        """Rank 0's parameters (and `extra` tensors, e.g. buffers) to every rank: the frame-parallel scheme assumes bit-identical
        replicas (the remesh and the template all-reduce depend on it) and must not rely on identical seeding."""
        if not is_distributed():
            return
        with torch.no_grad():
            for t in list(self.tensors) + list(extra):
                dist.broadcast(t.data, src=0)

    def _gather(self, g):
        ts = self.groups[g]
        if not ts:
            return None
        dev = ts[0].device
        if self.flat[g] is None or self.flat[g].device != dev:
            self.flat[g] = torch.zeros(sum(t.numel() for t in ts), dtype=torch.float32, device=dev)
            self.views[g], off = [], 0
            for t in ts:
                n = t.numel()
                self.views[g].append(self.flat[g][off:off + n].view_as(t))
                off += n
        have = [(v, t.grad) for v, t in zip(self.views[g], ts) if t.grad is not None]
        if len(have) != len(ts):
            self.flat[g].zero_()                           # parameters without a gradient on this rank contribute zeros
        if have:
            torch._foreach_copy_([v for v, _ in have], [gr.to(torch.float32) if gr.dtype != torch.float32 else gr for _, gr in have])
        return self.flat[g]

    def _scatter(self, g):
        ts = self.groups[g]
        if not ts:
            return
        self.flat[g].div_(dist.get_world_size())
        for t in ts:
            if t.grad is None:
                t.grad = torch.empty_like(t)
        torch._foreach_copy_([t.grad for t in ts], self.views[g])

    def start_early(self):
        if not is_distributed() or not self.groups[0] or self._pending is not None:
            return
        flat = self._gather(0)
        self._pending = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def all_reduce_mean(self):
        """Gather (one multi-tensor copy) -> all-reduce -> scatter back (one multi-tensor copy) per buffer: ~5 launches instead of
        two per parameter tensor (~260), which is what the weak-scaling efficiency pays for on top of the collective itself."""
        if not is_distributed() or not self.tensors:
            return
        if self._pending is None and self.groups[0]:
            self.start_early()
        flat = self._gather(1)
        if flat is not None:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            self._scatter(1)
        if self._pending is not None:
            self._pending.wait()
            self._pending = None
            self._scatter(0)


def all_reduce_mean_(tensor):
    """In-place mean over ranks (used for the template-vertex gradient before its SGD step, caveat A).  Every rank must call it
    with a tensor of the same shape: a rank without a gradient passes zeros, never None (a skipped collective deadlocks the rest)."""
    if is_distributed():
        if tensor is None:
            raise RuntimeError("all_reduce_mean_: tensor is None on this rank; pass zeros so that every rank joins the collective")
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
        tensor.div_(dist.get_world_size())
    return tensor


def assert_same_across_ranks(value, what):
    """Raises on every rank if the integer `value` differs between ranks (e.g. the template's vertex count after a remesh:
    a 1-ulp divergence of the replicas changes it, and the template all-reduce would then hang or corrupt memory)."""
    if not is_distributed():
        return
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.full((2,), int(value), dtype=torch.int64, device=dev)      # (fills, not a blocking host-to-device copy)
    t[1].neg_()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t[0]) != -int(t[1]):
        raise RuntimeError(f"{what} differs between ranks (min {-int(t[1])}, max {int(t[0])}): replicas have diverged")


def pooled_mean_weight(count, device):
    """Factor that turns the mean of per-rank means into the mean over the points of ALL ranks for a term averaged over
    `count` local points (SURVEY.md 8(e) caveat B): n_r * R / sum_r n_r, as a device scalar (no host sync)."""
    if not is_distributed():
        return None
    t = torch.full((1,), float(count), dtype=torch.float32, device=device)     # (a fill, not a host-to-device copy: torch.tensor(..., device=cuda)
    tot = t.clone()                                                             # does not return before the current stream has drained)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return (t * dist.get_world_size() / tot.clamp(min=1.0)).squeeze(0)


def shard_frames(global_frame_ids, rank, world):
    """rank r takes frames batch[r::R] of the global batch."""
    return global_frame_ids[rank::world]


# ------------------------------------------------------------------------------------------------
# Sharding of the REPLICATED template-sized work (strong scaling, configs[2]: 8 frames over 8 GPUs).  Frames shard by construction; what
# every rank would otherwise repeat is the template term mean|f(TmpVs)| (network.py:690-694: an SDF forward + backward over all V
# vertices, ~8 ms of a 29 ms one-frame step) and the SDF queries of the remesh (network.py:292-302).  Both are sums over independent
# points whose gradients land in parameters that are all-reduced anyway, so rank r takes points r::R (template term) or the r-th
# contiguous chunk (queries) and the results are combined by the gradient all-reduce / one all-gather.
SHARD_TEMPLATE_TERMS = os.environ.get("SR_SHARD_TEMPLATE", "1") != "0"
_SIM_WORLD = None        # (rank, world) of a SIMULATED group: bench.py's "one rank of R" workload record on a single GPU (no collectives)


def simulate_world(rank_world):
    global _SIM_WORLD
    _SIM_WORLD = None if rank_world is None else (int(rank_world[0]), int(rank_world[1]))


def shard_world():
    """(rank, world) over which the replicated template-sized work is split; (0, 1) = not split."""
    if not SHARD_TEMPLATE_TERMS:
        return 0, 1
    if _SIM_WORLD is not None:
        return _SIM_WORLD
    if is_distributed():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def is_simulated():
    return _SIM_WORLD is not None


def chunk_bounds(n, rank, world):
    """Contiguous chunk `rank` of `n` items split into `world` equal parts (the last ones may be shorter / empty)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per), per


def all_gather_chunks(mine, n, per):
    """`mine`: this rank's chunk (chunk_bounds) of a length-n float vector -> the whole vector on every rank.  RCCL: one
    all_gather_into_tensor of equal, zero-padded chunks; other backends (gloo has no GPU all-gather): an all-reduce of a zero-filled
    buffer -- x + 0 is exact, so both give every rank the same bits."""
    world, rank = dist.get_world_size(), dist.get_rank()
    buf = torch.zeros(world * per, dtype=mine.dtype, device=mine.device)
    if dist.get_backend() == "nccl":
        pad = torch.zeros(per, dtype=mine.dtype, device=mine.device)
        pad[:mine.numel()] = mine.reshape(-1)
        dist.all_gather_into_tensor(buf, pad)
    else:
        buf[rank * per:rank * per + mine.numel()] = mine.reshape(-1)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    return buf[:n]
