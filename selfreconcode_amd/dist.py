"""Frame-parallel data parallelism: one process per GPU, every rank optimises its own
frames of the global batch; ONE flat all-reduce of all gradients per optimizer step (RCCL over xGMI
when the backend is "nccl"; "gloo" for the CPU tests).  The reference has no distributed code at all
(single process, single GPU); the semantics preserved here are listed in DESIGN.md "multi-GPU".
"""
import os
import torch
import torch.distributed as dist


def init_from_env(device_type="cuda"):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if device_type == "cuda":
        if os.environ.get("SR_ALL_RANKS_ON_DEVICE0") == "1":      # functional test of the N>1 path on a 1-GPU box
            local = 0
        torch.cuda.set_device(local)
        device = torch.device("cuda", local)
    else:
        device = torch.device("cpu")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        backend = os.environ.get("SR_DIST_BACKEND", "nccl" if device_type == "cuda" else "gloo")
        kw = {"device_id": device} if backend == "nccl" else {}     # binds the communicator to this rank's GPU (RCCL)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class GradBucket:
    """All gradients of `tensors` as ONE contiguous fp32 buffer -> one collective per step.
    15.2 MB of MLP parameters + the dense per-frame learnables (poses, trans, codes: every row moves
    through Adam's moments each step, so they are reduced densely)."""

    def __init__(self, tensors):
        self.tensors = [t for t in tensors if t.requires_grad]
        self.numel = sum(t.numel() for t in self.tensors)
        self.flat = None

    def all_reduce_mean(self):
        """Gather (one multi-tensor copy) -> all-reduce -> scatter back (one multi-tensor copy): ~5 launches per step instead of
        two per parameter tensor (~260), which is what the weak-scaling efficiency pays for on top of the collective itself."""
        if not is_distributed() or not self.tensors:
            return
        dev = self.tensors[0].device
        if self.flat is None or self.flat.device != dev:
            self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
            self.views, off = [], 0
            for t in self.tensors:
                n = t.numel()
                self.views.append(self.flat[off:off + n].view_as(t))
                off += n
        have = [(v, t.grad) for v, t in zip(self.views, self.tensors) if t.grad is not None]
        if len(have) != len(self.tensors):
            self.flat.zero_()                              # parameters without a gradient on this rank contribute zeros
        if have:
            torch._foreach_copy_([v for v, _ in have], [g.to(torch.float32) if g.dtype != torch.float32 else g for _, g in have])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        self.flat.div_(dist.get_world_size())
        for v, t in zip(self.views, self.tensors):
            if t.grad is None:
                t.grad = torch.empty_like(t)
        torch._foreach_copy_([t.grad for t in self.tensors], self.views)


def all_reduce_mean_(tensor):
    """In-place mean over ranks (used for the template-vertex gradient before its SGD step, caveat A)."""
    if is_distributed() and tensor is not None:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
        tensor.div_(dist.get_world_size())
    return tensor


def shard_frames(global_frame_ids, rank, world):
    """rank r takes frames batch[r::R] of the global batch."""
    return global_frame_ids[rank::world]
