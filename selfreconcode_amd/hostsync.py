"""Host round trips of the iteration (bool mask -> index list).

`mask.nonzero()` is a count kernel, a small device-to-host copy and a blocking wait of the calling thread for that copy.  This module
is the one place the iteration's round trips go through (two per iteration with the fused selection, six without:
`nonzero_many` carries several counts in one), so that they can be traced (TRACE: tools/host_profile.py stamps the
device clock when the count copy is issued and the host clock when the count arrives).  (A polling form -- asynchronous copy to
pinned memory, hipEventQuery in a loop -- returned within the same ~20 us of the count being ready as torch's blocking wait and was
removed in round 6; what had looked like a 10 ms wake-up latency was the host running an iteration ahead, profiles/r03_host_vs_gpu.txt.)"""
import os
import torch

TRACE = None        # diagnostics (tools/host_profile.py): callable(label), called when the count copy has been issued and when the host has it
_pinned = {}


def _slot(device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _pinned.get(key)
    if buf is None:
        buf = _pinned[key] = torch.zeros(1, dtype=torch.int64).pin_memory()
    return buf


def count_to_host(count):
    """0-dim integer device tensor -> int, by async copy + event polling."""
    buf = _slot(count.device)
    buf.copy_(count.view(1), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    if TRACE is not None:
        TRACE('count copy issued')
    while not ev.query():
        pass
    if TRACE is not None:
        TRACE('count on the host')
    return int(buf[0])


def nonzero(mask, as_tuple=False):
    """mask.nonzero(as_tuple=...) for a bool CUDA tensor: same rows, same (lexicographic) order."""
    if not (TRACE is not None and mask.is_cuda):
        return mask.nonzero(as_tuple=as_tuple)
    n = count_to_host(mask.count_nonzero())
    idx = torch.nonzero_static(mask, size=n)
    return idx.unbind(1) if as_tuple else idx


def nonzero_many(masks, also=()):
    """Index lists (1-D int64) of several 1-D bool CUDA masks with ONE host round trip: the counts travel together, the lists are then
    made with their sizes known (torch.nonzero_static: no synchronisation).  Same rows, same order as mask.nonzero().view(-1).
    `also`: 0-dim integer device tensors that ride along; with them the result is (lists, their values as ints)."""
    counts = torch.stack([m.count_nonzero() for m in masks] + [a.to(torch.int64).view(()) for a in also])
    if TRACE is not None:
        TRACE('count copy issued')
    n = counts.tolist()
    if TRACE is not None:
        TRACE('count on the host')
    lists = [torch.nonzero_static(m, size=int(k)).view(-1) for m, k in zip(masks, n)]
    return (lists, [int(k) for k in n[len(masks):]]) if also else lists
