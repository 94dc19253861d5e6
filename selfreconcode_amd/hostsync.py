"""Host round trips of the iteration (mask -> index list) without a blocking runtime wait.

`mask.nonzero()` is a count kernel, a 4-byte device-to-host copy and a *blocking* wait of the calling thread for that copy
(hipMemcpyWithStream).  Measured on MI355X / ROCm 7.2 inside the training step (tools/host_profile.py): when the host reaches such a
wait BEFORE the GPU has produced the count, the thread is released ~10 ms after the copy completed (the side stream's event shows the
result ready at 5.3 ms, the call returns at 18.2 ms), while the same call on an already finished stream returns at once.  Two of
the iteration's index lists (the seeds of the ray selection, the converged rays after the refiner) are always of the first kind.
Here the count goes to pinned memory with an asynchronous copy and the host POLLS the event behind it (hipEventQuery, no sleep in
the runtime), then asks for the index list with the size known (torch.nonzero_static: no synchronisation)."""
import os
import torch

POLL = os.environ.get("SR_HOST_POLL", "1") != "0"
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
    if not (POLL and mask.is_cuda):
        return mask.nonzero(as_tuple=as_tuple)
    n = count_to_host(mask.count_nonzero())
    idx = torch.nonzero_static(mask, size=n)
    return idx.unbind(1) if as_tuple else idx
