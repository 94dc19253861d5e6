"""MI355X-native hot path of SelfRecon."""
import os as _os

if _os.environ.get("SR_AUTOGRAD_CALLER_THREAD", "0") == "1":
    # Opt-in experiment (round 6): run autograd traversals on the calling thread instead of handing each of the iteration's nine
    # traversals to the engine's device thread.  Saves 0.2 - 1.4 ms of thread hand-overs per step where the host paces the step (one
    # frame per rank), nothing at three frames.  NOT the default: with the worker thread the first node handed over starts at once,
    # on the calling thread all ready nodes are queued first and run in strict priority order -- a different (equally valid) order of
    # the first accumulations of a traversal, i.e. sums that differ in the last bit, and the long free-running parity trajectories
    # (tests/test_trajectory_full_gpu.py) are pinned to the default order.
    import torch as _torch
    _torch.autograd.set_multithreading_enabled(False)
