"""MI355X-native hot path of SelfRecon."""
import os as _os

if _os.environ.get("SR_AUTOGRAD_CALLER_THREAD", "1") != "0":
    # The iteration runs nine autograd traversals, each a few hundred nodes whose backward is a handful of C-ABI launches.  With the
    # engine's default threading every traversal is handed to the device's worker thread and the caller sleeps on a condition variable
    # until it is done: two thread wake-ups per traversal and a GIL hand-over per Python node.  On the calling thread the same nodes
    # run in the same order on the same streams (the engine's stream guards do not depend on the thread), without the hand-overs.
    # Thread-local to the importing thread; SR_AUTOGRAD_CALLER_THREAD=0 leaves torch's default.
    import torch as _torch
    _torch.autograd.set_multithreading_enabled(False)
