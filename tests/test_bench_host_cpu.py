"""Host logic of bench.py and of the one-launch optimizer that needs no GPU: the frames-per-rank rule of the two scaling modes, the
driver's contract on the command line (defaults that finish within minutes, the flags the driver passes), and FusedAdam's refusals
(it has no CPU path: it must say so, not fall back)."""
import argparse
import importlib
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(**kw):
    base = dict(scaling="weak", frames_per_gpu=0, global_frames=8)
    base.update(kw)
    return argparse.Namespace(**base)


def test_frames_per_rank_rules():
    bench = importlib.import_module("bench")
    # weak scaling: the stage's batch per rank, whatever the world size (configs[1]: 3 frames; the fine stage: 1)
    for world in (1, 2, 8):
        assert bench.frames_per_rank("coarse", _args(), world) == 3
        assert bench.frames_per_rank("fine", _args(), world) == 1
    # --frames-per-gpu 1 at N = 8 is configs[2]
    assert bench.frames_per_rank("coarse", _args(frames_per_gpu=1), 8) == 1
    # strong scaling: a fixed global batch split over the ranks, and a refusal when it does not divide
    assert [bench.frames_per_rank("coarse", _args(scaling="strong"), w) for w in (1, 2, 4, 8)] == [8, 4, 2, 1]
    with pytest.raises(SystemExit):
        bench.frames_per_rank("coarse", _args(scaling="strong"), 3)
    assert set(bench.STAGES) == {"coarse", "fine"} and bench.STAGES["coarse"] == dict(frames=3, rays=2048) and bench.STAGES["fine"] == dict(frames=1, rays=6144)


def test_bench_command_line_contract():
    """`python bench.py --help` works without a GPU and names the flags the driver uses; the defaults are one GPU and a step count that
    finishes within minutes."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    for flag in ("--gpus", "--steps", "--warmup", "--scaling", "--frames-per-gpu", "--global-frames", "--no-cpu-baseline", "--simulate-world"):
        assert flag in r.stdout, flag
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'add_argument("--gpus", type=int, default=1)' in src
    assert 'add_argument("--steps", type=int, default=30)' in src and 'add_argument("--warmup", type=int, default=5)' in src


def test_fused_adam_refuses_what_it_does_not_implement():
    from selfreconcode_amd.optim import FusedAdam
    p = torch.nn.Parameter(torch.zeros(4))
    with pytest.raises(NotImplementedError):
        FusedAdam([p], lr=1e-3, weight_decay=0.1)
    with pytest.raises(NotImplementedError):
        FusedAdam([p], lr=1e-3, amsgrad=True)
    opt = FusedAdam([p], lr=1e-3)
    assert opt.step() is None                                  # no gradient anywhere: nothing to do, no launch
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="GPU"):             # a CPU parameter: no silent fallback to a host loop
        opt.step()
    # the state layout is torch.optim.Adam's: its state_dict loads
    ref = torch.optim.Adam([torch.nn.Parameter(torch.zeros(4))], lr=2e-3)
    opt2 = FusedAdam([torch.nn.Parameter(torch.zeros(4))], lr=1e-3)
    opt2.load_state_dict(ref.state_dict())
    assert opt2.param_groups[0]["lr"] == 2e-3
