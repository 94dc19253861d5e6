"""The N > 1 path of bench.py end to end on ONE GPU: two ranks (gloo, both on device 0 through SR_ALL_RANKS_ON_DEVICE0) run the
real OptimNetwork step -- initial broadcast, frame sharding, the template-vertex all-reduce inside forward, the vertex-count check
after the remesh, the early (asynchronous) and the main gradient buffers.  A functional check of the collective path the
driver's multi-GPU bench takes over RCCL; scaling itself can only be measured on a multi-GPU node."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_run_the_real_step():
    env = dict(os.environ, SR_ALL_RANKS_ON_DEVICE0="1", SR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29577",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--settle", "3", "--settle-low", "1", "--noise-observations",
           "--no-cpu-baseline", "--no-gemm-events"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["value"] > 0 and rec["remesh"]["in_window"] == 1


def test_plain_bench_invocation_spawns_its_ranks():
    """`python bench.py --gpus 2` with NO launcher (the form the driver uses for --gpus 1): bench.py re-executes itself through
    torch.distributed.run; on this 1-GPU box both ranks share device 0 (functional run).  configs[2] shape: one frame per rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames-per-gpu", "1", "--steps", "2", "--warmup", "1", "--settle", "2", "--settle-low", "1",
           "--noise-observations", "--no-cpu-baseline", "--no-gemm-events"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["frames_per_gpu"] == 1 and rec["config"]["frames_per_step_all_ranks"] == 2
    assert abs(rec["value"] - rec["config"]["frames_per_s"] / 3.0) < 1e-3 * rec["value"]          # reference iterations = frames / 3 (coarse batch size)


def test_two_ranks_x_one_frame_equal_one_rank_x_two_frames(tmp_path):
    """SURVEY.md 8(e) caveats A/B on the REAL step: the template SGD step and every gradient after the all-reduce of two ranks with
    one frame each equal those of one rank with the two-frame batch (see tests/_dist_equiv_worker.py for what is compared and why)."""
    import numpy as np
    worker = os.path.join(ROOT, "tests", "_dist_equiv_worker.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    r = subprocess.run([sys.executable, worker, "--out", one], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env2 = dict(env, SR_ALL_RANKS_ON_DEVICE0="1", SR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29579",
                        worker, "--out", two, "--inject", one], env=env2, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = np.load(one), np.load(two)
    assert int(a["nrays"]) > 500 and int(a["nconv"]) > 100
    np.testing.assert_allclose(b["TmpVs"], a["TmpVs"], rtol=0, atol=2e-7)              # the shared template takes the SAME step (caveat A)
    bad = []
    for k in a.files:
        if not k.startswith("g_"):
            continue
        assert k in b.files, k
        err = np.abs(a[k] - b[k]).max() / max(np.abs(a[k]).max(), 1e-30)
        l2 = np.linalg.norm(a[k] - b[k]) / max(np.linalg.norm(a[k]), 1e-30)
        if err > 1e-4 or l2 > 1e-4:
            bad.append((k, float(err), float(l2)))
    assert not bad, bad


def test_rccl_backend_world_size_one():
    """backend="nccl" (RCCL) with the communicator bound to the device, at the only world size a one-GPU box allows: every helper of
    selfreconcode_amd/dist.py (initial broadcast, early / main gradient buffers, template all-reduce, pooled-mean weight, count check,
    describe) and two real steps with the collectives live (tests/_rccl_worker.py).  The other tests of this file force gloo."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("SR_DIST_BACKEND", "SR_ALL_RANKS_ON_DEVICE0")}
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SR_DIST_FORCE_INIT="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_worker.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["ok"] and rec["rccl"]["world"] == 1 and rec["rccl"]["backend"].startswith("rccl")
