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
