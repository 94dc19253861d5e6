"""Race amplifier: one full-size iteration with a stream held for a few ms at a tagged point (SR_DEBUG_DELAY, SR_DEBUG_TN_DELAY_MS)
against the same iteration without delays, bitwise.  A missing cross-stream dependency shows as a difference; correct ordering gives
identical bits whatever the timing.  python tools/race_amplifier.py   (runs every configuration in a child process)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import torch
import test_full_size_parity_gpu as T
from selfreconcode_amd import mlp_engine
mlp_engine.set_deferred_param_grads(True)
net, ds, conf = T._bench_scene('coarse')
fids = torch.tensor([3, 11, 40], device=T.DEV)
datas = ds.batch(fids)
rand = {k: v.to(T.DEV) for k, v in T._rand(700000).items()}
V0 = net.TmpVs.detach().clone()
outs = []
for rep in range(2):
    net.TmpVs = V0.clone().requires_grad_(True)
    net.TmpOptimizer = torch.optim.SGD([net.TmpVs], lr=0.05, momentum=0.9)
    for p in list(net.parameters()) + list(ds.learnable_weights()):
        p.grad = None
    dbg = {}
    loss = net(datas, 2048, T.RATIO, fids, rand=rand, debug=dbg)
    loss.backward(); net.propagateTmpPsGrad(fids, T.RATIO); torch.cuda.synchronize()
    out = T._collect(net, ds, loss)
    out['seeds'] = dbg['seeds'].clone(); out['initTmpPs'] = dbg['initTmpPs'].clone()
    outs.append({k: v.detach().cpu() for k, v in out.items()})
torch.save(outs, sys.argv[1])
''' % (ROOT, ROOT)
import torch
configs = [("baseline", {}), ("main held 6 ms after the fork", {"SR_DEBUG_DELAY": "main_after_fork:6"}), ("side held 8 ms after its wait", {"SR_DEBUG_DELAY": "side_after_wait:8"}),
           ("refiner held 10 ms", {"SR_DEBUG_DELAY": "refiner_start:10"}), ("selection stream held 10 ms after the index lists' round trip", {"SR_DEBUG_DELAY": "side_lists_made:10"}),
           ("sequential selection, vertex-subset stream held 10 ms", {"SR_DEBUG_DELAY": "aux_after_wait:10", "SR_FUSED_SELECTION": "0"}),
           ("main held 10 ms before it joins the side streams", {"SR_DEBUG_DELAY": "main_before_join:10"}),
           ("ray branch held 12 ms at its start (the main stream runs the sampled terms and the big backward meanwhile)", {"SR_DEBUG_DELAY": "ray_branch_start:12"}),
           ("implicit-gradient pass held 12 ms on the ray branch's stream", {"SR_DEBUG_DELAY": "ray_branch_propagate_start:12"}),
           ("main held 8 ms before it joins the ray branch", {"SR_DEBUG_DELAY": "main_before_ray_join:8"}),
           ("ray branch AND every weight-gradient launch held", {"SR_DEBUG_DELAY": "ray_branch_start:6", "SR_DEBUG_TN_DELAY_MS": "1"}), ("every weight-gradient launch held 1 ms", {"SR_DEBUG_TN_DELAY_MS": "1"})]
base = None
for mode in ("f32",):
    for name, env in configs:
        out = f"/tmp/race_{len(name)}.pt"
        r = subprocess.run([sys.executable, "-c", CHILD, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            print(name, "FAILED TO RUN", r.stderr[-800:]); continue
        a, b = torch.load(out)
        rep = [k for k in a if a[k].shape != b[k].shape or not torch.equal(a[k], b[k])]
        if base is None:
            base = a
        dif = [(k, float((a[k] - base[k]).abs().max())) for k in a if a[k].shape != base[k].shape or not torch.equal(a[k], base[k])]
        print(f"[{mode}] {name:52s} repeat-stable: {not rep}   differs from the baseline in {len(dif)} tensors {[(k, f'{d:.1e}') for k, d in dif[:6]]}", flush=True)
