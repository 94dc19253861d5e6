"""Where does the split-bf16 mode's rare first-iteration difference come from?  (VERDICT r03 "next" item 6.)
One process: builds the fine-stage bench scene, runs the timed schedule twice and reports WHICH intermediate differs first
(seeds -> refined points -> loss terms -> gradients); then repeats the iteration after each of several "make it a first
iteration again" resets (allocator cache emptied, weight packs dropped, refiner workspaces dropped, gc) and counts the
differences per reset kind.   SR_GEMM=bf16x3 python tools/bf16x3_hunt.py [trials]"""
import gc
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import test_full_size_parity_gpu as T
from selfreconcode_amd import mlp_engine
from selfreconcode_amd.utils import FindSurfacePs as FSP

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 20
stage = os.environ.get("SR_HUNT_STAGE", "fine")
mlp_engine.set_deferred_param_grads(True)
net, ds, conf = T._bench_scene(stage)
DEV = T.DEV
fids = torch.tensor([3, 11, 40] if stage == "coarse" else [17], device=DEV)
datas = ds.batch(fids)
rand = {k: v.to(DEV) for k, v in T._rand(700000).items()}
V0 = net.TmpVs.detach().clone()


def run():
    net.TmpVs = V0.clone().requires_grad_(True)
    net.TmpOptimizer = torch.optim.SGD([net.TmpVs], lr=0.05, momentum=0.9)
    for p in list(net.parameters()) + list(ds.learnable_weights()):
        p.grad = None
    net.refiner_stream = "side"
    net._camera_cache = None
    dbg = {}
    loss = net(datas, 2048, T.RATIO, fids, rand=rand, debug=dbg)
    loss.backward()
    net.propagateTmpPsGrad(fids, T.RATIO)
    torch.cuda.synchronize()
    out = T._collect(net, ds, loss)
    for k in ("defTmpVs", "proj_xy", "proj_z", "pix_to_face", "batch_inds", "row_inds", "col_inds", "seeds", "initTmpPs", "check", "rays"):
        out["dbg_" + k] = dbg[k].clone()
    return out


ORDER = ["dbg_defTmpVs", "dbg_proj_xy", "dbg_proj_z", "dbg_pix_to_face", "dbg_batch_inds", "dbg_row_inds", "dbg_col_inds", "dbg_seeds", "dbg_rays", "dbg_initTmpPs", "dbg_check", "TmpPs", "L_pc_loss_sdf", "TmpVs"]


def diff(a, b):
    keys = ORDER + [k for k in a if k not in ORDER]
    d = [k for k in keys if a[k].shape != b[k].shape or not torch.equal(a[k], b[k])]
    return d, [float((a[k].float() - b[k].float()).abs().max()) if a[k].shape == b[k].shape else -1.0 for k in d[:6]]


def drop_packs():
    for key in list(mlp_engine._PACK_CACHE):
        mlp_engine._drop_entry(key)
    for m in (net.sdf, net.deformer.defs[0], net.netRender):
        m.__dict__.pop('_sr_packs', None)


a1 = run()
a2 = run()
d, mx = diff(a1, a2)
print("FIRST-vs-SECOND", "equal" if not d else ("DIFF " + str(d[:6]) + " " + str(mx)), flush=True)
if d:
    a3 = run()
    print("   second-vs-third", "equal" if not diff(a2, a3)[0] else "DIFF", flush=True)
ref = run()
RESETS = {
    "none": lambda: None,
    "empty_cache": torch.cuda.empty_cache,
    "drop_packs": drop_packs,
    "drop_refiner_ws": FSP._WORKSPACES.clear,
    "gc": gc.collect,
    "all": lambda: (drop_packs(), FSP._WORKSPACES.clear(), gc.collect(), torch.cuda.empty_cache()),
}
want = os.environ.get("SR_HUNT_RESETS")
for name, fn in RESETS.items():
    if want and name not in want.split(","):
        continue
    bad, first = 0, None
    for i in range(trials):
        torch.cuda.synchronize()
        fn()
        r = run()
        d, mx = diff(ref, r)
        if d:
            bad += 1
            first = first or (d[:4], mx[:4])
            if "dbg_pix_to_face" in d:
                pa, pb = ref["dbg_pix_to_face"].view(-1), r["dbg_pix_to_face"].view(-1)
                ne = pa != pb
                print("   pix_to_face: %d pixels differ; reference hit / trial empty %d, reference empty / trial hit %d, both hit %d; first pixels %s"
                      % (int(ne.sum()), int((ne & (pa >= 0) & (pb < 0)).sum()), int((ne & (pa < 0) & (pb >= 0)).sum()), int((ne & (pa >= 0) & (pb >= 0)).sum()),
                         [(int(i), int(pa[i]), int(pb[i])) for i in ne.nonzero().view(-1)[:4]]), flush=True)
    print(f"reset={name}: {bad} of {trials} differ from the reference run", "" if first is None else first, flush=True)
