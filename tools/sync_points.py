"""Which calls of an iteration make the HOST wait for the GPU?  python tools/sync_points.py [frames]
torch.cuda.set_sync_debug_mode("warn") reports every synchronising call torch itself makes (nonzero / item / tolist, copies from or to
pageable host memory -- `torch.tensor(values, device=cuda)` is one, and waits for the whole current stream --, masked indexing ...);
each is attributed to the innermost frame inside this package.  Expected per iteration: the two count round trips (hostsync.py) and,
on a remesh iteration, marching cubes' count copy."""
import sys, os, warnings, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfreconcode_amd import mlp_engine
from selfreconcode_amd.synthetic import build_synthetic_scene
from selfreconcode_amd.optim import FusedAdam

FR = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda:0')
net, ds, conf = build_synthetic_scene(device=dev, frame_num=64, stage='coarse', consistent_masks=False)
params = [p for p in net.parameters() if p.requires_grad]
mlp_engine.set_deferred_param_grads(True)
opt = FusedAdam([{'params': ds.learnable_weights()}, {'params': params}], lr=1e-4)
ds.attach_rendered_observations(net, {'sdfRatio': 1., 'deformerRatio': 0.6, 'renderRatio': 1.})
state = {'it': 0}


def step():
    it = state['it']
    ratio = {'sdfRatio': 1., 'deformerRatio': min(1.0, it / 2500. + 0.5), 'renderRatio': 1.}
    f = torch.arange(FR * it % 56, FR * it % 56 + FR, device=dev)
    opt.zero_grad(set_to_none=True)
    loss = net(ds.batch(f), 2048, ratio, f)
    loss.backward()
    net.propagateTmpPsGrad(f, ratio)
    opt.step()
    state['it'] = it + 1


net.forward_time = 1
for _ in range(5):
    step()
torch.cuda.synchronize()
sites = collections.Counter()
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def showwarning(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message).lower():
        return
    stack = [fr for fr in traceback.extract_stack() if fr.filename.startswith(here) and "sync_points.py" not in fr.filename]
    fr = stack[-1] if stack else None
    sites[(os.path.relpath(fr.filename, here), fr.lineno, fr.name, fr.line) if fr else ("?", 0, "?", "?")] += 1


warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
N = 4
for _ in range(N):
    step()
torch.cuda.set_sync_debug_mode("default")
print(f"synchronising calls per iteration ({FR} frame(s) per step, {N} iterations, no remesh among them):")
for (fn, ln, name, line), c in sites.most_common():
    print(f"  {c / N:5.2f}  {fn}:{ln} {name}   {line}")
