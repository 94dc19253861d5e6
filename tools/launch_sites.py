"""Which source lines launch the torch kernels of an iteration?  python tools/launch_sites.py [iterations]
torch.profiler with Python stacks over a few iterations of bench.py's coarse-stage step; every operator that launched at least one
kernel is attributed to the innermost frame inside this package.  Prints launches per iteration by (site, op)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from selfreconcode_amd import mlp_engine
from selfreconcode_amd.synthetic import build_synthetic_scene
from selfreconcode_amd.optim import FusedAdam

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device('cuda:0')
net, ds, conf = build_synthetic_scene(device=dev, frame_num=64, stage='coarse', consistent_masks=False)
params = [p for p in net.parameters() if p.requires_grad]
mlp_engine.set_deferred_param_grads(True)
opt = FusedAdam([{'params': ds.learnable_weights()}, {'params': params}], lr=3.7e-6)
ratio = {'sdfRatio': 1., 'deformerRatio': 0.6, 'renderRatio': 1.}
ds.attach_rendered_observations(net, ratio)
state = {'it': 0}
def step():
    it = state['it']
    f = torch.arange(3 * it % 60, 3 * it % 60 + 3, device=dev)
    opt.zero_grad(set_to_none=True)
    loss = net(ds.batch(f), 2048, ratio, f)
    loss.backward()
    net.propagateTmpPsGrad(f, ratio)
    opt.step()
    state['it'] = it + 1
net.forward_time = 1
for _ in range(8):
    step()
torch.cuda.synchronize()
# --- part 1: aten operator calls by Python call site (TorchDispatchMode sees every dispatched operator on this thread and on the
#     autograd engine's thread, which inherits the mode; views / metadata operators are skipped)
import traceback, threading
from torch.utils._python_dispatch import TorchDispatchMode
VIEWS = ('view', 'reshape', 'expand', 'slice', 'select', 'as_strided', 'transpose', 'permute', 'detach', 'alias', 'unsqueeze', 'squeeze',
         '_unsafe_view', 'unbind', 'split', 't.', 'size', 'stride', 'is_', 'sym_', '_local_scalar', 'empty', 'record_stream', 'lift_fresh',
         'result_type', 'storage_offset', 'numel', 'dim', 'set_', 'resize', 'chunk', 'narrow', 'unfold', 'diagonal', 'real', 'conj', 'pin_memory', 'is_pinned', '_to_copy_cpu')
calls = collections.Counter()
class Sites(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace('aten.', '')
        if not any(name.startswith(v) for v in VIEWS):
            node = torch._C._current_autograd_node() if hasattr(torch._C, '_current_autograd_node') else None
            site = 'engine node: ' + (node.name() if node is not None else '?')
            for fr in reversed(traceback.extract_stack(limit=40)):
                if 'selfreconcode_amd' in fr.filename and 'launch_sites' not in fr.filename:
                    site = fr.filename.split('selfreconcode_amd/')[-1] + ':' + str(fr.lineno) + ' ' + fr.name
                    break
            calls[(site, name)] += 1
        return func(*args, **(kwargs or {}))
with Sites():
    for _ in range(iters):
        step()
torch.cuda.synchronize()
tot = sum(calls.values())
print(f'{tot / iters:.0f} non-view aten operator calls per iteration (an operator is one or more kernel launches); by call site:')
bysite = collections.Counter()
for (site, name), n in calls.items():
    bysite[site] += n
for site, n in bysite.most_common(70):
    ops = collections.Counter({name: k for (s_, name), k in calls.items() if s_ == site})
    print(f'{n / iters:7.1f}  {site[:100]:100s} ' + ', '.join(f'{k}x{v / iters:.0f}' for k, v in ops.most_common(6)))
print()
torch.cuda.synchronize()
sys.exit(0)
