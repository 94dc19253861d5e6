"""Where the HOST spends an iteration: python tools/host_profile.py [steps]
Builds bench.py's coarse-stage scene, warms up, then (1) times the enqueue of each step with no synchronisation other than the
path's own (host time per step against the GPU's), (2) runs the same steps under cProfile and prints the functions by own time."""
import cProfile, pstats, sys, time, io, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfreconcode_amd import mlp_engine
from selfreconcode_amd.synthetic import build_synthetic_scene
from selfreconcode_amd.optim import FusedAdam

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
FR = int(os.environ.get('SR_HP_FRAMES', '3'))                 # frames per step (1 + SR_HP_SIM_WORLD=8: the workload of one rank of configs[2])
if int(os.environ.get('SR_HP_SIM_WORLD', '0')) > 1:
    from selfreconcode_amd import dist as _srdist
    _srdist.simulate_world((0, int(os.environ['SR_HP_SIM_WORLD'])))
dev = torch.device('cuda:0')
if os.environ.get('SR_HP_BIND', '0') != '0':                  # the placement bench.py uses (selfreconcode_amd/affinity.py); SR_HP_SLOT = which core group
    from selfreconcode_amd import affinity
    print('host threads:', affinity.bind(0, slot=int(os.environ['SR_HP_SLOT']) if 'SR_HP_SLOT' in os.environ else None))
net, ds, conf = build_synthetic_scene(device=dev, frame_num=64, stage=os.environ.get('SR_HP_STAGE', 'coarse'), consistent_masks=False)
net.masked_ray_branch_below = int(os.environ.get('SR_HP_MASK', '0'))        # OptimNetwork.masked_ray_branch_below (bench.py's default: 4096)
params = [p for p in net.parameters() if p.requires_grad]
mlp_engine.set_deferred_param_grads(True)
opt = FusedAdam([{'params': ds.learnable_weights()}, {'params': params}], lr=float(os.environ.get('SR_HP_LR', '1e-4')))
ratio = {'sdfRatio': 1., 'deformerRatio': 0.6, 'renderRatio': 1.}
state = {'it': 0}
marks = []


def step(mark=False):
    it = state['it']
    f = torch.arange(FR * it % 56, FR * it % 56 + FR, device=dev)
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    loss = net(ds.batch(f), int(os.environ.get('SR_HP_RAYS', '2048')), ratio, f)
    t1 = time.perf_counter()
    loss.backward()
    net._mark('backward issued')
    t2 = time.perf_counter()
    net.propagateTmpPsGrad(f, ratio)
    net._mark('implicit-gradient pass issued')
    t3 = time.perf_counter()
    opt.step()
    net._mark('adam issued')
    t4 = time.perf_counter()
    if mark:
        marks.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    state['it'] = it + 1


ds.attach_rendered_observations(net, ratio)          # as bench.py: no host work in ds.batch
net.forward_time = 1
for _ in range(10):
    step()
# host clock against the GPU's at the marks of the iteration (OptimNetwork._mark): where the host leads and where the GPU waits for it
from selfreconcode_amd import hostsync, _lib
net.mark_stamps = torch.zeros(4096, dtype=torch.int64, device=dev)
# calibration of the device counter (100 MHz) against the host clock: one stamp on an idle stream
torch.cuda.synchronize(); _lib.call('sr_stream_stamp', net.mark_stamps.data_ptr() + 8 * 4095, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
cal_host, cal_dev = time.perf_counter(), int(net.mark_stamps[4095])
net.host_marks = []
hostsync.TRACE = net._mark
for _ in range(6):
    step()
torch.cuda.synchronize()
M, net.host_marks = net.host_marks, None
hostsync.TRACE = None
stamps = net.mark_stamps.cpu().tolist()
dev_ms = lambda i: (stamps[i] - cal_dev) / 1e5 + cal_host * 1e3        # device stamp on the host's clock (ms)
starts = [i for i, m in enumerate(M) if m[0] == 'start']
print(f'{"mark":34s} {"host ms":>8s} {"gpu ms":>8s}   since the HOST reached the start mark of the iteration (last of {len(starts)} iterations shown per row: mean); gpu = when the stream the mark was issued on got there')
rows = {}
for a_, b_ in zip(starts[:-1], starts[1:]):
    seen = {}
    for lab, t, i in M[a_:b_ + 1]:
        key = lab if (lab != 'start' or i == M[a_][2]) else 'next start'
        if key.startswith('count'):
            seen[key] = seen.get(key, 0) + 1
            key = f'  {key} #{seen[key]}'
        rows.setdefault(key, []).append(((t - M[a_][1]) * 1e3, dev_ms(i) - M[a_][1] * 1e3))
for lab, v in rows.items():
    print(f'{lab:34s} {sum(x[0] for x in v) / len(v):8.2f} {sum(x[1] for x in v) / len(v):8.2f}')
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(True)
th = time.perf_counter() - t0
torch.cuda.synchronize()
tw = time.perf_counter() - t0
print(f'host enqueue {th / steps * 1e3:.2f} ms / step, wall with final sync {tw / steps * 1e3:.2f} ms / step')
m = [sum(x[i] for x in marks) / len(marks) * 1e3 for i in range(4)]
print('host ms per step: forward %.2f  backward %.2f  propagate %.2f  adam %.2f' % tuple(m))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumulative'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    txt = s.getvalue()
    print(txt[txt.find('ncalls') - 4:])

# --- native stacks of this thread against the marks of the iteration (tools/_src/stack_sampler.c; build: see tools/sample_stacks.sh)
so = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_bin', 'libstack_sampler.so')
if os.path.exists(so) and os.environ.get('SR_STACKS', '1') != '0':
    import ctypes, collections, bisect
    L = ctypes.CDLL(so)
    net.host_marks = []
    net.mark_stamps = torch.zeros(65536, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    L.sampler_start(200, 200000)
    for _ in range(40):
        step()
    n = L.sampler_stop()
    torch.cuda.synchronize()
    out = os.environ.get('SR_STACKS_OUT', '/tmp/host_stacks.txt')
    L.sampler_dump(out.encode())
    M, net.host_marks = net.host_marks, None
    times = [m[1] for m in M]; labels = [m[0] for m in M]
    per = collections.defaultdict(collections.Counter)
    cur = None
    def flush(t, frames):
        i = bisect.bisect_right(times, t) - 1
        phase = labels[i] if i >= 0 else 'before'
        named = [f for f in frames if not f.endswith('!?') and not f.startswith('python!') and not f.startswith('?!')]
        per[phase][' < '.join(named[:7])] += 1
    frames = []; t = None
    for line in open(out):
        line = line.rstrip('\n')
        if line.startswith('--- '):
            if t is not None: flush(t, frames)
            t = float(line[4:]); frames = []
        else:
            frames.append(line)
    if t is not None: flush(t, frames)
    print(f'\n{n} native stack samples of the host thread over 40 iterations, by the mark that PRECEDES the sample (top signatures):')
    for phase in dict.fromkeys(labels):
        c = per.get(phase)
        if not c: continue
        tot = sum(c.values())
        print(f'== after "{phase}": {tot} samples')
        for sig, k in c.most_common(4):
            print(f'   {k:5d}  {sig[:420]}')
