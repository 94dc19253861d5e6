"""Side-stream round trip while the main stream carries a BACKLOG from the previous iteration (tools/stream_latency.py starts every
case from an idle GPU).  Loop: main: record(fork), N large GEMMs; side: wait_event(fork), short chain, count -> host (polling);
no device synchronisation between iterations, so from the second iteration on main still holds work when fork is recorded."""
import torch, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from selfreconcode_amd import _lib
dev = torch.device('cuda:0')
A = torch.randn(8192, 2048, device=dev); B = torch.randn(2048, 2048, device=dev)
x = torch.randn(1 << 16, device=dev)
pin = torch.zeros(1, dtype=torch.int64).pin_memory()
def big(n):
    for _ in range(n):
        torch.mm(A, B)
def chain(n):
    y = x
    for _ in range(n):
        y = y * 1.0001
    return y
side = torch.cuda.Stream(device=dev, priority=-1)
flag = torch.zeros(2, dtype=torch.int32, device=dev)      # [0] the flag, [1] time-out counter
epoch = [0]
big(5); chain(5); torch.cuda.synchronize()
def iteration(nbig, mode, early):
    main = torch.cuda.current_stream(dev)
    fork = torch.cuda.Event()
    chain(2)
    if early == 'none':
        pass
    elif early == 'flag':
        epoch[0] += 1
        _lib.call('sr_stream_flag_set', flag.data_ptr(), epoch[0], main.cuda_stream)
    else:
        fork.record(main)
    if early is True:
        side.wait_event(fork)          # the dependency is taken while fork is still the LAST command of the main stream
    big(nbig)
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        if early is False:
            side.wait_event(fork)
        elif early == 'flag':
            _lib.call('sr_stream_flag_wait', flag.data_ptr(), epoch[0], flag.data_ptr() + 4, 2000, side.cuda_stream)
        y = chain(10)
        c = (y > 0).count_nonzero()
        if mode == 'poll':
            pin.copy_(c.view(1), non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
            while not ev.query():
                pass
        else:
            int(c)
    t1 = time.perf_counter()
    # tail of the iteration on main, issued AFTER the round trip (as the rest of the training step is)
    big(nbig // 2)
    return (t1 - t0) * 1e3
NAMES = {'none': 'NO wait on the side stream  ', True: 'right after record', False: 'after the main work ', 'flag': 'device flag, no event'}
import contextlib
MAIN = os.environ.get('MAIN_STREAM', 'null')
ctx = torch.cuda.stream(torch.cuda.Stream(device=dev)) if MAIN != 'null' else contextlib.nullcontext()
print('main stream:', MAIN)
ctx.__enter__()
for mode, early in (('poll', 'none'), ('poll', False), ('poll', 'flag')):
    for nbig in (20, 60):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        r = [iteration(nbig, mode, early) for _ in range(8)]
        e1.record(); torch.cuda.synchronize()
        print(f'{mode:5s} wait issued {NAMES[early]} nbig={nbig:3d}: round trip per iteration (ms): ' + ' '.join(f'{v:6.2f}' for v in r) + f'   | GPU time per iteration {e0.elapsed_time(e1) / 8:.2f} ms')

print('flag wait time-outs:', int(flag[1]))
