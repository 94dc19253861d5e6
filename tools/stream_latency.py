"""How long does a side stream wait behind a busy main stream?  python tools/stream_latency.py
main: [short kernel] -> record(fork) -> N large GEMMs (~10 ms).   side: wait_event(fork) -> tiny kernel chain -> record(done).
Cases: side work issued AFTER the large kernels (the order of the iteration) or BEFORE them; high / default priority; fork event
with / without timing.  Reports when the side chain started and ended relative to fork, and when main finished."""
import torch, time
dev = torch.device('cuda:0')
A = torch.randn(8192, 4096, device=dev); B = torch.randn(4096, 4096, device=dev)
x = torch.randn(1 << 16, device=dev)
def big(n):
    for _ in range(n):
        torch.mm(A, B)
def chain(n):
    y = x
    for _ in range(n):
        y = y * 1.0001
    return y
lo, hi = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=-1)
for _ in range(3):
    big(5); chain(5)
torch.cuda.synchronize()
def run(side, before, timing_fork, nbig=40, nchain=20, host_sync=False):
    main = torch.cuda.current_stream(dev)
    t_fork = torch.cuda.Event(enable_timing=True)
    fork = torch.cuda.Event(enable_timing=True) if timing_fork else torch.cuda.Event()
    s0, s1, m1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    chain(2)
    t_fork.record(main); fork.record(main)
    def side_work():
        with torch.cuda.stream(side):
            side.wait_event(fork)
            s0.record(side)
            y = chain(nchain)
            if host_sync:
                y[:8].cpu()
                chain(nchain)
            s1.record(side)
    if before:
        side_work(); big(nbig)
    else:
        big(nbig); side_work()
    m1.record(main)
    torch.cuda.synchronize()
    return t_fork.elapsed_time(s0), t_fork.elapsed_time(s1), t_fork.elapsed_time(m1)
print('case                                              side start  side end  main end (ms after fork)')
for name, side in (('high priority', hi), ('default priority', lo)):
    for before in (False, True):
        for tf in (False, True):
            for hs in (False, True):
                r = [run(side, before, tf, host_sync=hs) for _ in range(3)][-1]
                print(f'{name:17s} issued {"before" if before else "after ":6s} main work, fork timing={tf!s:5s} host sync in chain={hs!s:5s}  {r[0]:8.3f} {r[1]:8.3f} {r[2]:8.3f}')
