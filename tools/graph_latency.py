"""Is a HIP graph of dependent small launches faster than the same launches issued eagerly?  (the refiner is ~230 dependent launches of
~13 us each: DESIGN.md 3.4)   python tools/graph_latency.py
200 dependent launches of the layer-pair kernel on 2048 live rows (the refiner's shape at one frame per rank), eager against a captured
torch.cuda.CUDAGraph replay; GPU time from events around the whole sequence, host time from perf_counter."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfreconcode_amd import _lib, mlp_engine as me

dev = torch.device("cuda:0")
M = 2048
A = torch.randn(M, 512, device=dev); B = torch.randn(512, 512, device=dev) * 0.04; bias = torch.zeros(512, device=dev)
C = [torch.empty(M, 512, device=dev) for _ in range(2)]


def chain(n):
    x = A
    for i in range(n):
        me._gemm_nt(x, 512, B, 512, C[i & 1], 512, M, 512, 512, bias, 1, me.ACT_RELU, me.EPI_FWD)
        x = C[i & 1]


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, host / reps * 1e3


N = 200
chain(N); torch.cuda.synchronize()
g_ms, h_ms = timeit(lambda: chain(N))
print(f"eager : {N} dependent launches: GPU {g_ms * 1e3 / N:.2f} us / launch, host issue {h_ms * 1e3 / N:.2f} us / launch")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    chain(4)
torch.cuda.current_stream().wait_stream(s)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    chain(N)
torch.cuda.synchronize()
g_ms, h_ms = timeit(graph.replay)
print(f"graph : {N} dependent launches: GPU {g_ms * 1e3 / N:.2f} us / launch, host issue {h_ms * 1e3 / N:.2f} us / launch")
for M2 in (256, 1024):
    M = M2
    g_ms, h_ms = timeit(lambda: chain(N))
    print(f"eager, {M2} rows: GPU {g_ms * 1e3 / N:.2f} us / launch")
