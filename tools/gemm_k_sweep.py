"""fp32 NT layer GEMM rate against the contraction length K (M = 131072, N = 512): what a tile's prologue + epilogue cost at K = 512."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from selfreconcode_amd import mlp_engine as me
dev = 'cuda:0'


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n


M, N = 131072, 512
for act, tag in ((me.ACT_RELU, 'relu'), (me.ACT_SOFTPLUS100, 'softplus')):
    for K in (128, 256, 512, 1024, 2048, 4096):
        A = torch.randn(M, K, device=dev) * 0.1; B = torch.randn(N, K, device=dev) * 0.05; C = torch.empty(M, N, device=dev); b = torch.zeros(N, device=dev)
        for _ in range(2):
            ms = timeit(lambda: me._gemm_nt(A, A.stride(0), B, B.stride(0), C, C.stride(0), M, N, K, b, 1, act, me.EPI_FWD))
        print(f"NT M={M} N={N} K={K:5d} {tag:9s}: {ms * 1e3:8.1f} us  {2 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
