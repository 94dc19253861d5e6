"""The split-bf16 (bf16x3) layer GEMM next to the exact-fp32 MFMA kernel on the layer shapes: TFLOP/s-equivalent (2 M N K / time)."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from selfreconcode_amd import mlp_engine as me
dev = 'cuda:0'


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n


for M in (262144, 86016, 8192):
    for (N, K, act, grp, tag) in ((512, 512, me.ACT_SOFTPLUS100, 1, 'softplus'), (512, 512, me.ACT_SOFTPLUS100, 4, 'softplus g4'), (512, 512, me.ACT_RELU, 1, 'relu'), (512, 167, me.ACT_RELU, 1, 'K=167'),
                                 (257, 512, me.ACT_NONE, 1, 'N=257')):
        A = torch.randn(M, me.pad4(K), device=dev); B = torch.randn(N, me.pad4(K), device=dev) * 0.05; C = torch.empty(M, me.pad4(N), device=dev); b = torch.zeros(N, device=dev)
        planes = me.split_bf16x3(B, K)
        res = {}
        for mode in ("f32", "bf16x3"):
            me.GEMM_MODE = mode
            if mode == "bf16x3":
                me._PLANES_BY_PTR[B.data_ptr()] = planes
            ms = timeit(lambda: me._gemm_nt(A, A.stride(0), B, B.stride(0), C, C.stride(0), M, N, K, b, grp, act, me.EPI_FWD))
            me._PLANES_BY_PTR.pop(B.data_ptr(), None)
            res[mode] = ms
        me.GEMM_MODE = "f32"
        print(f"NT M={M:7d} N={N} K={K} {tag:12s}: fp32 {res['f32'] * 1e3:8.1f} us {2 * M * N * K / res['f32'] / 1e9:7.1f} TF | bf16x3 {res['bf16x3'] * 1e3:8.1f} us {2 * M * N * K / res['bf16x3'] / 1e9:7.1f} TF-eq  x{res['f32'] / res['bf16x3']:.2f}", flush=True)
