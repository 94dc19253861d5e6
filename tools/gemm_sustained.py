"""One layer-GEMM shape in a loop for a few seconds (default: the 128x128-tile NT kernel, M = 262144, 512 x 512, Softplus epilogue), so
that the shader clock under THIS kernel's power draw can be read next to its rate:  tools/with_clocks.sh OUT python tools/gemm_sustained.py [nt|tn] [seconds] [M]"""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import time
import torch
from selfreconcode_amd import mlp_engine as me
dev = 'cuda:0'
kind = sys.argv[1] if len(sys.argv) > 1 else "nt"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
M, N, K = (int(sys.argv[3]) if len(sys.argv) > 3 else 262144), 512, 512
A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) * 0.05; C = torch.empty(M, N, device=dev); b = torch.zeros(N, device=dev)
Z = torch.randn(M, N, device=dev)
fn = (lambda: me._gemm_nt(A, K, B, K, C, N, M, N, K, b, 1, me.ACT_SOFTPLUS100, me.EPI_FWD)) if kind == "nt" else (lambda: me._gemm_tn(Z, N, A, K, M, N, K, K, 1))
for _ in range(5):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.perf_counter() - t0 < secs:
    for _ in range(50):
        fn()
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(f"{kind.upper()} M={M} {N}x{K}: {ms * 1e3:.1f} us per launch, {2 * M * N * K / ms / 1e9:.1f} TFLOP/s sustained over {n} launches ({secs:.0f} s)")
