"""One layer-GEMM shape through one kernel family a few times, for counter passes:  python tools/pmc_gemm_one.py f32 [M] [N] [K]"""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from selfreconcode_amd import mlp_engine as me
dev = 'cuda:0'
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"
M, N, K = (int(x) for x in (sys.argv[2:5] + [262144, 512, 512][len(sys.argv[2:5]):]))
A = torch.randn(M, me.pad4(K), device=dev); B = torch.randn(N, me.pad4(K), device=dev) * 0.05; C = torch.empty(M, me.pad4(N), device=dev); b = torch.zeros(N, device=dev)
for _ in range(6):
    me._gemm_nt(A, A.stride(0), B, B.stride(0), C, C.stride(0), M, N, K, b, 1, me.ACT_SOFTPLUS100, me.EPI_FWD)
torch.cuda.synchronize()
