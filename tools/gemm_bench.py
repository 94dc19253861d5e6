"""Layer-GEMM micro-benchmarks: the NT kernel (forward / backward-data, fused epilogues) and the TN weight-gradient kernel on
the shapes of the three MLPs, next to the vendor library (hipBLASLt / rocBLAS through torch.mm) on the same fp32 shapes."""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch, ctypes
from selfreconcode_amd import mlp_engine as me, _lib
dev='cuda:0'
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
for M in (262144, 49152, 24576, 6144, 512):
    for (N,K,act,grp,tag) in ((512,512,me.ACT_NONE,1,'none'),(512,512,me.ACT_SOFTPLUS100,1,'softplus'),(512,512,me.ACT_SOFTPLUS100,4,'softplus g4'),(512,512,me.ACT_RELU,1,'relu'),(512,40,me.ACT_RELU,1,'K=39'),(257,512,me.ACT_NONE,1,'N=257')):
        A=torch.randn(M,me.pad4(K),device=dev); B=torch.randn(N,me.pad4(K),device=dev)*0.05; C=torch.empty(M,me.pad4(N),device=dev); b=torch.zeros(N,device=dev)
        Kr = 39 if K==40 else K
        ms=timeit(lambda: me._gemm_nt(A,A.stride(0),B,B.stride(0),C,C.stride(0),M,N,Kr,b,grp,act,me.EPI_FWD))
        print(f"NT M={M:7d} N={N} K={Kr} {tag:12s}: {ms*1e3:8.1f} us  {2*M*N*Kr/ms/1e9:7.1f} TF/s", flush=True)
    Z=torch.randn(M,512,device=dev); X=torch.randn(M,512,device=dev)
    ms=timeit(lambda: me._gemm_tn(Z,512,X,512,M,512,512,512,1))
    print(f"TN R={M:7d} 512x512: {ms*1e3:8.1f} us  {2*M*512*512/ms/1e9:7.1f} TF/s", flush=True)
    X0=torch.randn(M,40,device=dev)                 # first-layer weight gradient: dW [512, 39] (256 x 64 tiles)
    ms=timeit(lambda: me._gemm_tn(Z,512,X0,40,M,512,39,40,1))
    print(f"TN R={M:7d} 512x39 : {ms*1e3:8.1f} us  {2*M*512*39/ms/1e9:7.1f} TF/s  {(M*552*4)/ms/1e6:7.1f} GB/s of operand rows", flush=True)
# vendor library (hipBLASLt / rocBLAS through torch) on the same fp32 shapes: the practical fp32 MFMA ceiling on this part
torch.backends.cuda.matmul.allow_tf32 = False
for M in (262144, 49152, 24576, 6144):
    A=torch.randn(M,512,device=dev); B=torch.randn(512,512,device=dev)
    ms=timeit(lambda: torch.mm(A,B.t()))
    print(f"torch.mm NT M={M:7d} 512x512: {ms*1e3:8.1f} us {2*M*512*512/ms/1e9:7.1f} TF/s", flush=True)
    Z=torch.randn(M,512,device=dev)
    ms=timeit(lambda: torch.mm(Z.t(),A))
    print(f"torch.mm TN R={M:7d} 512x512: {ms*1e3:8.1f} us {2*M*512*512/ms/1e9:7.1f} TF/s", flush=True)
