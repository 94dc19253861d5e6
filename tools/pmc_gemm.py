"""One large layer GEMM (M = 262144, N = K = 512) through our NT / TN kernels and through the vendor library, for SQ counter passes:
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
            --kernel-trace --output-format csv -d out -- python tools/pmc_gemm.py"""
import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import torch
from selfreconcode_amd import mlp_engine as me
dev='cuda:0'
torch.backends.cuda.matmul.allow_tf32 = False
M=262144
A=torch.randn(M,512,device=dev); B=torch.randn(512,512,device=dev)*0.05; C=torch.empty(M,512,device=dev); b=torch.zeros(512,device=dev)
for _ in range(4):
    me._gemm_nt(A,512,B,512,C,512,M,512,512,b,1,me.ACT_NONE,me.EPI_FWD)
    torch.mm(A,B.t())
Z=torch.randn(M,512,device=dev)
for _ in range(3):
    me._gemm_tn(Z,512,A,512,M,512,512,512,1)
    torch.mm(Z.t(),A)
torch.cuda.synchronize()
