"""Does a layer GEMM write outside its output?  The output matrix is a window inside a larger buffer filled with a sentinel; after the
launch every float outside the window's [M, pad4(N + nfill)] footprint (rows before / after, the pitch gap of every row) must still
hold the sentinel, and the operands must be unchanged.  Both arithmetic modes, the shapes of the sdf / deformer / render forward
and backward-data passes at template-sized row counts (ragged M).   python tools/gemm_guard.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from selfreconcode_amd import mlp_engine as me

DEV = "cuda:0"
SENT = 0x7FC12345 - (1 << 32) if 0x7FC12345 >= (1 << 31) else 0x7FC12345
torch.manual_seed(0)
bad = 0
for (M, N, K, act, group, mode, nfill) in ((196357, 512, 512, me.ACT_SOFTPLUS100, 1, me.EPI_FWD, 0), (196357, 512, 39, me.ACT_SOFTPLUS100, 1, me.EPI_FWD, 0),
                                          (196357, 473, 512, me.ACT_SOFTPLUS100, 1, me.EPI_FWD, 39), (196357, 257, 512, me.ACT_NONE, 1, me.EPI_FWD, 0),
                                          (87041, 512, 167, me.ACT_RELU, 1, me.EPI_FWD, 0), (196357, 512, 473, me.ACT_SOFTPLUS100, 1, me.EPI_BWD, 0),
                                          (196352, 512, 512, me.ACT_NONE, 1, me.EPI_FWD, 0), (4 * 49089, 512, 512, me.ACT_SOFTPLUS100, 4, me.EPI_FWD, 0),
                                          (2 * 98179, 512, 512, me.ACT_SOFTPLUS100, 2, me.EPI_BWD, 0), (196357, 128, 512, me.ACT_RELU, 1, me.EPI_FWD, 0),
                                          (196357, 512, 128, me.ACT_RELU, 1, me.EPI_BWD, 0), (196357, 256, 283, me.ACT_RELU, 1, me.EPI_FWD, 0)):
    A = (torch.randn(M, me.pad4(K), device=DEV) * 0.3).contiguous()
    B = (torch.randn(N, me.pad4(K), device=DEV) * 0.05).contiguous()
    A[:, K:] = 0; B[:, K:] = 0
    bias = torch.randn(N, device=DEV) * 0.01 if mode == me.EPI_FWD else None
    aux, kw = None, dict(out_scale=0.7)
    if mode == me.EPI_BWD:
        aux = (torch.rand(M, me.pad4(N), device=DEV) * 0.05).contiguous()
        kw = dict(out_scale=1.0, aux=aux, ldaux=aux.stride(0), nact_bwd=N, aux_scale=1.0)
    elif nfill:
        aux = torch.randn(M, me.pad4(nfill), device=DEV).contiguous()
        kw = dict(out_scale=0.7, aux=aux, ldaux=aux.stride(0), naux_fwd=nfill)
    planes = me.split_bf16x3(B, K)
    keep = [t.clone() for t in (A, B, planes[0]) + ((aux,) if aux is not None else ())]
    wcols = me.pad4(N + nfill)
    ldc = wcols + 64
    pre = 300
    for tag in ("f32", "bf16x3"):
        G = torch.full(((M + 2 * pre) * ldc,), SENT, dtype=torch.int32, device=DEV)
        C = G.view(torch.float32).view(M + 2 * pre, ldc)[pre:pre + M, :wcols]
        me.GEMM_MODE = tag
        if tag == "bf16x3":
            me._PLANES_BY_PTR[B.data_ptr()] = planes
        try:
            me._gemm_nt(A, A.stride(0), B, B.stride(0), C, ldc, M, N, K, bias, group, act, mode, **kw)
        finally:
            me.GEMM_MODE = "f32"
            me._PLANES_BY_PTR.pop(B.data_ptr(), None)
        torch.cuda.synchronize()
        g2 = G.view(M + 2 * pre, ldc)
        outside = int((g2[:pre] != SENT).sum()) + int((g2[pre + M:] != SENT).sum()) + int((g2[pre:pre + M, wcols:] != SENT).sum())
        inside_untouched = int((g2[pre:pre + M, :N + (nfill if mode == me.EPI_FWD else 0)] == SENT).sum())
        ops = all(torch.equal(a, b) for a, b in zip(keep, (A, B, planes[0]) + ((aux,) if aux is not None else ())))
        ok = outside == 0 and inside_untouched == 0 and ops
        bad += 0 if ok else 1
        print(f"{tag:7s} M={M} N={N} K={K} group={group} mode={mode} nfill={nfill}: outside writes {outside}, unwritten outputs {inside_untouched}, operands intact {ops}", flush=True)
print("GUARD", "OK" if bad == 0 else f"FAILED ({bad})")
