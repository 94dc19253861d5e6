"""Known-byte-count launches of the layer GEMM for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE in THIS kernel's access
pattern (MI355X_MICROARCH.md, HBM section: FETCH_SIZE is halved on gfx950 for wide coalesced reads, WRITE_SIZE and other widths are
uncalibrated).  Run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` (tools/profile_round.sh does);
the expected bytes are printed as JSON for the reader of the counter CSV.
  NT forward  M x 512 x 512, no activation:  reads A (M*512*4) + W (1 MB),  writes C (M*512*4)   -- operands far larger than the
  256 MB Infinity Cache at M = 524288 (1.07 GB each), so the memory-side counters see every byte once."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfreconcode_amd import mlp_engine as me
dev = "cuda:0"
out = []
for M in (524288, 131072):
    A = torch.randn(M, 512, device=dev); B = torch.randn(512, 512, device=dev) * 0.05; C = torch.empty(M, 512, device=dev); b = torch.zeros(512, device=dev)
    for _ in range(4):
        me._gemm_nt(A, 512, B, 512, C, 512, M, 512, 512, b, 1, me.ACT_NONE, me.EPI_FWD)
    torch.cuda.synchronize()
    out.append({"kernel": "gemm_nt_kernel", "M": M, "launches": 4, "read_bytes": M * 512 * 4 + 512 * 512 * 4, "write_bytes": M * 512 * 4})
    del A, C
# An INDEPENDENT streaming kernel with the same access width (16 bytes per lane, fully coalesced): torch's vectorised copy of a 1 GiB /
# 256 MiB tensor.  It fixes the counter-to-byte factors WITHOUT assuming anything about the GEMM; the GEMM launches above are then read
# with those factors (tools/summarize_profile.py), so "measured / algorithmic" of the GEMM is a measurement, not an identity.
for M in (524288, 131072):
    src = torch.randn(M, 512, device=dev); dst = torch.empty_like(src)
    for _ in range(4):
        torch.mul(src, 1.0, out=dst)          # vectorized_elementwise_kernel<4, ...>: float4 loads and stores (a plain copy_ would be a runtime memcpy)
    torch.cuda.synchronize()
    out.append({"kernel": "stream_copy", "M": M, "launches": 4, "read_bytes": M * 512 * 4, "write_bytes": M * 512 * 4})
    del src, dst
print(json.dumps(out))
