"""TEST INFRASTRUCTURE -- freezes the output of the reference's own marching-cubes kernels (host build of
/root/reference/MCGpu/CudaKernels.cu, contracted like nvcc's default; see oracle/Makefile) on the small volumes of
tests/test_mc_reference_pin.py into tests/golden/mc_ref.npz, canonicalised by lattice-edge key.
Run here (needs /root/reference):  python -m oracle.gen_mc_ref_golden"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mc as mco                      # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_mc_reference_pin import CASES, STEP, ORG, field   # noqa: E402

out = {}
for n, (shape, kind) in enumerate(CASES[:4]):
    v, k, f = mco.canonical(*mco.reference_marching_cubes(field(shape, kind), STEP, ORG, 0.0, "fma"))
    out[f"v{n}"], out[f"k{n}"], out[f"f{n}"] = v, k, f.astype(np.int32)
    print(shape, kind, v.shape, f.shape)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mc_ref.npz"), **out)
