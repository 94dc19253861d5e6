"""TEST INFRASTRUCTURE -- freezes the outputs of the reference's own grid-sampler kernels (host build of
/root/reference/MCAcc/cuda/GridSamplerMineKernel.cu, uncontracted; see oracle/Makefile `refgs`) on the case of
tests/test_gs_reference_pin.py into tests/golden/gs_ref.npz: forward, backward (grad_input, grad_grid) and double backward
(grad_input, grad_grid, grad_grad_output) in float32 and float64.
Run here (needs /root/reference):  python -m oracle.gen_gs_ref_golden"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gs_reference_pin import reference_outputs   # noqa: E402

out = {}
for dtype, tag in ((np.float32, "f32"), (np.float64, "f64")):
    for k, v in reference_outputs(dtype, "nofma").items():
        out[f"{tag}_{k}"] = v
        print(tag, k, v.shape, float(np.abs(v).max()))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "gs_ref.npz"), **out)
print("wrote gs_ref.npz", os.path.getsize(os.path.join(ROOT, "tests", "golden", "gs_ref.npz")))
