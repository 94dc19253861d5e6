"""TEST INFRASTRUCTURE -- the reference's own marching-cubes kernels (host build of /root/reference/MCGpu/CudaKernels.cu, contracted
like nvcc's default: oracle/_ref/libmc_ref_fma.so) run at the sizes BASELINE.json names -- 225 x 321 x 129 and 513^3 -- on the
volume of tests/test_mc_size_pin.py; counts, SHA-256 digests of the canonicalised output and a strided coordinate sample go to
tests/golden/mc_size.npz.  Run here (needs /root/reference or the prebuilt library; ~10 GB of memory, minutes):
    python -m oracle.gen_mc_size_golden"""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import mc as mco                                          # noqa: E402
from test_mc_size_pin import SIZES, STEP, ORG, volume, digest         # noqa: E402

out = {}
for name, shape in SIZES.items():
    t0 = time.time()
    s = volume(shape)
    t1 = time.time()
    v, k, f = mco.canonical(*mco.reference_marching_cubes(s, STEP, ORG, 0.0, "fma"))
    V, F, sv, sf = digest(v, f)
    assert f.min() >= 0, "the surface must stay inside the volume"
    out[name + "_shape"] = np.array(shape); out[name + "_V"] = np.array(V); out[name + "_F"] = np.array(F)
    out[name + "_sha_v"] = np.array(sv); out[name + "_sha_f"] = np.array(sf); out[name + "_v_sample"] = v[::997][:512].copy()
    print(name, shape, "V", V, "F", F, sv[:16], sf[:16], "volume %.1f s, reference kernels %.1f s" % (t1 - t0, time.time() - t1), flush=True)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "mc_size.npz"), **out)
