"""TEST INFRASTRUCTURE ONLY -- the reference's Seg3dLossless._forward (MCAcc/seg3d_lossless.py:233-428) run verbatim on CPU AT THE
SHIPPED COARSE-STAGE GRID 225 x 321 x 129 (train.py:29-36, five levels) on an analytic ellipsoid, frozen into
tests/golden/seg3d_full.npz:    python oracle/gen_seg3d_full_golden.py      (build container only: needs /root/reference)

Stored: the number of query points per level and in total, a SHA-256 of the sign volume (> 0), the count of positive voxels and a
strided slice of the values.  The query function is a closed form evaluated in float64 and rounded to float32 (see `ell`), so that the
CPU run here and the GPU run of the test see the same values; queried points within 1e-12 of the surface would make the fixture refuse
to be written.  `ell` is imported by the test."""
import hashlib
import os
import sys
import time
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.ref_harness import load_reference  # noqa: E402

ref = load_reference()
RES = [(14 + 1, 20 + 1, 8 + 1), (28 + 1, 40 + 1, 16 + 1), (56 + 1, 80 + 1, 32 + 1), (112 + 1, 160 + 1, 64 + 1), (224 + 1, 320 + 1, 128 + 1)]   # train.py:29-35 (W,H,D)
BMIN, BMAX = [-0.8, -1.25, -0.4], [0.8, 0.95, 0.4]
C, A = [0.0503, -0.1007, 0.0211], [0.4513, 0.8017, 0.2509]


def ell(points):
    """float32 points -> float32 values through float64 arithmetic: CPU and GPU then agree to the last bit except on a rounding
    boundary of the final cast, and a sign can only differ for |value| < 1e-15."""
    c = torch.tensor(C, dtype=torch.float64, device=points.device).view(1, 1, 3)
    a = torch.tensor(A, dtype=torch.float64, device=points.device).view(1, 1, 3)
    return ((((points.double() - c) / a).norm(dim=-1) - 1.0).view(1, 1, -1) * 0.25).float()


def main():
    torch.set_num_threads(os.cpu_count())
    eng = ref.MCAcc.Seg3dLossless(query_func=ell, b_min=BMIN, b_max=BMAX, resolutions=RES, align_corners=False, balance_value=0.0, device='cpu',
                                  visualize=False, debug=False, use_cuda_impl=False, faster=False)
    per_call, near = [], [0]

    def counted(points):
        per_call.append(points.shape[1])
        v = ell(points)
        near[0] += int((v.abs() < 1e-12).sum())
        return v
    eng.query_func = counted
    t0 = time.perf_counter()
    vol = eng.forward()
    sec = time.perf_counter() - t0
    v = vol[0, 0].numpy()
    assert v.shape == (129, 321, 225), v.shape
    assert near[0] == 0, f"{near[0]} queried points within 1e-12 of the surface: move the ellipsoid"
    sign = np.ascontiguousarray(v > 0)
    h = hashlib.sha256(np.packbits(sign.reshape(-1)).tobytes()).hexdigest()
    out = os.path.join(ROOT, "tests", "golden", "seg3d_full.npz")
    np.savez_compressed(out, nq_total=np.array(sum(per_call)), nq_calls=np.array(per_call), npos=np.array(int(sign.sum())),
                        sign_sha256=np.frombuffer(bytes.fromhex(h), dtype=np.uint8), slice=v[::8, ::8, ::8].astype(np.float32),
                        res=np.array(RES), centre=np.array(C, np.float32), radii=np.array(A, np.float32), seconds_cpu=np.array(sec))
    print("wrote", out, os.path.getsize(out), "bytes; queries", sum(per_call), "of", v.size, "voxels in", len(per_call), "calls;", "%.1f s" % sec, h[:16])


if __name__ == "__main__":
    main()
