"""a7 pinned to the reference's OWN kernels: /root/reference/FastMinv/Matrix3x3InvKernels.cu (cu3x3MInv, cu3x3MInv_backward) compiled
for the host by oracle/Makefile (oracle/_ref/libminv_ref_{fma,nofma}.so; the prebuilt files travel to the GPU box).

  * CPU: the oracle's restatement (oracle/torch_oracle.py::minv3x3 / minv3x3_backward) against them;
  * GPU: the HIP kernels (csrc/minv.hip, through ext/FastMinv.py) are BIT-EQUAL to the uncontracted host build in float32 and float64
    -- inverses, the singular flags (|det| < 1e-4 -> zeros + False) and the backward -- and within a few ulp of the build in which
    the host compiler contracted mul+add (what nvcc's -fmad=true default also does, for pairs of its own choosing)."""
import numpy as np
import pytest
import torch
from oracle import minv_ref
from oracle import torch_oracle as orc
from oracle import fixtures as fx

needs_ref = pytest.mark.skipif(not minv_ref.reference_available(), reason="oracle/_ref/libminv_ref_*.so not built (needs /root/reference once)")


def _matrices(n, dtype):
    m = fx.det_array((n, 3, 3), 71, 1.0, np.float64)
    m[::7] *= 0.03                                   # |det| ~ 1e-5: below the 1e-4 threshold -> zeros + False
    m[::11, 2] = m[::11, 0] * 0.5 + m[::11, 1]       # exactly dependent rows
    m[::13] *= 0.2                                   # |det| scattered around the threshold
    if n > 5:
        m[5] = 0.0
    return m.astype(dtype)


@needs_ref
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_restatement_vs_reference_kernels(dtype):
    m = _matrices(5000, dtype)
    inv_r, ok_r = minv_ref.forward(m, "nofma")
    inv_o, ok_o = orc.minv3x3(torch.from_numpy(m))
    assert 0.1 < ok_r.mean() < 0.9                                            # both branches are exercised
    assert np.array_equal(ok_o.numpy(), ok_r) and np.array_equal(inv_o.numpy(), inv_r)      # same operation order, no contraction: bit-equal
    g = fx.det_array((5000, 3, 3), 72, 1.0, np.float64).astype(dtype)
    out_r = minv_ref.backward(g, inv_r, "nofma")
    out_o = orc.minv3x3_backward(torch.from_numpy(g), torch.from_numpy(inv_r)).numpy()       # -(C^T G C^T) as two matrix products
    mag = (np.abs(inv_r).max(axis=(1, 2)) ** 2 * np.abs(g).max(axis=(1, 2)))[:, None, None] + 1e-30      # size of the summed terms |C|^2 |G|
    assert (np.abs(out_o - out_r) / mag).max() < (2e-6 if dtype == np.float32 else 4e-15)
    # the contracted host build differs from the uncontracted one by rounding only
    inv_f, ok_f = minv_ref.forward(m, "fma")
    same = ok_f == ok_r
    assert same.mean() > 0.999
    rel = np.abs(inv_f[same] - inv_r[same]).max(axis=(1, 2)) / (np.abs(inv_r[same]).max(axis=(1, 2)) + 1e-30)
    assert np.median(rel) < (1e-6 if dtype == np.float32 else 1e-15)


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_hip_kernels_bit_equal_to_reference_kernels(dtype):
    from selfreconcode_amd.ext import FastMinv
    for n in (1, 255, 256, 257, 10000):
        m = _matrices(n, dtype)
        inv_r, ok_r = minv_ref.forward(m, "nofma")
        inv, ok = FastMinv.Fast3x3Minv(torch.from_numpy(m).cuda())
        assert np.array_equal(ok.cpu().numpy(), ok_r), n
        assert np.array_equal(inv.cpu().numpy(), inv_r), (n, np.abs(inv.cpu().numpy() - inv_r).max())
        g = fx.det_array((n, 3, 3), 72, 1.0, np.float64).astype(dtype)
        out_r = minv_ref.backward(g, inv_r, "nofma")
        out = FastMinv.Fast3x3Minv_backward(torch.from_numpy(g).cuda(), inv)
        assert np.array_equal(out.cpu().numpy(), out_r), (n, np.abs(out.cpu().numpy() - out_r).max())
