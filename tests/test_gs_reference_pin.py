"""a6 pinned to the reference's OWN kernels: /root/reference/MCAcc/cuda/GridSamplerMineKernel.cu -- forward (K3, :162-328), backward
(K4, :333-570) and double backward (K5, :575-914) -- compiled for the host by oracle/Makefile (oracle/_ref/libgs_ref_{fma,nofma}.so; the
prebuilt files travel to the GPU box) and frozen into tests/golden/gs_ref.npz by oracle/gen_gs_ref_golden.py (which keeps the pin where
neither the reference nor oracle/_ref exists).

  * CPU: the oracle's restatement (oracle/torch_oracle.py::grid_sample_3d + torch autograd to second order) against the kernels;
  * GPU: the HIP kernels (csrc/gridsample.hip through ext/GridSamplerMine.py) against the kernels' outputs, float32 and float64,
    contiguous NCDHW and the channel-last layout the skinning volume uses, incl. points outside the volume (border rule :44-60).
Tolerances: float64 1e-12 of the output scale, float32 2e-6 (the kernels promote the unnormalisation to double, :210-212; sums of 8
corner terms are reordered)."""
import os
import numpy as np
import pytest
import torch
from oracle import gs_ref
from oracle import torch_oracle as orc
from oracle import fixtures as fx

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gs_ref.npz")
needs_ref = pytest.mark.skipif(not gs_ref.reference_available(), reason="oracle/_ref/libgs_ref_*.so not built (needs /root/reference once)")
C, D, H, W, P = 24, 9, 13, 11, 257          # 24 channels like the skinning-weight volume; P not a multiple of anything


def case(dtype):
    """Deterministic inputs: a smooth + noisy volume, points in [-1.15, 1.15]^3 (some outside: border rule), three on exact voxel
    centres / borders, and the three cotangents."""
    inp = fx.det_array((1, C, D, H, W), 501, 1.0, np.float64)
    grid = fx.det_array((1, 1, 1, P, 3), 502, 1.15, np.float64)
    grid[0, 0, 0, 0] = [-1.0, 1.0, 0.0]; grid[0, 0, 0, 1] = [1.0 - 1.0 / W, -1.0 + 1.0 / H, 0.0]; grid[0, 0, 0, 2] = [-1.3, 1.3, 2.0]
    gout = fx.det_array((1, C, 1, 1, P), 503, 1.0, np.float64)
    goi = fx.det_array((1, C, D, H, W), 504, 1.0, np.float64)
    gog = fx.det_array((1, 1, 1, P, 3), 505, 1.0, np.float64)
    return [a.astype(dtype) for a in (inp, grid, gout, goi, gog)]


def reference_outputs(dtype, mode="nofma"):
    inp, grid, gout, goi, gog = case(dtype)
    out = gs_ref.forward(inp, grid, mode)
    gi, gg = gs_ref.backward(inp, grid, gout, mode)
    di, dg, dgo = gs_ref.dbackward(goi, gog, inp, grid, gout, mode)
    return dict(out=out, gi=gi, gg=gg, di=di, dg=dg, dgo=dgo)


def oracle_outputs(dtype):
    inp, grid, gout, goi, gog = [torch.from_numpy(a).requires_grad_(i < 3) for i, a in enumerate(case(dtype))]
    out = orc.grid_sample_3d(inp, grid)
    gi, gg = torch.autograd.grad(out, (inp, grid), gout, create_graph=True)
    S = (gi * goi).sum() + (gg * gog).sum()
    di, dg, dgo = torch.autograd.grad(S, (inp, grid, gout), allow_unused=True)
    z = lambda t, like: torch.zeros_like(like) if t is None else t
    return {k: v.detach().numpy() for k, v in dict(out=out, gi=gi, gg=gg, di=z(di, inp), dg=z(dg, grid), dgo=z(dgo, gout)).items()}


def _close(a, b, tol, what):
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err < tol, (what, err)


TOL = {np.float32: 2e-6, np.float64: 1e-12}


@needs_ref
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_restatement_vs_reference_kernels(dtype):
    r, o = reference_outputs(dtype), oracle_outputs(dtype)
    assert np.isfinite(r["gg"]).all() and np.isfinite(r["dg"]).all()              # the kernels write every element of grad_grid
    assert np.abs(r["gg"][0, 0, 0, 2]).max() == 0.0                                # a point outside on all axes: zero gradient (border rule)
    for k in r:
        _close(o[k], r[k], TOL[dtype] * (4 if dtype == np.float32 else 1), k)
    # the double backward's grad_input is the part of the sampler linear in the volume: d/d input of <grad_grid, gog>
    assert np.abs(r["di"]).max() > 0 and np.abs(r["dgo"]).max() > 0
    f = reference_outputs(dtype, "fma")                                            # contraction changes rounding only
    for k in r:
        _close(f[k], r[k], TOL[dtype], k + " (fma build)")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_restatement_vs_frozen_reference_outputs(dtype):
    g = np.load(GOLDEN)
    tag = {np.float32: "f32", np.float64: "f64"}[dtype]
    o = oracle_outputs(dtype)
    for k in o:
        _close(o[k], g[f"{tag}_{k}"], TOL[dtype] * (4 if dtype == np.float32 else 1), k)
    if gs_ref.reference_available():                                               # the fixture IS the library's output
        r = reference_outputs(dtype)
        for k in r:
            assert np.array_equal(r[k], g[f"{tag}_{k}"]), k


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["ncdhw", "channels_last"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_hip_kernels_vs_reference_kernels(dtype, layout):
    from selfreconcode_amd.ext import GridSamplerMine
    g = np.load(GOLDEN)
    tag = {np.float32: "f32", np.float64: "f64"}[dtype]
    inp, grid, gout, goi, gog = [torch.from_numpy(a).cuda() for a in case(dtype)]
    if layout == "channels_last":
        inp = inp.contiguous(memory_format=torch.channels_last_3d)
        goi = goi.contiguous(memory_format=torch.channels_last_3d)
    out = GridSamplerMine.forward(inp, grid, 0, 1)
    gi, gg = GridSamplerMine.backward(inp, grid, gout, 0, 1)
    di, dg, dgo = GridSamplerMine.dbackward(goi, gog, inp, grid, gout, 0, 1)
    tol = TOL[dtype] * (4 if dtype == np.float32 else 1)
    for k, v in dict(out=out, gi=gi, gg=gg, di=di, dg=dg, dgo=dgo).items():
        _close(v.cpu().numpy(), g[f"{tag}_{k}"], tol, k)
    # without the volume-sized outputs (what the autograd glue asks for): same point gradients
    _, gg2 = GridSamplerMine.backward(inp, grid, gout, 0, 1, want_grad_input=False)
    _, dg2, dgo2 = GridSamplerMine.dbackward(None, gog, inp, grid, gout, 0, 1, want_grad_input=False)
    _close(gg2.cpu().numpy(), g[f"{tag}_gg"], tol, "gg (no grad_input)")          # (the channel-last fast path sums the corners in another order: not bit-equal to the general one)
    zero = dict(zip(("di", "dg", "dgo"), GridSamplerMine.dbackward(torch.zeros_like(goi), gog, inp, grid, gout, 0, 1)))
    _close(dg2.cpu().numpy(), zero["dg"].cpu().numpy(), tol, "dg (no grad_input)"); _close(dgo2.cpu().numpy(), zero["dgo"].cpu().numpy(), tol, "dgo (no grad_input)")
