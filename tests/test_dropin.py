"""`selfreconcode_amd.dropin.install()` -- the drop-in boundary of INTEGRATION.md section 2 -- exercised.

CPU (build container, /root/reference present): the reference's own files import the installed modules and every call site binds.
GPU: the reference's autograd-glue pattern (MCAcc/grid_sampler_mine.py:8-58: three calls into whatever module is registered as
`GridSamplerMine`; utils/utils.py:8-19 into `FastMinv`) runs forward -> backward -> double backward through `sys.modules` and agrees
with the CPU oracle."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="needs the reference tree (build container)")
def test_reference_files_resolve_their_extensions_to_this_package():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_dropin_check.py")], cwd=ROOT, capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONPATH=ROOT))
    assert r.returncode == 0, r.stderr[-3000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert len(rec["files"]) == 4


def test_install_registers_the_reference_import_names():
    import selfreconcode_amd.dropin as dropin
    saved = {k: sys.modules.get(k) for k in ("FastMinv", "GridSamplerMine", "MCGpu", "interp2x_boundary3d")}
    try:
        mods = dropin.install()
        import FastMinv, GridSamplerMine, MCGpu, interp2x_boundary3d          # noqa: E401  (the reference's import lines)
        assert (FastMinv, GridSamplerMine, MCGpu) == tuple(mods)
        assert all(m.__name__.startswith("selfreconcode_amd.ext.") for m in (FastMinv, GridSamplerMine, MCGpu, interp2x_boundary3d))
        for name in ("Fast3x3Minv", "Fast3x3Minv_backward"):
            assert callable(getattr(FastMinv, name))
        for name in ("forward", "backward", "dbackward"):
            assert callable(getattr(GridSamplerMine, name))
        assert callable(MCGpu.mc_gpu) and callable(MCGpu.mc_init) and callable(interp2x_boundary3d.forward) and callable(interp2x_boundary3d.backward)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


@pytest.mark.gpu
def test_reference_glue_pattern_through_the_installed_modules():
    from torch.autograd import Function
    import selfreconcode_amd.dropin as dropin
    from oracle import torch_oracle as orc
    from oracle import fixtures as fx
    dropin.install()
    G, M = sys.modules['GridSamplerMine'], sys.modules['FastMinv']

    # the call pattern of MCAcc/grid_sampler_mine.py:8-58 and utils/utils.py:8-19 (positional arguments as the reference passes them)
    class Sample(Function):
        @staticmethod
        def forward(ctx, vol, grid):
            ctx.save_for_backward(vol, grid)
            return G.forward(vol, grid, 0, 1)

        @staticmethod
        def backward(ctx, go):
            return SampleBackward.apply(*ctx.saved_tensors, go)

    class SampleBackward(Function):
        @staticmethod
        def forward(ctx, vol, grid, go):
            ctx.save_for_backward(vol, grid, go)
            return tuple(G.backward(vol, grid, go, 0, 1))

        @staticmethod
        def backward(ctx, g_vol, g_grid):
            vol, grid, go = ctx.saved_tensors
            return tuple(G.dbackward(g_vol, g_grid, vol, grid, go, 0, 1))

    class Minv(Function):
        @staticmethod
        def forward(ctx, m):
            inv, ok = M.Fast3x3Minv(m)
            ctx.save_for_backward(inv, ok)
            ctx.mark_non_differentiable(ok)
            return inv, ok

        @staticmethod
        def backward(ctx, g, _):
            return M.Fast3x3Minv_backward(g.contiguous(), ctx.saved_tensors[0]), None

    dev = "cuda:0"
    vol = fx.det_tensor((1, 5, 6, 7, 8), 1, 1.0, torch.float64)
    grid = fx.det_tensor((1, 1, 1, 50, 3), 2, 1.05, torch.float64)
    w = fx.det_tensor((1, 5, 1, 1, 50), 3, 1.0, torch.float64)
    u = fx.det_tensor((1, 1, 1, 50, 3), 4, 1.0, torch.float64)

    def second_order(sample, vol, grid, w, u):
        vol = vol.clone().requires_grad_(True); grid = grid.clone().requires_grad_(True)
        out = sample(vol, grid)
        gv, gg = torch.autograd.grad((out * w).sum(), (vol, grid), create_graph=True)        # backward
        hv, hg = torch.autograd.grad((gg * u).sum(), (vol, grid))                            # double backward
        return out.detach(), gv.detach(), gg.detach(), hv, hg
    got = second_order(Sample.apply, vol.to(dev), grid.to(dev), w.to(dev), u.to(dev))
    want = second_order(orc.grid_sample_3d, vol, grid, w, u)
    for a, b, name in zip(got, want, ("out", "grad_input", "grad_grid", "d2/dinput", "d2/dgrid")):
        torch.testing.assert_close(a.cpu(), b, rtol=1e-9, atol=1e-10, msg=lambda m, n=name: n + ": " + m)
    m = (torch.eye(3, dtype=torch.float64) * 1.5 + fx.det_tensor((40, 3, 3), 5, 0.4, torch.float64)).to(dev).requires_grad_(True)
    inv, ok = Minv.apply(m)
    assert bool(ok.all())
    (inv * fx.det_tensor((40, 3, 3), 6, 1.0, torch.float64).to(dev)).sum().backward()
    mo = m.detach().cpu().clone().requires_grad_(True)
    (torch.linalg.inv(mo) * fx.det_tensor((40, 3, 3), 6, 1.0, torch.float64)).sum().backward()
    torch.testing.assert_close(inv.detach().cpu(), torch.linalg.inv(mo.detach()), rtol=1e-9, atol=1e-10)
    torch.testing.assert_close(m.grad.cpu(), mo.grad, rtol=1e-8, atol=1e-9)
