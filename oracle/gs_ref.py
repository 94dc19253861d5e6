"""TEST INFRASTRUCTURE -- ctypes face of oracle/_ref/libgs_ref_{fma,nofma}.so: the REFERENCE's own grid-sampler kernels
(/root/reference/MCAcc/cuda/GridSamplerMineKernel.cu: grid_sampler_3d_kernel :162-328, grid_sampler_3d_backward_kernel :333-570,
grid_sampler_3d_backward_backward_kernel :575-914) compiled for the host by oracle/Makefile (`refgs`).  Arrays are numpy, float32 or
float64, any strides; the wrappers allocate the outputs the way the reference's launcher functions do (:918-1022: grad_input /
grad_grad_output zero-filled, grad_grid uninitialised -- the kernels write every element of it)."""
import ctypes
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = {mode: os.path.join(HERE, "_ref", f"libgs_ref_{mode}.so") for mode in ("fma", "nofma")}
REF_SRC = "/root/reference/MCAcc/cuda/GridSamplerMineKernel.cu"
BILINEAR, BORDER = 0, 1


def reference_available():
    if all(os.path.isfile(p) for p in REF_SO.values()):
        return True
    if os.path.isfile(REF_SRC):
        subprocess.check_call(["make", "-C", HERE, "refgs"], stdout=subprocess.DEVNULL)
        return True
    return False


_LIBS = {}


def _lib(mode):
    assert reference_available()
    if mode not in _LIBS:
        _LIBS[mode] = ctypes.CDLL(REF_SO[mode])
    return _LIBS[mode]


def _suffix(a):
    return {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[a.dtype]


def _desc(a):
    """(pointer, sizes[5], strides[5] in elements) of a 5-D array"""
    assert a.ndim == 5
    I5 = ctypes.c_int64 * 5
    return ctypes.c_void_p(a.ctypes.data), I5(*a.shape), I5(*[s // a.itemsize for s in a.strides])


def forward(inp, grid, mode="nofma"):
    N, C = inp.shape[:2]
    out = np.full((N, C) + grid.shape[1:4], np.nan, inp.dtype)
    pi, si, ti = _desc(inp); pg, sg, tg = _desc(grid); po, so, to = _desc(out)
    getattr(_lib(mode), "gs_ref_fwd_" + _suffix(inp))(pi, si, ti, pg, sg, tg, po, so, to, BILINEAR, BORDER)
    return out


def backward(inp, grid, grad_output, mode="nofma"):
    """-> (grad_input, grad_grid)"""
    ginput = np.zeros(inp.shape, inp.dtype)
    ggrid = np.full(grid.shape, np.nan, grid.dtype)
    pi, si, ti = _desc(inp); pg, sg, tg = _desc(grid); po, so, to = _desc(grad_output)
    pgi, _, tgi = _desc(ginput); pgg, _, tgg = _desc(ggrid)
    getattr(_lib(mode), "gs_ref_bwd_" + _suffix(inp))(po, so, to, pi, si, ti, pg, sg, tg, pgi, tgi, pgg, tgg, BILINEAR, BORDER)
    return ginput, ggrid


def dbackward(gout_input, gout_grid, inp, grid, grad_output, mode="nofma"):
    """cotangents (gout_input on grad_input, gout_grid on grad_grid) -> (grad_input, grad_grid, grad_grad_output)"""
    ginput = np.zeros(inp.shape, inp.dtype)
    ggrid = np.full(grid.shape, np.nan, grid.dtype)
    ggout = np.zeros(grad_output.shape, grad_output.dtype)
    pi, si, ti = _desc(inp); pg, sg, tg = _desc(grid); po, so, to = _desc(grad_output)
    pa, _, ta = _desc(gout_input); pb, _, tb = _desc(gout_grid)
    pgi, _, tgi = _desc(ginput); pgg, _, tgg = _desc(ggrid); pgo, _, tgo = _desc(ggout)
    getattr(_lib(mode), "gs_ref_dbwd_" + _suffix(inp))(pa, ta, pb, tb, po, so, to, pi, si, ti, pg, sg, tg, pgi, tgi, pgg, tgg, pgo, tgo, BILINEAR, BORDER)
    return ginput, ggrid, ggout
