"""TEST INFRASTRUCTURE -- ctypes face of oracle/_ref/libminv_ref_{fma,nofma}.so: the REFERENCE's own 3x3-inverse kernels
(/root/reference/FastMinv/Matrix3x3InvKernels.cu:22-141) compiled for the host by oracle/Makefile.
mode "nofma": every operation rounded on its own (the IEEE reading of the source); "fma": mul+add contracted by the host compiler,
which is what nvcc's default -fmad=true also does -- though not necessarily for the same pairs, so only "nofma" is a bit-level pin."""
import ctypes
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = {mode: os.path.join(HERE, "_ref", f"libminv_ref_{mode}.so") for mode in ("fma", "nofma")}
REF_SRC = "/root/reference/FastMinv/Matrix3x3InvKernels.cu"


def reference_available():
    if all(os.path.isfile(p) for p in REF_SO.values()):
        return True
    if os.path.isfile(REF_SRC):
        subprocess.check_call(["make", "-C", HERE, "refminv"], stdout=subprocess.DEVNULL)
        return True
    return False


def _lib(mode):
    assert reference_available()
    return ctypes.CDLL(REF_SO[mode])


def forward(ms, mode="nofma"):
    """ms [N,3,3] float32 / float64 numpy -> (invs [N,3,3], checks [N] bool) from the reference's cu3x3MInv."""
    ms = np.ascontiguousarray(ms)
    suffix = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[ms.dtype]
    invs = np.full_like(ms, np.nan); checks = np.zeros(ms.shape[0], np.uint8)
    getattr(_lib(mode), "minv_ref_fwd_" + suffix)(ctypes.c_void_p(ms.ctypes.data), ctypes.c_void_p(invs.ctypes.data), ctypes.c_void_p(checks.ctypes.data), ctypes.c_int(ms.shape[0]))
    return invs, checks.astype(bool)


def backward(grads, invs, mode="nofma"):
    grads, invs = np.ascontiguousarray(grads), np.ascontiguousarray(invs)
    suffix = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[grads.dtype]
    outs = np.full_like(grads, np.nan)
    getattr(_lib(mode), "minv_ref_bwd_" + suffix)(ctypes.c_void_p(grads.ctypes.data), ctypes.c_void_p(invs.ctypes.data), ctypes.c_void_p(outs.ctypes.data), ctypes.c_int(grads.shape[0]))
    return outs
