"""TEST INFRASTRUCTURE -- ctypes face of oracle/mc_oracle.c + canonicalisation helpers."""
import ctypes
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "_build", "libmc_oracle.so")


def build():
    if not os.path.isfile(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "mc_oracle.c")):
        subprocess.check_call(["make", "-C", HERE], stdout=subprocess.DEVNULL)
    return SO


def marching_cubes(sdf, step=(1., 1., 1.), origin=(0., 0., 0.), iso=0.0):
    """sdf: float32 [NX,NY,NZ] (C order).  Returns verts [V,3] f32, keys [V] i64, faces [F,3] i64
    in the oracle's sequential (cell-major) order."""
    lib = ctypes.CDLL(build())
    sdf = np.ascontiguousarray(sdf, dtype=np.float32)
    NX, NY, NZ = sdf.shape
    cap = max(1024, int(sdf.size * 0.5))
    verts = np.zeros((cap, 3), np.float32); keys = np.zeros(cap, np.int64); faces = np.zeros((cap * 2, 3), np.int64)
    nv, nf = ctypes.c_int64(0), ctypes.c_int64(0)
    f32 = ctypes.c_float
    lib.mc_oracle.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32, f32, f32, f32, f32, f32, f32,
                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                              ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    rc = lib.mc_oracle(sdf.ctypes.data, NX, NY, NZ, iso, step[0], step[1], step[2], origin[0], origin[1], origin[2],
                       verts.ctypes.data, keys.ctypes.data, cap, faces.ctypes.data, cap * 2, ctypes.byref(nv), ctypes.byref(nf))
    assert rc == 0, rc
    return verts[:nv.value].copy(), keys[:nv.value].copy(), faces[:nf.value].copy()


REF_SO = {mode: os.path.join(HERE, "_ref", f"libmc_ref_{mode}.so") for mode in ("fma", "nofma")}
REF_SRC = "/root/reference/MCGpu/CudaKernels.cu"


def reference_available():
    """True when the host build of the reference's own MC kernels exists (built here by `make -C oracle ref`;
    the prebuilt files travel to the GPU box) or can be built (the reference tree is present)."""
    if all(os.path.isfile(p) for p in REF_SO.values()):
        return True
    if os.path.isfile(REF_SRC):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)
        return True
    return False


def reference_marching_cubes(sdf, step=(1., 1., 1.), origin=(0., 0., 0.), iso=0.0, mode="fma"):
    """The REFERENCE's kernels (MCGpu/CudaKernels.cu, host build, see oracle/Makefile) on the same volume.
    mode "fma": mul+add contracted as nvcc does by default (-fmad=true); "nofma": every operation rounded separately.
    Returns verts [V,3], lattice-edge keys [V], faces [F,3] in the order the sequential emulation produced them."""
    assert reference_available()
    lib = ctypes.CDLL(REF_SO[mode])
    sdf = np.ascontiguousarray(sdf, dtype=np.float32)
    NX, NY, NZ = sdf.shape
    cv, cf = ctypes.c_int64(0), ctypes.c_int64(0)
    lib.mc_ref_capacity(NX, NY, NZ, ctypes.byref(cv), ctypes.byref(cf))
    # the reference never checks its 5 %-of-cells scratch (CudaKernels.cu:590-592): make sure this volume fits BEFORE running it
    vo, _, fo = marching_cubes(sdf, step, origin, iso)
    if len(vo) > cv.value or len(fo) > cf.value:
        raise ValueError(f"volume needs {len(vo)} vertices / {len(fo)} faces, the reference's scratch holds {cv.value} / {cf.value}")
    verts = np.zeros((cv.value, 3), np.float32); keys = np.zeros(cv.value, np.int64); faces = np.zeros((cf.value, 3), np.int64)
    nv, nf = ctypes.c_int64(0), ctypes.c_int64(0)
    f32 = ctypes.c_float
    lib.mc_ref_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, f32, f32, f32, f32, f32, f32, f32,
                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64,
                               ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    rc = lib.mc_ref_run(sdf.ctypes.data, NX, NY, NZ, iso, step[0], step[1], step[2], origin[0], origin[1], origin[2],
                        verts.ctypes.data, keys.ctypes.data, cv.value, faces.ctypes.data, cf.value, ctypes.byref(nv), ctypes.byref(nf))
    assert rc == 0, rc
    return verts[:nv.value].copy(), keys[:nv.value].copy(), faces[:nf.value].copy()


def canonical(verts, keys, faces):
    """Order vertices by lattice-edge key, rewrite faces to the new ids, sort face rows."""
    order = np.argsort(keys, kind="stable")
    rank = np.empty_like(order); rank[order] = np.arange(order.size)
    f = np.where(faces >= 0, rank[np.clip(faces, 0, None)], -1)
    f = f[np.lexsort((f[:, 2], f[:, 1], f[:, 0]))]
    return verts[order], keys[order], f
