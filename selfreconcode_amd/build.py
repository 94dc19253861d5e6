"""Builds libselfrecon_hip.so (all HIP kernels + the C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the source
snapshot to the GPU box (see .gitignore).  `python -m selfreconcode_amd.build [--force]`.
"""
import glob
import hashlib
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libselfrecon_hip.so")
STAMP = os.path.join(LIBDIR, ".build_stamp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _digest():
    h = hashlib.sha256()
    for f in _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(PKG, "..", "include", "selfrecon_hip.h")]:
        with open(f, "rb") as fh:
            h.update(os.path.relpath(f, PKG).encode()); h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def is_stale():
    if not os.path.isfile(LIB) or not os.path.isfile(STAMP):
        return True
    return open(STAMP).read().strip() != _digest()


def build_lib(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not is_stale():
        return LIB
    objs = []
    procs = []
    for src in _sources():
        obj = os.path.join(LIBDIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(_digest())
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
