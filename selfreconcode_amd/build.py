"""Builds libselfrecon_hip.so (all HIP kernels + the C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the source
snapshot to the GPU box (see .gitignore).  `python -m selfreconcode_amd.build [--force]`.

The library carries the digest of the sources it was built from (a marker string inside the
.so), so "is this .so the one these sources describe" needs no side file: a snapshot that ships
the .so alone is recognised as current.  Builds are serialised across processes (torchrun
ranks, pytest-xdist workers import the package at the same moment) by an flock, compile into a
private directory and are published with os.replace, so nobody ever dlopens a half-written file.
"""
import contextlib
import fcntl
import glob
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libselfrecon_hip.so")
LOCK = os.path.join(LIBDIR, ".build_lock")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
MARKER = b"SR_BUILD_DIGEST="


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _digest():
    h = hashlib.sha256()
    for f in _sources() + sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(PKG, "..", "include", "selfrecon_hip.h")]:
        with open(f, "rb") as fh:
            h.update(os.path.relpath(f, PKG).encode()); h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def embedded_digest(path=LIB):
    """The source digest the library at `path` was built from (None: no library, or one that predates the marker)."""
    try:
        with open(path, "rb") as fh:
            blob = fh.read()
    except OSError:
        return None
    i = blob.find(MARKER)
    if i < 0:
        return None
    return blob[i + len(MARKER):i + len(MARKER) + 64].decode("ascii", "replace")


def is_stale():
    return embedded_digest() != _digest()


def have_compiler():
    return os.path.isfile(HIPCC) and os.access(HIPCC, os.X_OK)


@contextlib.contextmanager
def _build_lock():
    os.makedirs(LIBDIR, exist_ok=True)
    with open(LOCK, "w") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)               # blocks while another process builds
        try:
            yield
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)


def build_lib(force=False, verbose=True):
    """Returns the path of an up-to-date library, building it if needed.  Safe to call from many processes at once: one
    builds, the others wait on the lock and then find a current library."""
    if not force and not is_stale():
        return LIB
    with _build_lock():
        if not force and not is_stale():             # somebody else built it while this process waited
            return LIB
        if not have_compiler():
            have = embedded_digest()
            raise RuntimeError(
                f"{LIB} " + ("does not exist" if not os.path.isfile(LIB) else
                             f"was built from other sources (digest {have and have[:12]}... != {_digest()[:12]}...): kernels and ctypes structs would disagree")
                + f", and {HIPCC} is not available to rebuild it.  Build on a machine with ROCm: python -m selfreconcode_amd.build")
        digest = _digest()
        tmp = tempfile.mkdtemp(prefix=".build_", dir=LIBDIR)
        try:
            objs, procs = [], []
            for src in _sources():
                obj = os.path.join(tmp, os.path.basename(src)[:-4] + ".o")
                objs.append(obj)
                cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
                if os.path.basename(src) == "minv.hip":          # (home of sr_abi_version / sr_build_arch / the digest marker)
                    cmd.insert(-4, f'-DSR_BUILD_DIGEST_STR="{digest}"')
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((src, subprocess.Popen(cmd)))
            failed = [src for src, p in procs if p.wait() != 0]
            if failed:
                raise RuntimeError(f"hipcc failed on {failed}")
            out = os.path.join(tmp, "libselfrecon_hip.so")
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            if embedded_digest(out) != digest:
                raise RuntimeError("built library does not carry the source digest")
            for obj in objs:                                     # (kept next to the library: the driver's build check looks for them)
                os.replace(obj, os.path.join(LIBDIR, os.path.basename(obj)))
            os.replace(out, LIB)                                 # atomic publish
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
