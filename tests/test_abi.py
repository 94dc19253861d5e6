"""CPU-side checks of the drop-in boundary: the shared library builds for gfx950, loads, and
exports exactly the entry points include/selfrecon_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = []
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", fn)).read()
            names += re.findall(r"^\s*(?:int|const char\*|int64_t)\s+(sr_\w+)\s*\(", txt, flags=re.M)
    return sorted(set(names))


def test_library_builds_and_exports_every_declared_symbol():
    from selfreconcode_amd.build import build_lib
    lib = ctypes.CDLL(build_lib(verbose=False))
    decl = _declared()
    assert len(decl) >= 12
    missing = [n for n in decl if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    from selfreconcode_amd import _lib
    decl = set(_declared()) - {"sr_abi_version", "sr_build_arch", "sr_build_digest"}
    assert decl == set(_lib.SIGNATURES), decl ^ set(_lib.SIGNATURES)
    assert _lib.build_arch() == "gfx950" and _lib.abi_version() >= 1
    from selfreconcode_amd import build
    assert _lib.build_digest() == build._digest() == build.embedded_digest()      # the loaded library is the one these sources describe


def test_no_cpu_fallback():
    """The HIP operators must refuse CPU tensors instead of silently computing elsewhere."""
    import torch
    from selfreconcode_amd.ext import FastMinv, GridSamplerMine
    with pytest.raises(RuntimeError):
        FastMinv.Fast3x3Minv(torch.eye(3).view(1, 3, 3))
    with pytest.raises(RuntimeError):
        GridSamplerMine.forward(torch.zeros(1, 2, 3, 3, 3), torch.zeros(1, 1, 1, 4, 3), 0, 1)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under selfreconcode_amd/ may reference it."""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "selfreconcode_amd")):
        for fn in fns:
            if fn.endswith(".py"):
                txt = open(os.path.join(dp, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "/root/reference" in txt:
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_ctypes_structs_have_the_size_the_header_declares(tmp_path):
    """Every argument struct crosses the boundary by pointer: the ctypes mirror in _lib.py and the C declaration must agree
    (gcc compiles the header as plain C -- it has to stay a C header -- and prints sizeof for each)."""
    import subprocess
    from selfreconcode_amd import _lib
    pairs = {"sr_tensor5": _lib.SrTensor5, "sr_gemm_args": _lib.SrGemmArgs, "sr_gemm_tn_args": _lib.SrGemmTnArgs, "sr_gemm_tn_group_args": _lib.SrGemmTnGroupArgs, "sr_lbs_args": _lib.SrLbsArgs,
             "sr_newton_args": _lib.SrNewtonArgs, "sr_newton2_args": _lib.SrNewton2Args, "sr_chain_args": _lib.SrChainArgs,
             "sr_refine_args": _lib.SrRefineArgs, "sr_pack_layer": _lib.SrPackLayer, "sr_pack_table": _lib.SrPackTable,
             "sr_unpack_layer": _lib.SrUnpackLayer, "sr_unpack_table": _lib.SrUnpackTable, "sr_camera": _lib.SrCamera,
             "sr_ray_pixels": _lib.SrRayPixels, "sr_adam_tensor": _lib.SrAdamTensor, "sr_adam_table": _lib.SrAdamTable}
    txt = open(os.path.join(ROOT, "include", "selfrecon_hip.h")).read()
    declared = set(re.findall(r"\}\s*(sr_\w+);", txt))
    assert declared == set(pairs), declared ^ set(pairs)
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "selfrecon_hip.h"\nint main(void){\n'
                   + "".join(f'printf("{n} %zu\\n", sizeof({n}));\n' for n in pairs) + "return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    sizes = dict(line.split() for line in out.strip().splitlines())
    for n, cls in pairs.items():
        assert int(sizes[n]) == ctypes.sizeof(cls), (n, sizes[n], ctypes.sizeof(cls))


def test_pack_cache_entries_die_with_their_parameters():
    """mlp_engine keeps packed weights / gradient buffers per parameter; the entry must not outlive (or keep alive) the module."""
    import gc
    import torch
    from selfreconcode_amd import mlp_engine as me

    class Stub:                       # what _pack_entry needs of an entry, without a GPU
        pass
    lin = torch.nn.Linear(8, 4)
    key = id(lin.weight)
    W = torch.zeros(4, 8)
    e = me._pack_entry(key, ("sig",), lambda: {"W": W, "WT": W.t().contiguous(), "norms": None}, owner=lin.weight)
    assert key in me._PACK_CACHE and me._ENTRY_BY_PTR[W.data_ptr()] is e
    import weakref
    e["src"] = (weakref.ref(lin.weight),)
    del lin
    gc.collect()
    assert key not in me._PACK_CACHE and W.data_ptr() not in me._ENTRY_BY_PTR and W.data_ptr() not in me._WT_BY_PTR
