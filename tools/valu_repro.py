"""VALU reproducibility next to the layer GEMMs of another stream (see tools/valu_repro.hip).  For each of 11 small out-of-line functions
(products, IEEE division, v_rcp, min/max, sqrt, 64-bit integer division, compares + selects, the rasteriser's pixel test with its
arguments in registers / on the stack / written to and read back from global memory / LDS): every thread evaluates it twice on the same
bits; counts evaluations whose two results differ -- alone, next to fp32 GEMMs, next to split-bf16 GEMMs.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/valu_repro.hip -o tools/_bin/libvalu_repro.so;  python tools/valu_repro.py [reps]"""
import ctypes
import os
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from selfreconcode_amd import mlp_engine

DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "_bin", "libvalu_repro.so"))
lib.valu_repro_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
NAMES = ["products", "IEEE division", "v_rcp * x", "min/max", "sqrt", "int64 div/mod", "compare+select", "pixel test (registers)", "pixel test (stack)", "pixel test (global round trip)", "pixel test (LDS round trip)", "pixel test (global round trip, 16-byte stores)"]
lib.valu_repro_set_buffer.argtypes = [ctypes.c_void_p]
roundtrip = torch.zeros(2048 * 256 * 16, device=DEV)
assert lib.valu_repro_set_buffer(roundtrip.data_ptr()) == 0
n = 1 << 20
torch.manual_seed(0)
inp = (torch.randn(n, 4, device=DEV) * torch.exp(torch.randn(n, 1, device=DEV) * 2.0)).contiguous()
side = torch.cuda.Stream(priority=-1)


def gemm_work(M, N=512, K=512):
    A = (torch.randn(M, K, device=DEV) * 0.3).contiguous(); B = (torch.randn(N, K, device=DEV) * 0.05).contiguous()
    C = torch.zeros(M, N, device=DEV); planes = mlp_engine.split_bf16x3(B, K); bias = torch.zeros(N, device=DEV)

    def work():
        mlp_engine._PLANES_BY_PTR[B.data_ptr()] = planes
        for _ in range(4):
            mlp_engine._gemm_nt(A, K, B, K, C, N, M, N, K, bias, 1, mlp_engine.ACT_NONE, mlp_engine.EPI_FWD)
    work.keep = (A, B, C, planes, bias)
    return work


def arm(name, work):
    counters = torch.zeros(32, dtype=torch.int64, device=DEV)
    torch.cuda.synchronize()
    for i in range(reps):
        if work is not None:
            work()
        with torch.cuda.stream(side):
            for v in range(12):
                rc = lib.valu_repro_launch(v, inp.data_ptr(), n, counters.data_ptr(), 8, side.cuda_stream)
                assert rc == 0, rc
    torch.cuda.synchronize()
    c = counters.tolist()
    total = reps * n * 8
    print(name + ":  " + ";  ".join("%s %d%s" % (NAMES[v], c[2 * v], "" if c[2 * v] == 0 else " (worst rel. diff %.1e)" % struct.unpack("f", struct.pack("I", c[2 * v + 1] & 0xffffffff))[0])
                                 for v in range(12)) + "   [of %.1e evaluations each]" % total, flush=True)


def torch_mm(dtype):
    a = torch.randn(16384, 4096, device=DEV, dtype=dtype); b = torch.randn(4096, 4096, device=DEV, dtype=dtype); out = torch.empty(16384, 4096, device=DEV, dtype=dtype)

    def work():
        for _ in range(2):
            torch.matmul(a, b, out=out)
    work.keep = (a, b, out)
    return work


lib.valu_repro_neighbour.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
nb_out = torch.zeros(2048 * 256, device=DEV)


def neighbour(kind, iters):
    def work():
        rc = lib.valu_repro_neighbour(kind, nb_out.data_ptr(), iters, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    return work


lib.valu_repro_neighbour.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
nb_out = torch.zeros(2048 * 256, device=DEV)


def neighbour(kind, iters):
    def work():
        rc = lib.valu_repro_neighbour(kind, nb_out.data_ptr(), iters, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
    return work


w = gemm_work(196608)
if os.environ.get("SR_VR_LOADS") == "1":
    lib.valu_repro_neighbour_loads.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    big = torch.randn(64 << 20, device=DEV)          # 256 MB

    def loads(kind, width, iters):
        def work():
            rc = lib.valu_repro_neighbour_loads(kind, width, big.data_ptr(), big.numel(), nb_out.data_ptr(), iters, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        return work
    lib.valu_repro_neighbour_loads_lds.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]

    def loads_lds(iters):
        def work():
            rc = lib.valu_repro_neighbour_loads_lds(big.data_ptr(), big.numel(), nb_out.data_ptr(), iters, torch.cuda.current_stream().cuda_stream)
            assert rc == 0, rc
        return work
    if os.environ.get("SR_VR_MFMA_KINDS") == "1":
        arm("next to the 16-byte load stream + v_mfma_f32_16x16x32_bf16", loads(3, 4, 600))
        arm("next to the 16-byte load stream + v_mfma_f32_32x32x16_f16", loads(4, 4, 600))
        arm("next to the 16-byte load stream + v_mfma_f32_32x32x8_bf16_1k", loads(5, 4, 600))
        arm("next to the 16-byte load stream + v_mfma_f32_32x32x16_bf16 (control)", loads(1, 4, 600))
        sys.exit(0)
    if os.environ.get("SR_VR_LDS_ONLY") == "1":
        arm("next to the stream of 16-byte loads with LDS as their destination + the bf16 MFMAs", loads_lds(600))
        arm("next to that stream + 24 v_mfma_f32_32x32x16_bf16 per step (VGPR destination, control)", loads(1, 4, 600))
        sys.exit(0)
    arm("next to a three-stage stream of 16-byte loads", loads(0, 4, 600))
    arm("next to that stream + 24 v_mfma_f32_32x32x16_bf16 per step", loads(1, 4, 600))
    arm("next to that stream + 12 v_mfma_f32_32x32x2_f32 per step", loads(2, 4, 600))
    arm("next to a stream of 8-byte loads + the bf16 MFMAs", loads(1, 2, 600))
    arm("next to a stream of 4-byte loads + the bf16 MFMAs", loads(1, 1, 600))
    sys.exit(0)
if os.environ.get("SR_VR_COMBO") == "1":
    third = torch.cuda.Stream()
    nb0, nb2 = neighbour(0, 6000), neighbour(2, 3000)

    def combo(parts):
        def work():
            if "gemm" in parts:
                w()
            with torch.cuda.stream(third):
                if "mfma" in parts:
                    nb0()
                if "lds" in parts:
                    nb2()
        return work
    mlp_engine.set_gemm_mode("f32")
    arm("next to fp32 GEMMs + the bf16 MFMA loop on a third stream", combo(("gemm", "mfma")))
    arm("next to fp32 GEMMs + the LDS loop on a third stream", combo(("gemm", "lds")))
    arm("next to the bf16 MFMA loop + the LDS loop", combo(("mfma", "lds")))
    sys.exit(0)
if os.environ.get("SR_VR_ONLY_BF16X3") == "1":
    mlp_engine.set_gemm_mode("bf16x3")
    arm("next to split-bf16 GEMMs", w)
    sys.exit(0)
arm("alone", None)
if os.environ.get("SR_VR_SYNTHETIC", "1") == "1":
    arm("next to a register-only loop of v_mfma_f32_32x32x16_bf16", neighbour(0, 6000))
    arm("next to a register-only loop of v_mfma_f32_32x32x2_f32", neighbour(1, 1500))
    arm("next to an LDS loop (ds_write_b64 / b128, ds_read_b128, s_barrier; 73.7 KB per workgroup)", neighbour(2, 3000))
    arm("next to a v_cvt_pk_bf16_f32 loop", neighbour(3, 20000))
arm("next to torch.matmul bf16 (hipBLASLt)", torch_mm(torch.bfloat16))
arm("next to torch.matmul fp16", torch_mm(torch.float16))
arm("next to torch.matmul fp32", torch_mm(torch.float32))
mlp_engine.set_gemm_mode("f32")
arm("next to fp32 GEMMs", w)
mlp_engine.set_gemm_mode("bf16x3")
arm("next to split-bf16 GEMMs", w)
