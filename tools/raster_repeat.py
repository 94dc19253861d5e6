"""Is the hard mesh rasteriser reproducible on one input, alone and next to other streams' work?  (Hunt for the rare ray-selection
difference of the split-bf16 mode)  Builds the fine-stage bench scene, deforms + projects the template once, then
rasterises the SAME (xy, z, faces) `reps` times on a side stream and counts pixels that differ from the first answer --
  alone | next to the sdf network's forward over the template on the main stream (f32, then bf16x3) |
  with the allocator cache emptied before every call.
    python tools/raster_repeat.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import test_full_size_parity_gpu as T
from selfreconcode_amd import mlp_engine
from selfreconcode_amd.ops import rasterize_meshes

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
mlp_engine.set_gemm_mode(os.environ.get("SR_RR_DEFORM_MODE", "bf16x3"))
mlp_engine.set_deferred_param_grads(True)
net, ds, conf = T._bench_scene("fine")
DEV = T.DEV
fids = torch.tensor([17], device=DEV)
datas = ds.batch(fids)
H, W = ds.H, ds.W
# the projection of the iteration itself: one forward with a debug dict
dbg = {}
rand = {k: v.to(DEV) for k, v in T._rand(700000).items()}
net.refiner_stream = "side"
loss = net(datas, 2048, T.RATIO, fids, rand=rand, debug=dbg)
loss.backward(); net.propagateTmpPsGrad(fids, T.RATIO)
xy, z = dbg["proj_xy"].clone(), dbg["proj_z"].clone()
torch.cuda.synchronize()
faces = net.Tmpfs
side = torch.cuda.Stream(priority=int(os.environ.get("SR_RR_PRIORITY", "-1")))
main = torch.cuda.current_stream()
V = net.TmpVs.detach().clone()


def loop(name, main_work=None, before=None):
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ref = rasterize_meshes(xy, z, faces, H, W).pix_to_face.clone()
    torch.cuda.synchronize()
    bad = torch.zeros((), dtype=torch.int64, device=DEV)
    calls = torch.zeros((), dtype=torch.int64, device=DEV)
    for i in range(reps):
        if before is not None:
            torch.cuda.synchronize(); before()
        if main_work is not None:
            main_work()
        with torch.cuda.stream(side):
            p = rasterize_meshes(xy, z, faces, H, W).pix_to_face
            n = (p != ref).sum()
            bad += n; calls += (n > 0)
    torch.cuda.synchronize()
    print(f"{name}: {int(calls)} of {reps} calls differ from the first answer ({int(bad)} pixels in total)", flush=True)


def sdf_forward():
    with torch.no_grad():
        net.sdf(V)


def gemm_work(M, N=512, K=512):
    A = (torch.randn(M, K, device=DEV) * 0.3).contiguous()
    B = (torch.randn(N, K, device=DEV) * 0.05).contiguous()
    C = torch.zeros(M, N, device=DEV)
    planes = mlp_engine.split_bf16x3(B, K)
    bias = torch.zeros(N, device=DEV)

    def work():
        mlp_engine._PLANES_BY_PTR[B.data_ptr()] = planes
        for _ in range(4):
            mlp_engine._gemm_nt(A, K, B, K, C, N, M, N, K, bias, 1, mlp_engine.ACT_NONE, mlp_engine.EPI_FWD)
    work.keep = (A, B, C, planes, bias)
    return work


def split_loop(name, main_work):
    """the C entry point directly, with the depth-buffer kept: does the z-buffer (pass 1: atomics) differ, or only what pass 2 makes of it?"""
    from selfreconcode_amd import _lib
    xyc, zc, fc = xy.contiguous().float(), z.contiguous().float(), faces.contiguous()
    N, Vn = xyc.shape[0], xyc.shape[1]

    def call():
        zb = torch.empty((N, H, W), dtype=torch.int64, device=DEV); p2f = torch.empty((N, H, W), dtype=torch.int64, device=DEV)
        bary = torch.empty((N, H, W, 3), dtype=torch.float32, device=DEV); zo = torch.empty((N, H, W), dtype=torch.float32, device=DEV)
        _lib.call("sr_rasterize_meshes", _lib.ptr(xyc), _lib.ptr(zc), _lib.ptr(fc), N, Vn, fc.shape[0], H, W, _lib.ptr(zb), _lib.ptr(p2f), _lib.ptr(bary), _lib.ptr(zo),
                  _lib.stream_of(xyc))
        return zb, p2f, bary
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        rz, rp, rb = call()
    torch.cuda.synchronize()
    cnt = torch.zeros((4,), dtype=torch.int64, device=DEV)
    for i in range(reps):
        main_work()
        with torch.cuda.stream(side):
            zb, p2f, bary = call()
            dz, dp = (zb != rz), (p2f != rp)
            cnt += torch.stack([dz.sum(), dp.sum(), (dp & ~dz).sum(), (dz.sum() > 0).long()])
    torch.cuda.synchronize()
    c = cnt.tolist()
    print(f"{name}: z-buffer words that differ {c[0]} (in {c[3]} of {reps} calls), pix_to_face entries that differ {c[1]}, of those with an EQUAL z-buffer word {c[2]}", flush=True)


only = os.environ.get("SR_RR_ONLY")
if only == "split2":
    mlp_engine.set_gemm_mode("bf16x3")
    split_loop("side-stream priority %s, next to 196608-row GEMMs (bf16x3)" % os.environ.get("SR_RR_PRIORITY", "-1"), gemm_work(196608))
    split_loop("next to 16384-row GEMMs (bf16x3: 512 workgroups, one round)", gemm_work(16384))
    split_loop("next to 65536-row GEMMs (bf16x3)", gemm_work(65536))
    side = main
    split_loop("SAME stream as the 196608-row GEMMs (bf16x3)", gemm_work(196608))
    sys.exit(0)
if only == "split":
    mlp_engine.set_gemm_mode("bf16x3")
    split_loop("next to plain 196608 x 512 x 512 GEMMs (bf16x3)", gemm_work(196608))
    x = torch.rand(1 << 22, device=DEV) + 0.5; y = torch.rand(1 << 22, device=DEV) + 0.5
    idx = torch.randint(0, 1 << 16, (1 << 22,), device=DEV); vals = torch.randint(0, 1 << 40, (1 << 22,), device=DEV)
    w = gemm_work(196608)
    with torch.cuda.stream(side):
        rq = x / y + torch.sqrt(x) * y
        ra = torch.full((1 << 16,), 1 << 62, dtype=torch.int64, device=DEV).scatter_reduce_(0, idx, vals, "amin")
    torch.cuda.synchronize()
    cnt = torch.zeros((2,), dtype=torch.int64, device=DEV)
    for i in range(reps):
        w()
        with torch.cuda.stream(side):
            q = x / y + torch.sqrt(x) * y
            a = torch.full((1 << 16,), 1 << 62, dtype=torch.int64, device=DEV).scatter_reduce_(0, idx, vals, "amin")
            cnt += torch.stack([(q != rq).sum(), (a != ra).sum()])
    torch.cuda.synchronize()
    print("next to the same GEMMs: torch elementwise (div, sqrt) results that differ %d, torch scatter-amin (int64 atomics) results that differ %d" % tuple(cnt.tolist()), flush=True)
    sys.exit(0)
if only == "gemm":
    mlp_engine.set_gemm_mode("bf16x3")
    loop("next to plain 196608 x 512 x 512 GEMMs (bf16x3, interior tiles only)", gemm_work(196608))
    loop("next to plain 196357 x 512 x 512 GEMMs (bf16x3, ragged rows)", gemm_work(196357))
    loop("next to plain 196608 x 512 x 39 GEMMs (bf16x3, K tail)", gemm_work(196608, 512, 40))
    mlp_engine.set_gemm_mode("f32")
    loop("next to plain 196608 x 512 x 512 GEMMs (f32)", gemm_work(196608))
    sys.exit(0)
loop("alone")
mlp_engine.set_gemm_mode("f32")
loop("next to the sdf forward over the template (f32)", sdf_forward)
mlp_engine.set_gemm_mode("bf16x3")
loop("next to the sdf forward over the template (bf16x3)", sdf_forward)
loop("allocator cache emptied before every call", None, torch.cuda.empty_cache)
loop("bf16x3 forward + cache emptied", sdf_forward, torch.cuda.empty_cache)
