"""Drop-in for the reference's `MCGpu` pybind module (MCGpu/MCGpu.cpp:14-60).

    mc_gpu(sdfs[NX,NY,NZ] f32 GPU contiguous, xstep=1, ystep=1, zstep=1, xmin=0, ymin=0, zmin=0, fTargetValue=0)
        -> [verts[V,3] f32, faces[F,3] int64]
    mc_init(device_id)

Error convention kept from the reference: wrong dtype / non-positive dims return an EMPTY LIST (MCGpu.cpp:41-48); non-GPU /
non-contiguous input raises (the CHECK_INPUT macro, :3-5,31).  The reference also returns [] for a device index outside [0, 8)
(:43-45) -- the size of its fixed array of per-device singletons.  There is no such array here (workspace and outputs are sized
per call, on the caller's stream, and the output order is deterministic), so every device index works: on a node that exposes more
than 8 logical devices (CPX partitions, 16-GPU boxes) the reference's limit would only turn the first remesh into an unpack error.
"""
import torch
from .. import _lib


def mc_init(device_id):
    return None      # nothing to pre-allocate: tables are uploaded on first use


def mc_gpu(sdfs, xstep=1.0, ystep=1.0, zstep=1.0, xmin=0.0, ymin=0.0, zmin=0.0, fTargetValue=0.0):
    if not sdfs.is_cuda:
        raise RuntimeError("sdfs must be a CUDA tensor")
    if not sdfs.is_contiguous():
        raise RuntimeError("sdfs must be contiguous")
    if sdfs.dtype != torch.float32 or sdfs.dim() != 3:
        return []
    nx, ny, nz = sdfs.shape
    if nx <= 0 or ny <= 0 or nz <= 0:
        return []
    dev = sdfs.device
    with _lib.on_device(dev):
        st = _lib.stream_of(sdfs)
        nbytes = _lib.raw("sr_mc_workspace_bytes")(nx, ny, nz)
        ws = torch.empty((nbytes // 4,), dtype=torch.int32, device=dev)
        counts = torch.zeros((2,), dtype=torch.int32, device=dev)
        _lib.call("sr_mc_count", _lib.ptr(sdfs), nx, ny, nz, float(fTargetValue), _lib.ptr(ws), _lib.ptr(counts), st)
        nv, nf = counts.tolist()                       # the one host sync (the reference has the same one)
        verts = torch.empty((nv, 3), dtype=torch.float32, device=dev)
        faces = torch.empty((nf, 3), dtype=torch.int64, device=dev)
        if nv or nf:
            _lib.call("sr_mc_emit", _lib.ptr(sdfs), nx, ny, nz, float(fTargetValue), _lib.ptr(ws), float(xstep), float(ystep),
                      float(zstep), float(xmin), float(ymin), float(zmin), _lib.ptr(verts), _lib.ptr(faces), st)
    return [verts, faces]
