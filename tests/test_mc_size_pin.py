"""Marching cubes pinned to the reference's OWN kernels AT THE SIZES BASELINE.json names: the shipped coarse grid 225 x 321 x 129
(train.py:29-35; `mc_gpu` receives the volume as [W, H, D]) and 513^3 (train.py:55-61, configs[3]).  The host build of
MCGpu/CudaKernels.cu (oracle/_ref/libmc_ref_fma.so, contracted like nvcc's default) is run once in the build container by
oracle/gen_mc_size_golden.py on a volume both sides construct from integers (below); its canonicalised output -- vertices in
lattice-edge-key order, face rows sorted -- is frozen as counts + SHA-256 digests in tests/golden/mc_size.npz (the meshes themselves
are 10-60 MB).  The HIP kernels must reproduce the digests: every vertex coordinate and every face index bit for bit.
(tests/test_mc_reference_pin.py holds the same comparison array-by-array up to 160 x 96 x 128.)"""
import hashlib
import os
import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mc_size.npz")
SIZES = {"coarse_225x321x129": (225, 321, 129), "cube_513": (513, 513, 513)}
STEP, ORG = (1.6 / 224, 2.2 / 320, 0.8 / 128), (-0.8, -1.25, -0.4)          # the coarse stage's spacing / origin (LBS box over the 224 x 320 x 128 cells)


def _splitmix(idx, seed):
    with np.errstate(over="ignore"):
        h = idx + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        h ^= h >> np.uint64(30); h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(27); h *= np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(31)
    return h


def volume(shape, seed=11):
    """float32 [NX,NY,NZ]: an ellipsoid's signed distance-like field plus voxel-scale roughness (0.6 of a cell: many of the 256 cases
    occur, the surface stays a thin shell so that the reference's unchecked 5 %-of-cells scratch holds it).  Built slab by slab from
    float64 arithmetic on integers and a splitmix64 hash -- a pure function of (shape, seed) on every machine."""
    NX, NY, NZ = shape
    out = np.empty(shape, np.float32)
    y = (np.arange(NY, dtype=np.float64) / (NY - 1) * 2.0 - 1.0)[:, None]
    z = (np.arange(NZ, dtype=np.float64) / (NZ - 1) * 2.0 - 1.0)[None, :]
    cell = 2.0 / (max(shape) - 1)
    yz = 0.8 * y * y + 1.1 * z * z
    plane = np.arange(NY * NZ, dtype=np.uint64).reshape(NY, NZ)
    for i in range(NX):
        x = i / (NX - 1) * 2.0 - 1.0
        u = (_splitmix(plane + np.uint64(i * NY * NZ), seed) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
        out[i] = (np.sqrt(x * x + yz) - 0.63 + (u * 2.0 - 1.0) * (0.6 * cell)).astype(np.float32)
    return out


def digest(verts, faces):
    """(V, F, sha256 of the float32 vertex bytes, sha256 of the int64 face bytes) of a CANONICAL mesh (vertices in key order, face rows
    sorted lexicographically)."""
    v = np.ascontiguousarray(verts, dtype=np.float32)
    f = np.ascontiguousarray(faces, dtype=np.int64)
    return v.shape[0], f.shape[0], hashlib.sha256(v.tobytes()).hexdigest(), hashlib.sha256(f.tobytes()).hexdigest()


def test_fixture_is_complete():
    g = np.load(GOLDEN)
    for name, shape in SIZES.items():
        assert tuple(g[name + "_shape"]) == shape and int(g[name + "_V"]) > 100_000 and int(g[name + "_F"]) > 200_000
        assert len(str(g[name + "_sha_v"])) == 64 and len(str(g[name + "_sha_f"])) == 64


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SIZES))
def test_hip_marching_cubes_reproduces_the_reference_digests(name):
    import torch
    from selfreconcode_amd.ext import MCGpu
    g = np.load(GOLDEN)
    s = torch.from_numpy(volume(SIZES[name])).to("cuda:0")
    verts, faces = MCGpu.mc_gpu(s, *STEP, *ORG, 0.0)
    v, f = verts.cpu().numpy(), faces.cpu().numpy()               # the HIP output is already in lattice-edge-key order
    f = f[np.lexsort((f[:, 2], f[:, 1], f[:, 0]))]
    V, F, sv, sf = digest(v, f)
    assert (V, F) == (int(g[name + "_V"]), int(g[name + "_F"]))
    assert sf == str(g[name + "_sha_f"]), "face indices differ from the reference kernels' output"
    assert sv == str(g[name + "_sha_v"]), "vertex coordinates differ from the reference kernels' output"
    # a strided sample of the coordinates is stored as well, so that a digest mismatch can be localised
    assert np.array_equal(v[::997][:len(g[name + "_v_sample"])], g[name + "_v_sample"])
