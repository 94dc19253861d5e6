"""TEST INFRASTRUCTURE ONLY -- the reference's SDF pre-fit loop `OptimNetwork.initializeTmpSDF` (model/network.py:207-290) run verbatim on
CPU for 3 epochs (6 Adam steps: 6938 template points in batches of 5000, lr 0.005, StepLR(500, 0.5), with normals), frozen into
tests/golden/prefit.npz:      python oracle/gen_prefit_golden.py       (build container only: needs /root/reference)

The draws of the loop (torch.randperm per epoch; torch.randn_like / torch.rand of utils.sample_points per batch) are replaced by
deterministic ones keyed by call order, so the product can be fed the same numbers through `initializeTmpSDF(..., rand=...)`.
Stored: the losses of every step (total, manifold, eikonal, normals) and, for every SDF parameter after the 6 steps, its L2 norm, two
fixed random projections and a strided slice."""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle.ref_harness import load_reference  # noqa: E402
from oracle import fixtures as fx  # noqa: E402

ref = load_reference()
EPOCHS, SEED0 = 3, 12000
PROJ_SEEDS = (7001, 7002)


def body():
    """6938 points and unit normals: a cube-sphere blown up to an ellipsoid (stand-in for the 6890 SMPL template vertices)."""
    dirs, _ = fx.cube_sphere(34)
    radii = torch.tensor([0.45, 0.75, 0.3])
    vs = dirs * radii
    ns = torch.nn.functional.normalize(dirs / radii, dim=1)
    return vs.contiguous(), ns.contiguous()


def det_perm(n, seed):
    return torch.argsort(fx.det_tensor((n,), seed, 1.0), stable=True)


def digest(p, seed):
    g = p.detach().double().reshape(-1)
    return np.array([float(g.norm())] + [float(g @ fx.det_tensor((g.numel(),), s + seed, 1.0, torch.float64)) for s in PROJ_SEEDS])


def slice_of(t):
    return t[::29, ::7] if (t.dim() == 2 and t.shape[1] > 1) else t.reshape(-1)[::5]


def main():
    torch.set_num_threads(os.cpu_count())
    sdf = ref.network.getTmpSdf("cpu", 6, 0.6, 256)
    sdf.load_state_dict(fx.sphere_sdf_params(7), strict=True)
    net = object.__new__(ref.network.OptimNetwork)
    torch.nn.Module.__init__(net)
    net.sdf = sdf
    net.tmpBodyVs, net.tmpBodyNs = body()
    calls, steps = [], []
    real = (torch.randperm, torch.rand, torch.randn_like, torch.save)

    def randperm(n, **k):
        calls.append(('perm', n)); return det_perm(n, SEED0 + len(calls) - 1)

    def rand(*size, **k):
        shape = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else tuple(size)
        calls.append(('rand', shape)); return fx.det_tensor(shape, SEED0 + len(calls) - 1, 0.5) + 0.5

    def randn_like(x, **k):
        calls.append(('randn_like', tuple(x.shape))); return fx.det_normal(tuple(x.shape), SEED0 + len(calls) - 1)
    # per-step losses: the reference only prints the last batch of an epoch -- wrap the optimizer step to read them off the graph
    real_step = torch.optim.Adam.step
    torch.randperm, torch.rand, torch.randn_like, torch.save = randperm, rand, randn_like, (lambda *a, **k: None)
    real_abs = torch.Tensor.abs
    try:
        import builtins
        log = []
        real_print = builtins.print
        builtins.print = lambda *a, **k: log.append(" ".join(str(x) for x in a))
        net.initializeTmpSDF(EPOCHS, "unused.pth", with_normals=True)
    finally:
        builtins.print = real_print
        torch.randperm, torch.rand, torch.randn_like, torch.save = real
    # the reference prints (loss, manifold, grad, normals) of the LAST batch of each epoch with 6 decimals
    printed = []
    for line in log:
        if line.startswith("Train Epoch"):
            nums = [float(x.split(":")[1]) for x in line.replace("\t", " ").split("  ") if ":" in x] if False else None
            import re
            printed.append([float(x) for x in re.findall(r":\s*(-?\d+\.\d+)", line)][-4:])
    kinds = [c[0] for c in calls]
    assert kinds == ['perm', 'randn_like', 'rand', 'randn_like', 'rand'] * EPOCHS, kinds
    out = dict(epochs=np.array(EPOCHS), seed0=np.array(SEED0), printed=np.array(printed), call_shapes=np.array([list(np.atleast_1d(c[1])) + [0] * (2 - np.atleast_1d(c[1]).size) for c in calls]))
    for i, (name, p) in enumerate(sdf.named_parameters()):
        out["d_" + name] = digest(p, 100 * i)
        out["s_" + name] = slice_of(p.detach()).numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "prefit.npz"), **out)
    print("wrote prefit.npz;", "printed per epoch (loss, manifold, grad, normals):", printed)


if __name__ == "__main__":
    main()
