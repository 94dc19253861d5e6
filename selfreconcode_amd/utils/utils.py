"""Math half of the reference's utils/utils.py over the HIP ops."""
import numpy as np
import torch
from torch.autograd import Function

from ..ext.FastMinv import Fast3x3Minv, Fast3x3Minv_backward
from ..mlp_engine import input_grads_only
from .. import step_ops


class FastDiff3x3MinvFunction(Function):
    """utils/utils.py:8-19."""

    @staticmethod
    def forward(ctx, input):
        invs, check = Fast3x3Minv(input.contiguous())
        ctx.save_for_backward(invs, check)
        ctx.mark_non_differentiable(check)
        return invs, check

    @staticmethod
    def backward(ctx, grad_input, grad_check):
        invs, check = ctx.saved_tensors
        return Fast3x3Minv_backward(grad_input.contiguous(), invs), None


def small_matmul(a, b):
    """Batched product of tiny matrices ([..., m, k] x [..., k, n], m, k, n <= 4) as one broadcast multiply and one reduction:
    the vendor batched GEMM (what `@` dispatches to) takes 40-130 us for a few thousand 3x3 products, two elementwise launches
    take ~10.  Same values up to the order of the k-sum; differentiable to any order."""
    return (a.unsqueeze(-1) * b.unsqueeze(-3)).sum(-2)


def small_matvec(a, v):
    """[..., m, k] x [..., k] -> [..., m] (see small_matmul)."""
    return (a * v.unsqueeze(-2)).sum(-1)


def annealing_weights(multires, ratio):
    """utils/utils.py:40-46."""
    alpha = ratio * multires
    out = []
    for ind in range(multires):
        w = (1. - np.cos(np.pi * min(max(alpha - float(ind), 0.), 1.))) / 2.
        out.extend([w, w])
    return out


def resolve_band_weights(multires, ratio):
    """The None / <=0 / annealed switch shared by the three networks (network.py:74-80,
    Deformer.py:51-57, RenderNet.py:56-62).  Returns None for 'all ones'."""
    if ratio is None:
        return None
    if ratio <= 0:
        return [0. for _ in range(multires * 2)]
    return annealing_weights(multires, ratio)


def GMRobustError(x, c, square=False):
    """utils/utils.py:48-52."""
    if square:
        return 2. * x / (c * c) / (x / (c * c) + 4)
    return 2. * x * x / (c * c) / (x * x / (c * c) + 4)


def quat2mat(quat):
    """utils/utils.py:21-38."""
    nq = quat / quat.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = nq[:, 0], nq[:, 1], nq[:, 2], nq[:, 3]
    B = quat.size(0)
    w2, x2, y2, z2 = w.pow(2), x.pow(2), y.pow(2), z.pow(2)
    wx, wy, wz = w * x, w * y, w * z
    xy, xz, yz = x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(B, 3, 3)


def smpl_tmp_Apose(init_pose_type=0):
    """utils/utils.py:56-72."""
    pose = np.zeros((24, 3))
    if init_pose_type == 0:
        a, b = 10., 45.
    elif init_pose_type == 1:
        a, b = 7., 55.
    else:
        assert False
    pose[1] = np.array([0, 0, a / 180. * np.pi])
    pose[2] = np.array([0, 0, -a / 180. * np.pi])
    pose[16] = np.array([0, 0, -b / 180. * np.pi])
    pose[17] = np.array([0, 0, b / 180. * np.pi])
    return pose.astype(np.float32)


def sample_points(pc_input, global_sigma, local_sigma, ratio=6, rand=None):
    """utils/utils.py:74-84.  `rand` (extension, parity tests): object with randn_like(x) / rand(n, dim) that supplies the two draws."""
    sample_size, dim = pc_input.shape
    noise = torch.randn_like(pc_input) if rand is None else rand.randn_like(pc_input)
    sample_local = pc_input + (noise * local_sigma)
    if ratio > 0:
        u = torch.rand(sample_size // ratio, dim, device=pc_input.device) if rand is None else rand.rand(sample_size // ratio, dim)
        sample_global = (u * (global_sigma * 2)) - global_sigma
        return torch.cat([sample_local, sample_global], dim=0)
    return sample_local


SINGULAR_COUNT = {}   # 'rays' / 'normals' -> bool mask [P] of the rows whose deformation Jacobian was INVERTIBLE in the last call: the
                      # reference tests it on the host and prints a warning (utils.py:145-150,162-167: one sync per call); here the
                      # fallback is selected on the device and `(~mask).sum()` is left to whoever wants the count


def compute_Jacobian(ps, ds, retain_graph, create_graph, allow_unused=False):
    """utils/utils.py:106-120 -- three reverse passes (generic drop-in path; the fused step uses the
    forward-mode group-4 kernels instead, see model/Deformer.py)."""
    grad_d_p = []
    grad_outputs = torch.ones_like(ds[..., 0])
    for c in range(3):
        rg = True if c < 2 else retain_graph
        with input_grads_only():
            out = torch.autograd.grad(ds[..., c], ps, grad_outputs, retain_graph=rg, create_graph=create_graph, allow_unused=allow_unused)
        grad_d_p.append(out[0].view(-1, 1, 3))
    return torch.cat(grad_d_p, dim=1)


FORWARD_MODE_JACOBIAN = True    # deformer value + Jacobian by one forward-mode pass (model/Deformer.py::deformer_value_jacobian)


def deformed_points_and_jacobian(deformer, ps, defconds, batch_inds, ratio, differentiable):
    """d(p) and dd/dp (utils.py:106-120 applied to the deformer): the fused forward-mode Function when the deformer is this
    package's MLPTranslator + LBSkinner composite and the points come with batch indices, the generic three reverse passes
    otherwise.  `differentiable`: the caller will back-propagate through the result (the 'train' phase)."""
    from ..model.Deformer import is_fused_composite, deformer_value_jacobian
    if FORWARD_MODE_JACOBIAN and batch_inds is not None and ps.dim() == 2 and is_fused_composite(deformer):
        if differentiable:
            return deformer_value_jacobian(deformer, ps, defconds, batch_inds, ratio)
        with torch.no_grad():
            return deformer_value_jacobian(deformer, ps, defconds, batch_inds, ratio)
    ds = deformer(ps, defconds, batch_inds, ratio=ratio)
    return ds, compute_Jacobian(ps, ds, differentiable, differentiable)


def compute_deformed_normals(sdf, deformer, ps, defconds, batch_inds, ratio, phase, cache=None, onx=None):
    """utils/utils.py:132-153.

    `cache` / `onx` (extensions): the colour + normal branch of one iteration evaluates the deformer Jacobian at the same points
    three times and the SDF gradient twice (network.py:608,610,623,630-631).  A caller may pass the dict filled by
    `compute_cardinal_rays(..., cache=...)` and the SDF gradient it already has; in 'test' phase (results used detached) their
    detached values are reused instead of being recomputed -- same numbers, a third of the work."""
    check = phase in ('train', 'Train')
    if onx is None or check:
        sdfs = sdf(ps, ratio)
        with input_grads_only():
            onx = torch.autograd.grad(sdfs, ps, torch.ones_like(sdfs), retain_graph=check, create_graph=check)[0]
    else:
        onx = onx.detach()
    if cache is not None and 'J' in cache and not check:
        ds, grad_d_p = cache['ds'].detach(), cache['J'].detach()
    else:
        ds, grad_d_p = deformed_points_and_jacobian(deformer, ps, defconds, batch_inds, ratio, check)
    if step_ops.ENABLED and not check and grad_d_p.is_cuda:      # 'test' phase: one kernel, no graph
        return step_ops.deformed_normals(grad_d_p, onx), ds
    grad_d_p_inv, inv_mask = FastDiff3x3MinvFunction.apply(grad_d_p)
    nx = small_matvec(grad_d_p_inv.transpose(-2, -1), onx.view(-1, 3))
    # singular Jacobians fall back to J n (reference :145-150), selected on the device
    SINGULAR_COUNT['normals'] = inv_mask
    nx = torch.where(inv_mask[:, None], nx, small_matvec(grad_d_p, onx.view(-1, 3)))
    nx = nx / nx.norm(dim=1, keepdim=True)
    return nx, ds


def compute_cardinal_rays(deformer, ps, rays, defconds, batch_inds, ratio, phase, cache=None):
    """utils/utils.py:155-169.  `cache` (extension): a dict that receives / supplies the deformed points and their Jacobian
    (keys 'ds', 'J') so that later uses at the same points share them (see compute_deformed_normals)."""
    check = phase in ('train', 'Train')
    if cache is not None and 'J' in cache:
        ds, grad_d_p = cache['ds'], cache['J']
    else:
        ds, grad_d_p = deformed_points_and_jacobian(deformer, ps, defconds, batch_inds, ratio, check)
        if cache is not None:
            cache['ds'], cache['J'] = ds, grad_d_p
    if step_ops.ENABLED and grad_d_p.is_cuda:
        crays, inv_mask = step_ops.CardinalRays.apply(grad_d_p, rays.view(-1, 3))
        SINGULAR_COUNT['rays'] = inv_mask
        return crays, ds
    grad_d_p_inv, inv_mask = FastDiff3x3MinvFunction.apply(grad_d_p)
    crays = small_matvec(grad_d_p_inv, rays.view(-1, 3))
    SINGULAR_COUNT['rays'] = inv_mask                    # (reference :162-167: host test + print)
    crays = torch.where(inv_mask[:, None], crays, rays.detach())
    crays = crays / crays.norm(dim=1, keepdim=True)
    return crays, ds


def compute_netRender_color(net, ps, ds, ns, vs, features, framefeatures, ratio):
    """utils/utils.py:171-172 (framefeatures and ds are ignored by the reference as well)."""
    return net(ps, ns, vs, features, ratio)


def DCTBasis(k, N):
    assert k < N
    basis = torch.tensor([np.pi * (float(n) + 0.5) * k / float(N) for n in range(N)]).float()
    return torch.cos(basis) * (1. / np.sqrt(float(N)) if k == 0 else np.sqrt(2. / float(N)))


def DCTNullSpace(k, N):
    """utils/utils.py:207-208."""
    return torch.stack([DCTBasis(ind, N) for ind in range(k, N)])


def DCTSpace(k, N):
    return torch.stack([DCTBasis(ind, N) for ind in range(0, k)])
