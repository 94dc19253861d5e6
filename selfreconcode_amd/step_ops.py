"""Autograd faces of csrc/step_ops.hip: the elementwise / small-matrix blocks between the MLP evaluations of one training
step (camera projection and rays, cardinal rays, the colour / normal / eikonal / deformation-regulariser / mask-IoU
reductions, the normal equations of the implicit differentiation), each one launch for the value and one for the gradient.

The composite torch formulations these replace stay in model/optim_network.py, utils/utils.py and model/CameraMine.py (they
are the reference's own lines, cited there); `ENABLED = False` (or SR_FUSED_STEP_OPS=0) switches back to them, which is how
tests/test_step_ops_gpu.py checks every fused op in value and gradient.  First-order only: none of these outputs is
differentiated twice by the step (the second-order path runs through the MLP engine, whose inputs here are plain cotangents).
"""
import ctypes
import os

import torch
from torch.autograd import Function

from . import _lib

ENABLED = os.environ.get("SR_FUSED_STEP_OPS", "1") != "0"
ENABLED_CAMERA = ENABLED          # model/CameraMine.py reads this one (projection / rays), everything else reads ENABLED
MAX_FRAMES = _lib.SR_STEP_MAX_FRAMES
_SLOTS = _lib.SR_STEP_LOSS_SLOTS


def _f32c(t):
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _i64c(t):
    t = t.detach()
    if t.dtype != torch.int64:
        t = t.long()
    return t if t.is_contiguous() else t.contiguous()


def _loss_buffers(rows, dev):
    blocks = _lib.raw("sr_step_reduce_blocks")(int(rows))
    out = torch.empty((_SLOTS,), dtype=torch.float32, device=dev)
    partial = torch.empty((blocks, _SLOTS - 1), dtype=torch.float32, device=dev) if blocks > 1 else None
    return out, partial


def _gscalar(g):
    g = g.detach()
    if g.dtype != torch.float32:
        g = g.float()
    return g.reshape(1) if g.is_contiguous() else g.contiguous().reshape(1)


# ------------------------------------------------------------------------------------------------ camera
class _CamArgs:
    """sr_camera for one (R, T, f, c, W, H): keeps the tensors it points to alive."""

    def __init__(self, R, T, f, c, W, H):
        self.keep = (_f32c(R), None if T is None else _f32c(T), _f32c(f), _f32c(c))
        a = _lib.SrCamera()
        a.R, a.T, a.f, a.c = _lib.ptr(self.keep[0]), _lib.ptr(self.keep[1]), _lib.ptr(self.keep[2]), _lib.ptr(self.keep[3])
        a.W, a.H = float(W), float(H)
        a.one_minus_inv_w, a.one_minus_inv_h = 1. - 1. / float(W), 1. - 1. / float(H)
        self.c = a

    def ref(self):
        return ctypes.byref(self.c)


def _param_grads(ctx_needs, gparams, R, T, f, c):
    gR = gparams[0:9].view(3, 3).to(R.dtype).view(R.shape) if ctx_needs[0] else None
    gT = gparams[9:12].to(T.dtype).view(T.shape) if ctx_needs[1] and T is not None else None
    gf = gparams[12:14].to(f.dtype).view(f.shape) if ctx_needs[2] else None
    gc = gparams[14:16].to(c.dtype).view(c.shape) if ctx_needs[3] else None
    return gR, gT, gf, gc


class ProjectNDC(Function):
    """model/CameraMine.py:44-70,171-262 for one camera: world points [..., 3] -> NDC xy [..., 2], view depth [...]."""

    @staticmethod
    def forward(ctx, ps, R, T, f, c, W, H):
        _lib.require_gpu(ps, R, f, c)
        p = _f32c(ps).view(-1, 3)
        n = p.shape[0]
        cam = _CamArgs(R, T, f, c, W, H)
        xy = torch.empty((n, 2), dtype=torch.float32, device=p.device)
        z = torch.empty((n,), dtype=torch.float32, device=p.device)
        with _lib.on_device(p.device):
            _lib.call("sr_cam_project_ndc_fwd", _lib.ptr(p), n, cam.ref(), _lib.ptr(xy), _lib.ptr(z), _lib.stream_of(p))
        ctx.save_for_backward(p, R, T, f, c)
        ctx.WH, ctx.pshape = (W, H), ps.shape
        ctx.set_materialize_grads(False)
        return xy.view(*ps.shape[:-1], 2), z.view(ps.shape[:-1])

    @staticmethod
    def backward(ctx, gxy, gz):
        p, R, T, f, c = ctx.saved_tensors
        if gxy is None and gz is None:
            return (None,) * 7
        n = p.shape[0]
        cam = _CamArgs(R, T, f, c, *ctx.WH)
        need_p = ctx.needs_input_grad[0]
        need_par = any(ctx.needs_input_grad[1:5])
        gps = torch.empty_like(p) if need_p else None
        gparams = partial = None
        if need_par:
            gparams = torch.empty((16,), dtype=torch.float32, device=p.device)
            partial = torch.empty((max(_lib.raw("sr_step_param_blocks")(n), 1), 16), dtype=torch.float32, device=p.device)
        gxy_c = None if gxy is None else _f32c(gxy).view(-1, 2)
        gz_c = None if gz is None else _f32c(gz).view(-1)
        with _lib.on_device(p.device):
            _lib.call("sr_cam_project_ndc_bwd", _lib.ptr(p), n, cam.ref(), _lib.ptr(gxy_c), _lib.ptr(gz_c), _lib.ptr(gps), _lib.ptr(partial),
                      _lib.ptr(gparams), _lib.stream_of(p))
        gR = gT = gf = gc = None
        if need_par:
            gR, gT, gf, gc = _param_grads(ctx.needs_input_grad[1:5], gparams, R, T, f, c)
        return (None if gps is None else gps.view(ctx.pshape)), gR, gT, gf, gc, None, None


class ViewRays(Function):
    """model/CameraMine.py:129-143: pixels [P,3] = (col, row, 1) -> unit world-space rays."""

    @staticmethod
    def forward(ctx, pixels, R, f, c):
        _lib.require_gpu(pixels, R, f, c)
        px = _f32c(pixels).view(-1, 3)
        n = px.shape[0]
        cam = _CamArgs(R, None, f, c, 2., 2.)
        rays = torch.empty((n, 3), dtype=torch.float32, device=px.device)
        with _lib.on_device(px.device):
            _lib.call("sr_cam_view_rays_fwd", _lib.ptr(px), n, cam.ref(), _lib.ptr(rays), _lib.stream_of(px))
        ctx.save_for_backward(px, R, f, c)
        ctx.set_materialize_grads(False)
        return rays

    @staticmethod
    def backward(ctx, grays):
        px, R, f, c = ctx.saved_tensors
        if grays is None or not any(ctx.needs_input_grad[1:]):
            return None, None, None, None
        n = px.shape[0]
        cam = _CamArgs(R, None, f, c, 2., 2.)
        gparams = torch.empty((16,), dtype=torch.float32, device=px.device)
        partial = torch.empty((max(_lib.raw("sr_step_param_blocks")(n), 1), 16), dtype=torch.float32, device=px.device)
        with _lib.on_device(px.device):
            _lib.call("sr_cam_view_rays_bwd", _lib.ptr(px), n, cam.ref(), _lib.ptr(_f32c(grays)), _lib.ptr(partial), _lib.ptr(gparams),
                      _lib.stream_of(px))
        needs = (ctx.needs_input_grad[1], False, ctx.needs_input_grad[2], ctx.needs_input_grad[3])
        gR, _, gf, gc = _param_grads(needs, gparams, R, None, f, c)
        return None, gR, gf, gc


# ------------------------------------------------------------------------------------------------ cardinal rays / normals
class CardinalRays(Function):
    """utils/utils.py:155-169 after the Jacobian: normalize(J^-1 v), rays.detach() where J is singular."""

    @staticmethod
    def forward(ctx, J, v):
        _lib.require_gpu(J, v)
        Jc, vc = _f32c(J).view(-1, 3, 3), _f32c(v).view(-1, 3)
        n = Jc.shape[0]
        out = torch.empty((n, 3), dtype=torch.float32, device=Jc.device)
        ok = torch.empty((n,), dtype=torch.bool, device=Jc.device)
        with _lib.on_device(Jc.device):
            _lib.call("sr_cardinal_rays_fwd", _lib.ptr(Jc), _lib.ptr(vc), n, _lib.ptr(out), _lib.ptr(ok), _lib.stream_of(Jc))
        ctx.save_for_backward(Jc, vc)
        ctx.mark_non_differentiable(ok)
        ctx.set_materialize_grads(False)
        return out, ok

    @staticmethod
    def backward(ctx, gout, _gok):
        Jc, vc = ctx.saved_tensors
        if gout is None:
            return None, None
        n = Jc.shape[0]
        gJ = torch.empty_like(Jc) if ctx.needs_input_grad[0] else None
        gv = torch.empty_like(vc) if ctx.needs_input_grad[1] else None
        if gJ is None and gv is None:
            return None, None
        with _lib.on_device(Jc.device):
            _lib.call("sr_cardinal_rays_bwd", _lib.ptr(Jc), _lib.ptr(vc), n, _lib.ptr(_f32c(gout)), _lib.ptr(gJ), _lib.ptr(gv), _lib.stream_of(Jc))
        return gJ, gv


def deformed_normals(J, onx):
    """utils/utils.py:132-153 in 'test' phase (no gradient): normalize(J^-T n), J n where J is singular."""
    _lib.require_gpu(J, onx)
    Jc, oc = _f32c(J).view(-1, 3, 3), _f32c(onx).view(-1, 3)
    out = torch.empty_like(oc)
    with _lib.on_device(Jc.device):
        _lib.call("sr_deformed_normals", _lib.ptr(Jc), _lib.ptr(oc), Jc.shape[0], _lib.ptr(out), _lib.stream_of(Jc))
    return out


# ------------------------------------------------------------------------------------------------ loss reductions
def _pixels(b, r, c, N, H, W):
    keep = (_i64c(b), _i64c(r), _i64c(c))
    px = _lib.SrRayPixels()
    px.b, px.r, px.c = _lib.ptr(keep[0]), _lib.ptr(keep[1]), _lib.ptr(keep[2])
    px.P, px.N, px.H, px.W = keep[0].shape[0], N, H, W
    return px, keep


class ColorLoss(Function):
    """model/network.py:611-618: scatter-mean over frames of |gt[b,r,c] - colour|_1, then the mean over frames."""

    @staticmethod
    def forward(ctx, colors, gt, b, r, c):
        _lib.require_gpu(colors, gt)
        col, g = _f32c(colors), _f32c(gt)
        N, H, W = g.shape[0], g.shape[1], g.shape[2]
        px, keep = _pixels(b, r, c, N, H, W)
        out, partial = _loss_buffers(px.P, col.device)
        with _lib.on_device(col.device):
            _lib.call("sr_color_loss_fwd", ctypes.byref(px), _lib.ptr(col), _lib.ptr(g), _lib.ptr(partial), _lib.ptr(out), _lib.stream_of(col))
        ctx.save_for_backward(col, g, out, *keep)
        ctx.dims = (N, H, W)
        return out[0]

    @staticmethod
    def backward(ctx, gloss):
        col, g, out, b, r, c = ctx.saved_tensors
        px, _ = _pixels(b, r, c, *ctx.dims)
        gcol = torch.empty_like(col)
        with _lib.on_device(col.device):
            _lib.call("sr_color_loss_bwd", ctypes.byref(px), _lib.ptr(col), _lib.ptr(g), _lib.ptr(out), _lib.ptr(_gscalar(gloss)), _lib.ptr(gcol),
                      _lib.stream_of(col))
        return gcol, None, None, None, None


class NormalLoss(Function):
    """model/network.py:620-639 (see sr_normal_loss_fwd in include/selfrecon_hip.h)."""

    @staticmethod
    def forward(ctx, nx_raw, J, gt_normals, R, rays, weighted, b, r, c):
        _lib.require_gpu(nx_raw, J, gt_normals, R)
        nx, Jc, g, Rc = _f32c(nx_raw).view(-1, 3), _f32c(J).view(-1, 3, 3), _f32c(gt_normals), _f32c(R).view(3, 3)
        rc = _f32c(rays).view(-1, 3) if weighted else None
        N, H, W = g.shape[0], g.shape[1], g.shape[2]
        px, keep = _pixels(b, r, c, N, H, W)
        out, partial = _loss_buffers(px.P, nx.device)
        with _lib.on_device(nx.device):
            _lib.call("sr_normal_loss_fwd", ctypes.byref(px), _lib.ptr(nx), _lib.ptr(Jc), _lib.ptr(g), _lib.ptr(Rc), _lib.ptr(rc), 1 if weighted else 0,
                      _lib.ptr(partial), _lib.ptr(out), _lib.stream_of(nx))
        ctx.save_for_backward(nx, Jc, g, Rc, out, *keep, *(() if rc is None else (rc,)))
        ctx.dims, ctx.weighted = (N, H, W), bool(weighted)
        return out[0]

    @staticmethod
    def backward(ctx, gloss):
        saved = ctx.saved_tensors
        nx, Jc, g, Rc, out, b, r, c = saved[:8]
        rc = saved[8] if ctx.weighted else None
        px, _ = _pixels(b, r, c, *ctx.dims)
        gnx = torch.empty_like(nx)
        gJ = torch.empty_like(Jc) if ctx.needs_input_grad[1] else None
        with _lib.on_device(nx.device):
            _lib.call("sr_normal_loss_bwd", ctypes.byref(px), _lib.ptr(nx), _lib.ptr(Jc), _lib.ptr(g), _lib.ptr(Rc), _lib.ptr(rc), 1 if ctx.weighted else 0,
                      _lib.ptr(out), _lib.ptr(_gscalar(gloss)), _lib.ptr(gnx), _lib.ptr(gJ), _lib.stream_of(nx))
        return gnx, gJ, None, None, None, None, None, None, None


class EikonalLoss(Function):
    """model/network.py:547-549: ((|g| - 1)^2).mean()."""

    @staticmethod
    def forward(ctx, g):
        _lib.require_gpu(g)
        gc = _f32c(g).view(-1, 3)
        out, partial = _loss_buffers(gc.shape[0], gc.device)
        with _lib.on_device(gc.device):
            _lib.call("sr_eikonal_loss_fwd", _lib.ptr(gc), gc.shape[0], _lib.ptr(partial), _lib.ptr(out), _lib.stream_of(gc))
        ctx.save_for_backward(gc)
        ctx.shape = g.shape
        return out[0]

    @staticmethod
    def backward(ctx, gloss):
        gc, = ctx.saved_tensors
        gg = torch.empty_like(gc)
        with _lib.on_device(gc.device):
            _lib.call("sr_eikonal_loss_bwd", _lib.ptr(gc), gc.shape[0], _lib.ptr(_gscalar(gloss)), _lib.ptr(gg), _lib.stream_of(gc))
        return gg.view(ctx.shape)


class DefReguLoss(Function):
    """model/network.py:565-582: GMRobustError(sum_k log(s_k(J))^2, c, True).mean() with the batched 3x3 SVD on device."""

    @staticmethod
    def forward(ctx, J, c):
        _lib.require_gpu(J)
        A = _f32c(J).view(-1, 3, 3)
        n = A.shape[0]
        U = torch.empty_like(A); V = torch.empty_like(A)
        S = torch.empty((n, 3), dtype=torch.float32, device=A.device)
        out, partial = _loss_buffers(n, A.device)
        with _lib.on_device(A.device):
            st = _lib.stream_of(A)
            _lib.call("sr_svd3x3", _lib.ptr(A), n, _lib.ptr(U), _lib.ptr(S), _lib.ptr(V), st)
            _lib.call("sr_def_regu_loss_fwd", _lib.ptr(S), n, float(c), _lib.ptr(partial), _lib.ptr(out), st)
        ctx.save_for_backward(U, S, V)
        ctx.c, ctx.shape = float(c), J.shape
        return out[0]

    @staticmethod
    def backward(ctx, gloss):
        U, S, V = ctx.saved_tensors
        gJ = torch.empty_like(U)
        with _lib.on_device(U.device):
            _lib.call("sr_def_regu_loss_bwd", _lib.ptr(U), _lib.ptr(S), _lib.ptr(V), U.shape[0], ctx.c, _lib.ptr(_gscalar(gloss)), _lib.ptr(gJ),
                      _lib.stream_of(U))
        return gJ.view(ctx.shape), None


class MaskIoULoss(Function):
    """model/network.py:652-654: (1 - sum(m g) / sum |m + g - m g|) per frame, mean over frames."""

    @staticmethod
    def forward(ctx, masks, gt):
        _lib.require_gpu(masks, gt)
        m, g = _f32c(masks), _f32c(gt)
        N = m.shape[0]
        hw = m.numel() // N
        out, partial = _loss_buffers(m.numel(), m.device)
        with _lib.on_device(m.device):
            _lib.call("sr_mask_iou_loss_fwd", _lib.ptr(m), _lib.ptr(g), N, hw, _lib.ptr(partial), _lib.ptr(out), _lib.stream_of(m))
        ctx.save_for_backward(m, g, out)
        return out[0]

    @staticmethod
    def backward(ctx, gloss):
        m, g, out = ctx.saved_tensors
        N = m.shape[0]
        gm = torch.empty_like(m)
        with _lib.on_device(m.device):
            _lib.call("sr_mask_iou_loss_bwd", _lib.ptr(m), _lib.ptr(g), N, m.numel() // N, _lib.ptr(out), _lib.ptr(_gscalar(gloss)), _lib.ptr(gm),
                      _lib.stream_of(m))
        return gm, None


# ------------------------------------------------------------------------------------------------ implicit differentiation
def implicit_solve(grad_f, J, v, grad_l):
    """model/network.py:702-771 (no gradient): -> cot_f [P], rhs_tail [P,3], temp [P,3], ok [P] bool."""
    _lib.require_gpu(grad_f, J, v, grad_l)
    gf, Jc, vc, gl = _f32c(grad_f).view(-1, 3), _f32c(J).view(-1, 3, 3), _f32c(v).view(-1, 3), _f32c(grad_l).view(-1, 3)
    n = gf.shape[0]
    dev = gf.device
    cot_f = torch.empty((n,), dtype=torch.float32, device=dev)
    tail = torch.empty((n, 3), dtype=torch.float32, device=dev)
    temp = torch.empty((n, 3), dtype=torch.float32, device=dev)
    ok = torch.empty((n,), dtype=torch.bool, device=dev)
    with _lib.on_device(dev):
        _lib.call("sr_implicit_solve", _lib.ptr(gf), _lib.ptr(Jc), _lib.ptr(vc), _lib.ptr(gl), n, _lib.ptr(cot_f), _lib.ptr(tail), _lib.ptr(temp),
                  _lib.ptr(ok), _lib.stream_of(gf))
    return cot_f, tail, temp, ok


def frames_supported(N):
    return ENABLED and 1 <= int(N) <= MAX_FRAMES
