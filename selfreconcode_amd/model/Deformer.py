"""Non-rigid deformation field -- drop-in for model/Deformer.py (CompositeDeformer :10-20,
MLPTranslator :22-76, LBSkinner :86-233)."""
import os

import numpy as np
import torch
import torch.nn as nn

from .Embedder import embed_rows
from ..mlp_engine import MLPSpec, mlp_apply, pad_cols, pad4, pack_linear
from ..utils.utils import resolve_band_weights


BATCH_FRAMES = True             # hoisted path: one batch over all frames instead of one MLP pass per frame
HOIST_FRAME_CODE = True     # frame-major batches: per-frame code product out of the deformer's first-layer GEMM (MLPTranslator.hoisted_first_layer)


_EYE3 = {}


def _eye3(device):
    e = _EYE3.get(device)
    if e is None:
        # created once, read from every stream afterwards (main, refiner, selection): filled on the host and copied synchronously, so
        # no stream can see the buffer before its contents
        e = _EYE3[device] = torch.eye(3).to(device)
        if e.is_cuda:
            torch.cuda.current_stream(device).synchronize()
    return e


class CompositeDeformer(nn.Module):
    def __init__(self, deformers):
        super().__init__()
        self.N = len(deformers)
        self.defs = nn.ModuleList(deformers)

    def forward(self, ps, conds, batch_inds=None, **kwargs):
        assert self.N == len(conds)
        out = ps
        for cond, deformer in zip(conds, self.defs):
            out = deformer(out, cond, batch_inds, **kwargs)
        return out


class MLPTranslator(nn.Module):
    """PE_6(p) (+) per-frame code -> 5 linears (ReLU) -> offset; returns p + offset and keeps
    `self.offset`.  The code columns ride in the fused embed kernel (gathered by batch index)."""

    def __init__(self, feature_vector_size, multires, weight_norm=False):
        super().__init__()
        assert multires > 0
        dims = [3 + 6 * multires + feature_vector_size, 512, 512, 512, 512, 3]
        self.feature_vector_size = feature_vector_size
        self.multires = multires
        self.num_layers = len(dims)
        for l in range(0, self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            if weight_norm:
                print('MLPTranslator:weight norm can influence weight initialization, can not produce small weights as '
                      'initialization. Now do not use weight_norm')
            if l == self.num_layers - 2:            # zero-translation start (Deformer.py:44-46)
                torch.nn.init.normal_(lin.weight, mean=0., std=0.001)
                torch.nn.init.constant_(lin.bias, 0.)
            setattr(self, "lin" + str(l), lin)
        self.spec = MLPSpec.relu_mlp(dims[0], dims[1:])
        self.offset = None

    def packed_weights(self):
        from ..mlp_engine import packed_weights_of
        return packed_weights_of(self, len(self.spec.layers))

    def hoisted_first_layer(self, conds, frames=None):
        """The per-frame code enters the first layer only through W0[:, PE:] code_f -- a constant per frame.  For batches laid out
        frame by frame ([N, V, 3]) that product is taken out of the per-point GEMM: the layer becomes [512 x 39] on PE(p) with a
        per-frame bias B_f = W0[:, 39:] code_f + b0 (-15 % of the deformer's FLOPs, K = 167 -> 39 in the first-layer
        forward, backward-data and weight-gradient GEMMs).  Plain torch ops: autograd carries the gradients of W0, the codes and b0."""
        npe = 3 + 6 * self.multires
        W0 = self.lin0.weight
        Bf = conds.reshape(-1, self.feature_vector_size) @ W0[:, npe:].t() + self.lin0.bias            # [N, 512]
        if frames is not None and Bf.shape[0] != frames:
            # (one bias row per row segment is what mlp_engine._bias_segments assumes; a code tensor with another number of rows would
            # still divide the batch -- and silently pair points with the wrong frame's code.  The reference's cat of the points with the expanded codes, Deformer.py:61, raises here too.)
            raise ValueError(f"MLPTranslator: {Bf.shape[0]} condition codes for a batch of {frames} frames")
        W0p = pad_cols(W0[:, :npe], pad4(npe))                                                          # [512, pad4(39)] (a copy: never a deferred sink)
        spec = getattr(self, "_spec_pe", None)
        if spec is None:
            spec = self._spec_pe = MLPSpec.relu_mlp(npe, [L.N for L in self.spec.layers])
        return spec, W0p, Bf

    def forward(self, ps, conds, batch_inds=None, **kwargs):
        ratio = kwargs['ratio']['deformerRatio']
        ws = resolve_band_weights(self.multires, ratio)
        if batch_inds is None and HOIST_FRAME_CODE and ps.dim() == 3:
            spec, W0p, Bf = self.hoisted_first_layer(conds, ps.shape[0])
            Ws, bs = self.packed_weights()
            if BATCH_FRAMES:
                # all frames in ONE batch: the first layer runs per frame (its bias is the frame's), every other layer once over
                # N x V rows -- an 85k-row launch is 10.4 tiles of 128 x 128 per CU (the last, partly filled round costs ~10 % of it),
                # three times that is 31 per CU; a third of the launches, slab reductions and autograd nodes
                A0 = embed_rows(ps.reshape(-1, 3), self.multires, ws)
                self.offset = mlp_apply(spec, A0, [W0p] + Ws[1:], [Bf] + bs[1:]).view(ps.shape[0], ps.shape[1], 3)
                return ps[..., :3] + self.offset
            outs = []
            for p_f, B_f in zip(ps.unbind(0), Bf.unbind(0)):                   # unbind: ONE backward node (a stack) instead of a zero-fill + copy per frame
                A0 = embed_rows(p_f, self.multires, ws)
                outs.append(mlp_apply(spec, A0, [W0p] + Ws[1:], [B_f] + bs[1:]))
            self.offset = torch.stack(outs, 0)
            return ps[..., :3] + self.offset
        if batch_inds is not None:
            flat, index, seg = ps, batch_inds, 0
        else:                                        # [N, V, 3] with one code per frame
            nb, nv = ps.shape[0], ps.shape[1]
            flat = ps.reshape(-1, 3)
            index = torch.arange(nb, device=ps.device).repeat_interleave(nv)
            seg = nv
        A0 = embed_rows(flat, self.multires, ws, extra=conds.reshape(-1, self.feature_vector_size), extra_index=index, segment=seg)
        Ws, bs = self.packed_weights()
        x = mlp_apply(self.spec, A0, Ws, bs)
        if batch_inds is not None:
            self.offset = x
            return ps[..., :3] + x
        self.offset = x.view(ps.shape[0], ps.shape[1], 3)
        return ps[..., :3] + self.offset


def getTranslatorNet(device, conf):
    return MLPTranslator(conf.get_int('condlen'), multires=conf.get_int('multires')).to(device)


# ------------------------------------------------------------------------------------------------
# SMPL linear-blend skinning on a sampled weight volume (model/Deformer.py:86-233)
import ctypes  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from .. import _lib  # noqa: E402
from ..MCAcc.grid_sampler_mine import GridSamplerMine3dFunction  # noqa: E402


def batch_rodrigues(theta):
    """axis-angle [M,3] -> R [M,3,3] through the half-angle quaternion, with the reference's
    `+1e-8` inside the norm only (smpl_pytorch/util.py:35-78)."""
    angle = torch.norm(theta + 1e-8, p=2, dim=1, keepdim=True)
    axis = theta / angle
    half = angle * 0.5
    q = torch.cat([torch.cos(half), torch.sin(half) * axis], dim=1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    return torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                        2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                        2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1).view(-1, 3, 3)


def _tree_levels(parents):
    depth = [0] * len(parents)
    for i in range(1, len(parents)):
        depth[i] = depth[int(parents[i])] + 1
    return [[i for i in range(len(parents)) if depth[i] == d] for d in range(1, max(depth) + 1)]


class LBSkinner(nn.Module):
    """Same constructor / buffers / call signature as the reference class.  `ws` keeps the reference's
    logical shape [1,24,D,H,W] but lives in channels_last_3d memory so that the kernels read a corner's
    24 weights as one 96-byte run."""

    def __init__(self, ws, bmins, bmaxs, Js, parents, init_pose=None, align_corners=False):
        super().__init__()

        def as_row(v):
            if isinstance(v, list):
                return torch.tensor(v, dtype=torch.float).view(1, 3)
            if isinstance(v, np.ndarray):
                return torch.from_numpy(v.astype(np.float32)).view(1, 3)
            return v.view(1, 3)
        self.register_buffer('b_min', as_row(bmins))
        self.register_buffer('b_max', as_row(bmaxs))
        ws = torch.from_numpy(ws.astype(np.float32)) if isinstance(ws, np.ndarray) else ws.to(torch.float)
        assert ws.dim() == 5 and ws.shape[0] == 1 and ws.shape[1] == 24
        self.register_buffer('ws', ws.contiguous(memory_format=torch.channels_last_3d))
        assert align_corners is False
        self.align_corners = align_corners
        self.register_buffer('Js', Js.view(24, 3).float())
        self.parents = [int(p) for p in parents]
        self._levels = _tree_levels(self.parents)
        if init_pose is None:
            raise NotImplementedError("the reference always builds the skinner with an init pose (network.py:843,851)")
        if isinstance(init_pose, np.ndarray):
            init_pose = torch.from_numpy(init_pose.astype(np.float32))
        if init_pose.numel() == 24 * 3:
            self.init_pose_inverse(batch_rodrigues(init_pose.view(-1, 3)).view(24, 3, 3), self.Js)
        else:
            self.register_buffer('init_pose', init_pose.view(24, 4, 4))

    def _apply(self, fn, *a, **k):        # keep the channel-last layout across .to(device)
        out = super()._apply(fn, *a, **k)
        if not self.ws.is_contiguous(memory_format=torch.channels_last_3d):
            self.ws = self.ws.contiguous(memory_format=torch.channels_last_3d)
        return out

    def init_pose_inverse(self, init_pose, Js):
        """Inverse of the rest-pose chain per joint (Deformer.py:125-141)."""
        R, T = [init_pose[0]], [Js[0]]
        for i in range(1, 24):
            pa = self.parents[i]
            R.append(R[pa] @ init_pose[i])
            T.append(R[pa] @ (Js[i] - Js[pa]) + T[pa])
        inv = torch.zeros(24, 4, 4)
        for i in range(24):
            inv[i, :3, :3] = R[i].t()
            inv[i, :3, 3] = -(R[i].t() @ T[i])
            inv[i, 3, 3] = 1.
        self.register_buffer('init_pose', inv)

    def _chain(self, poses):
        """Posed chain G_i = G_parent [R_i | J_i - J_parent] for a batch of frames -> [B,24,4,4];
        evaluated level by level of the kinematic tree (9 batched products instead of 23)."""
        B = poses.shape[0]
        R = batch_rodrigues(poses.reshape(-1, 3)).view(B, 24, 3, 3)
        rel = self.Js.clone()
        rel[1:] = self.Js[1:] - self.Js[self.parents[1:]]
        local = torch.cat([torch.cat([R, rel.view(1, 24, 3, 1).expand(B, 24, 3, 1)], 3),
                           torch.cat([rel.new_zeros(3), rel.new_ones(1)]).view(1, 1, 1, 4).expand(B, 24, 1, 4)], 2)
        G = [None] * 24
        G[0] = local[:, 0]
        for level in self._levels:
            pa = torch.stack([G[self.parents[i]] for i in level], 1)
            prod = pa @ local[:, level]
            for n, i in enumerate(level):
                G[i] = prod[:, n]
        return torch.stack(G, 1)

    def _buffers_key(self):
        # host copies of the buffers are keyed on (storage, version): load_state_dict / .to() / in-place edits refresh them
        return tuple((t.data_ptr(), t._version) for t in (self.Js, self.init_pose, self.b_min, self.b_max))

    def _host_consts(self):
        key = self._buffers_key()
        if getattr(self, "_hc_key", None) != key:
            js = (ctypes.c_float * 72)(*self.Js.detach().cpu().view(-1).tolist())
            pa = (ctypes.c_int32 * 24)(*[max(p, 0) if i else 0 for i, p in enumerate(self.parents)])
            ip = (ctypes.c_float * 384)(*self.init_pose.detach().cpu().view(-1).tolist())
            self._hc = (js, pa, ip)
            self._box = (self.b_min.view(-1).tolist(), self.b_max.view(-1).tolist())
            self._hc_key = key
        return self._hc

    def _box_consts(self):
        self._host_consts()
        return self._box

    def posed_chain(self, poses):
        """(G, A): posed kinematic chain and A = G @ init_pose, one fused launch each way (first-order autograd)."""
        if poses.is_cuda:
            return _PosedChain.apply(self, poses)
        G = self._chain(poses)
        return G, G @ self.init_pose.view(1, 24, 4, 4)

    def posed_transforms(self, poses):
        return self.posed_chain(poses)[1]

    def posedSkeleton(self, conds):
        poses, trans = conds
        assert poses.shape[0] == trans.shape[0]
        return self.posed_chain(poses)[0][:, :, :3, 3]

    def fused(self, ps, A, trans, batch_inds=None, with_jac=False, tps=None):
        """No-autograd fused kernel: y (and dy/dp) for flat points [P,3] (batch_inds) or [N,V,3]."""
        flat = ps.reshape(-1, 3).contiguous()
        P = flat.shape[0]
        a = _lib.SrLbsArgs()
        A12 = A[:, :, :3, :].contiguous()
        tr = trans.contiguous()
        y = torch.empty_like(flat)
        jac = torch.empty((P, 3, 3), device=flat.device) if with_jac else None
        tp = None if tps is None else tps.reshape(-1, 3).contiguous()
        vol = self.ws.permute(0, 2, 3, 4, 1)
        assert vol.is_contiguous()
        a.p, a.tp, a.P = _lib.ptr(flat), _lib.ptr(tp), P
        a.A, a.trans, a.nframes = _lib.ptr(A12), _lib.ptr(tr), A.shape[0]
        a.batch_inds = _lib.ptr(batch_inds)
        a.points_per_frame = 0 if batch_inds is not None else (ps.shape[1] if ps.dim() == 3 else P)
        a.vol, a.D, a.H, a.W = _lib.ptr(vol), vol.shape[1], vol.shape[2], vol.shape[3]
        box = self._box_consts()
        for i in range(3):
            a.bmin[i], a.bmax[i] = box[0][i], box[1][i]
        a.y, a.jac = _lib.ptr(y), _lib.ptr(jac)
        with _lib.on_device(flat.device):
            _lib.call("sr_lbs_fwd", ctypes.byref(a), _lib.stream_of(flat))
        return y.view(ps.shape), jac

    def forward(self, ps, conds, batch_inds=None, **kwargs):
        if type(ps) == list:
            tps, ps = ps
        else:
            tps = ps
        poses, trans = conds
        batch_size = poses.shape[0]
        assert batch_size == trans.shape[0]
        A = self.posed_transforms(poses)
        needs_graph = torch.is_grad_enabled() and (ps.requires_grad or tps.requires_grad or poses.requires_grad or trans.requires_grad)
        if not needs_graph:
            y, _ = self.fused(ps, A, trans, batch_inds, False, None if tps is ps else tps)
            return y
        if tps is ps:
            flat = ps.reshape(-1, 3)
            ppf = 0 if batch_inds is not None else ps.shape[1]
            y = _FusedLBS.apply(self, flat, A, trans, batch_inds, ppf)
            return y.view(ps.shape)
        return self._composite(ps, tps, A, trans, batch_inds)

    def _composite(self, ps, tps, A, trans, batch_inds):
        """Differentiable composition (any derivative order): sampler (HIP fwd/bwd/dbwd) -> blend in torch ops."""
        batch_size = A.shape[0]
        nps = 2. * (tps.reshape(-1, 3) - self.b_min) / (self.b_max - self.b_min) - 1.
        w = GridSamplerMine3dFunction.apply(self.ws, nps.reshape(1, 1, 1, -1, 3)).view(24, -1).transpose(0, 1)   # [P,24]
        A12 = A[:, :, :3, :].reshape(batch_size, 24, 12)
        if batch_inds is None:
            nb, pnum, _ = ps.shape
            assert nb == batch_size
            T = torch.matmul(w.view(nb, pnum, 24), A12).view(nb, pnum, 3, 4)
            return (T[..., :3] @ ps.unsqueeze(-1)).squeeze(-1) + T[..., 3] + trans.view(-1, 1, 3)
        p = ps.reshape(-1, 3)
        Tall = (w @ A12.permute(1, 0, 2).reshape(24, batch_size * 12)).view(-1, batch_size, 12)
        T = torch.gather(Tall, 1, batch_inds.view(-1, 1, 1).expand(-1, 1, 12)).view(-1, 3, 4)
        return (T[..., :3] @ p.unsqueeze(-1)).squeeze(-1) + T[..., 3] + trans[batch_inds]

    def fused_backward(self, flat, A, batch_inds, ppf, ybar, need_p, need_A, need_t):
        P = flat.shape[0]
        a = _lib.SrLbsArgs()
        A12 = A[:, :, :3, :].contiguous()
        vol = self.ws.permute(0, 2, 3, 4, 1)
        a.p, a.tp, a.P = _lib.ptr(flat), 0, P
        a.A, a.trans, a.nframes = _lib.ptr(A12), 0, A.shape[0]
        a.batch_inds, a.points_per_frame = _lib.ptr(batch_inds), ppf
        a.vol, a.D, a.H, a.W = _lib.ptr(vol), vol.shape[1], vol.shape[2], vol.shape[3]
        box = self._box_consts()
        for i in range(3):
            a.bmin[i], a.bmax[i] = box[0][i], box[1][i]
        yb = ybar.contiguous().float()
        pbar = torch.empty_like(flat) if need_p else None
        Abar = torch.empty((A.shape[0], 24, 12), device=flat.device) if need_A else None            # written, not accumulated
        tbar = torch.empty((A.shape[0], 3), device=flat.device) if need_t else None
        part = torch.empty((max(int(_lib.raw("sr_lbs_bwd_workspace_floats")(P, A.shape[0])), 1),), device=flat.device)
        with _lib.on_device(flat.device):
            _lib.call("sr_lbs_bwd", ctypes.byref(a), _lib.ptr(yb), _lib.ptr(pbar), _lib.ptr(Abar), _lib.ptr(tbar), _lib.ptr(part), _lib.stream_of(flat))
        return pbar, Abar, tbar


class _PosedChain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, skin, poses):
        p = poses.detach().contiguous().float().view(-1, 24, 3)
        B = p.shape[0]
        G = torch.empty((B, 24, 4, 4), dtype=torch.float32, device=p.device)
        A = torch.empty_like(G)
        js, pa, ip = skin._host_consts()
        with _lib.on_device(p.device):
            _lib.call("sr_lbs_chain_fwd", _lib.ptr(p), B, js, pa, ip, _lib.ptr(G), _lib.ptr(A), _lib.stream_of(p))
        ctx.skin = skin
        ctx.save_for_backward(p)
        ctx.pshape = poses.shape
        ctx.set_materialize_grads(False)
        return G, A

    @staticmethod
    def backward(ctx, Gbar, Abar):
        if Gbar is None and Abar is None:
            return None, None
        (p,) = ctx.saved_tensors
        js, pa, ip = ctx.skin._host_consts()
        out = torch.empty_like(p)
        gb = None if Gbar is None else Gbar.contiguous().float()
        ab = None if Abar is None else Abar.contiguous().float()
        with _lib.on_device(p.device):
            _lib.call("sr_lbs_chain_bwd", _lib.ptr(p), p.shape[0], js, pa, ip, _lib.ptr(ab), _lib.ptr(gb), _lib.ptr(out), _lib.stream_of(p))
        return None, out.view(ctx.pshape)


class _FusedLBS(torch.autograd.Function):
    """y = LBS(p; A, trans) with one fused kernel forward and one backward.  When the backward itself has to be
    differentiated (create_graph=True: Jacobian-based losses) it falls back to the differentiable composition."""

    @staticmethod
    def forward(ctx, skin, flat, A, trans, batch_inds, ppf):
        flat = flat.contiguous()
        y, _ = skin.fused(flat if batch_inds is not None else flat.view(A.shape[0], -1, 3), A, trans, batch_inds, False, None)
        ctx.skin, ctx.ppf = skin, ppf
        ctx.save_for_backward(flat, A, trans, batch_inds)
        return y.reshape(-1, 3)

    @staticmethod
    def backward(ctx, ybar):
        flat, A, trans, batch_inds = ctx.saved_tensors
        skin = ctx.skin
        need_p, need_A, need_t = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        if torch.is_grad_enabled():
            ps = flat if batch_inds is not None else flat.view(A.shape[0], -1, 3)
            y2 = skin._composite(ps, ps, A, trans, batch_inds).reshape(-1, 3)
            ins = [t for t, n in ((flat, need_p), (A, need_A), (trans, need_t)) if n]
            gs = list(torch.autograd.grad(y2, ins, ybar, create_graph=True, allow_unused=True))
            out = [gs.pop(0) if n else None for n in (need_p, need_A, need_t)]
            return None, out[0], out[1], out[2], None, None
        pbar, Abar, tbar = skin.fused_backward(flat, A, batch_inds, ctx.ppf, ybar, need_p, need_A, need_t)
        if Abar is not None:
            Abar = torch.nn.functional.pad(Abar.view(A.shape[0], 24, 3, 4), (0, 0, 0, 1))
        return None, pbar, Abar, tbar, None, None


class _LBSValueJacobian(torch.autograd.Function):
    """(y, J) = (LBS(q), dLBS/dq) from the fused kernel with its analytic Jacobian; first-order differentiable with respect to
    q, the posed transforms A and the translations through sr_lbs_jac_bwd (which carries the mixed second derivatives of the
    trilinear weight lookup)."""

    @staticmethod
    def forward(ctx, skin, q, A, trans, batch_inds, ppf):
        flat = q.reshape(-1, 3).contiguous().float()
        y, J = skin.fused(flat if batch_inds is not None else flat.view(A.shape[0], -1, 3), A, trans, batch_inds, True, None)
        ctx.skin, ctx.ppf = skin, ppf
        ctx.save_for_backward(flat, A, batch_inds)
        ctx.set_materialize_grads(False)
        return y.reshape(-1, 3), J

    @staticmethod
    def backward(ctx, ybar, Jbar):
        flat, A, batch_inds = ctx.saved_tensors
        skin = ctx.skin
        if ybar is None and Jbar is None:
            return (None,) * 6
        need_q, need_A, need_t = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        P = flat.shape[0]
        a = _lib.SrLbsArgs()
        A12 = A[:, :, :3, :].contiguous()
        vol = skin.ws.permute(0, 2, 3, 4, 1)
        a.p, a.tp, a.P = _lib.ptr(flat), 0, P
        a.A, a.trans, a.nframes = _lib.ptr(A12), 0, A.shape[0]
        a.batch_inds, a.points_per_frame = _lib.ptr(batch_inds), ctx.ppf
        a.vol, a.D, a.H, a.W = _lib.ptr(vol), vol.shape[1], vol.shape[2], vol.shape[3]
        box = skin._box_consts()
        for i in range(3):
            a.bmin[i], a.bmax[i] = box[0][i], box[1][i]
        yb = None if ybar is None else ybar.contiguous().float()
        Jb = torch.zeros((P, 3, 3), device=flat.device) if Jbar is None else Jbar.contiguous().float()
        qbar = torch.empty_like(flat) if need_q else None
        Abar = torch.empty((A.shape[0], 24, 12), device=flat.device) if need_A else None            # written, not accumulated
        tbar = torch.empty((A.shape[0], 3), device=flat.device) if (need_t and yb is not None) else None
        part = torch.empty((max(int(_lib.raw("sr_lbs_bwd_workspace_floats")(P, A.shape[0])), 1),), device=flat.device)
        with _lib.on_device(flat.device):
            _lib.call("sr_lbs_jac_bwd", ctypes.byref(a), _lib.ptr(yb), _lib.ptr(Jb), _lib.ptr(qbar), _lib.ptr(Abar), _lib.ptr(tbar), _lib.ptr(part),
                      _lib.stream_of(flat))
        if Abar is not None:
            Abar = torch.nn.functional.pad(Abar.view(A.shape[0], 24, 3, 4), (0, 0, 0, 1))
        return None, qbar, Abar, tbar, None, None


def deformer_value_jacobian(deformer, ps, defconds, batch_inds, ratio):
    """(d, J) = (d(p), dd/dp) of CompositeDeformer([MLPTranslator, LBSkinner]) for flat points with batch indices, by forward mode:
    one group-4 pass of the deformation MLP (value + three seed tangents), the fused LBS kernel with its analytic Jacobian and one
    3x3 product -- instead of a forward plus the three create_graph reverse passes of utils/utils.py:106-120 (compute_Jacobian)
    through both stages.  First-order differentiable in everything the reference's graph reaches (points, MLP weights, per-frame
    codes, poses, translations): what the colour / normal losses (network.py:599-639) and propagateTmpPsGrad (:702-814) need."""
    tr, skin = deformer.defs[0], deformer.defs[1]
    d_cond, (poses, trans) = defconds[0], defconds[1]
    q, Jq = translator_value_jacobian(tr, ps, d_cond, batch_inds, ratio)
    A = skin.posed_transforms(poses)
    y, Jl = _LBSValueJacobian.apply(skin, q, A, trans, batch_inds, 0)
    from ..utils.utils import small_matmul
    return y, small_matmul(Jl, Jq)


def is_fused_composite(deformer):
    return (isinstance(deformer, CompositeDeformer) and deformer.N == 2 and isinstance(deformer.defs[0], MLPTranslator)
            and isinstance(deformer.defs[1], LBSkinner))


class TranslatorValueJacobian(torch.autograd.Function):
    """(d, J) = (p + offset(p), I + d offset / d p) of the deformation MLP by FORWARD mode: one group-4 pass of
    the layer kernels (primal + 3 seed tangents per point) instead of a forward plus three reverse passes
    (utils/utils.py:106-120), and ONE group-4 reverse sweep as its backward instead of reverse-over-reverse.
    First-order differentiable (what the deformation regulariser needs, network.py:565-582)."""

    @staticmethod
    def forward(ctx, tr, ratio, x, conds, index, segment, spec, *wb):
        """`spec` None: the translator's own layer table with the per-frame code as input columns (`conds`, `index`);
        otherwise a table whose first layer takes PE(p) only (conds None; the code product sits in the first bias, see
        MLPTranslator.hoisted_first_layer)."""
        from .. import mlp_engine as me
        from .Embedder import band_weight_tensor
        spec = spec or tr.spec
        nl = len(spec.layers)
        Ws, bs = list(wb[:nl]), list(wb[nl:])
        flat = x.reshape(-1, 3).contiguous().float()
        P = flat.shape[0]
        wt, _ = band_weight_tensor(resolve_band_weights(tr.multires, ratio), tr.multires, flat.device)
        E = 0 if conds is None else tr.feature_vector_size
        ldo = me.pad4(3 + 6 * tr.multires + E)
        A0 = torch.empty((P * 4, ldo), dtype=torch.float32, device=flat.device)
        cd = None if conds is None else conds.reshape(-1, E).contiguous().float()
        with _lib.on_device(flat.device):
            _lib.call("sr_pe_embed", _lib.ptr(flat), P, tr.multires, _lib.ptr(wt), _lib.ptr(cd), 0 if cd is None else cd.stride(0), E,
                      _lib.ptr(index), 4, _lib.ptr(A0), ldo, _lib.stream_of(flat))
            acts = me.forward(spec, A0, Ws, bs, 4)
        out = acts[-1].view(P, 4, -1)[:, :, :3]
        d = flat + out[:, 0]
        J = out[:, 1:4].transpose(1, 2) + _eye3(flat.device)         # J[p, r, c] = delta + d off_r / d x_c
        ctx.tr, ctx.wt, ctx.segment, ctx.n_extra, ctx.xshape, ctx.spec, ctx.E = tr, wt, segment, 0 if cd is None else cd.shape[0], x.shape, spec, E
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(flat, index, A0, *wb, *acts[:-1])
        return d.view(x.shape), J

    @staticmethod
    def backward(ctx, dbar, Jbar):
        from .. import mlp_engine as me
        tr, spec = ctx.tr, ctx.spec
        nl = len(spec.layers)
        saved = ctx.saved_tensors
        flat, index, A0 = saved[0], saved[1], saved[2]
        wb, acts = saved[3:3 + 2 * nl], list(saved[3 + 2 * nl:])
        Ws = list(wb[:nl])
        P = flat.shape[0]
        if dbar is None and Jbar is None:
            return (None,) * (7 + 2 * nl)
        ybar = torch.zeros((P, 4, 4), dtype=torch.float32, device=flat.device)
        if dbar is not None:
            ybar[:, 0, :3] = dbar.reshape(-1, 3)
        if Jbar is not None:
            ybar[:, 1:4, :3] = Jbar.transpose(1, 2)
        WTs = [me.transposed_of(Ws[l], spec.layers[l].K) for l in range(nl)]
        acts_full = acts + [None]
        need_par = any(ctx.needs_input_grad[7:])
        # the input cotangent (first-layer backward-data GEMM + the encoding's backward) only when somebody asks for it: the points
        # or, through the code columns of the first layer's input, the per-frame codes (the regulariser's sample points need neither)
        need_x = ctx.needs_input_grad[2]
        need_in = need_x or (ctx.needs_input_grad[3] and ctx.E > 0)
        A0bar, dWs, dbs = me.reverse(spec, A0, WTs, acts_full, ybar.view(P * 4, 4), 4, need_in, need_par, Ws, list(wb[nl:]))
        if not need_par:
            dWs, dbs = [None] * nl, [None] * nl
        xbar = None
        if need_x:
            xbar = torch.empty_like(flat)
            with _lib.on_device(flat.device):
                _lib.call("sr_pe_embed_bwd", _lib.ptr(flat), P, tr.multires, _lib.ptr(ctx.wt), 4, _lib.ptr(A0bar), A0bar.stride(0),
                          _lib.ptr(xbar), _lib.stream_of(flat))
            if dbar is not None:
                xbar = xbar + dbar.reshape(-1, 3)
            xbar = xbar.view(ctx.xshape)
        gcond = None
        if ctx.needs_input_grad[3] and ctx.E:
            E = ctx.E
            ge = A0bar.view(P, 4, -1)[:, 0, 3 + 6 * tr.multires:3 + 6 * tr.multires + E]
            if ctx.segment:
                gcond = ge.reshape(ctx.n_extra, ctx.segment, E).sum(1)
            else:
                gcond = (me.rows_frame_sum(ge, index, ctx.n_extra) if ctx.n_extra <= 32       # deterministic (index_add: float atomics)
                         else torch.zeros((ctx.n_extra, E), dtype=ge.dtype, device=ge.device).index_add(0, index, ge))
        return (None, None, xbar, gcond, None, None, None) + tuple(dWs) + tuple(dbs)


def translator_value_jacobian(tr, ps, conds, batch_inds, ratio):
    """ps [P,3] with batch_inds, or [N,V,3] (one code per frame).  Returns d (like ps) and J [P,3,3]."""
    r = ratio['deformerRatio'] if isinstance(ratio, dict) else ratio
    if batch_inds is not None:
        index, seg = batch_inds, 0
    else:
        index = torch.arange(ps.shape[0], device=ps.device).repeat_interleave(ps.shape[1])
        seg = ps.shape[1]
    Ws, bs = tr.packed_weights()
    if batch_inds is None and HOIST_FRAME_CODE and ps.dim() == 3:            # frame-major batch: code product as a per-frame bias
        spec, W0p, Bf = tr.hoisted_first_layer(conds, ps.shape[0])
        if BATCH_FRAMES:                                                       # one group-4 batch over all frames (see MLPTranslator.forward)
            return TranslatorValueJacobian.apply(tr, r, ps.contiguous(), None, None, 0, spec, W0p, *Ws[1:], Bf, *bs[1:])
        ds_, Js_ = [], []
        for p_f, B_f in zip(ps.unbind(0), Bf.unbind(0)):                       # unbind: ONE backward node (a stack) instead of a zero-fill + copy per frame
            d_f, J_f = TranslatorValueJacobian.apply(tr, r, p_f.contiguous(), None, None, 0, spec, W0p, *Ws[1:], B_f, *bs[1:])
            ds_.append(d_f); Js_.append(J_f)
        return torch.stack(ds_, 0), torch.cat(Js_, 0)
    return TranslatorValueJacobian.apply(tr, r, ps, conds, index, seg, None, *Ws, *bs)
