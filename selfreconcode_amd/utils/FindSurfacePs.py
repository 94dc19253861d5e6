"""Ray/surface intersection helpers -- drop-ins for utils/FindSurfacePs.py of the reference.

FindSurfacePs   (:5-29)   rasteriser fragments -> canonical seed points.
OptimizeSurfacePs (:114-163) the masked Newton refiner ("the tracer").  When given this package's
ImplicitNetwork + CompositeDeformer([MLPTranslator, LBSkinner]) it runs the device-driven path
(`_optimize_device_driven`): the queue of unfinished rays is compacted on the GPU after every step and
every row count stays in device memory; per Newton step the host issues a fixed sequence -- first-layer
inputs of both networks, a chain over all forward layers of the sdf-only SDF MLP and the deformation MLP
side by side (one launch per layer pair on the device-side row count), LBS + Jacobian + convergence test + residual cotangents, the
chain over all reverse layers, Newton update + retirement + compaction -- with no autograd graph, no
per-frame Python loop and no host synchronisation.  The convergence test of step k and the gradient
of step k+1 come from the same evaluation (the reference evaluates the same points twice).
`DEVICE_DRIVEN = False` keeps the earlier layer-by-layer host loop (same arithmetic per ray).
"""
import ctypes
import numpy as np
import os
import torch

from .. import _lib
from .. import hostsync
from .. import mlp_engine as me
from .utils import resolve_band_weights


def FindSurfacePs(TmpVs, TmpFaces, frags):
    N, H, W, K = frags.pix_to_face.shape
    pix_to_face, bary_coords = frags.pix_to_face, frags.bary_coords
    inner = (bary_coords > 0.0).all(-1) & (pix_to_face >= 0)
    # first valid fragment per pixel (the reference: torch_scatter.scatter(cols, rows, 'min'))
    ks = torch.arange(K, device=inner.device).view(1, 1, 1, K).expand(N, H, W, K)
    index = torch.where(inner, ks, torch.full_like(ks, K)).amin(dim=-1)
    hit = index < K
    batch_inds, row_inds, col_inds = hostsync.nonzero(hit, as_tuple=True)
    sel = index[hit].view(-1, 1)
    finds = torch.gather(pix_to_face[hit], 1, sel).view(-1) % TmpFaces.shape[0]
    ws = torch.gather(bary_coords[hit], 1, sel.view(-1, 1, 1).expand(-1, 1, 3)).view(-1, 3)
    initTmpPs = (TmpVs[TmpFaces[finds].view(-1)].view(-1, 3, 3) * ws[:, :, None]).sum(1)
    return batch_inds, row_inds, col_inds, initTmpPs, finds


REVERSE_MODE = True     # derivatives of the refiner by reverse sweeps (2x rows) instead of group-4 forward tangents (4x rows)


class _FusedEval:
    """Holds the per-call constants of the fused refiner (weights are packed once per call)."""

    def __init__(self, sdf, deformer, defconds, ratio):
        from ..model.network import ImplicitNetwork
        from ..model.Deformer import MLPTranslator, LBSkinner, CompositeDeformer
        ok = (isinstance(sdf, ImplicitNetwork) and isinstance(deformer, CompositeDeformer) and deformer.N == 2 and
              isinstance(deformer.defs[0], MLPTranslator) and isinstance(deformer.defs[1], LBSkinner))
        if not ok:
            raise TypeError("fused refiner needs ImplicitNetwork + CompositeDeformer([MLPTranslator, LBSkinner])")
        self.sdf, self.tr, self.skin = sdf, deformer.defs[0], deformer.defs[1]
        dev = sdf.lin0.bias.device
        with torch.no_grad():
            r_sdf = ratio if isinstance(ratio, (float, int)) or ratio is None else ratio['sdfRatio']
            from ..model.Embedder import band_weight_tensor
            self.w_sdf, _ = band_weight_tensor(resolve_band_weights(sdf.multires, r_sdf), sdf.multires, dev)
            self.w_def, _ = band_weight_tensor(resolve_band_weights(self.tr.multires, ratio['deformerRatio']), self.tr.multires, dev)
            Ws, bs = sdf.packed_weights()
            self.sdf_spec = me.MLPSpec(sdf.spec.layers[:-1] + [me.LayerSpec(sdf.spec.layers[-1].K, sdf.d_out, me.ACT_NONE)], sdf.spec.K0)
            self.sdf_W = [w.contiguous() for w in Ws[:-1]] + [Ws[-1][:sdf.d_out].contiguous()]     # sdf-only last layer
            self.sdf_b = list(bs[:-1]) + [bs[-1][:sdf.d_out].contiguous()]
            Wd, bd = self.tr.packed_weights()
            self.def_W, self.def_b = [w.contiguous() for w in Wd], list(bd)
            self.sdf_WT = [me.transposed_of(w, L.K) for w, L in zip(self.sdf_W, self.sdf_spec.layers)]
            self.def_WT = [me.transposed_of(w, L.K) for w, L in zip(self.def_W, self.tr.spec.layers)]
            self.conds = defconds[0].detach().contiguous().float()
            poses, trans = defconds[1]
            self.A = self.skin.posed_transforms(poses.detach())
            self.trans = trans.detach().contiguous()

    def unit_cotangent(self, M):
        """[M,4] rows (1,0,0,0): the cotangent that turns the SDF reverse sweep into grad f; built once per call."""
        buf = getattr(self, "_unit", None)
        if buf is None or buf.shape[0] < M:
            buf = torch.zeros((M, 4), dtype=torch.float32, device=self.conds.device)
            buf[:, 0] = 1.0
            self._unit = buf
        return buf[:M]

    def _embed(self, x, L, wt, extra, index, group):
        P = x.shape[0]
        E = 0 if extra is None else extra.shape[1]
        ldo = me.pad4(3 + 6 * L + E)
        out = torch.empty((P * group, ldo), dtype=torch.float32, device=x.device)
        _lib.call("sr_pe_embed", _lib.ptr(x), P, L, _lib.ptr(wt), _lib.ptr(extra), 0 if extra is None else extra.stride(0), E,
                  _lib.ptr(index), group, _lib.ptr(out), ldo, _lib.stream_of(x))
        return out

    def evaluate(self, x, bi, group):
        """-> (sdf rows [M*group, ld], offset rows [M*group, ld], y [M,3], J_lbs [M,3,3] or None)"""
        with _lib.on_device(x.device):
            A0 = self._embed(x, self.sdf.multires, self.w_sdf, None, None, group)
            sdf_rows = me.forward(self.sdf_spec, A0, self.sdf_W, self.sdf_b, group)[-1]
            A0d = self._embed(x, self.tr.multires, self.w_def, self.conds, bi, group)
            off_rows = me.forward(self.tr.spec, A0d, self.def_W, self.def_b, group)[-1]
            q = x + off_rows.view(x.shape[0], group, -1)[:, 0, :3]
            y, jl = self.skin.fused(q, self.A, self.trans, bi, with_jac=(group == 4))
        return sdf_rows, off_rows, y, jl


def _newton_reverse(ev, x, bi, rays, cam, dthr, athr, w1, w2, update):
    """One refiner step with reverse-mode derivatives (2x the rows of a value pass instead of 4x)."""
    M = x.shape[0]
    dev = x.device
    with _lib.on_device(dev):
        A0 = ev._embed(x, ev.sdf.multires, ev.w_sdf, None, None, 1)
        acts = me.forward(ev.sdf_spec, A0, ev.sdf_W, ev.sdf_b, 1)
        A0d = ev._embed(x, ev.tr.multires, ev.w_def, ev.conds, bi, 1)
        actsd = me.forward(ev.tr.spec, A0d, ev.def_W, ev.def_b, 1)
        q = x + actsd[-1][:, :3]
        y, jl = ev.skin.fused(q, ev.A, ev.trans, bi, with_jac=update)
        conv = torch.empty((M,), dtype=torch.bool, device=dev)
        a = _lib.SrNewton2Args()
        a.M, a.sdf, a.ld_sdf = M, _lib.ptr(acts[-1]), acts[-1].stride(0)
        a.y, a.jlbs, a.rays, a.cam, a.converged = _lib.ptr(y), _lib.ptr(jl), _lib.ptr(rays), _lib.ptr(cam), _lib.ptr(conv)
        a.dthreshold, a.athreshold, a.w1, a.w2 = dthr, athr, w1, w2
        if not update:
            _lib.call("sr_newton_prepare", ctypes.byref(a), _lib.stream_of(x))
            return None, conv
        t = torch.empty((M, 4), dtype=torch.float32, device=dev)
        s = torch.empty((M,), dtype=torch.float32, device=dev)
        a.t_out, a.ld_t, a.s_out = _lib.ptr(t), 4, _lib.ptr(s)
        _lib.call("sr_newton_prepare", ctypes.byref(a), _lib.stream_of(x))
        ones = ev.unit_cotangent(M)
        A0bar, _, _ = me.reverse(ev.sdf_spec, A0, ev.sdf_WT, acts, ones, 1, True, False)
        gf = torch.empty_like(x)
        _lib.call("sr_pe_embed_bwd", _lib.ptr(x), M, ev.sdf.multires, _lib.ptr(ev.w_sdf), 1, _lib.ptr(A0bar), A0bar.stride(0), _lib.ptr(gf), _lib.stream_of(x))
        A0dbar, _, _ = me.reverse(ev.tr.spec, A0d, ev.def_WT, actsd, t, 1, True, False)
        goff = torch.empty_like(x)
        _lib.call("sr_pe_embed_bwd", _lib.ptr(x), M, ev.tr.multires, _lib.ptr(ev.w_def), 1, _lib.ptr(A0dbar), A0dbar.stride(0), _lib.ptr(goff), _lib.stream_of(x))
        xnew = torch.empty_like(x)
        a.grad_f, a.grad_off, a.p, a.p_out = _lib.ptr(gf), _lib.ptr(goff), _lib.ptr(x), _lib.ptr(xnew)
        _lib.call("sr_newton_apply", ctypes.byref(a), _lib.stream_of(x))
    return xnew, conv


def _newton(ev, x, bi, rays, cam, group, dthr, athr, w1, w2, update):
    sdf_rows, off_rows, y, jl = ev.evaluate(x, bi, group)
    M = x.shape[0]
    conv = torch.empty((M,), dtype=torch.bool, device=x.device)
    xnew = torch.empty_like(x) if update else None
    a = _lib.SrNewtonArgs()
    a.M, a.group = M, group
    a.sdf4, a.ld_sdf, a.off4, a.ld_off = _lib.ptr(sdf_rows), sdf_rows.stride(0), _lib.ptr(off_rows), off_rows.stride(0)
    a.y, a.jlbs, a.rays, a.cam = _lib.ptr(y), _lib.ptr(jl), _lib.ptr(rays), _lib.ptr(cam)
    a.p, a.p_out, a.converged = _lib.ptr(x), _lib.ptr(xnew), _lib.ptr(conv)
    a.dthreshold, a.athreshold, a.w1, a.w2 = dthr, athr, w1, w2
    with _lib.on_device(x.device):
        _lib.call("sr_newton_update", ctypes.byref(a), _lib.stream_of(x))
    return xnew, conv


# ------------------------------------------------------------------------------------------------
# Device-driven refiner: the queue of unfinished rays, its compaction and every row count stay on the GPU
# (csrc/refiner.hip + the layer chains of csrc/mlp_gemm.hip); the host issues a FIXED sequence of launches per call
# (3 small kernels + 2 chains per Newton step) and never synchronises.
DEVICE_DRIVEN = True
_WORKSPACES = {}     # (device, stream handle) -> _RefinerWorkspace (grow-only)


class _RefinerWorkspace:
    def __init__(self, dev, cap, ev, times_cap):
        if os.environ.get("SR_POISON_WORKSPACE") == "1":       # debugging aid: every buffer starts as NaN / -1 (a kernel that reads what no kernel wrote shows up)
            f = lambda *shape: torch.full(shape, float("nan"), dtype=torch.float32, device=dev)
            i = lambda *shape: torch.full(shape, -1, dtype=torch.int32, device=dev)
        else:
            f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
            i = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
        self.cap, self.times_cap = cap, times_cap
        self.live = torch.zeros(times_cap + 3, dtype=torch.int32, device=dev)
        self.x, self.v = [f(cap, 3), f(cap, 3)], [f(cap, 3), f(cap, 3)]
        self.frame, self.orig = [i(cap), i(cap)], [i(cap), i(cap)]
        self.unit, self.t, self.s = f(cap, 4), f(cap, 4), f(cap)
        self.conv = torch.empty(cap, dtype=torch.uint8, device=dev)
        sl, dl = ev.sdf_spec.layers, ev.tr.spec.layers
        self.a0, self.a0d = f(cap, me.pad4(ev.sdf_spec.K0)), f(cap, me.pad4(ev.tr.spec.K0))
        self.sdf_act = [f(cap, me.pad4(L.N + L.nfill)) for L in sl]
        self.def_act = [f(cap, me.pad4(L.N + L.nfill)) for L in dl]
        self.sdf_zbar = [None] + [f(cap, me.pad4(L.K)) for L in sl[1:]]       # cotangent of layer l's INPUT (own buffer per layer:
        self.def_zbar = [None] + [f(cap, me.pad4(L.K)) for L in dl[1:]]       # the skip part of one of them is read at the end)
        self.a0bar, self.a0dbar = f(cap, me.pad4(ev.sdf_spec.K0)), f(cap, me.pad4(ev.tr.spec.K0))


def _fill_gemm(g, A, B, C, N, K, bias, act, mode, out_scale=1.0, aux=None, naux_fwd=0, nact_bwd=0, aux_scale=1.0):
    g.A, g.lda, g.B, g.ldb, g.C, g.ldc = _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(C), C.stride(0)
    g.M, g.N, g.K, g.bias = 0, N, K, _lib.ptr(bias)
    g.group, g.act, g.mode, g.out_scale = 1, act, mode, out_scale
    g.aux, g.ldaux, g.naux_fwd, g.nact_bwd, g.aux_scale = _lib.ptr(aux), 0 if aux is None else aux.stride(0), naux_fwd, nact_bwd, aux_scale


def _forward_chain(ws, ev):
    c = _lib.SrChainArgs()
    nets = ((ev.sdf_spec, ev.sdf_W, ev.sdf_b, ws.a0, ws.sdf_act), (ev.tr.spec, ev.def_W, ev.def_b, ws.a0d, ws.def_act))
    c.nlayers = max(len(n[0].layers) for n in nets)
    for l in range(c.nlayers):
        k = 0
        for spec, W, b, a0, acts in nets:
            if l >= len(spec.layers):
                continue
            L = spec.layers[l]
            _fill_gemm(c.g[l][k], a0 if l == 0 else acts[l - 1], W[l], acts[l], L.N, L.K, b[l], L.act, me.EPI_FWD, out_scale=L.out_scale,
                       aux=a0 if L.nfill else None, naux_fwd=L.nfill)
            k += 1
        c.nprob[l] = k
    return c


def _reverse_chain(ws, ev):
    """me.reverse (input gradient only) of both networks, layer by layer from the outputs: SDF cotangent rows (1,0,0,0), deformer
    cotangent rows `t`."""
    c = _lib.SrChainArgs()
    nets = ((ev.sdf_spec, ev.sdf_WT, ws.a0, ws.sdf_act, ws.sdf_zbar, ws.unit, ws.a0bar), (ev.tr.spec, ev.def_WT, ws.a0d, ws.def_act, ws.def_zbar, ws.t, ws.a0dbar))
    c.nlayers = max(len(n[0].layers) for n in nets)
    for r in range(c.nlayers):
        k = 0
        for spec, WT, a0, acts, zbar, ybar, a0bar in nets:
            nl = len(spec.layers)
            l = nl - 1 - r
            if l < 0:
                continue
            L = spec.layers[l]
            src = ybar if l == nl - 1 else zbar[l + 1]
            if l > 0:
                Pv = spec.layers[l - 1]
                _fill_gemm(c.g[r][k], src, WT[l], zbar[l], L.K, L.N, None, Pv.act, me.EPI_BWD, out_scale=Pv.out_scale, aux=acts[l - 1], nact_bwd=Pv.N,
                           aux_scale=Pv.out_scale)
            else:
                _fill_gemm(c.g[r][k], src, WT[0], a0bar, L.K, L.N, None, me.ACT_NONE, me.EPI_FWD)
            k += 1
        c.nprob[r] = k
    return c


def _optimize_device_driven(ev, cam, rays, initTmpPs, batch_inds, dthreshold, athreshold, w1, w2, times):
    dev = initTmpPs.device
    P = initTmpPs.shape[0]
    # ONE grow-only workspace per (device, stream): capacity = the largest ray count seen + 1/8 headroom, in steps of 1024 rows (the
    # Bernoulli ray selection changes P every call by a few per cent; 57 KB per row, so a power-of-two rounding of 6145 rays would
    # hold 0.47 GB).  It is allocated and only ever used with its stream current, so the caching allocator may hand a replaced
    # workspace's blocks to that stream again without any cross-stream hazard.  A workspace made for ANOTHER stream of this device
    # is dropped when the stream changes (its blocks go back to that stream's pool; work still queued there is ordered before any reuse).
    key = (dev, torch.cuda.current_stream(dev).cuda_stream)
    ws = _WORKSPACES.get(key)
    if ws is None or ws.cap < P or ws.times_cap < times:
        cap = max(1024, (P + P // 8 + 1023) // 1024 * 1024, 0 if ws is None else ws.cap)
        ws = None
        for k in [k for k in _WORKSPACES if k[0] == dev]:
            _WORKSPACES.pop(k, None)
        ws = _WORKSPACES[key] = _RefinerWorkspace(dev, cap, ev, max(times, 30))
    x0 = initTmpPs.contiguous().float()
    bi = batch_inds.contiguous()
    p_out = torch.empty_like(x0)
    conv_out = torch.empty(P, dtype=torch.uint8, device=dev)
    a = _lib.SrRefineArgs()
    a.P, a.times = P, times
    a.p0, a.rays, a.batch_inds, a.cam = _lib.ptr(x0), _lib.ptr(rays), _lib.ptr(bi), _lib.ptr(cam)
    a.L_sdf, a.w_sdf, a.L_def, a.w_def = ev.sdf.multires, _lib.ptr(ev.w_sdf), ev.tr.multires, _lib.ptr(ev.w_def)
    a.conds, a.ld_conds, a.E = _lib.ptr(ev.conds), ev.conds.stride(0), ev.conds.shape[1]
    A12 = ev.A[:, :, :3, :].contiguous()
    vol = ev.skin.ws.permute(0, 2, 3, 4, 1)
    a.A, a.trans, a.nframes = _lib.ptr(A12), _lib.ptr(ev.trans), A12.shape[0]
    a.vol, a.D, a.H, a.W = _lib.ptr(vol), vol.shape[1], vol.shape[2], vol.shape[3]
    box = ev.skin._box_consts()
    for i in range(3):
        a.bmin[i], a.bmax[i] = box[0][i], box[1][i]
    a.dthreshold, a.athreshold, a.w1, a.w2 = dthreshold, athreshold, w1, w2
    a.live = _lib.ptr(ws.live)
    for i in range(2):
        a.x[i], a.v[i], a.frame[i], a.orig[i] = _lib.ptr(ws.x[i]), _lib.ptr(ws.v[i]), _lib.ptr(ws.frame[i]), _lib.ptr(ws.orig[i])
    a.unit, a.conv, a.t, a.s = _lib.ptr(ws.unit), _lib.ptr(ws.conv), _lib.ptr(ws.t), _lib.ptr(ws.s)
    a.a0, a.ld_a0, a.a0d, a.ld_a0d = _lib.ptr(ws.a0), ws.a0.stride(0), _lib.ptr(ws.a0d), ws.a0d.stride(0)
    a.sdf_out, a.ld_sdf, a.def_out, a.ld_def = _lib.ptr(ws.sdf_act[-1]), ws.sdf_act[-1].stride(0), _lib.ptr(ws.def_act[-1]), ws.def_act[-1].stride(0)
    a.a0bar, a.ld_a0bar, a.a0dbar, a.ld_a0dbar = _lib.ptr(ws.a0bar), ws.a0bar.stride(0), _lib.ptr(ws.a0dbar), ws.a0dbar.stride(0)
    a.skipbar, a.ld_skipbar, a.n_skip = 0, 0, 0
    for l, L in enumerate(ev.sdf_spec.layers):                      # the skip concat: cotangent of the filler columns of layer l's output
        if L.nfill:
            z = ws.sdf_zbar[l + 1]
            a.skipbar, a.ld_skipbar, a.n_skip = z.data_ptr() + 4 * L.N, z.stride(0), L.nfill
    a.p_out, a.conv_out = _lib.ptr(p_out), _lib.ptr(conv_out)
    fwd, rev = _forward_chain(ws, ev), _reverse_chain(ws, ev)
    for c in (fwd, rev):
        c.m_mul, c.m_cap = 1, P
    with _lib.on_device(dev):
        st = _lib.stream_of(x0)
        ra = ctypes.byref(a)
        _lib.call("sr_refine_init", ra, st)

        prof = me.PROFILE if me.PROFILE.enabled else None
        fwd_flop = sum(2.0 * g.N * g.K for l in range(fwd.nlayers) for g in fwd.g[l][:fwd.nprob[l]])
        rev_flop = sum(2.0 * g.N * g.K for l in range(rev.nlayers) for g in rev.g[l][:rev.nprob[l]])
        marks = []

        def chain(c, phase, flop_per_row):
            c.m_dev = ws.live.data_ptr() + 4 * phase
            if prof is not None:                           # bench.py's roofline leg: the chain's FLOPs are live[phase] x (sum over its layers)
                e0, e1 = prof.pair()
                e0.record()
            _lib.call("sr_mlp_chain", ctypes.byref(c), st)
            if prof is not None:
                e1.record()
                marks.append((e0, e1, phase, flop_per_row))

        def evaluate(phase):                               # the first-layer inputs were written by whoever enqueued the rays
            chain(fwd, phase, fwd_flop)
        watch = ENQUEUE_WATCH
        if watch is not None: watch(a, ws, 0, st)
        evaluate(0)
        _lib.call("sr_refine_mid", ra, 0, 0, st)
        if watch is not None: watch(a, ws, 1, st)
        for k in range(1, times + 1):
            evaluate(k)
            _lib.call("sr_refine_mid", ra, k, 1, st)
            chain(rev, k, rev_flop)
            _lib.call("sr_refine_finish", ra, k, st)
            if watch is not None: watch(a, ws, k + 1, st)
        evaluate(times + 1)
        _lib.call("sr_refine_mid", ra, times + 1, 2, st)
        if prof is not None:
            prof.chains.append((marks, ws.live[:times + 2].clone(), me.PROFILE.overlap))
    x0.record_stream(torch.cuda.current_stream(dev)); bi.record_stream(torch.cuda.current_stream(dev))
    initTmpPs.copy_(p_out)
    return initTmpPs.detach(), conv_out.bool(), ws


ENQUEUE_WATCH = None   # tests: called as (args, workspace, phase, stream) after each launch that filled the queue of `phase`
_PINNED = {}   # device -> pinned int64 scratch for the asynchronous live-ray counts
COMPACT_BELOW = 0.5   # compact the working set once fewer than this fraction of its rays are still unfinished


def OptimizeSurfacePs(cam_pos, rays, initTmpPs, batch_inds, tmpSdf, ratio, deformer, defconds, dthreshold=5.e-5,
                      athreshold=0.02, w1=3.05, w2=1., times=5):
    """Same contract as the reference: returns (ps, converged) and writes the refined points back
    into `initTmpPs` (FindSurfacePs.py:152).

    The reference drops converged rays from the batch after every step (boolean indexing = one host sync per step, 11 a call).
    Here finished rays stay in the working set, frozen by a device-side mask, and the set is compacted lazily: the count of
    unfinished rays travels to pinned host memory asynchronously, and only when a count that has ALREADY arrived says that
    less than COMPACT_BELOW of the set is live does the loop pay the sync of a compaction.  Results are identical -- a
    finished ray's position and flag never change again -- but the host can run ahead of the GPU through the whole call."""
    with torch.no_grad():
        ev = _FusedEval(tmpSdf, deformer, defconds, ratio)
        dev = initTmpPs.device
        cam = cam_pos.detach().float().contiguous().view(3)
        rays = rays.detach().float().contiguous()
        P = initTmpPs.shape[0]
        finished = torch.zeros(P, dtype=torch.bool, device=dev)
        if P == 0:
            return initTmpPs.detach(), finished
        if DEVICE_DRIVEN:
            ps, ok, _ = _optimize_device_driven(ev, cam, rays, initTmpPs, batch_inds, dthreshold, athreshold, w1, w2, times)
            return ps, ok
        pin = _PINNED.get(dev)
        if pin is None or pin.numel() < times + 2:
            pin = _PINNED[dev] = torch.zeros(times + 2, dtype=torch.int64).pin_memory()
        live = None                                   # None: the working set is all rays, in order
        x = initTmpPs.contiguous().float()
        bi_l, rays_l = batch_inds.contiguous(), rays
        done = torch.zeros(P, dtype=torch.bool, device=dev)
        counts = []                                   # (event, slot, size of the working set when it was taken)
        for it in range(times + 1):
            # lazy compaction on the newest count that has already landed on the host
            while len(counts) > 1 and counts[1][0].query():
                counts.pop(0)
            if counts and counts[0][0].query() and counts[0][2] == x.shape[0] and x.shape[0] - int(pin[counts[0][1]]) < COMPACT_BELOW * x.shape[0]:
                keep = (~done).nonzero(as_tuple=False).view(-1)          # the one sync, taken only when it pays
                if live is None:
                    initTmpPs.copy_(x); finished.copy_(done); live = keep
                else:
                    initTmpPs[live] = x; finished[live] = done; live = live[keep]
                x, bi_l, rays_l = x[keep].contiguous(), bi_l[keep].contiguous(), rays_l[keep].contiguous()
                done = torch.zeros(keep.numel(), dtype=torch.bool, device=dev)
                counts = []
            if x.shape[0] == 0:
                break
            last = it == times
            if REVERSE_MODE:
                xnew, conv = _newton_reverse(ev, x, bi_l, rays_l, cam, dthreshold, athreshold, w1, w2, not last)
            else:
                xnew, conv = _newton(ev, x, bi_l, rays_l, cam, 1 if last else 4, dthreshold, athreshold, w1, w2, not last)
            if not last:
                x = torch.where(done[:, None], x, xnew)    # rays finished before this step stay where they were
            done = done | conv
            if not last:
                pin[it].copy_(done.sum(), non_blocking=True)            # finished count; live = size - finished
                e = torch.cuda.Event(); e.record()
                counts.append((e, it, x.shape[0]))
        if live is None:
            initTmpPs.copy_(x); finished.copy_(done)
        else:
            initTmpPs[live] = x; finished[live] = done
    return initTmpPs.detach(), finished
