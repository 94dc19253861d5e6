"""Host side of the fused MLP engine: drives the fp32-MFMA layer kernels of csrc/mlp_gemm.hip.

One engine serves the three networks of the hot path (SDF a2/a3, deformer a4, render a10) and
all the derivative orders the reference reaches through autograd:

  * rows are tangent-interleaved: a sample owns `group` rows = primal + (group-1) forward tangents;
  * `forward(A0, ..., group)` runs the layers (activations, skip concat, tangent propagation fused
    in the GEMM epilogues) and keeps the stored activations;
  * `reverse(...)` is ONE reverse sweep over that (augmented) network: cotangents of the
    pre-activations, the weight/bias gradients and the input cotangent.

`MLPCoreFunction` wraps it for torch autograd to second order:
    y = MLP(A0; W, b)                    forward            (group 1)
    (dA0, dW, db) = backward(ybar)       reverse            (group 1)
    double backward with cotangent U on dA0:  S = <ybar, J_A0 U> is the forward tangent of y along U,
    so its gradients are one group-2 forward + one group-2 reverse (reverse-over-reverse == reverse
    over a 1-tangent forward).  Cotangents on dW/db (third-party code differentiating parameter
    gradients) are not supported -- the reference never does that.
"""
import math
import ctypes
import os
import weakref
import torch

from . import _lib

ACT_NONE, ACT_SOFTPLUS100, ACT_RELU = 0, 1, 2
EPI_FWD, EPI_BWD = 0, 1


def pad4(n):
    return (n + 3) // 4 * 4


class LayerSpec:
    __slots__ = ("K", "N", "act", "out_scale", "nfill")

    def __init__(self, K, N, act, out_scale=1.0, nfill=0):
        self.K, self.N, self.act, self.out_scale, self.nfill = K, N, act, float(out_scale), nfill


class MLPSpec:
    """Static description of one MLP: per layer (in, out, activation), optional skip concat:
    the layer BEFORE the skip writes [act(z) | A0[:, :nfill]] * out_scale (network.py:88-89)."""

    def __init__(self, layers, K0):
        self.layers = layers
        self.K0 = K0

    @staticmethod
    def sdf(multires=6, width=512, nlin=9, skip_in=(4,), d_out=257):
        K0 = 3 + 6 * multires
        layers = []
        k = K0
        for l in range(nlin):
            last = l == nlin - 1
            n = d_out if last else width
            if (l + 1) in skip_in:
                layers.append(LayerSpec(k, n - K0, ACT_SOFTPLUS100, 1.0 / math.sqrt(2.0), nfill=K0))
            else:
                layers.append(LayerSpec(k, n, ACT_NONE if last else ACT_SOFTPLUS100))
            k = n
        return MLPSpec(layers, K0)

    @staticmethod
    def relu_mlp(K0, widths):
        layers, k = [], K0
        for i, n in enumerate(widths):
            layers.append(LayerSpec(k, n, ACT_NONE if i == len(widths) - 1 else ACT_RELU))
            k = n
        return MLPSpec(layers, K0)


class _GemmProfile:
    """Optional HIP-event timing of every layer-GEMM launch on the launching stream (bench.py's roofline
    leg): achieved = sum(2 M N K) / sum(event time).  Off by default: zero overhead in the product path."""

    def __init__(self):
        self.enabled = False
        self.records = []
        self.pool = []
        self.overlap = False      # set by OptimNetwork.forward while two streams feed the GPU: an event pair then brackets a kernel
                                  # that shares the machine with the other stream's kernels, not a kernel's own duration

    def reset(self, enabled, reserve=0):
        """`reserve` event pairs are created (and recorded once, which is what allocates the HIP event) ahead of time, so that
        inside the measured region a pair costs two hipEventRecord calls and nothing else."""
        self.enabled = enabled
        self.records = []
        self.chains = []          # persistent layer-chain launches of the refiner: ([(e0, e1, phase, flop per row)], live counts, overlap flag)
        self.pool = []
        for _ in range(reserve if enabled else 0):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); e1.record()
            self.pool.append((e0, e1))

    def pair(self):
        return self.pool.pop() if self.pool else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def summary(self):
        if not self.records:
            return {"tflops": 0.0, "launches": 0, "avg_us": 0.0, "avg_flop": 0.0}
        torch.cuda.synchronize()
        for marks, live, overlap in self.chains:          # fold the chain launches in as records: FLOPs from the device-side row counts
            lv = live.tolist()
            for e0, e1, phase, fpr in marks:
                self.records.append((e0, e1, fpr * lv[phase], lv[phase], overlap, (0, 0, "chain", 1)))
        self.chains = []
        times = [r[0].elapsed_time(r[1]) for r in self.records]

        def rate(sel):
            f = sum(r[2] for r, t, k in zip(self.records, times, sel) if k)
            ms = sum(t for t, k in zip(times, sel) if k)
            n = sum(1 for k in sel if k)
            return (f / (ms * 1e-3) / 1e12 if ms > 0 else 0.0), n, (ms * 1e3 / n if n else 0.0), (f / n if n else 0.0)

        nt = [r[5][2] != "tn" for r in self.records]                  # the NT tile code (forward / backward-data GEMMs, refiner chains)
        alone = [k and not r[4] for k, r in zip(nt, self.records)]
        tf, n, us, fl = rate(alone)                                   # launches that had the GPU to themselves
        tf_all, n_all, us_all, _ = rate(nt)
        tf_big, n_big, _, _ = rate([a and r[3] >= 65536 for a, r in zip(alone, self.records)])
        tf_tn, n_tn, us_tn, _ = rate([not k for k in nt])             # the weight-gradient (TN) kernel + slab reduction
        self._times = times
        f_nt = float(sum(r[2] for r, k in zip(self.records, nt) if k))
        f_tn = float(sum(r[2] for r, k in zip(self.records, nt) if not k))
        return {"tflops": round(tf, 3), "launches": n, "avg_us": round(us, 3), "avg_flop": round(fl, 1), "tflops_large": round(tf_big, 3),
                "launches_large": n_big, "tflops_all": round(tf_all, 3), "launches_all": n_all, "avg_us_all": round(us_all, 3),
                "avg_flop_all": round(f_nt / max(n_all, 1), 1), "flops_total": f_nt, "flops_total_tn": f_tn,
                "tflops_tn": round(tf_tn, 3), "launches_tn": n_tn, "avg_us_tn": round(us_tn, 3)}

    def by_shape(self):
        """Per (M, N, K, epilogue mode, overlap flag): launches, total FLOP, total event time, TFLOP/s -- the table that makes
        the roofline figure reproducible next to the rocprof kernel trace (tools/summarize_profile.py joins the two)."""
        if not self.records:
            return []
        times = getattr(self, "_times", None) or [r[0].elapsed_time(r[1]) for r in self.records]
        acc = {}
        for r, t in zip(self.records, times):
            key = (r[3],) + tuple(r[5]) + (bool(r[4]),)
            a = acc.setdefault(key, [0, 0.0, 0.0])
            a[0] += 1; a[1] += r[2]; a[2] += t
        rows = [{"M": k[0], "N": k[1], "K": k[2], "mode": k[3], "group": k[4], "overlap": k[5], "launches": v[0], "flop": v[1], "ms": round(v[2], 4),
                 "tflops": round(v[1] / (v[2] * 1e-3) / 1e12, 2) if v[2] > 0 else 0.0} for k, v in acc.items()]
        rows.sort(key=lambda r: -r["ms"])
        return rows


PROFILE = _GemmProfile()

def _gemm_nt(A, lda, B, ldb, C, ldc, M, N, K, bias, group, act, mode, out_scale=1.0, aux=None, ldaux=0, naux_fwd=0,
             nact_bwd=0, aux_scale=1.0):
    a = _lib.SrGemmArgs()
    a.A, a.lda, a.B, a.ldb, a.C, a.ldc = _lib.ptr(A), lda, _lib.ptr(B), ldb, _lib.ptr(C), ldc
    a.M, a.N, a.K = M, N, K
    a.bias = _lib.ptr(bias)
    a.group, a.act, a.mode, a.out_scale = group, act, mode, out_scale
    a.aux, a.ldaux, a.naux_fwd, a.nact_bwd, a.aux_scale = _lib.ptr(aux), ldaux, naux_fwd, nact_bwd, aux_scale
    if PROFILE.enabled and M >= 128 and N > 32:      # the 128x128-tile kernel only
        e0, e1 = PROFILE.pair()
        e0.record()
        _lib.call("sr_mlp_gemm_nt", ctypes.byref(a), _lib.stream_of(C))
        e1.record()
        PROFILE.records.append((e0, e1, 2.0 * M * N * K, M, PROFILE.overlap, (N, K, mode, group)))
        return
    _lib.call("sr_mlp_gemm_nt", ctypes.byref(a), _lib.stream_of(C))


def _gemm_tn(Z, ldz, A, lda, R, N, K, lddw, group=1, dW=None, db=None, accumulate=False):
    """dW [N, lddw] (+)= Z[:, :N]^T A[:, :K] and db [N] (+)= sum of the primal rows of Z (deterministic slab reductions)."""
    splits = ctypes.c_int32(0)
    ws = _lib.raw("sr_mlp_gemm_tn_workspace_floats")(R, N, lddw, ctypes.byref(splits))
    if dW is None:
        dW = torch.empty((N, lddw), dtype=torch.float32, device=Z.device)
    partial = torch.empty((max(int(ws), 1) + splits.value * N,), dtype=torch.float32, device=Z.device)
    if db is None:
        db = torch.empty((N,), dtype=torch.float32, device=Z.device)
    a = _lib.SrGemmTnArgs()
    a.Z, a.ldz, a.A, a.lda, a.dW, a.lddw, a.partial = _lib.ptr(Z), ldz, _lib.ptr(A), lda, _lib.ptr(dW), lddw, _lib.ptr(partial)
    a.R, a.N, a.K, a.splits, a.accumulate = R, N, K, splits.value, 1 if accumulate else 0
    a.db, a.db_partial, a.group = _lib.ptr(db), _lib.ptr(partial) + 4 * max(int(ws), 1), group
    if PROFILE.enabled and R >= 128 and N > 32:        # weight-gradient GEMM + its slab reduction, as one interval
        e0, e1 = PROFILE.pair()
        e0.record()
        _lib.call("sr_mlp_gemm_tn", ctypes.byref(a), _lib.stream_of(Z))
        e1.record()
        PROFILE.records.append((e0, e1, 2.0 * R * N * K, R, PROFILE.overlap, (N, K, "tn", group)))
        return dW, db
    _lib.call("sr_mlp_gemm_tn", ctypes.byref(a), _lib.stream_of(Z))
    return dW, db


GROUP_TN_BELOW = 16384     # rows: the weight gradients of a reverse sweep on fewer rows than this go out as ONE grouped launch (0 = never)


def _tn_is_narrow(N, lddw):
    return lddw <= 64 and N >= 256          # (csrc/mlp_gemm.hip::tn_narrow: the 256 x 64 tile shape of the first layers is not grouped)


def _gemm_tn_group(problems):
    """`problems`: list of (Z, ldz, A, lda, R, N, K, lddw, group, dW, db, accumulate) with distinct dW / db -- the weight gradients of
    one reverse sweep.  One launch for all tiles and slabs + one for all slab reductions (sr_mlp_gemm_tn_group); each problem is computed
    exactly as `_gemm_tn` would compute it alone: same splits, same slab order, bit-identical results."""
    g = _lib.SrGemmTnGroupArgs()
    g.n = len(problems)
    sizes, total = [], 0
    for (Z, ldz, A, lda, R, N, K, lddw, group, dW, db, accumulate) in problems:
        splits = ctypes.c_int32(0)
        ws = max(int(_lib.raw("sr_mlp_gemm_tn_workspace_floats")(R, N, lddw, ctypes.byref(splits))), 1)
        sizes.append((ws, splits.value, total))
        total += (ws + splits.value * N + 3) // 4 * 4                     # (every problem's partial block stays 16-byte aligned)
    partial = torch.empty((total,), dtype=torch.float32, device=problems[0][0].device)
    for a, (Z, ldz, A, lda, R, N, K, lddw, group, dW, db, accumulate), (ws, splits, off) in zip(g.p, problems, sizes):
        a.Z, a.ldz, a.A, a.lda, a.dW, a.lddw, a.partial = _lib.ptr(Z), ldz, _lib.ptr(A), lda, _lib.ptr(dW), lddw, _lib.ptr(partial) + 4 * off
        a.R, a.N, a.K, a.splits, a.accumulate = R, N, K, splits, 1 if accumulate else 0
        a.db, a.db_partial, a.group = _lib.ptr(db), _lib.ptr(partial) + 4 * (off + ws), group
    _lib.call("sr_mlp_gemm_tn_group", ctypes.byref(g), _lib.stream_of(partial))


def _colsum(Z, ldz, R, N, group):
    out = torch.zeros((N,), dtype=torch.float32, device=Z.device)
    _lib.call("sr_colsum_rows", _lib.ptr(Z), ldz, R, N, group, _lib.ptr(out), _lib.stream_of(Z))
    return out


def _check_mat(t, name):
    if t.dtype != torch.float32 or not t.is_cuda or t.dim() != 2 or t.stride(1) != 1 or (t.stride(0) % 4) or (t.data_ptr() % 16):
        raise RuntimeError(f"mlp_engine: {name} must be a GPU fp32 matrix with unit column stride, row pitch % 4 == 0 "
                           f"and 16-byte alignment (got {tuple(t.shape)}, strides {t.stride()}, {t.dtype}, {t.device})")


def forward(spec, A0, Ws, bs, group):
    """A0 [R, >=K0] (pitch % 4 == 0); Ws[l] [N_l, pad4(K_l)]; returns the stored activations per layer
    (each [R, pad4(N_l + nfill_l)]); the last one holds the network output in its first N_L columns."""
    _check_mat(A0, "A0")
    R = A0.shape[0]
    acts, X = [], A0
    seg = _bias_segments(bs[0], R, group)
    with _lib.on_device(A0.device):
        for l, L in enumerate(spec.layers):
            _check_mat(Ws[l], f"W{l}")
            C = torch.empty((R, pad4(L.N + L.nfill)), dtype=torch.float32, device=A0.device)
            if l == 0 and seg is not None:
                # first layer with one bias row per row segment (frame-major batches whose per-frame code product was hoisted into
                # the bias): one launch per segment into its row slice -- a K = 39 launch is bound by the 2 KB-per-row write whatever
                # its row count; every later layer runs ONCE over all the segments
                S, rs = seg
                for s in range(S):
                    Xs, Cs = X[s * rs:(s + 1) * rs], C[s * rs:(s + 1) * rs]
                    _gemm_nt(Xs, X.stride(0), Ws[l], Ws[l].stride(0), Cs, C.stride(0), rs, L.N, L.K, bs[0][s], group, L.act, EPI_FWD,
                             out_scale=L.out_scale, aux=A0[s * rs:(s + 1) * rs] if L.nfill else None, ldaux=A0.stride(0), naux_fwd=L.nfill)
            else:
                _gemm_nt(X, X.stride(0), Ws[l], Ws[l].stride(0), C, C.stride(0), R, L.N, L.K, bs[l], group, L.act, EPI_FWD,
                         out_scale=L.out_scale, aux=A0 if L.nfill else None, ldaux=A0.stride(0), naux_fwd=L.nfill)
            acts.append(C)
            X = C
    return acts


def _bias_segments(b0, R, group):
    """-> (S, rows per segment) when the first-layer bias is a matrix [S, N_0] (row segment s of the batch uses bias row s), else None."""
    if b0 is None or b0.dim() != 2:
        return None
    S = b0.shape[0]
    if S < 1 or R % S or (R // S) % group or b0.stride(1) != 1:
        raise RuntimeError(f"mlp_engine: segmented first-layer bias {tuple(b0.shape)} does not divide {R} rows of group {group}")
    return S, R // S


# Weight-gradient GEMMs on a stream of their own.  In a reverse sweep dW_l = Zbar_l^T X_{l-1} and the backward-data GEMM
# Zbar_{l-1} = (Zbar_l W_l) . act' depend on the same Zbar_l and on nothing of each other; on one stream they alternate, every launch
# drains the machine before the next one fills it (the last partial wave of 128x128 tiles of an 85k-row batch leaves ~5 % of a launch
# idle, plus the launch gap).  With the weight gradients on a second stream the backward-data chain is the critical path and the
# weight-gradient workgroups fill its tails.  Only in deferred mode (the results land in the per-layer buffers, nobody reads them
# before flush_param_grads, which joins the stream); all weight-gradient launches share ONE stream, so the accumulation order into
# a buffer is the program order -- results are bit-identical to the one-stream schedule.
TN_SIDE_STREAM = True          # (module attribute: tests compare the one- and two-stream schedules bit for bit)
# (Row slabs of the deferred weight-gradient launches: all of them, two workgroups per CU -- since round 5 the weight-gradient stream is
# what the tail of the big backward waits for; rounds 3-4 ran half the count.)
_TN_STREAMS = {}
_TN_PENDING = set()


DEBUG_TN_DELAY_MS = int(os.environ.get("SR_DEBUG_TN_DELAY_MS", "0"))      # race amplifier: hold the weight-gradient stream before every launch
_DELAY_FLAG = {}


def _debug_delay(device, ms):
    f = _DELAY_FLAG.get(str(device))
    if f is None:
        f = _DELAY_FLAG[str(device)] = torch.zeros(2, dtype=torch.int32, device=device)
    _lib.call('sr_stream_flag_wait', f.data_ptr(), 0x40000000, 0, ms, torch.cuda.current_stream(device).cuda_stream)


def _tn_stream(device):
    key = str(device)
    st = _TN_STREAMS.get(key)
    if st is None:
        st = _TN_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def join_weight_gradient_stream():
    """The current stream of every device with outstanding weight-gradient launches waits for them."""
    for key in list(_TN_PENDING):
        st = _TN_STREAMS[key]
        torch.cuda.current_stream(st.device).wait_stream(st)
    _TN_PENDING.clear()


def reverse(spec, A0, WTs, acts, Ybar, group, need_input_grad=True, need_param_grad=True, Ws=None, bs=None):
    """One reverse sweep.  Ybar [R, >= N_L] (pitch % 4): cotangent of the output rows.  WTs[l] = W_l^T as
    [K_l, pad4(N_l)].  Returns (A0bar [R, pad4(K0)] or None, dWs [N_l, pad4(K_l)], dbs [N_l])."""
    _check_mat(Ybar, "Ybar")
    R = A0.shape[0]
    nl = len(spec.layers)
    dWs, dbs = [None] * nl, [None] * nl
    A0bar_extra = None
    Zbar = Ybar
    A0bar = None
    seg = _bias_segments(bs[0], R, group) if bs is not None else None
    # weight gradients of a SMALL sweep (deferred mode, their own stream): collected here and launched as one group after the last layer --
    # each is < 100 workgroups that end before the next starts (26 launches per iteration at 27 TFLOP/s in round 5)
    grouped = [] if (0 < R < GROUP_TN_BELOW and TN_SIDE_STREAM and not PROFILE.enabled) else None
    with _lib.on_device(A0.device):
        for l in range(nl - 1, -1, -1):
            L = spec.layers[l]
            X = A0 if l == 0 else acts[l - 1]
            if need_param_grad:
                sink = _deferred_sink(Ws[l], bs[l]) if (DEFERRED_PARAM_GRADS and Ws is not None and not (l == 0 and seg is not None)) else None
                if sink is None and DEFERRED_PARAM_GRADS and Ws is not None and not Ws[l].requires_grad:
                    e = _ENTRY_BY_PTR.get(Ws[l].data_ptr())
                    if e is not None and any(r() is not None and r().requires_grad for r in e.get("src", ())):
                        raise RuntimeError("mlp_engine: a packed weight handed out without autograd node (deferred mode) met a layer call "
                                           "that has no deferred sink (bias that is not the layer's own leaf parameter?): its gradient would be lost")
                if sink is not None and grouped is not None and not _tn_is_narrow(L.N, pad4(L.K)) and len(grouped) < _lib.SR_TN_GROUP_MAX:
                    grouped.append((Zbar, Zbar.stride(0), X, X.stride(0), R, L.N, L.K, pad4(L.K), group, sink[0], sink[1], sink[2]))
                elif sink is not None and TN_SIDE_STREAM and not PROFILE.enabled:
                    main, side = torch.cuda.current_stream(A0.device), _tn_stream(A0.device)
                    ready = torch.cuda.Event()
                    ready.record(main)                                   # Zbar (and, for a partial first use, the zeroed buffers) are final here
                    side.wait_event(ready)
                    with torch.cuda.stream(side):
                        if DEBUG_TN_DELAY_MS:
                            _debug_delay(A0.device, DEBUG_TN_DELAY_MS)
                        _gemm_tn(Zbar, Zbar.stride(0), X, X.stride(0), R, L.N, L.K, pad4(L.K), group, dW=sink[0], db=sink[1], accumulate=sink[2])
                    Zbar.record_stream(side); X.record_stream(side)      # both may be freed by the main stream's owner before the side stream has read them
                    _TN_PENDING.add(str(A0.device))
                elif sink is not None:        # accumulate straight into the per-step gradient buffers (no autograd traffic)
                    _gemm_tn(Zbar, Zbar.stride(0), X, X.stride(0), R, L.N, L.K, pad4(L.K), group, dW=sink[0], db=sink[1], accumulate=sink[2])
                elif l == 0 and seg is not None:
                    # segmented first-layer bias: the weight gradient is the sum over the segments (accumulated launch by launch, in
                    # segment order), the bias gradient one row per segment
                    S, rs = seg
                    dWs[0] = torch.empty((L.N, pad4(L.K)), dtype=torch.float32, device=A0.device)
                    dbs[0] = torch.zeros((S, L.N), dtype=torch.float32, device=A0.device)
                    for s in range(S):
                        _gemm_tn(Zbar[s * rs:(s + 1) * rs], Zbar.stride(0), X[s * rs:(s + 1) * rs], X.stride(0), rs, L.N, L.K, pad4(L.K), group,
                                 dW=dWs[0], db=dbs[0][s], accumulate=s > 0)
                else:
                    dWs[l], dbs[l] = _gemm_tn(Zbar, Zbar.stride(0), X, X.stride(0), R, L.N, L.K, pad4(L.K), group)
            if l > 0:
                Pv = spec.layers[l - 1]
                Znew = torch.empty((R, pad4(L.K)), dtype=torch.float32, device=A0.device)
                _gemm_nt(Zbar, Zbar.stride(0), WTs[l], WTs[l].stride(0), Znew, Znew.stride(0), R, L.K, L.N, None, group, Pv.act,
                         EPI_BWD, out_scale=Pv.out_scale, aux=acts[l - 1], ldaux=acts[l - 1].stride(0), nact_bwd=Pv.N,
                         aux_scale=Pv.out_scale)
                if Pv.nfill and need_input_grad:
                    A0bar_extra = Znew[:, Pv.N:Pv.N + Pv.nfill]
                Zbar = Znew
            elif need_input_grad:
                A0bar = torch.empty((R, pad4(L.K)), dtype=torch.float32, device=A0.device)
                _gemm_nt(Zbar, Zbar.stride(0), WTs[0], WTs[0].stride(0), A0bar, A0bar.stride(0), R, L.K, L.N, None, 1, ACT_NONE,
                         EPI_FWD)
                if A0bar_extra is not None:
                    A0bar[:, :A0bar_extra.shape[1]] += A0bar_extra
        if grouped:
            main, side = torch.cuda.current_stream(A0.device), _tn_stream(A0.device)
            ready = torch.cuda.Event()
            ready.record(main)                                           # every Zbar of the sweep is final here
            side.wait_event(ready)
            with torch.cuda.stream(side):
                if DEBUG_TN_DELAY_MS:
                    _debug_delay(A0.device, DEBUG_TN_DELAY_MS)
                if len(grouped) == 1:
                    p = grouped[0]
                    _gemm_tn(*p[:9], dW=p[9], db=p[10], accumulate=p[11])
                else:
                    _gemm_tn_group(grouped)
            for p in grouped:
                p[0].record_stream(side); p[2].record_stream(side)
            _TN_PENDING.add(str(A0.device))
    return A0bar, dWs, dbs


def transpose_padded(W, K):
    """W [N, pad4(K)] -> W^T as [K, pad4(N)] (zero padded)."""
    N = W.shape[0]
    WT = torch.zeros((K, pad4(N)), dtype=W.dtype, device=W.device)
    WT[:, :N] = W[:, :K].t()
    return WT


def pad_cols(t, width):
    if t.shape[1] == width and t.is_contiguous():
        return t
    return torch.nn.functional.pad(t, (0, width - t.shape[1]))


def rows_pad(a, width, b=None, pair=False):
    """No-autograd helper of the Functions below: `a` [R, n] -> [R, width] zero padded in one launch (returned as is when it
    already has that shape and unit column stride / padded pitch); with `pair`, the group-2 interleave [2R, width] of the rows
    (a, b), where None stands for zero rows."""
    ref = a if a is not None else b
    R = ref.shape[0]
    if not pair and a.shape[1] == width and a.stride(1) == 1 and a.stride(0) % 4 == 0 and a.data_ptr() % 16 == 0:
        return a
    srcs = []
    for t in (a, b):
        if t is not None and (t.stride(1) != 1 or t.dtype != torch.float32):
            t = t.contiguous().float()
        srcs.append(t)
    a, b = srcs
    out = torch.empty((R * (2 if pair else 1), width), dtype=torch.float32, device=ref.device)
    with _lib.on_device(ref.device):
        _lib.call("sr_rows_pad", _lib.ptr(a), 0 if a is None else a.stride(0), 0 if a is None else min(a.shape[1], width),
                  _lib.ptr(b), 0 if b is None else b.stride(0), 0 if b is None else min(b.shape[1], width),
                  R, 2 if pair else 1, _lib.ptr(out), width, width, _lib.stream_of(ref))
    return out


def rows_frame_sum(X, index, n):
    """[n, E] per-frame sums of the rows of X [P, E] (any row pitch), out[f] = sum of the rows with index == f -- the deterministic
    replacement of `zeros(n, E).index_add(0, index, X)` (float atomics) for the gradient of a per-frame code gathered into every row."""
    P, E = X.shape
    if X.dtype != torch.float32 or X.stride(1) != 1:
        X = X.float().contiguous()
    index = index.contiguous()
    out = torch.empty((n, E), dtype=torch.float32, device=X.device)
    part = torch.empty((max(int(_lib.raw("sr_rows_frame_sum_workspace_floats")(P, E, n)), 1),), dtype=torch.float32, device=X.device)
    with _lib.on_device(X.device):
        _lib.call("sr_rows_frame_sum", _lib.ptr(X), X.stride(0), P, E, _lib.ptr(index), n, _lib.ptr(part), _lib.ptr(out), _lib.stream_of(X))
    return out


def interleave(rows):
    """[R,K] x g -> [g*R, K] with sample-major interleaving (primal, tangent_1, ...)."""
    return torch.stack(rows, dim=1).reshape(rows[0].shape[0] * len(rows), rows[0].shape[1])


import contextlib

_INPUT_GRADS_ONLY = False


@contextlib.contextmanager
def input_grads_only():
    """Wrap `torch.autograd.grad(outputs, points, ...)` calls that only want d/d(input) (normals, Jacobians):
    a custom autograd Function cannot see which inputs the engine call targets (`needs_input_grad` is static),
    so without this hint every such call would also run the weight-gradient GEMMs -- wasted work in plain
    mode, and WRONG accumulation into the per-step buffers in deferred mode."""
    global _INPUT_GRADS_ONLY
    prev = _INPUT_GRADS_ONLY
    _INPUT_GRADS_ONLY = True
    try:
        yield
    finally:
        _INPUT_GRADS_ONLY = prev


class MLPCoreFunction(torch.autograd.Function):
    """y[P, N_L] = MLP(A0[P, pad4(K0)]; W_l [N_l, pad4(K_l)], b_l)."""

    @staticmethod
    def forward(ctx, spec, A0, *wb):
        nl = len(spec.layers)
        Ws, bs = list(wb[:nl]), list(wb[nl:])
        acts = forward(spec, A0, Ws, bs, 1)
        ctx.spec = spec
        ctx.save_for_backward(A0, *wb, *acts[:-1])
        ctx.set_materialize_grads(False)
        NL = spec.layers[-1].N
        out = acts[-1]
        return out if out.shape[1] == NL else out[:, :NL]

    @staticmethod
    def backward(ctx, ybar):
        if ybar is None:
            return (None,) * (2 + 2 * len(ctx.spec.layers))
        saved = ctx.saved_tensors
        nl = len(ctx.spec.layers)
        A0, wb, acts = saved[0], saved[1:1 + 2 * nl], saved[1 + 2 * nl:]
        need_par = any(ctx.needs_input_grad[2:]) and not _INPUT_GRADS_ONLY
        outs = MLPCoreBackward.apply(ctx.spec, ctx.needs_input_grad[1], need_par, A0, ybar, *wb, *acts)
        return (None,) + tuple(outs)


class MLPCoreBackward(torch.autograd.Function):
    """(A0bar, dW_0.., db_0..) = reverse sweep; itself differentiable w.r.t. (A0, ybar, W, b) for a
    cotangent on A0bar (see module docstring)."""

    @staticmethod
    def forward(ctx, spec, need_in, need_par, A0, ybar, *rest):
        nl = len(spec.layers)
        Ws, bs, acts = list(rest[:nl]), list(rest[nl:2 * nl]), list(rest[2 * nl:])
        NL = spec.layers[-1].N
        yb = rows_pad(ybar, pad4(NL))
        WTs = [transposed_of(Ws[l], spec.layers[l].K) for l in range(nl)]
        A0bar, dWs, dbs = reverse(spec, A0, WTs, acts, yb, 1, need_in, need_par, Ws, bs)
        ctx.spec = spec
        ctx.save_for_backward(A0, ybar, *Ws, *bs)
        ctx.set_materialize_grads(False)
        if not need_par:
            dWs, dbs = [None] * nl, [None] * nl
        if A0bar is not None and A0bar.shape[1] != A0.shape[1]:
            A0bar = rows_pad(A0bar[:, :spec.K0], A0.shape[1])
        return (A0bar,) + tuple(dWs) + tuple(dbs)

    @staticmethod
    def backward(ctx, U, *param_cots):
        spec = ctx.spec
        nl = len(spec.layers)
        if any(c is not None for c in param_cots):
            raise NotImplementedError("selfreconcode_amd: differentiating MLP parameter gradients again is not supported "
                                      "(the reference only differentiates input gradients, network.py:102-114)")
        n_in = 5 + 2 * nl + (nl - 1)
        if U is None:
            return (None,) * n_in
        saved = ctx.saved_tensors
        A0, ybar = saved[0], saved[1]
        Ws, bs = list(saved[2:2 + nl]), list(saved[2 + nl:2 + 2 * nl])
        NL = spec.layers[-1].N
        R = A0.shape[0]
        # group-2 augmented forward: rows (a0, U)
        A0i = rows_pad(A0, A0.shape[1], U, pair=True)                  # rows (a0, U), U zero padded to the pitch of A0
        acts2 = forward(spec, A0i, Ws, bs, 2)
        ydot = acts2[-1].view(R, 2, -1)[:, 1, :NL]                     # d S / d ybar
        ybi = rows_pad(None, pad4(NL), ybar, pair=True)                 # cotangent only on the tangent output
        WTs = [transposed_of(Ws[l], spec.layers[l].K) for l in range(nl)]
        need_in = ctx.needs_input_grad[3]
        need_par = any(ctx.needs_input_grad[5:5 + 2 * nl])
        A0bar2, dWs, dbs = reverse(spec, A0i, WTs, acts2, ybi, 2, need_in, need_par, Ws, bs)
        gA0 = A0bar2.view(R, 2, -1)[:, 0, :] if A0bar2 is not None else None
        if gA0 is not None and gA0.shape[1] != A0.shape[1]:
            gA0 = rows_pad(gA0[:, :spec.K0], A0.shape[1])
        return (None, None, None, gA0, ydot.contiguous()) + tuple(dWs) + tuple(dbs) + (None,) * (nl - 1)


# ------------------------------------------------------------------------------------------------
# Effective-weight packing with a per-parameter-version cache.  The reference recomputes
# W = g v/|v| in a forward pre-hook at EVERY module call (~30 calls per iteration); here the padded weight
# and its transpose are built once per optimizer step and handed out as aliases (no kernel launch on a hit),
# while each use still back-propagates into (g, v) through its own autograd node.
_PACK_CACHE = {}     # key: id(param) -> dict(sig=..., W=..., WT=..., norms=...)
_WT_BY_PTR = {}      # data_ptr of a packed W -> its transposed copy


def _sig(*ts):
    return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in ts)


def _drop_entry(key):
    e = _PACK_CACHE.pop(key, None)
    if e is not None:
        _WT_BY_PTR.pop(e["W"].data_ptr(), None)
        _ENTRY_BY_PTR.pop(e["W"].data_ptr(), None)


def _pack_entry(key, sig, build, owner=None):
    e = _PACK_CACHE.get(key)
    if e is None and owner is not None:
        weakref.finalize(owner, _drop_entry, key)      # the entry (packed weights, gradient buffers) goes when its parameter does
    if e is None or e["sig"] != sig:
        if e is not None:
            _WT_BY_PTR.pop(e["W"].data_ptr(), None)
        if e is not None and e.get("dirty"):
            raise RuntimeError("mlp_engine: parameters changed while deferred gradients were pending; call flush_param_grads() "
                               "before the optimizer step")
        if e is not None:
            _ENTRY_BY_PTR.pop(e["W"].data_ptr(), None)
        e = build()
        e["sig"] = sig
        _PACK_CACHE[key] = e
        _WT_BY_PTR[e["W"].data_ptr()] = e["WT"]
        _ENTRY_BY_PTR[e["W"].data_ptr()] = e
    return e


class PackWeightNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, g):
        K = v.shape[1]

        def build():
            with torch.no_grad():
                w, norms = torch._weight_norm_interface(v, g, 0)
                W = pad_cols(w, pad4(K)).contiguous()
                return {"W": W, "WT": transpose_padded(W, K), "norms": norms}
        e = _pack_entry(id(v), _sig(v, g), build, owner=v)
        e["src"] = (weakref.ref(v), weakref.ref(g))
        ctx.save_for_backward(v, g, e["norms"])
        ctx.set_materialize_grads(False)      # deferred mode hands back None for every use: no zero tensor, no weight-norm backward of zeros
        return e["W"].detach()

    @staticmethod
    def backward(ctx, gW):
        if gW is None:
            return None, None
        v, g, norms = ctx.saved_tensors
        gv, gg = torch.ops.aten._weight_norm_interface_backward(gW[:, :v.shape[1]].contiguous(), v, g, norms, 0)
        return gv, gg


class PackPlain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w):
        K = w.shape[1]

        def build():
            with torch.no_grad():
                W = pad_cols(w, pad4(K)).contiguous()
                if W.data_ptr() == w.data_ptr():
                    W = W.clone()
                return {"W": W, "WT": transpose_padded(W, K), "norms": None}
        e = _pack_entry(id(w), _sig(w), build, owner=w)
        e["src"] = (weakref.ref(w),)
        ctx.K = K
        ctx.set_materialize_grads(False)
        return e["W"].detach()

    @staticmethod
    def backward(ctx, gW):
        return None if gW is None else gW[:, :ctx.K]


FUSED_PACK = True       # refresh all stale packs of a network with ONE launch (sr_pack_weights) instead of ~5 torch launches per layer


def refresh_packs(lins):
    """Brings the pack entries of `lins` (the nn.Linear layers of one network) up to date with their parameters: entries that exist
    and are stale (the optimizer stepped) are re-filled IN PLACE by one launch; entries that do not exist yet are left to the
    per-layer torch path of pack_linear (first call only).  In-place refresh keeps every pointer (W, WT, gradient buffers)
    stable across optimizer steps.  A graph built on the old values must be back-propagated before the optimizer step that
    precedes this refresh -- as every training loop does; one that is not gets autograd's in-place error (the version counters of
    the refreshed buffers are bumped after the kernel), never stale weights."""
    if not FUSED_PACK or not lins:
        return
    first = lins[0].weight_v if hasattr(lins[0], "weight_g") else lins[0].weight
    if not first.is_cuda:
        return
    stale = []
    for lin in lins:
        wn = hasattr(lin, "weight_g")
        v, g = (lin.weight_v, lin.weight_g) if wn else (lin.weight, None)
        e = _PACK_CACHE.get(id(v))
        if e is None:
            continue
        sig = _sig(v, g) if wn else _sig(v)
        if e["sig"] != sig:
            if e.get("dirty"):
                raise RuntimeError("mlp_engine: parameters changed while deferred gradients were pending; call flush_param_grads() "
                                   "before the optimizer step")
            if e["W"].shape != (v.shape[0], pad4(v.shape[1])):
                continue                                   # shape changed: let pack_linear rebuild it
            stale.append((e, v, g, sig))
    for i in range(0, len(stale), _lib.SR_PACK_MAX_LAYERS):
        chunk = stale[i:i + _lib.SR_PACK_MAX_LAYERS]
        t = _lib.SrPackTable()
        t.nlayers = len(chunk)
        for j, (e, v, g, sig) in enumerate(chunk):
            L = t.layer[j]
            vv = v.detach()
            if not vv.is_contiguous():
                vv = vv.contiguous()
            L.v, L.g = _lib.ptr(vv), (0 if g is None else _lib.ptr(g.detach().contiguous()))
            L.W, L.WT, L.norms = _lib.ptr(e["W"]), _lib.ptr(e["WT"]), (0 if g is None else _lib.ptr(e["norms"]))
            L.N, L.K, L.ldw, L.ldwt = v.shape[0], v.shape[1], e["W"].stride(0), e["WT"].stride(0)
        with _lib.on_device(chunk[0][1].device), torch.no_grad():
            _lib.call("sr_pack_weights", ctypes.byref(t), _lib.stream_of(chunk[0][1]))
        for e, v, g, sig in chunk:
            e["sig"] = sig
        # The kernel wrote W / WT / norms behind torch's version counters.  Bump them (host-only, no launch): a graph that saved the
        # OLD values -- a retained graph, gradient accumulation over two forwards with an optimizer step in between, a delayed
        # backward -- now raises autograd's "modified by an inplace operation" error instead of silently back-propagating with
        # the new weights.
        torch.autograd.graph.increment_version([t for e, _, _, _ in chunk for t in (e["W"], e["WT"], e["norms"]) if t is not None])


PLAIN_PACKS = True


def pack_linear(lin):
    """Padded effective weight of an nn.Linear, weight-normed (network.py:65-66) or plain."""
    if DEFERRED_PARAM_GRADS and PLAIN_PACKS:
        # Deferred mode: the weight gradients of every use go into the entry's buffers and reach (v, g) through flush_param_grads,
        # so an up-to-date pack is handed out as a plain tensor -- no autograd node per layer and call (~12 us of host time each,
        # ~250 of them per iteration).  reverse() refuses to drop a gradient silently if such a weight ever comes without a sink.
        wn = hasattr(lin, "weight_g")
        v = lin.weight_v if wn else lin.weight
        e = _PACK_CACHE.get(id(v))
        b = lin.bias
        # (a layer whose bias is frozen or not a leaf has no deferred sink -- _deferred_sink keys the buffers on the bias -- so its
        # weight must keep an autograd node, or its gradient would be dropped without an error: take the autograd pack below)
        if (e is not None and e["sig"] == (_sig(v, lin.weight_g) if wn else _sig(v)) and "src" in e
                and b is not None and b.is_leaf and b.requires_grad):
            e["bias_param"] = b
            return e["W"]
    if hasattr(lin, "weight_g"):
        W = PackWeightNorm.apply(lin.weight_v, lin.weight_g)
    else:
        W = PackPlain.apply(lin.weight)
    e = _ENTRY_BY_PTR.get(W.data_ptr())
    if e is not None:
        e["bias_param"] = lin.bias
    return W


def packed_weights_of(module, nlayers):
    """(Ws, bs) of `module.lin0 .. lin{nlayers-1}`: refresh_packs + pack_linear per layer.  While no parameter of the module has changed
    (storage or version) and the packs are handed out as plain tensors (deferred mode), the previous answer is returned: the three networks
    are asked for their packs ~15 times per iteration, ~70 us each -- 1 ms of host time, which is wall time at one frame per rank."""
    d = module.__dict__
    mods = module._modules                                   # (plain dict lookups: nn.Module.__getattr__ is ~10x slower, and a layer or a
    lins = [mods["lin" + str(l)] for l in range(nlayers)]    #  parameter that was REPLACED since the last call must be seen)
    params = []
    for lin in lins:
        P = lin._parameters
        v = P.get('weight_v')
        if v is not None:
            params.append(v); params.append(P['weight_g'])
        else:
            params.append(P['weight'])
        b = P.get('bias')
        if b is not None:
            params.append(b)
    key = (DEFERRED_PARAM_GRADS, PLAIN_PACKS) + tuple((id(p), p.data_ptr(), p._version, p.requires_grad) for p in params)
    hit = d.get('_sr_packs')
    if hit is not None and hit[0] == key:
        return list(hit[1]), list(hit[2])
    refresh_packs(lins)
    Ws, bs = [pack_linear(lin) for lin in lins], [lin.bias for lin in lins]
    if DEFERRED_PARAM_GRADS and PLAIN_PACKS and all(W.grad_fn is None and not W.requires_grad and W.data_ptr() in _ENTRY_BY_PTR for W in Ws):
        d['_sr_packs'] = (key, list(Ws), list(bs))
    else:
        d.pop('_sr_packs', None)
    return Ws, bs


def transposed_of(W, K):
    """W^T for the backward-data GEMM: the packed transpose, or -- for a leading block of a packed weight (the first K input columns
    of a first layer whose per-frame code was hoisted out, the first row of the sdf-only last layer) -- the matching block of it
    (same row pitch; the GEMM masks the contraction tail, so the columns past N inside the 16-byte pad are never used)."""
    WT = _WT_BY_PTR.get(W.data_ptr())
    if WT is not None:
        if WT.shape == (K, pad4(W.shape[0])):
            return WT
        e = _ENTRY_BY_PTR.get(W.data_ptr())
        if (e is not None and e["WT"] is WT and W.stride(0) == e["W"].stride(0) and W.shape[0] <= e["W"].shape[0]
                and K <= WT.shape[0] and pad4(W.shape[0]) <= WT.shape[1]):
            return WT[:K, :pad4(W.shape[0])]
    return transpose_padded(W, K)


# ------------------------------------------------------------------------------------------------
# Deferred parameter gradients (opt-in, used by the training step): every weight-gradient GEMM adds
# into ONE persistent buffer per layer instead of returning a tensor that autograd then pushes through
# a weight-norm backward and an AccumulateGrad add for each of the ~30 network uses per iteration.
# `flush_param_grads()` runs the weight-norm backward once per layer and adds into the parameters'
# .grad; OptimNetwork.propagateTmpPsGrad (the last gradient producer of a step) calls it.
DEFERRED_PARAM_GRADS = False
_ENTRY_BY_PTR = {}


def set_deferred_param_grads(flag):
    global DEFERRED_PARAM_GRADS
    flush_param_grads()
    DEFERRED_PARAM_GRADS = bool(flag)


def _deferred_sink(W, b):
    """-> (dW buffer, db buffer, accumulate) or None.  Both buffers are private to the pack entry; the first weight-gradient GEMM
    after a flush OVERWRITES them (no zero fill), later ones add."""
    e = _ENTRY_BY_PTR.get(W.data_ptr())
    if e is None or b is None or not b.requires_grad or W.shape[0] > e["W"].shape[0] or W.stride(0) != e["W"].stride(0):
        return None
    n, full = W.shape[0], W.shape[0] == e["W"].shape[0]
    if full:
        if not b.is_leaf:
            return None
    else:
        # the leading rows of a layer (the sdf-only evaluation uses row 0 of the last SDF layer, with bias[:1]): same buffers, first
        # rows.  `b` must then be the head of the layer's own bias parameter (registered by pack_linear).
        base = e.get("bias_param")
        if base is None or not base.is_leaf or not base.requires_grad or b.data_ptr() != base.data_ptr():
            return None
        b = base
    if e.get("dW") is None:
        e["dW"] = torch.empty_like(e["W"])
        e["db"] = torch.empty((e["W"].shape[0],), dtype=torch.float32, device=e["W"].device)
        e["fresh"] = True
    if e["fresh"] and not full:                 # a partial first use: the rows it does not touch must read as zero at the flush
        e["dW"].zero_(); e["db"].zero_()
        e["fresh"] = False
    accumulate = not e["fresh"]
    e["fresh"] = False
    e["dirty"] = True
    e["bias"] = b
    return e["dW"][:n], e["db"][:n], accumulate


def flush_param_grads(only=None):
    """`only`: ids of parameter tensors (weight_v / weight) whose layers are flushed now; the rest stays pending.
    All pending layers are turned into parameter gradients by ONE launch (sr_unpack_grads: weight-norm backward / plain copy)."""
    todo = [e for e in _PACK_CACHE.values() if e.get("dirty") and (only is None or id(e["src"][0]()) in only)]
    if not todo:
        return
    join_weight_gradient_stream()
    for i in range(0, len(todo), _lib.SR_PACK_MAX_LAYERS):
        chunk = todo[i:i + _lib.SR_PACK_MAX_LAYERS]
        t = _lib.SrUnpackTable()
        t.nlayers = len(chunk)
        outs = []
        for j, e in enumerate(chunk):
            L = t.layer[j]
            src = tuple(r() for r in e["src"])          # (weak references: an entry must not keep its parameters alive)
            v = src[0]
            g = src[1] if len(src) == 2 else None
            acc = v.grad is not None and (g is None or g.grad is not None)
            gv = v.grad if acc else torch.empty_like(v, memory_format=torch.contiguous_format)
            gg = None
            if g is not None:
                gg = g.grad if acc else torch.empty_like(g, memory_format=torch.contiguous_format)
            if acc and (not gv.is_contiguous() or (gg is not None and not gg.is_contiguous())):
                acc = False                                  # (never the case for gradients made here) fall back to fresh tensors + add
                gv = torch.empty_like(v, memory_format=torch.contiguous_format)
                gg = None if g is None else torch.empty_like(g, memory_format=torch.contiguous_format)
            vd = v.detach()
            if not vd.is_contiguous():
                vd = vd.contiguous()
            L.dW, L.lddw = _lib.ptr(e["dW"]), e["dW"].stride(0)
            L.v, L.g, L.norms = _lib.ptr(vd), (0 if g is None else _lib.ptr(g.detach().contiguous())), (0 if g is None else _lib.ptr(e["norms"]))
            L.gv, L.gg, L.N, L.K, L.accumulate = _lib.ptr(gv), _lib.ptr(gg), v.shape[0], v.shape[1], 1 if acc else 0
            b = e["bias"]
            fold_bias = b.grad is not None and b.grad.is_contiguous() and b.grad.dtype == torch.float32       # else: the buffer is handed over below
            L.db, L.gb = (_lib.ptr(e["db"]), _lib.ptr(b.grad)) if fold_bias else (0, 0)
            outs.append((e, v, g, gv, gg, acc, vd, fold_bias))
        dev = chunk[0]["W"].device
        with _lib.on_device(dev), torch.no_grad():
            _lib.call("sr_unpack_grads", ctypes.byref(t), torch.cuda.current_stream(dev).cuda_stream)
        for e, v, g, gv, gg, acc, _, fold_bias in outs:
            if acc:
                pass                                         # added in place
            else:
                v.grad = gv if v.grad is None else v.grad.add_(gv)
                if g is not None:
                    g.grad = gg if g.grad is None else g.grad.add_(gg)
            b = e["bias"]
            if fold_bias:
                torch.autograd.graph.increment_version(b.grad)   # added in place by the launch above
            elif b.grad is None:                               # hand the buffer over instead of copying it; a new one is made on demand
                b.grad = e["db"]
                e["db"] = torch.empty_like(e["db"])
            else:
                b.grad.add_(e["db"])
            e["fresh"] = True                                  # the next GEMM overwrites the buffers
            e["dirty"] = False


def mlp_apply(spec, A0, Ws, bs):
    return MLPCoreFunction.apply(spec, A0, *Ws, *bs)
