"""Positional encoding -- mirrors model/Embedder.py:4-54 (get_embedder / Embedder.embed) on the fused
HIP embed kernel.  Band order [x | sin f0 | cos f0 | sin f1 | ...], bands 2^k, per-band weights
in (sin, cos) pairs from utils.annealing_weights (utils/utils.py:40-46)."""
import torch
from .. import _lib
from .. import mlp_engine
from ..mlp_engine import pad4

_WCACHE = {}


def band_weight_tensor(ws, multires, device):
    """2*L floats on the device; cached per (weights, device).  A new set of weights (the annealing ratio moves every iteration while it
    ramps, train.py:158-160) goes up through PINNED memory without blocking: `torch.tensor(values, device=cuda)` copies from pageable
    memory, and that call does not return before the current stream has drained (measured: 30 ms behind 30 ms of queued GEMMs, 0.01 ms
    for a pinned non-blocking copy) -- at the first deformer call of an iteration it made the host wait for the whole previous iteration
    and left the main stream idle until the host had issued again (round 6)."""
    if ws is None:
        ws = (1.0,) * (2 * multires)
    key = (tuple(float(w) for w in ws), str(device))
    t = _WCACHE.get(key)
    if t is None:
        if len(_WCACHE) > 4096:              # (one entry per distinct ratio: a ramp of thousands of iterations must not grow without bound)
            _WCACHE.clear()
        host = torch.tensor(key[0], dtype=torch.float32)
        t = host.pin_memory().to(device, non_blocking=True) if torch.device(device).type == "cuda" else host.to(device)
        _WCACHE[key] = t
    return t, key[0]


_FCACHE = {}


def _band_frequencies(L, device, dtype):
    key = (L, str(device), dtype)
    f = _FCACHE.get(key)
    if f is None:
        f = _FCACHE[key] = (2.0 ** torch.arange(L, device=device, dtype=dtype)).view(1, L, 1)
    return f


class PEFunction(torch.autograd.Function):
    """A0[P, pad4(3+6L+E)] = [x | PE_L(x) | extra[index] | 0].  The backward is written with
    differentiable torch ops on the saved OUTPUT (d sin = f cos, d cos = -f sin are again columns
    of A0), so any derivative order works (the one-kernel shortcuts are taken only when grad mode is off)."""

    @staticmethod
    def forward(ctx, x, wt, L, extra, extra_index, segment=0):
        _lib.require_gpu(x)
        x = x.contiguous().float()
        P = x.shape[0]
        E = 0 if extra is None else extra.shape[1]
        ldo = pad4(3 + 6 * L + E)
        out = torch.empty((P, ldo), dtype=torch.float32, device=x.device)
        ex = None if extra is None else extra.contiguous().float()
        with _lib.on_device(x.device):
            _lib.call("sr_pe_embed", _lib.ptr(x), P, L, _lib.ptr(wt), _lib.ptr(ex), 0 if ex is None else ex.stride(0), E,
                      _lib.ptr(extra_index), 1, _lib.ptr(out), ldo, _lib.stream_of(x))
        ctx.L, ctx.E, ctx.segment = L, E, segment
        ctx.n_extra = 0 if extra is None else extra.shape[0]
        ctx.wt = wt
        ctx.save_for_backward(out, extra_index, x)
        return out

    @staticmethod
    def backward(ctx, g):
        out, extra_index, x = ctx.saved_tensors
        L, E = ctx.L, ctx.E
        P = out.shape[0]
        if not torch.is_grad_enabled() and ctx.needs_input_grad[0]:
            # plain first-order backward: one kernel instead of ~10 elementwise launches (the composite below is kept for
            # create_graph=True, where the backward itself must be differentiable)
            g = g.contiguous()
            gx = torch.empty_like(x)
            with _lib.on_device(x.device):
                _lib.call("sr_pe_embed_bwd", _lib.ptr(x), P, L, _lib.ptr(ctx.wt), 1, _lib.ptr(g), g.stride(0), _lib.ptr(gx), _lib.stream_of(x))
        elif not ctx.needs_input_grad[0]:
            gx = None
        else:
            gx = g[:, :3]
        if L > 0 and gx is not None and torch.is_grad_enabled():
            Eb = out[:, 3:3 + 6 * L].reshape(P, L, 2, 3)
            gb = g[:, 3:3 + 6 * L].reshape(P, L, 2, 3)
            f = _band_frequencies(L, out.device, out.dtype)
            gx = gx + ((Eb[:, :, 1] * gb[:, :, 0] - Eb[:, :, 0] * gb[:, :, 1]) * f).sum(1)
        gextra = None
        if E > 0 and ctx.needs_input_grad[3]:
            ge = g[:, 3 + 6 * L:3 + 6 * L + E]
            if extra_index is None:
                gextra = ge
            elif ctx.segment:                       # rows of one frame are contiguous: plain segmented sum, no atomics
                gextra = ge.reshape(ctx.n_extra, ctx.segment, E).sum(1)
            else:
                # raw kernel (deterministic fold) only when nothing differentiates this backward again: under create_graph the
                # per-frame code's gradient must keep its grad_fn (index_add is differentiable)
                gextra = (mlp_engine.rows_frame_sum(ge, extra_index, ctx.n_extra).to(g.dtype)
                          if (g.is_cuda and ctx.n_extra <= 32 and not torch.is_grad_enabled())
                          else torch.zeros((ctx.n_extra, E), dtype=g.dtype, device=g.device).index_add(0, extra_index, ge))
        return gx, None, None, gextra, None, None


def embed_rows(x, multires, ws=None, extra=None, extra_index=None, segment=0):
    """First-layer input rows [P, pad4(3 + 6*multires + E)] (fused PE + concat)."""
    if ws is not None and any(float(ws[2 * k]) != float(ws[2 * k + 1]) for k in range(multires)):
        raise NotImplementedError("fused embedder needs equal (sin, cos) weights per band, as utils.annealing_weights yields")
    wt, _ = band_weight_tensor(ws, multires, x.device)
    return PEFunction.apply(x, wt, multires, extra, extra_index, segment)


class Embedder:
    """API mirror of the reference class (kwargs, out_dim, embed(inputs, ws))."""

    def __init__(self, **kwargs):
        self.kwargs = kwargs
        assert kwargs['include_input'] and kwargs['input_dims'] == 3 and kwargs['log_sampling']
        assert kwargs['max_freq_log2'] == kwargs['num_freqs'] - 1
        self.multires = kwargs['num_freqs']
        self.out_dim = 3 + 6 * self.multires

    def embed(self, inputs, ws=None):
        shp = inputs.shape
        out = embed_rows(inputs.reshape(-1, 3), self.multires, ws)[:, :self.out_dim]
        return out.reshape(*shp[:-1], self.out_dim)


def get_embedder(multires):
    embedder_obj = Embedder(include_input=True, input_dims=3, max_freq_log2=multires - 1, num_freqs=multires,
                            log_sampling=True, periodic_fns=[torch.sin, torch.cos])

    def embed(x, ws=None, eo=embedder_obj):
        return eo.embed(x, ws)
    return embed, embedder_obj.out_dim
