"""Adam for the training step (train.py:139: torch.optim.Adam over the dataset's learnable tensors and the three networks) with the
whole update in ONE kernel launch (csrc/small_ops.hip::adam_step_kernel) instead of torch's multi-tensor chain (~10 launches per step
with ~40 us of host time between them: 9 ms of GPU idle per 230 launches in the round-2 trace).  Same update rule, same state layout
(`step`, `exp_avg`, `exp_avg_sq` per parameter: a torch.optim.Adam state_dict loads, and this one's loads into torch.optim.Adam), same
`param_groups` (the learning-rate schedulers of torch work on it unchanged).  weight_decay / amsgrad / maximize are not implemented
(the reference uses none of them) and raise.  Inside the optimizer `step` is a Python float (torch keeps a host tensor and pays one
aten `add_` per parameter and step: ~60 operator calls per iteration here); `state_dict()` writes it out as the host tensor torch's Adam
keeps, `load_state_dict()` takes either form."""
import ctypes
import math

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0 or amsgrad:
            raise NotImplementedError("FusedAdam: weight_decay / amsgrad are not implemented (train.py:139 uses neither)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None, capturable=False,
                                      differentiable=False, fused=None))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # ONE table over all parameter groups (the learning rate and the bias corrections are per tensor): one launch per 64 tensors.
        # The table is cached: between two steps only the gradient pointers, the learning rates and the step-dependent scalars can
        # change, so a step rewrites those fields of the ctypes structure and nothing else (~60 tensors: 190 us -> 40 us of host time
        # at a point of the iteration where the GPU has nothing else queued).
        todo = []
        for group in self.param_groups:
            b1, b2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                        raise RuntimeError("FusedAdam: dense contiguous float32 GPU parameters only")
                    st['step'] = 0.0
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['step'] = float(st['step']) + 1.0                    # (a loaded state may hold torch's host tensor)
                todo.append((p, p.grad if p.grad.is_contiguous() else p.grad.contiguous(), st, group))
        if not todo:
            return loss
        betas, eps = todo[0][3]['betas'], todo[0][3]['eps']
        if any(t[3]['betas'] != betas or t[3]['eps'] != eps for t in todo):
            raise NotImplementedError("FusedAdam: all parameter groups must share betas and eps")
        dev = todo[0][0].device
        if any(p.device != dev for p, _, _, _ in todo):
            raise RuntimeError("FusedAdam: all parameters must live on one device")
        # (the table holds raw pointers: a parameter whose storage was swapped under the same Parameter object -- p.data = ...,
        # module.to() -- or whose state was reloaded must rebuild it)
        key = tuple((id(p), p.data_ptr(), p.numel(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()) for p, _, st, _ in todo)
        cache = getattr(self, '_tables', None)
        if cache is None or cache[0] != key:
            tables = []
            for i in range(0, len(todo), _lib.SR_ADAM_MAX_TENSORS):
                chunk = todo[i:i + _lib.SR_ADAM_MAX_TENSORS]
                t = _lib.SrAdamTable()
                t.ntensors, t.beta1, t.beta2, t.eps = len(chunk), betas[0], betas[1], eps
                t.one_minus_beta1, t.one_minus_beta2 = 1.0 - betas[0], 1.0 - betas[1]             # (double arithmetic, then rounded once)
                for j, (p, g, st, group) in enumerate(chunk):
                    T = t.tensor[j]
                    T.p, T.m, T.v, T.numel = _lib.ptr(p), _lib.ptr(st['exp_avg']), _lib.ptr(st['exp_avg_sq']), p.numel()
                tables.append(t)
            cache = self._tables = (key, tables)
        b1, b2 = betas
        corr = {}
        for i, (p, g, st, group) in enumerate(todo):
            T = cache[1][i // _lib.SR_ADAM_MAX_TENSORS].tensor[i % _lib.SR_ADAM_MAX_TENSORS]
            k = float(st['step'])
            c = corr.get(k)
            if c is None:
                c = corr[k] = (1.0 - b1 ** k, 1.0 / math.sqrt(1.0 - b2 ** k))
            T.g, T.lr, T.bias1, T.inv_sqrt_bias2 = g.data_ptr(), group['lr'], c[0], c[1]
        with _lib.on_device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            for t in cache[1]:
                _lib.call("sr_adam_step", ctypes.byref(t), stream)
        torch.autograd.graph.increment_version([p for p, _, _, _ in todo])      # the kernel wrote the parameters behind torch's version counters
        return loss

    def state_dict(self):
        sd = super().state_dict()
        # (copies of the per-parameter dicts: torch hands out the optimizer's own) with `step` in the form torch.optim.Adam saves (capturable=False)
        sd['state'] = {k: dict(st, step=torch.tensor(float(st['step']))) if 'step' in st else dict(st) for k, st in sd['state'].items()}
        return sd
