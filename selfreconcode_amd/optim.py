"""Adam for the training step (train.py:139: torch.optim.Adam over the dataset's learnable tensors and the three networks) with the
whole update in ONE kernel launch (csrc/small_ops.hip::adam_step_kernel) instead of torch's multi-tensor chain (~10 launches per step
with ~40 us of host time between them: 9 ms of GPU idle per 230 launches in the round-2 trace).  Same update rule, same state layout
(`step`, `exp_avg`, `exp_avg_sq` per parameter: a torch.optim.Adam state_dict loads, and this one's loads into torch.optim.Adam), same
`param_groups` (the learning-rate schedulers of torch work on it unchanged).  weight_decay / amsgrad / maximize are not implemented
(the reference uses none of them) and raise."""
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
        for group in self.param_groups:
            b1, b2 = group['betas']
            todo = []
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or p.dtype != torch.float32 or not p.is_cuda:
                    raise RuntimeError("FusedAdam: dense float32 GPU parameters only")
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0)                      # host tensor, as torch.optim.Adam keeps it (capturable=False)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st['step'] += 1
                if not p.is_contiguous():
                    raise RuntimeError("FusedAdam: parameters must be contiguous")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                todo.append((p, g, st))
            for i in range(0, len(todo), _lib.SR_ADAM_MAX_TENSORS):
                chunk = todo[i:i + _lib.SR_ADAM_MAX_TENSORS]
                t = _lib.SrAdamTable()
                t.ntensors, t.beta1, t.beta2, t.eps = len(chunk), b1, b2, group['eps']
                t.one_minus_beta1, t.one_minus_beta2 = 1.0 - b1, 1.0 - b2             # (double arithmetic, then rounded once)
                for j, (p, g, st) in enumerate(chunk):
                    k = float(st['step'])
                    T = t.tensor[j]
                    T.p, T.g, T.m, T.v, T.numel = _lib.ptr(p), _lib.ptr(g), _lib.ptr(st['exp_avg']), _lib.ptr(st['exp_avg_sq']), p.numel()
                    T.lr, T.bias1, T.inv_sqrt_bias2 = group['lr'], 1.0 - b1 ** k, 1.0 / math.sqrt(1.0 - b2 ** k)
                dev = chunk[0][0].device
                with torch.cuda.device(dev):
                    _lib.call("sr_adam_step", ctypes.byref(t), torch.cuda.current_stream(dev).cuda_stream)
                for p, _, _ in chunk:
                    torch.autograd.graph.increment_version(p)           # the kernel wrote the parameter behind torch's version counter
        return loss
