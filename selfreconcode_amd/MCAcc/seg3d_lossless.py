"""Coarse-to-fine "lossless" SDF volume evaluation -- drop-in for MCAcc/seg3d_lossless.py::Seg3dLossless
(_forward :233-428, batch_eval :89-108) in the configuration the reference ships
(use_cuda_impl=False, faster=False, align_corners=False, one channel, batch 1).

Same results, GPU-friendlier bookkeeping: the reference tracks evaluated voxels as a growing
coordinate list that it re-sorts with unique(dim=1) at every step; here that set is a boolean volume
per level (upsampled by striding), the 3x3x3 dilation is a max-pool, and conflict neighbourhoods
are scattered into a mask instead of sorted.  Every query goes through `query_func` exactly as in
the reference ([1,M,3] world points -> [1,1,M]); with this package's SDF network that is the fused
no-grad MFMA forward.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def create_grid3D(min, max, steps, device="cuda:0"):
    """MCAcc/utils.py:88-101 -- integer lattice coordinates [N,3] as (x,y,z), z slowest."""
    if type(min) is int:
        min = (min, min, min)
    if type(max) is int:
        max = (max, max, max)
    if type(steps) is int:
        steps = (steps, steps, steps)
    ax = [torch.linspace(min[i], max[i], steps[i]).long().to(device) for i in range(3)]
    gD, gH, gW = torch.meshgrid([ax[2], ax[1], ax[0]], indexing='ij')
    return torch.stack([gW, gH, gD]).view(3, -1).t()


class Seg3dLossless(nn.Module):
    def __init__(self, query_func, b_min, b_max, resolutions, channels=1, balance_value=0.5, align_corners=False,
                 visualize=False, debug=False, use_cuda_impl=False, faster=False, use_shadow=False, **kwargs):
        super().__init__()
        self.query_func = query_func
        # (host copies: utils.set_hierarchical_config hands over the previous engine's box, which lives on the GPU; the spacings below
        # are host scalars and `.to(device)` moves the buffers afterwards)
        self.register_buffer('b_min', torch.as_tensor(b_min).detach().float().cpu().view(1, 1, 3).clone())
        self.register_buffer('b_max', torch.as_tensor(b_max).detach().float().cpu().view(1, 1, 3).clone())
        if type(resolutions[0]) is int:
            resolutions = torch.tensor([(r, r, r) for r in resolutions])
        else:
            resolutions = torch.tensor(resolutions)
        self.register_buffer('resolutions', resolutions)
        self._res = [tuple(int(v) for v in r) for r in resolutions.tolist()]            # (W,H,D) per level, host copy
        tmp = (self.b_max.view(3) - self.b_min.view(3)) / self.resolutions[-1].view(3).float()
        self.spacing_x, self.spacing_y, self.spacing_z = tmp[0].item(), tmp[1].item(), tmp[2].item()
        self.bx = self.b_min.view(-1)[0].item() + self.spacing_x / 2.       # voxel-centre convention (:38-44)
        self.by = self.b_min.view(-1)[1].item() + self.spacing_y / 2.
        self.bz = self.b_min.view(-1)[2].item() + self.spacing_z / 2.
        assert self.b_min.size(0) == 1 and channels == 1 and align_corners is False and visualize is False
        assert not faster and not use_shadow, "faster / shadow modes of the reference are not built (never used by it)"
        self.use_cuda_impl = use_cuda_impl          # fused HIP upsample + boundary flag (the reference ships it switched off)
        self.balance_value = balance_value
        for r in self._res:
            assert r[0] % 2 == 1 and r[1] % 2 == 1, f"resolution {r} need to be odd becuase of align_corner."
        self.stats = {}

    def batch_eval(self, coords, **kwargs):
        """coords [1,M,3] integer (x,y,z) at the finest resolution -> query_func at voxel centres."""
        last = self.resolutions[-1].to(coords.device)
        step = 1.0 / last.float()
        c = coords.float() / last + step / 2
        c = c * (self.b_max - self.b_min) + self.b_min
        occ = self.query_func(**kwargs, points=c)
        if type(occ) is list:
            occ = torch.stack(occ)
        assert occ.dim() == 3, "query_func should return a occupancy with shape of [bz, C, N]"
        self.stats['queries'] = self.stats.get('queries', 0) + coords.shape[1]
        return occ

    def forward(self, **kwargs):
        return self._forward(**kwargs)

    def _forward(self, **kwargs):
        dev = self.b_min.device
        bal = self.balance_value
        self.stats = {}
        Wf, Hf, Df = self._res[-1]
        calculated = torch.zeros((Df, Hf, Wf), dtype=torch.bool, device=dev)      # at the finest lattice
        occ, done = None, None
        for lvl, (W, H, D) in enumerate(self._res):
            stride = tuple((f - 1) // (r - 1) for f, r in zip(self._res[-1], (W, H, D)))
            st = torch.tensor(stride, device=dev)
            if lvl == 0:
                coords = create_grid3D((0, 0, 0), (Wf - 1, Hf - 1, Df - 1), (W, H, D), device=dev).unsqueeze(0)
                occ = self.batch_eval(coords, **kwargs).view(1, 1, D, H, W)
                done = torch.ones((D, H, W), dtype=torch.bool, device=dev)
                calculated[coords[0, :, 2], coords[0, :, 1], coords[0, :, 0]] = True
                continue
            nd = torch.zeros((D, H, W), dtype=torch.bool, device=dev)
            nd[::2, ::2, ::2] = done                                             # evaluated voxels carry over (coords_accum *= 2)
            done = nd
            if self.use_cuda_impl:
                # fused HIP path: 2x upsample + "parents disagree" flag (K10), then dilation + exclusion + compaction in one pass
                from ..ext import interp2x_boundary3d
                from .. import _lib
                occ, bflag = interp2x_boundary3d.forward(occ.float().contiguous(), bal)
                cand = torch.empty(D * H * W, dtype=torch.int64, device=dev)
                cnt = torch.empty(1, dtype=torch.int64, device=dev)
                with _lib.on_device(dev):
                    _lib.call("sr_seg3d_candidates", _lib.ptr(bflag), _lib.ptr(done), D, H, W, _lib.ptr(cand), _lib.ptr(cnt), _lib.stream_of(occ))
                idx = cand[:int(cnt)].sort().values                             # (one host sync, as nonzero; sorted = the reference's order)
            else:
                valid = F.interpolate((occ > bal).float(), size=(D, H, W), mode="trilinear", align_corners=True)
                occ = F.interpolate(occ.float(), size=(D, H, W), mode="trilinear", align_corners=True)
                boundary = ((valid > 0.0) & (valid < 1.0)).float()
                boundary = F.max_pool3d(boundary, kernel_size=3, stride=1, padding=1)[0, 0] > 0     # == smooth_conv3x3(.) > 0
                boundary &= ~done
                idx = boundary.view(-1).nonzero(as_tuple=False).view(-1)         # flat index z*H*W + y*W + x
            if idx.numel() == 0:
                continue
            flat = occ.view(-1)
            interp = flat[idx]
            pc = torch.stack([idx % W, (idx // W) % H, idx // (W * H)], dim=1)   # (x,y,z) at this level
            coords = (pc * st).unsqueeze(0)
            val = self.batch_eval(coords, **kwargs).view(-1)
            flat[idx] = val
            done.view(-1)[idx] = True
            calculated[coords[0, :, 2], coords[0, :, 1], coords[0, :, 0]] = True
            conflicts = (interp - bal) * (val - bal) < 0
            offs = torch.stack(torch.meshgrid([torch.tensor([-1, 0, 1], device=dev)] * 3, indexing='ij')).view(3, -1).t()
            while bool(conflicts.any()):
                cc = coords[0, conflicts, :]
                nb = (cc.unsqueeze(1) + (offs * st).unsqueeze(0)).reshape(-1, 3)
                nb[:, 0].clamp_(0, Wf - 1); nb[:, 1].clamp_(0, Hf - 1); nb[:, 2].clamp_(0, Df - 1)
                cand = torch.zeros_like(calculated)
                cand[nb[:, 2], nb[:, 1], nb[:, 0]] = True
                cand &= ~calculated
                fidx = cand.view(-1).nonzero(as_tuple=False).view(-1)
                if fidx.numel() == 0:
                    break
                fc = torch.stack([fidx % Wf, (fidx // Wf) % Hf, fidx // (Wf * Hf)], dim=1)
                pc = fc // st                                                    # reference: point_coords = coords // stride
                idx = pc[:, 2] * H * W + pc[:, 1] * W + pc[:, 0]
                coords = (pc * st).unsqueeze(0)
                interp = flat[idx]
                val = self.batch_eval(coords, **kwargs).view(-1)
                conflicts = (interp - bal) * (val - bal) < 0
                flat[idx] = val
                done.view(-1)[idx] = True
                calculated[coords[0, :, 2], coords[0, :, 1], coords[0, :, 0]] = True
        return occ
