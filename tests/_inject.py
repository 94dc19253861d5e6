"""Shared by the iteration-level GPU tests: replace the refiner's output PER PIXEL.

The refiner's acceptance test |f| < 5e-5 flips on single ulps, so tests that compare what comes AFTER it feed both sides the same
refiner output.  When the two sides selected exactly the same rays that is `rand['refined']`; when the selections may differ by a ray
(perturbed inputs, a trajectory a few optimizer steps in) the override is matched by (frame, row, column): the product's own refiner
still runs on every ray, and the rays the other side also has take the other side's (point, flag)."""
import contextlib
import torch


@contextlib.contextmanager
def keyed_refiner(state, H, W):
    """`state`: dict the caller fills before every forward with  dbg (the `debug=` dict handed to forward),  ref = (frame [R], row [R],
    col [R], points [R,3], flags [R]) or None;  after the forward state['matched'] = (#overridden, #rays here, #rays there)."""
    from selfreconcode_amd.model import optim_network as onet
    real = onet.OptimizeSurfacePs

    def refiner(cam_pos, rays, p0, bi, *a, **kw):
        p, ok = real(cam_pos, rays, p0, bi, *a, **kw)
        ref, dbg = state.get('ref'), state['dbg']
        if ref is None:
            state['matched'] = (0, int(ok.numel()), 0)
            return p, ok
        dev = p.device
        key = (dbg['batch_inds'] * H + dbg['row_inds']) * W + dbg['col_inds']
        rkey = ((ref[0].long() * H + ref[1].long()) * W + ref[2].long()).to(dev)
        order = torch.argsort(rkey)
        pos = torch.searchsorted(rkey[order], key).clamp(max=max(rkey.numel() - 1, 0))
        hit = rkey[order][pos] == key if rkey.numel() else torch.zeros_like(key, dtype=torch.bool)
        src = order[pos[hit]]
        p, ok = p.clone(), ok.clone()
        p[hit] = ref[3].to(dev).float()[src]; ok[hit] = ref[4].to(dev).bool()[src]
        state['matched'] = (int(hit.sum()), int(key.numel()), int(rkey.numel()))
        return p, ok
    onet.OptimizeSurfacePs = refiner
    try:
        yield state
    finally:
        onet.OptimizeSurfacePs = real
