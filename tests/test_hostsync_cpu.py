"""hostsync.nonzero_many (the one round trip of the fused ray selection) on CPU tensors: same lists as mask.nonzero(), riders returned
as ints; and the device-side form of the reference's three ray filters (network.py:519-526) against the filters applied one after the
other, on random images -- the arithmetic OptimNetwork.forward uses, without a GPU."""
import torch

from selfreconcode_amd import hostsync


def test_nonzero_many_equals_nonzero_per_mask():
    g = torch.Generator().manual_seed(3)
    masks = [torch.rand(1000, generator=g) < 0.3, torch.rand(77, generator=g) < 0.9, torch.zeros(10, dtype=torch.bool), torch.ones(5, dtype=torch.bool)]
    lists = hostsync.nonzero_many(masks)
    assert isinstance(lists, list) and all(l.dtype == torch.int64 and l.dim() == 1 for l in lists)
    for l, m in zip(lists, masks):
        assert torch.equal(l, m.nonzero().view(-1))
    lists2, riders = hostsync.nonzero_many(masks[:2], also=[torch.tensor(41), torch.tensor(7, dtype=torch.int32)])
    assert riders == [41, 7] and torch.equal(lists2[1], lists[1])


def _sequential(hit, gt, u, cap):
    b, r, c = hit.nonzero(as_tuple=True)
    sel = (gt[b, r, c] > 0.).nonzero().view(-1)
    b, r, c = b[sel], r[sel], c[sel]
    pnum = b.shape[0]
    if pnum > cap:
        sel = (u[:pnum] < float(cap) / float(pnum)).nonzero().view(-1)
        b, r, c = b[sel], r[sel], c[sel]
    return b, r, c


def _fused(hit, gt, u, cap):
    N, H, W = hit.shape
    flag = (hit & (gt > 0.)).view(-1)
    rank = torch.cumsum(flag, 0) - 1
    pnum = rank[-1] + 1
    thr = torch.where(pnum > cap, (float(cap) / pnum.double()).float(), torch.full((), 2.))
    keep = flag & (u[rank.clamp(min=0, max=u.numel() - 1)] < thr)
    (lin,), (n,) = hostsync.nonzero_many([keep], also=[pnum])
    return (lin // (H * W), (lin // W) % H, lin % W), n


def test_device_form_of_the_three_ray_filters_equals_the_sequential_filters():
    g = torch.Generator().manual_seed(5)
    for trial in range(20):
        N, H, W = 3, 24 + trial, 31
        hit = torch.rand(N, H, W, generator=g) < (0.0 if trial == 0 else 0.55)
        gt = (torch.rand(N, H, W, generator=g) < 0.6).float()
        u = torch.rand(N * H * W, generator=g)
        for cap in (1, 40 * N, 10 ** 6):                 # subsample active (strongly, mildly) and inactive
            want = _sequential(hit, gt, u, cap)
            got, n = _fused(hit, gt, u, cap)
            assert n == int((hit & (gt > 0)).sum())
            for a, b in zip(got, want):
                assert torch.equal(a, b)
