"""The ray selection of the iteration with ONE host round trip (optim_network.FUSED_SELECTION) against the sequential form that mirrors
network.py:519-526 filter by filter (FindSurfacePs -> inside the ground-truth mask -> Bernoulli subsample) and utils.py:74-84's vertex
subsets: same rays in the same order, same seeds, same vertex subsets, and therefore the same iteration, bit for bit -- on the bench's
scene at full size (coarse stage, 540 x 540, 3 frames x 2048 rays), with the subsample active and with it inactive."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RATIO = {'sdfRatio': 1., 'deformerRatio': 0.6, 'renderRatio': 1.}


def _iteration(fused, sample_pix, rounds):
    from selfreconcode_amd.model import optim_network as on
    from selfreconcode_amd import hostsync
    from selfreconcode_amd.synthetic import build_synthetic_scene
    net, ds, conf = build_synthetic_scene(device=DEV, frame_num=64, stage='coarse', consistent_masks=False)
    ds.attach_rendered_observations(net, RATIO)
    f = torch.arange(2, 5, device=DEV)
    g = torch.Generator(device=DEV); g.manual_seed(11)
    V_max, P_max = 200_000, 3 * 540 * 540
    rand = {'ray_select': torch.rand(P_max, device=DEV, generator=g), 'vert_select': torch.rand(V_max, device=DEV, generator=g),
            'vert_select2': torch.rand(V_max, device=DEV, generator=g), 'eik_local': torch.randn(400_000, 3, device=DEV, generator=g),
            'eik_global': torch.rand(80_000, 3, device=DEV, generator=g), 'regu_local': torch.randn(400_000, 3, device=DEV, generator=g)}
    trips = []
    prev, hostsync.TRACE = hostsync.TRACE, (lambda label: trips.append(label))
    saved, on.FUSED_SELECTION = on.FUSED_SELECTION, fused
    try:
        dbg = {}
        loss = net(ds.batch(f), sample_pix, RATIO, f, rand=rand, debug=dbg)
        loss.backward()
        net.propagateTmpPsGrad(f, RATIO)
        torch.cuda.synchronize()
    finally:
        on.FUSED_SELECTION, hostsync.TRACE = saved, prev
    assert trips.count('count on the host') == rounds, trips
    grads = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.grad is not None])
    return dbg, loss.detach().clone(), net._eik_pts.clone(), grads, dict(net.info)


@pytest.mark.parametrize("sample_pix", [2048, 1 << 20])
def test_one_round_trip_selection_equals_the_sequential_filters(sample_pix):
    a = _iteration(True, sample_pix, rounds=2)            # selection + vertex subsets; converged rays
    b = _iteration(False, sample_pix, rounds=6 if sample_pix == 2048 else 5)
    da, db = a[0], b[0]
    n = da['batch_inds'].numel()
    assert n > 1000 and (n < 3 * 2048 * 1.2 if sample_pix == 2048 else n > 3 * 2048 * 2)
    for k in ('batch_inds', 'row_inds', 'col_inds', 'seeds', 'initTmpPs', 'check'):
        assert torch.equal(da[k], db[k]), k
    assert torch.equal(a[2], b[2])                        # eikonal sample points: [rays ; vertex subset ; uniform] -- the vertex subset too
    assert torch.equal(a[1], b[1])                        # the loss ...
    assert torch.equal(a[3], b[3])                        # ... and every network gradient of the iteration
    assert float(a[4]['def_loss']) == float(b[4]['def_loss'])
