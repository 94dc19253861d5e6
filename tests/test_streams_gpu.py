"""Stream plumbing of the iteration: the device-flag ordering primitives (sr_stream_flag_set / _wait), the device-clock stamp and
the polling count round trip (hostsync.nonzero) against torch's own nonzero."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_traced_nonzero_equals_torch_nonzero(monkeypatch):
    """The traced form of the iteration's round trips (tools/host_profile.py: count copy + event polling + nonzero_static) returns what
    torch's nonzero returns."""
    from selfreconcode_amd import hostsync
    seen = []
    monkeypatch.setattr(hostsync, 'TRACE', seen.append)
    g = torch.Generator(device=DEV); g.manual_seed(3)
    for shape in [(0,), (1,), (4097,), (3, 37, 41), (2, 5, 0)]:
        m = torch.rand(shape, device=DEV, generator=g) < 0.3
        a, b = hostsync.nonzero(m), m.nonzero()
        assert a.dtype == b.dtype and torch.equal(a, b)
        ta, tb = hostsync.nonzero(m, as_tuple=True), m.nonzero(as_tuple=True)
        assert len(ta) == len(tb) and all(torch.equal(x, y) for x, y in zip(ta, tb))
    none = torch.zeros(100, dtype=torch.bool, device=DEV)
    assert hostsync.nonzero(none).shape == (0, 1)
    side = torch.cuda.Stream(device=DEV)               # on a side stream: its own pinned slot, result ordered on that stream
    m = torch.rand(10000, device=DEV, generator=g) < 0.5
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        idx = hostsync.nonzero(m).view(-1)
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(idx, m.nonzero().view(-1))
    assert 'count copy issued' in seen and 'count on the host' in seen
    lists, (extra,) = hostsync.nonzero_many([m, ~m], also=[m.sum()])
    assert torch.equal(lists[0], m.nonzero().view(-1)) and torch.equal(lists[1], (~m).nonzero().view(-1)) and extra == int(m.sum())


def test_device_flag_orders_a_side_stream_behind_the_main_stream():
    from selfreconcode_amd import _lib
    flag = torch.zeros(2, dtype=torch.int32, device=DEV)            # [0] flag, [1] time-out counter
    main, side = torch.cuda.current_stream(), torch.cuda.Stream(device=DEV)
    x = torch.zeros(1 << 20, device=DEV)
    A = torch.randn(4096, 4096, device=DEV)
    torch.cuda.synchronize()
    for epoch in (1, 2, 3):
        # side: wait for the flag, then read x; main: long work, write x, set the flag.  The side stream is issued FIRST.
        with torch.cuda.stream(side):
            _lib.call("sr_stream_flag_wait", flag.data_ptr(), epoch, flag.data_ptr() + 4, 5000, side.cuda_stream)
            seen = x[:4].clone()
        for _ in range(5):
            A @ A
        x.fill_(float(epoch))
        _lib.call("sr_stream_flag_set", flag.data_ptr(), epoch, main.cuda_stream)
        torch.cuda.synchronize()
        assert seen.tolist() == [float(epoch)] * 4
    assert int(flag[1]) == 0
    # a wait that can never be satisfied gives up after its time-out and says so
    _lib.call("sr_stream_flag_wait", flag.data_ptr(), 1000, flag.data_ptr() + 4, 20, main.cuda_stream)
    torch.cuda.synchronize()
    assert int(flag[1]) == 1
    with pytest.raises(_lib.SrError):
        _lib.call("sr_stream_flag_wait", 0, 1, 0, 10, main.cuda_stream)


def test_device_clock_stamps_are_monotone_across_streams():
    from selfreconcode_amd import _lib
    st = torch.zeros(3, dtype=torch.int64, device=DEV)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream(device=DEV)
    A = torch.randn(4096, 4096, device=DEV)
    torch.cuda.synchronize()
    _lib.call("sr_stream_stamp", st.data_ptr(), main.cuda_stream)
    for _ in range(4):
        A @ A
    _lib.call("sr_stream_stamp", st.data_ptr() + 8, main.cuda_stream)
    side.wait_stream(main)
    _lib.call("sr_stream_stamp", st.data_ptr() + 16, side.cuda_stream)
    torch.cuda.synchronize()
    a, b, c = st.tolist()
    assert 0 < a < b <= c
    assert 1e2 < (b - a) < 1e8          # 100 MHz ticks: the four products take between a microsecond and a second
