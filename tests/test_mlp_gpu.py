"""GPU parity of the three MLP drop-ins (SDF a2/a3, deformer a4, render a10) and of the raw
tangent-interleaved engine, against the reference-generated golden vectors and the CPU oracle.
Tolerances: fp32 MFMA accumulates in a different order than the CPU GEMM -> values to 2e-5
relative / 2e-6 absolute, first derivatives to 1e-4, second-order parameter gradients to 1e-3."""
import pytest
import torch
from oracle import torch_oracle as orc
from oracle import fixtures as fx

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RATIO = {'sdfRatio': 1.0, 'deformerRatio': 0.62, 'renderRatio': 1.0}


def close(a, b, rtol=2e-5, atol=2e-6):
    torch.testing.assert_close(a.detach().float().cpu(), b.detach().float().cpu(), rtol=rtol, atol=atol)


def _sdf(seed=101):
    from selfreconcode_amd.model.network import getTmpSdf
    net = getTmpSdf(DEV, 6, 0.6, 256)
    net.load_state_dict(fx.det_params(fx.SDF_SPEC, seed), strict=True)
    return net


def test_sdf_golden_forward_gradient_eikonal(golden):
    g = golden("sdf")
    net = _sdf()
    for tag, ratio in [("r1", 1.0), ("r04", 0.4), ("dict", {'sdfRatio': 1.0, 'deformerRatio': 0.7, 'renderRatio': 1.0})]:
        x = g["x"].to(DEV).requires_grad_(True)
        y = net(x, ratio)
        assert y.shape == (48, 1) and net.rendcond.shape == (48, 256)
        gr = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True)[0]
        close(y, g["sdf_" + tag]); close(net.rendcond[:, ::16], g["rend_" + tag])
        close(gr, g["grad_" + tag], 1e-4, 1e-5)
        if tag == "r1":
            eik = ((gr.norm(2, dim=-1) - 1) ** 2).mean()
            pg = torch.autograd.grad(eik, [net.lin0.weight_v, net.lin4.weight_g, net.lin7.bias, x])
            close(eik, g["eik"], 1e-4)
            close(pg[0][::37, ::5], g["eik_dv0"], 1e-3, 1e-5)
            close(pg[1], g["eik_dg4"], 1e-3, 1e-5)
            close(pg[2], g["eik_db7"], 1e-3, 1e-5)
            close(pg[3], g["eik_dx"], 1e-3, 1e-5)


@pytest.mark.parametrize("P", [1, 7, 129, 1000])
def test_sdf_vs_oracle_ragged_sizes_and_param_grads(P):
    net = _sdf(55)
    sd = {k: v.clone().requires_grad_(True) for k, v in fx.det_params(fx.SDF_SPEC, 55).items()}
    x = fx.det_tensor((P, 3), 900 + P, 0.9)
    go = fx.det_tensor((P, 257), 901, 1.0)
    xr = x.clone().requires_grad_(True)
    yo, rc = orc.sdf_forward(sd, xr, 0.7)
    lo = (torch.cat([yo, rc], 1) * go).sum()
    ref = torch.autograd.grad(lo, [xr, sd["lin3.weight_v"], sd["lin3.weight_g"], sd["lin3.bias"], sd["lin8.weight_v"], sd["lin0.weight_v"]])
    xg = x.to(DEV).requires_grad_(True)
    y = net(xg, 0.7)
    l = (torch.cat([y, net.rendcond], 1) * go.to(DEV)).sum()
    ours = torch.autograd.grad(l, [xg, net.lin3.weight_v, net.lin3.weight_g, net.lin3.bias, net.lin8.weight_v, net.lin0.weight_v])
    close(y, yo); close(net.rendcond, rc)
    for a, b in zip(ours, ref):
        close(a, b, 2e-4, 2e-5 * max(1.0, float(b.abs().max())))


def test_sdf_empty_batch():
    net = _sdf()
    y = net(torch.zeros(0, 3, device=DEV), 1.0)
    assert y.shape == (0, 1)


def test_translator_golden(golden):
    from selfreconcode_amd.model.Deformer import MLPTranslator
    from selfreconcode_amd.utils import compute_Jacobian
    g = golden("translator")
    tr = MLPTranslator(128, 6).to(DEV)
    tr.load_state_dict(fx.det_params(fx.DEF_SPEC, 202), strict=True)
    y = tr(g["ps"].to(DEV), g["conds"].to(DEV), g["bi"].to(DEV), ratio=RATIO)
    close(y, g["y"]); close(tr.offset, g["off"])
    yb = tr(g["psb"].to(DEV), g["conds"].to(DEV), None, ratio=RATIO)
    assert yb.shape == (3, 10, 3) and tr.offset.shape == (3, 10, 3)
    close(yb, g["yb"])
    p = g["ps"].to(DEV).requires_grad_(True)
    d = tr(p, g["conds"].to(DEV), g["bi"].to(DEV), ratio=RATIO)
    close(compute_Jacobian(p, d, True, True), g["J"], 1e-4, 1e-5)


def test_translator_cond_and_second_order_grads():
    """def-regu style: a loss on the Jacobian differentiated w.r.t. weights and per-frame codes."""
    from selfreconcode_amd.model.Deformer import MLPTranslator
    from selfreconcode_amd.utils import compute_Jacobian
    tr = MLPTranslator(128, 6).to(DEV)
    sd = fx.det_params(fx.DEF_SPEC, 7)
    tr.load_state_dict(sd, strict=True)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ps = fx.det_tensor((37, 3), 1, 0.7); conds = fx.det_tensor((3, 128), 2, 0.1); bi = torch.arange(37) % 3
    po = ps.clone().requires_grad_(True); co = conds.clone().requires_grad_(True)
    do, _ = orc.translator_forward(sdo, po, co, bi, RATIO)
    Jo = orc.compute_jacobian(po, do, True, True)
    lo = (Jo ** 2).sum() + do.sum()
    ref = torch.autograd.grad(lo, [po, co, sdo["lin1.weight"], sdo["lin0.weight"], sdo["lin4.bias"]], allow_unused=True)
    pg = ps.to(DEV).requires_grad_(True); cg = conds.to(DEV).requires_grad_(True)
    d = tr(pg, cg, bi.to(DEV), ratio=RATIO)
    J = compute_Jacobian(pg, d, True, True)
    l = (J ** 2).sum() + d.sum()
    ours = torch.autograd.grad(l, [pg, cg, tr.lin1.weight, tr.lin0.weight, tr.lin4.bias], allow_unused=True)
    close(J, Jo, 1e-4, 1e-5)
    for a, b in zip(ours, ref):
        close(a, b, 1e-3, 1e-4 * max(1.0, float(b.abs().max())))


def test_render_golden_and_grads(golden):
    from selfreconcode_amd.model.RenderNet import RenderingNetwork_view_norm
    g = golden("render")
    rn = RenderingNetwork_view_norm(256, 'idr', 9, 3, [512] * 4, True, multires_n=0, multires_v=4).to(DEV)
    sd = fx.det_params(fx.REND_SPEC, 303)
    rn.load_state_dict(sd, strict=True)
    args = [g[k].to(DEV).requires_grad_(True) for k in ("pts", "nrm", "vd", "feat")]
    col = rn(*args, RATIO)
    close(col, g["col"])
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    argo = [g[k].clone().requires_grad_(True) for k in ("pts", "nrm", "vd", "feat")]
    co = orc.render_forward(sdo, *argo, RATIO)
    ref = torch.autograd.grad(co.abs().sum(), argo + [sdo["lin0.weight_v"], sdo["lin4.weight_g"]])
    ours = torch.autograd.grad(col.abs().sum(), args + [rn.lin0.weight_v, rn.lin4.weight_g])
    for a, b in zip(ours, ref):
        close(a, b, 2e-4, 2e-5 * max(1.0, float(b.abs().max())))


def test_engine_group4_tangents_and_reverse():
    """forward-mode Jacobian rows (group 4) == autograd Jacobian of the oracle; the group-4 reverse
    sweep == autograd gradient of a loss on (value, Jacobian)."""
    from selfreconcode_amd import mlp_engine as me
    from selfreconcode_amd import _lib
    net = _sdf(9)
    P = 50
    x = fx.det_tensor((P, 3), 4, 0.8)
    wt = torch.ones(12, device=DEV)
    A0 = torch.empty((P * 4, 40), device=DEV)
    xg = x.to(DEV)
    _lib.call("sr_pe_embed", xg.data_ptr(), P, 6, wt.data_ptr(), 0, 0, 0, 0, 4, A0.data_ptr(), 40, 0)
    Ws, bs = net.packed_weights()
    Ws = [w.detach().contiguous() for w in Ws]; bs = [b.detach() for b in bs]
    acts = me.forward(net.spec, A0, Ws, bs, 4)
    out = acts[-1].view(P, 4, -1)[:, :, :257]
    sd = {k: v.clone().requires_grad_(True) for k, v in fx.det_params(fx.SDF_SPEC, 9).items()}
    xo = x.clone().requires_grad_(True)
    yo, rc = orc.sdf_forward(sd, xo, None)
    full = torch.cat([yo, rc], 1)
    close(out[:, 0], full)
    cols = [0, 1, 100, 256]
    Jo = torch.stack([torch.autograd.grad(full[:, c].sum(), xo, retain_graph=True, create_graph=True)[0] for c in cols], 1)  # [P,4,3]
    close(out[:, 1:, cols].permute(0, 2, 1), Jo, 1e-4, 1e-5)
    # reverse sweep with cotangents on value and tangents of column 0
    cy = fx.det_tensor((P,), 5, 1.0); cj = fx.det_tensor((P, 3), 6, 1.0)
    lo = (yo[:, 0] * cy).sum() + (Jo[:, 0] * cj).sum()
    ref = torch.autograd.grad(lo, [sd["lin2.weight_v"], sd["lin5.bias"], sd["lin0.weight_v"]])
    ybar = torch.zeros((P, 4, 260), device=DEV)
    ybar[:, 0, 0] = cy.to(DEV); ybar[:, 1:, 0] = cj.to(DEV)
    WTs = [me.transpose_padded(Ws[l], net.spec.layers[l].K) for l in range(9)]
    A0bar, dWs, dbs = me.reverse(net.spec, A0, WTs, acts, ybar.view(P * 4, 260), 4)
    # chain dW_eff -> (g, v) through torch's own weight-norm autograd
    v2 = net.lin2.weight_v.detach().clone().requires_grad_(True); g2 = net.lin2.weight_g.detach()
    torch._weight_norm(v2, g2, 0).backward(dWs[2][:, :512])
    close(v2.grad, ref[0], 1e-3, 1e-4 * float(ref[0].abs().max()))
    close(dbs[5], ref[1], 1e-3, 1e-4 * float(ref[1].abs().max()))
    v0 = net.lin0.weight_v.detach().clone().requires_grad_(True); g0 = net.lin0.weight_g.detach()
    torch._weight_norm(v0, g0, 0).backward(dWs[0][:, :39])
    close(v0.grad, ref[2], 1e-3, 1e-4 * float(ref[2].abs().max()))


def test_sdf_only_variant_matches_full_network():
    from selfreconcode_amd import mlp_engine
    net = _sdf(21)
    x = fx.det_tensor((300, 3), 5, 0.8).to(DEV).requires_grad_(True)
    full = net(x, 0.8)
    g_full = torch.autograd.grad(full.abs().sum(), [x, net.lin8.weight_v, net.lin8.bias, net.lin2.weight_v])
    only = net(x, 0.8, sdf_only=True)
    assert net.rendcond is None and only.shape == (300, 1)
    g_only = torch.autograd.grad(only.abs().sum(), [x, net.lin8.weight_v, net.lin8.bias, net.lin2.weight_v])
    close(only, full, 1e-6, 1e-7)
    for a, b in zip(g_only, g_full):
        close(a, b, 1e-5, 1e-6 * max(1.0, float(b.abs().max())))
    # tiny batches take the 64x64-tile kernel: same numbers
    xs = x.detach()[:37].clone().requires_grad_(True)
    close(net(xs, 0.8), full[:37], 1e-6, 1e-7)


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (5, 3, 7), (63, 33, 39), (64, 32, 40), (65, 64, 33), (129, 257, 167), (300, 473, 512),
                                   (1000, 167, 289), (4097, 512, 41), (6129, 3, 512), (257, 31, 65), (8192, 130, 96),
                                   (5000, 512, 39), (3001, 257, 64), (70001, 300, 7), (1000, 256, 65)])      # + the 256 x 64 weight-gradient tile (K <= 64, N >= 256) and its edges
def test_gemm_kernels_ragged_shapes_vs_float64(M, N, K):
    """The layer-GEMM kernels on shapes that hit every clamp of the branch-free loaders (row tails, K tails inside and across
    float4s, narrow / ragged N, every tile configuration), with NaN planted in all padding the kernels are allowed to read:
    C = A B^T + b against float64, and dW = Z^T A, db = colsum(Z) likewise."""
    from selfreconcode_amd import mlp_engine as me
    g = torch.Generator().manual_seed(M * 7919 + N * 31 + K)
    ldk, ldn = me.pad4(K) + 4, me.pad4(N) + 8                      # row pitches wider than the logical widths
    A = torch.full((M, ldk), float("nan")); A[:, :K] = torch.randn(M, K, generator=g)
    B = torch.full((N, ldk), float("nan")); B[:, :K] = torch.randn(N, K, generator=g) * 0.1
    bias = torch.randn(N, generator=g)
    Ad, Bd, bd = A.to(DEV), B.to(DEV), bias.to(DEV)
    C = torch.full((M, ldn), float("nan"), device=DEV)
    me._gemm_nt(Ad, ldk, Bd, ldk, C, ldn, M, N, K, bd, 1, me.ACT_NONE, me.EPI_FWD)
    ref = A[:, :K].double() @ B[:, :K].double().t() + bias.double()
    torch.testing.assert_close(C[:, :N].cpu().double(), ref, rtol=2e-5, atol=2e-5 * max(1.0, K ** 0.5))
    # weight / bias gradient: Z [M, N] (pitch ldn), A [M, K] (pitch ldk)
    Z = torch.full((M, ldn), float("nan")); Z[:, :N] = torch.randn(M, N, generator=g)
    dW, db = me._gemm_tn(Z.to(DEV), ldn, Ad, ldk, M, N, K, me.pad4(K), 1)
    torch.testing.assert_close(dW[:, :K].cpu().double(), Z[:, :N].double().t() @ A[:, :K].double(), rtol=2e-5, atol=2e-5 * max(1.0, M ** 0.5))
    torch.testing.assert_close(db.cpu().double(), Z[:, :N].double().sum(0), rtol=2e-5, atol=2e-5 * max(1.0, M ** 0.5))
    assert torch.isfinite(dW).all()                                # padding columns [K, pad4(K)) are written as zeros


def test_pe_embed_kernel_vs_reference_golden(golden):
    """a1: the fused embed kernel itself against the reference's Embedder run (tests/golden/pe.npz): ratio None (all ones),
    annealed 0.35 / 1.0 and the <= 0 switch (all PE weights zero, used by initializeTmpSDF with ratio -1)."""
    from selfreconcode_amd.model.Embedder import embed_rows, get_embedder
    from selfreconcode_amd.utils.utils import resolve_band_weights, annealing_weights
    g = golden("pe")
    x = g["x"].to(DEV)
    for tag, ratio in (("none", None), ("r035", 0.35), ("r1", 1.0), ("neg", -1.0)):
        out = embed_rows(x, 6, resolve_band_weights(6, ratio))
        assert out.shape == (16, 40) and float(out[:, 39].abs().max()) == 0.0            # pad column is zero
        close(out[:, :39], g[tag], 1e-6, 1e-6)
    close(torch.tensor(annealing_weights(6, 0.35)), g["aw_035"], 1e-7, 1e-7)
    close(torch.tensor(annealing_weights(4, 0.7)), g["aw_07_4"], 1e-7, 1e-7)
    embed, dim = get_embedder(6)                                                         # API mirror (Embedder.py:44-54)
    assert dim == 39
    close(embed(x, annealing_weights(6, 0.35)), g["r035"], 1e-6, 1e-6)
    # first- and second-order input derivatives of the embed Function against autograd of the oracle's closed form
    xg = x.clone().requires_grad_(True)
    w = fx.det_tensor((16, 39), 77, 1.0).to(DEV)
    gx = torch.autograd.grad((embed_rows(xg, 6, annealing_weights(6, 0.8))[:, :39] * w).sum(), xg, create_graph=True)[0]
    g2 = torch.autograd.grad((gx * gx).sum(), xg)[0]
    xo = g["x"].clone().requires_grad_(True)
    go = torch.autograd.grad((orc.pe_embed(xo, 6, annealing_weights(6, 0.8)) * w.cpu()).sum(), xo, create_graph=True)[0]
    g2o = torch.autograd.grad((go * go).sum(), xo)[0]
    close(gx, go, 1e-5, 1e-5); close(g2, g2o, 1e-4, 1e-3)


def test_sdf_forward_backward_at_96k_rows_vs_fp64():
    """Value check (not a property check) on the paths only large batches reach: 128x128 NT tiles, several tiles per CU,
    weight-gradient GEMM with split-R > 1.  98 304 rows, full 257-wide output, against the CPU oracle in float64."""
    P = 98304
    net = _sdf(77)
    sd64 = {k: v.double().requires_grad_(True) for k, v in fx.det_params(fx.SDF_SPEC, 77).items()}
    x = fx.det_tensor((P, 3), 4242, 0.9)
    go = fx.det_tensor((P, 257), 4243, 1.0) / P
    xg = x.to(DEV).requires_grad_(True)
    y = net(xg, 1.0)
    l = (torch.cat([y, net.rendcond], 1) * go.to(DEV)).sum()
    names = ["lin0.weight_v", "lin1.weight_v", "lin3.weight_g", "lin4.weight_v", "lin4.bias", "lin8.weight_v", "lin8.bias"]
    ours = torch.autograd.grad(l, [xg] + [dict(net.named_parameters())[n] for n in names])
    xo = x.double().requires_grad_(True)
    torch.set_num_threads(max(1, torch.get_num_threads()))
    yo, rc = orc.sdf_forward(sd64, xo, 1.0)
    lo = (torch.cat([yo, rc], 1) * go.double()).sum()
    ref = torch.autograd.grad(lo, [xo] + [sd64[n] for n in names])
    close(y, yo.float(), 2e-5, 2e-6); close(net.rendcond, rc.float(), 2e-5, 2e-6)
    close(ours[0], ref[0].float(), 1e-4, 1e-4 * float(ref[0].abs().max()))
    for n, a, b in zip(names, ours[1:], ref[1:]):
        # sums over 98k rows: fp32 accumulation (fixed slab order) against float64
        torch.testing.assert_close(a.cpu(), b.float(), rtol=2e-4, atol=4e-5 * max(1e-3, float(b.abs().max())), msg=lambda m, n=n: n + ": " + m)      # (sums over 98k rows in float32: 4e-5 of the largest entry)


def test_fused_pack_refresh_and_fused_flush_match_torch_weight_norm():
    """One-launch refresh of all packed weights after an optimizer step (sr_pack_weights) and the one-launch weight-norm
    backward of the deferred gradients (sr_unpack_grads), against torch's own weight_norm (network.py:65-66) in float64."""
    import torch.nn as nn
    from selfreconcode_amd import mlp_engine as me
    torch.manual_seed(3)
    dev = "cuda:0"
    lins = [nn.utils.weight_norm(nn.Linear(39, 256)).to(dev), nn.utils.weight_norm(nn.Linear(256, 217)).to(dev), nn.Linear(217, 3).to(dev)]
    for l in lins:                      # first pack: per-layer torch path
        me.pack_linear(l)
    with torch.no_grad():               # "optimizer step"
        for l in lins:
            for p in l.parameters():
                p.add_(0.05 * torch.randn_like(p))
    me.refresh_packs(lins)
    for l in lins:
        W = me.pack_linear(l)
        if hasattr(l, "weight_g"):
            v, g = l.weight_v.double(), l.weight_g.double()
            ref = v * (g / v.norm(dim=1, keepdim=True))
        else:
            ref = l.weight.double()
        N, K = ref.shape
        assert W.shape == (N, me.pad4(K))
        assert torch.allclose(W[:, :K].double(), ref, rtol=0, atol=2e-7 * ref.abs().max().item())
        assert (W[:, K:] == 0).all()
        WT = me.transposed_of(W, K)
        assert torch.equal(WT[:, :N], W[:, :K].t()) and (WT[:, N:] == 0).all()

    # deferred gradients -> parameter gradients
    me.set_deferred_param_grads(True)
    try:
        refs = []
        for l in lins:
            W = me.pack_linear(l)
            sink = me._deferred_sink(W, l.bias)
            assert sink is not None
            dW, db, acc = sink
            assert not acc
            gW = torch.randn_like(dW); gb = torch.randn_like(db)
            dW.copy_(gW); db.copy_(gb)
            K = l.in_features
            if hasattr(l, "weight_g"):
                v = l.weight_v.detach().double().requires_grad_(True); g = l.weight_g.detach().double().requires_grad_(True)
                w = v * (g / v.norm(dim=1, keepdim=True))
                w.backward(gW[:, :K].double())
                refs.append((v.grad, g.grad, gb, gW))
            else:
                refs.append((gW[:, :K].double(), None, gb, gW))
        for l in lins:
            for p in l.parameters():
                p.grad = None
        me.flush_param_grads()
        for twice in range(2):
            for l, (rv, rg, rb, _) in zip(lins, refs):
                s = twice + 1
                v = l.weight_v if hasattr(l, "weight_g") else l.weight
                assert torch.allclose(v.grad.double(), s * rv, rtol=1e-5, atol=1e-5 * rv.abs().max().item())
                if rg is not None:
                    assert torch.allclose(l.weight_g.grad.double(), s * rg, rtol=1e-5, atol=1e-5 * rg.abs().max().item())
                assert torch.allclose(l.bias.grad, s * rb, rtol=1e-6, atol=1e-6)
            if twice == 0:              # a second flush of the same values ADDS into the existing gradients
                for l, (rv, rg, rb, gW) in zip(lins, refs):
                    W = me.pack_linear(l)
                    dW, db, acc = me._deferred_sink(W, l.bias)
                    assert not acc
                    dW.copy_(gW); db.copy_(rb)
                me.flush_param_grads()
    finally:
        me.set_deferred_param_grads(False)


def test_weight_gradient_stream_gives_bitwise_the_same_gradients():
    """Deferred mode with the weight-gradient GEMMs on their own stream (mlp_engine.TN_SIDE_STREAM) against the one-stream
    schedule: all weight-gradient launches share one stream, so the accumulation order into a layer's buffer is the program order
    and the flushed parameter gradients are bit-identical -- two uses of the network, one of them through the sdf-only head."""
    from selfreconcode_amd import mlp_engine as me
    from selfreconcode_amd.model.network import getTmpSdf
    torch.manual_seed(5)
    net = getTmpSdf(DEV, 6, 0.6, 256)
    xa = (torch.rand(20000, 3, device=DEV) - 0.5) * 1.4
    xb = (torch.rand(3000, 3, device=DEV) - 0.5) * 1.4
    grads = {}
    for flag in (True, False):
        me.TN_SIDE_STREAM = flag
        me.set_deferred_param_grads(True)
        try:
            for p in net.parameters():
                p.grad = None
            la = net(xa, 1.0, sdf_only=True).abs().mean()
            yb = net(xb, 1.0)
            lb = (yb ** 2).mean() + (net.rendcond ** 2).mean()
            (la + lb).backward()
            me.flush_param_grads()
            torch.cuda.synchronize()
            grads[flag] = [p.grad.clone() for p in net.parameters()]
        finally:
            me.set_deferred_param_grads(False)
            me.TN_SIDE_STREAM = True
    assert all(g is not None and torch.isfinite(g).all() for g in grads[True])
    for a, b in zip(grads[True], grads[False]):
        assert torch.equal(a, b)


def test_grouped_weight_gradients_equal_the_single_launches_bit_for_bit():
    """sr_mlp_gemm_tn_group (the weight gradients of one small reverse sweep in ONE launch + one slab-reduction launch) against one
    sr_mlp_gemm_tn per problem: ragged row counts and widths (K = 167, 473, 289 ..., N = 3 and 257), accumulate on and off, bias gradients
    with group 1 / 2 / 4, twelve problems at once."""
    from selfreconcode_amd import mlp_engine as me
    shapes = [(3000, 512, 512, 1, False), (3000, 512, 167, 1, True), (1777, 473, 512, 2, False), (4, 3, 512, 1, True), (6144, 257, 512, 4, False),
              (129, 512, 289, 1, True), (8188, 512, 512, 4, True), (33, 512, 512, 1, False), (2048, 512, 473, 2, True), (5000, 3, 512, 1, False),
              (1024, 512, 512, 1, True), (900, 257, 512, 1, True)]
    single, grouped, probs = [], [], []
    for i, (R, N, K, group, acc) in enumerate(shapes):
        R -= R % group
        Z = fx.det_tensor((R, me.pad4(N)), 400 + i, 1.0).to(DEV); A = fx.det_tensor((R, me.pad4(K)), 500 + i, 1.0).to(DEV)
        dW0 = fx.det_tensor((N, me.pad4(K)), 600 + i, 1.0).to(DEV); db0 = fx.det_tensor((N,), 700 + i, 1.0).to(DEV)
        a, b = dW0.clone(), db0.clone()
        me._gemm_tn(Z, Z.stride(0), A, A.stride(0), R, N, K, me.pad4(K), group, dW=a, db=b, accumulate=acc)
        single.append((a, b))
        c, d = dW0.clone(), db0.clone()
        grouped.append((c, d))
        probs.append((Z, Z.stride(0), A, A.stride(0), R, N, K, me.pad4(K), group, c, d, acc))
    me._gemm_tn_group(probs)
    torch.cuda.synchronize()
    for (a, b), (c, d), s in zip(single, grouped, shapes):
        assert torch.equal(a, c) and torch.equal(b, d), s
    me._gemm_tn_group(probs[:2])                      # a second, smaller group on the same buffers: accumulate semantics as the single launch
    me._gemm_tn(*probs[0][:9], dW=single[0][0], db=single[0][1], accumulate=probs[0][11])
    torch.cuda.synchronize()
    assert torch.equal(single[0][0], grouped[0][0]) and torch.equal(single[0][1], grouped[0][1])


def test_rows_frame_sum_is_exact_and_reproducible():
    """sr_rows_frame_sum (gradient of the per-frame code gather conds[batch_inds], model/Deformer.py:61,75) against float64 sums; ragged,
    empty and unsorted frames, a padded row pitch; two calls give the same bits (torch's index_add -- float atomics -- does not)."""
    from selfreconcode_amd import mlp_engine
    for P, E, n, sort in ((6144, 128, 3, True), (1, 5, 1, True), (0, 7, 2, True), (3001, 130, 8, False), (70000, 128, 3, True)):
        X = fx.det_tensor((P, E + 3), 5, 1.0).to(DEV)[:, :E]                       # row pitch E + 3
        idx = (fx.det_tensor((P,), 6, 0.5) + 0.5).mul(n).long().clamp(max=n - 1)
        if n > 1 and P > 10:
            idx[idx == 1] = 0                                                      # an empty frame
        if sort:
            idx = idx.sort().values
        idx = idx.to(DEV)
        a = mlp_engine.rows_frame_sum(X, idx, n)
        b = mlp_engine.rows_frame_sum(X, idx, n)
        assert torch.equal(a, b)
        want = torch.zeros(n, E, dtype=torch.float64, device=DEV).index_add(0, idx, X.double())
        torch.testing.assert_close(a.double(), want, rtol=1e-5, atol=4e-6 * max(1.0, P / n) ** 0.5)        # float32 partial sums of ~P/n terms of size <= 1
