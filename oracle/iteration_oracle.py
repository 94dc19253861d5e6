"""TEST INFRASTRUCTURE ONLY -- CPU restatement of one whole training iteration of the reference:
OptimNetwork.forward (model/network.py:451-644), computeTmpPcLoss (:647-697) and propagateTmpPsGrad (:702-814),
assembled from the row-level restatements of oracle/torch_oracle.py (each pinned to the reference's own modules through
tests/golden) and oracle/raster_oracle.py (pytorch3d 0.4.0 restated, parity unpinned).

PINNED as a whole: tests/test_oracle_golden.py holds `forward` + `propagate` to tests/golden/iteration.npz and `pc_loss` to
tests/golden/pcloss.npz -- the reference's own OptimNetwork.forward / backward / propagateTmpPsGrad / computeTmpPcLoss run verbatim
on CPU by oracle/gen_iteration_golden.py and oracle/gen_golden.py (reference modules; only the two pytorch3d renderers are the
restated ones, so the rasterisers remain the unpinned part).

Random draws are passed in (`rand`), exactly the tensors the product's `forward(..., rand=...)` takes, so both sides see
the same numbers (SURVEY.md 7 "Randomness").  Everything runs in the dtype of the parameters handed in (float32 or float64).
"""
import numpy as np
import torch

from . import torch_oracle as orc
from . import raster_oracle as ro


class Scene:
    """Plain container: CPU leaf tensors (requires_grad as the test wants them) + constants."""

    def __init__(self, sdf, tr, rnd, skin, poses, trans, dcond, rcond, cam, conf, point_radius, ang_thr):
        self.sdf, self.tr, self.rnd, self.skin = sdf, tr, rnd, skin           # state dicts; skin = dict(ws,b_min,b_max,Js,init_pose)
        self.poses, self.trans, self.dcond, self.rcond = poses, trans, dcond, rcond
        self.cam = cam                                                        # dict(focal[2], princ[2], R[3,3], T[3], H, W)
        self.conf, self.point_radius, self.ang_thr = conf, point_radius, ang_thr

    def deform(self, p, dcond, poses, trans, bi, ratio, lbs_only=False):
        """CompositeDeformer([MLPTranslator, LBSkinner]) (model/Deformer.py:10-20); `bi` None = [N,V,3] batch mode."""
        q = p if lbs_only else orc.translator_forward(self.tr, p, dcond, bi, ratio)[0]
        return orc.lbs_forward(q, poses, trans, batch_inds=bi, **self.skin)

    def rays(self, cols, rows):
        pix = torch.stack([cols, rows, torch.ones_like(cols)], -1).to(self.cam['focal'].dtype)
        return orc.view_rays(pix, self.cam['focal'], self.cam['princ'], self.cam['R'])


def pc_loss(sc, TmpVs, tmp_opt, defTmpVs, dcond, poses, trans, masks, gtMs, ratio, info):
    """computeTmpPcLoss (network.py:647-697): IoU mask loss (+ deformation consistency) -> INNER backward + template SGD
    step -> pc_weight * mean |f(TmpVs)| on the moved vertices."""
    conf = sc.conf
    N = gtMs.shape[0]
    mask_loss = orc.mask_iou_loss(masks, gtMs)
    info['mask_loss'] = mask_loss.detach()
    loss = mask_loss * (conf.get_float('pc_weight.mask_weight') if 'pc_weight.mask_weight' in conf else 1.)
    cw = conf.get_float('pc_weight.def_consistent.weight') if 'pc_weight.def_consistent' in conf else -1.
    if cw > 0.:
        lbs_only = sc.deform(TmpVs.view(1, -1, 3).expand(N, -1, 3), dcond, poses, trans, None, ratio, lbs_only=True)
        off2 = ((defTmpVs - lbs_only) ** 2).sum(-1)
        c = conf.get_float('pc_weight.def_consistent.c')
        closs = orc.gm_robust(off2, c, True).mean() if c > 0. else torch.sqrt(off2).mean()
        info['defconst_loss'] = closs.detach()
        loss = loss + closs * cw
    tmp_opt.zero_grad()
    loss.backward()
    tmp_opt.step()
    pred = orc.sdf_forward(sc.sdf, TmpVs, ratio)[0].view(-1)
    sdf_loss = pred.abs().mean()
    info['pc_loss_sdf'] = sdf_loss.detach()
    info['tmpl_pred'] = pred.detach()           # f at the moved template vertices: the tests look at its signs (gradient of the L1 term)
    return sdf_loss * (conf.get_float('pc_weight.weight') if 'pc_weight' in conf else 60.)


def color_normal(sc, TmpPs, rays, bi, rows, cols, dcond, poses, trans, rcond, gtCs, gtNs, ratio, N, info):
    """network.py:599-639."""
    conf = sc.conf
    total = 0.
    dfn = lambda p: sc.deform(p, dcond, poses, trans, bi, ratio)
    sdfs, feat = orc.sdf_forward(sc.sdf, TmpPs, ratio)
    nx = torch.autograd.grad(sdfs, TmpPs, torch.ones_like(sdfs), retain_graph=True, create_graph=True)[0]
    nx = nx / nx.norm(dim=1, keepdim=True)
    ds = dfn(TmpPs)                                                             # compute_cardinal_rays 'train' (utils.py:155-169)
    J = orc.compute_jacobian(TmpPs, ds, True, True)
    Ji, ok = orc.DiffMinv.apply(J)
    cr = (Ji @ rays.view(-1, 3, 1)).view(-1, 3)
    cr = torch.where(ok[:, None], cr, rays.detach())
    cr = cr / cr.norm(dim=1, keepdim=True)
    if conf.get_float('color_weight') > 0.:
        col = orc.render_forward(sc.rnd, TmpPs, nx, cr, feat, ratio)
        closs = orc.scatter_mean((gtCs[bi, rows, cols] - col).abs().sum(1), bi, N).mean()
        info['color_loss'] = closs.detach()
        total = total + conf.get_float('color_weight') * closs
    if gtNs is not None and conf.get_float('normal_weight') > 0.:
        if conf.get_bool('weighted_normal'):                                    # compute_deformed_normals 'test' (utils.py:132-153)
            y2, _ = orc.sdf_forward(sc.sdf, TmpPs, ratio)
            onx = torch.autograd.grad(y2, TmpPs, torch.ones_like(y2))[0]
            J2 = orc.compute_jacobian(TmpPs, dfn(TmpPs), False, False)
            Ji2, ok2 = orc.minv3x3(J2)
            cnx = (Ji2.transpose(-2, -1) @ onx.view(-1, 3, 1)).view(-1, 3)
            cnx = torch.where(ok2[:, None], cnx, (J2 @ onx.unsqueeze(-1)).view(-1, 3))
            cnx = cnx / cnx.norm(dim=1, keepdim=True)
            w = torch.clamp((-rays * cnx.detach()).sum(1).detach(), max=1., min=0.) ** 2           # :623-624: detached, also w.r.t. the rays
        else:
            w = torch.ones(nx.shape[0], dtype=nx.dtype)
        flip = torch.tensor([[-1., 0., 0.], [0., 1., 0.], [0., 0., -1.]], dtype=nx.dtype)
        gn = ((sc.cam['R'] @ flip) @ gtNs[bi, rows, cols].view(-1, 3, 1)).view(-1, 3)
        norms = gn.norm(dim=1, keepdim=True)
        valid = (norms > 0.0001)[..., 0]
        gn = torch.where(valid[:, None], gn / norms.clamp(min=1e-12), gn)
        J3 = orc.compute_jacobian(TmpPs, dfn(TmpPs), True, True)
        gn = (J3.transpose(-2, -1) @ gn.view(-1, 3, 1)).view(-1, 3)
        nl = (gn - nx).norm(2, dim=1) * w
        nloss = orc.scatter_mean(nl[valid], bi[valid], N).mean()
        info['normal_loss'] = nloss.detach()
        total = total + conf.get_float('normal_weight') * nloss
    return total


def forward(sc, TmpVs, Tmpfs, tmp_opt, datas, sample_pix, ratio, frame_ids, rand, dctnull=None, batchframe=None, inject=None):
    """OptimNetwork.forward (network.py:451-644) after the remesh block.  `inject` (optional): dict(initTmpPs, check) taken
    from the other side AFTER its refiner, for tests that want the downstream terms on identical ray sets (the |f| < 5e-5
    acceptance flips on single ulps).  Returns (total_loss, info, state) with state = what propagate() needs."""
    conf, cam = sc.conf, sc.cam
    info = {}
    gtCs, gtMs = datas['img'], datas['mask']
    N, H, W = gtCs.shape[0], cam['H'], cam['W']
    poses, trans, dcond, rcond = sc.poses[frame_ids], sc.trans[frame_ids], sc.dcond[frame_ids], sc.rcond[frame_ids]
    V = TmpVs.shape[0]
    defTmpVs = sc.deform(TmpVs[None].expand(N, -1, 3), dcond, poses, trans, None, ratio)
    with torch.no_grad():                                                       # :491-493 mesh rasteriser -> FindSurfacePs
        xy, z = ro.ndc_projection(defTmpVs, cam['focal'], cam['princ'], cam['R'], cam['T'], W, H)
        p2f, bary, _ = ro.rasterize_meshes(torch.cat([xy, z[..., None]], -1).float().numpy(), Tmpfs.numpy(), H, W)
        bi, rows, cols, p0, _ = orc.find_surface_ps(TmpVs.detach(), Tmpfs, torch.from_numpy(p2f), torch.from_numpy(bary).to(TmpVs.dtype))
    masks, _ = ro.render_point_silhouette(defTmpVs, cam['focal'], cam['princ'], cam['R'], cam['T'], H, W, sc.point_radius, 50)   # :495-497
    radius = int(np.round(sc.point_radius / 2. * float(min(H, W)) / 1.2))
    mgt = torch.nn.functional.max_pool2d(gtMs, kernel_size=2 * radius + 1, stride=1, padding=radius) if radius > 0 else gtMs
    total = pc_loss(sc, TmpVs, tmp_opt, defTmpVs, dcond, poses, trans, masks, mgt, ratio, info)
    poses, trans, dcond, rcond = sc.poses[frame_ids], sc.trans[frame_ids], sc.dcond[frame_ids], sc.rcond[frame_ids]   # :537 (fresh graph after the inner backward)
    sel = gtMs[bi, rows, cols] > 0.                                             # :507-526
    bi, rows, cols, p0 = bi[sel], rows[sel], cols[sel], p0[sel]
    if bi.shape[0] > sample_pix * N:
        sel = rand['ray_select'][:bi.shape[0]] < float(sample_pix * N) / float(bi.shape[0])
        bi, rows, cols, p0 = bi[sel], rows[sel], cols[sel], p0[sel]
    rays = sc.rays(cols, rows)
    campos = orc.cam_pos(cam['R'], cam['T'])
    if inject is None:
        p1, check = orc.optimize_surface_ps(campos.detach(), rays.detach(), p0.clone(), bi,
                                            lambda p: orc.sdf_forward(sc.sdf, p, ratio)[0],
                                            lambda p, b: sc.deform(p, dcond.detach(), poses.detach(), trans.detach(), b, ratio),
                                            5e-5, sc.ang_thr, 3.05, 1., 10)
    else:
        p1, check = inject['initTmpPs'].to(TmpVs.dtype), inject['check']
    info['rays'], info['p0'], info['p1'], info['check'] = bi.shape[0], p0, p1, check
    info['bi'], info['rows'], info['cols'] = bi, rows, cols
    # eikonal (:543-549)
    base = torch.cat([p1, TmpVs.detach()[rand['vert_select'][:V] < 4096. / float(V)]], 0)
    pts = torch.cat([base + rand['eik_local'][:base.shape[0]] * 0.01, rand['eik_global'][:base.shape[0] // 6] * (1.8 * 2) - 1.8], 0).requires_grad_(True)
    pred = orc.sdf_forward(sc.sdf, pts, ratio)[0]
    g = torch.autograd.grad(pred, pts, torch.ones_like(pred), create_graph=True)[0]
    grad_loss = ((g.norm(2, dim=-1) - 1) ** 2).mean()
    info['grad_loss'] = grad_loss.detach()
    total = total + grad_loss * conf.get_float('grad_weight')
    # deformation regulariser (:565-582)
    if 'def_regu' in conf and conf.get_float('def_regu.weight') > 0.:
        q = torch.cat([p1, TmpVs.detach()[rand['vert_select2'][:V] < 4096. / float(V)]], 0)
        q = torch.cat([q, q + rand['regu_local'][:q.shape[0]] * 0.01], 0).view(1, -1, 3).expand(N, -1, 3).contiguous().requires_grad_(True)
        dq = orc.translator_forward(sc.tr, q, dcond, None, ratio)[0]
        Jq = orc.compute_jacobian(q, dq, True, True)
        s = torch.log(torch.linalg.svdvals(Jq))
        def_loss = orc.gm_robust((s * s).sum(1), conf.get_float('def_regu.c'), True).mean()
        info['def_loss'] = def_loss.detach()
        total = total + def_loss * conf.get_float('def_regu.weight')
    # DCT (:585-593)
    if dctnull is not None and conf.get_float('dct_weight') > 0.:
        klen, Nlen = dctnull.shape
        idx = batchframe(frame_ids, Nlen)
        _, newJ = orc.lbs_transforms(sc.poses[idx].reshape(N * Nlen, 24, 3), sc.skin['Js'], sc.skin['init_pose'])
        dct = (dctnull[None] @ newJ.reshape(N, Nlen, 72)).abs().mean()
        info['dct_loss'] = dct.detach()
        total = total + dct * conf.get_float('dct_weight')
    state = None
    if int(check.sum()) > 0:
        TmpPs = p1[check].detach().clone().requires_grad_(True)
        state = dict(TmpPs=TmpPs, rays=rays[check], bi=bi[check], rows=rows[check], cols=cols[check])
        total = total + color_normal(sc, TmpPs, state['rays'], state['bi'], state['rows'], state['cols'], dcond, poses, trans, rcond,
                                     gtCs, datas.get('normal'), ratio, N, info)
    return total, info, state


def cross_matrix(v):
    z = torch.zeros_like(v[:, 0])
    return torch.stack([z, -v[:, 2], v[:, 1], v[:, 2], z, -v[:, 0], -v[:, 1], v[:, 0], z], dim=1).view(-1, 3, 3)


def propagate(sc, state, frame_ids, ratio):
    """propagateTmpPsGrad (network.py:702-814): returns (#systems, #invertible).  Camera tensors of `sc.cam` that require grad
    receive the ray / camera-centre terms of :798-813."""
    poses, trans, dcond = sc.poses[frame_ids], sc.trans[frame_ids], sc.dcond[frame_ids]
    p, bi = state['TmpPs'], state['bi']
    v_live = sc.rays(state['cols'], state['rows']) if state['rays'].requires_grad else state['rays']      # :715-718 (graph rebuilt)
    v = v_live.detach()
    c_live = orc.cam_pos(sc.cam['R'], sc.cam['T'])
    glp = p.grad
    pd = p.detach().clone().requires_grad_(True)
    f = orc.sdf_forward(sc.sdf, pd, ratio)[0]
    gfp = torch.autograd.grad(f, pd, torch.ones_like(f))[0]
    d = sc.deform(pd, dcond, poses, trans, bi, ratio)
    Jd = orc.compute_jacobian(pd, d, False, False)
    vx = cross_matrix(v)
    b = torch.cat([gfp.view(-1, 1, 3), vx @ Jd], 1)
    binv, ok = orc.minv3x3(b.permute(0, 2, 1) @ b)
    rhs = (glp.view(-1, 1, 3) @ (binv @ b.permute(0, 2, 1))).detach()
    f2 = orc.sdf_forward(sc.sdf, p.detach(), ratio)[0]
    d2 = sc.deform(p.detach(), dcond, poses, trans, bi, ratio)
    temp = (rhs[:, :, -3:] @ (-vx)).view(-1, 3)
    outs, cots = [f2, d2], [(-rhs[:, :, 0]).reshape(f2.shape), temp]
    if v_live.requires_grad:                                   # :798-809
        dc = d2.detach() - c_live.detach().view(1, 3)
        outs.append(v_live); cots.append((rhs[:, :, -3:] @ cross_matrix(dc)).view(-1, 3))
    if c_live.requires_grad:                                   # :811-813
        outs.append(c_live); cots.append(-temp.sum(0))
    torch.autograd.backward(outs, cots)
    return ok.numel(), int(ok.sum())
