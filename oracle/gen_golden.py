"""TEST INFRASTRUCTURE ONLY -- freezes outputs of the reference's *own* Python modules
(run on CPU through oracle/ref_harness.py) into small fixtures under tests/golden/.

Run here (the build container has /root/reference):   python oracle/gen_golden.py
The GPU box never runs this; it only reads the committed .npz files.

Network parameters are NOT stored (MBs): they are regenerated bit-identically on both
sides from `det_params` (a closed-form, RNG-free filler defined in oracle/fixtures.py).
"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_harness import load_reference  # noqa: E402
from oracle import fixtures as fx  # noqa: E402
from oracle import torch_oracle as orc  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
os.makedirs(OUT, exist_ok=True)
ref = load_reference()
torch.set_num_threads(8)


def save(name, **arrs):
    conv = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **conv)
    print("wrote", name, {k: v.shape for k, v in conv.items()})


class _SamplerSwap:
    """the CUDA grid sampler cannot run here; LBSkinner gets the oracle's
    gather (itself pinned against ATen's F.grid_sample in tests/test_oracle_golden.py)."""

    @staticmethod
    def apply(ws, grid):
        return orc.grid_sample_3d(ws, grid)


ref.Deformer.GridSamplerMine3dFunction = _SamplerSwap
ref.rutils.Fast3x3Minv = lambda m: list(orc.minv3x3(m))
ref.rutils.Fast3x3Minv_backward = lambda g, inv: orc.minv3x3_backward(g, inv)


def load_det(module, seed, spec):
    sd = fx.det_params(spec, seed)
    module.load_state_dict(sd, strict=True)
    return sd


# ---------------------------------------------------------------- a1 embedder / annealing
x = fx.det_tensor((16, 3), 11, 0.9)
pe = {}
for tag, ratio in [("none", None), ("r035", 0.35), ("r1", 1.0), ("neg", -1.0)]:
    embed, _ = ref.Embedder.get_embedder(6)
    if ratio is None:
        pe[tag] = embed(x)
    elif ratio <= 0:
        pe[tag] = embed(x, [0. for _ in range(12)])
    else:
        pe[tag] = embed(x, ref.rutils.annealing_weights(6, ratio))
save("pe", x=x, **pe, aw_035=np.array(ref.rutils.annealing_weights(6, 0.35)), aw_07_4=np.array(ref.rutils.annealing_weights(4, 0.7)))

# ---------------------------------------------------------------- a2/a3 SDF MLP (full size, det params)
sdf = ref.network.getTmpSdf("cpu", 6, 0.6, 256)
load_det(sdf, 101, fx.SDF_SPEC)
xs = fx.det_tensor((48, 3), 12, 0.8)
out = {}
for tag, ratio in [("r1", 1.0), ("r04", 0.4), ("dict", {'sdfRatio': 1.0, 'deformerRatio': 0.7, 'renderRatio': 1.0})]:
    xin = xs.clone().requires_grad_(True)
    y = sdf(xin, ratio)
    g = torch.autograd.grad(y, xin, torch.ones_like(y), create_graph=True)[0]
    out["sdf_" + tag] = y
    out["rend_" + tag] = sdf.rendcond[:, ::16]
    out["grad_" + tag] = g
    if tag == "r1":
        eik = ((g.norm(2, dim=-1) - 1) ** 2).mean()
        pg = torch.autograd.grad(eik, [sdf.lin0.weight_v, sdf.lin4.weight_g, sdf.lin7.bias, xin])
        out["eik"] = eik
        out["eik_dv0"] = pg[0][::37, ::5]
        out["eik_dg4"] = pg[1]
        out["eik_db7"] = pg[2]
        out["eik_dx"] = pg[3]
save("sdf", x=xs, **out)

# geometric init statistics of a fresh reference net (sphere of radius ~0.6, network.py:49-63)
torch.manual_seed(0)
fresh = ref.network.getTmpSdf("cpu", 6, 0.6, 256)
dirs = torch.nn.functional.normalize(fx.det_tensor((64, 3), 5, 1.0), dim=1)
save("sdf_init", dirs=dirs, f_at_r=torch.stack([fresh(dirs * r, 1.0)[:, 0] for r in (0.3, 0.6, 0.9)]))

# ---------------------------------------------------------------- a4 MLPTranslator
tr = ref.Deformer.MLPTranslator(128, 6)
load_det(tr, 202, fx.DEF_SPEC)
conds = fx.det_tensor((3, 128), 13, 0.1)
ps = fx.det_tensor((40, 3), 14, 0.7)
bi = torch.tensor([i % 3 for i in range(40)])
ratio = {'sdfRatio': 1.0, 'deformerRatio': 0.62, 'renderRatio': 1.0}
y1 = tr(ps, conds, bi, ratio=ratio)
off1 = tr.offset.clone()
psb = fx.det_tensor((3, 10, 3), 15, 0.7)
y2 = tr(psb, conds, None, ratio=ratio)
pj = ps.clone().requires_grad_(True)
dj = tr(pj, conds, bi, ratio=ratio)
J = ref.rutils.compute_Jacobian(pj, dj, True, True)
save("translator", ps=ps, conds=conds, bi=bi, y=y1, off=off1, psb=psb, yb=y2, J=J)

# ---------------------------------------------------------------- a10 render net
rn = ref.RenderNet.RenderingNetwork_view_norm(256, 'idr', 9, 3, [512, 512, 512, 512], True, multires_n=0, multires_v=4)
load_det(rn, 303, fx.REND_SPEC)
P = 24
pts = fx.det_tensor((P, 3), 16, 0.7)
nrm = torch.nn.functional.normalize(fx.det_tensor((P, 3), 17, 1.0), dim=1)
vd = torch.nn.functional.normalize(fx.det_tensor((P, 3), 18, 1.0), dim=1)
feat = fx.det_tensor((P, 256), 19, 0.5)
col = rn(pts, nrm, vd, feat, ratio)
save("render", pts=pts, nrm=nrm, vd=vd, feat=feat, col=col)

# ---------------------------------------------------------------- a5 LBS skinner (+ a6 through F.grid_sample)
vol = fx.synthetic_lbs_volume((7, 11, 9))          # (D,H,W) small
Js = fx.synthetic_joints()
bmin, bmax = fx.LBS_BMIN, fx.LBS_BMAX
init_pose = torch.from_numpy(ref.rutils.smpl_tmp_Apose(1))
skin = ref.Deformer.LBSkinner(vol, bmin, bmax, Js, np.array(orc.SMPL_PARENTS), init_pose=init_pose, align_corners=False)
poses = fx.det_tensor((3, 24, 3), 21, 0.15)
trans = fx.det_tensor((3, 3), 22, 0.05)
lp = fx.det_tensor((50, 3), 23, 0.6) * torch.tensor([0.7, 1.0, 0.3])
lbi = torch.tensor([i % 3 for i in range(50)])
yl = skin(lp, [poses, trans], lbi)
ylb = skin(lp[:48].view(3, 16, 3), [poses, trans], None)
newJ = skin.posedSkeleton([poses, trans])
# ATen's own 3-D sampler on the same volume: value and first derivative wrt the grid
gq = (fx.det_tensor((1, 1, 1, 60, 3), 24, 1.15)).requires_grad_(True)
vref = torch.nn.functional.grid_sample(vol, gq, mode='bilinear', padding_mode='border', align_corners=False)
gref = torch.autograd.grad(vref, gq, fx.det_tensor(tuple(vref.shape), 25, 1.0))[0]
save("lbs", poses=poses, trans=trans, p=lp, bi=lbi, y=yl, yb=ylb, newJ=newJ, init_pose=skin.init_pose,
     gq=gq, aten_val=vref, aten_ggrid=gref)

# ---------------------------------------------------------------- a9 cardinal rays / deformed normals
comp = ref.Deformer.CompositeDeformer([tr, skin])
defconds = [conds, [poses, trans]]
pc = (fx.det_tensor((30, 3), 26, 0.35) * torch.tensor([0.7, 1.0, 0.3])).requires_grad_(True)
bic = torch.tensor([i % 3 for i in range(30)])
rays = torch.nn.functional.normalize(fx.det_tensor((30, 3), 27, 1.0), dim=1)
crays, dsv = ref.rutils.compute_cardinal_rays(comp, pc, rays, defconds, bic, ratio, 'train')
nx, _ = ref.rutils.compute_deformed_normals(sdf, comp, pc, defconds, bic, ratio, 'train')
save("cardinal", p=pc, bi=bic, rays=rays, crays=crays, ds=dsv, nx=nx)

# ---------------------------------------------------------------- a12 OptimizeSurfacePs (fresh sphere SDF so that it converges)
tr2 = ref.Deformer.MLPTranslator(128, 6)
tr2.load_state_dict(fx.det_params(fx.DEF_SPEC, 202, last_scale=0.05), strict=True)
comp2 = ref.Deformer.CompositeDeformer([tr2, skin])
sph = ref.network.getTmpSdf("cpu", 6, 0.6, 256)
sph.load_state_dict(fx.sphere_sdf_params(7), strict=True)
campos = torch.tensor([0.05, -0.1, 2.4])
Pn = 96
dirs = torch.nn.functional.normalize(fx.det_tensor((Pn, 3), 31, 1.0) + torch.tensor([0., 0., 1.2]), dim=1)
with torch.no_grad():
    r0 = 0.6
    for _ in range(30):   # radial bisection-free fixed point to sit near the surface
        r0 = r0 - sph(dirs * r0, 1.0)[:, 0:1]
surf = (dirs * r0).detach()
p0 = surf + fx.det_tensor((Pn, 3), 32, 0.004)
bit = torch.tensor([i % 3 for i in range(Pn)])
with torch.no_grad():
    d0 = comp2(surf, defconds, bit, ratio=ratio)
rays_t = torch.nn.functional.normalize(d0 - campos.view(1, 3), dim=1)
p_in = p0.clone()
ps_out, ok = ref.FindSurfacePs.OptimizeSurfacePs(campos, rays_t, p_in, bit, sph, ratio, comp2, defconds,
                                                 dthreshold=5.e-5, athreshold=0.04, w1=3.05, w2=1., times=10)
save("tracer", campos=campos, rays=rays_t, p0=p0, bi=bit, ps=ps_out, ok=ok,
     surf=surf, sph_probe=sph(surf[:8] * 1.1, 1.0))

# ---------------------------------------------------------------- a13 camera closed forms
cam = object.__new__(ref.network.RectifiedPerspectiveCameras)
torch.nn.Module.__init__(cam)
cam.focal_length = torch.tensor([[648.0, 650.0]])
cam.principal_point = torch.tensor([[271.0, 268.5]])
cam.R = orc.quat2mat(torch.tensor([[0.02, 0.01, 0.999, 0.03]]))
cam.T = torch.tensor([[0.03, -0.2, 2.5]])
cam.image_size = torch.tensor([[540, 540]])
pix = torch.cat([fx.det_tensor((20, 2), 41, 250.0) + 270.0, torch.ones(20, 1)], 1)
save("camera", focal=cam.focal_length[0], princ=cam.principal_point[0], R=cam.R[0], T=cam.T[0], pix=pix,
     rays=cam.view_rays(pix), campos=cam.cam_pos(), ang=np.array(cam.angThreshold(0.5)))

# ---------------------------------------------------------------- a11 FindSurfacePs with a scatter(min) stand-in
def _scatter(src, index, reduce=None, out=None, dim_size=None, dim=0):
    if reduce == 'min':
        return out.scatter_reduce(0, index, src, reduce='amin', include_self=True)
    raise NotImplementedError
ref.FindSurfacePs.scatter = _scatter
N_, H_, W_, K_ = 2, 6, 7, 3
p2f = (fx.det_tensor((N_, H_, W_, K_), 51, 12.0)).long()
p2f[p2f < -3] = -1
p2f = p2f.clamp(min=-1) + torch.arange(N_).view(N_, 1, 1, 1) * 12 * (p2f >= 0)
bary = fx.det_tensor((N_, H_, W_, K_, 3), 52, 0.6) + 0.35
V_ = fx.det_tensor((20, 3), 53, 1.0)
Fc = (fx.det_tensor((12, 3), 54, 10.0).abs().long()) % 20


class _Frag:
    pix_to_face = p2f
    bary_coords = bary


b_, r_, c_, p0_, f_ = ref.FindSurfacePs.FindSurfacePs(V_, Fc, _Frag)
save("findsurf", p2f=p2f, bary=bary, V=V_, F=Fc, b=b_, r=r_, c=c_, p0=p0_, finds=f_)

# ---------------------------------------------------------------- a14 misc closed forms
xg = fx.det_tensor((50,), 61, 2.0).abs()
save("misc", xg=xg, gm_sq=ref.rutils.GMRobustError(xg, 0.5, True), gm=ref.rutils.GMRobustError(xg, 0.01, False),
     dctnull=ref.rutils.DCTNullSpace(10, 30), quat=fx.det_tensor((4, 4), 62, 1.0),
     qmat=ref.rutils.quat2mat(fx.det_tensor((4, 4), 62, 1.0)),
     rod_in=fx.det_tensor((9, 3), 63, 0.8), rod=ref.smpl_util.batch_rodrigues(fx.det_tensor((9, 3), 63, 0.8)))

# ---------------------------------------------------------------- a16 Seg3dLossless verbatim on an analytic SDF
def ell(points):
    c = torch.tensor([0.05, -0.1, 0.02]).view(1, 1, 3)
    a = torch.tensor([0.45, 0.8, 0.25]).view(1, 1, 3)
    return (((points - c) / a).norm(dim=-1) - 1.0).view(1, 1, -1) * 0.25


resolutions = [(5 + 1 - 1, 7 - 1 + 1, 3), (9, 13, 5), (17, 25, 9), (33, 49, 17)]
resolutions = [(5, 7, 3), (9, 13, 5), (17, 25, 9), (33, 49, 17)]
eng = ref.MCAcc.Seg3dLossless(query_func=ell, b_min=[-0.8, -1.25, -0.4], b_max=[0.8, 0.95, 0.4], resolutions=resolutions,
                              align_corners=False, balance_value=0.0, device='cpu', visualize=False, debug=False,
                              use_cuda_impl=False, faster=False)
nq = [0]


def ell_counted(points):
    nq[0] += points.shape[1]
    return ell(points)


eng.query_func = ell_counted
vol_out = eng.forward()
save("seg3d", vol=vol_out[0, 0], nq=np.array(nq[0]), spacing=np.array([eng.spacing_x, eng.spacing_y, eng.spacing_z]),
     origin=np.array([eng.bx, eng.by, eng.bz]))
print("done")

# ---------------------------------------------------------------- state_dict key/shape contract (checkpoint compatibility, utils/utils.py:257-316)
import json
keys = {}
for name, mod in (("sdf", ref.network.getTmpSdf("cpu", 6, 0.6, 256)), ("translator", ref.Deformer.MLPTranslator(128, 6)),
                  ("render", ref.RenderNet.RenderingNetwork_view_norm(256, 'idr', 9, 3, [512, 512, 512, 512], True, multires_n=0, multires_v=4)),
                  ("skinner", skin)):
    keys[name] = {k: list(v.shape) for k, v in mod.state_dict().items()}
with open(os.path.join(OUT, "state_keys.json"), "w") as fh:
    json.dump(keys, fh, indent=0, sort_keys=True)
print("wrote state_keys.json", {k: len(v) for k, v in keys.items()})

# ---------------------------------------------------------------- a15 propagateTmpPsGrad run verbatim (network.py:702-814)
# The method is called on a bare OptimNetwork object (its constructor needs the pytorch3d renderers): det-parameter SDF and
# translator, the small LBS volume, a 5-frame dataset stand-in with the reference's accessors (dataset/dataset.py:117-127) and
# learnable focal length / principal point / T as in config.conf:10-15, so the ray / camera-centre terms (:798-813) are exercised.
import types

ref.network.Fast3x3Minv = lambda m: list(orc.minv3x3(m))
Fn, Hh, Ww = 5, 64, 48


class _Seq:
    def __init__(self):
        leaf = lambda t: t.clone().requires_grad_(True)
        self.poses = leaf(fx.det_tensor((Fn, 24, 3), 71, 0.15)); self.trans = leaf(fx.det_tensor((Fn, 3), 72, 0.05))
        self.conds = [leaf(fx.det_tensor((Fn, 128), 73, 0.1)), leaf(fx.det_tensor((Fn, 256), 74, 0.1))]
        self.focal = leaf(torch.tensor([58.0, 60.0])); self.princ = leaf(torch.tensor([23.0, 33.5])); self.T = leaf(torch.tensor([0.03, -0.1, 2.5]))
        self.R = orc.quat2mat(torch.tensor([[0.02, 0.01, 0.999, 0.03]]))

    def get_grad_parameters(self, idxs, device):
        return self.poses[idxs], self.trans[idxs], self.conds[0][idxs], self.conds[1][idxs]

    def get_camera_parameters(self, N, device):
        return self.focal.view(1, 2).expand(N, 2), self.princ.view(1, 2).expand(N, 2), self.R.expand(N, 3, 3), self.T.view(1, 3).expand(N, 3), Hh, Ww


seq = _Seq()
for m in (sdf, tr):
    for prm in m.parameters():
        prm.grad = None
onet = object.__new__(ref.network.OptimNetwork)
torch.nn.Module.__init__(onet)
onet.sdf, onet.deformer, onet.dataset, onet.info = sdf, comp, seq, {}
onet.maskRender = types.SimpleNamespace(rasterizer=types.SimpleNamespace(cameras=None))
Pq = 36
fids = torch.tensor([4, 0, 2])
onet.batch_inds = torch.tensor([i % 3 for i in range(Pq)])
onet.col_inds = (fx.det_tensor((Pq,), 75, 0.5) * Ww + Ww / 2).long().clamp(0, Ww - 1)
onet.row_inds = (fx.det_tensor((Pq,), 76, 0.5) * Hh + Hh / 2).long().clamp(0, Hh - 1)
onet.TmpPs = (fx.det_tensor((Pq, 3), 77, 0.35) * torch.tensor([0.7, 1.0, 0.3])).requires_grad_(True)
onet.TmpPs.grad = fx.det_tensor((Pq, 3), 78, 1.0)
cam0 = ref.network.RectifiedPerspectiveCameras(*seq.get_camera_parameters(3, 'cpu')[:4], image_size=[(Ww, Hh)])
onet.rays = cam0.view_rays(torch.cat([onet.col_inds.view(-1, 1), onet.row_inds.view(-1, 1), torch.ones(Pq, 1, dtype=torch.long)], dim=-1).float())
assert onet.rays.requires_grad
onet.propagateTmpPsGrad(fids, ratio)
sp, tp = dict(sdf.named_parameters()), dict(tr.named_parameters())
save("propagate", fids=fids, bi=onet.batch_inds, cols=onet.col_inds, rows=onet.row_inds, p=onet.TmpPs.detach(), glp=fx.det_tensor((Pq, 3), 78, 1.0),
     poses=seq.poses.detach(), trans=seq.trans.detach(), dcond=seq.conds[0].detach(), rcond=seq.conds[1].detach(),
     focal=seq.focal.detach(), princ=seq.princ.detach(), T=seq.T.detach(), R=seq.R[0], HW=np.array([Hh, Ww]),
     inv_info=np.array(onet.info['invInfo']),
     g_poses=seq.poses.grad, g_trans=seq.trans.grad, g_dcond=seq.conds[0].grad, g_focal=seq.focal.grad, g_princ=seq.princ.grad, g_T=seq.T.grad,
     g_sdf_v0=sp['lin0.weight_v'].grad[::37, ::5], g_sdf_g4=sp['lin4.weight_g'].grad, g_sdf_b7=sp['lin7.bias'].grad, g_sdf_v8=sp['lin8.weight_v'].grad[:1, ::7],
     g_tr_w0=tp['lin0.weight'].grad[::41, ::9], g_tr_w2=tp['lin2.weight'].grad[::53, ::47], g_tr_b4=tp['lin4.bias'].grad, g_tr_w4=tp['lin4.weight'].grad[:, ::11])
print("rcond grad:", None if seq.conds[1].grad is None else float(seq.conds[1].grad.abs().max()))

# ---------------------------------------------------------------- a14 computeTmpPcLoss run verbatim (network.py:647-697)
# Bare OptimNetwork again; `imgs[..., -1]` (the point-silhouette masks, pytorch3d in the reference) come from the oracle's restated
# renderer so that the inner backward has a path into the deformer; everything the method itself does -- IoU loss, deformation
# consistency, inner backward, template SGD step, |f(TmpVs)| -- is the reference's code on the reference's modules.
from oracle import raster_oracle as ro  # noqa: E402


class _DictConf:
    """get_float / `in` on dotted keys, as pyhocon's ConfigTree serves them (config.conf:79-88 values)."""

    def __init__(self, d):
        self.d = d

    def _find(self, key):
        cur = self.d
        for part in key.split('.'):
            if not isinstance(cur, dict) or part not in cur:
                return None
            cur = cur[part]
        return cur

    def __contains__(self, key):
        return self._find(key) is not None

    def get_float(self, key):
        return float(self._find(key))


for m in (sdf, tr):
    for prm in m.parameters():
        prm.grad = None
seq2 = _Seq()
pcnet = object.__new__(ref.network.OptimNetwork)
torch.nn.Module.__init__(pcnet)
pcnet.sdf, pcnet.deformer, pcnet.info = sdf, comp, {'pc_loss': {}}
pcnet.conf = _DictConf({'pc_weight': {'weight': 60., 'laplacian_weight': -10., 'edge_weight': -10., 'norm_weight': -0.001,
                                      'def_consistent': {'weight': 0.6, 'c': 0.01}}})
pcnet.sdfShrinkRadius = 0.0
Vn, Np, Hp, Wp, rad = 400, 2, 40, 40, 0.08
dirs_v = torch.nn.functional.normalize(fx.det_tensor((Vn, 3), 81, 1.0), dim=1)
pcnet.TmpVs = (dirs_v * torch.tensor([0.28, 0.42, 0.16]) + fx.det_tensor((Vn, 3), 82, 0.01)).requires_grad_(True)
pcnet.Tmpfs = torch.zeros(1, 3, dtype=torch.long)
pcnet.TmpOptimizer = torch.optim.SGD([pcnet.TmpVs], lr=0.05, momentum=0.9)
V0 = pcnet.TmpVs.detach().clone()
fid2 = torch.tensor([3, 1])
poses2, trans2, dcond2, _ = seq2.get_grad_parameters(fid2, 'cpu')
defc = [dcond2, [poses2, trans2]]
defV = comp(pcnet.TmpVs[None].expand(Np, -1, 3), defc, ratio=ratio)
focal_p, princ_p, T_p = torch.tensor([52.0, 50.0]), torch.tensor([17.5, 20.5]), torch.tensor([0.02, -0.05, 2.4])
masks_p, _ = ro.render_point_silhouette(defV, focal_p, princ_p, seq2.R[0], T_p, Hp, Wp, rad, 50)
gt_p = (fx.det_tensor((Np, Hp, Wp), 83, 1.0) > 0.2).float()
out_p = pcnet.computeTmpPcLoss(types.SimpleNamespace(verts_padded=lambda: defV), defc, masks_p[..., None], gt_p, ratio)
inner = dict(g_tr_w0=tp['lin0.weight'].grad[::41, ::9].clone(), g_tr_w2=tp['lin2.weight'].grad[::53, ::47].clone(), g_tr_b4=tp['lin4.bias'].grad.clone(),
             g_poses=seq2.poses.grad.clone(), g_trans=seq2.trans.grad.clone(), g_dcond=seq2.conds[0].grad.clone())
out_p.backward()
save("pcloss", V0=V0, fids=fid2, poses=seq2.poses.detach(), trans=seq2.trans.detach(), dcond=seq2.conds[0].detach(), focal=focal_p, princ=princ_p,
     T=T_p, R=seq2.R[0], HW=np.array([Hp, Wp]), radius=np.array(rad), gt=gt_p, masks=masks_p.detach(),
     mask_loss=np.array(pcnet.info['pc_loss']['mask_loss']), defconst_loss=np.array(pcnet.info['pc_loss']['defconst_loss']),
     pc_loss_sdf=np.array(pcnet.info['pc_loss_sdf']), out=out_p.detach(), V1=pcnet.TmpVs.detach(), **inner,
     g_sdf_v0=sp['lin0.weight_v'].grad[::37, ::5], g_sdf_g4=sp['lin4.weight_g'].grad, g_sdf_b7=sp['lin7.bias'].grad, g_sdf_v8=sp['lin8.weight_v'].grad[:1, ::7])
