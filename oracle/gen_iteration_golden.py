"""TEST INFRASTRUCTURE ONLY -- one WHOLE training iteration of the reference, run verbatim on CPU, frozen into
tests/golden/iteration.npz:   python oracle/gen_iteration_golden.py     (build container only: needs /root/reference)

`OptimNetwork.forward` (model/network.py:451-644) + `loss.backward()` + `propagateTmpPsGrad` (:702-814) are called on a bare
reference `OptimNetwork` object built from the reference's own modules (ImplicitNetwork, MLPTranslator, LBSkinner,
RenderingNetwork_view_norm, utils.*).  What cannot run here is replaced by the stand-ins SURVEY.md 8(c) names, nothing else:
  * the two pytorch3d renderers (`maskRender`, `pcRender`) -> oracle/raster_oracle.py (pytorch3d 0.4.0 restated; parity unpinned),
    `Meshes` / `Pointclouds` -> containers with the three accessors the method uses;
  * CUDA extensions: the grid sampler -> the oracle's gather (pinned to ATen), FastMinv -> oracle.minv3x3 (pinned to M^-1 M = I);
  * torch_scatter.scatter -> index_add / scatter_reduce.
Every random draw of the iteration (torch.rand / torch.randn_like inside forward and utils.sample_points) is recorded in call order
and stored, so that the oracle and the product can be fed the same numbers through their `rand=` argument.
"""
import os
import sys
import types
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle.ref_harness import load_reference  # noqa: E402
from oracle import fixtures as fx  # noqa: E402
from oracle import torch_oracle as orc  # noqa: E402
from oracle import raster_oracle as ro  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
ref = load_reference()
torch.set_num_threads(8)
RATIO = {'sdfRatio': 1.0, 'deformerRatio': 0.62, 'renderRatio': 1.0}


class _SamplerSwap:
    @staticmethod
    def apply(ws, grid):
        return orc.grid_sample_3d(ws, grid)


ref.Deformer.GridSamplerMine3dFunction = _SamplerSwap
ref.rutils.Fast3x3Minv = lambda m: list(orc.minv3x3(m))
ref.rutils.Fast3x3Minv_backward = lambda g, inv: orc.minv3x3_backward(g, inv)
ref.network.Fast3x3Minv = lambda m: list(orc.minv3x3(m))


def _scatter(src, index, reduce=None, out=None, dim_size=None, dim=0):
    if reduce == 'min':
        return out.scatter_reduce(0, index, src, reduce='amin', include_self=True)
    if reduce == 'mean':
        return orc.scatter_mean(src, index, dim_size)
    raise NotImplementedError(reduce)


ref.FindSurfacePs.scatter = _scatter
ref.network.scatter = _scatter


class Meshes:                                   # pytorch3d.structures.Meshes: only what forward / computeTmpPcLoss touch
    def __init__(self, verts, faces, **kw):
        self._verts, self._faces = list(verts), list(faces)

    def verts_list(self):
        return self._verts

    def verts_padded(self):
        return torch.stack(self._verts, 0)


class Pointclouds:
    def __init__(self, points, features=None):
        self._points = list(points)

    def points_padded(self):
        return torch.stack(self._points, 0)


ref.network.Meshes, ref.network.Pointclouds = Meshes, Pointclouds


class _Frags:
    def __init__(self, p2f, bary):
        self.pix_to_face, self.bary_coords = p2f, bary


class MaskRender:
    """MeshRendererWithFragments(MeshRasterizer(blur 0, 1 face / pixel, perspective-correct)) -> (images, fragments)."""

    def __init__(self, H, W, faces):
        self.rasterizer = types.SimpleNamespace(cameras=None)
        self.H, self.W, self.faces = H, W, faces

    def __call__(self, meshes):
        cam = self.rasterizer.cameras
        V = meshes.verts_padded().detach()
        xy, z = ro.ndc_projection(V, cam.focal_length[0], cam.principal_point[0], cam.R[0], cam.T[0], self.W, self.H)
        p2f, bary, _ = ro.rasterize_meshes(torch.cat([xy, z[..., None]], -1).float().numpy(), self.faces.numpy(), self.H, self.W)
        return None, _Frags(torch.from_numpy(p2f), torch.from_numpy(bary).to(V.dtype))


class PcRender:
    """PointsRendererWithFrags(PointsRasterizer(radius, points_per_pixel=50), AlphaCompositor) -> (images [N,H,W,C], fragments)."""

    def __init__(self, H, W, radius):
        self.rasterizer = types.SimpleNamespace(cameras=None, raster_settings=types.SimpleNamespace(radius=radius))
        self.H, self.W = H, W

    def __call__(self, clouds):
        cam = self.rasterizer.cameras
        masks, _ = ro.render_point_silhouette(clouds.points_padded(), cam.focal_length[0], cam.principal_point[0], cam.R[0], cam.T[0], self.H, self.W,
                                              self.rasterizer.raster_settings.radius, 50)
        return masks[..., None], None


class DictConf:
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

    def get_int(self, key):
        return int(self._find(key))

    def get_bool(self, key):
        return bool(self._find(key))


LOSS_COARSE = {'color_weight': 0.5, 'normal_weight': 0.1, 'weighted_normal': True, 'grad_weight': 1., 'offset_weight': 0.,
               'def_regu': {'weight': 0.1, 'c': 0.5}, 'dct_weight': 2.,
               'pc_weight': {'weight': 60., 'laplacian_weight': -10., 'edge_weight': -10., 'norm_weight': -0.001, 'def_consistent': {'weight': 0.6, 'c': 0.01}}}


icosphere = fx.icosphere          # (shared with the GPU tests, which cannot import this module: it loads the reference)


def main():
    F, H, W, N, SP = 36, 48, 48, 2, 150
    # ---- the reference's modules
    sdf = ref.network.getTmpSdf("cpu", 6, 0.6, 256)
    sdf.load_state_dict(fx.sphere_sdf_params(7), strict=True)
    tr = ref.Deformer.MLPTranslator(128, 6)
    tr.load_state_dict(fx.det_params(fx.DEF_SPEC, 202, last_scale=0.05), strict=True)
    skin = ref.Deformer.LBSkinner(fx.synthetic_lbs_volume((7, 11, 9)), fx.LBS_BMIN, fx.LBS_BMAX, fx.synthetic_joints(), np.array(orc.SMPL_PARENTS),
                                  init_pose=torch.from_numpy(ref.rutils.smpl_tmp_Apose(1)), align_corners=False)
    comp = ref.Deformer.CompositeDeformer([tr, skin])
    rn = ref.RenderNet.RenderingNetwork_view_norm(256, 'idr', 9, 3, [512, 512, 512, 512], True, multires_n=0, multires_v=4)
    rn.load_state_dict(fx.det_params(fx.REND_SPEC, 303), strict=True)

    class Seq:
        frame_num = F
        video_segmented_index = []

        def __init__(self):
            leaf = lambda t: t.clone().requires_grad_(True)
            self.poses = leaf(fx.det_tensor((F, 24, 3), 91, 0.12)); self.trans = leaf(fx.det_tensor((F, 3), 92, 0.04))
            self.conds = [leaf(fx.det_tensor((F, 128), 93, 0.1)), leaf(fx.det_tensor((F, 256), 94, 0.1))]
            self.focal = leaf(torch.tensor([1.2 * W, 1.2 * W])); self.princ = leaf(torch.tensor([W / 2.0, H / 2.0])); self.T = leaf(torch.tensor([0., 0.1, 2.4]))
            self.R = orc.quat2mat(torch.tensor([[0., 0., 1., 0.]]))

        def get_grad_parameters(self, idxs, device):
            return self.poses[idxs], self.trans[idxs], self.conds[0][idxs], self.conds[1][idxs]

        def get_camera_parameters(self, n, device):
            return self.focal.view(1, 2).expand(n, 2), self.princ.view(1, 2).expand(n, 2), self.R.expand(n, 3, 3), self.T.view(1, 3).expand(n, 3), H, W

        def get_batchframe_data(self, name, fids, batchsize):                     # dataset/dataset.py:128-147, unsegmented video
            data = getattr(self, name)
            starts = (fids - batchsize // 2).clamp(min=0, max=self.frame_num - batchsize)
            return data[starts.view(-1, 1) + torch.arange(0, batchsize).view(1, batchsize)], fids - starts
    ds = Seq()
    # ---- template: an icosphere pulled onto the zero set of the SDF along the radius
    dirs, faces = icosphere(3)
    with torch.no_grad():
        r = torch.full((dirs.shape[0], 1), 0.6)
        for _ in range(30):
            r = r - sdf(dirs * r, 1.0)[:, 0:1]
    # ... and pushed off it again by up to 4e-3: the template term is an L1 of f at the vertices (network.py:690-694), its gradient is
    # sign(f) per vertex -- vertices sitting ON the zero set would make every parity comparison a coin toss on |f| ~ 1e-6
    TmpVs = (dirs * r + fx.det_tensor((dirs.shape[0], 3), 97, 0.004)).detach().clone().requires_grad_(True)
    net = object.__new__(ref.network.OptimNetwork)
    torch.nn.Module.__init__(net)
    net.conf = DictConf(LOSS_COARSE)
    net.sdf, net.deformer, net.netRender, net.dataset = sdf, comp, rn, ds
    net.maskRender, net.pcRender = MaskRender(H, W, faces), PcRender(H, W, 0.06)
    net.engine = None
    net.TmpVs, net.Tmpfs = TmpVs, faces
    net.TmpOptimizer = torch.optim.SGD([net.TmpVs], lr=0.05, momentum=0.9)
    net.forward_time, net.remesh_intersect, net.remesh_time = 1, 30, 0.
    net.next_conf = net.next_train_conf = None
    net.draw, net.enable_mesh_color, net.sdfShrinkRadius = False, True, 0.0
    net.dctnull = ref.rutils.DCTNullSpace(10, 30)
    cam0 = ref.network.RectifiedPerspectiveCameras(*ds.get_camera_parameters(N, 'cpu')[:4], image_size=[(W, H)])
    net.angThred = cam0.angThreshold(0.5)
    fids = torch.tensor([21, 7])
    datas = {'img': fx.det_tensor((N, H, W, 3), 95, 1.0), 'mask': torch.ones(N, H, W), 'normal': fx.det_tensor((N, H, W, 3), 96, 1.0)}
    datas['normal'][:, ::5] = 0.                                                   # rows without a ground-truth normal (invalid in the normal loss)
    V0 = TmpVs.detach().clone()
    # ---- record every random draw of the iteration, in call order
    draws = []
    real_rand, real_randn_like = torch.rand, torch.randn_like

    def rec_rand(*a, **k):
        t = real_rand(*a, **k); draws.append(('rand', t.clone())); return t

    def rec_randn_like(x, **k):
        t = real_randn_like(x, **k); draws.append(('randn_like', t.clone())); return t
    refined = {}
    real_refiner = ref.utils.OptimizeSurfacePs

    def rec_refiner(cam_pos, rays, p0, bi, *a, **k):                      # the refiner's own inputs / outputs (all selected rays)
        refined.update(cam_pos=cam_pos.clone(), rays=rays.clone(), p0=p0.clone(), bi=bi.clone())
        p1, check = real_refiner(cam_pos, rays, p0, bi, *a, **k)
        refined.update(p1=p1.detach().clone(), check=check.clone())
        return p1, check
    torch.manual_seed(1234)
    torch.rand, torch.randn_like = rec_rand, rec_randn_like
    ref.utils.OptimizeSurfacePs = rec_refiner
    try:
        loss = net(datas, SP, RATIO, fids)
    finally:
        torch.rand, torch.randn_like = real_rand, real_randn_like
        ref.utils.OptimizeSurfacePs = real_refiner
    kinds = [k for k, _ in draws]
    print('draws:', [(k, tuple(t.shape)) for k, t in draws])
    assert kinds == ['rand', 'rand', 'randn_like', 'rand', 'rand', 'randn_like'], kinds
    rand = dict(ray_select=draws[0][1], vert_select=draws[1][1], eik_local=draws[2][1], eik_global=draws[3][1], vert_select2=draws[4][1], regu_local=draws[5][1])
    info = dict(net.info)
    print({k: v for k, v in info.items() if k != 'pc_loss'}, info['pc_loss'])
    assert info['rayInfo'][1] > 10
    loss.backward()
    tmpps_grad = net.TmpPs.grad.clone()
    net.propagateTmpPsGrad(fids, RATIO)
    sp, tp, rp = dict(sdf.named_parameters()), dict(tr.named_parameters()), dict(rn.named_parameters())
    arrs = dict(fids=fids, V0=V0, faces=faces, V1=net.TmpVs.detach(), img=datas['img'], mask=datas['mask'], normal=datas['normal'],
                poses=ds.poses.detach(), trans=ds.trans.detach(), dcond=ds.conds[0].detach(), rcond=ds.conds[1].detach(),
                focal=ds.focal.detach(), princ=ds.princ.detach(), T=ds.T.detach(), R=ds.R[0], HW=np.array([H, W]), SP=np.array(SP), radius=np.array(0.06),
                ang_thr=np.array(net.angThred), loss=loss.detach(),
                ray_info=np.array(info['rayInfo']), inv_info=np.array(net.info['invInfo']),
                bi=net.batch_inds, rows=net.row_inds, cols=net.col_inds, TmpPs=net.TmpPs.detach(), g_TmpPs=tmpps_grad,
                sel_bi=refined['bi'], sel_rays=refined['rays'], sel_p0=refined['p0'], sel_p1=refined['p1'], sel_check=refined['check'], cam_pos=refined['cam_pos'],
                **{'rand_' + k: v for k, v in rand.items()},
                **{'L_' + k: np.array(info[k]) for k in ('grad_loss', 'def_loss', 'dct_loss', 'color_loss', 'normal_loss', 'offset_loss', 'pc_loss_sdf')},
                L_mask_loss=np.array(info['pc_loss']['mask_loss']), L_defconst_loss=np.array(info['pc_loss']['defconst_loss']),
                g_poses=ds.poses.grad, g_trans=ds.trans.grad, g_dcond=ds.conds[0].grad, g_focal=ds.focal.grad, g_princ=ds.princ.grad, g_T=ds.T.grad,
                g_sdf_v0=sp['lin0.weight_v'].grad[::37, ::5], g_sdf_g4=sp['lin4.weight_g'].grad, g_sdf_b7=sp['lin7.bias'].grad, g_sdf_v8=sp['lin8.weight_v'].grad[::16, ::7],
                g_tr_w0=tp['lin0.weight'].grad[::41, ::9], g_tr_w2=tp['lin2.weight'].grad[::53, ::47], g_tr_b4=tp['lin4.bias'].grad, g_tr_w4=tp['lin4.weight'].grad[:, ::11],
                g_rn_v0=rp['lin0.weight_v'].grad[::31, ::13], g_rn_g2=rp['lin2.weight_g'].grad, g_rn_b4=rp['lin4.bias'].grad)
    conv = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(OUT, "iteration.npz"), **conv)
    print("wrote iteration", {k: v.shape for k, v in conv.items()})
    print('rcond grad', None if ds.conds[1].grad is None else float(ds.conds[1].grad.abs().max()))


if __name__ == "__main__":
    main()
