"""Deterministic, RNG-free synthetic inputs for the hot path (SURVEY.md 8(d) "Synthetic inputs").

No dataset or checkpoint ships with the reference (SMPL assets / PeopleSnapshot are licensed
and absent), so tests, smoke() and bench.py build their inputs here: a splitmix64 integer
hash gives bit-identical tensors on every machine without touching torch's RNG.
"""
import numpy as np
import torch

SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
LBS_BMIN = [-0.8, -1.25, -0.4]
LBS_BMAX = [0.8, 0.95, 0.4]

# (name, out, in, weight_norm) per Linear -- shapes probed from the reference
SDF_SPEC = [("lin0", 512, 39, True), ("lin1", 512, 512, True), ("lin2", 512, 512, True), ("lin3", 473, 512, True),
            ("lin4", 512, 512, True), ("lin5", 512, 512, True), ("lin6", 512, 512, True), ("lin7", 512, 512, True),
            ("lin8", 257, 512, True)]
DEF_SPEC = [("lin0", 512, 167, False), ("lin1", 512, 512, False), ("lin2", 512, 512, False), ("lin3", 512, 512, False),
            ("lin4", 3, 512, False)]
REND_SPEC = [("lin0", 512, 289, True), ("lin1", 512, 512, True), ("lin2", 512, 512, True), ("lin3", 512, 512, True),
             ("lin4", 3, 512, True)]


def _splitmix(idx, seed):
    with np.errstate(over="ignore"):
        h = idx.astype(np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        h ^= h >> np.uint64(30)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(27)
        h *= np.uint64(0x94D049BB133111EB)
        h ^= h >> np.uint64(31)
    return h


def det_array(shape, seed, scale=1.0, dtype=np.float32):
    """Uniform in [-scale, scale), a pure function of (shape, seed)."""
    n = int(np.prod(shape))
    h = _splitmix(np.arange(n, dtype=np.uint64), seed)
    u = (h >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)
    return ((u * 2.0 - 1.0) * scale).astype(dtype).reshape(shape)


def det_tensor(shape, seed, scale=1.0, dtype=torch.float32):
    return torch.from_numpy(det_array(shape, seed, scale, np.float64)).to(dtype)


def det_params(spec, seed, last_scale=None):
    """state_dict (reference key names) filled with bounded pseudo-random values that keep
    activations O(1): rows of variance 2/out (the reference's own init scale, network.py:60-63)."""
    sd = {}
    for li, (name, out, inp, wn) in enumerate(spec):
        std = np.sqrt(2.0 / out) if out > 8 else 0.02
        w = det_tensor((out, inp), seed * 100 + li * 3, np.sqrt(3.0) * std)
        b = det_tensor((out,), seed * 100 + li * 3 + 1, 0.05)
        if last_scale is not None and li == len(spec) - 1:
            w, b = w * last_scale, b * last_scale
        if wn:
            sd[f"{name}.weight_g"] = 1.0 + 0.3 * det_tensor((out, 1), seed * 100 + li * 3 + 2, 1.0)
            sd[f"{name}.weight_v"] = w
        else:
            sd[f"{name}.weight"] = w
        sd[f"{name}.bias"] = b
    return sd


def synthetic_joints():
    """Fixed 24-point stick figure in the LBS box, SMPL joint order."""
    J = np.zeros((24, 3), np.float32)
    J[0] = [0.0, -0.25, 0.0]                      # pelvis
    J[1], J[2] = [0.09, -0.33, 0.0], [-0.09, -0.33, 0.0]
    J[3] = [0.0, -0.12, -0.02]
    J[4], J[5] = [0.10, -0.70, 0.0], [-0.10, -0.70, 0.0]
    J[6] = [0.0, 0.02, -0.01]
    J[7], J[8] = [0.09, -1.08, -0.03], [-0.09, -1.08, -0.03]
    J[9] = [0.0, 0.08, 0.0]
    J[10], J[11] = [0.11, -1.14, 0.08], [-0.11, -1.14, 0.08]
    J[12] = [0.0, 0.28, -0.02]
    J[13], J[14] = [0.08, 0.19, -0.01], [-0.08, 0.19, -0.01]
    J[15] = [0.0, 0.37, 0.03]
    J[16], J[17] = [0.19, 0.21, -0.02], [-0.19, 0.21, -0.02]
    J[18], J[19] = [0.44, 0.20, -0.03], [-0.44, 0.20, -0.03]
    J[20], J[21] = [0.69, 0.21, -0.02], [-0.69, 0.21, -0.02]
    J[22], J[23] = [0.77, 0.20, -0.03], [-0.77, 0.20, -0.03]
    return torch.from_numpy(J)


def synthetic_lbs_volume(shape_dhw=(65, 225, 129), device="cpu", chunk=1 << 20):
    """ws = softmax_j(-4 |voxel - J_j|^2 / 0.15^2), shape (1,24,D,H,W); voxel centres follow
    the reference's align_corners=False convention (model/Deformer.py:246-262)."""
    D, H, W = shape_dhw
    J = synthetic_joints().to(device)
    bmin = torch.tensor(LBS_BMIN, device=device)
    bmax = torch.tensor(LBS_BMAX, device=device)
    zs, ys, xs = torch.meshgrid(torch.arange(D, device=device), torch.arange(H, device=device),
                                torch.arange(W, device=device), indexing="ij")
    c = torch.stack([xs, ys, zs], -1).reshape(-1, 3).float()
    res = torch.tensor([W, H, D], device=device).float()
    c = (c / res + 0.5 / res) * (bmax - bmin) + bmin
    outs = []
    for part in torch.split(c, chunk):
        d2 = ((part[:, None, :] - J[None]) ** 2).sum(-1)
        outs.append(torch.softmax(-4.0 * d2 / (0.15 ** 2), dim=1))
    w = torch.cat(outs, 0)
    return w.t().reshape(1, 24, D, H, W).contiguous()


def det_normal(shape, seed, std=1.0, mean=0.0):
    """Approximately normal (Irwin-Hall, 4 uniforms): adds/muls only, so bit-reproducible."""
    s = sum(det_array(shape, seed * 4 + k, 1.0, np.float64) for k in range(4)) * (np.sqrt(3.0) / 2.0)
    return torch.from_numpy(s * std + mean).float()


def sphere_sdf_params(seed=7, bias=0.6, multires=6, feat=256):
    """A state_dict following the reference's geometric initialisation rules
    (model/network.py:49-63: sphere of radius `bias`), drawn from det_normal instead of
    torch's RNG so that tests / bench get the same near-sphere SDF everywhere."""
    dims = [3 + 6 * multires] + [512] * 8 + [1 + feat]
    skip_in = (4,)
    sd = {}
    nl = len(dims)
    for l in range(nl - 1):
        out_dim = dims[l + 1] - dims[0] if (l + 1) in skip_in else dims[l + 1]
        k = dims[l]
        if l == nl - 2:
            w = det_normal((out_dim, k), seed * 50 + l, 0.0001, np.sqrt(np.pi) / np.sqrt(k))
            b = torch.full((out_dim,), -bias)
        elif l == 0:
            w = torch.zeros(out_dim, k)
            w[:, :3] = det_normal((out_dim, 3), seed * 50 + l, np.sqrt(2) / np.sqrt(out_dim))
            b = torch.zeros(out_dim)
        elif l in skip_in:
            w = det_normal((out_dim, k), seed * 50 + l, np.sqrt(2) / np.sqrt(out_dim))
            w[:, -(dims[0] - 3):] = 0.0
            b = torch.zeros(out_dim)
        else:
            w = det_normal((out_dim, k), seed * 50 + l, np.sqrt(2) / np.sqrt(out_dim))
            b = torch.zeros(out_dim)
        sd[f"lin{l}.weight_g"] = w.double().norm(dim=1, keepdim=True).float()
        sd[f"lin{l}.weight_v"] = w
        sd[f"lin{l}.bias"] = b
    return sd


def icosphere(levels=2):
    t = (1.0 + 5 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2),
         (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, dtype=np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(levels):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m)); cache[key] = len(v) - 1
            return cache[key]
        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return torch.tensor(np.stack(v), dtype=torch.float32), torch.tensor(f, dtype=torch.long)


def cube_sphere(n):
    """Closed triangle mesh of the unit sphere from a cube with n x n cells per face: V = 6 n^2 + 2 vertices, F = 12 n^2 faces,
    outward winding.  A pure function of n (float64 numpy, exact integer lattice) -- the template of the full-size parity
    fixtures (n = 119 -> 84 968 vertices, the size marching cubes gives the coarse stage; n = 170 -> 173 402, fine stage)."""
    g = np.arange(n + 1)
    u, v = np.meshgrid(g, g, indexing="ij")
    z0, zn = np.zeros_like(u), np.full_like(u, n)
    # (lattice coordinates, and whether (du x dv) points outward) for the six faces
    sides = [(np.stack([z0, u, v], -1), False), (np.stack([zn, u, v], -1), True), (np.stack([u, z0, v], -1), True),
             (np.stack([u, zn, v], -1), False), (np.stack([u, v, z0], -1), False), (np.stack([u, v, zn], -1), True)]
    keys, quads = [], []
    for lat, outward in sides:
        k = (lat[..., 0] * (n + 1) + lat[..., 1]) * (n + 1) + lat[..., 2]
        base = len(keys) * (n + 1) ** 2
        idx = base + np.arange((n + 1) ** 2).reshape(n + 1, n + 1)
        a, b, c, d = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
        tri = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], 2).reshape(-1, 3)
        quads.append(tri if outward else tri[:, ::-1])
        keys.append(k.reshape(-1))
    keys = np.concatenate(keys)
    uniq, inv = np.unique(keys, return_inverse=True)
    faces = inv[np.concatenate(quads)]
    lat = np.stack([uniq // (n + 1) ** 2, (uniq // (n + 1)) % (n + 1), uniq % (n + 1)], -1).astype(np.float64)
    p = lat * (2.0 / n) - 1.0
    p = np.tan(p * (np.pi / 4.0))                              # equal-angle warp: cells of nearly equal size on the sphere
    dirs = p / np.linalg.norm(p, axis=1, keepdims=True)
    return torch.from_numpy(dirs).float(), torch.from_numpy(faces.astype(np.int64))


# ------------------------------------------------------------------------------------------------
# Synthetic sequence + scene builder: the tensor contract of
# dataset/dataset.py (poses / trans / per-frame codes as dense learnable tensors, one camera),
# resident on the GPU instead of being re-uploaded from the host at every call.
class SyntheticSequence:
    def __init__(self, frame_num=64, H=540, W=540, device="cuda:0", seed=0):
        self.frame_num, self.H, self.W = frame_num, H, W
        self.device = torch.device(device)
        dev = self.device
        # smooth pose / translation tracks (low-frequency sines so that the DCT term is meaningful)
        t = torch.linspace(0, 1, frame_num).view(-1, 1, 1)
        ph = det_tensor((1, 24, 3), 900 + seed, 3.14)
        self.poses = (0.12 * torch.sin(2 * np.pi * t + ph) * det_tensor((1, 24, 3), 901 + seed, 1.0)).to(dev).requires_grad_(True)
        self.trans = (0.04 * torch.sin(2 * np.pi * t.view(-1, 1) + det_tensor((1, 3), 902 + seed, 3.14))).to(dev).requires_grad_(True)
        self.shape = torch.zeros(10, device=dev)
        self.conds = [(0.1 * det_tensor((frame_num, 128), 903 + seed, 1.0)).to(dev).requires_grad_(True),
                      (0.1 * det_tensor((frame_num, 256), 904 + seed, 1.0)).to(dev).requires_grad_(True)]
        self.cond_ns = ['deformer', 'render']
        f = 1.2 * max(H, W)
        self.camera_params = {'focal_length': torch.tensor([f, f], device=dev),
                              'princeple_points': torch.tensor([W / 2.0, H / 2.0], device=dev),
                              'cam2world_coord_quat': torch.tensor([0., 0., 1., 0.], device=dev),   # R = diag(-1, 1, -1)
                              'world2cam_coord_trans': torch.tensor([0., 0.15, 2.4], device=dev)}
        self.video_segmented_index = []
        self._R_cache = None

    def opt_camera_params(self, conf):
        """dataset/dataset.py:64-74: which camera parameters are optimised (a bool for all, or train.opt_camera of the config)."""
        names = {'focal_length': 'focal_length', 'princeple_points': 'princeple_points', 'cam2world_coord_quat': 'quat', 'world2cam_coord_trans': 'T'}
        for key, cname in names.items():
            self.camera_params[key].requires_grad_(bool(conf) if isinstance(conf, bool) else conf.get_bool(cname))

    def learnable_weights(self):
        ws = [c for c in self.conds if c.requires_grad]
        ws += [v for v in self.camera_params.values() if v.requires_grad]
        ws += [v for v in (self.poses, self.trans) if v.requires_grad]
        return ws

    def get_grad_parameters(self, idxs, device=None):
        """dataset/dataset.py:117-122 (rows of the per-frame tables; index_select: its backward is one index_add)."""
        idxs = idxs.view(-1)
        return (torch.index_select(self.poses, 0, idxs), torch.index_select(self.trans, 0, idxs), torch.index_select(self.conds[0], 0, idxs),
                torch.index_select(self.conds[1], 0, idxs))

    def get_camera_parameters(self, N, device=None):
        """dataset/dataset.py:125-127.  The rotation of a quaternion that is not optimised (config.conf:13) is built once."""
        q = self.camera_params['cam2world_coord_quat'].view(1, 4)
        key = (q.data_ptr(), q._version)
        if not q.requires_grad and self._R_cache is not None and self._R_cache[0] == key:
            R = self._R_cache[1]
        else:
            q = q / q.norm(p=2, dim=1, keepdim=True)
            w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
            R = torch.stack([w * w + x * x - y * y - z * z, 2 * x * y - 2 * w * z, 2 * w * y + 2 * x * z,
                             2 * w * z + 2 * x * y, w * w - x * x + y * y - z * z, 2 * y * z - 2 * w * x,
                             2 * x * z - 2 * w * y, 2 * w * x + 2 * y * z, w * w - x * x - y * y + z * z], dim=1).view(1, 3, 3)
            if not q.requires_grad:
                self._R_cache = (key, R)
        return (self.camera_params['focal_length'].view(1, 2).expand(N, 2), self.camera_params['princeple_points'].view(1, 2).expand(N, 2),
                R.expand(N, 3, 3), self.camera_params['world2cam_coord_trans'].view(1, 3).expand(N, 3), self.H, self.W)

    def get_batchframe_data(self, name, fids, batchsize):
        """Window of `batchsize` frames around each id, clamped to the sequence (dataset.py:128-147)."""
        assert batchsize < self.frame_num
        data = getattr(self, name)
        starts = (fids - batchsize // 2).clamp(min=0, max=self.frame_num - batchsize)
        return data[starts.view(-1, 1) + torch.arange(0, batchsize, device=fids.device).view(1, batchsize)], fids - starts

    def attach_consistent_masks(self, net, ratio):
        """Ground-truth silhouettes that agree with the scene: the initial template, posed for every frame and splatted exactly as
        the training step does, thresholded at 0.5.  With the analytic ellipse the mask IoU loss starts at 0.5, the template SGD
        step tears the template off the SDF zero set (|f(TmpVs)| 1e-5 -> 0.13 after ONE step) and hardly a ray converges in the
        refiner any more -- a workload no real sequence produces: there the masks match the body and nearly every ray converges."""
        with torch.no_grad():
            verts, _ = net.discretizeSDF(ratio, None, -net.sdfShrinkRadius)
            cameras, H, W = net._cameras(1, self.device)
            masks = []
            for f in range(self.frame_num):
                defconds = [self.conds[0][f:f + 1], [self.poses[f:f + 1], self.trans[f:f + 1]]]
                dv = net.deformer(verts[None], defconds, ratio=ratio)
                masks.append(net._silhouette(dv, cameras, H, W, net.point_radius)[0] > 0.5)
            self._masks = torch.stack(masks)

    def attach_rendered_observations(self, net, ratio, frames_per_call=4):
        """Observations that the scene itself explains: colour, normal and silhouette images of EVERY frame rendered from the
        model as it is now (OptimNetwork.render_frames = the colour pass of the reference's `infer`, network.py:340-372).
        With them the colour / normal / mask terms sit at their optimum for the current weights, which is where a real
        sequence spends almost all of its ~10^5 iterations; uniform-noise targets (the default of `batch`) keep those terms at
        O(1) with gradients that never average out."""
        imgs, normals, masks = [], [], []
        for f0 in range(0, self.frame_num, frames_per_call):
            fids = torch.arange(f0, min(f0 + frames_per_call, self.frame_num), device=self.device)
            out = net.render_frames(fids, ratio)
            imgs.append(out['img']); normals.append(out['normal']); masks.append(out['mask'] > 0.5)
        self._imgs, self._normals, self._masks = torch.cat(imgs), torch.cat(normals), torch.cat(masks)

    def batch(self, frame_ids):
        """Synthetic observations: an elliptic ground-truth silhouette around the projected body and uniform-noise
        colour / normal images of the right shape."""
        N, H, W, dev = len(frame_ids), self.H, self.W, self.device
        if getattr(self, "_imgs", None) is not None:           # rendered from the scene itself, see attach_rendered_observations
            return {'img': self._imgs[frame_ids], 'mask': self._masks[frame_ids].float(), 'normal': self._normals[frame_ids]}
        g = torch.Generator(device=dev); g.manual_seed(1234 + int(frame_ids[0]))
        if getattr(self, "_masks", None) is not None:          # self-consistent observations, see attach_consistent_masks
            return {'img': torch.rand((N, H, W, 3), device=dev, generator=g) * 2 - 1, 'mask': self._masks[frame_ids].float(),
                    'normal': torch.rand((N, H, W, 3), device=dev, generator=g) * 2 - 1}
        ys, xs = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing='ij')
        cp = {k: v.detach() for k, v in self.camera_params.items()}
        f = float(cp['focal_length'][0]); cx = float(cp['princeple_points'][0]); cy = float(cp['princeple_points'][1])
        Tz = float(cp['world2cam_coord_trans'][2])
        masks = []
        for i in range(N):
            tr = self.trans[int(frame_ids[i])].detach()
            ux = cx + f * float(tr[0]) / Tz
            uy = cy - f * (float(tr[1]) + float(cp['world2cam_coord_trans'][1])) / Tz
            rx, ry = f * 0.56 / Tz, f * 0.63 / Tz
            masks.append((((xs - ux) / rx) ** 2 + ((ys - uy) / ry) ** 2 < 1.0).float())
        return {'img': torch.rand((N, H, W, 3), device=dev, generator=g) * 2 - 1, 'mask': torch.stack(masks),
                'normal': torch.rand((N, H, W, 3), device=dev, generator=g) * 2 - 1}


COARSE_RESOLUTIONS = [(14 + 1, 20 + 1, 8 + 1), (28 + 1, 40 + 1, 16 + 1), (56 + 1, 80 + 1, 32 + 1), (112 + 1, 160 + 1, 64 + 1),
                      (224 + 1, 320 + 1, 128 + 1)]                                    # train.py:29-35 (W,H,D)
MEDIUM_RESOLUTIONS = [(18 + 1, 24 + 1, 12 + 1), (36 + 1, 48 + 1, 24 + 1), (72 + 1, 96 + 1, 48 + 1), (144 + 1, 192 + 1, 96 + 1),
                      (288 + 1, 384 + 1, 192 + 1)]                                    # train.py:37-43
FINE_RESOLUTIONS = [(20 + 1, 26 + 1, 14 + 1), (40 + 1, 52 + 1, 28 + 1), (80 + 1, 104 + 1, 56 + 1), (160 + 1, 208 + 1, 112 + 1),
                    (320 + 1, 416 + 1, 224 + 1)]                                      # train.py:45-51
STAGE_RESOLUTIONS = {'coarse': COARSE_RESOLUTIONS, 'medium': MEDIUM_RESOLUTIONS, 'fine': FINE_RESOLUTIONS}


def build_synthetic_scene(device="cuda:0", frame_num=64, H=540, W=540, stage="coarse", resolutions=None, lbs_volume_shape=(65, 225, 129),
                          conf=None, seed=0, consistent_masks=True, opt_camera="config"):
    """SDF (near-sphere geometric init), deformer (MLPTranslator + LBS on a synthetic weight volume), render net,
    Seg3dLossless engine, orchestrator and dataset, wired like model/network.py::getOptNet (:828-909)."""
    from .config import default_config
    from .model.network import getTmpSdf
    from .model.Deformer import MLPTranslator, LBSkinner, CompositeDeformer
    from .model.RenderNet import RenderingNetwork_view_norm
    from .model.optim_network import OptimNetwork
    from .MCAcc import Seg3dLossless
    from .utils.utils import smpl_tmp_Apose, DCTNullSpace
    conf = conf or default_config()
    sdf = getTmpSdf(device, conf.get_int('sdf_net.multires'), 0.6, conf.get_int('render_net.condlen'))
    sdf.load_state_dict(sphere_sdf_params(7 + seed), strict=True)
    torch.manual_seed(seed)
    tr = MLPTranslator(conf.get_int('mlp_deformer.condlen'), conf.get_int('mlp_deformer.multires')).to(device)
    vol = synthetic_lbs_volume(lbs_volume_shape, device=device)
    skin = LBSkinner(vol, LBS_BMIN, LBS_BMAX, synthetic_joints(), np.array(SMPL_PARENTS),
                     init_pose=torch.from_numpy(smpl_tmp_Apose(conf.get_int('train.skinner_pose_type'))), align_corners=False).to(device)
    deformer = CompositeDeformer([tr, skin]).to(device)
    rend = RenderingNetwork_view_norm(conf.get_int('render_net.condlen'), 'idr', 9, 3, [512, 512, 512, 512], True,
                                      multires_n=conf.get_int('render_net.multires_n'), multires_v=conf.get_int('render_net.multires_v')).to(device)
    engine = Seg3dLossless(query_func=None, b_min=LBS_BMIN, b_max=LBS_BMAX, resolutions=resolutions or STAGE_RESOLUTIONS[stage],
                           align_corners=False, balance_value=0.0, use_cuda_impl=True).to(device)     # fused HIP upsample + candidate selection (same volume, same queries)
    net = OptimNetwork(sdf, deformer, engine, None, rend, conf=conf.get_config('loss_' + stage)).to(device)
    net.remesh_intersect = conf.get_int(f'train.{stage}.point_render.remesh_intersect')
    net.point_radius = conf.get_float(f'train.{stage}.point_render.radius')
    ds = SyntheticSequence(frame_num, H, W, device, seed)
    # train.py:86 -> dataset.opt_camera_params: the shipped configuration optimises focal length, principal point and T (config.conf:10-15)
    ds.opt_camera_params(conf.get_config('train.opt_camera') if opt_camera == "config" else opt_camera)
    net.dataset = ds
    net.dctnull = DCTNullSpace(10, 30).to(device)
    if consistent_masks:
        ds.attach_consistent_masks(net, {'sdfRatio': 1., 'deformerRatio': 0.5, 'renderRatio': 1.})
    return net, ds, conf
