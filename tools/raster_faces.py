"""Which faces make the mesh rasteriser's wave-per-face pass (rm_pass1b) expensive?  python tools/raster_faces.py
bench.py's coarse scene after a short settle: per (frame, face) the pixel-centre count of the bounding box and the projected area, for the
faces with more than 32 box pixels (the ones rm_pass1 hands to rm_pass1b)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from selfreconcode_amd import mlp_engine
from selfreconcode_amd.optim import FusedAdam
from selfreconcode_amd.synthetic import build_synthetic_scene

dev = torch.device('cuda:0')
net, ds, conf = build_synthetic_scene(device=dev, frame_num=64, stage='coarse', consistent_masks=False)
mlp_engine.set_deferred_param_grads(True)
opt = FusedAdam([{'params': ds.learnable_weights()}, {'params': [p for p in net.parameters() if p.requires_grad]}], lr=1e-4)
ratio = {'sdfRatio': 1., 'deformerRatio': 0.6, 'renderRatio': 1.}
ds.attach_rendered_observations(net, ratio)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    f = torch.arange(3 * it % 60, 3 * it % 60 + 3, device=dev)
    opt.zero_grad(set_to_none=True)
    loss = net(ds.batch(f), 2048, ratio, f)
    loss.backward(); net.propagateTmpPsGrad(f, ratio); opt.step()
with torch.no_grad():
    f = torch.arange(3, device=dev)
    poses, trans, d_cond, _ = ds.get_grad_parameters(f)
    cameras, H, W = net._cameras(3, dev)
    V = net.TmpVs.detach()
    defV = net.deformer(V[None].expand(3, -1, 3), [d_cond, [poses, trans]], ratio=ratio)
    xy, z = cameras.project_ndc(defV)
    F = net.Tmpfs
    p = xy[:, F]                                            # [3, F, 3, 2]
    xmin, xmax = p[..., 0].amin(-1), p[..., 0].amax(-1); ymin, ymax = p[..., 1].amin(-1), p[..., 1].amax(-1)
    bw = (torch.floor(((1 - xmin) * W - 1) / 2) - torch.ceil(((1 - xmax) * W - 1) / 2) + 1).clamp(min=0)
    bh = (torch.floor(((1 - ymin) * H - 1) / 2) - torch.ceil(((1 - ymax) * H - 1) / 2) + 1).clamp(min=0)
    box = bw * bh
    area = 0.5 * ((p[..., 1, 0] - p[..., 0, 0]) * (p[..., 2, 1] - p[..., 0, 1]) - (p[..., 1, 1] - p[..., 0, 1]) * (p[..., 2, 0] - p[..., 0, 0])).abs() * (W / 2) * (H / 2)
    big = box > 32
    print(f"faces per frame {F.shape[0]}, frames 3; faces with a box of > 32 pixel centres: {int(big.sum())}")
    for lo, hi in ((32, 100), (100, 1000), (1000, 10000), (10000, 1e9)):
        m = (box > lo) & (box <= hi)
        if int(m.sum()):
            print(f"  box in ({lo}, {hi:.0f}]: {int(m.sum()):6d} faces, box pixels {float(box[m].sum()):12.0f}, projected area {float(area[m].sum()):12.0f} px, "
                  f"longest side {float(torch.maximum(bw[m], bh[m]).max()):.0f}, median aspect (box / area) {float((box[m] / area[m].clamp(min=1e-3)).median()):.1f}")
    edge = (V[F[:, 0]] - V[F[:, 1]]).norm(dim=1)
    print("canonical edge length: median %.4f, max %.4f; deformed (frame 0): median %.4f, max %.4f" % (
        float(edge.median()), float(edge.max()), float((defV[0, F[:, 0]] - defV[0, F[:, 1]]).norm(dim=1).median()), float((defV[0, F[:, 0]] - defV[0, F[:, 1]]).norm(dim=1).max())))
