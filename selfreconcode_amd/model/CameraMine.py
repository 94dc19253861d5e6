"""The closed-form camera pieces the hot path uses: drop-in for
RectifiedPerspectiveCameras.view_rays / cam_pos / project / angThreshold
(model/CameraMine.py:129-170).  The pytorch3d CamerasBase plumbing of the reference class is out of scope."""
import numpy as np
import torch

from .. import step_ops


class RectifiedPerspectiveCameras:
    def __init__(self, focal_length, principal_point, R, T, image_size, device=None):
        self.focal_length = focal_length.view(-1, 2)
        self.principal_point = principal_point.view(-1, 2)
        self.R = R.view(-1, 3, 3)
        self.T = T.view(-1, 3)
        self.image_size = torch.as_tensor(image_size).view(-1, 2)        # (W, H)

    def to(self, device):
        self.focal_length, self.principal_point = self.focal_length.to(device), self.principal_point.to(device)
        self.R, self.T = self.R.to(device), self.T.to(device)
        return self

    def view_rays(self, ps, cam_id=0):
        """ps [P,3] = (col, row, 1): v = normalize([(cx-u)/fx, (cy-w)/fy, 1]) R^T."""
        f, c = self.focal_length[cam_id], self.principal_point[cam_id]
        if step_ops.ENABLED_CAMERA and ps.is_cuda and ps.dim() == 2:
            return step_ops.ViewRays.apply(ps, self.R[cam_id], f, c)
        rays = torch.stack([-ps[:, 0] / f[0] + ps[:, 2] * c[0] / f[0], -ps[:, 1] / f[1] + ps[:, 2] * c[1] / f[1], ps[:, 2]], dim=1)
        rays = rays / torch.norm(rays, p=2, dim=1, keepdim=True)
        return rays.matmul(self.R[cam_id].transpose(0, 1))

    def project(self, ps, cam_id=0):
        """world points -> pixel (x, y) and camera-space depth."""
        pc = ps.matmul(self.R[cam_id]) + self.T[cam_id].view(1, 3)
        x = self.principal_point[cam_id, 0] - pc[..., 0] * self.focal_length[cam_id, 0] / pc[..., 2]
        y = self.principal_point[cam_id, 1] - pc[..., 1] * self.focal_length[cam_id, 1] / pc[..., 2]
        return torch.stack([x, y], dim=-1), pc[..., 2]

    def project_ndc(self, ps, cam_id=0):
        """world points -> (x, y) in pytorch3d's NDC frame (+x left, +y up) and view-space depth: the reference's
        get_projection_transform with screen-space intrinsics (model/CameraMine.py:44-70, _get_sfm_calibration_matrix
        :171-262: fx_ndc = fx / (W/2), px_ndc = 1 - 1/W - cx / (W/2)) followed by the rasterisers' `z = z_view` override."""
        W, H = float(self.image_size[cam_id, 0]), float(self.image_size[cam_id, 1])
        if step_ops.ENABLED_CAMERA and ps.is_cuda:
            return step_ops.ProjectNDC.apply(ps, self.R[cam_id], self.T[cam_id], self.focal_length[cam_id], self.principal_point[cam_id], W, H)
        pc = ps.matmul(self.R[cam_id]) + self.T[cam_id].view(1, 3)
        f, c = self.focal_length[cam_id], self.principal_point[cam_id]
        x = (f[0] / (W / 2.)) * pc[..., 0] / pc[..., 2] + (1. - 1. / W - c[0] / (W / 2.))
        y = (f[1] / (H / 2.)) * pc[..., 1] / pc[..., 2] + (1. - 1. / H - c[1] / (H / 2.))
        return torch.stack([x, y], dim=-1), pc[..., 2]

    def cam_pos(self, cam_id=0):
        return -self.R[cam_id].matmul(self.T[cam_id].view(-1, 1)).view(-1)

    def angThreshold(self, pixoffset=0.4, cam_id=0):
        """Smallest angle (degrees) subtended by `pixoffset` pixels at the four image borders."""
        W, H = float(self.image_size[cam_id, 0]), float(self.image_size[cam_id, 1])
        cx, cy = (float(t) for t in self.principal_point[cam_id].detach())
        fx, fy = (float(t) for t in self.focal_length[cam_id].detach())

        def ang(a, b):
            a, b = torch.tensor(a), torch.tensor(b)
            return torch.arcsin(torch.linalg.cross(a, b).norm() / (a.norm() * b.norm())) / np.pi * 180.
        cands = [ang([(W - cx) / fx, 0., 1.], [(W + pixoffset - cx) / fx, 0., 1.]),
                 ang([-cx / fx, 0., 1.], [(pixoffset - cx) / fx, 0., 1.]),
                 ang([0., (H - cy) / fy, 1.], [0., (H + pixoffset - cy) / fy, 1.]),
                 ang([0., -cy / fy, 1.], [0., (pixoffset - cy) / fy, 1.])]
        return torch.stack(cands).min().item()
