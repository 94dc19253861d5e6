"""TEST INFRASTRUCTURE ONLY -- a SECOND, independent witness for the two rasterisation calls of the reference's iteration
(model/network.py:492,497: pytorch3d 0.4.0 `MeshRasterizer` and `PointsRasterizer` + `AlphaCompositor`).

PARITY UNPINNED, still: pytorch3d is neither in the reference repository nor installed.  oracle/raster_oracle.py restates its CUDA
kernels operation by operation in float32; this file is written WITHOUT looking at that restatement, from the renderer's DOCUMENTED
semantics only, in float64 and by brute force (every pixel against every primitive), with other formulas wherever the documentation
leaves the formula open:

  mesh    a pixel belongs to a face when its centre lies strictly inside the face's projected triangle; of the faces a pixel belongs
          to, the one with the smallest perspective-correct depth wins (faces_per_pixel = 1, blur_radius = 0, no culling of back
          faces); barycentric coordinates are perspective-correct:  b_i = (l_i / z_i) / sum_j (l_j / z_j)  with l the screen-space
          barycentrics (the textbook form; pytorch3d's kernel multiplies through by z_0 z_1 z_2), depth = 1 / sum_j (l_j / z_j);
          screen-space barycentrics from the SOLUTION OF THE 2 x 2 SYSTEM  p - v2 = l0 (v0 - v2) + l1 (v1 - v2)  (not edge functions);
          faces entirely behind the camera (all z < 0) are dropped; pixel (row, col) has its centre at NDC
          (1 - (2 col + 1) / W, 1 - (2 row + 1) / H): +x left, +y up;
  points  a pixel is covered by a point when the squared NDC distance of its centre to the point is < radius^2 and the point is in
          front of the camera (z >= 0); the points_per_pixel = K covering points nearest in z are kept; alpha_k = 1 - d_k^2 / r^2;
          `AlphaCompositor` with one all-ones feature: sum_k alpha_k prod_{j<k} (1 - alpha_j), front to back.

Decisions that sit on a float boundary (a pixel centre on an edge, two faces at the same depth, a point at distance exactly r, a tie in
z around rank K) cannot be the same in float32 and float64; every function therefore also returns an AMBIGUITY mask -- pixels whose
decision margin is below `eps` -- and callers compare outside it.  What the documentation does not define (the coverage of faces with
some but not all vertices behind the camera: pytorch3d 0.4.0 does not clip) is marked ambiguous as a whole."""
import numpy as np


def pixel_centres(H, W):
    xs = 1.0 - (2.0 * np.arange(W, dtype=np.float64) + 1.0) / W
    ys = 1.0 - (2.0 * np.arange(H, dtype=np.float64) + 1.0) / H
    return xs, ys


def mesh(verts, faces, H, W, eps=1e-5):
    """verts [V,3] = (x_ndc, y_ndc, z_view), faces [F,3] (rows with a negative index are skipped).
    -> face [H,W] (index or -1), bary [H,W,3], depth [H,W], ambiguous [H,W] bool."""
    v = np.asarray(verts, np.float64); faces = np.asarray(faces, np.int64)
    xs, ys = pixel_centres(H, W)
    PX, PY = np.meshgrid(xs, ys)                                  # [H,W]
    best = np.full((H, W), np.inf); second = np.full((H, W), np.inf)
    face = -np.ones((H, W), np.int64); bary = np.zeros((H, W, 3))
    amb = np.zeros((H, W), bool)
    for f, (i0, i1, i2) in enumerate(faces):
        if min(i0, i1, i2) < 0:
            continue
        p0, p1, p2 = v[i0], v[i1], v[i2]
        zs = np.array([p0[2], p1[2], p2[2]])
        if (zs < 0).all():
            continue
        # p - p2 = l0 (p0 - p2) + l1 (p1 - p2): Cramer's rule
        a, b, c, d = p0[0] - p2[0], p1[0] - p2[0], p0[1] - p2[1], p1[1] - p2[1]
        det = a * d - b * c
        box = (PX >= min(p0[0], p1[0], p2[0]) - eps) & (PX <= max(p0[0], p1[0], p2[0]) + eps) & (PY >= min(p0[1], p1[1], p2[1]) - eps) & (PY <= max(p0[1], p1[1], p2[1]) + eps)
        if abs(det) < 1e-7:
            amb |= box                                             # (nearly) degenerate in the image: whether an implementation drops it is its own choice
            continue
        rx, ry = PX - p2[0], PY - p2[1]
        l0 = (rx * d - b * ry) / det
        l1 = (a * ry - rx * c) / det
        l2 = 1.0 - l0 - l1
        if (zs < 0).any() or (np.abs(zs) < 1e-9).any():
            amb |= box                                             # not clipped by pytorch3d 0.4.0: coverage of such faces is not defined by the documentation
            continue
        inv = l0 / zs[0] + l1 / zs[1] + l2 / zs[2]
        inside = (l0 > 0) & (l1 > 0) & (l2 > 0) & box
        near_edge = (np.minimum(np.minimum(np.abs(l0), np.abs(l1)), np.abs(l2)) < eps) & box
        amb |= near_edge
        with np.errstate(divide='ignore', invalid='ignore'):
            depth = np.where(inside, 1.0 / inv, np.inf)
        better = depth < best
        second = np.where(better, best, np.minimum(second, depth))
        bb = np.stack([l0 / zs[0], l1 / zs[1], l2 / zs[2]], -1) / np.where(inside, inv, 1.0)[..., None]
        bary = np.where(better[..., None], bb, bary)
        face = np.where(better, f, face)
        best = np.where(better, depth, best)
    with np.errstate(invalid="ignore"):
        amb |= np.isfinite(second) & (second - best < eps * np.maximum(1.0, np.abs(best)))   # two faces at (nearly) the same depth
    depth = np.where(face >= 0, best, -1.0)
    return face, bary, depth, amb


def points(xy, z, H, W, radius, K=50, eps=1e-6):
    """xy [V,2] NDC, z [V] view depth -> mask [H,W] = 1 - prod over the K nearest covering points of d^2 / r^2 (the composite of
    alpha = 1 - d^2 / r^2 with an all-ones feature), count [H,W] of covering points, ambiguous [H,W] bool."""
    xy = np.asarray(xy, np.float64); z = np.asarray(z, np.float64)
    xs, ys = pixel_centres(H, W)
    r2 = float(radius) ** 2
    mask = np.zeros((H, W)); count = np.zeros((H, W), np.int64); amb = np.zeros((H, W), bool)
    ok = z >= 0
    for r in range(H):
        dy2 = (ys[r] - xy[:, 1]) ** 2
        cand = np.nonzero(ok & (dy2 < r2 * (1 + 1e-3) + 1e-12))[0]
        if cand.size == 0:
            continue
        d2 = (xs[None, :] - xy[cand, 0][:, None]) ** 2 + dy2[cand][:, None]          # [c, W]
        cov = d2 < r2
        amb[r] |= (np.abs(d2 - r2) < eps * r2).any(0)
        for c in np.nonzero(cov.any(0))[0]:
            who = cand[cov[:, c]]
            dd = d2[cov[:, c], c]
            order = np.argsort(z[who], kind='stable')
            count[r, c] = who.size
            if who.size > K:
                zs = z[who][order]
                if zs[K] - zs[K - 1] < eps * max(1.0, abs(zs[K - 1])):
                    amb[r, c] = True                                   # a tie in z around rank K: which point is kept is an implementation detail
                order = order[:K]
            alpha = 1.0 - dd[order] / r2
            acc, trans = 0.0, 1.0
            for a in alpha:                                            # front to back, as the compositor is documented
                acc += trans * a
                trans *= 1.0 - a
            mask[r, c] = acc
    return mask, count, amb
