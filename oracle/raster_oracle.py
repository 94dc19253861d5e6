"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the two pytorch3d rasterisation calls on the reference's iteration.

PARITY UNPINNED: the arithmetic lives in a third-party dependency that is not under /root/reference and is not
installed here -- pytorch3d 0.4.0 (pinned by the reference's README.md:10,21-25).  What follows restates its published
algorithm (pytorch3d/csrc/rasterize_points/rasterize_points.cu, csrc/rasterize_meshes/rasterize_meshes.cu,
csrc/utils/geometry_utils.cuh, csrc/compositing/alpha_composite.cu, renderer/points/rasterizer.py,
renderer/mesh/rasterizer.py at tag v0.4.0) and is anchored on the reference's own call sites:

  * model/network.py:178-190, 497     PointsRasterizer(radius, points_per_pixel=50) + AlphaCompositor, called through
  * model/CameraMine.py:285-305       PointsRendererWithFrags.forward: weights = 1 - dists2 / r^2, composite, [N,H,W,C]
  * model/network.py:877-892, 492     MeshRasterizer(blur_radius=0, faces_per_pixel=1, perspective_correct=True,
                                      clip_barycentric_coords=False, cull_backfaces=False) -> Fragments for FindSurfacePs
  * model/CameraMine.py:44-70,171-262 RectifiedPerspectiveCameras: screen-space intrinsics -> NDC calibration matrix

Conventions restated here:
  - view coordinates  X_view = X_world R + T (row vectors);  x_ndc = fx_ndc X/Z + px_ndc with fx_ndc = fx / (W/2),
    px_ndc = 1 - 1/W - cx / (W/2)  (CameraMine.py:236-241); the rasterisers take z = Z_view (rasterizer.py `transform`);
  - NDC +x points LEFT and +y UP: pixel (row, col) has its centre at  x = 1 - (2 col + 1)/W,  y = 1 - (2 row + 1)/H
    (PixToNdc after the  yi = H-1-row, xi = W-1-col  flip of the kernels), i.e. col = cx - fx X/Z in pixel units;
  - points: skipped when z < 0; covered when dist2 < radius^2 (strict); the K nearest in z are kept, sorted by z;
  - compositing front to back:  out = sum_k a_k f_k prod_{j<k} (1 - a_j),  idx < 0 entries skipped;
  - faces: skipped when max z < 0, when |signed area| <= 1e-8, when the pixel is outside the bounding box, or when the
    (perspective-corrected) depth is < 0; inside test on the perspective-corrected barycentrics, all three > 0 (strict).
Only square images are restated (the reference's data are 540^2 / 1080^2; pytorch3d 0.4.0 rescales the NDC range of the
longer side of a non-square image, which the reference's camera class does not follow)."""
import numpy as np
import torch

K_EPS = np.float32(1e-8)          # kEpsilon of geometry_utils.cuh


def ndc_projection(points_world, focal, princ, R, T, W, H):
    """RectifiedPerspectiveCameras.transform_points + the rasterisers' z override.  points_world [...,3] (torch, any
    float dtype; differentiable) -> xy_ndc [...,2], z_view [...]."""
    view = points_world @ R + T.view(*([1] * (points_world.dim() - 1)), 3)
    fx, fy = focal[0] / (W / 2.0), focal[1] / (H / 2.0)
    px, py = 1. - 1. / W - princ[0] / (W / 2.0), 1. - 1. / H - princ[1] / (H / 2.0)
    z = view[..., 2]
    return torch.stack([fx * view[..., 0] / z + px, fy * view[..., 1] / z + py], -1), z


def pixel_centres_ndc(H, W):
    assert H == W, "square images only (see the module docstring)"
    xs = (1.0 - (2.0 * np.arange(W, dtype=np.float32) + 1.0) / np.float32(W)).astype(np.float32)
    ys = (1.0 - (2.0 * np.arange(H, dtype=np.float32) + 1.0) / np.float32(H)).astype(np.float32)
    return xs, ys


def rasterize_points_loop(xy_ndc, z, H, W, radius, K=50):
    """rasterize_points.cu (naive kernel; the coarse-to-fine path gives the same fragments while no bin overflows).
    xy_ndc [N,V,2], z [N,V] float32 arrays.  Returns idx [N,H,W,K] int64 (PACKED index n*V + v, -1 padded),
    zbuf [N,H,W,K], dists2 [N,H,W,K] (both -1 padded)."""
    xy = np.asarray(xy_ndc, np.float32); z = np.asarray(z, np.float32)
    N, V = z.shape
    xs, ys = pixel_centres_ndc(H, W)
    r2 = np.float32(radius) * np.float32(radius)
    idx = -np.ones((N, H, W, K), np.int64); zb = -np.ones((N, H, W, K), np.float32); d2o = -np.ones((N, H, W, K), np.float32)
    rad_px_x, rad_px_y = float(radius) * W / 2.0 + 1.0, float(radius) * H / 2.0 + 1.0
    for n in range(N):
        buckets = {}
        col = (1.0 - xy[n, :, 0].astype(np.float64)) * W / 2.0 - 0.5
        row = (1.0 - xy[n, :, 1].astype(np.float64)) * H / 2.0 - 0.5
        for v in range(V):
            if not (z[n, v] >= 0) or not np.isfinite(col[v]) or not np.isfinite(row[v]):
                continue
            c0, c1 = int(np.floor(col[v] - rad_px_x)), int(np.ceil(col[v] + rad_px_x))
            r0, r1 = int(np.floor(row[v] - rad_px_y)), int(np.ceil(row[v] + rad_px_y))
            for r in range(max(r0, 0), min(r1, H - 1) + 1):
                dy = ys[r] - xy[n, v, 1]
                for c in range(max(c0, 0), min(c1, W - 1) + 1):
                    dx = xs[c] - xy[n, v, 0]
                    dist2 = np.float32(dx * dx) + np.float32(dy * dy)
                    if dist2 < r2:
                        buckets.setdefault((r, c), []).append((z[n, v], v, dist2))
        for (r, c), lst in buckets.items():
            lst.sort(key=lambda t: (t[0], t[1]))          # nearest in z first; equal depths: lower point index first
            for k, (zz, v, dd) in enumerate(lst[:K]):
                idx[n, r, c, k] = n * V + v; zb[n, r, c, k] = zz; d2o[n, r, c, k] = dd
    return idx, zb, d2o


def alpha_composite(idx, alphas, features):
    """alpha_composite.cu forward.  idx [N,K,H,W] int64 packed, alphas [N,K,H,W], features [C,P] -> [N,C,H,W]."""
    N, K, H, W = idx.shape
    out = torch.zeros(N, features.shape[0], H, W, dtype=alphas.dtype)
    cum = torch.ones(N, H, W, dtype=alphas.dtype)
    for k in range(K):
        valid = idx[:, k] >= 0
        a = torch.where(valid, alphas[:, k], torch.zeros_like(alphas[:, k]))
        f = features[:, idx[:, k].clamp(min=0)]                       # [C,N,H,W]
        out = out + (cum * a).unsqueeze(1) * f.permute(1, 0, 2, 3)
        cum = cum * (1 - a)
    return out


def render_point_silhouette(points_world, focal, princ, R, T, H, W, radius, K=50):
    """PointsRendererWithFrags.forward of the reference (CameraMine.py:285-305) with one all-ones feature channel, as
    network.py:495-497 calls it: masks [N,H,W] (= imgs[..., -1] of computeTmpPcLoss :649).  Differentiable with respect
    to `points_world` through dists2 (what pytorch3d's rasterize_points backward propagates); the fragment selection is
    made once, without gradient."""
    N, V = points_world.shape[0], points_world.shape[1]
    xy, z = ndc_projection(points_world, focal, princ, R, T, W, H)
    idx, _, _ = rasterize_points(xy.detach().float().numpy(), z.detach().float().numpy(), H, W, radius, K)
    idx_t = torch.from_numpy(idx)
    xs, ys = pixel_centres_ndc(H, W)
    xf = torch.from_numpy(xs).to(xy.dtype).view(1, 1, W, 1); yf = torch.from_numpy(ys).to(xy.dtype).view(1, H, 1, 1)
    flat = xy.reshape(N * V, 2)
    p = flat[idx_t.clamp(min=0)]                                       # [N,H,W,K,2]
    dists2 = (xf - p[..., 0]) ** 2 + (yf - p[..., 1]) ** 2
    weights = 1 - dists2 / (radius * radius)
    feats = torch.ones(1, N * V, dtype=xy.dtype)
    img = alpha_composite(idx_t.permute(0, 3, 1, 2), weights.permute(0, 3, 1, 2), feats)
    return img[:, 0], idx_t


def _edge(px, py, ax, ay, bx, by):
    """EdgeFunctionForward(p, v0, v1) of geometry_utils.cuh."""
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax)


def rasterize_meshes_loop(verts_ndc, faces, H, W):
    """rasterize_meshes.cu CheckPixelInsideFace with blur_radius 0, faces_per_pixel 1, perspective_correct True,
    clip_barycentric_coords False, cull_backfaces False.  verts_ndc [N,V,3] float32 (x_ndc, y_ndc, z_view), faces [F,3].
    Returns pix_to_face [N,H,W,1] int64 (packed n*F + f, -1), bary [N,H,W,1,3], zbuf [N,H,W,1] (both -1 padded)."""
    v = np.asarray(verts_ndc, np.float32); faces = np.asarray(faces, np.int64)
    N, F = v.shape[0], faces.shape[0]
    xs, ys = pixel_centres_ndc(H, W)
    p2f = -np.ones((N, H, W, 1), np.int64); bary = -np.ones((N, H, W, 1, 3), np.float32); zbuf = -np.ones((N, H, W, 1), np.float32)
    best = np.full((N, H, W), np.inf, np.float32)
    f32 = np.float32
    for n in range(N):
        for f in range(F):
            if faces[f].min() < 0:        # MCGpu leaves -1 where the owner cell of a vertex lies outside the volume (CudaKernels.cu:470):
                continue                  # such faces cannot be rasterised (pytorch3d would index out of range); the product skips them too
            a, b, c = v[n, faces[f, 0]], v[n, faces[f, 1]], v[n, faces[f, 2]]
            if max(a[2], b[2], c[2]) < 0:
                continue
            area = _edge(a[0], a[1], b[0], b[1], c[0], c[1])            # EdgeFunctionForward(v0, v1, v2)
            if -K_EPS <= area <= K_EPS:
                continue
            xmin, xmax = min(a[0], b[0], c[0]), max(a[0], b[0], c[0])
            ymin, ymax = min(a[1], b[1], c[1]), max(a[1], b[1], c[1])
            cols = np.nonzero((xs >= xmin) & (xs <= xmax))[0]
            rows = np.nonzero((ys >= ymin) & (ys <= ymax))[0]
            if cols.size == 0 or rows.size == 0:
                continue
            px = xs[cols][None, :].astype(f32); py = ys[rows][:, None].astype(f32)
            den = f32(_edge(c[0], c[1], a[0], a[1], b[0], b[1]) + K_EPS)                          # BarycentricCoordsForward
            w0 = (_edge(px, py, b[0], b[1], c[0], c[1]) / den).astype(f32)
            w1 = (_edge(px, py, c[0], c[1], a[0], a[1]) / den).astype(f32)
            w2 = (_edge(px, py, a[0], a[1], b[0], b[1]) / den).astype(f32)
            t0, t1, t2 = (w0 * b[2] * c[2]).astype(f32), (a[2] * w1 * c[2]).astype(f32), (a[2] * b[2] * w2).astype(f32)
            dn = np.maximum((t0 + t1 + t2).astype(f32), K_EPS)                                    # BarycentricPerspectiveCorrectionForward
            b0, b1, b2 = (t0 / dn).astype(f32), (t1 / dn).astype(f32), (t2 / dn).astype(f32)
            pz = (b0 * a[2] + b1 * b[2] + b2 * c[2]).astype(f32)
            inside = (b0 > 0) & (b1 > 0) & (b2 > 0) & (pz >= 0)
            rr, cc = np.nonzero(inside)
            for i, j in zip(rr, cc):
                r, cpx = rows[i], cols[j]
                if pz[i, j] < best[n, r, cpx]:
                    best[n, r, cpx] = pz[i, j]
                    p2f[n, r, cpx, 0] = n * F + f
                    bary[n, r, cpx, 0] = (b0[i, j], b1[i, j], b2[i, j]); zbuf[n, r, cpx, 0] = pz[i, j]
    return p2f, bary, zbuf


# ------------------------------------------------------------------------------------------------
# Vectorised forms of the two loops above.  Same float32 operations in the same order per (primitive, pixel) pair, so the
# results are BIT-IDENTICAL to the loop versions (tests/test_oracle_golden.py::test_raster_oracle_vectorised_equals_loops holds
# them to that); they exist because the loops take minutes at 540x540 / 85k vertices / 170k faces (the full-size parity test).
def rasterize_points(xy_ndc, z, H, W, radius, K=50):
    xy = np.asarray(xy_ndc, np.float32); z = np.asarray(z, np.float32)
    N, V = z.shape
    xs, ys = pixel_centres_ndc(H, W)
    r2 = np.float32(radius) * np.float32(radius)
    idx = -np.ones((N, H, W, K), np.int64); zb = -np.ones((N, H, W, K), np.float32); d2o = -np.ones((N, H, W, K), np.float32)
    rad_px_x, rad_px_y = float(radius) * W / 2.0 + 1.0, float(radius) * H / 2.0 + 1.0
    win_x, win_y = int(np.ceil(2 * rad_px_x)) + 3, int(np.ceil(2 * rad_px_y)) + 3
    for n in range(N):
        col = (1.0 - xy[n, :, 0].astype(np.float64)) * W / 2.0 - 0.5
        row = (1.0 - xy[n, :, 1].astype(np.float64)) * H / 2.0 - 0.5
        with np.errstate(invalid="ignore"):
            ok = (z[n] >= 0) & np.isfinite(col) & np.isfinite(row)
        vid = np.nonzero(ok)[0]
        if vid.size == 0:
            continue
        c0 = np.floor(col[vid] - rad_px_x).astype(np.int64); c1 = np.ceil(col[vid] + rad_px_x).astype(np.int64)
        r0 = np.floor(row[vid] - rad_px_y).astype(np.int64); r1 = np.ceil(row[vid] + rad_px_y).astype(np.int64)
        assert int((c1 - c0).max()) < win_x and int((r1 - r0).max()) < win_y
        cc = c0[:, None] + np.arange(win_x)[None, :]; rr = r0[:, None] + np.arange(win_y)[None, :]
        cok = (cc <= c1[:, None]) & (cc >= 0) & (cc <= W - 1); rok = (rr <= r1[:, None]) & (rr >= 0) & (rr <= H - 1)
        dx = xs[np.clip(cc, 0, W - 1)] - xy[n, vid, 0][:, None]                      # [v, win_x] float32
        dy = ys[np.clip(rr, 0, H - 1)] - xy[n, vid, 1][:, None]
        dist2 = (dx * dx)[:, None, :] + (dy * dy)[:, :, None]                        # [v, win_y, win_x] float32
        hit = (dist2 < r2) & cok[:, None, :] & rok[:, :, None]
        vi, ri, ci = np.nonzero(hit)
        pix = rr[vi, ri] * W + cc[vi, ci]
        zz, vv, dd = z[n, vid[vi]], vid[vi], dist2[vi, ri, ci]
        order = np.lexsort((vv, zz, pix))                                             # per pixel: nearest in z first, then lower index
        pix, zz, vv, dd = pix[order], zz[order], vv[order], dd[order]
        first = np.r_[True, pix[1:] != pix[:-1]]
        start = np.maximum.accumulate(np.where(first, np.arange(pix.size), 0))
        rank = np.arange(pix.size) - start
        keep = rank < K
        pr, pc = pix[keep] // W, pix[keep] % W
        idx[n, pr, pc, rank[keep]] = n * V + vv[keep]; zb[n, pr, pc, rank[keep]] = zz[keep]; d2o[n, pr, pc, rank[keep]] = dd[keep]
    return idx, zb, d2o


def rasterize_meshes(verts_ndc, faces, H, W):
    v = np.asarray(verts_ndc, np.float32); faces = np.asarray(faces, np.int64)
    N, F = v.shape[0], faces.shape[0]
    xs, ys = pixel_centres_ndc(H, W)
    xs_up, ys_up = xs[::-1].copy(), ys[::-1].copy()                                   # ascending copies for exact float32 range searches
    p2f = -np.ones((N, H, W, 1), np.int64); bary = -np.ones((N, H, W, 1, 3), np.float32); zbuf = -np.ones((N, H, W, 1), np.float32)
    f32 = np.float32
    fvalid = faces.min(1) >= 0
    fsafe = np.where(fvalid[:, None], faces, 0)
    for n in range(N):
        a, b, c = v[n, fsafe[:, 0]], v[n, fsafe[:, 1]], v[n, fsafe[:, 2]]
        area = _edge(a[:, 0], a[:, 1], b[:, 0], b[:, 1], c[:, 0], c[:, 1])
        ok = fvalid & ~(np.maximum(np.maximum(a[:, 2], b[:, 2]), c[:, 2]) < 0) & ~((-K_EPS <= area) & (area <= K_EPS))
        xmin = np.minimum(np.minimum(a[:, 0], b[:, 0]), c[:, 0]); xmax = np.maximum(np.maximum(a[:, 0], b[:, 0]), c[:, 0])
        ymin = np.minimum(np.minimum(a[:, 1], b[:, 1]), c[:, 1]); ymax = np.maximum(np.maximum(a[:, 1], b[:, 1]), c[:, 1])
        # columns with xmin <= xs[col] <= xmax: xs decreases with col, so the range in the ascending copy is [lo, hi) and col = W-1-i
        ilo, ihi = np.searchsorted(xs_up, xmin, 'left'), np.searchsorted(xs_up, xmax, 'right')
        jlo, jhi = np.searchsorted(ys_up, ymin, 'left'), np.searchsorted(ys_up, ymax, 'right')
        wdt, hgt = ihi - ilo, jhi - jlo
        ok &= (wdt > 0) & (hgt > 0)
        cstart, rstart = W - ihi, H - jhi                                             # first (smallest) column / row of the box
        size = np.maximum(wdt, hgt)
        cand = []
        fid_all = np.nonzero(ok)[0]
        edges = [0, 2, 4, 8, 16, 32, 64, 128, 256, max(H, W) + 1]
        for lo_b, B in zip(edges[:-1], edges[1:]):
            fid = fid_all[(size[fid_all] > lo_b) & (size[fid_all] <= B)]
            for chunk in np.array_split(fid, max(1, int(fid.size * B * B // (1 << 24)) + 1)):
                if chunk.size == 0:
                    continue
                A, Bv, Cv = a[chunk], b[chunk], c[chunk]
                cc = cstart[chunk][:, None] + np.arange(B)[None, :]; rr = rstart[chunk][:, None] + np.arange(B)[None, :]
                cok = np.arange(B)[None, :] < wdt[chunk][:, None]; rok = np.arange(B)[None, :] < hgt[chunk][:, None]
                px = xs[np.clip(cc, 0, W - 1)][:, None, :]; py = ys[np.clip(rr, 0, H - 1)][:, :, None]      # [f,1,B], [f,B,1]
                ax, ay, az = A[:, 0][:, None, None], A[:, 1][:, None, None], A[:, 2][:, None, None]
                bx, by, bz = Bv[:, 0][:, None, None], Bv[:, 1][:, None, None], Bv[:, 2][:, None, None]
                cx, cy, cz = Cv[:, 0][:, None, None], Cv[:, 1][:, None, None], Cv[:, 2][:, None, None]
                den = (_edge(cx, cy, ax, ay, bx, by) + K_EPS).astype(f32)
                w0 = (_edge(px, py, bx, by, cx, cy) / den).astype(f32)
                w1 = (_edge(px, py, cx, cy, ax, ay) / den).astype(f32)
                w2 = (_edge(px, py, ax, ay, bx, by) / den).astype(f32)
                t0, t1, t2 = (w0 * bz * cz).astype(f32), (az * w1 * cz).astype(f32), ((az * bz) * w2).astype(f32)
                dn = np.maximum((t0 + t1 + t2).astype(f32), K_EPS)
                b0, b1, b2 = (t0 / dn).astype(f32), (t1 / dn).astype(f32), (t2 / dn).astype(f32)
                pz = (b0 * az + b1 * bz + b2 * cz).astype(f32)
                inside = (b0 > 0) & (b1 > 0) & (b2 > 0) & (pz >= 0) & cok[:, None, :] & rok[:, :, None]
                fi, ri, ci = np.nonzero(inside)
                cand.append((rr[fi, ri] * W + cc[fi, ci], pz[fi, ri, ci], chunk[fi], b0[fi, ri, ci], b1[fi, ri, ci], b2[fi, ri, ci]))
        if not cand:
            continue
        pix, pz, fidx, b0, b1, b2 = [np.concatenate(t) for t in zip(*cand)]
        if pix.size == 0:                                                            # faces in the image, none covering a pixel centre
            continue
        order = np.lexsort((fidx, pz, pix))                                           # per pixel: smallest depth, ties -> lowest face index
        pix, pz, fidx, b0, b1, b2 = pix[order], pz[order], fidx[order], b0[order], b1[order], b2[order]
        first = np.r_[True, pix[1:] != pix[:-1]]
        pr, pc = pix[first] // W, pix[first] % W
        p2f[n, pr, pc, 0] = n * F + fidx[first]; zbuf[n, pr, pc, 0] = pz[first]
        bary[n, pr, pc, 0, 0] = b0[first]; bary[n, pr, pc, 0, 1] = b1[first]; bary[n, pr, pc, 0, 2] = b2[first]
    return p2f, bary, zbuf
