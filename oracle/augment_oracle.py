"""TEST INFRASTRUCTURE ONLY (imported by tests/): numpy / scipy restatements of the joint augmentation passes
(``smsut_warp_joint``, ``smsut_elastic_deform``).  The reference composes PIL / torchvision / elasticdeform
(data_loader/externalTransforms.py:45-90, order baseLoader.py:92-98).

Pinned: the rotation-only and crop-only warps of ``warp_joint`` match PIL's ``Image.rotate`` / ``Image.crop().resize()`` -- what
torchvision's ``F.rotate`` / ``F.resized_crop`` call on PIL images -- in direction, centre, pixel-centre convention and bilinear /
nearest interpolation (tests/test_augment_pil_cpu.py), and the device kernel itself matches committed PIL outputs including the
rotate -> crop order (tests/golden/augment_pil.npz, tests/test_data_loader_gpu.py).

``elastic_deform_grid`` restates ``elasticdeform.deform_grid`` (what ``deform_random_grid`` calls after drawing
``numpy.random.randn(2, points, points) * sigma``; externalTransforms.py:83: ``order=[0, 0]``, defaults mode='constant', cval=0,
prefilter=True) from its published algorithm: the displacement grid is spline-filtered along each axis
(``scipy.ndimage.spline_filter1d``, order 3, mirror) and evaluated per pixel at ``pixel * (points - 1) / (size - 1)`` by cubic
B-spline interpolation with mirror boundary; the inputs are sampled at ``pixel + displacement`` with the spline order given (0:
``floor(c + 0.5)``), a coordinate below 0 or above ``size - 1`` reads ``cval``.  The package is a THIRD-PARTY DEPENDENCY that is
absent here (not vendored, not pinned by the reference: no requirements file; current release 0.5.x): parity at that boundary is
UNPINNED -- the spline arithmetic itself is scipy's (``map_coordinates``), which elasticdeform's C extension is derived from.

    warp_joint:  source(yo, xo) = A * (xo, yo, 1) + bilinear(ctrl)(yo, xo);  image: bilinear, zeros outside;  labels: nearest.
"""
import numpy as np


def warp_joint(img, msk, aff, ctrl, Ho, Wo):
    img = np.asarray(img, dtype=np.float32)
    N, H, W = img.shape
    oimg = np.zeros((N, Ho, Wo), np.float32)
    omsk = None if msk is None else np.zeros((N, Ho, Wo), np.int64)
    yo, xo = np.meshgrid(np.arange(Ho, dtype=np.float32), np.arange(Wo, dtype=np.float32), indexing="ij")
    for n in range(N):
        a = np.asarray(aff[n], dtype=np.float32)
        xs = a[0] * xo + a[1] * yo + a[2]
        ys = a[3] * xo + a[4] * yo + a[5]
        if ctrl is not None:
            c = np.asarray(ctrl[n], dtype=np.float32)
            P = c.shape[-1]
            gy = yo * np.float32(P - 1) / np.float32(Ho - 1) if Ho > 1 else np.zeros_like(yo)
            gx = xo * np.float32(P - 1) / np.float32(Wo - 1) if Wo > 1 else np.zeros_like(xo)
            y0 = np.clip(np.floor(gy).astype(np.int64), 0, P - 2); x0 = np.clip(np.floor(gx).astype(np.int64), 0, P - 2)
            fy, fx = gy - y0, gx - x0
            y1, x1 = np.minimum(y0 + 1, P - 1), np.minimum(x0 + 1, P - 1)
            d = [(1 - fy) * ((1 - fx) * c[k][y0, x0] + fx * c[k][y0, x1]) + fy * ((1 - fx) * c[k][y1, x0] + fx * c[k][y1, x1])
                 for k in range(2)]
            ys = ys + d[0].astype(np.float32); xs = xs + d[1].astype(np.float32)
        fy0, fx0 = np.floor(ys), np.floor(xs)
        iy, ix = fy0.astype(np.int64), fx0.astype(np.int64)
        wy, wx = ys - fy0, xs - fx0

        def at(y, x):
            ok = (y >= 0) & (y < H) & (x >= 0) & (x < W)
            return np.where(ok, img[n][np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)], np.float32(0))
        oimg[n] = (1 - wy) * ((1 - wx) * at(iy, ix) + wx * at(iy, ix + 1)) + wy * ((1 - wx) * at(iy + 1, ix) + wx * at(iy + 1, ix + 1))
        if msk is not None:
            ny, nx = np.floor(ys + np.float32(0.5)).astype(np.int64), np.floor(xs + np.float32(0.5)).astype(np.int64)
            ok = (ny >= 0) & (ny < H) & (nx >= 0) & (nx < W)
            omsk[n] = np.where(ok, np.asarray(msk[n])[np.clip(ny, 0, H - 1), np.clip(nx, 0, W - 1)], 0)
    return oimg, omsk


def elastic_displacement(disp, H, W):
    """Per-pixel displacement field [2, H, W] (dy, dx) of a [2, P, P] control grid: scipy's cubic spline through the control values,
    mirror boundary, first / last control point on the first / last pixel."""
    from scipy import ndimage
    disp = np.asarray(disp, dtype=np.float64)
    P = disp.shape[-1]
    coef = disp.copy()
    for ax in (1, 2):
        coef = ndimage.spline_filter1d(coef, order=3, axis=ax, mode="mirror")
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    gy = yy * (P - 1) / (H - 1) if H > 1 else np.zeros_like(yy)
    gx = xx * (P - 1) / (W - 1) if W > 1 else np.zeros_like(xx)
    return np.stack([ndimage.map_coordinates(coef[k], [gy, gx], order=3, mode="mirror", prefilter=False) for k in range(2)])


def elastic_deform_grid(img, msk, disp):
    """``elasticdeform.deform_grid([img, msk], disp, order=[0, 0])`` for batches: img [N,H,W] float, msk [N,H,W] int or None, disp
    [N,2,P,P].  Returns (image, labels, per-pixel source coordinates [N,2,H,W] -- for tests that mask coordinates within rounding of a
    pixel boundary)."""
    img = np.asarray(img)
    N, H, W = img.shape
    oimg = np.zeros_like(img)
    omsk = None if msk is None else np.zeros_like(np.asarray(msk))
    coords = np.zeros((N, 2, H, W))
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    for n in range(N):
        d = elastic_displacement(disp[n], H, W)
        ys, xs = yy + d[0], xx + d[1]
        coords[n, 0], coords[n, 1] = ys, xs
        inside = (ys >= 0) & (ys <= H - 1) & (xs >= 0) & (xs <= W - 1)
        ny = np.clip(np.floor(ys + 0.5).astype(np.int64), 0, H - 1)
        nx = np.clip(np.floor(xs + 0.5).astype(np.int64), 0, W - 1)
        oimg[n] = np.where(inside, img[n][ny, nx], 0)
        if msk is not None:
            omsk[n] = np.where(inside, np.asarray(msk[n])[ny, nx], 0)
    return oimg, omsk, coords
