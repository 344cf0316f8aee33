"""TEST INFRASTRUCTURE ONLY (imported by tests/): numpy restatement of the joint augmentation resampling that
``smsut_warp_joint`` implements.  The reference composes PIL / torchvision / elasticdeform
(data_loader/externalTransforms.py:45-90).  Pinned (r02, tests/test_augment_pil_cpu.py): the rotation-only and crop-only warps of
this oracle match PIL's ``Image.rotate`` / ``Image.crop().resize()`` -- what torchvision's ``F.rotate`` / ``F.resized_crop`` call on
PIL images -- in direction, centre, pixel-centre convention and bilinear / nearest interpolation.  Still parity UNPINNED: the elastic
deformation (``elasticdeform`` is not installed here) and the composition (the device path resamples ONCE instead of three times).

    source(yo, xo) = A * (xo, yo, 1) + bilinear(ctrl)(yo, xo);  image: bilinear, zeros outside;  labels: nearest.
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
