"""Joint geometric augmentation on the device: the reference runs JointRotate(+-15 deg),
JointElasticDeform(sigma 9-13, 3x3 control points, p = 0.5) and JointRandomResizedCrop(size, scale 0.6-1.0,
ratio 3/4-4/3) per slice in PIL worker processes (data_loader/externalTransforms.py:45-90, config.py:60-71, order:
baseLoader.py:92-98); at thousands of slices per second per GPU that does not keep up, so they run as device passes per batch.

r04: the passes follow the reference's own structure.  When no slice of the batch drew the elastic deformation, rotation and crop
are composed into ONE bilinear / nearest resampling (``smsut_warp_joint``; pinned to PIL's rotate -> crop + resize outputs,
tests/golden/augment_pil.npz).  When one did, the batch takes the reference's three steps: rotate (bilinear image, nearest labels,
values rounded to the 8-bit grid as PIL stores them), ``elasticdeform.deform_random_grid(order=[0, 0])`` restated
(``smsut_elastic_deform``: cubic-B-spline interpolation of the P x P control displacements, ORDER-0 sampling of image AND labels,
zeros outside -- slices that did not draw it carry zero displacements, an exact copy), crop + resize.  Out-of-image samples read 0 =
black on the [0, 1] scale the passes run on (ToTensor's range; Normalize(0.5, 0.5) comes after, as in baseLoader.py:104-108).

Parameter draws follow the reference's distributions (uniform angle; torchvision's RandomResizedCrop.get_params: up to
10 tries of area ~ U(scale) * HW and log-uniform aspect ratio, central fallback; elastic control offsets ~ N(0, sigma)
with sigma ~ U(sigmas), applied with probability p).  ``elasticdeform`` is a third-party package that is not installed here (and
not pinned by the reference): its published algorithm is restated (oracle/augment_oracle.py::elastic_deform_grid builds it from
scipy.ndimage's spline routines, the code elasticdeform's C extension derives from); parity at that boundary is UNPINNED.
"""
import math
import random

import numpy as np
import torch

from .. import _hip as H


def spline_prefilter(ctrl):
    """Cubic B-spline coefficients of control values along the last two axes, mirror boundary (what
    ``scipy.ndimage.spline_filter1d(order=3, mode='mirror')`` computes along each axis, which is how elasticdeform's ``deform_grid``
    prefilters its displacement grid): after it, the spline through the coefficients INTERPOLATES the control values.  Host side
    (P x P values per slice); float64 inside.  ``ctrl`` [..., P, P] tensor -> same shape, float32."""
    c = np.asarray(ctrl, dtype=np.float64).copy()
    z = math.sqrt(3.0) - 2.0                                   # the pole of the cubic B-spline filter
    for ax in (-2, -1):
        c = np.moveaxis(c, ax, -1)
        n = c.shape[-1]
        if n > 1:
            c = c * ((1.0 - z) * (1.0 - 1.0 / z))              # gain
            # causal initialisation: sum over the mirrored signal, exact for short lines
            zn = z ** (n - 1)
            acc = c[..., 0] + zn * c[..., n - 1]
            z1, z2 = z, zn * zn / z
            for k in range(1, n - 1):
                acc = acc + (z1 + z2) * c[..., k]
                z1, z2 = z1 * z, z2 / z
            c[..., 0] = acc / (1.0 - zn * zn)
            for k in range(1, n):
                c[..., k] = c[..., k] + z * c[..., k - 1]
            c[..., n - 1] = (z / (z * z - 1.0)) * (c[..., n - 1] + z * c[..., n - 2])
            for k in range(n - 2, -1, -1):
                c[..., k] = z * (c[..., k + 1] - c[..., k])
        c = np.moveaxis(c, -1, ax)
    return torch.from_numpy(c.astype(np.float32))


def resized_crop_params(height, width, scale=(0.6, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """torchvision.transforms.RandomResizedCrop.get_params (as called by externalTransforms.py:51): (i, j, h, w)."""
    area = height * width
    log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
    for _ in range(10):
        target_area = area * random.uniform(scale[0], scale[1])
        aspect = math.exp(random.uniform(log_ratio[0], log_ratio[1]))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            return random.randint(0, height - h), random.randint(0, width - w), h, w
    in_ratio = float(width) / float(height)                    # fallback: central crop
    if in_ratio < min(ratio):
        w, h = width, int(round(width / min(ratio)))
    elif in_ratio > max(ratio):
        h, w = height, int(round(height * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def affine_for(angle_deg, crop, in_hw, out_hw):
    """2x3 matrix mapping an OUTPUT pixel (xo, yo) to SOURCE coordinates: output grid -> crop window (i, j, h, w) of the
    rotated image -> rotate by ``angle`` about the image centre back into the source image."""
    (i, j, h, w), (H_, W_), (Ho, Wo) = crop, in_hw, out_hw
    sx, sy = w / Wo, h / Ho                                      # pixel-centre aligned resize (PIL box resampling model)
    # position in the rotated image: xr = j + (xo + 0.5) * sx - 0.5
    cx, cy = (W_ - 1) / 2.0, (H_ - 1) / 2.0
    a = math.radians(angle_deg)
    ca, sa = math.cos(a), math.sin(a)
    # source = R(a) * (pr - c) + c, with pr = (xr, yr)
    bx, by = j + 0.5 * sx - 0.5 - cx, i + 0.5 * sy - 0.5 - cy
    return [ca * sx, -sa * sy, ca * bx - sa * by + cx,
            sa * sx, ca * sy, sa * bx + ca * by + cy]


class GpuJointAugment:
    def __init__(self, data_aug, out_size):
        self.cfg = dict(data_aug or {})
        self.out = int(self.cfg.get("resizeCrop_size", out_size)) if self.cfg.get("resizeCrop") else None
        self.points = int(self.cfg.get("elasticDeform_points", 3))

    def draw(self, n, in_hw):
        """Per-sample parameters (host RNG, reference draw order per sample: rotate, elastic, crop): (angles, crops, control
        displacements [n, 2, P, P] or None when no slice drew the deformation, output size)."""
        H_, W_ = in_hw
        out_hw = (self.out, self.out) if self.out else in_hw
        angs, crops, ctrl, any_el = [], [], [], False
        P = self.points
        for _ in range(n):
            ang = random.uniform(-self.cfg["rotate_degrees"], self.cfg["rotate_degrees"]) if self.cfg.get("rotate") else 0.0
            c = torch.zeros(2, P, P)
            if self.cfg.get("elasticDeform"):
                s = random.uniform(*self.cfg["elasticDeform_sigmas"])            # externalTransforms.py:80 (drawn before the coin)
                if random.random() < 0.5:
                    c = torch.tensor([[[random.gauss(0.0, s) for _ in range(P)] for _ in range(P)] for _ in range(2)])
                    any_el = True
            crops.append(resized_crop_params(H_, W_) if self.cfg.get("resizeCrop") else (0, 0, H_, W_))
            angs.append(ang)
            ctrl.append(c)
        return angs, crops, (torch.stack(ctrl) if any_el else None), out_hw

    def __call__(self, img, msk=None, params=None):
        """img [N,1,H,W] on the [0, 1] scale (zeros = black outside the image), msk [N,H,W] int64 or None.

        The resampling path is chosen PER SLICE (a slice's pixels must not depend on what its batchmates drew):
          * a slice that did not draw the elastic deformation: rotation and crop + resize composed into ONE bilinear pass -- the form
            ``tests/golden/augment_pil.npz`` pins against PIL's two 8-bit steps (``tests/test_augment_pil_cpu.py``: within 1.5 grey
            levels mean of rotate -> 8-bit -> resized_crop, externalTransforms.py:45-66);
          * a slice that did: the reference's three steps -- rotate, round to the 8-bit grid PIL hands on, elastic deformation
            (order 0), crop + resize."""
        n, _, H_, W_ = img.shape
        angs, crops, ctrl, (Ho, Wo) = params if params is not None else self.draw(n, (H_, W_))

        def composed(im, mk, idx):
            aff = torch.tensor([affine_for(angs[i], crops[i], (H_, W_), (Ho, Wo)) for i in idx], dtype=torch.float32)
            return warp_joint(im, mk, aff, None, Ho, Wo)

        def three_steps(im, mk, idx, ct):
            full = (0, 0, H_, W_)
            rot = torch.tensor([affine_for(angs[i], full, (H_, W_), (H_, W_)) for i in idx], dtype=torch.float32)
            im, mk = warp_joint(im, mk, rot, None, H_, W_)
            im = torch.round(im * 255.0) / 255.0             # PIL hands an 8-bit image on (F.rotate -> Image)
            im, mk = elastic_deform(im, mk, ct)
            crop = torch.tensor([affine_for(0.0, crops[i], (H_, W_), (Ho, Wo)) for i in idx], dtype=torch.float32)
            return warp_joint(im, mk, crop, None, Ho, Wo)

        if ctrl is None:                                     # nobody drew the deformation
            return composed(img, msk, range(n))
        drew = ctrl.flatten(1).abs().sum(1) > 0              # (a drawn grid of exact zeros has probability 0 and IS the identity)
        ie = [i for i in range(n) if bool(drew[i])]
        ip = [i for i in range(n) if not bool(drew[i])]
        if not ip:
            return three_steps(img, msk, ie, ctrl)
        oimg = torch.empty(n, 1, Ho, Wo, dtype=torch.float32, device=img.device)
        omsk = torch.empty(n, Ho, Wo, dtype=torch.int64, device=img.device) if msk is not None else None
        for idx, fn in ((ip, lambda im, mk: composed(im, mk, ip)), (ie, lambda im, mk: three_steps(im, mk, ie, ctrl[ie]))):
            if not idx:
                continue
            sel = torch.tensor(idx, device=img.device)
            a, b = fn(img.index_select(0, sel), msk.index_select(0, sel) if msk is not None else None)
            oimg.index_copy_(0, sel, a)
            if omsk is not None:
                omsk.index_copy_(0, sel, b)
        return oimg, omsk


def elastic_deform(img, msk, ctrl):
    """``elasticdeform.deform_random_grid([img, msk], order=[0, 0])`` given its control displacements ``ctrl`` [N,2,P,P] (dy, dx in
    pixels): one launch of ``smsut_elastic_deform``."""
    n, c, H_, W_ = img.shape
    assert c == 1, "slices are single-channel (config.img_channels)"
    dev = img.device
    img = img.contiguous()
    coef = spline_prefilter(ctrl).to(dev).contiguous()
    oimg = torch.empty_like(img)
    omsk = None
    if msk is not None:
        msk = msk.to(torch.int64).contiguous()
        omsk = torch.empty_like(msk)
    H.call("smsut_elastic_deform", img, msk, coef, oimg, omsk, n, H_, W_, int(ctrl.shape[-1]), H.stream_ptr())
    return oimg, omsk


def warp_joint(img, msk, aff, ctrl, Ho, Wo):
    """One launch of ``smsut_warp_joint``: img [N,1,H,W] fp32 (device), msk [N,H,W] int64 or None, aff [N,6], ctrl
    [N,2,P,P] or None."""
    n, c, H_, W_ = img.shape
    assert c == 1, "slices are single-channel (config.img_channels)"
    dev = img.device
    img = img.contiguous()
    aff = aff.to(dev, torch.float32).contiguous()
    P = 0 if ctrl is None else int(ctrl.shape[-1])
    if ctrl is not None:
        ctrl = ctrl.to(dev, torch.float32).contiguous()
    oimg = torch.empty(n, 1, Ho, Wo, dtype=torch.float32, device=dev)
    omsk = None
    if msk is not None:
        msk = msk.to(torch.int64).contiguous()
        omsk = torch.empty(n, Ho, Wo, dtype=torch.int64, device=dev)
    H.call("smsut_warp_joint", img, msk, aff, ctrl, oimg, omsk, n, H_, W_, Ho, Wo, P, H.stream_ptr())
    return oimg, omsk
