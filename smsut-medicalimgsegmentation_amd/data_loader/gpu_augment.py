"""Joint geometric augmentation on the device: the reference runs JointRotate(+-15 deg),
JointElasticDeform(sigma 9-13, 3x3 control points, p = 0.5) and JointRandomResizedCrop(size, scale 0.6-1.0,
ratio 3/4-4/3) per slice in PIL worker processes (data_loader/externalTransforms.py:45-90, config.py:60-71); at
thousands of slices per second per GPU that does not keep up, so the three are composed into ONE resampling pass per
batch (``smsut_warp_joint``): output pixel -> crop window -> rotation about the image centre (+ elastic offset).

Parameter draws follow the reference's distributions (uniform angle; torchvision's RandomResizedCrop.get_params: up to
10 tries of area ~ U(scale) * HW and log-uniform aspect ratio, central fallback; elastic control offsets ~ N(0, sigma)
with sigma ~ U(sigmas), applied with probability p).  The resampling itself is NOT bit-identical to PIL + elasticdeform
(single bilinear pass instead of three; bilinear instead of B-spline control-grid interpolation): parity for this row
is against ``oracle/augment_oracle.py`` (a numpy restatement of the kernel's definition), not against the reference.
"""
import math
import random

import torch

from .. import _hip as H


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
        """Per-sample parameters (host RNG, reference draw order per sample: rotate, elastic, crop)."""
        H_, W_ = in_hw
        out_hw = (self.out, self.out) if self.out else in_hw
        aff, ctrl, any_el = [], [], False
        P = self.points
        for _ in range(n):
            ang = random.uniform(-self.cfg["rotate_degrees"], self.cfg["rotate_degrees"]) if self.cfg.get("rotate") else 0.0
            c = torch.zeros(2, P, P)
            if self.cfg.get("elasticDeform"):
                s = random.uniform(*self.cfg["elasticDeform_sigmas"])
                if random.random() < 0.5:
                    c = torch.tensor([[[random.gauss(0.0, s) for _ in range(P)] for _ in range(P)] for _ in range(2)])
                    any_el = True
            crop = resized_crop_params(H_, W_) if self.cfg.get("resizeCrop") else (0, 0, H_, W_)
            aff.append(affine_for(ang, crop, in_hw, out_hw))
            ctrl.append(c)
        return torch.tensor(aff, dtype=torch.float32), (torch.stack(ctrl) if any_el else None), out_hw

    def __call__(self, img, msk=None, params=None):
        n, _, H_, W_ = img.shape
        aff, ctrl, (Ho, Wo) = params if params is not None else self.draw(n, (H_, W_))
        return warp_joint(img, msk, aff, ctrl, Ho, Wo)


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
