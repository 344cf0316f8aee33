"""Measured values behind the bars of tests/test_f16_gpu.py::test_f16_unet_512_vs_fp32_oracle (logits, loss, gradient errors, cosine),
with half storage on / off."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smsut_amd
from smsut_amd import ops
from conftest import rel_err, l2_rel
from oracle import recipe, smsut_oracle as O
from smsut_amd.network.unet import UNet
from smsut_amd.misc.loss import DiceAndCrossEntropyLoss
torch.set_num_threads(16)
H = 512
x = recipe.synth_images((2, 1, H, H), 501); y = recipe.synth_labels(2, H, H, 5, 502)
sd = recipe.fill(recipe.unet_shapes(1, 5, 16), 500)
leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
ref = O.unet_forward(leaf, x); ref_loss = O.dice_ce(ref, y); ref_loss.backward()
ops.set_conv_dtype("f16")
for store in (True, False):
    ops.F16_STORE = store
    net = UNet(1, 5, 16, norm_type="instance", act_type="lrelu"); net.load_state_dict(sd); net.cuda().train()
    out = net(x.cuda()); loss = DiceAndCrossEntropyLoss(0.5, 0.5, batch_dice=True)(out, y.cuda()); loss.backward()
    e = rel_err(out.detach().cpu().numpy(), ref.detach().numpy())
    errs = {k: l2_rel(p.grad.cpu().numpy(), leaf[k].grad.numpy()) for k, p in net.named_parameters()}
    ga = torch.cat([p.grad.detach().cpu().double().reshape(-1) for _, p in net.named_parameters()])
    gb = torch.cat([leaf[k].grad.double().reshape(-1) for k, _ in net.named_parameters()])
    cos = float(torch.dot(ga, gb) / (ga.norm() * gb.norm()))
    worst = max(errs.items(), key=lambda kv: kv[1])
    print(f"half storage {store}: logits {e:.3e}  loss rel {abs(loss.item() - ref_loss.item()) / abs(ref_loss.item()):.2e}  grad median {np.median(list(errs.values())):.4f} "
          f"worst {worst[1]:.4f} ({worst[0]})  cosine {cos:.5f}", flush=True)
