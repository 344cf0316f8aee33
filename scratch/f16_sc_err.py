"""Per-gradient error of the fp16-operand BasicBlock vs the fp32 one, with the fused fp16 shortcut on / off (env SMSUT_FUSE_SHORTCUT_F16)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_f16_gpu as T
import importlib
pkg = importlib.import_module("smsut_amd")
from smsut_amd import ops
for (n, h, ci, co) in [(5, 128, 16, 32), (4, 128, 64, 32), (9, 64, 32, 64), (16, 32, 32, 64)]:
    x = T.rnd(n, ci, h, h, seed=1).cuda().contiguous(memory_format=torch.channels_last)
    ws_ = [T.hwio(ops, T.rnd(co, ci, 3, 3, seed=2) / np.sqrt(9 * ci)), (1 + 0.1 * T.rnd(co, seed=3)).cuda(), (0.1 * T.rnd(co, seed=4)).cuda(),
           T.hwio(ops, T.rnd(co, co, 3, 3, seed=5) / np.sqrt(9 * co)), (1 + 0.1 * T.rnd(co, seed=6)).cuda(), (0.1 * T.rnd(co, seed=7)).cuda(),
           T.hwio(ops, T.rnd(co, ci, 1, 1, seed=8) / np.sqrt(ci)), (1 + 0.1 * T.rnd(co, seed=9)).cuda(), (0.1 * T.rnd(co, seed=10)).cuda()]
    gout = (T.rnd(n, co, h, h, seed=11) * 3e-7).cuda().contiguous(memory_format=torch.channels_last)
    o32, g32 = T._block(ops, x, ws_, gout, "f32")
    o16, g16 = T._block(ops, x, ws_, gout, "f16")
    print((n, h, ci, co), "fwd", T.rel_err(o16.cpu().numpy(), o32.cpu().numpy()),
          [round(float(T.l2_rel(a.cpu().numpy(), b.cpu().numpy())), 4) for a, b in zip(g16, g32)])
