"""Evidence run (not a test): tests/dice_proxy.py's UGANConsisTrainer task in the config-5 arithmetic (fp16 conv operands, half
storage) at 128x128, 4 + 4 slices, against the fp32 CPU oracle iteration.  Writes gpurun_out/r04_dice_ugan_f16.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import smsut_amd
from smsut_amd import ops
import dice_proxy
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
out = []
for dt, seed in [a.split(":") for a in (sys.argv[2:] or ["f16:2021"])]:          # e.g. f32:2021 f16:2022
    ops.set_conv_dtype(dt)
    res = dice_proxy.run_ugan(steps=steps, size=128, half=4, n_train=16, n_val=4, seed=int(seed), log=lambda *a: None)
    res = {k: v for k, v in res.items() if not k.startswith("loss_trace")}
    res["conv_dtype"], res["seed"] = dt, int(seed)
    out.append(res)
    print(dt, seed, {k: round(res[k], 5) for k in ("dice_mean_hip", "dice_mean_oracle", "delta_mean_dice_pt", "prediction_agreement")}, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"r04_dice_ugan_dtype_{steps}.json"), "w"), indent=1)
