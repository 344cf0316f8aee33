"""How far apart do two runs of the SAME uganConsis proxy training end when only rounding differs?  (tests/dice_proxy.py::run_ugan)
HIP vs oracle, and oracle vs oracle with its initial weights perturbed by 1e-6 relative -- the reference arithmetic's own chaos band."""
import json, os, sys, time, types
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dice_proxy as DP
from oracle import recipe
out = {}
orig_fill = recipe.fill
for seed in (2021, 2022):
    for steps in (300, 600):
        r = DP.run_ugan(steps=steps, seed=seed, log=lambda *a: None)
        out[f"hip_vs_oracle seed {seed} steps {steps}"] = (r["dice_mean_hip"], r["dice_mean_oracle"], r["delta_mean_dice_pt"])
        print(f"seed {seed} steps {steps}: HIP {r['dice_mean_hip']:.4f} oracle {r['dice_mean_oracle']:.4f} delta {r['delta_mean_dice_pt']:+.2f} pt", flush=True)
    # oracle vs perturbed oracle (the HIP leg is ignored): patch recipe.fill to perturb the generator weights
    def pert(shapes, s, _o=orig_fill):
        sd = _o(shapes, s)
        g = torch.Generator().manual_seed(99)
        return {k: v * (1 + 1e-6 * torch.randn(v.shape, generator=g)) for k, v in sd.items()}
    recipe.fill = pert
    r2 = DP.run_ugan(steps=300, seed=seed, log=lambda *a: None)
    recipe.fill = orig_fill
    r1 = DP.run_ugan(steps=300, seed=seed, log=lambda *a: None)
    print(f"seed {seed} steps 300: oracle {r1['dice_mean_oracle']:.4f} vs oracle(weights x (1 + 1e-6 N)) {r2['dice_mean_oracle']:.4f}: "
          f"{100 * (r2['dice_mean_oracle'] - r1['dice_mean_oracle']):+.2f} pt;  HIP {r1['dice_mean_hip']:.4f} vs HIP(perturbed) {r2['dice_mean_hip']:.4f}", flush=True)
