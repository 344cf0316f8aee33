"""tests/dice_proxy.run_ugan for a list of seeds under the current environment (SMSUT_WGRAD_PAIR / SMSUT_FIN are read at start-up):
HIP-vs-oracle Dice deltas, to tell a chaotic endpoint from an offset.  python scratch/dice_pair_ab.py 2021 2022 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import dice_proxy
tag = f"PAIR={os.environ.get('SMSUT_WGRAD_PAIR', '1')} FIN={os.environ.get('SMSUT_FIN', '1')}"
for s in [int(a) for a in sys.argv[1:]]:
    r = dice_proxy.run_ugan(steps=300, size=64, seed=s, log=lambda *a: None)
    print(f"{tag} seed {s}: HIP {r['dice_mean_hip']:.4f} oracle {r['dice_mean_oracle']:.4f} delta {r['delta_mean_dice_pt']:+.2f} pt "
          f"agreement {r['prediction_agreement']:.4f} G_seg end {r['g_seg_trace_hip'][-1][1]:.4f} / {r['g_seg_trace_oracle'][-1][1]:.4f}", flush=True)
