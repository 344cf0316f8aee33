"""Cost of the input-side InstanceNorm in the conv2 / wgrad2 kernels vs the plain kernels + the apply pass they replace."""
import sys; sys.path.insert(0, '.')
import torch, smsut_amd
from smsut_amd import _hip as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
st = H.stream_ptr()
for (h, c) in [(256, 16), (128, 32), (64, 64)]:
    n = B
    E = lambda *s: torch.randn(*s, device='cuda')
    y1, a1, y2, gy = E(n, h, h, c), E(n, h, h, c), E(n, h, h, c), E(n, h, h, c)
    w = E(9 * c * c) * 0.05; m, r, g, b = E(n, c), E(n, c).abs() + 0.5, E(c), E(c)
    tiles = H.call("smsut_conv2d_mfma_tiles", n, h, h, c, c, 3, 0)
    p = torch.zeros(n * tiles * c * 2, device='cuda'); gw = E(9 * c * c)
    ws = torch.empty(H.call("smsut_conv2d_wgrad_mfma_ws", n, h, h, c, c, 3), device='cuda')
    t = {
        "fwd": timeit(lambda: H.call("smsut_conv2d_fwd_mfma_stats", a1, w, y2, p, n, h, h, c, c, 3, st)),
        "fwd_inaff": timeit(lambda: H.call("smsut_conv2d_fwd_mfma_stats_inaff", y1, w, y2, p, m, r, g, b, 0.01, n, h, h, c, c, st)),
        "wgrad": timeit(lambda: H.call("smsut_conv2d_wgrad_mfma", a1, gy, gw, ws, n, h, h, c, c, 3, st)),
        "wgrad_inaff": timeit(lambda: H.call("smsut_conv2d_wgrad_mfma_inaff", y1, gy, gw, ws, m, r, g, b, 0.01, n, h, h, c, c, st)),
        "apply": timeit(lambda: H.call("smsut_instnorm_fwd_partials", y1, g, b, a1, m, r, p, tiles, n, h * h, c, 1e-5, 0.01, 1, st)),
        "finalize": timeit(lambda: H.call("smsut_in_finalize_fwd", p, tiles, m, r, n, h * h, c, 1e-5, st)),
    }
    print(f"B{n} {h}x{h} C{c}: " + "  ".join(f"{k} {v:.1f}" for k, v in t.items()), flush=True)
