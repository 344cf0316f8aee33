"""Persistent conv: time vs the number of workgroups per CU the grid is sized for (SMSUT_P_WGS_PER_CU, one process per value)."""
import os, sys, subprocess, json
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, '.')
    import torch, smsut_amd
    from smsut_amd import ops, _hip as H
    out = {}
    for (B, h, K, N, form) in [(16, 256, 32, 16, "cat"), (32, 256, 16, 16, "stats"), (32, 128, 32, 32, "stats"), (32, 64, 64, 64, "stats")]:
        x = torch.randn(B, h, h, K, device="cuda"); w = torch.randn(9, K, N, device="cuda") * 0.05
        y = torch.empty(B, h, h, N, device="cuda")
        tiles = H.call("smsut_conv2d_mfma_tiles", B, h, h, K, N, 3, 0)
        part = torch.empty(B * tiles * N * 2, device="cuda")
        xa, xb = x[..., :K // 2].contiguous(), x[..., K // 2:].contiguous()
        st = torch.cuda.current_stream().cuda_stream
        if form == "cat":
            f = lambda: H.call("smsut_conv2d_fwd_mfma_stats_cat", xa, xb, w, y, part, B, h, h, K, N, st)
        else:
            f = lambda: H.call("smsut_conv2d_fwd_mfma_stats", x, w, y, part, B, h, h, K, N, 3, st)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); [f() for _ in range(20)]; e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        out[f"B{B} H{h} {K}->{N} {form}"] = (round(us, 1), round(2.0 * B * h * h * K * N * 9 / us / 1e6, 1))
    print(json.dumps(out))
else:
    for occ in (os.environ.get("OCCS", "0,1,2").split(",")):
        env = dict(os.environ, SMSUT_P_WGS_PER_CU=occ)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        print("wgs/cu", occ, line[-1] if line else r.stderr[-300:], flush=True)
