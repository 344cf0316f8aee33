"""In-kernel clock of the register-row weight gradient (diagnostic build with -DSMSUT_STAMPS, scratch/build_variant_rr.sh):
per workgroup, s_memtime / s_memrealtime at the start of the kernel and at the end of wave 0's main loop (MICROARCH: clock =
d(s_memtime) / d(s_memrealtime) x 100 MHz).  Usage: python scratch/rr_clock.py scratch/bin/libsmsut_stamps.so [B]"""
import ctypes, sys, torch
lib = ctypes.CDLL(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
lib.smsut_conv2d_wgrad_mfma_ws.restype = ctypes.c_int64
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for zero in (0, 1):
    for h, ci, co in [(256, 16, 16), (128, 32, 32), (64, 64, 64), (16, 256, 256)]:
        x = torch.randn(B, h, h, ci, device="cuda"); gy = torch.randn(B, h, h, co, device="cuda")
        if zero: x.zero_(); gy.zero_()
        gw = torch.empty(9 * ci * co, device="cuda")
        ws = torch.empty(lib.smsut_conv2d_wgrad_mfma_ws(B, h, h, ci, co, 3), device="cuda")
        dbg = torch.zeros(4096 * 4, dtype=torch.int64, device="cuda")
        fn = lambda: lib.smsut_conv2d_wgrad_mfma(P(x), P(gy), P(gw), P(ws), B, h, h, ci, co, 3, st)
        for _ in range(200): fn()                    # warm: the clock settles under sustained load
        lib.smsut_dbg_rr_stamps(P(dbg)); fn(); torch.cuda.synchronize(); lib.smsut_dbg_rr_stamps(ctypes.c_void_p(0))
        d = dbg.view(-1, 4).cpu()
        d = d[d[:, 0] != 0]
        cyc = (d[:, 2] - d[:, 0]).double(); rt = (d[:, 3] - d[:, 1]).double()
        ghz = (cyc / rt * 0.1).median().item()
        span_us = (d[:, 3].max() - d[:, 1].min()).item() / 100.0
        nm = B * h * (h // 16) * 36 * (ci // 16) * (co // 16) / 1024          # MFMAs per SIMD
        print(f"{'zeros ' if zero else 'random'} H{h} {ci}->{co}: {len(d)} workgroups, loop {cyc.median().item():.0f} cycles (max {cyc.max().item():.0f}) at "
              f"{ghz:.2f} GHz, first start -> last loop end {span_us:.1f} us; {nm:.0f} MFMAs per SIMD = {nm * 32:.0f} cycles", flush=True)
