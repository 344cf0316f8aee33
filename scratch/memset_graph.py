"""Does smsut_scatter_rows zero the un-sampled rows when it runs as part of a replayed hipGraph?
    python scratch/memset_graph.py <path to libsmsut .so>"""
import ctypes, sys, torch
lib = ctypes.CDLL(sys.argv[1])
f = lib.smsut_scatter_rows
f.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
B, HW, C, P = 16, 256, 256, 64
dev = torch.device("cuda")
gout = torch.randn(B * P, C, device=dev)
ids = torch.randperm(HW, device=dev)[:P].contiguous()
gfeat = torch.full((B, HW, C), 1e30, device=dev)
junk = torch.empty(64 << 20, device=dev)
def run():
    st = torch.cuda.current_stream().cuda_stream
    # neighbours in the stream, like the real backward: a producer before, a consumer after
    junk.normal_()
    rc = f(gout.data_ptr(), ids.data_ptr(), gfeat.data_ptr(), B, HW, C, P, st)
    assert rc == 0
    return gfeat.abs().sum(dim=(0, 2))
run(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = run()
bad_total = 0
for rep in range(20):
    gfeat.fill_(1e30)
    g.replay()
    torch.cuda.synchronize()
    mask = torch.ones(HW, dtype=torch.bool, device=dev); mask[ids] = False
    bad = int((out[mask] != 0).sum())
    bad_total += bad
print(sys.argv[1].split("/")[-1], "un-sampled rows left non-zero over 20 replays:", bad_total)
