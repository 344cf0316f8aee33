"""Do two streams actually run concurrently on this box?  Small kernels (64 workgroups each), eager launches and captured graphs."""
import torch, time
dev = 'cuda'
a = torch.randn(64 * 256, 64, device=dev); b = torch.randn(64, 64, device=dev)
a2 = a.clone(); b2 = b.clone()
def work(x, w, n=200):
    for _ in range(n):
        x = torch.tanh(x)            # small elementwise kernel: 1M elements
    return x
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def serial():
    with torch.cuda.stream(s1): work(a, b); work(a2, b2)
def conc():
    with torch.cuda.stream(s1): work(a, b)
    with torch.cuda.stream(s2): work(a2, b2)
print("eager  one stream (2x200 kernels): %.3f ms   two streams: %.3f ms" % (timed(serial), timed(conc)))
# big tensors: each kernel fills the GPU
A = torch.randn(64 << 20, device=dev); A2 = A.clone()
def serial_big():
    with torch.cuda.stream(s1): work(A, b, 20); work(A2, b, 20)
def conc_big():
    with torch.cuda.stream(s1): work(A, b, 20)
    with torch.cuda.stream(s2): work(A2, b, 20)
print("eager big one stream: %.3f ms   two streams: %.3f ms" % (timed(serial_big), timed(conc_big)))
# graphs
g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
with torch.cuda.graph(g1): o1 = work(a, b)
with torch.cuda.graph(g2): o2 = work(a2, b2)
def gser():
    with torch.cuda.stream(s1): g1.replay(); g2.replay()
def gcon():
    with torch.cuda.stream(s1): g1.replay()
    with torch.cuda.stream(s2): g2.replay()
print("graphs one stream: %.3f ms   two streams: %.3f ms" % (timed(gser), timed(gcon)))
# mixed: a graph of big kernels on s1, a graph of small kernels on s2
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3): o3 = work(A, b, 20)
def mser():
    with torch.cuda.stream(s1): g3.replay(); g1.replay()
def mcon():
    with torch.cuda.stream(s1): g3.replay()
    with torch.cuda.stream(s2): g1.replay()
print("big graph + small graph, one stream: %.3f ms   two streams: %.3f ms" % (timed(mser), timed(mcon)))
