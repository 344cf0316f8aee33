// Is the split-slab reduction slow IN SITU because the producer's dirty L2 lines must be written back / probed?
// Times the PAIR (producer; sum) and the producer alone between HIP events, for plain / nontemporal / agent-scope stores.
//   hipcc --offload-arch=gfx950 -O3 -o scratch/bin/sum_bench3 scratch/ubench/sum_bench3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int TPB = 256;

template <int COLS>
__global__ void __launch_bounds__(TPB) sum_cur(const float* __restrict__ part, float* __restrict__ out, int wsize, int splits) {
  constexpr int LANES = TPB / COLS;
  __shared__ float4 sm[TPB];
  const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const int e = (blockIdx.x * COLS + col) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e + 3 < wsize) {
    float4 t0 = s, t1 = s, t2 = s, t3 = s;
    int c = sl;
    for (; c + 3 * LANES < splits; c += 4 * LANES) {
      const float4 v0 = *(const float4*)(part + (size_t)c * wsize + e);
      const float4 v1 = *(const float4*)(part + (size_t)(c + LANES) * wsize + e);
      const float4 v2 = *(const float4*)(part + (size_t)(c + 2 * LANES) * wsize + e);
      const float4 v3 = *(const float4*)(part + (size_t)(c + 3 * LANES) * wsize + e);
      t0.x += v0.x; t0.y += v0.y; t0.z += v0.z; t0.w += v0.w; t1.x += v1.x; t1.y += v1.y; t1.z += v1.z; t1.w += v1.w;
      t2.x += v2.x; t2.y += v2.y; t2.z += v2.z; t2.w += v2.w; t3.x += v3.x; t3.y += v3.y; t3.z += v3.z; t3.w += v3.w;
    }
    for (; c < splits; c += LANES) {
      const float4 v = *(const float4*)(part + (size_t)c * wsize + e);
      t0.x += v.x; t0.y += v.y; t0.z += v.z; t0.w += v.w;
    }
    s.x = (t0.x + t1.x) + (t2.x + t3.x); s.y = (t0.y + t1.y) + (t2.y + t3.y);
    s.z = (t0.z + t1.z) + (t2.z + t3.z); s.w = (t0.w + t1.w) + (t2.w + t3.w);
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (sl == 0 && e < wsize) {
    float4 t = sm[col];
    for (int l = 1; l < LANES; ++l) { const float4 v = sm[l * COLS + col]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    *(float4*)(out + e) = t;
  }
}

// producer: one workgroup per split writes its slab with scalar stores, 64-float rows (like the wgrad epilogue), after
// spinning for `spin` iterations of dependent FMAs (so that the stores of different workgroups are spread in time)
template <int MODE>
__global__ void __launch_bounds__(TPB) producer(float* __restrict__ part, int wsize, int spin) {
  float a = threadIdx.x * 1e-3f;
  for (int i = 0; i < spin + (int)(blockIdx.x & 7) * (spin >> 3); ++i) a = a * 1.0001f + 1e-7f;
  float* p = part + (size_t)blockIdx.x * wsize;
  for (int e = threadIdx.x; e < wsize; e += TPB) {
    if (MODE == 0) p[e] = a;
    else if (MODE == 1) __builtin_nontemporal_store(a, p + e);
    else if (MODE == 2) __hip_atomic_store(p + e, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(p + e, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

int main() {
  struct Case { int wsize, splits; } cases[] = {{36864, 128}, {9216, 512}, {147456, 32}, {589824, 8}, {2304, 745}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto cs : cases) {
    float *part, *out;
    size_t n = (size_t)cs.wsize * cs.splits;
    hipMalloc(&part, n * 4); hipMalloc(&out, cs.wsize * 4);
    hipMemset(part, 0, n * 4);
    auto time = [&](auto launch) {
      float tot = 0.f; const int reps = 20;
      for (int r = 0; r < 3; ++r) launch();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) launch();
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&tot, e0, e1);
      return tot / reps * 1e3f;
    };
    const int w = cs.wsize, s = cs.splits, spin = 20000;
    auto sum = [&] {
      if (w >= 32768) sum_cur<64><<<(w + 255) / 256, TPB>>>(part, out, w, s);
      else if (w >= 8192) sum_cur<32><<<(w + 127) / 128, TPB>>>(part, out, w, s);
      else sum_cur<16><<<(w + 63) / 64, TPB>>>(part, out, w, s);
    };
    const float t_sum = time(sum);
#define MODE(M, name) { const float tp = time([&] { producer<M><<<s, TPB>>>(part, w, spin); }); \
      const float tb = time([&] { producer<M><<<s, TPB>>>(part, w, spin); sum(); }); \
      printf("wsize %7d splits %4d (%4.1f MB) %-8s producer %6.1f us  pair %6.1f us  -> sum in situ %5.1f us (back-to-back sums alone %.1f us)\n", \
             w, s, n * 4 / 1e6, name, tp, tb, tb - tp, t_sum); }
    MODE(0, "plain") MODE(1, "nt") MODE(2, "agent") MODE(3, "system")
    hipFree(part); hipFree(out);
  }
  return 0;
}
