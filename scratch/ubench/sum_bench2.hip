// Second microbenchmark of the split-slab reduction: block size / loads-in-flight / two-stage variants.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sum_bench2 scratch/ubench/sum_bench2.hip && /tmp/sum_bench2
#include <hip/hip_runtime.h>
#include <cstdio>

// COLS float4 columns x (NT/COLS) split lanes, U loads in flight per thread; splits range [s0, s1) per blockIdx.y
template <int COLS, int NT, int U>
__global__ void __launch_bounds__(NT) sum_v(const float* __restrict__ part, float* __restrict__ out, int wsize, int splits,
                                            int per_group) {
  constexpr int LANES = NT / COLS;
  __shared__ float4 sm[NT];
  const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const int e = (blockIdx.x * COLS + col) * 4;
  const int s0 = blockIdx.y * per_group, s1 = min(splits, s0 + per_group);
  float4 acc[U];
#pragma unroll
  for (int k = 0; k < U; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e + 3 < wsize) {
    int c = s0 + sl;
    for (; c + (U - 1) * LANES < s1; c += U * LANES) {
#pragma unroll
      for (int k = 0; k < U; ++k) {
        const float4 v = *(const float4*)(part + (size_t)(c + k * LANES) * wsize + e);
        acc[k].x += v.x; acc[k].y += v.y; acc[k].z += v.z; acc[k].w += v.w;
      }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      if (c + k * LANES < s1) {
        const float4 v = *(const float4*)(part + (size_t)(c + k * LANES) * wsize + e);
        acc[k].x += v.x; acc[k].y += v.y; acc[k].z += v.z; acc[k].w += v.w;
      }
    }
  }
  float4 s = acc[0];
#pragma unroll
  for (int k = 1; k < U; ++k) { s.x += acc[k].x; s.y += acc[k].y; s.z += acc[k].z; s.w += acc[k].w; }
  sm[threadIdx.x] = s;
  __syncthreads();
  // tree over lanes (fixed order)
  for (int h = LANES / 2; h >= 1; h >>= 1) {
    if (sl < h) {
      float4 a = sm[sl * COLS + col]; const float4 b = sm[(sl + h) * COLS + col];
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; sm[sl * COLS + col] = a;
    }
    __syncthreads();
  }
  if (sl == 0 && e + 3 < wsize) *(float4*)(out + (size_t)blockIdx.y * wsize + e) = sm[col];
}

int main() {
  struct Case { int wsize, splits; } cases[] = {{2304, 745}, {4608, 512}, {9216, 512}, {18432, 256}, {36864, 128},
                                                {73728, 64}, {147456, 32}, {589824, 8}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float* junk; hipMalloc(&junk, 512u << 20);
  for (auto cs : cases) {
    float *part, *out, *mid;
    size_t n = (size_t)cs.wsize * cs.splits;
    hipMalloc(&part, n * 4); hipMalloc(&out, cs.wsize * 4); hipMalloc(&mid, (size_t)cs.wsize * 4 * 64);
    hipMemset(part, 0, n * 4);
    auto run = [&](const char* name, auto launch) {
      float tot = 0.f; const int reps = 20;
      for (int r = 0; r < reps + 2; ++r) {
        hipMemsetAsync(junk, r, 512u << 20, 0);
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) tot += ms;
      }
      printf("wsize %7d splits %4d (%5.1f MB) %-22s %6.1f us\n", cs.wsize, cs.splits, n * 4 / 1e6, name, tot / reps * 1e3);
    };
    const int w = cs.wsize, s = cs.splits;
#define ONE(C, NT, U) run("v<" #C "," #NT "," #U ">", [&] { sum_v<C, NT, U><<<dim3((w / 4 + C - 1) / C, 1), NT>>>(part, out, w, s, s); });
    ONE(64, 256, 4) ONE(64, 256, 8) ONE(64, 256, 16) ONE(32, 256, 4) ONE(32, 256, 8) ONE(16, 256, 4) ONE(16, 256, 8) ONE(16, 256, 16)
    ONE(64, 1024, 4) ONE(64, 1024, 8) ONE(32, 1024, 4) ONE(32, 1024, 8) ONE(16, 1024, 4) ONE(16, 1024, 8) ONE(16, 1024, 16) ONE(8, 1024, 8)
#define TWO(C, NT, U, G) run("2st<" #C "," #NT "," #U ">x" #G, [&] { const int pg = (s + G - 1) / G; \
      sum_v<C, NT, U><<<dim3((w / 4 + C - 1) / C, G), NT>>>(part, mid, w, s, pg); \
      sum_v<C, 256, 4><<<dim3((w / 4 + C - 1) / C, 1), 256>>>(mid, out, w, G, G); });
    if (s >= 64) { TWO(16, 256, 8, 8) TWO(16, 256, 8, 16) TWO(32, 256, 8, 8) TWO(64, 256, 8, 4) TWO(16, 256, 4, 32) }
    hipFree(part); hipFree(out); hipFree(mid);
  }
  return 0;
}
