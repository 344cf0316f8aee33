// Microbenchmark for the split-slab reduction out[e] = sum_s part[s][e] (wgrad second stage).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int TPB = 256;

// current design: COLS float4 columns x (TPB/COLS) split lanes per block
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

// alternative: one float4 column per WAVE-lane group: block = 64 columns (1 KB rows) x 4 split lanes, 8 loads in flight
__global__ void __launch_bounds__(TPB) sum_wide(const float* __restrict__ part, float* __restrict__ out, int wsize, int splits) {
  __shared__ float4 sm[TPB];
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int e = (blockIdx.x * 64 + col) * 4;
  float4 acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e + 3 < wsize) {
    int c = sl;
    for (; c + 28 < splits; c += 32) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 v = *(const float4*)(part + (size_t)(c + 4 * k) * wsize + e);
        acc[k].x += v.x; acc[k].y += v.y; acc[k].z += v.z; acc[k].w += v.w;
      }
    }
    for (; c < splits; c += 4) {
      const float4 v = *(const float4*)(part + (size_t)c * wsize + e);
      acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
    }
  }
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 8; ++k) { s.x += acc[k].x; s.y += acc[k].y; s.z += acc[k].z; s.w += acc[k].w; }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (sl == 0 && e + 3 < wsize) {
    float4 t = sm[col];
    for (int l = 1; l < 4; ++l) { const float4 v = sm[l * 64 + col]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    *(float4*)(out + e) = t;
  }
}

template <int NT>
__global__ void __launch_bounds__(TPB) producer(float* __restrict__ part, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (size_t)gridDim.x * TPB) {
    const float v = (float)(i & 1023) * 1e-3f;
    float* p = part + i * 4;
    if (NT) { __builtin_nontemporal_store(v, p); __builtin_nontemporal_store(v, p + 1); __builtin_nontemporal_store(v, p + 2); __builtin_nontemporal_store(v, p + 3); }
    else *(float4*)p = make_float4(v, v, v, v);
  }
}

int main() {
  struct Case { int wsize, splits; } cases[] = {{36864, 128}, {2304, 512}, {4608, 745}, {147456, 32}, {589824, 8}, {9216, 512}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto cs : cases) {
    float *part, *out, *junk;
    size_t n = (size_t)cs.wsize * cs.splits;
    hipMalloc(&part, n * 4); hipMalloc(&out, cs.wsize * 4); hipMalloc(&junk, 256u << 20);
    hipMemset(part, 0, n * 4);
    auto run = [&](const char* name, auto launch) {
      float tot = 0.f; const int reps = 20;
      for (int r = 0; r < reps + 2; ++r) {
        hipMemsetAsync(junk, r, 256u << 20, 0);           // evict caches between runs (cold partials, as after wgrad)
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) tot += ms;
      }
      printf("wsize %7d splits %4d (%.1f MB) %-10s %.1f us\n", cs.wsize, cs.splits, n * 4 / 1e6, name, tot / reps * 1e3);
    };
    auto run_after = [&](const char* name, int nt) {
      float tot = 0.f; const int reps = 20;
      for (int r = 0; r < reps + 2; ++r) {
        hipMemsetAsync(junk, r, 256u << 20, 0);
        if (nt) producer<1><<<512, TPB>>>(part, n / 4); else producer<0><<<512, TPB>>>(part, n / 4);
        hipEventRecord(e0); sum_cur<16><<<(cs.wsize + 63) / 64, TPB>>>(part, out, cs.wsize, cs.splits); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) tot += ms;
      }
      printf("wsize %7d splits %4d (%.1f MB) %-10s %.1f us\n", cs.wsize, cs.splits, n * 4 / 1e6, name, tot / reps * 1e3);
    };
    run_after("after-st", 0);
    run_after("after-nt", 1);
    run("cur<4>", [&] { sum_cur<4><<<(cs.wsize + 15) / 16, TPB>>>(part, out, cs.wsize, cs.splits); });
    run("cur<16>", [&] { sum_cur<16><<<(cs.wsize + 63) / 64, TPB>>>(part, out, cs.wsize, cs.splits); });
    run("cur<1>", [&] { sum_cur<1><<<(cs.wsize + 3) / 4, TPB>>>(part, out, cs.wsize, cs.splits); });
    run("wide", [&] { sum_wide<<<(cs.wsize + 255) / 256, TPB>>>(part, out, cs.wsize, cs.splits); });
    hipFree(part); hipFree(out); hipFree(junk);
  }
  return 0;
}
