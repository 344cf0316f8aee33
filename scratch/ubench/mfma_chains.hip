// Microbenchmark: fp32 MFMA issue rate vs number of independent accumulator chains and waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ void __launch_bounds__(256) k16(float* out, int iters, float a0, float b0) {
  f32x4 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) acc[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 123.456f) out[threadIdx.x] = s;
}
template <int CH>
__global__ void __launch_bounds__(256) k32(float* out, int iters, float a0, float b0) {
  f32x16 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[c][q];
  if (s == 123.456f) out[threadIdx.x] = s;
}
template <typename F>
void run(const char* name, F kern, int ch, int wgs_per_cu, double flop_per_mfma) {
  float* out; hipMalloc(&out, 4096);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256 * wgs_per_cu), dim3(256), 0, 0, out, iters, 1.0f, 2.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)256 * wgs_per_cu * 4 * iters * 8 * ch * flop_per_mfma;
  printf("%s chains=%d waves/SIMD=%d: %.1f TF (%.2f ms)\n", name, ch, wgs_per_cu, fl / ms / 1e9, ms);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4}) {
    run("16x16x4", k16<1>, 1, w, 2048); run("16x16x4", k16<2>, 2, w, 2048);
    run("16x16x4", k16<4>, 4, w, 2048); run("16x16x4", k16<8>, 8, w, 2048);
  }
  for (int w : {1, 2, 4}) {
    run("32x32x2", k32<1>, 1, w, 4096); run("32x32x2", k32<2>, 2, w, 4096); run("32x32x2", k32<4>, 4, w, 4096);
  }
  return 0;
}
