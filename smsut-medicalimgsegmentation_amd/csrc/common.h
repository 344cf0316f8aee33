// Shared helpers for the SMSUT gfx950 kernels.  All tensors are dense fp32 NHWC unless stated.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SMSUT_OK 0
#define SMSUT_EINVAL (-1)

#define SMSUT_REQUIRE(cond) \
  do {                      \
    if (!(cond)) return SMSUT_EINVAL; \
  } while (0)

#define SMSUT_LAUNCH_CHECK()                 \
  do {                                       \
    hipError_t e__ = hipGetLastError();      \
    if (e__ != hipSuccess) return (int)e__;  \
  } while (0)

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// grid-stride launch width for memory-bound elementwise kernels: 256 CUs x 8 blocks (guide G11)
#ifndef SMSUT_EW_GRID_CAP
#define SMSUT_EW_GRID_CAP 2048
#endif
static inline int ew_grid(int64_t work_items, int block = 256) {
  int64_t g = cdiv64(work_items, block);
  if (g > SMSUT_EW_GRID_CAP) g = SMSUT_EW_GRID_CAP;
  if (g < 1) g = 1;
  return (int)g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for 256-thread blocks (4 waves); result valid in every thread
__device__ __forceinline__ float block_sum_256(float v, float* sm4) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm4[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm4[0] + sm4[1] + sm4[2] + sm4[3];
}
__device__ __forceinline__ double block_sum_256_d(double v, double* sm4) {
  v = wave_sum_d(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm4[threadIdx.x >> 6] = v;
  __syncthreads();
  return sm4[0] + sm4[1] + sm4[2] + sm4[3];
}
__device__ __forceinline__ float block_max_256(float v, float* sm4) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sm4[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sm4[0], sm4[1]), fmaxf(sm4[2], sm4[3]));
}

__device__ __forceinline__ float lrelu_f(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float lrelu_mask(float y, float slope) { return y > 0.f ? 1.f : slope; }

// The normalised pre-activation of InstanceNorm.  ONE definition shared by the forward, the backward (which recomputes
// the LeakyReLU mask from x -- the sign of this value -- instead of reading the activation back from HBM) and the conv
// epilogue that folds the backward statistics in: it must be bit-identical everywhere.
__device__ __forceinline__ float in_affine(float x, float mean, float rstd, float gamma, float beta) {
  return __fmaf_rn(x - mean, rstd * gamma, beta);
}


// ---- virtual channel concat ------------------------------------------------------------------------------------------------
// A kernel input that is logically cat([xa, xb], channel) (UpSampleAndConcat, network/blocks.py:49-50) can be read from
// the two tensors in place: channel c of the cat lives in xa [.., ca] for c < ca, else in xb [.., Ctot - ca] at c - ca.
// Every kernel below picks the source per 4- or 16-channel unit, which never straddles the seam (ca % 16 == 0).
struct CatSrc { const float* p; int stride; int coff; };      // element (pix, c) = p[pix * stride + c - coff]
__device__ __forceinline__ CatSrc cat_src(const float* x, const float* x2, int Ctot, int ca, int c) {
  if (!x2) return CatSrc{x, Ctot, 0};
  return c < ca ? CatSrc{x, ca, 0} : CatSrc{x2, Ctot - ca, ca};
}
