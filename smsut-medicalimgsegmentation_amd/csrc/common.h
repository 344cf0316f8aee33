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

// ---- in-launch InstanceNorm finalize by the last-arriving workgroup (r05) -----------------------------------------------------
// A statistics-producing conv kernel leaves per-tile partials {sum, sum of squares} (or the backward pair) and a SEPARATE launch
// (in_moments_final, norm.hip: 4.7 us at the dependent-launch floor, ~220 of them per uganConsis iteration) used to combine
// them per (image, channel).  With a FinRef the workgroup whose partials complete an image runs that same fixed-order fp64
// combine itself, at the end of the producing kernel:
//   * partials are stored WRITE-THROUGH (8-byte sc1 stores: they leave the XCD's L2 at once, no release fence -- r02's variant
//     of this, `__threadfence()` in every workgroup, wrote back whole L2s and cost 15-27 % of the step);
//   * the storing wave drains (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, ONE lane adds the workgroup's tile count
//     to the image's ticket (agent-scope atomic; the eight XCD L2s do not see each other's plain stores);
//   * the workgroup whose add completes the image -- told by the value the add returned -- reads every partial of the image with
//     sc1 loads (they bypass this CU's L1, which may hold stale lines of the buffer) after a barrier the adding lane joins,
//     combines in the order in_moments_final uses (bit-identical results) and resets the ticket to 0 for the next launch.
// (MI355X_MICROARCH.md "inter-workgroup visibility", valid form: one agent-scope add per storing workgroup, last arriver loads
// sc1.)  tickets: int [N], ZERO on entry, zero again on exit; null = no in-launch finalize (plain stores, separate launch).
struct FinRef {
  int* tickets;
  float* o0; float* o1;          // [N][C]: (mean, rstd) of the forward statistics | (mean gz, mean gz * xhat) of the backward pair
  float* s0; float* s1;          // fused shortcut: the second statistics set's (mean, rstd); null otherwise
  float eps;
};

typedef unsigned long long smsut_u64;
__device__ __forceinline__ void st_sc1_f2(float* p, float a, float b) {      // 8-byte write-through store
  const smsut_u64 bits = ((smsut_u64)__float_as_uint(b) << 32) | (smsut_u64)__float_as_uint(a);
  __hip_atomic_store((smsut_u64*)p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// sc1 loads of the partials through a buffer descriptor over ONE image's block: the per-lane offset is one VGPR, the row a scalar
// offset (no 64-bit address per load in flight: the tail of a 72-register kernel must not raise its register count), and rows past
// the block's end read as 0 (descriptor range check) -- exactly the "+ 0.0" in_moments_final adds for them.
typedef __amdgpu_buffer_rsrc_t smsut_rsrc_t;
__device__ __forceinline__ smsut_rsrc_t fin_rsrc(const float* p, int bytes) {
  const smsut_u64 v = (smsut_u64)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((smsut_u64)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
typedef unsigned smsut_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ld_sc1_f2(smsut_rsrc_t r, int voff, int soff, float& a, float& b) {
  const smsut_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 16);       // aux 16 = sc1
  a = __uint_as_float(v[0]); b = __uint_as_float(v[1]);
}

// Combine one image's partials part[chunks][C][2] -> o0[C], o1[C].  FWD: (mean, rstd = 1/sqrt(var + eps)); else the two means.
// Whole workgroup (256 threads), sm: >= 256 * 2 doubles of LDS nobody else uses any more.  Summation order = in_moments_final_body
// (norm.hip): per channel, chunk lanes l = 0..15 each add rows l, l+16, ... in fp64, then the 16 lane sums in lane order -- with
// <= 16 chunks that is the plain sum in chunk order, done by ONE thread per channel without a barrier.
template <bool FWD>
__device__ __forceinline__ void fin_image(const float* part, int chunks, int C, int HW, float eps, float* o0, float* o1, double* sm) {
  const double inv = 1.0 / (double)HW;
  const smsut_rsrc_t rs = fin_rsrc(part, chunks * C * 8);
  auto emit = [&](int c, double t0, double t1) {
    if (FWD) {
      const double m = t0 * inv;
      double var = t1 * inv - m * m;
      if (var < 0.0) var = 0.0;
      o0[c] = (float)m;
      o1[c] = (float)(1.0 / sqrt(var + (double)eps));
    } else {
      o0[c] = (float)(t0 * inv);
      o1[c] = (float)(t1 * inv);
    }
  };
  constexpr int U = 16;
  if (chunks <= 16) {
    const int rowb = __builtin_amdgcn_readfirstlane(C * 8);
    for (int c = threadIdx.x; c < C; c += 256) {
      float a[U], b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) ld_sc1_f2(rs, c * 8, u * rowb, a[u], b[u]);
      double t0 = 0.0, t1 = 0.0;
#pragma unroll
      for (int u = 0; u < U; ++u) { t0 += (double)a[u]; t1 += (double)b[u]; }
      emit(c, t0, t1);
    }
    return;
  }
  const int col = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int rowb16 = __builtin_amdgcn_readfirstlane(C * 8 * 16);
  for (int cb = 0; cb < C; cb += 16) {                       // (uniform: barriers inside)
    const int c = cb + col;
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
      for (int ch0 = 0; ch0 < chunks; ch0 += U * 16) {       // rows cl + ch0 + 16 u: all U in flight
        float a[U], b[U];
        const int voff = ((cl + ch0) * C + c) * 8;
#pragma unroll
        for (int u = 0; u < U; ++u) ld_sc1_f2(rs, voff, u * rowb16, a[u], b[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) { s0 += (double)a[u]; s1 += (double)b[u]; }
      }
    }
    sm[threadIdx.x * 2] = s0; sm[threadIdx.x * 2 + 1] = s1;
    __syncthreads();
    if (cl == 0 && c < C) {
      double t0 = 0.0, t1 = 0.0;
      for (int l = 0; l < 16; ++l) { t0 += sm[(l * 16 + col) * 2]; t1 += sm[(l * 16 + col) * 2 + 1]; }
      emit(c, t0, t1);
    }
    __syncthreads();
  }
}

// End of a statistics-producing persistent kernel: this workgroup produced the partials of items [item0, item1) (item = image *
// tiles_img + tile) for its channel group; `units_per_image` = tiles_img * (channel groups of the grid).  stats / stats2: the
// partial buffers [N][tiles_img][C][2] (stats2 with fin.s0: the fused shortcut's).  Called by ALL threads, after the last partial
// store; `flag`: one int of LDS, sm as fin_image.
template <bool FWD>
__device__ __forceinline__ void fin_tail(const FinRef& fin, const float* stats, const float* stats2, int item0, int item1, int tiles_img,
                                         int units_per_image, int C, int HW, int* flag, double* sm) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave: its write-through partial stores have left
  __syncthreads();
  const int n0 = item0 / tiles_img, n1 = (item1 - 1) / tiles_img;
  for (int n = n0; n <= n1; ++n) {                           // (uniform)
    const int lo = max(item0, n * tiles_img), hi = min(item1, (n + 1) * tiles_img);
    if (threadIdx.x == 0) {
      const int k = hi - lo;
      const int t = __hip_atomic_fetch_add(fin.tickets + n, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = (t + k == units_per_image) ? 1 : 0;
    }
    __syncthreads();                                         // the adding lane joins: its add has returned
    const bool last = *flag != 0;
    __syncthreads();                                         // (flag is rewritten for the next image)
    if (last) {
      fin_image<FWD>(stats + (size_t)n * tiles_img * C * 2, tiles_img, C, HW, fin.eps, fin.o0 + (size_t)n * C, fin.o1 + (size_t)n * C, sm);
      if (FWD && fin.s0) {
        __syncthreads();
        fin_image<true>(stats2 + (size_t)n * tiles_img * C * 2, tiles_img, C, HW, fin.eps, fin.s0 + (size_t)n * C, fin.s1 + (size_t)n * C, sm);
      }
      if (threadIdx.x == 0) __hip_atomic_store(fin.tickets + n, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
    }
  }
}

// ... and for the THREE sums of the residual tail's backward (partials [chunks][C][3]; in_moments_final<2>'s order): the means go
// to o0, o1, o2 [C].  12-byte sc1 loads through the same descriptor scheme.
typedef unsigned smsut_u32x3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ void fin_image3(const float* part, int chunks, int C, int HW, float* o0, float* o1, float* o2, double* sm) {
  const double inv = 1.0 / (double)HW;
  const smsut_rsrc_t rs = fin_rsrc(part, chunks * C * 12);
  auto ld = [&](int voff, int soff, float (&v)[3]) __attribute__((always_inline)) {
    const smsut_u32x3 q = __builtin_amdgcn_raw_buffer_load_b96(rs, voff, soff, 16);    // aux 16 = sc1
    v[0] = __uint_as_float(q[0]); v[1] = __uint_as_float(q[1]); v[2] = __uint_as_float(q[2]);
  };
  constexpr int U = 16;
  if (chunks <= 16) {
    const int rowb = __builtin_amdgcn_readfirstlane(C * 12);
    for (int c = threadIdx.x; c < C; c += 256) {
      float v[U][3];
#pragma unroll
      for (int u = 0; u < U; ++u) ld(c * 12, u * rowb, v[u]);
      double t[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] += (double)v[u][k];
      o0[c] = (float)(t[0] * inv); o1[c] = (float)(t[1] * inv); o2[c] = (float)(t[2] * inv);
    }
    return;
  }
  const int col = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int rowb16 = __builtin_amdgcn_readfirstlane(C * 12 * 16);
  for (int cb = 0; cb < C; cb += 16) {                       // (uniform: barriers inside)
    const int c = cb + col;
    double s[3] = {0.0, 0.0, 0.0};
    if (c < C) {
      for (int ch0 = 0; ch0 < chunks; ch0 += U * 16) {
        float v[U][3];
        const int voff = ((cl + ch0) * C + c) * 12;
#pragma unroll
        for (int u = 0; u < U; ++u) ld(voff, u * rowb16, v[u]);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int k = 0; k < 3; ++k) s[k] += (double)v[u][k];
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) sm[threadIdx.x * 3 + k] = s[k];
    __syncthreads();
    if (cl == 0 && c < C) {
      double t[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double a = 0.0;
        for (int l = 0; l < 16; ++l) a += sm[(l * 16 + col) * 3 + k];
        t[k] = a;
      }
      o0[c] = (float)(t[0] * inv); o1[c] = (float)(t[1] * inv); o2[c] = (float)(t[2] * inv);
    }
    __syncthreads();
  }
}
__device__ __forceinline__ void st_sc1_f(float* p, float a) {                 // 4-byte write-through store
  __hip_atomic_store((unsigned*)p, __float_as_uint(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
