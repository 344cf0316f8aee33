// InstanceNorm2d(affine) on NHWC fp32: forward, backward and backward-of-backward (for WGAN-GP).
//
// Reference semantics: nn.InstanceNorm2d(C, affine=True), eps 1e-5, biased variance, no running
// stats (network/blocks.py:23), followed on the hot path by LeakyReLU(0.01) (blocks.py:28-32).
//
// Reductions are deterministic two-stage (per-chunk fp32 partials -> fp64 combine), no atomics.
// Per-(n,c) quantities with M = H*W, xh = (x-mean)*rstd, gz = gy * lrelu'(y):
//   fwd :  y  = act(xh*gamma + beta)          (lrelu'(y) below is recomputed from x: sign of xh*gamma + beta)
//   bwd :  a = mean(gz), b = mean(gz*xh);  gx = gamma*rstd*(gz - a - xh*b);
//          ggamma = sum_n M*b, gbeta = sum_n M*a
//   bwd2:  given v = d/dgx, ug = d/dggamma, ub = d/dgbeta:
//          cv = mean(v), dv = mean(v*xh), e = mean(v*gz), S = e - cv*a - dv*b
//          d/dgy    = lrelu'(y) * (gamma*rstd*(v - cv - xh*dv) + ug*xh + ub)
//          d/dx     = -gamma*rstd^2*(S*xh + b*(v-cv) + dv*(gz-a) - 2*b*dv*xh) + ug*rstd*(gz - a - xh*b)
//          d/dgamma = sum_n rstd*M*S
#include "common.h"

namespace {

constexpr int TPB = 256;

// in_affine() -- the normalised pre-activation, ONE definition for forward, backward and the conv epilogues: common.h

// MODE 0: sums of (x, x^2)           -- forward statistics
// MODE 1: sums of (gz, gz*xh)        -- backward
// MODE 2: sums of (v, v*xh, v*gz)    -- backward of backward
template <int MODE> struct NSums { static constexpr int n = (MODE == 2) ? 3 : 2; };

// One-chunk case (planes of <= 1024 pixels: pick_chunk): the workgroup that reduces (image n, channel c) holds the COMPLETE sums, so
// it writes the per-(n, c) means itself -- exactly what in_moments_final<MODE> would compute from this single chunk -- and that
// launch (4.9 us at the dependent-launch floor) is skipped.  o0 == null: not fused.
struct FinOut { float* o0; float* o1; float* o2; float eps; };
template <int MODE>
__device__ __forceinline__ void fin_emit(const FinOut& f, int HW, int k, const float* tot) {   // k = n * C + c; tot[NS]
  constexpr int NS = NSums<MODE>::n;
  const double inv = 1.0 / (double)HW;
  if (MODE == 0) {
    const double m = (double)tot[0] * inv;
    double var = (double)tot[1] * inv - m * m;
    if (var < 0.0) var = 0.0;
    f.o0[k] = (float)m;
    f.o1[k] = (float)(1.0 / sqrt(var + (double)f.eps));
  } else {
    f.o0[k] = (float)((double)tot[0] * inv);
    f.o1[k] = (float)((double)tot[1] * inv);
    if (MODE == 2) f.o2[k] = (float)((double)tot[NS - 1] * inv);
  }
}

// POOLG (MODE 1, r05): t0 is the gradient of the 2x2-AVERAGE-POOLED activation, [N, H/2, W/2, C]; the full-resolution gradient
// 0.25 * t0[h/2][w/2] (what k_avgpool_bwd, pointwise.hip, would have written) is formed while loading -- pw = W.
template <int MODE, int VEC, bool POOLG = false>
__global__ void __launch_bounds__(TPB, 4)   // <= 128 VGPRs: 4 waves per SIMD (it sat at 172 = 2 waves)
in_moments_partial(const float* __restrict__ t0,   // x | gy | v
                   const float* __restrict__ t1,   // - | x  | x
                   const float* __restrict__ t2,   // - | -  | gy
                   const float* __restrict__ gamma, const float* __restrict__ beta,   // beta == null: no activation
                   const float* __restrict__ mean, const float* __restrict__ rstd,
                   float* __restrict__ part,        // [N][chunks][C][NS]
                   int HW, int C, int pix_per_chunk, float slope, FinOut fin = FinOut{}, int pw = 0) {
  static_assert(!POOLG || MODE == 1, "pooled gradient: first-order backward sums only");
  constexpr int NS = NSums<MODE>::n;
  const bool emit = fin.o0 != nullptr;              // (host: only with gridDim.x == 1)
  const int n = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
  // gridDim.z channel slabs (host: slab_count): small planes with many channels (discriminator / bottleneck levels) had only
  // N * chunks workgroups, each walking ALL channels serially -- e.g. 64 workgroups for 16 x 32x32 x 128
  const int CVA = C / VEC;                    // all channel groups
  const int CV = CVA / gridDim.z;             // ... of this workgroup's slab (gridDim.z divides CVA)
  const int cvb = blockIdx.z * CV;
  const int TC = CV < TPB ? CV : TPB;
  const int rows = TPB / TC;
  const int tc = threadIdx.x % TC, trow = threadIdx.x / TC;
  const int p0 = chunk * pix_per_chunk;
  const int p1 = min(p0 + pix_per_chunk, HW);
  __shared__ float sm[TPB * 4 * 3];
  const size_t base = (size_t)n * HW * C;

  for (int cv0 = 0; cv0 < CV; cv0 += TC) {   // uniform trip count: barriers inside
    const int cv = cvb + cv0 + tc;
    const bool cv_ok = cv0 + tc < CV;
    float acc[NS][VEC];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[s][j] = 0.f;
    float mu[VEC], rs[VEC], gm[VEC], bt[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      mu[j] = (MODE == 0 || !cv_ok) ? 0.f : mean[n * C + cv * VEC + j];
      rs[j] = (MODE == 0 || !cv_ok) ? 1.f : rstd[n * C + cv * VEC + j];
      gm[j] = (MODE == 0 || !cv_ok || !beta) ? 1.f : gamma[cv * VEC + j];
      bt[j] = (MODE == 0 || !cv_ok || !beta) ? 0.f : beta[cv * VEC + j];
    }
    if (trow < rows && cv_ok) {
      // U rows in flight per thread (one float4 per tensor per row): with a single row the loop was load -> wait ->
      // accumulate, ~2 TB/s.  Rows are accumulated in the same order as before (bit-identical sums).
      constexpr int U = (MODE == 0) ? 4 : 2;
      auto load = [&](int p, float* a0, float* a1, float* a2) {
        const size_t off = base + (size_t)p * C + cv * VEC;
        if constexpr (POOLG) {
          const int ph = p / pw, pc = p - ph * pw;
          const size_t goff = ((size_t)n * (HW >> 2) + (size_t)(ph >> 1) * (pw >> 1) + (pc >> 1)) * C + cv * VEC;
          if constexpr (VEC == 4) { *(float4*)a0 = *(const float4*)(t0 + goff); *(float4*)a1 = *(const float4*)(t1 + off); }
          else { a0[0] = t0[goff]; a1[0] = t1[off]; }
#pragma unroll
          for (int j = 0; j < VEC; ++j) a0[j] *= 0.25f;
        } else if constexpr (VEC == 4) {
          *(float4*)a0 = *(const float4*)(t0 + off);
          if (MODE >= 1) *(float4*)a1 = *(const float4*)(t1 + off);
          if (MODE == 2) *(float4*)a2 = *(const float4*)(t2 + off);
        } else {
          a0[0] = t0[off];
          if (MODE >= 1) a1[0] = t1[off];
          if (MODE == 2) a2[0] = t2[off];
        }
      };
      auto accum = [&](const float* a0, const float* a1, const float* a2) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          if (MODE == 0) {
            acc[0][j] += a0[j];
            acc[1][j] += a0[j] * a0[j];
          } else if (MODE == 1) {
            const float gz = beta ? a0[j] * lrelu_mask(in_affine(a1[j], mu[j], rs[j], gm[j], bt[j]), slope) : a0[j];
            const float xh = (a1[j] - mu[j]) * rs[j];
            acc[0][j] += gz;
            acc[1][j] += gz * xh;
          } else {
            const float gz = beta ? a2[j] * lrelu_mask(in_affine(a1[j], mu[j], rs[j], gm[j], bt[j]), slope) : a2[j];
            const float xh = (a1[j] - mu[j]) * rs[j];
            acc[0][j] += a0[j];
            acc[1][j] += a0[j] * xh;
            acc[NS - 1][j] += a0[j] * gz;
          }
        }
      };
      int p = p0 + trow;
      for (; p + (U - 1) * rows < p1; p += U * rows) {
        float a0[U][VEC], a1[U][VEC], a2[U][VEC];
#pragma unroll
        for (int u = 0; u < U; ++u) load(p + u * rows, a0[u], a1[u], a2[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) accum(a0[u], a1[u], a2[u]);
      }
      for (; p < p1; p += rows) {
        float a0[VEC], a1[VEC], a2[VEC];
        load(p, a0, a1, a2);
        accum(a0, a1, a2);
      }
    }
    __syncthreads();
    if (TC <= 64 && (TC & (TC - 1)) == 0) {
      // lanes of a wave that share a channel group sit TC apart: xor-tree inside the wave, then 4 wave totals through
      // LDS (the old form below left 64 x 12 serial LDS reads to 4 threads of the block -- as long as the main loop)
      for (int off = TC; off < 64; off <<= 1) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[s][j] += __shfl_xor(acc[s][j], off, 64);
      }
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      if (lane < TC) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int j = 0; j < VEC; ++j) sm[((wv * TC + lane) * NS + s) * VEC + j] = acc[s][j];
      }
      __syncthreads();
      if (threadIdx.x < TC && cv_ok) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float tots[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            float tot = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) tot += sm[((w4 * TC + tc) * NS + s) * VEC + j];
            part[(((size_t)n * chunks + chunk) * C + cv * VEC + j) * NS + s] = tot;
            tots[s] = tot;
          }
          if (emit) fin_emit<MODE>(fin, HW, n * C + cv * VEC + j, tots);
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < VEC; ++j) sm[(threadIdx.x * NS + s) * VEC + j] = acc[s][j];
      __syncthreads();
      if (trow == 0 && cv_ok) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float tots[NS];
#pragma unroll
          for (int s = 0; s < NS; ++s) {
            float tot = 0.f;
            for (int r = 0; r < rows; ++r) tot += sm[((r * TC + tc) * NS + s) * VEC + j];
            part[(((size_t)n * chunks + chunk) * C + cv * VEC + j) * NS + s] = tot;
            tots[s] = tot;
          }
          if (emit) fin_emit<MODE>(fin, HW, n * C + cv * VEC + j, tots);
        }
      }
    }
    __syncthreads();
  }
}

// combine chunk partials in fp64 -> per-(n,c) means.  MODE 0 writes (mean, rstd).
// 16 channels x 16 chunk-lanes per block: independent loads in flight, fixed-order tree (deterministic).
template <int MODE>
__device__ __forceinline__ void in_moments_final_body(const float* __restrict__ part, int chunks, int C, int HW, float eps,
                                                      float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2) {
  constexpr int NS = NSums<MODE>::n;
  __shared__ double sm[TPB * 3];
  const int n = blockIdx.y;
  const int col = threadIdx.x & 15, cl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + col;
  double s[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) s[k] = 0.0;
  if (c < C) {
    // 8 chunk rows in flight per thread (the one-row loop was a chain of dependent load latencies: 6 us for a 16-block
    // grid that 150 launches per uganConsis step wait on); rows are added in the same order as before
    // (r03: 16 -- a 256-chunk image, the 256x256 planes, is ONE round of loads instead of two: 4.8 -> 4.1 us per launch)
    constexpr int U = 16;
    const float* p0 = part + ((size_t)n * chunks * C + c) * NS;
    const size_t rstride = (size_t)C * NS;
    for (int ch = cl; ch < chunks; ch += U * 16) {         // every round has all its U rows in flight; rows past the end add 0
      float v[U][NS];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = ch + u * 16;
        const float* q = p0 + (size_t)(r < chunks ? r : ch) * rstride;
#pragma unroll
        for (int k = 0; k < NS; ++k) v[u][k] = q[k];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int k = 0; k < NS; ++k) s[k] += (ch + u * 16 < chunks) ? (double)v[u][k] : 0.0;
    }
  }
#pragma unroll
  for (int k = 0; k < NS; ++k) sm[threadIdx.x * 3 + k] = s[k];
  __syncthreads();
  if (cl != 0 || c >= C) return;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    double t = 0.0;
    for (int l = 0; l < 16; ++l) t += sm[(l * 16 + col) * 3 + k];
    s[k] = t;
  }
  const double inv = 1.0 / (double)HW;
  if (MODE == 0) {
    const double m = s[0] * inv;
    double var = s[1] * inv - m * m;
    if (var < 0.0) var = 0.0;
    o0[n * C + c] = (float)m;
    o1[n * C + c] = (float)(1.0 / sqrt(var + (double)eps));
  } else {
    o0[n * C + c] = (float)(s[0] * inv);
    o1[n * C + c] = (float)(s[1] * inv);
    if (MODE == 2) o2[n * C + c] = (float)(s[NS - 1] * inv);
  }
}
template <int MODE>
__global__ void __launch_bounds__(TPB)
in_moments_final(const float* __restrict__ part, int chunks, int C, int HW, float eps,
                 float* __restrict__ o0, float* __restrict__ o1, float* __restrict__ o2) {
  in_moments_final_body<MODE>(part, chunks, C, HW, eps, o0, o1, o2);
}
// TWO statistics sets of the same (N, C, HW) in one launch (blockIdx.z picks the set): conv2's and the shortcut's partials of a
// BasicBlock are both complete when the residual tail needs them -- one launch of this latency-bound kernel instead of two.
__global__ void __launch_bounds__(TPB)
in_moments_final_pair(const float* __restrict__ pa, int chunks_a, float* __restrict__ mean_a, float* __restrict__ rstd_a,
                      const float* __restrict__ pb, int chunks_b, float* __restrict__ mean_b, float* __restrict__ rstd_b, int C,
                      int HW, float eps) {
  if (blockIdx.z == 0) in_moments_final_body<0>(pa, chunks_a, C, HW, eps, mean_a, rstd_a, nullptr);
  else in_moments_final_body<0>(pb, chunks_b, C, HW, eps, mean_b, rstd_b, nullptr);
}

// VEC consecutive per-channel values (statistics / affine parameters) as ONE load: c is a multiple of VEC and the
// tables are 16-byte aligned, so the 4 channels of a float4 lane are a float4 here too (4x fewer load instructions)
template <int VEC>
__device__ __forceinline__ void ldv(const float* __restrict__ p, int idx, float (&o)[VEC]) {
  if constexpr (VEC == 4) *(float4*)o = *(const float4*)(p + idx); else o[0] = p[idx];
}

// ---- per-image walk of the apply kernels ------------------------------------------------------------------------------
// grid = (blocks per image, N): the image index is blockIdx.y and a thread's vector units are li = x0 + k*stride inside
// the image (32-bit).  When the stride is a multiple of CV = C/VEC (always, for power-of-two channel counts: the stride is
// a multiple of 256), a thread keeps ONE channel group for the whole loop, so the per-(n, c) statistics and affine
// parameters are loaded once per thread instead of once per float4 of data, and there is no per-element division.
// (Before: i % CV and i / (CV*HW) in 64 bits per element -- ~250 integer instructions per 16 bytes, a VALU time equal
// to the HBM time of these kernels.)
inline dim3 img_grid(int64_t per_img_units, int N) {
  int cap = SMSUT_EW_GRID_CAP / (N > 0 ? N : 1);
  if (cap < 1) cap = 1;
  int64_t bx = cdiv64(per_img_units, TPB);
  if (bx > cap) bx = cap;
  if (bx < 1) bx = 1;
  return dim3((unsigned)bx, (unsigned)N);
}

template <class LoadP, class Body>
__device__ __forceinline__ void img_walk(int HW, int CV, LoadP&& load_params, Body&& body) {
  const int n = blockIdx.y;
  const int per_img = HW * CV;
  const int64_t ibase = (int64_t)n * per_img;
  const int stride = gridDim.x * TPB;
  const int x0 = blockIdx.x * TPB + threadIdx.x;
  if (stride % CV == 0) {
    if (x0 < per_img) {
      auto prm = load_params(n, x0 % CV);
      for (int li = x0; li < per_img; li += stride) body(ibase + li, prm);
    }
  } else {
    for (int li = x0; li < per_img; li += stride) {
      auto prm = load_params(n, li % CV);
      body(ibase + li, prm);
    }
  }
}

// HS ("half storage", config 5): the raw conv outputs y1, y2 and s of a BasicBlock live in HBM as fp16 (written by the conv epilogues,
// smsut_conv2d_fwd_mfma_stats*_f16_hs); the x / y2 / s pointers of the HS kernel variants then point at _Float16 data.  Arithmetic
// stays fp32.
typedef _Float16 hs4 __attribute__((ext_vector_type(4)));
template <int VEC, bool HS>
__device__ __forceinline__ void ld_act(const float* base, size_t off, float* v) {       // off in ELEMENTS
  if constexpr (HS) {
    const _Float16* h = reinterpret_cast<const _Float16*>(base) + off;
    if constexpr (VEC == 4) { const hs4 t = *(const hs4*)h; v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3]; }
    else v[0] = (float)h[0];
  } else {
    if constexpr (VEC == 4) *(float4*)v = *(const float4*)(base + off); else v[0] = base[off];
  }
}

template <int VEC, bool HS = false, bool YH = false>       // YH: y is stored as fp16 too
__global__ void __launch_bounds__(TPB)
in_apply_fwd(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
             const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
             int HW, int C, float slope, int has_act) {
  struct Prm { float mu[VEC], rs[VEC], gm[VEC], bt[VEC]; };
  img_walk(HW, C / VEC,
    [&](int n, int cv) {
      Prm p;
      ldv<VEC>(mean, n * C + cv * VEC, p.mu); ldv<VEC>(rstd, n * C + cv * VEC, p.rs);
      ldv<VEC>(gamma, cv * VEC, p.gm); ldv<VEC>(beta, cv * VEC, p.bt);
      return p;
    },
    [&](int64_t i, const Prm& p) {
      float v[VEC];
      ld_act<VEC, HS>(x, (size_t)i * VEC, v);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float r = in_affine(v[j], p.mu[j], p.rs[j], p.gm[j], p.bt[j]);
        v[j] = has_act ? lrelu_f(r, slope) : r;
      }
      if constexpr (YH) {
        static_assert(!YH || VEC == 4, "fp16 output: whole channel quads");
        *(hs4*)(reinterpret_cast<_Float16*>(y) + i * 4) = (hs4){(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
      } else if constexpr (VEC == 4) *(float4*)(y + i * 4) = *(float4*)v; else y[i] = v[0];
    });
}

// InstanceNorm + LeakyReLU + AvgPool2d(2) in ONE pass (r05): y[n, ho, wo, c] = mean over the 2x2 window of lrelu(IN(x)) -- bn1 -> relu ->
// avgpool of a stride-2 BottleBlock (reference network/blocks.py:99-107) without writing the activated full-resolution tensor (one
// read of x + a quarter-size write instead of read + write + read + quarter write).  Arithmetic = in_apply_fwd followed by
// k_avgpool_fwd (pointwise.hip): (a00 + a01 + a10 + a11) * 0.25f in that order -- bit-identical to the two kernels.
template <int VEC>
__global__ void __launch_bounds__(TPB)
in_apply_pool_fwd(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                  const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y, int H, int W, int C,
                  float slope) {
  struct Prm { float mu[VEC], rs[VEC], gm[VEC], bt[VEC]; };
  const int Wo = W >> 1, CV = C / VEC, HWo = (H >> 1) * Wo;
  const float* xin = x + (size_t)blockIdx.y * H * W * C;
  img_walk(HWo, CV,
    [&](int n, int cv) {
      Prm p;
      ldv<VEC>(mean, n * C + cv * VEC, p.mu); ldv<VEC>(rstd, n * C + cv * VEC, p.rs);
      ldv<VEC>(gamma, cv * VEC, p.gm); ldv<VEC>(beta, cv * VEC, p.bt);
      return p;
    },
    [&](int64_t i, const Prm& p) {
      const int li = (int)(i - (int64_t)blockIdx.y * HWo * CV);
      const int pp = li / CV, cv = li - pp * CV;
      const int ho = pp / Wo, wo = pp - ho * Wo;
      const float* b = xin + ((size_t)(2 * ho) * W + 2 * wo) * C + cv * VEC;
      float a[4][VEC];
      if constexpr (VEC == 4) {
        *(float4*)a[0] = *(const float4*)b; *(float4*)a[1] = *(const float4*)(b + C);
        *(float4*)a[2] = *(const float4*)(b + (size_t)W * C); *(float4*)a[3] = *(const float4*)(b + (size_t)W * C + C);
      } else {
        a[0][0] = b[0]; a[1][0] = b[C]; a[2][0] = b[(size_t)W * C]; a[3][0] = b[(size_t)W * C + C];
      }
      float v[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) q[k] = lrelu_f(in_affine(a[k][j], p.mu[j], p.rs[j], p.gm[j], p.bt[j]), slope);
        v[j] = (q[0] + q[1] + q[2] + q[3]) * 0.25f;
      }
      if constexpr (VEC == 4) *(float4*)(y + i * 4) = *(float4*)v; else y[i] = v[0];
    });
}

// fp16-operand convolutions scale a gradient operand by a power of two derived from its absolute maximum (smsut_absmax_scale).
// The kernels that WRITE such a gradient hand the maximum over (amax, nullable): every workgroup stores the maximum of what it
// wrote in its own slot amax[block] -- no atomics (a first version with one atomicMax per wave on a single float cost 3 ms per
// config-5 iteration in same-address contention), no zeroing, any order -- and smsut_absmax_finish reduces the slots.
__device__ __forceinline__ void amax_emit(float m, float* slots) {
  __shared__ float sm_amax[TPB / 64];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sm_amax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 1; k < TPB / 64; ++k) m = fmaxf(m, sm_amax[k]);
    slots[blockIdx.y * gridDim.x + blockIdx.x] = m;
  }
  __syncthreads();
}

template <int VEC, bool HS = false, bool AMAX = false, bool POOLG = false>   // AMAX: hand max|gx| over (amax non-null); compile-time, so
__global__ void __launch_bounds__(TPB)                       // that the fp32 path's instantiation carries nothing of it.  POOLG: gy is the
in_apply_bwd(                                                // gradient of the 2x2-average-pooled output (see in_moments_partial), pw = W
const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ beta,
             const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
             const float* __restrict__ am, const float* __restrict__ bm, float* __restrict__ gx,
             int HW, int C, float slope, int N, float* __restrict__ ggamma, float* __restrict__ gbeta,
             float* __restrict__ amax = nullptr, int pw = 0) {
  float mx = 0.f;
  if (ggamma && blockIdx.x == 0 && blockIdx.y == 0) {   // affine gradients ride along in one block: ggamma = sum_n M*b, gbeta = sum_n M*a
    for (int c = threadIdx.x; c < C; c += TPB) {
      double sa = 0.0, sb = 0.0;
      for (int n = 0; n < N; ++n) { sa += (double)am[n * C + c]; sb += (double)bm[n * C + c]; }
      ggamma[c] = (float)(sb * (double)HW);
      gbeta[c] = (float)(sa * (double)HW);
    }
  }
  struct Prm { float mu[VEC], rs[VEC], gm[VEC], bt[VEC], av[VEC], bv[VEC]; };
  img_walk(HW, C / VEC,
    [&](int n, int cv) {
      Prm p;
      const int k0 = n * C + cv * VEC;
      ldv<VEC>(mean, k0, p.mu); ldv<VEC>(rstd, k0, p.rs); ldv<VEC>(gamma, cv * VEC, p.gm);
      ldv<VEC>(am, k0, p.av); ldv<VEC>(bm, k0, p.bv);
      if (beta) ldv<VEC>(beta, cv * VEC, p.bt);
      return p;
    },
    [&](int64_t i, const Prm& p) {
      float g[VEC], xv[VEC];
      if constexpr (POOLG) {
        const int CV = C / VEC;
        const int li = (int)(i - (int64_t)blockIdx.y * HW * CV);          // unit inside the image: pixel * CV + channel group
        const int pix = li / CV, cv = li - pix * CV;
        const int ph = pix / pw, pc = pix - ph * pw;
        const size_t goff = ((size_t)blockIdx.y * (HW >> 2) + (size_t)(ph >> 1) * (pw >> 1) + (pc >> 1)) * C + cv * VEC;
        if constexpr (VEC == 4) *(float4*)g = *(const float4*)(gy + goff); else g[0] = gy[goff];
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[j] *= 0.25f;
      } else if constexpr (VEC == 4) *(float4*)g = *(const float4*)(gy + i * 4); else g[0] = gy[i];
      ld_act<VEC, HS>(x, (size_t)i * VEC, xv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float r = p.rs[j], gm = p.gm[j];
        const float gz = beta ? g[j] * lrelu_mask(in_affine(xv[j], p.mu[j], r, gm, p.bt[j]), slope) : g[j];
        const float xh = (xv[j] - p.mu[j]) * r;
        g[j] = gm * r * (gz - p.av[j] - xh * p.bv[j]);
        if constexpr (AMAX) mx = fmaxf(mx, fabsf(g[j]));
      }
      if constexpr (VEC == 4) *(float4*)(gx + i * 4) = *(float4*)g; else gx[i] = g[0];
    });
  if constexpr (AMAX) amax_emit(mx, amax);
}

template <int VEC>
__global__ void __launch_bounds__(TPB)
in_apply_bwd2(const float* __restrict__ v, const float* __restrict__ x, const float* __restrict__ gy,
              const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
              const float* __restrict__ gamma, const float* __restrict__ am, const float* __restrict__ bm,
              const float* __restrict__ cvm, const float* __restrict__ dvm, const float* __restrict__ em,
              const float* __restrict__ ug, const float* __restrict__ ub,
              float* __restrict__ d_gy, float* __restrict__ d_x,
              int64_t total_vec, int HW, int C, float slope) {
  const int CV = C / VEC;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total_vec; i += (int64_t)gridDim.x * TPB) {
    const int cv = (int)(i % CV);
    const int n = (int)(i / ((int64_t)CV * HW));
    float vv[VEC], xv[VEC], g[VEC], o1[VEC], o2[VEC];
    if constexpr (VEC == 4) {
      *(float4*)vv = *(const float4*)(v + i * 4);
      *(float4*)xv = *(const float4*)(x + i * 4);
      *(float4*)g = *(const float4*)(gy + i * 4);
    } else {
      vv[0] = v[i]; xv[0] = x[i]; g[0] = gy[i];
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = cv * VEC + j;
      const int k = n * C + c;
      const float r = rstd[k], gm = gamma[c];
      const float mk = beta ? lrelu_mask(in_affine(xv[j], mean[k], r, gm, beta[c]), slope) : 1.f;
      const float gz = g[j] * mk;
      const float xh = (xv[j] - mean[k]) * r;
      const float a = am[k], b = bm[k], cvv = cvm[k], dv = dvm[k];
      const float S = em[k] - cvv * a - dv * b;
      const float ugc = ug ? ug[c] : 0.f, ubc = ub ? ub[c] : 0.f;
      o1[j] = mk * (gm * r * (vv[j] - cvv - xh * dv) + ugc * xh + ubc);
      o2[j] = -gm * r * r * (S * xh + b * (vv[j] - cvv) + dv * (gz - a) - 2.f * b * dv * xh) +
              ugc * r * (gz - a - xh * b);
    }
    if constexpr (VEC == 4) {
      *(float4*)(d_gy + i * 4) = *(float4*)o1;
      *(float4*)(d_x + i * 4) = *(float4*)o2;
    } else {
      d_gy[i] = o1[0]; d_x[i] = o2[0];
    }
  }
}

// d/dgamma[c] = sum_n rstd*M*(e - cv*a - dv*b)
__global__ void in_bwd2_gamma(const float* __restrict__ rstd, const float* __restrict__ am, const float* __restrict__ bm,
                              const float* __restrict__ cvm, const float* __restrict__ dvm, const float* __restrict__ em,
                              int N, int C, int HW, float* __restrict__ d_gamma) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0;
  for (int n = 0; n < N; ++n) {
    const int k = n * C + c;
    s += (double)rstd[k] * ((double)em[k] - (double)cvm[k] * am[k] - (double)dvm[k] * bm[k]);
  }
  d_gamma[c] = (float)(s * (double)HW);
}

// ---- fused residual tail of BasicBlock (network/blocks.py:60-79): out = act(IN(y2) + (IN(s) | idn)) -------------------
// forward: one pass.  backward: gz = g_out * act'(out); per-(n,c) sums {gz, gz*y2hat, gz*shat} in one pass, then
// gy2 = g2*r2*(gz - a - y2hat*b2) and gs = gs_*rs*(gz - a - shat*bs) (or g_idn = gz) in one pass.
struct TailRef {
  const float* y2; const float* m2; const float* r2; const float* g2; const float* b2;
  const float* s;  const float* ms; const float* rs; const float* gs; const float* bs;   // ms == null: s is the identity
};

template <int VEC, bool HS = false>
__global__ void __launch_bounds__(TPB)
restail_fwd(TailRef t, float* __restrict__ out, int HW, int C, float slope) {
  struct Prm { float m2[VEC], r2[VEC], g2[VEC], b2[VEC], ms[VEC], rs[VEC], gs[VEC], bs[VEC]; };
  img_walk(HW, C / VEC,
    [&](int n, int cv) {
      Prm p;
      const int c0 = cv * VEC, k0 = n * C + c0;
      ldv<VEC>(t.m2, k0, p.m2); ldv<VEC>(t.r2, k0, p.r2); ldv<VEC>(t.g2, c0, p.g2); ldv<VEC>(t.b2, c0, p.b2);
      if (t.ms) { ldv<VEC>(t.ms, k0, p.ms); ldv<VEC>(t.rs, k0, p.rs); ldv<VEC>(t.gs, c0, p.gs); ldv<VEC>(t.bs, c0, p.bs); }
      return p;
    },
    [&](int64_t i, const Prm& p) {
      float a[VEC], b[VEC];
      ld_act<VEC, HS>(t.y2, (size_t)i * VEC, a); ld_act<VEC, HS>(t.s, (size_t)i * VEC, b);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float u = in_affine(a[j], p.m2[j], p.r2[j], p.g2[j], p.b2[j]);
        const float v = t.ms ? in_affine(b[j], p.ms[j], p.rs[j], p.gs[j], p.bs[j]) : b[j];
        a[j] = lrelu_f(u + v, slope);
      }
      if constexpr (VEC == 4) *(float4*)(out + i * 4) = *(float4*)a; else out[i] = a[0];
    });
}

// The tail of an ENCODER level's block and the level's MaxPool2d(2, 2) in ONE pass (r05; reference network/blocks.py:74-79 +
// 131-133): a thread owns a pooled unit = the 2x2 window of one channel quad, computes the four outputs as restail_fwd does, stores
// them (the skip connection reads `out`), their maximum in k_maxpool_fwd's scan order (pointwise.hip; NaN propagates) and WHERE the
// maximum sat (one byte per channel: 0..3 = (0,0),(0,1),(1,0),(1,1)) -- the pooling pass over `out` disappears, and the backward
// needs neither `out` nor a pass of its own to route the pooled gradient (MaxRef below).
template <bool HS>
__global__ void __launch_bounds__(TPB)
restail_fwd_pool(TailRef t, float* __restrict__ out, float* __restrict__ pooled, unsigned int* __restrict__ idx, int H, int W, int C,
                 float slope) {
  constexpr int VEC = 4;
  struct Prm { float m2[VEC], r2[VEC], g2[VEC], b2[VEC], ms[VEC], rs[VEC], gs[VEC], bs[VEC]; };
  const int Wo = W >> 1, CV = C / VEC, HWo = (H >> 1) * Wo;
  const size_t ibase = (size_t)blockIdx.y * H * W * C;
  img_walk(HWo, CV,
    [&](int n, int cv) {
      Prm p;
      const int c0 = cv * VEC, k0 = n * C + c0;
      ldv<VEC>(t.m2, k0, p.m2); ldv<VEC>(t.r2, k0, p.r2); ldv<VEC>(t.g2, c0, p.g2); ldv<VEC>(t.b2, c0, p.b2);
      if (t.ms) { ldv<VEC>(t.ms, k0, p.ms); ldv<VEC>(t.rs, k0, p.rs); ldv<VEC>(t.gs, c0, p.gs); ldv<VEC>(t.bs, c0, p.bs); }
      return p;
    },
    [&](int64_t i, const Prm& p) {
      const int li = (int)(i - (int64_t)blockIdx.y * HWo * CV);
      const int pp = li / CV, cv = li - pp * CV;
      const int ho = pp / Wo, wo = pp - ho * Wo;
      const size_t o0 = ibase + ((size_t)(2 * ho) * W + 2 * wo) * C + cv * VEC;
      const size_t offs[4] = {o0, o0 + C, o0 + (size_t)W * C, o0 + (size_t)W * C + C};
      float a[4][VEC], b[4][VEC];
#pragma unroll
      for (int k = 0; k < 4; ++k) { ld_act<VEC, HS>(t.y2, offs[k], a[k]); ld_act<VEC, HS>(t.s, offs[k], b[k]); }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float u = in_affine(a[k][j], p.m2[j], p.r2[j], p.g2[j], p.b2[j]);
          const float v = t.ms ? in_affine(b[k][j], p.ms[j], p.rs[j], p.gs[j], p.bs[j]) : b[k][j];
          a[k][j] = lrelu_f(u + v, slope);
        }
        *(float4*)(out + offs[k]) = *(float4*)a[k];
      }
      float m[VEC];
      if (!idx) {                                     // AVERAGE pooling (the discriminator's stride-2 blocks): k_avgpool_fwd's order
#pragma unroll
        for (int j = 0; j < VEC; ++j) m[j] = (a[0][j] + a[1][j] + a[2][j] + a[3][j]) * 0.25f;
        *(float4*)(pooled + i * 4) = *(float4*)m;
        return;
      }
      unsigned int where = 0;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        int kk = 0; float mv = a[0][j];
#pragma unroll
        for (int k = 1; k < 4; ++k) if (a[k][j] > mv || a[k][j] != a[k][j]) { mv = a[k][j]; kk = k; }
        m[j] = mv; where |= (unsigned int)kk << (8 * j);
      }
      *(float4*)(pooled + i * 4) = *(float4*)m;
      idx[i] = where;
    });
}

// MaxRef (r05): the gradient of a block output that went into MaxPool2d(2, 2) AND a skip connection, never materialised:
// g[n,h,w,c] = (idx[n,h/2,w/2,c] == 2 (h & 1) + (w & 1) ? gp[n,h/2,w/2,c] : 0) + gout[n,h,w,c] -- exactly what k_maxpool_bwd with
// `add` (pointwise.hip) would have written (idx from restail_fwd_pool).  gp == null: the plain tensor gout.
struct MaxRef { const float* gp; const unsigned int* idx; int W; };
__device__ __forceinline__ void add_pooled_grad(const MaxRef& mr, int n, int HW, int C, int p, int cv, float* g) {
  // n: image, p: pixel inside the image (h * W + w), cv: float4 channel group; g[4] holds gout's values on entry
  const int h = p / mr.W, w = p - h * mr.W;
  const size_t u = ((size_t)n * (HW >> 2) + (size_t)(h >> 1) * (mr.W >> 1) + (w >> 1)) * (C >> 2) + cv;      // pooled unit
  const float4 gp = *(const float4*)(mr.gp + u * 4);
  if (!mr.idx) {                                      // average pooling: every pixel of the window gets a quarter (k_avgpool_bwd)
    g[0] = gp.x * 0.25f + g[0]; g[1] = gp.y * 0.25f + g[1]; g[2] = gp.z * 0.25f + g[2]; g[3] = gp.w * 0.25f + g[3];
    return;
  }
  const unsigned int wh = mr.idx[u];
  const unsigned int pos = ((h & 1) << 1) | (w & 1);
  g[0] = (((wh) & 0xffu) == pos ? gp.x : 0.f) + g[0];
  g[1] = (((wh >> 8) & 0xffu) == pos ? gp.y : 0.f) + g[1];
  g[2] = (((wh >> 16) & 0xffu) == pos ? gp.z : 0.f) + g[2];
  g[3] = (((wh >> 24) & 0xffu) == pos ? gp.w : 0.f) + g[3];
}

// partial [N][chunks][C][3] = {sum gz, sum gz*y2hat, sum gz*shat}
// FIN (r05; several chunks per image): the workgroup whose partials complete an image (agent-scope ticket per image, common.h) runs
// in_moments_final<2>'s combine itself -- fin.o0 .. o2 then name the outputs of THAT finalize, tickets the image counters.
// MPG (r05): gout is the skip connection's gradient only; the pooled path's is routed in while loading (MaxRef above).
template <int VEC, bool REMASK, bool HS = false, bool FIN = false, bool MPG = false>
__global__ void __launch_bounds__(TPB)       // (a 128-VGPR cap spills here: 244 B scratch and +30 % time)
restail_bwd_partial(const float* __restrict__ gout, const float* __restrict__ out, TailRef t, float* __restrict__ part,
                    int HW, int C, int pix_per_chunk, float slope, FinOut fin = FinOut{}, int* tickets = nullptr,
                    MaxRef mr = MaxRef{nullptr, nullptr, 0}) {
  static_assert(!MPG || VEC == 4, "routed pooled gradient: whole channel quads");
  const bool emit = !FIN && fin.o0 != nullptr;      // one-chunk case: see fin_emit
  const int n = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
  const int CVA = C / VEC;                    // gridDim.z channel slabs, as in in_moments_partial
  const int CV = CVA / gridDim.z;
  const int cvb = blockIdx.z * CV;
  const int TC = CV < TPB ? CV : TPB;
  const int rows = TPB / TC;
  const int tc = threadIdx.x % TC, trow = threadIdx.x / TC;
  const int p0 = chunk * pix_per_chunk;
  const int p1 = min(p0 + pix_per_chunk, HW);
  __shared__ float sm[TPB * 4 * 3];
  const size_t base = (size_t)n * HW * C;
  for (int cv0 = 0; cv0 < CV; cv0 += TC) {
    const int cv = cvb + cv0 + tc;
    const bool cv_ok = cv0 + tc < CV;
    float acc[3][VEC];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[q][j] = 0.f;
    // two-IN tail with both betas given: the activation mask is recomputed from y2 and s exactly as restail_fwd formed
    // the pre-activation (no read of `out`: 3 tensor reads instead of 4)
    constexpr bool remask = REMASK;      // host: t.ms && t.b2 && t.bs
    float m2[VEC], r2[VEC], ms[VEC], rs[VEC], g2[VEC], b2[VEC], gs[VEC], bs[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int c = cv * VEC + j, k = n * C + c;
      m2[j] = cv_ok ? t.m2[k] : 0.f; r2[j] = cv_ok ? t.r2[k] : 1.f;
      ms[j] = (cv_ok && t.ms) ? t.ms[k] : 0.f; rs[j] = (cv_ok && t.ms) ? t.rs[k] : 1.f;
      g2[j] = (cv_ok && remask) ? t.g2[c] : 1.f; b2[j] = (cv_ok && remask) ? t.b2[c] : 0.f;
      gs[j] = (cv_ok && remask) ? t.gs[c] : 1.f; bs[j] = (cv_ok && remask) ? t.bs[c] : 0.f;
    }
    if (trow < rows && cv_ok) {
      auto load = [&](int p, float* g, float* o, float* y, float* sv) {
        const size_t off = base + (size_t)p * C + cv * VEC;
        if constexpr (VEC == 4) {
          *(float4*)g = *(const float4*)(gout + off);
          if (!remask) *(float4*)o = *(const float4*)(out + off);
        } else {
          g[0] = gout[off];
          if (!remask) o[0] = out[off];
        }
        if constexpr (MPG) add_pooled_grad(mr, n, HW, C, p, cv, g);
        ld_act<VEC, HS>(t.y2, off, y);
        if (t.ms) ld_act<VEC, HS>(t.s, off, sv);
      };
      auto accum = [&](const float* g, const float* o, const float* y, const float* sv) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float pre = remask ? in_affine(y[j], m2[j], r2[j], g2[j], b2[j]) + in_affine(sv[j], ms[j], rs[j], gs[j], bs[j])
                                   : o[j];
          const float gz = g[j] * lrelu_mask(pre, slope);
          acc[0][j] += gz;
          acc[1][j] += gz * ((y[j] - m2[j]) * r2[j]);
          if (t.ms) acc[2][j] += gz * ((sv[j] - ms[j]) * rs[j]);
        }
      };
      int p = p0 + trow;
      for (; p + rows < p1; p += 2 * rows) {         // two rows (8 float4 loads) in flight, accumulated in row order
        float g[2][VEC], o[2][VEC], y[2][VEC], sv[2][VEC];
        load(p, g[0], o[0], y[0], sv[0]); load(p + rows, g[1], o[1], y[1], sv[1]);
        accum(g[0], o[0], y[0], sv[0]); accum(g[1], o[1], y[1], sv[1]);
      }
      for (; p < p1; p += rows) {
        float g[VEC], o[VEC], y[VEC], sv[VEC];
        load(p, g, o, y, sv);
        accum(g, o, y, sv);
      }
    }
    __syncthreads();
    if (TC <= 64 && (TC & (TC - 1)) == 0) {          // in-wave xor-tree + 4 wave totals (see in_moments_partial)
      for (int off = TC; off < 64; off <<= 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int j = 0; j < VEC; ++j) acc[q][j] += __shfl_xor(acc[q][j], off, 64);
      }
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      if (lane < TC) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int j = 0; j < VEC; ++j) sm[((wv * TC + lane) * 3 + q) * VEC + j] = acc[q][j];
      }
      __syncthreads();
      if (threadIdx.x < TC && cv_ok) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float tots[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            float tot = 0.f;
#pragma unroll
            for (int w4 = 0; w4 < 4; ++w4) tot += sm[((w4 * TC + tc) * 3 + q) * VEC + j];
            if constexpr (FIN) st_sc1_f(part + (((size_t)n * chunks + chunk) * C + cv * VEC + j) * 3 + q, tot);
            else part[(((size_t)n * chunks + chunk) * C + cv * VEC + j) * 3 + q] = tot;
            tots[q] = tot;
          }
          if (emit) fin_emit<2>(fin, HW, n * C + cv * VEC + j, tots);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < VEC; ++j) sm[(threadIdx.x * 3 + q) * VEC + j] = acc[q][j];
      __syncthreads();
      if (trow == 0 && cv_ok) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float tots[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            float tot = 0.f;
            for (int r = 0; r < rows; ++r) tot += sm[((r * TC + tc) * 3 + q) * VEC + j];
            if constexpr (FIN) st_sc1_f(part + (((size_t)n * chunks + chunk) * C + cv * VEC + j) * 3 + q, tot);
            else part[(((size_t)n * chunks + chunk) * C + cv * VEC + j) * 3 + q] = tot;
            tots[q] = tot;
          }
          if (emit) fin_emit<2>(fin, HW, n * C + cv * VEC + j, tots);
        }
      }
    }
    __syncthreads();
  }
  if constexpr (FIN) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every storing wave: its write-through partials have left
    __syncthreads();
    int* flag = reinterpret_cast<int*>(sm);
    if (threadIdx.x == 0) {
      const int tk = __hip_atomic_fetch_add(tickets + n, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *flag = (tk + 1 == (int)(gridDim.x * gridDim.z)) ? 1 : 0;
    }
    __syncthreads();
    const bool last = *flag != 0;
    __syncthreads();
    if (last) {                                              // (sm: 12 KB = 256 x 3 doubles + slack)
      fin_image3(part + (size_t)n * chunks * C * 3, chunks, C, HW, fin.o0 + (size_t)n * C, fin.o1 + (size_t)n * C,
                 fin.o2 + (size_t)n * C, reinterpret_cast<double*>(sm));
      if (threadIdx.x == 0) __hip_atomic_store(tickets + n, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int VEC, bool REMASK, bool HS = false, bool AMAX = false, bool MPG = false>
__global__ void __launch_bounds__(TPB)
restail_bwd_apply(const float* __restrict__ gout, const float* __restrict__ out, TailRef t, const float* __restrict__ am,
                  const float* __restrict__ b2m, const float* __restrict__ bsm, float* __restrict__ gy2,
                  float* __restrict__ gs, int HW, int C, float slope, int N, float* __restrict__ gg2,
                  float* __restrict__ gb2, float* __restrict__ ggs, float* __restrict__ gbs, float* __restrict__ amax = nullptr,
                  MaxRef mr = MaxRef{nullptr, nullptr, 0}) {
  static_assert(!MPG || VEC == 4, "routed pooled gradient: whole channel quads");
  constexpr bool remask = REMASK;             // see restail_bwd_partial
  float mx1 = 0.f, mx2 = 0.f;                 // amax: {max |gy2|, max |gs|}
  if (blockIdx.x == 0 && blockIdx.y == 0) {   // affine gradients of the tail: gg2 = sum_n M*b2, gb = sum_n M*a (both norms), ggs = sum_n M*bs
    for (int c = threadIdx.x; c < C; c += TPB) {
      double sa = 0.0, s2 = 0.0, ss = 0.0;
      for (int n = 0; n < N; ++n) { sa += (double)am[n * C + c]; s2 += (double)b2m[n * C + c]; if (ggs) ss += (double)bsm[n * C + c]; }
      gg2[c] = (float)(s2 * HW); gb2[c] = (float)(sa * HW);
      if (ggs) { ggs[c] = (float)(ss * HW); gbs[c] = (float)(sa * HW); }
    }
  }
  struct Prm { float av[VEC], b2v[VEC], bsv[VEC], m2[VEC], r2[VEC], g2[VEC], msv[VEC], rsv[VEC], gsv[VEC], be2[VEC], bes[VEC]; };
  img_walk(HW, C / VEC,
    [&](int n, int cv) {
      Prm p;
      const int c0 = cv * VEC, k0 = n * C + c0;
      ldv<VEC>(am, k0, p.av); ldv<VEC>(b2m, k0, p.b2v); ldv<VEC>(t.m2, k0, p.m2); ldv<VEC>(t.r2, k0, p.r2); ldv<VEC>(t.g2, c0, p.g2);
      if (t.ms) { ldv<VEC>(bsm, k0, p.bsv); ldv<VEC>(t.ms, k0, p.msv); ldv<VEC>(t.rs, k0, p.rsv); ldv<VEC>(t.gs, c0, p.gsv); }
      if (remask) { ldv<VEC>(t.b2, c0, p.be2); ldv<VEC>(t.bs, c0, p.bes); }
      return p;
    },
    [&](int64_t i, const Prm& p) {
      float g[VEC], o[VEC], y[VEC], sv[VEC], o1[VEC], o2[VEC];
      if constexpr (VEC == 4) {
        *(float4*)g = *(const float4*)(gout + i * 4);
        if (!remask) *(float4*)o = *(const float4*)(out + i * 4);
      } else {
        g[0] = gout[i];
        if (!remask) o[0] = out[i];
      }
      if constexpr (MPG) {
        const int CV = C / VEC;
        const int li = (int)(i - (int64_t)blockIdx.y * HW * CV);
        const int pix = li / CV;
        add_pooled_grad(mr, blockIdx.y, HW, C, pix, li - pix * CV, g);
      }
      ld_act<VEC, HS>(t.y2, (size_t)i * VEC, y);
      if (t.ms) ld_act<VEC, HS>(t.s, (size_t)i * VEC, sv);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float pre = remask ? in_affine(y[j], p.m2[j], p.r2[j], p.g2[j], p.be2[j]) + in_affine(sv[j], p.msv[j], p.rsv[j], p.gsv[j], p.bes[j])
                                 : o[j];
        const float gz = g[j] * lrelu_mask(pre, slope);
        const float a = p.av[j];
        o1[j] = p.g2[j] * p.r2[j] * (gz - a - ((y[j] - p.m2[j]) * p.r2[j]) * p.b2v[j]);
        o2[j] = t.ms ? p.gsv[j] * p.rsv[j] * (gz - a - ((sv[j] - p.msv[j]) * p.rsv[j]) * p.bsv[j]) : gz;
        if constexpr (AMAX) { mx1 = fmaxf(mx1, fabsf(o1[j])); mx2 = fmaxf(mx2, fabsf(o2[j])); }
      }
      if constexpr (VEC == 4) { *(float4*)(gy2 + i * 4) = *(float4*)o1; *(float4*)(gs + i * 4) = *(float4*)o2; }
      else { gy2[i] = o1[0]; gs[i] = o2[0]; }
    });
  if constexpr (AMAX) { amax_emit(mx1, amax); amax_emit(mx2, amax + gridDim.x * gridDim.y); }
}

inline bool fin_emit_on() {                         // SMSUT_IN_ONE_CHUNK=0: always launch in_moments_final (A/B switch)
  static const bool on = [] { const char* e = getenv("SMSUT_IN_ONE_CHUNK"); return !e || atoi(e) != 0; }();
  return on;
}
inline int pick_chunk(int HW, int C, int N) {
  // aim for >= ~1024 blocks overall while keeping >= 256 pixels per chunk
#ifndef SMSUT_IN_BLOCKS
#define SMSUT_IN_BLOCKS 1024
#endif
  // planes of <= 256 pixels are ONE chunk: there the partial-sum kernel finalises by itself (fin_emit) and the in_moments_final
  // launch is skipped.  (1024-pixel planes as one chunk measured no better than four chunks + the extra launch.)
  if (HW <= 256 && C % 4 == 0) return HW;
  int ppc = 2048;
  while (ppc > 256 && (int64_t)N * cdiv64(HW, ppc) < SMSUT_IN_BLOCKS) ppc >>= 1;
  (void)C;
  return ppc;
}

// channel slabs for the partial-sum kernels: until ~512 workgroups, slabs of >= 16 channels (64 contiguous bytes per pixel)
inline int slab_count(int N, int chunks, int C, int vec) {
  static const bool on = [] { const char* e = getenv("SMSUT_IN_SLABS"); return !e || atoi(e) != 0; }();
  if (!on || vec != 4) return 1;
  const int cv = C / 4;
  int z = 1;
  static const int target = [] { const char* e = getenv("SMSUT_IN_SLAB_WGS"); return e ? atoi(e) : 512; }();
  while ((int64_t)N * chunks * z < target && cv % (2 * z) == 0 && cv / (2 * z) >= 4) z *= 2;
  return z;
}

}  // namespace

extern "C" {

int smsut_in_chunks(int N, int HW, int C) { return (int)cdiv64(HW, pick_chunk(HW, C, N)); }

// workspace: float[N * smsut_in_chunks * C * 3]
int smsut_instnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       float* workspace, int N, int HW, int C, float eps, float slope, int has_act, void* stream) {
  SMSUT_REQUIRE(x && gamma && beta && y && mean && rstd && workspace && N > 0 && HW > 0 && C > 0);
  hipStream_t st = (hipStream_t)stream;
  const int ppc = pick_chunk(HW, C, N);
  const int chunks = (int)cdiv64(HW, ppc);
  dim3 g(chunks, N, slab_count(N, chunks, C, C % 4 == 0 ? 4 : 1));
  const FinOut fin = (chunks == 1 && fin_emit_on()) ? FinOut{mean, rstd, nullptr, eps} : FinOut{};
  if (C % 4 == 0)
    in_moments_partial<0, 4><<<g, TPB, 0, st>>>(x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, HW, C, ppc, slope, fin);
  else
    in_moments_partial<0, 1><<<g, TPB, 0, st>>>(x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, workspace, HW, C, ppc, slope, fin);
  if (!fin.o0) in_moments_final<0><<<dim3((C + 15) / 16, N), TPB, 0, st>>>(workspace, chunks, C, HW, eps, mean, rstd, nullptr);
  const int64_t total = (int64_t)N * HW * C;
  SMSUT_REQUIRE((int64_t)HW * C < (1ll << 31));       // per-image walks index in 32 bits
  if (C % 4 == 0)
    in_apply_fwd<4><<<img_grid((int64_t)HW * (C / 4), N), TPB, 0, st>>>(x, mean, rstd, gamma, beta, y, HW, C, slope, has_act);
  else
    in_apply_fwd<1><<<img_grid((int64_t)HW * C, N), TPB, 0, st>>>(x, mean, rstd, gamma, beta, y, HW, C, slope, has_act);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

// Same as smsut_instnorm_fwd when the {sum, sum^2} partials [N][chunks][C][2] were already produced by the conv
// epilogue (smsut_conv2d_fwd_mfma_stats): finalise + normalise/activate only.
static int instnorm_fwd_partials_launch(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                        float* rstd, const float* partials, int chunks, int N, int HW, int C, float eps,
                                        float slope, int has_act, void* stream, int hs) {
  SMSUT_REQUIRE(x && gamma && beta && y && mean && rstd && partials && chunks > 0 && N > 0 && HW > 0 && C > 0);
  hipStream_t st = (hipStream_t)stream;
  in_moments_final<0><<<dim3((C + 15) / 16, N), TPB, 0, st>>>(partials, chunks, C, HW, eps, mean, rstd, nullptr);
  const int64_t total = (int64_t)N * HW * C;
  SMSUT_REQUIRE((int64_t)HW * C < (1ll << 31));       // per-image walks index in 32 bits
  if (hs == 2)
    in_apply_fwd<4, true, true><<<img_grid((int64_t)HW * (C / 4), N), TPB, 0, st>>>(x, mean, rstd, gamma, beta, y, HW, C, slope, has_act);
  else if (hs)
    in_apply_fwd<4, true><<<img_grid((int64_t)HW * (C / 4), N), TPB, 0, st>>>(x, mean, rstd, gamma, beta, y, HW, C, slope, has_act);
  else if (C % 4 == 0)
    in_apply_fwd<4><<<img_grid((int64_t)HW * (C / 4), N), TPB, 0, st>>>(x, mean, rstd, gamma, beta, y, HW, C, slope, has_act);
  else
    in_apply_fwd<1><<<img_grid((int64_t)HW * C, N), TPB, 0, st>>>(x, mean, rstd, gamma, beta, y, HW, C, slope, has_act);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}
int smsut_instnorm_fwd_partials(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                float* rstd, const float* partials, int chunks, int N, int HW, int C, float eps,
                                float slope, int has_act, void* stream) {
  return instnorm_fwd_partials_launch(x, gamma, beta, y, mean, rstd, partials, chunks, N, HW, C, eps, slope, has_act, stream, 0);
}
// "half storage" (config 5): x is fp16 [N,HW,C] (the raw conv output a conv epilogue stored), y stays fp32; C % 4 == 0
int smsut_instnorm_fwd_partials_hs(const void* x16, const float* gamma, const float* beta, float* y, float* mean,
                                   float* rstd, const float* partials, int chunks, int N, int HW, int C, float eps,
                                   float slope, int has_act, void* stream) {
  SMSUT_REQUIRE(C % 4 == 0);
  return instnorm_fwd_partials_launch((const float*)x16, gamma, beta, y, mean, rstd, partials, chunks, N, HW, C, eps, slope, has_act,
                                      stream, 1);
}
// ... and y stored as fp16 too (read by smsut_conv2d_fwd_mfma_stats_f16_hsx / smsut_conv2d_wgrad_f16_xh, which round to fp16 anyway)
int smsut_instnorm_fwd_partials_hs2(const void* x16, const float* gamma, const float* beta, void* y16, float* mean,
                                    float* rstd, const float* partials, int chunks, int N, int HW, int C, float eps,
                                    float slope, int has_act, void* stream) {
  SMSUT_REQUIRE(C % 4 == 0);
  return instnorm_fwd_partials_launch((const float*)x16, gamma, beta, (float*)y16, mean, rstd, partials, chunks, N, HW, C, eps, slope,
                                      has_act, stream, 2);
}

// beta == null: no activation; otherwise the LeakyReLU mask is recomputed from x (sign of the normalised
// pre-activation, bit-identical to the forward).  Outputs gx [N,HW,C], a/b [N,C] (saved for bwd2), ggamma/gbeta [C] (may be null).
int smsut_instnorm_bwd(const float* gy, const float* x, const float* beta, const float* mean, const float* rstd,
                       const float* gamma, float* gx, float* a_mean, float* b_mean, float* ggamma, float* gbeta,
                       float* workspace, int N, int HW, int C, float slope, void* stream) {
  SMSUT_REQUIRE(gy && x && mean && rstd && gamma && gx && a_mean && b_mean && workspace && N > 0 && HW > 0 && C > 0);
  hipStream_t st = (hipStream_t)stream;
  const int ppc = pick_chunk(HW, C, N);
  const int chunks = (int)cdiv64(HW, ppc);
  dim3 g(chunks, N, slab_count(N, chunks, C, C % 4 == 0 ? 4 : 1));
  const FinOut fin = (chunks == 1 && fin_emit_on()) ? FinOut{a_mean, b_mean, nullptr, 0.f} : FinOut{};
  if (C % 4 == 0)
    in_moments_partial<1, 4><<<g, TPB, 0, st>>>(gy, x, nullptr, gamma, beta, mean, rstd, workspace, HW, C, ppc, slope, fin);
  else
    in_moments_partial<1, 1><<<g, TPB, 0, st>>>(gy, x, nullptr, gamma, beta, mean, rstd, workspace, HW, C, ppc, slope, fin);
  if (!fin.o0) in_moments_final<1><<<dim3((C + 15) / 16, N), TPB, 0, st>>>(workspace, chunks, C, HW, 0.f, a_mean, b_mean, nullptr);
  float* gg = (ggamma && gbeta) ? ggamma : nullptr;       // affine gradients: computed by block 0 of the apply kernel
  const int64_t total = (int64_t)N * HW * C;
  SMSUT_REQUIRE((int64_t)HW * C < (1ll << 31));       // per-image walks index in 32 bits
  if (C % 4 == 0)
    in_apply_bwd<4><<<img_grid((int64_t)HW * (C / 4), N), TPB, 0, st>>>(gy, x, beta, mean, rstd, gamma, a_mean, b_mean, gx, HW, C, slope,
                                                        N, gg, gbeta);
  else
    in_apply_bwd<1><<<img_grid((int64_t)HW * C, N), TPB, 0, st>>>(gy, x, beta, mean, rstd, gamma, a_mean, b_mean, gx, HW, C, slope,
                                                    N, gg, gbeta);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

// InstanceNorm + LeakyReLU + AvgPool2d(2) as one op (r05; bn1 -> relu -> avgpool of a stride-2 BottleBlock, reference
// network/blocks.py:99-107; passes differentiated once).  Forward from the conv epilogue's statistics partials (as
// smsut_instnorm_fwd_partials): x [N,H,W,C] raw conv output -> y [N,H/2,W/2,C]; mean / rstd [N,C] are outputs.  H, W even.
int smsut_instnorm_pool_fwd_partials(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                     const float* partials, int chunks, int N, int H, int W, int C, float eps, float slope,
                                     void* stream) {
  SMSUT_REQUIRE(x && gamma && beta && y && mean && rstd && partials && chunks > 0 && N > 0 && H > 0 && W > 0 && C > 0 && !(H & 1) &&
                !(W & 1) && (int64_t)H * W * C < (1ll << 31));
  hipStream_t st = (hipStream_t)stream;
  in_moments_final<0><<<dim3((C + 15) / 16, N), TPB, 0, st>>>(partials, chunks, C, H * W, eps, mean, rstd, nullptr);
  const int64_t units = (int64_t)(H / 2) * (W / 2);
  if (C % 4 == 0) in_apply_pool_fwd<4><<<img_grid(units * (C / 4), N), TPB, 0, st>>>(x, mean, rstd, gamma, beta, y, H, W, C, slope);
  else in_apply_pool_fwd<1><<<img_grid(units * C, N), TPB, 0, st>>>(x, mean, rstd, gamma, beta, y, H, W, C, slope);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}
// ... its backward: gyp [N,H/2,W/2,C] is the gradient of the POOLED output; everything else as smsut_instnorm_bwd (beta non-null:
// the mask is recomputed from x).  The full-resolution gradient 0.25 * gyp[h/2][w/2] is formed while loading, in both passes --
// bit-identical to smsut_avgpool2_bwd followed by smsut_instnorm_bwd, without the full-resolution gradient tensor.
int smsut_instnorm_pool_bwd(const float* gyp, const float* x, const float* beta, const float* mean, const float* rstd,
                            const float* gamma, float* gx, float* a_mean, float* b_mean, float* ggamma, float* gbeta,
                            float* workspace, int N, int H, int W, int C, float slope, void* stream) {
  SMSUT_REQUIRE(gyp && x && beta && mean && rstd && gamma && gx && a_mean && b_mean && workspace && N > 0 && H > 0 && W > 0 && C > 0 &&
                !(H & 1) && !(W & 1));
  hipStream_t st = (hipStream_t)stream;
  const int HW = H * W;
  const int ppc = pick_chunk(HW, C, N);
  const int chunks = (int)cdiv64(HW, ppc);
  dim3 g(chunks, N, slab_count(N, chunks, C, C % 4 == 0 ? 4 : 1));
  const FinOut fin = (chunks == 1 && fin_emit_on()) ? FinOut{a_mean, b_mean, nullptr, 0.f} : FinOut{};
  if (C % 4 == 0)
    in_moments_partial<1, 4, true><<<g, TPB, 0, st>>>(gyp, x, nullptr, gamma, beta, mean, rstd, workspace, HW, C, ppc, slope, fin, W);
  else
    in_moments_partial<1, 1, true><<<g, TPB, 0, st>>>(gyp, x, nullptr, gamma, beta, mean, rstd, workspace, HW, C, ppc, slope, fin, W);
  if (!fin.o0) in_moments_final<1><<<dim3((C + 15) / 16, N), TPB, 0, st>>>(workspace, chunks, C, HW, 0.f, a_mean, b_mean, nullptr);
  float* gg = (ggamma && gbeta) ? ggamma : nullptr;
  SMSUT_REQUIRE((int64_t)HW * C < (1ll << 31));       // per-image walks index in 32 bits
  if (C % 4 == 0)
    in_apply_bwd<4, false, false, true><<<img_grid((int64_t)HW * (C / 4), N), TPB, 0, st>>>(gyp, x, beta, mean, rstd, gamma, a_mean, b_mean,
                                                                                         gx, HW, C, slope, N, gg, gbeta, nullptr, W);
  else
    in_apply_bwd<1, false, false, true><<<img_grid((int64_t)HW * C, N), TPB, 0, st>>>(gyp, x, beta, mean, rstd, gamma, a_mean, b_mean, gx,
                                                                                   HW, C, slope, N, gg, gbeta, nullptr, W);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

// Backward of smsut_instnorm_bwd (WGAN-GP double backward).  v = d/dgx; ug/ub = d/dggamma, d/dgbeta (may be null).
// scratch: float[3*N*C] for cv/dv/e.
int smsut_instnorm_bwd2(const float* v, const float* ug, const float* ub, const float* gy, const float* x,
                        const float* beta, const float* mean, const float* rstd, const float* gamma,
                        const float* a_mean, const float* b_mean, float* d_gy, float* d_x, float* d_gamma,
                        float* workspace, float* scratch, int N, int HW, int C, float slope, void* stream) {
  SMSUT_REQUIRE(v && gy && x && mean && rstd && gamma && a_mean && b_mean && d_gy && d_x && d_gamma && workspace &&
                scratch && N > 0 && HW > 0 && C > 0);
  hipStream_t st = (hipStream_t)stream;
  const int ppc = pick_chunk(HW, C, N);
  const int chunks = (int)cdiv64(HW, ppc);
  dim3 g(chunks, N, slab_count(N, chunks, C, C % 4 == 0 ? 4 : 1));
  float* cvm = scratch; float* dvm = scratch + (size_t)N * C; float* em = scratch + 2 * (size_t)N * C;
  const FinOut fin = (chunks == 1 && fin_emit_on()) ? FinOut{cvm, dvm, em, 0.f} : FinOut{};
  if (C % 4 == 0)
    in_moments_partial<2, 4><<<g, TPB, 0, st>>>(v, x, gy, gamma, beta, mean, rstd, workspace, HW, C, ppc, slope, fin);
  else
    in_moments_partial<2, 1><<<g, TPB, 0, st>>>(v, x, gy, gamma, beta, mean, rstd, workspace, HW, C, ppc, slope, fin);
  if (!fin.o0) in_moments_final<2><<<dim3((C + 15) / 16, N), TPB, 0, st>>>(workspace, chunks, C, HW, 0.f, cvm, dvm, em);
  in_bwd2_gamma<<<(C + 63) / 64, 64, 0, st>>>(rstd, a_mean, b_mean, cvm, dvm, em, N, C, HW, d_gamma);
  const int64_t total = (int64_t)N * HW * C;
  SMSUT_REQUIRE((int64_t)HW * C < (1ll << 31));       // per-image walks index in 32 bits
  if (C % 4 == 0)
    in_apply_bwd2<4><<<ew_grid(total / 4), TPB, 0, st>>>(v, x, gy, beta, mean, rstd, gamma, a_mean, b_mean, cvm, dvm, em,
                                                          ug, ub, d_gy, d_x, total / 4, HW, C, slope);
  else
    in_apply_bwd2<1><<<ew_grid(total), TPB, 0, st>>>(v, x, gy, beta, mean, rstd, gamma, a_mean, b_mean, cvm, dvm, em,
                                                      ug, ub, d_gy, d_x, total, HW, C, slope);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

// Finalise forward partials [N][chunks][C][2] = {sum, sum^2} (from smsut_conv2d_fwd_mfma_fused) into mean / rstd.
int smsut_in_finalize_fwd(const float* partials, int chunks, float* mean, float* rstd, int N, int HW, int C, float eps,
                          void* stream) {
  SMSUT_REQUIRE(partials && mean && rstd && chunks > 0 && N > 0 && HW > 0 && C > 0);
  in_moments_final<0><<<dim3((C + 15) / 16, N), TPB, 0, (hipStream_t)stream>>>(partials, chunks, C, HW, eps, mean, rstd, nullptr);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

// smsut_in_finalize_fwd for TWO partial sets of the same (N, HW, C) in one launch (chunk counts may differ)
int smsut_in_finalize_fwd2(const float* pa, int chunks_a, float* mean_a, float* rstd_a, const float* pb, int chunks_b, float* mean_b,
                           float* rstd_b, int N, int HW, int C, float eps, void* stream) {
  SMSUT_REQUIRE(pa && pb && mean_a && rstd_a && mean_b && rstd_b && chunks_a > 0 && chunks_b > 0 && N > 0 && HW > 0 && C > 0);
  in_moments_final_pair<<<dim3((C + 15) / 16, N, 2), TPB, 0, (hipStream_t)stream>>>(pa, chunks_a, mean_a, rstd_a, pb, chunks_b, mean_b,
                                                                                   rstd_b, C, HW, eps);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

// Finalise backward partials [N][chunks][C][2] = {sum gz, sum gz*xhat} (from smsut_conv2d_dgrad_mfma_bwdstats) into the
// per-(n,c) means a, b.
int smsut_in_finalize_bwd(const float* partials, int chunks, float* a_mean, float* b_mean, int N, int HW, int C,
                          void* stream) {
  SMSUT_REQUIRE(partials && a_mean && b_mean && chunks > 0 && N > 0 && HW > 0 && C > 0);
  in_moments_final<1><<<dim3((C + 15) / 16, N), TPB, 0, (hipStream_t)stream>>>(partials, chunks, C, HW, 0.f, a_mean, b_mean, nullptr);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

// gx = gamma*rstd*(gz - a - xhat*b) with gz ALREADY masked (output of smsut_conv2d_dgrad_mfma_bwdstats); ggamma / gbeta
// (nullable) are the affine gradients sum_n HW*b, sum_n HW*a.
static int in_apply_bwd_launch(const float* gz, const float* x, const float* mean, const float* rstd, const float* gamma,
                               const float* a_mean, const float* b_mean, float* gx, float* ggamma, float* gbeta, float* amax, int N,
                               int HW, int C, void* stream, bool hs = false) {
  SMSUT_REQUIRE(gz && x && mean && rstd && gamma && a_mean && b_mean && gx && N > 0 && HW > 0 && C > 0);
  hipStream_t st = (hipStream_t)stream;
  float* gg = (ggamma && gbeta) ? ggamma : nullptr;
  const int64_t total = (int64_t)N * HW * C;
  SMSUT_REQUIRE((int64_t)HW * C < (1ll << 31));       // per-image walks index in 32 bits
#define IN_APPLY_BWD(V, HSF, AM)                                                                                           \
  in_apply_bwd<V, HSF, AM><<<img_grid((int64_t)HW * (C / V), N), TPB, 0, st>>>(gz, x, nullptr, mean, rstd, gamma, a_mean, b_mean, gx, \
                                                                               HW, C, 0.f, N, gg, gbeta, amax)
  if (hs) { if (amax) IN_APPLY_BWD(4, true, true); else IN_APPLY_BWD(4, true, false); }
  else if (C % 4 == 0) { if (amax) IN_APPLY_BWD(4, false, true); else IN_APPLY_BWD(4, false, false); }
  else { if (amax) IN_APPLY_BWD(1, false, true); else IN_APPLY_BWD(1, false, false); }
#undef IN_APPLY_BWD
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}
int smsut_in_apply_bwd(const float* gz, const float* x, const float* mean, const float* rstd, const float* gamma,
                       const float* a_mean, const float* b_mean, float* gx, float* ggamma, float* gbeta, int N, int HW, int C,
                       void* stream) {
  return in_apply_bwd_launch(gz, x, mean, rstd, gamma, a_mean, b_mean, gx, ggamma, gbeta, nullptr, N, HW, C, stream);
}
// workgroups of the per-image apply kernels (= amax slots per output tensor of the _amax entry points)
int smsut_amax_blocks(int N, int HW, int C) {
  if (N <= 0 || HW <= 0 || C <= 0) return 0;
  const dim3 g = img_grid((int64_t)HW * (C % 4 == 0 ? C / 4 : C), N);
  return (int)(g.x * g.y);
}
// ... that also hands over max |gx| (amax: smsut_amax_blocks floats, one per workgroup, see amax_emit) for smsut_absmax_finish
int smsut_in_apply_bwd_amax(const float* gz, const float* x, const float* mean, const float* rstd, const float* gamma,
                            const float* a_mean, const float* b_mean, float* gx, float* ggamma, float* gbeta, float* amax, int N,
                            int HW, int C, void* stream) {
  SMSUT_REQUIRE(amax);
  return in_apply_bwd_launch(gz, x, mean, rstd, gamma, a_mean, b_mean, gx, ggamma, gbeta, amax, N, HW, C, stream);
}
// "half storage" (config 5): x is fp16 [N,HW,C]; amax nullable; C % 4 == 0
int smsut_in_apply_bwd_hs(const float* gz, const void* x16, const float* mean, const float* rstd, const float* gamma,
                          const float* a_mean, const float* b_mean, float* gx, float* ggamma, float* gbeta, float* amax, int N,
                          int HW, int C, void* stream) {
  SMSUT_REQUIRE(C % 4 == 0);
  return in_apply_bwd_launch(gz, (const float*)x16, mean, rstd, gamma, a_mean, b_mean, gx, ggamma, gbeta, amax, N, HW, C, stream, true);
}

// out = act(IN(y2) + (IN(s) | s)); ms == null: s is added as it is (identity shortcut).
static int restail_fwd_launch(const float* y2, const float* m2, const float* r2, const float* g2, const float* b2, const float* s,
                              const float* ms, const float* rs, const float* gs, const float* bs, float* out, bool hs, int N, int HW,
                              int C, float slope, void* stream) {
  SMSUT_REQUIRE(y2 && m2 && r2 && g2 && b2 && s && out && N > 0 && HW > 0 && C > 0 && (!ms || (rs && gs && bs)));
  TailRef t{y2, m2, r2, g2, b2, s, ms, rs, gs, bs};
  const int64_t total = (int64_t)N * HW * C;
  SMSUT_REQUIRE((int64_t)HW * C < (1ll << 31));       // per-image walks index in 32 bits
  hipStream_t st = (hipStream_t)stream;
  if (hs) restail_fwd<4, true><<<img_grid((int64_t)HW * (C / 4), N), TPB, 0, st>>>(t, out, HW, C, slope);
  else if (C % 4 == 0) restail_fwd<4><<<img_grid((int64_t)HW * (C / 4), N), TPB, 0, st>>>(t, out, HW, C, slope);
  else restail_fwd<1><<<img_grid((int64_t)HW * C, N), TPB, 0, st>>>(t, out, HW, C, slope);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}
int smsut_restail_fwd(const float* y2, const float* m2, const float* r2, const float* g2, const float* b2, const float* s,
                      const float* ms, const float* rs, const float* gs, const float* bs, float* out, int N, int HW, int C,
                      float slope, void* stream) {
  return restail_fwd_launch(y2, m2, r2, g2, b2, s, ms, rs, gs, bs, out, false, N, HW, C, slope, stream);
}
// "half storage" (config 5): y2 and s are fp16 [N,HW,C] (conv shortcut: ms != null), C % 4 == 0; everything else as above
int smsut_restail_fwd_hs(const void* y2, const float* m2, const float* r2, const float* g2, const float* b2, const void* s,
                         const float* ms, const float* rs, const float* gs, const float* bs, float* out, int N, int HW, int C,
                         float slope, void* stream) {
  SMSUT_REQUIRE(ms && C % 4 == 0);
  return restail_fwd_launch((const float*)y2, m2, r2, g2, b2, (const float*)s, ms, rs, gs, bs, out, true, N, HW, C, slope, stream);
}

// Backward of the tail.  workspace: float[N * smsut_in_chunks(N,HW,C) * C * 3]; a/b2/bs: [N,C] scratch outputs;
// b2 / bs (the two IN betas, nullable): with ms and both betas the activation mask is recomputed from y2 and s and `out`
// is not read (8 tensor passes instead of 10); otherwise the mask is the sign of `out`.
// gy2, gs: gradients w.r.t. the two raw conv outputs (gs = gradient of the identity when ms == null);
// gg2/gb2/ggs/gbs: affine gradients (ggs/gbs nullable when ms == null).
static int restail_bwd_launch(const float* gout, const float* out, const float* y2, const float* m2, const float* r2,
                              const float* g2, const float* b2, const float* s, const float* ms, const float* rs,
                              const float* gs_, const float* bs, float* gy2, float* gs, float* a_mean, float* b2_mean, float* bs_mean, float* gg2, float* gb2, float* ggs,
                              float* gbs, float* workspace, float* amax, int N, int HW, int C, float slope, void* stream,
                              bool hs = false, int* tickets = nullptr, const MaxRef* mr = nullptr) {
  SMSUT_REQUIRE(gout && out && y2 && m2 && r2 && g2 && s && gy2 && gs && a_mean && b2_mean && bs_mean && gg2 && gb2 &&
                workspace && N > 0 && HW > 0 && C > 0 && (!ms || (rs && gs_ && ggs && gbs)));
  // mr (r05): gout is the skip connection's gradient, the pooled path's is routed in while loading (two-IN tail, channel quads)
  SMSUT_REQUIRE(!mr || (mr->gp && mr->W > 0 && HW % mr->W == 0 && ms && b2 && bs && C % 4 == 0));
  const MaxRef mrv = mr ? *mr : MaxRef{nullptr, nullptr, 0};
  TailRef t{y2, m2, r2, g2, b2, s, ms, rs, gs_, bs};
  hipStream_t st = (hipStream_t)stream;
  const int ppc = pick_chunk(HW, C, N);
  const int chunks = (int)cdiv64(HW, ppc);
  dim3 g(chunks, N, slab_count(N, chunks, C, C % 4 == 0 ? 4 : 1));
  const bool remask = ms && b2 && bs;
  const FinOut fin = (chunks == 1 && fin_emit_on()) ? FinOut{a_mean, b2_mean, bs_mean, 0.f} : FinOut{};
#define TAIL_PARTIAL(V, R) restail_bwd_partial<V, R><<<g, TPB, 0, st>>>(gout, out, t, workspace, HW, C, ppc, slope, fin)
  // in-launch finalize (several chunks per image, fp32, the two-IN tail on whole float4 channel groups): the last-arriving
  // workgroup of an image combines its partials -- no in_moments_final<2> launch
  const bool fin_in = tickets && !fin.o0 && !hs && remask && C % 4 == 0;
  if (fin_in && mr) {
    restail_bwd_partial<4, true, false, true, true><<<g, TPB, 0, st>>>(gout, out, t, workspace, HW, C, ppc, slope,
                                                                       FinOut{a_mean, b2_mean, bs_mean, 0.f}, tickets, mrv);
  } else if (fin_in) {
    restail_bwd_partial<4, true, false, true><<<g, TPB, 0, st>>>(gout, out, t, workspace, HW, C, ppc, slope,
                                                                 FinOut{a_mean, b2_mean, bs_mean, 0.f}, tickets);
  } else if (hs) {
    SMSUT_REQUIRE(remask && C % 4 == 0);
    if (mr) restail_bwd_partial<4, true, true, false, true><<<g, TPB, 0, st>>>(gout, out, t, workspace, HW, C, ppc, slope, fin, nullptr, mrv);
    else restail_bwd_partial<4, true, true><<<g, TPB, 0, st>>>(gout, out, t, workspace, HW, C, ppc, slope, fin);
  } else if (mr) {
    restail_bwd_partial<4, true, false, false, true><<<g, TPB, 0, st>>>(gout, out, t, workspace, HW, C, ppc, slope, fin, nullptr, mrv);
  } else if (C % 4 == 0) { if (remask) TAIL_PARTIAL(4, true); else TAIL_PARTIAL(4, false); }
  else { if (remask) TAIL_PARTIAL(1, true); else TAIL_PARTIAL(1, false); }
#undef TAIL_PARTIAL
  if (!fin.o0 && !fin_in) in_moments_final<2><<<dim3((C + 15) / 16, N), TPB, 0, st>>>(workspace, chunks, C, HW, 0.f, a_mean, b2_mean, bs_mean);
  // the affine gradients (and the copy gbs = gb2) are written by block 0 of the apply kernel
  const int64_t total = (int64_t)N * HW * C;
  SMSUT_REQUIRE((int64_t)HW * C < (1ll << 31));       // per-image walks index in 32 bits
#define TAIL_APPLY(V, R, HSF, AM)                                                                                       \
  restail_bwd_apply<V, R, HSF, AM><<<img_grid((int64_t)HW * (C / V), N), TPB, 0, st>>>(gout, out, t, a_mean, b2_mean, bs_mean, gy2, gs, \
                                                                               HW, C, slope, N, gg2, gb2, ms ? ggs : nullptr,  \
                                                                               ms ? gbs : nullptr, amax)
#define TAIL_APPLY_AM(V, R, HSF) do { if (amax) TAIL_APPLY(V, R, HSF, true); else TAIL_APPLY(V, R, HSF, false); } while (0)
#define TAIL_APPLY_MP(HSF, AM)                                                                                              \
  restail_bwd_apply<4, true, HSF, AM, true><<<img_grid((int64_t)HW * (C / 4), N), TPB, 0, st>>>(gout, out, t, a_mean, b2_mean, bs_mean, gy2, \
                                                                                             gs, HW, C, slope, N, gg2, gb2, ggs, gbs, amax, mrv)
  if (mr) {
    if (hs) { if (amax) TAIL_APPLY_MP(true, true); else TAIL_APPLY_MP(true, false); }
    else { if (amax) TAIL_APPLY_MP(false, true); else TAIL_APPLY_MP(false, false); }
  } else
  if (hs) TAIL_APPLY_AM(4, true, true);
  else if (C % 4 == 0) { if (remask) TAIL_APPLY_AM(4, true, false); else TAIL_APPLY_AM(4, false, false); }
  else { if (remask) TAIL_APPLY_AM(1, true, false); else TAIL_APPLY_AM(1, false, false); }
#undef TAIL_APPLY_MP
#undef TAIL_APPLY_AM
#undef TAIL_APPLY
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}
// The tail of an encoder level's block + the level's MaxPool2d(2, 2) (r05, reference network/blocks.py:74-79 + 131-133 / network/ugan.py:36-39).
// Forward: out [N,H,W,C] (the skip connection), pooled [N,H/2,W/2,C] and idx [N,H/2,W/2,C] bytes (position of the maximum in its
// window) in one pass -- bit-identical to smsut_restail_fwd + smsut_maxpool2_fwd.  hs: y2 / s are fp16 (smsut_restail_fwd_hs).
// Conv shortcut (ms != null), C % 4 == 0, H and W even.  idx == null (both calls): AVERAGE pooling instead -- a stride-2 BottleBlock of the
// discriminator feeding the next one (blocks.py:83-117: its output goes to conv1 and to F.avg_pool2d): smsut_avgpool2_fwd's arithmetic
// forward; backward every window pixel gets gout + 0.25 gp, i.e. smsut_avgpool2_bwd + autograd's accumulation, bit for bit.
int smsut_restail_fwd_pool(const void* y2, const float* m2, const float* r2, const float* g2, const float* b2, const void* s,
                           const float* ms, const float* rs, const float* gs, const float* bs, float* out, float* pooled,
                           void* idx, int N, int H, int W, int C, float slope, int hs, void* stream) {
  SMSUT_REQUIRE(y2 && m2 && r2 && g2 && b2 && s && ms && rs && gs && bs && out && pooled && N > 0 && H > 0 && W > 0 && C > 0 &&
                C % 4 == 0 && !(H & 1) && !(W & 1) && (int64_t)H * W * C < (1ll << 31));
  TailRef t{(const float*)y2, m2, r2, g2, b2, (const float*)s, ms, rs, gs, bs};
  hipStream_t st = (hipStream_t)stream;
  const dim3 g = img_grid((int64_t)(H / 2) * (W / 2) * (C / 4), N);
  if (hs) restail_fwd_pool<true><<<g, TPB, 0, st>>>(t, out, pooled, (unsigned int*)idx, H, W, C, slope);
  else restail_fwd_pool<false><<<g, TPB, 0, st>>>(t, out, pooled, (unsigned int*)idx, H, W, C, slope);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}
// Backward of that pair: gout = gradient of `out` through the skip connection, gp [N,H/2,W/2,C] = gradient of `pooled`; the block
// output's total gradient (what smsut_maxpool2_bwd_add would write) is formed while loading, in both passes of the tail backward --
// no pooling-backward pass, no full-resolution gradient tensor.  tickets (nullable): in-launch finalize as smsut_restail_bwd_fin;
// amax (nullable): as smsut_restail_bwd_amax; hs: y2 / s fp16 as smsut_restail_bwd_hs.  Results bit-identical to
// smsut_maxpool2_bwd_add followed by the corresponding smsut_restail_bwd* call.
int smsut_restail_bwd_pool(const float* gout, const float* gp, const void* idx, const void* y2, const float* m2, const float* r2,
                           const float* g2, const float* b2, const void* s, const float* ms, const float* rs, const float* gs_,
                           const float* bs, float* gy2, float* gs, float* a_mean, float* b2_mean, float* bs_mean, float* gg2,
                           float* gb2, float* ggs, float* gbs, float* workspace, int* tickets, float* amax, int N, int H, int W, int C,
                           float slope, int hs, void* stream) {
  SMSUT_REQUIRE(gp && H > 0 && W > 0 && !(H & 1) && !(W & 1));
  const MaxRef mr{gp, (const unsigned int*)idx, W};
  return restail_bwd_launch(gout, gout, (const float*)y2, m2, r2, g2, b2, (const float*)s, ms, rs, gs_, bs, gy2, gs, a_mean, b2_mean,
                            bs_mean, gg2, gb2, ggs, gbs, workspace, amax, N, H * W, C, slope, stream, hs != 0, tickets, &mr);
}
int smsut_restail_bwd(const float* gout, const float* out, const float* y2, const float* m2, const float* r2,
                      const float* g2, const float* b2, const float* s, const float* ms, const float* rs,
                      const float* gs_, const float* bs, float* gy2, float* gs, float* a_mean, float* b2_mean, float* bs_mean, float* gg2, float* gb2, float* ggs,
                      float* gbs, float* workspace, int N, int HW, int C, float slope, void* stream) {
  return restail_bwd_launch(gout, out, y2, m2, r2, g2, b2, s, ms, rs, gs_, bs, gy2, gs, a_mean, b2_mean, bs_mean, gg2, gb2, ggs, gbs,
                            workspace, nullptr, N, HW, C, slope, stream);
}
// ... with the per-image means finalised INSIDE the partial-sum launch (common.h: write-through partials, agent-scope ticket per
// image, the last-arriving workgroup combines in in_moments_final<2>'s order: same bits) -- two launches instead of three.
// tickets: int [N], zero on entry, zero again on exit.  Falls back to the three-launch form where the fused form does not apply
// (one-chunk planes finalise in the partial kernel anyway; identity shortcut; C % 4 != 0).
int smsut_restail_bwd_fin(const float* gout, const float* out, const float* y2, const float* m2, const float* r2,
                          const float* g2, const float* b2, const float* s, const float* ms, const float* rs,
                          const float* gs_, const float* bs, float* gy2, float* gs, float* a_mean, float* b2_mean, float* bs_mean, float* gg2, float* gb2, float* ggs,
                          float* gbs, float* workspace, int* tickets, int N, int HW, int C, float slope, void* stream) {
  SMSUT_REQUIRE(tickets);
  return restail_bwd_launch(gout, out, y2, m2, r2, g2, b2, s, ms, rs, gs_, bs, gy2, gs, a_mean, b2_mean, bs_mean, gg2, gb2, ggs, gbs,
                            workspace, nullptr, N, HW, C, slope, stream, false, tickets);
}
// ... that also hands over max |gy2| (amax[0 .. B)) and max |gs| (amax[B .. 2B)), B = smsut_amax_blocks, for smsut_absmax_finish
int smsut_restail_bwd_amax(const float* gout, const float* out, const float* y2, const float* m2, const float* r2,
                           const float* g2, const float* b2, const float* s, const float* ms, const float* rs,
                           const float* gs_, const float* bs, float* gy2, float* gs, float* a_mean, float* b2_mean, float* bs_mean, float* gg2, float* gb2, float* ggs,
                           float* gbs, float* workspace, float* amax, int N, int HW, int C, float slope, void* stream) {
  SMSUT_REQUIRE(amax);
  return restail_bwd_launch(gout, out, y2, m2, r2, g2, b2, s, ms, rs, gs_, bs, gy2, gs, a_mean, b2_mean, bs_mean, gg2, gb2, ggs, gbs,
                            workspace, amax, N, HW, C, slope, stream);
}
// "half storage" (config 5): y2 and s are fp16 [N,HW,C]; two-IN tail with both betas (the mask is recomputed, `out` is not read),
// C % 4 == 0; amax nullable (see smsut_restail_bwd_amax)
int smsut_restail_bwd_hs(const float* gout, const float* out, const void* y2, const float* m2, const float* r2,
                         const float* g2, const float* b2, const void* s, const float* ms, const float* rs,
                         const float* gs_, const float* bs, float* gy2, float* gs, float* a_mean, float* b2_mean, float* bs_mean, float* gg2, float* gb2, float* ggs,
                         float* gbs, float* workspace, float* amax, int N, int HW, int C, float slope, void* stream) {
  SMSUT_REQUIRE(ms && b2 && bs && C % 4 == 0);
  return restail_bwd_launch(gout, out, (const float*)y2, m2, r2, g2, b2, (const float*)s, ms, rs, gs_, bs, gy2, gs, a_mean, b2_mean,
                            bs_mean, gg2, gb2, ggs, gbs, workspace, amax, N, HW, C, slope, stream, true);
}

}  // extern "C"
