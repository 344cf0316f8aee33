// Convolutions with a tiny channel count on one side: the 5x5 stems (Cin 1 / 1+n_modal -> base_width/2,
// network/blocks.py:123, ugan.py:26), D's 4x4 stride-2 stem (ugan.py:202) and the 1x1 heads (16 -> 5 / 1,
// blocks.py:166, ugan.py:70).  They carry < 1 % of the FLOPs but touch the full-resolution tensors, so they are
// HBM-bound: what matters is one pass over the activations, not the matrix cores.
//
//   forward / data-gradient: direct VALU kernels, one output pixel per thread, all output channels in registers,
//                            weights broadcast from LDS (every lane reads the same address);
//   weight-gradient        : MFMA GEMM with the pixels as K and the flattened (tap, ci) index as M
//                            ("im2col in the address": lane m adds its own tap offset to the pixel's LDS address),
//                            per-split slabs + fixed-order sum -> deterministic, no atomics.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TPB = 256;
constexpr int MAXW = 2048;          // weights held in LDS: KS*KS*Cin*Cout <= 2048 floats

struct SmallGeom { int N, H, W, Cin, Ho, Wo, Cout, KS, stride, pad; };

// y[n,ho,wo,:] = bias + sum_{kh,kw,ci} x[n,ho*s-p+kh,wo*s-p+kw,ci] * w[kh][kw][ci][:]      (COUT = 4*CQ channels)
template <int CQ>
__global__ void __launch_bounds__(TPB)
small_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
          float* __restrict__ y, SmallGeom g, int64_t npix) {
  __shared__ float4 ws[MAXW / 4];
  const int wq = g.KS * g.KS * g.Cin * CQ;                     // float4 units; Cout == 4*CQ
  for (int i = threadIdx.x; i < wq; i += TPB) ws[i] = ((const float4*)w)[i];
  __syncthreads();
  for (int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x; p < npix; p += (int64_t)gridDim.x * TPB) {
    const int wo = (int)(p % g.Wo);
    const int64_t q = p / g.Wo;
    const int ho = (int)(q % g.Ho);
    const int n = (int)(q / g.Ho);
    float4 acc[CQ];
#pragma unroll
    for (int c = 0; c < CQ; ++c) acc[c] = bias ? ((const float4*)bias)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float* xn = x + (size_t)n * g.H * g.W * g.Cin;
    for (int kh = 0; kh < g.KS; ++kh) {
      const int hi = ho * g.stride - g.pad + kh;
      if (hi < 0 || hi >= g.H) continue;
      for (int kw = 0; kw < g.KS; ++kw) {
        const int wi = wo * g.stride - g.pad + kw;
        if (wi < 0 || wi >= g.W) continue;
        const float* xp = xn + ((size_t)hi * g.W + wi) * g.Cin;
        const float4* wp = ws + (size_t)(kh * g.KS + kw) * g.Cin * CQ;
        for (int ci = 0; ci < g.Cin; ++ci) {
          const float v = xp[ci];
#pragma unroll
          for (int c = 0; c < CQ; ++c) {
            const float4 wv = wp[ci * CQ + c];
            acc[c].x = fmaf(v, wv.x, acc[c].x); acc[c].y = fmaf(v, wv.y, acc[c].y);
            acc[c].z = fmaf(v, wv.z, acc[c].z); acc[c].w = fmaf(v, wv.w, acc[c].w);
          }
        }
      }
    }
    float4* yp = (float4*)(y + (size_t)p * (4 * CQ));
#pragma unroll
    for (int c = 0; c < CQ; ++c) yp[c] = acc[c];
  }
}

// gx[n,hi,wi,ci] = sum_{kh,kw,co} gy[n,ho,wo,co] * w[kh][kw][ci][co],  ho*s - p + kh = hi      (Cin <= 8)
// S1: stride 1 (the 5x5 stems, reference network/blocks.py:123: the cycle pass differentiates them w.r.t. x_fake) -- without the
// per-tap `% stride` / `/ stride` of the general form (r05: 68 -> 64.6 us per launch; issuing a whole tap row's loads at once was tried
// and measured 2.5x SLOWER -- the kernel is bound by the latency chain of its L1 hits, not by instruction count).
template <int CQ, bool S1 = false>
__global__ void __launch_bounds__(TPB)
small_dgrad(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx, SmallGeom g,
            int64_t npix) {
  __shared__ float4 ws[MAXW / 4];
  const int wq = g.KS * g.KS * g.Cin * CQ;
  for (int i = threadIdx.x; i < wq; i += TPB) ws[i] = ((const float4*)w)[i];
  __syncthreads();
  for (int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x; p < npix; p += (int64_t)gridDim.x * TPB) {
    const int wi = (int)(p % g.W);
    const int64_t q = p / g.W;
    const int hi = (int)(q % g.H);
    const int n = (int)(q / g.H);
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    const float* gn = gy + (size_t)n * g.Ho * g.Wo * (4 * CQ);
    for (int kh = 0; kh < g.KS; ++kh) {
      const int hn = hi + g.pad - kh;
      if (hn < 0 || (!S1 && hn % g.stride)) continue;
      const int ho = S1 ? hn : hn / g.stride;
      if (ho >= g.Ho) continue;
      for (int kw = 0; kw < g.KS; ++kw) {
        const int wn = wi + g.pad - kw;
        if (wn < 0 || (!S1 && wn % g.stride)) continue;
        const int wo = S1 ? wn : wn / g.stride;
        if (wo >= g.Wo) continue;
        const float4* gp = (const float4*)(gn + ((size_t)ho * g.Wo + wo) * (4 * CQ));
        float4 gv[CQ];
#pragma unroll
        for (int c = 0; c < CQ; ++c) gv[c] = gp[c];
        const float4* wp = ws + (size_t)(kh * g.KS + kw) * g.Cin * CQ;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) {
          if (ci < g.Cin) {
#pragma unroll
            for (int c = 0; c < CQ; ++c) {
              const float4 wv = wp[ci * CQ + c];
              acc[ci] = fmaf(gv[c].x, wv.x, acc[ci]); acc[ci] = fmaf(gv[c].y, wv.y, acc[ci]);
              acc[ci] = fmaf(gv[c].z, wv.z, acc[ci]); acc[ci] = fmaf(gv[c].w, wv.w, acc[ci]);
            }
          }
        }
      }
    }
    float* op = gx + (size_t)p * g.Cin;
#pragma unroll
    for (int ci = 0; ci < 8; ++ci)
      if (ci < g.Cin) op[ci] = acc[ci];
  }
}

// ---- flattened-M MFMA weight gradient ---------------------------------------------------------------------
constexpr int FTH = 8, FTW = 16;           // output-pixel tile; each wave takes 2 rows
constexpr int MAXT = 4096;                 // staged input tile floats (IH*IW*Cin)

template <int MT>
__global__ void __launch_bounds__(TPB)
flat_wgrad(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ part, SmallGeom g,
           int tiles_x, int tiles_y, int tiles_per_split) {
  __shared__ float in_s[MAXT];
  __shared__ float gy_s[FTH * FTW * 16];
  __shared__ float red[MT * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, kq = lane >> 4;
  const int IH = (FTH - 1) * g.stride + g.KS, IW = (FTW - 1) * g.stride + g.KS;
  const int Mtot = g.KS * g.KS * g.Cin;
  // this lane's row m of the flattened (tap, ci) dimension per M tile: LDS offset of its tap / channel
  int moff[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = i * 16 + lm;
    const int tap = m / g.Cin, ci = m % g.Cin;
    moff[i] = m < Mtot ? ((tap / g.KS) * IW + (tap % g.KS)) * g.Cin + ci : -1;
  }
  f32x4 acc[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int tiles_img = tiles_x * tiles_y;
  const int total = g.N * tiles_img;
  const int t0 = blockIdx.x * tiles_per_split;
  const int t1 = min(t0 + tiles_per_split, total);
  for (int t = t0; t < t1; ++t) {
    const int n = t / tiles_img, rem = t % tiles_img;
    const int oy0 = (rem / tiles_x) * FTH, ox0 = (rem % tiles_x) * FTW;
    const int iy0 = oy0 * g.stride - g.pad, ix0 = ox0 * g.stride - g.pad;
    const float* xn = x + (size_t)n * g.H * g.W * g.Cin;
    const float* gn = gy + (size_t)n * g.Ho * g.Wo * g.Cout;
    __syncthreads();
    for (int u = tid; u < IH * IW * g.Cin; u += TPB) {
      const int ci = u % g.Cin, pix = u / g.Cin;
      const int iy = iy0 + pix / IW, ix = ix0 + pix % IW;
      in_s[u] = (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W) ? xn[((size_t)iy * g.W + ix) * g.Cin + ci] : 0.f;
    }
    for (int u = tid; u < FTH * FTW * 16; u += TPB) {
      const int c = u & 15, pix = u >> 4;
      const int oy = oy0 + pix / FTW, ox = ox0 + pix % FTW;
      gy_s[u] = (c < g.Cout && oy < g.Ho && ox < g.Wo) ? gn[((size_t)oy * g.Wo + ox) * g.Cout + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < FTH / 4; ++rr) {
      const int r = wave * (FTH / 4) + rr;
#pragma unroll
      for (int ks = 0; ks < FTW / 4; ++ks) {
        const int px = ks * 4 + kq;
        const float b = gy_s[(r * FTW + px) * 16 + lm];
        const int base = (r * g.stride * IW + px * g.stride) * g.Cin;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const float a = moff[i] >= 0 ? in_s[base + moff[i]] : 0.f;
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
      }
    }
  }
  // fixed-order combine of the 4 waves, then wave 0 writes the [Mtot][Cout] slab
  for (int src = 1; src < 4; ++src) {
    __syncthreads();
    if (wave == src) {
#pragma unroll
      for (int i = 0; i < MT; ++i) *(f32x4*)(red + ((size_t)i * 64 + lane) * 4) = acc[i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < MT; ++i) acc[i] += *(const f32x4*)(red + ((size_t)i * 64 + lane) * 4);
    }
  }
  if (wave == 0) {
    float* out = part + (size_t)blockIdx.x * Mtot * g.Cout;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = i * 16 + 4 * kq + r;               // accumulator row = M index, column lm = output channel
        if (m < Mtot && lm < g.Cout) out[(size_t)m * g.Cout + lm] = acc[i][r];
      }
  }
}

// out[e] = sum_s part[s][e], e < wsize (small, up to ~2000 splits): one block per 4 elements x 64 split lanes, four
// independent fp64 accumulators per lane (loads in flight), fixed-order LDS tree (deterministic)
constexpr int FS_COLS = 4;
__global__ void __launch_bounds__(TPB)
flat_sum(const float* __restrict__ part, float* __restrict__ out, int wsize, int splits) {
  constexpr int LANES = TPB / FS_COLS;
  __shared__ double sm[TPB];
  const int col = threadIdx.x % FS_COLS, sl = threadIdx.x / FS_COLS;
  const int e = blockIdx.x * FS_COLS + col;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (e < wsize) {
    int c = sl;
    for (; c + 3 * LANES < splits; c += 4 * LANES) {
      s0 += (double)part[(size_t)c * wsize + e];
      s1 += (double)part[(size_t)(c + LANES) * wsize + e];
      s2 += (double)part[(size_t)(c + 2 * LANES) * wsize + e];
      s3 += (double)part[(size_t)(c + 3 * LANES) * wsize + e];
    }
    for (; c < splits; c += LANES) s0 += (double)part[(size_t)c * wsize + e];
  }
  sm[threadIdx.x] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0 && e < wsize) {
    double t = 0.0;
    for (int l = 0; l < LANES; ++l) t += sm[l * FS_COLS + col];
    out[e] = (float)t;
  }
}

// ---- the 5x5 stems (Cin 1 or 1 + n_modal = 5 -> 8 channels, stride 1, pad 2: network/blocks.py:123, ugan.py:26) -----------------
// The generic kernels above do one global load and two LDS reads per 8 FMAs (forward: 17-25 TFLOP/s) or one LDS gather per
// MFMA with half the matrix padding (weight gradient: 10-18 TFLOP/s) -- 0.45 ms of the uganConsis iteration for convs whose
// tensors stream in 10 us.  Here a workgroup owns a 16 x 64 pixel tile: the input tile with its halo is staged ONCE into LDS as
// channel planes [ci][20][68] (zero padded), a thread computes 4 consecutive pixels x all 8 output channels, and a
// (kernel row, input channel) pair costs two 16-byte LDS reads of inputs + ten broadcast reads of weights for 160 FMAs.
constexpr int STY = 16, STX = 64, SKS = 5, SPD = 2, SCO = 8;
constexpr int SIH = STY + SKS - 1, SIW = STX + SKS - 1;          // 20 x 68 haloed tile

template <int CIN>
__device__ __forceinline__ void stem_stage(const float* __restrict__ x, float* __restrict__ in_s, int n, int y0, int x0,
                                           int H, int W) {
  // global [y][x][ci] -> LDS planes [ci][SIH][SIW]
  for (int u = threadIdx.x; u < SIH * SIW * CIN; u += TPB) {
    const int ci = u % CIN, pix = u / CIN;
    const int iy = pix / SIW, ix = pix % SIW;
    const int gy_ = y0 + iy - SPD, gx_ = x0 + ix - SPD;
    float v = 0.f;
    if (gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W) v = x[(((size_t)n * H + gy_) * W + gx_) * CIN + ci];
    in_s[(ci * SIH + iy) * SIW + ix] = v;
  }
}

template <int CIN>
__global__ void __launch_bounds__(TPB)
stem_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
         int N, int H, int W) {
  __shared__ __attribute__((aligned(16))) float in_s[CIN * SIH * SIW];
  __shared__ __attribute__((aligned(16))) float w_s[SKS * SKS * CIN * SCO];
  const int tiles_x = W / STX, tiles_y = H / STY;
  const int t = blockIdx.x;
  const int n = t / (tiles_x * tiles_y), rem = t % (tiles_x * tiles_y);
  const int y0 = (rem / tiles_x) * STY, x0 = (rem % tiles_x) * STX;
  for (int u = threadIdx.x; u < SKS * SKS * CIN * SCO; u += TPB) w_s[u] = w[u];
  stem_stage<CIN>(x, in_s, n, y0, x0, H, W);
  __syncthreads();
  const int row = threadIdx.x >> 4, cg = threadIdx.x & 15;
  float acc[4][SCO];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int c = 0; c < SCO; ++c) acc[p][c] = bias ? bias[c] : 0.f;
#pragma unroll
  for (int kh = 0; kh < SKS; ++kh)
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
      float v[8];
      const float* ip = in_s + (ci * SIH + row + kh) * SIW + 4 * cg;
      *(float4*)v = *(const float4*)ip; *(float4*)(v + 4) = *(const float4*)(ip + 4);
#pragma unroll
      for (int kw = 0; kw < SKS; ++kw) {
        float wv[SCO];
        const float* wp = w_s + ((kh * SKS + kw) * CIN + ci) * SCO;
        *(float4*)wv = *(const float4*)wp; *(float4*)(wv + 4) = *(const float4*)(wp + 4);
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int c = 0; c < SCO; ++c) acc[p][c] = fmaf(v[p + kw], wv[c], acc[p][c]);
      }
    }
  float* yp = y + (((size_t)n * H + y0 + row) * W + x0 + 4 * cg) * SCO;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    *(float4*)(yp + p * SCO) = *(float4*)acc[p];
    *(float4*)(yp + p * SCO + 4) = *(float4*)(acc[p] + 4);
  }
}

// data-gradient of the ONE-channel stems (r05; the cycle pass differentiates G's stems w.r.t. x_fake, reference
// trainer/uganConsisTrainer.py:159): gx[y][x] = sum_{kh,kw,co} gy[y + 2 - kh][x + 2 - kw][co] * w[kh][kw][0][co].  The mirror of
// stem_fwd: the gy tile with its halo staged once as channel planes [co][20][68], a thread computes 4 consecutive pixels; per (kernel
// row, output channel) two 16-byte LDS reads + five broadcast weight reads for 20 FMAs.  The general kernel (small_dgrad) walks 25 taps
// of dependent L1 hits per pixel: 65 us for tensors that stream in 10.
__global__ void __launch_bounds__(TPB)
stem_dgrad1(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx, int N, int H, int W) {
  __shared__ __attribute__((aligned(16))) float g_s[SCO * SIH * SIW];
  __shared__ __attribute__((aligned(16))) float w_s[SKS * SKS * SCO];
  const int tiles_x = W / STX, tiles_y = H / STY;
  const int t = blockIdx.x;
  const int n = t / (tiles_x * tiles_y), rem = t % (tiles_x * tiles_y);
  const int y0 = (rem / tiles_x) * STY, x0 = (rem % tiles_x) * STX;
  for (int u = threadIdx.x; u < SKS * SKS * SCO; u += TPB) w_s[u] = w[u];            // [kh][kw][ci = 0][co]
  // global [y][x][co] (two float4 per pixel) -> LDS planes [co][SIH][SIW], zero outside the image
  for (int u = threadIdx.x; u < SIH * SIW * 2; u += TPB) {
    const int h = u & 1, pix = u >> 1;
    const int iy = pix / SIW, ix = pix % SIW;
    const int gy_ = y0 + iy - SPD, gx_ = x0 + ix - SPD;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gy_ >= 0 && gy_ < H && gx_ >= 0 && gx_ < W) v = *(const float4*)(gy + (((size_t)n * H + gy_) * W + gx_) * SCO + 4 * h);
    float* d = g_s + ((4 * h) * SIH + iy) * SIW + ix;
    d[0] = v.x; d[SIH * SIW] = v.y; d[2 * SIH * SIW] = v.z; d[3 * SIH * SIW] = v.w;
  }
  __syncthreads();
  const int row = threadIdx.x >> 4, cg = threadIdx.x & 15;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  // output pixel (row, 4cg + p) takes gy at tile coordinates (row + 4 - kh, 4cg + p + 4 - kw): with jh = 4 - kh, jw = 4 - kw the window
  // is the forward kernel's, read against the flipped weights
#pragma unroll
  for (int jh = 0; jh < SKS; ++jh)
#pragma unroll
    for (int co = 0; co < SCO; ++co) {
      float v[8];
      const float* ip = g_s + (co * SIH + row + jh) * SIW + 4 * cg;
      *(float4*)v = *(const float4*)ip; *(float4*)(v + 4) = *(const float4*)(ip + 4);
#pragma unroll
      for (int jw = 0; jw < SKS; ++jw) {
        const float wv = w_s[((SKS - 1 - jh) * SKS + (SKS - 1 - jw)) * SCO + co];
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p] = fmaf(v[p + jw], wv, acc[p]);
      }
    }
  *(float4*)(gx + ((size_t)n * H + y0 + row) * W + x0 + 4 * cg) = *(float4*)acc;
}

// weight gradient: gw[kh][kw][ci][co] = sum over pixels of x[y + kh - 2][x + kw - 2][ci] * gy[y][x][co].  5 * CIN "pairs"
// q = (kh, ci); TPP threads share a pair (8 for CIN = 5: 200 live threads, 32 for CIN = 1: 160) and split the tile's 256 pixel
// quads; a thread keeps its pair's 5 kw x 8 co accumulators in registers ACROSS the tiles of its workgroup (two 16-byte input
// reads + eight of gy per 160 FMAs) and the TPP partial sums are combined once at the end (xor tree, fixed order) into the
// workgroup's slab.  gy tile in LDS with a 36-float quad stride (the eight quads a wave reads
// together hit disjoint bank quartets).
constexpr int SGQ = 36;
template <int CIN>
__global__ void __launch_bounds__(TPB)
stem_wgrad(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ part, int N, int H, int W,
           int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* in_s = smem;                                   // [CIN][SIH][SIW]
  float* gy_s = smem + CIN * SIH * SIW;                 // [256 quads][SGQ]: 4 pixels x 8 channels + pad
  constexpr int NACC = SKS * SCO;                       // per-thread accumulators
  constexpr int TPP = (CIN == 1) ? 32 : 8;              // threads per pair
  const int tiles_x = W / STX, tiles_y = H / STY, total = N * tiles_x * tiles_y;
  const int q = threadIdx.x / TPP, sub = threadIdx.x % TPP;       // pair, share of the quads
  const bool live = q < SKS * CIN;
  const int kh = q / CIN, k2 = q % CIN;                           // kernel row, input channel
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.f;
  const int t0 = blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, total);
  for (int t = t0; t < t1; ++t) {
    const int n = t / (tiles_x * tiles_y), rem = t % (tiles_x * tiles_y);
    const int y0 = (rem / tiles_x) * STY, x0 = (rem % tiles_x) * STX;
    __syncthreads();
    stem_stage<CIN>(x, in_s, n, y0, x0, H, W);
    for (int u = threadIdx.x; u < STY * (STX / 4) * 8; u += TPB) {           // float4 units: quad, pixel of the quad, half
      const int half = u & 1, px = (u >> 1) & 3, quad = u >> 3;
      const int row = quad >> 4, cgq = quad & 15;
      *(float4*)(gy_s + quad * SGQ + px * SCO + 4 * half) =
          *(const float4*)(gy + (((size_t)n * H + y0 + row) * W + x0 + 4 * cgq + px) * SCO + 4 * half);
    }
    __syncthreads();
    if (live) {
      for (int k = 0; k < 256 / TPP; ++k) {
        const int quad = sub + TPP * k;
        const int row = quad >> 4, cgq = quad & 15;
        float g[4][SCO];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          *(float4*)g[p] = *(const float4*)(gy_s + quad * SGQ + p * SCO);
          *(float4*)(g[p] + 4) = *(const float4*)(gy_s + quad * SGQ + p * SCO + 4);
        }
        {
          float v[8];
          const float* ip = in_s + (k2 * SIH + row + kh) * SIW + 4 * cgq;
          *(float4*)v = *(const float4*)ip; *(float4*)(v + 4) = *(const float4*)(ip + 4);
#pragma unroll
          for (int kw = 0; kw < SKS; ++kw)
#pragma unroll
            for (int c = 0; c < SCO; ++c)
              acc[kw * SCO + c] += v[kw] * g[0][c] + v[kw + 1] * g[1][c] + v[kw + 2] * g[2][c] + v[kw + 3] * g[3][c];
        }
      }
    }
  }
  // the eight threads of a pair: xor tree over sub (lanes 8q .. 8q+7 of one wave), then thread sub == 0 writes the pair's rows
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    float a = acc[i];
    a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 4, 64);
    if constexpr (TPP == 32) { a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 16, 64); }
    acc[i] = a;
  }
  if (live && sub == 0) {
    float* out = part + (size_t)blockIdx.x * SKS * SKS * CIN * SCO;
#pragma unroll
    for (int kw = 0; kw < SKS; ++kw)
#pragma unroll
      for (int c = 0; c < SCO; ++c) out[((kh * SKS + kw) * CIN + k2) * SCO + c] = acc[kw * SCO + c];
  }
}

inline bool stem_shape(const SmallGeom& g) {
  static const bool on = [] { const char* e = getenv("SMSUT_STEM"); return !e || atoi(e) != 0; }();
  return on && g.KS == SKS && g.stride == 1 && g.pad == SPD && g.Cout == SCO && (g.Cin == 1 || g.Cin == 5) && g.H % STY == 0 &&
         g.W % STX == 0 && g.Ho == g.H && g.Wo == g.W;
}
inline int stem_wgrad_wgs(const SmallGeom& g, int max_slabs) {
  const int total = g.N * (g.H / STY) * (g.W / STX);
  int wgs = total < 512 ? total : 512;
  if (wgs > max_slabs) wgs = max_slabs;
  return wgs < 1 ? 1 : wgs;
}

inline bool geom_ok(const SmallGeom& g) {
  return g.N > 0 && g.H > 0 && g.W > 0 && g.Cin > 0 && g.Cout > 0 && g.KS > 0 && g.stride > 0 && g.pad >= 0 &&
         g.Ho == (g.H + 2 * g.pad - g.KS) / g.stride + 1 && g.Wo == (g.W + 2 * g.pad - g.KS) / g.stride + 1 && g.Ho > 0 &&
         g.Wo > 0;
}

struct FlatPlan { int tiles_x, tiles_y, splits, tiles_per_split; };
inline FlatPlan flat_plan(const SmallGeom& g) {
  FlatPlan p;
  p.tiles_x = (g.Wo + FTW - 1) / FTW;
  p.tiles_y = (g.Ho + FTH - 1) / FTH;
  const int total = g.N * p.tiles_x * p.tiles_y;
  int want = total < 2048 ? total : 2048;
  p.tiles_per_split = (total + want - 1) / want;
  p.splits = (total + p.tiles_per_split - 1) / p.tiles_per_split;
  return p;
}

}  // namespace

extern "C" {

// forward / data-gradient eligibility: Cout in {4, 8, 12, 16}, Cin <= 8, weights fit the LDS image
int smsut_conv2d_small_supported(int KS, int Cin, int Cout) {
  return KS >= 1 && Cin >= 1 && Cin <= 8 && Cout >= 4 && Cout <= 16 && (Cout % 4) == 0 && KS * KS * Cin * Cout <= MAXW;
}

int smsut_conv2d_small_fwd(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin,
                           int Ho, int Wo, int Cout, int KS, int stride, int pad, void* stream) {
  SmallGeom g{N, H, W, Cin, Ho, Wo, Cout, KS, stride, pad};
  SMSUT_REQUIRE(x && w && y && geom_ok(g) && smsut_conv2d_small_supported(KS, Cin, Cout));
  const int64_t npix = (int64_t)N * Ho * Wo;
  const int grid = ew_grid(npix) * 2;
  hipStream_t st = (hipStream_t)stream;
  if (stem_shape(g)) {                                   // the 5x5 stems: tiled kernel (see stem_fwd)
    const int tiles = N * (H / STY) * (W / STX);
    if (Cin == 1) stem_fwd<1><<<tiles, TPB, 0, st>>>(x, w, bias, y, N, H, W);
    else stem_fwd<5><<<tiles, TPB, 0, st>>>(x, w, bias, y, N, H, W);
    SMSUT_LAUNCH_CHECK();
    return SMSUT_OK;
  }
  switch (Cout / 4) {
    case 1: small_fwd<1><<<grid, TPB, 0, st>>>(x, w, bias, y, g, npix); break;
    case 2: small_fwd<2><<<grid, TPB, 0, st>>>(x, w, bias, y, g, npix); break;
    case 3: small_fwd<3><<<grid, TPB, 0, st>>>(x, w, bias, y, g, npix); break;
    default: small_fwd<4><<<grid, TPB, 0, st>>>(x, w, bias, y, g, npix); break;
  }
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

int smsut_conv2d_small_dgrad(const float* gy, const float* w, float* gx, int N, int H, int W, int Cin, int Ho, int Wo,
                             int Cout, int KS, int stride, int pad, void* stream) {
  SmallGeom g{N, H, W, Cin, Ho, Wo, Cout, KS, stride, pad};
  SMSUT_REQUIRE(gy && w && gx && geom_ok(g) && smsut_conv2d_small_supported(KS, Cin, Cout));
  const int64_t npix = (int64_t)N * H * W;
  const int grid = ew_grid(npix) * 2;
  hipStream_t st = (hipStream_t)stream;
  if (stem_shape(g) && Cin == 1) {                       // the one-channel 5x5 stems: tiled kernel (see stem_dgrad1)
    stem_dgrad1<<<N * (H / STY) * (W / STX), TPB, 0, st>>>(gy, w, gx, N, H, W);
    SMSUT_LAUNCH_CHECK();
    return SMSUT_OK;
  }
#define SMALL_DG(Q) do { if (stride == 1) small_dgrad<Q, true><<<grid, TPB, 0, st>>>(gy, w, gx, g, npix);  \
                         else small_dgrad<Q, false><<<grid, TPB, 0, st>>>(gy, w, gx, g, npix); } while (0)
  switch (Cout / 4) {
    case 1: SMALL_DG(1); break;
    case 2: SMALL_DG(2); break;
    case 3: SMALL_DG(3); break;
    default: SMALL_DG(4); break;
  }
#undef SMALL_DG
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

// weight-gradient eligibility: KS*KS*Cin <= 128 rows of the flattened M dimension, Cout <= 16, tile fits LDS
int smsut_conv2d_flat_wgrad_supported(int KS, int stride, int Cin, int Cout) {
  const int IH = (FTH - 1) * stride + KS, IW = (FTW - 1) * stride + KS;
  return KS >= 1 && stride >= 1 && Cin >= 1 && Cout >= 1 && Cout <= 16 && KS * KS * Cin <= 128 && IH * IW * Cin <= MAXT;
}

int64_t smsut_conv2d_flat_wgrad_ws(int N, int Ho, int Wo, int Cin, int Cout, int KS) {
  SmallGeom g{N, 0, 0, Cin, Ho, Wo, Cout, KS, 1, 0};
  return (int64_t)flat_plan(g).splits * KS * KS * Cin * Cout;
}

int smsut_conv2d_flat_wgrad(const float* x, const float* gy, float* gw, float* workspace, int N, int H, int W, int Cin,
                            int Ho, int Wo, int Cout, int KS, int stride, int pad, void* stream) {
  SmallGeom g{N, H, W, Cin, Ho, Wo, Cout, KS, stride, pad};
  SMSUT_REQUIRE(x && gy && gw && workspace && geom_ok(g) && smsut_conv2d_flat_wgrad_supported(KS, stride, Cin, Cout));
  const FlatPlan p = flat_plan(g);
  hipStream_t st = (hipStream_t)stream;
  if (stem_shape(g)) {                                   // the 5x5 stems: register-tiled VALU kernel (see stem_wgrad)
    const int wgs = stem_wgrad_wgs(g, p.splits);         // (the workspace holds p.splits slabs)
    const int total = N * (H / STY) * (W / STX);
    const int tpw = (total + wgs - 1) / wgs;
    const int nwg = (total + tpw - 1) / tpw;
    const int wsz = KS * KS * Cin * Cout;
    if (Cin == 1) {
      constexpr size_t sh = (size_t)(1 * SIH * SIW + STY * (STX / 4) * SGQ) * sizeof(float);
      stem_wgrad<1><<<nwg, TPB, sh, st>>>(x, gy, workspace, N, H, W, tpw);
    } else {
      constexpr size_t sh = (size_t)(5 * SIH * SIW + STY * (STX / 4) * SGQ) * sizeof(float);
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute((const void*)stem_wgrad<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); attr = true; }
      stem_wgrad<5><<<nwg, TPB, sh, st>>>(x, gy, workspace, N, H, W, tpw);
    }
    flat_sum<<<(wsz + FS_COLS - 1) / FS_COLS, TPB, 0, st>>>(workspace, gw, wsz, nwg);
    SMSUT_LAUNCH_CHECK();
    return SMSUT_OK;
  }
  const int mt = (KS * KS * Cin + 15) / 16;
  if (mt <= 1) flat_wgrad<1><<<p.splits, TPB, 0, st>>>(x, gy, workspace, g, p.tiles_x, p.tiles_y, p.tiles_per_split);
  else if (mt <= 2) flat_wgrad<2><<<p.splits, TPB, 0, st>>>(x, gy, workspace, g, p.tiles_x, p.tiles_y, p.tiles_per_split);
  else if (mt <= 4) flat_wgrad<4><<<p.splits, TPB, 0, st>>>(x, gy, workspace, g, p.tiles_x, p.tiles_y, p.tiles_per_split);
  else flat_wgrad<8><<<p.splits, TPB, 0, st>>>(x, gy, workspace, g, p.tiles_x, p.tiles_y, p.tiles_per_split);
  const int wsize = KS * KS * Cin * Cout;
  flat_sum<<<(wsize + FS_COLS - 1) / FS_COLS, TPB, 0, st>>>(workspace, gw, wsize, p.splits);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

}  // extern "C"
