// 1x1 convolutions as streaming GEMMs on the matrix cores (fp32 in / fp32 accumulate).
//
// The 1x1 convs of the hot path -- BasicBlock / BottleBlock shortcuts (network/blocks.py:63,95-97), the translator's
// up-path (blocks.py:45), nn.Linear of PatchSampleF (ugan.py:295) -- carry ~5 % of the FLOPs but touch full-resolution
// tensors: they are HBM-bound (2.7 FLOP/B at 8->16 @256^2).  Unlike the 3x3 kernels there is no halo to share, so the
// activation operand never goes through LDS: every lane loads its MFMA A fragment (4 consecutive channels of one pixel,
// 16 B) straight from global memory, waves are independent (no barrier in the loop), and only the small weight matrix
// is staged in LDS once per workgroup.  r01 profile: the LDS-tiled kernel spent 64 us per call on these shapes
// (0.8 TB/s); this one is bounded by the memory pipe.
//
//   forward     y[p, n] = sum_k x[p, k] * W[k][n]                 (transposed = 0, W is [Cin][Cout])
//   data-grad   gx[p, k] = sum_n gy[p, n] * W[k][n]               (transposed = 1: same kernel, W read transposed)
//   weight-grad gW[k][n] = sum_p x[p, k] * gy[p, n]               (pixels are the GEMM K; per-split slabs + fixed-order sum)
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TPB = 256;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// MR x 16 pixels per wave step, NR x 16 output channels per workgroup.
// PS ("pixel shuffle"): ConvTranspose2d(k=2, s=2) as ONE 1x1 conv with Ndim = 4 * Cout result columns (tap-major), the result of
// column (tap, co) of input pixel (h, w) stored at output pixel (2h + tap/2, 2w + tap%2) -- x is read once per 16*NR-column slab
// instead of once per tap (network/blocks.py:41; weights [tap][Cin][Cout]; Cout % 16 == 0 so a 16-column tile is one tap; ps_w = W).
template <int MR, int NR, bool DUAL = false, bool PS = false>
__global__ void __launch_bounds__(TPB)
conv1x1_fwd(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, float* __restrict__ stats,
            int64_t P, int HW, int Kdim, int Ndim, int transposed, float* __restrict__ y2 = nullptr, int split = 0,
            const float* __restrict__ x2 = nullptr, int ca = 0, int ps_w = 0) {
  constexpr int CO_T = 16 * NR;
  extern __shared__ float w_s[];                 // [chunks][4 kq][CO_T][4]: k = 16*chunk + 4*kq + j
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, kq = lane >> 4;
  const int co0 = blockIdx.y * CO_T;
  const int chunks = (Kdim + 15) / 16;
  for (int u = tid; u < chunks * 4 * CO_T; u += TPB) {
    const int n = u % CO_T;
    const int kg = (u / CO_T) * 4;               // = 16*chunk + 4*kq
    const int ng = co0 + n;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ng < Ndim && kg < Kdim) {
      if constexpr (PS) {
        const int cout = Ndim >> 2, tap = ng / cout;
        const float* p = w + ((size_t)tap * Kdim + kg) * cout + (ng - tap * cout);
        v.x = p[0]; v.y = p[cout]; v.z = p[2 * (size_t)cout]; v.w = p[3 * (size_t)cout];
      } else if (!transposed) {
        const float* p = w + (size_t)kg * Ndim + ng;
        v.x = p[0]; v.y = p[Ndim]; v.z = p[2 * (size_t)Ndim]; v.w = p[3 * (size_t)Ndim];
      } else {
        v = *(const float4*)(w + (size_t)ng * Kdim + kg);
      }
    }
    *(float4*)(w_s + (size_t)u * 4) = v;
  }
  __syncthreads();

  const int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * (16 * MR);
  if (p0 >= P) return;
  f32x4 acc[MR][NR];
#pragma unroll
  for (int i = 0; i < MR; ++i)
#pragma unroll
    for (int j = 0; j < NR; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // x2 != null: the input is the virtual cat([x, x2]) (common.h): 16-channel chunks below ca come from x [P][ca], the
  // rest from x2 [P][Kdim - ca]
  const float* xp[MR];
  const float* xq[MR];
  bool pok[MR];
  const int sa = DUAL ? ca : Kdim;                 // (template flag: the plain form keeps one pointer set and no select)
#pragma unroll
  for (int i = 0; i < MR; ++i) {
    const int64_t p = p0 + i * 16 + lm;
    pok[i] = p < P;
    xp[i] = x + (size_t)(pok[i] ? p : 0) * sa + 4 * kq;
    xq[i] = DUAL ? x2 + (size_t)(pok[i] ? p : 0) * (Kdim - ca) + 4 * kq - ca : xp[i];
  }
// (runtime trip count: the partial unroll request is not honoured for every instantiation)
  for (int c = 0; c < chunks; ++c) {
    const bool kok = c * 16 + 4 * kq < Kdim;
    const bool second = DUAL && c * 16 >= ca;
    f32x4 a[MR], b[NR];
#pragma unroll
    for (int i = 0; i < MR; ++i)
      a[i] = (pok[i] && kok) ? *(const f32x4*)((second ? xq[i] : xp[i]) + c * 16) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NR; ++j) b[j] = *(const f32x4*)(w_s + ((size_t)(c * 4 + kq) * CO_T + j * 16 + lm) * 4);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j) acc[i][j] = mfma16(a[i][s], b[j][s], acc[i][j]);
  }
  // acc[i][j][r]: pixel p0 + 16 i + 4 kq + r, channel co0 + 16 j + lm
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int co = co0 + j * 16 + lm;
    // split output (y2 != null, split % 16 == 0): channels >= split go to y2 [P][Ndim - split] (see conv_mfma_fwd_p)
    const bool hi = y2 && co0 + j * 16 >= split;
    float* const yo = hi ? y2 : y;
    const int os = !y2 ? Ndim : (hi ? Ndim - split : split);
    const int oc = hi ? co - split : co;
    float s1 = 0.f, s2 = 0.f;
    if constexpr (PS) {
      const int cout = Ndim >> 2, tap = (co0 + j * 16) / cout, cc = co - tap * cout;
      const int W2 = 2 * ps_w;
#pragma unroll
      for (int i = 0; i < MR; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t p = p0 + i * 16 + 4 * kq + r;
          if (p < P && co < Ndim) {
            const int row = (int)p / ps_w;                         // = n * H + h   (host: P < 2^31 / (4 * cout))
            const int wx = (int)p - row * ps_w;
            y[((size_t)(2 * row + (tap >> 1)) * W2 + 2 * wx + (tap & 1)) * cout + cc] = acc[i][j][r];
          }
        }
      continue;
    }
#pragma unroll
    for (int i = 0; i < MR; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t p = p0 + i * 16 + 4 * kq + r;
        if (p < P && co < Ndim) {
          const float v = acc[i][j][r];
          yo[(size_t)p * os + oc] = v;
          s1 += v; s2 += v * v;
        }
      }
    if (stats) {                                  // InstanceNorm partials: one tile per wave step (HW % (16*MR) == 0)
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (kq == 0 && co < Ndim) {
        const int tiles = HW / (16 * MR);
        const int64_t n_img = p0 / HW;
        const int tile = (int)((p0 % HW) / (16 * MR));
        float* o = stats + (((size_t)n_img * tiles + tile) * Ndim + co) * 2;
        o[0] = s1; o[1] = s2;
      }
    }
  }
}

#ifndef W1_UNR
#define W1_UNR 4     // r02 A/B (scratch/w1_ab.py): 1 -> 4 groups: 107 -> 80 us at 32x256^2 (16+16)->16, 60 -> 38 us at 8->16; 8 and 16 no better
#endif
// weight gradient: each wave walks groups of 4 pixels of its workgroup's pixel range with direct global loads
// PS: weight gradient of ConvTranspose2d(k=2, s=2): gw[tap][ci][co] = sum_p x[p][ci] * gy[2p + tap][co] as the 1x1 weight gradient
// with Cout = 4 * cout gathered columns (the B operand of column (tap, co) is read at output pixel (2h + tap/2, 2w + tap%2));
// x is read once per 16*COT-column slab instead of once per tap, and a slab's two taps are the two halves of the same gy lines.
template <int CIT, int COT, bool PS = false>
__global__ void __launch_bounds__(TPB)
conv1x1_wgrad(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ part, int64_t P, int Cin,
              int Cout, int64_t pix_per_split, const float* __restrict__ x2 = nullptr, int ca = 0, int ps_w = 0) {
  __shared__ float red[CIT * COT * 64 * 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, kq = lane >> 4;
  const int ci0 = blockIdx.y * (16 * CIT), co0 = blockIdx.z * (16 * COT);
  const int64_t pb = (int64_t)blockIdx.x * pix_per_split;
  const int64_t pe = min(pb + pix_per_split, P);
  f32x4 acc[CIT][COT];
#pragma unroll
  for (int i = 0; i < CIT; ++i)
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bool iok[CIT], jok[COT];
#pragma unroll
  for (int i = 0; i < CIT; ++i) iok[i] = ci0 + i * 16 + lm < Cin;
#pragma unroll
  for (int j = 0; j < COT; ++j) jok[j] = co0 + j * 16 + lm < Cout;
  CatSrc xs[CIT];                                  // per 16-channel tile: which tensor of the (virtual) cat holds it
#pragma unroll
  for (int i = 0; i < CIT; ++i) xs[i] = cat_src(x, x2, Cin, ca, ci0 + i * 16);
  const float* gb = gy + co0 + lm;
  [[maybe_unused]] const int cout = Cout >> 2;             // PS: channels of the transposed conv's output
  [[maybe_unused]] int ps_tap[COT], ps_cc[COT];
  if constexpr (PS) {
#pragma unroll
    for (int j = 0; j < COT; ++j) { ps_tap[j] = (co0 + j * 16) / cout; ps_cc[j] = co0 + j * 16 - ps_tap[j] * cout + lm; }
  }
  // W1_UNR pixel groups per trip: all their loads are issued before the first MFMA (the rolled one-group loop kept ONE
  // 4-byte load per operand tile in flight per lane and ran at 3.4-3.8 TB/s; see profiles/r02_notes.md)
  constexpr int UNR = W1_UNR;
  for (int64_t p0 = pb + 4 * wave + kq; p0 - kq < pe; p0 += 16 * UNR) {      // this lane's pixel (the MFMA k index)
    float a[UNR][CIT], b[UNR][COT];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int64_t p = p0 + 16 * u;
      const bool ok = p < pe;
#pragma unroll
      for (int i = 0; i < CIT; ++i) a[u][i] = (ok && iok[i]) ? xs[i].p[(size_t)p * xs[i].stride + ci0 + i * 16 + lm - xs[i].coff] : 0.f;
      if constexpr (PS) {
        const int row = (int)p / ps_w;                             // = n * H + h   (host: P < 2^31 / (4 * cout))
        const int wx = (int)p - row * ps_w;
#pragma unroll
        for (int j = 0; j < COT; ++j)
          b[u][j] = (ok && jok[j]) ? gy[((size_t)(2 * row + (ps_tap[j] >> 1)) * (2 * ps_w) + 2 * wx + (ps_tap[j] & 1)) * cout + ps_cc[j]] : 0.f;
      } else {
#pragma unroll
        for (int j = 0; j < COT; ++j) b[u][j] = (ok && jok[j]) ? gb[(size_t)p * Cout + j * 16] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u)
#pragma unroll
      for (int i = 0; i < CIT; ++i)
#pragma unroll
        for (int j = 0; j < COT; ++j) acc[i][j] = mfma16(a[u][i], b[u][j], acc[i][j]);
  }
  for (int src = 1; src < 4; ++src) {             // fixed-order combine of the 4 waves
    __syncthreads();
    if (wave == src) {
#pragma unroll
      for (int i = 0; i < CIT; ++i)
#pragma unroll
        for (int j = 0; j < COT; ++j) *(f32x4*)(red + ((size_t)(i * COT + j) * 64 + lane) * 4) = acc[i][j];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int i = 0; i < CIT; ++i)
#pragma unroll
        for (int j = 0; j < COT; ++j) acc[i][j] += *(const f32x4*)(red + ((size_t)(i * COT + j) * 64 + lane) * 4);
    }
  }
  if (wave == 0) {
    float* out = part + (size_t)blockIdx.x * Cin * Cout;
#pragma unroll
    for (int i = 0; i < CIT; ++i)
#pragma unroll
      for (int j = 0; j < COT; ++j) {
        const int co = co0 + j * 16 + lm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ci = ci0 + i * 16 + 4 * kq + r;
          if constexpr (PS) {                                      // final layout [tap][ci][co]
            if (ci < Cin && co < Cout) out[((size_t)ps_tap[j] * Cin + ci) * cout + ps_cc[j]] = acc[i][j][r];
          } else if (ci < Cin && co < Cout) out[(size_t)ci * Cout + co] = acc[i][j][r];
        }
      }
  }
}

// out[e] = sum_s part[s][e]: 16 float4 columns x 16 split lanes per block, four independent accumulators per lane
// (loads in flight), fixed-order LDS tree (deterministic); wsize = Cin * Cout is a multiple of 4 on this path
__global__ void __launch_bounds__(TPB)
sum_parts(const float* __restrict__ part, float* __restrict__ out, int wsize, int splits) {
  __shared__ float4 sm[TPB];
  const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int e = (blockIdx.x * 16 + col) * 4;
  float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0, t2 = t0, t3 = t0;
  if (e < wsize) {
    int c = sl;
    for (; c + 48 < splits; c += 64) {
      const float4 v0 = *(const float4*)(part + (size_t)c * wsize + e);
      const float4 v1 = *(const float4*)(part + (size_t)(c + 16) * wsize + e);
      const float4 v2 = *(const float4*)(part + (size_t)(c + 32) * wsize + e);
      const float4 v3 = *(const float4*)(part + (size_t)(c + 48) * wsize + e);
      t0.x += v0.x; t0.y += v0.y; t0.z += v0.z; t0.w += v0.w; t1.x += v1.x; t1.y += v1.y; t1.z += v1.z; t1.w += v1.w;
      t2.x += v2.x; t2.y += v2.y; t2.z += v2.z; t2.w += v2.w; t3.x += v3.x; t3.y += v3.y; t3.z += v3.z; t3.w += v3.w;
    }
    for (; c < splits; c += 16) {
      const float4 v = *(const float4*)(part + (size_t)c * wsize + e);
      t0.x += v.x; t0.y += v.y; t0.z += v.z; t0.w += v.w;
    }
  }
  sm[threadIdx.x] = make_float4((t0.x + t1.x) + (t2.x + t3.x), (t0.y + t1.y) + (t2.y + t3.y), (t0.z + t1.z) + (t2.z + t3.z),
                                (t0.w + t1.w) + (t2.w + t3.w));
  __syncthreads();
  if (sl == 0 && e < wsize) {
    float4 t = sm[col];
    for (int l = 1; l < 16; ++l) { const float4 v = sm[l * 16 + col]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    *(float4*)(out + e) = t;
  }
}

struct Plan1 { int splits; int64_t pps; };
inline Plan1 plan_wgrad1(int64_t P, int Cin, int Cout, int cit, int cot) {
  const int slabs = ((Cin + 16 * cit - 1) / (16 * cit)) * ((Cout + 16 * cot - 1) / (16 * cot));
  int64_t want = (1024 + slabs - 1) / slabs;
  const int64_t maxs = P / 256 > 0 ? P / 256 : 1;           // >= 256 pixels per workgroup
  if (want > maxs) want = maxs;
  if (want < 1) want = 1;
  Plan1 p;
  p.pps = ((P + want - 1) / want + 15) / 16 * 16;           // multiple of 16: whole 4-pixel groups per wave
  p.splits = (int)((P + p.pps - 1) / p.pps);
  return p;
}


// ---- "thin" 1x1 layers: <= 8 output channels (decoder.fc 16 -> n_label+1, the translator's 16 -> 1 head;
// network/blocks.py:123-125, ugan.py:70).  Their data- and weight-gradients are pure streaming problems (16 -> 5 at
// 32x256^2: 176 MB, 0.3 GFLOP) that the generic direct kernels ran at ~1.1 / 1.5 TB/s.  One work item = (pixel, 4
// consecutive wide channels): the wide tensor moves as perfectly coalesced float4s, the thin tensor's <= 8 values per
// pixel are shared by CI/4 neighbouring lanes, the weights sit in LDS as [8][CI] (zero beyond Cout).
constexpr int THIN_MAX = 8;

template <int CI>
__global__ void __launch_bounds__(TPB)
thin1x1_dgrad(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx, int64_t P, int CO) {
  constexpr int Q = CI / 4;
  __shared__ float wt[THIN_MAX * CI];                    // wt[k][ci] = w[ci][k]
  for (int u = threadIdx.x; u < THIN_MAX * CI; u += TPB) {
    const int k = u / CI, ci = u % CI;
    wt[u] = k < CO ? w[ci * CO + k] : 0.f;
  }
  __syncthreads();
  const int64_t total = P * Q;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
    const int c4 = (int)(i % Q);
    const int64_t px = i / Q;
    float g[THIN_MAX];
#pragma unroll
    for (int k = 0; k < THIN_MAX; ++k) g[k] = k < CO ? gy[px * CO + k] : 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < THIN_MAX; ++k) o += g[k] * *(const f32x4*)(wt + k * CI + c4 * 4);
    *(f32x4*)(gx + i * 4) = o;
  }
}

// partial[b][ci][8] = sum over the block's pixels of x[p][ci] * gy[p][k]
template <int CI>
__global__ void __launch_bounds__(TPB)
thin1x1_wgrad(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ part, int64_t P, int CO) {
  constexpr int Q = CI / 4;
  static_assert(Q <= 64 && (Q & (Q - 1)) == 0, "lanes sharing a channel group sit Q apart inside a wave");
  __shared__ float sm[4 * Q * 4 * THIN_MAX];
  f32x4 acc[THIN_MAX];
#pragma unroll
  for (int k = 0; k < THIN_MAX; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int64_t total = P * Q;
  const int64_t stride = (int64_t)gridDim.x * TPB;       // multiple of Q: a thread keeps its channel group
  auto step = [&](int64_t i) {
    const int64_t px = i / Q;
    const f32x4 xv = *(const f32x4*)(x + i * 4);
#pragma unroll
    for (int k = 0; k < THIN_MAX; ++k) {
      const float g = k < CO ? gy[px * CO + k] : 0.f;
      acc[k] += g * xv;
    }
  };
  int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
  for (; i + stride < total; i += 2 * stride) { step(i); step(i + stride); }
  for (; i < total; i += stride) step(i);
  // lanes Q apart share the channel group: xor-tree in the wave, then the 4 waves through LDS (fixed order)
  for (int off = Q; off < 64; off <<= 1) {
#pragma unroll
    for (int k = 0; k < THIN_MAX; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[k][j] += __shfl_xor(acc[k][j], off, 64);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane < Q) {
#pragma unroll
    for (int k = 0; k < THIN_MAX; ++k) *(f32x4*)(sm + ((wv * Q + lane) * THIN_MAX + k) * 4) = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < Q) {
    const int c4 = threadIdx.x;                          // = (blockIdx.x*TPB + tid) % Q since TPB % Q == 0
#pragma unroll
    for (int k = 0; k < THIN_MAX; ++k) {
      f32x4 t = *(const f32x4*)(sm + ((0 * Q + c4) * THIN_MAX + k) * 4);
#pragma unroll
      for (int w4 = 1; w4 < 4; ++w4) t += *(const f32x4*)(sm + ((w4 * Q + c4) * THIN_MAX + k) * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) part[((size_t)blockIdx.x * CI + c4 * 4 + j) * THIN_MAX + k] = t[j];
    }
  }
}

// gw[ci][co] = sum_b part[b][ci][co] (co < CO): one wave per (ci, 4 k's), 64 lanes stride the blocks, xor-tree.
__global__ void __launch_bounds__(TPB)
thin1x1_wsum(const float* __restrict__ part, float* __restrict__ gw, int blocks, int CI, int CO) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);   // float4 column of the [CI][8] slab
  if (col >= CI * THIN_MAX / 4) return;
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  for (int b = lane; b < blocks; b += 64) t += *(const f32x4*)(part + ((size_t)b * CI * THIN_MAX) + col * 4);
  for (int off = 1; off < 64; off <<= 1)
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] += __shfl_xor(t[j], off, 64);
  if (lane == 0) {
    const int ci = col / 2, k0 = (col & 1) * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + j < CO) gw[ci * CO + k0 + j] = t[j];
  }
}

constexpr int THIN_WGRAD_BLOCKS = 1024;

}  // namespace

extern "C" {

// eligibility of the streaming 1x1 kernels: K %4 == 0 (16-byte A fragments), K <= 512 (weights of one 32-channel
// slab fit the 64 KiB LDS image)
int smsut_conv1x1_supported(int Kdim, int Ndim) { return Kdim >= 4 && (Kdim % 4) == 0 && Kdim <= 512 && Ndim >= 1; }

// number of statistics tiles per image the forward emits for this problem (0: statistics not available, e.g. HW % 64)
int smsut_conv1x1_tiles(int N, int HW, int Ndim) {
  const int64_t P = (int64_t)N * HW;
  const int mr = (P / 256 >= 512) ? 4 : 1;
  (void)Ndim;
  return (HW % (16 * mr) == 0) ? HW / (16 * mr) : 0;
}

// y = x * W (+ optional InstanceNorm statistics partials [N][tiles][Ndim][2]); transposed = 1: data-gradient
static int conv1x1_fwd_launch(const float* x, const float* w, float* y, float* stats, int N, int HW, int Kdim, int Ndim,
                              int transposed, void* stream, float* y2, int split, const float* x2 = nullptr, int ca = 0) {
  SMSUT_REQUIRE(x && w && y && N > 0 && HW > 0 && smsut_conv1x1_supported(Kdim, Ndim));
  SMSUT_REQUIRE(!y2 || (!stats && split > 0 && split < Ndim && split % 16 == 0));
  SMSUT_REQUIRE(!x2 || (!transposed && ca > 0 && ca < Kdim && ca % 16 == 0 && (Kdim - ca) % 4 == 0));
  const int64_t P = (int64_t)N * HW;
  const int mr = (P / 256 >= 512) ? 4 : 1;
  SMSUT_REQUIRE(!stats || HW % (16 * mr) == 0);
  const int nr = Ndim <= 16 ? 1 : 2;
  const int chunks = (Kdim + 15) / 16;
  const size_t sh = (size_t)chunks * 4 * 16 * nr * 4 * sizeof(float);
  dim3 grid((unsigned)cdiv64(P, 64 * mr), (Ndim + 16 * nr - 1) / (16 * nr));
  hipStream_t st = (hipStream_t)stream;
#define LAUNCH1X1(M, R)                                                                                                      \
  do {                                                                                                                       \
    if (x2) conv1x1_fwd<M, R, true><<<grid, TPB, sh, st>>>(x, w, y, stats, P, HW, Kdim, Ndim, transposed, y2, split, x2, ca); \
    else conv1x1_fwd<M, R, false><<<grid, TPB, sh, st>>>(x, w, y, stats, P, HW, Kdim, Ndim, transposed, y2, split, nullptr, 0); \
  } while (0)
  if (mr == 4 && nr == 2) LAUNCH1X1(4, 2);
  else if (mr == 4) LAUNCH1X1(4, 1);
  else if (nr == 2) LAUNCH1X1(1, 2);
  else LAUNCH1X1(1, 1);
#undef LAUNCH1X1
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

int smsut_conv1x1_fwd(const float* x, const float* w, float* y, float* stats, int N, int HW, int Kdim, int Ndim,
                      int transposed, void* stream) {
  return conv1x1_fwd_launch(x, w, y, stats, N, HW, Kdim, Ndim, transposed, stream, nullptr, 0);
}

// y = cat([xa, xb]) * W without materialising the cat (xa [P][ca], xb [P][Kdim - ca], ca % 16 == 0); forward form only.
int smsut_conv1x1_fwd_cat(const float* xa, const float* xb, int ca, const float* w, float* y, float* stats, int N, int HW,
                          int Kdim, int Ndim, void* stream) {
  SMSUT_REQUIRE(xb);
  return conv1x1_fwd_launch(xa, w, y, stats, N, HW, Kdim, Ndim, 0, stream, nullptr, 0, xb, ca);
}

// Same product with the result channels [0, split) written to ya [P][split] and [split, Ndim) to yb [P][Ndim - split]
// (split % 16 == 0): the shortcut's data-gradient of a block fed by cat([up, skip]).
int smsut_conv1x1_fwd_split(const float* x, const float* w, float* ya, float* yb, int split, int N, int HW, int Kdim,
                            int Ndim, int transposed, void* stream) {
  SMSUT_REQUIRE(yb);
  return conv1x1_fwd_launch(x, w, ya, nullptr, N, HW, Kdim, Ndim, transposed, stream, yb, split);
}

// ConvTranspose2d(k=2, s=2, bias=False) through the 1x1 kernels' pixel-shuffle forms (see conv1x1_fwd / conv1x1_wgrad, PS):
// x [N,H,W,Cin], w [2][2][Cin][Cout], y / gy [N,2H,2W,Cout].  Cout % 16 == 0, Cin % 4 == 0.  SMSUT_CONVT_PS=0/1.
int smsut_convT2x2_ps_supported(int Cin, int Cout) {
  static const bool on = [] { const char* e = getenv("SMSUT_CONVT_PS"); return !e || atoi(e) != 0; }();
  // Cout == 16 only (the 128^2 -> 256^2 level): there the four taps are one 64-column slab -- 32x128^2 32->16: forward 66.4 ->
  // 62.1 us, weight gradient 91.4 -> 53.2 us; from 32 output channels on the per-tap MFMA kernels win (64->32: 38.6 / 42.2 and
  // 48.2 / 49.8 us; 128->64: 32.2 / 34.1 and 40.4 / 49.0) -- scratch/convt_probe.py.  The kernels take any Cout % 16 == 0.
  return on && Cout == 16 && smsut_conv1x1_supported(Cin, 4 * Cout);
}

int smsut_convT2x2_fwd_ps(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, void* stream) {
  SMSUT_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0 && smsut_convT2x2_ps_supported(Cin, Cout));
  const int64_t P = (int64_t)N * H * W;
  SMSUT_REQUIRE(P * 4 * Cout < (1ll << 31));
  const int Nd = 4 * Cout;
  const int mr = (P / 256 >= 512) ? 4 : 1;
  const int nr = (Nd % 64 == 0) ? 4 : 2;                   // 64 columns per workgroup when they divide: x is read Nd / 64 times
  const int chunks = (Cin + 15) / 16;
  const size_t sh = (size_t)chunks * 4 * 16 * nr * 4 * sizeof(float);
  dim3 grid((unsigned)cdiv64(P, 64 * mr), Nd / (16 * nr));
  hipStream_t st = (hipStream_t)stream;
#define PS_GO(M, R) conv1x1_fwd<M, R, false, true><<<grid, TPB, sh, st>>>(x, w, y, nullptr, P, H * W, Cin, Nd, 0, nullptr, 0, nullptr, 0, W)
  if (mr == 4 && nr == 4) PS_GO(4, 4);
  else if (mr == 4) PS_GO(4, 2);
  else if (nr == 4) PS_GO(1, 4);
  else PS_GO(1, 2);
#undef PS_GO
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

int64_t smsut_convT2x2_wgrad_ps_ws(int N, int H, int W, int Cin, int Cout) {
  const int cit = Cin > 16 ? 2 : 1;
  return (int64_t)plan_wgrad1((int64_t)N * H * W, Cin, 4 * Cout, cit, 2).splits * Cin * 4 * Cout;
}

// gw [2][2][Cin][Cout]
int smsut_convT2x2_wgrad_ps(const float* x, const float* gy, float* gw, float* workspace, int N, int H, int W, int Cin,
                            int Cout, void* stream) {
  SMSUT_REQUIRE(x && gy && gw && workspace && N > 0 && H > 0 && W > 0 && smsut_convT2x2_ps_supported(Cin, Cout));
  const int64_t P = (int64_t)N * H * W;
  SMSUT_REQUIRE(P * 4 * Cout < (1ll << 31));
  const int Nd = 4 * Cout;
  const int cit = Cin > 16 ? 2 : 1;
  const Plan1 p = plan_wgrad1(P, Cin, Nd, cit, 2);
  dim3 grid(p.splits, (Cin + 16 * cit - 1) / (16 * cit), Nd / 32);
  hipStream_t st = (hipStream_t)stream;
  if (cit == 2) conv1x1_wgrad<2, 2, true><<<grid, TPB, 0, st>>>(x, gy, workspace, P, Cin, Nd, p.pps, nullptr, 0, W);
  else conv1x1_wgrad<1, 2, true><<<grid, TPB, 0, st>>>(x, gy, workspace, P, Cin, Nd, p.pps, nullptr, 0, W);
  const int wsize = Cin * Nd;
  sum_parts<<<(wsize + 63) / 64, TPB, 0, st>>>(workspace, gw, wsize, p.splits);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

int64_t smsut_conv1x1_wgrad_ws(int N, int HW, int Cin, int Cout) {
  const int cit = Cin > 16 ? 2 : 1, cot = Cout > 16 ? 2 : 1;
  return (int64_t)plan_wgrad1((int64_t)N * HW, Cin, Cout, cit, cot).splits * Cin * Cout;
}

// gw [Cin][Cout] = sum_p x[p][:]^T gy[p][:]
static int conv1x1_wgrad_launch(const float* x, const float* gy, float* gw, float* workspace, int N, int HW, int Cin,
                                int Cout, void* stream, const float* x2, int ca) {
  SMSUT_REQUIRE(x && gy && gw && workspace && N > 0 && HW > 0 && Cin > 0 && Cout > 0);
  SMSUT_REQUIRE(!x2 || (ca > 0 && ca < Cin && ca % 16 == 0));
  const int64_t P = (int64_t)N * HW;
  const int cit = Cin > 16 ? 2 : 1, cot = Cout > 16 ? 2 : 1;
  const Plan1 p = plan_wgrad1(P, Cin, Cout, cit, cot);
  dim3 grid(p.splits, (Cin + 16 * cit - 1) / (16 * cit), (Cout + 16 * cot - 1) / (16 * cot));
  hipStream_t st = (hipStream_t)stream;
  if (cit == 2 && cot == 2) conv1x1_wgrad<2, 2><<<grid, TPB, 0, st>>>(x, gy, workspace, P, Cin, Cout, p.pps, x2, ca);
  else if (cit == 2) conv1x1_wgrad<2, 1><<<grid, TPB, 0, st>>>(x, gy, workspace, P, Cin, Cout, p.pps, x2, ca);
  else if (cot == 2) conv1x1_wgrad<1, 2><<<grid, TPB, 0, st>>>(x, gy, workspace, P, Cin, Cout, p.pps, x2, ca);
  else conv1x1_wgrad<1, 1><<<grid, TPB, 0, st>>>(x, gy, workspace, P, Cin, Cout, p.pps, x2, ca);
  const int wsize = Cin * Cout;
  sum_parts<<<(wsize + 63) / 64, TPB, 0, st>>>(workspace, gw, wsize, p.splits);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

int smsut_conv1x1_wgrad(const float* x, const float* gy, float* gw, float* workspace, int N, int HW, int Cin, int Cout,
                        void* stream) {
  return conv1x1_wgrad_launch(x, gy, gw, workspace, N, HW, Cin, Cout, stream, nullptr, 0);
}

// weight gradient with x = cat([xa, xb]) read in place (same workspace size as smsut_conv1x1_wgrad_ws)
int smsut_conv1x1_wgrad_cat(const float* xa, const float* xb, int ca, const float* gy, float* gw, float* workspace, int N,
                            int HW, int Cin, int Cout, void* stream) {
  SMSUT_REQUIRE(xb);
  return conv1x1_wgrad_launch(xa, gy, gw, workspace, N, HW, Cin, Cout, stream, xb, ca);
}

// ---- thin 1x1 layers (Cout <= 8, Cin in {8, 16, 32, 64}): data- and weight-gradient as streaming kernels
int smsut_conv1x1_thin_supported(int Cin, int Cout) {
  return (Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64) && Cout >= 1 && Cout <= THIN_MAX;
}

// gx [P][Cin] = gy [P][Cout] * W^T, W = [Cin][Cout] (HWIO memory of the forward weights)
int smsut_conv1x1_thin_dgrad(const float* gy, const float* w, float* gx, int N, int HW, int Cin, int Cout, void* stream) {
  SMSUT_REQUIRE(gy && w && gx && N > 0 && HW > 0 && smsut_conv1x1_thin_supported(Cin, Cout));
  const int64_t P = (int64_t)N * HW;
  hipStream_t st = (hipStream_t)stream;
  const int grid = ew_grid(P * (Cin / 4));
  switch (Cin) {
    case 8: thin1x1_dgrad<8><<<grid, TPB, 0, st>>>(gy, w, gx, P, Cout); break;
    case 16: thin1x1_dgrad<16><<<grid, TPB, 0, st>>>(gy, w, gx, P, Cout); break;
    case 32: thin1x1_dgrad<32><<<grid, TPB, 0, st>>>(gy, w, gx, P, Cout); break;
    default: thin1x1_dgrad<64><<<grid, TPB, 0, st>>>(gy, w, gx, P, Cout); break;
  }
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

// workspace floats for smsut_conv1x1_thin_wgrad
int64_t smsut_conv1x1_thin_wgrad_ws(int Cin) { return (int64_t)THIN_WGRAD_BLOCKS * Cin * THIN_MAX; }

// gw [Cin][Cout] = sum_p x[p][:]^T gy[p][:]
int smsut_conv1x1_thin_wgrad(const float* x, const float* gy, float* gw, float* workspace, int N, int HW, int Cin,
                             int Cout, void* stream) {
  SMSUT_REQUIRE(x && gy && gw && workspace && N > 0 && HW > 0 && smsut_conv1x1_thin_supported(Cin, Cout));
  const int64_t P = (int64_t)N * HW;
  hipStream_t st = (hipStream_t)stream;
  int blocks = (int)cdiv64(P * (Cin / 4), 2 * TPB);
  if (blocks > THIN_WGRAD_BLOCKS) blocks = THIN_WGRAD_BLOCKS;
  switch (Cin) {
    case 8: thin1x1_wgrad<8><<<blocks, TPB, 0, st>>>(x, gy, workspace, P, Cout); break;
    case 16: thin1x1_wgrad<16><<<blocks, TPB, 0, st>>>(x, gy, workspace, P, Cout); break;
    case 32: thin1x1_wgrad<32><<<blocks, TPB, 0, st>>>(x, gy, workspace, P, Cout); break;
    default: thin1x1_wgrad<64><<<blocks, TPB, 0, st>>>(x, gy, workspace, P, Cout); break;
  }
  thin1x1_wsum<<<(Cin * THIN_MAX / 4 + 3) / 4, TPB, 0, st>>>(workspace, gw, blocks, Cin, Cout);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

}  // extern "C"

