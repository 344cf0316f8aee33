// 3x3 weight gradient with REGISTER-RESIDENT ROWS (r04) -- reference op: the gradient of conv3x3 w.r.t. its weight inside
// BasicBlock (/root/reference/network/blocks.py:10-12, :53-80), i.e. gw[tap][ci][co] = sum_p x[p + tap][ci] * gy[p][co].
//
// The LDS-staged kernels of conv_mfma.hip (conv_mfma_wgrad / conv_mfma_wgrad_ts) pay a per-tile skeleton -- two barriers and a
// register -> LDS publish per 8x16-pixel tile -- plus a prologue and a slab epilogue that every workgroup of the single resident
// round runs at the same time; they sit at 50-70 % of the fp32 MFMA peak (profiles/r03_notes.md).  This kernel has NO LDS staging
// and NO barrier in its main loop: with the pixels as the MFMA K dimension, lane (lm, kq) of v_mfma_f32_16x16x4_f32 needs ONE
// channel (lm) of ONE pixel per k-slot, which is exactly what a global_load_dword hands it (16 lanes = 64 contiguous bytes of a
// pixel's NHWC channel row).  A wave owns a 16-pixel-wide column strip of RC image rows:
//   * lane (lm, kq) covers the four consecutive pixels 4kq .. 4kq+3 of the strip (MFMA ks takes pixel 4kq + ks), so the three
//     horizontal taps of its pixels are the SIX pixels 4kq-1 .. 4kq+4 of a row: six registers per 16-channel tile of x;
//   * the three rows a tap column needs live in a ring of R row slots, ONE new row per step; the loads of row t+2+D are issued
//     D steps ahead (software pipeline through the in-order vmcnt counter -- the loop is unrolled R times so every slot is a
//     fixed register), gy rows alike;
//   * per step and (ci tile, co tile): 36 MFMAs on 6 + 4 loaded registers.
// Image borders: addresses are clamped into the tensor and the slot is zeroed when the row first becomes the lower neighbour
// (uniform branch for rows, per-lane select for the strip's first / last pixel at the image edge).  The input-side
// InstanceNorm + LeakyReLU form (INAFF) applies in_affine() + lrelu at that same point, BEFORE the zeroing (the padding zeros
// belong to the activated tensor), with the same fma as the forward.
// Waves of a workgroup: WI x WJ sub-slabs x WS = 4 / (WI WJ) neighbouring strips; the WS waves of a sub-slab combine in a fixed
// order through LDS at the end; one partial slab per workgroup ([split][9 (+1)][Cin][Cout], summed by sum_splits as before).
#include "conv_wgrad_rr.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TPB = 256;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// order fence for the software pipeline: the empty volatile asm keeps the (read-only, otherwise freely movable) buffer loads on
// their side at the IR / DAG level, sched_barrier does the same for the machine scheduler
#define FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Buffer descriptor over [p, p + bytes), built from wave-uniform words only (the wave id is uniform but not provably so: without
// the readfirstlanes every buffer_load would sit in a waterfall loop).  One buffer_load_dword = descriptor (4 SGPRs) + per-lane
// byte offset (1 VGPR, + immediate) + scalar byte offset (the image row): no 64-bit address arithmetic per load.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const float* p, int bytes) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, uni(bytes), 0x00020000);
}
__device__ __forceinline__ float ldb(rsrc_t r, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

struct RrArgs {
  const float* x; const float* x2; int ca;      // x2 != null: x = virtual cat([x, x2]), ca channels in x
  const float* gy; const float* gs;             // gs != null: fused shortcut (slab row 9)
  float* part;
  int N, H, W, Cin, Cout;
  int RC;                                       // rows per wave-unit (H % RC == 0, RC % R == 0)
  int groups_per_split;                         // groups of WS wave-units a workgroup walks
  int total_wu;                                 // N * (H / RC) * (W / 16)
  RrAff aff;
  // PAIRED launch (smsut_conv2d_wgrad_pair): images n >= N1 belong to a SECOND set of tensors that went through the same conv
  // (another forward pass of the same layer) -- one launch, one slab set, both sets summed in the accumulators.  N1 == N: one set.
  int N1;
  const float* xB; const float* x2B; const float* gyB; const float* gsB; const float* meanB; const float* rstdB;
#ifdef SMSUT_STAMPS                             // diagnostic build only (scratch/rr_clock.py): per-workgroup cycle / real-time stamps
  unsigned long long* dbg;                      // [workgroups][4]: s_memtime, s_memrealtime at the start and at the end of wave 0
#endif
};

#ifndef RR_SPREAD
#define RR_SPREAD 1
#endif
#ifndef RR_V4R8
#define RR_V4R8 0      // 32 x 32 per wave: ring of 8 rows, 3 (2 with the fused shortcut) steps of loads in flight
#endif
#ifndef RR_OCC4
#define RR_OCC4 1       // workgroups per CU the 32 x 32-per-wave form must allow (2 = at most 256 registers)
#endif
template <int CIW, int COW, int WI, int WJ, int R, int D, bool DUAL, bool INAFF, bool SC>
__global__ void __launch_bounds__(TPB, (CIW == 2 && COW == 2) ? RR_OCC4 : 1) wgrad_rr(const RrArgs a) {
  static_assert(WI * WJ == 1 || WI * WJ == 2 || WI * WJ == 4, "sub-slabs per workgroup");
  static_assert(R >= D + 3, "ring: three live rows + D in flight");
  constexpr int WS = 4 / (WI * WJ);
  constexpr int NT = 9 * CIW * COW;             // accumulator tiles per wave
  constexpr int NTS = SC ? CIW * COW : 0;       // ... of the fused shortcut
  extern __shared__ float smem[];

  const int lane = threadIdx.x & 63;
  const int wave = uni((int)(threadIdx.x >> 6));
  const int lm = lane & 15, kq = lane >> 4;
  const int sub = wave % (WI * WJ), strip = wave / (WI * WJ);
  const int wi = sub % WI, wj = sub / WI;
  const int ci0 = (blockIdx.y * WI + wi) * 16 * CIW, co0 = (blockIdx.z * WJ + wj) * 16 * COW;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, RC = a.RC;
  const int NXS = W >> 4, NYC = H / RC;

#ifdef SMSUT_STAMPS
  const int wg_lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
  if (a.dbg && threadIdx.x == 0) {
    a.dbg[wg_lin * 4 + 0] = __builtin_amdgcn_s_memtime();
    a.dbg[wg_lin * 4 + 1] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  f32x4 acc[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  [[maybe_unused]] f32x4 acs[NTS > 0 ? NTS : 1];
  if constexpr (SC) {
#pragma unroll
    for (int k = 0; k < NTS; ++k) acs[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // per ci tile: which source tensor (first / second part of a virtual cat), its channel stride, the tile's first channel inside
  // it (uniform; the tensors themselves are picked per wave-unit: a paired launch has two sets)
  bool second[CIW];
  int cs[CIW], cf[CIW];
#pragma unroll
  for (int i = 0; i < CIW; ++i) {
    const int c = ci0 + 16 * i;
    if (DUAL && a.x2) {
      if (c < a.ca) { second[i] = false; cs[i] = a.ca; cf[i] = c; }
      else { second[i] = true; cs[i] = Cin - a.ca; cf[i] = c - a.ca; }
    } else { second[i] = false; cs[i] = Cin; cf[i] = c; }
  }

  constexpr int G = (R % (D + 1) == 0) ? D + 1 : R;        // gy ring: the row in use + D in flight
  float xr[R][CIW][6];
  float gr[G][COW][4];
  [[maybe_unused]] float sr[SC ? G : 1][COW][4];

  for (int g = 0; g < a.groups_per_split; ++g) {
    const int wu = uni((int)((blockIdx.x * a.groups_per_split + g) * WS + strip));
    if (wu >= a.total_wu) break;                                   // (uniform per wave; the combine below is outside the loop)
    const int xsn = wu % NXS, ycn = (wu / NXS) % NYC, nall = wu / (NXS * NYC);
    const int x0 = xsn << 4, y0 = ycn * RC;
    const bool setb = nall >= a.N1;                                // (uniform) second image set of a paired launch
    const int n = setb ? nall - a.N1 : nall;
    const float* const px = setb ? a.xB : a.x;
    const float* const px2 = setb ? a.x2B : a.x2;
    const float* const pgy = setb ? a.gyB : a.gy;
    const float* const pgs = setb ? a.gsB : a.gs;
    const bool lz = (x0 == 0) && kq == 0, rz = (x0 + 16 == W) && kq == 3;
    // Buffer descriptors of this image (wave-uniform inputs, so every load is ONE buffer_load_dword with the row in the scalar
    // offset); lane byte offsets inside a row of the source / of gy, the two edge pixels clamped into the row.
    rsrc_t sx[DUAL ? CIW : 1];
    int rowb[DUAL ? CIW : 1];                                      // bytes per image row of the source
#pragma unroll
    for (int i = 0; i < (DUAL ? CIW : 1); ++i) {
      sx[i] = make_rsrc((second[i] ? px2 : px) + (size_t)n * H * W * cs[i], H * W * cs[i] * 4);
      rowb[i] = uni(W * cs[i] * 4);
    }
    const rsrc_t sg = make_rsrc(pgy + (size_t)n * H * W * Cout, H * W * Cout * 4);
    [[maybe_unused]] const rsrc_t ss = make_rsrc(SC ? pgs + (size_t)n * H * W * Cout : pgy, H * W * Cout * 4);
    const int rowg = uni(W * Cout * 4);
    int vx[DUAL ? CIW : 1][6], vg[4];
#pragma unroll
    for (int i = 0; i < (DUAL ? CIW : 1); ++i)
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        int px = x0 + 4 * kq + e - 1;
        px = px < 0 ? 0 : (px >= W ? W - 1 : px);
        vx[i][e] = (px * cs[i] + cf[i] + lm) * 4;
      }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) vg[ks] = ((x0 + 4 * kq + ks) * Cout + co0 + lm) * 4;
    [[maybe_unused]] float am[CIW], ar[CIW], ag[CIW], ab[CIW];
    if constexpr (INAFF) {
      const float* const pm = setb ? a.meanB : a.aff.mean;
      const float* const pr = setb ? a.rstdB : a.aff.rstd;
#pragma unroll
      for (int i = 0; i < CIW; ++i) {
        const int c = ci0 + 16 * i + lm;
        am[i] = pm[(size_t)n * Cin + c]; ar[i] = pr[(size_t)n * Cin + c];
        ag[i] = a.aff.gamma[c]; ab[i] = a.aff.beta[c];
      }
    }
    const int ylast = min(y0 + RC, H - 1);                          // last row any load may touch

    auto load_x = [&](int slot, int q) {                            // relative row q = absolute row y0 - 1 + q
      int r = y0 - 1 + q;
      r = r < 0 ? 0 : (r > ylast ? ylast : r);
#pragma unroll
      for (int i = 0; i < CIW; ++i)
#pragma unroll
        for (int e = 0; e < 6; ++e)                                 // (plain source: ci tile i is 64 bytes further in the pixel's row)
          xr[slot][i][e] = DUAL ? ldb(sx[i], vx[i][e], r * rowb[i]) : ldb(sx[0], vx[0][e] + 64 * i, r * rowb[0]);
    };
    auto load_g = [&](int slot, int t) {
      const int r = min(y0 + t, y0 + RC - 1);
#pragma unroll
      for (int j = 0; j < COW; ++j)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) gr[slot][j][ks] = ldb(sg, vg[ks] + 64 * j, r * rowg);
      if constexpr (SC) {
#pragma unroll
        for (int j = 0; j < COW; ++j)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) sr[slot][j][ks] = ldb(ss, vg[ks] + 64 * j, r * rowg);
      }
    };
    auto fix_x = [&](int slot, int q) {                             // activation, then the zero padding
      const int r = y0 - 1 + q;
      const bool rowz = r < 0 || r >= H;                            // (uniform; selects, not a branch: the step stays one block)
#pragma unroll
      for (int i = 0; i < CIW; ++i)
#pragma unroll
        for (int e = 0; e < 6; ++e) {
          float v = xr[slot][i][e];
          if constexpr (INAFF) {
            v = lrelu_f(in_affine(v, am[i], ar[i], ag[i], ab[i]), a.aff.slope);
            asm volatile("" : "+v"(v));        // (keeps the activation out of a branch on the uniform part of z: 12 branches per step)
          }
          const bool z = rowz || (e == 0 && lz) || (e == 5 && rz);
          xr[slot][i][e] = z ? 0.f : v;
        }
    };

    // prologue, in the order of the loop's own requests: rows 0, 1, then (x row 2 + d, gy row d) for d < D
    // (fenced pair by pair: the compiler would sort them by address, and the counted waits of the loop's first steps -- merged
    //  with this order at the loop header -- would then wait for younger loads than they need)
    load_x(0, 0);
    FENCE();
    load_x(1, 1);
    FENCE();
#pragma unroll
    for (int d = 0; d < D; ++d) {
      load_x((2 + d) % R, 2 + d);
      load_g(d % G, d);
      FENCE();
    }
    fix_x(0, 0);
    fix_x(1, 1);

    for (int tb = 0; tb < RC; tb += R) {
#pragma unroll
      for (int s = 0; s < R; ++s) {
        const int t = tb + s;
#ifndef RR_ABL_NOLOAD                                               // (ablation builds, scratch/wgrad_rr_ab.py: results wrong by construction)
        load_x((s + 2 + D) % R, t + 2 + D);
        load_g((s + D) % G, t + D);
#endif
#if !RR_SPREAD
        FENCE();                                                    // (the scheduler would sink the loads to their uses)
#endif
        fix_x((s + 2) % R, t + 2);
        // taps of rows t, t+1 first: the row that is fixed up in this step (t+2) feeds the last third of the MFMAs only
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int tap = half ? 6 : 0; tap < (half ? 9 : 6); ++tap)
#pragma unroll
              for (int i = 0; i < CIW; ++i)
#pragma unroll
                for (int j = 0; j < COW; ++j)
                  acc[(tap * CIW + i) * COW + j] =
                      mfma16(xr[(s + tap / 3) % R][i][ks + tap % 3], gr[s % G][j][ks], acc[(tap * CIW + i) * COW + j]);
            if constexpr (SC) {
              if (half == 0) {
#pragma unroll
                for (int i = 0; i < CIW; ++i)
#pragma unroll
                  for (int j = 0; j < COW; ++j)
                    acs[i * COW + j] = mfma16(xr[(s + 1) % R][i][ks + 1], sr[s % G][j][ks], acs[i * COW + j]);
              }
            }
          }
#if RR_SPREAD
        // issue order inside the step (one scheduling region): the step's loads and the fix-up VALU spread under the MFMAs of
        // rows t, t+1 instead of a burst in front of them (one wave per SIMD has nobody else to fill the matrix pipe meanwhile)
        {
          constexpr int NLD = 6 * CIW + 4 * COW * (SC ? 2 : 1);
#pragma unroll
          for (int q = 0; q < NLD; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);        // MFMA
            __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);       // VMEM read
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x2, INAFF ? 4 : 1, 0);   // VALU (fix-up)
          }
        }
#endif
        FENCE();
      }
    }
  }

#ifdef SMSUT_STAMPS
  if (a.dbg && threadIdx.x == 0) {
    a.dbg[wg_lin * 4 + 2] = __builtin_amdgcn_s_memtime();
    a.dbg[wg_lin * 4 + 3] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  // ---- epilogue: the WS strip waves of each sub-slab combine in a fixed order -- (w0 + w2) + (w1 + w3) for WS = 4, w0 + w1 for
  // WS = 2 -- and ALL 256 threads store the slab as 16-byte rows (one wave storing 36 tiles dword by dword was ~4 us of tail).
  constexpr int NTA = NT + NTS;
  constexpr int CIG = 16 * CIW * WI, COG = 16 * COW * WJ;        // the workgroup's slab
  constexpr int LROW = COG + 4;                                   // LDS row stride: rows 4kq apart land on disjoint bank groups
  constexpr int ROWS = (SC ? 10 : 9) * CIG;                       // slab rows [tap (9 = shortcut)][ci]
  constexpr int P = WS >= 2 ? 2 : 1;                              // partial sums left for the last stage
  if constexpr (WS == 4) {                                        // strips 2, 3 hand their tiles to strips 0, 1 (accumulator layout)
    if (strip >= 2) {
      float* dst = smem + ((size_t)((strip - 2) * WI * WJ + sub) * NTA) * 256;
#pragma unroll
      for (int k = 0; k < NT; ++k) *(f32x4*)(dst + ((size_t)k * 64 + lane) * 4) = acc[k];
      if constexpr (SC) {
#pragma unroll
        for (int k = 0; k < NTS; ++k) *(f32x4*)(dst + ((size_t)(NT + k) * 64 + lane) * 4) = acs[k];
      }
    }
    __syncthreads();
    if (strip < 2) {
      const float* src = smem + ((size_t)(strip * WI * WJ + sub) * NTA) * 256;
#pragma unroll
      for (int k = 0; k < NT; ++k) acc[k] += *(const f32x4*)(src + ((size_t)k * 64 + lane) * 4);
      if constexpr (SC) {
#pragma unroll
        for (int k = 0; k < NTS; ++k) acs[k] += *(const f32x4*)(src + ((size_t)(NT + k) * 64 + lane) * 4);
      }
    }
    __syncthreads();
  }
  if (strip < P) {                                                // row-major slab image [strip][tap][ci][co (+ pad)]
    float* dst = smem + (size_t)strip * ROWS * LROW + (size_t)(wi * 16 * CIW + 4 * kq) * LROW + wj * 16 * COW + lm;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int i = 0; i < CIW; ++i)
#pragma unroll
        for (int j = 0; j < COW; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dst[(size_t)(tap * CIG + 16 * i + r) * LROW + 16 * j] = acc[(tap * CIW + i) * COW + j][r];
    if constexpr (SC) {
#pragma unroll
      for (int i = 0; i < CIW; ++i)
#pragma unroll
        for (int j = 0; j < COW; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[(size_t)(9 * CIG + 16 * i + r) * LROW + 16 * j] = acs[i * COW + j][r];
    }
  }
  __syncthreads();
  {
    float* out = a.part + (size_t)blockIdx.x * (SC ? 10 : 9) * Cin * Cout + (size_t)(blockIdx.y * CIG) * Cout + blockIdx.z * COG;
    constexpr int Q = COG / 4;                                    // 16-byte units per slab row
#pragma unroll
    for (int f0 = 0; f0 < ROWS * Q; f0 += TPB) {
      const int f = f0 + threadIdx.x;
      if (ROWS * Q % TPB == 0 || f < ROWS * Q) {
        const int row = f / Q, q = f % Q;
        f32x4 v = *(const f32x4*)(smem + (size_t)row * LROW + 4 * q);
        if constexpr (P == 2) v += *(const f32x4*)(smem + (size_t)(ROWS + row) * LROW + 4 * q);
        const int tap = row / CIG, cil = row % CIG;
        *(f32x4*)(out + ((size_t)tap * Cin + cil) * Cout + 4 * q) = v;
      }
    }
  }
}

// ---- plan ---------------------------------------------------------------------------------------------------------------------
struct RrPlan {
  int variant;            // 0 = not covered; 1: 16x16 per wave, 4 strips; 2: 32x16; 3: 16x32; 4: 32x32 per wave, 4 strips;
                          // 5: 16x16 per wave, 2x2 sub-slabs, 1 strip; 6: 32x16 per wave, 1x2 sub-slabs, 2 strips
  int slab_ci, slab_co;   // channels per workgroup slab
  int ws;                 // strips per workgroup
  int R;                  // ring size = row-loop unroll
  int RC, groups_per_split, splits, total_wu;
};

int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

RrPlan plan_rr(int N, int H, int W, int Cin, int Cout, const float* x2, int ca, bool aff, bool sc) {
  RrPlan p{};
  static const int on = env_int("SMSUT_WGRAD_RR", 1);
  if (!on || N <= 0 || H < 4 || W < 16 || (W % 16) || (H % 4) || (Cin % 16) || (Cout % 16)) return p;
  if ((int64_t)H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 29)) return p;       // byte offsets inside an image are 32-bit
  if (x2 && (ca <= 0 || ca >= Cin || (ca % 16))) return p;
  // variant for the 32 x 32 slabs: 4, 5 or 6 (tuning hook; anything else falls back to 4, so that eligible() never promises a
  // plan launch() has no kernel for -- ADVICE r04)
  static const int v32 = [] { const int v = env_int("SMSUT_RR_V32", 4); return (v == 4 || v == 5 || v == 6) ? v : 4; }();
  if (Cin == 16 && Cout == 16) { p.variant = 1; p.slab_ci = 16; p.slab_co = 16; p.ws = 4; }
  else if (Cin == 32 && Cout == 16) { p.variant = 2; p.slab_ci = 32; p.slab_co = 16; p.ws = 4; }
  else if (Cin == 16 && Cout == 32) { p.variant = 3; p.slab_ci = 16; p.slab_co = 32; p.ws = 4; }
  else if (Cin % 32 == 0 && Cout % 32 == 0) {
    const int v = v32;
    p.variant = v; p.slab_ci = 32; p.slab_co = 32; p.ws = v == 4 ? 4 : (v == 5 ? 1 : 2);
  } else return p;
  if (aff && x2) return RrPlan{};
  p.R = (p.variant == 4 && !RR_V4R8) ? 4 : 8;
  if (H % p.R) {
    if (H % 4) return RrPlan{};
    p.R = 4;
  }
  const int slabs = (Cin / p.slab_ci) * (Cout / p.slab_co);
  static const int t_small = env_int("SMSUT_RR_TARGET", 512), t_big = env_int("SMSUT_RR_TARGET4", 256);
  const int target = p.variant == 4 ? t_big : t_small;
  int want = (target + slabs - 1) / slabs;
  if (want < 1) want = 1;
  static const int rc_max = env_int("SMSUT_RR_RCMAX", 64);      // (64-row units where the grid still fills: B32 256^2 16->16 83 -> 78 us)
  int rc = p.R;
  for (int c = rc_max; c >= p.R; c >>= 1) {
    if (c % p.R || H % c) continue;
    const int64_t groups = ((int64_t)N * (H / c) * (W / 16) + p.ws - 1) / p.ws;
    if (groups >= want || c == p.R) { rc = c; break; }
  }
  p.RC = rc;
  p.total_wu = N * (H / rc) * (W / 16);
  const int groups = (p.total_wu + p.ws - 1) / p.ws;
  if (want > groups) want = groups;
  // keep the slabs sum_splits re-reads bounded (as plan_wgrad: 8 M floats)
  const int64_t wsz = (int64_t)Cin * Cout * (sc ? 10 : 9);
  int cap = (int)(((int64_t)8 << 20) / wsz);
  if (cap < 1) cap = 1;
  if (want > cap) want = cap;
  p.groups_per_split = (groups + want - 1) / want;
  p.splits = (groups + p.groups_per_split - 1) / p.groups_per_split;
  return p;
}

template <int CIW, int COW, int WI, int WJ, int R, int D>
int launch_v(const RrArgs& a, const RrPlan& p, hipStream_t st) {
  constexpr int WS = 4 / (WI * WJ);
  const bool sc = a.gs != nullptr, aff = a.aff.mean != nullptr, dual = a.x2 != nullptr;
  constexpr int NTA_ = (9 + 1) * CIW * COW;                       // (with the shortcut's tiles: an upper bound without)
  constexpr size_t sh1 = WS == 4 ? (size_t)2 * WI * WJ * NTA_ * 256 * sizeof(float) : 0;
  constexpr size_t sh2 = (size_t)(WS >= 2 ? 2 : 1) * 10 * (16 * CIW * WI) * (16 * COW * WJ + 4) * sizeof(float);
  const size_t sh = sh1 > sh2 ? sh1 : sh2;
  dim3 grid(p.splits, a.Cin / p.slab_ci, a.Cout / p.slab_co);
#define RR_GO(DUAL, INAFF, SC)                                                                                          \
  do {                                                                                                                  \
    auto kfn = wgrad_rr<CIW, COW, WI, WJ, R, D, DUAL, INAFF, SC>;                                                     \
    if (sh > 48 * 1024) (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh); \
    kfn<<<grid, TPB, sh, st>>>(a);                                                                                      \
    return 0;                                                                                                           \
  } while (0)
  if (sc) { if (dual) RR_GO(true, false, true); else RR_GO(false, false, true); }
  if (aff) RR_GO(false, true, false);
  if (dual) RR_GO(true, false, false);
  RR_GO(false, false, false);
#undef RR_GO
}

}  // namespace

#ifdef SMSUT_STAMPS
static unsigned long long* g_rr_dbg = nullptr;
extern "C" void smsut_dbg_rr_stamps(unsigned long long* p) { g_rr_dbg = p; }
#endif

bool smsut_wgrad_rr_eligible(int N, int H, int W, int Cin, int Cout, const float* x2, int ca, bool aff, bool sc) {
  return plan_rr(N, H, W, Cin, Cout, x2, ca, aff, sc).variant != 0;
}
int smsut_wgrad_rr_splits(int N, int H, int W, int Cin, int Cout, const float* x2, int ca, bool aff, bool sc) {
  const RrPlan p = plan_rr(N, H, W, Cin, Cout, x2, ca, aff, sc);
  return p.variant ? p.splits : 0;
}

int smsut_wgrad_rr_launch(const float* x, const float* x2, int ca, const float* gy, const float* gs, float* part, int N, int H,
                          int W, int Cin, int Cout, const RrAff* aff, hipStream_t st, const RrSetB* b) {
  const RrPlan p = plan_rr(N, H, W, Cin, Cout, x2, ca, aff != nullptr, gs != nullptr);
  if (!p.variant) return -1;
  if (b && (b->n <= 0 || b->n >= N || !b->x || !b->gy || (x2 != nullptr) != (b->x2 != nullptr) || (gs != nullptr) != (b->gs != nullptr) ||
            (aff != nullptr) != (b->mean != nullptr && b->rstd != nullptr)))
    return -1;
  RrArgs a{};
  a.x = x; a.x2 = x2; a.ca = ca; a.gy = gy; a.gs = gs; a.part = part;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout;
  a.RC = p.RC; a.groups_per_split = p.groups_per_split; a.total_wu = p.total_wu;
  if (aff) a.aff = *aff;
  a.N1 = N;
  if (b) {
    a.N1 = N - b->n;
    a.xB = b->x; a.x2B = b->x2; a.gyB = b->gy; a.gsB = b->gs; a.meanB = b->mean; a.rstdB = b->rstd;
  }
#ifdef SMSUT_STAMPS
  a.dbg = g_rr_dbg;
#endif
  switch (p.variant) {
    case 1: return p.R == 8 ? launch_v<1, 1, 1, 1, 8, 3>(a, p, st) : launch_v<1, 1, 1, 1, 4, 1>(a, p, st);
    case 2: return p.R == 8 ? launch_v<2, 1, 1, 1, 8, 3>(a, p, st) : launch_v<2, 1, 1, 1, 4, 1>(a, p, st);
    case 3: return p.R == 8 ? launch_v<1, 2, 1, 1, 8, 3>(a, p, st) : launch_v<1, 2, 1, 1, 4, 1>(a, p, st);
#if RR_V4R8
    case 4: return p.R == 8 ? (gs ? launch_v<2, 2, 1, 1, 8, 2>(a, p, st) : launch_v<2, 2, 1, 1, 8, 3>(a, p, st))
                            : launch_v<2, 2, 1, 1, 4, 1>(a, p, st);
#else
    case 4: return launch_v<2, 2, 1, 1, 4, 1>(a, p, st);
#endif
    case 5: return p.R == 8 ? launch_v<1, 1, 2, 2, 8, 3>(a, p, st) : launch_v<1, 1, 2, 2, 4, 1>(a, p, st);
    case 6: return p.R == 8 ? launch_v<2, 1, 1, 2, 8, 3>(a, p, st) : launch_v<2, 1, 1, 2, 4, 1>(a, p, st);
  }
  return -1;
}
