// Winograd F(2x2, 3x3) convolution for LARGE reductions (Kdim >= 64) on the gfx950 matrix cores, fp32.
//
// conv_mfma.hip's Winograd form keeps the whole transformed weight block U = G g G^T resident in LDS, which stops at 32
// reduction channels (16 positions x Kdim x 16 output channels x 4 B).  The deep levels of the U-Net / ugan generator (64 ... 256
// channels on 64^2 ... 16^2 planes, network/blocks.py:120-174, ugan.py:22-83) carry 60 % of the 3x3 FLOPs; this kernel takes
// them with a different economy:
//
//   * an item is a 16x16-pixel output tile of one image for 16*NTN output channels; its reduction is walked in 16-channel
//     chunks, and per chunk BOTH operands are staged into one of two LDS buffers, by LDS-DMA (global_load_lds_dwordx4: no
//     staging registers, no ds_write): the haloed input tile, and the chunk's transformed weights from a PREPARED image when the
//     caller bound one (smsut_wino_prepare / smsut_wino_bind_many: U = G g G^T of the whole tensor, written once per phase in
//     this kernel's LDS order).  Without a binding every thread transforms NTN (reduction channel, output channel) pairs ON THE
//     FLY -- nine weights prefetched one chunk ahead, 28 flops, sixteen LDS words -- so the plain C ABI needs no workspace; the
//     input-side-IN form keeps register staging for its input (the values pass through the ALU);
//   * NTN = 2 output-channel slabs per wave share one input transform (B^T d B, in registers, lane = (tile, channel quad)):
//     128 position accumulators per lane, ONE wave per SIMD with the 512-register budget, and the latency hiding that a second
//     wave would give is written into the instruction stream instead -- the B-fragment reads and the transform arithmetic of
//     unit u+1 sit inside the 16-MFMA block of unit u, and a chunk is ROTATED around its barrier: the next chunk's first reads
//     go out under this chunk's last unit (mma_chunk);
//   * an item is long (>= 4 chunks x 128 MFMAs per wave), so its results are transformed (A^T m A), folded and stored right after
//     its last chunk -- no deferred epilogue, no second accumulator set (its statistics follow the next barrier).
//
// Fused forms, same contracts as conv_mfma_fwd_p (the entry points of conv_mfma.hip route here by shape):
//   STATS   per-tile {sum, sum of squares} of the result for the following InstanceNorm
//   ACC     result added to what y holds (second gradient path into a block input)
//   BST     conv2's data-gradient inside a BasicBlock: result x LeakyReLU mask recomputed from y1, InstanceNorm-backward partials
//   DUAL    input = virtual cat([x, x2]) of two Kdim/2-channel tensors
//   INAFF   input-side InstanceNorm + LeakyReLU applied while staging (the padding stays zero)
//   SC      fused 1x1 shortcut conv (forward): four more MFMA sets per chunk on the tile's raw pixels
//   SC2     fused shortcut data-gradient: the second reduction half is the shortcut's gradient against the 1x1 weights
//   y2      split output (channels [0, split) to y, the rest to y2)
// Lane <-> pixel map of the results: acc[i = 2*dy + dx][j][r] = tile 4*kq + r of the wave's 4-row strip, pixel (dy, dx) of it:
// strip row 2*(kq >> 1) + dy, column 8*(kq & 1) + 2*r + dx.
#include "conv_wino.h"
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TPB = 256;
constexpr int TH = 16, TW = 16;           // output tile of an item
constexpr int IH = TH + 2, IW = TW + 2;   // haloed input tile
constexpr int SPX = 20;                   // floats between pixels of the staged input tile (see SPIXW in conv_mfma.hip)
constexpr int UNITS = IH * IW * 4;        // float4 units of one 16-channel input chunk
constexpr int NI_R = (UNITS + TPB - 1) / TPB;
// LDS-DMA staging of the input tile (every form but INAFF, whose values pass through the ALU): `global_load_lds_dwordx4` writes
// wave-uniform base + lane * 16 B, so the tile image is cut into 16-byte SLOTS in LDS order -- five per pixel (four channel quads
// + the pad quad of the 20-float pixel stride) -- and lane l of wave-instruction q fills slot 64 q + l from its own source
// address: the pixel's channels, or a 16-byte block of zeros (padding pixels, pad quads, the slots past the tile)
constexpr int GSLOTS = IH * IW * 5;
constexpr int NI_G = (GSLOTS + TPB - 1) / TPB;
#ifndef SMSUT_WINO_GLDS
#define SMSUT_WINO_GLDS 1
#endif
__device__ __attribute__((aligned(16))) const float wino_zero16[4] = {0.f, 0.f, 0.f, 0.f};
typedef __attribute__((address_space(3))) void* lds_vp;
// The copies are issued from inline assembly: with the builtin (__builtin_amdgcn_global_load_lds) hipcc books the DMA as a second
// kind of pending LDS event and from the first one on turns EVERY `s_waitcnt lgkmcnt(n)` of the region into lgkmcnt(0) -- each
// fragment read then waits for the reads issued after it as well, six exposed LDS latencies per chunk with one wave per SIMD.
// The hardware counts an LDS-DMA on vmcnt only, so the partial waits are right; what the compiler no longer sees is waited for
// by hand (glds_wait before the barrier that publishes the buffer).  M0 = LDS byte address of the wave's 1 KiB piece, written
// in the statement that uses it and not restored: nothing else in these kernels reads M0 (hipcc only touches it for its own
// LDS-DMA builtin, indirect register moves, GWS and message instructions -- checked in the ISA of every instantiation by
// tests/test_cabi_cpu.py::test_wino_kernels_leave_m0_to_the_dma_statements).
__device__ __forceinline__ void glds16(const float* src, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_byte_addr) : "memory");
}
// ... with a uniform 64-bit base and a 32-bit byte offset per lane (no vector address arithmetic)
__device__ __forceinline__ void glds16_so(const float* base, unsigned byte_off, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"(byte_off), "s"(lds_byte_addr), "s"(base) : "memory");
}
#ifdef SMSUT_WL_STAMPS               // scratch builds: wave 0 of workgroup (0, 0) sums the cycles of the phases of its regions into y[0..7]
#define WL_STAMP(t)                                                                         \
  do {                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                      \
  } while (0)
#else
#define WL_STAMP(t) do { } while (0)
#endif
__device__ __forceinline__ void glds_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// compile-time loop: f(integral_constant<int, I>) for I in [I, N) -- the unit index of conv_wino_l's chunk loop must be a constant in
// EVERY instantiation (as a `#pragma unroll` loop the larger bodies were left rolled, and the accumulators went to scratch memory)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float aff1(float v, float m, float r, float g, float b, float slope) {
  return lrelu_f(in_affine(v, m, r, g, b), slope);
}

// LDS: TWO staging buffers {input tile, U chunk, (SC) 1x1 chunk} -- chunk t+1 is published while chunk t multiplies, one
// barrier per chunk -- plus the statistics scratch
constexpr int wino_l_in_floats(bool gli) { return gli ? NI_G * TPB * 4 : (IH * IW + 1) * SPX; }
template <int NTN>
constexpr int wino_l_buf_floats(bool sc, bool gli) { return wino_l_in_floats(gli) + 16 * 16 * 16 * NTN + (sc ? 16 * 16 * NTN : 0); }
template <int NTN>
constexpr size_t wino_l_lds(bool sc, bool gli) {
  return (size_t)(2 * wino_l_buf_floats<NTN>(sc, gli) + (4 * 16 * NTN * 2 + 8) * (sc ? 2 : 1)) * sizeof(float);
}

template <int NTN, bool STATS, bool ACC, bool BST, bool DUAL, bool INAFF, bool SC, bool SC2, bool PRE, bool FIN = false>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(1, 1)))
conv_wino_l(const float* __restrict__ x, const float* __restrict__ x2, const float* __restrict__ w, const float* __restrict__ wu,
            float* __restrict__ y,
            float* __restrict__ y2, int split, int N, int H, int W, int nch, int Ndim, int tiles_x, int tiles_img, int items_per_wg,
            int transposed, float* __restrict__ stats, WinoBst bst, WinoAff aff, WinoSc sc, FinRef fin) {
  static_assert(!(BST && (STATS || ACC)), "BST excludes the forward statistics and the accumulate form");
  static_assert(!FIN || STATS || BST, "in-launch finalize (common.h): statistics / BST forms");
  static_assert(!INAFF || (STATS && !ACC && !BST && !DUAL), "input-side IN: forward statistics form only");
  static_assert(!SC || (STATS && !ACC && !BST && !INAFF && !SC2), "fused shortcut: forward statistics forms");
  static_assert(!SC2 || (DUAL && !STATS && !ACC && !BST && !INAFF && !SC), "fused shortcut data-gradient");
  static_assert(!(PRE && SC2), "prepared weights: not for the fused shortcut data-gradient (two weight tensors in one reduction)");
  constexpr int CO_T = 16 * NTN, NR = NTN;
  constexpr bool GLI = !INAFF && SMSUT_WINO_GLDS != 0; // input tile by LDS-DMA
  constexpr int NI = GLI ? NI_G : NI_R;
  extern __shared__ float smem[];
  constexpr int BUF = wino_l_buf_floats<NTN>(SC, GLI); // floats per staging buffer
  constexpr int IN_F = wino_l_in_floats(GLI);          // [IH][IW][SPX] + one dummy pixel (sink of the padding units) | whole DMA slots
  constexpr int W_F = 16 * 16 * CO_T;                  // a chunk's U: [16 positions][4 channel quads][CO_T][4]
  float* red = smem + 2 * BUF;                         // [4 waves][CO_T][2] + dummy
  [[maybe_unused]] float* red_sc = red + 4 * CO_T * 2 + 8;
  auto in_b = [&](int b) { return smem + b * BUF; };
  auto w_b = [&](int b) { return smem + b * BUF + IN_F; };
  [[maybe_unused]] auto wsc_b = [&](int b) { return smem + b * BUF + IN_F + W_F; };   // SC: a chunk's 1x1 weights [4][CO_T][4]

#ifdef SMSUT_WL_STAMPS
  unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, ts5 = 0, ts_begin, acc_t[6] = {0, 0, 0, 0, 0, 0};
  WL_STAMP(ts_begin);
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, kq = lane >> 4;
  const int co0 = blockIdx.y * CO_T;
  const int Kd = 16 * nch;                              // reduction channels
  const int nh = nch >> 1;                              // DUAL: chunks per half
  const int KST = DUAL ? Kd / 2 : Kd;                   // pixel stride of the tensor(s) the input is read from
  const bool tr = SC2 || (transposed & 1);
  const int KROW = SC2 ? Kd / 2 : Kd;                   // row length of the transposed 3x3 weights
  float* const yo = (y2 && co0 >= split) ? y2 : y;
  const int os = !y2 ? Ndim : (co0 >= split ? Ndim - split : split);
  const int oc0 = (y2 && co0 >= split) ? co0 - split : co0;
  const int total_items = N * tiles_img;
  // XCD k gets the k-th contiguous eighth of the items (halo rows shared between neighbouring strips hit its own L2)
  const int wg = (gridDim.x & 7) == 0 ? (int)((blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int item0 = wg * items_per_wg;
  const int item1 = min(item0 + items_per_wg, total_items);
  if (item0 >= item1) return;
  const int tiles_y = tiles_img / tiles_x;

  // ---- input staging descriptors (tile-independent)
  int u_off[NI], u_flag[NI];
  [[maybe_unused]] int u_lds[NI];
  [[maybe_unused]] float4 rin[GLI ? 1 : NI];
  [[maybe_unused]] bool zero[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int u = tid + i * TPB;
    const bool real = GLI ? (u < GSLOTS && u % 5 < 4) : u < UNITS;
    const int uu = real ? u : 0;
    const int q = GLI ? uu % 5 : uu & 3, pix = GLI ? uu / 5 : uu >> 2;
    const int iy = pix / IW, ix = pix % IW;
    u_off[i] = (iy * W + ix) * KST + 4 * q;
    u_lds[i] = real ? pix * SPX + 4 * q : IH * IW * SPX;
    u_flag[i] = (iy < 1 ? 1 : 0) | (iy >= TH + 1 ? 2 : 0) | (ix < 1 ? 4 : 0) | (ix >= TW + 1 ? 8 : 0) | (real ? 0 : 16);
  }
  [[maybe_unused]] const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // LDS byte address of this wave's 1 KiB piece 0 of staging buffer 0 (uniform; pieces and buffers are constant steps from it)
  [[maybe_unused]] const unsigned la0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_vp)smem) + (unsigned)wave_u * 1024u;
  [[maybe_unused]] const float* xb1 = x;               // GLI: source base / border flags of the chunk one region ahead
  [[maybe_unused]] int fl1 = 0;
  [[maybe_unused]] const int safe_off = (W + 1) * KST;                  // first interior pixel of the tile: always inside the image

  // ---- weight staging: NTN (reduction channel ci, output channel n) pairs per thread.  Lanes run along the contiguous
  //      dimension of the weight tensor: n for [tap][ci][n] (forward), ci for [tap][n][ci] (transposed)
  int w_ci[NTN], w_n[NTN];
  [[maybe_unused]] unsigned w_off[NTN];                // element offset of the pair inside a (tap, chunk) block of the weight tensor
  [[maybe_unused]] unsigned w_off1[NTN];               // ... inside the 1x1 weights (SC: [ci][n]; SC2: [n][ci])
  [[maybe_unused]] float wreg[PRE ? 1 : NTN][9];
  [[maybe_unused]] float wx[NTN];
  // PRE: the chunk's transformed weights are copied by LDS-DMA from the image smsut_wino_prepare wrote,
  // [chunk][16-channel slab][position][channel quad][16][4]: 16-byte slot i * 256 + tid of the LDS block [pos][quad][CO_T][4]
  constexpr int NWG = PRE ? 4 * NTN : 0;               // DMA parts per thread and chunk
  constexpr int NWR = (!PRE || SC) ? NTN : 0;          // register parts (on-the-fly transform and / or the 1x1 shortcut weights)
  [[maybe_unused]] unsigned w_goff[PRE ? NWG : 1];
  [[maybe_unused]] const float* wu1 = wu;
  if constexpr (PRE) {
#pragma unroll
    for (int i = 0; i < NWG; ++i) {
      const int idx = i * TPB + tid, n = idx % CO_T, pk = idx / CO_T;
      w_goff[i] = (unsigned)((n >> 4) * 4096 + (pk * 16 + (n & 15)) * 4) * 4u;           // bytes
    }
  }
#pragma unroll
  for (int k = 0; k < NTN; ++k) {
    const int p = tid + k * TPB;
    if (tr) { w_ci[k] = p & 15; w_n[k] = p >> 4; }
    else { w_n[k] = p % CO_T; w_ci[k] = p / CO_T; }
    w_off[k] = tr ? (unsigned)((co0 + w_n[k]) * KROW + w_ci[k]) : (unsigned)(w_ci[k] * Ndim + co0 + w_n[k]);
    if constexpr (SC) w_off1[k] = (unsigned)(w_ci[k] * Ndim + co0 + w_n[k]);
    if constexpr (SC2) w_off1[k] = (unsigned)((co0 + w_n[k]) * KROW + w_ci[k]);
  }
  // (tap, chunk) blocks: forward w + (tap * Kd + 16 c) * Ndim, transposed w + (8 - tap) * Ndim * KROW + 16 c -- uniform, so a
  // weight load is one scalar base + the thread's 32-bit offset (no per-load 64-bit vector arithmetic); the selects between the
  // two layouts are hoisted out of the chunk loop (inside it they became branches that cut the MFMA units apart)
  const ptrdiff_t w_tstep = tr ? -(ptrdiff_t)Ndim * KROW : (ptrdiff_t)Kd * Ndim;       // tap -> tap + 1
  const ptrdiff_t w_cstep = tr ? 16 : (ptrdiff_t)16 * Ndim;                            // chunk -> chunk + 1
  const float* const w_tap0 = tr ? w + (size_t)8 * Ndim * KROW : w;                    // tap 0, chunk 0

  int pn = item0 / tiles_img, pty, ptx;                // item of the NEXT prefetch ...
  { const int t = item0 - pn * tiles_img; pty = t / tiles_x; ptx = t - pty * tiles_x; }
  int pch = 0;                                         // ... and its chunk
  int cn = pn, cty = pty, ctx = ptx;                   // item being computed
  int pubc = 0;                                        // chunk whose operands the staging registers hold (to be published)
  int pfc = 0, pfl = 0;                                // chunk being prefetched, borders its tile touches
  const float* xb = x;                                 // its input base (tile's first halo pixel, chunk's first channel)
  int left = (item1 - item0) * nch;                    // chunks not yet requested
  [[maybe_unused]] float4 a_m, a_r, a_g, a_b;

  auto advance = [&](int& n_, int& ty_, int& tx_) {
    if (++tx_ == tiles_x) { tx_ = 0; if (++ty_ == tiles_y) { ty_ = 0; ++n_; } }
  };
  // The staging work of a region is cut into NP = NI + NTN parts (an input unit or a weight pair each) so that it can be spread
  // over the MFMA units of the chunk that multiplies meanwhile: part p first PUBLISHES what its registers hold (operands of the
  // next chunk, requested a whole region ago) into the other LDS buffer, then REQUESTS the chunk after that into the same
  // registers.  Everything is unconditional straight-line code (a branch would cut the scheduling region): past the workgroup's
  // last chunk the cursor stops and the last chunk is simply requested again.
  auto pf_setup = [&]() {                              // uniform: where the next request reads from
    xb1 = xb; fl1 = pfl;
    pubc = pfc;
    if constexpr (PRE) wu1 = wu + ((size_t)pubc * (Ndim >> 4) + (co0 >> 4)) * 4096;
    pfc = pch;
    pfl = (pty == 0 ? 1 : 0) | (pty == tiles_y - 1 ? 2 : 0) | (ptx == 0 ? 4 : 0) | (ptx == tiles_x - 1 ? 8 : 0);
    const int cl = (DUAL && pch >= nh) ? pch - nh : pch;
    const int base = (((pn * H + pty * TH - 1) * W) + ptx * TW - 1) * KST + cl * 16;
    xb = ((DUAL && pch >= nh) ? x2 : x) + base;
  };
  auto pf_advance = [&]() {                            // after the region's requests have been issued
    if (left > 1) {
      --left;
      if (++pch == nch) { pch = 0; advance(pn, pty, ptx); }
    }
  };
  auto pf_in = [&](int i) {
    if constexpr (!GLI) {
      zero[i] = (u_flag[i] & pfl) != 0;
      rin[i] = *(const float4*)(xb + (unsigned)(zero[i] ? safe_off : u_off[i]));
    }
  };
  // GLI: the operands of the NEXT chunk (the one the registers of the other path would be publishing now) go straight from
  // global memory into LDS buffer b; the barrier that ends the region waits for them (hipcc drains vmcnt before it)
  auto gl_in = [&](int i, int b) {
    if constexpr (GLI) {
      const bool z = (u_flag[i] & (fl1 | 16)) != 0;
      const float* src = z ? (const float*)wino_zero16 : xb1 + u_off[i];
      glds16(src, la0 + (unsigned)(b * BUF + i * TPB * 4) * 4u);
    }
  };
  auto pf_aff = [&]() {
    if constexpr (INAFF) {
      const int ch = pfc * 16 + 4 * (tid & 3);
      a_m = *(const float4*)(aff.mean + (size_t)pn * Kd + ch);
      a_r = *(const float4*)(aff.rstd + (size_t)pn * Kd + ch);
      a_g = *(const float4*)(aff.gamma + ch);
      a_b = *(const float4*)(aff.beta + ch);
    }
  };
  auto gl_w = [&](int i, int b) {
    if constexpr (PRE) {
      glds16_so(wu1, w_goff[i], la0 + (unsigned)(b * BUF + IN_F + i * TPB * 4) * 4u);
    }
  };
  auto pf_w = [&](int k) {
    const int c = pfc;
    if constexpr (PRE) {
    } else if (SC2 && c >= nh) {
      wreg[k][0] = (sc.w + (c - nh) * 16)[w_off1[k]];
    } else {
      // tap t's block: forward w + (t * Kd + 16 c) * Ndim, transposed w + (8 - t) * Ndim * KROW + 16 c -- selected with scalar
      // arithmetic, not a branch (a branch here cuts the MFMA unit this part rides in out of its scheduling region)
      const float* w0 = w_tap0 + (ptrdiff_t)c * w_cstep;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) wreg[k][tap] = (w0 + tap * w_tstep)[w_off[k]];
    }
    if constexpr (SC) wx[k] = (sc.w + (size_t)(c * 16) * Ndim)[w_off1[k]];
  };
  auto pub_in = [&](int i, int b) {
    float4 v = rin[GLI ? 0 : i];
    if constexpr (INAFF) {
      v.x = aff1(v.x, a_m.x, a_r.x, a_g.x, a_b.x, aff.slope); v.y = aff1(v.y, a_m.y, a_r.y, a_g.y, a_b.y, aff.slope);
      v.z = aff1(v.z, a_m.z, a_r.z, a_g.z, a_b.z, aff.slope); v.w = aff1(v.w, a_m.w, a_r.w, a_g.w, a_b.w, aff.slope);
    }
    if (zero[i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    *(float4*)(in_b(b) + u_lds[i]) = v;
  };
  auto pub_w = [&](int k, int b) {
    const int ci = w_ci[k], n = w_n[k];
    float* dst = w_b(b) + ((size_t)(ci >> 2) * CO_T + n) * 4 + (ci & 3);        // + pos * 4 * CO_T * 4
    if constexpr (PRE) {
      (void)dst;
    } else if (SC2 && pubc >= nh) {
      dst[0] = wreg[k][0];                               // the 1x1 weights, parked in position 0's block
    } else {
      // U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
      float t[4][3];
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) {
        const float g0 = wreg[k][bb], g1 = wreg[k][3 + bb], g2 = wreg[k][6 + bb];
        t[0][bb] = g0; t[1][bb] = 0.5f * (g0 + g1 + g2); t[2][bb] = 0.5f * (g0 - g1 + g2); t[3][bb] = g2;
      }
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float uu[4] = {t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) dst[(size_t)(a * 4 + bb) * 4 * CO_T * 4] = uu[bb];
      }
    }
    if constexpr (SC) wsc_b(b)[((size_t)(ci >> 2) * CO_T + n) * 4 + (ci & 3)] = wx[k];
  };
  constexpr int NP = NI + NWG + NWR;                    // staging parts
  // part p of the region that multiplies from buffer b: publish into b ^ 1, then request
  auto stage_part = [&](int p_, int b) {
    if (p_ < NI) {
      if constexpr (GLI) gl_in(p_, b ^ 1);
      else { pub_in(p_, b ^ 1); pf_in(p_); }
    } else if (p_ < NI + NWG) {
      gl_w(p_ - NI, b ^ 1);
    } else { pub_w(p_ - NI - NWG, b ^ 1); pf_w(p_ - NI - NWG); }
  };

  f32x4 macc[16][NR], acc[4][NR];
  [[maybe_unused]] f32x4 qacc[(SC || SC2) ? 4 : 1][NR];
  [[maybe_unused]] f32x4 pold[(ACC || BST) ? 4 : 1][NR];
  [[maybe_unused]] float nm[NR], nr[NR], ng_[NR], nb[NR];
#pragma unroll
  for (int p_ = 0; p_ < 16; ++p_)
#pragma unroll
    for (int j = 0; j < NR; ++j) macc[p_][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (SC || SC2) {
#pragma unroll
    for (int q_ = 0; q_ < 4; ++q_)
#pragma unroll
      for (int j = 0; j < NR; ++j) qacc[q_][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  if constexpr (BST) {
#pragma unroll
    for (int j = 0; j < NR; ++j) { ng_[j] = bst.gamma[co0 + j * 16 + lm]; nb[j] = bst.beta[co0 + j * 16 + lm]; }
  }
  const unsigned o_lane = (unsigned)(((wave * 4 + 2 * (kq >> 1)) * W + 8 * (kq & 1)) * os + lm);
  auto pxo = [&](int i, int r) { return (unsigned)((i >> 1) * W + 2 * r + (i & 1)); };

  // the values the outputs of the current item hold now (ACC), or y1 at those positions (BST): loaded at the start of the
  // item's last chunk, consumed by its epilogue
  auto load_old = [&]() {
    if constexpr (BST) {
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        nm[j] = bst.mean[(size_t)cn * Ndim + co0 + j * 16 + lm];
        nr[j] = bst.rstd[(size_t)cn * Ndim + co0 + j * 16 + lm];
      }
    }
    if constexpr (ACC || BST) {
      const float* yb = (BST ? bst.y1 : yo) + (((size_t)cn * H + cty * TH) * W + ctx * TW) * os + oc0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) pold[i][j][r] = yb[o_lane + pxo(i, r) * os + j * 16];
    }
  };
  auto epilogue = [&]() {                              // item (cn, cty, ctx), results in acc (and qacc for SC)
    if constexpr (STATS || BST) {
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (BST) {
              const float yv = pold[i][j][r];
              const float gz = acc[i][j][r] * lrelu_mask(in_affine(yv, nm[j], nr[j], ng_[j], nb[j]), bst.slope);
              acc[i][j][r] = gz;
              s1 += gz; s2 += gz * ((yv - nm[j]) * nr[j]);
            } else {
              const float v = acc[i][j][r]; s1 += v; s2 += v * v;
            }
          }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        float* rd = kq == 0 ? red + (wave * CO_T + j * 16 + lm) * 2 : red + 4 * CO_T * 2;
        *(float2*)rd = make_float2(s1, s2);
      }
    }
    float* yb = yo + (((size_t)cn * H + cty * TH) * W + ctx * TW) * os + oc0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NR; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) yb[o_lane + pxo(i, r) * os + j * 16] = acc[i][j][r] + (ACC ? pold[i][j][r] : 0.f);
    if constexpr (SC) {
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) { const float v = qacc[i][j][r]; s1 += v; s2 += v * v; }
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        float* rd = kq == 0 ? red_sc + (wave * CO_T + j * 16 + lm) * 2 : red_sc + 4 * CO_T * 2;
        *(float2*)rd = make_float2(s1, s2);
      }
      float* ys = sc.y + (((size_t)cn * H + cty * TH) * W + ctx * TW) * Ndim + co0;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NR; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) ys[o_lane + pxo(i, r) * Ndim + j * 16] = qacc[i][j][r];
#pragma unroll
      for (int q_ = 0; q_ < 4; ++q_)
#pragma unroll
        for (int j = 0; j < NR; ++j) qacc[q_][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto stats_out = [&](int cn, int cty, int ctx) {      // item (cn, cty, ctx), after the barrier that completes red[]
    if ((STATS || BST) && tid < CO_T) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m) { s1 += red[(m * CO_T + tid) * 2]; s2 += red[(m * CO_T + tid) * 2 + 1]; }
      float* const po = stats + (((size_t)cn * tiles_img + cty * tiles_x + ctx) * Ndim + co0 + tid) * 2;
      if constexpr (FIN) st_sc1_f2(po, s1, s2);         // in-launch finalize (common.h): write-through partials
      else *(float2*)po = make_float2(s1, s2);
      if constexpr (SC) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) { t1 += red_sc[(m * CO_T + tid) * 2]; t2 += red_sc[(m * CO_T + tid) * 2 + 1]; }
        float* const ps = sc.stats + (((size_t)cn * tiles_img + cty * tiles_x + ctx) * Ndim + co0 + tid) * 2;
        if constexpr (FIN) st_sc1_f2(ps, t1, t2);
        else *(float2*)ps = make_float2(t1, t2);
      }
    }
  };

  // ---- one chunk of MFMAs: lane (lm = tile, kq = channel quad) of this wave's 4 x 16-pixel strip.
  // ROT (every form but SC2): the chunk is ROTATED around its barrier -- PH 2 (body): units 0 .. NU-2, everything that reads this
  // chunk's LDS buffer; then the barrier; PH 1 (head) of the NEXT chunk: its first window rows and B fragments, requested from the
  // other buffer; PH 3 (tail): the last unit of THIS chunk, whose operands are all in registers -- its 16 MFMAs run while the
  // head's reads are in flight (unrotated, 500 cycles per chunk went by between the barrier and the first MFMA: notes).  PH 0: whole
  // chunk between two barriers (SC2, whose second-half chunks have no units).
  f32x4 d0[4], d1[4], d2[4], d3[4];                      // window rows of the chunk (live across the barrier under ROT)
  f32x4 bf[2][4], vv[2][4];                              // B fragments / transformed inputs of the current and the next unit
  auto mma_chunk = [&](int c, int buf, auto last_tag, auto phase_tag) __attribute__((always_inline)) {
    constexpr bool last = decltype(last_tag)::value;
    constexpr int PH = decltype(phase_tag)::value;
    static_assert(PH == 0 || !SC2, "rotated chunks: not for the fused shortcut data-gradient");
    const float* in_s = in_b(buf);
    const float* w_s = w_b(buf);
    [[maybe_unused]] const float* wsc_s = wsc_b(buf);
    // window of tile (tr, tc) = (lm >> 3, lm & 7): rows wave*4 + 2*tr + a, columns 2*tc + b, channel quad kq
    const float* dp = in_s + (((wave * 4 + 2 * (lm >> 3)) * IW) + 2 * (lm & 7)) * SPX + 4 * kq;
    const float* wc = w_s + ((size_t)kq * CO_T + lm) * 4;              // + (pos * 4 * CO_T + j * 16) * 4
    if constexpr (SC2) {
      if (c >= nh) {                                     // the shortcut's gradient: 1x1 products of the tile's four pixels only
        const f32x4 px[4] = {*(const f32x4*)(dp + (1 * IW + 1) * SPX), *(const f32x4*)(dp + (1 * IW + 2) * SPX),
                             *(const f32x4*)(dp + (2 * IW + 1) * SPX), *(const f32x4*)(dp + (2 * IW + 2) * SPX)};
#pragma unroll
        for (int j = 0; j < NR; ++j) {
          const f32x4 bw = *(const f32x4*)(wc + (size_t)(j * 16) * 4);
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q_ = 0; q_ < 4; ++q_) qacc[q_][j] = mfma16(px[q_][s], bw[s], qacc[q_][j]);
        }
#pragma unroll
        for (int p_ = 0; p_ < NP; ++p_) stage_part(p_, buf);
        return;                                          // (SC2 has no ACC / BST operands to request)
      }
    }
    if constexpr (PH == 0 || PH == 1) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        d1[b] = *(const f32x4*)(dp + (1 * IW + b) * SPX);
        d2[b] = *(const f32x4*)(dp + (2 * IW + b) * SPX);
      }
    }
    if constexpr (SC && (PH == 0 || PH == 2)) {          // forward shortcut: the tile's four raw pixels x this chunk's 1x1 weights
#pragma unroll
      for (int j = 0; j < NR; ++j) {
        const f32x4 bw = *(const f32x4*)(wsc_s + (((size_t)kq * CO_T) + j * 16 + lm) * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          qacc[0][j] = mfma16(d1[1][s], bw[s], qacc[0][j]);
          qacc[1][j] = mfma16(d1[2][s], bw[s], qacc[1][j]);
          qacc[2][j] = mfma16(d2[1][s], bw[s], qacc[2][j]);
          qacc[3][j] = mfma16(d2[2][s], bw[s], qacc[3][j]);
        }
      }
    }
    // Software pipeline over units u = (row combination xi of B^T, output-channel slab j), xi in the order 1, 2, 0, 3 (the first
    // two need window rows 1 and 2 only; row 0 is read during group 0, row 3 during group 1): region u holds the B-fragment reads
    // of unit u+1, the transform arithmetic of the next xi and the 16 MFMAs of unit u.  The ACC / BST operands of the epilogue
    // are requested once the window rows are dead (last group of the item's last chunk).
    constexpr int XO[4] = {1, 2, 0, 3};
    constexpr int NU = 4 * NR;
    // which unit carries staging part p: the LDS-DMA pieces all go out in the first units of the chunk (SMSUT_WINO_DMA_UNITS,
    // default half of them) -- they have to have LANDED at the barrier that ends the chunk, and an input piece that misses L2
    // takes a good part of a chunk to arrive; the register parts (publish what was requested a chunk ago, request the next) are
    // spread over all units as before
#ifndef SMSUT_WL_IG_VALU
#define SMSUT_WL_IG_VALU 6
#endif
#ifndef SMSUT_WL_IG_DS
#define SMSUT_WL_IG_DS 2
#endif
#ifndef SMSUT_WINO_DMA_UNITS
#define SMSUT_WINO_DMA_UNITS (NU / 2)
#endif
    constexpr int NDMA = (GLI ? NI : 0) + NWG;           // DMA parts: [0, NI) when GLI, [NI, NI + NWG)
    auto part_unit = [&](int p_) {
      const bool dma = (GLI && p_ < NI) || (p_ >= NI && p_ < NI + NWG);
      if (dma) {
        const int k = (GLI || p_ < NI) ? p_ : p_ - NI;   // index among the DMA parts, inputs first
        return (k * (SMSUT_WINO_DMA_UNITS)) / (NDMA > 0 ? NDMA : 1);
      }
      return p_ % (PH == 0 ? NU : NU - 1);               // (rotated: the last unit runs after the barrier that publishes them)
    };
    // (sel: which half of bf / vv -- indexed here, not passed as a pointer: with the arrays at kernel scope a pointer
    //  parameter kept them in scratch memory in several instantiations)
    auto ld_b = [&](int u, int sel) __attribute__((always_inline)) {
      const int xi = XO[u / NR], j = u % NR;
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) bf[sel][nu] = *(const f32x4*)(wc + ((size_t)(xi * 4 + nu) * 4 * CO_T + j * 16) * 4);
    };
    auto xform = [&](int g, int sel) __attribute__((always_inline)) {
      const int xi = XO[g];
      f32x4 t[4];                                        // row combination xi of B^T: d0-d2 | d1+d2 | d2-d1 | d1-d3
#pragma unroll
      for (int b = 0; b < 4; ++b) t[b] = xi == 0 ? d0[b] - d2[b] : xi == 1 ? d1[b] + d2[b] : xi == 2 ? d2[b] - d1[b] : d1[b] - d3[b];
      vv[sel][0] = t[0] - t[2]; vv[sel][1] = t[1] + t[2]; vv[sel][2] = t[2] - t[1]; vv[sel][3] = t[1] - t[3];
    };
    if constexpr (PH == 0 || PH == 1) ld_b(0, 0);
    if constexpr (PH == 1) return;
    if constexpr (PH == 0 || PH == 2) {
      xform(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      WL_STAMP(ts1);
    }
    static_for<0, NU>([&](auto u_tag) __attribute__((always_inline)) {
      constexpr int u = decltype(u_tag)::value;
      constexpr int g = u / NR, j = u % NR;
      if constexpr ((PH == 2 && u == NU - 1) || (PH == 3 && u != NU - 1)) return;
#pragma unroll
      for (int p_ = 0; p_ < NP; ++p_)
        if (PH != 3)
        if (part_unit(p_) == u) stage_part(p_, buf);     // this unit's share of the staging work, in the shadow of its MFMAs
      if (PH != 3 && u + 1 < NU) {
        ld_b(u + 1, (u + 1) & 1);
        if ((u + 1) % NR == 0) xform(g + 1, (g + 1) & 1);
      }
      if (PH != 3 && j == NR - 1) {                      // (after the transform above in program order: its operands are dead)
        if (g == 0) {
#pragma unroll
          for (int b = 0; b < 4; ++b) d0[b] = *(const f32x4*)(dp + (0 * IW + b) * SPX);
        } else if (g == 1) {
#pragma unroll
          for (int b = 0; b < 4; ++b) d3[b] = *(const f32x4*)(dp + (3 * IW + b) * SPX);
        } else if (g == 2 && last) {
          load_old();
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
          macc[XO[g] * 4 + nu][j] = mfma16(vv[g & 1][nu][s], bf[u & 1][nu][s], macc[XO[g] * 4 + nu][j]);
      // SMSUT_WL_IGROUP: force an interleave -- after every MFMA up to six VALU, two LDS and two global-memory instructions of
      // this region.  Off: with the staging done by LDS-DMA the region has ~2 other instructions per MFMA and the scheduler's
      // own order measures 1-2 us (of 50) faster than any forced one (6/2, 4/1, 3/1, 2/1 tried: profiles/r03_notes.md)
      __builtin_amdgcn_sched_barrier(0);
    });
  };
  auto out_transform = [&]() {                          // A^T m A, element-wise over the lane's four tiles; resets the accumulators
#pragma unroll
    for (int j = 0; j < NR; ++j) {
      f32x4 c0[4], c1[4];
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        c0[nu] = macc[0 + nu][j] + macc[4 + nu][j] + macc[8 + nu][j];
        c1[nu] = macc[4 + nu][j] - macc[8 + nu][j] - macc[12 + nu][j];
      }
      acc[0][j] = c0[0] + c0[1] + c0[2]; acc[1][j] = c0[1] - c0[2] - c0[3];
      acc[2][j] = c1[0] + c1[1] + c1[2]; acc[3][j] = c1[1] - c1[2] - c1[3];
#pragma unroll
      for (int p_ = 0; p_ < 16; ++p_) macc[p_][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (SC2) {
#pragma unroll
        for (int q_ = 0; q_ < 4; ++q_) { acc[q_][j] += qacc[q_][j]; qacc[q_][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      }
    }
  };

#ifdef SMSUT_WL_STAMPS
  unsigned long long tp0, tp1, tp2, tp3;
  WL_STAMP(tp0);
#endif
  // prologue: chunk 0 -> buffer 0, chunk 1 -> registers
  pf_setup();
#pragma unroll
  for (int i = 0; i < NI; ++i) pf_in(i);
#pragma unroll
  for (int k = 0; k < NWR; ++k) pf_w(k);
  pf_aff();
  pf_advance();
  pf_setup();                                           // (pubc = 0: what the registers hold)
#pragma unroll
  for (int p_ = 0; p_ < NP; ++p_) stage_part(p_, 1);    // publishes into buffer 0, requests chunk 1
  pf_aff();
  pf_advance();
#ifdef SMSUT_WL_STAMPS
  WL_STAMP(tp1);
#endif
  if constexpr (GLI || PRE) glds_wait();
#ifdef SMSUT_WL_STAMPS
  WL_STAMP(tp2);
#endif
  __syncthreads();
#ifdef SMSUT_WL_STAMPS
  WL_STAMP(tp3);
#endif
  int buf = 0;
  // one chunk = one region = one barrier: staging parts + this chunk's MFMAs (+ on an item's last chunk: transform, statistics,
  // stores).  `last` is a compile-time tag: as a run-time condition the compiler predicated the whole epilogue into every chunk
  constexpr bool ROT = !SC2;
  using ph_whole = std::integral_constant<int, 0>;
  using ph_head = std::integral_constant<int, 1>;
  using ph_body = std::integral_constant<int, 2>;
  using ph_tail = std::integral_constant<int, 3>;
  [[maybe_unused]] int sn = cn, sty = cty, stx = ctx;    // ROT: the item whose statistics the next barrier completes
  [[maybe_unused]] bool st_pending = false;
  if constexpr (ROT) mma_chunk(0, 0, std::false_type{}, ph_head{});       // first rows / fragments of chunk 0
  auto region = [&](int c, auto last_tag, auto first_tag) __attribute__((always_inline)) {
    constexpr bool last = decltype(last_tag)::value;
    constexpr bool first = decltype(first_tag)::value;
    WL_STAMP(ts0);
    pf_setup();
    if constexpr (ROT) mma_chunk(c, buf, last_tag, ph_body{});
    else mma_chunk(c, buf, last_tag, ph_whole{});
    pf_aff();
    pf_advance();
    WL_STAMP(ts2);
    if constexpr (!ROT && last) { out_transform(); epilogue(); }
    WL_STAMP(ts3);
    if constexpr (GLI || PRE) glds_wait();               // this wave's LDS-DMA pieces of the next chunk have landed
    WL_STAMP(ts4);
    __syncthreads();                                     // buffer buf is free, buf ^ 1 is complete; red[] is complete
    if constexpr (!ROT) {
      if constexpr (last) stats_out(cn, cty, ctx);
    } else {
      // the previous item's epilogue ran in the tail of its last region, after that region's barrier: this one completes it
      if constexpr (first) {
        if (st_pending) stats_out(sn, sty, stx);
      }
      mma_chunk(c, buf ^ 1, last_tag, ph_head{});        // next chunk's first reads ...
      mma_chunk(c, buf, last_tag, ph_tail{});            // ... under this chunk's last unit
      if constexpr (last) {
        out_transform();
        epilogue();
        sn = cn; sty = cty; stx = ctx; st_pending = true;
      }
    }
    WL_STAMP(ts5);
#ifdef SMSUT_WL_STAMPS
    acc_t[0] += ts1 - ts0; acc_t[1] += ts2 - ts1; acc_t[2] += ts3 - ts2; acc_t[3] += ts4 - ts3; acc_t[4] += ts5 - ts4; acc_t[5] += 1;
#endif
    buf ^= 1;
  };
  for (int item = item0; item < item1; ++item) {
    region(0, std::false_type{}, std::true_type{});      // (nch >= 2: the first chunk of an item is never its last)
#pragma unroll 1
    for (int c = 1; c + 1 < nch; ++c) region(c, std::false_type{}, std::false_type{});
    region(nch - 1, std::true_type{}, std::false_type{});
    advance(cn, cty, ctx);
  }
  if constexpr (ROT && (STATS || BST)) {
    __syncthreads();
    stats_out(sn, sty, stx);
  }
  if constexpr (FIN)                                    // the image(s) this workgroup completes: finalised here
    fin_tail<!BST>(fin, stats, SC ? sc.stats : nullptr, item0, item1, tiles_img, tiles_img * (int)gridDim.y, Ndim, H * W,
                   reinterpret_cast<int*>(smem + 1024), reinterpret_cast<double*>(smem));
#ifdef SMSUT_WL_STAMPS
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    for (int i = 0; i < 6; ++i) y[i] = (float)acc_t[i];
    y[6] = (float)(ts5 - ts_begin);
    y[7] = (float)(tp0 - ts_begin); y[8] = (float)(tp1 - tp0); y[9] = (float)(tp2 - tp1); y[10] = (float)(tp3 - tp2);
  }
#endif
}

// ---- prepared weights: U = G g G^T of whole weight tensors, written once per weight version in the image order the kernel's
//      LDS-DMA copies (several tensors per launch: the table travels in the kernel arguments, so a captured launch is
//      self-contained)
constexpr int PREP_MAX = 32;
struct PrepEntry { const float* w; float* u; int kd, nd, tr, blk0; };
struct PrepTable { PrepEntry e[PREP_MAX]; int n; };

__global__ void __launch_bounds__(TPB) wino_u_prepare(PrepTable t) {
  int ei = 0;
  for (int i = 1; i < t.n; ++i)
    if ((int)blockIdx.x >= t.e[i].blk0) ei = i;
  const PrepEntry e = t.e[ei];
  const int lt = ((int)blockIdx.x - e.blk0) * TPB + (int)threadIdx.x;       // (chunk c, slab j, channel quad kq, channel n16)
  if (lt >= e.kd * (e.nd >> 2)) return;
  const int n16 = lt & 15, kq = (lt >> 4) & 3, cj = lt >> 6;
  const int nj = e.nd >> 4, c = cj / nj, j = cj - c * nj;
  const int n = j * 16 + n16, ci0 = c * 16 + kq * 4;
  float g[4][9];
  if (e.tr) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float4 v = *(const float4*)(e.w + ((size_t)(8 - tap) * e.nd + n) * e.kd + ci0);
      g[0][tap] = v.x; g[1][tap] = v.y; g[2][tap] = v.z; g[3][tap] = v.w;
    }
  } else {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int q = 0; q < 4; ++q) g[q][tap] = e.w[((size_t)tap * e.kd + ci0 + q) * e.nd + n];
  }
  float uu[4][16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {                          // the arithmetic of conv_wino_l's on-the-fly form, in its order
    float tt[4][3];
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      const float g0 = g[q][bb], g1 = g[q][3 + bb], g2 = g[q][6 + bb];
      tt[0][bb] = g0; tt[1][bb] = 0.5f * (g0 + g1 + g2); tt[2][bb] = 0.5f * (g0 - g1 + g2); tt[3][bb] = g2;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      uu[q][a * 4 + 0] = tt[a][0];
      uu[q][a * 4 + 1] = 0.5f * (tt[a][0] + tt[a][1] + tt[a][2]);
      uu[q][a * 4 + 2] = 0.5f * (tt[a][0] - tt[a][1] + tt[a][2]);
      uu[q][a * 4 + 3] = tt[a][2];
    }
  }
  float4* out = (float4*)e.u + ((size_t)cj * 64 + kq) * 16 + n16;           // + pos * 4 * 16
#pragma unroll
  for (int pos = 0; pos < 16; ++pos) out[(size_t)pos * 64] = make_float4(uu[0][pos], uu[1][pos], uu[2][pos], uu[3][pos]);
}

inline int device_cus() {
  static const int cus = [] {
    if (const char* e = getenv("SMSUT_CUS")) { const int v = atoi(e); if (v > 0) return v; }      // (tuning hook: see conv_mfma.hip)
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    return n > 0 ? n : 256;
  }();
  return cus;
}

template <auto Kern>
inline void allow_big_lds(size_t bytes) {
  // once per instantiation, safe when forward and autograd's backward thread arrive together (r03 had a plain flag here)
  static std::once_flag once;
  if (bytes > 64 * 1024)
    std::call_once(once, [] { (void)hipFuncSetAttribute((const void*)Kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
}

template <int NTN>
int launch_ntn(const float* x, const float* x2, const float* w, const float* wu, float* y, float* y2, int split, int N, int H, int W,
               int Kdim, int Ndim, int transposed, float* stats, const WinoBst* bst, const WinoAff* aff, const WinoSc* sc, hipStream_t st,
               const FinRef* fin) {
  const int tiles_x = W / TW, tiles_img = tiles_x * (H / TH);
  if (fin && (!fin->tickets || !fin->o0 || !fin->o1 || !stats || y2 || (transposed & 2) ||
              (fin->s0 && !(sc && !(transposed & 1) && sc->stats && fin->s1)) || (!fin->s0 && sc)))
    return -1;
  const FinRef fv = fin ? *fin : FinRef{};
  const int nz = Ndim / (16 * NTN);
  const int64_t items = (int64_t)N * tiles_img;
  // resident workgroups per CU: ONE (one wave per SIMD: registers, 2 LDS buffers).  SMSUT_WINO_WGS_PER_CU > 1 sizes the grid for that
  // many ROUNDS of smaller workgroups instead (tuning hook: with the discriminator's small kernels on a second queue a CU may be
  // taken when the grid launches, and a workgroup that waits for it holds the whole launch back by its own length)
  static const int per_cu = [] { const char* e = getenv("SMSUT_WINO_WGS_PER_CU"); const int v = e ? atoi(e) : 1; return v > 0 ? v : 1; }();
  const int64_t slots = (int64_t)device_cus() * per_cu;
  int ipw = (int)((items * nz + slots - 1) / slots);
  if (ipw < 1) ipw = 1;
  dim3 grid((unsigned)((items + ipw - 1) / ipw), nz);
  const int nch = Kdim / 16;
  const WinoBst bv = bst ? *bst : WinoBst{};
  const WinoAff av = aff ? *aff : WinoAff{};
  const WinoSc sv = sc ? *sc : WinoSc{};
  const bool sc2 = sc && (transposed & 1), scf = sc && !sc2;
  const size_t sh = wino_l_lds<NTN>(scf, !aff && SMSUT_WINO_GLDS != 0);
#define WGO2(ST, AC, BS, DU, IA, S1, S2, PR, FI)                                                                          \
  do {                                                                                                                     \
    allow_big_lds<conv_wino_l<NTN, ST, AC, BS, DU, IA, S1, S2, PR, FI>>(sh);                                               \
    conv_wino_l<NTN, ST, AC, BS, DU, IA, S1, S2, PR, FI><<<grid, TPB, sh, st>>>(x, x2, w, wu, y, y2, split, N, H, W, nch,  \
                                                                                Ndim, tiles_x, tiles_img, ipw, transposed, \
                                                                                stats, bv, av, sv, fv);                    \
  } while (0)
#define WGO1(ST, AC, BS, DU, IA, S1, S2, PR)                                                                              \
  do {                                                                                                                     \
    if constexpr ((ST) || (BS)) {                                                                                          \
      if (fin) { WGO2(ST, AC, BS, DU, IA, S1, S2, PR, true); break; }                                                      \
    }                                                                                                                      \
    WGO2(ST, AC, BS, DU, IA, S1, S2, PR, false);                                                                           \
  } while (0)
#define WGO(ST, AC, BS, DU, IA, S1, S2)                                                                                    \
  do {                                                                                                                     \
    if constexpr (!(S2)) {                                                                                                 \
      if (wu) { WGO1(ST, AC, BS, DU, IA, S1, S2, true); break; }                                                           \
    }                                                                                                                      \
    WGO1(ST, AC, BS, DU, IA, S1, S2, false);                                                                               \
  } while (0)
  if (sc2) {
    if (!x2 || !sc->w || stats || bst || aff || (transposed & 2) || nch % 2) return -1;
    WGO(false, false, false, true, false, false, true);
  } else if (scf) {
    if (!stats || bst || aff || y2 || transposed || !sc->w || !sc->y || !sc->stats || (x2 && nch % 2)) return -1;
    if (x2) WGO(true, false, false, true, false, true, false);
    else WGO(true, false, false, false, false, true, false);
  } else if (aff) {
    if (!stats || bst || y2 || x2 || transposed) return -1;
    WGO(true, false, false, false, true, false, false);
  } else if (x2) {
    if (!stats || bst || y2 || transposed || nch % 2) return -1;
    WGO(true, false, false, true, false, false, false);
  } else if (bst) {
    if (!stats || y2 || (transposed & 2)) return -1;
    WGO(false, false, true, false, false, false, false);
  } else if (transposed & 2) {
    if (stats) return -1;
    WGO(false, true, false, false, false, false, false);
  } else if (stats) {
    if (y2) return -1;
    WGO(true, false, false, false, false, false, false);
  } else {
    WGO(false, false, false, false, false, false, false);
  }
#undef WGO
#undef WGO1
#undef WGO2
  return 0;
}


// ------------------------------------------------------------------------------------------------ weight gradient (r03)
// Winograd F(3x3, 2x2) weight gradient.  The forward form Y = A^T [(G g G^T) (.) (B^T d B)] A is bilinear in (g, d), so
//   dg = G^T [ sum over tiles of (A dY A^T) (.) (B^T d B) ] G :
// 16 per-position GEMMs dU[pos][ci][co] = sum_tiles V[pos][tile][ci] Z[pos][tile][co] with the TILES as the reduction -- 16
// products per 2x2 tile of gy instead of 36.  MFMA roles: M = input channel, N = output channel, K = tile.  Lane (lm, kq) is
// channel lm of BOTH operands for tiles 4kq .. 4kq+3 of the wave's 4 x 16-pixel strip (k-slot kq of MFMA s = tile 4kq + s): it
// needs ONE channel of four neighbouring windows, the transpose of the pixel-major tensors -- so the operands are staged into
// channel-major LDS planes (a thread's float4 = four channels of a pixel becomes four ds_write_b32 into four planes; plane sizes
// 364 / 324 floats make the writes and the b128 row reads conflict-free).  Both transforms run in registers, position by position
// (row combination xi, then column combination nu -> four tiles' values -> 4 x CIT x COT MFMAs), so only the raw rows stay live.
// The negations of A's last row / column are left out (Z' = |A| dY |A|^T pattern with +) and applied as signs in the final pass.
// A workgroup owns a (16 CIT x 16 COT) slab of (ci, co) and a range of 16x16-pixel items; its 4 waves take a strip each (partial
// sums over different tiles), are combined through LDS at the end, and the split slabs are summed, sign-fixed and folded by
// G^T . G into gw by wino_wg_final (fixed order: deterministic).
constexpr int WRS = 20;                      // row stride (floats) of the channel-major planes
constexpr int PSX = IH * WRS + 4;            // x plane: 18 rows (+4: plane stride / 4 odd -> conflict-free b128 reads across channels)
constexpr int PSG = TH * WRS + 4;            // gy plane: 16 rows

template <int CIT, int COT>
constexpr int wino_wg_buf_floats() { return 16 * CIT * PSX + 16 * COT * PSG; }

template <int CIT, int COT, bool DUAL, bool INAFF>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(1, 1)))
conv_wino_wg(const float* __restrict__ x, const float* __restrict__ x2, int ca, const float* __restrict__ gy, float* __restrict__ part,
             int N, int H, int W, int Cin, int Cout, int tiles_x, int tiles_img, int items_per_split, WinoAff aff) {
  constexpr int CI_T = 16 * CIT, CO_T = 16 * COT;
  constexpr int XB = CI_T * PSX;
  constexpr int BUF = wino_wg_buf_floats<CIT, COT>();
  constexpr int NUX = IH * IW * 4 * CIT, NIX = (NUX + TPB - 1) / TPB;      // float4 units of the x tile, per thread
  constexpr int NIG = 4 * COT;                                              // ... of the gy tile (16 x 16 x 4 COT / 256)
  extern __shared__ float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, kq = lane >> 4;
  const int ci0 = blockIdx.y * CI_T, co0 = blockIdx.z * CO_T;
  const int total_items = N * tiles_img;
  const int item0 = blockIdx.x * items_per_split;
  const int item1 = min(item0 + items_per_split, total_items);
  const int tiles_y = tiles_img / tiles_x;
  // source of this slab's x channels (virtual cat: a slab never straddles the seam -- host)
  const bool second = DUAL && ci0 >= ca;
  const float* xs = second ? x2 : x;
  const int xst = DUAL ? (second ? Cin - ca : ca) : Cin;          // pixel stride of that tensor
  const int xc0 = second ? ci0 - ca : ci0;                         // first channel of the slab inside it

  int ux_off[NIX], ux_lds[NIX], ux_flag[NIX];
  float4 rx[NIX];
  bool zx[NIX];
#pragma unroll
  for (int i = 0; i < NIX; ++i) {
    const int u = tid + i * TPB;
    const bool real = u < NUX;
    const int uu = real ? u : 0;
    const int q = uu % (4 * CIT), pix = uu / (4 * CIT);
    const int iy = pix / IW, ix = pix % IW;
    ux_off[i] = (iy * W + ix) * xst + xc0 + 4 * q;
    ux_lds[i] = real ? (4 * q) * PSX + iy * WRS + ix : -1;
    ux_flag[i] = (iy < 1 ? 1 : 0) | (iy >= TH + 1 ? 2 : 0) | (ix < 1 ? 4 : 0) | (ix >= TW + 1 ? 8 : 0);
  }
  const int safe_off = (W + 1) * xst + xc0;
  int ug_off[NIG], ug_lds[NIG];
  float4 rg[NIG];
#pragma unroll
  for (int i = 0; i < NIG; ++i) {
    const int u = tid + i * TPB;
    const int q = u % (4 * COT), pix = u / (4 * COT);
    const int oy = pix / TW, ox = pix % TW;
    ug_off[i] = (oy * W + ox) * Cout + co0 + 4 * q;
    ug_lds[i] = XB + (4 * q) * PSG + oy * WRS + ox;
  }
  [[maybe_unused]] float4 a_m, a_r, a_g, a_b;
  if constexpr (INAFF) {
    const int ch = ci0 + 4 * (tid % (4 * CIT));
    a_g = *(const float4*)(aff.gamma + ch);
    a_b = *(const float4*)(aff.beta + ch);
  }

  f32x4 macc[16][CIT][COT];
#pragma unroll
  for (int p_ = 0; p_ < 16; ++p_)
#pragma unroll
    for (int a = 0; a < CIT; ++a)
#pragma unroll
      for (int b = 0; b < COT; ++b) macc[p_][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int pn = item0 / tiles_img, pty, ptx;
  { const int t = item0 - pn * tiles_img; pty = t / tiles_x; ptx = t - pty * tiles_x; }
  auto advance = [&](int& n_, int& ty_, int& tx_) {
    if (++tx_ == tiles_x) { tx_ = 0; if (++ty_ == tiles_y) { ty_ = 0; ++n_; } }
  };
  auto prefetch = [&]() {                               // item (pn, pty, ptx) -> registers
    const int pfl = (pty == 0 ? 1 : 0) | (pty == tiles_y - 1 ? 2 : 0) | (ptx == 0 ? 4 : 0) | (ptx == tiles_x - 1 ? 8 : 0);
    const float* xb = xs + (ptrdiff_t)(((pn * H + pty * TH - 1) * W) + ptx * TW - 1) * xst;   // (may point before the tensor: only
    const float* gb = gy + (ptrdiff_t)(((pn * H + pty * TH) * W) + ptx * TW) * Cout;            //  used with in-image offsets)
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
      zx[i] = (ux_flag[i] & pfl) != 0;
      rx[i] = *(const float4*)(xb + (unsigned)(zx[i] ? safe_off : ux_off[i]));
    }
#pragma unroll
    for (int i = 0; i < NIG; ++i) rg[i] = *(const float4*)(gb + (unsigned)ug_off[i]);
    if constexpr (INAFF) {
      const int ch = ci0 + 4 * (tid % (4 * CIT));
      a_m = *(const float4*)(aff.mean + (size_t)pn * Cin + ch);
      a_r = *(const float4*)(aff.rstd + (size_t)pn * Cin + ch);
    }
  };
  auto publish = [&](int b) {                           // registers -> channel-major planes of buffer b
    float* base = smem + b * BUF;
#pragma unroll
    for (int i = 0; i < NIX; ++i) {
      float4 v = rx[i];
      if constexpr (INAFF) {
        v.x = aff1(v.x, a_m.x, a_r.x, a_g.x, a_b.x, aff.slope); v.y = aff1(v.y, a_m.y, a_r.y, a_g.y, a_b.y, aff.slope);
        v.z = aff1(v.z, a_m.z, a_r.z, a_g.z, a_b.z, aff.slope); v.w = aff1(v.w, a_m.w, a_r.w, a_g.w, a_b.w, aff.slope);
      }
      if (zx[i]) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ux_lds[i] >= 0) {
        float* d = base + ux_lds[i];
        d[0] = v.x; d[PSX] = v.y; d[2 * PSX] = v.z; d[3 * PSX] = v.w;
      }
    }
#pragma unroll
    for (int i = 0; i < NIG; ++i) {
      float* d = base + ug_lds[i];
      d[0] = rg[i].x; d[PSG] = rg[i].y; d[2 * PSG] = rg[i].z; d[3 * PSG] = rg[i].w;
    }
  };
  auto compute = [&](int b) {
    const float* base = smem + b * BUF;
    const int ro = (wave * 4 + 2 * (kq >> 1)) * WRS + 8 * (kq & 1);     // first row / column of the lane's four tiles
    float xr[CIT][4][10];                               // channel lm of slab block a: window rows 0..3, columns 0..9
#pragma unroll
    for (int a = 0; a < CIT; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* p_ = base + (a * 16 + lm) * PSX + ro + r * WRS;
        const f32x4 v0 = *(const f32x4*)p_, v1 = *(const f32x4*)(p_ + 4);
        const float2 v2 = *(const float2*)(p_ + 8);
        xr[a][r][0] = v0[0]; xr[a][r][1] = v0[1]; xr[a][r][2] = v0[2]; xr[a][r][3] = v0[3];
        xr[a][r][4] = v1[0]; xr[a][r][5] = v1[1]; xr[a][r][6] = v1[2]; xr[a][r][7] = v1[3];
        xr[a][r][8] = v2.x; xr[a][r][9] = v2.y;
      }
    float gr[COT][2][8];                                // channel lm of slab block b: the strip's rows 2tr, 2tr+1, columns 0..7
#pragma unroll
    for (int b_ = 0; b_ < COT; ++b_)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float* p_ = base + XB + (b_ * 16 + lm) * PSG + ro + r * WRS;
        const f32x4 v0 = *(const f32x4*)p_, v1 = *(const f32x4*)(p_ + 4);
        gr[b_][r][0] = v0[0]; gr[b_][r][1] = v0[1]; gr[b_][r][2] = v0[2]; gr[b_][r][3] = v0[3];
        gr[b_][r][4] = v1[0]; gr[b_][r][5] = v1[1]; gr[b_][r][6] = v1[2]; gr[b_][r][7] = v1[3];
      }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      float tx[CIT][10], zr[COT][8];
#pragma unroll
      for (int a = 0; a < CIT; ++a)
#pragma unroll
        for (int c = 0; c < 10; ++c)
          tx[a][c] = xi == 0 ? xr[a][0][c] - xr[a][2][c] : xi == 1 ? xr[a][1][c] + xr[a][2][c]
                   : xi == 2 ? xr[a][2][c] - xr[a][1][c] : xr[a][1][c] - xr[a][3][c];
#pragma unroll
      for (int b_ = 0; b_ < COT; ++b_)
#pragma unroll
        for (int c = 0; c < 8; ++c)                     // rows of |A|: g0 | g0 + g1 | g0 - g1 | g1 (sign of the last one: final pass)
          zr[b_][c] = xi == 0 ? gr[b_][0][c] : xi == 1 ? gr[b_][0][c] + gr[b_][1][c] : xi == 2 ? gr[b_][0][c] - gr[b_][1][c] : gr[b_][1][c];
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) {
        f32x4 v[CIT], z[COT];
#pragma unroll
        for (int a = 0; a < CIT; ++a)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            v[a][s] = nu == 0 ? tx[a][2 * s] - tx[a][2 * s + 2] : nu == 1 ? tx[a][2 * s + 1] + tx[a][2 * s + 2]
                    : nu == 2 ? tx[a][2 * s + 2] - tx[a][2 * s + 1] : tx[a][2 * s + 1] - tx[a][2 * s + 3];
#pragma unroll
        for (int b_ = 0; b_ < COT; ++b_)
#pragma unroll
          for (int s = 0; s < 4; ++s)
            z[b_][s] = nu == 0 ? zr[b_][2 * s] : nu == 1 ? zr[b_][2 * s] + zr[b_][2 * s + 1]
                     : nu == 2 ? zr[b_][2 * s] - zr[b_][2 * s + 1] : zr[b_][2 * s + 1];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int a = 0; a < CIT; ++a)
#pragma unroll
            for (int b_ = 0; b_ < COT; ++b_) macc[xi * 4 + nu][a][b_] = mfma16(v[a][s], z[b_][s], macc[xi * 4 + nu][a][b_]);
      }
    }
  };

  if (item0 < item1) {
    prefetch();
    advance(pn, pty, ptx);
    publish(0);
    if (item0 + 1 < item1) { prefetch(); advance(pn, pty, ptx); }
    __syncthreads();
    int buf = 0;
    for (int item = item0; item < item1; ++item) {
      if (item + 1 < item1) publish(buf ^ 1);             // (operands requested a whole item ago)
      if (item + 2 < item1) { prefetch(); advance(pn, pty, ptx); }
      compute(buf);
      __syncthreads();
      buf ^= 1;
    }
  }
  // ---- combine the four waves' partial sums -- fixed order (0 + 1) + (2 + 3) -- and store the slab of this split
  f32x4* xch = reinterpret_cast<f32x4*>(smem);           // [2][16][CIT][COT][64 lanes]
  constexpr int XW = 16 * CIT * COT * 64;
#pragma unroll 1
  for (int round = 0; round < 2; ++round) {
    const bool writer = round == 0 ? (wave & 1) : wave == 2;
    const bool reader = round == 0 ? !(wave & 1) : wave == 0;
    const int slot = round == 0 ? (wave >> 1) : 0;
    __syncthreads();
    if (writer) {
#pragma unroll
      for (int p_ = 0; p_ < 16; ++p_)
#pragma unroll
        for (int a = 0; a < CIT; ++a)
#pragma unroll
          for (int b_ = 0; b_ < COT; ++b_) xch[slot * XW + ((p_ * CIT + a) * COT + b_) * 64 + lane] = macc[p_][a][b_];
    }
    __syncthreads();
    if (reader) {
#pragma unroll
      for (int p_ = 0; p_ < 16; ++p_)
#pragma unroll
        for (int a = 0; a < CIT; ++a)
#pragma unroll
          for (int b_ = 0; b_ < COT; ++b_) macc[p_][a][b_] += xch[slot * XW + ((p_ * CIT + a) * COT + b_) * 64 + lane];
    }
  }
  if (wave == 0) {                                        // D layout: lane (lm = co, kq) holds rows ci = 4 kq + r
    float* dst = part + (size_t)blockIdx.x * 16 * Cin * Cout;
#pragma unroll
    for (int p_ = 0; p_ < 16; ++p_)
#pragma unroll
      for (int a = 0; a < CIT; ++a)
#pragma unroll
        for (int b_ = 0; b_ < COT; ++b_)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dst[((size_t)p_ * Cin + ci0 + a * 16 + 4 * kq + r) * Cout + co0 + b_ * 16 + lm] = macc[p_][a][b_][r];
  }
}

// Sum of the split slabs, stage 1: group g of GRP consecutive splits -> part2[g][16 * CC] (one element per thread, eight loads in
// flight); stage 2 (wino_wg_final) sums the <= 16 groups in fixed order, restores the signs left out of A (sigma = (1, 1, 1, -1))
// and folds gw[tap][ci][co] = G^T (sigma dU sigma) G.  Fixed orders everywhere: deterministic.
__global__ void __launch_bounds__(TPB) wino_wg_reduce(const float* __restrict__ part, float* __restrict__ part2, int splits, int grp,
                                                      int64_t E) {
  const int64_t e = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (e >= E) return;
  const int k0 = blockIdx.y * grp, k1 = min(k0 + grp, splits);
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int k = k0;
  for (; k + 8 <= k1; k += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] += part[(size_t)(k + j) * E + e];
  }
  for (; k < k1; ++k) s[0] += part[(size_t)k * E + e];
  part2[(size_t)blockIdx.y * E + e] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

__global__ void __launch_bounds__(TPB) wino_wg_final(const float* __restrict__ part, float* __restrict__ gw, int groups, int CC) {
  const int idx = blockIdx.x * TPB + threadIdx.x;        // (ci, co) pair
  if (idx >= CC) return;
  float u[4][4];
#pragma unroll
  for (int p_ = 0; p_ < 16; ++p_) {
    float s = 0.f;
    for (int k = 0; k < groups; ++k) s += part[((size_t)k * 16 + p_) * CC + idx];
    const bool neg = ((p_ >> 2) == 3) != ((p_ & 3) == 3);
    u[p_ >> 2][p_ & 3] = neg ? -s : s;
  }
  // G^T = [[1, .5, .5, 0], [0, .5, -.5, 0], [0, .5, .5, 1]]
  float t[3][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    t[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
    t[1][b] = 0.5f * (u[1][b] - u[2][b]);
    t[2][b] = 0.5f * (u[1][b] + u[2][b]) + u[3][b];
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    gw[(size_t)(a * 3 + 0) * CC + idx] = t[a][0] + 0.5f * (t[a][1] + t[a][2]);
    gw[(size_t)(a * 3 + 1) * CC + idx] = 0.5f * (t[a][1] - t[a][2]);
    gw[(size_t)(a * 3 + 2) * CC + idx] = 0.5f * (t[a][1] + t[a][2]) + t[a][3];
  }
}

constexpr int WG_GROUPS = 16;                // stage-1 groups of the split sum (when there are more splits than that)

struct WgPlan { int cit, cot, splits, items_per_split; };
inline WgPlan plan_wino_wg(int N, int H, int W, int Cin, int Cout) {
  WgPlan p;
  // slab shapes: two output-channel blocks per workgroup when there are that many (the gy side of the transform is the cheap
  // one), else two input-channel blocks, else 16 x 16
  p.cit = 1; p.cot = 1;
  if (Cout % 32 == 0) p.cot = 2;
  else if (Cin % 32 == 0) p.cit = 2;
  const int slabs = (Cin / (16 * p.cit)) * (Cout / (16 * p.cot));
  const int items = N * (H / TH) * (W / TW);
  // ONE round of one workgroup per CU: a workgroup's fixed costs (first item's load latency, the cross-wave combine, the slab
  // store: ~10 us) want as many items behind them as the grid allows
  int want = (device_cus() + slabs - 1) / slabs;
  const int64_t cap = ((int64_t)8 << 20) / ((int64_t)16 * Cin * Cout);     // split slabs the final pass re-reads: <= 8 M floats
  if (want > cap) want = (int)(cap < 1 ? 1 : cap);
  if (want > items) want = items;
  if (want < 1) want = 1;
  p.items_per_split = (items + want - 1) / want;
  p.splits = (items + p.items_per_split - 1) / p.items_per_split;
  return p;
}

template <int CIT, int COT>
int launch_wino_wg(const float* x, const float* x2, int ca, const float* gy, float* ws, int N, int H, int W, int Cin, int Cout,
                   const WgPlan& p, const WinoAff* aff, hipStream_t st) {
  constexpr size_t sh = (size_t)2 * wino_wg_buf_floats<CIT, COT>() * sizeof(float);
  static_assert(sh <= 160 * 1024, "LDS budget");
  static_assert((size_t)2 * 16 * CIT * COT * 64 * 16 <= sh, "cross-wave exchange fits the staging buffers");
  dim3 grid(p.splits, Cin / (16 * CIT), Cout / (16 * COT));
  const int tiles_x = W / TW, tiles_img = tiles_x * (H / TH);
  const WinoAff av = aff ? *aff : WinoAff{};
  if (aff) {
    if (x2) return -1;
    allow_big_lds<conv_wino_wg<CIT, COT, false, true>>(sh);
    conv_wino_wg<CIT, COT, false, true><<<grid, TPB, sh, st>>>(x, nullptr, 0, gy, ws, N, H, W, Cin, Cout, tiles_x, tiles_img, p.items_per_split, av);
  } else if (x2) {
    allow_big_lds<conv_wino_wg<CIT, COT, true, false>>(sh);
    conv_wino_wg<CIT, COT, true, false><<<grid, TPB, sh, st>>>(x, x2, ca, gy, ws, N, H, W, Cin, Cout, tiles_x, tiles_img, p.items_per_split, av);
  } else {
    allow_big_lds<conv_wino_wg<CIT, COT, false, false>>(sh);
    conv_wino_wg<CIT, COT, false, false><<<grid, TPB, sh, st>>>(x, nullptr, 0, gy, ws, N, H, W, Cin, Cout, tiles_x, tiles_img, p.items_per_split, av);
  }
  return 0;
}

}  // namespace

bool smsut_wino_l_eligible(int N, int H, int W, int Kdim, int Ndim) {
  static const bool on = [] { const char* e = getenv("SMSUT_WINOGRAD_L"); return !e || atoi(e) != 0; }();
  return on && N > 0 && H >= 16 && W >= 16 && H % TH == 0 && W % TW == 0 && Kdim >= 32 && Kdim % 16 == 0 && Ndim >= 16 && Ndim % 16 == 0 &&
         (int64_t)N * H * W * (Kdim > Ndim ? Kdim : Ndim) < (1ll << 31);
}

int smsut_wino_l_launch(const float* x, const float* x2, const float* w, float* y, float* y2, int split, int N, int H, int W,
                        int Kdim, int Ndim, int transposed, float* stats, int* tiles_out, const WinoBst* bst, const WinoAff* aff,
                        const WinoSc* sc, hipStream_t st, const float* wu_in, const FinRef* fin) {
  if (!smsut_wino_l_eligible(N, H, W, Kdim, Ndim)) return -1;
  if (y2 && (split <= 0 || split >= Ndim || split % 16 != 0 || (Ndim - split) % 16 != 0 || stats || bst)) return -1;
  if (tiles_out) { *tiles_out = (W / TW) * (H / TH); return 0; }
  // output-channel slabs per workgroup: two (one input transform feeds 32 channels, one wave per SIMD) when the grid still
  // fills the chip and no slab straddles a split; SMSUT_WINO_NTN forces 1 / 2 (tuning hook)
  static const int force = [] { const char* e = getenv("SMSUT_WINO_NTN"); return e ? atoi(e) : 0; }();
  const int64_t items = (int64_t)N * (W / TW) * (H / TH);
  int ntn = (Ndim % 32 == 0 && !(y2 && split % 32 != 0) && items * (Ndim / 32) >= device_cus()) ? 2 : 1;
  if (force == 1 || (force == 2 && Ndim % 32 == 0 && !(y2 && split % 32 != 0))) ntn = force;
  // prepared weights (smsut_wino_prepare by the caller, passed with the call): copied by LDS-DMA instead of transformed per chunk
  const bool sc2 = sc && (transposed & 1);
  const float* wu = sc2 ? nullptr : wu_in;
  if (ntn == 2) return launch_ntn<2>(x, x2, w, wu, y, y2, split, N, H, W, Kdim, Ndim, transposed, stats, bst, aff, sc, st, fin);
  return launch_ntn<1>(x, x2, w, wu, y, y2, split, N, H, W, Kdim, Ndim, transposed, stats, bst, aff, sc, st, fin);
}

extern "C" {

int64_t smsut_wino_image_floats(int Kdim, int Ndim) {
  return (Kdim >= 16 && Kdim % 16 == 0 && Ndim >= 16 && Ndim % 16 == 0) ? (int64_t)16 * Kdim * Ndim : 0;
}

int smsut_wino_prepare(const float* const* w, float* const* u, const int* Kdim, const int* Ndim, const int* transposed, int count,
                       void* stream) {
  if (count < 0 || (count > 0 && (!w || !u || !Kdim || !Ndim || !transposed))) return -1;
  for (int i = 0; i < count; ++i)
    if (!w[i] || !u[i] || smsut_wino_image_floats(Kdim[i], Ndim[i]) == 0) return -1;
  for (int i0 = 0; i0 < count; i0 += PREP_MAX) {
    PrepTable t;
    t.n = count - i0 < PREP_MAX ? count - i0 : PREP_MAX;
    int blk = 0;
    for (int i = 0; i < t.n; ++i) {
      t.e[i] = PrepEntry{w[i0 + i], u[i0 + i], Kdim[i0 + i], Ndim[i0 + i], transposed[i0 + i] & 1, blk};
      blk += (Kdim[i0 + i] * (Ndim[i0 + i] / 4) + TPB - 1) / TPB;
    }
    wino_u_prepare<<<blk, TPB, 0, (hipStream_t)stream>>>(t);
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // extern "C"

bool smsut_wino_wg_eligible(int N, int H, int W, int Cin, int Cout, const float* x2, int ca) {
  // OFF by default (r03): correct (2-3e-7 of fp64) but only at parity with the direct weight-gradient kernels -- per-item staging
  // (both operands transposed into channel-major planes for 128 MFMAs per wave) costs what the 2.25x fewer MFMAs save, and the
  // split-slab pass adds to it on the 16-channel layers (profiles/r03_notes.md).  SMSUT_WINOGRAD_WG=1 enables it.
  static const bool on = [] { const char* e = getenv("SMSUT_WINOGRAD_WG"); return e && atoi(e) != 0; }();
  if (!on || N <= 0 || H < 16 || W < 16 || H % TH != 0 || W % TW != 0 || Cin < 16 || Cin % 16 != 0 || Cout < 16 || Cout % 16 != 0 ||
      (int64_t)N * H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 31))
    return false;
  if (x2) {
    const WgPlan p = plan_wino_wg(N, H, W, Cin, Cout);
    if (ca <= 0 || ca >= Cin || ca % (16 * p.cit) != 0 || (Cin - ca) % (16 * p.cit) != 0) return false;     // a slab never straddles the seam
  }
  return true;
}

int64_t smsut_wino_wg_ws(int N, int H, int W, int Cin, int Cout) {
  const int splits = plan_wino_wg(N, H, W, Cin, Cout).splits;
  return (int64_t)(splits + (splits > WG_GROUPS ? WG_GROUPS : 0)) * 16 * Cin * Cout;       // split slabs (+ the group sums)
}

int smsut_wino_wg_launch(const float* x, const float* x2, int ca, const float* gy, float* gw, float* workspace, int N, int H, int W,
                         int Cin, int Cout, const WinoAff* aff, hipStream_t st) {
  if (!smsut_wino_wg_eligible(N, H, W, Cin, Cout, x2, ca)) return -1;
  const WgPlan p = plan_wino_wg(N, H, W, Cin, Cout);
  int rc;
  if (p.cot == 2) rc = launch_wino_wg<1, 2>(x, x2, ca, gy, workspace, N, H, W, Cin, Cout, p, aff, st);
  else if (p.cit == 2) rc = launch_wino_wg<2, 1>(x, x2, ca, gy, workspace, N, H, W, Cin, Cout, p, aff, st);
  else rc = launch_wino_wg<1, 1>(x, x2, ca, gy, workspace, N, H, W, Cin, Cout, p, aff, st);
  if (rc != 0) return rc;
  const int CC = Cin * Cout;
  const float* src = workspace;
  int groups = p.splits;
  if (p.splits > WG_GROUPS) {
    const int64_t E = (int64_t)16 * CC;
    const int grp = (p.splits + WG_GROUPS - 1) / WG_GROUPS;
    groups = (p.splits + grp - 1) / grp;
    float* part2 = workspace + (size_t)p.splits * E;
    wino_wg_reduce<<<dim3((unsigned)((E + TPB - 1) / TPB), groups), TPB, 0, st>>>(workspace, part2, p.splits, grp, E);
    src = part2;
  }
  wino_wg_final<<<(CC + TPB - 1) / TPB, TPB, 0, st>>>(src, gw, groups, CC);
  return 0;
}
