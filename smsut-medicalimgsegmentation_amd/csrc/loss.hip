// Losses of the SMSUT hot path as fused kernels (fp32 data, fp64 scalar finalisation):
//   * softmax + soft-Dice statistics + cross-entropy in one pass over the logits (misc/loss.py:8-63)
//   * mean / L1 / small-batch cross-entropy / WGAN-GP norm term (uganConsisTrainer.py:130-162,
//     uganShp0Trainer.py:127-134)
//   * PatchNCE (network/patchnce.py:13-51), patch gather and L2 normalise (network/ugan.py:318-331,
//     network/networks.py:234-243)
// Logits are NHWC [N][HW][C] with C <= 32; labels int64 [N][HW].
#include "common.h"

namespace {
constexpr int TPB = 256;
constexpr int MAXC = 32;

// stats layout per group g (g = n when per-sample dice, g = 0 when batch dice):
//   part[block][G][C][3] = {tp, sum_p, count}; part_ce[block]
// CT > 0: the class count as a compile-time constant (loops unrolled, per-class arrays in registers); CT = 0: runtime C.
// (With runtime bounds the per-class arrays were indexed dynamically: 1.2 TB/s on a 59 MB pass.)
template <int CT>
__global__ void __launch_bounds__(TPB)
k_dicece_partial(const float* __restrict__ logits, const int64_t* __restrict__ labels, float* __restrict__ part,
                 float* __restrict__ part_ce, int N, int64_t HW, int C, int G) {
  __shared__ float sm4[4];
  __shared__ float acc_sm[MAXC * 3];
  constexpr int CC = CT ? CT : MAXC;
  if (CT) C = CT;
  const int n = blockIdx.y;
  const int g = G == 1 ? 0 : n;
  float tp[CC], sp[CC], cnt[CC];
#pragma unroll
  for (int c = 0; c < (CT ? CT : C); ++c) { tp[c] = 0.f; sp[c] = 0.f; cnt[c] = 0.f; }
  float ce = 0.f;
  for (int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x; p < HW; p += (int64_t)gridDim.x * TPB) {
    const float* z = logits + ((size_t)n * HW + p) * C;
    const int lab = (int)labels[(size_t)n * HW + p];
    float zc[CC];
#pragma unroll
    for (int c = 0; c < (CT ? CT : C); ++c) zc[c] = z[c];
    float m = zc[0], zl = zc[0];
#pragma unroll
    for (int c = 1; c < (CT ? CT : C); ++c) { m = fmaxf(m, zc[c]); zl = c == lab ? zc[c] : zl; }
    float e[CC], s = 0.f;
#pragma unroll
    for (int c = 0; c < (CT ? CT : C); ++c) { e[c] = __expf(zc[c] - m); s += e[c]; }
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < (CT ? CT : C); ++c) {
      const float pc = e[c] * inv;
      sp[c] += pc;
      if (c == lab) { tp[c] += pc; cnt[c] += 1.f; }
    }
    ce += (m + __logf(s)) - zl;
  }
  // block reduce each statistic
#pragma unroll
  for (int c = 0; c < (CT ? CT : C); ++c) {
    const float a = block_sum_256(tp[c], sm4);
    const float b = block_sum_256(sp[c], sm4);
    const float d = block_sum_256(cnt[c], sm4);
    if (threadIdx.x == 0) { acc_sm[c * 3] = a; acc_sm[c * 3 + 1] = b; acc_sm[c * 3 + 2] = d; }
  }
  const float cs = block_sum_256(ce, sm4);
  __syncthreads();
  const int blk = blockIdx.y * gridDim.x + blockIdx.x;
  // every block writes a full [G][C][3] slab (zeros for groups it does not own) so the finaliser is a plain sum
  for (int i = threadIdx.x; i < G * C * 3; i += TPB) {
    const int gg = i / (C * 3);
    part[(size_t)blk * G * C * 3 + i] = gg == g ? acc_sm[i % (C * 3)] : 0.f;
  }
  if (threadIdx.x == 0) part_ce[blk] = cs;
}

// stats[G][C][3] (fp32) and ce_sum[1]
// one 256-thread block per output word (GC3 statistics + 1 CE sum): fp64 tree over the per-block partials
__global__ void __launch_bounds__(TPB)
k_dicece_reduce(const float* __restrict__ part, const float* __restrict__ part_ce, int nblk, int GC3,
                float* __restrict__ stats, float* __restrict__ ce_sum) {
  __shared__ double sm4[4];
  const int i = blockIdx.x;
  double s = 0.0;
  if (i < GC3) {
    for (int b = threadIdx.x; b < nblk; b += TPB) s += (double)part[(size_t)b * GC3 + i];
  } else {
    for (int b = threadIdx.x; b < nblk; b += TPB) s += (double)part_ce[b];
  }
  s = block_sum_256_d(s, sm4);
  if (threadIdx.x == 0) {
    if (i < GC3) stats[i] = (float)s; else ce_sum[0] = (float)s;
  }
}

// loss = w_dc * (1 - mean_{g, c>=1} dc[g][c]) + w_ce * ce_sum / npix_total;  out[0]=loss, out[1]=dice, out[2]=ce
__global__ void k_dicece_final(const float* __restrict__ stats, const float* __restrict__ ce_sum, int G, int C,
                               double npix_total, float w_dc, float w_ce, float smooth, float eps,
                               float* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  double acc = 0.0;
  for (int g = 0; g < G; ++g)
    for (int c = 1; c < C; ++c) {
      const float* s = stats + ((size_t)g * C + c) * 3;
      const double tp = s[0], sp = s[1], cn = s[2];
      // 2tp + fp + fn = sum_p + count (misc/loss.py:33-35,54-55)
      acc += (2.0 * tp + smooth) / (sp + cn + (double)smooth + (double)eps);
    }
  const double dice = 1.0 - acc / (double)(G * (C - 1));
  const double ce = (double)ce_sum[0] / npix_total;
  out[0] = (float)(w_dc * dice + w_ce * ce);
  out[1] = (float)dice;
  out[2] = (float)ce;
}

template <int CT>
__global__ void __launch_bounds__(TPB)
k_dicece_bwd(const float* __restrict__ logits, const int64_t* __restrict__ labels, const float* __restrict__ stats,
             const float* __restrict__ gout, float* __restrict__ glogits, int N, int64_t HW, int C, int G,
             double npix_total, float w_dc, float w_ce, float smooth, float eps) {
  constexpr int CC = CT ? CT : MAXC;
  if (CT) C = CT;
  __shared__ float A[MAXC], Bc[MAXC];
  const int n = blockIdx.y;
  const int g = G == 1 ? 0 : n;
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    const float* s = stats + ((size_t)g * C + c) * 3;
    const double den = (double)s[1] + (double)s[2] + (double)smooth + (double)eps;
    const double k = c == 0 ? 0.0 : (double)w_dc / (double)(G * (C - 1));
    // dL/dp_c = -k * (2*onehot/den - (2tp+smooth)/den^2)
    A[c] = (float)(k * 2.0 / den);
    Bc[c] = (float)(k * (2.0 * (double)s[0] + (double)smooth) / (den * den));
  }
  __syncthreads();
  const float go = gout[0];
  const float cew = (float)((double)w_ce / npix_total);
  for (int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x; p < HW; p += (int64_t)gridDim.x * TPB) {
    const float* z = logits + ((size_t)n * HW + p) * C;
    float* gz = glogits + ((size_t)n * HW + p) * C;
    const int lab = (int)labels[(size_t)n * HW + p];
    float pr[CC], s = 0.f;
#pragma unroll
    for (int c = 0; c < (CT ? CT : C); ++c) pr[c] = z[c];
    float m = pr[0];
#pragma unroll
    for (int c = 1; c < (CT ? CT : C); ++c) m = fmaxf(m, pr[c]);
#pragma unroll
    for (int c = 0; c < (CT ? CT : C); ++c) { pr[c] = __expf(pr[c] - m); s += pr[c]; }
    const float inv = 1.f / s;
    float dot = 0.f, dp[CC];
#pragma unroll
    for (int c = 0; c < (CT ? CT : C); ++c) {
      pr[c] *= inv;
      dp[c] = Bc[c] - (c == lab ? A[c] : 0.f);
      dot += pr[c] * dp[c];
    }
#pragma unroll
    for (int c = 0; c < (CT ? CT : C); ++c)
      gz[c] = go * (pr[c] * (dp[c] - dot) + cew * (pr[c] - (c == lab ? 1.f : 0.f)));
  }
}

// ---- scalar reductions -------------------------------------------------------------------------------
// MODE 0: sum(a)   MODE 1: sum(|a-b|)   MODE 2: per-row sum(a^2) (rows = blockIdx.y)
template <int MODE>
__global__ void __launch_bounds__(TPB)
k_sum_partial(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part, int64_t n) {
  __shared__ float sm4[4];
  const int64_t base = (int64_t)blockIdx.y * n;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const float v = a[base + i];
    if (MODE == 0) acc += v;
    else if (MODE == 1) acc += fabsf(v - b[base + i]);
    else acc += v * v;
  }
  const float t = block_sum_256(acc, sm4);
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = t;
}

// out[0] = scale * sum(part[0..n))
__global__ void k_sum_final(const float* __restrict__ part, int n, double scale, float* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += (double)part[i];
  out[0] = (float)(s * scale);
}

// WGAN-GP: norms[r] = sqrt(sum_r), out = mean_r (norm-1)^2
__global__ void k_gp_final(const float* __restrict__ part, int rows, int nblk, float* __restrict__ norms,
                           float* __restrict__ out) {
  if (threadIdx.x || blockIdx.x) return;
  double acc = 0.0;
  for (int r = 0; r < rows; ++r) {
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += (double)part[(size_t)r * nblk + b];
    const double nr = sqrt(s);
    norms[r] = (float)nr;
    acc += (nr - 1.0) * (nr - 1.0);
  }
  out[0] = (float)(acc / rows);
}

// g_dydx[r][i] = gout * 2*(norm_r - 1)/(rows*norm_r) * dydx[r][i]
__global__ void __launch_bounds__(TPB)
k_gp_bwd(const float* __restrict__ dydx, const float* __restrict__ norms, const float* __restrict__ gout,
         float* __restrict__ g, int rows, int64_t n) {
  const int r = blockIdx.y;
  const float nr = norms[r];
  const float f = gout[0] * 2.f * (nr - 1.f) / ((float)rows * nr);
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
    g[(size_t)r * n + i] = f * dydx[(size_t)r * n + i];
}

// ga = gout*sign(a-b)/n ; gb = -ga
__global__ void __launch_bounds__(TPB)
k_l1_bwd(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gout,
         float* __restrict__ ga, float* __restrict__ gb, int64_t n) {
  const float f = gout[0] / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB) {
    const float d = a[i] - b[i];
    const float s = d > 0.f ? f : (d < 0.f ? -f : 0.f);
    if (ga) ga[i] = s;
    if (gb) gb[i] = -s;
  }
}

// small-batch cross-entropy (mean) over rows of logits [B][C] with int64 targets; one block
__global__ void k_ce_rows_fwd(const float* __restrict__ z, const int64_t* __restrict__ tgt, int B, int C,
                              float* __restrict__ out) {
  __shared__ float sm4[4];
  float acc = 0.f;
  for (int r = threadIdx.x; r < B; r += TPB) {
    const float* zr = z + (size_t)r * C;
    float m = zr[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, zr[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(zr[c] - m);
    acc += (m + logf(s)) - zr[(int)tgt[r]];
  }
  const float t = block_sum_256(acc, sm4);
  if (threadIdx.x == 0) out[0] = t / (float)B;
}
__global__ void k_ce_rows_bwd(const float* __restrict__ z, const int64_t* __restrict__ tgt,
                              const float* __restrict__ gout, int B, int C, float* __restrict__ gz) {
  const float f = gout[0] / (float)B;
  for (int r = blockIdx.x * TPB + threadIdx.x; r < B; r += gridDim.x * TPB) {
    const float* zr = z + (size_t)r * C;
    float m = zr[0];
    for (int c = 1; c < C; ++c) m = fmaxf(m, zr[c]);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(zr[c] - m);
    const int t = (int)tgt[r];
    for (int c = 0; c < C; ++c) gz[(size_t)r * C + c] = f * (expf(zr[c] - m) / s - (c == t ? 1.f : 0.f));
  }
}

// ---- patch gather, L2 normalise, PatchNCE ------------------------------------------------------------
// out[(b*P + j)][c] = feat[b][ids[j]][c]   (NHWC makes a patch one contiguous row)
__global__ void __launch_bounds__(TPB)
k_gather_rows(const float* __restrict__ feat, const int64_t* __restrict__ ids, float* __restrict__ out, int B,
              int64_t HW, int C, int P) {
  const int64_t total = (int64_t)B * P * C;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int j = (int)(r % P);
    const int b = (int)(r / P);
    out[i] = feat[((size_t)b * HW + (size_t)ids[j]) * C + c];
  }
}
// gfeat must be zeroed first; ids are unique within an image (randperm) so plain stores are race-free
__global__ void __launch_bounds__(TPB)
k_scatter_rows(const float* __restrict__ gout, const int64_t* __restrict__ ids, float* __restrict__ gfeat, int B,
               int64_t HW, int C, int P) {
  const int64_t total = (int64_t)B * P * C;
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
    const int c = (int)(i % C);
    const int64_t r = i / C;
    const int j = (int)(r % P);
    const int b = (int)(r / P);
    gfeat[((size_t)b * HW + (size_t)ids[j]) * C + c] = gout[i];
  }
}

__global__ void __launch_bounds__(TPB) k_zero_f32(float* __restrict__ p, int64_t n) {
  const int64_t n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < n4; i += (int64_t)gridDim.x * TPB)
    p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) p[(n4 << 2) + threadIdx.x] = 0.f;
}

// y = x / (||x||_2 + 1e-7) per row; one wave per row
__global__ void __launch_bounds__(TPB)
k_l2norm_fwd(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ norms, int rows, int C) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) { const float v = x[(size_t)r * C + c]; s += v * v; }
  s = wave_sum(s);
  const float nr = sqrtf(s);
  const float inv = 1.f / (nr + 1e-7f);
  for (int c = lane; c < C; c += 64) y[(size_t)r * C + c] = x[(size_t)r * C + c] * inv;
  if (lane == 0) norms[r] = nr;
}
// gx = gy/(n+e) - x * (sum(gy*x) / (n * (n+e)^2))
__global__ void __launch_bounds__(TPB)
k_l2norm_bwd(const float* __restrict__ gy, const float* __restrict__ x, const float* __restrict__ norms,
             float* __restrict__ gx, int rows, int C) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  float d = 0.f;
  for (int c = lane; c < C; c += 64) d += gy[(size_t)r * C + c] * x[(size_t)r * C + c];
  d = wave_sum(d);
  const float nr = norms[r];
  const float e = nr + 1e-7f;
  const float k = nr > 0.f ? d / (nr * e * e) : 0.f;
  for (int c = lane; c < C; c += 64) gx[(size_t)r * C + c] = gy[(size_t)r * C + c] / e - x[(size_t)r * C + c] * k;
}

// PatchNCE forward: one block per query row i (group g = i / np).  logits = [q_i.k_i, q_i.k_j (diag -> -10)] / T
// loss_i = logsumexp(logits) - logits[0];  also stores softmax probs [rows][np+1] for the backward.
__global__ void __launch_bounds__(TPB)
k_nce_fwd(const float* __restrict__ q, const float* __restrict__ k, float* __restrict__ loss, float* __restrict__ probs,
          int np, int dim, float invT) {
  extern __shared__ float sm[];           // q row [dim] + logits [np+1]
  __shared__ float sm4[4];
  float* qs = sm;
  float* lg = sm + dim;
  const int i = blockIdx.x;
  const int g0 = (i / np) * np;            // first row of this group
  for (int d = threadIdx.x; d < dim; d += TPB) qs[d] = q[(size_t)i * dim + d];
  __syncthreads();
  for (int j = threadIdx.x; j <= np; j += TPB) {
    const int kr = j == 0 ? i : g0 + (j - 1);
    const float* kp = k + (size_t)kr * dim;
    float acc = 0.f;
    for (int d = 0; d < dim; ++d) acc = fmaf(qs[d], kp[d], acc);
    if (j > 0 && kr == i) acc = -10.f;     // masked diagonal (patchnce.py:41-42)
    lg[j] = acc * invT;
  }
  __syncthreads();
  float m = -INFINITY;
  for (int j = threadIdx.x; j <= np; j += TPB) m = fmaxf(m, lg[j]);
  m = block_max_256(m, sm4);
  float s = 0.f;
  for (int j = threadIdx.x; j <= np; j += TPB) s += expf(lg[j] - m);
  s = block_sum_256(s, sm4);
  for (int j = threadIdx.x; j <= np; j += TPB) probs[(size_t)i * (np + 1) + j] = expf(lg[j] - m) / s;
  if (threadIdx.x == 0) loss[i] = (m + logf(s)) - lg[0];
}
// gq_i = gloss_i/T * [ (p0-1) k_i + sum_{j != i in group} p_j k_j ]   (k is detached, patchnce.py:16)
__global__ void __launch_bounds__(TPB)
k_nce_bwd(const float* __restrict__ gloss, const float* __restrict__ probs, const float* __restrict__ k,
          float* __restrict__ gq, int np, int dim, float invT) {
  extern __shared__ float co[];            // coefficient per key row of the group
  const int i = blockIdx.x;
  const int g0 = (i / np) * np;
  const float f = gloss[i] * invT;
  const float* pr = probs + (size_t)i * (np + 1);
  for (int j = threadIdx.x; j < np; j += TPB) {
    const int kr = g0 + j;
    co[j] = kr == i ? f * (pr[0] - 1.f) : f * pr[j + 1];
  }
  __syncthreads();
  for (int d = threadIdx.x; d < dim; d += TPB) {
    float acc = 0.f;
    for (int j = 0; j < np; ++j) acc = fmaf(co[j], k[(size_t)(g0 + j) * dim + d], acc);
    gq[(size_t)i * dim + d] = acc;
  }
}

// ---- mean-teacher consistency: mean((softmax(a) - softmax(b))^2) over all elements (reference
// trainer/meanTeacherTrainer.py:113-131), a and b NHWC logits [P pixels][C]; one thread per pixel, the C-vector in registers.
__global__ void __launch_bounds__(TPB)
k_softmax_mse_partial(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part, int64_t P, int C) {
  __shared__ float sm4[4];
  float acc = 0.f;
  for (int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x; p < P; p += (int64_t)gridDim.x * TPB) {
    float pa[MAXC], pb[MAXC];
    float ma = -INFINITY, mb = -INFINITY;
    for (int c = 0; c < C; ++c) { pa[c] = a[p * C + c]; pb[c] = b[p * C + c]; ma = fmaxf(ma, pa[c]); mb = fmaxf(mb, pb[c]); }
    float sa = 0.f, sb = 0.f;
    for (int c = 0; c < C; ++c) { pa[c] = __expf(pa[c] - ma); pb[c] = __expf(pb[c] - mb); sa += pa[c]; sb += pb[c]; }
    const float ia = 1.f / sa, ib = 1.f / sb;
    for (int c = 0; c < C; ++c) { const float d = pa[c] * ia - pb[c] * ib; acc += d * d; }
  }
  const float t = block_sum_256(acc, sm4);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// ga[p][k] = gout * (2 / (P*C)) * pa_k * ((pa_k - pb_k) - sum_c pa_c (pa_c - pb_c))    (b carries no gradient: EMA teacher)
__global__ void __launch_bounds__(TPB)
k_softmax_mse_bwd(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gout,
                  float* __restrict__ ga, int64_t P, int C) {
  const float scale = gout[0] * 2.f / ((float)P * (float)C);
  for (int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x; p < P; p += (int64_t)gridDim.x * TPB) {
    float pa[MAXC], pb[MAXC];
    float ma = -INFINITY, mb = -INFINITY;
    for (int c = 0; c < C; ++c) { pa[c] = a[p * C + c]; pb[c] = b[p * C + c]; ma = fmaxf(ma, pa[c]); mb = fmaxf(mb, pb[c]); }
    float sa = 0.f, sb = 0.f;
    for (int c = 0; c < C; ++c) { pa[c] = __expf(pa[c] - ma); pb[c] = __expf(pb[c] - mb); sa += pa[c]; sb += pb[c]; }
    const float ia = 1.f / sa, ib = 1.f / sb;
    float dot = 0.f;
    for (int c = 0; c < C; ++c) { pa[c] *= ia; pb[c] = pa[c] - pb[c] * ib; dot += pa[c] * pb[c]; }
    for (int c = 0; c < C; ++c) ga[p * C + c] = scale * pa[c] * (pb[c] - dot);
  }
}

// labels[p] = argmax_c logits[p][c] (first maximum, as torch.argmax): pseudo labels of the consistency / cross-pseudo
// losses (uganConsisTrainer.py:45-53, crossPseTrainer.py:122-127) and the prediction map of validate_epoch
__global__ void __launch_bounds__(TPB)
k_argmax_channels(const float* __restrict__ z, int64_t* __restrict__ out, int64_t P, int C) {
  for (int64_t p = (int64_t)blockIdx.x * TPB + threadIdx.x; p < P; p += (int64_t)gridDim.x * TPB) {
    float best = z[p * C];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
      const float v = z[p * C + c];
      if (v > best || (v != v && best == best)) { best = v; bi = c; }     // NaN wins, like torch
    }
    out[p] = bi;
  }
}

inline int pix_blocks(int64_t HW) {
  int64_t b = cdiv64(HW, TPB * 4);
  if (b > 64) b = 64;
  if (b < 1) b = 1;
  return (int)b;
}
inline int sum_blocks(int64_t n) {
  int64_t b = cdiv64(n, TPB * 8);
  if (b > 512) b = 512;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" {
#define ST ((hipStream_t)stream)

// workspace floats for dice/ce: nblk*(G*C*3 + 1)
int64_t smsut_dicece_ws(int N, int64_t HW, int C, int G) { return (int64_t)N * pix_blocks(HW) * (G * C * 3 + 1); }

// Stage 1: stats[G][C][3] = {tp, sum_p, count} and ce_sum[1].  G = 1 (batch dice) or N (per-sample dice).
// Under data parallelism the caller all-reduces stats / ce_sum between stage 1 and stage 2.
int smsut_dicece_stats(const float* logits, const int64_t* labels, float* stats, float* ce_sum, float* workspace, int N,
                       int64_t HW, int C, int G, void* stream) {
  SMSUT_REQUIRE(logits && labels && stats && ce_sum && workspace && N > 0 && HW > 0 && C >= 2 && C <= MAXC &&
                (G == 1 || G == N));
  const int pb = pix_blocks(HW);
  const int nblk = N * pb;
  float* part = workspace;
  float* part_ce = workspace + (size_t)nblk * G * C * 3;
#define DICE_P(CT) k_dicece_partial<CT><<<dim3(pb, N), TPB, 0, ST>>>(logits, labels, part, part_ce, N, HW, C, G)
  switch (C) { case 2: DICE_P(2); break; case 3: DICE_P(3); break; case 4: DICE_P(4); break; case 5: DICE_P(5); break;
               default: DICE_P(0); }
#undef DICE_P
  k_dicece_reduce<<<G * C * 3 + 1, TPB, 0, ST>>>(part, part_ce, nblk, G * C * 3, stats, ce_sum);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// Stage 2: out[0] = w_dc*dice + w_ce*ce, out[1] = dice loss, out[2] = ce.  npix_total = (global) N*HW.
int smsut_dicece_final(const float* stats, const float* ce_sum, float* out, int G, int C, double npix_total, float w_dc,
                       float w_ce, void* stream) {
  SMSUT_REQUIRE(stats && ce_sum && out && G > 0 && C >= 2 && npix_total > 0);
  k_dicece_final<<<1, 64, 0, ST>>>(stats, ce_sum, G, C, npix_total, w_dc, w_ce, 1e-5f, 1e-8f, out);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_dicece_bwd(const float* logits, const int64_t* labels, const float* stats, const float* gout, float* glogits,
                     int N, int64_t HW, int C, int G, double npix_total, float w_dc, float w_ce, void* stream) {
  SMSUT_REQUIRE(logits && labels && stats && gout && glogits && N > 0 && HW > 0 && C >= 2 && C <= MAXC);
#define DICE_B(CT) k_dicece_bwd<CT><<<dim3(pix_blocks(HW) * 4, N), TPB, 0, ST>>>(logits, labels, stats, gout, glogits, N, HW, C, G, \
                                                                          npix_total, w_dc, w_ce, 1e-5f, 1e-8f)
  switch (C) { case 2: DICE_B(2); break; case 3: DICE_B(3); break; case 4: DICE_B(4); break; case 5: DICE_B(5); break;
               default: DICE_B(0); }
#undef DICE_B
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}

int64_t smsut_sum_ws(int64_t n, int rows) { return (int64_t)rows * sum_blocks(n); }

// out = scale * sum(a)   (mean: scale = 1/n; -mean: scale = -1/n)
int smsut_sum(const float* a, float* out, float* workspace, int64_t n, double scale, void* stream) {
  SMSUT_REQUIRE(a && out && workspace && n > 0);
  const int nb = sum_blocks(n);
  k_sum_partial<0><<<dim3(nb, 1), TPB, 0, ST>>>(a, nullptr, workspace, n);
  k_sum_final<<<1, 64, 0, ST>>>(workspace, nb, scale, out);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// out = mean |a - b|
int smsut_l1_fwd(const float* a, const float* b, float* out, float* workspace, int64_t n, void* stream) {
  SMSUT_REQUIRE(a && b && out && workspace && n > 0);
  const int nb = sum_blocks(n);
  k_sum_partial<1><<<dim3(nb, 1), TPB, 0, ST>>>(a, b, workspace, n);
  k_sum_final<<<1, 64, 0, ST>>>(workspace, nb, 1.0 / (double)n, out);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_l1_bwd(const float* a, const float* b, const float* gout, float* ga, float* gb, int64_t n, void* stream) {
  SMSUT_REQUIRE(a && b && gout && (ga || gb) && n > 0);
  k_l1_bwd<<<ew_grid(n), TPB, 0, ST>>>(a, b, gout, ga, gb, n);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// WGAN-GP: out = mean_r (||dydx_r||_2 - 1)^2, norms[rows] saved for the backward
// out = mean((softmax_c(a) - softmax_c(b))^2); a, b [P][C] NHWC logits; workspace: smsut_sum_ws(P, 1) floats
int smsut_softmax_mse_fwd(const float* a, const float* b, float* out, float* workspace, int64_t P, int C, void* stream) {
  SMSUT_REQUIRE(a && b && out && workspace && P > 0 && C > 0 && C <= MAXC);
  const int nb = sum_blocks(P);
  k_softmax_mse_partial<<<nb, TPB, 0, ST>>>(a, b, workspace, P, C);
  k_sum_final<<<1, 64, 0, ST>>>(workspace, nb, 1.0 / ((double)P * C), out);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_softmax_mse_bwd(const float* a, const float* b, const float* gout, float* ga, int64_t P, int C, void* stream) {
  SMSUT_REQUIRE(a && b && gout && ga && P > 0 && C > 0 && C <= MAXC);
  k_softmax_mse_bwd<<<ew_grid(P), TPB, 0, ST>>>(a, b, gout, ga, P, C);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_argmax_channels(const float* z, int64_t* out, int64_t P, int C, void* stream) {
  SMSUT_REQUIRE(z && out && P > 0 && C > 0);
  k_argmax_channels<<<ew_grid(P), TPB, 0, ST>>>(z, out, P, C);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// WGAN-GP (kept below)
int smsut_gp_fwd(const float* dydx, float* out, float* norms, float* workspace, int rows, int64_t n, void* stream) {
  SMSUT_REQUIRE(dydx && out && norms && workspace && rows > 0 && n > 0);
  const int nb = sum_blocks(n);
  k_sum_partial<2><<<dim3(nb, rows), TPB, 0, ST>>>(dydx, nullptr, workspace, n);
  k_gp_final<<<1, 64, 0, ST>>>(workspace, rows, nb, norms, out);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_gp_bwd(const float* dydx, const float* norms, const float* gout, float* g, int rows, int64_t n, void* stream) {
  SMSUT_REQUIRE(dydx && norms && gout && g && rows > 0 && n > 0);
  k_gp_bwd<<<dim3(sum_blocks(n), rows), TPB, 0, ST>>>(dydx, norms, gout, g, rows, n);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_ce_rows_fwd(const float* z, const int64_t* tgt, float* out, int B, int C, void* stream) {
  SMSUT_REQUIRE(z && tgt && out && B > 0 && C > 0);
  k_ce_rows_fwd<<<1, TPB, 0, ST>>>(z, tgt, B, C, out);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_ce_rows_bwd(const float* z, const int64_t* tgt, const float* gout, float* gz, int B, int C, void* stream) {
  SMSUT_REQUIRE(z && tgt && gout && gz && B > 0 && C > 0);
  k_ce_rows_bwd<<<(B + TPB - 1) / TPB, TPB, 0, ST>>>(z, tgt, gout, B, C, gz);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_gather_rows(const float* feat, const int64_t* ids, float* out, int B, int64_t HW, int C, int P, void* stream) {
  SMSUT_REQUIRE(feat && ids && out && B > 0 && HW > 0 && C > 0 && P > 0);
  k_gather_rows<<<ew_grid((int64_t)B * P * C), TPB, 0, ST>>>(feat, ids, out, B, HW, C, P);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_scatter_rows(const float* gout, const int64_t* ids, float* gfeat, int B, int64_t HW, int C, int P,
                       void* stream) {
  SMSUT_REQUIRE(gout && ids && gfeat && B > 0 && HW > 0 && C > 0 && P > 0);
  // zero-fill by a KERNEL, not hipMemsetAsync: under stream capture the memset became a hipGraph memset node that was
  // not ordered against the kernel nodes around it on replay (r01: the un-sampled rows of d/dfeat kept whatever the
  // block held before -- 1e10-scale garbage into enc5 / tsl_encoder gradients from the first replayed iteration on)
  k_zero_f32<<<ew_grid(((int64_t)B * HW * C + 3) / 4), TPB, 0, ST>>>(gfeat, (int64_t)B * HW * C);
  k_scatter_rows<<<ew_grid((int64_t)B * P * C), TPB, 0, ST>>>(gout, ids, gfeat, B, HW, C, P);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_l2norm_fwd(const float* x, float* y, float* norms, int rows, int C, void* stream) {
  SMSUT_REQUIRE(x && y && norms && rows > 0 && C > 0);
  k_l2norm_fwd<<<(rows + 3) / 4, TPB, 0, ST>>>(x, y, norms, rows, C);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_l2norm_bwd(const float* gy, const float* x, const float* norms, float* gx, int rows, int C, void* stream) {
  SMSUT_REQUIRE(gy && x && norms && gx && rows > 0 && C > 0);
  k_l2norm_bwd<<<(rows + 3) / 4, TPB, 0, ST>>>(gy, x, norms, gx, rows, C);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// q, k: [rows][dim]; groups of np consecutive rows (np = rows / PatchNCELoss.batch_size); probs: [rows][np+1]
int smsut_patchnce_fwd(const float* q, const float* k, float* loss, float* probs, int rows, int np, int dim, float T,
                       void* stream) {
  SMSUT_REQUIRE(q && k && loss && probs && rows > 0 && np > 0 && rows % np == 0 && dim > 0 && T > 0.f);
  const size_t sh = (size_t)(dim + np + 1) * sizeof(float);
  SMSUT_REQUIRE(sh <= 64 * 1024);
  k_nce_fwd<<<rows, TPB, sh, ST>>>(q, k, loss, probs, np, dim, 1.f / T);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_patchnce_bwd(const float* gloss, const float* probs, const float* k, float* gq, int rows, int np, int dim,
                       float T, void* stream) {
  SMSUT_REQUIRE(gloss && probs && k && gq && rows > 0 && np > 0 && rows % np == 0 && dim > 0 && T > 0.f);
  SMSUT_REQUIRE((size_t)np * sizeof(float) <= 64 * 1024);
  k_nce_bwd<<<rows, TPB, (size_t)np * sizeof(float), ST>>>(gloss, probs, k, gq, np, dim, 1.f / T);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}

}  // extern "C"
