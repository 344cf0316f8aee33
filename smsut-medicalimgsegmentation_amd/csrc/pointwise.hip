// Memory-bound NHWC fp32 helpers of the SMSUT hot path: activations, residual add, pooling,
// bilinear x2, channel concat / split, modality planes.  All are grid-stride, 16 B/lane where the
// channel count allows (guide G13), and launch on the caller's stream.
#include "common.h"

namespace {
constexpr int TPB = 256;

#define GRID_STRIDE(i, total) \
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < (total); i += (int64_t)gridDim.x * TPB)

// y = lrelu(a (+ b)), vectorised when n4 covers the tensor
__global__ void __launch_bounds__(TPB)
k_add_act(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n, float slope) {
  const int64_t n4 = n >> 2;
  GRID_STRIDE(i, n4) {
    float4 v = ((const float4*)a)[i];
    if (b) { const float4 u = ((const float4*)b)[i]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    v.x = lrelu_f(v.x, slope); v.y = lrelu_f(v.y, slope); v.z = lrelu_f(v.z, slope); v.w = lrelu_f(v.w, slope);
    ((float4*)y)[i] = v;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
    y[i] = lrelu_f(a[i] + (b ? b[i] : 0.f), slope);
}

// gx = gy * lrelu'(y)
__global__ void __launch_bounds__(TPB)
k_act_bwd(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gx, int64_t n, float slope) {
  const int64_t n4 = n >> 2;
  GRID_STRIDE(i, n4) {
    float4 g = ((const float4*)gy)[i];
    const float4 v = ((const float4*)y)[i];
    g.x *= lrelu_mask(v.x, slope); g.y *= lrelu_mask(v.y, slope);
    g.z *= lrelu_mask(v.z, slope); g.w *= lrelu_mask(v.w, slope);
    ((float4*)gx)[i] = g;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TPB)
    gx[i] = gy[i] * lrelu_mask(y[i], slope);
}

__global__ void __launch_bounds__(TPB)
k_tanh_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  GRID_STRIDE(i, n) y[i] = tanhf(x[i]);
}
__global__ void __launch_bounds__(TPB)
k_tanh_bwd(const float* __restrict__ gy, const float* __restrict__ y, float* __restrict__ gx, int64_t n) {
  GRID_STRIDE(i, n) { const float t = y[i]; gx[i] = gy[i] * (1.f - t * t); }
}

// y[r][c] = x[r][c] + bias[c]
__global__ void __launch_bounds__(TPB)
k_bias_add(const float* __restrict__ x, const float* __restrict__ bias, float* __restrict__ y, int64_t n, int C) {
  GRID_STRIDE(i, n) y[i] = x[i] + bias[(int)(i % C)];
}

// out = alpha*a + beta*b  (optionally per-row alpha: x_hat of WGAN-GP, uganConsisTrainer.py:139)
__global__ void __launch_bounds__(TPB)
k_row_lerp(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ alpha,
           float* __restrict__ out, int64_t n, int64_t row) {
  GRID_STRIDE(i, n) { const float t = alpha[i / row]; out[i] = t * a[i] + (1.f - t) * b[i]; }
}

__global__ void __launch_bounds__(TPB)
k_fill(float* __restrict__ out, float v, int64_t n) { GRID_STRIDE(i, n) out[i] = v; }

__global__ void __launch_bounds__(TPB)
k_scale(const float* __restrict__ x, const float* __restrict__ s, float mul, float* __restrict__ out, int64_t n) {
  const float f = (s ? s[0] : 1.f) * mul;
  GRID_STRIDE(i, n) out[i] = x[i] * f;
}

// VEC channels of one pixel as one access (VEC = 4 when C % 4 == 0: the index arithmetic -- three div/mod -- and the
// memory instructions are then per float4, not per float; these kernels were instruction-bound, profiles/r01_notes.md)
template <int VEC> __device__ __forceinline__ void ldp(const float* __restrict__ p, float (&o)[VEC]) {
  if constexpr (VEC == 4) *(float4*)o = *(const float4*)p; else o[0] = p[0];
}
template <int VEC> __device__ __forceinline__ void stp(float* __restrict__ p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) *(float4*)p = *(const float4*)v; else p[0] = v[0];
}

// ---- 2x2 stride-2 pooling (H, W even) ------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(TPB)
k_maxpool_fwd(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, CV = C / VEC;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * VEC;
    int64_t p = i / CV;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const float* b = x + (((size_t)n * H + 2 * ho) * W + 2 * wo) * C + c;
    // scan order (0,0),(0,1),(1,0),(1,1); NaN propagates like at::max_pool2d
    float m[VEC], v[VEC];
    ldp<VEC>(b, m);
    ldp<VEC>(b + C, v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) if (v[j] > m[j] || v[j] != v[j]) m[j] = v[j];
    ldp<VEC>(b + (size_t)W * C, v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) if (v[j] > m[j] || v[j] != v[j]) m[j] = v[j];
    ldp<VEC>(b + (size_t)W * C + C, v);
#pragma unroll
    for (int j = 0; j < VEC; ++j) if (v[j] > m[j] || v[j] != v[j]) m[j] = v[j];
    stp<VEC>(y + i * VEC, m);
  }
}

// gradient goes to the first element (scan order) equal to the pooled max -- at::max_pool2d's argmax
template <int VEC>
__global__ void __launch_bounds__(TPB)
k_maxpool_bwd(const float* __restrict__ gy, const float* __restrict__ x, float* __restrict__ gx, int N, int H, int W,
              int C, const float* __restrict__ add = nullptr) {
  // add (nullable): a second gradient of x (the skip connection of the encoder level, network/blocks.py:131-133) summed in
  // the same pass -- replaces autograd's separate accumulation kernel (read 2, write 1 full-resolution tensors)
  const int Ho = H >> 1, Wo = W >> 1, CV = C / VEC;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * VEC;
    int64_t p = i / CV;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const size_t o = (((size_t)n * H + 2 * ho) * W + 2 * wo) * C + c;
    const size_t o1 = o + C, o2 = o + (size_t)W * C, o3 = o2 + C;
    float v0[VEC], v1[VEC], v2[VEC], v3[VEC], g[VEC];
    ldp<VEC>(x + o, v0); ldp<VEC>(x + o1, v1); ldp<VEC>(x + o2, v2); ldp<VEC>(x + o3, v3);
    ldp<VEC>(gy + i * VEC, g);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      int k = 0; float m = v0[j];
      if (v1[j] > m || v1[j] != v1[j]) { m = v1[j]; k = 1; }
      if (v2[j] > m || v2[j] != v2[j]) { m = v2[j]; k = 2; }
      if (v3[j] > m || v3[j] != v3[j]) { m = v3[j]; k = 3; }
      v0[j] = k == 0 ? g[j] : 0.f; v1[j] = k == 1 ? g[j] : 0.f; v2[j] = k == 2 ? g[j] : 0.f; v3[j] = k == 3 ? g[j] : 0.f;
    }
    if (add) {
      float a0[VEC], a1[VEC], a2[VEC], a3[VEC];
      ldp<VEC>(add + o, a0); ldp<VEC>(add + o1, a1); ldp<VEC>(add + o2, a2); ldp<VEC>(add + o3, a3);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { v0[j] += a0[j]; v1[j] += a1[j]; v2[j] += a2[j]; v3[j] += a3[j]; }
    }
    stp<VEC>(gx + o, v0); stp<VEC>(gx + o1, v1); stp<VEC>(gx + o2, v2); stp<VEC>(gx + o3, v3);
  }
}

template <int VEC>
__global__ void __launch_bounds__(TPB)
k_avgpool_fwd(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, CV = C / VEC;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * VEC;
    int64_t p = i / CV;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const float* b = x + (((size_t)n * H + 2 * ho) * W + 2 * wo) * C + c;
    float a0[VEC], a1[VEC], a2[VEC], a3[VEC];
    ldp<VEC>(b, a0); ldp<VEC>(b + C, a1); ldp<VEC>(b + (size_t)W * C, a2); ldp<VEC>(b + (size_t)W * C + C, a3);
#pragma unroll
    for (int j = 0; j < VEC; ++j) a0[j] = (a0[j] + a1[j] + a2[j] + a3[j]) * 0.25f;
    stp<VEC>(y + i * VEC, a0);
  }
}

// gx[n,h,w,c] = 0.25 * gy[n,h/2,w/2,c]   (adjoint of avgpool; its own adjoint is avgpool again)
template <int VEC>
__global__ void __launch_bounds__(TPB)
k_avgpool_bwd(const float* __restrict__ gy, float* __restrict__ gx, int N, int H, int W, int C) {
  const int Ho = H >> 1, Wo = W >> 1, CV = C / VEC;
  const int64_t total = (int64_t)N * H * W * CV;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * VEC;
    int64_t p = i / CV;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    float g[VEC];
    ldp<VEC>(gy + (((size_t)n * Ho + (h >> 1)) * Wo + (w >> 1)) * C + c, g);
#pragma unroll
    for (int j = 0; j < VEC; ++j) g[j] *= 0.25f;
    stp<VEC>(gx + i * VEC, g);
  }
}

// ---- bilinear x2, align_corners=False (nn.Upsample, network/blocks.py:44) ---------------------------
// src = max(0.5*(dst+0.5)-0.5, 0); i0 = floor(src); i1 = min(i0+1, size-1); lambda = src - i0
__device__ __forceinline__ void bil_src(int d, int size, int& i0, int& i1, float& l1) {
  float s = 0.5f * ((float)d + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  i1 = i0 + (i0 < size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

template <int VEC>
__global__ void __launch_bounds__(TPB)
k_bilinear2_fwd(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
  const int Ho = 2 * H, Wo = 2 * W, CV = C / VEC;
  const int64_t total = (int64_t)N * Ho * Wo * CV;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * VEC;
    int64_t p = i / CV;
    const int wo = (int)(p % Wo); p /= Wo;
    const int ho = (int)(p % Ho);
    const int n = (int)(p / Ho);
    int h0, h1, w0, w1; float lh, lw;
    bil_src(ho, H, h0, h1, lh);
    bil_src(wo, W, w0, w1, lw);
    const float* b = x + (size_t)n * H * W * C + c;
    float v00[VEC], v01[VEC], v10[VEC], v11[VEC];
    ldp<VEC>(b + ((size_t)h0 * W + w0) * C, v00); ldp<VEC>(b + ((size_t)h0 * W + w1) * C, v01);
    ldp<VEC>(b + ((size_t)h1 * W + w0) * C, v10); ldp<VEC>(b + ((size_t)h1 * W + w1) * C, v11);
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      v00[j] = (1.f - lh) * ((1.f - lw) * v00[j] + lw * v01[j]) + lh * ((1.f - lw) * v10[j] + lw * v11[j]);
    stp<VEC>(y + i * VEC, v00);
  }
}

// adjoint in gather form: each input pixel collects from the <=4x4 outputs that reference it
template <int VEC>
__global__ void __launch_bounds__(TPB)
k_bilinear2_bwd(const float* __restrict__ gy, float* __restrict__ gx, int N, int H, int W, int C) {
  const int Ho = 2 * H, Wo = 2 * W, CV = C / VEC;
  const int64_t total = (int64_t)N * H * W * CV;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % CV) * VEC;
    int64_t p = i / CV;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const int n = (int)(p / H);
    float wh[4], ww[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int oh = 2 * h - 1 + k, ow = 2 * w - 1 + k;
      wh[k] = 0.f; ww[k] = 0.f;
      if (oh >= 0 && oh < Ho) {
        int i0, i1; float l; bil_src(oh, H, i0, i1, l);
        wh[k] = (i0 == h ? 1.f - l : 0.f) + (i1 == h ? l : 0.f);
      }
      if (ow >= 0 && ow < Wo) {
        int i0, i1; float l; bil_src(ow, W, i0, i1, l);
        ww[k] = (i0 == w ? 1.f - l : 0.f) + (i1 == w ? l : 0.f);
      }
    }
    const float* b = gy + (size_t)n * Ho * Wo * C + c;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (wh[a] == 0.f) continue;
      const int oh = 2 * h - 1 + a;
      float r[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) r[j] = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (ww[k] == 0.f) continue;
        float v[VEC];
        ldp<VEC>(b + ((size_t)oh * Wo + (2 * w - 1 + k)) * C, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) r[j] += ww[k] * v[j];
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += wh[a] * r[j];
    }
    stp<VEC>(gx + i * VEC, acc);
  }
}


// ---- channel slices: dst[p][dst_off + c] = src[p][src_off + c], c < Cc --------------------------------
__global__ void __launch_bounds__(TPB)
k_copy_channels(const float* __restrict__ src, int Cs, int src_off, float* __restrict__ dst, int Cd, int dst_off,
                int Cc, int64_t P) {
  if (((Cc | Cs | Cd | src_off | dst_off) & 3) == 0) {
    const int q = Cc >> 2;
    const int64_t total = P * q;
    GRID_STRIDE(i, total) {
      const int c = (int)(i % q) << 2;
      const int64_t p = i / q;
      *(float4*)(dst + p * Cd + dst_off + c) = *(const float4*)(src + p * Cs + src_off + c);
    }
  } else {
    const int64_t total = P * Cc;
    GRID_STRIDE(i, total) {
      const int c = (int)(i % Cc);
      const int64_t p = i / Cc;
      dst[p * Cd + dst_off + c] = src[p * Cs + src_off + c];
    }
  }
}

// torch.cat([a, b], 1) / its split in ONE pass: every lane moves one float4 of a concatenated pixel row, so the wide
// tensor is read / written in full contiguous rows (two half-row copies touch every 128-B line of it twice).
// dir 0: y[p][0..Ca) = a[p], y[p][Ca..Ca+Cb) = b[p];  dir 1: the reverse (a / b nullable: that half is skipped).
__global__ void __launch_bounds__(TPB)
k_concat2(float* __restrict__ a, int Ca, float* __restrict__ b, int Cb, float* __restrict__ y, int64_t P, int dir) {
  const int q = (Ca + Cb) >> 2, qa = Ca >> 2;
  const int64_t total = P * q;
  GRID_STRIDE(i, total) {
    const int c4 = (int)(i % q);
    const int64_t p = i / q;
    float* part = c4 < qa ? (a ? a + p * Ca + 4 * c4 : nullptr) : (b ? b + p * Cb + 4 * (c4 - qa) : nullptr);
    float* wide = y + p * (Ca + Cb) + 4 * c4;
    if (!part) continue;
    if (dir == 0) *(float4*)wide = *(const float4*)part;
    else *(float4*)part = *(const float4*)wide;
  }
}

// tsl input: out[n][p][0..Cx) = x, out[n][p][Cx + j] = m[n][j]   (network/ugan.py:156-159)
__global__ void __launch_bounds__(TPB)
k_modal_planes(const float* __restrict__ x, const float* __restrict__ m, float* __restrict__ out, int N, int64_t HW,
               int Cx, int M) {
  const int Ct = Cx + M;
  const int64_t total = (int64_t)N * HW * Ct;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % Ct);
    const int64_t p = i / Ct;
    const int n = (int)(p / HW);
    out[i] = c < Cx ? x[p * Cx + c] : m[n * M + (c - Cx)];
  }
}

// ---- padding / cropping window copy and the anti-aliased (blur-pool) down-sampling of network/networks.py:37-60 ------
// mode 0 zero, 1 reflect (nn.ReflectionPad2d), 2 replicate (nn.ReplicationPad2d); offsets may be negative (crop).
__device__ __forceinline__ int map_index(int i, int size, int mode) {
  if (i >= 0 && i < size) return i;
  if (mode == 1) { if (i < 0) i = -i; if (i >= size) i = 2 * (size - 1) - i; return (i >= 0 && i < size) ? i : -1; }
  if (mode == 2) return i < 0 ? 0 : size - 1;
  return -1;
}

// dst[n,y,x,:] = src[n, map(y - oy), map(x - ox), :]
__global__ void __launch_bounds__(TPB)
k_window_fwd(const float* __restrict__ src, float* __restrict__ dst, int N, int Hs, int Ws, int Hd, int Wd, int C,
             int oy, int ox, int mode) {
  const int64_t total = (int64_t)N * Hd * Wd * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    int64_t p = i / C;
    const int x = (int)(p % Wd); p /= Wd;
    const int y = (int)(p % Hd);
    const int n = (int)(p / Hd);
    const int sy = map_index(y - oy, Hs, mode), sx = map_index(x - ox, Ws, mode);
    dst[i] = (sy >= 0 && sx >= 0) ? src[(((size_t)n * Hs + sy) * Ws + sx) * C + c] : 0.f;
  }
}

// adjoint: gsrc[n,sy,sx,:] = sum over every dst position that maps to (sy, sx); R = search radius (max |pad|)
__global__ void __launch_bounds__(TPB)
k_window_bwd(const float* __restrict__ gdst, float* __restrict__ gsrc, int N, int Hs, int Ws, int Hd, int Wd, int C,
             int oy, int ox, int mode, int R) {
  const int64_t total = (int64_t)N * Hs * Ws * C;
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    int64_t p = i / C;
    const int sx = (int)(p % Ws); p /= Ws;
    const int sy = (int)(p % Hs);
    const int n = (int)(p / Hs);
    float acc = 0.f;
    for (int y = max(0, sy + oy - 2 * R); y < min(Hd, sy + oy + 2 * R + 1); ++y) {
      if (map_index(y - oy, Hs, mode) != sy) continue;
      for (int x = max(0, sx + ox - 2 * R); x < min(Wd, sx + ox + 2 * R + 1); ++x) {
        if (map_index(x - ox, Ws, mode) != sx) continue;
        acc += gdst[(((size_t)n * Hd + y) * Wd + x) * C + c];
      }
    }
    gsrc[i] = acc;
  }
}

// Downsample(filt_size=3, stride=2, reflect pad 1): y[yo,xo] = sum_{i,j} f_i f_j x[refl(2yo-1+i), refl(2xo-1+j)], f = [1,2,1]/4
__global__ void __launch_bounds__(TPB)
k_blurdown_fwd(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)N * Ho * Wo * C;
  const float f[3] = {0.25f, 0.5f, 0.25f};
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    int64_t p = i / C;
    const int xo = (int)(p % Wo); p /= Wo;
    const int yo = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int sy = map_index(2 * yo - 1 + a, H, 1);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int sx = map_index(2 * xo - 1 + b, W, 1);
        acc += f[a] * f[b] * x[(((size_t)n * H + sy) * W + sx) * C + c];
      }
    }
    y[i] = acc;
  }
}
__global__ void __launch_bounds__(TPB)
k_blurdown_bwd(const float* __restrict__ gy, float* __restrict__ gx, int N, int H, int W, int C) {
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int64_t total = (int64_t)N * H * W * C;
  const float f[3] = {0.25f, 0.5f, 0.25f};
  GRID_STRIDE(i, total) {
    const int c = (int)(i % C);
    int64_t p = i / C;
    const int sx = (int)(p % W); p /= W;
    const int sy = (int)(p % H);
    const int n = (int)(p / H);
    float acc = 0.f;
    for (int yo = max(0, sy / 2 - 2); yo < min(Ho, sy / 2 + 3); ++yo)
      for (int a = 0; a < 3; ++a) {
        if (map_index(2 * yo - 1 + a, H, 1) != sy) continue;
        for (int xo = max(0, sx / 2 - 2); xo < min(Wo, sx / 2 + 3); ++xo)
          for (int b = 0; b < 3; ++b) {
            if (map_index(2 * xo - 1 + b, W, 1) != sx) continue;
            acc += f[a] * f[b] * gy[(((size_t)n * Ho + yo) * Wo + xo) * C + c];
          }
      }
    gx[i] = acc;
  }
}


// ---- joint geometric augmentation on the device (replaces the reference's PIL workers: JointRotate, JointElasticDeform,
// JointRandomResizedCrop of data_loader/externalTransforms.py:45-90 composed into ONE resampling pass).
// For output pixel (yo, xo) of sample n the source position is
//     (ys, xs) = A_n * (xo, yo, 1)  +  D_n(yo, xo)
// A_n: 2x3 affine [a00 a01 a02; a10 a11 a12] mapping output pixel centres to source pixel coordinates (crop window +
// rotation about the image centre, built on the host per sample); D_n: the elastic displacement, bilinear interpolation
// of a PxP control grid of (dy, dx) offsets spanning the output image (P = 0: none).  Image: bilinear, zeros outside;
// label map: nearest (round-half-away like lrintf's default would differ per platform, so floor(v + 0.5)), zeros outside.
__global__ void __launch_bounds__(TPB)
k_warp_joint(const float* __restrict__ img, const int64_t* __restrict__ msk, const float* __restrict__ aff,
             const float* __restrict__ ctrl, float* __restrict__ oimg, int64_t* __restrict__ omsk, int N, int H, int W,
             int Ho, int Wo, int P) {
  const int64_t total = (int64_t)N * Ho * Wo;
  GRID_STRIDE(i, total) {
    const int xo = (int)(i % Wo);
    const int yo = (int)((i / Wo) % Ho);
    const int n = (int)(i / ((int64_t)Wo * Ho));
    const float* a = aff + n * 6;
    float xs = a[0] * xo + a[1] * yo + a[2];
    float ys = a[3] * xo + a[4] * yo + a[5];
    if (P > 0) {
      const float gy = (Ho > 1) ? yo * (float)(P - 1) / (float)(Ho - 1) : 0.f;
      const float gx = (Wo > 1) ? xo * (float)(P - 1) / (float)(Wo - 1) : 0.f;
      int y0 = (int)floorf(gy), x0 = (int)floorf(gx);
      if (y0 > P - 2) y0 = P - 2;
      if (x0 > P - 2) x0 = P - 2;
      if (y0 < 0) y0 = 0;
      if (x0 < 0) x0 = 0;
      const float fy = gy - y0, fx = gx - x0;
      const float* c = ctrl + (size_t)n * 2 * P * P;
      const int y1 = min(y0 + 1, P - 1), x1 = min(x0 + 1, P - 1);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float* ck = c + k * P * P;
        const float d = (1.f - fy) * ((1.f - fx) * ck[y0 * P + x0] + fx * ck[y0 * P + x1]) +
                        fy * ((1.f - fx) * ck[y1 * P + x0] + fx * ck[y1 * P + x1]);
        if (k == 0) ys += d; else xs += d;
      }
    }
    // image: bilinear with zero fill
    const float fy0 = floorf(ys), fx0 = floorf(xs);
    const int iy = (int)fy0, ix = (int)fx0;
    const float wy = ys - fy0, wx = xs - fx0;
    const float* src = img + (size_t)n * H * W;
    auto at = [&](int y, int x) { return (y >= 0 && y < H && x >= 0 && x < W) ? src[(size_t)y * W + x] : 0.f; };
    oimg[i] = (1.f - wy) * ((1.f - wx) * at(iy, ix) + wx * at(iy, ix + 1)) + wy * ((1.f - wx) * at(iy + 1, ix) + wx * at(iy + 1, ix + 1));
    if (msk) {
      const int ny = (int)floorf(ys + 0.5f), nx = (int)floorf(xs + 0.5f);
      omsk[i] = (ny >= 0 && ny < H && nx >= 0 && nx < W) ? msk[(size_t)n * H * W + (size_t)ny * W + nx] : 0;
    }
  }
}

// JointElasticDeform (reference data_loader/externalTransforms.py:69-90: elasticdeform.deform_random_grid([img, msk], sigma,
// points = 3, order = [0, 0])), the resampling step: a P x P grid of control displacements (dy, dx), first / last control point
// on the first / last pixel, interpolated to every pixel by CUBIC B-SPLINES -- `coef` holds the spline coefficients of the
// control values (the caller prefilters them, mirror boundary: data_loader/gpu_augment.py::spline_prefilter) -- then image AND
// label map are sampled at (y + dy, x + dx) with ORDER 0 (nearest, floor(c + 0.5)); a source coordinate below 0 or above size - 1
// reads the constant 0 (mode 'constant', cval 0, the range test of the scipy interpolation code elasticdeform carries).
__device__ __forceinline__ void bspline3(float t, float* w) {       // weights of the four taps floor(t) - 1 .. floor(t) + 2
  const float f = t - floorf(t), g = 1.f - f;
  w[0] = g * g * g * (1.f / 6.f);
  w[1] = (4.f - 6.f * f * f + 3.f * f * f * f) * (1.f / 6.f);
  w[2] = (4.f - 6.f * g * g + 3.f * g * g * g) * (1.f / 6.f);
  w[3] = f * f * f * (1.f / 6.f);
}
__device__ __forceinline__ int mirror_idx(int i, int n) {           // whole-sample mirror: -1 -> 1, n -> n - 2 (n >= 2)
  const int period = 2 * (n - 1);
  i = i % period;
  if (i < 0) i += period;
  return i < n ? i : period - i;
}
__global__ void __launch_bounds__(TPB)
k_elastic_nearest(const float* __restrict__ img, const int64_t* __restrict__ msk, const float* __restrict__ coef,
                  float* __restrict__ oimg, int64_t* __restrict__ omsk, int N, int H, int W, int P) {
  const int64_t total = (int64_t)N * H * W;
  GRID_STRIDE(i, total) {
    const int x = (int)(i % W);
    const int y = (int)((i / W) % H);
    const int n = (int)(i / ((int64_t)W * H));
    const float gy = (H > 1) ? y * (float)(P - 1) / (float)(H - 1) : 0.f;
    const float gx = (W > 1) ? x * (float)(P - 1) / (float)(W - 1) : 0.f;
    float wy[4], wx[4];
    bspline3(gy, wy);
    bspline3(gx, wx);
    const int y0 = (int)floorf(gy) - 1, x0 = (int)floorf(gx) - 1;
    const float* c = coef + (size_t)n * 2 * P * P;
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int iy = mirror_idx(y0 + a, P);
      float r0 = 0.f, r1 = 0.f;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ix = mirror_idx(x0 + b, P);
        r0 += wx[b] * c[iy * P + ix];
        r1 += wx[b] * c[P * P + iy * P + ix];
      }
      d0 += wy[a] * r0;
      d1 += wy[a] * r1;
    }
    const float ys = (float)y + d0, xs = (float)x + d1;
    const bool in = ys >= 0.f && ys <= (float)(H - 1) && xs >= 0.f && xs <= (float)(W - 1);
    const int ny = (int)floorf(ys + 0.5f), nx = (int)floorf(xs + 0.5f);
    const size_t src = (size_t)n * H * W + (size_t)(in ? ny : 0) * W + (in ? nx : 0);
    oimg[i] = in ? img[src] : 0.f;
    if (msk) omsk[i] = in ? msk[src] : 0;
  }
}

}  // namespace

extern "C" {

#define ST ((hipStream_t)stream)

int smsut_add_act(const float* a, const float* b, float* y, int64_t n, float slope, void* stream) {
  SMSUT_REQUIRE(a && y && n > 0);
  k_add_act<<<ew_grid(n / 4 + 1), TPB, 0, ST>>>(a, b, y, n, slope);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_act_bwd(const float* gy, const float* y, float* gx, int64_t n, float slope, void* stream) {
  SMSUT_REQUIRE(gy && y && gx && n > 0);
  k_act_bwd<<<ew_grid(n / 4 + 1), TPB, 0, ST>>>(gy, y, gx, n, slope);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_tanh_fwd(const float* x, float* y, int64_t n, void* stream) {
  SMSUT_REQUIRE(x && y && n > 0);
  k_tanh_fwd<<<ew_grid(n), TPB, 0, ST>>>(x, y, n);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_tanh_bwd(const float* gy, const float* y, float* gx, int64_t n, void* stream) {
  SMSUT_REQUIRE(gy && y && gx && n > 0);
  k_tanh_bwd<<<ew_grid(n), TPB, 0, ST>>>(gy, y, gx, n);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_bias_add(const float* x, const float* bias, float* y, int64_t rows, int C, void* stream) {
  SMSUT_REQUIRE(x && bias && y && rows > 0 && C > 0);
  k_bias_add<<<ew_grid(rows * C), TPB, 0, ST>>>(x, bias, y, rows * C, C);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_row_lerp(const float* a, const float* b, const float* alpha, float* out, int64_t rows, int64_t row_len,
                   void* stream) {
  SMSUT_REQUIRE(a && b && alpha && out && rows > 0 && row_len > 0);
  k_row_lerp<<<ew_grid(rows * row_len), TPB, 0, ST>>>(a, b, alpha, out, rows * row_len, row_len);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_fill(float* out, float v, int64_t n, void* stream) {
  SMSUT_REQUIRE(out && n > 0);
  k_fill<<<ew_grid(n), TPB, 0, ST>>>(out, v, n);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// out = x * (*scale_dev or 1) * mul
int smsut_scale(const float* x, const float* scale_dev, float mul, float* out, int64_t n, void* stream) {
  SMSUT_REQUIRE(x && out && n > 0);
  k_scale<<<ew_grid(n), TPB, 0, ST>>>(x, scale_dev, mul, out, n);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// ---- SGD with momentum + weight decay over MANY tensors in one launch (r05).  The update rule of torch.optim.SGD(momentum, weight_decay,
// dampening 0, no Nesterov) -- the reference's optimizer (/root/reference/trainer/baseTrainer.py: SGD(lr, momentum=0.9, weight_decay)) --
// g' = g + wd p; buf = momentum buf + g'; p -= lr buf, element-wise, so a parameter, its gradient and its momentum buffer only need the SAME
// dense layout (they share the HWIO strides).  torch's own multi-tensor kernel launches ~48 blocks four times for the 12.6 MB of the
// generator (28 us each, latency-bound); here one block per 8192-element chunk of one tensor: ~1500 blocks, one launch.
struct SgdEnt { float* p; const float* g; float* buf; long long n; };
constexpr int SGD_CHUNK = 8192;
__global__ void __launch_bounds__(TPB)
k_sgd_multi(const SgdEnt* __restrict__ ents, const int* __restrict__ blk_ent, const int* __restrict__ blk_chunk, float lr, float momentum,
            float wd) {
  const SgdEnt e = ents[blk_ent[blockIdx.x]];
  const long long base = (long long)blk_chunk[blockIdx.x] * SGD_CHUNK;
  const long long end = base + SGD_CHUNK < e.n ? base + SGD_CHUNK : e.n;
  for (long long i = base + threadIdx.x; i < end; i += TPB) {
    const float pv = e.p[i];
    const float g = e.g[i] + pv * wd;
    const float b = momentum * e.buf[i] + g;
    e.buf[i] = b;
    e.p[i] = pv - lr * b;
  }
}
// ents: device array of `SgdEnt` {p, g, buf, n} (4 x 8 bytes each); blk_ent / blk_chunk: for every block its entry and its chunk index
// inside that entry's tensor (host-built: sum over entries of ceil(n / 8192) blocks).  All momentum buffers must exist (not the first step).
int smsut_sgd_momentum_multi(const void* ents, const int* blk_ent, const int* blk_chunk, int nblocks, float lr, float momentum, float wd,
                             void* stream) {
  SMSUT_REQUIRE(ents && blk_ent && blk_chunk && nblocks > 0);
  k_sgd_multi<<<nblocks, TPB, 0, ST>>>((const SgdEnt*)ents, blk_ent, blk_chunk, lr, momentum, wd);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_sgd_chunk(void) { return SGD_CHUNK; }
int smsut_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  SMSUT_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && !(H & 1) && !(W & 1));
  if (C % 4 == 0) k_maxpool_fwd<4><<<ew_grid(((int64_t)N * H * W * C / 4) / 4), TPB, 0, ST>>>(x, y, N, H, W, C);
  else k_maxpool_fwd<1><<<ew_grid((int64_t)N * H * W * C / 4), TPB, 0, ST>>>(x, y, N, H, W, C);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_maxpool2_bwd(const float* gy, const float* x, float* gx, int N, int H, int W, int C, void* stream) {
  SMSUT_REQUIRE(gy && x && gx && N > 0 && H > 0 && W > 0 && C > 0 && !(H & 1) && !(W & 1));
  if (C % 4 == 0) k_maxpool_bwd<4><<<ew_grid(((int64_t)N * H * W * C / 4) / 4), TPB, 0, ST>>>(gy, x, gx, N, H, W, C);
  else k_maxpool_bwd<1><<<ew_grid((int64_t)N * H * W * C / 4), TPB, 0, ST>>>(gy, x, gx, N, H, W, C);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// gx = maxpool2_bwd(gy; x) + add: the pooled path's and the skip connection's gradients of an encoder level in one pass
int smsut_maxpool2_bwd_add(const float* gy, const float* x, const float* add, float* gx, int N, int H, int W, int C,
                           void* stream) {
  SMSUT_REQUIRE(gy && x && add && gx && N > 0 && H > 0 && W > 0 && C > 0 && !(H & 1) && !(W & 1));
  if (C % 4 == 0) k_maxpool_bwd<4><<<ew_grid(((int64_t)N * H * W * C / 4) / 4), TPB, 0, ST>>>(gy, x, gx, N, H, W, C, add);
  else k_maxpool_bwd<1><<<ew_grid((int64_t)N * H * W * C / 4), TPB, 0, ST>>>(gy, x, gx, N, H, W, C, add);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_avgpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  SMSUT_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && !(H & 1) && !(W & 1));
  if (C % 4 == 0) k_avgpool_fwd<4><<<ew_grid(((int64_t)N * H * W * C / 4) / 4), TPB, 0, ST>>>(x, y, N, H, W, C);
  else k_avgpool_fwd<1><<<ew_grid((int64_t)N * H * W * C / 4), TPB, 0, ST>>>(x, y, N, H, W, C);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// (H, W) are the UN-pooled sizes: gy is [N,H/2,W/2,C], gx is [N,H,W,C]
int smsut_avgpool2_bwd(const float* gy, float* gx, int N, int H, int W, int C, void* stream) {
  SMSUT_REQUIRE(gy && gx && N > 0 && H > 0 && W > 0 && C > 0 && !(H & 1) && !(W & 1));
  if (C % 4 == 0) k_avgpool_bwd<4><<<ew_grid(((int64_t)N * H * W * C) / 4), TPB, 0, ST>>>(gy, gx, N, H, W, C);
  else k_avgpool_bwd<1><<<ew_grid((int64_t)N * H * W * C), TPB, 0, ST>>>(gy, gx, N, H, W, C);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// Joint geometric augmentation (see k_warp_joint): img [N,H,W] fp32 (one channel), msk [N,H,W] int64 (nullable),
// aff [N][6], ctrl [N][2][P][P] (nullable when P == 0) -> oimg [N,Ho,Wo], omsk [N,Ho,Wo].
int smsut_warp_joint(const float* img, const int64_t* msk, const float* aff, const float* ctrl, float* oimg,
                     int64_t* omsk, int N, int H, int W, int Ho, int Wo, int P, void* stream) {
  SMSUT_REQUIRE(img && aff && oimg && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && (P == 0 || (P >= 2 && ctrl)) &&
                (!msk || omsk));
  k_warp_joint<<<ew_grid((int64_t)N * Ho * Wo), TPB, 0, ST>>>(img, msk, aff, ctrl, oimg, omsk, N, H, W, Ho, Wo, P);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// JointElasticDeform's resampling (see k_elastic_nearest): img [N,H,W] fp32, msk [N,H,W] int64 (nullable), coef [N][2][P][P] cubic
// B-spline coefficients of the control displacements (dy, dx) -> oimg, omsk [N,H,W] (must not alias the inputs).
int smsut_elastic_deform(const float* img, const int64_t* msk, const float* coef, float* oimg, int64_t* omsk, int N, int H,
                         int W, int P, void* stream) {
  SMSUT_REQUIRE(img && coef && oimg && N > 0 && H > 0 && W > 0 && P >= 2 && (!msk || omsk) && oimg != img);
  k_elastic_nearest<<<ew_grid((int64_t)N * H * W), TPB, 0, ST>>>(img, msk, coef, oimg, omsk, N, H, W, P);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_bilinear2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  SMSUT_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0);
  if (C % 4 == 0) k_bilinear2_fwd<4><<<ew_grid(((int64_t)N * H * W * C * 4) / 4), TPB, 0, ST>>>(x, y, N, H, W, C);
  else k_bilinear2_fwd<1><<<ew_grid((int64_t)N * H * W * C * 4), TPB, 0, ST>>>(x, y, N, H, W, C);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_bilinear2_bwd(const float* gy, float* gx, int N, int H, int W, int C, void* stream) {
  SMSUT_REQUIRE(gy && gx && N > 0 && H > 0 && W > 0 && C > 0);
  if (C % 4 == 0) k_bilinear2_bwd<4><<<ew_grid(((int64_t)N * H * W * C) / 4), TPB, 0, ST>>>(gy, gx, N, H, W, C);
  else k_bilinear2_bwd<1><<<ew_grid((int64_t)N * H * W * C), TPB, 0, ST>>>(gy, gx, N, H, W, C);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// nn.ReflectionPad2d / nn.ReplicationPad2d / zero pad / crop (networks.py:95-105,618,835-851): dst is
// [N][Hs+top+bottom][Ws+left+right][C]; (oy, ox) = (top, left), negative values crop.  mode 0 zero, 1 reflect, 2 replicate.
int smsut_window_fwd(const float* src, float* dst, int N, int Hs, int Ws, int Hd, int Wd, int C, int oy, int ox, int mode,
                     void* stream) {
  SMSUT_REQUIRE(src && dst && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0 && mode >= 0 && mode <= 2);
  k_window_fwd<<<ew_grid((int64_t)N * Hd * Wd * C), TPB, 0, ST>>>(src, dst, N, Hs, Ws, Hd, Wd, C, oy, ox, mode);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_window_bwd(const float* gdst, float* gsrc, int N, int Hs, int Ws, int Hd, int Wd, int C, int oy, int ox, int mode,
                     void* stream) {
  SMSUT_REQUIRE(gdst && gsrc && N > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && C > 0 && mode >= 0 && mode <= 2);
  int R = abs(oy) > abs(ox) ? abs(oy) : abs(ox);
  const int ry = abs(Hd - Hs - oy), rx = abs(Wd - Ws - ox);
  if (ry > R) R = ry;
  if (rx > R) R = rx;
  k_window_bwd<<<ew_grid((int64_t)N * Hs * Ws * C), TPB, 0, ST>>>(gdst, gsrc, N, Hs, Ws, Hd, Wd, C, oy, ox, mode, R);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// networks.Downsample(channels, 'reflect', filt_size=3, stride=2) (networks.py:37-60): blur [1,2,1]x[1,2,1]/16, stride 2
int smsut_blurdown_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  SMSUT_REQUIRE(x && y && N > 0 && H > 1 && W > 1 && C > 0);
  k_blurdown_fwd<<<ew_grid((int64_t)N * H * W * C / 4 + 1), TPB, 0, ST>>>(x, y, N, H, W, C);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_blurdown_bwd(const float* gy, float* gx, int N, int H, int W, int C, void* stream) {
  SMSUT_REQUIRE(gy && gx && N > 0 && H > 1 && W > 1 && C > 0);
  k_blurdown_bwd<<<ew_grid((int64_t)N * H * W * C), TPB, 0, ST>>>(gy, gx, N, H, W, C);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_copy_channels(const float* src, int Cs, int src_off, float* dst, int Cd, int dst_off, int Cc, int64_t P,
                        void* stream) {
  SMSUT_REQUIRE(src && dst && P > 0 && Cc > 0 && src_off >= 0 && dst_off >= 0 && src_off + Cc <= Cs &&
                dst_off + Cc <= Cd);
  k_copy_channels<<<ew_grid(P * Cc / 4 + 1), TPB, 0, ST>>>(src, Cs, src_off, dst, Cd, dst_off, Cc, P);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
// One-pass concat (dir 0) / split (dir 1) of two channel blocks, both multiples of 4 channels; a / b nullable in dir 1.
int smsut_concat2(float* a, int Ca, float* b, int Cb, float* y, int64_t P, int dir, void* stream) {
  SMSUT_REQUIRE(y && P > 0 && Ca > 0 && Cb > 0 && (Ca & 3) == 0 && (Cb & 3) == 0 && (dir == 0 ? (a && b) : (a || b)));
  k_concat2<<<ew_grid(P * (Ca + Cb) / 4), TPB, 0, ST>>>(a, Ca, b, Cb, y, P, dir);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}
int smsut_modal_planes(const float* x, const float* m, float* out, int N, int64_t HW, int Cx, int M, void* stream) {
  SMSUT_REQUIRE(x && m && out && N > 0 && HW > 0 && Cx > 0 && M > 0);
  k_modal_planes<<<ew_grid((int64_t)N * HW * (Cx + M)), TPB, 0, ST>>>(x, m, out, N, HW, Cx, M);
  SMSUT_LAUNCH_CHECK(); return SMSUT_OK;
}

}  // extern "C"
