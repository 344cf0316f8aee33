// Generic direct convolution on NHWC fp32 (any kernel size / stride / zero padding): forward,
// data-gradient and weight-gradient.  These are the shape-complete kernels: they serve the layers
// whose FLOPs are negligible (stems with Cin 1/5, 1x1 heads with Cout 1/5, D's 4x4 s2 stem and
// k4 `conv_cls`, network/ugan.py:202,213-215) and are the on-device cross-check for the MFMA
// implicit-GEMM kernels in conv_mfma.hip, which take over every stride-1 "same" conv with
// Cin%4==0 && Cout%16==0 (network/blocks.py:10-16).
//
// Layouts:  x [N][H][W][Cin],  w [KH][KW][Cin][Cout] ("HWIO"),  y [N][Ho][Wo][Cout].
// ConvTranspose2d(k2,s2) (blocks.py:41) is the data-gradient form of a k2/s2/p0 conv, so the
// up-path reuses dgrad (forward), fwd (its data-gradient) and wgrad with swapped operands.
#include "common.h"

namespace {

constexpr int TPB = 256;

struct ConvGeom {
  int N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad;
};

__global__ void __launch_bounds__(TPB)
conv_fwd_naive(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
               float* __restrict__ y, ConvGeom g, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
    const int co = (int)(i % g.Cout);
    int64_t p = i / g.Cout;
    const int wo = (int)(p % g.Wo); p /= g.Wo;
    const int ho = (int)(p % g.Ho);
    const int n = (int)(p / g.Ho);
    float acc = bias ? bias[co] : 0.f;
    for (int kh = 0; kh < g.KH; ++kh) {
      const int hi = ho * g.stride - g.pad + kh;
      if (hi < 0 || hi >= g.H) continue;
      for (int kw = 0; kw < g.KW; ++kw) {
        const int wi = wo * g.stride - g.pad + kw;
        if (wi < 0 || wi >= g.W) continue;
        const float* xp = x + (((size_t)n * g.H + hi) * g.W + wi) * g.Cin;
        const float* wp = w + ((size_t)(kh * g.KW + kw) * g.Cin) * g.Cout + co;
        for (int ci = 0; ci < g.Cin; ++ci) acc = fmaf(xp[ci], wp[(size_t)ci * g.Cout], acc);
      }
    }
    y[i] = acc;
  }
}

// Full-window conv (KH == H, KW == W, pad 0 -> 1x1 output): D's conv_cls (ugan.py:213,215).  NHWC makes the input of
// one image the flat vector [KH*KW*Cin] in exactly the weight's [KH][KW][Cin] order: y[n][co] = sum_k x[n][k] w[k][co].
__global__ void __launch_bounds__(TPB)
conv_full_window(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                 float* __restrict__ y, int K, int Cout) {
  __shared__ float sm4[4];
  const int n = blockIdx.x;
  for (int co = 0; co < Cout; ++co) {
    float acc = 0.f;
    for (int k = threadIdx.x; k < K; k += TPB) acc = fmaf(x[(size_t)n * K + k], w[(size_t)k * Cout + co], acc);
    const float t = block_sum_256(acc, sm4);
    if (threadIdx.x == 0) y[(size_t)n * Cout + co] = t + (bias ? bias[co] : 0.f);
  }
}

__global__ void __launch_bounds__(TPB)
conv_dgrad_naive(const float* __restrict__ gy, const float* __restrict__ w, float* __restrict__ gx, ConvGeom g,
                 int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TPB) {
    const int ci = (int)(i % g.Cin);
    int64_t p = i / g.Cin;
    const int wi = (int)(p % g.W); p /= g.W;
    const int hi = (int)(p % g.H);
    const int n = (int)(p / g.H);
    float acc = 0.f;
    for (int kh = 0; kh < g.KH; ++kh) {
      const int hn = hi + g.pad - kh;
      if (hn < 0 || hn % g.stride) continue;
      const int ho = hn / g.stride;
      if (ho >= g.Ho) continue;
      for (int kw = 0; kw < g.KW; ++kw) {
        const int wn = wi + g.pad - kw;
        if (wn < 0 || wn % g.stride) continue;
        const int wo = wn / g.stride;
        if (wo >= g.Wo) continue;
        const float* gp = gy + (((size_t)n * g.Ho + ho) * g.Wo + wo) * g.Cout;
        const float* wp = w + ((size_t)(kh * g.KW + kw) * g.Cin + ci) * g.Cout;
        int co = 0;
        if ((g.Cout & 3) == 0) {
          for (; co < g.Cout; co += 4) {
            const float4 a = *(const float4*)(gp + co);
            const float4 b = *(const float4*)(wp + co);
            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc);
            acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
          }
        } else {
          for (; co < g.Cout; ++co) acc = fmaf(gp[co], wp[co], acc);
        }
      }
    }
    gx[i] = acc;
  }
}

// partial[chunk][KH*KW*Cin*Cout]: each thread owns one weight element and walks the chunk's output pixels.
__global__ void __launch_bounds__(TPB)
conv_wgrad_partial(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ part, ConvGeom g,
                   int wsize, int64_t npix, int pix_per_chunk) {
  const int e = blockIdx.x * TPB + threadIdx.x;
  const int chunk = blockIdx.y;
  if (e >= wsize) return;
  const int co = e % g.Cout;
  int t = e / g.Cout;
  const int ci = t % g.Cin; t /= g.Cin;
  const int kw = t % g.KW;
  const int kh = t / g.KW;
  const int64_t p0 = (int64_t)chunk * pix_per_chunk;
  const int64_t p1 = min(p0 + (int64_t)pix_per_chunk, npix);
  // walk (n, ho, wo) incrementally: no div/mod in the loop
  int wo = (int)(p0 % g.Wo);
  const int64_t q0 = p0 / g.Wo;
  int ho = (int)(q0 % g.Ho);
  int n = (int)(q0 / g.Ho);
  float acc = 0.f;
  const float* gp = gy + (size_t)p0 * g.Cout + co;
  for (int64_t p = p0; p < p1; ++p, gp += g.Cout) {
    const int hi = ho * g.stride - g.pad + kh;
    const int wi = wo * g.stride - g.pad + kw;
    if (hi >= 0 && hi < g.H && wi >= 0 && wi < g.W)
      acc = fmaf(x[(((size_t)n * g.H + hi) * g.W + wi) * g.Cin + ci], *gp, acc);
    if (++wo == g.Wo) { wo = 0; if (++ho == g.Ho) { ho = 0; ++n; } }
  }
  part[(size_t)chunk * wsize + e] = acc;
}

// out[e] = sum_c part[c][e]: 64 elements x 4 chunk-lanes per block, fp64 accumulation, fixed order
__global__ void __launch_bounds__(TPB)
reduce_chunks(const float* __restrict__ part, float* __restrict__ out, int wsize, int chunks) {
  __shared__ double sm[TPB];
  const int e = blockIdx.x * 64 + (threadIdx.x & 63);
  const int lane_c = threadIdx.x >> 6;
  double s = 0.0;
  if (e < wsize) {
    // eight chunk rows in flight, added in the original order (rolled, this loop was chunks / 4 dependent load latencies:
    // a constant ~25 us under every smsut_colsum / generic weight-gradient call)
    int c = lane_c;
    for (; c + 28 < chunks; c += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(c + 4 * u) * wsize + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; c < chunks; c += 4) s += (double)part[(size_t)c * wsize + e];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  if (lane_c == 0 && e < wsize)
    out[e] = (float)(sm[threadIdx.x] + sm[threadIdx.x + 64] + sm[threadIdx.x + 128] + sm[threadIdx.x + 192]);
}

// column sums of a [rows][C] matrix -> partial[chunk][C]
__global__ void __launch_bounds__(TPB)
colsum_partial(const float* __restrict__ x, float* __restrict__ part, int64_t rows, int C, int rows_per_chunk) {
  const int chunk = blockIdx.x;
  const int64_t r0 = (int64_t)chunk * rows_per_chunk;
  const int64_t r1 = min(r0 + (int64_t)rows_per_chunk, rows);
  const int TC = C < TPB ? C : TPB;
  const int nrow = TPB / TC;
  const int tc = threadIdx.x % TC, tr = threadIdx.x / TC;
  __shared__ float sm[TPB];
  for (int c0 = 0; c0 < C; c0 += TC) {
    const int c = c0 + tc;
    float acc = 0.f;
    if (tr < nrow && c < C) {
      // eight rows in flight per thread, added in the original order (the rolled loop kept ONE 4-byte load in flight: 0.8 TB/s)
      int64_t r = r0 + tr;
      for (; r + 7 * (int64_t)nrow < r1; r += 8 * (int64_t)nrow) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[(size_t)(r + u * (int64_t)nrow) * C + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
      }
      for (; r < r1; r += nrow) acc += x[(size_t)r * C + c];
    }
    __syncthreads();
    sm[threadIdx.x] = acc;
    __syncthreads();
    if (tr == 0 && c < C) {
      float tot = 0.f;
      for (int r = 0; r < nrow; ++r) tot += sm[r * TC + tc];
      part[(size_t)chunk * C + c] = tot;
    }
    __syncthreads();
  }
}

inline int wgrad_ppc(int64_t npix, int wsize) {
  // small weight tensors (stems, heads) get many short chunks so the grid still fills the chip
  const int64_t chunks_target = wsize <= 4096 ? 8192 : 1024;
  int64_t ppc = cdiv64(npix, chunks_target);
  if (ppc < 64) ppc = 64;
  return (int)ppc;
}

inline bool geom_ok(const ConvGeom& g) {
  if (g.N <= 0 || g.H <= 0 || g.W <= 0 || g.Cin <= 0 || g.Cout <= 0 || g.KH <= 0 || g.KW <= 0 || g.stride <= 0 ||
      g.pad < 0)
    return false;
  return g.Ho == (g.H + 2 * g.pad - g.KH) / g.stride + 1 && g.Wo == (g.W + 2 * g.pad - g.KW) / g.stride + 1 &&
         g.Ho > 0 && g.Wo > 0;
}

}  // namespace

extern "C" {

int smsut_conv2d_fwd_generic(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin,
                             int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, void* stream) {
  ConvGeom g{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad};
  SMSUT_REQUIRE(x && w && y && geom_ok(g));
  const int64_t total = (int64_t)N * Ho * Wo * Cout;
  if (KH == H && KW == W && pad == 0 && Cout <= 16)
    conv_full_window<<<N, TPB, 0, (hipStream_t)stream>>>(x, w, bias, y, KH * KW * Cin, Cout);
  else
    conv_fwd_naive<<<ew_grid(total) * 4, TPB, 0, (hipStream_t)stream>>>(x, w, bias, y, g, total);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

int smsut_conv2d_dgrad_generic(const float* gy, const float* w, float* gx, int N, int H, int W, int Cin, int Ho, int Wo,
                               int Cout, int KH, int KW, int stride, int pad, void* stream) {
  ConvGeom g{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad};
  SMSUT_REQUIRE(gy && w && gx && geom_ok(g));
  const int64_t total = (int64_t)N * H * W * Cin;
  conv_dgrad_naive<<<ew_grid(total) * 4, TPB, 0, (hipStream_t)stream>>>(gy, w, gx, g, total);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

// workspace floats needed by smsut_conv2d_wgrad_generic
int64_t smsut_conv2d_wgrad_generic_ws(int N, int Ho, int Wo, int Cin, int Cout, int KH, int KW) {
  const int64_t npix = (int64_t)N * Ho * Wo;
  const int ppc = wgrad_ppc(npix, KH * KW * Cin * Cout);
  return cdiv64(npix, ppc) * (int64_t)KH * KW * Cin * Cout;
}

int smsut_conv2d_wgrad_generic(const float* x, const float* gy, float* gw, float* workspace, int N, int H, int W,
                               int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, void* stream) {
  ConvGeom g{N, H, W, Cin, Ho, Wo, Cout, KH, KW, stride, pad};
  SMSUT_REQUIRE(x && gy && gw && workspace && geom_ok(g));
  const int64_t npix = (int64_t)N * Ho * Wo;
  const int wsize = KH * KW * Cin * Cout;
  const int ppc = wgrad_ppc(npix, wsize);
  const int chunks = (int)cdiv64(npix, ppc);
  hipStream_t st = (hipStream_t)stream;
  conv_wgrad_partial<<<dim3((wsize + TPB - 1) / TPB, chunks), TPB, 0, st>>>(x, gy, workspace, g, wsize, npix, ppc);
  reduce_chunks<<<(wsize + 63) / 64, TPB, 0, st>>>(workspace, gw, wsize, chunks);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

// out[c] = sum over rows of x[rows][C].  workspace: float[smsut_colsum_ws(rows, C)]
int64_t smsut_colsum_ws(int64_t rows, int C) {
  int64_t rpc = cdiv64(rows, 512);
  if (rpc < 64) rpc = 64;
  return cdiv64(rows, rpc) * C;
}

int smsut_colsum(const float* x, float* out, float* workspace, int64_t rows, int C, void* stream) {
  SMSUT_REQUIRE(x && out && workspace && rows > 0 && C > 0);
  int64_t rpc = cdiv64(rows, 512);
  if (rpc < 64) rpc = 64;
  const int chunks = (int)cdiv64(rows, rpc);
  hipStream_t st = (hipStream_t)stream;
  colsum_partial<<<chunks, TPB, 0, st>>>(x, workspace, rows, C, (int)rpc);
  reduce_chunks<<<(C + 63) / 64, TPB, 0, st>>>(workspace, out, C, chunks);
  SMSUT_LAUNCH_CHECK();
  return SMSUT_OK;
}

}  // extern "C"
