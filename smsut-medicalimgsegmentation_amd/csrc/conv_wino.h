// Internal interface between conv_mfma.hip (entry points, form selection) and conv_wino.hip (the large-reduction Winograd kernel).
#pragma once
#include "common.h"

struct WinoAff { const float* mean; const float* rstd; const float* gamma; const float* beta; float slope; };
struct WinoBst { const float* y1; const float* mean; const float* rstd; const float* gamma; const float* beta; float slope; };
struct WinoSc { const float* w; float* y; float* stats; };

// Shapes the kernel takes: 3x3 stride-1 "same", Kdim % 16 == 0 (>= 32), Ndim % 16 == 0, H % 16 == 0, W % 16 == 0.
bool smsut_wino_l_eligible(int N, int H, int W, int Kdim, int Ndim);

// One launch of conv_wino_l.  `transposed` bit 0: weights read transposed + tap-flipped (data-gradient), bit 1: the result is
// ADDED to what y holds.  x2: the input is the virtual cat([x, x2]) of two Kdim/2-channel tensors; with sc->w and bit 0 set it is
// the fused shortcut data-gradient instead (second half = the shortcut's gradient, 1x1 weights sc->w).  sc (forward): fused 1x1
// shortcut conv (sc->w, result sc->y, InstanceNorm partials sc->stats).  y2 / split: split output.  tiles_out: only report the
// statistics tiles per image.  wu: the caller's prepared image of w for THIS form (smsut_wino_prepare with the same Kdim, Ndim and
// transposed bit 0) or null = transform the weights on the fly; the library keeps no table of images (r03 did: SURVEY 8b rules
// process-wide mutable state out).  fin (statistics / BST forms): the InstanceNorm statistics are finalised inside the launch by the
// last-arriving workgroup (common.h).  Returns 0 when launched (or reported), -1 when the form is not covered (nothing launched).
int smsut_wino_l_launch(const float* x, const float* x2, const float* w, float* y, float* y2, int split, int N, int H, int W,
                        int Kdim, int Ndim, int transposed, float* stats, int* tiles_out, const WinoBst* bst, const WinoAff* aff,
                        const WinoSc* sc, hipStream_t st, const float* wu = nullptr, const FinRef* fin = nullptr);

// ---- Winograd weight gradient F(3x3, 2x2): gw[3][3][Cin][Cout] = sum_p x[p + tap] (x) gy[p] through
//      dg = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G  (16 products per 2x2 tile of gy instead of 36).
// x2 / ca: x is the virtual cat([x, x2]) with ca channels in x; aff: x is lrelu(IN(.)) of the tensor passed (zero padding after).
bool smsut_wino_wg_eligible(int N, int H, int W, int Cin, int Cout, const float* x2, int ca);
int64_t smsut_wino_wg_ws(int N, int H, int W, int Cin, int Cout);          // workspace floats
int smsut_wino_wg_launch(const float* x, const float* x2, int ca, const float* gy, float* gw, float* workspace, int N, int H, int W,
                         int Cin, int Cout, const WinoAff* aff, hipStream_t st);
