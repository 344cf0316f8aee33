// Internal interface between conv_mfma.hip (entry points, form selection) and conv_wgrad_rr.hip (the register-row weight gradient).
#pragma once
#include "common.h"

struct RrAff { const float* mean; const float* rstd; const float* gamma; const float* beta; float slope; };

// 3x3 stride-1 "same" weight gradient, slabs written per split as [split][9 (+1 with gs)][Cin][Cout] (the layout sum_splits reads).
// Shapes: W % 16 == 0, H % 4 == 0, Cin % 16 == 0, Cout % 16 == 0, 32-bit element offsets.
bool smsut_wgrad_rr_eligible(int N, int H, int W, int Cin, int Cout, const float* x2, int ca, bool aff, bool sc);
// number of split slabs the launch writes (0 = not eligible)
int smsut_wgrad_rr_splits(int N, int H, int W, int Cin, int Cout, const float* x2, int ca, bool aff, bool sc);
// x2 / ca: x is the virtual cat([x, x2]) with ca channels in x; aff: x is lrelu(IN(.)) of the tensor passed (zero padding after);
// gs: fused 1x1-shortcut weight gradient (slab row 9 = sum_p x[p] (x) gs[p]).  Returns 0 when launched, -1 when not covered.
// b (nullable): PAIRED launch -- the last b->n of the N images are a second set of tensors that went through the same conv (same
// form: x2 / gs / statistics present in both sets or in neither); one slab set holds the sum over both.
struct RrSetB { const float* x; const float* x2; const float* gy; const float* gs; const float* mean; const float* rstd; int n; };
int smsut_wgrad_rr_launch(const float* x, const float* x2, int ca, const float* gy, const float* gs, float* part, int N, int H,
                          int W, int Cin, int Cout, const RrAff* aff, hipStream_t st, const RrSetB* b = nullptr);
