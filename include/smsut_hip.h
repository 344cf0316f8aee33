/* smsut_hip.h -- C ABI of the MI355X (gfx950) kernels behind the SMSUT conv hot path.
 *
 * The reference (Sue1347/SMSUT-MedicalImgSegmentation) has no native code and no FFI: its boundary is the Python
 * module API (network.unet / network.ugan / network.networks constructors, trainer.uganConsisTrainer), whose
 * arithmetic it delegates to torch operators.  Each entry point below replaces the torch operator(s) named in
 * its comment at the cited reference call site(s); `smsut-medicalimgsegmentation_amd/ops.py` binds them behind
 * torch.autograd.Function objects and INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - every pointer is DEVICE memory (fp32 unless typed otherwise), allocated and owned by the caller; the
 *     library never allocates, frees or synchronises; `workspace` sizes come from the matching `*_ws` query;
 *   - activations are dense NHWC: [N][H][W][C]; conv weights are [KH][KW][Cin][Cout]; ConvTranspose2x2 weights
 *     are [kh][kw][Cin][Cout]; Linear weights are [in][out];
 *   - `stream` is a hipStream_t passed as void*; every kernel is enqueued on it (graph-capture safe);
 *   - return value: 0 = enqueued, -1 = invalid argument, > 0 = hipError_t of the launch; nothing throws;
 *   - no global mutable state; safe to call from several host threads on different streams.
 */
#ifndef SMSUT_HIP_H
#define SMSUT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------- convolution
 * nn.Conv2d as used at network/blocks.py:10-16 (conv3x3 / conv1x1), blocks.py:123 + ugan.py:26 (5x5 stems),
 * ugan.py:70 (1x1 heads with bias), ugan.py:202 (D stem k4 s2 p1 + bias), ugan.py:214-215 (conv_src / conv_cls).
 * *_generic: any kernel size / stride / zero padding (direct convolution).
 * *_mfma   : stride-1 "same" 1x1 / 3x3 on the matrix cores (v_mfma_f32_16x16x4_f32), the > 97 %-of-FLOPs path. */
int smsut_conv2d_fwd_generic(const float* x, const float* w, const float* bias /*nullable*/, float* y, int N, int H,
                             int W, int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad,
                             void* stream);
int smsut_conv2d_dgrad_generic(const float* gy, const float* w, float* gx, int N, int H, int W, int Cin, int Ho, int Wo,
                               int Cout, int KH, int KW, int stride, int pad, void* stream);
int64_t smsut_conv2d_wgrad_generic_ws(int N, int Ho, int Wo, int Cin, int Cout, int KH, int KW);
int smsut_conv2d_wgrad_generic(const float* x, const float* gy, float* gw, float* workspace, int N, int H, int W,
                               int Cin, int Ho, int Wo, int Cout, int KH, int KW, int stride, int pad, void* stream);

int smsut_conv2d_mfma_supported(int KS, int stride, int pad, int Kdim, int Ndim);
/* transposed = 0: y[N,H,W,Ndim] = conv(x[N,H,W,Kdim], w[KS*KS][Kdim][Ndim]);
 * transposed = 1: data-gradient, x = gy[N,H,W,Kdim=Cout], w = forward weights [KS*KS][Ndim=Cin][Kdim=Cout];
 * transposed | 2: the result is ADDED to what y already holds (second gradient path into a block input,
 *                 reference autograd's accumulation for x used by conv1 and the shortcut, network/blocks.py:66-78). */
int smsut_conv2d_fwd_mfma(const float* x, const float* w, float* y, int N, int H, int W, int Kdim, int Ndim, int KS,
                          int transposed, void* stream);
/* forward conv that also emits the InstanceNorm {sum, sum^2} partials of its output (fused statistics pass):
 * stats = float[N * smsut_conv2d_mfma_tiles(N, H, W, Kdim, Ndim, KS, f16) * Ndim * 2], consumed by smsut_instnorm_fwd_partials
 * (the tile shape is chosen per layer shape AND operand dtype -- f16 = 1 for the *_f16 entry points: the fp32 forms of a shape may
 * run a Winograd kernel on 16-row items where the fp16-operand ones keep the direct kernel's -- so both are part of the query). */
int smsut_conv2d_mfma_tiles(int N, int H, int W, int Kdim, int Ndim, int KS, int f16);
/* 1 when this shape runs one of the persistent kernels that carry the fused forms (statistics / input-side IN / BST epilogues):
 * conv_mfma_fwd_p (direct or Winograd F(2x2,3x3), resident weights, Kdim <= 64) or, fp32 only, conv_wino_l (Winograd, streamed
 * weights, Kdim >= 64); else 0 (per-tile kernel). */
int smsut_conv2d_mfma_persistent(int N, int H, int W, int Kdim, int Ndim, int KS, int f16);
/* data-gradient of conv2 inside a BasicBlock (its input was LeakyReLU(IN(y1)), blocks.py:66-72) with the InstanceNorm backward
 * folded into the epilogue: gz = g * mask(y1) is written instead of g, plus the per-tile partials {sum gz, sum gz*xhat}
 * [N][smsut_conv2d_mfma_tiles(N,H,W,Kdim,Ndim,3)][Ndim][2].  Persistent-kernel shapes only (-1 otherwise). */
int smsut_conv2d_dgrad_mfma_bwdstats(const float* gy, const float* w, float* gz, float* stats, const float* y1,
                                     const float* mean, const float* rstd, const float* gamma, const float* beta,
                                     float slope, int N, int H, int W, int Kdim, int Ndim, void* stream);
int smsut_conv2d_fwd_mfma_stats(const float* x, const float* w, float* y, float* stats, int N, int H, int W, int Kdim,
                                int Ndim, int KS, void* stream);
/* tuning hook: same as smsut_conv2d_fwd_mfma with a forced tile configuration (returns -1 for an unknown cfg) */
int smsut_conv2d_fwd_mfma_cfg(const float* x, const float* w, float* y, int N, int H, int W, int Kdim, int Ndim, int KS,
                              int transposed, int cfg, void* stream);
/* Prepared Winograd weights (optional; replaces nothing in the reference -- it is the cuDNN-internal filter transform of the
 * nn.Conv2d sites above made explicit, network/blocks.py:10-16): the large-reduction Winograd kernel (Kdim >= 64) transforms its
 * weights U = G g G^T per workgroup and chunk unless the caller hands it a prepared image.
 *   smsut_wino_image_floats(Kdim, Ndim)  floats of one image (16 * Kdim * Ndim; 0 when the shape has none);
 *   smsut_wino_prepare                  writes the images of `count` weight tensors in ONE launch.  w[i]: a 3x3 weight tensor in
 *                                       this library's layout [3][3][Cin][Cout]; transposed[i] = 0: image of the forward conv
 *                                       (Kdim[i] = Cin, Ndim[i] = Cout), 1: of its data-gradient (Kdim[i] = Cout, Ndim[i] = Cin);
 *                                       u[i]: device float[smsut_wino_image_floats(Kdim[i], Ndim[i])].  The five arrays are HOST arrays;
 *   *_pre entry points (below)         the conv entry points that may run that kernel, with one more argument: wu = the image the
 *                                       CALLER prepared for this weight tensor and form (same Kdim, Ndim, same bit 0 of `transposed`),
 *                                       or NULL = transform on the fly.  The caller keeps wu in step with w (re-prepare after every
 *                                       change of w).  The library holds NO table of images and no other process-wide mutable state
 *                                       (r03's smsut_wino_bind* registry is gone): an image is an argument like any other.
 * Results are bit-identical with and without an image (same arithmetic, done once instead of per workgroup). */
int64_t smsut_wino_image_floats(int Kdim, int Ndim);
int smsut_wino_prepare(const float* const* w, float* const* u, const int* Kdim, const int* Ndim, const int* transposed, int count,
                       void* stream);
int smsut_conv2d_fwd_mfma_pre(const float* x, const float* w, float* y, int N, int H, int W, int Kdim, int Ndim, int KS,
                              int transposed, const float* wu, void* stream);
int smsut_conv2d_fwd_mfma_stats_pre(const float* x, const float* w, float* y, float* stats, int N, int H, int W, int Kdim,
                                    int Ndim, int KS, const float* wu, void* stream);
int smsut_conv2d_dgrad_mfma_bwdstats_pre(const float* gy, const float* w, float* gz, float* stats, const float* y1,
                                         const float* mean, const float* rstd, const float* gamma, const float* beta,
                                         float slope, int N, int H, int W, int Kdim, int Ndim, const float* wu, void* stream);
int smsut_conv2d_fwd_mfma_stats_inaff_pre(const float* x, const float* w, float* y, float* stats, const float* mean,
                                          const float* rstd, const float* gamma, const float* beta, float slope, int N, int H,
                                          int W, int Kdim, int Ndim, const float* wu, void* stream);
int smsut_conv2d_fwd_mfma_stats_cat_pre(const float* xa, const float* xb, const float* w, float* y, float* stats, int N, int H,
                                        int W, int Kdim, int Ndim, const float* wu, void* stream);
int smsut_conv2d_fwd_mfma_split_pre(const float* x, const float* w, float* ya, float* yb, int split, int N, int H, int W,
                                    int Kdim, int Ndim, int transposed, const float* wu, void* stream);
int smsut_conv2d_fwd_mfma_stats_sc_pre(const float* x, const float* xb, const float* w, const float* wsc, float* y, float* ysc,
                                       float* stats, float* stats_sc, int N, int H, int W, int Kdim, int Ndim, const float* wu,
                                       void* stream);
int smsut_conv2d_wgrad_mfma_supported(int KS, int stride, int pad, int Cin, int Cout);
int64_t smsut_conv2d_wgrad_mfma_ws(int N, int H, int W, int Cin, int Cout, int KS);
int smsut_conv2d_wgrad_mfma(const float* x, const float* gy, float* gw, float* workspace, int N, int H, int W, int Cin,
                            int Cout, int KS, void* stream);

/* 1x1 convolutions as streaming GEMMs (shortcuts blocks.py:63,95-97; up-path blocks.py:45; nn.Linear ugan.py:295): the
 * activation operand goes global -> registers (no LDS, no barrier in the loop), only the weights are staged.
 * transposed = 1 is the data-gradient.  stats (nullable): InstanceNorm partials [N][smsut_conv1x1_tiles][Ndim][2]. */
int smsut_conv1x1_supported(int Kdim, int Ndim);
int smsut_conv1x1_tiles(int N, int HW, int Ndim);
int smsut_conv1x1_fwd(const float* x, const float* w, float* y, float* stats /*nullable*/, int N, int HW, int Kdim,
                      int Ndim, int transposed, void* stream);
int64_t smsut_conv1x1_wgrad_ws(int N, int HW, int Cin, int Cout);
int smsut_conv1x1_wgrad(const float* x, const float* gy, float* gw, float* workspace, int N, int HW, int Cin, int Cout,
                        void* stream);

/* input-side InstanceNorm + LeakyReLU: the operand is a = lrelu(IN(x; mean[N,C], rstd[N,C], gamma[C], beta[C])) of the tensor
   passed, normalised while the tiles are staged -- conv2 of a BasicBlock (network/blocks.py:70-72) and its weight gradient
   read the raw conv1 output, so the activation a1 is never written to or read from HBM.  Same in_affine() fma as the IN
   kernels: results equal the two-pass path bit for bit.  Forward: persistent-kernel shapes (smsut_conv2d_mfma_persistent). */
int smsut_conv2d_fwd_mfma_stats_inaff(const float* x, const float* w, float* y, float* stats, const float* mean,
                                      const float* rstd, const float* gamma, const float* beta, float slope, int N, int H,
                                      int W, int Kdim, int Ndim, void* stream);
int smsut_conv2d_wgrad_mfma_inaff(const float* x, const float* gy, float* gw, float* workspace, const float* mean,
                                  const float* rstd, const float* gamma, const float* beta, float slope, int N, int H, int W,
                                  int Cin, int Cout, void* stream);
/* measurement entry point (bench.py's roofline leg): the first of the two launches of smsut_conv2d_wgrad_mfma[_inaff] alone -- the
   register-row weight-gradient kernel (csrc/conv_wgrad_rr.hip) writing its per-split slabs, without the reduction; returns the
   number of slabs (> 0) or a negative value when the shape is not one that kernel takes.  mean == NULL: plain form. */
int smsut_conv2d_wgrad_mfma_slabs(const float* x, const float* gy, float* workspace, const float* mean, const float* rstd,
                                  const float* gamma, const float* beta, float slope, int N, int H, int W, int Cin, int Cout,
                                  void* stream);
/* which arithmetic a 3x3 stride-1 fp32 conv call of this shape runs: 0 direct, 1 Winograd F(2x2,3x3) resident weights, 2 Winograd
   streamed weights (16 instead of 36 products per 2x2 output tile); sc_dgrad: the fused shortcut data-gradient (Kdim = 2 Cout) */
int smsut_conv2d_mfma_form(int N, int H, int W, int Kdim, int Ndim, int sc_dgrad);

/* virtual-cat input forms: the logical input is cat([xa, xb], channel) (UpSampleAndConcat, network/blocks.py:49-50) read
   from the two tensors in place -- same chunk order and arithmetic as on a materialised cat, so results are bit-identical.
   ca % 16 == 0.  conv2d forward: persistent kernel, Kdim in {32, 64}, xa and xb of Kdim/2 channels each (ask _cat_supported). */
int smsut_conv2d_mfma_cat_supported(int N, int H, int W, int Kdim, int Ndim);
int smsut_conv2d_fwd_mfma_stats_cat(const float* xa, const float* xb, const float* w, float* y, float* stats, int N, int H,
                                    int W, int Kdim, int Ndim, void* stream);
/* conv1 of a BasicBlock fused with the block's 1x1 shortcut conv (reference network/blocks.py:66-80: `self.conv1(x)` and
 * `self.shortcut(x)` read the same x): y = conv3x3(x, w), ysc = conv1x1(x, wsc) and the InstanceNorm partials of both in one
 * pass.  xb nullable: non-null = x is the virtual cat([x, xb]) of two [N,H,W,Kdim/2] tensors.  _supported(..., cat) first. */
int smsut_conv2d_fwd_sc_supported(int N, int H, int W, int Kdim, int Ndim, int cat);
int smsut_conv2d_fwd_mfma_stats_sc(const float* x, const float* xb /*nullable*/, const float* w, const float* wsc, float* y,
                                   float* ysc, float* stats, float* stats_sc, int N, int H, int W, int Kdim, int Ndim,
                                   void* stream);
/* the same with fp16 operands (config 5): persistent kernel's direct form, Kdim in {16, 32, 64}; tiles with f16 = 1 */
int smsut_conv2d_fwd_sc_f16_supported(int N, int H, int W, int Kdim, int Ndim, int cat);
int smsut_conv2d_fwd_mfma_stats_sc_f16(const float* x, const float* xb, const float* w, const float* wsc, float* y, float* ysc,
                                       float* stats, float* stats_sc, int N, int H, int W, int Kdim, int Ndim, void* stream);
/* ... and the data-gradient of that pair in one pass: gx = dgrad3x3(gy, w) + dgrad1x1(gs, wsc) (the backward of
 * `out = conv1(x) ... + shortcut(x)` w.r.t. x, blocks.py:66-80).  gxb nullable: non-null = channels [0, split) of gx go to
 * gxa, the rest to gxb (the block input was cat([up, skip])).  _supported(..., split or 0) first. */
int smsut_conv2d_dgrad_sc_supported(int N, int H, int W, int Cout, int Cin, int split);
int smsut_conv2d_dgrad_mfma_sc(const float* gy, const float* gs, const float* w, const float* wsc, float* gxa,
                               float* gxb /*nullable*/, int split, int N, int H, int W, int Cout, int Cin, void* stream);
/* ... and both weight gradients in one pass: gw10 [10][Cin][Cout], rows 0..8 = conv1's 3x3 gradient (HWIO), row 9 = the 1x1
 * shortcut's; gs = gradient of the shortcut's output.  xb nullable (virtual cat, ca channels in xa). */
int smsut_conv2d_wgrad_sc_supported(int N, int H, int W, int Cin, int Cout);
int64_t smsut_conv2d_wgrad_sc_ws(int N, int H, int W, int Cin, int Cout);
int smsut_conv2d_wgrad_mfma_sc(const float* xa, const float* xb /*nullable*/, int ca, const float* gy, const float* gs,
                               float* gw10, float* workspace, int N, int H, int W, int Cin, int Cout, void* stream);
int smsut_conv2d_wgrad_mfma_cat(const float* xa, const float* xb, int ca, const float* gy, float* gw, float* workspace,
                                int N, int H, int W, int Cin, int Cout, int KS, void* stream);
/* `_fin` forms (r05) of the three statistics-producing convs of a fused BasicBlock: the same launch, with the InstanceNorm statistics
 * FINALISED INSIDE IT by the workgroup whose partials complete an image (fixed-order fp64 combine, the bits smsut_in_finalize_fwd /
 * _fwd2 / _bwd produce) -- the separate finalize launch behind each of them disappears (reference: conv -> nn.InstanceNorm2d,
 * network/blocks.py:70-79, and its backward).  tickets: int [N], ZERO on entry and zero again when the launch is over (the caller
 * keeps one zero-initialised pool and hands out slices; slices of launches that may run concurrently must not overlap).
 * wu: nullable prepared Winograd image (the `_pre` argument).  Partials are still written (stats / stats_sc), outputs [N][Ndim]. */
int smsut_conv2d_fwd_mfma_stats_sc_fin(const float* x, const float* xb /*nullable*/, const float* w, const float* wsc, float* y,
                                       float* ysc, float* stats, float* stats_sc, int* tickets, float* mean, float* rstd,
                                       float* mean_sc, float* rstd_sc, float eps, int N, int H, int W, int Kdim, int Ndim,
                                       const float* wu /*nullable*/, void* stream);
int smsut_conv2d_fwd_mfma_stats_inaff_fin(const float* x, const float* w, float* y, float* stats, const float* mean,
                                          const float* rstd, const float* gamma, const float* beta, float slope, int* tickets,
                                          float* mean_out, float* rstd_out, float eps, int N, int H, int W, int Kdim, int Ndim,
                                          const float* wu /*nullable*/, void* stream);
int smsut_conv2d_dgrad_mfma_bwdstats_fin(const float* gy, const float* w, float* gz, float* stats, const float* y1,
                                         const float* mean, const float* rstd, const float* gamma, const float* beta, float slope,
                                         int* tickets, float* a_mean, float* b_mean, int N, int H, int W, int Kdim, int Ndim,
                                         const float* wu /*nullable*/, void* stream);
/* PAIRED 3x3 weight gradient: gw = wgrad(set A) + wgrad(set B) in ONE launch, for two image sets that went through the same conv
 * (the reference differentiates every generator layer twice per iteration: G(x_real) and the cycle pass G(x_fake),
 * trainer/uganConsisTrainer.py:152,159,179 -- autograd sums the two weight gradients; here they are summed in the kernel's
 * accumulators).  Forms as the single-set entry points, present in both sets or in neither: x2 (virtual cat, ca channels in x),
 * mean / rstd [N, Cin] (x is the raw conv output, lrelu(IN(.)) applied in flight; gamma, beta, slope of the shared layer), gs (fused
 * 1x1 shortcut: gw holds 10 rows).  cat / aff / sc in the queries: 0 / 1 for those forms.  workspace: _ws floats. */
int smsut_conv2d_wgrad_pair_supported(int NA, int NB, int H, int W, int Cin, int Cout, int cat, int aff, int sc);
int64_t smsut_conv2d_wgrad_pair_ws(int NA, int NB, int H, int W, int Cin, int Cout, int cat, int aff, int sc);
int smsut_conv2d_wgrad_pair(const float* xA, const float* x2A /*nullable*/, const float* gyA, const float* gsA /*nullable*/,
                            const float* meanA /*nullable*/, const float* rstdA /*nullable*/, int NA, const float* xB,
                            const float* x2B /*nullable*/, const float* gyB, const float* gsB /*nullable*/,
                            const float* meanB /*nullable*/, const float* rstdB /*nullable*/, int NB, int ca,
                            const float* gamma /*nullable*/, const float* beta /*nullable*/, float slope, float* gw, float* workspace,
                            int H, int W, int Cin, int Cout, void* stream);
/* measurement twin of smsut_conv2d_wgrad_mfma_slabs for the paired launch (plain / input-side-IN form): the kernel alone, slabs left in
 * the workspace; returns the slab count (> 0) or a negative value */
int smsut_conv2d_wgrad_pair_slabs(const float* xA, const float* gyA, const float* meanA /*nullable*/, const float* rstdA /*nullable*/,
                                  int NA, const float* xB, const float* gyB, const float* meanB /*nullable*/,
                                  const float* rstdB /*nullable*/, int NB, const float* gamma /*nullable*/, const float* beta /*nullable*/,
                                  float slope, float* workspace, int H, int W, int Cin, int Cout, void* stream);
int smsut_conv1x1_fwd_cat(const float* xa, const float* xb, int ca, const float* w, float* y, float* stats /*nullable*/,
                          int N, int HW, int Kdim, int Ndim, void* stream);
int smsut_conv1x1_wgrad_cat(const float* xa, const float* xb, int ca, const float* gy, float* gw, float* workspace, int N,
                            int HW, int Cin, int Cout, void* stream);

/* split-output forms: result channels [0, split) -> ya (pixel stride split), [split, Ndim) -> yb (stride Ndim - split).
   They write the data-gradient of a block whose input was cat([up, skip]) (network/blocks.py:49) straight into the two
   gradient tensors.  conv2d form: persistent kernel shapes only (ask _split_supported), any `transposed` form of
   smsut_conv2d_fwd_mfma; conv1x1 form: split % 16 == 0. */
int smsut_conv2d_mfma_split_supported(int N, int H, int W, int Kdim, int Ndim, int split);
int smsut_conv2d_fwd_mfma_split(const float* x, const float* w, float* ya, float* yb, int split, int N, int H, int W,
                                int Kdim, int Ndim, int transposed, void* stream);
int smsut_conv1x1_fwd_split(const float* x, const float* w, float* ya, float* yb, int split, int N, int HW, int Kdim,
                            int Ndim, int transposed, void* stream);

/* fp16-operand convolutions (BASELINE config 5: 512x512 slices, "fp16 MFMA conv path with fp32 IN / loss accumulators"; the
 * reference has no AMP -- SURVEY.md 2.2 -- so these replace the same nn.Conv2d sites as the fp32 entry points: network/blocks.py:10-16,
 * 53-80 at the shapes of network/ugan.py:205-215, config.py:50).  Tensors stay fp32 in HBM; operands are converted to fp16 while a tile
 * is staged into LDS, products accumulate in fp32 (v_mfma_f32_16x16x16_f16), statistics / epilogues / outputs are the fp32 ones.
 * gsc (nullable): device float[2] = {s, 1/s} from smsut_absmax_scale -- a per-tensor power-of-two scale for a GRADIENT input operand
 * (it would underflow fp16 otherwise); the operand is multiplied by s before the conversion, the fp32 result by 1/s.
 * Statistics layout as the fp32 entry points with the tiles of smsut_conv2d_mfma_tiles(..., f16 = 1); `transposed` flags alike. */
int smsut_conv2d_f16_supported(int KS, int Kdim, int Ndim);
int smsut_conv2d_fwd_mfma_f16(const float* x, const float* w, float* y, const float* gsc /*nullable*/, int N, int H, int W,
                              int Kdim, int Ndim, int KS, int transposed, void* stream);
int smsut_conv2d_fwd_mfma_stats_f16(const float* x, const float* w, float* y, float* stats, int N, int H, int W, int Kdim,
                                    int Ndim, int KS, void* stream);
int smsut_conv2d_fwd_mfma_stats_cat_f16(const float* xa, const float* xb, const float* w, float* y, float* stats, int N, int H,
                                        int W, int Kdim, int Ndim, void* stream);
int smsut_conv2d_fwd_mfma_split_f16(const float* x, const float* w, float* ya, float* yb, const float* gsc /*nullable*/,
                                    int split, int N, int H, int W, int Kdim, int Ndim, int transposed, void* stream);
/* fp16 STORAGE of a BasicBlock's internal raw conv outputs y1, y2, s (config 5, "_hs" = half storage): the conv epilogues store
 * fp16 [N,H,W,Ndim] (InstanceNorm partials still from the fp32 accumulators), the consumers below and in the InstanceNorm section
 * read fp16 and compute in fp32.  Persistent-kernel shapes, Kdim in {16, 32, 64}; xb nullable (virtual cat).  Kdim == 8 (first
 * block after the stem): the fused-shortcut entry only, on fp32 operands (the 8-channel form has no fp16 twin), same storage. */
int smsut_conv2d_f16_hs_supported(int N, int H, int W, int Kdim, int Ndim, int cat);
int smsut_conv2d_fwd_mfma_stats_f16_hs(const float* x, const float* xb /*nullable*/, const float* w, void* y16, float* stats, int N,
                                       int H, int W, int Kdim, int Ndim, void* stream);
/* ... whose input is fp16 too (conv2 of the block reading the a1 that smsut_instnorm_fwd_partials_hs2 stored): same operand bits */
int smsut_conv2d_fwd_mfma_stats_f16_hsx(const void* x16, const float* w, void* y16, float* stats, int N, int H, int W, int Kdim,
                                        int Ndim, void* stream);
/* ... reading the RAW fp16 conv1 output and applying lrelu(IN(.)) while staging (half-storage twin of ..._stats_inaff) */
int smsut_conv2d_fwd_mfma_stats_inaff_f16_hsx(const void* y1_16, const float* w, void* y16, float* stats, const float* mean,
                                              const float* rstd, const float* gamma, const float* beta, float slope, int N, int H,
                                              int W, int Kdim, int Ndim, void* stream);
int smsut_conv2d_fwd_mfma_stats_sc_f16_hs(const float* x, const float* xb /*nullable*/, const float* w, const float* wsc, void* y16,
                                          void* ysc16, float* stats, float* stats_sc, int N, int H, int W, int Kdim, int Ndim,
                                          void* stream);
int smsut_conv2d_dgrad_mfma_bwdstats_f16_hs(const float* gy, const float* w, float* gz, float* stats, const void* y1_16,
                                            const float* mean, const float* rstd, const float* gamma, const float* beta,
                                            const float* gsc /*nullable*/, float slope, int N, int H, int W, int Kdim, int Ndim,
                                            void* stream);
/* the fused-shortcut data-gradient with fp16 operands (config 5): gsc = {s, 1/s} of smsut_absmax_scale2(gy, gs) (the two gradients
 * share accumulators, hence ONE scale); Cout in {16, 32}, Cin >= 16, persistent-kernel shapes */
int smsut_conv2d_dgrad_sc_f16_supported(int N, int H, int W, int Cout, int Cin, int split);
int smsut_conv2d_dgrad_mfma_sc_f16(const float* gy, const float* gs, const float* w, const float* wsc, float* gxa,
                                   float* gxb /*nullable*/, const float* gsc, int split, int N, int H, int W, int Cout, int Cin,
                                   void* stream);
int smsut_conv2d_dgrad_mfma_bwdstats_f16(const float* gy, const float* w, float* gz, float* stats, const float* y1,
                                         const float* mean, const float* rstd, const float* gamma, const float* beta,
                                         const float* gsc /*nullable*/, float slope, int N, int H, int W, int Kdim, int Ndim,
                                         void* stream);
/* 3x3 weight gradient, fp16 operands read transposed from LDS (ds_read_b64_tr_b16): H % 8 == 0, W % 16 == 0, Cin % 16 == 0,
 * Cout % 16 == 0.  x2 (nullable): x is the virtual cat([x, x2]) with ca channels in x.  workspace: _ws floats. */
int smsut_conv2d_wgrad_f16_supported(int N, int H, int W, int Cin, int Cout);
int64_t smsut_conv2d_wgrad_f16_ws(int N, int H, int W, int Cin, int Cout);
int smsut_conv2d_wgrad_f16(const float* x, const float* x2 /*nullable*/, int ca, const float* gy, float* gw, float* workspace,
                           const float* gsc /*nullable*/, int N, int H, int W, int Cin, int Cout, void* stream);
/* ... with x stored as fp16 (half storage: conv2's weight gradient reads the activated a1): same operand bits, same result */
int smsut_conv2d_wgrad_f16_xh(const void* x16, const float* gy, float* gw, float* workspace, const float* gsc /*nullable*/, int N,
                              int H, int W, int Cin, int Cout, void* stream);
/* ... with x the RAW fp16 conv1 output, lrelu(IN(.)) applied while staging (half-storage twin of smsut_conv2d_wgrad_mfma_inaff) */
int smsut_conv2d_wgrad_f16_xh_inaff(const void* y1_16, const float* gy, float* gw, float* workspace, const float* gsc /*nullable*/,
                                    const float* mean, const float* rstd, const float* gamma, const float* beta, float slope, int N,
                                    int H, int W, int Cin, int Cout, void* stream);
/* ... plus the 1x1 shortcut's weight gradient in the same pass (fp16 twin of smsut_conv2d_wgrad_mfma_sc): gw10 [10][Cin][Cout],
 * rows 0..8 the 3x3 taps, row 9 the shortcut; gsc from smsut_absmax_scale2(gy, gs) */
int smsut_conv2d_wgrad_sc_f16_supported(int N, int H, int W, int Cin, int Cout);
int64_t smsut_conv2d_wgrad_sc_f16_ws(int N, int H, int W, int Cin, int Cout);
int smsut_conv2d_wgrad_sc_f16(const float* x, const float* x2 /*nullable*/, int ca, const float* gy, const float* gs, float* gw10,
                              float* workspace, const float* gsc, int N, int H, int W, int Cin, int Cout, void* stream);
/* out2[2] = {s, 1/s}, s = 2^k with max|x| * s in [2^13, 2^14] (s = 1 for an all-zero tensor); workspace: _ws floats.
 * _scale2: one scale for two tensors (max over both; the same workspace size serves it) */
int64_t smsut_absmax_scale_ws(int64_t n);
int smsut_absmax_scale(const float* x, int64_t n, float* out2, float* workspace, void* stream);
int smsut_absmax_scale2(const float* x, int64_t n, const float* x2, int64_t n2, float* out2, float* workspace, void* stream);
/* the same scale from maxima the PRODUCING kernels handed over (smsut_restail_bwd_amax, smsut_in_apply_bwd_amax: one slot per
 * workgroup and output tensor): out2 = scale of max(amax[0..n)) -- no pass over the gradient tensor */
int smsut_absmax_finish(const float* amax, int n, float* out2, void* stream);

/* 4x4 stride-1 pad-1 convolutions on the matrix cores: networks.NLayerDiscriminator / PatchDiscriminator (reference
 * network/networks.py:977-1032, 64 -> 128 -> 256 -> 512 channels).  (H, W) = extents of the conv's forward input; the output
 * is (H-1) x (W-1).  transposed = 0: x [N,H,W,Cin] -> y [N,H-1,W-1,Cout]; transposed = 1 (data-gradient): x = gy
 * [N,H-1,W-1,Cout] -> y = gx [N,H,W,Cin].  Weights [4][4][Cin][Cout]; Cin, Cout multiples of 4. */
int smsut_conv2d_k4_supported(int Cin, int Cout);
int smsut_conv2d_k4_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, int transposed,
                        void* stream);
int64_t smsut_conv2d_k4_wgrad_ws(int N, int H, int W, int Cin, int Cout);
int smsut_conv2d_k4_wgrad(const float* x, const float* gy, float* gw, float* workspace, int N, int H, int W, int Cin, int Cout,
                          void* stream);

/* thin 1x1 layers (Cout <= 8, Cin in {8,16,32,64}; reference: the nn.Conv2d heads at network/blocks.py:123-125 and
   network/ugan.py:70): data-gradient gx[P][Cin] = gy[P][Cout] W^T and weight-gradient gw[Cin][Cout], both streaming. */
int smsut_conv1x1_thin_supported(int Cin, int Cout);
int smsut_conv1x1_thin_dgrad(const float* gy, const float* w, float* gx, int N, int HW, int Cin, int Cout, void* stream);
int64_t smsut_conv1x1_thin_wgrad_ws(int Cin);
int smsut_conv1x1_thin_wgrad(const float* x, const float* gy, float* gw, float* workspace, int N, int HW, int Cin,
                             int Cout, void* stream);

/* Tiny-channel convolutions (HBM-bound): 5x5 stems (blocks.py:123, ugan.py:26), D's k4 s2 stem (ugan.py:202), 1x1 heads
 * (blocks.py:166, ugan.py:70).  fwd/dgrad: direct, Cin <= 8, Cout in {4,8,12,16}; wgrad: MFMA with the flattened
 * (tap, ci) index as M, KS*KS*Cin <= 128, Cout <= 16. */
int smsut_conv2d_small_supported(int KS, int Cin, int Cout);
int smsut_conv2d_small_fwd(const float* x, const float* w, const float* bias /*nullable*/, float* y, int N, int H, int W,
                           int Cin, int Ho, int Wo, int Cout, int KS, int stride, int pad, void* stream);
int smsut_conv2d_small_dgrad(const float* gy, const float* w, float* gx, int N, int H, int W, int Cin, int Ho, int Wo,
                             int Cout, int KS, int stride, int pad, void* stream);
int smsut_conv2d_flat_wgrad_supported(int KS, int stride, int Cin, int Cout);
int64_t smsut_conv2d_flat_wgrad_ws(int N, int Ho, int Wo, int Cin, int Cout, int KS);
int smsut_conv2d_flat_wgrad(const float* x, const float* gy, float* gw, float* workspace, int N, int H, int W, int Cin,
                            int Ho, int Wo, int Cout, int KS, int stride, int pad, void* stream);

/* nn.ConvTranspose2d(in, out, kernel_size=2, stride=2, bias=False) -- network/blocks.py:41 (U-Net / seg decoder up path). */
int smsut_convT2x2_mfma_supported(int Cin, int Cout);
int smsut_convT2x2_fwd_mfma(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, void* stream);
int smsut_convT2x2_dgrad_mfma(const float* gy, const float* w, float* gx, int N, int H, int W, int Cin, int Cout,
                              void* stream);
int64_t smsut_convT2x2_wgrad_mfma_ws(int N, int H, int W, int Cin, int Cout);
int smsut_convT2x2_wgrad_mfma(const float* x, const float* gy, float* gw, float* workspace, int N, int H, int W, int Cin,
                              int Cout, void* stream);
/* The same transposed conv through the 1x1 kernels' pixel-shuffle forms (x read once per 64-column slab of the 4 * Cout tap-major
 * columns instead of once per tap; Cout % 16 == 0): forward and weight gradient; the data gradient is already one pass. */
int smsut_convT2x2_ps_supported(int Cin, int Cout);
int smsut_convT2x2_fwd_ps(const float* x, const float* w, float* y, int N, int H, int W, int Cin, int Cout, void* stream);
int64_t smsut_convT2x2_wgrad_ps_ws(int N, int H, int W, int Cin, int Cout);
int smsut_convT2x2_wgrad_ps(const float* x, const float* gy, float* gw, float* workspace, int N, int H, int W, int Cin,
                            int Cout, void* stream);

/* bias gradient: out[c] = sum over rows of x[rows][C] */
int64_t smsut_colsum_ws(int64_t rows, int C);
int smsut_colsum(const float* x, float* out, float* workspace, int64_t rows, int C, void* stream);

/* ---------------------------------------------------------------------------------------------- instance norm
 * nn.InstanceNorm2d(C, affine=True) (+ the following LeakyReLU/ReLU) -- network/blocks.py:19-32,57-60,64.
 * _bwd2 is the backward of _bwd, needed because WGAN-GP differentiates D's backward
 * (trainer/uganShp0Trainer.py:127-134, create_graph=True). */
int smsut_in_chunks(int N, int HW, int C); /* workspace = N * chunks * C * 3 floats */
int smsut_instnorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       float* workspace, int N, int HW, int C, float eps, float slope, int has_act, void* stream);
int smsut_instnorm_fwd_partials(const float* x, const float* gamma, const float* beta, float* y, float* mean,
                                float* rstd, const float* partials, int chunks, int N, int HW, int C, float eps,
                                float slope, int has_act, void* stream);
/* ... reading an fp16 x (half storage, see smsut_conv2d_f16_hs_supported); y stays fp32; C % 4 == 0 */
int smsut_instnorm_fwd_partials_hs(const void* x16, const float* gamma, const float* beta, float* y, float* mean,
                                   float* rstd, const float* partials, int chunks, int N, int HW, int C, float eps,
                                   float slope, int has_act, void* stream);
/* ... and y stored as fp16 as well (consumers: smsut_conv2d_fwd_mfma_stats_f16_hsx, smsut_conv2d_wgrad_f16_xh) */
int smsut_instnorm_fwd_partials_hs2(const void* x16, const float* gamma, const float* beta, void* y16, float* mean,
                                    float* rstd, const float* partials, int chunks, int N, int HW, int C, float eps,
                                    float slope, int has_act, void* stream);
int smsut_in_finalize_fwd(const float* partials, int chunks, float* mean, float* rstd, int N, int HW, int C, float eps,
                          void* stream);
/* the same for TWO partial sets of one (N, HW, C) in one launch (conv2's and the shortcut's statistics of a BasicBlock,
   network/blocks.py:66-80, are both due when the residual tail starts) */
int smsut_in_finalize_fwd2(const float* pa, int chunks_a, float* mean_a, float* rstd_a, const float* pb, int chunks_b, float* mean_b,
                           float* rstd_b, int N, int HW, int C, float eps, void* stream);
/* InstanceNorm backward fed by the dgrad epilogue (smsut_conv2d_dgrad_mfma_bwdstats): finalise the {sum gz, sum gz*xhat}
 * partials into a / b, then gx = gamma*rstd*(gz - a - xhat*b) on the ALREADY masked gz (+ affine gradients, nullable). */
int smsut_in_finalize_bwd(const float* partials, int chunks, float* a_mean, float* b_mean, int N, int HW, int C, void* stream);
int smsut_in_apply_bwd(const float* gz, const float* x, const float* mean, const float* rstd, const float* gamma,
                       const float* a_mean, const float* b_mean, float* gx, float* ggamma /*nullable*/,
                       float* gbeta /*nullable*/, int N, int HW, int C, void* stream);
/* ... that also hands over max|gx|: every workgroup writes the maximum of what it stored into its own slot, amax[0 .. B),
 * B = smsut_amax_blocks(N, HW, C) (no atomics, no zeroing); fp16-operand consumers reduce the slots with smsut_absmax_finish */
int smsut_amax_blocks(int N, int HW, int C);
int smsut_in_apply_bwd_amax(const float* gz, const float* x, const float* mean, const float* rstd, const float* gamma,
                            const float* a_mean, const float* b_mean, float* gx, float* ggamma /*nullable*/,
                            float* gbeta /*nullable*/, float* amax, int N, int HW, int C, void* stream);
/* ... reading an fp16 x (half storage, see smsut_conv2d_f16_hs_supported); amax nullable; C % 4 == 0 */
int smsut_in_apply_bwd_hs(const float* gz, const void* x16, const float* mean, const float* rstd, const float* gamma,
                          const float* a_mean, const float* b_mean, float* gx, float* ggamma /*nullable*/,
                          float* gbeta /*nullable*/, float* amax /*nullable*/, int N, int HW, int C, void* stream);
/* residual tail of BasicBlock (blocks.py:60-79): out = act(IN(y2) + (IN(s) | s)), forward and backward in one pass each */
int smsut_restail_fwd(const float* y2, const float* m2, const float* r2, const float* g2, const float* b2, const float* s,
                      const float* ms /*nullable: identity*/, const float* rs, const float* gs, const float* bs, float* out,
                      int N, int HW, int C, float slope, void* stream);
int smsut_restail_bwd(const float* gout, const float* out, const float* y2, const float* m2, const float* r2,
                      const float* g2, const float* b2 /*nullable*/, const float* s, const float* ms /*nullable*/,
                      const float* rs, const float* gs_, const float* bs /*nullable*/, float* gy2, float* gs, float* a_mean, float* b2_mean, float* bs_mean, float* gg2, float* gb2,
                      float* ggs /*nullable*/, float* gbs /*nullable*/, float* workspace, int N, int HW, int C,
                      float slope, void* stream);
/* ... with the per-image means finalised inside the partial-sum launch (r05; tickets: int [N], zero on entry and on exit): two
 * launches instead of three, same bits */
int smsut_restail_bwd_fin(const float* gout, const float* out, const float* y2, const float* m2, const float* r2,
                      const float* g2, const float* b2 /*nullable*/, const float* s, const float* ms /*nullable*/,
                      const float* rs, const float* gs_, const float* bs /*nullable*/, float* gy2, float* gs, float* a_mean, float* b2_mean, float* bs_mean, float* gg2, float* gb2,
                      float* ggs /*nullable*/, float* gbs /*nullable*/, float* workspace, int* tickets, int N, int HW, int C,
                      float slope, void* stream);
/* half storage: y2 and s are fp16 [N,HW,C]; conv shortcut with both betas (ms, b2, bs non-null); C % 4 == 0; amax nullable */
int smsut_restail_fwd_hs(const void* y2_16, const float* m2, const float* r2, const float* g2, const float* b2, const void* s16,
                         const float* ms, const float* rs, const float* gs, const float* bs, float* out, int N, int HW, int C,
                         float slope, void* stream);
int smsut_restail_bwd_hs(const float* gout, const float* out, const void* y2_16, const float* m2, const float* r2,
                         const float* g2, const float* b2, const void* s16, const float* ms, const float* rs, const float* gs_,
                         const float* bs, float* gy2, float* gs, float* a_mean, float* b2_mean, float* bs_mean, float* gg2,
                         float* gb2, float* ggs, float* gbs, float* workspace, float* amax /*nullable*/, int N, int HW, int C,
                         float slope, void* stream);
/* ... that also hands over max|gy2| in amax[0 .. B) and max|gs| in amax[B .. 2B), B = smsut_amax_blocks(N, HW, C) */
int smsut_restail_bwd_amax(const float* gout, const float* out, const float* y2, const float* m2, const float* r2,
                           const float* g2, const float* b2 /*nullable*/, const float* s, const float* ms /*nullable*/,
                           const float* rs, const float* gs_, const float* bs /*nullable*/, float* gy2, float* gs, float* a_mean, float* b2_mean, float* bs_mean, float* gg2, float* gb2,
                           float* ggs /*nullable*/, float* gbs /*nullable*/, float* workspace, float* amax, int N, int HW,
                           int C, float slope, void* stream);
/* The tail of an ENCODER level's block together with the level's MaxPool2d(2, 2) (r05; /root/reference/network/blocks.py:74-79 +
 * 131-133, network/ugan.py:36-39).  Forward: out [N,H,W,C] (the skip connection), pooled [N,H/2,W/2,C] and idx (one byte per pooled
 * element: where the maximum sat in its window) in ONE pass -- bit-identical to smsut_restail_fwd + smsut_maxpool2_fwd.  Backward:
 * gout = gradient of out through the skip connection, gp = gradient of pooled; the block output's total gradient (what
 * smsut_maxpool2_bwd_add writes) is formed while loading -- bit-identical to that call followed by smsut_restail_bwd / _fin (tickets
 * non-null) / _amax (amax non-null) / _hs (hs != 0: y2, s are _Float16).  Conv shortcut only (ms, b2, bs non-null), C % 4 == 0, H, W even.
 * idx == NULL in both calls: AVERAGE pooling instead (a stride-2 BottleBlock feeding the next one, blocks.py:99-107: its output goes to
 * conv1 and to F.avg_pool2d) -- smsut_avgpool2_fwd's arithmetic forward, gout + 0.25 gp per window pixel backward. */
int smsut_restail_fwd_pool(const void* y2, const float* m2, const float* r2, const float* g2, const float* b2, const void* s,
                           const float* ms, const float* rs, const float* gs, const float* bs, float* out, float* pooled,
                           void* idx /*nullable*/, int N, int H, int W, int C, float slope, int hs, void* stream);
int smsut_restail_bwd_pool(const float* gout, const float* gp, const void* idx /*nullable*/, const void* y2, const float* m2, const float* r2,
                           const float* g2, const float* b2, const void* s, const float* ms, const float* rs, const float* gs_,
                           const float* bs, float* gy2, float* gs, float* a_mean, float* b2_mean, float* bs_mean, float* gg2,
                           float* gb2, float* ggs, float* gbs, float* workspace, int* tickets /*nullable*/, float* amax /*nullable*/,
                           int N, int H, int W, int C, float slope, int hs, void* stream);
int smsut_instnorm_bwd(const float* gy, const float* x, const float* beta /*nullable: no activation*/, const float* mean,
                       const float* rstd, const float* gamma, float* gx, float* a_mean, float* b_mean,
                       float* ggamma /*nullable*/, float* gbeta /*nullable*/, float* workspace, int N, int HW, int C,
                       float slope, void* stream);
/* InstanceNorm + LeakyReLU + AvgPool2d(2) as ONE op (r05): bn1 -> relu -> avgpool of a stride-2 BottleBlock
 * (/root/reference/network/blocks.py:99-107), for passes differentiated once.  Forward from a conv epilogue's statistics partials
 * (as smsut_instnorm_fwd_partials): x [N,H,W,C] raw conv output -> y [N,H/2,W/2,C]; mean / rstd [N,C] are outputs; H, W even.
 * Backward: gyp [N,H/2,W/2,C] = gradient of the POOLED output; other arguments as smsut_instnorm_bwd.  Both are bit-identical to
 * the two-op compositions (instnorm + avgpool2; avgpool2_bwd + instnorm_bwd) without the full-resolution intermediate. */
int smsut_instnorm_pool_fwd_partials(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                                     const float* partials, int chunks, int N, int H, int W, int C, float eps, float slope,
                                     void* stream);
int smsut_instnorm_pool_bwd(const float* gyp, const float* x, const float* beta, const float* mean, const float* rstd,
                            const float* gamma, float* gx, float* a_mean, float* b_mean, float* ggamma /*nullable*/,
                            float* gbeta /*nullable*/, float* workspace, int N, int H, int W, int C, float slope, void* stream);
int smsut_instnorm_bwd2(const float* v, const float* ug /*nullable*/, const float* ub /*nullable*/, const float* gy,
                        const float* x, const float* beta /*nullable: no activation*/, const float* mean, const float* rstd,
                        const float* gamma, const float* a_mean, const float* b_mean, float* d_gy, float* d_x,
                        float* d_gamma, float* workspace, float* scratch /*3*N*C*/, int N, int HW, int C, float slope,
                        void* stream);

/* ---------------------------------------------------------------------------------------------- pointwise / resampling
 * LeakyReLU + residual add (blocks.py:78-79,115-116), nn.Tanh (ugan.py:72-73), conv bias, F.avg_pool2d(x,2)
 * (blocks.py:101,107), nn.MaxPool2d(2,2) (blocks.py:128-134), nn.Upsample(x2, bilinear, align_corners=False)
 * (blocks.py:44), torch.cat (blocks.py:50, ugan.py:159), modality planes (ugan.py:156-157), x_hat (uganConsisTrainer.py:139). */
int smsut_add_act(const float* a, const float* b /*nullable*/, float* y, int64_t n, float slope, void* stream);
int smsut_act_bwd(const float* gy, const float* y, float* gx, int64_t n, float slope, void* stream);
int smsut_tanh_fwd(const float* x, float* y, int64_t n, void* stream);
int smsut_tanh_bwd(const float* gy, const float* y, float* gx, int64_t n, void* stream);
int smsut_bias_add(const float* x, const float* bias, float* y, int64_t rows, int C, void* stream);
int smsut_row_lerp(const float* a, const float* b, const float* alpha, float* out, int64_t rows, int64_t row_len,
                   void* stream);
int smsut_fill(float* out, float v, int64_t n, void* stream);
int smsut_scale(const float* x, const float* scale_dev /*nullable*/, float mul, float* out, int64_t n, void* stream);
/* SGD with momentum + weight decay over many tensors in ONE launch (r05): torch.optim.SGD's rule (dampening 0, no Nesterov; the
 * reference's optimizer, /root/reference/trainer/baseTrainer.py) -- g' = g + wd p; buf = momentum buf + g'; p -= lr buf.  ents: device
 * array of {float* p; const float* g; float* buf; long long n;} (parameter, gradient, momentum buffer: same dense layout); blk_ent /
 * blk_chunk: per block its entry and its chunk (of smsut_sgd_chunk() elements) inside that tensor.  Not for the first step (the buffers
 * must exist). */
int smsut_sgd_momentum_multi(const void* ents, const int* blk_ent, const int* blk_chunk, int nblocks, float lr, float momentum, float wd,
                             void* stream);
int smsut_sgd_chunk(void);
int smsut_maxpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int smsut_maxpool2_bwd(const float* gy, const float* x, float* gx, int N, int H, int W, int C, void* stream);
/* gx = maxpool2_bwd(gy; x) + add: pooled-path and skip-connection gradients of an encoder level (blocks.py:131-133) in one pass */
int smsut_maxpool2_bwd_add(const float* gy, const float* x, const float* add, float* gx, int N, int H, int W, int C,
                           void* stream);
int smsut_avgpool2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int smsut_avgpool2_bwd(const float* gy, float* gx, int N, int H, int W, int C, void* stream);
/* Device-side joint augmentation (rotate + elastic + random-resized-crop of data_loader/externalTransforms.py:45-90 as one
 * resampling pass): source position = aff[n] * (xo, yo, 1) + bilinear(ctrl[n][2][P][P]); image bilinear, labels nearest,
 * zeros outside.  img [N,H,W], msk [N,H,W] (nullable), aff [N][6], ctrl nullable when P == 0. */
int smsut_warp_joint(const float* img, const int64_t* msk, const float* aff, const float* ctrl, float* oimg,
                     int64_t* omsk, int N, int H, int W, int Ho, int Wo, int P, void* stream);
/* JointElasticDeform's resampling (reference data_loader/externalTransforms.py:69-90 -> elasticdeform.deform_random_grid(order = [0, 0])):
   coef [N][2][P][P] = cubic B-spline coefficients (mirror boundary) of the P x P control displacements (dy, dx), first / last control
   point on the first / last pixel; image and label map sampled at (y + dy, x + dx) with order 0, constant 0 outside. */
int smsut_elastic_deform(const float* img, const int64_t* msk, const float* coef, float* oimg, int64_t* omsk, int N, int H,
                         int W, int P, void* stream);
int smsut_bilinear2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int smsut_bilinear2_bwd(const float* gy, float* gx, int N, int H, int W, int C, void* stream);
/* networks.py building blocks of ResnetGenerator / NLayerDiscriminator: reflection / replication /
 * zero padding and cropping (mode 0 zero, 1 reflect, 2 replicate; (oy, ox) = (top, left) pad, negative = crop) and the
 * anti-aliased Downsample(filt 3, stride 2, reflect) -- networks.py:37-60,95-105.  Upsample(filt 4, 'repl', stride 2)
 * (networks.py:73-93) is arithmetically the x2 bilinear kernel above. */
int smsut_window_fwd(const float* src, float* dst, int N, int Hs, int Ws, int Hd, int Wd, int C, int oy, int ox, int mode,
                     void* stream);
int smsut_window_bwd(const float* gdst, float* gsrc, int N, int Hs, int Ws, int Hd, int Wd, int C, int oy, int ox, int mode,
                     void* stream);
int smsut_blurdown_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int smsut_blurdown_bwd(const float* gy, float* gx, int N, int H, int W, int C, void* stream);
/* torch.cat([a, b], 1) (dir 0) and its backward split (dir 1; a / b nullable) in one pass over the wide tensor. */
int smsut_concat2(float* a, int Ca, float* b, int Cb, float* y, int64_t P, int dir, void* stream);
int smsut_copy_channels(const float* src, int Cs, int src_off, float* dst, int Cd, int dst_off, int Cc, int64_t P,
                        void* stream);
int smsut_modal_planes(const float* x, const float* m, float* out, int N, int64_t HW, int Cx, int M, void* stream);

/* ---------------------------------------------------------------------------------------------- losses
 * DiceAndCrossEntropyLoss / SoftDiceLoss (misc/loss.py:8-63): stage 1 = per-class {tp, sum_p, count} + CE sum,
 * (optional all-reduce by the caller under data parallelism), stage 2 = scalar; G = 1 batch dice, G = N per sample. */
int64_t smsut_dicece_ws(int N, int64_t HW, int C, int G);
int smsut_dicece_stats(const float* logits, const int64_t* labels, float* stats, float* ce_sum, float* workspace, int N,
                       int64_t HW, int C, int G, void* stream);
int smsut_dicece_final(const float* stats, const float* ce_sum, float* out /*3*/, int G, int C, double npix_total,
                       float w_dc, float w_ce, void* stream);
int smsut_dicece_bwd(const float* logits, const int64_t* labels, const float* stats, const float* gout, float* glogits,
                     int N, int64_t HW, int C, int G, double npix_total, float w_dc, float w_ce, void* stream);
/* -/+ mean(out_src) (uganConsisTrainer.py:130,136,154), mean|a-b| (:162), F.cross_entropy on [B,n_modal] (:131,155),
 * gradient penalty norm term (uganShp0Trainer.py:131-134). */
int64_t smsut_sum_ws(int64_t n, int rows);
int smsut_sum(const float* a, float* out, float* workspace, int64_t n, double scale, void* stream);
int smsut_l1_fwd(const float* a, const float* b, float* out, float* workspace, int64_t n, void* stream);
int smsut_l1_bwd(const float* a, const float* b, const float* gout, float* ga /*nullable*/, float* gb /*nullable*/,
                 int64_t n, void* stream);
/* mean((softmax(a) - softmax(b))^2) over [P pixels][C] NHWC logits, gradient to a only: the mean-teacher consistency
 * term (trainer/meanTeacherTrainer.py:113-131).  workspace: smsut_sum_ws(P, 1) floats. */
int smsut_softmax_mse_fwd(const float* a, const float* b, float* out, float* workspace, int64_t P, int C, void* stream);
int smsut_softmax_mse_bwd(const float* a, const float* b, const float* gout, float* ga, int64_t P, int C, void* stream);
/* labels[p] = argmax_c logits[p][c] (first maximum): pseudo labels (uganConsisTrainer.py:45-53, crossPseTrainer.py:122-127)
 * and the prediction map of validate_epoch (uganShp0Trainer.py:262-266). */
int smsut_argmax_channels(const float* logits, int64_t* labels, int64_t P, int C, void* stream);
int smsut_gp_fwd(const float* dydx, float* out, float* norms, float* workspace, int rows, int64_t n, void* stream);
int smsut_gp_bwd(const float* dydx, const float* norms, const float* gout, float* g, int rows, int64_t n, void* stream);
int smsut_ce_rows_fwd(const float* z, const int64_t* tgt, float* out, int B, int C, void* stream);
int smsut_ce_rows_bwd(const float* z, const int64_t* tgt, const float* gout, float* gz, int B, int C, void* stream);
/* PatchSampleF gather (ugan.py:318-327), networks.Normalize (networks.py:234-243), PatchNCELoss (patchnce.py:13-51). */
int smsut_gather_rows(const float* feat, const int64_t* ids, float* out, int B, int64_t HW, int C, int P, void* stream);
int smsut_scatter_rows(const float* gout, const int64_t* ids, float* gfeat, int B, int64_t HW, int C, int P,
                       void* stream);
int smsut_l2norm_fwd(const float* x, float* y, float* norms, int rows, int C, void* stream);
int smsut_l2norm_bwd(const float* gy, const float* x, const float* norms, float* gx, int rows, int C, void* stream);
int smsut_patchnce_fwd(const float* q, const float* k, float* loss, float* probs, int rows, int np, int dim, float T,
                       void* stream);
int smsut_patchnce_bwd(const float* gloss, const float* probs, const float* k, float* gq, int rows, int np, int dim,
                       float T, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMSUT_HIP_H */
