/*
 * vegans_hip.h — C ABI of libvegans_hip.so: the MI355X (gfx950) kernels behind the
 * VeGANs adversarial training step.
 *
 * The reference (tneuer/GAN-pytorch) has no FFI of its own: every arithmetic call on its
 * training path is a torch call.  Each entry point below names the reference call site
 * (file:line under the reference tree) whose arithmetic it replaces.
 *
 * Conventions
 *  - all tensors are float32 device memory; activations are NHWC ([N][H][W][C]),
 *    conv weights are [Co][kh][kw][Ci] (torch "channels_last" storage of a [Co,Ci,kh,kw]
 *    parameter; a ConvTranspose2d weight [Cin,Cout,kh,kw] stored channels_last is the
 *    [Co][kh][kw][Ci] weight of the convolution it is the transpose of);
 *  - `stream` is a hipStream_t passed as void*; every call only enqueues work on it;
 *  - the library never allocates, frees or synchronises; scratch memory is passed in by
 *    the caller and sized by the matching *_workspace_bytes query;
 *  - return value 0 = success, negative = error, text via vg_last_error() (thread local).
 */
#ifndef VEGANS_HIP_H
#define VEGANS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VG_OK 0
#define VG_ERR_INVALID (-1)
#define VG_ERR_LAUNCH (-2)
#define VG_ERR_WORKSPACE (-3)

/* activation codes shared by the fused norm/activation kernels */
#define VG_ACT_NONE 0
#define VG_ACT_RELU 1
#define VG_ACT_LRELU 2
#define VG_ACT_TANH 3
#define VG_ACT_SIGMOID 4

const char* vg_last_error(void);
int vg_abi_version(void);

/* Geometry of a 2-d cross-correlation y = conv(x, w): x [N][H][W][Ci], w [Co][kh][kw][Ci],
 * y [N][Ho][Wo][Co], Ho = (H + 2*ph - kh)/sh + 1 (dilation 1, groups 1). */
typedef struct vg_conv_geom {
  int32_t N, H, W, Ci;
  int32_t Co, kh, kw;
  int32_t Ho, Wo;
  int32_t sh, sw, ph, pw;
} vg_conv_geom;

/* ---- dense contractions (MFMA implicit GEMM, exact f32) -------------------------------
 * Replace torch.nn.Conv2d / ConvTranspose2d / Linear forward and their autograd formulas as
 * invoked by `self.generator(z)` / `self.adversariat(x)`
 * (vegans/models/unconditional/AbstractGenerativeModel.py:615,671; vegans/utils/networks.py:45-47)
 * and by `loss.backward` / `torch.autograd.grad`
 * (AbstractGenerativeModel.py:367-372; WassersteinGANGP.py:72-79).
 *
 *  vg_conv2d_fwd   y  = conv(x, w) (+ bias[Co], nullable)
 *  vg_conv2d_dgrad gx = conv_transpose(gy, w): gradient of conv w.r.t. x; also the forward
 *                  of ConvTranspose2d (x plays gy) (+ bias[Ci], nullable)
 *  vg_conv2d_wgrad gw = sum over pixels of gy (x) x-window: gradient w.r.t. w; needs a
 *                  workspace of vg_conv2d_wgrad_workspace_bytes(g) bytes (split-K slabs,
 *                  reduced in a fixed order: results are run-to-run deterministic)
 * A Linear layer is the geometry H=W=kh=kw=1.
 */
int vg_conv2d_fwd(const vg_conv_geom* g, const float* x, const float* w, const float* bias,
                  float* y, void* stream);
int vg_conv2d_dgrad(const vg_conv_geom* g, const float* gy, const float* w, const float* bias,
                    float* gx, void* stream);
size_t vg_conv2d_wgrad_workspace_bytes(const vg_conv_geom* g);
int vg_conv2d_wgrad(const vg_conv_geom* g, const float* gy, const float* x, float* gw,
                    void* workspace, size_t workspace_bytes, void* stream);
/* y = act(conv(x, w) + bias) with act = VG_ACT_RELU / VG_ACT_LRELU applied as the result is stored, for the
 * convolutions vg_conv2d_fwd_act_applies reports (today: image stems of at most 4 input channels, as the first layer
 * of a critic without BatchNorm, `Conv2d -> LeakyReLU`); the pre-activation tensor is never written. */
int vg_conv2d_fwd_act_applies(const vg_conv_geom* g, int act);
int vg_conv2d_fwd_act(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act,
                      float slope, void* stream);
/* The two phases of vg_conv2d_wgrad as separate calls (contraction into split-K slabs, then the ordered slab sum), so
 * that a caller can time or overlap them individually; same arguments, same results. */
int vg_conv2d_wgrad_partial(const vg_conv_geom* g, const float* gy, const float* x, float* gw,
                            void* workspace, size_t workspace_bytes, void* stream);
int vg_conv2d_wgrad_reduce(const vg_conv_geom* g, float* gw, const void* workspace, void* stream);
/* The weight gradient of one layer summed over `pairs` (1..4) pairs (gy[i], x[i]) of one geometry in one launch: what
 * the reference's optimizer finds in weight.grad after every pass of a training step that ran through the layer has
 * back-propagated (WassersteinGANGP.py:88-107: fake batch, real batch, and the penalty's double backward twice;
 * torch's AccumulateGrad adds them one by one). gy and x are host arrays of device pointers. The splits of the launch
 * are dealt out over the pairs, so the slab traffic and the slab sum are those of a single pass; the slabs are added
 * in pair order, split order within a pair. vg_conv2d_wgrad_multi_applies: f32 tensors below 2 GiB, not the image-stem
 * geometries (those have a kernel of their own). */
int vg_conv2d_wgrad_multi_applies(const vg_conv_geom* g);
size_t vg_conv2d_wgrad_multi_workspace_bytes(const vg_conv_geom* g, int pairs);
int vg_conv2d_wgrad_partial_multi(const vg_conv_geom* g, int pairs, const float* const* gy, const float* const* x, float* gw,
                                  void* workspace, size_t workspace_bytes, void* stream);
int vg_conv2d_wgrad_reduce_multi(const vg_conv_geom* g, int pairs, float* gw, const void* workspace, void* stream);

/* vg_conv2d_wgrad for callers that cannot allocate device memory or hand over a workspace (code running under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc). It has no counterpart in the reference. Same
 * arguments, kernels and bit-identical results as vg_conv2d_wgrad; the split-K slabs go to a 32 MiB device array built
 * into the library (one copy per GPU, held until the process ends). A geometry whose slabs exceed that
 * (vg_conv2d_wgrad_workspace_bytes(g) > 32 MiB) is refused with VG_ERR_WORKSPACE. The call allocates and frees
 * nothing. All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (different streams) race;
 * calls on one stream, or otherwise ordered, are safe, and nothing needs clearing between them. The first call on a
 * device looks up the array's device address (hipGetSymbolAddress): make it outside a stream capture. */
int vg_conv2d_wgrad_noalloc(const vg_conv_geom* g, const float* gy, const float* x, float* gw, void* stream);

/* Measurement aid, no counterpart in the reference: the next vg_conv2d_* call made by this thread launches its
 * contraction kernel (not its helper kernels: operand conversion, slab sum) with these two hipEvent_t, created by the
 * caller with timing enabled; after the stream has drained, hipEventElapsedTime(start, stop) is that one kernel's own
 * execution time on the device. One-shot. vg_time_pending() tells whether a pair set earlier is still waiting for a
 * kernel (1) or was used (0), and vg_time_next_conv(NULL, NULL) withdraws it. Not for use under stream capture. */
int vg_time_next_conv(void* start_event, void* stop_event);
int vg_time_pending(void);

/* The same three contractions with a choice of multiplier precision (the north-star configuration names a bf16 mode
 * next to fp32): VG_PREC_F32 = exact f32 products (what the un-suffixed entry points do); VG_PREC_BF16 = operands
 * rounded to bf16 (nearest even) on their way into the matrix cores, f32 accumulation, f32 tensors in memory;
 * VG_PREC_F32_SPLIT = f32-grade products on the bf16 matrix cores: every operand is split exactly into three bf16
 * terms and the six leading cross products are accumulated in f32 (the neglected ones lie below 2^-25 of |a b|, less
 * than the rounding of one f32 multiplication), so results agree with VG_PREC_F32 to f32 rounding noise.
 * Shapes the bf16 tiles do not cover (operands that are not 16-byte vectors, e.g. a 3-channel image stem, and the
 * bandwidth-bound thin kernels) run in f32 under every setting.
 * Workspace: vg_conv2d_prec_workspace_bytes(mode, g, precision) bytes, 256-byte aligned (0 = none needed). The split
 * and bf16 modes keep the bf16 terms of both operands there (written by a converting kernel ahead of the contraction,
 * unless the caller hands them over; VG_BF16_PLANES=0 in the environment makes the bf16 mode convert inside the
 * contraction's tile loader instead); for
 * mode 2 the size includes the split-K slabs of vg_conv2d_wgrad_workspace_bytes, which come first, so
 * vg_conv2d_wgrad_reduce takes the same pointer. */
enum { VG_PREC_F32 = 0, VG_PREC_BF16 = 1, VG_PREC_F32_SPLIT = 2 };
size_t vg_conv2d_prec_workspace_bytes(int mode, const vg_conv_geom* g, int precision);   /* mode 0 fwd, 1 dgrad, 2 wgrad */
/* The `*_planes` arguments (nullable) hand over an operand that the caller has already split with vg_split_planes_f32,
 * e.g. an activation that feeds both a forward contraction and, later, a weight gradient; they are only read for
 * (mode, geometry, precision) combinations vg_conv2d_split_applies reports. */
int vg_conv2d_fwd_prec(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                       int precision, void* workspace, size_t workspace_bytes, const void* x_planes,
                       const void* w_planes, void* stream);
int vg_conv2d_dgrad_prec(const vg_conv_geom* g, const float* gy, const float* w, const float* bias, float* gx,
                         int precision, void* workspace, size_t workspace_bytes, const void* gy_planes,
                         const void* w_planes, void* stream);
int vg_conv2d_wgrad_partial_prec(const vg_conv_geom* g, const float* gy, const float* x, float* gw, void* workspace,
                                 size_t workspace_bytes, int precision, const void* gy_planes, const void* x_planes,
                                 void* stream);
/* The bf16 terms of `elements` f32 values under `precision` (three in the split mode, the rounded value alone in the
 * bf16 mode), written one plane after the other into `planes` (vg_split_planes_bytes bytes, 16-byte aligned); and
 * whether (mode, geometry) runs on such pre-converted operands under `precision`. */
size_t vg_split_planes_bytes(long long elements, int precision);
int vg_split_planes_f32(const float* in, long long elements, int precision, void* planes, void* stream);
int vg_conv2d_split_applies(int mode, const vg_conv_geom* g, int precision);

/* Name of the kernel template instance the call for (mode, geometry) dispatches to (mode 0 fwd, 1 dgrad, 2 wgrad;
 * 16-byte aligned tensors assumed), as a kernel trace prints it.  No counterpart in the reference: it exists so a
 * benchmark can attribute event-timed durations to the kernel the profiler lists. */
int vg_conv2d_variant(int mode, const vg_conv_geom* g, char* buf, size_t len);

/* ---- max pooling -----------------------------------------------------------------------
 * nn.MaxPool2d(kernel, stride, padding) as the reference's image-to-image critic uses it
 * (examples/example_image_to_image.py:78-90), over NHWC. The geometry struct is the convolution one with
 * Co == Ci == channels; Ho/Wo = floor((in + 2p - k)/s) + 1 (no ceil_mode, no dilation). `idx` receives, per
 * output element, the winning input pixel hi*W + wi (first maximum in (kh, kw) scan order; NaN wins).
 * vg_maxpool2d_bwd: gx = gradient with respect to x given gy and idx (gather form, deterministic).
 * vg_maxpool2d_pick: out[n][ho][wo][c] = t[n][idx][c], the derivative of vg_maxpool2d_bwd with respect to gy. */
int vg_maxpool2d_fwd(const vg_conv_geom* g, const float* x, float* y, int* idx, void* stream);
int vg_maxpool2d_bwd(const vg_conv_geom* g, const float* gy, const int* idx, float* gx, void* stream);
int vg_maxpool2d_pick(const vg_conv_geom* g, const float* t, const int* idx, float* out, void* stream);

/* ---- layout ---------------------------------------------------------------------------
 * NCHW <-> NHWC copies at the API boundary (user arrays are NCHW: AbstractGenerativeModel.py:319,
 * :672-674) and the label broadcast-concatenate of vegans/utils/utils.py:52-90. */
int vg_nchw_to_nhwc(const float* src, float* dst, int N, int C, int HW, void* stream);
int vg_nhwc_to_nchw(const float* src, float* dst, int N, int C, int HW, void* stream);
/* The same boundary for float64 arrays (numpy's default dtype; the reference keeps such a batch as float64 up to
 * `X_batch.float()`, AbstractGAN1v1.py:77, AbstractConditionalGenerativeModel.py:269): uploaded as they are and
 * rounded to f32 (round to nearest even, what `.float()` does) on the device, images while being re-laid to NHWC. */
int vg_nchw_to_nhwc_f64(const double* src, float* dst, int N, int C, int HW, void* stream);
int vg_cast_f64_to_f32(const double* src, float* dst, size_t n, void* stream);
/* out[n][p][0:C1] = a[n][p][0:C1]  (a_bcast=0)  or a[n][0:C1] tiled over the HW pixels (a_bcast=1);
 * out[n][p][C1:C1+C2] likewise from b.  NHWC; bit-exact copies. */
int vg_concat_channels(const float* a, int C1, int a_bcast, const float* b, int C2, int b_bcast,
                       float* out, int N, int HW, void* stream);
/* dst[p][0:Cp] = src[p][c0:c0+Cp] over `pixels` rows of Ctot channels: the gradient of vg_concat_channels with
 * respect to a non-broadcast part (labels and latent vectors receive no gradient on this path). */
int vg_slice_channels(const float* src, int Ctot, int c0, int Cp, float* dst, size_t pixels, void* stream);
/* out[c] = sum_m x[m][c]  (bias gradients); rows reduced in a fixed order. workspace: vg_colsum_workspace_bytes */
size_t vg_colsum_workspace_bytes(int M, int C);
int vg_colsum(const float* x, float* out, int M, int C, void* workspace, size_t workspace_bytes, void* stream);
/* out[m][c] = v[c]  (backward of vg_colsum) */
int vg_bcast_rows(const float* v, float* out, int M, int C, void* stream);
/* out = a + b ; out = a * s (plain elementwise helpers used by gradient accumulation) */
int vg_add(const float* a, const float* b, float* out, size_t n, void* stream);
int vg_scale(const float* a, float s, float* out, size_t n, void* stream);
int vg_fill(float* out, float v, size_t n, void* stream);
/* out[0] = c0*a[0] + c1*b[0] + c2*c[0] (b, c nullable) and, if ratio_b_over_a is non-null, b[0]/a[0]: the scalar
 * epilogue `0.5*(fake+real) + lmbda_grad*GP` and `RealFakeRatio` of AbstractGAN1v1.py:85-91 /
 * WassersteinGANGP.py:100-107 without a host round trip. */
int vg_scalar_combine(const float* a, float c0, const float* b, float c1, const float* c, float c2, float* out,
                      float* ratio_b_over_a, void* stream);

/* ---- BatchNorm (training and eval mode) + activations ----------------------------------
 * Replace torch.nn.BatchNorm1d/2d, LeakyReLU, ReLU, Tanh, Sigmoid inside the user networks
 * (examples/example_input_formats.py:52-77,141-165; notebooks/02_mnist.ipynb:219-227),
 * including the first and second derivative formulas autograd applies for WassersteinGANGP.py:72-82.
 * x is [M][C] (M = N*H*W pixels). */
size_t vg_bn_workspace_bytes(int M, int C);
/* batch statistics: mean[c], invstd[c] = 1/sqrt(biased var + eps); if running_mean/var are non-null they
 * are updated in place: r = (1-momentum)*r + momentum*stat (unbiased variance), as torch does; if
 * num_batches_tracked (the module's int64 counter, on the device) is non-null it is incremented by one. */
int vg_bn_stats(const float* x, int M, int C, float eps, float momentum, float* mean, float* invstd,
                float* running_mean, float* running_var, long long* num_batches_tracked, void* workspace,
                size_t workspace_bytes, void* stream);
/* Batch statistics without a pass over the activation: a convolution (mode 0: vg_conv2d_fwd_bnstats) or transposed
 * convolution (mode 1: vg_conv2d_dgrad_bnstats) that feeds a BatchNorm leaves, per row tile of its output and per
 * channel, a shift, the sum of (y - shift), the sum of (y - shift)^2 and the tile's row count in `stats`
 * (vg_conv2d_bnstats_workspace_bytes(mode, g) bytes, 16-byte aligned); vg_bn_stats_from_conv merges the tiles (in
 * double, fixed order) into what vg_bn_stats computes: mean, invstd, running statistics, batch counter. The convolution
 * entry points compute exactly what vg_conv2d_fwd / vg_conv2d_dgrad compute; *stats_written is 1 when the geometry ran
 * on a kernel that leaves the statistics (the eight-wave f32 MFMA tiles) and 0 when it did not (thin, stem and small
 * shapes: the caller then uses vg_bn_stats). Reference: nn.BatchNorm2d behind nn.Conv2d / nn.ConvTranspose2d in the
 * user networks (examples/example_input_formats.py:52-77). */
size_t vg_conv2d_bnstats_workspace_bytes(int mode, const vg_conv_geom* g);
int vg_conv2d_fwd_bnstats(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                          void* stats, size_t stats_bytes, int* stats_written, void* stream);
int vg_conv2d_dgrad_bnstats(const vg_conv_geom* g, const float* gy, const float* w, const float* bias, float* gx,
                            void* stats, size_t stats_bytes, int* stats_written, void* stream);
int vg_bn_stats_from_conv(int mode, const vg_conv_geom* g, const void* stats, float eps, float momentum,
                          float* mean, float* invstd, float* running_mean, float* running_var,
                          long long* num_batches_tracked, void* stream);
/* invstd[c] = 1/sqrt(var[c] + eps) (eval mode uses the running statistics) */
int vg_bn_invstd_from_var(const float* var, float* invstd, int C, float eps, void* stream);
/* y = act((x - mean) * invstd * gamma + beta); gamma/beta nullable (affine=False) */
int vg_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma,
                const float* beta, float* y, int M, int C, int act, float slope, void* stream);
/* training-mode backward: sums[0:C] = sum gy, sums[C:2C] = sum gy*xhat (=> gbeta, ggamma), then
 * gx = gamma*invstd*(gy - mean(gy) - xhat*mean(gy*xhat)) */
/* When a ReLU / LeakyReLU is fused behind the norm (vg_bn_apply with act), pass its code and slope: the incoming
 * gradient is then multiplied by the activation's slope mask on the fly. The mask is read off the fused OUTPUT when
 * act_out is given; with act_out null it is rebuilt from x as the sign of (x - mean) * (invstd * gamma) + beta, computed
 * by the very instruction sequence vg_bn_apply uses (gamma, beta: the norm's affine parameters, nullable as there), so
 * the backward pass reads one activation-sized tensor less. act = VG_ACT_NONE: no activation. */
int vg_bn_bwd_reduce(const float* gy, const float* x, const float* mean, const float* invstd, int M, int C,
                     float* sums, void* workspace, size_t workspace_bytes, const float* act_out, int act, float slope,
                     const float* gamma, const float* beta, void* stream);
int vg_bn_bwd_apply(const float* gy, const float* x, const float* mean, const float* invstd,
                    const float* gamma, const float* sums, float* gx, int M, int C, const float* act_out, int act,
                    float slope, const float* beta, void* stream);
/* vg_conv2d_dgrad followed by the backward of a ReLU / LeakyReLU that sits in FRONT of the convolution (layer order
 * conv_a -> act -> conv_b; this is conv_b's input gradient): gx = act'(act_out) * conv_transpose(gy, w), the mask
 * applied in the kernel's epilogue, so the gradient behind the activation is never written. act_out is the activation's
 * output (conv_b's input), laid out like gx. *masked is 1 when the geometry ran on a kernel with this epilogue (the
 * eight-wave f32 MFMA tiles); 0 means gx holds the plain input gradient and the caller applies vg_act_bwd itself.
 * Reference: autograd through nn.LeakyReLU between two nn.Conv2d (examples/example_input_formats.py:64-77). */
int vg_conv2d_dgrad_actmask(const vg_conv_geom* g, const float* gy, const float* w, float* gx, const float* act_out,
                            int act, float slope, int* masked, void* stream);
/* The two sums of vg_bn_bwd_reduce without a pass over (gy, x): when the gradient that arrives at a BatchNorm's output
 * is itself the output of a convolution kernel (the input gradient of the Conv2d behind the norm: vg_conv2d_dgrad_bnbwd;
 * the input gradient of a ConvTranspose2d behind it, which is a forward convolution: vg_conv2d_fwd_bnbwd), that kernel's
 * epilogue reads the norm's input bn_x at the positions it writes ([rows][channels], laid out like the convolution's
 * output), rebuilds the slope mask of the activation fused behind the norm (act, slope; bn_gamma / bn_beta nullable as
 * in vg_bn_apply) and leaves per row tile and channel sum (g m) and sum (g m xhat) in `sums`
 * (vg_conv2d_bnstats_workspace_bytes(mode, g) bytes, 16-byte aligned); vg_bn_bwd_sums_from_conv adds the tiles (double,
 * fixed order) into the f32 [2][C] that vg_bn_bwd_apply takes. The convolution's own result is exactly that of
 * vg_conv2d_fwd / vg_conv2d_dgrad; *sums_written is 0 when the geometry ran on a kernel without this epilogue (the
 * caller then uses vg_bn_bwd_reduce). Reference: the backward pass autograd runs through nn.BatchNorm2d + ReLU /
 * LeakyReLU in front of nn.Conv2d / nn.ConvTranspose2d (examples/example_input_formats.py:52-77). */
int vg_conv2d_fwd_bnbwd(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                        const float* bn_x, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                        const float* bn_beta, int act, float slope, void* sums, size_t sums_bytes, int* sums_written,
                        void* stream);
int vg_conv2d_dgrad_bnbwd(const vg_conv_geom* g, const float* gy, const float* w, const float* bias, float* gx,
                          const float* bn_x, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                          const float* bn_beta, int act, float slope, void* sums, size_t sums_bytes, int* sums_written,
                          void* stream);
int vg_bn_bwd_sums_from_conv(int mode, const vg_conv_geom* g, const void* tile_sums, float* sums, void* stream);
/* eval-mode backward: gx = gy * gamma * invstd */
int vg_bn_bwd_eval(const float* gy, const float* invstd, const float* gamma, float* gx, int M, int C,
                   void* stream);
/* second derivative of the training-mode backward. Inputs: gy, x, the incoming gradients
 * u = d/d(gx) [M][C] (nullable), v = d/d(ggamma) [C] (nullable), w = d/d(gbeta) [C] (nullable).
 * sums5[0:5C] = sum gy, sum gy*xhat, sum u, sum u*xhat, sum u*gy.
 * Outputs: d_gy [M][C], d_x [M][C], d_gamma [C]. */
int vg_bn_dbwd_reduce(const float* gy, const float* x, const float* u, const float* mean, const float* invstd,
                      int M, int C, float* sums5, void* workspace, size_t workspace_bytes, const float* act_out, int act,
                      float slope, const float* gamma, const float* beta, void* stream);
int vg_bn_dbwd_apply(const float* gy, const float* x, const float* u, const float* v, const float* w,
                     const float* mean, const float* invstd, const float* gamma, const float* sums5,
                     float* d_gy, float* d_x, float* d_gamma, int M, int C, const float* act_out, int act, float slope,
                     const float* beta, void* stream);

int vg_act_fwd(const float* x, float* y, size_t n, int act, float slope, void* stream);
/* gx = gy * act'(.) ; `ref` is the forward INPUT for RELU/LRELU and the forward OUTPUT for TANH/SIGMOID */
int vg_act_bwd(const float* gy, const float* ref, float* gx, size_t n, int act, float slope, void* stream);
/* derivative of vg_act_bwd w.r.t. `ref` for TANH/SIGMOID: d_ref = gg * gy * act''-factor(ref) */
int vg_act_dbwd_ref(const float* gg, const float* gy, const float* ref, float* d_ref, size_t n, int act,
                    void* stream);

/* ---- losses -----------------------------------------------------------------------------
 * Each forward writes one scalar out[0]; each backward writes dpred = gout[0] * dloss/dpred.
 *  mean : sign * mean(pred)                utils.py:45-49 (wasserstein_loss; sign -1 for target ones, +1 zeros)
 *  bce  : mean(-(t*max(log p,-100) + (1-t)*max(log(1-p),-100)))   VanillaGAN.py:55-56
 *  mse  : mean((p - t)^2)                                         LSGAN.py:52-53
 *  l1   : mean(|a - b|)                     LRGAN.py:73-74,103-105; ConditionalPix2Pix.py:57-58,66-68
 * workspace: vg_reduce_workspace_bytes(n). */
size_t vg_reduce_workspace_bytes(size_t n);
int vg_loss_mean_fwd(const float* pred, size_t n, float sign, float* out, void* ws, size_t ws_bytes, void* stream);
int vg_loss_mean_bwd(const float* gout, size_t n, float sign, float* dpred, void* stream);
/* `target` is a tensor of n targets, or null for the constant target `tconst` (ones_like / zeros_like call sites) */
int vg_loss_bce_fwd(const float* pred, const float* target, float tconst, size_t n, float* out, void* ws, size_t ws_bytes,
                    void* stream);
int vg_loss_bce_bwd(const float* gout, const float* pred, const float* target, float tconst, size_t n, float* dpred,
                    void* stream);
int vg_loss_mse_fwd(const float* pred, const float* target, float tconst, size_t n, float* out, void* ws, size_t ws_bytes,
                    void* stream);
int vg_loss_mse_bwd(const float* gout, const float* pred, const float* target, float tconst, size_t n, float* dpred,
                    void* stream);
int vg_loss_l1_fwd(const float* a, const float* b, size_t n, float* out, void* ws, size_t ws_bytes, void* stream);
/* da and db are each nullable (only one side may need a gradient) */
int vg_loss_l1_bwd(const float* gout, const float* a, const float* b, size_t n, float* da, float* db, void* stream);

/* ---- WGAN-GP penalty ----------------------------------------------------------------------
 * WassersteinGANGP.py:67-82 / ConditionalWassersteinGANGP.py:68-83.
 *  interp  : out[b][i] = alpha[b]*real[b][i] + (1-alpha[b])*fake[b][i]   (B rows of D elements)
 *  rownorm : norms[b] = sqrt(sum_i g[b][i]^2)
 *  penalty : out[0] = mean_b (norms[b]-1)^2
 *  penalty_bwd : dg[b][i] = gout[0] * 2*(norms[b]-1)/(B*norms[b]) * g[b][i]
 */
int vg_gp_interp(const float* alpha, const float* real, const float* fake, float* out, int B, size_t D,
                 void* stream);
int vg_gp_rownorm(const float* g, float* norms, int B, size_t D, void* stream);
int vg_gp_penalty_fwd(const float* norms, int B, float* out, void* stream);
int vg_gp_penalty_bwd(const float* gout, const float* g, const float* norms, float* dg, int B, size_t D,
                      void* stream);

/* ---- optimizers --------------------------------------------------------------------------
 * torch.optim.RMSprop / Adam single-tensor arithmetic as configured by
 * AbstractGenerativeModel.py:91-117 and stepped at :374-378, with the WGAN weight clamp of
 * WassersteinGAN.py:74-81 fused in (clamp_lo > clamp_hi disables it).  grad_scale multiplies the
 * gradient first (1/world_size for data parallel sums).  State tensors are updated in place. */
typedef struct vg_rmsprop_cfg {
  float lr, alpha, eps, weight_decay, momentum;
  int32_t centered;
  float grad_scale, clamp_lo, clamp_hi;
} vg_rmsprop_cfg;
int vg_rmsprop_step(float* param, const float* grad, float* square_avg, float* momentum_buf, float* grad_avg,
                    size_t n, const vg_rmsprop_cfg* cfg, void* stream);
/* The same update for `count` tensors in one launch (host arrays of device pointers and element counts; the
 * momentum / grad_avg arrays may be null when the configuration does not use them). */
int vg_rmsprop_step_multi(int count, float* const* params, const float* const* grads, float* const* square_avgs,
                          float* const* momentum_bufs, float* const* grad_avgs, const size_t* n,
                          const vg_rmsprop_cfg* cfg, void* stream);
typedef struct vg_adam_cfg {
  float lr, beta1, beta2, eps, weight_decay;
  float step_size;              /* lr / (1 - beta1^step), computed by the host in double like torch does */
  float bias_correction2_sqrt;  /* sqrt(1 - beta2^step) */
  int32_t amsgrad;
  int32_t decoupled_wd;         /* AdamW: p *= 1 - lr*wd before the update */
  float grad_scale, clamp_lo, clamp_hi;
  const float* step_scalars;    /* null, or device {step_size, bias_correction2_sqrt} overriding the two fields above:
                                   lets a captured HIP graph be replayed for later steps */
} vg_adam_cfg;
int vg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                 size_t n, const vg_adam_cfg* cfg, void* stream);
/* The same update for `count` tensors that share one configuration (hence one step count) in one launch. */
int vg_adam_step_multi(int count, float* const* params, const float* const* grads, float* const* exp_avgs,
                       float* const* exp_avg_sqs, float* const* max_exp_avg_sqs, const size_t* n,
                       const vg_adam_cfg* cfg, void* stream);
/* Sum of the gradient contributions a parameter receives within one backward pass, replacing the `AccumulateGrad`
 * additions torch performs for `loss.backward(retain_graph=True)` (AbstractGenerativeModel.py:367-372) when a network is
 * applied to several batches in one sub-step (WassersteinGANGP.py:88-107: fake, real, interpolates, and the penalty's
 * double backward). For every tensor t: outs[t] = scales[t] * (((src 0 + src 1) + src 2) + ...), the nsrc[t] (1..8)
 * sources added in the order given, so the caller fixes the floating-point result by fixing the order. `srcs` holds
 * count*8 device pointers (tensor t's sources at srcs[8*t ..]); outs[t] may alias one of its sources; host arrays. */
int vg_grad_sum_multi(int count, float* const* outs, const float* const* srcs, const int* nsrc, const size_t* n,
                      const float* scales, void* stream);
typedef struct vg_sgd_cfg {
  float lr, momentum, dampening, weight_decay;
  int32_t nesterov;
  int32_t first_step;           /* momentum buffer is initialised to the gradient on the first step */
  float grad_scale, clamp_lo, clamp_hi;
} vg_sgd_cfg;
int vg_sgd_step(float* param, const float* grad, float* momentum_buf, size_t n, const vg_sgd_cfg* cfg, void* stream);
/* p = min(max(p, lo), hi), NaN preserved (torch clamp_) */
int vg_clamp(float* param, size_t n, float lo, float hi, void* stream);

/* ---- activations stored as bf16 ("_h" entry points) ---------------------------------------------------------------
 * The bf16 mode named beside fp32 in the reference-facing configuration (torch autocast / bf16 tensors where the
 * reference would call nn.Conv2d / nn.BatchNorm2d / nn.LeakyReLU on them; the reference itself, utils/networks.py:19-154,
 * only ever runs f32) as a DATA PATH: feature maps and their gradients live in HBM as bf16 (round to nearest even on
 * every store), parameters, their gradients, optimizer state, batch statistics, per-channel sums and losses stay f32,
 * and all arithmetic between a load and a store is f32. bf16 tensors cross this interface as void*; their memory
 * order is that of the f32 entry point of the same name (NHWC activations, [Co][kh][kw][Ci] weights). Weights are
 * handed over as a bf16 copy of the f32 master (vg_cast_f32_to_bf16, once per parameter update).
 * vg_conv2d_h_applies(mode, g) tells whether contraction `mode` (0 fwd, 1 dgrad, 2 wgrad) of geometry g runs on bf16
 * tensors (channel counts in multiples of 8, more than 32 GEMM columns, not one of the thin / stem / 1x1 special
 * kernels); other shapes go through the f32 entry points on converted copies. No workspace except the split-K slabs
 * of the weight gradient (vg_conv2d_wgrad_h_workspace_bytes, fewer and longer splits than the f32 plan;
 * vg_conv2d_wgrad_reduce_h adds them in split order). Every tensor handed to these entry points is addressed through
 * 31-bit byte offsets: an operand or a bf16 result of 2 GiB or more is refused with VG_ERR_INVALID. */
int vg_conv2d_h_applies(int mode, const vg_conv_geom* g);
int vg_conv2d_fwd_h(const vg_conv_geom* g, const void* x, const void* w, const float* bias, void* y, void* stream);
int vg_conv2d_dgrad_h(const vg_conv_geom* g, const void* gy, const void* w, const float* bias, void* gx, void* stream);
/* vg_conv2d_dgrad_h with the slope mask of a ReLU / LeakyReLU in FRONT of the convolution applied before the store
 * (as vg_conv2d_dgrad_actmask does for f32): gx = act'(act_out) * conv_transpose(gy, w); act_out is that activation's
 * output, bf16, laid out like gx. Served wherever vg_conv2d_h_applies(1, g). */
int vg_conv2d_dgrad_h_actmask(const vg_conv_geom* g, const void* gy, const void* w, void* gx, const void* act_out,
                              int act, float slope, void* stream);
size_t vg_conv2d_wgrad_h_workspace_bytes(const vg_conv_geom* g);
int vg_conv2d_wgrad_partial_h(const vg_conv_geom* g, const void* gy, const void* x, float* gw, void* workspace,
                              size_t workspace_bytes, void* stream);
int vg_conv2d_wgrad_reduce_h(const vg_conv_geom* g, float* gw, const void* workspace, void* stream);
/* The weight gradient of one layer summed over `pairs` (1..4) pairs (gy[i], x[i]) of one geometry, in one launch: what
 * the reference's optimizer sees in weight.grad after the passes of a training step that ran through the layer have
 * all back-propagated (WassersteinGANGP.py:88-107: the critic sees the fake batch, the real batch and the interpolates,
 * and torch's AccumulateGrad adds the contributions one by one). gy and x are host arrays of device pointers; the
 * slabs of all pairs lie in one workspace (vg_conv2d_wgrad_h_multi_workspace_bytes) and vg_conv2d_wgrad_reduce_h_multi
 * adds them in pair order, split order within a pair. */
size_t vg_conv2d_wgrad_h_multi_workspace_bytes(const vg_conv_geom* g, int pairs);
int vg_conv2d_wgrad_partial_h_multi(const vg_conv_geom* g, int pairs, const void* const* gy, const void* const* x, float* gw,
                                    void* workspace, size_t workspace_bytes, void* stream);
int vg_conv2d_wgrad_reduce_h_multi(const vg_conv_geom* g, int pairs, float* gw, const void* workspace, void* stream);
/* The layers next to an image in such a network (reference: the first nn.Conv2d of a critic and the last
 * nn.ConvTranspose2d of a generator, e.g. examples/example_input_formats.py:64-77) have an f32 image on one side and a
 * bf16 feature map on the other: forward f32 x -> bf16 y with an optional ReLU / LeakyReLU, input gradient bf16 gy ->
 * f32 gx, weight gradient from bf16 gy and f32 x into the split-K slabs of vg_conv2d_wgrad_workspace_bytes (added by
 * vg_conv2d_wgrad_reduce). vg_conv2d_stem_h_applies(mode, g): whether the stem / scatter kernels serve (mode, g). */
int vg_conv2d_stem_h_applies(int mode, const vg_conv_geom* g);
int vg_conv2d_stem_fwd_h(const vg_conv_geom* g, const float* x, const float* w, const float* bias, void* y, int act,
                         float slope, void* stream);
int vg_conv2d_stem_dgrad_h(const vg_conv_geom* g, const void* gy, const float* w, const float* bias, float* gx,
                           void* stream);
int vg_conv2d_stem_wgrad_partial_h(const vg_conv_geom* g, const void* gy, const float* x, void* workspace,
                                   size_t workspace_bytes, void* stream);
int vg_cast_f32_to_bf16(const float* in, void* out, size_t n, void* stream);
int vg_cast_bf16_to_f32(const void* in, float* out, size_t n, void* stream);
int vg_add_h(const void* a, const void* b, void* out, size_t n, void* stream);
int vg_fill_h(void* out, float v, size_t n, void* stream);
int vg_colsum_h(const void* x, float* out, int M, int C, void* workspace, size_t workspace_bytes, void* stream);
/* BatchNorm and activation kernels on bf16 activations: arguments as in the f32 entry points of the same name */
int vg_bn_stats_h(const void* x, int M, int C, float eps, float momentum, float* mean, float* invstd,
                  float* running_mean, float* running_var, long long* num_batches_tracked, void* workspace,
                  size_t workspace_bytes, void* stream);
int vg_bn_apply_h(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta, void* y,
                  int M, int C, int act, float slope, void* stream);
int vg_bn_bwd_reduce_h(const void* gy, const void* x, const float* mean, const float* invstd, int M, int C, float* sums,
                       void* workspace, size_t workspace_bytes, const void* act_out, int act, float slope,
                       const float* gamma, const float* beta, void* stream);
int vg_bn_bwd_apply_h(const void* gy, const void* x, const float* mean, const float* invstd, const float* gamma,
                      const float* sums, void* gx, int M, int C, const void* act_out, int act, float slope,
                      const float* beta, void* stream);
int vg_bn_bwd_eval_h(const void* gy, const float* invstd, const float* gamma, void* gx, int M, int C, void* stream);
int vg_bn_dbwd_reduce_h(const void* gy, const void* x, const void* u, const float* mean, const float* invstd, int M,
                        int C, float* sums5, void* workspace, size_t workspace_bytes, const void* act_out, int act,
                        float slope, const float* gamma, const float* beta, void* stream);
int vg_bn_dbwd_apply_h(const void* gy, const void* x, const void* u, const float* v, const float* w, const float* mean,
                       const float* invstd, const float* gamma, const float* sums5, void* d_gy, void* d_x,
                       float* d_gamma, int M, int C, const void* act_out, int act, float slope, const float* beta,
                       void* stream);
int vg_act_fwd_h(const void* x, void* y, size_t n, int act, float slope, void* stream);
int vg_act_bwd_h(const void* gy, const void* ref, void* gx, size_t n, int act, float slope, void* stream);
int vg_act_dbwd_ref_h(const void* gg, const void* gy, const void* ref, void* d_ref, size_t n, int act, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VEGANS_HIP_H */
