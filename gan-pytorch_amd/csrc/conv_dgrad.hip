// vg_conv2d_dgrad: gx = conv_transpose(gy, w) (+ bias): the gradient of a convolution with respect to its
// input, which is also the forward pass of ConvTranspose2d.
#include "pgemm.h"

int vg_gemm_kn(const float* A, const float* B, const float* bias, int bias_mod, float* out, int M, int Ncols, int K,
               hipStream_t stream, int precision);
int vg_try_dgrad_thin(const vg_conv_geom* g, const float* gy, const float* w, const float* bias, float* gx,
                      hipStream_t st);

extern "C" int vg_conv2d_dgrad(const vg_conv_geom* g, const float* gy, const float* w, const float* bias,
                               float* gx, void* stream) {
  return vg_conv2d_dgrad_prec(g, gy, w, bias, gx, VG_PREC_F32, nullptr, 0, nullptr, nullptr, stream);
}

static int conv_dgrad_impl(const vg_conv_geom* g, const float* gy, const float* w, const float* bias, float* gx,
                           int precision, void* workspace, size_t workspace_bytes, const void* gy_planes,
                           const void* w_planes, double* stats, int* stats_written, const VgNormBwdSrc* nb, void* stream,
                           bool io_h = false);
int vg_check_norm_bwd_src(const VgNormBwdSrc* nb, const void* stats, size_t stats_bytes, size_t need, const int* written,
                          const char* who);

extern "C" int vg_conv2d_dgrad_prec(const vg_conv_geom* g, const float* gy, const float* w, const float* bias,
                                    float* gx, int precision, void* workspace, size_t workspace_bytes, const void* gy_planes,
                                    const void* w_planes, void* stream) {
  return conv_dgrad_impl(g, gy, w, bias, gx, precision, workspace, workspace_bytes, gy_planes, w_planes, nullptr, nullptr,
                         nullptr, stream);
}

extern "C" int vg_conv2d_dgrad_bnstats(const vg_conv_geom* g, const float* gy, const float* w, const float* bias, float* gx,
                                       void* stats, size_t stats_bytes, int* stats_written, void* stream) {
  VG_REQUIRE(stats && stats_written && vg_aligned16(stats), "vg_conv2d_dgrad_bnstats: null or misaligned statistics buffer");
  if (stats_bytes < vg_conv2d_bnstats_workspace_bytes(1, g))
    return vg_fail(VG_ERR_WORKSPACE, "vg_conv2d_dgrad_bnstats: statistics buffer of %zu bytes required, got %zu",
                   vg_conv2d_bnstats_workspace_bytes(1, g), stats_bytes);
  return conv_dgrad_impl(g, gy, w, bias, gx, VG_PREC_F32, nullptr, 0, nullptr, nullptr, (double*)stats, stats_written, nullptr, stream);
}

extern "C" int vg_conv2d_dgrad_bnbwd(const vg_conv_geom* g, const float* gy, const float* w, const float* bias, float* gx,
                                     const float* bn_x, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                     const float* bn_beta, int act, float slope, void* sums, size_t sums_bytes,
                                     int* sums_written, void* stream) {
  const VgNormBwdSrc nb{bn_x, bn_mean, bn_invstd, bn_gamma, bn_beta, act, slope};
  if (int e = vg_check_norm_bwd_src(&nb, sums, sums_bytes, vg_conv2d_bnstats_workspace_bytes(1, g), sums_written,
                                    "vg_conv2d_dgrad_bnbwd"))
    return e;
  return conv_dgrad_impl(g, gy, w, bias, gx, VG_PREC_F32, nullptr, 0, nullptr, nullptr, (double*)sums, sums_written, &nb, stream);
}

extern "C" int vg_conv2d_dgrad_actmask(const vg_conv_geom* g, const float* gy, const float* w, float* gx,
                                       const float* act_out, int act, float slope, int* masked, void* stream) {
  VG_REQUIRE(act_out && masked, "vg_conv2d_dgrad_actmask: null activation output or flag pointer");
  VG_REQUIRE(act == VG_ACT_RELU || act == VG_ACT_LRELU, "vg_conv2d_dgrad_actmask: only ReLU / LeakyReLU have a slope mask");
  const VgNormBwdSrc nb{act_out, nullptr, nullptr, nullptr, nullptr, act, slope};
  return conv_dgrad_impl(g, gy, w, nullptr, gx, VG_PREC_F32, nullptr, 0, nullptr, nullptr, nullptr, masked, &nb, stream);
}

extern "C" int vg_conv2d_dgrad_h(const vg_conv_geom* g, const void* gy, const void* w, const float* bias, void* gx,
                                 void* stream) {
  if (!vg_conv2d_h_applies(1, g))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_dgrad_h: this geometry is not served on bf16 tensors (vg_conv2d_h_applies)");
  VG_REQUIRE(gy && w && gx && vg_aligned16(gy) && vg_aligned16(w) && vg_aligned16(gx), "vg_conv2d_dgrad_h: null or misaligned tensor");
  return conv_dgrad_impl(g, (const float*)gy, (const float*)w, bias, (float*)gx, VG_PREC_BF16, nullptr, 0, gy, w, nullptr,
                         nullptr, nullptr, stream, true);
}

// vg_conv2d_dgrad_h followed by the backward of a ReLU / LeakyReLU that sits in FRONT of the convolution (its output
// act_out, bf16, laid out like gx): the mask is applied to the f32 result before it is rounded and stored.
extern "C" int vg_conv2d_dgrad_h_actmask(const vg_conv_geom* g, const void* gy, const void* w, void* gx, const void* act_out,
                                         int act, float slope, void* stream) {
  if (!vg_conv2d_h_applies(1, g))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_dgrad_h_actmask: this geometry is not served on bf16 tensors (vg_conv2d_h_applies)");
  VG_REQUIRE(gy && w && gx && act_out && vg_aligned16(gy) && vg_aligned16(w) && vg_aligned16(gx) && vg_aligned16(act_out),
             "vg_conv2d_dgrad_h_actmask: null or misaligned tensor");
  VG_REQUIRE(act == VG_ACT_RELU || act == VG_ACT_LRELU, "vg_conv2d_dgrad_h_actmask: only ReLU / LeakyReLU have a slope mask");
  const VgNormBwdSrc nb{(const float*)act_out, nullptr, nullptr, nullptr, nullptr, act, slope};
  return conv_dgrad_impl(g, (const float*)gy, (const float*)w, nullptr, (float*)gx, VG_PREC_BF16, nullptr, 0, gy, w, nullptr,
                         nullptr, &nb, stream, true);
}

static int conv_dgrad_impl(const vg_conv_geom* g, const float* gy, const float* w, const float* bias, float* gx,
                           int precision, void* workspace, size_t workspace_bytes, const void* gy_planes,
                           const void* w_planes, double* stats, int* stats_written, const VgNormBwdSrc* nb, void* stream,
                           bool io_h) {
  if (stats_written) *stats_written = 0;
  if (int e = vg_check_geom(g, "vg_conv2d_dgrad")) return e;
  if (precision < VG_PREC_F32 || precision > VG_PREC_F32_SPLIT)
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_dgrad: unknown precision %d", precision);
  VG_REQUIRE(gy && w && gx, "vg_conv2d_dgrad: null tensor pointer");

  // 1x1 gradient image covering the whole input window: one GEMM [N][Co] x [Co][kh*kw*Ci].
  if (!io_h && g->Ho == 1 && g->Wo == 1 && g->ph == 0 && g->pw == 0 && g->H == g->kh && g->W == g->kw)
    return vg_gemm_kn(gy, w, bias, g->Ci, gx, g->N, g->kh * g->kw * g->Ci, g->Co, (hipStream_t)stream, precision);

  if (!io_h) {
    if (int r = vg_try_dgrad_thin(g, gy, w, bias, gx, (hipStream_t)stream)) return r < 0 ? r : VG_OK;
  }

  if (precision == VG_PREC_F32_SPLIT && !vg_planes_applicable(1, g)) precision = VG_PREC_F32;
  const bool planes = io_h || pgemm_serves(precision, 1, g);
  IgemmParams p{};
  p.A = gy; p.B = w; p.bias = bias; p.out = gx;
  p.N = g->N; p.H = g->H; p.W = g->W; p.Ci = g->Ci; p.Co = g->Co; p.KH = g->kh; p.KW = g->kw;
  p.Ho = g->Ho; p.Wo = g->Wo; p.sh = g->sh; p.sw = g->sw; p.ph = g->ph; p.pw = g->pw;
  p.Ncols = g->Ci;
  p.bias_mod = g->Ci;
  // Largest parity class bounds the grid; smaller classes retire their surplus blocks at entry.
  const int Hc = (g->H + g->sh - 1) / g->sh, Wc = (g->W + g->sw - 1) / g->sw;
  p.M = g->N * Hc * Wc;
  p.fast_div = p.M < (1 << 22) ? 1 : 0;
  p.Ktot = 0;  // per class, derived in the kernel
  igemm_set_classes(p);
  const int bn = igemm_pick_bn(p.Ncols, precision);
  p.ntiles = (p.Ncols + bn - 1) / bn;
  const int bm = igemm_pick_bm(p.M, p.ntiles, g->sh * g->sw, bn, planes ? PREC_SPLIT : precision);
  p.mtiles = (p.M + bm - 1) / bm;
  const int va = (g->Co % 4 == 0 && vg_aligned16(gy)) ? 4 : 1;
  const int vb = (g->Ci % 4 == 0 && vg_aligned16(w)) ? 4 : 1;
  dim3 grid((unsigned)(p.mtiles * p.ntiles), 1, (unsigned)(g->sh * g->sw));
  if (planes) {
    const int np = vg_planes_of(precision);
    PgemmParams pp{};
    pp.g = p;
    if (int e = pgemm_prepare(pp, gy, (long long)g->N * g->Ho * g->Wo * g->Co, w, (long long)g->Co * g->kh * g->kw * g->Ci,
                              np, workspace, workspace_bytes, (hipStream_t)stream, "vg_conv2d_dgrad", gy_planes, w_planes))
      return e;
    if (io_h) {
      if (int e = pgemm_set_out_h(pp, (long long)g->N * g->H * g->W * g->Ci, "vg_conv2d_dgrad")) return e;
    }
    if (io_h && nb) {      // with the slope mask of the activation in front of the convolution (vg_conv2d_dgrad_h_actmask)
      pp.mask_ref = (const __bf16*)(const void*)nb->x;
      pp.mask_act = nb->act;
      pp.mask_slope = nb->slope;
      if (pgemm_launch<MODE_DGRAD, 1, true, true>(pp, bm, bn, grid, (hipStream_t)stream) != 0)
        return vg_fail(VG_ERR_INVALID, "vg_conv2d_dgrad: no plane kernel for tile %dx%d", bm, bn);
      VG_CHECK_LAUNCH("vg_conv2d_dgrad(planes, slope mask)");
      return VG_OK;
    }
    if ((io_h ? pgemm_launch<MODE_DGRAD, 1, true>(pp, bm, bn, grid, (hipStream_t)stream)
              : np == 3 ? pgemm_launch<MODE_DGRAD, 3>(pp, bm, bn, grid, (hipStream_t)stream)
                        : pgemm_launch<MODE_DGRAD, 1>(pp, bm, bn, grid, (hipStream_t)stream)) != 0)
      return vg_fail(VG_ERR_INVALID, "vg_conv2d_dgrad: no plane kernel for tile %dx%d", bm, bn);
    VG_CHECK_LAUNCH("vg_conv2d_dgrad(planes)");
    return VG_OK;
  }
  if (igemm_set_operands(p, (long long)g->N * g->Ho * g->Wo * g->Co, (long long)g->Co * g->kh * g->kw * g->Ci))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_dgrad: the MFMA kernels address operands with 31-bit byte offsets; gy or w is 2 GiB or larger");
  if (stats) {   // with column statistics where such a kernel exists for the plan, else plainly
    p.stats = stats;
    p.stats_tiles = p.mtiles * g->sh * g->sw;
    if (nb) igemm_set_norm_bwd(p, *nb);
    if (igemm_launch<MODE_DGRAD, true>(p, bm, bn, va, vb, grid, (hipStream_t)stream, precision) == 0) {
      VG_CHECK_LAUNCH("vg_conv2d_dgrad(statistics)");
      *stats_written = 1;
      return VG_OK;
    }
    p.stats = nullptr;
    p.bx = nullptr;
  } else if (nb) {   // no sums asked for: the slope mask alone, where the plan has such a kernel
    igemm_set_norm_bwd(p, *nb);
    if (igemm_launch<MODE_DGRAD, true>(p, bm, bn, va, vb, grid, (hipStream_t)stream, precision) == 0) {
      VG_CHECK_LAUNCH("vg_conv2d_dgrad(slope mask)");
      *stats_written = 1;
      return VG_OK;
    }
    p.bx = nullptr;
  }
  if (igemm_launch<MODE_DGRAD, true>(p, bm, bn, va, vb, grid, (hipStream_t)stream, precision) != 0)
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_dgrad: no kernel variant for bn=%d va=%d vb=%d", bn, va, vb);
  VG_CHECK_LAUNCH("vg_conv2d_dgrad");
  return VG_OK;
}
