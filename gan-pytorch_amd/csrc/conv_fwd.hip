// vg_conv2d_fwd: y = conv(x, w) + bias on the f32 MFMA implicit-GEMM core.
#include "pgemm.h"

int vg_check_geom(const vg_conv_geom* g, const char* who) {
  if (!g) return vg_fail(VG_ERR_INVALID, "%s: null geometry", who);
  if (g->N <= 0 || g->H <= 0 || g->W <= 0 || g->Ci <= 0 || g->Co <= 0 || g->kh <= 0 || g->kw <= 0 ||
      g->sh <= 0 || g->sw <= 0 || g->ph < 0 || g->pw < 0 || g->Ho <= 0 || g->Wo <= 0)
    return vg_fail(VG_ERR_INVALID, "%s: non-positive extent in geometry", who);
  if ((g->H + 2 * g->ph - g->kh) / g->sh + 1 != g->Ho || (g->W + 2 * g->pw - g->kw) / g->sw + 1 != g->Wo ||
      g->H + 2 * g->ph < g->kh || g->W + 2 * g->pw < g->kw)
    return vg_fail(VG_ERR_INVALID, "%s: Ho/Wo (%d,%d) inconsistent with H,W,k,s,p", who, g->Ho, g->Wo);
  const long long in_elems = (long long)g->N * g->H * g->W * g->Ci;
  const long long out_elems = (long long)g->N * g->Ho * g->Wo * g->Co;
  const long long w_elems = (long long)g->Co * g->kh * g->kw * g->Ci;
  if (in_elems >= (1LL << 31) || out_elems >= (1LL << 31) || w_elems >= (1LL << 31))
    return vg_fail(VG_ERR_INVALID, "%s: tensor exceeds 2^31 elements", who);
  return 0;
}

int vg_try_fwd_thin(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, hipStream_t st);
int vg_thin_dgrad_applies(const vg_conv_geom* g);
int vg_stem_fwd_applies(const vg_conv_geom* g);
int vg_stem_wgrad_applies(const vg_conv_geom* g);

static int check_precision(int precision, const char* who) {
  if (precision < VG_PREC_F32 || precision > VG_PREC_F32_SPLIT) return vg_fail(VG_ERR_INVALID, "%s: unknown precision %d", who, precision);
  return 0;
}

extern "C" int vg_conv2d_fwd(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                             void* stream) {
  return vg_conv2d_fwd_prec(g, x, w, bias, y, VG_PREC_F32, nullptr, 0, nullptr, nullptr, stream);
}

// Row tiles of the forward implicit GEMM for this geometry (the plan vg_conv2d_fwd runs in f32).
static void fwd_tiles(const vg_conv_geom* g, int* bm, int* bn, int* mtiles, int* ntiles) {
  const int M = g->N * g->Ho * g->Wo;
  *bn = igemm_pick_bn(g->Co, VG_PREC_F32);
  *ntiles = (g->Co + *bn - 1) / *bn;
  *bm = igemm_pick_bm(M, *ntiles, 1, *bn, VG_PREC_F32);
  *mtiles = (M + *bm - 1) / *bm;
}

static int conv_fwd_impl(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int precision,
                         void* workspace, size_t workspace_bytes, const void* x_planes, const void* w_planes,
                         double* stats, int* stats_written, const VgNormBwdSrc* nb, void* stream, bool io_h = false);

extern "C" int vg_conv2d_fwd_prec(const vg_conv_geom* g, const float* x, const float* w, const float* bias,
                                  float* y, int precision, void* workspace, size_t workspace_bytes, const void* x_planes,
                                  const void* w_planes, void* stream) {
  return conv_fwd_impl(g, x, w, bias, y, precision, workspace, workspace_bytes, x_planes, w_planes, nullptr, nullptr, nullptr, stream);
}

extern "C" size_t vg_conv2d_bnstats_workspace_bytes(int mode, const vg_conv_geom* g) {
  if (vg_check_geom(g, "vg_conv2d_bnstats_workspace_bytes") || mode < 0 || mode > 1) return 0;
  // an upper bound that does not depend on the tile the plan picks: 64-row tiles
  long long rows, classes = 1, cols;
  if (mode == 0) {
    rows = (long long)g->N * g->Ho * g->Wo;
    cols = g->Co;
  } else {
    rows = (long long)g->N * ((g->H + g->sh - 1) / g->sh) * ((g->W + g->sw - 1) / g->sw);
    classes = (long long)g->sh * g->sw;
    cols = g->Ci;
  }
  const long long tiles = ((rows + 63) / 64) * classes;
  return (size_t)(tiles * (3 * cols + 1)) * sizeof(double);
}

extern "C" int vg_conv2d_fwd_bnstats(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                                     void* stats, size_t stats_bytes, int* stats_written, void* stream) {
  VG_REQUIRE(stats && stats_written && vg_aligned16(stats), "vg_conv2d_fwd_bnstats: null or misaligned statistics buffer");
  if (stats_bytes < vg_conv2d_bnstats_workspace_bytes(0, g))
    return vg_fail(VG_ERR_WORKSPACE, "vg_conv2d_fwd_bnstats: statistics buffer of %zu bytes required, got %zu",
                   vg_conv2d_bnstats_workspace_bytes(0, g), stats_bytes);
  return conv_fwd_impl(g, x, w, bias, y, VG_PREC_F32, nullptr, 0, nullptr, nullptr, (double*)stats, stats_written, nullptr, stream);
}

int vg_check_norm_bwd_src(const VgNormBwdSrc* nb, const void* stats, size_t stats_bytes, size_t need, const int* written,
                          const char* who) {
  VG_REQUIRE(stats && written && vg_aligned16(stats), "%s: null or misaligned buffer for the per-tile sums", who);
  VG_REQUIRE(nb->x && nb->mean && nb->invstd, "%s: the norm's input and batch statistics are required", who);
  VG_REQUIRE(nb->act == VG_ACT_NONE || nb->act == VG_ACT_RELU || nb->act == VG_ACT_LRELU,
             "%s: only ReLU / LeakyReLU can sit between the norm and the convolution", who);
  if (stats_bytes < need)
    return vg_fail(VG_ERR_WORKSPACE, "%s: buffer of %zu bytes required for the per-tile sums, got %zu", who, need, stats_bytes);
  return VG_OK;
}

extern "C" int vg_conv2d_fwd_bnbwd(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                                   const float* bn_x, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                   const float* bn_beta, int act, float slope, void* sums, size_t sums_bytes,
                                   int* sums_written, void* stream) {
  const VgNormBwdSrc nb{bn_x, bn_mean, bn_invstd, bn_gamma, bn_beta, act, slope};
  if (int e = vg_check_norm_bwd_src(&nb, sums, sums_bytes, vg_conv2d_bnstats_workspace_bytes(0, g), sums_written,
                                    "vg_conv2d_fwd_bnbwd"))
    return e;
  return conv_fwd_impl(g, x, w, bias, y, VG_PREC_F32, nullptr, 0, nullptr, nullptr, (double*)sums, sums_written, &nb, stream);
}

static int conv_fwd_impl(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int precision,
                         void* workspace, size_t workspace_bytes, const void* x_planes, const void* w_planes,
                         double* stats, int* stats_written, const VgNormBwdSrc* nb, void* stream, bool io_h) {
  if (stats_written) *stats_written = 0;
  if (int e = vg_check_geom(g, "vg_conv2d_fwd")) return e;
  if (int e = check_precision(precision, "vg_conv2d_fwd")) return e;
  VG_REQUIRE(x && w && y, "vg_conv2d_fwd: null tensor pointer");
  if (!io_h) {
    if (int r = vg_try_fwd_thin(g, x, w, bias, y, (hipStream_t)stream)) return r < 0 ? r : VG_OK;
  }
  if (precision == VG_PREC_F32_SPLIT && !vg_planes_applicable(0, g)) precision = VG_PREC_F32;
  const bool planes = io_h || pgemm_serves(precision, 0, g);
  IgemmParams p{};
  p.A = x; p.B = w; p.bias = bias; p.out = y;
  p.N = g->N; p.H = g->H; p.W = g->W; p.Ci = g->Ci; p.Co = g->Co; p.KH = g->kh; p.KW = g->kw;
  p.Ho = g->Ho; p.Wo = g->Wo; p.sh = g->sh; p.sw = g->sw; p.ph = g->ph; p.pw = g->pw;
  p.M = g->N * g->Ho * g->Wo;
  p.fast_div = p.M < (1 << 22) ? 1 : 0;
  p.Ncols = g->Co;
  p.Ktot = g->kh * g->kw * g->Ci;
  p.bias_mod = g->Co;
  const int bn = igemm_pick_bn(p.Ncols, precision);
  p.ntiles = (p.Ncols + bn - 1) / bn;
  const int bm = igemm_pick_bm(p.M, p.ntiles, 1, bn, planes ? PREC_SPLIT : precision);
  p.mtiles = (p.M + bm - 1) / bm;
  const int va = (g->Ci % 4 == 0 && vg_aligned16(x)) ? 4 : 1;
  const int vb = (p.Ktot % 4 == 0 && vg_aligned16(w)) ? 4 : 1;
  dim3 grid((unsigned)(p.mtiles * p.ntiles), 1, 1);
  if (planes) {
    const int np = vg_planes_of(precision);
    PgemmParams pp{};
    pp.g = p;
    if (int e = pgemm_prepare(pp, x, (long long)g->N * g->H * g->W * g->Ci, w, (long long)g->Co * p.Ktot, np, workspace,
                              workspace_bytes, (hipStream_t)stream, "vg_conv2d_fwd", x_planes, w_planes))
      return e;
    if (io_h) {
      if (int e = pgemm_set_out_h(pp, (long long)p.M * p.Ncols, "vg_conv2d_fwd")) return e;
    }
    if ((io_h ? pgemm_launch<MODE_FWD, 1, true>(pp, bm, bn, grid, (hipStream_t)stream)
              : np == 3 ? pgemm_launch<MODE_FWD, 3>(pp, bm, bn, grid, (hipStream_t)stream)
                        : pgemm_launch<MODE_FWD, 1>(pp, bm, bn, grid, (hipStream_t)stream)) != 0)
      return vg_fail(VG_ERR_INVALID, "vg_conv2d_fwd: no plane kernel for tile %dx%d", bm, bn);
    VG_CHECK_LAUNCH("vg_conv2d_fwd(planes)");
    return VG_OK;
  }
  if (igemm_set_operands(p, (long long)g->N * g->H * g->W * g->Ci, (long long)g->Co * p.Ktot))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_fwd: the MFMA kernels address operands with 31-bit byte offsets; x or w is 2 GiB or larger");
  if (stats) {   // with column statistics where such a kernel exists for the plan, else plainly
    p.stats = stats;
    p.stats_tiles = p.mtiles;
    if (nb) igemm_set_norm_bwd(p, *nb);
    if (igemm_launch<MODE_FWD, false>(p, bm, bn, va, vb, grid, (hipStream_t)stream, precision) == 0) {
      VG_CHECK_LAUNCH("vg_conv2d_fwd(statistics)");
      *stats_written = 1;
      return VG_OK;
    }
    p.stats = nullptr;
    p.bx = nullptr;
  }
  if (igemm_launch<MODE_FWD, false>(p, bm, bn, va, vb, grid, (hipStream_t)stream, precision) != 0)
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_fwd: no kernel variant for bn=%d va=%d vb=%d", bn, va, vb);
  VG_CHECK_LAUNCH("vg_conv2d_fwd");
  return VG_OK;
}

// ---- activations stored as bf16 --------------------------------------------------------------------------------------
// Which contractions run entirely on bf16 tensors (operands read as they lie in memory, result written as bf16; the
// weight gradient leaves as f32): the shapes the plane GEMM serves, minus those the library routes to special kernels
// (thin layers, image stems, the 1x1 latent GEMM), which a caller runs through the f32 entry points on converted copies.
bool vg_wgrad_narrow_tile(const vg_conv_geom* g);   // conv_wgrad.hip: the 32-column plan the plane GEMM has no tile for

extern "C" int vg_conv2d_h_applies(int mode, const vg_conv_geom* g) {
  if (vg_check_geom(g, "vg_conv2d_h_applies") || mode < 0 || mode > 2) return 0;
  if (!vg_planes_applicable(mode, g)) return 0;          // channel counts in multiples of 8, more than 32 columns
  if (g->Co <= 4 || g->Ci <= 4 || vg_stem_fwd_applies(g) || vg_thin_dgrad_applies(g)) return 0;
  if (mode == 1 && g->Ho == 1 && g->Wo == 1 && g->ph == 0 && g->pw == 0 && g->H == g->kh && g->W == g->kw) return 0;
  if (mode == 2 && vg_wgrad_narrow_tile(g)) return 0;
  return 1;
}

extern "C" int vg_conv2d_fwd_h(const vg_conv_geom* g, const void* x, const void* w, const float* bias, void* y,
                               void* stream) {
  if (!vg_conv2d_h_applies(0, g))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_fwd_h: this geometry is not served on bf16 tensors (vg_conv2d_h_applies)");
  VG_REQUIRE(x && w && y && vg_aligned16(x) && vg_aligned16(w) && vg_aligned16(y), "vg_conv2d_fwd_h: null or misaligned tensor");
  return conv_fwd_impl(g, (const float*)x, (const float*)w, bias, (float*)y, VG_PREC_BF16, nullptr, 0, x, w, nullptr, nullptr,
                       nullptr, stream, true);
}

// Tiles and rows of the statistics vg_conv2d_fwd_bnstats / vg_conv2d_dgrad_bnstats left for this geometry.
void vg_bnstats_layout(int mode, const vg_conv_geom* g, int* tiles, long long* rows, int* cols) {
  int bm, bn, mtiles, ntiles;
  if (mode == 0) {
    fwd_tiles(g, &bm, &bn, &mtiles, &ntiles);
    *tiles = mtiles;
    *rows = (long long)g->N * g->Ho * g->Wo;
    *cols = g->Co;
  } else {
    const int Hc = (g->H + g->sh - 1) / g->sh, Wc = (g->W + g->sw - 1) / g->sw;
    const int M = g->N * Hc * Wc;
    bn = igemm_pick_bn(g->Ci, VG_PREC_F32);
    ntiles = (g->Ci + bn - 1) / bn;
    bm = igemm_pick_bm(M, ntiles, g->sh * g->sw, bn, VG_PREC_F32);
    *tiles = ((M + bm - 1) / bm) * g->sh * g->sw;
    *rows = (long long)g->N * g->H * g->W;
    *cols = g->Ci;
  }
}

int vg_try_fwd_stem_act(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act,
                        float slope, hipStream_t st);

extern "C" int vg_conv2d_fwd_act_applies(const vg_conv_geom* g, int act) {
  if (vg_check_geom(g, "vg_conv2d_fwd_act_applies")) return 0;
  return (act == VG_ACT_RELU || act == VG_ACT_LRELU) && vg_stem_fwd_applies(g) ? 1 : 0;
}

extern "C" int vg_conv2d_fwd_act(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y,
                                 int act, float slope, void* stream) {
  if (int e = vg_check_geom(g, "vg_conv2d_fwd_act")) return e;
  VG_REQUIRE(x && w && y, "vg_conv2d_fwd_act: null tensor pointer");
  if (!vg_conv2d_fwd_act_applies(g, act))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_fwd_act: no fused kernel for this geometry / activation");
  const int r = vg_try_fwd_stem_act(g, x, w, bias, y, act, slope, (hipStream_t)stream);
  return r < 0 ? r : VG_OK;
}

// Plain GEMM out[M][Ncols] = A[M][K] * B[K][Ncols] (+ bias[col % bias_mod]); used by vg_conv2d_dgrad when the
// gradient image is 1x1 (ConvTranspose2d applied to a latent vector): every tap then hits exactly one output
// pixel and the transposed convolution is a single GEMM against the weight viewed as [Co][kh*kw*Ci].
int vg_gemm_kn(const float* A, const float* B, const float* bias, int bias_mod, float* out, int M, int Ncols, int K,
               hipStream_t stream, int precision) {
  if (precision == VG_PREC_F32_SPLIT) precision = VG_PREC_F32;   // small and rare: ConvTranspose2d of a latent vector
  IgemmParams p{};
  p.A = A; p.B = B; p.bias = bias; p.out = out;
  p.N = M; p.H = 1; p.W = 1; p.Ci = K; p.Co = Ncols; p.KH = 1; p.KW = 1;
  p.Ho = 1; p.Wo = 1; p.sh = 1; p.sw = 1; p.ph = 0; p.pw = 0;
  p.M = M; p.Ncols = Ncols; p.Ktot = K; p.bias_mod = bias_mod;
  p.fast_div = p.M < (1 << 22) ? 1 : 0;
  const int bn = igemm_pick_bn(Ncols, precision);
  p.ntiles = (Ncols + bn - 1) / bn;
  const int bm = igemm_pick_bm(M, p.ntiles, 1, bn, precision);
  p.mtiles = (M + bm - 1) / bm;
  const int va = (K % 4 == 0 && vg_aligned16(A)) ? 4 : 1;
  const int vb = (Ncols % 4 == 0 && vg_aligned16(B)) ? 4 : 1;
  dim3 grid((unsigned)(p.mtiles * p.ntiles), 1, 1);
  if (igemm_set_operands(p, (long long)M * K, (long long)K * Ncols))
    return vg_fail(VG_ERR_INVALID, "vg_gemm_kn: the MFMA kernels address operands with 31-bit byte offsets; an operand is 2 GiB or larger");
  if (igemm_launch<MODE_FWD, true>(p, bm, bn, va, vb, grid, stream, precision) != 0)
    return vg_fail(VG_ERR_INVALID, "vg_gemm_kn: no kernel variant for bn=%d va=%d vb=%d", bn, va, vb);
  VG_CHECK_LAUNCH("vg_gemm_kn");
  return VG_OK;
}

// Name of the kernel variant a contraction dispatches to (mode 0 fwd, 1 dgrad, 2 wgrad), for profilers that want to
// attribute measured time to a kernel: "igemm<MODE,BM,BN,WAVES_M,B_MC,VA,VB>" as it appears in a kernel trace.
extern "C" int vg_conv2d_variant(int mode, const vg_conv_geom* g, char* buf, size_t len) {
  if (int e = vg_check_geom(g, "vg_conv2d_variant")) return e;
  VG_REQUIRE(buf && len > 0 && mode >= 0 && mode <= 2, "vg_conv2d_variant: bad arguments");
  int kmode = mode, ncols, va, vb, bmc, zdim = 1;
  long long rows = (long long)g->N * g->Ho * g->Wo;
  if (mode == 0 && g->Co <= 4 && g->Ci % 4 == 0) {
    snprintf(buf, len, "fwd_thin_kernel<%d, %d>", g->Co, g->kh * g->kw * g->Ci >= 2048 ? 256 : 64);
    return VG_OK;
  }
  if (mode == 0 && vg_stem_fwd_applies(g)) {
    snprintf(buf, len, "stem_fwd_kernel<%d, %d, %d>", g->Ci, g->kh, g->sh);
    return VG_OK;
  }
  if (mode == 2 && vg_stem_wgrad_applies(g) && rows >= 512) {     // (more than one split-K slab)
    snprintf(buf, len, "stem_wgrad_kernel<%d, %d, %d>", g->Ci, g->kh, g->sh);
    return VG_OK;
  }
  const bool gemm_dgrad = g->Ho == 1 && g->Wo == 1 && g->ph == 0 && g->pw == 0 && g->H == g->kh && g->W == g->kw;
  if (mode == 1 && !gemm_dgrad && vg_thin_dgrad_applies(g)) {
    const int taps = g->kh * g->kw;
    if (g->Co % 4 == 0 && (taps == 9 || taps == 16 || taps == 25))
      snprintf(buf, len, "dgrad_scatter_kernel<%d, %d>", g->Ci, taps);
    else
      snprintf(buf, len, "dgrad_thin_kernel<%d>", g->Ci);
    return VG_OK;
  }
  if (mode == 0) {
    ncols = g->Co; va = g->Ci % 4 == 0 ? 4 : 1; vb = (g->kh * g->kw * g->Ci) % 4 == 0 ? 4 : 1; bmc = 0;
  } else if (mode == 1) {
    if (g->Ho == 1 && g->Wo == 1 && g->ph == 0 && g->pw == 0 && g->H == g->kh && g->W == g->kw) {
      kmode = 0; ncols = g->kh * g->kw * g->Ci; va = g->Co % 4 == 0 ? 4 : 1; vb = ncols % 4 == 0 ? 4 : 1;
      rows = g->N;
    } else {
      ncols = g->Ci; va = g->Co % 4 == 0 ? 4 : 1; vb = g->Ci % 4 == 0 ? 4 : 1;
      zdim = g->sh * g->sw;
      rows = (long long)g->N * ((g->H + g->sh - 1) / g->sh) * ((g->W + g->sw - 1) / g->sw);
    }
    bmc = 1;
  } else {
    ncols = g->kh * g->kw * g->Ci; va = g->Co % 4 == 0 ? 4 : 1; vb = g->Ci % 4 == 0 ? 4 : 1; bmc = 1;
    rows = g->Co;
  }
  const int bn = igemm_pick_bn(ncols);
  const int bm = mode == 2 ? (rows <= 64 && bn != 32 ? 64 : 128) : igemm_pick_bm(rows, (ncols + bn - 1) / bn, zdim, bn);
  // trailing arguments: precision 0 = the f32 instance (what the un-suffixed entry points run), then waves per block
  const int nw = igemm_waves(bm, bn, va, vb, PREC_F32,
                             igemm_small24(g->N, g->H, g->W, g->Ci, g->Co, g->kh, g->kw, g->Ho, g->Wo));
  // (last argument: the variant without column statistics; a convolution asked to leave BatchNorm statistics runs the
  // instance that ends in "true")
  snprintf(buf, len, "igemm_kernel<%d, %d, %d, %d, %s, %d, %d, 0, %d, 0>", kmode, bm, bn, igemm_waves_m(bm, bn, nw),
           bmc ? "true" : "false", va, vb, nw);
  return VG_OK;
}
