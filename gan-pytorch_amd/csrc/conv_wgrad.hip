// vg_conv2d_wgrad: gw[co][kh][kw][ci] = sum over (n,ho,wo) of gy[n][ho][wo][co] * x-window.
// The contraction runs over all output pixels, so it is split along k across blockIdx.z; every split writes
// its own f32 slab and a second kernel adds the slabs in split order (bitwise reproducible, no atomics).
#include "pgemm.h"

#include <mutex>

int vg_try_wgrad_stem(const vg_conv_geom* g, const float* gy, const float* x, float* slabs, int nslabs, hipStream_t st);
int vg_stem_wgrad_applies(const vg_conv_geom* g);

namespace {

struct WgradPlan {
  int M, Ncols, Ktot, bm, bn, mtiles, ntiles, nsplit, kchunk;
};

// target_blocks: how many blocks the split along k aims at. Two per CU for the f32 kernels (matrix-core-bound: the
// second block hides the first one's stalls); one per CU for the contraction on bf16 tensors, which is bound by memory
// traffic and gains more from halving its slab writes (measured: 512 better in f32, 256 better on bf16 tensors).
// kalign: what a split's k extent is rounded up to (the k-step of the kernel that will run it).
WgradPlan wgrad_plan(const vg_conv_geom* g, int target_blocks = 512, int kalign = 32) {
  WgradPlan pl;
  pl.M = g->Co;
  pl.Ncols = g->kh * g->kw * g->Ci;
  pl.Ktot = g->N * g->Ho * g->Wo;
  pl.bn = igemm_pick_bn(pl.Ncols);
  pl.bm = (pl.M <= 64 && pl.bn != 32) ? 64 : 128;
  pl.mtiles = (pl.M + pl.bm - 1) / pl.bm;
  pl.ntiles = (pl.Ncols + pl.bn - 1) / pl.bn;
  const int tiles = pl.mtiles * pl.ntiles;
  // aim at two blocks per CU (what the double-buffered tiles allow to be resident; every extra split costs a slab
  // write + read), at least 8 k-steps (256 pixels) per split
  int want = (target_blocks + tiles / 2) / tiles;
  int max_split = pl.Ktot / 256;
  if (max_split < 1) max_split = 1;
  int ns = want < max_split ? want : max_split;
  if (ns < 1) ns = 1;
  if (ns > 1024) ns = 1024;
  int kchunk = (pl.Ktot + ns - 1) / ns;
  kchunk = ((kchunk + kalign - 1) / kalign) * kalign;
  pl.kchunk = kchunk;
  pl.nsplit = (pl.Ktot + kchunk - 1) / kchunk;
  return pl;
}

// out[i] = sum_z slabs[z][i].  A block owns 256 / LANES consecutive float4 columns; LANES threads per column each add
// every LANES-th slab, then the lane partials are added in lane order through LDS (fixed order: deterministic).
// Many slabs over a small tile (a stem layer's weight gradient: hundreds of splits of a 64 x 48 tile) would
// otherwise be a serial chain per element.
template <int LANES>
__global__ void __launch_bounds__(256) slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                          size_t count, int nsplit, int vec) {
  __shared__ f32x4 red[256];
  constexpr int COLS = 256 / LANES;
  const int lane = threadIdx.x / COLS, col = threadIdx.x % COLS;
  const size_t ncols = vec ? count >> 2 : count;
  const size_t i = (size_t)blockIdx.x * COLS + col;
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (i < ncols) {
    if (vec) {
      // eight loads in flight, added in slab order (the sum's order is what makes the result reproducible; issued
      // one at a time the loop is a chain of memory latencies)
      const f32x4* s4 = reinterpret_cast<const f32x4*>(slabs) + i;
      int z = lane;
      for (; z + 7 * LANES < nsplit; z += 8 * LANES) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s4[(size_t)(z + u * LANES) * ncols];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
      }
      for (; z < nsplit; z += LANES) a += s4[(size_t)z * ncols];
    } else {
      const float* s1 = slabs + i;
      int z = lane;
      for (; z + 7 * LANES < nsplit; z += 8 * LANES) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s1[(size_t)(z + u * LANES) * count];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[0] += v[u];
      }
      for (; z < nsplit; z += LANES) a[0] += s1[(size_t)z * count];
    }
  }
  if constexpr (LANES > 1) {
    red[threadIdx.x] = a;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int l = 1; l < LANES; ++l) a += red[l * COLS + col];
    }
  }
  if (lane == 0 && i < ncols) {
    if (vec) reinterpret_cast<f32x4*>(out)[i] = a;
    else out[i] = a[0];
  }
}

int launch_slab_reduce(const float* slabs, float* out, size_t count, int nsplit, hipStream_t st) {
  const int vec = ((count & 3) == 0 && vg_aligned16(slabs) && vg_aligned16(out)) ? 1 : 0;
  const size_t ncols = vec ? count >> 2 : count;
  // enough lanes to keep ~2048+ threads busy on small tiles, one lane when the tile alone fills the chip
  int lanes = 1;
  while (lanes < 16 && lanes * 2 <= nsplit && ncols * lanes < 65536) lanes *= 2;
#define SLAB_CASE(L_)                                                                                              \
  if (lanes == L_) {                                                                                               \
    const size_t blocks = (ncols + (256 / L_) - 1) / (256 / L_);                                                   \
    hipLaunchKernelGGL((slab_reduce_kernel<L_>), dim3((unsigned)blocks), dim3(256), 0, st, slabs, out, count, nsplit, vec); \
  }
  SLAB_CASE(1) SLAB_CASE(2) SLAB_CASE(4) SLAB_CASE(8) SLAB_CASE(16)
#undef SLAB_CASE
  return 0;
}

}  // namespace

bool vg_wgrad_narrow_tile(const vg_conv_geom* g) { return wgrad_plan(g).bn == 32; }

extern "C" size_t vg_conv2d_wgrad_workspace_bytes(const vg_conv_geom* g) {
  if (vg_check_geom(g, "vg_conv2d_wgrad_workspace_bytes")) return 0;
  const WgradPlan pl = wgrad_plan(g);
  if (pl.nsplit <= 1) return 0;
  return (size_t)pl.nsplit * pl.M * pl.Ncols * sizeof(float);
}

static size_t slab_bytes(const WgradPlan& pl) {
  const size_t raw = pl.nsplit > 1 ? (size_t)pl.nsplit * pl.M * pl.Ncols * sizeof(float) : 0;
  return (raw + 255) / 256 * 256;
}

extern "C" size_t vg_conv2d_prec_workspace_bytes(int mode, const vg_conv_geom* g, int precision) {
  if (vg_check_geom(g, "vg_conv2d_prec_workspace_bytes") || mode < 0 || mode > 2) return 0;
  size_t bytes = mode == 2 ? slab_bytes(wgrad_plan(g)) : 0;
  if (pgemm_serves(precision, mode, g)) bytes += vg_planes_operand_bytes(mode, g, vg_planes_of(precision));
  return bytes;
}

// 1 when (mode, geometry) runs on pre-converted operands under `precision`, i.e. when operand planes are of use.
extern "C" int vg_conv2d_split_applies(int mode, const vg_conv_geom* g, int precision) {
  if (vg_check_geom(g, "vg_conv2d_split_applies") || mode < 0 || mode > 2) return 0;
  if (mode == 2 && wgrad_plan(g).bn == 32) return 0;
  return pgemm_serves(precision, mode, g) ? 1 : 0;
}

// Phase 1: the contraction. With more than one split the per-split slabs land in `workspace`, otherwise `gw` is final.
extern "C" int vg_conv2d_wgrad_partial(const vg_conv_geom* g, const float* gy, const float* x, float* gw, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  return vg_conv2d_wgrad_partial_prec(g, gy, x, gw, workspace, workspace_bytes, VG_PREC_F32, nullptr, nullptr, stream);
}

static int wgrad_partial_impl(const vg_conv_geom* g, const float* gy, const float* x, float* gw, void* workspace,
                              size_t workspace_bytes, int precision, const void* gy_planes, const void* x_planes,
                              void* stream, bool io_h);

extern "C" int vg_conv2d_wgrad_partial_prec(const vg_conv_geom* g, const float* gy, const float* x, float* gw,
                                            void* workspace, size_t workspace_bytes, int precision, const void* gy_planes,
                                            const void* x_planes, void* stream) {
  return wgrad_partial_impl(g, gy, x, gw, workspace, workspace_bytes, precision, gy_planes, x_planes, stream, false);
}

// gy and x stored as bf16, gw f32: the contraction reads the tensors as they lie in memory; split-K slabs (f32) in
// `workspace` (vg_conv2d_wgrad_h_workspace_bytes), added by vg_conv2d_wgrad_reduce_h.
extern "C" int vg_conv2d_wgrad_partial_h(const vg_conv_geom* g, const void* gy, const void* x, float* gw, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (!vg_conv2d_h_applies(2, g))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_wgrad_partial_h: this geometry is not served on bf16 tensors (vg_conv2d_h_applies)");
  VG_REQUIRE(gy && x && vg_aligned16(gy) && vg_aligned16(x), "vg_conv2d_wgrad_partial_h: null or misaligned tensor");
  return wgrad_partial_impl(g, (const float*)gy, (const float*)x, gw, workspace, workspace_bytes, VG_PREC_BF16, gy, x, stream, true);
}

constexpr int kTargetBlocksH = 256;
// the plan of the contraction on bf16 tensors: 64-deep k-steps (measured again with the register-lean uniform k-walk,
// which would fit two blocks per CU: 3.19-3.20 ms per iteration at 256 blocks, 3.27 at 384, 3.27-3.29 at 512)
static WgradPlan wgrad_plan_h(const vg_conv_geom* g) { return wgrad_plan(g, kTargetBlocksH, 64); }

extern "C" size_t vg_conv2d_wgrad_h_workspace_bytes(const vg_conv_geom* g) {
  if (vg_check_geom(g, "vg_conv2d_wgrad_h_workspace_bytes")) return 0;
  const WgradPlan pl = wgrad_plan_h(g);
  if (pl.nsplit <= 1) return 0;
  return (size_t)pl.nsplit * pl.M * pl.Ncols * sizeof(float);
}

extern "C" int vg_conv2d_wgrad_reduce_h(const vg_conv_geom* g, float* gw, const void* workspace, void* stream) {
  if (int e = vg_check_geom(g, "vg_conv2d_wgrad_reduce_h")) return e;
  const WgradPlan pl = wgrad_plan_h(g);
  if (pl.nsplit <= 1) return VG_OK;
  VG_REQUIRE(gw && workspace, "vg_conv2d_wgrad_reduce_h: null pointer");
  launch_slab_reduce((const float*)workspace, gw, (size_t)pl.M * pl.Ncols, pl.nsplit, (hipStream_t)stream);
  VG_CHECK_LAUNCH("vg_conv2d_wgrad_h(slab reduce)");
  return VG_OK;
}

// Several (gy, x) pairs of one geometry in one launch: the weight gradient of a layer that several passes of a
// training step ran through is the sum over the passes, and one launch over all of them has the slab traffic and the
// slab sum of a single pass (its splits are dealt out over the pairs). Up to four pairs.
constexpr int kMaxPairsH = 4;
static WgradPlan wgrad_plan_h_multi(const vg_conv_geom* g, int pairs) {
  return pairs <= 1 ? wgrad_plan_h(g) : wgrad_plan(g, (kTargetBlocksH + pairs - 1) / pairs, 64);
}

extern "C" size_t vg_conv2d_wgrad_h_multi_workspace_bytes(const vg_conv_geom* g, int pairs) {
  if (vg_check_geom(g, "vg_conv2d_wgrad_h_multi_workspace_bytes") || pairs < 1 || pairs > kMaxPairsH) return 0;
  const WgradPlan pl = wgrad_plan_h_multi(g, pairs);
  if (pl.nsplit * pairs <= 1) return 0;
  return (size_t)pl.nsplit * pairs * pl.M * pl.Ncols * sizeof(float);
}

extern "C" int vg_conv2d_wgrad_reduce_h_multi(const vg_conv_geom* g, int pairs, float* gw, const void* workspace, void* stream) {
  if (int e = vg_check_geom(g, "vg_conv2d_wgrad_reduce_h_multi")) return e;
  VG_REQUIRE(pairs >= 1 && pairs <= kMaxPairsH, "vg_conv2d_wgrad_reduce_h_multi: one to four pairs");
  const WgradPlan pl = wgrad_plan_h_multi(g, pairs);
  if (pl.nsplit * pairs <= 1) return VG_OK;
  VG_REQUIRE(gw && workspace, "vg_conv2d_wgrad_reduce_h_multi: null pointer");
  launch_slab_reduce((const float*)workspace, gw, (size_t)pl.M * pl.Ncols, pl.nsplit * pairs, (hipStream_t)stream);
  VG_CHECK_LAUNCH("vg_conv2d_wgrad_h_multi(slab reduce)");
  return VG_OK;
}

extern "C" int vg_conv2d_wgrad_partial_h_multi(const vg_conv_geom* g, int pairs, const void* const* gy, const void* const* x,
                                               float* gw, void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = vg_check_geom(g, "vg_conv2d_wgrad_partial_h_multi")) return e;
  if (!vg_conv2d_h_applies(2, g))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_wgrad_partial_h_multi: this geometry is not served on bf16 tensors (vg_conv2d_h_applies)");
  VG_REQUIRE(pairs >= 1 && pairs <= kMaxPairsH && gy && x && gw, "vg_conv2d_wgrad_partial_h_multi: one to four pairs, no null pointers");
  for (int i = 0; i < pairs; ++i)
    VG_REQUIRE(gy[i] && x[i] && vg_aligned16(gy[i]) && vg_aligned16(x[i]), "vg_conv2d_wgrad_partial_h_multi: null or misaligned tensor");
  const WgradPlan pl = wgrad_plan_h_multi(g, pairs);
  const int nsplit = pl.nsplit * pairs;
  const size_t need = nsplit > 1 ? (size_t)nsplit * pl.M * pl.Ncols * sizeof(float) : 0;
  if (need > 0 && (!workspace || workspace_bytes < need || !vg_aligned16(workspace)))
    return vg_fail(VG_ERR_WORKSPACE, "vg_conv2d_wgrad_partial_h_multi: workspace of %zu bytes (16-byte aligned) required, got %zu",
                   need, workspace_bytes);
  IgemmParams p{};
  p.out = nsplit > 1 ? (float*)workspace : gw;
  p.N = g->N; p.H = g->H; p.W = g->W; p.Ci = g->Ci; p.Co = g->Co; p.KH = g->kh; p.KW = g->kw;
  p.Ho = g->Ho; p.Wo = g->Wo; p.sh = g->sh; p.sw = g->sw; p.ph = g->ph; p.pw = g->pw;
  p.M = pl.M; p.Ncols = pl.Ncols; p.Ktot = pl.Ktot;
  p.mtiles = pl.mtiles; p.ntiles = pl.ntiles;
  p.kchunk = pl.kchunk;
  p.bias_mod = 1;
  p.slab_stride = nsplit > 1 ? (long long)pl.M * pl.Ncols : 0;
  PgemmParams pp{};
  pp.g = p;
  if (int e = pgemm_prepare(pp, nullptr, (long long)g->N * g->Ho * g->Wo * g->Co, nullptr, (long long)g->N * g->H * g->W * g->Ci, 1,
                            nullptr, 0, (hipStream_t)stream, "vg_conv2d_wgrad_partial_h_multi", gy[0], x[0]))
    return e;
  for (int i = 0; i < pairs; ++i) {
    pp.Aseg[i] = (const __bf16*)gy[i];
    pp.Bseg[i] = (const __bf16*)x[i];
  }
  pp.seg_splits = pl.nsplit;
  dim3 grid((unsigned)(pl.mtiles * pl.ntiles), 1, (unsigned)nsplit);
  if (pgemm_launch<MODE_WGRAD, 1>(pp, pl.bm, pl.bn, grid, (hipStream_t)stream) != 0)
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_wgrad_partial_h_multi: no plane kernel for tile %dx%d", pl.bm, pl.bn);
  VG_CHECK_LAUNCH("vg_conv2d_wgrad_partial_h_multi");
  return VG_OK;
}

// The same on f32 tensors (fp32 mode, the MFMA kernel; not the image stems, which have a kernel of their own).
static WgradPlan wgrad_plan_multi(const vg_conv_geom* g, int pairs) {
  return pairs <= 1 ? wgrad_plan(g) : wgrad_plan(g, (512 + pairs - 1) / pairs);
}

extern "C" int vg_conv2d_wgrad_multi_applies(const vg_conv_geom* g) {
  if (vg_check_geom(g, "vg_conv2d_wgrad_multi_applies")) return 0;
  if (vg_stem_wgrad_applies(g)) return 0;
  return (long long)g->N * g->Ho * g->Wo * g->Co * 4 < (1LL << 31) && (long long)g->N * g->H * g->W * g->Ci * 4 < (1LL << 31);
}

extern "C" size_t vg_conv2d_wgrad_multi_workspace_bytes(const vg_conv_geom* g, int pairs) {
  if (vg_check_geom(g, "vg_conv2d_wgrad_multi_workspace_bytes") || pairs < 1 || pairs > kMaxPairsH) return 0;
  const WgradPlan pl = wgrad_plan_multi(g, pairs);
  if (pl.nsplit * pairs <= 1) return 0;
  return (size_t)pl.nsplit * pairs * pl.M * pl.Ncols * sizeof(float);
}

extern "C" int vg_conv2d_wgrad_reduce_multi(const vg_conv_geom* g, int pairs, float* gw, const void* workspace, void* stream) {
  if (int e = vg_check_geom(g, "vg_conv2d_wgrad_reduce_multi")) return e;
  VG_REQUIRE(pairs >= 1 && pairs <= kMaxPairsH, "vg_conv2d_wgrad_reduce_multi: one to four pairs");
  const WgradPlan pl = wgrad_plan_multi(g, pairs);
  if (pl.nsplit * pairs <= 1) return VG_OK;
  VG_REQUIRE(gw && workspace, "vg_conv2d_wgrad_reduce_multi: null pointer");
  launch_slab_reduce((const float*)workspace, gw, (size_t)pl.M * pl.Ncols, pl.nsplit * pairs, (hipStream_t)stream);
  VG_CHECK_LAUNCH("vg_conv2d_wgrad_multi(slab reduce)");
  return VG_OK;
}

extern "C" int vg_conv2d_wgrad_partial_multi(const vg_conv_geom* g, int pairs, const float* const* gy, const float* const* x,
                                             float* gw, void* workspace, size_t workspace_bytes, void* stream) {
  if (!vg_conv2d_wgrad_multi_applies(g))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_wgrad_partial_multi: this geometry is not served (vg_conv2d_wgrad_multi_applies)");
  VG_REQUIRE(pairs >= 1 && pairs <= kMaxPairsH && gy && x && gw, "vg_conv2d_wgrad_partial_multi: one to four pairs, no null pointers");
  for (int i = 0; i < pairs; ++i) VG_REQUIRE(gy[i] && x[i], "vg_conv2d_wgrad_partial_multi: null tensor");
  const WgradPlan pl = wgrad_plan_multi(g, pairs);
  const int nsplit = pl.nsplit * pairs;
  const size_t need = nsplit > 1 ? (size_t)nsplit * pl.M * pl.Ncols * sizeof(float) : 0;
  if (need > 0 && (!workspace || workspace_bytes < need || !vg_aligned16(workspace)))
    return vg_fail(VG_ERR_WORKSPACE, "vg_conv2d_wgrad_partial_multi: workspace of %zu bytes (16-byte aligned) required, got %zu",
                   need, workspace_bytes);
  IgemmParams p{};
  p.A = gy[0]; p.B = x[0]; p.bias = nullptr;
  p.out = nsplit > 1 ? (float*)workspace : gw;
  p.N = g->N; p.H = g->H; p.W = g->W; p.Ci = g->Ci; p.Co = g->Co; p.KH = g->kh; p.KW = g->kw;
  p.Ho = g->Ho; p.Wo = g->Wo; p.sh = g->sh; p.sw = g->sw; p.ph = g->ph; p.pw = g->pw;
  p.M = pl.M; p.Ncols = pl.Ncols; p.Ktot = pl.Ktot;
  p.mtiles = pl.mtiles; p.ntiles = pl.ntiles;
  p.kchunk = pl.kchunk;
  p.bias_mod = 1;
  p.slab_stride = nsplit > 1 ? (long long)pl.M * pl.Ncols : 0;
  bool a16 = true, b16 = true;
  for (int i = 0; i < pairs; ++i) {
    p.Aseg[i] = gy[i];
    p.Bseg[i] = x[i];
    a16 = a16 && vg_aligned16(gy[i]);
    b16 = b16 && vg_aligned16(x[i]);
  }
  p.seg_splits = pl.nsplit;
  const int va = (g->Co % 4 == 0 && a16) ? 4 : 1;
  const int vb = (g->Ci % 4 == 0 && b16) ? 4 : 1;
  if (igemm_set_operands(p, (long long)g->N * g->Ho * g->Wo * g->Co, (long long)g->N * g->H * g->W * g->Ci))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_wgrad_partial_multi: gy or x is 2 GiB or larger");
  dim3 grid((unsigned)(pl.mtiles * pl.ntiles), 1, (unsigned)nsplit);
  if (igemm_launch<MODE_WGRAD, true>(p, pl.bm, pl.bn, va, vb, grid, (hipStream_t)stream, VG_PREC_F32) != 0)
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_wgrad_partial_multi: no kernel variant for bn=%d va=%d vb=%d", pl.bn, va, vb);
  VG_CHECK_LAUNCH("vg_conv2d_wgrad_partial_multi");
  return VG_OK;
}

static int wgrad_partial_impl(const vg_conv_geom* g, const float* gy, const float* x, float* gw, void* workspace,
                              size_t workspace_bytes, int precision, const void* gy_planes, const void* x_planes,
                              void* stream, bool io_h) {
  if (int e = vg_check_geom(g, "vg_conv2d_wgrad")) return e;
  if (precision < VG_PREC_F32 || precision > VG_PREC_F32_SPLIT)
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_wgrad: unknown precision %d", precision);
  VG_REQUIRE(gy && x && gw, "vg_conv2d_wgrad: null tensor pointer");
  const WgradPlan pl = io_h ? wgrad_plan_h(g) : wgrad_plan(g);
  const size_t need = pl.nsplit > 1 ? (size_t)pl.nsplit * pl.M * pl.Ncols * sizeof(float) : 0;
  if (need > 0 && (!workspace || workspace_bytes < need || !vg_aligned16(workspace)))
    return vg_fail(VG_ERR_WORKSPACE, "vg_conv2d_wgrad: workspace of %zu bytes (16-byte aligned) required, got %zu",
                   need, workspace_bytes);

  // image stems (3 or 1 input channels) have a kernel of their own that fills the same slabs
  if (pl.nsplit > 1 && !io_h) {
    if (int r = vg_try_wgrad_stem(g, gy, x, (float*)workspace, pl.nsplit, (hipStream_t)stream)) return r < 0 ? r : VG_OK;
  }

  IgemmParams p{};
  p.A = gy; p.B = x; p.bias = nullptr;
  p.out = pl.nsplit > 1 ? (float*)workspace : gw;
  p.N = g->N; p.H = g->H; p.W = g->W; p.Ci = g->Ci; p.Co = g->Co; p.KH = g->kh; p.KW = g->kw;
  p.Ho = g->Ho; p.Wo = g->Wo; p.sh = g->sh; p.sw = g->sw; p.ph = g->ph; p.pw = g->pw;
  p.M = pl.M; p.Ncols = pl.Ncols; p.Ktot = pl.Ktot;
  p.mtiles = pl.mtiles; p.ntiles = pl.ntiles;
  p.kchunk = pl.kchunk;
  p.bias_mod = 1;
  p.slab_stride = pl.nsplit > 1 ? (long long)pl.M * pl.Ncols : 0;
  const int va = (g->Co % 4 == 0 && vg_aligned16(gy)) ? 4 : 1;
  const int vb = (g->Ci % 4 == 0 && vg_aligned16(x)) ? 4 : 1;
  dim3 grid((unsigned)(pl.mtiles * pl.ntiles), 1, (unsigned)pl.nsplit);
  if (io_h || (pgemm_serves(precision, 2, g) && pl.bn != 32)) {
    const int np = vg_planes_of(precision);
    PgemmParams pp{};
    pp.g = p;
    const size_t skip = io_h ? 0 : slab_bytes(pl);     // (both operands handed over: nothing is split into the workspace)
    if (workspace_bytes < skip) return vg_fail(VG_ERR_WORKSPACE, "vg_conv2d_wgrad: workspace too small for split mode");
    if (int e = pgemm_prepare(pp, gy, (long long)g->N * g->Ho * g->Wo * g->Co, x, (long long)g->N * g->H * g->W * g->Ci, np,
                              (char*)workspace + skip, workspace_bytes - skip, (hipStream_t)stream, "vg_conv2d_wgrad", gy_planes,
                              x_planes))
      return e;
    if ((np == 3 ? pgemm_launch<MODE_WGRAD, 3>(pp, pl.bm, pl.bn, grid, (hipStream_t)stream)
                 : pgemm_launch<MODE_WGRAD, 1>(pp, pl.bm, pl.bn, grid, (hipStream_t)stream)) != 0)
      return vg_fail(VG_ERR_INVALID, "vg_conv2d_wgrad: no plane kernel for tile %dx%d", pl.bm, pl.bn);
    VG_CHECK_LAUNCH("vg_conv2d_wgrad(planes)");
    return VG_OK;
  }
  if (igemm_set_operands(p, (long long)g->N * g->Ho * g->Wo * g->Co, (long long)g->N * g->H * g->W * g->Ci))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_wgrad: the MFMA kernels address operands with 31-bit byte offsets; gy or x is 2 GiB or larger");
  if (igemm_launch<MODE_WGRAD, true>(p, pl.bm, pl.bn, va, vb, grid, (hipStream_t)stream, precision) != 0)
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_wgrad: no kernel variant for bn=%d va=%d vb=%d", pl.bn, va, vb);
  VG_CHECK_LAUNCH("vg_conv2d_wgrad");
  return VG_OK;
}

// Phase 2: add the slabs (no-op when the plan has a single split).
extern "C" int vg_conv2d_wgrad_reduce(const vg_conv_geom* g, float* gw, const void* workspace, void* stream) {
  if (int e = vg_check_geom(g, "vg_conv2d_wgrad_reduce")) return e;
  const WgradPlan pl = wgrad_plan(g);
  if (pl.nsplit <= 1) return VG_OK;
  VG_REQUIRE(gw && workspace, "vg_conv2d_wgrad_reduce: null pointer");
  launch_slab_reduce((const float*)workspace, gw, (size_t)pl.M * pl.Ncols, pl.nsplit, (hipStream_t)stream);
  VG_CHECK_LAUNCH("vg_conv2d_wgrad(slab reduce)");
  return VG_OK;
}

extern "C" int vg_conv2d_wgrad(const vg_conv_geom* g, const float* gy, const float* x, float* gw, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (int e = vg_conv2d_wgrad_partial(g, gy, x, gw, workspace, workspace_bytes, stream)) return e;
  return vg_conv2d_wgrad_reduce(g, gw, workspace, stream);
}

// ---- the weight gradient for callers that cannot allocate -----------------------------------------------------------
// Same kernels as vg_conv2d_wgrad; the split-K slabs live in a file-scope device array instead of a caller-provided
// workspace. Every GPU the library runs on holds its own copy of the array (at its own address) until the process
// ends. The array's name is only a host-side handle: the device address is looked up once per device with
// hipGetSymbolAddress and that pointer is what the kernels receive.
constexpr size_t kStaticScratchBytes = (size_t)32 << 20;      // 32 MiB: the largest slab set of a DCGAN-64 layer at batch 256
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {
constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices] = {nullptr};
std::mutex g_scratch_mutex;

// Device address of `scratch` on the current device (looked up on the first call per device, with that device current).
int static_scratch(void** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return vg_fail(VG_ERR_LAUNCH, "vg_conv2d_wgrad_noalloc: hipGetDevice: %s", hipGetErrorString(e));
  if (dev < 0 || dev >= kMaxDevices) return vg_fail(VG_ERR_INVALID, "vg_conv2d_wgrad_noalloc: device index %d out of range", dev);
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr)
      return vg_fail(VG_ERR_LAUNCH, "vg_conv2d_wgrad_noalloc: hipGetSymbolAddress(scratch) on device %d: %s", dev,
                     hipGetErrorString(e));
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return VG_OK;
}
}  // namespace

extern "C" int vg_conv2d_wgrad_noalloc(const vg_conv_geom* g, const float* gy, const float* x, float* gw, void* stream) {
  if (int e = vg_check_geom(g, "vg_conv2d_wgrad_noalloc")) return e;
  VG_REQUIRE(gy && x && gw, "vg_conv2d_wgrad_noalloc: null tensor pointer");
  const size_t need = vg_conv2d_wgrad_workspace_bytes(g);
  if (need > kStaticScratchBytes)
    return vg_fail(VG_ERR_WORKSPACE,
                   "vg_conv2d_wgrad_noalloc: this geometry needs %zu bytes of split-K slabs, the built-in scratch holds %zu; "
                   "use vg_conv2d_wgrad with a caller-provided workspace",
                   need, kStaticScratchBytes);
  void* ws = nullptr;
  if (int e = static_scratch(&ws)) return e;
  // every split writes its whole slab before the reduce reads it: nothing has to start at zero, whatever the previous
  // call left behind
  return vg_conv2d_wgrad(g, gy, x, gw, ws, kStaticScratchBytes, stream);
}
