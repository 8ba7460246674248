// vg_conv2d_wgrad: gw[co][kh][kw][ci] = sum over (n,ho,wo) of gy[n][ho][wo][co] * x-window.
// The contraction runs over all output pixels, so it is split along k across blockIdx.z; every split writes
// its own f32 slab and a second kernel adds the slabs in split order (bitwise reproducible, no atomics).
#include "igemm.h"

namespace {

struct WgradPlan {
  int M, Ncols, Ktot, bn, mtiles, ntiles, nsplit, kchunk;
};

WgradPlan wgrad_plan(const vg_conv_geom* g) {
  WgradPlan pl;
  pl.M = g->Co;
  pl.Ncols = g->kh * g->kw * g->Ci;
  pl.Ktot = g->N * g->Ho * g->Wo;
  pl.bn = igemm_pick_bn(pl.Ncols);
  pl.mtiles = (pl.M + 127) / 128;
  pl.ntiles = (pl.Ncols + pl.bn - 1) / pl.bn;
  const int tiles = pl.mtiles * pl.ntiles;
  // aim at ~3 blocks per CU, but keep at least 8 k-steps (256 pixels) per split
  int want = (768 + tiles - 1) / tiles;
  int max_split = pl.Ktot / 256;
  if (max_split < 1) max_split = 1;
  int ns = want < max_split ? want : max_split;
  if (ns < 1) ns = 1;
  if (ns > 1024) ns = 1024;
  int kchunk = (pl.Ktot + ns - 1) / ns;
  kchunk = ((kchunk + 31) / 32) * 32;
  pl.kchunk = kchunk;
  pl.nsplit = (pl.Ktot + kchunk - 1) / kchunk;
  return pl;
}

__global__ void __launch_bounds__(256) slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                          size_t count, int nsplit) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  if ((count & 3) == 0) {
    const size_t n4 = count >> 2;
    const f32x4* s4 = reinterpret_cast<const f32x4*>(slabs);
    f32x4* o4 = reinterpret_cast<f32x4*>(out);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
      f32x4 a = s4[i];
      for (int z = 1; z < nsplit; ++z) a += s4[(size_t)z * n4 + i];
      o4[i] = a;
    }
  } else {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
      float a = slabs[i];
      for (int z = 1; z < nsplit; ++z) a += slabs[(size_t)z * count + i];
      out[i] = a;
    }
  }
}

}  // namespace

extern "C" size_t vg_conv2d_wgrad_workspace_bytes(const vg_conv_geom* g) {
  if (vg_check_geom(g, "vg_conv2d_wgrad_workspace_bytes")) return 0;
  const WgradPlan pl = wgrad_plan(g);
  if (pl.nsplit <= 1) return 0;
  return (size_t)pl.nsplit * pl.M * pl.Ncols * sizeof(float);
}

extern "C" int vg_conv2d_wgrad(const vg_conv_geom* g, const float* gy, const float* x, float* gw, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (int e = vg_check_geom(g, "vg_conv2d_wgrad")) return e;
  VG_REQUIRE(gy && x && gw, "vg_conv2d_wgrad: null tensor pointer");
  const WgradPlan pl = wgrad_plan(g);
  const size_t need = pl.nsplit > 1 ? (size_t)pl.nsplit * pl.M * pl.Ncols * sizeof(float) : 0;
  if (need > 0 && (!workspace || workspace_bytes < need || !vg_aligned16(workspace)))
    return vg_fail(VG_ERR_WORKSPACE, "vg_conv2d_wgrad: workspace of %zu bytes (16-byte aligned) required, got %zu",
                   need, workspace_bytes);

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
  hipStream_t st = (hipStream_t)stream;
  if (igemm_launch<MODE_WGRAD, true>(p, pl.bn, va, vb, grid, st) != 0)
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_wgrad: no kernel variant for bn=%d va=%d vb=%d", pl.bn, va, vb);
  VG_CHECK_LAUNCH("vg_conv2d_wgrad");
  if (pl.nsplit > 1) {
    const size_t count = (size_t)pl.M * pl.Ncols;
    const int blocks = vg_stream_grid((count + 3) / 4, 256);
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)workspace, gw, count,
                       pl.nsplit);
    VG_CHECK_LAUNCH("vg_conv2d_wgrad(slab reduce)");
  }
  return VG_OK;
}
