// Max pooling over NHWC activations (the reference's image-to-image critic uses nn.MaxPool2d(5, 2, 2),
// examples/example_image_to_image.py:78-90).  HBM-bound: one thread per (output pixel, 4-channel group), 16-byte
// accesses along the channel dimension.  The forward pass records, per output element, the input pixel (hi*W+wi)
// that won; the backward pass is a gather over the windows that cover an input pixel, so overlapping windows need
// no atomics and the sum order is fixed.
#include "vg_internal.h"

namespace {

template <int VEC>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int* __restrict__ idx, vg_conv_geom g, size_t total) {
  const int CG = g.Ci / VEC;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int cg = (int)(i % CG);
    size_t p = i / CG;
    const int wo = (int)(p % g.Wo); p /= g.Wo;
    const int ho = (int)(p % g.Ho);
    const int n = (int)(p / g.Ho);
    float best[VEC];
    int where[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { best[e] = -INFINITY; where[e] = -1; }
    for (int a = 0; a < g.kh; ++a) {
      const int hi = ho * g.sh - g.ph + a;
      if ((unsigned)hi >= (unsigned)g.H) continue;
      for (int b = 0; b < g.kw; ++b) {
        const int wi = wo * g.sw - g.pw + b;
        if ((unsigned)wi >= (unsigned)g.W) continue;
        const float* src = x + ((size_t)(n * g.H + hi) * g.W + wi) * g.Ci + (size_t)cg * VEC;
        float v[VEC];
        if constexpr (VEC == 4) {
          const f32x4 t = *reinterpret_cast<const f32x4*>(src);
          v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        } else {
          v[0] = src[0];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          // first maximum in scan order wins; a NaN wins over everything (torch's max_pool2d rule)
          if (v[e] > best[e] || v[e] != v[e] || where[e] < 0) { best[e] = v[e]; where[e] = hi * g.W + wi; }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      y[i * VEC + e] = best[e];
      idx[i * VEC + e] = where[e];
    }
  }
}

// gx[n][hi][wi][c] = sum over output pixels (ho,wo) whose window covers (hi,wi) and whose winner is (hi,wi)
template <int VEC>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const float* __restrict__ gy, const int* __restrict__ idx,
                                                          float* __restrict__ gx, vg_conv_geom g, size_t total) {
  const int CG = g.Ci / VEC;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int cg = (int)(i % CG);
    size_t p = i / CG;
    const int wi = (int)(p % g.W); p /= g.W;
    const int hi = (int)(p % g.H);
    const int n = (int)(p / g.H);
    const int me = hi * g.W + wi;
    float sum[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) sum[e] = 0.f;
    const int ho_hi = min((hi + g.ph) / g.sh, g.Ho - 1), wo_hi = min((wi + g.pw) / g.sw, g.Wo - 1);
    for (int ho = ho_hi; ho >= 0 && ho * g.sh - g.ph + g.kh > hi; --ho) {
      for (int wo = wo_hi; wo >= 0 && wo * g.sw - g.pw + g.kw > wi; --wo) {
        const size_t o = (((size_t)(n * g.Ho + ho) * g.Wo + wo) * CG + cg) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (idx[o + e] == me) sum[e] += gy[o + e];
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) gx[i * VEC + e] = sum[e];
  }
}

// out[n][ho][wo][c] = t[n][winner pixel of (ho,wo,c)][c]: the derivative of the backward pass with respect to gy
__global__ void __launch_bounds__(256) maxpool_pick_kernel(const float* __restrict__ t, const int* __restrict__ idx,
                                                           float* __restrict__ out, vg_conv_geom g, size_t total) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int c = (int)(i % g.Ci);
    const int n = (int)(i / ((size_t)g.Ho * g.Wo * g.Ci));
    const int w = idx[i];
    out[i] = w >= 0 ? t[((size_t)n * g.H * g.W + w) * g.Ci + c] : 0.f;
  }
}

int check_pool(const vg_conv_geom* g, const char* who) {
  if (!g || g->N <= 0 || g->H <= 0 || g->W <= 0 || g->Ci <= 0 || g->Co != g->Ci || g->kh <= 0 || g->kw <= 0 ||
      g->sh <= 0 || g->sw <= 0 || g->ph < 0 || g->pw < 0 || 2 * g->ph > g->kh || 2 * g->pw > g->kw)
    return vg_fail(VG_ERR_INVALID, "%s: bad pooling geometry", who);
  if (g->Ho != (g->H + 2 * g->ph - g->kh) / g->sh + 1 || g->Wo != (g->W + 2 * g->pw - g->kw) / g->sw + 1 || g->Ho <= 0 ||
      g->Wo <= 0)
    return vg_fail(VG_ERR_INVALID, "%s: output size does not match floor((in + 2p - k) / s) + 1", who);
  return 0;
}

}  // namespace

extern "C" int vg_maxpool2d_fwd(const vg_conv_geom* g, const float* x, float* y, int* idx, void* stream) {
  if (int e = check_pool(g, "vg_maxpool2d_fwd")) return e;
  VG_REQUIRE(x && y && idx, "vg_maxpool2d_fwd: null pointer");
  const size_t out = (size_t)g->N * g->Ho * g->Wo * g->Ci;
  if (g->Ci % 4 == 0 && vg_aligned16(x))
    hipLaunchKernelGGL((maxpool_fwd_kernel<4>), dim3(vg_stream_grid(out / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, y,
                       idx, *g, out / 4);
  else
    hipLaunchKernelGGL((maxpool_fwd_kernel<1>), dim3(vg_stream_grid(out, 256)), dim3(256), 0, (hipStream_t)stream, x, y, idx,
                       *g, out);
  VG_CHECK_LAUNCH("vg_maxpool2d_fwd");
  return VG_OK;
}

extern "C" int vg_maxpool2d_bwd(const vg_conv_geom* g, const float* gy, const int* idx, float* gx, void* stream) {
  if (int e = check_pool(g, "vg_maxpool2d_bwd")) return e;
  VG_REQUIRE(gy && gx && idx, "vg_maxpool2d_bwd: null pointer");
  const size_t in = (size_t)g->N * g->H * g->W * g->Ci;
  if (g->Ci % 4 == 0)
    hipLaunchKernelGGL((maxpool_bwd_kernel<4>), dim3(vg_stream_grid(in / 4, 256)), dim3(256), 0, (hipStream_t)stream, gy, idx,
                       gx, *g, in / 4);
  else
    hipLaunchKernelGGL((maxpool_bwd_kernel<1>), dim3(vg_stream_grid(in, 256)), dim3(256), 0, (hipStream_t)stream, gy, idx, gx,
                       *g, in);
  VG_CHECK_LAUNCH("vg_maxpool2d_bwd");
  return VG_OK;
}

extern "C" int vg_maxpool2d_pick(const vg_conv_geom* g, const float* t, const int* idx, float* out, void* stream) {
  if (int e = check_pool(g, "vg_maxpool2d_pick")) return e;
  VG_REQUIRE(t && out && idx, "vg_maxpool2d_pick: null pointer");
  const size_t n = (size_t)g->N * g->Ho * g->Wo * g->Ci;
  hipLaunchKernelGGL(maxpool_pick_kernel, dim3(vg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, t, idx, out, *g, n);
  VG_CHECK_LAUNCH("vg_maxpool2d_pick");
  return VG_OK;
}
