// GAN loss reductions and the WGAN-GP penalty kernels.  Scalar reductions are two-level (per-block partials in
// the caller's workspace, then one block adds them in order) so the result does not depend on scheduling.
#include "vg_internal.h"

namespace {

constexpr int kMaxReduceBlocks = 256;

inline int reduce_blocks(size_t n) {
  size_t nb = (n + 256 * 16 - 1) / (256 * 16);
  if (nb < 1) nb = 1;
  if (nb > kMaxReduceBlocks) nb = kMaxReduceBlocks;
  return (int)nb;
}

template <class F>
__global__ void __launch_bounds__(256) scalar_partial(F f, size_t n, float* partial, float scale, float* out) {
  __shared__ float red[4];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += f(i);
  const float s = vg_block_sum_256(acc, red);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) out[0] = s * scale;
    else partial[blockIdx.x] = s;
  }
}

__global__ void __launch_bounds__(256) scalar_final(const float* partial, int nb, float scale, float* out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nb; i += 256) acc += partial[i];
  const float s = vg_block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = s * scale;
}

template <class F>
int run_scalar_reduce(const char* who, F f, size_t n, float scale, float* out, void* ws, size_t ws_bytes, hipStream_t st) {
  const int nb = reduce_blocks(n);
  if (nb > 1 && (!ws || ws_bytes < nb * sizeof(float)))
    return vg_fail(VG_ERR_WORKSPACE, "%s: workspace of %zu bytes required, got %zu", who, nb * sizeof(float), ws_bytes);
  hipLaunchKernelGGL((scalar_partial<F>), dim3(nb), dim3(256), 0, st, f, n, (float*)ws, scale, out);
  VG_CHECK_LAUNCH(who);
  if (nb > 1) {
    hipLaunchKernelGGL(scalar_final, dim3(1), dim3(256), 0, st, (const float*)ws, nb, scale, out);
    VG_CHECK_LAUNCH(who);
  }
  return VG_OK;
}

struct IdentF {
  const float* p;
  __device__ float operator()(size_t i) const { return p[i]; }
};
struct BceF {
  const float* p;
  const float* t;
  float tc;
  __device__ float operator()(size_t i) const {
    const float x = p[i], y = t ? t[i] : tc;
    const float l1 = fmaxf(logf(x), -100.f), l2 = fmaxf(log1pf(-x), -100.f);
    return -(y * l1 + (1.f - y) * l2);
  }
};
struct MseF {
  const float* p;
  const float* t;
  float tc;
  __device__ float operator()(size_t i) const {
    const float d = p[i] - (t ? t[i] : tc);
    return d * d;
  }
};
struct L1F {
  const float *a, *b;
  __device__ float operator()(size_t i) const { return fabsf(a[i] - b[i]); }
};

__global__ void __launch_bounds__(256) fill_scaled_kernel(const float* gout, float s, float* out, size_t n) {
  const float v = gout[0] * s;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}

__global__ void __launch_bounds__(256) bce_bwd_kernel(const float* gout, const float* p, const float* t, float tc, float inv_n,
                                                      float* dp, size_t n) {
  const float g = gout[0] * inv_n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float x = p[i], y = t ? t[i] : tc;
    dp[i] = g * (x - y) / fmaxf((1.f - x) * x, 1e-12f);
  }
}

__global__ void __launch_bounds__(256) mse_bwd_kernel(const float* gout, const float* p, const float* t, float tc, float inv_n,
                                                      float* dp, size_t n) {
  const float g = 2.f * gout[0] * inv_n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dp[i] = g * (p[i] - (t ? t[i] : tc));
}

__global__ void __launch_bounds__(256) l1_bwd_kernel(const float* gout, const float* a, const float* b, float inv_n, float* da,
                                                     float* db, size_t n) {
  const float g = gout[0] * inv_n;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float d = a[i] - b[i];
    const float s = d > 0.f ? g : (d < 0.f ? -g : 0.f);
    if (da) da[i] = s;
    if (db) db[i] = -s;
  }
}

// ---- WGAN-GP ---------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) gp_interp_kernel(const float* __restrict__ alpha, const float* __restrict__ real,
                                                        const float* __restrict__ fake, float* __restrict__ out, size_t nvec,
                                                        size_t dvec) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const float a = alpha[i / dvec];
    const float na = 1.f - a;
    if constexpr (VEC == 4) {
      const f32x4 r = reinterpret_cast<const f32x4*>(real)[i], f = reinterpret_cast<const f32x4*>(fake)[i];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = a * r[e] + na * f[e];
      reinterpret_cast<f32x4*>(out)[i] = o;
    } else {
      out[i] = a * real[i] + na * fake[i];
    }
  }
}

// one block per sample: norms[b] = ||g[b][:]||_2
__global__ void __launch_bounds__(256) gp_rownorm_kernel(const float* __restrict__ g, float* __restrict__ norms, size_t D,
                                                         int vec4) {
  __shared__ float red[4];
  const float* row = g + (size_t)blockIdx.x * D;
  float acc = 0.f;
  if (vec4) {
    const f32x4* r4 = reinterpret_cast<const f32x4*>(row);
    for (size_t i = threadIdx.x; i < D / 4; i += 256) {
      const f32x4 v = r4[i];
      acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
  } else {
    for (size_t i = threadIdx.x; i < D; i += 256) acc = fmaf(row[i], row[i], acc);
  }
  const float s = vg_block_sum_256(acc, red);
  if (threadIdx.x == 0) norms[blockIdx.x] = sqrtf(s);
}

__global__ void __launch_bounds__(256) gp_penalty_fwd_kernel(const float* norms, int B, float* out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float d = norms[i] - 1.f;
    acc = fmaf(d, d, acc);
  }
  const float s = vg_block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = s / (float)B;
}

template <int VEC>
__global__ void __launch_bounds__(256) gp_penalty_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ g,
                                                             const float* __restrict__ norms, float* __restrict__ dg,
                                                             size_t nvec, size_t dvec, float inv_b) {
  const float go = gout[0];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    const float nb = norms[i / dvec];
    const float coef = nb > 0.f ? go * 2.f * (nb - 1.f) * inv_b / nb : 0.f;
    if constexpr (VEC == 4) {
      const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = coef * v[e];
      reinterpret_cast<f32x4*>(dg)[i] = o;
    } else {
      dg[i] = coef * g[i];
    }
  }
}

}  // namespace

extern "C" size_t vg_reduce_workspace_bytes(size_t n) { return (size_t)reduce_blocks(n) * sizeof(float); }

extern "C" int vg_loss_mean_fwd(const float* pred, size_t n, float sign, float* out, void* ws, size_t ws_bytes, void* stream) {
  VG_REQUIRE(pred && out && n > 0, "vg_loss_mean_fwd: bad arguments");
  return run_scalar_reduce("vg_loss_mean_fwd", IdentF{pred}, n, sign / (float)n, out, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int vg_loss_mean_bwd(const float* gout, size_t n, float sign, float* dpred, void* stream) {
  VG_REQUIRE(gout && dpred && n > 0, "vg_loss_mean_bwd: bad arguments");
  hipLaunchKernelGGL(fill_scaled_kernel, dim3(vg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, gout,
                     sign / (float)n, dpred, n);
  VG_CHECK_LAUNCH("vg_loss_mean_bwd");
  return VG_OK;
}

extern "C" int vg_loss_bce_fwd(const float* pred, const float* target, float tconst, size_t n, float* out, void* ws,
                               size_t ws_bytes, void* stream) {
  VG_REQUIRE(pred && out && n > 0, "vg_loss_bce_fwd: bad arguments");
  return run_scalar_reduce("vg_loss_bce_fwd", BceF{pred, target, tconst}, n, 1.f / (float)n, out, ws, ws_bytes,
                           (hipStream_t)stream);
}

extern "C" int vg_loss_bce_bwd(const float* gout, const float* pred, const float* target, float tconst, size_t n,
                               float* dpred, void* stream) {
  VG_REQUIRE(gout && pred && dpred && n > 0, "vg_loss_bce_bwd: bad arguments");
  hipLaunchKernelGGL(bce_bwd_kernel, dim3(vg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, gout, pred, target,
                     tconst, 1.f / (float)n, dpred, n);
  VG_CHECK_LAUNCH("vg_loss_bce_bwd");
  return VG_OK;
}

extern "C" int vg_loss_mse_fwd(const float* pred, const float* target, float tconst, size_t n, float* out, void* ws,
                               size_t ws_bytes, void* stream) {
  VG_REQUIRE(pred && out && n > 0, "vg_loss_mse_fwd: bad arguments");
  return run_scalar_reduce("vg_loss_mse_fwd", MseF{pred, target, tconst}, n, 1.f / (float)n, out, ws, ws_bytes,
                           (hipStream_t)stream);
}

extern "C" int vg_loss_mse_bwd(const float* gout, const float* pred, const float* target, float tconst, size_t n,
                               float* dpred, void* stream) {
  VG_REQUIRE(gout && pred && dpred && n > 0, "vg_loss_mse_bwd: bad arguments");
  hipLaunchKernelGGL(mse_bwd_kernel, dim3(vg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, gout, pred, target,
                     tconst, 1.f / (float)n, dpred, n);
  VG_CHECK_LAUNCH("vg_loss_mse_bwd");
  return VG_OK;
}

extern "C" int vg_loss_l1_fwd(const float* a, const float* b, size_t n, float* out, void* ws, size_t ws_bytes, void* stream) {
  VG_REQUIRE(a && b && out && n > 0, "vg_loss_l1_fwd: bad arguments");
  return run_scalar_reduce("vg_loss_l1_fwd", L1F{a, b}, n, 1.f / (float)n, out, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int vg_loss_l1_bwd(const float* gout, const float* a, const float* b, size_t n, float* da, float* db,
                              void* stream) {
  VG_REQUIRE(gout && a && b && (da || db) && n > 0, "vg_loss_l1_bwd: bad arguments");
  hipLaunchKernelGGL(l1_bwd_kernel, dim3(vg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, gout, a, b,
                     1.f / (float)n, da, db, n);
  VG_CHECK_LAUNCH("vg_loss_l1_bwd");
  return VG_OK;
}

extern "C" int vg_gp_interp(const float* alpha, const float* real, const float* fake, float* out, int B, size_t D,
                            void* stream) {
  VG_REQUIRE(alpha && real && fake && out && B > 0 && D > 0, "vg_gp_interp: bad arguments");
  const bool v4 = D % 4 == 0 && vg_aligned16(real) && vg_aligned16(fake) && vg_aligned16(out);
  hipStream_t st = (hipStream_t)stream;
  if (v4) {
    const size_t nvec = (size_t)B * D / 4;
    hipLaunchKernelGGL((gp_interp_kernel<4>), dim3(vg_stream_grid(nvec, 256)), dim3(256), 0, st, alpha, real, fake, out, nvec,
                       D / 4);
  } else {
    const size_t nvec = (size_t)B * D;
    hipLaunchKernelGGL((gp_interp_kernel<1>), dim3(vg_stream_grid(nvec, 256)), dim3(256), 0, st, alpha, real, fake, out, nvec,
                       D);
  }
  VG_CHECK_LAUNCH("vg_gp_interp");
  return VG_OK;
}

extern "C" int vg_gp_rownorm(const float* g, float* norms, int B, size_t D, void* stream) {
  VG_REQUIRE(g && norms && B > 0 && D > 0, "vg_gp_rownorm: bad arguments");
  const int v4 = (D % 4 == 0 && vg_aligned16(g)) ? 1 : 0;
  hipLaunchKernelGGL(gp_rownorm_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, g, norms, D, v4);
  VG_CHECK_LAUNCH("vg_gp_rownorm");
  return VG_OK;
}

extern "C" int vg_gp_penalty_fwd(const float* norms, int B, float* out, void* stream) {
  VG_REQUIRE(norms && out && B > 0, "vg_gp_penalty_fwd: bad arguments");
  hipLaunchKernelGGL(gp_penalty_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, norms, B, out);
  VG_CHECK_LAUNCH("vg_gp_penalty_fwd");
  return VG_OK;
}

extern "C" int vg_gp_penalty_bwd(const float* gout, const float* g, const float* norms, float* dg, int B, size_t D,
                                 void* stream) {
  VG_REQUIRE(gout && g && norms && dg && B > 0 && D > 0, "vg_gp_penalty_bwd: bad arguments");
  const bool v4 = D % 4 == 0 && vg_aligned16(g) && vg_aligned16(dg);
  hipStream_t st = (hipStream_t)stream;
  const float inv_b = 1.f / (float)B;
  if (v4) {
    const size_t nvec = (size_t)B * D / 4;
    hipLaunchKernelGGL((gp_penalty_bwd_kernel<4>), dim3(vg_stream_grid(nvec, 256)), dim3(256), 0, st, gout, g, norms, dg, nvec,
                       D / 4, inv_b);
  } else {
    const size_t nvec = (size_t)B * D;
    hipLaunchKernelGGL((gp_penalty_bwd_kernel<1>), dim3(vg_stream_grid(nvec, 256)), dim3(256), 0, st, gout, g, norms, dg, nvec,
                       D, inv_b);
  }
  VG_CHECK_LAUNCH("vg_gp_penalty_bwd");
  return VG_OK;
}
