// Fused single-pass RMSprop / Adam / SGD parameter updates with optional gradient pre-scale (data-parallel
// sums -> means) and the WGAN weight clamp folded into the same pass.  HBM-bound: every state tensor is read
// and written exactly once with 16-byte accesses.
#include "vg_internal.h"

namespace {

__device__ __forceinline__ float clamp_keep_nan(float v, float lo, float hi) {
  // matches torch clamp_: NaN stays NaN, otherwise min(max(v, lo), hi)
  if (v != v) return v;
  v = v < lo ? lo : v;
  return v > hi ? hi : v;
}

// one VEC-wide element group i of one tensor
template <int VEC>
__device__ __forceinline__ void rmsprop_update(float* __restrict__ param, const float* __restrict__ grad,
                                               float* __restrict__ sq, float* __restrict__ mbuf,
                                               float* __restrict__ gavg, size_t i, const vg_rmsprop_cfg& c, bool do_clamp) {
    float p[VEC], g[VEC], s[VEC], m[VEC], ga[VEC];
    if constexpr (VEC == 4) {
      const f32x4 pv = reinterpret_cast<f32x4*>(param)[i], gv = reinterpret_cast<const f32x4*>(grad)[i],
                  sv = reinterpret_cast<f32x4*>(sq)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) { p[e] = pv[e]; g[e] = gv[e]; s[e] = sv[e]; }
      if (mbuf) { const f32x4 mv = reinterpret_cast<f32x4*>(mbuf)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = mv[e]; }
      if (gavg) { const f32x4 av = reinterpret_cast<f32x4*>(gavg)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) ga[e] = av[e]; }
    } else {
      p[0] = param[i]; g[0] = grad[i]; s[0] = sq[i];
      if (mbuf) m[0] = mbuf[i];
      if (gavg) ga[0] = gavg[i];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float gr = g[e] * c.grad_scale;
      if (c.weight_decay != 0.f) gr = gr + c.weight_decay * p[e];
      s[e] = s[e] * c.alpha + (1.f - c.alpha) * gr * gr;
      float avg;
      if (c.centered) {
        ga[e] = ga[e] + (1.f - c.alpha) * (gr - ga[e]);
        avg = sqrtf(s[e] - ga[e] * ga[e]) + c.eps;
      } else {
        avg = sqrtf(s[e]) + c.eps;
      }
      if (c.momentum > 0.f) {
        m[e] = m[e] * c.momentum + gr / avg;
        p[e] = p[e] - c.lr * m[e];
      } else {
        p[e] = p[e] + (-c.lr * gr) / avg;
      }
      if (do_clamp) p[e] = clamp_keep_nan(p[e], c.clamp_lo, c.clamp_hi);
    }
    if constexpr (VEC == 4) {
      reinterpret_cast<f32x4*>(param)[i] = f32x4{p[0], p[1], p[2], p[3]};
      reinterpret_cast<f32x4*>(sq)[i] = f32x4{s[0], s[1], s[2], s[3]};
      if (mbuf) reinterpret_cast<f32x4*>(mbuf)[i] = f32x4{m[0], m[1], m[2], m[3]};
      if (gavg) reinterpret_cast<f32x4*>(gavg)[i] = f32x4{ga[0], ga[1], ga[2], ga[3]};
    } else {
      param[i] = p[0]; sq[i] = s[0];
      if (mbuf) mbuf[i] = m[0];
      if (gavg) gavg[i] = ga[0];
    }
}

template <int VEC>
__global__ void __launch_bounds__(256) rmsprop_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                      float* __restrict__ sq, float* __restrict__ mbuf,
                                                      float* __restrict__ gavg, size_t nvec, vg_rmsprop_cfg c) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const bool do_clamp = c.clamp_lo <= c.clamp_hi;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride)
    rmsprop_update<VEC>(param, grad, sq, mbuf, gavg, i, c, do_clamp);
}

// Many tensors, one launch: block b works on one 2048-element chunk of one tensor (two 16-byte groups per thread: with
// 8192 the few hundred blocks of a DCGAN network left the chip latency-bound) (first_block[] is the running
// count of chunks). A network's update is otherwise a dozen launches of a few microseconds each.
constexpr int kMultiMax = 24;
constexpr int kMultiChunk = 2048;
struct RmspropMulti {
  float* param[kMultiMax];
  const float* grad[kMultiMax];
  float* sq[kMultiMax];
  float* mbuf[kMultiMax];
  float* gavg[kMultiMax];
  unsigned long long n[kMultiMax];
  int first_block[kMultiMax + 1];
  int vec[kMultiMax];
  int count;
};

__global__ void __launch_bounds__(256) rmsprop_multi_kernel(RmspropMulti t, vg_rmsprop_cfg c) {
  int k = 0;
  while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;    // block-uniform
  const size_t e0 = (size_t)((int)blockIdx.x - t.first_block[k]) * kMultiChunk;
  const size_t e1 = e0 + kMultiChunk < t.n[k] ? e0 + kMultiChunk : (size_t)t.n[k];
  const bool do_clamp = c.clamp_lo <= c.clamp_hi;
  if (t.vec[k]) {
    for (size_t i = e0 / 4 + threadIdx.x; i < e1 / 4; i += 256)
      rmsprop_update<4>(t.param[k], t.grad[k], t.sq[k], t.mbuf[k], t.gavg[k], i, c, do_clamp);
  } else {
    for (size_t i = e0 + threadIdx.x; i < e1; i += 256)
      rmsprop_update<1>(t.param[k], t.grad[k], t.sq[k], t.mbuf[k], t.gavg[k], i, c, do_clamp);
  }
}

// one VEC-wide element group i of one tensor
template <int VEC>
__device__ __forceinline__ void adam_update(float* __restrict__ param, const float* __restrict__ grad,
                                            float* __restrict__ ea, float* __restrict__ eas, float* __restrict__ mx,
                                            size_t i, const vg_adam_cfg& c, bool do_clamp, float w1, float step_size,
                                            float bc2_sqrt) {
    float p[VEC], g[VEC], m[VEC], v[VEC], x[VEC];
    if constexpr (VEC == 4) {
      const f32x4 pv = reinterpret_cast<f32x4*>(param)[i], gv = reinterpret_cast<const f32x4*>(grad)[i],
                  mv = reinterpret_cast<f32x4*>(ea)[i], vv = reinterpret_cast<f32x4*>(eas)[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) { p[e] = pv[e]; g[e] = gv[e]; m[e] = mv[e]; v[e] = vv[e]; }
      if (mx) { const f32x4 xv = reinterpret_cast<f32x4*>(mx)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) x[e] = xv[e]; }
    } else {
      p[0] = param[i]; g[0] = grad[i]; m[0] = ea[i]; v[0] = eas[i];
      if (mx) x[0] = mx[i];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float gr = g[e] * c.grad_scale;
      if (c.decoupled_wd) {
        p[e] = p[e] * (1.f - c.lr * c.weight_decay);
      } else if (c.weight_decay != 0.f) {
        gr = gr + c.weight_decay * p[e];
      }
      // torch lerp_: start + w*(end-start) for w < 0.5, end - (end-start)*(1-w) otherwise
      m[e] = w1 < 0.5f ? m[e] + w1 * (gr - m[e]) : gr - (gr - m[e]) * (1.f - w1);
      v[e] = v[e] * c.beta2 + (1.f - c.beta2) * gr * gr;
      float denom;
      if (mx) {
        x[e] = fmaxf(x[e], v[e]);
        denom = sqrtf(x[e]) / bc2_sqrt + c.eps;
      } else {
        denom = sqrtf(v[e]) / bc2_sqrt + c.eps;
      }
      p[e] = p[e] + (-step_size * m[e]) / denom;
      if (do_clamp) p[e] = clamp_keep_nan(p[e], c.clamp_lo, c.clamp_hi);
    }
    if constexpr (VEC == 4) {
      reinterpret_cast<f32x4*>(param)[i] = f32x4{p[0], p[1], p[2], p[3]};
      reinterpret_cast<f32x4*>(ea)[i] = f32x4{m[0], m[1], m[2], m[3]};
      reinterpret_cast<f32x4*>(eas)[i] = f32x4{v[0], v[1], v[2], v[3]};
      if (mx) reinterpret_cast<f32x4*>(mx)[i] = f32x4{x[0], x[1], x[2], x[3]};
    } else {
      param[i] = p[0]; ea[i] = m[0]; eas[i] = v[0];
      if (mx) mx[i] = x[0];
    }
}

template <int VEC>
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                   float* __restrict__ ea, float* __restrict__ eas, float* __restrict__ mx,
                                                   size_t nvec, vg_adam_cfg c) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const bool do_clamp = c.clamp_lo <= c.clamp_hi;
  const float w1 = 1.f - c.beta1;
  // step-dependent scalars: by value, or (replayed graphs) from two floats the host refreshes before every launch
  const float step_size = c.step_scalars ? c.step_scalars[0] : c.step_size;
  const float bc2_sqrt = c.step_scalars ? c.step_scalars[1] : c.bias_correction2_sqrt;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride)
    adam_update<VEC>(param, grad, ea, eas, mx, i, c, do_clamp, w1, step_size, bc2_sqrt);
}

struct AdamMulti {
  float* param[kMultiMax];
  const float* grad[kMultiMax];
  float* ea[kMultiMax];
  float* eas[kMultiMax];
  float* mx[kMultiMax];
  unsigned long long n[kMultiMax];
  int first_block[kMultiMax + 1];
  int vec[kMultiMax];
  int count;
};

__global__ void __launch_bounds__(256) adam_multi_kernel(AdamMulti t, vg_adam_cfg c) {
  int k = 0;
  while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;    // block-uniform
  const size_t e0 = (size_t)((int)blockIdx.x - t.first_block[k]) * kMultiChunk;
  const size_t e1 = e0 + kMultiChunk < t.n[k] ? e0 + kMultiChunk : (size_t)t.n[k];
  const bool do_clamp = c.clamp_lo <= c.clamp_hi;
  const float w1 = 1.f - c.beta1;
  const float step_size = c.step_scalars ? c.step_scalars[0] : c.step_size;
  const float bc2_sqrt = c.step_scalars ? c.step_scalars[1] : c.bias_correction2_sqrt;
  if (t.vec[k]) {
    for (size_t i = e0 / 4 + threadIdx.x; i < e1 / 4; i += 256)
      adam_update<4>(t.param[k], t.grad[k], t.ea[k], t.eas[k], t.mx[k], i, c, do_clamp, w1, step_size, bc2_sqrt);
  } else {
    for (size_t i = e0 + threadIdx.x; i < e1; i += 256)
      adam_update<1>(t.param[k], t.grad[k], t.ea[k], t.eas[k], t.mx[k], i, c, do_clamp, w1, step_size, bc2_sqrt);
  }
}

// Gradient accumulation owned by the engine: out = scale * (((src0 + src1) + src2) + ...), the sources added left to
// right in the order the host lists them (the order is part of the result in floating point). One launch serves many
// tensors, laid out over blocks like the optimizer kernels above; `out` may be one of the sources.
constexpr int kSumMaxSrc = 8;
struct GradSumMulti {
  float* out[kMultiMax];
  const float* src[kMultiMax][kSumMaxSrc];
  unsigned long long n[kMultiMax];
  float scale[kMultiMax];
  int first_block[kMultiMax + 1];
  int nsrc[kMultiMax];
  int vec[kMultiMax];
  int count;
};

__global__ void __launch_bounds__(256) grad_sum_multi_kernel(GradSumMulti t) {
  int k = 0;
  while (k + 1 < t.count && (int)blockIdx.x >= t.first_block[k + 1]) ++k;    // block-uniform
  const size_t e0 = (size_t)((int)blockIdx.x - t.first_block[k]) * kMultiChunk;
  const size_t e1 = e0 + kMultiChunk < t.n[k] ? e0 + kMultiChunk : (size_t)t.n[k];
  const int ns = t.nsrc[k];
  const float scale = t.scale[k];
  if (t.vec[k]) {
    for (size_t i = e0 / 4 + threadIdx.x; i < e1 / 4; i += 256) {
      f32x4 v[kSumMaxSrc];
#pragma unroll
      for (int s = 0; s < kSumMaxSrc; ++s)
        if (s < ns) v[s] = reinterpret_cast<const f32x4*>(t.src[k][s])[i];
      f32x4 a = v[0];
#pragma unroll
      for (int s = 1; s < kSumMaxSrc; ++s)
        if (s < ns) a += v[s];
      if (scale != 1.f) a *= scale;
      reinterpret_cast<f32x4*>(t.out[k])[i] = a;
    }
  } else {
    for (size_t i = e0 + threadIdx.x; i < e1; i += 256) {
      float a = t.src[k][0][i];
      for (int s = 1; s < ns; ++s) a += t.src[k][s][i];
      if (scale != 1.f) a *= scale;
      t.out[k][i] = a;
    }
  }
}

__global__ void __launch_bounds__(256) sgd_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                  float* __restrict__ mbuf, size_t n, vg_sgd_cfg c) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const bool do_clamp = c.clamp_lo <= c.clamp_hi;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float p = param[i];
    float g = grad[i] * c.grad_scale;
    if (c.weight_decay != 0.f) g = g + c.weight_decay * p;
    if (mbuf) {
      float b = c.first_step ? g : mbuf[i] * c.momentum + (1.f - c.dampening) * g;
      mbuf[i] = b;
      g = c.nesterov ? g + c.momentum * b : b;
    }
    p = p - c.lr * g;
    if (do_clamp) p = clamp_keep_nan(p, c.clamp_lo, c.clamp_hi);
    param[i] = p;
  }
}

__global__ void __launch_bounds__(256) clamp_kernel(float* __restrict__ p, size_t n, float lo, float hi) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = clamp_keep_nan(p[i], lo, hi);
}

}  // namespace

extern "C" int vg_rmsprop_step(float* param, const float* grad, float* square_avg, float* momentum_buf, float* grad_avg,
                               size_t n, const vg_rmsprop_cfg* cfg, void* stream) {
  VG_REQUIRE(param && grad && square_avg && cfg && n > 0, "vg_rmsprop_step: bad arguments");
  VG_REQUIRE(!(cfg->momentum > 0.f) || momentum_buf, "vg_rmsprop_step: momentum > 0 needs momentum_buf");
  VG_REQUIRE(!cfg->centered || grad_avg, "vg_rmsprop_step: centered needs grad_avg");
  float* mb = cfg->momentum > 0.f ? momentum_buf : nullptr;
  float* ga = cfg->centered ? grad_avg : nullptr;
  const bool v4 = n % 4 == 0 && vg_aligned16(param) && vg_aligned16(grad) && vg_aligned16(square_avg) &&
                  (!mb || vg_aligned16(mb)) && (!ga || vg_aligned16(ga));
  hipStream_t st = (hipStream_t)stream;
  if (v4)
    hipLaunchKernelGGL((rmsprop_kernel<4>), dim3(vg_stream_grid(n / 4, 256)), dim3(256), 0, st, param, grad, square_avg, mb, ga,
                       n / 4, *cfg);
  else
    hipLaunchKernelGGL((rmsprop_kernel<1>), dim3(vg_stream_grid(n, 256)), dim3(256), 0, st, param, grad, square_avg, mb, ga, n,
                       *cfg);
  VG_CHECK_LAUNCH("vg_rmsprop_step");
  return VG_OK;
}

extern "C" int vg_rmsprop_step_multi(int count, float* const* params, const float* const* grads, float* const* square_avgs,
                                     float* const* momentum_bufs, float* const* grad_avgs, const size_t* n,
                                     const vg_rmsprop_cfg* cfg, void* stream) {
  VG_REQUIRE(count > 0 && params && grads && square_avgs && n && cfg, "vg_rmsprop_step_multi: bad arguments");
  VG_REQUIRE(!(cfg->momentum > 0.f) || momentum_bufs, "vg_rmsprop_step_multi: momentum > 0 needs momentum buffers");
  VG_REQUIRE(!cfg->centered || grad_avgs, "vg_rmsprop_step_multi: centered needs grad_avg buffers");
  for (int base = 0; base < count; base += kMultiMax) {
    RmspropMulti t;
    t.count = count - base < kMultiMax ? count - base : kMultiMax;
    int blocks = 0;
    for (int k = 0; k < t.count; ++k) {
      const int j = base + k;
      VG_REQUIRE(params[j] && grads[j] && square_avgs[j] && n[j] > 0, "vg_rmsprop_step_multi: null tensor or empty tensor");
      t.param[k] = params[j];
      t.grad[k] = grads[j];
      t.sq[k] = square_avgs[j];
      t.mbuf[k] = cfg->momentum > 0.f ? momentum_bufs[j] : nullptr;
      t.gavg[k] = cfg->centered ? grad_avgs[j] : nullptr;
      VG_REQUIRE(!(cfg->momentum > 0.f) || t.mbuf[k], "vg_rmsprop_step_multi: missing momentum buffer");
      VG_REQUIRE(!cfg->centered || t.gavg[k], "vg_rmsprop_step_multi: missing grad_avg buffer");
      t.n[k] = n[j];
      t.vec[k] = n[j] % 4 == 0 && vg_aligned16(t.param[k]) && vg_aligned16(t.grad[k]) && vg_aligned16(t.sq[k]) &&
                 (!t.mbuf[k] || vg_aligned16(t.mbuf[k])) && (!t.gavg[k] || vg_aligned16(t.gavg[k]));
      t.first_block[k] = blocks;
      blocks += (int)((n[j] + kMultiChunk - 1) / kMultiChunk);
    }
    t.first_block[t.count] = blocks;
    hipLaunchKernelGGL(rmsprop_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, *cfg);
    VG_CHECK_LAUNCH("vg_rmsprop_step_multi");
  }
  return VG_OK;
}

extern "C" int vg_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                            size_t n, const vg_adam_cfg* cfg, void* stream) {
  VG_REQUIRE(param && grad && exp_avg && exp_avg_sq && cfg && n > 0, "vg_adam_step: bad arguments");
  VG_REQUIRE(!cfg->amsgrad || max_exp_avg_sq, "vg_adam_step: amsgrad needs max_exp_avg_sq");
  float* mx = cfg->amsgrad ? max_exp_avg_sq : nullptr;
  const bool v4 = n % 4 == 0 && vg_aligned16(param) && vg_aligned16(grad) && vg_aligned16(exp_avg) &&
                  vg_aligned16(exp_avg_sq) && (!mx || vg_aligned16(mx));
  hipStream_t st = (hipStream_t)stream;
  if (v4)
    hipLaunchKernelGGL((adam_kernel<4>), dim3(vg_stream_grid(n / 4, 256)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq,
                       mx, n / 4, *cfg);
  else
    hipLaunchKernelGGL((adam_kernel<1>), dim3(vg_stream_grid(n, 256)), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, mx,
                       n, *cfg);
  VG_CHECK_LAUNCH("vg_adam_step");
  return VG_OK;
}

extern "C" int vg_adam_step_multi(int count, float* const* params, const float* const* grads, float* const* exp_avgs,
                                  float* const* exp_avg_sqs, float* const* max_exp_avg_sqs, const size_t* n,
                                  const vg_adam_cfg* cfg, void* stream) {
  VG_REQUIRE(count > 0 && params && grads && exp_avgs && exp_avg_sqs && n && cfg, "vg_adam_step_multi: bad arguments");
  VG_REQUIRE(!cfg->amsgrad || max_exp_avg_sqs, "vg_adam_step_multi: amsgrad needs max_exp_avg_sq buffers");
  for (int base = 0; base < count; base += kMultiMax) {
    AdamMulti t;
    t.count = count - base < kMultiMax ? count - base : kMultiMax;
    int blocks = 0;
    for (int k = 0; k < t.count; ++k) {
      const int j = base + k;
      VG_REQUIRE(params[j] && grads[j] && exp_avgs[j] && exp_avg_sqs[j] && n[j] > 0,
                 "vg_adam_step_multi: null tensor or empty tensor");
      t.param[k] = params[j];
      t.grad[k] = grads[j];
      t.ea[k] = exp_avgs[j];
      t.eas[k] = exp_avg_sqs[j];
      t.mx[k] = cfg->amsgrad ? max_exp_avg_sqs[j] : nullptr;
      VG_REQUIRE(!cfg->amsgrad || t.mx[k], "vg_adam_step_multi: missing max_exp_avg_sq buffer");
      t.n[k] = n[j];
      t.vec[k] = n[j] % 4 == 0 && vg_aligned16(t.param[k]) && vg_aligned16(t.grad[k]) && vg_aligned16(t.ea[k]) &&
                 vg_aligned16(t.eas[k]) && (!t.mx[k] || vg_aligned16(t.mx[k]));
      t.first_block[k] = blocks;
      blocks += (int)((n[j] + kMultiChunk - 1) / kMultiChunk);
    }
    t.first_block[t.count] = blocks;
    hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t, *cfg);
    VG_CHECK_LAUNCH("vg_adam_step_multi");
  }
  return VG_OK;
}

extern "C" int vg_grad_sum_multi(int count, float* const* outs, const float* const* srcs, const int* nsrc, const size_t* n,
                                 const float* scales, void* stream) {
  VG_REQUIRE(count > 0 && outs && srcs && nsrc && n && scales, "vg_grad_sum_multi: bad arguments");
  for (int base = 0; base < count; base += kMultiMax) {
    GradSumMulti t;
    t.count = count - base < kMultiMax ? count - base : kMultiMax;
    int blocks = 0;
    for (int k = 0; k < t.count; ++k) {
      const int j = base + k;
      VG_REQUIRE(outs[j] && n[j] > 0 && nsrc[j] >= 1 && nsrc[j] <= kSumMaxSrc,
                 "vg_grad_sum_multi: tensor %d: null output, empty tensor or source count outside 1..%d", j, kSumMaxSrc);
      t.out[k] = outs[j];
      bool vec = n[j] % 4 == 0 && vg_aligned16(outs[j]);
      for (int s = 0; s < kSumMaxSrc; ++s) {
        const float* p = s < nsrc[j] ? srcs[(size_t)j * kSumMaxSrc + s] : nullptr;
        VG_REQUIRE(s >= nsrc[j] || p, "vg_grad_sum_multi: tensor %d: null source %d", j, s);
        t.src[k][s] = p;
        vec = vec && (!p || vg_aligned16(p));
      }
      t.n[k] = n[j];
      t.nsrc[k] = nsrc[j];
      t.scale[k] = scales[j];
      t.vec[k] = vec ? 1 : 0;
      t.first_block[k] = blocks;
      blocks += (int)((n[j] + kMultiChunk - 1) / kMultiChunk);
    }
    t.first_block[t.count] = blocks;
    hipLaunchKernelGGL(grad_sum_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
    VG_CHECK_LAUNCH("vg_grad_sum_multi");
  }
  return VG_OK;
}

extern "C" int vg_sgd_step(float* param, const float* grad, float* momentum_buf, size_t n, const vg_sgd_cfg* cfg,
                           void* stream) {
  VG_REQUIRE(param && grad && cfg && n > 0, "vg_sgd_step: bad arguments");
  VG_REQUIRE(!(cfg->momentum != 0.f) || momentum_buf, "vg_sgd_step: momentum needs momentum_buf");
  float* mb = cfg->momentum != 0.f ? momentum_buf : nullptr;
  hipLaunchKernelGGL(sgd_kernel, dim3(vg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, param, grad, mb, n, *cfg);
  VG_CHECK_LAUNCH("vg_sgd_step");
  return VG_OK;
}

extern "C" int vg_clamp(float* param, size_t n, float lo, float hi, void* stream) {
  VG_REQUIRE(param && n > 0, "vg_clamp: bad arguments");
  hipLaunchKernelGGL(clamp_kernel, dim3(vg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, param, n, lo, hi);
  VG_CHECK_LAUNCH("vg_clamp");
  return VG_OK;
}
