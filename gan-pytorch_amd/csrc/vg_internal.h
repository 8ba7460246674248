// Internal helpers shared by the gfx950 kernel translation units of libvegans_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/vegans_hip.h"

#include <type_traits>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Thread-local error text returned by vg_last_error().
char* vg_error_buffer();
int vg_fail(int code, const char* fmt, ...);

#define VG_REQUIRE(cond, ...)                          \
  do {                                                 \
    if (!(cond)) return vg_fail(VG_ERR_INVALID, __VA_ARGS__); \
  } while (0)

// Checks the launch itself (not completion: the library never synchronises).
#define VG_CHECK_LAUNCH(what)                                                        \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess)                                                           \
      return vg_fail(VG_ERR_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e__)); \
  } while (0)

static inline bool vg_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Memory-bound kernels: cap the grid and grid-stride the rest.
static inline int vg_stream_grid(size_t work_items, int block) {
  size_t blocks = (work_items + block - 1) / block;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  return (int)blocks;
}

// The BatchNorm pre-activation value, v = (x - mean) * (invstd * gamma) + beta, in ONE fixed sequence of operations (a
// subtraction, a product for the scale, a fused multiply-add). The forward pass and every kernel that rebuilds the
// fused activation's slope mask from x instead of reading the activation's output call this, so the mask is the sign
// the forward pass saw, bit for bit.
__device__ __forceinline__ float vg_bn_affine(float a, float mu, float r, const float* gamma, float ga, const float* beta,
                                              float be) {
  const float sc = gamma ? r * ga : r;
  const float d = a - mu;
  return beta ? __builtin_fmaf(d, sc, be) : d * sc;
}

// A BatchNorm (+ fused ReLU / LeakyReLU) whose output gradient a convolution is about to produce: its input x (laid out
// like the convolution's output), batch statistics, affine parameters (null = none) and activation.
struct VgNormBwdSrc {
  const float *x, *mean, *invstd, *gamma, *beta;
  int act;
  float slope;
};

// slope of ReLU / LeakyReLU at a value of that sign (other codes: 1)
__device__ __forceinline__ float vg_mask_grad(float ref, int act, float slope) {
  if (act == VG_ACT_RELU) return ref > 0.f ? 1.f : 0.f;
  if (act == VG_ACT_LRELU) return ref > 0.f ? 1.f : slope;
  return 1.f;
}

__device__ __forceinline__ float vg_wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Block-wide sum for 256-thread blocks; result valid in thread 0. `red` holds >= 4 floats.
__device__ __forceinline__ float vg_block_sum_256(float v, float* red) {
  v = vg_wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = (red[0] + red[1]) + (red[2] + red[3]);
  return r;
}

// One-shot kernel timing (vg_time_next_conv): when a pair of events is pending on this thread, the next contraction
// kernel is launched with them, and they then hold the begin and end of that kernel's execution on the device (the
// dispatch's own timestamps, not the moments a marker before and after it drained from the queue).
struct VgLaunchTiming {
  hipEvent_t start, stop;
};
VgLaunchTiming* vg_launch_timing();   // thread-local slot

template <class K, class... Args>
static inline void vg_launch_timed(K kernel, dim3 grid, dim3 block, unsigned lds, hipStream_t st, Args... args) {
  VgLaunchTiming* t = vg_launch_timing();
  if (t->start || t->stop) {
    hipExtLaunchKernelGGL(kernel, grid, block, lds, st, t->start, t->stop, 0, args...);
    t->start = t->stop = nullptr;
  } else {
    hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
  }
}
