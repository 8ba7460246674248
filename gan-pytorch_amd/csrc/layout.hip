// Layout copies (NCHW <-> NHWC), channel concatenate / slice for conditional inputs, small elementwise helpers,
// and the library's error reporting.
#include "vg_internal.h"

#include <string.h>

char* vg_error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

int vg_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(vg_error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* vg_last_error(void) { return vg_error_buffer(); }
extern "C" int vg_abi_version(void) { return 1; }

VgLaunchTiming* vg_launch_timing() {
  static thread_local VgLaunchTiming slot = {nullptr, nullptr};
  return &slot;
}

extern "C" int vg_time_next_conv(void* start_event, void* stop_event) {
  VgLaunchTiming* t = vg_launch_timing();
  t->start = (hipEvent_t)start_event;
  t->stop = (hipEvent_t)stop_event;
  return VG_OK;
}

extern "C" int vg_time_pending(void) {
  const VgLaunchTiming* t = vg_launch_timing();
  return (t->start || t->stop) ? 1 : 0;
}

namespace {

// Per image: src [R][S] -> dst [S][R] through a 32x33 LDS tile (coalesced on both sides).
// SRC = double: the batch arrives as float64 (a numpy default) and is rounded to f32 on its way through the tile.
template <typename SRC>
__global__ void __launch_bounds__(256) transpose_kernel(const SRC* __restrict__ src, float* __restrict__ dst, int R, int S) {
  __shared__ float tile[32][33];
  const size_t img = (size_t)blockIdx.z * R * S;
  const int s0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int rr = r0 + ty + j, ss = s0 + tx;
    if (rr < R && ss < S) tile[ty + j][tx] = (float)src[img + (size_t)rr * S + ss];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 32; j += 8) {
    const int ss = s0 + ty + j, rr = r0 + tx;
    if (rr < R && ss < S) dst[img + (size_t)ss * R + rr] = tile[tx][ty + j];
  }
}

template <typename SRC>
int launch_transpose(const char* who, const SRC* src, float* dst, int N, int R, int S, hipStream_t st) {
  if (!(src && dst && N > 0 && R > 0 && S > 0)) return vg_fail(VG_ERR_INVALID, "%s: bad arguments", who);
  if (N > 65535) return vg_fail(VG_ERR_INVALID, "%s: batch %d exceeds the grid z limit", who, N);
  dim3 grid((S + 31) / 32, (R + 31) / 32, N);
  hipLaunchKernelGGL(transpose_kernel<SRC>, grid, dim3(256), 0, st, src, dst, R, S);
  VG_CHECK_LAUNCH(who);
  return VG_OK;
}

__global__ void __launch_bounds__(256) concat_kernel(const float* __restrict__ a, int C1, int a_bcast, const float* __restrict__ b,
                                                     int C2, int b_bcast, float* __restrict__ out, size_t total, int HW) {
  const int Ct = C1 + C2;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t pix = i / Ct;
    const int c = (int)(i - pix * Ct);
    const size_t n = pix / HW;
    float v;
    if (c < C1) v = a_bcast ? a[n * C1 + c] : a[pix * C1 + c];
    else v = b_bcast ? b[n * C2 + (c - C1)] : b[pix * C2 + (c - C1)];
    out[i] = v;
  }
}

__global__ void __launch_bounds__(256) slice_kernel(const float* __restrict__ src, int Ct, int c0, int Cp, float* __restrict__ dst,
                                                    size_t total) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const size_t pix = i / Cp;
    const int c = (int)(i - pix * Cp);
    dst[i] = src[pix * Ct + c0 + c];
  }
}

__global__ void __launch_bounds__(256) add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                  size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a[i] + b[i];
}

__global__ void __launch_bounds__(256) scale_kernel(const float* __restrict__ a, float s, float* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a[i] * s;
}

__global__ void __launch_bounds__(256) fill_kernel(float* __restrict__ out, float v, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = v;
}

// ---- bf16 storage: conversions and the elementwise helpers on bf16 tensors; a lane handles 8 consecutive values ----
__device__ __forceinline__ void widen8(const u32x4 t, float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[2 * j] = __builtin_bit_cast(float, t[j] << 16);
    v[2 * j + 1] = __builtin_bit_cast(float, t[j] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 narrow8(const float (&v)[8]) {   // round to nearest even
  u32x4 t;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 p = {v[2 * j], v[2 * j + 1]};
    t[j] = __builtin_bit_cast(unsigned int, __builtin_convertvector(p, bf16x2));
  }
  return t;
}

__global__ void __launch_bounds__(256) narrow_kernel(const float* __restrict__ in, __bf16* __restrict__ out, size_t n, int vec) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * 8;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (vec && i + 8 <= n) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(in + i), b = *reinterpret_cast<const f32x4*>(in + i + 4);
      const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
      *reinterpret_cast<u32x4*>(out + i) = narrow8(v);
    } else {
      for (size_t j = i; j < n && j < i + 8; ++j) out[j] = (__bf16)in[j];
    }
  }
}

__global__ void __launch_bounds__(256) widen_kernel(const __bf16* __restrict__ in, float* __restrict__ out, size_t n, int vec) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * 8;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (vec && i + 8 <= n) {
      float v[8];
      widen8(*reinterpret_cast<const u32x4*>(in + i), v);
      const f32x4 a = {v[0], v[1], v[2], v[3]}, b = {v[4], v[5], v[6], v[7]};
      *reinterpret_cast<f32x4*>(out + i) = a;
      *reinterpret_cast<f32x4*>(out + i + 4) = b;
    } else {
      for (size_t j = i; j < n && j < i + 8; ++j) out[j] = (float)in[j];
    }
  }
}

__global__ void __launch_bounds__(256) add_h_kernel(const __bf16* __restrict__ a, const __bf16* __restrict__ b,
                                                    __bf16* __restrict__ out, size_t n, int vec) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * 8;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (vec && i + 8 <= n) {
      float x[8], y[8];
      widen8(*reinterpret_cast<const u32x4*>(a + i), x);
      widen8(*reinterpret_cast<const u32x4*>(b + i), y);
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] += y[j];
      *reinterpret_cast<u32x4*>(out + i) = narrow8(x);
    } else {
      for (size_t j = i; j < n && j < i + 8; ++j) out[j] = (__bf16)((float)a[j] + (float)b[j]);
    }
  }
}

__global__ void __launch_bounds__(256) fill_h_kernel(__bf16* __restrict__ out, float v, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const __bf16 h = (__bf16)v;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = h;
}

// out[0] = c0*a[0] + c1*b[0] + c2*c[0] (b, c nullable); aux[0] = a[0]/b[0] if aux: the scalar epilogue of the
// adversary loss: 0.5*(fake+real) + lambda*penalty and the logged real/fake ratio, without leaving the device.
__global__ void scalar_combine_kernel(const float* a, float c0, const float* b, float c1, const float* c, float c2, float* out,
                                      float* ratio_b_over_a) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float v = c0 * a[0];
    if (b) v += c1 * b[0];
    if (c) v += c2 * c[0];
    out[0] = v;
    if (ratio_b_over_a && b) ratio_b_over_a[0] = b[0] / a[0];
  }
}

}  // namespace

extern "C" int vg_nchw_to_nhwc(const float* src, float* dst, int N, int C, int HW, void* stream) {
  return launch_transpose("vg_nchw_to_nhwc", src, dst, N, C, HW, (hipStream_t)stream);
}

extern "C" int vg_nchw_to_nhwc_f64(const double* src, float* dst, int N, int C, int HW, void* stream) {
  return launch_transpose("vg_nchw_to_nhwc_f64", src, dst, N, C, HW, (hipStream_t)stream);
}

namespace {
__global__ void __launch_bounds__(256) narrow_kernel(const double* __restrict__ src, float* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (float)src[i];
}
}  // namespace

extern "C" int vg_cast_f64_to_f32(const double* src, float* dst, size_t n, void* stream) {
  if (!(src && dst)) return vg_fail(VG_ERR_INVALID, "vg_cast_f64_to_f32: null pointer");
  if (n == 0) return VG_OK;
  const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(narrow_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, n);
  VG_CHECK_LAUNCH("vg_cast_f64_to_f32");
  return VG_OK;
}

extern "C" int vg_nhwc_to_nchw(const float* src, float* dst, int N, int C, int HW, void* stream) {
  return launch_transpose("vg_nhwc_to_nchw", src, dst, N, HW, C, (hipStream_t)stream);
}

extern "C" int vg_concat_channels(const float* a, int C1, int a_bcast, const float* b, int C2, int b_bcast, float* out, int N,
                                  int HW, void* stream) {
  VG_REQUIRE(a && b && out && C1 > 0 && C2 > 0 && N > 0 && HW > 0, "vg_concat_channels: bad arguments");
  const size_t total = (size_t)N * HW * (C1 + C2);
  hipLaunchKernelGGL(concat_kernel, dim3(vg_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, a, C1, a_bcast, b, C2,
                     b_bcast, out, total, HW);
  VG_CHECK_LAUNCH("vg_concat_channels");
  return VG_OK;
}

extern "C" int vg_slice_channels(const float* src, int Ctot, int c0, int Cp, float* dst, size_t pixels, void* stream) {
  VG_REQUIRE(src && dst && Ctot > 0 && c0 >= 0 && Cp > 0 && c0 + Cp <= Ctot && pixels > 0, "vg_slice_channels: bad arguments");
  const size_t total = pixels * Cp;
  hipLaunchKernelGGL(slice_kernel, dim3(vg_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, src, Ctot, c0, Cp, dst,
                     total);
  VG_CHECK_LAUNCH("vg_slice_channels");
  return VG_OK;
}

extern "C" int vg_add(const float* a, const float* b, float* out, size_t n, void* stream) {
  VG_REQUIRE(a && b && out && n > 0, "vg_add: bad arguments");
  hipLaunchKernelGGL(add_kernel, dim3(vg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, a, b, out, n);
  VG_CHECK_LAUNCH("vg_add");
  return VG_OK;
}

extern "C" int vg_add_h(const void* a, const void* b, void* out, size_t n, void* stream) {
  VG_REQUIRE(a && b && out && n > 0, "vg_add_h: bad arguments");
  const int vec = vg_aligned16(a) && vg_aligned16(b) && vg_aligned16(out) ? 1 : 0;
  hipLaunchKernelGGL(add_h_kernel, dim3(vg_stream_grid((n + 7) / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)a,
                     (const __bf16*)b, (__bf16*)out, n, vec);
  VG_CHECK_LAUNCH("vg_add_h");
  return VG_OK;
}

extern "C" int vg_fill_h(void* out, float v, size_t n, void* stream) {
  VG_REQUIRE(out && n > 0, "vg_fill_h: bad arguments");
  hipLaunchKernelGGL(fill_h_kernel, dim3(vg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, (__bf16*)out, v, n);
  VG_CHECK_LAUNCH("vg_fill_h");
  return VG_OK;
}

extern "C" int vg_cast_f32_to_bf16(const float* in, void* out, size_t n, void* stream) {
  VG_REQUIRE(in && out && n > 0, "vg_cast_f32_to_bf16: bad arguments");
  const int vec = vg_aligned16(in) && vg_aligned16(out) ? 1 : 0;
  hipLaunchKernelGGL(narrow_kernel, dim3(vg_stream_grid((n + 7) / 8, 256)), dim3(256), 0, (hipStream_t)stream, in, (__bf16*)out, n, vec);
  VG_CHECK_LAUNCH("vg_cast_f32_to_bf16");
  return VG_OK;
}

extern "C" int vg_cast_bf16_to_f32(const void* in, float* out, size_t n, void* stream) {
  VG_REQUIRE(in && out && n > 0, "vg_cast_bf16_to_f32: bad arguments");
  const int vec = vg_aligned16(in) && vg_aligned16(out) ? 1 : 0;
  hipLaunchKernelGGL(widen_kernel, dim3(vg_stream_grid((n + 7) / 8, 256)), dim3(256), 0, (hipStream_t)stream, (const __bf16*)in, out, n, vec);
  VG_CHECK_LAUNCH("vg_cast_bf16_to_f32");
  return VG_OK;
}

extern "C" int vg_scale(const float* a, float s, float* out, size_t n, void* stream) {
  VG_REQUIRE(a && out && n > 0, "vg_scale: bad arguments");
  hipLaunchKernelGGL(scale_kernel, dim3(vg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, a, s, out, n);
  VG_CHECK_LAUNCH("vg_scale");
  return VG_OK;
}

extern "C" int vg_fill(float* out, float v, size_t n, void* stream) {
  VG_REQUIRE(out && n > 0, "vg_fill: bad arguments");
  hipLaunchKernelGGL(fill_kernel, dim3(vg_stream_grid(n, 256)), dim3(256), 0, (hipStream_t)stream, out, v, n);
  VG_CHECK_LAUNCH("vg_fill");
  return VG_OK;
}

extern "C" int vg_scalar_combine(const float* a, float c0, const float* b, float c1, const float* c, float c2, float* out,
                                 float* ratio_b_over_a, void* stream) {
  VG_REQUIRE(a && out, "vg_scalar_combine: bad arguments");
  hipLaunchKernelGGL(scalar_combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, c0, b, c1, c, c2, out, ratio_b_over_a);
  VG_CHECK_LAUNCH("vg_scalar_combine");
  return VG_OK;
}
