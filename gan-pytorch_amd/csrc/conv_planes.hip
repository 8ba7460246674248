// Splitting f32 tensors into bf16 planes for the plane GEMM (pgemm.h), and the rules that decide which contractions it
// serves.
#include "pgemm.h"

long long vg_plane_stride(long long elements) { return (elements + 127) / 128 * 128; }

size_t vg_planes_bytes(long long elements, int np) { return (size_t)np * (size_t)vg_plane_stride(elements) * sizeof(__bf16); }

namespace {

// plane q + 1 holds the rounding residual of plane q (round to nearest even each time): the planes add up to the input
template <int NP>
__device__ __forceinline__ void split2(float x, float y, unsigned int (&out)[NP]) {
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const f32x2 v = {x, y};
    out[q] = __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2));
    if (q + 1 < NP) {
      x -= __builtin_bit_cast(float, out[q] << 16);
      y -= __builtin_bit_cast(float, out[q] & 0xffff0000u);
    }
  }
}

template <int NP>
__global__ void __launch_bounds__(256) split_planes_kernel(const float* __restrict__ in, __bf16* __restrict__ out,
                                                           long long n, long long pstride, int vec) {
  const long long i8 = ((long long)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i8 >= n) return;
  if (vec && i8 + 8 <= n) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(in + i8), b = *reinterpret_cast<const f32x4*>(in + i8 + 4);
    unsigned int d[4][NP];
    split2<NP>(a[0], a[1], d[0]);
    split2<NP>(a[2], a[3], d[1]);
    split2<NP>(b[0], b[1], d[2]);
    split2<NP>(b[2], b[3], d[3]);
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      const u32x4 v = {d[0][q], d[1][q], d[2][q], d[3][q]};
      *reinterpret_cast<u32x4*>(out + (size_t)q * pstride + i8) = v;
    }
  } else {
    for (long long i = i8; i < n && i < i8 + 8; ++i) {
      float x = in[i];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
        const __bf16 b = (__bf16)x;
        out[(size_t)q * pstride + i] = b;
        x -= (float)b;
      }
    }
  }
}

}  // namespace

int vg_split_planes(const float* in, long long elements, int np, void* planes, hipStream_t st) {
  const long long pstride = vg_plane_stride(elements);
  const unsigned blocks = (unsigned)((elements + 2047) / 2048);
  const int vec = vg_aligned16(in) ? 1 : 0;
  if (np == 3)
    hipLaunchKernelGGL((split_planes_kernel<3>), dim3(blocks), dim3(256), 0, st, in, (__bf16*)planes, elements, pstride, vec);
  else
    hipLaunchKernelGGL((split_planes_kernel<1>), dim3(blocks), dim3(256), 0, st, in, (__bf16*)planes, elements, pstride, vec);
  return 0;
}

// bf16 terms per value under a precision setting: 3 in the split mode, 1 (the rounded value) in the bf16 mode
int vg_planes_of(int precision) { return precision == VG_PREC_F32_SPLIT ? 3 : 1; }

// whether the bf16 mode also runs on pre-converted operands (default) or converts inside the contraction's loader
bool vg_bf16_uses_planes() {
  static const bool on = !(getenv("VG_BF16_PLANES") && atoi(getenv("VG_BF16_PLANES")) == 0);
  return on;
}

extern "C" size_t vg_split_planes_bytes(long long elements, int precision) {
  return elements > 0 && precision != VG_PREC_F32 ? vg_planes_bytes(elements, vg_planes_of(precision)) : 0;
}

extern "C" int vg_split_planes_f32(const float* in, long long elements, int precision, void* planes, void* stream) {
  VG_REQUIRE(in && planes && elements > 0 && ((uintptr_t)planes & 15) == 0 &&
                 (precision == VG_PREC_BF16 || precision == VG_PREC_F32_SPLIT),
             "vg_split_planes_f32: bad arguments");
  vg_split_planes(in, elements, vg_planes_of(precision), planes, (hipStream_t)stream);
  VG_CHECK_LAUNCH("vg_split_planes_f32");
  return VG_OK;
}

// Which contractions the plane GEMM serves: 16-byte vectors of 8 bf16 along every contiguous index, and a tile
// family of 64 / 128 columns. mode 0 fwd, 1 dgrad, 2 wgrad; the caller has already routed the thin and 1x1 special
// cases elsewhere.
int vg_planes_applicable(int mode, const vg_conv_geom* g) {
  if (g->Ci % 8 != 0 || g->Co % 8 != 0) return 0;
  // the kernel's index arithmetic uses 24-bit products: rows of images, taps times channels and the like stay below 2^24
  const long long lim = 1LL << 24;
  if ((long long)g->N * g->H >= lim || (long long)g->N * g->Ho >= lim || (long long)g->Co * g->kh * g->kw >= lim ||
      (long long)g->kh * g->W + g->kw >= lim || (long long)g->Ci >= lim || (long long)g->W >= lim)
    return 0;
  const int ncols = mode == 0 ? g->Co : (mode == 1 ? g->Ci : g->kh * g->kw * g->Ci);
  return ncols > 32 ? 1 : 0;
}

size_t vg_planes_operand_bytes(int mode, const vg_conv_geom* g, int np) {
  const long long x = (long long)g->N * g->H * g->W * g->Ci, y = (long long)g->N * g->Ho * g->Wo * g->Co,
                  w = (long long)g->Co * g->kh * g->kw * g->Ci;
  const long long a = mode == 0 ? x : y, b = mode == 2 ? x : w;
  return vg_planes_bytes(a, np) + vg_planes_bytes(b, np);
}
