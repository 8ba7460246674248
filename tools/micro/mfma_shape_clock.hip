// Which f32 MFMA shape sustains more FLOP/s on this chip under dense load: v_mfma_f32_32x32x2_f32 or
// v_mfma_f32_16x16x4_f32? Both retire 64 FLOP per cycle per SIMD on paper; the chip lowers its clock under dense matrix
// load, and the clock it holds can depend on the instruction shape. Bare loops, operands in registers (random data),
// accumulators independent, `waves` waves per SIMD on every CU, run for a couple of seconds each.
//   hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_shape_clock.hip -o /tmp/mfma_shape_clock && /tmp/mfma_shape_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ void __launch_bounds__(256) mfma_loop(const float* in, float* out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = in[(t * 8 + i) & 0xFFFFF]; b[i] = in[(t * 8 + 4 + i) & 0xFFFFF]; }
  if constexpr (SHAPE == 32) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + 1) & 3], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + 1) & 3], b[j], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(j + 2) & 3], b[(j + 3) & 3], c3, 0, 0, 0);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e] + c2[e] + c3[e];
    out[t] = s;
  } else {
    // the same FLOPs per loop trip: 16 MFMAs of 32x32x2 (2048 MAC each) = 32 MFMAs of 16x16x4 (1024 MAC each)
    f32x4 c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + j) & 3], b[(i >> 2) & 3], c[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[t] = s;
  }
}

template <int SHAPE>
double run(const float* in, float* out, int blocks, int iters, double seconds) {
  hipLaunchKernelGGL((mfma_loop<SHAPE>), dim3(blocks), dim3(256), 0, 0, in, out, iters);
  hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  int launches = 0;
  double el = 0;
  while (el < seconds) {
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((mfma_loop<SHAPE>), dim3(blocks), dim3(256), 0, 0, in, out, iters);
    hipDeviceSynchronize();
    launches += 4;
    el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const double flops = (double)launches * blocks * 4 /*waves*/ * iters * 16.0 * 2048.0 * 2.0;
  return flops / el / 1e12;
}

int main() {
  const int n = 1 << 20;
  float* h = (float*)malloc(n * sizeof(float));
  srand(1);
  for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
  float *in, *out;
  hipMalloc(&in, n * sizeof(float));
  hipMalloc(&out, 256 * 4 * 256 * 8 * sizeof(float));
  hipMemcpy(in, h, n * sizeof(float), hipMemcpyHostToDevice);
  for (int wavesPerSimd = 1; wavesPerSimd <= 4; wavesPerSimd *= 2) {
    const int blocks = 256 * wavesPerSimd;   // 256 CUs x (4 waves per block = one per SIMD) x wavesPerSimd
    for (int rep = 0; rep < 2; ++rep) {
      const double a = run<32>(in, out, blocks, 4000, 2.0);
      const double b = run<16>(in, out, blocks, 4000, 2.0);
      printf("waves/SIMD %d: 32x32x2 %.1f TFLOP/s   16x16x4 %.1f TFLOP/s   ratio %.3f\n", wavesPerSimd, a, b, b / a);
      fflush(stdout);
    }
  }
  return 0;
}
