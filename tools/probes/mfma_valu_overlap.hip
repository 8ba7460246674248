// Do the matrix-core phase of one wave and the vector-ALU phase of another wave on the same SIMD overlap?
// Every wave alternates P times between a run of NM v_mfma_f32_32x32x2_f32 (64 cycles each) and a run of NV dependent-free
// v_fma_f32 (4 cycles each), sized so that both runs take about the same time. Launched with 1 and with 2 waves per SIMD
// (one block per compute unit, of 256 or 512 threads), the second wave of a SIMD either in step with the
// first (`offset` = 0: both in their matrix phase at the same time) or half a period behind it (`offset` = 1: it starts
// with a vector phase). If the units overlap, two waves in opposite phases take about as long as one wave; in step they
// take twice as long.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_valu_overlap.hip -o /tmp/mfma_valu_overlap && /tmp/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#ifndef VOP
#define VOP 0      // vector instruction of the vector phase: 0 v_pk_fma_f32 (what -O3 makes of fmaf), 1 v_fma_f32, 2 v_add_u32
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NV, int KIND = 0>
__global__ void __launch_bounds__(512) phases(const float* in, float* out, int periods, int offset, int half) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  float a = in[t & 1023], b = in[(t + 7) & 1023];
  f32x16 c0 = {0}, c1 = {0};
  typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
  bf16x8 ah, bh;
#pragma unroll
  for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)in[(t + i) & 1023]; bh[i] = (__bf16)in[(t + 3 * i) & 1023]; }
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = in[(t + i) & 1023];
  // which waves start with the vector phase: with `half` = 256 the second four waves of a 512-thread block (a block's
  // waves go round the four SIMDs, so waves w and w + 4 share one); with `half` = 64 every other wave
  const bool late = offset && ((int)threadIdx.x / half) % 2 == 1;
  auto vec = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NV / 8; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#if VOP == 0
        v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);        // plain -O3 packs pairs of these into v_pk_fma_f32
#elif VOP == 1
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#else
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
#endif
      }
  };
  if (late) vec();
  for (int p = 0; p < periods; ++p) {
    if constexpr (KIND == 2) {
      // the same instructions, one matrix instruction followed by its share of the vector ones (same wave)
#pragma unroll
      for (int k = 0; k < NM; ++k) {
        if (k & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        else c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NV / NM; ++i) {
#if VOP == 0
          v[i & 7] = __builtin_fmaf(v[i & 7], 1.0001f, 0.5f);
#elif VOP == 1
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(a), "v"(b));
#else
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i & 7]) : "v"(a));
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NM / 2; ++k) {
        if constexpr (KIND == 1) {
          c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, ah, c1, 0, 0, 0);
        } else {
          c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      vec();
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[t] = s;
}

template <int NM, int NV, int KIND = 0>
float run(const float* in, float* out, int blocks, int periods, int offset) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int threads = blocks > 256 ? 512 : 256;      // one block per compute unit, 4 or 8 waves
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((phases<NM, NV, KIND>), dim3(256), dim3(threads), 0, 0, in, out, periods, offset & 1, offset & 2 ? 64 : 256);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((phases<NM, NV, KIND>), dim3(256), dim3(threads), 0, 0, in, out, periods, offset & 1, offset & 2 ? 64 : 256);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 100.f;   // us per launch
}

int main() {
  float *in, *out;
  hipMalloc(&in, 1024 * sizeof(float));
  hipMalloc(&out, 512 * 256 * sizeof(float));
  float h[1024];
  for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (i % 97);
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  const int periods = 40;
  // 48 MFMAs = 3072 cycles; 768 FMAs = 3072 cycles
  printf("VOP = %d; per launch, %d periods of (48 MFMA 32x32x2 | 768 vector instructions):\n", VOP, periods);
  printf("  matrix only   1 wave/SIMD  %.1f us\n", run<48, 0>(in, out, 256, periods, 0));
  printf("  vector only   1 wave/SIMD  %.1f us\n", run<0, 768>(in, out, 256, periods, 0));
  printf("  both          1 wave/SIMD  %.1f us\n", run<48, 768>(in, out, 256, periods, 0));
  printf("  both          2 waves/SIMD in step        %.1f us\n", run<48, 768>(in, out, 512, periods, 0));
  printf("  both          2 waves/SIMD half a period apart (waves 4-7 late)  %.1f us\n", run<48, 768>(in, out, 512, periods, 1));
  printf("  both          2 waves/SIMD half a period apart (odd waves late)  %.1f us\n", run<48, 768>(in, out, 512, periods, 3));
  printf("  matrix only   2 waves/SIMD %.1f us\n", run<48, 0>(in, out, 512, periods, 0));
  printf("  vector only   2 waves/SIMD %.1f us\n", run<0, 768>(in, out, 512, periods, 0));
  printf("same wave, each MFMA followed by 16 v_fma:  1 wave/SIMD %.1f us, 2 waves/SIMD %.1f us\n",
         run<48, 768, 2>(in, out, 256, periods, 0), run<48, 768, 2>(in, out, 512, periods, 0));
  printf("same wave, each MFMA followed by n vector instructions, 1 wave/SIMD: n=1 %.1f  n=2 %.1f  n=4 %.1f  n=8 %.1f  n=12 %.1f us (matrix only %.1f)\n",
         run<48, 48, 2>(in, out, 256, periods, 0), run<48, 96, 2>(in, out, 256, periods, 0), run<48, 192, 2>(in, out, 256, periods, 0),
         run<48, 384, 2>(in, out, 256, periods, 0), run<48, 576, 2>(in, out, 256, periods, 0), run<48, 0>(in, out, 256, periods, 0));
  printf("the same with 2 waves/SIMD: n=1 %.1f  n=2 %.1f  n=4 %.1f  n=8 %.1f us (matrix only %.1f)\n",
         run<48, 48, 2>(in, out, 512, periods, 0), run<48, 96, 2>(in, out, 512, periods, 0), run<48, 192, 2>(in, out, 512, periods, 0),
         run<48, 384, 2>(in, out, 512, periods, 0), run<48, 0>(in, out, 512, periods, 0));
  // 96 bf16 MFMAs of 32x32x16 (32 cycles each on paper) = 3072 cycles
  printf("bf16 MFMA 32x32x16 (96 per period) | 768 v_fma:\n");
  printf("  matrix only   1 wave/SIMD  %.1f us\n", run<96, 0, 1>(in, out, 256, periods, 0));
  printf("  both          1 wave/SIMD  %.1f us\n", run<96, 768, 1>(in, out, 256, periods, 0));
  printf("  both          2 waves/SIMD in step        %.1f us\n", run<96, 768, 1>(in, out, 512, periods, 0));
  printf("  both          2 waves/SIMD half a period apart (waves 4-7 late)  %.1f us\n", run<96, 768, 1>(in, out, 512, periods, 1));
  printf("  both          2 waves/SIMD half a period apart (odd waves late)  %.1f us\n", run<96, 768, 1>(in, out, 512, periods, 3));
  return 0;
}
