// Probe of ds_read_b64_tr_b16 lane semantics: prints, for every lane, the four 16-bit values it receives from a
// [row][64] image whose element (row, col) holds row * 256 + col.   hipcc --offload-arch=gfx950 -o tr_probe tr_read_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short t[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) t[i] = (unsigned short)((i / 64) * 256 + (i % 64));
  __syncthreads();
  const int L = threadIdx.x, g = L >> 4, i = L & 15, q = i >> 2, p = i & 3;
  const int k0 = 8 * (g >> 1), nbase = 16 * (g & 1);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(&t[(k0 + q) * 64 + nbase + 4 * p]));
  for (int e = 0; e < 4; ++e) out[L * 4 + e] = (unsigned short)v[e];
}
int main() {
  unsigned short* d; unsigned short h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int L = 0; L < 64; ++L) {
    const int g = L >> 4, i = L & 15, k0 = 8 * (g >> 1), nbase = 16 * (g & 1);
    printf("lane %2d:", L);
    for (int e = 0; e < 4; ++e) {
      printf(" (%d,%d)", h[L * 4 + e] >> 8, h[L * 4 + e] & 255);
      if (h[L * 4 + e] != (k0 + e) * 256 + nbase + i) ++bad;
    }
    printf("\n");
  }
  printf("mismatches vs expectation (lane i of a 16-lane group gets column nbase+i of rows k0..k0+3): %d\n", bad);
  return 0;
}
