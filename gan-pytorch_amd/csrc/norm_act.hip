// BatchNorm (training/eval), activations and per-channel reductions over NHWC activations viewed as
// x[M][C] (M = N*H*W).  All of these are HBM-bound: 16-byte accesses, consecutive lanes on consecutive
// addresses, and every cross-block reduction goes through per-block partials that a second kernel adds in
// block order, so results are bitwise reproducible (no float atomics).
#include "vg_internal.h"

namespace {

constexpr int kMaxRowBlocks = 512;

// A block of the partial kernel reduces a slab of rows x 64 channels (16 threads across the channels, 16 down the
// rows): the grid is (row slabs, channel chunks), so a short, wide activation (4096 x 512) still fills the chip.
constexpr int kChunkChannels = 64;

inline int row_blocks(int M) {
  // at least 128 rows per block (8 per thread); at most kMaxRowBlocks blocks
  int nb = (M + 127) / 128;
  if (nb > kMaxRowBlocks) nb = kMaxRowBlocks;
  if (nb < 1) nb = 1;
  return nb;
}

// ---- generic column reduction: partial[v][b][c] = sum over the rows of block b of f_v(row, c) -------------
// A thread adds its own rows (8 ... 32 terms) in f32; from there on - across the threads of a block, across the blocks,
// into the epilogue - the sums are carried in double precision and rounded to f32 once. The long part of such a sum
// (10^3 ... 10^5 rows per channel) therefore adds no rounding of its own, as in the reference's CPU kernels, which
// accumulate batch statistics and the BatchNorm backward sums in double: a gradient sum whose terms nearly cancel
// (sum g over a channel of the critic) otherwise loses three to four digits, and every gradient below that layer
// inherits the error through mean(g). The per-element work is unchanged, so the kernels run as fast as before.
template <int NV, int VEC, class F>
__global__ void __launch_bounds__(256) colreduce_partial(F f, int M, int C, int rows_per_block, double* partial) {
  __shared__ double red[NV * VEC * 256];
  const int CG = C / VEC;
  constexpr int CGB = kChunkChannels / VEC;          // channel groups per block
  const int cg_lo = blockIdx.y * CGB, cg_hi = min(cg_lo + CGB, CG);
  const int span = cg_hi - cg_lo;
  const int TX = span < 256 ? span : 256;
  const int TY = 256 / TX;
  const int tid = threadIdx.x;
  const int tx = tid % TX, ty = tid / TX;
  const int nblk = gridDim.x;
  const int row0 = blockIdx.x * rows_per_block;
  const int row1 = min(row0 + rows_per_block, M);
  for (int cg0 = cg_lo; cg0 < cg_hi; cg0 += TX) {
    const int cg = cg0 + tx;
    float acc[NV][VEC];
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[v][e] = 0.f;
    if (ty < TY && cg < cg_hi) {
      typename F::template Ch<VEC> ch;
      f.template channel<VEC>(cg * VEC, ch);
#pragma unroll 4
      for (int row = row0 + ty; row < row1; row += TY) f.template eval<VEC>((size_t)row * C + (size_t)cg * VEC, ch, acc);
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int e = 0; e < VEC; ++e) red[(v * VEC + e) * 256 + tid] = (double)acc[v][e];
    __syncthreads();
    // NV * VEC * TX independent sums over the TY row lanes, one per thread (each still added in row-lane order, so the
    // result does not depend on how the sums are dealt out): done by 16 threads this tail took as long as the loads
    for (int q = tid; q < NV * VEC * TX; q += 256) {
      const int ve = q / TX, qx = q - ve * TX;
      if (cg0 + qx < cg_hi) {
        double s = 0.0;
        for (int y = 0; y < TY; ++y) s += red[ve * 256 + y * TX + qx];
        const int v = ve / VEC, e = ve - v * VEC;
        partial[((size_t)v * nblk + blockIdx.x) * C + (cg0 + qx) * VEC + e] = s;
      }
    }
    __syncthreads();
  }
}

// out[v][c] = sum_b partial[v][b][c]; 16 channels x 16 block-lanes per workgroup, fixed summation order.
struct StoreSumsE {   // plain result: out[v][c]
  float* out;
  int C;
  template <int NV>
  __device__ void operator()(int c, const double (&s)[NV]) const {
#pragma unroll
    for (int v = 0; v < NV; ++v) out[(size_t)v * C + c] = (float)s[v];
  }
};

// 8 channels x 32 block-lanes per workgroup. The partials were written by other compute units and come from memory: a
// lane requests all of its (at most 16 per value: kMaxRowBlocks / 32) before it adds the first, in fixed order, so the
// kernel is one memory round trip and not a chain of them (16 lanes with four loads in flight took eight trips).
template <int NV, class E>
__global__ void __launch_bounds__(256) colreduce_final(const double* partial, int nblk, int C, E epilogue) {
  constexpr int COLS = 8, LANES = 32, PER = kMaxRowBlocks / LANES;
  constexpr int GROUP = NV <= 2 ? PER : 8;         // blocks per lane requested together (registers: 2 per double)
  __shared__ double red[NV * 256];
  const int cl = threadIdx.x % COLS, q = threadIdx.x / COLS;
  const int c = blockIdx.x * COLS + cl;
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.0;
  if (c < C) {
    for (int b0 = q; b0 < nblk; b0 += LANES * GROUP) {
      double t[NV][GROUP];
#pragma unroll
      for (int u = 0; u < GROUP; ++u) {
        const int b = b0 + LANES * u;
#pragma unroll
        for (int v = 0; v < NV; ++v) t[v][u] = b < nblk ? partial[((size_t)v * nblk + b) * C + c] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < GROUP; ++u)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] += t[v][u];
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) red[v * 256 + threadIdx.x] = acc[v];
  __syncthreads();
  if (q == 0 && c < C) {
    double tot[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const double* rr = red + v * 256 + cl;
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < LANES; ++j) s += rr[COLS * j];
      tot[v] = s;
    }
    epilogue.template operator()<NV>(c, tot);
  }
}

template <int NV, class F, class E>
int run_colreduce_with(const char* who, F f, int M, int C, bool vec_ok, E epilogue, void* ws, size_t ws_bytes,
                       hipStream_t st) {
  const int nb = row_blocks(M);
  const size_t need = (size_t)NV * nb * C * sizeof(double);
  if (!ws || ws_bytes < need || (reinterpret_cast<uintptr_t>(ws) & 7u))
    return vg_fail(VG_ERR_WORKSPACE, "%s: workspace of %zu bytes (8-byte aligned) required, got %zu", who, need, ws_bytes);
  const int rpb = (M + nb - 1) / nb;
  double* partial = (double*)ws;
  const dim3 grid(nb, (C + kChunkChannels - 1) / kChunkChannels);
  if (vec_ok && C % 4 == 0)
    hipLaunchKernelGGL((colreduce_partial<NV, 4, F>), grid, dim3(256), 0, st, f, M, C, rpb, partial);
  else
    hipLaunchKernelGGL((colreduce_partial<NV, 1, F>), grid, dim3(256), 0, st, f, M, C, rpb, partial);
  VG_CHECK_LAUNCH(who);
  hipLaunchKernelGGL((colreduce_final<NV, E>), dim3((C + 7) / 8), dim3(256), 0, st, (const double*)partial, nb, C, epilogue);
  VG_CHECK_LAUNCH(who);
  return VG_OK;
}

template <int NV, class F>
int run_colreduce(const char* who, F f, int M, int C, bool vec_ok, float* out, void* ws, size_t ws_bytes,
                  hipStream_t st) {
  return run_colreduce_with<NV>(who, f, M, C, vec_ok, StoreSumsE{out, C}, ws, ws_bytes, st);
}

template <int VEC>
__device__ __forceinline__ void ldv(const float* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(p);
    v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
  } else {
    v[0] = *p;
  }
}
template <int VEC>
__device__ __forceinline__ void stv(float* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    f32x4 t = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p) = t;
  } else {
    *p = v[0];
  }
}

// bf16 storage: four values are one 8-byte access; widening is a shift, narrowing rounds to nearest even
// (v_cvt_pk_bf16_f32).
template <int VEC>
__device__ __forceinline__ void ldv(const __bf16* p, float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const u32x2 t = *reinterpret_cast<const u32x2*>(p);
    v[0] = __builtin_bit_cast(float, t[0] << 16);
    v[1] = __builtin_bit_cast(float, t[0] & 0xffff0000u);
    v[2] = __builtin_bit_cast(float, t[1] << 16);
    v[3] = __builtin_bit_cast(float, t[1] & 0xffff0000u);
  } else {
    v[0] = (float)*p;
  }
}
template <int VEC>
__device__ __forceinline__ void stv(__bf16* p, const float (&v)[VEC]) {
  if constexpr (VEC == 4) {
    const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    const u32x2 t = {__builtin_bit_cast(unsigned int, __builtin_convertvector(a, bf16x2)),
                     __builtin_bit_cast(unsigned int, __builtin_convertvector(b, bf16x2))};
    *reinterpret_cast<u32x2*>(p) = t;
  } else {
    *p = (__bf16)v[0];
  }
}

// ---- elementwise -----------------------------------------------------------------------------------
__device__ __forceinline__ float act_apply(float v, int act, float slope) {
  switch (act) {
    case VG_ACT_RELU: return v > 0.f ? v : 0.f;
    case VG_ACT_LRELU: return v > 0.f ? v : v * slope;
    case VG_ACT_TANH: return tanhf(v);
    case VG_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
    default: return v;
  }
}

// gx factor; `ref` = forward input (relu/lrelu) or forward output (tanh/sigmoid)
__device__ __forceinline__ float act_grad(float ref, int act, float slope) {
  switch (act) {
    case VG_ACT_RELU: return ref > 0.f ? 1.f : 0.f;
    case VG_ACT_LRELU: return ref > 0.f ? 1.f : slope;
    case VG_ACT_TANH: return 1.f - ref * ref;
    case VG_ACT_SIGMOID: return ref * (1.f - ref);
    default: return 1.f;
  }
}

// The slope mask of a ReLU / LeakyReLU (the only activations fused behind a norm: check_mask_act) without the switch:
// inside a per-element loop the switch is a chain of scalar branches per element.
__device__ __forceinline__ float mask_low(int act, float slope) { return act == VG_ACT_RELU ? 0.f : slope; }
__device__ __forceinline__ float mask_of(float ref, float low) { return ref > 0.f ? 1.f : low; }

// The channel group of element group i of an [M][C] tensor walked with a grid stride, without a division per step.
// When the stride is a multiple of the groups per row (every power-of-two channel count), a thread stays on one
// channel group and loads its per-channel values once: on bf16 tensors the four to seven 16-byte parameter loads per
// 8 bytes of data, not HBM, bounded these kernels.
struct ChannelWalk {
  int cg, step, CG;
  __device__ ChannelWalk(size_t i0, size_t stride, int CG_) : cg((int)(i0 % (size_t)CG_)), step((int)(stride % (size_t)CG_)), CG(CG_) {}
  __device__ bool fixed() const { return step == 0; }
  __device__ void next() {
    cg += step;
    if (cg >= CG) cg -= CG;
  }
};

// ---- functors -------------------------------------------------------------------------------------
// A functor of the column reduction has a per-channel part, fetched once per thread before its rows (channel()), and a
// per-row part (eval()). Everything eval() loads, it loads unconditionally: with the per-channel loads and the
// activation-dependent branches inside the row loop, the compiler put a wait for ALL outstanding loads into every
// row, so a thread's rows went to memory one after the other instead of four at a time.
template <class T>
struct SumF {  // colsum
  const T* x;
  template <int VEC>
  struct Ch {};
  template <int VEC>
  __device__ void channel(int c, Ch<VEC>& ch) const {}
  template <int VEC>
  __device__ void eval(size_t off, const Ch<VEC>& ch, float (&acc)[1][VEC]) const {
    float a[VEC];
    ldv<VEC>(x + off, a);
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[0][e] += a[e];
  }
};

template <class T>
struct ShiftedMomentsF {  // sums of (x - x[0][c]) and its square
  const T* x;
  template <int VEC>
  struct Ch {
    float s[VEC];
  };
  template <int VEC>
  __device__ void channel(int c, Ch<VEC>& ch) const {
    ldv<VEC>(x + c, ch.s);
  }
  template <int VEC>
  __device__ void eval(size_t off, const Ch<VEC>& ch, float (&acc)[2][VEC]) const {
    float a[VEC];
    ldv<VEC>(x + off, a);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float d = a[e] - ch.s[e];
      acc[0][e] += d;
      acc[1][e] = fmaf(d, d, acc[1][e]);
    }
  }
};

// Where the slope mask of an activation fused behind the norm comes from: nowhere (no activation), the activation's
// output, or the pre-activation recomputed from the norm's input (bit for bit what the forward pass saw).
enum { MASK_NONE = 0, MASK_ACT_OUT = 1, MASK_FROM_X = 2 };
static inline int mask_source(const void* act_out, int act) {
  return act_out ? MASK_ACT_OUT : (act != VG_ACT_NONE ? MASK_FROM_X : MASK_NONE);
}

template <int VEC>
struct BnChannel {
  float mu[VEC], r[VEC], ga[VEC], be[VEC];
};

template <class T, int MASK>
struct BnBwdF {  // sum g, sum g*xhat with g = gy * act'(.) when an activation is fused behind the norm
  const T *gy, *x;
  const float *mean, *invstd;
  const T* act_out;
  const float *gamma, *beta;
  int act;
  float slope;
  template <int VEC>
  using Ch = BnChannel<VEC>;
  template <int VEC>
  __device__ void channel(int c, Ch<VEC>& ch) const {
    ldv<VEC>(mean + c, ch.mu);
    ldv<VEC>(invstd + c, ch.r);
    if constexpr (MASK == MASK_FROM_X) {
      if (gamma) ldv<VEC>(gamma + c, ch.ga);
      if (beta) ldv<VEC>(beta + c, ch.be);
    }
  }
  template <int VEC>
  __device__ void eval(size_t off, const Ch<VEC>& ch, float (&acc)[2][VEC]) const {
    float g[VEC], a[VEC];
    ldv<VEC>(gy + off, g);
    ldv<VEC>(x + off, a);
    const float low = mask_low(act, slope);
    if constexpr (MASK == MASK_ACT_OUT) {
      float y[VEC];
      ldv<VEC>(act_out + off, y);
#pragma unroll
      for (int e = 0; e < VEC; ++e) g[e] *= mask_of(y[e], low);
    } else if constexpr (MASK == MASK_FROM_X) {
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        g[e] *= mask_of(vg_bn_affine(a[e], ch.mu[e], ch.r[e], gamma, ch.ga[e], beta, ch.be[e]), low);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float xh = (a[e] - ch.mu[e]) * ch.r[e];
      acc[0][e] += g[e];
      acc[1][e] = fmaf(g[e], xh, acc[1][e]);
    }
  }
};

template <class T, int MASK>
struct BnDbwdF {  // sum g, sum g*xhat, sum u, sum u*xhat, sum u*g   (g as in BnBwdF)
  const T *gy, *x, *u;
  const float *mean, *invstd;
  const T* act_out;
  const float *gamma, *beta;
  int act;
  float slope;
  template <int VEC>
  using Ch = BnChannel<VEC>;
  template <int VEC>
  __device__ void channel(int c, Ch<VEC>& ch) const {
    ldv<VEC>(mean + c, ch.mu);
    ldv<VEC>(invstd + c, ch.r);
    if constexpr (MASK == MASK_FROM_X) {
      if (gamma) ldv<VEC>(gamma + c, ch.ga);
      if (beta) ldv<VEC>(beta + c, ch.be);
    }
  }
  template <int VEC>
  __device__ void eval(size_t off, const Ch<VEC>& ch, float (&acc)[5][VEC]) const {
    float g[VEC], a[VEC], uu[VEC];
    ldv<VEC>(gy + off, g);
    ldv<VEC>(x + off, a);
    ldv<VEC>(u + off, uu);
    const float low = mask_low(act, slope);
    if constexpr (MASK == MASK_ACT_OUT) {
      float y[VEC];
      ldv<VEC>(act_out + off, y);
#pragma unroll
      for (int e = 0; e < VEC; ++e) g[e] *= mask_of(y[e], low);
    } else if constexpr (MASK == MASK_FROM_X) {
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        g[e] *= mask_of(vg_bn_affine(a[e], ch.mu[e], ch.r[e], gamma, ch.ga[e], beta, ch.be[e]), low);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float xh = (a[e] - ch.mu[e]) * ch.r[e];
      acc[0][e] += g[e];
      acc[1][e] = fmaf(g[e], xh, acc[1][e]);
      acc[2][e] += uu[e];
      acc[3][e] = fmaf(uu[e], xh, acc[3][e]);
      acc[4][e] = fmaf(uu[e], g[e], acc[4][e]);
    }
  }
};

// ---- BatchNorm statistics -------------------------------------------------------------------------
// Epilogue of the moments reduction: batch mean / inverse std from the shifted sums, running-statistics update and the
// batch counter, all in the kernel that finishes the reduction.
struct BnFinalizeE {
  const float* x;        // first row: the shift of the one-pass moments (f32 storage)
  const __bf16* xh;      // the same for bf16 storage (exactly one of the two is set by vg_bn_stats)
  int M;
  float eps, momentum;
  float *mean, *invstd, *running_mean, *running_var;
  long long* num_batches_tracked;
  template <int NV>
  __device__ void operator()(int c, const double (&s)[NV]) const {
    const double n = (double)M;
    const double s1 = s[0] / n;
    const double shift = xh ? (double)(float)xh[c] : (double)x[c];
    finish(c, shift + s1, s[1] / n - s1 * s1);
  }
  // mean and biased variance of column c (in double) -> the norm's statistics and running statistics
  __device__ void finish(int c, double mean_d, double var) const {
    const double n = (double)M;
    var = var < 0.0 ? 0.0 : var;
    const float mu = (float)mean_d;
    const float varf = (float)var;
    mean[c] = mu;
    invstd[c] = 1.0f / sqrtf(varf + eps);
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) {
      const float unbiased = M > 1 ? (float)(var * (n / (n - 1.0))) : varf;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * unbiased;
    }
    if (num_batches_tracked && c == 0) *num_batches_tracked += 1;
  }
};

// Batch statistics from the per-tile column statistics a convolution's epilogue left (igemm.h, STATS): tile t holds,
// per column, a shift s_t, S1_t = sum (y - s_t) and S2_t = sum (y - s_t)^2 over its n_t rows. Against the reference
// shift s_0 of the first tile: sum (y - s_0) = sum_t [S1_t + n_t d_t] and sum (y - s_0)^2 = sum_t [S2_t + 2 d_t S1_t +
// n_t d_t^2] with d_t = s_t - s_0, added in double (s_0 is a value of the column, so the two sums are of the size of
// the spread and their combination into the variance does not cancel). One pass, 4 columns x 256 tile lanes per block,
// four tiles in flight per lane, lanes added in a fixed tree. Same epilogue as the moments reduction (mean, invstd,
// running statistics, batch counter).
__global__ void __launch_bounds__(1024) bn_tiles_final_kernel(const double* __restrict__ stats, int tiles, int C, double n_rows,
                                                              BnFinalizeE fin) {
  constexpr int COLS = 4, LANES = 1024 / COLS;      // few columns per block: a 64-channel layer still gets 16 blocks
  __shared__ double red[2][1024];
  const int cl = threadIdx.x % COLS, q = threadIdx.x / COLS;
  const int c = blockIdx.x * COLS + cl;
  const double* counts = stats + (size_t)tiles * 3 * C;
  double a = 0.0, b = 0.0, s0 = 0.0;
  if (c < C) {
    s0 = stats[c];
    int t = q;
    for (; t + 3 * LANES < tiles; t += 4 * LANES) {
      double n[4], s[4], s1[4], s2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t base = ((size_t)(t + LANES * u) * 3) * C + c;
        n[u] = counts[t + LANES * u];
        s[u] = stats[base];
        s1[u] = stats[base + C];
        s2[u] = stats[base + 2 * (size_t)C];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (n[u] > 0.0) {
          const double d = s[u] - s0;
          a += s1[u] + n[u] * d;
          b += s2[u] + 2.0 * d * s1[u] + n[u] * d * d;
        }
      }
    }
    for (; t < tiles; t += LANES) {
      const double n = counts[t];
      if (n > 0.0) {
        const size_t base = ((size_t)t * 3) * C + c;
        const double d = stats[base] - s0, s1 = stats[base + C];
        a += s1 + n * d;
        b += stats[base + 2 * (size_t)C] + 2.0 * d * s1 + n * d * d;
      }
    }
  }
  // lanes of a column added pairwise in a fixed tree (LDS), then the column's result
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  __syncthreads();
  for (int half = LANES / 2; half > 0; half >>= 1) {
    if (q < half) {
      red[0][threadIdx.x] += red[0][threadIdx.x + half * COLS];
      red[1][threadIdx.x] += red[1][threadIdx.x + half * COLS];
    }
    __syncthreads();
  }
  if (q == 0 && c < C) {
    const double m1 = red[0][cl] / n_rows;
    fin.finish(c, s0 + m1, red[1][cl] / n_rows - m1 * m1);
  }
}

// The two column sums of a BatchNorm's backward pass from the per-tile sums a convolution's epilogue left (igemm.h,
// STATS = 2): [tiles][2][C] doubles, then [tiles] row counts (0 = the tile wrote nothing). Same block shape and fixed
// tree as above; the result leaves as f32 [2][C] (sum g m, sum g m xhat), what vg_bn_bwd_reduce leaves.
__global__ void __launch_bounds__(1024) bn_bwd_tiles_final_kernel(const double* __restrict__ part, int tiles, int C,
                                                                  float* __restrict__ sums) {
  constexpr int COLS = 4, LANES = 1024 / COLS;
  __shared__ double red[2][1024];
  const int cl = threadIdx.x % COLS, q = threadIdx.x / COLS;
  const int c = blockIdx.x * COLS + cl;
  const double* counts = part + (size_t)tiles * 2 * C;
  double a = 0.0, b = 0.0;
  if (c < C) {
    int t = q;
    for (; t + 3 * LANES < tiles; t += 4 * LANES) {
      double n[4], s1[4], s2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const size_t base = ((size_t)(t + LANES * u) * 2) * C + c;
        n[u] = counts[t + LANES * u];
        s1[u] = part[base];
        s2[u] = part[base + C];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (n[u] > 0.0) {
          a += s1[u];
          b += s2[u];
        }
      }
    }
    for (; t < tiles; t += LANES) {
      if (counts[t] > 0.0) {
        const size_t base = ((size_t)t * 2) * C + c;
        a += part[base];
        b += part[base + C];
      }
    }
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  __syncthreads();
  for (int half = LANES / 2; half > 0; half >>= 1) {
    if (q < half) {
      red[0][threadIdx.x] += red[0][threadIdx.x + half * COLS];
      red[1][threadIdx.x] += red[1][threadIdx.x + half * COLS];
    }
    __syncthreads();
  }
  if (q == 0 && c < C) {
    sums[c] = (float)red[0][cl];
    sums[C + c] = (float)red[1][cl];
  }
}

__global__ void __launch_bounds__(256) invstd_kernel(const float* var, float* invstd, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) invstd[c] = 1.0f / sqrtf(var[c] + eps);
}

template <int VEC, class T>
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, T* __restrict__ y, size_t nvec,
                                                       int C, int act, float slope) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvec) return;
  ChannelWalk cw(i, stride, C / VEC);
  float mu[VEC], r[VEC], ga[VEC], be[VEC];
  auto load_channel = [&]() {
    const int c = cw.cg * VEC;
    ldv<VEC>(mean + c, mu);
    ldv<VEC>(invstd + c, r);
    if (gamma) ldv<VEC>(gamma + c, ga);
    if (beta) ldv<VEC>(beta + c, be);
  };
  load_channel();
  const bool simple = act == VG_ACT_NONE || act == VG_ACT_RELU || act == VG_ACT_LRELU;
  const bool none = act == VG_ACT_NONE, relu = act == VG_ACT_RELU;
  for (; i < nvec; i += stride) {
    if (!cw.fixed()) load_channel();
    float a[VEC], o[VEC];
    ldv<VEC>(x + i * VEC, a);
    if (simple) {         // no switch per element
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float v = vg_bn_affine(a[e], mu[e], r[e], gamma, ga[e], beta, be[e]);
        const float neg = relu ? 0.f : v * slope;
        o[e] = (none || v > 0.f) ? v : neg;
      }
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = act_apply(vg_bn_affine(a[e], mu[e], r[e], gamma, ga[e], beta, be[e]), act, slope);
    }
    stv<VEC>(y + i * VEC, o);
    if (!cw.fixed()) cw.next();
  }
}

template <int VEC, class T>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ gy, const T* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ sums,
                                                           T* __restrict__ gx, size_t nvec, int C, float inv_n,
                                                           const T* __restrict__ act_out, int act, float slope,
                                                           const float* __restrict__ beta) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvec) return;
  ChannelWalk cw(i, stride, C / VEC);
  float mu[VEC], r[VEC], ga[VEC], be[VEC], sg[VEC], sgx[VEC];
  const bool from_x = !act_out && act != VG_ACT_NONE;      // the mask is rebuilt from the norm's input
  auto load_channel = [&]() {
    const int c = cw.cg * VEC;
    ldv<VEC>(mean + c, mu);
    ldv<VEC>(invstd + c, r);
    if (gamma) ldv<VEC>(gamma + c, ga);
    if (from_x && beta) ldv<VEC>(beta + c, be);
    ldv<VEC>(sums + c, sg);
    ldv<VEC>(sums + C + c, sgx);
  };
  load_channel();
  const float low = mask_low(act, slope);
  for (; i < nvec; i += stride) {
    if (!cw.fixed()) load_channel();
    float g[VEC], a[VEC], o[VEC];
    ldv<VEC>(gy + i * VEC, g);
    ldv<VEC>(x + i * VEC, a);
    if (act_out) {
      float y[VEC];
      ldv<VEC>(act_out + i * VEC, y);
#pragma unroll
      for (int e = 0; e < VEC; ++e) g[e] *= mask_of(y[e], low);
    } else if (from_x) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) g[e] *= mask_of(vg_bn_affine(a[e], mu[e], r[e], gamma, ga[e], beta, be[e]), low);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float xh = (a[e] - mu[e]) * r[e];
      const float sc = gamma ? r[e] * ga[e] : r[e];
      o[e] = sc * (g[e] - sg[e] * inv_n - xh * (sgx[e] * inv_n));
    }
    stv<VEC>(gx + i * VEC, o);
    if (!cw.fixed()) cw.next();
  }
}

template <int VEC, class T>
__global__ void __launch_bounds__(256) bn_bwd_eval_kernel(const T* __restrict__ gy, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, T* __restrict__ gx,
                                                          size_t nvec, int C) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvec) return;
  ChannelWalk cw(i, stride, C / VEC);
  float r[VEC], ga[VEC];
  auto load_channel = [&]() {
    const int c = cw.cg * VEC;
    ldv<VEC>(invstd + c, r);
    if (gamma) ldv<VEC>(gamma + c, ga);
  };
  load_channel();
  for (; i < nvec; i += stride) {
    if (!cw.fixed()) load_channel();
    float g[VEC], o[VEC];
    ldv<VEC>(gy + i * VEC, g);
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = g[e] * (gamma ? r[e] * ga[e] : r[e]);
    stv<VEC>(gx + i * VEC, o);
    if (!cw.fixed()) cw.next();
  }
}

// second derivative of the training-mode backward (see vegans_hip.h for the meaning of u, v, w)
template <int VEC, class T>
__global__ void __launch_bounds__(256) bn_dbwd_apply_kernel(
    const T* __restrict__ gy, const T* __restrict__ x, const T* __restrict__ u, const float* __restrict__ v,
    const float* __restrict__ w, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ gamma, const float* __restrict__ sums5, T* __restrict__ d_gy, T* __restrict__ d_x,
    size_t nvec, int C, float inv_n, const T* __restrict__ act_out, int act, float slope,
    const float* __restrict__ beta) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvec) return;
  ChannelWalk cw(i, stride, C / VEC);
  float mu[VEC], r[VEC], ga[VEC], be[VEC], vv[VEC], ww[VEC], s0[VEC], s1[VEC], s2[VEC], s3[VEC], s4[VEC];
  const bool from_x = !act_out && act != VG_ACT_NONE;
  auto load_channel = [&]() {
    const int c = cw.cg * VEC;
    ldv<VEC>(mean + c, mu);
    ldv<VEC>(invstd + c, r);
    if (gamma) ldv<VEC>(gamma + c, ga);
    if (from_x && beta) ldv<VEC>(beta + c, be);
    if (v) ldv<VEC>(v + c, vv);
    if (w) ldv<VEC>(w + c, ww);
    ldv<VEC>(sums5 + c, s0);
    ldv<VEC>(sums5 + C + c, s1);
    ldv<VEC>(sums5 + 2 * C + c, s2);
    ldv<VEC>(sums5 + 3 * C + c, s3);
    ldv<VEC>(sums5 + 4 * C + c, s4);
  };
  load_channel();
  const float low = mask_low(act, slope);
  for (; i < nvec; i += stride) {
    if (!cw.fixed()) load_channel();
    float g[VEC], a[VEC], uu[VEC], o1[VEC], o2[VEC], mk[VEC];
    ldv<VEC>(gy + i * VEC, g);
    ldv<VEC>(x + i * VEC, a);
    ldv<VEC>(u + i * VEC, uu);
#pragma unroll
    for (int e = 0; e < VEC; ++e) mk[e] = 1.f;
    if (act_out) {
      float y[VEC];
      ldv<VEC>(act_out + i * VEC, y);
#pragma unroll
      for (int e = 0; e < VEC; ++e) mk[e] = mask_of(y[e], low);
    } else if (from_x) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) mk[e] = mask_of(vg_bn_affine(a[e], mu[e], r[e], gamma, ga[e], beta, be[e]), low);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) g[e] *= mk[e];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float xh = (a[e] - mu[e]) * r[e];
      const float gam = gamma ? ga[e] : 1.f;
      const float gbar = s0[e] * inv_n, m = s1[e] * inv_n, ubar = s2[e] * inv_n, q = s3[e] * inv_n;
      const float P = s4[e] * inv_n - ubar * gbar - q * m;
      const float du = uu[e] - ubar, dg = g[e] - gbar;
      float ogy = gam * r[e] * (du - xh * q);
      float ox = -gam * r[e] * r[e] * (m * du + q * dg + xh * (P - 2.f * m * q));
      if (v) {
        ogy += vv[e] * xh;
        ox += vv[e] * r[e] * (dg - xh * m);
      }
      if (w) ogy += ww[e];
      o1[e] = ogy * mk[e];
      o2[e] = ox;
    }
    stv<VEC>(d_gy + i * VEC, o1);
    stv<VEC>(d_x + i * VEC, o2);
    if (!cw.fixed()) cw.next();
  }
}

__global__ void __launch_bounds__(256) bn_dbwd_gamma_kernel(const float* invstd, const float* sums5, float* d_gamma, int C,
                                                            float n) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float inv_n = 1.f / n;
  const float gbar = sums5[c] * inv_n, m = sums5[C + c] * inv_n, ubar = sums5[2 * C + c] * inv_n, q = sums5[3 * C + c] * inv_n;
  const float P = sums5[4 * C + c] * inv_n - ubar * gbar - q * m;
  d_gamma[c] = invstd[c] * n * P;
}

template <int VEC, class T>
__global__ void __launch_bounds__(256) act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, size_t nvec, int act,
                                                      float slope) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float a[VEC], o[VEC];
    ldv<VEC>(x + i * VEC, a);
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = act_apply(a[e], act, slope);
    stv<VEC>(y + i * VEC, o);
  }
}

template <int VEC, class T>
__global__ void __launch_bounds__(256) act_bwd_kernel(const T* __restrict__ gy, const T* __restrict__ ref,
                                                      T* __restrict__ gx, size_t nvec, int act, float slope) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float g[VEC], a[VEC], o[VEC];
    ldv<VEC>(gy + i * VEC, g);
    ldv<VEC>(ref + i * VEC, a);
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = g[e] * act_grad(a[e], act, slope);
    stv<VEC>(gx + i * VEC, o);
  }
}

template <int VEC, class T>
__global__ void __launch_bounds__(256) act_dbwd_ref_kernel(const T* __restrict__ gg, const T* __restrict__ gy,
                                                           const T* __restrict__ ref, T* __restrict__ d_ref,
                                                           size_t nvec, int act) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float a[VEC], b[VEC], y[VEC], o[VEC];
    ldv<VEC>(gg + i * VEC, a);
    ldv<VEC>(gy + i * VEC, b);
    ldv<VEC>(ref + i * VEC, y);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float f = act == VG_ACT_TANH ? -2.f * y[e] : (act == VG_ACT_SIGMOID ? 1.f - 2.f * y[e] : 0.f);
      o[e] = a[e] * b[e] * f;
    }
    stv<VEC>(d_ref + i * VEC, o);
  }
}

template <int VEC>
__global__ void __launch_bounds__(256) bcast_rows_kernel(const float* __restrict__ v, float* __restrict__ out, size_t nvec,
                                                         int C) {
  const int CG = C / VEC;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float a[VEC];
    ldv<VEC>(v + (int)(i % CG) * VEC, a);
    stv<VEC>(out + i * VEC, a);
  }
}

inline bool all_aligned(std::initializer_list<const void*> ps) {
  for (const void* p : ps)
    if (p && !vg_aligned16(p)) return false;
  return true;
}

}  // namespace

#define VG_LAUNCH_VEC(kernel, T_, vec4, total, st, ...)                                                      \
  do {                                                                                                       \
    if (vec4) {                                                                                              \
      const size_t nvec__ = (total) / 4;                                                                     \
      hipLaunchKernelGGL((kernel<4, T_>), dim3(vg_stream_grid(nvec__, 256)), dim3(256), 0, st, __VA_ARGS__); \
    } else {                                                                                                 \
      const size_t nvec__ = (total);                                                                         \
      hipLaunchKernelGGL((kernel<1, T_>), dim3(vg_stream_grid(nvec__, 256)), dim3(256), 0, st, __VA_ARGS__); \
    }                                                                                                        \
  } while (0)

extern "C" size_t vg_bn_workspace_bytes(int M, int C) {
  if (M <= 0 || C <= 0) return 0;
  return (size_t)5 * row_blocks(M) * C * sizeof(double) + (size_t)5 * C * sizeof(float);
}

extern "C" size_t vg_colsum_workspace_bytes(int M, int C) {
  if (M <= 0 || C <= 0) return 0;
  return (size_t)row_blocks(M) * C * sizeof(double);
}

// Every entry point below exists twice: for f32 activations (the un-suffixed names) and, with the suffix _h, for
// activations stored as bf16 (same arithmetic in f32 registers; per-channel vectors, sums and parameters stay f32).
namespace {

template <class T>
int colsum_impl(const char* who, const T* x, float* out, int M, int C, void* ws, size_t ws_bytes, void* stream) {
  VG_REQUIRE(x && out && M > 0 && C > 0, "%s: bad arguments", who);
  SumF<T> f{x};
  return run_colreduce<1>(who, f, M, C, all_aligned({x}), out, ws, ws_bytes, (hipStream_t)stream);
}

template <class T>
int bn_stats_impl(const char* who, const T* x, int M, int C, float eps, float momentum, float* mean, float* invstd,
                  float* running_mean, float* running_var, long long* num_batches_tracked, void* ws, size_t ws_bytes,
                  void* stream) {
  VG_REQUIRE(x && mean && invstd && M > 0 && C > 0, "%s: bad arguments", who);
  ShiftedMomentsF<T> f{x};
  BnFinalizeE fin{nullptr, nullptr, M, eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked};
  if constexpr (std::is_same<T, float>::value) fin.x = x;
  else fin.xh = x;
  return run_colreduce_with<2>(who, f, M, C, all_aligned({x}), fin, ws, ws_bytes, (hipStream_t)stream);
}

template <class T>
int bn_apply_impl(const char* who, const T* x, const float* mean, const float* invstd, const float* gamma,
                  const float* beta, T* y, int M, int C, int act, float slope, void* stream) {
  VG_REQUIRE(x && mean && invstd && y && M > 0 && C > 0, "%s: bad arguments", who);
  const bool v4 = C % 4 == 0 && all_aligned({x, mean, invstd, gamma, beta, y});
  const size_t total = (size_t)M * C;
  VG_LAUNCH_VEC(bn_apply_kernel, T, v4, total, (hipStream_t)stream, x, mean, invstd, gamma, beta, y, nvec__, C, act, slope);
  VG_CHECK_LAUNCH(who);
  return VG_OK;
}

int check_mask_act(const void* act_out, int act, const char* who) {
  if (act != VG_ACT_NONE && act != VG_ACT_RELU && act != VG_ACT_LRELU)
    return vg_fail(VG_ERR_INVALID, "%s: only relu / leaky relu can be fused behind the norm", who);
  if (act_out && act == VG_ACT_NONE)
    return vg_fail(VG_ERR_INVALID, "%s: an activation output was passed without an activation code", who);
  return 0;
}

template <class T>
int bn_bwd_reduce_impl(const char* who, const T* gy, const T* x, const float* mean, const float* invstd, int M, int C,
                       float* sums, void* ws, size_t ws_bytes, const T* act_out, int act, float slope,
                       const float* gamma, const float* beta, void* stream) {
  VG_REQUIRE(gy && x && mean && invstd && sums && M > 0 && C > 0, "%s: bad arguments", who);
  if (int e = check_mask_act(act_out, act, who)) return e;
  const bool vec_ok = all_aligned({gy, x, mean, invstd, act_out, gamma, beta});
  switch (mask_source(act_out, act)) {
    case MASK_ACT_OUT:
      return run_colreduce<2>(who, BnBwdF<T, MASK_ACT_OUT>{gy, x, mean, invstd, act_out, gamma, beta, act, slope}, M, C, vec_ok,
                              sums, ws, ws_bytes, (hipStream_t)stream);
    case MASK_FROM_X:
      return run_colreduce<2>(who, BnBwdF<T, MASK_FROM_X>{gy, x, mean, invstd, act_out, gamma, beta, act, slope}, M, C, vec_ok,
                              sums, ws, ws_bytes, (hipStream_t)stream);
    default:
      return run_colreduce<2>(who, BnBwdF<T, MASK_NONE>{gy, x, mean, invstd, act_out, gamma, beta, act, slope}, M, C, vec_ok,
                              sums, ws, ws_bytes, (hipStream_t)stream);
  }
}

template <class T>
int bn_bwd_apply_impl(const char* who, const T* gy, const T* x, const float* mean, const float* invstd,
                      const float* gamma, const float* sums, T* gx, int M, int C, const T* act_out, int act, float slope,
                      const float* beta, void* stream) {
  VG_REQUIRE(gy && x && mean && invstd && sums && gx && M > 0 && C > 0, "%s: bad arguments", who);
  if (int e = check_mask_act(act_out, act, who)) return e;
  const bool v4 = C % 4 == 0 && all_aligned({gy, x, mean, invstd, gamma, sums, gx, act_out, beta});
  const size_t total = (size_t)M * C;
  const float inv_n = 1.0f / (float)M;
  VG_LAUNCH_VEC(bn_bwd_apply_kernel, T, v4, total, (hipStream_t)stream, gy, x, mean, invstd, gamma, sums, gx, nvec__, C,
                inv_n, act_out, act, slope, beta);
  VG_CHECK_LAUNCH(who);
  return VG_OK;
}

template <class T>
int bn_bwd_eval_impl(const char* who, const T* gy, const float* invstd, const float* gamma, T* gx, int M, int C,
                     void* stream) {
  VG_REQUIRE(gy && invstd && gx && M > 0 && C > 0, "%s: bad arguments", who);
  const bool v4 = C % 4 == 0 && all_aligned({gy, invstd, gamma, gx});
  const size_t total = (size_t)M * C;
  VG_LAUNCH_VEC(bn_bwd_eval_kernel, T, v4, total, (hipStream_t)stream, gy, invstd, gamma, gx, nvec__, C);
  VG_CHECK_LAUNCH(who);
  return VG_OK;
}

template <class T>
int bn_dbwd_reduce_impl(const char* who, const T* gy, const T* x, const T* u, const float* mean, const float* invstd,
                        int M, int C, float* sums5, void* ws, size_t ws_bytes, const T* act_out, int act, float slope,
                        const float* gamma, const float* beta, void* stream) {
  VG_REQUIRE(gy && x && u && mean && invstd && sums5 && M > 0 && C > 0, "%s: bad arguments", who);
  if (int e = check_mask_act(act_out, act, who)) return e;
  const bool vec_ok = all_aligned({gy, x, u, mean, invstd, act_out, gamma, beta});
  switch (mask_source(act_out, act)) {
    case MASK_ACT_OUT:
      return run_colreduce<5>(who, BnDbwdF<T, MASK_ACT_OUT>{gy, x, u, mean, invstd, act_out, gamma, beta, act, slope}, M, C,
                              vec_ok, sums5, ws, ws_bytes, (hipStream_t)stream);
    case MASK_FROM_X:
      return run_colreduce<5>(who, BnDbwdF<T, MASK_FROM_X>{gy, x, u, mean, invstd, act_out, gamma, beta, act, slope}, M, C,
                              vec_ok, sums5, ws, ws_bytes, (hipStream_t)stream);
    default:
      return run_colreduce<5>(who, BnDbwdF<T, MASK_NONE>{gy, x, u, mean, invstd, act_out, gamma, beta, act, slope}, M, C,
                              vec_ok, sums5, ws, ws_bytes, (hipStream_t)stream);
  }
}

template <class T>
int bn_dbwd_apply_impl(const char* who, const T* gy, const T* x, const T* u, const float* v, const float* w,
                       const float* mean, const float* invstd, const float* gamma, const float* sums5, T* d_gy, T* d_x,
                       float* d_gamma, int M, int C, const T* act_out, int act, float slope, const float* beta,
                       void* stream) {
  VG_REQUIRE(gy && x && u && mean && invstd && sums5 && d_gy && d_x && M > 0 && C > 0, "%s: bad arguments", who);
  if (int e = check_mask_act(act_out, act, who)) return e;
  const bool v4 = C % 4 == 0 && all_aligned({gy, x, u, v, w, mean, invstd, gamma, sums5, d_gy, d_x, act_out, beta});
  const size_t total = (size_t)M * C;
  const float inv_n = 1.0f / (float)M;
  hipStream_t st = (hipStream_t)stream;
  VG_LAUNCH_VEC(bn_dbwd_apply_kernel, T, v4, total, st, gy, x, u, v, w, mean, invstd, gamma, sums5, d_gy, d_x, nvec__, C,
                inv_n, act_out, act, slope, beta);
  VG_CHECK_LAUNCH(who);
  if (d_gamma) {
    hipLaunchKernelGGL(bn_dbwd_gamma_kernel, dim3((C + 255) / 256), dim3(256), 0, st, invstd, sums5, d_gamma, C, (float)M);
    VG_CHECK_LAUNCH(who);
  }
  return VG_OK;
}

template <class T>
int act_fwd_impl(const char* who, const T* x, T* y, size_t n, int act, float slope, void* stream) {
  VG_REQUIRE(x && y && n > 0, "%s: bad arguments", who);
  const bool v4 = n % 4 == 0 && all_aligned({x, y});
  VG_LAUNCH_VEC(act_fwd_kernel, T, v4, n, (hipStream_t)stream, x, y, nvec__, act, slope);
  VG_CHECK_LAUNCH(who);
  return VG_OK;
}

template <class T>
int act_bwd_impl(const char* who, const T* gy, const T* ref, T* gx, size_t n, int act, float slope, void* stream) {
  VG_REQUIRE(gy && ref && gx && n > 0, "%s: bad arguments", who);
  const bool v4 = n % 4 == 0 && all_aligned({gy, ref, gx});
  VG_LAUNCH_VEC(act_bwd_kernel, T, v4, n, (hipStream_t)stream, gy, ref, gx, nvec__, act, slope);
  VG_CHECK_LAUNCH(who);
  return VG_OK;
}

template <class T>
int act_dbwd_ref_impl(const char* who, const T* gg, const T* gy, const T* ref, T* d_ref, size_t n, int act,
                      void* stream) {
  VG_REQUIRE(gg && gy && ref && d_ref && n > 0, "%s: bad arguments", who);
  VG_REQUIRE(act == VG_ACT_TANH || act == VG_ACT_SIGMOID, "%s: only tanh/sigmoid have a second derivative", who);
  const bool v4 = n % 4 == 0 && all_aligned({gg, gy, ref, d_ref});
  VG_LAUNCH_VEC(act_dbwd_ref_kernel, T, v4, n, (hipStream_t)stream, gg, gy, ref, d_ref, nvec__, act);
  VG_CHECK_LAUNCH(who);
  return VG_OK;
}

typedef const __bf16* HC;   // bf16 tensors cross the C ABI as void*
typedef __bf16* HM;

}  // namespace

extern "C" int vg_colsum(const float* x, float* out, int M, int C, void* ws, size_t ws_bytes, void* stream) {
  return colsum_impl<float>("vg_colsum", x, out, M, C, ws, ws_bytes, stream);
}
extern "C" int vg_colsum_h(const void* x, float* out, int M, int C, void* ws, size_t ws_bytes, void* stream) {
  return colsum_impl<__bf16>("vg_colsum_h", (HC)x, out, M, C, ws, ws_bytes, stream);
}

extern "C" int vg_bn_stats(const float* x, int M, int C, float eps, float momentum, float* mean, float* invstd,
                           float* running_mean, float* running_var, long long* num_batches_tracked, void* ws,
                           size_t ws_bytes, void* stream) {
  return bn_stats_impl<float>("vg_bn_stats", x, M, C, eps, momentum, mean, invstd, running_mean, running_var,
                              num_batches_tracked, ws, ws_bytes, stream);
}
extern "C" int vg_bn_stats_h(const void* x, int M, int C, float eps, float momentum, float* mean, float* invstd,
                             float* running_mean, float* running_var, long long* num_batches_tracked, void* ws,
                             size_t ws_bytes, void* stream) {
  return bn_stats_impl<__bf16>("vg_bn_stats_h", (HC)x, M, C, eps, momentum, mean, invstd, running_mean, running_var,
                               num_batches_tracked, ws, ws_bytes, stream);
}

void vg_bnstats_layout(int mode, const vg_conv_geom* g, int* tiles, long long* rows, int* cols);

extern "C" int vg_bn_stats_from_conv(int mode, const vg_conv_geom* g, const void* stats, float eps, float momentum,
                                     float* mean, float* invstd, float* running_mean, float* running_var,
                                     long long* num_batches_tracked, void* stream) {
  VG_REQUIRE(g && stats && mean && invstd && (mode == 0 || mode == 1), "vg_bn_stats_from_conv: bad arguments");
  int tiles, C;
  long long rows;
  vg_bnstats_layout(mode, g, &tiles, &rows, &C);
  VG_REQUIRE(rows > 0 && rows < (1LL << 31), "vg_bn_stats_from_conv: bad geometry");
  BnFinalizeE fin{nullptr, nullptr, (int)rows, eps, momentum, mean, invstd, running_mean, running_var, num_batches_tracked};
  hipLaunchKernelGGL(bn_tiles_final_kernel, dim3((C + 3) / 4), dim3(1024), 0, (hipStream_t)stream, (const double*)stats, tiles,
                     C, (double)rows, fin);
  VG_CHECK_LAUNCH("vg_bn_stats_from_conv");
  return VG_OK;
}

extern "C" int vg_bn_bwd_sums_from_conv(int mode, const vg_conv_geom* g, const void* tile_sums, float* sums, void* stream) {
  VG_REQUIRE(g && tile_sums && sums && (mode == 0 || mode == 1), "vg_bn_bwd_sums_from_conv: bad arguments");
  int tiles, C;
  long long rows;
  vg_bnstats_layout(mode, g, &tiles, &rows, &C);
  hipLaunchKernelGGL(bn_bwd_tiles_final_kernel, dim3((C + 3) / 4), dim3(1024), 0, (hipStream_t)stream,
                     (const double*)tile_sums, tiles, C, sums);
  VG_CHECK_LAUNCH("vg_bn_bwd_sums_from_conv");
  return VG_OK;
}

extern "C" int vg_bn_invstd_from_var(const float* var, float* invstd, int C, float eps, void* stream) {
  VG_REQUIRE(var && invstd && C > 0, "vg_bn_invstd_from_var: bad arguments");
  hipLaunchKernelGGL(invstd_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, var, invstd, C, eps);
  VG_CHECK_LAUNCH("vg_bn_invstd_from_var");
  return VG_OK;
}

extern "C" int vg_bn_apply(const float* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           float* y, int M, int C, int act, float slope, void* stream) {
  return bn_apply_impl<float>("vg_bn_apply", x, mean, invstd, gamma, beta, y, M, C, act, slope, stream);
}
extern "C" int vg_bn_apply_h(const void* x, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             void* y, int M, int C, int act, float slope, void* stream) {
  return bn_apply_impl<__bf16>("vg_bn_apply_h", (HC)x, mean, invstd, gamma, beta, (HM)y, M, C, act, slope, stream);
}

extern "C" int vg_bn_bwd_reduce(const float* gy, const float* x, const float* mean, const float* invstd, int M, int C,
                                float* sums, void* ws, size_t ws_bytes, const float* act_out, int act, float slope,
                                const float* gamma, const float* beta, void* stream) {
  return bn_bwd_reduce_impl<float>("vg_bn_bwd_reduce", gy, x, mean, invstd, M, C, sums, ws, ws_bytes, act_out, act, slope,
                                   gamma, beta, stream);
}
extern "C" int vg_bn_bwd_reduce_h(const void* gy, const void* x, const float* mean, const float* invstd, int M, int C,
                                  float* sums, void* ws, size_t ws_bytes, const void* act_out, int act, float slope,
                                  const float* gamma, const float* beta, void* stream) {
  return bn_bwd_reduce_impl<__bf16>("vg_bn_bwd_reduce_h", (HC)gy, (HC)x, mean, invstd, M, C, sums, ws, ws_bytes,
                                    (HC)act_out, act, slope, gamma, beta, stream);
}

extern "C" int vg_bn_bwd_apply(const float* gy, const float* x, const float* mean, const float* invstd, const float* gamma,
                               const float* sums, float* gx, int M, int C, const float* act_out, int act, float slope,
                               const float* beta, void* stream) {
  return bn_bwd_apply_impl<float>("vg_bn_bwd_apply", gy, x, mean, invstd, gamma, sums, gx, M, C, act_out, act, slope, beta,
                                  stream);
}
extern "C" int vg_bn_bwd_apply_h(const void* gy, const void* x, const float* mean, const float* invstd, const float* gamma,
                                 const float* sums, void* gx, int M, int C, const void* act_out, int act, float slope,
                                 const float* beta, void* stream) {
  return bn_bwd_apply_impl<__bf16>("vg_bn_bwd_apply_h", (HC)gy, (HC)x, mean, invstd, gamma, sums, (HM)gx, M, C,
                                   (HC)act_out, act, slope, beta, stream);
}

extern "C" int vg_bn_bwd_eval(const float* gy, const float* invstd, const float* gamma, float* gx, int M, int C,
                              void* stream) {
  return bn_bwd_eval_impl<float>("vg_bn_bwd_eval", gy, invstd, gamma, gx, M, C, stream);
}
extern "C" int vg_bn_bwd_eval_h(const void* gy, const float* invstd, const float* gamma, void* gx, int M, int C,
                                void* stream) {
  return bn_bwd_eval_impl<__bf16>("vg_bn_bwd_eval_h", (HC)gy, invstd, gamma, (HM)gx, M, C, stream);
}

extern "C" int vg_bn_dbwd_reduce(const float* gy, const float* x, const float* u, const float* mean, const float* invstd,
                                 int M, int C, float* sums5, void* ws, size_t ws_bytes, const float* act_out, int act,
                                 float slope, const float* gamma, const float* beta, void* stream) {
  return bn_dbwd_reduce_impl<float>("vg_bn_dbwd_reduce", gy, x, u, mean, invstd, M, C, sums5, ws, ws_bytes, act_out, act,
                                    slope, gamma, beta, stream);
}
extern "C" int vg_bn_dbwd_reduce_h(const void* gy, const void* x, const void* u, const float* mean, const float* invstd,
                                   int M, int C, float* sums5, void* ws, size_t ws_bytes, const void* act_out, int act,
                                   float slope, const float* gamma, const float* beta, void* stream) {
  return bn_dbwd_reduce_impl<__bf16>("vg_bn_dbwd_reduce_h", (HC)gy, (HC)x, (HC)u, mean, invstd, M, C, sums5, ws, ws_bytes,
                                     (HC)act_out, act, slope, gamma, beta, stream);
}

extern "C" int vg_bn_dbwd_apply(const float* gy, const float* x, const float* u, const float* v, const float* w,
                                const float* mean, const float* invstd, const float* gamma, const float* sums5, float* d_gy,
                                float* d_x, float* d_gamma, int M, int C, const float* act_out, int act, float slope,
                                const float* beta, void* stream) {
  return bn_dbwd_apply_impl<float>("vg_bn_dbwd_apply", gy, x, u, v, w, mean, invstd, gamma, sums5, d_gy, d_x, d_gamma, M, C,
                                   act_out, act, slope, beta, stream);
}
extern "C" int vg_bn_dbwd_apply_h(const void* gy, const void* x, const void* u, const float* v, const float* w,
                                  const float* mean, const float* invstd, const float* gamma, const float* sums5, void* d_gy,
                                  void* d_x, float* d_gamma, int M, int C, const void* act_out, int act, float slope,
                                  const float* beta, void* stream) {
  return bn_dbwd_apply_impl<__bf16>("vg_bn_dbwd_apply_h", (HC)gy, (HC)x, (HC)u, v, w, mean, invstd, gamma, sums5, (HM)d_gy,
                                    (HM)d_x, d_gamma, M, C, (HC)act_out, act, slope, beta, stream);
}

extern "C" int vg_act_fwd(const float* x, float* y, size_t n, int act, float slope, void* stream) {
  return act_fwd_impl<float>("vg_act_fwd", x, y, n, act, slope, stream);
}
extern "C" int vg_act_fwd_h(const void* x, void* y, size_t n, int act, float slope, void* stream) {
  return act_fwd_impl<__bf16>("vg_act_fwd_h", (HC)x, (HM)y, n, act, slope, stream);
}

extern "C" int vg_act_bwd(const float* gy, const float* ref, float* gx, size_t n, int act, float slope, void* stream) {
  return act_bwd_impl<float>("vg_act_bwd", gy, ref, gx, n, act, slope, stream);
}
extern "C" int vg_act_bwd_h(const void* gy, const void* ref, void* gx, size_t n, int act, float slope, void* stream) {
  return act_bwd_impl<__bf16>("vg_act_bwd_h", (HC)gy, (HC)ref, (HM)gx, n, act, slope, stream);
}

extern "C" int vg_act_dbwd_ref(const float* gg, const float* gy, const float* ref, float* d_ref, size_t n, int act,
                               void* stream) {
  return act_dbwd_ref_impl<float>("vg_act_dbwd_ref", gg, gy, ref, d_ref, n, act, stream);
}
extern "C" int vg_act_dbwd_ref_h(const void* gg, const void* gy, const void* ref, void* d_ref, size_t n, int act,
                                 void* stream) {
  return act_dbwd_ref_impl<__bf16>("vg_act_dbwd_ref_h", (HC)gg, (HC)gy, (HC)ref, (HM)d_ref, n, act, stream);
}

extern "C" int vg_bcast_rows(const float* v, float* out, int M, int C, void* stream) {
  VG_REQUIRE(v && out && M > 0 && C > 0, "vg_bcast_rows: bad arguments");
  const bool v4 = C % 4 == 0 && all_aligned({v, out});
  const size_t total = (size_t)M * C;
  if (v4) {
    const size_t nvec = total / 4;
    hipLaunchKernelGGL((bcast_rows_kernel<4>), dim3(vg_stream_grid(nvec, 256)), dim3(256), 0, (hipStream_t)stream, v, out, nvec, C);
  } else {
    hipLaunchKernelGGL((bcast_rows_kernel<1>), dim3(vg_stream_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, v, out, total, C);
  }
  VG_CHECK_LAUNCH("vg_bcast_rows");
  return VG_OK;
}
