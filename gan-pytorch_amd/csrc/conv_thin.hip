// Bandwidth-bound convolution shapes that an MFMA tile cannot use: very few output channels (the critic head
// Conv(512->1), Linear(->1)) in the forward direction, and very few input channels in the dgrad direction (the
// generator's ConvTranspose2d(64->3) image head, the critic stem's input gradient).  A 32-wide MFMA column block
// would be 90 % padding there; these kernels instead stream the activation once with 16-byte accesses and
// do the few multiply-adds per element on the vector ALU.
#include "igemm.h"

namespace {

// ---- forward, Co <= 4 ----------------------------------------------------------------------------------------
// TPP threads cooperate on one output pixel: each walks the (kh,kw,ci) window in float4 steps, then the partial
// dot products are reduced in a fixed order (wave shuffles, then LDS across waves).
template <int CO, int TPP>
__global__ void __launch_bounds__(256) fwd_thin_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ y, int H, int W,
                                                       int Ci, int KW, int Ho, int Wo, int sh, int sw, int ph, int pw,
                                                       int M, int K) {
  constexpr int PPB = 256 / TPP;
  __shared__ float red[4 * CO];
  const int t = threadIdx.x % TPP;
  const int pix = blockIdx.x * PPB + threadIdx.x / TPP;
  float acc[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) acc[c] = 0.f;
  if (pix < M) {
    const int hw = Ho * Wo;
    const int n = pix / hw, rem = pix - n * hw;
    const int ho = rem / Wo, wo = rem - ho * Wo;
    const int bh = ho * sh - ph, bw = wo * sw - pw;
    const size_t img = (size_t)n * H * W;
    for (int k4 = 4 * t; k4 < K; k4 += 4 * TPP) {
      const int tap = k4 / Ci, c = k4 - tap * Ci;
      const int a = tap / KW, b = tap - a * KW;
      const int hi = bh + a, wi = bw + b;
      if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (img + (size_t)hi * W + wi) * Ci + c);
#pragma unroll
        for (int co = 0; co < CO; ++co) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(w + (size_t)co * K + k4);
          acc[co] = fmaf(xv[0], wv[0], fmaf(xv[1], wv[1], fmaf(xv[2], wv[2], fmaf(xv[3], wv[3], acc[co]))));
        }
      }
    }
  }
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = vg_wave_sum(acc[co]);
  if constexpr (TPP == 64) {
    if ((threadIdx.x & 63) == 0 && pix < M) {
#pragma unroll
      for (int co = 0; co < CO; ++co) y[(size_t)pix * CO + co] = acc[co] + (bias ? bias[co] : 0.f);
    }
  } else {
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int co = 0; co < CO; ++co) red[wid * CO + co] = acc[co];
    }
    __syncthreads();
    if (threadIdx.x == 0 && pix < M) {
#pragma unroll
      for (int co = 0; co < CO; ++co)
        y[(size_t)pix * CO + co] = ((red[co] + red[CO + co]) + (red[2 * CO + co] + red[3 * CO + co])) + (bias ? bias[co] : 0.f);
    }
  }
}

// ---- dgrad, Ci <= 4 ----------------------------------------------------------------------------------------------
// One block = one parity class (blockIdx.z), one image, an 8 x 32 patch of that class's output pixels.  The gy
// pixels the patch touches are staged in LDS once per chunk of CC output channels (row stride CC+4 floats:
// conflict-free ds_read_b128 for consecutive pixels); every thread then owns one output pixel and walks its taps.
// The weights of a class are block-uniform, so the compiler keeps them in scalar registers.
constexpr int TI = 8, TJ = 32;

template <int CI>
__global__ void __launch_bounds__(256) dgrad_thin_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ gx, int H,
                                                         int W, int Co, int KH, int KW, int Ho, int Wo, int sh, int sw,
                                                         int ph, int pw, int CC, int itiles, int jtiles) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  const int cls_h = blockIdx.z / sw, cls_w = blockIdx.z - cls_h * sw;
  const int off_h = ((cls_h - ph) % sh + sh) % sh, off_w = ((cls_w - pw) % sw + sw) % sw;
  const int Hc = off_h < H ? (H - off_h + sh - 1) / sh : 0;
  const int Wc = off_w < W ? (W - off_w + sw - 1) / sw : 0;
  const int nth = cls_h < KH ? (KH - cls_h + sh - 1) / sh : 0;
  const int ntw = cls_w < KW ? (KW - cls_w + sw - 1) / sw : 0;
  const int base_h = (off_h + ph - cls_h) / sh, base_w = (off_w + pw - cls_w) / sw;
  int bid = blockIdx.x;
  const int jt = bid % jtiles; bid /= jtiles;
  const int it = bid % itiles;
  const int n = bid / itiles;
  const int i0 = it * TI, j0 = jt * TJ;
  if (i0 >= Hc || j0 >= Wc) return;  // block-uniform
  const int ti = threadIdx.x / TJ, tj = threadIdx.x % TJ;
  const int i = i0 + ti, j = j0 + tj;
  const bool live = i < Hc && j < Wc;
  float acc[CI];
#pragma unroll
  for (int c = 0; c < CI; ++c) acc[c] = 0.f;

  if (nth > 0 && ntw > 0) {
    const int RH = TI + nth - 1, RW = TJ + ntw - 1;
    const int ho_lo = i0 + base_h - (nth - 1), wo_lo = j0 + base_w - (ntw - 1);
    const int LD = CC + 4;
    const int cq = CC / 4;
    for (int cc0 = 0; cc0 < Co; cc0 += CC) {
      __syncthreads();
      for (int s = threadIdx.x; s < RH * RW * cq; s += 256) {
        const int p = s / cq, c4 = s - p * cq;
        const int rh = p / RW, rw = p - rh * RW;
        const int ho = ho_lo + rh, wo = wo_lo + rw;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if ((unsigned)ho < (unsigned)Ho && (unsigned)wo < (unsigned)Wo && cc0 + 4 * c4 < Co)
          v = *reinterpret_cast<const f32x4*>(gy + ((size_t)(n * Ho + ho) * Wo + wo) * Co + cc0 + 4 * c4);
        *reinterpret_cast<f32x4*>(&tile[p * LD + 4 * c4]) = v;
      }
      __syncthreads();
      if (live) {
        for (int a = 0; a < nth; ++a) {
          const int kh = cls_h + sh * a;
          for (int b = 0; b < ntw; ++b) {
            const int kw = cls_w + sw * b;
            const float* row = &tile[((ti + nth - 1 - a) * RW + (tj + ntw - 1 - b)) * LD];
            const float* wt = w + ((size_t)(cc0 * KH + kh) * KW + kw) * CI;   // + co * KH*KW*CI
            const size_t wstep = (size_t)KH * KW * CI;
            const int cmax = min(CC, Co - cc0);
#pragma unroll 4
            for (int c4 = 0; c4 < cmax; c4 += 4) {
              const f32x4 g = *reinterpret_cast<const f32x4*>(row + c4);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float* wp = wt + (size_t)(c4 + e) * wstep;
#pragma unroll
                for (int ci = 0; ci < CI; ++ci) acc[ci] = fmaf(g[e], wp[ci], acc[ci]);
              }
            }
          }
        }
      }
    }
  }
  if (live) {
    const int hi = i * sh + off_h, wi = j * sw + off_w;
    float* o = gx + ((size_t)(n * H + hi) * W + wi) * CI;
#pragma unroll
    for (int ci = 0; ci < CI; ++ci) o[ci] = acc[ci] + (bias ? bias[ci] : 0.f);
  }
}

// ---- dgrad, Ci <= 4, scatter form ---------------------------------------------------------------------------------
// One thread per gy pixel: it reads its Co values straight from global memory (16-byte loads, no staging) and forms
// all KH*KW*CI products with the weights, which are uniform over the block and arrive as scalar operands in
// 16-dword loads ([Co][kh][kw][Ci] keeps the KH*KW*CI weights of one co contiguous). Every gy value is read once
// and used KH*KW*CI times from a register. The per-pixel partial results go to LDS, and a second phase lets every
// output pixel of the block's tile add up the (at most ceil(KH/sh)*ceil(KW/sw)) partials that land on it, in tap
// order. A tile is RH x RW gy pixels; neighbouring tiles overlap by ceil(K/s)-1 rows/columns so that every output
// pixel finds all its partials inside one tile (no atomics); a tile that spans a whole dimension needs no overlap.
// four consecutive values of a tensor stored as f32 or as bf16 (widened: a shift)
__device__ __forceinline__ f32x4 vg_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 vg_ld4(const __bf16* p) {
  const u32x2 t = *reinterpret_cast<const u32x2*>(p);
  const f32x4 v = {__builtin_bit_cast(float, t[0] << 16), __builtin_bit_cast(float, t[0] & 0xffff0000u),
                   __builtin_bit_cast(float, t[1] << 16), __builtin_bit_cast(float, t[1] & 0xffff0000u)};
  return v;
}

// TG: storage type of the gradient that comes in (f32, or bf16 when the feature maps of the network are stored as bf16;
// the image-sized result is f32 either way)
template <int CI, int TAPS, class TG = float>
__global__ void __launch_bounds__(256) dgrad_scatter_kernel(const TG* __restrict__ gy, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ gx, int H,
                                                            int W, int Co, int KH, int KW, int Ho, int Wo, int sh, int sw,
                                                            int ph, int pw, int RW, int tiles_h, int tiles_w, int step_h,
                                                            int step_w, int org_h, int org_w) {
  constexpr int NP = TAPS * CI;
  constexpr int LDP = NP | 1;  // odd row stride: conflict-free column writes
  __shared__ float part[256 * LDP + 256 * CI];   // partials, then one pass of finished output pixels
  const int RH = 256 / RW;
  int bid = blockIdx.x;
  const int tw = bid % tiles_w; bid /= tiles_w;
  const int th = bid % tiles_h;
  const int n = bid / tiles_h;
  const int h0 = th * step_h + org_h, w0 = tw * step_w + org_w;   // first gy row / column of the tile
  const int ty = threadIdx.x / RW, tx = threadIdx.x - ty * RW;
  const int ho = h0 + ty, wo = w0 + tx;

  float acc[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) acc[k] = 0.f;
  // The tile's gy values come in through LDS, 32 channels at a time: 8 consecutive threads fetch the 128 contiguous
  // bytes of one pixel (a thread reading its own pixel straight from memory touches 16 bytes of a different line per
  // lane, and the lines are evicted before the next 16 bytes are wanted). The staging area shares `part`.
  constexpr int CCH = 32, LDS_ = CCH + 4;
  float* stage = part;
  constexpr bool kStaged = 256 * LDS_ <= 256 * LDP;
  const bool mine = (unsigned)ho < (unsigned)Ho && (unsigned)wo < (unsigned)Wo;
  if constexpr (kStaged) {
    // the 8 (pixel, 4-channel group) slots this thread fetches per chunk
    long long src[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pp = (threadIdx.x >> 3) + 32 * j;
      const int py = pp / RW, px = pp - py * RW;
      const int qh = h0 + py, qw = w0 + px;
      src[j] = ((unsigned)qh < (unsigned)Ho && (unsigned)qw < (unsigned)Wo)
                   ? ((long long)(n * Ho + qh) * Wo + qw) * Co + 4 * (threadIdx.x & 7) : -1;
    }
    // partial[pixel][k] = sum over co of gy[pixel][co] * w[co][k] on the matrix cores, as 16 x 16 x 4 products (NP is a
    // multiple of 16 for the common 3-channel 4 x 4 layer; 32-wide tiles would pad it by a third): wave v owns pixels
    // 64 v .. 64 v + 63 (four 16-row tiles) and all ceil(NP / 16) column tiles; per 32-channel chunk its A values come
    // from the staged tile and its B values (lane = column, coalesced rows of w) straight from memory.
    constexpr int NTN = (NP + 15) / 16;
    typedef float f32x4_acc __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kq = lane >> 4;
    f32x4_acc macc[4][NTN];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt) macc[mt][nt] = f32x4_acc{0.f, 0.f, 0.f, 0.f};
    // the next chunk's values are fetched into registers while the matrix cores work on the current one
    f32x4 pre[8];
    auto fetch = [&](int cc0) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (src[j] >= 0 && cc0 + 4 * (int)(threadIdx.x & 7) < Co) v = vg_ld4(gy + src[j] + cc0);
        pre[j] = v;
      }
    };
    fetch(0);
    for (int cc0 = 0; cc0 < Co; cc0 += CCH) {
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j)
        *reinterpret_cast<f32x4*>(&stage[((threadIdx.x >> 3) + 32 * j) * LDS_ + 4 * (threadIdx.x & 7)]) = pre[j];
      float bfr[CCH / 4][NTN];
#pragma unroll
      for (int j = 0; j < CCH / 4; ++j) {
        const int co = cc0 + 4 * j + kq;
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
          const int col = 16 * nt + r;
          bfr[j][nt] = (co < Co && col < NP) ? w[(size_t)co * NP + col] : 0.f;
        }
      }
      if (cc0 + CCH < Co) fetch(cc0 + CCH);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < CCH / 4; ++j) {
        float av[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) av[mt] = stage[(64 * wave + 16 * mt + r) * LDS_ + 4 * j + kq];
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
            macc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bfr[j][nt], macc[mt][nt], 0, 0, 0);
      }
    }
    __syncthreads();
    // accumulator element e of a lane is row 4 kq + e, column r of its 16 x 16 tile
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt) {
        const int col = 16 * nt + r;
        if (col < NP) {
#pragma unroll
          for (int e = 0; e < 4; ++e) part[(64 * wave + 16 * mt + 4 * kq + e) * LDP + col] = macc[mt][nt][e];
        }
      }
  } else if (mine) {
    const TG* g = gy + ((size_t)(n * Ho + ho) * Wo + wo) * Co;
    // the next quad of channels is fetched while the current one is being used
    f32x4 gv = vg_ld4(g);
    for (int co = 0; co < Co; co += 4) {
      const int nx = co + 4 < Co ? co + 4 : co;
      const f32x4 gn = vg_ld4(g + nx);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* wc = w + (size_t)(co + e) * NP;
#pragma unroll
        for (int k = 0; k < NP; ++k) acc[k] = fmaf(gv[e], wc[k], acc[k]);
      }
      gv = gn;
    }
  }
  if constexpr (!kStaged) {
#pragma unroll
    for (int k = 0; k < NP; ++k) part[threadIdx.x * LDP + k] = acc[k];
  }
  __syncthreads();

  // output pixels owned by this tile: rows whose lowest contributing gy row lies in [h0, h0 + step_h) (all rows when
  // the tile spans the dimension), likewise for columns
  const int hi_lo = tiles_h == 1 ? 0 : sh * (h0 - 1) - ph + KH, OH = tiles_h == 1 ? H : sh * step_h;
  const int wi_lo = tiles_w == 1 ? 0 : sw * (w0 - 1) - pw + KW, OW = tiles_w == 1 ? W : sw * step_w;
  float b[CI];
#pragma unroll
  for (int c = 0; c < CI; ++c) b[c] = bias ? bias[c] : 0.f;
  const int nth = (KH + sh - 1) / sh, ntw = (KW + sw - 1) / sw;
  const float inv_ow = 1.0f / (float)OW, inv_sh = 1.0f / (float)sh, inv_sw = 1.0f / (float)sw;
  // sums the partials that land on output pixel (hi, wi), in tap order
  auto gather = [&](int hi, int wi, float (&sum)[CI]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < CI; ++c) sum[c] = b[c];
    const int qh = vg_div_small(hi + ph, inv_sh), qw = vg_div_small(wi + pw, inv_sw);
    for (int a = 0; a < nth; ++a) {
      const int q = qh - a, kh = hi + ph - q * sh;
      if (kh >= KH || q < 0) break;
      const int ry = q - h0;
      if (q >= Ho || (unsigned)ry >= (unsigned)RH) continue;
      for (int bb = 0; bb < ntw; ++bb) {
        const int v = qw - bb, kw = wi + pw - v * sw;
        if (kw >= KW || v < 0) break;
        const int rx = v - w0;
        if (v >= Wo || (unsigned)rx >= (unsigned)RW) continue;
        const float* pp = &part[(ry * RW + rx) * LDP + (kh * KW + kw) * CI];
#pragma unroll
        for (int c = 0; c < CI; ++c) sum[c] += pp[c];
      }
    }
  };
  if (tiles_w == 1) {
    // The tile's output rows are whole image rows, adjacent in memory: 256 pixels per pass go through LDS and leave
    // as consecutive floats on consecutive lanes (a lane storing its own pixel writes CI floats at a CI-float stride).
    float* outbuf = part + 256 * LDP;
    const int total = OH * OW;
    for (int o0 = 0; o0 < total; o0 += 256) {
      const int o = o0 + threadIdx.x;
      const int oy = vg_div_small(o, inv_ow), ox = o - oy * OW;
      const int hi = hi_lo + oy;
      float sum[CI];
#pragma unroll
      for (int c = 0; c < CI; ++c) sum[c] = 0.f;
      if (o < total && (unsigned)hi < (unsigned)H) gather(hi, ox, sum);
#pragma unroll
      for (int c = 0; c < CI; ++c) outbuf[threadIdx.x * CI + c] = sum[c];
      __syncthreads();
      float* dst = gx + ((long long)(n * H + hi_lo) * W + o0) * CI;
#pragma unroll
      for (int k = 0; k < CI; ++k) {
        const int f = threadIdx.x + 256 * k;
        const int oo = o0 + f / CI;
        const int hh2 = hi_lo + vg_div_small(oo, inv_ow);
        if (oo < total && (unsigned)hh2 < (unsigned)H) dst[f] = outbuf[f];
      }
      __syncthreads();
    }
    return;
  }
  for (int o = threadIdx.x; o < OH * OW; o += 256) {
    const int oy = vg_div_small(o, inv_ow), ox = o - oy * OW;
    const int hi = hi_lo + oy, wi = wi_lo + ox;
    if ((unsigned)hi >= (unsigned)H || (unsigned)wi >= (unsigned)W) continue;
    float sum[CI];
#pragma unroll
    for (int c = 0; c < CI; ++c) sum[c] = b[c];
    // gy rows that reach output row hi: q = floor((hi + ph) / sh) - a with tap kh = hi + ph - q * sh < KH
    const int qh = vg_div_small(hi + ph, inv_sh), qw = vg_div_small(wi + pw, inv_sw);
    for (int a = 0; a < nth; ++a) {
      const int q = qh - a, kh = hi + ph - q * sh;
      if (kh >= KH || q < 0) break;
      const int ry = q - h0;
      if (q >= Ho || (unsigned)ry >= (unsigned)RH) continue;
      for (int bb = 0; bb < ntw; ++bb) {
        const int v = qw - bb, kw = wi + pw - v * sw;
        if (kw >= KW || v < 0) break;
        const int rx = v - w0;
        if (v >= Wo || (unsigned)rx >= (unsigned)RW) continue;
        const float* pp = &part[(ry * RW + rx) * LDP + (kh * KW + kw) * CI];
#pragma unroll
        for (int c = 0; c < CI; ++c) sum[c] += pp[c];
      }
    }
    float* op = gx + ((size_t)(n * H + hi) * W + wi) * CI;
#pragma unroll
    for (int c = 0; c < CI; ++c) op[c] = sum[c];
  }
}


// ---- dgrad, Ci <= 4, scatter form, persistent blocks (4 x 4 kernel, stride 2, 32 or 64 gradient channels) ----------------
// The arithmetic of dgrad_scatter_kernel for the shape image heads have, arranged so that a launch is not a sequence of
// phases all blocks go through together (fetch, products, gather: each of them idle for the other two thirds of the time,
// which is what the phase-by-phase timing of that kernel shows): a block walks several tiles, the w fragments of all
// channels stay in registers for the block's lifetime, the next tile's gradient values are in flight while the current
// tile's products run, and the blocks of a compute unit drift apart so that one block's products overlap another's
// gather. A fragments are read 16 bytes at a time: the k order inside a 32-channel chunk is free as long as both operands
// agree, so lane (r, kq) takes channels 16 jj + 4 kq + e (e = 0..3) for its eight products of a chunk. The gather phase
// has compile-time tap loops and writes 16 bytes per lane, straight from the sums (an output row segment of the tile is
// contiguous in memory, so lane l's four floats follow lane l - 1's).
template <int CI, int NCH, class TG>
__global__ void __launch_bounds__(256) dgrad_scatter42_kernel(const TG* __restrict__ gy, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ gx, int H,
                                                              int W, int Ho, int Wo, int ph, int pw, int rw_log2,
                                                              int tiles_h, int step_h, int org_h, int total_tiles) {
  constexpr int KK = 4, S = 2, NT = 2, Co = 32 * NCH;
  constexpr int NP = KK * KK * CI, LDP = NP | 1, NTN = (NP + 15) / 16, LDS_ = 36;
  static_assert(LDS_ <= LDP, "the staging tile shares the partials' memory");
  __shared__ float part[256 * LDP];
  float* stage = part;
  typedef float f32x4_acc __attribute__((ext_vector_type(4)));
  const int RW = 1 << rw_log2, RH = 256 >> rw_log2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 15, kq = lane >> 4;
  if ((int)blockIdx.x >= total_tiles) return;

  float bfr[NCH][2][4][NTN];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
          const int co = 32 * ch + 16 * jj + 4 * kq + e, col = 16 * nt + r;
          bfr[ch][jj][e][nt] = col < NP ? w[co * NP + col] : 0.f;
        }
  float b[CI];
#pragma unroll
  for (int c = 0; c < CI; ++c) b[c] = bias ? bias[c] : 0.f;

  // the 8 (pixel, 4-channel group) slots this thread fetches per 32-channel chunk: element offsets into gy, -1 = none
  int src[8];
  unsigned valid = 0;
  auto offsets = [&](int tile) __attribute__((always_inline)) {
    const int n = tile / tiles_h, h0 = (tile - n * tiles_h) * step_h + org_h;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int pp = (threadIdx.x >> 3) + 32 * j;
      const int qh = h0 + (pp >> rw_log2), qw = pp & (RW - 1);
      const bool ok = (unsigned)qh < (unsigned)Ho && qw < Wo;
      src[j] = ok ? ((n * Ho + qh) * Wo + qw) * Co + 4 * (int)(threadIdx.x & 7) : 0;
      valid = ok ? valid | (1u << j) : valid & ~(1u << j);
    }
  };
  // Loads without conditions around them (a slot outside the image reads element 0 and is zeroed when it is written to
  // LDS): the compiler can then count the loads in flight, and the products of a chunk do not wait for the next chunk's
  // values.
  f32x4 pre[8];
  unsigned pre_valid = 0;
  auto fetch = [&](int cc0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) pre[j] = vg_ld4(gy + src[j] + cc0);
    pre_valid = valid;
  };
  offsets(blockIdx.x);
  fetch(0);
  // the weight fragments and the bias have arrived before the tile loop starts: nothing inside it waits for them
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) asm volatile("" : "+v"(bfr[ch][jj][e][nt]));
#pragma unroll
  for (int c = 0; c < CI; ++c) asm volatile("" : "+v"(b[c]));
  const int rowf = W * CI;                                   // floats of one output row
  // Gather phase: a thread owns four consecutive floats of an output row, the same column position in every pass (rpp rows
  // per pass), so everything that depends on the column is worked out once: per float the offset of its (gy column
  // qw, tap kw = (wi + pw) % 2) partial within a gy row of `part`, whether columns qw and qw - 1 exist, and its bias.
  const int t4 = rowf >> 2, rpp = 256 / t4;
  const int grow = threadIdx.x / t4, gcol = threadIdx.x - grow * t4;
  const bool gactive = grow < rpp;
  int goff[4];
  unsigned gok = 0;
  float gbias[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int idx = 4 * gcol + e;
    const int wi = idx / CI, c = idx - wi * CI;
    const int qw = (wi + pw) >> 1, kw0 = (wi + pw) & 1;
    goff[e] = qw * LDP + kw0 * CI + c;
    gok |= (qw < Wo ? 1u : 0u) << (2 * e);                 // bb = 0: column qw
    gok |= (qw >= 1 && qw - 1 < Wo ? 1u : 0u) << (2 * e + 1);   // bb = 1: column qw - 1, tap kw0 + 2
    float bv = b[0];
#pragma unroll
    for (int cc = 1; cc < CI; ++cc) bv = c == cc ? b[cc] : bv;
    gbias[e] = bv;
  }
  for (int tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const int n = tile / tiles_h, h0 = (tile - n * tiles_h) * step_h + org_h;
    f32x4_acc macc[4][NTN];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt) macc[mt][nt] = f32x4_acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      __syncthreads();            // the readers of the memory `stage` shares (this tile's products, the last tile's gather)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(&stage[((threadIdx.x >> 3) + 32 * j) * LDS_ + 4 * (threadIdx.x & 7)]) =
            (pre_valid >> j) & 1u ? pre[j] : zero;
      }
      if (ch + 1 < NCH) {
        fetch(32 * (ch + 1));
      } else if (tile + (int)gridDim.x < total_tiles) {
        offsets(tile + gridDim.x);
        fetch(0);
      }
      __syncthreads();
      f32x4 a4[4][2];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
          a4[mt][jj] = *reinterpret_cast<const f32x4*>(&stage[(64 * wave + 16 * mt + r) * LDS_ + 16 * jj + 4 * kq]);
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
              macc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[mt][jj][e], bfr[ch][jj][e][nt], macc[mt][nt], 0, 0, 0);
      // what follows (the next chunk's values arriving, their way into LDS) waits for memory: not among the products
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    // accumulator element e of a lane is row 4 kq + e, column r of its 16 x 16 tile
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < NTN; ++nt) {
        const int col = 16 * nt + r;
        if (col < NP) {
#pragma unroll
          for (int e = 0; e < 4; ++e) part[(64 * wave + 16 * mt + 4 * kq + e) * LDP + col] = macc[mt][nt][e];
        }
      }
    __syncthreads();
    // output rows owned by this tile: those whose lowest contributing gy row lies in [h0, h0 + step_h) (all rows when one
    // tile spans the image); a row segment is W * CI consecutive floats, handled four at a time
    const int hi_lo = tiles_h == 1 ? 0 : S * (h0 - 1) - ph + KK, OH = tiles_h == 1 ? H : S * step_h;
    float* const obase = gx + ((long long)n * H + hi_lo) * rowf;
    for (int oy = grow; oy < OH; oy += rpp) {
      const int hi = hi_lo + oy;
      if ((unsigned)hi >= (unsigned)H || !gactive) continue;
      // gy rows that reach output row hi: qq = floor((hi + ph) / 2) - a with tap kh = (hi + ph) % 2 + 2 a
      const int qh = (hi + ph) >> 1, kh0 = (hi + ph) & 1;
      f32x4 out;
#pragma unroll
      for (int e = 0; e < 4; ++e) out[e] = gbias[e];
#pragma unroll
      for (int a = 0; a < NT; ++a) {
        const int qq = qh - a, ry = qq - h0;
        const bool rowok = qq >= 0 && qq < Ho && (unsigned)ry < (unsigned)RH;
        const int rowoff = (ry << rw_log2) * LDP + (kh0 + 2 * a) * KK * CI;
#pragma unroll
        for (int bb = 0; bb < NT; ++bb)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = rowok && ((gok >> (2 * e + bb)) & 1u);
            const float t = part[ok ? rowoff + goff[e] + bb * (2 * CI - LDP) : 0];
            out[e] += ok ? t : 0.f;
          }
      }
      *reinterpret_cast<f32x4*>(obase + oy * rowf + 4 * gcol) = out;
    }
  }
}

}  // namespace

// Returns 1 if the thin forward kernel took the call, 0 if the shape does not qualify, < 0 on error.
// the (channels, taps, stride) combinations stem_fwd_kernel is instantiated for
int vg_stem_fwd_applies(const vg_conv_geom* g) {
  if (g->Ci > 4 || g->Co % 32 != 0 || g->kh != g->kw || g->sh != g->sw) return 0;
  const bool c = g->Ci == 3 || g->Ci == 1;
  return c && ((g->kh == 4 && g->sh == 2) || (g->kh == 3 && g->sh == 1) || (g->kh == 5 && g->sh == 2));
}

namespace {

// Forward convolution of an image stem: at most 4 input channels (K = KK*KK*CI is a few dozen), output channels in
// multiples of 32. The generic implicit GEMM gathers such an operand with one bounds-checked 4-byte load per element;
// here a block stages the input patch of its output pixels in LDS with coalesced row loads (zero filled where the
// window leaves the image), and every lane reads its A values of v_mfma_f32_32x32x2_f32 from the patch at compile-time
// offsets, while the weight fragments of the block's 64 output channels stay in registers. A block is 4 waves; wave v
// owns output row ho0 + v of the tile, 32 consecutive output columns; blockIdx.y walks the channel tiles.
// TO: storage type of the result (f32, or bf16 rounded to nearest even at the store).
template <int CI, int KK, int S, class TO = float>
__global__ void __launch_bounds__(256) stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, TO* __restrict__ y, int H, int W,
                                                       int Co, int Ho, int Wo, int ph, int pw, int tiles_w, int tiles_h,
                                                       int tiles_total, int act, float slope) {
  constexpr int K = KK * KK * CI, KP = (K + 1) / 2;
  constexpr int TW = 32, TH = 4;
  constexpr int PWID = (TW - 1) * S + KK, PH = (TH - 1) * S + KK;   // patch extent in input pixels
  constexpr int ROWF = PWID * CI;                                   // floats per patch row
  constexpr int LDW = K | 1;                                        // odd weight row stride: conflict-free column reads
  constexpr int PATCH = (PH * ROWF + 3) & ~3;
  __shared__ float patch[2 * PATCH];
  __shared__ float wsm[64 * LDW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int co0 = blockIdx.y * 64;
  const bool two = co0 + 32 < Co;     // the second 32-channel tile exists

  // patch element i of a thread: row pr, pixel px, channel pc (the same for every tile)
  constexpr int PPT = (PH * ROWF + 255) / 256;
  int p_row[PPT], p_px[PPT], p_c[PPT];
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const int i = threadIdx.x + 256 * u;
    p_row[u] = i / ROWF;
    const int pc = i - p_row[u] * ROWF;
    p_px[u] = pc / CI;
    p_c[u] = pc - p_px[u] * CI;
    if (i >= PH * ROWF) p_row[u] = -(1 << 28);
  }
  float pre[PPT];
  auto fetch_patch = [&](int tile) {       // zero where the window leaves the image
    int bid = tile;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h;
    const int n = bid / tiles_h;
    const int hi0 = th * TH * S - ph, wi0 = tw * TW * S - pw;
    const float* img = x + (size_t)n * H * W * CI;
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int hi = hi0 + p_row[u], wi = wi0 + p_px[u];
      // no branch around the load and no use of the loaded value here: a lane outside the image reads the zero page
      const bool in = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      const float* ptr = in ? img + ((size_t)hi * W + wi) * CI + p_c[u] : vg_zero_page;
      pre[u] = *ptr;
    }
  };
  // the first tile's patch is on its way while the weights come in
  if ((int)blockIdx.x < tiles_total) fetch_patch(blockIdx.x);
  // the block's 64 weight rows come in coalesced through LDS; lane (col = r, half = h) keeps w[co0 + 32 t + r][2 j + h]
  // in registers for all the tiles the block walks (all loads first, without conditions around them, then the LDS
  // writes: written as one loop, every element waited for its own trip to memory)
  constexpr int WPT = (64 * K + 255) / 256;
  float wv[WPT];
#pragma unroll
  for (int u = 0; u < WPT; ++u) {
    const int i = threadIdx.x + 256 * u;
    const int row = i / K;
    const bool ok = i < 64 * K && co0 + row < Co;
    wv[u] = w[ok ? (size_t)co0 * K + i : 0];
  }
#pragma unroll
  for (int u = 0; u < WPT; ++u) {
    const int i = threadIdx.x + 256 * u;
    const int row = i / K, k = i - row * K;
    if (i < 64 * K) wsm[row * LDW + k] = co0 + row < Co ? wv[u] : 0.f;
  }
  __syncthreads();
  float bw[2][KP];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      const int k = 2 * j + h;
      bw[t][j] = k < K ? wsm[(32 * t + r) * LDW + k] : 0.f;
    }
  float bv[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) bv[t] = (bias && co0 + 32 * t + r < Co) ? bias[co0 + 32 * t + r] : 0.f;
  // Pin the arrival of the bias values here. Their first use is inside the conditionally executed store blocks below;
  // left there, every one of those blocks gets its own wait for all outstanding memory operations, which also waits
  // for the store before it and serialises the stores.
  asm volatile("" ::"v"(bv[0]), "v"(bv[1]));

  // Two patch buffers. Per tile: the next tile's patch is requested, the matrix cores run on the current buffer, the
  // arrived values go to the other buffer and only then are this tile's results stored. The stores thus drain under
  // the next tile's products: the wait for the patch loads (the memory counter completes in order) falls behind them,
  // where the previous tile's stores have long finished, instead of directly after a tile's own stores.
  if ((int)blockIdx.x < tiles_total) {
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int i = threadIdx.x + 256 * u;
      if (i < PH * ROWF) patch[i] = pre[u];
    }
  }
  __syncthreads();
  // the activation as two uniform values, so that the store loop below stays free of branches on `act`:
  // non-positive results are multiplied by `neg` (1: no activation) or, for ReLU, replaced by 0
  const bool relu = act == VG_ACT_RELU;
  const float neg = act == VG_ACT_LRELU ? slope : 1.f;
  int cur = 0;
  for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
    int bid = tile;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h;
    const int n = bid / tiles_h;
    const int ho0 = th * TH, wo0 = tw * TW;
    const bool more = tile + (int)gridDim.x < tiles_total;
    if (more) fetch_patch(tile + gridDim.x);   // in flight under the MFMAs below

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    // A[pixel r of this wave's row][k] = patch[(wave*S + kh) * ROWF + (r*S + kw) * CI + ci], k = (kh*KK + kw)*CI + ci
    const float* base = patch + cur * PATCH + wave * S * ROWF + r * S * CI;
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      constexpr int KC = KK * CI;
      const int k0 = 2 * j, k1 = 2 * j + 1;
      const int o0 = (k0 / KC) * ROWF + (k0 % KC);
      const int o1 = k1 < K ? (k1 / KC) * ROWF + (k1 % KC) : o0;
      float a = base[h ? o1 : o0];
      if (k1 >= K && h) a = 0.f;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[0][j], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[1][j], acc[1], 0, 0, 0);
    }
    if (more) {
      float* nxt = patch + (cur ^ 1) * PATCH;
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i < PH * ROWF) nxt[i] = pre[u];
      }
    }
    const int ho = ho0 + wave;
    if (ho < Ho) {
      // lane (r, h) owns channel co0 + 32 t + r of pixels wo0 + 4 h + {0..3} + 8 g: one 64-bit base per tile, then
      // offsets that are compile-time multiples of Co
      TO* yb = y + ((size_t)(n * Ho + ho) * Wo + wo0 + 4 * h) * Co + co0 + r;
      auto put = [&](auto checked) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          if (t == 1 && !two) break;
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int px = (e & 3) + 8 * (e >> 2);
            float v = acc[t][e] + bv[t];
            const float lo = relu ? 0.f : v * neg;
            v = v > 0.f ? v : lo;
            if (!decltype(checked)::value || wo0 + 4 * h + px < Wo) yb[px * Co + 32 * t] = (TO)v;
          }
        }
      };
      // uniform: when no pixel of the tile lies beyond the row the stores carry no conditions at all
      if (wo0 + TW <= Wo) put(std::false_type{});
      else put(std::true_type{});
    }
    __syncthreads();      // the other buffer is complete, this one is free
    cur ^= 1;
  }
}

// Weight gradient of an image stem (the layers stem_fwd_kernel serves, output channels in multiples of 64):
// gw[co][k] = sum over output pixels of gy[pixel][co] * patch[pixel][k], K = KK*KK*CI a few dozen. The generic
// implicit GEMM gathers the patch operand with one 4-byte load per element; here the input patch of a tile of output
// pixels is staged in LDS exactly as in the forward kernel and read from there as the B operand of 16 x 16 x 4
// products (lane = column k, four consecutive pixels of the wave's row), while the A operand comes straight from
// memory: lane (i, q) reads the 16 bytes gy[pixel 4 j + q][co0 + 4 i .. + 3], whose element m feeds row tile m, so row
// i of tile m is channel co0 + 4 i + m and a wave's load covers whole 256-byte pixel rows. A block (4 waves, wave v =
// row v of a 4 x 32 pixel tile) walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... with the next tile's operands in
// flight, keeps its 64 x K partial result in registers, adds the four waves' parts in wave order through LDS and writes
// one slab; the slabs are then added in slab order by the split-K reduce of the generic path (same layout and count).
template <int CI, int KK, int S, class TG = float>
__global__ void __launch_bounds__(256) stem_wgrad_kernel(const TG* __restrict__ gy, const float* __restrict__ x,
                                                         float* __restrict__ slabs, int H, int W, int Co, int Ho, int Wo,
                                                         int ph, int pw, int tiles_w, int tiles_h, int tiles_total) {
  constexpr int K = KK * KK * CI, NT = (K + 15) / 16, KC = KK * CI;
  constexpr int TW = 32, TH = 4;
  constexpr int PWID = (TW - 1) * S + KK, PH = (TH - 1) * S + KK;
  constexpr int ROWF = PWID * CI;
  constexpr int PATCH = (PH * ROWF + 3) & ~3;
  typedef float f32x4_acc __attribute__((ext_vector_type(4)));
  __shared__ float patch[2 * PATCH];
  __shared__ float red[4 * 64 * K];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i16 = lane & 15, q = lane >> 4;
  const int co0 = blockIdx.y * 64;

  // column k = 16 nt + i16 of this lane: offset of its patch element relative to the pixel's window origin
  int offc[NT];
  bool cok[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = 16 * nt + i16;
    cok[nt] = c < K;
    const int cc = cok[nt] ? c : 0;
    offc[nt] = (cc / KC) * ROWF + (cc % KC);
  }

  constexpr int PPT = (PH * ROWF + 255) / 256;
  int p_row[PPT], p_px[PPT], p_c[PPT];
#pragma unroll
  for (int u = 0; u < PPT; ++u) {
    const int i = threadIdx.x + 256 * u;
    p_row[u] = i / ROWF;
    const int pc = i - p_row[u] * ROWF;
    p_px[u] = pc / CI;
    p_c[u] = pc - p_px[u] * CI;
    if (i >= PH * ROWF) p_row[u] = -(1 << 28);
  }
  float pre[PPT];
  f32x4 nxt[8];
  // requests the operands of `tile`: its input patch (zero where the window leaves the image) and this lane's gy values
  auto fetch = [&](int tile) __attribute__((always_inline)) {
    int bid = tile;
    const int tw = bid % tiles_w; bid /= tiles_w;
    const int th = bid % tiles_h;
    const int n = bid / tiles_h;
    const int ho = th * TH + wave, wo0 = tw * TW;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int wo = wo0 + 4 * j + q;
      // an invalid lane reads the zero page: the loaded value itself is not touched, so nothing waits for it here
      const bool ok = ho < Ho && wo < Wo;
      const TG* ptr = ok ? gy + ((size_t)(n * Ho + ho) * Wo + wo) * Co + co0 + 4 * i16 : reinterpret_cast<const TG*>(vg_zero_page);
      nxt[j] = vg_ld4(ptr);
    }
    const int hi0 = th * TH * S - ph, wi0 = tw * TW * S - pw;
    const float* img = x + (size_t)n * H * W * CI;
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int hi = hi0 + p_row[u], wi = wi0 + p_px[u];
      const bool in = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
      const float* ptr = in ? img + ((size_t)hi * W + wi) * CI + p_c[u] : vg_zero_page;
      pre[u] = *ptr;
    }
  };

  f32x4_acc acc[4][NT];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4_acc{0.f, 0.f, 0.f, 0.f};

  if ((int)blockIdx.x < tiles_total) {
    fetch(blockIdx.x);
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int i = threadIdx.x + 256 * u;
      if (i < PH * ROWF) patch[i] = pre[u];
    }
  }
  __syncthreads();
  int cur = 0;
  for (int tile = blockIdx.x; tile < tiles_total; tile += gridDim.x) {
    f32x4 a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = nxt[j];
    const bool more = tile + (int)gridDim.x < tiles_total;
    if (more) fetch(tile + gridDim.x);       // in flight under the products below
    const float* base = patch + cur * PATCH + wave * S * ROWF + q * S * CI;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const float v = base[offc[nt] + 4 * j * S * CI];
        b[nt] = cok[nt] ? v : 0.f;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][mt], b[nt], acc[mt][nt], 0, 0, 0);
    }
    if (more) {
      float* other = patch + (cur ^ 1) * PATCH;
#pragma unroll
      for (int u = 0; u < PPT; ++u) {
        const int i = threadIdx.x + 256 * u;
        if (i < PH * ROWF) other[i] = pre[u];
      }
    }
    __syncthreads();
    cur ^= 1;
  }

  // The four waves' parts go to LDS side by side (element e of a lane is row 4 q + e, column i16 of its 16 x 16 tile);
  // then every thread adds the four parts of its elements in wave order and writes them to the slab.
  float* mine = red + wave * (64 * K);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (!cok[nt]) continue;
#pragma unroll
      for (int e = 0; e < 4; ++e) mine[(4 * (4 * q + e) + mt) * K + 16 * nt + i16] = acc[mt][nt][e];
    }
  __syncthreads();
  float* slab = slabs + ((size_t)blockIdx.x * Co + co0) * K;
  for (int i = threadIdx.x; i < 64 * K; i += 256)
    slab[i] = ((red[i] + red[64 * K + i]) + red[2 * 64 * K + i]) + red[3 * 64 * K + i];
}

int try_fwd_stem(const vg_conv_geom* g, const float* x, const float* w, const float* bias, void* y, hipStream_t st,
                 int act, float slope, bool out_h = false) {
  if (!vg_stem_fwd_applies(g)) return 0;
  const int tiles_w = (g->Wo + 31) / 32, tiles_h = (g->Ho + 3) / 4;
  const long long blocks = (long long)tiles_w * tiles_h * g->N;
  if (blocks > 0x7fffffffLL) return 0;
  // a block keeps its weight fragments in registers and walks tiles, a tile's stores draining under the next tile's
  // products: two blocks per CU, so that every block sees several tiles
  const long long cap = 512;
  dim3 grid((unsigned)(blocks < cap ? blocks : cap), (unsigned)((g->Co + 63) / 64));
  bool done = false;
#define STEM_FWD(CI_, KK_, S_)                                                                                          \
  if (!done && g->Ci == CI_ && g->kh == KK_ && g->sh == S_) {                                                          \
    if (out_h)                                                                                                         \
      vg_launch_timed(stem_fwd_kernel<CI_, KK_, S_, __bf16>, grid, dim3(256), 0, st, x, w, bias, (__bf16*)y, g->H,     \
                      g->W, g->Co, g->Ho, g->Wo, g->ph, g->pw, tiles_w, tiles_h, (int)blocks, act, slope);            \
    else                                                                                                               \
      vg_launch_timed(stem_fwd_kernel<CI_, KK_, S_, float>, grid, dim3(256), 0, st, x, w, bias, (float*)y, g->H, g->W, \
                      g->Co, g->Ho, g->Wo, g->ph, g->pw, tiles_w, tiles_h, (int)blocks, act, slope);                  \
    done = true;                                                                                                       \
  }
  STEM_FWD(3, 4, 2) STEM_FWD(1, 4, 2) STEM_FWD(3, 3, 1) STEM_FWD(1, 3, 1) STEM_FWD(3, 5, 2) STEM_FWD(1, 5, 2)
#undef STEM_FWD
  if (!done) return 0;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vg_fail(VG_ERR_LAUNCH, "vg_conv2d_fwd(stem): launch failed: %s", hipGetErrorString(e));
  return 1;
}

}  // namespace

// Stem weight gradient into `nslabs` slabs of [Co][K] floats (the generic split-K layout); 1 if taken, 0 if the shape
// does not qualify, < 0 on error.
int vg_stem_wgrad_applies(const vg_conv_geom* g) {
  return vg_stem_fwd_applies(g) && g->Co % 64 == 0 && g->kh == g->kw && g->sh == g->sw;
}

static int wgrad_stem(const vg_conv_geom* g, const void* gy, bool gy_h, const float* x, float* slabs, int nslabs, hipStream_t st);

int vg_try_wgrad_stem(const vg_conv_geom* g, const float* gy, const float* x, float* slabs, int nslabs, hipStream_t st) {
  return wgrad_stem(g, gy, false, x, slabs, nslabs, st);
}

static int wgrad_stem(const vg_conv_geom* g, const void* gy, bool gy_h, const float* x, float* slabs, int nslabs, hipStream_t st) {
  if (!vg_stem_wgrad_applies(g) || nslabs < 2 || !vg_aligned16(gy)) return 0;
  const int tiles_w = (g->Wo + 31) / 32, tiles_h = (g->Ho + 3) / 4;
  const long long tiles = (long long)tiles_w * tiles_h * g->N;
  if (tiles > 0x7fffffffLL) return 0;
  dim3 grid((unsigned)nslabs, (unsigned)(g->Co / 64));
  bool done = false;
#define STEM_WGRAD(CI_, KK_, S_)                                                                                       \
  if (!done && g->Ci == CI_ && g->kh == KK_ && g->sh == S_) {                                                          \
    if (gy_h)                                                                                                          \
      vg_launch_timed(stem_wgrad_kernel<CI_, KK_, S_, __bf16>, grid, dim3(256), 0, st, (const __bf16*)gy, x, slabs,    \
                      g->H, g->W, g->Co, g->Ho, g->Wo, g->ph, g->pw, tiles_w, tiles_h, (int)tiles);                    \
    else                                                                                                               \
      vg_launch_timed(stem_wgrad_kernel<CI_, KK_, S_, float>, grid, dim3(256), 0, st, (const float*)gy, x, slabs,      \
                      g->H, g->W, g->Co, g->Ho, g->Wo, g->ph, g->pw, tiles_w, tiles_h, (int)tiles);                    \
    done = true;                                                                                                       \
  }
  STEM_WGRAD(3, 4, 2) STEM_WGRAD(1, 4, 2) STEM_WGRAD(3, 3, 1) STEM_WGRAD(1, 3, 1) STEM_WGRAD(3, 5, 2) STEM_WGRAD(1, 5, 2)
#undef STEM_WGRAD
  if (!done) return 0;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vg_fail(VG_ERR_LAUNCH, "vg_conv2d_wgrad(stem): launch failed: %s", hipGetErrorString(e));
  return 1;
}

// Stem forward with ReLU / LeakyReLU applied to the result before it is stored; 0 when the stem kernel does not apply.
int vg_try_fwd_stem_act(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, int act,
                        float slope, hipStream_t st) {
  return try_fwd_stem(g, x, w, bias, y, st, act, slope);
}

int vg_try_fwd_thin(const vg_conv_geom* g, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
  if (int r = try_fwd_stem(g, x, w, bias, y, st, VG_ACT_NONE, 0.f)) return r;
  const int K = g->kh * g->kw * g->Ci;
  if (g->Co > 4 || g->Ci % 4 != 0 || !vg_aligned16(x) || !vg_aligned16(w)) return 0;
  const int M = g->N * g->Ho * g->Wo;
  const bool wide = K >= 2048;
  const int ppb = wide ? 1 : 4;
  dim3 grid((unsigned)((M + ppb - 1) / ppb));
#define THIN_FWD(CO_)                                                                                                  \
  if (g->Co == CO_) {                                                                                                  \
    if (wide)                                                                                                          \
      vg_launch_timed(fwd_thin_kernel<CO_, 256>, grid, dim3(256), 0, st, x, w, bias, y, g->H, g->W, g->Ci, g->kw,  \
                         g->Ho, g->Wo, g->sh, g->sw, g->ph, g->pw, M, K);                                             \
    else                                                                                                               \
      vg_launch_timed(fwd_thin_kernel<CO_, 64>, grid, dim3(256), 0, st, x, w, bias, y, g->H, g->W, g->Ci, g->kw,   \
                         g->Ho, g->Wo, g->sh, g->sw, g->ph, g->pw, M, K);                                             \
  }
  THIN_FWD(1) THIN_FWD(2) THIN_FWD(3) THIN_FWD(4)
#undef THIN_FWD
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vg_fail(VG_ERR_LAUNCH, "vg_conv2d_fwd(thin): launch failed: %s", hipGetErrorString(e));
  return 1;
}

static int thin_dgrad_chunk(const vg_conv_geom* g) {
  // widest co chunk (multiple of 4) whose staged patch fits 64 KB of LDS; 0 if even 4 channels do not fit
  const int nth = (g->kh + g->sh - 1) / g->sh, ntw = (g->kw + g->sw - 1) / g->sw;
  const int pixels = (TI + nth - 1) * (TJ + ntw - 1);
  for (int cc = 64; cc >= 4; cc /= 2)
    if ((long long)pixels * (cc + 4) * 4 <= 64 * 1024) return cc;
  return 0;
}

int vg_thin_dgrad_applies(const vg_conv_geom* g) {
  return g->Ci <= 4 && g->Co % 4 == 0 && thin_dgrad_chunk(g) > 0 && g->sh * g->sw <= 65535;
}


// Tiling of one gy dimension of `extent` pixels for the scatter kernel with `r` tile rows and `nt` = ceil(K/s):
// one tile from 0 when it covers the dimension, else tiles stepping by r - (nt - 1) from -(nt - 1).
static void scatter_tiling(int extent, int r, int nt, int* tiles, int* step, int* org) {
  if (extent <= r) {
    *tiles = 1; *step = r; *org = 0;
  } else {
    *step = r - (nt - 1);
    *org = -(nt - 1);
    *tiles = (extent + nt - 1 + *step - 1) / *step;
  }
}

static int try_dgrad_scatter(const vg_conv_geom* g, const void* gy, const float* w, const float* bias, float* gx,
                             hipStream_t st, bool gy_h = false) {
  const int taps = g->kh * g->kw;
  if (g->Ci > 4 || g->Co % 4 != 0 || !vg_aligned16(gy) || !vg_aligned16(w)) return 0;
  if (taps != 9 && taps != 16 && taps != 25) return 0;
  const int nth = (g->kh + g->sh - 1) / g->sh, ntw = (g->kw + g->sw - 1) / g->sw;
  // tile shape with the least idle / duplicated threads
  int best_rw = 0, bt_h = 0, bt_w = 0, bs_h = 0, bs_w = 0, bo_h = 0, bo_w = 0;
  long long best_blocks = 0;
  for (int rw = 8; rw <= 64; rw *= 2) {
    const int rh = 256 / rw;
    if (rh < nth + 1 && g->Ho > rh) continue;
    if (rw < ntw + 1 && g->Wo > rw) continue;
    int th, sh_, oh, tw, sw_, ow;
    scatter_tiling(g->Ho, rh, nth, &th, &sh_, &oh);
    scatter_tiling(g->Wo, rw, ntw, &tw, &sw_, &ow);
    const long long blocks = (long long)th * tw;
    if (best_rw == 0 || blocks < best_blocks || (blocks == best_blocks && tw == 1)) {
      best_rw = rw; best_blocks = blocks; bt_h = th; bt_w = tw; bs_h = sh_; bs_w = sw_; bo_h = oh; bo_w = ow;
    }
  }
  if (best_rw == 0) return 0;
  const long long grid = best_blocks * g->N;
  if (grid > 0x7fffffffLL) return 0;
  // the image-head shape (4 x 4, stride 2, 32 or 64 gradient channels, one tile across the width): persistent blocks
  if (taps == 16 && g->kh == 4 && g->sh == 2 && g->sw == 2 && bt_w == 1 && (g->Ci == 3 || g->Ci == 4) &&
      (g->Co == 32 || g->Co == 64) && (g->W * g->Ci) % 4 == 0 && g->W * g->Ci <= 1024 && vg_aligned16(gx) &&
      (long long)g->N * g->Ho * g->Wo * g->Co < 0x7fffffffLL) {     // (a row of the image fits one gather pass: 256 lanes x 4 floats)
    int lg = 0;
    while ((1 << lg) < best_rw) ++lg;
    // two blocks per compute unit (what the kernel's registers allow), each walking every 512th tile: with the 1280
    // tiles of a batch of 256 images of 64 x 64 a compute unit gets 3 + 2 tiles. Measured on that shape: 27.5 us at 512
    // blocks, 31.4 at 256, 29.3 at 768, 33.2 with one tile per block (44 for dgrad_scatter_kernel).
    const long long cap = 512;
    const unsigned blocks = (unsigned)(grid < cap ? grid : cap);
#define THIN_SCATTER42(CI_, NCH_)                                                                                        \
    if (g->Ci == CI_ && g->Co == 32 * NCH_) {                                                                            \
      if (gy_h)                                                                                                          \
        vg_launch_timed(dgrad_scatter42_kernel<CI_, NCH_, __bf16>, dim3(blocks), dim3(256), 0, st, (const __bf16*)gy, w,  \
                        bias, gx, g->H, g->W, g->Ho, g->Wo, g->ph, g->pw, lg, bt_h, bs_h, bo_h, (int)grid);                \
      else                                                                                                               \
        vg_launch_timed(dgrad_scatter42_kernel<CI_, NCH_, float>, dim3(blocks), dim3(256), 0, st, (const float*)gy, w,    \
                        bias, gx, g->H, g->W, g->Ho, g->Wo, g->ph, g->pw, lg, bt_h, bs_h, bo_h, (int)grid);                \
    }
    THIN_SCATTER42(3, 1) THIN_SCATTER42(3, 2) THIN_SCATTER42(4, 1) THIN_SCATTER42(4, 2)
#undef THIN_SCATTER42
    hipError_t e42 = hipGetLastError();
    if (e42 != hipSuccess)
      return vg_fail(VG_ERR_LAUNCH, "vg_conv2d_dgrad(scatter): launch failed: %s", hipGetErrorString(e42));
    return 1;
  }
#define THIN_SCATTER(CI_, TAPS_)                                                                                        \
  if (g->Ci == CI_ && taps == TAPS_) {                                                                                  \
    if (gy_h)                                                                                                           \
      vg_launch_timed(dgrad_scatter_kernel<CI_, TAPS_, __bf16>, dim3((unsigned)grid), dim3(256), 0, st,                 \
                      (const __bf16*)gy, w, bias, gx, g->H, g->W, g->Co, g->kh, g->kw, g->Ho, g->Wo, g->sh, g->sw,      \
                      g->ph, g->pw, best_rw, bt_h, bt_w, bs_h, bs_w, bo_h, bo_w);                                       \
    else                                                                                                                \
      vg_launch_timed(dgrad_scatter_kernel<CI_, TAPS_, float>, dim3((unsigned)grid), dim3(256), 0, st,                  \
                      (const float*)gy, w, bias, gx, g->H, g->W, g->Co, g->kh, g->kw, g->Ho, g->Wo, g->sh, g->sw,       \
                      g->ph, g->pw, best_rw, bt_h, bt_w, bs_h, bs_w, bo_h, bo_w);                                       \
  }
  THIN_SCATTER(1, 9) THIN_SCATTER(2, 9) THIN_SCATTER(3, 9) THIN_SCATTER(4, 9)
  THIN_SCATTER(1, 16) THIN_SCATTER(2, 16) THIN_SCATTER(3, 16) THIN_SCATTER(4, 16)
  THIN_SCATTER(1, 25) THIN_SCATTER(2, 25) THIN_SCATTER(3, 25) THIN_SCATTER(4, 25)
#undef THIN_SCATTER
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vg_fail(VG_ERR_LAUNCH, "vg_conv2d_dgrad(scatter): launch failed: %s", hipGetErrorString(e));
  return 1;
}

int vg_try_dgrad_thin(const vg_conv_geom* g, const float* gy, const float* w, const float* bias, float* gx,
                      hipStream_t st) {
  if (int r = try_dgrad_scatter(g, gy, w, bias, gx, st)) return r;
  if (!vg_thin_dgrad_applies(g) || !vg_aligned16(gy)) return 0;
  int cc = thin_dgrad_chunk(g);
  while (cc > g->Co && cc > 4) cc /= 2;
  const int nth = (g->kh + g->sh - 1) / g->sh, ntw = (g->kw + g->sw - 1) / g->sw;
  const size_t lds = (size_t)(TI + nth - 1) * (TJ + ntw - 1) * (cc + 4) * sizeof(float);
  const int Hc = (g->H + g->sh - 1) / g->sh, Wc = (g->W + g->sw - 1) / g->sw;
  const int itiles = (Hc + TI - 1) / TI, jtiles = (Wc + TJ - 1) / TJ;
  dim3 grid((unsigned)(g->N * itiles * jtiles), 1, (unsigned)(g->sh * g->sw));
#define THIN_DGRAD(CI_)                                                                                                 \
  if (g->Ci == CI_)                                                                                                     \
    vg_launch_timed(dgrad_thin_kernel<CI_>, grid, dim3(256), lds, st, gy, w, bias, gx, g->H, g->W, g->Co, g->kh,    \
                       g->kw, g->Ho, g->Wo, g->sh, g->sw, g->ph, g->pw, cc, itiles, jtiles);
  THIN_DGRAD(1) THIN_DGRAD(2) THIN_DGRAD(3) THIN_DGRAD(4)
#undef THIN_DGRAD
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vg_fail(VG_ERR_LAUNCH, "vg_conv2d_dgrad(thin): launch failed: %s", hipGetErrorString(e));
  return 1;
}

// ---- image stems of a network whose feature maps are stored as bf16 -----------------------------------------------------
// The layers next to an image (the critic's first convolution, the generator's last transposed convolution) have an f32
// image on one side and a bf16 feature map on the other: forward f32 x -> bf16 y (optionally through ReLU / LeakyReLU),
// input gradient bf16 gy -> f32 gx, weight gradient from bf16 gy and f32 x. Same kernels as the f32 entry points, the
// feature-map side loaded / stored in its storage type.
extern "C" int vg_conv2d_stem_h_applies(int mode, const vg_conv_geom* g) {
  if (vg_check_geom(g, "vg_conv2d_stem_h_applies") || mode < 0 || mode > 2) return 0;
  if (mode == 0) return vg_stem_fwd_applies(g);
  if (mode == 1) {
    const int taps = g->kh * g->kw;
    const bool gemm = g->Ho == 1 && g->Wo == 1 && g->ph == 0 && g->pw == 0 && g->H == g->kh && g->W == g->kw;
    return !gemm && g->Ci <= 4 && g->Co % 4 == 0 && (taps == 9 || taps == 16 || taps == 25);
  }
  return vg_stem_wgrad_applies(g) && vg_conv2d_wgrad_workspace_bytes(g) > 0;
}

extern "C" int vg_conv2d_stem_fwd_h(const vg_conv_geom* g, const float* x, const float* w, const float* bias, void* y,
                                    int act, float slope, void* stream) {
  if (!vg_conv2d_stem_h_applies(0, g))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_stem_fwd_h: not an image stem the stem kernel serves (vg_conv2d_stem_h_applies)");
  VG_REQUIRE(x && w && y, "vg_conv2d_stem_fwd_h: null tensor pointer");
  VG_REQUIRE(act == VG_ACT_NONE || act == VG_ACT_RELU || act == VG_ACT_LRELU, "vg_conv2d_stem_fwd_h: only ReLU / LeakyReLU can be fused");
  const int r = try_fwd_stem(g, x, w, bias, y, (hipStream_t)stream, act, slope, true);
  if (r == 0) return vg_fail(VG_ERR_INVALID, "vg_conv2d_stem_fwd_h: geometry too large for the stem kernel");
  return r < 0 ? r : VG_OK;
}

extern "C" int vg_conv2d_stem_dgrad_h(const vg_conv_geom* g, const void* gy, const float* w, const float* bias, float* gx,
                                      void* stream) {
  if (!vg_conv2d_stem_h_applies(1, g))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_stem_dgrad_h: geometry not served (vg_conv2d_stem_h_applies)");
  VG_REQUIRE(gy && w && gx && vg_aligned16(gy) && vg_aligned16(w), "vg_conv2d_stem_dgrad_h: null or misaligned tensor");
  const int r = try_dgrad_scatter(g, gy, w, bias, gx, (hipStream_t)stream, true);
  if (r == 0) return vg_fail(VG_ERR_INVALID, "vg_conv2d_stem_dgrad_h: no tile shape for this geometry");
  return r < 0 ? r : VG_OK;
}

extern "C" int vg_conv2d_stem_wgrad_partial_h(const vg_conv_geom* g, const void* gy, const float* x, void* workspace,
                                              size_t workspace_bytes, void* stream) {
  if (!vg_conv2d_stem_h_applies(2, g))
    return vg_fail(VG_ERR_INVALID, "vg_conv2d_stem_wgrad_partial_h: geometry not served (vg_conv2d_stem_h_applies)");
  const size_t need = vg_conv2d_wgrad_workspace_bytes(g);
  VG_REQUIRE(gy && x && vg_aligned16(gy), "vg_conv2d_stem_wgrad_partial_h: null or misaligned tensor");
  if (!workspace || workspace_bytes < need || !vg_aligned16(workspace))
    return vg_fail(VG_ERR_WORKSPACE, "vg_conv2d_stem_wgrad_partial_h: workspace of %zu bytes (16-byte aligned) required, got %zu",
                   need, workspace_bytes);
  const int nslabs = (int)(need / ((size_t)g->Co * g->kh * g->kw * g->Ci * sizeof(float)));
  const int r = wgrad_stem(g, gy, true, x, (float*)workspace, nslabs, (hipStream_t)stream);
  if (r == 0) return vg_fail(VG_ERR_INVALID, "vg_conv2d_stem_wgrad_partial_h: geometry too large for the stem kernel");
  return r < 0 ? r : VG_OK;
}
