// Implicit-GEMM convolution core for gfx950, exact f32 on v_mfma_f32_32x32x2_f32.
//
// One kernel template serves the three contractions of a 2-d convolution over NHWC activations and
// [Co][kh][kw][Ci] weights:
//   MODE_FWD   y[m=(n,ho,wo)][co]      = sum_{k=(kh,kw,ci)} x-window[m][k] * w[co][k]
//   MODE_DGRAD gx[m=(n,hi,wi)][ci]     = sum_{k=(tap,co)}   gy-window[m][k] * w[co][tap][ci]
//              (stride > 1: output pixels are grouped by parity class, one class per blockIdx.z,
//               so each class only walks the taps that actually hit it)
//   MODE_WGRAD gw[m=co][n=(kh,kw,ci)]  = sum_{k=(n,ho,wo)}  gy[k][m] * x-window[k][n]   (split-K over blockIdx.z)
//
// Block = 256 threads = 4 waves; block tile BM x BN x 32; each wave owns TM x TN MFMA tiles of 32x32.
// Operand tiles are staged global -> registers -> LDS (next tile's loads are issued before the MFMAs of
// the current one).  A tile whose global image is contiguous along k is kept [row][k] (row stride 36
// floats: conflict-free ds_read_b128); one that is contiguous along the row index is kept [k][row]
// (ds_read_b32, consecutive lanes -> consecutive banks).  The k order inside a 32-deep step is permuted
// identically for A and B (lane half h of MFMA j in group s holds k = 8s + 4h + j), which a sum over k
// allows.
#pragma once
#include "vg_internal.h"

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };

struct IgemmParams {
  const float* A;     // fwd: x      dgrad: gy     wgrad: gy
  const float* B;     // fwd: w      dgrad: w      wgrad: x
  const float* bias;  // nullable; added to column (col % bias_mod)
  float* out;         // fwd: y      dgrad: gx     wgrad: gw or split-K slabs
  int N, H, W, Ci, Co, KH, KW, Ho, Wo, sh, sw, ph, pw;
  int M, Ncols, Ktot;  // fwd/wgrad GEMM extents (dgrad derives M and Ktot per parity class)
  int mtiles, ntiles;
  int kchunk;          // wgrad: k extent per split (multiple of 32)
  int bias_mod;
  long long slab_stride;  // wgrad: elements per split-K slab
};

#define IGEMM_BK 32
#define IGEMM_LDK 36

template <int MODE, int BM, int BN, int WAVES_M, bool B_MC, int VA, int VB>
__global__ void __launch_bounds__(256) igemm_kernel(const IgemmParams p) {
  constexpr int BK = IGEMM_BK, LDK = IGEMM_LDK;
  constexpr int WAVES_N = 4 / WAVES_M;
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr bool A_MC = (MODE == MODE_WGRAD);
  constexpr bool B_IS_MC = B_MC || (MODE != MODE_FWD);
  static_assert(TM >= 1 && TN >= 1, "tile too small for the wave arrangement");

  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK];
  float* As = smem;
  float* Bs = smem + BM * LDK;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;

  // XCD-aware tile order: blocks b and b+8 share an L2, so give each XCD a contiguous run of tiles
  // (bijective for any grid size).
  int mt, nt;
  {
    const int bid = blockIdx.x, nblk = gridDim.x;
    const int xcd = bid & 7, q = nblk >> 3, rem = nblk & 7;
    const int id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (bid >> 3);
    mt = id / p.ntiles;
    nt = id - mt * p.ntiles;
  }
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-mode extents -------------------------------------------------------------------
  int M = p.M, Ktot = p.Ktot;
  int kbeg = 0, kend = p.Ktot;
  // dgrad parity class
  int cls_h = 0, cls_w = 0, off_h = 0, off_w = 0, Hc = 0, Wc = 0, nth = 0, ntw = 0, base_h = 0, base_w = 0;
  if constexpr (MODE == MODE_DGRAD) {
    cls_h = blockIdx.z / p.sw;
    cls_w = blockIdx.z - cls_h * p.sw;
    off_h = ((cls_h - p.ph) % p.sh + p.sh) % p.sh;
    off_w = ((cls_w - p.pw) % p.sw + p.sw) % p.sw;
    Hc = off_h < p.H ? (p.H - off_h + p.sh - 1) / p.sh : 0;
    Wc = off_w < p.W ? (p.W - off_w + p.sw - 1) / p.sw : 0;
    nth = cls_h < p.KH ? (p.KH - cls_h + p.sh - 1) / p.sh : 0;
    ntw = cls_w < p.KW ? (p.KW - cls_w + p.sw - 1) / p.sw : 0;
    base_h = (off_h + p.ph - cls_h) / p.sh;
    base_w = (off_w + p.pw - cls_w) / p.sw;
    M = p.N * Hc * Wc;
    Ktot = nth * ntw * p.Co;
    kend = Ktot;
    if (m0 >= M) return;  // block-uniform
  } else if constexpr (MODE == MODE_WGRAD) {
    kbeg = blockIdx.z * p.kchunk;
    kend = min(kbeg + p.kchunk, p.Ktot);
  } else {
    ntw = p.KW;
  }
  const int nk = (kend - kbeg + BK - 1) / BK;

  // ---- A loader state -----------------------------------------------------------------------
  constexpr int RA = BM / 32;  // rows per thread, [row][k] layout
  constexpr int TPRA = BM / 4, RPA = 256 / TPRA, PA = BK / RPA;  // [k][row] layout
  int a_img[A_MC ? 1 : RA], a_h[A_MC ? 1 : RA], a_w[A_MC ? 1 : RA];
  if constexpr (!A_MC) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int m = m0 + (tid >> 3) + 32 * i;
      a_img[i] = 0;
      a_w[i] = 0;
      a_h[i] = -(1 << 28);
      if (m < M) {
        if constexpr (MODE == MODE_FWD) {
          const int hw = p.Ho * p.Wo;
          const int n = m / hw, rem = m - n * hw;
          const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
          a_img[i] = n * p.H * p.W;
          a_h[i] = ho * p.sh - p.ph;
          a_w[i] = wo * p.sw - p.pw;
        } else {
          const int hw = Hc * Wc;
          const int n = m / hw, rem = m - n * hw;
          const int ii = rem / Wc, jj = rem - ii * Wc;
          a_img[i] = n * p.Ho * p.Wo;
          a_h[i] = ii + base_h;
          a_w[i] = jj + base_w;
        }
      }
    }
  }

  // ---- B loader state -----------------------------------------------------------------------
  constexpr int RB = BN / 32;
  constexpr int TPRB = BN / 4, RPB = 256 / TPRB, PB = BK / RPB;
  // wgrad: the tile columns (kh,kw,ci) of this thread never change along k
  int b_ta[(MODE == MODE_WGRAD) ? VB == 4 ? 1 : 4 : 1], b_tb[(MODE == MODE_WGRAD) ? VB == 4 ? 1 : 4 : 1],
      b_ci[(MODE == MODE_WGRAD) ? VB == 4 ? 1 : 4 : 1];
  const int bn4 = n0 + 4 * (tid % TPRB);  // first column of this thread in [k][row] layout
  if constexpr (MODE == MODE_WGRAD) {
    constexpr int NE = VB == 4 ? 1 : 4;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int n = bn4 + e;
      const int tap = n / p.Ci;
      b_ci[e] = n - tap * p.Ci;
      b_ta[e] = tap / p.KW;
      b_tb[e] = tap - b_ta[e] * p.KW;
      if (n >= p.Ncols) b_ta[e] = -(1 << 28);
    }
  }

  f32x4 ra[A_MC ? PA : RA];
  f32x4 rb[B_IS_MC ? PB : RB];

  auto load_tiles = [&](int kt) {
    const int kb = kbeg + kt * BK;
    // ---------------- A ----------------
    if constexpr (!A_MC) {
      const int k4 = kb + 4 * (tid & 7);
      const int Cch = (MODE == MODE_FWD) ? p.Ci : p.Co;
      if constexpr (VA == 4) {
        const bool kvalid = k4 < kend;
        const int tap = k4 / Cch, c = k4 - tap * Cch;
        const int ta = tap / ntw, tb = tap - ta * ntw;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if constexpr (MODE == MODE_FWD) {
            const int hi = a_h[i] + ta, wi = a_w[i] + tb;
            if (kvalid && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
              v = *reinterpret_cast<const f32x4*>(p.A + (size_t)(a_img[i] + hi * p.W + wi) * p.Ci + c);
          } else {
            const int ho = a_h[i] - ta, wo = a_w[i] - tb;
            if (kvalid && (unsigned)ho < (unsigned)p.Ho && (unsigned)wo < (unsigned)p.Wo)
              v = *reinterpret_cast<const f32x4*>(p.A + (size_t)(a_img[i] + ho * p.Wo + wo) * p.Co + c);
          }
          ra[i] = v;
        }
      } else {
        int ta[4], tb[4], c[4];
        bool kv[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = k4 + e;
          kv[e] = k < kend;
          const int tap = k / Cch;
          c[e] = k - tap * Cch;
          ta[e] = tap / ntw;
          tb[e] = tap - ta[e] * ntw;
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (MODE == MODE_FWD) {
              const int hi = a_h[i] + ta[e], wi = a_w[i] + tb[e];
              if (kv[e] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                v[e] = p.A[(size_t)(a_img[i] + hi * p.W + wi) * p.Ci + c[e]];
            } else {
              const int ho = a_h[i] - ta[e], wo = a_w[i] - tb[e];
              if (kv[e] && (unsigned)ho < (unsigned)p.Ho && (unsigned)wo < (unsigned)p.Wo)
                v[e] = p.A[(size_t)(a_img[i] + ho * p.Wo + wo) * p.Co + c[e]];
            }
          }
          ra[i] = v;
        }
      }
    } else {
      // wgrad A = gy[pixel][co], tile [k][m]
      const int m4 = m0 + 4 * (tid % TPRA);
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int pix = kb + tid / TPRA + RPA * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pix < kend) {
          if constexpr (VA == 4) {
            if (m4 < p.Co) v = *reinterpret_cast<const f32x4*>(p.A + (size_t)pix * p.Co + m4);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (m4 + e < p.Co) v[e] = p.A[(size_t)pix * p.Co + m4 + e];
          }
        }
        ra[i] = v;
      }
    }
    // ---------------- B ----------------
    if constexpr (!B_IS_MC) {
      // fwd weights w[co][k]: tile [n][k]
      const int k4 = kb + 4 * (tid & 7);
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int n = n0 + (tid >> 3) + 32 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < p.Ncols) {
          if constexpr (VB == 4) {
            if (k4 < kend) v = *reinterpret_cast<const f32x4*>(p.B + (size_t)n * Ktot + k4);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (k4 + e < kend) v[e] = p.B[(size_t)n * Ktot + k4 + e];
          }
        }
        rb[i] = v;
      }
    } else if constexpr (MODE == MODE_FWD) {
      // plain GEMM operand B[k][n] (row stride Ncols)
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int k = kb + tid / TPRB + RPB * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < kend) {
          if constexpr (VB == 4) {
            if (bn4 < p.Ncols) v = *reinterpret_cast<const f32x4*>(p.B + (size_t)k * p.Ncols + bn4);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (bn4 + e < p.Ncols) v[e] = p.B[(size_t)k * p.Ncols + bn4 + e];
          }
        }
        rb[i] = v;
      }
    } else if constexpr (MODE == MODE_DGRAD) {
      // weights w[co][kh][kw][ci] read as B[k=(tap,co)][n=ci]
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int k = kb + tid / TPRB + RPB * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < kend) {
          const int tap = k / p.Co, co = k - tap * p.Co;
          const int ta = tap / ntw, tb = tap - ta * ntw;
          const int kh = cls_h + p.sh * ta, kw = cls_w + p.sw * tb;
          const size_t base = ((size_t)(co * p.KH + kh) * p.KW + kw) * p.Ci;
          if constexpr (VB == 4) {
            if (bn4 < p.Ncols) v = *reinterpret_cast<const f32x4*>(p.B + base + bn4);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (bn4 + e < p.Ncols) v[e] = p.B[base + bn4 + e];
          }
        }
        rb[i] = v;
      }
    } else {
      // wgrad B = x window, tile [k=pixel][n=(kh,kw,ci)]
      const int hw = p.Ho * p.Wo;
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int pix = kb + tid / TPRB + RPB * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pix < kend) {
          const int n = pix / hw, rem = pix - n * hw;
          const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
          const int hb = ho * p.sh - p.ph, wb = wo * p.sw - p.pw;
          if constexpr (VB == 4) {
            const int hi = hb + b_ta[0], wi = wb + b_tb[0];
            if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
              v = *reinterpret_cast<const f32x4*>(p.B + (size_t)((n * p.H + hi) * p.W + wi) * p.Ci + b_ci[0]);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int hi = hb + b_ta[e], wi = wb + b_tb[e];
              if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W)
                v[e] = p.B[(size_t)((n * p.H + hi) * p.W + wi) * p.Ci + b_ci[e]];
            }
          }
        }
        rb[i] = v;
      }
    }
  };

  auto store_tiles = [&]() {
    if constexpr (!A_MC) {
#pragma unroll
      for (int i = 0; i < RA; ++i)
        *reinterpret_cast<f32x4*>(&As[((tid >> 3) + 32 * i) * LDK + 4 * (tid & 7)]) = ra[i];
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i)
        *reinterpret_cast<f32x4*>(&As[(tid / TPRA + RPA * i) * BM + 4 * (tid % TPRA)]) = ra[i];
    }
    if constexpr (!B_IS_MC) {
#pragma unroll
      for (int i = 0; i < RB; ++i)
        *reinterpret_cast<f32x4*>(&Bs[((tid >> 3) + 32 * i) * LDK + 4 * (tid & 7)]) = rb[i];
    } else {
#pragma unroll
      for (int i = 0; i < PB; ++i)
        *reinterpret_cast<f32x4*>(&Bs[(tid / TPRB + RPB * i) * BN + 4 * (tid % TPRB)]) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  if (nk > 0) {
    load_tiles(0);
    store_tiles();
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tiles(kt + 1);  // global loads stay in flight under the MFMAs below

#pragma unroll
    for (int s = 0; s < 4; ++s) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int row = (wm * TM + mi) * 32 + r;
        if constexpr (!A_MC) {
          af[mi] = *reinterpret_cast<const f32x4*>(&As[row * LDK + 8 * s + 4 * h]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) af[mi][j] = As[(8 * s + 4 * h + j) * BM + row];
        }
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int col = (wn * TN + ni) * 32 + r;
        if constexpr (!B_IS_MC) {
          bf[ni] = *reinterpret_cast<const f32x4*>(&Bs[col * LDK + 8 * s + 4 * h]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[ni][j] = Bs[(8 * s + 4 * h + j) * BN + col];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][j], bf[ni][j], acc[mi][ni], 0, 0, 0);
    }

    __syncthreads();
    if (kt + 1 < nk) {
      store_tiles();
      __syncthreads();
    }
  }

  // ---- epilogue: lane holds column (lane&31) and rows (e&3)+8*(e>>2)+4*(lane>>5) of each 32x32 tile ----
  float* outp = p.out;
  if constexpr (MODE == MODE_WGRAD) outp += (size_t)blockIdx.z * p.slab_stride;
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = m0 + (wm * TM + mi) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      if (row >= M) continue;
      size_t rowoff;
      if constexpr (MODE == MODE_DGRAD) {
        const int hw = Hc * Wc;
        const int n = row / hw, rem = row - n * hw;
        const int ii = rem / Wc, jj = rem - ii * Wc;
        rowoff = (size_t)((n * p.H + ii * p.sh + off_h) * p.W + jj * p.sw + off_w) * p.Ncols;
      } else {
        rowoff = (size_t)row * p.Ncols;
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int col = n0 + (wn * TN + ni) * 32 + r;
        if (col < p.Ncols) {
          float v = acc[mi][ni][e];
          if constexpr (MODE != MODE_WGRAD) {
            if (p.bias) v += p.bias[col % p.bias_mod];
          }
          outp[rowoff + col] = v;
        }
      }
    }
  }
}

// Launch one (BN, VA, VB) variant chosen at run time. BM is 128; the wave arrangement follows BN.
template <int MODE, bool B_MC>
static inline int igemm_launch(const IgemmParams& p, int bn, int va, int vb, dim3 grid, hipStream_t st) {
#define IGEMM_CASE(BN_, WM_, VA_, VB_)                                                             \
  if (bn == BN_ && va == VA_ && vb == VB_) {                                                       \
    hipLaunchKernelGGL((igemm_kernel<MODE, 128, BN_, WM_, B_MC, VA_, VB_>), grid, dim3(256), 0, st, p); \
    return 0;                                                                                      \
  }
  IGEMM_CASE(128, 2, 4, 4)
  IGEMM_CASE(128, 2, 4, 1)
  IGEMM_CASE(128, 2, 1, 4)
  IGEMM_CASE(128, 2, 1, 1)
  IGEMM_CASE(64, 2, 4, 4)
  IGEMM_CASE(64, 2, 4, 1)
  IGEMM_CASE(64, 2, 1, 4)
  IGEMM_CASE(64, 2, 1, 1)
  IGEMM_CASE(32, 4, 4, 4)
  IGEMM_CASE(32, 4, 4, 1)
  IGEMM_CASE(32, 4, 1, 4)
  IGEMM_CASE(32, 4, 1, 1)
#undef IGEMM_CASE
  return -1;
}

static inline int igemm_pick_bn(int ncols) { return ncols > 64 ? 128 : (ncols > 32 ? 64 : 32); }

// shared by the conv translation units
int vg_check_geom(const vg_conv_geom* g, const char* who);
