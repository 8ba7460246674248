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
//
// PREC = 1 (bf16 mode): the same kernel with the operands rounded to bf16 (round to nearest even) when a tile is
// written to LDS and v_mfma_f32_32x32x16_bf16 doing the products; tensors in global memory, accumulation and the
// epilogue stay f32.  A [row][k] tile is then kept as bf16 with a row stride of 40 elements (80 bytes:
// conflict-free ds_read_b128 of the 8 consecutive k one lane feeds the MFMA); a [k][row] tile is kept as
// [k/2][row] dwords holding the pair (k, k+1), written 16 bytes at a time and read with four ds_read_b32 per
// fragment; a thread that loads such a tile owns both rows of a pair.
#pragma once
#include <type_traits>
#include "vg_internal.h"

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { PREC_F32 = 0, PREC_BF16 = 1, PREC_SPLIT = 2 };   // PREC_SPLIT is served by pgemm.h; here it runs as f32

enum { MODE_FWD = 0, MODE_DGRAD = 1, MODE_WGRAD = 2 };

// A predicated-off lane of the thin kernels reads these 16 bytes instead of branching around its load.
static __device__ __attribute__((aligned(16))) float vg_zero_page[4] = {0.f, 0.f, 0.f, 0.f};  // never written

// n / d for 0 <= n < 2^22 and small positive d, with inv = 1.0f / d: exact because (n + 0.5) / d is at least
// 0.5 / d away from the next integer, far more than the float rounding of the product.
__device__ __forceinline__ int vg_div_small(int n, float inv) { return (int)(((float)n + 0.5f) * inv); }
// the same when `fast` (a kernel-uniform flag saying n stays below 2^22), else a true integer division
__device__ __forceinline__ int vg_div_row(int n, int d, float inv, bool fast) {
  return fast ? vg_div_small(n, inv) : n / d;
}

struct IgemmParams {
  const float* A;     // fwd: x      dgrad: gy     wgrad: gy
  const float* B;     // fwd: w      dgrad: w      wgrad: x
  const float* bias;  // nullable; added to column (col % bias_mod)
  float* out;         // fwd: y      dgrad: gx     wgrad: gw or split-K slabs
  int N, H, W, Ci, Co, KH, KW, Ho, Wo, sh, sw, ph, pw;
  int M, Ncols, Ktot;  // fwd/wgrad GEMM extents (dgrad derives M and Ktot per parity class)
  int mtiles, ntiles;
  int kchunk;          // wgrad: k extent per split (multiple of 32)
  int fast_div;        // all row indices < 2^22: row -> (image, y, x) decoding may use float reciprocals
  int bias_mod;
  long long slab_stride;  // wgrad: elements per split-K slab
  unsigned a_bytes, b_bytes;  // sizes of A and B (below 2 GiB each: igemm_set_operands)
  double* stats;              // STATS kernels: per-tile column statistics [tiles][3 or 2][Ncols], then [tiles] row counts
  int stats_tiles;            // row tiles over all parity classes
  // STATS == 2: the output is the gradient arriving at a BatchNorm's output; bx is that norm's input (laid out like
  // the output), the rest its statistics, affine parameters and fused activation
  const float *bx, *bmean, *binvstd, *bgamma, *bbeta;
  int bact;
  float bslope;
  // dgrad: what a block needs to know about its parity class, worked out on the host for the first ncls classes (0: the
  // kernel derives it, a dozen block-uniform integer divisions that the compiler runs on the vector ALU)
  int ncls;
  struct Class {
    int cls_h, cls_w, off_h, off_w, Hc, Wc, nth, ntw, base_h, base_w;
    float inv_roww, inv_plane, inv_ntw;
  } cls[4];
  // wgrad over several (gy, x) pairs of one geometry in one launch (the passes of a training step that share a layer):
  // split z belongs to pair z / seg_splits and is split z % seg_splits of it; every split writes its own slab, so the
  // slab sum is the sum over all pairs. seg_splits == 0: one pair, A / B.
  const float* Aseg[4];
  const float* Bseg[4];
  int seg_splits;
};

// Fills p.cls for an input-gradient launch (strides whose sh * sw classes fit the table).
static inline void igemm_set_classes(IgemmParams& p) {
  p.ncls = 0;
  if (p.sh * p.sw > 4) return;
  for (int zz = 0; zz < p.sh * p.sw; ++zz) {
    IgemmParams::Class& c = p.cls[zz];
    c.cls_h = zz / p.sw;
    c.cls_w = zz - c.cls_h * p.sw;
    c.off_h = ((c.cls_h - p.ph) % p.sh + p.sh) % p.sh;
    c.off_w = ((c.cls_w - p.pw) % p.sw + p.sw) % p.sw;
    c.Hc = c.off_h < p.H ? (p.H - c.off_h + p.sh - 1) / p.sh : 0;
    c.Wc = c.off_w < p.W ? (p.W - c.off_w + p.sw - 1) / p.sw : 0;
    c.nth = c.cls_h < p.KH ? (p.KH - c.cls_h + p.sh - 1) / p.sh : 0;
    c.ntw = c.cls_w < p.KW ? (p.KW - c.cls_w + p.sw - 1) / p.sw : 0;
    c.base_h = (c.off_h + p.ph - c.cls_h) / p.sh;
    c.base_w = (c.off_w + p.pw - c.cls_w) / p.sw;
    const int plane = c.Hc * c.Wc;
    c.inv_roww = c.Wc > 0 ? 1.0f / (float)c.Wc : 0.f;
    c.inv_plane = plane > 0 ? 1.0f / (float)plane : 0.f;
    c.inv_ntw = c.ntw > 0 ? 1.0f / (float)c.ntw : 0.f;
  }
  p.ncls = p.sh * p.sw;
}

// Operand sizes for the kernel's range-checked loads; -1 when an operand cannot be addressed with a 31-bit byte offset.
static inline int igemm_set_operands(IgemmParams& p, long long a_elems, long long b_elems) {
  if (a_elems * 4 >= (1LL << 31) || b_elems * 4 >= (1LL << 31)) return -1;
  p.a_bytes = (unsigned)(a_elems * 4);
  p.b_bytes = (unsigned)(b_elems * 4);
  return 0;
}

#define IGEMM_BK 32
#define IGEMM_LDK 36

// STATS = 1: the epilogue also leaves, per row tile and column, the statistics a BatchNorm behind the convolution needs
// (a shift value, the sum of (y - shift) and the sum of (y - shift)^2 over the tile's valid rows, and the tile's row
// count), so the norm does not have to read the output again to get its batch mean and variance.
// STATS = 3 (input-gradient mode): the output is the gradient arriving at the output of a ReLU / LeakyReLU; it is
// multiplied by that activation's slope mask before it is stored (no sums).
// STATS = 2: the output is the gradient g arriving at the output of a BatchNorm (+ ReLU / LeakyReLU) whose input x is
// at hand: the epilogue leaves, per row tile and column, sum (g m) and sum (g m xhat) (m the activation's slope mask,
// xhat the normalised input), the two sums that norm's backward pass needs before it can form its input gradient.
template <int MODE, int BM, int BN, int WAVES_M, bool B_MC, int VA, int VB, int PREC = PREC_F32, int NW = 4, int STATS = 0>
// (eight-wave blocks are meant to run two to a CU, four waves per SIMD: the register allocator is told so)
__global__ void __launch_bounds__(NW * 64, NW == 8 ? 4 : 1) igemm_kernel(const IgemmParams p) {
  constexpr int NT = NW * 64;      // threads per block
  constexpr int RPT = NT / 8;      // rows one pass of the [row][k] loaders covers (8 threads x 16 bytes span the 32 k)
  constexpr int BK = IGEMM_BK, LDK = IGEMM_LDK;
  constexpr bool BF = PREC == PREC_BF16;
  constexpr int LDB = 40;   // bf16 elements per row of a [row][k] bf16 tile
  constexpr int WAVES_N = NW / WAVES_M;
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr bool A_MC = (MODE == MODE_WGRAD);
  constexpr bool B_IS_MC = B_MC || (MODE != MODE_FWD);
  static_assert(TM >= 1 && TN >= 1, "tile too small for the wave arrangement");
  // Products of the address arithmetic inside the k loop. A full 32-bit integer multiply runs at a quarter of the vector
  // rate (16 cycles per wave), and vector instructions do not overlap the matrix instructions of the same SIMD, not
  // even another wave's (measured: tools/probes/mfma_valu_overlap.hip), so the 8 to 14 products of a k-step cost as
  // much as two or three MFMAs. The eight-wave variants multiply 24-bit operands (one full-rate instruction, fusable
  // with the addition behind it); the host hands them only geometries whose factors are below 2^24 (igemm_small24)
  // and everything else to the four-wave variants.
  // (Written as an instruction: through __mul24 the compiler sees that only the low bits of an inner product are used
  // by the outer one, drops the 24-bit extensions and is back at v_mul_lo_u32. The second factor is block-uniform.)
  constexpr bool M24 = NW == 8 && !BF;
  auto mul = [](int a, int uniform) -> int {
    if constexpr (M24) {
      int r;
      asm("v_mul_i32_i24 %0, %2, %1" : "=v"(r) : "v"(a), "s"(uniform));
      return r;
    } else {
      return a * uniform;
    }
  };
  // a * uniform + c in one instruction (the compiler cannot fuse an addition into the assembly above)
  auto mad = [](int a, int uniform, int c) -> int {
    if constexpr (M24) {
      int r;
      asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(uniform), "v"(c));
      return r;
    } else {
      return a * uniform + c;
    }
  };

  // two staging buffers: tile t+1 is written while tile t is being read, one barrier per k-step
  constexpr int A_FLOATS = BF ? (A_MC ? (BK / 2) * BM : BM * LDB / 2) : (A_MC ? BK * BM : BM * LDK);
  constexpr int B_FLOATS = BF ? (B_IS_MC ? (BK / 2) * BN : BN * LDB / 2) : (B_IS_MC ? BK * BN : BN * LDK);
  constexpr int BUF = A_FLOATS + B_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;

  // XCD-aware block order. Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest, then z), and
  // each XCD has its own L2. Every XCD is given a contiguous run of the logical order "z-major, then tile" (bijective
  // for any grid size): for the weight gradient that is whole split-K chunks, whose tiles all read the same pixels, so
  // the activations stream through one L2 once instead of through all eight; for the forward mode a run of row tiles
  // with all their column tiles; for the input gradient "tile-major, then class".
  int mt, nt, zz;
  {
    const int tiles = gridDim.x, nblk = gridDim.x * gridDim.z;
    const int lin = blockIdx.x + gridDim.x * blockIdx.z;
    const int xcd = lin & 7, q = nblk >> 3, rem = nblk & 7;
    const int id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (lin >> 3);
    int t;
    if constexpr (MODE == MODE_DGRAD) {   // the parity classes of one row tile read the same gradient pixels: keep them together
      if (gridDim.z == 4) {
        t = id >> 2;
        zz = id & 3;
      } else {
        t = id / (int)gridDim.z;
        zz = id - t * (int)gridDim.z;
      }
    } else {
      zz = id / tiles;
      t = id - zz * tiles;
    }
    if (p.ntiles == 1) {
      mt = t;
      nt = 0;
    } else {
      mt = t / p.ntiles;
      nt = t - mt * p.ntiles;
    }
  }
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-mode extents -------------------------------------------------------------------
  int M = p.M, Ktot = p.Ktot;
  int kbeg = 0, kend = p.Ktot;
  const float *opA = p.A, *opB = p.B;          // this block's operands
  // dgrad parity class
  int cls_h = 0, cls_w = 0, off_h = 0, off_w = 0, Hc = 0, Wc = 0, nth = 0, ntw = 0, base_h = 0, base_w = 0;
  float tab_inv_roww = 0.f, tab_inv_plane = 0.f, tab_inv_ntw = 0.f;
  const bool tab = MODE == MODE_DGRAD && zz < p.ncls;        // the host has worked this class out
  if constexpr (MODE == MODE_DGRAD) {
    if (tab) {
      const IgemmParams::Class& c = p.cls[zz];
      cls_h = c.cls_h; cls_w = c.cls_w; off_h = c.off_h; off_w = c.off_w; Hc = c.Hc; Wc = c.Wc;
      nth = c.nth; ntw = c.ntw; base_h = c.base_h; base_w = c.base_w;
      tab_inv_roww = c.inv_roww; tab_inv_plane = c.inv_plane; tab_inv_ntw = c.inv_ntw;
    } else {
      cls_h = zz / p.sw;
      cls_w = zz - cls_h * p.sw;
      off_h = ((cls_h - p.ph) % p.sh + p.sh) % p.sh;
      off_w = ((cls_w - p.pw) % p.sw + p.sw) % p.sw;
      Hc = off_h < p.H ? (p.H - off_h + p.sh - 1) / p.sh : 0;
      Wc = off_w < p.W ? (p.W - off_w + p.sw - 1) / p.sw : 0;
      nth = cls_h < p.KH ? (p.KH - cls_h + p.sh - 1) / p.sh : 0;
      ntw = cls_w < p.KW ? (p.KW - cls_w + p.sw - 1) / p.sw : 0;
      base_h = (off_h + p.ph - cls_h) / p.sh;
      base_w = (off_w + p.pw - cls_w) / p.sw;
    }
    M = p.N * Hc * Wc;
    Ktot = nth * ntw * p.Co;
    kend = Ktot;
    if (m0 >= M) {        // block-uniform
      if constexpr (STATS == 1 || STATS == 2) {      // a tile of this class's surplus: it has no rows
        constexpr int NQ = STATS == 1 ? 3 : 2;
        if (nt == 0 && tid == 0) p.stats[(size_t)p.stats_tiles * NQ * p.Ncols + (zz * p.mtiles + mt)] = 0.0;
      }
      return;
    }
  } else if constexpr (MODE == MODE_WGRAD) {
    int zl = zz;
    if (p.seg_splits > 0) {
      const int seg = zz / p.seg_splits;
      zl = zz - seg * p.seg_splits;
      // (constant indices: a dynamically indexed array of a by-value kernel argument sends the whole structure to
      // scratch memory)
      opA = seg == 0 ? p.Aseg[0] : seg == 1 ? p.Aseg[1] : seg == 2 ? p.Aseg[2] : p.Aseg[3];
      opB = seg == 0 ? p.Bseg[0] : seg == 1 ? p.Bseg[1] : seg == 2 ? p.Bseg[2] : p.Bseg[3];
    }
    kbeg = zl * p.kchunk;
    kend = min(kbeg + p.kchunk, p.Ktot);
  } else {
    ntw = p.KW;
  }
  const int nk = (kend - kbeg + BK - 1) / BK;
  // The common cases in which the k-walk is block-uniform (explained where the state of each is set up, below): known
  // here so that the per-thread walk state, integer divisions included, is only worked out when it is needed.
  const bool full_steps = (kend - kbeg) % BK == 0;
  const bool ua_fast = !(MODE == MODE_WGRAD) && VA == 4 && MODE == MODE_DGRAD && p.Co % BK == 0 && full_steps && kbeg == 0;
  const bool dg_fast = (MODE == MODE_DGRAD) && VB == 4 && (p.Co % BK == 0);
  const int wg_R = (MODE == MODE_WGRAD && VA == 4 && VB == 4 && p.Wo > 0 && BK % p.Wo == 0) ? BK / p.Wo : 0;   // output rows per k-step
  const bool wg_fast = wg_R > 0 && (p.Ho % wg_R == 0 || wg_R % p.Ho == 0) && full_steps;

  // ---- A loader state -----------------------------------------------------------------------
  constexpr int RA = BM / RPT;  // rows per thread, [row][k] layout
  constexpr int TPRA = BM / 4, RPA = NT / TPRA, PA = BK / RPA;  // [k][row] layout
  int a_img[A_MC ? 1 : RA], a_h[A_MC ? 1 : RA], a_w[A_MC ? 1 : RA];
  unsigned a_base[A_MC ? 1 : RA];
  // row index -> (image, y, x): plane = pixels per image, roww = pixels per image row, of the tensor the rows walk
  const bool fast = p.fast_div != 0;
  const int roww = (MODE == MODE_FWD) ? p.Wo : Wc;
  const int plane = (MODE == MODE_FWD) ? p.Ho * p.Wo : Hc * Wc;
  float inv_roww, inv_plane;
  if (tab) {
    inv_roww = tab_inv_roww;
    inv_plane = tab_inv_plane;
  } else {
    inv_roww = roww > 0 ? 1.0f / (float)roww : 0.f;
    inv_plane = plane > 0 ? 1.0f / (float)plane : 0.f;
  }
  if constexpr (!A_MC) {
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int m = m0 + (tid >> 3) + RPT * i;
      a_img[i] = 0;
      a_w[i] = 0;
      a_h[i] = -(1 << 28);
      if (m < M) {
        if constexpr (MODE == MODE_FWD) {
          const int n = vg_div_row(m, plane, inv_plane, fast), rem = m - n * plane;
          const int ho = vg_div_row(rem, roww, inv_roww, fast), wo = rem - ho * roww;
          a_img[i] = n * p.H * p.W;
          a_h[i] = ho * p.sh - p.ph;
          a_w[i] = wo * p.sw - p.pw;
        } else {
          const int n = vg_div_row(m, plane, inv_plane, fast), rem = m - n * plane;
          const int ii = vg_div_row(rem, roww, inv_roww, fast), jj = rem - ii * roww;
          a_img[i] = n * p.Ho * p.Wo;
          a_h[i] = ii + base_h;
          a_w[i] = jj + base_w;
        }
      }
      // byte offset of the row's window origin (channel 0 of tap (0,0)); meaningless, and never used, for rows >= M
      const int roww_in = (MODE == MODE_FWD) ? p.W : p.Wo;
      a_base[i] = ((unsigned)(a_img[i] + a_h[i] * roww_in + a_w[i]) * (unsigned)((MODE == MODE_FWD) ? p.Ci : p.Co)) * 4u;
    }
  }

  // ---- B loader state -----------------------------------------------------------------------
  constexpr int RB = BN / RPT;
  constexpr int TPRB = BN / 4, RPB = NT / TPRB, PB = BK / RPB;
  static_assert(!BF || ((!A_MC || PA % 2 == 0) && (!B_IS_MC || PB % 2 == 0)), "bf16 [k][row] tiles pair adjacent k rows");
  // k row (inside a 32-deep step) of the i-th 16-byte load of a [k][row] tile; in bf16 mode loads 2j and 2j+1 of a
  // thread are the two rows of one packed pair
  auto krowA = [&](int i) -> int {
    if constexpr (BF) return 2 * (tid / TPRA) + (i & 1) + 2 * RPA * (i >> 1);
    else return tid / TPRA + RPA * i;
  };
  auto krowB = [&](int i) -> int {
    if constexpr (BF) return 2 * (tid / TPRB) + (i & 1) + 2 * RPB * (i >> 1);
    else return tid / TPRB + RPB * i;
  };
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

  // ---- k-walk state: the tile loaders are called for k-steps 0, 1, 2, ... in order and step this state by 32
  // instead of re-deriving (tap, channel) / (image, row, column) with integer divisions every k-step ----------
  int kpos = kbeg;                                    // first k of the tile the next load_tiles() call fetches
  // A of fwd/dgrad: this thread's 4-wide k chunk -> (linear tap index, channel inside the tap)
  int ak_c = 0, ak_t = 0;
  const int a_cch = (MODE == MODE_FWD) ? p.Ci : p.Co;
  const int a_steps = BK / a_cch, a_rem = BK - a_steps * a_cch;   // one k-step = a_steps whole taps + a_rem channels
  float inv_ntw;
  if (tab) inv_ntw = tab_inv_ntw;
  else inv_ntw = ntw > 0 ? 1.0f / (float)ntw : 0.f;
  if constexpr (!A_MC && VA == 4) {
    if (!ua_fast) {
      const int k4 = kbeg + 4 * (tid & 7);
      ak_t = k4 / a_cch;
      ak_c = k4 - ak_t * a_cch;
    }
  }
  // fwd / dgrad A, common case (channels per tap a multiple of 32, every k-step full): the 32 k of a k-step lie in one
  // tap, so the tap and the first channel of the step are block-uniform (scalar state stepped with the k-walk) and what
  // they add to a row's offset is one scalar; a thread's own share, its 16-byte channel chunk, is folded into a_base.
  constexpr bool UA = !A_MC && VA == 4;
  // (input-gradient mode only: measured on the forward kernels it gains nothing, 130.2 / 132.5 / 139.9 against
  // 129.9 / 132.6 / 138.0 us on the three DCGAN layers)
  int ua_c0 = 0, ua_ta = 0, ua_tb = 0;
  if constexpr (UA) {
    if (ua_fast) {
#pragma unroll
      for (int i = 0; i < RA; ++i) a_base[i] += 16u * (unsigned)(tid & 7);
    }
  }
  // dgrad weights B[k=(tap,co)][n]: one (linear tap, co) per k row owned by this thread
  int bk_co[(MODE == MODE_DGRAD) ? PB : 1], bk_t[(MODE == MODE_DGRAD) ? PB : 1];
  if constexpr (MODE == MODE_DGRAD) {
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      bk_t[i] = bk_co[i] = 0;
      if (!dg_fast) {
        const int k = kbeg + krowB(i);
        bk_t[i] = k / p.Co;
        bk_co[i] = k - bk_t[i] * p.Co;
      }
    }
  }
  // dgrad weights, common case Co % 32 == 0: the 32 k rows of a k-step lie in ONE tap, so (kh, kw) and what they add to
  // the offset are the same for the whole block: kept as block-uniform state (scalar registers, scalar arithmetic) and
  // stepped with the k-walk; a thread then forms a row's offset from its channel with one product
  // (co * KH*KW*Ci + (kh*KW + kw)*Ci + column). Other channel counts take the per-row decomposition.
  int dg_co0 = 0, dg_ta = 0, dg_tb = 0;                         // first channel of the k-step, its tap (row, column) in the class
  const int dg_costride4 = p.KH * p.KW * p.Ci * 4;              // bytes between consecutive co
  // wgrad x-window B[k=pixel][n]: one output pixel (image*Ho + ho, wo) per k row owned by this thread
  int bp_row[(MODE == MODE_WGRAD) ? PB : 1], bp_wo[(MODE == MODE_WGRAD) ? PB : 1];
  const int w_steps = (MODE == MODE_WGRAD) ? BK / p.Wo : 0, w_rem = (MODE == MODE_WGRAD) ? BK - w_steps * p.Wo : 0;
  const float inv_ho = 1.0f / (float)p.Ho;
  if constexpr (MODE == MODE_WGRAD) {
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      bp_row[i] = bp_wo[i] = 0;
      if (!wg_fast) {
        const int pix = kbeg + krowB(i);
        bp_row[i] = pix / p.Wo;
        bp_wo[i] = pix - bp_row[i] * p.Wo;
      }
    }
  }

  // wgrad, common case (the 32 pixels of a k-step are whole output rows, BK % Wo == 0, and those rows either lie in one
  // image or are whole images; every k-step of the block is full): image and first row of a k-step are block-uniform
  // (scalar state, stepped with the k-walk), and everything a thread adds to them is constant along k: its row and
  // column inside the step, hence its window column, that column's validity and most of its offset. A B load then
  // costs an addition, a comparison, one product and a select instead of a division and nine products and sums; an A
  // load one addition.
  constexpr bool WG = MODE == MODE_WGRAD && VA == 4 && VB == 4;
  int wg_n0 = 0, wg_ho0 = 0;                                    // image and row of the k-step's first pixel
  const int wg_step_n = wg_R > 0 ? wg_R / p.Ho : 0, wg_step_h = wg_R > 0 ? wg_R % p.Ho : 0;
  const int wg_wci4 = p.W * p.Ci * 4;                           // bytes between input rows
  int wg_hic[WG ? PB : 1], wg_cst[WG ? PB : 1], wg_ca[WG ? PA : 1];
  bool wg_wok[WG ? PB : 1];
  if constexpr (WG) {
    if (wg_fast) {
      const int row0 = kbeg / p.Wo;
      wg_n0 = row0 / p.Ho;
      wg_ho0 = row0 - wg_n0 * p.Ho;
      const float inv_wo = 1.0f / (float)p.Wo;
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int kr = krowB(i);
        const int ri = vg_div_small(kr, inv_wo), wo = kr - ri * p.Wo;          // row inside the step, output column
        const int dn = vg_div_small(ri, inv_ho), hoc = ri - dn * p.Ho;         // image inside the step, row inside it
        wg_hic[i] = hoc * p.sh - p.ph + b_ta[0];
        const int wi = wo * p.sw - p.pw + b_tb[0];
        wg_wok[i] = (unsigned)wi < (unsigned)p.W;
        wg_cst[i] = (((dn * p.H) * p.W + wi) * p.Ci + b_ci[0]) * 4;
      }
#pragma unroll
      for (int i = 0; i < PA; ++i) wg_ca[i] = (krowA(i) * p.Co + m0 + 4 * (tid % TPRA)) * 4;
    }
  }

  f32x4 ra[A_MC ? PA : RA];
  f32x4 rb[B_IS_MC ? PB : RB];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // ---- operand loads: buffer loads with 32-bit byte offsets --------------------------------------------------------
  // A load that must not happen (row or k outside the operand, a window position in the padding) is given an offset
  // with the top bit set: that is beyond any operand (both are smaller than 2 GiB), and the hardware returns zeros for
  // it. So a tile's loads are one straight-line sequence without branches, selects on 64-bit pointers or a dummy page.
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(opA), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(opB), 0, (int)p.b_bytes, 0x00020000);
  auto ld4 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned off) -> f32x4 {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
  };
  // 4-byte loads of operands that cannot be read as vectors (3-channel images, odd channel counts); elem = element index
  auto ld1 = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned elem, bool ok) -> float {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(ok ? elem * 4u : OOB), 0, 0));
  };

  // Fetching a tile is split in two. point_piece(q) works out the offsets of the 16-byte loads of the next tile, a few
  // loads per call: pure integer arithmetic, issued between the MFMAs of one slice (see k_step), because a wave that
  // runs through all of a tile's address arithmetic in one stretch feeds the matrix pipe nothing meanwhile, and the waves
  // of a block reach that stretch together. fetch_tiles() then issues the loads (after the staging registers have been
  // written to LDS) and steps the k-walk. Operands that need 4-byte loads do all their work in fetch_tiles().
  constexpr int NPA = VA == 4 ? (A_MC ? PA : RA) : 0, NPB = VB == 4 ? (B_IS_MC ? PB : RB) : 0, NP = NPA + NPB;
  constexpr int NPIECE = BF ? TM * TN : 4;     // MFMA groups of one slice the address work is spread over
  unsigned oa[NPA > 0 ? NPA : 1], ob[NPB > 0 ? NPB : 1];
  // row-invariant part of the fwd/dgrad A offsets of one k-step, set by the first piece
  int pt_ta = 0, pt_tb = 0;
  unsigned pt_tap = 0;
  bool pt_kvalid = false;
  const unsigned m4ok_mask = (A_MC && m0 + 4 * (tid % TPRA) < p.Co) ? 0u : OOB;   // wgrad A: this thread's 4 columns exist
  // fwd weights w[co][k], tile [n][k]: byte offset of this thread's rows (k-invariant)
  unsigned b_rowbase[(!B_IS_MC && VB == 4) ? RB : 1];
  if constexpr (!B_IS_MC && VB == 4) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int n = n0 + (tid >> 3) + RPT * i;
      b_rowbase[i] = n < p.Ncols ? (unsigned)n * (unsigned)Ktot * 4u : OOB;
    }
  }
  const unsigned bn4_mask = bn4 < p.Ncols ? 0u : OOB;

  auto point_a = [&](int i) {
    if constexpr (NPA > 0) {
      if constexpr (!A_MC) {
        const int hi = (MODE == MODE_FWD) ? a_h[i] + pt_ta : a_h[i] - pt_ta;
        const int wi = (MODE == MODE_FWD) ? a_w[i] + pt_tb : a_w[i] - pt_tb;
        const unsigned lim_h = (MODE == MODE_FWD) ? p.H : p.Ho, lim_w = (MODE == MODE_FWD) ? p.W : p.Wo;
        const bool ok = pt_kvalid & ((unsigned)hi < lim_h) & ((unsigned)wi < lim_w);
        oa[i] = ok ? a_base[i] + pt_tap : OOB;
      } else {
        // wgrad A = gy[pixel][co], tile [k][m]
        if (WG && wg_fast) {       // block-uniform
          oa[i] = (unsigned)(kpos * p.Co * 4 + wg_ca[i]) | m4ok_mask;
        } else {
          const int pix = kpos + krowA(i);
          const unsigned off = (unsigned)mad(pix, p.Co, m0 + 4 * (tid % TPRA)) * 4u;
          oa[i] = (pix < kend ? off : OOB) | m4ok_mask;
        }
      }
    }
  };
  auto point_b = [&](int i) {
    if constexpr (NPB > 0) {
      if constexpr (!B_IS_MC) {
        const int k4 = kpos + 4 * (tid & 7);
        ob[i] = (b_rowbase[i] + (unsigned)k4 * 4u) | (k4 < kend ? 0u : OOB);
      } else if constexpr (MODE == MODE_FWD) {
        // plain GEMM operand B[k][n] (row stride Ncols)
        const int k = kpos + krowB(i);
        ob[i] = (k < kend ? (unsigned)mad(k, p.Ncols, bn4) * 4u : OOB) | bn4_mask;
      } else if constexpr (MODE == MODE_DGRAD) {
        // weights w[co][kh][kw][ci] read as B[k=(tap,co)][n=ci]
        if (dg_fast) {       // block-uniform
          const int tap4 = (((cls_h + p.sh * dg_ta) * p.KW + cls_w + p.sw * dg_tb) * p.Ci) * 4;   // scalar
          ob[i] = (unsigned)mad(dg_co0 + krowB(i), dg_costride4, tap4 + 4 * bn4) | bn4_mask;      // (every k row exists)
        } else {
          const int k = kpos + krowB(i);
          const int bta = vg_div_small(bk_t[i], inv_ntw), btb = mad(bta, -ntw, bk_t[i]);
          const int kh = mad(bta, p.sh, cls_h), kw = mad(btb, p.sw, cls_w);
          const unsigned off = (unsigned)mad(mad(mad(bk_co[i], p.KH, kh), p.KW, kw), p.Ci, bn4) * 4u;
          ob[i] = (k < kend ? off : OOB) | bn4_mask;
        }
      } else {
        // wgrad B = x window, tile [k=pixel][n=(kh,kw,ci)]
        if (WG && wg_fast) {       // block-uniform
          const int hi = wg_ho0 * p.sh + wg_hic[i];
          const bool ok = wg_wok[i] & ((unsigned)hi < (unsigned)p.H);
          ob[i] = ok ? (unsigned)(mad(hi, wg_wci4, wg_cst[i]) + wg_n0 * (p.H * wg_wci4)) : OOB;
        } else {
          const int pix = kpos + krowB(i);
          const int bpn = vg_div_small(bp_row[i], inv_ho), bpho = mad(bpn, -p.Ho, bp_row[i]);
          const int hi = mad(bpho, p.sh, b_ta[0] - p.ph), wi = mad(bp_wo[i], p.sw, b_tb[0] - p.pw);
          const bool ok = (pix < kend) & ((unsigned)hi < (unsigned)p.H) & ((unsigned)wi < (unsigned)p.W);
          ob[i] = ok ? (unsigned)mad(mad(mad(bpn, p.H, hi), p.W, wi), p.Ci, b_ci[0]) * 4u : OOB;
        }
      }
    }
  };
  auto point_piece = [&](int q) {
    if constexpr (NPA > 0 && !A_MC) {
      if (q == 0) {
        if (ua_fast) {       // block-uniform: scalar arithmetic
          pt_kvalid = true;
          pt_ta = ua_ta;
          pt_tb = ua_tb;
          if constexpr (MODE == MODE_FWD) pt_tap = (unsigned)(((ua_ta * p.W + ua_tb) * p.Ci + ua_c0) * 4);
          else pt_tap = (unsigned)((ua_c0 - (ua_ta * p.Wo + ua_tb) * p.Co) * 4);
        } else {
          pt_kvalid = kpos + 4 * (tid & 7) < kend;
          pt_ta = vg_div_small(ak_t, inv_ntw);
          pt_tb = mad(pt_ta, -ntw, ak_t);
          if constexpr (MODE == MODE_FWD) pt_tap = (unsigned)mad(mad(pt_ta, p.W, pt_tb), p.Ci, ak_c) * 4u;
          else pt_tap = (unsigned)mad(mad(pt_ta, p.Wo, pt_tb), -p.Co, ak_c) * 4u;
        }
      }
    }
#pragma unroll
    for (int idx = 0; idx < NP; ++idx) {
      if (idx * NPIECE / (NP > 0 ? NP : 1) == q) {
        if (idx < NPA) point_a(idx);
        else point_b(idx - NPA);
      }
    }
  };
  auto point_tiles = [&]() {
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) point_piece(q);
  };

  auto fetch_tiles = [&]() {
    const int kb = kpos;
    // ---------------- A ----------------
    if constexpr (VA == 4) {
#pragma unroll
      for (int i = 0; i < NPA; ++i) ra[i] = ld4(rsrcA, oa[i]);
    } else if constexpr (!A_MC) {
      const int k4 = kb + 4 * (tid & 7);
      const int Cch = (MODE == MODE_FWD) ? p.Ci : p.Co;
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
        f32x4 v = zero4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if constexpr (MODE == MODE_FWD) {
            const int hi = a_h[i] + ta[e], wi = a_w[i] + tb[e];
            v[e] = ld1(rsrcA, (unsigned)(a_img[i] + hi * p.W + wi) * (unsigned)p.Ci + (unsigned)c[e],
                       kv[e] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W);
          } else {
            const int ho = a_h[i] - ta[e], wo = a_w[i] - tb[e];
            v[e] = ld1(rsrcA, (unsigned)(a_img[i] + ho * p.Wo + wo) * (unsigned)p.Co + (unsigned)c[e],
                       kv[e] && (unsigned)ho < (unsigned)p.Ho && (unsigned)wo < (unsigned)p.Wo);
          }
        }
        ra[i] = v;
      }
    } else {
      const int m4 = m0 + 4 * (tid % TPRA);
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int pix = kb + krowA(i);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = ld1(rsrcA, (unsigned)pix * (unsigned)p.Co + (unsigned)(m4 + e), pix < kend && m4 + e < p.Co);
        ra[i] = v;
      }
    }
    // ---------------- B ----------------
    if constexpr (VB == 4) {
#pragma unroll
      for (int i = 0; i < NPB; ++i) rb[i] = ld4(rsrcB, ob[i]);
    } else if constexpr (!B_IS_MC) {
      const int k4 = kb + 4 * (tid & 7);
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const int n = n0 + (tid >> 3) + RPT * i;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = ld1(rsrcB, (unsigned)n * (unsigned)Ktot + (unsigned)(k4 + e), n < p.Ncols && k4 + e < kend);
        rb[i] = v;
      }
    } else if constexpr (MODE == MODE_FWD) {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int k = kb + krowB(i);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[e] = ld1(rsrcB, (unsigned)k * (unsigned)p.Ncols + (unsigned)(bn4 + e), k < kend && bn4 + e < p.Ncols);
        rb[i] = v;
      }
    } else if constexpr (MODE == MODE_DGRAD) {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int k = kb + krowB(i);
        const int bta = vg_div_small(bk_t[i], inv_ntw), btb = bk_t[i] - bta * ntw;
        const int kh = cls_h + p.sh * bta, kw = cls_w + p.sw * btb;
        const unsigned base = (unsigned)((bk_co[i] * p.KH + kh) * p.KW + kw) * (unsigned)p.Ci;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ld1(rsrcB, base + (unsigned)(bn4 + e), k < kend && bn4 + e < p.Ncols);
        rb[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int pix = kb + krowB(i);
        const int bpn = vg_div_small(bp_row[i], inv_ho), bpho = bp_row[i] - bpn * p.Ho;
        const int hb = bpho * p.sh - p.ph, wb = bp_wo[i] * p.sw - p.pw;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int hi = hb + b_ta[e], wi = wb + b_tb[e];
          v[e] = ld1(rsrcB, (unsigned)((bpn * p.H + hi) * p.W + wi) * (unsigned)p.Ci + (unsigned)b_ci[e],
                     pix < kend && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W);
        }
        rb[i] = v;
      }
    }
    // ---------------- advance the k-walk by one tile (branch-free) ----------------
    kpos += BK;
    if constexpr (!A_MC && VA == 4) {
      if (ua_fast) {         // block-uniform, scalar
        ua_c0 += BK;
        if (ua_c0 >= a_cch) {
          ua_c0 = 0;
          if (++ua_tb == ntw) {
            ua_tb = 0;
            ++ua_ta;
          }
        }
      } else {
        ak_c += a_rem;
        const int carry = ak_c >= a_cch ? 1 : 0;
        ak_c -= carry ? a_cch : 0;
        ak_t += a_steps + carry;
      }
    }
    if constexpr (MODE == MODE_DGRAD) {
      if (dg_fast) {         // block-uniform, scalar
        dg_co0 += BK;
        if (dg_co0 >= p.Co) {
          dg_co0 = 0;
          if (++dg_tb == ntw) {
            dg_tb = 0;
            ++dg_ta;
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          bk_co[i] += a_rem;
          const int carry = bk_co[i] >= a_cch ? 1 : 0;
          bk_co[i] -= carry ? a_cch : 0;
          bk_t[i] += a_steps + carry;
        }
      }
    }
    if constexpr (MODE == MODE_WGRAD) {
      if (WG && wg_fast) {       // block-uniform, scalar
        wg_ho0 += wg_step_h;
        if (wg_ho0 >= p.Ho) {
          wg_ho0 -= p.Ho;
          ++wg_n0;
        }
        wg_n0 += wg_step_n;
      } else {
#pragma unroll
        for (int i = 0; i < PB; ++i) {
          bp_wo[i] += w_rem;
          const int carry = bp_wo[i] >= p.Wo ? 1 : 0;
          bp_wo[i] -= carry ? p.Wo : 0;
          bp_row[i] += w_steps + carry;
        }
      }
    }
  };

  auto store_tiles = [&](int buf) {
    float* As = smem + buf * BUF;
    float* Bs = As + A_FLOATS;
    if constexpr (!BF) {
      if constexpr (!A_MC) {
#pragma unroll
        for (int i = 0; i < RA; ++i)
          *reinterpret_cast<f32x4*>(&As[((tid >> 3) + RPT * i) * LDK + 4 * (tid & 7)]) = ra[i];
      } else {
#pragma unroll
        for (int i = 0; i < PA; ++i)
          *reinterpret_cast<f32x4*>(&As[(tid / TPRA + RPA * i) * BM + 4 * (tid % TPRA)]) = ra[i];
      }
      if constexpr (!B_IS_MC) {
#pragma unroll
        for (int i = 0; i < RB; ++i)
          *reinterpret_cast<f32x4*>(&Bs[((tid >> 3) + RPT * i) * LDK + 4 * (tid & 7)]) = rb[i];
      } else {
#pragma unroll
        for (int i = 0; i < PB; ++i)
          *reinterpret_cast<f32x4*>(&Bs[(tid / TPRB + RPB * i) * BN + 4 * (tid % TPRB)]) = rb[i];
      }
    } else {
      auto pack = [](float lo, float hi) -> unsigned int {
        const bf16x2 v = {(__bf16)lo, (__bf16)hi};
        return __builtin_bit_cast(unsigned int, v);
      };
      if constexpr (!A_MC) {
        __bf16* Ah = reinterpret_cast<__bf16*>(As);
#pragma unroll
        for (int i = 0; i < RA; ++i)
          *reinterpret_cast<bf16x4*>(&Ah[((tid >> 3) + RPT * i) * LDB + 4 * (tid & 7)]) = __builtin_convertvector(ra[i], bf16x4);
      } else {
        unsigned int* Au = reinterpret_cast<unsigned int*>(As);
#pragma unroll
        for (int i = 0; i + 1 < PA; i += 2) {
          const u32x4 v = {pack(ra[i][0], ra[i + 1][0]), pack(ra[i][1], ra[i + 1][1]), pack(ra[i][2], ra[i + 1][2]),
                           pack(ra[i][3], ra[i + 1][3])};
          *reinterpret_cast<u32x4*>(&Au[(krowA(i) >> 1) * BM + 4 * (tid % TPRA)]) = v;
        }
      }
      if constexpr (!B_IS_MC) {
        __bf16* Bh = reinterpret_cast<__bf16*>(Bs);
#pragma unroll
        for (int i = 0; i < RB; ++i)
          *reinterpret_cast<bf16x4*>(&Bh[((tid >> 3) + RPT * i) * LDB + 4 * (tid & 7)]) = __builtin_convertvector(rb[i], bf16x4);
      } else {
        unsigned int* Bu = reinterpret_cast<unsigned int*>(Bs);
#pragma unroll
        for (int i = 0; i + 1 < PB; i += 2) {
          const u32x4 v = {pack(rb[i][0], rb[i + 1][0]), pack(rb[i][1], rb[i + 1][1]), pack(rb[i][2], rb[i + 1][2]),
                           pack(rb[i][3], rb[i + 1][3])};
          *reinterpret_cast<u32x4*>(&Bu[(krowB(i) >> 1) * BN + 4 * (tid % TPRB)]) = v;
        }
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int mi = 0; mi < TM; ++mi)
#pragma unroll
    for (int ni = 0; ni < TN; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  // Fragments of one slice of a staged tile (f32: 8 deep, 4 per k-step; bf16: 16 deep, 2 per k-step) and the MFMAs
  // that consume them. The two are separate so the reads of slice s+1 are in flight while the MFMAs of slice s run
  // (two fragment register sets): a block that is alone on its CU has no other wave to hide the LDS latency behind.
  using Frag = std::conditional_t<BF, bf16x8, f32x4>;
  auto load_frags = [&](const float* As, const float* Bs, int s, Frag (&af)[TM], Frag (&bf)[TN]) {
    if constexpr (!BF) {
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
    } else {
      // lane (r, h) feeds k = 16 s + 8 h + 0..7
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int row = (wm * TM + mi) * 32 + r;
        if constexpr (!A_MC) {
          af[mi] = *reinterpret_cast<const bf16x8*>(&reinterpret_cast<const __bf16*>(As)[row * LDB + 16 * s + 8 * h]);
        } else {
          const unsigned int* Au = reinterpret_cast<const unsigned int*>(As);
          u32x4 d;
#pragma unroll
          for (int j = 0; j < 4; ++j) d[j] = Au[(8 * s + 4 * h + j) * BM + row];
          af[mi] = __builtin_bit_cast(bf16x8, d);
        }
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int col = (wn * TN + ni) * 32 + r;
        if constexpr (!B_IS_MC) {
          bf[ni] = *reinterpret_cast<const bf16x8*>(&reinterpret_cast<const __bf16*>(Bs)[col * LDB + 16 * s + 8 * h]);
        } else {
          const unsigned int* Bu = reinterpret_cast<const unsigned int*>(Bs);
          u32x4 d;
#pragma unroll
          for (int j = 0; j < 4; ++j) d[j] = Bu[(8 * s + 4 * h + j) * BN + col];
          bf[ni] = __builtin_bit_cast(bf16x8, d);
        }
      }
    }
  };
  // MFMAs of one slice, in NPIECE groups (f32: the four k sub-steps; bf16: one 32x32 tile each)
  auto mma_piece = [&](const Frag (&af)[TM], const Frag (&bf)[TN], int q) {
    if constexpr (!BF) {
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][q], bf[ni][q], acc[mi][ni], 0, 0, 0);
    } else {
      const int mi = q / TN, ni = q % TN;
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    }
  };
  auto mma_frags = [&](const Frag (&af)[TM], const Frag (&bf)[TN]) {
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) mma_piece(af, bf, q);
  };
  // the same with the address arithmetic of the next tile's loads issued between the groups: an MFMA occupies the
  // matrix pipe for 16 passes, during which the wave can issue that much other work without the pipe running dry
  auto mma_frags_pointing = [&](const Frag (&af)[TM], const Frag (&bf)[TN]) {
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) {
      point_piece(q);
      mma_piece(af, bf, q);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  Frag afx[TM], bfx[TN], afy[TM], bfy[TN];

  // One k-step on buffer `buf`, without a branch: every k-step stages "the tile after next" even when there is none
  // (its loads are predicated off by k >= kend and read the zero page; the LDS buffer they are written to is not read
  // again), so the whole step is one scheduling region per barrier interval. On entry afx/bfx hold slice 0 of `buf` and
  // the staging registers hold tile kt+1, whose global loads were issued most of a k-step ago. The addresses of tile
  // kt+2 are worked out under the MFMAs of one slice; then tile kt+1 is written to the other buffer (last read one
  // k-step ago, behind a barrier) and the loads of tile kt+2 are issued into the freed registers. After the barrier
  // the first slice of the next buffer is fetched under the MFMAs of the last slice.
  auto k_step = [&](int buf) {
    const float* As = smem + buf * BUF;
    const float* Bs = As + A_FLOATS;
    load_frags(As, Bs, 1, afy, bfy);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (BF) {
      mma_frags_pointing(afx, bfx);
    } else {
      mma_frags(afx, bfx);
      load_frags(As, Bs, 2, afx, bfx);
      __builtin_amdgcn_sched_barrier(0);
      mma_frags_pointing(afy, bfy);
    }
    store_tiles(buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);   // the loads reuse the staging registers: hoisted above the LDS writes they would need a second set
    fetch_tiles();
    if constexpr (!BF) {
      load_frags(As, Bs, 3, afy, bfy);
      __builtin_amdgcn_sched_barrier(0);
      mma_frags(afx, bfx);
    }
    __syncthreads();
    {
      const float* An = smem + (buf ^ 1) * BUF;
      load_frags(An, An + A_FLOATS, 0, afx, bfx);
    }
    __builtin_amdgcn_sched_barrier(0);
    mma_frags(afy, bfy);
  };

  // element offset of row `row` of this block's row space in the output tensor (NO_ROW past the last row); in the
  // input-gradient mode the rows are the pixels of one parity class, scattered over the output image
  constexpr unsigned NO_ROW = 0xffffffffu;      // offsets stay below 2^31 (tensor size limit of the geometry check)
  auto out_row_offset = [&](int row) -> unsigned {
    if (row >= M) return NO_ROW;
    if constexpr (MODE == MODE_DGRAD) {
      const int n = vg_div_row(row, plane, inv_plane, fast), rem = row - n * plane;
      const int ii = vg_div_row(rem, roww, inv_roww, fast), jj = rem - ii * roww;
      return (unsigned)((n * p.H + ii * p.sh + off_h) * p.W + jj * p.sw + off_w) * (unsigned)p.Ncols;
    } else {
      return (unsigned)row * (unsigned)p.Ncols;
    }
  };
  // STATS = 2, 3: the block's tile of the side tensor bx (laid out like the output) is fetched with 16-byte loads (rows
  // whole, four columns a thread) issued one at a time early in the k-walk, where they travel with the operand tiles
  // and nothing waits for them on their own; the epilogue hands them round through LDS (STATS = 3 keeps one bit per
  // element, value > 0, and shrinks the values to that right after the walk).
  constexpr bool SIDE = (STATS == 3 && MODE == MODE_DGRAD) || (STATS == 2 && MODE != MODE_WGRAD);
  constexpr int SIDE_C4 = BN / 4, SIDE_PER = SIDE ? BM * SIDE_C4 / NT : 1;
  // (the largest tile has no registers to carry the values through the walk: it reads them in the epilogue)
  constexpr bool SIDE_EARLY = SIDE && SIDE_PER <= 4;
  unsigned mask_bits = 0;
  float4 side_rv[SIDE_PER];
  auto fetch_side_one = [&](int k, auto&& row_offset_of) {
    const int idx = tid + k * NT, rr = idx / SIDE_C4, col = n0 + (idx - rr * SIDE_C4) * 4;
    const unsigned o = row_offset_of(rr);
    side_rv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (o != NO_ROW && col < p.Ncols) side_rv[k] = *reinterpret_cast<const float4*>(p.bx + (size_t)o + col);
  };
  auto fetch_side = [&](auto&& row_offset_of) {
#pragma unroll
    for (int k = 0; k < SIDE_PER; ++k) fetch_side_one(k, row_offset_of);
  };
  auto early_row = [&](int rr) { return out_row_offset(m0 + rr); };
  auto pack_mask = [&]() {
#pragma unroll
    for (int k = 0; k < SIDE_PER; ++k)
      mask_bits |= (unsigned)((side_rv[k].x > 0.f ? 1 : 0) | (side_rv[k].y > 0.f ? 2 : 0) | (side_rv[k].z > 0.f ? 4 : 0) |
                              (side_rv[k].w > 0.f ? 8 : 0)) << (4 * k);
  };
  static_assert(!SIDE || (BM * SIDE_C4 % NT == 0 && SIDE_PER <= 8), "tile not a multiple of the block");
  if (nk > 0) {
    point_tiles();
    fetch_tiles();
    store_tiles(0);
    point_tiles();
    fetch_tiles();
  }
  __syncthreads();
  if (nk > 0) load_frags(smem, smem + A_FLOATS, 0, afx, bfx);

  int kt = 0;
  for (; kt + 1 < nk; kt += 2) {
    if constexpr (SIDE_EARLY) {      // one 16-byte load of the side tile every other k-step
#pragma unroll
      for (int k = 0; k < SIDE_PER; ++k)
        if (kt == 2 * k) fetch_side_one(k, early_row);
    }
    k_step(0);
    k_step(1);
  }
  if (kt < nk) k_step(0);
  if constexpr (SIDE_EARLY) {
    if (nk < 2 * SIDE_PER) fetch_side(early_row);      // a k-walk too short to have issued them all
    if constexpr (STATS == 3) pack_mask();
  }

  // ---- epilogue: lane holds column (lane&31) and rows (e&3)+8*(e>>2)+4*(lane>>5) of each 32x32 tile ----
  float* outp = p.out;
  if constexpr (MODE == MODE_WGRAD) outp += (size_t)zz * p.slab_stride;
  // Per column tile: validity and the bias value, fetched once and pinned here. A load whose first use sits inside the
  // conditionally executed store blocks below gives every one of them its own wait for all outstanding memory
  // operations, which also waits for the store before it: the stores of a tile would leave one by one.
  bool cok[TN];
  float bcol[TN];
  bool has_bias = false;
#pragma unroll
  for (int ni = 0; ni < TN; ++ni) {
    const int col = n0 + (wn * TN + ni) * 32 + r;
    cok[ni] = col < p.Ncols;
    bcol[ni] = 0.f;
  }
  if constexpr (MODE != MODE_WGRAD) {
    has_bias = p.bias != nullptr;
    if (has_bias) {
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int col = n0 + (wn * TN + ni) * 32 + r;
        if (cok[ni]) bcol[ni] = p.bias[col % p.bias_mod];
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) asm volatile("" ::"v"(bcol[ni]));
    }
  }
  if constexpr (STATS == 1) {
    // The shift is the tile's first row (always a valid one), so the sums are of deviations of the size of the spread,
    // not of the mean; lanes add their 16 * TM rows in f32, the wave halves and the waves of a column are added in a
    // fixed order, and the tile's three numbers per column leave as doubles.
    const int tile = (MODE == MODE_DGRAD) ? zz * p.mtiles + mt : mt;
    float* sh = smem;               // [BN]
    float* part = smem + BN;        // [WAVES_M][BN][2]
    __syncthreads();                // every wave is done with the staged tiles
    if (wm == 0 && h == 0) {
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) sh[(wn * TN + ni) * 32 + r] = acc[0][ni][0] + (has_bias ? bcol[ni] : 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const float s = sh[(wn * TN + ni) * 32 + r];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int row_t = m0 + (wm * TM + mi) * 32 + 4 * h;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          if (row_t + (e & 3) + 8 * (e >> 2) < M) {
            const float d = acc[mi][ni][e] + (has_bias ? bcol[ni] : 0.f) - s;
            s1 += d;
            s2 = fmaf(d, d, s2);
          }
        }
      }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (h == 0) {
        part[(wm * BN + (wn * TN + ni) * 32 + r) * 2] = s1;
        part[(wm * BN + (wn * TN + ni) * 32 + r) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Ncols) {
      double S1 = 0.0, S2 = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) {
        S1 += (double)part[(w * BN + tid) * 2];
        S2 += (double)part[(w * BN + tid) * 2 + 1];
      }
      double* st = p.stats + (size_t)tile * 3 * p.Ncols + n0 + tid;
      st[0] = (double)sh[tid];
      st[p.Ncols] = S1;
      st[2 * (size_t)p.Ncols] = S2;
    }
    if (tid == 0 && nt == 0) p.stats[(size_t)p.stats_tiles * 3 * p.Ncols + tile] = (double)min(BM, M - m0);
  }
  float* colp = outp + n0 + wn * TN * 32 + r;
  // dgrad: the rows of a tile are pixels of one parity class, scattered over the output image. Each row's element
  // offset is worked out once per block (two divisions per row) and left in LDS, which the k loop no longer needs,
  // instead of thirty-two times per lane inside the store loop.
  unsigned* row_off = reinterpret_cast<unsigned*>(smem);
  if constexpr (MODE == MODE_DGRAD) {
    __syncthreads();                            // every wave is done with the staged tiles
    for (int rr = tid; rr < BM; rr += NT) row_off[rr] = out_row_offset(m0 + rr);
    if constexpr (STATS == 3 && !SIDE_EARLY) {
      __syncthreads();
      fetch_side([&](int rr) { return row_off[rr]; });
      pack_mask();
    }
    if constexpr (STATS == 3) {
      unsigned char* mbits = reinterpret_cast<unsigned char*>(smem + BM);      // [BM][BN / 4], behind the row offsets
#pragma unroll
      for (int k = 0; k < SIDE_PER; ++k) mbits[tid + k * NT] = (unsigned char)((mask_bits >> (4 * k)) & 15u);
    }
    __syncthreads();
  }
  if constexpr (STATS == 2 && MODE != MODE_WGRAD) {
    const int tile = (MODE == MODE_DGRAD) ? zz * p.mtiles + mt : mt;
    constexpr int LDX = BN + 4;
    static_assert(BM + BM * LDX <= 2 * BUF && 2 * WAVES_M * BN <= BM * LDX, "the staging buffers cannot hold the side tile");
    float* xt = smem + BM;            // the block's tile of the norm's input, [BM][LDX], behind the row offsets
    if constexpr (MODE != MODE_DGRAD) __syncthreads();       // every wave is done with the staged tiles
    if constexpr (!SIDE_EARLY) fetch_side([&](int rr) { return MODE == MODE_DGRAD ? row_off[rr] : out_row_offset(m0 + rr); });
#pragma unroll
    for (int k = 0; k < SIDE_PER; ++k) {
      const int idx = tid + k * NT, rr = idx / SIDE_C4;
      *reinterpret_cast<float4*>(&xt[rr * LDX + (idx - rr * SIDE_C4) * 4]) = side_rv[k];
    }
    float mu[TN], rs[TN], ga[TN], be[TN], s1[TN], s2[TN];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      const int col = n0 + (wn * TN + ni) * 32 + r;
      mu[ni] = rs[ni] = ga[ni] = be[ni] = s1[ni] = s2[ni] = 0.f;
      if (cok[ni]) {
        mu[ni] = p.bmean[col];
        rs[ni] = p.binvstd[col];
        if (p.bgamma) ga[ni] = p.bgamma[col];
        if (p.bbeta) be[ni] = p.bbeta[col];
      }
    }
    __syncthreads();
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rr = (wm * TM + mi) * 32 + 4 * h + (e & 3) + 8 * (e >> 2);
        if (m0 + rr < M) {
#pragma unroll
          for (int ni = 0; ni < TN; ++ni) {
            if (cok[ni]) {
              const float xv = xt[rr * LDX + (wn * TN + ni) * 32 + r];
              float g = acc[mi][ni][e];
              if (has_bias) g += bcol[ni];
              if (p.bact != VG_ACT_NONE)
                g *= vg_mask_grad(vg_bn_affine(xv, mu[ni], rs[ni], p.bgamma, ga[ni], p.bbeta, be[ni]), p.bact, p.bslope);
              s1[ni] += g;
              s2[ni] = fmaf(g, (xv - mu[ni]) * rs[ni], s2[ni]);
            }
          }
        }
      }
    }
    __syncthreads();                  // the tile has been read: its space takes the waves' partial sums
    float* part = xt;                 // [WAVES_M][BN][2]
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      s1[ni] += __shfl_xor(s1[ni], 32, 64);
      s2[ni] += __shfl_xor(s2[ni], 32, 64);
      if (h == 0) {
        part[(wm * BN + (wn * TN + ni) * 32 + r) * 2] = s1[ni];
        part[(wm * BN + (wn * TN + ni) * 32 + r) * 2 + 1] = s2[ni];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Ncols) {
      double S1 = 0.0, S2 = 0.0;
#pragma unroll
      for (int w = 0; w < WAVES_M; ++w) {
        S1 += (double)part[(w * BN + tid) * 2];
        S2 += (double)part[(w * BN + tid) * 2 + 1];
      }
      double* st = p.stats + (size_t)tile * 2 * p.Ncols + n0 + tid;
      st[0] = S1;
      st[p.Ncols] = S2;
    }
    if (tid == 0 && nt == 0) p.stats[(size_t)p.stats_tiles * 2 * p.Ncols + tile] = (double)min(BM, M - m0);
  }
  if constexpr (STATS == 3 && MODE == MODE_DGRAD) {
    // The output is a gradient arriving at the output of a ReLU / LeakyReLU (bx): multiply by the activation's slope at
    // each position, so what is stored is the gradient in front of it. Every lane picks, from the bits left in LDS
    // above, those of the elements its accumulators hold.
    const unsigned char* mbits = reinterpret_cast<const unsigned char*>(smem + BM);
    const float low = p.bact == VG_ACT_RELU ? 0.f : p.bslope;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int rr = (wm * TM + mi) * 32 + 4 * h + (e & 3) + 8 * (e >> 2);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
          const int c = (wn * TN + ni) * 32 + r;
          const bool up = (mbits[rr * SIDE_C4 + (c >> 2)] >> (c & 3)) & 1;
          if (has_bias) acc[mi][ni][e] += bcol[ni];
          acc[mi][ni][e] *= up ? 1.f : low;
        }
      }
    }
    has_bias = false;
  }
  // dgrad: the lane's row offsets come out of LDS in one go, four at a time (rows 4 h + 8 q + 0..3 are neighbours), ahead
  // of the store loop: read one by one inside it, every store waited for an LDS round trip of its own
  unsigned offs[MODE == MODE_DGRAD ? TM : 1][16];
  if constexpr (MODE == MODE_DGRAD) {
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint4 t = *reinterpret_cast<const uint4*>(&row_off[(wm * TM + mi) * 32 + 4 * h + 8 * q]);
        offs[mi][4 * q] = t.x; offs[mi][4 * q + 1] = t.y; offs[mi][4 * q + 2] = t.z; offs[mi][4 * q + 3] = t.w;
      }
  }
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int row_t = m0 + (wm * TM + mi) * 32 + 4 * h;      // first of the lane's 16 rows of this tile
    float* tilep = colp;
    if constexpr (MODE != MODE_DGRAD) tilep += (size_t)row_t * p.Ncols;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int dr = (e & 3) + 8 * (e >> 2);
      const int row = row_t + dr;
      float* rp;
      if constexpr (MODE == MODE_DGRAD) {
        const unsigned off = offs[mi][e];
        if (off == NO_ROW) continue;
        rp = tilep + off;
      } else {
        if (row >= M) continue;
        rp = tilep + dr * p.Ncols;
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        if (cok[ni]) {
          float v = acc[mi][ni][e];
          if (has_bias) v += bcol[ni];
          rp[ni * 32] = v;
        }
      }
    }
  }
}

static inline void igemm_set_norm_bwd(IgemmParams& p, const VgNormBwdSrc& nb) {
  p.bx = nb.x; p.bmean = nb.mean; p.binvstd = nb.invstd; p.bgamma = nb.gamma; p.bbeta = nb.beta;
  p.bact = nb.act; p.bslope = nb.slope;
}

// Whether every factor of the kernels' address products is below 2^23 (the eight-wave f32 variants multiply with
// v_mul_i32_i24): pixel counts of both tensors, weights per input / output channel, the extents of the GEMM.
static inline bool igemm_small24(long long N, long long H, long long W, long long Ci, long long Co, long long KH,
                                 long long KW, long long Ho, long long Wo, long long Ktot = 0, long long Ncols = 0) {
  const long long lim = 1LL << 23;
  return N * H * W < lim && N * Ho * Wo < lim && Co * KH * KW < lim && Ci * KH * KW * 4 < lim && W * Ci * 4 < lim && Ktot < lim && Ncols < lim;
}
static inline bool igemm_small24(const IgemmParams& p) {
  return igemm_small24(p.N, p.H, p.W, p.Ci, p.Co, p.KH, p.KW, p.Ho, p.Wo, p.Ktot, p.Ncols);
}

// Waves per block of the variant that serves (bm, bn, va, vb, prec); small24: igemm_small24 of the problem.
static inline int igemm_waves(int bm, int bn, int va, int vb, int prec, bool small24) {
  const bool large = (bm == 128 && bn == 128) || (bm == 128 && bn == 64) || (bm == 64 && bn == 128);
  return prec == PREC_F32 && va == 4 && vb == 4 && large && small24 ? 8 : 4;
}

// Launch one (BM, BN, VA, VB) variant chosen at run time; the wave arrangement follows the tile. prec = PREC_BF16 is
// served for 16-byte-vector operands on the tiles that keep [k][row] pairs whole, anything else runs in f32.
template <int MODE, bool B_MC>
static inline int igemm_launch(const IgemmParams& p, int bm, int bn, int va, int vb, dim3 grid, hipStream_t st,
                               int prec = PREC_F32) {
  if (prec == PREC_SPLIT) prec = PREC_F32;   // shapes the plane GEMM does not serve
#define IGEMM_CASE_P(BM_, BN_, WM_, VA_, VB_, P_)                                                          \
  if (bm == BM_ && bn == BN_ && va == VA_ && vb == VB_) {                                                  \
    vg_launch_timed(igemm_kernel<MODE, BM_, BN_, WM_, B_MC, VA_, VB_, P_>, grid, dim3(256), 0, st, p); \
    return 0;                                                                                              \
  }
  // f32, 16-byte-vector operands: eight waves per block on the three large tiles (half the accumulators per wave, so
  // two blocks bring four waves to every SIMD instead of two; measured +2..10 % on the DCGAN layers)
  if (p.bx && !p.stats) {      // slope mask of an activation applied to the stored gradient
    if constexpr (MODE == MODE_DGRAD) {
      if (igemm_waves(bm, bn, va, vb, prec, igemm_small24(p)) != 8) return -3;
      if (bm == 128 && bn == 128) vg_launch_timed(igemm_kernel<MODE, 128, 128, 2, B_MC, 4, 4, PREC_F32, 8, 3>, grid, dim3(512), 0, st, p);
      else if (bm == 128) vg_launch_timed(igemm_kernel<MODE, 128, 64, 4, B_MC, 4, 4, PREC_F32, 8, 3>, grid, dim3(512), 0, st, p);
      else vg_launch_timed(igemm_kernel<MODE, 64, 128, 2, B_MC, 4, 4, PREC_F32, 8, 3>, grid, dim3(512), 0, st, p);
      return 0;
    }
    return -3;
  }
  if (igemm_waves(bm, bn, va, vb, prec, igemm_small24(p)) == 8 && p.stats) {
    if constexpr (MODE != MODE_WGRAD) {
      if (p.bx) {
        if (bm == 128 && bn == 128) vg_launch_timed(igemm_kernel<MODE, 128, 128, 2, B_MC, 4, 4, PREC_F32, 8, 2>, grid, dim3(512), 0, st, p);
        else if (bm == 128) vg_launch_timed(igemm_kernel<MODE, 128, 64, 4, B_MC, 4, 4, PREC_F32, 8, 2>, grid, dim3(512), 0, st, p);
        else vg_launch_timed(igemm_kernel<MODE, 64, 128, 2, B_MC, 4, 4, PREC_F32, 8, 2>, grid, dim3(512), 0, st, p);
      } else {
        if (bm == 128 && bn == 128) vg_launch_timed(igemm_kernel<MODE, 128, 128, 2, B_MC, 4, 4, PREC_F32, 8, 1>, grid, dim3(512), 0, st, p);
        else if (bm == 128) vg_launch_timed(igemm_kernel<MODE, 128, 64, 4, B_MC, 4, 4, PREC_F32, 8, 1>, grid, dim3(512), 0, st, p);
        else vg_launch_timed(igemm_kernel<MODE, 64, 128, 2, B_MC, 4, 4, PREC_F32, 8, 1>, grid, dim3(512), 0, st, p);
      }
      return 0;
    }
  }
  if (p.stats) return -3;    // column statistics exist for the eight-wave f32 kernels only: the caller runs without them
  if (igemm_waves(bm, bn, va, vb, prec, igemm_small24(p)) == 8) {
    if (bm == 128 && bn == 128) vg_launch_timed(igemm_kernel<MODE, 128, 128, 2, B_MC, 4, 4, PREC_F32, 8>, grid, dim3(512), 0, st, p);
    else if (bm == 128) vg_launch_timed(igemm_kernel<MODE, 128, 64, 4, B_MC, 4, 4, PREC_F32, 8>, grid, dim3(512), 0, st, p);
    else vg_launch_timed(igemm_kernel<MODE, 64, 128, 2, B_MC, 4, 4, PREC_F32, 8>, grid, dim3(512), 0, st, p);
    return 0;
  }
  if (prec == PREC_BF16 && va == 4 && vb == 4 && bm == 128 && bn == 128) {   // eight waves here too (the other tiles
    // would leave a thread with a single row of a packed [k][row] pair)
    vg_launch_timed(igemm_kernel<MODE, 128, 128, 2, B_MC, 4, 4, PREC_BF16, 8>, grid, dim3(512), 0, st, p);
    return 0;
  }
  if (prec == PREC_BF16 && va == 4 && vb == 4) {
    IGEMM_CASE_P(128, 64, 2, 4, 4, PREC_BF16)
    IGEMM_CASE_P(64, 128, 2, 4, 4, PREC_BF16)
    IGEMM_CASE_P(64, 64, 2, 4, 4, PREC_BF16)
  }
#define IGEMM_CASE(BM_, BN_, WM_, VA_, VB_) IGEMM_CASE_P(BM_, BN_, WM_, VA_, VB_, PREC_F32)
#define IGEMM_TILE(BM_, BN_, WM_) \
  IGEMM_CASE(BM_, BN_, WM_, 4, 4) IGEMM_CASE(BM_, BN_, WM_, 4, 1) IGEMM_CASE(BM_, BN_, WM_, 1, 4) IGEMM_CASE(BM_, BN_, WM_, 1, 1)
  IGEMM_TILE(128, 128, 2)
  IGEMM_TILE(128, 64, 2)
  IGEMM_TILE(128, 32, 4)
  IGEMM_TILE(64, 128, 2)
  IGEMM_TILE(64, 64, 2)
#undef IGEMM_TILE
#undef IGEMM_CASE
#undef IGEMM_CASE_P
  return -1;
}

// WAVES_M template argument of the variant (for naming the instance a launch ran)
static inline int igemm_waves_m(int bm, int bn, int nw = 4) {
  if (nw == 8) return (bm == 128 && bn == 64) ? 4 : 2;
  return (bm == 128 && bn == 32) ? 4 : 2;
}

// Rows per tile: 128 unless that leaves the 256 CUs with at most one block each and a 64-row tile exists for this BN
// (measured on the DCGAN layers: 256 blocks of 128 rows reach 99-110 TFLOP/s, 512 blocks of 64 rows 102-117). The
// bf16 kernels were measured with smaller tiles / more blocks as well (64x64 up to 2000 blocks): no gain, same rule.
static inline int igemm_pick_bm(long long rows, int ntiles, int zdim, int bn, int prec = PREC_F32) {
  if (bn == 32 || rows <= 64) return rows <= 64 && bn != 32 ? 64 : 128;
  // the split mode is bound by operand traffic into the CUs, which the larger tile halves per flop
  if (prec == PREC_SPLIT) return ((rows + 127) / 128) * ntiles * zdim < 200 ? 64 : 128;
  const long long blocks128 = ((rows + 127) / 128) * ntiles * zdim;
  return blocks128 < 300 ? 64 : 128;
}

static inline int igemm_pick_bn(int ncols, int prec = PREC_F32) {
  (void)prec;
  return ncols > 64 ? 128 : (ncols > 32 ? 64 : 32);
}

// shared by the conv translation units
int vg_check_geom(const vg_conv_geom* g, const char* who);
