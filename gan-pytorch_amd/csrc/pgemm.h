// Implicit-GEMM convolution core on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16) over operands that were split
// into bf16 "planes" beforehand.
//
// A tensor a (f32) is represented by NP bf16 tensors of the same shape: a0 = bf16(a), a1 = bf16(a - a0),
// a2 = bf16(a - a0 - a1) (round to nearest even; a0 + a1 + a2 == a exactly).  split_planes_kernel (conv_planes.hip)
// writes them once per operand, one after the other in a workspace; this kernel then contracts
//     sum over k of  sum_{i + j < NP}  A_i[m][k] * B_j[k][n]
// in f32 accumulators.  NP = 3 gives f32-grade products (the dropped terms are below 2^-25 |a b|), NP = 1 is plain
// bf16.  Taking the split out of the contraction matters because the implicit GEMM reads every activation value
// kh*kw/(sh*sw) * (number of column tiles) times: converted in the tile loader, the conversions cost more issue
// slots than the MFMAs they feed.
//
// The three contractions and their row/column/k meanings are those of igemm.h.  Tiles are BM x BN x BK (BK = 32 with
// three planes, 64 with one), staged
// global -> registers -> LDS in 16-byte vectors of 8 bf16 (so every channel count involved must be a multiple
// of 8), double buffered, one barrier per k-step.  A tile whose global image is contiguous along k is kept
// [row][k] with 2 * BK + 16 byte rows (conflict-free ds_read_b128 of a lane's 8 k values); one that is contiguous along
// the row index is kept [k][row] as loaded and read back with ds_read_b64_tr_b16, which hands lane i of a 16-lane
// group column i of a 4 x 16 block, i.e. 4 consecutive k of one row - the MFMA operand order - without any register
// shuffling.  Its row stride is (rows / 2 + 16) dwords, which puts the four k rows of a block 16 banks apart.
#pragma once
#include "igemm.h"

struct PgemmParams {
  IgemmParams g;                 // geometry, extents, output; g.A / g.B unused
  const __bf16* Ap;              // plane 0 of operand A (fwd: x, dgrad: gy, wgrad: gy)
  const __bf16* Bp;              // plane 0 of operand B (fwd: w, dgrad: w, wgrad: x)
  long long a_plane, b_plane;    // elements from one plane to the next
  unsigned a_bytes, b_bytes;     // bytes from plane 0 to the end of the last plane (below 2 GiB: pgemm_prepare)
  // MASKED kernels (input gradient on bf16 tensors): the result is the gradient arriving at the OUTPUT of a ReLU /
  // LeakyReLU whose output is mask_ref (laid out like the result); it is multiplied by the activation's slope at each
  // position before it is stored, so what leaves is the gradient in front of the activation.
  const __bf16* mask_ref;
  int mask_act;
  float mask_slope;
  unsigned out_bytes;            // bf16 results: size of the result tensor in bytes (below 2 GiB: pgemm_set_out_h)
  // Weight gradient over several (gy, x) pairs of one geometry in one launch (the passes of a training step that share
  // a layer): split z belongs to pair z / seg_splits and is split z % seg_splits of it; every split writes its own
  // slab as before, so the slab sum is the sum over all pairs. seg_splits == 0: one pair, Ap / Bp.
  const __bf16* Aseg[4];
  const __bf16* Bseg[4];
  int seg_splits;
};

static __device__ __attribute__((aligned(16))) unsigned int vg_zero_page_u[4] = {0u, 0u, 0u, 0u};  // never written

// Diagnostic builds only (tools/micro/pgemm_ablate.py compiles the library with -DPG_ABLATE=<bits> to see where the time
// of the loop goes; results are then wrong by construction): 1 no global loads, 2 no LDS stores, 4 no fragment reads,
// 8 no MFMAs, 16 no result stores.
#ifndef PG_ABLATE
#define PG_ABLATE 0
#endif

typedef short s16x4 __attribute__((ext_vector_type(4)));

// OUT_H: the result leaves as bf16 (round to nearest even) instead of f32 - the convolution of a network whose
// activations are stored as bf16. Two neighbouring lanes (columns c, c + 1) exchange one accumulator per pair of rows,
// so that every lane stores a packed pair of columns: 4-byte stores, 16 lanes per 64-byte row segment.
// Single-plane forward / input-gradient instances are held to 128 registers (four waves per SIMD, i.e. two 8-wave blocks
// per CU: the 128 x 128 tile otherwise takes 136-146 and runs one block per CU; measured 25.7 vs 28.0 us and 28.4 vs
// 31.9 us). Not the weight gradient: its grid is one block per CU anyway and the few spills cost it 15%. Nor the input
// gradient with the general k-walk, whose per-unit tap state no longer fits: under the bound it spills a staged
// tile inside the loop (77 against 74 us on a 128 -> 160 channel layer).
// KU ("k-steps uniform", forward and input gradient): every k-step lies inside one filter tap (the channel count along k
// is a multiple of BK) and a tap index fits a 32-bit mask (pgemm_uniform_k vouches for both). The walk over taps and
// channels is then block-uniform and lives in scalar registers; a thread keeps, per tile row, a byte offset and a mask
// of the taps that read inside the tensor, per weight unit a byte offset, and a load's address is one addition. Vector
// instructions do not overlap the matrix instructions of a SIMD, so the fifty-odd of the general walk cost a k-step as
// much as its MFMAs.
// KU, weight gradient (k = gradient pixels): a k-step is whole rows of the gradient image (BK a multiple of Wo), its
// rows lie in one image or are whole images, and every chunk is whole k-steps. Image and first row of a k-step are
// then block-uniform; what a thread adds (its row inside the step, its column, its tap) never changes.
template <int MODE, int BM, int BN, int WAVES_M, int NP, int NW, int BK_ = 32, bool OUT_H = false, bool MASKED = false,
          bool KU = false>
__global__ void __launch_bounds__(NW * 64, (NP == 1 && NW == 8 && MODE != MODE_WGRAD && (KU || MODE == MODE_FWD)) ? 4 : 1) pgemm_kernel(const PgemmParams pp) {
  static_assert(!MASKED || (OUT_H && MODE == MODE_DGRAD), "the slope mask belongs to the input gradient on bf16 tensors");
  const IgemmParams& p = pp.g;
  constexpr int NT = NW * 64;
  constexpr int BK = BK_;                          // k extent of a staged tile: 32, or 64 where LDS allows (one plane)
  constexpr int KV = BK / 8;                       // 16-byte vectors per row of a [row][k] tile
  constexpr int NS = BK / 16;                      // 16-deep MFMA slices per k-step
  constexpr int LDB = BK + 8;                      // bf16 per row of a [row][k] tile (80- or 144-byte rows: conflict-free)
  static_assert(BK == 32 || BK == 64, "two or four slices per k-step");
  constexpr int LDMA = BM + 32, LDMB = BN + 32;    // bf16 per k row of a [k][row] tile
  constexpr int WAVES_N = NW / WAVES_M;
  constexpr int TM = BM / (WAVES_M * 32), TN = BN / (WAVES_N * 32);
  constexpr bool A_MC = (MODE == MODE_WGRAD);
  constexpr bool B_MC = (MODE != MODE_FWD);
  static_assert(TM >= 1 && TN >= 1, "tile too small for the wave arrangement");

  constexpr int A_EL = A_MC ? BK * LDMA : BM * LDB;   // bf16 elements per plane
  constexpr int B_EL = B_MC ? BK * LDMB : BN * LDB;
  constexpr int A_ALL = NP * A_EL, BUF = NP * (A_EL + B_EL);
  __shared__ __attribute__((aligned(16))) __bf16 smem[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wid / WAVES_N, wn = wid % WAVES_N;

  // XCD-aware block order. Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest, then z), and
  // each XCD has its own L2. Every XCD is given a contiguous run of the logical order "z-major, then tile": for the
  // weight gradient that is whole split-K chunks, whose tiles all read the same pixels (2 + 16 operand tiles feed 32
  // blocks); for the other modes a run of row tiles with all their column tiles.
  int mt, nt, zz;
  {
    const int tiles = gridDim.x, nblk = gridDim.x * gridDim.z;
    const int lin = blockIdx.x + gridDim.x * blockIdx.z;
    const int xcd = lin & 7, q = nblk >> 3, rem = nblk & 7;
    const int id = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (lin >> 3);
    int t;
    if constexpr (MODE == MODE_DGRAD) {   // the parity classes of one row tile read the same gradient pixels: keep them together
      t = id / (int)gridDim.z;
      zz = id - t * (int)gridDim.z;
    } else {
      zz = id / tiles;
      t = id - zz * tiles;
    }
    mt = t / p.ntiles;
    nt = t - mt * p.ntiles;
  }
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-mode extents (as igemm.h) ----
  int M = p.M, Ktot = p.Ktot;
  int kbeg = 0, kend = p.Ktot;
  int cls_h = 0, cls_w = 0, off_h = 0, off_w = 0, Hc = 0, Wc = 0, nth = 0, ntw = 0, base_h = 0, base_w = 0;
  const __bf16 *opA = pp.Ap, *opB = pp.Bp;       // plane 0 of this block's operands
  float tab_inv_roww = 0.f, tab_inv_plane = 0.f, tab_inv_ntw = 0.f;
  const bool tab = MODE == MODE_DGRAD && zz < p.ncls;     // the class's geometry as the host worked it out
  if constexpr (MODE == MODE_DGRAD) {
   if (tab) {
    const IgemmParams::Class& c = p.cls[zz];
    cls_h = c.cls_h; cls_w = c.cls_w; off_h = c.off_h; off_w = c.off_w; Hc = c.Hc; Wc = c.Wc; nth = c.nth; ntw = c.ntw;
    base_h = c.base_h; base_w = c.base_w;
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
    if (m0 >= M) return;  // block-uniform
  } else if constexpr (MODE == MODE_WGRAD) {
    int zl = zz;
    if (pp.seg_splits > 0) {
      const int seg = zz / pp.seg_splits;
      zl = zz - seg * pp.seg_splits;
      // (constant indices: a dynamically indexed array of a by-value kernel argument sends the whole structure to
      // scratch memory)
      opA = seg == 0 ? pp.Aseg[0] : seg == 1 ? pp.Aseg[1] : seg == 2 ? pp.Aseg[2] : pp.Aseg[3];
      opB = seg == 0 ? pp.Bseg[0] : seg == 1 ? pp.Bseg[1] : seg == 2 ? pp.Bseg[2] : pp.Bseg[3];
    }
    kbeg = zl * p.kchunk;
    kend = min(kbeg + p.kchunk, p.Ktot);
  } else {
    ntw = p.KW;
  }
  const int nk = (kend - kbeg + BK - 1) / BK;

  // ---- loader geometry: a thread owns 16-byte units (8 bf16) of each plane ----
  // [row][k] tile: unit = (row, k8 = 8 * (tid % KV)), rows (tid / KV) + (NT / KV) * i
  // [k][row] tile: unit = (k, 8 rows), TPK threads per k row, k rows tid / TPK + (NT / TPK) * i
  constexpr int UA_RK = (BM * KV + NT - 1) / NT, UB_RK = (BN * KV + NT - 1) / NT;
  constexpr int TPKA = BM / 8, TPKB = BN / 8;
  constexpr int UA_MC = (BK * TPKA + NT - 1) / NT, UB_MC = (BK * TPKB + NT - 1) / NT;
  constexpr int UA = A_MC ? UA_MC : UA_RK, UB = B_MC ? UB_MC : UB_RK;
  auto rowA = [&](int i) -> int { return (tid / KV) + (NT / KV) * i; };    // [row][k] tiles
  auto rowB = [&](int i) -> int { return (tid / KV) + (NT / KV) * i; };
  auto krowA = [&](int i) -> int { return tid / TPKA + (NT / TPKA) * i; };  // [k][row] tiles
  auto krowB = [&](int i) -> int { return tid / TPKB + (NT / TPKB) * i; };

  // Products of two non-negative factors below 2^24 (vg_planes_applicable vouches for the extents): v_mul_u32_u24 runs at
  // full rate, the 32-bit v_mul_lo_u32 at a quarter of it, and the k-walk below is made of such products.
  auto m24 = [](int a, int b) -> int { return (int)__umul24((unsigned)a, (unsigned)b); };

  // ---- A loader state ([row][k]: fwd / dgrad): the row's element offset at tap (0, 0), channel 0 (taps and channels
  // are added per k-step with one product shared by all rows of the thread), and its window origin for the bounds ----
  unsigned a_base[A_MC ? 1 : UA];
  int a_h[A_MC ? 1 : UA], a_w[A_MC ? 1 : UA];
  const bool fast = p.fast_div != 0;
  const int roww = (MODE == MODE_FWD) ? p.Wo : Wc;
  const int plane = (MODE == MODE_FWD) ? p.Ho * p.Wo : Hc * Wc;
  const float inv_roww = tab ? tab_inv_roww : roww > 0 ? 1.0f / (float)roww : 0.f;
  const float inv_plane = tab ? tab_inv_plane : plane > 0 ? 1.0f / (float)plane : 0.f;
  unsigned a_mask[(KU && !A_MC) ? UA : 1];      // KU: bit t set = tap t of this row reads inside the tensor
  const int k8b = 16 * (tid % KV);              // KU: byte offset of the thread's 8 k inside a k-step of a [row][k] tile
  if constexpr (!A_MC) {
#pragma unroll
    for (int i = 0; i < UA; ++i) {
      const int m = m0 + rowA(i);
      a_base[i] = 0u;
      a_w[i] = 0;
      a_h[i] = -(1 << 28);
      if (rowA(i) < BM && m < M) {
        const int n = vg_div_row(m, plane, inv_plane, fast), rem = m - n * plane;
        const int a = vg_div_row(rem, roww, inv_roww, fast), b = rem - a * roww;
        if constexpr (MODE == MODE_FWD) {
          a_h[i] = a * p.sh - p.ph;
          a_w[i] = b * p.sw - p.pw;
          a_base[i] = (unsigned)((n * p.H + a_h[i]) * p.W + a_w[i]) * (unsigned)p.Ci;      // (wraps for a padded origin)
        } else {
          a_h[i] = a + base_h;
          a_w[i] = b + base_w;
          a_base[i] = (unsigned)((n * p.Ho + a_h[i]) * p.Wo + a_w[i]) * (unsigned)p.Co;
        }
      }
      if constexpr (KU) {
        // taps (ta, tb), numbered ta * ntw + tb as the k-walk numbers them: which of them this row may read
        const int tw = (MODE == MODE_FWD) ? p.KW : ntw, th = (MODE == MODE_FWD) ? p.KH : nth;
        const unsigned lim_h = (unsigned)((MODE == MODE_FWD) ? p.H : p.Ho), lim_w = (unsigned)((MODE == MODE_FWD) ? p.W : p.Wo);
        unsigned mw = 0u, mk = 0u;
        for (int tb = 0; tb < tw; ++tb) mw |= ((unsigned)((MODE == MODE_FWD) ? a_w[i] + tb : a_w[i] - tb) < lim_w ? 1u : 0u) << tb;
        for (int ta = 0; ta < th; ++ta)
          mk |= ((unsigned)((MODE == MODE_FWD) ? a_h[i] + ta : a_h[i] - ta) < lim_h ? mw : 0u) << (ta * tw);
        a_mask[i] = mk;                        // (a row past the end has a_h far below zero: no tap)
        a_base[i] = a_base[i] * 2u + (unsigned)k8b;
      }
    }
  }

  // ---- B loader state ----
  // wgrad: the 8 tile columns (kh, kw, ci .. ci+7) of this thread never change along k
  const int bn8 = n0 + 8 * (tid % TPKB);
  int b_ta = 0, b_tb = 0, b_ci = 0;
  if constexpr (MODE == MODE_WGRAD) {
    const int tap = bn8 / p.Ci;
    b_ci = bn8 - tap * p.Ci;
    b_ta = tap / p.KW;
    b_tb = tap - b_ta * p.KW;
    if (bn8 >= p.Ncols) b_ta = -(1 << 28);
  }

  // ---- k-walk state (stepped by 32 per tile, no divisions in the loop) ----
  int kpos = kbeg;
  int ak_c = 0, ak_t = 0;
  const int a_cch = (MODE == MODE_FWD) ? p.Ci : p.Co;
  const int a_steps = BK / a_cch, a_rem = BK - a_steps * a_cch;
  const float inv_ntw = tab ? tab_inv_ntw : ntw > 0 ? 1.0f / (float)ntw : 0.f;
  if constexpr (!A_MC && !KU) {
    const int k8 = kbeg + 8 * (tid % KV);
    ak_t = k8 / a_cch;
    ak_c = k8 - ak_t * a_cch;
  }
  int bk_co[(MODE == MODE_DGRAD) ? UB : 1], bk_t[(MODE == MODE_DGRAD) ? UB : 1];
  if constexpr (MODE == MODE_DGRAD && !KU) {
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int k = kbeg + krowB(i);
      bk_t[i] = k / p.Co;
      bk_co[i] = k - bk_t[i] * p.Co;
    }
  }
  // wgrad A = gy[pixel][co]: the unit's element offset, advanced by BK pixels per tile
  unsigned a_off[A_MC ? UA : 1];
  if constexpr (A_MC) {
#pragma unroll
    for (int i = 0; i < UA; ++i) a_off[i] = (unsigned)(kbeg + krowA(i)) * (unsigned)p.Co + (unsigned)(m0 + 8 * (tid % TPKA));
  }
  // fwd B = w[co][k]: the row's first element
  unsigned b_rowbase[B_MC ? 1 : UB];
  if constexpr (!B_MC) {
#pragma unroll
    for (int i = 0; i < UB; ++i) b_rowbase[i] = (unsigned)(n0 + rowB(i)) * (unsigned)Ktot;
  }
  const int khkw = p.KH * p.KW;
  constexpr unsigned OOB = 0x80000000u;         // a byte offset no operand reaches (pgemm_prepare): the load returns zeros
  // KU: the unit's byte offset at k-step 0 (tap 0, channel 0), OOB for a unit that is never read (adding the uniform
  // part, below 2^31, leaves it out of range)
  unsigned b_unit[KU ? UB : 1];
  int s_c0 = 0, s_ta = 0, s_tb = 0, s_t = 0;    // KU: channel, tap row, tap column, tap number of the next k-step to load
  if constexpr (KU && MODE == MODE_FWD) {
#pragma unroll
    for (int i = 0; i < UB; ++i)
      b_unit[i] = (rowB(i) < BN && n0 + rowB(i) < p.Ncols) ? b_rowbase[i] * 2u + (unsigned)k8b : OOB;
  }
  if constexpr (KU && MODE == MODE_DGRAD) {
#pragma unroll
    for (int i = 0; i < UB; ++i)
      b_unit[i] = (krowB(i) < BK && bn8 < p.Ncols) ? (unsigned)(krowB(i) * khkw * p.Ci + bn8) * 2u : OOB;
  }
  int bp_row[(MODE == MODE_WGRAD) ? UB : 1], bp_wo[(MODE == MODE_WGRAD) ? UB : 1];
  const int w_steps = (MODE == MODE_WGRAD) ? BK / p.Wo : 0, w_rem = (MODE == MODE_WGRAD) ? BK - w_steps * p.Wo : 0;
  const float inv_ho = 1.0f / (float)p.Ho;
  // KU weight gradient. Uniform: image s_n and row s_ho of the k-step's first pixel. Per B unit: wg_hic = what the
  // unit adds to s_ho * sh to give its window row, wg_c = its byte offset relative to the window row s_ho * sh of
  // image s_n (OOB where its column or tap column falls outside). Per A unit: its byte offset relative to the step's
  // first pixel.
  int wg_hic[(KU && MODE == MODE_WGRAD) ? UB : 1];
  unsigned wg_c[(KU && MODE == MODE_WGRAD) ? UB : 1], wg_a[(KU && MODE == MODE_WGRAD) ? UA : 1];
  int s_n = 0, s_ho = 0;
  if constexpr (KU && MODE == MODE_WGRAD) {
    const int row0 = kbeg / p.Wo;               // kbeg is a multiple of BK, BK of Wo
    s_n = row0 / p.Ho;
    s_ho = row0 - s_n * p.Ho;
    const int wci = p.W * p.Ci;
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int j = krowB(i), dr = j / p.Wo, wo = j - dr * p.Wo;
      const int dn = dr / p.Ho, dh = dr - dn * p.Ho;          // (dn = 0 unless the step spans whole images)
      const int wi = wo * p.sw - p.pw + b_tb;
      wg_hic[i] = dh * p.sh - p.ph + b_ta;
      const bool ok = j < BK && (unsigned)wi < (unsigned)p.W;  // (b_ta is far below zero for a column past the end)
      wg_c[i] = ok ? (unsigned)(((dn * p.H + wg_hic[i]) * p.W + wi) * p.Ci + b_ci) * 2u : OOB;
      if (!ok) wg_hic[i] = -(1 << 28);
    }
    const int m8 = m0 + 8 * (tid % TPKA);
#pragma unroll
    for (int i = 0; i < UA; ++i) wg_a[i] = (krowA(i) < BK && m8 < p.Co) ? (unsigned)(krowA(i) * p.Co + m8) * 2u : OOB;
  }
  if constexpr (MODE == MODE_WGRAD && !KU) {
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int pix = kbeg + krowB(i);
      bp_row[i] = pix / p.Wo;
      bp_wo[i] = pix - bp_row[i] * p.Wo;
    }
  }

  // two register stages: tile t travels through stage t % 2, so its loads are in flight for two k-steps
  u32x4 ra[2][UA][NP], rb[2][UB][NP];

  // Predicated 16-byte load of the same element offset in every plane, as a range-checked buffer load on a 32-bit
  // byte offset: a unit that must not be read gets an offset beyond the buffer and comes back as zeros (no 64-bit
  // address arithmetic and no pointer select in the k-walk; with bf16 MFMAs a k-step's address arithmetic on the vector
  // ALU takes as long as its matrix work).
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(opA), 0, (int)pp.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcB = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(opB), 0, (int)pp.b_bytes, 0x00020000);
  const unsigned a_pbytes = (unsigned)pp.a_plane * 2u, b_pbytes = (unsigned)pp.b_plane * 2u;
  auto ld_planes = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned pbytes, unsigned off, bool ok, u32x4 (&dst)[NP]) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      if constexpr ((PG_ABLATE & 1) != 0) {
        dst[q] = u32x4{off, 0u, 0u, 0u};
      } else {
        dst[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ok ? off * 2u + q * pbytes : OOB), 0, 0));
      }
    }
  };

  auto ld_bytes = [&](const __amdgpu_buffer_rsrc_t& rs, unsigned pbytes, unsigned boff, u32x4 (&dst)[NP]) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      if constexpr ((PG_ABLATE & 1) != 0) {
        dst[q] = u32x4{boff, 0u, 0u, 0u};
      } else {
        dst[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(boff + q * pbytes), 0, 0));
      }
    }
  };

  auto load_tiles = [&](int st) {
    if constexpr (KU && MODE == MODE_WGRAD) {
      const int hrow = s_ho * p.sh;
      const unsigned ub = (unsigned)((s_n * p.H + hrow) * p.W * p.Ci) * 2u, ua = (unsigned)kpos * (unsigned)p.Co * 2u;
#pragma unroll
      for (int i = 0; i < UA; ++i) ld_bytes(rsrcA, a_pbytes, wg_a[i] + ua, ra[st][i]);
#pragma unroll
      for (int i = 0; i < UB; ++i)
        ld_bytes(rsrcB, b_pbytes, (unsigned)(hrow + wg_hic[i]) < (unsigned)p.H ? wg_c[i] + ub : OOB, rb[st][i]);
      kpos += BK;
      s_ho += w_steps;
      if (s_ho >= p.Ho) {                       // the next step starts w_steps rows on: same image, or whole images on
        s_n += w_steps > p.Ho ? w_steps / p.Ho : 1;
        s_ho = 0;
      }
      return;
    }
    if constexpr (KU) {
      // block-uniform part of the k-step: scalar registers
      const unsigned bit = 1u << s_t;
      unsigned ua, ub;
      if constexpr (MODE == MODE_FWD) {
        ua = (unsigned)((s_ta * p.W + s_tb) * p.Ci + s_c0) * 2u;
        ub = (unsigned)kpos * 2u;
      } else {
        ua = (unsigned)(s_c0 - (s_ta * p.Wo + s_tb) * p.Co) * 2u;
        ub = (unsigned)((s_c0 * khkw + (cls_h + p.sh * s_ta) * p.KW + cls_w + p.sw * s_tb) * p.Ci) * 2u;
      }
#pragma unroll
      for (int i = 0; i < UA; ++i) ld_bytes(rsrcA, a_pbytes, (a_mask[i] & bit) ? a_base[i] + ua : OOB, ra[st][i]);
#pragma unroll
      for (int i = 0; i < UB; ++i) ld_bytes(rsrcB, b_pbytes, b_unit[i] + ub, rb[st][i]);
      kpos += BK;
      s_c0 += BK;
      if (s_c0 >= a_cch) {
        s_c0 = 0;
        ++s_t;
        ++s_tb;
        const int tw = (MODE == MODE_FWD) ? p.KW : ntw;
        if (s_tb >= tw) {
          s_tb = 0;
          ++s_ta;
        }
      }
      return;
    }
    const int kb = kpos;
    // ---------------- A ----------------
    if constexpr (!A_MC) {
      const bool kvalid = kb + 8 * (tid % KV) < kend;
      const int ak_ta = vg_div_small(ak_t, inv_ntw), ak_tb = ak_t - m24(ak_ta, ntw);
      // what tap (ak_ta, ak_tb) and channel ak_c add to a row's offset: one product pair per k-step, not per row
      const unsigned tap_off = (MODE == MODE_FWD)
                                   ? (unsigned)m24(m24(ak_ta, p.W) + ak_tb, p.Ci) + (unsigned)ak_c
                                   : (unsigned)ak_c - (unsigned)m24(m24(ak_ta, p.Wo) + ak_tb, p.Co);
#pragma unroll
      for (int i = 0; i < UA; ++i) {
        if constexpr (MODE == MODE_FWD) {
          const int hi = a_h[i] + ak_ta, wi = a_w[i] + ak_tb;
          const bool ok = kvalid && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
          ld_planes(rsrcA, a_pbytes, a_base[i] + tap_off, ok, ra[st][i]);
        } else {
          const int ho = a_h[i] - ak_ta, wo = a_w[i] - ak_tb;
          const bool ok = kvalid && (unsigned)ho < (unsigned)p.Ho && (unsigned)wo < (unsigned)p.Wo;
          ld_planes(rsrcA, a_pbytes, a_base[i] + tap_off, ok, ra[st][i]);
        }
      }
    } else {
      // wgrad A = gy[pixel][co], tile [k][m]
      const int m8 = m0 + 8 * (tid % TPKA);
#pragma unroll
      for (int i = 0; i < UA; ++i) {
        const int pix = kb + krowA(i);
        ld_planes(rsrcA, a_pbytes, a_off[i], krowA(i) < BK && pix < kend && m8 < p.Co, ra[st][i]);
        a_off[i] += (unsigned)BK * (unsigned)p.Co;
      }
    }
    // ---------------- B ----------------
    if constexpr (!B_MC) {
      // fwd weights w[co][k]: tile [n][k]
      const int k8 = kb + 8 * (tid % KV);
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        const int n = n0 + rowB(i);
        ld_planes(rsrcB, b_pbytes, b_rowbase[i] + (unsigned)k8, rowB(i) < BN && n < p.Ncols && k8 < kend, rb[st][i]);
      }
    } else if constexpr (MODE == MODE_DGRAD) {
      // weights w[co][kh][kw][ci] read as B[k=(tap,co)][n=ci]
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        const int k = kb + krowB(i);
        const int bta = vg_div_small(bk_t[i], inv_ntw), btb = bk_t[i] - m24(bta, ntw);
        const int kh = cls_h + m24(p.sh, bta), kw = cls_w + m24(p.sw, btb);
        const unsigned base = (unsigned)m24(m24(bk_co[i], khkw) + m24(kh, p.KW) + kw, p.Ci);
        ld_planes(rsrcB, b_pbytes, base + (unsigned)bn8, krowB(i) < BK && k < kend && bn8 < p.Ncols, rb[st][i]);
      }
    } else {
      // wgrad B = x window, tile [k=pixel][n=(kh,kw,ci)]
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        const int pix = kb + krowB(i);
        const int bpn = vg_div_small(bp_row[i], inv_ho), bpho = bp_row[i] - m24(bpn, p.Ho);
        const int hi = m24(bpho, p.sh) - p.ph + b_ta, wi = m24(bp_wo[i], p.sw) - p.pw + b_tb;
        const bool ok = krowB(i) < BK && pix < kend && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        // (hi, wi are in range wherever the unit is read, so the factors below are non-negative there)
        ld_planes(rsrcB, b_pbytes, (unsigned)(m24(m24(bpn, p.H) + hi, p.W) + wi) * (unsigned)p.Ci + (unsigned)b_ci, ok, rb[st][i]);
      }
    }
    // ---------------- advance the k-walk by one tile ----------------
    kpos += BK;
    if constexpr (!A_MC) {
      ak_c += a_rem;
      const int carry = ak_c >= a_cch ? 1 : 0;
      ak_c -= carry ? a_cch : 0;
      ak_t += a_steps + carry;
    }
    if constexpr (MODE == MODE_DGRAD) {
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        bk_co[i] += a_rem;
        const int carry = bk_co[i] >= a_cch ? 1 : 0;
        bk_co[i] -= carry ? a_cch : 0;
        bk_t[i] += a_steps + carry;
      }
    }
    if constexpr (MODE == MODE_WGRAD) {
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        bp_wo[i] += w_rem;
        const int carry = bp_wo[i] >= p.Wo ? 1 : 0;
        bp_wo[i] -= carry ? p.Wo : 0;
        bp_row[i] += w_steps + carry;
      }
    }
  };

  auto store_tiles = [&](int buf) {   // tile in stage `buf` -> LDS buffer `buf`
    if constexpr ((PG_ABLATE & 2) != 0) {
#pragma unroll
      for (int i = 0; i < UA; ++i) asm volatile("" ::"v"(ra[buf][i][0]));
#pragma unroll
      for (int i = 0; i < UB; ++i) asm volatile("" ::"v"(rb[buf][i][0]));
      return;
    }
    __bf16* As = smem + buf * BUF;
    __bf16* Bs = As + A_ALL;
#pragma unroll
    for (int i = 0; i < UA; ++i) {
      if constexpr (!A_MC) {
        if (BM * KV % NT == 0 || rowA(i) < BM) {
#pragma unroll
          for (int q = 0; q < NP; ++q)
            *reinterpret_cast<u32x4*>(&As[q * A_EL + rowA(i) * LDB + 8 * (tid % KV)]) = ra[buf][i][q];
        }
      } else {
        if (BK * TPKA % NT == 0 || krowA(i) < BK) {
#pragma unroll
          for (int q = 0; q < NP; ++q)
            *reinterpret_cast<u32x4*>(&As[q * A_EL + krowA(i) * LDMA + 8 * (tid % TPKA)]) = ra[buf][i][q];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      if constexpr (!B_MC) {
        if (BN * KV % NT == 0 || rowB(i) < BN) {
#pragma unroll
          for (int q = 0; q < NP; ++q)
            *reinterpret_cast<u32x4*>(&Bs[q * B_EL + rowB(i) * LDB + 8 * (tid % KV)]) = rb[buf][i][q];
        }
      } else {
        if (BK * TPKB % NT == 0 || krowB(i) < BK) {
#pragma unroll
          for (int q = 0; q < NP; ++q)
            *reinterpret_cast<u32x4*>(&Bs[q * B_EL + krowB(i) * LDMB + 8 * (tid % TPKB)]) = rb[buf][i][q];
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

  // Fragment of a [k][row] tile: lane (r, h) needs k = 16 s + 8 h + 0..7 of row `row0 + r`. Two transposed reads of
  // 4 k each; in a 16-lane group, lane 4 q + c supplies the address of k row q, rows 4 c .. 4 c + 3 of the group's 16.
  const int tr_q = (lane & 15) >> 2, tr_c = lane & 3, tr_half = (lane >> 4) & 1;
  auto read_tr = [&](const __bf16* T, int ldm, int row0, int s) -> bf16x8 {
    const __bf16* a0 = T + (16 * s + 8 * h + tr_q) * ldm + row0 + 16 * tr_half + 4 * tr_c;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * ldm));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  // fragment q * T + i is plane q of 32-row tile i
  auto load_frags = [&](int buf, int s, bf16x8 (&af)[NP * TM], bf16x8 (&bf)[NP * TN]) {
    if constexpr ((PG_ABLATE & 4) != 0) return;
    const __bf16* As = smem + buf * BUF;
    const __bf16* Bs = As + A_ALL;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
#pragma unroll
      for (int mi = 0; mi < TM; ++mi) {
        const int row0 = (wm * TM + mi) * 32;
        if constexpr (!A_MC)
          af[q * TM + mi] = *reinterpret_cast<const bf16x8*>(&As[q * A_EL + (row0 + r) * LDB + 16 * s + 8 * h]);
        else
          af[q * TM + mi] = read_tr(As + q * A_EL, LDMA, row0, s);
      }
#pragma unroll
      for (int ni = 0; ni < TN; ++ni) {
        const int col0 = (wn * TN + ni) * 32;
        if constexpr (!B_MC)
          bf[q * TN + ni] = *reinterpret_cast<const bf16x8*>(&Bs[q * B_EL + (col0 + r) * LDB + 16 * s + 8 * h]);
        else
          bf[q * TN + ni] = read_tr(Bs + q * B_EL, LDMB, col0, s);
      }
    }
  };
  // every plane pair (qa, qb) with qa + qb < NP, smallest terms first
  auto mma_frags = [&](const bf16x8 (&af)[NP * TM], const bf16x8 (&bf)[NP * TN]) {
    if constexpr ((PG_ABLATE & 8) != 0) {
#pragma unroll
      for (int i = 0; i < NP * TM; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
      for (int i = 0; i < NP * TN; ++i) asm volatile("" ::"v"(bf[i]));
      return;
    }
#pragma unroll
    for (int order = NP - 1; order >= 0; --order)
#pragma unroll
      for (int qa = 0; qa <= order; ++qa)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qa * TM + mi], bf[(order - qa) * TN + ni],
                                                                  acc[mi][ni], 0, 0, 0);
  };

  bf16x8 afx[NP * TM], bfx[NP * TN], afy[NP * TM], bfy[NP * TN];
  if constexpr ((PG_ABLATE & 4) != 0) {
#pragma unroll
    for (int i = 0; i < NP * TM; ++i) afx[i] = afy[i] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)tid, 1u, 2u, 3u});
#pragma unroll
    for (int i = 0; i < NP * TN; ++i) bfx[i] = bfy[i] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)tid, 1u, 2u, 3u});
  }

  // One k-step on buffer `buf` (NS 16-deep slices). On entry afx/bfx hold slice 0 and the other register stage holds
  // tile kt+1 (loads issued two k-steps ago); it is written to the other buffer under the MFMAs, the loads of tile
  // kt+3 are issued into the freed stage, and after the barrier slice 0 of the next buffer is fetched under the MFMAs
  // of slice 1.
  // `more` = tiles kt+1 and kt+3 are known to exist (the branch-free form the steady-state loop uses: with
  // conditional loads in the loop body the compiler can no longer count which loads are outstanding at the LDS store
  // and waits for all of them, the stage loaded one k-step ago included).
  auto k_step = [&](int kt, int buf, bool more) {
#pragma unroll
    for (int sl = 0; sl + 1 < NS; ++sl) {      // slices 0 .. NS-2: even ones sit in afx/bfx, odd ones in afy/bfy
      if (sl & 1) load_frags(buf, sl + 1, afx, bfx);
      else load_frags(buf, sl + 1, afy, bfy);
      __builtin_amdgcn_sched_barrier(0);
      if (sl & 1) mma_frags(afy, bfy);
      else mma_frags(afx, bfx);
      if (sl == 0) {
        if (more || kt + 1 < nk) store_tiles(buf ^ 1);
        if (more || kt + 3 < nk) load_tiles(buf ^ 1);
      }
    }
    __syncthreads();
    if (more || kt + 1 < nk) load_frags(buf ^ 1, 0, afx, bfx);
    __builtin_amdgcn_sched_barrier(0);
    mma_frags(afy, bfy);                       // the last slice (odd)
  };

  if (nk > 0) {
    load_tiles(0);
    store_tiles(0);
    if (nk > 1) load_tiles(1);
    if (nk > 2) load_tiles(0);
  }
  __syncthreads();
  if (nk > 0) load_frags(0, 0, afx, bfx);

  int kt = 0;
  for (; kt + 4 < nk; kt += 2) {     // steady state: every tile touched exists
    k_step(kt, 0, true);
    k_step(kt + 1, 1, true);
  }
  for (; kt + 1 < nk; kt += 2) {
    k_step(kt, 0, false);
    k_step(kt + 1, 1, false);
  }
  if (kt < nk) k_step(kt, 0, false);

  // ---- epilogue: lane holds column (lane&31) and rows (e&3)+8*(e>>2)+4*(lane>>5) of each 32x32 tile ----
  if constexpr ((PG_ABLATE & 16) != 0) {
    float keep = 0.f;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
      for (int ni = 0; ni < TN; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) keep += acc[mi][ni][e];
    if (keep == 123.456f) p.out[0] = keep;      // (practically never: keeps the accumulators alive)
    return;
  }
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
  if constexpr (OUT_H) {
    static_assert(MODE != MODE_WGRAD, "weight gradients leave as f32 slabs");
    // A lane holds column c = (lane & 31) of 16 rows; rows 2 e2 and 2 e2 + 1 of its accumulators are neighbours, and so
    // are the columns of lanes l and l ^ 1. The even lane of a pair stores the first row's two columns as one word, the
    // odd lane the second row's. Results leave through range-checked buffer stores on 32-bit byte offsets: a row past
    // the end has an offset beyond the tensor (below 2 GiB: checked by the host) and its store is dropped, so the
    // store loop carries no conditions; a lane whose columns lie past the last one sits the whole loop out.
    const bool odd = (lane & 1) != 0;
    const __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)pp.out_bytes, 0x00020000);
    const unsigned colb = 2u * (unsigned)(n0 + wn * TN * 32 + (r & ~1));
    unsigned roffb[TM][8];                       // byte offsets of the rows this lane stores
    unsigned yy[MASKED ? TM : 1][MASKED ? 8 : 1][MASKED ? TN : 1];
    if (has_bias) {
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[mi][ni][e] += bcol[ni];
    }
    // Input gradient: a tile's rows are pixels of one parity class, scattered over the image. Each row's offset takes
    // two divisions; it is worked out once per block and left in LDS (the k loop is done with it), not sixteen times
    // per lane. Within a group of four rows the table is kept in the order 0, 2, 1, 3, so that a lane's two rows of a
    // group (0 and 2 for an even lane, 1 and 3 for an odd one) are one 8-byte read.
    unsigned* row_off = reinterpret_cast<unsigned*>(smem);
    if constexpr (MODE == MODE_DGRAD) {
      static_assert(NT >= BM, "one table entry per thread");
      __syncthreads();                          // every wave is done with the staged tiles
      if (tid < BM) {
        const int row = m0 + tid;
        unsigned roff = OOB;
        if (row < M) {
          const int n = vg_div_row(row, plane, inv_plane, fast), rem = row - n * plane;
          const int ii = vg_div_row(rem, roww, inv_roww, fast), jj = rem - ii * roww;
          roff = (unsigned)((n * p.H + ii * p.sh + off_h) * p.W + jj * p.sw + off_w) * (unsigned)p.Ncols * 2u;
        }
        row_off[(tid & ~3) | ((tid & 1) << 1) | ((tid >> 1) & 1)] = roff;
      }
      __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
      if constexpr (MODE == MODE_DGRAD) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const uint2 t = *reinterpret_cast<const uint2*>(&row_off[(wm * TM + mi) * 32 + 4 * h + 8 * q + (odd ? 2 : 0)]);
          roffb[mi][2 * q] = t.x;
          roffb[mi][2 * q + 1] = t.y;
        }
      } else {
        const int row_t = m0 + (wm * TM + mi) * 32 + 4 * h + (odd ? 1 : 0);
        const unsigned rowb = 2u * (unsigned)p.Ncols;
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2) {
          const int row = row_t + ((2 * e2) & 3) + 8 * ((2 * e2) >> 2);
          roffb[mi][e2] = row < M ? (unsigned)row * rowb : OOB;
        }
      }
    }
    if constexpr (MASKED) {
      // the words of the activation's output behind the words this lane stores, all requested before the first is used
      const __amdgpu_buffer_rsrc_t rsrcM =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(pp.mask_ref), 0, (int)pp.out_bytes, 0x00020000);
#pragma unroll
      for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int e2 = 0; e2 < 8; ++e2)
#pragma unroll
          for (int ni = 0; ni < TN; ++ni)
            yy[mi][e2][ni] = cok[ni] ? (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrcM, (int)(roffb[mi][e2] + colb + ni * 64), 0, 0) : 0u;
    }
    const float low = pp.mask_act == VG_ACT_RELU ? 0.f : pp.mask_slope;
    const unsigned sel = odd ? 0x03020706u : 0x05040100u;   // v_perm_b32(neighbour, own): (neighbour, own) high halves / (own, neighbour) low halves
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
      if (cok[ni]) {                             // (the lanes of a pair agree: column counts are multiples of 8)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
          for (int e2 = 0; e2 < 8; ++e2) {
            const float v0 = acc[mi][ni][2 * e2], v1 = acc[mi][ni][2 * e2 + 1];
            unsigned word;
            if constexpr (MASKED) {
              // the slope is applied to the f32 value at the position it is stored at: values change lanes first
              const float keep = odd ? v1 : v0, send = odd ? v0 : v1;
              const float recv = __builtin_bit_cast(
                  float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0xB1, 0xF, 0xF, false));   // lane ^ 1
              f32x2 pr = {odd ? recv : keep, odd ? keep : recv};
              pr[0] *= __builtin_bit_cast(float, yy[mi][e2][ni] << 16) > 0.f ? 1.f : low;
              pr[1] *= __builtin_bit_cast(float, yy[mi][e2][ni] & 0xffff0000u) > 0.f ? 1.f : low;
              word = __builtin_bit_cast(unsigned int, __builtin_convertvector(pr, bf16x2));
            } else {
              // round both rows of the own column, fetch the neighbour's pair, keep the halves of this lane's row
              const f32x2 own2 = {v0, v1};
              const unsigned own = __builtin_bit_cast(unsigned int, __builtin_convertvector(own2, bf16x2));
              const unsigned nbr = (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0xB1, 0xF, 0xF, false);     // lane ^ 1
              word = __builtin_amdgcn_perm(nbr, own, sel);
            }
            __builtin_amdgcn_raw_buffer_store_b32(word, rsrcO, (int)(roffb[mi][e2] + colb + ni * 64), 0, 0);
          }
        }
      }
    }
    return;
  }
  float* colp = outp + n0 + wn * TN * 32 + r;
#pragma unroll
  for (int mi = 0; mi < TM; ++mi) {
    const int row_t = m0 + (wm * TM + mi) * 32 + 4 * h;      // first of the lane's 16 rows of this tile
    float* tilep = colp;
    if constexpr (MODE != MODE_DGRAD) tilep += (size_t)row_t * p.Ncols;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int dr = (e & 3) + 8 * (e >> 2);
      const int row = row_t + dr;
      if (row >= M) continue;
      float* rp;
      if constexpr (MODE == MODE_DGRAD) {
        const int n = vg_div_row(row, plane, inv_plane, fast), rem = row - n * plane;
        const int ii = vg_div_row(rem, roww, inv_roww, fast), jj = rem - ii * roww;
        rp = tilep + (size_t)((n * p.H + ii * p.sh + off_h) * p.W + jj * p.sw + off_w) * p.Ncols;
      } else {
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

// Result tensor of a launch that stores bf16 (`elements` of them): its size for the range-checked stores.
static inline int pgemm_set_out_h(PgemmParams& pp, long long elements, const char* who) {
  if (elements * 2 >= (1LL << 31))
    return vg_fail(VG_ERR_INVALID, "%s: the bf16 kernels store through 31-bit byte offsets; the result is 2 GiB or larger", who);
  pp.out_bytes = (unsigned)(elements * 2);
  return 0;
}

// True when a forward / input-gradient launch may run the KU form of the kernel with k-steps of `bk`.
static inline bool pgemm_uniform_k(int mode, const IgemmParams& p, int bk) {
  if (mode == MODE_FWD) return p.Ci % bk == 0 && p.KH * p.KW <= 32;
  if (mode == MODE_DGRAD) return p.Co % bk == 0 && ((p.KH + p.sh - 1) / p.sh) * ((p.KW + p.sw - 1) / p.sw) <= 32;
  if (p.Wo <= 0 || bk % p.Wo != 0) return false;
  const int rows = bk / p.Wo;                   // gradient rows per k-step
  return (p.Ho % rows == 0 || rows % p.Ho == 0) && p.Ktot % bk == 0 && p.kchunk % bk == 0;
}

// Tiles served: 128x128, 128x64, 64x128 (eight waves) and 64x64 (four waves). The KU form exists for one plane: the
// forward and input-gradient kernels with bf16 results and the weight gradient.
template <int MODE, int NP, bool OUT_H, bool MASKED, bool KU>
static inline int pgemm_launch_ku(const PgemmParams& pp, int bm, int bn, dim3 grid, hipStream_t st) {
  constexpr int BK = NP == 1 ? 64 : 32;     // a single plane leaves room for 64-deep tiles: half the barriers per flop
  if (bm == 128 && bn == 128) vg_launch_timed(pgemm_kernel<MODE, 128, 128, 2, NP, 8, BK, OUT_H, MASKED, KU>, grid, dim3(512), 0, st, pp);
  else if (bm == 128 && bn == 64) vg_launch_timed(pgemm_kernel<MODE, 128, 64, 4, NP, 8, BK, OUT_H, MASKED, KU>, grid, dim3(512), 0, st, pp);
  else if (bm == 64 && bn == 128) vg_launch_timed(pgemm_kernel<MODE, 64, 128, 2, NP, 8, BK, OUT_H, MASKED, KU>, grid, dim3(512), 0, st, pp);
  else if (bm == 64 && bn == 64) vg_launch_timed(pgemm_kernel<MODE, 64, 64, 2, NP, 4, BK, OUT_H, MASKED, KU>, grid, dim3(256), 0, st, pp);
  else return -1;
  return 0;
}
template <int MODE, int NP, bool OUT_H = false, bool MASKED = false>
static inline int pgemm_launch(const PgemmParams& pp, int bm, int bn, dim3 grid, hipStream_t st) {
  if constexpr (NP == 1 && (OUT_H || MODE == MODE_WGRAD)) {
    if (pgemm_uniform_k(MODE, pp.g, 64)) return pgemm_launch_ku<MODE, NP, OUT_H, MASKED, true>(pp, bm, bn, grid, st);
  }
  return pgemm_launch_ku<MODE, NP, OUT_H, MASKED, false>(pp, bm, bn, grid, st);
}

// conv_planes.hip
size_t vg_planes_bytes(long long elements, int np);   // workspace bytes for the planes of one operand
int vg_split_planes(const float* in, long long elements, int np, void* planes, hipStream_t st);
long long vg_plane_stride(long long elements);
int vg_planes_applicable(int mode, const vg_conv_geom* g);
int vg_planes_of(int precision);
bool vg_bf16_uses_planes();

// true when this precision runs (mode, g) on pre-converted operands
static inline bool pgemm_serves(int precision, int mode, const vg_conv_geom* g) {
  if (precision == VG_PREC_F32 || !vg_planes_applicable(mode, g)) return false;
  return precision == VG_PREC_F32_SPLIT || vg_bf16_uses_planes();
}
size_t vg_planes_operand_bytes(int mode, const vg_conv_geom* g, int np);

// Splits both operands into `workspace` (A's planes first) and fills the plane fields of `pp`.
// An operand whose planes the caller already holds (a_planes / b_planes, as vg_split_planes wrote them) is taken
// from there.
static inline int pgemm_prepare(PgemmParams& pp, const float* A, long long a_elems, const float* B, long long b_elems,
                                int np, void* workspace, size_t workspace_bytes, hipStream_t st, const char* who,
                                const void* a_planes = nullptr, const void* b_planes = nullptr) {
  const size_t need = (a_planes ? 0 : vg_planes_bytes(a_elems, np)) + (b_planes ? 0 : vg_planes_bytes(b_elems, np));
  if (need > 0 && (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 255) != 0))
    return vg_fail(VG_ERR_WORKSPACE, "%s: split mode needs a 256-byte aligned workspace of %zu bytes, got %zu", who, need,
                   workspace_bytes);
  if (((uintptr_t)a_planes & 15) != 0 || ((uintptr_t)b_planes & 15) != 0)
    return vg_fail(VG_ERR_INVALID, "%s: operand planes must be 16-byte aligned", who);
  char* ws = (char*)workspace;
  if (!a_planes) {
    vg_split_planes(A, a_elems, np, ws, st);
    a_planes = ws;
    ws += vg_planes_bytes(a_elems, np);
  }
  if (!b_planes) {
    vg_split_planes(B, b_elems, np, ws, st);
    b_planes = ws;
  }
  pp.Ap = (const __bf16*)a_planes;
  pp.Bp = (const __bf16*)b_planes;
  pp.a_plane = vg_plane_stride(a_elems);
  pp.b_plane = vg_plane_stride(b_elems);
  // the kernel addresses every operand with 32-bit byte offsets into a range-checked buffer; an offset with the top bit
  // set stands for "do not read", so the planes of an operand together must stay below 2 GiB
  const long long a_bytes = ((long long)(np - 1) * pp.a_plane + a_elems) * 2, b_bytes = ((long long)(np - 1) * pp.b_plane + b_elems) * 2;
  if (a_bytes >= (1LL << 31) || b_bytes >= (1LL << 31))
    return vg_fail(VG_ERR_INVALID, "%s: the bf16 kernels address operands with 31-bit byte offsets; an operand's planes are 2 GiB or larger", who);
  pp.a_bytes = (unsigned)a_bytes;
  pp.b_bytes = (unsigned)b_bytes;
  return 0;
}
