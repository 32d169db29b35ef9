// igemm_bf16.h — the implicit-GEMM convolution of igemm.h on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16),
// for fp32 tensors in HBM.  Two precisions:
//   X3 = false : operands rounded to bf16, fp32 accumulate (BASELINE config 5's compute: ~3 significant digits).
//   X3 = true  : each fp32 operand split into hi = bf16(x), lo = bf16(x - hi); products hi*hi + hi*lo + lo*hi
//                (the lo*lo term, 2^-16 relative, is dropped).  ~1e-5 relative accuracy at 16/3 x the fp32 MFMA rate.
// Same three contraction modes, pixel tables, loaders, XCD-aware tile order, split-K and epilogues as the fp32 kernel;
// what changes is the staging: fp32 global -> registers -> convert -> bf16 planes in LDS, and the fragment reads.
// MFMA 32x32x16 wants 8 consecutive k per lane for BOTH operands:
//   * operands whose tile is k-contiguous in memory (im2col rows in FWD / BWD_D, the filter in BWD_D) are stored
//     [row][k] and read with one ds_read_b128 per fragment (row stride 80 B: conflict-free);
//   * operands whose tile has k as the SLOW axis (filter [k][n] in FWD; im2col [pixel][rsc] and dz [pixel][n] in
//     BWD_F) are stored as they arrive and read with the hardware-transposing ds_read_b64_tr_b16 (two per fragment;
//     row stride = cols + 32 elements puts the 4 rows of a block 16 banks apart: conflict-free).
#pragma once
#include "igemm.h"

namespace a3d {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int MODE, int BM, int BN, bool X3>
struct Bf16Cfg {
  // k-tile: 64 for the plain bf16 kernel (half as many barriers and LDS round trips per MAC as 32; 78 KB of LDS, two
  // blocks per CU), 32 for the two-plane x3 variant (its planes would not fit otherwise)
  static constexpr int BK = X3 ? 32 : 64;
#ifndef A3D_BF16_WAVES
#define A3D_BF16_WAVES 8
#endif
  // 8 waves of 32 x (BN/2), or (diagnostic / tuning build: -DA3D_BF16_WAVES=4) 4 waves of 64 x (BN/2): twice the MFMAs
  // per wave, barrier and fragment read, half the wavefronts per CU
  static constexpr int NWAVES = A3D_BF16_WAVES, NT = 64 * NWAVES, WAVES_M = NWAVES / 2, WAVES_N = 2;
  static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  static constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1 && TM * 32 * WAVES_M == BM && TN * 32 * WAVES_N == BN, "tile");
  static constexpr bool A_KC = (MODE != MODE_BWD_F);     // A tile k-contiguous?
  static constexpr bool B_KC = (MODE == MODE_BWD_D);
  static constexpr int A_ROWS = A_KC ? BM : BK, A_COLS = A_KC ? BK : BM;     // same tile shapes as the fp32 kernel
  static constexpr int B_ROWS = B_KC ? BN : BK, B_COLS = B_KC ? BK : BN;
  static constexpr int A_LD = A_KC ? BK + 8 : BM + 32;                         // bf16 elements
  static constexpr int B_LD = B_KC ? BK + 8 : BN + 32;
  static constexpr int A_ELEMS = A_ROWS * A_LD, B_ELEMS = B_ROWS * B_LD;
  static constexpr int PLANES = X3 ? 2 : 1;
  static constexpr int BUF_ELEMS = PLANES * (A_ELEMS + B_ELEMS);
  static constexpr int PIX = A_ROWS;
  static constexpr size_t TILE_BYTES = (size_t)2 * BUF_ELEMS * 2;
  static constexpr size_t LDS_BYTES = TILE_BYTES + (size_t)2 * PIX * 16;
  static_assert((A_ELEMS * 2) % 16 == 0 && (B_ELEMS * 2) % 16 == 0, "plane alignment");
  static_assert(TILE_BYTES >= (size_t)(NT / (BN / 4)) * BN * 4, "bias-gradient scratch must fit in the tile buffers");
};

// fp32 registers of a loader tile -> bf16 plane(s), same [row][col] orientation as the global tile
template <class Tile, bool X3, bool PLAIN, int NT>
__device__ __forceinline__ void store_bf16(const float (&regs)[Tile::NL][4], __bf16* hi, __bf16* lo, int ld, int tid) {
#pragma unroll
  for (int j = 0; j < Tile::NL; ++j) {
    int row, cq;
    if constexpr (PLAIN) {
      const int idx = tid + j * NT;
      if (Tile::TOTAL % NT != 0 && idx >= Tile::TOTAL) continue;
      row = idx / Tile::CPR;
      cq = idx % Tile::CPR;
    } else {
      static_assert(!Tile::PARTIAL, "tile must cover all threads");
      row = tid / Tile::CPR + j * Tile::RPP;
      cq = tid % Tile::CPR;
    }
    bf16x4 h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h[e] = (__bf16)regs[j][e];
      if (X3) l[e] = (__bf16)(regs[j][e] - (float)h[e]);
    }
    *reinterpret_cast<bf16x4*>(hi + row * ld + cq * 4) = h;
    if (X3) *reinterpret_cast<bf16x4*>(lo + row * ld + cq * 4) = l;
  }
}

// registers of a loader tile that was gathered from a bf16 tensor through its "float view" (a float = two adjacent
// bf16 channels; 4 floats = 8 bf16 = one 16-byte chunk) -> bf16 plane, bit for bit
template <class Tile, bool PLAIN, int NT>
__device__ __forceinline__ void store_raw16(const float (&regs)[Tile::NL][4], __bf16* hi, int ld, int tid) {
#pragma unroll
  for (int j = 0; j < Tile::NL; ++j) {
    int row, cq;
    if constexpr (PLAIN) {
      const int idx = tid + j * NT;
      if (Tile::TOTAL % NT != 0 && idx >= Tile::TOTAL) continue;
      row = idx / Tile::CPR;
      cq = idx % Tile::CPR;
    } else {
      row = tid / Tile::CPR + j * Tile::RPP;
      cq = tid % Tile::CPR;
      if (Tile::PARTIAL && row >= Tile::NROWS) continue;
    }
    const f32x4 v = {regs[j][0], regs[j][1], regs[j][2], regs[j][3]};
    *reinterpret_cast<f32x4*>(hi + row * ld + cq * 8) = v;
  }
}
__device__ __forceinline__ float bf16_lo(float packed) { return __uint_as_float(__float_as_uint(packed) << 16); }
__device__ __forceinline__ float bf16_hi(float packed) { return __uint_as_float(__float_as_uint(packed) & 0xffff0000u); }

__device__ __forceinline__ bf16x4 lds_read_tr(const __bf16* p) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  return __builtin_bit_cast(bf16x4, v);
}

// fragment of a k-contiguous plane: 8 bf16 at [row][koff]
__device__ __forceinline__ bf16x8 frag_kc(const __bf16* plane, int ld, int row, int koff) {
  return *reinterpret_cast<const bf16x8*>(plane + row * ld + koff);
}
// fragment of a [k][col] plane for the 32 columns col0..col0+31: lane l gets column col0 + (l & 31),
// k = kbase + 8*(l >> 5) + 0..7
__device__ __forceinline__ bf16x8 frag_tr(const __bf16* plane, int ld, int col0, int kbase, int lane) {
  const int g = lane >> 4, l16 = lane & 15;
  const int q = l16 >> 2, pp = l16 & 3, h = g >> 1, cb = g & 1;
  const __bf16* a0 = plane + (kbase + 8 * h + q) * ld + col0 + 16 * cb + 4 * pp;
  bf16x4 lo4 = lds_read_tr(a0);
  bf16x4 hi4 = lds_read_tr(a0 + 4 * ld);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = lo4[e]; r[4 + e] = hi4[e]; }
  return r;
}

// A16 / B16: the A / B operand tensor is stored as bf16 in HBM (BASELINE config 5: bf16 activations and weight copies);
// it is gathered through its float view (half as many "channels", 16-byte chunks of 8 bf16) and lands in the LDS plane
// unconverted.  C16: the output tensor (and, in BWD_D, the ReluGrad mask, which is the same activation) is bf16; split-K
// slabs and filter gradients stay fp32.
// The whole GEMM of one block: `nwg` blocks work on problem `p`, this one is number `bid_in`.
template <int MODE, int BM, int BN, bool X3, bool A16, bool B16, bool C16>
__device__ __forceinline__ void igemm_bf16_body(const IgemmParams& p, const uint32_t nwg, const uint32_t bid_in) {
  using Cfg = Bf16Cfg<MODE, BM, BN, X3>;
  constexpr int BK = Cfg::BK, NT = Cfg::NT, TM = Cfg::TM, TN = Cfg::TN;
  constexpr bool TRANSPOSED = (MODE == MODE_BWD_D);
  static_assert(!(X3 && (A16 || B16)), "the split-operand mode needs fp32 sources");
  using ATile = Im2colTile<NT, Cfg::A_ROWS, A16 ? Cfg::A_COLS / 2 : Cfg::A_COLS, 4, TRANSPOSED>;
  using BTile = typename std::conditional<MODE == MODE_BWD_D, FilterTTile<NT, Cfg::B_ROWS, B16 ? Cfg::B_COLS / 2 : Cfg::B_COLS, 4>,
                                          PlainTile<NT, Cfg::B_ROWS, B16 ? Cfg::B_COLS / 2 : Cfg::B_COLS, 4>>::type;
  static_assert(!ATile::PARTIAL, "A tile must cover all threads");
  constexpr bool B_PLAIN = (MODE != MODE_BWD_D);
  // parameter view of a bf16 tensor as floats: half the channels per pixel / per filter tap
  IgemmParams ph = p;
  ph.ld = p.ld / 2; ph.Cg = p.Cg / 2; ph.nrsc = p.nrsc / 2; ph.K = p.K / 2; ph.div_c = p.div_c_half;
  const IgemmParams& pA = A16 ? ph : p;
  const IgemmParams& pB = B16 ? ph : p;
  constexpr int AKS = A16 ? BK / 2 : BK;       // k-tile extent of the A / B gather in its own units
  constexpr int BKS = B16 ? BK / 2 : BK;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* tiles = reinterpret_cast<__bf16*>(smem_raw);
  int4* pixtab = reinterpret_cast<int4*>(smem_raw + Cfg::TILE_BYTES);
  auto A_hi = [&](int buf) { return tiles + buf * Cfg::BUF_ELEMS; };
  auto A_lo = [&](int buf) { return tiles + buf * Cfg::BUF_ELEMS + Cfg::A_ELEMS; };
  auto B_hi = [&](int buf) { return tiles + buf * Cfg::BUF_ELEMS + Cfg::PLANES * Cfg::A_ELEMS; };
  auto B_lo = [&](int buf) { return tiles + buf * Cfg::BUF_ELEMS + Cfg::PLANES * Cfg::A_ELEMS + Cfg::B_ELEMS; };

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

  uint32_t bid = bid_in;
  {
    uint32_t q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_mn = p.tiles_m * p.tiles_n;
  const int split = bid / tiles_mn;
  const int tmn = bid - split * tiles_mn;
  const int tile_m = tmn / p.tiles_n, tile_n = tmn - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int nk_total = (p.K + BK - 1) / BK;
  const int kt_begin = split * p.ktiles_per_split;
  int kt_end = kt_begin + p.ktiles_per_split;
  if (kt_end > nk_total) kt_end = nk_total;
  const int nkt = kt_end - kt_begin;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  float ra[ATile::NL][4];
  float rb[BTile::NL][4];
  // BiasAddGrad (BWD_F): every thread sums, in fp32 and over the whole K range, the dz values it stages (its 4
  // columns, its rows of each tile); one LDS reduction in the epilogue
  const bool do_bias = (MODE == MODE_BWD_F) && p.dbias != nullptr && tile_m == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // B16: the thread's chunk is 8 columns

  const int a_cq = tid % ATile::CPR;
  const int b_cq = tid % BTile::CPR;

  // ===== staging, as in igemm_body (igemm.h): whatever does not change from k-tile to k-tile is computed once per lane,
  // the filter tap is decoded with scalar instructions when it is wave-uniform, and every global access is a raw buffer
  // load whose out-of-range offset reads as zero — no predicated branches, no zero-line selects.  A bf16 tensor is seen
  // through its float view (pA / pB: half the channels), so the same code serves fp32 and bf16 operands.
  constexpr int A_CPR = ATile::CPR, A_RPP = ATile::RPP, A_NL = ATile::NL;
  constexpr int SGN = TRANSPOSED ? -1 : 1;
  // wave-uniform filter tap of a k-tile: the host's decode parameters for this kernel's k-tile length
  const bool t_uni = BK == 64 ? p.uni64 != 0 : p.uni != 0, t_kperm = BK == 64 ? p.kperm64 != 0 : p.kperm != 0;
  const int t_cpt = BK == 64 ? p.cpt64 : p.cpt;
  const FastDiv t_div_cpt = BK == 64 ? p.div_cpt64 : p.div_cpt;
  const int a_r0 = tid / A_CPR;
  const int pW = p.W, pldA = pA.ld;
  const float* Abase = p.A;
  const unsigned long long a_total = A16 ? p.a_elems / 2 : p.a_elems;      // extent of A in view floats
  const unsigned long long b_total = B16 ? p.b_elems / 2 : p.b_elems;
  auto image_of = [&](int pixel) -> uint32_t {
    const int px = pixel < p.npix ? pixel : p.npix - 1;
    return fdiv((uint32_t)px, p.div_phw);
  };
  auto row_entry = [&](int pixel, uint32_t nf) -> int4 {      // {byte offset from image nf, y0, x0, valid}
    int4 e = make_pix<TRANSPOSED>(p, pixel);
    e.x = ((e.x - (int)(nf * (uint32_t)p.pHW)) + e.y * pW + e.z) * pldA * 4;
    return e;
  };
  int a_rowoff[A_NL], a_y0[A_NL], a_x0[A_NL];
  int a_dy = 0, a_dx = 0, a_coloff = 0;
  bool a_cvalid = true;
  const float* a_base = Abase;
  unsigned long long a_bytes = 0;
  if constexpr (MODE == MODE_BWD_F) {
    if (tid < 2 * Cfg::PIX) {
      const int which = tid / Cfg::PIX, e = tid % Cfg::PIX;
      const int pix0 = (kt_begin + which) * BK;
      pixtab[which * Cfg::PIX + e] = row_entry(pix0 + e, image_of(pix0));
    }
    const ColDec d = decode_col(pA, (A16 ? m0 / 2 : m0) + a_cq * 4);
    a_dy = d.r; a_dx = d.s;
    a_coloff = ((d.r * pW + d.s) * pldA + d.c) * 4;
    a_cvalid = d.valid;
  } else {
    const uint32_t nf = image_of(m0);
    if (tid < Cfg::PIX) pixtab[tid] = row_entry(m0 + tid, nf);
    const unsigned long long boff = (unsigned long long)nf * (unsigned long long)p.pHW * (unsigned long long)pldA;
    a_base = Abase + boff;
    a_bytes = (a_total - boff) * 4ull;
  }
  // ---- k-tile table (FWD / BWD_D, wave-uniform taps): what changes from k-tile to k-tile — the tap's (dy, dx), the A
  // gather's byte offset, the B tile's byte offset — is decoded ONCE per block, one thread per k-tile, into the unused
  // half of the row table; the loop reads one 16-byte entry (broadcast) instead of redoing ~70 scalar instructions per
  // wave and tile on the CU's one scalar unit (16 waves: more scalar cycles per tile than its 8 MFMAs take).  The entry
  // past the last tile makes every offset out of range: the loads of a tile that does not exist return zeros.
  int4* kttab = pixtab + Cfg::PIX;
  const bool use_tab = (MODE != MODE_BWD_F) && t_uni && nkt + 1 <= Cfg::PIX && b_total * 4ull < 0x80000000ull;
  if (use_tab && tid <= nkt) {
    int4 e = {0, 0, (int)kOOB, (int)kOOB};
    if (tid < nkt) {
      const uint32_t c1 = fdiv((uint32_t)(kt_begin + tid), p.div_taps), r1 = (uint32_t)(kt_begin + tid) - c1 * (uint32_t)p.ntaps;
      const uint32_t r2 = fdiv((uint32_t)(kt_begin + tid), t_div_cpt), c2 = (uint32_t)(kt_begin + tid) - r2 * (uint32_t)t_cpt;
      const uint32_t trs = t_kperm ? r1 : r2, chunk = t_kperm ? c1 : c2;
      const uint32_t r = fdiv(trs, p.div_s), sx = trs - r * p.div_s.d;
      e.x = SGN * (int)r;
      e.y = SGN * (int)sx;
      e.z = ((e.x * pW + e.y) * pldA + (int)chunk * AKS) * 4;
      if constexpr (MODE == MODE_BWD_D) {
        const uint32_t rs = (p.tap_r0 + p.sub_step * r) * p.S_full + p.tap_s0 + p.sub_step * sx;
        e.w = (int)((rs * (uint32_t)(p.Cn * pB.Cg) + (uint32_t)n0 * (uint32_t)pB.Cg + chunk * (uint32_t)BKS) * 4u);
      } else {
        const int ldb_vv = B16 ? p.ldb / 2 : p.ldb;
        e.w = (int)((trs * (uint32_t)p.Cg + chunk * (uint32_t)BK) * (uint32_t)ldb_vv * 4u);
      }
    }
    kttab[tid] = e;
  }
  __syncthreads();
  if constexpr (MODE != MODE_BWD_F) {
#pragma unroll
    for (int j = 0; j < A_NL; ++j) {
      const int4 pt = pixtab[a_r0 + j * A_RPP];
      a_rowoff[j] = pt.x + (use_tab ? a_cq * 16 : 0);
      a_y0[j] = pt.w ? pt.y : -(1 << 30);
      a_x0[j] = pt.z;
    }
  }
  // a bf16 B tile is read in 16-byte chunks up to its row stride: pad columns (fine/first's 64th channel) are zeros
  const int ldb_v = B16 ? p.ldb / 2 : p.ldb, n0_v = B16 ? n0 / 2 : n0, nn_v = B16 ? p.ldb / 2 : p.N;
  constexpr int B_CPR = BTile::CPR;
  uint32_t b_voff[BTile::NL];
  if constexpr (MODE == MODE_BWD_D) {
    const int r0 = tid / B_CPR;
#pragma unroll
    for (int j = 0; j < BTile::NL; ++j) {
      const int row = r0 + j * BTile::RPP;
      const bool ok = (!BTile::PARTIAL || r0 < Cfg::B_ROWS) && n0 + row < p.N;
      b_voff[j] = ok ? (uint32_t)((row * pB.Cg + b_cq * 4) * 4) : kOOB;
    }
  } else {
#pragma unroll
    for (int j = 0; j < BTile::NL; ++j) {
      const int idx = tid + j * NT;
      const int r = idx / B_CPR, cq = idx % B_CPR;
      const bool ok = (BTile::TOTAL % NT == 0 || idx < BTile::TOTAL) && n0_v + cq * 4 < nn_v;
      b_voff[j] = ok ? (uint32_t)((r * ldb_v + cq * 4) * 4) : kOOB;
    }
  }

  // loads of k-tile kt into ra / rb (slot: row-table buffer of that tile, BWD_F); live = false: a tile that does not exist
  auto stage = [&](auto uni_c, int kt_real, int slot, bool live) {
    constexpr bool UNI = decltype(uni_c)::value;
#ifdef A3D_HACK_FIXED_KT
    // timing-only diagnostic build (never shipped, results are wrong): every iteration stages the FIRST k-tile, so all of
    // the per-tile address generation is loop-invariant and leaves the loop — what the kernel would cost without it
    const int kt = kt_begin;
    (void)kt_real;
#else
    const int kt = kt_real;
#endif
    __amdgpu_buffer_rsrc_t rsA, rsB;
    uint32_t tab_boff = 0;        // UNI: byte offset of the B tile (k-tile table)
    // ---------- A ----------
    if constexpr (MODE == MODE_BWD_F) {
      const int pix0 = kt * BK;
      const uint32_t nf = image_of(pix0);
      const unsigned long long boff = (unsigned long long)nf * (unsigned long long)p.pHW * (unsigned long long)pldA;
      rsA = make_rsrc(Abase + boff, live ? (a_total - boff) * 4ull : 0ull);
      const int4* ptab = pixtab + slot * Cfg::PIX;
#pragma unroll
      for (int j = 0; j < A_NL; ++j) {
        const int4 pt = ptab[a_r0 + j * A_RPP];
        const int y = pt.y + a_dy, x = pt.z + a_dx;
        const bool ok = a_cvalid & (pt.w != 0) & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)pW);
        load_vec_buf<4>(rsA, ok ? (uint32_t)(pt.x + a_coloff) : kOOB, ra[j]);
      }
    } else {
      int dy, dx, coloff;
      bool cv = true;
      if constexpr (UNI) {
        const int4 e = kttab[live ? kt - kt_begin : nkt];      // one address for the whole wave: a broadcast read
        dy = e.x; dx = e.y; coloff = e.z;
        tab_boff = (uint32_t)e.w;
      } else {
        const ColDec d = decode_col(pA, kt * AKS + a_cq * 4);
        dy = SGN * d.r; dx = SGN * d.s;
        coloff = ((dy * pW + dx) * pldA + d.c) * 4;
        cv = d.valid;
      }
      rsA = make_rsrc(a_base, (UNI || live) ? a_bytes : 0ull);     // table entries past the end point out of range
#pragma unroll
      for (int j = 0; j < A_NL; ++j) {
        const int y = a_y0[j] + dy, x = a_x0[j] + dx;
        const bool ok = cv & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)pW);
        load_vec_buf<4>(rsA, ok ? (uint32_t)(a_rowoff[j] + coloff) : kOOB, ra[j]);
      }
    }
    // ---------- B ----------
    if constexpr (MODE == MODE_BWD_D) {
      // filter W[rs][cin][cout] read as rows = cin, columns = k = (rs, cout), in view floats
      if constexpr (UNI) {
        rsB = make_rsrc(p.B, b_total * 4ull);                     // the tile's offset comes from the table
#pragma unroll
        for (int j = 0; j < BTile::NL; ++j)
          load_vec_buf<4>(rsB, ((b_voff[j] | tab_boff) >> 31) ? kOOB : b_voff[j] + tab_boff, rb[j]);
      } else {
        const int kcol = kt * BKS + b_cq * 4;
        const bool kvalid = kcol < pB.K;
        const uint32_t k = kvalid ? (uint32_t)kcol : 0u;
        const uint32_t rs0 = fdiv(k, pB.div_c);
        const int ko = (int)(k - rs0 * pB.div_c.d);
        const uint32_t rp = fdiv(rs0, p.div_s), sp = rs0 - rp * p.div_s.d;
        const uint32_t rs = (p.tap_r0 + p.sub_step * rp) * p.S_full + p.tap_s0 + p.sub_step * sp;
        const uint32_t base = (rs * (uint32_t)(p.Cn * pB.Cg) + (uint32_t)n0 * (uint32_t)pB.Cg + (uint32_t)ko) * 4u;
        rsB = make_rsrc(p.B, live ? b_total * 4ull : 0ull);
#pragma unroll
        for (int j = 0; j < BTile::NL; ++j)
          load_vec_buf<4>(rsB, (kvalid && b_voff[j] != kOOB) ? base + (b_voff[j] - (uint32_t)(b_cq * 16)) : kOOB, rb[j]);
      }
    } else {
      if constexpr (MODE == MODE_FWD && UNI) {
        // columns n0.. of the whole [K][ldb] filter; the tile's first row comes from the table (rows past K - 1 lie
        // past the descriptor's end)
        const long long rec = ((long long)p.K * ldb_v - n0_v) * 4ll;
        rsB = make_rsrc(p.B + n0_v, (unsigned long long)(rec > 0 ? rec : 0ll));
#pragma unroll
        for (int j = 0; j < BTile::NL; ++j)
          load_vec_buf<4>(rsB, ((b_voff[j] | tab_boff) >> 31) ? kOOB : b_voff[j] + tab_boff, rb[j]);
        return;
      }
      // plain [K][ldb] tile at rows row0.., columns n0..: the descriptor is re-based and ends with row K-1
      const int row0 = kt * BK;
      const int rows = p.K - row0;
      const long long rec = ((long long)(rows > 0 ? rows : 0) * ldb_v - n0_v) * 4ll;
      rsB = make_rsrc(p.B + ((unsigned long long)row0 * (unsigned long long)ldb_v + (unsigned long long)n0_v),
                      (unsigned long long)((live && rec > 0) ? rec : 0ll));
#pragma unroll
      for (int j = 0; j < BTile::NL; ++j) load_vec_buf<4>(rsB, b_voff[j], rb[j]);
    }
  };
  // BiasAddGrad (BWD_F): summed from the staged dz registers when they are parked in LDS (their data has arrived then)
  auto add_bias = [&]() {
    if (MODE == MODE_BWD_F && do_bias) {
#pragma unroll
      for (int j = 0; j < BTile::NL; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (B16) { bsum[2 * e] += bf16_lo(rb[j][e]); bsum[2 * e + 1] += bf16_hi(rb[j][e]); }
          else bsum[e] += rb[j][e];
        }
    }
  };
  auto store_tiles = [&](int buf) {
    if constexpr (A16) store_raw16<ATile, false, NT>(ra, A_hi(buf), Cfg::A_LD, tid);
    else store_bf16<ATile, X3, false, NT>(ra, A_hi(buf), A_lo(buf), Cfg::A_LD, tid);
    if constexpr (B16) store_raw16<BTile, B_PLAIN, NT>(rb, B_hi(buf), Cfg::B_LD, tid);
    else store_bf16<BTile, X3, B_PLAIN, NT>(rb, B_hi(buf), B_lo(buf), Cfg::B_LD, tid);
  };

  auto k_loop = [&](auto uni_c) {
  // FWD / BWD_D: a tile's loads are in flight for a whole iteration — requested at the top of iteration it - 1 (right
  // after the registers they land in were parked in LDS), parked at the top of iteration it, read from LDS in iteration
  // it + 1.  (BWD_F keeps the shorter schedule below: its row table is double-buffered on the same cadence.)
  constexpr bool EARLY = MODE != MODE_BWD_F;
  if (nkt > 0) {
    stage(uni_c, kt_begin, 0, true);
    add_bias();
    store_tiles(0);
  }
  if constexpr (EARLY) stage(uni_c, kt_begin + 1, 1, nkt > 1);
  __syncthreads();

  int cur = 0;
  for (int it = 0; it < nkt; ++it) {
    const int kt = kt_begin + it;
    const bool more = it + 1 < nkt;
    if constexpr (EARLY) {
      if (more) store_tiles(cur ^ 1);              // tile it + 1; buffer cur ^ 1 was last read before the barrier above
      stage(uni_c, kt + 2, 0, it + 2 < nkt);       // a tile that does not exist: offsets out of range / 0 records
    } else {
      stage(uni_c, kt + 1, (it + 1) & 1, more);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == MODE_BWD_F) {
      if (tid < Cfg::PIX) pixtab[(it & 1) * Cfg::PIX + tid] = row_entry((kt + 2) * BK + tid, image_of((kt + 2) * BK));
    }
    const __bf16* ah = A_hi(cur);
    const __bf16* al = A_lo(cur);
    const __bf16* bh = B_hi(cur);
    const __bf16* bl = B_lo(cur);
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 a_hi[TM], a_lo[TM], b_hi[TN], b_lo[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const int arow0 = wm * Cfg::WM + a * 32;
        if (Cfg::A_KC) {
          a_hi[a] = frag_kc(ah, Cfg::A_LD, arow0 + li, 16 * s + 8 * lh);
          if (X3) a_lo[a] = frag_kc(al, Cfg::A_LD, arow0 + li, 16 * s + 8 * lh);
        } else {
          a_hi[a] = frag_tr(ah, Cfg::A_LD, arow0, 16 * s, lane);
          if (X3) a_lo[a] = frag_tr(al, Cfg::A_LD, arow0, 16 * s, lane);
        }
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int bcol0 = wn * Cfg::WN + b * 32;
        if (Cfg::B_KC) {
          b_hi[b] = frag_kc(bh, Cfg::B_LD, bcol0 + li, 16 * s + 8 * lh);
          if (X3) b_lo[b] = frag_kc(bl, Cfg::B_LD, bcol0 + li, 16 * s + 8 * lh);
        } else {
          b_hi[b] = frag_tr(bh, Cfg::B_LD, bcol0, 16 * s, lane);
          if (X3) b_lo[b] = frag_tr(bl, Cfg::B_LD, bcol0, 16 * s, lane);
        }
      }
      if (!EARLY && s == BK / 16 - 1 && more) {
        add_bias();
        store_tiles(cur ^ 1);
      }
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          if (X3) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[a], b_hi[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[a], b_lo[b], acc[a][b], 0, 0, 0);
          }
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[a], b_hi[b], acc[a][b], 0, 0, 0);
        }
    }
    __syncthreads();
    cur ^= 1;
  }
  };
  if constexpr (MODE == MODE_BWD_F) {
    k_loop(std::false_type{});
  } else {
    if (use_tab) k_loop(std::true_type{});       // wave-uniform taps through the k-tile table
    else k_loop(std::false_type{});
  }

  // ---- epilogue (as igemm_kernel) ----
  float* Cout = p.C;
  int ldc = p.ldc;
  const bool partial = p.splitk > 1;
  if (partial) {
    Cout = p.C + (size_t)split * p.slab;
    ldc = p.N;
  }
  if (MODE == MODE_BWD_F && p.dbias != nullptr && tile_m == 0) {        // block-uniform branch
    float* red = reinterpret_cast<float*>(smem_raw);                       // tile buffers are free now
    constexpr int CW = B16 ? 8 : 4;                                        // columns per thread chunk
    constexpr int RG = NT / (BN / CW);                                     // row groups: threads sharing a column chunk
    const int rg = tid / (BN / CW);
#pragma unroll
    for (int e = 0; e < CW; ++e) red[rg * BN + b_cq * CW + e] = bsum[e];
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float s = 0.f;
      for (int r = 0; r < RG; ++r) s += red[r * BN + tid];
      p.dbias[(partial ? (size_t)split * p.N : 0) + n0 + tid] = s;
    }
  }
  if (MODE == MODE_FWD && p.pool) {      // never split (host): rows 4w .. 4w+3 are the conv outputs of pool window w (make_pix)
    // the values a separate conv would have stored (rounded to bf16 if its output tensor is), compared the way MaxPool /
    // MaxPoolGrad scan them: first maximum wins (igemm.h: the epilogues by raw buffer stores)
    store_tile_pool_buf<BM, TM, TN, Cfg::WM, Cfg::WN, C16, false>(p, acc, m0, n0, wm, wn, li, lh, Cout, ldc, C16);
    return;
  }
  if (epi_buf_ok<MODE>(p, ldc, partial)) {
    store_tile_buf<MODE, BM, TM, TN, Cfg::WM, Cfg::WN, C16, false>(p, acc, m0, n0, wm, wn, li, lh, Cout, ldc, partial, C16 && !partial);
    return;
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int col = n0 + wn * Cfg::WN + b * 32 + li;
    if (col >= p.N) continue;
    float bias = 0.f;
    if (!partial && MODE == MODE_FWD && p.bias) bias = p.bias[col];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = m0 + wm * Cfg::WM + a * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
      if (row >= p.M) continue;
      float val = acc[a][b][v];
      const size_t o = (partial || MODE != MODE_BWD_D
                            ? (size_t)row
                            : remap_row(row, p.sub_step, p.sub_ph, p.sub_pw, p.outW, p.outHW, p.div_phw, p.div_pw)) *
                           ldc + col;
      if (!partial) {
        if (MODE == MODE_FWD) {
          val += bias;
          if (p.act == EPI_RELU) val = fmaxf(val, 0.f);
          else if (p.act == EPI_SIGMOID) val = 1.f / (1.f + expf(-val));
          if (p.keep) val = p.keep[(size_t)row * p.N + col] ? val * p.mask_scale : 0.f;
        } else if (MODE == MODE_BWD_D) {
          if (p.mask) {
            const float y = C16 ? (float)reinterpret_cast<const __bf16*>(p.mask)[o] : p.mask[o];
            val = apply_act_grad(val, y, p.mask_act, p.mask_scale);
          }
        }
      }
      if (C16 && !partial) reinterpret_cast<__bf16*>(Cout)[o] = (__bf16)val;
      else Cout[o] = val;
    }
  }
}

template <int MODE, int BM, int BN, bool X3, bool A16 = false, bool B16 = false, bool C16 = false>
__global__ __launch_bounds__(64 * A3D_BF16_WAVES, A3D_BF16_WAVES / 2) void igemm_bf16_kernel(const IgemmParams p) {
  igemm_bf16_body<MODE, BM, BN, X3, A16, B16, C16>(p, gridDim.x, blockIdx.x);
}

// Up to four independent problems in ONE launch (blockIdx.y selects the problem), as igemm_multi_kernel: the parity
// classes of a strided bwd-data (conv2d_4 at batch 64: four GEMMs of 96 tiles each, 14 - 40 us plus a split-K reduction
// apiece as separate launches).
template <int MODE, int BM, int BN, bool X3, bool A16, bool B16, bool C16>
__global__ __launch_bounds__(64 * A3D_BF16_WAVES, A3D_BF16_WAVES / 2) void igemm_bf16_multi_kernel(const IgemmMulti ps) {
  const IgemmParams& p = ps.p[blockIdx.y];
  const uint32_t nwg = (uint32_t)(p.tiles_m * p.tiles_n * p.splitk);
  if (blockIdx.x >= nwg) return;
  igemm_bf16_body<MODE, BM, BN, X3, A16, B16, C16>(p, nwg, blockIdx.x);
}

}  // namespace a3d
