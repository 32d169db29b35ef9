// igemm_ring.h — the implicit-GEMM convolution on the bf16 matrix cores for bf16-STORED operands (BASELINE config 5:
// bf16 activations and weight copies in HBM), forward, stride-1 bwd-data and bwd-filter, with its tiles staged by LDS-DMA.
//
// Why a second bf16 kernel.  igemm_bf16.h stages global -> registers -> LDS: per k-tile a wave walks the serial chain
// buffer_load x4 -> s_waitcnt vmcnt -> ds_write x4 -> s_barrier -> ds_read x20 -> 8 MFMA, all eight waves in lockstep behind
// the barrier; counters put 40 % of a wave's cycles at a wait or the barrier and the LDS write path (ds_write_b128: ~79
// bytes per clock and CU) costs as many cycles per k-tile as the MFMAs.  Here
//   * a k-tile (BK = 64) of both operands is written into LDS by `buffer_load_dwordx4 ... lds`: no VGPR round trip, no
//     ds_write, no conversion (the tensors are bf16 already); out-of-range pieces (padding halo, K / M / N tails) are
//     offsets past the descriptor's end and arrive as zeros;
//   * the tile of iteration it + 1 is requested right after the ONE barrier of iteration it and has that whole iteration
//     — 2-4 k LDS-read + MFMA cycles — to land; the wait at the next barrier is for loads issued an iteration ago;
//   * blocks are large (256 rows x 64 ... 256 columns, one block of eight waves per CU, wave tiles of 64 x 64 ... 128): half
//     the LDS read bytes per MFMA of the 32 x 64 wave tiles, and the registers to keep the fragments of k-step s + 1 in
//     flight behind the MFMAs of k-step s;
//   * address generation per lane and k-tile: one fast division (k -> filter tap, channel), one 16-byte table read
//     (the tap's dy, dx, byte offset), and an add + range test per piece.  Any channel count that is a multiple of 8 works
//     (conv2d_1's 96: a k-tile then straddles two taps, each 16-byte piece lies in one).
// LDS images (no padding is possible: a wave-instruction writes 1 KiB linearly, lane l at base + 16 l):
//   * k-contiguous operands (im2col rows; the filter in bwd-data) as [row][64 k] = 128-byte rows, 16-byte chunk c of row
//     r at chunk position c ^ ((r >> 1) & 7), applied to the per-lane SOURCE and again by the ds_read_b128 fragment read:
//     every 16-lane group of that read touches 16 different 16-byte slots;
//   * the forward filter tile as it lies in memory, [64 k][BN], read with the transposing ds_read_b64_tr_b16; 16-byte
//     chunk c of row k at c ^ ((k & 3) << 2) (BN = 64: c ^ (((k >> 1) & 1) << 2)): the four k rows a 16-lane group reads
//     land 64 bytes apart in the bank row.
#pragma once
#include "igemm_bf16.h"

namespace a3d {

typedef __attribute__((address_space(3))) void* ring_lptr_t;

template <int MODE, int BM, int BN, int WAVES_M>
struct RingCfg {
  static constexpr int BK = 64, NWAVES = 8, NT = 512, WAVES_N = NWAVES / WAVES_M;
  static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1 && TM * 32 * WAVES_M == BM && TN * 32 * WAVES_N == BN, "tile");
  static constexpr bool A_KC = MODE != MODE_BWD_F;           // A tile k-contiguous ([BM][64]) or pixel-major ([64 pixels][BM])
  static constexpr bool B_KC = MODE == MODE_BWD_D;           // B tile k-contiguous ([BN][64]) or as stored ([64][BN])
  static_assert(B_KC || BN == 64 || BN == 128 || BN == 256, "[64][BN] tile: 128-, 256- or 512-byte rows");
  static_assert(A_KC || BM == 128 || BM == 256, "[64][BM] tile: 256- or 512-byte rows");
  static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
  static constexpr int A_PIECES = A_BYTES / 1024, B_PIECES = B_BYTES / 1024;     // 1-KiB LDS-DMA wave-instructions
  static constexpr int A_NI = (A_PIECES + 7) / 8, B_NI = (B_PIECES + 7) / 8;     // ... per wave
  static_assert(A_PIECES % 8 == 0, "A pieces divide over the eight waves");
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int NSTAGE = 2;
  static constexpr int MAXTAPS = 128;
  // (the epilogue stages 32 x WN outputs per wave in the tile buffers: at most 8 x 32 x (4 WN + 16) bytes)
  static constexpr int TAB_ROWS = MODE == MODE_BWD_F ? 256 : BM;      // row table (BM entries) / bwd-filter: four k-tiles' pixel tables
  static constexpr size_t TILE_AND_TABLES = (size_t)NSTAGE * STAGE + (size_t)TAB_ROWS * 16 + (size_t)MAXTAPS * 16;
  static constexpr size_t EPI_BYTES = (size_t)8 * 32 * (WN * 4 + 16);
  static constexpr size_t LDS_BYTES = TILE_AND_TABLES > EPI_BYTES ? TILE_AND_TABLES : EPI_BYTES;
  // registers: one block per CU for the big tiles (two wavefronts per SIMD, up to 256 registers each)
  static constexpr int MIN_WAVES = (TM * TN <= 2) ? 4 : 2;
};

__device__ __forceinline__ int ring_swz(int row) { return (row >> 1) & 7; }

// Buffer descriptor in four scalar registers (make_rsrc's words; every word made provably wave-uniform: it is an "s"
// operand of the statement below).
typedef uint32_t ring_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ ring_u32x4 ring_rsrc(const void* base, unsigned long long bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  ring_u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes > 0x7fffffffull ? 0x7fffffffu : (uint32_t)bytes);
  r[3] = 0x00020000u;
  return r;
}
// One LDS-DMA wave-instruction: lane l's 16 bytes at byte offset `voff` of the buffer go to LDS byte address lds_addr + 16 l
// (zeros when voff is past the descriptor's end).  Inline assembly ON PURPOSE: the compiler orders every LDS read it cannot
// prove disjoint (the transposing ds_read_b64_tr_b16 among them) behind an LDS-DMA it knows of with s_waitcnt vmcnt(0) —
// right after the requests of the next tile, i.e. no overlap at all (seen in the ISA of the builtin form).  Hidden in an asm
// statement the request is invisible to that bookkeeping; the kernel waits for it itself, once per k-tile, ahead of the
// barrier that precedes the first read (ring_landed).  M0 is saved and restored (the compiler may hold a value in it); the
// s_nop covers both the M0 write -> LDS-DMA and the SGPR write -> VMEM descriptor read wait states.
__device__ __forceinline__ void ring_dma(ring_u32x4 rs, uint32_t voff, uint32_t lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_addr), "v"(voff), "s"(rs)
               : "memory");
}
// every LDS-DMA this wave has requested has landed
__device__ __forceinline__ void ring_landed() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int MODE, int BM, int BN, int WAVES_M, bool C16>
__global__ __launch_bounds__(512, (RingCfg<MODE, BM, BN, WAVES_M>::MIN_WAVES)) void igemm_ring_kernel(const IgemmParams p) {
  using Cfg = RingCfg<MODE, BM, BN, WAVES_M>;
  constexpr int BK = Cfg::BK, TM = Cfg::TM, TN = Cfg::TN;
  constexpr bool TRANSPOSED = (MODE == MODE_BWD_D);
  constexpr int SGN = TRANSPOSED ? -1 : 1;
  constexpr int ROWB = BN * 2;                    // bytes per k row of the forward filter tile
  constexpr int CPR = BN / 8;                     // its 16-byte chunks per row

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int4* rowtab = reinterpret_cast<int4*>(smem_raw + Cfg::NSTAGE * Cfg::STAGE);
  int4* taptab = rowtab + Cfg::TAB_ROWS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

  const uint32_t nwg = gridDim.x;
  uint32_t bid = blockIdx.x;
  {
    const uint32_t q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
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

  // ---- tables: one entry per row of the im2col tile {byte offset of its reference pixel from the base image, y0, x0,
  //      valid}, one per filter tap {dy, dx, byte offset of the tap in the gathered image, byte offset of the tap in the
  //      filter (bwd-data)} ----
  const int pW = p.W, pld = p.ld;
  uint32_t nf = 0;
  // bwd-filter: the pixel axis is K.  Table of k-tile kt (64 entries {byte offset of the pixel's reference position in the
  // whole tensor, y0, x0}) lives in slot kt & 3; it is written three iterations before the tile's offsets are computed
  auto write_pix = [&](int kt) {
    if (tid < 64) {
      int4 e = make_pix<false>(p, kt * BK + tid);
      e.x = (e.x + e.y * pW + e.z) * pld * 2;
      if (!e.w || kt >= kt_end) e.y = -(1 << 30);
      rowtab[(kt & 3) * 64 + tid] = e;
    }
  };
  if constexpr (MODE == MODE_BWD_F) {
#pragma unroll
    for (int t = 0; t < 4; ++t) write_pix(kt_begin + t);
  } else {
    const int px = m0 < p.npix ? m0 : p.npix - 1;
    nf = __builtin_amdgcn_readfirstlane(fdiv((uint32_t)px, p.div_phw));
  }
  if (MODE != MODE_BWD_F && tid < BM) {
    int4 e = make_pix<TRANSPOSED>(p, m0 + tid);
    e.x = ((e.x - (int)(nf * (uint32_t)p.pHW)) + e.y * pW + e.z) * pld * 2;
    if (!e.w) e.y = -(1 << 30);                   // rows past M fail the range test like any halo pixel
    rowtab[tid] = e;
  }
  for (int t = tid; MODE != MODE_BWD_F && t < p.ntaps; t += Cfg::NT) {
    const uint32_t r = fdiv((uint32_t)t, p.div_s), sx = (uint32_t)t - r * p.div_s.d;
    int4 e;
    e.x = SGN * (int)r;
    e.y = SGN * (int)sx;
    e.z = (e.x * pW + e.y) * pld * 2;
    e.w = TRANSPOSED ? (int)((uint32_t)t * (uint32_t)(p.Cn * p.Cg) * 2u) : 0;
    taptab[t] = e;
  }
  __syncthreads();

  // ---- per-lane constants of the LDS-DMA pieces.  Piece (j, wave) of a [rows][64 k] image covers rows 8 (8 j + wave) ..
  //      + 7, lane l its row (l >> 3) and chunk position l & 7; successive j are 64 rows apart: the same swizzle ----
  const int row0 = wave * 8 + (lane >> 3);
  const int kc = (lane & 7) ^ ring_swz(row0);     // this lane's SOURCE k-chunk (8 bf16) inside every k-tile
  int a_rowoff[Cfg::A_NI], a_y0[Cfg::A_NI], a_x0[Cfg::A_NI];
  // bwd-filter: the A image is [64 pixels][BM] — a lane's pieces are one fixed 16-byte column chunk (one filter tap, eight
  // channels: decoded once) of the pixel rows a_krow + 8 j (512 / BM) of every k-tile
  constexpr int CPRA = BM / 8;
  int a_krow = 0, a_dy = 0, a_dx = 0, a_coloff = 0;
  bool a_mvalid = true;
  if constexpr (MODE == MODE_BWD_F) {
    const int id = wave * 64 + lane;
    a_krow = id / CPRA;
    const int cpos = id % CPRA, sw = (a_krow & 3) << 2;
    const ColDec dc = decode_col(p, m0 + 8 * (cpos ^ sw));
    a_dy = dc.r; a_dx = dc.s;
    a_coloff = ((dc.r * pW + dc.s) * pld + dc.c) * 2;
    a_mvalid = dc.valid;
  } else {
#pragma unroll
    for (int j = 0; j < Cfg::A_NI; ++j) {
      const int4 e = rowtab[row0 + 64 * j];
      a_rowoff[j] = e.x; a_y0[j] = e.y; a_x0[j] = e.z;
    }
  }
  uint32_t b_voff[Cfg::B_NI];                     // loop-invariant part of the B pieces' offsets (kOOB: never valid)
#pragma unroll
  for (int j = 0; j < Cfg::B_NI; ++j) {
    if constexpr (Cfg::B_KC) {                    // rows = cin (GEMM N), k-contiguous: W[rs][cin][cout]
      const int row = row0 + 64 * j;
      b_voff[j] = (row < BN && n0 + row < p.N) ? (uint32_t)((n0 + row) * p.Cg) * 2u : kOOB;
    } else {                                      // [64 k][BN] as stored: filter [K][ldb]
      const int id = (j * 8 + wave) * 64 + lane;
      const int krow = id / CPR, cpos = id % CPR;
      const int sw = CPR >= 16 ? ((krow & 3) << 2) : (((krow >> 1) & 1) << 2);
      const int col = n0 + 8 * (cpos ^ sw);
      b_voff[j] = col < p.ldb ? (uint32_t)(krow * p.ldb + col) * 2u : kOOB;      // (pad columns of a row are zeros)
    }
  }
  const unsigned long long a_boff = (unsigned long long)nf * (unsigned long long)p.pHW * (unsigned long long)pld;
  const ring_u32x4 rsA = ring_rsrc(reinterpret_cast<const __bf16*>(p.A) + a_boff, (p.a_elems - a_boff) * 2ull);   // (bwd-filter: the whole tensor, < 2^31 bytes: host)
  const ring_u32x4 rsB = ring_rsrc(p.B, p.b_elems * 2ull);
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((ring_lptr_t)smem_raw);      // LDS byte address of the first stage

  // The requests of a k-tile in two steps, so that neither sits between the barrier and the first MFMA of an iteration
  // (where all eight waves would do it at once with the matrix pipe idle: counters of the first version, 48 % of the wave
  // cycles parked, 34 % matrix-pipe busy): prep() computes the pieces' offsets ONE TILE EARLY, in the shadow of the
  // MFMAs of the third k-step; fire_a / fire_b are then nothing but the LDS-DMAs, issued behind the first / second k-step.
  uint32_t aoff[Cfg::A_NI], boff[Cfg::B_NI];
  auto prep = [&](int kt) {
    if constexpr (MODE == MODE_BWD_F) {
      const int4* tab = rowtab + (kt & 3) * 64;
#pragma unroll
      for (int j = 0; j < Cfg::A_NI; ++j) {
        const int4 e = tab[a_krow + j * (8 * 64 / CPRA)];
        const int y = e.y + a_dy, x = e.z + a_dx;
        const bool ok = a_mvalid & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)pW);
        aoff[j] = ok ? (uint32_t)(e.x + a_coloff) : kOOB;
      }
#pragma unroll
      for (int j = 0; j < Cfg::B_NI; ++j)
        boff[j] = (b_voff[j] != kOOB && kt < kt_end) ? b_voff[j] + (uint32_t)kt * (uint32_t)(BK * p.ldb * 2) : kOOB;   // pixels past the end: past the end
      return;
    }
    const int k0 = kt * BK + kc * 8;
    const uint32_t tap = fdiv((uint32_t)k0, p.div_c);
    const int ch = k0 - (int)(tap * p.div_c.d);
    const bool kvalid = (k0 < p.K) & (kt < kt_end);
    const int4 tt = taptab[kvalid ? tap : 0u];
#pragma unroll
    for (int j = 0; j < Cfg::A_NI; ++j) {
      const int y = a_y0[j] + tt.x, x = a_x0[j] + tt.y;
      const bool ok = kvalid & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)pW);
      aoff[j] = ok ? (uint32_t)(a_rowoff[j] + tt.z + ch * 2) : kOOB;
    }
#pragma unroll
    for (int j = 0; j < Cfg::B_NI; ++j) {
      if constexpr (Cfg::B_KC) boff[j] = (kvalid && b_voff[j] != kOOB) ? b_voff[j] + (uint32_t)tt.w + (uint32_t)ch * 2u : kOOB;
      else boff[j] = (b_voff[j] != kOOB && kt < kt_end) ? b_voff[j] + (uint32_t)kt * (uint32_t)(BK * p.ldb * 2) : kOOB;   // rows past K: past the end
    }
  };
  auto fire_a = [&](int stg) {
    const uint32_t sa = lds0 + (uint32_t)(stg * Cfg::STAGE);
#pragma unroll
    for (int j = 0; j < Cfg::A_NI; ++j) ring_dma(rsA, aoff[j], sa + (uint32_t)((j * 8 + wave) * 1024));
  };
  auto fire_b = [&](int stg) {
    const uint32_t sb = lds0 + (uint32_t)(stg * Cfg::STAGE + Cfg::A_BYTES);
#pragma unroll
    for (int j = 0; j < Cfg::B_NI; ++j)
      if (Cfg::B_PIECES % 8 == 0 || j * 8 + wave < Cfg::B_PIECES) ring_dma(rsB, boff[j], sb + (uint32_t)((j * 8 + wave) * 1024));
  };

  // ---- fragment addresses (bytes inside a stage) ----
  int a_fr[TM], b_fr[TN];
  constexpr int ROWA = BM * 2;                    // bytes per pixel row of the bwd-filter A tile
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    if constexpr (Cfg::A_KC) {
      const int row = wm * Cfg::WM + a * 32 + li;
      a_fr[a] = row * 128 + ((lh ^ ring_swz(row)) << 4);        // k-step s: ^ (s << 5)
    } else {
      const int g = lane >> 4, l16 = lane & 15, q = l16 >> 2, pp = l16 & 3, h = g >> 1, cb = g & 1;
      const int c = ((wm * Cfg::WM + a * 32) >> 3) + 2 * cb + (pp >> 1);
      a_fr[a] = (8 * h + q) * ROWA + ((c ^ (q << 2)) << 4) + (pp & 1) * 8;                  // k-step s: + 16 s ROWA
    }
  }
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    if constexpr (Cfg::B_KC) {
      const int row = wn * Cfg::WN + b * 32 + li;
      b_fr[b] = Cfg::A_BYTES + row * 128 + ((lh ^ ring_swz(row)) << 4);
    } else {
      const int g = lane >> 4, l16 = lane & 15, q = l16 >> 2, pp = l16 & 3, h = g >> 1, cb = g & 1;
      const int c = ((wn * Cfg::WN + b * 32) >> 3) + 2 * cb + (pp >> 1);
      const int sw = CPR >= 16 ? (q << 2) : ((q >> 1) << 2);      // of rows 16 s + 8 h + q and + 4: the same
      b_fr[b] = Cfg::A_BYTES + (8 * h + q) * ROWB + ((c ^ sw) << 4) + (pp & 1) * 8;     // k-step s: + 16 s ROWB
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  // ---- main loop, rotated: the ONE barrier of an iteration sits BEFORE its last k-step.  At that point the wave holds the
  //      last k-step's fragments in registers (nothing of stage it & 1 will be read again) and tile it + 1 has landed, so
  //      behind the barrier the first fragments of tile it + 1 are read and the requests of tile it + 2 (into stage it & 1)
  //      are issued IN THE SHADOW of the last k-step's MFMAs.  With the barrier at the top of the iteration the first
  //      fragment reads, the requests' issue (60 - 180 cycles apiece) and the offsets' arithmetic all ran with the matrix
  //      pipe idle on all eight waves at once: a build without requests ran as long as (empty loop) + (MFMA time). ----
  bf16x8 af[2][TM], bf[2][TN];
  auto read_frags = [&](const unsigned char* st, int s, int buf) {
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      if constexpr (Cfg::A_KC) {
        af[buf][a] = *reinterpret_cast<const bf16x8*>(st + (a_fr[a] ^ (s << 5)));
      } else {
        const __bf16* q0 = reinterpret_cast<const __bf16*>(st + a_fr[a] + s * 16 * ROWA);
        const bf16x4 lo4 = lds_read_tr(q0), hi4 = lds_read_tr(q0 + 2 * ROWA);
#pragma unroll
        for (int e = 0; e < 4; ++e) { af[buf][a][e] = lo4[e]; af[buf][a][4 + e] = hi4[e]; }
      }
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      if constexpr (Cfg::B_KC) {
        bf[buf][b] = *reinterpret_cast<const bf16x8*>(st + (b_fr[b] ^ (s << 5)));
      } else {
        const __bf16* q0 = reinterpret_cast<const __bf16*>(st + b_fr[b] + s * 16 * ROWB);
        const bf16x4 lo4 = lds_read_tr(q0), hi4 = lds_read_tr(q0 + 2 * ROWB);      // rows k and k + 4 (ROWB / 2 elements each)
#pragma unroll
        for (int e = 0; e < 4; ++e) { bf[buf][b][e] = lo4[e]; bf[buf][b][4 + e] = hi4[e]; }
      }
    }
  };
  auto mfmas = [&](int buf) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[buf][a], bf[buf][b], acc[a][b], 0, 0, 0);
  };
  constexpr int NREADS = (Cfg::A_KC ? TM : 2 * TM) + (Cfg::B_KC ? TN : 2 * TN), NMFMA = TM * TN;
  constexpr int PER = (NREADS + NMFMA - 1) / NMFMA;
  auto interleave = [&]() {                       // one MFMA, then its share of the LDS reads issued ahead of the group
#pragma unroll
    for (int i = 0; i < NMFMA; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, PER, 0);
    }
  };
#ifdef A3D_RING_DIAG      // timing-only diagnostic build (results wrong): A3D_DBG bit 0 / 1 no A / B requests inside the loop, bit 2 no math
  const bool dma_a = !(p.dbg & 1), dma_b = !(p.dbg & 2), math = !(p.dbg & 4);
#else
  constexpr bool dma_a = true, dma_b = true, math = true;
#endif
  if (nkt > 0) {
    prep(kt_begin);
    fire_a(0);
    fire_b(0);
    prep(kt_begin + 1);
    ring_landed();
    __syncthreads();
    if (nkt > 1) { fire_a(1); fire_b(1); }
    prep(kt_begin + 2);
    read_frags(smem_raw, 0, 0);
  }
  for (int it = 0; it < nkt; ++it) {
    const unsigned char* st = smem_raw + (it & 1) * Cfg::STAGE;
    const unsigned char* stn = smem_raw + ((it + 1) & 1) * Cfg::STAGE;
    if (math) {
#pragma unroll
      for (int s = 0; s < BK / 16 - 1; ++s) {
        read_frags(st, s + 1, (s + 1) & 1);
        mfmas(s & 1);
        interleave();
      }
    }
    // tile it + 1 has landed (requested most of an iteration ago) on every wave, and no wave will read stage it & 1 again
    ring_landed();
    __syncthreads();
    if (math) {
      if (it + 1 < nkt) read_frags(stn, 0, 0);
      mfmas((BK / 16 - 1) & 1);
    }
    if (it + 2 < nkt) {
      if (dma_a) fire_a(it & 1);
      if (dma_b) fire_b(it & 1);
    }
    prep(kt_begin + it + 3);
    if constexpr (MODE == MODE_BWD_F) write_pix(kt_begin + it + 4);      // (slot of tile `it`: read three iterations ago)
  }

  // ---- epilogue: bias / activation in registers, then the tile leaves as WHOLE 16-byte row pieces.  A lane of the MFMA
  //      result holds one column: storing from there means 2-byte stores, 64 bytes of each cache line per instruction —
  //      measured (a build of this kernel with requests and math removed): 35 of conv2d_1's 95 us.  Each wave passes its
  //      32 x WN sub-tiles through its own slice of the (now idle) stage buffers: packed 4-byte LDS writes (two columns of
  //      one row per lane after a lane-pair exchange), 16-byte reads of eight (bf16) / four (fp32) adjacent columns, the
  //      ReluGrad mask of bwd-data fetched and applied in that form too. ----
  constexpr int ESZ = C16 ? 2 : 4;                 // output element size
  constexpr int EP = Cfg::WN * ESZ + 16;           // row pitch of a wave's staging rows
  constexpr int LPR = Cfg::WN * ESZ / 16;          // 16-byte pieces per row
  constexpr int LPRP = LPR <= 4 ? 4 : LPR <= 8 ? 8 : LPR <= 16 ? 16 : LPR <= 32 ? 32 : 64;      // lanes given to a row (96 columns: 12 of 16 busy)
  constexpr int RPI = 64 / LPRP;                   // rows per wave-instruction there
  static_assert(LPR <= 64 && 32 % RPI == 0, "staging rows divide over the lanes");
  static_assert((size_t)8 * 32 * EP <= Cfg::LDS_BYTES, "staging rows fit the tile buffers");
  __syncthreads();                                 // every wave is done with the stages
  float* Cout = p.C;                               // under split-K: this split's slab of raw sums, rows of N floats (C16 = false: host)
  int ldc = p.ldc;
  const bool partial = p.splitk > 1;
  if (partial) {
    Cout = p.C + (size_t)split * p.slab;
    ldc = p.N;
  }
  unsigned char* eb = smem_raw + wave * (32 * EP);
  const int er = lane / LPRP, ec = lane % LPRP;    // 16-byte phase: this lane's row within a group of RPI, its piece of the row
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + wn * Cfg::WN + b * 32 + li;
      float bias = 0.f;
      if (MODE == MODE_FWD && !partial && p.bias && col < p.N) bias = p.bias[col];
      float val[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        val[v] = acc[a][b][v];
        if (MODE == MODE_FWD && !partial) {
          val[v] += bias;
          if (p.act == EPI_RELU) val[v] = fmaxf(val[v], 0.f);
          else if (p.act == EPI_SIGMOID) val[v] = 1.f / (1.f + expf(-val[v]));
          if (p.keep) {                            // tf.layers.dropout fused (dense layers: a handful of rows)
            const int row = m0 + wm * Cfg::WM + a * 32 + (v & 3) + 8 * (v >> 2) + 4 * lh;
            if (row < p.M && col < p.N) val[v] = p.keep[(size_t)row * p.N + col] ? val[v] * p.mask_scale : 0.f;
          }
        }
      }
      if constexpr (C16) {
        // rows r (register v even) and r + 1 (v + 1) of column li: the even lane takes over its neighbour's row-r value and
        // writes columns li, li + 1 of row r; the odd lane gets the neighbour's row-(r + 1) value: columns li - 1, li
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bool odd = li & 1;
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          const float give = odd ? val[v] : val[v + 1];
          const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
          const bf16x2 pk = odd ? bf16x2{(__bf16)got, (__bf16)val[v + 1]} : bf16x2{(__bf16)val[v], (__bf16)got};
          const int row = (v & 3) + 8 * (v >> 2) + 4 * lh + (odd ? 1 : 0);
          *reinterpret_cast<bf16x2*>(eb + row * EP + (b * 32 + (li & ~1)) * 2) = pk;
        }
      } else {
#pragma unroll
        for (int v = 0; v < 16; ++v)
          *reinterpret_cast<float*>(eb + ((v & 3) + 8 * (v >> 2) + 4 * lh) * EP + (b * 32 + li) * 4) = val[v];
      }
    }
    // the same wave reads what it wrote: LDS serves a wave's accesses in order
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
      const int r = i * RPI + er;
      const int row = m0 + wm * Cfg::WM + a * 32 + r, col0 = n0 + wn * Cfg::WN + ec * (16 / ESZ);
      if (LPR != LPRP && ec >= LPR) continue;
      u32x4 q = *reinterpret_cast<const u32x4*>(eb + r * EP + ec * 16);
      if (row < p.M && col0 < p.N) {               // (N is a multiple of the piece: host)
        const size_t o = (size_t)row * ldc + col0;
        if (MODE == MODE_BWD_D && !partial && p.mask) {        // ReluGrad of the layer below: dx = 0 where its activation is not positive
          if constexpr (C16) {
            const u32x4 mk = *reinterpret_cast<const u32x4*>(reinterpret_cast<const __bf16*>(p.mask) + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const bool lo = __uint_as_float(mk[e] << 16) > 0.f, hi = __uint_as_float(mk[e] & 0xffff0000u) > 0.f;
              q[e] &= (lo ? 0x0000ffffu : 0u) | (hi ? 0xffff0000u : 0u);
            }
          } else {
            const f32x4 mk = *reinterpret_cast<const f32x4*>(p.mask + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = mk[e] > 0.f ? q[e] : 0u;
          }
        }
        if constexpr (C16) *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(Cout) + o) = q;
        else *reinterpret_cast<u32x4*>(Cout + o) = q;
      }
    }
  }
}

}  // namespace a3d
