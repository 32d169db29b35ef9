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
#include <type_traits>
#include <utility>

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
  static constexpr int TAB_ROWS = MODE == MODE_BWD_F ? 512 : BM;      // row table (BM entries) / bwd-filter: eight k-tiles' pixel tables
  static constexpr size_t TILE_AND_TABLES = (size_t)NSTAGE * STAGE + (size_t)TAB_ROWS * 16 + (size_t)MAXTAPS * 16;
  static constexpr size_t EPI_BYTES = (size_t)8 * 32 * (WN * 4 + 16);
  static constexpr size_t LDS_BYTES = TILE_AND_TABLES > EPI_BYTES ? TILE_AND_TABLES : EPI_BYTES;
  // registers: one block per CU for the big tiles (two wavefronts per SIMD, up to 256 registers each)
  static constexpr int MIN_WAVES = (TM * TN <= 2) ? 4 : 2;
};

__device__ __forceinline__ int ring_swz(int row) { return (row >> 1) & 7; }

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a constant expression inside the body
template <class F, int... I>
__device__ __forceinline__ void ring_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void ring_static_for(F&& f) {
  ring_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

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

// Which k rows of a [64 k][columns] image move a 16-byte chunk, and where to (XOR on the chunk index), so that the transposing
// read of a fragment touches every bank once.  32x32x16 fragments (a half-wave reads rows q = 0..3 at two column halves):
// the row's low two bits move a chunk by 4 (128-byte rows: bit 1 only).  16x16x32 fragments (a half-wave reads rows q and
// 8 + q, 32 bytes of each): row bits 0, 1 and 3 move a chunk PAIR by 1..7 pairs (128-byte rows: bits 1 and 3, 1..3 pairs).
template <bool M16>
__device__ __forceinline__ int ring_swz_tr(int k, int chunks_per_row) {
  if constexpr (M16)
    return chunks_per_row >= 16 ? (((k & 3) | (((k >> 3) & 1) << 2)) << 1) : ((((k >> 1) & 1) | (((k >> 3) & 1) << 1)) << 1);
  else
    return chunks_per_row >= 16 ? ((k & 3) << 2) : (((k >> 1) & 1) << 2);
}

// M16: the contraction runs on v_mfma_f32_16x16x32_bf16 (sixteen-row sub-tiles, k-steps of 32) instead of 32x32x16 — the same
// FLOPs per cycle, the same LDS bytes, the same results up to the order of the fp32 additions; the chip holds a higher
// clock on this shape in MFMA-dense loops.
template <int MODE, int BM, int BN, int WAVES_M, bool C16, bool M16 = false>
__global__ __launch_bounds__(512, (RingCfg<MODE, BM, BN, WAVES_M>::MIN_WAVES)) void igemm_ring_kernel(const IgemmParams p) {
  using Cfg = RingCfg<MODE, BM, BN, WAVES_M>;
  constexpr int BK = Cfg::BK, TM = Cfg::TM, TN = Cfg::TN;
  constexpr bool TRANSPOSED = (MODE == MODE_BWD_D);
  constexpr int SGN = TRANSPOSED ? -1 : 1;
  constexpr int ROWB = BN * 2;                    // bytes per k row of the forward filter tile
  constexpr int CPR = BN / 8;                     // its 16-byte chunks per row

#ifdef A3D_STAMPS           // diagnostic build (never shipped): cycle stamps of the loop's phases, per wave (tools/stamps_ring.py)
  unsigned long long st_entry = 0, st_beg = 0, st_end = 0, st_s0 = 0, st_s1 = 0, st_s2 = 0, st_d1 = 0, st_dw = 0, st_d2 = 0, st_rt0 = 0, st_rt1 = 0, st_tab = 0, st_fire = 0;
  A3D_STAMP(st_entry);
#endif
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
  // whole tensor, y0, x0}) lives in slot kt & 7.  Iteration it writes the table of tile it + 6 and reads (prep_head, in the
  // shadow of its first k-steps) the one of tile it + 3: barriers lie between a table's write and its read, and between
  // its read and the slot's next write.  Every wave writes the same 64 entries — identical values to identical
  // addresses — so that the write is unconditional: a wave-0-only store is a branch, and a branch inside the loop body ends
  // the scheduling region the MFMA shadows are filled from.
  auto write_pix = [&](int kt) {
    int4 e = make_pix<false>(p, kt * BK + lane);
    e.x = (e.x + e.y * pW + e.z) * pld * 2;
    if (!e.w || kt >= kt_end) e.y = -(1 << 30);
    rowtab[(kt & 7) * 64 + lane] = e;
  };
  if constexpr (MODE == MODE_BWD_F) {
#pragma unroll
    for (int t = 0; t < 6; ++t) write_pix(kt_begin + t);
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
#ifdef A3D_STAMPS
  A3D_STAMP(st_tab);
#endif

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
    const int cpos = id % CPRA, sw = ring_swz_tr<M16>(a_krow, CPRA);
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
      const int sw = ring_swz_tr<M16>(krow, CPR);
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
  // cycles parked, 34 % matrix-pipe busy): the pieces' offsets are computed in the shadow of the MFMAs of k-steps 0 .. 2,
  // the LDS-DMAs themselves are issued one by one behind the MFMAs of the last k-step.
  uint32_t aoff[Cfg::A_NI], boff[Cfg::B_NI];
#ifdef A3D_RING_DIAG      // timing-only diagnostic build (results wrong): A3D_DBG bit 0 / 1: the A / B requests inside the loop fetch nothing
  const uint32_t dbg_a = (p.dbg & 1) ? kOOB : 0u, dbg_b = (p.dbg & 2) ? kOOB : 0u;
#else
  constexpr uint32_t dbg_a = 0u, dbg_b = 0u;
#endif
  // Offsets of a k-tile's pieces, in pieces of work that fit an MFMA's shadow.  prep_head(kt) decodes this lane's k-chunk of
  // tile kt (filter tap, channel) and REQUESTS the tap's table entry (bwd-filter: the pixel entries of its A_NI rows); the
  // pieces that follow — an iteration later inside the loop — turn it into the A_NI + B_NI offsets.
  int4 tt = make_int4(0, 0, 0, 0), pe[MODE == MODE_BWD_F ? Cfg::A_NI : 1];
  int p_ch = 0, p_kt = 0;
  bool p_kvalid = false;
  auto prep_head = [&](int kt) {
    p_kt = kt;
    if constexpr (MODE == MODE_BWD_F) {
      const int4* tab = rowtab + (kt & 7) * 64;
#pragma unroll
      for (int j = 0; j < Cfg::A_NI; ++j) pe[j] = tab[a_krow + j * (8 * 64 / CPRA)];
    } else {
      const int k0 = kt * BK + kc * 8;
      const uint32_t tap = fdiv((uint32_t)k0, p.div_c);
      p_ch = k0 - (int)(tap * p.div_c.d);
      p_kvalid = (k0 < p.K) & (kt < kt_end);
      tt = taptab[p_kvalid ? tap : 0u];
    }
  };
  constexpr int NPREP = Cfg::A_NI + Cfg::B_NI + 1;
  // piece pc of the tile prep_head() was last called for; the last piece is prep_head(kt_next)
  auto prep_piece = [&](int pc, int kt_next, uint32_t (&ao)[Cfg::A_NI], uint32_t (&bo)[Cfg::B_NI]) {
    if (pc < Cfg::A_NI) {
      const int j = pc;
      if constexpr (MODE == MODE_BWD_F) {
        const int y = pe[j].y + a_dy, x = pe[j].z + a_dx;
        const bool ok = a_mvalid & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)pW);
        ao[j] = (ok ? (uint32_t)(pe[j].x + a_coloff) : kOOB) | dbg_a;
      } else {
        const int y = a_y0[j] + tt.x, x = a_x0[j] + tt.y;
        const bool ok = p_kvalid & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)pW);
        ao[j] = (ok ? (uint32_t)(a_rowoff[j] + tt.z + p_ch * 2) : kOOB) | dbg_a;
      }
    } else if (pc < Cfg::A_NI + Cfg::B_NI) {
      const int j = pc - Cfg::A_NI;
      // (b_voff = kOOB stays past the end under the additions: filter and slab offsets are below 2^31, checked on the host)
      if constexpr (Cfg::B_KC) bo[j] = (b_voff[j] + (uint32_t)tt.w + (uint32_t)p_ch * 2u) | (p_kvalid ? 0u : kOOB) | dbg_b;
      else bo[j] = (b_voff[j] + (uint32_t)p_kt * (uint32_t)(BK * p.ldb * 2)) | (p_kt < kt_end ? 0u : kOOB) | dbg_b;   // rows (bwd-filter: pixels) past the end: past the end
    } else {
      prep_head(kt_next);
    }
  };
  auto prep_all = [&](uint32_t (&ao)[Cfg::A_NI], uint32_t (&bo)[Cfg::B_NI]) {
#pragma unroll
    for (int pc = 0; pc < Cfg::A_NI + Cfg::B_NI; ++pc) prep_piece(pc, 0, ao, bo);
  };
  // request number d of a k-tile's A_NI + B_NI (this wave's share), into stage `stg`
  auto fire_one = [&](int stg, int d) {
    if (d < Cfg::A_NI) {
      ring_dma(rsA, aoff[d], lds0 + (uint32_t)(stg * Cfg::STAGE) + (uint32_t)((d * 8 + wave) * 1024));
    } else {
      const int j = d - Cfg::A_NI;
      if (Cfg::B_PIECES % 8 == 0 || j * 8 + wave < Cfg::B_PIECES)
        ring_dma(rsB, boff[j], lds0 + (uint32_t)(stg * Cfg::STAGE + Cfg::A_BYTES) + (uint32_t)((j * 8 + wave) * 1024));
    }
  };
  constexpr int NDMA = Cfg::A_NI + Cfg::B_NI;
  auto fire = [&](int stg) {
#pragma unroll
    for (int d = 0; d < NDMA; ++d) fire_one(stg, d);
  };

  // The first tile's requests leave NOW: everything between here and the loop (fragment addresses, 128 accumulator
  // registers to clear) happens while they are in flight instead of ahead of them — one block per CU, nothing else covers
  // the prologue.
  if (nkt > 0) {
    prep_head(kt_begin);
    prep_all(aoff, boff);
    fire(0);
    prep_head(kt_begin + 1);
    prep_all(aoff, boff);
  }
#ifdef A3D_STAMPS
  A3D_STAMP(st_fire);
#endif

  // ---- fragment addresses: LDS byte offsets per STAGE, loop-invariant registers (made opaque so that the compiler keeps them
  //      instead of re-deriving one from another with a VALU add ahead of every read: a read without address arithmetic
  //      can be issued in any MFMA shadow).  k-contiguous images: one register per (stage, k-step) — the k-step is an XOR
  //      inside the swizzled row — with the sub-tiles 4 KiB apart as immediates; [64 k][columns] images: one per (stage,
  //      sub-tile), the k-steps as immediates ----
  constexpr int ROWA = BM * 2;                    // bytes per pixel row of the bwd-filter A tile
  constexpr int TM16 = 2 * TM, TN16 = 2 * TN;     // M16: sixteen-row sub-tiles of the wave tile
  // 16x16x32: k-contiguous images one register per (stage, k-step of 32), the sub-tiles 2 KiB apart as immediates;
  // [64 k][columns] images ONE per stage: sub-tile t is an XOR with 32 t bytes, the k-step an immediate
  constexpr int NAF = M16 ? (Cfg::A_KC ? 2 : 1) : (Cfg::A_KC ? 4 : TM), NBF = M16 ? (Cfg::B_KC ? 2 : 1) : (Cfg::B_KC ? 4 : TN);
  int a_fr[2][NAF], b_fr[2][NBF];
  {
    const int g = lane >> 4, l16 = lane & 15, q = l16 >> 2, pp = l16 & 3, h = g >> 1, cb = g & 1;
#pragma unroll
    for (int stg = 0; stg < 2; ++stg) {
#pragma unroll
      for (int i = 0; i < NAF; ++i) {
        if constexpr (M16 && Cfg::A_KC) {
          const int row = wm * Cfg::WM + l16;       // (+ 16 a: the same swizzle); this lane's k-chunk of k-step i: 4 i + g
          a_fr[stg][i] = stg * Cfg::STAGE + row * 128 + (((4 * i + g) ^ ring_swz(row)) << 4);
        } else if constexpr (M16) {
          const int krow = 8 * g + q, c = ((wm * Cfg::WM) >> 3) + (pp >> 1);
          a_fr[stg][i] = stg * Cfg::STAGE + krow * ROWA + ((c ^ ring_swz_tr<true>(krow, CPRA)) << 4) + (pp & 1) * 8;   // k-step s: + 32 s ROWA; sub-tile t: ^ 32 t
        } else if constexpr (Cfg::A_KC) {
          const int row = wm * Cfg::WM + li;        // (+ 32 a: the same swizzle)
          a_fr[stg][i] = stg * Cfg::STAGE + ((row * 128 + ((lh ^ ring_swz(row)) << 4)) ^ (i << 5));
        } else {
          const int c = ((wm * Cfg::WM + i * 32) >> 3) + 2 * cb + (pp >> 1);
          a_fr[stg][i] = stg * Cfg::STAGE + (8 * h + q) * ROWA + ((c ^ (q << 2)) << 4) + (pp & 1) * 8;      // k-step s: + 16 s ROWA
        }
        asm volatile("" : "+v"(a_fr[stg][i]));
      }
#pragma unroll
      for (int i = 0; i < NBF; ++i) {
        if constexpr (M16 && Cfg::B_KC) {
          const int row = wn * Cfg::WN + l16;
          b_fr[stg][i] = stg * Cfg::STAGE + Cfg::A_BYTES + row * 128 + (((4 * i + g) ^ ring_swz(row)) << 4);
        } else if constexpr (M16) {
          const int krow = 8 * g + q, c = ((wn * Cfg::WN) >> 3) + (pp >> 1);
          b_fr[stg][i] = stg * Cfg::STAGE + Cfg::A_BYTES + krow * ROWB + ((c ^ ring_swz_tr<true>(krow, CPR)) << 4) + (pp & 1) * 8;
        } else if constexpr (Cfg::B_KC) {
          const int row = wn * Cfg::WN + li;
          b_fr[stg][i] = stg * Cfg::STAGE + Cfg::A_BYTES + ((row * 128 + ((lh ^ ring_swz(row)) << 4)) ^ (i << 5));
        } else {
          const int c = ((wn * Cfg::WN + i * 32) >> 3) + 2 * cb + (pp >> 1);
          const int sw = CPR >= 16 ? (q << 2) : ((q >> 1) << 2);      // of rows 16 s + 8 h + q and + 4: the same
          b_fr[stg][i] = stg * Cfg::STAGE + Cfg::A_BYTES + (8 * h + q) * ROWB + ((c ^ sw) << 4) + (pp & 1) * 8;     // k-step s: + 16 s ROWB
        }
        asm volatile("" : "+v"(b_fr[stg][i]));
      }
    }
  }

  // accumulators: 32x32 sub-tiles of sixteen registers, or (M16) 16x16 sub-tiles of four — the same count either way
  f32x16 acc[M16 ? 1 : TM][M16 ? 1 : TN];
  f32x4 acc16[M16 ? TM16 : 1][M16 ? TN16 : 1];
  if constexpr (M16) {
#pragma unroll
    for (int a = 0; a < TM16; ++a)
#pragma unroll
      for (int b = 0; b < TN16; ++b)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc16[a][b][v] = 0.f;
  }
  if constexpr (!M16) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
  }

  // ---- main loop, rotated: the ONE barrier of an iteration sits BEFORE its last k-step.  At that point the wave holds the
  //      last k-step's fragments in registers (nothing of stage it & 1 will be read again) and tile it + 1 has landed, so
  //      behind the barrier the first fragments of tile it + 1 are read and the requests of tile it + 2 (into stage it & 1)
  //      are issued IN THE SHADOW of the last k-step's MFMAs.  With the barrier at the top of the iteration the first
  //      fragment reads, the requests' issue (60 - 180 cycles apiece) and the offsets' arithmetic all ran with the matrix
  //      pipe idle on all eight waves at once: a build without requests ran as long as (empty loop) + (MFMA time). ----
  if constexpr (M16) {
    // ---- 16x16x32: the k-tile is a sequence of NF = 2 TN16 B fragments (k-step 0's, then k-step 1's), each multiplied
    //      into the TM16 accumulators of its column by one MFMA per A fragment: NSLOT = NF TM16 MFMA slots of 16 cycles.
    //      B fragments live in a window of four registers sets, fragment f + 3 requested behind the first MFMA of fragment f
    //      (past the end of the tile: the next tile's, behind the barrier); A fragments of k-step 1 are requested during
    //      k-step 0, the next tile's k-step 0 behind the barrier.  The barrier follows the slot in which the tile's last B
    //      fragment is requested: sixteen MFMAs' worth of operands are then in registers (or in flight), and in their
    //      shadow go the requests of tile it + 2 and the first reads of tile it + 1, as in the 32x32x16 form. ----
    constexpr int NF = 2 * TN16, NB = 4, AHEAD = NB - 1, NSLOT = NF * TM16, TB = (NF - NB) * TM16, NPOST = NSLOT - TB - 1;
    static_assert(TN16 >= 4 && TM16 >= 2 && NF % NB == 0, "16x16x32 form: wave tiles of at least 32 x 64");
    static_assert(NPOST >= TM16 && TB >= 4, "room behind and ahead of the barrier");
    bf16x8 a16[2][TM16], bw[NB];
    auto read_a = [&](int stg, int s, int a) {
      if constexpr (Cfg::A_KC) {
        a16[s & 1][a] = *reinterpret_cast<const bf16x8*>(smem_raw + a_fr[stg][s] + a * 2048);
      } else {
        const __bf16* q0 = reinterpret_cast<const __bf16*>(smem_raw + (a_fr[stg][0] ^ (a << 5)) + s * 32 * ROWA);
        const bf16x4 lo4 = lds_read_tr(q0), hi4 = lds_read_tr(q0 + 2 * ROWA);
#pragma unroll
        for (int e = 0; e < 4; ++e) { a16[s & 1][a][e] = lo4[e]; a16[s & 1][a][4 + e] = hi4[e]; }
      }
    };
    auto read_b = [&](int stg, int f) {
      const int s = f / TN16, b = f % TN16;
      if constexpr (Cfg::B_KC) {
        bw[f % NB] = *reinterpret_cast<const bf16x8*>(smem_raw + b_fr[stg][s] + b * 2048);
      } else {
        const __bf16* q0 = reinterpret_cast<const __bf16*>(smem_raw + (b_fr[stg][0] ^ (b << 5)) + s * 32 * ROWB);
        const bf16x4 lo4 = lds_read_tr(q0), hi4 = lds_read_tr(q0 + 2 * ROWB);
#pragma unroll
        for (int e = 0; e < 4; ++e) { bw[f % NB][e] = lo4[e]; bw[f % NB][4 + e] = hi4[e]; }
      }
    };
    if (nkt > 0) {
      ring_landed();
      __syncthreads();
      if (nkt > 1) fire(1);
      prep_head(kt_begin + 2);
#pragma unroll
      for (int a = 0; a < TM16; ++a) read_a(0, 0, a);
#pragma unroll
      for (int f = 0; f < AHEAD; ++f) read_b(0, f);
    }
    auto body16 = [&](int it, auto steady) {
      constexpr bool STEADY = decltype(steady)::value;
      const int par = it & 1;
      const bool rd = STEADY || it + 1 < nkt, rq = STEADY || it + 2 < nkt;
      if constexpr (MODE == MODE_BWD_F) write_pix(kt_begin + it + 6);
      ring_static_for<NSLOT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int f = t / TM16, a = t % TM16, s = f / TN16, b = f % TN16;
        acc16[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a16[s & 1][a], bw[f % NB], acc16[a][b], 0, 0, 0);
        if (a == 0) {                                           // the window moves on
          if (f + AHEAD < NF) read_b(0, f + AHEAD);
          else if (rd) read_b(1, f + AHEAD - NF);
        }
#pragma unroll
        for (int j = 0; j < TM16; ++j) {
          if (t == (TM16 * j + 1 < TB ? TM16 * j + 1 : TB)) read_a(0, 1, j);        // k-step 1 of this tile
          if (t == TB + 1 + j && rd) read_a(1, 0, j);                               // k-step 0 of the next one
        }
        if (t < TB) {
#pragma unroll
          for (int pc = t * NPREP / TB; pc < (t + 1) * NPREP / TB; ++pc) prep_piece(pc, kt_begin + it + 3, aoff, boff);
        } else if (t > TB && rq) {
#pragma unroll
          for (int d = (t - TB - 1) * NDMA / NPOST; d < (t - TB) * NDMA / NPOST; ++d) fire_one(par, d);
        }
        if (t == TB) {
          ring_landed();
          __syncthreads();
        } else if (STEADY) {
          __builtin_amdgcn_sched_barrier(0);
        }
      });
#pragma unroll
      for (int i = 0; i < NAF; ++i) { const int t = a_fr[0][i]; a_fr[0][i] = a_fr[1][i]; a_fr[1][i] = t; }
#pragma unroll
      for (int i = 0; i < NBF; ++i) { const int t = b_fr[0][i]; b_fr[0][i] = b_fr[1][i]; b_fr[1][i] = t; }
    };
    int it = 0;
    for (; it + 2 < nkt; ++it) body16(it, std::true_type{});
    for (; it < nkt; ++it) body16(it, std::false_type{});
  } else {
  bf16x8 af[2][TM], bf[2][TN];
  // fragment number u of k-step s of this iteration's tile (stg 0) or the next one's (stg 1): A sub-tile u (u < TM) or B sub-tile u - TM
  auto read_unit = [&](int stg, int s, int buf, int u) {
    if (u < TM) {
      const int a = u;
      if constexpr (Cfg::A_KC) {
        af[buf][a] = *reinterpret_cast<const bf16x8*>(smem_raw + a_fr[stg][s] + a * 4096);
      } else {
        const __bf16* q0 = reinterpret_cast<const __bf16*>(smem_raw + a_fr[stg][a] + s * 16 * ROWA);
        const bf16x4 lo4 = lds_read_tr(q0), hi4 = lds_read_tr(q0 + 2 * ROWA);
#pragma unroll
        for (int e = 0; e < 4; ++e) { af[buf][a][e] = lo4[e]; af[buf][a][4 + e] = hi4[e]; }
      }
    } else {
      const int b = u - TM;
      if constexpr (Cfg::B_KC) {
        bf[buf][b] = *reinterpret_cast<const bf16x8*>(smem_raw + b_fr[stg][s] + b * 4096);
      } else {
        const __bf16* q0 = reinterpret_cast<const __bf16*>(smem_raw + b_fr[stg][b] + s * 16 * ROWB);
        const bf16x4 lo4 = lds_read_tr(q0), hi4 = lds_read_tr(q0 + 2 * ROWB);      // rows k and k + 4 (ROWB / 2 elements each)
#pragma unroll
        for (int e = 0; e < 4; ++e) { bf[buf][b][e] = lo4[e]; bf[buf][b][4 + e] = hi4[e]; }
      }
    }
  };
  constexpr int NMFMA = TM * TN, NU = TM + TN, NS1 = 3 * NMFMA;
  // The CU's address unit takes 16 cycles per request (64 lanes x 16 bytes), 64 requests per k-tile of a 256 x 256 block:
  // 1 024 of the 2 048 cycles the matrix pipe needs.  All of them behind the barrier — the only place the two-stage ring
  // allows a whole tile's requests — made the last k-step (512 MFMA cycles per SIMD) take 1 050 (in-kernel stamps).  So only
  // NEARLY requests per wave go there (what their shadow hides: half as many as MFMA slots); the other NLATE follow in the
  // first slots of the NEXT iteration, still an iteration's k-steps 0 .. 2 ahead of the barrier that waits for them.
#ifdef A3D_RING_NEARLY
  constexpr int NEARLY_ = A3D_RING_NEARLY;
#else
  constexpr int NEARLY_ = NMFMA / 2;
#endif
  constexpr int NEARLY = NEARLY_ < 1 ? 1 : NEARLY_ > NDMA ? NDMA : (NDMA - NEARLY_ >= NS1 ? NDMA - NS1 + 1 : NEARLY_), NLATE = NDMA - NEARLY;
  static_assert(NLATE < NS1, "a slot is left for the offsets");
#ifdef A3D_RING_DIAG      // A3D_DBG bit 2: no fragment reads, no MFMAs; bit 3: no barrier; bit 4: no wait for the requests; bit 5: no fragment reads
  const bool math = !(p.dbg & 4), dbar = !(p.dbg & 8), dland = !(p.dbg & 16), dreads = !(p.dbg & 32);
#else
  constexpr bool math = true, dbar = true, dland = true, dreads = true;
#endif
  if (nkt > 0) {
    ring_landed();
    __syncthreads();
    if (nkt > 1) {
#pragma unroll
      for (int d = 0; d < NEARLY; ++d) fire_one(1, d);       // (the rest of tile 1: first slots of iteration 0)
    }
    prep_head(kt_begin + 2);
#pragma unroll
    for (int u = 0; u < NU; ++u) read_unit(0, 0, 0, u);
  }
  // One iteration.  The instruction order is pinned slot by slot: a slot is ONE MFMA and what rides
  // in its shadow; the compiler may not move anything across a slot's end.
  //   k-steps 0 .. 2   the slot's share of the next k-step's fragment reads; in the first NLATE slots the remaining requests
  //                    of tile it + 1, in the others the offset arithmetic of tile it + 2 (whose tap entry was read an
  //                    iteration ago; the last piece requests the entry of tile it + 3)
  //   barrier          tile it + 1 has landed on every wave, no wave reads stage it & 1 again
  //   last k-step      the slot's share of the first fragment reads of tile it + 1 and of the first NEARLY requests of tile it + 2
  // STEADY (tiles it + 1 and it + 2 exist): no branch between the top and the bottom of the iteration.
  auto body = [&](int it, auto steady) {
    constexpr bool STEADY = decltype(steady)::value;
    const int par = it & 1;
#ifdef A3D_STAMPS
    A3D_STAMP(st_s0);
    if (it > 0) st_d2 += st_s0 - st_s2;
#endif
    if constexpr (MODE == MODE_BWD_F) write_pix(kt_begin + it + 6);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
#pragma unroll
      for (int i = 0; i < NMFMA; ++i) {
        const int a = i / TN, b = i % TN, q = s * NMFMA + i;
        if (math) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][a], bf[s & 1][b], acc[a][b], 0, 0, 0);
#pragma unroll
          for (int u = i * NU / NMFMA; dreads && u < (i + 1) * NU / NMFMA; ++u) read_unit(0, s + 1, (s + 1) & 1, u);
        }
        if (q < NLATE) {
          if (STEADY || it + 1 < nkt) fire_one(par ^ 1, NEARLY + q);
        } else {
#pragma unroll
          for (int pc = (q - NLATE) * NPREP / (NS1 - NLATE); pc < (q - NLATE + 1) * NPREP / (NS1 - NLATE); ++pc)
            prep_piece(pc, kt_begin + it + 3, aoff, boff);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef A3D_STAMPS
    A3D_STAMP(st_s1);
    st_d1 += st_s1 - st_s0;
#endif
    if (dland) ring_landed();
    if (dbar) __syncthreads();
#ifdef A3D_STAMPS
    A3D_STAMP(st_s2);
    st_dw += st_s2 - st_s1;
#endif
    const bool rd = (STEADY || it + 1 < nkt) && dreads, rq = STEADY || it + 2 < nkt;
    if (math) {
#pragma unroll
      for (int i = 0; i < NMFMA; ++i) {
        const int a = i / TN, b = i % TN;
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][a], bf[1][b], acc[a][b], 0, 0, 0);
        if (rd) {
#pragma unroll
          for (int u = i * NU / NMFMA; u < (i + 1) * NU / NMFMA; ++u) read_unit(1, 0, 0, u);
        }
        if (rq) {
#pragma unroll
          for (int d = i * NEARLY / NMFMA; d < (i + 1) * NEARLY / NMFMA; ++d) fire_one(par, d);
        }
        if constexpr (STEADY) __builtin_amdgcn_sched_barrier(0);
      }
    } else if (rq) {
#pragma unroll
      for (int d = 0; d < NEARLY; ++d) fire_one(par, d);
    }
    // the stages trade places: the address registers of "this tile's stage" (index 0) and "the next tile's" (index 1)
#pragma unroll
    for (int i = 0; i < NAF; ++i) { const int t = a_fr[0][i]; a_fr[0][i] = a_fr[1][i]; a_fr[1][i] = t; }
#pragma unroll
    for (int i = 0; i < NBF; ++i) { const int t = b_fr[0][i]; b_fr[0][i] = b_fr[1][i]; b_fr[1][i] = t; }
  };
#ifdef A3D_STAMPS
  A3D_STAMP(st_beg);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt0)::"memory");
#endif
  int it = 0;
  for (; it + 2 < nkt; ++it) body(it, std::true_type{});
  for (; it < nkt; ++it) body(it, std::false_type{});
#ifdef A3D_STAMPS
  A3D_STAMP(st_end);
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt1)::"memory");
  st_d2 += st_end - st_s2;
#endif

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
  // the fused 2x2 max pool (forward only, never split: host) leaves through the same staging rows, eight pooled rows per group
  constexpr bool POOLABLE = MODE == MODE_FWD && !M16 && 8 % RPI == 0 && Cfg::WN % 16 == 0 && 8 * (Cfg::WN / 16) <= 64;
  const bool pooled = POOLABLE && p.pool != 0;
  unsigned char* eb = smem_raw + wave * (32 * EP);
  const int er = lane / LPRP, ec = lane % LPRP;    // 16-byte phase: this lane's row within a group of RPI, its piece of the row
  // element e (0 .. 15) of a lane's share of a 32 x 32 sub-tile: its row and column inside the sub-tile.  32x32x16: column li,
  // rows (e & 3) + 8 (e >> 2) + 4 lh.  16x16x32: four 16 x 16 accumulators (a2, b2), e = 8 a2 + 4 b2 + v: row 16 a2 + 4 (lane >> 4) + v,
  // column 16 b2 + (lane & 15).  Either way e and e + 1 (e even) are rows r, r + 1 of one column, and lane ^ 1 holds the
  // neighbouring column of the same rows.
  auto erow = [&](int e) { return M16 ? 16 * (e >> 3) + 4 * (lane >> 4) + (e & 3) : (e & 3) + 8 * (e >> 2) + 4 * lh; };
  auto ecol = [&](int e) { return M16 ? 16 * ((e >> 2) & 1) + (lane & 15) : li; };
  // this lane's bias values, all requested before the first sub-tile: loaded where they are used, each was a dependent
  // global load (an L2 round trip) at the head of its sub-tile — eight in a row for a 64 x 128 wave tile
  float bias_all[TN][2];
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      const int col = n0 + wn * Cfg::WN + b * 32 + ecol(4 * h2);
      bias_all[b][h2] = (MODE == MODE_FWD && !partial && p.bias && (M16 || h2 == 0) && col < p.N) ? p.bias[col] : 0.f;
    }
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const float bias[2] = {bias_all[b][0], bias_all[b][1]};
      // every uniform choice (activation, dropout) branches once per 32 x 32 sub-tile, not once per value: with the branches
      // inside the value loop the sub-tile's code was ~1 200 instructions of which the ReLU path executes 100, jumping
      // over the sigmoid's and the dropout's sixteen times — the epilogue of the whole kernel 50 KB of code walked in
      // hops, each hop an instruction-cache miss on every CU at once (in-kernel stamps: 25.7 k cycles for conv2d_1's
      // forward at batch 64, a fifth of its loop)
      float val[16];
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        if constexpr (M16) val[v] = acc16[2 * a + (v >> 3)][2 * b + ((v >> 2) & 1)][v & 3];
        else val[v] = acc[a][b][v];
      }
      if (MODE == MODE_BWD_D && !partial && p.mask) {       // (dropout's gradient factor; the mask itself: 16-byte phase)
#pragma unroll
        for (int v = 0; v < 16; ++v) val[v] *= p.mask_scale;
      }
      if (MODE == MODE_FWD && !partial) {
#pragma unroll
        for (int v = 0; v < 16; ++v) val[v] += bias[M16 ? (v >> 2) & 1 : 0];
        if (p.act == EPI_RELU) {
#pragma unroll
          for (int v = 0; v < 16; ++v) val[v] = fmaxf(val[v], 0.f);
        } else if (__builtin_expect(p.act == EPI_SIGMOID, 0)) {
#pragma unroll
          for (int v = 0; v < 16; ++v) val[v] = 1.f / (1.f + expf(-val[v]));
        }
        if (__builtin_expect(p.keep != nullptr, 0)) {      // tf.layers.dropout fused (dense layers: a handful of rows; out of line)
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int row = m0 + wm * Cfg::WM + a * 32 + erow(v), col = n0 + wn * Cfg::WN + b * 32 + ecol(v);
            if (row < p.M && col < p.N) val[v] = p.keep[(size_t)row * p.N + col] ? val[v] * p.mask_scale : 0.f;
          }
        }
      }
      if constexpr (POOLABLE) {
        if (pooled) {
          // forward with the 2x2 max pool fused (rows enumerated window by window, make_pix): registers 4 j .. 4 j + 3 are the
          // four conv outputs of window 2 j + lh of this 32-row group.  The values a separate conv would have stored (rounded
          // to bf16 if its output tensor is), compared the way MaxPool / MaxPoolGrad scan them: first maximum wins.
          float pv[4];
          int pa[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float v4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v4[i] = C16 ? (float)(__bf16)val[4 * j + i] : val[4 * j + i];
            pv[j] = v4[0];
            pa[j] = 0;
#pragma unroll
            for (int i = 1; i < 4; ++i)
              if (v4[i] > pv[j]) { pv[j] = v4[i]; pa[j] = i; }
          }
          unsigned char* eb8 = eb + 8 * EP;          // the argmax bytes of the eight windows: rows of WN + 16 bytes
          if constexpr (C16) {
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            const bool odd = lane & 1;
#pragma unroll
            for (int jp = 0; jp < 4; jp += 2) {
              const float give = odd ? pv[jp] : pv[jp + 1];
              const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
              const bf16x2 pk = odd ? bf16x2{(__bf16)got, (__bf16)pv[jp + 1]} : bf16x2{(__bf16)pv[jp], (__bf16)got};
              const int wrow = 2 * (jp + (odd ? 1 : 0)) + lh;
              *reinterpret_cast<bf16x2*>(eb + wrow * EP + (b * 32 + (li & ~1)) * 2) = pk;
            }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<float*>(eb + (2 * j + lh) * EP + (b * 32 + li) * 4) = pv[j];
          }
          if (p.argmax) {
#pragma unroll
            for (int j = 0; j < 4; ++j) eb8[(2 * j + lh) * (Cfg::WN + 16) + b * 32 + li] = (unsigned char)pa[j];
          }
          continue;
        }
      }
      if constexpr (C16) {
        // rows r (register v even) and r + 1 (v + 1) of one column: the even lane takes over its neighbour's row-r value and
        // writes columns c, c + 1 of row r; the odd lane gets the neighbour's row-(r + 1) value: columns c - 1, c
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        const bool odd = lane & 1;
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          const float give = odd ? val[v] : val[v + 1];
          const float got = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, give), 0xB1, 0xf, 0xf, true));
          const bf16x2 pk = odd ? bf16x2{(__bf16)got, (__bf16)val[v + 1]} : bf16x2{(__bf16)val[v], (__bf16)got};
          const int row = erow(v) + (odd ? 1 : 0);
          *reinterpret_cast<bf16x2*>(eb + row * EP + (b * 32 + (ecol(v) & ~1)) * 2) = pk;
        }
      } else {
#pragma unroll
        for (int v = 0; v < 16; ++v)
          *reinterpret_cast<float*>(eb + erow(v) * EP + (b * 32 + ecol(v)) * 4) = val[v];
      }
    }
    // the same wave reads what it wrote: LDS serves a wave's accesses in order
    if constexpr (POOLABLE) {
      if (pooled) {
        const int prow0 = (m0 + wm * Cfg::WM + a * 32) >> 2;      // first pooled pixel of this group of eight windows
#pragma unroll
        for (int i = 0; i < 8 / RPI; ++i) {
          const int r = i * RPI + er;
          const int col0 = n0 + wn * Cfg::WN + ec * (16 / ESZ);
          if (LPR != LPRP && ec >= LPR) continue;
          const u32x4 q = *reinterpret_cast<const u32x4*>(eb + r * EP + ec * 16);
          if (4 * (prow0 + r) < p.M && col0 < p.N) {
            const size_t o = (size_t)(prow0 + r) * ldc + col0;
            if constexpr (C16) *reinterpret_cast<u32x4*>(reinterpret_cast<__bf16*>(Cout) + o) = q;
            else *reinterpret_cast<u32x4*>(Cout + o) = q;
          }
        }
        if (p.argmax) {                                            // sixteen window positions per lane, whole 16-byte pieces (N % 16 == 0: host)
          constexpr int APR = Cfg::WN / 16;                       // pieces per row
          const int r = lane / APR, c16 = lane % APR, col0 = n0 + wn * Cfg::WN + c16 * 16;
          if (lane < 8 * APR) {
            const u32x4 q = *reinterpret_cast<const u32x4*>(eb + 8 * EP + r * (Cfg::WN + 16) + c16 * 16);
            if (4 * (prow0 + r) < p.M && col0 < p.N) *reinterpret_cast<u32x4*>(p.argmax + (size_t)(prow0 + r) * p.N + col0) = q;
          }
        }
        continue;
      }
    }
    if (MODE == MODE_BWD_D && !partial && p.mask && (unsigned)ldc < (1u << 20)) {
      // ReluGrad of the layer below: dx = 0 where its activation is not positive.  The mask pieces of the whole 32-row group
      // are requested first and the stores follow them: with the load inside the store loop every store waited for its
      // mask's round trip AND for the store before it (vmcnt counts both; seen in the ISA as one s_waitcnt vmcnt(0) per
      // store).  Descriptors based at the tile's first row: rows >= M and invalid columns lie past their ends (igemm.h).
      int rows_left = p.M - m0;
      rows_left = rows_left < 0 ? 0 : (rows_left > BM ? BM : rows_left);
      const __amdgpu_buffer_rsrc_t rsM = epi_rsrc(p.mask, (size_t)m0 * ldc * ESZ, (uint32_t)(rows_left * ldc) * ESZ);
      const __amdgpu_buffer_rsrc_t rsC = epi_rsrc(Cout, (size_t)m0 * ldc * ESZ, (uint32_t)(rows_left * ldc) * ESZ);
      const int col0 = n0 + wn * Cfg::WN + ec * (16 / ESZ);
      const bool lane_ok = (LPR == LPRP || ec < LPR) && col0 < p.N;
      const uint32_t v0 = lane_ok ? (uint32_t)((wm * Cfg::WM + a * 32 + er) * ldc + col0) * ESZ : kOOB;
      u32x4 mk[32 / RPI];
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i)
        mk[i] = __builtin_amdgcn_raw_buffer_load_b128(rsM, (int)(v0 + (uint32_t)(i * RPI * ldc) * ESZ), 0, 0);
#pragma unroll
      for (int i = 0; i < 32 / RPI; ++i) {
        const int r = i * RPI + er;
        u32x4 q = *reinterpret_cast<const u32x4*>(eb + r * EP + (LPR != LPRP && ec >= LPR ? 0 : ec) * 16);
        if constexpr (C16) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool lo = __uint_as_float(mk[i][e] << 16) > 0.f, hi = __uint_as_float(mk[i][e] & 0xffff0000u) > 0.f;
            q[e] &= (lo ? 0x0000ffffu : 0u) | (hi ? 0xffff0000u : 0u);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = __uint_as_float(mk[i][e]) > 0.f ? q[e] : 0u;
        }
        __builtin_amdgcn_raw_buffer_store_b128(q, rsC, (int)(v0 + (uint32_t)(i * RPI * ldc) * ESZ), 0, 0);
      }
      continue;
    }
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
#ifdef A3D_STAMPS
  {
    unsigned long long st_exit = 0;
    A3D_STAMP(st_exit);
    if (p.stamps && lane == 0) {
      unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 16;
      o[0] = st_d1; o[1] = st_dw; o[2] = st_d2; o[3] = st_end - st_beg; o[4] = st_rt1 - st_rt0; o[5] = (unsigned long long)nkt;
      o[6] = st_beg - st_entry; o[7] = st_exit - st_end; o[8] = st_entry; o[9] = st_exit; o[10] = st_tab - st_entry; o[11] = st_fire - st_tab;
    }
  }
#endif
}

}  // namespace a3d
