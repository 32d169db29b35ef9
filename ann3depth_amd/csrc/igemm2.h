// igemm2.h — second-generation fp32 implicit GEMM (round 6): the contractions of igemm.h on v_mfma_f32_32x32x2_f32 with
//   * 64 x 64 per wave (four accumulators), four waves = a 128 x 128 block tile, two blocks per CU;
//   * both operand tiles written into LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no VGPR round trip, no ds_write, the
//     waves' instruction streams carry the MFMAs, the fragment reads and a handful of address instructions per k-tile;
//   * the order of a k-tile's instructions pinned (sched_barrier fences): fragments of chunk u + 1 are read among the MFMAs
//     of chunk u, the DMA requests of tile t + 1 go out between the MFMA quads of tile t's first two chunks, ONE barrier per
//     k-tile (64 MFMAs per wave) with two MFMA quads held back behind it to cover the first fragment reads of the next tile.
// tools/micro/gemm{2,3,4}.hip are the plain-GEMM prototypes this loop was measured on (profiles/r06_micro_gemm*.txt): 0.96 of
// the matrix pipe's cycles inside the loop; what a launch then reaches is set by the clock the chip holds under the load.
//
// LDS images (a DMA wave-instruction writes 1 KiB linearly, lane l at base + 16 l — no padding inside a piece):
//   KC ("k-contiguous": im2col rows in forward / bwd-data, the filter in bwd-data): [128 rows][32 k] = 128-byte rows, 16-byte
//       chunk c of row r at chunk position c ^ ((r >> 1) & 7) (applied to the SOURCE address of the DMA and again by the
//       ds_read_b128 fragment read: one read = four k-pairs of a 32-row group);
//   MC ("as stored": the filter [k][n] in forward, x and dz [pixel][channel] in bwd-filter): [32 k][128 columns]; a piece =
//       two k rows, pieces 1088 bytes apart so that the rows of the two lane halves (k and k + 4: two pieces on) fall into
//       opposite halves of the 64 banks; fragments by ds_read2_b32 (one read = columns li and 32 + li of one k).
// No transposes anywhere: each mode reads its operands the way they lie in memory.
//
// Takes what the planner gives it only when (host, gen2_applicable): float32 tensors, 16-byte vectorisable operands, and for
// forward / bwd-data a gathered channel count that is a multiple of 32 with K = taps x channels (a k-tile lies inside one
// filter tap: the tap is decoded with scalar instructions), stride-1 bwd-data.  Epilogues, stream-K shares, slabs and
// fix-ups are igemm.h's (the accumulator layout is that of its 4-wave 128 x 128 configuration).
#pragma once
#include "igemm.h"

namespace a3d {

constexpr int G2_BM = 128, G2_BN = 128, G2_BK = 32, G2_NT = 256;
constexpr int G2_MC_PIECE = 1088;                 // bytes between the two-row pieces of an MC tile

template <int MODE>
struct Gen2Cfg {
  static constexpr bool A_KC = MODE != MODE_BWD_F;
  static constexpr bool B_KC = MODE == MODE_BWD_D;
  static constexpr int A_BYTES = A_KC ? G2_BM * 128 : 16 * G2_MC_PIECE;
  static constexpr int B_BYTES = B_KC ? G2_BN * 128 : 16 * G2_MC_PIECE;
  static constexpr int A_STEP = A_KC ? 1024 : G2_MC_PIECE, B_STEP = B_KC ? 1024 : G2_MC_PIECE;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  // row table: forward / bwd-data one entry per tile row; bwd-filter three k-tiles' pixel tables (32 entries each)
  static constexpr int TAB_BYTES = MODE == MODE_BWD_F ? 3 * G2_BK * 16 : G2_BM * 16;
  static constexpr size_t LDS_BYTES = (size_t)2 * STAGE + TAB_BYTES;
};

typedef uint32_t g2_u32x4 __attribute__((ext_vector_type(4)));
// buffer descriptor in four scalar registers (every word provably wave-uniform: it is an "s" operand of g2_dma)
__device__ __forceinline__ g2_u32x4 g2_rsrc(const void* base, unsigned long long bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  g2_u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes > 0x7fffffffull ? 0x7fffffffu : (uint32_t)bytes);
  r[3] = 0x00020000u;
  return r;
}
// One LDS-DMA wave-instruction: lane l's 16 bytes at byte offset voff (+ the scalar soff) of the buffer go to LDS byte
// address lds_addr + 16 l; zeros when voff is past the descriptor's end (kOOB).  Inline assembly on purpose (igemm_ring.h):
// the compiler would order every LDS read behind a DMA it knows of with s_waitcnt vmcnt(0).  M0 is saved and restored; the
// s_nop covers M0 write -> LDS-DMA and SGPR write -> VMEM read.
__device__ __forceinline__ void g2_dma(g2_u32x4 rs, uint32_t voff, uint32_t soff, uint32_t lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
               : "memory");
}
#define G2_FENCE() __builtin_amdgcn_sched_barrier(0)

typedef float g2_f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(G2_NT, 2) void igemm2_kernel(const IgemmParams p) {
  using Cfg = Gen2Cfg<MODE>;
  constexpr int BM = G2_BM, BN = G2_BN, BK = G2_BK;
  constexpr bool TRANSPOSED = (MODE == MODE_BWD_D);
  constexpr int SGN = TRANSPOSED ? -1 : 1;
  constexpr bool A_KC = Cfg::A_KC, B_KC = Cfg::B_KC;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int4* pixtab = reinterpret_cast<int4*>(smem_raw + 2 * Cfg::STAGE);
  const uint32_t lds0 = (uint32_t)(size_t)smem_raw;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  // ---- block -> shares of the (tile, k-tile) iteration space: igemm_body's ----
  const uint32_t nwg = gridDim.x;
  uint32_t bid = blockIdx.x;
  {
    uint32_t q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_mn = p.tiles_m * p.tiles_n;
  const int nk_total = (p.K + BK - 1) / BK;
  const SkSpace sp = sk_space(p.streamk, bid, nwg, (uint32_t)tiles_mn, (uint32_t)nk_total);
  uint32_t sk_cur = p.streamk ? sk_first(sp.j, sp.blocks, sp.total) : 0u;
  const uint32_t sk_end = p.streamk ? sk_first(sp.j + 1, sp.blocks, sp.total) : 1u;
  const int pW = p.W, pld = p.ld;
#ifdef A3D_STAMPS          // diagnostic build (never shipped): tools/stamps_layer.py, the slots of igemm_body's stamps
  unsigned long long st_entry = 0, st_beg = 0, st_end = 0, st_wait = 0, st_bar = 0, st_s0 = 0, st_s1 = 0, st_s2 = 0, st_pro = 0, st_loop = 0,
                     st_nkt = 0, st_rt0 = 0, st_rt1 = 0, st_rt = 0, st_rtfirst = 0;
  A3D_STAMP(st_entry);
#endif

  for (int seg = 0; sk_cur < sk_end; ++seg) {
    int split = 0, tmn, kt_begin, kt_end;
    if (p.streamk) {
      tmn = p.streamk == 1 ? (int)fdiv(sk_cur, p.div_nk) : (int)(sk_cur / sp.nk);
      const uint32_t kl = sk_cur - (uint32_t)tmn * sp.nk;
      const uint32_t n = min(sk_end - sk_cur, sp.nk - kl);
      kt_begin = (int)(sp.k0 + kl);
      kt_end = kt_begin + (int)n;
      sk_cur += n;
    } else {
      split = bid / tiles_mn;
      tmn = bid - split * tiles_mn;
      kt_begin = split * p.ktiles_per_split;
      kt_end = kt_begin + p.ktiles_per_split;
      if (kt_end > nk_total) kt_end = nk_total;
      sk_cur = sk_end;
    }
    const int tile_m = tmn / p.tiles_n, tile_n = tmn - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nkt = kt_end - kt_begin;
    if (seg > 0) __syncthreads();                  // the previous share's tiles and tables are dead from here on

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
    // BiasAddGrad (bwd-filter): the wm = 0 waves of the first M-tile multiply a row of ones into every dz fragment they read —
    // all 32 rows of accb[b] then hold the column sums of the wave's 64 columns over this share's pixels
    const bool do_bias = (MODE == MODE_BWD_F) && p.dbias != nullptr && tile_m == 0 && wm == 0;
    f32x16 accb[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) accb[b][v] = 0.f;

    auto image_of = [&](int pixel) -> uint32_t {
      const int px = pixel < p.npix ? pixel : p.npix - 1;
      return fdiv((uint32_t)px, p.div_phw);
    };
    // row table entry: {byte offset of the row's reference pixel from the base image, y0, x0, valid}
    auto row_entry = [&](int pixel, uint32_t nf) -> int4 {
      int4 e = make_pix<TRANSPOSED>(p, pixel);
      e.x = ((e.x - (int)(nf * (uint32_t)p.pHW)) + e.y * pW + e.z) * pld * 4;
      return e;
    };
    auto tap_of = [&](int kt, uint32_t& rs, uint32_t& chunk) {
      const uint32_t c1 = fdiv((uint32_t)kt, p.div_taps), r1 = (uint32_t)kt - c1 * (uint32_t)p.ntaps;
      const uint32_t r2 = fdiv((uint32_t)kt, p.div_cpt), c2 = (uint32_t)kt - r2 * (uint32_t)p.cpt;
      rs = p.kperm ? r1 : r2;
      chunk = p.kperm ? c1 : c2;
    };

    // ================= loop-invariant DMA state: this wave issues pieces 4 j + wave (j = 0..3) of each operand tile =================
    // ---- A ----
    int a_voff[4];                 // KC: byte offset of (row, source chunk) from the base image.  MC: unused
    int a_y0[4], a_x0[4];          // KC: the row's window origin.  MC: a_y0[j] = this lane's pixel row 0..31 of piece j
    int a_dy = 0, a_dx = 0, a_coloff = 0;          // MC (bwd-filter): this lane's fixed window element
    bool a_cvalid = true;
    unsigned long long a_boff = 0;                 // KC: element offset of the tile's base image
    if constexpr (A_KC) {
      const uint32_t nf = image_of(m0);
      if (tid < BM) pixtab[tid] = row_entry(m0 + tid, nf);
      a_boff = (unsigned long long)nf * (unsigned long long)p.pHW * (unsigned long long)pld;
    } else {
      if (tid < 2 * BK) {
        const int which = tid / BK, e = tid % BK;
        const int pix0 = (kt_begin + which) * BK;
        pixtab[which * BK + e] = row_entry(pix0 + e, image_of(pix0));
      }
      const ColDec d = decode_col(p, m0 + 4 * (lane & 31));
      a_dy = d.r; a_dx = d.s;
      a_coloff = ((d.r * pW + d.s) * pld + d.c) * 4;
      a_cvalid = d.valid;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = 4 * j + wave;
      if constexpr (A_KC) {
        const int row = 8 * pc + (lane >> 3);
        const int4 pt = pixtab[row];
        a_voff[j] = pt.x + 16 * ((lane & 7) ^ ((row >> 1) & 7));
        a_y0[j] = pt.w ? pt.y : -(1 << 30);
        a_x0[j] = pt.z;
      } else {
        a_voff[j] = 0; a_x0[j] = 0;
        a_y0[j] = 2 * pc + (lane >> 5);
      }
    }
    // ---- B ----
    uint32_t b_voff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int pc = 4 * j + wave;
      if constexpr (B_KC) {              // bwd-data: filter W[rs][cin][cout] read as rows = cin, k = (rs, cout)
        const int row = 8 * pc + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        b_voff[j] = (n0 + row < p.N) ? (uint32_t)(((n0 + row) * p.Cg + 4 * c) * 4) : kOOB;
      } else {                           // forward: filter rows k; bwd-filter: dz rows = pixels.  Columns n0 + 4 (lane & 31) ..
        const int col = n0 + 4 * (lane & 31);
        b_voff[j] = (col < p.N) ? (uint32_t)(((2 * pc + (lane >> 5)) * p.ldb + col) * 4) : kOOB;
      }
    }
    const g2_u32x4 rsB_all = g2_rsrc(p.B, p.b_elems * 4ull);

    // requests of tile kt into stage st: split into the A pieces and the B pieces so that the loop can place them apart
    int t_dy = 0, t_dx = 0, t_coloff = 0;
    uint32_t t_soffB = 0;
    g2_u32x4 t_rsA = g2_rsrc(p.A, 0), t_rsB = g2_rsrc(p.B, 0);
    auto prepare_tile = [&](int kt, bool live) {           // scalar work of tile kt's requests
      if constexpr (A_KC) {
        uint32_t rs, chunk;
        tap_of(kt, rs, chunk);
        const uint32_t r = fdiv(rs, p.div_s), sx = rs - r * p.div_s.d;
        t_dy = SGN * (int)r; t_dx = SGN * (int)sx;
        t_coloff = ((t_dy * pW + t_dx) * pld + (int)chunk * BK) * 4;
        t_rsA = g2_rsrc(p.A + a_boff, live ? (p.a_elems - a_boff) * 4ull : 0ull);
        if constexpr (MODE == MODE_FWD) {
          t_soffB = (rs * (uint32_t)p.Cg + chunk * BK) * (uint32_t)p.ldb * 4u;
        } else {
          const uint32_t rp = fdiv(rs, p.div_s), sp2 = rs - rp * p.div_s.d;
          const uint32_t rsf = (p.tap_r0 + p.sub_step * rp) * p.S_full + p.tap_s0 + p.sub_step * sp2;
          t_soffB = (rsf * (uint32_t)(p.Cn * p.Cg) + chunk * BK) * 4u;
        }
        t_rsB = live ? rsB_all : g2_rsrc(p.B, 0);
      } else {
        const int pix0 = kt * BK;
        const uint32_t nf = image_of(pix0);
        const unsigned long long boff = (unsigned long long)nf * (unsigned long long)p.pHW * (unsigned long long)pld;
        t_rsA = g2_rsrc(p.A + boff, live ? (p.a_elems - boff) * 4ull : 0ull);
        // dz rows kt*32 ..: the descriptor is re-based and ends with the tensor, so pixels past the last read as zeros
        const unsigned long long brow = (unsigned long long)pix0 * (unsigned long long)p.ldb;
        const bool some = live && brow < p.b_elems;
        t_rsB = g2_rsrc(p.B + (some ? brow : 0ull), some ? (p.b_elems - brow) * 4ull : 0ull);
        t_soffB = 0;
      }
    };
    auto dma_a = [&](int j, int st, int slot) {            // slot: pixel-table buffer of the tile (bwd-filter)
      uint32_t off;
      if constexpr (A_KC) {
        const int y = a_y0[j] + t_dy, x = a_x0[j] + t_dx;
        const bool ok = ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)pW);
        off = ok ? (uint32_t)(a_voff[j] + t_coloff) : kOOB;
      } else {
        const int4 pt = pixtab[slot * BK + a_y0[j]];
        const int y = pt.y + a_dy, x = pt.z + a_dx;
        const bool ok = a_cvalid & (pt.w != 0) & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)pW);
        off = (uint32_t)(pt.x + a_coloff) | (ok ? 0u : kOOB);
      }
      g2_dma(t_rsA, off, 0u, lds0 + (uint32_t)(st * Cfg::STAGE + (4 * j + wave) * Cfg::A_STEP));
    };
    auto dma_b = [&](int j, int st) {
      g2_dma(t_rsB, b_voff[j], t_soffB, lds0 + (uint32_t)(st * Cfg::STAGE + Cfg::A_BYTES + (4 * j + wave) * Cfg::B_STEP));
    };

    // ================= fragments =================
    // KC: f32x4 = k 8u + 4 lh + (0..3) of row g*32 + li.  MC: f32x2 = columns li, 32 + li of k row 8u + 4 lh + j.
    f32x4 akq[2][2], bkq[2][2];
    g2_f32x2 amq[2][4], bmq[2][4];
    const int a_kc_row[2] = {wm * 64 + li, wm * 64 + 32 + li}, b_kc_row[2] = {wn * 64 + li, wn * 64 + 32 + li};
    const int a_mc_base = lh * 2 * G2_MC_PIECE + (wm * 64 + li) * 4, b_mc_base = lh * 2 * G2_MC_PIECE + (wn * 64 + li) * 4;
    constexpr int NRA = A_KC ? 2 : 4, NRB = B_KC ? 2 : 4, NR = NRA + NRB;
    auto read_one = [&](int st, int u, int set, int idx) {           // idx-th fragment read of chunk u (A's first)
      const unsigned char* As = smem_raw + st * Cfg::STAGE;
      const unsigned char* Bs = As + Cfg::A_BYTES;
      if (idx < NRA) {
        if constexpr (A_KC) {
          akq[set][idx] = *reinterpret_cast<const f32x4*>(As + a_kc_row[idx] * 128 + 16 * ((2 * u + lh) ^ ((a_kc_row[idx] >> 1) & 7)));
        } else {
          const float* q = reinterpret_cast<const float*>(As + a_mc_base + (4 * u + (idx >> 1)) * G2_MC_PIECE + (idx & 1) * 512);
          amq[set][idx] = g2_f32x2{q[0], q[32]};
        }
      } else {
        const int i = idx - NRA;
        if constexpr (B_KC) {
          bkq[set][i] = *reinterpret_cast<const f32x4*>(Bs + b_kc_row[i] * 128 + 16 * ((2 * u + lh) ^ ((b_kc_row[i] >> 1) & 7)));
        } else {
          const float* q = reinterpret_cast<const float*>(Bs + b_mc_base + (4 * u + (i >> 1)) * G2_MC_PIECE + (i & 1) * 512);
          bmq[set][i] = g2_f32x2{q[0], q[32]};
        }
      }
    };
    auto reads_of_quad = [&](int st, int u, int set, int q) {        // quad q's share of chunk u's reads
#pragma unroll
      for (int idx = 0; idx < NR; ++idx)
        if (idx * 4 / NR == q) read_one(st, u, set, idx);
    };
    auto mfma_q = [&](auto bias_c, int set, int j) {
      constexpr bool BIAS = decltype(bias_c)::value;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(A_KC ? akq[set][a][j] : amq[set][j][a],
                                                           B_KC ? bkq[set][b][j] : bmq[set][j][b], acc[a][b], 0, 0, 0);
      if constexpr (BIAS) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
          accb[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f, B_KC ? bkq[set][b][j] : bmq[set][j][b], accb[b], 0, 0, 0);
      }
    };

    // ================= the K loop =================
    auto k_loop = [&](auto bias_c) {
      if (nkt > 0) {
        prepare_tile(kt_begin, true);
#pragma unroll
        for (int j = 0; j < 4; ++j) { dma_a(j, 0, 0); dma_b(j, 0); }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (nkt > 0) {
#pragma unroll
        for (int idx = 0; idx < NR; ++idx) read_one(0, 0, 0, idx);
      }
#ifdef A3D_STAMPS
      A3D_STAMP(st_beg);
      A3D_RTSTAMP(st_rt0);
      if (seg == 0) { st_pro = st_beg - st_entry; st_rtfirst = st_rt0; }
#endif
      auto tile_body = [&](const int it, auto cur_c) {
        constexpr int cur = decltype(cur_c)::value;
        const int kt = kt_begin + it;
        G2_FENCE();
        prepare_tile(kt + 1, it + 1 < nkt);
        G2_FENCE();
        // chunks 0..2: the MFMA quads of chunk u, each followed by its share of chunk u + 1's fragment reads; the A requests
        // of tile kt + 1 ride in chunk 0, the B requests in chunk 1, the pixel table of tile kt + 2 (bwd-filter) in chunk 2
#pragma unroll
        for (int u = 0; u < 3; ++u) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            mfma_q(bias_c, u & 1, q);
            G2_FENCE();
            reads_of_quad(cur, u + 1, (u + 1) & 1, q);
            if (u == 0) dma_a(q, cur ^ 1, (it + 1) % 3);
            if (u == 1) dma_b(q, cur ^ 1);
            if (MODE == MODE_BWD_F && u == 2 && q == 0) {
              if (wave == (it & 3) && lane < BK)
                pixtab[((it + 2) % 3) * BK + lane] = row_entry((kt + 2) * BK + lane, image_of((kt + 2) * BK));
            }
            G2_FENCE();
          }
        }
        // chunk 3: two quads, then the tile's barrier (every request of tile kt + 1 has landed: they were issued three chunks
        // ago), then the first fragments of tile kt + 1, covered by the last two quads
        mfma_q(bias_c, 1, 0);
        mfma_q(bias_c, 1, 1);
        G2_FENCE();
#ifdef A3D_STAMPS                              // (no LDS read is outstanding here: the stamps' lgkmcnt(0) waits for nothing else)
        A3D_STAMP(st_s0);
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef A3D_STAMPS
        A3D_STAMP(st_s1);
#endif
        __builtin_amdgcn_s_barrier();
#ifdef A3D_STAMPS
        A3D_STAMP(st_s2);
        st_wait += st_s1 - st_s0; st_bar += st_s2 - st_s1;
#endif
        G2_FENCE();
#pragma unroll
        for (int idx = 0; idx < NR; ++idx) read_one(cur ^ 1, 0, 0, idx);
        G2_FENCE();
        mfma_q(bias_c, 1, 2);
        mfma_q(bias_c, 1, 3);
        G2_FENCE();
      };
      for (int it = 0; it < nkt; it += 2) {
        tile_body(it, std::integral_constant<int, 0>{});
        if (it + 1 < nkt) tile_body(it + 1, std::integral_constant<int, 1>{});
      }
#ifdef A3D_STAMPS
      A3D_STAMP(st_end);
      A3D_RTSTAMP(st_rt1);
      st_loop += st_end - st_beg; st_nkt += (unsigned long long)nkt; st_rt += st_rt1 - st_rt0;
#endif
    };
    if (do_bias) k_loop(std::true_type{});
    else k_loop(std::false_type{});

    // ================= epilogue =================
    if (MODE == MODE_BWD_F && do_bias && lh == 0) {
      // row 0 of accb[b] (register 0 of the lanes with lh = 0) = sum over this share's pixels of dz[:, n0 + wn*64 + b*32 + li]
      const bool partial = !p.streamk && p.splitk > 1;
      const bool slab = p.streamk && (kt_begin != 0 || kt_end != nk_total);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int col = wn * 64 + b * 32 + li;
        if (slab) p.sk_bias[((size_t)2 * bid + (seg > 0 ? 1 : 0)) * BN + col] = accb[b][0];
        else if (n0 + col < p.N) p.dbias[(partial ? (size_t)split * p.N : 0) + n0 + col] = accb[b][0];
      }
    }
    igemm_epilogue<MODE, BM, BN, 2, 2, 64, 64>(p, acc, split, bid, seg, kt_begin, kt_end, nk_total, false, 0.f, tid, wave, lane, m0, n0,
                                               wm, wn);
  }   // shares of this block
#ifdef A3D_STAMPS
  {
    unsigned long long st_exit = 0;
    A3D_STAMP(st_exit);
    if (p.stamps && lane == 0) {               // slots as igemm_body writes them (eight waves per block there: four stay empty)
      unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 16;
      o[0] = 0; o[1] = st_loop - st_wait - st_bar; o[2] = st_wait; o[3] = 0; o[4] = st_bar; o[5] = st_loop; o[6] = st_nkt; o[7] = 0;
      o[8] = st_pro; o[9] = st_entry; o[10] = st_end; o[11] = st_exit; o[12] = st_rt; o[13] = st_rtfirst; o[14] = st_rt1;
    }
  }
#endif
}

}  // namespace a3d
