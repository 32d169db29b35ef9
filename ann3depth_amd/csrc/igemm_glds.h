// igemm_glds.h — the fp32 implicit GEMM of igemm.h with its tiles staged by LDS-DMA (global_load_lds_dwordx4):
// no VGPR round trip, no ds_write, no packing VALU.  An ablation of the register-staged kernel (conv2d_1 forward,
// 8-wave 128x128) showed the matrix pipe at 131 TFLOP/s with the staging removed against 106 with it; this variant
// takes the staging off the waves' instruction streams.
//
// LDS-DMA writes a wave-instruction's 64 x 16 B contiguously (base + lane*16), so tiles cannot be padded:
//   * K-contiguous tiles ([rows][32 floats] = 128-B rows: im2col in FWD / BWD_D, filter in BWD_D) are XOR-swizzled
//     instead: 16-B chunk c of row r lives at chunk position c ^ ((r >> 1) & 7).  The swizzle is applied to the
//     per-lane SOURCE address (the destination stays lane-linear) and again when the fragment is read with
//     ds_read_b128: every 16-lane group of that read then touches 16 different 16-B slots of the 256-B bank row.
//   * tiles read along their rows with ds_read_b32 (filter [k][n] in FWD, both operands in BWD_F) need no padding.
// Out-of-range elements are fetched from a zero line.  Vector (16-B) operands only: Cin % 4 == Cout % 4 == 0.
#pragma once
#include "igemm.h"

namespace a3d {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

template <int MODE, int BM, int BN, int WAVES_M, int NWAVES>
struct GldsCfg {
  static constexpr int BK = 32;
  static constexpr int NT = 64 * NWAVES;
  static constexpr int WAVES_N = NWAVES / WAVES_M;
  static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  static constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1 && TM * 32 * WAVES_M == BM && TN * 32 * WAVES_N == BN, "tile");
  static constexpr int A_ROWS = (MODE == MODE_BWD_F) ? BK : BM;
  static constexpr int A_COLS = (MODE == MODE_BWD_F) ? BM : BK;
  static constexpr int B_ROWS = (MODE == MODE_BWD_D) ? BN : BK;
  static constexpr int B_COLS = (MODE == MODE_BWD_D) ? BK : BN;
  static constexpr int A_ELEMS = A_ROWS * A_COLS, B_ELEMS = B_ROWS * B_COLS;
  static constexpr int A_CHUNKS = A_ELEMS / 4, B_CHUNKS = B_ELEMS / 4;
  static constexpr int A_NI = (A_CHUNKS + NT - 1) / NT, B_NI = (B_CHUNKS + NT - 1) / NT;   // LDS-DMAs per thread
  static_assert(A_CHUNKS % 64 == 0 && B_CHUNKS % 64 == 0, "whole wave-instructions");
  static constexpr int PIX = A_ROWS;
  static constexpr size_t LDS_BYTES = (size_t)(2 * (A_ELEMS + B_ELEMS)) * 4 + (size_t)2 * PIX * 16;
};

template <int MODE, int BM, int BN, int WAVES_M, int NWAVES>
__global__ __launch_bounds__(64 * NWAVES, NWAVES / 2) void igemm_glds_kernel(const IgemmParams p) {
  using Cfg = GldsCfg<MODE, BM, BN, WAVES_M, NWAVES>;
  constexpr int BK = Cfg::BK, TM = Cfg::TM, TN = Cfg::TN;
  constexpr bool TRANSPOSED = (MODE == MODE_BWD_D);
  constexpr bool A_SWZ = (MODE != MODE_BWD_F);        // K-contiguous 128-B rows
  constexpr bool B_SWZ = (MODE == MODE_BWD_D);
  constexpr int A_CPR = Cfg::A_COLS / 4, B_CPR = Cfg::B_COLS / 4;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* As = reinterpret_cast<float*>(smem_raw);
  float* Bs = As + 2 * Cfg::A_ELEMS;
  int4* pixtab = reinterpret_cast<int4*>(Bs + 2 * Cfg::B_ELEMS);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

  const uint32_t nwg = gridDim.x;
  uint32_t bid = blockIdx.x;
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

  const bool do_bias = (MODE == MODE_BWD_F) && p.dbias != nullptr && tile_m == 0 && tid < BN;
  float bsum = 0.f;

  if (MODE == MODE_BWD_F) {
    if (tid < 2 * Cfg::PIX) {
      int which = tid / Cfg::PIX, e = tid % Cfg::PIX;
      pixtab[which * Cfg::PIX + e] = make_pix<false>(p, (kt_begin + which) * BK + e);
    }
  } else {
    if (tid < Cfg::PIX) pixtab[tid] = make_pix<TRANSPOSED>(p, m0 + tid);
  }
  __syncthreads();

  // Chunk id of this lane in LDS-DMA j of a tile with CPR chunks per row: ((j*NWAVES + wave)*64 + lane).
  // For the swizzled tiles rows of successive j differ by a multiple of 16, so the source chunk is the same for all j.
  const int a_id0 = wave * 64 + lane, b_id0 = wave * 64 + lane;
  const int a_row0 = a_id0 / A_CPR, a_cpos = a_id0 % A_CPR;
  const int a_chunk = A_SWZ ? (a_cpos ^ swz(a_row0)) : a_cpos;          // source chunk along the tile's columns
  const int b_row0 = b_id0 / B_CPR, b_cpos = b_id0 % B_CPR;
  const int b_chunk = B_SWZ ? (b_cpos ^ swz(b_row0)) : b_cpos;
  static_assert(!A_SWZ || (Cfg::NT / A_CPR) % 16 == 0, "row step between DMAs must keep the swizzle");
  static_assert(!B_SWZ || (Cfg::NT / B_CPR) % 16 == 0, "row step between DMAs must keep the swizzle");
  constexpr int A_RSTEP = Cfg::NT / A_CPR;     // rows between successive DMAs of one thread (NT % CPR == 0 here)
  static_assert(Cfg::NT % A_CPR == 0, "A chunk columns fixed per thread");

  ColDec cdec;
  if (MODE == MODE_BWD_F) cdec = decode_col(p, m0 + a_chunk * 4);

  auto issue_a = [&](float* dst, const int4* ptab, const ColDec& cd) {
#pragma unroll
    for (int j = 0; j < Cfg::A_NI; ++j) {
      if (Cfg::A_CHUNKS % Cfg::NT != 0 && (j * NWAVES + wave) * 64 >= Cfg::A_CHUNKS) break;     // wave-uniform
      const int row = a_row0 + j * A_RSTEP;
      const int4 pt = ptab[row];
      int y = TRANSPOSED ? pt.y - cd.r : pt.y + cd.r;
      int x = TRANSPOSED ? pt.z - cd.s : pt.z + cd.s;
      bool ok = cd.valid && pt.w;
      if (TRANSPOSED) {
        ok = ok && (((y | x) & (p.stride - 1)) == 0) && y >= 0 && x >= 0;
        y >>= p.lstride;
        x >>= p.lstride;
      }
      ok = ok && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const uint32_t off = (uint32_t)(pt.x + y * p.W + x) * (uint32_t)p.ld + (uint32_t)cd.c;
      glds16(ok ? p.A + off : g_zero_line, dst + (j * NWAVES + wave) * 256);
    }
  };
  auto issue_b = [&](float* dst, int kt) {
#pragma unroll
    for (int j = 0; j < Cfg::B_NI; ++j) {
      if (Cfg::B_CHUNKS % Cfg::NT != 0 && (j * NWAVES + wave) * 64 >= Cfg::B_CHUNKS) break;     // wave-uniform
      const int id = (j * NWAVES + wave) * 64 + lane;
      const float* src;
      if constexpr (MODE == MODE_BWD_D) {
        // filter tile [BN rows = cin][32 k], swizzled; k = (rs, cout)
        const int row = id / B_CPR, cpos = id % B_CPR;
        const int kcol = kt * BK + 4 * (cpos ^ swz(row));
        const bool kvalid = kcol < p.K;
        const uint32_t k = kvalid ? (uint32_t)kcol : 0u;
        uint32_t rs = fdiv(k, p.div_c);
        const int ko = (int)(k - rs * p.div_c.d);
        if (p.sub_step > 1) {
          const uint32_t rp = fdiv(rs, p.div_s);
          const uint32_t sp = rs - rp * p.div_s.d;
          rs = (p.tap_r0 + p.sub_step * rp) * p.S_full + p.tap_s0 + p.sub_step * sp;
        }
        const int cin = n0 + row;
        const uint32_t off = rs * (uint32_t)(p.Cn * p.Cg) + (uint32_t)cin * (uint32_t)p.Cg + (uint32_t)ko;
        src = (kvalid && cin < p.N) ? p.B + off : g_zero_line;
      } else {
        const int row = id / B_CPR, c = id % B_CPR;
        const int gr = kt * BK + row, gc = n0 + 4 * c;
        src = (gr < p.K && gc < p.N) ? p.B + (uint32_t)gr * (uint32_t)p.ldb + (uint32_t)gc : g_zero_line;
      }
      glds16(src, dst + (j * NWAVES + wave) * 256);
    }
  };
  auto issue_tiles = [&](int kt, int buf, int pbuf) {
    if constexpr (MODE == MODE_BWD_F) {
      issue_a(As + buf * Cfg::A_ELEMS, pixtab + pbuf * Cfg::PIX, cdec);
    } else {
      const ColDec cd = decode_col(p, kt * BK + a_chunk * 4);
      issue_a(As + buf * Cfg::A_ELEMS, pixtab, cd);
    }
    issue_b(Bs + buf * Cfg::B_ELEMS, kt);
  };
  (void)b_chunk;

  if (nkt > 0) issue_tiles(kt_begin, 0, 0);
  __syncthreads();                       // includes the vmcnt(0) that lands the LDS-DMAs

  int cur = 0;
  for (int it = 0; it < nkt; ++it) {
    const int kt = kt_begin + it;
    const bool more = it + 1 < nkt;
    // the other buffer was last read in iteration it-1, and every wave has passed that iteration's barrier
    if (more) issue_tiles(kt + 1, cur ^ 1, (it + 1) & 1);
    if (MODE == MODE_BWD_F) {
      if (tid < Cfg::PIX) pixtab[(it & 1) * Cfg::PIX + tid] = make_pix<false>(p, (kt + 2) * BK + tid);
    }
    const float* Ac = As + cur * Cfg::A_ELEMS;
    const float* Bc = Bs + cur * Cfg::B_ELEMS;
    if (MODE == MODE_BWD_F && do_bias) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) bsum += Bc[k * BN + tid];
    }
#pragma unroll
    for (int u = 0; u < BK / 8; ++u) {
      f32x4 af[TM], bf[TN];
      const int kk = 8 * u + 4 * lh;
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const int row = wm * Cfg::WM + a * 32 + li;
        if (MODE == MODE_BWD_F) {
#pragma unroll
          for (int j = 0; j < 4; ++j) af[a][j] = Ac[(kk + j) * BM + row];
        } else {
          af[a] = *reinterpret_cast<const f32x4*>(Ac + row * BK + 4 * ((2 * u + lh) ^ swz(row)));
        }
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = wn * Cfg::WN + b * 32 + li;
        if (MODE == MODE_BWD_D) {
          bf[b] = *reinterpret_cast<const f32x4*>(Bc + col * BK + 4 * ((2 * u + lh) ^ swz(col)));
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[b][j] = Bc[(kk + j) * BN + col];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][j], bf[b][j], acc[a][b], 0, 0, 0);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue (as igemm_kernel) ----
  float* Cout = p.C;
  int ldc = p.ldc;
  const bool partial = p.splitk > 1;
  if (partial) {
    Cout = p.C + (size_t)split * p.slab;
    ldc = p.N;
  }
  if (MODE == MODE_BWD_F && do_bias && n0 + tid < p.N) p.dbias[(partial ? (size_t)split * p.N : 0) + n0 + tid] = bsum;
#pragma unroll
  for (int a = 0; a < TM; ++a) {
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
            if (p.mask) val = apply_act_grad(val, p.mask[o], p.mask_act, p.mask_scale);
          }
        }
        Cout[o] = val;
      }
    }
  }
}

}  // namespace a3d
