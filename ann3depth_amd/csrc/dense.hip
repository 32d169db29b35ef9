// dense.hip — weight-streaming kernels for the dense layers of a small batch (m <= 64 rows: coarse/dense/dense_0 and
// dense_1 of MSDN, src/models.py:228,231).  At m = 32 each layer is a 201 MB / 67 MB sweep over its weights (or their
// gradient / Adam slot) with 16 FLOP per byte: HBM-bound, and a 128-wide MFMA tile pipeline built for conv layers is the
// wrong tool (0.29-0.46 of 8 TB/s).  Here a wave owns a strip of the weight matrix, takes its operands
// straight from global memory (x and dz are a few hundred KB: L2-resident) and streams the strip once.
//
//   dense_dw_kernel<ADAM = false> : dw[k][n] = sum_m x[m][k] * dz[m][n]               (BiasAddGrad rides along)
//   dense_dw_kernel<ADAM = true>  : the same sum goes straight into ApplyAdam's m slot for the reference's optimizer
//                                   AdamOptimizer(rate, 0.9, beta2 = 1) (src/models.py:309): alpha = 0 and 1-beta2 = 0,
//                                   so only m moves (adam_frozen_kernel, pointwise.hip); dw is never written and not
//                                   re-read: 2 HBM streams (m in, m out) instead of 4 (dw out; dw, m in; m out).
//
// fp32 MFMA 32x32x2 with the batch as the contraction axis: D[k][n] += A[k][m] * B[m][n], A = x^T, B = dz; the sum runs
// over m in ascending order, one fmaf per term — the order of the BiasAddGrad sum too.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "a3d_internal.h"
#include "igemm.h"

// cache policy of the once-read streams (weights, Adam slots); -DA3D_DENSE_AUX=0 for the A/B
#ifndef A3D_DENSE_AUX
#define A3D_DENSE_AUX kAuxStream
#endif

namespace a3d {


// A wave owns 32 weight rows x 32*CW columns.  Lane li takes dz[m][n0 + CW li .. + CW-1] with ONE 4*CW-byte load and
// feeds the CW values to CW MFMAs, so accumulator j holds the columns n0 + CW li + j: the CW accumulators of a lane are
// ADJACENT columns of one row, and a wave-instruction of the epilogue moves two row segments of 128*CW bytes — for the m
// slot's read (issued before the contraction starts, so its HBM latency passes under the operand loads and the MFMAs)
// as for the write.  No LDS.  CW = 4 for the plain gradient (16-byte accesses); the Adam form holds the m tile in
// registers on top of the accumulators and runs CW = 2 (half the registers, twice the waves per SIMD: what hides the
// latency of a read-modify-write stream is waves in flight).
template <int CW>
struct VecOf;
template <>
struct VecOf<4> { typedef float type __attribute__((ext_vector_type(4), aligned(4))); };   // rows of [.., 4070] are 8-byte aligned
template <>
struct VecOf<2> { typedef float type __attribute__((ext_vector_type(2), aligned(4))); };

template <bool ADAM, int CW>
__global__ __launch_bounds__(256) void dense_dw_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                       float* __restrict__ dw, float* __restrict__ db,
                                                       float* __restrict__ var_w, float* __restrict__ m_w,
                                                       float* __restrict__ v_w, float* __restrict__ var_b,
                                                       float* __restrict__ m_b, float* __restrict__ v_b, int M, int K,
                                                       int N, float omb1, float gscale) {
  typedef typename VecOf<CW>::type vec;
  constexpr int TILE_N = 32 * CW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * TILE_N, k0 = (blockIdx.y * 4 + wave) * 32;
  const bool use_scale = gscale != 1.f;
  const float qnan = __builtin_nanf("");

  // BiasAddGrad (+ its Adam step): the blocks of the first row group, one column per thread
  if (blockIdx.y == 0 && tid < TILE_N && n0 + tid < N && (ADAM ? m_b != nullptr : db != nullptr)) {
    const int col = n0 + tid;
    float s = 0.f;
    for (int m0 = 0; m0 < M; m0 += 8) {       // eight loads in flight, added in row order
      float t[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) t[i] = m0 + i < M ? dz[(size_t)(m0 + i) * N + col] : 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (m0 + i < M) s += t[i];
    }
    if (ADAM) {
      const float g = use_scale ? __fmul_rn(s, gscale) : s;
      const float mo = m_b[col];
      const float mn = __fadd_rn(mo, __fmul_rn(__fsub_rn(g, mo), omb1));
      m_b[col] = mn;
      const bool poison = !isfinite(__fmul_rn(g, g));
      if (poison) v_b[col] = qnan;
      if (poison || !isfinite(mn)) var_b[col] = qnan;
    } else {
      db[col] = s;
    }
  }
  if (k0 >= K) return;      // wave-uniform

  const int col0 = n0 + CW * li;                  // this lane's CW columns
  const bool cfull = col0 + CW - 1 < N;           // all exist (only the last lanes of the last column group may not)
  auto row_of = [&](int v) -> int { return k0 + (v & 3) + 8 * (v >> 2) + 4 * lh; };

  vec mold[ADAM ? 16 : 1];
  if (ADAM) {
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const int row = row_of(v);
#pragma unroll
      for (int j = 0; j < CW; ++j) mold[v][j] = 0.f;
      if (row < K && cfull) mold[v] = *reinterpret_cast<const vec*>(m_w + (size_t)row * N + col0);
    }
  }

  f32x16 acc[CW];
#pragma unroll
  for (int j = 0; j < CW; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;

  // A[i = weight row k0+li][kk = batch row 2t+lh] = x[2t+lh][k0+li];  B_j[kk][li] = dz[2t+lh][col0 + j]
  const int krow = k0 + li;
  const bool kok = krow < K;
  const int T = (M + 1) / 2;
  for (int t0 = 0; t0 < T; t0 += 8) {
    float a[8];
    vec bq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = 2 * (t0 + u) + lh;
      const bool mok = m < M;
      a[u] = (mok && kok) ? x[(size_t)m * K + krow] : 0.f;
#pragma unroll
      for (int j = 0; j < CW; ++j) bq[u][j] = 0.f;
      if (mok) {
        if (cfull) {
          bq[u] = *reinterpret_cast<const vec*>(dz + (size_t)m * N + col0);
        } else {
#pragma unroll
          for (int j = 0; j < CW; ++j)
            if (col0 + j < N) bq[u][j] = dz[(size_t)m * N + col0 + j];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < CW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bq[u][j], acc[j], 0, 0, 0);
  }

  // epilogue: register v of the CW accumulators = CW adjacent columns of row row_of(v)
#pragma unroll
  for (int v = 0; v < 16; ++v) {
    const int row = row_of(v);
    if (row >= K) continue;
    const size_t o = (size_t)row * N + col0;
    if (ADAM) {
      if (cfull) {
        // ApplyAdam with alpha = 0, 1 - beta2 = 0 (adam_frozen_kernel): m moves; v / var only where a non-finite g or m
        // poisons them.  The poison test is one sum per row segment here — a non-finite term makes the sum non-finite —
        // and the per-element work happens only behind it.
        vec mn;
        float chk = 0.f;
#pragma unroll
        for (int j = 0; j < CW; ++j) {
          const float g = use_scale ? __fmul_rn(acc[j][v], gscale) : acc[j][v];
          mn[j] = __fadd_rn(mold[v][j], __fmul_rn(__fsub_rn(g, mold[v][j]), omb1));
          chk += __fmul_rn(g, g) + fabsf(mn[j]);
        }
        *reinterpret_cast<vec*>(m_w + o) = mn;
        if (!isfinite(chk)) {
#pragma unroll
          for (int j = 0; j < CW; ++j) {
            const float g = use_scale ? __fmul_rn(acc[j][v], gscale) : acc[j][v];
            const bool poison = !isfinite(__fmul_rn(g, g));
            if (poison) v_w[o + j] = qnan;
            if (poison || !isfinite(mn[j])) var_w[o + j] = qnan;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < CW; ++j) {
          if (col0 + j >= N) continue;
          const float g = use_scale ? __fmul_rn(acc[j][v], gscale) : acc[j][v];
          const float mo = m_w[o + j];
          const float mn = __fadd_rn(mo, __fmul_rn(__fsub_rn(g, mo), omb1));
          m_w[o + j] = mn;
          const bool poison = !isfinite(__fmul_rn(g, g));
          if (poison) v_w[o + j] = qnan;
          if (poison || !isfinite(mn)) var_w[o + j] = qnan;
        }
      }
    } else if (cfull) {
      vec out;
#pragma unroll
      for (int j = 0; j < CW; ++j) out[j] = acc[j][v];
      *reinterpret_cast<vec*>(dw + o) = out;
    } else {
#pragma unroll
      for (int j = 0; j < CW; ++j)
        if (col0 + j < N) dw[o + j] = acc[j][v];
    }
  }
}

// The Adam form of the big layers: the same contraction, but the block's 32 x 512 gradient tile crosses LDS so that the
// m slot is read and written as WHOLE 2-KiB row segments (256*CW contiguous bytes per wave-instruction) instead of
// 512-byte pieces 16 KB apart: the read-modify-write stream of a 201 MB slot is what this kernel costs, and HBM serves
// long runs far better (measured on dense_0: 426 us with 256-byte pieces, 100 us with 512-byte pieces).  The wave's share
// of the m tile is requested FIRST, so its HBM latency passes under the contraction (operands from L2) instead of after
// it: the LDS tile allows two blocks per CU, so the registers that hold m across the contraction cost no occupancy.
// CW = 4 needs N % 4 == 0 and 16-byte slots; CW = 2 serves rows of 8-byte alignment (dense_1: [4096, 4070]).
constexpr int kRowsLd = 512 + 4;
typedef float f32x4v __attribute__((ext_vector_type(4)));
#ifdef A3D_STAMPS
__device__ unsigned long long g_dense_stamps[4096 * 4 * 8];       // diagnostic build: [block][wave][8]
#endif
template <int CW>
__global__ __launch_bounds__(256) void dense_dw_adam_rows_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                 float* __restrict__ var_w, float* __restrict__ m_w,
                                                                 float* __restrict__ v_w, float* __restrict__ var_b,
                                                                 float* __restrict__ m_b, float* __restrict__ v_b, int M,
                                                                 int K, int N, float omb1, float gscale) {
  typedef float vec __attribute__((ext_vector_type(CW)));
  constexpr int G = 4 / CW;              // column groups of 32*CW per wave (128 columns)
  constexpr int NI = 512 / (64 * CW);    // wave-instructions per 2-KiB row segment
  __shared__ __attribute__((aligned(16))) float tile[32 * kRowsLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nb = blockIdx.x * 512, n0 = nb + wave * 128, k0 = blockIdx.y * 32;
  const bool use_scale = gscale != 1.f;
  const float qnan = __builtin_nanf("");

#ifdef A3D_STAMPS
  unsigned long long s_t0, s_t1, s_t2, s_t3, s_t4, s_t5;
#endif
  A3D_STAMP(s_t0);
  // this wave's share of the m tile: rows 8*wave .. 8*wave+7, NI pieces each
  vec mold[8 * NI];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int h = 0; h < NI; ++h) {
      const int row = k0 + wave * 8 + r, col = nb + (h * 64 + lane) * CW;
#pragma unroll
      for (int j = 0; j < CW; ++j) mold[NI * r + h][j] = 0.f;
      if (row < K && col < N) mold[NI * r + h] = *reinterpret_cast<const vec*>(m_w + (size_t)row * N + col);
    }

  if (blockIdx.y == 0 && m_b != nullptr) {        // BiasAddGrad + its Adam step, two columns per thread
    for (int col = nb + tid; col < nb + 512 && col < N; col += 256) {
      float s = 0.f;
      for (int m0 = 0; m0 < M; m0 += 8) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = m0 + i < M ? dz[(size_t)(m0 + i) * N + col] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (m0 + i < M) s += t[i];
      }
      const float g = use_scale ? __fmul_rn(s, gscale) : s;
      const float mo = m_b[col];
      const float mn = __fadd_rn(mo, __fmul_rn(__fsub_rn(g, mo), omb1));
      m_b[col] = mn;
      const bool poison = !isfinite(__fmul_rn(g, g));
      if (poison) v_b[col] = qnan;
      if (poison || !isfinite(mn)) var_b[col] = qnan;
    }
  }

  A3D_STAMP(s_t1);
  f32x16 acc[G][CW];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int j = 0; j < CW; ++j)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[g][j][v] = 0.f;
  const int krow = k0 + li;
  const bool kok = krow < K;                       // N % CW == 0: a lane's CW columns exist together
#ifdef A3D_DENSE_NOGEMM
  const int T = 0;
#else
  const int T = (M + 1) / 2;
#endif
  for (int t0 = 0; t0 < T; t0 += 8) {
    float a[8];
    vec bq[8][G];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = 2 * (t0 + u) + lh;
      const bool mok = m < M;
      a[u] = (mok && kok) ? x[(size_t)m * K + krow] : 0.f;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int col = n0 + (32 * g + li) * CW;
#pragma unroll
        for (int j = 0; j < CW; ++j) bq[u][g][j] = 0.f;
        if (mok && col < N) bq[u][g] = *reinterpret_cast<const vec*>(dz + (size_t)m * N + col);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < CW; ++j)
          acc[g][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bq[u][g][j], acc[g][j], 0, 0, 0);
  }
  A3D_STAMP(s_t2);
  // gradient tile -> LDS: register v of group g's CW accumulators = columns 128*wave + (32 g + li)*CW .. of row
  // (v&3) + 8*(v>>2) + 4*lh
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      vec t;
#pragma unroll
      for (int j = 0; j < CW; ++j) t[j] = acc[g][j][v];
      *reinterpret_cast<vec*>(&tile[((v & 3) + 8 * (v >> 2) + 4 * lh) * kRowsLd + wave * 128 + (32 * g + li) * CW]) = t;
    }
  __syncthreads();
  A3D_STAMP(s_t3);
#ifdef A3D_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  A3D_STAMP(s_t4);
#endif
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int h = 0; h < NI; ++h) {
      const int lr = wave * 8 + r, row = k0 + lr, c = (h * 64 + lane) * CW, col = nb + c;
      if (row >= K || col >= N) continue;
      const vec g4 = *reinterpret_cast<const vec*>(&tile[lr * kRowsLd + c]);
      const vec mo = mold[NI * r + h];
      const size_t o = (size_t)row * N + col;
      vec mn;
      float chk = 0.f;
#pragma unroll
      for (int j = 0; j < CW; ++j) {
        const float g = use_scale ? __fmul_rn(g4[j], gscale) : g4[j];
        mn[j] = __fadd_rn(mo[j], __fmul_rn(__fsub_rn(g, mo[j]), omb1));
        chk += __fmul_rn(g, g) + fabsf(mn[j]);
      }
      *reinterpret_cast<vec*>(m_w + o) = mn;
      if (!isfinite(chk)) {
#pragma unroll
        for (int j = 0; j < CW; ++j) {
          const float g = use_scale ? __fmul_rn(g4[j], gscale) : g4[j];
          const bool poison = !isfinite(__fmul_rn(g, g));
          if (poison) v_w[o + j] = qnan;
          if (poison || !isfinite(mn[j])) var_w[o + j] = qnan;
        }
      }
    }
#ifdef A3D_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  A3D_STAMP(s_t5);
  if (lane == 0 && blockIdx.y * gridDim.x + blockIdx.x < 4096) {
    unsigned long long* o = g_dense_stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 8;
    o[0] = s_t0; o[1] = s_t1; o[2] = s_t2; o[3] = s_t3; o[4] = s_t4; o[5] = s_t5;
    unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); o[6] = hw;
  }
#endif
}

// The same update for batches of at most 32 rows, as a stream: a block keeps its 512 columns of dz in REGISTERS (64 per
// lane) and walks down the weight rows, 32 at a time.  Measured on dense_0 with the kernel above: 87 us, and 56 us
// (7.2 TB/s, the rate of an in-place multiply) with its contraction removed — the 64 KB of dz each tile re-read from L2
// and the operand loads queued behind the m requests (loads return in order) cost more than the HBM stream itself.
// Here dz is read once per block, and the next group's x rows and m tile are requested while the current group is
// updated and stored, x FIRST, so that the contraction of the next group never waits on HBM.  Buffer loads and stores
// throughout (rows and columns past the end go to the out-of-range offset): the loop body is straight-line code.
// SL = 2 ("slim", CW = 2 only): half the columns per wave — 64 instead of 128, so half the dz, accumulator and m registers
// (~120 instead of ~220): beside an 8-wave GEMM block of the other stream that holds half of every SIMD's register file
// (A3D_HINT_SHARE_CU) TWO of these blocks are resident per CU instead of one.
// BF (precision A3D_PREC_BF16, batches of up to 64 rows: BASELINE config 5): the contraction on the bf16 matrix cores —
// x and dz rounded to bf16 once (dz into packed registers, x while it is parked in LDS, transposed so that a lane's eight
// batch rows are one 16-byte read), fp32 accumulation, four v_mfma_f32_32x32x16_bf16 per tile instead of 32
// v_mfma_f32_32x32x2_f32 of 64 cycles each: at 64 rows the fp32 form needs 56 % of the matrix pipe to keep up with HBM and
// shares it with the other stream's GEMM (3.8 TB/s alone, 2.7 in the step); this one is a stream again.  Columns per block
// as for 32 rows (512, 128 per wave).
typedef __bf16 dbf16x8 __attribute__((ext_vector_type(8)));
template <int CW, int MB, int SL = 1, bool BF = false>
__global__ __launch_bounds__(256, SL == 2 ? 4 : 2) void dense_dw_adam_stream_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                      float* __restrict__ var_w, float* __restrict__ m_w,
                                                                      float* __restrict__ v_w, float* __restrict__ var_b,
                                                                      float* __restrict__ m_b, float* __restrict__ v_b,
                                                                      int M, int K, int N, float omb1, float gscale,
                                                                      int gpb) {
  // MB = 1: up to 32 batch rows, a block owns 512 columns (128 per wave); MB = 2: up to 64 rows and 256 columns, so that
  // the dz registers stay at 64 per lane (batch rows x columns per wave is the same in both)
  constexpr int CDIV = BF ? SL : MB * SL;
  constexpr int BCOLS = 512 / CDIV, WCOLS = 128 / CDIV, TSTEPS = BF ? 1 : 16 * MB, TLD = BCOLS + 4;
  constexpr int G = WCOLS / (32 * CW), NI = BCOLS / (64 * CW);
  static_assert(G >= 1 && NI >= 1, "columns per lane");
  static_assert(!BF || MB == 2, "the bf16 form takes up to 64 batch rows");
  constexpr int XLD = 72;                        // bf16 elements per weight row of the transposed x tile (144 bytes: conflict-free b128 reads)
  typedef float vec __attribute__((ext_vector_type(CW)));
  typedef uint32_t uvec __attribute__((ext_vector_type(CW)));
  // the gradient tile passes through LDS sixteen rows at a time (33 KiB at MB = 1, not 66: two of these blocks fit beside
  // a GEMM block of the other stream that leaves half of the CU's LDS free); wave w updates rows 4w .. 4w+3 of each half
  __shared__ __attribute__((aligned(16))) float tile[16 * TLD];
  __shared__ __attribute__((aligned(16))) float xs[32 * MB * 32];   // x[batch row][weight row of the group]; BF: bf16 [weight row][XLD batch rows]
  __bf16* xsb = reinterpret_cast<__bf16*>(xs);
  static_assert(32 * XLD * 2 <= 32 * 2 * 32 * 4, "transposed tile fits");
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int nb = blockIdx.x * BCOLS, n0 = nb + wave * WCOLS;
  const int ngroups = (K + 31) / 32, g0 = blockIdx.y * gpb, g1 = min(g0 + gpb, ngroups);
  const bool use_scale = gscale != 1.f;
  const float qnan = __builtin_nanf("");
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, (unsigned long long)M * K * 4);
  const __amdgpu_buffer_rsrc_t rz = make_rsrc(dz, (unsigned long long)M * N * 4);

  // dz: lane's CW columns of group gq, batch rows 2u + lh (BF: rows 16 t + 8 lh + e as element e of the bf16 fragment of step t)
  float bq[TSTEPS][G][CW];
  dbf16x8 bqb[BF ? 4 : 1][G][CW];
  if constexpr (BF) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int gq = 0; gq < G; ++gq)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int m = 16 * t + 8 * lh + e, col = n0 + (32 * gq + li) * CW;
          float v[CW];
          load_vec_buf<CW>(rz, ((m < M) & (col < N)) ? (uint32_t)(((size_t)m * N + col) * 4) : kOOB, v);
#pragma unroll
          for (int j = 0; j < CW; ++j) bqb[t][gq][j][e] = (__bf16)v[j];
        }
  } else {
#pragma unroll
    for (int u = 0; u < TSTEPS; ++u)
#pragma unroll
      for (int gq = 0; gq < G; ++gq) {
        const int m = 2 * u + lh, col = n0 + (32 * gq + li) * CW;
        load_vec_buf<CW>(rz, ((m < M) & (col < N)) ? (uint32_t)(((size_t)m * N + col) * 4) : kOOB, bq[u][gq]);
      }
  }
  // x rows of a group: 32*MB batch rows x 32 weight rows = MB 16-byte loads per thread (K % 4 == 0)
  const int xm = tid >> 3, xk = (tid & 7) * 4;
  auto load_x = [&](float (&xv)[MB][4], int g) {
    const int krow = 32 * g + xk;
#pragma unroll
    for (int b = 0; b < MB; ++b)
      load_vec_buf<4>(rx, ((g < g1) & (xm + 32 * b < M) & (krow < K)) ? (uint32_t)(((size_t)(xm + 32 * b) * K + krow) * 4) : kOOB,
                      xv[b]);
  };
  auto park_x = [&](const float (&xv)[MB][4]) {
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      if constexpr (BF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xsb[(xk + i) * XLD + xm + 32 * b] = (__bf16)xv[b][i];
      } else {
        *reinterpret_cast<f32x4v*>(&xs[(xm + 32 * b) * 32 + xk]) = (f32x4v){xv[b][0], xv[b][1], xv[b][2], xv[b][3]};
      }
    }
  };
  // m tile of a group: this wave's eight rows (wrow: 4 wave .. +3 of either half of the group), NI pieces each.  One
  // resource per row (scalar arithmetic: a row that does not exist gets zero records), one lane offset for all of them.
  const int mcol = nb + lane * CW;
  const uint32_t mvoff = mcol < N ? (uint32_t)mcol * 4u : kOOB;       // N % CW == 0; later pieces: + 256*CW bytes each
  auto wrow = [&](int r) { return 16 * (r >> 2) + 4 * wave + (r & 3); };
  auto row_rsrc = [&](int g, int r) {
    const int row = 32 * g + wrow(r);
    const bool ok = (g < g1) & (row < K);
    return make_rsrc(m_w + (size_t)(ok ? row : 0) * N, ok ? (unsigned long long)N * 4 : 0ull);
  };
  auto piece_off = [&](int h) -> uint32_t { return nb + (h * 64 + lane) * CW < N ? mvoff + h * 256 * CW : kOOB; };
  auto load_m = [&](float (&mo)[8 * NI][CW], int g) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const __amdgpu_buffer_rsrc_t rr = row_rsrc(g, r);
#pragma unroll
      for (int h = 0; h < NI; ++h) load_vec_buf<CW, A3D_DENSE_AUX>(rr, piece_off(h), mo[NI * r + h]);
    }
  };

  float xv[MB][4], mreg[8 * NI][CW];
  load_x(xv, g0);
  __builtin_amdgcn_sched_barrier(0);
  load_m(mreg, g0);
  park_x(xv);

  if (blockIdx.y == 0 && m_b != nullptr) {        // BiasAddGrad + its Adam step, two columns per thread
    for (int col = nb + tid; col < nb + BCOLS && col < N; col += 256) {
      float s = 0.f;
      for (int m0 = 0; m0 < M; m0 += 8) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = m0 + i < M ? dz[(size_t)(m0 + i) * N + col] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (m0 + i < M) s += t[i];
      }
      const float g = use_scale ? __fmul_rn(s, gscale) : s;
      const float mo = m_b[col];
      const float mn = __fadd_rn(mo, __fmul_rn(__fsub_rn(g, mo), omb1));
      m_b[col] = mn;
      const bool poison = !isfinite(__fmul_rn(g, g));
      if (poison) v_b[col] = qnan;
      if (poison || !isfinite(mn)) var_b[col] = qnan;
    }
  }
  __syncthreads();

  // one row group: contraction from xs / bq, tile through LDS, then row by row: update and store this group's m and
  // request the next group's row INTO THE SAME REGISTERS (one m tile per lane, not two: the kernel fits 256 registers)
  for (int g = g0; g < g1; ++g) {
    f32x16 acc[G][CW];
#pragma unroll
    for (int gq = 0; gq < G; ++gq)
#pragma unroll
      for (int j = 0; j < CW; ++j)
#pragma unroll
        for (int v = 0; v < 16; ++v) acc[gq][j][v] = 0.f;
    if constexpr (BF) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const dbf16x8 a = *reinterpret_cast<const dbf16x8*>(&xsb[li * XLD + 16 * t + 8 * lh]);
#pragma unroll
        for (int gq = 0; gq < G; ++gq)
#pragma unroll
          for (int j = 0; j < CW; ++j)
            acc[gq][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bqb[t][gq][j], acc[gq][j], 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int u = 0; u < TSTEPS; ++u) {
      const float a = xs[(2 * u + lh) * 32 + li];
#pragma unroll
      for (int gq = 0; gq < G; ++gq)
#pragma unroll
        for (int j = 0; j < CW; ++j)
          acc[gq][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq[u][gq][j], acc[gq][j], 0, 0, 0);
    }
    }
    // gradient tile -> LDS, sixteen rows per pass: register v = 8 hh + vv of group gq's CW accumulators = columns
    // 128*wave + (32 gq + li)*CW .. of row 16 hh + (vv&3) + 8*(vv>>2) + 4*lh
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (hh) __syncthreads();               // the first half's rows have been read
#pragma unroll
      for (int gq = 0; gq < G; ++gq)
#pragma unroll
        for (int vv = 0; vv < 8; ++vv) {
          vec t;
#pragma unroll
          for (int j = 0; j < CW; ++j) t[j] = acc[gq][j][8 * hh + vv];
          *reinterpret_cast<vec*>(&tile[((vv & 3) + 8 * (vv >> 2) + 4 * lh) * TLD + wave * WCOLS + (32 * gq + li) * CW]) = t;
        }
      __syncthreads();
      if (hh == 0) {
        load_x(xv, g + 1);      // before the m requests: loads come back in issue order
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 4 * hh; r < 4 * hh + 4; ++r) {
        const __amdgpu_buffer_rsrc_t rr = row_rsrc(g, r), rn = row_rsrc(g + 1, r);
        const int row = 32 * g + wrow(r);
#pragma unroll
        for (int h = 0; h < NI; ++h) {
          const int c = (h * 64 + lane) * CW;
          const vec g4 = *reinterpret_cast<const vec*>(&tile[(4 * wave + (r & 3)) * TLD + c]);
          const uint32_t off = piece_off(h);
          uvec mn;
          float chk = 0.f;
#pragma unroll
          for (int j = 0; j < CW; ++j) {
            const float gr = use_scale ? __fmul_rn(g4[j], gscale) : g4[j];
            const float mo = mreg[NI * r + h][j];
            const float t = __fadd_rn(mo, __fmul_rn(__fsub_rn(gr, mo), omb1));
            mn[j] = __float_as_uint(t);
            chk += __fmul_rn(gr, gr) + fabsf(t);
          }
          if constexpr (CW == 4) __builtin_amdgcn_raw_buffer_store_b128(mn, rr, (int)off, 0, A3D_DENSE_AUX);
          else __builtin_amdgcn_raw_buffer_store_b64(mn, rr, (int)off, 0, A3D_DENSE_AUX);
          load_vec_buf<CW, A3D_DENSE_AUX>(rn, off, mreg[NI * r + h]);
          // ApplyAdam's v and var take a NaN where g*g or the new m is not finite (adam_frozen_kernel); a non-finite
          // term makes the piece's sum non-finite, and the per-element work happens only behind that test
          if (!isfinite(chk) & (off != kOOB) & (row < K)) {
#pragma unroll
            for (int j = 0; j < CW; ++j) {
              const float gr = use_scale ? __fmul_rn(g4[j], gscale) : g4[j];
              const bool poison = !isfinite(__fmul_rn(gr, gr));
              const size_t o = (size_t)row * N + nb + c + j;
              if (poison) v_w[o] = qnan;
              if (poison || !isfinite(__uint_as_float(mn[j]))) var_w[o] = qnan;
            }
          }
        }
      }
    }
    park_x(xv);
    __syncthreads();
  }
}

// ===================================================================================================================
// forward  y[M][N] = act(x[M][K] W[K][N] + b) (* dropout)   and   bwd-data  dx[M][K] = (dz[M][N] W[K][N]^T) * act'(mask)
// for M <= 64: one pass over W, 16 bytes per lane, the batch on the 32 rows of the fp32 MFMA.
// ===================================================================================================================
typedef float f32x4a __attribute__((ext_vector_type(4), aligned(4)));

struct DenseStreamParams {
  const float* x;        // fwd: x [M][K];  bwd-data: dz [M][N]
  const float* w;        // [K][N]
  float* out;            // y / dx, or the split slabs [S][M][cols]
  const float* bias;     // fwd
  const uint8_t* keep;   // fwd: dropout keep mask [M][N]
  const float* mask;     // bwd-data: activation whose gradient is applied, [M][K]
  float scale;           // fwd: survivors' factor; bwd-data: gradient factor
  int act;               // fwd: EPI_*; bwd-data: mask_act
  int M, K, N, splits, span;      // span: rows of W (fwd) / columns (bwd-data) per split
};

// Forward.  Block = 4 waves on the same 128 columns, each taking a quarter of the block's K span in chunks of 16 rows
// (8 KB of W in flight per wave beside the chunk being multiplied); lane li holds W[k][n0 + 4 li .. + 3] and feeds four
// column-interleaved MFMAs.  The four partial tiles meet in LDS and leave as rows (bias / activation / dropout applied
// when the launch is not split; otherwise one slab per split for splitk_reduce_kernel).
template <int MB>
__global__ __launch_bounds__(256) void dense_fwd_stream_kernel(const DenseStreamParams p) {
  // (16 rows at a time: 33 KiB instead of 66 — two of these blocks fit beside a GEMM block of the other stream that
  // leaves half of the CU's LDS free, A3D_HINT_SHARE_CU)
  __shared__ __attribute__((aligned(16))) float red[4 * 16 * 132];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.x * 128, split = blockIdx.y;
  const int kb0 = split * p.span, kb1 = min(p.K, kb0 + p.span);
  const int wspan = ((kb1 - kb0 + 3) / 4 + 15) / 16 * 16;         // rows per wave, whole chunks
  const int k0 = kb0 + wave * wspan, k1 = min(kb1, k0 + wspan);
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(p.w, (unsigned long long)p.K * p.N * 4ull);
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (unsigned long long)p.M * p.K * 4ull);
  const int col0 = n0 + 4 * li;

  f32x16 acc[MB][4];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[b][c][v] = 0.f;

  // a chunk = U sub-chunks of 8 rows; two chunks in registers (one being multiplied, one in flight).  With 64 batch rows
  // the accumulators take 128 registers: chunks of 8 rows then keep the kernel out of scratch
  constexpr int U = MB == 1 ? 2 : 1, ROWS = 8 * U;
  struct Chunk { float w[U][4][4]; float a[MB][U][4]; };
  auto load = [&](Chunk& c, int kc) {            // rows kc .. kc+ROWS-1: sub-chunk u, MFMA j, lane half lh -> row kc + 8u + 4lh + j
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // 16 bytes at dword alignment; where a lane's four columns run past N they are the next row's first (or
        // past the buffer: zeros): accumulators of columns that do not exist, never stored
        const int k = kc + 8 * u + 4 * lh + j;
        load_vec_buf<4, A3D_DENSE_AUX>(rw, ((k < k1) & (col0 < p.N)) ? (uint32_t)(((size_t)k * p.N + col0) * 4) : kOOB, c.w[u][j]);
      }
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        const int m = 32 * b + li, k = kc + 8 * u + 4 * lh;   // K % 4 == 0 and spans of whole chunks: k .. k+3 < k1 together
        load_vec_buf<4>(rx, ((m < p.M) & (k < k1)) ? (uint32_t)(((size_t)m * p.K + k) * 4) : kOOB, c.a[b][u]);
      }
    }
  };
  auto compute = [&](const Chunk& c) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[b][e] = __builtin_amdgcn_mfma_f32_32x32x2f32(c.a[b][u][j], c.w[u][j][e], acc[b][e], 0, 0, 0);
  };
  if (k0 < k1) {
    Chunk ca, cb;
    load(ca, k0);
    for (int kc = k0; kc < k1; kc += 2 * ROWS) {
      load(cb, kc + ROWS);
      compute(ca);
      load(ca, kc + 2 * ROWS);
      compute(cb);
    }
  }

  // the four waves' tiles -> LDS -> rows, sixteen rows of the batch at a time (accumulator registers 8 hh .. 8 hh + 7 hold
  // rows 16 hh .. 16 hh + 15); the four partial sums are added in wave order
#pragma unroll
  for (int b = 0; b < MB; ++b) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      if (b || hh) __syncthreads();
      float* mine = red + wave * (16 * 132);
#pragma unroll
      for (int vv = 0; vv < 8; ++vv) {
        const int v = 8 * hh + vv;
        const f32x4 q = {acc[b][0][v], acc[b][1][v], acc[b][2][v], acc[b][3][v]};
        *reinterpret_cast<f32x4*>(mine + ((vv & 3) + 8 * (vv >> 2) + 4 * lh) * 132 + 4 * li) = q;
      }
      __syncthreads();
      // the eight sums of this thread first, then its stores back to back: with the split / bias / dropout branches around
      // each store the compiler put s_waitcnt vmcnt(0) between them, and every block of the one-round grid ended with
      // sixteen store round trips in a row
      float sv[4][2];
      bool okv[4][2];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = wave * 4 + r, m = 32 * b + 16 * hh + lr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int c = lane + 64 * h;
          okv[r][h] = m < p.M && n0 + c < p.N;
          float s = red[lr * 132 + c];
          s += red[(16 + lr) * 132 + c];
          s += red[(32 + lr) * 132 + c];
          s += red[(48 + lr) * 132 + c];
          sv[r][h] = s;
        }
      }
      if (p.splits > 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if (okv[r][h]) p.out[((size_t)split * p.M + 32 * b + 16 * hh + wave * 4 + r) * p.N + n0 + lane + 64 * h] = sv[r][h];
      } else {
        float bias[2] = {0.f, 0.f};
        uint8_t kp[4][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (p.bias && n0 + lane + 64 * h < p.N) bias[h] = p.bias[n0 + lane + 64 * h];
        if (p.keep) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int h = 0; h < 2; ++h)
              kp[r][h] = okv[r][h] ? p.keep[(size_t)(32 * b + 16 * hh + wave * 4 + r) * p.N + n0 + lane + 64 * h] : (uint8_t)0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float s = sv[r][h];
            if (p.bias) s += bias[h];
            if (p.act == EPI_RELU) s = fmaxf(s, 0.f);
            else if (p.act == EPI_SIGMOID) s = 1.f / (1.f + expf(-s));
            if (p.keep) s = kp[r][h] ? s * p.scale : 0.f;
            sv[r][h] = s;
          }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if (okv[r][h]) p.out[(size_t)(32 * b + 16 * hh + wave * 4 + r) * p.N + n0 + lane + 64 * h] = sv[r][h];
      }
    }
  }
}

static int pick_span(int extent, int groups, int quantum) {
  const int target = 512;                              // two resident blocks per CU
  const int want = std::max(1, target / std::max(1, groups));
  int span = (extent + want - 1) / want;
  span = std::max(quantum, (span + quantum - 1) / quantum * quantum);
  return span;
}
bool dense_stream_applicable(int m, int k, int n) {
  return m >= 1 && m <= 64 && (long)k * n >= (1L << 20) && k % 4 == 0;      // 16-byte rows of x; W rows of any length
}
size_t dense_stream_ws_bytes(int m, int k, int n) {
  if (!dense_stream_applicable(m, k, n)) return 0;
  const int sf = pick_span(k, (n + 127) / 128, 64);
  return (size_t)((k + sf - 1) / sf) * m * n * 4;
}
int dense_fwd_stream(int m, int k, int n, const float* x, const float* w, const float* bias, float* y, int act,
                     const uint8_t* keep, float keep_scale, void* ws, size_t ws_bytes, hipStream_t st) {
  DenseStreamParams p{};
  p.x = x; p.w = w; p.bias = bias; p.keep = keep; p.scale = keep_scale; p.act = act; p.M = m; p.K = k; p.N = n;
  p.span = pick_span(k, (n + 127) / 128, 64);
  p.splits = (k + p.span - 1) / p.span;
  if (p.splits > 1 && (size_t)p.splits * m * n * 4 > ws_bytes) return set_error(A3D_EWORKSPACE, "dense_fwd: workspace too small");
  p.out = p.splits > 1 ? static_cast<float*>(ws) : y;
  clear_stale_error();
  const dim3 grid((n + 127) / 128, p.splits);
  if (m <= 32) hipLaunchKernelGGL(dense_fwd_stream_kernel<1>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(dense_fwd_stream_kernel<2>, grid, dim3(256), 0, st, p);
  int rc = check_launch("dense_fwd_stream");
  if (rc != A3D_OK || p.splits == 1) return rc;
  ReduceParams r{};
  r.ws = p.out; r.C = y; r.bias = bias; r.keep = keep; r.mask_scale = keep_scale; r.M = m; r.N = n; r.ldc = n;
  r.splitk = p.splits; r.act = act; r.mode = MODE_FWD; r.slab = (size_t)m * n; r.sub_step = 1;
  r.div_phw = make_fastdiv(1); r.div_pw = make_fastdiv(1);
  return launch_splitk_reduce(r, st);
}
bool dense_dw_applicable(int m, int k, int n) { return m >= 1 && m <= 64 && k >= 1 && n >= 1; }

int dense_dw_launch(int m, int k, int n, const float* x, const float* dz, float* dw, float* db, hipStream_t st) {
  clear_stale_error();
  hipLaunchKernelGGL((dense_dw_kernel<false, 4>), dim3((n + 127) / 128, (k + 127) / 128), dim3(256), 0, st, x, dz, dw, db,
                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, m, k, n, 0.f, 1.f);
  return check_launch("dense_dw");
}

}  // namespace a3d

using namespace a3d;

#ifdef A3D_STAMPS
extern "C" int a3d_debug_dense_stamps(unsigned long long* out, size_t bytes) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dense_stamps), std::min(bytes, sizeof(g_dense_stamps)));
}
#endif

extern "C" int a3d_dense_bwd_filter_adam_tf1(int m, int k, int n, const float* x, const float* dz, float* var_w, float* m_w,
                                             float* v_w, float* var_b, float* m_b, float* v_b, float lr, float beta1,
                                             float beta2, float beta1_power, float beta2_power, float grad_scale,
                                             void* stream) {
  return a3d_dense_bwd_filter_adam_tf1_ex(m, k, n, x, dz, var_w, m_w, v_w, var_b, m_b, v_b, lr, beta1, beta2, beta1_power,
                                          beta2_power, grad_scale, A3D_PREC_F32, stream);
}

extern "C" int a3d_dense_bwd_filter_adam_tf1_ex(int m, int k, int n, const float* x, const float* dz, float* var_w, float* m_w,
                                                float* v_w, float* var_b, float* m_b, float* v_b, float lr, float beta1,
                                                float beta2, float beta1_power, float beta2_power, float grad_scale,
                                                int precision, void* stream) {
  A3D_CHECK_ARG(precision == A3D_PREC_F32 || precision == A3D_PREC_BF16, "dense_bwd_filter_adam: precision %d (float32 or bf16)", precision);
  A3D_CHECK_ARG(m > 0 && k > 0 && n > 0 && x && dz && var_w && m_w && v_w, "dense_bwd_filter_adam: bad arguments");
  A3D_CHECK_ARG((var_b && m_b && v_b) || (!var_b && !m_b && !v_b), "dense_bwd_filter_adam: bias slots come as a set");
  A3D_CHECK_ARG(dense_dw_applicable(m, k, n), "dense_bwd_filter_adam: batches of at most 64 rows only");
  const float alpha = lr * sqrtf(1.f - beta2_power) / (1.f - beta1_power);
  A3D_CHECK_ARG(alpha == 0.f && 1.f - beta2 == 0.f,
                "dense_bwd_filter_adam: only the reference's frozen optimizer (beta2 == 1); otherwise call "
                "a3d_dense_bwd_filter and a3d_adam_apply_tf1");
  clear_stale_error();
  const uintptr_t slots = reinterpret_cast<uintptr_t>(m_w) | reinterpret_cast<uintptr_t>(dz);
  const dim3 rows_grid((n + 511) / 512, (k + 31) / 32);
  // stream form: two blocks per CU, each walking down its share of the row groups; 512 columns per block for batches of
  // at most 32 rows, 256 for up to 64
  static const bool no_stream = tune_int("A3D_NO_DENSE_STREAM", 0) != 0;   // A/B aid (tuning processes only)
  // the slim form (124 registers, four blocks per CU) for batches of at most 32 rows: measured in the step, beside the fine
  // network's hinted GEMMs
  const bool slim = m <= 32 && n % 2 == 0;
  const bool bf_cols = precision == A3D_PREC_BF16 && m > 32;      // the bf16 form: 512 columns per block at any batch
  const int bcols = slim ? 256 : ((m <= 32 || bf_cols) ? 512 : 256), colblocks = (n + bcols - 1) / bcols;
  const int gpb = std::max(1, ((k + 31) / 32 * colblocks + (slim ? 1023 : 511)) / (slim ? 1024 : 512));
  const dim3 stream_grid(colblocks, ((k + 31) / 32 + gpb - 1) / gpb);
  const bool stream_ok = m <= 64 && k % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && !no_stream;
  const hipStream_t hst = static_cast<hipStream_t>(stream);
#define A3D_DW_STREAM(CWV, MBV, ...)                                                                                     \
  hipLaunchKernelGGL((dense_dw_adam_stream_kernel<CWV, MBV, ##__VA_ARGS__>), stream_grid, dim3(256), 0, hst, x, dz, var_w, m_w, v_w, var_b, \
                     m_b, v_b, m, k, n, 1.f - beta1, grad_scale, gpb)
  // bf16 arithmetic (config 5): the bf16 matrix cores take the batch axis 16 rows per instruction — worth it above 32 rows
  const bool bf = precision == A3D_PREC_BF16 && m > 32 && stream_ok;
  if (bf && n % 4 == 0 && (slots & 15) == 0) A3D_DW_STREAM(4, 2, 1, true);
  else if (bf && n % 2 == 0 && (slots & 7) == 0) A3D_DW_STREAM(2, 2, 1, true);
  else if (stream_ok && slim && (slots & 7) == 0) A3D_DW_STREAM(2, 1, 2);
  else if (stream_ok && m <= 32 && n % 4 == 0 && (slots & 15) == 0) A3D_DW_STREAM(4, 1);
  else if (stream_ok && m <= 32 && n % 2 == 0 && (slots & 7) == 0) A3D_DW_STREAM(2, 1);
  else if (stream_ok && n % 2 == 0 && (slots & 7) == 0) A3D_DW_STREAM(2, 2);
#undef A3D_DW_STREAM
  else if (n % 4 == 0 && (slots & 15) == 0)
    hipLaunchKernelGGL(dense_dw_adam_rows_kernel<4>, rows_grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, dz, var_w,
                       m_w, v_w, var_b, m_b, v_b, m, k, n, 1.f - beta1, grad_scale);
  else if (n % 2 == 0 && (slots & 7) == 0)
    hipLaunchKernelGGL(dense_dw_adam_rows_kernel<2>, rows_grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, dz, var_w,
                       m_w, v_w, var_b, m_b, v_b, m, k, n, 1.f - beta1, grad_scale);
  else
    hipLaunchKernelGGL((dense_dw_kernel<true, 4>), dim3((n + 127) / 128, (k + 127) / 128), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, dz, nullptr, nullptr, var_w, m_w, v_w, var_b, m_b, v_b, m, k, n,
                       1.f - beta1, grad_scale);
  return check_launch("dense_dw_adam");
}
