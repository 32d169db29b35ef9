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

namespace a3d {

typedef float f32x16 __attribute__((ext_vector_type(16)));

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
// m slot is read and written as WHOLE 2-KiB row segments (1 KiB contiguous per wave-instruction) instead of 512-byte
// pieces 16 KB apart: the read-modify-write stream of a 201 MB slot is what this kernel costs, and HBM serves long runs
// far better (measured on dense_0: 426 us with 256-byte pieces, 100 us with 512-byte pieces).  Needs N % 4 == 0.
constexpr int kRowsLd = 512 + 4;
__global__ __launch_bounds__(256) void dense_dw_adam_rows_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                                 float* __restrict__ var_w, float* __restrict__ m_w,
                                                                 float* __restrict__ v_w, float* __restrict__ var_b,
                                                                 float* __restrict__ m_b, float* __restrict__ v_b, int M,
                                                                 int K, int N, float omb1, float gscale) {
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) float tile[32 * kRowsLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int nb = blockIdx.x * 512, n0 = nb + wave * 128, k0 = blockIdx.y * 32;
  const bool use_scale = gscale != 1.f;
  const float qnan = __builtin_nanf("");

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

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
  const int krow = k0 + li, col0 = n0 + 4 * li;
  const bool kok = krow < K, cok = col0 < N;       // N % 4 == 0: a lane's four columns exist together
  const int T = (M + 1) / 2;
  for (int t0 = 0; t0 < T; t0 += 8) {
    float a[8];
    f32x4 bq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int m = 2 * (t0 + u) + lh;
      const bool mok = m < M;
      a[u] = (mok && kok) ? x[(size_t)m * K + krow] : 0.f;
      bq[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (mok && cok) bq[u] = *reinterpret_cast<const f32x4*>(dz + (size_t)m * N + col0);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bq[u][j], acc[j], 0, 0, 0);
  }
  // gradient tile -> LDS: register v of the four accumulators = columns 128*wave + 4*li .. +3 of row (v&3)+8*(v>>2)+4*lh
#pragma unroll
  for (int v = 0; v < 16; ++v)
    *reinterpret_cast<f32x4*>(&tile[((v & 3) + 8 * (v >> 2) + 4 * lh) * kRowsLd + wave * 128 + 4 * li]) =
        (f32x4){acc[0][v], acc[1][v], acc[2][v], acc[3][v]};
  // this wave's share of the m tile — rows 8*wave .. 8*wave+7, two 1-KiB halves each — requested once the accumulators
  // are in LDS: the kernel then never holds both (under 128 registers: four waves per SIMD hide the HBM latency, and the
  // waves fit beside the GEMM blocks of the fine network that run on the second stream at the same time)
  f32x4 mold[16];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = k0 + wave * 8 + r, col = nb + h * 256 + lane * 4;
      mold[2 * r + h] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (row < K && col < N) mold[2 * r + h] = *reinterpret_cast<const f32x4*>(m_w + (size_t)row * N + col);
    }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int lr = wave * 8 + r, row = k0 + lr, c = h * 256 + lane * 4, col = nb + c;
      if (row >= K || col >= N) continue;
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(&tile[lr * kRowsLd + c]);
      const size_t o = (size_t)row * N + col;
      f32x4 mn;
      float chk = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = use_scale ? __fmul_rn(g4[j], gscale) : g4[j];
        mn[j] = __fadd_rn(mold[2 * r + h][j], __fmul_rn(__fsub_rn(g, mold[2 * r + h][j]), omb1));
        chk += __fmul_rn(g, g) + fabsf(mn[j]);
      }
      *reinterpret_cast<f32x4*>(m_w + o) = mn;
      if (!isfinite(chk)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g = use_scale ? __fmul_rn(g4[j], gscale) : g4[j];
          const bool poison = !isfinite(__fmul_rn(g, g));
          if (poison) v_w[o + j] = qnan;
          if (poison || !isfinite(mn[j])) var_w[o + j] = qnan;
        }
      }
    }
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

extern "C" int a3d_dense_bwd_filter_adam_tf1(int m, int k, int n, const float* x, const float* dz, float* var_w, float* m_w,
                                             float* v_w, float* var_b, float* m_b, float* v_b, float lr, float beta1,
                                             float beta2, float beta1_power, float beta2_power, float grad_scale,
                                             void* stream) {
  A3D_CHECK_ARG(m > 0 && k > 0 && n > 0 && x && dz && var_w && m_w && v_w, "dense_bwd_filter_adam: bad arguments");
  A3D_CHECK_ARG((var_b && m_b && v_b) || (!var_b && !m_b && !v_b), "dense_bwd_filter_adam: bias slots come as a set");
  A3D_CHECK_ARG(dense_dw_applicable(m, k, n), "dense_bwd_filter_adam: batches of at most 64 rows only");
  const float alpha = lr * sqrtf(1.f - beta2_power) / (1.f - beta1_power);
  A3D_CHECK_ARG(alpha == 0.f && 1.f - beta2 == 0.f,
                "dense_bwd_filter_adam: only the reference's frozen optimizer (beta2 == 1); otherwise call "
                "a3d_dense_bwd_filter and a3d_adam_apply_tf1");
  clear_stale_error();
  static const int cw = getenv("A3D_DW_CW") ? atoi(getenv("A3D_DW_CW")) : 0;      // tuning aid
  if (cw != 4 && cw != 2 && n % 4 == 0 && (reinterpret_cast<uintptr_t>(m_w) & 15) == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0)
    hipLaunchKernelGGL(dense_dw_adam_rows_kernel, dim3((n + 511) / 512, (k + 31) / 32), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, dz, var_w, m_w, v_w, var_b, m_b, v_b, m, k, n, 1.f - beta1,
                       grad_scale);
  else if (cw != 2)
    hipLaunchKernelGGL((dense_dw_kernel<true, 4>), dim3((n + 127) / 128, (k + 127) / 128), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, dz, nullptr, nullptr, var_w, m_w, v_w, var_b, m_b, v_b, m, k, n,
                       1.f - beta1, grad_scale);
  else
    hipLaunchKernelGGL((dense_dw_kernel<true, 2>), dim3((n + 63) / 64, (k + 127) / 128), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, dz, nullptr, nullptr, var_w, m_w, v_w, var_b, m_b, v_b, m, k, n,
                       1.f - beta1, grad_scale);
  return check_launch("dense_dw_adam");
}
