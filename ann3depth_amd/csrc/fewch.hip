// fewch.hip — Conv2DBackpropFilter (+ BiasAddGrad) of the few-channel layers from LDS-staged input rows: conv2d_0
// (src/models.py:211: 11x11 stride 4 on a 3-channel image), fine/first (:241: 9x9 stride 2) and DCNF's first conv
// (:64: 11x11 stride 1).  dw[(r,s,c)][k] = sum over pixels x[oy*st + r][ox*st + s][c] * dz[oy][ox][k] is a GEMM with
// M = R*S*C (243 / 363), N = k (63 / 64 / 96) and the PIXELS as its contraction axis; the generic kernel gathers its A
// operand (im2col(x) transposed) with 8-byte window runs, a row-table lookup and a bound test per piece — 4 vector and 3
// scalar instructions per MFMA, matrix pipe half idle (profiles/r04_pmc_mfma_busy.json).  Here:
//   * a block owns 128 rows of M (four waves x one 32-row tile x all of N) and a contiguous range of output rows;
//   * per output row the few input rows its filter rows touch (<= 6 x W*C floats) and the row of dz go to LDS once, as
//     16-byte pieces; lane (m, half) of v_mfma_f32_32x32x2_f32 then reads its A element for the pixel pair (2kp, 2kp+1) at
//     `lane constant + kp * 2*stride*C` — ONE ds_read_b32 and one add per MFMA group, no decode, no bound test (VALID
//     convolutions: every tap of every output pixel lies inside the image);
//   * BiasAddGrad rides in the GEMM: row M of the padded tile reads a constant 1.0, so dw[M][k] = sum dz[.][k];
//   * the pool in front of these layers' gradients is fused into the staging of dz (SRC_POOLED): the row of dz is built
//     from the POOLED gradient, the argmax byte and the sign of the pooled activation (MaxPoolGrad + ReluGrad,
//     src/models.py:213,243,65), so the full-resolution gradient (131 MB for fine/first at B = 32, 1.6 GB for DCNF) is
//     never written or read;
//   * the pixel axis is split over blocks; partial tiles go to slabs and one reduction adds them in split order (the
//     same bits on every run).
#include <algorithm>

#include "a3d_internal.h"
#include "igemm.h"

namespace a3d {

enum { FEW_SRC_DZ = 0, FEW_SRC_POOLED = 1, FEW_SRC_POOLED_BF16 = 2 };
constexpr int kFewRows = 6;          // input rows one 128-row group of M can touch (128 / (S*C) + 2 for S*C >= 27)

struct FewchParams {
  const float* x;          // [n, h, w, c], pixels densely packed
  const void* dz;          // FEW_SRC_DZ: [n, ho, wo, ldz] float32; pooled: the pooled gradient [n, ho/2, wo/2, ldz]
  const void* pooled;      // pooled sources: the pooled activation (ReluGrad: > 0), same layout as dz; null = no mask
  const uint8_t* argmax;   // pooled sources: [n, ho/2, wo/2, ld_arg] window positions 0..3
  float* slabs;            // [splits][Mp][NP]
  int n, h, w, c, R, S, stride, ho, wo, N, M, Mp, NP;
  int ldz, ld_arg;
  int rows_per_img, rows_total, splits, mgroups;
  int xpitch, rowlen, kpairs, wo_pad;
  int xvec4, dvec4;
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kPX = 6;               // 16-byte pieces of input rows per thread and output row (6 rows x 228 pieces / 256 threads)
constexpr int kPDPlain = 10;         // ... of the dz row (148 pixels x 16 pieces)
constexpr int kPDPooled = 5;         // 4-channel groups of the pooled row (74 windows x 16 groups), three registers each

#ifdef A3D_STAMPS
__device__ unsigned long long g_fewch_stamps[1024 * 4 * 8];       // diagnostic build (never shipped): [block][wave][8] phase cycle sums
#endif
// STEP: stride * C of the layer as a compile-time constant (12 / 6 / 3 for conv2d_0 / fine/first / DCNF's first conv), 0 = read
// it from the descriptor.  With it — and with the tile's row length NP = 32 * TN — every operand read of the MFMA loop is
// `lane constant + immediate`: beside fp32 MFMAs a v_add per read is paid in full, and there were 16 per 12 MFMAs.
template <int TN, int SRC, bool VEC, int STEP = 0>
__global__ __launch_bounds__(256, 2) void fewch_bwdf_kernel(const FewchParams p) {
  constexpr int kPD = SRC == FEW_SRC_DZ ? kPDPlain : kPDPooled;
  constexpr int NP = 32 * TN;                         // == p.NP
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                                   // [kFewRows][xpitch]
  float* dzs = xs + kFewRows * p.xpitch;              // [wo_pad][NP]
  float* consts = dzs + p.wo_pad * NP;                // [2][xpitch]: a row of 1.0 (the bias tap's "input row") and a row of 0.0 (taps past M)
  // every LDS write of the staging is unconditional: a piece a thread does not have goes to its own 16 bytes of this scrap area
  // (a pooled source's second piece: NP floats further on, still inside it).  A guarded write is a basic block of its own and
  // its lane mask a pair of scalar registers for the whole loop: the kernel spilled 50-180 of them into vector lanes and read
  // them back one v_readlane at a time.
  const int scrap = (int)(consts - xs) + 2 * p.xpitch + threadIdx.x * 4;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
#ifdef A3D_STAMPS
  unsigned long long t_entry = 0;
  A3D_STAMP(t_entry);
#endif
  // XCD-aware order: the blocks an XCD receives (ids congruent mod 8) are consecutive tasks, so the m-groups of one pixel range —
  // which stage the same input rows and the same row of dz — share an L2 (the counters had 199 MB of fabric reads per launch
  // for conv2d_0 against ~80 MB algorithmic with the m-groups dealt round-robin over the XCDs)
  uint32_t bid = blockIdx.x;
  {
    const uint32_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = (int)bid / p.mgroups, mg = (int)bid - split * p.mgroups;
  const int SC = p.S * p.c;
  const int rlo = (mg * 128) / SC;
  const int rhi = min(p.M - 1, mg * 128 + 127) / SC;
  const int nr = rhi - rlo + 1;
  // ---- this lane's A element: row m of the tile, pixel parity lh
  const int m = mg * 128 + wv * 32 + li;
  // (every lane walks its row with the same step: the bias tap and the taps past M walk rows of constants)
  const int stC = STEP ? STEP : p.stride * p.c;
  const int a_step = 2 * stC;
  int a_off;
  if (m < p.M) {
    const int r = m / SC, j = m - r * SC;
    a_off = (r - rlo) * p.xpitch + j + lh * stC;
  } else {
    a_off = (int)(consts - xs) + (m == p.M ? 0 : p.xpitch);  // the bias row reads 1.0, the rest of the padding 0.0
  }
  // zero what the staging never writes: the tails of the input rows (read against dz = 0 when wo is odd) and the columns
  // N..NP / the pad pixel of the dz row
  for (int i = tid; i < kFewRows * p.xpitch; i += 256) xs[i] = 0.f;
  for (int i = tid; i < p.wo_pad * NP; i += 256) dzs[i] = 0.f;
  for (int i = tid; i < p.xpitch; i += 256) { consts[i] = 1.f; consts[p.xpitch + i] = 0.f; }
  f32x16 acc[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[b][v] = 0.f;

  // ---- staging in two halves: fetch() puts a row's operands into registers (issued behind the first MFMAs of the previous
  //      row), commit() writes them to LDS behind the barrier that ends those MFMAs.  Every load is a raw buffer load at
  //      `row base + per-thread constant`; a piece the thread does not have carries the out-of-range offset (zeros, no branch).
  const int r4 = p.rowlen >> 2;
  const int xtotal = nr * r4;                          // 16-byte pieces of the input rows
  const int n4 = (p.N + 3) >> 2;
  const int pw = p.wo >> 1, ph = p.ho >> 1;
  const int dtotal = SRC == FEW_SRC_DZ ? p.wo * n4 : pw * n4;
  constexpr int ESZ = SRC == FEW_SRC_POOLED_BF16 ? 2 : 4;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (unsigned long long)p.n * p.h * p.rowlen * 4ull);
  const unsigned long long dz_elems = SRC == FEW_SRC_DZ ? (unsigned long long)p.n * p.ho * p.wo * p.ldz
                                                         : (unsigned long long)p.n * ph * pw * p.ldz;
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(static_cast<const float*>(p.dz), dz_elems * ESZ);
  const __amdgpu_buffer_rsrc_t rp = make_rsrc(static_cast<const float*>(p.pooled ? p.pooled : p.dz), dz_elems * ESZ);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(reinterpret_cast<const float*>(p.argmax),
                                              SRC == FEW_SRC_DZ ? 0ull : (unsigned long long)p.n * ph * pw * p.ld_arg);
  int xdst[kPX];
  uint32_t xoff[kPX];
#pragma unroll
  for (int i = 0; i < kPX; ++i) {
    const int e = tid + i * 256, rr = e / r4, q = e - rr * r4;
    xdst[i] = e < xtotal ? rr * p.xpitch + 4 * q : scrap;
    xoff[i] = e < xtotal ? (uint32_t)((rr * p.rowlen + 4 * q) * 4) : kOOB;
  }
  int ddst[kPD];                                       // LDS float index of the piece (pooled: of its even pixel; the odd one NP floats on)
  uint32_t doff[kPD], aoff[kPD];                       // byte offsets into dz (or dpool / pooled) and argmax (none: 2^31, outside every buffer)
#pragma unroll
  for (int i = 0; i < kPD; ++i) {
    const int e = tid + i * 256, px = e / n4, q = 4 * (e - px * n4);
    const bool ok = e < dtotal;
    ddst[i] = ok ? (SRC == FEW_SRC_DZ ? px : 2 * px) * NP + q : scrap - kFewRows * p.xpitch;      // (float index from dzs)
    doff[i] = ok ? (uint32_t)((px * p.ldz + q) * ESZ) : kOOB;
    aoff[i] = ok ? (uint32_t)(px * p.ld_arg + q) : kOOB;
  }
  f32x4 xv[kPX], dv[kPD], av[kPD];
  uint32_t argv[kPD];
  auto load4 = [&](const __amdgpu_buffer_rsrc_t r, uint32_t off) -> f32x4 {
    if constexpr (SRC == FEW_SRC_POOLED_BF16) {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      const bf16x4 g = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0));
      return f32x4{(float)g[0], (float)g[1], (float)g[2], (float)g[3]};
    } else {
      return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    }
  };
  auto fetch = [&](int row) {
    const int img = row / p.rows_per_img, oy = row - img * p.rows_per_img;
    const uint32_t xbase = (uint32_t)((((size_t)img * p.h + (size_t)oy * p.stride + rlo) * p.rowlen) * 4);
#pragma unroll
    for (int i = 0; i < kPX; ++i)
      if (i * 256 < xtotal) xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(xbase + xoff[i]), 0, 0));
    if constexpr (SRC == FEW_SRC_DZ) {
      const uint32_t dbase = (uint32_t)((((size_t)img * p.ho + oy) * p.wo * p.ldz) * 4);
#pragma unroll
      for (int i = 0; i < kPD; ++i)
        if (i * 256 < dtotal) {
          if constexpr (VEC) {
            dv[i] = load4(rd, dbase + doff[i]);
          } else {                                     // pixel strides / filter counts off the 16-byte grid: element by element
#pragma unroll
            for (int e = 0; e < 4; ++e)
              dv[i][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rd, (int)(dbase + doff[i] + 4 * e), 0, 0));      // (past N: the next pixel's, into a column nobody reads)
          }
        }
    } else {
      const size_t prow = ((size_t)img * ph + (oy >> 1)) * pw;
      const uint32_t dbase = (uint32_t)(prow * p.ldz * ESZ), abase = (uint32_t)(prow * p.ld_arg);
#pragma unroll
      for (int i = 0; i < kPD; ++i)
        if (i * 256 < dtotal) {
          dv[i] = load4(rd, dbase + doff[i]);
          if (p.pooled) av[i] = load4(rp, dbase + doff[i]);
          if constexpr (VEC) {                         // (here: argmax rows of whole, aligned 4-byte groups)
            argv[i] = __builtin_amdgcn_raw_buffer_load_b32(ra, (int)(abase + aoff[i]), 0, 0);
          } else {
            uint32_t w = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e)
              w |= (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(ra, (int)(abase + aoff[i] + e), 0, 0) << (8 * e);
            argv[i] = w;
          }
        }
    }
  };
#ifdef A3D_STAMPS
  unsigned long long s_mid = 0, d_xw = 0, s_ld = 0, d_ld = 0;
#endif
  auto commit = [&](int row) {
#pragma unroll
    for (int i = 0; i < kPX; ++i)
      if (i * 256 < xtotal) *reinterpret_cast<f32x4*>(xs + xdst[i]) = xv[i];
#ifdef A3D_STAMPS
    A3D_STAMP(s_mid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    A3D_STAMP(s_ld);
#endif
    if constexpr (SRC == FEW_SRC_DZ) {
#pragma unroll
      for (int i = 0; i < kPD; ++i)
        if (i * 256 < dtotal) *reinterpret_cast<f32x4*>(dzs + ddst[i]) = dv[i];
    } else {
      // MaxPoolGrad + ReluGrad: window (oy/2, px) hands its gradient to position argmax, if the maximum was > 0
      const int oy = row % p.rows_per_img;
      const uint32_t want = (uint32_t)(oy & 1) * 2u;
#pragma unroll
      for (int i = 0; i < kPD; ++i)
        if (i * 256 < dtotal) {
          // (no test against N: a channel past N of the last group lands in a column >= N of the tile, whose sums the
          // reduction never reads; where the group is loaded element by element such a channel is the next pixel's)
          // on bit masks instead of compare / select pairs (each of those is a VCC round trip with its wait states): byte e of
          // argmax == want <=> byte e of t is zero, flagged exactly in bit 7 of the byte by the carry trick
          const uint32_t t = argv[i] ^ (want * 0x01010101u), t1 = t ^ 0x01010101u;
          const uint32_t zl = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu);
          const uint32_t zh = ~(((t1 & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t1 | 0x7f7f7f7fu);
          f32x4 lo, hi;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const uint32_t pos = (!p.pooled || av[i][e] > 0.f) ? 0xffffffffu : 0u;
            const uint32_t g = __float_as_uint(dv[i][e]) & pos;
            lo[e] = __uint_as_float(g & (uint32_t)__builtin_amdgcn_sbfe((int)zl, 8 * e + 7, 1));
            hi[e] = __uint_as_float(g & (uint32_t)__builtin_amdgcn_sbfe((int)zh, 8 * e + 7, 1));
          }
          *reinterpret_cast<f32x4*>(dzs + ddst[i]) = lo;
          *reinterpret_cast<f32x4*>(dzs + ddst[i] + NP) = hi;
        }
    }
  };

  const int row_lo = (int)((long)split * p.rows_total / p.splits), row_hi = (int)((long)(split + 1) * p.rows_total / p.splits);
#ifdef A3D_STAMPS
  unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, d_bar1 = 0, d_commit = 0, d_bar2 = 0, d_mfma = 0;
#endif
  if (row_lo < row_hi) fetch(row_lo);
  for (int row = row_lo; row < row_hi; ++row) {
    A3D_STAMP(s0);
    __syncthreads();                                  // the previous row's MFMAs have read their operands
    A3D_STAMP(s1);
    commit(row);
    A3D_STAMP(s2);
    __syncthreads();
    A3D_STAMP(s3);
    // ---- 2 pixels per MFMA: A = x at the lane's tap for pixel 2kp + lh, B = dz[2kp + lh][column li]; groups of four
    //      pixel pairs, the next group's operands read from LDS while this group's MFMAs run; the NEXT row's global loads
    //      are issued behind the first group's MFMAs
    const float* ap = xs + a_off;
    const float* bp = dzs + lh * NP + li;
    const int groups = p.kpairs >> 2;
    const bool more = row + 1 < row_hi;
    float a[2][4], b[2][4][TN];
    auto read = [&](int buf) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[buf][u] = ap[u * a_step];
#pragma unroll
        for (int t = 0; t < TN; ++t) b[buf][u][t] = bp[u * 2 * NP + t * 32];
      }
      ap += 4 * a_step;
      bp += 8 * NP;
    };
    auto mul = [&](int buf) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[buf][u], b[buf][u][t], acc[t], 0, 0, 0);
    };
    int g = 0;
    if (groups >= 2) {
      read(0);
      read(1);
      mul(0);
      if (more) fetch(row + 1);
      if (groups > 2) read(0);
      mul(1);
      for (g = 2; g + 2 <= groups; g += 2) {
        read(1);
        mul(0);
        if (g + 2 < groups) read(0);
        mul(1);
      }
      if (g < groups) mul(0);
      g = groups;
    } else if (more) {
      fetch(row + 1);
    }
    for (int kp = g * 4; kp < p.kpairs; ++kp) {
      const float av1 = ap[0];
#pragma unroll
      for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bp[t * 32], acc[t], 0, 0, 0);
      ap += a_step;
      bp += 2 * NP;
    }
#ifdef A3D_STAMPS
    A3D_STAMP(s4);
    d_ld += s_ld - s_mid; d_xw += s_mid - s1; d_bar1 += s1 - s0; d_commit += s2 - s1; d_bar2 += s3 - s2; d_mfma += s4 - s3;
#endif
  }
#ifdef A3D_STAMPS
  if (lane == 0 && blockIdx.x < 1024) {
    unsigned long long t_end;
    A3D_STAMP(t_end);
    unsigned long long* o = g_fewch_stamps + ((size_t)blockIdx.x * 4 + wv) * 8;
    o[0] = d_bar1; o[1] = d_commit; o[2] = d_bar2; o[3] = d_mfma; o[4] = (unsigned long long)(row_hi - row_lo); o[5] = d_xw; o[6] = t_end;
    o[7] = d_ld;      // (slot 7: waiting for the gradient row's loads)
  }
  unsigned long long t_end2 = 0;
  A3D_STAMP(t_end2);
#endif
  // ---- the block's partial tile -> its slab (register 4g+i of a lane: row 8g + 4*lh + i, column li)
  float* slab = p.slabs + ((size_t)split * p.Mp + mg * 128 + wv * 32) * NP;
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) slab[(size_t)(8 * g + 4 * lh + i) * NP + t * 32 + li] = acc[t][4 * g + i];
#ifdef A3D_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t_exit = 0;
  A3D_STAMP(t_exit);
  if (lane == 0 && blockIdx.x < 1024) {
    unsigned long long* o = g_fewch_stamps + ((size_t)blockIdx.x * 4 + wv) * 8;
    o[5] = t_exit - t_end2;      // (slot 5 reused: the epilogue's slab stores)
    o[6] = t_end2 - t_entry;     // (slot 6 reused: entry -> loop end)
  }
#endif
}

// dw[m][n] (m < M), db[n] (row M) = sum over splits, in split order
__global__ __launch_bounds__(256) void fewch_reduce_kernel(const float* __restrict__ slabs, int splits, int Mp, int NP, int M, int N,
                                                           float* __restrict__ dw, float* __restrict__ db) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= (M + 1) * N) return;
  const int m = idx / N, n = idx - m * N;
  const float* s = slabs + (size_t)m * NP + n;
  const size_t pitch = (size_t)Mp * NP;
  float a0 = 0.f;
  int k = 0;
  for (; k + 8 <= splits; k += 8) {                   // eight loads in flight, added in split order
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = s[(size_t)(k + u) * pitch];
#pragma unroll
    for (int u = 0; u < 8; ++u) a0 += v[u];
  }
  for (; k < splits; ++k) a0 += s[(size_t)k * pitch];
  if (m < M) dw[(size_t)m * N + n] = a0;
  else if (db) db[n] = a0;
}

// ---- host side ----
struct FewchShape {
  int M, Mp, mgroups, TN, NP, rows_per_img, rows_total, splits, xpitch, wo_pad, kpairs;
  size_t lds;
};

static FewchShape fewch_shape(const a3d_conv_desc* d, bool pooled) {
  FewchShape s;
  s.M = d->r * d->s * d->c;
  s.mgroups = (s.M + 1 + 127) / 128;                  // + the bias row
  s.Mp = s.mgroups * 128;
  s.TN = (d->k + 31) / 32;
  s.NP = s.TN * 32;
  s.rows_per_img = pooled ? (d->ho / 2) * 2 : d->ho;  // (a last odd row has no pool window: its gradient is zero)
  s.rows_total = d->n * s.rows_per_img;
  s.wo_pad = (d->wo + 1) / 2 * 2;
  s.kpairs = s.wo_pad / 2;
  const int reach = (s.wo_pad - 1) * d->stride * d->c + d->s * d->c;
  s.xpitch = (std::max(d->w * d->c, reach) + 3) / 4 * 4;
  // two resident blocks per CU; a block needs a few rows to amortise its slab (Mp x NP floats)
  s.splits = std::max(1, std::min(s.rows_total / 4, tune_int("A3D_FEWCH_BLOCKS", 512) / s.mgroups));
  s.lds = (size_t)((kFewRows + 2) * s.xpitch + s.wo_pad * s.NP + 256 * 4 + s.NP + 4) * 4;      // rows of x, row of dz, two constant rows, scrap
  return s;
}

bool fewch_bwdf_applicable(const a3d_conv_desc* d, bool pooled) {
  if (d->precision != A3D_PREC_F32 || d->c > 4 || d->pad_t || d->pad_l || d->ldx != d->c) return false;
  if (d->k < 33 || d->k > 96) return false;
  if (d->s * d->c < 27) return false;                 // kFewRows input rows per 128 rows of M
  if ((d->ho - 1) * d->stride + d->r > d->h || (d->wo - 1) * d->stride + d->s > d->w) return false;   // VALID geometry
  if (pooled && (d->ho < 2 || d->wo < 2)) return false;
  if ((d->w * d->c) % 4 != 0) return false;           // input rows as whole 16-byte pieces
  // one buffer descriptor per whole tensor, 31-bit byte offsets (kOOB = 2^31 marks a piece as out of range): x here, the
  // gradient tensors with their real pixel strides in fewch_extents_ok (the callers: igemm_host.hip)
  if ((double)d->n * d->h * d->w * d->c * 4.0 >= 2147483647.0) return false;
  const FewchShape s = fewch_shape(d, pooled);
  if (s.rows_total < 1 || s.lds > 78 * 1024) return false;
  // the per-thread staging registers: kPX pieces of input rows, kPD pieces of the dz row (or 4-channel groups of the pooled row)
  if (kFewRows * (d->w * d->c / 4) > kPX * 256) return false;
  if ((pooled ? d->wo / 2 : d->wo) * ((d->k + 3) / 4) > (pooled ? kPDPooled : kPDPlain) * 256) return false;
  return true;
}

// The gradient-side tensors of a launch against the kernels' 31-bit byte offsets: dz (plain source: n x ho x wo pixels of ldz
// elements of esz bytes) or the pooled gradient and the pooled activations (n x ho/2 x wo/2 pixels of ldz elements) with their
// argmax bytes (ld_arg per pixel).  Shared by fewch16.hip (bf16 gradient tensors: esz 2).
bool fewch_extents_ok(const a3d_conv_desc* d, bool pooled, int ldz, int esz, int ld_arg) {
  const double pixels = pooled ? (double)d->n * (d->ho / 2) * (d->wo / 2) : (double)d->n * d->ho * d->wo;
  return pixels * ldz * esz < 2147483647.0 && (!pooled || pixels * ld_arg < 2147483647.0);
}

size_t fewch_bwdf_ws_bytes(const a3d_conv_desc* d, bool pooled) {
  const FewchShape s = fewch_shape(d, pooled);
  return (size_t)s.splits * s.Mp * s.NP * 4 + 16;
}

template <int TN, int SRC, bool VEC, int STEP>
static void fewch_launch4(const FewchParams& p, int blocks, size_t lds, hipStream_t st) {
  // above 64 KiB of dynamic LDS is possible: the attribute is per device and cheap, set on every call
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fewch_bwdf_kernel<TN, SRC, VEC, STEP>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            80 * 1024);
  hipLaunchKernelGGL((fewch_bwdf_kernel<TN, SRC, VEC, STEP>), dim3(blocks), dim3(256), lds, st, p);
}
template <int TN, int SRC, bool VEC>
static void fewch_launch3(const FewchParams& p, int blocks, size_t lds, hipStream_t st) {
  // the step of the three layers this kernel was built for (with their tile widths) as compile-time constants; anything else reads it
  const int step = p.stride * p.c;
  if (TN == 3 && step == 12) return fewch_launch4<TN, SRC, VEC, TN == 3 ? 12 : 0>(p, blocks, lds, st);      // conv2d_0
  if (TN == 2 && step == 6) return fewch_launch4<TN, SRC, VEC, TN == 2 ? 6 : 0>(p, blocks, lds, st);        // fine/first
  if (TN == 2 && step == 3) return fewch_launch4<TN, SRC, VEC, TN == 2 ? 3 : 0>(p, blocks, lds, st);        // DCNF's first conv
  fewch_launch4<TN, SRC, VEC, 0>(p, blocks, lds, st);
}
template <int TN>
static void fewch_launch(int src, const FewchParams& p, int blocks, size_t lds, hipStream_t st) {
  if (src == FEW_SRC_DZ) p.dvec4 ? fewch_launch3<TN, FEW_SRC_DZ, true>(p, blocks, lds, st) : fewch_launch3<TN, FEW_SRC_DZ, false>(p, blocks, lds, st);
  else if (src == FEW_SRC_POOLED) p.dvec4 ? fewch_launch3<TN, FEW_SRC_POOLED, true>(p, blocks, lds, st) : fewch_launch3<TN, FEW_SRC_POOLED, false>(p, blocks, lds, st);
  else p.dvec4 ? fewch_launch3<TN, FEW_SRC_POOLED_BF16, true>(p, blocks, lds, st) : fewch_launch3<TN, FEW_SRC_POOLED_BF16, false>(p, blocks, lds, st);
}

// src: FEW_SRC_*; dz / ldz: the gradient tensor (or the pooled gradient) and its pixel stride
int fewch_bwd_filter(const a3d_conv_desc* d, const float* x, int src, const void* dz, int ldz, const void* pooled_act,
                     const uint8_t* argmax, int ld_arg, float* dw, float* db, void* ws, hipStream_t st) {
  const bool pooled = src != FEW_SRC_DZ;
  const FewchShape s = fewch_shape(d, pooled);
  FewchParams p{};
  p.x = x; p.dz = dz; p.pooled = pooled_act; p.argmax = argmax;
  p.slabs = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 15) & ~(uintptr_t)15);
  p.n = d->n; p.h = d->h; p.w = d->w; p.c = d->c; p.R = d->r; p.S = d->s; p.stride = d->stride; p.ho = d->ho; p.wo = d->wo;
  p.N = d->k; p.M = s.M; p.Mp = s.Mp; p.NP = s.NP; p.ldz = ldz; p.ld_arg = ld_arg;
  p.rows_per_img = s.rows_per_img; p.rows_total = s.rows_total; p.splits = s.splits; p.mgroups = s.mgroups;
  p.xpitch = s.xpitch; p.rowlen = d->w * d->c; p.kpairs = s.kpairs; p.wo_pad = s.wo_pad;
  p.xvec4 = p.rowlen % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  // whole aligned 16-byte pieces of dz (plain source) / 4-byte groups of argmax bytes (pooled sources)
  p.dvec4 = pooled ? (d->k % 4 == 0 && ld_arg % 4 == 0 && (reinterpret_cast<uintptr_t>(argmax) & 3) == 0)
                   : (d->k % 4 == 0 && ldz % 4 == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0);
  const int blocks = s.splits * s.mgroups;
  clear_stale_error();
  if (s.TN == 3) fewch_launch<3>(src, p, blocks, s.lds, st);
  else fewch_launch<2>(src, p, blocks, s.lds, st);
  int rc = check_launch("fewch_bwd_filter");
  if (rc != A3D_OK) return rc;
  return fewch_reduce_launch(p.slabs, s.splits, s.Mp, s.NP, s.M, d->k, dw, db, st);
}

#ifdef A3D_STAMPS
extern "C" int a3d_debug_fewch_stamps(unsigned long long* out, size_t bytes) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fewch_stamps), std::min(bytes, sizeof(g_fewch_stamps)));
}
#endif

int fewch_reduce_launch(const float* slabs, int splits, int Mp, int NP, int M, int N, float* dw, float* db, hipStream_t st) {
  const int outs = (M + 1) * N;
  clear_stale_error();
  hipLaunchKernelGGL(fewch_reduce_kernel, dim3((outs + 255) / 256), dim3(256), 0, st, slabs, splits, Mp, NP, M, N, dw, db);
  return check_launch("fewch_reduce");
}

}  // namespace a3d
