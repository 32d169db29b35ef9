// Third prototype of the gen-2 fp32 GEMM loop (see gemm2.hip / gemm3.hip): LDS-DMA staging, fenced schedule, and now both
// LDS layouts per operand, the way the three conv modes will need them:
//   KC ("k-contiguous"): tile [128 rows][32 k], 128-byte rows, 16-byte chunk c of row r at chunk position c ^ ((r >> 1) & 7),
//       fragments by ds_read_b128 (one read = four k-pairs of a 32-row group).        Operand given as [rows][K].
//   MC ("row-major in k"): tile [32 k][128 rows]; a DMA piece = two k rows (1 KiB) and pieces lie 1088 bytes apart, so the
//       k rows of the two lane halves (k, k + 4 = two pieces on) fall into opposite bank halves; fragments by ds_read2_b32
//       (one read = rows li and 32 + li of ONE k).                                      Operand given as [K][rows].
//   forward = A KC, B MC;  bwd-data = KC, KC;  bwd-filter = MC, MC.
// DM 0: one M0 write per DMA piece, pieces issued between the MFMA quads of chunks 0 and 1.
// DM 1: a wave owns four CONSECUTIVE pieces of a tile: one M0 write, the pieces told apart by the instruction's immediate
//       offset (added to the LDS address AND the memory address: the lane offsets carry the negated immediate).
// ABL bit 0: no DMA in the loop, bit 1: no fragment reads in the loop (ablations: results meaningless).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/gemm4.hip -o tools/micro/bin/gemm4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include <type_traits>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ u32x4 rsrc_of(const void* base, unsigned long long bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base);
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes > 0x7fffffffull ? 0x7fffffffu : (unsigned)bytes);
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ void dma16(u32x4 rs, unsigned voff, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
               : "memory");
}
// four pieces behind one M0 write; piece i goes to lds_addr + i * STEP and reads lane offset voff[i] + i * STEP (the caller
// has subtracted i * STEP from voff[i])
template <int STEP>
__device__ __forceinline__ void dma16x4(u32x4 rs, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned soff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 2\n\t"
               "buffer_load_dwordx4 %2, %6, %7 offen lds\n\t"
               "buffer_load_dwordx4 %3, %6, %7 offen offset:%c8 lds\n\t"
               "buffer_load_dwordx4 %4, %6, %7 offen offset:%c9 lds\n\t"
               "buffer_load_dwordx4 %5, %6, %7 offen offset:%c10 lds\n\t"
               "s_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_addr), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(rs), "s"(soff), "i"(STEP), "i"(2 * STEP), "i"(3 * STEP)
               : "memory");
}
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ unsigned long long realtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ unsigned remap_xcd(unsigned bid, unsigned nwg) {
  const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

constexpr int KC = 0, MC = 1, MX = 2;      // MX: as MC, but unpadded 1-KiB pieces; the column chunks of k rows with (k >> 2) & 1 XOR 8 (= columns XOR 32), reads by ds_read2st64_b32
constexpr int MC_PIECE = 1088;                                  // bytes between the two-row pieces of an MC tile
constexpr int tile_bytes(int lay) { return lay == MC ? 16 * MC_PIECE : 128 * 128; }

template <int NS, int AL, int BL, int DM, int ABL, bool STAMPS, int MINB, int EPI = 0>
__global__ __launch_bounds__(256, MINB) void g4(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                int M, int N, int K, unsigned long long* stamps) {
  constexpr int BM = 128, BN = 128, BK = 32;
  constexpr int A_BYTES = tile_bytes(AL), B_BYTES = tile_bytes(BL), STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int tiles_n = N / BN;
  const unsigned bid = remap_xcd(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = K / BK;
  unsigned long long rt_entry = 0;
  if (STAMPS) rt_entry = realtime();

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  // ---- LDS-DMA.  Piece p of an operand tile, p = 0..15; this wave issues pieces pc(j), j = 0..3. ----
  auto pc = [&](int j) { return DM == 1 ? 4 * wave + j : 4 * j + wave; };
  auto rs_for = [&](const float* P, int lay, int r0, int R) {       // operand with R rows in all, block rows r0 .. r0 + 127
    return lay == KC ? rsrc_of(P + (size_t)r0 * K, (unsigned long long)128 * K * 4) : rsrc_of(P + r0, ((unsigned long long)K * R - r0) * 4);
  };
  const u32x4 rsA = rs_for(A, AL, m0, M), rsB = rs_for(B, BL, n0, N);
  auto voff_for = [&](int lay, int j, int R) -> unsigned {
    const int p = pc(j);
    if (lay == KC) {
      const int row = 8 * p + (lane >> 3);
      return (unsigned)((row * K + 4 * ((lane & 7) ^ ((row >> 1) & 7))) * 4);
    }
    const int krow = 2 * p + (lane >> 5);
    const int chunk = lay == MX ? ((lane & 31) ^ (8 * ((krow >> 2) & 1))) : (lane & 31);
    return (unsigned)((krow * R + 4 * chunk) * 4);
  };
  constexpr int A_STEP = AL == MC ? MC_PIECE : 1024, B_STEP = BL == MC ? MC_PIECE : 1024;
  unsigned voffA[4], voffB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    voffA[j] = voff_for(AL, j, M) - (DM == 1 ? (unsigned)(j * A_STEP) : 0u);
    voffB[j] = voff_for(BL, j, N) - (DM == 1 ? (unsigned)(j * B_STEP) : 0u);
  }
  const unsigned lds0 = (unsigned)(size_t)smem;
  auto soffA = [&](int kt) { return AL == KC ? (unsigned)(kt * BK * 4) : (unsigned)(kt * BK * M * 4); };
  auto soffB = [&](int kt) { return BL == KC ? (unsigned)(kt * BK * 4) : (unsigned)(kt * BK * N * 4); };
  auto dma_a = [&](int kt, int st, int j) { dma16(rsA, voffA[j], soffA(kt), lds0 + (unsigned)(st * STAGE + pc(j) * A_STEP)); };
  auto dma_b = [&](int kt, int st, int j) { dma16(rsB, voffB[j], soffB(kt), lds0 + (unsigned)(st * STAGE + A_BYTES + pc(j) * B_STEP)); };
  auto dma_a4 = [&](int kt, int st) {
    dma16x4<A_STEP>(rsA, voffA[0], voffA[1], voffA[2], voffA[3], soffA(kt), lds0 + (unsigned)(st * STAGE + pc(0) * A_STEP));
  };
  auto dma_b4 = [&](int kt, int st) {
    dma16x4<B_STEP>(rsB, voffB[0], voffB[1], voffB[2], voffB[3], soffB(kt), lds0 + (unsigned)(st * STAGE + A_BYTES + pc(0) * B_STEP));
  };

  // ---- fragments ----
  // KC: f32x4 kq[set][g] = k 8u + 4 lh + (0..3) of row g * 32 + li.  MC: f32x2 mq[set][j] = rows li, 32 + li of k row 8u + 4 lh + j.
  f32x4 akq[2][2], bkq[2][2];
  f32x2 amq[2][4], bmq[2][4];
  const int a_kc_row[2] = {wm * 64 + li, wm * 64 + 32 + li}, b_kc_row[2] = {wn * 64 + li, wn * 64 + 32 + li};
  const int a_mc_base = lh * 2 * MC_PIECE + (wm * 64 + li) * 4, b_mc_base = lh * 2 * MC_PIECE + (wn * 64 + li) * 4;   // bytes
  f32x2 axq[2][2][2], bxq[2][2][2];          // MX: [set][column group][j pair] = k rows 2h, 2h + 1 of one column
  const int a_mx_base[2] = {lh * 2048 + ((wm * 64 + li) ^ (32 * lh)) * 4, lh * 2048 + ((wm * 64 + 32 + li) ^ (32 * lh)) * 4};
  const int b_mx_base[2] = {lh * 2048 + ((wn * 64 + li) ^ (32 * lh)) * 4, lh * 2048 + ((wn * 64 + 32 + li) ^ (32 * lh)) * 4};
  constexpr int NRA = AL == KC ? 2 : 4, NRB = BL == KC ? 2 : 4, NR = NRA + NRB;
  auto read_one = [&](int st, int u, int set, int idx) {            // idx-th fragment read of chunk u (A's first)
    const unsigned char* As = smem + st * STAGE;
    const unsigned char* Bs = As + A_BYTES;
    if (idx < NRA) {
      if constexpr (AL == KC) {
        akq[set][idx] = *reinterpret_cast<const f32x4*>(As + a_kc_row[idx] * 128 + 16 * ((2 * u + lh) ^ ((a_kc_row[idx] >> 1) & 7)));
      } else if constexpr (AL == MX) {
        const float* p = reinterpret_cast<const float*>(As + a_mx_base[idx >> 1] + u * 4096 + (idx & 1) * 1024);
        axq[set][idx >> 1][idx & 1] = f32x2{p[0], p[128]};
      } else {
        const float* p = reinterpret_cast<const float*>(As + a_mc_base + (4 * u + (idx >> 1)) * MC_PIECE + (idx & 1) * 512);
        amq[set][idx] = f32x2{p[0], p[32]};
      }
    } else {
      const int i = idx - NRA;
      if constexpr (BL == KC) {
        bkq[set][i] = *reinterpret_cast<const f32x4*>(Bs + b_kc_row[i] * 128 + 16 * ((2 * u + lh) ^ ((b_kc_row[i] >> 1) & 7)));
      } else if constexpr (BL == MX) {
        const float* p = reinterpret_cast<const float*>(Bs + b_mx_base[i >> 1] + u * 4096 + (i & 1) * 1024);
        bxq[set][i >> 1][i & 1] = f32x2{p[0], p[128]};
      } else {
        const float* p = reinterpret_cast<const float*>(Bs + b_mc_base + (4 * u + (i >> 1)) * MC_PIECE + (i & 1) * 512);
        bmq[set][i] = f32x2{p[0], p[32]};
      }
    }
  };
  auto mfma_j = [&](int set, int j) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(AL == KC ? akq[set][a][j] : AL == MX ? axq[set][a][j >> 1][j & 1] : amq[set][j][a],
                                                         BL == KC ? bkq[set][b][j] : BL == MX ? bxq[set][b][j >> 1][j & 1] : bmq[set][j][b],
                                                         acc[a][b], 0, 0, 0);
  };
  // the reads of chunk (st, u) spread over the four MFMA quads of the chunk before it: quad q issues reads q * NR / 4 ...
  auto reads_of_quad = [&](int st, int u, int set, int q) {
    if (ABL & 2) return;
#pragma unroll
    for (int idx = 0; idx < NR; ++idx)
      if (idx * 4 / NR == q) read_one(st, u, set, idx);
  };

  unsigned long long t_wait = 0, t_bar = 0, t_loop0 = 0, r0 = 0;
  // ---- prologue ----
#pragma unroll
  for (int j = 0; j < 4; ++j) { dma_a(0, 0, j); dma_b(0, 0, j); }
  if (NS == 3 && nkt > 1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { dma_a(1, 1, j); dma_b(1, 1, j); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int idx = 0; idx < NR; ++idx) read_one(0, 0, 0, idx);
  if (STAMPS) { t_loop0 = stamp(); r0 = realtime(); }

  auto tile_body = [&](int it, auto st_c) {
    constexpr int ST = decltype(st_c)::value;
    constexpr int NXT = (ST + 1) % NS, FILL = (ST + NS - 1) % NS;
    const int kt_fill = it + NS - 1;
    FENCE();
    // chunks 0..2: MFMA quads of chunk u, each followed by its share of chunk u + 1's fragment reads and (chunks 0, 1) of the DMA
#pragma unroll
    for (int u = 0; u < 3; ++u) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mfma_j(u & 1, q);
        FENCE();
        reads_of_quad(ST, u + 1, (u + 1) & 1, q);
        if (!(ABL & 1)) {
          if (DM == 0 && u == 0) { dma_a(kt_fill, FILL, q); }
          if (DM == 0 && u == 1) { dma_b(kt_fill, FILL, q); }
          if (DM == 1 && u == 0 && q == 0) dma_a4(kt_fill, FILL);
          if (DM == 1 && u == 1 && q == 0) dma_b4(kt_fill, FILL);
        }
        FENCE();
      }
    }
    // chunk 3
    unsigned long long s0 = 0, s1 = 0, s2 = 0;
    if constexpr (NS == 3) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mfma_j(1, q);
        FENCE();
        reads_of_quad(NXT, 0, 0, q);      // tile it+1 was published by the previous barrier
        FENCE();
      }
      if (STAMPS) s0 = stamp();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (STAMPS) s1 = stamp();
      __builtin_amdgcn_s_barrier();
      if (STAMPS) { s2 = stamp(); t_wait += s1 - s0; t_bar += s2 - s1; }
    } else {
      mfma_j(1, 0);
      mfma_j(1, 1);
      FENCE();
      if (STAMPS) s0 = stamp();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (STAMPS) s1 = stamp();
      __builtin_amdgcn_s_barrier();
      if (STAMPS) { s2 = stamp(); t_wait += s1 - s0; t_bar += s2 - s1; }
      FENCE();
#pragma unroll
      for (int q = 0; q < 4; ++q) reads_of_quad(NXT, 0, 0, q);
      FENCE();
      mfma_j(1, 2);
      mfma_j(1, 3);
    }
    FENCE();
  };
  int it = 0;
  if constexpr (NS == 2) {
    for (; it + 1 < nkt; it += 2) {
      tile_body(it, std::integral_constant<int, 0>{});
      tile_body(it + 1, std::integral_constant<int, 1>{});
    }
    if (it < nkt) tile_body(it, std::integral_constant<int, 0>{});
  } else {
    for (; it + 2 < nkt; it += 3) {
      tile_body(it, std::integral_constant<int, 0>{});
      tile_body(it + 1, std::integral_constant<int, 1>{});
      tile_body(it + 2, std::integral_constant<int, 2>{});
    }
    if (it < nkt) { tile_body(it, std::integral_constant<int, 0>{}); ++it; }
    if (it < nkt) { tile_body(it, std::integral_constant<int, 1>{}); ++it; }
  }
  unsigned long long t1 = 0, r1 = 0;
  if (STAMPS) { t1 = stamp(); r1 = realtime(); }

  // ---- epilogue ----   DM 2: through LDS, 16-byte stores (a wave's 32 x 64 half tile at a time, rows 68 floats apart);
  //                        DM 3: no stores at all (ablation)
  if constexpr (EPI == 2) {
    __syncthreads();                                   // every wave is done with the operand tiles
    float* stage = reinterpret_cast<float*>(smem) + wave * (32 * 68);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int v = 0; v < 16; ++v) stage[(8 * (v >> 2) + 4 * lh + (v & 3)) * 68 + b * 32 + li] = acc[a][b][v];
      __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): the wave's own writes have landed
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = (lane >> 4) + 4 * i, c4 = lane & 15;
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(stage + r * 68 + 4 * c4);
        *reinterpret_cast<f32x4*>(C + (size_t)(m0 + wm * 64 + a * 32 + r) * N + n0 + wn * 64 + 4 * c4) = v4;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
    }
  } else if constexpr (EPI == 3) {       // one store per lane: the accumulators stay live, nothing else is written
    C[(size_t)(m0 + wm * 64 + li) * N + n0 + wn * 64 + lh] = acc[0][0][0] + acc[0][1][5] + acc[1][0][9] + acc[1][1][15];
  } else {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = m0 + wm * 64 + a * 32 + 8 * (v >> 2) + 4 * lh + (v & 3);
        const int col = n0 + wn * 64 + b * 32 + li;
        C[(size_t)row * N + col] = acc[a][b][v];
      }
  }
  if (STAMPS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long r2 = realtime();
    if (lane == 0) {
      unsigned long long* o = stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
      o[0] = t1 - t_loop0; o[1] = t_wait; o[2] = t_bar; o[3] = r1 - r0; o[4] = rt_entry; o[5] = r0; o[6] = r1; o[7] = r2;
    }
  }
}

// ---- the same KC / KC loop on v_mfma_f32_16x16x4_f32 (sixteen 16 x 16 accumulators per wave, k-steps of 4): the same FLOPs
//      per cycle and the same LDS bytes, half the accumulator-register traffic per FLOP — does the chip hold a higher clock? ----
template <bool STAMPS>
__global__ __launch_bounds__(256, 2) void g5(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                             int M, int N, int K, unsigned long long* stamps) {
  constexpr int BM = 128, BN = 128, BK = 32, STAGE = 2 * 128 * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, l16 = lane & 15, g = lane >> 4;
  const int tiles_n = N / BN;
  const unsigned bid = remap_xcd(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = K / BK;
  unsigned long long rt_entry = 0;
  if (STAMPS) rt_entry = realtime();
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4 rsA = rsrc_of(A + (size_t)m0 * K, (unsigned long long)128 * K * 4), rsB = rsrc_of(B + (size_t)n0 * K, (unsigned long long)128 * K * 4);
  unsigned voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 8 * (4 * j + wave) + (lane >> 3);
    voff[j] = (unsigned)((row * K + 4 * ((lane & 7) ^ ((row >> 1) & 7))) * 4);
  }
  const unsigned lds0 = (unsigned)(size_t)smem;
  auto dma_a = [&](int kt, int st, int j) { dma16(rsA, voff[j], (unsigned)(kt * BK * 4), lds0 + (unsigned)(st * STAGE + (4 * j + wave) * 1024)); };
  auto dma_b = [&](int kt, int st, int j) { dma16(rsB, voff[j], (unsigned)(kt * BK * 4), lds0 + (unsigned)(st * STAGE + 16384 + (4 * j + wave) * 1024)); };
  // fragments: f32x4 = k 16 u + 4 g + (0..3) of row rg * 16 + l16 (u = 0, 1: the two halves of a k-tile)
  f32x4 af[2][4], bf[2][4];
  int a_row[4], b_row[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { a_row[r] = wm * 64 + r * 16 + l16; b_row[r] = wn * 64 + r * 16 + l16; }
  auto read_one = [&](int st, int u, int set, int idx) {
    const unsigned char* As = smem + st * STAGE;
    if (idx < 4) af[set][idx] = *reinterpret_cast<const f32x4*>(As + a_row[idx] * 128 + 16 * ((4 * u + g) ^ ((a_row[idx] >> 1) & 7)));
    else bf[set][idx - 4] = *reinterpret_cast<const f32x4*>(As + 16384 + b_row[idx - 4] * 128 + 16 * ((4 * u + g) ^ ((b_row[idx - 4] >> 1) & 7)));
  };
  auto mfma_ja = [&](int set, int j, int a) {
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[set][a][j], bf[set][b][j], acc[a][b], 0, 0, 0);
  };
  unsigned long long t_wait = 0, t_bar = 0, t_loop0 = 0, r0 = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) { dma_a(0, 0, j); dma_b(0, 0, j); }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int idx = 0; idx < 8; ++idx) read_one(0, 0, 0, idx);
  if (STAMPS) { t_loop0 = stamp(); r0 = realtime(); }
  auto tile_body = [&](int it, auto st_c) {
    constexpr int ST = decltype(st_c)::value;
    FENCE();
    // half 0 (k 0..15): sixteen groups of four MFMAs; the eight reads of half 1 and the eight requests of the next tile between them
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        mfma_ja(0, j, a);
        FENCE();
        const int s = j * 4 + a;
        if (s < 8) read_one(ST, 1, 1, s);
        else if (s < 12) dma_a(it + 1, ST ^ 1, s - 8);
        else dma_b(it + 1, ST ^ 1, s - 12);
        FENCE();
      }
    // half 1: eight groups, the barrier, the first reads of the next tile, eight groups
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int a = 0; a < 4; ++a) mfma_ja(1, j, a);
    FENCE();
    unsigned long long s0 = 0, s1 = 0, s2 = 0;
    if (STAMPS) s0 = stamp();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (STAMPS) s1 = stamp();
    __builtin_amdgcn_s_barrier();
    if (STAMPS) { s2 = stamp(); t_wait += s1 - s0; t_bar += s2 - s1; }
    FENCE();
#pragma unroll
    for (int idx = 0; idx < 8; ++idx) read_one(ST ^ 1, 0, 0, idx);
    FENCE();
#pragma unroll
    for (int j = 2; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < 4; ++a) mfma_ja(1, j, a);
    FENCE();
  };
  int it = 0;
  for (; it + 1 < nkt; it += 2) {
    tile_body(it, std::integral_constant<int, 0>{});
    tile_body(it + 1, std::integral_constant<int, 1>{});
  }
  if (it < nkt) tile_body(it, std::integral_constant<int, 0>{});
  unsigned long long t1 = 0, r1 = 0;
  if (STAMPS) { t1 = stamp(); r1 = realtime(); }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int v = 0; v < 4; ++v)
        C[(size_t)(m0 + wm * 64 + a * 16 + 4 * g + v) * N + n0 + wn * 64 + b * 16 + l16] = acc[a][b][v];
  if (STAMPS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long r2 = realtime();
    if (lane == 0) {
      unsigned long long* o = stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
      o[0] = t1 - t_loop0; o[1] = t_wait; o[2] = t_bar; o[3] = r1 - r0; o[4] = rt_entry; o[5] = r0; o[6] = r1; o[7] = r2;
    }
  }
}

// ---- KC / KC fragments for operands that are NOT k-contiguous in memory (the bwd-filter case: x and dz are [pixel][channel]):
//      register staging with a free 4 x 4 transpose.  A thread loads four k rows x four columns (four 16-byte loads of the
//      [K][rows] operand), and writes four ds_write_b128: row (4 mq + j) gets its four consecutive k.  Same LDS image and the same
//      fragment reads as the LDS-DMA KC tiles. ----
template <bool STAMPS>
__global__ __launch_bounds__(256, 2) void g6(const float* __restrict__ At, const float* __restrict__ Bt, float* __restrict__ C,
                                             int M, int N, int K, unsigned long long* stamps) {
  constexpr int BM = 128, BN = 128, BK = 32, STAGE = 2 * 128 * 128;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int tiles_n = N / BN;
  const unsigned bid = remap_xcd(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = K / BK;
  unsigned long long rt_entry = 0;
  if (STAMPS) rt_entry = realtime();
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
  // staging: thread -> (k quad kq = tid / 32, column quad mq = tid % 32)
  const int kq = tid >> 5, mq = tid & 31;
  const __amdgpu_buffer_rsrc_t brA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(At + m0), 0, (int)(((long long)K * M - m0) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t brB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Bt + n0), 0, (int)(((long long)K * N - n0) * 4), 0x00020000);
  unsigned voffA[4], voffB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    voffA[i] = (unsigned)(((4 * kq + i) * M + 4 * mq) * 4);
    voffB[i] = (unsigned)(((4 * kq + i) * N + 4 * mq) * 4);
  }
  unsigned wr[4];                    // LDS byte offset of row 4 mq + j, chunk kq (swizzled), inside an operand tile
#pragma unroll
  for (int j = 0; j < 4; ++j) { const int row = 4 * mq + j; wr[j] = (unsigned)(row * 128 + 16 * (kq ^ ((row >> 1) & 7))); }
  f32x4 ra[4], rb[4];
  auto load_regs = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brA, (int)voffA[i], kt * BK * M * 4, 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) rb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brB, (int)voffB[i], kt * BK * N * 4, 0));
  };
  auto store_one = [&](int st, int j, bool isb) {
    const f32x4 (&r)[4] = isb ? rb : ra;
    const f32x4 w = {r[0][j], r[1][j], r[2][j], r[3][j]};
    *reinterpret_cast<f32x4*>(smem + st * STAGE + (isb ? 16384 : 0) + wr[j]) = w;
  };
  f32x4 af[2][2], bf[2][2];
  const int a_row[2] = {wm * 64 + li, wm * 64 + 32 + li}, b_row[2] = {wn * 64 + li, wn * 64 + 32 + li};
  auto read_one = [&](int st, int u, int set, int idx) {
    const unsigned char* As = smem + st * STAGE;
    if (idx < 2) af[set][idx] = *reinterpret_cast<const f32x4*>(As + a_row[idx] * 128 + 16 * ((2 * u + lh) ^ ((a_row[idx] >> 1) & 7)));
    else bf[set][idx - 2] = *reinterpret_cast<const f32x4*>(As + 16384 + b_row[idx - 2] * 128 + 16 * ((2 * u + lh) ^ ((b_row[idx - 2] >> 1) & 7)));
  };
  auto mfma_j = [&](int set, int j) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][a][j], bf[set][b][j], acc[a][b], 0, 0, 0);
  };
  unsigned long long t_wait = 0, t_bar = 0, t_loop0 = 0, r0 = 0;
  load_regs(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) { store_one(0, j, false); store_one(0, j, true); }
  __syncthreads();
#pragma unroll
  for (int idx = 0; idx < 4; ++idx) read_one(0, 0, 0, idx);
  if (STAMPS) { t_loop0 = stamp(); r0 = realtime(); }
  auto tile_body = [&](int it, auto st_c) {
    constexpr int ST = decltype(st_c)::value;
    FENCE();
    load_regs(it + 1 < nkt ? it + 1 : it);        // (the last iteration re-reads its own tile: never used)
    FENCE();
    // chunks 0, 1: MFMAs and the next chunk's fragment reads
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        mfma_j(u & 1, q);
        FENCE();
        read_one(ST, u + 1, (u + 1) & 1, q);
        FENCE();
      }
    // chunk 2: MFMAs, chunk 3's reads and the first half of the transposed LDS writes of tile it + 1 (loads issued 32 MFMAs ago)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      mfma_j(0, q);
      FENCE();
      read_one(ST, 3, 1, q);
      store_one(ST ^ 1, q, false);
      FENCE();
    }
    // chunk 3: the other half of the writes, the barrier, the next tile's first fragments
    mfma_j(1, 0);
    FENCE();
    store_one(ST ^ 1, 0, true); store_one(ST ^ 1, 1, true);
    FENCE();
    mfma_j(1, 1);
    FENCE();
    store_one(ST ^ 1, 2, true); store_one(ST ^ 1, 3, true);
    FENCE();
    unsigned long long s0 = 0, s2 = 0;
    if (STAMPS) s0 = stamp();
    __syncthreads();
    if (STAMPS) { s2 = stamp(); t_bar += s2 - s0; }
    FENCE();
#pragma unroll
    for (int idx = 0; idx < 4; ++idx) read_one(ST ^ 1, 0, 0, idx);
    FENCE();
    mfma_j(1, 2);
    mfma_j(1, 3);
    FENCE();
  };
  int it = 0;
  for (; it + 1 < nkt; it += 2) {
    tile_body(it, std::integral_constant<int, 0>{});
    tile_body(it + 1, std::integral_constant<int, 1>{});
  }
  if (it < nkt) tile_body(it, std::integral_constant<int, 0>{});
  unsigned long long t1 = 0, r1 = 0;
  if (STAMPS) { t1 = stamp(); r1 = realtime(); }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v)
        C[(size_t)(m0 + wm * 64 + a * 32 + 8 * (v >> 2) + 4 * lh + (v & 3)) * N + n0 + wn * 64 + b * 32 + li] = acc[a][b][v];
  if (STAMPS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long r2 = realtime();
    if (lane == 0) {
      unsigned long long* o = stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
      o[0] = t1 - t_loop0; o[1] = t_wait; o[2] = t_bar; o[3] = r1 - r0; o[4] = rt_entry; o[5] = r0; o[6] = r1; o[7] = r2;
    }
  }
}

struct Problem { int M, N, K; const char* what; };
struct Bufs { const float *A, *At, *B, *Bt; float* C; unsigned long long* st; const std::vector<float>*hA, *hB; };

template <int NS, int AL, int BL, int DM, int ABL, bool STAMPS, int MINB, int EPI = 0>
static void run(const char* name, const Problem& pr, const Bufs& bf) {
  const size_t lds = (size_t)NS * (tile_bytes(AL) + tile_bytes(BL));
  auto kern = g4<NS, AL, BL, DM, ABL, STAMPS, MINB, EPI>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = (pr.M / 128) * (pr.N / 128);
  const float* Ap = AL == KC ? bf.A : bf.At;
  const float* Bp = BL == KC ? bf.B : bf.Bt;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, Ap, Bp, bf.C, pr.M, pr.N, pr.K, bf.st);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0, 0));
  // (stamped launches write their stamps to slices of their own: the gaps between consecutive launches can be read off)
  for (int i = 0; i < reps; ++i)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, Ap, Bp, bf.C, pr.M, pr.N, pr.K, bf.st + (STAMPS ? (size_t)i * grid * 32 : 0));
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, tf = 2.0 * pr.M * pr.N * pr.K / (us * 1e-6) / 1e12;
  printf("%-46s %-14s %8.1f us %6.1f TF (%.3f)", name, pr.what, us, tf, tf / 157.3);
  if (ABL == 0) {
    std::vector<float> hC((size_t)pr.M * pr.N);
    CK(hipMemcpy(hC.data(), bf.C, hC.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    unsigned s = 12345;
    for (int t = 0; t < 3000; ++t) {
      s = s * 1664525u + 1013904223u; const int m = (s >> 8) % pr.M;
      s = s * 1664525u + 1013904223u; const int n = (s >> 8) % pr.N;
      double ref = 0;
      for (int k = 0; k < pr.K; ++k) ref += (double)(*bf.hA)[(size_t)m * pr.K + k] * (*bf.hB)[(size_t)n * pr.K + k];
      maxerr = std::max(maxerr, fabs(ref - hC[(size_t)m * pr.N + n]) / (fabs(ref) + 1.0));
    }
    printf(" err %.1e %s", maxerr, maxerr < 1e-4 ? "ok" : "WRONG");
  } else {
    printf(" [ablation]");
  }
  if (STAMPS) {
    std::vector<unsigned long long> h((size_t)grid * 32);
    CK(hipMemcpy(h.data(), bf.st, h.size() * 8, hipMemcpyDeviceToHost));
    double tl = 0, tw = 0, tb = 0, tr = 0;
    unsigned long long first = ~0ull, last = 0;
    std::vector<double> pro, loop, epi, start;
    for (int i = 0; i < grid * 4; ++i) {
      const unsigned long long* o = &h[(size_t)i * 8];
      tl += o[0]; tw += o[1]; tb += o[2]; tr += o[3];
      first = std::min(first, o[4]); last = std::max(last, o[7]);
    }
    for (int i = 0; i < grid * 4; ++i) {
      const unsigned long long* o = &h[(size_t)i * 8];
      start.push_back((o[4] - first) * 0.01); pro.push_back((o[5] - o[4]) * 0.01); loop.push_back((o[6] - o[5]) * 0.01); epi.push_back((o[7] - o[6]) * 0.01);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    auto mx = [](std::vector<double> v) { return *std::max_element(v.begin(), v.end()); };
    const double nk = pr.K / 32.0;
    {
      std::vector<unsigned long long> all((size_t)reps * grid * 32);
      CK(hipMemcpy(all.data(), bf.st, all.size() * 8, hipMemcpyDeviceToHost));
      double gap = 0, span = 0;
      unsigned long long prev_end = 0;
      for (int i = 0; i < reps; ++i) {
        unsigned long long f = ~0ull, l = 0;
        for (int w = 0; w < grid * 4; ++w) { const unsigned long long* o = &all[((size_t)i * grid * 4 + w) * 8]; f = std::min(f, o[4]); l = std::max(l, o[7]); }
        span += (l - f) * 0.01;
        if (i) gap += (double)(long long)(f - prev_end) * 0.01;
        prev_end = l;
      }
      printf("\n      20 launches: first wave in -> last wave out %.1f us on average, last wave out -> next launch's first wave in %.1f us", span / reps, gap / (reps - 1));
    }
    printf("\n      per k-tile %.0f cyc (dma wait %.0f, barrier %.0f), clock %.2f GHz | us: kernel span %.1f, wave start med %.1f max %.1f, prologue med %.1f, loop med %.1f max %.1f, epilogue med %.1f max %.1f",
           tl / (grid * 4) / nk, tw / (grid * 4) / nk, tb / (grid * 4) / nk, tl / tr * 0.1, (last - first) * 0.01, med(start), mx(start),
           med(pro), med(loop), mx(loop), med(epi), mx(epi));
  }
  printf("\n");
  fflush(stdout);
}

template <bool STAMPS, int WHICH = 5>
static void run5(const char* name, const Problem& pr, const Bufs& bf_in) {
  Bufs bf = bf_in;
  if (WHICH == 6) { bf.A = bf_in.At; bf.B = bf_in.Bt; }          // g6 takes the [K][rows] operands
  const size_t lds = 2 * 2 * 128 * 128;
  auto kern = WHICH == 6 ? g6<STAMPS> : g5<STAMPS>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = (pr.M / 128) * (pr.N / 128);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, bf.A, bf.B, bf.C, pr.M, pr.N, pr.K, bf.st);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, bf.A, bf.B, bf.C, pr.M, pr.N, pr.K, bf.st);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, tf = 2.0 * pr.M * pr.N * pr.K / (us * 1e-6) / 1e12;
  std::vector<float> hC((size_t)pr.M * pr.N);
  CK(hipMemcpy(hC.data(), bf.C, hC.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  unsigned s = 12345;
  for (int t = 0; t < 3000; ++t) {
    s = s * 1664525u + 1013904223u; const int m = (s >> 8) % pr.M;
    s = s * 1664525u + 1013904223u; const int n = (s >> 8) % pr.N;
    double ref = 0;
    for (int k = 0; k < pr.K; ++k) ref += (double)(*bf.hA)[(size_t)m * pr.K + k] * (*bf.hB)[(size_t)n * pr.K + k];
    maxerr = std::max(maxerr, fabs(ref - hC[(size_t)m * pr.N + n]) / (fabs(ref) + 1.0));
  }
  printf("%-46s %-14s %8.1f us %6.1f TF (%.3f) err %.1e %s", name, pr.what, us, tf, tf / 157.3, maxerr, maxerr < 1e-4 ? "ok" : "WRONG");
  if (STAMPS) {
    std::vector<unsigned long long> h((size_t)grid * 32);
    CK(hipMemcpy(h.data(), bf.st, h.size() * 8, hipMemcpyDeviceToHost));
    double tl = 0, tw = 0, tb = 0, tr = 0;
    for (int i = 0; i < grid * 4; ++i) { const unsigned long long* o = &h[(size_t)i * 8]; tl += o[0]; tw += o[1]; tb += o[2]; tr += o[3]; }
    printf("\n      per k-tile %.0f cyc (dma wait %.0f, barrier %.0f), clock %.2f GHz", tl / (grid * 4) / (pr.K / 32.0), tw / (grid * 4) / (pr.K / 32.0),
           tb / (grid * 4) / (pr.K / 32.0), tl / tr * 0.1);
  }
  printf("\n");
  fflush(stdout);
}

int main() {
  const Problem probs[] = {{32000, 256, 2400, "conv2d_1-like"}, {7424, 384, 3456, "conv2d_3-like"}};
  for (const Problem& pr : probs) {
    std::vector<float> hA((size_t)pr.M * pr.K), hB((size_t)pr.N * pr.K), hAt(hA.size()), hBt(hB.size());
    unsigned s = 777;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = ((s >> 9) & 0xffff) / 65536.f - 0.5f; }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = ((s >> 9) & 0xffff) / 65536.f - 0.5f; }
    for (int m = 0; m < pr.M; ++m)
      for (int k = 0; k < pr.K; ++k) hAt[(size_t)k * pr.M + m] = hA[(size_t)m * pr.K + k];
    for (int n = 0; n < pr.N; ++n)
      for (int k = 0; k < pr.K; ++k) hBt[(size_t)k * pr.N + n] = hB[(size_t)n * pr.K + k];
    float *dA, *dAt, *dB, *dBt, *dC;
    unsigned long long* dst;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dAt, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dBt, hB.size() * 4));
    CK(hipMalloc(&dC, (size_t)pr.M * pr.N * 4)); CK(hipMalloc(&dst, (size_t)20 * 512 * 32 * 8));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dAt, hAt.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dBt, hBt.data(), hBt.size() * 4, hipMemcpyHostToDevice));
    const Bufs bf{dA, dAt, dB, dBt, dC, dst, &hA, &hB};
    //  NS  AL  BL  DM ABL STAMPS MINB
    if (getenv("G4_CLOCK")) {
      // the clock question: the same loop without DMA, fragment reads as ds_read_b128 (KC) or ds_read2_b32 (MC), in alternating order
      for (int rep = 0; rep < 3; ++rep) {
        run<2, KC, KC, 0, 0, false, 2>("2st KC/KC, LDS-DMA", pr, bf);
        run<2, MX, MX, 0, 0, false, 2>("2st MX/MX, LDS-DMA (operands as stored)", pr, bf);
        run5<false, 6>("2st KC/KC, registers + transposed writes", pr, bf);
        run5<true, 6>("2st KC/KC, registers + transposed writes, stamps", pr, bf);
      }
    } else {
    run<2, KC, KC, 0, 0, false, 2>("2st KC/KC (bwd-data form)", pr, bf);
    run<2, KC, MC, 0, 0, false, 2>("2st KC/MC (forward form)", pr, bf);
    run<2, MC, MC, 0, 0, false, 2>("2st MC/MC (bwd-filter form)", pr, bf);
    run<3, KC, KC, 0, 0, false, 1>("3st KC/KC 1 block/CU", pr, bf);
    run<3, KC, MC, 0, 0, false, 1>("3st KC/MC 1 block/CU", pr, bf);
    run<3, MC, MC, 0, 0, false, 1>("3st MC/MC 1 block/CU", pr, bf);
    run<2, KC, KC, 0, 0, true, 2>("2st KC/KC stamps", pr, bf);
    run<2, MC, MC, 0, 0, true, 2>("2st MC/MC stamps", pr, bf);
    run<3, KC, KC, 0, 0, true, 1>("3st KC/KC stamps", pr, bf);
    run<3, MC, MC, 0, 0, true, 1>("3st MC/MC stamps", pr, bf);
    run<3, KC, KC, 0, 1, true, 1>("3st KC/KC stamps, no DMA", pr, bf);
    run<3, KC, KC, 0, 2, true, 1>("3st KC/KC stamps, no frag reads", pr, bf);
    run<3, KC, KC, 0, 3, true, 1>("3st KC/KC stamps, MFMAs + barrier only", pr, bf);
    run<3, MC, MC, 0, 1, true, 1>("3st MC/MC stamps, no DMA", pr, bf);
    }
    CK(hipFree(dA)); CK(hipFree(dAt)); CK(hipFree(dB)); CK(hipFree(dBt)); CK(hipFree(dC)); CK(hipFree(dst));
  }
  return 0;
}
