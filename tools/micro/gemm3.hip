// Second prototype of the gen-2 fp32 GEMM loop: LDS-DMA staging only, the order of a k-tile's instructions pinned by
// sched_barrier fences (gemm2.hip showed the compiler's own order exposing every LDS round trip), optional cycle stamps.
//   C[M][N] = A[M][K] * B^T,  A k-contiguous.  BMC = false: B given as [N][K] (k-contiguous tile, ds_read_b128 fragments);
//   BMC = true: B given as [K][N] (tile rows = k, 512-byte rows of 128 n, ds_read_b64 fragments: lane li holds columns 2 li, 2 li + 1).
//   NS: LDS stages (2: tile t+1 lands while tile t is multiplied, its first fragments are read after the barrier;
//                   3: tile t+2 lands meanwhile and the first fragments of tile t+1 are read BEFORE the barrier of tile t).
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/gemm3.hip -o tools/micro/bin/gemm3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
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

// XCD-aware bijective remap: the blocks an XCD runs are consecutive tiles (tile_n fastest)
__device__ __forceinline__ unsigned remap_xcd(unsigned bid, unsigned nwg) {
  const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <int NS, bool BMC, bool STAMPS, int MINB>
__global__ __launch_bounds__(256, MINB) void g3(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                int M, int N, int K, unsigned long long* stamps) {
  constexpr int BM = 128, BN = 128, BK = 32;
  constexpr int TILE = BM * BK;                 // floats per operand tile (both layouts: 16 KiB)
  constexpr int STAGE = 2 * TILE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int tiles_n = N / BN;
  const unsigned bid = remap_xcd(blockIdx.x, gridDim.x);
  const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = K / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  // ---- LDS-DMA state.  A wave issues pieces j = 0..3 of each operand: piece (4 j + wave) of the tile's sixteen KiB. ----
  // k-contiguous tile [128 rows][32 k]: piece p = rows 8 p .. 8 p + 7, lane l -> row 8 p + (l >> 3), chunk POSITION l & 7 holds
  // source chunk (l & 7) ^ ((row >> 1) & 7).
  // row-major [32 k][128 n] tile: piece p = k rows 2 p, 2 p + 1, lane l -> k row 2 p + (l >> 5), columns 4 (l & 31) ..
  const u32x4 rsA = rsrc_of(A + (size_t)m0 * K, (unsigned long long)BM * K * 4);
  const u32x4 rsB = BMC ? rsrc_of(B + n0, ((unsigned long long)K * N - n0) * 4) : rsrc_of(B + (size_t)n0 * K, (unsigned long long)BN * K * 4);
  unsigned voffA[4], voffB[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = 4 * j + wave;
    const int row = 8 * p + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    voffA[j] = (unsigned)((row * K + 4 * chunk) * 4);
    if (BMC) voffB[j] = (unsigned)(((2 * p + (lane >> 5)) * N + 4 * (lane & 31)) * 4);
    else voffB[j] = voffA[j];
  }
  const unsigned lds0 = (unsigned)(size_t)smem;
  auto dma_a = [&](int kt, int st, int j) {
    dma16(rsA, voffA[j], (unsigned)(kt * BK * 4), lds0 + (unsigned)((st * STAGE) * 4 + (4 * j + wave) * 1024));
  };
  auto dma_b = [&](int kt, int st, int j) {
    dma16(rsB, voffB[j], BMC ? (unsigned)(kt * BK * N * 4) : (unsigned)(kt * BK * 4),
          lds0 + (unsigned)((st * STAGE + TILE) * 4 + (4 * j + wave) * 1024));
  };

  // ---- fragments.  k-contiguous: f32x4 = k 8u + 4 lh + (0..3) of row li;  row-major B: f32x2 per k-pair s: columns 2 li, 2 li + 1
  //      of k row 2 s + lh.  The four MFMAs of k-pair index j in a chunk use k rows 8u + 4 lh + j on the A side, so the B side
  //      reads k row 8u + 4 lh + j as well.
  const int a_row[2] = {wm * 64 + li, wm * 64 + 32 + li};
  const int b_row[2] = {wn * 64 + li, wn * 64 + 32 + li};
  f32x4 af[2][2];               // [register set][row group]
  f32x4 bk[2][2];               // B k-contiguous: [set][column group]
  f32x2 bm[2][4];               // B row-major:    [set][k-pair j] = columns (2 li, 2 li + 1) of the wave's 64
  auto read_frags = [&](int st, int u, int set) {
    const float* As = smem + st * STAGE;
    const float* Bs = As + TILE;
#pragma unroll
    for (int a = 0; a < 2; ++a)
      af[set][a] = *reinterpret_cast<const f32x4*>(As + a_row[a] * BK + 4 * ((2 * u + lh) ^ ((a_row[a] >> 1) & 7)));
    if constexpr (BMC) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bm[set][j] = *reinterpret_cast<const f32x2*>(Bs + (8 * u + 4 * lh + j) * BN + wn * 64 + 2 * li);
    } else {
#pragma unroll
      for (int b = 0; b < 2; ++b)
        bk[set][b] = *reinterpret_cast<const f32x4*>(Bs + b_row[b] * BK + 4 * ((2 * u + lh) ^ ((b_row[b] >> 1) & 7)));
    }
  };
  auto mfma_j = [&](int set, int j) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][a][j], BMC ? bm[set][j][b] : bk[set][b][j], acc[a][b], 0, 0, 0);
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
  read_frags(0, 0, 0);
  if (STAMPS) { t_loop0 = stamp(); r0 = realtime(); }

  // One k-tile.  ST = stage of tile `it` (compile-time: the loop is unrolled NS times).
  auto tile_body = [&](int it, auto st_c) {
    constexpr int ST = decltype(st_c)::value;
    constexpr int NXT = (ST + 1) % NS, FILL = (ST + NS - 1) % NS;     // stage of tile it+1; stage the DMAs of this iteration fill
    const int kt_fill = it + NS - 1;                                  // tile requested in this iteration (past the end: zeros)
    FENCE();
    // chunk 0: the eight DMA requests go out between its MFMAs, fragments of chunk 1 are read
    read_frags(ST, 1, 1);
    FENCE();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mfma_j(0, j);
      FENCE();
      dma_a(kt_fill, FILL, j);
      dma_b(kt_fill, FILL, j);
      FENCE();
    }
    // chunk 1
    read_frags(ST, 2, 0);
    FENCE();
#pragma unroll
    for (int j = 0; j < 4; ++j) mfma_j(1, j);
    FENCE();
    // chunk 2
    read_frags(ST, 3, 1);
    FENCE();
#pragma unroll
    for (int j = 0; j < 4; ++j) mfma_j(0, j);
    FENCE();
    // chunk 3
    if constexpr (NS == 3) {
      // tile it+1 landed an iteration ago and was published by the previous barrier: its first fragments are read now, and
      // the barrier at the end only has to publish tile it+2
      read_frags(NXT, 0, 0);
      FENCE();
#pragma unroll
      for (int j = 0; j < 4; ++j) mfma_j(1, j);
      FENCE();
      unsigned long long s0 = 0, s1 = 0, s2 = 0;
      if (STAMPS) s0 = stamp();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (STAMPS) s1 = stamp();
      __builtin_amdgcn_s_barrier();
      if (STAMPS) { s2 = stamp(); t_wait += s1 - s0; t_bar += s2 - s1; }
    } else {
      mfma_j(1, 0);
      mfma_j(1, 1);
      FENCE();
      unsigned long long s0 = 0, s1 = 0, s2 = 0;
      if (STAMPS) s0 = stamp();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (STAMPS) s1 = stamp();
      __builtin_amdgcn_s_barrier();
      if (STAMPS) { s2 = stamp(); t_wait += s1 - s0; t_bar += s2 - s1; }
      FENCE();
      read_frags(NXT, 0, 0);
      FENCE();
      mfma_j(1, 2);
      mfma_j(1, 3);
    }
    FENCE();
  };
  // all tiles but the last NS-1 positions run unrolled by NS; reads / requests past the last tile touch valid LDS and
  // out-of-range (zero) source addresses only
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
  if (STAMPS) {
    const unsigned long long t1 = stamp(), r1 = realtime();
    if (lane == 0) {
      unsigned long long* o = stamps + ((size_t)blockIdx.x * 4 + wave) * 4;
      o[0] = t1 - t_loop0; o[1] = t_wait; o[2] = t_bar; o[3] = r1 - r0;
    }
  }

  // ---- epilogue ----
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int row = m0 + wm * 64 + a * 32 + 8 * (v >> 2) + 4 * lh + (v & 3);
        const int col = n0 + wn * 64 + (BMC ? 2 * li + b : b * 32 + li);
        C[(size_t)row * N + col] = acc[a][b][v];
      }
}

struct Problem { int M, N, K; const char* what; };

template <int NS, bool BMC, bool STAMPS, int MINB>
static void run(const char* name, const Problem& pr, const float* dA, const float* dB, const float* dBt, float* dC,
                const std::vector<float>& hA, const std::vector<float>& hB, unsigned long long* dstamps) {
  const size_t lds = (size_t)NS * 2 * 128 * 32 * 4;
  auto kern = g3<NS, BMC, STAMPS, MINB>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = (pr.M / 128) * (pr.N / 128);
  const float* Bp = BMC ? dBt : dB;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, Bp, dC, pr.M, pr.N, pr.K, dstamps);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, Bp, dC, pr.M, pr.N, pr.K, dstamps);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, tf = 2.0 * pr.M * pr.N * pr.K / (us * 1e-6) / 1e12;
  std::vector<float> hC((size_t)pr.M * pr.N);
  CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  unsigned s = 12345;
  for (int t = 0; t < 4000; ++t) {
    s = s * 1664525u + 1013904223u; const int m = (s >> 8) % pr.M;
    s = s * 1664525u + 1013904223u; const int n = (s >> 8) % pr.N;
    double ref = 0;
    for (int k = 0; k < pr.K; ++k) ref += (double)hA[(size_t)m * pr.K + k] * hB[(size_t)n * pr.K + k];
    const double err = fabs(ref - hC[(size_t)m * pr.N + n]) / (fabs(ref) + 1.0);
    if (err > maxerr) maxerr = err;
  }
  printf("%-40s %-18s grid %4d  %8.1f us  %6.1f TF  (%.3f)  err %.1e %s", name, pr.what, grid, us, tf, tf / 157.3, maxerr,
         maxerr < 1e-4 ? "ok" : "WRONG");
  if (STAMPS) {
    std::vector<unsigned long long> h((size_t)grid * 16);
    CK(hipMemcpy(h.data(), dstamps, h.size() * 8, hipMemcpyDeviceToHost));
    double tl = 0, tw = 0, tb = 0, tr = 0;
    for (int i = 0; i < grid * 4; ++i) { tl += h[i * 4]; tw += h[i * 4 + 1]; tb += h[i * 4 + 2]; tr += h[i * 4 + 3]; }
    const double nk = pr.K / 32.0;
    printf("  | per k-tile: %.0f cycles (MFMA time 4096 x waves/SIMD), dma wait %.0f, barrier %.0f; clock %.2f GHz",
           tl / (grid * 4) / nk, tw / (grid * 4) / nk, tb / (grid * 4) / nk, tl / tr * 0.1);
  }
  printf("\n");
  fflush(stdout);
}

int main() {
  const Problem probs[] = {{32000, 256, 2400, "conv2d_1-like"}, {65536, 128, 1600, "fine2-like"}, {7424, 384, 3456, "conv2d_3-like"}};
  for (const Problem& pr : probs) {
    std::vector<float> hA((size_t)pr.M * pr.K), hB((size_t)pr.N * pr.K), hBt((size_t)pr.N * pr.K);
    unsigned s = 777;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = ((s >> 9) & 0xffff) / 65536.f - 0.5f; }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = ((s >> 9) & 0xffff) / 65536.f - 0.5f; }
    for (int n = 0; n < pr.N; ++n)
      for (int k = 0; k < pr.K; ++k) hBt[(size_t)k * pr.N + n] = hB[(size_t)n * pr.K + k];
    float *dA, *dB, *dBt, *dC;
    unsigned long long* dst;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dBt, hB.size() * 4));
    CK(hipMalloc(&dC, (size_t)pr.M * pr.N * 4)); CK(hipMalloc(&dst, (size_t)4096 * 16 * 8));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dBt, hBt.data(), hBt.size() * 4, hipMemcpyHostToDevice));
    run<2, false, false, 2>("2 stages, B k-contiguous", pr, dA, dB, dBt, dC, hA, hB, dst);
    run<2, true, false, 2>("2 stages, B row-major (b64 frags)", pr, dA, dB, dBt, dC, hA, hB, dst);
    run<3, false, false, 1>("3 stages, B k-contiguous, 1 block/CU", pr, dA, dB, dBt, dC, hA, hB, dst);
    run<3, true, false, 1>("3 stages, B row-major, 1 block/CU", pr, dA, dB, dBt, dC, hA, hB, dst);
    run<2, false, true, 2>("2 stages, B k-contiguous, stamps", pr, dA, dB, dBt, dC, hA, hB, dst);
    run<2, true, true, 2>("2 stages, B row-major, stamps", pr, dA, dB, dBt, dC, hA, hB, dst);
    run<3, false, true, 1>("3 stages, B k-contiguous, stamps", pr, dA, dB, dBt, dC, hA, hB, dst);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dBt)); CK(hipFree(dC)); CK(hipFree(dst));
  }
  return 0;
}
