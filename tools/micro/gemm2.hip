// Prototype of the second-generation fp32 GEMM inner loop: a plain C[M][N] = A[M][K] * B[N][K]^T on
// v_mfma_f32_32x32x2_f32 with 64x64 per wave (four accumulators), BOTH operands k-contiguous in LDS so that one ds_read_b128
// feeds four k-steps of a 32-row group: 4 fragment reads per 16 MFMAs.  No im2col addressing here — this file only answers
// what the loop structure can reach on an MI355X before it is married to igemm.h's gather.
//   STAGE 0: global -> registers -> ds_write_b128 (rows padded to 36 floats)
//   STAGE 1: LDS-DMA (buffer_load_dwordx4 ... lds), 128-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7
//   STAGE 2: no staging at all (ablation: the tiles are written once)
//   PIPE  : fragments of chunk u + 1 are read while chunk u's MFMAs issue
//   NOFRAG: ablation — fragments read once, the loop is MFMAs (+ staging) only
// Build: hipcc -O3 --offload-arch=gfx950 tools/micro/gemm2.hip -o gpurun_out/gemm2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

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
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_addr), "v"(voff), "s"(rs), "s"(soff)
               : "memory");
}

template <int STAGE, bool PIPE, bool NOFRAG, int MINB, int BK>
__global__ __launch_bounds__(256, MINB) void g2(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                int M, int N, int K) {
  constexpr int BM = 128, BN = 128;
  constexpr int LD = (STAGE == 1) ? BK : BK + 4;
  constexpr int CPR = BK / 4;                    // 16-byte chunks per row
  constexpr int RPI = 256 / CPR;                 // rows per pass of the 256 threads
  constexpr int NL = BM / RPI;                   // passes per operand
  constexpr int NCH = BK / 8;                    // fragment chunks per tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * BM * LD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int tiles_n = N / BN;
  const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = K / BK;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  // ---- staging state ----
  const int s_c = tid % CPR, s_r = tid / CPR;
  const u32x4 rsA = rsrc_of(A + (size_t)m0 * K, (unsigned long long)BM * K * 4);
  const u32x4 rsB = rsrc_of(B + (size_t)n0 * K, (unsigned long long)BN * K * 4);
  const __amdgpu_buffer_rsrc_t brA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (size_t)m0 * K), 0, BM * K * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t brB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B + (size_t)n0 * K), 0, BN * K * 4, 0x00020000);
  constexpr int SWS = CPR >= 16 ? 0 : 1;         // 128-byte rows: two rows span the 64 banks
  unsigned voff[NL];              // STAGE 0: row s_r + RPI j, chunk s_c.  STAGE 1: row = s_r + RPI j too, SOURCE chunk swizzled
#pragma unroll
  for (int j = 0; j < NL; ++j) {
    const int row = s_r + RPI * j;
    const int chunk = (STAGE == 1) ? (s_c ^ ((row >> SWS) & (CPR - 1))) : s_c;
    voff[j] = (unsigned)((row * K + 4 * chunk) * 4);
  }
  f32x4 ra[NL], rb[NL];
  auto load_regs = [&](int kt) {
    const int soff = kt * BK * 4;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(brA, (int)voff[j], soff, 0);
      ra[j] = __builtin_bit_cast(f32x4, va);
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(brB, (int)voff[j], soff, 0);
      rb[j] = __builtin_bit_cast(f32x4, vb);
    }
  };
  auto store_regs = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      *reinterpret_cast<f32x4*>(As + buf * BM * LD + (s_r + RPI * j) * LD + 4 * s_c) = ra[j];
      *reinterpret_cast<f32x4*>(Bs + buf * BN * LD + (s_r + RPI * j) * LD + 4 * s_c) = rb[j];
    }
  };
  // LDS-DMA: wave w issues passes j (each pass of the block = 4 wave-instructions of 1 KiB; this wave's lies at
  // (RPI j + 64 w / CPR) rows = byte (j * 256 + w * 64) * 16 of the tile)
  const unsigned lds_a = (unsigned)(size_t)As, lds_b = (unsigned)(size_t)Bs;
  auto dma_tile = [&](int kt, int buf) {
    const unsigned soff = (unsigned)(kt * BK * 4);
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      dma16(rsA, voff[j], soff, lds_a + (unsigned)(buf * BM * LD * 4) + (unsigned)((j * 256 + wave * 64) * 16));
      dma16(rsB, voff[j], soff, lds_b + (unsigned)(buf * BN * LD * 4) + (unsigned)((j * 256 + wave * 64) * 16));
    }
  };

  // ---- fragment addressing ----
  // padded rows: address = row * LD + 8 u + 4 lh; swizzled rows: row * BK + 4 * ((2 u + lh) ^ swz(row))
  auto frag_ptr = [&](const float* base, int row, int u) -> const f32x4* {
    if constexpr (STAGE == 1) return reinterpret_cast<const f32x4*>(base + row * LD + 4 * ((2 * u + lh) ^ ((row >> SWS) & (CPR - 1))));
    else return reinterpret_cast<const f32x4*>(base + row * LD + 8 * u + 4 * lh);
  };

  if (STAGE == 1) {
    dma_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    load_regs(0);
    store_regs(0);
  }
  __syncthreads();

  f32x4 af[2][2], bf[2][2];     // [register buffer][row group]
  auto read_frags = [&](int cur, int u, int rb_) {
    const float* Ac = As + cur * BM * LD;
    const float* Bc = Bs + cur * BN * LD;
#pragma unroll
    for (int a = 0; a < 2; ++a) af[rb_][a] = *frag_ptr(Ac, wm * 64 + a * 32 + li, u);
#pragma unroll
    for (int b = 0; b < 2; ++b) bf[rb_][b] = *frag_ptr(Bc, wn * 64 + b * 32 + li, u);
  };
  auto mfmas = [&](int rb_) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[rb_][a][j], bf[rb_][b][j], acc[a][b], 0, 0, 0);
  };

  if (NOFRAG) read_frags(0, 0, 0);
  auto tile_body = [&](int it, auto cur_c) {
    constexpr int cur = decltype(cur_c)::value;
    const bool more = it + 1 < nkt;
    if (STAGE == 0) { if (more) load_regs(it + 1); }
    if (STAGE == 1) { if (more) dma_tile(it + 1, cur ^ 1); }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NOFRAG) {
#pragma unroll
      for (int u = 0; u < NCH; ++u) {
        if (u == NCH - 1 && STAGE == 0 && more) store_regs(cur ^ 1);
        mfmas(0);
      }
    } else if constexpr (PIPE) {
      read_frags(cur, 0, 0);
#pragma unroll
      for (int u = 0; u < NCH; ++u) {
        if (u + 1 < NCH) read_frags(cur, u + 1, (u + 1) & 1);
        if (u == NCH - 1 && STAGE == 0 && more) store_regs(cur ^ 1);
        mfmas(u & 1);
        if (u + 1 < NCH) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // two MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read of the next chunk
          }
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < NCH; ++u) {
        read_frags(cur, u, 0);
        if (u == NCH - 1 && STAGE == 0 && more) store_regs(cur ^ 1);
        mfmas(0);
      }
    }
    if (STAGE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };
  for (int it = 0; it < nkt; it += 2) {
    tile_body(it, std::integral_constant<int, 0>{});
    if (it + 1 < nkt) tile_body(it + 1, std::integral_constant<int, 1>{});
  }

  // ---- epilogue ----
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

struct Problem { int M, N, K; const char* what; };

template <int STAGE, bool PIPE, bool NOFRAG, int MINB, int BK>
static void run(const char* name, const Problem& pr, const float* dA, const float* dB, float* dC, const std::vector<float>& hA,
                const std::vector<float>& hB, bool check) {
  constexpr int LD = (STAGE == 1) ? BK : BK + 4;
  const size_t lds = (size_t)2 * (128 + 128) * LD * 4;
  auto kern = g2<STAGE, PIPE, NOFRAG, MINB, BK>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = (pr.M / 128) * (pr.N / 128);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, dB, dC, pr.M, pr.N, pr.K);
  CK(hipDeviceSynchronize());
  const int reps = 20;
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, dA, dB, dC, pr.M, pr.N, pr.K);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, tf = 2.0 * pr.M * pr.N * pr.K / (us * 1e-6) / 1e12;
  double maxerr = -1;
  if (check) {
    std::vector<float> hC((size_t)pr.M * pr.N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    maxerr = 0;
    unsigned s = 12345;
    for (int t = 0; t < 4000; ++t) {
      s = s * 1664525u + 1013904223u; const int m = (s >> 8) % pr.M;
      s = s * 1664525u + 1013904223u; const int n = (s >> 8) % pr.N;
      double ref = 0;
      for (int k = 0; k < pr.K; ++k) ref += (double)hA[(size_t)m * pr.K + k] * hB[(size_t)n * pr.K + k];
      const double err = fabs(ref - hC[(size_t)m * pr.N + n]) / (fabs(ref) + 1.0);
      if (err > maxerr) maxerr = err;
    }
  }
  printf("%-44s %-22s grid %4d  %8.1f us  %6.1f TF  (%.3f of 157.3)%s", name, pr.what, grid, us, tf, tf / 157.3,
         check ? "" : "  [ablation: results not meaningful]\n");
  if (check) printf("  max rel err %.2e %s\n", maxerr, maxerr < 1e-4 ? "ok" : "WRONG");
  fflush(stdout);
}

int main() {
  const Problem probs[] = {{32000, 256, 2400, "conv2d_1-like"}, {65536, 128, 1600, "fine2-like(N=128)"}};
  for (const Problem& pr : probs) {
    std::vector<float> hA((size_t)pr.M * pr.K), hB((size_t)pr.N * pr.K);
    unsigned s = 777;
    for (auto& v : hA) { s = s * 1664525u + 1013904223u; v = ((s >> 9) & 0xffff) / 65536.f - 0.5f; }
    for (auto& v : hB) { s = s * 1664525u + 1013904223u; v = ((s >> 9) & 0xffff) / 65536.f - 0.5f; }
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)pr.M * pr.N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    //   STAGE PIPE NOFRAG MINB BK
    run<0, false, false, 2, 32>("reg-staged", pr, dA, dB, dC, hA, hB, true);
    run<0, true, false, 2, 32>("reg-staged, fragment pipe", pr, dA, dB, dC, hA, hB, true);
    run<1, false, false, 2, 32>("lds-dma", pr, dA, dB, dC, hA, hB, true);
    run<1, true, false, 2, 32>("lds-dma, fragment pipe", pr, dA, dB, dC, hA, hB, true);
    run<0, true, false, 1, 32>("reg-staged, fragment pipe, 1 block/CU", pr, dA, dB, dC, hA, hB, true);
    run<1, true, false, 1, 32>("lds-dma, fragment pipe, 1 block/CU", pr, dA, dB, dC, hA, hB, true);
    run<0, true, false, 2, 16>("reg-staged, fragment pipe, BK 16", pr, dA, dB, dC, hA, hB, true);
    run<1, true, false, 2, 64>("lds-dma, fragment pipe, BK 64 (1 block fits)", pr, dA, dB, dC, hA, hB, true);
    run<2, false, false, 2, 32>("ablation: no staging", pr, dA, dB, dC, hA, hB, false);
    run<2, true, false, 2, 32>("ablation: no staging, fragment pipe", pr, dA, dB, dC, hA, hB, false);
    run<2, false, true, 2, 32>("ablation: MFMAs only", pr, dA, dB, dC, hA, hB, false);
    run<0, false, true, 2, 32>("ablation: reg staging + MFMAs, no frag reads", pr, dA, dB, dC, hA, hB, false);
    run<1, false, true, 2, 32>("ablation: lds-dma + MFMAs, no frag reads", pr, dA, dB, dC, hA, hB, false);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
  }
  return 0;
}
