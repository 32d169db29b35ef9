// Micro-benchmark: what a vector instruction costs beside v_mfma_f32_32x32x2_f32 on gfx950.
//   (a) one wave per SIMD: MFMAs back to back with k independent v_fma_f32 between two of them — cycles per MFMA by k;
//   (b) two waves per SIMD: wave A issues only MFMAs, wave B only v_fma_f32 (or ds_read_b32 / nothing) — cycles per MFMA of A
//       and cycles per instruction of B.
// Build:  hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f32_fillers.hip -o gpurun_out/mfma_f32_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND 0: v_fma_f32, 1: ds_read_b32 (no wait inside the loop), 3: ds_read_b128
template <int K, int KIND = 0>
__global__ __launch_bounds__(256) void one_wave(unsigned long long* out, float* sink, int iters) {
  __shared__ float lds1[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds1[i] = i;
  __syncthreads();
  const unsigned lp = (unsigned)(size_t)(lds1 + (threadIdx.x & 63) * (KIND == 3 ? 4 : 1));
  unsigned sreg = 1;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 q[4] = {};
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
  float a = threadIdx.x * 0.5f, b = 1.0f + threadIdx.x;
  float f[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[k % 8]) : "v"(a));
        else if (KIND == 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f[k % 8]) : "v"(lp), "n"((k % 8) * 256) : "memory");
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[k % 4]) : "v"(lp), "n"((k % 4) * 1024) : "memory");
      }
    }
    if (KIND == 1 || KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f[0] += (float)sreg + q[0][0] + q[1][1] + q[2][2] + q[3][3];
  float s = 0.f;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][7];
  for (int k = 0; k < 8; ++k) s += f[k];
  sink[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// 512 threads = 8 waves = 2 per SIMD; waves 0-3 MFMA only, waves 4-7 the filler stream (MODE 0: v_fma, 1: ds_read_b32, 2: idle)
template <int MODE>
__global__ __launch_bounds__(512) void two_waves(unsigned long long* out, float* sink, int iters) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  unsigned long long t0 = 0, t1 = 0;
  float s = 0.f;
  if (wave < 4) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
      for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
    float a = threadIdx.x * 0.5f, b = 1.0f + threadIdx.x;
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    t1 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][7];
  } else if (MODE != 2) {
    float f[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    float a = threadIdx.x * 0.5f;
    const float* p = lds + (threadIdx.x & 63);
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[k % 8]) : "v"(a));
        else asm volatile("ds_read_b32 %0, %1 offset:%2\n\ts_waitcnt lgkmcnt(0)" : "=v"(f[k % 8]) : "v"((unsigned)(size_t)p), "n"(k * 256) : "memory");
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
    for (int k = 0; k < 8; ++k) s += f[k];
  }
  sink[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
}

// (c) NW waves per SIMD (block of NW * 256 threads), every wave the same mix: one MFMA, K v_fma_f32 — SIMD cycles per MFMA
template <int K, int NW>
__global__ __launch_bounds__(NW * 256) void mixed(unsigned long long* out, float* sink, int iters) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;
  float a = threadIdx.x * 0.5f, b = 1.0f + threadIdx.x;
  float f[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[k % 8]) : "v"(a));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][7];
  for (int k = 0; k < 8; ++k) s += f[k];
  sink[(blockIdx.x * NW * 256 + threadIdx.x) % (256 * 512)] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  const int blocks = 256, iters = 2000;
  unsigned long long* out;
  float* sink;
  hipMalloc(&out, blocks * 16 * 8);
  hipMalloc(&sink, blocks * 512 * 4);
  std::vector<unsigned long long> h(blocks * 16);
  auto report1 = [&](int k) {
    hipMemcpy(h.data(), out, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks * 4; ++i) s += (double)h[i];
    printf("one wave per SIMD, %2d v_fma_f32 per MFMA: %6.1f cycles per MFMA\n", k, s / (blocks * 4) / (iters * 4.0));
  };
#define RUN1(K) hipLaunchKernelGGL(one_wave<K>, dim3(blocks), dim3(256), 0, 0, out, sink, iters); hipDeviceSynchronize(); report1(K);
  RUN1(0) RUN1(0) RUN1(2) RUN1(4) RUN1(8) RUN1(12) RUN1(16) RUN1(24)
#define RUNK(K, KIND, WHAT) hipLaunchKernelGGL((one_wave<K, KIND>), dim3(blocks), dim3(256), 0, 0, out, sink, iters); hipDeviceSynchronize(); \
  { hipMemcpy(h.data(), out, blocks * 4 * 8, hipMemcpyDeviceToHost); double s = 0; for (int i = 0; i < blocks * 4; ++i) s += (double)h[i]; \
    printf("one wave per SIMD, %2d %-12s per MFMA: %6.1f cycles per MFMA\n", K, WHAT, s / (blocks * 4) / (iters * 4.0)); }
  RUNK(1, 1, "ds_read_b32") RUNK(2, 1, "ds_read_b32") RUNK(4, 1, "ds_read_b32") RUNK(8, 1, "ds_read_b32")
  RUNK(1, 3, "ds_read_b128") RUNK(2, 3, "ds_read_b128") RUNK(4, 3, "ds_read_b128")

  auto report2 = [&](const char* what, int per_iter) {
    hipMemcpy(h.data(), out, blocks * 8 * 8, hipMemcpyDeviceToHost);
    double sa = 0, sb = 0;
    for (int b = 0; b < blocks; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? sa : sb) += (double)h[b * 8 + w];
    printf("two waves per SIMD, partner %-12s: %6.1f cycles per MFMA", what, sa / (blocks * 4) / (iters * 4.0));
    if (per_iter) printf(", partner %6.1f cycles per instruction", sb / (blocks * 4) / (iters * (double)per_iter));
    printf("\n");
  };
  hipLaunchKernelGGL(two_waves<2>, dim3(blocks), dim3(512), 0, 0, out, sink, iters); hipDeviceSynchronize(); report2("idle", 0);
  hipLaunchKernelGGL(two_waves<0>, dim3(blocks), dim3(512), 0, 0, out, sink, iters); hipDeviceSynchronize(); report2("v_fma_f32", 16);
  hipLaunchKernelGGL(two_waves<1>, dim3(blocks), dim3(512), 0, 0, out, sink, iters); hipDeviceSynchronize(); report2("ds_read_b32", 16);
#define RUNM(K, NW) hipLaunchKernelGGL((mixed<K, NW>), dim3(blocks), dim3(NW * 256), 0, 0, out, sink, iters); hipDeviceSynchronize(); \
  { hipMemcpy(h.data(), out, blocks * 16 * 8, hipMemcpyDeviceToHost); double s = 0; for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4 * NW; ++w) s += (double)h[b * 16 + w]; \
    printf("%d waves per SIMD, each one MFMA + %2d v_fma_f32: %6.1f SIMD cycles per MFMA\n", NW, K, s / (blocks * 4 * NW) / (iters * 4.0) / NW); }
  RUNM(0, 2) RUNM(2, 2) RUNM(4, 2) RUNM(8, 2) RUNM(16, 2)
  return 0;
}
