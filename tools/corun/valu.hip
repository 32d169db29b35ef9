// Test-only: a streaming kernel with VALU work per byte, to study whether a weight stream whose arithmetic runs on the
// VECTOR pipe (32 fp32 FMAs per weight element, the dense layers' ratio at batch 32) keeps its bandwidth beside an
// MFMA-bound GEMM of another stream — the matrix pipe is the GEMM's, the vector pipe is mostly idle under it.
// mode 0: read-modify-write stream (the fused dW + ApplyAdam's shape: 16 B in, 16 B out, 128 FMAs per lane and piece)
// mode 1: read-only stream (the forward's shape: 16 B in, 128 FMAs, one store per lane at the end)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
struct Coef { float s[32]; };
template <int FMAS, bool RMW>
__global__ __launch_bounds__(256) void valu_stream_kernel(f4* __restrict__ p, size_t n4, Coef c, float* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  f4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  f4 cur = i < n4 ? __builtin_nontemporal_load(&p[i]) : f4{0.f, 0.f, 0.f, 0.f};
  for (; i < n4; i += stride) {
    const size_t nx = i + stride;
    const f4 nxt = nx < n4 ? __builtin_nontemporal_load(&p[nx]) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < FMAS; ++b) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[b & 7][j] = __builtin_fmaf(c.s[b & 31], cur[j], acc[b & 7][j]);
    }
    if (RMW) {
      f4 o = acc[0];
#pragma unroll
      for (int q = 1; q < 8; ++q) o += acc[q];
      __builtin_nontemporal_store(o, &p[i]);
    }
    cur = nxt;
  }
  if (!RMW) {
    f4 o = acc[0];
#pragma unroll
    for (int q = 1; q < 8; ++q) o += acc[q];
    sink[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = o[0] + o[1] + o[2] + o[3];
  }
}
extern "C" int corun_valu(void* p, size_t n4, int grid, int fmas, int rmw, void* sink, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  Coef c;
  for (int i = 0; i < 32; ++i) c.s[i] = 1e-3f * (i + 1);
  f4* q = static_cast<f4*>(p);
  float* sk = static_cast<float*>(sink);
#define L(F, R) hipLaunchKernelGGL((valu_stream_kernel<F, R>), dim3(grid), dim3(256), 0, st, q, n4, c, sk)
  if (fmas == 0) { if (rmw) L(0, true); else L(0, false); }
  else if (fmas == 16) { if (rmw) L(16, true); else L(16, false); }
  else { if (rmw) L(32, true); else L(32, false); }
  return (int)hipGetLastError();
}
