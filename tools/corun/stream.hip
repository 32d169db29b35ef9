// Test-only: a configurable streaming kernel (reads `n` float4, adds, writes back) to study how a bandwidth-bound grid of a
// given shape slows a GEMM of another stream.  Built by tools/corun/run.py with hipcc; never part of liba3d.so.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int UNROLL, bool NT>
__global__ void stream_kernel(f4* __restrict__ p, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * UNROLL;
  for (size_t i = (size_t)blockIdx.x * blockDim.x * UNROLL + threadIdx.x; i < n4; i += stride) {
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t j = i + (size_t)u * blockDim.x;
      if (j < n4) v[u] = NT ? __builtin_nontemporal_load(&p[j]) : p[j];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t j = i + (size_t)u * blockDim.x;
      if (j < n4) {
        v[u] += 1.0f;
        if (NT) __builtin_nontemporal_store(v[u], &p[j]); else p[j] = v[u];
      }
    }
  }
}
extern "C" int corun_stream(void* p, size_t n4, int grid, int block, int unroll, int nt, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  f4* q = static_cast<f4*>(p);
#define L(U, N) hipLaunchKernelGGL((stream_kernel<U, N>), dim3(grid), dim3(block), 0, st, q, n4)
  if (unroll == 8) { if (nt) L(8, true); else L(8, false); }
  else if (unroll == 4) { if (nt) L(4, true); else L(4, false); }
  else { if (nt) L(1, true); else L(1, false); }
  return (int)hipGetLastError();
}
