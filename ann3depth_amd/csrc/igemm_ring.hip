// igemm_ring.hip — instantiations of the LDS-DMA bf16 kernel for bf16-stored operands (igemm_ring.h).
#include <algorithm>

#include "a3d_internal.h"
#include "igemm_ring.h"

namespace a3d {

// index, BM, BN, WAVES_M  (keep in step with kRingCfgs in igemm_host.hip)
#define A3D_RING_CFGS(X) X(0, 256, 128, 4) X(1, 256, 64, 8) X(2, 256, 256, 4) X(3, 128, 128, 4) X(5, 512, 64, 8) X(6, 64, 128, 2)

// A3D_RING_M16 (tuning processes, read once): 0 = every launch on v_mfma_f32_32x32x16_bf16, 1 = on 16x16x32 wherever that form
// exists (wave tiles of at least 32 x 64), unset = the shipped choice per tile (ring_m16_default)
static int ring_m16_env() {
  static const int v = tune_int("A3D_RING_M16", -1);
  return v;
}

template <int MODE, int BM, int BN, int WAVES_M, bool C16, bool M16>
static int launch_ring_one(IgemmParams& p, unsigned grid, hipStream_t st) {
  using Cfg = RingCfg<MODE, BM, BN, WAVES_M>;
  auto kern = igemm_ring_kernel<MODE, BM, BN, WAVES_M, C16, M16>;
  // the raised LDS limit is a per-DEVICE attribute of the kernel: one flag per device (a benign race at worst sets it twice)
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)Cfg::LDS_BYTES);
    if (e != hipSuccess) return set_error(A3D_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  clear_stale_error();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, st, p);
  return check_launch("igemm_ring");
}

static bool ring_m16_default(int mode, int cfg) { return false; }

template <int MODE, int BM, int BN, int WAVES_M>
static int launch_ring_tile(int cfg, bool c16, IgemmParams& p, unsigned grid, hipStream_t st) {
  // The 16x16x32 form (igemm_ring.h, M16) is parity-green and measured: within 1 % of the 32x32x16 form on every layer
  // (DESIGN.md 6, round 4).  It is instantiated only in builds made with -DA3D_RING_M16 (twice the kernels to compile).
#ifdef A3D_RING_M16
  constexpr bool HAS16 = RingCfg<MODE, BM, BN, WAVES_M>::TN >= 2;      // it needs four 16-column sub-tiles per wave
#else
  constexpr bool HAS16 = false;
#endif
  const int env = ring_m16_env();
  const bool m16 = HAS16 && (env < 0 ? ring_m16_default(MODE, cfg) : env != 0);
  if constexpr (HAS16) {
    if (m16) {
      if constexpr (MODE != MODE_BWD_F)
        if (c16) return launch_ring_one<MODE, BM, BN, WAVES_M, true, true>(p, grid, st);
      return launch_ring_one<MODE, BM, BN, WAVES_M, false, true>(p, grid, st);
    }
  }
  if constexpr (MODE != MODE_BWD_F)
    if (c16) return launch_ring_one<MODE, BM, BN, WAVES_M, true, false>(p, grid, st);
  return launch_ring_one<MODE, BM, BN, WAVES_M, false, false>(p, grid, st);
}

template <int MODE>
static int launch_ring_mode(int cfg, IgemmParams& p, unsigned grid, hipStream_t st) {
  const bool c16 = p.c16 && p.splitk == 1;       // split-K slabs are float32; the reduction writes the bf16 tensor
  switch (cfg) {
#define X(i, bm, bn, wm) \
  case i: return launch_ring_tile<MODE, bm, bn, wm>(cfg, c16, p, grid, st);
    A3D_RING_CFGS(X)
#undef X
  }
  if (MODE == MODE_BWD_D && cfg == 4)      // 96 input channels (conv2d_1's bwd-data): one 96-column tile, k-contiguous filter rows
    return launch_ring_tile<MODE_BWD_D, 256, 96, 8>(cfg, c16, p, grid, st);
  return set_error(A3D_EINVAL, "igemm ring: unknown config %d", cfg);
}

int launch_igemm_ring(int mode, int cfg, IgemmParams& p, unsigned grid, hipStream_t st) {
  if (mode == MODE_FWD) return launch_ring_mode<MODE_FWD>(cfg, p, grid, st);
  if (mode == MODE_BWD_D) return launch_ring_mode<MODE_BWD_D>(cfg, p, grid, st);
  switch (cfg) {                               // bwd-filter: float32 gradient (or split-K slabs), tiles of at least 128 rows / 64 columns
    case 0: return launch_ring_tile<MODE_BWD_F, 256, 128, 4>(cfg, false, p, grid, st);
    case 1: return launch_ring_tile<MODE_BWD_F, 256, 64, 8>(cfg, false, p, grid, st);
    case 2: return launch_ring_tile<MODE_BWD_F, 256, 256, 4>(cfg, false, p, grid, st);
    case 3: return launch_ring_tile<MODE_BWD_F, 128, 128, 4>(cfg, false, p, grid, st);
  }
  return set_error(A3D_EINVAL, "igemm ring: unknown bwd-filter config %d", cfg);
}

// BiasAddGrad of a bf16 gradient tensor dz [rows][ld] (the LDS-DMA bwd-filter never sees dz in registers): column sums in
// float32.  Pass 1: block b sums its slab of rows, thread = one 16-byte piece (8 columns) of a row, rows of the slab in
// ascending order, the block's row groups added in group order through LDS -> partial[b][n].  Pass 2 adds the partials in
// block order.  The same bits on every run.
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const __bf16* __restrict__ dz, float* __restrict__ partial, int rows,
                                                          int n, int ld, int rows_per_block) {
  __shared__ float red[256 * 8];
  const int tpr = n / 8, groups = 256 / tpr;            // threads per row, rows in flight per block
  const int g = threadIdx.x / tpr, c = threadIdx.x % tpr;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  if (g < groups) {
    for (int r = r0 + g; r < r1; r += groups) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(dz + (size_t)r * ld + 8 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[2 * e] += __uint_as_float(v[e] << 16);
        s[2 * e + 1] += __uint_as_float(v[e] & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x * 8 + e] = s[e];
  __syncthreads();
  for (int col = threadIdx.x; col < n; col += 256) {
    const int cc = col / 8, e = col % 8;
    float t = 0.f;
    for (int gg = 0; gg < groups; ++gg) t += red[(gg * tpr + cc) * 8 + e];
    partial[(size_t)blockIdx.x * n + col] = t;
  }
}
// 64 columns per block; four threads per column take the partials b = part, part + 4, ... (eight loads in flight, added in
// ascending b), their four sums meet in LDS in part order
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ partial, float* __restrict__ out, int n, int blocks) {
  __shared__ float red[256];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
  float t = 0.f;
  if (col < n) {
    int b = part;
    for (; b + 28 < blocks; b += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(b + 4 * u) * n + col];
#pragma unroll
      for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; b < blocks; b += 4) t += partial[(size_t)b * n + col];
  }
  red[threadIdx.x] = t;
  __syncthreads();
  if (part == 0 && col < n) out[col] = ((red[threadIdx.x] + red[threadIdx.x + 64]) + red[threadIdx.x + 128]) + red[threadIdx.x + 192];
}
size_t colsum_bf16_ws_bytes(int n) { return (size_t)256 * n * 4; }
bool colsum_bf16_ok(int n) { return n % 8 == 0 && n >= 8 && n / 8 <= 256; }
// out == nullptr: pass 1 only — the caller adds the *nparts rows of `ws` itself (the split-K reduction of the launch beside it)
int colsum_bf16(const void* dz, int rows, int n, int ld, float* out, void* ws, int* nparts, hipStream_t st) {
  const int blocks = std::min(256, std::max(1, rows / 64));
  const int rpb = (rows + blocks - 1) / blocks;
  const int used = (rows + rpb - 1) / rpb;
  clear_stale_error();
  hipLaunchKernelGGL(colsum_bf16_kernel, dim3(used), dim3(256), 0, st, static_cast<const __bf16*>(dz), static_cast<float*>(ws), rows, n,
                     ld, rpb);
  int rc = check_launch("colsum_bf16");
  if (nparts) *nparts = used;
  if (rc != A3D_OK || !out) return rc;
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((n + 63) / 64), dim3(256), 0, st, static_cast<const float*>(ws), out, n, used);
  return check_launch("colsum_finish");
}

}  // namespace a3d
