// igemm_ring.hip — instantiations of the LDS-DMA bf16 kernel for bf16-stored operands (igemm_ring.h).
#include "a3d_internal.h"
#include "igemm_ring.h"

namespace a3d {

// index, BM, BN, WAVES_M  (keep in step with kRingCfgs in igemm_host.hip)
#define A3D_RING_CFGS(X) X(0, 256, 128, 4) X(1, 256, 64, 8) X(2, 256, 256, 4) X(3, 128, 128, 4)

template <int MODE, int BM, int BN, int WAVES_M, bool C16>
static int launch_ring_one(IgemmParams& p, unsigned grid, hipStream_t st) {
  using Cfg = RingCfg<MODE, BM, BN, WAVES_M>;
  auto kern = igemm_ring_kernel<MODE, BM, BN, WAVES_M, C16>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)Cfg::LDS_BYTES);
    if (e != hipSuccess) return set_error(A3D_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
  clear_stale_error();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, st, p);
  return check_launch("igemm_ring");
}

template <int MODE>
static int launch_ring_mode(int cfg, IgemmParams& p, unsigned grid, hipStream_t st) {
  switch (cfg) {
#define X(i, bm, bn, wm) \
  case i: return p.c16 ? launch_ring_one<MODE, bm, bn, wm, true>(p, grid, st) : launch_ring_one<MODE, bm, bn, wm, false>(p, grid, st);
    A3D_RING_CFGS(X)
#undef X
  }
  if (MODE == MODE_BWD_D && cfg == 4)      // 96 input channels (conv2d_1's bwd-data): one 96-column tile, k-contiguous filter rows
    return p.c16 ? launch_ring_one<MODE_BWD_D, 256, 96, 8, true>(p, grid, st) : launch_ring_one<MODE_BWD_D, 256, 96, 8, false>(p, grid, st);
  return set_error(A3D_EINVAL, "igemm ring: unknown config %d", cfg);
}

int launch_igemm_ring(int mode, int cfg, IgemmParams& p, unsigned grid, hipStream_t st) {
  if (mode == MODE_FWD) return launch_ring_mode<MODE_FWD>(cfg, p, grid, st);
  if (mode == MODE_BWD_D) return launch_ring_mode<MODE_BWD_D>(cfg, p, grid, st);
  return set_error(A3D_EINVAL, "igemm ring: forward and bwd-data only");
}

}  // namespace a3d
