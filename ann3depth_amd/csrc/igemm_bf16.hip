// igemm_bf16.hip — instantiations of the bf16 / bf16x3 implicit-GEMM kernel (igemm_bf16.h): BM = 128, BN in {128, 64}.
#include "a3d_internal.h"
#include "igemm_bf16.h"

namespace a3d {

template <int MODE, int BN, bool X3>
static int launch_bf16_one(IgemmParams& p, unsigned grid, hipStream_t st) {
  using Cfg = Bf16Cfg<MODE, 128, BN, X3>;
  auto kern = igemm_bf16_kernel<MODE, 128, BN, X3>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    if (e != hipSuccess) return set_error(A3D_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
  clear_stale_error();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, st, p);
  return check_launch("igemm_bf16");
}

template <int MODE>
static int launch_bf16_mode(int bn, bool x3, IgemmParams& p, unsigned grid, hipStream_t st) {
  if (bn == 128) return x3 ? launch_bf16_one<MODE, 128, true>(p, grid, st) : launch_bf16_one<MODE, 128, false>(p, grid, st);
  return x3 ? launch_bf16_one<MODE, 64, true>(p, grid, st) : launch_bf16_one<MODE, 64, false>(p, grid, st);
}

int launch_igemm_bf16(int mode, int bn, bool x3, IgemmParams& p, unsigned grid, hipStream_t st) {
  if (mode == MODE_FWD) return launch_bf16_mode<MODE_FWD>(bn, x3, p, grid, st);
  if (mode == MODE_BWD_D) return launch_bf16_mode<MODE_BWD_D>(bn, x3, p, grid, st);
  return launch_bf16_mode<MODE_BWD_F>(bn, x3, p, grid, st);
}

}  // namespace a3d
