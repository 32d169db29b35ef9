// igemm_bf16.hip — instantiations of the bf16 / bf16x3 implicit-GEMM kernel (igemm_bf16.h): BM = 128, BN in {128, 64},
// fp32 tensors (both precisions) and the bf16-storage operand combinations of BASELINE config 5.
#include "a3d_internal.h"
#include "igemm_bf16.h"

namespace a3d {

template <int MODE, int BN, bool X3, bool A16, bool B16, bool C16>
static int launch_bf16_one(IgemmParams& p, unsigned grid, hipStream_t st) {
  using Cfg = Bf16Cfg<MODE, 128, BN, X3>;
  auto kern = igemm_bf16_kernel<MODE, 128, BN, X3, A16, B16, C16>;
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

// storage combinations the path uses (a3d_conv_desc.storage): fp32 everywhere; weights (or a bf16 gradient) as B with an
// fp32 or bf16 A and an fp32 or bf16 output; a bf16 activation against an fp32 gradient (fine/second's filter gradient)
template <int MODE, int BN>
static int launch_bf16_bn(bool x3, IgemmParams& p, unsigned grid, hipStream_t st) {
  const int combo = (p.a16 ? 4 : 0) | (p.b16 ? 2 : 0) | (p.c16 ? 1 : 0);
  if (x3) {
    if (combo) return set_error(A3D_EINVAL, "igemm bf16x3: split operands need fp32 tensors");
    return launch_bf16_one<MODE, BN, true, false, false, false>(p, grid, st);
  }
  switch (combo) {
    case 0: return launch_bf16_one<MODE, BN, false, false, false, false>(p, grid, st);
    case 1: return launch_bf16_one<MODE, BN, false, false, false, true>(p, grid, st);
    case 2: return launch_bf16_one<MODE, BN, false, false, true, false>(p, grid, st);
    case 3: return launch_bf16_one<MODE, BN, false, false, true, true>(p, grid, st);
    case 4: return launch_bf16_one<MODE, BN, false, true, false, false>(p, grid, st);
    case 6: return launch_bf16_one<MODE, BN, false, true, true, false>(p, grid, st);
    case 7: return launch_bf16_one<MODE, BN, false, true, true, true>(p, grid, st);
  }
  return set_error(A3D_EINVAL, "igemm bf16: storage combination %d is not built", combo);
}

template <int MODE>
static int launch_bf16_mode(int bn, bool x3, IgemmParams& p, unsigned grid, hipStream_t st) {
  if (bn == 128) return launch_bf16_bn<MODE, 128>(x3, p, grid, st);
  return launch_bf16_bn<MODE, 64>(x3, p, grid, st);
}

// the parity classes of a strided bwd-data on bf16-stored tensors (dz, filter copy and dx all bf16) as one launch
template <int BN>
static int launch_bf16_multi_one(IgemmMulti& ps, unsigned grid_x, unsigned count, hipStream_t st) {
  using Cfg = Bf16Cfg<MODE_BWD_D, 128, BN, false>;
  auto kern = igemm_bf16_multi_kernel<MODE_BWD_D, 128, BN, false, true, true, true>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    if (e != hipSuccess) return set_error(A3D_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
  clear_stale_error();
  hipLaunchKernelGGL(kern, dim3(grid_x, count), dim3(Cfg::NT), Cfg::LDS_BYTES, st, ps);
  return check_launch("igemm_bf16_multi");
}
int launch_igemm_bf16_multi_bwd_d(int bn, IgemmMulti& ps, unsigned grid_x, unsigned count, hipStream_t st) {
  if (bn == 128) return launch_bf16_multi_one<128>(ps, grid_x, count, st);
  return launch_bf16_multi_one<64>(ps, grid_x, count, st);
}

int launch_igemm_bf16(int mode, int bn, bool x3, IgemmParams& p, unsigned grid, hipStream_t st) {
  if (mode == MODE_FWD) return launch_bf16_mode<MODE_FWD>(bn, x3, p, grid, st);
  if (mode == MODE_BWD_D) return launch_bf16_mode<MODE_BWD_D>(bn, x3, p, grid, st);
  return launch_bf16_mode<MODE_BWD_F>(bn, x3, p, grid, st);
}

}  // namespace a3d
