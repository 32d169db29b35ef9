// igemm_inst.h — instantiates every (tile config x vector width) of one MODE; included by igemm_{fwd,bwd_d,bwd_f}.hip
// with A3D_MODE defined, so the three modes compile in parallel.
#include "a3d_internal.h"
#include "igemm.h"
#include "igemm_glds.h"
#include "igemm2.h"

namespace a3d {

// index, BM, BN, WAVES_M, NWAVES, BK  (keep in step with kCfgs in igemm_host.hip)
#define A3D_CFGS(X) X(0, 128, 128, 2, 4, 32) X(1, 128, 96, 4, 4, 32) X(2, 128, 64, 4, 4, 32) X(3, 128, 32, 4, 4, 32) \
                    X(4, 64, 64, 2, 4, 32) X(5, 32, 128, 1, 4, 32) X(6, 64, 128, 1, 4, 32) X(7, 128, 128, 4, 8, 32) \
                    X(8, 128, 64, 4, 8, 32)

// A3D_HINT_SHARE_CU: the dynamic LDS request is raised until only 8 / NWAVES blocks (two wavefronts per SIMD) fit the
// CU's 160 KiB, so the other stream's bandwidth-bound kernels find free registers and wave slots on every CU.
template <int NWAVES>
constexpr size_t share_lds_bytes(size_t need) {
  constexpr size_t blocks = NWAVES >= 8 ? 1 : 8 / NWAVES;
  const size_t pad = (size_t)163840 / (blocks + 1) + 1024;
  return need > pad ? need : pad;
}

template <int BM, int BN, int WAVES_M, int NWAVES, int BK, int AVEC, int BVEC>
static int launch_one(IgemmParams& p, unsigned grid, hipStream_t st) {
  using Cfg = IgemmCfg<A3D_MODE, BM, BN, WAVES_M, NWAVES, BK, AVEC, BVEC>;
  auto kern = igemm_kernel<A3D_MODE, BM, BN, WAVES_M, NWAVES, BK, AVEC, BVEC>;
  constexpr size_t kShared = share_lds_bytes<NWAVES>(Cfg::LDS_BYTES);
  static bool attr_done = false;   // idempotent attribute, benign if set twice
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kShared);
    if (e != hipSuccess) return set_error(A3D_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
  clear_stale_error();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::NT), p.share ? kShared : Cfg::LDS_BYTES, st, p);
  return check_launch("igemm");
}

template <int BM, int BN, int WAVES_M, int NWAVES, int BK>
static int launch_vec(int avec, int bvec, IgemmParams& p, unsigned grid, hipStream_t st) {
  if (avec == 4 && bvec == 4) return launch_one<BM, BN, WAVES_M, NWAVES, BK, 4, 4>(p, grid, st);
  if (avec == 4 && bvec == 1) return launch_one<BM, BN, WAVES_M, NWAVES, BK, 4, 1>(p, grid, st);
  if (avec == 1 && bvec == 4) return launch_one<BM, BN, WAVES_M, NWAVES, BK, 1, 4>(p, grid, st);
  if (avec == 2 && bvec == 4) return launch_one<BM, BN, WAVES_M, NWAVES, BK, 2, 4>(p, grid, st);   // 8-byte window runs
  if (avec == 2 && bvec == 1) return launch_one<BM, BN, WAVES_M, NWAVES, BK, 2, 1>(p, grid, st);
  return launch_one<BM, BN, WAVES_M, NWAVES, BK, 1, 1>(p, grid, st);
}

// LDS-DMA staged variants (igemm_glds.h): index, BM, BN, WAVES_M, NWAVES, index of the register-staged twin used
// when an operand is not 16-byte vectorisable
#define A3D_GLDS_CFGS(X) X(9, 128, 128, 4, 8, 7) X(10, 128, 64, 4, 8, 8)

template <int BM, int BN, int WAVES_M, int NWAVES>
static int launch_glds(IgemmParams& p, unsigned grid, hipStream_t st) {
  using Cfg = GldsCfg<A3D_MODE, BM, BN, WAVES_M, NWAVES>;
  auto kern = igemm_glds_kernel<A3D_MODE, BM, BN, WAVES_M, NWAVES>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    if (e != hipSuccess) return set_error(A3D_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
  clear_stale_error();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, st, p);
  return check_launch("igemm_glds");
}

// second-generation kernel (igemm2.h): config index 11, fix-ups and register-staged twin of config 0 (same wave layout)
constexpr int kGen2CfgIndex = 11;
static bool gen2_applicable(const IgemmParams& p, int avec, int bvec) {
  if (avec != 4 || bvec != 4 || p.a16 || p.b16 || p.c16) return false;
  if (A3D_MODE == MODE_BWD_F) return p.Cg % 4 == 0 && p.N % 4 == 0;
  return p.uni && p.Cg % 32 == 0 && p.stride >= 1 && (A3D_MODE == MODE_FWD || p.stride == 1);
}
static int launch_gen2(IgemmParams& p, unsigned grid, hipStream_t st) {
  using Cfg = Gen2Cfg<A3D_MODE>;
  auto kern = igemm2_kernel<A3D_MODE>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    if (e != hipSuccess) return set_error(A3D_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
  clear_stale_error();
  hipLaunchKernelGGL(kern, dim3(grid), dim3(G2_NT), Cfg::LDS_BYTES, st, p);
  return check_launch("igemm2");
}

template <int BM, int BN, int WAVES_M, int NWAVES>
static int launch_fixup_one(IgemmParams& p, unsigned tiles, unsigned nblk, hipStream_t st) {
  clear_stale_error();
  hipLaunchKernelGGL((igemm_fixup_kernel<A3D_MODE, BM, BN, WAVES_M, NWAVES>), dim3(tiles, (BM * BN / 256 + 3) / 4), dim3(256),
                     0, st, p, nblk);
  return check_launch("igemm_fixup");
}

#if A3D_MODE == 1
// multi-problem launch (parity classes of a strided bwd-data): 64x64 4-wave tiles only
template <int AVEC, int BVEC>
static int launch_multi_one(IgemmMulti& ps, unsigned grid_x, unsigned count, hipStream_t st) {
  using Cfg = IgemmCfg<MODE_BWD_D, 64, 64, 2, 4, 32, AVEC, BVEC>;
  auto kern = igemm_multi_kernel<MODE_BWD_D, 64, 64, 2, 4, 32, AVEC, BVEC>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cfg::LDS_BYTES);
    if (e != hipSuccess) return set_error(A3D_ELAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    attr_done = true;
  }
  clear_stale_error();
  hipLaunchKernelGGL(kern, dim3(grid_x, count), dim3(Cfg::NT), Cfg::LDS_BYTES, st, ps);
  return check_launch("igemm_multi");
}
int launch_igemm_multi_bwd_d(int avec, int bvec, IgemmMulti& ps, unsigned grid_x, unsigned count, hipStream_t st) {
  if (avec == 4 && bvec == 4) return launch_multi_one<4, 4>(ps, grid_x, count, st);
  if (avec == 4 && bvec == 1) return launch_multi_one<4, 1>(ps, grid_x, count, st);
  if (avec == 1 && bvec == 4) return launch_multi_one<1, 4>(ps, grid_x, count, st);
  return launch_multi_one<1, 1>(ps, grid_x, count, st);
}
#endif

#define A3D_CAT_(a, b) a##b
#define A3D_CAT(a, b) A3D_CAT_(a, b)

// stream-K fixup of a launch made with register-staged config `cfg`
int A3D_CAT(launch_fixup_mode, A3D_MODE)(int cfg, IgemmParams& p, unsigned tiles, unsigned nblk, hipStream_t st) {
  if (cfg == kGen2CfgIndex) cfg = 0;
  switch (cfg) {
#define X(i, bm, bn, wm, nw, bk) \
  case i: return launch_fixup_one<bm, bn, wm, nw>(p, tiles, nblk, st);
    A3D_CFGS(X)
#undef X
  }
  return set_error(A3D_EINVAL, "igemm fixup: unknown config %d", cfg);
}

int A3D_CAT(launch_igemm_mode, A3D_MODE)(int cfg, int avec, int bvec, IgemmParams& p, unsigned grid, hipStream_t st) {
  if (cfg == kGen2CfgIndex) {
    if (gen2_applicable(p, avec, bvec)) return launch_gen2(p, grid, st);
    cfg = 0;
  }
  switch (cfg) {
#define X(i, bm, bn, wm, nw, twin) \
  case i:                          \
    if (avec == 4 && bvec == 4) return launch_glds<bm, bn, wm, nw>(p, grid, st); \
    cfg = twin;                    \
    break;
    A3D_GLDS_CFGS(X)
#undef X
  }
  switch (cfg) {
#define X(i, bm, bn, wm, nw, bk) \
  case i: return launch_vec<bm, bn, wm, nw, bk>(avec, bvec, p, grid, st);
    A3D_CFGS(X)
#undef X
  }
  return set_error(A3D_EINVAL, "igemm: unknown config %d", cfg);
}

}  // namespace a3d
