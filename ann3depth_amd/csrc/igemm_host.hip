// igemm_host.hip — tile/split-K planning, dispatch, split-K reduction, bias-gradient column sums, and the
// conv2d / dense entry points of include/a3d.h.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

#include "a3d_internal.h"
#include "igemm.h"

namespace a3d {

int launch_igemm_mode0(int cfg, int avec, int bvec, IgemmParams& p, unsigned grid, hipStream_t st);
int launch_igemm_mode1(int cfg, int avec, int bvec, IgemmParams& p, unsigned grid, hipStream_t st);
int launch_igemm_mode2(int cfg, int avec, int bvec, IgemmParams& p, unsigned grid, hipStream_t st);
int launch_igemm_bf16(int mode, int bn, bool x3, IgemmParams& p, unsigned grid, hipStream_t st);   // p.a16/b16/c16 pick the storage variant
int launch_fixup_mode0(int cfg, IgemmParams& p, unsigned tiles, unsigned nblk, hipStream_t st);
int launch_fixup_mode1(int cfg, IgemmParams& p, unsigned tiles, unsigned nblk, hipStream_t st);
int launch_fixup_mode2(int cfg, IgemmParams& p, unsigned tiles, unsigned nblk, hipStream_t st);
int launch_igemm_multi_bwd_d(int avec, int bvec, IgemmMulti& ps, unsigned grid_x, unsigned count, hipStream_t st);
int launch_igemm_bf16_multi_bwd_d(int bn, IgemmMulti& ps, unsigned grid_x, unsigned count, hipStream_t st);
int launch_igemm_ring(int mode, int cfg, IgemmParams& p, unsigned grid, hipStream_t st);

// tile configurations of the LDS-DMA bf16 kernel (igemm_ring.hip: A3D_RING_CFGS + the 96-column bwd-data tile)
struct RingTile { int bm, bn; };
static const RingTile kRingCfgs[] = {{256, 128}, {256, 64}, {256, 256}, {128, 128}, {256, 96}, {512, 64}, {64, 128}};

struct TileCfg {
  int bm, bn;
  float eff;   // relative efficiency of the configuration, fitted (tools/fit_planner.py); 0 = only via A3D_FORCE_CFG
  int bk;
};
static const TileCfg kCfgs[] = {{128, 128, 1.00f, 32}, {128, 96, 1.00f, 32}, {128, 64, 0.92f, 32}, {128, 32, 0.60f, 32},
                                {64, 64, 0.98f, 32},   {32, 128, 0.90f, 32}, {64, 128, 1.00f, 32},
                                {128, 128, 1.15f, 32}, {128, 64, 1.05f, 32},      // 8-wave blocks
                                // LDS-DMA staged (igemm_glds.h), same order as A3D_GLDS_CFGS; forward only.  Round 1:
                                // +3-5 % over the register-staged twins; since those stage through buffer loads with
                                // addresses computed a tile ahead (round 2) they are the faster ones (fine/second
                                // forward 245 vs 259 us, conv2d_1 325 vs 339 us: profiles/r02_sweep_hot.txt)
                                {128, 128, 1.10f, 32}, {128, 64, 1.00f, 32},
                                // second-generation kernel (igemm2.h: LDS-DMA, 4 waves of 64x64, pinned schedule), all three
                                // modes; takes a launch only when gen2_ok (GemmProblem) — else its register-staged twin, cfg 0.
                                // Round 6: parity-green and measured on every MSDN layer — 2-5 % behind the 8-wave kernels on
                                // forward / bwd-data, 15 % behind on bwd-filter (DESIGN.md 3.1i says why): eff 0 = pinned
                                // plans only (A3D_FORCE_CFG=11 in a tuning process; tools/gen2_check.py, tools/sweep_hot.py)
                                {128, 128, 0.f, 32}};
static const int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);
static const int kSlots = 512;               // 256 CUs x 2 resident blocks
static const size_t kMaxSlabBytes = (size_t)192 << 20;

#ifdef A3D_STAMPS
static unsigned long long* g_stamps = nullptr;
static unsigned g_stamp_grid = 0;
static const size_t kStampBytes = (size_t)8 << 20;
#endif

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// Tuning aids (tools/sweep_igemm.py): A3D_FORCE_CFG / A3D_FORCE_SPLITK pin the tile config / split-K factor.  Every switch
// below goes through tune_int(): none of them is read in a process that was not started with A3D_TUNING=1.
// A/B switches of the staging fast paths (tools/bench_layers.py); read once per tuning process
// The A3D_FORCE_* / A3D_NO_* / A3D_BF16_BN switches of the sweep and fuzz tools are consulted on every launch ONLY in
// a process started with A3D_TUNING=1 (the tools set it); otherwise nothing below reads the environment after its first
// call, and plans are cached per problem (plan_gemm).
// (tuning() / tune_int(): a3d_internal.h, capi.cc)
static bool env_flag_no_uni() { static const bool v = tune_int("A3D_NO_UNI", 0) != 0; return v; }
// K-sliced stream-K is built and fuzzed but OFF by default: it removes the bwd-filter launches' fabric over-fetch and
// costs 0.7 % of the step; A3D_SK_SLICED=1 turns it on
static bool env_flag_no_sliced() { static const bool v = tune_int("A3D_SK_SLICED", 0) == 0; return v; }
// K-sliced stream-K (SkSpace): a block's share must fit inside one slice of one tile's K range (at most two partial
// tiles per block = its two slab slots) and a tile may not have more contributors than the fixup lists (96).
static bool sk_sliced_ok(int mode, long tiles, int nk, int grid, long* meet_out) {
  if (mode != MODE_BWD_F || grid % 8 != 0 || nk < 64) return false;
  const long q = grid / 8, nk_lo = nk / 8, nk_hi = (nk + 7) / 8;
  const long per = (tiles * nk_hi + q - 1) / q;                 // longest share of any XCD
  if (per > nk_lo || per < 1) return false;
  const long meet = 8 * ((nk_hi + per - 1) / per + 1);
  if (meet > 256) return false;
  if (meet_out) *meet_out = meet;
  return true;
}
static bool env_flag_no_streamk() { static const bool v = tune_int("A3D_NO_STREAMK", 0) != 0; return v; }
static bool env_flag_no_kperm() { static const bool v = tune_int("A3D_NO_KPERM", 0) != 0; return v; }

// ---- opt-in launch timing (a3d_timing_*) ----
struct TimingSlot {
  hipEvent_t start, stop;
  a3d_timing_record rec;
};
static std::mutex g_timing_mu;
static bool g_timing_on = false;
static bool g_timing_only = false;          // a3d_timing_select: bracket only launches of one kernel
static a3d_timing_record g_timing_like{};
static std::vector<TimingSlot> g_timing;

// Is this launch to be bracketed?  `r` names the kernel about to be launched.
static bool timing_wanted(const a3d_timing_record& r) {
  std::lock_guard<std::mutex> lk(g_timing_mu);
  if (!g_timing_on) return false;
  if (!g_timing_only) return true;
  const a3d_timing_record& l = g_timing_like;
  return r.mode == l.mode && r.prec == l.prec && r.bm == l.bm && r.bn == l.bn && r.waves_m == l.waves_m &&
         r.nwaves == l.nwaves && r.bk == l.bk && r.avec == l.avec && r.bvec == l.bvec && r.lds_dma == l.lds_dma;
}
static int timing_begin(TimingSlot& slot, hipStream_t st) {
  if (hipEventCreate(&slot.start) != hipSuccess || hipEventCreate(&slot.stop) != hipSuccess)
    return set_error(A3D_ELAUNCH, "timing: hipEventCreate failed");
  (void)hipEventRecord(slot.start, st);
  return A3D_OK;
}
static void timing_end(TimingSlot& slot, hipStream_t st) {
  (void)hipEventRecord(slot.stop, st);
  std::lock_guard<std::mutex> lk(g_timing_mu);
  g_timing.push_back(slot);
}
static const int kCfgWavesM[] = {2, 4, 4, 4, 2, 1, 1, 4, 4, 4, 4, 2};
static const int kCfgNWaves[] = {4, 4, 4, 4, 4, 4, 4, 8, 8, 8, 8, 4};
static const int kFirstGldsCfg = 9;
static const int kGen2Cfg = 11;
// the forward-only LDS-DMA kernels of round 1 (igemm_glds.h)
static inline bool is_glds_cfg(int c) { return c >= kFirstGldsCfg && c < kGen2Cfg; }
// configurations that run stream-K shares (slabs + igemm_fixup_kernel)
static inline bool streamk_cfg(int c) { return c < kFirstGldsCfg || c == kGen2Cfg; }
static bool env_flag_no_gen2() { static const bool v = tune_int("A3D_NO_GEN2", 0) != 0; return v; }

// bf16 / bf16x3 kernel: BM = 128.  Staging-bound rather than MFMA-bound: the wider tile wins whenever N allows it
// (even at one block per CU for the two-plane x3 variant); split-K factors: x3 by round 1's sweep
// (profiles/r01_sweep_bf16.txt: ~600 blocks with >= 12 k-tiles each), plain bf16 by round 3's, below.
// bf16-stored operands, forward / stride-1 bwd-data: igemm_ring.h.  One block of eight waves per CU works on a 256-row tile
// (128 rows, two blocks per CU, where 256-row tiles would leave a third of the chip idle); never split (a launch that small
// stays on igemm_bf16's split-K).  A3D_RING=0 turns the kernel off, A3D_RING_CFG pins a tile (tuning processes).
static bool ring_plan(const GemmProblem& g, GemmPlan& pl) {
  static const bool off = tune_int("A3D_RING", 1) == 0;
  if (!g.ring_ok || (g.plain && g.mode != MODE_FWD)) return false;      // (plain = the forward with the 2x2 max pool fused: never split)
  if (g.mode != MODE_BWD_F && g.M <= 64 && g.N >= 1024 && g.K >= 1024) {      // (not under A3D_RING: bf16 x / dz have no other kernel)
    // a dense layer of a small batch: a weight stream.  64-row tiles of 128 columns, K split until ~512 blocks (three per
    // CU) pull on HBM; the slabs are a few MB
    pl.ring = 1 + 6;
    pl.tiles_m = 1;
    pl.tiles_n = (g.N + 127) / 128;
    const int nk = std::max(1, (g.K + 63) / 64);
    int splitk = (int)std::min<long>(std::max<long>(512 / pl.tiles_n, 1), std::max(1, nk / 4));
    if (tune_int("A3D_FORCE_SPLITK", 0) > 0) splitk = std::min(tune_int("A3D_FORCE_SPLITK", 0), nk);
    if (g.need_reduce && nk >= 2) splitk = std::max(splitk, 2);      // rows narrower than the GEMM: stored by the reduction
    const int kps = (nk + splitk - 1) / splitk;
    pl.splitk = (nk + kps - 1) / kps;
    pl.ktiles_per_split = kps;
    pl.ws_bytes = pl.splitk > 1 ? (size_t)pl.splitk * g.M * g.N * 4 : 0;
    return true;
  }
  if (off && !g.plain) return false;              // (a pooled forward on bf16 inputs has no other kernel: ADVICE r4)
  if (g.need_reduce) return false;                // the tiles below are never split: igemm_bf16's split-K stores the narrower rows
  if (g.mode == MODE_BWD_F) {
    // filter gradient: few tiles (M = r s Cin rows) over a long pixel axis — 256-row tiles, split-K until every CU has one block
    static const bool off_f = tune_int("A3D_RING_BWDF", 1) == 0;
    if (off_f || g.M < 512 || g.N < 64 || g.K < 64 * 64) return false;      // (short pixel axes stay with igemm_bf16's finer tiles)
    int cfg = g.N <= 64 ? 1 : (g.N % 256 == 0 ? 2 : 0);
    const int forced = tune_int("A3D_RING_CFG", -1);
    if (forced >= 0 && forced <= 3) cfg = forced;
    const int bm = kRingCfgs[cfg].bm, bn = kRingCfgs[cfg].bn;
    pl.tiles_m = (g.M + bm - 1) / bm;
    pl.tiles_n = (g.N + bn - 1) / bn;
    const long tiles = (long)pl.tiles_m * pl.tiles_n;
    const int nk = std::max(1, (g.K + 63) / 64);
    int splitk = (int)std::min<long>(std::max<long>(256 / tiles, 1), std::max(1, nk / 8));
    if (tune_int("A3D_FORCE_SPLITK", 0) > 0) splitk = std::min(tune_int("A3D_FORCE_SPLITK", 0), nk);
    while (splitk > 1 && (size_t)splitk * g.M * g.N * 4 > kMaxSlabBytes) --splitk;
    const int kps = (nk + splitk - 1) / splitk;
    pl.ring = 1 + cfg;
    pl.splitk = (nk + kps - 1) / kps;
    pl.ktiles_per_split = kps;
    pl.ws_bytes = pl.splitk > 1 ? (size_t)pl.splitk * g.M * g.N * 4 : 0;
    return true;
  }
  int cfg;
  if (g.mode == MODE_BWD_D && g.N <= 96 && g.N > 64) cfg = 4;
  else if (g.N <= 64) cfg = (g.M + 511) / 512 >= 192 ? 5 : 1;      // 64 columns: 512-row tiles (wave tiles of 64 x 64) when they fill the chip
  else if (g.N % 256 == 0 && (long)((g.M + 255) / 256) * (g.N / 256) >= 192) cfg = 2;
  else cfg = (long)((g.M + 255) / 256) * ((g.N + 127) / 128) >= 192 ? 0 : 3;
  const int forced = tune_int("A3D_RING_CFG", -1);
  if (forced >= 0 && forced <= 6 && !(forced == 4 && g.mode != MODE_BWD_D) && !(forced == 6 && g.plain)) cfg = forced;
  const int bm = kRingCfgs[cfg].bm, bn = kRingCfgs[cfg].bn;
  const long tiles = (long)((g.M + bm - 1) / bm) * ((g.N + bn - 1) / bn);
  if (tiles < 96 && forced < 0 && !g.plain) return false;      // (a pooled forward on bf16 inputs has no other kernel)
  pl.ring = 1 + cfg;
  pl.tiles_m = (g.M + bm - 1) / bm;
  pl.tiles_n = (g.N + bn - 1) / bn;
  pl.splitk = 1;
  pl.ktiles_per_split = std::max(1, (g.K + 63) / 64);
  pl.ws_bytes = 0;
  return true;
}

static GemmPlan plan_gemm_bf16(const GemmProblem& g, int precision) {
  GemmPlan pl{};
  pl.prec = precision;
  if (precision == A3D_PREC_BF16 && ring_plan(g, pl)) return pl;
  pl.bf16_bn = g.N <= 64 ? 64 : 128;
  if (tune_int("A3D_BF16_BN", 0) == 64 || tune_int("A3D_BF16_BN", 0) == 128) pl.bf16_bn = tune_int("A3D_BF16_BN", 0);
  pl.tiles_m = (g.M + 127) / 128;
  pl.tiles_n = (g.N + pl.bf16_bn - 1) / pl.bf16_bn;
  const int bk = precision == A3D_PREC_BF16X3 ? 32 : 64;        // Bf16Cfg::BK: the plain bf16 kernel takes k-tiles of 64
  const int nk = std::max(1, (g.K + bk - 1) / bk);
  const long tiles = (long)pl.tiles_m * pl.tiles_n;
  int splitk = (int)std::min<long>(std::max<long>(768 / tiles, 1), std::max(1, nk * (bk / 32) / 12));
  if (precision == A3D_PREC_BF16) {
    // Plain bf16 kernel, measured (profiles/r03_bf16_splitk_sweep.txt): one 128 x 128 block keeps a CU's staging path
    // nearly as busy as two do (0.87 us per k-tile alone, 1.47 us each when two share the CU), so splitting K pays only
    // until every CU has ONE block (two of the lighter 128 x 64 ones); past that it just adds a slab of the whole
    // output per factor (conv2d_2 / conv2d_3 forward and bwd-data at batch 64: 351 tiles, 53 / 71 us unsplit against
    // 72 / 85 us split in two).
    const long target = pl.bf16_bn == 128 ? 256 : 512;
    splitk = (int)std::min<long>(std::max<long>(target / tiles, 1), std::max(1, nk / 4));
  }
  if (tune_int("A3D_FORCE_SPLITK", 0) > 0) splitk = std::min(tune_int("A3D_FORCE_SPLITK", 0), std::max(1, nk));
  if (g.need_reduce && nk >= 2) splitk = std::max(splitk, 2);
  if (g.plain) splitk = 1;                      // fused pool: whole K ranges only
  while (splitk > 1 && (size_t)splitk * g.M * g.N * 4 > kMaxSlabBytes) --splitk;
  const int kps = (nk + splitk - 1) / splitk;
  pl.splitk = (nk + kps - 1) / kps;
  pl.ktiles_per_split = kps;
  pl.ws_bytes = pl.splitk > 1 ? (size_t)pl.splitk * g.M * g.N * 4 : 0;
  return pl;
}

static GemmPlan plan_gemm_search(const GemmProblem& g, int precision);

// The search below walks 11 tile configurations x ~50 split factors: once per distinct problem, not once per launch
// (a training step launches the same ~50 problems over and over).
GemmPlan plan_gemm(const GemmProblem& g, int precision) {
  if (tuning()) return plan_gemm_search(g, precision);
  static std::mutex mu;
  static std::map<std::array<int, 11>, GemmPlan> cache;
  const std::array<int, 11> key = {g.mode, g.M, g.N, g.K, g.avec, g.bvec, g.plain + 2 * g.need_reduce, g.no_glds, precision, g.ring_ok, g.gen2_ok};
  std::lock_guard<std::mutex> lk(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  const GemmPlan plan = plan_gemm_search(g, precision);
  cache.emplace(key, plan);
  return plan;
}

// Measured winners for shapes where the cost model below loses more than 3 % to a configuration of the sweep
// (tools/sweep_hot.py --assert-auto-within 0.03; profiles/r03_sweep_hot.txt).  The model's block-slot count is that of
// the 8-wave tiles (two per CU); 4-wave 64x64 blocks sit four to a CU and want several thousand blocks when a handful
// of tiles carry a K of 10^5 — fine/second's bwd-filter: 64x64 x 128 splits 290 us, the model's pick 315-363 us.
struct TunedPlan { int mode, M, N, K, cfg, splitk; };
static const TunedPlan kTuned[] = {
    {MODE_BWD_F, 1600, 64, 130240, 4, 128},        // fine/second/conv2d bwd-filter at batch 32
};

static GemmPlan plan_gemm_search(const GemmProblem& g, int precision) {
  if (precision != A3D_PREC_F32 && g.avec == 4 && g.bvec == 4) return plan_gemm_bf16(g, precision);
  if (tune_int("A3D_FORCE_CFG", -1) < 0 && !g.plain) {
    for (const TunedPlan& t : kTuned) {
      if (t.mode != g.mode || t.M != g.M || t.N != g.N || t.K != g.K) continue;
      GemmPlan pl{};
      const int nk = std::max(1, (g.K + 31) / 32), kps = (nk + t.splitk - 1) / t.splitk;
      pl.cfg = t.cfg; pl.ktiles_per_split = kps; pl.splitk = (nk + kps - 1) / kps;
      pl.tiles_m = (g.M + kCfgs[t.cfg].bm - 1) / kCfgs[t.cfg].bm; pl.tiles_n = (g.N + kCfgs[t.cfg].bn - 1) / kCfgs[t.cfg].bn;
      pl.ws_bytes = pl.splitk > 1 ? (size_t)pl.splitk * g.M * g.N * 4 : 0;
      if (pl.ws_bytes <= kMaxSlabBytes) return pl;
    }
  }
  GemmPlan best{};
  double best_t = 1e300;
  const int nk = std::max(1, (g.K + 31) / 32);
  const int force_cfg = tune_int("A3D_FORCE_CFG", -1), force_split = tune_int("A3D_FORCE_SPLITK", -1);
  const int force_streamk = std::min(tune_int("A3D_FORCE_STREAMK", 0), 1024);      // tuning aid: stream-K with this many blocks (the fixup lists at most 1024 contributors per tile)
  if (force_cfg >= 0 && force_cfg < kNumCfgs) {
    // a pinned configuration still has to be one this problem can run on: the LDS-DMA kernels are forward-only, take
    // 16-byte operands, and have neither the pooling epilogue nor a bf16 output — their register-staged twins do
    int cfg = force_cfg;
    if (is_glds_cfg(cfg) && (g.mode != MODE_FWD || g.avec != 4 || g.bvec != 4 || g.plain || g.no_glds)) cfg -= 2;
    if (cfg == kGen2Cfg && !g.gen2_ok) cfg = 0;
    const int force_cfg = cfg;
    const int bm = kCfgs[force_cfg].bm, bn = kCfgs[force_cfg].bn;
    const int nk = std::max(1, (g.K + kCfgs[force_cfg].bk - 1) / kCfgs[force_cfg].bk);
    int splitk = std::max(1, std::min(force_split > 0 ? force_split : 1, nk));
    if (g.need_reduce && nk >= 2) splitk = std::max(splitk, 2);      // a narrower output is stored by the split-K reduction
    if (g.plain) splitk = 1;                     // the fused pool takes whole K ranges
    while (splitk > 1 && (size_t)splitk * g.M * g.N * 4 > kMaxSlabBytes) --splitk;
    int kps = (nk + splitk - 1) / splitk;
    splitk = (nk + kps - 1) / kps;
    best.cfg = force_cfg; best.splitk = splitk; best.ktiles_per_split = kps;
    best.tiles_m = (g.M + bm - 1) / bm; best.tiles_n = (g.N + bn - 1) / bn;
    best.ws_bytes = splitk > 1 ? (size_t)splitk * g.M * g.N * 4 : 0;
    if (force_streamk > 0 && streamk_cfg(force_cfg) && !g.plain && !g.need_reduce) {
      best.splitk = 1; best.ktiles_per_split = nk; best.streamk = force_streamk;
      best.ws_bytes = (size_t)2 * force_streamk * ((size_t)bm * bn + bn) * 4;
      best.sk_sliced = tune_int("A3D_FORCE_SK_SLICED", 0) &&
                       sk_sliced_ok(g.mode, (long)best.tiles_m * best.tiles_n, nk, force_streamk, nullptr);
    }
    return best;
  }
  // Cost model calibrated on MI355X with tools/sweep_igemm.py (profiles/r01_sweep_igemm.txt): a block progresses at
  // ~96.5 GMAC/s when two share a CU and ~1.6x that when alone; split-K costs one slab write+read at ~3 TB/s.
  // Model picks are within 1.15x (mostly 1.05x) of the best measured configuration for every MSDN layer/direction.
  for (int c = 0; c < kNumCfgs; ++c) {
    if (kCfgs[c].eff <= 0.f) continue;
    if (is_glds_cfg(c) && (g.mode != MODE_FWD || g.avec != 4 || g.bvec != 4 || g.plain || g.no_glds)) continue;
    if (c == kGen2Cfg && (!g.gen2_ok || env_flag_no_gen2())) continue;
    const int bm = kCfgs[c].bm, bn = kCfgs[c].bn;
    const int tm = (g.M + bm - 1) / bm, tn = (g.N + bn - 1) / bn;
    const long tiles = (long)tm * tn;
    // split-K candidates: every small factor, coarser steps above, and the two factors that fill the chip's block
    // slots exactly once or twice (tiles * splitk just below 512 / 1024: a kernel of 486 blocks beats one of 648)
    int wants[48];
    int nw = 0;
    for (int w = 1; w <= 24; ++w) wants[nw++] = w;
    for (int w = 32; w <= 256; w *= 2) { wants[nw++] = w; if (w < 256) wants[nw++] = w + w / 2; }
    wants[nw++] = (int)std::max<long>(1, kSlots / tiles);
    wants[nw++] = (int)std::max<long>(1, 2 * kSlots / tiles);
    for (int wi = 0; wi < nw; ++wi) {
      const int want = wants[wi];
      if (want > 1 && g.plain) continue;
      if (want == 1 && g.need_reduce && nk >= 2) continue;      // (nk == 1: launch_igemm refuses the call, nothing is enqueued)
      if (want > 1 && want > nk / 2 && !(g.need_reduce && want == 2)) continue;
      if (want > 1 && (size_t)want * g.M * g.N * 4 > kMaxSlabBytes) continue;
      const int kps = (nk + want - 1) / want;
      const int splitk = (nk + kps - 1) / kps;
      const long blocks = tiles * splitk;
      double t_b = (double)bm * bn * kps * 32.0 / (96.5e3 * kCfgs[c].eff);                // us
      if (is_glds_cfg(c) && kps < 40) t_b *= 1.08;     // LDS-DMA pays off on long K ranges only (conv2d_2: 158 vs 146 us)
      double f;      // kernel time in units of t_b: the slowest CU decides (tools/fit_planner.py)
      if (blocks <= 256) f = 0.62;
      else if (blocks <= kSlots) f = 1.0;
      else f = std::max((double)blocks / kSlots + 0.08, 1.45);
      double t = t_b * f;
      if (splitk > 1) t += 2.5 + (double)g.M * g.N * 4.0 * (splitk + 1) / 3.0e6;
      if (t < best_t) {
        best_t = t;
        best.cfg = c;
        best.splitk = splitk;
        best.ktiles_per_split = kps;
        best.tiles_m = tm;
        best.tiles_n = tn;
        best.ws_bytes = splitk > 1 ? (size_t)splitk * g.M * g.N * 4 : 0;
        best.streamk = 0;
        best.sk_sliced = 0;
      }
    }
    // stream-K: equal shares of the (tile, k-tile) iterations for one or two blocks per CU — no tile quantisation, and
    // at most two partial slabs per block (register-staged kernels only; the pooling forward never splits)
    if (streamk_cfg(c) && !g.plain && !g.need_reduce && !env_flag_no_streamk()) {
      const long iters = tiles * nk;
      for (int G = 256; G <= 512; G += 256) {
        if (force_streamk > 0 && G != 256) continue;
        const int grid = force_streamk > 0 ? force_streamk : G;
        if (iters < 4L * grid) continue;
        const long per = (iters + grid - 1) / grid;
        // blocks meeting in one tile: their slabs are added one after the other by the fixup, so few tiles with very
        // long K (bwd-filter of the 3-channel layers, dense layers) stay with classic split-K and its flat reduction
        long meet = (nk + per - 1) / per + 1;
        // bwd-filter: K (the pixel axis) in eight slices, one per XCD — each XCD streams its own eighth of x and dz
        long meet_sliced = 0;
        const bool sliced = !env_flag_no_sliced() && sk_sliced_ok(g.mode, tiles, nk, grid, &meet_sliced);
        if (force_streamk <= 0 && ((sliced ? meet_sliced > 48 : meet > 16) || tiles < 24)) continue;
        if (sliced) meet = meet_sliced / 4;
        double t = (double)bm * bn * per * 32.0 / (96.5e3 * kCfgs[c].eff) * (grid <= 256 ? 0.62 : 1.0);
        // ~1.5 slabs per block are written and read back, then the split tiles are written once more
        const double slabs = std::min<double>(1.5 * grid, 2.0 * tiles) * bm * bn * 4.0;
        // (round 3 refit: 4.5 TB/s for the slab traffic — the fixup reads eight slabs deep; at 3 TB/s the model preferred
        // split-K 2 for conv2d_1's bwd-data, measured 349 us against 334 for stream-K: profiles/r03_sweep_hot.txt)
        t += 3.0 + 0.4 * meet + (2.0 * slabs + (double)std::min<long>(tiles, grid) * bm * bn * 4.0) / 4.5e6;
        if (force_streamk > 0) t = 0;
        if (t < best_t) {
          best_t = t;
          best.cfg = c;
          best.splitk = 1;
          best.ktiles_per_split = nk;
          best.tiles_m = tm;
          best.tiles_n = tn;
          best.streamk = grid;
          best.sk_sliced = sliced ? 1 : 0;
          best.ws_bytes = (size_t)2 * grid * ((size_t)bm * bn + bn) * 4;
        }
      }
    }
  }
  return best;
}

// Sums the split-K slabs in slab order (deterministic) and applies the epilogue.  Slab reads are issued four at a
// time into independent registers: a thread's serial chain of `splitk` dependent loads was the cost of this kernel.
template <typename V>
__device__ __forceinline__ V slab_sum(const V* ws, size_t slab, int splitk, size_t i) {
  V s = V(0.f);
  int z = 0;
  for (; z + 8 <= splitk; z += 8) {       // eight slabs in flight, added in slab order
    V t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = ws[(size_t)(z + u) * slab + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  if (z + 4 <= splitk) {
    const V a = ws[(size_t)z * slab + i], b = ws[(size_t)(z + 1) * slab + i];
    const V c = ws[(size_t)(z + 2) * slab + i], d = ws[(size_t)(z + 3) * slab + i];
    s += a; s += b; s += c; s += d;
    z += 4;
  }
  for (; z < splitk; ++z) s += ws[(size_t)z * slab + i];
  return s;
}

// C position (in units of V) of slab position i when the slabs' rows are padded window runs (ReduceParams::row_rlp), or
// SIZE_MAX for a pad row; nv = V's per row
__device__ __forceinline__ size_t unpadded_pos(size_t i, int nv, int rl, int rlp) {
  if (rlp == 0) return i;
  const size_t row = i / (size_t)nv, c = i - row * (size_t)nv;
  const size_t rr = row / (size_t)rlp, q = row - rr * (size_t)rlp;
  return q < (size_t)rl ? (rr * (size_t)rl + q) * (size_t)nv + c : SIZE_MAX;
}
template <typename V>
__device__ __forceinline__ void reduce_wide(const V* src, V* dst, size_t slab, size_t count, int splitk, size_t first, V* lds, int nv,
                                            int rl, int rlp);
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ReduceParams p) {
  const size_t total = (size_t)p.M * p.N;
  unsigned nblk = gridDim.x;
  if (p.dbias_splits > 0) {      // up to 256 rows of column-sum partials: blocks of their own, sixteen threads per column
    __shared__ float lds[256];
    if (blockIdx.x >= p.dbias_block0) {
      reduce_wide<float>(p.dbias_ws, p.dbias_out, (size_t)p.N, (size_t)p.N, p.dbias_splits, (size_t)(blockIdx.x - p.dbias_block0) * 16, lds,
                         0, 0, 0);
      return;
    }
    nblk = p.dbias_block0;
  } else if (p.dbias_out) {       // N sums of `splitk` values: the grid's first threads do them on the side
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)p.N; i += (size_t)gridDim.x * 256)
      p.dbias_out[i] = slab_sum<float>(p.dbias_ws, (size_t)p.N, p.splitk, i);
  }
  if (p.vec4) {   // plain sum of 16-byte columns: bwd-filter slabs (no epilogue, contiguous output)
    typedef float f4 __attribute__((ext_vector_type(4)));
    const size_t nv = total / 4;
    const f4* ws4 = reinterpret_cast<const f4*>(p.ws);
    const size_t slab4 = p.slab / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (size_t)nblk * 256) {
      const size_t o = unpadded_pos(i, p.N / 4, p.row_rl, p.row_rlp);
      if (o != SIZE_MAX) reinterpret_cast<f4*>(p.C)[o] = slab_sum<f4>(ws4, slab4, p.splitk, i);
    }
    return;
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)nblk * 256) {
    float s = slab_sum<float>(p.ws, p.slab, p.splitk, i);
    const int row = (int)(i / p.N), col = (int)(i - (size_t)row * p.N);
    const size_t o = remap_row(row, p.mode == MODE_BWD_D ? p.sub_step : 1, p.sub_ph, p.sub_pw, p.outW, p.outHW, p.div_phw,
                               p.div_pw) * p.ldc + col;
    if (p.mode == MODE_FWD) {
      if (p.bias) s += p.bias[col];
      if (p.act == EPI_RELU) s = fmaxf(s, 0.f);
      else if (p.act == EPI_SIGMOID) s = 1.f / (1.f + expf(-s));
      if (p.keep) s = p.keep[i] ? s * p.mask_scale : 0.f;
    } else if (p.mode == MODE_BWD_D) {
      if (p.mask) {
        const float y = p.c16 ? (float)reinterpret_cast<const __bf16*>(p.mask)[o] : p.mask[o];
        s = apply_act_grad(s, y, p.mask_act, p.mask_scale);
      }
    }
    if (p.c_cols == 0 || col < p.c_cols) {
      if (p.c16) reinterpret_cast<__bf16*>(p.C)[o] = (__bf16)s;
      else p.C[o] = s;
    }
    if (p.C2 && col < p.cols2) {
      const size_t o2 = ((size_t)row * p.ld2 + col) * p.step2 + p.off2;
      if (p.c2_16) static_cast<__bf16*>(p.C2)[o2] = (__bf16)(p.c16 ? (float)(__bf16)s : s);
      else static_cast<float*>(p.C2)[o2] = p.c16 ? (float)(__bf16)s : s;
    }
  }
}

// the second output of a launch that had no reduction stage to write it: a copy of the finished tensor
__global__ __launch_bounds__(256) void second_output_kernel(const void* C, int c16, int M, int ldc, void* C2, int ld2, int step2, int off2,
                                                            int c2_16, int cols2) {
  const size_t total = (size_t)M * cols2;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int row = (int)(i / cols2), col = (int)(i - (size_t)row * cols2);
    const size_t o = (size_t)row * ldc + col;
    const float s = c16 ? (float)static_cast<const __bf16*>(C)[o] : static_cast<const float*>(C)[o];
    const size_t o2 = ((size_t)row * ld2 + col) * step2 + off2;
    if (c2_16) static_cast<__bf16*>(C2)[o2] = (__bf16)s;
    else static_cast<float*>(C2)[o2] = s;
  }
}

// Many slabs, few outputs (conv2d_0's bwd-filter: 128 slabs of 36 k floats, and its 96 bias sums): one thread per output
// means 35 blocks on 256 CUs, each walking 128 dependent-latency rounds — 28 us for 18 MB.  Here sixteen threads share
// an output: thread `part` adds its contiguous sixteenth of the slabs in slab order, the sixteen partial sums meet in
// LDS and are added in part order.  A fixed order, so still the same bits on every run (a different order than the
// one-thread sum above: a launch uses one or the other by shape alone, never by timing).
template <typename V>
__device__ __forceinline__ void reduce_wide(const V* src, V* dst, size_t slab, size_t count, int splitk, size_t first,
                                            V* lds, int nv, int rl, int rlp) {
  const int tid = threadIdx.x, out = tid & 15, part = tid >> 4;
  const size_t i = first + out;
  const int per = (splitk + 15) / 16, z0 = part * per, z1 = min(splitk, z0 + per);
  V s = V(0.f);
  if (i < count) {
    int z = z0;
    for (; z + 4 <= z1; z += 4) {
      const V a = src[(size_t)z * slab + i], b = src[(size_t)(z + 1) * slab + i];
      const V c = src[(size_t)(z + 2) * slab + i], d = src[(size_t)(z + 3) * slab + i];
      s += a; s += b; s += c; s += d;
    }
    for (; z < z1; ++z) s += src[(size_t)z * slab + i];
  }
  lds[part * 16 + out] = s;
  __syncthreads();
  if (part == 0 && i < count) {
    V t = lds[out];
#pragma unroll
    for (int q = 1; q < 16; ++q) t += lds[q * 16 + out];
    const size_t o = unpadded_pos(i, nv, rl, rlp);
    if (o != SIZE_MAX) dst[o] = t;
  }
}
__global__ __launch_bounds__(256) void splitk_reduce_wide_kernel(const ReduceParams p, unsigned blocks_c) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  __shared__ f4 lds[256];
  if (blockIdx.x < blocks_c)
    reduce_wide<f4>(reinterpret_cast<const f4*>(p.ws), reinterpret_cast<f4*>(p.C), p.slab / 4, (size_t)p.M * p.N / 4,
                    p.splitk, (size_t)blockIdx.x * 16, lds, p.N / 4, p.row_rl, p.row_rlp);
  else
    reduce_wide<float>(p.dbias_ws, p.dbias_out, (size_t)p.N, (size_t)p.N, p.dbias_splits > 0 ? p.dbias_splits : p.splitk, (size_t)(blockIdx.x - blocks_c) * 16,
                       reinterpret_cast<float*>(lds), 0, 0, 0);
}

int launch_splitk_reduce(const ReduceParams& r, hipStream_t st) {
  const size_t total = (size_t)r.M * r.N;
  clear_stale_error();
  static const bool no_wide = tune_int("A3D_NO_WIDE_REDUCE", 0) != 0;      // A/B aid
  if (!no_wide && r.vec4 && r.splitk >= 16 && total / 4 <= (size_t)1 << 16) {
    const unsigned blocks_c = (unsigned)((total / 4 + 15) / 16), blocks_b = r.dbias_out ? (unsigned)((r.N + 15) / 16) : 0u;
    hipLaunchKernelGGL(splitk_reduce_wide_kernel, dim3(blocks_c + blocks_b), dim3(256), 0, st, r, blocks_c);
    return check_launch("splitk_reduce_wide");
  }
  const unsigned g = (unsigned)std::min<size_t>(((r.vec4 ? total / 4 : total) + 255) / 256, 2048);
  ReduceParams rr = r;
  rr.dbias_block0 = g;
  const unsigned gb = (r.dbias_out && r.dbias_splits > 0) ? (unsigned)((r.N + 15) / 16) : 0u;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(g + gb), dim3(256), 0, st, rr);
  return check_launch("splitk_reduce");
}

int launch_igemm(int mode, const GemmPlan& plan, int avec, int bvec, IgemmParams& p, void* ws, hipStream_t st) {
  if (mode == MODE_BWD_D && p.stride != 1)      // strided bwd-data always arrives as stride-1 parity classes
    return set_error(A3D_EINVAL, "igemm: bwd-data launches are stride-1 problems");
  p.splitk = plan.splitk;
  p.ktiles_per_split = plan.ktiles_per_split;
  p.tiles_m = plan.tiles_m;
  p.tiles_n = plan.tiles_n;
  p.slab = (size_t)p.M * p.N;
  float* final_c = p.C;
  float* final_dbias = p.dbias;
  if (plan.splitk > 1) {
    if (!ws) return set_error(A3D_EWORKSPACE, "igemm: split-K needs a workspace");
    p.C = static_cast<float*>(ws);
    if (p.dbias) p.dbias = static_cast<float*>(ws) + (size_t)plan.splitk * p.slab;   // [splitk][N] after the C slabs
  }
  unsigned grid = (unsigned)((long)plan.tiles_m * plan.tiles_n * plan.splitk);
  if (plan.streamk > 0) {
    if (!ws) return set_error(A3D_EWORKSPACE, "igemm: stream-K needs a workspace");
    const size_t tile_elems = (size_t)kCfgs[plan.cfg].bm * kCfgs[plan.cfg].bn;
    p.streamk = plan.sk_sliced ? 2 : 1;
    p.sk_ws = static_cast<float*>(ws);
    p.sk_bias = p.sk_ws + (size_t)2 * plan.streamk * tile_elems;
    p.div_nk = make_fastdiv((uint32_t)std::max(1, (p.K + 31) / 32));
    grid = (unsigned)plan.streamk;
  }
  int rc;
#ifdef A3D_STAMPS
  if (!g_stamps) (void)hipMalloc(&g_stamps, kStampBytes);
  (void)hipMemsetAsync(g_stamps, 0, kStampBytes, st);
  p.stamps = grid * 8ull * 16 * 8 <= kStampBytes ? g_stamps : nullptr;
  g_stamp_grid = grid;
#endif
  p.dbg = tune_int("A3D_DBG", 0);
  static const bool plan_log = tune_int("A3D_PLAN_LOG", 0) != 0;       // tuning aid: one line per launch on stderr
  if (plan_log)
    fprintf(stderr, "a3d plan: mode %d M %d N %d K %d -> %s %d (%dx%d) splitk %d streamk %d%s grid %u\n", mode, p.M, p.N, p.K,
            plan.ring ? "ring" : "cfg", plan.ring ? plan.ring - 1 : plan.cfg, plan.ring ? kRingCfgs[plan.ring - 1].bm : kCfgs[plan.cfg].bm,
            plan.ring ? kRingCfgs[plan.ring - 1].bn : kCfgs[plan.cfg].bn, plan.splitk, plan.streamk, plan.sk_sliced ? " (K-sliced)" : "", grid);
  TimingSlot slot{};
  {
    a3d_timing_record& r = slot.rec;
    r.mode = mode; r.prec = plan.prec;
    if (plan.ring) {
      r.bm = kRingCfgs[plan.ring - 1].bm; r.bn = kRingCfgs[plan.ring - 1].bn; r.waves_m = 0; r.nwaves = 8; r.bk = 64;
      r.lds_dma = 3;                                        // 3: igemm_ring_kernel
    } else if (plan.prec != A3D_PREC_F32) {
      r.bm = 128; r.bn = plan.bf16_bn; r.waves_m = 4; r.nwaves = 8; r.bk = plan.prec == A3D_PREC_BF16X3 ? 32 : 64;
    } else {
      r.bm = kCfgs[plan.cfg].bm; r.bn = kCfgs[plan.cfg].bn; r.waves_m = kCfgWavesM[plan.cfg];
      r.nwaves = kCfgNWaves[plan.cfg]; r.bk = kCfgs[plan.cfg].bk;
      r.lds_dma = plan.cfg == kGen2Cfg ? 5 : (plan.cfg >= kFirstGldsCfg && avec == 4 && bvec == 4);
    }
    r.avec = avec; r.bvec = bvec; r.splitk = plan.splitk; r.m = p.M; r.n = p.N; r.k = p.K; r.ms = 0.f;
    r.flops = 2.0 * p.M * p.N * p.K;
  }
  // a second output / a narrower output are written by the reduction stage of a classic split-K forward or bwd-data launch
  // (or, the second output, by a copy launch behind an unsplit one): refuse the other combinations BEFORE anything is
  // enqueued — a GEMM that has already stored N columns into rows of c_cols floats cannot be taken back
  const bool reduce_vec4 = plan.splitk > 1 && mode == MODE_BWD_F && p.ldc == p.N && (p.slab % 4) == 0 && aligned16(final_c) && aligned16(ws);
  if (plan.streamk > 0) {
    A3D_CHECK_ARG(!p.out2 && !p.c_cols, "second output: not on stream-K launches");
  } else if (plan.splitk > 1) {
    if (reduce_vec4 || mode == MODE_BWD_F) A3D_CHECK_ARG(!p.out2 && !p.c_cols, "a second output belongs to a forward or bwd-data launch");
  } else {
    A3D_CHECK_ARG(!p.c_cols || p.c_cols == p.N, "this launch has no reduction stage: the output cannot be narrower than the GEMM");
    A3D_CHECK_ARG(!p.out2 || (p.sub_step == 1 && !p.pool), "second output: plain forward / bwd-data launches only");
  }
  const bool timed = timing_wanted(slot.rec);
  if (timed && (rc = timing_begin(slot, st)) != A3D_OK) return rc;
  if (plan.ring) rc = launch_igemm_ring(mode, plan.ring - 1, p, grid, st);
  else if (plan.prec != A3D_PREC_F32) rc = launch_igemm_bf16(mode, plan.bf16_bn, plan.prec == A3D_PREC_BF16X3, p, grid, st);
  else if (mode == MODE_FWD) rc = launch_igemm_mode0(plan.cfg, avec, bvec, p, grid, st);
  else if (mode == MODE_BWD_D) rc = launch_igemm_mode1(plan.cfg, avec, bvec, p, grid, st);
  else rc = launch_igemm_mode2(plan.cfg, avec, bvec, p, grid, st);
  if (timed) timing_end(slot, st);
  if (rc != A3D_OK) return rc;
  if (plan.streamk > 0) {
    const unsigned tiles = (unsigned)(plan.tiles_m * plan.tiles_n);
    if (mode == MODE_FWD) return launch_fixup_mode0(plan.cfg, p, tiles, grid, st);
    if (mode == MODE_BWD_D) return launch_fixup_mode1(plan.cfg, p, tiles, grid, st);
    return launch_fixup_mode2(plan.cfg, p, tiles, grid, st);
  }
  if (plan.splitk > 1) {
    ReduceParams r{};
    r.ws = static_cast<const float*>(ws); r.C = final_c; r.bias = p.bias; r.mask = p.mask; r.keep = p.keep;
    r.mask_scale = p.mask_scale; r.M = p.M; r.N = p.N; r.ldc = p.ldc; r.splitk = plan.splitk; r.act = p.act;
    r.mode = mode; r.slab = p.slab; r.mask_act = p.mask_act; r.c16 = p.c16;
    r.vec4 = reduce_vec4;
    r.sub_step = p.sub_step; r.sub_ph = p.sub_ph; r.sub_pw = p.sub_pw; r.outW = p.outW; r.outHW = p.outHW;
    r.div_phw = p.div_phw; r.div_pw = p.div_pw;
    r.dbias_ws = final_dbias ? p.dbias : nullptr;
    r.dbias_out = final_dbias;
    if (p.dbias_parts) { r.dbias_ws = p.dbias_parts; r.dbias_out = p.dbias_parts_out; r.dbias_splits = p.dbias_parts_n; }
    if (r.vec4 && p.unpad_dst && p.unpad_rlp > 0 && p.N % 4 == 0 && aligned16(p.unpad_dst)) {
      r.C = p.unpad_dst; r.row_rl = p.unpad_rl; r.row_rlp = p.unpad_rlp;
      p.unpad_done = 1;
    }
    r.C2 = p.out2; r.ld2 = p.out2_ld; r.step2 = p.out2_step; r.off2 = p.out2_off; r.c2_16 = p.out2_bf16; r.cols2 = p.out2_cols;
    r.c_cols = p.c_cols;
    rc = launch_splitk_reduce(r, st);
    return rc;
  }
  if (p.out2) {                                       // no reduction stage wrote it: one copy launch (what the caller saved otherwise)
    const size_t total = (size_t)p.M * p.out2_cols;
    clear_stale_error();
    hipLaunchKernelGGL(second_output_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 2048)), dim3(256), 0, st,
                       static_cast<const void*>(final_c), p.c16, p.M, p.ldc, p.out2, p.out2_ld, p.out2_step, p.out2_off, p.out2_bf16,
                       p.out2_cols);
    rc = check_launch("second_output");
  }
  return rc;
}

// a3d_second_output -> the host-only fields of IgemmParams
static int take_second_output(IgemmParams& p, const a3d_second_output* o) {
  if (!o || !o->ptr) return A3D_OK;
  A3D_CHECK_ARG(o->ld > 0 && o->step > 0 && o->offset >= 0 && o->cols > 0 && o->cols <= p.N, "second output: bad geometry");
  p.out2 = o->ptr; p.out2_ld = o->ld; p.out2_step = o->step; p.out2_off = o->offset; p.out2_bf16 = o->bf16 ? 1 : 0; p.out2_cols = o->cols;
  return A3D_OK;
}

// ---- "window-run" form of few-channel VALID convolutions (Cin = 3: conv2d_0, fine/first) ----
// For an unpadded conv with densely packed pixels a filter row (s, c) is ONE contiguous run of S*Cin floats of the
// input, starting stride*Cin floats after the previous output pixel's run.  When those starts are 16- (or 8-) byte
// aligned the run is gathered with vector loads instead of float by float: the K axis becomes (r, q) with q padded
// to a multiple of the vector width, the padded filter rows are zero, and the generic kernel sees a conv with S' = 1
// and Cin' = padded run length.  The few floats read past a run belong to the next pixels of the same image row
// (guaranteed in-bounds by run_form_ok) and meet zero weights.
struct RunForm {
  int vec;      // 4 or 2
  int rl, rlp;  // run length S*Cin and its padded length
  int kp;       // R * rlp
};
// bwd_filter: the filter-gradient GEMM may also take runs that start at any 4-byte address (DCNF's first conv: stride 1,
// runs 12 bytes apart) as 8-byte buffer loads — the hardware only asks a multi-dword buffer load for dword alignment —
// and runs whose padding reaches past the row's end: what a pad element multiplies ends up in a pad row of the padded
// gradient, which is never stored (past the tensor's end the descriptor returns zeros).
static bool run_form_ok(const a3d_conv_desc* d, const float* x, RunForm* rf, bool bwd_filter = false) {
  if (d->pad_t || d->pad_l || d->ldx != d->c || d->c % 4 == 0 || d->c > 4) return false;
  if (tune_int("A3D_NO_RUNFORM", 0)) return false;
  const int step = d->stride * d->c, row = d->w * d->c;
  int vec = 0;
  bool anywhere = false;
  if (step % 4 == 0 && row % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) vec = 4;
  else if (step % 2 == 0 && row % 2 == 0 && (reinterpret_cast<uintptr_t>(x) & 7) == 0) vec = 2;
  else if (bwd_filter && d->precision == A3D_PREC_F32 && !d->storage && (reinterpret_cast<uintptr_t>(x) & 3) == 0 &&
           !tune_int("A3D_NO_RUNFORM_ANYWHERE", 0)) {
    vec = 2;
    anywhere = true;
  }
  if (vec == 4 && d->precision == A3D_PREC_F32) {      // (the bf16 kernels take 16-byte operands only)
    // 8-byte runs pad less (conv2d_0: 33 -> 34 instead of 36 floats per filter row).  Worth the narrower loads when it
    // saves a whole 128-row tile of the bwd-filter GEMM (374 vs 396 rows: 3 tiles instead of 4; 172 -> 148 us, and the
    // forward loses a k-tile: 135 -> 129 us)
    const int rl = d->s * d->c;
    const int kp4 = d->r * ((rl + 3) / 4 * 4), kp2 = d->r * ((rl + 1) / 2 * 2);
    if ((kp2 + 127) / 128 < (kp4 + 127) / 128) vec = 2;
  }
  if (!vec) return false;
  rf->vec = vec;
  rf->rl = d->s * d->c;
  rf->rlp = (rf->rl + vec - 1) / vec * vec;
  rf->kp = d->r * rf->rlp;
  // the padded run of the last output column stays inside its row; `anywhere`: the run itself does (no implicit padding
  // on the right / below: a SAME convolution with an even kernel has pad_t = pad_l = 0 and still pads), its padding may not
  if (anywhere) return (d->wo - 1) * step + rf->rl <= row && (d->ho - 1) * d->stride + d->r <= d->h;
  return (d->wo - 1) * step + rf->rlp <= row;
}

// filter [R*RL][N] <-> padded [R*RLP][N] (pad rows zero)
// (forward: the copy's rows are also padded to `np` >= n columns, zeros: fine/first's 63 filters become 16-byte rows)
__global__ __launch_bounds__(256) void pad_filter_kernel(const float* w, float* wp, int r, int rl, int rlp, int n, int np) {
  const int total = r * rlp * np;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int col = i % np, row = i / np;
    const int q = row % rlp, rr = row / rlp;
    wp[i] = (q < rl && col < n) ? w[(size_t)(rr * rl + q) * n + col] : 0.f;
  }
}
__global__ __launch_bounds__(256) void unpad_filter_kernel(const float* wp, float* w, int r, int rl, int rlp, int n) {
  const int total = r * rl * n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int col = i % n, row = i / n;
    const int q = row % rl, rr = row / rl;
    w[i] = wp[(size_t)(rr * rlp + q) * n + col];
  }
}
static size_t run_filter_bytes(const a3d_conv_desc* d, const RunForm& rf) {
  return ((size_t)rf.kp * ((d->k + 3) / 4 * 4) * 4 + 255) / 256 * 256;      // forward: rows padded to 16 bytes
}

// ---- descriptor checks / parameter assembly ----
static int ilog2_exact(int v) {
  for (int l = 0; l < 8; ++l)
    if ((1 << l) == v) return l;
  return -1;
}

static int check_desc(const a3d_conv_desc* d) {
  A3D_CHECK_ARG(d != nullptr, "conv: null descriptor");
  A3D_CHECK_ARG(d->n > 0 && d->h > 0 && d->w > 0 && d->c > 0 && d->k > 0 && d->r > 0 && d->s > 0, "conv: bad dims");
  A3D_CHECK_ARG(ilog2_exact(d->stride) >= 0 && d->stride <= 4, "conv: stride %d unsupported (1, 2 or 4)", d->stride);
  A3D_CHECK_ARG(d->pad_t >= 0 && d->pad_l >= 0 && d->pad_t < d->r && d->pad_l < d->s, "conv: bad padding");
  A3D_CHECK_ARG(d->ho > 0 && d->wo > 0, "conv: bad output size");
  A3D_CHECK_ARG((d->ho - 1) * d->stride - d->pad_t < d->h && (d->wo - 1) * d->stride - d->pad_l < d->w,
                "conv: output size inconsistent with input");
  A3D_CHECK_ARG(d->ldx >= d->c && d->ldy >= d->k, "conv: pixel strides smaller than channel counts");
  A3D_CHECK_ARG(d->precision >= A3D_PREC_F32 && d->precision <= A3D_PREC_BF16, "conv: unknown precision %d", d->precision);
  const double lim = 2147483647.0;
  A3D_CHECK_ARG((double)d->n * d->h * d->w * d->ldx < lim && (double)d->n * d->ho * d->wo * d->ldy < lim &&
                    (double)d->r * d->s * d->c * d->k < lim,
                "conv: tensor too large for 31-bit indexing");
  return A3D_OK;
}


struct ConvProblem {
  GemmProblem g;
};

// May this layer / direction run on the second-generation kernel (igemm2.h)?  By the descriptor alone: the callers clear the
// flag again when an operand is not 16-byte aligned or the launch takes another form (window runs, dense streams).
static bool gen2_desc_ok(const a3d_conv_desc* d, int mode) {
  if (d->precision != A3D_PREC_F32 || d->storage) return false;
  if (mode == MODE_FWD) return d->c % 32 == 0 && d->ldx % 4 == 0 && d->k % 4 == 0;
  if (mode == MODE_BWD_D) return d->k % 32 == 0 && d->ldy % 4 == 0;
  return d->c % 4 == 0 && d->ldx % 4 == 0 && d->k % 4 == 0 && d->ldy % 4 == 0;
}

static GemmProblem fwd_problem(const a3d_conv_desc* d) {
  GemmProblem g;
  g.mode = MODE_FWD;
  g.M = d->n * d->ho * d->wo; g.N = d->k; g.K = d->r * d->s * d->c;
  g.avec = (d->c % 4 == 0 && d->ldx % 4 == 0) ? 4 : 1;
  g.bvec = (d->k % 4 == 0) ? 4 : 1;
  g.gen2_ok = gen2_desc_ok(d, MODE_FWD);
  return g;
}
// workspace of a problem whichever of the two kernel generations a launch ends up on
static size_t plan_ws_either(GemmProblem g, int precision) {
  size_t need = plan_gemm(g, precision).ws_bytes;
  if (g.gen2_ok) { g.gen2_ok = 0; need = std::max(need, plan_gemm(g, precision).ws_bytes); }
  return need;
}
static GemmProblem bwd_d_problem(const a3d_conv_desc* d) {
  GemmProblem g;
  g.mode = MODE_BWD_D;
  g.M = d->n * d->h * d->w; g.N = d->c; g.K = d->r * d->s * d->k;
  g.avec = (d->k % 4 == 0 && d->ldy % 4 == 0) ? 4 : 1;
  g.bvec = (d->k % 4 == 0) ? 4 : 1;
  g.gen2_ok = gen2_desc_ok(d, MODE_BWD_D);
  return g;
}
static GemmProblem bwd_f_problem(const a3d_conv_desc* d) {
  GemmProblem g;
  g.mode = MODE_BWD_F;
  g.M = d->r * d->s * d->c; g.N = d->k; g.K = d->n * d->ho * d->wo;
  g.avec = (d->c % 4 == 0 && d->ldx % 4 == 0) ? 4 : 1;
  g.bvec = (d->k % 4 == 0 && d->ldy % 4 == 0) ? 4 : 1;
  g.gen2_ok = gen2_desc_ok(d, MODE_BWD_F);
  return g;
}

// a3d_conv_desc.storage -> which GEMM operands are bf16, with the checks a 16-byte bf16 gather needs
static int apply_storage(IgemmParams& p, GemmProblem& g, int precision, bool a16, bool b16, bool c16, int a_c, int a_ld,
                         int b_c, int b_ld, const void* a, const void* b, const void* c, int c_c, int c_ld) {
  if (!a16 && !b16 && !c16) return A3D_OK;
  // a bf16 OUTPUT alone is also served by the fp32 kernels (the 3-channel layers of config 5 keep fp32 arithmetic)
  A3D_CHECK_ARG(precision == A3D_PREC_BF16 || (precision == A3D_PREC_F32 && !a16 && !b16),
                "bf16 operands need precision A3D_PREC_BF16");
  A3D_CHECK_ARG(!a16 || (a_c % 8 == 0 && a_ld % 8 == 0 && aligned16(a)), "bf16 operand: channels / stride must be multiples of 8, base 16-byte aligned");
  // b_c < 0: a row-major [rows][b_ld] operand read in whole 16-byte chunks (its pad columns must be readable zeros)
  A3D_CHECK_ARG(!b16 || ((b_c < 0 || b_c % 8 == 0) && b_ld % 8 == 0 && aligned16(b)), "bf16 operand: channels / stride must be multiples of 8, base 16-byte aligned");
  A3D_CHECK_ARG(!c16 || aligned16(c), "bf16 output: base must be 16-byte aligned");
  (void)c_c; (void)c_ld;
  p.a16 = a16; p.b16 = b16; p.c16 = c16;
  if (c16) g.no_glds = 1;
  if (a16) g.avec = 4;
  if (b16) g.bvec = 4;
  return A3D_OK;
}

static int take_second_output(IgemmParams& p, const a3d_second_output* o);
static void fill_common(IgemmParams& p, const GemmProblem& g) {
  p = IgemmParams{};
  p.M = g.M; p.N = g.N; p.K = g.K;
  p.mask_scale = 1.f;
  p.mask_act = EPI_RELU;
  p.sub_step = 1;
}

// Staging parameters of igemm_body (call after S, Cg, H, W, stride and the paddings are set): operand extents for the
// buffer descriptors, the wave-uniform tap decode for layers whose gathered channel count is a multiple of BK = 32, and
// whether the gather can leave the image at all.
static void fill_staging(IgemmParams& p, int mode, unsigned long long a_elems, unsigned long long b_elems, int taps_r,
                         int taps_s, int filt_r, int filt_s) {
  p.a_elems = a_elems;
  p.b_elems = b_elems;
  p.ntaps = std::max(1, taps_r * taps_s);
  p.cpt = std::max(1, p.Cg / 32);
  p.uni = (mode != MODE_BWD_F) && p.Cg % 32 == 0 && p.K == p.ntaps * p.Cg && !env_flag_no_uni();
  p.kperm = p.uni && p.ntaps > 1 && p.cpt > 1 && !env_flag_no_kperm();
  p.div_cpt = make_fastdiv(p.cpt);
  p.div_taps = make_fastdiv(p.ntaps);
  p.cpt64 = std::max(1, p.Cg / 64);
  p.uni64 = p.uni && p.Cg % 64 == 0;
  p.kperm64 = p.uni64 && p.ntaps > 1 && p.cpt64 > 1 && !env_flag_no_kperm();
  p.div_cpt64 = make_fastdiv(p.cpt64);
  (void)filt_r; (void)filt_s;
}

}  // namespace a3d

using namespace a3d;

extern "C" {

#ifdef A3D_STAMPS
// diagnostic build only: copies the last launch's [grid][8 waves][8] phase sums to the host; returns the grid size
int a3d_debug_stamps(unsigned long long* out, size_t cap_bytes) {
  (void)hipDeviceSynchronize();
  if (g_stamps && out) (void)hipMemcpy(out, g_stamps, std::min(cap_bytes, kStampBytes), hipMemcpyDeviceToHost);
  return (int)g_stamp_grid;
}
#endif

int a3d_timing_enable(int on) {
  std::lock_guard<std::mutex> lk(g_timing_mu);
  g_timing_on = on != 0;
  return A3D_OK;
}

int a3d_timing_select(const a3d_timing_record* like) {
  std::lock_guard<std::mutex> lk(g_timing_mu);
  g_timing_only = like != nullptr;
  if (like) g_timing_like = *like;
  return A3D_OK;
}

int a3d_timing_collect(a3d_timing_record* out, int cap) {
  std::vector<TimingSlot> slots;
  {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    slots.swap(g_timing);
  }
  int n = 0;
  for (TimingSlot& s : slots) {
    (void)hipEventSynchronize(s.stop);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, s.start, s.stop);
    s.rec.ms = ms;
    if (out && n < cap) out[n++] = s.rec;
    (void)hipEventDestroy(s.start);
    (void)hipEventDestroy(s.stop);
  }
  return n;
}

static size_t bf16_image_filter_bytes(const a3d_conv_desc* d);

size_t a3d_conv2d_fwd_ws_bytes(const a3d_conv_desc* d) {
  if (check_desc(d) != A3D_OK) return 0;
  if (stencil1_applicable(d)) return 0;
  // (a maximum over the paths a launch may take: which one it is also depends on the operands' alignment)
  size_t need = plan_ws_either(fwd_problem(d), d->precision);
  {                                              // ... the LDS-DMA kernel's plan among them (its split-K slabs: dense layers)
    const int both = A3D_STORE_X_BF16 | A3D_STORE_W_BF16;
    GemmProblem g = fwd_problem(d);
    g.ring_ok = d->precision == A3D_PREC_BF16 && (d->storage & both) == both && d->c % 8 == 0 && d->k % 8 == 0;
    g.avec = g.bvec = 4;
    if (g.ring_ok) need = std::max(need, plan_gemm(g, d->precision).ws_bytes);
  }
  if (conv3_applicable(d, nullptr)) need = std::max(need, conv3_ws_bytes(d));
  RunForm rf;
  if (run_form_ok(d, nullptr, &rf)) {
    GemmProblem g = fwd_problem(d);
    g.K = rf.kp; g.avec = rf.vec;
    need = std::max(need, run_filter_bytes(d, rf) + plan_gemm(g, d->precision).ws_bytes);
  }
  if (d->storage & A3D_STORE_X_BF16 && d->c == 4 && d->ldx == 4) {      // bf16 image form (same test minus the pointer)
    GemmProblem g = fwd_problem(d);
    g.K = d->r * ((d->s * 2 + 3) / 4 * 4 * 2); g.avec = 4; g.bvec = 4; g.no_glds = 1;
    need = std::max(need, bf16_image_filter_bytes(d) + plan_gemm(g, d->precision).ws_bytes);
    need = std::max(need, conv3b_filter_bytes(d));
  }
  return need;
}

// ---- window-run form over a bf16 image of 8-byte pixels (4 channels; a3d_pad_channels_bf16) ----
// Seen through its float view a pixel is 2 floats, so at an even stride the runs of a filter row start 16 bytes apart and
// the bf16 kernel gathers them with its 16-byte loads: K = (r, run of s*4 bf16 padded to a multiple of 8), the filter
// [r][s][4][k] float32 is repacked per call as bf16 [r][run][k padded to 8] (pad rows / columns zero).
static bool bf16_image_form_ok(const a3d_conv_desc* d, const void* x) {
  if (!(d->storage & A3D_STORE_X_BF16) || (d->storage & A3D_STORE_W_BF16) || d->precision != A3D_PREC_BF16) return false;
  if (d->c != 4 || d->ldx != 4 || d->pad_t || d->pad_l || d->stride % 2 || (d->w * 2) % 4 || !aligned16(x)) return false;
  const int rlv = (d->s * 2 + 3) / 4 * 4;                       // padded run in view floats
  return (d->wo - 1) * d->stride * 2 + rlv <= d->w * 2;        // the last column's padded run stays inside its row
}
static size_t bf16_image_filter_bytes(const a3d_conv_desc* d) {
  const int rlp = (d->s * 2 + 3) / 4 * 4 * 2, np = (d->k + 7) / 8 * 8;      // bf16 elements
  return ((size_t)d->r * rlp * np * 2 + 255) / 256 * 256;
}
// filter [r][rl][n] float32 -> bf16 [r][rlp][np], pad rows / columns zero
__global__ __launch_bounds__(256) void pad_filter_bf16_kernel(const float* w, __bf16* wp, int r, int rl, int rlp, int n, int np) {
  const int total = r * rlp * np;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int col = i % np, row = i / np;
    const int q = row % rlp, rr = row / rlp;
    wp[i] = (__bf16)((q < rl && col < n) ? w[(size_t)(rr * rl + q) * n + col] : 0.f);
  }
}
static int pad_filter_bf16(const a3d_conv_desc* d, const float* w, __bf16* wp, hipStream_t st) {
  const int rl = d->s * 4, rlp = (d->s * 2 + 3) / 4 * 4 * 2, np = (d->k + 7) / 8 * 8;      // bf16 elements
  clear_stale_error();
  hipLaunchKernelGGL(pad_filter_bf16_kernel, dim3((d->r * rlp * np + 255) / 256), dim3(256), 0, st, w, wp, d->r, rl, rlp,
                     d->k, np);
  return check_launch("pad_filter_bf16");
}
static int conv_fwd_bf16_image(const a3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int act,
                               int pool, int ld_out, uint8_t* argmax, void* ws, size_t ws_bytes, hipStream_t st) {
  const int rlp = (d->s * 2 + 3) / 4 * 4 * 2, np = (d->k + 7) / 8 * 8;      // bf16 elements
  GemmProblem g = fwd_problem(d);
  g.K = d->r * rlp; g.avec = 4; g.bvec = 4; g.no_glds = 1;
  const int ph = d->ho / 2, pw = d->wo / 2;
  if (pool) {
    g.M = d->n * ph * pw * 4;
    g.plain = 1;
  }
  const bool prepared = (d->hints & A3D_HINT_W_PREPARED) != 0;      // w is already the padded bf16 filter (a3d_conv2d_fwd_prepare_filter)
  const size_t ws_used = prepared ? 0 : bf16_image_filter_bytes(d);
  GemmPlan plan = plan_gemm(g, d->precision);
  if ((!ws && ws_used + plan.ws_bytes > 0) || ws_used + plan.ws_bytes > ws_bytes)
    return set_error(A3D_EWORKSPACE, "conv2d_fwd: need %zu workspace bytes", ws_used + plan.ws_bytes);
  A3D_CHECK_ARG(plan.prec == A3D_PREC_BF16 && aligned16(y), "conv2d_fwd: bf16 image form needs the bf16 kernel");
  const __bf16* wp = reinterpret_cast<const __bf16*>(w);
  if (!prepared) {
    int rc = pad_filter_bf16(d, w, static_cast<__bf16*>(ws), st);
    if (rc != A3D_OK) return rc;
    wp = static_cast<const __bf16*>(ws);
  } else {
    A3D_CHECK_ARG(aligned16(w), "conv2d_fwd: a prepared filter is 16-byte aligned");
  }
  IgemmParams p;
  fill_common(p, g);
  p.a16 = 1; p.b16 = 1; p.c16 = (d->storage & A3D_STORE_Y_BF16) ? 1 : 0;
  p.A = x; p.B = reinterpret_cast<const float*>(wp); p.C = y; p.bias = bias; p.act = act;
  p.npix = g.M; p.nrsc = g.K;
  p.H = d->h; p.W = d->w; p.ld = 4; p.pHW = d->h * d->w;
  p.stride = d->stride; p.lstride = ilog2_exact(d->stride); p.pad_t = 0; p.pad_l = 0;
  p.S = 1; p.Cg = rlp;
  p.div_phw = make_fastdiv(d->ho * d->wo); p.div_pw = make_fastdiv(d->wo);
  p.div_c = make_fastdiv(p.Cg); p.div_s = make_fastdiv(1);
  p.div_c_half = make_fastdiv(p.Cg / 2);
  p.ldb = np; p.ldc = d->ldy;
  if (pool) {
    p.pool = 1;
    p.argmax = argmax;
    p.div_phw = make_fastdiv(ph * pw * 4); p.div_pw = make_fastdiv(pw);
    p.ldc = ld_out;
  }
  fill_staging(p, MODE_FWD, (unsigned long long)d->n * d->h * d->w * 4, (unsigned long long)g.K * np, d->r, 1, d->r, 1);
  return launch_igemm(MODE_FWD, plan, 4, 4, p, static_cast<char*>(ws) + ws_used, st);
}

// conv2d forward; pool != 0: y is the 2x2 / stride-2 max pool of the activated conv output, [n, ho/2, wo/2, k] with
// pixel stride ld_out, and the conv output itself is never written
static int conv_fwd_impl(const a3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int act,
                         int pool, int ld_out, uint8_t* argmax, void* ws, size_t ws_bytes, void* stream,
                         const a3d_second_output* out2 = nullptr) {
  int rc = check_desc(d);
  if (rc != A3D_OK) return rc;
  A3D_CHECK_ARG(!(out2 && out2->ptr) || (!pool && !stencil1_applicable(d) && !bf16_image_form_ok(d, x) && !conv3_applicable(d, x)),
                "conv2d_fwd_ex2: a second output on the implicit-GEMM forwards only (no fused pool, no few-channel / one-filter kernels)");
  A3D_CHECK_ARG(x && w && y, "conv2d_fwd: null tensor");
  A3D_CHECK_ARG(act == A3D_ACT_NONE || act == A3D_ACT_RELU || act == A3D_ACT_SIGMOID, "conv2d_fwd: bad act");
  if (pool) {
    const int all16 = A3D_STORE_X_BF16 | A3D_STORE_W_BF16 | A3D_STORE_Y_BF16;
    A3D_CHECK_ARG(d->precision == A3D_PREC_F32 || bf16_image_form_ok(d, x) ||
                      (d->precision == A3D_PREC_BF16 && ((d->storage & all16) == all16 || d->storage == A3D_STORE_Y_BF16)),
                  "conv2d_pool_fwd: fp32, the bf16 image form (4-channel bf16 image, a3d_pad_channels_bf16), or bf16 arithmetic on "
                  "bf16 x, w and y (LDS-DMA kernel)");
    A3D_CHECK_ARG(d->ho >= 2 && d->wo >= 2 && ld_out >= d->k, "conv2d_pool_fwd: output smaller than one pool window");
    A3D_CHECK_ARG(!stencil1_applicable(d), "conv2d_pool_fwd: single-output-channel convs are not supported");
  } else if (stencil1_applicable(d)) {
    A3D_CHECK_ARG(!d->storage, "conv2d_fwd: single-output-channel convs take float32 tensors");
    return stencil1_fwd(d, x, w, bias, y, act, static_cast<hipStream_t>(stream));
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (bf16_image_form_ok(d, x)) {
    if (!conv3b_applicable(d)) return conv_fwd_bf16_image(d, x, w, bias, y, act, pool, ld_out, argmax, ws, ws_bytes, st);
    TimingSlot slot{};                          // straight from L2 on the bf16 matrix cores (conv3.hip, conv3b_fwd_kernel)
    {
      a3d_timing_record& r = slot.rec;
      r.mode = MODE_FWD; r.prec = A3D_PREC_BF16; r.bm = 64; r.bn = d->k > 64 ? 96 : 64; r.waves_m = 1; r.nwaves = 1; r.bk = 16;
      r.avec = 4; r.bvec = 4; r.splitk = 1; r.lds_dma = 2;
      r.m = pool ? d->n * (d->ho / 2) * (d->wo / 2) * 4 : d->n * d->ho * d->wo; r.n = d->k; r.k = d->r * d->s * 3; r.ms = 0.f;
      r.flops = 2.0 * r.m * r.n * r.k;          // (algorithmic: the image's three real channels)
    }
    const bool timed = timing_wanted(slot.rec);
    if (timed && (rc = timing_begin(slot, st)) != A3D_OK) return rc;
    rc = conv3b_fwd(d, x, w, bias, y, act, pool, ld_out, argmax, ws, ws_bytes, st, (d->hints & A3D_HINT_W_PREPARED) != 0);
    if (timed) timing_end(slot, st);
    return rc;
  }
  if (conv3_applicable(d, x)) {                  // few-channel layers: operands straight from L2 (conv3.hip)
    TimingSlot slot{};
    {
      a3d_timing_record& r = slot.rec;
      r.mode = MODE_FWD; r.prec = A3D_PREC_F32; r.bm = 64; r.bn = d->k > 64 ? 96 : 64; r.waves_m = 1; r.nwaves = 1; r.bk = 8;
      r.avec = 4; r.bvec = 4; r.splitk = 1; r.lds_dma = 2;          // 2: conv3_fwd_kernel (filter repack included in ms)
      r.m = pool ? d->n * (d->ho / 2) * (d->wo / 2) * 4 : d->n * d->ho * d->wo; r.n = d->k; r.k = d->r * d->s * d->c; r.ms = 0.f;
      r.flops = 2.0 * r.m * r.n * r.k;
    }
    const bool timed = timing_wanted(slot.rec);
    if (timed && (rc = timing_begin(slot, st)) != A3D_OK) return rc;
    rc = conv3_fwd(d, x, w, bias, y, act, pool, ld_out, argmax, ws, ws_bytes, st, (d->hints & A3D_HINT_W_PREPARED) != 0);
    if (timed) timing_end(slot, st);
    return rc;
  }
  GemmProblem g = fwd_problem(d);
  const int ph = d->ho / 2, pw = d->wo / 2;
  if (pool) {
    g.M = d->n * ph * pw * 4;
    g.plain = 1;
  }
  if (!aligned16(x)) g.avec = 1;
  if (!aligned16(w)) g.bvec = 1;
  if (d->hints & A3D_HINT_SHARE_CU) g.no_glds = 1;      // the register-staged kernels take the hint (launch_one)
  RunForm rf{};
  const bool prepared = (d->hints & A3D_HINT_W_PREPARED) != 0;
  const bool run = run_form_ok(d, x, &rf) && (prepared || (ws && ws_bytes >= run_filter_bytes(d, rf)));
  if (run || g.avec != 4 || g.bvec != 4 || (out2 && out2->ptr)) g.gen2_ok = 0;
  // (the prepared layout was chosen for a 16-byte aligned x — fwd_filter_form — and this launch reads it with THAT row padding)
  A3D_CHECK_ARG(!prepared || (run && aligned16(w) && aligned16(x)),
                "conv2d_fwd: A3D_HINT_W_PREPARED on a launch that reads the filter as stored, or with x off the 16-byte grid");
  size_t ws_used = 0;
  const float* filter = w;
  int run_ldb = d->k;
  if (run) {                                   // window-run form: K = (r, padded run), filter padded into the workspace
    g.K = rf.kp; g.avec = rf.vec;
    const int np = (d->storage & A3D_STORE_W_BF16) ? d->k : (d->k + 3) / 4 * 4;
    if (!prepared) {                           // (prepared: w already is the padded filter, a3d_conv2d_fwd_prepare_filter)
      ws_used = run_filter_bytes(d, rf);
      float* wp = static_cast<float*>(ws);
      clear_stale_error();
      hipLaunchKernelGGL(pad_filter_kernel, dim3((rf.kp * np + 255) / 256), dim3(256), 0, st, w, wp, d->r, rf.rl,
                         rf.rlp, d->k, np);
      rc = check_launch("pad_filter");
      if (rc != A3D_OK) return rc;
      filter = wp;
    }
    run_ldb = np;
    g.bvec = (np % 4 == 0) ? 4 : 1;             // the padded copy is 256-byte aligned
  }
  IgemmParams p;
  fill_common(p, g);
  {
    const int sb = d->storage;
    const int all16 = A3D_STORE_X_BF16 | A3D_STORE_W_BF16 | A3D_STORE_Y_BF16;
    A3D_CHECK_ARG(!sb || ((!pool || sb == A3D_STORE_Y_BF16 || (sb == all16 && d->precision == A3D_PREC_BF16)) &&
                          !(run && (sb & A3D_STORE_W_BF16))),
                  "conv2d_fwd: the fused pool takes float32 inputs (its output may be bf16) or bf16 x, w and y; 3-channel filters stay float32");
    const IgemmParams keep = p;
    rc = apply_storage(p, g, d->precision, sb & A3D_STORE_X_BF16, sb & A3D_STORE_W_BF16, sb & A3D_STORE_Y_BF16, d->c,
                       d->ldx, d->k, d->k, x, w, y, d->k, d->ldy);
    if (rc != A3D_OK) return rc;
    (void)keep;
    // both operands bf16 tensors, whole 16-byte pieces inside one tap: the LDS-DMA kernel may take the launch
    // (pooled: the pooled map's pixel stride and the argmax rows in whole 16-byte pieces too)
    g.ring_ok = p.a16 && p.b16 && !run && d->precision == A3D_PREC_BF16 && d->c % 8 == 0 && d->ldx % 8 == 0 &&
                d->k % 8 == 0 && d->ldy % 8 == 0 && d->r * d->s <= 128 && aligned16(y) && act != A3D_ACT_SIGMOID &&
                (!pool || (p.c16 && ld_out % 8 == 0 && (!argmax || (d->k % 16 == 0 && aligned16(argmax)))));
  }
  GemmPlan plan = plan_gemm(g, d->precision);
  A3D_CHECK_ARG(!(pool && (p.a16 || p.b16)) || plan.ring,
                "conv2d_pool_fwd: on bf16 inputs the fused pool is the LDS-DMA kernel's (channels and strides in whole 16-byte pieces, k %% 16 == 0)");
  if (ws_used + plan.ws_bytes > ws_bytes)
    return set_error(A3D_EWORKSPACE, "conv2d_fwd: need %zu workspace bytes", ws_used + plan.ws_bytes);
  A3D_CHECK_ARG(!(p.a16 || p.b16) || plan.prec == A3D_PREC_BF16, "conv2d_fwd: bf16 operands need vectorisable tensors");
  A3D_CHECK_ARG(!p.c16 || plan.prec == A3D_PREC_BF16 || plan.cfg < kFirstGldsCfg, "conv2d_fwd: no bf16 output from the LDS-DMA kernels");
  p.A = x; p.B = filter; p.C = y; p.bias = bias; p.act = act;
  p.share = (d->hints & A3D_HINT_SHARE_CU) ? 1 : 0;
  p.npix = g.M; p.nrsc = g.K;
  p.H = d->h; p.W = d->w; p.ld = d->ldx; p.pHW = d->h * d->w;
  p.stride = d->stride; p.lstride = ilog2_exact(d->stride); p.pad_t = d->pad_t; p.pad_l = d->pad_l;
  p.S = run ? 1 : d->s; p.Cg = run ? rf.rlp : d->c;
  p.div_phw = make_fastdiv(d->ho * d->wo); p.div_pw = make_fastdiv(d->wo);
  p.div_c = make_fastdiv(p.Cg); p.div_s = make_fastdiv(p.S);
  p.div_c_half = make_fastdiv(std::max(1, p.Cg / 2));
  p.ldb = run ? run_ldb : d->k; p.ldc = d->ldy;
  if (pool) {
    p.pool = 1;
    p.argmax = argmax;
    p.div_phw = make_fastdiv(ph * pw * 4); p.div_pw = make_fastdiv(pw);
    p.ldc = ld_out;
  }
  {
    // nocheck needs the LAST window inside the image: rows (ho-1)*stride + r - 1 < h; columns: the last gathered float
    const bool inside = (d->ho - 1) * d->stride + d->r <= d->h && (d->wo - 1) * d->stride + d->s <= d->w;
    fill_staging(p, MODE_FWD, (unsigned long long)d->n * d->h * d->w * d->ldx, (unsigned long long)g.K * (run ? run_ldb : d->k),
                 run ? d->r : d->r, run ? 1 : d->s, inside ? d->r : 0, inside ? d->s : 0);
  }
  rc = take_second_output(p, out2);
  if (rc != A3D_OK) return rc;
  return launch_igemm(MODE_FWD, plan, g.avec, g.bvec, p, static_cast<char*>(ws) + ws_used, st);
}

int a3d_conv2d_fwd(const a3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int act,
                   void* ws, size_t ws_bytes, void* stream) {
  return conv_fwd_impl(d, x, w, bias, y, act, 0, 0, nullptr, ws, ws_bytes, stream);
}

// Which per-call repack of the filter does this forward do?  0 none, 1 bf16 image form, 2 conv3's [K/4][N][4], 3 window-run padding
static int fwd_filter_form(const a3d_conv_desc* d, RunForm* rf) {
  if (check_desc(d) != A3D_OK || stencil1_applicable(d)) return 0;
  if (bf16_image_form_ok(d, nullptr)) return 1;
  if (conv3_applicable(d, nullptr)) return 2;
  if (run_form_ok(d, nullptr, rf)) return 3;
  return 0;
}

size_t a3d_conv2d_fwd_prepared_filter_bytes(const a3d_conv_desc* d) {
  RunForm rf{};
  switch (fwd_filter_form(d, &rf)) {
    case 1: return conv3b_applicable(d) ? conv3b_filter_bytes(d) : bf16_image_filter_bytes(d);
    case 2: return conv3_ws_bytes(d);
    case 3: return run_filter_bytes(d, rf);
    default: return 0;
  }
}

int a3d_conv2d_fwd_prepare_filter(const a3d_conv_desc* d, const float* w, void* prepared, size_t prepared_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc != A3D_OK) return rc;
  A3D_CHECK_ARG(w && prepared && aligned16(prepared), "conv2d_fwd_prepare_filter: null or misaligned buffer");
  const size_t need = a3d_conv2d_fwd_prepared_filter_bytes(d);
  A3D_CHECK_ARG(need > 0, "conv2d_fwd_prepare_filter: this forward reads the filter as stored");
  if (prepared_bytes < need) return set_error(A3D_EWORKSPACE, "conv2d_fwd_prepare_filter: need %zu bytes", need);
  hipStream_t st = static_cast<hipStream_t>(stream);
  RunForm rf{};
  switch (fwd_filter_form(d, &rf)) {
    case 1: return conv3b_applicable(d) ? conv3b_pack(d, w, prepared, st) : pad_filter_bf16(d, w, static_cast<__bf16*>(prepared), st);
    case 2: return conv3_pack(d, w, static_cast<float*>(prepared), st);
    default: {
      const int np = (d->storage & A3D_STORE_W_BF16) ? d->k : (d->k + 3) / 4 * 4;
      clear_stale_error();
      hipLaunchKernelGGL(pad_filter_kernel, dim3((rf.kp * np + 255) / 256), dim3(256), 0, st, w, static_cast<float*>(prepared), d->r,
                         rf.rl, rf.rlp, d->k, np);
      return check_launch("pad_filter");
    }
  }
}

int a3d_conv2d_fwd_ex2(const a3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int act,
                       const a3d_second_output* out2, void* ws, size_t ws_bytes, void* stream) {
  return conv_fwd_impl(d, x, w, bias, y, act, 0, 0, nullptr, ws, ws_bytes, stream, out2);
}

int a3d_conv2d_pool_fwd(const a3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y_pooled,
                        int ld_pooled, uint8_t* argmax, int act, void* ws, size_t ws_bytes, void* stream) {
  return conv_fwd_impl(d, x, w, bias, y_pooled, act, 1, ld_pooled, argmax, ws, ws_bytes, stream);
}

// One parity class (ph, pw) of a strided bwd-data as a stride-1 problem; false if the class has no pixels.
struct BwdDClass {
  int hc, wc, r0, s0, rp, sp, off_y, off_x;
  GemmProblem g;
};
static bool bwd_d_class(const a3d_conv_desc* d, int ph, int pw, BwdDClass* c) {
  const int st = d->stride;
  c->hc = (d->h - ph + st - 1) / st;
  c->wc = (d->w - pw + st - 1) / st;
  if (c->hc <= 0 || c->wc <= 0) return false;
  c->r0 = (ph + d->pad_t) % st;
  c->s0 = (pw + d->pad_l) % st;
  c->rp = c->r0 < d->r ? (d->r - c->r0 + st - 1) / st : 0;
  c->sp = c->s0 < d->s ? (d->s - c->s0 + st - 1) / st : 0;
  c->off_y = (ph + d->pad_t - c->r0) / st;
  c->off_x = (pw + d->pad_l - c->s0) / st;
  c->g = bwd_d_problem(d);
  c->g.M = d->n * c->hc * c->wc;
  c->g.K = c->rp * c->sp * d->k;          // 0: no tap reaches this class, the launch only writes zeros
  return true;
}

size_t a3d_conv2d_bwd_data_ws_bytes(const a3d_conv_desc* d) {
  if (check_desc(d) != A3D_OK) return 0;
  if (d->stride == 1) {
    size_t need = plan_ws_either(bwd_d_problem(d), d->precision);
    const int both = A3D_STORE_Y_BF16 | A3D_STORE_W_BF16;
    GemmProblem g = bwd_d_problem(d);
    g.ring_ok = d->precision == A3D_PREC_BF16 && (d->storage & both) == both && d->c % 8 == 0 && d->k % 8 == 0;
    g.avec = g.bvec = 4;
    if (g.ring_ok) need = std::max(need, plan_gemm(g, d->precision).ws_bytes);
    return need;
  }
  size_t need = 0;
  for (int ph = 0; ph < d->stride; ++ph)
    for (int pw = 0; pw < d->stride; ++pw) {
      BwdDClass c;
      if (bwd_d_class(d, ph, pw, &c)) need = std::max(need, plan_ws_either(c.g, d->precision));
    }
  return need;
}

int a3d_conv2d_bwd_data(const a3d_conv_desc* d, const float* dz, const float* w, float* dx, const float* relu_mask,
                        void* ws, size_t ws_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc != A3D_OK) return rc;
  A3D_CHECK_ARG(dz && w && dx, "conv2d_bwd_data: null tensor");
  const bool vec_ok_a = aligned16(dz), vec_ok_b = aligned16(w);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // stride 2, fp32: the (up to) four classes as ONE launch of 64x64 tiles when that alone fills the chip
  IgemmMulti multi;
  int n_multi = 0, multi_avec = 4, multi_bvec = 4;
  unsigned multi_grid = 0;
  long multi_tiles = 0;
  double multi_flops = 0;
  const bool try_multi = d->stride == 2 && d->precision == A3D_PREC_F32 && !d->storage && !tune_int("A3D_NO_MULTI", 0) &&
                         tune_int("A3D_FORCE_CFG", -1) < 0;
  // ... and on bf16-stored tensors (BASELINE config 5's conv2d_4): one launch of the bf16 kernel's 128-row tiles
  const int all16 = A3D_STORE_X_BF16 | A3D_STORE_W_BF16 | A3D_STORE_Y_BF16;
  const bool try_multi16 = d->stride == 2 && d->precision == A3D_PREC_BF16 && (d->storage & all16) == all16 && vec_ok_a &&
                           vec_ok_b && !tune_int("A3D_NO_MULTI", 0) && tune_int("A3D_FORCE_SPLITK", 0) <= 0;
  int multi16_bn = 0;
  for (int ph = 0; ph < d->stride; ++ph) {
    for (int pw = 0; pw < d->stride; ++pw) {
      BwdDClass c;
      if (!bwd_d_class(d, ph, pw, &c)) continue;
      GemmProblem g = c.g;
      if (!vec_ok_a) g.avec = 1;
      if (!vec_ok_b) g.bvec = 1;
      if (g.avec != 4 || g.bvec != 4 || g.K == 0) g.gen2_ok = 0;
      IgemmParams p;
      fill_common(p, g);
      rc = apply_storage(p, g, d->precision, d->storage & A3D_STORE_Y_BF16, d->storage & A3D_STORE_W_BF16,
                         d->storage & A3D_STORE_X_BF16, d->k, d->ldy, d->k, d->k, dz, w, dx, d->c, d->ldx);
      if (rc != A3D_OK) return rc;
      A3D_CHECK_ARG(!p.c16 || !relu_mask || aligned16(relu_mask), "conv2d_bwd_data: bf16 mask must be 16-byte aligned");
      g.ring_ok = d->stride == 1 && p.a16 && p.b16 && d->precision == A3D_PREC_BF16 && d->k % 8 == 0 && d->ldy % 8 == 0 &&
                  d->c % 8 == 0 && d->ldx % 8 == 0 && aligned16(dx) && (!relu_mask || aligned16(relu_mask)) && c.rp * c.sp <= 128;
      GemmPlan plan = plan_gemm(g, d->precision);
      A3D_CHECK_ARG(!d->storage || plan.prec == A3D_PREC_BF16, "conv2d_bwd_data: bf16 storage needs vectorisable operands");
      if (try_multi) {                                   // 64x64 tiles, no split-K
        plan = GemmPlan{};
        plan.cfg = 4; plan.splitk = 1; plan.ktiles_per_split = std::max(1, (g.K + 31) / 32);
        plan.tiles_m = (g.M + 63) / 64; plan.tiles_n = (g.N + 63) / 64;
      }
      const bool multi16 = try_multi16 && plan.prec == A3D_PREC_BF16;
      if (multi16) {                                     // the bf16 kernel's tiles, no split-K
        multi16_bn = plan.bf16_bn;
        plan.splitk = 1; plan.ktiles_per_split = std::max(1, (g.K + 63) / 64); plan.ws_bytes = 0;
      }
      if (plan.ws_bytes > ws_bytes)
        return set_error(A3D_EWORKSPACE, "conv2d_bwd_data: need %zu workspace bytes", plan.ws_bytes);
      p.M = g.M; p.N = g.N; p.K = g.K;
      p.A = dz; p.B = w; p.C = dx; p.mask = relu_mask;
      p.npix = g.M; p.nrsc = g.K;
      p.H = d->ho; p.W = d->wo; p.ld = d->ldy; p.pHW = d->ho * d->wo;
      p.stride = 1; p.lstride = 0; p.pad_t = c.off_y; p.pad_l = c.off_x;      // the class is a stride-1 problem
      p.S = std::max(c.sp, 1); p.Cg = d->k; p.Cn = d->c;
      p.div_phw = make_fastdiv(c.hc * c.wc); p.div_pw = make_fastdiv(c.wc);
      p.div_c = make_fastdiv(d->k); p.div_s = make_fastdiv(std::max(c.sp, 1));
      p.div_c_half = make_fastdiv(std::max(1, d->k / 2));
      p.ldb = 0; p.ldc = d->ldx;
      p.sub_step = d->stride; p.sub_ph = ph; p.sub_pw = pw; p.tap_r0 = c.r0; p.tap_s0 = c.s0; p.S_full = d->s;
      p.outW = d->w; p.outHW = d->h * d->w;
      fill_staging(p, MODE_BWD_D, (unsigned long long)d->n * d->ho * d->wo * d->ldy,
                   (unsigned long long)d->r * d->s * d->c * d->k, c.rp, c.sp, 0, 0);
      if (try_multi || multi16) {
        p.splitk = 1; p.ktiles_per_split = plan.ktiles_per_split; p.tiles_m = plan.tiles_m; p.tiles_n = plan.tiles_n;
        p.slab = (size_t)p.M * p.N;
        multi.p[n_multi++] = p;
        multi_avec = std::min(multi_avec, g.avec); multi_bvec = std::min(multi_bvec, g.bvec);
        multi_grid = std::max(multi_grid, (unsigned)((long)plan.tiles_m * plan.tiles_n));
        multi_tiles += (long)plan.tiles_m * plan.tiles_n;
        multi_flops += 2.0 * p.M * p.N * p.K;
        continue;
      }
      rc = launch_igemm(MODE_BWD_D, plan, g.avec, g.bvec, p, ws, st);
      if (rc != A3D_OK) return rc;
    }
  }
  if (try_multi16 && n_multi > 0) {
    TimingSlot slot{};
    {
      a3d_timing_record& r = slot.rec;
      r.mode = MODE_BWD_D; r.prec = A3D_PREC_BF16; r.bm = 128; r.bn = multi16_bn; r.waves_m = 4; r.nwaves = 8; r.bk = 64;
      r.avec = 4; r.bvec = 4; r.splitk = 1; r.lds_dma = 0;
      r.m = d->n * d->h * d->w; r.n = d->c; r.k = d->r * d->s * d->k; r.ms = 0.f;
      r.flops = multi_flops;
    }
    const bool timed = timing_wanted(slot.rec);
    if (timed && (rc = timing_begin(slot, st)) != A3D_OK) return rc;
    rc = launch_igemm_bf16_multi_bwd_d(multi16_bn, multi, multi_grid, (unsigned)n_multi, st);
    if (timed) timing_end(slot, st);
    return rc;
  }
  if (try_multi && n_multi > 0) {
    if (multi_tiles < 256) {                 // too little work for 64x64 tiles without split-K: one launch per class
      for (int i = 0; i < n_multi; ++i) {
        IgemmParams& p = multi.p[i];
        GemmProblem g = bwd_d_problem(d);
        g.M = p.M; g.K = p.K; g.avec = multi_avec; g.bvec = multi_bvec;
        if (g.avec != 4 || g.bvec != 4 || g.K == 0) g.gen2_ok = 0;
        GemmPlan plan = plan_gemm(g, d->precision);
        if (plan.ws_bytes > ws_bytes)
          return set_error(A3D_EWORKSPACE, "conv2d_bwd_data: need %zu workspace bytes", plan.ws_bytes);
        rc = launch_igemm(MODE_BWD_D, plan, g.avec, g.bvec, p, ws, st);
        if (rc != A3D_OK) return rc;
      }
      return A3D_OK;
    }
    TimingSlot slot{};
    {
      a3d_timing_record& r = slot.rec;
      r.mode = MODE_BWD_D; r.prec = A3D_PREC_F32; r.bm = 64; r.bn = 64; r.waves_m = 2; r.nwaves = 4; r.bk = 32;
      r.avec = multi_avec; r.bvec = multi_bvec; r.splitk = 1; r.lds_dma = 0;
      r.m = d->n * d->h * d->w; r.n = d->c; r.k = d->r * d->s * d->k; r.ms = 0.f;
      r.flops = multi_flops;
    }
    const bool timed = timing_wanted(slot.rec);
    if (timed && (rc = timing_begin(slot, st)) != A3D_OK) return rc;
    rc = launch_igemm_multi_bwd_d(multi_avec, multi_bvec, multi, multi_grid, (unsigned)n_multi, st);
    if (timed) timing_end(slot, st);
    return rc;
  }
  return A3D_OK;
}

// bf16 x and bf16 dz, whole 16-byte pieces inside one filter tap, 31-bit byte offsets into x: the LDS-DMA kernel may take the
// filter gradient (BiasAddGrad then comes from colsum_bf16)
static bool bwd_f_ring_ok(const a3d_conv_desc* d) {
  const int both = A3D_STORE_X_BF16 | A3D_STORE_Y_BF16;
  return d->precision == A3D_PREC_BF16 && (d->storage & both) == both && d->c % 8 == 0 && d->ldx % 8 == 0 && d->k % 8 == 0 &&
         d->ldy % 8 == 0 && (double)d->n * d->h * d->w * d->ldx * 2.0 < 2147483647.0 && colsum_bf16_ok(d->k) && d->ldy == d->k;
}

int a3d_conv2d_bwd_both_supported(const a3d_conv_desc* d) {
  return check_desc(d) == A3D_OK && !d->storage && stencil1_bwd_both_applicable(d);
}

size_t a3d_conv2d_bwd_both_ws_bytes(const a3d_conv_desc* d) {
  return a3d_conv2d_bwd_both_supported(d) ? stencil1_bwd_both_ws_bytes(d) : 0;
}

int a3d_conv2d_bwd_both(const a3d_conv_desc* d, const float* x, const float* dz, const float* w, float* dw, float* db,
                        void* dx, int lddx, int dx_bf16, int relu_mask, uint32_t* state, void* ws, size_t ws_bytes,
                        void* stream) {
  int rc = check_desc(d);
  if (rc != A3D_OK) return rc;
  A3D_CHECK_ARG(x && dz && w && dw && dx && state, "conv2d_bwd_both: null tensor");
  A3D_CHECK_ARG(a3d_conv2d_bwd_both_supported(d),
                "conv2d_bwd_both: one output channel, 5x5, stride 1, an even channel count <= 64, float32 tensors, x below 1 GiB");
  A3D_CHECK_ARG(lddx >= d->c && lddx % 2 == 0, "conv2d_bwd_both: lddx must be even and >= c");
  A3D_CHECK_ARG((reinterpret_cast<uintptr_t>(x) & 7) == 0 && (reinterpret_cast<uintptr_t>(w) & 7) == 0 &&
                    (reinterpret_cast<uintptr_t>(dx) & (dx_bf16 ? 3 : 7)) == 0 && (reinterpret_cast<uintptr_t>(ws) & 3) == 0,
                "conv2d_bwd_both: x, w and dx must be aligned to a channel pair");
  if (stencil1_bwd_both_ws_bytes(d) > ws_bytes) return set_error(A3D_EWORKSPACE, "conv2d_bwd_both: workspace too small");
  return stencil1_bwd_both(d, x, dz, w, dw, db, dx, lddx, dx_bf16, relu_mask, state, ws, static_cast<hipStream_t>(stream));
}

// the few-channel layers' filter gradient from LDS-staged input rows (fewch.hip); A3D_FEWCH=0 (tuning processes) keeps the
// window-run form of the generic kernel for A/B runs
static bool fewch_wanted(const a3d_conv_desc* d, bool pooled) {
  return !d->storage && fewch_bwdf_applicable(d, pooled) && tune_int("A3D_FEWCH", 1) != 0;
}

// the launch as the timing list sees it (a3d_timing_*): lds_dma 4 = fewch_bwdf_kernel, ms includes the slab reduction
static int fewch_timed(const a3d_conv_desc* d, const float* x, int src, const void* dz, int ldz, const void* pooled_act,
                       const uint8_t* argmax, int ld_arg, float* dw, float* db, void* ws, hipStream_t st) {
  TimingSlot slot{};
  {
    a3d_timing_record& r = slot.rec;
    r.mode = MODE_BWD_F; r.prec = A3D_PREC_F32; r.bm = 128; r.bn = (d->k + 31) / 32 * 32; r.waves_m = 4; r.nwaves = 4; r.bk = 2;
    r.avec = 1; r.bvec = 1; r.splitk = 1; r.lds_dma = 4;
    r.m = d->r * d->s * d->c; r.n = d->k; r.k = d->n * d->ho * d->wo; r.ms = 0.f;
    r.flops = 2.0 * r.m * r.n * r.k;
  }
  const bool timed = timing_wanted(slot.rec);
  int rc;
  if (timed && (rc = timing_begin(slot, st)) != A3D_OK) return rc;
  rc = fewch_bwd_filter(d, x, src, dz, ldz, pooled_act, argmax, ld_arg, dw, db, ws, st);
  if (timed) timing_end(slot, st);
  return rc;
}

// config 5: bf16 arithmetic on the fp32 image and bf16 pooled tensors (fewch16.hip); A3D_FEWCH16=0 (tuning processes): refused
static bool fewch16_wanted(const a3d_conv_desc* d, bool pooled) {
  return !d->storage && fewch16_bwdf_applicable(d, pooled) && tune_int("A3D_FEWCH16", 1) != 0;
}

size_t a3d_conv2d_bwd_filter_pooled_ws_bytes(const a3d_conv_desc* d) {
  if (check_desc(d) != A3D_OK || d->storage) return 0;
  if (fewch16_wanted(d, true)) return fewch16_bwdf_ws_bytes(d, true);
  if (!fewch_bwdf_applicable(d, true)) return 0;
  return fewch_bwdf_ws_bytes(d, true);
}

int a3d_conv2d_bwd_filter_pooled(const a3d_conv_desc* d, const float* x, const void* dpool, int ld_dpool, const void* pooled,
                                 const uint8_t* argmax, int ld_argmax, int pooled_bf16, float* dw, float* db, void* ws,
                                 size_t ws_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc != A3D_OK) return rc;
  A3D_CHECK_ARG(x && dpool && argmax && dw, "conv2d_bwd_filter_pooled: null tensor");
  if (fewch16_wanted(d, true)) {                    // bf16 arithmetic: float32 image, bf16 pooled tensors
    A3D_CHECK_ARG(pooled_bf16 && ld_dpool >= d->k && ld_argmax >= d->k && ld_dpool % 4 == 0 &&
                      (reinterpret_cast<uintptr_t>(dpool) & 7) == 0 && (reinterpret_cast<uintptr_t>(pooled) & 7) == 0 &&
                      (reinterpret_cast<uintptr_t>(x) & 15) == 0,
                  "conv2d_bwd_filter_pooled: bf16 arithmetic takes bf16 pooled tensors in whole aligned 4-channel groups");
    A3D_CHECK_ARG(fewch_extents_ok(d, true, ld_dpool, 2, ld_argmax), "conv2d_bwd_filter_pooled: pooled tensors of 2 GiB or more");
    if (fewch16_bwdf_ws_bytes(d, true) > ws_bytes) return set_error(A3D_EWORKSPACE, "conv2d_bwd_filter_pooled: workspace too small");
    TimingSlot slot{};
    {
      a3d_timing_record& r = slot.rec;
      r.mode = MODE_BWD_F; r.prec = A3D_PREC_BF16; r.bm = 128; r.bn = (d->k + 31) / 32 * 32; r.waves_m = 4; r.nwaves = 4; r.bk = 16;
      r.avec = 1; r.bvec = 1; r.splitk = 1; r.lds_dma = 4;
      r.m = d->r * d->s * d->c; r.n = d->k; r.k = d->n * d->ho * d->wo; r.ms = 0.f;
      r.flops = 2.0 * r.m * r.n * r.k;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool timed = timing_wanted(slot.rec);
    if (timed && (rc = timing_begin(slot, st)) != A3D_OK) return rc;
    rc = fewch16_bwd_filter(d, x, true, dpool, ld_dpool, pooled, argmax, ld_argmax, dw, db, ws, st);
    if (timed) timing_end(slot, st);
    return rc;
  }
  A3D_CHECK_ARG(!d->storage && fewch_bwdf_applicable(d, true),
                "conv2d_bwd_filter_pooled: an unpadded conv of <= 4 densely packed float32 input channels and 33..96 filters (fp32 "
                "arithmetic; or bf16 arithmetic with bf16 pooled tensors)");
  A3D_CHECK_ARG(ld_dpool >= d->k && ld_argmax >= d->k, "conv2d_bwd_filter_pooled: pixel strides below the filter count");
  A3D_CHECK_ARG(ld_dpool % 4 == 0 && aligned16(x) && (reinterpret_cast<uintptr_t>(dpool) & (pooled_bf16 ? 7 : 15)) == 0 &&
                    (reinterpret_cast<uintptr_t>(pooled) & (pooled_bf16 ? 7 : 15)) == 0,
                "conv2d_bwd_filter_pooled: x, dpool and pooled in whole aligned 4-channel groups (ld_dpool % 4 == 0)");
  A3D_CHECK_ARG(fewch_extents_ok(d, true, ld_dpool, pooled_bf16 ? 2 : 4, ld_argmax), "conv2d_bwd_filter_pooled: pooled tensors of 2 GiB or more");
  if (fewch_bwdf_ws_bytes(d, true) > ws_bytes) return set_error(A3D_EWORKSPACE, "conv2d_bwd_filter_pooled: workspace too small");
  return fewch_timed(d, x, pooled_bf16 ? 2 : 1, dpool, ld_dpool, pooled, argmax, ld_argmax, dw, db, ws,
                     static_cast<hipStream_t>(stream));
}

size_t a3d_conv2d_bwd_filter_ws_bytes(const a3d_conv_desc* d) {
  if (check_desc(d) != A3D_OK) return 0;
  if (stencil1_applicable(d)) return stencil1_bwdf_ws_bytes(d);
  const size_t few = fewch_wanted(d, false) ? fewch_bwdf_ws_bytes(d, false) : 0;      // (an unaligned x falls back to the plans below)
  GemmProblem g0 = bwd_f_problem(d);
  GemmPlan plan = plan_gemm(g0, d->precision);       // the plan without the LDS-DMA kernel: a launch may fall back to it (a dw
  size_t need = plan.ws_bytes + (plan.splitk > 1 ? (size_t)plan.splitk * d->k * 4 : 0);      // off the 16-byte grid: ADVICE r4)
  if (g0.gen2_ok) {                                  // ... or to the first-generation kernel (operands off the 16-byte grid)
    g0.gen2_ok = 0;
    const GemmPlan p1 = plan_gemm(g0, d->precision);
    need = std::max(need, p1.ws_bytes + (p1.splitk > 1 ? (size_t)p1.splitk * d->k * 4 : 0));
  }
  if (bwd_f_ring_ok(d)) {
    g0.ring_ok = 1;
    GemmPlan pr = plan_gemm(g0, d->precision);
    if (pr.ring) need = std::max(need, pr.ws_bytes + colsum_bf16_ws_bytes(d->k));
  }
  RunForm rf;
  if (run_form_ok(d, nullptr, &rf, true)) {
    GemmProblem g = bwd_f_problem(d);
    g.M = rf.kp; g.avec = rf.vec;
    GemmPlan pr = plan_gemm(g, d->precision);
    need = std::max(need, run_filter_bytes(d, rf) + pr.ws_bytes + (pr.splitk > 1 ? (size_t)pr.splitk * d->k * 4 : 0));
  }
  return std::max(need, few);
}

int a3d_conv2d_bwd_filter(const a3d_conv_desc* d, const float* x, const float* dz, float* dw, float* db, void* ws,
                          size_t ws_bytes, void* stream) {
  int rc = check_desc(d);
  if (rc != A3D_OK) return rc;
  A3D_CHECK_ARG(x && dz && dw, "conv2d_bwd_filter: null tensor");
  if (stencil1_applicable(d)) {
    A3D_CHECK_ARG(!d->storage, "conv2d_bwd_filter: single-output-channel convs take float32 tensors");
    if (stencil1_bwdf_ws_bytes(d) > ws_bytes) return set_error(A3D_EWORKSPACE, "conv2d_bwd_filter: workspace too small");
    return stencil1_bwd_filter(d, x, dz, dw, db, ws, static_cast<hipStream_t>(stream));
  }
  if (fewch_wanted(d, false) && aligned16(x) && fewch_extents_ok(d, false, d->ldy, 4, 0)) {
    if (fewch_bwdf_ws_bytes(d, false) > ws_bytes) return set_error(A3D_EWORKSPACE, "conv2d_bwd_filter: workspace too small");
    return fewch_timed(d, x, 0, dz, d->ldy, nullptr, nullptr, 0, dw, db, ws, static_cast<hipStream_t>(stream));
  }
  GemmProblem g = bwd_f_problem(d);
  if (!aligned16(x)) g.avec = 1;
  if (!aligned16(dz)) g.bvec = 1;
  RunForm rf{};
  const bool run = run_form_ok(d, x, &rf, true) && ws && ws_bytes >= run_filter_bytes(d, rf);
  size_t ws_used = 0;
  float* out = dw;
  if (run) {                                   // window-run form: gradient of the PADDED filter, unpadded afterwards
    g.M = rf.kp; g.avec = rf.vec;
    ws_used = run_filter_bytes(d, rf);
    out = static_cast<float*>(ws);
  }
  if (run || g.avec != 4 || g.bvec != 4) g.gen2_ok = 0;
  IgemmParams p;
  fill_common(p, g);
  rc = apply_storage(p, g, d->precision, d->storage & A3D_STORE_X_BF16, d->storage & A3D_STORE_Y_BF16, false, d->c, d->ldx,
                     -1, d->ldy, x, dz, dw, d->k, d->k);
  if (rc != A3D_OK) return rc;
  g.ring_ok = !run && p.a16 && p.b16 && bwd_f_ring_ok(d) && aligned16(dw);
  GemmPlan plan = plan_gemm(g, d->precision);
  A3D_CHECK_ARG(!d->storage || plan.prec == A3D_PREC_BF16, "conv2d_bwd_filter: bf16 storage needs vectorisable operands");
  size_t need = ws_used + plan.ws_bytes + (plan.splitk > 1 && db ? (size_t)plan.splitk * g.N * 4 : 0);
  if (plan.ring) need = plan.ws_bytes + (db ? colsum_bf16_ws_bytes(d->k) : 0);
  if (need > ws_bytes) return set_error(A3D_EWORKSPACE, "conv2d_bwd_filter: need %zu workspace bytes", need);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (plan.ring && db) {                           // the LDS-DMA kernel never holds dz in registers: BiasAddGrad on the side
    // a split launch ends in a reduction kernel anyway: that one adds the partial sums (one launch fewer per layer)
    float* parts = reinterpret_cast<float*>(static_cast<char*>(ws) + plan.ws_bytes);
    int nparts = 0;
    rc = colsum_bf16(dz, g.K, d->k, d->ldy, plan.splitk > 1 ? nullptr : db, parts, &nparts, st);
    if (rc != A3D_OK) return rc;
    if (plan.splitk > 1) { p.dbias_parts = parts; p.dbias_parts_out = db; p.dbias_parts_n = nparts; }
    db = nullptr;
  }
  p.A = x; p.B = dz; p.C = out; p.dbias = db;     // BiasAddGrad = column sums of dz, fused into the same kernel
  p.npix = g.K; p.nrsc = g.M;
  p.H = d->h; p.W = d->w; p.ld = d->ldx; p.pHW = d->h * d->w;
  p.stride = d->stride; p.lstride = ilog2_exact(d->stride); p.pad_t = d->pad_t; p.pad_l = d->pad_l;
  p.S = run ? 1 : d->s; p.Cg = run ? rf.rlp : d->c;
  p.div_phw = make_fastdiv(d->ho * d->wo); p.div_pw = make_fastdiv(d->wo);
  p.div_c = make_fastdiv(p.Cg); p.div_s = make_fastdiv(p.S);
  p.div_c_half = make_fastdiv(std::max(1, p.Cg / 2));
  p.ldb = d->ldy; p.ldc = d->k;
  {
    const bool inside = (d->ho - 1) * d->stride + d->r <= d->h && (d->wo - 1) * d->stride + d->s <= d->w;
    fill_staging(p, MODE_BWD_F, (unsigned long long)d->n * d->h * d->w * d->ldx,
                 (unsigned long long)d->n * d->ho * d->wo * d->ldy, d->r, run ? 1 : d->s, inside ? d->r : 0,
                 inside ? d->s : 0);
  }
  if (run) { p.unpad_dst = dw; p.unpad_rl = rf.rl; p.unpad_rlp = rf.rlp; }
  rc = launch_igemm(MODE_BWD_F, plan, g.avec, g.bvec, p, static_cast<char*>(ws) + ws_used, st);
  if (rc != A3D_OK || !run || p.unpad_done) return rc;       // the split-K reduction stored the unpadded filter itself
  clear_stale_error();
  hipLaunchKernelGGL(unpad_filter_kernel, dim3((d->r * rf.rl * d->k + 255) / 256), dim3(256), 0, st, out, dw, d->r,
                     rf.rl, rf.rlp, d->k);
  return check_launch("unpad_filter");
}

// ---- dense = 1x1 conv over a 1x1 image ----
static a3d_conv_desc dense_desc(int m, int k, int n) {
  a3d_conv_desc d{};
  d.n = m; d.h = 1; d.w = 1; d.c = k; d.k = n; d.r = 1; d.s = 1; d.stride = 1; d.ho = 1; d.wo = 1;
  d.ldx = k; d.ldy = n;
  return d;
}

size_t a3d_dense_fwd_ws_bytes(int m, int k, int n) {
  a3d_conv_desc d = dense_desc(m, k, n);
  if (check_desc(&d) != A3D_OK) return 0;
  // the GEMM a3d_dense_fwd plans (never the few-channel convolution paths: a dense layer of <= 4 inputs is still a GEMM)
  return std::max(plan_ws_either(fwd_problem(&d), A3D_PREC_F32), dense_stream_ws_bytes(m, k, n));
}

int a3d_dense_fwd(int m, int k, int n, const float* x, const float* w, const float* bias, float* y, int act,
                  const uint8_t* drop_keep, void* ws, size_t ws_bytes, void* stream) {
  return a3d_dense_fwd_ex(m, k, n, x, w, bias, y, act, drop_keep, A3D_PREC_F32, 0, ws, ws_bytes, stream);
}

int a3d_dense_fwd_ex(int m, int k, int n, const float* x, const float* w, const float* bias, float* y, int act,
                     const uint8_t* drop_keep, int precision, int storage, void* ws, size_t ws_bytes, void* stream) {
  return a3d_dense_fwd_ex2(m, k, n, x, w, bias, y, n, n, act, drop_keep, precision, storage, nullptr, ws, ws_bytes, stream);
}

int a3d_dense_fwd_ex2(int m, int k, int n, const float* x, const float* w, const float* bias, float* y, int ldy, int ncols_y, int act,
                      const uint8_t* drop_keep, int precision, int storage, const a3d_second_output* out2, void* ws,
                      size_t ws_bytes, void* stream) {
  A3D_CHECK_ARG(m > 0 && k > 0 && n > 0, "dense_fwd: bad dims");
  A3D_CHECK_ARG(ncols_y > 0 && ncols_y <= n && ldy >= ncols_y, "dense_fwd: y rows of %d columns at pitch %d from a GEMM of %d", ncols_y, ldy, n);
  A3D_CHECK_ARG((storage & ~(A3D_STORE_W_BF16 | A3D_STORE_X_BF16)) == 0, "dense_fwd: the weights and x may be bf16, y is float32");
  a3d_conv_desc d = dense_desc(m, k, n);
  d.precision = precision;
  int rc = check_desc(&d);
  if (rc != A3D_OK) return rc;
  A3D_CHECK_ARG(x && w && y, "dense_fwd: null tensor");
  if (precision == A3D_PREC_F32 && !storage && dense_stream_applicable(m, k, n) && aligned16(x) && !out2 && ldy == n && ncols_y == n &&
      !tune_int("A3D_NO_DENSE_KERNELS", 0))
    return dense_fwd_stream(m, k, n, x, w, bias, y, act, drop_keep, 2.f, ws, ws_bytes, static_cast<hipStream_t>(stream));
  GemmProblem g = fwd_problem(&d);
  g.gen2_ok = 0;                                // dense layers: rows of weights, not 128-row pixel tiles
  if (!aligned16(x)) g.avec = 1;
  if (!aligned16(w)) g.bvec = 1;
  IgemmParams p;
  fill_common(p, g);
  rc = apply_storage(p, g, precision, storage & A3D_STORE_X_BF16, storage & A3D_STORE_W_BF16, false, k, k, n, n, x, w, y, n, n);
  if (rc != A3D_OK) return rc;
  g.ring_ok = p.a16 && p.b16 && precision == A3D_PREC_BF16 && k % 8 == 0 && n % 8 == 0 && aligned16(y) && act != A3D_ACT_SIGMOID;
  g.need_reduce = ncols_y != n;                  // rows narrower than the GEMM are stored by the split-K reduction: plan one
  GemmPlan plan = plan_gemm(g, precision);
  A3D_CHECK_ARG(!storage || plan.prec == A3D_PREC_BF16, "dense_fwd: bf16 weights need vectorisable operands");
  A3D_CHECK_ARG(!p.a16 || plan.ring, "dense_fwd: a bf16 x is taken by the LDS-DMA kernel only (bf16 weights, k and n multiples of 8)");
  if (plan.ws_bytes > ws_bytes) return set_error(A3D_EWORKSPACE, "dense_fwd: need %zu workspace bytes", plan.ws_bytes);
  p.A = x; p.B = w; p.C = y; p.bias = bias; p.act = act; p.keep = drop_keep; p.mask_scale = 2.f;
  p.npix = m; p.nrsc = k; p.H = 1; p.W = 1; p.ld = k; p.pHW = 1; p.stride = 1; p.lstride = 0; p.S = 1; p.Cg = k;
  p.div_phw = make_fastdiv(1); p.div_pw = make_fastdiv(1); p.div_c = make_fastdiv(k); p.div_s = make_fastdiv(1);
  p.div_c_half = make_fastdiv(std::max(1, k / 2));
  p.ldb = n; p.ldc = ldy; p.c_cols = ncols_y == n ? 0 : ncols_y;
  rc = take_second_output(p, out2);
  if (rc != A3D_OK) return rc;
  fill_staging(p, MODE_FWD, (unsigned long long)m * k, (unsigned long long)k * n, 1, 1, 1, 1);
  return launch_igemm(MODE_FWD, plan, g.avec, g.bvec, p, ws, static_cast<hipStream_t>(stream));
}

size_t a3d_dense_bwd_data_ws_bytes(int m, int k, int n) {
  a3d_conv_desc d = dense_desc(m, k, n);
  return a3d_conv2d_bwd_data_ws_bytes(&d);
}

int a3d_dense_bwd_data(int m, int k, int n, const float* dz, const float* w, float* dx, const float* mask,
                       int mask_act, float scale, void* ws, size_t ws_bytes, void* stream) {
  return a3d_dense_bwd_data_ex(m, k, n, dz, w, dx, mask, mask_act, scale, A3D_PREC_F32, 0, ws, ws_bytes, stream);
}

int a3d_dense_bwd_data_ex(int m, int k, int n, const float* dz, const float* w, float* dx, const float* mask, int mask_act,
                          float scale, int precision, int storage, void* ws, size_t ws_bytes, void* stream) {
  return a3d_dense_bwd_data_ex2(m, k, n, dz, w, dx, mask, mask_act, scale, precision, storage, nullptr, ws, ws_bytes, stream);
}

int a3d_dense_bwd_data_ex2(int m, int k, int n, const float* dz, const float* w, float* dx, const float* mask, int mask_act,
                           float scale, int precision, int storage, const a3d_second_output* out2, void* ws, size_t ws_bytes,
                           void* stream) {
  A3D_CHECK_ARG(m > 0 && k > 0 && n > 0, "dense_bwd_data: bad dims");
  // storage bits as a3d_conv2d_bwd_data's: Y = dz, W = w, X = dx and the mask
  A3D_CHECK_ARG(precision >= A3D_PREC_F32 && precision <= A3D_PREC_BF16, "dense_bwd_data: unknown precision %d", precision);
  A3D_CHECK_ARG(!mask || mask_act == A3D_ACT_RELU || mask_act == A3D_ACT_SIGMOID, "dense_bwd_data: bad mask_act");
  a3d_conv_desc d = dense_desc(m, k, n);
  int rc = check_desc(&d);
  if (rc != A3D_OK) return rc;
  A3D_CHECK_ARG(dz && w && dx, "dense_bwd_data: null tensor");
  GemmProblem g = bwd_d_problem(&d);
  g.gen2_ok = 0;                                // dense layers: rows of weights, not 128-row pixel tiles
  if (!aligned16(dz)) g.avec = 1;
  if (!aligned16(w)) g.bvec = 1;
  IgemmParams p;
  fill_common(p, g);
  rc = apply_storage(p, g, precision, storage & A3D_STORE_Y_BF16, storage & A3D_STORE_W_BF16, storage & A3D_STORE_X_BF16, n, n, n, n,
                     dz, w, dx, k, k);
  if (rc != A3D_OK) return rc;
  g.ring_ok = p.a16 && p.b16 && precision == A3D_PREC_BF16 && k % 8 == 0 && n % 8 == 0 && aligned16(dx) &&
              (!mask || (aligned16(mask) && mask_act == A3D_ACT_RELU));
  GemmPlan plan = plan_gemm(g, precision);
  A3D_CHECK_ARG(!storage || plan.prec == A3D_PREC_BF16, "dense_bwd_data: bf16 weights need vectorisable operands");
  A3D_CHECK_ARG(!(p.a16 || p.c16) || plan.ring, "dense_bwd_data: bf16 dz / dx are taken by the LDS-DMA kernel only (bf16 weights, k and n multiples of 8, ReLU mask)");
  if (plan.ws_bytes > ws_bytes) return set_error(A3D_EWORKSPACE, "dense_bwd_data: need %zu workspace bytes", plan.ws_bytes);
  p.A = dz; p.B = w; p.C = dx; p.mask = mask; p.mask_scale = scale; p.mask_act = mask_act;
  p.npix = m; p.nrsc = n; p.H = 1; p.W = 1; p.ld = n; p.pHW = 1; p.stride = 1; p.lstride = 0; p.S = 1;
  p.Cg = n; p.Cn = k;
  p.div_phw = make_fastdiv(1); p.div_pw = make_fastdiv(1); p.div_c = make_fastdiv(n); p.div_s = make_fastdiv(1);
  p.div_c_half = make_fastdiv(std::max(1, n / 2));
  p.ldc = k;
  p.S_full = 1;
  rc = take_second_output(p, out2);
  if (rc != A3D_OK) return rc;
  fill_staging(p, MODE_BWD_D, (unsigned long long)m * n, (unsigned long long)k * n, 1, 1, 0, 0);
  return launch_igemm(MODE_BWD_D, plan, g.avec, g.bvec, p, ws, static_cast<hipStream_t>(stream));
}

size_t a3d_dense_bwd_filter_ws_bytes(int m, int k, int n) {
  a3d_conv_desc d = dense_desc(m, k, n);
  return a3d_conv2d_bwd_filter_ws_bytes(&d);
}

int a3d_dense_bwd_filter(int m, int k, int n, const float* x, const float* dz, float* dw, float* db, void* ws,
                         size_t ws_bytes, void* stream) {
  A3D_CHECK_ARG(m > 0 && k > 0 && n > 0, "dense_bwd_filter: bad dims");
  if (dense_dw_applicable(m, k, n) && (long)k * n >= (1L << 16) && !tune_int("A3D_NO_DENSE_KERNELS", 0)) {      // small batch: stream dw once (dense.hip)
    A3D_CHECK_ARG(x && dz && dw, "dense_bwd_filter: null tensor");
    return dense_dw_launch(m, k, n, x, dz, dw, db, static_cast<hipStream_t>(stream));
  }
  a3d_conv_desc d = dense_desc(m, k, n);
  return a3d_conv2d_bwd_filter(&d, x, dz, dw, db, ws, ws_bytes, stream);
}

}  // extern "C"
