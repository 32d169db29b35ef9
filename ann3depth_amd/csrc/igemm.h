// igemm.h — implicit-GEMM convolution on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One kernel template covers the three contractions a conv layer needs, each as C[M,N] = A[M,K] * B[K,N]:
//   FWD   : M = n*ho*wo pixels, N = Cout,  K = r*s*Cin   A = im2col(x)        B = filter [K][N] as stored (HWIO)
//   BWD_D : M = n*h*w  pixels, N = Cin,   K = r*s*Cout  A = im2col^T(dz)     B = filter read as [Cin][(rs,Cout)]
//   BWD_F : M = r*s*Cin,       N = Cout,  K = n*ho*wo   A = im2col(x)^T      B = dz [K][N]
// Dense layers are the 1x1 / 1-pixel special case.  fp32 MFMA is bit-for-bit an fmaf chain, so results carry
// plain fp32 rounding (no reduced-precision inputs).
//
// Block = 4 or 8 waves; each wave owns TM x TN accumulators of 32x32.  The K loop runs in tiles of BK=32
// with a register-prefetched, double-buffered LDS pipeline (global loads of tile t+1 are in flight while the MFMAs
// of tile t run; one __syncthreads per tile).  Inside a tile K is consumed in chunks of 8: lane half h supplies
// k = 8u+4h+j to MFMA j of the chunk, for both operands, so a K-contiguous operand is read with one ds_read_b128.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace a3d {

// In-kernel phase stamps for a separate diagnostic build (make STAMPS=1 -> tools/ab/): where a K-tile iteration spends
// its cycles.  No stamp executes in the shipped library.
#ifdef A3D_STAMPS
#define A3D_STAMP(var)                                                                         \
  do {                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    unsigned long long t_;                                                                     \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                 \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    var = t_;                                                                                  \
  } while (0)
// the constant 100 MHz counter: loop cycles / loop ticks x 100 MHz = the clock the chip held during the loop
#define A3D_RTSTAMP(var)                                                                       \
  do {                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    unsigned long long t_;                                                                     \
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    var = t_;                                                                                  \
  } while (0)
#else
#define A3D_STAMP(var) do { } while (0)
#define A3D_RTSTAMP(var) do { } while (0)
#endif

#ifndef A3D_PIPE_ALL
#define A3D_PIPE_ALL 0
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct FastDiv {
  uint32_t mul, shr, d;
  uint32_t id;       // all ones when d <= 1 (the quotient is n itself), else 0: keeps fdiv branch-free
};

inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d;
  f.id = 0;
  if (d <= 1) { f.mul = 0; f.shr = 0; f.id = 0xffffffffu; return f; }
  uint32_t l = 0;
  while ((1u << l) < d) ++l;                  // l = ceil(log2 d), 1..31
  uint64_t p = 31 + l;
  f.mul = (uint32_t)((((uint64_t)1 << p) + d - 1) / d);
  f.shr = l - 1;
  return f;
}
// exact for n < 2^31
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
  return (__umulhi(n, f.mul) >> f.shr) + (n & f.id);
}

// Out-of-range tile elements (padding halo, K / M / N tails) are LOADED from this zero line instead of being
// branched around: the select is one v_cndmask on the address, the load itself is unconditional.
static __device__ __attribute__((aligned(16))) float g_zero_line[4] = {0.f, 0.f, 0.f, 0.f};

typedef float f32x2 __attribute__((ext_vector_type(2)));

// VEC (1, 2 or 4) consecutive floats <-> registers, as one 4 / 8 / 16-byte access
template <int VEC>
__device__ __forceinline__ void load_vec(const float* src, float (&out)[VEC]) {
  if constexpr (VEC == 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src);
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
  } else if constexpr (VEC == 2) {
    const f32x2 v = *reinterpret_cast<const f32x2*>(src);
    out[0] = v[0]; out[1] = v[1];
  } else {
    out[0] = *src;
  }
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* dst, const float (&in)[VEC]) {
  if constexpr (VEC == 4) {
    const f32x4 v = {in[0], in[1], in[2], in[3]};
    *reinterpret_cast<f32x4*>(dst) = v;
  } else if constexpr (VEC == 2) {
    const f32x2 v = {in[0], in[1]};
    *reinterpret_cast<f32x2*>(dst) = v;
  } else {
    *dst = in[0];
  }
}

// ---- raw buffer loads: out-of-range lanes get kOOB as their offset and the hardware returns zeros for them (no
//      select on a 64-bit address, no zero line, vmcnt only) ----
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t kOOB = 0x80000000u;           // >= every descriptor's num_records (those are clamped to 2^31 - 1)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, unsigned long long bytes) {
  const uint32_t nr = bytes > 0x7fffffffull ? 0x7fffffffu : (uint32_t)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)nr, 0x00020000);
}
// AUX: the instruction's cache policy bits (gfx942+: 1 = sc0, 2 = nt, 16 = sc1).  kAuxStream marks data read once by a
// kernel that streams hundreds of MB (dense-layer weights, Adam slots): non-temporal, so that it does not push the tiles
// a GEMM of the other stream re-reads out of the L2.
constexpr int kAuxStream = 2;
template <int VEC, int AUX = 0>
__device__ __forceinline__ void load_vec_buf(__amdgpu_buffer_rsrc_t r, uint32_t off, float (&out)[VEC]) {
  if constexpr (VEC == 4) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, AUX);
    out[0] = __uint_as_float(v[0]); out[1] = __uint_as_float(v[1]);
    out[2] = __uint_as_float(v[2]); out[3] = __uint_as_float(v[3]);
  } else if constexpr (VEC == 2) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, AUX);
    out[0] = __uint_as_float(v[0]); out[1] = __uint_as_float(v[1]);
  } else {
    out[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, AUX));
  }
}

enum { MODE_FWD = 0, MODE_BWD_D = 1, MODE_BWD_F = 2 };
enum { EPI_RELU = 1, EPI_SIGMOID = 2 };

struct IgemmParams {
  const float* A;       // im2col source tensor (x or dz)
  const float* B;       // FWD: filter; BWD_D: filter; BWD_F: dz
  float* C;             // output (or split-K slabs)
  const float* bias;    // [N] or null            (FWD epilogue)
  const float* mask;    // [M, ldc] or null        (BWD_D epilogue: *= mask>0)
  const uint8_t* keep;  // [M, N] or null          (FWD epilogue: *= keep*mask_scale)
  float* dbias;         // BWD_F: column sums of B (= BiasAddGrad), [N] or [splitk][N] slabs; null = not wanted
  float mask_scale;
  int mask_act;         // EPI_RELU: *= (mask > 0);  EPI_SIGMOID: *= mask * (1 - mask)
  int M, N, K;
  // geometry of the im2col operand
  int npix;             // number of pixels on the pixel axis (FWD/BWD_D: M, BWD_F: K)
  int nrsc;             // r*s*channels on the filter-window axis (FWD/BWD_D: K, BWD_F: M)
  int H, W;             // spatial extent of the SOURCE tensor being gathered
  int ld;               // elements per source pixel
  int stride, lstride, pad_t, pad_l;
  int S;                // filter width
  int Cg;               // channels of the gathered tensor (Cin for FWD/BWD_F, Cout for BWD_D)
  int Cn;               // BWD_D: Cin (row count of B); unused otherwise
  FastDiv div_phw, div_pw, div_c, div_s;   // pixel grid (PH*PW, PW) and window decode (Cg, S)
  int pHW;              // source pixels per image = H*W
  int ldb;              // FWD: N of filter rows; BWD_F: elements per dz pixel
  int ldc;              // elements per output row
  int act;
  int splitk, ktiles_per_split;
  size_t slab;          // elements per split-K slab
  int tiles_m, tiles_n;
  // FWD with the 2x2 / stride-2 max pool fused into the epilogue: rows are enumerated pool window by pool window
  // (row = 4 * window + position), so the four conv outputs of a window sit in ONE lane's accumulator registers and
  // only their maximum is written, to pooled pixel `row / 4`.  div_phw / div_pw then describe the POOLED grid.
  int pool;
  uint8_t* argmax;      // pool: position (0..3, row-major in the window, first maximum) of each written maximum, or null
  // BWD_D with stride > 1 runs one launch per output-parity class (h % stride, w % stride): only the filter taps
  // r = tap_r0 + stride*r', s = tap_s0 + stride*s' reach such a pixel, so each class is a stride-1 problem over a
  // sub-sampled pixel grid and a sub-sampled filter.  sub_step == 1: plain launch.
  int sub_step, sub_ph, sub_pw, tap_r0, tap_s0, S_full, outW, outHW;
  // staging by raw buffer loads (igemm_body): element counts of the two operand tensors (bounds of the descriptors),
  // and the fast column decode of layers whose gathered channel count is a multiple of BK: a K-tile then lies inside ONE
  // filter tap, so (r, s, first channel) are wave-uniform.  kperm walks the taps of a channel chunk before moving to the
  // next chunk (the 25 taps of a 5x5 layer re-read the same few KB of each pixel row: L1/L2-resident) instead of all
  // channels of a tap first.
  unsigned long long a_elems, b_elems;
  int uni, kperm, cpt, ntaps;
  // bf16 storage (igemm_bf16_kernel<.., A16, B16, C16>): which operands are bf16 in HBM, and div_c for the float view
  int a16, b16, c16;
  FastDiv div_c_half;
  FastDiv div_cpt, div_taps;
  // the same tap decode for k-tiles of 64 (the bf16 kernel): uniform when Cg % 64 == 0
  int uni64, kperm64, cpt64;
  FastDiv div_cpt64;
  // stream-K (p.streamk): the (tile, k-tile) iterations, tile-major, are dealt to the blocks in equal contiguous shares
  // (no tile quantisation: 177 tiles x 4 splits = 708 blocks on 512 slots was a 1.4-round launch).  A block that owns a
  // whole tile writes it with the fused epilogue; shares that end inside a tile leave their accumulators, in register
  // order, in slab 2*block (the block's first share) / 2*block + 1 (its last), and igemm_fixup_kernel adds a tile's slabs
  // in block order = ascending k (deterministic) and applies the epilogue.
  int streamk;
  float* sk_ws;                 // [2 * grid][BM * BN] accumulator slabs
  float* sk_bias;               // [2 * grid][BN] BiasAddGrad partial sums (bwd-filter)
  // host only (launch_igemm): a window-run filter gradient whose split-K reduction stores straight into the unpadded
  // filter (run rows of unpad_rl floats out of padded rows of unpad_rlp); unpad_done reports that it did
  float* unpad_dst;
  int unpad_rl, unpad_rlp, unpad_done;
  // host only (launch_igemm): a second copy of the finished output, written by the split-K reduction that writes the first
  // (a3d_second_output: another type, pitch or place — the cast / copy launch that would follow otherwise); c_cols > 0:
  // only the first c_cols columns of the GEMM's N are stored to C (rows of a tensor narrower than the padded GEMM)
  void* out2;
  int out2_ld, out2_step, out2_off, out2_bf16, out2_cols, c_cols;
  // host only (launch_igemm): BiasAddGrad partial sums [dbias_parts_n][N] computed beside the launch (the LDS-DMA bwd-filter never
  // holds dz in registers); the split-K reduction adds them into dbias_parts_out on the side instead of a launch of its own
  const float* dbias_parts; float* dbias_parts_out; int dbias_parts_n;
  FastDiv div_nk;               // k-tiles per tile
  int share;                    // host only: A3D_HINT_SHARE_CU — launch with enough dynamic LDS that <= 8 waves fit a CU
  int dbg;                      // diagnostic builds only: bit 0 / 1 = A / B tile loads fetch nothing
  unsigned long long* stamps;   // diagnostic builds (-DA3D_STAMPS) only: per-wave phase cycle sums; null otherwise
};

// linear pixel of the (sub-)problem -> pixel index in the full output tensor
__device__ __forceinline__ size_t remap_row(int row, int sub_step, int sub_ph, int sub_pw, int outW, int outHW,
                                            const FastDiv& div_phw, const FastDiv& div_pw) {
  if (sub_step <= 1) return (size_t)row;
  const uint32_t n = fdiv((uint32_t)row, div_phw);
  const uint32_t rem = (uint32_t)row - n * div_phw.d;
  const uint32_t i = fdiv(rem, div_pw);
  const uint32_t j = rem - i * div_pw.d;
  return (size_t)n * outHW + (size_t)(sub_step * i + sub_ph) * outW + sub_step * j + sub_pw;
}

template <int MODE, int BM, int BN, int WAVES_M, int NWAVES, int BKT, int AVEC, int BVEC>
struct IgemmCfg {
  static constexpr int BK = BKT;
  static constexpr int NT = 64 * NWAVES;
  static constexpr int WAVES_N = NWAVES / WAVES_M;
  static constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  static constexpr int TM = WM / 32, TN = WN / 32;
  static_assert(TM >= 1 && TN >= 1 && TM * 32 * WAVES_M == BM && TN * 32 * WAVES_N == BN, "tile");
  // LDS images
  // A: FWD/BWD_D [BM][BK+4] (K contiguous, b128 fragment reads); BWD_F [BK][BM+4] (M contiguous, b32 reads)
  static constexpr int A_ROWS = (MODE == MODE_BWD_F) ? BK : BM;
  static constexpr int A_COLS = (MODE == MODE_BWD_F) ? BM : BK;
  // Tiles read k-major (b32 fragment reads: row (k) * LD + column) whose rows are a multiple of 64 floats carry NO pad: rows are
  // then a multiple of 256 bytes apart, and every fragment read is `one lane-constant address + immediate` (ds_read2st64_b32:
  // offsets in units of 256 bytes) — with LD = COLS + 4 each ds_read2_b32 needed a v_add_u32 for its base (its offsets reach
  // 1 KiB only), 23 per tile of 24 MFMAs in the bwd-filter kernel, and beside fp32 MFMAs none of them is hidden.
  // The pad kept the two k halves of a wave (rows 4 apart) off each other's banks; without it the rows of the odd 4-groups
  // store their columns XOR 32 (SWZ), which does the same.
#ifndef A3D_NO_SWZ
#define A3D_NO_SWZ 0
#endif
  static constexpr bool A_SWZ = !A3D_NO_SWZ && (MODE == MODE_BWD_F) && (A_COLS % 64 == 0);
  static constexpr int A_LD = A_SWZ ? A_COLS : A_COLS + 4;
  // B: FWD/BWD_F [BK][BN+4] (N contiguous, b32 reads); BWD_D [BN][BK+4] (K contiguous, b128 reads)
  static constexpr int B_ROWS = (MODE == MODE_BWD_D) ? BN : BK;
  static constexpr int B_COLS = (MODE == MODE_BWD_D) ? BK : BN;
  static constexpr bool B_SWZ = !A3D_NO_SWZ && (MODE != MODE_BWD_D) && (B_COLS % 64 == 0);
  static constexpr int B_LD = B_SWZ ? B_COLS : B_COLS + 4;
  static constexpr int A_ELEMS = A_ROWS * A_LD, B_ELEMS = B_ROWS * B_LD;
  static constexpr int PIX = A_ROWS;   // pixel-table entries (rows of the im2col tile)
  static constexpr size_t LDS_BYTES = (size_t)(2 * (A_ELEMS + B_ELEMS)) * 4 + (size_t)3 * PIX * 16;
};

// ---- pixel table: one int4 per row of the im2col tile: {image base in pixels, y0, x0, valid} ----
template <bool TRANSPOSED>
__device__ __forceinline__ int4 make_pix(const IgemmParams& p, int pixel) {
  int4 e;
  bool valid = pixel < p.npix;
  uint32_t m = valid ? (uint32_t)pixel : 0u;
  uint32_t n = fdiv(m, p.div_phw);
  uint32_t rem = m - n * p.div_phw.d;
  uint32_t po, qo;
  if (!TRANSPOSED && p.pool) {           // rem = 4 * (pooled pixel) + position in its 2x2 window
    const uint32_t win = rem >> 2, sub = rem & 3u;
    const uint32_t wy = fdiv(win, p.div_pw);
    po = 2 * wy + (sub >> 1);
    qo = 2 * (win - wy * p.div_pw.d) + (sub & 1u);
  } else {
    po = fdiv(rem, p.div_pw);
    qo = rem - po * p.div_pw.d;
  }
  e.x = (int)(n * (uint32_t)p.pHW);
  if (TRANSPOSED) {
    e.y = (int)po + p.pad_t;
    e.z = (int)qo + p.pad_l;
  } else {
    e.y = (int)po * p.stride - p.pad_t;
    e.z = (int)qo * p.stride - p.pad_l;
  }
  e.w = valid ? 1 : 0;
  return e;
}

__device__ __forceinline__ float apply_act_grad(float g, float y, int act, float scale) {
  if (act == EPI_SIGMOID) return g * scale * (y * (1.f - y));
  return y > 0.f ? g * scale : 0.f;
}

struct ColDec {
  int r, s, c;
  bool valid;
};
__device__ __forceinline__ ColDec decode_col(const IgemmParams& p, int kcol) {
  ColDec d;
  d.valid = kcol < p.nrsc;
  uint32_t k = d.valid ? (uint32_t)kcol : 0u;
  uint32_t q = fdiv(k, p.div_c);
  d.c = (int)(k - q * p.div_c.d);
  uint32_t r = fdiv(q, p.div_s);
  d.r = (int)r;
  d.s = (int)(q - r * p.div_s.d);
  return d;
}

// Gather ROWS x COLS (pixels x window elements) into registers. Thread t owns column chunk t % CPR and rows
// t / CPR + j * RPP.
template <int NT, int ROWS, int COLS, int VEC, bool TRANSPOSED>
struct Im2colTile {
  static constexpr int CPR = COLS / VEC;
  static_assert(NT % CPR == 0, "cpr");
  static constexpr int RPP = NT / CPR;
  static constexpr int NL = (ROWS + RPP - 1) / RPP;
  static constexpr int NROWS = ROWS;
  static constexpr bool PARTIAL = RPP > ROWS;      // fewer chunks than threads: the upper threads idle
  static_assert(PARTIAL || ROWS % RPP == 0, "rows");

  __device__ __forceinline__ static void load(float (&regs)[NL][VEC], const IgemmParams& p, const int4* pixtab,
                                              const ColDec& cd, int tid) {
    const int r0 = tid / CPR;
    if (PARTIAL && r0 >= ROWS) return;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int4 pt = pixtab[r0 + j * RPP];
      int y = TRANSPOSED ? pt.y - cd.r : pt.y + cd.r;
      int x = TRANSPOSED ? pt.z - cd.s : pt.z + cd.s;
      bool ok = cd.valid && pt.w;
      if (TRANSPOSED) {
        ok = ok && (((y | x) & (p.stride - 1)) == 0) && y >= 0 && x >= 0;
        y >>= p.lstride;
        x >>= p.lstride;
      }
      ok = ok && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
      const uint32_t off = (uint32_t)(pt.x + y * p.W + x) * (uint32_t)p.ld + (uint32_t)cd.c;   // < 2^31 (check_desc)
      load_vec<VEC>(ok ? p.A + off : g_zero_line, regs[j]);
    }
  }
  template <bool SWZ = false>
  __device__ __forceinline__ static void store(const float (&regs)[NL][VEC], float* lds, int ld, int tid) {
    const int r0 = tid / CPR, cq = tid % CPR;
    if (PARTIAL && r0 >= ROWS) return;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int r = r0 + j * RPP;
      float* dst = lds + r * ld + ((cq * VEC) ^ (SWZ ? 32 * ((r >> 2) & 1) : 0));
      store_vec<VEC>(dst, regs[j]);
    }
  }
};

// Plain 2-D tile: src[(row0+r)*ld + col0+c], zero outside [0,rmax) x [0,cmax).
template <int NT, int ROWS, int COLS, int VEC>
struct PlainTile {
  static constexpr int CPR = COLS / VEC;
  static constexpr int TOTAL = ROWS * CPR;
  static constexpr int NL = (TOTAL + NT - 1) / NT;

  __device__ __forceinline__ static void load(float (&regs)[NL][VEC], const float* src, int ld, int row0, int col0,
                                              int rmax, int cmax, int tid) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int idx = tid + j * NT;
      int r = idx / CPR, cq = idx % CPR;
      int gr = row0 + r, gc = col0 + cq * VEC;
      bool ok = (TOTAL % NT == 0 || idx < TOTAL) && gr < rmax && gc < cmax;   // VEC=4 requires cmax % 4 == 0
      const uint32_t off = (uint32_t)gr * (uint32_t)ld + (uint32_t)gc;
      load_vec<VEC>(ok ? src + off : g_zero_line, regs[j]);
    }
  }
  template <bool SWZ = false>
  __device__ __forceinline__ static void store(const float (&regs)[NL][VEC], float* lds, int ld, int tid) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int idx = tid + j * NT;
      if (TOTAL % NT != 0 && idx >= TOTAL) continue;
      int r = idx / CPR, cq = idx % CPR;
      float* dst = lds + r * ld + ((cq * VEC) ^ (SWZ ? 32 * ((r >> 2) & 1) : 0));
      store_vec<VEC>(dst, regs[j]);
    }
  }
};

// BWD_D filter tile: rows = Cin (GEMM N), cols = (rs, Cout) (GEMM K): W[rs][cin][cout].
template <int NT, int ROWS, int COLS, int VEC>
struct FilterTTile {
  static constexpr int CPR = COLS / VEC;
  static constexpr int RPP = NT / CPR;
  static constexpr int NROWS = ROWS;
  static constexpr int NL = (ROWS + RPP - 1) / RPP;
  static constexpr bool PARTIAL = RPP > ROWS;
  static_assert(PARTIAL || ROWS % RPP == 0, "rows");
  __device__ __forceinline__ static void load(float (&regs)[NL][VEC], const IgemmParams& p, int n0, int kcol,
                                              int tid) {
    const int r0 = tid / CPR;
    if (PARTIAL && r0 >= ROWS) return;
    bool kvalid = kcol < p.K;
    uint32_t k = kvalid ? (uint32_t)kcol : 0u;
    uint32_t rs = fdiv(k, p.div_c);              // div_c.d == Cout here
    int ko = (int)(k - rs * p.div_c.d);
    if (p.sub_step > 1) {                        // tap of the sub-sampled filter -> tap of the stored filter
      const uint32_t rp = fdiv(rs, p.div_s);
      const uint32_t sp = rs - rp * p.div_s.d;
      rs = (p.tap_r0 + p.sub_step * rp) * p.S_full + p.tap_s0 + p.sub_step * sp;
    }
    const uint32_t base = rs * (uint32_t)(p.Cn * p.Cg) + (uint32_t)ko;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      int cin = n0 + r0 + j * RPP;
      bool ok = kvalid && cin < p.N;
      const uint32_t off = base + (uint32_t)cin * (uint32_t)p.Cg;
      load_vec<VEC>(ok ? p.B + off : g_zero_line, regs[j]);
    }
  }
  __device__ __forceinline__ static void store(const float (&regs)[NL][VEC], float* lds, int ld, int tid) {
    const int r0 = tid / CPR, cq = tid % CPR;
    if (PARTIAL && r0 >= ROWS) return;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      float* dst = lds + (r0 + j * RPP) * ld + cq * VEC;
      store_vec<VEC>(dst, regs[j]);
    }
  }
};

// stream-K share of block b of `nblk`: iterations [first, last) of the tile-major (tile, k-tile) order
__device__ __forceinline__ uint32_t sk_first(uint32_t b, uint32_t nblk, uint32_t total) {
  return (uint32_t)(((unsigned long long)b * total) / nblk);
}
// Iteration space a block's share is cut from.  streamk 1: all `tiles` x `nk` iterations, dealt to all blocks.
// streamk 2 (bwd-filter, grid a multiple of 8): the K axis — the pixels — is cut into eight slices, one per XCD (the
// remapped ids [x q, (x+1) q) are the blocks of XCD x), and the blocks of an XCD share tiles x (their slice): an XCD then
// streams only its eighth of x and dz through its L2 instead of all of both (tile-major shares: 601 MB of fabric reads
// per conv2d_1..3 bwd-filter launch against 47 MB algorithmic).  Every tile then has contributors on all eight XCDs;
// ascending block id is still ascending k.  The host makes sure a share is no longer than a slice (per <= nk / 8), so a
// block still ends up with at most two partial tiles = its two slab slots.
struct SkSpace { uint32_t blocks, j, k0, nk, total; };
__device__ __forceinline__ SkSpace sk_space(int streamk, uint32_t b, uint32_t nblk, uint32_t tiles, uint32_t nk_total) {
  SkSpace s;
  if (streamk == 2) {
    const uint32_t q = nblk / 8, x = b / q;
    s.blocks = q; s.j = b - x * q;
    s.k0 = x * nk_total / 8;
    s.nk = (x + 1) * nk_total / 8 - s.k0;
  } else {
    s.blocks = nblk; s.j = b; s.k0 = 0; s.nk = nk_total;
  }
  s.total = tiles * s.nk;
  return s;
}

// accumulators of one wave <-> slab, register order: 16 bytes per lane, 1 KiB per wave-instruction
template <int TM, int TN>
__device__ __forceinline__ void slab_store(float* slab, const f32x16 (&acc)[TM][TN], int wave, int lane) {
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
        *reinterpret_cast<f32x4*>(slab + ((((size_t)(wave * TM + a) * TN + b) * 4 + q) * 64 + lane) * 4) = v;
      }
}
template <int TM, int TN>
__device__ __forceinline__ void slab_add(const float* slab, f32x16 (&acc)[TM][TN], int wave, int lane) {
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(slab + ((((size_t)(wave * TM + a) * TN + b) * 4 + q) * 64 + lane) * 4);
        acc[a][b][4 * q] += v[0]; acc[a][b][4 * q + 1] += v[1]; acc[a][b][4 * q + 2] += v[2]; acc[a][b][4 * q + 3] += v[3];
      }
}

// The plain (non-pooling) epilogue of four accumulator registers (rows row0 .. row0+3 of column col): bias / activation /
// dropout (FWD), activation gradient and parity-class row remap (BWD_D), or the raw sums of a classic split-K slab.
template <int MODE>
__device__ __forceinline__ void store_quad(const IgemmParams& p, const f32x4 q, int row0, int col, float* Cout, int ldc,
                                           bool partial) {
  if (col >= p.N) return;
  // values first (every uniform choice branches once per quad, the loads of a mask are issued together), then the four
  // stores back to back: with the branches inside the per-value loop each store was followed by s_waitcnt vmcnt(0) — the
  // mask path's load joins there — and waited for the round trip of the one before (see store_tile_buf)
  float v[4];
  size_t o[4];
  bool ok[4];
  const bool remap = !partial && MODE == MODE_BWD_D;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = row0 + i;
    v[i] = q[i];
    ok[i] = row < p.M;
    o[i] = (remap ? remap_row(ok[i] ? row : 0, p.sub_step, p.sub_ph, p.sub_pw, p.outW, p.outHW, p.div_phw, p.div_pw) : (size_t)row) *
               ldc + col;
  }
  if (!partial) {
    if (MODE == MODE_FWD) {
      const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += bias;
      if (p.act == EPI_RELU) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
      } else if (p.act == EPI_SIGMOID) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = 1.f / (1.f + __expf(-v[i]));
      }
      if (p.keep) {
        uint8_t k[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) k[i] = ok[i] ? p.keep[(size_t)(row0 + i) * p.N + col] : (uint8_t)0;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = k[i] ? v[i] * p.mask_scale : 0.f;
      }
    } else if (MODE == MODE_BWD_D) {
      if (p.mask) {
        float y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) y[i] = ok[i] ? p.mask[o[i]] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = apply_act_grad(v[i], y[i], p.mask_act, p.mask_scale);
      }
    }
  }
  if (p.c16 && !partial) {      // bf16-stored output tensor
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (ok[i]) reinterpret_cast<__bf16*>(Cout)[o[i]] = (__bf16)v[i];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (ok[i]) Cout[o[i]] = v[i];
  }
}
template <int MODE, int TM, int TN, int WM, int WN>
__device__ __forceinline__ void store_tile(const IgemmParams& p, const f32x16 (&acc)[TM][TN], int m0, int n0, int wm,
                                           int wn, int li, int lh, float* Cout, int ldc, bool partial) {
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = {acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]};
        store_quad<MODE>(p, v, m0 + wm * WM + a * 32 + 8 * q + 4 * lh, n0 + wn * WN + b * 32 + li, Cout, ldc, partial);
      }
}

// ---- the same epilogues by raw buffer stores ----
// store_quad branches per value (activation, dropout, output type, row in range), and the compiler closes each value's
// control flow with s_waitcnt vmcnt(0) — the loads of the dropout / mask paths join there — so every store waited for the
// round trip of the one before: 41 k cycles for the 64 stores of a lane against 3.4 k for a stream-K slab of the same
// bytes, 8 - 20 us of every unsplit launch (in-kernel stamps).  Here every uniform choice branches once
// per tile, the accumulators are finished in place (bias, activation, activation gradient: the mask's loads issued
// together), and the stores follow each other with one v_add between them: the descriptor starts at the tile's first
// row, so a lane's offset is `lane constant + uniform row offset`, and rows >= M / columns >= N fall outside
// num_records (kOOB) instead of being branched around.
// (base and extent are block-uniform, but the tile coordinates they come from live in vector registers — an integer
// division — and a descriptor in vector registers makes the compiler wrap every access in a waterfall loop)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t epi_rsrc(const void* base, size_t byte_off, uint32_t bytes) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(base) + byte_off;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)hi << 32) | lo), 0,
                                           (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
// Which tiles may take it: row remaps (the parity classes of a strided bwd-data) and dropout (dense layers only) keep the
// per-value form, and so does a pitch whose 128 rows would not fit a 31-bit offset.
template <int MODE>
__device__ __forceinline__ bool epi_buf_ok(const IgemmParams& p, int ldc, bool partial) {
  if (ldc >= (1 << 21)) return false;
  if (partial) return true;
  if (MODE == MODE_BWD_D && p.sub_step > 1) return false;
  if (MODE == MODE_FWD && p.keep) return false;
  return true;
}
template <int MODE, int BM, int TM, int TN, int WM, int WN, bool MASK16, bool FASTEXP = true>
__device__ __forceinline__ void store_tile_buf(const IgemmParams& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                               int li, int lh, float* Cout, int ldc, bool partial, bool c16) {
  int rows_left = p.M - m0;
  rows_left = rows_left < 0 ? 0 : (rows_left > BM ? BM : rows_left);
  const uint32_t row_el = (uint32_t)((wm * WM + 4 * lh) * ldc);      // this lane's first row of the tile, in elements
  uint32_t elb[TN];
  bool okb[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int col = n0 + wn * WN + b * 32 + li;
    okb[b] = col < p.N;
    elb[b] = row_el + (uint32_t)col;
  }
  auto row_off = [&](int a, int e) -> uint32_t { return (uint32_t)((a * 32 + 8 * (e >> 2) + (e & 3)) * ldc); };   // uniform
  if (!partial) {
    if (MODE == MODE_FWD) {
      float bias[TN];
#pragma unroll
      for (int b = 0; b < TN; ++b) bias[b] = (p.bias && okb[b]) ? p.bias[n0 + wn * WN + b * 32 + li] : 0.f;
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[a][b][e] += bias[b];
      if (p.act == EPI_RELU) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = fmaxf(acc[a][b][e], 0.f);
      } else if (p.act == EPI_SIGMOID) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 1.f / (1.f + (FASTEXP ? __expf(-acc[a][b][e]) : expf(-acc[a][b][e])));
      }
    } else if (MODE == MODE_BWD_D) {
      if (p.mask) {
        constexpr uint32_t MSZ = MASK16 ? 2u : 4u;
        const __amdgpu_buffer_rsrc_t rsM = epi_rsrc(p.mask, (size_t)m0 * ldc * MSZ, (uint32_t)(rows_left * ldc) * MSZ);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            const uint32_t vb = okb[b] ? elb[b] * MSZ : kOOB;
            float y[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
              if constexpr (MASK16)
                y[e] = (float)__builtin_bit_cast(
                    __bf16, (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsM, (int)(vb + row_off(a, e) * MSZ), 0, 0));
              else
                y[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsM, (int)(vb + row_off(a, e) * MSZ), 0, 0));
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = apply_act_grad(acc[a][b][e], y[e], p.mask_act, p.mask_scale);
          }
      }
    }
  }
  if (c16) {
    const __amdgpu_buffer_rsrc_t rsC = epi_rsrc(Cout, (size_t)m0 * ldc * 2u, (uint32_t)(rows_left * ldc) * 2u);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const uint32_t vb = okb[b] ? elb[b] * 2u : kOOB;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)acc[a][b][e]), rsC,
                                                (int)(vb + row_off(a, e) * 2u), 0, 0);
      }
  } else {
    const __amdgpu_buffer_rsrc_t rsC = epi_rsrc(Cout, (size_t)m0 * ldc * 4u, (uint32_t)(rows_left * ldc) * 4u);
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const uint32_t vb = okb[b] ? elb[b] * 4u : kOOB;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[a][b][e]), rsC, (int)(vb + row_off(a, e) * 4u), 0, 0);
      }
  }
}
// The fused 2x2 max pool: window g of an accumulator is its registers 4 g .. 4 g + 3 (conv rows row, .., row + 3), written
// to pooled row `row / 4`.  Descriptors start at the tile's first pooled row.
template <int BM, int TM, int TN, int WM, int WN, bool ROUND16 = false, bool FASTEXP = true>
__device__ __forceinline__ void store_tile_pool_buf(const IgemmParams& p, f32x16 (&acc)[TM][TN], int m0, int n0, int wm, int wn,
                                                    int li, int lh, float* Cout, int ldc, bool c16) {
  static_assert(BM % 4 == 0 && WM % 4 == 0, "pool windows");
  int rows_left = (p.M - m0) >> 2;      // pooled rows of this tile (host: M % 4 == 0)
  rows_left = rows_left < 0 ? 0 : (rows_left > BM / 4 ? BM / 4 : rows_left);
  const int prow = wm * (WM / 4) + lh;      // this lane's first pooled row of the tile
  float val[TM][TN][4];
  uint32_t arg[TM][TN];                     // four 2-bit positions
  float bias[TN];
  bool okb[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int col = n0 + wn * WN + b * 32 + li;
    okb[b] = col < p.N;
    bias[b] = (p.bias && okb[b]) ? p.bias[col] : 0.f;
  }
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      arg[a][b] = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // the values a separate conv would have stored, compared the way MaxPool / MaxPoolGrad scan them
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[a][b][4 * g + i] + bias[b];
        if (p.act == EPI_RELU) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        } else if (p.act == EPI_SIGMOID) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = 1.f / (1.f + (FASTEXP ? __expf(-v[i]) : expf(-v[i])));
        }
        if (ROUND16) {      // the bf16-storage kernels compare what a separate conv would have stored: rounded values
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = (float)(__bf16)v[i];
        }
        float m = v[0];
        uint32_t w = 0;
#pragma unroll
        for (int i = 1; i < 4; ++i)
          if (v[i] > m) { m = v[i]; w = (uint32_t)i; }
        val[a][b][g] = m;
        arg[a][b] |= w << (2 * g);
      }
    }
  auto row_off = [&](int a, int g, int ld) -> uint32_t { return (uint32_t)((a * 8 + 2 * g) * ld); };      // uniform
  const uint32_t esz = c16 ? 2u : 4u;
  const __amdgpu_buffer_rsrc_t rsC = epi_rsrc(Cout, (size_t)(m0 >> 2) * ldc * esz, (uint32_t)(rows_left * ldc) * esz);
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const uint32_t el = (uint32_t)(prow * ldc + n0 + wn * WN + b * 32 + li);
      if (c16) {
        const uint32_t vb = okb[b] ? el * 2u : kOOB;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (__bf16)val[a][b][g]), rsC,
                                                (int)(vb + row_off(a, g, ldc) * 2u), 0, 0);
      } else {
        const uint32_t vb = okb[b] ? el * 4u : kOOB;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(val[a][b][g]), rsC, (int)(vb + row_off(a, g, ldc) * 4u), 0, 0);
      }
    }
  if (p.argmax) {
    const __amdgpu_buffer_rsrc_t rsA = epi_rsrc(p.argmax, (size_t)(m0 >> 2) * p.N, (uint32_t)(rows_left * p.N));
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const uint32_t vb = okb[b] ? (uint32_t)(prow * p.N + n0 + wn * WN + b * 32 + li) : kOOB;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          __builtin_amdgcn_raw_buffer_store_b8((uint8_t)((arg[a][b] >> (2 * g)) & 3u), rsA, (int)(vb + row_off(a, g, p.N)), 0, 0);
      }
  }
}

// What a block does with the accumulators of one share of one tile: a share that covers the tile's whole K range stores it
// through the mode's epilogue (bias / activation / dropout, the fused 2x2 max pool, the activation gradient), a classic
// split-K share stores raw sums into its slab, a stream-K share that ends inside the tile leaves the accumulators (and the
// bias-gradient sums of `do_bias` threads) in the block's slab slot for igemm_fixup_kernel.  Shared by igemm_body and the
// LDS-DMA kernel of igemm2.h (same accumulator layout: wave (wm, wn), accumulators [TM][TN] of 32x32).
template <int MODE, int BM, int BN, int TM, int TN, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, f32x16 (&acc)[TM][TN], const int split, const uint32_t bid,
                                               const int seg, const int kt_begin, const int kt_end, const int nk_total,
                                               const bool do_bias, const float bsum, const int tid, const int wave, const int lane,
                                               const int m0, const int n0, const int wm, const int wn) {
  const int li = lane & 31, lh = lane >> 5;
  float* Cout = p.C;
  int ldc = p.ldc;
  const bool partial = !p.streamk && p.splitk > 1;
  if (partial) {
    Cout = p.C + (size_t)split * p.slab;
    ldc = p.N;
  }
  if (p.streamk && (kt_begin != 0 || kt_end != nk_total)) {
    // a share that ends inside the tile: accumulators (and the bias-gradient sums) go to this block's slab as they are
    const size_t slot = (size_t)2 * bid + (seg > 0 ? 1 : 0);
    slab_store<TM, TN>(p.sk_ws + slot * (size_t)(BM * BN), acc, wave, lane);
    if (MODE == MODE_BWD_F && do_bias) p.sk_bias[slot * BN + tid] = bsum;
  } else if (MODE == MODE_FWD && p.pool) {      // never split (host)
    store_tile_pool_buf<BM, TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, li, lh, Cout, ldc, p.c16 != 0);
  } else {
    if (MODE == MODE_BWD_F && do_bias && n0 + tid < p.N) p.dbias[(partial ? (size_t)split * p.N : 0) + n0 + tid] = bsum;
    if (epi_buf_ok<MODE>(p, ldc, partial))
      store_tile_buf<MODE, BM, TM, TN, WM, WN, false>(p, acc, m0, n0, wm, wn, li, lh, Cout, ldc, partial, p.c16 && !partial);
    else
      store_tile<MODE, TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, li, lh, Cout, ldc, partial);
  }
}

// The whole GEMM of one block: `nwg` blocks work on problem `p`, this one is number `bid_in`.
template <int MODE, int BM, int BN, int WAVES_M, int NWAVES, int BKT, int AVEC, int BVEC>
__device__ __forceinline__ void igemm_body(const IgemmParams& p, const uint32_t nwg, const uint32_t bid_in) {
  using Cfg = IgemmCfg<MODE, BM, BN, WAVES_M, NWAVES, BKT, AVEC, BVEC>;
  constexpr int NT = Cfg::NT;
  constexpr int BK = Cfg::BK;
  constexpr int TM = Cfg::TM, TN = Cfg::TN;
  constexpr bool TRANSPOSED = (MODE == MODE_BWD_D);
  using ATile = Im2colTile<NT, Cfg::A_ROWS, Cfg::A_COLS, AVEC, TRANSPOSED>;
  using BTile = typename std::conditional<MODE == MODE_BWD_D, FilterTTile<NT, Cfg::B_ROWS, Cfg::B_COLS, BVEC>,
                                          PlainTile<NT, Cfg::B_ROWS, Cfg::B_COLS, BVEC>>::type;
  constexpr int BNL = BTile::NL;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* As = reinterpret_cast<float*>(smem_raw);
  float* Bs = As + 2 * Cfg::A_ELEMS;
  int4* pixtab = reinterpret_cast<int4*>(Bs + 2 * Cfg::B_ELEMS);
#ifdef A3D_STAMPS
  unsigned long long t_entry = 0;
  A3D_STAMP(t_entry);
#endif

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const int li = lane & 31, lh = lane >> 5;

  // ---- block -> (tile_m, tile_n, split): XCD-aware bijective remap of the linear id, tile_n fastest so the
  //      blocks that re-read one im2col panel share an XCD's L2 ----
  uint32_t bid = bid_in;
  {
    uint32_t q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_mn = p.tiles_m * p.tiles_n;
  const int nk_total = (p.K + BK - 1) / BK;      // p.ktiles_per_split is in units of this kernel's BK
  // stream-K share of this block: a contiguous range of the tile-major (tile, k-tile) iterations of its iteration space —
  // the whole problem (streamk 1), or (streamk 2, bwd-filter) the K slice of the block's XCD: see SkSpace
  const SkSpace sp = sk_space(p.streamk, bid, nwg, (uint32_t)tiles_mn, (uint32_t)nk_total);
  uint32_t sk_cur = p.streamk ? sk_first(sp.j, sp.blocks, sp.total) : 0u;
  const uint32_t sk_end = p.streamk ? sk_first(sp.j + 1, sp.blocks, sp.total) : 1u;
  for (int seg = 0; sk_cur < sk_end; ++seg) {      // classic launches: exactly one pass
  int split = 0, tmn, kt_begin, kt_end;
  if (p.streamk) {
    tmn = p.streamk == 1 ? (int)fdiv(sk_cur, p.div_nk) : (int)(sk_cur / sp.nk);
    const uint32_t kl = sk_cur - (uint32_t)tmn * sp.nk;
    const uint32_t n = min(sk_end - sk_cur, sp.nk - kl);
    kt_begin = (int)(sp.k0 + kl);
    kt_end = kt_begin + (int)n;
    sk_cur += n;
  } else {
    split = bid / tiles_mn;
    tmn = bid - split * tiles_mn;
    kt_begin = split * p.ktiles_per_split;
    kt_end = kt_begin + p.ktiles_per_split;
    if (kt_end > nk_total) kt_end = nk_total;
    sk_cur = sk_end;
  }
  const int tile_m = tmn / p.tiles_n, tile_n = tmn - tile_m * p.tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = kt_end - kt_begin;
  if (seg > 0) __syncthreads();                    // the previous share's tiles and row table are dead from here on

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  float ra[ATile::NL][AVEC];
  float rb[BNL][BVEC];
  // BiasAddGrad rides along in BWD_F: the blocks of the first M-tile also sum the dz tiles they stage (column n0+tid)
  const bool do_bias = (MODE == MODE_BWD_F) && p.dbias != nullptr && tile_m == 0 && tid < BN;
  float bsum = 0.f;

  // ===== staging: global -> registers by raw buffer loads.  Whatever does not change from K-tile to K-tile is computed
  // ONCE per lane (row offsets, pixel coordinates, column validity); per tile only the filter tap moves — one scalar
  // decode when the tap is wave-uniform (p.uni), one per-lane decode otherwise — and the B tiles move by re-basing their
  // descriptor (scalar).  The previous form recomputed every address per tile with ~75 VALU instructions and several
  // dependent LDS reads per wave: a third of each wave's time with no MFMA issued (in-kernel stamps).
  constexpr int A_CPR = ATile::CPR, A_RPP = ATile::RPP, A_NL = ATile::NL;
  const int a_r0 = tid / A_CPR, a_cq = tid % A_CPR;
  const bool a_on = !ATile::PARTIAL || a_r0 < Cfg::A_ROWS;
  constexpr int SGN = TRANSPOSED ? -1 : 1;
  const int pW = p.W, pld = p.ld;

  // image of the tile's first pixel: descriptors are based there, so lane offsets stay far below 2^31
  auto image_of = [&](int pixel) -> uint32_t {
    const int px = pixel < p.npix ? pixel : p.npix - 1;
    return fdiv((uint32_t)px, p.div_phw);
  };
  // row table entry: {byte offset of the row's reference pixel from the base image, y0, x0, valid}
  auto row_entry = [&](int pixel, uint32_t nf) -> int4 {
    int4 e = make_pix<TRANSPOSED>(p, pixel);
    e.x = ((e.x - (int)(nf * (uint32_t)p.pHW)) + e.y * pW + e.z) * pld * 4;
    return e;
  };

  // ---- A operand ----
  int a_rowoff[A_NL], a_y0[A_NL], a_x0[A_NL];
  int a_dy = 0, a_dx = 0, a_coloff = 0;    // BWD_F: this lane's fixed window element
  bool a_cvalid = true;
  const float* a_base = p.A;
  unsigned long long a_bytes = 0;
  if constexpr (MODE == MODE_BWD_F) {
    // pixel axis is K: row tables of tiles kt_begin and kt_begin+1; three buffers indexed by (tile - kt_begin) % 3
    if (tid < 2 * Cfg::PIX) {
      const int which = tid / Cfg::PIX, e = tid % Cfg::PIX;
      const int pix0 = (kt_begin + which) * BK;
      pixtab[which * Cfg::PIX + e] = row_entry(pix0 + e, image_of(pix0));
    }
    const ColDec d = decode_col(p, m0 + a_cq * AVEC);
    a_dy = d.r; a_dx = d.s;
    a_coloff = ((d.r * pW + d.s) * pld + d.c) * 4;
    a_cvalid = d.valid;
  } else {
    const uint32_t nf = image_of(m0);
    if (tid < Cfg::PIX) pixtab[tid] = row_entry(m0 + tid, nf);
    const unsigned long long boff = (unsigned long long)nf * (unsigned long long)p.pHW * (unsigned long long)pld;
    a_base = p.A + boff;
    a_bytes = (p.a_elems - boff) * 4ull;
  }
  __syncthreads();
  if constexpr (MODE != MODE_BWD_F) {
    if (a_on) {
#pragma unroll
      for (int j = 0; j < A_NL; ++j) {
        const int4 pt = pixtab[a_r0 + j * A_RPP];
        a_rowoff[j] = pt.x + (p.uni ? a_cq * AVEC * 4 : 0);
        a_y0[j] = pt.w ? pt.y : -(1 << 30);
        a_x0[j] = pt.z;
      }
    }
  }

  // k-tile of iteration index kt: (filter tap, first channel) when the tap is wave-uniform
  struct TapPos { uint32_t rs, chunk; };
  auto tap_of = [&](int kt) -> TapPos {
    // both orders are a handful of scalar operations: computed side by side and selected, no branch in the K loop
    const uint32_t c1 = fdiv((uint32_t)kt, p.div_taps), r1 = (uint32_t)kt - c1 * (uint32_t)p.ntaps;
    const uint32_t r2 = fdiv((uint32_t)kt, p.div_cpt), c2 = (uint32_t)kt - r2 * (uint32_t)p.cpt;
    TapPos t;
    t.rs = p.kperm ? r1 : r2;
    t.chunk = p.kperm ? c1 : c2;
    return t;
  };

  // ---- B operand ----
  constexpr int B_CPR = BTile::CPR;
  uint32_t b_voff[BNL];                    // loop-invariant lane offsets (kOOB: outside the tensor's columns / rows)
  const int b_cq = tid % B_CPR;            // BWD_D
  if constexpr (MODE == MODE_BWD_D) {
    const int r0 = tid / B_CPR;
#pragma unroll
    for (int j = 0; j < BNL; ++j) {
      const int row = r0 + j * BTile::RPP;
      const bool ok = (!BTile::PARTIAL || r0 < Cfg::B_ROWS) && n0 + row < p.N;
      b_voff[j] = ok ? (uint32_t)((row * p.Cg + b_cq * BVEC) * 4) : kOOB;
    }
  } else {
#pragma unroll
    for (int j = 0; j < BNL; ++j) {
      const int idx = tid + j * NT;
      const int r = idx / B_CPR, cq = idx % B_CPR;
      const bool ok = (BTile::TOTAL % NT == 0 || idx < BTile::TOTAL) && n0 + cq * BVEC < p.N;   // BVEC = 4: N % 4 == 0
      b_voff[j] = ok ? (uint32_t)((r * p.ldb + cq * BVEC) * 4) : kOOB;
    }
  }

  // ---- state of the NEXT tile to fetch: lane offsets and the two descriptors.  prepare() computes it (scalar tap
  // decode / row-table reads / a few VALU per row) and is placed in the shadow of the current tile's MFMAs; issue() at the
  // top of an iteration is then nothing but the buffer loads.
  uint32_t a_off[A_NL];
  uint32_t b_off[BNL];
  __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A, 0), rsB = make_rsrc(p.B, 0);
  auto prepare = [&](auto uni_c, auto fetch_c, int kt, int slot, bool live) {       // slot: row-table buffer of tile kt (BWD_F)
    constexpr bool FETCH = decltype(fetch_c)::value;  // issue each load as soon as its offset exists (no state kept)
    constexpr bool UNI = decltype(uni_c)::value;      // wave-uniform tap: hoisted out of the K loop (two loop bodies)
    // ---------- A ----------
    if constexpr (MODE == MODE_BWD_F) {
      const int pix0 = kt * BK;
      const uint32_t nf = image_of(pix0);
      const unsigned long long boff = (unsigned long long)nf * (unsigned long long)p.pHW * (unsigned long long)pld;
      rsA = make_rsrc(p.A + boff, live ? (p.a_elems - boff) * 4ull : 0ull);
      if (a_on) {
        const int4* ptab = pixtab + slot * Cfg::PIX;
#pragma unroll
        for (int j = 0; j < A_NL; ++j) {
          const int4 pt = ptab[a_r0 + j * A_RPP];
          const int y = pt.y + a_dy, x = pt.z + a_dx;
          const bool ok = a_cvalid & (pt.w != 0) & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)pW);
          // (all four fields of the entry used unconditionally: ONE 16-byte LDS read — selecting `pt.x + a_coloff` made the
          // compiler read x under an exec-mask branch of its own, a second LDS round trip per piece)
          a_off[j] = (uint32_t)(pt.x + a_coloff) | (ok ? 0u : kOOB);
          if (FETCH) load_vec_buf<AVEC>(rsA, a_off[j], ra[j]);
        }
      }
    } else {
      int dy, dx, coloff;
      bool cv = true;
      if constexpr (UNI) {
        const TapPos t = tap_of(kt);
        const uint32_t r = fdiv(t.rs, p.div_s), sx = t.rs - r * p.div_s.d;
        dy = SGN * (int)r; dx = SGN * (int)sx;
        coloff = ((dy * pW + dx) * pld + (int)t.chunk * BK) * 4;
      } else {
        const ColDec d = decode_col(p, kt * BK + a_cq * AVEC);
        dy = SGN * d.r; dx = SGN * d.s;
        coloff = ((dy * pW + dx) * pld + d.c) * 4;
        cv = d.valid;
      }
      rsA = make_rsrc(a_base, live ? a_bytes : 0ull);
      if (a_on) {
#pragma unroll
        for (int j = 0; j < A_NL; ++j) {
          // rows that do not exist carry y0 = INT_MIN/2 and fail the range test like any halo pixel
          const int y = a_y0[j] + dy, x = a_x0[j] + dx;
          const bool ok = cv & ((unsigned)y < (unsigned)p.H) & ((unsigned)x < (unsigned)pW);
          a_off[j] = ok ? (uint32_t)(a_rowoff[j] + coloff) : kOOB;
          if (FETCH) load_vec_buf<AVEC>(rsA, a_off[j], ra[j]);
        }
      }
    }
    // ---------- B ----------
    if constexpr (MODE == MODE_BWD_D) {
      // filter W[rs][cin][cout] read as rows = cin, columns = k = (rs, cout)
      if constexpr (UNI) {
        const TapPos t = tap_of(kt);
        // tap of the (sub-sampled, see sub_step) filter -> tap of the stored filter; the identity for plain launches
        const uint32_t rp = fdiv(t.rs, p.div_s), sp = t.rs - rp * p.div_s.d;
        const uint32_t rs = (p.tap_r0 + p.sub_step * rp) * p.S_full + p.tap_s0 + p.sub_step * sp;
        const unsigned long long boff = (unsigned long long)rs * (unsigned long long)(p.Cn * p.Cg) +
                                        (unsigned long long)n0 * p.Cg + t.chunk * BK;
        rsB = make_rsrc(p.B + boff, live ? (p.b_elems - boff) * 4ull : 0ull);
#pragma unroll
        for (int j = 0; j < BNL; ++j) b_off[j] = b_voff[j];
      } else {
        const int kcol = kt * BK + b_cq * BVEC;
        const bool kvalid = kcol < p.K;
        const uint32_t k = kvalid ? (uint32_t)kcol : 0u;
        const uint32_t rs0 = fdiv(k, p.div_c);       // div_c.d == Cout here
        const int ko = (int)(k - rs0 * p.div_c.d);
        const uint32_t rp = fdiv(rs0, p.div_s), sp = rs0 - rp * p.div_s.d;
        const uint32_t rs = (p.tap_r0 + p.sub_step * rp) * p.S_full + p.tap_s0 + p.sub_step * sp;
        const uint32_t base = (rs * (uint32_t)(p.Cn * p.Cg) + (uint32_t)n0 * (uint32_t)p.Cg + (uint32_t)ko) * 4u;
        rsB = make_rsrc(p.B, live ? p.b_elems * 4ull : 0ull);
#pragma unroll
        for (int j = 0; j < BNL; ++j)
          b_off[j] = (kvalid && b_voff[j] != kOOB) ? base + (b_voff[j] - (uint32_t)(b_cq * BVEC * 4)) : kOOB;
      }
    } else {
      // plain [K][ldb] tile at rows row0.., columns n0..: the descriptor is re-based (scalar) and ends with the tensor's
      // row K-1, so rows of the K tail read as zeros
      int row0 = kt * BK;
      if constexpr (MODE == MODE_FWD && UNI) {
        const TapPos t = tap_of(kt);
        row0 = (int)(t.rs * (uint32_t)p.Cg + t.chunk * BK);
      }
      // selects, not branches: the K loop stays one basic block
      const int rows = p.K - row0;
      const long long rec = ((long long)(rows > 0 ? rows : 0) * p.ldb - n0) * 4ll;
      rsB = make_rsrc(p.B + ((unsigned long long)row0 * (unsigned long long)p.ldb + (unsigned long long)n0),
                      (unsigned long long)((live && rec > 0) ? rec : 0ll));
    }
    if (FETCH) {
#pragma unroll
      for (int j = 0; j < BNL; ++j) load_vec_buf<BVEC>(rsB, MODE == MODE_BWD_D ? b_off[j] : b_voff[j], rb[j]);
    }
  };
  auto issue = [&]() {
    if (a_on) {
#pragma unroll
      for (int j = 0; j < A_NL; ++j) load_vec_buf<AVEC>(rsA, a_off[j], ra[j]);
    }
#pragma unroll
    for (int j = 0; j < BNL; ++j) load_vec_buf<BVEC>(rsB, MODE == MODE_BWD_D ? b_off[j] : b_voff[j], rb[j]);
  };
  auto store_tiles = [&](int buf) {
    ATile::template store<Cfg::A_SWZ>(ra, As + buf * Cfg::A_ELEMS, Cfg::A_LD, tid);
    if constexpr (MODE == MODE_BWD_D) BTile::store(rb, Bs + buf * Cfg::B_ELEMS, Cfg::B_LD, tid);
    else BTile::template store<Cfg::B_SWZ>(rb, Bs + buf * Cfg::B_ELEMS, Cfg::B_LD, tid);
  };

#ifdef A3D_STAMPS
  unsigned long long s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, d01 = 0, d12 = 0, d23 = 0, d34 = 0, d45 = 0, tbeg = 0, tend = 0, rtbeg = 0, rtend = 0, tp1 = 0, tp2 = 0;
#endif
  auto k_loop = [&](auto uni_c) {
  // AHEAD: the next tile's addresses are computed one tile early, in the shadow of the MFMAs (forward / bwd-data: a
  // handful of scalar operations).  The bwd-filter gather reads its row table from LDS and keeps more lane state; there
  // the addresses are computed where they are used (measured: +4-7 % the other way round).
  constexpr bool AHEAD = MODE != MODE_BWD_F;
  A3D_STAMP(tp1);
  if (nkt > 0) {
    prepare(uni_c, std::true_type{}, kt_begin, 0, true);
    if (AHEAD) prepare(uni_c, std::false_type{}, kt_begin + 1, 1, nkt > 1);
    A3D_STAMP(tp2);
    store_tiles(0);
  }
  __syncthreads();

#ifdef A3D_STAMPS
  A3D_STAMP(tbeg);
  A3D_RTSTAMP(rtbeg);
#endif
  // one K tile; `cur` (which half of the double-buffered LDS tiles it reads) is a compile-time constant: the loop below is
  // unrolled by two, so every fragment read is `lane constant + immediate` instead of an address rebuilt per tile (20 of the
  // 31 vector instructions a wave issued per tile — and beside fp32 MFMAs none of them is hidden: DESIGN.md 3.1)
  auto tile_body = [&](const int it, auto cur_c) {
    constexpr int cur = decltype(cur_c)::value;
    const int kt = kt_begin + it;
    const bool more = it + 1 < nkt;
    A3D_STAMP(s0);
    // The staging instructions are few, but issued at priority 0 they queue behind the MFMAs of the three other waves of
    // this SIMD (stamps: 20-30 cycles per instruction); raised, they slip into the matrix pipe's 64-cycle shadows.
    if (AHEAD) issue();                          // tile kt+1 (a tile that does not exist: descriptors of 0 records)
    else prepare(uni_c, std::true_type{}, kt + 1, (it + 1) % 3, more);
    __builtin_amdgcn_sched_barrier(0);           // the loads stay here: a whole tile of MFMAs ahead of their first use
    A3D_STAMP(s1);
    // addresses of tile kt+2: no dependence on anything below, free to sink among the MFMAs
    if (AHEAD) prepare(uni_c, std::false_type{}, kt + 2, (it + 2) % 3, it + 2 < nkt);
    if (MODE == MODE_BWD_F) {
      // row table of tile kt+2 into the buffer tile kt-1 used (three buffers: tiles kt and kt+1 are still live)
      if (tid < Cfg::PIX) pixtab[((it + 2) % 3) * Cfg::PIX + tid] = row_entry((kt + 2) * BK + tid, image_of((kt + 2) * BK));
    }
    const float* Ac = As + cur * Cfg::A_ELEMS;
    const float* Bc = Bs + cur * Cfg::B_ELEMS;
    if (MODE == MODE_BWD_F && do_bias) {
#pragma unroll 8
      for (int k = 0; k < BK; ++k) bsum += Bc[k * Cfg::B_LD + (tid ^ (Cfg::B_SWZ ? 32 * ((k >> 2) & 1) : 0))];
    }
    // PIPE (8-wave 128-wide backward kernels): fragments double-buffered in registers — chunk u+1 is read from LDS while
    // chunk u's MFMAs issue, in the requested interleave of one MFMA and its share of the next chunk's ds_reads
    // (bwd-filter 128x128 187 -> 182 us, bwd-data 170 -> 163 us).  The forward kernels and the other tiles are 0-4 %
    // slower that way and keep the plain read-then-multiply form below.
    constexpr bool PIPE = A3D_PIPE_ALL ? (NWAVES == 8) : ((MODE != MODE_FWD) && NWAVES == 8 && BN == 128);
    if constexpr (PIPE) {
      f32x4 af[2][TM], bf[2][TN];
      auto read_frags = [&](int u, int buf) {
        const int kk = 8 * u + 4 * lh;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
          const int row = (wm * Cfg::WM + a * 32 + li) ^ (Cfg::A_SWZ ? 32 * lh : 0);      // (kk + j) >> 2 is odd exactly for lh = 1
          if (MODE == MODE_BWD_F) {
#pragma unroll
            for (int j = 0; j < 4; ++j) af[buf][a][j] = Ac[(kk + j) * Cfg::A_LD + row];
          } else {
            af[buf][a] = *reinterpret_cast<const f32x4*>(Ac + row * Cfg::A_LD + kk);
          }
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int col = (wn * Cfg::WN + b * 32 + li) ^ (Cfg::B_SWZ ? 32 * lh : 0);
          if (MODE == MODE_BWD_D) {
            bf[buf][b] = *reinterpret_cast<const f32x4*>(Bc + col * Cfg::B_LD + kk);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[buf][b][j] = Bc[(kk + j) * Cfg::B_LD + col];
          }
        }
      };
      read_frags(0, 0);
#pragma unroll
      for (int u = 0; u < BK / 8; ++u) {
        constexpr int LAST = BK / 8 - 1;
        if (u < LAST) read_frags(u + 1, (u + 1) & 1);
        if (u == LAST) {
          A3D_STAMP(s2);
          if (more) store_tiles(cur ^ 1);
          A3D_STAMP(s3);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[u & 1][a][j], bf[u & 1][b][j], acc[a][b], 0, 0, 0);
        if (u < LAST) {
          constexpr int NREADS = (MODE == MODE_BWD_F ? 4 * (TM + TN) : TM + TN);
          constexpr int NMFMA = 4 * TM * TN;
          constexpr int PER = (NREADS + NMFMA - 1) / NMFMA;
#pragma unroll
          for (int i = 0; i < NMFMA; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, PER, 0);    // then this many LDS reads of the next chunk
          }
        }
      }
    } else {
#pragma unroll
    for (int u = 0; u < BK / 8; ++u) {
      f32x4 af[TM], bf[TN];
      const int kk = 8 * u + 4 * lh;
#pragma unroll
      for (int a = 0; a < TM; ++a) {
        const int row = (wm * Cfg::WM + a * 32 + li) ^ (Cfg::A_SWZ ? 32 * lh : 0);
        if (MODE == MODE_BWD_F) {
#pragma unroll
          for (int j = 0; j < 4; ++j) af[a][j] = Ac[(kk + j) * Cfg::A_LD + row];
        } else {
          af[a] = *reinterpret_cast<const f32x4*>(Ac + row * Cfg::A_LD + kk);
        }
      }
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int col = (wn * Cfg::WN + b * 32 + li) ^ (Cfg::B_SWZ ? 32 * lh : 0);
        if (MODE == MODE_BWD_D) {
          bf[b] = *reinterpret_cast<const f32x4*>(Bc + col * Cfg::B_LD + kk);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) bf[b][j] = Bc[(kk + j) * Cfg::B_LD + col];
        }
      }
      // the next tile's global loads were issued before chunk 0; park them in the other LDS buffer ahead of the
      // last chunk's MFMAs so that only the barrier is left at the end of the tile
      if (u == BK / 8 - 1) {
        A3D_STAMP(s2);
        if (more) store_tiles(cur ^ 1);
        A3D_STAMP(s3);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[a][j], bf[b][j], acc[a][b], 0, 0, 0);
    }
    }
    A3D_STAMP(s4);
    __syncthreads();
    A3D_STAMP(s5);
#ifdef A3D_STAMPS
    d01 += s1 - s0; d12 += s2 - s1; d23 += s3 - s2; d34 += s4 - s3; d45 += s5 - s4;
#endif
  };
  for (int it = 0; it < nkt; it += 2) {
    tile_body(it, std::integral_constant<int, 0>{});
    if (it + 1 < nkt) tile_body(it + 1, std::integral_constant<int, 1>{});
  }
  };
  if constexpr (MODE == MODE_BWD_F) {        // the tap belongs to the lane there: one loop body
    k_loop(std::false_type{});
  } else {
    if (p.uni) k_loop(std::true_type{});
    else k_loop(std::false_type{});
  }
#ifdef A3D_STAMPS
  A3D_STAMP(tend);
  A3D_RTSTAMP(rtend);
  if (p.stamps && lane == 0) {
    unsigned long long* o = p.stamps + ((size_t)bid_in * NWAVES + wave) * 16;
    o[0] = d01; o[1] = d12; o[2] = d23; o[3] = d34; o[4] = d45; o[5] = tend - tbeg; o[6] = (unsigned long long)nkt; o[7] = tp1 - t_entry; o[15] = tp2 - tp1;
    o[8] = tbeg - t_entry; o[9] = t_entry; o[10] = tend; o[12] = rtend - rtbeg; o[13] = rtbeg; o[14] = rtend;
  }
#define A3D_STAMP_EXIT()                                                                                   \
  do {                                                                                                     \
    unsigned long long t_exit_;                                                                            \
    A3D_STAMP(t_exit_);                                                                                    \
    if (p.stamps && lane == 0) p.stamps[((size_t)bid_in * NWAVES + wave) * 16 + 11] = t_exit_;              \
  } while (0)
#else
#define A3D_STAMP_EXIT() do { } while (0)
#endif

  // ---- epilogue ----
  igemm_epilogue<MODE, BM, BN, TM, TN, Cfg::WM, Cfg::WN>(p, acc, split, bid, seg, kt_begin, kt_end, nk_total, do_bias, bsum, tid, wave, lane,
                                                          m0, n0, wm, wn);
  }   // shares of this block
  A3D_STAMP_EXIT();
}

// Adds the slabs of every tile that more than one block worked on, in block order (= ascending k: the same sum whatever
// the timing), and applies the epilogue the owning kernel would have applied.  A slab is a sequence of 1-KiB units, one per
// (wave, accumulator, register quad) of the GEMM kernel; a wave of this kernel owns one unit of one tile, so a tile is
// finished by BM*BN/256 independent waves (grid.y = groups of four units), each reading 16 bytes per lane per slab with
// four slabs in flight.
template <int MODE, int BM, int BN, int WAVES_M, int NWAVES>
__global__ __launch_bounds__(256) void igemm_fixup_kernel(const IgemmParams p, const uint32_t nblk) {
  constexpr int WAVES_N = NWAVES / WAVES_M, WM = BM / WAVES_M, WN = BN / WAVES_N, TM = WM / 32, TN = WN / 32;
  constexpr int UNITS = NWAVES * TM * TN * 4;
  constexpr int MAXC = 1024;                                // contributors of one tile: at most one per block (host: grid <= 1024)
  __shared__ uint32_t slots[MAXC];
  __shared__ uint32_t nslots;
  const int lane = threadIdx.x & 63;
  const int unit = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6);
  const uint32_t tile = blockIdx.x;
  const uint32_t nk = p.div_nk.d, tiles = (uint32_t)(p.tiles_m * p.tiles_n);
  // the tile's contributors in block order (= ascending k), as slab slots: first share of the block if that share starts
  // in this tile, else its last share.  One thread lists them; a launch with more blocks than iterations has blocks
  // without work in between.
  if (threadIdx.x == 0) {
    uint32_t n = 0;
    const uint32_t nx = p.streamk == 2 ? 8u : 1u;
    for (uint32_t x = 0; x < nx; ++x) {
      const SkSpace sp = sk_space(p.streamk, x * (nblk / nx), nblk, tiles, nk);
      if (sp.nk == 0) continue;
      const uint32_t t0 = tile * sp.nk, t1 = t0 + sp.nk - 1;
      uint32_t jf = (uint32_t)(((unsigned long long)t0 * sp.blocks) / sp.total), jl = (uint32_t)(((unsigned long long)t1 * sp.blocks) / sp.total);
      while (jf + 1 < sp.blocks && sk_first(jf + 1, sp.blocks, sp.total) <= t0) ++jf;
      while (jl + 1 < sp.blocks && sk_first(jl + 1, sp.blocks, sp.total) <= t1) ++jl;
      for (uint32_t j = jf; j <= jl; ++j) {
        const uint32_t f = sk_first(j, sp.blocks, sp.total);
        if (f == sk_first(j + 1, sp.blocks, sp.total)) continue;                      // idle block
        const uint32_t b = x * (nblk / nx) + j;
        if (n < MAXC) slots[n] = 2 * b + (f / sp.nk == tile ? 0u : 1u);
        ++n;
      }
    }
    nslots = n < MAXC ? n : MAXC;
  }
  __syncthreads();
  const uint32_t n = nslots;
  // one contributor that covered the whole K range wrote the tile itself (tile-major shares only)
  if (n <= 1 && p.streamk == 1) return;
  const int tile_m = (int)tile / p.tiles_n, tile_n = (int)tile - tile_m * p.tiles_n;
  if (unit < UNITS) {
    const size_t uoff = ((size_t)unit * 64 + lane) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    auto slab_of = [&](uint32_t i) -> f32x4 {
      return *reinterpret_cast<const f32x4*>(p.sk_ws + (size_t)slots[i] * (size_t)(BM * BN) + uoff);
    };
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) {          // eight slabs in flight, added in contributor order
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = slab_of(i + u);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    if (i + 4 <= n) {
      const f32x4 v0 = slab_of(i), v1 = slab_of(i + 1), v2 = slab_of(i + 2), v3 = slab_of(i + 3);
      s += v0; s += v1; s += v2; s += v3;
      i += 4;
    }
    for (; i < n; ++i) s += slab_of(i);
    // unit -> (wave, a, b, q) -> rows / column, as slab_store laid them out
    const int q = unit & 3, ab = unit >> 2, bb = ab % TN, a = (ab / TN) % TM, wave = ab / (TN * TM);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N, li = lane & 31, lh = lane >> 5;
    store_quad<MODE>(p, s, tile_m * BM + wm * WM + a * 32 + 8 * q + 4 * lh, tile_n * BN + wn * WN + bb * 32 + li, p.C,
                     p.ldc, false);
  }
  if (MODE == MODE_BWD_F && p.dbias != nullptr && tile_m == 0 && blockIdx.y == 0 && (int)threadIdx.x < BN) {
    static_assert(BN <= 256, "bias sums by the first 256 threads");
    float bsum = 0.f;
    for (uint32_t i = 0; i < n; ++i) bsum += p.sk_bias[(size_t)slots[i] * BN + threadIdx.x];
    if (tile_n * BN + (int)threadIdx.x < p.N) p.dbias[tile_n * BN + threadIdx.x] = bsum;
  }
}

template <int MODE, int BM, int BN, int WAVES_M, int NWAVES, int BKT, int AVEC, int BVEC>
__global__ __launch_bounds__(64 * NWAVES, (BKT == 16 ? 3 : 2) * NWAVES / 4) void igemm_kernel(const IgemmParams p) {
  igemm_body<MODE, BM, BN, WAVES_M, NWAVES, BKT, AVEC, BVEC>(p, gridDim.x, blockIdx.x);
}

// Up to four independent problems of one tile configuration in ONE launch (blockIdx.y selects the problem): the
// parity classes of a strided bwd-data are ~190-block GEMMs each, too small to fill the chip one after the other.
struct IgemmMulti {
  IgemmParams p[4];
};
template <int MODE, int BM, int BN, int WAVES_M, int NWAVES, int BKT, int AVEC, int BVEC>
__global__ __launch_bounds__(64 * NWAVES, (BKT == 16 ? 3 : 2) * NWAVES / 4) void igemm_multi_kernel(const IgemmMulti ps) {
  const IgemmParams& p = ps.p[blockIdx.y];
  const uint32_t nwg = (uint32_t)(p.tiles_m * p.tiles_n * p.splitk);
  if (blockIdx.x >= nwg) return;
  igemm_body<MODE, BM, BN, WAVES_M, NWAVES, BKT, AVEC, BVEC>(p, nwg, blockIdx.x);
}

// split-K slab reduction + the same epilogue
struct ReduceParams {
  const float* ws; float* C; const float* bias; const float* mask; const uint8_t* keep; float mask_scale;
  int M, N, ldc, splitk, act, mode, mask_act; size_t slab;
  int sub_step, sub_ph, sub_pw, outW, outHW; FastDiv div_phw, div_pw;     // BWD_D parity-class row remap
  int vec4;              // plain 16-byte sum (bwd-filter slabs)
  int c16;               // output (and BWD_D mask) tensors are bf16
  const float* dbias_ws; float* dbias_out;     // bwd-filter: the [splitk][N] BiasAddGrad slabs ride along, or null
  int dbias_splits;      // > 0: dbias_ws holds that many rows instead of splitk (the column-sum partials of colsum_bf16), added by
  unsigned dbias_block0; //      the blocks dbias_block0 .. of the grid, sixteen threads per column (reduce_wide)
  void* C2; int ld2, step2, off2, c2_16, cols2;      // second output: element (row, col < cols2) at (row * ld2 + col) * step2 + off2
  int c_cols;            // > 0: columns >= c_cols are not stored to C
  int row_rl, row_rlp;   // vec4 sums of a window-run filter gradient: row rr*row_rlp + q of the slabs is row rr*row_rl + q of
                         // C for q < row_rl and a pad row otherwise (not stored); row_rlp == 0: rows as they are
};
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ReduceParams p);

}  // namespace a3d
