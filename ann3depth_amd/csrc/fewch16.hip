// fewch16.hip — the few-channel layers' filter gradient (fewch.hip) on the bf16 matrix cores, for BASELINE config 5
// ("bf16 activations + bf16 weight copies, fp32 master and accumulate"): conv2d_0 and fine/first of src/models.py:211,241 at
// batch 64, where the gradient tensors are bf16 and the fp32 form of the kernel was the largest launch of the fine phase.
//
// v_mfma_f32_32x32x16_bf16 wants 8 consecutive k per lane for BOTH operands, and k is the PIXEL axis here: neither x (a tap's
// pixels are stride*C floats apart) nor dz (pixel-major, channel-minor) has that in memory.  So both are TRANSPOSED on their
// way into LDS, a segment of one output row (<= 80 pixels) at a time:
//   At [128 taps][pixels]  bf16: At[m][p] = x[oy*st + r_m][(p*st + s_m)*C + c_m], gathered with 4-byte loads (L1-resident rows),
//                          two pixels per thread -> one v_cvt_pk_bf16_f32 and one 4-byte LDS write;
//   dzT[N][pixels]         bf16: from a bf16 dz tensor, or — MaxPoolGrad + ReluGrad fused, as in fewch.hip — from the pooled
//                          gradient, the argmax bytes and the pooled activation (two pixels of a window row per 4-byte write).
// A wave then reads its A fragment and TN B fragments with one 16-byte LDS read each per 16 pixels.  BiasAddGrad is row M of
// the tile again (At[M][p] = 1).  The MFMAs are a sixth of a segment's time: the kernel is bound by the staging (≈65 gathered
// loads and as many LDS writes per thread and segment), which is what makes it 4-5x the fp32 form, not 16x.
// Partial tiles per pixel range go to fp32 slabs and fewch.hip's reduction adds them in split order.
#include <algorithm>

#include "a3d_internal.h"
#include "igemm.h"

namespace a3d {

struct Few16Params {
  const float* x;          // [n, h, w, c] float32, pixels densely packed
  const void* dz;          // bf16: [n, ho, wo, ldz], or (pooled) the pooled gradient [n, ho/2, wo/2, ldz]
  const void* pooled;      // pooled source: the pooled activation (ReluGrad: > 0), same layout; null = no mask
  const uint8_t* argmax;   // pooled source: [n, ho/2, wo/2, ld_arg]
  float* slabs;            // [splits][Mp][NP]
  int n, h, w, c, R, S, stride, ho, wo, N, M, Mp, NP;
  int ldz, ld_arg;
  int rows_per_img, rows_total, splits, mgroups;
  int rowlen, seg, nseg, PP, ksteps, hp, n4;
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kKA = 19;              // pixel pairs of the A tile per thread and segment (128 taps x 37 pairs / 256)
constexpr int kKD = 4;               // 4-channel groups of the dz segment per thread (37 windows x 24 groups / 256)

template <int TN, bool POOLED, bool VEC>
__global__ __launch_bounds__(256, 2) void fewch16_bwdf_kernel(const Few16Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
  __bf16* At = reinterpret_cast<__bf16*>(smem16);            // [128][PP]
  __bf16* dzT = At + 128 * p.PP;                             // [NP][PP]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  uint32_t bid = blockIdx.x;
  {                                                          // XCD-aware order: the m-groups of one pixel range share an L2
    const uint32_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = (int)bid / p.mgroups, mg = (int)bid - split * p.mgroups;
  const int SC = p.S * p.c, stC = p.stride * p.c;
  const int rlo = (mg * 128) / SC;
  // zero both tiles (pad pixels, pad taps, pad channels), then the bias row: ones
  for (int i = tid; i < (128 + p.NP) * p.PP / 2; i += 256) reinterpret_cast<uint32_t*>(At)[i] = 0u;
  __syncthreads();
  {
    const int mb = p.M - mg * 128;                           // the bias row's position in this m-group, if it has it
    if (mb >= 0 && mb < 128)
      for (int i = tid; i < p.PP; i += 256) At[mb * p.PP + i] = (__bf16)1.f;
  }
  // ---- per-thread constants of the staging: which pixel pairs of which taps, which channel groups of which windows
  int adst[kKA];
  uint32_t xoff[kKA];
#pragma unroll
  for (int i = 0; i < kKA; ++i) {
    const int e = tid + i * 256, ml = e / p.hp, pp = e - ml * p.hp, m = mg * 128 + ml;
    const bool ok = ml < 128 && m < p.M;
    const int r = ok ? m / SC : 0, j = ok ? m - r * SC : 0;
    adst[i] = ok ? (ml * p.PP + 2 * pp) * 2 : -1;
    xoff[i] = ok ? (uint32_t)(((r - rlo) * p.rowlen + j + 2 * pp * stC) * 4) : kOOB;
  }
  int ddst[kKD];
  uint32_t doff[kKD], aoff[kKD], dlive[kKD];
  const int dtotal = p.hp * p.n4;
#pragma unroll
  for (int i = 0; i < kKD; ++i) {
    const int e = tid + i * 256, px = e / p.n4, q = 4 * (e - px * p.n4);
    const bool ok = e < dtotal;
    ddst[i] = ok ? (q * p.PP + 2 * px) * 2 : -1;
    // plain source: the pixel PAIR (2px, 2px+1) of the segment; pooled: window px of the segment's half
    doff[i] = ok ? (uint32_t)(((POOLED ? px : 2 * px) * p.ldz + q) * 2) : kOOB;
    aoff[i] = ok ? (uint32_t)(px * p.ld_arg + q) : kOOB;
    dlive[i] = (uint32_t)max(0, min(4, p.N - q));
  }
  const int pw = p.wo >> 1, ph = p.ho >> 1;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x, (unsigned long long)p.n * p.h * p.rowlen * 4ull);
  const unsigned long long dz_elems = POOLED ? (unsigned long long)p.n * ph * pw * p.ldz : (unsigned long long)p.n * p.ho * p.wo * p.ldz;
  const __amdgpu_buffer_rsrc_t rd = make_rsrc(static_cast<const float*>(p.dz), dz_elems * 2ull);
  const __amdgpu_buffer_rsrc_t rp = make_rsrc(static_cast<const float*>(p.pooled ? p.pooled : p.dz), dz_elems * 2ull);
  const __amdgpu_buffer_rsrc_t ra = make_rsrc(reinterpret_cast<const float*>(p.argmax),
                                              POOLED ? (unsigned long long)p.n * ph * pw * p.ld_arg : 0ull);
  f32x16 acc[TN];
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[t][v] = 0.f;

  const long stages_total = (long)p.rows_total * p.nseg;
  const long st_lo = (long)split * stages_total / p.splits, st_hi = (long)(split + 1) * stages_total / p.splits;
  for (long stg = st_lo; stg < st_hi; ++stg) {
    const int row = (int)(stg / p.nseg), segi = (int)(stg - (long)row * p.nseg);
    const int img = row / p.rows_per_img, oy = row - img * p.rows_per_img;
    const int seg0 = segi * p.seg;
    // ---- every load of the segment in flight, then convert and write
    const uint32_t xbase = (uint32_t)((((size_t)img * p.h + (size_t)oy * p.stride + rlo) * p.rowlen + (size_t)seg0 * stC) * 4);
    float a0[kKA], a1[kKA];
#pragma unroll
    for (int i = 0; i < kKA; ++i) {
      a0[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, (int)(xbase + xoff[i]), 0, 0));
      a1[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, (int)(xoff[i] == kOOB ? kOOB : xbase + xoff[i] + (uint32_t)stC * 4u), 0, 0));
    }
    bf16x4 g0[kKD], g1[kKD];
    uint32_t argv[kKD];
    if constexpr (POOLED) {
      const size_t prow = ((size_t)img * ph + (oy >> 1)) * pw + (seg0 >> 1);
      const uint32_t dbase = (uint32_t)(prow * p.ldz * 2), abase = (uint32_t)(prow * p.ld_arg);
#pragma unroll
      for (int i = 0; i < kKD; ++i) {
        g0[i] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rd, (int)(dbase + doff[i]), 0, 0));
        if (p.pooled) g1[i] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rp, (int)(dbase + doff[i]), 0, 0));
        if constexpr (VEC) {
          argv[i] = __builtin_amdgcn_raw_buffer_load_b32(ra, (int)(abase + aoff[i]), 0, 0);
        } else {
          uint32_t w = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            w |= (uint32_t)(uint8_t)__builtin_amdgcn_raw_buffer_load_b8(ra, (uint32_t)e < dlive[i] ? (int)(abase + aoff[i] + e) : (int)kOOB, 0, 0) << (8 * e);
          argv[i] = w;
        }
      }
    } else {
      const uint32_t dbase = (uint32_t)(((((size_t)img * p.ho + oy) * p.wo + seg0) * p.ldz) * 2);
#pragma unroll
      for (int i = 0; i < kKD; ++i) {
        g0[i] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rd, (int)(dbase + doff[i]), 0, 0));
        g1[i] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rd, (int)(doff[i] == kOOB ? kOOB : dbase + doff[i] + (uint32_t)p.ldz * 2u), 0, 0));
      }
    }
    __syncthreads();                                  // the previous segment's MFMAs have read their operands
#pragma unroll
    for (int i = 0; i < kKA; ++i)
      if (adst[i] >= 0)
        *reinterpret_cast<bf16x2*>(reinterpret_cast<unsigned char*>(At) + adst[i]) = bf16x2{(__bf16)a0[i], (__bf16)a1[i]};
#pragma unroll
    for (int i = 0; i < kKD; ++i)
      if (ddst[i] >= 0) {
        unsigned char* dst = reinterpret_cast<unsigned char*>(dzT) + ddst[i];
        if constexpr (POOLED) {
          // MaxPoolGrad + ReluGrad: window px hands its gradient to position argmax, if the maximum was > 0
          const uint32_t want = (uint32_t)(oy & 1) * 2u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool live = (uint32_t)e < dlive[i] && (!p.pooled || (float)g1[i][e] > 0.f);
            const __bf16 g = live ? g0[i][e] : (__bf16)0.f;
            const uint32_t a = (argv[i] >> (8 * e)) & 0xffu;
            *reinterpret_cast<bf16x2*>(dst + (size_t)e * p.PP * 2) = bf16x2{a == want ? g : (__bf16)0.f, a == want + 1u ? g : (__bf16)0.f};
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool live = (uint32_t)e < dlive[i];
            *reinterpret_cast<bf16x2*>(dst + (size_t)e * p.PP * 2) = bf16x2{live ? g0[i][e] : (__bf16)0.f, live ? g1[i][e] : (__bf16)0.f};
          }
        }
      }
    __syncthreads();
    // ---- 16 pixels per MFMA: one 16-byte LDS read per fragment
    const __bf16* ap = At + (wv * 32 + li) * p.PP + 8 * lh;
    const __bf16* bp = dzT + li * p.PP + 8 * lh;
    for (int u = 0; u < p.ksteps; ++u) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + 16 * u);
#pragma unroll
      for (int t = 0; t < TN; ++t) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(bp + t * 32 * p.PP + 16 * u);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
      }
    }
  }
  float* slab = p.slabs + ((size_t)split * p.Mp + mg * 128 + wv * 32) * p.NP;
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) slab[(size_t)(8 * g + 4 * lh + i) * p.NP + t * 32 + li] = acc[t][4 * g + i];
}

// ---- host side ----
struct Few16Shape {
  int M, Mp, mgroups, TN, NP, rows_per_img, rows_total, splits, seg, nseg, PP, ksteps, hp, n4;
  size_t lds;
  bool ok;
};

static Few16Shape few16_shape(const a3d_conv_desc* d, bool pooled) {
  Few16Shape s{};
  s.M = d->r * d->s * d->c;
  s.mgroups = (s.M + 1 + 127) / 128;
  s.Mp = s.mgroups * 128;
  s.TN = (d->k + 31) / 32;
  s.NP = s.TN * 32;
  s.rows_per_img = pooled ? (d->ho / 2) * 2 : d->ho;
  s.rows_total = d->n * s.rows_per_img;
  // segments: equal, even, at most 80 pixels (the staging registers of a thread: kKA pixel pairs, kKD channel groups)
  s.ok = false;
  for (int nseg = (d->wo + 79) / 80; nseg <= 8 && nseg >= 1; ++nseg)
    if (d->wo % nseg == 0 && (d->wo / nseg) % 2 == 0) { s.nseg = nseg; s.seg = d->wo / nseg; s.ok = true; break; }
  if (!s.ok) return s;
  s.hp = s.seg / 2;
  s.n4 = (d->k + 3) / 4;
  s.ksteps = (s.seg + 15) / 16;
  s.PP = s.ksteps * 16 + 8;
  s.splits = std::max(1, std::min(s.rows_total * s.nseg / 4, tune_int("A3D_FEWCH_BLOCKS", 512) / s.mgroups));
  s.lds = (size_t)(128 + s.NP) * s.PP * 2;
  s.ok = 128 * s.hp <= kKA * 256 && s.hp * s.n4 <= kKD * 256 && s.lds <= 64 * 1024;
  return s;
}

// bf16 arithmetic on a float32 image and bf16 gradient tensors (config 5)
bool fewch16_bwdf_applicable(const a3d_conv_desc* d, bool pooled) {
  if (d->precision != A3D_PREC_BF16 || d->c > 4 || d->pad_t || d->pad_l || d->ldx != d->c) return false;
  if (d->k < 33 || d->k > 96) return false;
  if (d->s * d->c < 27) return false;
  if ((d->ho - 1) * d->stride + d->r > d->h || (d->wo - 1) * d->stride + d->s > d->w) return false;
  if (pooled && (d->ho < 2 || d->wo < 2)) return false;
  if ((double)d->n * d->h * d->w * d->c * 4.0 >= 2147483647.0) return false;
  return few16_shape(d, pooled).ok;
}

size_t fewch16_bwdf_ws_bytes(const a3d_conv_desc* d, bool pooled) {
  const Few16Shape s = few16_shape(d, pooled);
  return (size_t)s.splits * s.Mp * s.NP * 4 + 16;
}

template <int TN, bool POOLED, bool VEC>
static void few16_launch(const Few16Params& p, int blocks, size_t lds, hipStream_t st) {
  hipLaunchKernelGGL((fewch16_bwdf_kernel<TN, POOLED, VEC>), dim3(blocks), dim3(256), lds, st, p);
}

int fewch16_bwd_filter(const a3d_conv_desc* d, const float* x, bool pooled, const void* dz, int ldz, const void* pooled_act,
                       const uint8_t* argmax, int ld_arg, float* dw, float* db, void* ws, hipStream_t st) {
  const Few16Shape s = few16_shape(d, pooled);
  Few16Params p{};
  p.x = x; p.dz = dz; p.pooled = pooled_act; p.argmax = argmax;
  p.slabs = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 15) & ~(uintptr_t)15);
  p.n = d->n; p.h = d->h; p.w = d->w; p.c = d->c; p.R = d->r; p.S = d->s; p.stride = d->stride; p.ho = d->ho; p.wo = d->wo;
  p.N = d->k; p.M = s.M; p.Mp = s.Mp; p.NP = s.NP; p.ldz = ldz; p.ld_arg = ld_arg;
  p.rows_per_img = s.rows_per_img; p.rows_total = s.rows_total; p.splits = s.splits; p.mgroups = s.mgroups;
  p.rowlen = d->w * d->c; p.seg = s.seg; p.nseg = s.nseg; p.PP = s.PP; p.ksteps = s.ksteps; p.hp = s.hp; p.n4 = s.n4;
  const bool vec = pooled && d->k % 4 == 0 && ld_arg % 4 == 0 && (reinterpret_cast<uintptr_t>(argmax) & 3) == 0;
  const int blocks = s.splits * s.mgroups;
  clear_stale_error();
  if (s.TN == 3) {
    if (!pooled) few16_launch<3, false, true>(p, blocks, s.lds, st);
    else if (vec) few16_launch<3, true, true>(p, blocks, s.lds, st);
    else few16_launch<3, true, false>(p, blocks, s.lds, st);
  } else {
    if (!pooled) few16_launch<2, false, true>(p, blocks, s.lds, st);
    else if (vec) few16_launch<2, true, true>(p, blocks, s.lds, st);
    else few16_launch<2, true, false>(p, blocks, s.lds, st);
  }
  int rc = check_launch("fewch16_bwd_filter");
  if (rc != A3D_OK) return rc;
  return fewch_reduce_launch(p.slabs, s.splits, s.Mp, s.NP, s.M, d->k, dw, db, st);
}

}  // namespace a3d
