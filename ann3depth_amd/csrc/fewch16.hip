// fewch16.hip — the few-channel layers' filter gradient (fewch.hip) on the bf16 matrix cores, for BASELINE config 5
// ("bf16 activations + bf16 weight copies, fp32 master and accumulate"): conv2d_0 and fine/first of src/models.py:211,241 at
// batch 64, where the gradient tensors are bf16 and the fp32 form of the kernel was the largest launch of the fine phase.
//
// v_mfma_f32_32x32x16_bf16 wants 8 consecutive k per lane for BOTH operands, and k is the PIXEL axis here: neither x (a tap's
// pixels are stride*C floats apart) nor dz (pixel-major, channel-minor) has that in memory.  Both tiles therefore sit in LDS
// pixel-major — as they are produced — and the fragments are read with the transposing LDS read (ds_read_b64_tr_b16), a
// segment of one output row (<= 80 pixels) at a time:
//   xs  [<= 6 rows][floats]  the input rows' share of the segment, float32, as 16-byte pieces (coalesced);
//   Ats [pixels][128 taps]   bf16: Ats[p][m] = x[oy*st + r_m][(p*st + s_m)*C + c_m], built from xs — a thread takes two
//                            neighbouring taps of one pixel: two LDS reads, one v_cvt_pk_bf16_f32, one 4-byte write;
//   dzs [pixels][N]          bf16: a bf16 dz tensor as it lies in memory, or — MaxPoolGrad + ReluGrad fused, as in fewch.hip —
//                            built from the pooled gradient, the argmax bytes and the pooled activation (8-byte writes).
// (Transposing while WRITING instead — [tap][pixel] / [channel][pixel] tiles, 16-byte fragment reads — put 12- to 32-way
// bank conflicts on the channel-major scatter: the LDS pipe was busy 7.5 k of a segment's 9.7 k cycles, 92 us for conv2d_0
// at B = 64.)  BiasAddGrad is tap M of the tile again (Ats[p][M] = 1).  The operands of segment s + 1 are fetched into
// registers behind the barrier that publishes segment s.  Partial tiles per pixel range go to fp32 slabs and fewch.hip's
// reduction adds them in split order.
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
  int XP, xl4;             // floats per staged input row in LDS, 16-byte pieces fetched per row and segment
  int LDD;                 // row stride of the [pixel][channel] tile
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ bf16x4 few16_read_tr(const __bf16* p) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  return __builtin_bit_cast(bf16x4, v);
}
// fragment of a [k][col] plane for the 32 columns col0 .. col0+31: lane l gets column col0 + (l & 31), k = kbase + 8*(l >> 5) + 0..7
__device__ __forceinline__ bf16x8 few16_frag_tr(const __bf16* plane, int ld, int col0, int kbase, int lane) {
  const int g = lane >> 4, l16 = lane & 15;
  const int q = l16 >> 2, pp = l16 & 3, h = g >> 1, cb = g & 1;
  const __bf16* a0 = plane + (kbase + 8 * h + q) * ld + col0 + 16 * cb + 4 * pp;
  const bf16x4 lo4 = few16_read_tr(a0), hi4 = few16_read_tr(a0 + 4 * ld);
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = lo4[e]; r[4 + e] = hi4[e]; }
  return r;
}

// A workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also waits for every outstanding global
// load (s_waitcnt vmcnt(0)): the next segment's operands, fetched into registers a segment ahead, would be waited for at the
// very next barrier and each segment would expose one memory latency (115 us for conv2d_0 at B = 64, half of all wave cycles
// waiting: profiles/r05_pmc_fewch16.txt).  Registers that a buffer load still owes are guarded by the compiler's own vmcnt
// wait at their first use.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int kNT = 256;             // (eight waves sharing the staging, halves of a segment's k-steps each, were slower: 108 vs 88 us)
constexpr int kKA = 19;              // (tap pair, pixel) items of the A tile per thread and segment (64 pairs x 74 pixels / 256)
constexpr int kKD = 4;               // 4-channel groups of the dz segment per thread (37 windows x 24 groups / 256)
constexpr int kKX = 6;               // 16-byte pieces of the input rows per thread and segment (6 rows x 228 pieces / 256)
constexpr int kRows16 = 6;           // input rows one 128-row group of M can touch

template <int TN, bool POOLED, bool VEC>
__global__ __launch_bounds__(kNT, 2) void fewch16_bwdf_kernel(const Few16Params p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 31, lh = lane >> 5;
  constexpr int LDA = 128 + 32;                              // row stride of the [pixel][tap] tile (cols + 32: the transposing read's rows 16 banks apart)
  const int LDD = p.LDD;                                     // (>= NP, 32 or 96 mod 128: see few16_ldd)
  __bf16* At = reinterpret_cast<__bf16*>(smem16);            // [PPk pixels][LDA]
  __bf16* dzT = At + p.PP * LDA;                             // [PPk pixels][LDD]
  float* xs = reinterpret_cast<float*>(dzT + p.PP * LDD);    // [kRows16][XP]: the input rows' segment, float32 as it lies in memory
  // every LDS access of the staging is UNCONDITIONAL: a slot a thread does not have reads index 0 and writes its own 16 bytes
  // of this scrap area.  Guarded accesses are one basic block each — the compiler cannot batch reads across them, and every
  // slot paid an LDS round trip of its own (387 scalar instructions per wave and segment were exec-mask branches).
  unsigned char* scrap = reinterpret_cast<unsigned char*>(xs + kRows16 * p.XP) + tid * 16;
  const int scrapA = (int)(scrap - reinterpret_cast<unsigned char*>(At)), scrapD = (int)(scrap - reinterpret_cast<unsigned char*>(dzT));
  const int scrapX = (int)((scrap - reinterpret_cast<unsigned char*>(xs)) >> 2);
  uint32_t bid = blockIdx.x;
  {                                                          // XCD-aware order: the m-groups of one pixel range share an L2
    const uint32_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int split = (int)bid / p.mgroups, mg = (int)bid - split * p.mgroups;
  const int SC = p.S * p.c, stC = p.stride * p.c;
  const int rlo = (mg * 128) / SC;
  // zero both tiles (pad pixels, pad taps, pad channels), then the bias row: ones
  for (int i = tid; i < p.PP * (LDA + LDD) / 2; i += kNT) reinterpret_cast<uint32_t*>(At)[i] = 0u;
  __syncthreads();
  {
    const int mb = p.M - mg * 128;                           // the bias tap's position in this m-group, if it has it
    if (mb >= 0 && mb < 128)                                 // (tile column of tap t: 2 * (t % 64) + t / 64, see below)
      for (int i = tid; i < p.PP; i += kNT) At[i * LDA + 2 * (mb & 63) + (mb >> 6)] = (__bf16)1.f;
  }
  // ---- per-thread constants of the staging: which pixel pairs of which taps, which channel groups of which windows
  // consecutive threads take consecutive tap PAIRS of one pixel (conflict-free reads of xs, conflict-free 4-byte writes of the
  // tile).  256 is a multiple of 64, so a thread keeps ONE tap pair for all its slots and walks the pixels in steps of four: no
  // per-slot tables (they were 95 registers and pushed the kernel into spilling).
  // A thread's pair is taps mp and mp + 64 of the group, stored side by side: tile column 2 * (t % 64) + t / 64 holds tap t (the
  // slabs are written back in tap order).  With neighbouring taps 2mp, 2mp + 1 per thread a wave read every other float of a
  // filter row — a 2-way bank conflict on each of the segment's 38 reads, half of the LDS pipe's busy cycles.
  const int mp = tid & 63, px0 = tid >> 6, m0 = mg * 128 + mp, m1 = m0 + 64;
  const bool ok0 = m0 < p.M, ok1 = m1 < p.M;
  const int r0 = ok0 ? m0 / SC : 0, r1 = ok1 ? m1 / SC : 0;
  const int base0 = ok0 ? (r0 - rlo) * p.XP + (m0 - r0 * SC) : 0;                 // float index into xs of pixel 0 (plus the segment's shift)
  const int base1 = ok1 ? (r1 - rlo) * p.XP + (m1 - r1 * SC) : base0;
  const float fill1 = (ok0 && m1 == p.M) ? 1.f : 0.f;                             // the pair's second tap is the bias tap (ones) or padding
  // The build below has NO per-slot validity test (nineteen lane masks were 38 scalar registers: the kernel spilled 90-170 of
  // them into vector lanes and read them back with one v_readlane each, 85 per segment, and every address went through a
  // select): a slot past the segment's last pixel is CLAMPED to that pixel and writes the same values to the same place as the
  // slot that owns it; a thread whose tap pair lies past M reads pixel addresses of tap 0 and writes all its slots to its own
  // scrap bytes (stride 0).
  const int rstride = stC * 4, rdelta = (base1 - base0) * 4;
  const int wbase = ok0 ? 4 * mp : scrapA, wstride = ok0 ? LDA * 2 : 0;
  int pxc[kKA];
#pragma unroll
  for (int k = 0; k < kKA; ++k) pxc[k] = min(px0 + 4 * k, p.seg - 1);
  // the input rows reach LDS as 16-byte pieces (coalesced), the taps are gathered from there: 4-byte gathers straight from
  // global memory kept the address unit busy for the whole segment (103 us for conv2d_0 at B = 64)
  const int nr = min(p.M - 1, mg * 128 + 127) / SC - rlo + 1;
  int xsdst[kKX];
  uint32_t xsrc[kKX];
#pragma unroll
  for (int i = 0; i < kKX; ++i) {
    const int e = tid + i * kNT, rr = e / p.xl4, q = e - rr * p.xl4;
    const bool ok = rr < nr;
    xsdst[i] = ok ? rr * p.XP + 4 * q : scrapX;
    xsrc[i] = ok ? (uint32_t)((rr * p.rowlen + 4 * q) * 4) : kOOB;
  }
  int ddst[kKD], ddst2[kKD];
  uint32_t doff[kKD], aoff[kKD], dlive[kKD];
  const int dtotal = p.hp * p.n4;
#pragma unroll
  for (int i = 0; i < kKD; ++i) {
    const int e = tid + i * kNT, px = e / p.n4, q = 4 * (e - px * p.n4);
    const bool ok = e < dtotal;
    ddst[i] = ok ? (2 * px * LDD + q) * 2 : scrapD;           // byte offset of dzs[2px][q]; the second pixel: + LDD
    ddst2[i] = ok ? ddst[i] + LDD * 2 : scrapD + 8;
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

  // ---- a segment's operands: fetch() issues every global load of it into registers — one segment ahead, behind the barrier
  //      that publishes the previous one, so the loads fly while At is built and the MFMAs run (without it every segment
  //      exposed one memory latency: 112 us for conv2d_0 at B = 64) — commit() writes them to LDS
  u32x4 xv[kKX];
  bf16x4 g0[kKD], g1[kKD];
  uint32_t argv[kKD];
  auto where = [&](int stg, int& img, int& oy, int& seg0) {      // (32-bit: a 64-bit division here cost more than the segment's MFMAs)
    const int row = stg / p.nseg, segi = stg - row * p.nseg;
    img = row / p.rows_per_img;
    oy = row - img * p.rows_per_img;
    seg0 = segi * p.seg;
  };
  auto fetch = [&](int stg) {
    int img, oy, seg0;
    where(stg, img, oy, seg0);
    const int start4 = (seg0 * stC) & ~3;              // the segment's first float, rounded down to a 16-byte piece
    const uint32_t xbase = (uint32_t)((((size_t)img * p.h + (size_t)oy * p.stride + rlo) * p.rowlen + (size_t)start4) * 4);
#pragma unroll
    for (int i = 0; i < kKX; ++i) xv[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)(xbase + xsrc[i]), 0, 0);     // (a slot without a piece: 2^31 past a base below 2^31, outside every buffer)
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
        g1[i] = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(rd, (int)(dbase + doff[i] + (uint32_t)p.ldz * 2u), 0, 0));
      }
    }
  };
  // 16-bit lanes of a dword: all ones where ...
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  auto positive = [](uint32_t two_bf16) -> uint32_t {        // ... the bf16 is > 0 (+inf yes; NaN, -0, 0 no): u - 1 < 0x7f80 unsigned
    const u16x2 u = __builtin_bit_cast(u16x2, two_bf16);
    const u16x2 t = __builtin_elementwise_min((u16x2)(u - (u16x2){1, 1}), (u16x2){0x7f80, 0x7f80});
    return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, (u16x2)(t - (u16x2){0x7f80, 0x7f80})) >> (s16x2){15, 15}));
  };
  auto is_zero = [](uint32_t two_u16) -> uint32_t {          // ... the 16-bit value is 0 (values below 2^15)
    return __builtin_bit_cast(uint32_t, (s16x2)((__builtin_bit_cast(s16x2, two_u16) - (s16x2){1, 1}) >> (s16x2){15, 15}));
  };
  // per-slot channel masks of a group that straddles N (N % 4 != 0): stage-invariant
  uint32_t dm0[kKD], dm1[kKD];
#pragma unroll
  for (int i = 0; i < kKD; ++i) {
    dm0[i] = dlive[i] >= 2 ? 0xffffffffu : dlive[i] == 1 ? 0xffffu : 0u;
    dm1[i] = dlive[i] >= 4 ? 0xffffffffu : dlive[i] == 3 ? 0xffffu : 0u;
  }
  auto commit = [&](int stg) {
    int img, oy, seg0;
    where(stg, img, oy, seg0);
#pragma unroll
    for (int i = 0; i < kKX; ++i)
      *reinterpret_cast<u32x4*>(xs + xsdst[i]) = xv[i];
    // MaxPoolGrad + ReluGrad on packed 16-bit lanes, no lane masks: window px hands its gradient to position argmax, if the
    // maximum was > 0.  This output row holds positions `want` (left pixel of the pair) and `want + 1` (right pixel).
    const uint32_t wantx = (uint32_t)(oy & 1) * 0x02020202u;
#pragma unroll
    for (int i = 0; i < kKD; ++i) {
      typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
      u32x2 g = __builtin_bit_cast(u32x2, g0[i]);
      u32x2 lo, hi;
      if constexpr (POOLED) {
        if (p.pooled) {
          const u32x2 a = __builtin_bit_cast(u32x2, g1[i]);
          g[0] &= positive(a[0]); g[1] &= positive(a[1]);
        }
        g[0] &= dm0[i]; g[1] &= dm1[i];                            // (channels past N of the last group)
        const uint32_t t = argv[i] ^ wantx;                        // byte e: 0 = this row's left pixel, 1 = its right pixel
        const uint32_t t01 = __builtin_amdgcn_perm(0u, t, 0x0c010c00u), t23 = __builtin_amdgcn_perm(0u, t, 0x0c030c02u);   // bytes -> 16-bit lanes
        lo[0] = g[0] & is_zero(t01); lo[1] = g[1] & is_zero(t23);
        hi[0] = g[0] & is_zero(t01 ^ 0x00010001u); hi[1] = g[1] & is_zero(t23 ^ 0x00010001u);
      } else {
        u32x2 h = __builtin_bit_cast(u32x2, g1[i]);
        g[0] &= dm0[i]; g[1] &= dm1[i]; h[0] &= dm0[i]; h[1] &= dm1[i];
        lo = g; hi = h;
      }
      *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(dzT) + ddst[i]) = lo;
      *reinterpret_cast<u32x2*>(reinterpret_cast<unsigned char*>(dzT) + ddst2[i]) = hi;
    }
  };

  const int stages_total = p.rows_total * p.nseg;
  const int st_lo = (int)((long)split * stages_total / p.splits), st_hi = (int)((long)(split + 1) * stages_total / p.splits);
  if (st_lo < st_hi) fetch(st_lo);
  for (int stg = st_lo; stg < st_hi; ++stg) {
    int img, oy, seg0;
    where(stg, img, oy, seg0);
    const int shift = seg0 * stC - ((seg0 * stC) & ~3);
    lds_barrier();                                  // the previous segment's MFMAs (and its At build) have read their operands
    commit(stg);
    lds_barrier();
    if (stg + 1 < st_hi) fetch(stg + 1);
    // ---- the [pixel][tap] tile from the staged rows: two neighbouring taps of one pixel per thread and slot.  All reads
    //      first, then all writes: interleaved, every write had to precede the next read (both are LDS: they may alias as far
    //      as the compiler knows) and the build was nineteen serial LDS round trips.
    // (five slots at a time: reads batched, their lane masks not kept alive across all nineteen)
    const int rbs = (base0 + shift) * 4;
#pragma unroll
    for (int c = 0; c < kKA; c += 5) {
      float t0[5], t1[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        if (c + i < kKA) {
          const unsigned char* a = reinterpret_cast<const unsigned char*>(xs) + (pxc[c + i] * rstride + rbs);
          t0[i] = *reinterpret_cast<const float*>(a);
          t1[i] = *reinterpret_cast<const float*>(a + rdelta);
        }
      }
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        if (c + i < kKA) {
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          const f32x2 v = {t0[i], ok1 ? t1[i] : fill1};
          *reinterpret_cast<bf16x2*>(reinterpret_cast<unsigned char*>(At) + (pxc[c + i] * wstride + wbase)) = __builtin_convertvector(v, bf16x2);
        }
      }
    }
    lds_barrier();
    // ---- 16 pixels per MFMA: fragments through the transposing LDS read (two 8-byte reads each), the next k-step's
    //      fragments read while this one's MFMAs run
    bf16x8 fa[2], fb[2][TN];
    auto frags = [&](int u, int buf) {
      fa[buf] = few16_frag_tr(At, LDA, wv * 32, 16 * u, lane);
#pragma unroll
      for (int t = 0; t < TN; ++t) fb[buf][t] = few16_frag_tr(dzT, LDD, t * 32, 16 * u, lane);
    };
    auto mul = [&](int buf) {
#pragma unroll
      for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[buf], fb[buf][t], acc[t], 0, 0, 0);
    };
    const int u_lo = 0, u_hi = p.ksteps;
    if (u_lo < u_hi) frags(u_lo, 0);
    int u = u_lo;
    for (; u + 2 <= u_hi; u += 2) {
      frags(u + 1, 1);
      mul(0);
      if (u + 2 < u_hi) frags(u + 2, 0);
      mul(1);
    }
    if (u < u_hi) mul(0);
  }
  float* slab = p.slabs + ((size_t)split * p.Mp + mg * 128) * p.NP;
#pragma unroll
  for (int t = 0; t < TN; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = wv * 32 + 8 * g + 4 * lh + i, tap = (col >> 1) + 64 * (col & 1);      // tile column -> tap of the group
        slab[(size_t)tap * p.NP + t * 32 + li] = acc[t][4 * g + i];
      }
}

// ---- host side ----
// Row stride (bf16 elements) of the [pixel][channel] tile: the transposing read takes four consecutive rows per 16 lanes, 64 bytes
// of each — conflict-free when the rows start 16 banks apart, i.e. a stride of 64 or 192 bytes modulo 256 (NP + 32 = 128
// elements put all four rows of conv2d_0's 96-channel tile on the same banks).
static int few16_ldd(int NP) {
  int ld = NP;
  while (ld % 128 != 32 && ld % 128 != 96) ld += 32;
  return ld;
}
struct Few16Shape {
  int M, Mp, mgroups, TN, NP, rows_per_img, rows_total, splits, seg, nseg, PP, ksteps, hp, n4, XP, xl4;
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
  s.PP = s.ksteps * 16;                               // pixels of a segment, padded to whole MFMA k-steps
  s.splits = std::max(1, std::min(s.rows_total * s.nseg / 4, tune_int("A3D_FEWCH_BLOCKS", 512) / s.mgroups));
  // the input rows' share of a segment: (seg - 1) * st*C + S*C floats after a start rounded down to a 16-byte piece
  const int need = 3 + (s.seg - 1) * d->stride * d->c + d->s * d->c;
  s.xl4 = std::min((need + 3) / 4, (d->w * d->c) / 4);
  s.XP = s.xl4 * 4 + 4;
  s.lds = (size_t)s.PP * (160 + few16_ldd(s.NP)) * 2 + (size_t)kRows16 * s.XP * 4 + kNT * 16;
  s.ok = 64 * s.seg <= kKA * kNT && s.hp * s.n4 <= kKD * kNT && kRows16 * s.xl4 <= kKX * kNT && s.lds <= 78 * 1024 &&
         (d->w * d->c) % 4 == 0;
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
  // a little above 64 KiB of dynamic LDS: the attribute is per device and cheap, set on every call
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fewch16_bwdf_kernel<TN, POOLED, VEC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            80 * 1024);
  hipLaunchKernelGGL((fewch16_bwdf_kernel<TN, POOLED, VEC>), dim3(blocks), dim3(kNT), lds, st, p);
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
  p.XP = s.XP; p.xl4 = s.xl4; p.LDD = few16_ldd(s.NP);
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
