// stencil1.hip — direct kernels for convolutions with ONE output channel (fine/third, src/models.py:250-251:
// 5x5 SAME, 64 -> 1).  With N = 1 the implicit GEMM wastes 31/32 of every MFMA tile; the layer is a dot-product
// stencil and HBM/L2-bound, so it runs on the vector ALUs instead: lane = input channel (Cin <= 64), one wave slides
// over P consecutive output pixels of a row and reuses each loaded input column for up to S taps.
//   forward : y[p]       = b + sum_lanes sum_{r,s} x[p+(r,s)][lane] * w[r][s][lane]      (wave-shuffle reduction)
//   filter  : dw[r][s][lane] = sum_p x[p+(r,s)][lane] * dz[p] ;  db = sum_p dz[p]        (per-block slabs, then reduce)
#include <algorithm>

#include "a3d_internal.h"
#include "igemm.h"

namespace a3d {

static constexpr int P = 8;   // outputs per wave iteration

struct Stencil1Params {
  const float* x; const float* w; const float* bias; const float* dz; float* y; float* slabs;
  int n, h, w_in, c, ldx, ho, wo, pad_t, pad_l, ldy;
  int strips_per_row, total_strips, act;
};

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// One wave computes a tile of RT output rows x P output columns: every loaded input column (64 channels of one pixel)
// serves up to KS taps along the row AND up to RT of the KS taps down the column, so the tile reads
// (RT+KS-1) x (P+KS-1) pixels for RT x P outputs (3x per output at RT = 4 instead of 7.5x with single rows).
template <int KS>
__global__ __launch_bounds__(256) void stencil1_fwd_kernel(const Stencil1Params p) {
  constexpr int RT = 4;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  const bool live = lane < p.c;
  float wreg[KS * KS];
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) wreg[t] = live ? p.w[t * p.c + lane] : 0.f;
  const float b = p.bias ? p.bias[0] : 0.f;
  const int row_groups = (p.ho + RT - 1) / RT;
  const int total = p.n * row_groups * p.strips_per_row;
  for (int tile = wave; tile < total; tile += nwaves) {
    const int rg = tile / p.strips_per_row;               // (n, row group)
    const int q0 = (tile - rg * p.strips_per_row) * P;
    const int img = rg / row_groups, oy0 = (rg - img * row_groups) * RT;
    float acc[RT][P];
#pragma unroll
    for (int j = 0; j < RT; ++j)
#pragma unroll
      for (int i = 0; i < P; ++i) acc[j][i] = 0.f;
#pragma unroll
    for (int rr = 0; rr < RT + KS - 1; ++rr) {            // input row oy0 + rr - pad_t feeds output row j with tap r = rr - j
      const int iy = oy0 + rr - p.pad_t;
      if ((unsigned)iy >= (unsigned)p.h) continue;        // wave-uniform
      const float* xrow = p.x + ((size_t)img * p.h + iy) * p.w_in * p.ldx + lane;
#pragma unroll
      for (int col = 0; col < P + KS - 1; ++col) {
        const int ix = q0 + col - p.pad_l;
        float v = 0.f;
        if (live && (unsigned)ix < (unsigned)p.w_in) v = xrow[(size_t)ix * p.ldx];
#pragma unroll
        for (int j = 0; j < RT; ++j) {
          const int r = rr - j;
          if (r < 0 || r >= KS) continue;
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const int o = col - s;                         // output column within the strip
            if (o >= 0 && o < P) acc[j][o] += v * wreg[r * KS + s];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < RT; ++j) {
#pragma unroll
      for (int i = 0; i < P; ++i) acc[j][i] = wave_sum64(acc[j][i]);
      const int oy = oy0 + j;
      if (oy < p.ho && lane < P && q0 + lane < p.wo) {
        float out = acc[j][0];
#pragma unroll
        for (int i = 1; i < P; ++i) out = lane == i ? acc[j][i] : out;
        out += b;
        if (p.act == EPI_RELU) out = fmaxf(out, 0.f);
        else if (p.act == EPI_SIGMOID) out = 1.f / (1.f + expf(-out));
        p.y[(((size_t)img * p.ho + oy) * p.wo + q0 + lane) * p.ldy] = out;
      }
    }
  }
}

// slabs[block][KS*KS*64 + 1]: per-block partial dw (lane-major inside a tap) and partial db
template <int KS>
__global__ __launch_bounds__(256) void stencil1_bwdf_kernel(const Stencil1Params p) {
  __shared__ float red[4][KS * KS * 64 + 1];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv;
  const int nwaves = gridDim.x * 4;
  const bool live = lane < p.c;
  float acc[KS * KS];
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) acc[t] = 0.f;
  float dbsum = 0.f;
  for (int strip = wave; strip < p.total_strips; strip += nwaves) {
    const int row = strip / p.strips_per_row;
    const int q0 = (strip - row * p.strips_per_row) * P;
    const int img = row / p.ho, oy = row - img * p.ho;
    float g[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
      g[i] = (q0 + i < p.wo) ? p.dz[((size_t)row * p.wo + q0 + i) * p.ldy] : 0.f;     // wave-uniform broadcast loads
      dbsum += g[i];
    }
#pragma unroll
    for (int r = 0; r < KS; ++r) {
      const int iy = oy + r - p.pad_t;
      if ((unsigned)iy >= (unsigned)p.h) continue;
      const float* xrow = p.x + ((size_t)img * p.h + iy) * p.w_in * p.ldx + lane;
#pragma unroll
      for (int col = 0; col < P + KS - 1; ++col) {
        const int ix = q0 + col - p.pad_l;
        float v = 0.f;
        if (live && (unsigned)ix < (unsigned)p.w_in) v = xrow[(size_t)ix * p.ldx];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int o = col - s;
          if (o >= 0 && o < P) acc[r * KS + s] += v * g[o];
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) red[wv][t * 64 + lane] = acc[t];
  if (lane == 0) red[wv][KS * KS * 64] = dbsum;
  __syncthreads();
  float* slab = p.slabs + (size_t)blockIdx.x * (KS * KS * 64 + 1);
  for (int i = threadIdx.x; i < KS * KS * 64 + 1; i += 256) slab[i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

// Two-level deterministic slab reduction.  Level 1: out[g][i] = sum of `per` consecutive slabs (grid.y = groups).
__global__ __launch_bounds__(256) void stencil1_slab_group_kernel(const float* slabs, int blocks, int per, int width,
                                                                  float* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= width) return;
  const int b0 = blockIdx.y * per, b1 = min(blocks, b0 + per);
  float s = 0.f;
  for (int b = b0; b < b1; ++b) s += slabs[(size_t)b * width + i];
  out[(size_t)blockIdx.y * width + i] = s;
}

// Level 2: dw[t][c] (c < C) and db from the remaining [groups][taps*64+1] slabs
__global__ __launch_bounds__(256) void stencil1_bwdf_reduce_kernel(const float* slabs, int blocks, int taps, int c,
                                                                   float* dw, float* db) {
  const int width = taps * 64 + 1;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= width) return;
  float s = 0.f;
  for (int b = 0; b < blocks; ++b) s += slabs[(size_t)b * width + i];
  if (i == taps * 64) {
    if (db) db[0] = s;
  } else {
    const int t = i / 64, ch = i % 64;
    if (ch < c) dw[t * c + ch] = s;
  }
}


// ================================================================================================================
// Round 5: the layer's BACKWARD on packed fp32 FMAs (v_pk_fma_f32), half a wave per pixel.
// A lane owns TWO adjacent channels (one 8-byte piece), 32 lanes cover the 64 channels of a pixel, and every
// multiply-add is a packed one over the channel pair.  Needs an even channel count and 8-byte aligned pixels.
// (A forward of the same shape — 32 outputs per half-wave, the partial sums reduced by a halving exchange — was built
// and measured at the time of the lane = channel kernel above, 21.5 vs 21.5 us at B = 32: neither is bound by its vector
// instructions, both pull each pixel through the L1 three times.)
// ================================================================================================================
typedef float pk2 __attribute__((ext_vector_type(2)));

// acc += a * g with ONE scalar g for both halves: g is the low / the high dword of an aligned register pair (as a 16-byte LDS
// read leaves it) and the instruction's op_sel bits broadcast it — spelled in asm because the compiler builds the splat
// with two v_mov per multiply-add otherwise (243 moves beside 200 packed FMAs in stencil1_bwd_kernel's loop).
__device__ __forceinline__ void pkfma_lo(pk2& acc, pk2 a, pk2 gpair) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(gpair));
}
__device__ __forceinline__ void pkfma_hi(pk2& acc, pk2 a, pk2 gpair) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(a), "v"(gpair));
}

// Backward, both gradients in ONE pass over x: for an input pixel q and tap t the SAME output
// gradient g = dz[q - t + pad] enters  dw[t][c] += x[q][c] * g  and  dx[q][c] += w[t][c] * g, so a lane that holds x[q][c..c+1]
// does both with the g it fetched once (50 packed FMAs per pixel and channel pair), writes dx (masked by x > 0: the
// ReluGrad of the layer below, whose output x is) and keeps dw in registers; x is read once, dx written once.
//   block = (image, band of RB input rows); dz of the band's rows +- the filter reach sits zero-padded in LDS;
//   half a wave walks runs of 4 consecutive pixels (the 5 x 8 window of dz it needs: ten 16-byte LDS reads);
//   dw / db: halves -> waves (LDS) -> one slab per block -> groups of blocks -> total, in index order at every level, the
//   last-arriving block of a group / the last group doing the adding (tickets in `state`, which every call leaves zero).
struct Stencil1BwdParams {
  const float* x; const float* dz; const float* w; void* dx; float* slabs; float* gslabs; float* dw; float* db;
  unsigned* state;
  int n, h, w_in, c, ldx, ho, wo, pad_t, pad_l, ldy, lddx;
  int bands, runs_per_row, dwid, group, ngroups, mask, xvec4;
};

// The end of a backward block: the four waves' filter-gradient sums (red[4][TW], (tap, channel) order) and bias sums become the
// block's slab, the last-arriving block of a group adds the group's slabs, the last group the groups' — in index order at every
// level (the same bits on every run); tickets in p.state, which every call leaves zero.
template <int KS>
__device__ __forceinline__ void stencil1_bwd_reduce_tail(const Stencil1BwdParams& p, const float* red, const float* dbred, unsigned& last) {
  constexpr int TW = KS * KS * 64;
  constexpr int SLAB4 = TW / 4 + 1;
  // slabs cross CUs (and XCDs) as 16-byte write-through stores, drained before the block's ticket, and are read back with
  // sc1 loads (MI355X_MICROARCH.md, inter-workgroup visibility: the last-arriver form)
  const int nblocks = gridDim.x;
  const size_t slab_bytes = (size_t)SLAB4 * 16;
  const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.slabs, (unsigned long long)nblocks * slab_bytes);
  const __amdgpu_buffer_rsrc_t rg = make_rsrc(p.gslabs, (unsigned long long)p.ngroups * slab_bytes);
  for (int i4 = threadIdx.x; i4 < SLAB4; i4 += 256) {
    f32x4 s;
    if (i4 < TW / 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(red + 4 * i4), b = *reinterpret_cast<const f32x4*>(red + TW + 4 * i4);
      const f32x4 c = *reinterpret_cast<const f32x4*>(red + 2 * TW + 4 * i4), d = *reinterpret_cast<const f32x4*>(red + 3 * TW + 4 * i4);
      s = (a + b) + (c + d);
    } else {
      s = f32x4{(dbred[0] + dbred[1]) + (dbred[2] + dbred[3]), 0.f, 0.f, 0.f};
    }
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s), rs, (int)(blockIdx.x * slab_bytes + i4 * 16), 0, 16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int grp = blockIdx.x / p.group;
  const int gfirst = grp * p.group, gcount = min(p.group, nblocks - gfirst);
  if (threadIdx.x == 0) last = atomicInc(&p.state[1 + grp], (unsigned)gcount - 1u) == (unsigned)gcount - 1u;
  __syncthreads();
  if (!last) return;
  auto sum_slabs = [&](const __amdgpu_buffer_rsrc_t r, int firstslab, int count, int i4) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    constexpr int U = 8;
    for (int b0 = 0; b0 < count; b0 += U) {            // eight loads in flight, added in slab order
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
        v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, b0 + u < count ? (int)((firstslab + b0 + u) * slab_bytes + i4 * 16) : (int)kOOB, 0, 16);
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (b0 + u < count) s += __builtin_bit_cast(f32x4, v[u]);
    }
    return s;
  };
  for (int i4 = threadIdx.x; i4 < SLAB4; i4 += 256) {
    const f32x4 s = sum_slabs(rs, gfirst, gcount, i4);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, s), rg, (int)(grp * slab_bytes + i4 * 16), 0, 16);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) last = atomicInc(&p.state[0], (unsigned)p.ngroups - 1u) == (unsigned)p.ngroups - 1u;
  __syncthreads();
  if (!last) return;
  for (int i4 = threadIdx.x; i4 < SLAB4; i4 += 256) {
    const f32x4 s = sum_slabs(rg, 0, p.ngroups, i4);
    if (i4 == TW / 4) {
      if (p.db) p.db[0] = s[0];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int i = 4 * i4 + e, t = i / 64, ch = i % 64;
        if (ch < p.c) p.dw[t * p.c + ch] = s[e];
      }
    }
  }
}


template <int KS, int RB, bool DX16>
__global__ __launch_bounds__(256, 2) void stencil1_bwd_kernel(const Stencil1BwdParams p) {
  constexpr int DROWS = RB + KS - 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* D = smem;                                   // [DROWS][dwid]
  float* X = smem + DROWS * p.dwid;                  // [RB][w_in][c]: the band of x; later the four waves' filter-gradient sums
  float* red = X;
  __shared__ float dbred[4];
  __shared__ unsigned last;
  const int lane = threadIdx.x & 63, half = lane >> 5, cl = lane & 31, wv = threadIdx.x >> 6;
  const int img = blockIdx.x / p.bands, iy0 = (blockIdx.x - img * p.bands) * RB;
  const bool live = 2 * cl < p.c;
  // ---- the band of x -> LDS: every thread's loads in flight at once (rows past the image: the buffer's end, i.e. zeros)
  {
    const int rows = min(RB, p.h - iy0);
    const size_t first = ((size_t)img * p.h + iy0) * p.w_in * p.ldx;
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.x + first, ((unsigned long long)(rows * p.w_in - 1) * p.ldx + p.c) * 4ull);
    if (p.xvec4) {                                   // c, ldx multiples of 4, 16-byte aligned pixels
      const int c4 = p.c >> 2, total = RB * p.w_in * c4;
      constexpr int U = 8;
      for (int e0 = threadIdx.x; e0 < total; e0 += 256 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = e0 + u * 256, px = e / c4, q = e - px * c4;
          v[u] = __builtin_amdgcn_raw_buffer_load_b128(rb, e < total ? (px * p.ldx + 4 * q) * 4 : (int)kOOB, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int e = e0 + u * 256;
          if (e < total) *reinterpret_cast<u32x4*>(X + 4 * e) = v[u];
        }
      }
    } else {
      const int c2 = p.c >> 1, total = RB * p.w_in * c2;
      for (int e = threadIdx.x; e < total; e += 256) {
        const int px = e / c2, q = e - px * c2;
        *reinterpret_cast<u32x2*>(X + 2 * e) = __builtin_amdgcn_raw_buffer_load_b64(rb, (px * p.ldx + 2 * q) * 4, 0, 0);
      }
    }
  }
  // ---- dz window of the band -> LDS (zeros outside the output), and this block's share of BiasAddGrad: the output rows
  //      iy0 .. iy0+RB-1 (every output row belongs to exactly one band: ho <= h)
  float dbp = 0.f;
  for (int i = threadIdx.x; i < DROWS * p.dwid; i += 256) {
    const int a = i / p.dwid, bcol = i - a * p.dwid;
    const int oy = iy0 + p.pad_t - (KS - 1) + a, ox = bcol - (KS - 1) + p.pad_l;
    float g = 0.f;
    if ((unsigned)oy < (unsigned)p.ho && (unsigned)ox < (unsigned)p.wo) {
      g = p.dz[(((size_t)img * p.ho + oy) * p.wo + ox) * p.ldy];
      if (oy >= iy0 && oy < iy0 + RB) dbp += g;
    }
    D[i] = g;
  }
  dbp = wave_sum64(dbp);
  if (lane == 0) dbred[wv] = dbp;
  pk2 wreg[KS * KS], acc[KS * KS];
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) {
    wreg[t] = live ? *reinterpret_cast<const pk2*>(p.w + t * p.c + 2 * cl) : pk2{0.f, 0.f};
    acc[t] = pk2{0.f, 0.f};
  }
  __syncthreads();
  const int units = RB * p.runs_per_row;             // (row of the band, run of 4 pixels)
  const int hwb = wv * 2 + half;
  const int iters = (units + 7) / 8;                 // the same trip count for every half-wave of the block
  for (int k = 0; k < iters; ++k) {
    const int u = min(k * 8 + hwb, units - 1);       // (a half-wave past the end repeats the last run with x = 0)
    const bool uok = k * 8 + hwb < units;
    const int j = u / p.runs_per_row, run = u - j * p.runs_per_row;
    const int iy = iy0 + j;
    pk2 v[4];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int ix = min(4 * run + pp, p.w_in - 1);
      const pk2 t = *reinterpret_cast<const pk2*>(X + (j * p.w_in + ix) * p.c + (live ? 2 * cl : 0));
      const bool ok = uok && live && 4 * run + pp < p.w_in;
      v[pp] = pk2{ok ? t[0] : 0.f, ok ? t[1] : 0.f};
    }
    pk2 G[KS][4];                                      // G[r][m] = dz window columns 2m, 2m+1 of filter row r
#pragma unroll
    for (int r = 0; r < KS; ++r) {
      const f32x4* src = reinterpret_cast<const f32x4*>(D + (j + (KS - 1) - r) * p.dwid + 4 * run);
      const f32x4 g0 = src[0], g1 = src[1];
      G[r][0] = pk2{g0[0], g0[1]}; G[r][1] = pk2{g0[2], g0[3]};
      G[r][2] = pk2{g1[0], g1[1]}; G[r][3] = pk2{g1[2], g1[3]};
    }
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      pk2 d = {0.f, 0.f};
#pragma unroll
      for (int r = 0; r < KS; ++r)
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const int col = pp + (KS - 1) - s;           // window column of this pixel's tap
          if (col & 1) {
            pkfma_hi(acc[r * KS + s], v[pp], G[r][col >> 1]);
            pkfma_hi(d, wreg[r * KS + s], G[r][col >> 1]);
          } else {
            pkfma_lo(acc[r * KS + s], v[pp], G[r][col >> 1]);
            pkfma_lo(d, wreg[r * KS + s], G[r][col >> 1]);
          }
        }
      const int ix = 4 * run + pp;
      if (live && uok && iy < p.h && ix < p.w_in) {
        if (p.mask) {
          d[0] = v[pp][0] > 0.f ? d[0] : 0.f;
          d[1] = v[pp][1] > 0.f ? d[1] : 0.f;
        }
        const size_t o = (((size_t)img * p.h + iy) * p.w_in + ix) * p.lddx + 2 * cl;
        if constexpr (DX16) {
          typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
          *reinterpret_cast<bf16x2*>(static_cast<__bf16*>(p.dx) + o) = bf16x2{(__bf16)d[0], (__bf16)d[1]};
        } else {
          *reinterpret_cast<pk2*>(static_cast<float*>(p.dx) + o) = d;
        }
      }
    }
  }
  __syncthreads();                                     // the band of x is dead: its LDS takes the waves' sums
  // ---- dw: the two halves of a wave, then the four waves, then the block's slab
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) {
    const float a0 = acc[t][0] + __shfl_xor(acc[t][0], 32, 64);
    const float a1 = acc[t][1] + __shfl_xor(acc[t][1], 32, 64);
    if (half == 0) *reinterpret_cast<pk2*>(red + (wv * KS * KS + t) * 64 + 2 * cl) = pk2{a0, a1};
  }
  __syncthreads();
  stencil1_bwd_reduce_tail<KS>(p, red, dbred, last);
}


// ---- forward on the matrix cores through a taps-as-columns product (round 6) ----
// y[p] = b + sum_{r,s} ( sum_c x[p + (r,s)][c] w[r][s][c] ): the inner sum over the 64 channels of ONE input pixel for all 25
// taps is a [pixels][64] x [64][25] product — the input as it lies in memory times a 6.4-KB matrix that lives in registers —
// and the outer sum then adds 25 numbers per output pixel out of LDS.  A block owns a band of RB output rows of one image: it
// reads the RB + 4 input rows ONCE with 16-byte loads straight into MFMA A fragments (lane = pixel of a 32-pixel piece of a
// row, its 64 channels in eight float4: no LDS staging of x, no cross-lane reduction: the lane-per-channel kernel above spends a
// quarter of its wave cycles in ds_bpermute and reads every pixel three times through the L1), leaves Z[pixel][tap] in LDS
// (25 floats per pixel) and sums, skipping the tap columns that fall outside a row (SAME padding).  Input rows outside the
// image are loaded as zeros (out-of-range buffer offsets).  Bands of one image are neighbours on one XCD (block remap), so the
// four halo rows two bands share come from that L2.
constexpr int kS1mRB = 8;                           // output rows per block
constexpr int kS1mZP = 25;                          // floats per pixel of Z (odd: consecutive pixels fall into different banks)
struct Stencil1MfmaParams {
  const float* x; const float* w; const float* bias; float* y;
  int n, h, w_in, ho, wo, pad_t, pad_l, ldy, act, bands;
};
__global__ __launch_bounds__(256, 1) void stencil1_fwd_mfma_kernel(const Stencil1MfmaParams p) {
  extern __shared__ __attribute__((aligned(16))) float Zs[];        // [(RB + 4) * w_in pixels][25], the band's pixels in memory order
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  uint32_t bid = blockIdx.x;
  {
    const uint32_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int img = (int)bid / p.bands, band = (int)bid - img * p.bands;
  const int oy0 = band * kS1mRB;
  const int rows_out = min(kS1mRB, p.ho - oy0);
  const int iy0 = oy0 - p.pad_t, nrows = rows_out + 4;
  // B fragments: lane (tap li, half lh) holds w[tap][8 u + 4 lh .. + 3] for the eight channel chunks u
  f32x4 wf[8];
#pragma unroll
  for (int u = 0; u < 8; ++u)
    wf[u] = li < 25 ? *reinterpret_cast<const f32x4*>(p.w + li * 64 + 8 * u + 4 * lh) : f32x4{0.f, 0.f, 0.f, 0.f};
  // The band's input rows are ONE contiguous run of nrows * w_in pixels of 256 bytes (rows above / below the image: a run that
  // starts before / ends after the image's pixels — those pixels are out of range for the descriptor of THIS image and load as
  // zeros).  Pieces of 32 pixels of that run, no padding at row ends: 28 pieces for 12 rows of 74 instead of 36.
  const long long img_px = (long long)img * p.h * p.w_in;
  const int px_lo = max(0, -iy0) * p.w_in;                                     // first band pixel that exists
  const int px_hi = (min(p.h, iy0 + nrows) - iy0) * p.w_in;                    // one past the last
  const int npx = nrows * p.w_in, ntiles = (npx + 31) / 32;
  const __amdgpu_buffer_rsrc_t rx = make_rsrc(p.x + (img_px + (long long)iy0 * p.w_in) * 64, 0x7fffffffull);      // base may lie before the image: offsets below px_lo are never used
  auto tile_off = [&](int t) -> uint32_t {
    const int px = t * 32 + li;
    return (t < ntiles && px >= px_lo && px < px_hi) ? (uint32_t)((px * 64 + 4 * lh) * 4) : kOOB;
  };
  f32x4 af[2][8];
  auto fetch = [&](int t, auto set_c) {
    constexpr int set = decltype(set_c)::value;
    const uint32_t off = tile_off(t);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 32 * u, 0);
      af[set][u] = __builtin_bit_cast(f32x4, v);
    }
  };
  auto tile = [&](int t, auto set_c, auto next_c) {
    constexpr int set = decltype(set_c)::value;
    fetch(t + 4, next_c);                            // the wave's next piece is in flight while this one is multiplied
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][u][j], wf[u][j], acc, 0, 0, 0);
    if (li < 25) {
#pragma unroll
      for (int v = 0; v < 16; ++v) {
        const int px = t * 32 + 8 * (v >> 2) + 4 * lh + (v & 3);
        if (px < npx) Zs[px * kS1mZP + li] = acc[v];
      }
    }
  };
  fetch(wave, std::integral_constant<int, 0>{});
  for (int t = wave; t < ntiles; t += 8) {
    tile(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
    if (t + 4 < ntiles) tile(t + 4, std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  }
  __syncthreads();
  const float b = p.bias ? p.bias[0] : 0.f;
  for (int o = tid; o < rows_out * p.wo; o += 256) {
    const int oyl = o / p.wo, ox = o - oyl * p.wo;
    float sum = b;
#pragma unroll
    for (int s2 = 0; s2 < 5; ++s2) {
      const int ix = ox + s2 - p.pad_l;                                        // input column of tap column s2: outside the row = padding
      if ((unsigned)ix >= (unsigned)p.w_in) continue;
      const float* z = Zs + (oyl * p.w_in + ix) * kS1mZP + s2;
#pragma unroll
      for (int r = 0; r < 5; ++r) sum += z[(r * p.w_in) * kS1mZP + r * 5];
    }
    if (p.act == EPI_RELU) sum = fmaxf(sum, 0.f);
    else if (p.act == EPI_SIGMOID) sum = 1.f / (1.f + expf(-sum));
    p.y[(((size_t)img * p.ho + oy0 + oyl) * p.wo + ox) * p.ldy] = sum;
  }
}

static size_t s1m_lds_bytes(const a3d_conv_desc* d) { return (size_t)(kS1mRB + 4) * d->w * kS1mZP * 4 + 16; }
// 64 densely packed channels, whole 16-byte pieces, one descriptor over x (31-bit byte offsets), Z of a band in LDS
static bool stencil1_mfma_applicable(const a3d_conv_desc* d, const float* x, const float* w) {
  if (!stencil1_applicable(d) || d->c != 64 || d->ldx != 64 || d->precision != A3D_PREC_F32 || d->storage) return false;
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(w) & 15)) return false;
  if ((double)d->n * d->h * d->w * 64.0 * 4.0 >= 2147483647.0) return false;
  if (s1m_lds_bytes(d) > 150 * 1024) return false;
  return tune_int("A3D_STENCIL_MFMA", 1) != 0;
}

static const int kStencilBlocks = 1024;
static const int kStencilGroup = 32;

bool stencil1_applicable(const a3d_conv_desc* d) {
  return d->k == 1 && d->r == 5 && d->s == 5 && d->stride == 1 && d->c <= 64;
}

size_t stencil1_bwdf_ws_bytes(const a3d_conv_desc* d) {
  (void)d;
  return (size_t)(kStencilBlocks + kStencilBlocks / kStencilGroup) * (25 * 64 + 1) * 4;
}

static Stencil1Params make_params(const a3d_conv_desc* d) {
  Stencil1Params p{};
  p.n = d->n; p.h = d->h; p.w_in = d->w; p.c = d->c; p.ldx = d->ldx; p.ho = d->ho; p.wo = d->wo;
  p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.ldy = d->ldy;
  p.strips_per_row = (d->wo + P - 1) / P;
  p.total_strips = d->n * d->ho * p.strips_per_row;
  return p;
}

int stencil1_fwd(const a3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int act,
                 hipStream_t st) {
  if (stencil1_mfma_applicable(d, x, w)) {
    Stencil1MfmaParams q{};
    q.x = x; q.w = w; q.bias = bias; q.y = y; q.act = act;
    q.n = d->n; q.h = d->h; q.w_in = d->w; q.ho = d->ho; q.wo = d->wo; q.pad_t = d->pad_t; q.pad_l = d->pad_l; q.ldy = d->ldy;
    q.bands = (d->ho + kS1mRB - 1) / kS1mRB;
    const size_t lds = s1m_lds_bytes(d);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stencil1_fwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    clear_stale_error();
    hipLaunchKernelGGL(stencil1_fwd_mfma_kernel, dim3(d->n * q.bands), dim3(256), lds, st, q);
    return check_launch("stencil1_fwd_mfma");
  }
  Stencil1Params p = make_params(d);
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.act = act;
  const int tiles = p.n * ((p.ho + 3) / 4) * p.strips_per_row;      // 4 output rows x 8 columns per wave iteration
  const int blocks = std::min((tiles + 3) / 4, 4096);
  clear_stale_error();
  hipLaunchKernelGGL(stencil1_fwd_kernel<5>, dim3(blocks), dim3(256), 0, st, p);
  return check_launch("stencil1_fwd");
}

// ---- both gradients in one pass (stencil1_bwd_kernel) ----
static constexpr int kBwdRB = 4;
static void bwd_geometry(const a3d_conv_desc* d, int& bands, int& blocks, int& group, int& ngroups) {
  bands = (d->h + kBwdRB - 1) / kBwdRB;
  blocks = d->n * bands;
  group = 32;
  if ((blocks + group - 1) / group > 63) group = (blocks + 62) / 63;
  ngroups = (blocks + group - 1) / group;
}

// dz window of the band + the band of x (later: four waves' sums), two blocks per CU
static size_t bwd_lds_bytes(const a3d_conv_desc* d) {
  const int dwid = 4 * ((d->w + 3) / 4) + 4;
  return (size_t)((kBwdRB + 4) * dwid + std::max(4 * 25 * 64, kBwdRB * d->w * d->c)) * 4;
}

bool stencil1_bwd_both_applicable(const a3d_conv_desc* d) {
  return stencil1_applicable(d) && d->c % 2 == 0 && d->ldx % 2 == 0 && d->pad_t >= 0 && d->pad_t < 5 && d->pad_l >= 0 &&
         d->pad_l < 5 && d->ho <= d->h && d->wo <= d->w && bwd_lds_bytes(d) <= 79 * 1024 &&
         (unsigned long long)d->n * d->h * d->w * d->ldx * 4ull <= 0x40000000ull;
}

size_t stencil1_bwd_both_ws_bytes(const a3d_conv_desc* d) {
  int bands, blocks, group, ngroups;
  bwd_geometry(d, bands, blocks, group, ngroups);
  return (size_t)(blocks + ngroups) * (25 * 64 + 4) * 4 + 16;
}

int stencil1_bwd_both(const a3d_conv_desc* d, const float* x, const float* dz, const float* w, float* dw, float* db,
                      void* dx, int lddx, int dx_bf16, int relu_mask, unsigned* state, void* ws, hipStream_t st) {
  Stencil1BwdParams p{};
  p.x = x; p.dz = dz; p.w = w; p.dx = dx; p.dw = dw; p.db = db; p.state = state;
  p.n = d->n; p.h = d->h; p.w_in = d->w; p.c = d->c; p.ldx = d->ldx; p.ho = d->ho; p.wo = d->wo;
  p.pad_t = d->pad_t; p.pad_l = d->pad_l; p.ldy = d->ldy; p.lddx = lddx; p.mask = relu_mask;
  int blocks;
  bwd_geometry(d, p.bands, blocks, p.group, p.ngroups);
  p.runs_per_row = (d->w + 3) / 4;
  p.dwid = 4 * p.runs_per_row + 4;
  p.slabs = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 15) & ~(uintptr_t)15);
  p.gslabs = p.slabs + (size_t)blocks * (25 * 64 + 4);
  p.xvec4 = d->c % 4 == 0 && d->ldx % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  const size_t lds = bwd_lds_bytes(d);
  // above 64 KiB of dynamic LDS: the attribute is per device and cheap, set on every call
  if (dx_bf16) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stencil1_bwd_kernel<5, kBwdRB, true>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  else (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stencil1_bwd_kernel<5, kBwdRB, false>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  clear_stale_error();
  if (dx_bf16) hipLaunchKernelGGL((stencil1_bwd_kernel<5, kBwdRB, true>), dim3(blocks), dim3(256), lds, st, p);
  else hipLaunchKernelGGL((stencil1_bwd_kernel<5, kBwdRB, false>), dim3(blocks), dim3(256), lds, st, p);
  return check_launch("stencil1_bwd_both");
}

int stencil1_bwd_filter(const a3d_conv_desc* d, const float* x, const float* dz, float* dw, float* db, void* ws,
                        hipStream_t st) {
  Stencil1Params p = make_params(d);
  p.x = x; p.dz = dz; p.slabs = static_cast<float*>(ws);
  const int blocks = std::min((p.total_strips + 3) / 4, kStencilBlocks);
  clear_stale_error();
  hipLaunchKernelGGL(stencil1_bwdf_kernel<5>, dim3(blocks), dim3(256), 0, st, p);
  int rc = check_launch("stencil1_bwd_filter");
  if (rc != A3D_OK) return rc;
  const int width = 25 * 64 + 1;
  const int groups = (blocks + kStencilGroup - 1) / kStencilGroup;
  float* level1 = static_cast<float*>(ws) + (size_t)kStencilBlocks * width;
  clear_stale_error();
  hipLaunchKernelGGL(stencil1_slab_group_kernel, dim3((width + 255) / 256, groups), dim3(256), 0, st,
                     static_cast<const float*>(ws), blocks, kStencilGroup, width, level1);
  rc = check_launch("stencil1_slab_group");
  if (rc != A3D_OK) return rc;
  clear_stale_error();
  hipLaunchKernelGGL(stencil1_bwdf_reduce_kernel, dim3((width + 255) / 256), dim3(256), 0, st,
                     static_cast<const float*>(level1), groups, 25, d->c, dw, db);
  return check_launch("stencil1_bwd_filter_reduce");
}

}  // namespace a3d
