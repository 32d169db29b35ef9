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
  Stencil1Params p = make_params(d);
  p.x = x; p.w = w; p.bias = bias; p.y = y; p.act = act;
  const int tiles = p.n * ((p.ho + 3) / 4) * p.strips_per_row;      // 4 output rows x 8 columns per wave iteration
  const int blocks = std::min((tiles + 3) / 4, 4096);
  clear_stale_error();
  hipLaunchKernelGGL(stencil1_fwd_kernel<5>, dim3(blocks), dim3(256), 0, st, p);
  return check_launch("stencil1_fwd");
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
