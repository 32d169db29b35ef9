// conv3.hip — forward convolution of the few-channel layers (Cin <= 4, unpadded: coarse/conv/conv2d_0 11x11 s4,
// fine/first 9x9 s2, src/models.py:211,241; DCNF's first conv 11x11 s1, src/models.py:64) on the fp32 matrix cores,
// with operands taken straight from L1/L2: no LDS, no barriers, no im2col tile.
//
// With few channels a filter row (s, c) of one output pixel is ONE contiguous run of S*Cin floats of the image.  The K
// axis is (r, q): filter row r, position q inside the run, the run padded to a multiple of 4 floats (the filter gets zero
// rows for the pad; the image floats read there — the next pixels of the same row — are replaced by zeros before they
// reach the matrix cores: a non-finite pixel OUTSIDE a window must not turn into 0 * inf = NaN inside it, TensorFlow's
// Conv2D never touches it).  Then
//   * the k -> (r, q) decode is the same for every pixel, so a lane's A address is `pixel base + offset(k)`: one add;
//   * four consecutive k of one pixel are 16 contiguous bytes: lane (row li, half h) of a v_mfma_f32_32x32x2_f32 quad
//     loads k = 8u+4h .. +3 of its row with ONE buffer_load_dwordx4 and feeds MFMA j with element j;
//   * the filter is repacked per call as [K/4][N][4] (150 KB, one tiny kernel), so the B fragment of a lane (column li)
//     is one coalesced 16-byte load as well.  (Reading the stored HWIO filter with four 4-byte loads per fragment
//     instead — no repack, no workspace — was measured: 14 instead of 5 load instructions per 24 MFMAs cost 20 %,
//     conv2d_0 96 -> 118 us, fine/first 163 -> 187 us.)
// A wave owns TM x TN accumulator tiles of 32x32 (64 or 128 output pixels x 64 or 96 filters) and walks all of K; its
// loads for chunk u+1 are in flight while chunk u's 4*TM*TN MFMAs issue.  Waves never meet: the input (26 MB at B = 32)
// and the packed filter (<= 160 KB) live in L2, the XCD-aware block order keeps an XCD on one eighth of the images.
// The generic implicit-GEMM kernel staged these layers through LDS with 4- and 8-byte loads and a per-lane tap decode:
// matrix pipe 0.50-0.55 busy; this form has nothing but loads and MFMAs in its loop.
//
// Fused 2x2 / stride-2 max pool (the train step never writes the pre-pool activation): GEMM rows are enumerated pool
// window by pool window (row = 4 * window + position), so the four conv outputs of a window are four accumulator
// registers of one lane; only their maximum and (optionally) its position are stored — the contract of
// a3d_conv2d_pool_fwd, same as the generic kernel's.
#include <algorithm>

#include "a3d_internal.h"
#include "igemm.h"

namespace a3d {

struct Conv3Params {
  const float* x; const float* wp; const float* bias; float* y; uint8_t* argmax;
  unsigned long long x_bytes, wp_bytes;
  int M;                 // GEMM rows: n*ho*wo, or 4 * n*(ho/2)*(wo/2) with the fused pool
  int N, Np;             // filters, and the packed filter's column count (a multiple of 32*TN)
  int Kp;                // padded K = multiple of 8 >= R * RLP
  int Kreal;             // R * RLP: chunks beyond it (the K tail) load nothing
  int RL, RLP;           // run length S*Cin and its padded length (multiple of 4)
  int rowpitch;          // floats per image row = W * Cin
  int imgpitch;          // floats per image = H * W * Cin
  int step;              // floats between the runs of neighbouring output pixels = stride * Cin
  int stride;
  int ldc, act, pool, c16;
  int m_tiles;           // wave tasks along M
  FastDiv div_img, div_row;      // (pooled) pixels per image, per row
  FastDiv div_rlp;
};

// filter [R][RL][N] (HWIO with s, c fused) -> [Kp/4][Np][4], zero where q >= RL, r >= R or n >= N
__global__ __launch_bounds__(256) void conv3_pack_kernel(const float* __restrict__ w, float* __restrict__ wp, int R, int RL,
                                                         int RLP, int N, int Np, int Kp) {
  const int total = Kp * Np;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int j = i & 3, col = (i >> 2) % Np, k4 = (i >> 2) / Np;
    const int k = 4 * k4 + j, r = k / RLP, q = k - r * RLP;
    wp[i] = (r < R && q < RL && col < N) ? w[(size_t)(r * RL + q) * N + col] : 0.f;
  }
}

// ---- epilogue shared by the fp32 and the bf16 form: bias, activation, (max pool + argmax), store ----
template <int TM, int TN, bool POOL>
__device__ __forceinline__ void conv3_epilogue(const f32x16 (&acc)[TM][TN], const Conv3Params& p, int m0, int n0, int li, int lh) {
#pragma unroll
  for (int a = 0; a < TM; ++a) {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int col = n0 + b * 32 + li;
      if (col >= p.N) continue;
      const float bias = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = m0 + a * 32 + 8 * g + 4 * lh;       // the lane holds rows row .. row+3 of this column
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = acc[a][b][4 * g + i] + bias;
          if (p.act == EPI_RELU) v[i] = fmaxf(v[i], 0.f);
          else if (p.act == EPI_SIGMOID) v[i] = 1.f / (1.f + __expf(-v[i]));
        }
        if (POOL) {
          if (row >= p.M) continue;
          // the values a separate conv would have stored (rounded to bf16 where the output is), compared the way MaxPool /
          // MaxPoolGrad scan them: the first maximum wins
          if (p.c16) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (float)(__bf16)v[i];
          }
          float val = v[0];
          int arg = 0;
#pragma unroll
          for (int i = 1; i < 4; ++i)
            if (v[i] > val) { val = v[i]; arg = i; }
          const size_t o = (size_t)(row >> 2) * p.ldc + col;
          if (p.c16) reinterpret_cast<__bf16*>(p.y)[o] = (__bf16)val;
          else p.y[o] = val;
          if (p.argmax) p.argmax[(size_t)(row >> 2) * p.N + col] = (uint8_t)arg;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (row + i >= p.M) continue;
            const size_t o = (size_t)(row + i) * p.ldc + col;
            if (p.c16) reinterpret_cast<__bf16*>(p.y)[o] = (__bf16)v[i];
            else p.y[o] = v[i];
          }
        }
      }
    }
  }
}

// DEPTH: chunks of 8 k in registers per wave (one being multiplied, DEPTH - 1 in flight).  2 for the kernel that runs
// alone — four wavefronts per SIMD cover each other's load latency; 4 under A3D_HINT_SHARE_CU, where two wavefronts per
// SIMD have to cover it themselves so that the other stream's bandwidth-bound kernels find half the register file free.
template <int TM, int TN, bool POOL, int DEPTH = 2>
__global__ __launch_bounds__(DEPTH > 2 ? 512 : 256, DEPTH > 2 ? 2 : (TM * TN > 4 ? 3 : 4)) void conv3_fwd_kernel(const Conv3Params p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  // XCD-aware order: the blocks an XCD receives (ids congruent mod 8) work on one contiguous eighth of the tiles
  uint32_t bid = blockIdx.x;
  {
    const uint32_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile = (int)bid * (DEPTH > 2 ? 8 : 4) + wave;       // wave task: tile_n fastest (8 waves per block under the hint)
  const int tiles_n = p.Np / (32 * TN);
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  if (tile_m >= p.m_tiles) return;
  const int m0 = tile_m * (32 * TM), n0 = tile_n * (32 * TN);

  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.x, p.x_bytes), rsB = make_rsrc(p.wp, p.wp_bytes);
  // byte offset of each of this lane's rows' reference pixel
  uint32_t a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int row = m0 + a * 32 + li;
    uint32_t off = kOOB;
    if (row < p.M) {
      uint32_t pix, sub = 0;
      if (POOL) { pix = (uint32_t)row >> 2; sub = (uint32_t)row & 3u; } else pix = (uint32_t)row;
      const uint32_t img = fdiv(pix, p.div_img), rem = pix - img * p.div_img.d;
      uint32_t oy = fdiv(rem, p.div_row), ox = rem - oy * p.div_row.d;
      if (POOL) { oy = 2 * oy + (sub >> 1); ox = 2 * ox + (sub & 1u); }
      off = (img * (uint32_t)p.imgpitch + oy * (uint32_t)(p.stride * p.rowpitch) + ox * (uint32_t)p.step) * 4u;
    }
    a_base[a] = off;
  }
  const uint32_t b_base = (uint32_t)((n0 + li) * 16);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;

  const int nchunks = p.Kp / 8;
  f32x4 af[DEPTH][TM], bf[DEPTH][TN];
  int nv[DEPTH];                                      // how many of a chunk's four k lie inside the run (>= 4: all)
  // Beside fp32 MFMAs no vector instruction of the same wave is hidden (5-6 cycles each against the MFMA's 64), so
  // a chunk's addresses are kept off the vector unit: chunks are fetched in order and everything that depends only on the chunk
  // and on the lane's k half lives in SCALAR registers for both halves — position in the filter row (one compare-and-wrap per
  // chunk instead of a division), byte offset, how many of its four k are inside the run — and a lane picks its half's with one
  // select each.  "Outside" is 2^31 in either summand of an offset and the two are added with unsigned saturation (out of range
  // either way, no select); a chunk past the end is fetched like any other — its A at 2^31: zeros — so the ring needs no branch
  // and no registers to clear; the filter's offset is a scalar (the load's soffset), clamped to the last chunk.
  int s_q = 0, s_k = 0;                               // scalar: element q of the run and k of the NEXT chunk's first half
  uint32_t s_row = 0;                                 // ... byte offset of its filter row
  const uint32_t rowstep = (uint32_t)p.rowpitch * 4u;
  const uint32_t b_lane = (uint32_t)(lh * p.Np) * 16u + b_base;
  const int bstep = 2 * p.Np * 16, b_last = (nchunks - 1) * bstep;
  int s_b = 0;
  bool s_tail[DEPTH];                                 // scalar: the chunk has pad positions to clear (either half)
  auto fetch = [&](int buf) {                         // the next chunk in order
    int q1 = s_q + 4;                                 // the second half's four k: the next 4-group of the run, or the next row's first
    uint32_t row1 = s_row;
    if (q1 >= p.RLP) { q1 -= p.RLP; row1 += rowstep; }
    const uint32_t koff0 = s_k < p.Kreal ? s_row + (uint32_t)s_q * 4u : kOOB;       // K tail and chunks past the end: nothing to read
    const uint32_t koff1 = s_k + 4 < p.Kreal ? row1 + (uint32_t)q1 * 4u : kOOB;
    const int nv0 = p.RL - s_q, nv1 = p.RL - q1;      // q <= RLP - 4 < RL: element 0 is always inside
    s_tail[buf] = nv0 < 4 || nv1 < 4;
    const uint32_t koff = lh ? koff1 : koff0;
    nv[buf] = lh ? nv1 : nv0;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)__builtin_elementwise_add_sat(a_base[a], koff), 0, 0);
      af[buf][a] = {__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    }
    const int sb = s_b < b_last ? s_b : b_last;       // (past the end: the last chunk's weights again, against zeros)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(b_lane + (uint32_t)b * 512u), sb, 0);
      bf[buf][b] = {__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
    }
    s_k += 8;
    s_b += bstep;
    s_q += 8;                                         // RLP % 4 == 0: one wrap at most unless the run is a single 4-group
    if (s_q >= p.RLP) { s_q -= p.RLP; s_row += rowstep; }
    if (s_q >= p.RLP) { s_q -= p.RLP; s_row += rowstep; }
  };
  auto multiply = [&](int buf) {
    // pad positions of the run (only the last 4-group of a filter row has any) hold neighbouring pixels: zero them here,
    // at the point of use — the selects sit where the wait for this chunk's loads is anyway, not behind the fetch
    if (s_tail[buf]) {                                // (scalar: only the chunk that holds a filter row's last 4-group)
#pragma unroll
      for (int j = 1; j < 4; ++j)
#pragma unroll
        for (int a = 0; a < TM; ++a) af[buf][a][j] = nv[buf] > j ? af[buf][a][j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[buf][a][j], bf[buf][b][j], acc[a][b], 0, 0, 0);
  };
  if constexpr (DEPTH == 2) {
    fetch(0);
    int u = 0;
    for (; u + 2 <= nchunks; u += 2) {                  // two chunks per trip: the register double buffer is static
      fetch(1);
      __builtin_amdgcn_sched_barrier(0);
      multiply(0);
      fetch(0);                                         // (past the end: zeros, never multiplied)
      __builtin_amdgcn_sched_barrier(0);
      multiply(1);
    }
    if (u < nchunks) multiply(0);
  } else {
    // ring of DEPTH register sets, DEPTH chunks per trip (static indices); a chunk past the end is fetched with every
    // offset out of range (zeros: multiplying them adds nothing), so the trip needs no tail logic
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i) fetch(i);
    for (int u = 0; u < nchunks; u += DEPTH) {
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) {
        fetch((i + DEPTH - 1) % DEPTH);
        __builtin_amdgcn_sched_barrier(0);            // (the loads stay here, DEPTH - 1 chunks ahead of their use)
        multiply(i);
      }
    }
  }

  conv3_epilogue<TM, TN, POOL>(acc, p, m0, n0, li, lh);
}

// ================================================================================================================
// The same forward on the bf16 matrix cores (round 5; BASELINE config 5: conv2d_0 and fine/first at batch 64), from the
// 4-channel bf16 copy of the image (a3d_pad_channels_bf16: 8-byte pixels, so a filter row's run of S*4 bf16 starts on a
// 16-byte boundary for every even stride).  K = (r, q) with the run padded to a multiple of 8 elements: lane (row li,
// half lh) of v_mfma_f32_32x32x16_bf16 holds k = 16u + 8 lh .. + 7 — ONE 16-byte load straight from L2, as is its B
// fragment from the filter packed [K/8][N][8].  No LDS, no barrier; per k-step a wave issues TM + TN loads for TM * TN
// MFMAs of 32 cycles.  The generic bf16 kernel (igemm_bf16.h) staged these layers through LDS at 0.06-0.11 of the bf16
// peak (profiles/r04_bench_layers_bf16.txt).  Pad positions of a run hold the next pixels of the row and meet zero
// weights (the stated deviation of the bf16 modes for non-finite pixels, include/a3d.h).
// ================================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// filter [R][S][4][N] float32 (HWIO, 4 channels: the 4th is the pad channel) -> bf16 [Kp/8][Np][8], zero where q >= RL, r >= R, n >= N
__global__ __launch_bounds__(256) void conv3b_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int R, int RL,
                                                          int RLP, int N, int Np, int Kp) {
  const int total = Kp * Np;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int j = i & 7, col = (i >> 3) % Np, k8 = (i >> 3) / Np;
    const int k = 8 * k8 + j, r = k / RLP, q = k - r * RLP;
    wp[i] = (__bf16)((r < R && q < RL && col < N) ? w[(size_t)(r * RL + q) * N + col] : 0.f);
  }
}

template <int TM, int TN, bool POOL>
// (the six-tile form at most two wavefronts per SIMD: with its addressing in scalar registers it would fit three, and ran 12 % slower
// that way — three k-steps of loads in flight per wave are what covers the latency here, not more waves on the same L1)
__global__ __launch_bounds__(256, TM * TN > 4 ? 2 : 3) __attribute__((amdgpu_waves_per_eu(TM * TN > 4 ? 2 : 3, TM * TN > 4 ? 2 : 3)))
void conv3b_fwd_kernel(const Conv3Params p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lh = lane >> 5;
  uint32_t bid = blockIdx.x;
  {
    const uint32_t nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile = (int)bid * 4 + wave;
  const int tiles_n = p.Np / (32 * TN);
  const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
  if (tile_m >= p.m_tiles) return;
  const int m0 = tile_m * (32 * TM), n0 = tile_n * (32 * TN);
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.x, p.x_bytes), rsB = make_rsrc(p.wp, p.wp_bytes);
  uint32_t a_base[TM];
#pragma unroll
  for (int a = 0; a < TM; ++a) {
    const int row = m0 + a * 32 + li;
    uint32_t off = kOOB;
    if (row < p.M) {
      uint32_t pix, sub = 0;
      if (POOL) { pix = (uint32_t)row >> 2; sub = (uint32_t)row & 3u; } else pix = (uint32_t)row;
      const uint32_t img = fdiv(pix, p.div_img), rem = pix - img * p.div_img.d;
      uint32_t oy = fdiv(rem, p.div_row), ox = rem - oy * p.div_row.d;
      if (POOL) { oy = 2 * oy + (sub >> 1); ox = 2 * ox + (sub & 1u); }
      off = (img * (uint32_t)p.imgpitch + oy * (uint32_t)(p.stride * p.rowpitch) + ox * (uint32_t)p.step) * 2u;     // bf16 elements
    }
    a_base[a] = off;
  }
  const uint32_t b_base = (uint32_t)((n0 + li) * 16);
  f32x16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int v = 0; v < 16; ++v) acc[a][b][v] = 0.f;
  const int nsteps = p.Kp / 16;
  constexpr int DEPTH = 3;                            // k-steps in registers: one being multiplied, two in flight
  bf16x8 af[DEPTH][TM], bf[DEPTH][TN];
  // chunk addressing in scalar registers for both k halves, as in conv3_fwd_kernel: k-steps are fetched in order, a lane picks
  // its half's offset with one select; saturating adds keep "outside" (2^31) outside; the filter's offset is the load's scalar
  // offset, clamped to the last k-step (past the end A reads zeros)
  int s_q = 0, s_k = 0;
  uint32_t s_row = 0;
  const uint32_t rowstep = (uint32_t)p.rowpitch * 2u;
  const uint32_t b_lane = (uint32_t)(lh * p.Np) * 16u + b_base;
  const int bstep = 2 * p.Np * 16, b_last = (nsteps - 1) * bstep;
  int s_b = 0;
  auto fetch = [&](int buf) {                         // the next k-step in order
    int q1 = s_q + 8;                                 // the second half's eight k: the next 8-group of the run, or the next row's first
    uint32_t row1 = s_row;
    if (q1 >= p.RLP) { q1 -= p.RLP; row1 += rowstep; }
    const uint32_t koff0 = s_k < p.Kreal ? s_row + (uint32_t)s_q * 2u : kOOB;
    const uint32_t koff1 = s_k + 8 < p.Kreal ? row1 + (uint32_t)q1 * 2u : kOOB;
    const uint32_t koff = lh ? koff1 : koff0;
#pragma unroll
    for (int a = 0; a < TM; ++a)
      af[buf][a] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)__builtin_elementwise_add_sat(a_base[a], koff), 0, 0));
    const int sb = s_b < b_last ? s_b : b_last;
#pragma unroll
    for (int b = 0; b < TN; ++b)
      bf[buf][b] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsB, (int)(b_lane + (uint32_t)b * 512u), sb, 0));
    s_k += 16;
    s_b += bstep;
    s_q += 16;                                        // RLP % 8 == 0: two wraps at most (a run of one 8-group)
    if (s_q >= p.RLP) { s_q -= p.RLP; s_row += rowstep; }
    if (s_q >= p.RLP) { s_q -= p.RLP; s_row += rowstep; }
  };
  auto multiply = [&](int buf) {
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[buf][a], bf[buf][b], acc[a][b], 0, 0, 0);
  };
  // a ring of DEPTH register sets, DEPTH k-steps per trip (static indices); a k-step past the end fetches zeros
#pragma unroll
  for (int i = 0; i < DEPTH - 1; ++i) fetch(i);
  for (int u = 0; u < nsteps; u += DEPTH) {
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
      fetch((i + DEPTH - 1) % DEPTH);
      // the loads stay HERE, two k-steps ahead of their use: with nothing on the vector unit tying them down the scheduler
      // moved all three steps' loads behind the trip's first MFMAs and opened the trip with s_waitcnt vmcnt(0)
      __builtin_amdgcn_sched_barrier(0);
      multiply(i);
    }
  }
  conv3_epilogue<TM, TN, POOL>(acc, p, m0, n0, li, lh);
}

// ---- host side ----
struct Conv3Shape {
  int RL, RLP, Kreal, Kp, TN, Np;
};
static Conv3Shape conv3_shape(const a3d_conv_desc* d) {
  Conv3Shape s;
  s.RL = d->s * d->c;
  s.RLP = (s.RL + 3) / 4 * 4;
  s.Kreal = d->r * s.RLP;
  s.Kp = (s.Kreal + 7) / 8 * 8;
  s.TN = (d->k + 31) / 32 >= 3 ? 3 : 2;               // 96 filters: one 3-tile wave column; 63 / 64: two tiles
  s.Np = (d->k + 32 * s.TN - 1) / (32 * s.TN) * (32 * s.TN);
  return s;
}

bool conv3_applicable(const a3d_conv_desc* d, const void* x) {
  static const bool off = tune_int("A3D_NO_CONV3", 0) != 0;      // A/B aid (tuning processes only)
  if (off) return false;
  if (d->precision != A3D_PREC_F32 || d->c > 4 || d->pad_t || d->pad_l || d->ldx != d->c) return false;
  if (d->storage & ~A3D_STORE_Y_BF16) return false;
  if (d->k < 33) return false;                          // (one-output-channel layers have their own stencil kernel)
  if (x && (reinterpret_cast<uintptr_t>(x) & 3)) return false;
  if ((d->ho - 1) * d->stride + d->r > d->h || (d->wo - 1) * d->stride + d->s > d->w) return false;   // VALID geometry
  if ((double)d->n * d->h * d->w * d->c * 4.0 >= 2147483647.0) return false;     // 31-bit byte offsets into the image
  return true;
}

size_t conv3_ws_bytes(const a3d_conv_desc* d) {
  const Conv3Shape s = conv3_shape(d);
  return ((size_t)s.Kp * s.Np * 4 + 255) / 256 * 256;
}

// share: A3D_HINT_SHARE_CU — the kernel uses no LDS, so a dynamic request of 80 KiB + 1 KiB caps it at ONE block per CU:
// eight waves (two per SIMD), and the other half of the CU's LDS stays free for the other stream's kernels (the dense
// layers' streaming kernels take 66 - 70 KiB per block: beside two 4-wave blocks of 54 KiB each they did not fit at all)
template <int TM, int TN>
static int conv3_launch(const Conv3Params& p, bool pool, unsigned blocks, bool share, hipStream_t st) {
  constexpr size_t kShare = (size_t)163840 / 2 + 1024;
  const size_t lds = share ? kShare : 0;
  static bool attr_done = false;
  if (share && !attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_fwd_kernel<TM, TN, true, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kShare) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3_fwd_kernel<TM, TN, false, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kShare) != hipSuccess)
      return set_error(A3D_ELAUNCH, "conv3: hipFuncSetAttribute failed");
    attr_done = true;
  }
  if (share) {                 // one 8-wave block per CU, three chunks in flight per wave
    if (pool) hipLaunchKernelGGL((conv3_fwd_kernel<TM, TN, true, 4>), dim3((blocks + 1) / 2), dim3(512), lds, st, p);
    else hipLaunchKernelGGL((conv3_fwd_kernel<TM, TN, false, 4>), dim3((blocks + 1) / 2), dim3(512), lds, st, p);
  } else {
    if (pool) hipLaunchKernelGGL((conv3_fwd_kernel<TM, TN, true>), dim3(blocks), dim3(256), lds, st, p);
    else hipLaunchKernelGGL((conv3_fwd_kernel<TM, TN, false>), dim3(blocks), dim3(256), lds, st, p);
  }
  return A3D_OK;
}

// the filter as conv3_fwd_kernel reads it: [K/4][Np][4] (a3d_conv2d_fwd_prepare_filter, or per call into the workspace)
int conv3_pack(const a3d_conv_desc* d, const float* w, float* wp, hipStream_t st) {
  const Conv3Shape s = conv3_shape(d);
  clear_stale_error();
  hipLaunchKernelGGL(conv3_pack_kernel, dim3(std::min((s.Kp * s.Np + 255) / 256, 1024)), dim3(256), 0, st, w, wp, d->r, s.RL,
                     s.RLP, d->k, s.Np, s.Kp);
  return check_launch("conv3_pack");
}

int conv3_fwd(const a3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int act, int pool,
              int ld_out, uint8_t* argmax, void* ws, size_t ws_bytes, hipStream_t st, bool prepared) {
  const Conv3Shape s = conv3_shape(d);
  const float* wp = w;                                 // prepared: w already is the packed filter
  int rc;
  if (!prepared) {
    if (!ws || ws_bytes < conv3_ws_bytes(d)) return set_error(A3D_EWORKSPACE, "conv3_fwd: need %zu workspace bytes", conv3_ws_bytes(d));
    rc = conv3_pack(d, w, static_cast<float*>(ws), st);
    if (rc != A3D_OK) return rc;
    wp = static_cast<const float*>(ws);
  } else if (reinterpret_cast<uintptr_t>(w) & 15) {
    return set_error(A3D_EINVAL, "conv3_fwd: a prepared filter is 16-byte aligned");
  }
  Conv3Params p{};
  p.x = x; p.wp = wp; p.bias = bias; p.y = y; p.argmax = argmax;
  p.x_bytes = (unsigned long long)d->n * d->h * d->w * d->c * 4ull;
  p.wp_bytes = (unsigned long long)s.Kp * s.Np * 4ull;
  const int ph = d->ho / 2, pw = d->wo / 2;
  p.M = pool ? d->n * ph * pw * 4 : d->n * d->ho * d->wo;
  p.N = d->k; p.Np = s.Np; p.Kp = s.Kp; p.Kreal = s.Kreal; p.RL = s.RL; p.RLP = s.RLP;
  p.rowpitch = d->w * d->c; p.imgpitch = d->h * d->w * d->c; p.step = d->stride * d->c; p.stride = d->stride;
  p.ldc = pool ? ld_out : d->ldy; p.act = act; p.pool = pool; p.c16 = (d->storage & A3D_STORE_Y_BF16) ? 1 : 0;
  p.div_img = make_fastdiv(pool ? ph * pw : d->ho * d->wo);
  p.div_row = make_fastdiv(pool ? pw : d->wo);
  p.div_rlp = make_fastdiv(s.RLP);
  const int tiles_n = s.Np / (32 * s.TN);
  const int TM = 2;                                    // 64 output pixels (16 pool windows) per wave
  p.m_tiles = (p.M + 32 * TM - 1) / (32 * TM);
  const unsigned blocks = (unsigned)(((long)p.m_tiles * tiles_n + 3) / 4);
  clear_stale_error();
  const bool share = (d->hints & A3D_HINT_SHARE_CU) != 0;
  rc = s.TN == 3 ? conv3_launch<2, 3>(p, pool != 0, blocks, share, st) : conv3_launch<2, 2>(p, pool != 0, blocks, share, st);
  if (rc != A3D_OK) return rc;
  return check_launch("conv3_fwd");
}



// ---- bf16 form from the 4-channel bf16 image (conv3b_fwd_kernel) ----
struct Conv3bShape {
  int RL, RLP, Kreal, Kp, TN, Np;
};
static Conv3bShape conv3b_shape(const a3d_conv_desc* d) {
  Conv3bShape s;
  s.RL = d->s * 4;
  s.RLP = (s.RL + 7) / 8 * 8;
  s.Kreal = d->r * s.RLP;
  s.Kp = (s.Kreal + 15) / 16 * 16;
  s.TN = (d->k + 31) / 32 >= 3 ? 3 : 2;
  s.Np = (d->k + 32 * s.TN - 1) / (32 * s.TN) * (32 * s.TN);
  return s;
}

// (the caller has checked the image form: 4 bf16 channels, densely packed, no padding, even stride, padded runs inside their row)
bool conv3b_applicable(const a3d_conv_desc* d) {
  if (tune_int("A3D_NO_CONV3B", 0) != 0) return false;       // A/B aid (tuning processes only)
  if (d->k < 33) return false;
  if ((double)d->n * d->h * d->w * 4 * 2.0 >= 2147483647.0) return false;        // 31-bit byte offsets into the image
  return true;
}

size_t conv3b_filter_bytes(const a3d_conv_desc* d) {
  const Conv3bShape s = conv3b_shape(d);
  return ((size_t)s.Kp * s.Np * 2 + 255) / 256 * 256;
}

int conv3b_pack(const a3d_conv_desc* d, const float* w, void* wp, hipStream_t st) {
  const Conv3bShape s = conv3b_shape(d);
  clear_stale_error();
  hipLaunchKernelGGL(conv3b_pack_kernel, dim3(std::min((s.Kp * s.Np + 255) / 256, 1024)), dim3(256), 0, st, w, static_cast<__bf16*>(wp),
                     d->r, s.RL, s.RLP, d->k, s.Np, s.Kp);
  return check_launch("conv3b_pack");
}

int conv3b_fwd(const a3d_conv_desc* d, const void* x, const float* w, const float* bias, void* y, int act, int pool, int ld_out,
               uint8_t* argmax, void* ws, size_t ws_bytes, hipStream_t st, bool prepared) {
  const Conv3bShape s = conv3b_shape(d);
  const void* wp = w;
  int rc;
  if (!prepared) {
    if (!ws || ws_bytes < conv3b_filter_bytes(d)) return set_error(A3D_EWORKSPACE, "conv3b_fwd: need %zu workspace bytes", conv3b_filter_bytes(d));
    rc = conv3b_pack(d, w, ws, st);
    if (rc != A3D_OK) return rc;
    wp = ws;
  } else if (reinterpret_cast<uintptr_t>(w) & 15) {
    return set_error(A3D_EINVAL, "conv3b_fwd: a prepared filter is 16-byte aligned");
  }
  Conv3Params p{};
  p.x = static_cast<const float*>(x); p.wp = static_cast<const float*>(wp); p.bias = bias; p.y = static_cast<float*>(y); p.argmax = argmax;
  p.x_bytes = (unsigned long long)d->n * d->h * d->w * 4 * 2ull;
  p.wp_bytes = (unsigned long long)s.Kp * s.Np * 2ull;
  const int ph = d->ho / 2, pw = d->wo / 2;
  p.M = pool ? d->n * ph * pw * 4 : d->n * d->ho * d->wo;
  p.N = d->k; p.Np = s.Np; p.Kp = s.Kp; p.Kreal = s.Kreal; p.RL = s.RL; p.RLP = s.RLP;
  p.rowpitch = d->w * 4; p.imgpitch = d->h * d->w * 4; p.step = d->stride * 4; p.stride = d->stride;
  p.ldc = pool ? ld_out : d->ldy; p.act = act; p.pool = pool; p.c16 = (d->storage & A3D_STORE_Y_BF16) ? 1 : 0;
  p.div_img = make_fastdiv(pool ? ph * pw : d->ho * d->wo);
  p.div_row = make_fastdiv(pool ? pw : d->wo);
  p.div_rlp = make_fastdiv(s.RLP);
  const int tiles_n = s.Np / (32 * s.TN);
  const int TM = 2;
  p.m_tiles = (p.M + 32 * TM - 1) / (32 * TM);
  const unsigned blocks = (unsigned)(((long)p.m_tiles * tiles_n + 3) / 4);
  clear_stale_error();
  if (s.TN == 3) {
    if (pool) hipLaunchKernelGGL((conv3b_fwd_kernel<2, 3, true>), dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv3b_fwd_kernel<2, 3, false>), dim3(blocks), dim3(256), 0, st, p);
  } else {
    if (pool) hipLaunchKernelGGL((conv3b_fwd_kernel<2, 2, true>), dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv3b_fwd_kernel<2, 2, false>), dim3(blocks), dim3(256), 0, st, p);
  }
  return check_launch("conv3b_fwd");
}

}  // namespace a3d
