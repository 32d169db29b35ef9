// pointwise.hip — the HBM-bound kernels of the path: max-pool (+concat) and its gradient (+ReluGrad), legacy
// bilinear resize, patch extraction, scale-invariant log loss (wavefront-shuffle reductions) and TF-1.3 ApplyAdam.
// This file is compiled with -ffp-contract=off (see Makefile): the oracle does these computations as separate
// fp32 operations, and without FMA contraction the kernels are bit-exact against the numpy restatement (HIP's
// __f*_rn intrinsics are plain operators on ROCm 7.2 and do not prevent contraction by themselves).
#include <algorithm>

#include "a3d_internal.h"

namespace a3d {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static inline unsigned grid_for(size_t total, int per_block = 256, unsigned cap = 8192) {
  size_t g = (total + per_block - 1) / per_block;
  return (unsigned)std::min<size_t>(std::max<size_t>(g, 1), cap);
}

// ------------------------------------------------------------------ max pool 2x2/2 VALID
// one thread per output element; channel index fastest -> coalesced along NHWC's C.
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          const float* __restrict__ extra, int n, int h, int w, int c,
                                                          int ho, int wo, int ldy) {
  const int cout = c + (extra ? 1 : 0);
  const size_t total = (size_t)n * ho * wo * cout;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % cout);
    const size_t pix = i / cout;
    float v;
    if (ch < c) {
      const int q = (int)(pix % wo);
      const size_t t = pix / wo;
      const int p = (int)(t % ho);
      const int b = (int)(t / ho);
      const float* s = x + (((size_t)b * h + 2 * p) * w + 2 * q) * c + ch;
      v = fmaxf(fmaxf(s[0], s[c]), fmaxf(s[(size_t)w * c], s[(size_t)w * c + c]));
    } else {
      v = extra[pix];
    }
    y[pix * ldy + ch] = v;
  }
}

// one thread per 2x2 cell of the INPUT grid (ceil(h/2) x ceil(w/2)); cells cut by VALID flooring write zeros.
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dx, int n, int h, int w, int c, int ho,
                                                          int wo, int lddy, int relu_mask) {
  const int hc = (h + 1) / 2, wc = (w + 1) / 2;
  const size_t total = (size_t)n * hc * wc * c;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    size_t t = i / c;
    const int q = (int)(t % wc);
    t /= wc;
    const int p = (int)(t % hc);
    const int b = (int)(t / hc);
    const size_t base = (((size_t)b * h + 2 * p) * w + 2 * q) * c + ch;
    const bool full = p < ho && q < wo;
    if (full) {
      const float v0 = x[base], v1 = x[base + c], v2 = x[base + (size_t)w * c], v3 = x[base + (size_t)w * c + c];
      int arg = 0;
      float best = v0;
      if (v1 > best) { best = v1; arg = 1; }
      if (v2 > best) { best = v2; arg = 2; }
      if (v3 > best) { best = v3; arg = 3; }
      float g = dy[(((size_t)b * ho + p) * wo + q) * lddy + ch];
      if (relu_mask && !(best > 0.f)) g = 0.f;
      dx[base] = arg == 0 ? g : 0.f;
      dx[base + c] = arg == 1 ? g : 0.f;
      dx[base + (size_t)w * c] = arg == 2 ? g : 0.f;
      dx[base + (size_t)w * c + c] = arg == 3 ? g : 0.f;
    } else {
      const bool has_r = 2 * p + 1 < h, has_c = 2 * q + 1 < w;
      dx[base] = 0.f;
      if (has_c) dx[base + c] = 0.f;
      if (has_r) dx[base + (size_t)w * c] = 0.f;
      if (has_r && has_c) dx[base + (size_t)w * c + c] = 0.f;
    }
  }
}

// the same from the recorded argmax position and the pooled value (a3d_conv2d_pool_fwd): one thread per 2x2 cell of
// the input grid, like maxpool_bwd_kernel (cells cut by VALID flooring write zeros)
__global__ __launch_bounds__(256) void maxpool_bwd_idx_kernel(const uint8_t* __restrict__ argmax,
                                                              const float* __restrict__ y, const float* __restrict__ dy,
                                                              float* __restrict__ dx, int n, int h, int w, int c, int ho,
                                                              int wo, int ldy, int lddy, int relu_mask) {
  const int hc = (h + 1) / 2, wc = (w + 1) / 2;
  const size_t total = (size_t)n * hc * wc * c;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    size_t t = i / c;
    const int q = (int)(t % wc);
    t /= wc;
    const int p = (int)(t % hc);
    const int b = (int)(t / hc);
    const size_t base = (((size_t)b * h + 2 * p) * w + 2 * q) * c + ch;
    if (p < ho && q < wo) {
      const size_t win = ((size_t)b * ho + p) * wo + q;
      const int arg = argmax[win * c + ch];
      float g = dy[win * lddy + ch];
      if (relu_mask && !(y[win * ldy + ch] > 0.f)) g = 0.f;
      dx[base] = arg == 0 ? g : 0.f;
      dx[base + c] = arg == 1 ? g : 0.f;
      dx[base + (size_t)w * c] = arg == 2 ? g : 0.f;
      dx[base + (size_t)w * c + c] = arg == 3 ? g : 0.f;
    } else {
      const bool has_r = 2 * p + 1 < h, has_c = 2 * q + 1 < w;
      dx[base] = 0.f;
      if (has_c) dx[base + c] = 0.f;
      if (has_r) dx[base + (size_t)w * c] = 0.f;
      if (has_r && has_c) dx[base + (size_t)w * c + c] = 0.f;
    }
  }
}

// four channels per thread, 16-byte accesses (c, ldy, lddy multiples of 4; 16-byte aligned tensors)
__global__ __launch_bounds__(256) void maxpool_bwd_idx_vec4_kernel(const uint8_t* __restrict__ argmax,
                                                                   const float* __restrict__ y, const float* __restrict__ dy,
                                                                   float* __restrict__ dx, int n, int h, int w, int c, int ho,
                                                                   int wo, int ldy, int lddy, int relu_mask) {
  const int hc = (h + 1) / 2, wc = (w + 1) / 2, c4 = c / 4;
  const uint32_t total = (uint32_t)n * hc * wc * c4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const uint32_t ch = (i % c4) * 4;
    uint32_t t = i / c4;
    const uint32_t q = t % wc;
    t /= wc;
    const uint32_t p = t % hc, b = t / hc;
    const size_t base = (((size_t)b * h + 2 * p) * w + 2 * q) * c + ch;
    const bool has_r = 2 * p + 1 < (uint32_t)h, has_c = 2 * q + 1 < (uint32_t)w;
    f32x4 o0 = zero, o1 = zero, o2 = zero, o3 = zero;
    if (p < (uint32_t)ho && q < (uint32_t)wo) {
      const size_t win = ((size_t)b * ho + p) * wo + q;
      const uint32_t a4 = *reinterpret_cast<const uint32_t*>(argmax + win * c + ch);
      const f32x4 g4 = *reinterpret_cast<const f32x4*>(dy + win * lddy + ch);
      const f32x4 y4 = *reinterpret_cast<const f32x4*>(y + win * ldy + ch);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t arg = (a4 >> (8 * j)) & 0xffu;
        const float g = (relu_mask && !(y4[j] > 0.f)) ? 0.f : g4[j];
        o0[j] = arg == 0 ? g : 0.f; o1[j] = arg == 1 ? g : 0.f; o2[j] = arg == 2 ? g : 0.f; o3[j] = arg == 3 ? g : 0.f;
      }
    }
    *reinterpret_cast<f32x4*>(dx + base) = o0;
    if (has_c) *reinterpret_cast<f32x4*>(dx + base + c) = o1;
    if (has_r) *reinterpret_cast<f32x4*>(dx + base + (size_t)w * c) = o2;
    if (has_r && has_c) *reinterpret_cast<f32x4*>(dx + base + (size_t)w * c + c) = o3;
  }
}

// ------------------------------------------------------------------ bf16 storage (BASELINE config 5)
// float32 [pixels][c_src] -> bf16 [pixels][4], missing channels zero: an 8-byte pixel, so that the window runs of a
// few-channel conv start 16 bytes apart at even strides and the bf16 kernel can gather them (igemm_host.hip)
__global__ __launch_bounds__(256) void pad_channels_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst,
                                                                size_t pixels, int c_src) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < pixels; i += (size_t)gridDim.x * 256) {
    bf16x4 o = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    for (int c = 0; c < c_src; ++c) o[c] = (__bf16)src[i * c_src + c];
    reinterpret_cast<bf16x4*>(dst)[i] = o;
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const void* __restrict__ src, void* __restrict__ dst, size_t count,
                                                        int to_bf16) {
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  const size_t nvec = count / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    if (to_bf16) {
      const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
      const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
      reinterpret_cast<bf16x4*>(dst)[i] = o;
    } else {
      const bf16x4 v = reinterpret_cast<const bf16x4*>(src)[i];
      const f32x4 o = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
      reinterpret_cast<f32x4*>(dst)[i] = o;
    }
  }
  if (blockIdx.x == 0) {
    const size_t i = nvec * 4 + threadIdx.x;
    if (i < count) {
      if (to_bf16) reinterpret_cast<__bf16*>(dst)[i] = (__bf16) reinterpret_cast<const float*>(src)[i];
      else reinterpret_cast<float*>(dst)[i] = (float)reinterpret_cast<const __bf16*>(src)[i];
    }
  }
}

// max pool 2x2/2 and its gradient on bf16 tensors: the fp32 kernels above with 2-byte elements (max of bf16 values is
// exact in either type; first maximum in scan order)
__global__ __launch_bounds__(256) void maxpool_fwd_bf16_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y,
                                                               const float* __restrict__ extra, int n, int h, int w, int c,
                                                               int ho, int wo, int ldx, int ldy) {
  const int cout = c + (extra ? 1 : 0);
  const size_t total = (size_t)n * ho * wo * cout;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % cout);
    const size_t pix = i / cout;
    __bf16 v;
    if (ch < c) {
      const int q = (int)(pix % wo);
      const size_t t = pix / wo;
      const int p = (int)(t % ho);
      const int b = (int)(t / ho);
      const __bf16* s = x + (((size_t)b * h + 2 * p) * w + 2 * q) * ldx + ch;
      const float m = fmaxf(fmaxf((float)s[0], (float)s[ldx]), fmaxf((float)s[(size_t)w * ldx], (float)s[(size_t)w * ldx + ldx]));
      v = (__bf16)m;
    } else {
      v = (__bf16)extra[pix];
    }
    y[pix * ldy + ch] = v;
  }
}

__global__ __launch_bounds__(256) void maxpool_bwd_bf16_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dy,
                                                               __bf16* __restrict__ dx, int n, int h, int w, int c, int ho,
                                                               int wo, int ldx, int lddy, int relu_mask) {
  const int hc = (h + 1) / 2, wc = (w + 1) / 2;
  const size_t total = (size_t)n * hc * wc * c;
  const __bf16 zero = (__bf16)0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    size_t t = i / c;
    const int q = (int)(t % wc);
    t /= wc;
    const int p = (int)(t % hc);
    const int b = (int)(t / hc);
    const size_t base = (((size_t)b * h + 2 * p) * w + 2 * q) * ldx + ch;
    const size_t c_ = (size_t)ldx;      // x and dx share the pixel stride
    const bool full = p < ho && q < wo;
    if (full) {
      const float v0 = (float)x[base], v1 = (float)x[base + c_], v2 = (float)x[base + (size_t)w * c_],
                  v3 = (float)x[base + (size_t)w * c_ + c_];
      int arg = 0;
      float best = v0;
      if (v1 > best) { best = v1; arg = 1; }
      if (v2 > best) { best = v2; arg = 2; }
      if (v3 > best) { best = v3; arg = 3; }
      __bf16 g = dy[(((size_t)b * ho + p) * wo + q) * lddy + ch];
      if (relu_mask && !(best > 0.f)) g = zero;
      dx[base] = arg == 0 ? g : zero;
      dx[base + c_] = arg == 1 ? g : zero;
      dx[base + (size_t)w * c_] = arg == 2 ? g : zero;
      dx[base + (size_t)w * c_ + c_] = arg == 3 ? g : zero;
    } else {
      const bool has_r = 2 * p + 1 < h, has_c = 2 * q + 1 < w;
      dx[base] = zero;
      if (has_c) dx[base + c_] = zero;
      if (has_r) dx[base + (size_t)w * c_] = zero;
      if (has_r && has_c) dx[base + (size_t)w * c_ + c_] = zero;
    }
  }
}

// ------------------------------------------------------------------ ResizeBilinear (legacy, align_corners=False)
// One block per output row of one tensor (blockIdx.y selects the tensor of a pair: the step resizes the image and its
// depth map, src/models.py:282-283, in ONE launch): the row's source lines and vertical weight are scalars, a thread walks
// the row's (pixel, channel) elements with 32-bit arithmetic.  Same separate fp32 operations as before: bit-exact.
struct ResizeOne { const float* x; float* y; int h, w, c, oh, ow; float sy, sx; int u8; };
struct ResizePair { ResizeOne t[2]; int n; };
template <typename SRC>
__device__ __forceinline__ void resize_rows(const ResizeOne& r, int n, const float* lut) {
  const int rows = n * r.oh;
  const SRC* src = reinterpret_cast<const SRC*>(r.x);
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int b = row / r.oh, oy = row - b * r.oh;
    const float fy = __fmul_rn((float)oy, r.sy);
    const int y0 = (int)fy, y1 = min(y0 + 1, r.h - 1);
    const float ly = __fsub_rn(fy, (float)y0);
    const SRC* l0 = src + ((size_t)b * r.h + y0) * r.w * r.c;
    const SRC* l1 = src + ((size_t)b * r.h + y1) * r.w * r.c;
    float* out = r.y + (size_t)row * r.ow * r.c;
    const int ne = r.ow * r.c;
    auto tap = [&](const SRC* line, int i) -> float {
      if constexpr (sizeof(SRC) == 1) return lut[line[i]];
      else return line[i];
    };
    for (int e = threadIdx.x; e < ne; e += 256) {
      const int ox = e / r.c, ch = e - ox * r.c;
      const float fx = __fmul_rn((float)ox, r.sx);
      const int x0 = (int)fx, x1 = min(x0 + 1, r.w - 1);
      const float lx = __fsub_rn(fx, (float)x0);
      const float tl = tap(l0, x0 * r.c + ch), tr = tap(l0, x1 * r.c + ch);
      const float bl = tap(l1, x0 * r.c + ch), br = tap(l1, x1 * r.c + ch);
      const float top = __fadd_rn(tl, __fmul_rn(__fsub_rn(tr, tl), lx));
      const float bot = __fadd_rn(bl, __fmul_rn(__fsub_rn(br, bl), lx));
      out[e] = __fadd_rn(top, __fmul_rn(__fsub_rn(bot, top), ly));
    }
  }
}
__global__ __launch_bounds__(256) void resize_kernel(const ResizePair p) {
  const ResizeOne& r = p.t[blockIdx.y];
  if (r.u8) {
    // pixel value k -> the float the converter stored, plus the loader's 0.5: fl(fl(fl(k / 255) - 0.5) + 0.5), in the
    // separate correctly-rounded fp32 operations numpy performed (tools/data_tf_converter.py:36-37, src/data.py:84-85)
    __shared__ float lut[256];
    lut[threadIdx.x] = __fadd_rn(__fsub_rn(__fdiv_rn((float)threadIdx.x, 255.f), 0.5f), 0.5f);
    __syncthreads();
    resize_rows<uint8_t>(r, p.n, lut);
  } else {
    resize_rows<float>(r, p.n, nullptr);
  }
}

// ------------------------------------------------------------------ extract_image_patches (SAME, zero fill)
__global__ __launch_bounds__(256) void patches_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int h,
                                                      int w, int c, int k, int stride, int ph, int pw, int pad_t,
                                                      int pad_l) {
  const size_t total = (size_t)n * ph * pw * k * k * c;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    size_t t = i / c;
    const int j = (int)(t % k);
    t /= k;
    const int ii = (int)(t % k);
    t /= k;
    const int pc = (int)(t % pw);
    t /= pw;
    const int pr = (int)(t % ph);
    const int b = (int)(t / ph);
    const int sy = pr * stride - pad_t + ii, sx = pc * stride - pad_l + j;
    float v = 0.f;
    if ((unsigned)sy < (unsigned)h && (unsigned)sx < (unsigned)w) v = x[(((size_t)b * h + sy) * w + sx) * c + ch];
    y[i] = v;
  }
}

// ------------------------------------------------------------------ scale-invariant log loss
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

__device__ __forceinline__ float masked_log(float v) {
  float l = logf(__fadd_rn(v, 1e-8f));
  return isnan(l) ? 0.f : l;     // tf.where(tf.is_nan(log), 0, log): -inf is kept
}

// kSilogParts blocks per sample (one sample is 4070 pixels at MSDN's size: with one block per sample 32 CUs each walked a
// chain of load latencies and logarithms — 29 us on the step's critical path; now every CU holds one short piece).  Block
// (b, part) leaves its partial sums in ws[2nb+1 + 2(b*parts+part) ..]; the block that finishes last (ticket in ws[2nb],
// which wraps back to 0: nothing to re-initialise between calls) adds each sample's parts in part order — the same sum
// whatever the timing — writes ws[2b] = sum d^2, ws[2b+1] = sum d for the backward kernel, and the batch mean.  The
// partials cross CUs as write-through stores, drained before the ticket, and are read back past the L1
// (MI355X_MICROARCH.md, inter-workgroup visibility).
constexpr int kSilogParts = A3D_SILOG_PARTS;
__global__ __launch_bounds__(256) void silog_fwd_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                        float* __restrict__ ws, float* __restrict__ loss, int npix, int nb,
                                                        float c) {
  __shared__ float red[2][4];
  __shared__ unsigned last;
  const int b = blockIdx.x / kSilogParts, part = blockIdx.x % kSilogParts;
  const int chunk = (npix + kSilogParts - 1) / kSilogParts;
  const int lo = part * chunk, hi = min(npix, lo + chunk);
  const float* o = out + (size_t)b * npix;
  const float* t = tgt + (size_t)b * npix;
  float* partials = ws + 2 * nb + 1;
  float s2 = 0.f, s1 = 0.f;
  // four elements per thread and round, all eight loads issued before the first logarithm
  for (int i0 = lo + threadIdx.x; i0 < hi; i0 += 4 * 256) {
    float ov[4], tv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 256;
      ov[u] = i < hi ? o[i] : 0.f;
      tv[u] = i < hi ? t[i] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (i0 + u * 256 >= hi) break;
      const float d = __fsub_rn(masked_log(ov[u]), masked_log(tv[u]));
      s2 += d * d;
      s1 += d;
    }
  }
  s2 = wave_sum(s2);
  s1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = s2;
    red[1][threadIdx.x >> 6] = s1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned total = (unsigned)(nb * kSilogParts);
    __hip_atomic_store(&partials[2 * blockIdx.x], red[0][0] + red[0][1] + red[0][2] + red[0][3], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&partials[2 * blockIdx.x + 1], red[1][0] + red[1][1] + red[1][2] + red[1][3], __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    last = atomicInc(reinterpret_cast<unsigned*>(ws + 2 * nb), total - 1u) == total - 1u;
  }
  __syncthreads();
  if (!last || threadIdx.x >= 64) return;
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 64) {
    float a2 = 0.f, a1 = 0.f;
#pragma unroll
    for (int q = 0; q < kSilogParts; ++q) {
      a2 += __hip_atomic_load(&partials[2 * (i * kSilogParts + q)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      a1 += __hip_atomic_load(&partials[2 * (i * kSilogParts + q) + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ws[2 * i] = a2;
    ws[2 * i + 1] = a1;
    s += a2 - c * (a1 * a1);
  }
  s = wave_sum(s);
  if (threadIdx.x == 0) loss[0] = s / (float)nb;
}

__global__ __launch_bounds__(256) void silog_bwd_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                        const float* __restrict__ ws, float* __restrict__ dout, int b,
                                                        int npix, float c, float inv_b, __bf16* __restrict__ dout16 = nullptr,
                                                        int ld16 = 0) {
  const size_t total = (size_t)b * npix;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int smp = (int)(i / npix);
    const float o = out[i];
    const float arg = __fadd_rn(o, 1e-8f);
    const float lo = logf(arg);
    float g = 0.f;
    if (!isnan(lo)) {
      const float d = __fsub_rn(lo, masked_log(tgt[i]));
      const float sd = ws[2 * smp + 1];
      g = __fdiv_rn(__fmul_rn(__fsub_rn(__fmul_rn(2.f, d), __fmul_rn(__fmul_rn(2.f, c), sd)), inv_b), arg);
    }
    dout[i] = g;
    if (dout16) dout16[(size_t)smp * ld16 + (i - (size_t)smp * npix)] = (__bf16)g;      // the copy the bf16 dense layer reads (a3d_silog_loss_bwd_ex)
  }
}

// ------------------------------------------------------------------ ApplyAdam (TF 1.3 formula)
__device__ __forceinline__ void adam_one(float& var, float& m, float& v, float g, float omb1, float omb2, float alpha,
                                         float eps) {
  m = __fadd_rn(m, __fmul_rn(__fsub_rn(g, m), omb1));
  v = __fadd_rn(v, __fmul_rn(__fsub_rn(__fmul_rn(g, g), v), omb2));
  // v_sqrt_f32 is 1-ulp, not correctly rounded, and hipcc emits it bare; the f64 square root rounded to f32 is
  // correctly rounded (53 >= 2*24+2 bits), which is what the numpy/Eigen reference computes.
  const float sq = (float)sqrt((double)v);
  var = __fsub_rn(var, __fdiv_rn(__fmul_rn(m, alpha), __fadd_rn(sq, eps)));
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ var, float* __restrict__ m,
                                                   float* __restrict__ v, const float* __restrict__ g, size_t count,
                                                   float omb1, float omb2, float alpha, float eps, float gscale,
                                                   unsigned int* __restrict__ poisoned) {
  const size_t nvec = count / 4;
  const bool use_scale = gscale != 1.f;
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    f32x4 w4 = reinterpret_cast<f32x4*>(var)[i], m4 = reinterpret_cast<f32x4*>(m)[i];
    f32x4 v4 = reinterpret_cast<f32x4*>(v)[i], g4 = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gj = use_scale ? __fmul_rn(g4[j], gscale) : g4[j];
      float wj = w4[j], mj = m4[j], vj = v4[j];
      adam_one(wj, mj, vj, gj, omb1, omb2, alpha, eps);
      bad |= !isfinite(wj);
      w4[j] = wj; m4[j] = mj; v4[j] = vj;
    }
    reinterpret_cast<f32x4*>(var)[i] = w4;
    reinterpret_cast<f32x4*>(m)[i] = m4;
    reinterpret_cast<f32x4*>(v)[i] = v4;
  }
  if (blockIdx.x == 0) {
    const size_t i = nvec * 4 + threadIdx.x;
    if (i < count) {
      float gj = use_scale ? __fmul_rn(g[i], gscale) : g[i];
      adam_one(var[i], m[i], v[i], gj, omb1, omb2, alpha, eps);
      bad |= !isfinite(var[i]);
    }
  }
  if (poisoned && bad) atomicOr(poisoned, 1u);
}

// ApplyAdam specialised for alpha == 0 and 1-beta2 == 0 — what the reference's AdamOptimizer(rate, 0.9, 1) always is
// (src/models.py:309).  Then v += (g*g - v)*0 and var -= (m*0)/(sqrt(v)+eps) leave v and var bit-identical unless a
// non-finite value poisons them (inf*0 = NaN), so only m needs the full read-modify-write: 3 HBM streams instead of
// 7.  Poisoned elements get exactly the NaNs the general formula would produce.  (Assumes v holds no +-inf from a
// foreign checkpoint; v is only ever written by these two kernels, which never produce one.)
__global__ __launch_bounds__(256) void adam_frozen_kernel(float* __restrict__ var, float* __restrict__ m,
                                                          float* __restrict__ v, const float* __restrict__ g,
                                                          size_t count, float omb1, float gscale,
                                                          unsigned int* __restrict__ poisoned) {
  const size_t nvec = count / 4;
  const bool use_scale = gscale != 1.f;
  const float qnan = __builtin_nanf("");
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
    f32x4 m4 = reinterpret_cast<f32x4*>(m)[i], g4 = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = use_scale ? __fmul_rn(g4[j], gscale) : g4[j];
      m4[j] = __fadd_rn(m4[j], __fmul_rn(__fsub_rn(gj, m4[j]), omb1));
      const bool g_poison = !isfinite(__fmul_rn(gj, gj));
      if (g_poison || !isfinite(m4[j])) {             // rare; `bad` only when it changes something (a NaN stays a NaN)
        const size_t e = 4 * i + j;
        bad |= !isnan(var[e]) || (g_poison && !isnan(v[e]));
        if (g_poison) v[e] = qnan;
        var[e] = qnan;
      }
    }
    reinterpret_cast<f32x4*>(m)[i] = m4;
  }
  if (blockIdx.x == 0) {
    const size_t i = nvec * 4 + threadIdx.x;
    if (i < count) {
      const float gj = use_scale ? __fmul_rn(g[i], gscale) : g[i];
      const float mj = __fadd_rn(m[i], __fmul_rn(__fsub_rn(gj, m[i]), omb1));
      m[i] = mj;
      const bool g_poison = !isfinite(__fmul_rn(gj, gj));
      if (g_poison || !isfinite(mj)) {
        bad |= !isnan(var[i]) || (g_poison && !isnan(v[i]));
        if (g_poison) v[i] = qnan;
        var[i] = qnan;
      }
    }
  }
  if (poisoned && bad) atomicOr(poisoned, 1u);      // rare: a non-finite gradient reached this slice
}

// ------------------------------------------------------------------ dropout keep mask (Philox4x32-10)
// TF draws U[0,1) from its own Philox stream, which is not reproducible outside TF; this is the same generator
// family keyed by (seed, step), one counter per 4 mask bytes.  keep = floor(keep_prob + u)  (nn.dropout, TF 1.3).
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
  const uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
  const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
  c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ keep, size_t count, uint32_t seed_lo,
                                                           uint32_t seed_hi, uint32_t step_lo, uint32_t step_hi,
                                                           float keep_prob) {
  const size_t nquad = (count + 3) / 4;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < nquad; q += (size_t)gridDim.x * 256) {
    uint32_t c[4] = {(uint32_t)q, (uint32_t)(q >> 32), step_lo, step_hi};
    uint32_t k0 = seed_lo, k1 = seed_hi;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const size_t i = q * 4 + j;
      if (i < count) {
        const float u = (float)(c[j] >> 8) * (1.0f / 16777216.0f);      // 24 random bits -> [0,1)
        keep[i] = (uint8_t)floorf(keep_prob + u);
      }
    }
  }
}

}  // namespace a3d

namespace a3d {
// A measurement aid, not part of the training path: what a collective costs the kernels it runs beside.  Shaped like one rank's
// share of RCCL's reduce-scatter at N = 8 (src/ann3depth.py:77-92's replacement, dp.py): a few workgroups read `read_bytes`,
// add what they read, write `write_bytes`, and pace themselves to `bytes_per_tick` (s_memrealtime runs at 100 MHz) so that the
// launch lasts as long as the exchange would over xGMI.  bench.py --dp-rank-standin launches it on a second stream wherever a
// data-parallel rank would start a collective.
__global__ __launch_bounds__(256) void comm_standin_kernel(const float* __restrict__ src, size_t read_f4, float* __restrict__ dst,
                                                           size_t write_f4, float f4_per_tick) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const f4* s4 = reinterpret_cast<const f4*>(src);
  f4* d4 = reinterpret_cast<f4*>(dst);
  const size_t per = (read_f4 + gridDim.x - 1) / gridDim.x, lo = (size_t)blockIdx.x * per, hi = min(read_f4, lo + per);
  const size_t wper = (write_f4 + gridDim.x - 1) / gridDim.x, wlo = (size_t)blockIdx.x * wper, whi = min(write_f4, wlo + wper);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  constexpr int U = 8;                                  // 8 x 256 x 16 B = 32 KiB per block and round
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  size_t wpos = wlo + threadIdx.x;
  for (size_t i = lo; i < hi; i += 256 * U) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t e = i + u * 256 + threadIdx.x;
      v[u] = e < hi ? __builtin_nontemporal_load(s4 + e) : f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
    // writes keep pace with the reads: after reading a fraction f of its share a block has written the same fraction of what
    // it has to write (reduce-scatter: one piece per eight read; all-reduce stand-in, write_f4 == read_f4: eight per eight)
    const size_t wdue = wlo + (size_t)((double)(min(hi, i + 256 * U) - lo) * (double)(whi - wlo) / (double)(hi - lo));
    while (wpos < wdue) {
      __builtin_nontemporal_store(acc, d4 + wpos);
      wpos += 256;
    }
    // pace: this block's share of the rate
    const float due = (float)(i - lo + 256 * U) * (float)gridDim.x / f4_per_tick;
    while ((float)(__builtin_amdgcn_s_memrealtime() - t0) < due) __builtin_amdgcn_s_sleep(8);
  }
}

__global__ __launch_bounds__(256) void copy_channel_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           size_t npix, int ld_src, int c_src, int ld_dst, int c_dst) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256)
    dst[i * ld_dst + c_dst] = src[i * ld_src + c_src];
}
__global__ __launch_bounds__(256) void copy_channel_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst,
                                                                size_t npix, int ld_src, int c_src, int ld_dst, int c_dst) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256)
    dst[i * ld_dst + c_dst] = (__bf16)src[i * ld_src + c_src];
}
// a3d_maxpool2x2_bwd_idx with bf16 pooled values and bf16 dy; dx float32 (the 3-channel layers' filter gradient takes it)
__global__ __launch_bounds__(256) void maxpool_bwd_idx_bf16_kernel(const uint8_t* __restrict__ argmax,
                                                                   const __bf16* __restrict__ y, const __bf16* __restrict__ dy,
                                                                   float* __restrict__ dx, int n, int h, int w, int c, int ho,
                                                                   int wo, int ldy, int lddy, int relu_mask) {
  const int hc = (h + 1) / 2, wc = (w + 1) / 2;
  const size_t total = (size_t)n * hc * wc * c;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % c);
    size_t t = i / c;
    const int q = (int)(t % wc);
    t /= wc;
    const int p = (int)(t % hc);
    const int b = (int)(t / hc);
    const size_t base = (((size_t)b * h + 2 * p) * w + 2 * q) * c + ch;
    if (p < ho && q < wo) {
      const size_t win = ((size_t)b * ho + p) * wo + q;
      const int arg = argmax[win * c + ch];
      float g = (float)dy[win * lddy + ch];
      if (relu_mask && !((float)y[win * ldy + ch] > 0.f)) g = 0.f;
      dx[base] = arg == 0 ? g : 0.f;
      dx[base + c] = arg == 1 ? g : 0.f;
      dx[base + (size_t)w * c] = arg == 2 ? g : 0.f;
      dx[base + (size_t)w * c + c] = arg == 3 ? g : 0.f;
    } else {
      const bool has_r = 2 * p + 1 < h, has_c = 2 * q + 1 < w;
      dx[base] = 0.f;
      if (has_c) dx[base + c] = 0.f;
      if (has_r) dx[base + (size_t)w * c] = 0.f;
      if (has_r && has_c) dx[base + (size_t)w * c + c] = 0.f;
    }
  }
}
// dst[r][c] = src[r][c] for c < cols (float32 or bf16 on either side, round to nearest even), dst[r][c] = 0 for
// cols <= c < ld_dst: row pitches change (a padded bf16 copy of a matrix whose rows are not whole 16-byte pieces, and back)
template <bool S16, bool D16>
__global__ __launch_bounds__(256) void cast_rows_kernel(const void* __restrict__ src, void* __restrict__ dst, size_t rows, int cols,
                                                        int ld_src, int ld_dst) {
  const size_t total = rows * (size_t)ld_dst;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t r = i / (size_t)ld_dst;
    const int c = (int)(i - r * (size_t)ld_dst);
    float v = 0.f;
    if (c < cols) v = S16 ? (float)static_cast<const __bf16*>(src)[r * ld_src + c] : static_cast<const float*>(src)[r * ld_src + c];
    if (D16) static_cast<__bf16*>(dst)[i] = (__bf16)v;
    else static_cast<float*>(dst)[i] = v;
  }
}
// The same with a bf16 dx (config 5: the conv stack's activation gradients are bf16 tensors), eight channels per thread:
// 8 argmax bytes, 16 bytes of pooled values and of dy in, four 16-byte pieces of dx out.
__global__ __launch_bounds__(256) void maxpool_bwd_idx_bf16s_kernel(const uint8_t* __restrict__ argmax,
                                                                    const __bf16* __restrict__ y, const __bf16* __restrict__ dy,
                                                                    __bf16* __restrict__ dx, int n, int h, int w, int c, int ho,
                                                                    int wo, int ldy, int lddy, int relu_mask) {
  const int hc = (h + 1) / 2, wc = (w + 1) / 2, c8 = c / 8;
  const size_t total = (size_t)n * hc * wc * c8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % c8) * 8;
    size_t t = i / c8;
    const int q = (int)(t % wc);
    t /= wc;
    const int p = (int)(t % hc);
    const int b = (int)(t / hc);
    const size_t base = (((size_t)b * h + 2 * p) * w + 2 * q) * c + ch;
    const bool has_r = 2 * p + 1 < h, has_c = 2 * q + 1 < w;
    u32x4 o[4] = {u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}, u32x4{0, 0, 0, 0}};
    if (p < ho && q < wo) {
      const size_t win = ((size_t)b * ho + p) * wo + q;
      const uint2 a8 = *reinterpret_cast<const uint2*>(argmax + win * c + ch);
      const u32x4 yv = *reinterpret_cast<const u32x4*>(y + win * ldy + ch);
      const u32x4 gv = *reinterpret_cast<const u32x4*>(dy + win * lddy + ch);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t sh = (e & 1) ? 0xffff0000u : 0x0000ffffu;
        const uint32_t yb = (e & 1) ? (yv[e >> 1] & 0xffff0000u) : (yv[e >> 1] << 16);
        uint32_t g = gv[e >> 1] & sh;                                   // the gradient's 16 bits, in place
        if (relu_mask && !(__uint_as_float(yb) > 0.f)) g = 0u;
        const uint32_t arg = ((e < 4 ? a8.x : a8.y) >> (8 * (e & 3))) & 0xffu;
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (arg == (uint32_t)k) o[k][e >> 1] |= g;
      }
    }
    *reinterpret_cast<u32x4*>(dx + base) = o[0];
    if (has_c) *reinterpret_cast<u32x4*>(dx + base + c) = o[1];
    if (has_r) *reinterpret_cast<u32x4*>(dx + base + (size_t)w * c) = o[2];
    if (has_r && has_c) *reinterpret_cast<u32x4*>(dx + base + (size_t)w * c + c) = o[3];
  }
}
}  // namespace a3d

using namespace a3d;

extern "C" {

int a3d_maxpool2x2_fwd(int n, int h, int w, int c, const float* x, float* y, int ldy, const float* extra,
                       void* stream) {
  A3D_CHECK_ARG(n > 0 && h >= 2 && w >= 2 && c > 0 && x && y, "maxpool_fwd: bad arguments");
  A3D_CHECK_ARG(ldy >= c + (extra ? 1 : 0), "maxpool_fwd: ldy %d too small", ldy);
  const int ho = h / 2, wo = w / 2;
  const size_t total = (size_t)n * ho * wo * (c + (extra ? 1 : 0));
  clear_stale_error();
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y,
                     extra, n, h, w, c, ho, wo, ldy);
  return check_launch("maxpool_fwd");
}

int a3d_maxpool2x2_bwd(int n, int h, int w, int c, const float* x, const float* dy, int lddy, float* dx,
                       int relu_mask, void* stream) {
  A3D_CHECK_ARG(n > 0 && h >= 2 && w >= 2 && c > 0 && x && dy && dx && lddy >= c, "maxpool_bwd: bad arguments");
  const size_t total = (size_t)n * ((h + 1) / 2) * ((w + 1) / 2) * c;
  clear_stale_error();
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), x, dy,
                     dx, n, h, w, c, h / 2, w / 2, lddy, relu_mask);
  return check_launch("maxpool_bwd");
}

int a3d_stream_create(int level, void** stream) {
  A3D_CHECK_ARG(stream != nullptr, "stream_create: null output");
  int least = 0, greatest = 0;                        // numerically: greatest priority <= least priority
  if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return set_error(A3D_ELAUNCH, "stream_create: no priority range");
  const int prio = std::min(least, std::max(greatest, level));
  hipStream_t st = nullptr;
  if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio) != hipSuccess)
    return set_error(A3D_ELAUNCH, "stream_create: hipStreamCreateWithPriority failed");
  *stream = st;
  return A3D_OK;
}

int a3d_stream_destroy(void* stream) {
  A3D_CHECK_ARG(stream != nullptr, "stream_destroy: null stream");
  return hipStreamDestroy(static_cast<hipStream_t>(stream)) == hipSuccess ? A3D_OK : set_error(A3D_ELAUNCH, "stream_destroy failed");
}

int a3d_pad_channels_bf16(size_t pixels, int c_src, const float* src, int c_dst, void* dst, void* stream) {
  A3D_CHECK_ARG(pixels > 0 && src && dst && c_src >= 1 && c_dst == 4 && c_src <= 4, "pad_channels_bf16: 1..4 channels to 4");
  A3D_CHECK_ARG((reinterpret_cast<uintptr_t>(dst) & 7) == 0, "pad_channels_bf16: 8-byte aligned destination");
  clear_stale_error();
  hipLaunchKernelGGL(pad_channels_bf16_kernel, dim3(grid_for(pixels, 256, 4096)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     src, static_cast<__bf16*>(dst), pixels, c_src);
  return check_launch("pad_channels_bf16");
}

int a3d_cast_bf16(size_t count, const void* src, void* dst, int to_bf16, void* stream) {
  A3D_CHECK_ARG(count > 0 && src && dst, "cast_bf16: bad arguments");
  A3D_CHECK_ARG(((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0, "cast_bf16: 16-byte aligned buffers");
  clear_stale_error();
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(count / 4 + 1, 256, 4096)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, dst, count, to_bf16);
  return check_launch("cast_bf16");
}

int a3d_maxpool2x2_fwd_bf16(int n, int h, int w, int c, const void* x, int ldx, void* y, int ldy, const float* extra,
                            void* stream) {
  A3D_CHECK_ARG(n > 0 && h >= 2 && w >= 2 && c > 0 && x && y && ldx >= c && ldy >= c + (extra ? 1 : 0),
                "maxpool_fwd_bf16: bad arguments");
  const size_t total = (size_t)n * (h / 2) * (w / 2) * (c + (extra ? 1 : 0));
  clear_stale_error();
  hipLaunchKernelGGL(maxpool_fwd_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const __bf16*>(x), static_cast<__bf16*>(y), extra, n, h, w, c, h / 2, w / 2, ldx, ldy);
  return check_launch("maxpool_fwd_bf16");
}

int a3d_maxpool2x2_bwd_bf16(int n, int h, int w, int c, const void* x, int ldx, const void* dy, int lddy, void* dx,
                            int relu_mask, void* stream) {
  A3D_CHECK_ARG(n > 0 && h >= 2 && w >= 2 && c > 0 && x && dy && dx && lddy >= c && ldx >= c, "maxpool_bwd_bf16: bad arguments");
  const size_t total = (size_t)n * ((h + 1) / 2) * ((w + 1) / 2) * c;
  clear_stale_error();
  hipLaunchKernelGGL(maxpool_bwd_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const __bf16*>(x), static_cast<const __bf16*>(dy), static_cast<__bf16*>(dx), n, h, w, c,
                     h / 2, w / 2, ldx, lddy, relu_mask);
  return check_launch("maxpool_bwd_bf16");
}

static ResizeOne resize_one(int h, int w, int c, const float* x, int oh, int ow, float* y, int u8 = 0) {
  ResizeOne r;
  r.u8 = u8;
  r.x = x; r.y = y; r.h = h; r.w = w; r.c = c; r.oh = oh; r.ow = ow;
  r.sy = (float)h / (float)oh; r.sx = (float)w / (float)ow;
  return r;
}

int a3d_resize_bilinear_tf1(int n, int h, int w, int c, const float* x, int oh, int ow, float* y, void* stream) {
  A3D_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && oh > 0 && ow > 0 && x && y, "resize: bad arguments");
  ResizePair p;
  p.n = n;
  p.t[0] = p.t[1] = resize_one(h, w, c, x, oh, ow, y);
  clear_stale_error();
  hipLaunchKernelGGL(resize_kernel, dim3((unsigned)std::min(n * oh, 16384), 1), dim3(256), 0, static_cast<hipStream_t>(stream), p);
  return check_launch("resize");
}

int a3d_resize_bilinear_tf1_pair(int n, int h, int w, int c0, const float* x0, int oh0, int ow0, float* y0, int c1,
                                 const float* x1, int oh1, int ow1, float* y1, void* stream) {
  A3D_CHECK_ARG(n > 0 && h > 0 && w > 0 && c0 > 0 && c1 > 0 && oh0 > 0 && ow0 > 0 && oh1 > 0 && ow1 > 0 && x0 && y0 && x1 && y1,
                "resize_pair: bad arguments");
  ResizePair p;
  p.n = n;
  p.t[0] = resize_one(h, w, c0, x0, oh0, ow0, y0);
  p.t[1] = resize_one(h, w, c1, x1, oh1, ow1, y1);
  clear_stale_error();
  hipLaunchKernelGGL(resize_kernel, dim3((unsigned)std::min(n * std::max(oh0, oh1), 16384), 2), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  return check_launch("resize_pair");
}

int a3d_resize_bilinear_tf1_ex(int n, int h, int w, int c0, const void* x0, int u8_0, int oh0, int ow0, float* y0, int c1,
                               const void* x1, int u8_1, int oh1, int ow1, float* y1, void* stream) {
  A3D_CHECK_ARG(n > 0 && h > 0 && w > 0 && c0 > 0 && oh0 > 0 && ow0 > 0 && x0 && y0, "resize_ex: bad arguments");
  A3D_CHECK_ARG(!x1 || (c1 > 0 && oh1 > 0 && ow1 > 0 && y1), "resize_ex: bad second tensor");
  ResizePair p;
  p.n = n;
  p.t[0] = resize_one(h, w, c0, static_cast<const float*>(x0), oh0, ow0, y0, u8_0 ? 1 : 0);
  p.t[1] = x1 ? resize_one(h, w, c1, static_cast<const float*>(x1), oh1, ow1, y1, u8_1 ? 1 : 0) : p.t[0];
  clear_stale_error();
  hipLaunchKernelGGL(resize_kernel, dim3((unsigned)std::min(n * std::max(oh0, x1 ? oh1 : oh0), 16384), x1 ? 2 : 1), dim3(256), 0,
                     static_cast<hipStream_t>(stream), p);
  return check_launch("resize_ex");
}

int a3d_extract_patches(int n, int h, int w, int c, const float* x, int k, int stride, float* y, void* stream) {
  A3D_CHECK_ARG(n > 0 && h > 0 && w > 0 && c > 0 && k > 0 && stride > 0 && x && y, "patches: bad arguments");
  const int ph = (h + stride - 1) / stride, pw = (w + stride - 1) / stride;
  const int pad_h = std::max((ph - 1) * stride + k - h, 0), pad_w = std::max((pw - 1) * stride + k - w, 0);
  const size_t total = (size_t)n * ph * pw * k * k * c;
  clear_stale_error();
  hipLaunchKernelGGL(patches_kernel, dim3(grid_for(total, 256, 16384)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, y, n, h, w, c, k, stride, ph, pw, pad_h / 2, pad_w / 2);
  return check_launch("patches");
}

static const float kSilogC = (float)(0.5 / (74 * 55));   // src/models.py:269, folded constant

int a3d_silog_loss_fwd(int b, int npix, const float* out, const float* tgt, float* loss, float* ws, void* stream) {
  A3D_CHECK_ARG(b > 0 && npix > 0 && out && tgt && loss && ws, "silog_fwd: bad arguments");
  hipStream_t st = static_cast<hipStream_t>(stream);
  clear_stale_error();
  hipLaunchKernelGGL(silog_fwd_kernel, dim3(b * kSilogParts), dim3(256), 0, st, out, tgt, ws, loss, npix, b, kSilogC);
  return check_launch("silog_fwd");
}

int a3d_silog_loss_bwd(int b, int npix, const float* out, const float* tgt, const float* ws, float* dout,
                       void* stream) {
  A3D_CHECK_ARG(b > 0 && npix > 0 && out && tgt && ws && dout, "silog_bwd: bad arguments");
  const size_t total = (size_t)b * npix;
  clear_stale_error();
  hipLaunchKernelGGL(silog_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), out, tgt,
                     ws, dout, b, npix, kSilogC, 1.0f / (float)b);
  return check_launch("silog_bwd");
}

int a3d_silog_loss_bwd_ex(int b, int npix, const float* out, const float* tgt, const float* ws, float* dout, void* dout_bf16,
                          int ld_bf16, void* stream) {
  A3D_CHECK_ARG(b > 0 && npix > 0 && out && tgt && ws && dout && (!dout_bf16 || ld_bf16 >= npix), "silog_bwd: bad arguments");
  const size_t total = (size_t)b * npix;
  clear_stale_error();
  hipLaunchKernelGGL(silog_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), out, tgt,
                     ws, dout, b, npix, kSilogC, 1.0f / (float)b, static_cast<__bf16*>(dout_bf16), ld_bf16);
  return check_launch("silog_bwd");
}

int a3d_dropout_keep_mask(size_t count, uint64_t seed, uint64_t step, float rate, uint8_t* keep, void* stream) {
  A3D_CHECK_ARG(count > 0 && keep && rate >= 0.f && rate < 1.f, "dropout_keep_mask: bad arguments");
  clear_stale_error();
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for((count + 3) / 4)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), keep, count, (uint32_t)seed, (uint32_t)(seed >> 32),
                     (uint32_t)step, (uint32_t)(step >> 32), 1.0f - rate);
  return check_launch("dropout_mask");
}

int a3d_adam_apply_tf1(size_t count, float* var, float* m, float* v, const float* g, float lr, float beta1,
                       float beta2, float eps, float beta1_power, float beta2_power, float grad_scale,
                       void* stream) {
  return a3d_adam_apply_tf1_flag(count, var, m, v, g, lr, beta1, beta2, eps, beta1_power, beta2_power, grad_scale,
                                 nullptr, stream);
}

int a3d_adam_apply_tf1_flag(size_t count, float* var, float* m, float* v, const float* g, float lr, float beta1,
                            float beta2, float eps, float beta1_power, float beta2_power, float grad_scale,
                            unsigned int* poisoned, void* stream) {
  A3D_CHECK_ARG(count > 0 && var && m && v && g, "adam: bad arguments");
  A3D_CHECK_ARG(((reinterpret_cast<uintptr_t>(var) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
                  reinterpret_cast<uintptr_t>(g)) & 15) == 0, "adam: buffers must be 16-byte aligned");
  const float alpha = lr * sqrtf(1.f - beta2_power) / (1.f - beta1_power);
  if (alpha == 0.f && 1.f - beta2 == 0.f) {
    clear_stale_error();
    hipLaunchKernelGGL(adam_frozen_kernel, dim3(grid_for(count / 4 + 1, 256, 4096)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), var, m, v, g, count, 1.f - beta1, grad_scale, poisoned);
    return check_launch("adam_frozen");
  }
  clear_stale_error();
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(count / 4 + 1, 256, 4096)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), var, m, v, g, count, 1.f - beta1, 1.f - beta2, alpha, eps,
                     grad_scale, poisoned);
  return check_launch("adam");
}

int a3d_comm_standin(const float* src, size_t read_bytes, float* dst, size_t write_bytes, int workgroups, float gbytes_per_s,
                     void* stream) {
  A3D_CHECK_ARG(src && dst && read_bytes >= 16 && write_bytes >= 16 && workgroups >= 1 && workgroups <= 256 && gbytes_per_s > 0.f &&
                    ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0,
                "comm_standin: bad arguments");
  // bytes per 10-ns tick of s_memrealtime: both directions count against the rate
  const float f4_per_tick = gbytes_per_s * 10.f / 16.f * ((float)read_bytes / (float)(read_bytes + write_bytes));
  clear_stale_error();
  hipLaunchKernelGGL(comm_standin_kernel, dim3(workgroups), dim3(256), 0, static_cast<hipStream_t>(stream), src, read_bytes / 16,
                     dst, write_bytes / 16, f4_per_tick);
  return check_launch("comm_standin");
}

int a3d_copy_channel(size_t npix, const float* src, int ld_src, int c_src, float* dst, int ld_dst, int c_dst,
                     void* stream) {
  A3D_CHECK_ARG(npix > 0 && src && dst && c_src >= 0 && c_src < ld_src && c_dst >= 0 && c_dst < ld_dst,
                "copy_channel: bad arguments");
  clear_stale_error();
  hipLaunchKernelGGL(copy_channel_kernel, dim3((unsigned)std::min<size_t>((npix + 255) / 256, 2048)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, dst, npix, ld_src, c_src, ld_dst, c_dst);
  return check_launch("copy_channel");
}

int a3d_copy_channel_bf16(size_t npix, const float* src, int ld_src, int c_src, void* dst, int ld_dst, int c_dst, void* stream) {
  A3D_CHECK_ARG(npix > 0 && src && dst && c_src >= 0 && c_src < ld_src && c_dst >= 0 && c_dst < ld_dst,
                "copy_channel_bf16: bad arguments");
  clear_stale_error();
  hipLaunchKernelGGL(copy_channel_bf16_kernel, dim3((unsigned)std::min<size_t>((npix + 255) / 256, 2048)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), src, static_cast<__bf16*>(dst), npix, ld_src, c_src, ld_dst, c_dst);
  return check_launch("copy_channel_bf16");
}

int a3d_maxpool2x2_bwd_idx_bf16(int n, int h, int w, int c, const uint8_t* argmax, const void* y, int ldy, const void* dy,
                                int lddy, float* dx, int relu_mask, void* stream) {
  A3D_CHECK_ARG(n > 0 && h >= 2 && w >= 2 && c > 0 && argmax && y && dy && dx && ldy >= c && lddy >= c,
                "maxpool_bwd_idx_bf16: bad arguments");
  const size_t total = (size_t)n * ((h + 1) / 2) * ((w + 1) / 2) * c;
  clear_stale_error();
  hipLaunchKernelGGL(maxpool_bwd_idx_bf16_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     argmax, static_cast<const __bf16*>(y), static_cast<const __bf16*>(dy), dx, n, h, w, c, h / 2, w / 2, ldy,
                     lddy, relu_mask);
  return check_launch("maxpool_bwd_idx_bf16");
}

int a3d_cast_rows(size_t rows, int cols, const void* src, int ld_src, int src_bf16, void* dst, int ld_dst, int dst_bf16,
                  void* stream) {
  A3D_CHECK_ARG(rows > 0 && cols > 0 && src && dst && ld_src >= cols && ld_dst >= cols, "cast_rows: bad arguments");
  const size_t total = rows * (size_t)ld_dst;
  hipStream_t st = static_cast<hipStream_t>(stream);
  clear_stale_error();
  const dim3 grid(grid_for(total)), block(256);
  if (src_bf16 && dst_bf16) hipLaunchKernelGGL((cast_rows_kernel<true, true>), grid, block, 0, st, src, dst, rows, cols, ld_src, ld_dst);
  else if (src_bf16) hipLaunchKernelGGL((cast_rows_kernel<true, false>), grid, block, 0, st, src, dst, rows, cols, ld_src, ld_dst);
  else if (dst_bf16) hipLaunchKernelGGL((cast_rows_kernel<false, true>), grid, block, 0, st, src, dst, rows, cols, ld_src, ld_dst);
  else hipLaunchKernelGGL((cast_rows_kernel<false, false>), grid, block, 0, st, src, dst, rows, cols, ld_src, ld_dst);
  return check_launch("cast_rows");
}

int a3d_maxpool2x2_bwd_idx_bf16s(int n, int h, int w, int c, const uint8_t* argmax, const void* y, int ldy, const void* dy,
                                 int lddy, void* dx, int relu_mask, void* stream) {
  A3D_CHECK_ARG(n > 0 && h >= 2 && w >= 2 && c > 0 && argmax && y && dy && dx && ldy >= c && lddy >= c,
                "maxpool_bwd_idx_bf16s: bad arguments");
  const uintptr_t al = reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dx);
  A3D_CHECK_ARG(c % 8 == 0 && ldy % 8 == 0 && lddy % 8 == 0 && (al & 15) == 0 && (reinterpret_cast<uintptr_t>(argmax) & 7) == 0,
                "maxpool_bwd_idx_bf16s: channels and pixel strides in whole 16-byte pieces");
  const size_t total = (size_t)n * ((h + 1) / 2) * ((w + 1) / 2) * (c / 8);
  clear_stale_error();
  hipLaunchKernelGGL(maxpool_bwd_idx_bf16s_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     argmax, static_cast<const __bf16*>(y), static_cast<const __bf16*>(dy), static_cast<__bf16*>(dx), n, h, w, c,
                     h / 2, w / 2, ldy, lddy, relu_mask);
  return check_launch("maxpool_bwd_idx_bf16s");
}

int a3d_maxpool2x2_bwd_idx(int n, int h, int w, int c, const uint8_t* argmax, const float* y, int ldy, const float* dy,
                           int lddy, float* dx, int relu_mask, void* stream) {
  A3D_CHECK_ARG(n > 0 && h >= 2 && w >= 2 && c > 0 && argmax && y && dy && dx && ldy >= c && lddy >= c,
                "maxpool_bwd_idx: bad arguments");
  const size_t total = (size_t)n * ((h + 1) / 2) * ((w + 1) / 2) * c;
  const uintptr_t al = reinterpret_cast<uintptr_t>(argmax) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(dy) |
                       reinterpret_cast<uintptr_t>(dx);
  if (c % 4 == 0 && ldy % 4 == 0 && lddy % 4 == 0 && (al & 15) == 0 && total / 4 < (1u << 31)) {
    clear_stale_error();
    hipLaunchKernelGGL(maxpool_bwd_idx_vec4_kernel, dim3(grid_for(total / 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       argmax, y, dy, dx, n, h, w, c, h / 2, w / 2, ldy, lddy, relu_mask);
    return check_launch("maxpool_bwd_idx_vec4");
  }
  clear_stale_error();
  hipLaunchKernelGGL(maxpool_bwd_idx_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     argmax, y, dy, dx, n, h, w, c, h / 2, w / 2, ldy, lddy, relu_mask);
  return check_launch("maxpool_bwd_idx");
}

}  // extern "C"
