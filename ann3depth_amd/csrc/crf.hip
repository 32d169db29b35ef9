// crf.hip — the pairwise part and the CRF negative log-likelihood of ann3depth's DCNF model
// (src/models.py:20-48,91-177): 40x40 "superpixel" statistics, pair similarities, the 48x48 system A = I + D - R per
// image (LU with partial pivoting in LDS, one wavefront per image), the loss and its gradient wrt the unary output z.
// A is a constant for the gradient: TF 1.3 registers no gradient for scatter_nd_update (oracle/dcnf.py states the
// assumption).  All of it is tiny next to the unary conv stack; the kernels are written for clarity, not speed.
#include <algorithm>

#include "a3d_internal.h"

namespace a3d {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum_f(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float s = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return s;
}

// mean over each sp x sp block: x [n,h,w,c] -> out [n, (h/sp)*(w/sp), c]          (reduce_mean(superpixels, axis=2))
__global__ __launch_bounds__(256) void superpixel_mean_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                              int h, int w, int c, int sp) {
  __shared__ float red[4];
  const int cols = w / sp, rows = h / sp;
  const int p = blockIdx.x % (rows * cols), b = blockIdx.x / (rows * cols);
  const int pr = p / cols, pc = p % cols;
  for (int ch = 0; ch < c; ++ch) {
    float s = 0.f;
    for (int i = threadIdx.x; i < sp * sp; i += 256) {
      const int yy = pr * sp + i / sp, xx = pc * sp + i % sp;
      s += x[(((size_t)b * h + yy) * w + xx) * c + ch];
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) out[((size_t)b * rows * cols + p) * c + ch] = s / (float)(sp * sp);
  }
}

// color_histogram (src/models.py:95-100): 256 bins of r*2^24 + g*2^16 + b*2^8 over [0, 2^24)
__global__ __launch_bounds__(256) void superpixel_hist_kernel(const float* __restrict__ x, float* __restrict__ hist,
                                                              int h, int w, int sp) {
  __shared__ int bins[256];
  const int cols = w / sp, rows = h / sp;
  const int p = blockIdx.x % (rows * cols), b = blockIdx.x / (rows * cols);
  const int pr = p / cols, pc = p % cols;
  bins[threadIdx.x] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < sp * sp; i += 256) {
    const int yy = pr * sp + i / sp, xx = pc * sp + i % sp;
    const float* px = x + (((size_t)b * h + yy) * w + xx) * 3;
    const float v = __fadd_rn(__fadd_rn(__fmul_rn(px[0], 16777216.f), __fmul_rn(px[1], 65536.f)), __fmul_rn(px[2], 256.f));
    const float scaled = __fdiv_rn(v, 16777216.f);
    int idx = (int)floorf(__fmul_rn(256.f, scaled));
    idx = min(max(idx, 0), 255);
    atomicAdd(&bins[idx], 1);
  }
  __syncthreads();
  hist[((size_t)b * rows * cols + p) * 256 + threadIdx.x] = (float)bins[threadIdx.x];
}

// similarity() of both feature kinds for one (image, pair) + the pairwise dense layer (2 -> 1)
__global__ __launch_bounds__(256) void pair_similarity_kernel(const float* __restrict__ x, const float* __restrict__ hist,
                                                              const int* __restrict__ left, const int* __restrict__ right,
                                                              const float* __restrict__ dw, const float* __restrict__ db,
                                                              float* __restrict__ sims, float* __restrict__ r, int h,
                                                              int w, int sp, int npairs, float gamma) {
  __shared__ float red[4];
  const int cols = w / sp, nsp = (h / sp) * cols;
  const int q = blockIdx.x % npairs, b = blockIdx.x / npairs;
  const int pl = left[q], pr = right[q];
  float sc = 0.f;
  for (int i = threadIdx.x; i < sp * sp; i += 256) {
    const int dy = i / sp, dx = i % sp;
    const float* a = x + (((size_t)b * h + (pl / cols) * sp + dy) * w + (pl % cols) * sp + dx) * 3;
    const float* c = x + (((size_t)b * h + (pr / cols) * sp + dy) * w + (pr % cols) * sp + dx) * 3;
    const float ga = (a[0] + a[1] + a[2]) / 3.f, gc = (c[0] + c[1] + c[2]) / 3.f;     // reduce_mean over channels
    const float d = ga - gc;
    sc += d * d;
  }
  sc = block_sum_256(sc, red);
  const float* hl = hist + ((size_t)b * nsp + pl) * 256;
  const float* hr = hist + ((size_t)b * nsp + pr) * 256;
  const float dh = hl[threadIdx.x] - hr[threadIdx.x];
  const float sh = block_sum_256(dh * dh, red);
  if (threadIdx.x == 0) {
    const float cdiff = expf(-gamma * sqrtf(sc)), hdiff = expf(-gamma * sqrtf(sh));
    const size_t o = (size_t)b * npairs + q;
    sims[2 * o] = cdiff;
    sims[2 * o + 1] = hdiff;
    r[o] = cdiff * dw[0] + hdiff * dw[1] + db[0];
  }
}

// One wavefront per image.  A = I + D - R from the pair weights, LU with partial pivoting on [A | z] in LDS:
// det(A) = +-prod(pivots), w = A^-1 z by back substitution; then energy, partition function, loss, d loss / d z.
constexpr int kMaxSp = 64;
__global__ __launch_bounds__(64) void crf_loss_kernel(const float* __restrict__ z, const float* __restrict__ y,
                                                      const float* __restrict__ r, const int* __restrict__ left,
                                                      const int* __restrict__ right, float* __restrict__ loss_img,
                                                      float* __restrict__ dz, int n, int npairs, float eps, float fac0,
                                                      float inv_batch) {
  __shared__ float A[kMaxSp][kMaxSp + 1];
  __shared__ float U[kMaxSp][kMaxSp + 2];      // working copy, column n holds the right-hand side
  __shared__ float wv[kMaxSp];
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < n * n; i += 64) A[i / n][i % n] = 0.f;
  __syncthreads();
  if (lane == 0) {                              // scatter the pair weights: R[l][r] = R[r][l] = r_q (no duplicates)
    for (int q = 0; q < npairs; ++q) {
      const float v = r[(size_t)b * npairs + q];
      A[left[q]][right[q]] = v;
      A[right[q]][left[q]] = v;
    }
  }
  __syncthreads();
  const float zi = lane < n ? z[(size_t)b * n + lane] : 0.f;
  const float yi = lane < n ? y[(size_t)b * n + lane] : 0.f;
  if (lane < n) {                               // A = I + diag(row sums of R) - R
    float rs = 0.f;
    for (int j = 0; j < n; ++j) rs += A[lane][j];
    for (int j = 0; j < n; ++j) A[lane][j] = (j == lane ? 1.f + rs : 0.f) - A[lane][j];
    for (int j = 0; j < n; ++j) U[lane][j] = A[lane][j];
    U[lane][n] = zi;
  }
  __syncthreads();
  // energy = y^T A y - 2 z^T y + z^T z
  float ay = 0.f;
  if (lane < n)
    for (int j = 0; j < n; ++j) ay += A[lane][j] * y[(size_t)b * n + j];
  const float yAy = wave_sum_f(yi * ay), zy = wave_sum_f(zi * yi), zz = wave_sum_f(zi * zi), zsum = wave_sum_f(zi);
  const float energy = yAy - 2.f * zy + zz;
  // LU, one lane per row
  float det = 1.f;
  for (int k = 0; k < n; ++k) {
    float best = (lane >= k && lane < n) ? fabsf(U[lane][k]) : -1.f;
    int arg = lane;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ob = __shfl_xor(best, off, 64);
      const int oa = __shfl_xor(arg, off, 64);
      if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (arg != k) {
      for (int j = lane; j <= n; j += 64) { const float t = U[k][j]; U[k][j] = U[arg][j]; U[arg][j] = t; }
      det = -det;
    }
    __syncthreads();
    const float piv = U[k][k];
    det *= piv;
    if (lane > k && lane < n) {
      const float f = U[lane][k] / piv;
      for (int j = k; j <= n; ++j) U[lane][j] -= f * U[k][j];
    }
    __syncthreads();
  }
  if (lane == 0) {                              // back substitution (48 x 48: serial is fine)
    for (int i = n - 1; i >= 0; --i) {
      float s = U[i][n];
      for (int j = i + 1; j < n; ++j) s -= U[i][j] * wv[j];
      wv[i] = s / U[i][i];
    }
  }
  __syncthreads();
  const float wi = lane < n ? wv[lane] : 0.f;
  // g = z^T (A^-1 + eps) z - z^T z ;  inverseA = inv(A) + eps adds eps to EVERY element (src/models.py:165)
  const float zw = wave_sum_f(zi * wi);
  const float g = zw + eps * zsum * zsum - zz;
  const float fac = fac0 / (sqrtf(det) + eps);
  const float ex = expf(g);
  const float Z = fac * ex + eps;
  const float u = expf(-energy) / Z;
  if (lane == 0) loss_img[b] = -logf(u + eps);
  if (lane < n) {
    const float dE = -2.f * yi + 2.f * zi;
    const float dg = 2.f * wi + 2.f * eps * zsum - 2.f * zi;
    const float du = u * (-dE) - (u / Z) * (fac * ex * dg);
    dz[(size_t)b * n + lane] = (-du / (u + eps)) * inv_batch;
  }
}

__global__ __launch_bounds__(64) void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) s += v[i];
  s = wave_sum_f(s);
  if (threadIdx.x == 0) out[0] = s / (float)n;
}

// tf.train.GradientDescentOptimizer(lr) (src/models.py:198): var -= lr * g
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ var, const float* __restrict__ g, size_t count,
                                                  float lr) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256)
    var[i] = __fsub_rn(var[i], __fmul_rn(lr, g[i]));
}

}  // namespace a3d

using namespace a3d;

extern "C" {

int a3d_superpixel_mean(int n, int h, int w, int c, const float* x, int sp, float* out, void* stream) {
  A3D_CHECK_ARG(n > 0 && sp > 0 && h % sp == 0 && w % sp == 0 && c > 0 && x && out, "superpixel_mean: bad arguments");
  clear_stale_error();
  hipLaunchKernelGGL(superpixel_mean_kernel, dim3(n * (h / sp) * (w / sp)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, out, h, w, c, sp);
  return check_launch("superpixel_mean");
}

int a3d_superpixel_hist(int n, int h, int w, const float* x, int sp, float* hist, void* stream) {
  A3D_CHECK_ARG(n > 0 && sp > 0 && h % sp == 0 && w % sp == 0 && x && hist, "superpixel_hist: bad arguments");
  clear_stale_error();
  hipLaunchKernelGGL(superpixel_hist_kernel, dim3(n * (h / sp) * (w / sp)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, hist, h, w, sp);
  return check_launch("superpixel_hist");
}

int a3d_pair_similarity(int n, int h, int w, const float* x, int sp, const float* hist, const int32_t* left,
                        const int32_t* right, int npairs, const float* dense_w, const float* dense_b, float gamma,
                        float* sims, float* r, void* stream) {
  A3D_CHECK_ARG(n > 0 && sp > 0 && h % sp == 0 && w % sp == 0 && npairs > 0 && x && hist && left && right && dense_w &&
                    dense_b && sims && r, "pair_similarity: bad arguments");
  clear_stale_error();
  hipLaunchKernelGGL(pair_similarity_kernel, dim3(n * npairs), dim3(256), 0, static_cast<hipStream_t>(stream), x, hist,
                     left, right, dense_w, dense_b, sims, r, h, w, sp, npairs, gamma);
  return check_launch("pair_similarity");
}

int a3d_crf_loss(int n, int nsp, const float* z, const float* y, const float* r, const int32_t* left,
                 const int32_t* right, int npairs, float eps, float* loss_per_image, float* loss_mean, float* dz,
                 void* stream) {
  A3D_CHECK_ARG(n > 0 && nsp > 0 && nsp <= kMaxSp && npairs > 0 && z && y && r && left && right && loss_per_image &&
                    loss_mean && dz, "crf_loss: bad arguments (at most %d superpixels)", kMaxSp);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float fac0 = (float)pow(3.14159265358979323846, nsp / 2.0);
  clear_stale_error();
  hipLaunchKernelGGL(crf_loss_kernel, dim3(n), dim3(64), 0, st, z, y, r, left, right, loss_per_image, dz, nsp, npairs,
                     eps, fac0, 1.0f / (float)n);
  int rc = check_launch("crf_loss");
  if (rc != A3D_OK) return rc;
  clear_stale_error();
  hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(64), 0, st, loss_per_image, n, loss_mean);
  return check_launch("crf_loss_mean");
}

int a3d_sgd_apply(size_t count, float* var, const float* g, float lr, void* stream) {
  A3D_CHECK_ARG(count > 0 && var && g, "sgd: bad arguments");
  clear_stale_error();
  hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)std::min<size_t>((count + 255) / 256, 4096)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), var, g, count, lr);
  return check_launch("sgd");
}

}  // extern "C"
