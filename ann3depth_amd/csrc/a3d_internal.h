// a3d_internal.h — shared between the translation units of liba3d.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/a3d.h"

namespace a3d {

int set_error(int code, const char* fmt, ...);

#define A3D_CHECK_ARG(cond, ...)                                 \
  do {                                                           \
    if (!(cond)) return ::a3d::set_error(A3D_EINVAL, __VA_ARGS__); \
  } while (0)

// hipGetLastError() is sticky across unrelated runtime calls made by the process (PyTorch probes devices, etc.):
// clear it before launching so check_launch() reports only this launch.
inline void clear_stale_error() { (void)hipGetLastError(); }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(A3D_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return A3D_OK;
}

// Environment switches (capi.cc).  tune_int: A/B and sweep switches — the environment is consulted ONLY in a process started
// with A3D_TUNING=1 (the tools under tools/ set it); anywhere else the default is returned and nothing reads the environment.
bool tuning();
int tune_int(const char* name, int dflt);

// ---- implicit-GEMM front end (igemm_host.hip) ----
struct GemmPlan {
  int prec;          // A3D_PREC_*: 0 = fp32 kernel (cfg valid), else bf16 kernel (bf16_bn valid)
  int bf16_bn;       // 128 or 64
  int cfg;           // index into the config table
  int splitk;
  int ktiles_per_split;
  int tiles_m, tiles_n;
  int streamk;       // > 0: stream-K launch of this many blocks (splitk == 1)
  int sk_sliced;     // stream-K shares cut per XCD from eighths of the K axis (bwd-filter; SkSpace in igemm.h)
  size_t ws_bytes;   // split-K / stream-K slabs (0 if neither)
  int ring;          // bf16 plans: 1 + tile configuration of the LDS-DMA kernel for bf16-stored operands (igemm_ring.h), 0 = igemm_bf16
};

struct GemmProblem {
  int mode;          // MODE_*
  int M, N, K;
  int avec, bvec;    // 1 or 4
  int plain = 0;     // 1: register-staged kernel without split-K only (fused-pool forward)
  int need_reduce = 0;   // 1: the output is stored by the split-K reduction only (rows narrower than the GEMM's N): split-K >= 2, no stream-K
  int no_glds = 0;   // 1: not the LDS-DMA kernels (bf16 output)
  int gen2_ok = 0;   // 1: the launch may run on the second-generation LDS-DMA kernel (igemm2.h): float32 tensors, 16-byte
                     //    operands, forward / bwd-data: gathered channels a multiple of 32 and K = taps x channels
  int ring_ok = 0;   // 1: both operands are bf16 tensors whose 16-byte pieces lie inside one filter tap (channels % 8 == 0):
                     //    forward / stride-1 bwd-data may run on igemm_ring.h
};

GemmPlan plan_gemm(const GemmProblem& g, int precision = 0);

struct IgemmParams;
int launch_igemm(int mode, const GemmPlan& plan, int avec, int bvec, IgemmParams& p, void* ws, hipStream_t st);

// ---- weight-streaming dense kernels for batches of at most 64 rows (dense.hip) ----
bool dense_dw_applicable(int m, int k, int n);
bool dense_stream_applicable(int m, int k, int n);
size_t dense_stream_ws_bytes(int m, int k, int n);
int dense_fwd_stream(int m, int k, int n, const float* x, const float* w, const float* bias, float* y, int act,
                     const uint8_t* keep, float keep_scale, void* ws, size_t ws_bytes, hipStream_t st);
struct ReduceParams;
int launch_splitk_reduce(const ReduceParams& r, hipStream_t st);
int dense_dw_launch(int m, int k, int n, const float* x, const float* dz, float* dw, float* db, hipStream_t st);

// ---- BiasAddGrad of a bf16 gradient tensor beside the LDS-DMA bwd-filter (igemm_ring.hip) ----
size_t colsum_bf16_ws_bytes(int n);
bool colsum_bf16_ok(int n);
int colsum_bf16(const void* dz, int rows, int n, int ld, float* out, void* ws, int* nparts, hipStream_t st);

// ---- single-output-channel 5x5 stencil (stencil1.hip) ----
bool stencil1_applicable(const a3d_conv_desc* d);
size_t stencil1_bwdf_ws_bytes(const a3d_conv_desc* d);
int stencil1_fwd(const a3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int act,
                 hipStream_t st);
int stencil1_bwd_filter(const a3d_conv_desc* d, const float* x, const float* dz, float* dw, float* db, void* ws,
                        hipStream_t st);
bool stencil1_bwd_both_applicable(const a3d_conv_desc* d);
size_t stencil1_bwd_both_ws_bytes(const a3d_conv_desc* d);
int stencil1_bwd_both(const a3d_conv_desc* d, const float* x, const float* dz, const float* w, float* dw, float* db,
                      void* dx, int lddx, int dx_bf16, int relu_mask, unsigned* state, void* ws, hipStream_t st);

// ---- few-channel filter gradient from LDS-staged input rows, optionally with the max pool's gradient fused (fewch.hip) ----
bool fewch_bwdf_applicable(const a3d_conv_desc* d, bool pooled);
size_t fewch_bwdf_ws_bytes(const a3d_conv_desc* d, bool pooled);
bool fewch_extents_ok(const a3d_conv_desc* d, bool pooled, int ldz, int esz, int ld_arg);
int fewch_bwd_filter(const a3d_conv_desc* d, const float* x, int src, const void* dz, int ldz, const void* pooled_act,
                     const uint8_t* argmax, int ld_arg, float* dw, float* db, void* ws, hipStream_t st);

int fewch_reduce_launch(const float* slabs, int splits, int Mp, int NP, int M, int N, float* dw, float* db, hipStream_t st);
// ... on the bf16 matrix cores (float32 image, bf16 gradient tensors: config 5; fewch16.hip)
bool fewch16_bwdf_applicable(const a3d_conv_desc* d, bool pooled);
size_t fewch16_bwdf_ws_bytes(const a3d_conv_desc* d, bool pooled);
int fewch16_bwd_filter(const a3d_conv_desc* d, const float* x, bool pooled, const void* dz, int ldz, const void* pooled_act,
                       const uint8_t* argmax, int ld_arg, float* dw, float* db, void* ws, hipStream_t st);

// ---- few-channel forward convolution straight from L2 (conv3.hip) ----
bool conv3_applicable(const a3d_conv_desc* d, const void* x);
size_t conv3_ws_bytes(const a3d_conv_desc* d);
int conv3_pack(const a3d_conv_desc* d, const float* w, float* wp, hipStream_t st);
int conv3_fwd(const a3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int act, int pool,
              int ld_out, uint8_t* argmax, void* ws, size_t ws_bytes, hipStream_t st, bool prepared = false);

// ---- the same from a 4-channel bf16 image on the bf16 matrix cores (conv3.hip, conv3b_*) ----
bool conv3b_applicable(const a3d_conv_desc* d);
size_t conv3b_filter_bytes(const a3d_conv_desc* d);
int conv3b_pack(const a3d_conv_desc* d, const float* w, void* wp, hipStream_t st);
int conv3b_fwd(const a3d_conv_desc* d, const void* x, const float* w, const float* bias, void* y, int act, int pool, int ld_out,
               uint8_t* argmax, void* ws, size_t ws_bytes, hipStream_t st, bool prepared);

}  // namespace a3d
