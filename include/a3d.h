/* a3d.h — C ABI of liba3d.so: the MI355X (gfx950) kernels behind ann3depth's depth-regression training path.
 *
 * The reference (shoeffner/ann3depth) has no FFI of its own: its arithmetic is whatever TensorFlow-1.3 ops the
 * model functions in src/models.py instantiate.  Each entry point below replaces the TF op (or op group) created
 * by the cited reference call site; the Python host (ann3depth_amd/) is the only caller.
 *
 * Conventions
 *   - Layouts are TensorFlow's: activations NHWC, conv filters HWIO ([R][S][Cin][Cout]), dense kernels [in][out].
 *   - All tensors are float32 device pointers owned by the caller (PyTorch is only the allocator).  The library
 *     allocates no device memory and never synchronises: every compute call only enqueues work on the caller's
 *     hipStream_t (passed as void*), so it is stream-ordered and graph-capturable.
 *   - Process-wide state, all of it host-side: the thread-local error message; a cache of launch plans keyed by
 *     problem shape (mutex-guarded, never invalidated: a plan depends on nothing else); the launch-timing list of
 *     a3d_timing_enable / a3d_timing_collect, which is ONE list for every stream and thread — enable it from one
 *     place, and not during graph capture (it records hipEvents).  The environment: A3D_TUNING is read once; every other
 *     switch (A/B and sweep aids: INTEGRATION.md section 5) is read ONLY in a process started with A3D_TUNING=1 — without
 *     it the library never calls getenv and a plan depends on the problem alone.
 *   - `ws` is caller-provided scratch of at least the size the matching *_ws_bytes() query returns.
 *   - Return value: 0 (A3D_OK) or a negative A3D_E* code; a3d_last_error() gives a thread-local message.
 *     Nothing throws across the ABI and nothing calls exit().
 */
#ifndef A3D_H_
#define A3D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A3D_OK 0
#define A3D_EINVAL (-1)   /* bad argument / unsupported shape */
#define A3D_EWORKSPACE (-2) /* workspace too small */
#define A3D_ELAUNCH (-3)  /* hipLaunchKernel failed */
#define A3D_EFORMAT (-4)  /* corrupt TFRecord / Example */
#define A3D_EIO (-5)

#define A3D_ACT_NONE 0
#define A3D_ACT_RELU 1
#define A3D_ACT_SIGMOID 2

/* Geometry of one tf.layers.conv2d call (src/models.py:64-72,211-223,241-251).
 * pad_t/pad_l are TF's "before" paddings (SAME: pad_total//2; VALID: 0); ho/wo the output extent. */
typedef struct a3d_conv_desc {
  int32_t n, h, w, c;        /* input  [n,h,w,c]  */
  int32_t k, r, s;           /* filter [r,s,c,k]  */
  int32_t stride;            /* same for both axes; must be 1, 2 or 4 */
  int32_t pad_t, pad_l;
  int32_t ho, wo;            /* output [n,ho,wo,k] */
  int32_t ldx;               /* elements between consecutive input pixels  (>= c; c if dense-packed) */
  int32_t ldy;               /* elements between consecutive output pixels (>= k) */
  int32_t precision;         /* A3D_PREC_*: arithmetic of the contraction */
  int32_t storage;           /* A3D_STORE_* bits: which tensors are bf16 in memory (0: all float32); needs A3D_PREC_BF16 */
  int32_t hints;             /* A3D_HINT_* bits: scheduling advice, never changes a result */
} a3d_conv_desc;

/* The forward launch will run beside bandwidth-bound kernels of ANOTHER stream (the fine network's forward beside the
 * coarse network's dense layers: src/models.py:289-290 builds both in one graph): keep at most two wavefronts per SIMD
 * resident — for the 8-wave kernels one block per CU, which also leaves half of the CU's LDS unclaimed — so that the
 * other stream's kernels find free registers, wave slots and LDS on every CU instead of waiting for GEMM blocks to
 * retire.  The result is bit-identical with and without the hint. */
#define A3D_HINT_SHARE_CU 1
/* `w` of this FORWARD call is the filter as a3d_conv2d_fwd_prepare_filter laid it out for this descriptor (the few-channel
 * layers repack or pad their filter — [K/4][N][4], zero rows for the window runs' pad positions, a bf16 copy for the image
 * form — which otherwise happens on every call, 5 us per launch for weights that the reference's optimizer never moves:
 * src/models.py:309).  The caller refreshes the prepared copy whenever the weights change.  Same results bit for bit. */
#define A3D_HINT_W_PREPARED 2

/* BASELINE config 5 ("bf16 activations + bf16 weight copies, fp32 master and accumulate"): tensors marked here are bf16 in
 * HBM (pass their pointers through the float* parameters); channel counts and pixel strides of a bf16 tensor must be
 * multiples of 8 and its base 16-byte aligned.  Filter / bias gradients, split-K slabs and biases are always float32.
 *   forward      : X = x, W = w, Y = y
 *   bwd_data     : Y = dz, W = w, X = dx and relu_mask (the same activation tensor)
 *   bwd_filter   : X = x, Y = dz (dw, db float32)                                                     */
#define A3D_STORE_X_BF16 1
#define A3D_STORE_W_BF16 2
#define A3D_STORE_Y_BF16 4

#define A3D_PREC_F32 0     /* exact fp32 on the fp32 matrix cores (default; what parity is stated for) */
#define A3D_PREC_BF16X3 1  /* fp32 operands split into bf16 hi+lo, 3 bf16 MFMAs per product: ~1e-5 relative */
#define A3D_PREC_BF16 2    /* operands rounded to bf16, fp32 accumulate (BASELINE config 5's arithmetic; storage: a3d_conv_desc.storage) */

const char* a3d_version(void);
int a3d_last_error(char* buf, size_t len);
/* sizeof(a3d_conv_desc) as THIS build of the library reads it: a binding written against an older header (the struct has
 * grown: storage, hints) can compare it with the size of its own mirror of the struct before the first call. */
size_t a3d_sizeof_conv_desc(void);

/* A HIP stream of a given queue priority (hipStreamCreateWithPriority, non-blocking): `level` -1 = above, 0 = the same
 * as, +1 = below the streams the host framework hands out, clamped to what the device offers.  For the second stream of
 * a training step (src/models.py:289-305 builds coarse and fine in one graph; here the MFMA-bound fine forward runs
 * beside the HBM-bound dense stretch of the coarse chain): with the lower priority the dispatcher gives a freed CU slot
 * to the HBM-bound kernel first.  *stream is a hipStream_t for every `void* stream` argument of this header. */
int a3d_stream_create(int level, void** stream);
int a3d_stream_destroy(void* stream);

/* Conv2D + BiasAdd (+ Relu)  — tf.layers.conv2d forward.  bias may be NULL.
 * Non-finite inputs (the reference gives them meaning: src/models.py:262-264): with A3D_PREC_F32 an output is non-finite
 * only if its own receptive field holds a non-finite value, as with TensorFlow's Conv2D — taps outside the image, K / M
 * tails and the pad positions of the few-channel kernels' window runs all enter the matrix cores as zeros on the
 * ACTIVATION side.  One stated deviation, opt-in modes only: A3D_PREC_BF16 / _BF16X3 on an unpadded conv of <= 4 input
 * channels gather whole window runs whose pad positions hold the NEXT pixels of the image row and meet zero weights
 * (0 * inf = NaN): there a non-finite pixel can also reach the windows immediately to its left. */
size_t a3d_conv2d_fwd_ws_bytes(const a3d_conv_desc* d);
int a3d_conv2d_fwd(const a3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y,
                   int act, void* ws, size_t ws_bytes, void* stream);

/* The per-call filter repack of a3d_conv2d_fwd / a3d_conv2d_pool_fwd, hoisted (A3D_HINT_W_PREPARED): bytes of the prepared
 * form for this descriptor with the hint bit clear (0: this forward reads the filter as stored — the hint is refused), and the
 * repack itself into a caller-owned, 16-byte aligned buffer.  The descriptor's x is assumed 16-byte aligned. */
size_t a3d_conv2d_fwd_prepared_filter_bytes(const a3d_conv_desc* d);
int a3d_conv2d_fwd_prepare_filter(const a3d_conv_desc* d, const float* w, void* prepared, size_t prepared_bytes, void* stream);

/* tf.layers.max_pooling2d(tf.layers.conv2d(x, ..., activation), 2, 2) in one kernel (src/models.py:211-216,241-243):
 * y_pooled[n, ho/2, wo/2, k] (pixel stride ld_pooled >= k) = 2x2 / stride-2 VALID max pool of act(conv + bias); the conv
 * output itself is never written.  argmax (may be NULL): [n, ho/2, wo/2, k] bytes, the position 0..3 (row-major in the
 * window) of the first maximum — all that MaxPoolGrad + ReluGrad need besides the pooled value, so training does not
 * need the conv output either (a3d_maxpool2x2_bwd_idx).  Accepted: fp32 arithmetic and inputs; float32 x and w with a
 * bf16 pooled map (storage A3D_STORE_Y_BF16, fp32 or bf16 arithmetic: conv2d_0 of BASELINE config 5); the bf16 kernel's
 * image form (a 4-channel bf16 image, see a3d_pad_channels_bf16); bf16 x, w AND y with bf16 arithmetic (all three storage
 * bits; channels, pixel strides and ld_pooled multiples of 8, k a multiple of 16 when argmax is wanted: the LDS-DMA kernel,
 * conv2d_1 of config 5).  Wherever the output tensor is bf16 the values compared are the ones a separate conv would have
 * stored, i.e. rounded to bf16.  Same workspace as a3d_conv2d_fwd. */
int a3d_conv2d_pool_fwd(const a3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y_pooled,
                        int ld_pooled, uint8_t* argmax, int act, void* ws, size_t ws_bytes, void* stream);

/* Conv2DBackpropInput.  dz = gradient wrt the pre-activation output [n,ho,wo,k] (pixel stride ldy).
 * If relu_mask != NULL (same shape/stride as dx) the result is multiplied by (relu_mask > 0): this fuses the
 * ReluGrad of the PREVIOUS layer, whose output is this layer's input. */
size_t a3d_conv2d_bwd_data_ws_bytes(const a3d_conv_desc* d);
int a3d_conv2d_bwd_data(const a3d_conv_desc* d, const float* dz, const float* w, float* dx,
                        const float* relu_mask, void* ws, size_t ws_bytes, void* stream);

/* Conv2DBackpropFilter + BiasAddGrad.  dw [r,s,c,k]; db [k] may be NULL. */
size_t a3d_conv2d_bwd_filter_ws_bytes(const a3d_conv_desc* d);
int a3d_conv2d_bwd_filter(const a3d_conv_desc* d, const float* x, const float* dz, float* dw, float* db,
                          void* ws, size_t ws_bytes, void* stream);

/* MaxPoolGrad + ReluGrad + Conv2DBackpropFilter + BiasAddGrad of a conv -> ReLU -> 2x2 max pool block in one launch
 * (src/models.py:211-213 conv2d_0, :241-243 fine/first, :64-65 DCNF's first conv; the gradients asked for at :314,:198):
 * the gradient of the conv's pre-activation output is never materialised.  dpool / pooled: gradient wrt the POOLED map and
 * the pooled activation itself [n, ho/2, wo/2, .] with pixel stride ld_dpool (float32, or bf16 when pooled_bf16 != 0);
 * pooled == NULL: no ReluGrad.  argmax: the window positions a3d_conv2d_pool_fwd recorded, pixel stride ld_argmax.  A window
 * hands its gradient to the position of its maximum if that maximum is > 0, exactly as a3d_maxpool2x2_bwd_idx followed by
 * a3d_conv2d_bwd_filter.  Accepted: fp32 arithmetic, float32 x with <= 4 densely packed channels (ldx == c), no padding,
 * 33..96 filters (a3d_conv2d_bwd_filter_pooled_ws_bytes returns 0 otherwise).  Sums are taken in a fixed order. */
size_t a3d_conv2d_bwd_filter_pooled_ws_bytes(const a3d_conv_desc* d);
int a3d_conv2d_bwd_filter_pooled(const a3d_conv_desc* d, const float* x, const void* dpool, int ld_dpool, const void* pooled,
                                 const uint8_t* argmax, int ld_argmax, int pooled_bf16, float* dw, float* db, void* ws,
                                 size_t ws_bytes, void* stream);

/* Conv2DBackpropFilter + BiasAddGrad + Conv2DBackpropInput (+ the ReluGrad of the layer below) of a convolution with ONE
 * output channel, in one pass over x — fine/third (src/models.py:250-251; its gradients are asked for at :333-338 through
 * compute_gradients over the fine variables, the input gradient by the layer below): the same dz[q - tap + pad] enters
 * dw[tap][c] += x[q][c] * dz and dx[q][c] += w[tap][c] * dz, so x is read once (for the filter gradient AND as the ReLU mask
 * of dx when relu_mask != 0: dx *= (x > 0)) and dx written once, float32 or (dx_bf16 != 0) bf16, pixel stride lddx.
 * Supported (a3d_conv2d_bwd_both_supported): k = 1, 5x5, stride 1, an even c <= 64, even ldx, float32 x / dz / w, x below
 * 1 GiB; otherwise call a3d_conv2d_bwd_filter and a3d_conv2d_bwd_data.  dw and db are summed in a fixed order (the same
 * bits on every run).  `state`: 64 uint32 owned by the caller, zero before the first call; every call leaves them zero
 * (arrival counters of the launch's blocks; one buffer per stream that may run this call concurrently). */
int a3d_conv2d_bwd_both_supported(const a3d_conv_desc* d);
size_t a3d_conv2d_bwd_both_ws_bytes(const a3d_conv_desc* d);
int a3d_conv2d_bwd_both(const a3d_conv_desc* d, const float* x, const float* dz, const float* w, float* dw, float* db,
                        void* dx, int lddx, int dx_bf16, int relu_mask, uint32_t* state, void* ws, size_t ws_bytes,
                        void* stream);

/* A SECOND copy of a launch's finished output (after bias / activation / dropout / mask), in another type, pitch or place —
 * written by the split-K reduction that writes the first, instead of the cast or copy launch that would otherwise follow
 * (BASELINE config 5 keeps five such tensors in two types: the bf16 side of the conv stacks and the fp32 side of the dense
 * layers and the loss; src/models.py:222-234).  Element (row, col), col < cols, goes to ptr[(row * ld + col) * step + offset],
 * as bf16 (bf16 != 0) or float32.  Rows are GEMM rows: samples for a dense layer, output pixels for a conv.  Where the launch
 * has no reduction stage the library adds the copy launch itself.  NULL / ptr == NULL: none. */
typedef struct a3d_second_output {
  void* ptr;
  int32_t ld, step, offset;
  int32_t bf16;
  int32_t cols;
} a3d_second_output;

/* a3d_conv2d_fwd with a second output (implicit-GEMM forwards: no fused pool, not the few-channel / one-filter kernels). */
int a3d_conv2d_fwd_ex2(const a3d_conv_desc* d, const float* x, const float* w, const float* bias, float* y, int act,
                       const a3d_second_output* out2, void* ws, size_t ws_bytes, void* stream);

/* tf.layers.dense (src/models.py:80-82,228,231): y[m,n] = act(x[m,:] @ w[:,n] + b[n]).
 * If drop_keep != NULL (uint8 [m,n]) the tf.layers.dropout(rate=.5, training=True) of src/models.py:230 is fused:
 * y *= 2 * keep. */
size_t a3d_dense_fwd_ws_bytes(int m, int k, int n);
int a3d_dense_fwd(int m, int k, int n, const float* x, const float* w, const float* bias, float* y, int act,
                  const uint8_t* drop_keep, void* ws, size_t ws_bytes, void* stream);
/* dx[m,k] = dz[m,:] @ w[k,:]^T, optionally fused with the activation gradient of the layer below, whose OUTPUT is
 * `mask` [m,k]:  mask_act = A3D_ACT_RELU: dx *= scale * (mask > 0)  (with mask = the dropped-out activations this is
 * dropout-grad and ReluGrad in one);  A3D_ACT_SIGMOID: dx *= scale * mask * (1 - mask). */
size_t a3d_dense_bwd_data_ws_bytes(int m, int k, int n);
int a3d_dense_bwd_data(int m, int k, int n, const float* dz, const float* w, float* dx, const float* mask,
                       int mask_act, float scale, void* ws, size_t ws_bytes, void* stream);
/* dw[k,n] = x^T @ dz ; db[n] = sum_m dz  (db may be NULL) */
size_t a3d_dense_bwd_filter_ws_bytes(int m, int k, int n);
int a3d_dense_bwd_filter(int m, int k, int n, const float* x, const float* dz, float* dw, float* db,
                         void* ws, size_t ws_bytes, void* stream);

/* tf.layers.max_pooling2d(x, 2, 2) VALID (src/models.py:65,68,73,213,216,243).
 * y has pixel stride ldy >= c.  If extra != NULL, y channel c is filled from extra[n,ho,wo] (fuses the
 * tf.concat([pooled, coarse], -1) of src/models.py:246; requires ldy >= c+1). */
int a3d_maxpool2x2_fwd(int n, int h, int w, int c, const float* x, float* y, int ldy, const float* extra,
                       void* stream);
/* The same MaxPoolGrad (+ ReluGrad) from what a3d_conv2d_pool_fwd leaves: dx[n,h,w,c] = dy at the recorded position of
 * each window if (!relu_mask || pooled > 0), zero elsewhere (incl. the odd last row / column the pool dropped).
 * y: pooled values [n,h/2,w/2] with pixel stride ldy; argmax dense [n,h/2,w/2,c]; dy pixel stride lddy. */
int a3d_maxpool2x2_bwd_idx(int n, int h, int w, int c, const uint8_t* argmax, const float* y, int ldy, const float* dy,
                           int lddy, float* dx, int relu_mask, void* stream);
/* MEASUREMENT AID (bench.py --dp-rank-standin; nothing in the training path calls it): a launch shaped like one rank's share
 * of the gradient exchange that replaces the parameter servers of src/ann3depth.py:77-92 — `workgroups` blocks read read_bytes
 * from src, add them up, write write_bytes to dst (16-byte aligned, multiples of 16), and pace themselves so that the launch
 * lasts (read_bytes + write_bytes) / gbytes_per_s.  Run on a second stream beside a data-parallel rank's step it bounds what
 * a collective of that size and rate costs the kernels it overlaps, on ONE GPU. */
int a3d_comm_standin(const float* src, size_t read_bytes, float* dst, size_t write_bytes, int workgroups, float gbytes_per_s,
                     void* stream);

/* dst[pix, c_dst] = src[pix, c_src] for npix pixels (pixel strides ld_src / ld_dst): the coarse map into channel 63 of
 * the fine network's concat buffer (tf.concat, src/models.py:246) when the pooling kernel that normally does it is
 * fused away. */
int a3d_copy_channel(size_t npix, const float* src, int ld_src, int c_src, float* dst, int ld_dst, int c_dst,
                     void* stream);
/* MaxPoolGrad (first maximum in scan order) fused with the ReluGrad of the conv that produced x:
 * dx = (argmax ? dy : 0) * (x > 0 if relu_mask else 1).  dy has pixel stride lddy. */
int a3d_maxpool2x2_bwd(int n, int h, int w, int c, const float* x, const float* dy, int lddy, float* dx,
                       int relu_mask, void* stream);

/* tf.image.resize_images = ResizeBilinear(align_corners=False), legacy src = dst*in/out mapping
 * (src/models.py:180-181,282-283). */
int a3d_resize_bilinear_tf1(int n, int h, int w, int c, const float* x, int oh, int ow, float* y, void* stream);
/* The two resizes of a training step (image and depth map of the same stored size, src/models.py:282-283) in one launch. */
int a3d_resize_bilinear_tf1_pair(int n, int h, int w, int c0, const float* x0, int oh0, int ow0, float* y0, int c1,
                                 const float* x1, int oh1, int ow1, float* y1, void* stream);
/* The same one or two resizes (x1 == NULL: one) with each source either float32 or the uint8 pixel values k of
 * a3d_record_decode_u8 (u8_0 / u8_1 != 0): a tap then reads fl(fl(fl(k / 255) - 0.5) + 0.5), the float the converter and
 * the loader's `+ 0.5` (src/data.py:84-85) produce from that pixel, from a 256-entry table built in LDS — the output is
 * bit-identical to resizing the float32 record. */
int a3d_resize_bilinear_tf1_ex(int n, int h, int w, int c0, const void* x0, int u8_0, int oh0, int ow0, float* y0, int c1,
                               const void* x1, int u8_1, int oh1, int ow1, float* y1, void* stream);

/* tf.extract_image_patches(k x k, stride, SAME) + reshape (src/models.py:53-59): y [n*ph*pw, k, k, c]. */
int a3d_extract_patches(int n, int h, int w, int c, const float* x, int k, int stride, float* y, void* stream);

/* Scale-invariant log loss (src/models.py:255-275).  out/tgt [b, npix]; loss: 1 float.  ws: A3D_SILOG_WS_FLOATS(b)
 * floats — [0, 2b) the per-sample sums the backward call reads, [2b] the arrival ticket of the single launch (zero before
 * the FIRST call, zero again after every call), then the partial sums of the A3D_SILOG_PARTS blocks that share a sample. */
#define A3D_SILOG_PARTS 8
#define A3D_SILOG_WS_FLOATS(b) ((b) * 2 + 1 + (b) * 2 * A3D_SILOG_PARTS)
int a3d_silog_loss_fwd(int b, int npix, const float* out, const float* tgt, float* loss, float* ws, void* stream);
/* d loss / d out, using the per-sample sums left in ws by the forward call. */
int a3d_silog_loss_bwd(int b, int npix, const float* out, const float* tgt, const float* ws, float* dout,
                       void* stream);
/* ... and the same gradient a second time as bf16 rows of pitch ld_bf16 >= npix (the form config 5's dense_1 reads dz in: rows
 * of whole 16-byte pieces; columns npix .. ld_bf16 are the caller's, kept zero).  dout_bf16 == NULL: a3d_silog_loss_bwd. */
int a3d_silog_loss_bwd_ex(int b, int npix, const float* out, const float* tgt, const float* ws, float* dout, void* dout_bf16,
                          int ld_bf16, void* stream);

/* Keep mask of tf.layers.dropout(rate, training=True) (src/models.py:230): keep[i] = floor((1-rate) + u_i),
 * u from Philox4x32-10 keyed by (seed, step).  TF's own random stream is not reproducible, so parity tests pass
 * the mask in; the training driver draws it with this kernel. */
int a3d_dropout_keep_mask(size_t count, uint64_t seed, uint64_t step, float rate, uint8_t* keep, void* stream);

/* a3d_dense_fwd / a3d_dense_bwd_data with the arithmetic and storage of a3d_conv_desc: precision A3D_PREC_*, storage
 * A3D_STORE_W_BF16 when w is the layer's bf16 weight copy (k and n multiples of 8).  x / dz / y / dx stay float32: at
 * batch <= 64 they are a few MB against the layer's 100-200 MB of weights. */
int a3d_dense_fwd_ex(int m, int k, int n, const float* x, const float* w, const float* bias, float* y, int act,
                     const uint8_t* drop_keep, int precision, int storage, void* ws, size_t ws_bytes, void* stream);
int a3d_dense_bwd_data_ex(int m, int k, int n, const float* dz, const float* w, float* dx, const float* mask, int mask_act,
                          float scale, int precision, int storage, void* ws, size_t ws_bytes, void* stream);
/* ... with a second output (a3d_second_output); the forward may also store y as rows of ncols_y <= n columns at pitch ldy
 * (a GEMM padded to whole 16-byte pieces — dense_1: 4072 for 4070 — writing the unpadded tensor; needs a launch with a
 * reduction stage, A3D_EINVAL otherwise). */
int a3d_dense_fwd_ex2(int m, int k, int n, const float* x, const float* w, const float* bias, float* y, int ldy, int ncols_y, int act,
                      const uint8_t* drop_keep, int precision, int storage, const a3d_second_output* out2, void* ws,
                      size_t ws_bytes, void* stream);
int a3d_dense_bwd_data_ex2(int m, int k, int n, const float* dz, const float* w, float* dx, const float* mask, int mask_act,
                           float scale, int precision, int storage, const a3d_second_output* out2, void* ws, size_t ws_bytes,
                           void* stream);

/* float32 <-> bf16 (round to nearest even) of `count` elements: weight copies after ApplyAdam, and the two small tensors
 * that cross between the bf16 conv stack and the float32 dense layers (to_bf16 != 0: src float32 -> dst bf16). */
int a3d_cast_bf16(size_t count, const void* src, void* dst, int to_bf16, void* stream);
/* The same between matrices of different row pitches: dst[r][c] = src[r][c] for c < cols, zero for cols <= c < ld_dst
 * (src_bf16 / dst_bf16: the element types).  The bf16 copy of tf.layers.dense's [4096, 4070] kernel (src/models.py:231)
 * is kept with rows of 4072 elements — whole 16-byte pieces — and the layer's x, dz and y cross to it and back. */
int a3d_cast_rows(size_t rows, int cols, const void* src, int ld_src, int src_bf16, void* dst, int ld_dst, int dst_bf16,
                  void* stream);

/* float32 [pixels][c_src] -> bf16 [pixels][4] with the missing channels zero (c_src <= 4): the network image as 8-byte
 * pixels.  a3d_conv2d_fwd with A3D_STORE_X_BF16, c = ldx = 4, no padding, an even stride and precision A3D_PREC_BF16
 * gathers such an image in whole window runs through the bf16 kernel (fine/first, src/models.py:241, whose float32
 * window runs start 24 bytes apart and cannot); its filter is the float32 [r][s][4][k] one (channel 3: anything). */
int a3d_pad_channels_bf16(size_t pixels, int c_src, const float* src, int c_dst, void* dst, void* stream);

/* a3d_maxpool2x2_fwd / a3d_maxpool2x2_bwd on bf16 tensors (x, y, dy, dx bf16, pixel strides ldx / ldy / lddy >= c; `extra`, the concatenated channel, stays
 * float32: it is the coarse network's output).  Same semantics, first maximum in scan order. */
int a3d_maxpool2x2_fwd_bf16(int n, int h, int w, int c, const void* x, int ldx, void* y, int ldy, const float* extra,
                            void* stream);
int a3d_maxpool2x2_bwd_bf16(int n, int h, int w, int c, const void* x, int ldx, const void* dy, int lddy, void* dx,
                            int relu_mask, void* stream);     /* dx has x's pixel stride */

/* a3d_copy_channel into a bf16 tensor, and a3d_maxpool2x2_bwd_idx from bf16 pooled values / bf16 dy to a float32 dx: the
 * fused conv + pool of the two 3-channel layers keeps fp32 arithmetic under bf16 storage and writes only its pooled map
 * (bf16) and argmax bytes. */
int a3d_copy_channel_bf16(size_t npix, const float* src, int ld_src, int c_src, void* dst, int ld_dst, int c_dst, void* stream);
int a3d_maxpool2x2_bwd_idx_bf16(int n, int h, int w, int c, const uint8_t* argmax, const void* y, int ldy, const void* dy,
                                int lddy, float* dx, int relu_mask, void* stream);
/* ... and to a bf16 dx (dense [n,h,w,c]): the backward of a3d_conv2d_pool_fwd on bf16 x, w and y (config 5's conv2d_1:
 * src/models.py:214-215), c, ldy, lddy multiples of 8 and 16-byte aligned tensors. */
int a3d_maxpool2x2_bwd_idx_bf16s(int n, int h, int w, int c, const uint8_t* argmax, const void* y, int ldy, const void* dy,
                                 int lddy, void* dx, int relu_mask, void* stream);

/* dense_bwd_filter and ApplyAdam of one dense layer in ONE pass, for the optimizer the reference actually builds:
 * AdamOptimizer(rate, 0.9, beta2 = 1) (src/models.py:309) has alpha = 0 and 1 - beta2 = 0, so ApplyAdam moves only the
 * m slot: m += (g - m)(1 - beta1) with g = grad_scale * x^T dz (kernel) / grad_scale * sum_rows dz (bias; pass the three
 * bias pointers as NULL to leave the bias alone).  v / var are touched only where a non-finite g or m poisons them,
 * exactly as a3d_adam_apply_tf1 does.  The gradient itself is never written: single-GPU training only (a data-parallel
 * replica needs it for the all-reduce and calls a3d_dense_bwd_filter + a3d_adam_apply_tf1).  A3D_EINVAL unless
 * alpha == 0 and beta2 == 1, and for m > 64.  Replaces compute_gradients + apply_gradients of the coarse/dense layers
 * (src/models.py:319-330). */
int a3d_dense_bwd_filter_adam_tf1(int m, int k, int n, const float* x, const float* dz, float* var_w, float* m_w,
                                  float* v_w, float* var_b, float* m_b, float* v_b, float lr, float beta1, float beta2,
                                  float beta1_power, float beta2_power, float grad_scale, void* stream);
/* The same with the arithmetic of the contraction chosen: A3D_PREC_BF16 (BASELINE config 5: the conv stack's arithmetic)
 * rounds x and dz to bf16 and accumulates in float32 on the bf16 matrix cores when m > 32 — at 64 rows the float32 form is
 * no longer a pure weight stream; A3D_PREC_F32 is a3d_dense_bwd_filter_adam_tf1. */
int a3d_dense_bwd_filter_adam_tf1_ex(int m, int k, int n, const float* x, const float* dz, float* var_w, float* m_w,
                                  float* v_w, float* var_b, float* m_b, float* v_b, float lr, float beta1, float beta2,
                                  float beta1_power, float beta2_power, float grad_scale, int precision, void* stream);

/* tf.train.AdamOptimizer ApplyAdam (src/models.py:309): alpha = lr*sqrt(1-b2p)/(1-b1p);
 * m += (g-m)(1-b1); v += (g*g-v)(1-b2); var -= m*alpha/(sqrt(v)+eps).  grad_scale multiplies g first
 * (1/world_size after an all-reduce sum). */
int a3d_adam_apply_tf1(size_t count, float* var, float* m, float* v, const float* g, float lr, float beta1,
                       float beta2, float eps, float beta1_power, float beta2_power, float grad_scale,
                       void* stream);

/* The same ApplyAdam over one rank's SLICE of a parameter group (data-parallel replicas that reduce-scatter their
 * gradients instead of all-reducing them: each rank updates only the slice whose gradient sum it received — what
 * replaces the parameter server's per-variable update of src/ann3depth.py:77-92).  `poisoned` (device, may be NULL)
 * gets bit 0 set when the update left a non-finite value in var[0..count) — the only case in which the reference's
 * frozen optimizer (beta2 = 1) changes a weight at all, and the one the other ranks must then be told about. */
int a3d_adam_apply_tf1_flag(size_t count, float* var, float* m, float* v, const float* g, float lr, float beta1,
                            float beta2, float eps, float beta1_power, float beta2_power, float grad_scale,
                            unsigned int* poisoned, void* stream);

/* ---- DCNF pairwise part + CRF negative log-likelihood (src/models.py:20-48,91-177,185-200) ----
 * Superpixels are the sp x sp (40 x 40) non-overlapping blocks of the 240x320 image, row-major (src/models.py:37-48). */

/* tf.reduce_mean(superpixels, axis=2) (src/models.py:110,132): x [n,h,w,c] -> out [n,(h/sp)*(w/sp),c]. */
int a3d_superpixel_mean(int n, int h, int w, int c, const float* x, int sp, float* out, void* stream);

/* color_histogram (src/models.py:95-100): 256-bin tf.histogram_fixed_width of r*2^24+g*2^16+b*2^8 over [0,2^24) per
 * superpixel: x [n,h,w,3] -> hist [n,(h/sp)*(w/sp),256] (float counts). */
int a3d_superpixel_hist(int n, int h, int w, const float* x, int sp, float* hist, void* stream);

/* similarity() of the grayscale superpixels and of the histograms for every (left,right) pair (src/models.py:102-119)
 * and the pairwise dense layer 2->1 (src/models.py:121-127): sims [n,npairs,2], r [n,npairs].
 * left/right: device int32 superpixel indices (src/models.py:20-30). */
int a3d_pair_similarity(int n, int h, int w, const float* x, int sp, const float* hist, const int32_t* left,
                        const int32_t* right, int npairs, const float* dense_w, const float* dense_b, float gamma,
                        float* sims, float* r, void* stream);

/* loss_part (src/models.py:129-177): per image A = I + D - R (get_A, :136-143), energy, partition function
 * (matrix_determinant + matrix_inverse, here one LU with partial pivoting), loss_b = -log(exp(-E)/Z + eps);
 * loss_mean = mean_b; dz = d loss_mean / d z with A held constant (TF 1.3 has no gradient for scatter_nd_update).
 * z, y, dz: [n,nsp]; r: [n,npairs]; nsp <= 64. */
int a3d_crf_loss(int n, int nsp, const float* z, const float* y, const float* r, const int32_t* left,
                 const int32_t* right, int npairs, float eps, float* loss_per_image, float* loss_mean, float* dz,
                 void* stream);

/* tf.train.GradientDescentOptimizer (src/models.py:198): var -= lr * g. */
int a3d_sgd_apply(size_t count, float* var, const float* g, float lr, void* stream);

/* ---- opt-in kernel timing for bench.py's roofline line (the only process-global state in the library) ----
 * While enabled, every implicit-GEMM launch (conv / dense, any direction; or one kernel's: a3d_timing_select) is
 * bracketed by a hipEvent pair recorded
 * on the launch stream.  a3d_timing_collect() synchronises those events, returns up to `cap` records (oldest first),
 * and clears the list.  Not for use inside graph capture. */
typedef struct a3d_timing_record {
  int32_t mode;        /* 0 fwd, 1 bwd-data, 2 bwd-filter */
  int32_t bm, bn, waves_m, nwaves, bk, avec, bvec;   /* igemm_kernel<mode,bm,bn,waves_m,nwaves,bk,avec,bvec> */
  int32_t prec;        /* A3D_PREC_*; for bf16 modes the kernel is igemm_bf16_kernel<mode,bm,bn,x3> */
  int32_t lds_dma;     /* 1: igemm_glds_kernel<mode,bm,bn,waves_m,nwaves> (tiles staged by global_load_lds);
                        * 2: conv3_fwd_kernel<bm/32,bn/32,pool> (few-channel layers, one wave per bm x bn tile, no LDS);
                        * 3: igemm_ring_kernel<mode,bm,bn> (bf16-stored operands, LDS-DMA);
                        * 4: fewch_bwdf_kernel<bn/32> (few-channel filter gradient from LDS-staged rows; ms includes its slab reduction);
                        * 5: igemm2_kernel<mode> (second-generation fp32 kernel, LDS-DMA staging, 128 x 128 tiles: pinned plans only) */
  int32_t splitk;
  int32_t m, n, k;     /* GEMM extents of the launch */
  float ms;            /* duration of the igemm kernel alone (split-K reduction excluded) */
  double flops;        /* algorithmic 2*m*n*k */
} a3d_timing_record;
int a3d_timing_enable(int on);
int a3d_timing_collect(a3d_timing_record* out, int cap);
/* Bracket only launches of ONE kernel: the one `like` names by its template fields (mode, prec, bm, bn, waves_m, nwaves,
 * bk, avec, bvec, lds_dma; the other fields are ignored); NULL = every launch again.  An event pair per launch costs the
 * stream a few microseconds (30 GEMM launches per step: 3.5 % of the fp32 step, 8 % of the bf16-storage one), so
 * bench.py times every kernel during its warm-up steps and, inside the timed region, only the dominant one. */
int a3d_timing_select(const a3d_timing_record* like);

/* ---- host side of the dataset plugin: TFRecord container + tf.train.Example (src/data.py:62-86,
 *      tools/data_tf_converter.py:27-53) ---- */
uint32_t a3d_crc32c(const void* data, size_t len);
uint32_t a3d_masked_crc32c(const void* data, size_t len);

typedef struct a3d_example_view {
  int64_t image_height, image_width, image_channels;
  int64_t depth_height, depth_width, depth_channels;
  const uint8_t* image; size_t image_bytes;   /* point into the record payload */
  const uint8_t* depth; size_t depth_bytes;
} a3d_example_view;

/* Frame at buf[0..len): returns payload offset/length; A3D_EFORMAT on CRC mismatch or truncation.
 * *consumed = bytes of the whole frame. verify_crc=0 skips the payload CRC. */
int a3d_tfrecord_next(const uint8_t* buf, size_t len, int verify_crc, size_t* payload_off, size_t* payload_len,
                      size_t* consumed);
/* Parse the 8-feature Example of tools/data_tf_converter.py:41-51. */
int a3d_example_parse(const uint8_t* payload, size_t len, a3d_example_view* out);
/* data._convert_img_depth (src/data.py:82-85): dst[i] = src_le_f32[i] + 0.5 */
int a3d_decode_raw_plus_half(const uint8_t* src, size_t bytes, float* dst);
/* Reader fast path: one framed record at `frame` -> payload CRC check (if verify_crc), Example parse and
 * decode_raw + 0.5 of both features into image_dst / depth_dst (float counts must match the record) in a single pass
 * over the bytes.  *view (optional) receives the size features. */
int a3d_record_decode(const uint8_t* frame, size_t len, int verify_crc, float* image_dst, size_t image_floats,
                      float* depth_dst, size_t depth_floats, a3d_example_view* view);
/* a3d_record_decode for records written by the converter (tools/data_tf_converter.py:36-37: png_u8 / 255 - 0.5): a feature
 * whose floats ALL have that form, checked bit for bit, is delivered as the uint8 values k instead (*kinds bit 0: image in
 * image_u8, bit 1: depth in depth_u8) — a quarter of the bytes for the pinned staging pool, the host link and the
 * device-side read; a3d_resize_bilinear_tf1_ex rebuilds exactly the float32 value `stored + 0.5` of src/data.py:84-85 from
 * k.  Any other feature is decoded to its float32 destination as a3d_record_decode does. */
int a3d_record_decode_u8(const uint8_t* frame, size_t len, int verify_crc, uint8_t* image_u8, float* image_f32,
                         size_t image_count, uint8_t* depth_u8, float* depth_f32, size_t depth_count,
                         a3d_example_view* view, int* kinds);
/* n framed records into slots slots[i] of the staging pool (dense arrays of image / depth features: float32, and uint8
 * twins or NULL) in one call: a3d_record_decode_u8 per record when the twins exist, a3d_record_decode otherwise; kinds[i]
 * as a3d_record_decode_u8's.  dims = {image h, w, c, depth h, w, c} (all positive): a record of any other size is an
 * error.  The pools hold `nslots` slots; a slot number outside [0, nslots) is A3D_EINVAL and nothing is written. */
int a3d_records_decode(const void* const* frames, const size_t* lens, int n, int verify_crc, const int64_t* dims,
                       uint8_t* image_u8_pool, float* image_f32_pool, uint8_t* depth_u8_pool, float* depth_f32_pool,
                       const int32_t* slots, int nslots, int32_t* kinds);
/* One dequeued batch from the pinned staging pool to its device buffer: record b of the batch is slot slots[b] of the pool
 * (`bytes_each` bytes per slot, both sides dense), n asynchronous host-to-device copies on `stream` issued from ONE call —
 * the shuffle queue hands out slot numbers (src/data.py:51-55: tf.train.shuffle_batch), and 2 x 32 copies per step issued
 * one by one from the host language cost more of the step's launch budget than the copies themselves.  The pool holds
 * `nslots` slots; a slot number outside [0, nslots) is A3D_EINVAL and no copy is enqueued. */
int a3d_h2d_gather(void* dst, const void* src_pool, const int32_t* slots, int n, int nslots, size_t bytes_each, void* stream);
/* Serialise one framed record (writer side).  Returns bytes written, or the needed size if cap is too small
 * (nothing written then), or a negative error. */
int64_t a3d_example_write(const float* image, int ih, int iw, int ic, const float* depth, int dh, int dw, int dc,
                          uint8_t* dst, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* A3D_H_ */
