/* nq_hip.h -- C ABI of libnqhip.so: the MI355X (gfx950) kernels behind NeuroQuant's network-wise
 * calibration hot path.
 *
 * The reference (Eric-qi/NeuroQuant) has no FFI layer: its boundary is the Python object API of
 * quantization/{quantizer,quant_layer,quant_model,calib_model}.py and models/{HNeRV,NeRV}.py.  The Python
 * host in neuroquant_amd/ keeps that API and reaches every entry point below through ctypes on
 * torch.cuda.current_stream().  Each entry point names the reference code (file:line under
 * /root/reference) whose arithmetic it replaces.
 *
 * Conventions (all entry points):
 *   - plain C: device pointers + sizes, no torch types; fp32 tensors, contiguous, NCHW / OIHW;
 *   - no allocation, no ownership transfer, no host synchronisation: the caller passes outputs and
 *     workspaces; work is enqueued on `stream` (a hipStream_t) and the call returns immediately;
 *   - returns NQ_OK (0) or a negative NQ_ERR_* code; nothing throws across the boundary;
 *   - re-entrant from any thread and device; the only thing the library remembers is, per kernel and per device, that
 *     the >64 KB dynamic-LDS opt-in has been granted (lock-free, idempotent); pointers must be 16-byte aligned unless
 *     stated otherwise.
 *
 * "rows x row_len" tensors: a weight (C_out, C_in, k, k) is rows=C_out, row_len=C_in*k*k; a bias is
 * rows=1, row_len=C_out.  per_row=1 -> delta/zp hold one value per row (channel-wise weight),
 * per_row=0 -> one scalar for the whole tensor (bias, layer-wise weight)  (quantizer.py:129-153).
 */
#ifndef NQ_HIP_H
#define NQ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* nq_stream_t; /* hipStream_t */

/* The library is built with -fvisibility=hidden: exactly the entry points declared in this header are exported
 * (`nm -D --defined-only libnqhip.so` == this file; checked by tests/test_cabi_cpu.py). */
#define NQ_API __attribute__((visibility("default")))

#define NQ_OK 0
#define NQ_ERR_INVALID (-1)     /* bad argument (null pointer, non power-of-two length, size <= 0 ...) */
#define NQ_ERR_UNSUPPORTED (-2) /* shape outside what the kernels are built for */
#define NQ_ERR_LAUNCH (-3)      /* HIP reported a launch error */

NQ_API int nq_abi_version(void);
NQ_API const char* nq_error_string(int code);

/* ---------------------------------------------------------------- quantiser parameter side ---- */

/* UniformAffineQuantizer.init_quantization_scale, 'max' branch (quantizer.py:127-168): per row
 * delta = max((max(x,0)-min(x,0))/(n_levels-1), 1e-8) (division in double, stored fp32),
 * zp = rint(-min/delta).  One value per row is written; a bias / layer-wise tensor is rows=1. */
NQ_API int nq_scale_init_max(const float* x, int64_t rows, int64_t row_len, int n_levels, float* delta, float* zp,
                      nq_stream_t stream);

/* UniformAffineQuantizer.forward (quantizer.py:117-119): y = (clamp(rint(x/delta)+zp, 0, L-1) - zp)*delta. */
NQ_API int nq_uaq_forward(const float* x, const float* delta, const float* zp, float* y, int64_t rows, int64_t row_len,
                   int per_row, int n_levels, nq_stream_t stream);

/* Backward of the above (round_ste, quantizer.py:53-57): ddelta[row] = sum gy*((xq-zp) -
 * 1{0<=rint(x/delta)+zp<=L-1} * x/delta), overwritten (rows values, or 1 if !per_row); dx (may be NULL; same shape as x)
 * = gy * 1{0<=rint(x/delta)+zp<=L-1}, the straight-through gradient w.r.t. the quantiser input. */
NQ_API int nq_uaq_backward(const float* x, const float* gy, const float* delta, const float* zp, float* ddelta, float* dx,
                    int64_t rows, int64_t row_len, int per_row, int n_levels, nq_stream_t stream);

/* AdaRoundQuantizer.__init__/init_alpha (quantizer.py:264-265, 305-314): delta/zp through an fp16 round
 * trip, alpha = -log((zeta-gamma)/(frac(x/delta)-gamma) - 1).  delta_out/zp_out have the size of delta_in. */
NQ_API int nq_adaround_init(const float* x, const float* delta_in, const float* zp_in, float* delta_out, float* zp_out,
                     float* alpha, int64_t rows, int64_t row_len, int per_row, nq_stream_t stream);

/* AdaRoundQuantizer.forward 'learned_hard_sigmoid' (quantizer.py:288-300): soft!=0 -> floor(x/delta)+h(alpha),
 * else floor(x/delta)+1{alpha>=0}; xq (may be NULL) receives the clamped integer grid value x_quant. */
NQ_API int nq_adaround_forward(const float* x, const float* alpha, const float* delta, const float* zp, float* y, float* xq,
                        int64_t rows, int64_t row_len, int per_row, int n_levels, int soft, nq_stream_t stream);

/* d/dalpha of the soft forward, plus (reg_weight != 0) the gradient of the rounding regulariser
 * reg_weight * sum(1 - |2h(alpha)-1|^reg_b) (calib_model.py:39-47).  dalpha is overwritten. */
NQ_API int nq_adaround_backward(const float* x, const float* gy, const float* alpha, const float* delta, const float* zp,
                         float* dalpha, int64_t rows, int64_t row_len, int per_row, int n_levels, float reg_weight,
                         float reg_b, nq_stream_t stream);

/* Value of the regulariser over one alpha tensor: out[0] (+)= weight*sum(1-|2h-1|^b) (calib_model.py:45).
 * ws: >= nq_reduce_ws_floats(n) floats of scratch; accumulate!=0 adds to out[0]. Deterministic. */
NQ_API int64_t nq_reduce_ws_floats(int64_t n);
NQ_API int nq_round_loss(const float* alpha, int64_t n, float b, float weight, float* ws, float* out, int accumulate,
                  nq_stream_t stream);

/* Gradient of the regulariser alone: dalpha (+)= gscale[0] * weight * d/dalpha sum(1-|2h-1|^b); gscale is a
 * device scalar (the upstream gradient of the loss term) or NULL for 1. */
NQ_API int nq_round_loss_backward(const float* alpha, int64_t n, float b, float weight, const float* gscale, float* dalpha,
                           int accumulate, nq_stream_t stream);

/* torch.optim.Adam single-tensor step, no weight decay / amsgrad (used by calib_model.py:134, 195):
 * m += (g-m)*(1-beta1); v = v*beta2 + (1-beta2)*g*g; p -= step_size * m / (sqrt(v)/bc2_sqrt + eps),
 * step_size = lr/(1-beta1^t), bc2_sqrt = sqrt(1-beta2^t) computed by the host in double. */
NQ_API int nq_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float step_size, float beta1, float beta2,
                 float eps, float bc2_sqrt, nq_stream_t stream);

/* ---- multi-tensor variants: ONE launch for all layers (the per-iteration parameter side is 14 small tensors; 42
 * launch-bound kernels become 3).  `segs` is a HOST array (copied into the kernel arguments, <= 16 segments per launch,
 * longer lists are chunked); every pointer inside is a device pointer.  Per-element arithmetic is identical to the
 * single-tensor entry points above (same results bit for bit). */
typedef struct nq_ada_seg {
  const float* x;      /* weights (rows x row_len) */
  const float* gy;     /* backward only: gradient w.r.t. the fake-quantised output */
  const float* alpha;
  const float* delta;  /* (rows) when per_row, else (1) */
  const float* zp;
  float* out;          /* forward: fake-quantised weights; backward: d(alpha) (+ regulariser gradient) */
  int64_t rows, row_len;
  int per_row, n_levels, soft; /* soft = soft_targets (forward only) */
  float reg_weight;    /* backward only: regulariser weight lambda, 0 = none (bias quantisers, warm-up) */
} nq_ada_seg;
typedef struct nq_adam_seg {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
} nq_adam_seg;
/* d(alpha) (+ regulariser gradient where reg_weight != 0) and torch.optim.Adam's update of alpha in ONE pass over the
 * rounding variables (calib_model.py:170-226: loss.backward(); optimizer.step()): per element the arithmetic of
 * nq_adaround_backward_multi followed by nq_adam_step_multi, d(alpha) is never stored; alpha, m, v are updated in place and
 * bit-identical to the two launches.  dyn != NULL: {reg_b, regulariser gate, lr/(1-beta1^t), sqrt(1-beta2^t)} are read from
 * that device array (nq_step_prologue) instead of the host arguments. */
typedef struct nq_ada_adam_seg {
  const float* x;
  const float* gy;
  float* alpha;
  const float* delta;
  const float* zp;
  float* m;
  float* v;
  int64_t rows, row_len;
  int per_row, n_levels;
  float reg_weight;
} nq_ada_adam_seg;
NQ_API int nq_adaround_adam_multi(const nq_ada_adam_seg* segs, int nseg, float reg_b, float step_size, float beta1, float beta2, float eps,
                           float bc2_sqrt, const float* dyn, nq_stream_t stream);
NQ_API int nq_adaround_forward_multi(const nq_ada_seg* segs, int nseg, nq_stream_t stream);
/* The UAQ fake-quant (nq_uaq_forward) / its d(delta) (nq_uaq_backward without dx) for several tensors in one launch:
 * phase 1 of the calibration (calib_model.py:119-165).  Uses x, gy (backward), delta, zp, out (forward: y; backward:
 * d(delta), one value per reduction row), rows, row_len, per_row, n_levels of nq_ada_seg; alpha / soft / reg_weight are
 * ignored.  Bit-identical to the single-tensor entry points. */
NQ_API int nq_uaq_forward_multi(const nq_ada_seg* segs, int nseg, nq_stream_t stream);
NQ_API int nq_uaq_backward_multi(const nq_ada_seg* segs, int nseg, nq_stream_t stream);
NQ_API int nq_adaround_backward_multi(const nq_ada_seg* segs, int nseg, float reg_b, nq_stream_t stream);
NQ_API int nq_adam_step_multi(const nq_adam_seg* segs, int nseg, float step_size, float beta1, float beta2, float eps,
                       float bc2_sqrt, nq_stream_t stream);

/* ---- captured iterations (hipGraph): what changes from one calibration iteration to the next is the batch (frame
 * indices), the temperature b / regulariser gate (LinearTempDecay + warm-up, calib_model.py:62-81) and Adam's two bias
 * corrections.  The host writes them for a whole epoch into device tables once; nq_step_prologue copies row *step into
 * fixed slots (cur_idx: B int64 frame indices; cur_scal: nscal floats) and increments *step, and the _dyn variants read
 * their scalars from cur_scal = {reg_b, regulariser gate (0 or 1), lr/(1-beta1^t), sqrt(1-beta2^t)} instead of from
 * host arguments -- same values, same arithmetic, so a replayed iteration is bit-identical to an eagerly launched one. */
NQ_API int nq_step_prologue(const int64_t* order, const float* scal, int* step, int64_t* cur_idx, float* cur_scal, int B, int nscal,
                     nq_stream_t stream);
/* The same, plus the batch's rows of a table gathered in the same launch: out[t] = table[cur_idx[t]] (row_len floats each),
 * i.e. the decoder inputs cali_data[idx] (calib_model.py:150, :201) without a separate index_select launch. */
NQ_API int nq_step_prologue_gather(const int64_t* order, const float* scal, int* step, int64_t* cur_idx, float* cur_scal, int B, int nscal,
                            const float* table, int64_t table_rows, int64_t row_len, float* out, nq_stream_t stream);
NQ_API int nq_adaround_backward_multi_dyn(const nq_ada_seg* segs, int nseg, const float* dyn, nq_stream_t stream);
NQ_API int nq_adam_step_multi_dyn(const nq_adam_seg* segs, int nseg, const float* dyn, float beta1, float beta2, float eps,
                           nq_stream_t stream);

/* Orthonormal Walsh-Hadamard transform along the middle axis (hadamard_along_channel_weight,
 * quant_layer.py:16-22, with the zero-padding of :45-49 and the slice of :71 folded in):
 * x is [outer][n_in][inner] (read as zero for index >= n_in), y is [outer][n_out][inner], transform
 * length n = 2^k <= 1024, n_in <= n, n_out <= n.  x and y must not alias. */
NQ_API int nq_fwht(const float* x, float* y, int64_t outer, int n, int64_t inner, int n_in, int n_out, nq_stream_t stream);

/* nq_fwht for several tensors in ONE launch (all layers of a decoder); `segs` is a host array, the pointers inside are
 * device pointers; fields as the arguments of nq_fwht (same arithmetic, bit-identical results). */
typedef struct nq_fwht_seg {
  const float* x;
  float* y;
  int64_t outer, inner;
  int n, n_in, n_out;
} nq_fwht_seg;
NQ_API int nq_fwht_multi(const nq_fwht_seg* segs, int nseg, nq_stream_t stream);

/* The AdaRound fake-quant fused with the transform (ABI v4; QuantModule.forward with hadamard=True, quant_layer.py:70-71, and
 * its backward + Adam, calib_model.py:170-226): one launch per direction for all layers of a decoder.
 *   nq_adaround_fwht_multi        y = H(Q(x))[:, :c_in] per weight tensor (x, alpha: (outer, n, inner) transform-domain weight and
 *                                 its rounding variables; y: (outer, c_in, inner)); a segment with n == 0 is a plain tensor
 *                                 (a bias: y = Q(x), outer * inner elements, one scalar delta / zp)
 *   nq_fwht_adaround_adam_multi   g = H(pad(gy)) (gy: (outer, c_in, inner)), then d(alpha) (+ regulariser gradient, reg_weight)
 *                                 and Adam's update of alpha / m / v in place; scalars and `dyn` as nq_adaround_adam_multi
 * Bit-identical to nq_adaround_forward_multi + nq_fwht_multi and to nq_fwht_multi + nq_adaround_adam_multi (same butterflies,
 * same per-element arithmetic); the transform-domain tensors no longer cross HBM between the two.  Rows with n * inner >
 * 8192 are NQ_ERR_UNSUPPORTED (callers use the separate launches). */
typedef struct nq_fq_fwht_seg {
  const float* x;
  float* alpha;
  const float* delta;
  const float* zp;
  float* m;          /* backward only */
  float* v;
  const float* gy;   /* backward only */
  float* y;          /* forward only */
  int64_t outer, inner;
  int n, c_in, per_row, n_levels, soft;
  float reg_weight;
} nq_fq_fwht_seg;
NQ_API int nq_adaround_fwht_multi(const nq_fq_fwht_seg* segs, int nseg, nq_stream_t stream);
NQ_API int nq_fwht_adaround_adam_multi(const nq_fq_fwht_seg* segs, int nseg, float reg_b, float step_size, float beta1, float beta2,
                                float eps, float bc2_sqrt, const float* dyn, nq_stream_t stream);

/* ---------------------------------------------------------------- convolution side ------------ */

/* Re-layout an OIHW weight (Cout,Cin,k,k) into the two GEMM operands the conv kernels read:
 *   wt_fwd [krows_fwd][ld_fwd]: row (ci*k+kh)*k+kw, column co            (forward)
 *   wt_bwd [krows_bwd][ld_bwd]: row (co*k+kh)*k+kw, column ci, taps flipped (data gradient)
 * rows/columns beyond the real extents are written as zero.  Either output may be NULL. */
NQ_API int nq_weight_layouts(const float* w, float* wt_fwd, float* wt_bwd, int Cout, int Cin, int k, int krows_fwd,
                      int ld_fwd, int krows_bwd, int ld_bwd, nq_stream_t stream);

/* nq_weight_layouts for several layers in ONE launch (the per-layer form is launch-bound); `segs` is a host array, the
 * pointers inside are device pointers; wt_fwd / wt_bwd may be NULL per segment; dims as for nq_weight_layouts. */
typedef struct nq_wl_seg {
  const float* w;
  float* wt_fwd;
  float* wt_bwd;
  int Cout, Cin, k, krows_fwd, ld_fwd, krows_bwd, ld_bwd;
} nq_wl_seg;
NQ_API int nq_weight_layouts_multi(const nq_wl_seg* segs, int nseg, nq_stream_t stream);

/* Padded operand sizes the conv kernels expect for a (Cin -> Cout, k) convolution. */
NQ_API int nq_conv_operand_dims(int Cin, int Cout, int k, int* krows, int* ld);

/* Implicit-GEMM convolution on the fp32 MFMA pipe (replaces F.conv2d in QuantModule.forward,
 * quant_layer.py:80, fused with what follows it in the decoder):
 *   stride 1, padding k/2, x (B,Cin,H,W), wt = wt_fwd layout from nq_weight_layouts, bias (Cout) or NULL.
 *   epilogue NQ_EPI_PLAIN     : y (B,Cout,H,W) = conv + bias
 *            NQ_EPI_PS_GELU   : PixelShuffle(r) + exact-erf GELU (quant_block.py:31-35, _layers.py:20-36):
 *                               with v = shuffled conv+bias, (B,Cout/r^2,H*r,W*r): y = gelu(v) and z = gelu'(v), the
 *                               derivative saved for the backward pass (one erf serves both; no backward kernel
 *                               evaluates erf/exp again)
 *            NQ_EPI_TANH      : y = tanh(conv+bias)*0.5+0.5 (OutImg, _layers.py:10-16)
 * The data gradient of a convolution is the same call with wt = wt_bwd and Cin/Cout swapped.
 * ws: scratch of >= nq_conv_forward_ws_floats(...) floats (may be NULL when that is 0): layers with few pixel
 * tiles split their K loop across workgroups; the partial sums are added in fixed order (deterministic). */
#define NQ_EPI_PLAIN 0
#define NQ_EPI_PS_GELU 1
#define NQ_EPI_TANH 2
#define NQ_EPI_PS 3         /* PixelShuffle(r) only: z = shuffled conv+bias (no activation) */
#define NQ_EPI_DGRAD_GELU 4 /* data gradient: y = conv * zprev, zprev (B,Cout,H,W) = the z a NQ_EPI_PS_GELU forward saved
                             * (= gelu' of the pre-activation), stored PixelUnshuffle(r)-ed, i.e. as the
                             * (B,Cout*r*r,H/r,W/r) output gradient of the convolution below */
/* Split {hi | lo} word interchange between the bf16x3 kernels (ABI v5; OR-ed into `epilogue`): a float v travels as one
 * 32-bit word {bf16 hi = bf16(v) in the upper half, bf16 lo = bf16(v - hi) in the lower half}, i.e. the two operands the
 * bf16x3 kernels derive from v when they stage it -- a consumer that is handed the word re-packs halves instead of converting.
 * Same bytes per element, same NCHW indexing, same results bit for bit.
 *   NQ_EPI_X_SPLIT : the input x of nq_conv_forward3 holds such words        (where nq_conv3_split_io has the bit)
 *   NQ_EPI_Y_SPLIT : the output y is written as such words (never z)         (nq_conv3_split_io / nq_conv_split_out)
 * nq_split_words converts a float tensor to that form (elementwise; for callers that feed such a kernel themselves). */
#define NQ_EPI_X_SPLIT 0x100
#define NQ_EPI_Y_SPLIT 0x200
NQ_API int nq_split_words(const float* x, float* y, int64_t n, nq_stream_t stream);
NQ_API int64_t nq_conv_forward_ws_floats(int B, int Cin, int H, int W, int Cout, int k);
/* 1 when nq_conv_forward can write y as split words (NQ_EPI_Y_SPLIT) for this call: the streaming data gradient of the head */
NQ_API int nq_conv_split_out(int B, int Cin, int H, int W, int Cout, int k, int r, int epilogue, int in_gelu, int has_bias);
/* in_gelu != 0: x holds pre-activations and exact GELU is applied while the input tile is staged. */
NQ_API int nq_conv_forward(const float* x, const float* wt, const float* bias, float* y, float* z, float* ws, int B, int Cin, int H,
                    int W, int Cout, int k, int krows, int ld, int r, int epilogue, int in_gelu, const float* zprev,
                    nq_stream_t stream);

/* ---- "bf16x3" variant of nq_conv_forward: fp32-equivalent accuracy on the BF16 matrix pipe ------------------------
 * Every fp32 operand is split into bf16 hi + lo and each product formed as hi*hi + hi*lo + lo*hi with fp32
 * accumulation (v_mfma_f32_16x16x32_bf16): relative error ~2^-16 per product before accumulation, i.e. the same
 * order as the fp32 rounding noise of these K~1000 contractions; 5.3x the fp32-MFMA rate.  k in {3,5} only.
 *   nq_conv3_supported    : 1 when the shape is served by this path (enough workgroups, > 4 channels each side)
 *   nq_conv3_weight_bytes : size of the pre-split operand of a (Cin -> Cout, k) convolution
 *   nq_weight_layout3     : builds it from the OIHW weight w; transposed = 1 builds the operand of the data gradient
 *                           (then Cin/Cout are those of the gradient convolution: Cin = w's C_out, Cout = w's C_in)
 *   nq_conv_forward3      : same contract as nq_conv_forward (epilogues, zprev); ws = nq_conv_forward3_ws_floats
 *                           floats (0 -> may be NULL): deep low-resolution layers are split over channel chunks */
NQ_API int nq_conv3_supported(int B, int Cin, int H, int W, int Cout, int k);
/* bit mask NQ_EPI_X_SPLIT | NQ_EPI_Y_SPLIT: the sides of nq_conv_forward3 that may travel as split words for this shape */
NQ_API int nq_conv3_split_io(int B, int Cin, int H, int W, int Cout, int k);
NQ_API int64_t nq_conv_forward3_ws_floats(int B, int Cin, int H, int W, int Cout, int k);
NQ_API int64_t nq_conv3_weight_bytes(int Cin, int Cout, int k);
NQ_API int nq_weight_layout3(const float* w, void* wt3, int Cin, int Cout, int k, int transposed, nq_stream_t stream);
/* several operands (all layers, forward and data-gradient) in ONE launch; `segs` is a host array, pointers inside are
 * device pointers; Cin/Cout/transposed per segment as for nq_weight_layout3 */
typedef struct nq_wl3_seg {
  const float* w;
  void* wt3;
  int Cin, Cout, k, transposed;
} nq_wl3_seg;
NQ_API int nq_weight_layout3_multi(const nq_wl3_seg* segs, int nseg, nq_stream_t stream);
/* nq_weight_layout3_multi + nq_weight_layouts_multi in ONE launch (ABI v4): every operand a decoder needs per iteration, the
 * same bytes as the two calls (which it falls back to when a table does not fit one kernel-argument block). */
NQ_API int nq_weight_layouts_all(const nq_wl3_seg* segs3, int n3, const nq_wl_seg* segsf, int nf, nq_stream_t stream);
NQ_API int nq_conv_forward3(const float* x, const void* wt3, const float* bias, float* y, float* z, const float* zprev, float* ws, int B,
                     int Cin, int H, int W, int Cout, int k, int r, int epilogue, nq_stream_t stream);

/* bf16x3 variant of nq_conv_wgrad (same contract, x_gelu not offered): */
NQ_API int nq_conv_wgrad3_supported(int B, int Cin, int H, int W, int Cout, int k);
NQ_API int64_t nq_conv_wgrad3_ws_floats(int B, int Cin, int H, int W, int Cout, int k);
/* The launch plan nq_conv_wgrad3 will use for this shape (pure host function): MT = 16*mi channels x NT = 64*ni columns per
 * workgroup, nsplit K-splits, pc != 0 -> the 8-wave producer/consumer kernel (32-bit buffer offsets: only for operands
 * below 2 GiB), pc == 0 -> the 4-wave kernel (64-bit pointers). */
NQ_API int nq_conv_wgrad3_plan(int B, int Cin, int H, int W, int Cout, int k, int* mi, int* ni, int* nsplit, int* pc);
NQ_API int nq_conv_wgrad3(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W, int Cout,
                   int k, nq_stream_t stream);
/* The same with operands as split {hi | lo} words (see NQ_EPI_X_SPLIT): fmt bit 0 -- x, bit 1 -- dy, only the bits
 * nq_conv_wgrad3_split_io returns for the shape (the row-segment producer/consumer kernel).  dw is bit-identical to the float
 * call on the un-split tensors; db sums hi + lo of every dy value (what the matrix pipe sees of it). */
NQ_API int nq_conv_wgrad3_split_io(int B, int Cin, int H, int W, int Cout, int k);
NQ_API int nq_conv_wgrad3_fmt(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W, int Cout,
                       int k, int fmt, nq_stream_t stream);
/* The same weight gradient dw (Cout,Cin,k,k) for a convolution with very FEW output channels (the 3-channel head, HNeRV.py:42)
 * by exchanged operand roles: R[ci][(co,tap)] = sum_p x[ci][p] * dy[co][p+tap] is the weight gradient of the convolution
 * dy -> x-channels and dW[co][ci][tap] = R[ci][co][k*k-1-tap]; the big tensor x is then the un-shifted GEMM operand read
 * exactly once.  ws: nq_conv_wgrad3_ws_floats(B, Cout, H, W, Cin, k) floats (the exchanged problem).  No bias gradient
 * (use nq_channel_sum on dy). */
NQ_API int nq_conv_wgrad3_swapped(const float* x, const float* dy, float* dw, float* ws, int B, int Cin, int H, int W, int Cout, int k,
                           nq_stream_t stream);

/* Weight + bias gradient of the same convolution: dw (Cout,Cin,k,k), db (Cout) (db may be NULL),
 * from x (B,Cin,H,W) and dy (B,Cout,H,W).  ws: scratch of >= nq_conv_wgrad_ws_floats(...) floats.
 * x_gelu != 0: x holds pre-activations, exact GELU is applied while staging.  Deterministic (fixed split-K order). */
NQ_API int64_t nq_conv_wgrad_ws_floats(int B, int Cin, int H, int W, int Cout, int k);
NQ_API int nq_conv_wgrad(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W, int Cout,
                  int k, int x_gelu, nq_stream_t stream);

/* Deferred slab reduction (round 3).  The weight-gradient kernels above split K over workgroups into slabs and finish with a
 * fixed-order reduction launch each (five ~10 us launches per HNeRV-3M iteration).  The *_slabs variants run ONLY the
 * split kernel and describe the pending reduction in *seg (slab / dw / db pointers stay owned by the caller until it ran);
 * nq_wgrad_reduce_multi performs up to 16 pending reductions per launch, each with the same summation order as the
 * single-tensor entry point -- results are bit-identical to nq_conv_wgrad3 / nq_conv_wgrad3_swapped / nq_conv_wgrad.
 * seg->nsplit == 0 on return: the kernel wrote dw / db itself (tiny 1x1 problems), nothing is pending. */
typedef struct nq_wgr_seg {
  const float* slab;     /* [nsplit][co_pad][n_pad] */
  const float* slab_db;  /* [nsplit][co_pad] or NULL */
  float* dw;
  float* db;             /* or NULL */
  int Cout, N, co_pad, n_pad, nsplit;
  int swap_kk;           /* > 0: role-swapped problem, dw[ci][co][kk-1-tap] = R[co][ci][tap] (nq_conv_wgrad3_swapped) */
  int sg;                /* split groups per output (1, 4 or 16): fixes the summation order */
} nq_wgr_seg;
NQ_API int nq_conv_wgrad3_slabs(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W, int Cout,
                         int k, nq_wgr_seg* seg, nq_stream_t stream);
NQ_API int nq_conv_wgrad3_slabs_fmt(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W,
                             int Cout, int k, nq_wgr_seg* seg, int fmt, nq_stream_t stream);   /* fmt: nq_conv_wgrad3_fmt */
NQ_API int nq_conv_wgrad3_swapped_slabs(const float* x, const float* dy, float* dw, float* ws, int B, int Cin, int H, int W, int Cout,
                                 int k, nq_wgr_seg* seg, nq_stream_t stream);
NQ_API int nq_conv_wgrad_slabs(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W, int Cout,
                        int k, int x_gelu, nq_wgr_seg* seg, nq_stream_t stream);
NQ_API int nq_wgrad_reduce_multi(const nq_wgr_seg* segs, int nseg, nq_stream_t stream);

/* Backward of PixelShuffle(r)+GELU: dconv (B,C*r*r,H,W) = unshuffle(da * z), da and z (B,C,H*r,W*r), z = the saved
 * derivative output of a NQ_EPI_PS_GELU forward. */
NQ_API int nq_ps_gelu_backward(const float* da, const float* z, float* dconv, int B, int C, int H, int W, int r,
                        nq_stream_t stream);

/* Backward of OutImg 'tanh': dconv = dimg * 0.5 * (1 - t^2), t = 2*img - 1. */
NQ_API int nq_tanh_out_backward(const float* dimg, const float* img, float* dconv, int64_t n, nq_stream_t stream);

/* lp_loss p=2 (quantizer.py:66-71): loss[0] = sum_{b,c,h,w}(pred-tgt)^2 / (B*H*W); dpred (may be NULL) =
 * 2*(pred-tgt)/(B*H*W) * gscale.  ws: >= nq_reduce_ws_floats(n) floats.  Deterministic. */
NQ_API int nq_l2_loss(const float* pred, const float* tgt, float* loss, float* dpred, float* ws, int64_t n, int64_t mean_count,
               float gscale, nq_stream_t stream);

/* Per-channel sums of an NCHW tensor, out[c] = sum_{b,h,w} x[b][c][h][w] (bias gradient of a convolution);
 * ws: >= 512*C floats.  Deterministic. */
NQ_API int nq_channel_sum(const float* x, float* out, float* ws, int B, int C, int64_t HW, nq_stream_t stream);

/* nq_l2_loss + nq_tanh_out_backward + nq_channel_sum of a tanh-headed decoder in one pass over the image (the tail of
 * calib_model.py:219-226 for OutImg 'tanh', models/_layers.py): loss[0] as nq_l2_loss (bit-identical), dconv (B,C,H,W) =
 * [2*(pred-tgt)/(B*H*W)*gscale] * 0.5 * (1 - (2*pred-1)^2) = the gradient at the head conv's output, db[c] = sum of dconv
 * over frames and pixels (the head's bias gradient).  The target is either float frames `tgt` (B,C,H,W) or the uint8
 * frame cache `cache_u8` (N,C,H,W) with frame indices idx[B] (tgt = cache[idx]/255, videosets/datasets.py:19-24);
 * exactly one of the two is non-NULL.  ws: >= 2*nq_reduce_ws_floats(B*C*HW) floats.  NQ_ERR_UNSUPPORTED unless
 * HW % 4096 == 0 (callers then use the three separate entry points).  Deterministic. */
NQ_API int nq_l2_loss_tanh_head(const float* pred, const float* tgt, const uint8_t* cache_u8, const int64_t* idx, float* loss,
                         float* dconv, float* db, float* ws, int B, int C, int64_t HW, int64_t mean_count, float gscale,
                         nq_stream_t stream);

/* The decoder's 3-channel tanh head (HNeRV.py:42, 64-66; NeRV.py:37, 58-60; OutImg models/_layers.py:10-16) AND the loss
 * tail of nq_l2_loss_tanh_head in one pass over the image (ABI v4): y (B,3,H,W) = tanh(conv3x3(x, w) + bias) * 0.5 + 0.5 is
 * still written; loss, dconv and db as nq_l2_loss_tanh_head defines them (the same per-element arithmetic; the loss and db
 * are summed per image strip and then in strip order: deterministic, last-bit different from the chunked sums of
 * nq_l2_loss_tanh_head).  wt / ld: the fp32 forward operand of nq_weight_layouts for the (Cin -> 3, k = 3) head.
 * ws: nq_head_forward_loss_ws_floats(B, H, W) floats.  NQ_ERR_UNSUPPORTED unless W % 4 == 0, ld % 4 == 0 and x < 4 GiB
 * (callers then run nq_conv_forward + nq_l2_loss_tanh_head). */
NQ_API int64_t nq_head_forward_loss_ws_floats(int B, int H, int W);
NQ_API int nq_head_forward_loss(const float* x, const float* wt, int ld, const float* bias, float* y, const float* tgt,
                         const uint8_t* cache_u8, const int64_t* idx, float* loss, float* dconv, float* db, float* ws, int B,
                         int Cin, int H, int W, int64_t mean_count, float gscale, nq_stream_t stream);

/* Per-frame PSNR pieces (utils.py:148-151): sse[f] = sum over one frame of (out-gt)^2, frames of frame_len floats. */
NQ_API int nq_frame_sse(const float* out, const float* gt, float* sse, int64_t frames, int64_t frame_len, nq_stream_t stream);

/* Frame gather: dst[i] = float(src_u8[idx[i]]) / 255 for frames of frame_len bytes (videosets/datasets.py:19-24);
 * idx is a device int64 array of n entries. */
NQ_API int nq_gather_frames_u8(const uint8_t* src, const int64_t* idx, float* dst, int64_t n, int64_t frame_len,
                        nq_stream_t stream);

/* ---- twice-differentiable elementwise pieces of the decoder (ABI v4): the Omega bit-allocation criterion differentiates
 * the decoder twice (Hessian-vector products by double backward, methods/bit_assign.py:57-118, 171-217), through
 * PixelShuffle, the exact-erf GELU (models/_layers.py:20-36, 104-105) and OutImg's tanh (models/_layers.py:10-16).
 *   nq_act_dd        y[i] = f(x[i]) * (g ? g[i] : 1) * (g2 ? g2[i] : 1);  mode 0 / 1 / 2: gelu, gelu', gelu''
 *                    (gelu'' = phi(x) (2 - x^2));  mode 3 / 4 / 5: t = tanh(x): 0.5 t + 0.5, 0.5 (1 - t^2), -t (1 - t^2)
 *   nq_pixel_shuffle inverse = 0: (B, C*r*r, H, W) -> (B, C, H*r, W*r) (torch.nn.PixelShuffle); inverse = 1: the un-shuffle
 *   nq_bias_add      y[b][c][p] = (x ? x[b][c][p] : 0) + bias[c]  (the bias term of F.conv2d, quant_layer.py:80; x = NULL
 *                    broadcasts the bias: the backward of a channel sum) */
NQ_API int nq_act_dd(const float* x, const float* g, const float* g2, float* y, int64_t n, int mode, nq_stream_t stream);
NQ_API int nq_pixel_shuffle(const float* x, float* y, int B, int C, int H, int W, int r, int inverse, nq_stream_t stream);
NQ_API int nq_bias_add(const float* x, const float* bias, float* y, int B, int C, int64_t HW, nq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NQ_HIP_H */
