// C-ABI entry points of the convolution side: tile selection + dispatch to the per-kernel-size
// implicit-GEMM / weight-gradient kernels, weight re-layout, split-K slab reduction.
#include <algorithm>

#include "nq_common.h"

extern "C" {
int nq_tiny_pw_supported(int B, int Cin, int H, int W, int Cout, int k);
int nq_tiny_pw_forward(const float* x, const float* wt, const float* bias, float* y, float* z, const float* zprev, int B,
                       int Cin, int H, int W, int Cout, int ld, int r, int epi, hipStream_t st);
int nq_tiny_pw_wgrad(const float* x, const float* dy, float* dw, float* db, int B, int Cin, int H, int W, int Cout,
                     hipStream_t st);
}

extern "C" {
int nq_conv_igemm_k1(const float*, const float*, const float*, float*, float*, int, int, int, int, int, int, int, int, int,
                     int, float*, int, const float*, hipStream_t);
int nq_conv_igemm_k3(const float*, const float*, const float*, float*, float*, int, int, int, int, int, int, int, int, int,
                     int, float*, int, const float*, hipStream_t);
int nq_conv_igemm_k5(const float*, const float*, const float*, float*, float*, int, int, int, int, int, int, int, int, int,
                     int, float*, int, const float*, hipStream_t);
int nq_conv_splitk_finish(const float*, const float*, float*, float*, const float*, int, int, int, int, int, int, int,
                          hipStream_t);
int nq_head_supported(int, int);
int nq_head_forward(const float*, const float*, int, const float*, float*, int, int, int, int, int, int, int, hipStream_t);
int nq_head_dgrad(const float*, const float*, int, const float*, float*, int, int, int, int, int, int, int, int, hipStream_t);
int nq_head_dgrad_streams(int, int, int, int, int, int, int, int);
int nq_conv_wgrad_k1(const float*, const float*, float*, float*, int, int, int, int, int, int, int, int, int, int,
                     int, hipStream_t);
int nq_conv_wgrad_k3(const float*, const float*, float*, float*, int, int, int, int, int, int, int, int, int, int,
                     int, hipStream_t);
int nq_conv_wgrad_k5(const float*, const float*, float*, float*, int, int, int, int, int, int, int, int, int, int,
                     int, hipStream_t);
}

namespace {

constexpr int kMiOptions[] = {11, 10, 9, 8, 6, 5, 4, 3, 2, 1};

inline int ci_per_slice(int k) { return k == 5 ? 4 : (k == 3 ? 8 : 16); }
inline bool ks_ok(int k) { return k == 1 || k == 3 || k == 5; }

// forward / data-gradient: fewest padded output channels, then the widest tile
inline int pick_mi_fwd(int Cout) {
  int best = 1, best_pad = 1 << 30;
  for (int mi : kMiOptions) {
    int mt = 16 * mi, pad = (Cout + mt - 1) / mt * mt;
    if (pad < best_pad) {
      best_pad = pad;
      best = mi;
    }
  }
  return best;
}

// split-K factor of the implicit-GEMM kernel: layers with few pixel tiles (dec0..dec3 and their data gradients)
// would otherwise occupy a handful of the 256 CUs with a very long K loop
inline int pick_nsplit(int B, int Cin, int H, int W, int Cout, int k) {
  const int mi = pick_mi_fwd(Cout);
  const int64_t base = (int64_t)((W + 31) / 32) * ((H + 3) / 4) * ((Cout + 16 * mi - 1) / (16 * mi)) * B;
  if (base >= 384) return 1;
  const int ncg = (Cin + ci_per_slice(k) - 1) / ci_per_slice(k);
  int ns = (int)((512 + base - 1) / base);
  if (ns > ncg) ns = ncg;
  if (ns > 32) ns = 32;
  return ns < 1 ? 1 : ns;
}

inline bool use_head_fwd(int Cout, int k, int epi, int in_gelu) {
  return nq_head_supported(Cout, k) && (epi == NQ_EPI_PLAIN || epi == NQ_EPI_TANH) && !in_gelu;
}
inline bool use_head_dgrad(int Cin, int k, int epi, int in_gelu, const float* bias) {
  return nq_head_supported(Cin, k) && (epi == NQ_EPI_PLAIN || epi == NQ_EPI_DGRAD_GELU) && !in_gelu && !bias;
}

inline int wgrad_ni(int mi, int k) { return k == 1 ? 2 : (mi <= 3 ? 6 : (mi <= 6 ? 4 : 3)); }

struct WgradPlan {
  int mi, ni, co_pad, n_pad, nsplit;
};

inline WgradPlan plan_wgrad(int B, int Cin, int H, int W, int Cout, int k) {
  WgradPlan p{};
  int64_t best_cost = INT64_MAX;
  const int N = Cin * k * k;
  for (int mi : kMiOptions) {
    int ni = wgrad_ni(mi, k);
    int mt = 16 * mi, nt = 64 * ni;
    int co_pad = (Cout + mt - 1) / mt * mt, n_pad = (N + nt - 1) / nt * nt;
    int64_t cost = (int64_t)co_pad * n_pad;
    if (cost < best_cost) {
      best_cost = cost;
      p.mi = mi;
      p.ni = ni;
      p.co_pad = co_pad;
      p.n_pad = n_pad;
    }
  }
  int tiles = (p.co_pad / (16 * p.mi)) * (p.n_pad / (64 * p.ni));
  int nseg = ((W + 31) / 32) * H * B;
  // one full wave of workgroups: 256 CUs x 2 resident workgroups (the kernels sit at 170-230 VGPRs), no ragged tail
  int ns = 512 / tiles;
  if (ns > nseg) ns = nseg;
  // (fewer, longer splits for problems with few segments -- HNeRV-3M dec2: 20 segments, 16 splits x 3.1 MB of slab -- were
  // tried: the reduction got 5 us shorter and the kernel 8 us longer)
  if (ns > 256) ns = 256;
  if (ns < 1) ns = 1;
  p.nsplit = ns;
  return p;
}

__global__ __launch_bounds__(256) void weight_fwd_layout_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                                int Cout, int K, int krows, int ld) {
  // wt[k][co] = w[co][k]  (k = (ci*KS+kh)*KS+kw is the OIHW inner index); 32x32 LDS transpose tile
  __shared__ float tile[32][33];
  const int k0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int co = c0 + j, k = k0 + tx;
    tile[j][tx] = (co < Cout && k < K) ? w[(int64_t)co * K + k] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int k = k0 + j, co = c0 + tx;
    if (k < krows && co < ld) wt[(int64_t)k * ld + co] = tile[tx][j];
  }
}

__global__ __launch_bounds__(256) void weight_bwd_layout_kernel(const float* __restrict__ w, float* __restrict__ wt,
                                                                int Cout, int Cin, int KK, int krows, int ld) {
  // wt[(co*KK + tap')][ci] = w[co][ci][KK-1-tap']   (both spatial taps flipped)
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)krows * ld) return;
  int row = (int)(i / ld), ci = (int)(i - (int64_t)row * ld);
  int co = row / KK, tap = row - co * KK;
  float v = 0.f;
  if (co < Cout && ci < Cin) v = w[((int64_t)co * Cin + ci) * KK + (KK - 1 - tap)];
  wt[i] = v;
}

// Both fp32 operands of several layers in ONE launch (the per-layer kernels above are launch-bound: five ~5 us launches per
// iteration for the layers that stay on the fp32 kernels).  Flat element index per segment; same values as the two
// kernels above.
constexpr int WL_MAXSEG = 16;
struct WLSeg {
  const float* w;
  float* wt;
  int Cout, Cin, KK, krows, ld, bwd;
};
struct WLMulti {
  WLSeg s[WL_MAXSEG];
  int blk0[WL_MAXSEG + 1];
  int nseg;
};
__global__ __launch_bounds__(256) void weight_layouts_multi_kernel(WLMulti t) {
  int k = 0;
  while (k + 1 < t.nseg && (int)blockIdx.x >= t.blk0[k + 1]) ++k;
  const WLSeg& g = t.s[k];
  const int64_t i = (int64_t)(blockIdx.x - t.blk0[k]) * 256 + threadIdx.x;
  if (i >= (int64_t)g.krows * g.ld) return;
  const int row = (int)(i / g.ld), col = (int)(i - (int64_t)row * g.ld);
  float v = 0.f;
  if (g.bwd) {   // wt[(co*KK + tap')][ci] = w[co][ci][KK-1-tap']
    const int co = row / g.KK, tap = row - co * g.KK;
    if (co < g.Cout && col < g.Cin) v = g.w[((int64_t)co * g.Cin + col) * g.KK + (g.KK - 1 - tap)];
  } else {       // wt[k][co] = w[co][k]
    if (col < g.Cout && row < g.Cin * g.KK) v = g.w[(int64_t)col * (g.Cin * g.KK) + row];
  }
  g.wt[i] = v;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ slab_db,
                                                           float* __restrict__ dw, float* __restrict__ db, int Cout, int N,
                                                           int co_pad, int n_pad, int nsplit) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t total = (int64_t)Cout * N;
  if (i < total) {
    int co = (int)(i / N), n = (int)(i - (int64_t)co * N);
    const float* p = slab + (int64_t)co * n_pad + n;
    const int64_t stride = (int64_t)co_pad * n_pad;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += p[k * stride];
    dw[i] = s;
  } else if (db && i < total + Cout) {
    int co = (int)(i - total);
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += slab_db[(int64_t)k * co_pad + co];
    db[co] = s;
  }
}

}  // namespace

extern "C" {

int nq_conv_operand_dims(int Cin, int Cout, int k, int* krows, int* ld) {
  if (!ks_ok(k) || Cin <= 0 || Cout <= 0 || !krows || !ld) return NQ_ERR_INVALID;
  int ci = ci_per_slice(k);
  *krows = (Cin + ci - 1) / ci * ci * k * k;
  int mt = 16 * pick_mi_fwd(Cout);
  *ld = (Cout + mt - 1) / mt * mt;
  return NQ_OK;
}

int nq_weight_layouts(const float* w, float* wt_fwd, float* wt_bwd, int Cout, int Cin, int k, int krows_fwd, int ld_fwd,
                      int krows_bwd, int ld_bwd, nq_stream_t stream) {
  if (!w || Cout <= 0 || Cin <= 0 || k <= 0) return NQ_ERR_INVALID;
  const int KK = k * k;
  if (wt_fwd) {
    if (krows_fwd < Cin * KK || ld_fwd < Cout) return NQ_ERR_INVALID;
    dim3 g((unsigned)((krows_fwd + 31) / 32), (unsigned)((ld_fwd + 31) / 32));
    hipLaunchKernelGGL(weight_fwd_layout_kernel, g, dim3(256), 0, nq_s(stream), w, wt_fwd, Cout, Cin * KK, krows_fwd,
                       ld_fwd);
  }
  if (wt_bwd) {
    if (krows_bwd < Cout * KK || ld_bwd < Cin) return NQ_ERR_INVALID;
    int64_t total = (int64_t)krows_bwd * ld_bwd;
    hipLaunchKernelGGL(weight_bwd_layout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nq_s(stream), w,
                       wt_bwd, Cout, Cin, KK, krows_bwd, ld_bwd);
  }
  return nq_launch_status();
}

int nq_weight_layouts_multi(const nq_wl_seg* segs, int nseg, nq_stream_t stream) {
  if (!segs || nseg <= 0) return NQ_ERR_INVALID;
  WLMulti t;
  t.nseg = 0;
  int blocks = 0;
  auto flush = [&]() {
    if (t.nseg == 0) return;
    t.blk0[t.nseg] = blocks;
    hipLaunchKernelGGL(weight_layouts_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, nq_s(stream), t);
    t.nseg = 0;
    blocks = 0;
  };
  for (int i = 0; i < nseg; ++i) {
    const nq_wl_seg& h = segs[i];
    if (!h.w || h.Cout <= 0 || h.Cin <= 0 || h.k <= 0) return NQ_ERR_INVALID;
    const int KK = h.k * h.k;
    for (int bwd = 0; bwd < 2; ++bwd) {
      float* wt = bwd ? h.wt_bwd : h.wt_fwd;
      if (!wt) continue;
      const int krows = bwd ? h.krows_bwd : h.krows_fwd, ld = bwd ? h.ld_bwd : h.ld_fwd;
      if (bwd ? (krows < h.Cout * KK || ld < h.Cin) : (krows < h.Cin * KK || ld < h.Cout)) return NQ_ERR_INVALID;
      if (t.nseg == WL_MAXSEG) flush();
      t.s[t.nseg] = WLSeg{h.w, wt, h.Cout, h.Cin, KK, krows, ld, bwd};
      t.blk0[t.nseg] = blocks;
      blocks += (int)(((int64_t)krows * ld + 255) / 256);
      ++t.nseg;
    }
  }
  flush();
  return nq_launch_status();
}

int64_t nq_conv_forward_ws_floats(int B, int Cin, int H, int W, int Cout, int k) {
  if (!ks_ok(k) || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  if (nq_head_supported(Cout, k) || nq_head_supported(Cin, k)) return 0;
  int ns = pick_nsplit(B, Cin, H, W, Cout, k);
  return ns > 1 ? (int64_t)ns * B * Cout * H * W : 0;
}

// 1 when nq_conv_forward can write its output y as split {hi | lo} words for this call (include/nq_hip.h)
int nq_conv_split_out(int B, int Cin, int H, int W, int Cout, int k, int r, int epilogue, int in_gelu, int has_bias) {
  if (!ks_ok(k) || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  if (use_head_fwd(Cout, k, epilogue, in_gelu)) return 0;
  if (!(nq_head_supported(Cin, k) && (epilogue == NQ_EPI_PLAIN || epilogue == NQ_EPI_DGRAD_GELU) && !in_gelu && !has_bias)) return 0;
  return nq_head_dgrad_streams(B, Cout, H, W, Cin, k, epilogue == NQ_EPI_DGRAD_GELU ? r : 1, epilogue == NQ_EPI_DGRAD_GELU);
}

int nq_conv_forward(const float* x, const float* wt, const float* bias, float* y, float* z, float* ws, int B, int Cin, int H,
                    int W, int Cout, int k, int krows, int ld, int r, int epilogue, int in_gelu, const float* zprev,
                    nq_stream_t stream) {
  if (!x || !wt || (!y && epilogue != NQ_EPI_PS) || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0)
    return NQ_ERR_INVALID;
  if (!ks_ok(k)) return NQ_ERR_UNSUPPORTED;
  // NQ_EPI_Y_SPLIT: the output as split {hi | lo} words -- only the streaming head data gradient offers it (nq_conv_split_out)
  const int y_split = (epilogue & NQ_EPI_Y_SPLIT) ? 1 : 0;
  if (epilogue & NQ_EPI_X_SPLIT) return NQ_ERR_UNSUPPORTED;
  epilogue &= ~NQ_EPI_Y_SPLIT;
  if (y_split && !nq_conv_split_out(B, Cin, H, W, Cout, k, r, epilogue, in_gelu, bias != nullptr)) return NQ_ERR_UNSUPPORTED;
  if (epilogue < 0 || epilogue > NQ_EPI_DGRAD_GELU) return NQ_ERR_INVALID;
  if ((epilogue == NQ_EPI_PS_GELU || epilogue == NQ_EPI_PS) && (!z || r <= 0 || Cout % (r * r) != 0)) return NQ_ERR_INVALID;
  if (epilogue == NQ_EPI_DGRAD_GELU && (!zprev || r <= 0 || H % r != 0 || W % r != 0)) return NQ_ERR_INVALID;
  int need_rows, need_ld;
  nq_conv_operand_dims(Cin, Cout, k, &need_rows, &need_ld);
  if (krows < need_rows || ld < need_ld || (ld & 3)) return NQ_ERR_INVALID;
  if (B > 65535) return NQ_ERR_UNSUPPORTED;
  hipStream_t st0 = nq_s(stream);
  // <= 4 output channels (the decoder head) / <= 4 input channels (its data gradient): HBM-bound vector kernels
  if (use_head_fwd(Cout, k, epilogue, in_gelu))
    return nq_head_forward(x, wt, ld, bias, y, B, Cin, H, W, Cout, k, epilogue, st0);
  if (use_head_dgrad(Cin, k, epilogue, in_gelu, bias))
    return nq_head_dgrad(x, wt, ld, epilogue == NQ_EPI_DGRAD_GELU ? zprev : nullptr, y, B, Cout, H, W, Cin, k,
                         epilogue == NQ_EPI_DGRAD_GELU ? r : 1, y_split, st0);
  // 1x1 convolutions over a handful of pixels (decoder stem / first block): compact thread-per-output kernel
  if (!in_gelu && nq_tiny_pw_supported(B, Cin, H, W, Cout, k))
    return nq_tiny_pw_forward(x, wt, bias, y, z, zprev, B, Cin, H, W, Cout, ld, r, epilogue, st0);
  const int mi = pick_mi_fwd(Cout);
  const int ns = pick_nsplit(B, Cin, H, W, Cout, k);
  if (ns > 1 && !ws) return NQ_ERR_INVALID;
  if ((int64_t)B * ns > 65535) return NQ_ERR_UNSUPPORTED;
  hipStream_t st = nq_s(stream);
  int rc;
  switch (k) {
    case 1: rc = nq_conv_igemm_k1(x, wt, bias, y, z, B, Cin, H, W, Cout, ld, r, epilogue, mi, ns, ws, in_gelu, zprev, st); break;
    case 3: rc = nq_conv_igemm_k3(x, wt, bias, y, z, B, Cin, H, W, Cout, ld, r, epilogue, mi, ns, ws, in_gelu, zprev, st); break;
    default: rc = nq_conv_igemm_k5(x, wt, bias, y, z, B, Cin, H, W, Cout, ld, r, epilogue, mi, ns, ws, in_gelu, zprev, st); break;
  }
  if (rc != NQ_OK || ns == 1) return rc;
  return nq_conv_splitk_finish(ws, bias, y, z, zprev, B, H, W, Cout, r, epilogue, ns, st);
}

int64_t nq_conv_wgrad_ws_floats(int B, int Cin, int H, int W, int Cout, int k) {
  if (!ks_ok(k) || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  WgradPlan p = plan_wgrad(B, Cin, H, W, Cout, k);
  return (int64_t)p.nsplit * p.co_pad * ((int64_t)p.n_pad + 1);
}

static int conv_wgrad_impl(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W,
                           int Cout, int k, int x_gelu, nq_wgr_seg* seg, nq_stream_t stream);

int nq_conv_wgrad(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W, int Cout,
                  int k, int x_gelu, nq_stream_t stream) {
  return conv_wgrad_impl(x, dy, dw, db, ws, B, Cin, H, W, Cout, k, x_gelu, nullptr, stream);
}

int nq_conv_wgrad_slabs(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W, int Cout,
                        int k, int x_gelu, nq_wgr_seg* seg, nq_stream_t stream) {
  if (!seg) return NQ_ERR_INVALID;
  return conv_wgrad_impl(x, dy, dw, db, ws, B, Cin, H, W, Cout, k, x_gelu, seg, stream);
}

static int conv_wgrad_impl(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W,
                           int Cout, int k, int x_gelu, nq_wgr_seg* seg, nq_stream_t stream) {
  if (!x || !dy || !dw || !ws || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return NQ_ERR_INVALID;
  if (!ks_ok(k)) return NQ_ERR_UNSUPPORTED;
  if (seg) *seg = nq_wgr_seg{nullptr, nullptr, dw, db, Cout, Cin * k * k, 0, 0, 0, 0, 1};
  if (!x_gelu && nq_tiny_pw_supported(B, Cin, H, W, Cout, k))
    return nq_tiny_pw_wgrad(x, dy, dw, db, B, Cin, H, W, Cout, nq_s(stream));
  WgradPlan p = plan_wgrad(B, Cin, H, W, Cout, k);
  float* slab = ws;
  float* slab_db = ws + (int64_t)p.nsplit * p.co_pad * p.n_pad;
  hipStream_t st = nq_s(stream);
  int rc;
  switch (k) {
    case 1: rc = nq_conv_wgrad_k1(x, dy, slab, slab_db, B, Cin, H, W, Cout, p.co_pad, p.n_pad, p.nsplit, p.mi, p.ni, x_gelu, st); break;
    case 3: rc = nq_conv_wgrad_k3(x, dy, slab, slab_db, B, Cin, H, W, Cout, p.co_pad, p.n_pad, p.nsplit, p.mi, p.ni, x_gelu, st); break;
    default: rc = nq_conv_wgrad_k5(x, dy, slab, slab_db, B, Cin, H, W, Cout, p.co_pad, p.n_pad, p.nsplit, p.mi, p.ni, x_gelu, st); break;
  }
  if (rc != NQ_OK) return rc;
  const int N = Cin * k * k;
  if (seg) {   // deferred: the sequential fixed-order sum over the splits (sg = 1) runs in nq_wgrad_reduce_multi
    *seg = nq_wgr_seg{slab, db ? slab_db : nullptr, dw, db, Cout, N, p.co_pad, p.n_pad, p.nsplit, 0, 1};
    return NQ_OK;
  }
  int64_t total = (int64_t)Cout * N + Cout;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, slab, slab_db, dw, db, Cout,
                     N, p.co_pad, p.n_pad, p.nsplit);
  return nq_launch_status();
}

}  // extern "C"
