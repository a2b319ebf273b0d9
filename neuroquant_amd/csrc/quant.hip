// Parameter-side kernels: scale init, uniform-affine and AdaRound fake-quant (forward / backward),
// rounding regulariser, Adam.  All HBM-bound elementwise / row-reduction work; compiled with
// -ffp-contract=off so that the arithmetic is the reference's op-by-op fp32 sequence
// (quantization/quantizer.py, calib_model.py:39-47).
#include "nq_common.h"
#include "quant_elem.h"

namespace {

constexpr int TPB = 256;
constexpr int EPT = 4;  // elements per thread per block chunk

// ------------------------------------------------------------------------------------------------
// scale init: one workgroup per row (quantizer.py:156-168)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void scale_init_kernel(const float* __restrict__ x, int64_t row_len, int n_levels,
                                                         float* __restrict__ delta, float* __restrict__ zp) {
  __shared__ float smin[4], smax[4];
  const float* xr = x + (int64_t)blockIdx.x * row_len;
  float mn = 0.f, mx = 0.f;  // min(x_min, 0), max(x_max, 0)
  for (int64_t i = threadIdx.x; i < row_len; i += TPB) {
    float v = xr[i];
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
  mn = nq_wave_min(mn);
  mx = nq_wave_max(mx);
  if ((threadIdx.x & 63) == 0) {
    smin[threadIdx.x >> 6] = mn;
    smax[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < TPB / 64; ++i) {
      mn = fminf(mn, smin[i]);
      mx = fmaxf(mx, smax[i]);
    }
    // Python-double division, then cast to fp32 (quantizer.py:163)
    float d = (float)(((double)mx - (double)mn) / (double)(n_levels - 1));
    d = fmaxf(d, 1e-8f);
    delta[blockIdx.x] = d;
    zp[blockIdx.x] = rintf((-mn) / d);
  }
}

// ------------------------------------------------------------------------------------------------
// UAQ forward / backward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void uaq_fwd_kernel(const float* __restrict__ x, const float* __restrict__ delta,
                                                      const float* __restrict__ zp, float* __restrict__ y,
                                                      int64_t row_len, int per_row, float qmax, unsigned bpr) {
  const int64_t row = blockIdx.x / bpr;   // rows are folded into grid.x (no 65535-row limit of grid.y)
  const float d = per_row ? delta[row] : delta[0];
  const float z = per_row ? zp[row] : zp[0];
  const int64_t base = row * row_len;
  int64_t i = (int64_t)(blockIdx.x - row * bpr) * (TPB * EPT) + threadIdx.x;
#pragma unroll
  for (int e = 0; e < EPT; ++e, i += TPB) {
    if (i < row_len) {
      float xi = rintf(x[base + i] / d) + z;
      float xq = fminf(fmaxf(xi, 0.f), qmax);
      y[base + i] = (xq - z) * d;
    }
  }
}

__global__ __launch_bounds__(TPB) void uaq_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                      const float* __restrict__ delta, const float* __restrict__ zp,
                                                      float* __restrict__ ddelta, float* __restrict__ dx,
                                                      int64_t row_len, int per_row, float qmax) {
  __shared__ float red[16];
  const int64_t row = blockIdx.x;
  const float d = per_row ? delta[row] : delta[0];
  const float z = per_row ? zp[row] : zp[0];
  const int64_t base = row * row_len;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < row_len; i += TPB) {
    float u = x[base + i] / d;
    float xi = rintf(u) + z;
    float xq = fminf(fmaxf(xi, 0.f), qmax);
    float inside = (xi >= 0.f && xi <= qmax) ? 1.f : 0.f;
    const float g = gy[base + i];
    acc += g * ((xq - z) - inside * u);
    // round_ste (quantizer.py:53-57, 117-119): autograd multiplies by delta (dequantisation), masks by the clamp and
    // divides by delta again (x/delta): (g*d)/d, not g, in fp32
    if (dx) dx[base + i] = inside != 0.f ? (g * d) / d : 0.f;
  }
  float s = nq_block_sum(acc, red);
  if (threadIdx.x == 0) ddelta[row] = s;
}

// ------------------------------------------------------------------------------------------------
// AdaRound
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float half_round_trip(float v) { return (float)(_Float16)v; }  // round-to-nearest-even, like Tensor.half()

__global__ void adaround_scale_kernel(const float* __restrict__ din, const float* __restrict__ zin,
                                      float* __restrict__ dout, float* __restrict__ zout, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    dout[i] = half_round_trip(din[i]);  // quantizer.py:264
    zout[i] = half_round_trip(zin[i]);  // quantizer.py:265
  }
}

__global__ __launch_bounds__(TPB) void adaround_alpha_init_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ delta,
                                                                  float* __restrict__ alpha, int64_t row_len,
                                                                  int per_row, unsigned bpr) {
  const int64_t row = blockIdx.x / bpr;
  const float d = per_row ? delta[row] : delta[0];
  const int64_t base = row * row_len;
  int64_t i = (int64_t)(blockIdx.x - row * bpr) * (TPB * EPT) + threadIdx.x;
#pragma unroll
  for (int e = 0; e < EPT; ++e, i += TPB) {
    if (i < row_len) {
      float u = x[base + i] / d;
      float rest = u - floorf(u);
      alpha[base + i] = -logf((NQ_ZETA - NQ_GAMMA) / (rest - NQ_GAMMA) - 1.0f);  // quantizer.py:312
    }
  }
}

__global__ __launch_bounds__(TPB) void adaround_fwd_kernel(const float* __restrict__ x, const float* __restrict__ alpha,
                                                           const float* __restrict__ delta,
                                                           const float* __restrict__ zp, float* __restrict__ y,
                                                           float* __restrict__ xq_out, int64_t row_len, int per_row,
                                                           float qmax, int soft, unsigned bpr) {
  const int64_t row = blockIdx.x / bpr;   // rows are folded into grid.x (no 65535-row limit of grid.y)
  const float d = per_row ? delta[row] : delta[0];
  const float z = per_row ? zp[row] : zp[0];
  const int64_t base = row * row_len;
  int64_t i = (int64_t)(blockIdx.x - row * bpr) * (TPB * EPT) + threadIdx.x;
#pragma unroll
  for (int e = 0; e < EPT; ++e, i += TPB) {
    if (i < row_len) {
      float xq;
      y[base + i] = ada_fwd_elem(x[base + i], alpha[base + i], d, z, qmax, soft, xq);
      if (xq_out) xq_out[base + i] = xq;
    }
  }
}

__global__ __launch_bounds__(TPB) void adaround_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                           const float* __restrict__ alpha,
                                                           const float* __restrict__ delta,
                                                           const float* __restrict__ zp, float* __restrict__ dalpha,
                                                           int64_t row_len, int per_row, float qmax, float reg_weight,
                                                           float reg_b, unsigned bpr) {
  const int64_t row = blockIdx.x / bpr;   // rows are folded into grid.x (no 65535-row limit of grid.y)
  const float d = per_row ? delta[row] : delta[0];
  const float z = per_row ? zp[row] : zp[0];
  const int64_t base = row * row_len;
  int64_t i = (int64_t)(blockIdx.x - row * bpr) * (TPB * EPT) + threadIdx.x;
#pragma unroll
  for (int e = 0; e < EPT; ++e, i += TPB) {
    if (i < row_len) {
      float g = ada_bwd_elem(x[base + i], gy[base + i], alpha[base + i], d, z, qmax, reg_weight, reg_b);
      dalpha[base + i] = g;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// deterministic two-stage sums
// ------------------------------------------------------------------------------------------------
constexpr int RED_CHUNK = NQ_RED_CHUNK;

__global__ __launch_bounds__(TPB) void round_loss_stage1(const float* __restrict__ alpha, int64_t n, float b,
                                                         float* __restrict__ ws) {
  __shared__ float red[16];
  int64_t i0 = (int64_t)blockIdx.x * RED_CHUNK;
  float acc = 0.f;
  for (int k = 0; k < 16; ++k) {
    int64_t i = i0 + k * TPB + threadIdx.x;
    if (i < n) {
      float h = fminf(fmaxf(soft_target_lin(alpha[i]), 0.f), 1.f);
      acc += 1.f - powf(fabsf(h - 0.5f) * 2.f, b);
    }
  }
  float s = nq_block_sum(acc, red);
  if (threadIdx.x == 0) ws[blockIdx.x] = s;
}

__global__ __launch_bounds__(TPB) void round_loss_bwd_kernel(const float* __restrict__ alpha, int64_t n, float b,
                                                             float weight, const float* __restrict__ gscale,
                                                             float* __restrict__ dalpha, int accumulate) {
  int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  float s = nq_sigmoid(alpha[i]);
  float lin = s * (NQ_ZETA - NQ_GAMMA) + NQ_GAMMA;
  float h = fminf(fmaxf(lin, 0.f), 1.f);
  float hp = (lin >= 0.f && lin <= 1.f) ? (NQ_ZETA - NQ_GAMMA) * (s * (1.f - s)) : 0.f;
  float c = h - 0.5f;
  float sg = (c > 0.f) ? 1.f : ((c < 0.f) ? -1.f : 0.f);
  float g = -(weight * (gscale ? gscale[0] : 1.f)) * (b * powf(fabsf(c) * 2.f, b - 1.f)) * 2.f * sg * hp;
  dalpha[i] = accumulate ? dalpha[i] + g : g;
}

// ------------------------------------------------------------------------------------------------
// Adam (torch/optim/adam.py _single_tensor_adam, no weight decay, no amsgrad)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void adam_elem(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                          float* __restrict__ v, int64_t i, float step_size, float beta1, float beta2,
                                          float eps, float bc2_sqrt) {
  float gi = g[i];
  float mi = m[i];
  mi = mi + (1.f - beta1) * (gi - mi);         // exp_avg.lerp_(grad, 1-beta1)
  float vi = v[i] * beta2 + (1.f - beta2) * (gi * gi);  // mul_(beta2).addcmul_(g, g, value=1-beta2)
  m[i] = mi;
  v[i] = vi;
  float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] - step_size * (mi / denom);      // addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ __launch_bounds__(TPB) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float step_size, float beta1, float beta2, float eps,
                                                   float bc2_sqrt) {
  int64_t i = (int64_t)blockIdx.x * TPB + threadIdx.x;
  if (i >= n) return;
  adam_elem(p, g, m, v, i, step_size, beta1, beta2, eps, bc2_sqrt);
}


// ------------------------------------------------------------------------------------------------
// multi-tensor variants: ONE launch for all layers' weight and bias tensors (14 per decoder).  The segment table
// travels as a kernel argument (<= 16 segments, ~1.5 KB); a block finds its segment by a short scan of the block
// prefix, then works on 1024 consecutive elements of that tensor.  Per-element arithmetic = the single-tensor kernels'.
// ------------------------------------------------------------------------------------------------
constexpr int MAXSEG = 16;
struct AdaSegD {
  const float* x;
  const float* gy;
  const float* alpha;
  const float* delta;
  const float* zp;
  float* out;
  int64_t n;
  int row_len, per_row, soft;
  float qmax, reg_weight;
  int vec;  // every pointer of the segment is 16-byte aligned -> 16-byte accesses
};
struct AdaMulti {
  AdaSegD s[MAXSEG];
  int blk0[MAXSEG + 1];
  int nseg;
  float reg_b;
  const float* dyn;  // optional device scalars of the current step (nq_step_prologue): [0] = reg_b, [1] = regulariser gate
};
struct AdamSegD {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
  int vec;
};
struct AdamMulti {
  AdamSegD s[MAXSEG];
  int blk0[MAXSEG + 1];
  int nseg;
};

template <class T>
__device__ __forceinline__ int find_seg(const T& t, int blk) {
  int k = 0;
  while (k + 1 < t.nseg && blk >= t.blk0[k + 1]) ++k;
  return k;
}

// A thread owns 4 CONSECUTIVE elements (one 16-byte load / store per tensor when the segment's pointers are 16-byte
// aligned: these kernels are bound by the number of memory instructions, 23 -> 10 us for the 2.65 M elements of HNeRV-3M);
// per-element arithmetic unchanged.
template <bool BWD>
__global__ __launch_bounds__(TPB) void adaround_multi_kernel(AdaMulti t) {
  const int k = find_seg(t, blockIdx.x);
  const AdaSegD& sg = t.s[k];
  const int64_t i0 = (int64_t)(blockIdx.x - t.blk0[k]) * (TPB * EPT) + 4 * threadIdx.x;
  if (i0 >= sg.n) return;
  const bool vec = sg.vec && (i0 + 3 < sg.n);
  float xv[4], av[4], gv[4] = {0.f, 0.f, 0.f, 0.f}, ov[4];
  if (vec) {
    const float4 x4 = *reinterpret_cast<const float4*>(sg.x + i0), a4 = *reinterpret_cast<const float4*>(sg.alpha + i0);
    xv[0] = x4.x; xv[1] = x4.y; xv[2] = x4.z; xv[3] = x4.w;
    av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w;
    if (BWD) {
      const float4 g4 = *reinterpret_cast<const float4*>(sg.gy + i0);
      gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool in = i0 + j < sg.n;
      xv[j] = in ? sg.x[i0 + j] : 0.f;
      av[j] = in ? sg.alpha[i0 + j] : 0.f;
      if (BWD) gv[j] = in ? sg.gy[i0 + j] : 0.f;
    }
  }
  const int64_t row0 = sg.per_row ? i0 / sg.row_len : 0;
  const int rem0 = sg.per_row ? (int)(i0 - row0 * sg.row_len) : 0;
  const float rb = (BWD && t.dyn) ? t.dyn[0] : t.reg_b;
  const float rw = (BWD && t.dyn) ? sg.reg_weight * t.dyn[1] : sg.reg_weight;   // gate is exactly 0 or 1
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int64_t row = 0;
    if (sg.per_row) row = (sg.row_len >= 4) ? row0 + ((rem0 + j >= sg.row_len) ? 1 : 0) : (i0 + j) / sg.row_len;
    if (i0 + j >= sg.n) row = 0;
    const float d = sg.delta[row], z = sg.zp[row];
    if (BWD) {
      ov[j] = ada_bwd_elem(xv[j], gv[j], av[j], d, z, sg.qmax, rw, rb);
    } else {
      float xq;
      ov[j] = ada_fwd_elem(xv[j], av[j], d, z, sg.qmax, sg.soft, xq);
    }
  }
  if (vec) {
    *reinterpret_cast<float4*>(sg.out + i0) = make_float4(ov[0], ov[1], ov[2], ov[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i0 + j < sg.n) sg.out[i0 + j] = ov[j];
  }
}

__global__ __launch_bounds__(TPB) void adam_multi_kernel(AdamMulti t, float step_size, float beta1, float beta2, float eps,
                                                         float bc2_sqrt, const float* __restrict__ dyn) {
  if (dyn) {  // per-step scalars from device memory (graph replays): [2] = lr/(1-beta1^t), [3] = sqrt(1-beta2^t)
    step_size = dyn[2];
    bc2_sqrt = dyn[3];
  }
  const int k = find_seg(t, blockIdx.x);
  const AdamSegD& sg = t.s[k];
  const int64_t i0 = (int64_t)(blockIdx.x - t.blk0[k]) * (TPB * EPT) + 4 * threadIdx.x;
  if (i0 >= sg.n) return;
  if (sg.vec && i0 + 3 < sg.n) {   // same per-element arithmetic as adam_elem, four elements per 16-byte access
    const float4 g4 = *reinterpret_cast<const float4*>(sg.g + i0);
    float4 m4 = *reinterpret_cast<const float4*>(sg.m + i0), v4 = *reinterpret_cast<const float4*>(sg.v + i0);
    float4 p4 = *reinterpret_cast<const float4*>(sg.p + i0);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    float m[4] = {m4.x, m4.y, m4.z, m4.w}, v[4] = {v4.x, v4.y, v4.z, v4.w}, p[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[j] = m[j] + (1.f - beta1) * (g[j] - m[j]);
      v[j] = v[j] * beta2 + (1.f - beta2) * (g[j] * g[j]);
      const float denom = sqrtf(v[j]) / bc2_sqrt + eps;
      p[j] = p[j] - step_size * (m[j] / denom);
    }
    *reinterpret_cast<float4*>(sg.m + i0) = make_float4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<float4*>(sg.v + i0) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(sg.p + i0) = make_float4(p[0], p[1], p[2], p[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i0 + j < sg.n) adam_elem(sg.p, sg.g, sg.m, sg.v, i0 + j, step_size, beta1, beta2, eps, bc2_sqrt);
  }
}

// d(alpha) (+ regulariser gradient) and the Adam update of alpha in ONE pass (round 3): alpha is both the input of the
// fake-quant backward and Adam's parameter, so d(alpha) never goes to memory (x, gy, alpha, m, v: 85 MB per HNeRV-3M
// iteration instead of 117 MB in two launches).  Per element exactly ada_bwd_elem followed by adam_elem's arithmetic:
// alpha / m / v are bit-identical to nq_adaround_backward_multi + nq_adam_step_multi.
struct AdaAdamSegD {
  const float* x;
  const float* gy;
  float* alpha;
  const float* delta;
  const float* zp;
  float* m;
  float* v;
  int64_t n;
  int row_len, per_row;
  float qmax, reg_weight;
  int vec;
};
struct AdaAdamMulti {
  AdaAdamSegD s[MAXSEG];
  int blk0[MAXSEG + 1];
  int nseg;
};
__global__ __launch_bounds__(TPB) void adaround_adam_multi_kernel(AdaAdamMulti t, float reg_b, float step_size, float beta1,
                                                                  float beta2, float eps, float bc2_sqrt,
                                                                  const float* __restrict__ dyn) {
  float gate = 1.f;
  if (dyn) {   // per-step scalars from device memory (graph replays): {reg_b, regulariser gate, lr/(1-beta1^t), sqrt(1-beta2^t)}
    reg_b = dyn[0];
    gate = dyn[1];
    step_size = dyn[2];
    bc2_sqrt = dyn[3];
  }
  const int k = find_seg(t, blockIdx.x);
  const AdaAdamSegD& sg = t.s[k];
  const int64_t i0 = (int64_t)(blockIdx.x - t.blk0[k]) * (TPB * EPT) + 4 * threadIdx.x;
  if (i0 >= sg.n) return;
  const bool vec = sg.vec && (i0 + 3 < sg.n);
  float xv[4], av[4], gv[4], mv[4], vv[4];
  if (vec) {
    const float4 x4 = *reinterpret_cast<const float4*>(sg.x + i0), a4 = *reinterpret_cast<const float4*>(sg.alpha + i0);
    const float4 g4 = *reinterpret_cast<const float4*>(sg.gy + i0);
    const float4 m4 = *reinterpret_cast<const float4*>(sg.m + i0), v4 = *reinterpret_cast<const float4*>(sg.v + i0);
    xv[0] = x4.x; xv[1] = x4.y; xv[2] = x4.z; xv[3] = x4.w;
    av[0] = a4.x; av[1] = a4.y; av[2] = a4.z; av[3] = a4.w;
    gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
    mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
    vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool in = i0 + j < sg.n;
      xv[j] = in ? sg.x[i0 + j] : 0.f;
      av[j] = in ? sg.alpha[i0 + j] : 0.f;
      gv[j] = in ? sg.gy[i0 + j] : 0.f;
      mv[j] = in ? sg.m[i0 + j] : 0.f;
      vv[j] = in ? sg.v[i0 + j] : 0.f;
    }
  }
  const int64_t row0 = sg.per_row ? i0 / sg.row_len : 0;
  const int rem0 = sg.per_row ? (int)(i0 - row0 * sg.row_len) : 0;
  const float rw = dyn ? sg.reg_weight * gate : sg.reg_weight;   // gate is exactly 0 or 1
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int64_t row = 0;
    if (sg.per_row) row = (sg.row_len >= 4) ? row0 + ((rem0 + j >= sg.row_len) ? 1 : 0) : (i0 + j) / sg.row_len;
    if (i0 + j >= sg.n) row = 0;
    const float g = ada_bwd_elem(xv[j], gv[j], av[j], sg.delta[row], sg.zp[row], sg.qmax, rw, reg_b);
    mv[j] = mv[j] + (1.f - beta1) * (g - mv[j]);
    vv[j] = vv[j] * beta2 + (1.f - beta2) * (g * g);
    const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
    av[j] = av[j] - step_size * (mv[j] / denom);
  }
  if (vec) {
    *reinterpret_cast<float4*>(sg.m + i0) = make_float4(mv[0], mv[1], mv[2], mv[3]);
    *reinterpret_cast<float4*>(sg.v + i0) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    *reinterpret_cast<float4*>(sg.alpha + i0) = make_float4(av[0], av[1], av[2], av[3]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i0 + j < sg.n) {
        sg.m[i0 + j] = mv[j];
        sg.v[i0 + j] = vv[j];
        sg.alpha[i0 + j] = av[j];
      }
  }
}

// UAQ fake-quant of several tensors in ONE launch (phase 1 of the calibration, calib_model.py:119-165): the forward is the
// per-element arithmetic of uaq_fwd_kernel, the backward one workgroup per reduction row with uaq_bwd_kernel's loop and
// block sum -- bit-identical to the single-tensor launches.
struct UaqSegD {
  const float* x;
  const float* gy;      // backward
  const float* delta;
  const float* zp;
  float* out;           // forward: y; backward: d(delta)
  int64_t n;            // forward: elements
  int64_t rows;         // backward: reduction rows
  int row_len, per_row;
  float qmax;
};
struct UaqMulti {
  UaqSegD s[MAXSEG];
  int blk0[MAXSEG + 1];
  int nseg;
};
__global__ __launch_bounds__(TPB) void uaq_fwd_multi_kernel(UaqMulti t) {
  const int k = find_seg(t, blockIdx.x);
  const UaqSegD& sg = t.s[k];
  int64_t i = (int64_t)(blockIdx.x - t.blk0[k]) * (TPB * EPT) + threadIdx.x;
#pragma unroll
  for (int e = 0; e < EPT; ++e, i += TPB) {
    if (i < sg.n) {
      const int64_t row = sg.per_row ? i / sg.row_len : 0;
      const float d = sg.delta[row], z = sg.zp[row];
      float xi = rintf(sg.x[i] / d) + z;
      float xq = fminf(fmaxf(xi, 0.f), sg.qmax);
      sg.out[i] = (xq - z) * d;
    }
  }
}
__global__ __launch_bounds__(TPB) void uaq_bwd_multi_kernel(UaqMulti t) {
  __shared__ float red[16];
  const int k = find_seg(t, blockIdx.x);
  const UaqSegD& sg = t.s[k];
  const int64_t row = blockIdx.x - t.blk0[k];
  const float d = sg.delta[row], z = sg.zp[row];
  const int64_t base = row * sg.row_len;
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < sg.row_len; i += TPB) {
    float u = sg.x[base + i] / d;
    float xi = rintf(u) + z;
    float xq = fminf(fmaxf(xi, 0.f), sg.qmax);
    float inside = (xi >= 0.f && xi <= sg.qmax) ? 1.f : 0.f;
    acc += sg.gy[base + i] * ((xq - z) - inside * u);
  }
  float s = nq_block_sum(acc, red);
  if (threadIdx.x == 0) sg.out[row] = s;
}

inline unsigned blocks_per_row(int64_t row_len) { return (unsigned)((row_len + TPB * EPT - 1) / (TPB * EPT)); }
inline dim3 row_grid(int64_t rows, int64_t row_len) { return dim3((unsigned)(rows * blocks_per_row(row_len)), 1, 1); }
// rows x blocks-per-row must fit grid.x (2^31 - 1); any channel count a decoder can have does
inline bool bad_rows(int64_t rows, int64_t row_len) {
  return rows <= 0 || row_len <= 0 || rows * (int64_t)blocks_per_row(row_len) > 0x7fffffffLL;
}

}  // namespace

extern "C" {

int nq_abi_version(void) { return 5; }

const char* nq_error_string(int code) {
  switch (code) {
    case NQ_OK: return "ok";
    case NQ_ERR_INVALID: return "invalid argument";
    case NQ_ERR_UNSUPPORTED: return "unsupported shape";
    case NQ_ERR_LAUNCH: return "HIP launch error";
    default: return "unknown error";
  }
}

int nq_scale_init_max(const float* x, int64_t rows, int64_t row_len, int n_levels, float* delta, float* zp,
                      nq_stream_t stream) {
  if (!x || !delta || !zp || rows <= 0 || row_len <= 0 || n_levels < 2) return NQ_ERR_INVALID;
  hipLaunchKernelGGL(scale_init_kernel, dim3((unsigned)rows), dim3(TPB), 0, nq_s(stream), x, row_len, n_levels, delta,
                     zp);
  return nq_launch_status();
}

int nq_uaq_forward(const float* x, const float* delta, const float* zp, float* y, int64_t rows, int64_t row_len,
                   int per_row, int n_levels, nq_stream_t stream) {
  if (!x || !delta || !zp || !y || bad_rows(rows, row_len)) return NQ_ERR_INVALID;
  hipLaunchKernelGGL(uaq_fwd_kernel, row_grid(rows, row_len), dim3(TPB), 0, nq_s(stream), x, delta, zp, y, row_len,
                     per_row, (float)(n_levels - 1), blocks_per_row(row_len));
  return nq_launch_status();
}

int nq_uaq_backward(const float* x, const float* gy, const float* delta, const float* zp, float* ddelta, float* dx,
                    int64_t rows, int64_t row_len, int per_row, int n_levels, nq_stream_t stream) {
  if (!x || !gy || !delta || !zp || !ddelta || rows <= 0 || row_len <= 0) return NQ_ERR_INVALID;
  // per_row=0 -> the whole tensor is one reduction row
  int64_t r = per_row ? rows : 1, len = per_row ? row_len : rows * row_len;
  hipLaunchKernelGGL(uaq_bwd_kernel, dim3((unsigned)r), dim3(TPB), 0, nq_s(stream), x, gy, delta, zp, ddelta, dx, len,
                     per_row, (float)(n_levels - 1));
  return nq_launch_status();
}

static int uaq_multi(const nq_ada_seg* segs, int nseg, bool bwd, nq_stream_t stream) {
  if (!segs || nseg <= 0) return NQ_ERR_INVALID;
  for (int base = 0; base < nseg; base += MAXSEG) {
    UaqMulti t;
    t.nseg = (nseg - base < MAXSEG) ? nseg - base : MAXSEG;
    int64_t blocks = 0;
    for (int k = 0; k < t.nseg; ++k) {
      const nq_ada_seg& h = segs[base + k];
      if (!h.x || !h.delta || !h.zp || !h.out || (bwd && !h.gy) || h.rows <= 0 || h.row_len <= 0) return NQ_ERR_INVALID;
      UaqSegD& d = t.s[k];
      d.x = h.x; d.gy = h.gy; d.delta = h.delta; d.zp = h.zp; d.out = h.out;
      d.n = h.rows * h.row_len;
      // per_row = 0: the whole tensor is one reduction row with one scale
      d.rows = h.per_row ? h.rows : 1;
      const int64_t rl = h.per_row ? h.row_len : h.rows * h.row_len;
      if (rl > 0x7fffffffLL) return NQ_ERR_INVALID;
      d.row_len = (int)rl; d.per_row = h.per_row; d.qmax = (float)(h.n_levels - 1);
      t.blk0[k] = (int)blocks;
      blocks += bwd ? d.rows : (d.n + TPB * EPT - 1) / (TPB * EPT);
      if (blocks > 0x7fffffffLL) return NQ_ERR_INVALID;
    }
    t.blk0[t.nseg] = (int)blocks;
    if (bwd)
      hipLaunchKernelGGL(uaq_bwd_multi_kernel, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), t);
    else
      hipLaunchKernelGGL(uaq_fwd_multi_kernel, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), t);
  }
  return nq_launch_status();
}

int nq_uaq_forward_multi(const nq_ada_seg* segs, int nseg, nq_stream_t stream) { return uaq_multi(segs, nseg, false, stream); }
int nq_uaq_backward_multi(const nq_ada_seg* segs, int nseg, nq_stream_t stream) { return uaq_multi(segs, nseg, true, stream); }

int nq_adaround_init(const float* x, const float* delta_in, const float* zp_in, float* delta_out, float* zp_out,
                     float* alpha, int64_t rows, int64_t row_len, int per_row, nq_stream_t stream) {
  if (!x || !delta_in || !zp_in || !delta_out || !zp_out || !alpha || bad_rows(rows, row_len)) return NQ_ERR_INVALID;
  int64_t ns = per_row ? rows : 1;
  hipLaunchKernelGGL(adaround_scale_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, nq_s(stream), delta_in,
                     zp_in, delta_out, zp_out, ns);
  hipLaunchKernelGGL(adaround_alpha_init_kernel, row_grid(rows, row_len), dim3(TPB), 0, nq_s(stream), x, delta_out,
                     alpha, row_len, per_row, blocks_per_row(row_len));
  return nq_launch_status();
}

int nq_adaround_forward(const float* x, const float* alpha, const float* delta, const float* zp, float* y, float* xq,
                        int64_t rows, int64_t row_len, int per_row, int n_levels, int soft, nq_stream_t stream) {
  if (!x || !alpha || !delta || !zp || !y || bad_rows(rows, row_len)) return NQ_ERR_INVALID;
  hipLaunchKernelGGL(adaround_fwd_kernel, row_grid(rows, row_len), dim3(TPB), 0, nq_s(stream), x, alpha, delta, zp, y,
                     xq, row_len, per_row, (float)(n_levels - 1), soft, blocks_per_row(row_len));
  return nq_launch_status();
}

int nq_adaround_backward(const float* x, const float* gy, const float* alpha, const float* delta, const float* zp,
                         float* dalpha, int64_t rows, int64_t row_len, int per_row, int n_levels, float reg_weight,
                         float reg_b, nq_stream_t stream) {
  if (!x || !gy || !alpha || !delta || !zp || !dalpha || bad_rows(rows, row_len)) return NQ_ERR_INVALID;
  hipLaunchKernelGGL(adaround_bwd_kernel, row_grid(rows, row_len), dim3(TPB), 0, nq_s(stream), x, gy, alpha, delta, zp,
                     dalpha, row_len, per_row, (float)(n_levels - 1), reg_weight, reg_b, blocks_per_row(row_len));
  return nq_launch_status();
}

int64_t nq_reduce_ws_floats(int64_t n) { return n <= 0 ? 1 : (n + RED_CHUNK - 1) / RED_CHUNK; }

int nq_round_loss(const float* alpha, int64_t n, float b, float weight, float* ws, float* out, int accumulate,
                  nq_stream_t stream) {
  if (!alpha || !ws || !out || n <= 0) return NQ_ERR_INVALID;
  int64_t parts = nq_reduce_ws_floats(n);
  hipLaunchKernelGGL(round_loss_stage1, dim3((unsigned)parts), dim3(TPB), 0, nq_s(stream), alpha, n, b, ws);
  hipLaunchKernelGGL(nq_sum_stage2, dim3(1), dim3(TPB), 0, nq_s(stream), ws, parts, weight, out, accumulate);
  return nq_launch_status();
}

int nq_round_loss_backward(const float* alpha, int64_t n, float b, float weight, const float* gscale, float* dalpha,
                           int accumulate, nq_stream_t stream) {
  if (!alpha || !dalpha || n <= 0) return NQ_ERR_INVALID;
  hipLaunchKernelGGL(round_loss_bwd_kernel, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, nq_s(stream), alpha, n,
                     b, weight, gscale, dalpha, accumulate);
  return nq_launch_status();
}

int nq_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float step_size, float beta1, float beta2,
                 float eps, float bc2_sqrt, nq_stream_t stream) {
  if (!p || !g || !m || !v || n <= 0) return NQ_ERR_INVALID;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, nq_s(stream), p, g, m, v, n,
                     step_size, beta1, beta2, eps, bc2_sqrt);
  return nq_launch_status();
}

// One block: copies row `*step` of the per-step tables into the fixed "current step" slots every other kernel of the
// iteration reads, then advances the counter -- the only thing that changes between two replays of a captured iteration.
__global__ __launch_bounds__(256) void step_prologue_kernel(const int64_t* __restrict__ order, const float* __restrict__ scal,
                                                            int* __restrict__ step, int64_t* __restrict__ cur_idx,
                                                            float* __restrict__ cur_scal, int B, int nscal) {
  const int s = *step;
  const int t = threadIdx.x;
  int64_t vi = 0;
  float vs = 0.f;
  if (t < B) vi = order[(int64_t)s * B + t];
  if (t < nscal) vs = scal[(int64_t)s * nscal + t];
  __syncthreads();   // every thread has read *step before it is advanced
  if (t < B) cur_idx[t] = vi;
  if (t < nscal) cur_scal[t] = vs;
  if (t == 0) *step = s + 1;
}

// The prologue plus the gather of the batch's decoder inputs: blocks 1.. copy out[t] = table[order[step*B + t]] (they read
// the step counter BEFORE block 0 can have advanced it?  No ordering exists between blocks, so block 0 does NOT advance it
// here: every block reads *step, and the counter is advanced by the LAST block to finish -- a ticket in step[1]).
__global__ __launch_bounds__(256) void step_prologue_gather_kernel(const int64_t* __restrict__ order, const float* __restrict__ scal,
                                                                   int* __restrict__ step, int64_t* __restrict__ cur_idx,
                                                                   float* __restrict__ cur_scal, int B, int nscal,
                                                                   const float* __restrict__ table, int64_t table_rows,
                                                                   int64_t row_len, float* __restrict__ out) {
  __shared__ int last;
  const int s = __hip_atomic_load(step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int t = threadIdx.x;
  if (blockIdx.x == 0) {
    if (t < B) cur_idx[t] = order[(int64_t)s * B + t];
    if (t < nscal) cur_scal[t] = scal[(int64_t)s * nscal + t];
  } else {
    const int64_t per = ((int64_t)B * row_len + (gridDim.x - 2)) / (gridDim.x - 1);   // elements per gather block
    const int64_t lo = (int64_t)(blockIdx.x - 1) * per, hi = min(lo + per, (int64_t)B * row_len);
    for (int64_t e = lo + t; e < hi; e += 256) {
      const int64_t f = e / row_len, j = e - f * row_len;
      int64_t idx = order[(int64_t)s * B + f];
      idx = idx < 0 ? 0 : (idx >= table_rows ? table_rows - 1 : idx);
      out[e] = table[idx * row_len + j];
    }
  }
  // the last block to arrive advances the step counter and re-arms the ticket (nobody reads *step after its ticket)
  __syncthreads();
  if (t == 0) last = (atomicAdd(step + 1, 1) == (int)gridDim.x - 1);
  __syncthreads();
  if (last && t == 0) {
    step[1] = 0;
    __hip_atomic_store(step, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

static int ada_multi(const nq_ada_seg* segs, int nseg, float reg_b, const float* dyn, bool bwd, nq_stream_t stream) {
  if (!segs || nseg <= 0) return NQ_ERR_INVALID;
  for (int base = 0; base < nseg; base += MAXSEG) {
    AdaMulti t;
    t.nseg = (nseg - base < MAXSEG) ? nseg - base : MAXSEG;
    t.reg_b = reg_b;
    t.dyn = dyn;
    int blocks = 0;
    for (int k = 0; k < t.nseg; ++k) {
      const nq_ada_seg& h = segs[base + k];
      if (!h.x || !h.alpha || !h.delta || !h.zp || !h.out || (bwd && !h.gy) || h.rows <= 0 || h.row_len <= 0 ||
          h.row_len > 0x7fffffffLL)
        return NQ_ERR_INVALID;
      AdaSegD& d = t.s[k];
      d.x = h.x; d.gy = h.gy; d.alpha = h.alpha; d.delta = h.delta; d.zp = h.zp; d.out = h.out;
      d.n = h.rows * h.row_len; d.row_len = (int)h.row_len; d.per_row = h.per_row; d.soft = h.soft;
      d.qmax = (float)(h.n_levels - 1); d.reg_weight = h.reg_weight;
      d.vec = (((uintptr_t)h.x | (uintptr_t)h.alpha | (uintptr_t)h.out | (uintptr_t)(bwd ? h.gy : h.x)) & 15) == 0;
      t.blk0[k] = blocks;
      blocks += (int)((d.n + TPB * EPT - 1) / (TPB * EPT));
    }
    t.blk0[t.nseg] = blocks;
    if (bwd)
      hipLaunchKernelGGL(adaround_multi_kernel<true>, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), t);
    else
      hipLaunchKernelGGL(adaround_multi_kernel<false>, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), t);
  }
  return nq_launch_status();
}

int nq_adaround_forward_multi(const nq_ada_seg* segs, int nseg, nq_stream_t stream) {
  return ada_multi(segs, nseg, 0.f, nullptr, false, stream);
}

int nq_adaround_backward_multi(const nq_ada_seg* segs, int nseg, float reg_b, nq_stream_t stream) {
  return ada_multi(segs, nseg, reg_b, nullptr, true, stream);
}

int nq_adaround_backward_multi_dyn(const nq_ada_seg* segs, int nseg, const float* dyn, nq_stream_t stream) {
  if (!dyn) return NQ_ERR_INVALID;
  return ada_multi(segs, nseg, 0.f, dyn, true, stream);
}

int nq_step_prologue(const int64_t* order, const float* scal, int* step, int64_t* cur_idx, float* cur_scal, int B, int nscal,
                     nq_stream_t stream) {
  if (!order || !scal || !step || !cur_idx || !cur_scal || B <= 0 || B > 256 || nscal <= 0 || nscal > 256) return NQ_ERR_INVALID;
  hipLaunchKernelGGL(step_prologue_kernel, dim3(1), dim3(256), 0, nq_s(stream), order, scal, step, cur_idx, cur_scal, B, nscal);
  return nq_launch_status();
}

int nq_step_prologue_gather(const int64_t* order, const float* scal, int* step, int64_t* cur_idx, float* cur_scal, int B, int nscal,
                            const float* table, int64_t table_rows, int64_t row_len, float* out, nq_stream_t stream) {
  if (!order || !scal || !step || !cur_idx || !cur_scal || B <= 0 || B > 256 || nscal <= 0 || nscal > 256 || !table || !out ||
      table_rows <= 0 || row_len <= 0)
    return NQ_ERR_INVALID;
  // step = {counter, ticket}: two ints (the caller zeroes both)
  const int64_t total = (int64_t)B * row_len;
  int gb = (int)((total + 4095) / 4096);
  if (gb < 1) gb = 1;
  if (gb > 64) gb = 64;
  hipLaunchKernelGGL(step_prologue_gather_kernel, dim3(1 + gb), dim3(256), 0, nq_s(stream), order, scal, step, cur_idx, cur_scal, B,
                     nscal, table, table_rows, row_len, out);
  return nq_launch_status();
}

static int adam_multi(const nq_adam_seg* segs, int nseg, float step_size, float beta1, float beta2, float eps, float bc2_sqrt,
                      const float* dyn, nq_stream_t stream);

int nq_adaround_adam_multi(const nq_ada_adam_seg* segs, int nseg, float reg_b, float step_size, float beta1, float beta2, float eps,
                           float bc2_sqrt, const float* dyn, nq_stream_t stream) {
  if (!segs || nseg <= 0) return NQ_ERR_INVALID;
  for (int base = 0; base < nseg; base += MAXSEG) {
    AdaAdamMulti t;
    t.nseg = (nseg - base < MAXSEG) ? nseg - base : MAXSEG;
    int blocks = 0;
    for (int k = 0; k < t.nseg; ++k) {
      const nq_ada_adam_seg& h = segs[base + k];
      if (!h.x || !h.gy || !h.alpha || !h.delta || !h.zp || !h.m || !h.v || h.rows <= 0 || h.row_len <= 0 ||
          h.row_len > 0x7fffffffLL)
        return NQ_ERR_INVALID;
      AdaAdamSegD& d = t.s[k];
      d.x = h.x; d.gy = h.gy; d.alpha = h.alpha; d.delta = h.delta; d.zp = h.zp; d.m = h.m; d.v = h.v;
      d.n = h.rows * h.row_len; d.row_len = (int)h.row_len; d.per_row = h.per_row;
      d.qmax = (float)(h.n_levels - 1); d.reg_weight = h.reg_weight;
      d.vec = (((uintptr_t)h.x | (uintptr_t)h.gy | (uintptr_t)h.alpha | (uintptr_t)h.m | (uintptr_t)h.v) & 15) == 0;
      t.blk0[k] = blocks;
      blocks += (int)((d.n + TPB * EPT - 1) / (TPB * EPT));
    }
    t.blk0[t.nseg] = blocks;
    hipLaunchKernelGGL(adaround_adam_multi_kernel, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), t, reg_b, step_size,
                       beta1, beta2, eps, bc2_sqrt, dyn);
  }
  return nq_launch_status();
}

int nq_adam_step_multi(const nq_adam_seg* segs, int nseg, float step_size, float beta1, float beta2, float eps,
                       float bc2_sqrt, nq_stream_t stream) {
  return adam_multi(segs, nseg, step_size, beta1, beta2, eps, bc2_sqrt, nullptr, stream);
}

int nq_adam_step_multi_dyn(const nq_adam_seg* segs, int nseg, const float* dyn, float beta1, float beta2, float eps,
                           nq_stream_t stream) {
  if (!dyn) return NQ_ERR_INVALID;
  return adam_multi(segs, nseg, 0.f, beta1, beta2, eps, 1.f, dyn, stream);
}

static int adam_multi(const nq_adam_seg* segs, int nseg, float step_size, float beta1, float beta2, float eps, float bc2_sqrt,
                      const float* dyn, nq_stream_t stream) {
  if (!segs || nseg <= 0) return NQ_ERR_INVALID;
  for (int base = 0; base < nseg; base += MAXSEG) {
    AdamMulti t;
    t.nseg = (nseg - base < MAXSEG) ? nseg - base : MAXSEG;
    int blocks = 0;
    for (int k = 0; k < t.nseg; ++k) {
      const nq_adam_seg& h = segs[base + k];
      if (!h.p || !h.g || !h.m || !h.v || h.n <= 0) return NQ_ERR_INVALID;
      t.s[k].p = h.p; t.s[k].g = h.g; t.s[k].m = h.m; t.s[k].v = h.v; t.s[k].n = h.n;
      t.s[k].vec = (((uintptr_t)h.p | (uintptr_t)h.g | (uintptr_t)h.m | (uintptr_t)h.v) & 15) == 0;
      t.blk0[k] = blocks;
      blocks += (int)((h.n + TPB * EPT - 1) / (TPB * EPT));
    }
    t.blk0[t.nseg] = blocks;
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), t, step_size, beta1, beta2,
                       eps, bc2_sqrt, dyn);
  }
  return nq_launch_status();
}

}  // extern "C"
