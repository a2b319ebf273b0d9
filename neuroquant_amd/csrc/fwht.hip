// Orthonormal Sylvester-ordered Walsh-Hadamard transform along the C_in axis of a conv weight
// (hadamard_along_channel_weight, quantization/quant_layer.py:16-22), with the zero padding to 2^k
// (:45-49) folded into the load and the [:, :C] slice (:71) folded into the store.
//
// Layout: x [outer][n_in][inner], y [outer][n_out][inner]; a "column" is one (outer, inner) pair and is
// transformed along its n entries.  A workgroup owns OPB whole `outer` rows = OPB*inner columns, i.e. ONE contiguous
// piece of x (OPB*n_in*inner floats) and of y: global loads / stores run linearly over it (fully coalesced; the
// column-major gather this replaces touched ~8 separate 36-byte pieces per wave instruction).  In LDS the tile is
// [n][TC+1] (lanes run along columns -> conflict-free butterflies); log2(n) in-place stages (a,b)->(a+b,a-b) in the
// same order as the oracle, scaled by 1/sqrt(n) on the way out.  HBM-bound (one read + one write of the weight).
#include "nq_common.h"
#include "quant_elem.h"

namespace {

// Workgroup size / tile / columns per workgroup (compile-time; A/B of whole libraries inside a NeRV-3M + Hadamard iteration,
// us forward / backward launch): 256 threads, 8192-float tile, ~32 columns (rounds 2-3) 23.0 / 24.1; 512, 8192, 32: 17.5 / 20.1;
// 1024, 8192, 32: 19.6 / 22.0; 512, 8192, 16: 16.3 / 17.9; 256 or 512, 4096, 16: 18.6 / 19.4-20.2; 512, 2560, 9: 19.9 / 21.4.
// The phases of a tile (load | three butterfly passes | store) are separated by barriers, so a launch lasts as long as the
// longest workgroup's chain: twice the waves halve the chain's per-thread work, smaller tiles did not help further.
#ifndef NQ_FWHT_TPB
#define NQ_FWHT_TPB 512
#endif
constexpr int TPB = NQ_FWHT_TPB;
constexpr int FW_MAXSEG_FQ = 16;
#ifndef NQ_FWHT_TILE
#define NQ_FWHT_TILE 8192
#endif
#ifndef NQ_FWHT_COLS
#define NQ_FWHT_COLS 16
#endif
#ifndef NQ_FWHT_U
#define NQ_FWHT_U 3
#endif
constexpr int FQ_U = NQ_FWHT_U;                // elements per thread and round of the fused launches' element loops
constexpr int TILE = NQ_FWHT_TILE;             // floats of a workgroup's tile (n * columns)
constexpr int LDS_FLOATS = TILE + 1024;    // n * (columns + 1) floats, n <= 1024

// G butterfly stages s0 .. s0+G-1 in one pass: a thread takes the 2^G rows that differ in bits s0 .. s0+G-1 of one column
template <int G>
__device__ __forceinline__ void fwht_pass(float* lds, int n, int s0, int TC, int LD) {
  constexpr int R = 1 << G;
  const int groups = (n >> G) * TC, h0 = 1 << s0;
  for (int e = threadIdx.x; e < groups; e += TPB) {
    const int q = e / TC, t = e - q * TC;
    const int base = ((q >> s0) << (s0 + G)) | (q & (h0 - 1));   // row with bits s0 .. s0+G-1 clear
    float v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = lds[(base + j * h0) * LD + t];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        if ((j & (1 << g)) == 0) {
          const float a = v[j], b = v[j | (1 << g)];
          v[j] = a + b;
          v[j | (1 << g)] = a - b;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) lds[(base + j * h0) * LD + t] = v[j];
  }
  __syncthreads();
}
__device__ __forceinline__ void fwht_stages(float* lds, int n, int log2n, int TC, int LD) {
  int s = 0;
  for (; s + 3 <= log2n; s += 3) fwht_pass<3>(lds, n, s, TC, LD);
  if (log2n - s == 2) fwht_pass<2>(lds, n, s, TC, LD);
  else if (log2n - s == 1) fwht_pass<1>(lds, n, s, TC, LD);
}

__device__ __forceinline__ void fwht_block(float* lds, int blk, const float* __restrict__ x, float* __restrict__ y,
                                           int64_t outer, int n, int log2n, int inner, int n_in, int n_out, int OPB,
                                           float sqrt_n) {
  const int64_t o0 = (int64_t)blk * OPB;
  const int nob = (int)min((int64_t)OPB, outer - o0);   // outer rows of this block
  const int TC = OPB * inner, LD = TC + 1;
  // rows [n_in, n): the zero padding
  for (int e = threadIdx.x; e < (n - n_in) * TC; e += TPB) {
    const int c = n_in + e / TC, t = e - (e / TC) * TC;
    lds[c * LD + t] = 0.f;
  }
  // rows [0, n_in): linear sweep over the block's contiguous input.  The (outer row, channel, inner) coordinates of element
  // e = tid + 256 j are STEPPED from one iteration to the next (two compares instead of two integer divisions per element:
  // the divisions were most of this kernel's instructions -- 23.5 us for 43 MB on NeRV-3M's layers)
  const int d_c = TPB / inner, d_i = TPB - d_c * inner;   // 256 = d_c * inner + d_i
  {
    const int per_o = n_in * inner, total = nob * per_o;
    const float* __restrict__ xb = x + o0 * per_o;
    int ol = threadIdx.x / per_o, r = threadIdx.x - ol * per_o;
    int c = r / inner, ii = r - c * inner;
    for (int e = threadIdx.x; e < OPB * per_o; e += TPB) {
      lds[c * LD + ol * inner + ii] = (e < total) ? xb[e] : 0.f;
      ii += d_i;
      c += d_c;
      if (ii >= inner) {
        ii -= inner;
        ++c;
      }
      while (c >= n_in) {
        c -= n_in;
        ++ol;
      }
    }
  }
  __syncthreads();
  // butterflies: up to three stages per pass over LDS (an element's 8 / 4 / 2 partners in registers), each stage the same
  // (a, b) -> (a + b, a - b) on the same operands as the one-stage-per-pass loop (bit-identical), with a third of the
  // LDS round trips and barriers
  fwht_stages(lds, n, log2n, TC, LD);
  // store the first n_out entries: linear sweep over the block's contiguous output (coordinates stepped as above).  The
  // scale is the oracle's division by sqrt(n); for n = 4^k it is an exact power of two and a multiplication gives the same bits
  {
    const int per_o = n_out * inner, total = nob * per_o;
    float* __restrict__ yb = y + o0 * per_o;
    const bool pow2 = (log2n & 1) == 0;
    const float inv = 1.0f / sqrt_n;
    int ol = threadIdx.x / per_o, r = threadIdx.x - ol * per_o;
    int c = r / inner, ii = r - c * inner;
    for (int e = threadIdx.x; e < total; e += TPB) {
      const float v = lds[c * LD + ol * inner + ii];
      yb[e] = pow2 ? v * inv : v / sqrt_n;
      ii += d_i;
      c += d_c;
      if (ii >= inner) {
        ii -= inner;
        ++c;
      }
      while (c >= n_out) {
        c -= n_out;
        ++ol;
      }
    }
  }
}

// ---- round 4: the AdaRound fake-quant / its backward + Adam fused with the transform (NeRV-3M + Hadamard spends a sixth of an
//      iteration on the parameter side: fake-quant 15 us -> FWHT 17 us forward, FWHT 18 us -> d(alpha)+Adam 33 us backward; the
//      transform-domain tensors are 1.75x the weights, and each of them crossed HBM twice between the two launches) ----
// One tensor of a fused launch.  n > 0: a transformed weight, tiles of OPB outer rows as fwht_block; n == 0: a plain tensor
// (the biases: no transform), blocks of TPB * 4 elements.
struct FqSeg {
  const float* x;       // transform-domain FP weight (outer, n, inner) / bias
  float* alpha;         // rounding variables, same shape (const in the forward launch)
  const float* delta;   // per outer row (per_row) or one scalar
  const float* zp;
  float* m;             // Adam moments (backward launch)
  float* v;
  const float* gy;      // backward: d(loss)/d(W^) (outer, c_in, inner) / d(loss)/d(b^)
  float* y;             // forward: W^ (outer, c_in, inner) / b^
  int64_t outer;
  int n, log2n, inner, c_in, OPB, per_row, soft;
  float sqrt_n, qmax, reg_weight;
};
struct FqMulti {
  FqSeg s[FW_MAXSEG_FQ];
  int blk0[FW_MAXSEG_FQ + 1];
  int nseg;
};

// forward: y = H(Q(x))[:, :c_in]  (quant_layer.py:70-71 with AdaRoundQuantizer.forward, quantizer.py:288-300)
__device__ __forceinline__ void fq_fwht_tile(float* lds, int blk, const FqSeg& g) {
  const int n = g.n, inner = g.inner, OPB = g.OPB;
  const int64_t o0 = (int64_t)blk * OPB;
  const int nob = (int)min((int64_t)OPB, g.outer - o0);
  const int TC = OPB * inner, LD = TC + 1;
  const int d_c = TPB / inner, d_i = TPB - d_c * inner;
  {
    const int per_o = n * inner, total = nob * per_o;
    const float* __restrict__ xb = g.x + o0 * per_o;
    const float* __restrict__ ab = g.alpha + o0 * per_o;
    int ol = threadIdx.x / per_o, r = threadIdx.x - ol * per_o;
    int c = r / inner, ii = r - c * inner;
    for (int e = threadIdx.x; e < OPB * per_o; e += TPB) {
      float q = 0.f;
      if (e < total) {
        const int64_t row = g.per_row ? o0 + ol : 0;
        float xq;
        q = ada_fwd_elem(xb[e], ab[e], g.delta[row], g.zp[row], g.qmax, g.soft, xq);
      }
      lds[c * LD + ol * inner + ii] = q;
      ii += d_i;
      c += d_c;
      if (ii >= inner) {
        ii -= inner;
        ++c;
      }
      while (c >= n) {
        c -= n;
        ++ol;
      }
    }
  }
  __syncthreads();
  fwht_stages(lds, n, g.log2n, TC, LD);
  {
    const int n_out = g.c_in;
    const int per_o = n_out * inner, total = nob * per_o;
    float* __restrict__ yb = g.y + o0 * per_o;
    const bool pow2 = (g.log2n & 1) == 0;
    const float inv = 1.0f / g.sqrt_n;
    int ol = threadIdx.x / per_o, r = threadIdx.x - ol * per_o;
    int c = r / inner, ii = r - c * inner;
    for (int e = threadIdx.x; e < total; e += TPB) {
      const float v = lds[c * LD + ol * inner + ii];
      yb[e] = pow2 ? v * inv : v / g.sqrt_n;
      ii += d_i;
      c += d_c;
      if (ii >= inner) {
        ii -= inner;
        ++c;
      }
      while (c >= n_out) {
        c -= n_out;
        ++ol;
      }
    }
  }
}

__global__ __launch_bounds__(TPB) void adaround_fwht_multi_kernel(FqMulti t) {
  __shared__ float lds[LDS_FLOATS];
  int k = 0;
  while (k + 1 < t.nseg && (int)blockIdx.x >= t.blk0[k + 1]) ++k;
  const FqSeg& g = t.s[k];
  const int blk = (int)blockIdx.x - t.blk0[k];
  if (g.n > 0) {
    fq_fwht_tile(lds, blk, g);
    return;
  }
  const int64_t total = g.outer;   // plain tensor: outer = element count
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = (int64_t)blk * (TPB * 4) + u * TPB + threadIdx.x;
    if (i < total) {
      float xq;
      g.y[i] = ada_fwd_elem(g.x[i], g.alpha[i], g.delta[0], g.zp[0], g.qmax, g.soft, xq);
    }
  }
}

// backward: g_T = H(pad(gy)) (the transform is its own transpose), then d(alpha) = ada_bwd_elem(x, g_T, alpha, ...) (+ the
// regulariser's gradient) and Adam's update of alpha -- the arithmetic of adaround_adam_multi_kernel (quant.hip) element by
// element, so alpha / m / v come out bit-identical to nq_fwht_multi + nq_adaround_adam_multi.
__device__ __forceinline__ void fwht_ada_adam_tile(float* lds, int blk, const FqSeg& g, float reg_b, float rw, float step_size,
                                                   float beta1, float beta2, float eps, float bc2_sqrt) {
  const int n = g.n, inner = g.inner, OPB = g.OPB, n_in = g.c_in;
  const int64_t o0 = (int64_t)blk * OPB;
  const int nob = (int)min((int64_t)OPB, g.outer - o0);
  const int TC = OPB * inner, LD = TC + 1;
  for (int e = threadIdx.x; e < (n - n_in) * TC; e += TPB) {
    const int c = n_in + e / TC, tt = e - (e / TC) * TC;
    lds[c * LD + tt] = 0.f;
  }
  const int d_c = TPB / inner, d_i = TPB - d_c * inner;
  {
    const int per_o = n_in * inner, total = nob * per_o;
    const float* __restrict__ xb = g.gy + o0 * per_o;
    int ol = threadIdx.x / per_o, r = threadIdx.x - ol * per_o;
    int c = r / inner, ii = r - c * inner;
    for (int e = threadIdx.x; e < OPB * per_o; e += TPB) {
      lds[c * LD + ol * inner + ii] = (e < total) ? xb[e] : 0.f;
      ii += d_i;
      c += d_c;
      if (ii >= inner) {
        ii -= inner;
        ++c;
      }
      while (c >= n_in) {
        c -= n_in;
        ++ol;
      }
    }
  }
  __syncthreads();
  fwht_stages(lds, n, g.log2n, TC, LD);
  {
    // d(alpha) + Adam over the tile, FQ_U elements per thread and round: ALL of a round's global loads are issued before its
    // first store (the stores to m / v / alpha may alias the next loads as far as the compiler knows, so one element per loop
    // iteration was one exposed memory round trip per element -- 16 per workgroup and launch).  Elements past the tile load
    // the tile's last element (unconditional loads, no branches) and store nothing.  Same arithmetic per element.
    const int per_o = n * inner, total = nob * per_o;
    const int64_t base = o0 * per_o;
    const bool pow2 = (g.log2n & 1) == 0;
    const float inv = 1.0f / g.sqrt_n;
    int ol = threadIdx.x / per_o, r = threadIdx.x - ol * per_o;
    int c = r / inner, ii = r - c * inner;
    for (int e0 = threadIdx.x; e0 < total; e0 += FQ_U * TPB) {
      float xv[FQ_U], av[FQ_U], mv[FQ_U], vv[FQ_U], dl[FQ_U], zq[FQ_U], tv[FQ_U];
#pragma unroll
      for (int u = 0; u < FQ_U; ++u) {
        const int e = e0 + u * TPB;
        const bool ok = e < total;
        const int64_t i = base + (ok ? e : total - 1);
        const int64_t row = g.per_row ? o0 + (ok ? ol : nob - 1) : 0;
        xv[u] = g.x[i];
        av[u] = g.alpha[i];
        mv[u] = g.m[i];
        vv[u] = g.v[i];
        dl[u] = g.delta[row];
        zq[u] = g.zp[row];
        tv[u] = lds[ok ? c * LD + ol * inner + ii : 0];
        ii += d_i;
        c += d_c;
        if (ii >= inner) {
          ii -= inner;
          ++c;
        }
        while (c >= n) {
          c -= n;
          ++ol;
        }
      }
#pragma unroll
      for (int u = 0; u < FQ_U; ++u) {
        const int e = e0 + u * TPB;
        if (e < total) {
          const int64_t i = base + e;
          const float gv = pow2 ? tv[u] * inv : tv[u] / g.sqrt_n;          // what nq_fwht stores
          const float gr = ada_bwd_elem(xv[u], gv, av[u], dl[u], zq[u], g.qmax, rw, reg_b);
          float mi = mv[u], vi = vv[u];
          mi = mi + (1.f - beta1) * (gr - mi);
          vi = vi * beta2 + (1.f - beta2) * (gr * gr);
          const float denom = sqrtf(vi) / bc2_sqrt + eps;
          g.m[i] = mi;
          g.v[i] = vi;
          g.alpha[i] = av[u] - step_size * (mi / denom);
        }
      }
    }
  }
}

__global__ __launch_bounds__(TPB) void fwht_adaround_adam_multi_kernel(FqMulti t, float reg_b, float step_size, float beta1,
                                                                       float beta2, float eps, float bc2_sqrt,
                                                                       const float* __restrict__ dyn) {
  __shared__ float lds[LDS_FLOATS];
  float gate = 1.f;
  if (dyn) {   // per-step scalars from device memory (graph replays): {reg_b, regulariser gate, lr/(1-beta1^t), sqrt(1-beta2^t)}
    reg_b = dyn[0];
    gate = dyn[1];
    step_size = dyn[2];
    bc2_sqrt = dyn[3];
  }
  int k = 0;
  while (k + 1 < t.nseg && (int)blockIdx.x >= t.blk0[k + 1]) ++k;
  const FqSeg& g = t.s[k];
  const int blk = (int)blockIdx.x - t.blk0[k];
  const float rw = dyn ? g.reg_weight * gate : g.reg_weight;   // gate is exactly 0 or 1
  if (g.n > 0) {
    fwht_ada_adam_tile(lds, blk, g, reg_b, rw, step_size, beta1, beta2, eps, bc2_sqrt);
    return;
  }
  const int64_t total = g.outer;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = (int64_t)blk * (TPB * 4) + u * TPB + threadIdx.x;
    if (i < total) {
      const float gr = ada_bwd_elem(g.x[i], g.gy[i], g.alpha[i], g.delta[0], g.zp[0], g.qmax, rw, reg_b);
      float mi = g.m[i], vi = g.v[i];
      mi = mi + (1.f - beta1) * (gr - mi);
      vi = vi * beta2 + (1.f - beta2) * (gr * gr);
      const float denom = sqrtf(vi) / bc2_sqrt + eps;
      g.m[i] = mi;
      g.v[i] = vi;
      g.alpha[i] = g.alpha[i] - step_size * (mi / denom);
    }
  }
}

__global__ __launch_bounds__(TPB) void fwht_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t outer,
                                                   int n, int log2n, int inner, int n_in, int n_out, int OPB,
                                                   float sqrt_n) {
  __shared__ float lds[LDS_FLOATS];
  fwht_block(lds, (int)blockIdx.x, x, y, outer, n, log2n, inner, n_in, n_out, OPB, sqrt_n);
}

// several tensors in ONE launch (all layers of a decoder: the per-layer launches are 10 us each for ~1 us of traffic)
constexpr int FW_MAXSEG = 16;
struct FwSeg {
  const float* x;
  float* y;
  int64_t outer;
  int n, log2n, inner, n_in, n_out, OPB;
  float sqrt_n;
};
struct FwMulti {
  FwSeg s[FW_MAXSEG];
  int blk0[FW_MAXSEG + 1];
  int nseg;
};
__global__ __launch_bounds__(TPB) void fwht_multi_kernel(FwMulti t) {
  __shared__ float lds[LDS_FLOATS];
  int k = 0;
  while (k + 1 < t.nseg && (int)blockIdx.x >= t.blk0[k + 1]) ++k;
  const FwSeg& g = t.s[k];
  fwht_block(lds, (int)blockIdx.x - t.blk0[k], g.x, g.y, g.outer, g.n, g.log2n, g.inner, g.n_in, g.n_out, g.OPB, g.sqrt_n);
}

// Fallback for rows too long for the tile above (n*inner > 8192): TC arbitrary columns per workgroup, gathered.
__global__ __launch_bounds__(TPB) void fwht_cols_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t ncols,
                                                   int n, int log2n, int64_t inner, int n_in, int n_out, int TC,
                                                   float sqrt_n) {
  __shared__ float lds[LDS_FLOATS];
  const int LD = TC + 1;
  const int64_t col0 = (int64_t)blockIdx.x * TC;
  // load (zero beyond n_in or beyond the last column)
  for (int e = threadIdx.x; e < n * TC; e += TPB) {
    int c = e / TC, t = e - c * TC;
    int64_t col = col0 + t;
    float v = 0.f;
    if (col < ncols && c < n_in) {
      int64_t o = col / inner, ii = col - o * inner;
      v = x[(o * n_in + c) * inner + ii];
    }
    lds[c * LD + t] = v;
  }
  __syncthreads();
  fwht_stages(lds, n, log2n, TC, LD);
  // store first n_out entries
  for (int e = threadIdx.x; e < n_out * TC; e += TPB) {
    int c = e / TC, t = e - c * TC;
    int64_t col = col0 + t;
    if (col < ncols) {
      int64_t o = col / inner, ii = col - o * inner;
      y[(o * n_out + c) * inner + ii] = lds[c * LD + t] / sqrt_n;
    }
  }
}

}  // namespace

extern "C" int nq_fwht(const float* x, float* y, int64_t outer, int n, int64_t inner, int n_in, int n_out,
                       nq_stream_t stream) {
  if (!x || !y || x == y || outer <= 0 || inner <= 0 || n <= 0 || (n & (n - 1)) != 0) return NQ_ERR_INVALID;
  if (n_in <= 0 || n_in > n || n_out <= 0 || n_out > n) return NQ_ERR_INVALID;
  if (n > 1024) return NQ_ERR_UNSUPPORTED;
  int log2n = 0;
  while ((1 << log2n) < n) ++log2n;
  if ((int64_t)n * inner > TILE) {   // one outer row does not fit the LDS tile: column-gather variant
    int TC = TILE / n;
    if (TC > 32) TC = 32;
    int64_t ncols = outer * inner;
    int64_t blocks = (ncols + TC - 1) / TC;
    if (blocks > 0x7fffffffLL) return NQ_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fwht_cols_kernel, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), x, y, ncols, n, log2n, inner,
                       n_in, n_out, TC, sqrtf((float)n));
    return nq_launch_status();
  }
  int OPB = (int)(TILE / ((int64_t)n * inner));   // outer rows per workgroup
  const int cap = (int)((NQ_FWHT_COLS + inner - 1) / inner); // ~32 columns per workgroup keeps the grid large
  if (OPB > cap) OPB = cap;
  if (OPB < 1) OPB = 1;
  int64_t blocks = (outer + OPB - 1) / OPB;
  if (blocks > 0x7fffffffLL) return NQ_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(fwht_kernel, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), x, y, outer, n, log2n, (int)inner, n_in,
                     n_out, OPB, sqrtf((float)n));
  return nq_launch_status();
}

extern "C" int nq_fwht_multi(const nq_fwht_seg* segs, int nseg, nq_stream_t stream) {
  if (!segs || nseg <= 0) return NQ_ERR_INVALID;
  FwMulti t;
  t.nseg = 0;
  int blocks = 0;
  auto flush = [&]() {
    if (t.nseg == 0) return;
    t.blk0[t.nseg] = blocks;
    hipLaunchKernelGGL(fwht_multi_kernel, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), t);
    t.nseg = 0;
    blocks = 0;
  };
  for (int i = 0; i < nseg; ++i) {
    const nq_fwht_seg& h = segs[i];
    if (!h.x || !h.y || h.x == h.y || h.outer <= 0 || h.inner <= 0 || h.n <= 0 || (h.n & (h.n - 1)) != 0) return NQ_ERR_INVALID;
    if (h.n_in <= 0 || h.n_in > h.n || h.n_out <= 0 || h.n_out > h.n) return NQ_ERR_INVALID;
    if (h.n > 1024) return NQ_ERR_UNSUPPORTED;
    if ((int64_t)h.n * h.inner > TILE) {   // long rows: the single-tensor column-gather variant
      int rc = nq_fwht(h.x, h.y, h.outer, h.n, h.inner, h.n_in, h.n_out, stream);
      if (rc != NQ_OK) return rc;
      continue;
    }
    int log2n = 0;
    while ((1 << log2n) < h.n) ++log2n;
    int OPB = (int)(TILE / ((int64_t)h.n * h.inner));
    const int cap = (int)((NQ_FWHT_COLS + h.inner - 1) / h.inner);
    if (OPB > cap) OPB = cap;
    if (OPB < 1) OPB = 1;
    const int64_t nb = (h.outer + OPB - 1) / OPB;
    if (nb + blocks > 0x7fffffffLL) return NQ_ERR_UNSUPPORTED;
    if (t.nseg == FW_MAXSEG) flush();
    t.s[t.nseg] = FwSeg{h.x, h.y, h.outer, h.n, log2n, (int)h.inner, h.n_in, h.n_out, OPB, sqrtf((float)h.n)};
    t.blk0[t.nseg] = blocks;
    blocks += (int)nb;
    ++t.nseg;
  }
  flush();
  return nq_launch_status();
}

// One tensor of the fused launches (include/nq_hip.h: nq_fq_fwht_seg) -> device segment; 0 on success
static int fq_fill(FqSeg& d, const nq_fq_fwht_seg& h, bool bwd, int* blocks) {
  if (!h.x || !h.alpha || !h.delta || !h.zp || h.outer <= 0 || h.inner <= 0 || h.n_levels < 2) return NQ_ERR_INVALID;
  if (bwd ? (!h.gy || !h.m || !h.v) : !h.y) return NQ_ERR_INVALID;
  d.x = h.x; d.alpha = h.alpha; d.delta = h.delta; d.zp = h.zp; d.m = h.m; d.v = h.v; d.gy = h.gy; d.y = h.y;
  d.per_row = h.per_row; d.soft = h.soft; d.qmax = (float)(h.n_levels - 1); d.reg_weight = h.reg_weight;
  if (h.n == 0) {   // plain tensor (a bias): no transform
    d.n = 0; d.log2n = 0; d.inner = 1; d.c_in = 0; d.OPB = 0; d.sqrt_n = 1.f;
    d.outer = h.outer * h.inner;
    if (d.per_row) return NQ_ERR_UNSUPPORTED;
    *blocks = (int)((d.outer + TPB * 4 - 1) / (TPB * 4));
    return NQ_OK;
  }
  if (h.n < 0 || (h.n & (h.n - 1)) != 0 || h.c_in <= 0 || h.c_in > h.n) return NQ_ERR_INVALID;
  if (h.n > 1024 || (int64_t)h.n * h.inner > TILE) return NQ_ERR_UNSUPPORTED;   // (longer rows: the unfused launches)
  int log2n = 0;
  while ((1 << log2n) < h.n) ++log2n;
  int OPB = (int)(TILE / ((int64_t)h.n * h.inner));
  const int cap = (int)((NQ_FWHT_COLS + h.inner - 1) / h.inner);
  if (OPB > cap) OPB = cap;
  if (OPB < 1) OPB = 1;
  d.outer = h.outer; d.n = h.n; d.log2n = log2n; d.inner = (int)h.inner; d.c_in = h.c_in; d.OPB = OPB; d.sqrt_n = sqrtf((float)h.n);
  const int64_t nb = (h.outer + OPB - 1) / OPB;
  if (nb > 0x3fffffffLL) return NQ_ERR_UNSUPPORTED;
  *blocks = (int)nb;
  return NQ_OK;
}

extern "C" int nq_adaround_fwht_multi(const nq_fq_fwht_seg* segs, int nseg, nq_stream_t stream) {
  if (!segs || nseg <= 0) return NQ_ERR_INVALID;
  for (int base = 0; base < nseg; base += FW_MAXSEG_FQ) {
    FqMulti t;
    t.nseg = (nseg - base < FW_MAXSEG_FQ) ? nseg - base : FW_MAXSEG_FQ;
    int blocks = 0;
    for (int i = 0; i < t.nseg; ++i) {
      int nb = 0;
      if (int rc = fq_fill(t.s[i], segs[base + i], false, &nb)) return rc;
      t.blk0[i] = blocks;
      blocks += nb;
    }
    t.blk0[t.nseg] = blocks;
    hipLaunchKernelGGL(adaround_fwht_multi_kernel, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), t);
  }
  return nq_launch_status();
}

extern "C" int nq_fwht_adaround_adam_multi(const nq_fq_fwht_seg* segs, int nseg, float reg_b, float step_size, float beta1,
                                           float beta2, float eps, float bc2_sqrt, const float* dyn, nq_stream_t stream) {
  if (!segs || nseg <= 0) return NQ_ERR_INVALID;
  for (int base = 0; base < nseg; base += FW_MAXSEG_FQ) {
    FqMulti t;
    t.nseg = (nseg - base < FW_MAXSEG_FQ) ? nseg - base : FW_MAXSEG_FQ;
    int blocks = 0;
    for (int i = 0; i < t.nseg; ++i) {
      int nb = 0;
      if (int rc = fq_fill(t.s[i], segs[base + i], true, &nb)) return rc;
      t.blk0[i] = blocks;
      blocks += nb;
    }
    t.blk0[t.nseg] = blocks;
    hipLaunchKernelGGL(fwht_adaround_adam_multi_kernel, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), t, reg_b, step_size,
                       beta1, beta2, eps, bc2_sqrt, dyn);
  }
  return nq_launch_status();
}
