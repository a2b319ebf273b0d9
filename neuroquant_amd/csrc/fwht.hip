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

namespace {

constexpr int TPB = 256;
constexpr int LDS_FLOATS = 8192 + 1024;

__global__ __launch_bounds__(TPB) void fwht_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t outer,
                                                   int n, int log2n, int inner, int n_in, int n_out, int OPB,
                                                   float sqrt_n) {
  __shared__ float lds[LDS_FLOATS];
  const int64_t o0 = (int64_t)blockIdx.x * OPB;
  const int nob = (int)min((int64_t)OPB, outer - o0);   // outer rows of this block
  const int TC = OPB * inner, LD = TC + 1;
  // rows [n_in, n): the zero padding
  for (int e = threadIdx.x; e < (n - n_in) * TC; e += TPB) {
    const int c = n_in + e / TC, t = e - (e / TC) * TC;
    lds[c * LD + t] = 0.f;
  }
  // rows [0, n_in): linear sweep over the block's contiguous input
  {
    const int per_o = n_in * inner, total = nob * per_o;
    const float* __restrict__ xb = x + o0 * per_o;
    for (int e = threadIdx.x; e < OPB * per_o; e += TPB) {
      const int ol = e / per_o, r = e - ol * per_o;
      const int c = r / inner, ii = r - c * inner;
      lds[c * LD + ol * inner + ii] = (e < total) ? xb[e] : 0.f;
    }
  }
  __syncthreads();
  // butterflies
  const int pairs = (n >> 1) * TC;
  for (int s = 0; s < log2n; ++s) {
    const int h = 1 << s;
    for (int e = threadIdx.x; e < pairs; e += TPB) {
      int p = e / TC, t = e - p * TC;
      int c = ((p >> s) << (s + 1)) | (p & (h - 1));  // index with bit s clear
      float a = lds[c * LD + t], b = lds[(c + h) * LD + t];
      lds[c * LD + t] = a + b;
      lds[(c + h) * LD + t] = a - b;
    }
    __syncthreads();
  }
  // store the first n_out entries: linear sweep over the block's contiguous output
  {
    const int per_o = n_out * inner, total = nob * per_o;
    float* __restrict__ yb = y + o0 * per_o;
    for (int e = threadIdx.x; e < total; e += TPB) {
      const int ol = e / per_o, r = e - ol * per_o;
      const int c = r / inner, ii = r - c * inner;
      yb[e] = lds[c * LD + ol * inner + ii] / sqrt_n;
    }
  }
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
  // butterflies
  const int pairs = (n >> 1) * TC;
  for (int s = 0; s < log2n; ++s) {
    const int h = 1 << s;
    for (int e = threadIdx.x; e < pairs; e += TPB) {
      int p = e / TC, t = e - p * TC;
      int c = ((p >> s) << (s + 1)) | (p & (h - 1));  // index with bit s clear
      float a = lds[c * LD + t], b = lds[(c + h) * LD + t];
      lds[c * LD + t] = a + b;
      lds[(c + h) * LD + t] = a - b;
    }
    __syncthreads();
  }
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
  if ((int64_t)n * inner > 8192) {   // one outer row does not fit the LDS tile: column-gather variant
    int TC = 8192 / n;
    if (TC > 32) TC = 32;
    int64_t ncols = outer * inner;
    int64_t blocks = (ncols + TC - 1) / TC;
    if (blocks > 0x7fffffffLL) return NQ_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(fwht_cols_kernel, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), x, y, ncols, n, log2n, inner,
                       n_in, n_out, TC, sqrtf((float)n));
    return nq_launch_status();
  }
  int OPB = (int)(8192 / ((int64_t)n * inner));   // outer rows per workgroup
  const int cap = (int)((32 + inner - 1) / inner); // ~32 columns per workgroup keeps the grid large
  if (OPB > cap) OPB = cap;
  if (OPB < 1) OPB = 1;
  int64_t blocks = (outer + OPB - 1) / OPB;
  if (blocks > 0x7fffffffLL) return NQ_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(fwht_kernel, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), x, y, outer, n, log2n, (int)inner, n_in,
                     n_out, OPB, sqrtf((float)n));
  return nq_launch_status();
}
