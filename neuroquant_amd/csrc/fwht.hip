// Orthonormal Sylvester-ordered Walsh-Hadamard transform along the C_in axis of a conv weight
// (hadamard_along_channel_weight, quantization/quant_layer.py:16-22), with the zero padding to 2^k
// (:45-49) folded into the load and the [:, :C] slice (:71) folded into the store.
//
// Layout: x [outer][n_in][inner], y [outer][n_out][inner]; a "column" is one (outer, inner) pair and is
// transformed along its n entries.  A workgroup stages TC columns in LDS as [n][TC+1] (lanes run along
// columns -> conflict-free butterflies), runs log2(n) in-place stages (a,b)->(a+b,a-b) in the same order
// as the oracle, scales by 1/sqrt(n) and writes back.  HBM-bound (one read + one write of the weight).
#include "nq_common.h"

namespace {

constexpr int TPB = 256;
constexpr int LDS_FLOATS = 8192 + 1024;

__global__ __launch_bounds__(TPB) void fwht_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t ncols,
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
  int TC = 8192 / n;
  if (TC > 32) TC = 32;
  int64_t ncols = outer * inner;
  int64_t blocks = (ncols + TC - 1) / TC;
  if (blocks > 0x7fffffffLL) return NQ_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(fwht_kernel, dim3((unsigned)blocks), dim3(TPB), 0, nq_s(stream), x, y, ncols, n, log2n, inner, n_in,
                     n_out, TC, sqrtf((float)n));
  return nq_launch_status();
}
