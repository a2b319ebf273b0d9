// Shared device/host helpers for libnqhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/nq_hip.h"

#define NQ_WAVE 64

static inline int nq_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? NQ_OK : NQ_ERR_LAUNCH;
}

static inline hipStream_t nq_s(nq_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// More than 64 KB of dynamic LDS needs an explicit opt-in on the kernel, per device.  One table per kernel (the template
// argument IS the kernel) and per device; lock-free, and the attribute call is idempotent, so concurrent callers on any
// thread / device are safe: the library keeps no other state.
template <auto Kernel>
static inline int nq_lds_optin(size_t lds) {
  if (lds <= 64 * 1024) return NQ_OK;
  static std::atomic<size_t> granted[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return NQ_ERR_LAUNCH;
  std::atomic<size_t>& g = granted[dev & 63];
  if (g.load(std::memory_order_acquire) >= lds) return NQ_OK;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return NQ_ERR_LAUNCH;
  size_t cur = g.load(std::memory_order_relaxed);
  while (cur < lds && !g.compare_exchange_weak(cur, lds, std::memory_order_release)) {
  }
  return NQ_OK;
}

// XCD-aware workgroup order.  The dispatcher deals consecutive workgroup ids round-robin over the 8 XCDs, each with its
// own 4 MiB L2 (MI355X_MICROARCH.md, L2): with the plain id -> tile map, tiles that share halo rows / cache lines /
// an operand panel never meet in one L2 and every shared line is fetched once per XCD.  This bijection hands XCD k (the
// ids with id % 8 == k, in dispatch order) a CONTIGUOUS range of logical ids, so neighbouring tiles run at the same
// time behind the same L2.  Speed and HBM traffic only -- never correctness.
__device__ __forceinline__ int nq_xcd_chunk(int id, int n) {
  const int q = n >> 3, r = n & 7;          // XCD k owns q + (k < r) ids
  const int k = id & 7, i = id >> 3;
  return k * q + (k < r ? k : r) + i;
}

// exact-erf GELU and its derivative (nn.GELU(), reference models/_layers.py:104-105) from ONE exponential: the
// Gaussian exp(-v^2/2) that gelu' needs is also the exponential of erf(v/sqrt 2) = 1 - P(t) exp(-v^2/2), t = 1/(1 + p|v|/sqrt 2)
// (Abramowitz & Stegun 7.1.26, |error| <= 1.5e-7 -- at the rounding level of an fp32 erf; measured against float64 over
// [-12, 12]: gelu 4.7e-7 abs / 3.2e-7 relative, gelu' 3.1e-7, the same as erff+expf gives in fp32).  ~18 VALU
// instructions with two transcendentals instead of ~40: the forward epilogues evaluate this for every activation
// (121 M per dec5 launch).  The forward epilogues store the derivative next to the activation, so no backward kernel
// evaluates erf/exp again.
__device__ __forceinline__ void nq_gelu_pair(float v, float& g, float& dg) {
  const float ax = fabsf(v) * 0.70710678118654752440f;
  const float e = __builtin_amdgcn_exp2f((v * v) * -0.72134752044448170368f);   // exp(-v^2/2) = 2^(-v^2/2 * log2 e)
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, ax, 1.0f));
  float p = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  p = __builtin_fmaf(p, t, 1.421413741f);
  p = __builtin_fmaf(p, t, -0.284496736f);
  p = __builtin_fmaf(p, t, 0.254829592f);
  const float erf_abs = 1.0f - (p * t) * e;
  const float er = __builtin_copysignf(erf_abs, v);
  const float cdf = __builtin_fmaf(0.5f, er, 0.5f);
  g = v * cdf;
  dg = __builtin_fmaf(v, 0.39894228040143267794f * e, cdf);
}

// ---- split ("hi | lo") words: the interchange format of activations / gradients between the bf16x3 kernels (round 4) ----
// A float v travels as ONE 32-bit word {bf16 hi in the upper half, bf16 lo in the lower half}, hi = bf16(v), lo = bf16(v - hi),
// both round-to-nearest-even: exactly the two operands the bf16x3 kernels derive from v when they stage it (split8 in
// conv_igemm3_impl.h, split_word in conv_wgrad3_impl.h), so a consumer that receives the word only re-packs halves (one
// v_perm_b32 per pair) instead of converting (two conversions + a subtraction per value).  Same bytes per element as fp32.
__device__ __forceinline__ unsigned nq_split_word_u(float v) {
  const __bf16 h = (__bf16)v;
  const __bf16 l = (__bf16)(v - (float)h);
  return ((unsigned)__builtin_bit_cast(unsigned short, h) << 16) | (unsigned)__builtin_bit_cast(unsigned short, l);
}
__device__ __forceinline__ float nq_split_word_f(float v) { return __builtin_bit_cast(float, nq_split_word_u(v)); }
// value a split word stands for: hi + lo (what the matrix pipe sees of v)
__device__ __forceinline__ float nq_split_word_value(unsigned w) {
  return __builtin_bit_cast(float, w & 0xffff0000u) + __builtin_bit_cast(float, w << 16);
}

// ---- wave / block reductions (wave = 64 lanes) ----
__device__ __forceinline__ float nq_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ float nq_wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_down(v, o, 64));
  return v;
}
__device__ __forceinline__ float nq_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). Result valid in thread 0. Fixed order -> deterministic.
__device__ __forceinline__ float nq_block_sum(float v, float* smem /* >= 16 floats */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = nq_wave_sum(v);
  __syncthreads();
  if (lane == 0) smem[wave] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0)
    for (int i = 0; i < nw; ++i) r += smem[i];
  return r;
}

// Rectified-sigmoid constants of AdaRound (quantizer.py:274)
#define NQ_GAMMA (-0.1f)
#define NQ_ZETA (1.1f)

__device__ __forceinline__ float nq_sigmoid(float a) { return 1.0f / (1.0f + expf(-a)); }

// Stage 2 of the deterministic two-stage sums: out[0] = (accumulate ? out[0] : 0) + scale * sum(ws[0..nparts)).
static __global__ __launch_bounds__(256) void nq_sum_stage2(const float* __restrict__ ws, int64_t nparts, float scale,
                                                           float* __restrict__ out, int accumulate) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < nparts; i += 256) acc += ws[i];
  float s = nq_block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + s * scale;
}
#define NQ_RED_CHUNK 4096  // elements per stage-1 block (256 threads x 16)
