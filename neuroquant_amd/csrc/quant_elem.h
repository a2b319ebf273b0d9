// Per-element arithmetic of the AdaRound fake-quant (reference quantization/quantizer.py:288-303) and of its backward with the
// rounding regulariser's gradient (calib_model.py:39-47), shared by quant.hip (single- and multi-tensor launches) and fwht.hip
// (the launches fused with the Hadamard transform): ONE op sequence, so every caller produces the same bits.
#pragma once
#include "nq_common.h"

__device__ __forceinline__ float soft_target_lin(float a) { return nq_sigmoid(a) * (NQ_ZETA - NQ_GAMMA) + NQ_GAMMA; }

// per-element arithmetic shared by the single-tensor and the multi-tensor kernels (identical op sequence)
__device__ __forceinline__ float ada_fwd_elem(float xv, float a, float d, float z, float qmax, int soft, float& xq) {
  float h = soft ? fminf(fmaxf(soft_target_lin(a), 0.f), 1.f) : (a >= 0.f ? 1.f : 0.f);
  float xi = (floorf(xv / d) + h) + z;
  xq = fminf(fmaxf(xi, 0.f), qmax);
  return (xq - z) * d;
}
__device__ __forceinline__ float ada_bwd_elem(float xv, float gyv, float a, float d, float z, float qmax, float reg_weight,
                                              float reg_b) {
    float s = nq_sigmoid(a);
  float lin = s * (NQ_ZETA - NQ_GAMMA) + NQ_GAMMA;
  float h = fminf(fmaxf(lin, 0.f), 1.f);
  float hp = (lin >= 0.f && lin <= 1.f) ? (NQ_ZETA - NQ_GAMMA) * (s * (1.f - s)) : 0.f;
  float xi = (floorf(xv / d) + h) + z;
  float inside = (xi >= 0.f && xi <= qmax) ? 1.f : 0.f;
  float g = gyv * d * inside * hp;
  if (reg_weight != 0.f) {
    // R = w * sum(1 - (2|h-.5|)^b);  dR/dh = -w * b * (2|h-.5|)^(b-1) * 2 * sign(h-.5)
    float c = h - 0.5f;
    float t = fabsf(c) * 2.f;
    float sg = (c > 0.f) ? 1.f : ((c < 0.f) ? -1.f : 0.f);
    // t^(b-1), t in [0,1], b-1 >= 1, as exp2((b-1) * log2 t) on the two hardware transcendentals (relative error ~2e-6 at
    // b = 20; t = 0 -> 2^-inf = 0 like powf): the library powf was half of this kernel's time
    const float tp = __builtin_amdgcn_exp2f((reg_b - 1.f) * __builtin_amdgcn_logf(t));
    float dRdh = -reg_weight * (reg_b * tp) * 2.f * sg;
    g += dRdh * hp;
  }
  return g;
}
