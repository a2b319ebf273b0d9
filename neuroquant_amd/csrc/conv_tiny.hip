// Compact kernels for 1x1 convolutions over a handful of pixels (B*H*W <= 256): the stem and first block of the decoders
// (HNeRV dec0 16->92 and dec1 92->1925 on a 2x4 grid; NeRV stem 160->1160 on 1x1).  A few kFLOP each: the MFMA
// implicit-GEMM kernels spend 10-27 us per launch on them (two workgroups walking a long serial loop with a cold
// instruction cache), a thread-per-output kernel with a short body a few.  Same operands and epilogues as
// nq_conv_forward / nq_conv_wgrad (the data gradient is the forward kernel on the wt_bwd operand); plain fp32 FMA
// chains in channel order (deterministic).
#include "nq_common.h"

namespace {

// y[b][co][p] = bias[co] + sum_ci wt[ci*ld + co] * x[b][ci][p];  thread = one output, co fastest (coalesced weights,
// the activation value is a broadcast)
// KSL > 1 (long reductions, e.g. the data gradient 1925 -> 92): KSL threads share one output, each walks every KSL-th
// input channel, partial sums are combined through LDS in slice order (deterministic).
template <int KSL>
__global__ __launch_bounds__(256) void tiny_pw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                          const float* __restrict__ bias, float* __restrict__ y,
                                                          float* __restrict__ z, const float* __restrict__ zprev, int B,
                                                          int Cin, int H, int W, int Cout, int ld, int r, int epi) {
  constexpr int OPB = 256 / KSL;   // outputs per workgroup
  __shared__ float part[KSL][OPB];
  const int HW = H * W;
  const int o = threadIdx.x % OPB, ks = threadIdx.x / OPB;
  const int64_t t = (int64_t)blockIdx.x * OPB + o;
  const bool live = t < (int64_t)B * HW * Cout;
  const int co = live ? (int)(t % Cout) : 0, bp = live ? (int)(t / Cout) : 0;
  const int b = bp / HW, p = bp - b * HW;
  const float* __restrict__ xp = x + (int64_t)b * Cin * HW + p;
  float acc = 0.f;
#pragma unroll 8
  for (int ci = ks; ci < Cin; ci += KSL) acc = fmaf(wt[(int64_t)ci * ld + co], xp[(int64_t)ci * HW], acc);
  if (KSL > 1) {
    part[ks][o] = acc;
    __syncthreads();
    if (ks != 0) return;
    acc = part[0][o];
#pragma unroll
    for (int j = 1; j < KSL; ++j) acc += part[j][o];
  }
  if (!live) return;
  float v = acc + (bias ? bias[co] : 0.f);
  const int py = p / W, px = p - py * W;
  if (epi == NQ_EPI_PS_GELU || epi == NQ_EPI_PS) {
    const int rr = r * r, C = Cout / rr;
    const int c = co / rr, rem = co - c * rr, si = rem / r, sj = rem - si * r;
    const int64_t o = (((int64_t)b * C + c) * (H * r) + (int64_t)py * r + si) * ((int64_t)W * r) + (int64_t)px * r + sj;
    if (epi == NQ_EPI_PS_GELU) {
      float gv, dv;
      nq_gelu_pair(v, gv, dv);
      y[o] = gv;
      z[o] = dv;
    } else {
      z[o] = v;
    }
  } else if (epi == NQ_EPI_DGRAD_GELU) {
    const int64_t i = ((int64_t)b * Cout + co) * HW + p;
    v *= zprev[i];
    if (r == 1) {
      y[i] = v;
    } else {
      const int rr = r * r, yq = py / r, xq = px / r;
      const int ch = co * rr + (py - yq * r) * r + (px - xq * r);
      y[(((int64_t)b * Cout * rr + ch) * (H / r) + yq) * (int64_t)(W / r) + xq] = v;
    }
  } else {
    const int64_t i = ((int64_t)b * Cout + co) * HW + p;
    y[i] = (epi == NQ_EPI_TANH) ? tanhf(v) * 0.5f + 0.5f : v;
  }
}

// dw[co][ci] = sum_{b,p} dy[b][co][p] * x[b][ci][p];  thread = one weight (ci fastest); the last Cout threads do db
__global__ __launch_bounds__(256) void tiny_pw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dw, float* __restrict__ db, int B, int Cin,
                                                            int HW, int Cout) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t nw = (int64_t)Cout * Cin;
  if (t < nw) {
    const int co = (int)(t / Cin), ci = (int)(t - (int64_t)co * Cin);
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
      const float* __restrict__ dp = dy + ((int64_t)b * Cout + co) * HW;
      const float* __restrict__ xp = x + ((int64_t)b * Cin + ci) * HW;
      if ((HW & 7) == 0) {
        // eight pixels per step, their four 16-byte loads in flight together (the scalar loop waited for two L2 round trips
        // per multiply-add: 8.8 us for HNeRV's dec1); the additions keep the pixel order
        for (int p = 0; p < HW; p += 8) {
          const float4 d0 = *reinterpret_cast<const float4*>(dp + p), d1 = *reinterpret_cast<const float4*>(dp + p + 4);
          const float4 x0 = *reinterpret_cast<const float4*>(xp + p), x1 = *reinterpret_cast<const float4*>(xp + p + 4);
          acc = fmaf(d0.x, x0.x, acc); acc = fmaf(d0.y, x0.y, acc); acc = fmaf(d0.z, x0.z, acc); acc = fmaf(d0.w, x0.w, acc);
          acc = fmaf(d1.x, x1.x, acc); acc = fmaf(d1.y, x1.y, acc); acc = fmaf(d1.z, x1.z, acc); acc = fmaf(d1.w, x1.w, acc);
        }
      } else {
        for (int p = 0; p < HW; ++p) acc = fmaf(dp[p], xp[p], acc);
      }
    }
    dw[t] = acc;
  } else if (db && t < nw + Cout) {
    const int co = (int)(t - nw);
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
      const float* __restrict__ dp = dy + ((int64_t)b * Cout + co) * HW;
      for (int p = 0; p < HW; ++p) acc += dp[p];
    }
    db[co] = acc;
  }
}

}  // namespace

extern "C" {

int nq_tiny_pw_supported(int B, int Cin, int H, int W, int Cout, int k) {
  return k == 1 && (int64_t)B * H * W <= 256 && (int64_t)B * H * W * Cout < (1ll << 31);
}

int nq_tiny_pw_forward(const float* x, const float* wt, const float* bias, float* y, float* z, const float* zprev, int B,
                       int Cin, int H, int W, int Cout, int ld, int r, int epi, hipStream_t st) {
  const int64_t total = (int64_t)B * H * W * Cout;
  // threads per output = slices of the channel sum: each thread walks Cin / KSL channels through a chain of dependent
  // fused multiply-adds fed by two L2-latency loads, so the kernel's time is that chain's length (round 2: 16 slices for
  // the 1925-channel data gradient of HNeRV's dec1 = 120 links, 17.8 us; 64 slices = 30 links; 4 slices for the 92- and
  // 160-channel forwards)
  if (Cin > 1024)
    hipLaunchKernelGGL(tiny_pw_fwd_kernel<64>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, x, wt, bias, y, z, zprev,
                       B, Cin, H, W, Cout, ld, r, epi);
  else if (Cin > 256)
    hipLaunchKernelGGL(tiny_pw_fwd_kernel<16>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, x, wt, bias, y, z, zprev,
                       B, Cin, H, W, Cout, ld, r, epi);
  else if (Cin > 48)
    hipLaunchKernelGGL(tiny_pw_fwd_kernel<4>, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, x, wt, bias, y, z, zprev,
                       B, Cin, H, W, Cout, ld, r, epi);
  else
    hipLaunchKernelGGL(tiny_pw_fwd_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, wt, bias, y, z, zprev,
                       B, Cin, H, W, Cout, ld, r, epi);
  return nq_launch_status();
}

int nq_tiny_pw_wgrad(const float* x, const float* dy, float* dw, float* db, int B, int Cin, int H, int W, int Cout,
                     hipStream_t st) {
  const int64_t total = (int64_t)Cout * Cin + (db ? Cout : 0);
  hipLaunchKernelGGL(tiny_pw_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, dy, dw, db, B, Cin,
                     H * W, Cout);
  return nq_launch_status();
}

}  // extern "C"
