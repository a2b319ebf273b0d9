// Implicit-GEMM convolution on the fp32 MFMA pipe (v_mfma_f32_16x16x4_f32), stride 1, pad KS/2, NCHW.
// Included once per kernel size (NQ_KS = 1, 3, 5) so the three sizes compile in parallel.
//
// GEMM view:  D[co][pixel] = sum_k  Wt[k][co] * Xpatch[k][pixel],   k = (ci, kh, kw)
//   A operand = weights (MFMA rows = output channels), B operand = activations (MFMA columns = pixels), so
//   the accumulator has 16 consecutive pixels of one row on 16 lanes (coalesced stores) and 4 consecutive
//   output channels in the 4 registers of a lane (= the r*r sub-pixels of PixelShuffle for r=2).
//
// Workgroup = 4 waves = 4 image rows x 32 columns of one frame, all MT = 16*MI output channels of one
// channel tile.  Wave w owns row w: 2 pixel blocks x MI channel blocks = 2*MI accumulators (8*MI VGPRs).
//
// K loop: a "slice" = CI input channels x KS taps of one kernel row kh.  The 4 lane groups (lane>>4) of the
// 16x16x4 MFMA each walk their own CI/4 channels, so every LDS read is base-VGPR + immediate offset
// (sum order over k is free in a GEMM).  Per slice the weights [CI*KS][MT] are streamed from the k-major
// copy of the (fake-quantised) weight; the activation patch [CI][4+KS-1][32+KS-1] (zero-filled halo) is
// staged once per channel group and reused for all KS kernel rows.  Both are double-buffered through
// registers: global loads for slice s+1 are issued before the MFMAs of slice s, written to LDS after them,
// one barrier per slice.  LDS plane/row strides are padded so that the two lane groups sharing an LDS
// service cycle hit disjoint bank halves.
//
// Roofline: MFMA-bound (fp32 matrix peak 157.3 TFLOP/s); algorithmic flops = 2*Cout*Cin*KS^2*H*W*B.
#include <type_traits>

#include "nq_common.h"

#ifndef NQ_KS
#error "define NQ_KS before including conv_igemm_impl.h"
#endif

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct ConvArgs {
  const float* x;
  const float* wt;
  const float* bias;
  float* y;
  float* z;
  int B, Cin, H, W, Cout, ld, r, epi, tiles_x, tiles, co_tiles, ncg, nsplit, in_gelu;
  const float* zprev;  // NQ_EPI_DGRAD_GELU: pre-activation of the layer below, (B,Cout,H,W)
  float* slab;  // [nsplit][B][Cout][H][W] partial sums when nsplit > 1
};

constexpr int KS = NQ_KS;
constexpr int KK = KS * KS;
constexpr int PAD = KS / 2;
constexpr int CI = (KS == 5) ? 4 : (KS == 3) ? 8 : 16;  // input channels per slice
constexpr int CIQ = CI / 4;                              // channels per lane group
constexpr int TH = 4, TW = 32;
constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
constexpr int WROWS1 = CI * KS;  // weight rows per kernel row kh

constexpr int pad_to_mod32(int v, int mult, int want) {  // smallest v' >= v with (mult*v') % 32 == want
  while ((mult * v) % 32 != want) ++v;
  return v;
}
constexpr int PS = pad_to_mod32(PH * PW, CIQ, 16);  // patch plane stride (floats)
constexpr int PATCH_FLOATS = CI * PS;
constexpr int PE = CI * PH * PW;               // patch elements to stage
constexpr int PPT = (PE + 255) / 256;          // per thread

// compile-time loop: f(integral_constant<int, I>) for I in [I0, N)
template <int I0, int N, class F>
__device__ __forceinline__ void igemm_steps(F&& f) {
  if constexpr (I0 < N) {
    f(std::integral_constant<int, I0>{});
    igemm_steps<I0 + 1, N>(f);
  }
}

__device__ __forceinline__ float gelu_exact(float v) { return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f)); }

// Kernel rows per slice: narrow channel tiles (MI <= 4: the data gradients 148->44, 176->53, 848->64) would otherwise
// run only 6*MI MFMAs between barriers; they take a whole channel group (all KS kernel rows) per slice instead.
constexpr int khs_for(int mi) { return (mi <= 4) ? KS : 1; }

template <int MI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_igemm_kernel(ConvArgs a) {
  constexpr int MT = 16 * MI;
  constexpr int KHS = khs_for(MI);
  constexpr int WROWS = WROWS1 * KHS;
  // (CIQ*KS*LDW) % 32 == 16 and LDW % 4 == 0
  constexpr int LDW = [] {
    int v = MT;
    while ((CIQ * KS * v) % 32 != 16 || (v % 4) != 0) ++v;
    return v;
  }();
  constexpr int W_FLOATS = WROWS * LDW;
  constexpr int WF4 = WROWS * (MT / 4);          // float4 loads per slice
  constexpr int WPT = (WF4 + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const patch0 = smem;
  float* const wl0 = smem + 2 * PATCH_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  // 1-D grid, XCD-chunked (nq_xcd_chunk): logical id = (z * tiles + tile) * co_tiles + co_tile
  const int lid = nq_xcd_chunk((int)blockIdx.x, (int)gridDim.x);
  const int lt = lid / a.co_tiles, tile = lt % a.tiles, bz = lt / a.tiles;
  const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
  const int x0 = tile_x * TW, y0 = tile_y * TH;
  const int co0 = (lid % a.co_tiles) * MT;
  const int b = bz / a.nsplit, split = bz - b * a.nsplit;
  const int H = a.H, W = a.W, Cin = a.Cin;
  const float* __restrict__ xb = a.x + (int64_t)b * Cin * H * W;
  const float* __restrict__ wt = a.wt + co0;
  const int ld = a.ld;
  const bool in_gelu = a.in_gelu != 0;  // the input is a pre-activation: apply GELU while staging (gelu(0)=0 keeps the halo)

  // ---- staging helpers ------------------------------------------------------------------------
  float pv[PPT];
  f32x4 wv[WPT];
  // per-thread patch element coordinates are slice-invariant: precompute offsets (or -1 when outside the image)
  int poff[PPT];   // offset inside one input plane, -1 = zero fill
  int pci[PPT];    // channel inside the group
  int plds[PPT];   // LDS offset
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    int e = tid + i * 256;
    int ci = e / (PH * PW), rem = e - ci * (PH * PW);
    int r = rem / PW, c = rem - r * PW;
    int gy = y0 - PAD + r, gx = x0 - PAD + c;
    bool ok = (e < PE) && gy >= 0 && gy < H && gx >= 0 && gx < W;
    poff[i] = ok ? gy * W + gx : -1;
    pci[i] = ci;
    plds[i] = (e < PE) ? ci * PS + r * PW + c : -1;
  }
#define NQ_LOAD_PATCH(CG)                                                                         \
  _Pragma("unroll") for (int i = 0; i < PPT; ++i) {                                               \
    int cig = (CG) * CI + pci[i];                                                                 \
    pv[i] = (poff[i] >= 0 && cig < Cin) ? xb[(int64_t)cig * H * W + poff[i]] : 0.f;               \
  }
#define NQ_STORE_PATCH(DST)                                                                       \
  _Pragma("unroll") for (int i = 0; i < PPT; ++i) {                                               \
    if (plds[i] >= 0) (DST)[plds[i]] = in_gelu ? gelu_exact(pv[i]) : pv[i];                       \
  }
#define NQ_LOAD_W(CG, KH)                                                                         \
  _Pragma("unroll") for (int i = 0; i < WPT; ++i) {                                               \
    int f = tid + i * 256;                                                                        \
    if (i + 1 < WPT || f < WF4) {                                                                 \
      int row = f / (MT / 4), c4 = f - row * (MT / 4);                                            \
      int khl = row / WROWS1, r1 = row - khl * WROWS1;                                            \
      int ci = r1 / KS, kw = r1 - ci * KS;                                                        \
      int grow = (((CG) * CI + ci) * KS + (KH) + khl) * KS + kw;                                  \
      wv[i] = *reinterpret_cast<const f32x4*>(wt + (int64_t)grow * ld + c4 * 4);                 \
    }                                                                                             \
  }
#define NQ_STORE_W(DST)                                                                           \
  _Pragma("unroll") for (int i = 0; i < WPT; ++i) {                                               \
    int f = tid + i * 256;                                                                        \
    if (i + 1 < WPT || f < WF4) {                                                                 \
      int row = f / (MT / 4), c4 = f - row * (MT / 4);                                            \
      *reinterpret_cast<f32x4*>((DST) + row * LDW + c4 * 4) = wv[i];                             \
    }                                                                                             \
  }

  f32x4 acc[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    acc[mi][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    acc[mi][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int a_base = (kq * CIQ * KS) * LDW + l16;
  const int b_base = (kq * CIQ) * PS + wave * PW + l16;

  // ---- prologue -------------------------------------------------------------------------------
  // split-K: this workgroup covers channel groups [cg_lo, cg_hi)
  const int cg_lo = (int)(((int64_t)a.ncg * split) / a.nsplit);
  const int cg_hi = (int)(((int64_t)a.ncg * (split + 1)) / a.nsplit);
  const int nslices = (cg_hi - cg_lo) * (KS / KHS);
  NQ_LOAD_PATCH(cg_lo)
  NQ_LOAD_W(cg_lo, 0)
  NQ_STORE_PATCH(patch0)
  NQ_STORE_W(wl0)
  __syncthreads();

  int cg = cg_lo, kh = 0;
  for (int s = 0; s < nslices; ++s) {
    int ncg_ = cg, nkh = kh + KHS;
    if (nkh == KS) {
      nkh = 0;
      ncg_ = cg + 1;
    }
    const bool more = (s + 1 < nslices);
    const bool new_patch = more && (nkh == 0);
    if (more) {
      NQ_LOAD_W(ncg_, nkh)
    }
    if (new_patch) {
      NQ_LOAD_PATCH(ncg_)
    }

    const float* __restrict__ pb = patch0 + ((cg - cg_lo) & 1) * PATCH_FLOATS + b_base + kh * PW;
    const float* __restrict__ wb = wl0 + (s & 1) * W_FLOATS + a_base;
    // software-pipelined fragments: the LDS reads of step st+1 are issued ahead of the MFMAs of step st
    // (hipcc otherwise funnels every A fragment through one register pair and exposes the LDS latency
    // after every 4 MFMAs); sched_barrier pins "reads of the next step, then MFMAs of this step".
    {
      constexpr int STEPS = KHS * CIQ * KS;
      float af0[MI], af1[MI], bf0[2], bf1[2];
#define NQ_LDFRAG(AF, BF, ST)                                                                         \
  {                                                                                                   \
    constexpr int khl_ = (ST) / (CIQ * KS), s1_ = (ST)-khl_ * (CIQ * KS);                             \
    constexpr int t_ = s1_ / KS, kw_ = s1_ - t_ * KS;                                                 \
    BF[0] = pb[t_ * PS + khl_ * PW + kw_];                                                            \
    BF[1] = pb[t_ * PS + khl_ * PW + kw_ + 16];                                                       \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                                 \
        AF[mi] = wb[(khl_ * WROWS1 + t_ * KS + kw_) * LDW + mi * 16];                                 \
  }
#define NQ_MFMAS(AF, BF)                                                                              \
  _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                                 \
    acc[mi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[mi], BF[0], acc[mi][0], 0, 0, 0);            \
    acc[mi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[mi], BF[1], acc[mi][1], 0, 0, 0);            \
  }
      NQ_LDFRAG(af0, bf0, 0)
      igemm_steps<0, STEPS>([&](auto st_c) {
        constexpr int st = decltype(st_c)::value;
        if constexpr ((st & 1) == 0) {
          if constexpr (st + 1 < STEPS) NQ_LDFRAG(af1, bf1, st + 1)
          __builtin_amdgcn_sched_barrier(0);
          NQ_MFMAS(af0, bf0)
          __builtin_amdgcn_sched_barrier(0);
        } else {
          if constexpr (st + 1 < STEPS) NQ_LDFRAG(af0, bf0, st + 1)
          __builtin_amdgcn_sched_barrier(0);
          NQ_MFMAS(af1, bf1)
          __builtin_amdgcn_sched_barrier(0);
        }
      });
#undef NQ_LDFRAG
#undef NQ_MFMAS
    }
    if (more) {
      float* wdst = wl0 + ((s + 1) & 1) * W_FLOATS;
      NQ_STORE_W(wdst)
    }
    if (new_patch) {
      float* pdst = patch0 + ((ncg_ - cg_lo) & 1) * PATCH_FLOATS;
      NQ_STORE_PATCH(pdst)
    }
    __syncthreads();
    cg = ncg_;
    kh = nkh;
  }

  // ---- epilogue -------------------------------------------------------------------------------
  const int py = y0 + wave;
  if (py >= H) return;
  const int Cout = a.Cout;
  if (a.nsplit > 1) {  // raw partial sums; bias + epilogue are applied by conv_splitk_finish_kernel
    float* __restrict__ slab = a.slab + ((int64_t)split * a.B + b) * Cout * H * W;
    igemm_steps<0, MI>([&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = co0 + mi * 16 + 4 * kq + reg;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int px = x0 + ni * 16 + l16;
          if (co < Cout && px < W) slab[((int64_t)co * H + py) * W + px] = acc[mi][ni][reg];
        }
      }
    });
    return;
  }
  const int epi = a.epi;
  const int r = a.r, rr = a.r * a.r;
  const int64_t HW = (int64_t)H * W;
  const int cob = co0 + 4 * kq;  // this lane's first channel of block mi is cob + 16*mi (multiple of 4)
  if ((epi == NQ_EPI_PS || epi == NQ_EPI_PS_GELU) && (r == 2 || r == 4)) {
    // PixelShuffle fast paths: the 4 registers of a lane are 4 consecutive conv channels = a full 2x2 output
    // block (r=2) or one output row segment of 4 pixels (r=4) -> 8/16-byte stores, contiguous across lanes
    const int C = Cout / rr;
    const bool want_act = (epi == NQ_EPI_PS_GELU);
    igemm_steps<0, MI>([&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
      const int co = cob + mi * 16;
      if (co >= Cout) return;  // Cout % 4 == 0 for r in {2,4}
      const float4 bv = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
      const int c = co / rr;
      const int si = (r == 4) ? ((co >> 2) & 3) : 0;
      const int64_t rowbase = (((int64_t)b * C + c) * (H * r) + (int64_t)py * r + si) * ((int64_t)W * r);
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int px = x0 + ni * 16 + l16;
        if (px >= W) continue;
        const f32x4 v = acc[mi][ni];
        const float v0 = v[0] + bv.x, v1 = v[1] + bv.y, v2 = v[2] + bv.z, v3 = v[3] + bv.w;
        float g0, g1, g2, g3, d0 = v0, d1 = v1, d2 = v2, d3 = v3;  // PS: z = conv; PS_GELU: z = gelu'(conv)
        if (want_act) {
          nq_gelu_pair(v0, g0, d0); nq_gelu_pair(v1, g1, d1); nq_gelu_pair(v2, g2, d2); nq_gelu_pair(v3, g3, d3);
        }
        if (r == 2) {
          const int64_t o0 = rowbase + (int64_t)px * 2, o1 = o0 + (int64_t)W * 2;
          *reinterpret_cast<float2*>(a.z + o0) = make_float2(d0, d1);
          *reinterpret_cast<float2*>(a.z + o1) = make_float2(d2, d3);
          if (want_act) {
            *reinterpret_cast<float2*>(a.y + o0) = make_float2(g0, g1);
            *reinterpret_cast<float2*>(a.y + o1) = make_float2(g2, g3);
          }
        } else {
          const int64_t o0 = rowbase + (int64_t)px * 4;
          *reinterpret_cast<float4*>(a.z + o0) = make_float4(d0, d1, d2, d3);
          if (want_act) *reinterpret_cast<float4*>(a.y + o0) = make_float4(g0, g1, g2, g3);
        }
      }
    });
    return;
  }
  if (epi == NQ_EPI_DGRAD_GELU) {
    // data gradient w.r.t. the pre-activation below: acc * gelu'(z) (zprev = the derivative saved by the forward
    // epilogue, same NCHW layout as this conv's output), stored un-shuffled (conv layout of the layer below: channel
    // c*r*r + (y%r)*r + x%r at (y/r, x/r)); r == 1 keeps the layout.  Per-channel base + per-pixel offsets.
    const int Wo = W / r, plane = (int)(HW / rr), yq = py / r;
    int in_off[2], out_off[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int px = x0 + ni * 16 + l16, xq = px / r;
      in_off[ni] = (px < W) ? py * W + px : -1;
      out_off[ni] = ((py - yq * r) * r + (px - xq * r)) * plane + yq * Wo + xq;
    }
    igemm_steps<0, MI>([&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = cob + mi * 16 + reg;
        if (co >= Cout) continue;
        const float bv = a.bias ? a.bias[co] : 0.f;
        const int64_t cbase = ((int64_t)b * Cout + co) * HW;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          if (in_off[ni] >= 0) a.y[cbase + out_off[ni]] = (acc[mi][ni][reg] + bv) * a.zprev[cbase + in_off[ni]];
      }
    });
    return;
  }
  igemm_steps<0, MI>([&](auto mi_c) {
    constexpr int mi = decltype(mi_c)::value;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int co = cob + mi * 16 + reg;
      if (co >= Cout) continue;
      const float bv = a.bias ? a.bias[co] : 0.f;
      int c = 0, si = 0, sj = 0;
      if (epi == NQ_EPI_PS_GELU || epi == NQ_EPI_PS) {
        c = co / rr;
        int rem = co - c * rr;
        si = rem / r;
        sj = rem - si * r;
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int px = x0 + ni * 16 + l16;
        if (px >= W) continue;
        float v = acc[mi][ni][reg] + bv;
        if (epi == NQ_EPI_PS_GELU || epi == NQ_EPI_PS) {
          const int C = Cout / rr;
          int64_t o = (((int64_t)b * C + c) * (H * r) + (int64_t)py * r + si) * ((int64_t)W * r) + (int64_t)px * r + sj;
          if (epi == NQ_EPI_PS_GELU) {
            float gv, dv;
            nq_gelu_pair(v, gv, dv);
            a.y[o] = gv;
            a.z[o] = dv;
          } else {
            a.z[o] = v;
          }
        } else {
          int64_t o = ((int64_t)b * Cout + co) * HW + (int64_t)py * W + px;
          a.y[o] = (epi == NQ_EPI_TANH) ? tanhf(v) * 0.5f + 0.5f : v;
        }
      }
    }
  });
}

#if NQ_KS == 1
// Sums the split-K slabs and applies bias + epilogue (x fastest).  SG = 1: one thread per conv output element.
// SG = 8 (small outputs with many slabs, e.g. 31k outputs x 128 slabs): 8 threads share an output, thread g adds slabs
// g, g+8, ... and the 8 partial sums are combined through LDS in index order -- still a fixed order, 8x the parallelism.
template <int SG>
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(ConvArgs a) {
  constexpr int OG = 256 / SG;
  __shared__ float part[SG][OG];
  const int64_t HW = (int64_t)a.H * a.W;
  const int64_t total = (int64_t)a.B * a.Cout * HW;
  const int o = threadIdx.x % OG, g = threadIdx.x / OG;
  int64_t i = (int64_t)blockIdx.x * OG + o;
  float v = 0.f;
  if (i < total) {
#pragma unroll 8
    for (int s = g; s < a.nsplit; s += SG) v += a.slab[s * total + i];
  }
  if (SG > 1) {
    part[g][o] = v;
    __syncthreads();
    if (g != 0) return;
    v = part[0][o];
#pragma unroll
    for (int j = 1; j < SG; ++j) v += part[j][o];
  }
  if (i >= total) return;
  const int px = (int)(i % a.W);
  const int py = (int)((i / a.W) % a.H);
  const int co = (int)((i / HW) % a.Cout);
  const int b = (int)(i / (HW * a.Cout));
  if (a.bias) v += a.bias[co];
  if (a.epi == NQ_EPI_PS_GELU || a.epi == NQ_EPI_PS) {
    const int r = a.r, rr = r * r, C = a.Cout / rr;
    const int c = co / rr, rem = co - c * rr, si = rem / r, sj = rem - si * r;
    int64_t o = (((int64_t)b * C + c) * (a.H * r) + (int64_t)py * r + si) * ((int64_t)a.W * r) + (int64_t)px * r + sj;
    if (a.epi == NQ_EPI_PS_GELU) {
      float gv, dv;
      nq_gelu_pair(v, gv, dv);
      a.y[o] = gv;
      a.z[o] = dv;
    } else {
      a.z[o] = v;
    }
  } else if (a.epi == NQ_EPI_DGRAD_GELU) {
    const int r = a.r, rr = r * r;
    v *= a.zprev[i];
    if (r == 1) {
      a.y[i] = v;
    } else {
      const int yq = py / r, xq = px / r;
      const int ch = co * rr + (py - yq * r) * r + (px - xq * r);
      a.y[(((int64_t)b * a.Cout * rr + ch) * (a.H / r) + yq) * (int64_t)(a.W / r) + xq] = v;
    }
  } else {
    a.y[i] = (a.epi == NQ_EPI_TANH) ? tanhf(v) * 0.5f + 0.5f : v;
  }
}

extern "C" int nq_conv_splitk_finish(const float* slab, const float* bias, float* y, float* z, const float* zprev, int B,
                                     int H, int W, int Cout, int r, int epi, int nsplit, hipStream_t st) {
  ConvArgs a{};
  a.slab = const_cast<float*>(slab); a.bias = bias; a.y = y; a.z = z; a.zprev = zprev;
  a.B = B; a.H = H; a.W = W; a.Cout = Cout; a.r = r; a.epi = epi; a.nsplit = nsplit;
  int64_t total = (int64_t)B * Cout * H * W;
  if (total < 131072 && nsplit >= 16)
    hipLaunchKernelGGL(conv_splitk_finish_kernel<8>, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, a);
  else
    hipLaunchKernelGGL(conv_splitk_finish_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
  return nq_launch_status();
}
#endif

template <int MI>
int launch_igemm(const ConvArgs& a_in, int tiles, int co_tiles, hipStream_t st) {
  ConvArgs a = a_in;
  a.tiles = tiles; a.co_tiles = co_tiles;
  constexpr int MT = 16 * MI;
  constexpr int LDW = [] {
    int v = MT;
    while ((CIQ * KS * v) % 32 != 16 || (v % 4) != 0) ++v;
    return v;
  }();
  size_t lds = (size_t)(2 * PATCH_FLOATS + 2 * WROWS1 * khs_for(MI) * LDW) * sizeof(float);
  if (int rc = nq_lds_optin<&conv_igemm_kernel<MI>>(lds)) return rc;
  hipLaunchKernelGGL(conv_igemm_kernel<MI>, dim3((unsigned)(tiles * co_tiles * a.B * a.nsplit)), dim3(256), lds, st, a);
  return nq_launch_status();
}

}  // namespace

#define NQ_CAT2(a, b) a##b
#define NQ_CAT(a, b) NQ_CAT2(a, b)

// mi_sel: channel blocks (of 16) per workgroup, chosen by nq_conv_pick_mi().
extern "C" int NQ_CAT(nq_conv_igemm_k, NQ_KS)(const float* x, const float* wt, const float* bias, float* y, float* z,
                                               int B, int Cin, int H, int W, int Cout, int ld, int r, int epi,
                                               int mi_sel, int nsplit, float* slab, int in_gelu, const float* zprev,
                                               hipStream_t st) {
  ConvArgs a;
  a.x = x; a.wt = wt; a.bias = bias; a.y = y; a.z = z;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.ld = ld; a.r = r; a.epi = epi;
  a.tiles_x = (W + TW - 1) / TW;
  a.ncg = (Cin + CI - 1) / CI;
  a.nsplit = nsplit;
  a.slab = slab;
  a.in_gelu = in_gelu;
  a.zprev = zprev;
  int tiles = a.tiles_x * ((H + TH - 1) / TH);
  int co_tiles = (Cout + 16 * mi_sel - 1) / (16 * mi_sel);
  switch (mi_sel) {
    case 1: return launch_igemm<1>(a, tiles, co_tiles, st);
    case 2: return launch_igemm<2>(a, tiles, co_tiles, st);
    case 3: return launch_igemm<3>(a, tiles, co_tiles, st);
    case 4: return launch_igemm<4>(a, tiles, co_tiles, st);
    case 5: return launch_igemm<5>(a, tiles, co_tiles, st);
    case 6: return launch_igemm<6>(a, tiles, co_tiles, st);
    case 8: return launch_igemm<8>(a, tiles, co_tiles, st);
    case 9: return launch_igemm<9>(a, tiles, co_tiles, st);
    case 10: return launch_igemm<10>(a, tiles, co_tiles, st);
    case 11: return launch_igemm<11>(a, tiles, co_tiles, st);
    default: return NQ_ERR_UNSUPPORTED;
  }
}
