// Vector (VALU) kernels for the decoder's head convolution: C_in -> C_out <= 4 channels, 3x3 or 5x5 or 1x1, over the
// full-resolution image (HNeRV/NeRV head_layer: 37|24 -> 3, k=3, 640x1280; reference models/HNeRV.py:42, :63-64).
//
// With 3 output channels a GEMM tile is >80 % padding, and the op is HBM-bound anyway (arithmetic intensity
// 2*3*9 = 54 flop per 4-byte input element, ~13 flop/B << the 19.7 flop/B fp32 ridge): forward and data gradient
// each read or write the 242 MB activation tensor once.  So these are streaming kernels:
//   head_fwd   : thread = 4 consecutive pixels, input patch staged in LDS per 4-channel chunk, weights via
//                scalar loads, fused bias + tanh*0.5+0.5 (OutImg, models/_layers.py:10-16)
//   head_dgrad : thread = 4 pixels, the C_out x k x (4+k-1) neighbourhood of dY lives in registers, loop over C_in,
//                fused multiply by the saved gelu'(z) and PixelUnshuffle store (same contract as NQ_EPI_DGRAD_GELU)
// (The weight gradient of the head stays on the MFMA split-K kernel of conv_wgrad_impl.h: a VALU variant with one
//  thread per (ci,kh,kw) was measured slower, 0.66-0.80 ms vs 0.56 ms.)
// Roofline: HBM (8 TB/s spec / 6.3 TB/s achievable); algorithmic bytes = 4*B*H*W*(C_in + C_out) per launch
// (+ the same again for z in head_dgrad).
#include <cstdlib>
#include <type_traits>

#include "nq_common.h"

namespace {

constexpr int MAXCO = 4;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int I0, int N, class F>
__device__ __forceinline__ void hf2_rows(F&& f) {
  if constexpr (I0 < N) {
    f(std::integral_constant<int, I0>{});
    hf2_rows<I0 + 1, N>(f);
  }
}

// ------------------------------------------------------------------------------------------------ forward
template <int KS>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wt, int ld,
                                                       const float* __restrict__ bias, float* __restrict__ y, int Cin,
                                                       int H, int W, int CO, int epi, int tiles_x) {
  constexpr int KK = KS * KS, PAD = KS / 2;
#ifndef NQ_HEAD_CCH
#define NQ_HEAD_CCH 4
#endif
  constexpr int TH = 16, TW = 64, CCH = NQ_HEAD_CCH;
  constexpr int PH = TH + KS - 1;
  // LDS rows: [OFF-PAD, OFF) left halo | [OFF, OFF+64) interior (16-byte aligned) | [OFF+64, OFF+64+PAD) right halo.
  // The interior is staged with 16-byte global loads / ds_write_b128 (4.5 per thread per chunk instead of 18 scalar
  // loads: the kernel is bound by the number of memory instructions), only the 2*PAD halo columns are scalar.
  constexpr int OFF = 4, PWS = 72;
  constexpr int NQUAD = CCH * PH * (TW / 4), NHALO = CCH * PH * 2 * PAD;
  __shared__ __attribute__((aligned(16))) float patch[CCH * PH * PWS];

  const int tid = threadIdx.x;
  const int row = tid >> 4, c4 = (tid & 15) * 4;
  // 1-D grid over (image, tile), XCD-chunked (nq_xcd_chunk): neighbouring tiles share halo rows / lines behind one L2
  const int lid = nq_xcd_chunk((int)blockIdx.x, (int)gridDim.x);
  const int tiles = tiles_x * ((H + TH - 1) / TH), tile = lid % tiles;
  const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
  const int x0 = tile_x * TW, y0 = tile_y * TH, b = lid / tiles;
  const int64_t HW = (int64_t)H * W;
  const float* __restrict__ xb = x + (int64_t)b * Cin * HW;
  const bool w4 = (W & 3) == 0;

  f32x2 acc2[MAXCO][2];
#pragma unroll
  for (int co = 0; co < MAXCO; ++co) acc2[co][0] = acc2[co][1] = f32x2{0.f, 0.f};

  for (int c0 = 0; c0 < Cin; c0 += CCH) {
    __syncthreads();
    for (int e = tid; e < NQUAD; e += 256) {
      const int ci = e / (PH * (TW / 4)), rem = e - ci * (PH * (TW / 4));
      const int r = rem / (TW / 4), q = rem - r * (TW / 4);
      const int gy = y0 - PAD + r, gx = x0 + 4 * q, cig = c0 + ci;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (cig < Cin && gy >= 0 && gy < H && gx < W) {
        const float* p = xb + (int64_t)cig * HW + (int64_t)gy * W + gx;
        if (w4) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          v.x = p[0];
          if (gx + 1 < W) v.y = p[1];
          if (gx + 2 < W) v.z = p[2];
          if (gx + 3 < W) v.w = p[3];
        }
      }
      *reinterpret_cast<float4*>(patch + (ci * PH + r) * PWS + OFF + 4 * q) = v;
    }
    if constexpr (PAD > 0) {
      for (int e = tid; e < NHALO; e += 256) {
        const int ci = e / (PH * 2 * PAD), rem = e - ci * (PH * 2 * PAD);
        const int r = rem / (2 * PAD), h = rem - r * (2 * PAD);
        const int c = (h < PAD) ? h - PAD : TW + (h - PAD);   // column relative to x0: [-PAD, 0) or [TW, TW+PAD)
        const int gy = y0 - PAD + r, gx = x0 + c, cig = c0 + ci;
        float v = 0.f;
        if (cig < Cin && gy >= 0 && gy < H && gx >= 0 && gx < W) v = xb[(int64_t)cig * HW + (int64_t)gy * W + gx];
        patch[(ci * PH + r) * PWS + OFF + c] = v;
      }
    }
    __syncthreads();
    const int cmax = min(CCH, Cin - c0);
    for (int ci = 0; ci < cmax; ++ci) {
      const float* __restrict__ wc = wt + (int64_t)(c0 + ci) * KK * ld;  // wt[(ci*KK + tap)*ld + co]
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
        float in[4 + KS - 1];
        const float* pr = patch + (ci * PH + row + kh) * PWS + (OFF - PAD) + c4;
#pragma unroll
        for (int j = 0; j < 4 + KS - 1; ++j) in[j] = pr[j];
#pragma unroll
        for (int co = 0; co < MAXCO; ++co) {
          if (co < CO) {
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
              const float wv = wc[(kh * KS + kw) * ld + co];  // wave-uniform -> scalar load
              // two pixels per instruction (v_pk_fma_f32: the fp32 vector peak assumes packed math); per element the
              // same fused multiply-add as fmaf
              const f32x2 w2 = {wv, wv};
              acc2[co][0] = __builtin_elementwise_fma(w2, f32x2{in[kw], in[kw + 1]}, acc2[co][0]);
              acc2[co][1] = __builtin_elementwise_fma(w2, f32x2{in[kw + 2], in[kw + 3]}, acc2[co][1]);
            }
          }
        }
      }
    }
  }
  const int gy = y0 + row;
  if (gy >= H) return;
#pragma unroll
  for (int co = 0; co < MAXCO; ++co) {
    if (co >= CO) break;
    const float bv = bias ? bias[co] : 0.f;
    float* __restrict__ yo = y + ((int64_t)b * CO + co) * HW + (int64_t)gy * W + x0 + c4;
    float o[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float v = acc2[co][p >> 1][p & 1] + bv;
      o[p] = (epi == NQ_EPI_TANH) ? tanhf(v) * 0.5f + 0.5f : v;
    }
    if (w4 && x0 + c4 + 3 < W) {
      *reinterpret_cast<float4*>(yo) = make_float4(o[0], o[1], o[2], o[3]);   // one 16-byte store per channel
    } else {
#pragma unroll
      for (int p = 0; p < 4; ++p)
        if (x0 + c4 + p < W) yo[p] = o[p];
    }
  }
}

// ------------------------------------------------------------------------------------------------ forward, streaming
// Round 3.  The LDS-staged kernel above runs at 2.3 TB/s (wait_any 0.64: 4-channel staging rounds between two barriers,
// nothing in flight while a workgroup computes).  This one has no LDS and no barrier: a WAVE owns a strip of 256 columns
// (one 16-byte load per lane and (channel, row): 1 KiB per wave-instruction) and slides down R output rows with the
// 3 x CO x 4 partial sums of the three output rows an input row contributes to in registers, so every activation is
// fetched from memory ONCE per R + 2 rows (R = 5: 1.4 x the tensor through L2, the halo rows shared with the row blocks
// above / below, which the XCD-chunked order runs behind the same L2).  The horizontal halo comes from the neighbouring
// lanes by DPP wave shifts; the two pixels outside the strip are one extra dword load per (channel, row) with two live
// lanes.  Loads are bounds-checked buffer loads (rows / columns outside the image read as zero, no branch), issued D
// steps ahead through a register queue; the weights sit in LDS ({co0..co3} per tap, one fill per workgroup) and are read
// one channel ahead by broadcast 16-byte reads.  Same fp32 fused multiply-adds in the same (ci, kh, kw) order per output as head_fwd_kernel.
struct HeadFwd2Args {
  int B, Cin, H, W, ld, strips, rblocks;
  unsigned x_bytes;
};

// (the pointers are kernel PARAMETERS with __restrict__: only then does the compiler know that the stores to y cannot
// change the weights, and keeps their loads on the scalar unit -- as vector loads they shared the vmcnt queue with the
// activation loads and every use drained it)
// wt[(ci*9 + tap)*ld + co], zero for co >= CO (nq_weight_layouts)
template <int R, int D, int NCO, bool TANH>
__global__ __launch_bounds__(256) void head_fwd2_kernel(const float* __restrict__ x_, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, float* __restrict__ y_,
                                                        HeadFwd2Args a) {
  constexpr int KS = 3, KK = 9;
  constexpr unsigned OOB = 0xFFFFFF00u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the weights, {co0..co3} per (channel, tap), once per workgroup (Cin * 9 * 16 bytes); the only barrier of the kernel
  extern __shared__ __attribute__((aligned(16))) f32x4 wl[];
  for (int e = threadIdx.x; e < a.Cin * KK; e += 256) wl[e] = *reinterpret_cast<const f32x4*>(wt + (int64_t)e * a.ld);
  __syncthreads();
  const int total = a.B * a.strips * a.rblocks;
  const int wid = nq_xcd_chunk((int)blockIdx.x, (int)gridDim.x) * 4 + wave;
  if (wid >= total) return;   // whole wave
  const int by = wid % a.rblocks, t0 = wid / a.rblocks;
  const int sx = t0 % a.strips, b = t0 / a.strips;
  const int H = a.H, W = a.W, Cin = a.Cin;
  const int y0 = by * R, x0 = sx * 256, gx = x0 + 4 * lane;
  const unsigned HWb = (unsigned)H * (unsigned)W * 4u, Wb = (unsigned)W * 4u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x_), 0, (int)a.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(y_, 0, (int)((unsigned)a.B * NCO * HWb), 0x00020000);
  // per-lane column offsets: the quad, and the one pixel outside the strip that lane 0 (left) / lane 63 (right) fetches
  const unsigned col_q = (gx < W) ? (unsigned)gx * 4u : OOB;
  const int ex = (lane == 0) ? x0 - 1 : x0 + 256;
  const unsigned col_e = ((lane == 0 || lane == 63) && ex >= 0 && ex < W) ? (unsigned)ex * 4u : OOB;
  const unsigned base_b = (unsigned)b * (unsigned)Cin * HWb;

  // queue of loads in flight: linear step t = row * cpad + ci over the R + 2 input rows
  const int cpad = (Cin + D - 1) / D * D;
  f32x4 qv[D];
  float qe[D];
  int l_ci = 0, l_row = 0;   // (channel, input row relative to y0 - 1) of the NEXT load to issue: wave-uniform
  auto issue = [&](f32x4& v, float& e) {
    const int iy = y0 - 1 + l_row;
    const bool ok = l_ci < Cin && iy >= 0 && iy < H && l_row < R + 2;
    const unsigned ro = base_b + (unsigned)l_ci * HWb + (unsigned)iy * Wb;
    const unsigned m = ok ? 0xFFFFFFFFu : 0u;   // bit masks, not ?: on the offsets -- no branch around the loads
    const unsigned oq = ((ro + col_q) & m) | (OOB & ~m), oe = ((ro + col_e) & m) | (OOB & ~m);
    v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (col_q == OOB) ? OOB : oq, 0, 0));
    e = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (col_e == OOB) ? OOB : oe, 0, 0));
    const int wrap = (l_ci + 1 == cpad) ? 1 : 0;   // scalar selects, no branch
    l_ci = wrap ? 0 : l_ci + 1;
    l_row += wrap;
  };
#pragma unroll
  for (int j = 0; j < D; ++j) issue(qv[j], qe[j]);

  f32x2 acc[3][NCO][2];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int co = 0; co < NCO; ++co) acc[s][co][0] = acc[s][co][1] = f32x2{0.f, 0.f};

  float bv[NCO];
#pragma unroll
  for (int co = 0; co < NCO; ++co) bv[co] = bias ? bias[co] : 0.f;

  hf2_rows<0, R + 2>([&](auto iyr_c) {
    constexpr int iyr = decltype(iyr_c)::value;   // input row y0 - 1 + iyr feeds output rows y0 + iyr - kh, kh = 0..2
    // the weights of one channel: 9 broadcast 16-byte LDS reads {co0..co3} per tap, fetched ONE channel ahead into the
    // other register set (LDS returns in order: the compiler waits with a counted lgkmcnt, the next set stays in flight)
    auto load_w = [&](int ci, f32x4 (&w)[KK]) {
      const f32x4* __restrict__ wr = wl + min(ci, Cin - 1) * KK;
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
        if (iyr - kh < 0 || iyr - kh >= R) continue;   // compile-time: only the taps this input row uses
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) w[kh * KS + kw] = wr[kh * KS + kw];
      }
    };
    auto slot = [&](int ci, int j, f32x4 (&w)[KK], f32x4 (&wn)[KK]) {
      const f32x4 v = qv[j];
      const float e = qe[j];
      issue(qv[j], qe[j]);
      load_w(ci + 1, wn);
      // halo pixels from the neighbouring lanes; the strip's outer pixels from the edge load
      float lf = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[3]), 0x138, 0xF, 0xF, false));
      float rt = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[0]), 0x130, 0xF, 0xF, false));
      lf = (lane == 0) ? e : lf;
      rt = (lane == 63) ? e : rt;
      const float in[6] = {lf, v[0], v[1], v[2], v[3], rt};
#pragma unroll
      for (int kh = 0; kh < KS; ++kh) {
        const int rel = iyr - kh;               // output row relative to y0
        if (rel < 0 || rel >= R) continue;      // compile-time after unrolling
        const int s = rel % 3;
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
#pragma unroll
          for (int co = 0; co < NCO; ++co) {
            const float wv = w[kh * KS + kw][co];
            const f32x2 w2 = {wv, wv};
            acc[s][co][0] = __builtin_elementwise_fma(w2, f32x2{in[kw], in[kw + 1]}, acc[s][co][0]);
            acc[s][co][1] = __builtin_elementwise_fma(w2, f32x2{in[kw + 2], in[kw + 3]}, acc[s][co][1]);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the queue slots in program order: the younger loads stay in flight
    };
    f32x4 wA[KK], wB[KK];
    load_w(0, wA);
    static_assert(D % 2 == 0, "the two weight register sets alternate per queue slot");
#pragma unroll 1
    for (int c0 = 0; c0 < cpad; c0 += D) {
#pragma unroll
      for (int j = 0; j < D; j += 2) {
        slot(c0 + j, j, wA, wB);
        slot(c0 + j + 1, j + 1, wB, wA);
      }
    }
    // output row y0 + iyr - 2 has now received its three input rows
    if constexpr (iyr >= 2) {
      constexpr int rel = iyr - 2, s = rel % 3;
      const int oy = y0 + rel;
      // bounds-checked buffer stores at an out-of-range offset for rows / columns outside the image: no branch around a
      // memory instruction anywhere in the kernel (a divergent branch with stores in it made the compiler fall back to
      // vmcnt(0) for every later wait, i.e. drain the load queue at every step)
      const unsigned so = (oy < H && gx < W) ? ((unsigned)b * NCO * (unsigned)H + (unsigned)oy) * Wb + (unsigned)gx * 4u : OOB;
#pragma unroll
      for (int co = 0; co < NCO; ++co) {
        f32x4 o;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const float v = acc[s][co][p >> 1][p & 1] + bv[co];
          o[p] = TANH ? tanhf(v) * 0.5f + 0.5f : v;
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o), rs_y,
                                               so == OOB ? OOB : so + (unsigned)co * HWb, 0, 0);
        acc[s][co][0] = acc[s][co][1] = f32x2{0.f, 0.f};
      }
    }
  });
}

// ------------------------------------------------------------------------------------------------ data gradient
// R2 = 1: r == 2 with W % 4 == 0 and H % 2 == 0 checked by the host -> only the 8-byte un-shuffle stores are compiled
// (the generic path's 64-bit per-element addressing otherwise sets the kernel's VGPR count and halves the occupancy).
template <int KS, int R2, int NCO, int PX>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((KS <= 3 && NCO <= 3 && R2 && PX == 4) ? 4 : 1, 8))) void head_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ wt, int ld,
                                                         const float* __restrict__ zprev, float* __restrict__ out,
                                                         int Cin, int H, int W, int CO, int r, int tiles_x) {
  // NB naming follows the data-gradient use: "CO" (<= 4) = channels of the INPUT dy, "Cin" = channels of the OUTPUT;
  // wt[(co*KK + tap)*ld + ci] is the (already tap-flipped) wt_bwd operand of nq_weight_layouts.
  constexpr int KK = KS * KS, PAD = KS / 2;
  // PX pixels per thread (4, or 8 for the r == 2 fast path: un-shuffled stores become 16 bytes, 2 loads + 2 stores per
  // 8 pixels and channel instead of 2 + 4 -- the kernel is bound by the number of global memory instructions)
  constexpr int TW = 64, TPR = TW / PX, TH = 256 / TPR;
  constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
  constexpr int PWS = (PW + 3) / 4 * 4;
  __shared__ __attribute__((aligned(16))) float patch[NCO * PH * PWS];
  // weights of one output channel ci, contiguous: wl[ci][co*KK + tap] (row padded to a multiple of 4 floats).  The
  // operand arrives k-major ([co*KK + tap][ld]): reading it per ci would be NCO*KK separate scalar loads; from LDS it is
  // NCO*KK/4 broadcast ds_read_b128.
  constexpr int KKP = (KK + 3) / 4 * 4;        // taps of one input channel, padded to whole 16-byte vectors
  constexpr int WROW = NCO * KKP;
  constexpr int WL_MAX = 64;   // output channels cached per pass (64 * 112 * 4 B = 28 KB for k = 5)
  __shared__ __attribute__((aligned(16))) float wl[WL_MAX * WROW];

  const int tid = threadIdx.x;
  const int row = tid / TPR, c4 = (tid % TPR) * PX;
  // 1-D grid over (image, tile), XCD-chunked (nq_xcd_chunk): neighbouring tiles share halo rows / lines behind one L2
  const int lid = nq_xcd_chunk((int)blockIdx.x, (int)gridDim.x);
  const int tiles = tiles_x * ((H + TH - 1) / TH), tile = lid % tiles;
  const int tile_x = tile % tiles_x, tile_y = tile / tiles_x;
  const int x0 = tile_x * TW, y0 = tile_y * TH, b = lid / tiles;
  const int64_t HW = (int64_t)H * W;
  const float* __restrict__ dyb = dy + (int64_t)b * CO * HW;

  for (int e = tid; e < CO * PH * PW; e += 256) {
    int co = e / (PH * PW), rem = e - co * (PH * PW);
    int rr = rem / PW, c = rem - rr * PW;
    int gy = y0 - PAD + rr, gx = x0 - PAD + c;
    float v = 0.f;
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = dyb[(int64_t)co * HW + (int64_t)gy * W + gx];
    patch[(co * PH + rr) * PWS + c] = v;
  }
  __syncthreads();
  // the thread's dY neighbourhood: CO x KS rows x (4+KS-1) columns, kept in registers for the whole C_in loop
  float nb[NCO][KS][PX + KS - 1];
#pragma unroll
  for (int co = 0; co < NCO; ++co)
#pragma unroll
    for (int kh = 0; kh < KS; ++kh)
#pragma unroll
      for (int j = 0; j < PX + KS - 1; ++j) nb[co][kh][j] = (co < CO) ? patch[(co * PH + row + kh) * PWS + c4 + j] : 0.f;

  const int gy = y0 + row, gx0 = x0 + c4;
  const bool active = (gy < H && gx0 < W);   // inactive threads still help staging the weights and join the barriers
  const bool full = (gx0 + PX - 1 < W);
#pragma unroll 1
  for (int ci = 0; ci < Cin; ++ci) {
    if (ci % WL_MAX == 0) {   // (re)fill the weight cache for output channels [ci, ci + WL_MAX)
      __syncthreads();
      const int nci = min(WL_MAX, Cin - ci);
      for (int e = tid; e < nci * WROW; e += 256) {
        const int cl = e / WROW, j = e - cl * WROW;
        const int co = j / KKP, tap = j - co * KKP;
        wl[e] = (co < CO && tap < KK) ? wt[(int64_t)(co * KK + tap) * ld + ci + cl] : 0.f;
      }
      __syncthreads();
    }
    if (!active) continue;
    float acc[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) acc[p] = 0.f;
    const float* __restrict__ wrow = wl + (ci % WL_MAX) * WROW;
#pragma unroll
    for (int co = 0; co < NCO; ++co) {
      if (co < CO) {
        // the KK weights of (ci, co): KKP/4 broadcast 16-byte LDS reads (same address in every lane), consumed before the
        // next input channel's are requested -- keeps ~28 weight registers live instead of all NCO*KK
        float wv[KKP];
#pragma unroll
        for (int q = 0; q < KKP / 4; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(wrow + co * KKP + 4 * q);
          wv[4 * q] = t.x; wv[4 * q + 1] = t.y; wv[4 * q + 2] = t.z; wv[4 * q + 3] = t.w;
        }
#pragma unroll
        for (int kh = 0; kh < KS; ++kh)
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
#pragma unroll
            for (int p = 0; p < PX; ++p) acc[p] = fmaf(wv[kh * KS + kw], nb[co][kh][p + kw], acc[p]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int64_t zi = ((int64_t)b * Cin + ci) * HW + (int64_t)gy * W + gx0;
    if (zprev) {
      if (full && (W & 3) == 0) {
#pragma unroll
        for (int q = 0; q < PX / 4; ++q) {
          const float4 zv = *reinterpret_cast<const float4*>(zprev + zi + 4 * q);
          acc[4 * q] *= zv.x; acc[4 * q + 1] *= zv.y; acc[4 * q + 2] *= zv.z; acc[4 * q + 3] *= zv.w;
        }
      } else {
#pragma unroll
        for (int p = 0; p < PX; ++p)
          if (gx0 + p < W) acc[p] *= zprev[zi + p];
      }
    }
    if constexpr (R2) {
      // un-shuffle: channel ci*4 + (y%2)*2 + x%2 at (y/2, x/2): even pixels -> plane j=0, odd pixels -> plane j=1
      const int Ho = H >> 1, Wo = W >> 1;
      const int64_t base = ((((int64_t)b * Cin + ci) * 4 + (gy & 1) * 2) * Ho + (gy >> 1)) * (int64_t)Wo + (gx0 >> 1);
      if constexpr (PX == 8) {
        *reinterpret_cast<float4*>(out + base) = make_float4(acc[0], acc[2], acc[4], acc[6]);
        *reinterpret_cast<float4*>(out + base + (int64_t)Ho * Wo) = make_float4(acc[1], acc[3], acc[5], acc[7]);
      } else {
        *reinterpret_cast<float2*>(out + base) = make_float2(acc[0], acc[2]);
        *reinterpret_cast<float2*>(out + base + (int64_t)Ho * Wo) = make_float2(acc[1], acc[3]);
      }
    } else if (r == 1) {
#pragma unroll
      for (int p = 0; p < PX; ++p)
        if (gx0 + p < W) out[zi + p] = acc[p];
    } else if (PX == 4 && r == 2 && full && (W & 3) == 0 && (H & 1) == 0) {
      // un-shuffle: channel ci*4 + (y%2)*2 + x%2 at (y/2, x/2): pixels {0,2} -> plane j=0, {1,3} -> plane j=1
      const int Ho = H >> 1, Wo = W >> 1;
      const int64_t base = ((((int64_t)b * Cin + ci) * 4 + (gy & 1) * 2) * Ho + (gy >> 1)) * (int64_t)Wo + (gx0 >> 1);
      *reinterpret_cast<float2*>(out + base) = make_float2(acc[0], acc[2]);
      *reinterpret_cast<float2*>(out + base + (int64_t)Ho * Wo) = make_float2(acc[1], acc[3]);
    } else {
      const int Ho = H / r, Wo = W / r, rr2 = r * r;
#pragma unroll
      for (int p = 0; p < PX; ++p) {
        const int gx = gx0 + p;
        if (gx < W) {
          const int yq = gy / r, xq = gx / r;
          const int ch = ci * rr2 + (gy - yq * r) * r + (gx - xq * r);
          out[(((int64_t)b * Cin * rr2 + ch) * Ho + yq) * (int64_t)Wo + xq] = acc[p];
        }
      }
    }
  }
}

}  // namespace

extern "C" {

// Selection rule shared with conv.hip: the vector path serves C_out <= 4.
int nq_head_supported(int Cout, int k) { return Cout <= MAXCO && (k == 1 || k == 3 || k == 5); }

int nq_head_forward(const float* x, const float* wt, int ld, const float* bias, float* y, int B, int Cin, int H, int W,
                    int Cout, int k, int epi, hipStream_t st) {
  // 3x3 heads with whole 16-byte quads per row and a tensor below 4 GiB (32-bit buffer offsets): the streaming kernel
  const char* hv = getenv("NQ_HEAD_FWD");   // NQ_HEAD_FWD=1: the LDS-staged kernel (A/B runs; read per call)
  const bool v1 = hv && hv[0] == '1';
  if (k == 3 && Cout == 3 && (W & 3) == 0 && (ld & 3) == 0 && (int64_t)B * Cin * H * W * 4 < 0xFFFFFF00ll && !v1) {
    constexpr int R = 5, D = 4;
    HeadFwd2Args a;
    a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.ld = ld;
    a.strips = (W + 255) / 256; a.rblocks = (H + R - 1) / R;
    a.x_bytes = (unsigned)((int64_t)B * Cin * H * W * 4);
    const int waves = B * a.strips * a.rblocks;
    dim3 g2((unsigned)((waves + 3) / 4)), blk2(256);
    const size_t lds = (size_t)Cin * 9 * 16;
    if (lds > 64 * 1024) return NQ_ERR_UNSUPPORTED;
    if (epi == NQ_EPI_TANH) hipLaunchKernelGGL((head_fwd2_kernel<R, D, 3, true>), g2, blk2, lds, st, x, wt, bias, y, a);
    else hipLaunchKernelGGL((head_fwd2_kernel<R, D, 3, false>), g2, blk2, lds, st, x, wt, bias, y, a);
    return nq_launch_status();
  }
  const int tiles_x = (W + 63) / 64, tiles = tiles_x * ((H + 15) / 16);
  dim3 g((unsigned)(tiles * B)), blk(256);
  switch (k) {
    case 1: hipLaunchKernelGGL(head_fwd_kernel<1>, g, blk, 0, st, x, wt, ld, bias, y, Cin, H, W, Cout, epi, tiles_x); break;
    case 3: hipLaunchKernelGGL(head_fwd_kernel<3>, g, blk, 0, st, x, wt, ld, bias, y, Cin, H, W, Cout, epi, tiles_x); break;
    default: hipLaunchKernelGGL(head_fwd_kernel<5>, g, blk, 0, st, x, wt, ld, bias, y, Cin, H, W, Cout, epi, tiles_x); break;
  }
  return nq_launch_status();
}

// dy (B,Cout,H,W) -> out = d/dx (B,Cin,H,W) [* gelu'(zprev)] stored PixelUnshuffle(r)-ed
int nq_head_dgrad(const float* dy, const float* wt, int ld, const float* zprev, float* out, int B, int Cin, int H, int W,
                  int Cout, int k, int r, hipStream_t st) {
  // r == 2 fast path: 8 pixels per thread (tile 32 x 64) when W % 8 == 0 and k <= 3, else 4 (tile 16 x 64)
  const bool r2 = (r == 2) && (W % 4 == 0) && (H % 2 == 0);
  const bool px8 = r2 && (W % 8 == 0) && k <= 3;
  const int th = px8 ? 32 : 16;
  const int tiles_x = (W + 63) / 64, tiles = tiles_x * ((H + th - 1) / th);
  dim3 g((unsigned)(tiles * B)), blk(256);
#define NQ_HD(KS_, R2_, PX_)                                                                                          \
  do {                                                                                                                \
    if (Cout <= 3)                                                                                                    \
      hipLaunchKernelGGL((head_dgrad_kernel<KS_, R2_, 3, PX_>), g, blk, 0, st, dy, wt, ld, zprev, out, Cin, H, W, Cout, r, \
                         tiles_x);                                                                                    \
    else                                                                                                              \
      hipLaunchKernelGGL((head_dgrad_kernel<KS_, R2_, 4, PX_>), g, blk, 0, st, dy, wt, ld, zprev, out, Cin, H, W, Cout, r, \
                         tiles_x);                                                                                    \
  } while (0)
  switch (k) {
    case 1: if (px8) NQ_HD(1, 1, 8); else if (r2) NQ_HD(1, 1, 4); else NQ_HD(1, 0, 4); break;
    case 3: if (px8) NQ_HD(3, 1, 8); else if (r2) NQ_HD(3, 1, 4); else NQ_HD(3, 0, 4); break;
    default: if (r2) NQ_HD(5, 1, 4); else NQ_HD(5, 0, 4); break;
  }
#undef NQ_HD
  return nq_launch_status();
}

}  // extern "C"
