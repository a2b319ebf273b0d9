// Weight/bias gradient of the stride-1 convolution with fp32-equivalent accuracy on the BF16 matrix pipe ("bf16x3",
// see conv_igemm3_impl.h for the operand split).  Included once per kernel size (NQ_KS = 3, 5).
//
// GEMM view as conv_wgrad_impl.h:  dW[co][n] = sum_pixels dY[co][p] * X[n][p],  n = (ci,kh,kw), K = pixels, split over
// workgroups into slabs that are reduced in fixed order (deterministic).  One MFMA k-step = 32 consecutive pixels of
// an image row: lane group kq = lane>>4 holds pixels 8*kq .. 8*kq+7.
//
// Workgroup = 4 waves; tile = MT = 16*MI channels x NT = 64*NI n-values; wave w owns n-columns [16*NI*w, 16*NI*(w+1)).
// LDS per 32-pixel segment (double-buffered, one barrier per segment):
//   dY  [plane hi/lo][kq][MT][8 px] bf16  -> A fragments are conflict-free 16-byte reads
//   x   [CIT][KS][32+KS-1] 32-bit words {hi16,lo16}, row stride == KS, plane stride == KS*KS (mod 32) so that
//       bank(n) = n mod 32: a B fragment = 8 conflict-free 4-byte reads at the lane's (ci,kh,kw) offset + 8 v_perm
// Operands are split (two v_cvt_pk_bf16_f32 + one subtract per element) once, while staging.
//
// Roofline: MFMA bf16, 3 MFMA flops per algorithmic flop (833 TFLOP/s fp32-equivalent); algorithmic flops as fp32.
#include <cstdlib>
#include <type_traits>

#include "nq_common.h"

#ifndef NQ_KS
#error "define NQ_KS before including conv_wgrad3_impl.h"
#endif

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef f32x4 f32x4_u __attribute__((aligned(4)));
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

struct Wgrad3Args {
  const float* x;
  const float* dy;
  float* slab;     // [nsplit][co_pad][n_pad]
  float* slab_db;  // [nsplit][co_pad]
  int B, Cin, H, W, Cout, N, co_pad, n_pad, segs_x, nseg, nsplit;
  int x_split, dy_split;   // producer/consumer kernel: x / dy hold split {hi | lo} words (nq_common.h) instead of floats
};

// Timing experiments only (never in the product build): -DNQ_WG3_ABL=n compiles conv_wgrad3p_kernel WITHOUT one of its
// parts (results are wrong): 1 = load one x row of five, 2 = producers skip the conversion + LDS stores, 3 = consumers
// skip the MFMAs, 4 = one B fragment for all n-blocks, 5 = 2 + 4 (tools/ablate_igemm3.sh builds such libraries).
#ifndef NQ_WG3_ABL
#define NQ_WG3_ABL 0
#endif
constexpr int KS = NQ_KS;
constexpr int KK = KS * KS;
constexpr int PAD = KS / 2;
constexpr int SEG = 32;
constexpr int RW = SEG + KS - 1;
constexpr int PWS = [] {  // x row stride (words): >= RW and == KS (mod 32)
  int v = RW;
  while (v % 32 != KS % 32) ++v;
  return v;
}();
constexpr int PSX = KS * PWS;  // plane stride (== KS*KS mod 32)

template <int I0, int N, class F>
__device__ __forceinline__ void wg3_steps(F&& f) {
  if constexpr (I0 < N) {
    f(std::integral_constant<int, I0>{});
    wg3_steps<I0 + 1, N>(f);
  }
}

// Lane group kq of a k-step holds pixel octet oct_of(kq) = (0, 2, 1, 3)[kq]: the two groups that share an LDS service
// cycle (kq = 0, 1) then read x words 16 apart (disjoint bank halves) instead of 8 apart (2-way conflict).  Any
// bijection works as long as A (dY) and B (x) agree; oct_of is its own inverse.
__device__ __forceinline__ int oct_of(int kq) { return ((kq & 1) << 1) | (kq >> 1); }

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
// {bf16(f0) in the low half, bf16(f1) in the high half}: one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pk_bf16(float f0, float f1) {
  bf16x2_t t = {(__bf16)f0, (__bf16)f1};
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ float lo_as_f32(unsigned pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float hi_as_f32(unsigned pk) { return __builtin_bit_cast(float, pk & 0xffff0000u); }

__device__ __forceinline__ unsigned split_word(float v) {  // {hi16, lo16}
  __bf16 h = (__bf16)v;
  __bf16 l = (__bf16)(v - (float)h);
  return ((unsigned)__builtin_bit_cast(unsigned short, h) << 16) | (unsigned)__builtin_bit_cast(unsigned short, l);
}

template <int MI, int NI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_wgrad3_kernel(Wgrad3Args a) {
  constexpr int MT = 16 * MI, NT = 64 * NI;
  constexpr int CIT = (NT + KK - 2) / KK + 1;
  constexpr int DZ_U4 = 2 * 4 * MT;              // 16-byte units: [plane][kq][MT]
  constexpr int X_WORDS = CIT * PSX;
  constexpr int BUF_BYTES = DZ_U4 * 16 + ((X_WORDS * 4 + 15) / 16) * 16;
  constexpr int DITEMS = 4 * MT;                 // (co, kq) staging items of 8 pixels
  constexpr int DPT = (DITEMS + 255) / 256;
  constexpr int Q = (RW + 3) / 4;
  constexpr int XF = CIT * KS * Q;               // 16-byte x staging slots
  constexpr int XPT4 = (XF + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  // 1-D grid, XCD-chunked (nq_xcd_chunk): logical id = (split * n_tiles + n_tile) * co_tiles + co_tile -- the
  // workgroups that read the same segments (same x rows for every co tile, same dY rows for every n tile) and the splits
  // next to them (x halo rows, shared 128-byte lines at the segment ends) run together behind one L2
  const int lid = nq_xcd_chunk((int)blockIdx.x, (int)gridDim.x);
  const int co_tiles = a.co_pad / MT, n_tiles = a.n_pad / NT;
  const int split = lid / (co_tiles * n_tiles);
  const int n_tile = (lid / co_tiles) % n_tiles;
  const int n0 = n_tile * NT;
  const int co0 = (lid % co_tiles) * MT;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, N = a.N;
  const int ci0 = n0 / KK;
  const int64_t HW = (int64_t)H * W;

  // per-lane B-fragment constants (word offsets)
  int lc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    int n = n0 + (wave * NI + ni) * 16 + l16;
    if (n > N - 1) n = N - 1;
    const int ci = n / KK, rem = n - ci * KK;
    const int kh = rem / KS, kw = rem - kh * KS;
    lc[ni] = (ci - ci0) * PSX + kh * PWS + kw + 8 * oct_of(kq);
  }
  const int a_lane = kq * MT + l16;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db_part[DPT];
#pragma unroll
  for (int i = 0; i < DPT; ++i) db_part[i] = 0.f;
  const bool do_db = (n_tile == 0) && a.slab_db != nullptr;

  const int seg_lo = (int)(((int64_t)a.nseg * split) / a.nsplit);
  const int seg_hi = (int)(((int64_t)a.nseg * (split + 1)) / a.nsplit);

  // ---- staging plan ----
  // dY item e = tid + 256*i -> (co = e >> 2, kq = e & 3): pixels 8*kq..8*kq+7 of row co (two 16-byte loads)
  // x slot  e -> (row = e / Q, q = e % Q), row = (ci_l, r)
  int xoff[XPT4], xlds[XPT4], xrc[XPT4];
#pragma unroll
  for (int i = 0; i < XPT4; ++i) {
    const int e = tid + i * 256;
    const int row = e / Q, q = e - row * Q;
    const int ci_l = row / KS, r = row - ci_l * KS;
    const bool ok = (e < XF) && (ci0 + ci_l < Cin);
    xoff[i] = (ci0 + ci_l) * (int)HW + (r - PAD) * W + 4 * q - PAD;
    xlds[i] = ok ? ci_l * PSX + r * PWS + 4 * q : -1;
    xrc[i] = r * 4096 + 4 * q;
  }
  f32x4 dv[DPT][2], xv[XPT4];
  auto load_seg = [&](int seg) {
    const int xs = seg % a.segs_x;
    const int by = seg / a.segs_x;
    const int y = by % H, b = by / H;
    const int x0 = xs * SEG;
    const float* __restrict__ dyp = a.dy + (int64_t)b * Cout * HW + (int64_t)y * W + x0;
    const float* __restrict__ xp = a.x + (int64_t)b * Cin * HW + (int64_t)y * W + x0;
    const bool seg_full = (x0 + SEG <= W);
    if (y >= PAD && y + PAD < H && x0 >= PAD && x0 + SEG + PAD <= W) {
      // interior segment (the common case): every halo element exists -> no per-element tests, no exec-mask branches
#pragma unroll
      for (int i = 0; i < DPT; ++i) {
        const int e = tid + i * 256;
        const int co = e >> 2, q8 = (e & 3) * 8;
        f32x4 v0 = f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (e < DITEMS && co0 + co < Cout) {
          const float* p = dyp + (int64_t)(co0 + co) * HW + q8;
          v0 = *reinterpret_cast<const f32x4_u*>(p);
          v1 = *reinterpret_cast<const f32x4_u*>(p + 4);
        }
        dv[i][0] = v0;
        dv[i][1] = v1;
      }
#pragma unroll
      for (int i = 0; i < XPT4; ++i) {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (xlds[i] >= 0) v = *reinterpret_cast<const f32x4_u*>(xp + xoff[i]);
        xv[i] = v;
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      const int e = tid + i * 256;
      const int co = e >> 2, q8 = (e & 3) * 8;
      f32x4 v0 = f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
      if (e < DITEMS && co0 + co < Cout) {
        const float* p = dyp + (int64_t)(co0 + co) * HW + q8;
        if (seg_full) {
          v0 = *reinterpret_cast<const f32x4_u*>(p);
          v1 = *reinterpret_cast<const f32x4_u*>(p + 4);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (x0 + q8 + j < W) v0[j] = p[j];
            if (x0 + q8 + 4 + j < W) v1[j] = p[4 + j];
          }
        }
      }
      dv[i][0] = v0;
      dv[i][1] = v1;
    }
#pragma unroll
    for (int i = 0; i < XPT4; ++i) {
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (xlds[i] >= 0) {
        const int gy = y + (xrc[i] >> 12) - PAD, gx0 = x0 + (xrc[i] & 4095) - PAD;
        if (gy >= 0 && gy < H) {
          const float* p = xp + xoff[i];
          if (gx0 >= 0 && gx0 + 3 < W) {
            v = *reinterpret_cast<const f32x4_u*>(p);
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (gx0 + j >= 0 && gx0 + j < W) v[j] = p[j];
          }
        }
      }
      xv[i] = v;
    }
  };
  auto store_seg = [&](unsigned char* buf) {
    u32x4* dz = reinterpret_cast<u32x4*>(buf);
    unsigned* xw = reinterpret_cast<unsigned*>(buf + DZ_U4 * 16);
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      const int e = tid + i * 256;
      if (e < DITEMS) {
        const int co = e >> 2, q = e & 3;
        u32x4 hi, lo;
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float f0 = (j < 2) ? dv[i][0][2 * j] : dv[i][1][2 * j - 4];
          const float f1 = (j < 2) ? dv[i][0][2 * j + 1] : dv[i][1][2 * j - 3];
          s += f0 + f1;
          const unsigned h2 = pk_bf16(f0, f1);   // same values as split_word, packed pairwise: 6 VALU per 2 elements
          hi[j] = h2;
          lo[j] = pk_bf16(f0 - lo_as_f32(h2), f1 - hi_as_f32(h2));
        }
        db_part[i] += s;
        dz[oct_of(q) * MT + co] = hi;       // pixel octet q is held by lane group oct_of(q)
        dz[4 * MT + oct_of(q) * MT + co] = lo;
      }
    }
#pragma unroll
    for (int i = 0; i < XPT4; ++i) {
      if (xlds[i] >= 0) {
        const int cc = xrc[i] & 4095;
        unsigned* d = xw + xlds[i];
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
          const float f0 = xv[i][2 * jp], f1 = xv[i][2 * jp + 1];
          const unsigned h2 = pk_bf16(f0, f1);
          const unsigned l2 = pk_bf16(f0 - lo_as_f32(h2), f1 - hi_as_f32(h2));
          const unsigned w0 = __builtin_amdgcn_perm(h2, l2, 0x05040100u);  // {hi16(f0) << 16 | lo16(f0)}
          const unsigned w1 = __builtin_amdgcn_perm(h2, l2, 0x07060302u);  // same for f1
          if constexpr (4 * Q == RW) {
            d[2 * jp] = w0;
            d[2 * jp + 1] = w1;
          } else {
            if (cc + 2 * jp < RW) d[2 * jp] = w0;
            if (cc + 2 * jp + 1 < RW) d[2 * jp + 1] = w1;
          }
        }
      }
    }
  };

  if (seg_lo < seg_hi) {
    load_seg(seg_lo);
    store_seg(smem);
    __syncthreads();
    if (seg_lo + 1 < seg_hi) load_seg(seg_lo + 1);
  }
  for (int seg = seg_lo; seg < seg_hi; ++seg) {
    const int cur = (seg - seg_lo) & 1;
    const u32x4* __restrict__ dz = reinterpret_cast<const u32x4*>(smem + cur * BUF_BYTES) + a_lane;
    const unsigned* __restrict__ xw = reinterpret_cast<const unsigned*>(smem + cur * BUF_BYTES + DZ_U4 * 16);
    // A fragments (dY) of all channel blocks stay in registers (2 x 4 x MI VGPRs); B fragments (x) are built one n-block
    // at a time, just before their MFMAs, and the next one is assembled while the current block's MFMAs run: fewer
    // live registers than holding all NI B fragments (the 5x6 tile no longer spills) and the MFMAs that accumulate into
    // one register are MI issues apart (product-major order).
    bf16x8 ah[MI], al[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      ah[mi] = __builtin_bit_cast(bf16x8, dz[mi * 16]);
      al[mi] = __builtin_bit_cast(bf16x8, dz[4 * MT + mi * 16]);
    }
    auto build_B = [&](int ni, bf16x8& h, bf16x8& l) {
      unsigned w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] = xw[lc[ni] + j];
      u32x4 hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hi[j] = __builtin_amdgcn_perm(w[2 * j + 1], w[2 * j], 0x07060302u);  // {hi16(w0), hi16(w1) << 16}
        lo[j] = __builtin_amdgcn_perm(w[2 * j + 1], w[2 * j], 0x05040100u);  // {lo16(w0), lo16(w1) << 16}
      }
      h = __builtin_bit_cast(bf16x8, hi);
      l = __builtin_bit_cast(bf16x8, lo);
    };
    bf16x8 bh0, bl0;
    build_B(0, bh0, bl0);
    wg3_steps<0, NI>([&](auto ni_c) {
      constexpr int ni = decltype(ni_c)::value;
      const bf16x8 bh = bh0, bl = bl0;
      if constexpr (ni + 1 < NI) build_B(ni + 1, bh0, bl0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh, acc[mi][ni], 0, 0, 0);
    });
    if (seg + 1 < seg_hi) store_seg(smem + (cur ^ 1) * BUF_BYTES);
    __syncthreads();
    if (seg + 2 < seg_hi) load_seg(seg + 2);
  }

  // ---- write the partial slab ----
  float* __restrict__ slab = a.slab + (int64_t)split * a.co_pad * a.n_pad;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + (wave * NI + ni) * 16 + l16;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = co0 + mi * 16 + 4 * kq + reg;
        slab[(int64_t)co * a.n_pad + n] = acc[mi][ni][reg];
      }
    }
  if (do_db) {  // bias gradient: per-thread fp32 sums of the staged dY values, combined in fixed order
    float* red = reinterpret_cast<float*>(smem);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      const int e = tid + i * 256;
      if (e < DITEMS) red[e] = db_part[i];   // red[co*4 + kq]
    }
    __syncthreads();
    if (tid < MT) a.slab_db[(int64_t)split * a.co_pad + co0 + tid] = (red[4 * tid] + red[4 * tid + 1]) + (red[4 * tid + 2] + red[4 * tid + 3]);
  }
}

// dY staging item e of the producer/consumer kernel -> (channel, k-step, pixel octet); see the image comment in the kernel
struct DItem { int co, sub, q; };
template <int NSUB>
__device__ __forceinline__ DItem ditem(int e) {
  constexpr int G = 4 / NSUB;
  const int qlo = e & 1;
  int t = e >> 1;
  const int sub = t % NSUB;
  t /= NSUB;
  const int co_lo = t % G;
  t /= G;
  const int qhi = t & 1;
  return DItem{(t >> 1) * G + co_lo, sub, 2 * qhi + qlo};
}

constexpr int ni3_for(int mi) { return mi <= 3 ? 6 : (mi <= 4 ? 6 : 6); }

// ------------------------------------------------------------------------------------------------------------------
// Producer / consumer variant (8 waves, one workgroup per CU).  The loop above is bound by instruction ISSUE, not by the
// matrix pipe: per 32-pixel segment a wave spends ~250 instructions on staging (global loads, the bf16 hi/lo split,
// LDS stores) next to the ~200 that feed its 90 MFMAs.  Here the two jobs sit in different waves of the same SIMD:
//   waves 0..3 (consumers): A fragments, just-in-time B fragments, MFMAs -- nothing else, the whole accumulator tile;
//   waves 4..7 (producers): global loads two segments ahead (two register sets), split, LDS stores, bias-gradient sums.
// Wave i and wave i+4 share a SIMD (waves are dealt to SIMDs cyclically), so the producer's VALU work issues in the
// slots the consumer's MFMAs leave free (an MFMA holds the SIMD's vector issue for 8 of its 16 cycles) instead of
// standing in front of them in one in-order stream.  Same LDS images, same arithmetic, same slab layout as above: results
// are bit-identical for an equal split plan.  One barrier per segment, executed by all eight waves.
template <int MI, int NI, int SS, int SY>
// (the narrow streaming variant, NI == 1: few registers, latency-bound -> up to two workgroups per CU, conv3.hip plan_wgrad3)
// (tiles of more than 48 channels hold too many registers for that: one workgroup per CU)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((NI == 1 && MI <= 3) ? 4 : 2, (NI == 1 && MI <= 3) ? 4 : 2))) void conv_wgrad3p_kernel(Wgrad3Args a) {
  // A staged segment = SY image rows x 32*SS pixels = SS*SY MFMA k-steps of 32 pixels, one barrier each.
  //   SY > 1 (rows): the KS-row x halo is shared -- KS+SY-1 rows are staged for SY row-steps instead of KS per step
  //   (SY = 4: 2 rows per step instead of 5, the producers' x work and the x traffic drop by 60 %);
  //   SS > 1 (columns): more bytes in flight per barrier for the narrow head problem (SS = 4).
  constexpr int MT = 16 * MI, NT = 64 * NI;
  constexpr int NR = KS + SY - 1;              // staged x rows per channel
  constexpr int SEG = 32 * SS;
  constexpr int RW = SEG + KS - 1;
  constexpr int PWS = [] {  // x row stride (words): >= RW and == KS (mod 32)
    int v = RW;
    while (v % 32 != KS % 32) ++v;
    return v;
  }();
  constexpr int PSX = [] {  // plane (channel) stride in words: >= NR rows and == KS*KS (mod 32), so that bank(n) = n mod 32
    int v = NR * PWS;
    while (v % 32 != KK % 32) ++v;
    return v;
  }();
  constexpr int CIT = (NT + KK - 2) / KK + 1;
  constexpr int NSUB = SS * SY;                // k-steps per segment; sub = sy*SS + sx
  // dY image, 16-byte units: [k-step: SUBS][plane: 4*MT][kq: MT][co ^ (kq >= 2 ? 4 : 0)].  The A-fragment reads
  // (ds_read_b128, served in 16-lane groups that pair kq 0|1 and 2|3) need the kq stride == 0 (mod 16 units); the
  // producers' ds_write_b128 is served in groups of 8 CONTIGUOUS lanes with bank = unit mod 8, so the 8 items of a group
  // must land on 8 different units mod 8 -- with the plain [k-step][plane][kq][co] image and items ordered (co, row,
  // octet) the 8 lanes of a group held ONE channel: an 8-way conflict, 64 LDS cycles per store instead of 8 (dec5:
  // 10 stores per producer thread and segment).  Here the item order is (co_hi, q_hi, co_lo, k-step, q_lo) (fastest
  // last), the k-step stride is == G (mod 8) and channels of lane groups kq >= 2 are XOR-ed with 4: a group's 8 items
  // {G channels} x {NSUB k-steps} x {octets q, q+1 -> kq = (0,2) or (1,3)} cover all 8 residues.
  static_assert(NSUB == 1 || NSUB == 2 || NSUB == 4, "item order below assumes 1, 2 or 4 k-steps per segment");
  constexpr int G = 4 / NSUB;                  // channels per 8-lane store group
  constexpr int SUBS = 8 * MT + G;             // k-step stride
  constexpr int DZ_U4 = NSUB * SUBS;
  constexpr int X_WORDS = CIT * PSX;
  constexpr int BUF_BYTES = DZ_U4 * 16 + ((X_WORDS * 4 + 15) / 16) * 16;
  constexpr int DITEMS = NSUB * 4 * MT;        // (co, row, pixel octet) staging items
  constexpr int DPT = (DITEMS + 255) / 256;
  constexpr int Q = (RW + 3) / 4;
  constexpr int XF = CIT * NR * Q;
  constexpr int XPT4 = (XF + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid, XCD-chunked (nq_xcd_chunk): logical id = (split * n_tiles + n_tile) * co_tiles + co_tile -- the
  // workgroups that read the same segments (same x rows for every co tile, same dY rows for every n tile) and the splits
  // next to them (x halo rows, shared 128-byte lines at the segment ends) run together behind one L2
  const int lid = nq_xcd_chunk((int)blockIdx.x, (int)gridDim.x);
  const int co_tiles = a.co_pad / MT, n_tiles = a.n_pad / NT;
  const int split = lid / (co_tiles * n_tiles);
  const int n_tile = (lid / co_tiles) % n_tiles;
  const int n0 = n_tile * NT;
  const int co0 = (lid % co_tiles) * MT;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, N = a.N;
  const int ci0 = n0 / KK;
  const int64_t HW = (int64_t)H * W;
  const int seg_lo = (int)(((int64_t)a.nseg * split) / a.nsplit);
  const int seg_hi = (int)(((int64_t)a.nseg * (split + 1)) / a.nsplit);

  if (wave >= 4) {
    // ============================================ producers ============================================
    const int ptid = tid - 256;
    // All global loads are bounds-checked BUFFER loads at 32-bit byte offsets with NO branch around them: an invalid
    // item / an x row outside the image gets an out-of-range offset and reads as zero, only the partial quads at the
    // left / right image edge (and the ragged tail of a row) are masked afterwards, under a wave-uniform branch.  A
    // straight-line load sequence is what lets the compiler wait with a counted vmcnt(N) (see the loop below).
    constexpr unsigned OOB = 0xFFFFFF00u;
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.dy), 0, (int)(unsigned)((int64_t)a.B * Cout * HW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (int)(unsigned)((int64_t)a.B * Cin * HW * 4), 0x00020000);
    unsigned doff[DPT];
    int dq8[DPT];
    bool dok[DPT];
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      const int e = ptid + i * 256;
      const DItem it = ditem<NSUB>(e);
      const int sy = it.sub / SS, sx = it.sub % SS;
      dq8[i] = (sx * 4 + it.q) * 8;
      dok[i] = (e < DITEMS) && (co0 + it.co < Cout);
      doff[i] = (unsigned)(((int64_t)(co0 + it.co) * HW + (int64_t)sy * W + dq8[i]) * 4);
    }
    int xoff[XPT4], xlds[XPT4], xrc[XPT4];
#pragma unroll
    for (int i = 0; i < XPT4; ++i) {
      const int e = ptid + i * 256;
      const int row = e / Q, q = e - row * Q;
      const int ci_l = row / NR, r = row - ci_l * NR;
      const bool ok = (e < XF) && (ci0 + ci_l < Cin);
      xoff[i] = ((ci0 + ci_l) * (int)HW + (r - PAD) * W + 4 * q - PAD) * 4;   // bytes, relative to (b, channel 0, y, x0)
      xlds[i] = ok ? ci_l * PSX + r * PWS + 4 * q : -1;
      xrc[i] = r * 4096 + 4 * q;
    }
    float db_part[DPT];
#pragma unroll
    for (int i = 0; i < DPT; ++i) db_part[i] = 0.f;
    const bool do_db = (n_tile == 0) && a.slab_db != nullptr;

    auto load_seg = [&](int seg, f32x4 (&dv)[DPT][2], f32x4 (&xv)[XPT4]) {
      const int xs = seg % a.segs_x;
      const int by = seg / a.segs_x;
      const int y = (by % (H / SY)) * SY, b = by / (H / SY);   // first of the segment's SY rows
      const int x0 = xs * SEG;
      const unsigned dy_base = (unsigned)((((int64_t)b * Cout) * HW + (int64_t)y * W + x0) * 4);
      const int x_base = (int)((((int64_t)b * Cin) * HW + (int64_t)y * W + x0) * 4);
#pragma unroll
      for (int i = 0; i < DPT; ++i) {
        const unsigned off = dok[i] ? dy_base + doff[i] : OOB;
        dv[i][0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, off, 0, 0));
        dv[i][1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, off + 16, 0, 0));
      }
#pragma unroll
      for (int i = 0; i < XPT4; ++i) {
        const int gy = y + (xrc[i] >> 12) - PAD;
        const bool ok = (xlds[i] >= 0) && gy >= 0 && gy < H && (NQ_WG3_ABL != 1 || (xrc[i] >> 12) == PAD);
        // a quad that starts before the very first element of the tensor (frame 0, channel 0, row 0, left halo) is loaded
        // from offset 0 and shifted into place in store_seg: a negative offset would make the whole load read as zero
        const unsigned off = ok ? (unsigned)max(x_base + xoff[i], 0) : OOB;
        xv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
      }
    };
    // the edge masks are applied HERE, when the values are consumed (masking right after the loads would wait for them)
    auto store_seg = [&](unsigned char* buf, int seg, f32x4 (&dv)[DPT][2], f32x4 (&xv)[XPT4]) {
      const int x0 = (seg % a.segs_x) * SEG;
      if (x0 + SEG > W) {   // ragged last segment of a row: pixels beyond the row end belong to the next row
#pragma unroll
        for (int i = 0; i < DPT; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (x0 + dq8[i] + j >= W) dv[i][0][j] = 0.f;
            if (x0 + dq8[i] + 4 + j >= W) dv[i][1][j] = 0.f;
          }
      }
      if (x0 < PAD || x0 + SEG + PAD > W) {   // halo columns outside the image (left / right edge segments)
        const int by_ = seg / a.segs_x;
        const int by = (by_ / (H / SY)) * H + (by_ % (H / SY)) * SY;   // b*H + y of the segment's first row
        if (ci0 == 0 && by <= PAD && x0 == 0) {   // first rows of frame 0: the quad loaded from offset 0 (see load_seg)
          const int x_base = by * W * 4;
#pragma unroll
          for (int i = 0; i < XPT4; ++i) {
            if (xlds[i] >= 0 && x_base + xoff[i] < 0 && by + (xrc[i] >> 12) - PAD >= 0) {
              const f32x4 v = xv[i];
              if constexpr (PAD == 2) xv[i] = f32x4{0.f, 0.f, v[0], v[1]};
              else if constexpr (PAD == 1) xv[i] = f32x4{0.f, v[0], v[1], v[2]};
            }
          }
        }
#pragma unroll
        for (int i = 0; i < XPT4; ++i) {
          const int gx0 = x0 + (xrc[i] & 4095) - PAD;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (gx0 + j < 0 || gx0 + j >= W) xv[i][j] = 0.f;
        }
      }
      u32x4* dz = reinterpret_cast<u32x4*>(buf);
      unsigned* xw = reinterpret_cast<unsigned*>(buf + DZ_U4 * 16);
#pragma unroll
      for (int i = 0; i < DPT; ++i) {
        const int e = ptid + i * 256;
        if (e < DITEMS) {
          const DItem it = ditem<NSUB>(e);
          const int kqd = oct_of(it.q);             // pixel octet q of a k-step is held by lane group oct_of(q)
          u32x4* const dst = dz + it.sub * SUBS + kqd * MT + (it.co ^ ((kqd & 2) << 1));
          u32x4 hi, lo;
          float s = 0.f;
          if (a.dy_split) {
            // the values are {hi | lo} words already: two v_perm_b32 per pixel pair; the bias gradient sums hi + lo (what the
            // matrix pipe sees of dY) with one packed dot product per half, only in the workgroups that own it
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned w0 = __builtin_bit_cast(unsigned, (j < 2) ? dv[i][0][2 * j] : dv[i][1][2 * j - 4]);
              const unsigned w1 = __builtin_bit_cast(unsigned, (j < 2) ? dv[i][0][2 * j + 1] : dv[i][1][2 * j - 3]);
              hi[j] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);
              lo[j] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);
            }
            if (do_db) {
              const bf16x2_t one = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                // (through named scalars: __builtin_bit_cast applied directly to a vector element reads element 0 with hipcc 7.2)
                const unsigned hj = hi[j], lj = lo[j];
                s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, hj), one, s, false);
                s = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, lj), one, s, false);
              }
            }
          } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float f0 = (j < 2) ? dv[i][0][2 * j] : dv[i][1][2 * j - 4];
            const float f1 = (j < 2) ? dv[i][0][2 * j + 1] : dv[i][1][2 * j - 3];
            s += f0 + f1;
            const unsigned h2 = pk_bf16(f0, f1);
            hi[j] = h2;
            lo[j] = pk_bf16(f0 - lo_as_f32(h2), f1 - hi_as_f32(h2));
          }
          }
          db_part[i] += s;
          dst[0] = hi;
          dst[4 * MT] = lo;
        }
      }
#pragma unroll
      for (int i = 0; i < XPT4; ++i) {
        if (xlds[i] >= 0) {
          const int cc = xrc[i] & 4095;
          unsigned* d = xw + xlds[i];
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            const float f0 = xv[i][2 * jp], f1 = xv[i][2 * jp + 1];
            unsigned w0, w1;
            if (a.x_split) {   // already the {hi | lo} word this image holds per pixel
              w0 = __builtin_bit_cast(unsigned, f0);
              w1 = __builtin_bit_cast(unsigned, f1);
            } else {
              const unsigned h2 = pk_bf16(f0, f1);
              const unsigned l2 = pk_bf16(f0 - lo_as_f32(h2), f1 - hi_as_f32(h2));
              w0 = __builtin_amdgcn_perm(h2, l2, 0x05040100u);
              w1 = __builtin_amdgcn_perm(h2, l2, 0x07060302u);
            }
            if constexpr (4 * Q == RW) {
              d[2 * jp] = w0;
              d[2 * jp + 1] = w1;
            } else {
              if (cc + 2 * jp < RW) d[2 * jp] = w0;
              if (cc + 2 * jp + 1 < RW) d[2 * jp + 1] = w1;
            }
          }
        }
      }
    };

    f32x4 dvA[DPT][2], xvA[XPT4], dvB[DPT][2], xvB[XPT4];
    unsigned char* const buf0 = smem;
    unsigned char* const buf1 = smem + BUF_BYTES;
    // Loads are issued UNCONDITIONALLY (past the end of the split the last segment is simply loaded again and never
    // stored): with a fixed number of younger loads in flight the compiler waits with a counted vmcnt(N) for the set it
    // is about to convert, i.e. the other set's loads stay in flight across the barrier -- a conditional issue forces
    // vmcnt(0) and exposes a full memory latency per segment.
    const int seg_last = seg_hi - 1;
    if (seg_lo < seg_hi) {
      load_seg(seg_lo, dvA, xvA);
      load_seg(min(seg_lo + 1, seg_last), dvB, xvB);
      store_seg(buf0, seg_lo, dvA, xvA);
      load_seg(min(seg_lo + 2, seg_last), dvA, xvA);
    }
    __syncthreads();  // P: segment seg_lo is staged
    // Segments are walked in PAIRS with two barriers per pair on both sides (an odd tail just passes the second barrier):
    // no early exit inside the body, so the in-flight load count the compiler reasons about is the same on every path.
    for (int seg = seg_lo; seg < seg_hi; seg += 2) {
      // consumers work on buf0 (segment seg); segment seg+1 (set B) goes to buf1, then set B is re-armed with seg+3
      if (seg + 1 < seg_hi && NQ_WG3_ABL != 2 && NQ_WG3_ABL != 5) store_seg(buf1, seg + 1, dvB, xvB);
      load_seg(min(seg + 3, seg_last), dvB, xvB);
      __syncthreads();
      // consumers work on buf1 (segment seg+1); segment seg+2 (set A) goes to buf0, set A re-armed with seg+4
      if (seg + 2 < seg_hi && NQ_WG3_ABL != 2 && NQ_WG3_ABL != 5) store_seg(buf0, seg + 2, dvA, xvA);
      load_seg(min(seg + 4, seg_last), dvA, xvA);
      __syncthreads();
    }
    // bias gradient: per-thread fp32 sums of the staged dY values, combined in fixed order (all 8 waves pass the two
    // barriers; the consumers are done with the LDS buffers after the loop's last barrier)
    float* red = reinterpret_cast<float*>(smem);
    if (do_db) {
#pragma unroll
      for (int i = 0; i < DPT; ++i) {
        const int e = ptid + i * 256;
        const DItem it = ditem<NSUB>(e);
        if (e < DITEMS) red[it.co * (4 * NSUB) + it.sub * 4 + it.q] = db_part[i];   // red[co][k-step][octet]
      }
    }
    __syncthreads();
    if (do_db && ptid < MT)
    {
      float t = 0.f;
#pragma unroll
      for (int sub = 0; sub < NSUB; ++sub) {
        const float* r4 = red + (4 * NSUB) * ptid + 4 * sub;
        t += (r4[0] + r4[1]) + (r4[2] + r4[3]);
      }
      a.slab_db[(int64_t)split * a.co_pad + co0 + ptid] = t;
    }
    return;
  }

  // ============================================== consumers ==============================================
  const int l16 = lane & 15, kq = lane >> 4;
  int lc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    int n = n0 + (wave * NI + ni) * 16 + l16;
    if (n > N - 1) n = N - 1;
    const int ci = n / KK, rem = n - ci * KK;
    const int kh = rem / KS, kw = rem - kh * KS;
    lc[ni] = (ci - ci0) * PSX + kh * PWS + kw + 8 * oct_of(kq);
  }
  const int a_lane = kq * MT + (l16 ^ ((kq & 2) << 1));   // the producers' image: channels of kq >= 2 are XOR-ed with 4
  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  __builtin_amdgcn_s_setprio(1);   // the MFMA stream wins issue arbitration against its producer partner
  __syncthreads();  // P
  auto compute = [&](int cur) {
    const u32x4* __restrict__ dz0 = reinterpret_cast<const u32x4*>(smem + cur * BUF_BYTES) + a_lane;
    const unsigned* __restrict__ xw0 = reinterpret_cast<const unsigned*>(smem + cur * BUF_BYTES + DZ_U4 * 16);
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
      const u32x4* __restrict__ dz = dz0 + sub * SUBS;
      const unsigned* __restrict__ xw = xw0 + (sub / SS) * PWS + 32 * (sub % SS);   // row step sy, column step sx
      bf16x8 ah[MI], al[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        ah[mi] = __builtin_bit_cast(bf16x8, dz[mi * 16]);
        al[mi] = __builtin_bit_cast(bf16x8, dz[4 * MT + mi * 16]);
      }
      auto build_B = [&](int ni, bf16x8& h, bf16x8& l) {
        unsigned w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = xw[lc[ni] + j];
        u32x4 hi, lo;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          hi[j] = __builtin_amdgcn_perm(w[2 * j + 1], w[2 * j], 0x07060302u);
          lo[j] = __builtin_amdgcn_perm(w[2 * j + 1], w[2 * j], 0x05040100u);
        }
        h = __builtin_bit_cast(bf16x8, hi);
        l = __builtin_bit_cast(bf16x8, lo);
      };
      bf16x8 bh0, bl0;
      build_B(0, bh0, bl0);
      wg3_steps<0, NI>([&](auto ni_c) {
        constexpr int ni = decltype(ni_c)::value;
        const bf16x8 bh = bh0, bl = bl0;
        if constexpr (ni + 1 < NI) {
          if (NQ_WG3_ABL != 4 && NQ_WG3_ABL != 5) build_B(ni + 1, bh0, bl0);   // ablation 4 (timing only): one B fragment for all n-blocks
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bh, acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[mi], bl, acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mi], bh, acc[mi][ni], 0, 0, 0);
      });
    }
  };
  for (int seg = seg_lo; seg < seg_hi; seg += 2) {   // pairs, two barriers per pair (mirrors the producers)
    if (NQ_WG3_ABL != 3) compute(0);
    __syncthreads();
    if (seg + 1 < seg_hi && NQ_WG3_ABL != 3) compute(1);
    __syncthreads();
  }
  __builtin_amdgcn_s_setprio(0);
  __syncthreads();   // pairs with the producers' bias-gradient barrier

  float* __restrict__ slab = a.slab + (int64_t)split * a.co_pad * a.n_pad;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = n0 + (wave * NI + ni) * 16 + l16;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int co = co0 + mi * 16 + 4 * kq + reg;
        slab[(int64_t)co * a.n_pad + n] = acc[mi][ni][reg];
      }
    }
}

template <int MI, int NI, int SS, int SY>
int launch_wgrad3p(const Wgrad3Args& a_in, hipStream_t st) {
  constexpr int MT = 16 * MI, NT = 64 * NI;
  constexpr int CIT = (NT + KK - 2) / KK + 1;
  constexpr int SEGP = 32 * SS, RWP = SEGP + KS - 1;
  constexpr int PWSP = [] {
    int v = RWP;
    while (v % 32 != KS % 32) ++v;
    return v;
  }();
  constexpr int PSXP = [] {
    int v = (KS + SY - 1) * PWSP;
    while (v % 32 != KK % 32) ++v;
    return v;
  }();
  constexpr int BUF_BYTES = SS * SY * (8 * MT + 4 / (SS * SY)) * 16 + ((CIT * PSXP * 4 + 15) / 16) * 16;   // as in the kernel
  size_t lds = (size_t)2 * BUF_BYTES;
  if (a_in.H % SY != 0) return NQ_ERR_UNSUPPORTED;
  Wgrad3Args a = a_in;
  a.segs_x = (a.W + SEGP - 1) / SEGP;       // segments of SY rows x 32*SS pixels
  a.nseg = a.segs_x * (a.H / SY) * a.B;
  dim3 grid((unsigned)(a.nsplit * (a.n_pad / NT) * (a.co_pad / MT)));
  if (int rc = nq_lds_optin<&conv_wgrad3p_kernel<MI, NI, SS, SY>>(lds)) return rc;
  hipLaunchKernelGGL((conv_wgrad3p_kernel<MI, NI, SS, SY>), grid, dim3(512), lds, st, a);
  return nq_launch_status();
}


template <int MI, int NI>
int launch_wgrad3(const Wgrad3Args& a, hipStream_t st) {
  constexpr int MT = 16 * MI, NT = 64 * NI;
  constexpr int CIT = (NT + KK - 2) / KK + 1;
  constexpr int BUF_BYTES = 2 * 4 * MT * 16 + ((CIT * PSX * 4 + 15) / 16) * 16;
  size_t lds = (size_t)2 * BUF_BYTES;
  dim3 grid((unsigned)(a.nsplit * (a.n_pad / NT) * (a.co_pad / MT)));
  if (int rc = nq_lds_optin<&conv_wgrad3_kernel<MI, NI>>(lds)) return rc;
  hipLaunchKernelGGL((conv_wgrad3_kernel<MI, NI>), grid, dim3(256), lds, st, a);
  return nq_launch_status();
}

}  // namespace

#define NQ_CAT2(a, b) a##b
#define NQ_CAT(a, b) NQ_CAT2(a, b)

// tile: MT = 16*mi_sel channels (mi_sel in 1..5), NT = 64*ni_sel n-values (ni_sel in {5,6,7}: least padding of C_in*k*k,
// or 1 for C_in*k*k <= 64)
extern "C" int NQ_CAT(nq_conv_wgrad3_k, NQ_KS)(const float* x, const float* dy, float* slab, float* slab_db, int B, int Cin,
                                                int H, int W, int Cout, int co_pad, int n_pad, int nsplit, int mi_sel,
                                                int ni_sel, int pc, int fmt, hipStream_t st) {
  Wgrad3Args a;
  a.x_split = fmt & 1; a.dy_split = (fmt >> 1) & 1;   // split {hi | lo} word operands: the row-segment producer/consumer variants only
  if (fmt && !(pc == 1 || pc == 12 || pc == 14)) return NQ_ERR_UNSUPPORTED;
  a.x = x; a.dy = dy; a.slab = slab; a.slab_db = slab_db;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.N = Cin * KK;
  a.co_pad = co_pad; a.n_pad = n_pad;
  a.segs_x = (W + SEG - 1) / SEG;
  a.nseg = a.segs_x * H * B;
  a.nsplit = nsplit;
  if (pc) {   // producer / consumer variant (wide n-tiles only; the plan sized nsplit for one 8-wave workgroup per CU);
              // pc: 1 = one row x 32 pixels per segment, 12 / 14 = 2 / 4 rows x 32 pixels, 4 = one row x 128 pixels (head)
#define NQ_WG3P(MI_, NI_)                                                                   \
  return pc == 14 ? launch_wgrad3p<MI_, NI_, 1, 4>(a, st)                                   \
                  : (pc == 12 ? launch_wgrad3p<MI_, NI_, 1, 2>(a, st) : launch_wgrad3p<MI_, NI_, 1, 1>(a, st));
    if (ni_sel == 1) {   // narrow problems (the role-swapped head gradient): 128-pixel segments keep enough bytes in flight
      if (pc != 4) return NQ_ERR_UNSUPPORTED;
      switch (mi_sel) {
        case 2: return launch_wgrad3p<2, 1, 4, 1>(a, st);
        case 3: return launch_wgrad3p<3, 1, 4, 1>(a, st);
        case 4: return launch_wgrad3p<4, 1, 4, 1>(a, st);   // (heads with 49 .. 80 input channels: the UVG-12M shape has 74)
        case 5: return launch_wgrad3p<5, 1, 4, 1>(a, st);
        default: return NQ_ERR_UNSUPPORTED;
      }
    }
    if (ni_sel == 5) {
      switch (mi_sel) {
        case 3: NQ_WG3P(3, 5)
        case 4: NQ_WG3P(4, 5)
        case 5: NQ_WG3P(5, 5)
        default: return NQ_ERR_UNSUPPORTED;
      }
    }
    if (ni_sel == 6) {
      switch (mi_sel) {
        case 3: NQ_WG3P(3, 6)
        case 4: NQ_WG3P(4, 6)
        case 5: NQ_WG3P(5, 6)
        default: return NQ_ERR_UNSUPPORTED;
      }
    }
    if (ni_sel == 7) {
      switch (mi_sel) {
        case 3: NQ_WG3P(3, 7)
        case 4: NQ_WG3P(4, 7)
        default: return NQ_ERR_UNSUPPORTED;
      }
    }
#undef NQ_WG3P
    return NQ_ERR_UNSUPPORTED;
  }
  if (ni_sel == 1) {
    switch (mi_sel) {
      case 1: return launch_wgrad3<1, 1>(a, st);
      case 2: return launch_wgrad3<2, 1>(a, st);
      case 3: return launch_wgrad3<3, 1>(a, st);
      case 4: return launch_wgrad3<4, 1>(a, st);
      case 5: return launch_wgrad3<5, 1>(a, st);
      default: return NQ_ERR_UNSUPPORTED;
    }
  }
#define NQ_WG3_CASES(NI_)                                  \
  switch (mi_sel) {                                        \
    case 1: return launch_wgrad3<1, NI_>(a, st);           \
    case 2: return launch_wgrad3<2, NI_>(a, st);           \
    case 3: return launch_wgrad3<3, NI_>(a, st);           \
    case 4: return launch_wgrad3<4, NI_>(a, st);           \
    case 5: return launch_wgrad3<5, NI_>(a, st);           \
    default: return NQ_ERR_UNSUPPORTED;                    \
  }
  if (ni_sel == 5) {
    NQ_WG3_CASES(5)
  }
  if (ni_sel == 7) {   // 7 n-blocks only with <= 4 channel blocks (accumulators: 16*MI*NI/... = 112 VGPRs at 4x7)
    switch (mi_sel) {
      case 1: return launch_wgrad3<1, 7>(a, st);
      case 2: return launch_wgrad3<2, 7>(a, st);
      case 3: return launch_wgrad3<3, 7>(a, st);
      case 4: return launch_wgrad3<4, 7>(a, st);
      default: return NQ_ERR_UNSUPPORTED;
    }
  }
  NQ_WG3_CASES(6)
#undef NQ_WG3_CASES
}
