// Weight/bias gradient of the stride-1 convolution on the fp32 MFMA pipe (v_mfma_f32_16x16x4_f32).
// Included once per kernel size (NQ_KS = 1, 3, 5).
//
// GEMM view:  dW[co][n] = sum_pixels  dY[co][p] * X[n][p],   n = (ci, kh, kw) = the OIHW inner index,
//   X[n][p] = x[ci][py + kh - pad][px + kw - pad].  K dimension = pixels (B*H*W): split across workgroups,
//   each writes a partial [co][n] slab; a second kernel sums the slabs in fixed order (deterministic).
//
// Workgroup = 4 waves; tile = MT = 16*MI output channels x NT = 64*NI n-values; wave w owns n-columns
// [w*16*NI, (w+1)*16*NI) for all MT channels: MI*NI accumulators.  It walks "segments" = 32 consecutive
// pixels of one image row: stages dY[MT][32] and the x rows [CIT][KS][32+KS-1] in LDS, then 8 MFMA k-steps.
// Lane j of a B fragment is one n = (ci,kh,kw): its LDS address is a per-lane constant + the pixel offset,
// and the LDS strides are chosen (row stride == KS, plane stride == KS*KS mod 32) so bank(n) = n mod 32.
// The bias gradient (row sums of dY) is accumulated by the n-tile-0 workgroups from the staged dY tile.
//
// Roofline: MFMA-bound; algorithmic flops = 2*Cout*Cin*KS^2*H*W*B.
#include <type_traits>

#include "nq_common.h"

#ifndef NQ_KS
#error "define NQ_KS before including conv_wgrad_impl.h"
#endif

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef f32x4 f32x4_u __attribute__((aligned(4)));  // 16-byte global load from a 4-byte aligned address

struct WgradArgs {
  const float* x;
  const float* dy;
  float* slab;     // [nsplit][co_pad][n_pad]
  float* slab_db;  // [nsplit][co_pad]
  int B, Cin, H, W, Cout, N, co_pad, n_pad, segs_x, nseg, nsplit, x_gelu;
};

constexpr int KS = NQ_KS;
constexpr int KK = KS * KS;
constexpr int PAD = KS / 2;
constexpr int SEG = 32;
constexpr int LDP = 33;  // dY tile row stride
constexpr int PWS = [] {  // x row stride: >= SEG+KS-1 and == KS (mod 32)
  int v = SEG + KS - 1;
  while (v % 32 != KS % 32) ++v;
  return v;
}();
constexpr int PSX = KS * PWS;  // plane stride (== KS*KS mod 32)

__device__ __forceinline__ float wgrad_gelu(float v) { return v * 0.5f * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int I0, int N, class F>
__device__ __forceinline__ void wgrad_steps(F&& f) {
  if constexpr (I0 < N) {
    f(std::integral_constant<int, I0>{});
    wgrad_steps<I0 + 1, N>(f);
  }
}

template <int MI, int NI>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int MT = 16 * MI, NT = 64 * NI;
  constexpr int CIT = (NT + KK - 2) / KK + 1;  // max input channels spanned by an n-tile
  constexpr int DZ_FLOATS = MT * LDP;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int BUF_FLOATS = DZ_FLOATS + CIT * PSX;  // one staging buffer: dY tile + x rows

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  // 1-D grid, XCD-chunked (nq_xcd_chunk): logical id = (split * n_tiles + n_tile) * co_tiles + co_tile
  const int lid = nq_xcd_chunk((int)blockIdx.x, (int)gridDim.x);
  const int co_tiles = a.co_pad / MT, n_tiles = a.n_pad / NT;
  const int split = lid / (co_tiles * n_tiles);
  const int n_tile = (lid / co_tiles) % n_tiles;
  const int n0 = n_tile * NT;
  const int co0 = (lid % co_tiles) * MT;
  const int H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, N = a.N;
  const int ci0 = n0 / KK;
  const int64_t HW = (int64_t)H * W;

  // per-lane B-fragment constants
  int lc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    int n = n0 + (wave * NI + ni) * 16 + l16;
    if (n > N - 1) n = N - 1;  // padding columns recompute the last one; discarded at the store
    int ci = n / KK, rem = n - ci * KK;
    int kh = rem / KS, kw = rem - kh * KS;
    lc[ni] = (ci - ci0) * PSX + kh * PWS + kw;
  }
  const int pxo = ((kq & 1) << 4) + ((kq >> 1) << 3);

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  float db_acc = 0.f;
  const bool do_db = (n_tile == 0) && a.slab_db != nullptr;

  const int seg_lo = (int)(((int64_t)a.nseg * split) / a.nsplit);
  const int seg_hi = (int)(((int64_t)a.nseg * (split + 1)) / a.nsplit);

  // ---- staging plan: 16-byte global loads, per-thread invariants hoisted by hand (a dozen VGPRs) -------------
  // dY tile: row = drow0 + 32*i (i < DROWS), float4 index dq within the 32-pixel row
  // x rows : slot e = tid + 256*i -> (row = e / Q, q = e % Q), row = (ci_l, r); 4-byte aligned 16-byte loads
  constexpr int RW = SEG + KS - 1;
  constexpr int Q = (RW + 3) / 4;
  constexpr int XF = CIT * KS * Q;
  constexpr int XPT4 = (XF + 255) / 256;
  constexpr int DROWS = (MT + 31) / 32;
  const int dq = tid & 7, drow0 = tid >> 3;
  const int d_off0 = (co0 + drow0) * (int)HW + 4 * dq;  // host guarantees Cout*H*W < 2^31
  int xoff[XPT4], xlds[XPT4], xrc[XPT4];                // xrc = r * 4096 + 4q   (both >= 0)
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
  f32x4 dv[DROWS], xv[XPT4];
  auto load_seg = [&](int seg) {
    const int xs = seg % a.segs_x;
    const int by = seg / a.segs_x;
    const int y = by % H, b = by / H;
    const int x0 = xs * SEG;
    const float* __restrict__ dyp = a.dy + (int64_t)b * Cout * HW + (int64_t)y * W + x0;
    const float* __restrict__ xp = a.x + (int64_t)b * Cin * HW + (int64_t)y * W + x0;
    const bool seg_full = (x0 + SEG <= W);
#pragma unroll
    for (int i = 0; i < DROWS; ++i) {
      const int row = drow0 + 32 * i;
      const bool ok = (row < MT) && (co0 + row < Cout);
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (ok) {
        const float* p = dyp + d_off0 + i * 32 * (int)HW;
        if (seg_full) {
          v = *reinterpret_cast<const f32x4_u*>(p);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (x0 + 4 * dq + j < W) v[j] = p[j];
        }
      }
      dv[i] = v;
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
  auto store_seg = [&](float* dzl, float* xl) {
#pragma unroll
    for (int i = 0; i < DROWS; ++i) {
      const int row = drow0 + 32 * i;
      if (row < MT) {
        float* d = dzl + row * LDP + 4 * dq;
        d[0] = dv[i][0]; d[1] = dv[i][1]; d[2] = dv[i][2]; d[3] = dv[i][3];
      }
    }
#pragma unroll
    for (int i = 0; i < XPT4; ++i) {
      if (xlds[i] >= 0) {
        const int cc = xrc[i] & 4095;  // 4q
        float* d = xl + xlds[i];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (cc + j < RW) d[j] = a.x_gelu ? wgrad_gelu(xv[i][j]) : xv[i][j];
      }
    }
  };

  // double-buffered LDS: MFMAs of segment s read buf[s&1] while the registers holding segment s+1 are written
  // to buf[(s+1)&1] right after them; one barrier per segment, global loads run one segment ahead of the stores
  if (seg_lo < seg_hi) {
    load_seg(seg_lo);
    store_seg(smem, smem + DZ_FLOATS);
    __syncthreads();
    if (seg_lo + 1 < seg_hi) load_seg(seg_lo + 1);
  }
  for (int seg = seg_lo; seg < seg_hi; ++seg) {
    const int cur = (seg - seg_lo) & 1;
    const float* __restrict__ dzl = smem + cur * BUF_FLOATS;
    const float* __restrict__ xl = dzl + DZ_FLOATS;
    // ---- 8 k-steps of 4 pixels, fragments software-pipelined one step ahead (see conv_igemm_impl.h) ----
    {
      float af0[MI], af1[MI], bf0[NI], bf1[NI];
#define NQ_LDFRAG(AF, BF, S)                                                                            \
  {                                                                                                     \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) BF[ni] = xl[lc[ni] + pxo + (S)];                  \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) AF[mi] = dzl[(mi * 16 + l16) * LDP + pxo + (S)];  \
  }
#define NQ_MFMAS(AF, BF)                                                                                \
  _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) {                                                   \
    _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                   \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(AF[mi], BF[ni], acc[mi][ni], 0, 0, 0);       \
  }
      NQ_LDFRAG(af0, bf0, 0)
      wgrad_steps<0, 8>([&](auto st_c) {
        constexpr int st = decltype(st_c)::value;
        if constexpr ((st & 1) == 0) {
          if constexpr (st + 1 < 8) NQ_LDFRAG(af1, bf1, st + 1)
          __builtin_amdgcn_sched_barrier(0);
          NQ_MFMAS(af0, bf0)
          __builtin_amdgcn_sched_barrier(0);
        } else {
          if constexpr (st + 1 < 8) NQ_LDFRAG(af0, bf0, st + 1)
          __builtin_amdgcn_sched_barrier(0);
          NQ_MFMAS(af1, bf1)
          __builtin_amdgcn_sched_barrier(0);
        }
      });
#undef NQ_LDFRAG
#undef NQ_MFMAS
    }
    if (do_db && tid < MT) {
      float sacc = 0.f;
#pragma unroll
      for (int px = 0; px < SEG; ++px) sacc += dzl[tid * LDP + px];
      db_acc += sacc;
    }
    if (seg + 1 < seg_hi) {
      float* nb = smem + (cur ^ 1) * BUF_FLOATS;
      store_seg(nb, nb + DZ_FLOATS);
    }
    __syncthreads();
    if (seg + 2 < seg_hi) load_seg(seg + 2);
  }

  // ---- write the partial slab ----
  float* __restrict__ slab = a.slab + (int64_t)split * a.co_pad * a.n_pad;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      int n = n0 + (wave * NI + ni) * 16 + l16;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        int co = co0 + mi * 16 + 4 * kq + reg;
        slab[(int64_t)co * a.n_pad + n] = acc[mi][ni][reg];
      }
    }
  if (do_db && tid < MT) a.slab_db[(int64_t)split * a.co_pad + co0 + tid] = db_acc;
}

constexpr int ni_for(int mi) { return KS == 1 ? 2 : (mi <= 3 ? 6 : (mi <= 6 ? 4 : 3)); }

template <int MI, int NI>
int launch_wgrad(const WgradArgs& a, hipStream_t st) {
  constexpr int MT = 16 * MI, NT = 64 * NI;
  constexpr int CIT = (NT + KK - 2) / KK + 1;
  size_t lds = (size_t)2 * (MT * LDP + CIT * PSX) * sizeof(float);
  dim3 grid((unsigned)(a.nsplit * (a.n_pad / NT) * (a.co_pad / MT)));
  if (int rc = nq_lds_optin<&conv_wgrad_kernel<MI, NI>>(lds)) return rc;
  hipLaunchKernelGGL((conv_wgrad_kernel<MI, NI>), grid, dim3(256), lds, st, a);
  return nq_launch_status();
}

}  // namespace

#define NQ_CAT2(a, b) a##b
#define NQ_CAT(a, b) NQ_CAT2(a, b)

// (mi_sel, ni_sel) chosen by nq_wgrad_pick(); co_pad % (16*mi) == 0 and n_pad % (64*ni) == 0.
extern "C" int NQ_CAT(nq_conv_wgrad_k, NQ_KS)(const float* x, const float* dy, float* slab, float* slab_db, int B,
                                               int Cin, int H, int W, int Cout, int co_pad, int n_pad, int nsplit,
                                               int mi_sel, int ni_sel, int x_gelu, hipStream_t st) {
  WgradArgs a;
  a.x = x; a.dy = dy; a.slab = slab; a.slab_db = slab_db;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.N = Cin * KK;
  a.co_pad = co_pad; a.n_pad = n_pad;
  a.segs_x = (W + SEG - 1) / SEG;
  a.nseg = a.segs_x * H * B;
  a.nsplit = nsplit;
  a.x_gelu = x_gelu;
#define NQ_WG_CASE(MI_) \
  if (mi_sel == MI_ && ni_sel == ni_for(MI_)) return launch_wgrad<MI_, ni_for(MI_)>(a, st);
  NQ_WG_CASE(1)
  NQ_WG_CASE(2)
  NQ_WG_CASE(3)
  NQ_WG_CASE(4)
  NQ_WG_CASE(5)
  NQ_WG_CASE(6)
  NQ_WG_CASE(8)
  NQ_WG_CASE(9)
  NQ_WG_CASE(10)
  NQ_WG_CASE(11)
#undef NQ_WG_CASE
  return NQ_ERR_UNSUPPORTED;
}
