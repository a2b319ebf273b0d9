// C-ABI entry points of the bf16x3 convolution path (conv_igemm3_impl.h): weight pre-split/re-order + dispatch.
#include <cstdlib>
#include "nq_common.h"
#include "conv3_layout.h"

extern "C" {
int nq_conv_igemm3_k3(const float*, const void*, const float*, float*, float*, const float*, int, int, int, int, int, int, int,
                      int, int, int, float*, hipStream_t);
int nq_conv_igemm3_k5(const float*, const void*, const float*, float*, float*, const float*, int, int, int, int, int, int, int,
                      int, int, int, float*, hipStream_t);
int nq_conv_splitk_finish(const float*, const float*, float*, float*, const float*, int, int, int, int, int, int, int, hipStream_t);
int nq_conv3_nst_k3();
int nq_conv3_nst_k5();
int nq_conv3_nst8_k3();
int nq_conv3_nst8_k5();
int nq_conv3_nstk_k3(int);
int nq_conv3_nstk_k5(int);
int nq_conv_flat3_plan(int, int, int, int, int, int, int*, int*, int*, int*);
int nq_conv_flat3(const float*, const void*, const float*, float*, float*, const float*, float*, int, int, int, int, int, int, int, int,
                  int, int, int, int, hipStream_t);
int nq_conv_wgrad_flat3_ok(int, int, int, int, int, int);
int nq_conv_wgrad_flat3(const float*, const float*, float*, float*, int, int, int, int, int, int, hipStream_t);
int nq_conv_wgrad3_k3(const float*, const float*, float*, float*, int, int, int, int, int, int, int, int, int, int, int, int, hipStream_t);
int nq_conv_wgrad3_k5(const float*, const float*, float*, float*, int, int, int, int, int, int, int, int, int, int, int, int, hipStream_t);
}

namespace {

constexpr int CC = 16;

// channel blocks (of 16) per workgroup: fewest padded channels among the wide tiles (narrow tiles re-read the
// activation patch once per tile and run too few MFMAs per barrier), narrow tiles only for narrow layers
inline int pick_mi3(int Cout) {
  if (Cout <= 16) return 1;
  if (Cout <= 32) return 2;
  int best = 3, best_pad = 1 << 30;
  for (int mi = 5; mi >= 3; --mi) {
    int mt = 16 * mi, pad = (Cout + mt - 1) / mt * mt;
    if (pad < best_pad) {
      best_pad = pad;
      best = mi;
    }
  }
  // many-channel layers are the low-resolution ones (few pixel tiles): 64-channel tiles give a workgroup count that fills
  // the chip in one round where 48 needs a second, nearly empty one and 80 leaves a third of the slots idle (dec3
  // 64->848 at 40x80: forward 0.093 -> 0.075 ms, weight gradient 0.126 -> 0.108 ms), at <= 6 % more padding
  if (Cout > 256 && (Cout + 63) / 64 * 64 <= best_pad + best_pad * 6 / 100) return 4;
  return best;
}
inline int nst_of(int k) { return k == 5 ? nq_conv3_nst_k5() : nq_conv3_nst_k3(); }
inline int nstk_of(int k, int kind) { return k == 5 ? nq_conv3_nstk_k5(kind) : nq_conv3_nstk_k3(kind); }
// shape of the last 16-channel chunk by the channels r it really holds (Conv3Args::tail): 1: r <= 4, 2: r <= 8, 3: r <= 12,
// 0: a full chunk -- it is laid out (and run) with fewer, denser k-steps
inline int tail_kind_of(int Cin, int mi) {
  const int r = Cin - CC * ((Cin + CC - 1) / CC - 1);
  return r <= 4 ? 1 : (r <= 8 ? 2 : (r <= 12 ? 3 : 0));
}
inline int64_t total_steps3(int Cin, int k, int mi) {
  const int nchunk = (Cin + CC - 1) / CC;
  return (int64_t)(nchunk - 1) * nst_of(k) + nstk_of(k, tail_kind_of(Cin, mi));
}

// Operand of the bf16x3 kernels: 16-byte fragment slots [global k-step][plane hi/lo][co tile][kq][MT] of 8 bf16 k-values.
// transposed = 0: logical conv == the stored conv, src(co, ch, tap) = w[co][ch][tap]
// transposed = 1: data gradient, logical (Cin=Cout_w -> Cout=Cin_w): src(co, ch, tap) = w[ch][co][KK-1-tap]
struct WL3 {
  const float* w;
  uint4* out;
  int Cin, Cout, KK, NST, NSTT, nchunk, tail, co_tiles, MT, transposed;   // tail: kind of the last chunk, NSTT its k-steps
  int co16;        // 16-channel groups of output channels: ceil(co_tiles*MT / 16)
  int64_t slots;
};
// (k-value e of lane group kq at k-step s -> (channel, tap): wl3_elem, conv3_layout.h)

// One workgroup per (16-channel chunk c of the logical input channels, group of 16 logical output channels): the 16 x 16 x KK
// weights it needs are 16 CONTIGUOUS runs of 16*KK floats in the stored tensor (rows = co for the forward operand, = ch
// for the transposed one), loaded coalesced into LDS; the fragment slots are then assembled from LDS and stored 16 bytes
// per thread, 256 contiguous bytes per 16 lanes.  (Round 2's kernel gathered the 8 values of a slot straight from global
// memory at a stride of KK or Cout*KK floats: 22.6 us for the layers of HNeRV-3M, most of it uncoalesced reads.)
constexpr int WL3_KKMAX = 25;
// (KK is a template parameter: the index arithmetic -- e / (16*KK), rem / KK per loaded float, eight (channel, tap) pairs per
// slot -- was most of the kernel's time with a run-time divisor)
template <int KK>
__device__ __forceinline__ void wl3_tile(const WL3& p, int c, int g16, float* __restrict__ T /* [16][16*KK (+1)] */) {
  constexpr int RS = 16 * KK + 1;   // row stride: odd -> the 16 rows of a column land in different banks
  const int tid = threadIdx.x;
  const int nfull = p.nchunk - (p.tail ? 1 : 0);
  const int kind = (c == nfull) ? p.tail : 0;
  const int nst = kind ? p.NSTT : p.NST;
  const int64_t gs0 = (int64_t)c * p.NST;   // every chunk in front of c is a full one
  // rows a = 0..15, columns (b, tap): forward a = co (g16*16 + a), b = ch (c*16 + b); transposed a = ch, b = co
  const int Ra = p.transposed ? p.Cin : p.Cout;            // valid extent of the row index (stored tensor's dim 0)
  const int Cb = p.transposed ? p.Cout : p.Cin;            // stored tensor's dim 1
  const int a0 = p.transposed ? c * 16 : g16 * 16, b0 = p.transposed ? g16 * 16 : c * 16;
  // all KK loads of a thread in flight together (branch-free: an out-of-range element reads element 0 and is zeroed
  // afterwards) -- with a conditional load the compiler waited for each group of five before issuing the next
  {
    float v[KK];
#pragma unroll
    for (int j = 0; j < KK; ++j) {
      const int e = tid + j * 256;
      const int a = e / (16 * KK), rem = e - a * (16 * KK);
      const int bb = rem / KK;
      const bool ok = a0 + a < Ra && b0 + bb < Cb;
      const float t = p.w[ok ? ((int64_t)(a0 + a) * Cb + b0) * KK + rem : 0];
      v[j] = ok ? t : 0.f;
    }
#pragma unroll
    for (int j = 0; j < KK; ++j) {
      const int e = tid + j * 256;
      const int a = e / (16 * KK), rem = e - a * (16 * KK);
      T[a * RS + rem] = v[j];
    }
  }
  __syncthreads();
  const int MT = p.MT;
  const int64_t plane_stride = (int64_t)p.co_tiles * 4 * MT;
  for (int sl = tid; sl < nst * 4 * 16; sl += 256) {
    const int col = sl & 15, kq = (sl >> 4) & 3, st = sl >> 6;
    const int co = g16 * 16 + col;
    if (co >= p.co_tiles * MT) continue;
    unsigned hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        int ch, tap;
        wl3_elem(kind, KK, st, kq, 2 * j + t, ch, tap);
        float x = 0.f;
        if (tap < KK) x = p.transposed ? T[ch * RS + col * KK + (KK - 1 - tap)] : T[col * RS + ch * KK + tap];
        v[t] = x;
      }
      __bf16 h0 = (__bf16)v[0], h1 = (__bf16)v[1];
      __bf16 l0 = (__bf16)(v[0] - (float)h0), l1 = (__bf16)(v[1] - (float)h1);
      hi[j] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
      lo[j] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
    }
    const int tile = co / MT, co_l = co - tile * MT;
    const int64_t base = (gs0 + st) * 2 * plane_stride + (int64_t)tile * 4 * MT + kq * MT + co_l;
    p.out[base] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    p.out[base + plane_stride] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

__global__ __launch_bounds__(256) void weight_layout3_kernel(WL3 p) {
  __shared__ float T[16 * (16 * WL3_KKMAX + 1)];
  if (p.KK == 25) wl3_tile<25>(p, (int)blockIdx.x / p.co16, (int)blockIdx.x % p.co16, T);
  else wl3_tile<9>(p, (int)blockIdx.x / p.co16, (int)blockIdx.x % p.co16, T);
}

// all layers' operands (forward and data-gradient) in ONE launch: the table travels as a kernel argument
constexpr int WL3_MAXSEG = 16;
struct WL3Multi {
  WL3 s[WL3_MAXSEG];
  int blk0[WL3_MAXSEG + 1];
  int nseg;
};
__global__ __launch_bounds__(256) void weight_layout3_multi_kernel(WL3Multi t) {
  __shared__ float T[16 * (16 * WL3_KKMAX + 1)];
  int k = 0;
  while (k + 1 < t.nseg && (int)blockIdx.x >= t.blk0[k + 1]) ++k;
  const int blk = (int)blockIdx.x - t.blk0[k];
  if (t.s[k].KK == 25) wl3_tile<25>(t.s[k], blk / t.s[k].co16, blk % t.s[k].co16, T);
  else wl3_tile<9>(t.s[k], blk / t.s[k].co16, blk % t.s[k].co16, T);
}

// Round 4: the fp32 operands of the layers that stay on the fp32 / vector kernels (the 1 x 1 stem, the head) in the SAME launch
// as the bf16x3 operands: blocks past the bf16x3 tiles run the element code of weight_layouts_multi_kernel (conv.hip) -- the
// same values, one launch (~5 us of launch floor) less per iteration.
struct WLF {
  const float* w;
  float* wt;
  int Cout, Cin, KK, krows, ld, bwd;
};
struct WLAll {
  WL3 s3[WL3_MAXSEG];
  int blk3[WL3_MAXSEG + 1];
  int n3;
  WLF sf[WL3_MAXSEG];
  int blkf[WL3_MAXSEG + 1];
  int nf;
};
__global__ __launch_bounds__(256) void weight_layouts_all_kernel(WLAll t) {
  __shared__ float T[16 * (16 * WL3_KKMAX + 1)];
  const int nb3 = t.blk3[t.n3];
  if ((int)blockIdx.x < nb3) {
    int k = 0;
    while (k + 1 < t.n3 && (int)blockIdx.x >= t.blk3[k + 1]) ++k;
    const int blk = (int)blockIdx.x - t.blk3[k];
    if (t.s3[k].KK == 25) wl3_tile<25>(t.s3[k], blk / t.s3[k].co16, blk % t.s3[k].co16, T);
    else wl3_tile<9>(t.s3[k], blk / t.s3[k].co16, blk % t.s3[k].co16, T);
    return;
  }
  const int bf = (int)blockIdx.x - nb3;
  int k = 0;
  while (k + 1 < t.nf && bf >= t.blkf[k + 1]) ++k;
  const WLF& g = t.sf[k];
  const int64_t i = (int64_t)(bf - t.blkf[k]) * 256 + threadIdx.x;
  if (i >= (int64_t)g.krows * g.ld) return;
  const int row = (int)(i / g.ld), col = (int)(i - (int64_t)row * g.ld);
  float v = 0.f;
  if (g.bwd) {   // wt[(co*KK + tap')][ci] = w[co][ci][KK-1-tap']
    const int co = row / g.KK, tap = row - co * g.KK;
    if (co < g.Cout && col < g.Cin) v = g.w[((int64_t)co * g.Cin + col) * g.KK + (g.KK - 1 - tap)];
  } else {       // wt[k][co] = w[co][k]
    if (col < g.Cout && row < g.Cin * g.KK) v = g.w[(int64_t)col * (g.Cin * g.KK) + row];
  }
  g.wt[i] = v;
}

// split-K of the forward / data-gradient kernel over 16-channel chunks when the pixel x channel grid alone cannot fill
// the chip (deep, low-resolution layers): ~512 workgroups, every split non-empty
struct Fwd3Plan {
  int mi, nsplit, per;
  int64_t wgs;
};
inline Fwd3Plan plan_fwd3(int B, int Cin, int H, int W, int Cout) {
  Fwd3Plan p;
  p.mi = pick_mi3(Cout);
  p.wgs = (int64_t)((W + 31) / 32) * ((H + 7) / 8) * ((Cout + 16 * p.mi - 1) / (16 * p.mi)) * B;
  const int nchunk = (Cin + CC - 1) / CC;
  p.nsplit = 1;
  p.per = nchunk;
  // (a K loop of <= 4 chunks is not worth splitting: the slabs and the finish launch cost more than the half-empty round --
  // NeRV's 36 -> 384 layer at 40 x 80: 180 workgroups, 20.0 + 23.2 us split in two vs one launch without slabs)
  static const int min_chunks = [] { const char* e = getenv("NQ_SPLITK_MIN_CHUNKS"); return e ? atoi(e) : 5; }();
  if (p.wgs < 256 && nchunk >= min_chunks) {
    int want = (int)(512 / p.wgs);   // floor: stay within ONE round of 512 resident workgroups (a second, nearly empty
                                     // round costs as much as the first)
    if (want > nchunk) want = nchunk;
    p.per = (nchunk + want - 1) / want;
    p.nsplit = (nchunk + p.per - 1) / p.per;
  }
  return p;
}

// Few-pixel layers (conv_flat3.hip): taken when the flat kernel needs no slabs (forward convolutions of the deep layers: no
// finish launch), or when the tiled kernel's grid would not fill the chip even with split-K (it would fall back to the fp32
// kernels).  Long K loops that the tiled split-K path already handles stay there: measured (us, kernel + finish) HNeRV dec2
// data gradient 21.0 flat vs 21.7 tiled, NeRV dec1 data gradient 22.4 vs 18.7, NeRV dec2 data gradient 19.3 vs 25.6 (fp32).
inline bool use_flat3(int B, int Cin, int H, int W, int Cout, int k, int* fns) {
  int ns = 1;
  if (!nq_conv_flat3_plan(B, Cin, H, W, Cout, k, nullptr, nullptr, &ns, nullptr)) return false;
  if (fns) *fns = ns;
  if (ns == 1) return true;
  const Fwd3Plan p = plan_fwd3(B, Cin, H, W, Cout);
  return p.wgs * p.nsplit < 128;
}

struct Wg3Plan {
  int mi, ni, co_pad, n_pad, nsplit, pc;
};
// NQ_WGRAD3_PC=0 keeps every layer on the 4-wave kernel (A/B runs of the two structures in one build)
inline bool wgrad3_ss1() {   // NQ_WGRAD3_SS=1: 32-pixel segments only (A/B runs)
  const char* e = std::getenv("NQ_WGRAD3_SS");
  return e && e[0] == '1';
}
inline bool wgrad3_pc_enabled() {   // read per call: tools/bench_kernels.py flips it between launches of one process
  const char* e = std::getenv("NQ_WGRAD3_PC");
  return !(e && e[0] == '0');
}
inline Wg3Plan plan_wgrad3(int B, int Cin, int H, int W, int Cout, int k) {
  Wg3Plan p;
  p.mi = pick_mi3(Cout);
  const int N = Cin * k * k;
  if (N <= 64) {
    p.ni = 1;
  } else {   // n-tile of 64*ni columns: the ni in {5,6,7} that pads C_in*k*k least (7 only with <= 4 channel blocks);
             // ties go to the wider tile (fewer tiles re-staging dY)
    int best = 6, best_pad = 1 << 30;
    for (int ni = 5; ni <= (p.mi <= 4 ? 7 : 6); ++ni) {
      const int nt = 64 * ni, pad = (N + nt - 1) / nt * nt;
      if (pad <= best_pad) {
        best_pad = pad;
        best = ni;
      }
    }
    p.ni = best;
  }
  int mt = 16 * p.mi, nt = 64 * p.ni;
  p.co_pad = (Cout + mt - 1) / mt * mt;
  p.n_pad = (N + nt - 1) / nt * nt;
  int tiles = (p.co_pad / mt) * (p.n_pad / nt);
  // Very wide layers (more than 256 output tiles: the UVG-12M shape's 128 -> 1712) fell off the producer/consumer kernel, which
  // wants one workgroup per CU in ONE round: the 80-channel tile gets them back under 256 when its extra padding is small
  // (preferring them wherever they pad no more -- HNeRV-3M's 64 -> 848: 880 instead of 896 rows -- measured 66-68 -> 65 us with one
  // more slab to reduce: not kept)
  if (tiles > 256 && p.mi < 5 && p.ni >= 5 && p.ni <= 6 /* (no 80 x 448 tile) */) {
    const int co5 = (Cout + 79) / 80 * 80, tiles5 = (co5 / 80) * (p.n_pad / nt);
    if (tiles5 <= 256 && co5 * 100 <= p.co_pad * 105) {
      p.mi = 5;
      mt = 80;
      p.co_pad = co5;
      tiles = tiles5;
    }
  }
  const int nseg = ((W + 31) / 32) * H * B;
  int ns = 512 / tiles;  // one full wave of workgroups (2 resident per CU)
  if (ns > nseg) ns = nseg;
  if (ns < 1) ns = 1;
  p.nsplit = ns;
  p.pc = 0;
  // the producer/consumer kernel addresses x through SIGNED and dy through unsigned 32-bit byte offsets (raw buffer loads):
  // operands of 2 GiB and more stay on the 4-wave kernel (64-bit pointers)
  const bool pc_ok = wgrad3_pc_enabled() && (int64_t)B * Cin * H * W * 4 < (1ll << 31) && (int64_t)B * Cout * H * W * 4 < (1ll << 31);
  // producer/consumer kernel (8 waves, ONE workgroup per CU): wide tiles with a long K loop per workgroup
  if (pc_ok && p.ni >= 5 && p.mi >= 3 && tiles <= 256) {
    // rows per staged segment: 4 (2) where H divides and every workgroup still gets >= 16 segments; NQ_WGRAD3_SS=1 -> 1
    int sy = 1;
    int ns_pc = 256 / tiles;
    const int segs_x = (W + 31) / 32;
    // segments per workgroup a taller segment must leave (NQ_WGRAD3_MINSEG; 16 until late round 4: NeRV-3M's 24 -> 96 at
    // 160 x 320 stayed on one-row segments for it, 44.2 us; 12 -> two rows 42.3; 6 -> four rows 41.3; HNeRV dec3 84.0 -> 80.9)
    static const int minseg = [] { const char* e = std::getenv("NQ_WGRAD3_MINSEG"); return e ? atoi(e) : 6; }();
    if (!wgrad3_ss1() && ns_pc >= 1) {
      if (H % 4 == 0 && segs_x * (H / 4) * B / ns_pc >= minseg) sy = 4;
      else if (H % 2 == 0 && segs_x * (H / 2) * B / ns_pc >= minseg) sy = 2;
    }
    const int nseg_pc = segs_x * (H / sy) * B;
    const int need = minseg < 16 ? minseg : 16;
    // a little short of segments for one workgroup per CU (NeRV-3M's 36 -> 384 at 40 x 80: 240 one-row segments for 42 splits):
    // fewer splits, as long as three quarters of the CUs still get a workgroup (NQ_WGRAD3_SHRINK=0: the 4-wave kernel as before)
    static const int shrink = [] { const char* e = std::getenv("NQ_WGRAD3_SHRINK"); return e ? atoi(e) : 1; }();
    if (shrink && ns_pc >= 1 && nseg_pc / ns_pc < need && (nseg_pc / need) * tiles >= 192) ns_pc = nseg_pc / need;
    if (ns_pc >= 1 && nseg_pc / ns_pc >= need) {
      p.pc = sy == 1 ? 1 : 10 + sy;
      p.nsplit = ns_pc;
    }
  }
  // narrow problems streaming a big tensor (the role-swapped head gradient: 3 x 9 n-values, K = all pixels): the 4-wave
  // kernel runs 18 MFMAs per barrier with one 6 KB segment in flight per workgroup (latency-bound, 2.1 TB/s); the
  // producer/consumer kernel with 128-pixel segments keeps 2 x 24 KB in flight per CU
  if (pc_ok && p.ni == 1 && p.mi >= 2 && p.mi <= 5 && tiles == 1 && W % 128 == 0) {
    const int nseg_pc = (W / 128) * H * B;
    if (nseg_pc / 256 >= 8) {
      p.pc = 4;
      // NQ_WGRAD3_HEAD_SPLITS (timing runs): 256 = one 8-wave workgroup per CU (round 2), 512 = two
      static const int hs = [] { const char* e = std::getenv("NQ_WGRAD3_HEAD_SPLITS"); return e ? atoi(e) : 512; }();
      p.nsplit = (p.mi <= 3 && nseg_pc / hs >= 8) ? hs : 256;   // (64- / 80-channel tiles: one workgroup per CU)
    }
  }
  return p;
}

// Fixed-order reduction of the split slabs: a workgroup owns 256/SG consecutive outputs, split group g adds slabs
// g, g+SG, g+2SG, ... in order, and the SG partial sums are combined in LDS in index order -> run-to-run identical.
// swap_kk > 0: the slabs hold R[co'][ci'][tap] of the ROLE-SWAPPED problem (x and dy exchanged); the result is written as
// dW[ci'][co'][KK-1-tap], KK = swap_kk -- the weight gradient of the original convolution (ops.conv_wgrad_swapped3).
template <int SG>
__global__ __launch_bounds__(256) void wgrad3_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ slab_db,
                                                            float* __restrict__ dw, float* __restrict__ db, int Cout, int N,
                                                            int co_pad, int n_pad, int nsplit, int swap_kk) {
  constexpr int OG = 256 / SG;
  __shared__ float part[SG][OG];
  const int o = threadIdx.x % OG, g = threadIdx.x / OG;
  const int64_t i = (int64_t)blockIdx.x * OG + o;
  const int64_t total = (int64_t)Cout * N;
  float s = 0.f;
  if (i < total) {
    const int co = (int)(i / N), n = (int)(i - (int64_t)co * N);
    const float* p = slab + (int64_t)co * n_pad + n;
    const int64_t stride = (int64_t)co_pad * n_pad;
#pragma unroll 4
    for (int k = g; k < nsplit; k += SG) s += p[k * stride];
  } else if (db && i < total + Cout) {
    const int co = (int)(i - total);
#pragma unroll 4
    for (int k = g; k < nsplit; k += SG) s += slab_db[(int64_t)k * co_pad + co];
  }
  part[g][o] = s;
  __syncthreads();
  if (g == 0) {
    float t = part[0][o];
#pragma unroll
    for (int j = 1; j < SG; ++j) t += part[j][o];
    if (i < total) {
      if (swap_kk > 0) {
        const int co = (int)(i / N), n = (int)(i - (int64_t)co * N);
        const int ci = n / swap_kk, tap = n - ci * swap_kk;
        dw[((int64_t)ci * Cout + co) * swap_kk + (swap_kk - 1 - tap)] = t;
      } else {
        dw[i] = t;
      }
    } else if (db && i < total + Cout) {
      db[(int)(i - total)] = t;
    }
  }
}

// Several pending reductions in ONE launch (nq_wgrad_reduce_multi): per segment the arithmetic of wgrad3_reduce_kernel<sg>
// (sg = 4 / 16) or of the sequential fp32 reduction (sg = 1), so results are bit-identical to the single-tensor launches.
constexpr int WGR_MAXSEG = 16;
struct WGRMulti {
  nq_wgr_seg s[WGR_MAXSEG];
  int blk0[WGR_MAXSEG + 1];
  int nseg;
};
// (round 3: a thread owns FOUR consecutive n of one channel -- one 16-byte load per slab instead of four 4-byte ones; the
// launch read 115 MB of slabs at 2.8 TB/s with scalar loads.  Per output the slabs are still added in the order
// j = g, g + SG, ... and the SG partial sums in the order 0 .. SG-1: bit-identical to the scalar form.)
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(WGRMulti t) {
  __shared__ float4 part[256];
  int k = 0;
  while (k + 1 < t.nseg && (int)blockIdx.x >= t.blk0[k + 1]) ++k;
  const nq_wgr_seg& q = t.s[k];
  const int SG = q.sg, OG = 256 / SG;
  const int o = threadIdx.x % OG, g = threadIdx.x / OG;
  const int N = q.N, Cout = q.Cout, nsplit = q.nsplit;
  const int NQ = (N + 3) >> 2;                                   // quads per channel (n_pad is a multiple of 16)
  const int64_t i = (int64_t)((int)blockIdx.x - t.blk0[k]) * OG + o;   // quad index, then the bias entries
  const int64_t total = (int64_t)Cout * NQ;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  int co = 0, n = 0;
  if (i < total) {
    co = (int)(i / NQ);
    n = 4 * (int)(i - (int64_t)co * NQ);
    const float* p = q.slab + (int64_t)co * q.n_pad + n;
    const int64_t stride = (int64_t)q.co_pad * q.n_pad;
#pragma unroll 8
    for (int j = g; j < nsplit; j += SG) {
      const float4 v = *reinterpret_cast<const float4*>(p + j * stride);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  } else if (q.db && i < total + Cout) {
    const int cb = (int)(i - total);
#pragma unroll 8
    for (int j = g; j < nsplit; j += SG) s.x += q.slab_db[(int64_t)j * q.co_pad + cb];
  }
  if (SG > 1) {   // block-uniform
    part[g * OG + o] = s;
    __syncthreads();
    if (g != 0) return;
    s = part[o];
    for (int j = 1; j < SG; ++j) {
      const float4 v = part[j * OG + o];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  if (i < total) {
    const float sv[4] = {s.x, s.y, s.z, s.w};
    if (q.swap_kk > 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e >= N) break;
        const int ci = (n + e) / q.swap_kk, tap = (n + e) - ci * q.swap_kk;
        q.dw[((int64_t)ci * Cout + co) * q.swap_kk + (q.swap_kk - 1 - tap)] = sv[e];
      }
    } else if ((N & 3) == 0) {
      *reinterpret_cast<float4*>(q.dw + (int64_t)co * N + n) = s;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (n + e < N) q.dw[(int64_t)co * N + n + e] = sv[e];
    }
  } else if (q.db && i < total + Cout) {
    q.db[(int)(i - total)] = s.x;
  }
}

}  // namespace

extern "C" {

int nq_conv3_supported(int B, int Cin, int H, int W, int Cout, int k) {
  // <= 4 channels on either side (the image head): the streaming VALU kernels of conv_head.hip are faster than a
  // >80 %-padded MFMA tile (measured 0.17 / 0.23 ms vs 0.21 / 0.38 ms, forward / data gradient of 37->3 at 640x1280)
  if (!(k == 3 || k == 5) || B <= 0 || Cin <= 0 || Cout <= 4 || Cin <= 4) return 0;
  // the kernel addresses the input with 32-bit buffer offsets (conv_igemm3_impl.h): tensors of 4 GiB and more stay on
  // the fp32 kernels
  if ((int64_t)B * Cin * H * W * 4 >= 0xFFFFFF00ll) return 0;
  // few-pixel kernel (conv_flat3.hip) -- offered for layers whose weight tensor is worth streaming; a toy convolution (a few
  // hundred weights) stays on the exact-fp32 kernels: nothing to gain, and its callers keep fp32-level results
  if (use_flat3(B, Cin, H, W, Cout, k, nullptr) && (int64_t)Cin * k * k * Cout >= 65536) return 1;
  const Fwd3Plan p = plan_fwd3(B, Cin, H, W, Cout);
  return p.wgs * p.nsplit >= 128;  // still smaller grids stay on the fp32 split-K kernel
}

int64_t nq_conv_forward3_ws_floats(int B, int Cin, int H, int W, int Cout, int k) {
  if (!(k == 3 || k == 5) || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  int fns = 1;
  if (use_flat3(B, Cin, H, W, Cout, k, &fns)) return fns > 1 ? (int64_t)fns * B * Cout * H * W : 0;
  const Fwd3Plan p = plan_fwd3(B, Cin, H, W, Cout);
  return p.nsplit > 1 ? (int64_t)p.nsplit * B * Cout * H * W : 0;
}

int64_t nq_conv3_weight_bytes(int Cin, int Cout, int k) {
  if (!(k == 3 || k == 5)) return 0;
  const int mi = pick_mi3(Cout), mt = 16 * mi;
  const int64_t co_tiles = (Cout + mt - 1) / mt, nchunk = (Cin + CC - 1) / CC;
  (void)nchunk;
  return total_steps3(Cin, k, mi) * 2 * co_tiles * 4 * mt * 16;
}

// w: OIHW weight tensor of the STORED conv (Cout_w, Cin_w, k, k).  transposed = 0 -> operand of the forward conv
// (Cin = Cin_w, Cout = Cout_w); transposed = 1 -> operand of its data gradient (Cin = Cout_w, Cout = Cin_w).
static bool wl3_fill(WL3& p, const float* w, void* wt3, int Cin, int Cout, int k, int transposed) {
  if (!w || !wt3 || !(k == 3 || k == 5) || Cin <= 0 || Cout <= 0) return false;
  const int mi = pick_mi3(Cout);
  p.w = w; p.out = reinterpret_cast<uint4*>(wt3);
  p.Cin = Cin; p.Cout = Cout; p.KK = k * k; p.NST = nst_of(k); p.tail = tail_kind_of(Cin, mi); p.NSTT = nstk_of(k, p.tail);
  p.MT = 16 * mi; p.co_tiles = (Cout + p.MT - 1) / p.MT; p.nchunk = (Cin + CC - 1) / CC; p.transposed = transposed;
  p.slots = total_steps3(Cin, k, mi) * p.co_tiles * 4 * p.MT;
  p.co16 = (p.co_tiles * p.MT + 15) / 16;
  return true;
}

int nq_weight_layout3(const float* w, void* wt3, int Cin, int Cout, int k, int transposed, nq_stream_t stream) {
  WL3 p;
  if (!wl3_fill(p, w, wt3, Cin, Cout, k, transposed)) return NQ_ERR_INVALID;
  hipLaunchKernelGGL(weight_layout3_kernel, dim3((unsigned)(p.nchunk * p.co16)), dim3(256), 0, nq_s(stream), p);
  return nq_launch_status();
}

int nq_weight_layout3_multi(const nq_wl3_seg* segs, int nseg, nq_stream_t stream) {
  if (!segs || nseg <= 0) return NQ_ERR_INVALID;
  for (int base = 0; base < nseg; base += WL3_MAXSEG) {
    WL3Multi t;
    t.nseg = (nseg - base < WL3_MAXSEG) ? nseg - base : WL3_MAXSEG;
    int blocks = 0;
    for (int i = 0; i < t.nseg; ++i) {
      const nq_wl3_seg& h = segs[base + i];
      if (!wl3_fill(t.s[i], h.w, h.wt3, h.Cin, h.Cout, h.k, h.transposed)) return NQ_ERR_INVALID;
      t.blk0[i] = blocks;
      blocks += t.s[i].nchunk * t.s[i].co16;
    }
    t.blk0[t.nseg] = blocks;
    hipLaunchKernelGGL(weight_layout3_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, nq_s(stream), t);
  }
  return nq_launch_status();
}

int nq_weight_layouts_all(const nq_wl3_seg* segs3, int n3, const nq_wl_seg* segsf, int nf, nq_stream_t stream) {
  if ((n3 > 0 && !segs3) || (nf > 0 && !segsf) || n3 < 0 || nf < 0 || n3 + nf == 0) return NQ_ERR_INVALID;
  // count the fp32 operands (one per direction)
  int nfo = 0;
  for (int i = 0; i < nf; ++i) nfo += (segsf[i].wt_fwd ? 1 : 0) + (segsf[i].wt_bwd ? 1 : 0);
  if (n3 > WL3_MAXSEG || nfo > WL3_MAXSEG || n3 == 0 || nfo == 0) {   // does not fit one argument block: the two launches
    if (n3 > 0)
      if (int rc = nq_weight_layout3_multi(segs3, n3, stream)) return rc;
    if (nf > 0) return nq_weight_layouts_multi(segsf, nf, stream);
    return NQ_OK;
  }
  WLAll t;
  t.n3 = n3;
  int blocks = 0;
  for (int i = 0; i < n3; ++i) {
    const nq_wl3_seg& h = segs3[i];
    if (!wl3_fill(t.s3[i], h.w, h.wt3, h.Cin, h.Cout, h.k, h.transposed)) return NQ_ERR_INVALID;
    t.blk3[i] = blocks;
    blocks += t.s3[i].nchunk * t.s3[i].co16;
  }
  t.blk3[n3] = blocks;
  t.nf = 0;
  int bf = 0;
  for (int i = 0; i < nf; ++i) {
    const nq_wl_seg& h = segsf[i];
    if (!h.w || h.Cout <= 0 || h.Cin <= 0 || h.k <= 0) return NQ_ERR_INVALID;
    const int KK = h.k * h.k;
    for (int bwd = 0; bwd < 2; ++bwd) {
      float* wt = bwd ? h.wt_bwd : h.wt_fwd;
      if (!wt) continue;
      const int krows = bwd ? h.krows_bwd : h.krows_fwd, ld = bwd ? h.ld_bwd : h.ld_fwd;
      if (bwd ? (krows < h.Cout * KK || ld < h.Cin) : (krows < h.Cin * KK || ld < h.Cout)) return NQ_ERR_INVALID;
      t.sf[t.nf] = WLF{h.w, wt, h.Cout, h.Cin, KK, krows, ld, bwd};
      t.blkf[t.nf] = bf;
      bf += (int)(((int64_t)krows * ld + 255) / 256);
      ++t.nf;
    }
  }
  t.blkf[t.nf] = bf;
  hipLaunchKernelGGL(weight_layouts_all_kernel, dim3((unsigned)(blocks + bf)), dim3(256), 0, nq_s(stream), t);
  return nq_launch_status();
}

// Which sides of nq_conv_forward3 may travel as split {hi | lo} words for this shape: bit NQ_EPI_X_SPLIT -- the input (the tiled
// kernel with >= 32-channel tiles stages it), bit NQ_EPI_Y_SPLIT -- the output y (the tiled kernel's own epilogue writes it: no
// split-K; the few-pixel kernel when it leaves no slabs).  0 for shapes this path does not serve.
int nq_conv3_split_io(int B, int Cin, int H, int W, int Cout, int k) {
  if (!nq_conv3_supported(B, Cin, H, W, Cout, k)) return 0;
  {   // few-pixel kernel: reads floats only; writes the words when it needs no slabs (its own epilogue runs)
    int fns = 1;
    if (use_flat3(B, Cin, H, W, Cout, k, &fns)) return fns == 1 ? NQ_EPI_Y_SPLIT : 0;
  }
  const Fwd3Plan p = plan_fwd3(B, Cin, H, W, Cout);
  return (p.mi >= 2 ? NQ_EPI_X_SPLIT : 0) | (p.nsplit == 1 ? NQ_EPI_Y_SPLIT : 0);
}

int nq_conv_forward3(const float* x, const void* wt3, const float* bias, float* y, float* z, const float* zprev, float* ws, int B,
                     int Cin, int H, int W, int Cout, int k, int r, int epilogue, nq_stream_t stream) {
  if (!x || !wt3 || (!y && epilogue != NQ_EPI_PS) || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return NQ_ERR_INVALID;
  if (!(k == 3 || k == 5)) return NQ_ERR_UNSUPPORTED;
  // split {hi | lo} word interchange (include/nq_hip.h: NQ_EPI_X_SPLIT / NQ_EPI_Y_SPLIT): only where nq_conv3_split_io says so
  const int fmt = epilogue & (NQ_EPI_X_SPLIT | NQ_EPI_Y_SPLIT);
  epilogue &= ~(NQ_EPI_X_SPLIT | NQ_EPI_Y_SPLIT);
  if (fmt & ~nq_conv3_split_io(B, Cin, H, W, Cout, k)) return NQ_ERR_UNSUPPORTED;
  if (epilogue < 0 || epilogue > NQ_EPI_DGRAD_GELU) return NQ_ERR_INVALID;
  if ((epilogue == NQ_EPI_PS_GELU || epilogue == NQ_EPI_PS) && (!z || r <= 0 || Cout % (r * r) != 0)) return NQ_ERR_INVALID;
  if (epilogue == NQ_EPI_DGRAD_GELU && (!zprev || r <= 0 || H % r != 0 || W % r != 0)) return NQ_ERR_INVALID;
  if (B > 65535) return NQ_ERR_UNSUPPORTED;
  hipStream_t st = nq_s(stream);
  {   // few-pixel layers: pixels of all frames as one flat GEMM dimension, waves split the K loop (conv_flat3.hip)
    int fns = 1;
    if (use_flat3(B, Cin, H, W, Cout, k, &fns)) {
      if (fns > 1 && !ws) return NQ_ERR_INVALID;
      const int mi = pick_mi3(Cout), kind = tail_kind_of(Cin, mi);
      int rc = nq_conv_flat3(x, wt3, bias, y, z, zprev, ws, B, Cin, H, W, Cout, k, r, epilogue | fmt, 16 * mi, kind, nst_of(k),
                             nstk_of(k, kind), st);
      if (rc != NQ_OK || fns == 1) return rc;
      return nq_conv_splitk_finish(ws, bias, y, z, zprev, B, H, W, Cout, r, epilogue, fns, st);
    }
  }
  const Fwd3Plan p = plan_fwd3(B, Cin, H, W, Cout);
  if (p.nsplit > 1 && !ws) return NQ_ERR_INVALID;
  int rc = (k == 3) ? nq_conv_igemm3_k3(x, wt3, bias, y, z, zprev, B, Cin, H, W, Cout, r, epilogue | fmt, p.mi, p.nsplit, p.per, ws, st)
                    : nq_conv_igemm3_k5(x, wt3, bias, y, z, zprev, B, Cin, H, W, Cout, r, epilogue | fmt, p.mi, p.nsplit, p.per, ws, st);
  if (rc != NQ_OK || p.nsplit == 1) return rc;
  return nq_conv_splitk_finish(ws, bias, y, z, zprev, B, H, W, Cout, r, epilogue, p.nsplit, st);
}

int nq_conv_wgrad3_supported(int B, int Cin, int H, int W, int Cout, int k) {
  if (!(k == 3 || k == 5) || B <= 0 || Cin <= 0 || Cout <= 4) return 0;
  if (nq_conv_wgrad_flat3_ok(B, Cin, H, W, Cout, k)) return 1;   // few-pixel layers: conv_wgrad_flat3.hip (no slabs)
  return (int64_t)((W + 31) / 32) * H * B >= 128;  // enough 32-pixel segments to split over
}

int nq_conv_wgrad3_plan(int B, int Cin, int H, int W, int Cout, int k, int* mi, int* ni, int* nsplit, int* pc) {
  if (!(k == 3 || k == 5) || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0 || !mi || !ni || !nsplit || !pc) return NQ_ERR_INVALID;
  const Wg3Plan p = plan_wgrad3(B, Cin, H, W, Cout, k);
  *mi = p.mi; *ni = p.ni; *nsplit = p.nsplit; *pc = p.pc;
  return NQ_OK;
}

int64_t nq_conv_wgrad3_ws_floats(int B, int Cin, int H, int W, int Cout, int k) {
  if (!(k == 3 || k == 5) || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  if (nq_conv_wgrad_flat3_ok(B, Cin, H, W, Cout, k)) return 4;   // no slabs (a token workspace keeps the pointer non-NULL)
  Wg3Plan p = plan_wgrad3(B, Cin, H, W, Cout, k);
  return (int64_t)p.nsplit * p.co_pad * ((int64_t)p.n_pad + 1);
}

static int conv_wgrad3_impl(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W,
                            int Cout, int k, int swap_kk, nq_wgr_seg* seg, nq_stream_t stream, int fmt = 0);

int nq_conv_wgrad3(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W, int Cout,
                   int k, nq_stream_t stream) {
  return conv_wgrad3_impl(x, dy, dw, db, ws, B, Cin, H, W, Cout, k, 0, nullptr, stream);
}

// Operands as split {hi | lo} words (include/nq_hip.h): fmt bit 0 -- x, bit 1 -- dy; where nq_conv_wgrad3_split_io says so
int nq_conv_wgrad3_split_io(int B, int Cin, int H, int W, int Cout, int k) {
  if (!nq_conv_wgrad3_supported(B, Cin, H, W, Cout, k) || nq_conv_wgrad_flat3_ok(B, Cin, H, W, Cout, k)) return 0;
  const Wg3Plan p = plan_wgrad3(B, Cin, H, W, Cout, k);
  return (p.pc == 1 || p.pc == 12 || p.pc == 14) ? 3 : 0;
}
int nq_conv_wgrad3_fmt(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W, int Cout,
                       int k, int fmt, nq_stream_t stream) {
  if (fmt & ~nq_conv_wgrad3_split_io(B, Cin, H, W, Cout, k)) return NQ_ERR_UNSUPPORTED;
  return conv_wgrad3_impl(x, dy, dw, db, ws, B, Cin, H, W, Cout, k, 0, nullptr, stream, fmt);
}
int nq_conv_wgrad3_slabs_fmt(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W,
                             int Cout, int k, nq_wgr_seg* seg, int fmt, nq_stream_t stream) {
  if (!seg) return NQ_ERR_INVALID;
  if (fmt & ~nq_conv_wgrad3_split_io(B, Cin, H, W, Cout, k)) return NQ_ERR_UNSUPPORTED;
  return conv_wgrad3_impl(x, dy, dw, db, ws, B, Cin, H, W, Cout, k, 0, seg, stream, fmt);
}

int nq_conv_wgrad3_swapped(const float* x, const float* dy, float* dw, float* ws, int B, int Cin, int H, int W, int Cout, int k,
                           nq_stream_t stream) {
  // the kernel sees the exchanged problem: "x" = dy (Cout channels), "dy" = x (Cin channels)
  return conv_wgrad3_impl(dy, x, dw, nullptr, ws, B, Cout, H, W, Cin, k, k * k, nullptr, stream);
}

int nq_conv_wgrad3_slabs(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W, int Cout,
                         int k, nq_wgr_seg* seg, nq_stream_t stream) {
  if (!seg) return NQ_ERR_INVALID;
  return conv_wgrad3_impl(x, dy, dw, db, ws, B, Cin, H, W, Cout, k, 0, seg, stream);
}

int nq_conv_wgrad3_swapped_slabs(const float* x, const float* dy, float* dw, float* ws, int B, int Cin, int H, int W, int Cout,
                                 int k, nq_wgr_seg* seg, nq_stream_t stream) {
  if (!seg) return NQ_ERR_INVALID;
  return conv_wgrad3_impl(dy, x, dw, nullptr, ws, B, Cout, H, W, Cin, k, k * k, seg, stream);
}

int nq_wgrad_reduce_multi(const nq_wgr_seg* segs, int nseg, nq_stream_t stream) {
  if (!segs || nseg < 0) return NQ_ERR_INVALID;
  WGRMulti t;
  t.nseg = 0;
  int blocks = 0;
  auto flush = [&]() {
    if (t.nseg == 0) return;
    t.blk0[t.nseg] = blocks;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, nq_s(stream), t);
    t.nseg = 0;
    blocks = 0;
  };
  for (int i = 0; i < nseg; ++i) {
    const nq_wgr_seg& h = segs[i];
    if (h.nsplit == 0) continue;   // nothing pending (the kernel wrote dw itself)
    if (!h.slab || !h.dw || h.Cout <= 0 || h.N <= 0 || h.nsplit < 0 || !(h.sg == 1 || h.sg == 4 || h.sg == 16) ||
        (h.db && !h.slab_db))
      return NQ_ERR_INVALID;
    if (t.nseg == WGR_MAXSEG) flush();
    t.s[t.nseg] = h;
    t.blk0[t.nseg] = blocks;
    const int og = 256 / h.sg;
    blocks += (int)(((int64_t)h.Cout * ((h.N + 3) / 4) + h.Cout + og - 1) / og);   // quads of n, then the bias entries
    ++t.nseg;
  }
  flush();
  return nq_launch_status();
}

static int conv_wgrad3_impl(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int Cin, int H, int W,
                            int Cout, int k, int swap_kk, nq_wgr_seg* seg, nq_stream_t stream, int fmt) {
  if (!x || !dy || !dw || !ws || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || Cout <= 0) return NQ_ERR_INVALID;
  if (!(k == 3 || k == 5)) return NQ_ERR_UNSUPPORTED;
  if ((int64_t)Cout * H * W >= (1ll << 31) || (int64_t)Cin * H * W >= (1ll << 31)) return NQ_ERR_UNSUPPORTED;
  if (swap_kk == 0 && nq_conv_wgrad_flat3_ok(B, Cin, H, W, Cout, k)) {
    // few-pixel layer: every output block is owned by one wave for the whole K range -- dw / db are final, nothing pending
    if (fmt) return NQ_ERR_UNSUPPORTED;
    if (seg) *seg = nq_wgr_seg{nullptr, nullptr, dw, db, Cout, Cin * k * k, 0, 0, 0, 0, 1};
    return nq_conv_wgrad_flat3(x, dy, dw, db, B, Cin, H, W, Cout, k, nq_s(stream));
  }
  Wg3Plan p = plan_wgrad3(B, Cin, H, W, Cout, k);
  float* slab = ws;
  float* slab_db = ws + (int64_t)p.nsplit * p.co_pad * p.n_pad;
  hipStream_t st = nq_s(stream);
  int rc = (k == 3) ? nq_conv_wgrad3_k3(x, dy, slab, slab_db, B, Cin, H, W, Cout, p.co_pad, p.n_pad, p.nsplit, p.mi, p.ni, p.pc, fmt, st)
                    : nq_conv_wgrad3_k5(x, dy, slab, slab_db, B, Cin, H, W, Cout, p.co_pad, p.n_pad, p.nsplit, p.mi, p.ni, p.pc, fmt, st);
  if (rc != NQ_OK) return rc;
  const int N = Cin * k * k;
  int64_t total = (int64_t)Cout * N + Cout;
  if (seg) {   // deferred: describe the reduction, nq_wgrad_reduce_multi runs it (same split groups as below)
    *seg = nq_wgr_seg{slab, db ? slab_db : nullptr, dw, db, Cout, N, p.co_pad, p.n_pad, p.nsplit, swap_kk,
                      (total < 65536 && p.nsplit >= 16) ? 16 : 4};
    return NQ_OK;
  }
  if (total < 65536 && p.nsplit >= 16) {
    hipLaunchKernelGGL(wgrad3_reduce_kernel<16>, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, slab, slab_db, dw, db,
                       Cout, N, p.co_pad, p.n_pad, p.nsplit, swap_kk);
  } else {
    hipLaunchKernelGGL(wgrad3_reduce_kernel<4>, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, slab, slab_db, dw, db,
                       Cout, N, p.co_pad, p.n_pad, p.nsplit, swap_kk);
  }
  return nq_launch_status();
}

}  // extern "C"
