// Few-pixel convolution, forward + data gradient, bf16x3 (round 4).
//
// The deep decoder layers (HNeRV dec2: 77 -> 1024, k3 on 10x20 pixels; NeRV dec1 / dec2: 145 -> 1800 on 2x4, 72 -> 576 on 10x20,
// reference models/HNeRV.py:29-42, NeRV.py:24-37) and their data gradients hold most of the PARAMETERS and almost none of
// the pixels.  On the 8 x 32-pixel tiles of conv_igemm3 a 10 x 20 frame fills 39 % of two tiles, the grid only fills the
// chip through split-K slabs, and a finish launch adds them up: 15-25 us per layer for < 0.6 GFLOP.  Here the pixels of ALL
// frames are ONE flat GEMM dimension (B*H*W = 16 ... 400: blocks of 16 consecutive flat pixels, 96 % full), a workgroup
// owns 16*MI output channels x 16*NB pixels, and its NW waves split the K loop between them by 16-channel chunks:
//   * weights: the pre-split operand of nq_weight_layout3 ([k-step][plane][co tile][kq][MT] 16-byte fragments) is read
//     straight from global memory into the A registers -- one coalesced 16-byte load per fragment, no LDS, no barrier;
//   * activations: each wave stages the patch of ITS chunk (the image rows its pixels touch + halo, every frame padded
//     separately: rows of different frames never alias) into a wave-private LDS region as [plane hi/lo][octet][pixel][8 ch]
//     bf16 -- the layout of conv_igemm3 -- and reads B fragments as two 8-byte halves at run-time (channel, tap) offsets
//     (conv3_layout.h: the same K order, tail chunks included);
//   * no workgroup barrier in the K loop (waves are independent); ONE barrier before the cross-wave reduction of the
//     accumulators through LDS in fixed wave order (deterministic), then bias / PixelShuffle / GELU / un-shuffle epilogues.
// Long K loops with few output tiles (data gradients: K = C_out*k*k of the layer) can additionally be split over
// workgroups into slabs that nq_conv_splitk_finish adds (same contract as conv_igemm3).
// Same arithmetic as conv_igemm3: x = hi + lo (two bf16 roundings), hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16,
// fp32 accumulation.  Roofline: latency (a few us per launch); algorithmic flops 2*B*Cout*Cin*k*k*H*W.
#include <cstdlib>

#include "conv3_layout.h"
#include "nq_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

// n / d for the small non-negative integers of this kernel (pixel, item and channel indices < 65536) by a host-computed
// reciprocal: one v_mul_hi instead of the ~35 instructions of a run-time integer division -- a lone wave issues one
// instruction per ~5 cycles, and the ~30 divisions of the first build were a fifth of its 15 us
struct FDiv {
  unsigned d, m;   // m = floor(2^32 / d) + 1 (d >= 2); d == 1 -> identity
};
static inline FDiv make_fdiv(int d) { return FDiv{(unsigned)d, d > 1 ? (unsigned)(0x100000000ull / (unsigned)d) + 1u : 0u}; }
__device__ __forceinline__ int fdiv(int n, FDiv f) { return f.d == 1 ? n : (int)__umulhi((unsigned)n, f.m); }

struct FlatArgs {
  const float* x;
  const u32x4* wt3;
  const float* bias;
  float* y;
  float* z;
  const float* zprev;
  float* slab;             // [nsplit][B][Cout][H][W] when nsplit > 1
  int B, Cin, H, W, Cout, KS, r, epi;
  int P;                   // B*H*W flat pixels
  int MT, co_tiles;        // operand layout: channel tile and number of tiles
  int nchunk, tail, NST, NSTT;
  int ngroups, cgroups;    // pixel groups (16*NB pixels) and channel groups (16*MI channels) of the grid
  int nsplit, per_split;   // workgroup-level split over chunks
  int y_split;             // y as split {hi | lo} words (only with nsplit == 1)
  int ppix;                // 16-byte units per (plane, octet) of a wave's patch
  unsigned x_bytes;
  FDiv dHW, dW, dVH, dNQ, dR, dRR, dNG, dCG;
};

__device__ __forceinline__ void split8f(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __bf16 h0 = (__bf16)v[2 * j], h1 = (__bf16)v[2 * j + 1];
    __bf16 l0 = (__bf16)(v[2 * j] - (float)h0), l1 = (__bf16)(v[2 * j + 1] - (float)h1);
    hi[j] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    lo[j] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
  }
}

template <int KS, int NW, int MI, int NB>
__global__ __launch_bounds__(NW * 64) void conv_flat3_kernel(FlatArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kq = lane >> 4;
  constexpr int KK = KS * KS, pad = KS >> 1;
  constexpr int NSTMAX = (KK + 1) / 2;   // k-steps of a full chunk (tail chunks have fewer)
  const int H = a.H, W = a.W, HW = H * W, PW = W + 2 * pad, VH = H + 2 * pad;
  int id = (int)blockIdx.x;
  const int id1 = fdiv(id, a.dNG), pg = id - id1 * a.ngroups;
  const int split = fdiv(id1, a.dCG), cg = id1 - split * a.cgroups;
  const int c_lo = split * a.per_split, c_hi = min(c_lo + a.per_split, a.nchunk);

  // ---- pixels of this workgroup: flat range [p0, p1) over (frame, y, x) ----
  const int p0 = pg * 16 * NB, p1 = min(p0 + 16 * NB, a.P);
  auto vrow_of = [&](int p, int& x_out) {
    const int b = fdiv(p, a.dHW), rem = p - b * HW, yy = fdiv(rem, a.dW);
    x_out = rem - yy * W;
    return b * VH + yy + pad;   // row of the pixel in the stack of separately padded frames
  };
  int xdummy;
  const int vr0 = vrow_of(p0, xdummy), vr1 = vrow_of(p1 - 1, xdummy);
  const int prows = vr1 - vr0 + 1 + 2 * pad;   // patch row pr <-> virtual row vr0 - pad + pr
  const int ppix = a.ppix;
  int base[NB];   // 16-byte unit of the lane's pixel at tap (0, 0), per pixel block (padding lanes: the last valid pixel)
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    int xx;
    const int p = min(p0 + 16 * nb + l16, a.P - 1);
    const int vr = vrow_of(p, xx);
    base[nb] = (vr - vr0) * PW + xx;
  }
  u32x4* const patch = smem + wave * (4 * ppix);   // [plane][octet][ppix]
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)a.x_bytes, 0x00020000);

  f32x4 acc[MI][NB];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[mi][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // A fragments: unit of (global k-step g, plane, this lane) for channel block mi
  const int64_t plane_stride = (int64_t)a.co_tiles * 4 * a.MT;
  int a_unit[MI];
  bool a_ok[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int cob = (cg * MI + mi) * 16;
    a_ok[mi] = cob < a.co_tiles * a.MT;
    const int cobc = a_ok[mi] ? cob : 0;
    const int tile = cobc / a.MT, co_l = cobc - tile * a.MT;   // (uniform: scalar division)
    a_unit[mi] = tile * 4 * a.MT + kq * a.MT + co_l + l16;
  }

  // ---- staging items (octet, patch row, 4-pixel quad) of this lane, decoded ONCE: the first two rounds (128 items: every
  //      layer of the 3M models) keep their loads in flight together; further rounds take the generic loop ----
  const int nq = (PW + 3) >> 2;   // 4-pixel quads per patch row
  const int nitem = 2 * prows * nq;
  struct Item {
    int off;      // element offset of (frame, channel o*8, row, first pixel of the quad); + (c*16 + j)*HW per channel
    int lds;      // unit index o*ppix + pr*PW + 4q
    int gx0, o;
    bool ok;      // a row inside a frame
  };
  auto decode = [&](int it) {
    Item t;
    const int per_o = prows * nq;
    const int o = it >= per_o ? 1 : 0, rem = it - o * per_o;
    const int pr = fdiv(rem, a.dNQ), q = rem - pr * nq;
    const int vr = vr0 - pad + pr;
    const int b = fdiv(vr, a.dVH), yy = vr - b * VH - pad;
    t.ok = it < nitem && b < a.B && yy >= 0 && yy < H;
    t.gx0 = 4 * q - pad;
    t.o = o;
    t.off = ((b * a.Cin + o * 8) * H + yy) * W + t.gx0;   // < 2^29 elements (host check); negative only in front of the tensor
    t.lds = o * ppix + pr * PW + 4 * q;
    return t;
  };
  const Item it0 = decode(lane), it1 = decode(lane + 64);
  auto load_item = [&](const Item& t, int c, f32x4 (&pv)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = t.ok && c * 16 + t.o * 8 + j < a.Cin;
      const int e0 = t.off + (c * 16 + j) * HW;
      const bool neg = e0 < 0;   // only the left halo of the very first row of the tensor
      const unsigned off = ok ? (unsigned)(neg ? 0 : e0) * 4u : 0xFFFFFF00u;
      f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
      if (neg) v = (e0 == -1) ? f32x4{0.f, v[0], v[1], v[2]} : f32x4{0.f, 0.f, v[0], v[1]};
      pv[j] = v;
    }
  };
  auto store_item = [&](const Item& t, const f32x4 (&pv)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int gx = t.gx0 + e;
      if (gx + pad < PW) {
        const bool cok = gx >= 0 && gx < W;
        float cv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) cv[j] = cok ? pv[j][e] : 0.f;
        u32x4 hi, lo;
        split8f(cv, hi, lo);
        patch[t.lds + e] = hi;
        patch[2 * ppix + t.lds + e] = lo;
      }
    }
  };
  for (int c = c_lo + wave; c < c_hi; c += NW) {
    const int kind = (c == a.nchunk - 1) ? a.tail : 0;
    const int nst = kind ? a.NSTT : a.NST;
    const int64_t g0 = (int64_t)c * a.NST;   // every chunk in front of c is a full one
    // ALL of the chunk's weight fragments on their way before the patch is staged: a k-step is 3*MI*NB MFMAs (~0.1 us),
    // a global load ~1-2 us -- fetched step by step the loop waited for every one of them (first build: 22 us for HNeRV's
    // dec2 forward).  The k-step loop below is branch-free (one basic block from these loads to their uses, so the
    // compiler cannot sink a load into a conditional block behind the patch staging): a tail chunk runs all NSTMAX steps,
    // those past its end with zero weights (their loads re-read its last step).
    u32x4 ah[NSTMAX][MI], al[NSTMAX][MI];
#pragma unroll
    for (int s = 0; s < NSTMAX; ++s)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const u32x4* wp = a.wt3 + (g0 + min(s, nst - 1)) * 2 * plane_stride + a_unit[mi];
        ah[s][mi] = wp[0];
        al[s][mi] = wp[plane_stride];
      }
    __builtin_amdgcn_sched_barrier(0);
    // ---- stage the chunk's patch: 8 channels x 4 pixels per item, both rounds' loads in flight together ----
    {
      f32x4 pv0[8], pv1[8];
      load_item(it0, c, pv0);
      if (nitem > 64) load_item(it1, c, pv1);
      if (lane < nitem) store_item(it0, pv0);
      if (lane + 64 < nitem) store_item(it1, pv1);
      for (int it = lane + 128; it < nitem; it += 64) {   // (bigger patches than any 3M layer has)
        const Item t = decode(it);
        f32x4 pv[8];
        load_item(t, c, pv);
        store_item(t, pv);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    const u32x2* __restrict__ p2 = reinterpret_cast<const u32x2*>(patch);
#pragma unroll
    for (int s = 0; s < NSTMAX; ++s) {
      const bool live = s < nst;   // wave-uniform; dead steps multiply by zero weights
      // the lane group's 8 k-values = two halves of 4 channels at one tap each (conv3_layout.h)
      int chA, tapA, chB, tapB;
      wl3_elem(kind, KK, min(s, nst - 1), kq, 0, chA, tapA);
      wl3_elem(kind, KK, min(s, nst - 1), kq, 4, chB, tapB);
      tapA = min(tapA, KK - 1);   // padded taps carry zero weights
      tapB = min(tapB, KK - 1);
      const int tyA = tapA / KS, tyB = tapB / KS;
      const int huA = 2 * ((chA >> 3) * ppix + tyA * PW + (tapA - tyA * KS)) + ((chA >> 2) & 1);
      const int huB = 2 * ((chB >> 3) * ppix + tyB * PW + (tapB - tyB * KS)) + ((chB >> 2) & 1);
      bf16x8 bh[NB], bl[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int o2 = 2 * base[nb];
        const u32x2 h0 = p2[huA + o2], h1 = p2[huB + o2];
        const u32x2 l0 = p2[4 * ppix + huA + o2], l1 = p2[4 * ppix + huB + o2];
        bh[nb] = __builtin_bit_cast(bf16x8, u32x4{h0.x, h0.y, h1.x, h1.y});
        bl[nb] = __builtin_bit_cast(bf16x8, u32x4{l0.x, l0.y, l1.x, l1.y});
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const u32x4 zero4 = u32x4{0u, 0u, 0u, 0u};
        const bf16x8 fh = __builtin_bit_cast(bf16x8, live ? ah[s][mi] : zero4), fl = __builtin_bit_cast(bf16x8, live ? al[s][mi] : zero4);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, bh[nb], acc[mi][nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh, bl[nb], acc[mi][nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fl, bh[nb], acc[mi][nb], 0, 0, 0);
      }
    }
    // the next chunk's staging overwrites the patch: this wave's reads above have been issued in order before its writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // ---- cross-wave reduction through LDS, fixed wave order ----
  __syncthreads();   // every wave is done with its patch (the regions are re-used below)
  float* const red = reinterpret_cast<float*>(smem);   // [wave][MI*NB][4][64]
  constexpr int TPW = MI * NB * 4 * 64;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) red[wave * TPW + ((mi * NB + nb) * 4 + reg) * 64 + lane] = acc[mi][nb][reg];
  __syncthreads();

  const int Cout = a.Cout, r = a.r, rr = r * r, epi = a.epi;
  for (int t = wave; t < MI * NB; t += NW) {
    const int mi = t / NB, nb = t - mi * NB;
    f32x4 v;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      float sum = red[(t * 4 + reg) * 64 + lane];
      for (int w = 1; w < NW; ++w) sum += red[w * TPW + (t * 4 + reg) * 64 + lane];
      v[reg] = sum;
    }
    const int p = p0 + 16 * nb + l16;
    if (p >= a.P) continue;
    const int b = fdiv(p, a.dHW), rem = p - b * HW, py = fdiv(rem, a.dW), px = rem - py * W;
    const int cob = (cg * MI + mi) * 16 + 4 * kq;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int co = cob + reg;
      if (co >= Cout) continue;
      const int64_t i = (((int64_t)b * Cout + co) * H + py) * W + px;
      if (a.nsplit > 1) {   // raw partial sums; bias / epilogue in nq_conv_splitk_finish
        a.slab[(int64_t)split * a.B * Cout * HW + i] = v[reg];
        continue;
      }
      float val = v[reg] + (a.bias ? a.bias[co] : 0.f);
      if (epi == NQ_EPI_PS_GELU || epi == NQ_EPI_PS) {
        const int C = Cout / rr, cc = fdiv(co, a.dRR), rm = co - cc * rr, si = fdiv(rm, a.dR), sj = rm - si * r;
        const int64_t o = (((int64_t)b * C + cc) * (H * r) + (int64_t)py * r + si) * ((int64_t)W * r) + (int64_t)px * r + sj;
        if (epi == NQ_EPI_PS_GELU) {
          float gv, dv;
          nq_gelu_pair(val, gv, dv);
          a.y[o] = a.y_split ? nq_split_word_f(gv) : gv;
          a.z[o] = dv;
        } else {
          a.z[o] = val;
        }
      } else if (epi == NQ_EPI_DGRAD_GELU) {
        val *= a.zprev[i];
        if (a.y_split) val = nq_split_word_f(val);
        if (r == 1) {
          a.y[i] = val;
        } else {
          const int yq = fdiv(py, a.dR), xq = fdiv(px, a.dR);
          const int ch = co * rr + (py - yq * r) * r + (px - xq * r);
          a.y[(((int64_t)b * Cout * rr + ch) * (H / r) + yq) * (int64_t)(W / r) + xq] = val;
        }
      } else {
        const float ov = (epi == NQ_EPI_TANH) ? tanhf(val) * 0.5f + 0.5f : val;
        a.y[i] = a.y_split ? nq_split_word_f(ov) : ov;
      }
    }
  }
}

#define KS_MI2_OK(k_) ((k_) == 3)   /* two fragment sets of a 5 x 5 chunk (52 x 4 registers) do not fit */

template <int KS, int NW, int MI, int NB>
int launch_flat3(const FlatArgs& a, hipStream_t st) {
  const size_t patch_b = (size_t)NW * 4 * a.ppix * 16, red_b = (size_t)NW * MI * NB * 4 * 64 * 4;
  const size_t lds = patch_b > red_b ? patch_b : red_b;
  if (lds > 160 * 1024) return NQ_ERR_UNSUPPORTED;
  if (int rc = nq_lds_optin<&conv_flat3_kernel<KS, NW, MI, NB>>(lds)) return rc;
  hipLaunchKernelGGL((conv_flat3_kernel<KS, NW, MI, NB>), dim3((unsigned)(a.ngroups * a.cgroups * a.nsplit)), dim3(NW * 64), lds, st, a);
  return nq_launch_status();
}

}  // namespace

// Launch plan of the few-pixel kernel for a (Cin -> Cout, k) convolution over B frames of H x W (pure host function, shared
// with conv3.hip): returns 0 when the shape is not one of its shapes.
extern "C" int nq_conv_flat3_plan(int B, int Cin, int H, int W, int Cout, int k, int* nw, int* nb, int* nsplit, int* per_split) {
  static const int enabled = [] { const char* e = getenv("NQ_FLAT3"); return !(e && e[0] == '0'); }();
  if (!enabled || !(k == 3 || k == 5) || B <= 0 || Cin <= 4 || Cout <= 4 || H <= 0 || W <= 0) return 0;
  const int64_t P = (int64_t)B * H * W;
  if (P > 512 || W > 32 || (int64_t)B * Cin * H * W * 4 >= 0x7FFFFF00ll) return 0;
  const int NBv = P > 16 ? 5 : 1;
  const int nchunk = (Cin + 15) / 16;
  // waves per workgroup = K-splits inside it; 16 only with single-block patches (16 wave-private patches of a 5-block group
  // would not fit the LDS)
  // (16 waves = 128 VGPRs per lane: only with the 10 weight fragments of a 3 x 3 chunk in flight, not the 26 of a 5 x 5 one)
  const int NWv = (nchunk > 8 && NBv == 1 && k == 3) ? 16 : (nchunk > 4 ? 8 : 4);
  const int ngroups = (int)((P + 16 * NBv - 1) / (16 * NBv)), cgroups = (Cout + 15) / 16;
  // workgroup-level split of long K loops (data gradients) while the grid is small: <= ~4 chunks per wave, <= ~512 workgroups
  static const int max_per_wave = [] { const char* e = getenv("NQ_FLAT3_CPW"); return e ? atoi(e) : 1; }();
  int ns = 1;
  const int per_wave = (nchunk + NWv - 1) / NWv;
  if (per_wave > max_per_wave) {
    ns = (per_wave + max_per_wave - 1) / max_per_wave;
    const int cap = 1024 / (ngroups * cgroups);
    if (ns > cap) ns = cap;
    if (ns < 1) ns = 1;
  }
  int per = (nchunk + ns - 1) / ns;
  per = (per + NWv - 1) / NWv * NWv;          // whole rounds of the waves
  ns = (nchunk + per - 1) / per;
  if (nw) *nw = NWv;
  if (nb) *nb = NBv;
  if (nsplit) *nsplit = ns;
  if (per_split) *per_split = per;
  return 1;
}

extern "C" int nq_conv_flat3(const float* x, const void* wt3, const float* bias, float* y, float* z, const float* zprev, float* slab,
                             int B, int Cin, int H, int W, int Cout, int k, int r, int epi, int MT, int tail, int NST, int NSTT,
                             hipStream_t st) {
  int NWv, NBv, ns, per;
  if (!nq_conv_flat3_plan(B, Cin, H, W, Cout, k, &NWv, &NBv, &ns, &per)) return NQ_ERR_UNSUPPORTED;
  FlatArgs a{};
  a.x = x; a.wt3 = reinterpret_cast<const u32x4*>(wt3); a.bias = bias; a.y = y; a.z = z; a.zprev = zprev; a.slab = slab;
  // bit 9 of `epi` (NQ_EPI_Y_SPLIT): y is written as split {hi | lo} words (nq_common.h) -- only without split-K
  a.y_split = (epi & 0x200) ? 1 : 0;
  epi &= 0xFF;
  if (a.y_split && ns > 1) return NQ_ERR_UNSUPPORTED;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.KS = k; a.r = r; a.epi = epi;
  a.P = B * H * W;
  a.MT = MT; a.co_tiles = (Cout + MT - 1) / MT;
  a.nchunk = (Cin + 15) / 16; a.tail = tail; a.NST = NST; a.NSTT = NSTT;
  a.ngroups = (a.P + 16 * NBv - 1) / (16 * NBv);
  a.cgroups = (Cout + 15) / 16;
  a.nsplit = ns; a.per_split = per;
  a.x_bytes = (unsigned)((int64_t)B * Cin * H * W * 4);
  a.dHW = make_fdiv(H * W); a.dW = make_fdiv(W); a.dR = make_fdiv(r > 0 ? r : 1); a.dRR = make_fdiv(r > 0 ? r * r : 1);
  if (ns > 1 && !slab) return NQ_ERR_INVALID;
  // largest patch of any pixel group: rows from the first to the last pixel of the group in the stack of padded frames
  const int pad = k / 2, VH = H + 2 * pad, PW = W + 2 * pad, HW = H * W;
  int maxrows = 0;
  for (int g = 0; g < a.ngroups; ++g) {
    const int p0 = g * 16 * NBv, p1 = (p0 + 16 * NBv < a.P ? p0 + 16 * NBv : a.P) - 1;
    const int v0 = (p0 / HW) * VH + (p0 % HW) / W, v1 = (p1 / HW) * VH + (p1 % HW) / W;
    if (v1 - v0 + 1 + 2 * pad > maxrows) maxrows = v1 - v0 + 1 + 2 * pad;
  }
  a.ppix = (maxrows * PW + 3) / 4 * 4;
  a.dVH = make_fdiv(VH); a.dNQ = make_fdiv((PW + 3) / 4);
  // 32 channels per workgroup (two A-fragment sets per wave) when 16 would need more than one workgroup per CU: the
  // wave-private patches of a 5-block group allow one resident workgroup per CU, and a second round costs a whole
  // workgroup latency (HNeRV dec2 forward: 320 workgroups of 16 channels 31 us, 160 of 32 ...)
  const int MIv = (NBv == 5 && a.cgroups * a.ngroups * ns > 256 && KS_MI2_OK(k)) ? 2 : 1;
  if (MIv == 2) a.cgroups = (Cout + 31) / 32;
  a.dNG = make_fdiv(a.ngroups); a.dCG = make_fdiv(a.cgroups);
#define NQ_FLAT_CASE(NW_, NB_)                                                           \
  if (NWv == NW_ && NBv == NB_ && MIv == 1) return k == 3 ? launch_flat3<3, NW_, 1, NB_>(a, st) : launch_flat3<5, NW_, 1, NB_>(a, st);
  if (MIv == 2 && NBv == 5 && NWv == 4) return launch_flat3<3, 4, 2, 5>(a, st);
  if (MIv == 2 && NBv == 5 && NWv == 8) return launch_flat3<3, 8, 2, 5>(a, st);
  NQ_FLAT_CASE(4, 1) NQ_FLAT_CASE(8, 1) NQ_FLAT_CASE(4, 5) NQ_FLAT_CASE(8, 5)
  if (NWv == 16 && NBv == 1 && k == 3) return launch_flat3<3, 16, 1, 1>(a, st);
#undef NQ_FLAT_CASE
  return NQ_ERR_UNSUPPORTED;
}
