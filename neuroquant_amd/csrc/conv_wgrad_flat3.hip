// Few-pixel weight gradient, bf16x3 (round 4; companion of conv_flat3.hip).
//
// dW[co][n] = sum_p dY[co][p] * X[n][p],  n = (ci, kh, kw), p = the pixels of ALL frames as one flat K dimension (B*H*W = 16 ...
// 512): for the deep layers (HNeRV dec2: 1024 x 693 outputs over 400 pixels; NeRV dec1: 1800 x 1305 over 16) the K loop is
// 1 ... 16 MFMA steps, so NO split-K: every 16 x 16 output block is owned by one wave for the whole K range and written to dW
// directly -- no slabs (dec2's were 50 MB), no reduction launch.  (The tiled kernels split K = pixels over the chip and leave
// [split][C_out][C_in k^2] slabs: right for 10^5 ... 10^6 pixels, wrong for 400.)
//   * workgroup = 4 waves = 64 output channels x 64 columns n; wave w owns channels 16w .. 16w+15 x 4 column blocks;
//   * X: the <= 8 input channels its 64 columns touch are staged ONCE into LDS as {bf16 hi | bf16 lo} words, every frame zero-
//     padded separately ([channel][frame][(H+2p) x (W+2p)]), plus a table ptab[p] = padded index of flat pixel p; a B fragment
//     (column n = (ci, tap), 8 consecutive flat pixels) is 8 ds_read_b32 at column base + ptab[p .. p+7] and 8 v_perm;
//   * dY: A fragments (8 consecutive pixels of one channel: two 16-byte loads; H*W % 8 == 0 keeps a group inside one frame)
//     straight from global memory, split into bf16 hi / lo in registers; their row sums give the bias gradient;
//   * 3 MFMAs per (A, B) pair: hi*hi + hi*lo + lo*hi, fp32 accumulation, fixed order: deterministic.
// Same arithmetic as conv_wgrad3 (split operands, fp32 accumulate); only the summation order over pixels differs.
#include <cstdlib>

#include "nq_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using i32x4 = __attribute__((ext_vector_type(4))) int;

struct FDivW {
  unsigned d, m;
};
static inline FDivW make_fdivw(int d) { return FDivW{(unsigned)d, d > 1 ? (unsigned)(0x100000000ull / (unsigned)d) + 1u : 0u}; }
__device__ __forceinline__ int fdivw(int n, FDivW f) { return f.d == 1 ? n : (int)__umulhi((unsigned)n, f.m); }

struct WFlatArgs {
  const float* x;    // (B, Cin, H, W)
  const float* dy;   // (B, Cout, H, W)
  float* dw;         // (Cout, Cin, k, k)
  float* db;         // (Cout) or NULL
  int B, Cin, H, W, Cout, KS, N, P, Ppad, ntiles, ctiles, maxch;
  FDivW dHW, dW, dKK, dKS, dNT, dPPIX, dPW, dB;
};

__device__ __forceinline__ unsigned pack_split(float v) {   // {bf16 hi | bf16 lo} of one value
  const __bf16 h = (__bf16)v;
  const __bf16 l = (__bf16)(v - (float)h);
  return ((unsigned)__builtin_bit_cast(unsigned short, h) << 16) | (unsigned)__builtin_bit_cast(unsigned short, l);
}

__global__ __launch_bounds__(256) void conv_wgrad_flat3_kernel(WFlatArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned smem_w[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, kq = lane >> 4;
  const int KS = a.KS, KK = KS * KS, pad = KS >> 1;
  const int H = a.H, W = a.W, HW = H * W, PW = W + 2 * pad, PPIX = (H + 2 * pad) * PW;
  const int ct = fdivw((int)blockIdx.x, a.dNT), nt = (int)blockIdx.x - ct * a.ntiles;   // n-tile fastest: the co-tile's dY stays in L2
  const int n0 = nt * 64, co0 = ct * 64 + wave * 16;
  const int ci0 = fdivw(n0, a.dKK);
  const int nch = min(fdivw(min(n0 + 63, a.N - 1), a.dKK) - ci0 + 1, a.maxch);
  // LDS: ptab[Ppad] ints | xs[nch][B][PPIX] words.  (Pixels past the end of the last k-step point at word 0: their dY values
  // are zero, so whatever finite x they multiply does not matter.)
  int* const ptab = reinterpret_cast<int*>(smem_w);
  unsigned* const xs = smem_w + a.Ppad;
  const int xwords = nch * a.B * PPIX;
  for (int p = tid; p < a.Ppad; p += 256) {
    int v = 0;
    if (p < a.P) {
      const int b = fdivw(p, a.dHW), rem = p - b * HW, py = fdivw(rem, a.dW), px = rem - py * W;
      v = b * PPIX + py * PW + px;   // pixel (py, px) at tap (0, 0) of the padded frame: tap (dy, dx) adds dy*PW + dx
    }
    ptab[p] = v;
  }
  // (eight loads in flight per thread: one load per iteration serialised ~17 global-memory round trips per workgroup)
  for (int e0 = tid; e0 < xwords; e0 += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 256;
      const int ec = min(e, xwords - 1);
      const int cb = fdivw(ec, a.dPPIX), q = ec - cb * PPIX;      // (channel, frame) plane, padded pixel
      const int c = fdivw(cb, a.dB), b = cb - c * a.B;
      const int qy = fdivw(q, a.dPW), yy = qy - pad, xx = q - qy * PW - pad;
      const bool ok = e < xwords && yy >= 0 && yy < H && xx >= 0 && xx < W && ci0 + c < a.Cin;
      const float* src = a.x + (ok ? (((int64_t)b * a.Cin + ci0 + c) * H + yy) * W + xx : 0);
      const float t = *src;                                       // always a valid address
      v[u] = ok ? t : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + u * 256;
      if (e < xwords) xs[e] = pack_split(v[u]);
    }
  }
  __syncthreads();

  // column bases of this lane's 4 B fragments: n = n0 + 16 j + l16 -> (ci, tap) -> word offset of tap in channel plane
  int nbase[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = min(n0 + 16 * j + l16, a.N - 1);   // padded columns read column N-1 (masked at the store)
    const int ci = fdivw(n, a.dKK), tap = n - ci * KK, ty = fdivw(tap, a.dKS), tx = tap - ty * KS;
    nbase[j] = (ci - ci0) * a.B * PPIX + ty * PW + tx;
  }
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float rsum = 0.f;   // bias gradient: row sum of this lane's dY values
  const int co = co0 + l16;
  const bool co_ok = co < a.Cout;
  const int steps = a.Ppad >> 5;
  // A fragment of k-step s: 8 consecutive pixels of channel co (inside one frame: HW % 8 == 0); fetched ONE STEP AHEAD -- a
  // k-step is ~0.3 us of issue time, a global load 1-2 us: loaded where it was used, every step waited for it (24 us for
  // HNeRV's dec2: 13 steps)
  auto load_a = [&](int s, f32x4& v0, f32x4& v1) {
    const int p = 32 * s + 8 * kq;
    const int b = fdivw(p, a.dHW), rem = p - b * HW;
    const bool ok = co_ok && p < a.P;
    const float* src = a.dy + ((int64_t)(ok ? b : 0) * a.Cout + (co_ok ? co : 0)) * HW + (ok ? rem : 0);
    v0 = *reinterpret_cast<const f32x4*>(src);        // (always a valid address; masked below)
    v1 = *reinterpret_cast<const f32x4*>(src + 4);
    if (!ok) v0 = v1 = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  f32x4 n0v, n1v;
  load_a(0, n0v, n1v);
  for (int s = 0; s < steps; ++s) {
    const int p = 32 * s + 8 * kq;
    float av[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      av[e] = n0v[e];
      av[4 + e] = n1v[e];
    }
    load_a(min(s + 1, steps - 1), n0v, n1v);
    u32x4 ah, al;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const __bf16 h0 = (__bf16)av[2 * e], h1 = (__bf16)av[2 * e + 1];
      const __bf16 l0 = (__bf16)(av[2 * e] - (float)h0), l1 = (__bf16)(av[2 * e + 1] - (float)h1);
      ah[e] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
      al[e] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
      rsum += av[2 * e] + av[2 * e + 1];
    }
    const i32x4 q0 = *reinterpret_cast<const i32x4*>(ptab + p), q1 = *reinterpret_cast<const i32x4*>(ptab + p + 4);
    const int po[8] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
    const bf16x8 fah = __builtin_bit_cast(bf16x8, ah), fal = __builtin_bit_cast(bf16x8, al);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned w[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) w[e] = xs[nbase[j] + po[e]];
      u32x4 bh, bl;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bh[e] = __builtin_amdgcn_perm(w[2 * e + 1], w[2 * e], 0x07060302u);   // {hi(w[2e+1]), hi(w[2e])}
        bl[e] = __builtin_amdgcn_perm(w[2 * e + 1], w[2 * e], 0x05040100u);   // {lo(w[2e+1]), lo(w[2e])}
      }
      const bf16x8 fbh = __builtin_bit_cast(bf16x8, bh), fbl = __builtin_bit_cast(bf16x8, bl);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah, fbh, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah, fbl, acc[j], 0, 0, 0);
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal, fbh, acc[j], 0, 0, 0);
    }
  }
  // acc[j][r] = dW[co0 + 4 kq + r][n0 + 16 j + l16]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = n0 + 16 * j + l16;
    if (n >= a.N) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = co0 + 4 * kq + r;
      if (c < a.Cout) a.dw[(int64_t)c * a.N + n] = acc[j][r];
    }
  }
  if (a.db && nt == 0) {   // bias gradient: the four lane groups hold the four pixel octets of every k-step
    rsum += __shfl_xor(rsum, 16, 64);
    rsum += __shfl_xor(rsum, 32, 64);
    if (kq == 0 && co_ok) a.db[co] = rsum;
  }
}

}  // namespace

// 1 when the few-pixel weight-gradient kernel takes this shape (pure host function, shared with conv3.hip)
extern "C" int nq_conv_wgrad_flat3_ok(int B, int Cin, int H, int W, int Cout, int k) {
  static const int enabled = [] { const char* e = getenv("NQ_WGRAD_FLAT3"); return !(e && e[0] == '0'); }();
  if (!enabled || !(k == 3 || k == 5) || B <= 0 || Cin <= 0 || Cout <= 4 || H <= 0 || W <= 0) return 0;
  const int64_t P = (int64_t)B * H * W;
  if (P > 512 || (H * W) % 8 != 0) return 0;
  if ((int64_t)Cin * k * k * Cout < 65536) return 0;   // toy layers stay on the exact-fp32 kernel
  const int pad = k / 2, maxch = (64 + k * k - 2) / (k * k) + 1;
  const int64_t words = (int64_t)((P + 31) / 32 * 32) + (int64_t)maxch * B * (H + 2 * pad) * (W + 2 * pad) + 1;
  return words * 4 <= 160 * 1024;
}

extern "C" int nq_conv_wgrad_flat3(const float* x, const float* dy, float* dw, float* db, int B, int Cin, int H, int W, int Cout,
                                   int k, hipStream_t st) {
  if (!nq_conv_wgrad_flat3_ok(B, Cin, H, W, Cout, k)) return NQ_ERR_UNSUPPORTED;
  WFlatArgs a{};
  a.x = x; a.dy = dy; a.dw = dw; a.db = db;
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.KS = k;
  a.N = Cin * k * k; a.P = B * H * W; a.Ppad = (a.P + 31) / 32 * 32;
  a.ntiles = (a.N + 63) / 64; a.ctiles = (Cout + 63) / 64;
  a.maxch = (64 + k * k - 2) / (k * k) + 1;
  a.dHW = make_fdivw(H * W); a.dW = make_fdivw(W); a.dKK = make_fdivw(k * k); a.dKS = make_fdivw(k); a.dNT = make_fdivw(a.ntiles);
  const int pad = k / 2;
  a.dPPIX = make_fdivw((H + 2 * pad) * (W + 2 * pad)); a.dPW = make_fdivw(W + 2 * pad); a.dB = make_fdivw(B);
  const size_t lds = ((size_t)a.Ppad + (size_t)a.maxch * B * (H + 2 * pad) * (W + 2 * pad) + 1) * 4;
  if (int rc = nq_lds_optin<&conv_wgrad_flat3_kernel>(lds)) return rc;
  hipLaunchKernelGGL(conv_wgrad_flat3_kernel, dim3((unsigned)(a.ntiles * a.ctiles)), dim3(256), lds, st, a);
  return nq_launch_status();
}
