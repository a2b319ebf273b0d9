// Implicit-GEMM convolution with fp32-equivalent accuracy on the BF16 matrix pipe ("bf16x3"):
//   every fp32 operand x is split into x_hi = bf16(x), x_lo = bf16(x - x_hi) and each product is formed as
//   a_hi*b_hi + a_hi*b_lo + a_lo*b_hi with fp32 accumulation in v_mfma_f32_16x16x32_bf16 (the dropped lo*lo term is
//   2^-16 relative, the same order as the fp32 rounding of a length-1000 dot product).  3 MFMAs of 16 cycles replace
//   8 fp32 MFMAs of 32 cycles per 16x16x32 block: 5.3x the fp32-MFMA rate; stride 1, pad KS/2, NCHW fp32 in HBM.
//   Included once per kernel size (NQ_KS = 3, 5).
//
// GEMM view as in conv_igemm_impl.h: D[co][pixel] = sum_k W[co][k] * X[k][pixel]; A = weights, B = activations, so the
// accumulator/epilogue layout is identical to the fp32 kernel (16 pixels of a row on 16 lanes, 4 conv channels in a
// lane's 4 registers).
//
// K order: input channels are processed in chunks of 16 (2 octets); one MFMA k-step (32) = 2 taps x 2 octets: lane
// group kq = lane>>4 holds the 8 channels of octet (kq&1) at tap 2*step + (kq>>1).  (KS*KS is odd: the last step's
// second tap has zero weights.)
//
// Workgroup = 4 waves = 8 rows x 32 columns of one frame x MT = 16*MI output channels (MI <= 5); wave w owns rows
// 2w, 2w+1 = 4 pixel blocks: 4*MI accumulators.
// LDS (bf16, hi and lo planes):
//   patch  [2 buf][plane][octet][PH*PW pixels][8 ch]   16-byte fragment reads, 16 consecutive pixels -> conflict-free
//   weights[2 buf][plane][kq][MT][8 k]                  streamed per k-step from a pre-split, pre-ordered global copy
// The patch of the next channel chunk and the weights of the next step are loaded into registers before the MFMAs of
// the current step and written to the other buffer after them; one barrier per step.
//
// Roofline: MFMA bf16 (2.5 PFLOP/s dense, 3 MFMA flops per algorithmic flop -> 833 TFLOP/s fp32-equivalent);
// algorithmic flops = 2*Cout*Cin*KS^2*H*W*B.
#include <cstdlib>
#include <type_traits>

#include "nq_common.h"

#ifndef NQ_KS
#error "define NQ_KS before including conv_igemm3_impl.h"
#endif

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef f32x4 f32x4_u __attribute__((aligned(4)));
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

struct Conv3Args {
  const float* x;
  const void* wt3;  // [chunk][step][plane][co tile][kq][MT][8] bf16 (nq_weight_layout3)
  const float* bias;
  float* y;
  float* z;
  const float* zprev;
  int B, Cin, H, W, Cout, r, epi, tiles_x, tiles, nchunk, co_tiles;
  int nsplit, per_split;  // split-K over channel chunks: blockIdx.z = b*nsplit + split, chunks [split*per, +per)
  float* slab;            // [nsplit][B][Cout][H][W] raw partial sums when nsplit > 1 (nq_conv_splitk_finish adds them)
  int lds_epi;            // the launch reserved MT*1024 B of LDS: the data-gradient epilogue may transpose through it
  int y_split;            // y is written as split {hi | lo} words (nq_common.h) instead of floats; x: template parameter XS
  unsigned w_bytes;       // size of the wt3 operand (buffer resource range)
  int tail;               // shape of the LAST 16-channel chunk, by the channels r it really holds (no MFMAs on all-zero k-values):
                          //   0: r > 12   NST   k-steps of (2 octets x 2 taps)                                  13 for k = 5
                          //   3: r <= 12  NST12 steps: octet 0 as (1 octet x 1 tap) slots, the half octet 1 as
                          //               (4 channels x 2 taps) slots, 4 slots per step                          10
                          //   2: r <= 8   NST8  steps of (1 octet x 4 taps)                                       7
                          //   1: r <= 4   NST4  steps of (4 channels x 8 taps)                                    4
};

// Timing experiments only (never in the product build): -DNQ_IG3_ABL=n compiles the kernel WITHOUT one of its parts
// (results are wrong): 1 patch global loads, 2 patch conversion + LDS stores, 3 the MFMAs, 4 the epilogue's global
// traffic, 5 weight LDS stores, 6 the B-fragment LDS reads, 7 patch loads confined to a 256 KiB window (L2 hits),
// 8 no patch re-staging after the first chunk, 9 no weight staging after the prologue, 10 both (operands stay random),
// 11 (LDS-DMA build) no barrier after even k-steps
// (tools/ablate_igemm3.sh).
#ifndef NQ_IG3_ABL
#define NQ_IG3_ABL 0
#endif
#ifndef NQ_IG3_OCC3_DEFAULT
#define NQ_IG3_OCC3_DEFAULT 40
#endif
#ifndef NQ_IG3_SPREAD
#define NQ_IG3_SPREAD 1
#endif
#ifndef NQ_IG3_PSTD
#define NQ_IG3_PSTD 3   // the next chunk's patch loads are issued this many k-steps before the chunk ends
#endif
constexpr int KS = NQ_KS;
constexpr int KK = KS * KS;
constexpr int PAD = KS / 2;
constexpr int TH = 8, TW = 32;
constexpr int PH = TH + KS - 1, PW = TW + KS - 1;
constexpr int NPIX = PH * PW;
constexpr int PP = (NPIX + 15) / 16 * 16;  // octet-plane stride in pixels (16-byte units): multiple of 256 B
constexpr int NST = (KK + 1) / 2;          // k-steps per channel chunk
constexpr int NST8 = (KK + 3) / 4;         // k-steps of a tail chunk of <= 8 channels (lane group kq -> tap 4*step + kq)
constexpr int NHALF = (KK + 1) / 2;        // (4 channels x 2 taps) slots of a half octet
constexpr int NST4 = (NHALF + 3) / 4;      // k-steps of a tail chunk of <= 4 channels
constexpr int NST12 = (KK + NHALF + 3) / 4;  // k-steps of a tail chunk of 9..12 channels
constexpr int nst_of_kind(int kind) { return kind == 1 ? NST4 : (kind == 2 ? NST8 : (kind == 3 ? NST12 : NST)); }
// kind of the last 16-channel chunk (Conv3Args::tail) for Cin input channels on the 16*mi-channel tile; shared with
// nq_weight_layout3 (conv3.hip), which must lay the operand out the same way
static inline int nq_conv3_tail_kind(int Cin, int mi) {
  const int r = Cin - 16 * ((Cin + 15) / 16 - 1);
  return r <= 4 ? 1 : (r <= 8 ? 2 : (r <= 12 ? 3 : 0));
}
constexpr int CC = 16;                     // channels per chunk
constexpr int PATCH_U4 = 2 * 2 * PP;       // 16-byte units per patch buffer: [plane][octet][PP]
constexpr int NITEM = 2 * NPIX;            // staging items (octet, pixel)
constexpr int IPT = (NITEM + 255) / 256;

template <int I0, int N, class F>
__device__ __forceinline__ void steps3(F&& f) {
  if constexpr (I0 < N) {
    f(std::integral_constant<int, I0>{});
    steps3<I0 + 1, N>(f);
  }
}


// split 8 floats into packed bf16 hi / lo vectors (round-to-nearest-even both times)
__device__ __forceinline__ void split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    __bf16 h0 = (__bf16)v[2 * j], h1 = (__bf16)v[2 * j + 1];
    __bf16 l0 = (__bf16)(v[2 * j] - (float)h0), l1 = (__bf16)(v[2 * j + 1] - (float)h1);
    hi[j] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    lo[j] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
  }
}

// WPE = waves per SIMD the register allocation aims at: 2 for the long K loops (the MFMA-bound layers: a third wave only
// queues for the same matrix pipe), 3 for the tiles of <= 48 channels on short K loops (NeRV's last blocks: <= 30 k-steps
// between a prologue and an epilogue of global-memory latency -- bound by latency, not by issue slots; tools/bench_nerv_tail.py).
// With the weights travelling by LDS-DMA the 48-channel tile holds 167 VGPRs and 48.5 KB of LDS: three workgroups per CU
// (with two weight register sets it needed 264 B of scratch for that: 132 -> 202 us).
// XS: the input x holds split {hi | lo} words (written by a producer with y_split): staging re-packs halves, no conversion.
template <int MI, int WPE = 2, bool XS = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void conv_igemm3_kernel(Conv3Args a) {
  constexpr int MT = 16 * MI;
  constexpr int W_U4 = 2 * 4 * MT;            // 16-byte units per weight buffer: [plane][kq][MT]
  constexpr int WPT = (W_U4 + 255) / 256;     // per thread

  extern __shared__ __attribute__((aligned(16))) u32x4 smem[];
  u32x4* const patch0 = smem;                  // 1 buffer of PATCH_U4 (re-filled between two barriers per chunk)
  // Tiles of >= 48 channels fetch their weights by LDS-DMA (buffer_load_dwordx4 ... lds) into a ring of four buffers, three
  // k-steps ahead; the 16-/32-channel tiles stage them through two register sets into two buffers, one barrier per step.
  // (Round 2's register-staged ring of four for the 80-channel tile was superseded by the DMA ring and is gone.)
  constexpr bool WDMA = MI >= 3;
  // WDMA ring: NB buffers, k-step g+D fetched during step g (D = NB - 1): 3 steps of ~1 us cover an L2 round trip.
  // The 16-/32-channel tiles keep the register staging: their k-steps hold 12-24 MFMAs, and one LDS-DMA piece per step
  // costs a wave more issue time than a load + ds_write pair (NeRV's 96 -> 24 data gradient at 320x640: 90 us with
  // registers at three waves per SIMD, 97 us with a ring of four, 103 us with a ring of eight).
  constexpr int NB = 4, WD = NB - 1;
  constexpr int WMASK = WDMA ? NB - 1 : 1;
  // WDMA: the weight operand is already the LDS image ([k-step][plane][kq][MT] 16-byte units, bf16 hi / lo split done by
  // nq_weight_layout3), so a k-step's W_U4 units go global -> LDS directly: WPT `buffer_load_dwordx4 ... lds` per wave
  // (1 KiB each: wave-uniform LDS base + lane * 16), no registers, no ds_write, and a prefetch distance of THREE k-steps
  // into a ring of four buffers at no register cost.  Every wave issues exactly WPT pieces per k-step, so that the counted
  // waits below are the same for all waves: the pieces past W_U4 in the last round (W_U4 is a multiple of 128 units, a
  // piece is 64: whole pieces) REPEAT the wave's first piece -- same source, same destination, the same bytes written twice
  // (round 4; a 2 KB dump area behind the ring before: without it the 48-channel tile of a 5 x 5 layer is 52.2 KB, three
  // workgroups per CU fit).
  constexpr int WS = W_U4;                     // buffer stride in 16-byte units
  constexpr int WDUMP = WPT * 256 - W_U4;      // units of the dump area (0 for the 32- and 64-channel tiles)
  u32x4* const wl0 = smem + PATCH_U4;          // weight buffers: k-step g lives in buffer g & WMASK

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  // 1-D grid, XCD-chunked: logical id = (z * tiles + tile) * co_tiles + cot -- the co-tiles of one patch and the tiles
  // next to it (shared halo rows, shared 128-byte lines at the row ends) run together behind one L2
  const int lid = nq_xcd_chunk((int)blockIdx.x, (int)gridDim.x);
  const int cot = lid % a.co_tiles, lt = lid / a.co_tiles;
  const int tile = lt % a.tiles, bz = lt / a.tiles;
  const int tile_x = tile % a.tiles_x, tile_y = tile / a.tiles_x;
  const int x0 = tile_x * TW, y0 = tile_y * TH;
  const int co0 = cot * MT;
  const int b = bz / a.nsplit, split = bz - b * a.nsplit;
  const int c_lo = split * a.per_split;  // first chunk of this split
  const int H = a.H, W = a.W, Cin = a.Cin - c_lo * CC;  // channels from this split's first one on
  const int64_t HW = (int64_t)H * W;
  const float* __restrict__ xb = a.x + ((int64_t)b * a.Cin + (int64_t)c_lo * CC) * HW;
  // weights of (chunk c, step s): base + ((c*NST + s) * co_tiles*2 ... see nq_weight_layout3: [c][s][plane][tile][kq][MT]
  const int64_t w_plane_stride = (int64_t)a.co_tiles * 4 * MT;   // in 16-byte units
  const int64_t w_step_stride = 2 * w_plane_stride;

  // ---- staging state: weights are prefetched TWO k-steps ahead (two register sets; an L2 round trip under load is
  //      ~2-3 k-steps of MFMA time), the next chunk's patch three steps before it is needed ----
  // Patch staging item = (channel octet, patch row, 4-pixel quad): ONE 16-byte global load per channel (8 per thread
  // and chunk instead of 32 scalar ones -- global memory instructions cost ~150 cycles of CU time each, whatever their
  // width), split into bf16 hi/lo per pixel afterwards.  2 * PH * QP items <= 256: one item per thread.
  constexpr int QP = (PW + 3) / 4;
  static_assert(2 * PH * QP <= 256, "one staging item per thread");
  const int it_oct = tid / (PH * QP), it_rem = tid - it_oct * (PH * QP);
  const int it_r = it_rem / QP, it_q = it_rem - it_r * QP;
  const int it_gy = y0 - PAD + it_r, it_gx = x0 - PAD + 4 * it_q;
  const bool it_act = tid < 2 * PH * QP;
  const bool it_row = it_act && it_gy >= 0 && it_gy < H;
  // Every global load of the K loop is UNCONDITIONAL and branch-free: the compiler can then wait for the oldest register
  // set with a counted s_waitcnt vmcnt(N) while the younger sets stay in flight.  (With a branch around any load it falls
  // back to vmcnt(0): publishing the weights of step g+1 then also waited for the loads of step g+2 issued half a step
  // earlier -- a full L2 round trip exposed per k-step.  Ablation: without the patch loads OR without the weight loads the
  // dec5 data gradient ran 23 % faster.)  Patch quads are bounds-checked BUFFER loads at 32-bit byte offsets: a row
  // outside the image or a channel >= Cin gets an out-of-range offset and reads as zero; the partial quads at the left /
  // right image edge are masked when the values are converted (NQ3_STORE_PATCH), under a block-uniform branch.
  constexpr unsigned OOB = 0xFFFFFF00u;      // >= num_records (checked by the host: the activation tensor is < 4 GiB)
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x), 0, (int)(unsigned)((int64_t)a.B * a.Cin * HW * 4), 0x00020000);
  // byte offset of (this split's channel it_oct*8, row, quad); negative only for the quad that starts in the left halo
  // of the very first row of the tensor: that one is loaded from offset 0 and shifted into place (prologue)
  // (the sign is taken from the 64-bit value: offsets of tensors between 2 and 4 GiB are valid 32-bit UNSIGNED numbers whose
  // truncation to int is negative; all later offset arithmetic is unsigned and exact modulo 2^32)
  const int64_t it_base64 = (((int64_t)b * a.Cin + (int64_t)c_lo * CC + it_oct * 8) * HW + (int64_t)it_gy * W + it_gx) * 4;
  const unsigned it_base = (unsigned)it_base64;
  const unsigned HWb = (unsigned)(HW * 4);
  const bool it_neg = it_row && it_base64 < 0;
  const bool x_edge = (x0 == 0) || (x0 + TW + PAD > W);   // block-uniform: some halo columns are outside the image
  f32x4 pv[8];
  u32x4 wvA[WPT], wvB[WPT];
#define NQ3_LOAD_PATCH_J(CH, J)                                                                       \
  {                                                                                                   \
    /* channels of this item's octet that exist (0 for a thread without an item / a row outside the image): the   \
       offsets are selected with bit masks, not with ?: -- a conditional here comes back as a branch around the load */ \
    const int nv_ = (NQ_IG3_ABL != 1 && it_row) ? min(max(Cin - ((CH) * CC + it_oct * 8), 0), 8) : 0;  \
    unsigned o_ = it_base + (unsigned)((CH) * CC + (J)) * HWb;                                        \
    if ((J) == 0 && (CH) == 0 && it_neg) o_ = 0u;                                                     \
    if (NQ_IG3_ABL == 7) o_ &= 0x3FFF0u;   /* timing only: every patch load hits a 256 KiB window (L2) */ \
    const unsigned sel_ = (unsigned)(((J) - nv_) >> 31);   /* all ones when J < nv_ */                  \
    const unsigned off_ = (o_ & sel_) | (OOB & ~sel_);                                                \
    pv[J] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, off_, 0, 0));       \
  }
#define NQ3_LOAD_PATCH(CH)                                                                            \
  { _Pragma("unroll") for (int j = 0; j < 8; ++j) NQ3_LOAD_PATCH_J(CH, j) }
#define NQ3_STORE_PATCH(DST)                                                                          \
  if (NQ_IG3_ABL != 2 && it_act) {                                                                    \
    if (x_edge) {                                                                                     \
      _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_)                                                \
        if (it_gx + e_ < 0 || it_gx + e_ >= W) {                                                      \
          _Pragma("unroll") for (int j = 0; j < 8; ++j) pv[j][e_] = 0.f;                              \
        }                                                                                             \
    }                                                                                                 \
    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) {                                                \
      if (4 * QP == PW || 4 * it_q + e_ < PW) {                                                       \
        const float c_[8] = {pv[0][e_], pv[1][e_], pv[2][e_], pv[3][e_], pv[4][e_], pv[5][e_], pv[6][e_], pv[7][e_]}; \
        u32x4 hi_, lo_;                                                                               \
        if constexpr (XS) {                                                                           \
          _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                          \
            const unsigned w0_ = __builtin_bit_cast(unsigned, c_[2 * j_]), w1_ = __builtin_bit_cast(unsigned, c_[2 * j_ + 1]); \
            hi_[j_] = __builtin_amdgcn_perm(w1_, w0_, 0x07060302u);                                   \
            lo_[j_] = __builtin_amdgcn_perm(w1_, w0_, 0x05040100u);                                   \
          }                                                                                           \
        } else {                                                                                      \
          split8(c_, hi_, lo_);                                                                       \
        }                                                                                             \
        (DST)[it_oct * PP + it_r * PW + 4 * it_q + e_] = hi_;                                         \
        (DST)[2 * PP + it_oct * PP + it_r * PW + 4 * it_q + e_] = lo_;                                \
      }                                                                                               \
    }                                                                                                 \
  }
  // weights of global step G (= chunk*NST + step): [G][plane][tile][kq][MT]
#define NQ3_LOAD_W(SET, G)                                                                            \
  {                                                                                                   \
    /* uniform byte offset of the k-step (past the end: the last step again, never stored) + fixed per-thread offsets */ \
    const unsigned so_ = (unsigned)(c_lo * NST + min((int)(G), Gm1)) * w_step_bytes;                  \
    _Pragma("unroll") for (int i = 0; i < WPT; ++i)                                                   \
      SET[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wvo[i], so_, 0)); \
  }
#define NQ3_STORE_W(SET, DST)                                                                         \
  _Pragma("unroll") for (int i = 0; i < WPT; ++i) {                                                   \
    const int f_ = tid + i * 256;                                                                     \
    if (NQ_IG3_ABL != 5 && (i + 1 < WPT || f_ < W_U4)) (DST)[f_] = SET[i];                            \
  }

  // (the address-space cast of the builtin's LDS operand only exists in the device pass; the host pass of hipcc parses this
  // body too)
#if defined(__HIP_DEVICE_COMPILE__)
#define NQ3_DMA_PIECE(DST, VOFF, SOFF) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(DST), 16, VOFF, SOFF, 0, 0);
#else
#define NQ3_DMA_PIECE(DST, VOFF, SOFF) (void)(DST);
#endif
  // WDMA: k-step G -> LDS buffer G & (NB - 1) (past the end of the split: the last step again, into a buffer nobody reads)
#define NQ3_DMA_W(G)                                                                                  \
  {                                                                                                   \
    const unsigned so_ = (unsigned)(c_lo * NST + min((int)(G), Gm1)) * w_step_bytes;                  \
    u32x4* const d_ = wl0 + ((G) & (NB - 1)) * WS + wave_u * 64;                                             \
    _Pragma("unroll") for (int i = 0; i < WPT; ++i) {                                                 \
      if (WDUMP == 0 || i + 1 < WPT) NQ3_DMA_PIECE(d_ + i * 256, wvo[i], so_)                         \
      else NQ3_DMA_PIECE((i * 256 + wave_u * 64 < W_U4) ? d_ + i * 256 : d_, wvo[i], so_) \
    }                                                                                                 \
  }
  // counted wait for this wave's LDS-DMA pieces: at most N VMEM operations still in flight
#define NQ3_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
  // workgroup barrier that does NOT drain the VM counter (__syncthreads() waits vmcnt(0) while an LDS-DMA is in flight):
  // this wave's LDS reads / writes retired, then the bare barrier
#define NQ3_BARRIER_LDS()                                    \
  {                                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
    __builtin_amdgcn_s_barrier();                            \
    asm volatile("" ::: "memory");                           \
  }

  f32x4 acc[MI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) acc[mi][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // B fragment base (16-byte units): octet (kq&1), pixel (row 2*wave, col l16); + tap offset + block offset
  const int b_lane = (kq & 1) * PP + (2 * wave) * PW + l16;
  const bool odd_tap = (kq >> 1) != 0;
  // A fragment base (16-byte units): [kq][MT] + l16
  const int a_lane = kq * MT + l16;
  const u32x4* __restrict__ pb = patch0 + b_lane;

  // ---- prologue: step 0 -> LDS buffer 0 (via set A), step 1 in flight in set B ----
  const int nchunk = min(a.per_split, a.nchunk - c_lo);
  const int tail = (c_lo + nchunk == a.nchunk) ? a.tail : 0;   // this split ends with a short tail chunk of that kind
  const int nfull = nchunk - (tail ? 1 : 0);
  const int G = nfull * NST + (tail == 1 ? NST4 : (tail == 2 ? NST8 : (tail == 3 ? NST12 : 0)));
  const int Gm1 = G - 1;
  // per-thread byte offsets of its weight units within one k-step (fixed): unit f = tid + i*256 of [plane][kq][MT]; a
  // thread without an item in the last round repeats its first unit (see WDUMP above)
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wt3), 0, (int)a.w_bytes, 0x00020000);
  const unsigned w_step_bytes = (unsigned)(w_step_stride * 16);
  unsigned wvo[WPT];
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int f_ = tid + i * 256;
    const bool ok_ = NQ_IG3_ABL != 5 && (i + 1 < WPT || f_ < W_U4);
    const int fe_ = (ok_ || !WDMA) ? f_ : tid;   // (register-staged tiles: the value is never stored)
    const int pl_ = fe_ / (4 * MT), rem_ = fe_ - pl_ * (4 * MT);
    wvo[i] = (unsigned)(cot * 4 * MT + ((ok_ || WDMA) ? pl_ * (int)w_plane_stride + rem_ : 0)) * 16u;
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  if constexpr (WDMA) {   // steps 0..D-1 on their way before the patch loads (whose first use below waits for everything)
    steps3<0, WD>([&](auto q_c) { NQ3_DMA_W(decltype(q_c)::value) });
  }
  NQ3_LOAD_PATCH(0)
  if constexpr (!WDMA) {
    NQ3_LOAD_W(wvA, 0)
    NQ3_LOAD_W(wvB, 1)
  }
  if (it_neg) {   // first row of the tensor, left halo: the quad was loaded from offset 0 (see it_base)
    const f32x4 v_ = pv[0];
    if constexpr (PAD == 2) pv[0] = f32x4{0.f, 0.f, v_[0], v_[1]};
    else if constexpr (PAD == 1) pv[0] = f32x4{0.f, v_[0], v_[1], v_[2]};
  }
  NQ3_STORE_PATCH(patch0)
  if constexpr (WDMA) {
    NQ3_WAIT_VM(0)          // the three k-steps have landed (the patch conversion above waited for its own loads already)
    NQ3_BARRIER_LDS()
  } else {
  NQ3_STORE_W(wvA, wl0)
  __syncthreads();
  }

  // Register-staged weight pipeline (!WDMA): LDS buffer g & 1 holds step g, register set (g+1) & 1 holds step g+1 and is
  // published during step g, the loads of step g+2 are issued into the other set at the start of step g; a barrier after
  // every step.
  // one chunk; PAR = parity of its first global step
  auto run_chunk = [&](auto par_c, auto tail_c, int ch) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr int TK = decltype(tail_c)::value;   // tail kind (Conv3Args::tail), 0 = full chunk
    constexpr bool TAIL = TK != 0;
    constexpr int NSTC = nst_of_kind(TK);
    const int g0 = ch * NST;   // every chunk in front of this one is a full chunk
    steps3<0, NSTC>([&](auto st_c) {
      constexpr int st = decltype(st_c)::value;
      constexpr int gp = (PAR + st) & 1;  // parity of the global step
      const int g = g0 + st;
      // WDMA: k-step g+D into buffer (g+D) & (NB-1) = (g-1) & (NB-1), whose last readers passed the barrier that ended step g-1.  In
      // the last step of a full chunk the pieces are issued AFTER the patch conversion instead (below): the conversion is the
      // first use of ordinary loads, where hipcc drains the VM counter -- with the newest pieces not yet issued that costs
      // nothing that was not needed anyway.
      constexpr bool LATE_DMA = WDMA && !TAIL && st == NST - 1;
      if constexpr (WDMA && !LATE_DMA) NQ3_DMA_W(g + WD)
      if constexpr (!WDMA && NQ_IG3_ABL != 9 && NQ_IG3_ABL != 10) {   // ablations 9 / 10: the weights of the first steps for all
        if constexpr (gp == 0) {
          NQ3_LOAD_W(wvA, g + 2)
        } else {
          NQ3_LOAD_W(wvB, g + 2)
        }
      }
      if constexpr (!TAIL) {
        // The next chunk's 8 patch loads are SPREAD over the first NST - NQ_IG3_PSTD k-steps: a wave's loads return in
        // order, so the weight loads of the following steps queue behind whatever patch loads were issued before them
        // (past the last chunk of the tensor the channels are >= Cin: nothing is fetched)
        constexpr int LS = (NST > NQ_IG3_PSTD) ? NST - NQ_IG3_PSTD : 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (NQ_IG3_ABL != 8 && NQ_IG3_ABL != 10 && (NQ_IG3_SPREAD ? (j * LS) / 8 : LS - 1) == st) NQ3_LOAD_PATCH_J(ch + 1, j)
        }
      }
      bf16x8 bh[4], bl[4];
      if constexpr (TK == 1 || TK == 3) {
        // slots with HALF octets: a lane group's 8 k-values are (4 channels x tap A, 4 channels x tap B) = the low 8 bytes
        // of two patch units; a whole-octet slot of a 12-channel tail is the two halves of ONE unit.  All B fragments of
        // such a chunk are two 8-byte LDS reads at per-lane-group offsets (in 8-byte units, relative to the lane's pixel).
        constexpr auto tapoff = [](int tap) { return (tap < KK ? tap : KK - 1) / KS * PW + (tap < KK ? tap : KK - 1) % KS; };
        constexpr auto off0 = [tapoff](int j) {   // first 8 bytes of slot j
          if (TK == 1) return 2 * tapoff(2 * j);
          return j < KK ? 2 * tapoff(j) : 2 * (PP + tapoff(2 * (j - KK)));
        };
        constexpr auto off1 = [tapoff](int j) {   // second 8 bytes of slot j
          if (TK == 1) return 2 * tapoff(2 * j + 1);
          return j < KK ? 2 * tapoff(j) + 1 : 2 * (PP + tapoff(2 * (j - KK) + 1));
        };
        constexpr int j0 = 4 * st;
        constexpr int a0 = off0(j0), a1 = off0(j0 + 1), a2 = off0(j0 + 2), a3 = off0(j0 + 3);
        constexpr int b0 = off1(j0), b1 = off1(j0 + 1), b2 = off1(j0 + 2), b3 = off1(j0 + 3);
        const int o0 = (kq == 0) ? a0 : (kq == 1) ? a1 : (kq == 2) ? a2 : a3;
        const int o1 = (kq == 0) ? b0 : (kq == 1) ? b1 : (kq == 2) ? b2 : b3;
        const uint2* __restrict__ p2 = reinterpret_cast<const uint2*>(patch0) + 2 * ((2 * wave) * PW + l16);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          const int o = 2 * ((nb >> 1) * PW + (nb & 1) * 16);
          const uint2 h0 = p2[o0 + o], h1 = p2[o1 + o];
          const uint2 l0 = p2[4 * PP + o0 + o], l1 = p2[4 * PP + o1 + o];   // lo plane: 2 * PP units of 16 bytes further
          bh[nb] = __builtin_bit_cast(bf16x8, u32x4{h0.x, h0.y, h1.x, h1.y});
          bl[nb] = __builtin_bit_cast(bf16x8, u32x4{l0.x, l0.y, l1.x, l1.y});
        }
      } else {
      const u32x4* __restrict__ pbt;
      if constexpr (TAIL) {
        // one octet, four taps: lane group kq -> tap 4*st + kq (clamped; its weights are zero when padded)
        constexpr int t0 = 4 * st;
        constexpr int ta = t0 < KK ? t0 : KK - 1, tb = t0 + 1 < KK ? t0 + 1 : KK - 1, tc = t0 + 2 < KK ? t0 + 2 : KK - 1,
                      td = t0 + 3 < KK ? t0 + 3 : KK - 1;
        constexpr int oa = (ta / KS) * PW + (ta % KS), ob = (tb / KS) * PW + (tb % KS), oc = (tc / KS) * PW + (tc % KS),
                      od = (td / KS) * PW + (td % KS);
        const int off = (kq == 0) ? oa : (kq == 1) ? ob : (kq == 2) ? oc : od;
        pbt = patch0 + (2 * wave) * PW + l16 + off;
      } else {
        // taps of this step: even lane groups -> tap 2*st, odd -> tap 2*st+1 (clamped; its weights are zero when padded)
        constexpr int te = 2 * st, to_ = (2 * st + 1 < KK) ? 2 * st + 1 : KK - 1;
        constexpr int off_e = (te / KS) * PW + (te % KS), off_o = (to_ / KS) * PW + (to_ % KS);
        pbt = pb + (odd_tap ? off_o : off_e);
      }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int o = (NQ_IG3_ABL == 6) ? 0 : (nb >> 1) * PW + (nb & 1) * 16;   // ablation 6: one fragment pair for all four
        bh[nb] = __builtin_bit_cast(bf16x8, pbt[o]);
        bl[nb] = __builtin_bit_cast(bf16x8, pbt[2 * PP + o]);
      }
      }
      const u32x4* __restrict__ wb = wl0 + (g & WMASK) * WS + a_lane;
      bf16x8 ah0 = __builtin_bit_cast(bf16x8, wb[0]), al0 = __builtin_bit_cast(bf16x8, wb[4 * MT]);
      steps3<0, MI>([&](auto mi_c) {
        constexpr int mi = decltype(mi_c)::value;
        bf16x8 ah = ah0, al = al0;
        if constexpr (mi + 1 < MI) {  // prefetch the next channel block's fragments
          ah0 = __builtin_bit_cast(bf16x8, wb[(mi + 1) * 16]);
          al0 = __builtin_bit_cast(bf16x8, wb[4 * MT + (mi + 1) * 16]);
        }
        // product-major order: the three MFMAs that accumulate into one register are 4 issues apart (a dependent MFMA
        // issued back to back waits for its predecessor's result: the compiler otherwise chains them through a temporary)
        if constexpr (NQ_IG3_ABL == 3) {   // keep the fragment reads alive without the matrix pipe
#pragma unroll
          for (int nb = 0; nb < 4; ++nb) {
            const u32x4 t0 = __builtin_bit_cast(u32x4, ah), t1 = __builtin_bit_cast(u32x4, bh[nb]), t2 = __builtin_bit_cast(u32x4, bl[nb]),
                        t3 = __builtin_bit_cast(u32x4, al);
            acc[mi][nb][0] += __builtin_bit_cast(float, t0[0] ^ t1[0] ^ t2[0] ^ t3[0]);
          }
        } else {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nb], acc[mi][nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nb], acc[mi][nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nb], acc[mi][nb], 0, 0, 0);
        }
        if constexpr (mi == (MI - 1) / 2) {
          // publish the next weights in the MIDDLE of the MFMA block (the LDS write latency is covered by the remaining
          // MFMAs instead of sitting in front of the barrier)
          if constexpr (NQ_IG3_ABL == 9 || NQ_IG3_ABL == 10 || WDMA) {
          } else {              // step g+1 (loaded one step ago) from set (g+1) & 1
            if (g + 1 < G) {
              u32x4* wdst = wl0 + (gp ^ 1) * W_U4;
              if constexpr (gp == 0) {
                NQ3_STORE_W(wvB, wdst)
              } else {
                NQ3_STORE_W(wvA, wdst)
              }
            }
          }
          if constexpr (!WDMA) __builtin_amdgcn_sched_barrier(0);
        }
      });
      if constexpr (WDMA) {
        // End of step g: this wave's pieces of step g+1 must have landed before the barrier; every wave's, after it -- the
        // fragments of step g+1 are read one barrier AFTER the wait that retires them.  In flight behind them: steps g+2 ..
        // g+D ((D-1) x WPT pieces; patch loads issued in between only make the count conservative).
        if constexpr (LATE_DMA) {
          NQ3_WAIT_VM((WD - 2) * WPT)   // only steps g+2 .. g+D-1 behind step g+1 (the chunk's patch loads are older than all)
          NQ3_BARRIER_LDS()       // every wave is done with the current patch
          if (ch + 1 < nchunk) {
            NQ3_STORE_PATCH(patch0)
          }
          NQ3_DMA_W(g + WD)
          NQ3_BARRIER_LDS()
        } else if constexpr (NQ_IG3_ABL == 11 && (st & 1) == 0) {   // timing only: no barrier after even steps (races)
          NQ3_WAIT_VM((WD - 1) * WPT)
        } else {
          NQ3_WAIT_VM((WD - 1) * WPT)
          NQ3_BARRIER_LDS()
        }
      } else {
      if constexpr (!TAIL && st == NST - 1) {
        if (NQ_IG3_ABL != 8 && NQ_IG3_ABL != 10 && ch + 1 < nchunk) {   // ablations 8 / 10: every chunk re-uses the first patch
          __syncthreads();  // every wave is done with the current patch
          NQ3_STORE_PATCH(patch0)
        }
      }
      __syncthreads();
      }
    });
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  for (int ch = 0; ch < nfull; ch += 2) {
    run_chunk(C0{}, C0{}, ch);
    if (ch + 1 < nfull) run_chunk(std::integral_constant<int, (NST & 1)>{}, C0{}, ch + 1);
  }
  if (tail) {   // parity of its first global step = parity of nfull * NST
    const bool odd = ((nfull & NST) & 1) != 0;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;
    // (the half-octet kinds on the 80-channel tile too since the weights travel by LDS-DMA: with the two weight register sets
    // gone the kernel holds 252 VGPRs without scratch; with them these kinds spilled into their k-steps, 387 -> 476 us)
    {
      if (tail == 1) {
        if (odd) run_chunk(C1{}, K1{}, nfull);
        else run_chunk(C0{}, K1{}, nfull);
      } else if (tail == 3) {
        if (odd) run_chunk(C1{}, K3{}, nfull);
        else run_chunk(C0{}, K3{}, nfull);
      }
    }
    if (tail == 2) {
      if (odd) run_chunk(C1{}, K2{}, nfull);
      else run_chunk(C0{}, K2{}, nfull);
    }
  }
  if constexpr (WDMA) {   // the pieces issued past the end of the split must not land in LDS the epilogue re-uses
    NQ3_WAIT_VM(0)
    NQ3_BARRIER_LDS()
  }
#undef NQ3_DMA_W
#undef NQ3_DMA_PIECE
#undef NQ3_WAIT_VM
#undef NQ3_BARRIER_LDS
#undef NQ3_LOAD_PATCH
#undef NQ3_STORE_PATCH
#undef NQ3_LOAD_W
#undef NQ3_STORE_W

  // ---- epilogue (same element mapping as conv_igemm_impl.h; nb = (row 2w + nb/2, column half nb%2)) ----
  if constexpr (NQ_IG3_ABL == 4) {   // no epilogue traffic: one conditional store keeps the accumulators alive
    float t = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) t += (acc[mi][nb][0] + acc[mi][nb][1]) + (acc[mi][nb][2] + acc[mi][nb][3]);
    if (t == 1.2345e-30f) a.y[tid] = t;
    return;
  }
  const int Cout = a.Cout, r = a.r, rr = a.r * a.r;
  const int cob = co0 + 4 * kq;
  if (a.nsplit > 1) {  // raw partial sums of this split; bias / activation happen in the finish kernel
    float* __restrict__ ys = a.slab + ((int64_t)split * a.B + b) * Cout * HW;
    steps3<0, MI>([&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int py = y0 + 2 * wave + (nb >> 1), px = x0 + (nb & 1) * 16 + l16;
        if (py >= H || px >= W) continue;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int cc = cob + mi * 16 + reg;
          if (cc < Cout) ys[(int64_t)cc * HW + (int64_t)py * W + px] = acc[mi][nb][reg];
        }
      }
    });
    return;
  }
  const int epi = a.epi;
  if (epi == NQ_EPI_DGRAD_GELU && a.lds_epi && (W & 3) == 0) {
    // Wide variant (the epilogue is bound by the NUMBER of global load/store instructions, not by bytes): every wave
    // transposes its 2 x 32 pixel x MT channel tile through its own LDS region so that a lane then owns 4 consecutive
    // pixels of one channel: ONE 16-byte load of gelu' and, for r = 2, TWO 8-byte stores (even / odd pixels go to
    // different un-shuffled planes) per 4 outputs, instead of 4 loads + 4 stores.  LDS is free here: the K loop ended
    // with a barrier.  Region: [MT][2 rows][32 px] floats = MT*256 B per wave.
    float* const stage = reinterpret_cast<float*>(smem) + wave * (MT * 64);
    steps3<0, MI>([&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int cl = mi * 16 + 4 * kq + reg;
        const float bv = (a.bias && co0 + cl < Cout) ? a.bias[co0 + cl] : 0.f;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) stage[cl * 64 + (nb >> 1) * 32 + (nb & 1) * 16 + l16] = acc[mi][nb][reg] + bv;
      }
    });
    const int Wo = W / r;
    const int plane = (int)(HW / rr);
#pragma unroll
    for (int it = 0; it < MT / 4; ++it) {
      const int e = it * 64 + lane;
      const int q = e & 7, row = (e >> 3) & 1, cl = e >> 4;
      const int cc = co0 + cl, py = y0 + 2 * wave + row, px0 = x0 + q * 4;
      if (cc >= Cout || py >= H || px0 >= W) continue;   // W % 4 == 0: a quad is inside or outside as a whole
      f32x4 v = *reinterpret_cast<const f32x4*>(stage + cl * 64 + row * 32 + q * 4);
      const int64_t cbase = ((int64_t)b * Cout + cc) * HW;
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(a.zprev + cbase + (int64_t)py * W + px0);
      v = v * d4;
      if (a.y_split) v = f32x4{nq_split_word_f(v[0]), nq_split_word_f(v[1]), nq_split_word_f(v[2]), nq_split_word_f(v[3])};
      float* __restrict__ yo = a.y + cbase;
      if (r == 2) {
        const int o = ((py & 1) * 2) * plane + (py >> 1) * Wo + (px0 >> 1);
        *reinterpret_cast<float2*>(yo + o) = make_float2(v[0], v[2]);
        *reinterpret_cast<float2*>(yo + o + plane) = make_float2(v[1], v[3]);
      } else if (r == 4) {
        const int o = ((py & 3) * 4) * plane + (py >> 2) * Wo + (px0 >> 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) yo[o + j * plane] = v[j];
      } else {
        const int yq = py / r;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int px = px0 + j, xq = px / r;
          yo[((py - yq * r) * r + (px - xq * r)) * plane + yq * Wo + xq] = v[j];
        }
      }
    }
    return;
  }
  if (epi == NQ_EPI_DGRAD_GELU) {
    // data gradient w.r.t. the pre-activation below: acc * gelu'(z) (zprev = the derivative saved by the forward
    // epilogue, same NCHW layout as this conv's output), stored un-shuffled: channel c*r*r + (y%r)*r + x%r at
    // (y/r, x/r).  Both offsets split into a per-channel base (b*Cout + c)*H*W and a per-pixel part.
    const int Wo = W / r;
    const int plane = (int)(HW / rr);
    int in_off[4], out_off[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int py = y0 + 2 * wave + (nb >> 1), px = x0 + (nb & 1) * 16 + l16;
      const int yq = py / r, xq = px / r;
      in_off[nb] = (py < H && px < W) ? py * W + px : -1;
      out_off[nb] = ((py - yq * r) * r + (px - xq * r)) * plane + yq * Wo + xq;
    }
    steps3<0, MI>([&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int cc = cob + mi * 16 + reg;
        if (cc >= Cout) continue;
        const float bv = a.bias ? a.bias[cc] : 0.f;
        const int64_t cbase = ((int64_t)b * Cout + cc) * HW;
        const float* __restrict__ zp = a.zprev + cbase;
        float* __restrict__ yo = a.y + cbase;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          if (in_off[nb] >= 0) {
            const float v = (acc[mi][nb][reg] + bv) * zp[in_off[nb]];
            yo[out_off[nb]] = a.y_split ? nq_split_word_f(v) : v;
          }
      }
    });
    return;
  }
  if ((epi == NQ_EPI_PS || epi == NQ_EPI_PS_GELU) && r == 2 && (W & 1) == 0) {
    // PixelShuffle(2) with 16-byte stores.  A lane's 4 registers are the 2x2 output block of its pixel; the even lane of
    // a pixel pair takes both upper rows, the odd lane both lower rows (one quad_perm DPP exchange per value), so each
    // lane issues ONE dwordx4 store per tensor instead of two dwordx2: the epilogue is bound by the number of store
    // instructions, not by bytes (measured: time scales with the instruction count at equal bytes).
    const bool want_act = (epi == NQ_EPI_PS_GELU);
    const bool odd = (lane & 1) != 0;
    const int C = Cout >> 2;
    const int64_t W2 = (int64_t)W * 2;
    auto xchg = [](float v) {
      return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    };
    steps3<0, MI>([&](auto mi_c) {
      constexpr int mi = decltype(mi_c)::value;
      const int co = cob + mi * 16;  // first of the lane's 4 channels (multiple of 4); the pair partner has the same co
      const bool cok = co < Cout;
      const float4 bv = (a.bias && cok) ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        const int py = y0 + 2 * wave + (nb >> 1), px = x0 + (nb & 1) * 16 + l16;
        const bool valid = cok && py < H && px < W;   // W even: both lanes of a pair agree
        const f32x4 v4 = acc[mi][nb];
        const float v0 = v4[0] + bv.x, v1 = v4[1] + bv.y, v2 = v4[2] + bv.z, v3 = v4[3] + bv.w;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f, d0 = v0, d1 = v1, d2 = v2, d3 = v3;  // PS: z = conv; PS_GELU: z = gelu'
        if (want_act) {
          nq_gelu_pair(v0, g0, d0); nq_gelu_pair(v1, g1, d1); nq_gelu_pair(v2, g2, d2); nq_gelu_pair(v3, g3, d3);
          if (a.y_split) { g0 = nq_split_word_f(g0); g1 = nq_split_word_f(g1); g2 = nq_split_word_f(g2); g3 = nq_split_word_f(g3); }
        }
        const int64_t o = (((int64_t)b * C + (co >> 2)) * (2 * H) + 2 * py + (odd ? 1 : 0)) * W2 + 2 * (px - (odd ? 1 : 0));
        {
          const float r0 = xchg(odd ? d0 : d2), r1 = xchg(odd ? d1 : d3);
          const float4 out = odd ? make_float4(r0, r1, d2, d3) : make_float4(d0, d1, r0, r1);
          if (valid) *reinterpret_cast<float4*>(a.z + o) = out;
        }
        if (want_act) {
          const float r0 = xchg(odd ? g0 : g2), r1 = xchg(odd ? g1 : g3);
          const float4 out = odd ? make_float4(r0, r1, g2, g3) : make_float4(g0, g1, r0, r1);
          if (valid) *reinterpret_cast<float4*>(a.y + o) = out;
        }
      }
    });
    return;
  }
  steps3<0, MI>([&](auto mi_c) {
    constexpr int mi = decltype(mi_c)::value;
    const int co = cob + mi * 16;  // first of the lane's 4 channels (multiple of 4)
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
      const int py = y0 + 2 * wave + (nb >> 1), px = x0 + (nb & 1) * 16 + l16;
      if (py >= H || px >= W) continue;
      const f32x4 v4 = acc[mi][nb];
      if ((epi == NQ_EPI_PS || epi == NQ_EPI_PS_GELU) && (r == 2 || r == 4)) {
        if (co >= Cout) continue;  // Cout % 4 == 0 here
        const float4 bv = a.bias ? *reinterpret_cast<const float4*>(a.bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float v0 = v4[0] + bv.x, v1 = v4[1] + bv.y, v2 = v4[2] + bv.z, v3 = v4[3] + bv.w;
        const int C = Cout / rr, c = co / rr;
        const int si = (r == 4) ? ((co >> 2) & 3) : 0;
        const int64_t rowbase = (((int64_t)b * C + c) * (H * r) + (int64_t)py * r + si) * ((int64_t)W * r);
        const bool want_act = (epi == NQ_EPI_PS_GELU);
        float g0, g1, g2, g3, d0 = v0, d1 = v1, d2 = v2, d3 = v3;  // PS: z = conv; PS_GELU: z = gelu'(conv)
        if (want_act) {
          nq_gelu_pair(v0, g0, d0); nq_gelu_pair(v1, g1, d1); nq_gelu_pair(v2, g2, d2); nq_gelu_pair(v3, g3, d3);
          if (a.y_split) { g0 = nq_split_word_f(g0); g1 = nq_split_word_f(g1); g2 = nq_split_word_f(g2); g3 = nq_split_word_f(g3); }
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
        continue;
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int cc = co + reg;
        if (cc >= Cout) continue;
        float v = v4[reg] + (a.bias ? a.bias[cc] : 0.f);
        if (epi == NQ_EPI_PS_GELU || epi == NQ_EPI_PS) {
          const int C = Cout / rr, c = cc / rr, rem = cc - c * rr, si = rem / r, sj = rem - si * r;
          const int64_t o = (((int64_t)b * C + c) * (H * r) + (int64_t)py * r + si) * ((int64_t)W * r) + (int64_t)px * r + sj;
          if (epi == NQ_EPI_PS_GELU) {
            float gv, dv;
            nq_gelu_pair(v, gv, dv);
            a.y[o] = a.y_split ? nq_split_word_f(gv) : gv;
            a.z[o] = dv;
          } else {
            a.z[o] = v;
          }
        } else {
          const int64_t o = ((int64_t)b * Cout + cc) * HW + (int64_t)py * W + px;
          const float ov = (epi == NQ_EPI_TANH) ? tanhf(v) * 0.5f + 0.5f : v;
          a.y[o] = a.y_split ? nq_split_word_f(ov) : ov;
        }
      }
    }
  });
}

template <int MI, bool XS = false>
int launch_igemm3(const Conv3Args& a_in, int tiles, hipStream_t st) {
  constexpr int MT = 16 * MI;
  size_t lds = (size_t)(PATCH_U4 + 2 * 2 * 4 * MT) * 16;   // patch + two weight buffers (register-staged tiles)
  if (MI >= 3) lds = (size_t)(PATCH_U4 + 4 * (2 * 4 * MT)) * 16;  // DMA ring of 4
  Conv3Args a = a_in;
  a.lds_epi = 0;
  if (a.epi == NQ_EPI_DGRAD_GELU && a.nsplit == 1 && MI <= 4) {   // 4 waves x [MT][64] floats for the wide epilogue
    a.lds_epi = 1;
    if (lds < (size_t)MT * 1024) lds = (size_t)MT * 1024;
  }
  a.tiles = tiles;
  if constexpr (MI <= 3) {
    // short K loop -> the three-waves-per-SIMD build (NQ_IG3_OCC3_STEPS: largest k-step count that takes it; 0 = never)
    static const int occ3_steps = [] { const char* e = getenv("NQ_IG3_OCC3_STEPS"); return e ? atoi(e) : NQ_IG3_OCC3_DEFAULT; }();
    const int ksteps = (a.nsplit > 1 ? a.per_split : a.nchunk - 1) * NST + (a.nsplit > 1 ? 0 : nst_of_kind(a.tail));
    if (ksteps <= occ3_steps) {
      if (int rc = nq_lds_optin<&conv_igemm3_kernel<MI, 3, XS>>(lds)) return rc;
      hipLaunchKernelGGL((conv_igemm3_kernel<MI, 3, XS>), dim3((unsigned)(tiles * a.co_tiles * a.B * a.nsplit)), dim3(256), lds, st, a);
      return nq_launch_status();
    }
  }
  if (int rc = nq_lds_optin<&conv_igemm3_kernel<MI, 2, XS>>(lds)) return rc;
  hipLaunchKernelGGL((conv_igemm3_kernel<MI, 2, XS>), dim3((unsigned)(tiles * a.co_tiles * a.B * a.nsplit)), dim3(256), lds, st, a);
  return nq_launch_status();
}

}  // namespace

#define NQ_CAT2(a, b) a##b
#define NQ_CAT(a, b) NQ_CAT2(a, b)

extern "C" int NQ_CAT(nq_conv_igemm3_k, NQ_KS)(const float* x, const void* wt3, const float* bias, float* y, float* z,
                                                const float* zprev, int B, int Cin, int H, int W, int Cout, int r, int epi,
                                                int mi_sel, int nsplit, int per_split, float* slab, hipStream_t st) {
  if ((int64_t)B * Cin * H * W * 4 >= 0xFFFFFF00ll) return NQ_ERR_UNSUPPORTED;   // 32-bit buffer offsets into x
  Conv3Args a;
  a.nsplit = nsplit; a.per_split = per_split; a.slab = slab;
  a.x = x; a.wt3 = wt3; a.bias = bias; a.y = y; a.z = z; a.zprev = zprev;
  // bits 8 / 9 of `epi`: x holds / y is written as split {hi | lo} words (NQ_EPI_X_SPLIT / NQ_EPI_Y_SPLIT, include/nq_hip.h)
  const bool xs = (epi & 0x100) != 0;
  a.y_split = (epi & 0x200) ? 1 : 0;
  epi &= 0xFF;
  if (a.y_split && nsplit > 1) return NQ_ERR_UNSUPPORTED;   // (the slabs are finished by another kernel)
  a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.r = r; a.epi = epi;
  a.tiles_x = (W + TW - 1) / TW;
  a.nchunk = (Cin + CC - 1) / CC;
  a.tail = nq_conv3_tail_kind(Cin, mi_sel);
  a.co_tiles = (Cout + 16 * mi_sel - 1) / (16 * mi_sel);
  a.w_bytes = (unsigned)(((int64_t)(a.nchunk - 1) * NST + nst_of_kind(a.tail)) * 2 * a.co_tiles * 4 * (16 * mi_sel) * 16);
  const int tiles = a.tiles_x * ((H + TH - 1) / TH);
  if (xs) {   // (the 16-channel tile is not offered with split input: no layer of the shipped models pairs them)
    switch (mi_sel) {
      case 2: return launch_igemm3<2, true>(a, tiles, st);
      case 3: return launch_igemm3<3, true>(a, tiles, st);
      case 4: return launch_igemm3<4, true>(a, tiles, st);
      case 5: return launch_igemm3<5, true>(a, tiles, st);
      default: return NQ_ERR_UNSUPPORTED;
    }
  }
  switch (mi_sel) {
    case 1: return launch_igemm3<1>(a, tiles, st);
    case 2: return launch_igemm3<2>(a, tiles, st);
    case 3: return launch_igemm3<3>(a, tiles, st);
    case 4: return launch_igemm3<4>(a, tiles, st);
    case 5: return launch_igemm3<5>(a, tiles, st);
    default: return NQ_ERR_UNSUPPORTED;
  }
}

// steps per chunk for this kernel size (used by nq_weight_layout3)
extern "C" int NQ_CAT(nq_conv3_nst_k, NQ_KS)() { return NST; }
extern "C" int NQ_CAT(nq_conv3_nst8_k, NQ_KS)() { return NST8; }
extern "C" int NQ_CAT(nq_conv3_nstk_k, NQ_KS)(int kind) { return nst_of_kind(kind); }
