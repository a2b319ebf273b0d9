// K order of the pre-split bf16x3 weight operand (nq_weight_layout3, conv3.hip), shared by the kernels that read it
// (conv_igemm3_impl.h mirrors it with compile-time offsets; conv_flat3.hip evaluates it at run time).
#pragma once
#include <hip/hip_runtime.h>

// k-value e (0..7) of lane group kq at k-step s of a 16-channel chunk -> (channel within the chunk, tap); tap >= KK: zero weight.
//   kind 0 full chunk : 2 octets x 2 taps per step          ch = (kq&1)*8 + e, tap = 2s + (kq>>1)
//   kind 2 <= 8 ch    : 1 octet x 4 taps per step           ch = e,            tap = 4s + kq
//   kind 1 <= 4 ch    : slot j = 4s + kq = 4 channels x taps (2j, 2j+1)        ch = e & 3, tap = 2j + (e >> 2)
//   kind 3 <= 12 ch   : slots j < KK: octet 0 at tap j; slots KK + h: channels 8..11 x taps (2h, 2h+1)
__host__ __device__ __forceinline__ void wl3_elem(int kind, int KK, int s, int kq, int e, int& ch, int& tap) {
  const int j = 4 * s + kq;
  if (kind == 2) {
    ch = e;
    tap = j;
  } else if (kind == 1) {
    ch = e & 3;
    tap = 2 * j + (e >> 2);
  } else if (kind == 3) {
    if (j < KK) {
      ch = e;
      tap = j;
    } else {
      ch = 8 + (e & 3);
      tap = 2 * (j - KK) + (e >> 2);
    }
  } else {
    ch = (kq & 1) * 8 + e;
    tap = 2 * s + (kq >> 1);
  }
}
