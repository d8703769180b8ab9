// LDS images of the K / V tiles of the attention kernel (attention.hip): 64-key tiles of D-wide bf16 rows,
// filled by LDS-DMA (lane-linear destination, so the bank swizzle is applied to the per-lane SOURCE chunk and again on the read).
#pragma once

template <int D> struct Lay;
template <> struct Lay<64> {
  static constexpr int ROWB = 128;
  // K tile: 32x32 row reads (ds_read_b128) are conflict-free with chunk ^= (row>>1)&7 on 128-B rows
  __device__ static int kchunk(int row, int ch) { return ch ^ ((row >> 1) & 7); }
  // V tile: transposed reads take 4 consecutive keys x 64 B per 32-lane half
  __device__ static int vchunk(int row, int ch) { return ch ^ (((row >> 1) & 1) << 2); }
};
template <> struct Lay<128> {
  static constexpr int ROWB = 256;
  __device__ static int kchunk(int row, int ch) { return ch ^ (row & 15); }
  __device__ static int vchunk(int row, int ch) { return ch ^ (((row & 3) << 2) | ((row >> 2) & 3)); }
};
