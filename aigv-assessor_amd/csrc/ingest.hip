// Frame ingest: uint8 HWC RGB frames, already resized to the model resolution on the host,
// -> normalised bf16 NCHW `pixel_values`.  Restates torchvision's ToTensor + Normalize + the call-site bf16 cast of the
// reference's eval transform (internvl/train/dataset.py:267-274 build_transform(is_train=False), constants.py:10-11,
// stage2_eval.py:484-485,932) with the same fp32 operation order: (u / 255 - mean) / std, one bf16 rounding.
// HBM-bound byte mover: 12 contiguous input bytes (4 pixels) per thread, one 8-byte store per channel plane.
#include "common.h"
#include "kernels.h"

namespace {

struct IngestArgs {
  const uint8_t* in;   // [F, H, W, 3]
  bf16_t* out;         // [F, 3, H, W]
  long quads;          // F*H*W/4
  int hw;              // H*W
  float mean[3], std[3];
};

__global__ __launch_bounds__(256) void frame_ingest_kernel(const IngestArgs a) {
  const int hw4 = a.hw >> 2;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < a.quads; q += (long)gridDim.x * blockDim.x) {
    const long f = q / hw4;
    const int p4 = (int)(q - f * hw4);          // pixel quad inside the frame
    const uint32_t* src = (const uint32_t*)(a.in + (size_t)q * 12);
    const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
    uint8_t b[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) { b[i] = (w0 >> (8 * i)) & 255; b[4 + i] = (w1 >> (8 * i)) & 255; b[8 + i] = (w2 >> (8 * i)) & 255; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      u16x4 o;
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const float x = __fdiv_rn((float)b[px * 3 + c], 255.0f);       // ToTensor
        o[px] = f2bf(__fdiv_rn(x - a.mean[c], a.std[c]));               // Normalize, then .to(bfloat16)
      }
      *(u16x4*)(a.out + ((size_t)f * 3 + c) * a.hw + (size_t)p4 * 4) = o;
    }
  }
}

}  // namespace

hipError_t aigv_launch_frame_ingest(const uint8_t* hwc, int n_frames, int H, int W, const float* mean, const float* stdv,
                                    bf16_t* out, hipStream_t s) {
  if (n_frames <= 0) return hipSuccess;
  if ((H * W) % 4 || !hwc || !out || !mean || !stdv) return hipErrorInvalidValue;
  IngestArgs a{};
  a.in = hwc; a.out = out; a.hw = H * W; a.quads = (long)n_frames * H * W / 4;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.std[c] = stdv[c]; }
  const long blocks = (a.quads + 255) / 256;
  hipLaunchKernelGGL(frame_ingest_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, a);
  return hipGetLastError();
}
