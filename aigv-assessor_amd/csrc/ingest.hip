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

// ---- frame resize + ingest (the part in front of the kernel above) ---------------------------------------
// uint8 HWC RGB frames at the video's resolution -> Pillow's BICUBIC resize to the model resolution -> ToTensor / Normalize /
// bf16 NCHW.  The reference resizes each sampled frame with PIL `image.resize((448, 448))` (dynamic_preprocess with max_num = 1,
// internvl/train/dataset.py:702-738; stage2_eval.py:453-456); `T.Resize` of build_transform is then an identity.  Pillow's
// ImagingResample for 8-bit images (src/libImaging/Resample.c) is integer arithmetic with 22-bit fixed-point coefficients and a
// uint8 intermediate between the horizontal and the vertical pass; both passes are restated here bit-exactly (oracle/resize.py
// is the CPU restatement, pinned byte-for-byte against PIL).  Coefficient tables are computed on the host in double precision
// exactly as precompute_coeffs / normalize_coeffs_8bpc do and cached on the device per (input size, output size).
// HBM-bound byte movers: one thread per output pixel, 3 channels.
#include <map>
#include <math.h>
#include <mutex>
#include <vector>

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ int clip8(int acc) {
  const int v = acc >> PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: in [F, in_h, in_w, 3] rows y_first .. y_first + rows - 1 -> tmp [F, rows, out_w, 3]
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ tmp, int in_h, int in_w,
                                                       int out_w, int rows, int y_first, const int* __restrict__ bounds,
                                                       const int* __restrict__ kk, int ksize, long total) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % out_w);
    const long fy = i / out_w;
    const int y = (int)(fy % rows);
    const long f = fy / rows;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = kk + (size_t)xx * ksize;
    const uint8_t* p = in + ((size_t)(f * in_h + y_first + y) * in_w + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
      const int c = k[x];
      s0 += (int)p[3 * x] * c;
      s1 += (int)p[3 * x + 1] * c;
      s2 += (int)p[3 * x + 2] * c;
    }
    uint8_t* o = tmp + (size_t)i * 3;
    o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
  }
}

struct ResizeVArgs {
  const uint8_t* tmp;   // [F, rows, out_w, 3]
  uint8_t* out_u8;      // [F, out_h, out_w, 3] or null
  bf16_t* out_nchw;     // [F, 3, out_h, out_w] or null
  const int* bounds;    // vertical, already shifted by the first row of tmp
  const int* kk;
  int ksize, rows, out_h, out_w;
  long total;           // F * out_h * out_w
  float mean[3], std[3];
};

// vertical pass + ToTensor / Normalize / bf16 (the same fp32 operation order as frame_ingest_kernel)
__global__ __launch_bounds__(256) void resize_v_kernel(const ResizeVArgs a) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < a.total; i += (long)gridDim.x * blockDim.x) {
    const int xx = (int)(i % a.out_w);
    const long fy = i / a.out_w;
    const int yy = (int)(fy % a.out_h);
    const long f = fy / a.out_h;
    const int ymin = a.bounds[2 * yy], n = a.bounds[2 * yy + 1];
    const int* k = a.kk + (size_t)yy * a.ksize;
    const size_t stride = (size_t)a.out_w * 3;
    const uint8_t* p = a.tmp + ((size_t)(f * a.rows + ymin) * a.out_w + xx) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
      const int c = k[y];
      s0 += (int)p[0] * c;
      s1 += (int)p[1] * c;
      s2 += (int)p[2] * c;
      p += stride;
    }
    const int u[3] = {clip8(s0), clip8(s1), clip8(s2)};
    if (a.out_u8) {
      uint8_t* o = a.out_u8 + (size_t)i * 3;
      o[0] = (uint8_t)u[0]; o[1] = (uint8_t)u[1]; o[2] = (uint8_t)u[2];
    }
    if (a.out_nchw) {
      const size_t plane = (size_t)a.out_h * a.out_w;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float x = __fdiv_rn((float)u[c], 255.0f);
        a.out_nchw[((size_t)f * 3 + c) * plane + (size_t)yy * a.out_w + xx] = f2bf(__fdiv_rn(x - a.mean[c], a.std[c]));
      }
    }
  }
}

// Fast forms (the generic kernels above stay as the fallback for odd sizes).
// Horizontal: one source row per workgroup iteration staged in LDS with coalesced dword loads (the per-pixel taps are 39-57
// contiguous bytes at an arbitrary byte offset: from global memory that is one byte load per tap and lane), then every thread
// produces output pixels from LDS.  `in_end` = one past the last input byte: the staging never reads beyond it.
__global__ __launch_bounds__(256) void resize_h_lds_kernel(const uint8_t* __restrict__ in, const uint8_t* in_end, uint8_t* __restrict__ tmp,
                                                           int in_h, int in_w, int out_w, int rows, int y_first,
                                                           const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                           long n_rows) {
  extern __shared__ __attribute__((aligned(16))) uint8_t srow[];
  const int row_bytes = in_w * 3;
  for (long r = blockIdx.x; r < n_rows; r += gridDim.x) {
    const long f = r / rows;
    const int y = (int)(r - f * rows);
    const uint8_t* src = in + (size_t)(f * in_h + y_first + y) * row_bytes;
    const int mis = (int)((uintptr_t)src & 3);                      // stage from the dword-aligned address below src
    const uint32_t* base = (const uint32_t*)(src - mis);
    const int n_dw = (mis + row_bytes + 3) >> 2;
    for (int i = threadIdx.x; i < n_dw; i += 256) {
      const uint8_t* a = (const uint8_t*)(base + i);
      uint32_t v;
      if (a + 4 <= in_end) v = base[i];
      else { v = 0; for (int j = 0; j < 4 && a + j < in_end; ++j) v |= (uint32_t)a[j] << (8 * j); }
      ((uint32_t*)srow)[i] = v;
    }
    __syncthreads();
    uint8_t* orow = tmp + (size_t)r * out_w * 3;
    for (int xx = threadIdx.x; xx < out_w; xx += 256) {
      const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
      const int* k = kk + (size_t)xx * ksize;
      const uint8_t* p = srow + mis + xmin * 3;
      int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
      for (int x = 0; x < n; ++x) {
        const int c = k[x];
        s0 += (int)p[3 * x] * c;
        s1 += (int)p[3 * x + 1] * c;
        s2 += (int)p[3 * x + 2] * c;
      }
      orow[3 * xx] = (uint8_t)clip8(s0); orow[3 * xx + 1] = (uint8_t)clip8(s1); orow[3 * xx + 2] = (uint8_t)clip8(s2);
    }
    __syncthreads();
  }
}

// Vertical: a thread owns 4 consecutive BYTES of an output row (rows are flat [out_w * 3] byte arrays, out_w * 3 % 4 == 0): one
// dword load per tap instead of three byte loads per pixel, four accumulators.
__global__ __launch_bounds__(256) void resize_v_dword_kernel(const ResizeVArgs a) {
  const int rb = a.out_w * 3, dw_per_row = rb >> 2;
  const long total_dw = a.total * 3 / 4;      // F * out_h * dw_per_row
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total_dw; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % dw_per_row);
    const long fy = i / dw_per_row;
    const int yy = (int)(fy % a.out_h);
    const long f = fy / a.out_h;
    const int ymin = a.bounds[2 * yy], n = a.bounds[2 * yy + 1];
    const int* k = a.kk + (size_t)yy * a.ksize;
    const uint8_t* p = a.tmp + (size_t)(f * a.rows + ymin) * rb + (size_t)t * 4;
    int s[4] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
    for (int y = 0; y < n; ++y) {
      const uint32_t v = *(const uint32_t*)p;
      const int c = k[y];
      s[0] += (int)(v & 255) * c;
      s[1] += (int)((v >> 8) & 255) * c;
      s[2] += (int)((v >> 16) & 255) * c;
      s[3] += (int)(v >> 24) * c;
      p += rb;
    }
    int u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = clip8(s[j]);
    if (a.out_u8) *(uint32_t*)(a.out_u8 + (size_t)(f * a.out_h + yy) * rb + (size_t)t * 4) = (uint32_t)u[0] | ((uint32_t)u[1] << 8) | ((uint32_t)u[2] << 16) | ((uint32_t)u[3] << 24);
    if (a.out_nchw) {
      const size_t plane = (size_t)a.out_h * a.out_w;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int b = t * 4 + j, xx = b / 3, c = b - xx * 3;
        const float x = __fdiv_rn((float)u[j], 255.0f);
        a.out_nchw[((size_t)f * 3 + c) * plane + (size_t)yy * a.out_w + xx] = f2bf(__fdiv_rn(x - a.mean[c], a.std[c]));
      }
    }
  }
}

double bicubic_filter(double x) {   // Resample.c bicubic_filter, a = -0.5
  const double a = -0.5;
  if (x < 0.0) x = -x;
  if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
  if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
  return 0.0;
}

struct AxisTable { int ksize = 0; std::vector<int> bounds, kk; };

// Resample.c precompute_coeffs (box = the whole axis) followed by normalize_coeffs_8bpc, operation for operation
AxisTable precompute_axis(int in_size, int out_size) {
  AxisTable t;
  const float in0 = 0.0f, in1 = (float)in_size;
  double scale = (double)(in1 - in0) / out_size, filterscale = scale;
  if (filterscale < 1.0) filterscale = 1.0;
  const double support = 2.0 * filterscale;
  t.ksize = (int)ceil(support) * 2 + 1;
  t.bounds.assign((size_t)out_size * 2, 0);
  t.kk.assign((size_t)out_size * t.ksize, 0);
  std::vector<double> k((size_t)t.ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = in0 + (xx + 0.5) * scale;
    double ww = 0.0;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < xmax; ++x) {
      const double w = bicubic_filter((x + xmin - center + 0.5) * ss);
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      t.kk[(size_t)xx * t.ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << PRECISION_BITS));
    }
    t.bounds[2 * xx] = xmin;
    t.bounds[2 * xx + 1] = xmax;
  }
  return t;
}

struct DeviceTables {
  int *bounds_h = nullptr, *kk_h = nullptr, *bounds_v = nullptr, *kk_v = nullptr;
  int ksize_h = 0, ksize_v = 0, y_first = 0, rows = 0;
};
// keyed by (device, (in_h, in_w), (out_h, out_w)): a table lives in the memory of the device it was built on; process lifetime,
// written once under the lock and never regrown (a cached entry is immutable, so launches on any stream may share it)
std::map<std::pair<int, std::pair<long long, long long>>, DeviceTables> g_resize_tables;
std::mutex g_resize_mutex;

hipError_t upload(const std::vector<int>& v, int** out) {
  hipError_t e = hipMalloc((void**)out, v.size() * sizeof(int));
  if (e != hipSuccess) return e;
  return hipMemcpy(*out, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice);
}

}  // namespace

hipError_t aigv_launch_frame_resize_ingest(const uint8_t* hwc, int n_frames, int in_h, int in_w, int out_h, int out_w,
                                           const float* mean, const float* stdv, uint8_t* tmp_u8, uint8_t* out_u8, bf16_t* out_nchw,
                                           hipStream_t s) {
  if (n_frames <= 0) return hipSuccess;
  if (!hwc || !tmp_u8 || (!out_u8 && !out_nchw) || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0 || (out_nchw && (!mean || !stdv)))
    return hipErrorInvalidValue;
  DeviceTables t;
  {
    std::lock_guard<std::mutex> lock(g_resize_mutex);
    int dev = 0;
    hipError_t de = hipGetDevice(&dev);
    if (de != hipSuccess) return de;
    const auto key = std::make_pair(dev, std::make_pair(((long long)in_h << 32) | (unsigned)in_w, ((long long)out_h << 32) | (unsigned)out_w));
    auto it = g_resize_tables.find(key);
    if (it == g_resize_tables.end()) {
      const AxisTable h = precompute_axis(in_w, out_w);
      AxisTable v = precompute_axis(in_h, out_h);
      // the horizontal pass produces only the source rows the vertical pass reads (ybox_first .. ybox_last of ImagingResample)
      t.y_first = v.bounds[0];
      t.rows = v.bounds[(size_t)out_h * 2 - 2] + v.bounds[(size_t)out_h * 2 - 1] - t.y_first;
      for (int i = 0; i < out_h; ++i) v.bounds[2 * i] -= t.y_first;
      t.ksize_h = h.ksize; t.ksize_v = v.ksize;
      hipError_t e;
      if ((e = upload(h.bounds, &t.bounds_h)) != hipSuccess || (e = upload(h.kk, &t.kk_h)) != hipSuccess ||
          (e = upload(v.bounds, &t.bounds_v)) != hipSuccess || (e = upload(v.kk, &t.kk_v)) != hipSuccess)
        return e;
      g_resize_tables[key] = t;
    } else {
      t = it->second;
    }
  }
  const long n_rows = (long)n_frames * t.rows;
  const size_t lds = ((size_t)in_w * 3 + 3 + 15) & ~(size_t)15;
  if (lds <= 60 * 1024) {
    hipLaunchKernelGGL(resize_h_lds_kernel, dim3((unsigned)(n_rows < 8192 ? n_rows : 8192)), dim3(256), lds, s, hwc,
                       hwc + (size_t)n_frames * in_h * in_w * 3, tmp_u8, in_h, in_w, out_w, t.rows, t.y_first, t.bounds_h, t.kk_h,
                       t.ksize_h, n_rows);
  } else {
    const long total_h = n_rows * out_w;
    const long bh = (total_h + 255) / 256;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)(bh < 16384 ? bh : 16384)), dim3(256), 0, s, hwc, tmp_u8, in_h, in_w, out_w, t.rows,
                       t.y_first, t.bounds_h, t.kk_h, t.ksize_h, total_h);
  }
  ResizeVArgs a{};
  a.tmp = tmp_u8; a.out_u8 = out_u8; a.out_nchw = out_nchw; a.bounds = t.bounds_v; a.kk = t.kk_v; a.ksize = t.ksize_v;
  a.rows = t.rows; a.out_h = out_h; a.out_w = out_w; a.total = (long)n_frames * out_h * out_w;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean ? mean[c] : 0.f; a.std[c] = stdv ? stdv[c] : 1.f; }
  const bool dword_rows = (out_w * 3) % 4 == 0 && ((uintptr_t)tmp_u8 & 3) == 0 && (!out_u8 || ((uintptr_t)out_u8 & 3) == 0);
  const long bv = ((dword_rows ? a.total * 3 / 4 : a.total) + 255) / 256;
  if (dword_rows) hipLaunchKernelGGL(resize_v_dword_kernel, dim3((unsigned)(bv < 16384 ? bv : 16384)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)(bv < 16384 ? bv : 16384)), dim3(256), 0, s, a);
  return hipGetLastError();
}
