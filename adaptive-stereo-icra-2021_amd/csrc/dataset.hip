// Dataset layer on the device: what the reference does per sample on the host with torchvision /
// numpy — ToTensor, horizontal flip, crop, disparity decoding (datasets/stereo_dataset.py:49-96,
// utils/dataset_utils.py:26-57) — as two gather kernels over the raw decoded file contents.  The host only parses
// the container formats (PNG via PIL, PFM / NPY headers) and uploads the raw samples once; the multi-scale pyramid
// (stereo_dataset.py:98-135) reuses as_upsample_bilinear_fwd, which is a general align_corners=False resize.
#include "as_common.h"

// dst[c][y][x] = src[i0+y][col][c] / 255,  col = j0+x, or W0-1-(j0+x) when the pair was flipped before cropping
__global__ __launch_bounds__(256) void decode_rgb8_kernel(const uint8_t* __restrict__ src, int H0, int W0, int i0, int j0,
                                                           int H, int W, int hflip, float* __restrict__ dst) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= H * W) return;
  const int y = o / W, x = o - y * W;
  const int col = hflip ? W0 - 1 - (j0 + x) : j0 + x;
  const uint8_t* px = src + ((long)(i0 + y) * W0 + col) * 3;
  const long plane = (long)H * W;
  dst[o] = (float)px[0] / 255.f;
  dst[plane + o] = (float)px[1] / 255.f;
  dst[2 * plane + o] = (float)px[2] / 255.f;
}

// dst[y][x] = f(src[row][col]);  row = i0+y, or H0-1-(i0+y) for bottom-up files (PFM); col as above;
// f(v) = v*scale (reciprocal == 0) or scale/v (reciprocal == 1: depth -> disparity)
template <typename T>
__global__ __launch_bounds__(256) void decode_plane_kernel(const T* __restrict__ src, int H0, int W0, int i0, int j0, int H,
                                                            int W, int hflip, int vflip, float scale, int reciprocal,
                                                            float* __restrict__ dst) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= H * W) return;
  const int y = o / W, x = o - y * W;
  const int col = hflip ? W0 - 1 - (j0 + x) : j0 + x;
  const int row = vflip ? H0 - 1 - (i0 + y) : i0 + y;
  const float v = (float)src[(long)row * W0 + col];
  dst[o] = reciprocal ? scale / v : v * scale;
}

static bool window_ok(int H0, int W0, int i0, int j0, int H, int W) {
  return H0 > 0 && W0 > 0 && H > 0 && W > 0 && i0 >= 0 && j0 >= 0 && i0 + H <= H0 && j0 + W <= W0 &&
         (long)H * W < (1L << 31);
}

extern "C" int as_decode_rgb8(const uint8_t* src, int H0, int W0, int i0, int j0, int H, int W, int hflip, float* dst,
                              void* stream) {
  AS_CHECK_ARG(src && dst && window_ok(H0, W0, i0, j0, H, W), "as_decode_rgb8: bad argument");
  hipLaunchKernelGGL(decode_rgb8_kernel, dim3(as_div_up((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream, src, H0,
                     W0, i0, j0, H, W, hflip, dst);
  AS_CHECK_LAUNCH("as_decode_rgb8");
  return AS_OK;
}

extern "C" int as_decode_plane(const void* src, int dtype, int H0, int W0, int i0, int j0, int H, int W, int hflip,
                               int vflip, float scale, int reciprocal, float* dst, void* stream) {
  AS_CHECK_ARG(src && dst && window_ok(H0, W0, i0, j0, H, W), "as_decode_plane: bad argument");
  AS_CHECK_ARG(dtype >= 0 && dtype <= 2, "as_decode_plane: dtype %d (0 = f32, 1 = u16, 2 = u8)", dtype);
  const dim3 grid(as_div_up((long)H * W, 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == 0)
    hipLaunchKernelGGL(decode_plane_kernel<float>, grid, block, 0, st, static_cast<const float*>(src), H0, W0, i0, j0, H, W,
                       hflip, vflip, scale, reciprocal, dst);
  else if (dtype == 1)
    hipLaunchKernelGGL(decode_plane_kernel<uint16_t>, grid, block, 0, st, static_cast<const uint16_t*>(src), H0, W0, i0,
                       j0, H, W, hflip, vflip, scale, reciprocal, dst);
  else
    hipLaunchKernelGGL(decode_plane_kernel<uint8_t>, grid, block, 0, st, static_cast<const uint8_t*>(src), H0, W0, i0, j0,
                       H, W, hflip, vflip, scale, reciprocal, dst);
  AS_CHECK_LAUNCH("as_decode_plane");
  return AS_OK;
}
