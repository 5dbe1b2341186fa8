// Image preparation of LineFeatureTracker::readImage (feature_tracker/src/line_feature_tracker.cpp:62-68) for a batch of
// frames: cv::remap(INTER_LINEAR, float maps, BORDER_CONSTANT 0) and CLAHE(clip, tiles).  Integer / float32 arithmetic
// in the order of OpenCV 3.4.2's imgwarp.cpp / clahe.cpp, so the result is bit-identical to oracle/preproc.cpp.
// All three kernels are streaming passes (remap: 8 B map + 1 B out per pixel, gathers from the raw frame through L2;
// histogram: 1 B per pixel; interpolation: 1 B in, 1 B out, LUTs of 16 KB per frame from L1/L2): HBM-bound, about
// 12 B per pixel in total -- three orders of magnitude below the EDLines stages that follow.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vpl {

#pragma clang fp contract(off)

struct PreBatch {
  int N, W, H;
  const uint8_t* raw;    // [N][H][W]
  const float* mapx;     // [H][W]
  const float* mapy;
  uint8_t* mid;          // [N][H][W] remapped (input of CLAHE)
  uint8_t* out;          // [N][H][W] = EdBatch::img
  uint8_t* lut;          // [N][tilesY*tilesX][256]
  int tilesX, tilesY, tw, th, clipLimit;
  float lutScale, inv_tw, inv_th;
};

__device__ inline uint32_t pre_remap_px(const uint8_t* __restrict__ src, int W, int H, float mx, float my) {
  const int sx = __float2int_rn(mx * 32), sy = __float2int_rn(my * 32);
  int ix = sx >> 5, iy = sy >> 5;
  ix = min(max(ix, -32768), 32767);
  iy = min(max(iy, -32768), 32767);
  const int fx = sx & 31, fy = sy & 31;
  const bool x0 = ix >= 0 && ix < W, x1 = ix + 1 >= 0 && ix + 1 < W, y0 = iy >= 0 && iy < H, y1 = iy + 1 >= 0 && iy + 1 < H;
  const uint8_t* p = src + (ptrdiff_t)iy * W + ix;
  const int v00 = (x0 && y0) ? p[0] : 0, v01 = (x1 && y0) ? p[1] : 0, v10 = (x0 && y1) ? p[W] : 0, v11 = (x1 && y1) ? p[W + 1] : 0;
  const int acc = v00 * ((32 - fx) * (32 - fy) * 32) + v01 * (fx * (32 - fy) * 32) + v10 * ((32 - fx) * fy * 32) + v11 * (fx * fy * 32);
  return (uint32_t)min(max((acc + (1 << 14)) >> 15, 0), 255);
}

// one lane per 4 consecutive pixels of a row
__global__ __launch_bounds__(256) void k_pre_remap(PreBatch B, uint8_t* __restrict__ dst) {
  const int W4 = (B.W + 3) >> 2;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= W4 * B.H) return;
  const int y = g / W4, x0 = (g - y * W4) * 4;
  const size_t PX = (size_t)B.W * B.H;
  const uint8_t* src = B.raw + blockIdx.y * PX;
  uint8_t* d = dst + blockIdx.y * PX + (size_t)y * B.W + x0;
  const float* mx = B.mapx + (size_t)y * B.W + x0;
  const float* my = B.mapy + (size_t)y * B.W + x0;
  if ((B.W & 3) == 0) {
    const float4 ax = *(const float4*)mx, ay = *(const float4*)my;
    const uint32_t r = pre_remap_px(src, B.W, B.H, ax.x, ay.x) | pre_remap_px(src, B.W, B.H, ax.y, ay.y) << 8 |
                       pre_remap_px(src, B.W, B.H, ax.z, ay.z) << 16 | pre_remap_px(src, B.W, B.H, ax.w, ay.w) << 24;
    *(uint32_t*)d = r;
  } else {
    for (int k = 0; k < 4 && x0 + k < B.W; ++k) d[k] = (uint8_t)pre_remap_px(src, B.W, B.H, mx[k], my[k]);
  }
}

__device__ inline int pre_reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

// CLAHE_CalcLut_Body: one workgroup per (tile, frame), one lane per histogram bin afterwards
__global__ __launch_bounds__(256) void k_pre_clahe_lut(PreBatch B, const uint8_t* __restrict__ in) {
  __shared__ int hist[256];
  __shared__ int wsum[4];
  const int t = threadIdx.x;
  const int tile = blockIdx.x, tx = tile % B.tilesX, ty = tile / B.tilesX;
  const uint8_t* src = in + (size_t)blockIdx.y * B.W * B.H;
  hist[t] = 0;
  __syncthreads();
  const int area = B.tw * B.th;
  for (int i = t; i < area; i += 256) {
    const int r = i / B.tw, c = i - r * B.tw;
    const int y = pre_reflect101(ty * B.th + r, B.H), x = pre_reflect101(tx * B.tw + c, B.W);
    atomicAdd(&hist[src[(size_t)y * B.W + x]], 1);
  }
  __syncthreads();
  int h = hist[t];
  if (B.clipLimit > 0) {
    int ex = max(h - B.clipLimit, 0);
    h = min(h, B.clipLimit);
    for (int o = 32; o; o >>= 1) ex += __shfl_xor(ex, o);
    if ((t & 63) == 0) wsum[t >> 6] = ex;
    __syncthreads();
    const int clipped = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    const int redistBatch = clipped / 256;
    const int residual = clipped - redistBatch * 256;
    h += redistBatch;
    if (residual != 0) {
      const int step = max(256 / residual, 1);
      if (t % step == 0 && t / step < residual) h += 1;
    }
  }
  // inclusive prefix sum over the 256 bins
  int s = h;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(s, o);
    if ((t & 63) >= o) s += v;
  }
  if ((t & 63) == 63) wsum[t >> 6] = s;
  __syncthreads();
  for (int w = 0; w < (t >> 6); ++w) s += wsum[w];
  const int v = __float2int_rn((float)s * B.lutScale);
  B.lut[((size_t)blockIdx.y * B.tilesX * B.tilesY + tile) * 256 + t] = (uint8_t)min(max(v, 0), 255);
}

__device__ inline uint32_t pre_interp_px(const uint8_t* __restrict__ p1, const uint8_t* __restrict__ p2, int tilesX, float inv_tw,
                                         int x, int v, float ya, float ya1) {
  const float txf = x * inv_tw - 0.5f;
  int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
  const float xa = txf - tx1, xa1 = 1.0f - xa;
  tx1 = max(tx1, 0); tx2 = min(tx2, tilesX - 1);
  const int i1 = tx1 * 256 + v, i2 = tx2 * 256 + v;
  const float res = (p1[i1] * xa1 + p1[i2] * xa) * ya1 + (p2[i1] * xa1 + p2[i2] * xa) * ya;
  return (uint32_t)min(max(__float2int_rn(res), 0), 255);
}

// CLAHE_Interpolation_Body: one lane per 4 consecutive pixels of a row
__global__ __launch_bounds__(256) void k_pre_clahe_interp(PreBatch B, const uint8_t* __restrict__ in) {
  const int W4 = (B.W + 3) >> 2;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= W4 * B.H) return;
  const int y = g / W4, x0 = (g - y * W4) * 4;
  const size_t PX = (size_t)B.W * B.H;
  const uint8_t* s = in + blockIdx.y * PX + (size_t)y * B.W + x0;
  uint8_t* d = B.out + blockIdx.y * PX + (size_t)y * B.W + x0;
  const float tyf = y * B.inv_th - 0.5f;
  int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
  const float ya = tyf - ty1, ya1 = 1.0f - ya;
  ty1 = max(ty1, 0); ty2 = min(ty2, B.tilesY - 1);
  const uint8_t* lut = B.lut + (size_t)blockIdx.y * B.tilesX * B.tilesY * 256;
  const uint8_t* p1 = lut + (size_t)ty1 * B.tilesX * 256;
  const uint8_t* p2 = lut + (size_t)ty2 * B.tilesX * 256;
  if ((B.W & 3) == 0) {
    const uint32_t v = *(const uint32_t*)s;
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) r |= pre_interp_px(p1, p2, B.tilesX, B.inv_tw, x0 + k, (v >> (8 * k)) & 255, ya, ya1) << (8 * k);
    *(uint32_t*)d = r;
  } else {
    for (int k = 0; k < 4 && x0 + k < B.W; ++k) d[k] = (uint8_t)pre_interp_px(p1, p2, B.tilesX, B.inv_tw, x0 + k, s[k], ya, ya1);
  }
}

}  // namespace vpl
