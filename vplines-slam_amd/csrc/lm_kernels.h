// HIP kernels of the KLT line matcher (line front-end), batch of frame pairs of one size.
//   k_lm_level0 / k_lm_down : padded image pyramid (pyrDown [1 4 6 4 1]^2 / 256, BORDER_REFLECT_101 border of 13 px)
//   k_lm_scharr             : Scharr 3/10/3 derivative planes, zero border (klt.cpp:42-122, :613)
//   k_lm_anchors            : LineMatching::Anchors (line_matching.cpp:532-602), one lane per reference line
//   k_lm_klt                : pyramidal LK with per-iteration gain/bias (lk_tracker_invoker_2d.cpp:28-479), ONE LANE PER
//                             KEY POINT, all four levels inside the lane; persistent waves pull chunks of 64 points.
//                             The I/dI window of a lane is a coalesced [pixel][lane] column in global scratch, the J
//                             window an LDS column.  The float accumulations run in the reference's pixel order, so
//                             the tracked positions are bit-identical to the CPU path.
//   k_lm_vote               : ClosestLine / Point2Line / TopologicalFilter (:48-133, :266-436), one workgroup per pair
//   k_lm_line_filter        : LineMatching::LineFilter (:167-264), one workgroup per frame, on the detector's lines in HBM
// Streaming kernels are HBM-bound; k_lm_klt is latency bound (byte gathers from L2-resident pyramids).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vplines_frontend.h"

namespace vpl {

constexpr int LM_WIN = 13;
constexpr int LM_NPX = LM_WIN * LM_WIN;
constexpr int LM_LEVELS = 4;      // maxLevel 3 (line_matching.cpp:14)
constexpr int LM_MAXCOUNT = 30;
constexpr int LM_KLT_SMEM = LM_NPX * 64 * 2;       // J windows of the 64 lanes of a wave
constexpr int LM_KLT_GRID = 256 * 7;              // persistent waves: 7 per CU fit the LDS

struct LmBatch {
  int N, W, H;                 // images (shared with the EDLines batch)
  int nLevels;                 // levels built (buildOpticalFlowPyramid's return value + 1)
  int lw[LM_LEVELS], lh[LM_LEVELS], ls[LM_LEVELS];   // level width / height / padded stride
  size_t loff[LM_LEVELS];      // offset of the level inside one image's pyramid block (pixels)
  size_t pyrSize;              // padded pixels of all levels of one image
  const uint8_t* img;          // [N][H][W]
  uint8_t* pyr;                // [N][pyrSize]
  int16_t* der;                // [N][pyrSize][2]
  // pairs
  int nPairs, maxLines, maxK;
  const int *refImg, *curImg;  // [P]
  const vpl_line *linesRef, *linesCur;   // [P][maxLines]
  const int *nRef, *nCur;      // [P]
  float2 *kpsRef, *kpsCur;     // [P][maxK]
  uint8_t* status;             // [P][maxK]
  float* err;                  // [P][maxK]
  int* kp2lineCur;             // [P][maxK]
  int *kpOff, *kpNum;          // [P][maxLines]
  int* nK;                     // [P]
  int* r2c;                    // [P][maxLines]
  int* chunkOff;               // [P+1] prefix of ceil(nK/64)
  int* workCounter;            // [1]
  uint2* winScratch;           // [LM_KLT_GRID][169][64]  I | dIx<<16 , dIy  per lane column
  int* valid;                  // [P]  1 matched, 0 Matching() returned false, -1 key-point capacity exceeded
  vpl_match_param prm;
  double epsilon;              // criteria_.epsilon^2 (klt.cpp:28-33)
};

__device__ __forceinline__ int lm_r101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// level 0: copy with a reflect-101 border
__global__ __launch_bounds__(256) void k_lm_level0(LmBatch B) {
  const int n = blockIdx.y;
  const int S = B.ls[0], HP = B.lh[0] + 2 * LM_WIN;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= S * HP) return;
  const int py = i / S, px = i - py * S;
  const int x = lm_r101(px - LM_WIN, B.W), y = lm_r101(py - LM_WIN, B.H);
  B.pyr[(size_t)n * B.pyrSize + B.loff[0] + i] = B.img[((size_t)n * B.H + y) * B.W + x];
}

// level l from the padded level l-1 (reads reach 2 px into the border, which already holds the reflected pixels)
__global__ __launch_bounds__(256) void k_lm_down(LmBatch B, int l) {
  const int n = blockIdx.y;
  const int S = B.ls[l], HP = B.lh[l] + 2 * LM_WIN;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= S * HP) return;
  const int py = i / S, px = i - py * S;
  const int x = lm_r101(px - LM_WIN, B.lw[l]), y = lm_r101(py - LM_WIN, B.lh[l]);
  const int SP = B.ls[l - 1];
  const uint8_t* src = B.pyr + (size_t)n * B.pyrSize + B.loff[l - 1] + (size_t)(2 * y - 2 + LM_WIN) * SP + (2 * x - 2 + LM_WIN);
  int s = 0;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const uint8_t* r = src + (size_t)j * SP;
    const int rs = r[0] + r[4] + 4 * (r[1] + r[3]) + 6 * r[2];
    s += (j == 0 || j == 4) ? rs : (j == 2 ? 6 * rs : 4 * rs);
  }
  B.pyr[(size_t)n * B.pyrSize + B.loff[l] + i] = (uint8_t)((s + 128) >> 8);
}

// Scharr planes of every level (blockIdx.z = level); zero outside the image
__global__ __launch_bounds__(256) void k_lm_scharr(LmBatch B) {
  const int n = blockIdx.y, l = blockIdx.z;
  if (l >= B.nLevels) return;
  const int S = B.ls[l], HP = B.lh[l] + 2 * LM_WIN;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= S * HP) return;
  const int py = i / S, px = i - py * S;
  const int x = px - LM_WIN, y = py - LM_WIN;
  short dx = 0, dy = 0;
  if (x >= 0 && x < B.lw[l] && y >= 0 && y < B.lh[l]) {
    const uint8_t* c = B.pyr + (size_t)n * B.pyrSize + B.loff[l] + i;
    const uint8_t *r0 = c - S, *r2 = c + S;
    // vertical 3/10/3 smoothing then horizontal difference; vertical difference then horizontal 3/10/3 (int16 wrap as deriv_type)
    const int sm = (short)((r0[-1] + r2[-1]) * 3 + c[-1] * 10), sp = (short)((r0[1] + r2[1]) * 3 + c[1] * 10);
    const int dm = (short)(r2[-1] - r0[-1]), d0 = (short)(r2[0] - r0[0]), dp = (short)(r2[1] - r0[1]);
    dx = (short)(sp - sm);
    dy = (short)((dp + dm) * 3 + d0 * 10);
  }
  int16_t* o = B.der + ((size_t)n * B.pyrSize + B.loff[l] + i) * 2;
  o[0] = dx;
  o[1] = dy;
}

#pragma clang fp contract(off)   // keep the reference's (and the oracle's) un-fused float arithmetic

// Anchors: lane per reference line, serial prefix over the (<= maxLines) counts
__global__ __launch_bounds__(256) void k_lm_anchors(LmBatch B) {
  const int p = blockIdx.x;
  const int nr = B.nRef[p], nc = B.nCur[p];
  __shared__ int total;
  if (nr == 0 || nc == 0) {
    if (threadIdx.x == 0) { B.valid[p] = 0; B.nK[p] = 0; }
    return;
  }
  const vpl_line* L = B.linesRef + (size_t)p * B.maxLines;
  int* num = B.kpNum + (size_t)p * B.maxLines;
  int* off = B.kpOff + (size_t)p * B.maxLines;
  const int step = B.prm.step;
  extern __shared__ int lm_num[];   // maxLines: the counts, then their exclusive prefix (serial over LDS, not over HBM)
  for (int i = threadIdx.x; i < nr; i += 256) { const int c = int(L[i].length / step) + 2; num[i] = c; lm_num[i] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int s = 0;
    for (int i = 0; i < nr; ++i) { const int c = lm_num[i]; lm_num[i] = s; s += c; }
    total = s;
    B.nK[p] = s <= B.maxK ? s : 0;
    B.valid[p] = s <= B.maxK ? 1 : -1;
  }
  __syncthreads();
  if (total > B.maxK) return;
  for (int i = threadIdx.x; i < nr; i += 256) off[i] = lm_num[i];
  float2* kps = B.kpsRef + (size_t)p * B.maxK;
  for (int i = threadIdx.x; i < nr; i += 256) {
    const float x1 = L[i].line_endpoint[0], y1 = L[i].line_endpoint[1], x2 = L[i].line_endpoint[2], y2 = L[i].line_endpoint[3];
    const float len = L[i].length;
    const float dirx = (x2 - x1) / len, diry = (y2 - y1) / len;
    const float ddx = step * dirx, ddy = step * diry;
    const int iter = num[i] - 2;
    float px = x1, py = y1;
    float2* o = kps + lm_num[i];
    for (int j = 0; j <= iter; ++j) {
      o[j] = make_float2(px, py);
      px += ddx;
      py += ddy;
    }
    o[iter + 1] = make_float2(x2, y2);
  }
}

struct LmWeights { int w00, w01, w10, w11; };
__device__ __forceinline__ LmWeights lm_weights(float a, float b) {
  LmWeights w;
  w.w00 = __float2int_rn((1.f - a) * (1.f - b) * (1 << 14));
  w.w01 = __float2int_rn(a * (1.f - b) * (1 << 14));
  w.w10 = __float2int_rn((1.f - a) * b * (1 << 14));
  w.w11 = (1 << 14) - w.w00 - w.w01 - w.w10;
  return w;
}

// 14 consecutive bytes of one window row from an arbitrarily aligned address: five aligned dword loads + v_alignbyte
// (a byte gather per pixel costs one cache-line lookup per lane per byte; this is 5 lookups per 14 pixels)
__device__ __forceinline__ void lm_load_row(const uint8_t* p, int (&v)[LM_WIN + 1]) {
  const uintptr_t a = (uintptr_t)p;
  const unsigned* q = (const unsigned*)(a & ~(uintptr_t)3);
  const unsigned sh = (unsigned)(a & 3);
  const unsigned d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
  unsigned w[4];
  w[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
  w[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
  w[2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
  w[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
#pragma unroll
  for (int x = 0; x <= LM_WIN; ++x) v[x] = (int)((w[x >> 2] >> ((x & 3) * 8)) & 255u);
}

// bilinear J window (scaled x32) into the lane's LDS column; returns sum and sum of squares (exact integers).
// Row r contributes to window row r-1 as its lower neighbour and to row r as its upper one: one load per row.
__device__ __forceinline__ void lm_sample_J(const uint8_t* Jp, int S, int ix, int iy, LmWeights w, short* Jw, int& sJ,
                                            long long& qJ) {
  sJ = 0;
  unsigned long long q = 0;
  const uint8_t* r = Jp + (size_t)(iy + LM_WIN) * S + (ix + LM_WIN);
  int v[LM_WIN + 1], up[LM_WIN];
  lm_load_row(r, v);
#pragma unroll
  for (int x = 0; x < LM_WIN; ++x) up[x] = v[x] * w.w00 + v[x + 1] * w.w01 + (1 << 8);
#pragma unroll 1
  for (int y = 0; y < LM_WIN; ++y) {
    r += S;
    lm_load_row(r, v);
#pragma unroll
    for (int x = 0; x < LM_WIN; ++x) {
      const int val = (up[x] + v[x] * w.w10 + v[x + 1] * w.w11) >> 9;
      Jw[(y * LM_WIN + x) * 64] = (short)val;
      sJ += val;
      q += (unsigned)(val * val);
      up[x] = v[x] * w.w00 + v[x + 1] * w.w01 + (1 << 8);
    }
  }
  qJ = (long long)q;
}

// getImageNormParams (klt.cpp:4-10) from the integer moments of the two windows
__device__ __forceinline__ void lm_norm_params(int sI, long long qI, int sJ, long long qJ, float& alpha, float& beta) {
  const double scale = 1.0 / LM_NPX;
  const double mI = (double)sI * scale, mJ = (double)sJ * scale;
  const double sdI = sqrt(fmax((double)qI * scale - mI * mI, 0.0));
  const double sdJ = sqrt(fmax((double)qJ * scale - mJ * mJ, 0.0));
  alpha = float(sdI / sdJ);
  beta = float(mI - alpha * mJ);
}

// chunk plan: 64 key points per wave, chunks of all pairs in one list (prefix over the pairs), work counter reset
__global__ __launch_bounds__(64) void k_lm_plan(LmBatch B) {
  const int lane = threadIdx.x;
  int base = 0;
  for (int p0 = 0; p0 < B.nPairs; p0 += 64) {   // 64 pairs per trip: one load latency, inclusive scan by shuffles
    const int p = p0 + lane;
    const int c = p < B.nPairs ? (B.nK[p] + 63) >> 6 : 0;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (p < B.nPairs) B.chunkOff[p] = base + incl - c;
    base += __shfl(incl, 63, 64);
  }
  if (lane == 0) {
    B.chunkOff[B.nPairs] = base;
    *B.workCounter = 0;
  }
}

// Persistent waves: each takes chunks of 64 key points from the work counter.  Per lane: the I / dIx / dIy window of
// the current level lives in a global scratch column (uint2 per pixel, [pixel][lane] => one coalesced 512-B access per
// wave per pixel), the J window of the current iteration in LDS.
__global__ __launch_bounds__(64) void k_lm_klt(LmBatch B) {
  extern __shared__ short lm_sm[];
  __shared__ int chunkS;
  const int lane = threadIdx.x;
  short* Jw = lm_sm + lane;
  uint2* win = B.winScratch + (size_t)blockIdx.x * LM_NPX * 64 + lane;
  const bool illum = B.prm.illumination_adapt != 0;
  const float halfWin = (LM_WIN - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (1 << 20);
  const int maxLevel = B.nLevels - 1;
  const int totalChunks = B.chunkOff[B.nPairs];

  for (;;) {
    __syncthreads();
    if (lane == 0) chunkS = atomicAdd(B.workCounter, 1);
    __syncthreads();
    const int chunk = chunkS;
    if (chunk >= totalChunks) return;
    int lo = 0, hi = B.nPairs - 1;          // last pair whose first chunk is <= chunk
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (B.chunkOff[mid] <= chunk) lo = mid; else hi = mid - 1;
    }
    const int p = lo;
    const int idx = (chunk - B.chunkOff[p]) * 64 + lane;
    if (idx < B.nK[p]) {                    // (lanes past the end of a ragged chunk idle until the next chunk)

    const uint8_t* pyrI = B.pyr + (size_t)B.refImg[p] * B.pyrSize;
    const uint8_t* pyrJ = B.pyr + (size_t)B.curImg[p] * B.pyrSize;
    const int16_t* derI = B.der + (size_t)B.refImg[p] * B.pyrSize * 2;
    const float2 prev0 = B.kpsRef[(size_t)p * B.maxK + idx];
    float outx = 0.f, outy = 0.f, errv = 0.f;
    bool st = true;

    for (int level = maxLevel; level >= 0; --level) {
      const int S = B.ls[level], w = B.lw[level], h = B.lh[level];
      const uint8_t* Ip = pyrI + B.loff[level];
      const uint8_t* Jp = pyrJ + B.loff[level];
      const int16_t* Dp = derI + B.loff[level] * 2;
      const float lscale = (float)(1. / (1 << level));
      float prevx = prev0.x * lscale, prevy = prev0.y * lscale;
      float nx, ny;
      if (level == maxLevel) { nx = prevx; ny = prevy; }
      else { nx = outx * 2.f; ny = outy * 2.f; }
      outx = nx; outy = ny;

      prevx -= halfWin; prevy -= halfWin;
      const int ipx = (int)floorf(prevx), ipy = (int)floorf(prevy);
      if (ipx < -LM_WIN || ipx >= w || ipy < -LM_WIN || ipy >= h) {
        if (level == 0) { st = false; errv = 0.f; }
        continue;
      }
      const LmWeights wt = lm_weights(prevx - ipx, prevy - ipy);
      float iA11 = 0, iA12 = 0, iA22 = 0;
      int sI = 0;
      unsigned long long qIu = 0;
      {
        const size_t o = (size_t)(ipy + LM_WIN) * S + (ipx + LM_WIN);
        const uint8_t* r = Ip + o;
        const int* d = (const int*)(Dp + o * 2);   // (dx, dy) pairs, 4-byte aligned
        int v[LM_WIN + 1], up[LM_WIN], e[LM_WIN + 1], upx[LM_WIN], upy[LM_WIN];
        lm_load_row(r, v);
#pragma unroll
        for (int x = 0; x <= LM_WIN; ++x) e[x] = d[x];
#pragma unroll
        for (int x = 0; x < LM_WIN; ++x) {
          up[x] = v[x] * wt.w00 + v[x + 1] * wt.w01 + (1 << 8);
          upx[x] = (short)e[x] * wt.w00 + (short)e[x + 1] * wt.w01 + (1 << 13);
          upy[x] = (e[x] >> 16) * wt.w00 + (e[x + 1] >> 16) * wt.w01 + (1 << 13);
        }
#pragma unroll 1
        for (int y = 0; y < LM_WIN; ++y) {
          r += S; d += S;
          lm_load_row(r, v);
#pragma unroll
          for (int x = 0; x <= LM_WIN; ++x) e[x] = d[x];
#pragma unroll
          for (int x = 0; x < LM_WIN; ++x) {
            const int ival = (up[x] + v[x] * wt.w10 + v[x + 1] * wt.w11) >> 9;
            const int ixval = (upx[x] + (short)e[x] * wt.w10 + (short)e[x + 1] * wt.w11) >> 14;
            const int iyval = (upy[x] + (e[x] >> 16) * wt.w10 + (e[x + 1] >> 16) * wt.w11) >> 14;
            win[(size_t)(y * LM_WIN + x) * 64] = make_uint2((unsigned)(ival & 0xffff) | ((unsigned)ixval << 16), (unsigned)iyval);
            sI += ival;
            qIu += (unsigned)(ival * ival);
            iA11 += (float)(ixval * ixval);
            iA12 += (float)(ixval * iyval);
            iA22 += (float)(iyval * iyval);
            up[x] = v[x] * wt.w00 + v[x + 1] * wt.w01 + (1 << 8);
            upx[x] = (short)e[x] * wt.w00 + (short)e[x + 1] * wt.w01 + (1 << 13);
            upy[x] = (e[x] >> 16) * wt.w00 + (e[x + 1] >> 16) * wt.w01 + (1 << 13);
          }
        }
      }
      const long long qI = (long long)qIu;
      const float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
      float D = A11 * A22 - A12 * A12;
      const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * LM_WIN * LM_WIN);
      if (minEig < 1e-4f || D < 1.1920928955078125e-07f) {
        if (level == 0) st = false;
        continue;
      }
      D = 1.f / D;
      nx -= halfWin; ny -= halfWin;
      float pdx = 0, pdy = 0;
      int j;
      for (j = 0; j < LM_MAXCOUNT; ++j) {
        const int inx = (int)floorf(nx), iny = (int)floorf(ny);
        if (inx < -halfWin || inx >= w || iny < -halfWin || iny >= h) {
          if (level == 0) st = false;
          break;
        }
        int sJ; long long qJ;
        lm_sample_J(Jp, S, inx, iny, lm_weights(nx - inx, ny - iny), Jw, sJ, qJ);
        float alpha = 1.0f, beta = 0.0f;
        if (illum) lm_norm_params(sI, qI, sJ, qJ, alpha, beta);
        float ib1 = 0, ib2 = 0;
#pragma unroll 13
        for (int k = 0; k < LM_NPX; ++k) {
          const uint2 t = win[(size_t)k * 64];
          const float diff = alpha * (float)Jw[k * 64] + beta - (float)(int)(t.x & 0xffff);
          ib1 += diff * (float)((int)t.x >> 16);
          ib2 += diff * (float)(int)t.y;
        }
        const float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
        const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
        nx += dx; ny += dy;
        outx = nx + halfWin; outy = ny + halfWin;
        if ((double)dx * dx + (double)dy * dy <= B.epsilon) break;
        if (j > 0 && fabsf(dx + pdx) < 0.01 && fabsf(dy + pdy) < 0.01) {
          outx -= dx * 0.5f; outy -= dy * 0.5f;
          break;
        }
        pdx = dx; pdy = dy;
      }
      if (j == LM_MAXCOUNT && level == 0) st = false;
      if (level == 0 && st) {
        const float fx = outx - halfWin, fy = outy - halfWin;
        const int ix = (int)floorf(fx), iy = (int)floorf(fy);
        if (ix < -LM_WIN || ix >= w || iy < -LM_WIN || iy >= h) { st = false; continue; }
        int sJ; long long qJ;
        lm_sample_J(Jp, S, ix, iy, lm_weights(fx - ix, fy - iy), Jw, sJ, qJ);
        float alpha = 1.0f, beta = 0.0f;
        if (illum) lm_norm_params(sI, qI, sJ, qJ, alpha, beta);
        float e = 0.f;
#pragma unroll 13
        for (int k = 0; k < LM_NPX; ++k)
          e += fabsf(alpha * (float)Jw[k * 64] + beta - (float)(int)(win[(size_t)k * 64].x & 0xffff));
        errv = e * 1.f / (32 * LM_WIN * LM_WIN);
      }
    }
    const size_t o = (size_t)p * B.maxK + idx;
    B.kpsCur[o] = make_float2(outx, outy);
    B.status[o] = st ? 1 : 0;
    B.err[o] = errv;
    }
  }
}

// LineMatching::PointLineDistance :21-41
__device__ __forceinline__ float lm_point_line_distance(float x, float y, const float* e) {
  const float vx = e[2] - e[0], vy = e[3] - e[1];
  const float ux = e[0] - x, uy = e[1] - y;
  float t = -(vx * ux + vy * uy) / (vx * vx + vy * vy);
  if (t < 0) t = 0; else if (t > 1) t = 1;
  const float dx = t * vx + ux, dy = t * vy + uy;
  return sqrtf(dx * dx + dy * dy);
}

// ClosestLine + Point2Line + TopologicalFilter for one pair; dynamic LDS: r2c[maxLines] + cnt[maxLines] + 1
__global__ __launch_bounds__(256) void k_lm_vote(LmBatch B) {
  extern __shared__ int lm_vs[];
  const int p = blockIdx.x;
  if (B.valid[p] != 1) return;
  const int nr = B.nRef[p], nc = B.nCur[p], nk = B.nK[p];
  const vpl_line* LR = B.linesRef + (size_t)p * B.maxLines;
  const vpl_line* LC = B.linesCur + (size_t)p * B.maxLines;
  const float2* kc = B.kpsCur + (size_t)p * B.maxK;
  const uint8_t* st = B.status + (size_t)p * B.maxK;
  const float* er = B.err + (size_t)p * B.maxK;
  int* k2l = B.kp2lineCur + (size_t)p * B.maxK;
  int* r2c = lm_vs;
  int* cnt = lm_vs + B.maxLines;
  int* matchNum = cnt + B.maxLines;

  // ClosestLine :48-86
  for (int i = threadIdx.x; i < nk; i += 256) {
    int best = -1;
    if (st[i] && !(er[i] > B.prm.klt_error_threshold)) {
      float mind = 1000000;
      int mi = -1;
      const float2 pt = kc[i];
      for (int j = 0; j < nc; ++j) {
        const float d = lm_point_line_distance(pt.x, pt.y, LC[j].line_endpoint);
        if (d < mind) { mind = d; mi = j; }
      }
      if (mind < B.prm.closest_line_threshold) best = mi;
    }
    k2l[i] = best;
  }
  if (threadIdx.x == 0) *matchNum = 0;
  __syncthreads();

  // Point2Line :88-133 -- the arg-max of the vote histogram (lowest index wins ties) without the histogram
  const int* off = B.kpOff + (size_t)p * B.maxLines;
  const int* num = B.kpNum + (size_t)p * B.maxLines;
  for (int i = threadIdx.x; i < nr; i += 256) {
    const int* v = k2l + off[i];
    const int kn = num[i];
    int maxv = 0, maxi = 0;   // an all-zero histogram elects line 0 with value 0 (max_value starts at -1)
    for (int a = 0; a < kn; ++a) {
      const int c = v[a];
      if (c < 0) continue;
      int n = 0;
      for (int b = 0; b < kn; ++b) n += (v[b] == c);
      if (n > maxv || (n == maxv && c < maxi)) { maxv = n; maxi = c; }
    }
    int m = -1;
    if (!(maxv <= 2 || float(maxv) / kn < B.prm.line_matching_ratio ||
          LC[maxi].length > LR[i].length * B.prm.line_distance_error_ratio ||
          LC[maxi].length < LR[i].length / B.prm.line_distance_error_ratio))
      m = maxi;
    r2c[i] = m;
    cnt[i] = 0;
    if (m != -1) atomicAdd(matchNum, 1);
  }
  __syncthreads();

  // TopologicalFilter :266-397: ordered pair (r1, r2) in violation adds one to both counters
  if (B.prm.topological_filter) {
    for (int i = threadIdx.x; i < nr; i += 256) {
      const int ci = r2c[i];
      if (ci == -1) continue;
      int c = 0;
      for (int o = 0; o < nr; ++o) {
        const int co = r2c[o];
        if (o == i || co == -1) continue;
#pragma unroll
        for (int dirn = 0; dirn < 2; ++dirn) {
          const int r1 = dirn ? o : i, r2 = dirn ? i : o, c1 = dirn ? co : ci, c2 = dirn ? ci : co;
          const float ldr = fabsf(LR[r2].length - LC[c2].length) / LR[r2].length;
          if (ldr > B.prm.topo_length_tolerate_ratio) continue;
          // SidenessCheck :399-436
          const double a1 = LR[r1].line_equation[0], b1 = LR[r1].line_equation[1], c1e = LR[r1].line_equation[2];
          const double px1 = LR[r2].center[0], py1 = LR[r2].center[1];
          double a2 = LC[c1].line_equation[0], b2 = LC[c1].line_equation[1], c2e = LC[c1].line_equation[2];
          const double px2 = LC[c2].center[0], py2 = LC[c2].center[1];
          if ((fabs(a1 - a2) + fabs(b1 - b2)) > (fabs(a1 + a2) + fabs(b1 + b2))) { a2 = -a2; b2 = -b2; c2e = -c2e; }
          const float d1 = (float)((px1 * a1 + py1 * b1 + c1e) / sqrt(a1 * a1 + b1 * b1));
          const float d2 = (float)((px2 * a2 + py2 * b2 + c2e) / sqrt(a2 * a2 + b2 * b2));
          if (d1 * d2 < 0 && fabsf(d1) > B.prm.topo_distance_threshold && fabsf(d2) > B.prm.topo_distance_threshold) ++c;
        }
      }
      cnt[i] = c;
    }
    __syncthreads();
    float threshold = B.prm.topo_violation_ratio * (*matchNum - 1);
    if (threshold < 2) threshold = 2;
    for (int i = threadIdx.x; i < nr; i += 256)
      if (cnt[i] > threshold) r2c[i] = -1;
    __syncthreads();
  }
  int* out = B.r2c + (size_t)p * B.maxLines;
  for (int i = threadIdx.x; i < nr; i += 256) out[i] = r2c[i];
}

// LineMatching::LineFilter (line_matching.cpp:167-264) on the line table of a batch of frames (vpl_line [N][ML], counts [N]),
// in place.  One work-group per frame.  Lines by decreasing length (equal lengths in index order: the reference's std::sort
// leaves that open); the outer loop over the longer line is sequential (a removed line removes nothing), the inner loop over
// the shorter lines is the work-group's.  Dynamic LDS: order[ML] ints + len[ML] floats.  Same float expressions as the
// reference (file-scope contract(off) above).
__global__ __launch_bounds__(256) void k_lm_line_filter(vpl_line* lines, int* counts, int ML, float distTh, float parTh) {
  extern __shared__ int lf_smem[];
  int* order = lf_smem;
  float* len = reinterpret_cast<float*>(lf_smem + ML);
  const int n = blockIdx.x, tid = threadIdx.x;
  vpl_line* L = lines + (size_t)n * ML;
  const int m = min(counts[n], ML);
  for (int k = tid; k < m; k += 256) len[k] = L[k].length;
  __syncthreads();
  for (int k = tid; k < m; k += 256) {   // rank = lines that come before k
    const float lk = len[k];
    int rank = 0;
    for (int j = 0; j < m; ++j) rank += (len[j] > lk || (len[j] == lk && j < k)) ? 1 : 0;
    order[rank] = k;
  }
  __syncthreads();
  for (int i = 0; i < m; ++i) {
    const int i1 = order[i];
    const float u_dist = len[i1];
    if (u_dist != -1.f) {   // uniform
      const float e1[4] = {L[i1].line_endpoint[0], L[i1].line_endpoint[1], L[i1].line_endpoint[2], L[i1].line_endpoint[3]};
      const float ux = e1[2] - e1[0], uy = e1[3] - e1[1];
      for (int j = i + 1 + tid; j < m; j += 256) {
        const int i2 = order[j];
        const float v_dist = len[i2];
        if (v_dist == -1.f) continue;
        const float* e2 = L[i2].line_endpoint;
        const float vx = e2[2] - e2[0], vy = e2[3] - e2[1];
        if (fabsf(ux * vy - vx * uy) > (u_dist * v_dist * parTh)) continue;
        const float d1 = lm_point_line_distance(e2[0], e2[1], e1);
        const float d2 = lm_point_line_distance(e2[2], e2[3], e1);
        if (d1 < distTh || d2 < distTh) len[i2] = -1.f;
      }
    }
    __syncthreads();
  }
  // compaction in the original order; a line only ever moves towards the front, chunks of 256 in increasing order
  for (int base = 0; base < m; base += 256) {
    const int k = base + tid;
    vpl_line v;
    int dst = -1;
    if (k < m && len[k] != -1.f) {
      v = L[k];
      dst = 0;
      for (int j = 0; j < k; ++j) dst += len[j] != -1.f ? 1 : 0;
    }
    __syncthreads();
    if (dst >= 0) L[dst] = v;
    __syncthreads();
  }
  if (tid == 0) {
    int c = 0;
    for (int j = 0; j < m; ++j) c += len[j] != -1.f ? 1 : 0;
    counts[n] = c;
  }
}

}  // namespace vpl
