// HIP kernels of the EDLines extractor (line front-end), batch of N frames of one size.
//   k_ed_grad   : Sobel 3x3 (BORDER_REFLECT_101) -> |dx|+|dy| -> threshold -> /4 (half-to-even) -> direction
//                 (edline_detector.cpp:125-136); streaming, HBM-bound: 1 byte read, 7 bytes written per pixel
//   k_ed_blur_grad : the same with cv::GaussianBlur(image, Size(ksize, ksize), sigma) in front (EdgeDrawing with
//                 smoothed = false, edline_detector.cpp:81-86: the reference's default): raw tile + halo in LDS, OpenCV's
//                 8.8 fixed-point row / column passes, Sobel of the blurred tile -- the blurred frame never goes to HBM
//   k_ed_anchor : anchor test on the scan lattice + ORDERED compaction (w outer, h inner, :148-164)
//   k_ed_code   : per-pixel routing byte (walkable, direction, arg-max forward neighbour for both senses of travel)
//   k_ed_route  : smart routing (:166-707): inherently serial and order dependent per frame -> one wave per
//                 frame; edge bitmap + a full-height 128-column strip of routing bytes in LDS; throughput comes from the batch
//   k_ed_fit    : per edge chain least-squares fit / extension / Helmholtz validation (:729-1174), one wave per chain
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vpl {

struct EdBatch {
  int N, W, H;
  int cap;          // W*H/5  (edgePixelArraySize)
  int capEdges;     // cap/20 (maxNumOfEdge)
  int maxLines;
  int gradTh, anchorTh, scan, minLineLen;
  double fitErr;
  const uint8_t* img;      // [N][H][W]
  int16_t *dx, *dy, *g;    // [N][H][W]
  uint8_t* dir;            // [N][H][W]
  uint8_t* code;           // [N][H][Wc] routing bytes (k_ed_code)
  int Wc;                  // W rounded up to the routing strip width (128)
  int routeHS;             // rows of the routing strip k_ed_route keeps in LDS (a multiple of 32; >= H: the full height)
  unsigned long long* rstats; // [N][4] routing counters: steps, tile loads, walks, cycles (diagnostic)
  uint32_t *anchX, *anchY; // [N][cap]
  int* nAnch;              // [N]
  uint32_t *fX, *fY;       // [N][cap]  scratch: first part of the chain under construction
  uint32_t *cX, *cY;       // [N][2*cap] chains
  uint32_t* sId;           // [N][capEdges+2]
  int* nEdges;             // [N]
  // lines out: 10 doubles each + key
  double* lines;           // [N][maxLines][10]
  uint32_t* lkey;          // [N][maxLines]  (chain id << 8 | ordinal) for a deterministic order
  int* nLines;             // [N]
  // Gaussian pre-blur (smoothed = false): 2 * blurR + 1 taps in 8.8 fixed point (host: gauss_kernel_q8), optional copy of
  // the blurred frames for the tests
  int blurR;
  int blurK[7];
  uint8_t* blurOut;        // [N][H][W] or nullptr
};

__device__ __forceinline__ int refl101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

__global__ __launch_bounds__(256) void k_ed_grad(EdBatch B) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int W = B.W, H = B.H;
  if (i >= W * H) return;
  const int y = i / W, x = i - y * W;
  const uint8_t* im = B.img + (size_t)n * W * H;
  const uint8_t* r0 = im + (size_t)refl101(y - 1, H) * W;
  const uint8_t* r1 = im + (size_t)y * W;
  const uint8_t* r2 = im + (size_t)refl101(y + 1, H) * W;
  const int xm = refl101(x - 1, W), xp = refl101(x + 1, W);
  const int gx = (r0[xp] + 2 * r1[xp] + r2[xp]) - (r0[xm] + 2 * r1[xm] + r2[xm]);
  const int gy = (r2[xm] + 2 * r2[x] + r2[xp]) - (r0[xm] + 2 * r0[x] + r0[xp]);
  const int ax = gx < 0 ? -gx : gx, ay = gy < 0 ? -gy : gy;
  const int sum = ax + ay;
  const int t = sum > B.gradTh + 1 ? sum : 0;
  // round half to even of t/4 (t >= 0): q = t>>2, r = t&3 ; r==3 -> q+1 ; r==2 -> q + (q&1) ; else q
  const int q = t >> 2, r = t & 3;
  const int gq = r == 3 ? q + 1 : (r == 2 ? q + (q & 1) : q);
  const size_t o = (size_t)n * W * H + i;
  B.dx[o] = (int16_t)gx;
  B.dy[o] = (int16_t)gy;
  B.g[o] = (int16_t)gq;
  B.dir[o] = ax < ay ? 255 : 0;
}

// EdgeDrawing with smoothed = false: GaussianBlur (fixedSmoothInvoker<uint8_t, ufixedpoint16> of OpenCV's smooth.cpp: taps in
// 8.8 fixed point, row pass into 16 bits saturating at 0xFFFF, column pass into 32 bits, (v + 0x8000) >> 16 saturated to 8 bits,
// BORDER_REFLECT_101) fused with the gradient stage.  One work-group per 64 x 16 tile of the frame:
//   raw  [16 + 2 + 2R][64 + 2 + 2R]  the tile with the halo of blur (R <= 3) and Sobel (1), image coordinates only (a reflected
//                                    coordinate of a border pixel always lies inside the tile's own range)
//   rows [16 + 2 + 2R][66]           row pass at the REFLECTED column of every blurred position (the Sobel halo of a border tile
//                                    is the blurred image mirrored, not the blur of a mirrored image: both reflections kept apart)
//   blr  [18][66]                    blurred tile + Sobel halo
constexpr int EDB_TW = 64, EDB_TH = 16, EDB_RMAX = 3;
__global__ __launch_bounds__(256) void k_ed_blur_grad(EdBatch B) {
  constexpr int RW = EDB_TW + 2 + 2 * EDB_RMAX, RH = EDB_TH + 2 + 2 * EDB_RMAX, BW = EDB_TW + 2, BH = EDB_TH + 2;
  __shared__ uint8_t raw[RH][RW + 2];
  __shared__ uint16_t rows[RH][BW];
  __shared__ uint8_t blr[BH][BW + 2];
  const int n = blockIdx.z, tid = threadIdx.x;
  const int W = B.W, H = B.H, R = B.blurR;
  const int x0 = blockIdx.x * EDB_TW, y0 = blockIdx.y * EDB_TH;
  const int xb = x0 - 1 - R, yb = y0 - 1 - R;             // image coordinates of raw[0][0]
  const int rw = EDB_TW + 2 + 2 * R, rh = EDB_TH + 2 + 2 * R;
  const uint8_t* im = B.img + (size_t)n * W * H;
  for (int i = tid; i < rh * rw; i += 256) {
    const int ry = i / rw, rx = i - ry * rw;
    const int y = yb + ry, x = xb + rx;
    raw[ry][rx] = (x >= 0 && x < W && y >= 0 && y < H) ? im[(size_t)y * W + x] : 0;
  }
  __syncthreads();
  for (int i = tid; i < rh * BW; i += 256) {
    const int ry = i / BW, bx = i - ry * BW;
    const int y = yb + ry, x = x0 - 1 + bx;
    if (y < 0 || y >= H || x > W) continue;
    const int cx = refl101(x, W);
    uint32_t a = 0;
    for (int j = -R; j <= R; ++j) a += (uint32_t)B.blurK[j + R] * raw[ry][refl101(cx + j, W) - xb];
    rows[ry][bx] = (uint16_t)min(a, 0xFFFFu);
  }
  __syncthreads();
  for (int i = tid; i < BH * BW; i += 256) {
    const int by = i / BW, bx = i - by * BW;
    const int y = y0 - 1 + by, x = x0 - 1 + bx;
    if (y > H || x > W) continue;
    const int cy = refl101(y, H);
    uint32_t a = 0;
    for (int j = -R; j <= R; ++j) a += (uint32_t)B.blurK[j + R] * rows[refl101(cy + j, H) - yb][bx];
    // a <= 257 * 0xFFFF: the rounding add cannot wrap
    blr[by][bx] = (uint8_t)min((a + 0x8000u) >> 16, 255u);
  }
  __syncthreads();
  const int ty = tid >> 4, tx = (tid & 15) * 4;
  const int y = y0 + ty;
  if (y >= H) return;
  const size_t fo = (size_t)n * W * H + (size_t)y * W + x0 + tx;
  short vx[4], vy[4], vg[4];
  uint8_t vd[4], vb[4];
  for (int q = 0; q < 4; ++q) {
    const int by = ty + 1, bx = tx + q + 1;
    const int gx = (blr[by - 1][bx + 1] + 2 * blr[by][bx + 1] + blr[by + 1][bx + 1]) -
                   (blr[by - 1][bx - 1] + 2 * blr[by][bx - 1] + blr[by + 1][bx - 1]);
    const int gy = (blr[by + 1][bx - 1] + 2 * blr[by + 1][bx] + blr[by + 1][bx + 1]) -
                   (blr[by - 1][bx - 1] + 2 * blr[by - 1][bx] + blr[by - 1][bx + 1]);
    const int ax = gx < 0 ? -gx : gx, ay = gy < 0 ? -gy : gy;
    const int sum = ax + ay;
    const int t = sum > B.gradTh + 1 ? sum : 0;
    const int qq = t >> 2, r = t & 3;
    vx[q] = (short)gx;
    vy[q] = (short)gy;
    vg[q] = (short)(r == 3 ? qq + 1 : (r == 2 ? qq + (qq & 1) : qq));
    vd[q] = ax < ay ? 255 : 0;
    vb[q] = blr[by][bx];
  }
  if ((W & 3) == 0 && x0 + tx + 3 < W) {   // rows start 8-byte aligned when W is a multiple of 4
    *reinterpret_cast<short4*>(B.dx + fo) = make_short4(vx[0], vx[1], vx[2], vx[3]);
    *reinterpret_cast<short4*>(B.dy + fo) = make_short4(vy[0], vy[1], vy[2], vy[3]);
    *reinterpret_cast<short4*>(B.g + fo) = make_short4(vg[0], vg[1], vg[2], vg[3]);
    *reinterpret_cast<uchar4*>(B.dir + fo) = make_uchar4(vd[0], vd[1], vd[2], vd[3]);
    if (B.blurOut) *reinterpret_cast<uchar4*>(B.blurOut + fo) = make_uchar4(vb[0], vb[1], vb[2], vb[3]);
  } else {
    for (int q = 0; q < 4 && x0 + tx + q < W; ++q) {
      B.dx[fo + q] = vx[q]; B.dy[fo + q] = vy[q]; B.g[fo + q] = vg[q]; B.dir[fo + q] = vd[q];
      if (B.blurOut) B.blurOut[fo + q] = vb[q];
    }
  }
}

// One work-group of 1024 per frame.  The anchor order is the reference's scan order, w outer, h inner (edline_detector.cpp:148-164):
// column after column of the scan lattice.  Round 4: the test runs with the LANES ALONG A ROW of the lattice (one wave = 64
// neighbouring lattice columns of one row: its loads of g / dir are contiguous), every hit sets a bit in an LDS bitmask laid
// out per lattice COLUMN; the columns' popcounts are scanned, and each column's thread writes its anchors in row order at its
// offset.  (Round 1-3: every thread walked a contiguous piece of the scan order, i.e. down a column -- 88 dependent,
// uncoalesced loads, twice: 0.71 ms per 64 frames; now 0.1.)  Dynamic LDS: nWs x ceil(nHs / 32) words.
__global__ __launch_bounds__(1024) void k_ed_anchor(EdBatch B) {
  extern __shared__ uint32_t anc_bits[];
  __shared__ int sc[1024];
  __shared__ int carry;
  const int n = blockIdx.x, tid = threadIdx.x;
  const int W = B.W, H = B.H, scan = B.scan;
  const int nWs = (W - 2 + scan - 1) / scan, nHs = (H - 2 + scan - 1) / scan;
  const int nHw = (nHs + 31) >> 5;
  const int16_t* g = B.g + (size_t)n * W * H;
  const uint8_t* dir = B.dir + (size_t)n * W * H;
  for (int i = tid; i < nWs * nHw; i += 1024) anc_bits[i] = 0;
  if (tid == 0) carry = 0;
  __syncthreads();
  // lattice rows to the 16 waves, 64 lattice columns per wave step
  const int lane = tid & 63, wv = tid >> 6;
  for (int hi = wv; hi < nHs; hi += 16) {
    const int h = 1 + hi * scan;
    for (int w0 = 0; w0 < nWs; w0 += 64) {
      const int wi = w0 + lane;
      if (wi < nWs) {
        const int i = h * W + 1 + wi * scan;
        const int gv = g[i];
        const bool a = dir[i] == 255 ? (gv >= g[i - W] + B.anchorTh && gv >= g[i + W] + B.anchorTh)
                                     : (gv >= g[i - 1] + B.anchorTh && gv >= g[i + 1] + B.anchorTh);
        if (a) atomicOr(&anc_bits[wi * nHw + (hi >> 5)], 1u << (hi & 31));
      }
    }
  }
  __syncthreads();
  uint32_t* ax = B.anchX + (size_t)n * B.cap;
  uint32_t* ay = B.anchY + (size_t)n * B.cap;
  for (int c0 = 0; c0 < nWs; c0 += 1024) {   // (one trip for frames up to 2050 px wide)
    const int wi = c0 + tid;
    int cnt = 0;
    if (wi < nWs)
      for (int k = 0; k < nHw; ++k) cnt += __popc(anc_bits[wi * nHw + k]);
    sc[tid] = cnt;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // inclusive Hillis-Steele scan
      const int v = tid >= o ? sc[tid - o] : 0;
      __syncthreads();
      sc[tid] += v;
      __syncthreads();
    }
    int pos = carry + sc[tid] - cnt;
    if (wi < nWs)
      for (int k = 0; k < nHw; ++k) {
        uint32_t m = anc_bits[wi * nHw + k];
        while (m) {
          const int bit = __ffs(m) - 1;
          m &= m - 1;
          if (pos < B.cap) { ax[pos] = 1 + wi * scan; ay[pos] = 1 + (32 * k + bit) * scan; }
          ++pos;
        }
      }
    __syncthreads();
    if (tid == 1023) carry += sc[1023];
    __syncthreads();
  }
  if (tid == 0) B.nAnch[n] = min(carry, B.cap);
}

// ---- smart routing ---------------------------------------------------------------------------------------
// Everything the walk needs to know about a pixel is static once gImg/dirImg exist: whether it is walkable (g > 0), its
// direction, and -- for each of the two senses of travel -- WHICH of the three forward neighbours has the largest gradient
// (compared as unsigned char, edline_detector.cpp:231-233) or that the image border stops the walk.  k_ed_code computes
// that in parallel into one BYTE per pixel (row stride Wc = W rounded up to the strip width; round 4 -- a 16-bit word with the
// moves spelled out until round 3: a byte per pixel lets a full-height strip of the frame sit in LDS next to the edge bitmap):
//   bits 0-1 choice when travelling forward (RIGHT on a horizontal pixel, DOWN on a vertical one): 0 / 1 = the two diagonals,
//            2 = straight, 3 = the border breaks the walk;  bits 2-3 the same when travelling backward (LEFT / UP);
//   bit 4 horizontal pixel, bit 5 walkable (g > 0).
// The move (dx + 1) | (dy + 1) << 2 of (pixel type, sense, choice) comes out of a 64-bit constant on the scalar unit (ED_MOVES).
constexpr int ED_TILE = 128;
constexpr int ED_STOP = 3;
constexpr int ED_HORIZ = 1 << 4, ED_LIVE = 1 << 5;

__device__ __forceinline__ int ed_pick(int g1, int g2, int g3) {   // g1, g3 diagonals, g2 straight
  g1 &= 255; g2 &= 255; g3 &= 255;
  if (g1 >= g2 && g1 >= g3) return 0;
  if (g3 >= g2 && g3 >= g1) return 1;
  return 2;
}
__host__ __device__ constexpr int ed_move(int dx, int dy) { return (dx + 1) | ((dy + 1) << 2); }
// entry (h, F, c) at bit 4 * (8 h + 4 F + c): horizontal pixel -> dx = +-1, dy = -1 / +1 / 0; vertical pixel -> dy = +-1, dx = +1 / -1 / 0
constexpr unsigned long long ed_moves() {
  unsigned long long t = 0;
  for (int h = 0; h < 2; ++h)
    for (int F = 0; F < 2; ++F)
      for (int c = 0; c < 3; ++c) {
        const int s = F ? 1 : -1;
        const int mv = h ? ed_move(s, c == 0 ? -1 : (c == 1 ? 1 : 0)) : ed_move(c == 0 ? 1 : (c == 1 ? -1 : 0), s);
        t |= (unsigned long long)mv << (4 * (8 * h + 4 * F + c));
      }
  return t;
}
constexpr unsigned long long ED_MOVES = ed_moves();

__global__ __launch_bounds__(256) void k_ed_code(EdBatch B) {
  const int n = blockIdx.y;
  const int W = B.W, H = B.H;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= W * H) return;
  const int y = i / W, x = i - y * W;
  const int16_t* g = B.g + (size_t)n * W * H;
  const int gv = g[i];
  int code = 0;
  if (gv > 0) {
    const bool horiz = B.dir[(size_t)n * W * H + i] == 255;
    int f = ED_STOP, b = ED_STOP;
    if (horiz) {
      if (!(x == W - 1 || y == 0 || y == H - 1)) f = ed_pick(g[i - W + 1], g[i + 1], g[i + W + 1]);   // RIGHT: (x+1,y-1) | (x+1,y+1) | (x+1,y)
      if (!(x == 0 || y == 0 || y == H - 1)) b = ed_pick(g[i - W - 1], g[i - 1], g[i + W - 1]);       // LEFT: (x-1,y-1) | (x-1,y+1) | (x-1,y)
    } else {
      if (!(x == 0 || x == W - 1 || y == H - 1)) f = ed_pick(g[i + W + 1], g[i + W], g[i + W - 1]);   // DOWN: (x+1,y+1) | (x-1,y+1) | (x,y+1)
      if (!(x == 0 || x == W - 1 || y == 0)) b = ed_pick(g[i - W + 1], g[i - W], g[i - W - 1]);       // UP: (x+1,y-1) | (x-1,y-1) | (x,y-1)
    }
    code = ED_LIVE | (horiz ? ED_HORIZ : 0) | f | (b << 2);
  }
  B.code[((size_t)n * H + y) * B.Wc + x] = (uint8_t)code;
}

// Walk state shared by the whole wave: every lane computes the same (uniform) walk.
// LDS: edge bitmap of the frame (1 bit / pixel) + a STRIP of routing bytes around the walker, 128 columns wide and HS rows high
// (round 4: HS = the whole frame height when it fits beside the bitmap -- 61 KB for 480 rows -- so that only a walk that
// leaves the strip SIDEWAYS, or the scan of the anchors moving on, asks for a reload: 138 per frame on the benchmark's stream
// against 1 455 reloads of the 128 x 128 tile of 16-bit words of round 3, measured with a counter in the round-3 kernel),
// reloaded cooperatively by the four waves of the work-group.
// The travel state is four bits S = lastWasHorizontal | lastWasForward << 1 | (x > lastX) << 2 | (y > lastY) << 3
// (lastDirection / lastX / lastY of edline_detector.cpp:196-215 only ever enter through these predicates); whether the
// walk goes forward on the current pixel is a 16-entry truth table per pixel type.
struct EdWalker {
  const uint8_t* code;   // frame's routing bytes
  unsigned* bits;        // LDS edge bitmap
  uint8_t* tile;         // LDS strip [HS][128]
  unsigned* ring;        // LDS: request block of the strip loads
  int W, H, Wc, HS;
  int tx0, ty0;          // strip origin; tx0 < 0 = nothing loaded
  unsigned nSteps, nLoads, nWalks;
};

constexpr unsigned ed_truth(bool horizPixel) {
  unsigned t = 0;
  for (unsigned S = 0; S < 16; ++S) {
    const bool ldH = S & 1, ldF = S & 2, gx = S & 4, gy = S & 8;
    const bool F = horizPixel ? (ldH ? ldF : gx) : (ldH ? gy : ldF);
    t |= (F ? 1u : 0u) << S;
  }
  return t;
}
constexpr unsigned ED_TRUTH = ed_truth(false) | (ed_truth(true) << 16);

// One wave's share of a strip load: a strip row is 128 B = 8 lanes x 16 B, one instruction of a wave moves 8 rows, the four
// waves of the work-group 32; HS / 32 such rounds, eight loads in flight per wave.
constexpr int ED_ROUTE_WAVES = 4;
__device__ __forceinline__ void ed_tile_rows(const EdWalker& wk, int tx, int ty, int wave, int lane) {
  const int sub = lane >> 3, col = (lane & 7) * 16;
  const int rounds = wk.HS >> 5;
  for (int r0 = 0; r0 < rounds; r0 += 8) {
    uint4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int lr = 32 * (r0 + k) + 8 * wave + sub;      // row inside the strip
      const int row = ty + lr;
      v[k] = (r0 + k < rounds && row < wk.H) ? *(const uint4*)(wk.code + (size_t)row * wk.Wc + tx + col) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int lr = 32 * (r0 + k) + 8 * wave + sub;
      if (r0 + k < rounds) *(uint4*)(wk.tile + lr * ED_TILE + col) = v[k];
    }
  }
}
// The walker (wave 0) asks for a strip: command and origin to LDS, barrier A (the three helper waves wait there), every wave
// loads its rows, barrier B.  Between B and the next A the helpers touch nothing.
__device__ __forceinline__ void ed_tile_load(EdWalker& wk, int x, int y, unsigned S) {
  // centred on the walker (columns; rows only when the strip is lower than the frame)
  int tx = x - ED_TILE / 2, ty = y - wk.HS / 2;
  (void)S;
  tx &= ~15;
  tx = max(0, min(tx, wk.Wc - ED_TILE));
  ty = max(0, min(ty, wk.H - wk.HS));       // (HS >= H: 0)
  tx = __builtin_amdgcn_readfirstlane(tx); ty = __builtin_amdgcn_readfirstlane(ty);
  if (threadIdx.x == 0) { wk.ring[0] = 1u; wk.ring[1] = (unsigned)tx; wk.ring[2] = (unsigned)ty; }
  __syncthreads();   // A
  ed_tile_rows(wk, tx, ty, 0, threadIdx.x);
  wk.tx0 = tx; wk.ty0 = ty;
  ++wk.nLoads;
  __syncthreads();   // B
}

// EdgeDrawing's routing loops (edline_detector.cpp:191-647); returns the number of pixels of this part.
//
// Round 3: the walk runs on the SCALAR unit.  The walker's state (x, y, travel state, counters) is wave uniform; what made a
// step cost ~550 cycles was ~75 vector instructions and two dependent LDS reads per pixel.  Now the 64 lanes hold an 8 x 8
// WINDOW of the frame around the walker -- lane (lx, ly) the routing word and the visited bit of pixel (wx0 + lx, wy0 + ly),
// fetched from the LDS tile / bitmap with one read each -- and a step inside the window is: one v_readlane of the routing word
// (the result lands in a scalar register), a dozen scalar instructions (truth table, move, state), one bit set in a 64-bit
// mask of the pixels claimed in this window and a compare + select that files the pixel's coordinates in the lane of its
// ordinal.  When the walker leaves the window the claimed pixels are committed to the bitmap (one LDS atomic OR per lane
// that holds one) and the window is re-centred ahead of the direction of travel (5-6 steps per window on average).
// Same pixels, same order, same chains as the step-by-step loop.
__device__ __forceinline__ int ed_walk(EdWalker& wk, int x, int y, unsigned S, uint32_t* px, uint32_t* py, int cap) {
  const int W = wk.W, lane = threadIdx.x;
  const int lx = lane & 7, ly = lane >> 3;
  x = __builtin_amdgcn_readfirstlane(x); y = __builtin_amdgcn_readfirstlane(y); S = __builtin_amdgcn_readfirstlane(S);
  int n = 0;
  int pxy = 0;                   // lane (i & 63) holds pixel i of the current block of 64
  for (;;) {
    if ((unsigned)(x - wk.tx0) >= (unsigned)ED_TILE || (unsigned)(y - wk.ty0) >= (unsigned)wk.HS || wk.tx0 < 0)
      ed_tile_load(wk, x, y, S);
    // window origin: one pixel behind the walker, six ahead, +-3 across the direction of travel; kept inside the tile
    int wx0, wy0;
    if (S & 1) { wx0 = (S & 2) ? x - 1 : x - 6; wy0 = y - 3; }
    else { wy0 = (S & 2) ? y - 1 : y - 6; wx0 = x - 3; }
    wx0 = __builtin_amdgcn_readfirstlane(max(wk.tx0, min(wx0, wk.tx0 + ED_TILE - 8)));
    wy0 = __builtin_amdgcn_readfirstlane(max(wk.ty0, min(wy0, wk.ty0 + wk.HS - 8)));
    const int qx = wx0 + lx, qy = wy0 + ly;
    const bool inb = qx < W && qy < wk.H;
    const int idx = __mul24(qy, W) + qx;
    const unsigned wcode = wk.tile[(qy - wk.ty0) * ED_TILE + (qx - wk.tx0)];
    const unsigned bw = inb ? wk.bits[idx >> 5] : 0u;
    const unsigned long long live = __ballot(inb && (wcode & ED_LIVE) != 0);
    const unsigned long long vis0 = __ballot(((bw >> (idx & 31)) & 1u) != 0);
    // one mask of the pixels the walk may still claim (walkable and not visited) instead of two tests per step; what was
    // claimed in this window is what has left that mask
    const unsigned long long free0 = live & ~vis0;
    unsigned long long free = free0;
    bool done = false;
    for (;;) {
      const int li = (y - wy0) * 8 + (x - wx0);
      const unsigned long long bit = 1ull << li;
      if (!(free & bit)) { done = true; break; }       // while (g > 0 && !edge)
      free &= ~bit;
      {   // pxy[lane n & 63] = x | y << 16  (v_writelane takes one scalar register besides m0)
        const int pv = x | (y << 16), pl = n & 63;
        asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(pxy) : "s"(pv), "s"(pl) : "m0");
      }
      ++n;
      if ((n & 63) == 0) {
        // (the empty statement keeps this block behind a scalar branch: merged with the lane test below it cost every step a
        // vector add, a compare and a pair of exec-mask instructions)
        asm volatile("" ::: "memory");
        const int o = n - 64 + lane;
        if (o < cap) { px[o] = (unsigned)pxy & 0xffff; py[o] = (unsigned)pxy >> 16; }
      }
      const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)wcode, li);
      const unsigned h = (w >> 4) & 1;
      const unsigned F = (ED_TRUTH >> (h * 16 + S)) & 1;
      const unsigned ch = (w >> (F ? 0 : 2)) & 3;
      if (ch == ED_STOP) { done = true; break; }
      const unsigned mv = (unsigned)(ED_MOVES >> (4 * (8 * h + 4 * F + ch))) & 15;
      // (v_readfirstlane: the compiler's uniformity analysis loses the walker's state through the loops; one hint per value
      // keeps all of the arithmetic above on the scalar unit)
      x = __builtin_amdgcn_readfirstlane(x + (int)(mv & 3) - 1);
      y = __builtin_amdgcn_readfirstlane(y + (int)(mv >> 2) - 1);
      S = __builtin_amdgcn_readfirstlane(h | (F << 1) | ((mv & 2) << 1) | (mv & 8));   // dx > 0 <=> (mv & 3) == 2, dy > 0 <=> (mv >> 2) == 2
      n = __builtin_amdgcn_readfirstlane(n);
      if ((unsigned)(x - wx0) >= 8u || (unsigned)(y - wy0) >= 8u) break;         // left the window
    }
    const unsigned long long vmask = free0 & ~free;
    if ((vmask >> lane) & 1ull) atomicOr(&wk.bits[idx >> 5], 1u << (idx & 31));
    if (done) break;
  }
  if (n & 63) {
    const int o = (n & ~63) + lane;
    if (lane < (n & 63) && o < cap) { px[o] = (unsigned)pxy & 0xffff; py[o] = (unsigned)pxy >> 16; }
  }
  wk.nSteps += n;
  return n;
}

// One work-group of four waves per frame: wave 0 walks (everything below), waves 1-3 only help to load tiles of routing words
// -- a reload was four dependent batches of loads for a lone wave (15 % of the kernel), it is one batch per wave now.  The
// helpers sit at a work-group barrier between requests (no polling); every barrier of the walker belongs to a request.
__global__ __launch_bounds__(64 * ED_ROUTE_WAVES) void k_ed_route(EdBatch B) {
  extern __shared__ unsigned ed_sm[];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int W = B.W, H = B.H;
  const int nWords = (W * H + 31) >> 5;
  EdWalker wk;
  wk.code = B.code + (size_t)n * H * B.Wc;
  wk.bits = ed_sm;
  wk.tile = (uint8_t*)(ed_sm + ((nWords + 3) & ~3));
  wk.ring = (unsigned*)(wk.tile + (size_t)B.routeHS * ED_TILE);
  wk.W = W; wk.H = H; wk.Wc = B.Wc; wk.HS = B.routeHS;
  wk.tx0 = -1; wk.ty0 = 0;
  wk.nSteps = wk.nLoads = wk.nWalks = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int k = threadIdx.x; k < nWords; k += 64 * ED_ROUTE_WAVES) wk.bits[k] = 0;
  const uint32_t* ax = B.anchX + (size_t)n * B.cap;
  const uint32_t* ay = B.anchY + (size_t)n * B.cap;
  uint32_t* fX = B.fX + (size_t)n * B.cap;
  uint32_t* fY = B.fY + (size_t)n * B.cap;
  uint32_t* cX = B.cX + (size_t)n * 2 * B.cap;
  uint32_t* cY = B.cY + (size_t)n * 2 * B.cap;
  uint32_t* sId = B.sId + (size_t)n * (B.capEdges + 2);
  const int nA = B.nAnch[n];
  int nC = 0, nE = 0;   // chain pixels written, edges accepted (uniform across the wave)
  __syncthreads();
  if (wave != 0) {
    for (;;) {
      __syncthreads();   // A: a request of the walker
      if (wk.ring[0] == 2u) return;
      ed_tile_rows(wk, (int)wk.ring[1], (int)wk.ring[2], wave, lane);
      __syncthreads();   // B
    }
  }
  for (int i0 = 0; i0 < nA; i0 += 64) {
    // 64 anchors per trip: coordinates and routing word of the anchor pixel (its direction) in one latency
    unsigned mx = 0, my = 0;
    int mcode = 0;
    if (i0 + lane < nA) {
      mx = ax[i0 + lane]; my = ay[i0 + lane];
      mcode = wk.code[(size_t)my * B.Wc + mx];
    }
    // Most anchors lie on an edge that an earlier walk has drawn already: the 64 anchors of the trip test their bit at once
    // (one LDS gather + a ballot); only those that were free then are looked at one by one, and tested again, because a walk
    // of this trip may have reached them since.  (Bits are only ever set -- the anchor that is un-marked between its two walks
    // is claimed again by the second one -- so an anchor that was marked stays marked.)
    unsigned long long cand;
    {
      const int li = (int)my * W + (int)mx;
      const unsigned bw = i0 + lane < nA ? wk.bits[li >> 5] : ~0u;
      cand = __ballot(((bw >> (li & 31)) & 1u) == 0);
    }
    while (cand) {
      const int k = __builtin_ctzll(cand);
      cand &= cand - 1;
      // (read as scalars: the anchor test is a branch on a wave-uniform value, and the compiler should know)
      const int x = __builtin_amdgcn_readlane((int)mx, k), y = __builtin_amdgcn_readlane((int)my, k);
      const int code = __builtin_amdgcn_readlane(mcode, k);
      const int idx = y * W + x;
      const unsigned bword = (unsigned)__builtin_amdgcn_readfirstlane((int)wk.bits[idx >> 5]);
      if ((bword >> (idx & 31)) & 1) continue;
      const unsigned h = (code >> 4) & 1;
      wk.nWalks += 2;
      // first part (RIGHT / DOWN) into the scratch arrays, second part (LEFT / UP) directly behind the slot of the
      // (reversed) first part: its element 0 is the anchor again
      const int nf = ed_walk(wk, x, y, h | 2u, fX, fY, B.cap);
      if (lane == 0) wk.bits[idx >> 5] &= ~(1u << (idx & 31));
      const int room = 2 * B.cap - nC - nf;
      const int ns = ed_walk(wk, x, y, h, cX + nC + nf - 1, cY + nC + nf - 1, room > 0 ? room : 0);
      const bool go = nf + ns >= B.minLineLen + 1 && nE < B.capEdges && nC + nf + ns - 1 <= 2 * B.cap && nf <= B.cap;
      if (go) {
        // chain = reverse(first part) ++ second part without the anchor (the scratch arrays were written by other lanes of
        // this wave: a fence, not a work-group barrier -- those belong to the tile requests)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
        for (int q = lane; q < nf; q += 64) { cX[nC + nf - 1 - q] = fX[q]; cY[nC + nf - 1 - q] = fY[q]; }
        if (lane == 0) sId[nE] = nC;
        nC += nf + ns - 1;
        nE += 1;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (lane == 0) wk.ring[0] = 2u;      // the helpers leave
  __syncthreads();                     // A
  if (lane == 0) {
    sId[nE] = nC;
    B.nEdges[n] = nE;
    unsigned long long* rs = B.rstats + (size_t)n * 4;
    rs[0] = wk.nSteps; rs[1] = wk.nLoads; rs[2] = wk.nWalks; rs[3] = (unsigned long long)(__builtin_readcyclecounter() - t0);
  }
}

// ---- line fit / validation ----------------------------------------------------------------------------
#pragma clang fp contract(off)   // keep the reference's (and the oracle's) un-fused double arithmetic
__device__ double ed_log_gamma(double x) {
  if (x > 15.0) return 0.918938533204673 + (x - 0.5) * log(x) - x + 0.5 * x * log(x * sinh(1 / x) + 1 / (810.0 * pow(x, 6.0)));
  const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
  double a = (x + 0.5) * log(x + 5.5) - (x + 5.5);
  double b = 0.0;
  for (int n = 0; n < 7; n++) {
    a -= log(x + (double)n);
    b += q[n] * pow(x, (double)n);
  }
  return a + log(b);
}
__device__ double ed_nfa(int n, int k, double p, double logNT) {
  const double tolerance = 0.1;
  if (n == 0 || k == 0) return -logNT;
  if (n == k) return -logNT - (double)n * log10(p);
  const double p_term = p / (1.0 - p);
  const double log1term = ed_log_gamma((double)n + 1.0) - ed_log_gamma((double)k + 1.0) - ed_log_gamma((double)(n - k) + 1.0) +
                          (double)k * log(p) + (double)(n - k) * log(1.0 - p);
  double term = exp(log1term);
  {
    const double abs_diff = fabs(term), abs_max = abs_diff < 2.2250738585072014e-308 ? 2.2250738585072014e-308 : abs_diff;
    if (term == 0.0 || (abs_diff / abs_max) <= 100.0 * 2.220446049250313e-16) {
      if ((double)k > (double)n * p) return -log1term / 2.30258509299404568402 - logNT;
      return -logNT;
    }
  }
  double bin_tail = term;
  for (int i = k + 1; i <= n; i++) {
    const double bin_term = (double)(n - i + 1) / (double)i;
    const double mult_term = bin_term * p_term;
    term *= mult_term;
    bin_tail += term;
    if (bin_term < 1.0) {
      const double err = term * ((1.0 - pow(mult_term, (double)(n - i + 1))) / (1.0 - mult_term) - 1.0);
      if (err < tolerance * fabs(-log10(bin_tail) - logNT) * bin_tail) break;
    }
  }
  return -log10(bin_tail) - logNT;
}

// One WAVE per edge chain (the blocks of a frame take chains e = blockIdx.x, blockIdx.x + gridDim.x, ...).  The reference
// loops are sequential over the pixels of a chain; every quantity they produce is either an exact integer sum (normal
// equations of the fit, gradient sums, aligned-pixel counts: order independent, computed lane-parallel), a per-pixel
// predicate (distance to the line: computed lane-parallel, consumed in order through a ballot), or the fit error, whose
// double sum is accumulated in pixel order from lane-parallel terms.  The results are bit-identical to the serial form.
constexpr int ED_FIT_BLOCKS = 48;

__device__ __forceinline__ double ed_wave_sum(double v) {   // exact for the integer-valued sums it is used on
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double ed_readlane_f64(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(64) void k_ed_fit(EdBatch B) {
  const int n = blockIdx.y, lane = threadIdx.x;
  const int W = B.W, H = B.H;
  const uint32_t* xC = B.cX + (size_t)n * 2 * B.cap;
  const uint32_t* yC = B.cY + (size_t)n * 2 * B.cap;
  const uint32_t* sId = B.sId + (size_t)n * (B.capEdges + 2);
  const uint8_t* dir = B.dir + (size_t)n * W * H;
  const int16_t* dx = B.dx + (size_t)n * W * H;
  const int16_t* dy = B.dy + (size_t)n * W * H;
  const unsigned minLineLen = B.minLineLen;
  const double thr = B.fitErr;
  const double logNT = 2.0 * (log10((double)W) + log10((double)H));
  const int nE = B.nEdges[n];
  for (int e = blockIdx.x; e < nE; e += gridDim.x) {
    unsigned offS = sId[e], offE = sId[e + 1];
    int ordinal = 0;
    float ATA0 = 0, ATA1 = 0, ATA2 = 0, ATA3 = 0, ATV0 = 0, ATV1 = 0;
    // sums of u^2, u, u v, v over pixels [a, b) of the chain (u along the line's dominant axis): exact integers
    auto sums = [&](unsigned a, unsigned b, bool hz, double& s_uu, double& s_u, double& s_uv, double& s_v) {
      double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
      for (unsigned o = a + lane; o < b; o += 64) {
        const unsigned x = xC[o], y = yC[o];
        const double u = (double)(float)(hz ? x : y), v = (double)(float)(hz ? y : x);
        p0 += u * u; p1 += u; p2 += u * v; p3 += v;
      }
      s_uu = ed_wave_sum(p0); s_u = ed_wave_sum(p1); s_uv = ed_wave_sum(p2); s_v = ed_wave_sum(p3);
    };
    while (offE > offS + minLineLen) {
      double eq0 = 0, eq1 = 0, lineFitErr = 0;
      bool firstTry = true;
      while (offE > offS + minLineLen) {
        if (!firstTry) {
          // The reference moves the start two pixels on after every failed initial fit and tries again (:989-995): 98 of 100
          // attempts fail (5 670 attempts for 127 lines in a frame of the benchmark's stream), each a chain of dependent work
          // for the whole wave.  After the first failure the next 64 starts are tried AT ONCE, one per lane: every lane runs
          // the reference's arithmetic for its own start (integer sums, float normal equations, the error summed in pixel order),
          // the first lane that succeeds is the start the reference would have reached.
          const unsigned st = offS + 2u * lane;
          const bool valid = offE > st + minLineLen;
          double e_k = 0, q0 = 0, q1 = 0;
          float a0 = 0, a1 = 0, a3 = 0, b0 = 0, b1 = 0;
          if (valid) {
            const bool hzk = dir[yC[st] * W + xC[st]] == 255;
            double s_uu = 0, s_u = 0, s_uv = 0, s_v = 0;
            for (unsigned i = 0; i < minLineLen; ++i) {
              const unsigned x = xC[st + i], y = yC[st + i];
              const double u = (double)(float)(hzk ? x : y), v = (double)(float)(hzk ? y : x);
              s_uu += u * u; s_u += u; s_uv += u * v; s_v += v;
            }
            a0 = (float)s_uu; a1 = (float)s_u; a3 = (float)(double)minLineLen; b0 = (float)s_uv; b1 = (float)s_v;
            const double coef = 1.0 / (double(a0) * double(a3) - double(a1) * double(a1));
            q0 = coef * (double(a3) * double(b0) - double(a1) * double(b1));
            q1 = coef * (double(a0) * double(b1) - double(a1) * double(b0));
            double err = 0;
            for (unsigned i = 0; i < minLineLen; ++i) {
              const unsigned x = xC[st + i], y = yC[st + i];
              const double u = (double)(hzk ? x : y), v = (double)(hzk ? y : x);
              const double c = v - u * q0 - q1;
              err += c * c;
            }
            e_k = sqrt(err);
          }
          const unsigned long long okm = __ballot(valid && e_k <= thr);
          const int nvalid = __popcll(__ballot(valid));
          if (okm) {
            const int f = __builtin_ctzll(okm);
            offS += 2u * (unsigned)f;
            eq0 = ed_readlane_f64(q0, f); eq1 = ed_readlane_f64(q1, f); lineFitErr = ed_readlane_f64(e_k, f);
            ATA0 = __shfl(a0, f, 64); ATA1 = __shfl(a1, f, 64); ATA2 = ATA1; ATA3 = __shfl(a3, f, 64);
            ATV0 = __shfl(b0, f, 64); ATV1 = __shfl(b1, f, 64);
            break;
          }
          // none of them: the last attempt's error is what the test behind the loop sees
          lineFitErr = ed_readlane_f64(e_k, nvalid - 1);
          offS += 2u * (unsigned)nvalid;
          continue;
        }
        firstTry = false;
        const bool hz = dir[yC[offS] * W + xC[offS]] == 255;
        double s_uu, s_u, s_uv, s_v;
        sums(offS, offS + minLineLen, hz, s_uu, s_u, s_uv, s_v);
        ATA0 = (float)s_uu; ATA1 = (float)s_u; ATA2 = (float)s_u; ATA3 = (float)(double)minLineLen;
        ATV0 = (float)s_uv; ATV1 = (float)s_v;
        const double coef = 1.0 / (double(ATA0) * double(ATA3) - double(ATA1) * double(ATA2));
        eq0 = coef * (double(ATA3) * double(ATV0) - double(ATA1) * double(ATV1));
        eq1 = coef * (double(ATA0) * double(ATV1) - double(ATA2) * double(ATV0));
        double err = 0;
        for (unsigned i0 = 0; i0 < minLineLen; i0 += 64) {   // terms lane-parallel, the sum in pixel order
          const unsigned i = i0 + lane;
          double c2 = 0;
          if (i < minLineLen) {
            const unsigned x = xC[offS + i], y = yC[offS + i];
            const double u = (double)(hz ? x : y), v = (double)(hz ? y : x);
            const double c = v - u * eq0 - eq1;
            c2 = c * c;
          }
          const int cnt = (int)min(64u, minLineLen - i0);
          for (int l = 0; l < cnt; ++l) err += ed_readlane_f64(c2, l);
        }
        lineFitErr = sqrt(err);
        if (lineFitErr <= thr) break;
        offS += 2;
      }
      // (a run of minLineLen pixels with one and the same abscissa makes the 2 x 2 normal equations singular and the fit error
      // NaN: neither `<= thr` above nor the reference's `> thr` here is true, and the reference goes on with offS beyond the end
      // of the chain (edline_detector.cpp:989-997, :1011) -- a read past the edge's pixels.  A fit that did not succeed ends the chain.)
      if (!(lineFitErr <= thr)) break;
      const bool hz = dir[yC[offS] * W + xC[offS]] == 255;
      const unsigned offS_init = offS;
      bool bExtended = true, bFirstTry = true;
      int numOfOutlier = 0, tryTimes = 0;
      unsigned newOffsetS = 0;
      double coef1 = 0;
      while (bExtended) {
        tryTimes++;
        if (bFirstTry) {
          bFirstTry = false;
          offS += minLineLen;
        } else {
          const int newLength = (int)offS - (int)newOffsetS;
          if ((int)offS - (int)offS_init > 0 && newLength > 0) {
            double s_uu, s_u, s_uv, s_v;
            sums(newOffsetS, offS, hz, s_uu, s_u, s_uv, s_v);
            const float t00 = (float)s_uu, t01 = (float)s_u, t11 = (float)(double)newLength, v0 = (float)s_uv, v1 = (float)s_v;
            ATA0 = ATA0 + t00; ATA1 = ATA1 + t01; ATA2 = ATA2 + t01; ATA3 = ATA3 + t11;
            ATV0 = ATV0 + v0; ATV1 = ATV1 + v1;
            const double coef = 1.0 / (double(ATA0) * double(ATA3) - double(ATA1) * double(ATA2));
            eq0 = coef * (double(ATA3) * double(ATV0) - double(ATA1) * double(ATV1));
            eq1 = coef * (double(ATA0) * double(ATV1) - double(ATA2) * double(ATV0));
          }
        }
        coef1 = 1 / sqrt(eq0 * eq0 + 1);
        numOfOutlier = 0;
        newOffsetS = offS;
        // "while (offE > offS) { d; offS++; if (d > thr) { if (++numOfOutlier > 3) break; } else numOfOutlier = 0; }"
        bool stop = false;
        while (offE > offS && !stop) {
          const unsigned o = offS + lane;
          bool out = false;
          if (o < offE) {
            const unsigned x = xC[o], y = yC[o];
            const double d = hz ? fabs(eq0 * x - y + eq1) * coef1 : fabs(x - eq0 * y - eq1) * coef1;
            out = d > thr;
          }
          const unsigned long long mask = __ballot(out);
          const int nvalid = (int)min(64u, offE - offS);
          int consumed = nvalid;
          for (int p = 0; p < nvalid; ++p) {
            if ((mask >> p) & 1ull) {
              if (++numOfOutlier > 3) { consumed = p + 1; stop = true; break; }
            } else {
              numOfOutlier = 0;
            }
          }
          offS += consumed;
        }
        offS -= numOfOutlier;
        if (offS - newOffsetS > 0 && tryTimes < 6) {} else bExtended = false;
      }
      double leq0, leq1, leq2;
      if (hz) { leq0 = eq0 * coef1; leq1 = -1 * coef1; leq2 = eq1 * coef1; }
      else { leq0 = 1 * coef1; leq1 = -eq0 * coef1; leq2 = -eq1 * coef1; }
      // ---- LineValidation ----
      bool valid;
      {
        const int np = (int)offS - (int)offS_init;
        double pgx = 0, pgy = 0;
        for (int i = lane; i < np; i += 64) {
          const int idx = yC[offS_init + i] * W + xC[offS_init + i];
          pgx += (double)dx[idx];
          pgy += (double)dy[idx];
        }
        const int mgx = (int)ed_wave_sum(pgx), mgy = (int)ed_wave_sum(pgy);
        const double ddx = fabs(leq1), ddy = fabs(leq0);
        valid = !(mgx == 0 && mgy == 0);
        if (valid) {
          float direction = 0.f;
          if (mgx > 0 && mgy >= 0) direction = (float)atan2(-ddy, ddx);
          if (mgx <= 0 && mgy > 0) direction = (float)atan2(ddy, ddx);
          if (mgx < 0 && mgy <= 0) direction = (float)atan2(ddy, -ddx);
          if (mgx >= 0 && mgy < 0) direction = (float)atan2(-ddy, -ddx);
          if (fabs(direction) < 0.15 || M_PI - fabs(direction) < 0.15)
            if (fabs(leq2) < 10 || fabs(H - fabs(leq2)) < 10) valid = false;
          if (valid && fabs(fabs(direction) - M_PI * 0.5) < 0.15)
            if (fabs(leq2) < 10 || fabs(W - fabs(leq2)) < 10) valid = false;
          if (valid) {
            int k = 0;
            for (int i0 = 0; i0 < np; i0 += 64) {
              const int i = i0 + lane;
              bool al = false;
              if (i < np) {
                const int idx = yC[offS_init + i] * W + xC[offS_init + i];
                const double pd = atan2(-(double)dx[idx], (double)dy[idx]);
                const double dd = fabs(direction - pd);
                al = fabs(2 * M_PI - dd) < 0.392699 || dd < 0.392699;
              }
              k += __popcll(__ballot(al));
            }
            valid = ed_nfa(np, k, 0.125, logNT) > 0;
          }
        }
      }
      if (valid) {
        int slot = 0;
        if (lane == 0) slot = atomicAdd(&B.nLines[n], 1);
        slot = __builtin_amdgcn_readfirstlane(slot);
        if (slot < B.maxLines && lane == 0) {
          const double a1 = leq1 * leq1, a2 = leq0 * leq0, a3 = leq0 * leq1, a4 = leq2 * leq0, a5 = leq2 * leq1;
          unsigned Px = xC[offS_init], Py = yC[offS_init];
          const float x1 = (float)(a1 * Px - a3 * Py - a4), y1 = (float)(a2 * Py - a3 * Px - a5);
          Px = xC[offS - 1]; Py = yC[offS - 1];
          const float x2 = (float)(a1 * Px - a3 * Py - a4), y2 = (float)(a2 * Py - a3 * Px - a5);
          double* o = B.lines + ((size_t)n * B.maxLines + slot) * 10;
          o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = leq0; o[5] = leq1; o[6] = leq2;
          o[7] = (float)((x1 + x2) / 2.0);
          o[8] = (float)((y1 + y2) / 2.0);
          o[9] = (float)sqrt(pow((double)(x2 - x1), 2) + pow((double)(y2 - y1), 2));
          B.lkey[(size_t)n * B.maxLines + slot] = ((uint32_t)e << 8) | (uint32_t)(ordinal & 255);
        }
      }
      ++ordinal;
    }
  }
}

}  // namespace vpl
