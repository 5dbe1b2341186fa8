// Vanishing-point stage of the line tracker (feature_tracker/src/vanishing_point_detection.cpp, called from
// line_feature_tracker.cpp:237-266) for a batch of frames.
//   k_vp_grid   (2 waves per frame)  wave 0: getSphereGrids' accumulation (:180-247) -- 64 line pairs per trip evaluated in
//                                    parallel, committed to the 90 x 360 grid in PAIR ORDER per cell (lanes that share a
//                                    cell with an earlier pending lane wait a round; the adds themselves are f64 L2
//                                    atomics, which a wave issues in order) so that every cell is the reference's
//                                    sequential double sum, bit for bit;
//                                    wave 1: the 2-line draws of getVPHypVia2Lines (:107-128) with glibc's rand() stream.
//   k_vp_smooth (grid cells)         3x3 box term of :249-275
//   k_vp_score  (one lane per hypothesis, 105 x 360 per frame)  vp1/vp2/vp3 of :130-176 recomputed from the drawn pair,
//                                    the three grid look-ups of :286-313, block arg-max (first maximum)
//   k_vp_pick   (1 wave per frame)   final arg-max, the vps[1]/vps[2] swap of :318-339, lines2Vps (:347-466): the angle
//                                    of every line to the three VPs lane-parallel, the list logic with its rand() calls
//                                    on one lane (it is sequential by construction).
// The elementary functions are the fixed IEEE-double formulas of oracle/detmath.h (same constants, same operation order):
// the reference's result hangs on values that sit on cell borders, see the oracle's header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vpl {

#pragma clang fp contract(off)

constexpr int VP_IT = 105;            // int(log(1 - 0.9999) / log(1 - (1/3) * 0.5^2)), :96-100
constexpr int VP_NHYP = VP_IT * 360;
constexpr int VP_LA = 90, VP_LO = 360, VP_CELLS = VP_LA * VP_LO;
constexpr int VP_SCORE_BLOCKS = (VP_NHYP + 255) / 256;
constexpr int VP_MAX_DRAWS = 100000;

struct VpBatch {
  int N, maxL;
  const float* hypEnds;   // [N][maxL][4]
  const int* nHyp;        // [N]
  const float* allEnds;   // [N][maxL][4]
  const int* nAll;        // [N]
  const uint32_t* seed;   // [N]
  const int* firstFrame;  // [N]
  double f, ppx, ppy;
  double* g;              // [N][VP_CELLS] raw grid
  double* grid;           // [N][VP_CELLS] smoothed
  int* pairs;             // [N][VP_IT][2]
  uint32_t* rng;          // [N][36]: ring of 34, position, numbers drawn
  int* status;            // [N] 0 ok, -1 no hypothesis possible
  double* partScore;      // [N][VP_SCORE_BLOCKS]
  int* partIdx;           // [N][VP_SCORE_BLOCKS]
  double* vps;            // [N][9]
  int* ids;               // [N][maxL]
  int* bestIdx;           // [N]
};

// ---- elementary functions (oracle/detmath.h) ----
constexpr double VPD_PIO2_HI = 1.57079632673412561417e+00, VPD_PIO2_LO = 6.07710050650619224932e-11;
constexpr double VPD_INV_PIO2 = 6.36619772367581382433e-01;
constexpr double VPD_PI = 3.14159265358979311600e+00, VPD_PI_LO = 1.2246467991473531772e-16;

__device__ inline double vpd_ksin(double x) {
  const double z = x * x;
  const double r = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                   z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  return x + x * z * (-1.66666666666666324348e-01 + z * r);
}
__device__ inline double vpd_kcos(double x) {
  const double z = x * x;
  const double r = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                   z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  return 1.0 - (0.5 * z - z * r);
}
__device__ inline int vpd_reduce(double x, double* r) {
  const double fn = floor(x * VPD_INV_PIO2 + 0.5);
  *r = (x - fn * VPD_PIO2_HI) - fn * VPD_PIO2_LO;
  return (int)fn;
}
__device__ inline double vpd_sin(double x) {
  double r;
  const int n = vpd_reduce(x, &r) & 3;
  return n == 0 ? vpd_ksin(r) : n == 1 ? vpd_kcos(r) : n == 2 ? -vpd_ksin(r) : -vpd_kcos(r);
}
__device__ inline double vpd_cos(double x) {
  double r;
  const int n = vpd_reduce(x, &r) & 3;
  return n == 0 ? vpd_kcos(r) : n == 1 ? -vpd_ksin(r) : n == 2 ? -vpd_kcos(r) : vpd_ksin(r);
}
__device__ inline double vpd_atan(double x) {
  const double ax = x < 0 ? -x : x;
  if (ax != ax) return x;
  double t, hi, lo;
  int id;
  if (ax < 0.4375) { t = ax; id = -1; hi = 0; lo = 0; }
  else if (ax < 0.6875) { t = (2.0 * ax - 1.0) / (2.0 + ax); id = 0; hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
  else if (ax < 1.1875) { t = (ax - 1.0) / (ax + 1.0); id = 1; hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
  else if (ax < 2.4375) { t = (ax - 1.5) / (1.0 + 1.5 * ax); id = 2; hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; }
  else { t = -1.0 / ax; id = 3; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; }
  const double z = t * t, w = z * z;
  const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 +
                    w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
  const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 +
                    w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
  const double res = id < 0 ? t - t * (s1 + s2) : hi - ((t * (s1 + s2) - lo) - t);
  return x < 0 ? -res : res;
}
__device__ inline double vpd_atan2(double y, double x) {
  if (x != x || y != y) return x + y;
  if (y == 0.0) return x < 0 || (x == 0.0 && signbit(x)) ? (signbit(y) ? -VPD_PI : VPD_PI) : y;
  if (x == 0.0) return y < 0 ? -VPD_PIO2_HI - VPD_PIO2_LO : VPD_PIO2_HI + VPD_PIO2_LO;
  const double a = vpd_atan((y < 0 ? -y : y) / (x < 0 ? -x : x));
  const double q = x > 0 ? a : VPD_PI - (a - VPD_PI_LO);
  return y < 0 ? -q : q;
}
__device__ inline double vpd_acos(double x) {
  if (x >= 1.0) return 0.0;
  if (x <= -1.0) return VPD_PI;
  return 2.0 * vpd_atan(sqrt((1.0 - x) / (1.0 + x)));
}

struct VpV3 { double x, y, z; };
__device__ inline VpV3 vp_cross(VpV3 a, VpV3 b) { return VpV3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ inline VpV3 vp_para(const float* e) { return vp_cross(VpV3{e[0], e[1], 1.0}, VpV3{e[2], e[3], 1.0}); }
// segAngle (:22-27): float arguments, the reference's call is atan2f
__device__ inline float vp_seg_angle(const float* e) {
  if (e[2] > e[0]) return (float)vpd_atan2((double)(e[3] - e[1]), (double)(e[2] - e[0]));
  return (float)vpd_atan2((double)(e[1] - e[3]), (double)(e[0] - e[2]));
}

// glibc rand() on a 34-entry ring (random_r.c TYPE_3), one lane
__device__ inline void vp_rng_seed(uint32_t* ring, uint32_t seed) {
  int32_t word = seed ? (int32_t)seed : 1;
  ring[0] = (uint32_t)word;
  for (int i = 1; i < 31; ++i) {
    const long long hi = word / 127773, lo = word % 127773;
    long long w = 16807 * lo - 2836 * hi;
    if (w < 0) w += 2147483647;
    word = (int32_t)w;
    ring[i] = (uint32_t)word;
  }
  for (int i = 31; i < 34; ++i) ring[i] = ring[i - 31];
  // state index n = 34: entry n lives at n % 34
  for (int n = 34; n < 344; ++n) ring[n % 34] = ring[(n - 31) % 34] + ring[(n - 3) % 34];
}
__device__ inline int vp_rng_next(uint32_t* ring, uint32_t& pos) {   // pos = index of the next state entry (mod 34)
  const uint32_t v = ring[(pos + 3) % 34] + ring[(pos + 31) % 34];    // n-31 = n+3, n-3 = n+31 (mod 34)
  ring[pos] = v;
  pos = (pos + 1) % 34;
  return (int)(v >> 1);
}

// number of pairs before row i of the (i < j) enumeration over n lines
__device__ inline long long vp_row_start(int i, int n) { return (long long)i * (2 * n - i - 1) / 2; }

__global__ __launch_bounds__(64) void k_vp_grid(VpBatch B) {
  __shared__ uint32_t ring[34];
  const int fr = blockIdx.x, lane = threadIdx.x;
  const int num = B.nHyp[fr];
  const float* ends = B.hypEnds + (size_t)fr * B.maxL * 4;
  if (blockIdx.y == 1) {
    // ---- the draws of getVPHypVia2Lines ----
    if (lane != 0) return;
    int* pairs = B.pairs + (size_t)fr * VP_IT * 2;
    uint32_t* st = B.rng + (size_t)fr * 36;
    if (num < 2) { B.status[fr] = -1; return; }
    vp_rng_seed(ring, B.seed[fr]);
    uint32_t pos = 344 % 34;
    int drawn = 0, status = 0;
    for (int i = 0; i < VP_IT; ++i) {
      if (drawn > VP_MAX_DRAWS) { status = -1; break; }
      const int idx1 = vp_rng_next(ring, pos) % num;
      int idx2 = vp_rng_next(ring, pos) % num;
      drawn += 2;
      while (idx2 == idx1 && drawn <= VP_MAX_DRAWS) { idx2 = vp_rng_next(ring, pos) % num; ++drawn; }
      if (idx2 == idx1) { status = -1; break; }
      const VpV3 v = vp_cross(vp_para(ends + 4 * idx1), vp_para(ends + 4 * idx2));
      if (v.z == 0) { --i; continue; }
      pairs[2 * i] = idx1;
      pairs[2 * i + 1] = idx2;
    }
    for (int k = 0; k < 34; ++k) st[k] = ring[k];
    st[34] = pos;
    st[35] = (uint32_t)drawn;
    B.status[fr] = status;
    return;
  }
  // ---- sphere grid accumulation ----
  if (num < 2) return;
  double* g = B.g + (size_t)fr * VP_CELLS;
  const double angelAccuracy = 1.0 / 180.0 * VPD_PI;
  const double angelTolerance = 60.0 / 180.0 * VPD_PI;
  const long long nPairs = (long long)num * (num - 1) / 2;
  for (long long base = 0; base < nPairs; base += 64) {
    const long long p = base + lane;
    int cell = -1;
    double val = 0.0;
    if (p < nPairs) {
      const double tn = 2.0 * num - 1.0;
      int i = (int)((tn - sqrt(tn * tn - 8.0 * (double)p)) * 0.5);
      i = min(max(i, 0), num - 2);
      while (i + 1 <= num - 2 && vp_row_start(i + 1, num) <= p) ++i;
      while (i > 0 && vp_row_start(i, num) > p) --i;
      const int j = i + 1 + (int)(p - vp_row_start(i, num));
      const float* ei = ends + 4 * i;
      const float* ej = ends + 4 * j;
      const VpV3 pt = vp_cross(vp_para(ei), vp_para(ej));
      if (pt.z != 0) {
        const double x = pt.x / pt.z, y = pt.y / pt.z;
        const double X = x - B.ppx, Y = y - B.ppy, Z = B.f;
        const double N = sqrt(X * X + Y * Y + Z * Z);
        const double latitude = vpd_acos(Z / N);
        const double longitude = vpd_atan2(X, Y) + VPD_PI;
        int LA = (int)(latitude / angelAccuracy);
        if (LA >= VP_LA) LA = VP_LA - 1;
        int LO = (int)(longitude / angelAccuracy);
        if (LO >= VP_LO) LO = VP_LO - 1;
        // lineinfo (:69-92) of the two lines: dx = x1 - y1, dy = x2 - y2 as written there
        const double dxi = ei[0] - ei[1], dyi = ei[2] - ei[3], dxj = ej[0] - ej[1], dyj = ej[2] - ej[3];
        double oi = vpd_atan2(dyi, dxi), oj = vpd_atan2(dyj, dxj);
        if (oi < 0) oi += VPD_PI;
        if (oj < 0) oj += VPD_PI;
        double angleDev = fabs(oi - oj);
        angleDev = fmin(VPD_PI - angleDev, angleDev);
        if (!(angleDev > angelTolerance) && LA >= 0 && LO >= 0) {
          const double li = sqrt(dxi * dxi + dyi * dyi), lj = sqrt(dxj * dxj + dyj * dyj);
          cell = LA * VP_LO + LO;
          val = sqrt(li * lj) * (vpd_sin(2.0 * angleDev) + 0.2);
        }
      }
    }
    // lower lanes that hit the same cell
    unsigned long long dupLower = 0;
    for (int l = 0; l < 63; ++l) {
      const int c = __shfl(cell, l);
      if (l < lane && c == cell && cell >= 0) dupLower |= 1ull << l;
    }
    unsigned long long pending = __ballot(cell >= 0);
    while (pending) {
      const bool ready = ((pending >> lane) & 1) && !(dupLower & pending);
      if (ready) atomicAdd(&g[cell], val);
      pending &= ~__ballot(ready);
    }
  }
}

__global__ __launch_bounds__(256) void k_vp_smooth(VpBatch B) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= VP_CELLS) return;
  const int fr = blockIdx.y;
  const double* g = B.g + (size_t)fr * VP_CELLS;
  const int i = c / VP_LO, j = c - i * VP_LO;
  double out = 0.0;
  if (i >= 1 && i < VP_LA - 1 && j >= 1 && j < VP_LO - 1) {
    double neighborTotal = 0.0;
    for (int m = 0; m < 3; ++m)
      for (int n = 0; n < 3; ++n) neighborTotal += g[(i - 1 + m) * VP_LO + (j - 1 + n)];
    out = g[c] + neighborTotal / 9;
  }
  B.grid[(size_t)fr * VP_CELLS + c] = out;
}

// hypothesis h of frame fr: vp1 | vp2 | vp3 (:130-176)
__device__ inline void vp_hypothesis(const VpBatch& B, int fr, int h, double* v9) {
  const int i = h / 360, j = h - i * 360;
  const int* pr = B.pairs + ((size_t)fr * VP_IT + i) * 2;
  const float* ends = B.hypEnds + (size_t)fr * B.maxL * 4;
  const VpV3 img = vp_cross(vp_para(ends + 4 * pr[0]), vp_para(ends + 4 * pr[1]));
  VpV3 vp1{img.x / img.z - B.ppx, img.y / img.z - B.ppy, B.f};
  if (vp1.z == 0) vp1.z = 0.0011;
  double N = sqrt(vp1.x * vp1.x + vp1.y * vp1.y + vp1.z * vp1.z);
  vp1.x *= 1.0 / N; vp1.y *= 1.0 / N; vp1.z *= 1.0 / N;
  const double stepVp2 = 2.0 * VPD_PI / 360;
  const double lambda = j * stepVp2;
  const double sl = vpd_sin(lambda), cl = vpd_cos(lambda);
  const double k1 = vp1.x * sl + vp1.y * cl;
  const double k2 = vp1.z;
  const double phi = vpd_atan(-k2 / k1);
  const double sp = vpd_sin(phi);
  VpV3 vp2{sp * sl, sp * cl, vpd_cos(phi)};
  if (vp2.z == 0.0) vp2.z = 0.0011;
  N = sqrt(vp2.x * vp2.x + vp2.y * vp2.y + vp2.z * vp2.z);
  vp2.x *= 1.0 / N; vp2.y *= 1.0 / N; vp2.z *= 1.0 / N;
  if (vp2.z < 0) { vp2.x *= -1.0; vp2.y *= -1.0; vp2.z *= -1.0; }
  VpV3 vp3 = vp_cross(vp1, vp2);
  if (vp3.z == 0.0) vp3.z = 0.0011;
  N = sqrt(vp3.x * vp3.x + vp3.y * vp3.y + vp3.z * vp3.z);
  vp3.x *= 1.0 / N; vp3.y *= 1.0 / N; vp3.z *= 1.0 / N;
  if (vp3.z < 0) { vp3.x *= -1.0; vp3.y *= -1.0; vp3.z *= -1.0; }
  v9[0] = vp1.x; v9[1] = vp1.y; v9[2] = vp1.z;
  v9[3] = vp2.x; v9[4] = vp2.y; v9[5] = vp2.z;
  v9[6] = vp3.x; v9[7] = vp3.y; v9[8] = vp3.z;
}

__global__ __launch_bounds__(256) void k_vp_score(VpBatch B) {
  __shared__ double sS[4];
  __shared__ int sI[4];
  const int fr = blockIdx.y, t = threadIdx.x;
  const int h = blockIdx.x * 256 + t;
  double score = -1.0;
  int idx = 0x7fffffff;
  if (B.status[fr] == 0 && h < VP_NHYP) {
    double v[9];
    vp_hypothesis(B, fr, h, v);
    const double* grid = B.grid + (size_t)fr * VP_CELLS;
    const double oneDegree = 1.0 / 180.0 * VPD_PI;
    double len = 0.0;
    for (int j = 0; j < 3; ++j) {
      if (v[3 * j + 2] == 0.0) continue;
      const double latitude = vpd_acos(v[3 * j + 2]);
      const double longitude = vpd_atan2(v[3 * j], v[3 * j + 1]) + VPD_PI;
      int la = (int)(latitude / oneDegree);
      if (la == 90) la = 89;
      int lo = (int)(longitude / oneDegree);
      if (lo == 360) lo = 359;
      la = min(max(la, 0), VP_LA - 1);          // (a NaN would index outside in the reference)
      lo = min(max(lo, 0), VP_LO - 1);
      len += grid[la * VP_LO + lo];
    }
    score = len;
    idx = h;
  }
  // arg-max, smallest index among equal scores (= the reference's first strict maximum)
  for (int o = 32; o; o >>= 1) {
    const double s2 = __shfl_xor(score, o);
    const int i2 = __shfl_xor(idx, o);
    if (s2 > score || (s2 == score && i2 < idx)) { score = s2; idx = i2; }
  }
  if ((t & 63) == 0) { sS[t >> 6] = score; sI[t >> 6] = idx; }
  __syncthreads();
  if (t == 0) {
    for (int w = 1; w < 4; ++w)
      if (sS[w] > score || (sS[w] == score && sI[w] < idx)) { score = sS[w]; idx = sI[w]; }
    B.partScore[(size_t)fr * VP_SCORE_BLOCKS + blockIdx.x] = score;
    B.partIdx[(size_t)fr * VP_SCORE_BLOCKS + blockIdx.x] = idx;
  }
}

// dynamic LDS: maxL * (3 doubles + 1 float) + 3 * maxL ints
__global__ __launch_bounds__(64) void k_vp_pick(VpBatch B) {
  extern __shared__ double vp_sm[];
  __shared__ uint32_t ring[34];
  __shared__ double sVps[9];
  const int fr = blockIdx.x, lane = threadIdx.x;
  const int num = B.nAll[fr];
  int* ids = B.ids + (size_t)fr * B.maxL;
  double* vout = B.vps + (size_t)fr * 9;
  if (B.status[fr] != 0) {
    for (int i = lane; i < num; i += 64) ids[i] = 3;
    if (lane < 9) vout[lane] = 0.0;
    if (lane == 0) B.bestIdx[fr] = -1;
    return;
  }
  double* ang = vp_sm;                              // [maxL][3]
  float* seg = (float*)(ang + 3 * (size_t)B.maxL);  // [maxL]
  int* lx = (int*)(seg + B.maxL);                   // [maxL] each
  int* ly = lx + B.maxL;
  int* lz = ly + B.maxL;
  // final arg-max over the blocks of k_vp_score
  double score = -1.0;
  int idx = 0x7fffffff;
  for (int b = lane; b < VP_SCORE_BLOCKS; b += 64) {
    const double s2 = B.partScore[(size_t)fr * VP_SCORE_BLOCKS + b];
    const int i2 = B.partIdx[(size_t)fr * VP_SCORE_BLOCKS + b];
    if (s2 > score || (s2 == score && i2 < idx)) { score = s2; idx = i2; }
  }
  for (int o = 32; o; o >>= 1) {
    const double s2 = __shfl_xor(score, o);
    const int i2 = __shfl_xor(idx, o);
    if (s2 > score || (s2 == score && i2 < idx)) { score = s2; idx = i2; }
  }
  if (!(score > 0.0)) idx = 0;                      // maxLength starts at 0.0 with bestIdx = 0 (:316-325)
  if (lane == 0) {
    double v[9];
    vp_hypothesis(B, fr, idx, v);
    if (!B.firstFrame[fr] && !(fabs(v[4]) > 0.8))   // row_f == 1 always, row_v = |vps[1].y| > 0.8 ? 1 : 2 (:318-339)
      for (int c = 0; c < 3; ++c) { const double t = v[3 + c]; v[3 + c] = v[6 + c]; v[6 + c] = t; }
    for (int c = 0; c < 9; ++c) { sVps[c] = v[c]; vout[c] = v[c]; }
    B.bestIdx[fr] = idx;
  }
  __syncthreads();
  // lines2Vps: the angles first, lane-parallel
  const float* ends = B.allEnds + (size_t)fr * B.maxL * 4;
  double vp2D[3][2];
  for (int i = 0; i < 3; ++i) {
    vp2D[i][0] = sVps[3 * i] * B.f / sVps[3 * i + 2] + B.ppx;
    vp2D[i][1] = sVps[3 * i + 1] * B.f / sVps[3 * i + 2] + B.ppy;
  }
  for (int i = lane; i < num; i += 64) {
    const float* e = ends + 4 * i;
    const double x1 = e[0], y1 = e[1], x2 = e[2], y2 = e[3];
    const double xm = (x1 + x2) / 2.0, ym = (y1 + y2) / 2.0;
    double v1x = x1 - x2, v1y = y1 - y2;
    const double N1 = sqrt(v1x * v1x + v1y * v1y);
    v1x /= N1; v1y /= N1;
    for (int j = 0; j < 3; ++j) {
      double v2x = vp2D[j][0] - xm, v2y = vp2D[j][1] - ym;
      const double N2 = sqrt(v2x * v2x + v2y * v2y);
      v2x /= N2; v2y /= N2;
      double crossValue = v1x * v2x + v1y * v2y;
      if (crossValue > 1.0) crossValue = 1.0;
      if (crossValue < -1.0) crossValue = -1.0;
      double angle = vpd_acos(crossValue);
      angle = fmin(VPD_PI - angle, angle);
      ang[3 * i + j] = angle;
    }
    seg[i] = vp_seg_angle(e);
  }
  __syncthreads();
  if (lane != 0) return;
  const uint32_t* st = B.rng + (size_t)fr * 36;
  for (int k = 0; k < 34; ++k) ring[k] = st[k];
  uint32_t pos = st[34];
  const double thAngle = 1.0 / 180.0 * VPD_PI;
  int nx = 0, ny = 0, nz = 0;
  for (int i = 0; i < num; ++i) {
    double minAngle = 1000;
    int bestIdx = 0;
    for (int j = 0; j < 3; ++j) {
      const double angle = ang[3 * i + j];
      if (!(angle < minAngle)) continue;
      bool flag = false;
      const int sized = j == 0 ? ny : j == 1 ? nz : nx;   // the list whose size is tested and drawn from (:405,:425,:441)
      if (sized > 1) {
        const int q = vp_rng_next(ring, pos) % sized;
        if (q < nx) {                                      // the query line is always lx[idx]; past its end: no query
          const float delta_angle = fabsf(seg[i] - seg[lx[q]]);
          if (delta_angle < 0.175) flag = true;
        }
      }
      if (!flag) {
        minAngle = angle;
        bestIdx = j;
        if (j == 0) lx[nx++] = i; else if (j == 1) ly[ny++] = i; else lz[nz++] = i;
      }
    }
    ids[i] = minAngle < thAngle ? bestIdx : 3;
  }
}

}  // namespace vpl
