// HIP kernels of the EDLines extractor (line front-end), batch of N frames of one size.
//   k_ed_grad   : Sobel 3x3 (BORDER_REFLECT_101) -> |dx|+|dy| -> threshold -> /4 (half-to-even) -> direction
//                 (edline_detector.cpp:125-136); streaming, HBM-bound: 1 byte read, 7 bytes written per pixel
//   k_ed_anchor : anchor test on the scan lattice + ORDERED compaction (w outer, h inner, :148-164)
//   k_ed_route  : smart routing (:166-707): inherently serial and order dependent per frame -> one wave per
//                 frame, lane 0 walks, the wave copies accepted chains; throughput comes from the batch
//   k_ed_fit    : per edge chain least-squares fit / extension / Helmholtz validation (:729-1174), one lane per chain
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
  uint8_t *dir, *edge;     // [N][H][W]
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
  B.edge[o] = 0;
}

// one workgroup of 1024 per frame; thread t owns a contiguous slice of the scan order
__global__ __launch_bounds__(1024) void k_ed_anchor(EdBatch B) {
  const int n = blockIdx.x, tid = threadIdx.x;
  const int W = B.W, H = B.H, scan = B.scan;
  const int nWs = (W - 2 + scan - 1) / scan, nHs = (H - 2 + scan - 1) / scan;
  const int total = nWs * nHs;
  const int chunk = (total + 1023) / 1024;
  const int16_t* g = B.g + (size_t)n * W * H;
  const uint8_t* dir = B.dir + (size_t)n * W * H;
  const int s0 = tid * chunk, s1 = min(total, s0 + chunk);
  auto is_anchor = [&](int s) {
    const int wi = s / nHs, hi = s - wi * nHs;
    const int w = 1 + wi * scan, h = 1 + hi * scan;
    const int i = h * W + w;
    const int gv = g[i];
    if (dir[i] == 255) return gv >= g[i - W] + B.anchorTh && gv >= g[i + W] + B.anchorTh;
    return gv >= g[i - 1] + B.anchorTh && gv >= g[i + 1] + B.anchorTh;
  };
  int cnt = 0;
  for (int s = s0; s < s1; ++s) cnt += is_anchor(s) ? 1 : 0;
  __shared__ int sc[1024];
  sc[tid] = cnt;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {   // inclusive Hillis-Steele scan
    int v = tid >= o ? sc[tid - o] : 0;
    __syncthreads();
    sc[tid] += v;
    __syncthreads();
  }
  int pos = sc[tid] - cnt;
  uint32_t* ax = B.anchX + (size_t)n * B.cap;
  uint32_t* ay = B.anchY + (size_t)n * B.cap;
  for (int s = s0; s < s1; ++s)
    if (is_anchor(s)) {
      const int wi = s / nHs, hi = s - wi * nHs;
      if (pos < B.cap) { ax[pos] = 1 + wi * scan; ay[pos] = 1 + hi * scan; }
      ++pos;
    }
  if (tid == 1023) B.nAnch[n] = min(sc[1023], B.cap);
}

enum { ED_UP = 1, ED_RIGHT = 2, ED_DOWN = 3, ED_LEFT = 4 };

// one walk (the loop body that the reference repeats four times); lane 0 only
__device__ int ed_walk(const int16_t* g, const uint8_t* dir, uint8_t* edge, int W, int H, unsigned x, unsigned y,
                       int lastDirection, unsigned& lastX, unsigned& lastY, uint32_t* px, uint32_t* py, int n, int cap) {
  int idx = y * W + x;
  while (g[idx] > 0 && !edge[idx]) {
    edge[idx] = 1;
    if (n < cap) { px[n] = x; py[n] = y; }
    ++n;
    int shouldGo = 0;
    if (dir[idx] == 255) {
      if (lastDirection == ED_UP || lastDirection == ED_DOWN) shouldGo = x > lastX ? ED_RIGHT : ED_LEFT;
      lastX = x; lastY = y;
      if (lastDirection == ED_RIGHT || shouldGo == ED_RIGHT) {
        if (x == (unsigned)W - 1 || y == 0 || y == (unsigned)H - 1) break;
        const unsigned char g1 = (unsigned char)g[idx - W + 1], g2 = (unsigned char)g[idx + 1], g3 = (unsigned char)g[idx + W + 1];
        if (g1 >= g2 && g1 >= g3) { x = x + 1; y = y - 1; }
        else if (g3 >= g2 && g3 >= g1) { x = x + 1; y = y + 1; }
        else x = x + 1;
        lastDirection = ED_RIGHT;
      } else if (lastDirection == ED_LEFT || shouldGo == ED_LEFT) {
        if (x == 0 || y == 0 || y == (unsigned)H - 1) break;
        const unsigned char g1 = (unsigned char)g[idx - W - 1], g2 = (unsigned char)g[idx - 1], g3 = (unsigned char)g[idx + W - 1];
        if (g1 >= g2 && g1 >= g3) { x = x - 1; y = y - 1; }
        else if (g3 >= g2 && g3 >= g1) { x = x - 1; y = y + 1; }
        else x = x - 1;
        lastDirection = ED_LEFT;
      }
    } else {
      if (lastDirection == ED_RIGHT || lastDirection == ED_LEFT) shouldGo = y > lastY ? ED_DOWN : ED_UP;
      lastX = x; lastY = y;
      if (lastDirection == ED_DOWN || shouldGo == ED_DOWN) {
        if (x == 0 || x == (unsigned)W - 1 || y == (unsigned)H - 1) break;
        const unsigned char g1 = (unsigned char)g[idx + W + 1], g2 = (unsigned char)g[idx + W], g3 = (unsigned char)g[idx + W - 1];
        if (g1 >= g2 && g1 >= g3) { x = x + 1; y = y + 1; }
        else if (g3 >= g2 && g3 >= g1) { x = x - 1; y = y + 1; }
        else y = y + 1;
        lastDirection = ED_DOWN;
      } else if (lastDirection == ED_UP || shouldGo == ED_UP) {
        if (x == 0 || x == (unsigned)W - 1 || y == 0) break;
        const unsigned char g1 = (unsigned char)g[idx - W + 1], g2 = (unsigned char)g[idx - W], g3 = (unsigned char)g[idx - W - 1];
        if (g1 >= g2 && g1 >= g3) { x = x + 1; y = y - 1; }
        else if (g3 >= g2 && g3 >= g1) { x = x - 1; y = y - 1; }
        else y = y - 1;
        lastDirection = ED_UP;
      }
    }
    idx = y * W + x;
  }
  return n;
}

__global__ __launch_bounds__(64) void k_ed_route(EdBatch B) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const int W = B.W, H = B.H;
  const int16_t* g = B.g + (size_t)n * W * H;
  const uint8_t* dir = B.dir + (size_t)n * W * H;
  uint8_t* edge = B.edge + (size_t)n * W * H;
  const uint32_t* ax = B.anchX + (size_t)n * B.cap;
  const uint32_t* ay = B.anchY + (size_t)n * B.cap;
  uint32_t* fX = B.fX + (size_t)n * B.cap;
  uint32_t* fY = B.fY + (size_t)n * B.cap;
  uint32_t* cX = B.cX + (size_t)n * 2 * B.cap;
  uint32_t* cY = B.cY + (size_t)n * 2 * B.cap;
  uint32_t* sId = B.sId + (size_t)n * (B.capEdges + 2);
  const int nA = B.nAnch[n];
  __shared__ int s_nf, s_ns, s_go;
  unsigned lastX = 0, lastY = 0;
  int nC = 0, nE = 0;   // chain pixels written, edges accepted (uniform across the wave)
  for (int i = 0; i < nA; ++i) {
    if (lane == 0) {
      const unsigned x = ax[i], y = ay[i];
      const int idx = y * W + x;
      int nf = 0, ns = 0, go = 0;
      if (!edge[idx]) {
        const bool horiz = dir[idx] == 255;
        // first part into the scratch arrays, second part directly behind the slot of the (reversed) first part
        nf = ed_walk(g, dir, edge, W, H, x, y, horiz ? ED_RIGHT : ED_DOWN, lastX, lastY, fX, fY, 0, B.cap);
        edge[idx] = 0;
        const int room = 2 * B.cap - nC - nf;
        // second part: element 0 is the anchor again; it is written at slot nC + nf - 1 and overwritten... see copy below
        ns = ed_walk(g, dir, edge, W, H, x, y, horiz ? ED_LEFT : ED_UP, lastX, lastY, cX + nC + nf - 1, cY + nC + nf - 1, 0,
                     room > 0 ? room : 0);
        go = (nf + ns >= B.minLineLen + 1 && nE < B.capEdges && nC + nf + ns - 1 <= 2 * B.cap && nf <= B.cap) ? 1 : 0;
      }
      s_nf = nf; s_ns = ns; s_go = go;
    }
    __syncthreads();
    const int nf = s_nf, ns = s_ns, go = s_go;
    if (go) {
      // chain = reverse(first part) ++ second part without the anchor.  The second part already sits at
      // [nC + nf - 1, ...) with its element 0 (the anchor) in the slot of the first part's element 0.
      for (int k = lane; k < nf; k += 64) { cX[nC + nf - 1 - k] = fX[k]; cY[nC + nf - 1 - k] = fY[k]; }
      if (lane == 0) sId[nE] = nC;
      nC += nf + ns - 1;
      nE += 1;
    }
    __syncthreads();
  }
  if (lane == 0) {
    sId[nE] = nC;
    B.nEdges[n] = nE;
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

__global__ __launch_bounds__(64) void k_ed_fit(EdBatch B) {
  const int n = blockIdx.y;
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= B.nEdges[n]) return;
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
  unsigned offS = sId[e], offE = sId[e + 1];
  int ordinal = 0;
  float ATA0, ATA1, ATA2, ATA3, ATV0, ATV1;
  while (offE > offS + minLineLen) {
    double eq0 = 0, eq1 = 0, lineFitErr = 0;
    while (offE > offS + minLineLen) {
      const bool hz = dir[yC[offS] * W + xC[offS]] == 255;
      double s_uu = 0, s_u = 0, s_uv = 0, s_v = 0;
      for (unsigned i = 0; i < minLineLen; ++i) {
        const double u = (double)(float)(hz ? xC[offS + i] : yC[offS + i]);
        const double v = (double)(float)(hz ? yC[offS + i] : xC[offS + i]);
        s_uu += u * u; s_u += u; s_uv += u * v; s_v += v;
      }
      ATA0 = (float)s_uu; ATA1 = (float)s_u; ATA2 = (float)s_u; ATA3 = (float)(double)minLineLen;
      ATV0 = (float)s_uv; ATV1 = (float)s_v;
      const double coef = 1.0 / (double(ATA0) * double(ATA3) - double(ATA1) * double(ATA2));
      eq0 = coef * (double(ATA3) * double(ATV0) - double(ATA1) * double(ATV1));
      eq1 = coef * (double(ATA0) * double(ATV1) - double(ATA2) * double(ATV0));
      double err = 0;
      for (unsigned i = 0; i < minLineLen; ++i) {
        const double u = (double)(hz ? xC[offS + i] : yC[offS + i]), v = (double)(hz ? yC[offS + i] : xC[offS + i]);
        const double c = v - u * eq0 - eq1;
        err += c * c;
      }
      lineFitErr = sqrt(err);
      if (lineFitErr <= thr) break;
      offS += 2;
    }
    if (lineFitErr > thr) break;
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
          double s_uu = 0, s_u = 0, s_uv = 0, s_v = 0;
          for (unsigned o = newOffsetS; o < offS; ++o) {
            const double u = (double)(float)(hz ? xC[o] : yC[o]);
            const double v = (double)(float)(hz ? yC[o] : xC[o]);
            s_uu += u * u; s_u += u; s_uv += u * v; s_v += v;
          }
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
      while (offE > offS) {
        const double d = hz ? fabs(eq0 * xC[offS] - yC[offS] + eq1) * coef1 : fabs(xC[offS] - eq0 * yC[offS] - eq1) * coef1;
        offS++;
        if (d > thr) {
          numOfOutlier++;
          if (numOfOutlier > 3) break;
        } else {
          numOfOutlier = 0;
        }
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
      int mgx = 0, mgy = 0;
      for (int i = 0; i < np; ++i) {
        const int idx = yC[offS_init + i] * W + xC[offS_init + i];
        mgx += dx[idx];
        mgy += dy[idx];
      }
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
          for (int i = 0; i < np; ++i) {
            const int idx = yC[offS_init + i] * W + xC[offS_init + i];
            const double pd = atan2(-(double)dx[idx], (double)dy[idx]);
            const double dd = fabs(direction - pd);
            if (fabs(2 * M_PI - dd) < 0.392699 || dd < 0.392699) k++;
          }
          valid = ed_nfa(np, k, 0.125, logNT) > 0;
        }
      }
    }
    if (valid) {
      const int slot = atomicAdd(&B.nLines[n], 1);
      if (slot < B.maxLines) {
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

}  // namespace vpl
