// TEST INFRASTRUCTURE -- CPU oracle (see smallmat.h).
// Restates the KLT line matcher of the reference's line front-end:
//   LineMatching::Matching            line_matching/src/line_matching.cpp:605-690
//   LineMatching::Anchors             :532-602
//   KLT::calc2D                       line_matching/src/klt.cpp:491-628  (calcSharrDeriv :42-122, getImageNormParams :4-10)
//   LKTrackerInvoker2D::operator()    line_matching/src/lk_tracker_invoker_2d.cpp:28-479  (affines == nullptr branch)
//   LineMatching::ClosestLine :48-86, Point2Line :88-133, TopologicalFilter :266-397, SidenessCheck :399-436
// in the configuration the tracker uses (feature_tracker/src/line_feature_tracker.cpp:299-311):
//   KLT(Size(9,9) -> setWinSize(13,13), maxLevel 3, TermCriteria(COUNT|EPS, 30, 0.001), minEig 1e-4, flags 0).
//
// OpenCV 3.4 (third party, not in the tree) supplies buildOpticalFlowPyramid / pyrDown / copyMakeBorder /
// meanStdDev.  Their arithmetic is restated from the published behaviour:
//   pyrDown CV_8U: separable [1 4 6 4 1], exact integers, dst = (sum + 128) >> 8, size (w+1)/2 x (h+1)/2, BORDER_REFLECT_101;
//   buildOpticalFlowPyramid(img, pyr, winSize, maxLevel, withDerivatives=false): every level padded by winSize with
//     BORDER_REFLECT_101; stops before a level whose width <= winSize.width or height <= winSize.height and returns the
//     index of the last level built;
//   meanStdDev on CV_16S: exact integer sum / sum of squares, mean = s * (1/N), sd = sqrt(max(sq * (1/N) - mean^2, 0)) in double.
// PARITY UNPINNED: the reference holds no numeric fixture for this path.
#include <cmath>
#include <cfloat>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <vector>

namespace orc {

struct LMLine {           // struct Line, line.h:8-17 (numeric part)
  float endpoint[4];
  double equation[3];
  float center[2];
  float length;
};

struct LMParam {          // LineMatching ctor defaults line_matching.h:14-18, TopologicalFilter defaults :45-47
  int step;                         // 10
  float closest_line_threshold;     // 0.5
  float line_matching_ratio;        // 0.4
  float line_distance_error_ratio;  // 3
  float klt_error_threshold;        // 40
  int illumination_adapt;           // true in the tracker
  int topological_filter;           // true in the tracker
  float topo_distance_threshold;    // 15
  float topo_length_tolerate_ratio; // 0.2
  float topo_violation_ratio;       // 0.05
};

constexpr int LM_WIN = 13;          // line_matching.cpp:631
constexpr int LM_MAXLEVEL = 3;      // line_matching.cpp:14
constexpr int LM_MAXCOUNT = 30;
constexpr float LM_MINEIG = 1e-4f;

static inline int r101(int i, int n) {
  if (i < 0) return -i;
  if (i >= n) return 2 * n - 2 - i;
  return i;
}
static inline int cv_round(float v) { return (int)lrintf(v); }   // cvRound: round half to even
static inline int cv_floor(float v) { return (int)floorf(v); }
static inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// one pyramid level stored with a border of LM_WIN pixels on every side
struct Level {
  int w = 0, h = 0, stride = 0;
  std::vector<uint8_t> px;       // (h + 2 WIN) x (w + 2 WIN), BORDER_REFLECT_101
  std::vector<short> d;          // same geometry x 2 (dI/dx, dI/dy interleaved), BORDER_CONSTANT 0
  const uint8_t* at(int x, int y) const { return px.data() + (size_t)(y + LM_WIN) * stride + (x + LM_WIN); }
  const short* dat(int x, int y) const { return d.data() + ((size_t)(y + LM_WIN) * stride + (x + LM_WIN)) * 2; }
};

static void pad_reflect(Level& L, const uint8_t* src) {
  L.stride = L.w + 2 * LM_WIN;
  L.px.assign((size_t)L.stride * (L.h + 2 * LM_WIN), 0);
  for (int y = -LM_WIN; y < L.h + LM_WIN; ++y)
    for (int x = -LM_WIN; x < L.w + LM_WIN; ++x)
      L.px[(size_t)(y + LM_WIN) * L.stride + (x + LM_WIN)] = src[(size_t)r101(y, L.h) * L.w + r101(x, L.w)];
}

// cv::pyrDown for CV_8UC1
void lm_pyr_down(const uint8_t* src, int w, int h, uint8_t* dst) {
  const int dw = (w + 1) / 2, dh = (h + 1) / 2;
  for (int y = 0; y < dh; ++y)
    for (int x = 0; x < dw; ++x) {
      static const int k[5] = {1, 4, 6, 4, 1};
      int s = 0;
      for (int j = 0; j < 5; ++j) {
        const uint8_t* row = src + (size_t)r101(2 * y + j - 2, h) * w;
        int rs = 0;
        for (int i = 0; i < 5; ++i) rs += k[i] * row[r101(2 * x + i - 2, w)];
        s += k[j] * rs;
      }
      dst[(size_t)y * dw + x] = (uint8_t)((s + 128) >> 8);
    }
}

// KLT::calcSharrDeriv klt.cpp:42-122 (3/10/3 smoothing, central difference, reflect-101 inside the image)
void lm_scharr(const uint8_t* src, int w, int h, short* dst /* [h][w][2] */) {
  std::vector<int> t0(w + 2), t1(w + 2);
  for (int y = 0; y < h; ++y) {
    const uint8_t* r0 = src + (size_t)(y > 0 ? y - 1 : h > 1 ? 1 : 0) * w;
    const uint8_t* r1 = src + (size_t)y * w;
    const uint8_t* r2 = src + (size_t)(y < h - 1 ? y + 1 : h > 1 ? h - 2 : 0) * w;
    for (int x = 0; x < w; ++x) {
      t0[x + 1] = (short)((r0[x] + r2[x]) * 3 + r1[x] * 10);
      t1[x + 1] = (short)(r2[x] - r0[x]);
    }
    const int x0 = (w > 1 ? 1 : 0), x1 = (w > 1 ? w - 2 : 0);
    t0[0] = t0[x0 + 1]; t0[w + 1] = t0[x1 + 1];
    t1[0] = t1[x0 + 1]; t1[w + 1] = t1[x1 + 1];
    for (int x = 0; x < w; ++x) {
      dst[((size_t)y * w + x) * 2] = (short)(t0[x + 2] - t0[x]);
      dst[((size_t)y * w + x) * 2 + 1] = (short)((t1[x + 2] + t1[x]) * 3 + t1[x + 1] * 10);
    }
  }
}

// buildOpticalFlowPyramid (levels + borders) and, when `deriv`, the Scharr planes of klt.cpp:602-614
static int build_pyramid(const uint8_t* img, int W, int H, bool deriv, std::vector<Level>& pyr) {
  pyr.clear();
  std::vector<uint8_t> cur(img, img + (size_t)W * H), nxt;
  int w = W, h = H, level = 0;
  for (;;) {
    Level L;
    L.w = w; L.h = h;
    pad_reflect(L, cur.data());
    if (deriv) {
      std::vector<short> d((size_t)w * h * 2);
      lm_scharr(cur.data(), w, h, d.data());
      L.d.assign((size_t)L.stride * (h + 2 * LM_WIN) * 2, 0);
      for (int y = 0; y < h; ++y)
        std::memcpy(&L.d[((size_t)(y + LM_WIN) * L.stride + LM_WIN) * 2], &d[(size_t)y * w * 2], (size_t)w * 4);
    }
    pyr.push_back(std::move(L));
    if (level == LM_MAXLEVEL) break;
    const int nw = (w + 1) / 2, nh = (h + 1) / 2;
    if (nw <= LM_WIN || nh <= LM_WIN) break;
    nxt.resize((size_t)nw * nh);
    lm_pyr_down(cur.data(), w, h, nxt.data());
    cur.swap(nxt);
    w = nw; h = nh; ++level;
  }
  return level;
}

// getImageNormParams klt.cpp:4-10 on two CV_16S windows
static void norm_params(const short* I, const short* J, int n, float& alpha, float& beta) {
  long long sI = 0, qI = 0, sJ = 0, qJ = 0;
  for (int i = 0; i < n; ++i) {
    sI += I[i]; qI += (long long)I[i] * I[i];
    sJ += J[i]; qJ += (long long)J[i] * J[i];
  }
  const double scale = 1.0 / n;
  const double mI = (double)sI * scale, mJ = (double)sJ * scale;
  const double sdI = std::sqrt(std::max((double)qI * scale - mI * mI, 0.0));
  const double sdJ = std::sqrt(std::max((double)qJ * scale - mJ * mJ, 0.0));
  alpha = float(sdI / sdJ);
  beta = float(mI - alpha * mJ);
}

static void sample_J(const Level& J, int ix, int iy, int iw00, int iw01, int iw10, int iw11, short* Jw) {
  for (int y = 0; y < LM_WIN; ++y) {
    const uint8_t* dst = J.at(ix, iy + y);
    for (int x = 0; x < LM_WIN; ++x)
      Jw[y * LM_WIN + x] = (short)descale(dst[x] * iw00 + dst[x + 1] * iw01 + dst[x + J.stride] * iw10 + dst[x + J.stride + 1] * iw11, 14 - 5);
  }
}

// LKTrackerInvoker2D::operator() for one level (lk_tracker_invoker_2d.cpp:28-479), affines == nullptr, flags == 0
static void lk_level(const Level& I, const Level& J, const float* prevPts, float* nextPts, uint8_t* status, float* err,
                     int npoints, int level, int maxLevel, bool illum, double epsilon) {
  const float halfWin = (LM_WIN - 1) * 0.5f;
  const float FLT_SCALE = 1.f / (1 << 20);
  const int W_BITS = 14;
  short Iw[LM_WIN * LM_WIN], Jw[LM_WIN * LM_WIN], dIw[LM_WIN * LM_WIN * 2];
  for (int p = 0; p < npoints; ++p) {
    const float lscale = (float)(1. / (1 << level));
    float prevx = prevPts[2 * p] * lscale, prevy = prevPts[2 * p + 1] * lscale;
    float nx, ny;
    if (level == maxLevel) { nx = prevx; ny = prevy; }
    else { nx = nextPts[2 * p] * 2.f; ny = nextPts[2 * p + 1] * 2.f; }
    nextPts[2 * p] = nx; nextPts[2 * p + 1] = ny;

    prevx -= halfWin; prevy -= halfWin;
    const int ipx = cv_floor(prevx), ipy = cv_floor(prevy);
    if (ipx < -LM_WIN || ipx >= I.w || ipy < -LM_WIN || ipy >= I.h) {
      if (level == 0) { status[p] = 0; err[p] = 0; }
      continue;
    }
    float a = prevx - ipx, b = prevy - ipy;
    int iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
    int iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
    int iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
    int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
    float iA11 = 0, iA12 = 0, iA22 = 0;
    for (int y = 0; y < LM_WIN; ++y) {
      const uint8_t* src = I.at(ipx, ipy + y);
      const short* ds = I.dat(ipx, ipy + y);
      const int ss = I.stride, dss = I.stride * 2;
      for (int x = 0; x < LM_WIN; ++x, ds += 2) {
        const int ival = descale(src[x] * iw00 + src[x + 1] * iw01 + src[x + ss] * iw10 + src[x + ss + 1] * iw11, W_BITS - 5);
        const int ixval = descale(ds[0] * iw00 + ds[2] * iw01 + ds[dss] * iw10 + ds[dss + 2] * iw11, W_BITS);
        const int iyval = descale(ds[1] * iw00 + ds[3] * iw01 + ds[dss + 1] * iw10 + ds[dss + 3] * iw11, W_BITS);
        Iw[y * LM_WIN + x] = (short)ival;
        dIw[(y * LM_WIN + x) * 2] = (short)ixval;
        dIw[(y * LM_WIN + x) * 2 + 1] = (short)iyval;
        iA11 += (float)(ixval * ixval);
        iA12 += (float)(ixval * iyval);
        iA22 += (float)(iyval * iyval);
      }
    }
    const float A11 = iA11 * FLT_SCALE, A12 = iA12 * FLT_SCALE, A22 = iA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * LM_WIN * LM_WIN);
    if (minEig < LM_MINEIG || D < FLT_EPSILON) {
      if (level == 0) status[p] = 0;
      continue;
    }
    D = 1.f / D;
    nx -= halfWin; ny -= halfWin;
    float pdx = 0, pdy = 0;
    int j;
    for (j = 0; j < LM_MAXCOUNT; ++j) {
      const int inx = cv_floor(nx), iny = cv_floor(ny);
      if (inx < -halfWin || inx >= J.w || iny < -halfWin || iny >= J.h) {
        if (level == 0) status[p] = 0;
        break;
      }
      a = nx - inx; b = ny - iny;
      iw00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
      iw01 = cv_round(a * (1.f - b) * (1 << W_BITS));
      iw10 = cv_round((1.f - a) * b * (1 << W_BITS));
      iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      sample_J(J, inx, iny, iw00, iw01, iw10, iw11, Jw);
      float alpha = 1.0f, beta = 0.0f;
      if (illum) norm_params(Iw, Jw, LM_WIN * LM_WIN, alpha, beta);
      float ib1 = 0, ib2 = 0;
      for (int k = 0; k < LM_WIN * LM_WIN; ++k) {
        const float diff = (float)(alpha * Jw[k] + beta - Iw[k]);
        ib1 += (float)(diff * dIw[2 * k]);
        ib2 += (float)(diff * dIw[2 * k + 1]);
      }
      const float b1 = ib1 * FLT_SCALE, b2 = ib2 * FLT_SCALE;
      const float dx = (float)((A12 * b2 - A22 * b1) * D), dy = (float)((A12 * b1 - A11 * b2) * D);
      nx += dx; ny += dy;
      nextPts[2 * p] = nx + halfWin; nextPts[2 * p + 1] = ny + halfWin;
      if ((double)dx * dx + (double)dy * dy <= epsilon) break;
      if (j > 0 && std::abs(dx + pdx) < 0.01 && std::abs(dy + pdy) < 0.01) {
        nextPts[2 * p] -= dx * 0.5f; nextPts[2 * p + 1] -= dy * 0.5f;
        break;
      }
      pdx = dx; pdy = dy;
    }
    if (j == LM_MAXCOUNT && level == 0) status[p] = 0;
    if (level == 0 && status[p]) {
      const float fx = nextPts[2 * p] - halfWin, fy = nextPts[2 * p + 1] - halfWin;
      const int ix = cv_floor(fx), iy = cv_floor(fy);
      if (ix < -LM_WIN || ix >= J.w || iy < -LM_WIN || iy >= J.h) { status[p] = 0; continue; }
      const float aa = fx - ix, bb = fy - iy;
      iw00 = cv_round((1.f - aa) * (1.f - bb) * (1 << W_BITS));
      iw01 = cv_round(aa * (1.f - bb) * (1 << W_BITS));
      iw10 = cv_round((1.f - aa) * bb * (1 << W_BITS));
      iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
      sample_J(J, ix, iy, iw00, iw01, iw10, iw11, Jw);
      float alpha = 1.0f, beta = 0.0f;
      if (illum) norm_params(Iw, Jw, LM_WIN * LM_WIN, alpha, beta);
      float errval = 0.f;
      for (int k = 0; k < LM_WIN * LM_WIN; ++k) errval += std::abs((float)(alpha * Jw[k] + beta - Iw[k]));
      err[p] = errval * 1.f / (32 * LM_WIN * LM_WIN);
    }
  }
}

// LineMatching::Anchors :532-602
static void lm_anchors(const LMLine* lines, int n, int step, std::vector<float>& kps, std::vector<int>& kp2line,
                       std::vector<int>& kpnum) {
  kps.clear(); kp2line.clear(); kpnum.assign(n, 0);
  for (int i = 0; i < n; ++i) {
    float px = lines[i].endpoint[0], py = lines[i].endpoint[1];
    const float x1 = px, y1 = py, x2 = lines[i].endpoint[2], y2 = lines[i].endpoint[3];
    const float len = lines[i].length;
    const float dirx = (x2 - x1) / len, diry = (y2 - y1) / len;
    const float ddx = step * dirx, ddy = step * diry;
    const int iter = int(len / step);
    for (int j = 0; j <= iter; ++j) {
      kps.push_back(px); kps.push_back(py); kp2line.push_back(i);
      px += ddx; py += ddy;
    }
    kps.push_back(x2); kps.push_back(y2); kp2line.push_back(i);
    kpnum[i] = iter + 2;
  }
}

// LineMatching::PointLineDistance :21-41
static float point_line_distance(float x, float y, const float* e) {
  const float vx = e[2] - e[0], vy = e[3] - e[1];
  const float ux = e[0] - x, uy = e[1] - y;
  float t = -(vx * ux + vy * uy) / (vx * vx + vy * vy);
  if (t < 0) t = 0; else if (t > 1) t = 1;
  const float dx = t * vx + ux, dy = t * vy + uy;
  return sqrt(dx * dx + dy * dy);   // double sqrt of a float, result narrowed to float (as the reference's unqualified sqrt)
}

// LineMatching::SidenessCheck :399-436
static bool sideness(const LMLine& l1r, const LMLine& l2r, const LMLine& l1c, const LMLine& l2c, float& d1, float& d2) {
  const double a1 = l1r.equation[0], b1 = l1r.equation[1], c1 = l1r.equation[2];
  const double px1 = l2r.center[0], py1 = l2r.center[1];
  double a2 = l1c.equation[0], b2 = l1c.equation[1], c2 = l1c.equation[2];
  const double px2 = l2c.center[0], py2 = l2c.center[1];
  if ((fabs(a1 - a2) + fabs(b1 - b2)) > (fabs(a1 + a2) + fabs(b1 + b2))) { a2 = -a2; b2 = -b2; c2 = -c2; }
  d1 = (px1 * a1 + py1 * b1 + c1) / sqrt(a1 * a1 + b1 * b1);
  d2 = (px2 * a2 + py2 * b2 + c2) / sqrt(a2 * a2 + b2 * b2);
  return !(d1 * d2 < 0);
}

// LineMatching::LineFilter :167-264 (called by the reference's demo test_line_matching.cpp:57,74 between detector and matcher):
// lines by decreasing length; a longer line removes every shorter one that is nearly parallel to it (|u x v| <= |u||v| sin 3 deg
// with the stored `length` members as |u|, |v|) and has an end point closer than distance_threshold to it.
// The reference orders with std::sort (unstable): lines of EQUAL length are taken in index order here.
int lm_line_filter(LMLine* lines, int num, float distance_threshold, float parallel_threshold) {
  std::vector<int> idx(num);
  for (int i = 0; i < num; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return lines[a].length > lines[b].length; });
  for (int i = 0; i < num; ++i) {
    const int i1 = idx[i];
    const float u_dist = lines[i1].length;
    if (u_dist == -1) continue;
    const float* e1 = lines[i1].endpoint;
    const float ux = e1[2] - e1[0], uy = e1[3] - e1[1];
    for (int j = i + 1; j < num; ++j) {
      const int i2 = idx[j];
      const float v_dist = lines[i2].length;
      if (v_dist == -1) continue;
      const float* e2 = lines[i2].endpoint;
      const float vx = e2[2] - e2[0], vy = e2[3] - e2[1];
      if (fabsf(ux * vy - vx * uy) > (u_dist * v_dist * parallel_threshold)) continue;
      const float d1 = point_line_distance(e2[0], e2[1], e1);
      const float d2 = point_line_distance(e2[2], e2[3], e1);
      if (d1 < distance_threshold || d2 < distance_threshold) lines[i2].length = -1;
    }
  }
  int m = 0;
  for (int i = 0; i < num; ++i)
    if (lines[i].length != -1) lines[m++] = lines[i];
  return m;
}

}  // namespace orc

using namespace orc;

extern "C" {

int orc_line_filter(LMLine* lines, int n, float distance_threshold, float parallel_threshold) {
  return lm_line_filter(lines, n, distance_threshold, parallel_threshold);
}

int orc_lm_pyr_down(const uint8_t* src, int w, int h, uint8_t* dst) { lm_pyr_down(src, w, h, dst); return 0; }
int orc_lm_scharr(const uint8_t* src, int w, int h, short* dst) { lm_scharr(src, w, h, dst); return 0; }

// LineMatching::Matching.  Returns 1 (true) / 0 (false: a line list or the key-point list is empty; outputs untouched).
// Optional outputs (NULL to skip), cap_kps entries each: kps_ref[2], kps_cur[2], status, err, kp2line_cur; *n_kps.
int orc_line_match(const uint8_t* img_ref, const uint8_t* img_cur, int W, int H, const LMLine* lines_ref, int n_ref,
                   const LMLine* lines_cur, int n_cur, const LMParam* prm, int* ref_to_cur, int cap_kps, int* n_kps,
                   float* kps_ref_o, float* kps_cur_o, uint8_t* status_o, float* err_o, int* kp2line_cur_o) {
  if (n_kps) *n_kps = 0;
  if (n_ref == 0 || n_cur == 0) return 0;
  std::vector<float> kps;
  std::vector<int> kp2line_ref, kpnum;
  lm_anchors(lines_ref, n_ref, prm->step, kps, kp2line_ref, kpnum);
  const int np = (int)kp2line_ref.size();
  if (np == 0) return 0;

  // KLT::calc2D klt.cpp:491-628
  std::vector<Level> pI, pJ;
  int maxLevel = build_pyramid(img_ref, W, H, true, pI);
  maxLevel = std::min(maxLevel, build_pyramid(img_cur, W, H, false, pJ));
  std::vector<float> nextPts(2 * np, 0.f), err(np, 0.f);
  std::vector<uint8_t> status(np, 1);
  double eps = std::min(std::max(0.001, 0.), 10.);
  eps *= eps;                                             // KLT ctor klt.cpp:28-33
  for (int level = maxLevel; level >= 0; --level)
    lk_level(pI[level], pJ[level], kps.data(), nextPts.data(), status.data(), err.data(), np, level, maxLevel,
             prm->illumination_adapt != 0, eps);

  // ClosestLine :48-86
  std::vector<int> kp2line_cur(np, -1);
  for (int i = 0; i < np; ++i) {
    if (!status[i] || err[i] > prm->klt_error_threshold) continue;
    int min_idx = -1;
    float min_d = 1000000;
    for (int j = 0; j < n_cur; ++j) {
      const float d = point_line_distance(nextPts[2 * i], nextPts[2 * i + 1], lines_cur[j].endpoint);
      if (d < min_d) { min_d = d; min_idx = j; }
    }
    if (min_d < prm->closest_line_threshold) kp2line_cur[i] = min_idx;
  }

  // Point2Line :88-133
  std::vector<int> count(n_cur);
  int idx = 0, max_idx = 0;
  for (int i = 0; i < n_ref; ++i) {
    ref_to_cur[i] = -1;
    std::fill(count.begin(), count.end(), 0);
    for (int j = 0; j < kpnum[i]; ++j, ++idx)
      if (kp2line_cur[idx] != -1) count[kp2line_cur[idx]]++;
    int max_value = -1;
    for (int j = 0; j < n_cur; ++j)
      if (count[j] > max_value) { max_value = count[j]; max_idx = j; }
    if (max_value <= 2 || float(max_value) / kpnum[i] < prm->line_matching_ratio ||
        lines_cur[max_idx].length > lines_ref[i].length * prm->line_distance_error_ratio ||
        lines_cur[max_idx].length < lines_ref[i].length / prm->line_distance_error_ratio)
      continue;
    ref_to_cur[i] = max_idx;
  }

  // TopologicalFilter :266-397
  if (prm->topological_filter) {
    std::vector<int> cnt(n_ref, 0);
    int match_num = 0;
    for (int r1 = 0; r1 < n_ref; ++r1) {
      const int c1 = ref_to_cur[r1];
      if (c1 == -1) continue;
      ++match_num;
      for (int r2 = 0; r2 < n_ref; ++r2) {
        if (r1 == r2) continue;
        const int c2 = ref_to_cur[r2];
        if (c2 == -1) continue;
        const float ldr = fabs(lines_ref[r2].length - lines_cur[c2].length) / lines_ref[r2].length;
        if (ldr > prm->topo_length_tolerate_ratio) continue;
        float d1, d2;
        const bool ok = sideness(lines_ref[r1], lines_ref[r2], lines_cur[c1], lines_cur[c2], d1, d2);
        if (!ok && fabs(d1) > prm->topo_distance_threshold && fabs(d2) > prm->topo_distance_threshold) {
          cnt[r1] += 1;
          cnt[r2] += 1;
        }
      }
    }
    float threshold = prm->topo_violation_ratio * (match_num - 1);
    if (threshold < 2) threshold = 2;
    for (int r = 0; r < n_ref; ++r)
      if (cnt[r] > threshold) ref_to_cur[r] = -1;
  }

  if (n_kps) *n_kps = np;
  const int m = std::min(np, cap_kps);
  if (kps_ref_o) std::memcpy(kps_ref_o, kps.data(), (size_t)m * 8);
  if (kps_cur_o) std::memcpy(kps_cur_o, nextPts.data(), (size_t)m * 8);
  if (status_o) std::memcpy(status_o, status.data(), (size_t)m);
  if (err_o) std::memcpy(err_o, err.data(), (size_t)m * 4);
  if (kp2line_cur_o) std::memcpy(kp2line_cur_o, kp2line_cur.data(), (size_t)m * 4);
  return 1;
}

}  // extern "C"
