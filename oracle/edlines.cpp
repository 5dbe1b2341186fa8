// TEST INFRASTRUCTURE -- CPU oracle (see smallmat.h).
// Restates the EDLines line extractor of the reference's line front-end:
//   EDLineDetector::EdgeDrawing          line_matching/src/edline_detector.cpp:81-710
//   EDLineDetectorParallel::operator()   :960-1174   (LeastSquaresLineFit :729-891, LineValidation :893-958)
//   EDLineDetector::EDline               :1176-1198  ; nfa / log_gamma  edline_detector.h:186-348
// Production passes smoothed = true (no Gaussian blur, feature_tracker/src/line_feature_tracker.cpp:87 ->
// edline_detector.cpp:82-86); the reference's default and both of its demo programs pass smoothed = false
// (edline_detector.h:79-81, test_edline_detector.cpp:27, test_line_matching.cpp:54,74): cv::GaussianBlur(image, image_,
// Size(ksize, ksize), sigma) first -- ed_gaussian_blur below.
//
// OpenCV 3.4 (third party, not in the tree) supplies Sobel / absdiff / add / threshold / Mat division /
// compare / the float GEMM of the line fit.  Their arithmetic is restated from the published behaviour:
//   Sobel 3x3, CV_8U -> CV_16S, BORDER_REFLECT_101, exact integer;
//   threshold(THRESH_TOZERO) on CV_16S: dst = src > thresh ? src : 0;
//   Mat / 4 on CV_16S: saturate_cast<short>(src * 0.25) = round half to even;
//   Mat_<float> products (gemm, small sizes): products and sums in double, result cast to float.
//   GaussianBlur on CV_8UC1 (imgproc/smooth.cpp, the bit-exact fixed-point path OpenCV takes for 8-bit images since
//   3.4.1 when no IPP / OpenVX / HAL replacement is compiled in): the 1-D kernel in unsigned 8.8 fixed point, a row pass
//   into 8.8 values (ufixedpoint16), a column pass into 16.16 (ufixedpoint32), (v + 0x8000) >> 16 saturated to 8 bits,
//   BORDER_REFLECT_101.  Two published roundings of the kernel exist:
//     3.4.1 .. 3.4.8 / 4.0 .. 4.1:  k_i = cvRound(256 g_i)                         (5, sigma 1: 14 63 103 63 14, sum 257)
//     3.4.9+ / 4.2+ (getGaussianKernelFixedPoint_ED):  error diffusion from the ends to the centre, the centre tap takes
//                    what is left of 256                                          (5, sigma 1: 14 62 104 62 14)
//   The reference's README names OpenCV 3.4.2; its own rendered output (below) is reproduced only by the SECOND kernel,
//   which is therefore the default here (ED_BLUR_NORMALISED); the first is kept as ED_BLUR_OPENCV_341.
// PINNED (front-end, smoothed = false path): line_matching/data/edline_result.png is the reference's own output of
// test_edline_detector.cpp (mh04/imgs/1.png, {5, 1, 30, 5, 2, 25, 1.8}, smoothed = false).  The segments recovered from
// that picture (tests/golden/make_edline_result_segments.py -> tests/golden/edline_result_segments.npz) are checked
// against this restatement in tests/test_edline_reference_picture.py: same number of lines, every pixel of every
// restated line painted in the picture, every painted pixel of the picture explained by a restated line or its arrows,
// and -- the picture's colours being rand() triples whose srand(time(0)) seed tests/golden/find_srand_seed.c recovers --
// the ORDER of the list: the reference's is an interleaving of 3 in-order runs of the one produced here (its
// cv::parallel_for_ stripes push lines under a lock).  line_matching/data/line_matching_result.png (frames 5 | 10,
// LineFilter(3.0), colours after srand(0)) pins the same for two more frames and says which lines LineFilter keeps
// (tests/test_line_matching_reference_picture.py).  Nothing else of the front-end has a reference-held expected output.
#include <array>
#include <cmath>
#include <cfloat>
#include <cstdint>
#include <cstring>
#include <vector>

namespace orc {

struct EDParam {
  int gradientThreshold;    // 30 in production (line_feature_tracker_node.cpp:203)
  int anchorThreshold;      // 5
  int scanIntervals;        // 2
  int minLineLen;           // 35
  double lineFitErrThreshold;  // 1.8
};

struct EDLine {
  float endpoint[4];
  double equation[3];
  float center[2];
  float length;
};

static inline int refl101(int i, int n) {   // BORDER_REFLECT_101
  if (i < 0) return -i;
  if (i >= n) return 2 * n - 2 - i;
  return i;
}
static inline short div4_half_even(short v) { return (short)std::nearbyint((double)v * 0.25); }

// ---- cv::GaussianBlur on 8-bit images (edline_detector.cpp:82-84) ------------------------------------------------------
enum { ED_BLUR_NORMALISED = 0, ED_BLUR_OPENCV_341 = 1 };

static inline int border101(int p, int len) {   // cv::borderInterpolate(p, len, BORDER_REFLECT_101)
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do { p = p < 0 ? -p : 2 * len - p - 2; } while ((unsigned)p >= (unsigned)len);
  return p;
}

// createGaussianKernels + getFixedpointGaussianKernel<ufixedpoint16> (smooth.cpp): ksize <= 0 with sigma > 0 picks
// cvRound(sigma * 3 * 2 + 1) | 1 (8-bit depth); sigma <= 0 takes the fixed tables for n = 1, 3, 5, 7 and
// sigma = 0.15 n + 0.35 otherwise.  Returns the kernel size (odd), or -1 for an even / non-positive size.
// OpenCV evaluates exp() in its softfloat double; the libm value can differ from it in the last place, which moves a tap
// only if 256 g_i sits within 1e-13 of a rounding boundary.
int ed_gauss_kernel_q8(int ksize, double sigma, int mode, int* k /* >= ksize entries */, int cap) {
  if (sigma < 0) sigma = 0;
  if (ksize <= 0 && sigma > 0) ksize = (int)std::nearbyint(sigma * 3 * 2 + 1) | 1;
  if (ksize <= 0 || (ksize & 1) == 0 || ksize > cap) return -1;
  const int n = ksize;
  std::vector<double> v(n);
  if (sigma <= 0 && n <= 7) {
    static const double t1[] = {1.0}, t3[] = {0.25, 0.5, 0.25}, t5[] = {0.0625, 0.25, 0.375, 0.25, 0.0625},
                        t7[] = {0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125};
    const double* t = n == 1 ? t1 : n == 3 ? t3 : n == 5 ? t5 : t7;
    for (int i = 0; i < n; ++i) v[i] = t[i];
  } else {
    const double sx = sigma > 0 ? sigma : 0.15 * n + 0.35;
    const double scale2X = -0.5 * 0.25 / (sx * sx);
    double sum = 0;
    for (int i = 0, x = 1 - n; i < n; ++i, x += 2) { v[i] = std::exp((double)(x * x) * scale2X); sum += v[i]; }
    sum = 1.0 / sum;
    for (int i = 0; i < n; ++i) v[i] *= sum;
  }
  if (mode == ED_BLUR_OPENCV_341) {
    for (int i = 0; i < n; ++i) k[i] = (int)std::nearbyint(v[i] * 256.0);   // ufixedpoint16(softdouble): cvRound(v * 256)
  } else {   // getGaussianKernelFixedPoint_ED(result, kernel, 8)
    const int h = n / 2;
    double err = 0;
    int sum = 0;
    for (int i = 0; i < h; ++i) {
      const double adj = v[i] * 256.0 + err;
      const int v0 = (int)std::nearbyint(adj);
      err = adj - v0;
      k[i] = k[n - 1 - i] = v0;
      sum += v0;
    }
    k[h] = 256 - 2 * sum;
  }
  return n;
}

// fixedSmoothInvoker<uint8_t, ufixedpoint16>: rows then columns, every product and sum exact in 16 / 32 bits
// (ufixedpoint16 saturates at 0xFFFF, which a kernel summing to <= 257/256 reaches only as the exact value)
void ed_gaussian_blur(const uint8_t* img, int W, int H, const int* k, int n, uint8_t* out) {
  const int r = n / 2;
  std::vector<uint16_t> rows((size_t)W * H);
  for (int y = 0; y < H; ++y) {
    const uint8_t* s = img + (size_t)y * W;
    for (int x = 0; x < W; ++x) {
      uint32_t a = 0;
      for (int j = 0; j < n; ++j) a += (uint32_t)k[j] * s[border101(x + j - r, W)];
      rows[(size_t)y * W + x] = (uint16_t)(a > 0xFFFFu ? 0xFFFFu : a);
    }
  }
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      uint64_t a = 0;
      for (int j = 0; j < n; ++j) a += (uint64_t)k[j] * rows[(size_t)border101(y + j - r, H) * W + x];
      const uint64_t v = (a + 0x8000u) >> 16;
      out[(size_t)y * W + x] = (uint8_t)(v > 255 ? 255 : v);
    }
}

// edline_detector.cpp:125-136
void ed_gradient(const uint8_t* img, int W, int H, int gradTh, short* dx, short* dy, short* gImg, uint8_t* dirImg) {
  for (int y = 0; y < H; ++y) {
    const uint8_t* r0 = img + (size_t)refl101(y - 1, H) * W;
    const uint8_t* r1 = img + (size_t)y * W;
    const uint8_t* r2 = img + (size_t)refl101(y + 1, H) * W;
    for (int x = 0; x < W; ++x) {
      const int xm = refl101(x - 1, W), xp = refl101(x + 1, W);
      const int gx = (r0[xp] + 2 * r1[xp] + r2[xp]) - (r0[xm] + 2 * r1[xm] + r2[xm]);
      const int gy = (r2[xm] + 2 * r2[x] + r2[xp]) - (r0[xm] + 2 * r0[x] + r0[xp]);
      const size_t i = (size_t)y * W + x;
      dx[i] = (short)gx;
      dy[i] = (short)gy;
      const int ax = gx < 0 ? -gx : gx, ay = gy < 0 ? -gy : gy;
      const int sum = ax + ay;
      const short t = sum > gradTh + 1 ? (short)sum : (short)0;   // THRESH_TOZERO at gradienThreshold_ + 1
      gImg[i] = div4_half_even(t);
      dirImg[i] = ax < ay ? 255 : 0;   // Horizontal = 255 if |dx| < |dy|
    }
  }
}

// edline_detector.cpp:148-164 ; scan order: w outer, h inner
int ed_anchors(const short* g, const uint8_t* dir, int W, int H, int scan, int anchorTh, unsigned* ax, unsigned* ay, int cap) {
  int n = 0;
  for (int w = 1; w < W - 1; w += scan)
    for (int h = 1; h < H - 1; h += scan) {
      const int i = h * W + w;
      bool a;
      if (dir[i] == 255) a = g[i] >= g[i - W] + anchorTh && g[i] >= g[i + W] + anchorTh;
      else a = g[i] >= g[i - 1] + anchorTh && g[i] >= g[i + 1] + anchorTh;
      if (a) {
        if (n < cap) { ax[n] = w; ay[n] = h; }
        ++n;
      }
    }
  return n;
}

namespace {
enum { UpDir = 1, RightDir = 2, DownDir = 3, LeftDir = 4 };
struct Walker {
  const short* g;
  const uint8_t* dir;
  uint8_t* edge;
  int W, H;
  unsigned lastX = 0, lastY = 0;   // persist across walks exactly like the reference's locals
  // one walk (the body that appears four times at :209-647), appending to (px, py)
  void walk(unsigned x, unsigned y, unsigned char lastDirection, std::vector<unsigned>& px, std::vector<unsigned>& py) {
    int idx = y * W + x;
    while (g[idx] > 0 && !edge[idx]) {
      edge[idx] = 1;
      px.push_back(x);
      py.push_back(y);
      unsigned char shouldGo = 0;
      if (dir[idx] == 255) {
        if (lastDirection == UpDir || lastDirection == DownDir) shouldGo = x > lastX ? RightDir : LeftDir;
        lastX = x; lastY = y;
        if (lastDirection == RightDir || shouldGo == RightDir) {
          if (x == (unsigned)W - 1 || y == 0 || y == (unsigned)H - 1) break;
          const unsigned char g1 = (unsigned char)g[idx - W + 1], g2 = (unsigned char)g[idx + 1], g3 = (unsigned char)g[idx + W + 1];
          if (g1 >= g2 && g1 >= g3) { x = x + 1; y = y - 1; }
          else if (g3 >= g2 && g3 >= g1) { x = x + 1; y = y + 1; }
          else x = x + 1;
          lastDirection = RightDir;
        } else if (lastDirection == LeftDir || shouldGo == LeftDir) {
          if (x == 0 || y == 0 || y == (unsigned)H - 1) break;
          const unsigned char g1 = (unsigned char)g[idx - W - 1], g2 = (unsigned char)g[idx - 1], g3 = (unsigned char)g[idx + W - 1];
          if (g1 >= g2 && g1 >= g3) { x = x - 1; y = y - 1; }
          else if (g3 >= g2 && g3 >= g1) { x = x - 1; y = y + 1; }
          else x = x - 1;
          lastDirection = LeftDir;
        }
      } else {
        if (lastDirection == RightDir || lastDirection == LeftDir) shouldGo = y > lastY ? DownDir : UpDir;
        lastX = x; lastY = y;
        if (lastDirection == DownDir || shouldGo == DownDir) {
          if (x == 0 || x == (unsigned)W - 1 || y == (unsigned)H - 1) break;
          const unsigned char g1 = (unsigned char)g[idx + W + 1], g2 = (unsigned char)g[idx + W], g3 = (unsigned char)g[idx + W - 1];
          if (g1 >= g2 && g1 >= g3) { x = x + 1; y = y + 1; }
          else if (g3 >= g2 && g3 >= g1) { x = x - 1; y = y + 1; }
          else y = y + 1;
          lastDirection = DownDir;
        } else if (lastDirection == UpDir || shouldGo == UpDir) {
          if (x == 0 || x == (unsigned)W - 1 || y == 0) break;
          const unsigned char g1 = (unsigned char)g[idx - W + 1], g2 = (unsigned char)g[idx - W], g3 = (unsigned char)g[idx - W - 1];
          if (g1 >= g2 && g1 >= g3) { x = x + 1; y = y - 1; }
          else if (g3 >= g2 && g3 >= g1) { x = x - 1; y = y - 1; }
          else y = y - 1;
          lastDirection = UpDir;
        }
      }
      idx = y * W + x;
    }
  }
};
}  // namespace

// edline_detector.cpp:166-707 : returns chains as (xCors, yCors, sId)
void ed_route(const short* g, const uint8_t* dir, int W, int H, const unsigned* ax, const unsigned* ay, int nAnchors,
              int minLineLen, uint8_t* edge, std::vector<unsigned>& xC, std::vector<unsigned>& yC, std::vector<unsigned>& sId) {
  std::memset(edge, 0, (size_t)W * H);
  Walker wk{g, dir, edge, W, H};
  xC.clear(); yC.clear(); sId.clear();
  std::vector<unsigned> fx, fy, sx, sy;
  for (int i = 0; i < nAnchors; ++i) {
    const unsigned x = ax[i], y = ay[i];
    const int idx = y * W + x;
    if (edge[idx]) continue;
    fx.clear(); fy.clear(); sx.clear(); sy.clear();
    const bool horiz = dir[idx] == 255;
    wk.walk(x, y, horiz ? RightDir : DownDir, fx, fy);
    edge[idx] = 0;   // the anchor is un-marked so that the second walk starts on it (:326, :535)
    wk.walk(x, y, horiz ? LeftDir : UpDir, sx, sy);
    if ((int)(fx.size() + sx.size()) < minLineLen + 1) continue;   // short edge dropped (its pixels stay marked)
    // The reference's buffers hold W H / 5 pixels per part and W H / 100 edges (:92-93, :113-118); beyond that it writes past
    // them and reports "Edge drawing Error" afterwards (:655-665).  Defined here (and on the device, k_ed_route): an edge that
    // does not fit is dropped like a short one.  Frames of noise with a small minLineLen get there (tools/fuzz_frontend.py).
    {
      const size_t cap = (size_t)W * H / 5, capEdges = cap / 20;
      if (sId.size() >= capEdges || xC.size() + fx.size() + sx.size() - 1 > 2 * cap || fx.size() > cap) continue;
    }
    sId.push_back((unsigned)xC.size());
    for (int k = (int)fx.size() - 1; k >= 0; --k) { xC.push_back(fx[k]); yC.push_back(fy[k]); }   // first part reversed
    for (size_t k = 1; k < sx.size(); ++k) { xC.push_back(sx[k]); yC.push_back(sy[k]); }          // second part without the anchor
  }
  sId.push_back((unsigned)xC.size());
}

// ---- NFA (edline_detector.h:186-348) -------------------------------------------------------------------
static double log_gamma_lanczos(double x) {
  static const double q[7] = {75122.6331530, 80916.6278952, 36308.2951477, 8687.24529705, 1168.92649479, 83.8676043424, 2.50662827511};
  double a = (x + 0.5) * std::log(x + 5.5) - (x + 5.5);
  double b = 0.0;
  for (int n = 0; n < 7; n++) {
    a -= std::log(x + (double)n);
    b += q[n] * std::pow(x, (double)n);
  }
  return a + std::log(b);
}
static double log_gamma_windschitl(double x) {
  return 0.918938533204673 + (x - 0.5) * std::log(x) - x + 0.5 * x * std::log(x * std::sinh(1 / x) + 1 / (810.0 * std::pow(x, 6.0)));
}
static double log_gamma(double x) { return x > 15.0 ? log_gamma_windschitl(x) : log_gamma_lanczos(x); }
static bool double_equal(double a, double b) {
  if (a == b) return true;
  double abs_diff = std::fabs(a - b), aa = std::fabs(a), bb = std::fabs(b);
  double abs_max = aa > bb ? aa : bb;
  if (abs_max < DBL_MIN) abs_max = DBL_MIN;
  return (abs_diff / abs_max) <= (100.0 * DBL_EPSILON);
}
double ed_nfa(int n, int k, double p, double logNT) {
  const double tolerance = 0.1;
  if (n == 0 || k == 0) return -logNT;
  if (n == k) return -logNT - (double)n * std::log10(p);
  const double p_term = p / (1.0 - p);
  const double log1term = log_gamma((double)n + 1.0) - log_gamma((double)k + 1.0) - log_gamma((double)(n - k) + 1.0) +
                          (double)k * std::log(p) + (double)(n - k) * std::log(1.0 - p);
  double term = std::exp(log1term);
  if (double_equal(term, 0.0)) {
    if ((double)k > (double)n * p) return -log1term / 2.30258509299404568402 - logNT;
    return -logNT;
  }
  double bin_tail = term;
  for (int i = k + 1; i <= n; i++) {
    const double bin_term = (double)(n - i + 1) / (double)i;
    const double mult_term = bin_term * p_term;
    term *= mult_term;
    bin_tail += term;
    if (bin_term < 1.0) {
      const double err = term * ((1.0 - std::pow(mult_term, (double)(n - i + 1))) / (1.0 - mult_term) - 1.0);
      if (err < tolerance * std::fabs(-std::log10(bin_tail) - logNT) * bin_tail) break;
    }
  }
  return -std::log10(bin_tail) - logNT;
}

namespace {
struct Fitter {
  const unsigned* xC;
  const unsigned* yC;
  const uint8_t* dir;
  const short* dx;
  const short* dy;
  int W, H, minLineLen;
  double thr, logNT;
  float ATA[4], ATV[2];   // cv::Mat_<float>

  bool horiz(unsigned off) const { return dir[yC[off] * W + xC[off]] == 255; }
  void solve(double eq[2]) const {
    const double coef = 1.0 / (double(ATA[0]) * double(ATA[3]) - double(ATA[1]) * double(ATA[2]));
    eq[0] = coef * (double(ATA[3]) * double(ATV[0]) - double(ATA[1]) * double(ATV[1]));
    eq[1] = coef * (double(ATA[0]) * double(ATV[1]) - double(ATA[2]) * double(ATV[0]));
  }
  // :729-807 initial fit over minLineLen pixels; float inputs, double accumulation, float result (cv::gemm)
  double fit_initial(unsigned offS, double eq[2]) {
    const bool hz = horiz(offS);
    double s_uu = 0, s_u = 0, s_uv = 0, s_v = 0;
    for (int i = 0; i < minLineLen; ++i) {
      const double u = (double)(float)(hz ? xC[offS + i] : yC[offS + i]);
      const double v = (double)(float)(hz ? yC[offS + i] : xC[offS + i]);
      s_uu += u * u; s_u += u; s_uv += u * v; s_v += v;
    }
    ATA[0] = (float)s_uu; ATA[1] = (float)s_u; ATA[2] = (float)s_u; ATA[3] = (float)(double)minLineLen;
    ATV[0] = (float)s_uv; ATV[1] = (float)s_v;
    solve(eq);
    double err = 0;
    for (int i = 0; i < minLineLen; ++i) {
      const double u = (double)(hz ? xC[offS + i] : yC[offS + i]), v = (double)(hz ? yC[offS + i] : xC[offS + i]);
      const double c = v - u * eq[0] - eq[1];
      err += c * c;
    }
    return std::sqrt(err);
  }
  // :809-891 incremental update with the points [newS, offE)
  void fit_update(unsigned offS_init, unsigned newS, unsigned offE, double eq[2]) {
    const int newLength = (int)offE - (int)newS;
    if ((int)offE - (int)offS_init <= 0 || newLength <= 0) return;   // the reference prints and returns -1, equation unchanged
    const bool hz = horiz(offS_init);
    double s_uu = 0, s_u = 0, s_uv = 0, s_v = 0;
    for (unsigned o = newS; o < offE; ++o) {
      const double u = (double)(float)(hz ? xC[o] : yC[o]);
      const double v = (double)(float)(hz ? yC[o] : xC[o]);
      s_uu += u * u; s_u += u; s_uv += u * v; s_v += v;
    }
    const float t00 = (float)s_uu, t01 = (float)s_u, t11 = (float)(double)newLength, v0 = (float)s_uv, v1 = (float)s_v;
    ATA[0] = ATA[0] + t00; ATA[1] = ATA[1] + t01; ATA[2] = ATA[2] + t01; ATA[3] = ATA[3] + t11;
    ATV[0] = ATV[0] + v0; ATV[1] = ATV[1] + v1;
    solve(eq);
  }
  // :893-958
  bool validate(unsigned offS, unsigned offE, const double leq[3]) const {
    const int n = (int)offE - (int)offS;
    int mgx = 0, mgy = 0;
    std::vector<double> pd;
    for (int i = 0; i < n; ++i) {
      const int idx = yC[offS + i] * W + xC[offS + i];
      mgx += dx[idx];
      mgy += dy[idx];
      pd.push_back(std::atan2(-(double)dx[idx], (double)dy[idx]));
    }
    const double ddx = std::fabs(leq[1]), ddy = std::fabs(leq[0]);
    if (mgx == 0 && mgy == 0) return false;
    float direction = 0.f;   // the reference leaves it uninitialised when no branch fires; all four cover every non-zero case
    if (mgx > 0 && mgy >= 0) direction = (float)std::atan2(-ddy, ddx);
    if (mgx <= 0 && mgy > 0) direction = (float)std::atan2(ddy, ddx);
    if (mgx < 0 && mgy <= 0) direction = (float)std::atan2(ddy, -ddx);
    if (mgx >= 0 && mgy < 0) direction = (float)std::atan2(-ddy, -ddx);
    if (std::fabs(direction) < 0.15 || M_PI - std::fabs(direction) < 0.15)
      if (std::fabs(leq[2]) < 10 || std::fabs(H - std::fabs(leq[2])) < 10) return false;
    if (std::fabs(std::fabs(direction) - M_PI * 0.5) < 0.15)
      if (std::fabs(leq[2]) < 10 || std::fabs(W - std::fabs(leq[2])) < 10) return false;
    int k = 0;
    for (int i = 0; i < n; ++i) {
      const double dd = std::fabs(direction - pd[i]);
      if (std::fabs(2 * M_PI - dd) < 0.392699 || dd < 0.392699) k++;
    }
    return ed_nfa(n, k, 0.125, logNT) > 0;
  }
};
}  // namespace

// EDLineDetectorParallel::operator() over all chains, sequentially: the serial order of the reference's loop.  (The
// reference's own list is an interleaving of in-order runs of this order -- parallel_for_ stripes under a lock,
// edline_detector.cpp:1081-1083, 1195 -- which both reference pictures confirm, see the header.)
void ed_fit(const unsigned* xC, const unsigned* yC, const unsigned* sId, int nEdges, const uint8_t* dir, const short* dx,
            const short* dy, int W, int H, const EDParam& P, std::vector<EDLine>& lines) {
  Fitter F{xC, yC, dir, dx, dy, W, H, P.minLineLen, P.lineFitErrThreshold,
           2.0 * (std::log10((double)W) + std::log10((double)H))};
  const unsigned minLineLen = P.minLineLen;
  for (int e = 0; e < nEdges; ++e) {
    unsigned offS = sId[e], offE = sId[e + 1];
    while (offE > offS + minLineLen) {
      double eq[2] = {0, 0}, lineFitErr = 0;
      while (offE > offS + minLineLen) {
        lineFitErr = F.fit_initial(offS, eq);
        if (lineFitErr <= P.lineFitErrThreshold) break;
        offS += 2;   // SkipEdgePoint
      }
      // The reference tests `lineFitErr > threshold` (:996).  A run of minLineLen pixels with one and the same abscissa makes
      // the normal equations singular and the error NaN: neither that test nor `<=` above is true, and the reference goes on
      // with offS beyond the end of the chain -- it reads past the edge's pixels (found with the sanitizers over the shapes of
      // tools/fuzz_frontend.py).  Defined here and on the device: a fit that did not succeed ends the chain.
      if (!(lineFitErr <= P.lineFitErrThreshold)) break;
      const bool hz = F.horiz(offS);
      const unsigned offS_init = offS;
      bool bExtended = true, bFirstTry = true;
      int numOfOutlier = 0, tryTimes = 0;
      unsigned newOffsetS = 0;
      double coef1 = 0;
      while (bExtended) {
        tryTimes++;
        if (bFirstTry) { bFirstTry = false; offS += minLineLen; }
        else F.fit_update(offS_init, newOffsetS, offS, eq);
        coef1 = 1 / std::sqrt(eq[0] * eq[0] + 1);
        numOfOutlier = 0;
        newOffsetS = offS;
        while (offE > offS) {
          const double d = hz ? std::fabs(eq[0] * xC[offS] - yC[offS] + eq[1]) * coef1
                              : std::fabs(xC[offS] - eq[0] * yC[offS] - eq[1]) * coef1;
          offS++;
          if (d > P.lineFitErrThreshold) {
            numOfOutlier++;
            if (numOfOutlier > 3) break;
          } else {
            numOfOutlier = 0;
          }
        }
        offS -= numOfOutlier;
        if (offS - newOffsetS > 0 && tryTimes < 6) {} else bExtended = false;
      }
      double leq[3];
      if (hz) { leq[0] = eq[0] * coef1; leq[1] = -1 * coef1; leq[2] = eq[1] * coef1; }
      else { leq[0] = 1 * coef1; leq[1] = -eq[0] * coef1; leq[2] = -eq[1] * coef1; }
      if (F.validate(offS_init, offS, leq)) {
        EDLine L;
        for (int k = 0; k < 3; ++k) L.equation[k] = leq[k];
        const double a1 = leq[1] * leq[1], a2 = leq[0] * leq[0], a3 = leq[0] * leq[1], a4 = leq[2] * leq[0], a5 = leq[2] * leq[1];
        unsigned Px = xC[offS_init], Py = yC[offS_init];
        const float x1 = (float)(a1 * Px - a3 * Py - a4), y1 = (float)(a2 * Py - a3 * Px - a5);
        Px = xC[offS - 1]; Py = yC[offS - 1];
        const float x2 = (float)(a1 * Px - a3 * Py - a4), y2 = (float)(a2 * Py - a3 * Px - a5);
        L.endpoint[0] = x1; L.endpoint[1] = y1; L.endpoint[2] = x2; L.endpoint[3] = y2;
        L.center[0] = (float)((x1 + x2) / 2.0);
        L.center[1] = (float)((y1 + y2) / 2.0);
        L.length = (float)std::sqrt(std::pow((double)(x2 - x1), 2) + std::pow((double)(y2 - y1), 2));
        lines.push_back(L);
      }
    }
  }
}

}  // namespace orc

using namespace orc;
extern "C" {
// Whole EDline() on one 8-bit image.  Optional outputs (may be NULL): dx, dy, gImg [W*H shorts], dirImg [W*H],
// anchors (x,y pairs, cap entries) + *nAnchors, chains (xC,yC cap entries, sId cap/20 entries) + *nEdges.
// lines: cap_lines * 10 doubles = x1,y1,x2,y2, eq0,eq1,eq2, cx,cy, length ; returns the number of lines.
int orc_edlines(const uint8_t* img, int W, int H, int gradTh, int anchorTh, int scan, int minLineLen, double fitErr,
                short* dx_o, short* dy_o, short* g_o, uint8_t* dir_o, unsigned* anchors_o, int* nAnchors_o,
                unsigned* chainX_o, unsigned* chainY_o, unsigned* sId_o, int* nEdges_o, int cap_px, double* lines_o, int cap_lines) {
  const size_t N = (size_t)W * H;
  std::vector<short> dx(N), dy(N), g(N);
  std::vector<uint8_t> dir(N), edge(N);
  ed_gradient(img, W, H, gradTh, dx.data(), dy.data(), g.data(), dir.data());
  const int cap = (int)(N / 5);
  std::vector<unsigned> ax(cap), ay(cap);
  int nA = ed_anchors(g.data(), dir.data(), W, H, scan, anchorTh, ax.data(), ay.data(), cap);
  if (nA > cap) nA = cap;
  std::vector<unsigned> xC, yC, sId;
  ed_route(g.data(), dir.data(), W, H, ax.data(), ay.data(), nA, minLineLen, edge.data(), xC, yC, sId);
  const int nE = (int)sId.size() - 1;
  // (capacity = size: a read one past the last chain pixel then lies outside the allocation, where a sanitizer build sees it)
  xC.shrink_to_fit(); yC.shrink_to_fit(); sId.shrink_to_fit();
  EDParam P{gradTh, anchorTh, scan, minLineLen, fitErr};
  std::vector<EDLine> lines;
  ed_fit(xC.data(), yC.data(), sId.data(), nE, dir.data(), dx.data(), dy.data(), W, H, P, lines);
  if (dx_o) std::memcpy(dx_o, dx.data(), N * 2);
  if (dy_o) std::memcpy(dy_o, dy.data(), N * 2);
  if (g_o) std::memcpy(g_o, g.data(), N * 2);
  if (dir_o) std::memcpy(dir_o, dir.data(), N);
  if (anchors_o) for (int i = 0; i < nA && i < cap_px; ++i) { anchors_o[2 * i] = ax[i]; anchors_o[2 * i + 1] = ay[i]; }
  if (nAnchors_o) *nAnchors_o = nA;
  if (chainX_o) for (size_t i = 0; i < xC.size() && (int)i < cap_px; ++i) { chainX_o[i] = xC[i]; chainY_o[i] = yC[i]; }
  if (sId_o) for (size_t i = 0; i < sId.size(); ++i) sId_o[i] = sId[i];
  if (nEdges_o) *nEdges_o = nE;
  int n = 0;
  for (auto& L : lines) {
    if (n >= cap_lines) break;
    double* o = lines_o + 10 * n++;
    for (int k = 0; k < 4; ++k) o[k] = L.endpoint[k];
    for (int k = 0; k < 3; ++k) o[4 + k] = L.equation[k];
    o[7] = L.center[0]; o[8] = L.center[1]; o[9] = L.length;
  }
  return (int)lines.size();
}
double orc_nfa(int n, int k, double p, double logNT) { return ed_nfa(n, k, p, logNT); }

// cv::GaussianBlur(image, image_, Size(ksize, ksize), sigma) of EdgeDrawing (edline_detector.cpp:82-84) on its own.
// kernel_out (may be NULL): the 8.8 fixed-point taps.  Returns the kernel size or -1.
int orc_gaussian_blur_u8(const uint8_t* img, int W, int H, int ksize, double sigma, int mode, uint8_t* out, int* kernel_out) {
  int k[64];
  const int n = ed_gauss_kernel_q8(ksize, sigma, mode, k, 63);
  if (n < 0) return -1;
  if (kernel_out) for (int i = 0; i < n; ++i) kernel_out[i] = k[i];
  if (n == 1) { std::memcpy(out, img, (size_t)W * H); return n; }   // GaussianBlur: a 1 x 1 kernel copies
  ed_gaussian_blur(img, W, H, k, n, out);
  return n;
}

// EDline(image, lines, smoothed): smoothed == 0 blurs first (the reference's default), then the same path
int orc_edlines_ex(const uint8_t* img, int W, int H, int smoothed, int ksize, double sigma, int blur_mode, int gradTh, int anchorTh,
                   int scan, int minLineLen, double fitErr, short* dx_o, short* dy_o, short* g_o, uint8_t* dir_o,
                   unsigned* anchors_o, int* nAnchors_o, unsigned* chainX_o, unsigned* chainY_o, unsigned* sId_o, int* nEdges_o,
                   int cap_px, double* lines_o, int cap_lines) {
  if (smoothed)
    return orc_edlines(img, W, H, gradTh, anchorTh, scan, minLineLen, fitErr, dx_o, dy_o, g_o, dir_o, anchors_o, nAnchors_o,
                       chainX_o, chainY_o, sId_o, nEdges_o, cap_px, lines_o, cap_lines);
  std::vector<uint8_t> b((size_t)W * H);
  if (orc_gaussian_blur_u8(img, W, H, ksize, sigma, blur_mode, b.data(), nullptr) < 0) return -1;
  return orc_edlines(b.data(), W, H, gradTh, anchorTh, scan, minLineLen, fitErr, dx_o, dy_o, g_o, dir_o, anchors_o, nAnchors_o,
                     chainX_o, chainY_o, sId_o, nEdges_o, cap_px, lines_o, cap_lines);
}
}
