// TEST INFRASTRUCTURE -- CPU oracle of the vanishing-point stage of the line tracker
// (feature_tracker/src/vanishing_point_detection.cpp; called from line_feature_tracker.cpp:237-266).
//
// PARITY UNPINNED, and not pinnable: the reference seeds rand() with time(NULL) in every call (:107) and reads
// `lx[idx]` with an index drawn for another list (:411-416, :428-431), i.e. past the end of a vector.  This restatement
//   * takes the seed as an argument and generates glibc's rand() stream for it (random_r.c TYPE_3: r[i] = r[i-3] + r[i-31],
//     output >> 1, 310 outputs discarded after srand) -- so a run of the reference with srand(seed) draws the same pairs;
//   * defines the out-of-range read as "no query line" (flag stays false);
//   * gives up (returns -1) after 100000 draws when every drawn pair of lines is parallel, where the reference loops
//     forever (:124-128).
//   * evaluates sin / cos / atan / atan2 / acos with the fixed IEEE-double formulas of detmath.h instead of the platform
//     libm: the longitude of vp2 is lambda = j degrees up to rounding (:137-147, :298-311), i.e. it sits ON the cell borders
//     of the sphere grid, so the reference's choice depends on the last bit of whatever libm it runs with.
// Everything else follows the cited lines, quirks included (dx = x1 - y1, dy = x2 - y2 in lineinfo :80-81; the local
// row_f = 1 that makes the first-frame VPs irrelevant :318-339).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "detmath.h"

namespace {

constexpr double kPi = 3.1415926535897932384626433832795;   // CV_PI

struct GlibcRand {   // srand(seed); rand()
  std::vector<uint32_t> s;
  size_t n = 0;
  int drawn = 0;
  explicit GlibcRand(uint32_t seed) {
    s.resize(344);
    int32_t word = seed ? (int32_t)seed : 1;
    s[0] = (uint32_t)word;
    for (int i = 1; i < 31; ++i) {
      const long hi = word / 127773, lo = word % 127773;
      long w = 16807 * lo - 2836 * hi;
      if (w < 0) w += 2147483647;
      word = (int32_t)w;
      s[i] = (uint32_t)word;
    }
    for (int i = 31; i < 34; ++i) s[i] = s[i - 31];
    for (int i = 34; i < 344; ++i) s[i] = s[i - 31] + s[i - 3];
    n = 344;
  }
  int next() {
    const uint32_t v = s[n - 31] + s[n - 3];
    s.push_back(v);
    ++n;
    ++drawn;
    return (int)(v >> 1);
  }
};

struct V3 { double x, y, z; };
inline V3 cross(const V3& a, const V3& b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// segAngle (:22-27)
double seg_angle(const float* e) {   // float arguments: the reference's call resolves to atan2f
  if (e[2] > e[0]) return (float)det::atan2((double)(e[3] - e[1]), (double)(e[2] - e[0]));
  return (float)det::atan2((double)(e[1] - e[3]), (double)(e[0] - e[2]));
}

struct LineInfo { std::vector<V3> para; std::vector<double> length, orientation; };

// lineinfo (:69-92)
LineInfo lineinfo(const float* ends, int num) {
  LineInfo li;
  for (int i = 0; i < num; ++i) {
    const float* e = ends + 4 * i;
    const V3 p1{e[0], e[1], 1.0}, p2{e[2], e[3], 1.0};
    li.para.push_back(cross(p1, p2));
    const double dx = e[0] - e[1];
    const double dy = e[2] - e[3];
    li.length.push_back(std::sqrt(dx * dx + dy * dy));
    double orientation = det::atan2(dy, dx);
    if (orientation < 0) orientation += kPi;
    li.orientation.push_back(orientation);
  }
  return li;
}

// getVPHypVia2Lines (:93-177).  hyp: [it*360][3][3].  Returns it, or -1.
int hypotheses(const LineInfo& li, int num, double f, double ppx, double ppy, GlibcRand& rng, std::vector<double>& hyp,
               std::vector<int>* pairs) {
  const double noiseRatio = 0.5;
  const double p = 1.0 / 3.0 * std::pow(1.0 - noiseRatio, 2);
  const double confEfficience = 0.9999;
  const int it = (int)(std::log(1 - confEfficience) / std::log(1.0 - p));
  const int numVp2 = 360;
  const double stepVp2 = 2.0 * kPi / numVp2;
  hyp.assign((size_t)it * numVp2 * 9, 0.0);
  int count = 0;
  for (int i = 0; i < it; ++i) {
    if (rng.drawn > 100000) return -1;
    int idx1 = rng.next() % num;
    int idx2 = rng.next() % num;
    while (idx2 == idx1) {
      if (rng.drawn > 100000) return -1;
      idx2 = rng.next() % num;
    }
    const V3 vp1_Img = cross(li.para[idx1], li.para[idx2]);
    if (vp1_Img.z == 0) { --i; continue; }
    if (pairs) { pairs->push_back(idx1); pairs->push_back(idx2); }
    V3 vp1{vp1_Img.x / vp1_Img.z - ppx, vp1_Img.y / vp1_Img.z - ppy, f};
    if (vp1.z == 0) vp1.z = 0.0011;
    double N = std::sqrt(vp1.x * vp1.x + vp1.y * vp1.y + vp1.z * vp1.z);
    vp1.x *= 1.0 / N; vp1.y *= 1.0 / N; vp1.z *= 1.0 / N;
    for (int j = 0; j < numVp2; ++j) {
      const double lambda = j * stepVp2;
      const double k1 = vp1.x * det::sin(lambda) + vp1.y * det::cos(lambda);
      const double k2 = vp1.z;
      const double phi = det::atan(-k2 / k1);
      const double Z = det::cos(phi);
      const double X = det::sin(phi) * det::sin(lambda);
      const double Y = det::sin(phi) * det::cos(lambda);
      V3 vp2{X, Y, Z};
      if (vp2.z == 0.0) vp2.z = 0.0011;
      N = std::sqrt(vp2.x * vp2.x + vp2.y * vp2.y + vp2.z * vp2.z);
      vp2.x *= 1.0 / N; vp2.y *= 1.0 / N; vp2.z *= 1.0 / N;
      if (vp2.z < 0) { vp2.x *= -1.0; vp2.y *= -1.0; vp2.z *= -1.0; }
      V3 vp3 = cross(vp1, vp2);
      if (vp3.z == 0.0) vp3.z = 0.0011;
      N = std::sqrt(vp3.x * vp3.x + vp3.y * vp3.y + vp3.z * vp3.z);
      vp3.x *= 1.0 / N; vp3.y *= 1.0 / N; vp3.z *= 1.0 / N;
      if (vp3.z < 0) { vp3.x *= -1.0; vp3.y *= -1.0; vp3.z *= -1.0; }
      double* h = &hyp[(size_t)count * 9];
      h[0] = vp1.x; h[1] = vp1.y; h[2] = vp1.z;
      h[3] = vp2.x; h[4] = vp2.y; h[5] = vp2.z;
      h[6] = vp3.x; h[7] = vp3.y; h[8] = vp3.z;
      ++count;
    }
  }
  return it;
}

// getSphereGrids (:180-276): grid [90][360]
void sphere_grid(const LineInfo& li, int num, double f, double ppx, double ppy, std::vector<double>& grid) {
  const double angelAccuracy = 1.0 / 180.0 * kPi;
  const int gridLA = (int)((kPi / 2.0) / angelAccuracy), gridLO = (int)((kPi * 2.0) / angelAccuracy);
  std::vector<double> g((size_t)gridLA * gridLO, 0.0);
  const double angelTolerance = 60.0 / 180.0 * kPi;
  for (int i = 0; i < num - 1; ++i)
    for (int j = i + 1; j < num; ++j) {
      const V3 pt = cross(li.para[i], li.para[j]);
      if (pt.z == 0) continue;
      const double x = pt.x / pt.z, y = pt.y / pt.z;
      const double X = x - ppx, Y = y - ppy, Z = f;
      const double N = std::sqrt(X * X + Y * Y + Z * Z);
      const double latitude = det::acos(Z / N);
      const double longitude = det::atan2(X, Y) + kPi;
      int LA = (int)(latitude / angelAccuracy);
      if (LA >= gridLA) LA = gridLA - 1;
      int LO = (int)(longitude / angelAccuracy);
      if (LO >= gridLO) LO = gridLO - 1;
      double angleDev = std::fabs(li.orientation[i] - li.orientation[j]);
      angleDev = std::min(kPi - angleDev, angleDev);
      if (angleDev > angelTolerance) continue;
      g[(size_t)LA * gridLO + LO] += std::sqrt(li.length[i] * li.length[j]) * (det::sin(2.0 * angleDev) + 0.2);
    }
  grid.assign((size_t)gridLA * gridLO, 0.0);
  for (int i = 1; i < gridLA - 1; ++i)
    for (int j = 1; j < gridLO - 1; ++j) {
      double neighborTotal = 0.0;
      for (int m = 0; m < 3; ++m)
        for (int n = 0; n < 3; ++n) neighborTotal += g[(size_t)(i - 1 + m) * gridLO + (j - 1 + n)];
      grid[(size_t)i * gridLO + j] = g[(size_t)i * gridLO + j] + neighborTotal / 9;
    }
}

// getBestVpsHyp (:278-345): scores of all hypotheses, first maximum, the frame_count > 0 swap
int best_hypothesis(const std::vector<double>& grid, const std::vector<double>& hyp, int first_frame, double* vps,
                    std::vector<double>& lineLength) {
  const int num = (int)(hyp.size() / 9);
  const double oneDegree = 1.0 / 180.0 * kPi;
  lineLength.assign(num, 0.0);
  for (int i = 0; i < num; ++i)
    for (int j = 0; j < 3; ++j) {
      const double* v = &hyp[(size_t)i * 9 + 3 * j];
      if (v[2] == 0.0) continue;
      const double latitude = det::acos(v[2]);
      const double longitude = det::atan2(v[0], v[1]) + kPi;
      int gridLA = (int)(latitude / oneDegree);
      if (gridLA == 90) gridLA = 89;
      int gridLO = (int)(longitude / oneDegree);
      if (gridLO == 360) gridLO = 359;
      lineLength[i] += grid[(size_t)gridLA * 360 + gridLO];
    }
  int bestIdx = 0;
  double maxLength = 0.0;
  for (int i = 0; i < num; ++i)
    if (lineLength[i] > maxLength) { maxLength = lineLength[i]; bestIdx = i; }
  std::memcpy(vps, &hyp[(size_t)bestIdx * 9], 72);
  if (!first_frame) {
    // row_f is the local 1 in every later frame (:318), row_v from vps[1].y (:331-334)
    const int row_v = std::fabs(vps[4]) > 0.8 ? 1 : 2;
    if (1 != row_v)
      for (int c = 0; c < 3; ++c) std::swap(vps[3 + c], vps[6 + c]);
  }
  return bestIdx;
}

// lines2Vps (:347-466)
void lines2vps(const float* ends, int num, double thAngle, const double* vps, double f, double ppx, double ppy, GlibcRand& rng,
               int* vp_idx) {
  std::vector<int> lx, ly, lz;
  double vp2D[3][2];
  for (int i = 0; i < 3; ++i) {
    vp2D[i][0] = vps[3 * i] * f / vps[3 * i + 2] + ppx;
    vp2D[i][1] = vps[3 * i + 1] * f / vps[3 * i + 2] + ppy;
  }
  for (int i = 0; i < num; ++i) {
    const float* e = ends + 4 * i;
    const double x1 = e[0], y1 = e[1], x2 = e[2], y2 = e[3];
    const double xm = (x1 + x2) / 2.0, ym = (y1 + y2) / 2.0;
    double v1x = x1 - x2, v1y = y1 - y2;
    const double N1 = std::sqrt(v1x * v1x + v1y * v1y);
    v1x /= N1; v1y /= N1;
    double minAngle = 1000;
    int bestIdx = 0;
    for (int j = 0; j < 3; ++j) {
      double v2x = vp2D[j][0] - xm, v2y = vp2D[j][1] - ym;
      const double N2 = std::sqrt(v2x * v2x + v2y * v2y);
      v2x /= N2; v2y /= N2;
      double crossValue = v1x * v2x + v1y * v2y;
      if (crossValue > 1.0) crossValue = 1.0;
      if (crossValue < -1.0) crossValue = -1.0;
      double angle = det::acos(crossValue);
      angle = std::min(kPi - angle, angle);
      bool flag = false;
      if (angle < minAngle) {
        const std::vector<int>& sized = j == 0 ? ly : j == 1 ? lz : lx;    // the list whose size is tested and drawn from
        if (sized.size() > 1) {
          const int idx = rng.next() % (int)sized.size();
          if (idx < (int)lx.size()) {                                     // the query is always taken from lx
            const float cur_angle = (float)seg_angle(e);
            const float query_angle = (float)seg_angle(ends + 4 * lx[idx]);
            const float delta_angle = std::fabs(cur_angle - query_angle);
            if (delta_angle < 0.175) flag = true;
          }
        }
        if (!flag) {
          minAngle = angle;
          bestIdx = j;
          (j == 0 ? lx : j == 1 ? ly : lz).push_back(i);
        }
      }
    }
    vp_idx[i] = minAngle < thAngle ? bestIdx : 3;
  }
}

}  // namespace

extern "C" {

// the elementary functions of detmath.h (test access): fn 0 sin, 1 cos, 2 atan, 3 acos, 4 atan2(x, y)
int orc_detmath(int fn, int n, const double* x, const double* y, double* out) {
  for (int i = 0; i < n; ++i)
    out[i] = fn == 0 ? det::sin(x[i]) : fn == 1 ? det::cos(x[i]) : fn == 2 ? det::atan(x[i]) : fn == 3 ? det::acos(x[i]) : det::atan2(x[i], y[i]);
  return 0;
}

// glibc rand() stream (test access)
int orc_glibc_rand(uint32_t seed, int n, int* out) {
  GlibcRand r(seed);
  for (int i = 0; i < n; ++i) out[i] = r.next();
  return 0;
}

// run_vanishing_point_detection (:37-66).  hyp_ends: the lines the hypotheses / the sphere grid are built from
// (`lines`), all_ends: the lines that are classified (`all_lines`).  Optional outputs (NULL to skip): hyp [it*360*9],
// grid [90*360], scores [it*360], best_idx, pairs [2*it], drawn [2] = rand() calls after the hypotheses / in total.  Returns it (105), or -1 (degenerate input).
int orc_vp_detect(const float* hyp_ends, int n_hyp, const float* all_ends, int n_all, double f, double ppx, double ppy, uint32_t seed,
                  int first_frame, double* vps, int* vp_ids, double* hyp_out, double* grid_out, double* scores_out, int* best_idx,
                  int* pairs_out, int* drawn_out) {
  if (n_hyp < 2) return -1;
  const LineInfo li = lineinfo(hyp_ends, n_hyp);
  GlibcRand rng(seed);
  std::vector<double> hyp, grid, scores;
  std::vector<int> pairs;
  const int it = hypotheses(li, n_hyp, f, ppx, ppy, rng, hyp, &pairs);
  if (it < 0) return -1;
  if (drawn_out) drawn_out[0] = rng.drawn;
  sphere_grid(li, n_hyp, f, ppx, ppy, grid);
  const int b = best_hypothesis(grid, hyp, first_frame, vps, scores);
  lines2vps(all_ends, n_all, 1.0 / 180.0 * kPi, vps, f, ppx, ppy, rng, vp_ids);
  if (hyp_out) std::memcpy(hyp_out, hyp.data(), hyp.size() * 8);
  if (grid_out) std::memcpy(grid_out, grid.data(), grid.size() * 8);
  if (scores_out) std::memcpy(scores_out, scores.data(), scores.size() * 8);
  if (best_idx) *best_idx = b;
  if (pairs_out) std::memcpy(pairs_out, pairs.data(), pairs.size() * 4);
  if (drawn_out) drawn_out[1] = rng.drawn;
  return it;
}

// lines2Vps alone, from given VPs and a generator that has already produced `skip` numbers (test access: lets the
// classification be replayed from the device's choice when two hypotheses tie)
int orc_vp_lines2vps(const float* all_ends, int n_all, const double* vps, double f, double ppx, double ppy, uint32_t seed, int skip,
                     int* vp_ids) {
  GlibcRand rng(seed);
  for (int i = 0; i < skip; ++i) rng.next();
  lines2vps(all_ends, n_all, 1.0 / 180.0 * kPi, vps, f, ppx, ppy, rng, vp_ids);
  return rng.drawn;
}

}  // extern "C"
