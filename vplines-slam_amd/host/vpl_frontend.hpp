// Host-side mirror of the reference's line front-end interface, forwarding to the C ABI of
// include/vplines_frontend.h (HIP kernels).  Same class names, constructor arguments, member-function names and
// argument meaning as
//   line_matching/src/line.h:8-17                    struct Line
//   line_matching/src/edline_detector.h:32-40,55-84  EDLineParam, EDLineDetector::EDline
//   line_matching/src/line_matching.h:14-33          LineMatching::LineMatching, LineMatching::Matching
//   feature_tracker/include/linefeature_tracker.h    FrameLines, LineFeatureTracker::readImage / undistortedLineEndPoints
// so the tracker code (feature_tracker/src/line_feature_tracker.cpp:87,291-314) keeps its shape.
// Frames are raw 8-bit single-channel buffers; with -DVPL_USE_OPENCV overloads taking cv::Mat are added
// (cv::Mat::data of a continuous CV_8UC1 matrix is that buffer).  Nothing computes on the CPU: without a HIP device
// the constructors throw.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <ctime>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "vplines_frontend.h"

#ifdef VPL_USE_OPENCV
#include <opencv2/core.hpp>
#endif

namespace vplhost {

struct Line {   // numeric members of the reference's struct Line (the debug key-point vectors are not carried)
  std::array<float, 4> line_endpoint;
  std::array<double, 3> line_equation;
  std::array<float, 2> center;
  float length;
};

struct EDLineParam {
  int ksize;
  float sigma;
  float gradientThreshold;
  float anchorThreshold;
  int scanIntervals;
  int minLineLen;
  double lineFitErrThreshold;
};

// One device context shared by the detector and the matcher of a tracker (two frames in flight).
class FrontendDevice {
 public:
  FrontendDevice(int width, int height, int max_lines = 1024, int max_kps = 8192, int device = 0)
      : w_(width), h_(height), max_lines_(max_lines) {
    if (vpl_fe_create(&fe_, device, 2, width, height, max_lines) != 0)
      throw std::runtime_error("vpl_fe_create failed (no HIP device? there is no CPU fallback)");
    if (vpl_match_reserve(fe_, 1, max_kps) != 0) {
      std::string m = vpl_fe_last_error(fe_);
      vpl_fe_destroy(fe_);
      throw std::runtime_error("vpl_match_reserve: " + m);
    }
  }
  ~FrontendDevice() { vpl_fe_destroy(fe_); }
  FrontendDevice(const FrontendDevice&) = delete;
  FrontendDevice& operator=(const FrontendDevice&) = delete;
  vpl_fe_ctx* ctx() const { return fe_; }
  int width() const { return w_; }
  int height() const { return h_; }
  int max_lines() const { return max_lines_; }

 private:
  vpl_fe_ctx* fe_ = nullptr;
  int w_, h_, max_lines_;
};

inline Line to_line(const vpl_line& v) {
  Line l;
  for (int k = 0; k < 4; ++k) l.line_endpoint[k] = v.line_endpoint[k];
  for (int k = 0; k < 3; ++k) l.line_equation[k] = v.line_equation[k];
  l.center = {v.center[0], v.center[1]};
  l.length = v.length;
  return l;
}
inline vpl_line from_line(const Line& l) {
  vpl_line v;
  std::memset(&v, 0, sizeof(v));
  for (int k = 0; k < 4; ++k) v.line_endpoint[k] = l.line_endpoint[k];
  for (int k = 0; k < 3; ++k) v.line_equation[k] = l.line_equation[k];
  v.center[0] = l.center[0]; v.center[1] = l.center[1];
  v.length = l.length;
  return v;
}

class EDLineDetector {
 public:
  EDLineDetector(FrontendDevice& dev, EDLineParam param) : dev_(dev) {
    p_.ksize = param.ksize; p_.sigma = param.sigma; p_.gradientThreshold = param.gradientThreshold;
    p_.anchorThreshold = param.anchorThreshold; p_.scanIntervals = param.scanIntervals;
    p_.minLineLen = param.minLineLen; p_.lineFitErrThreshold = param.lineFitErrThreshold;
  }
  // int EDLineDetector::EDline(cv::Mat& image, std::vector<Line>& lines, bool smoothed = false): returns 1, or -1 on error
  // (edline_detector.h:79-81, edline_detector.cpp:1176-1198).  smoothed = false, the reference's default, runs the Gaussian
  // pre-blur of EdgeDrawing first (edline_detector.cpp:82-84); production passes true (line_feature_tracker.cpp:87).
  int EDline(const uint8_t* image, std::vector<Line>& lines, bool smoothed = false) {
    if (!image) return -1;
    std::vector<vpl_line> out(dev_.max_lines());
    int n = 0;
    if (vpl_edlines_detect_batch_ex(dev_.ctx(), 1, image, &p_, smoothed ? 1 : 0, out.data(), &n) != 0) return -1;
    lines.clear();
    for (int i = 0; i < n; ++i) lines.push_back(to_line(out[i]));
    return 1;
  }
#ifdef VPL_USE_OPENCV
  int EDline(cv::Mat& image, std::vector<Line>& lines, bool smoothed = false) {
    if (image.type() != CV_8UC1 || !image.isContinuous() || image.cols != dev_.width() || image.rows != dev_.height()) return -1;
    return EDline(image.data, lines, smoothed);
  }
#endif

  const vpl_edline_param& param() const { return p_; }

 private:
  FrontendDevice& dev_;
  vpl_edline_param p_;
};

class LineMatching {
 public:
  LineMatching(FrontendDevice& dev, int step = 10, float closest_line_threshold = 0.5f, float line_matching_ratio = 0.4f,
               float line_distance_error_ratio = 3.f, float klt_error_threshold = 40.f)
      : dev_(dev) {
    vpl_match_default_param(&p_);
    p_.step = step; p_.closest_line_threshold = closest_line_threshold; p_.line_matching_ratio = line_matching_ratio;
    p_.line_distance_error_ratio = line_distance_error_ratio; p_.klt_error_threshold = klt_error_threshold;
  }
  // bool LineMatching::Matching(img_ref, img_cur, lines_ref, lines_cur, line_ref_to_line_cur, K_ref, K_cur, T_cur_ref,
  //                             illumination_adapt, topological_filter, ...) with K/T = NULL (line_matching.cpp:605-690).
  // Returns false (and leaves line_ref_to_line_cur untouched) when a line list is empty; throws on device errors.
  bool Matching(const uint8_t* img_ref, const uint8_t* img_cur, const std::vector<Line>& lines_ref,
                const std::vector<Line>& lines_cur, std::vector<int>& line_ref_to_line_cur,
                bool illumination_adapt = false, bool topological_filter = true) {
    const size_t px = (size_t)dev_.width() * dev_.height();
    const int ML = dev_.max_lines();
    if ((int)lines_ref.size() > ML || (int)lines_cur.size() > ML) throw std::runtime_error("more lines than max_lines");
    std::vector<uint8_t> two(2 * px);
    std::memcpy(two.data(), img_ref, px);
    std::memcpy(two.data() + px, img_cur, px);
    std::vector<vpl_line> lr(ML), lc(ML);
    for (size_t i = 0; i < lines_ref.size(); ++i) lr[i] = from_line(lines_ref[i]);
    for (size_t i = 0; i < lines_cur.size(); ++i) lc[i] = from_line(lines_cur[i]);
    const int ref = 0, cur = 1, nr = (int)lines_ref.size(), nc = (int)lines_cur.size();
    vpl_match_param p = p_;
    p.illumination_adapt = illumination_adapt ? 1 : 0;
    p.topological_filter = topological_filter ? 1 : 0;
    std::vector<int> r2c(ML, -1);
    int matched = 0;
    if (vpl_line_match_batch(dev_.ctx(), 2, two.data(), 1, &ref, &cur, lr.data(), &nr, lc.data(), &nc, &p, r2c.data(),
                             &matched) != 0)
      throw std::runtime_error(std::string("vpl_line_match_batch: ") + vpl_fe_last_error(dev_.ctx()));
    if (!matched) return false;
    line_ref_to_line_cur.assign(r2c.begin(), r2c.begin() + nr);
    return true;
  }
  // void LineMatching::LineFilter(std::vector<Line>& lines, float distance_threshold, float parallel_threshold = sin 3 deg)
  // (line_matching.h:35-37, line_matching.cpp:167-264); throws on device errors
  void LineFilter(std::vector<Line>& lines, float distance_threshold, float parallel_threshold = 0.0348994967f) {
    const int ML = dev_.max_lines();
    if ((int)lines.size() > ML) throw std::runtime_error("more lines than max_lines");
    std::vector<vpl_line> l(ML);
    for (size_t i = 0; i < lines.size(); ++i) l[i] = from_line(lines[i]);
    int n = (int)lines.size();
    if (vpl_line_filter_batch(dev_.ctx(), 1, l.data(), &n, distance_threshold, parallel_threshold) != 0)
      throw std::runtime_error(std::string("vpl_line_filter_batch: ") + vpl_fe_last_error(dev_.ctx()));
    lines.clear();
    for (int i = 0; i < n; ++i) lines.push_back(to_line(l[i]));
  }
#ifdef VPL_USE_OPENCV
  bool Matching(const cv::Mat& img_ref, const cv::Mat& img_cur, const std::vector<Line>& lines_ref,
                const std::vector<Line>& lines_cur, std::vector<int>& line_ref_to_line_cur,
                bool illumination_adapt = false, bool topological_filter = true) {
    return Matching(img_ref.data, img_cur.data, lines_ref, lines_cur, line_ref_to_line_cur, illumination_adapt,
                    topological_filter);
  }
#endif

 private:
  FrontendDevice& dev_;
  vpl_match_param p_;
};

// FrameLines / LineFeatureTracker (feature_tracker/include/linefeature_tracker.h:45-96, src/line_feature_tracker.cpp).
// readImage = remap + CLAHE (device) -> EDline (device) -> Matching against the previous frame (device) -> id / quota
// lists (vpl_line_track_ids) -> vanishing points (device, vpl_vp_detect_batch).  The reference seeds rand() with
// time(NULL) inside the VP stage; here the seed comes from vp_seed() (default: time(NULL), settable for reproducible runs).
struct FrameLines {
  std::vector<uint8_t> img;                       // prepared frame (undistorted, equalised)
  std::vector<Line> vecLine;
  std::vector<int> lineID;
  std::vector<int> t_cnt;
  std::vector<std::array<double, 4>> vps;
};

class LineFeatureTracker {
 public:
  LineFeatureTracker(FrontendDevice& dev, EDLineParam edparam, int max_h_lines, int max_v_lines, bool equalize = true)
      : dev_(dev), line_detctor(dev, edparam), line_matching(dev), max_h_lines_(max_h_lines), max_v_lines_(max_v_lines),
        equalize_(equalize) {}

  // readIntrinsicParameter (:26-35): the maps of PinholeCamera::initUndistortRectifyMap and its K_rect
  void setUndistortMaps(const float* map_x, const float* map_y, float fx, float fy, float cx, float cy) {
    if (vpl_pre_set_maps(dev_.ctx(), map_x, map_y) != 0)
      throw std::runtime_error(std::string("vpl_pre_set_maps: ") + vpl_fe_last_error(dev_.ctx()));
    fx_ = fx; fy_ = fy; cx_ = cx; cy_ = cy;
  }

  // undistortedLineEndPoints (:37-53): pixel end points of curframe_ to the normalised plane of K_
  std::vector<Line> undistortedLineEndPoints() const {
    std::vector<Line> un = curframe_->vecLine;
    for (Line& l : un) {
      l.line_endpoint[0] = (l.line_endpoint[0] - cx_) / fx_;
      l.line_endpoint[1] = (l.line_endpoint[1] - cy_) / fy_;
      l.line_endpoint[2] = (l.line_endpoint[2] - cx_) / fx_;
      l.line_endpoint[3] = (l.line_endpoint[3] - cy_) / fy_;
    }
    return un;
  }

  // readImage (:57-286), raw = the 8-bit frame as it arrives from the camera
  void readImage(const uint8_t* raw) {
    const size_t px = (size_t)dev_.width() * dev_.height();
    lines_exit = true;
    last_match.clear();
    std::vector<uint8_t> img(px);
    if (vpl_pre_batch(dev_.ctx(), 1, raw, equalize_ ? 1 : 0, 3.0, 8, 8, img.data()) != 0)
      throw std::runtime_error(std::string("vpl_pre_batch: ") + vpl_fe_last_error(dev_.ctx()));
    bool first_img = false;
    if (!forwframe_) {
      forwframe_.reset(new FrameLines);
      curframe_.reset(new FrameLines);
      forwframe_->img = img;
      curframe_->img = img;
      first_img = true;
    } else {
      forwframe_.reset(new FrameLines);
      forwframe_->img = img;
    }
    // the prepared frame is still resident: detect without a second upload
    std::vector<vpl_line> det(dev_.max_lines());
    int n_det = 0;
    vpl_edline_param ep = line_detctor.param();
    if (vpl_edlines_detect(dev_.ctx(), &ep) != 0 || vpl_fe_synchronize(dev_.ctx()) != 0 ||
        vpl_edlines_download(dev_.ctx(), 1, det.data(), &n_det) != 0)
      throw std::runtime_error(std::string("vpl_edlines_detect: ") + vpl_fe_last_error(dev_.ctx()));
    for (int i = 0; i < n_det; ++i) forwframe_->vecLine.push_back(to_line(det[i]));
    last_detected = forwframe_->vecLine;
    if (forwframe_->vecLine.empty()) { lines_exit = false; return; }
    for (size_t i = 0; i < forwframe_->vecLine.size(); ++i) {
      forwframe_->lineID.push_back(first_img ? allfeature_cnt++ : -1);
      forwframe_->t_cnt.push_back(0);
    }
    if (!curframe_->vecLine.empty()) {
      std::vector<int> line_prev_to_line_cur;
      line_matching.Matching(curframe_->img.data(), forwframe_->img.data(), curframe_->vecLine, forwframe_->vecLine,
                             line_prev_to_line_cur, true, true);
      last_match = line_prev_to_line_cur;
      const int n_new = (int)forwframe_->vecLine.size();
      const int n_prev = (int)line_prev_to_line_cur.size();     // 0 when Matching() returned false
      std::vector<float> ends(4 * (size_t)n_new);
      for (int i = 0; i < n_new; ++i)
        for (int k = 0; k < 4; ++k) ends[4 * i + k] = forwframe_->vecLine[i].line_endpoint[k];
      std::vector<int> keep(n_new), ids(n_new), tc(n_new), vert(n_new);
      int n_vert = 0;
      const int n_keep = vpl_line_track_ids(n_new, ends.data(), n_prev, curframe_->lineID.data(), curframe_->t_cnt.data(),
                                            (int)curframe_->t_cnt.size(), line_prev_to_line_cur.data(), max_h_lines_,
                                            max_v_lines_, &allfeature_cnt, keep.data(), ids.data(), tc.data(), vert.data(),
                                            &n_vert);
      if (n_keep < 0) throw std::runtime_error("vpl_line_track_ids failed");
      std::vector<Line> kept, verticalLine;
      for (int k = 0; k < n_keep; ++k) kept.push_back(forwframe_->vecLine[keep[k]]);
      for (int k = 0; k < n_vert; ++k) verticalLine.push_back(forwframe_->vecLine[vert[k]]);
      forwframe_->vecLine.swap(kept);
      forwframe_->lineID.assign(ids.begin(), ids.begin() + n_keep);
      forwframe_->t_cnt = tc;                                   // stays in detection order (:227-228 swap only two vectors)
      // vanishing points (:233-279)
      const std::array<double, 4> none{0.0, 0.0, 0.0, 0.0};
      forwframe_->vps.clear();
      last_vp_ids.clear();
      if (forwframe_->vecLine.size() > 2) {
        const std::vector<Line>& hyp = verticalLine.size() > 2 ? verticalLine : forwframe_->vecLine;
        const int ML = dev_.max_lines();
        std::vector<vpl_line> lh(ML), la(ML);
        for (size_t i = 0; i < hyp.size(); ++i) lh[i] = from_line(hyp[i]);
        for (size_t i = 0; i < forwframe_->vecLine.size(); ++i) la[i] = from_line(forwframe_->vecLine[i]);
        const int nh = (int)hyp.size(), na = (int)forwframe_->vecLine.size();
        const uint32_t seed = vp_seed();
        const int first = vp_frame_count_ == 0 ? 1 : 0;
        double vps[9];
        int status = 0;
        std::vector<int> local_vp_ids(ML, 3);
        if (vpl_vp_detect_batch(dev_.ctx(), 1, lh.data(), &nh, la.data(), &na, fx_, cx_, cy_, &seed, &first, vps,
                                local_vp_ids.data(), &status) != 0)
          throw std::runtime_error(std::string("vpl_vp_detect_batch: ") + vpl_fe_last_error(dev_.ctx()));
        ++vp_frame_count_;
        last_vp_seed = seed;
        std::memcpy(last_vps, vps, sizeof(vps));
        for (int i = 0; i < na; ++i) {
          const int id = status == 0 ? local_vp_ids[i] : 3;
          last_vp_ids.push_back(id);
          if (id == 3) { forwframe_->vps.push_back(none); continue; }
          const double* v = vps + 3 * id;
          forwframe_->vps.push_back(std::array<double, 4>{v[0], v[1], v[2], v[2] / v[2]});      // (:256)
        }
      } else {
        forwframe_->vps.assign(forwframe_->vecLine.size(), none);
      }
    }
    curframe_.swap(forwframe_);
  }

  // The wire format of line_feature_tracker_node.cpp:77-153 for one camera: what Estimator::processImage receives
  // per line id (estimator_node.cpp:375-407): x1, y1, x2, y2 normalised, vp x y z, vp flag.  As at :104-109 every
  // line carries vps[camera index], not its own entry.
  struct LineObservation { int id; double v[8]; };
  std::vector<LineObservation> lineObservations() const {
    std::vector<LineObservation> out;
    if (!lines_exit || !curframe_) return out;
    const std::vector<Line> un = undistortedLineEndPoints();
    for (size_t j = 0; j < curframe_->lineID.size(); ++j) {
      LineObservation ob;
      ob.id = curframe_->lineID[j];
      for (int k = 0; k < 4; ++k) ob.v[k] = un[j].line_endpoint[k];
      for (int k = 0; k < 4; ++k) ob.v[4 + k] = curframe_->vps.empty() ? 0.0 : curframe_->vps[0][k];
      out.push_back(ob);
    }
    return out;
  }

  std::shared_ptr<FrameLines> curframe_, forwframe_;
  int allfeature_cnt = 0;
  bool lines_exit = true;
  std::function<uint32_t()> vp_seed = [] { return (uint32_t)std::time(nullptr); };   // srand((unsigned)time(NULL)), :107
  std::vector<int> last_vp_ids;                   // local_vp_ids of the last readImage (test access)
  double last_vps[9] = {0};
  uint32_t last_vp_seed = 0;
  std::vector<int> last_match;                    // line_prev_to_line_cur of the last readImage (test access)
  std::vector<Line> last_detected;                // the last frame's detections before the quota (test access)

 private:
  FrontendDevice& dev_;
  EDLineDetector line_detctor;
  LineMatching line_matching;
  int max_h_lines_, max_v_lines_;
  bool equalize_;
  float fx_ = 1.f, fy_ = 1.f, cx_ = 0.f, cy_ = 0.f;
  int vp_frame_count_ = 0;
};

}  // namespace vplhost
