// TEST INFRASTRUCTURE -- CPU oracle (see smallmat.h).
// Restates the body of Estimator::optimizationwithLine()
//   vins_estimator/src/estimator.cpp:1043-1453
// with vector2double (:650-705), double2vector2 (:810-900),
// FeatureManager::{getLineOrthVector,setLineOrth,removeLineOutlier}
//   (feature_manager.cpp:341-388,702-798)
// over the window container of include/vplines_ba.h (data format only; no
// product code is linked here).
#include <array>
#include "../include/vplines_ba.h"
#include "marginalization.h"
#include "problem.h"

namespace orc {
// TEST HOOK: scales ceres' three termination tolerances in the main solve (tests/test_oracle_fixed_point.py runs the solve to
// its fixed point to compare it with an independent least-squares solver).  0 = off.
double g_test_tolerance_scale = 0.0;


namespace {

struct LineTrack {
  int start_frame;
  std::vector<Mat<8, 1>> obs;  // lineobs (4) + lineobs_vp (4)
  Vec6 line_plucker;           // start camera frame
  bool is_triangulation = true;
  int index = 0;               // position in the vpl_window arrays
};
struct PointTrack {
  int start_frame;
  std::vector<Vec3> obs;
  double estimated_depth;
  int solve_flag = 0;
  int index = 0;               // position in the vpl_window arrays
};

// The slice of Estimator state that optimizationwithLine touches.  Member order
// of the para_* arrays follows estimator.h:139-143 (it defines the block order
// of the marginalisation, see marginalization.h).
struct Est {
  Vec3 Ps[VPL_NFRAMES], Vs[VPL_NFRAMES], Bas[VPL_NFRAMES], Bgs[VPL_NFRAMES];
  Mat3 Rs[VPL_NFRAMES];
  Vec3 tic;
  Mat3 ric;
  std::vector<PointTrack> feature;
  std::vector<LineTrack> linefeature;
  IntegrationBase* pre_integrations[VPL_NFRAMES] = {nullptr};
  Vec3 G;

  double para_Pose[VPL_NFRAMES][7];
  double para_SpeedBias[VPL_NFRAMES][9];
  std::vector<std::array<double, 1>> para_Feature;
  std::vector<std::array<double, 4>> para_LineFeature;
  double para_Ex_Pose[1][7];

  MarginalizationInfo* last_marginalization_info = nullptr;
  std::vector<double*> last_marginalization_parameter_blocks;
  bool failure_occur = false;   // estimator.h: failure_occur, last_R0, last_P0
  Mat3 last_R0;
  Vec3 last_P0;

  ~Est() {
    for (auto* p : pre_integrations) delete p;
    delete last_marginalization_info;
  }

  // estimator.cpp:650-705
  void vector2double() {
    for (int i = 0; i <= VPL_WINDOW_SIZE; i++) {
      para_Pose[i][0] = Ps[i][0]; para_Pose[i][1] = Ps[i][1]; para_Pose[i][2] = Ps[i][2];
      Quat q = Quat::fromRotationMatrix(Rs[i]);
      para_Pose[i][3] = q.x; para_Pose[i][4] = q.y; para_Pose[i][5] = q.z; para_Pose[i][6] = q.w;
      for (int k = 0; k < 3; ++k) {
        para_SpeedBias[i][k] = Vs[i][k];
        para_SpeedBias[i][3 + k] = Bas[i][k];
        para_SpeedBias[i][6 + k] = Bgs[i][k];
      }
    }
    para_Ex_Pose[0][0] = tic[0]; para_Ex_Pose[0][1] = tic[1]; para_Ex_Pose[0][2] = tic[2];
    Quat q = Quat::fromRotationMatrix(ric);
    para_Ex_Pose[0][3] = q.x; para_Ex_Pose[0][4] = q.y; para_Ex_Pose[0][5] = q.z; para_Ex_Pose[0][6] = q.w;
    // getDepthVector (feature_manager.cpp:277-293)
    for (size_t i = 0; i < feature.size(); ++i) para_Feature[i][0] = 1. / feature[i].estimated_depth;
    // getLineOrthVector (feature_manager.cpp:341-365)
    for (size_t i = 0; i < linefeature.size(); ++i) {
      int imu_i = linefeature[i].start_frame;
      Vec3 twc = Ps[imu_i] + Rs[imu_i] * tic;
      Mat3 Rwc = Rs[imu_i] * ric;
      Vec6 line_w = plk_to_pose(linefeature[i].line_plucker, Rwc, twc);
      Vec4 o = plk_to_orth(line_w);
      for (int k = 0; k < 4; ++k) para_LineFeature[i][k] = o[k];
    }
  }

  // estimator.cpp:810-900 (double2vector2) and :707-808 (double2vector: identical except that the lines are not carried
  // through the yaw/position gauge transform -- setLineOrth receives the optimised orth as it is)
  void double2vector2(bool line_gauge = true) {
    Vec3 origin_R0 = R2ypr(Rs[0]);
    Vec3 origin_P0 = Ps[0];
    if (failure_occur) {   // estimator.cpp:818-823
      origin_R0 = R2ypr(last_R0);
      origin_P0 = last_P0;
      failure_occur = 0;
    }
    Vec3 origin_R00 = R2ypr(Quat(para_Pose[0][6], para_Pose[0][3], para_Pose[0][4], para_Pose[0][5]).toRotationMatrix());
    double y_diff = origin_R0[0] - origin_R00[0];
    Mat3 rot_diff = ypr2R(Vec3{y_diff, 0, 0});
    for (int i = 0; i <= VPL_WINDOW_SIZE; i++) {
      Rs[i] = rot_diff * Quat(para_Pose[i][6], para_Pose[i][3], para_Pose[i][4], para_Pose[i][5]).normalized().toRotationMatrix();
      Ps[i] = rot_diff * Vec3{para_Pose[i][0] - para_Pose[0][0], para_Pose[i][1] - para_Pose[0][1], para_Pose[i][2] - para_Pose[0][2]} + origin_P0;
      Vs[i] = rot_diff * Vec3{para_SpeedBias[i][0], para_SpeedBias[i][1], para_SpeedBias[i][2]};
      Bas[i] = Vec3{para_SpeedBias[i][3], para_SpeedBias[i][4], para_SpeedBias[i][5]};
      Bgs[i] = Vec3{para_SpeedBias[i][6], para_SpeedBias[i][7], para_SpeedBias[i][8]};
    }
    tic = Vec3{para_Ex_Pose[0][0], para_Ex_Pose[0][1], para_Ex_Pose[0][2]};
    ric = Quat(para_Ex_Pose[0][6], para_Ex_Pose[0][3], para_Ex_Pose[0][4], para_Ex_Pose[0][5]).toRotationMatrix();

    Mat3 Rwow1 = rot_diff;
    Vec3 tw1b{para_Pose[0][0], para_Pose[0][1], para_Pose[0][2]};
    Vec3 twow1 = -(Rwow1 * tw1b) + origin_P0;
    for (size_t i = 0; i < linefeature.size(); ++i) {
      Vec4 orth{para_LineFeature[i][0], para_LineFeature[i][1], para_LineFeature[i][2], para_LineFeature[i][3]};
      if (line_gauge) {
        Vec6 line_w1 = orth_to_plk(orth);
        Vec6 line_wo = plk_to_pose(line_w1, Rwow1, twow1);
        orth = plk_to_orth(line_wo);
      }
      // setLineOrth (feature_manager.cpp:367-388)
      Vec6 line_w = orth_to_plk(orth);
      int imu_i = linefeature[i].start_frame;
      Vec3 twc = Ps[imu_i] + Rs[imu_i] * tic;
      Mat3 Rwc = Rs[imu_i] * ric;
      linefeature[i].line_plucker = plk_from_pose(line_w, Rwc, twc);
    }
    // setDepth (feature_manager.cpp:229-251)
    for (size_t i = 0; i < feature.size(); ++i) {
      feature[i].estimated_depth = 1.0 / para_Feature[i][0];
      feature[i].solve_flag = feature[i].estimated_depth < 0 ? 2 : 1;
    }
  }

  // FeatureManager::triangulate, feature_manager.cpp:565-621: DLT over all observations of a track in the start camera
  // frame; the depth is V(2)/V(3) of the right singular vector of the smallest singular value (Eigen::JacobiSVD in the
  // reference, third party; restated as a one-sided Jacobi SVD on the four columns).  Returns the number of tracks done.
  int triangulate(double init_depth) {
    int done = 0;
    for (PointTrack& F : feature) {
      if (F.estimated_depth > 0) continue;
      const int imu_i = F.start_frame;
      int imu_j = imu_i - 1;
      Vec3 t0 = Ps[imu_i] + Rs[imu_i] * tic;
      Mat3 R0 = Rs[imu_i] * ric;
      const int m = 2 * (int)F.obs.size();
      std::vector<double> A((size_t)m * 4);
      int row = 0;
      for (auto& ob : F.obs) {
        imu_j++;
        Vec3 t1 = Ps[imu_j] + Rs[imu_j] * tic;
        Mat3 R1 = Rs[imu_j] * ric;
        Vec3 t = R0.T() * (t1 - t0);
        Mat3 R = R0.T() * R1;
        Mat3 Rt = R.T();
        Vec3 mt = -(Rt * t);
        double P[3][4];
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) P[r][c] = Rt(r, c); P[r][3] = mt[r]; }
        Vec3 f = ob / ob.norm();
        for (int c = 0; c < 4; ++c) A[(size_t)row * 4 + c] = f[0] * P[2][c] - f[2] * P[0][c];
        ++row;
        for (int c = 0; c < 4; ++c) A[(size_t)row * 4 + c] = f[1] * P[2][c] - f[2] * P[1][c];
        ++row;
      }
      double V[16];
      svd4_right(A.data(), m, V);
      // columns of V sorted by decreasing singular value: the last one belongs to the smallest
      double depth = V[2 * 4 + 3] / V[3 * 4 + 3];
      if (depth < 0.1) depth = init_depth;
      F.estimated_depth = depth;
      ++done;
    }
    return done;
  }

  // FeatureManager::triangulateLine, feature_manager.cpp:413-563; returns the number of lines triangulated now
  int triangulateLine() {
    int done = 0;
    for (LineTrack& L : linefeature) {
      if (L.is_triangulation) continue;
      int imu_i = L.start_frame, imu_j = imu_i - 1;
      Vec3 t0 = Ps[imu_i] + Rs[imu_i] * tic;
      Mat3 R0 = Rs[imu_i] * ric;
      double min_cos_theta = 1.0;
      Vec3 tij{0, 0, 0};
      Mat3 Rij;
      Vec4 obsj{0, 0, 0, 0};
      Vec4 pii{0, 0, 0, 0};
      Vec3 ni{0, 0, 0};
      for (auto& ob : L.obs) {
        imu_j++;
        if (imu_j == imu_i) {
          Vec3 p1{ob[0], ob[1], 1}, p2{ob[2], ob[3], 1};
          pii = pi_from_ppp(p1, p2, Vec3{0, 0, 0});
          ni = Vec3{pii[0], pii[1], pii[2]};
          ni = ni / ni.norm();
          continue;
        }
        Vec3 t1 = Ps[imu_j] + Rs[imu_j] * tic;
        Mat3 R1 = Rs[imu_j] * ric;
        Vec3 t = R0.T() * (t1 - t0);
        Mat3 R = R0.T() * R1;
        Vec3 p3{ob[0], ob[1], 1}, p4{ob[2], ob[3], 1};
        p3 = R * p3 + t;
        p4 = R * p4 + t;
        Vec4 pij = pi_from_ppp(p3, p4, t);
        Vec3 nj{pij[0], pij[1], pij[2]};
        nj = nj / nj.norm();
        double cos_theta = ni.dot(nj);
        if (cos_theta < min_cos_theta) {
          min_cos_theta = cos_theta;
          tij = t;
          Rij = R;
          obsj = Vec4{ob[0], ob[1], ob[2], ob[3]};
        }
      }
      if (min_cos_theta > 0.998) continue;
      Vec3 p3{obsj[0], obsj[1], 1}, p4{obsj[2], obsj[3], 1};
      p3 = Rij * p3 + tij;
      p4 = Rij * p4 + tij;
      Vec4 pij = pi_from_ppp(p3, p4, tij);
      L.line_plucker = pipi_plk(pii, pij);
      L.is_triangulation = true;
      ++done;
    }
    return done;
  }

  // feature_manager.cpp:390-411
  static double reprojection_error(const Vec4& obs, const Mat3& Rwc, const Vec3& twc, const Vec6& line_w) {
    Vec3 p1{obs[0], obs[1], 1}, p2{obs[2], obs[3], 1};
    Vec6 line_c = plk_from_pose(line_w, Rwc, twc);
    Vec3 nc{line_c[0], line_c[1], line_c[2]};
    double sql = std::sqrt(nc[0] * nc[0] + nc[1] * nc[1]);
    nc = nc / sql;
    return (std::fabs(nc.dot(p1)) + std::fabs(nc.dot(p2))) / 2.0;
  }

  // feature_manager.cpp:702-798 ; returns the number of erased lines
  int removeLineOutlier(std::vector<int>* removed_index) {
    int removed = 0;
    std::vector<LineTrack> kept;
    for (size_t li = 0; li < linefeature.size(); ++li) {
      LineTrack& L = linefeature[li];
      int imu_i = L.start_frame, imu_j = imu_i - 1;
      Vec3 twc = Ps[imu_i] + Rs[imu_i] * tic;
      Mat3 Rwc = Rs[imu_i] * ric;
      Vec3 nc{L.line_plucker[0], L.line_plucker[1], L.line_plucker[2]};
      Vec3 vc{L.line_plucker[3], L.line_plucker[4], L.line_plucker[5]};
      Mat<4, 4> Lc;
      Lc.setBlock<3, 3>(0, 0, skew(nc));
      for (int k = 0; k < 3; ++k) { Lc(k, 3) = vc[k]; Lc(3, k) = -vc[k]; }
      const Mat<8, 1>& o0 = L.obs[0];
      Vec3 p11{o0[0], o0[1], 1.0}, p21{o0[2], o0[3], 1.0};
      Vec3 cr = cross(p11, p21);
      Vec2 ln{cr[0], cr[1]};
      ln = ln / ln.norm();
      Vec3 p12{p11[0] + ln[0], p11[1] + ln[1], 1.0};
      Vec3 p22{p21[0] + ln[0], p21[1] + ln[1], 1.0};
      Vec3 cam{0, 0, 0};
      Vec4 pi1 = pi_from_ppp(cam, p11, p12);
      Vec4 pi2 = pi_from_ppp(cam, p21, p22);
      Vec4 e1 = Lc * pi1, e2 = Lc * pi2;
      e1 = e1 / e1[3];
      e2 = e2 / e2[3];
      bool erase = false;
      if (e1[2] < 0 || e2[2] < 0) erase = true;
      else if ((e1 - e2).norm() > 10) erase = true;
      else {
        Vec6 line_w = plk_to_pose(L.line_plucker, Rwc, twc);
        double allerr = 0;
        for (auto& ob : L.obs) {
          imu_j++;
          Vec4 obs{ob[0], ob[1], ob[2], ob[3]};
          Vec3 t1 = Ps[imu_j] + Rs[imu_j] * tic;
          Mat3 R1 = Rs[imu_j] * ric;
          double err = reprojection_error(obs, R1, t1, line_w);
          if (allerr < err) allerr = err;
        }
        if (allerr > 3.0 / 500.0) erase = true;
      }
      if (erase) { ++removed; if (removed_index) removed_index->push_back(L.index); }
      else kept.push_back(L);
    }
    linefeature.swap(kept);
    para_LineFeature.resize(linefeature.size());
    return removed;
  }
};

void load_window(const vpl_window& w, const vpl_ba_options& opt, Est& e, bool all_lines = false) {
  for (int i = 0; i < VPL_NFRAMES; ++i) {
    e.Ps[i] = Vec3{w.pose[i][0], w.pose[i][1], w.pose[i][2]};
    e.Rs[i] = Quat(w.pose[i][6], w.pose[i][3], w.pose[i][4], w.pose[i][5]).normalized().toRotationMatrix();
    e.Vs[i] = Vec3{w.speed_bias[i][0], w.speed_bias[i][1], w.speed_bias[i][2]};
    e.Bas[i] = Vec3{w.speed_bias[i][3], w.speed_bias[i][4], w.speed_bias[i][5]};
    e.Bgs[i] = Vec3{w.speed_bias[i][6], w.speed_bias[i][7], w.speed_bias[i][8]};
  }
  e.failure_occur = w.failure_occur != 0;
  if (e.failure_occur) {
    e.last_P0 = Vec3{w.last_P0[0], w.last_P0[1], w.last_P0[2]};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) e.last_R0(r, c) = w.last_R0[3 * r + c];
  }
  e.tic = Vec3{w.ex_pose[0], w.ex_pose[1], w.ex_pose[2]};
  e.ric = Quat(w.ex_pose[6], w.ex_pose[3], w.ex_pose[4], w.ex_pose[5]).normalized().toRotationMatrix();
  e.G = Vec3{0, 0, opt.g_norm};
  int off = 0;
  for (int i = 0; i < w.n_points; ++i) {
    PointTrack t;
    t.start_frame = w.point_start[i];
    for (int k = 0; k < w.point_nobs[i]; ++k, ++off)
      t.obs.push_back(Vec3{w.point_obs[3 * off], w.point_obs[3 * off + 1], w.point_obs[3 * off + 2]});
    t.estimated_depth = 1.0 / w.inv_depth[i];
    t.index = i;
    e.feature.push_back(t);
  }
  off = 0;
  for (int i = 0; i < w.n_lines; ++i) {
    LineTrack t;
    t.start_frame = w.line_start[i];
    for (int k = 0; k < w.line_nobs[i]; ++k, ++off) {
      Mat<8, 1> o;
      for (int c = 0; c < 8; ++c) o[c] = w.line_obs[8 * off + c];
      t.obs.push_back(o);
    }
    for (int c = 0; c < 6; ++c) t.line_plucker[c] = w.line_plk[6 * i + c];
    t.index = i;
    t.is_triangulation = !w.line_triangulated || w.line_triangulated[i] != 0;
    // the solve sees the lines that pass the filter of estimator.cpp:1132-1133, of which is_triangulation is a part
    if (all_lines || t.is_triangulation) e.linefeature.push_back(t);
  }
  e.para_Feature.resize(e.feature.size());
  e.para_LineFeature.resize(e.linefeature.size());
  ImuNoise nz{opt.acc_n, opt.gyr_n, opt.acc_w, opt.gyr_w};
  for (int j = 1; j < VPL_NFRAMES; ++j) {
    const vpl_preintegration& p = w.preint[j];
    IntegrationBase* ib = new IntegrationBase(Vec3{}, Vec3{}, Vec3{p.linearized_ba[0], p.linearized_ba[1], p.linearized_ba[2]},
                                              Vec3{p.linearized_bg[0], p.linearized_bg[1], p.linearized_bg[2]}, nz);
    ib->sum_dt = p.sum_dt;
    ib->delta_p = Vec3{p.delta_p[0], p.delta_p[1], p.delta_p[2]};
    ib->delta_q = Quat(p.delta_q[3], p.delta_q[0], p.delta_q[1], p.delta_q[2]);
    ib->delta_v = Vec3{p.delta_v[0], p.delta_v[1], p.delta_v[2]};
    for (int k = 0; k < 225; ++k) { ib->jacobian.a[k] = p.jacobian[k]; ib->covariance.a[k] = p.covariance[k]; }
    e.pre_integrations[j] = ib;
  }
  if (w.has_prior && w.prior) {
    const vpl_prior& pr = *w.prior;
    MarginalizationInfo* mi = new MarginalizationInfo();
    mi->n = pr.n;
    mi->m = 0;
    mi->linearized_jacobians.resize(pr.n, pr.n);
    for (int r = 0; r < pr.n; ++r)
      for (int c = 0; c < pr.n; ++c) mi->linearized_jacobians(r, c) = pr.J0[(size_t)r * pr.n + c];
    mi->linearized_residuals.assign(pr.r0, pr.r0 + pr.n);
    mi->owned_keep_data.resize(pr.n_blocks);
    for (int b = 0; b < pr.n_blocks; ++b) {
      int size = pr.block_kind[b] == VPL_BLOCK_SPEEDBIAS ? 9 : 7;
      mi->keep_block_size.push_back(size);
      mi->keep_block_idx.push_back(pr.block_idx[b]);
      mi->owned_keep_data[b].assign(pr.x0[b], pr.x0[b] + size);
    }
    for (int b = 0; b < pr.n_blocks; ++b) {
      mi->keep_block_data.push_back(mi->owned_keep_data[b].data());
      double* addr = pr.block_kind[b] == VPL_BLOCK_POSE ? e.para_Pose[pr.block_frame[b]]
                     : pr.block_kind[b] == VPL_BLOCK_SPEEDBIAS ? e.para_SpeedBias[pr.block_frame[b]]
                                                                : e.para_Ex_Pose[0];
      e.last_marginalization_parameter_blocks.push_back(addr);
    }
    e.last_marginalization_info = mi;
  }
}

void export_prior(Est& e, const std::vector<double*>& blocks, vpl_prior* out) {
  MarginalizationInfo* mi = e.last_marginalization_info;
  std::memset(out, 0, sizeof(*out));
  out->n = mi->n;
  out->n_blocks = (int)blocks.size();
  for (size_t b = 0; b < blocks.size(); ++b) {
    double* a = blocks[b];
    int kind = -1, frame = 0;
    for (int i = 0; i < VPL_NFRAMES; ++i) {
      if (a == e.para_Pose[i]) { kind = VPL_BLOCK_POSE; frame = i; }
      if (a == e.para_SpeedBias[i]) { kind = VPL_BLOCK_SPEEDBIAS; frame = i; }
    }
    if (a == e.para_Ex_Pose[0]) { kind = VPL_BLOCK_EXPOSE; frame = 0; }
    out->block_kind[b] = kind;
    out->block_frame[b] = frame;
    out->block_idx[b] = mi->keep_block_idx[b] - mi->m;
    for (int k = 0; k < mi->keep_block_size[b]; ++k) out->x0[b][k] = mi->keep_block_data[b][k];
  }
  for (int r = 0; r < mi->n; ++r) {
    out->r0[r] = mi->linearized_residuals[r];
    for (int c = 0; c < mi->n; ++c) out->J0[(size_t)r * mi->n + c] = mi->linearized_jacobians(r, c);
  }
}

}  // namespace

// Estimator::optimizationwithLine, estimator.cpp:1043-1453
int solve_window(vpl_window* w, const vpl_ba_options* opt, vpl_prior* prior_out, vpl_solve_report* rep,
                 double* A_final_out /* optional n*n */, double* b_final_out /* optional n */) {
  Est e;
  load_window(*w, *opt, e);
  ProjectionFactor::sqrt_info = opt->focal_length / 1.5;   // estimator.cpp:18
  lineProjectionFactor::sqrt_info = opt->line_factor;      // :19
  vpProjectionFactor::sqrt_info = opt->vp_factor;          // :20
  vpl_solve_report report;
  std::memset(&report, 0, sizeof(report));

  LossFunction* loss_function = new HuberLoss(opt->huber_delta);
  {
    Problem problem;
    for (int i = 0; i < VPL_WINDOW_SIZE + 1; i++) {
      problem.AddParameterBlock(e.para_Pose[i], 7, new PoseLocalParameterization());
      problem.AddParameterBlock(e.para_SpeedBias[i], 9);
    }
    problem.AddParameterBlock(e.para_Ex_Pose[0], 7, new PoseLocalParameterization());
    if (!opt->estimate_extrinsic) problem.SetParameterBlockConstant(e.para_Ex_Pose[0]);

    e.vector2double();

    if (e.last_marginalization_info) {
      MarginalizationFactor* mf = new MarginalizationFactor(e.last_marginalization_info);
      problem.AddResidualBlock(mf, nullptr, e.last_marginalization_parameter_blocks);
    }
    for (int i = 0; i < VPL_WINDOW_SIZE; i++) {
      int j = i + 1;
      if (e.pre_integrations[j]->sum_dt > 10.0) continue;
      problem.AddResidualBlock(new IMUFactor(e.pre_integrations[j], e.G), nullptr,
                               {e.para_Pose[i], e.para_SpeedBias[i], e.para_Pose[j], e.para_SpeedBias[j]});
    }
    int feature_index = -1;
    for (auto& it_per_id : e.feature) {
      ++feature_index;
      int imu_i = it_per_id.start_frame, imu_j = imu_i - 1;
      Vec3 pts_i = it_per_id.obs[0];
      for (auto& pts_j : it_per_id.obs) {
        imu_j++;
        if (imu_i == imu_j) continue;
        problem.AddResidualBlock(new ProjectionFactor(pts_i, pts_j), loss_function,
                                 {e.para_Pose[imu_i], e.para_Pose[imu_j], e.para_Ex_Pose[0], e.para_Feature[feature_index].data()});
      }
    }
    int linefeature_index = -1;
    for (auto& it_per_id : e.linefeature) {
      ++linefeature_index;
      problem.AddParameterBlock(e.para_LineFeature[linefeature_index].data(), 4, new LineOrthParameterization());
      int imu_i = it_per_id.start_frame, imu_j = imu_i - 1;
      for (auto& ob : it_per_id.obs) {
        imu_j++;
        Vec4 obs{ob[0], ob[1], ob[2], ob[3]};
        problem.AddResidualBlock(new lineProjectionFactor(obs), loss_function,
                                 {e.para_Pose[imu_j], e.para_Ex_Pose[0], e.para_LineFeature[linefeature_index].data()});
        if (ob[7] == 1.0) {
          Vec3 vp_obs{ob[4], ob[5], ob[6]};
          problem.AddResidualBlock(new vpProjectionFactor(vp_obs), loss_function,
                                   {e.para_Pose[imu_j], e.para_Ex_Pose[0], e.para_LineFeature[linefeature_index].data()});
        }
      }
    }
    SolverOptions options;
    if (g_test_tolerance_scale > 0.0) {   // test hook (oracle_capi.cpp orc_set_tolerance_scale): 0 = ceres' defaults
      options.function_tolerance *= g_test_tolerance_scale;
      options.parameter_tolerance *= g_test_tolerance_scale;
      options.gradient_tolerance *= g_test_tolerance_scale;
    }
    options.max_num_iterations = opt->num_iterations;
    SolverSummary summary;
    // the prior factor and the loss are owned elsewhere (marginalization info / below)
    Solve(options, &problem, &summary);
    report.iterations = (int)summary.iterations.size() - 1;
    report.num_successful_steps = summary.num_successful_steps - 1;
    report.termination = (int)summary.termination_type;
    report.initial_cost = summary.initial_cost;
    report.final_cost = summary.final_cost;
    // detach objects the problem must not free
    for (auto& rb : problem.residuals_) {
      if (rb.loss == loss_function) rb.loss = nullptr;
    }
  }

  e.double2vector2();
  if (opt->remove_line_outliers) {
    std::vector<int> removed_index;
    report.n_lines_removed = e.removeLineOutlier(&removed_index);
    if (w->line_removed) {
      for (int i = 0; i < w->n_lines; ++i) w->line_removed[i] = 0;
      for (int i : removed_index) w->line_removed[i] = 1;
    }
  } else if (w->line_removed) {
    for (int i = 0; i < w->n_lines; ++i) w->line_removed[i] = 0;
  }

  if (opt->marginalization_flag == VPL_MARGIN_OLD) {
    MarginalizationInfo* marginalization_info = new MarginalizationInfo();
    e.vector2double();
    if (e.last_marginalization_info) {
      std::vector<int> drop_set;
      for (int i = 0; i < (int)e.last_marginalization_parameter_blocks.size(); i++)
        if (e.last_marginalization_parameter_blocks[i] == e.para_Pose[0] ||
            e.last_marginalization_parameter_blocks[i] == e.para_SpeedBias[0])
          drop_set.push_back(i);
      MarginalizationFactor* mf = new MarginalizationFactor(e.last_marginalization_info);
      marginalization_info->addResidualBlockInfo(
          new ResidualBlockInfo(mf, nullptr, e.last_marginalization_parameter_blocks, drop_set));
    }
    if (e.pre_integrations[1]->sum_dt < 10.0) {
      marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(
          new IMUFactor(e.pre_integrations[1], e.G), nullptr,
          std::vector<double*>{e.para_Pose[0], e.para_SpeedBias[0], e.para_Pose[1], e.para_SpeedBias[1]},
          std::vector<int>{0, 1}));
    }
    {
      int feature_index = -1;
      for (auto& it_per_id : e.feature) {
        ++feature_index;
        int imu_i = it_per_id.start_frame, imu_j = imu_i - 1;
        if (imu_i != 0) continue;
        Vec3 pts_i = it_per_id.obs[0];
        for (auto& pts_j : it_per_id.obs) {
          imu_j++;
          if (imu_i == imu_j) continue;
          marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(
              new ProjectionFactor(pts_i, pts_j), loss_function,
              std::vector<double*>{e.para_Pose[imu_i], e.para_Pose[imu_j], e.para_Ex_Pose[0], e.para_Feature[feature_index].data()},
              std::vector<int>{0, 3}));
        }
      }
    }
    {
      int linefeature_index = -1;
      for (auto& it_per_id : e.linefeature) {
        ++linefeature_index;
        int imu_i = it_per_id.start_frame, imu_j = imu_i - 1;
        if (imu_i != 0) continue;
        for (auto& ob : it_per_id.obs) {
          imu_j++;
          if (imu_i == imu_j) continue;
          Vec4 obs{ob[0], ob[1], ob[2], ob[3]};
          marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(
              new lineProjectionFactor(obs), loss_function,
              std::vector<double*>{e.para_Pose[imu_j], e.para_Ex_Pose[0], e.para_LineFeature[linefeature_index].data()},
              std::vector<int>{2}));
          // estimator.cpp:1341-1353: the VP factor is added to the already solved ceres
          // problem, NOT to marginalization_info -- it has no effect and is not restated.
        }
      }
    }
    marginalization_info->preMarginalize();
    marginalization_info->marginalize();

    std::map<long, double*> addr_shift;
    for (int i = 1; i <= VPL_WINDOW_SIZE; i++) {
      addr_shift[reinterpret_cast<long>(e.para_Pose[i])] = e.para_Pose[i - 1];
      addr_shift[reinterpret_cast<long>(e.para_SpeedBias[i])] = e.para_SpeedBias[i - 1];
    }
    addr_shift[reinterpret_cast<long>(e.para_Ex_Pose[0])] = e.para_Ex_Pose[0];
    std::vector<double*> parameter_blocks = marginalization_info->getParameterBlocks(addr_shift);
    // the old info is referenced by the MarginalizationFactor just added: the reference
    // deletes it here as well (estimator.cpp:1379-1381); safe because Evaluate already ran.
    if (e.last_marginalization_info) {
      // the new info owns a MarginalizationFactor pointing at the old info; null the pointer use
      delete e.last_marginalization_info;
    }
    e.last_marginalization_info = marginalization_info;
    e.last_marginalization_parameter_blocks = parameter_blocks;
    report.prior_m = marginalization_info->m;
    report.prior_n = marginalization_info->n;
    if (prior_out) export_prior(e, parameter_blocks, prior_out);
    if (A_final_out)
      for (int r = 0; r < marginalization_info->n; ++r)
        for (int c = 0; c < marginalization_info->n; ++c)
          A_final_out[(size_t)r * marginalization_info->n + c] = marginalization_info->A_final(r, c);
    if (b_final_out)
      for (int r = 0; r < marginalization_info->n; ++r) b_final_out[r] = marginalization_info->b_final[r];
  } else if (opt->marginalization_flag == VPL_MARGIN_SECOND_NEW) {
    // estimator.cpp:1385-1447
    bool has = false;
    for (double* b : e.last_marginalization_parameter_blocks)
      if (b == e.para_Pose[VPL_WINDOW_SIZE - 1]) has = true;
    if (e.last_marginalization_info && has) {
      MarginalizationInfo* marginalization_info = new MarginalizationInfo();
      e.vector2double();
      std::vector<int> drop_set;
      for (int i = 0; i < (int)e.last_marginalization_parameter_blocks.size(); i++)
        if (e.last_marginalization_parameter_blocks[i] == e.para_Pose[VPL_WINDOW_SIZE - 1]) drop_set.push_back(i);
      MarginalizationFactor* mf = new MarginalizationFactor(e.last_marginalization_info);
      marginalization_info->addResidualBlockInfo(
          new ResidualBlockInfo(mf, nullptr, e.last_marginalization_parameter_blocks, drop_set));
      marginalization_info->preMarginalize();
      marginalization_info->marginalize();
      std::map<long, double*> addr_shift;
      for (int i = 0; i <= VPL_WINDOW_SIZE; i++) {
        if (i == VPL_WINDOW_SIZE - 1) continue;
        else if (i == VPL_WINDOW_SIZE) {
          addr_shift[reinterpret_cast<long>(e.para_Pose[i])] = e.para_Pose[i - 1];
          addr_shift[reinterpret_cast<long>(e.para_SpeedBias[i])] = e.para_SpeedBias[i - 1];
        } else {
          addr_shift[reinterpret_cast<long>(e.para_Pose[i])] = e.para_Pose[i];
          addr_shift[reinterpret_cast<long>(e.para_SpeedBias[i])] = e.para_SpeedBias[i];
        }
      }
      addr_shift[reinterpret_cast<long>(e.para_Ex_Pose[0])] = e.para_Ex_Pose[0];
      std::vector<double*> parameter_blocks = marginalization_info->getParameterBlocks(addr_shift);
      delete e.last_marginalization_info;
      e.last_marginalization_info = marginalization_info;
      e.last_marginalization_parameter_blocks = parameter_blocks;
      report.prior_m = marginalization_info->m;
      report.prior_n = marginalization_info->n;
      if (prior_out) export_prior(e, parameter_blocks, prior_out);
    } else if (prior_out) {
      // the reference keeps last_marginalization_info / _parameter_blocks as they are (estimator.cpp:1385)
      if (w->has_prior && w->prior) { *prior_out = *w->prior; report.prior_n = w->prior->n; }
      else std::memset(prior_out, 0, sizeof(*prior_out));
    }
  }
  delete loss_function;

  // write the state back in para_* form (what the next vector2double() would produce)
  e.vector2double();
  for (int i = 0; i < VPL_NFRAMES; ++i) {
    for (int k = 0; k < 7; ++k) w->pose[i][k] = e.para_Pose[i][k];
    for (int k = 0; k < 9; ++k) w->speed_bias[i][k] = e.para_SpeedBias[i][k];
  }
  for (int k = 0; k < 7; ++k) w->ex_pose[k] = e.para_Ex_Pose[0][k];
  for (size_t i = 0; i < e.feature.size(); ++i) w->inv_depth[i] = 1.0 / e.feature[i].estimated_depth;
  // erased tracks are gone from e.linefeature: their line_plk keeps the caller's value
  for (size_t i = 0; i < e.linefeature.size(); ++i)
      for (int c = 0; c < 6; ++c) w->line_plk[6 * e.linefeature[i].index + c] = e.linefeature[i].line_plucker[c];
  if (rep) *rep = report;
  return 0;
}

// FeatureManager::triangulateLine (feature_manager.cpp:413-563) on the lines of the window whose line_triangulated flag is
// 0; writes line_plk and the flag of every line it triangulated.  Returns the number of such lines.
int triangulate_lines(vpl_window* w, const vpl_ba_options* opt) {
  Est e;
  load_window(*w, *opt, e, /*all_lines=*/true);
  const int done = e.triangulateLine();
  for (auto& L : e.linefeature) {
    if (!L.is_triangulation) continue;
    for (int c = 0; c < 6; ++c) w->line_plk[6 * L.index + c] = L.line_plucker[c];
    if (w->line_triangulated) w->line_triangulated[L.index] = 1;
  }
  return done;
}

// FeatureManager::triangulate on the tracks of the window whose inv_depth is <= 0 (estimated_depth not set: the reference
// initialises it to -1); writes inv_depth.  Returns the number of tracks triangulated.
int triangulate_points(vpl_window* w, const vpl_ba_options* opt, double init_depth) {
  Est e;
  load_window(*w, *opt, e, true);
  const int done = e.triangulate(init_depth);
  for (size_t i = 0; i < e.feature.size(); ++i)
    if (w->inv_depth[i] < 0.0) w->inv_depth[i] = 1.0 / e.feature[i].estimated_depth;   // the others were not touched
  return done;
}

// Estimator::slideWindow (estimator.cpp:1731-1851) for a full window in the NON_LINEAR state, with the FeatureManager
// calls behind it: MARGIN_OLD -> state swaps + removeBackShiftDepth (feature_manager.cpp:800-874),
// otherwise -> frame 10 copied over frame 9 + removeFront (feature_manager.cpp:915-956).
// The IMU buffers / IntegrationBase objects stay with the caller (they are re-integrated sample by sample, :1786-1797).
int slide_window(vpl_window* w, const vpl_ba_options* opt, int marginalization_flag, double init_depth, vpl_slide_tracks* out) {
  Est e;
  load_window(*w, *opt, e, true);
  constexpr int WINDOW_SIZE = VPL_NFRAMES - 1;
  const int frame_count = WINDOW_SIZE;
  for (int i = 0; i < w->n_points; ++i) { out->point_start[i] = w->point_start[i]; out->point_nobs[i] = 0; out->point_drop[i] = -1; }
  for (int i = 0; i < w->n_lines; ++i) { out->line_start[i] = w->line_start[i]; out->line_nobs[i] = 0; out->line_drop[i] = -1; }
  if (marginalization_flag == VPL_MARGIN_OLD) {
    const Mat3 back_R0 = e.Rs[0];
    const Vec3 back_P0 = e.Ps[0];
    for (int i = 0; i < WINDOW_SIZE; ++i) {
      std::swap(e.Rs[i], e.Rs[i + 1]);
      std::swap(e.Ps[i], e.Ps[i + 1]);
      for (int c = 0; c < 7; ++c) std::swap(w->pose[i][c], w->pose[i + 1][c]);
      for (int c = 0; c < 9; ++c) std::swap(w->speed_bias[i][c], w->speed_bias[i + 1][c]);
    }
    e.Rs[WINDOW_SIZE] = e.Rs[WINDOW_SIZE - 1];
    e.Ps[WINDOW_SIZE] = e.Ps[WINDOW_SIZE - 1];
    for (int c = 0; c < 7; ++c) w->pose[WINDOW_SIZE][c] = w->pose[WINDOW_SIZE - 1][c];
    for (int c = 0; c < 9; ++c) w->speed_bias[WINDOW_SIZE][c] = w->speed_bias[WINDOW_SIZE - 1][c];
    // slideWindowOld (:1828-1851), shift_depth = true
    const Mat3 marg_R = back_R0 * e.ric, new_R = e.Rs[0] * e.ric;
    const Vec3 marg_P = back_P0 + back_R0 * e.tic, new_P = e.Ps[0] + e.Rs[0] * e.tic;
    for (size_t k = 0; k < e.feature.size();) {
      PointTrack& it = e.feature[k];
      if (it.start_frame != 0) { it.start_frame--; ++k; continue; }
      const Vec3 uv_i = it.obs[0];
      it.obs.erase(it.obs.begin());
      out->point_drop[it.index] = 0;
      if (it.obs.size() < 2) { e.feature.erase(e.feature.begin() + k); continue; }
      const Vec3 pts_i = uv_i * it.estimated_depth;
      const Vec3 w_pts_i = marg_R * pts_i + marg_P;
      const Vec3 pts_j = new_R.T() * (w_pts_i - new_P);
      const double dep_j = pts_j[2];
      it.estimated_depth = dep_j > 0 ? dep_j : init_depth;
      ++k;
    }
    for (size_t k = 0; k < e.linefeature.size();) {
      LineTrack& it = e.linefeature[k];
      if (it.start_frame != 0) { it.start_frame--; ++k; continue; }
      it.obs.erase(it.obs.begin());
      out->line_drop[it.index] = 0;
      if (it.obs.size() < 2) { e.linefeature.erase(e.linefeature.begin() + k); continue; }
      const Mat3 Rji = new_R.T() * marg_R;
      const Vec3 tji = new_R.T() * (marg_P - new_P);
      it.line_plucker = plk_to_pose(it.line_plucker, Rji, tji);
      ++k;
    }
  } else {
    for (int c = 0; c < 7; ++c) w->pose[frame_count - 1][c] = w->pose[frame_count][c];
    for (int c = 0; c < 9; ++c) w->speed_bias[frame_count - 1][c] = w->speed_bias[frame_count][c];
    // slideWindowNew -> removeFront(frame_count)
    for (size_t k = 0; k < e.feature.size();) {
      PointTrack& it = e.feature[k];
      if (it.start_frame == frame_count) { it.start_frame--; ++k; continue; }
      const int j = WINDOW_SIZE - 1 - it.start_frame;
      if (it.start_frame + (int)it.obs.size() - 1 < frame_count - 1) { ++k; continue; }
      it.obs.erase(it.obs.begin() + j);
      out->point_drop[it.index] = j;
      if (it.obs.empty()) { e.feature.erase(e.feature.begin() + k); continue; }
      ++k;
    }
    for (size_t k = 0; k < e.linefeature.size();) {
      LineTrack& it = e.linefeature[k];
      if (it.start_frame == frame_count) { it.start_frame--; ++k; continue; }
      const int j = WINDOW_SIZE - 1 - it.start_frame;
      if (it.start_frame + (int)it.obs.size() - 1 < frame_count - 1) { ++k; continue; }
      it.obs.erase(it.obs.begin() + j);
      out->line_drop[it.index] = j;
      if (it.obs.empty()) { e.linefeature.erase(e.linefeature.begin() + k); continue; }
      ++k;
    }
  }
  for (const PointTrack& t : e.feature) {
    out->point_start[t.index] = t.start_frame;
    out->point_nobs[t.index] = (int)t.obs.size();
    if (marginalization_flag == VPL_MARGIN_OLD && out->point_drop[t.index] == 0) w->inv_depth[t.index] = 1.0 / t.estimated_depth;
  }
  for (const LineTrack& t : e.linefeature) {
    out->line_start[t.index] = t.start_frame;
    out->line_nobs[t.index] = (int)t.obs.size();
    if (marginalization_flag == VPL_MARGIN_OLD && out->line_drop[t.index] == 0)
      for (int c = 0; c < 6; ++c) w->line_plk[6 * t.index + c] = t.line_plucker[c];
  }
  return 0;
}

// Estimator::onlyLineOpt (estimator.cpp:950-1039): poses and extrinsic constant, line factors only (no VP factors),
// CauchyLoss(1.0), ceres defaults otherwise (LEVENBERG_MARQUARDT, DENSE_SCHUR), NUM_ITERATIONS; then double2vector() and
// removeLineOutlier.  With fewer than four line tracks the function returns before the solve (:1019-1022).
int only_line_opt(vpl_window* w, const vpl_ba_options* opt, vpl_solve_report* rep) {
  Est e;
  load_window(*w, *opt, e);
  lineProjectionFactor::sqrt_info = opt->line_factor;
  vpl_solve_report report;
  std::memset(&report, 0, sizeof(report));
  if (w->line_removed)
    for (int i = 0; i < w->n_lines; ++i) w->line_removed[i] = 0;
  e.vector2double();
  const int feature_index = (int)e.linefeature.size() - 1;
  if (feature_index >= 3) {
    LossFunction* loss_function = new CauchyLoss(1.0);
    Problem problem;
    for (int i = 0; i < VPL_WINDOW_SIZE + 1; i++) {
      problem.AddParameterBlock(e.para_Pose[i], 7, new PoseLocalParameterization());
      problem.SetParameterBlockConstant(e.para_Pose[i]);
    }
    problem.AddParameterBlock(e.para_Ex_Pose[0], 7, new PoseLocalParameterization());
    problem.SetParameterBlockConstant(e.para_Ex_Pose[0]);
    for (size_t li = 0; li < e.linefeature.size(); ++li) {
      problem.AddParameterBlock(e.para_LineFeature[li].data(), 4, new LineOrthParameterization());
      int imu_j = e.linefeature[li].start_frame - 1;
      for (auto& ob : e.linefeature[li].obs) {
        imu_j++;
        Vec4 obs{ob[0], ob[1], ob[2], ob[3]};
        problem.AddResidualBlock(new lineProjectionFactor(obs), loss_function,
                                 {e.para_Pose[imu_j], e.para_Ex_Pose[0], e.para_LineFeature[li].data()});
      }
    }
    SolverOptions options;
    options.use_dogleg = false;
    options.max_num_iterations = opt->num_iterations;
    SolverSummary summary;
    Solve(options, &problem, &summary);
    report.iterations = (int)summary.iterations.size() - 1;
    report.num_successful_steps = summary.num_successful_steps - 1;
    report.termination = (int)summary.termination_type;
    report.initial_cost = summary.initial_cost;
    report.final_cost = summary.final_cost;
    for (auto& rb : problem.residuals_)
      if (rb.loss == loss_function) rb.loss = nullptr;
    delete loss_function;
    e.double2vector2(/*line_gauge=*/false);
    std::vector<int> removed_index;
    report.n_lines_removed = e.removeLineOutlier(&removed_index);
    if (w->line_removed)
      for (int i : removed_index) w->line_removed[i] = 1;
    for (auto& L : e.linefeature)
      for (int c = 0; c < 6; ++c) w->line_plk[6 * L.index + c] = L.line_plucker[c];
  }
  if (rep) *rep = report;
  return 0;
}

}  // namespace orc
