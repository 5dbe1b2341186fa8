// TEST INFRASTRUCTURE -- CPU oracle (see smallmat.h / factors.h for the reference map).
#include "factors.h"

namespace orc {

double ProjectionFactor::sqrt_info = 460.0 / 1.5;
double lineProjectionFactor::sqrt_info = 460.0 / 1.5;
double vpProjectionFactor::sqrt_info = 10.0;

static inline Vec3 v3(const double* p) { return Vec3{p[0], p[1], p[2]}; }
static inline Quat q4(const double* p) { return Quat(p[6], p[3], p[4], p[5]); }  // pose block: qx,qy,qz,qw at 3..6

// ---- parameterisations ------------------------------------------------------
// pose_local_parameterization.cpp:3-19
bool PoseLocalParameterization::Plus(const double* x, const double* delta, double* xpd) const {
  Vec3 p = v3(x);
  Quat q(x[6], x[3], x[4], x[5]);
  Vec3 dp = v3(delta);
  Quat dq = deltaQ(v3(delta + 3));
  Vec3 pn = p + dp;
  Quat qn = (q * dq).normalized();
  xpd[0] = pn[0]; xpd[1] = pn[1]; xpd[2] = pn[2];
  xpd[3] = qn.x; xpd[4] = qn.y; xpd[5] = qn.z; xpd[6] = qn.w;
  return true;
}
// pose_local_parameterization.cpp:20-27
bool PoseLocalParameterization::ComputeJacobian(const double*, double* j) const {
  for (int i = 0; i < 42; ++i) j[i] = 0.0;
  for (int i = 0; i < 6; ++i) j[i * 6 + i] = 1.0;
  return true;
}
// line_parameterization.cpp:7-63
bool LineOrthParameterization::Plus(const double* x, const double* delta, double* xpd) const {
  Mat3 R = orth_R(x[0], x[1], x[2]);
  double phi = x[3];
  double w1 = std::cos(phi), w2 = std::sin(phi);
  double d0 = delta[0], d1 = delta[1], d2 = delta[2], dphi = delta[3];
  Mat3 Rz{std::cos(d2), -std::sin(d2), 0, std::sin(d2), std::cos(d2), 0, 0, 0, 1};
  Mat3 Ry{std::cos(d1), 0., std::sin(d1), 0., 1., 0., -std::sin(d1), 0., std::cos(d1)};
  Mat3 Rx{1., 0., 0., 0., std::cos(d0), -std::sin(d0), 0., std::sin(d0), std::cos(d0)};
  R = R * Rx * Ry * Rz;
  Mat<2, 2> W{w1, -w2, w2, w1};
  Mat<2, 2> dW{std::cos(dphi), -std::sin(dphi), std::sin(dphi), std::cos(dphi)};
  W = W * dW;
  // u1 = R.col(0), u2 = R.col(1), u3 = R.col(2)
  xpd[0] = std::atan2(R(2, 1), R(2, 2));
  xpd[1] = std::asin(-R(2, 0));
  xpd[2] = std::atan2(R(1, 0), R(0, 0));
  xpd[3] = std::asin(W(1, 0));
  return true;
}
// line_parameterization.cpp:94-100
bool LineOrthParameterization::ComputeJacobian(const double*, double* j) const {
  for (int i = 0; i < 16; ++i) j[i] = 0.0;
  for (int i = 0; i < 4; ++i) j[i * 4 + i] = 1.0;
  return true;
}

// ---- ProjectionFactor (UNIT_SPHERE_ERROR is defined, parameters.h:27) -------
ProjectionFactor::ProjectionFactor(const Vec3& pi, const Vec3& pj) : pts_i(pi), pts_j(pj) {
  sizes_ = {7, 7, 7, 1};
  nres_ = 2;
  // projection_factor.cpp:9-18
  Vec3 a = pts_j.normalized();
  Vec3 tmp{0, 0, 1};
  if (a[0] == tmp[0] && a[1] == tmp[1] && a[2] == tmp[2]) tmp = Vec3{1, 0, 0};
  Vec3 b1 = (tmp - a * a.dot(tmp)).normalized();
  Vec3 b2 = cross(a, b1);
  for (int c = 0; c < 3; ++c) { tangent_base(0, c) = b1[c]; tangent_base(1, c) = b2[c]; }
}

bool ProjectionFactor::Evaluate(double const* const* P, double* residuals, double** jacobians) const {
  Vec3 Pi = v3(P[0]); Quat Qi = q4(P[0]);
  Vec3 Pj = v3(P[1]); Quat Qj = q4(P[1]);
  Vec3 tic = v3(P[2]); Quat qic = q4(P[2]);
  double inv_dep_i = P[3][0];

  Vec3 pts_camera_i = pts_i / inv_dep_i;
  Vec3 pts_imu_i = qic.rotate(pts_camera_i) + tic;
  Vec3 pts_w = Qi.rotate(pts_imu_i) + Pi;
  Vec3 pts_imu_j = Qj.inverse().rotate(pts_w - Pj);
  Vec3 pts_camera_j = qic.inverse().rotate(pts_imu_j - tic);

  Vec2 residual = tangent_base * (pts_camera_j.normalized() - pts_j.normalized());
  residual = residual * sqrt_info;
  residuals[0] = residual[0];
  residuals[1] = residual[1];

  if (jacobians) {
    Mat3 Ri = Qi.toRotationMatrix();
    Mat3 Rj = Qj.toRotationMatrix();
    Mat3 ric = qic.toRotationMatrix();
    double norm = pts_camera_j.norm();
    double n3 = std::pow(norm, 3);
    double x1 = pts_camera_j[0], x2 = pts_camera_j[1], x3 = pts_camera_j[2];
    Mat3 norm_jaco{1.0 / norm - x1 * x1 / n3, -x1 * x2 / n3,             -x1 * x3 / n3,
                   -x1 * x2 / n3,             1.0 / norm - x2 * x2 / n3, -x2 * x3 / n3,
                   -x1 * x3 / n3,             -x2 * x3 / n3,             1.0 / norm - x3 * x3 / n3};
    Mat<2, 3> reduce = tangent_base * norm_jaco;
    reduce = reduce * sqrt_info;

    if (jacobians[0]) {
      Mat<3, 6> jaco_i;
      jaco_i.setBlock<3, 3>(0, 0, ric.T() * Rj.T());
      jaco_i.setBlock<3, 3>(0, 3, ric.T() * Rj.T() * Ri * (-skew(pts_imu_i)));
      Mat<2, 6> J = reduce * jaco_i;
      for (int r = 0; r < 2; ++r) {
        for (int c = 0; c < 6; ++c) jacobians[0][r * 7 + c] = J(r, c);
        jacobians[0][r * 7 + 6] = 0.0;
      }
    }
    if (jacobians[1]) {
      Mat<3, 6> jaco_j;
      jaco_j.setBlock<3, 3>(0, 0, ric.T() * (-Rj.T()));
      jaco_j.setBlock<3, 3>(0, 3, ric.T() * skew(pts_imu_j));
      Mat<2, 6> J = reduce * jaco_j;
      for (int r = 0; r < 2; ++r) {
        for (int c = 0; c < 6; ++c) jacobians[1][r * 7 + c] = J(r, c);
        jacobians[1][r * 7 + 6] = 0.0;
      }
    }
    if (jacobians[2]) {
      Mat<3, 6> jaco_ex;
      jaco_ex.setBlock<3, 3>(0, 0, ric.T() * (Rj.T() * Ri - Mat3::Identity()));
      Mat3 tmp_r = ric.T() * Rj.T() * Ri * ric;
      Mat3 right = -(tmp_r * skew(pts_camera_i)) + skew(tmp_r * pts_camera_i) +
                   skew(ric.T() * (Rj.T() * (Ri * tic + Pi - Pj) - tic));
      jaco_ex.setBlock<3, 3>(0, 3, right);
      Mat<2, 6> J = reduce * jaco_ex;
      for (int r = 0; r < 2; ++r) {
        for (int c = 0; c < 6; ++c) jacobians[2][r * 7 + c] = J(r, c);
        jacobians[2][r * 7 + 6] = 0.0;
      }
    }
    if (jacobians[3]) {
      Vec2 J = reduce * ric.T() * Rj.T() * Ri * ric * pts_i * (-1.0 / (inv_dep_i * inv_dep_i));
      jacobians[3][0] = J[0];
      jacobians[3][1] = J[1];
    }
  }
  return true;
}

// ---- shared chain of the line and VP factors --------------------------------
namespace {
struct LineChain {
  Vec6 line_w, line_b, line_c;
  Mat3 Rwb, Rbc;
  Vec3 twb, tbc;
};
LineChain line_chain(double const* const* P) {
  LineChain c;
  c.twb = v3(P[0]);
  c.Rwb = q4(P[0]).toRotationMatrix();
  c.tbc = v3(P[1]);
  c.Rbc = q4(P[1]).toRotationMatrix();
  Vec4 orth{P[2][0], P[2][1], P[2][2], P[2][3]};
  c.line_w = orth_to_plk(orth);
  c.line_b = plk_from_pose(c.line_w, c.Rwb, c.twb);
  c.line_c = plk_from_pose(c.line_b, c.Rbc, c.tbc);
  return c;
}
// the three Jacobian blocks common to line_projection_factor.cpp:75-151 and :304-378,
// given jaco_e_Lc (2x6)
void line_jacobians(const LineChain& c, const Mat<2, 6>& jaco_e_Lc, double** jacobians) {
  if (jacobians[0]) {
    Mat6 invTbc;
    invTbc.setBlock<3, 3>(0, 0, c.Rbc.T());
    invTbc.setBlock<3, 3>(0, 3, -(c.Rbc.T() * skew(c.tbc)));
    invTbc.setBlock<3, 3>(3, 3, c.Rbc.T());
    Vec3 nw{c.line_w[0], c.line_w[1], c.line_w[2]}, dw{c.line_w[3], c.line_w[4], c.line_w[5]};
    Mat6 jaco_Lc_pose;
    jaco_Lc_pose.setBlock<3, 3>(0, 0, c.Rwb.T() * skew(dw));
    jaco_Lc_pose.setBlock<3, 3>(0, 3, skew(c.Rwb.T() * (nw + skew(dw) * c.twb)));
    jaco_Lc_pose.setBlock<3, 3>(3, 3, skew(c.Rwb.T() * dw));
    jaco_Lc_pose = invTbc * jaco_Lc_pose;
    Mat<2, 6> J = jaco_e_Lc * jaco_Lc_pose;
    for (int r = 0; r < 2; ++r) {
      for (int k = 0; k < 6; ++k) jacobians[0][r * 7 + k] = J(r, k);
      jacobians[0][r * 7 + 6] = 0.0;
    }
  }
  if (jacobians[1]) {
    Vec3 nb{c.line_b[0], c.line_b[1], c.line_b[2]}, db{c.line_b[3], c.line_b[4], c.line_b[5]};
    Mat6 jaco_Lc_ex;
    jaco_Lc_ex.setBlock<3, 3>(0, 0, c.Rbc.T() * skew(db));
    jaco_Lc_ex.setBlock<3, 3>(0, 3, skew(c.Rbc.T() * (nb + skew(db) * c.tbc)));
    jaco_Lc_ex.setBlock<3, 3>(3, 3, skew(c.Rbc.T() * db));
    Mat<2, 6> J = jaco_e_Lc * jaco_Lc_ex;
    for (int r = 0; r < 2; ++r) {
      for (int k = 0; k < 6; ++k) jacobians[1][r * 7 + k] = J(r, k);
      jacobians[1][r * 7 + 6] = 0.0;
    }
  }
  if (jacobians[2]) {
    Mat3 Rwc = c.Rwb * c.Rbc;
    Vec3 twc = c.Rwb * c.tbc + c.twb;
    Mat6 invTwc;
    invTwc.setBlock<3, 3>(0, 0, Rwc.T());
    invTwc.setBlock<3, 3>(0, 3, -(Rwc.T() * skew(twc)));
    invTwc.setBlock<3, 3>(3, 3, Rwc.T());
    Vec3 nw{c.line_w[0], c.line_w[1], c.line_w[2]}, vw{c.line_w[3], c.line_w[4], c.line_w[5]};
    Vec3 u1 = nw / nw.norm();
    Vec3 u2 = vw / vw.norm();
    Vec3 u3 = cross(u1, u2);
    Vec2 w{nw.norm(), vw.norm()};
    w = w / w.norm();
    Mat<6, 4> K;
    for (int i = 0; i < 3; ++i) {
      K(3 + i, 0) = w[1] * u3[i];
      K(i, 1) = -w[0] * u3[i];
      K(i, 2) = w[0] * u2[i];
      K(3 + i, 2) = -w[1] * u1[i];
      K(i, 3) = -w[1] * u1[i];
      K(3 + i, 3) = w[0] * u2[i];
    }
    Mat<2, 4> J = jaco_e_Lc * invTwc * K;
    for (int r = 0; r < 2; ++r)
      for (int k = 0; k < 4; ++k) jacobians[2][r * 4 + k] = J(r, k);
  }
}
}  // namespace

lineProjectionFactor::lineProjectionFactor(const Vec4& obs) : obs_i(obs) {
  sizes_ = {7, 7, 4};
  nres_ = 2;
}

// line_projection_factor.cpp:251-380
bool lineProjectionFactor::Evaluate(double const* const* P, double* residuals, double** jacobians) const {
  LineChain c = line_chain(P);
  Vec3 nc{c.line_c[0], c.line_c[1], c.line_c[2]};
  double l_norm = nc[0] * nc[0] + nc[1] * nc[1];
  double l_sqrtnorm = std::sqrt(l_norm);
  double l_trinorm = l_norm * l_sqrtnorm;
  double e1 = obs_i[0] * nc[0] + obs_i[1] * nc[1] + nc[2];
  double e2 = obs_i[2] * nc[0] + obs_i[3] * nc[1] + nc[2];
  residuals[0] = sqrt_info * (e1 / l_sqrtnorm);
  residuals[1] = sqrt_info * (e2 / l_sqrtnorm);
  if (jacobians) {
    Mat<2, 3> jaco_e_l{obs_i[0] / l_sqrtnorm - nc[0] * e1 / l_trinorm, obs_i[1] / l_sqrtnorm - nc[1] * e1 / l_trinorm, 1.0 / l_sqrtnorm,
                       obs_i[2] / l_sqrtnorm - nc[0] * e2 / l_trinorm, obs_i[3] / l_sqrtnorm - nc[1] * e2 / l_trinorm, 1.0 / l_sqrtnorm};
    jaco_e_l = jaco_e_l * sqrt_info;
    Mat<2, 6> jaco_e_Lc;
    jaco_e_Lc.setBlock<2, 3>(0, 0, jaco_e_l);  // jaco_l_Lc = [I3 0]
    line_jacobians(c, jaco_e_Lc, jacobians);
  }
  return true;
}

vpProjectionFactor::vpProjectionFactor(const Vec3& vp) : obs_i(vp) {
  sizes_ = {7, 7, 4};
  nres_ = 2;
}

// line_projection_factor.cpp:11-153.  jaco_e_l is the reference's literal matrix
// (derivative w.r.t. the OBSERVATION, :62-64) -- kept as is.
bool vpProjectionFactor::Evaluate(double const* const* P, double* residuals, double** jacobians) const {
  LineChain c = line_chain(P);
  Vec3 d_c{c.line_c[3], c.line_c[4], c.line_c[5]};
  Vec2 d_c_2d{d_c[0] / d_c[2], d_c[1] / d_c[2]};
  Vec2 vp_2d{obs_i[0] / obs_i[2], obs_i[1] / obs_i[2]};
  double v1v3_inv = vp_2d[0], v2v3_inv = vp_2d[1];
  residuals[0] = sqrt_info * (d_c_2d[0] - vp_2d[0]);
  residuals[1] = sqrt_info * (d_c_2d[1] - vp_2d[1]);
  if (jacobians) {
    Mat<2, 3> jaco_e_l{-1 / obs_i[2], 0.0, v1v3_inv, 0.0, -1 / obs_i[2], v2v3_inv};
    jaco_e_l = jaco_e_l * sqrt_info;
    Mat<2, 6> jaco_e_Lc;
    jaco_e_Lc.setBlock<2, 3>(0, 3, jaco_e_l);  // jaco_l_Lc = [0 I3]
    line_jacobians(c, jaco_e_Lc, jacobians);
  }
  return true;
}

// ---- IntegrationBase ---------------------------------------------------------
IntegrationBase::IntegrationBase(const Vec3& a0, const Vec3& g0, const Vec3& ba, const Vec3& bg, const ImuNoise& nz)
    : acc_0(a0), gyr_0(g0), linearized_ba(ba), linearized_bg(bg) {
  jacobian = Mat<15, 15>::Identity();
  covariance = Mat<15, 15>::Zero();
  for (int i = 0; i < 3; ++i) {
    noise(i, i) = nz.acc_n * nz.acc_n;
    noise(3 + i, 3 + i) = nz.gyr_n * nz.gyr_n;
    noise(6 + i, 6 + i) = nz.acc_n * nz.acc_n;
    noise(9 + i, 9 + i) = nz.gyr_n * nz.gyr_n;
    noise(12 + i, 12 + i) = nz.acc_w * nz.acc_w;
    noise(15 + i, 15 + i) = nz.gyr_w * nz.gyr_w;
  }
}

void IntegrationBase::push_back(double dt_, const Vec3& acc, const Vec3& gyr) { propagate(dt_, acc, gyr); }

// integration_base.h:54-168
void IntegrationBase::midPointIntegration(double _dt, const Vec3& _acc_0, const Vec3& _gyr_0, const Vec3& _acc_1,
                                          const Vec3& _gyr_1, const Vec3& dp, const Quat& dq, const Vec3& dv,
                                          const Vec3& lba, const Vec3& lbg, Vec3& res_p, Quat& res_q, Vec3& res_v,
                                          bool update_jacobian) {
  Vec3 un_acc_0 = dq.rotate(_acc_0 - lba);
  Vec3 un_gyr = (_gyr_0 + _gyr_1) * 0.5 - lbg;
  res_q = dq * Quat(1, un_gyr[0] * _dt / 2, un_gyr[1] * _dt / 2, un_gyr[2] * _dt / 2);
  Vec3 un_acc_1 = res_q.rotate(_acc_1 - lba);
  Vec3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
  res_p = dp + dv * _dt + un_acc * (0.5 * _dt * _dt);
  res_v = dv + un_acc * _dt;

  if (update_jacobian) {
    Vec3 w_x = (_gyr_0 + _gyr_1) * 0.5 - lbg;
    Vec3 a_0_x = _acc_0 - lba;
    Vec3 a_1_x = _acc_1 - lba;
    Mat3 R_w_x = skew(w_x), R_a_0_x = skew(a_0_x), R_a_1_x = skew(a_1_x);
    Mat3 I3 = Mat3::Identity();
    Mat3 Rq = dq.toRotationMatrix();
    Mat3 Rr = res_q.toRotationMatrix();

    Mat<15, 15> F;
    F.setBlock<3, 3>(0, 0, I3);
    F.setBlock<3, 3>(0, 3, Rq * R_a_0_x * (-0.25 * _dt * _dt) + Rr * R_a_1_x * (I3 - R_w_x * _dt) * (-0.25 * _dt * _dt));
    F.setBlock<3, 3>(0, 6, I3 * _dt);
    F.setBlock<3, 3>(0, 9, (Rq + Rr) * (-0.25 * _dt * _dt));
    F.setBlock<3, 3>(0, 12, Rr * R_a_1_x * (-0.25 * _dt * _dt * -_dt));
    F.setBlock<3, 3>(3, 3, I3 - R_w_x * _dt);
    F.setBlock<3, 3>(3, 12, I3 * (-1.0 * _dt));
    F.setBlock<3, 3>(6, 3, Rq * R_a_0_x * (-0.5 * _dt) + Rr * R_a_1_x * (I3 - R_w_x * _dt) * (-0.5 * _dt));
    F.setBlock<3, 3>(6, 6, I3);
    F.setBlock<3, 3>(6, 9, (Rq + Rr) * (-0.5 * _dt));
    F.setBlock<3, 3>(6, 12, Rr * R_a_1_x * (-0.5 * _dt * -_dt));
    F.setBlock<3, 3>(9, 9, I3);
    F.setBlock<3, 3>(12, 12, I3);

    Mat<15, 18> V;
    Mat3 V03 = (-Rr) * R_a_1_x * (0.25 * _dt * _dt * 0.5 * _dt);
    Mat3 V63 = (-Rr) * R_a_1_x * (0.5 * _dt * 0.5 * _dt);
    V.setBlock<3, 3>(0, 0, Rq * (0.25 * _dt * _dt));
    V.setBlock<3, 3>(0, 3, V03);
    V.setBlock<3, 3>(0, 6, Rr * (0.25 * _dt * _dt));
    V.setBlock<3, 3>(0, 9, V03);
    V.setBlock<3, 3>(3, 3, I3 * (0.5 * _dt));
    V.setBlock<3, 3>(3, 9, I3 * (0.5 * _dt));
    V.setBlock<3, 3>(6, 0, Rq * (0.5 * _dt));
    V.setBlock<3, 3>(6, 3, V63);
    V.setBlock<3, 3>(6, 6, Rr * (0.5 * _dt));
    V.setBlock<3, 3>(6, 9, V63);
    V.setBlock<3, 3>(9, 12, I3 * _dt);
    V.setBlock<3, 3>(12, 15, I3 * _dt);

    jacobian = F * jacobian;
    covariance = F * covariance * F.T() + V * noise * V.T();
  }
}

// integration_base.h:170-198
void IntegrationBase::propagate(double _dt, const Vec3& _acc_1, const Vec3& _gyr_1) {
  dt = _dt;
  acc_1 = _acc_1;
  gyr_1 = _gyr_1;
  Vec3 rp, rv;
  Quat rq;
  midPointIntegration(_dt, acc_0, gyr_0, _acc_1, _gyr_1, delta_p, delta_q, delta_v, linearized_ba, linearized_bg,
                      rp, rq, rv, true);
  delta_p = rp;
  delta_q = rq;
  delta_v = rv;
  delta_q.normalize();
  sum_dt += dt;
  acc_0 = acc_1;
  gyr_0 = gyr_1;
}

// integration_base.h:200-226
Mat<15, 1> IntegrationBase::evaluate(const Vec3& Pi, const Quat& Qi, const Vec3& Vi, const Vec3& Bai,
                                     const Vec3& Bgi, const Vec3& Pj, const Quat& Qj, const Vec3& Vj,
                                     const Vec3& Baj, const Vec3& Bgj, const Vec3& G) const {
  Mat<15, 1> residuals;
  Mat3 dp_dba = jacobian.block<3, 3>(0, 9);
  Mat3 dp_dbg = jacobian.block<3, 3>(0, 12);
  Mat3 dq_dbg = jacobian.block<3, 3>(3, 12);
  Mat3 dv_dba = jacobian.block<3, 3>(6, 9);
  Mat3 dv_dbg = jacobian.block<3, 3>(6, 12);
  Vec3 dba = Bai - linearized_ba;
  Vec3 dbg = Bgi - linearized_bg;
  Quat corrected_delta_q = delta_q * deltaQ(dq_dbg * dbg);
  Vec3 corrected_delta_v = delta_v + dv_dba * dba + dv_dbg * dbg;
  Vec3 corrected_delta_p = delta_p + dp_dba * dba + dp_dbg * dbg;
  Vec3 rp = Qi.inverse().rotate(G * (0.5 * sum_dt * sum_dt) + Pj - Pi - Vi * sum_dt) - corrected_delta_p;
  Vec3 rq = (corrected_delta_q.inverse() * (Qi.inverse() * Qj)).vec() * 2.0;
  Vec3 rv = Qi.inverse().rotate(G * sum_dt + Vj - Vi) - corrected_delta_v;
  Vec3 rba = Baj - Bai;
  Vec3 rbg = Bgj - Bgi;
  for (int i = 0; i < 3; ++i) {
    residuals[i] = rp[i];
    residuals[3 + i] = rq[i];
    residuals[6 + i] = rv[i];
    residuals[9 + i] = rba[i];
    residuals[12 + i] = rbg[i];
  }
  return residuals;
}

// ---- IMUFactor ---------------------------------------------------------------
IMUFactor::IMUFactor(const IntegrationBase* pre, const Vec3& G_) : pre_integration(pre), G(G_) {
  sizes_ = {7, 9, 7, 9};
  nres_ = 15;
  // imu_factor.h:68  LLT(covariance.inverse()).matrixL().transpose()
  MatX C(15, 15), Ci;
  for (int i = 0; i < 15; ++i)
    for (int j = 0; j < 15; ++j) C(i, j) = pre->covariance(i, j);
  inverse_lu(C, Ci);
  // LLT reads the lower triangle
  for (int i = 0; i < 15; ++i)
    for (int j = i + 1; j < 15; ++j) Ci(i, j) = Ci(j, i);
  cholesky_lower(Ci);
  for (int i = 0; i < 15; ++i)
    for (int j = 0; j < 15; ++j) sqrt_info(i, j) = Ci(j, i);  // L^T
}

bool IMUFactor::Evaluate(double const* const* P, double* residuals, double** jacobians) const {
  Vec3 Pi = v3(P[0]); Quat Qi = q4(P[0]);
  Vec3 Vi = v3(P[1]), Bai = v3(P[1] + 3), Bgi = v3(P[1] + 6);
  Vec3 Pj = v3(P[2]); Quat Qj = q4(P[2]);
  Vec3 Vj = v3(P[3]), Baj = v3(P[3] + 3), Bgj = v3(P[3] + 6);
  const IntegrationBase* pre = pre_integration;

  Mat<15, 1> residual = pre->evaluate(Pi, Qi, Vi, Bai, Bgi, Pj, Qj, Vj, Baj, Bgj, G);
  residual = sqrt_info * residual;
  for (int i = 0; i < 15; ++i) residuals[i] = residual[i];

  if (jacobians) {
    double sum_dt = pre->sum_dt;
    Mat3 dp_dba = pre->jacobian.block<3, 3>(0, 9);
    Mat3 dp_dbg = pre->jacobian.block<3, 3>(0, 12);
    Mat3 dq_dbg = pre->jacobian.block<3, 3>(3, 12);
    Mat3 dv_dba = pre->jacobian.block<3, 3>(6, 9);
    Mat3 dv_dbg = pre->jacobian.block<3, 3>(6, 12);
    Mat3 RiT = Qi.inverse().toRotationMatrix();
    Quat corrected_delta_q = pre->delta_q * deltaQ(dq_dbg * (Bgi - pre->linearized_bg));

    if (jacobians[0]) {
      Mat<15, 7> J;
      J.setBlock<3, 3>(0, 0, -RiT);
      J.setBlock<3, 3>(0, 3, skew(Qi.inverse().rotate(G * (0.5 * sum_dt * sum_dt) + Pj - Pi - Vi * sum_dt)));
      Mat<4, 4> M = Qleft(Qj.inverse() * Qi) * Qright(corrected_delta_q);
      J.setBlock<3, 3>(3, 3, -(M.block<3, 3>(1, 1)));
      J.setBlock<3, 3>(6, 3, skew(Qi.inverse().rotate(G * sum_dt + Vj - Vi)));
      J = sqrt_info * J;
      for (int i = 0; i < 105; ++i) jacobians[0][i] = J.a[i];
    }
    if (jacobians[1]) {
      Mat<15, 9> J;
      J.setBlock<3, 3>(0, 0, -RiT * sum_dt);
      J.setBlock<3, 3>(0, 3, -dp_dba);
      J.setBlock<3, 3>(0, 6, -dp_dbg);
      Mat<4, 4> M = Qleft(Qj.inverse() * Qi * corrected_delta_q);
      J.setBlock<3, 3>(3, 6, -(M.block<3, 3>(1, 1)) * dq_dbg);
      J.setBlock<3, 3>(6, 0, -RiT);
      J.setBlock<3, 3>(6, 3, -dv_dba);
      J.setBlock<3, 3>(6, 6, -dv_dbg);
      J.setBlock<3, 3>(9, 3, -Mat3::Identity());
      J.setBlock<3, 3>(12, 6, -Mat3::Identity());
      J = sqrt_info * J;
      for (int i = 0; i < 135; ++i) jacobians[1][i] = J.a[i];
    }
    if (jacobians[2]) {
      Mat<15, 7> J;
      J.setBlock<3, 3>(0, 0, RiT);
      Mat<4, 4> M = Qleft(corrected_delta_q.inverse() * Qi.inverse() * Qj);
      J.setBlock<3, 3>(3, 3, M.block<3, 3>(1, 1));
      J = sqrt_info * J;
      for (int i = 0; i < 105; ++i) jacobians[2][i] = J.a[i];
    }
    if (jacobians[3]) {
      Mat<15, 9> J;
      J.setBlock<3, 3>(6, 0, RiT);
      J.setBlock<3, 3>(9, 3, Mat3::Identity());
      J.setBlock<3, 3>(12, 6, Mat3::Identity());
      J = sqrt_info * J;
      for (int i = 0; i < 135; ++i) jacobians[3][i] = J.a[i];
    }
  }
  return true;
}

}  // namespace orc
