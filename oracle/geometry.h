// TEST INFRASTRUCTURE -- CPU oracle (see smallmat.h).
// Restates vins_estimator/src/utility/utility.h:12-108 (quaternion helpers) and
// vins_estimator/src/utility/line_geometry.cpp:62-126,155-216 (Pluecker geometry).
#pragma once
#include "smallmat.h"

namespace orc {

// Utility::deltaQ (utility.h:12-24): first-order, NOT normalised.
inline Quat deltaQ(const Vec3& theta) { return Quat(1.0, theta[0] / 2.0, theta[1] / 2.0, theta[2] / 2.0); }

// Utility::skewSymmetric (utility.h:27-34) == skew_symmetric (line_geometry.cpp:155-159)
inline Mat3 skew(const Vec3& q) {
  Mat3 S;
  S(0, 0) = 0;     S(0, 1) = -q[2]; S(0, 2) = q[1];
  S(1, 0) = q[2];  S(1, 1) = 0;     S(1, 2) = -q[0];
  S(2, 0) = -q[1]; S(2, 1) = q[0];  S(2, 2) = 0;
  return S;
}

// Utility::Qleft / Qright (utility.h:46-64); positify is the identity (utility.h:37-44).
inline Mat<4, 4> Qleft(const Quat& q) {
  Mat<4, 4> ans;
  Vec3 v = q.vec();
  ans(0, 0) = q.w;
  for (int i = 0; i < 3; ++i) { ans(0, 1 + i) = -v[i]; ans(1 + i, 0) = v[i]; }
  Mat3 B = Mat3::Identity() * q.w + skew(v);
  ans.setBlock<3, 3>(1, 1, B);
  return ans;
}
inline Mat<4, 4> Qright(const Quat& p) {
  Mat<4, 4> ans;
  Vec3 v = p.vec();
  ans(0, 0) = p.w;
  for (int i = 0; i < 3; ++i) { ans(0, 1 + i) = -v[i]; ans(1 + i, 0) = v[i]; }
  Mat3 B = Mat3::Identity() * p.w - skew(v);
  ans.setBlock<3, 3>(1, 1, B);
  return ans;
}

// Utility::R2ypr (utility.h:66-81): degrees.
inline Vec3 R2ypr(const Mat3& R) {
  Vec3 n{R(0, 0), R(1, 0), R(2, 0)};
  Vec3 o{R(0, 1), R(1, 1), R(2, 1)};
  Vec3 a{R(0, 2), R(1, 2), R(2, 2)};
  double y = std::atan2(n[1], n[0]);
  double p = std::atan2(-n[2], n[0] * std::cos(y) + n[1] * std::sin(y));
  double r = std::atan2(a[0] * std::sin(y) - a[1] * std::cos(y), -o[0] * std::sin(y) + o[1] * std::cos(y));
  return Vec3{y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0};
}

// Utility::ypr2R (utility.h:83-108): degrees in, Rz*Ry*Rx.
inline Mat3 ypr2R(const Vec3& ypr) {
  double y = ypr[0] / 180.0 * M_PI, p = ypr[1] / 180.0 * M_PI, r = ypr[2] / 180.0 * M_PI;
  Mat3 Rz{std::cos(y), -std::sin(y), 0, std::sin(y), std::cos(y), 0, 0, 0, 1};
  Mat3 Ry{std::cos(p), 0., std::sin(p), 0., 1., 0., -std::sin(p), 0., std::cos(p)};
  Mat3 Rx{1., 0., 0., 0., std::cos(r), -std::sin(r), 0., std::sin(r), std::cos(r)};
  return Rz * Ry * Rx;
}

// Rotation matrix from the 3 orthonormal angles, as written out at line_geometry.cpp:100-104.
inline Mat3 orth_R(double t0, double t1, double t2) {
  double s1 = std::sin(t0), c1 = std::cos(t0);
  double s2 = std::sin(t1), c2 = std::cos(t1);
  double s3 = std::sin(t2), c3 = std::cos(t2);
  return Mat3{c2 * c3, s1 * s2 * c3 - c1 * s3, c1 * s2 * c3 + s1 * s3,
              c2 * s3, s1 * s2 * s3 + c1 * c3, c1 * s2 * s3 - s1 * c3,
              -s2,     s1 * c2,                c1 * c2};
}

// plk_to_orth (line_geometry.cpp:62-83)
inline Vec4 plk_to_orth(const Vec6& plk) {
  Vec3 n{plk[0], plk[1], plk[2]}, v{plk[3], plk[4], plk[5]};
  Vec3 u1 = n / n.norm();
  Vec3 u2 = v / v.norm();
  Vec3 u3 = cross(u1, u2);
  Vec4 orth;
  orth[0] = std::atan2(u2[2], u3[2]);
  orth[1] = std::asin(-u1[2]);
  orth[2] = std::atan2(u1[1], u1[0]);
  Vec2 w{n.norm(), v.norm()};
  w = w / w.norm();
  orth[3] = std::asin(w[1]);
  return orth;
}

// orth_to_plk (line_geometry.cpp:86-126)
inline Vec6 orth_to_plk(const Vec4& orth) {
  Mat3 R = orth_R(orth[0], orth[1], orth[2]);
  double w1 = std::cos(orth[3]), w2 = std::sin(orth[3]);
  Vec6 plk;
  for (int i = 0; i < 3; ++i) { plk[i] = w1 * R(i, 0); plk[3 + i] = w2 * R(i, 1); }
  return plk;
}

// plk_to_pose (line_geometry.cpp:198-209): nc = Rcw nw + [tcw]x Rcw vw ; vc = Rcw vw
inline Vec6 plk_to_pose(const Vec6& plk_w, const Mat3& Rcw, const Vec3& tcw) {
  Vec3 nw{plk_w[0], plk_w[1], plk_w[2]}, vw{plk_w[3], plk_w[4], plk_w[5]};
  Vec3 nc = Rcw * nw + skew(tcw) * Rcw * vw;
  Vec3 vc = Rcw * vw;
  return Vec6{nc[0], nc[1], nc[2], vc[0], vc[1], vc[2]};
}
// plk_from_pose (line_geometry.cpp:211-216)
inline Vec6 plk_from_pose(const Vec6& plk_c, const Mat3& Rcw, const Vec3& tcw) {
  Mat3 Rwc = Rcw.T();
  Vec3 twc = -(Rwc * tcw);
  return plk_to_pose(plk_c, Rwc, twc);
}

// pi_from_ppp (line_geometry.cpp:134-139)
inline Vec4 pi_from_ppp(const Vec3& x1, const Vec3& x2, const Vec3& x3) {
  Vec3 n = cross(x1 - x3, x2 - x3);
  return Vec4{n[0], n[1], n[2], -x3.dot(cross(x1, x2))};
}

// pipi_plk (line_geometry.cpp:142-148): Pluecker line of the intersection of two planes, dp = pi1 pi2^T - pi2 pi1^T
inline Vec6 pipi_plk(const Vec4& a, const Vec4& b) {
  auto dp = [&](int i, int j) { return a[i] * b[j] - b[i] * a[j]; };
  return Vec6{dp(0, 3), dp(1, 3), dp(2, 3), -dp(1, 2), dp(0, 2), -dp(0, 1)};
}

}  // namespace orc
