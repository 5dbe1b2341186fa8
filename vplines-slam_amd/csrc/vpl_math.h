// Device math of the bundle-adjustment path: small fixed-size vector algebra,
// quaternion helpers, Pluecker/orthonormal line geometry and the four factor
// linearisations.  Everything is a __device__ inline working on registers.
//
// Reference behaviour reproduced (path:line relative to the reference tree):
//   quaternion helpers            vins_estimator/src/utility/utility.h:12-108
//   line geometry                 vins_estimator/src/utility/line_geometry.cpp:62-126,198-216
//   ProjectionFactor              vins_estimator/src/factor/projection_factor.cpp:6-126
//   lineProjectionFactor          vins_estimator/src/factor/line_projection_factor.cpp:251-380
//   vpProjectionFactor            vins_estimator/src/factor/line_projection_factor.cpp:11-153
//   IMUFactor / IntegrationBase   vins_estimator/src/factor/imu_factor.h:23-182, integration_base.h:200-226
//   parameterisations             pose_local_parameterization.cpp:3-27, line_parameterization.cpp:7-100
//   Triggs corrector              marginalization_factor.cpp:37-68
//
// VPL_HD is __host__ __device__ under hipcc; tests/native compiles this header
// with a host compiler (VPL_HD empty) to finite-difference the formulas without
// a GPU.  The product library only ever calls them from kernels.
#pragma once
#include <math.h>

#ifdef __HIPCC__
#define VPL_HD __host__ __device__ __forceinline__
#else
#define VPL_HD inline
#endif

namespace vpl {

struct V3 {
  double x, y, z;
};
struct M3 {  // row-major
  double m[9];
};

VPL_HD V3 v3(double x, double y, double z) { return V3{x, y, z}; }
VPL_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
VPL_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
VPL_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
VPL_HD V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }
VPL_HD V3 operator*(double s, V3 a) { return V3{a.x * s, a.y * s, a.z * s}; }
VPL_HD double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
VPL_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
VPL_HD double norm(V3 a) { return sqrt(dot(a, a)); }
VPL_HD double get(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

VPL_HD V3 mul(const M3& A, V3 v) {
  return V3{A.m[0] * v.x + A.m[1] * v.y + A.m[2] * v.z, A.m[3] * v.x + A.m[4] * v.y + A.m[5] * v.z,
            A.m[6] * v.x + A.m[7] * v.y + A.m[8] * v.z};
}
VPL_HD V3 mulT(const M3& A, V3 v) {  // A^T v
  return V3{A.m[0] * v.x + A.m[3] * v.y + A.m[6] * v.z, A.m[1] * v.x + A.m[4] * v.y + A.m[7] * v.z,
            A.m[2] * v.x + A.m[5] * v.y + A.m[8] * v.z};
}
VPL_HD M3 mul(const M3& A, const M3& B) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
  return C;
}
VPL_HD M3 mulTA(const M3& A, const M3& B) {  // A^T B
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[i] * B.m[j] + A.m[3 + i] * B.m[3 + j] + A.m[6 + i] * B.m[6 + j];
  return C;
}
VPL_HD M3 transpose(const M3& A) {
  M3 C;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * j + i];
  return C;
}
VPL_HD M3 skew(V3 q) {  // utility.h:27-34
  M3 S;
  S.m[0] = 0;    S.m[1] = -q.z; S.m[2] = q.y;
  S.m[3] = q.z;  S.m[4] = 0;    S.m[5] = -q.x;
  S.m[6] = -q.y; S.m[7] = q.x;  S.m[8] = 0;
  return S;
}
// A * skew(v): column j of the product is A (e_j-th column of skew)
VPL_HD M3 mul_skew(const M3& A, V3 v) { return mul(A, skew(v)); }

struct Q4 {  // Hamilton quaternion, Eigen conventions
  double w, x, y, z;
};
VPL_HD Q4 qmul(Q4 a, Q4 b) {
  return Q4{a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
VPL_HD Q4 qinv(Q4 q) {  // Eigen: conjugate / squaredNorm
  double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  return Q4{q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}
VPL_HD Q4 qnormalized(Q4 q) {
  double n = sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return Q4{q.w / n, q.x / n, q.y / n, q.z / n};
}
VPL_HD Q4 qpose(const double* p) { return Q4{p[6], p[3], p[4], p[5]}; }  // block layout qx,qy,qz,qw
VPL_HD M3 qmat(Q4 q) {  // Eigen toRotationMatrix (no normalisation)
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
  const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  M3 R;
  R.m[0] = 1 - (tyy + tzz); R.m[1] = txy - twz;       R.m[2] = txz + twy;
  R.m[3] = txy + twz;       R.m[4] = 1 - (txx + tzz); R.m[5] = tyz - twx;
  R.m[6] = txz - twy;       R.m[7] = tyz + twx;       R.m[8] = 1 - (txx + tyy);
  return R;
}
VPL_HD V3 qrot(Q4 q, V3 v) {  // Eigen _transformVector
  V3 qv{q.x, q.y, q.z};
  V3 uv = cross(qv, v);
  uv = uv + uv;
  return v + uv * q.w + cross(qv, uv);
}
VPL_HD Q4 mat2q(const M3& m) {  // Eigen Quaternion(Matrix3)
  Q4 q;
  double t = m.m[0] + m.m[4] + m.m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q.w = 0.5 * t;
    t = 0.5 / t;
    q.x = (m.m[7] - m.m[5]) * t;
    q.y = (m.m[2] - m.m[6]) * t;
    q.z = (m.m[3] - m.m[1]) * t;
  } else {
    int i = 0;
    if (m.m[4] > m.m[0]) i = 1;
    if (m.m[8] > m.m[4 * i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = sqrt(m.m[4 * i] - m.m[4 * j] - m.m[4 * k] + 1.0);
    double qq[3];
    qq[i] = 0.5 * t;
    t = 0.5 / t;
    q.w = (m.m[3 * k + j] - m.m[3 * j + k]) * t;
    qq[j] = (m.m[3 * j + i] + m.m[3 * i + j]) * t;
    qq[k] = (m.m[3 * k + i] + m.m[3 * i + k]) * t;
    q.x = qq[0]; q.y = qq[1]; q.z = qq[2];
  }
  return q;
}
VPL_HD Q4 deltaQ(V3 th) { return Q4{1.0, th.x / 2.0, th.y / 2.0, th.z / 2.0}; }  // utility.h:12-24

// Utility::R2ypr / ypr2R (utility.h:66-108), degrees
VPL_HD V3 R2ypr(const M3& R) {
  V3 n{R.m[0], R.m[3], R.m[6]}, o{R.m[1], R.m[4], R.m[7]}, a{R.m[2], R.m[5], R.m[8]};
  double y = atan2(n.y, n.x);
  double p = atan2(-n.z, n.x * cos(y) + n.y * sin(y));
  double r = atan2(a.x * sin(y) - a.y * cos(y), -o.x * sin(y) + o.y * cos(y));
  return V3{y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0};
}
VPL_HD M3 ypr2R(V3 ypr) {
  double y = ypr.x / 180.0 * M_PI, p = ypr.y / 180.0 * M_PI, r = ypr.z / 180.0 * M_PI;
  M3 Rz{{cos(y), -sin(y), 0, sin(y), cos(y), 0, 0, 0, 1}};
  M3 Ry{{cos(p), 0., sin(p), 0., 1., 0., -sin(p), 0., cos(p)}};
  M3 Rx{{1., 0., 0., 0., cos(r), -sin(r), 0., sin(r), cos(r)}};
  return mul(mul(Rz, Ry), Rx);
}

// ---- pose / line parameterisations ------------------------------------------
// PoseLocalParameterization::Plus
VPL_HD void pose_plus(const double* x, const double* d, double* out) {
  Q4 q = qpose(x);
  Q4 dq = deltaQ(V3{d[3], d[4], d[5]});
  Q4 qn = qnormalized(qmul(q, dq));
  out[0] = x[0] + d[0]; out[1] = x[1] + d[1]; out[2] = x[2] + d[2];
  out[3] = qn.x; out[4] = qn.y; out[5] = qn.z; out[6] = qn.w;
}
// rotation matrix of the 3 orthonormal angles (line_geometry.cpp:100-104)
VPL_HD M3 orth_R(double t0, double t1, double t2) {
  double s1 = sin(t0), c1 = cos(t0), s2 = sin(t1), c2 = cos(t1), s3 = sin(t2), c3 = cos(t2);
  M3 R{{c2 * c3, s1 * s2 * c3 - c1 * s3, c1 * s2 * c3 + s1 * s3,
        c2 * s3, s1 * s2 * s3 + c1 * c3, c1 * s2 * s3 - s1 * c3,
        -s2,     s1 * c2,                c1 * c2}};
  return R;
}
// LineOrthParameterization::Plus
VPL_HD void line_orth_plus(const double* x, const double* d, double* out) {
  M3 R = orth_R(x[0], x[1], x[2]);
  double w1 = cos(x[3]), w2 = sin(x[3]);
  M3 Rz{{cos(d[2]), -sin(d[2]), 0, sin(d[2]), cos(d[2]), 0, 0, 0, 1}};
  M3 Ry{{cos(d[1]), 0., sin(d[1]), 0., 1., 0., -sin(d[1]), 0., cos(d[1])}};
  M3 Rx{{1., 0., 0., 0., cos(d[0]), -sin(d[0]), 0., sin(d[0]), cos(d[0])}};
  R = mul(mul(mul(R, Rx), Ry), Rz);
  // W = [[w1,-w2],[w2,w1]] * [[c,-s],[s,c]] ; W(1,0) = w2*c + w1*s
  double W10 = w2 * cos(d[3]) + w1 * sin(d[3]);
  out[0] = atan2(R.m[7], R.m[8]);
  out[1] = asin(-R.m[6]);
  out[2] = atan2(R.m[3], R.m[0]);
  out[3] = asin(W10);
}
struct Plk {
  V3 n, v;
};
VPL_HD Plk orth_to_plk(const double* o) {
  M3 R = orth_R(o[0], o[1], o[2]);
  double w1 = cos(o[3]), w2 = sin(o[3]);
  Plk L;
  L.n = V3{w1 * R.m[0], w1 * R.m[3], w1 * R.m[6]};
  L.v = V3{w2 * R.m[1], w2 * R.m[4], w2 * R.m[7]};
  return L;
}
VPL_HD void plk_to_orth(Plk L, double* o) {
  double nn = norm(L.n), vn = norm(L.v);
  V3 u1{L.n.x / nn, L.n.y / nn, L.n.z / nn};
  V3 u2{L.v.x / vn, L.v.y / vn, L.v.z / vn};
  V3 u3 = cross(u1, u2);
  o[0] = atan2(u2.z, u3.z);
  o[1] = asin(-u1.z);
  o[2] = atan2(u1.y, u1.x);
  double wn = sqrt(nn * nn + vn * vn);
  o[3] = asin(vn / wn);
}
// plk_to_pose: n' = R n + t x (R v), v' = R v
VPL_HD Plk plk_to_pose(Plk L, const M3& R, V3 t) {
  Plk o;
  o.v = mul(R, L.v);
  o.n = mul(R, L.n) + cross(t, o.v);
  return o;
}
// plk_from_pose(L, R, t) = plk_to_pose(L, R^T, -R^T t)
VPL_HD Plk plk_from_pose(Plk L, const M3& R, V3 t) {
  Plk o;
  o.v = mulT(R, L.v);
  V3 twc = -mulT(R, t);
  o.n = mulT(R, L.n) + cross(twc, o.v);
  return o;
}

// ---- robust loss (ceres HuberLoss + corrector; rho'' <= 0 always => alpha = 0) ----
// returns rho0; *scale = sqrt(rho1): residuals and Jacobians are both multiplied by it
VPL_HD double huber(double s, double delta, double* scale) {
  const double b = delta * delta;
  if (s > b) {
    const double r = sqrt(s);
    double rho1 = delta / r;
    if (rho1 < 2.2250738585072014e-308) rho1 = 2.2250738585072014e-308;
    *scale = sqrt(rho1);
    return 2.0 * delta * r - b;
  }
  *scale = 1.0;
  return s;
}

// ---- ProjectionFactor ---------------------------------------------------------
// inputs: pose_i, pose_j, ex (7 each), inverse depth, pts_i, pts_j (z = 1)
// outputs: r[2]; if J: Ji[2][6], Jj[2][6], Je[2][6], Jl[2]   (local columns; the 7th
// global column is zero by construction)
struct PoseR {  // a pose block unpacked once per iteration
  V3 p;
  M3 R;
};
VPL_HD PoseR unpack_pose(const double* x) {
  PoseR o;
  o.p = V3{x[0], x[1], x[2]};
  o.R = qmat(qpose(x));
  return o;
}

VPL_HD void tangent_basis(V3 pts_j, V3* b1, V3* b2) {  // projection_factor.cpp:9-18
  const double rn = 1.0 / norm(pts_j);   // one division, three multiplies (an f64 divide is ~12 instructions)
  V3 a{pts_j.x * rn, pts_j.y * rn, pts_j.z * rn};
  V3 tmp{0, 0, 1};
  if (a.x == 0.0 && a.y == 0.0 && a.z == 1.0) tmp = V3{1, 0, 0};
  double at = dot(a, tmp);
  V3 t = tmp - a * at;
  const double rtn = 1.0 / norm(t);
  *b1 = V3{t.x * rtn, t.y * rtn, t.z * rtn};
  *b2 = cross(a, *b1);
}

// Works on quaternion-rotated points exactly like the reference for the residual
// (q * v), and on rotation matrices for the Jacobians.
VPL_HD void projection_factor(const double* pi, const double* pj, const double* ex, double inv_dep, V3 pts_i,
                              V3 pts_j, double sqrt_info, double* r, bool want_jac, double* Ji, double* Jj,
                              double* Je, double* Jl) {
  V3 Pi{pi[0], pi[1], pi[2]}, Pj{pj[0], pj[1], pj[2]}, tic{ex[0], ex[1], ex[2]};
  Q4 Qi = qpose(pi), Qj = qpose(pj), qic = qpose(ex);
  const double dep = 1.0 / inv_dep;
  V3 pts_camera_i{pts_i.x * dep, pts_i.y * dep, pts_i.z * dep};
  V3 pts_imu_i = qrot(qic, pts_camera_i) + tic;
  V3 pts_w = qrot(Qi, pts_imu_i) + Pi;
  V3 pts_imu_j = qrot(qinv(Qj), pts_w - Pj);
  V3 pts_camera_j = qrot(qinv(qic), pts_imu_j - tic);
  V3 b1, b2;
  tangent_basis(pts_j, &b1, &b2);
  const double ncj = norm(pts_camera_j), rncj = 1.0 / ncj, rnj = 1.0 / norm(pts_j);
  V3 diff{pts_camera_j.x * rncj - pts_j.x * rnj, pts_camera_j.y * rncj - pts_j.y * rnj, pts_camera_j.z * rncj - pts_j.z * rnj};
  r[0] = sqrt_info * dot(b1, diff);
  r[1] = sqrt_info * dot(b2, diff);
  if (!want_jac) return;

  M3 Ri = qmat(Qi), Rj = qmat(Qj), ric = qmat(qic);
  const double rn3 = rncj * rncj * rncj;
  double x1 = pts_camera_j.x, x2 = pts_camera_j.y, x3 = pts_camera_j.z;
  M3 nj_{{rncj - x1 * x1 * rn3, -x1 * x2 * rn3, -x1 * x3 * rn3,
          -x1 * x2 * rn3, rncj - x2 * x2 * rn3, -x2 * x3 * rn3,
          -x1 * x3 * rn3, -x2 * x3 * rn3, rncj - x3 * x3 * rn3}};
  // reduce (2x3) = sqrt_info * tangent_base * norm_jaco
  double red[6];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    red[c] = sqrt_info * (b1.x * nj_.m[c] + b1.y * nj_.m[3 + c] + b1.z * nj_.m[6 + c]);
    red[3 + c] = sqrt_info * (b2.x * nj_.m[c] + b2.y * nj_.m[3 + c] + b2.z * nj_.m[6 + c]);
  }
  M3 ricT_RjT = mulTA(ric, transpose(Rj));   // ric^T Rj^T
  M3 A = mul(ricT_RjT, Ri);                  // ric^T Rj^T Ri
  // pose i : [ricT_RjT | -A [pts_imu_i]x]
  M3 Bi = mul_skew(A, pts_imu_i);
  // pose j : [-ricT_RjT | ric^T [pts_imu_j]x]
  M3 Bj = mulTA(ric, skew(pts_imu_j));
  // ex : [ric^T (Rj^T Ri - I) | -T [pc_i]x + [T pc_i]x + [ric^T (Rj^T (Ri tic + Pi - Pj) - tic)]x],  T = A ric
  M3 RjT_Ri = mulTA(Rj, Ri);
  M3 RjT_Ri_mI = RjT_Ri;
  RjT_Ri_mI.m[0] -= 1.0; RjT_Ri_mI.m[4] -= 1.0; RjT_Ri_mI.m[8] -= 1.0;
  M3 Ce = mulTA(ric, RjT_Ri_mI);
  M3 T = mul(A, ric);
  M3 TS = mul_skew(T, pts_camera_i);
  M3 S2 = skew(mul(T, pts_camera_i));
  V3 inner = mulT(Rj, mul(Ri, tic) + Pi - Pj) - tic;
  M3 S3 = skew(mulT(ric, inner));
  M3 De;
#pragma unroll
  for (int k = 0; k < 9; ++k) De.m[k] = -TS.m[k] + S2.m[k] + S3.m[k];
  V3 tl = mul(T, pts_i);
  double sl = -(dep * dep);
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const double a0 = red[3 * rr], a1 = red[3 * rr + 1], a2 = red[3 * rr + 2];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      Ji[6 * rr + c] = a0 * ricT_RjT.m[c] + a1 * ricT_RjT.m[3 + c] + a2 * ricT_RjT.m[6 + c];
      Ji[6 * rr + 3 + c] = -(a0 * Bi.m[c] + a1 * Bi.m[3 + c] + a2 * Bi.m[6 + c]);
      Jj[6 * rr + c] = -(a0 * ricT_RjT.m[c] + a1 * ricT_RjT.m[3 + c] + a2 * ricT_RjT.m[6 + c]);
      Jj[6 * rr + 3 + c] = a0 * Bj.m[c] + a1 * Bj.m[3 + c] + a2 * Bj.m[6 + c];
      Je[6 * rr + c] = a0 * Ce.m[c] + a1 * Ce.m[3 + c] + a2 * Ce.m[6 + c];
      Je[6 * rr + 3 + c] = a0 * De.m[c] + a1 * De.m[3 + c] + a2 * De.m[6 + c];
    }
    Jl[rr] = (a0 * tl.x + a1 * tl.y + a2 * tl.z) * sl;
  }
}

// ---- line / VP factors --------------------------------------------------------
// Shared transform chain.  jel (2x3) is d r / d nc for the line factor and the
// reference's literal "jaco_e_l" for the VP factor; sel selects which half of the
// 6-vector L_c it multiplies (0: normal part, 1: direction part).
struct LineCtx {
  Plk Lw, Lb, Lc;
  M3 Rwb, Rbc;
  V3 twb, tbc;
};
VPL_HD LineCtx line_ctx_plk(const double* pose, const double* ex, const Plk& Lw) {   // world line given in Pluecker coordinates
  LineCtx c;
  c.twb = V3{pose[0], pose[1], pose[2]};
  c.Rwb = qmat(qpose(pose));
  c.tbc = V3{ex[0], ex[1], ex[2]};
  c.Rbc = qmat(qpose(ex));
  c.Lw = Lw;
  c.Lb = plk_from_pose(c.Lw, c.Rwb, c.twb);
  c.Lc = plk_from_pose(c.Lb, c.Rbc, c.tbc);
  return c;
}
VPL_HD LineCtx line_ctx(const double* pose, const double* ex, const double* orth) {
  return line_ctx_plk(pose, ex, orth_to_plk(orth));
}
// rows of a 2x3 matrix times 3x3 blocks -> accumulate into 2x6 / 2x4 outputs
VPL_HD void line_chain_jac(const LineCtx& c, const double* jel, int sel, double* Jp, double* Je, double* Jo) {
  // dLc/d(pose) = invTbc * [[Rwb^T [dw]x , [Rwb^T (nw + dw x twb)]x],[0, [Rwb^T dw]x]]
  // invTbc = [[Rbc^T, -Rbc^T [tbc]x],[0, Rbc^T]]
  M3 RbcT = transpose(c.Rbc);
  M3 P00 = mulTA(c.Rwb, skew(c.Lw.v));                                   // Rwb^T [dw]x
  M3 P01 = skew(mulT(c.Rwb, c.Lw.n + cross(c.Lw.v, c.twb)));            // [Rwb^T (nw + [dw]x twb)]x
  M3 P11 = skew(mulT(c.Rwb, c.Lw.v));                                    // [Rwb^T dw]x
  M3 RbcT_St = mul(RbcT, skew(c.tbc));                                   // Rbc^T [tbc]x
  // top rows (normal part) of invTbc*P: [RbcT P00 , RbcT P01 - RbcT_St P11] ; bottom: [0, RbcT P11]
  M3 T00 = mul(RbcT, P00);
  M3 T01a = mul(RbcT, P01), T01b = mul(RbcT_St, P11);
  M3 T11 = mul(RbcT, P11);
  // ex: [[Rbc^T [db]x, [Rbc^T (nb + db x tbc)]x],[0,[Rbc^T db]x]]  (no left factor)
  M3 E00 = mulTA(c.Rbc, skew(c.Lb.v));
  M3 E01 = skew(mulT(c.Rbc, c.Lb.n + cross(c.Lb.v, c.tbc)));
  M3 E11 = skew(mulT(c.Rbc, c.Lb.v));
  // orth: invTwc * K ; invTwc = [[Rwc^T, -Rwc^T [twc]x],[0,Rwc^T]]
  M3 Rwc = mul(c.Rwb, c.Rbc);
  V3 twc = mul(c.Rwb, c.tbc) + c.twb;
  double nn = norm(c.Lw.n), vn = norm(c.Lw.v);
  const double rnn = 1.0 / nn, rvn = 1.0 / vn;
  V3 u1{c.Lw.n.x * rnn, c.Lw.n.y * rnn, c.Lw.n.z * rnn}, u2{c.Lw.v.x * rvn, c.Lw.v.y * rvn, c.Lw.v.z * rvn};
  V3 u3 = cross(u1, u2);
  const double rwn = 1.0 / sqrt(nn * nn + vn * vn);
  double w0 = nn * rwn, w1 = vn * rwn;
  // K columns (6x4): c0 = [0; w1 u3], c1 = [-w0 u3; 0], c2 = [w0 u2; -w1 u1], c3 = [-w1 u1; w0 u2]
  V3 Kt[4] = {V3{0, 0, 0}, u3 * (-w0), u2 * w0, u1 * (-w1)};   // top (normal) halves
  V3 Kb[4] = {u3 * w1, V3{0, 0, 0}, u1 * (-w1), u2 * w0};      // bottom (direction) halves
  V3 Ot[4], Ob[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    Ob[k] = mulT(Rwc, Kb[k]);
    Ot[k] = mulT(Rwc, Kt[k]) - mulT(Rwc, cross(twc, Kb[k]));
  }
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const double a0 = jel[3 * rr], a1 = jel[3 * rr + 1], a2 = jel[3 * rr + 2];
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      if (sel == 0) {
        Jp[6 * rr + cc] = a0 * T00.m[cc] + a1 * T00.m[3 + cc] + a2 * T00.m[6 + cc];
        Jp[6 * rr + 3 + cc] = a0 * (T01a.m[cc] - T01b.m[cc]) + a1 * (T01a.m[3 + cc] - T01b.m[3 + cc]) + a2 * (T01a.m[6 + cc] - T01b.m[6 + cc]);
        Je[6 * rr + cc] = a0 * E00.m[cc] + a1 * E00.m[3 + cc] + a2 * E00.m[6 + cc];
        Je[6 * rr + 3 + cc] = a0 * E01.m[cc] + a1 * E01.m[3 + cc] + a2 * E01.m[6 + cc];
      } else {
        Jp[6 * rr + cc] = 0.0;
        Jp[6 * rr + 3 + cc] = a0 * T11.m[cc] + a1 * T11.m[3 + cc] + a2 * T11.m[6 + cc];
        Je[6 * rr + cc] = 0.0;
        Je[6 * rr + 3 + cc] = a0 * E11.m[cc] + a1 * E11.m[3 + cc] + a2 * E11.m[6 + cc];
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      V3 o = sel == 0 ? Ot[k] : Ob[k];
      Jo[4 * rr + k] = a0 * o.x + a1 * o.y + a2 * o.z;
    }
  }
}

// lineProjectionFactor: obs = (x1,y1,x2,y2)
VPL_HD void line_factor_res(const LineCtx& c, const double* obs, double sqrt_info, double* r, double* jel) {
  V3 nc = c.Lc.n;
  double l_norm = nc.x * nc.x + nc.y * nc.y;
  double l_sqrtnorm = sqrt(l_norm);
  double l_trinorm = l_norm * l_sqrtnorm;
  double e1 = obs[0] * nc.x + obs[1] * nc.y + nc.z;
  double e2 = obs[2] * nc.x + obs[3] * nc.y + nc.z;
  const double rs = 1.0 / l_sqrtnorm;
  r[0] = sqrt_info * (e1 * rs);
  r[1] = sqrt_info * (e2 * rs);
  if (jel) {
    const double rt = 1.0 / l_trinorm;
    jel[0] = sqrt_info * (obs[0] * rs - nc.x * e1 * rt);
    jel[1] = sqrt_info * (obs[1] * rs - nc.y * e1 * rt);
    jel[2] = sqrt_info * rs;
    jel[3] = sqrt_info * (obs[2] * rs - nc.x * e2 * rt);
    jel[4] = sqrt_info * (obs[3] * rs - nc.y * e2 * rt);
    jel[5] = sqrt_info * rs;
  }
}
// vpProjectionFactor: vp = (x,y,z); jel is the reference's literal matrix (sic)
VPL_HD void vp_factor_res(const LineCtx& c, const double* vp, double sqrt_info, double* r, double* jel) {
  V3 d = c.Lc.v;
  double v1 = vp[0] / vp[2], v2 = vp[1] / vp[2];
  r[0] = sqrt_info * (d.x / d.z - v1);
  r[1] = sqrt_info * (d.y / d.z - v2);
  if (jel) {
    jel[0] = sqrt_info * (-1 / vp[2]); jel[1] = sqrt_info * 0.0; jel[2] = sqrt_info * v1;
    jel[3] = sqrt_info * 0.0; jel[4] = sqrt_info * (-1 / vp[2]); jel[5] = sqrt_info * v2;
  }
}

// ---- IMU factor -----------------------------------------------------------------
struct PreInt {      // what IMUFactor::Evaluate reads from IntegrationBase
  double sum_dt;
  V3 dp, dv, lba, lbg;
  Q4 dq;
  M3 dp_dba, dp_dbg, dq_dbg, dv_dba, dv_dbg;
};
// raw (un-whitened) residual, IntegrationBase::evaluate
VPL_HD void imu_residual_raw(const PreInt& p, const double* pi, const double* sbi, const double* pj, const double* sbj,
                             double g_norm, double* r) {
  V3 Pi{pi[0], pi[1], pi[2]}, Pj{pj[0], pj[1], pj[2]};
  Q4 Qi = qpose(pi), Qj = qpose(pj);
  V3 Vi{sbi[0], sbi[1], sbi[2]}, Bai{sbi[3], sbi[4], sbi[5]}, Bgi{sbi[6], sbi[7], sbi[8]};
  V3 Vj{sbj[0], sbj[1], sbj[2]}, Baj{sbj[3], sbj[4], sbj[5]}, Bgj{sbj[6], sbj[7], sbj[8]};
  V3 G{0, 0, g_norm};
  V3 dba = Bai - p.lba, dbg = Bgi - p.lbg;
  Q4 cq = qmul(p.dq, deltaQ(mul(p.dq_dbg, dbg)));
  V3 cv = p.dv + mul(p.dv_dba, dba) + mul(p.dv_dbg, dbg);
  V3 cp = p.dp + mul(p.dp_dba, dba) + mul(p.dp_dbg, dbg);
  double dt = p.sum_dt;
  Q4 Qii = qinv(Qi);
  V3 rp = qrot(Qii, G * (0.5 * dt * dt) + Pj - Pi - Vi * dt) - cp;
  Q4 qe = qmul(qinv(cq), qmul(Qii, Qj));
  V3 rq{2 * qe.x, 2 * qe.y, 2 * qe.z};
  V3 rv = qrot(Qii, G * dt + Vj - Vi) - cv;
  V3 rba = Baj - Bai, rbg = Bgj - Bgi;
  r[0] = rp.x; r[1] = rp.y; r[2] = rp.z; r[3] = rq.x; r[4] = rq.y; r[5] = rq.z;
  r[6] = rv.x; r[7] = rv.y; r[8] = rv.z; r[9] = rba.x; r[10] = rba.y; r[11] = rba.z;
  r[12] = rbg.x; r[13] = rbg.y; r[14] = rbg.z;
}
VPL_HD M3 qleft_br(Q4 q) {  // bottomRightCorner<3,3> of Utility::Qleft: w I + [v]x
  M3 S = skew(V3{q.x, q.y, q.z});
  S.m[0] += q.w; S.m[4] += q.w; S.m[8] += q.w;
  return S;
}
VPL_HD M3 qright_br(Q4 q) {  // w I - [v]x
  M3 S = skew(V3{-q.x, -q.y, -q.z});
  S.m[0] += q.w; S.m[4] += q.w; S.m[8] += q.w;
  return S;
}
// The non-zero 3x3 blocks of the raw (un-whitened) IMU Jacobian, imu_factor.h:90-178.
// Local column layout of the 30 columns: [pose_i 0..5 | sb_i 6..14 | pose_j 15..20 | sb_j 21..29].
struct ImuJac {
  M3 pi_pp, pi_pr, pi_rr, pi_vr;                                  // pose_i : (P,P) (P,R) (R,R) (V,R)
  M3 si_pv, si_pba, si_pbg, si_rbg, si_vv, si_vba, si_vbg;        // sb_i
  M3 pj_pp, pj_rr;                                                // pose_j
  M3 sj_vv;                                                       // sb_j (V,V); (BA,BA)=(BG,BG)=I ; sb_i (BA,BA)=(BG,BG)=-I
};
VPL_HD M3 neg(const M3& A) {
  M3 C;
#pragma unroll
  for (int k = 0; k < 9; ++k) C.m[k] = -A.m[k];
  return C;
}
VPL_HD M3 scale(const M3& A, double s) {
  M3 C;
#pragma unroll
  for (int k = 0; k < 9; ++k) C.m[k] = A.m[k] * s;
  return C;
}
VPL_HD ImuJac imu_jacobian_raw(const PreInt& p, const double* pi, const double* sbi, const double* pj,
                               const double* sbj, double g_norm) {
  V3 Pi{pi[0], pi[1], pi[2]}, Pj{pj[0], pj[1], pj[2]};
  Q4 Qi = qpose(pi), Qj = qpose(pj);
  V3 Vi{sbi[0], sbi[1], sbi[2]}, Bgi{sbi[6], sbi[7], sbi[8]};
  V3 Vj{sbj[0], sbj[1], sbj[2]};
  V3 G{0, 0, g_norm};
  double dt = p.sum_dt;
  Q4 Qii = qinv(Qi);
  M3 RiT = qmat(Qii);
  Q4 cq = qmul(p.dq, deltaQ(mul(p.dq_dbg, Bgi - p.lbg)));
  ImuJac J;
  J.pi_pp = neg(RiT);
  J.pi_pr = skew(qrot(Qii, G * (0.5 * dt * dt) + Pj - Pi - Vi * dt));
  // -(Qleft(Qj^-1 Qi) * Qright(cq)).bottomRight: the 4x4 product's lower-right 3x3 block is
  //   v_l v_r^T + (w_l I + [v_l]x)(w_r I - [v_r]x)
  {
    Q4 ql = qmul(qinv(Qj), Qi);
    M3 A = qleft_br(ql), B = qright_br(cq);
    M3 AB = mul(A, B);
    V3 vl{ql.x, ql.y, ql.z}, vr{cq.x, cq.y, cq.z};
    // Qleft(1:3,0) = v_l (column), Qright(0,1:3) = -v_r^T (row)  => outer product term is -v_l v_r^T
    double vlv[3] = {vl.x, vl.y, vl.z}, vrv[3] = {vr.x, vr.y, vr.z};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) AB.m[3 * i + j] += -vlv[i] * vrv[j];
    J.pi_rr = neg(AB);
  }
  J.pi_vr = skew(qrot(Qii, G * dt + Vj - Vi));
  J.si_pv = scale(RiT, -dt);
  J.si_pba = neg(p.dp_dba);
  J.si_pbg = neg(p.dp_dbg);
  J.si_rbg = neg(mul(qleft_br(qmul(qmul(qinv(Qj), Qi), cq)), p.dq_dbg));
  J.si_vv = neg(RiT);
  J.si_vba = neg(p.dv_dba);
  J.si_vbg = neg(p.dv_dbg);
  J.pj_pp = RiT;
  J.pj_rr = qleft_br(qmul(qmul(qinv(cq), Qii), Qj));
  J.sj_vv = RiT;
  return J;
}
// scatter the block form into a dense 15x30 row-major matrix (zero-filled by the caller)
VPL_HD void put33(double* J, int ld, int r0, int c0, const M3& B) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) J[(r0 + i) * ld + c0 + j] = B.m[3 * i + j];
}
VPL_HD void imu_jac_dense(const ImuJac& B, double* J /*15x30, pre-zeroed*/) {
  put33(J, 30, 0, 0, B.pi_pp); put33(J, 30, 0, 3, B.pi_pr); put33(J, 30, 3, 3, B.pi_rr); put33(J, 30, 6, 3, B.pi_vr);
  put33(J, 30, 0, 6, B.si_pv); put33(J, 30, 0, 9, B.si_pba); put33(J, 30, 0, 12, B.si_pbg);
  put33(J, 30, 3, 12, B.si_rbg);
  put33(J, 30, 6, 6, B.si_vv); put33(J, 30, 6, 9, B.si_vba); put33(J, 30, 6, 12, B.si_vbg);
  J[9 * 30 + 9] = -1; J[10 * 30 + 10] = -1; J[11 * 30 + 11] = -1;
  J[12 * 30 + 12] = -1; J[13 * 30 + 13] = -1; J[14 * 30 + 14] = -1;
  put33(J, 30, 0, 15, B.pj_pp); put33(J, 30, 3, 18, B.pj_rr);
  put33(J, 30, 6, 21, B.sj_vv);
  J[9 * 30 + 24] = 1; J[10 * 30 + 25] = 1; J[11 * 30 + 26] = 1;
  J[12 * 30 + 27] = 1; J[13 * 30 + 28] = 1; J[14 * 30 + 29] = 1;
}

}  // namespace vpl
