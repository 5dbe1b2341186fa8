// TEST INFRASTRUCTURE -- CPU oracle of the VPLines-SLAM bundle-adjustment path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.  The product (vplines-slam_amd/) never links it.
//
// smallmat.h: dependency-free fixed-size / dynamic dense matrices (the
// reference uses Eigen, which is not available in this image).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <cassert>
#include <algorithm>
#include <initializer_list>

namespace orc {

template <int R, int C>
struct Mat {
  double a[R * C];
  Mat() { for (int i = 0; i < R * C; ++i) a[i] = 0.0; }
  Mat(std::initializer_list<double> l) {
    int i = 0;
    for (double v : l) a[i++] = v;
    for (; i < R * C; ++i) a[i] = 0.0;
  }
  static Mat Zero() { return Mat(); }
  static Mat Identity() {
    Mat m;
    for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = 1.0;
    return m;
  }
  double& operator()(int r, int c) { return a[r * C + c]; }
  double operator()(int r, int c) const { return a[r * C + c]; }
  double& operator[](int i) { return a[i]; }
  double operator[](int i) const { return a[i]; }
  Mat<C, R> T() const {
    Mat<C, R> t;
    for (int r = 0; r < R; ++r)
      for (int c = 0; c < C; ++c) t(c, r) = (*this)(r, c);
    return t;
  }
  Mat operator+(const Mat& o) const { Mat m; for (int i = 0; i < R * C; ++i) m.a[i] = a[i] + o.a[i]; return m; }
  Mat operator-(const Mat& o) const { Mat m; for (int i = 0; i < R * C; ++i) m.a[i] = a[i] - o.a[i]; return m; }
  Mat operator-() const { Mat m; for (int i = 0; i < R * C; ++i) m.a[i] = -a[i]; return m; }
  Mat operator*(double s) const { Mat m; for (int i = 0; i < R * C; ++i) m.a[i] = a[i] * s; return m; }
  Mat operator/(double s) const { Mat m; for (int i = 0; i < R * C; ++i) m.a[i] = a[i] / s; return m; }
  Mat& operator+=(const Mat& o) { for (int i = 0; i < R * C; ++i) a[i] += o.a[i]; return *this; }
  Mat& operator-=(const Mat& o) { for (int i = 0; i < R * C; ++i) a[i] -= o.a[i]; return *this; }
  Mat& operator*=(double s) { for (int i = 0; i < R * C; ++i) a[i] *= s; return *this; }
  double squaredNorm() const { double s = 0; for (int i = 0; i < R * C; ++i) s += a[i] * a[i]; return s; }
  double norm() const { return std::sqrt(squaredNorm()); }
  Mat normalized() const { return (*this) / norm(); }
  double dot(const Mat& o) const { double s = 0; for (int i = 0; i < R * C; ++i) s += a[i] * o.a[i]; return s; }
  double maxCoeff() const { double m = a[0]; for (int i = 1; i < R * C; ++i) m = std::max(m, a[i]); return m; }
  double minCoeff() const { double m = a[0]; for (int i = 1; i < R * C; ++i) m = std::min(m, a[i]); return m; }
  template <int BR, int BC>
  Mat<BR, BC> block(int r0, int c0) const {
    Mat<BR, BC> b;
    for (int r = 0; r < BR; ++r)
      for (int c = 0; c < BC; ++c) b(r, c) = (*this)(r0 + r, c0 + c);
    return b;
  }
  template <int BR, int BC>
  void setBlock(int r0, int c0, const Mat<BR, BC>& b) {
    for (int r = 0; r < BR; ++r)
      for (int c = 0; c < BC; ++c) (*this)(r0 + r, c0 + c) = b(r, c);
  }
};

template <int R, int K, int C>
inline Mat<R, C> operator*(const Mat<R, K>& A, const Mat<K, C>& B) {
  Mat<R, C> m;
  for (int r = 0; r < R; ++r)
    for (int c = 0; c < C; ++c) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += A(r, k) * B(k, c);
      m(r, c) = s;
    }
  return m;
}
template <int R, int C>
inline Mat<R, C> operator*(double s, const Mat<R, C>& A) { return A * s; }

using Vec2 = Mat<2, 1>;
using Vec3 = Mat<3, 1>;
using Vec4 = Mat<4, 1>;
using Vec6 = Mat<6, 1>;
using Mat3 = Mat<3, 3>;
using Mat6 = Mat<6, 6>;

inline Vec3 cross(const Vec3& a, const Vec3& b) {
  return Vec3{a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
}

// Quaternion with Eigen's semantics (w,x,y,z ctor order, Hamilton product).
struct Quat {
  double w, x, y, z;
  Quat() : w(1), x(0), y(0), z(0) {}
  Quat(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}
  Quat operator*(const Quat& b) const {
    return Quat(w * b.w - x * b.x - y * b.y - z * b.z, w * b.x + x * b.w + y * b.z - z * b.y,
                w * b.y + y * b.w + z * b.x - x * b.z, w * b.z + z * b.w + x * b.y - y * b.x);
  }
  double squaredNorm() const { return w * w + x * x + y * y + z * z; }
  double norm() const { return std::sqrt(squaredNorm()); }
  Quat normalized() const { double n = norm(); return Quat(w / n, x / n, y / n, z / n); }
  void normalize() { *this = normalized(); }
  Quat conjugate() const { return Quat(w, -x, -y, -z); }
  // Eigen::Quaternion::inverse(): conjugate / squaredNorm
  Quat inverse() const {
    double n2 = squaredNorm();
    return Quat(w / n2, -x / n2, -y / n2, -z / n2);
  }
  Vec3 vec() const { return Vec3{x, y, z}; }
  // Eigen::QuaternionBase::toRotationMatrix (no normalisation performed)
  Mat3 toRotationMatrix() const {
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    Mat3 R;
    R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz;       R(0, 2) = txz + twy;
    R(1, 0) = txy + twz;       R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy;       R(2, 1) = tyz + twx;       R(2, 2) = 1 - (txx + tyy);
    return R;
  }
  // Eigen: q * v  ==  v + w*uv*2 + cross(q.vec, uv)*2, uv = cross(q.vec, v)
  Vec3 rotate(const Vec3& v) const {
    Vec3 qv = vec();
    Vec3 uv = cross(qv, v);
    uv = uv + uv;
    return v + uv * w + cross(qv, uv);
  }
  // Eigen::Quaternion(Matrix3) : rotation matrix -> quaternion
  static Quat fromRotationMatrix(const Mat3& m) {
    Quat q;
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0) {
      t = std::sqrt(t + 1.0);
      q.w = 0.5 * t;
      t = 0.5 / t;
      q.x = (m(2, 1) - m(1, 2)) * t;
      q.y = (m(0, 2) - m(2, 0)) * t;
      q.z = (m(1, 0) - m(0, 1)) * t;
    } else {
      int i = 0;
      if (m(1, 1) > m(0, 0)) i = 1;
      if (m(2, 2) > m(i, i)) i = 2;
      int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
      double qq[3];
      qq[i] = 0.5 * t;
      t = 0.5 / t;
      q.w = (m(k, j) - m(j, k)) * t;
      qq[j] = (m(j, i) + m(i, j)) * t;
      qq[k] = (m(k, i) + m(i, k)) * t;
      q.x = qq[0]; q.y = qq[1]; q.z = qq[2];
    }
    return q;
  }
};

// Dynamic dense matrix, row-major.
struct MatX {
  int r = 0, c = 0;
  std::vector<double> a;
  MatX() {}
  MatX(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
  void resize(int r_, int c_) { r = r_; c = c_; a.assign((size_t)r_ * c_, 0.0); }
  double& operator()(int i, int j) { return a[(size_t)i * c + j]; }
  double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
  void setZero() { std::fill(a.begin(), a.end(), 0.0); }
};
using VecX = std::vector<double>;

// Cholesky A = L L^T (lower), in place on a dense symmetric matrix; returns false if not PD.
bool cholesky_lower(MatX& A);
// Solve L L^T x = b given the factor from cholesky_lower.
void cholesky_solve(const MatX& L, VecX& b);
// Dense inverse by LU with partial pivoting (Eigen's MatrixXd::inverse() is PartialPivLU).
bool inverse_lu(const MatX& A, MatX& Ainv);
// Symmetric eigen-decomposition (cyclic Jacobi): A = V diag(w) V^T, eigenvalues ascending.
void sym_eigen(const MatX& A, VecX& w, MatX& V);
// right singular vectors (columns of V, 4 x 4 row-major, decreasing singular value) of an m x 4 row-major matrix
void svd4_right(const double* A, int m, double* V);

}  // namespace orc
