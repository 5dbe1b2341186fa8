// TEST INFRASTRUCTURE -- CPU oracle (see smallmat.h).  Dense linear algebra the
// reference obtains from Eigen: LLT, PartialPivLU inverse, SelfAdjointEigenSolver.
#include "smallmat.h"

namespace orc {

bool cholesky_lower(MatX& A) {
  const int n = A.r;
  for (int j = 0; j < n; ++j) {
    double d = A(j, j);
    for (int k = 0; k < j; ++k) d -= A(j, k) * A(j, k);
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A(j, j) = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A(i, j);
      for (int k = 0; k < j; ++k) s -= A(i, k) * A(j, k);
      A(i, j) = s / d;
    }
  }
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) A(i, j) = 0.0;
  return true;
}

void cholesky_solve(const MatX& L, VecX& b) {
  const int n = L.r;
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L(i, k) * b[k];
    b[i] = s / L(i, i);
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L(k, i) * b[k];
    b[i] = s / L(i, i);
  }
}

bool inverse_lu(const MatX& A, MatX& Ainv) {
  const int n = A.r;
  MatX LU = A;
  std::vector<int> piv(n);
  bool singular = false;
  for (int i = 0; i < n; ++i) piv[i] = i;
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = std::fabs(LU(k, k));
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(LU(i, k)) > best) { best = std::fabs(LU(i, k)); p = i; }
    // an exactly singular column: Eigen's PartialPivLU (partial_lu_impl::unblocked_lu) notes the zero pivot, leaves the
    // column unscaled and goes on; the inverse then divides by it and comes out as inf / NaN.  The same here (it was a
    // `return false` that left Ainv empty: a landmark block without information made the caller read through a null matrix)
    if (best == 0.0) { singular = true; continue; }
    if (p != k) {
      for (int j = 0; j < n; ++j) std::swap(LU(k, j), LU(p, j));
      std::swap(piv[k], piv[p]);
    }
    for (int i = k + 1; i < n; ++i) {
      LU(i, k) /= LU(k, k);
      const double f = LU(i, k);
      if (f != 0.0)
        for (int j = k + 1; j < n; ++j) LU(i, j) -= f * LU(k, j);
    }
  }
  Ainv.resize(n, n);
  std::vector<double> col(n);
  for (int c = 0; c < n; ++c) {
    for (int i = 0; i < n; ++i) col[i] = (piv[i] == c) ? 1.0 : 0.0;
    for (int i = 0; i < n; ++i) {
      double s = col[i];
      for (int k = 0; k < i; ++k) s -= LU(i, k) * col[k];
      col[i] = s;
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = col[i];
      for (int k = i + 1; k < n; ++k) s -= LU(i, k) * col[k];
      col[i] = s / LU(i, i);
    }
    for (int i = 0; i < n; ++i) Ainv(i, c) = col[i];
  }
  return !singular;
}

// Cyclic Jacobi on the full symmetric matrix.  Converges to machine precision;
// eigenvalues returned ascending (Eigen's SelfAdjointEigenSolver order).
void sym_eigen(const MatX& Ain, VecX& w, MatX& V) {
  const int n = Ain.r;
  MatX A = Ain;
  // symmetrise (the solver reads one triangle in Eigen; inputs here are symmetric to rounding)
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) A(i, j) = A(j, i);
  V.resize(n, n);
  for (int i = 0; i < n; ++i) V(i, i) = 1.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; ++i) {
      diag += A(i, i) * A(i, i);
      for (int j = i + 1; j < n; ++j) off += A(i, j) * A(i, j);
    }
    if (off <= 1e-60 || off <= 1e-34 * diag) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A(p, q);
        if (apq == 0.0) continue;
        const double app = A(p, p), aqq = A(q, q);
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < n; ++k) {
          const double akp = A(k, p), akq = A(k, q);
          A(k, p) = cs * akp - sn * akq;
          A(k, q) = sn * akp + cs * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A(p, k), aqk = A(q, k);
          A(p, k) = cs * apk - sn * aqk;
          A(q, k) = sn * apk + cs * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V(k, p), vkq = V(k, q);
          V(k, p) = cs * vkp - sn * vkq;
          V(k, q) = sn * vkp + cs * vkq;
        }
      }
  }
  w.resize(n);
  std::vector<int> idx(n);
  for (int i = 0; i < n; ++i) { w[i] = A(i, i); idx[i] = i; }
  std::sort(idx.begin(), idx.end(), [&](int a, int b) { return w[a] < w[b]; });
  VecX ws(n);
  MatX Vs(n, n);
  for (int j = 0; j < n; ++j) {
    ws[j] = w[idx[j]];
    for (int i = 0; i < n; ++i) Vs(i, j) = V(i, idx[j]);
  }
  w = ws;
  V = Vs;
}

// Right singular vectors of an m x 4 matrix (row-major) by one-sided (Hestenes) Jacobi on its columns: the columns are
// rotated until mutually orthogonal, V accumulates the rotations.  V (4 x 4, row-major) is returned with its columns
// sorted by decreasing singular value, as Eigen::JacobiSVD::matrixV() is.
void svd4_right(const double* A_in, int m, double* V) {
  std::vector<double> A(A_in, A_in + (size_t)m * 4);
  for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int r = 0; r < m; ++r) { al += A[r * 4 + p] * A[r * 4 + p]; be += A[r * 4 + q] * A[r * 4 + q]; ga += A[r * 4 + p] * A[r * 4 + q]; }
        if (ga == 0.0 || std::fabs(ga) <= 1e-15 * std::sqrt(al * be)) continue;
        rotated = true;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (int r = 0; r < m; ++r) {
          const double a = A[r * 4 + p], b = A[r * 4 + q];
          A[r * 4 + p] = c * a - sn * b;
          A[r * 4 + q] = sn * a + c * b;
        }
        for (int r = 0; r < 4; ++r) {
          const double a = V[r * 4 + p], b = V[r * 4 + q];
          V[r * 4 + p] = c * a - sn * b;
          V[r * 4 + q] = sn * a + c * b;
        }
      }
    if (!rotated) break;
  }
  double sv[4];
  for (int c = 0; c < 4; ++c) { double s = 0; for (int r = 0; r < m; ++r) s += A[r * 4 + c] * A[r * 4 + c]; sv[c] = s; }
  for (int i = 0; i < 3; ++i)           // selection sort of the columns, largest singular value first
    for (int j = i + 1; j < 4; ++j)
      if (sv[j] > sv[i]) {
        std::swap(sv[i], sv[j]);
        for (int r = 0; r < 4; ++r) std::swap(V[r * 4 + i], V[r * 4 + j]);
      }
}

}  // namespace orc
