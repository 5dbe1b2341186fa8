// TEST INFRASTRUCTURE -- CPU oracle (see smallmat.h).
// Ceres-shaped cost functions restating the reference factors:
//   ProjectionFactor        vins_estimator/src/factor/projection_factor.cpp:6-126
//   lineProjectionFactor    vins_estimator/src/factor/line_projection_factor.cpp:251-380
//   vpProjectionFactor      vins_estimator/src/factor/line_projection_factor.cpp:11-153
//   IntegrationBase         vins_estimator/src/factor/integration_base.h:9-249
//   IMUFactor               vins_estimator/src/factor/imu_factor.h:23-182
//   Pose/LineOrth parameterisation  pose_local_parameterization.cpp:3-27, line_parameterization.cpp:7-100
//   HuberLoss / CauchyLoss  ceres-solver 1.12.0 loss_function.cc (third party, absent; published formulas)
#pragma once
#include <vector>
#include <limits>
#include "geometry.h"

namespace orc {

// ---- minimal Ceres-like interfaces ----------------------------------------
struct CostFunction {
  virtual ~CostFunction() {}
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  const std::vector<int>& parameter_block_sizes() const { return sizes_; }
  int num_residuals() const { return nres_; }
 protected:
  std::vector<int> sizes_;
  int nres_ = 0;
};

struct LossFunction {
  virtual ~LossFunction() {}
  virtual void Evaluate(double s, double rho[3]) const = 0;
};
struct HuberLoss : LossFunction {
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (s > b_) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a_ * r - b_;
      rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
  }
  double a_, b_;
};
struct CauchyLoss : LossFunction {
  explicit CauchyLoss(double a) : b_(a * a), c_(1.0 / (a * a)) {}
  void Evaluate(double s, double rho[3]) const override {
    const double sum = 1.0 + s * c_;
    const double inv = 1.0 / sum;
    rho[0] = b_ * std::log(sum);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -c_ * (inv * inv);
  }
  double b_, c_;
};

// Triggs corrector; identical in ceres corrector.cc and in the reference's
// ResidualBlockInfo::Evaluate (marginalization_factor.cpp:37-68).
struct Corrector {
  double sqrt_rho1, residual_scaling, alpha_sq_norm;
  Corrector(double sq_norm, const double rho[3]) {
    sqrt_rho1 = std::sqrt(rho[1]);
    if ((sq_norm == 0.0) || (rho[2] <= 0.0)) {
      residual_scaling = sqrt_rho1;
      alpha_sq_norm = 0.0;
    } else {
      const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
      const double alpha = 1.0 - std::sqrt(D);
      residual_scaling = sqrt_rho1 / (1 - alpha);
      alpha_sq_norm = alpha / sq_norm;
    }
  }
  // J <- sqrt_rho1 (J - alpha_sq_norm r r^T J), J row-major nres x ncols
  void CorrectJacobian(int nres, int ncols, const double* r, double* J) const {
    if (alpha_sq_norm == 0.0) {
      for (int i = 0; i < nres * ncols; ++i) J[i] *= sqrt_rho1;
      return;
    }
    for (int c = 0; c < ncols; ++c) {
      double rtj = 0;
      for (int k = 0; k < nres; ++k) rtj += J[k * ncols + c] * r[k];
      for (int k = 0; k < nres; ++k) J[k * ncols + c] = sqrt_rho1 * (J[k * ncols + c] - alpha_sq_norm * r[k] * rtj);
    }
  }
  void CorrectResiduals(int nres, double* r) const {
    for (int i = 0; i < nres; ++i) r[i] *= residual_scaling;
  }
};

struct LocalParameterization {
  virtual ~LocalParameterization() {}
  virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const = 0;
  virtual bool ComputeJacobian(const double* x, double* jacobian) const = 0;
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
};

struct PoseLocalParameterization : LocalParameterization {
  bool Plus(const double* x, const double* delta, double* xpd) const override;
  bool ComputeJacobian(const double* x, double* jacobian) const override;
  int GlobalSize() const override { return 7; }
  int LocalSize() const override { return 6; }
};
struct LineOrthParameterization : LocalParameterization {
  bool Plus(const double* x, const double* delta, double* xpd) const override;
  bool ComputeJacobian(const double* x, double* jacobian) const override;
  int GlobalSize() const override { return 4; }
  int LocalSize() const override { return 4; }
};

// ---- factors ---------------------------------------------------------------
struct ProjectionFactor : CostFunction {
  ProjectionFactor(const Vec3& pts_i, const Vec3& pts_j);
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override;
  Vec3 pts_i, pts_j;
  Mat<2, 3> tangent_base;
  static double sqrt_info;  // scalar * I2 (estimator.cpp:18)
};

struct lineProjectionFactor : CostFunction {
  explicit lineProjectionFactor(const Vec4& obs);
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override;
  Vec4 obs_i;
  static double sqrt_info;  // estimator.cpp:19
};

struct vpProjectionFactor : CostFunction {
  explicit vpProjectionFactor(const Vec3& vp);
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override;
  Vec3 obs_i;
  static double sqrt_info;  // estimator.cpp:20
};

struct ImuNoise { double acc_n, gyr_n, acc_w, gyr_w; };

struct IntegrationBase {
  IntegrationBase(const Vec3& acc_0, const Vec3& gyr_0, const Vec3& lin_ba, const Vec3& lin_bg, const ImuNoise& nz);
  void push_back(double dt, const Vec3& acc, const Vec3& gyr);
  void propagate(double dt, const Vec3& acc_1, const Vec3& gyr_1);
  void midPointIntegration(double dt, const Vec3& acc_0, const Vec3& gyr_0, const Vec3& acc_1, const Vec3& gyr_1,
                           const Vec3& delta_p, const Quat& delta_q, const Vec3& delta_v, const Vec3& lin_ba,
                           const Vec3& lin_bg, Vec3& res_p, Quat& res_q, Vec3& res_v, bool update_jacobian);
  Mat<15, 1> evaluate(const Vec3& Pi, const Quat& Qi, const Vec3& Vi, const Vec3& Bai, const Vec3& Bgi,
                      const Vec3& Pj, const Quat& Qj, const Vec3& Vj, const Vec3& Baj, const Vec3& Bgj,
                      const Vec3& G) const;
  double dt = 0;
  Vec3 acc_0, gyr_0, acc_1, gyr_1;
  Vec3 linearized_ba, linearized_bg;
  Mat<15, 15> jacobian, covariance;
  Mat<18, 18> noise;
  double sum_dt = 0;
  Vec3 delta_p;
  Quat delta_q;
  Vec3 delta_v;
};

struct IMUFactor : CostFunction {
  IMUFactor(const IntegrationBase* pre, const Vec3& G);
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override;
  const IntegrationBase* pre_integration;
  Vec3 G;
  Mat<15, 15> sqrt_info;  // chol(cov^-1).L^T ; the reference recomputes it per call (imu_factor.h:68)
};

}  // namespace orc
