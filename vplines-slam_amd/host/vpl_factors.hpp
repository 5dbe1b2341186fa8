// Host-side mirror of the reference's factor / parameterisation / marginalisation interface for the
// bundle-adjustment path, forwarding to the C ABI of include/vplines_ba.h (HIP kernels).
//
// Same class names, constructor arguments, Evaluate()/Plus()/ComputeJacobian() signatures, block sizes,
// static sqrt_info members and Jacobian layouts as
//   vins_estimator/src/factor/projection_factor.h:10-24
//   vins_estimator/src/factor/line_projection_factor.h:12-36
//   vins_estimator/src/factor/imu_factor.h:16-23
//   vins_estimator/src/factor/marginalization_factor.h:15-83
//   vins_estimator/src/factor/pose_local_parameterization.h:7-13, line_parameterization.h:6-12
// so reference-shaped host code (and the parity tests) read like the reference's own.
//
// With -DVPL_USE_CERES the classes derive from the real ceres:: bases (the integration build);
// without it a minimal stand-in base is used so the header compiles where Ceres is absent.
// A single Evaluate() call is a batch of one on the device: correct but latency-bound -- the
// throughput path is vpl_ba_solve_windows(), see INTEGRATION.md.
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "vplines_ba.h"

#ifdef VPL_USE_CERES
#include <ceres/ceres.h>
namespace vplhost {
using CostFunctionBase = ceres::CostFunction;
using LocalParameterizationBase = ceres::LocalParameterization;
}
#else
namespace vplhost {
class CostFunctionBase {
 public:
  virtual ~CostFunctionBase() {}
  virtual bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const = 0;
  const std::vector<int>& parameter_block_sizes() const { return sizes_; }
  int num_residuals() const { return nres_; }
 protected:
  std::vector<int>* mutable_parameter_block_sizes() { return &sizes_; }
  void set_num_residuals(int n) { nres_ = n; }
  std::vector<int> sizes_;
  int nres_ = 0;
};
class LocalParameterizationBase {
 public:
  virtual ~LocalParameterizationBase() {}
  virtual bool Plus(const double* x, const double* delta, double* x_plus_delta) const = 0;
  virtual bool ComputeJacobian(const double* x, double* jacobian) const = 0;
  virtual int GlobalSize() const = 0;
  virtual int LocalSize() const = 0;
};
}  // namespace vplhost
#endif

namespace vplhost {

// One process-wide device context for the single-factor entry points (the reference evaluates its
// factors from one thread, estimator_node.cpp:229).
inline vpl_ctx* default_ctx() {
  static vpl_ctx* ctx = nullptr;
  if (!ctx) {
    int rc = vpl_ctx_create(&ctx, 0, 1, 8, 64, 8, 64);
    if (rc != VPL_OK) throw std::runtime_error("vplines: no HIP device / context (code " + std::to_string(rc) + "); there is no CPU fallback");
  }
  return ctx;
}
inline void check(int rc, const char* what) {
  if (rc != VPL_OK) throw std::runtime_error(std::string(what) + " failed: " + vpl_last_error(default_ctx()));
}

class PoseLocalParameterization : public LocalParameterizationBase {
 public:
  bool Plus(const double* x, const double* delta, double* x_plus_delta) const override {
    check(vpl_pose_plus(default_ctx(), 1, x, delta, x_plus_delta), "vpl_pose_plus");
    return true;
  }
  bool ComputeJacobian(const double*, double* jacobian) const override {   // [I6; 0], row-major 7x6
    std::memset(jacobian, 0, sizeof(double) * 42);
    for (int i = 0; i < 6; ++i) jacobian[i * 6 + i] = 1.0;
    return true;
  }
  int GlobalSize() const override { return 7; }
  int LocalSize() const override { return 6; }
};

class LineOrthParameterization : public LocalParameterizationBase {
 public:
  bool Plus(const double* x, const double* delta, double* x_plus_delta) const override {
    check(vpl_line_orth_plus(default_ctx(), 1, x, delta, x_plus_delta), "vpl_line_orth_plus");
    return true;
  }
  bool ComputeJacobian(const double*, double* jacobian) const override {   // I4
    std::memset(jacobian, 0, sizeof(double) * 16);
    for (int i = 0; i < 4; ++i) jacobian[i * 4 + i] = 1.0;
    return true;
  }
  int GlobalSize() const override { return 4; }
  int LocalSize() const override { return 4; }
};

namespace detail {
// scatter a packed jacobian buffer into the (possibly NULL) per-block pointers of the ceres ABI
inline void scatter(const double* packed, const int* sizes, int nblocks, int nres, double** jacobians) {
  size_t off = 0;
  for (int b = 0; b < nblocks; ++b) {
    const size_t len = (size_t)nres * sizes[b];
    if (jacobians[b]) std::memcpy(jacobians[b], packed + off, len * sizeof(double));
    off += len;
  }
}
}  // namespace detail

// SizedCostFunction<2, 7, 7, 7, 1>
class ProjectionFactor : public CostFunctionBase {
 public:
  ProjectionFactor(const double pts_i_[3], const double pts_j_[3]) {
    std::memcpy(pts, pts_i_, 24);
    std::memcpy(pts + 3, pts_j_, 24);
    *mutable_parameter_block_sizes() = {7, 7, 7, 1};
    set_num_residuals(2);
  }
  bool Evaluate(double const* const* p, double* residuals, double** jacobians) const override {
    double params[22], jac[44];
    std::memcpy(params, p[0], 56); std::memcpy(params + 7, p[1], 56); std::memcpy(params + 14, p[2], 56);
    params[21] = p[3][0];
    check(vpl_projection_factor_evaluate(default_ctx(), 1, params, pts, sqrt_info, residuals, jacobians ? jac : nullptr),
          "vpl_projection_factor_evaluate");
    const int sz[4] = {7, 7, 7, 1};
    if (jacobians) detail::scatter(jac, sz, 4, 2, jacobians);
    return true;
  }
  double pts[6];
  static inline double sqrt_info = 460.0 / 1.5;   // FOCAL_LENGTH / 1.5 (estimator.cpp:18); scalar * I2
};

// SizedCostFunction<2, 7, 7, 4>
class lineProjectionFactor : public CostFunctionBase {
 public:
  explicit lineProjectionFactor(const double obs_i_[4]) {
    std::memcpy(obs_i, obs_i_, 32);
    *mutable_parameter_block_sizes() = {7, 7, 4};
    set_num_residuals(2);
  }
  bool Evaluate(double const* const* p, double* residuals, double** jacobians) const override {
    double params[18], jac[36];
    std::memcpy(params, p[0], 56); std::memcpy(params + 7, p[1], 56); std::memcpy(params + 14, p[2], 32);
    check(vpl_line_factor_evaluate(default_ctx(), 1, params, obs_i, sqrt_info, residuals, jacobians ? jac : nullptr),
          "vpl_line_factor_evaluate");
    const int sz[3] = {7, 7, 4};
    if (jacobians) detail::scatter(jac, sz, 3, 2, jacobians);
    return true;
  }
  double obs_i[4];
  static inline double sqrt_info = 306.666666667;   // line_factor (estimator.cpp:19)
};

// SizedCostFunction<2, 7, 7, 4>
class vpProjectionFactor : public CostFunctionBase {
 public:
  explicit vpProjectionFactor(const double vp_[3]) {
    std::memcpy(obs_i, vp_, 24);
    *mutable_parameter_block_sizes() = {7, 7, 4};
    set_num_residuals(2);
  }
  bool Evaluate(double const* const* p, double* residuals, double** jacobians) const override {
    double params[18], jac[36];
    std::memcpy(params, p[0], 56); std::memcpy(params + 7, p[1], 56); std::memcpy(params + 14, p[2], 32);
    check(vpl_vp_factor_evaluate(default_ctx(), 1, params, obs_i, sqrt_info, residuals, jacobians ? jac : nullptr),
          "vpl_vp_factor_evaluate");
    const int sz[3] = {7, 7, 4};
    if (jacobians) detail::scatter(jac, sz, 3, 2, jacobians);
    return true;
  }
  double obs_i[3];
  static inline double sqrt_info = 10.0;   // vp_factor (estimator.cpp:20)
};

// Mirrors IntegrationBase (integration_base.h:9-249): push_back() buffers samples, the device
// pre-integrates them on demand.
class IntegrationBase {
 public:
  IntegrationBase(const double acc_0_[3], const double gyr_0_[3], const double ba[3], const double bg[3],
                  const vpl_ba_options& opt_) : opt(opt_) {
    std::memcpy(acc_0, acc_0_, 24); std::memcpy(gyr_0, gyr_0_, 24);
    std::memcpy(linearized_ba, ba, 24); std::memcpy(linearized_bg, bg, 24);
  }
  void push_back(double dt, const double acc[3], const double gyr[3]) {
    samples.push_back(dt);
    samples.insert(samples.end(), acc, acc + 3);
    samples.insert(samples.end(), gyr, gyr + 3);
    dirty = true;
  }
  const vpl_preintegration& result() const {
    if (dirty) {
      const int off = 0, n = (int)(samples.size() / 7);
      check(vpl_preintegrate_batch(default_ctx(), 1, &off, &n, samples.data(), acc_0, gyr_0, linearized_ba,
                                   linearized_bg, &opt, &pre), "vpl_preintegrate_batch");
      dirty = false;
    }
    return pre;
  }
  double acc_0[3], gyr_0[3], linearized_ba[3], linearized_bg[3];
  vpl_ba_options opt;
  std::vector<double> samples;
  mutable vpl_preintegration pre;
  mutable bool dirty = true;
};

// SizedCostFunction<15, 7, 9, 7, 9>
class IMUFactor : public CostFunctionBase {
 public:
  explicit IMUFactor(IntegrationBase* pre_) : pre_integration(pre_) {
    *mutable_parameter_block_sizes() = {7, 9, 7, 9};
    set_num_residuals(15);
  }
  bool Evaluate(double const* const* p, double* residuals, double** jacobians) const override {
    double params[32];
    std::vector<double> jac(jacobians ? 480 : 0);
    std::memcpy(params, p[0], 56); std::memcpy(params + 7, p[1], 72); std::memcpy(params + 16, p[2], 56);
    std::memcpy(params + 23, p[3], 72);
    check(vpl_imu_factor_evaluate(default_ctx(), 1, params, &pre_integration->result(), pre_integration->opt.g_norm,
                                  residuals, jacobians ? jac.data() : nullptr), "vpl_imu_factor_evaluate");
    const int sz[4] = {7, 9, 7, 9};
    if (jacobians) detail::scatter(jac.data(), sz, 4, 15, jacobians);
    return true;
  }
  IntegrationBase* pre_integration;
};

// MarginalizationFactor over a vpl_prior (= the fields of MarginalizationInfo the reference's
// factor reads: n, keep_block_{size,idx,data}, linearized_jacobians, linearized_residuals)
class MarginalizationFactor : public CostFunctionBase {
 public:
  explicit MarginalizationFactor(const vpl_prior* info) : marginalization_info(info) {
    for (int b = 0; b < info->n_blocks; ++b)
      mutable_parameter_block_sizes()->push_back(info->block_kind[b] == VPL_BLOCK_SPEEDBIAS ? 9 : 7);
    set_num_residuals(info->n);
  }
  bool Evaluate(double const* const* p, double* residuals, double** jacobians) const override {
    const vpl_prior* pr = marginalization_info;
    std::vector<double> params;
    std::vector<int> sz;
    for (int b = 0; b < pr->n_blocks; ++b) {
      const int s = pr->block_kind[b] == VPL_BLOCK_SPEEDBIAS ? 9 : 7;
      sz.push_back(s);
      params.insert(params.end(), p[b], p[b] + s);
    }
    std::vector<double> jac(jacobians ? (size_t)pr->n * params.size() : 0);
    check(vpl_prior_factor_evaluate(default_ctx(), pr, params.data(), residuals, jacobians ? jac.data() : nullptr),
          "vpl_prior_factor_evaluate");
    if (jacobians) detail::scatter(jac.data(), sz.data(), pr->n_blocks, pr->n, jacobians);
    return true;
  }
  const vpl_prior* marginalization_info;
};

}  // namespace vplhost
