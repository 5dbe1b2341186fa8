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
#include <algorithm>
#include <cmath>
#include <limits>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "vplines_ba.h"

#ifdef VPL_USE_CERES
#include <ceres/ceres.h>
namespace vplhost {
using CostFunctionBase = ceres::CostFunction;
using LocalParameterizationBase = ceres::LocalParameterization;
using LossFunctionBase = ceres::LossFunction;
using HuberLoss = ceres::HuberLoss;
}
#else
namespace vplhost {
// stand-ins for ceres::LossFunction / ceres::HuberLoss (estimator.cpp:1048); the device applies the loss, these carry `a`
class LossFunctionBase {
 public:
  virtual ~LossFunctionBase() {}
  virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class HuberLoss : public LossFunctionBase {
 public:
  explicit HuberLoss(double a) : a_(a), b_(a * a) {}
  void Evaluate(double s, double rho[3]) const override {
    if (s > b_) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * a_ * r - b_; rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r); rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
  }
  double a_, b_;
};
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

// ---- MarginalizationInfo (marginalization_factor.h:15-72) ------------------------------------------------------------
// Same members and the same four calls as the reference; the arithmetic of preMarginalize (factor evaluation + loss
// correction) and marginalize (A = sum J^T J, Schur complement, factorisation of the kept block) runs on the device through
// vpl_ba_marginalize.  The reference identifies parameter blocks by ADDRESS only; the device path needs to know which
// block is which, so the window's parameter arrays are named once with setWindowArrays() (the one call a caller adds; the
// arrays are Estimator's para_Pose / para_SpeedBias / para_Ex_Pose, estimator.h:139-143).  Supported factor sets are the
// two the reference builds: MARGIN_OLD (estimator.cpp:1229-1358: the last prior, IMUFactor(0,1), ProjectionFactor /
// lineProjectionFactor blocks of the tracks that start in frame 0) and MARGIN_SECOND_NEW (:1387-1405: the prior alone).
struct ResidualBlockInfo {
  ResidualBlockInfo(CostFunctionBase* _cost_function, LossFunctionBase* _loss_function, std::vector<double*> _parameter_blocks,
                    std::vector<int> _drop_set)
      : cost_function(_cost_function), loss_function(_loss_function), parameter_blocks(_parameter_blocks), drop_set(_drop_set) {}
  CostFunctionBase* cost_function;
  LossFunctionBase* loss_function;
  std::vector<double*> parameter_blocks;
  std::vector<int> drop_set;
};

class MarginalizationInfo {
 public:
  explicit MarginalizationInfo(const vpl_ba_options& opt_) : opt(opt_) { std::memset(&prior, 0, sizeof(int) * 2); }
  ~MarginalizationInfo() {   // owns the factors, their cost functions and the x0 copies; not the loss functions (:71-87)
    for (auto& it : parameter_block_data) delete[] it.second;
    for (auto* f : factors) { delete f->cost_function; delete f; }
  }
  int localSize(int size) const { return size == 7 ? 6 : size; }
  int globalSize(int size) const { return size == 6 ? 7 : size; }

  void setWindowArrays(double (*pose)[7], double (*speed_bias)[9], double* ex_pose) {
    para_Pose = pose; para_SpeedBias = speed_bias; para_Ex_Pose = ex_pose;
  }
  // marginalization_factor.cpp:89-108
  void addResidualBlockInfo(ResidualBlockInfo* info) {
    factors.emplace_back(info);
    const std::vector<int>& sizes = info->cost_function->parameter_block_sizes();
    for (size_t i = 0; i < info->parameter_blocks.size(); ++i)
      parameter_block_size[reinterpret_cast<long>(info->parameter_blocks[i])] = sizes[i];
    for (int d : info->drop_set) parameter_block_idx[reinterpret_cast<long>(info->parameter_blocks[d])] = 0;
  }
  // marginalization_factor.cpp:110-129: the linearisation point x0 of every block is frozen here; the factor evaluation the
  // reference also does here happens inside marginalize() on the device, at the same values
  void preMarginalize() {
    for (auto* it : factors) {
      const std::vector<int>& sizes = it->cost_function->parameter_block_sizes();
      for (size_t i = 0; i < sizes.size(); ++i) {
        const long addr = reinterpret_cast<long>(it->parameter_blocks[i]);
        if (parameter_block_data.find(addr) == parameter_block_data.end()) {
          double* data = new double[sizes[i]];
          std::memcpy(data, it->parameter_blocks[i], sizeof(double) * sizes[i]);
          parameter_block_data[addr] = data;
        }
      }
    }
  }
  // marginalization_factor.cpp:177-363
  void marginalize();
  // marginalization_factor.cpp:458-478: kept blocks in prior order, their addresses shifted by the caller's map
  std::vector<double*> getParameterBlocks(std::unordered_map<long, double*>& addr_shift) {
    std::vector<double*> keep_block_addr;
    keep_block_size.clear(); keep_block_idx.clear(); keep_block_data.clear();
    for (int b = 0; b < prior.n_blocks; ++b) {
      double* cur = block_address(prior.block_kind[b], kept_current_frame[b]);
      keep_block_size.push_back(prior.block_kind[b] == VPL_BLOCK_SPEEDBIAS ? 9 : 7);
      keep_block_idx.push_back(prior.block_idx[b] + m);
      keep_block_data.push_back(prior.x0[b]);
      keep_block_addr.push_back(addr_shift.at(reinterpret_cast<long>(cur)));
    }
    return keep_block_addr;
  }

  std::vector<ResidualBlockInfo*> factors;
  int m = 0, n = 0;
  std::unordered_map<long, int> parameter_block_size;      // global size
  std::unordered_map<long, int> parameter_block_idx;       // dropped blocks (value unused here)
  std::unordered_map<long, double*> parameter_block_data;  // x0 copies
  std::vector<int> keep_block_size, keep_block_idx;
  std::vector<double*> keep_block_data;
  std::vector<double> linearized_jacobians;                // row-major n x n
  std::vector<double> linearized_residuals;
  vpl_prior prior;                                         // the same prior in the C ABI's form (MarginalizationFactor reads it)
  vpl_ba_options opt;

 private:
  double (*para_Pose)[7] = nullptr;
  double (*para_SpeedBias)[9] = nullptr;
  double* para_Ex_Pose = nullptr;
  int kept_current_frame[VPL_MAX_PRIOR_BLOCKS] = {0};
  int pose_frame(const double* p) const {
    for (int f = 0; f < VPL_NFRAMES; ++f) if (p == para_Pose[f]) return f;
    return -1;
  }
  int sb_frame(const double* p) const {
    for (int f = 0; f < VPL_NFRAMES; ++f) if (p == para_SpeedBias[f]) return f;
    return -1;
  }
  double* block_address(int kind, int frame) const {
    return kind == VPL_BLOCK_POSE ? para_Pose[frame] : kind == VPL_BLOCK_SPEEDBIAS ? para_SpeedBias[frame] : para_Ex_Pose;
  }
};

// MarginalizationFactor(MarginalizationInfo*), the reference's constructor (marginalization_factor.cpp:480-490)
inline MarginalizationFactor make_marginalization_factor(const MarginalizationInfo* info) { return MarginalizationFactor(&info->prior); }

inline void MarginalizationInfo::marginalize() {
  if (!para_Pose || !para_SpeedBias || !para_Ex_Pose) throw std::runtime_error("MarginalizationInfo: setWindowArrays() first");
  vpl_window w;
  std::memset(&w, 0, sizeof(w));
  std::memcpy(w.pose, para_Pose, sizeof(w.pose));
  std::memcpy(w.speed_bias, para_SpeedBias, sizeof(w.speed_bias));
  std::memcpy(w.ex_pose, para_Ex_Pose, sizeof(w.ex_pose));
  for (int j = 0; j < VPL_NFRAMES; ++j) w.preint[j].sum_dt = 1e30;     // "no IMU factor" unless one is added below
  struct PTrack { int start; std::map<int, const double*> obs; const double* pts_i; double lambda; };
  struct LTrack { int start; std::map<int, const double*> obs; const double* orth; };
  std::map<const double*, PTrack> ptracks;
  std::map<const double*, LTrack> ltracks;
  std::vector<const double*> porder, lorder;
  const MarginalizationFactor* last_prior = nullptr;
  bool other = false;
  for (auto* f : factors) {
    if (auto* pf = dynamic_cast<ProjectionFactor*>(f->cost_function)) {
      const int fi = pose_frame(f->parameter_blocks[0]), fj = pose_frame(f->parameter_blocks[1]);
      if (fi < 0 || fj <= fi || f->parameter_blocks[2] != para_Ex_Pose) throw std::runtime_error("MarginalizationInfo: ProjectionFactor blocks are not (pose_i, pose_j, ex_pose, lambda) of this window");
      const double* key = f->parameter_blocks[3];
      if (!ptracks.count(key)) { ptracks[key] = PTrack{fi, {}, pf->pts, *key}; porder.push_back(key); }
      ptracks[key].obs[fj] = pf->pts + 3;
      other = true;
    } else if (auto* lf = dynamic_cast<lineProjectionFactor*>(f->cost_function)) {
      const int fj = pose_frame(f->parameter_blocks[0]);
      if (fj < 0 || f->parameter_blocks[1] != para_Ex_Pose) throw std::runtime_error("MarginalizationInfo: lineProjectionFactor blocks are not (pose, ex_pose, orth) of this window");
      const double* key = f->parameter_blocks[2];
      if (!ltracks.count(key)) { ltracks[key] = LTrack{0, {}, key}; lorder.push_back(key); }
      ltracks[key].obs[fj] = lf->obs_i;
      other = true;
    } else if (auto* imu = dynamic_cast<IMUFactor*>(f->cost_function)) {
      const int fi = pose_frame(f->parameter_blocks[0]), fj = pose_frame(f->parameter_blocks[2]);
      if (fi != 0 || fj != 1 || sb_frame(f->parameter_blocks[1]) != 0 || sb_frame(f->parameter_blocks[3]) != 1)
        throw std::runtime_error("MarginalizationInfo: only IMUFactor(0, 1) is marginalised (estimator.cpp:1261)");
      w.preint[1] = imu->pre_integration->result();
      other = true;
    } else if (auto* mf = dynamic_cast<MarginalizationFactor*>(f->cost_function)) {
      last_prior = mf;
    } else {
      throw std::runtime_error("MarginalizationInfo: unsupported cost function (the reference marginalises no VP factors, estimator.cpp:1341-1353)");
    }
  }
  if (last_prior) { w.has_prior = 1; w.prior = last_prior->marginalization_info; }
  // tracks -> the vpl_window layout (consecutive frames from the start frame; the start observation of a line takes no part
  // in the marginalisation, estimator.cpp:1322-1326, its slot stays zero)
  std::vector<int> pstart, pnobs, lstart, lnobs;
  std::vector<double> pobs, invd, lobs, lorth;
  for (const double* key : porder) {
    const PTrack& t = ptracks[key];
    const int last = t.obs.rbegin()->first;
    pstart.push_back(t.start); pnobs.push_back(last - t.start + 1); invd.push_back(t.lambda);
    pobs.insert(pobs.end(), t.pts_i, t.pts_i + 3);
    for (int fr = t.start + 1; fr <= last; ++fr) {
      auto it = t.obs.find(fr);
      if (it == t.obs.end()) throw std::runtime_error("MarginalizationInfo: point track with a gap");
      pobs.insert(pobs.end(), it->second, it->second + 3);
    }
  }
  for (const double* key : lorder) {
    const LTrack& t = ltracks[key];
    const int last = t.obs.rbegin()->first;
    lstart.push_back(0); lnobs.push_back(last + 1);
    for (int fr = 0; fr <= last; ++fr) {
      auto it = t.obs.find(fr);
      double o8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (it != t.obs.end()) std::memcpy(o8, it->second, 32);
      else if (fr != 0) throw std::runtime_error("MarginalizationInfo: line track with a gap");
      lobs.insert(lobs.end(), o8, o8 + 8);
    }
    lorth.insert(lorth.end(), key, key + 4);
  }
  std::vector<double> plk_dummy(6 * std::max<size_t>(1, lorder.size()), 0.0);
  w.n_points = (int)porder.size(); w.point_start = pstart.data(); w.point_nobs = pnobs.data(); w.point_obs = pobs.data();
  w.inv_depth = invd.data();
  w.n_lines = (int)lorder.size(); w.line_start = lstart.data(); w.line_nobs = lnobs.data(); w.line_obs = lobs.data();
  w.line_plk = plk_dummy.data(); w.line_orth = lorth.data();
  const int flag = other ? VPL_MARGIN_OLD : VPL_MARGIN_SECOND_NEW;
  // a context sized for this window (the single-factor default context is too small)
  vpl_ctx* ctx = nullptr;
  const int npo = (int)(pobs.size() / 3), nlo = (int)(lobs.size() / 8);
  int rc = vpl_ctx_create(&ctx, 0, 1, std::max(1, w.n_points), std::max(1, npo), std::max(1, w.n_lines), std::max(1, nlo));
  if (rc != VPL_OK) throw std::runtime_error("MarginalizationInfo: vpl_ctx_create failed (" + std::to_string(rc) + "); there is no CPU fallback");
  rc = vpl_ba_marginalize(ctx, 1, &w, &opt, flag, &prior, &m, &n);
  const std::string err = rc == VPL_OK ? "" : vpl_last_error(ctx);
  vpl_ctx_destroy(ctx);
  if (rc != VPL_OK) throw std::runtime_error("vpl_ba_marginalize failed: " + err);
  linearized_jacobians.assign(prior.J0, prior.J0 + (size_t)n * n);
  linearized_residuals.assign(prior.r0, prior.r0 + n);
  // frames of the kept blocks in THIS window (the prior numbers them for the next one)
  for (int b = 0; b < prior.n_blocks; ++b)
    kept_current_frame[b] = prior.block_kind[b] == VPL_BLOCK_EXPOSE ? 0
                            : (flag == VPL_MARGIN_OLD ? prior.block_frame[b] + 1      // frame f of this window is f - 1 of the next
                               // MARGIN_SECOND_NEW: pose WINDOW_SIZE-1 is gone, frame WINDOW_SIZE was renumbered to it
                               : (prior.block_frame[b] == VPL_NFRAMES - 2 ? VPL_NFRAMES - 1 : prior.block_frame[b]));
}

}  // namespace vplhost
