// TEST INFRASTRUCTURE -- CPU oracle (see smallmat.h).
// A minimal ceres::Problem / ceres::Solve restatement covering exactly what the
// reference configures at vins_estimator/src/estimator.cpp:1046-1215:
//   TRUST_REGION + DOGLEG(TRADITIONAL) + DENSE_SCHUR, jacobi scaling, monotonic steps.
// ceres-solver is a third-party dependency pinned at 1.12.0 (docker/Dockerfile:3)
// and is ABSENT from /root/reference: this file restates its published algorithm
// (trust_region_minimizer.cc, dogleg_strategy.cc, schur_eliminator_impl.h,
// corrector.cc, residual_block.cc) from the builder's knowledge of that release.
// PARITY UNPINNED at this boundary: the reference holds no golden vectors.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "factors.h"

namespace orc {

struct SolverOptions {
  int max_num_iterations = 50;
  // ceres 1.12 defaults (solver.h)
  double initial_trust_region_radius = 1e4;
  double max_trust_region_radius = 1e16;
  double min_trust_region_radius = 1e-32;
  double min_relative_decrease = 1e-3;
  double min_lm_diagonal = 1e-6;
  double max_lm_diagonal = 1e32;
  double function_tolerance = 1e-6;
  double gradient_tolerance = 1e-10;
  double parameter_tolerance = 1e-8;
  int max_num_consecutive_invalid_steps = 5;
  bool jacobi_scaling = true;
  bool use_dogleg = true;  // false => LEVENBERG_MARQUARDT (onlyLineOpt, estimator.cpp:1024-1030)
};

enum Termination { NO_CONVERGENCE = 0, CONVERGENCE = 1, FAILURE = 2 };

struct IterationSummary {
  int iteration = 0;
  bool step_is_valid = false, step_is_successful = false;
  double cost = 0, cost_change = 0, gradient_max_norm = 0, step_norm = 0, relative_decrease = 0,
         trust_region_radius = 0;
};
struct SolverSummary {
  double initial_cost = 0, final_cost = 0;
  int num_successful_steps = 0, num_unsuccessful_steps = 0;
  Termination termination_type = NO_CONVERGENCE;
  std::string message;
  std::vector<IterationSummary> iterations;
};

class Problem {
 public:
  // Ownership follows ceres defaults: the problem owns costs, losses and parameterisations.
  ~Problem();
  void AddParameterBlock(double* values, int size, LocalParameterization* lp = nullptr);
  void SetParameterBlockConstant(double* values);
  void AddResidualBlock(CostFunction* cost, LossFunction* loss, const std::vector<double*>& blocks);

  struct ParamBlock {
    double* user = nullptr;
    int size = 0;
    LocalParameterization* lp = nullptr;
    bool constant = false;
    int local_size() const { return lp ? lp->LocalSize() : size; }
  };
  struct ResidualBlock {
    CostFunction* cost;
    LossFunction* loss;
    std::vector<int> blocks;  // indices into params_
  };
  std::vector<ParamBlock> params_;
  std::vector<ResidualBlock> residuals_;
  std::map<double*, int> index_;
  std::vector<LossFunction*> owned_losses_;
};

void Solve(const SolverOptions& options, Problem* problem, SolverSummary* summary);

}  // namespace orc
