// TEST INFRASTRUCTURE -- CPU oracle (see smallmat.h).
// Restates vins_estimator/src/factor/marginalization_factor.{h,cpp}:
//   ResidualBlockInfo::Evaluate            marginalization_factor.cpp:3-69
//   MarginalizationInfo::addResidualBlockInfo / preMarginalize / marginalize /
//   getParameterBlocks                     marginalization_factor.cpp:89-129,177-363,458-478
//   MarginalizationFactor::Evaluate        marginalization_factor.cpp:492-542
// Difference kept deliberate: the reference iterates std::unordered_map keyed by
// parameter ADDRESS (platform-dependent order); here the maps are ordered by
// address, which for the reference's member layout (estimator.h:139-143:
// para_Pose, para_SpeedBias, para_Feature, para_LineFeature, para_Ex_Pose) is the
// deterministic order documented in DESIGN.md.
#pragma once
#include <map>
#include <unordered_map>
#include <vector>
#include "factors.h"

namespace orc {

extern double g_marg_rel_eps;   // test switch, see marginalization.cpp
extern int g_marg_reverse_sums;   // test switch, see marginalization.cpp
extern int g_marg_threads;   // threads of the A, b assembly (ThreadsConstructA); set through orc_set_marg_threads


struct ResidualBlockInfo {
  ResidualBlockInfo(CostFunction* c, LossFunction* l, std::vector<double*> pb, std::vector<int> ds)
      : cost_function(c), loss_function(l), parameter_blocks(std::move(pb)), drop_set(std::move(ds)) {}
  void Evaluate();
  CostFunction* cost_function;
  LossFunction* loss_function;
  std::vector<double*> parameter_blocks;
  std::vector<int> drop_set;
  std::vector<std::vector<double>> jacobians;  // row-major nres x global size
  std::vector<double> residuals;
};

class MarginalizationInfo {
 public:
  ~MarginalizationInfo();
  static int localSize(int size) { return size == 7 ? 6 : size; }
  void addResidualBlockInfo(ResidualBlockInfo* info);
  void preMarginalize();
  void marginalize();
  std::vector<double*> getParameterBlocks(std::map<long, double*>& addr_shift);

  std::vector<ResidualBlockInfo*> factors;
  int m = 0, n = 0;
  std::map<long, int> parameter_block_size;  // global size
  std::map<long, int> parameter_block_idx;   // local idx
  std::map<long, double*> parameter_block_data;
  int sum_block_size = 0;

  std::vector<int> keep_block_size;
  std::vector<int> keep_block_idx;
  std::vector<double*> keep_block_data;
  // when rebuilt from a vpl_prior the data are owned here
  std::vector<std::vector<double>> owned_keep_data;

  MatX linearized_jacobians;  // n x n
  VecX linearized_residuals;
  // exposed for the invariant tests (the reference's commented check, marginalization_factor.cpp:361-362)
  MatX A_final;
  VecX b_final;
  const double eps = 1e-8;
};

struct MarginalizationFactor : CostFunction {
  explicit MarginalizationFactor(MarginalizationInfo* info);
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override;
  MarginalizationInfo* marginalization_info;
};

}  // namespace orc
