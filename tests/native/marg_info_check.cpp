// GPU test program: the reference-shaped marginalisation (vplhost::MarginalizationInfo, ResidualBlockInfo, IMUFactor,
// IntegrationBase, MarginalizationFactor of vplines-slam_amd/host/vpl_factors.hpp) driven exactly as
// Estimator::optimizationwithLine() drives the reference's classes for MARGIN_OLD (estimator.cpp:1229-1378).  Reads one
// window from a text dump written by tests/test_gpu_host_adapter.py, prints the results for the comparison with the oracle.
#include <cstdio>
#include <cstdlib>
#include <memory>
#include "../../vplines-slam_amd/host/vpl_factors.hpp"

using namespace vplhost;

static double rd(FILE* f) { double v; if (fscanf(f, "%lf", &v) != 1) { fprintf(stderr, "short dump\n"); exit(2); } return v; }
static int ri(FILE* f) { return (int)rd(f); }

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "r");
  if (!f) return 2;
  static double para_Pose[VPL_NFRAMES][7], para_SpeedBias[VPL_NFRAMES][9], para_Ex_Pose[1][7];
  for (auto& p : para_Pose) for (double& v : p) v = rd(f);
  for (auto& p : para_SpeedBias) for (double& v : p) v = rd(f);
  for (double& v : para_Ex_Pose[0]) v = rd(f);
  vpl_ba_options opt;
  vpl_ba_default_options(&opt);

  // ---- IntegrationBase + IMUFactor of the interval (0, 1) ----
  const int ns = ri(f);
  double acc0[3], gyr0[3], ba[3], bg[3];
  for (double& v : acc0) v = rd(f);
  for (double& v : gyr0) v = rd(f);
  for (double& v : ba) v = rd(f);
  for (double& v : bg) v = rd(f);
  IntegrationBase pre(acc0, gyr0, ba, bg, opt);
  for (int k = 0; k < ns; ++k) {
    double s[7];
    for (double& v : s) v = rd(f);
    pre.push_back(s[0], s + 1, s + 4);
  }
  const vpl_preintegration& pr = pre.result();
  std::printf("preint %.17g", pr.sum_dt);
  for (int k = 0; k < 3; ++k) std::printf(" %.17g", pr.delta_p[k]);
  for (int k = 0; k < 4; ++k) std::printf(" %.17g", pr.delta_q[k]);
  for (int k = 0; k < 3; ++k) std::printf(" %.17g", pr.delta_v[k]);
  std::printf("\n");
  {
    IMUFactor imu(&pre);
    const double* params[4] = {para_Pose[0], para_SpeedBias[0], para_Pose[1], para_SpeedBias[1]};
    double r[15], J0[105], J1[135], J2[105], J3[135];
    double* J[4] = {J0, J1, J2, J3};
    imu.Evaluate(params, r, J);
    std::printf("imu");
    for (double v : r) std::printf(" %.17g", v);
    for (double v : J0) std::printf(" %.17g", v);
    for (double v : J3) std::printf(" %.17g", v);
    std::printf("\n");
  }

  // ---- tracks that start in frame 0 ----
  const int nP = ri(f);
  std::vector<std::vector<double>> pobs(nP);
  std::vector<int> pn(nP);
  static double para_Feature[1000][1];
  for (int i = 0; i < nP; ++i) {
    pn[i] = ri(f);
    pobs[i].resize(3 * pn[i]);
    for (double& v : pobs[i]) v = rd(f);
    para_Feature[i][0] = rd(f);
  }
  const int nL = ri(f);
  std::vector<std::vector<double>> lobs(nL);
  std::vector<int> ln(nL);
  static double para_LineFeature[1000][4];
  for (int i = 0; i < nL; ++i) {
    ln[i] = ri(f);
    lobs[i].resize(8 * ln[i]);
    for (double& v : lobs[i]) v = rd(f);
    for (double& v : para_LineFeature[i]) v = rd(f);
  }
  // ---- the last prior ----
  static vpl_prior last;
  const int has_prior = ri(f);
  if (has_prior) {
    last.n = ri(f); last.n_blocks = ri(f);
    for (int b = 0; b < last.n_blocks; ++b) { last.block_kind[b] = ri(f); last.block_frame[b] = ri(f); last.block_idx[b] = ri(f); }
    for (int b = 0; b < last.n_blocks; ++b) for (int k = 0; k < 9; ++k) last.x0[b][k] = rd(f);
    for (int k = 0; k < last.n * last.n; ++k) last.J0[k] = rd(f);
    for (int k = 0; k < last.n; ++k) last.r0[k] = rd(f);
  }
  fclose(f);

  // ---- estimator.cpp:1229-1378, MARGIN_OLD ----
  HuberLoss* loss_function = new HuberLoss(opt.huber_delta);
  MarginalizationInfo* marginalization_info = new MarginalizationInfo(opt);
  marginalization_info->setWindowArrays(para_Pose, para_SpeedBias, para_Ex_Pose[0]);
  if (has_prior) {
    std::vector<double*> last_marginalization_parameter_blocks;
    std::vector<int> drop_set;
    for (int b = 0; b < last.n_blocks; ++b) {
      double* p = last.block_kind[b] == VPL_BLOCK_POSE ? para_Pose[last.block_frame[b]]
                  : last.block_kind[b] == VPL_BLOCK_SPEEDBIAS ? para_SpeedBias[last.block_frame[b]] : para_Ex_Pose[0];
      if (p == para_Pose[0] || p == para_SpeedBias[0]) drop_set.push_back(b);
      last_marginalization_parameter_blocks.push_back(p);
    }
    MarginalizationFactor* marginalization_factor = new MarginalizationFactor(&last);
    marginalization_info->addResidualBlockInfo(
        new ResidualBlockInfo(marginalization_factor, NULL, last_marginalization_parameter_blocks, drop_set));
  }
  if (pre.result().sum_dt < 10.0) {
    IMUFactor* imu_factor = new IMUFactor(&pre);
    marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(
        imu_factor, NULL, std::vector<double*>{para_Pose[0], para_SpeedBias[0], para_Pose[1], para_SpeedBias[1]},
        std::vector<int>{0, 1}));
  }
  for (int i = 0; i < nP; ++i)
    for (int j = 1; j < pn[i]; ++j) {
      ProjectionFactor* pf = new ProjectionFactor(&pobs[i][0], &pobs[i][3 * j]);
      marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(
          pf, loss_function, std::vector<double*>{para_Pose[0], para_Pose[j], para_Ex_Pose[0], para_Feature[i]},
          std::vector<int>{0, 3}));
    }
  for (int i = 0; i < nL; ++i)
    for (int j = 1; j < ln[i]; ++j) {       // the start-frame observation is skipped (:1322-1326)
      lineProjectionFactor* lf = new lineProjectionFactor(&lobs[i][8 * j]);
      marginalization_info->addResidualBlockInfo(new ResidualBlockInfo(
          lf, loss_function, std::vector<double*>{para_Pose[j], para_Ex_Pose[0], para_LineFeature[i]}, std::vector<int>{2}));
    }
  marginalization_info->preMarginalize();
  marginalization_info->marginalize();
  std::unordered_map<long, double*> addr_shift;
  for (int i = 1; i <= VPL_WINDOW_SIZE; i++) {
    addr_shift[reinterpret_cast<long>(para_Pose[i])] = para_Pose[i - 1];
    addr_shift[reinterpret_cast<long>(para_SpeedBias[i])] = para_SpeedBias[i - 1];
  }
  addr_shift[reinterpret_cast<long>(para_Ex_Pose[0])] = para_Ex_Pose[0];
  std::vector<double*> parameter_blocks = marginalization_info->getParameterBlocks(addr_shift);

  const int n = marginalization_info->n;
  std::printf("mn %d %d %zu\n", marginalization_info->m, n, parameter_blocks.size());
  std::printf("blocks");
  for (size_t b = 0; b < parameter_blocks.size(); ++b) {
    // which block of the NEXT window each returned address is: kind, frame
    int kind = -1, frame = -1;
    for (int fr = 0; fr < VPL_NFRAMES; ++fr) {
      if (parameter_blocks[b] == para_Pose[fr]) { kind = 0; frame = fr; }
      if (parameter_blocks[b] == para_SpeedBias[fr]) { kind = 1; frame = fr; }
    }
    if (parameter_blocks[b] == para_Ex_Pose[0]) { kind = 2; frame = 0; }
    std::printf(" %d %d %d %d", kind, frame, marginalization_info->keep_block_size[b], marginalization_info->keep_block_idx[b]);
  }
  std::printf("\nJ0");
  for (double v : marginalization_info->linearized_jacobians) std::printf(" %.17g", v);
  std::printf("\nr0");
  for (double v : marginalization_info->linearized_residuals) std::printf(" %.17g", v);
  std::printf("\nx0");
  for (size_t b = 0; b < parameter_blocks.size(); ++b)
    for (int k = 0; k < marginalization_info->keep_block_size[b]; ++k) std::printf(" %.17g", marginalization_info->keep_block_data[b][k]);
  // the new prior as a cost function, evaluated at its own linearisation point: residual = r0, Jacobian blocks = J0 columns
  {
    MarginalizationFactor nf = make_marginalization_factor(marginalization_info);
    std::vector<const double*> p;
    for (size_t b = 0; b < parameter_blocks.size(); ++b) p.push_back(marginalization_info->keep_block_data[b]);
    std::vector<double> r(n);
    nf.Evaluate(p.data(), r.data(), nullptr);
    std::printf("\nprior_at_x0");
    for (double v : r) std::printf(" %.17g", v);
  }
  std::printf("\n");
  delete marginalization_info;   // deletes the factors and their cost functions, not the loss
  delete loss_function;
  return 0;
}
