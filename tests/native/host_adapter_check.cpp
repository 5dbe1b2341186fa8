// GPU test program for the C++ host adapter (vplines-slam_amd/host/vpl_factors.hpp): drives the
// reference-shaped classes through the Ceres-style Evaluate()/Plus() ABI and prints the results so
// that tests/test_gpu_host_adapter.py can compare them with the oracle.
#include <cstdio>
#include "../../vplines-slam_amd/host/vpl_factors.hpp"

using namespace vplhost;

int main() {
  double Pi[7] = {0.1, -0.2, 0.3, 0.01, 0.02, -0.03, 0.9993}, Pj[7] = {0.3, -0.1, 0.35, 0.03, 0.01, -0.02, 0.9993};
  double ex[7] = {-0.02, -0.06, 0.01, 0.0, 0.0, 0.7071067811865476, 0.7071067811865476}, lam[1] = {0.25};
  double pts_i[3] = {0.12, -0.08, 1.0}, pts_j[3] = {0.05, -0.11, 1.0};
  ProjectionFactor f(pts_i, pts_j);
  const double* params[4] = {Pi, Pj, ex, lam};
  double r[2], J0[14], J1[14], J2[14], J3[2];
  double* J[4] = {J0, nullptr, J2, J3};   // a NULL block must be skipped, as ceres allows
  f.Evaluate(params, r, J);
  std::printf("proj %.17g %.17g", r[0], r[1]);
  for (int k = 0; k < 14; ++k) std::printf(" %.17g", J0[k]);
  for (int k = 0; k < 14; ++k) std::printf(" %.17g", J2[k]);
  std::printf(" %.17g %.17g\n", J3[0], J3[1]);
  f.Evaluate(params, r, nullptr);
  std::printf("proj_nojac %.17g %.17g\n", r[0], r[1]);

  double orth[4] = {0.3, -0.2, 1.1, 0.2}, obs[4] = {0.1, 0.2, -0.15, 0.22}, vp[3] = {0.3, -0.2, 0.9};
  const double* lp[3] = {Pi, ex, orth};
  double L0[14], L1[14], L2[8];
  double* LJ[3] = {L0, L1, L2};
  lineProjectionFactor lf(obs);
  lf.Evaluate(lp, r, LJ);
  std::printf("line %.17g %.17g", r[0], r[1]);
  for (int k = 0; k < 8; ++k) std::printf(" %.17g", L2[k]);
  std::printf("\n");
  vpProjectionFactor vf(vp);
  vf.Evaluate(lp, r, LJ);
  std::printf("vp %.17g %.17g", r[0], r[1]);
  for (int k = 0; k < 8; ++k) std::printf(" %.17g", L2[k]);
  std::printf("\n");

  PoseLocalParameterization pl;
  double d6[6] = {0.01, -0.02, 0.03, 0.004, -0.005, 0.006}, xp[7];
  pl.Plus(Pi, d6, xp);
  std::printf("pose_plus");
  for (int k = 0; k < 7; ++k) std::printf(" %.17g", xp[k]);
  std::printf("\n");
  LineOrthParameterization ll;
  double d4[4] = {0.01, -0.02, 0.03, 0.004}, op[4];
  ll.Plus(orth, d4, op);
  std::printf("orth_plus %.17g %.17g %.17g %.17g\n", op[0], op[1], op[2], op[3]);
  std::printf("sizes %d %d %d %d\n", pl.GlobalSize(), pl.LocalSize(), ll.GlobalSize(), ll.LocalSize());
  return 0;
}
