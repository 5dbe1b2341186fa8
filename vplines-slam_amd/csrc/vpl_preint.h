// IMU mid-point pre-integration, one keyframe interval per call.
// Reproduces IntegrationBase::{push_back,propagate,midPointIntegration}
// (vins_estimator/src/factor/integration_base.h:30-36,54-198): per sample
//   J <- F J,  P <- F P F^T + V Q V^T,  delta_q normalised after every sample.
#pragma once
#include "vpl_math.h"

namespace vpl {

struct PreintState {
  V3 dp, dv;
  Q4 dq;
  double J[225], P[225];
  double sum_dt;
};

VPL_HD void set33(double* A, int ld, int r0, int c0, const M3& B) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) A[(r0 + i) * ld + c0 + j] = B.m[3 * i + j];
}
VPL_HD M3 add(const M3& A, const M3& B) {
  M3 C;
  for (int k = 0; k < 9; ++k) C.m[k] = A.m[k] + B.m[k];
  return C;
}
VPL_HD M3 ident() { return M3{{1, 0, 0, 0, 1, 0, 0, 0, 1}}; }

// one sample; (a0,g0) previous measurement, (a1,g1) new one; ba,bg linearisation biases;
// nz2 = squared noise densities {acc_n^2, gyr_n^2, acc_w^2, gyr_w^2}
VPL_HD void preint_step(PreintState& s, double dt, V3 a0, V3 g0, V3 a1, V3 g1, V3 ba, V3 bg, const double* nz2,
                        double* F /*225 scratch*/, double* V /*270 scratch*/, double* T /*225 scratch*/) {
  V3 un_acc_0 = qrot(s.dq, a0 - ba);
  V3 un_gyr = (g0 + g1) * 0.5 - bg;
  Q4 rq = qmul(s.dq, Q4{1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2});
  V3 un_acc_1 = qrot(rq, a1 - ba);
  V3 un_acc = (un_acc_0 + un_acc_1) * 0.5;
  V3 rp = s.dp + s.dv * dt + un_acc * (0.5 * dt * dt);
  V3 rv = s.dv + un_acc * dt;

  M3 R_w_x = skew(un_gyr), R_a_0_x = skew(a0 - ba), R_a_1_x = skew(a1 - ba);
  M3 Rq = qmat(s.dq), Rr = qmat(rq);
  M3 I3 = ident();
  M3 ImW = add(I3, scale(R_w_x, -dt));
  for (int k = 0; k < 225; ++k) F[k] = 0.0;
  for (int k = 0; k < 270; ++k) V[k] = 0.0;
  M3 RrA1 = mul(Rr, R_a_1_x);
  set33(F, 15, 0, 0, I3);
  set33(F, 15, 0, 3, add(scale(mul(Rq, R_a_0_x), -0.25 * dt * dt), scale(mul(RrA1, ImW), -0.25 * dt * dt)));
  set33(F, 15, 0, 6, scale(I3, dt));
  set33(F, 15, 0, 9, scale(add(Rq, Rr), -0.25 * dt * dt));
  set33(F, 15, 0, 12, scale(RrA1, -0.25 * dt * dt * -dt));
  set33(F, 15, 3, 3, ImW);
  set33(F, 15, 3, 12, scale(I3, -1.0 * dt));
  set33(F, 15, 6, 3, add(scale(mul(Rq, R_a_0_x), -0.5 * dt), scale(mul(RrA1, ImW), -0.5 * dt)));
  set33(F, 15, 6, 6, I3);
  set33(F, 15, 6, 9, scale(add(Rq, Rr), -0.5 * dt));
  set33(F, 15, 6, 12, scale(RrA1, -0.5 * dt * -dt));
  set33(F, 15, 9, 9, I3);
  set33(F, 15, 12, 12, I3);

  M3 V03 = scale(RrA1, -(0.25 * dt * dt * 0.5 * dt));
  M3 V63 = scale(RrA1, -(0.5 * dt * 0.5 * dt));
  set33(V, 18, 0, 0, scale(Rq, 0.25 * dt * dt));
  set33(V, 18, 0, 3, V03);
  set33(V, 18, 0, 6, scale(Rr, 0.25 * dt * dt));
  set33(V, 18, 0, 9, V03);
  set33(V, 18, 3, 3, scale(I3, 0.5 * dt));
  set33(V, 18, 3, 9, scale(I3, 0.5 * dt));
  set33(V, 18, 6, 0, scale(Rq, 0.5 * dt));
  set33(V, 18, 6, 3, V63);
  set33(V, 18, 6, 6, scale(Rr, 0.5 * dt));
  set33(V, 18, 6, 9, V63);
  set33(V, 18, 9, 12, scale(I3, dt));
  set33(V, 18, 12, 15, scale(I3, dt));

  // J <- F J
  for (int i = 0; i < 15; ++i)
    for (int j = 0; j < 15; ++j) {
      double acc = 0;
      for (int k = 0; k < 15; ++k) acc += F[i * 15 + k] * s.J[k * 15 + j];
      T[i * 15 + j] = acc;
    }
  for (int k = 0; k < 225; ++k) s.J[k] = T[k];
  // P <- F P F^T + V Q V^T
  for (int i = 0; i < 15; ++i)
    for (int j = 0; j < 15; ++j) {
      double acc = 0;
      for (int k = 0; k < 15; ++k) acc += F[i * 15 + k] * s.P[k * 15 + j];
      T[i * 15 + j] = acc;
    }
  for (int i = 0; i < 15; ++i)
    for (int j = 0; j < 15; ++j) {
      double acc = 0;
      for (int k = 0; k < 15; ++k) acc += T[i * 15 + k] * F[j * 15 + k];
      double vq = 0;
      for (int k = 0; k < 18; ++k) {
        const double q = nz2[k < 3 ? 0 : k < 6 ? 1 : k < 9 ? 0 : k < 12 ? 1 : k < 15 ? 2 : 3];
        vq += V[i * 18 + k] * q * V[j * 18 + k];
      }
      s.P[i * 15 + j] = acc + vq;
    }
  s.dp = rp;
  s.dv = rv;
  s.dq = qnormalized(rq);
  s.sum_dt += dt;
}

}  // namespace vpl
