// TEST-ONLY: compiles the DEVICE math headers (vplines-slam_amd/csrc/vpl_math.h,
// vpl_preint.h) with the host compiler so that the formulas the HIP kernels use can
// be checked against the oracle on a machine without a GPU.  This file is never
// linked into the product library; the product has no CPU path.
#include <cstring>
#include "../../include/vplines_ba.h"
#include "../../vplines-slam_amd/csrc/vpl_math.h"
#include "../../vplines-slam_amd/csrc/vpl_preint.h"

using namespace vpl;

static void expand6to7(const double* J6, double* J7) {
  for (int r = 0; r < 2; ++r) {
    for (int c = 0; c < 6; ++c) J7[7 * r + c] = J6[6 * r + c];
    J7[7 * r + 6] = 0.0;
  }
}

extern "C" {
int hc_projection_factor(const double* params, const double* pts, double sqrt_info, double* res, double* jac) {
  double Ji[12], Jj[12], Je[12], Jl[2];
  projection_factor(params, params + 7, params + 14, params[21], V3{pts[0], pts[1], pts[2]}, V3{pts[3], pts[4], pts[5]},
                    sqrt_info, res, jac != nullptr, Ji, Jj, Je, Jl);
  if (jac) {
    expand6to7(Ji, jac);
    expand6to7(Jj, jac + 14);
    expand6to7(Je, jac + 28);
    jac[42] = Jl[0];
    jac[43] = Jl[1];
  }
  return 0;
}
int hc_line_factor(const double* params, const double* obs, double sqrt_info, double* res, double* jac) {
  LineCtx c = line_ctx(params, params + 7, params + 14);
  double jel[6], Jp[12], Je[12], Jo[8];
  line_factor_res(c, obs, sqrt_info, res, jel);
  if (jac) {
    line_chain_jac(c, jel, 0, Jp, Je, Jo);
    expand6to7(Jp, jac);
    expand6to7(Je, jac + 14);
    for (int k = 0; k < 8; ++k) jac[28 + k] = Jo[k];
  }
  return 0;
}
int hc_vp_factor(const double* params, const double* vp, double sqrt_info, double* res, double* jac) {
  LineCtx c = line_ctx(params, params + 7, params + 14);
  double jel[6], Jp[12], Je[12], Jo[8];
  vp_factor_res(c, vp, sqrt_info, res, jel);
  if (jac) {
    line_chain_jac(c, jel, 1, Jp, Je, Jo);
    expand6to7(Jp, jac);
    expand6to7(Je, jac + 14);
    for (int k = 0; k < 8; ++k) jac[28 + k] = Jo[k];
  }
  return 0;
}
static PreInt to_pre(const vpl_preintegration* p) {
  PreInt o;
  o.sum_dt = p->sum_dt;
  o.dp = V3{p->delta_p[0], p->delta_p[1], p->delta_p[2]};
  o.dv = V3{p->delta_v[0], p->delta_v[1], p->delta_v[2]};
  o.dq = Q4{p->delta_q[3], p->delta_q[0], p->delta_q[1], p->delta_q[2]};
  o.lba = V3{p->linearized_ba[0], p->linearized_ba[1], p->linearized_ba[2]};
  o.lbg = V3{p->linearized_bg[0], p->linearized_bg[1], p->linearized_bg[2]};
  auto blk = [&](int r0, int c0) {
    M3 B;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) B.m[3 * i + j] = p->jacobian[(r0 + i) * 15 + c0 + j];
    return B;
  };
  o.dp_dba = blk(0, 9); o.dp_dbg = blk(0, 12); o.dq_dbg = blk(3, 12); o.dv_dba = blk(6, 9); o.dv_dbg = blk(6, 12);
  return o;
}
// raw (un-whitened) residual and Jacobian of the IMU factor; whitening is checked on the GPU
int hc_imu_factor_raw(const double* params, const vpl_preintegration* pre, double g_norm, double* res, double* jac30) {
  PreInt p = to_pre(pre);
  imu_residual_raw(p, params, params + 7, params + 16, params + 23, g_norm, res);
  if (jac30) {
    std::memset(jac30, 0, sizeof(double) * 450);
    ImuJac B = imu_jacobian_raw(p, params, params + 7, params + 16, params + 23, g_norm);
    imu_jac_dense(B, jac30);
  }
  return 0;
}
int hc_pose_plus(const double* x, const double* d, double* out) { pose_plus(x, d, out); return 0; }
int hc_line_orth_plus(const double* x, const double* d, double* out) { line_orth_plus(x, d, out); return 0; }
int hc_preintegrate(int nsamples, const double* samples, const double* acc0, const double* gyr0, const double* ba,
                    const double* bg, const vpl_ba_options* opt, vpl_preintegration* out) {
  PreintState s;
  s.dp = V3{0, 0, 0}; s.dv = V3{0, 0, 0}; s.dq = Q4{1, 0, 0, 0}; s.sum_dt = 0;
  for (int k = 0; k < 225; ++k) { s.J[k] = (k / 15 == k % 15) ? 1.0 : 0.0; s.P[k] = 0.0; }
  double nz2[4] = {opt->acc_n * opt->acc_n, opt->gyr_n * opt->gyr_n, opt->acc_w * opt->acc_w, opt->gyr_w * opt->gyr_w};
  double F[225], V[270], T[225];
  V3 a0{acc0[0], acc0[1], acc0[2]}, g0{gyr0[0], gyr0[1], gyr0[2]};
  V3 vba{ba[0], ba[1], ba[2]}, vbg{bg[0], bg[1], bg[2]};
  for (int i = 0; i < nsamples; ++i) {
    const double* p = samples + 7 * i;
    V3 a1{p[1], p[2], p[3]}, g1{p[4], p[5], p[6]};
    preint_step(s, p[0], a0, g0, a1, g1, vba, vbg, nz2, F, V, T);
    a0 = a1; g0 = g1;
  }
  out->sum_dt = s.sum_dt;
  out->delta_p[0] = s.dp.x; out->delta_p[1] = s.dp.y; out->delta_p[2] = s.dp.z;
  out->delta_v[0] = s.dv.x; out->delta_v[1] = s.dv.y; out->delta_v[2] = s.dv.z;
  out->delta_q[0] = s.dq.x; out->delta_q[1] = s.dq.y; out->delta_q[2] = s.dq.z; out->delta_q[3] = s.dq.w;
  for (int k = 0; k < 3; ++k) { out->linearized_ba[k] = ba[k]; out->linearized_bg[k] = bg[k]; }
  std::memcpy(out->jacobian, s.J, sizeof(s.J));
  std::memcpy(out->covariance, s.P, sizeof(s.P));
  return 0;
}
}
