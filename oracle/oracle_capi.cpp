// TEST INFRASTRUCTURE -- CPU oracle (see smallmat.h).  extern "C" entry points
// used by tests/ (ctypes) and by bench.py's cpu_baseline leg only.
#include <cstring>
#include <thread>
#include <vector>
#include "../include/vplines_ba.h"
#include "marginalization.h"
#include "problem.h"

namespace orc {
extern double g_test_tolerance_scale;
int triangulate_lines(vpl_window* w, const vpl_ba_options* opt);
int triangulate_points(vpl_window* w, const vpl_ba_options* opt, double init_depth);
int slide_window(vpl_window* w, const vpl_ba_options* opt, int marginalization_flag, double init_depth, vpl_slide_tracks* out);
int only_line_opt(vpl_window* w, const vpl_ba_options* opt, vpl_solve_report* rep);
int solve_window(vpl_window* w, const vpl_ba_options* opt, vpl_prior* prior_out, vpl_solve_report* rep,
                 double* A_final_out, double* b_final_out);
}
using namespace orc;

static void fill_pre(const IntegrationBase& ib, vpl_preintegration* o) {
  o->sum_dt = ib.sum_dt;
  for (int k = 0; k < 3; ++k) {
    o->delta_p[k] = ib.delta_p[k];
    o->delta_v[k] = ib.delta_v[k];
    o->linearized_ba[k] = ib.linearized_ba[k];
    o->linearized_bg[k] = ib.linearized_bg[k];
  }
  o->delta_q[0] = ib.delta_q.x; o->delta_q[1] = ib.delta_q.y; o->delta_q[2] = ib.delta_q.z; o->delta_q[3] = ib.delta_q.w;
  for (int k = 0; k < 225; ++k) { o->jacobian[k] = ib.jacobian.a[k]; o->covariance[k] = ib.covariance.a[k]; }
}
static IntegrationBase* make_pre(const vpl_preintegration& p) {
  ImuNoise nz{0, 0, 0, 0};
  IntegrationBase* ib = new IntegrationBase(Vec3{}, Vec3{}, Vec3{p.linearized_ba[0], p.linearized_ba[1], p.linearized_ba[2]},
                                            Vec3{p.linearized_bg[0], p.linearized_bg[1], p.linearized_bg[2]}, nz);
  ib->sum_dt = p.sum_dt;
  ib->delta_p = Vec3{p.delta_p[0], p.delta_p[1], p.delta_p[2]};
  ib->delta_q = Quat(p.delta_q[3], p.delta_q[0], p.delta_q[1], p.delta_q[2]);
  ib->delta_v = Vec3{p.delta_v[0], p.delta_v[1], p.delta_v[2]};
  for (int k = 0; k < 225; ++k) { ib->jacobian.a[k] = p.jacobian[k]; ib->covariance.a[k] = p.covariance[k]; }
  return ib;
}

extern "C" {

// IntegrationBase ctor + push_back per sample (estimator.cpp:82-117 call pattern)
int orc_preintegrate(int nsamples, const double* samples, const double* acc0, const double* gyr0, const double* ba,
                     const double* bg, const vpl_ba_options* opt, vpl_preintegration* out) {
  ImuNoise nz{opt->acc_n, opt->gyr_n, opt->acc_w, opt->gyr_w};
  IntegrationBase ib(Vec3{acc0[0], acc0[1], acc0[2]}, Vec3{gyr0[0], gyr0[1], gyr0[2]}, Vec3{ba[0], ba[1], ba[2]},
                     Vec3{bg[0], bg[1], bg[2]}, nz);
  for (int s = 0; s < nsamples; ++s) {
    const double* p = samples + 7 * s;
    ib.push_back(p[0], Vec3{p[1], p[2], p[3]}, Vec3{p[4], p[5], p[6]});
  }
  fill_pre(ib, out);
  return 0;
}

int orc_projection_factor(const double* params, const double* pts, double sqrt_info, double* res, double* jac) {
  ProjectionFactor::sqrt_info = sqrt_info;
  ProjectionFactor f(Vec3{pts[0], pts[1], pts[2]}, Vec3{pts[3], pts[4], pts[5]});
  const double* P[4] = {params, params + 7, params + 14, params + 21};
  double* J[4] = {jac, jac ? jac + 14 : nullptr, jac ? jac + 28 : nullptr, jac ? jac + 42 : nullptr};
  f.Evaluate(P, res, jac ? J : nullptr);
  return 0;
}
int orc_line_factor(const double* params, const double* obs, double sqrt_info, double* res, double* jac) {
  lineProjectionFactor::sqrt_info = sqrt_info;
  lineProjectionFactor f(Vec4{obs[0], obs[1], obs[2], obs[3]});
  const double* P[3] = {params, params + 7, params + 14};
  double* J[3] = {jac, jac ? jac + 14 : nullptr, jac ? jac + 28 : nullptr};
  f.Evaluate(P, res, jac ? J : nullptr);
  return 0;
}
int orc_vp_factor(const double* params, const double* vp, double sqrt_info, double* res, double* jac) {
  vpProjectionFactor::sqrt_info = sqrt_info;
  vpProjectionFactor f(Vec3{vp[0], vp[1], vp[2]});
  const double* P[3] = {params, params + 7, params + 14};
  double* J[3] = {jac, jac ? jac + 14 : nullptr, jac ? jac + 28 : nullptr};
  f.Evaluate(P, res, jac ? J : nullptr);
  return 0;
}
int orc_imu_factor(const double* params, const vpl_preintegration* pre, double g_norm, double* res, double* jac) {
  IntegrationBase* ib = make_pre(*pre);
  IMUFactor f(ib, Vec3{0, 0, g_norm});
  const double* P[4] = {params, params + 7, params + 16, params + 23};
  double* J[4] = {jac, jac ? jac + 105 : nullptr, jac ? jac + 240 : nullptr, jac ? jac + 345 : nullptr};
  f.Evaluate(P, res, jac ? J : nullptr);
  delete ib;
  return 0;
}
int orc_prior_factor(const vpl_prior* pr, const double* params, double* res, double* jac) {
  MarginalizationInfo mi;
  mi.n = pr->n;
  mi.m = 0;
  mi.linearized_jacobians.resize(pr->n, pr->n);
  for (int r = 0; r < pr->n; ++r)
    for (int c = 0; c < pr->n; ++c) mi.linearized_jacobians(r, c) = pr->J0[(size_t)r * pr->n + c];
  mi.linearized_residuals.assign(pr->r0, pr->r0 + pr->n);
  mi.owned_keep_data.resize(pr->n_blocks);
  std::vector<const double*> P;
  std::vector<double*> J;
  const double* pp = params;
  double* jp = jac;
  for (int b = 0; b < pr->n_blocks; ++b) {
    int size = pr->block_kind[b] == VPL_BLOCK_SPEEDBIAS ? 9 : 7;
    mi.keep_block_size.push_back(size);
    mi.keep_block_idx.push_back(pr->block_idx[b]);
    mi.owned_keep_data[b].assign(pr->x0[b], pr->x0[b] + size);
    P.push_back(pp);
    pp += size;
    J.push_back(jp);
    if (jp) jp += (size_t)pr->n * size;
  }
  for (int b = 0; b < pr->n_blocks; ++b) mi.keep_block_data.push_back(mi.owned_keep_data[b].data());
  MarginalizationFactor f(&mi);
  f.Evaluate(P.data(), res, jac ? J.data() : nullptr);
  return 0;
}
int orc_pose_plus(const double* x, const double* delta, double* out) {
  PoseLocalParameterization p;
  p.Plus(x, delta, out);
  return 0;
}
int orc_line_orth_plus(const double* x, const double* delta, double* out) {
  LineOrthParameterization p;
  p.Plus(x, delta, out);
  return 0;
}
int orc_orth_to_plk(const double* orth, double* plk) {
  Vec6 p = orth_to_plk(Vec4{orth[0], orth[1], orth[2], orth[3]});
  for (int i = 0; i < 6; ++i) plk[i] = p[i];
  return 0;
}
int orc_plk_to_orth(const double* plk, double* orth) {
  Vec4 o = plk_to_orth(Vec6{plk[0], plk[1], plk[2], plk[3], plk[4], plk[5]});
  for (int i = 0; i < 4; ++i) orth[i] = o[i];
  return 0;
}

int orc_solve_window(vpl_window* w, const vpl_ba_options* opt, vpl_prior* prior_out, vpl_solve_report* rep,
                     double* A_final, double* b_final) {
  return solve_window(w, opt, prior_out, rep, A_final, b_final);
}

int orc_triangulate_lines(vpl_window* w, const vpl_ba_options* opt) { return triangulate_lines(w, opt); }
int orc_triangulate_points(vpl_window* w, const vpl_ba_options* opt, double init_depth) { return triangulate_points(w, opt, init_depth); }
int orc_slide_window(vpl_window* w, const vpl_ba_options* opt, int flag, double init_depth, vpl_slide_tracks* out) {
  return slide_window(w, opt, flag, init_depth, out);
}
int orc_only_line_opt(vpl_window* w, const vpl_ba_options* opt, vpl_solve_report* rep) { return only_line_opt(w, opt, rep); }

// threads of MarginalizationInfo's A, b assembly (the reference runs 4, marginalization_factor.h:13)
void orc_set_tolerance_scale(double sc) { orc::g_test_tolerance_scale = sc > 0.0 ? sc : 0.0; }   // test hook, see window.cpp
void orc_set_marg_threads(int n) { g_marg_threads = n < 1 ? 1 : n; }
// test switch: relative truncation of the kept block's spectrum (0 = the reference's absolute 1e-8), see marginalization.cpp
void orc_set_marg_truncation(double rel) { g_marg_rel_eps = rel > 0.0 ? rel : 0.0; }

// test switch: summation order of the last Schur complement of the marginalisation, see marginalization.cpp
void orc_set_marg_reverse_sums(int mode) { g_marg_reverse_sums = mode; }

// windows fanned over `threads` host threads (cpu_baseline leg of bench.py)
int orc_solve_windows(int n, vpl_window* w, const vpl_ba_options* opt, vpl_prior* priors_out, vpl_solve_report* reps,
                      int threads) {
  if (threads <= 1) {
    for (int i = 0; i < n; ++i) solve_window(&w[i], opt, priors_out ? &priors_out[i] : nullptr, reps ? &reps[i] : nullptr, nullptr, nullptr);
    return 0;
  }
  // NOTE: the static sqrt_info members are set to the same value by every thread
  std::vector<std::thread> th;
  for (int t = 0; t < threads; ++t)
    th.emplace_back([=]() {
      for (int i = t; i < n; i += threads)
        solve_window(&w[i], opt, priors_out ? &priors_out[i] : nullptr, reps ? &reps[i] : nullptr, nullptr, nullptr);
    });
  for (auto& t : th) t.join();
  return 0;
}
}
