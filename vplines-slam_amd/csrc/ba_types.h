// Device-resident layout of a batch of sliding windows (one vpl_ctx).
//
// Local (tangent) index spaces used by every kernel
//   cam index  c in [0,171): frame f pose -> 15 f + 0..5, frame f speed/bias -> 15 f + 6..14,
//                            extrinsic -> 165..170              (IMU blocks are contiguous 30-wide bands)
//   vis index  v in [0,72) : pose of frame f -> 6 f + 0..5, extrinsic -> 66..71
//                            (the only cam dims visual factors touch)
//   full index k in [0,171 + maxP + 4 maxL): cam | inverse depths | line orth (4 each)
#pragma once
#include <stdint.h>

namespace vpl {

constexpr int NF = 11;                 // WINDOW_SIZE + 1 frames
constexpr int NC = 171;                // cam dims
constexpr int NV = 72;                 // vis dims
constexpr int NCP = NC * (NC + 1) / 2; // packed lower triangle of the cam Hessian (14706)
constexpr int MAXPB = 23;              // prior blocks
constexpr int MAXPN = 171;             // prior dim
constexpr int MAXKEEP = 80;            // new prior dim after MARGIN_OLD: <= 10*6 + 9 + 6 = 75

__host__ __device__ inline int vis2cam(int v) { return v < 66 ? 15 * (v / 6) + (v % 6) : 165 + (v - 66); }
__host__ __device__ inline int tri(int r, int c) { return r * (r + 1) / 2 + c; }  // r >= c

struct DevOptions {
  int num_iterations, estimate_extrinsic, marginalization_flag, remove_line_outliers;
  double sqrt_info_point, sqrt_info_line, sqrt_info_vp, g_norm, huber_delta;
};

// pre-integration as consumed by the IMU factor (+ whitening matrix, computed once per solve:
// the reference recomputes it in every Evaluate, imu_factor.h:68)
struct DevPreint {
  double sum_dt;
  double dp[3], dq[4] /*x,y,z,w*/, dv[3], lba[3], lbg[3];
  double dp_dba[9], dp_dbg[9], dq_dbg[9], dv_dba[9], dv_dbg[9];
  double cov[225];
  double sqrt_info[225];  // upper triangular L^T, row-major
};

// per-window trust-region state (ceres TrustRegionMinimizer + DoglegStrategy members)
struct TrState {
  double radius, mu, x_cost, cand_cost, model_cost_change, x_norm, step_norm, dogleg_step_norm;
  double alpha, a1 /*|g~|^2*/, a2 /*|gn|^2*/, a3 /*g~.gn*/, initial_cost;
  int iter;            // iterations performed so far (0 after the initial evaluation)
  int status;          // 0 running, 1 convergence, 2 failure, 3 max iterations
  int reuse;           // DoglegStrategy::reuse_
  int step_valid;      // the candidate written by k_solve is to be evaluated by k_cost
  int fresh_lin;       // linearisation buffers correspond to the current x
  int num_successful, num_invalid;
  int pad;
};

struct DevBatch {
  int nW;
  int maxP, maxPO, maxL, maxLO;
  int nfull;  // NC + maxP + 4 maxL
  DevOptions opt;

  // ---- states: current x, candidate, uploaded initial copy ----
  double *pose, *sb, *ex, *invd, *orth;          // [W][11][7] [W][11][9] [W][7] [W][maxP] [W][maxL][4]
  double *pose_c, *sb_c, *ex_c, *invd_c, *orth_c;
  double *pose_0, *sb_0, *ex_0, *invd_0, *plk_0; // as uploaded (vpl_ba_reset_state)
  double *plk;                                   // [W][maxL][6] start-camera-frame Pluecker (in/out)
  double *gauge;                                 // [W][4]: yaw of R0 before (deg), P0 before

  // ---- tracks ----
  int *nP, *nL;                                  // [W]
  int *pt_start, *pt_nobs, *pt_off;              // [W][maxP]
  double *pt_obs;                                // [W][maxPO][3]
  int *ps_list, *ps_cnt;                         // [W][maxP] track ids sorted by start frame ; [W][12] prefix offsets
  int *ln_start, *ln_nobs, *ln_off;              // [W][maxL]
  double *ln_obs;                                // [W][maxLO][8]
  int *nLO, *lo_ln;                              // [W] line observation count ; [W][maxLO] observation -> line

  DevPreint *pre;                                // [W][11]

  // ---- prior in ----
  int *pr_n, *pr_nb;                             // [W]
  int *pr_kind, *pr_frame, *pr_idx;              // [W][23]
  double *pr_x0;                                 // [W][23][9]
  double *pr_J0, *pr_r0;                         // [W][171*171] [W][171]
  double *pr_H;                                  // [W][171*171]  J0^T J0 (prior-local indexing)
  int *pr_map;                                   // [W][171] prior-local column -> cam index

  // ---- linearisation (one buffer set; written by k_lin at the current x) ----
  double *Hcc, *gc;                              // [W][NCP] [W][NC]
  double *Hpp, *gp, *Wp;                         // [W][maxP] [W][maxP] [W][maxP][NV]
  double *Hll, *gl, *Wl;                         // [W][maxL][16] [W][maxL][4] [W][maxL][4][NV]
  double *lchol;                                 // [W][maxL][10] Cholesky factors of the regularised line blocks

  // ---- trust region vectors over the full index ----
  TrState *tr;                                   // [W]
  double *scale, *diag, *grad, *gn, *delta;      // [W][nfull]

  // ---- marginalisation ----
  int *mg_n, *mg_nb;                             // [W] new prior size / blocks
  int *mg_kind, *mg_frame, *mg_idx, *mg_cam;     // [W][23] kept blocks (frame = index in the NEXT window), cam base
  double *mg_x0;                                 // [W][23][9]
  double *mg_J0, *mg_r0;                         // [W][MAXKEEP*MAXKEEP] [W][MAXKEEP]
  double *mg_A, *mg_b;                           // [W][MAXKEEP*MAXKEEP] [W][MAXKEEP]  (invariant check: A, b before the eig)
  int *mg_m;                                     // [W] MarginalizationInfo::m
  long long *dbg;                                // [W][64] phase stamps (diagnostic builds: -DVPL_STAMPS)
};

}  // namespace vpl
