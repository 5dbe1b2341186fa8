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
// column of vis dim c (0..71) in the compact W row of a track that starts in frame s; -1: structurally zero
__host__ __device__ inline int wcol(int c, int s, int WS) {
  if (c >= 66) return WS - 72 + c;
  const int cc = c - 6 * s;
  return (cc >= 0 && cc < WS - 6) ? cc : -1;
}
constexpr int NCP = NC * (NC + 1) / 2; // packed lower triangle of the cam Hessian (14706)
constexpr int MAXPB = 23;              // prior blocks
constexpr int MAXPN = 171;             // prior dim
constexpr int ACT_SLOTS = 64;             // launches of one solve whose activity is counted
constexpr int SK_WSTRIDE = 64;         // ints per wave in sk_wave: chunk count, then up to 21 x (first entry, end, ticket)
// non-zeros of the packed cam Hessian on the three-kernel path: 72 x 73 / 2 + 11 x 99 + 10 x 189 + 60 x 9
constexpr int NZ_N = 2628 + 1089 + 1890 + 540;
constexpr int SACC_N = 74 * 75 / 2;     // packed lower triangle of the compact Schur product: 72 vis dims, rhs row, Cauchy row
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
  // world Pluecker coordinates (n, v) of orth / orth_c, [W][maxL][6]: written wherever the orthonormal parameters are (k_prep,
  // the candidate of k_back / k_solve, k_cost's copy on acceptance) -- the factors of a line's <= 11 observations, in k_lin
  // and in k_cost, read them instead of evaluating eight sin / cos each
  double *lw, *lw_c;
  // k_lin2 (two work-groups per window): the point work-group's part of the visual Hessian | gradient [W][HV + NV], of the
  // cost [W], and the flag that hands them over [W] (0 between launches)
  double *lin_part, *lin_pcost;
  int* lin_flag;
  int ncu;                                       // compute units of the device (k_lin2 splits a window over two of them while they last)
  double *pose_0, *sb_0, *ex_0, *invd_0, *plk_0; // as uploaded (vpl_ba_reset_state)
  double *plk;                                   // [W][maxL][6] start-camera-frame Pluecker (in/out)
  double *gauge;                                 // [W][4]: yaw of R0 before (deg), P0 before
  int *orth_in;                                  // [W] 1 = B.orth was uploaded (vpl_window.line_orth): k_prep leaves it alone
  double *fail_ref;                              // [W][13]: failure_occur flag, last_P0, last_R0 (row-major) -- the gauge reference
                                                 // of double2vector2 after a failure (estimator.cpp:818-823)

  // ---- tracks ----
  int *nP, *nL;                                  // [W]
  int *pt_start, *pt_nobs, *pt_off;              // [W][maxP]
  double *pt_obs;                                // [W][maxPO][3]
  int *ps_list, *ps_cnt;                         // [W][maxP] track ids sorted by start frame ; [W][12] prefix offsets
  // point phase of k_lin: rounds of 512 lanes = 32 quarter-wave slots; a slot holds up to 8 work units (start frame, k,
  // tracks) packed on even lane boundaries.  pu_lane: (track | k << 16 | s << 20, observation offset) of every lane (-1: idle).  pu_sub: per slot 8 x
  // (descriptor s | j << 4 | ks0 << 8 | ks1 << 12 | 1 << 16, commit ticket), descriptor 0 ends the list.  All units of a
  // chunk of tracks sit in the same half of the work-group (waves 0..3 / 4..7).
  int *pu_lane, *pu_sub;                         // [W][maxPR][512][2] ; [W][maxPR][32][8][2]
  int *pu_cnt, *pu_cnt0;                         // [W] rounds ; [W] rounds that hold the units of start frame 0 (they come first)
  int maxPR;
  int prhN;                                      // largest prior whose J0^T J0 k_lin keeps in LDS (lin_prh_n)
  int *ln_start, *ln_nobs, *ln_off;              // [W][maxL]
  double *ln_obs;                                // [W][maxLO][8]
  int *nLO, *lo_ln;                              // [W] line observation count ; [W][maxLO] observation -> line
  // lane layout of the line phase of k_lin: every wave holds whole tracks, k-major.  Slot pass * 512 + wave * 64 + k * llNLW + i
  // = observation k of the i-th line of that wave (index into ln_obs) or -1; lines sorted by start frame.  llK = longest
  // line track of the batch, llNLW = 64 / llK lines per wave, ll_np[w] passes of 8 waves.
  int *ll_tab, *ll_np;                           // [W][llSlots][2] (observation, line | k << 16 | start << 20) ; [W]
  int llNLW, llK, llSlots;
  int *ln_removed;                               // [W][maxL] 1 = erased by removeLineOutlier (k_gauge)
  int *ln_tri;                                   // [W][maxL] lineFeaturePerId::is_triangulation (k_triangulate)

  DevPreint *pre;                                // [W][11]

  // ---- prior in ----
  int *pr_n, *pr_nb;                             // [W]
  int *pr_kind, *pr_frame, *pr_idx;              // [W][23]
  double *pr_x0;                                 // [W][23][9]
  double *pr_J0, *pr_r0;                         // [W][171*171] [W][171]
  double *pr_H;                                  // [W][171*171]  J0^T J0 (prior-local indexing)
  double *pr_g0;                                 // [W][171]      J0^T r0
  int prS;                                       // stride (doubles) of one window's J0 / H: the batch's largest prior squared,
                                                 // not 171^2 -- the priors of a batch stay within a few MB instead of 120 MB
  int *pr_map;                                   // [W][171] prior-local column -> cam index

  // ---- linearisation (one buffer set; written by k_lin at the current x) ----
  double *Hcc, *gc;                              // [W][NCP] [W][NC]
  int *asm_tab;                                  // [NCP][2] static: sources of every packed cam-Hessian entry, see lin_asm_entry()
  double *Hpp, *gp, *Wp;                         // [W][maxP] [W][maxP] [W][maxP][WS]
  double *Hll, *gl, *Wl;                         // [W][maxL][16] [W][maxL][4] [W][maxL][4][WS]
  // W rows are stored compact: the 6 x 6 blocks of the frames start .. start + maxTrack - 1 of the track, then the 6
  // extrinsic columns: WS = 6 * maxTrack + 6 doubles (42 for 6-frame tracks, 72 = NV at most).  wfill: rows have slots no
  // factor writes (shorter tracks, erased lines) and must be zeroed before the factors are accumulated.
  int WS, wfill;
  // ---- launch order of the trust-region iterations ----
  // A window whose last step was rejected neither re-linearises nor re-factors; one whose step was accepted does both
  // (~150 us of one CU).  Work-groups are dispatched in blockIdx order, so k_cost sorts the windows for the next iteration:
  // those that will do the heavy work first.  order[it & 1][i] = window of work-group i in iteration `it`; ord_cnt[it & 1]
  // = {filled from the front, filled from the back}.  ord_it: iteration of this launch (0 = identity order).
  int *order;                                    // [2][W]
  int *ord_cnt;                                  // [2][2]
  int ord_it;
  double *lchol;                                 // [W][maxL][10] Cholesky factors of the regularised line blocks

  // ---- the trust-region step as three kernels (ba_step.h): k_schur -> k_chol -> k_back ----
  // k_schur: landmark elimination.  The rows of X = C^-1 S [W | g | e] are consumed straight from HBM by the matrix cores,
  // one wave per span of K-steps (4 rows each), in the COMPACT coordinates of the rows' start frame (WS + 2 columns: the
  // frames start .. start + maxTrack - 1, extrinsic, g, e); a wave adds its product to the window's compact 74 x 74 system
  // when the start frame changes.  sk_tab: K-steps sorted by group (points by start frame, then lines by start frame):
  // {group, id0 | id1 << 16, id2 | id3 << 16, 0} for four point rows (0xffff: none) or {group | 32, line, 0, 0} for the four
  // rows of one line.  sk_wave[wave]: {number of chunks, then (first entry, end, ticket) per chunk} -- see ba_pack.h; the adds
  // into the shared system are committed in ticket order so that every sum has a fixed order of terms.
  int *sk_tab, *sk_wave;                         // [W][maxKS][4] ; [W][8][SK_WSTRIDE]
  int maxKS;
  double *sacc;                                  // [W][SACC_N] packed lower triangle of the compact Schur product (rows 0..73)
  double *ycs;                                   // [W][176] S_c y_c: the camera part of the Gauss-Newton step, unscaled (k_chol -> k_back)
  double *sx;                                    // [W][8] scalars handed from kernel to kernel: a2, a3 of the camera dims
  // 0: the three-kernel path; 1: general path for the whole solve (the prior holds a speed/bias block other than frame 0's,
  // which k_chol's elimination order does not cover); 2: general path for this iteration (a factorisation failed: the retry
  // with a larger mu runs in k_solve).  k_solve (ba_solve.h) is that general path.
  int *path;                                     // [W]
  // Entries of the packed camera Hessian that the factors of a fast-path window can fill (vis x vis, the 15-dim frame blocks
  // and their sub-diagonal neighbours, speed/bias 0 against everything: 6147 of 14706), as idx | r << 14 | c << 22: the
  // passes of k_schur that need every non-zero of Hcc walk this list instead of the whole triangle.
  int *nz_tab;                                   // [NZ_N]

  // ---- trust region vectors over the full index ----
  TrState *tr;                                   // [W]
  double *scale, *diag, *grad, *gn;              // [W][nfull]

  // ---- marginalisation ----
  int *mg_n, *mg_nb;                             // [W] new prior size / blocks
  int *mg_kind, *mg_frame, *mg_idx, *mg_cam;     // [W][23] kept blocks (frame = index in the NEXT window), cam base
  double *mg_x0;                                 // [W][23][9]
  double *mg_J0, *mg_r0;                         // [W][MAXKEEP*MAXKEEP] [W][MAXKEEP]
  double *mg_A, *mg_b;                           // [W][MAXKEEP*MAXKEEP] [W][MAXKEEP]  (invariant check: A, b before the eig)
  int *mg_m;                                     // [W] MarginalizationInfo::m
  long long *dbg;                                // [W][64] phase stamps (diagnostic builds: -DVPL_STAMPS)
  // ---- activity of the kernel launches of one solve (bench.py prices a launch by the windows that did work in it) ----
  int *act;                                      // [ACT_SLOTS][4]: windows that ran k_lin, k_solve (new step), k_solve (re-used
                                                 // Gauss-Newton step), k_cost in launch `launch`; null = not counted
  int launch;                                    // slot of this launch (set by the host per launch)
};


// ---- kept-block tables of the marginalisation -------------------------------------------------------------
// The reference orders the kept parameter blocks by address (std::map<long, ...>, marginalization_factor.cpp:133-175),
// i.e. para_Pose[0..10] < para_SpeedBias[0..10] < para_Ex_Pose (estimator.h member order); the same canonical order is
// produced here from which blocks the added factors touch.  Shared by the host (upload) and the device (after
// removeLineOutlier changed the set of line factors).
struct KeepSrc {
  int nP; const int *pt_start, *pt_nobs;
  int nL; const int *ln_start, *ln_nobs, *ln_removed;   // ln_removed may be null
  int pr_nb; const int *pr_kind, *pr_frame;
  bool imu01;                                           // pre_integrations[1]->sum_dt < 10 (estimator.cpp:1261)
};

// MARGIN_OLD (estimator.cpp:1229-1378): drops pose_0 / speed-bias_0 and every landmark that starts in frame 0
__host__ __device__ inline void keep_tables_old(const KeepSrc& S, int* kind, int* frame, int* idx, int* cam, int* n_out,
                                                int* nb_out, int* m_out) {
  bool pose_t[NF], sb_t[NF];
  for (int f = 0; f < NF; ++f) { pose_t[f] = false; sb_t[f] = false; }
  bool ex_t = false, frame0 = false;
  int m = 0;
  for (int p = 0; p < S.nP; ++p)
    if (S.pt_start[p] == 0) { for (int k = 1; k < S.pt_nobs[p]; ++k) pose_t[k] = true; ex_t = frame0 = true; m += 1; }
  for (int l = 0; l < S.nL; ++l)
    if (S.ln_start[l] == 0 && S.ln_nobs[l] >= 2 && !(S.ln_removed && S.ln_removed[l])) {
      for (int k = 1; k < S.ln_nobs[l]; ++k) pose_t[k] = true;
      ex_t = frame0 = true; m += 4;
    }
  for (int b = 0; b < S.pr_nb; ++b) {
    if (S.pr_kind[b] == 0) pose_t[S.pr_frame[b]] = true;
    else if (S.pr_kind[b] == 1) sb_t[S.pr_frame[b]] = true;
    else ex_t = true;
  }
  if (S.imu01) { pose_t[0] = pose_t[1] = true; sb_t[0] = sb_t[1] = true; }
  int nb = 0, n = 0;
  for (int f = 1; f < NF; ++f) if (pose_t[f]) { kind[nb] = 0; frame[nb] = f - 1; idx[nb] = n; cam[nb] = 15 * f; n += 6; ++nb; }
  for (int f = 1; f < NF; ++f) if (sb_t[f]) { kind[nb] = 1; frame[nb] = f - 1; idx[nb] = n; cam[nb] = 15 * f + 6; n += 9; ++nb; }
  if (ex_t) { kind[nb] = 2; frame[nb] = 0; idx[nb] = n; cam[nb] = 165; n += 6; ++nb; }
  *n_out = n; *nb_out = nb;
  *m_out = m + ((pose_t[0] || frame0) ? 6 : 0) + (sb_t[0] ? 9 : 0);
}

// MARGIN_SECOND_NEW (estimator.cpp:1380-1447): only the prior, pose of frame WINDOW_SIZE-1 dropped, frame 10 -> 9.
// Returns 0 when the prior does not hold that pose (the reference then leaves its prior untouched), -1 when it holds
// the speed/bias of frame WINDOW_SIZE-1 (ROS_ASSERT at :1395), 1 otherwise.
__host__ __device__ inline int keep_tables_second_new(int pr_nb, const int* pr_kind, const int* pr_frame, int* kind,
                                                       int* frame, int* idx, int* cam, int* n_out, int* nb_out) {
  bool has = false;
  for (int b = 0; b < pr_nb; ++b) {
    if (pr_kind[b] == 0 && pr_frame[b] == NF - 2) has = true;
    if (pr_kind[b] == 1 && pr_frame[b] == NF - 2) return -1;
  }
  *n_out = 0; *nb_out = 0;
  if (!has) return 0;
  int nb = 0, n = 0;
  for (int k = 0; k < 3; ++k)
    for (int f = 0; f < NF; ++f)
      for (int b = 0; b < pr_nb; ++b) {
        if (pr_kind[b] != k || (k < 2 && pr_frame[b] != f) || (k == 2 && f != 0)) continue;
        if (k == 0 && f == NF - 2) continue;
        kind[nb] = k; frame[nb] = (k < 2 && f == NF - 1) ? NF - 2 : (k == 2 ? 0 : f); idx[nb] = n;
        cam[nb] = k == 0 ? 15 * f : (k == 1 ? 15 * f + 6 : 165);
        n += k == 1 ? 9 : 6; ++nb;
      }
  *n_out = n; *nb_out = nb;
  return 1;
}

}  // namespace vpl
