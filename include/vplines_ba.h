/*
 * vplines_ba.h -- C ABI of the MI355X-native sliding-window bundle-adjustment path.
 *
 * This header is the drop-in boundary for the back-end hot path of
 * multiplefish/VPLines-SLAM (vins_estimator).  Every entry point names the
 * reference interface it replaces (path:line relative to the reference tree).
 * All signatures use plain pointers and sizes; no C++/torch types cross it.
 *
 * Conventions shared with the reference
 *   - parameter blocks keep the reference's global layouts
 *       pose        [7] = px,py,pz,qx,qy,qz,qw      (estimator.cpp:650-705)
 *       speed/bias  [9] = v, ba, bg
 *       inv. depth  [1]                             (feature_manager.cpp:277-293)
 *       line orth   [4] = psi1,psi2,psi3,phi        (line_geometry.cpp:62-126)
 *   - Jacobians are row-major  num_residuals x global_size, column 6 of every
 *     7-sized pose block is written as zero (projection_factor.cpp:81-88)
 *   - WINDOW_SIZE = 10  => 11 frames  (parameters.h:20)
 *
 * Error handling: every function returns 0 on success and a negative
 * VPL_E_* code otherwise; nothing aborts, nothing falls back to a CPU path.
 * Thread model: one vpl_ctx per host thread / GPU; functions on one context
 * are not re-entrant (the reference calls its solve from exactly one thread,
 * estimator_node.cpp:229).
 */
#ifndef VPLINES_BA_H
#define VPLINES_BA_H

#ifdef __cplusplus
extern "C" {
#endif

#define VPL_WINDOW_SIZE 10
#define VPL_NFRAMES 11
#define VPL_MAX_PRIOR_BLOCKS 23 /* 11 poses + 11 speed/bias + 1 extrinsic */
#define VPL_MAX_PRIOR_DIM 171   /* 11*6 + 11*9 + 6 */

#define VPL_OK 0
#define VPL_E_INVALID -1   /* bad argument / shape mismatch */
#define VPL_E_NODEVICE -2  /* no HIP device or kernel image for it */
#define VPL_E_HIP -3       /* a HIP runtime call failed */
#define VPL_E_CAPACITY -4  /* batch exceeds the capacity given at create */
#define VPL_E_NUMERIC -5   /* non-finite value met where the reference would ROS_BREAK */

/* marginalization_flag of Estimator (estimator.h MarginalizationFlag) */
#define VPL_MARGIN_OLD 0
#define VPL_MARGIN_SECOND_NEW 1
#define VPL_MARGIN_NONE -1

/* kind of a kept parameter block inside a prior */
#define VPL_BLOCK_POSE 0
#define VPL_BLOCK_SPEEDBIAS 1
#define VPL_BLOCK_EXPOSE 2

/* Solver / model constants the reference reads from YAML and parameters.h
 * (parameters.cpp:58-154, estimator.cpp:18-20). */
typedef struct vpl_ba_options {
  int num_iterations;       /* NUM_ITERATIONS -> options.max_num_iterations (estimator.cpp:1211) */
  int estimate_extrinsic;   /* ESTIMATE_EXTRINSIC; 0 => para_Ex_Pose constant (estimator.cpp:1060) */
  int marginalization_flag; /* VPL_MARGIN_* */
  int remove_line_outliers; /* run FeatureManager::removeLineOutlier between solve and marginalisation (0/1) */
  double focal_length;      /* FOCAL_LENGTH; ProjectionFactor::sqrt_info = f/1.5 * I */
  double line_factor;       /* lineProjectionFactor::sqrt_info = line_factor * I */
  double vp_factor;         /* vpProjectionFactor::sqrt_info = vp_factor * I */
  double g_norm;            /* G = (0,0,g_norm) */
  double acc_n, gyr_n, acc_w, gyr_w; /* IMU noise densities (integration_base.h:21-27) */
  double huber_delta;       /* ceres::HuberLoss(1.0) (estimator.cpp:1048) */
} vpl_ba_options;

/* Fills the EuRoC values (config/euroc/euroc_config.yaml:55-63,84-91) with
 * num_iterations = 5 (the BASELINE metric), estimate_extrinsic = 1. */
void vpl_ba_default_options(vpl_ba_options* opt);

/* Result of IntegrationBase (integration_base.h:9-249) for one keyframe interval. */
typedef struct vpl_preintegration {
  double sum_dt;
  double delta_p[3];
  double delta_q[4];          /* x,y,z,w */
  double delta_v[3];
  double linearized_ba[3];
  double linearized_bg[3];
  double jacobian[15 * 15];   /* row-major, state order P,R,V,BA,BG (parameters.h:82-89) */
  double covariance[15 * 15]; /* row-major */
} vpl_preintegration;

/* The linearised prior left by MarginalizationInfo (marginalization_factor.h:46-72):
 * keep_block_{size,idx,data}, linearized_jacobians, linearized_residuals.
 * Blocks are identified by (kind, frame) instead of by address; the order of
 * blocks is deterministic (see DESIGN.md "prior ordering"). */
typedef struct vpl_prior {
  int n;                                   /* number of residuals = kept local dims */
  int n_blocks;
  int block_kind[VPL_MAX_PRIOR_BLOCKS];    /* VPL_BLOCK_* */
  int block_frame[VPL_MAX_PRIOR_BLOCKS];   /* frame index in the window that will USE the prior */
  int block_idx[VPL_MAX_PRIOR_BLOCKS];     /* keep_block_idx - m : first local column */
  double x0[VPL_MAX_PRIOR_BLOCKS][9];      /* keep_block_data (global size 7 or 9) */
  double J0[VPL_MAX_PRIOR_DIM * VPL_MAX_PRIOR_DIM]; /* linearized_jacobians, row-major n x n */
  double r0[VPL_MAX_PRIOR_DIM];            /* linearized_residuals */
} vpl_prior;

/* One sliding window exactly as Estimator::optimizationwithLine() sees it
 * (estimator.cpp:1043-1453): para_* arrays after vector2double(), the feature
 * tracks of FeatureManager that pass the reference's filters, pre-integrations
 * and the last prior.  All arrays are caller-owned host memory. */
typedef struct vpl_window {
  double pose[VPL_NFRAMES][7];       /* para_Pose        (in: initial, out: after double2vector2) */
  double speed_bias[VPL_NFRAMES][9]; /* para_SpeedBias   (in/out) */
  double ex_pose[7];                 /* para_Ex_Pose[0]  (in/out) */

  /* point tracks: FeaturePerId with used_num>=2 && start_frame<WINDOW_SIZE-2
   * (estimator.cpp:1100-1102), in f_manager.feature order */
  int n_points;
  const int* point_start;   /* [n_points] start_frame */
  const int* point_nobs;    /* [n_points] feature_per_frame.size(), consecutive frames from start */
  const double* point_obs;  /* [sum nobs][3]  FeaturePerFrame::point (x,y,1) */
  double* inv_depth;        /* [n_points] para_Feature (in/out) = 1/estimated_depth */

  /* line tracks: used_num>=LINE_MIN_OBS && start_frame<WINDOW_SIZE-2 && is_triangulation
   * (estimator.cpp:1132-1133) */
  int n_lines;
  const int* line_start;    /* [n_lines] */
  const int* line_nobs;     /* [n_lines] */
  const double* line_obs;   /* [sum nobs][8] x1,y1,x2,y2, vp_x,vp_y,vp_z, vp_flag (estimator_node.cpp:375-407) */
  double* line_plk;         /* [n_lines][6] lineFeaturePerId::line_plucker, start-CAMERA frame (in/out) */
  int* line_removed;        /* [n_lines] out, may be NULL: 1 = erased by removeLineOutlier (options.remove_line_outliers) */
  int* line_triangulated;   /* [n_lines] lineFeaturePerId::is_triangulation; NULL = every line is triangulated.  Lines with 0
                             * take no part in the solves (estimator.cpp:1133); vpl_ba_triangulate_lines sets it (in/out) */

  /* pre_integrations[1..10]; entry 0 unused (estimator.cpp:1085-1093) */
  vpl_preintegration preint[VPL_NFRAMES];

  int has_prior;            /* last_marginalization_info != nullptr */
  const vpl_prior* prior;   /* used when has_prior */

  /* Estimator::failure_occur (estimator.cpp:818-823): after a detected failure the gauge fix of double2vector2 restores the
   * yaw / position of last_R0 / last_P0 (the first frame of the last good window) instead of those of the first frame
   * before this solve.  0 = normal operation; last_R0 is row-major 3x3. */
  int failure_occur;
  double last_P0[3];
  double last_R0[9];

  /* Optional: para_LineFeature as the caller holds it -- [n_lines][4] WORLD-frame orthonormal (psi1,psi2,psi3,phi),
   * the block lineProjectionFactor / vpProjectionFactor::Evaluate take (feature_manager.cpp:341-365).  NULL (the normal
   * case): it is derived from line_plk and the start pose, as vector2double does.  Non-NULL: used as is (vpl_ba_marginalize
   * driven factor by factor, host/vpl_factors.hpp MarginalizationInfo); line_plk is then not read. */
  const double* line_orth;
} vpl_window;

/* Per-window solve report (mirrors the fields of ceres::Solver::Summary the
 * reference could log, plus what the marginalisation produced). */
typedef struct vpl_solve_report {
  int iterations;           /* trust-region iterations performed (successful + unsuccessful) */
  int num_successful_steps;
  int termination;          /* 0 no_convergence(max iters), 1 convergence, 2 failure */
  double initial_cost;
  double final_cost;
  int n_lines_removed;      /* by removeLineOutlier */
  int prior_m, prior_n;     /* MarginalizationInfo::m, n */
} vpl_solve_report;

typedef struct vpl_ctx vpl_ctx; /* opaque: device buffers + stream for one GPU */

/* ---- context ------------------------------------------------------------ */
/* Creates a context on HIP device `device` sized for up to `max_windows`
 * windows of up to `max_points` point tracks, `max_point_obs` point
 * observations, `max_lines` / `max_line_obs` likewise per window. */
int vpl_ctx_create(vpl_ctx** out, int device, int max_windows, int max_points, int max_point_obs,
                   int max_lines, int max_line_obs);
void vpl_ctx_destroy(vpl_ctx* ctx);
/* Launch all work of this context on `hip_stream` (a hipStream_t; NULL = default stream).  Work in flight on the stream used
 * so far is completed first (an enqueued call is collected, the old stream synchronised). */
int vpl_ctx_set_stream(vpl_ctx* ctx, void* hip_stream);
const char* vpl_last_error(const vpl_ctx* ctx);

/* ---- IMU pre-integration  (replaces IntegrationBase::push_back/propagate/
 *      midPointIntegration, integration_base.h:30-36,54-198) --------------- */
/* n intervals; interval k has nsamples[k] samples stored at samples + 7*offset[k]
 * as (dt, ax,ay,az, gx,gy,gz); acc0/gyr0/ba/bg are [n][3]. Host in, host out. */
int vpl_preintegrate_batch(vpl_ctx* ctx, int n, const int* offset, const int* nsamples,
                           const double* samples, const double* acc0, const double* gyr0,
                           const double* lin_ba, const double* lin_bg, const vpl_ba_options* opt,
                           vpl_preintegration* out);

/* ---- single-factor batch evaluators ------------------------------------- *
 * Each replaces one CostFunction::Evaluate (signature
 *   bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const )
 * for n factors at once.  Parameter blocks are packed contiguously per factor
 * in the reference's order; residuals [n][R]; jacobians may be NULL, otherwise
 * row-major blocks concatenated per factor in parameter order.              */

/* ProjectionFactor::Evaluate (projection_factor.cpp:26-126), SizedCostFunction<2,7,7,7,1>.
 * params [n][22] = pose_i, pose_j, ex_pose, inv_dep ; pts [n][6] = pts_i, pts_j ;
 * residuals [n][2] ; jac [n][44] = 2x7, 2x7, 2x7, 2x1.  Host pointers. */
int vpl_projection_factor_evaluate(vpl_ctx* ctx, int n, const double* params, const double* pts,
                                   double sqrt_info, double* residuals, double* jac);
/* lineProjectionFactor::Evaluate (line_projection_factor.cpp:251-380), <2,7,7,4>.
 * params [n][18] = pose, ex_pose, orth ; obs [n][4] ; jac [n][36] = 2x7,2x7,2x4. */
int vpl_line_factor_evaluate(vpl_ctx* ctx, int n, const double* params, const double* obs,
                             double sqrt_info, double* residuals, double* jac);
/* vpProjectionFactor::Evaluate (line_projection_factor.cpp:11-153), <2,7,7,4>. obs [n][3]. */
int vpl_vp_factor_evaluate(vpl_ctx* ctx, int n, const double* params, const double* vp,
                           double sqrt_info, double* residuals, double* jac);
/* IMUFactor::Evaluate (imu_factor.h:23-182), <15,7,9,7,9>.
 * params [n][32] = pose_i, sb_i, pose_j, sb_j ; jac [n][480] = 15x7,15x9,15x7,15x9. */
int vpl_imu_factor_evaluate(vpl_ctx* ctx, int n, const double* params, const vpl_preintegration* pre,
                            double g_norm, double* residuals, double* jac);
/* MarginalizationFactor::Evaluate (marginalization_factor.cpp:492-542).
 * params: concatenated kept blocks in prior order (global sizes); residuals [prior->n];
 * jac: for block b a row-major n x global_size matrix, concatenated in block order. */
int vpl_prior_factor_evaluate(vpl_ctx* ctx, const vpl_prior* prior, const double* params,
                              double* residuals, double* jac);

/* PoseLocalParameterization::Plus (pose_local_parameterization.cpp:3-19): x[n][7], delta[n][6]. */
int vpl_pose_plus(vpl_ctx* ctx, int n, const double* x, const double* delta, double* x_plus_delta);
/* LineOrthParameterization::Plus (line_parameterization.cpp:7-93): x[n][4], delta[n][4]. */
int vpl_line_orth_plus(vpl_ctx* ctx, int n, const double* x, const double* delta, double* x_plus_delta);

/* ---- the window solve ---------------------------------------------------- *
 * Replaces the body of Estimator::optimizationwithLine() (estimator.cpp:1043-1453):
 * vector2double -> ceres::Solve(DENSE_SCHUR, DOGLEG, max_num_iterations) ->
 * double2vector2 -> [removeLineOutlier] -> marginalisation -> new prior.
 *
 * Three-phase form so that a bench can keep inputs resident in HBM:
 *   upload  : host windows -> device SoA        (PCIe, not in the timed region)
 *   solve   : everything on device, no host round trip, asynchronous on the stream
 *   download: device -> host windows / priors / reports
 * Refusals (VPL_E_INVALID / VPL_E_CAPACITY, message in vpl_last_error): more windows / points / observations / lines than the
 * context was created for, a track that leaves the window, a point track of one observation, an unknown marginalization_flag,
 * a prior of impossible size.  A refusal that comes after the upload has begun to rewrite the batch leaves the context WITHOUT
 * a batch: solve / download / reset_state return VPL_E_INVALID until the next successful upload.
 * A window with a NaN / inf among its inputs is not refused: its solve fails as ceres fails it (report: termination 2,
 * iterations = num_successful_steps = -1, costs 0; states untouched), the other windows of the batch are unaffected.
 */
int vpl_ba_upload(vpl_ctx* ctx, int n_windows, const vpl_window* windows, const vpl_ba_options* opt);
/* The next windows of a sequence: as vpl_ba_upload, but window i takes the prior that the context's PREVIOUS solve left for
 * window i (MarginalizationInfo of that solve, estimator.cpp:1229-1447 -> last_marginalization_info) -- it stays in HBM,
 * handed from the marginalisation's output to the next solve's input on the device; windows[i].prior / has_prior are
 * ignored.  Needs a previous solve (or vpl_ba_marginalize) of the same batch size with a marginalisation flag other than
 * VPL_MARGIN_NONE and NO upload on the context since (vpl_ba_triangulate_*, vpl_ba_only_line_opt, vpl_ba_slide_window and the
 * stages of vpl_ba_solve_odometry upload too): VPL_E_INVALID otherwise. */
int vpl_ba_upload_chained(vpl_ctx* ctx, int n_windows, const vpl_window* windows, const vpl_ba_options* opt);
int vpl_ba_solve(vpl_ctx* ctx);           /* enqueue one batched solve of the uploaded windows */
int vpl_ba_reset_state(vpl_ctx* ctx);     /* restore the uploaded initial states on device (for repeated timing) */
int vpl_ba_download(vpl_ctx* ctx, int n_windows, vpl_window* windows, vpl_prior* priors_out,
                    vpl_solve_report* reports);
int vpl_ctx_synchronize(vpl_ctx* ctx);
/* Host <-> device traffic of upload / download: the arrays are packed into one PINNED staging arena per context; an upload is
 * one host-to-device copy + one scatter kernel, a download one gather kernel + one device-to-host copy.  No pageable memory
 * (neither the library's nor the caller's) is handed to the HIP runtime on the upload / solve / download path.
 * Leg timing: with it enabled the three calls bracket their device work with hipEvents on the context's stream;
 * vpl_ctx_leg_times returns {upload (copy + scatter), solve (all launches), download (gather + copy)} of the last calls in
 * milliseconds of DEVICE time (-1 for a leg that has not run), to be read next to the wall clock of the calls.  While it is on,
 * vpl_ba_solve launches kernel by kernel (no graph replay). */
int vpl_ctx_enable_leg_timing(vpl_ctx* ctx, int enable);
int vpl_ctx_leg_times(vpl_ctx* ctx, double* ms3);
/* States of the solved batch into a caller-owned DEVICE buffer, [n_windows][183] doubles per window: pose[11][7],
 * speed_bias[11][9], ex_pose[7] -- after double2vector2.  Asynchronous on the context's stream.  This is what a multi-GPU
 * run hands to its collective (RCCL all-gather of the batch's results) without a host staging copy. */
int vpl_ba_pack_states_device(vpl_ctx* ctx, int n_windows, void* d_states);

/* Convenience: upload + solve + synchronize + download. */
int vpl_ba_solve_windows(vpl_ctx* ctx, int n_windows, vpl_window* windows, const vpl_ba_options* opt,
                         vpl_prior* priors_out, vpl_solve_report* reports);

/* ---- marginalisation on its own -------------------------------------------------------------------------- *
 * Replaces MarginalizationInfo::{addResidualBlockInfo, preMarginalize, marginalize, getParameterBlocks}
 * (marginalization_factor.cpp:89-129,177-363,458-478) as Estimator::optimizationwithLine() drives them
 * (estimator.cpp:1229-1378 for VPL_MARGIN_OLD, :1380-1447 for VPL_MARGIN_SECOND_NEW), WITHOUT a solve: the factor subset of
 * the flag (MARGIN_OLD: last prior, IMU factor (0,1), the point and line factors of the tracks that start in frame 0; no VP
 * factors -- MARGIN_SECOND_NEW: the last prior alone) is linearised at the windows' CURRENT states, landmarks and dropped
 * blocks are eliminated, and the kept block is returned as the next prior: priors_out[w] = (n, kept-block table with the
 * frames renumbered for the next window, x0, J0, r0), m[w] / n[w] = MarginalizationInfo::m / n (either may be NULL).
 * The windows are not modified.  With VPL_MARGIN_SECOND_NEW a window whose prior does not hold pose WINDOW_SIZE-1 gets its
 * input prior back (estimator.cpp:1385).  Synchronous (upload, kernels, download). */
int vpl_ba_marginalize(vpl_ctx* ctx, int n_windows, const vpl_window* windows, const vpl_ba_options* opt,
                       int marginalization_flag, vpl_prior* priors_out, int* m, int* n);

/* ---- line map maintenance that precedes the main solve (estimator.cpp:635-638) -------------------------------- */
/* FeatureManager::triangulateLine (feature_manager.cpp:413-563): lines with line_triangulated[i] == 0 are triangulated
 * from the two observations with the largest plane angle (skipped when cos > 0.998); on success line_plk[i] is written
 * and line_triangulated[i] set.  Synchronous (upload, kernel, download). */
int vpl_ba_triangulate_lines(vpl_ctx* ctx, int n_windows, vpl_window* windows);
/* What Estimator::slideWindow did to the track lists of one window (the lists themselves stay with the caller) */
typedef struct vpl_slide_tracks {
  int* point_start; /* [n_points] start_frame after the slide */
  int* point_nobs;  /* [n_points] observations left; 0 = the track was erased */
  int* point_drop;  /* [n_points] index (within the track) of the observation that was removed, -1 = none */
  int* line_start;  /* [n_lines] */
  int* line_nobs;   /* [n_lines] */
  int* line_drop;   /* [n_lines] */
} vpl_slide_tracks;
/* Estimator::slideWindow (estimator.cpp:1731-1851) of full windows in the NON_LINEAR state.
 * VPL_MARGIN_OLD: pose / speed_bias move one frame down (frame 10 keeps the newest state), then
 *   FeatureManager::removeBackShiftDepth (feature_manager.cpp:800-874): tracks that started in frame 0 lose that observation,
 *   are erased when fewer than two remain, else are re-anchored in the next frame -- inv_depth from the point carried over
 *   (non-positive depth -> init_depth), line_plk by plk_to_pose; all other tracks start one frame earlier.
 * VPL_MARGIN_SECOND_NEW: frame 10 is copied over frame 9, then FeatureManager::removeFront (feature_manager.cpp:915-956).
 * pose, speed_bias, inv_depth, line_plk are updated in place; the new track layout is reported in `tracks`.  The IMU
 * buffers (preint) are the caller's: the reference re-integrates them sample by sample (estimator.cpp:1786-1797).
 * Every track of the FeatureManager may be passed (any length >= 1, any start frame).  Synchronous. */
int vpl_ba_slide_window(vpl_ctx* ctx, int n_windows, vpl_window* windows, int marginalization_flag, double init_depth,
                        vpl_slide_tracks* tracks);
/* FeatureManager::triangulate (feature_manager.cpp:565-621): tracks whose inv_depth is negative (estimated_depth not set,
 * -1 in the reference) get the depth of the DLT / SVD over all their observations; a result below 0.1 becomes init_depth
 * (INIT_DEPTH = 5.0, parameters.cpp:138).  inv_depth is written.  Synchronous. */
int vpl_ba_triangulate_points(vpl_ctx* ctx, int n_windows, vpl_window* windows, double init_depth);
/* Estimator::onlyLineOpt (estimator.cpp:950-1039): line-only Levenberg-Marquardt with the poses and the extrinsic
 * constant, CauchyLoss(1.0), at most options.num_iterations iterations, then double2vector + removeLineOutlier.
 * Updates line_plk / line_removed of the triangulated lines; windows with fewer than four such lines are left untouched
 * (as :1019-1022).  Synchronous (asynchronous variant below). */
int vpl_ba_only_line_opt(vpl_ctx* ctx, int n_windows, vpl_window* windows, const vpl_ba_options* opt,
                         vpl_solve_report* reports);

/* Estimator::solveOdometry (estimator.cpp:624-648) in one call:  f_manager.triangulate  ||  (f_manager.triangulateLine ->
 * onlyLineOpt)  ->  optimizationwithLine.  `windows` are packed as for the single stages (every point track with
 * inv_depth < 0 where estimated_depth is unset; every line track that passes the LINE_MIN_OBS filter with its
 * is_triangulation flag; line_removed writable).  On return: inv_depth / line_plk / line_triangulated / line_removed as the
 * three map stages leave them -- a track erased by removeLineOutlier has line_removed = 1 and line_triangulated = 0 and took
 * no part in the solve -- and states, priors_out and reports as vpl_ba_solve_windows leaves them; line_reports (may be NULL)
 * are onlyLineOpt's.  The stages run back to back (the two triangulations on one upload): which lines take part changes between them
 * and the kernels' layout tables are built on the host from that set. */
int vpl_ba_solve_odometry(vpl_ctx* ctx, int n_windows, vpl_window* windows, const vpl_ba_options* opt, double init_depth,
                          vpl_prior* priors_out, vpl_solve_report* line_reports, vpl_solve_report* reports);
/* wall clock of the three stages of the context's last vpl_ba_solve_odometry, in ms: the two triangulations | onlyLineOpt |
 * optimizationwithLine (tools/time_odometry.py) */
int vpl_ba_debug_odometry_ms(vpl_ctx* ctx, double* ms3);

/* ---- asynchronous variants of the five entry points above -------------------------------------------------------------- *
 * Same arguments, same results, but the call returns as soon as its uploads, kernels and read-backs are ENQUEUED on the
 * context's stream; the results are written into the caller's arrays (windows, tracks, priors_out, m, n, reports) by
 * vpl_ba_collect -- or by the next call on the context that touches the batch (any upload / solve / download, another of
 * these calls, vpl_ctx_synchronize), which completes a pending call first.  The host is free in between (the reference's
 * processImage() does its feature-manager bookkeeping on the host, estimator.cpp:121-200); the arrays a call was given must stay
 * alive, and must not be read, until it has been collected.  One call can be pending per context.  The integer track
 * bookkeeping of vpl_ba_slide_window (`tracks`) is host work and is complete when the call returns; its states, inverse
 * depths and Pluecker lines are not. */
int vpl_ba_triangulate_points_async(vpl_ctx* ctx, int n_windows, vpl_window* windows, double init_depth);
int vpl_ba_triangulate_lines_async(vpl_ctx* ctx, int n_windows, vpl_window* windows);
int vpl_ba_only_line_opt_async(vpl_ctx* ctx, int n_windows, vpl_window* windows, const vpl_ba_options* opt,
                               vpl_solve_report* reports);
int vpl_ba_slide_window_async(vpl_ctx* ctx, int n_windows, vpl_window* windows, int marginalization_flag, double init_depth,
                              vpl_slide_tracks* tracks);
int vpl_ba_marginalize_async(vpl_ctx* ctx, int n_windows, const vpl_window* windows, const vpl_ba_options* opt,
                             int marginalization_flag, vpl_prior* priors_out, int* m, int* n);
/* Completes the pending asynchronous call, if any (waits for the stream, writes the results).  Returns its status. */
int vpl_ba_collect(vpl_ctx* ctx);

/* ---- instrumentation (bench.py) ------------------------------------------ */
/* Per-kernel device time of the last solve measured with hipEvents on the
 * context's stream. names/ms arrays of length *count on input; count updated. */
int vpl_ba_enable_kernel_timing(vpl_ctx* ctx, int enable);
int vpl_ba_kernel_times(vpl_ctx* ctx, int* count, const char** names, double* total_ms, int* launches);
/* Per-launch profile of the last solve that ran with kernel timing enabled: kernel name, device time [ms] and the
 * number of windows that did work in that launch, active[i][4] = (linearised, new Gauss-Newton step, re-used step of a
 * rejected iteration, candidate evaluated).  Arrays of length *count on input; count updated. */
int vpl_ba_launch_profile(vpl_ctx* ctx, int* count, const char** names, double* ms, int* active);

/* ---- layout self-check (host only, no device call; tests/test_point_units.py) ---------------------------------- *
 * The point factors of a window are packed into work units whose Hessian tiles are committed in a host-assigned ticket
 * order (csrc/ba_pack.h).  vpl_ba_debug_point_units builds the tables exactly as vpl_ba_upload does:
 * lane_table [max_rounds][512][2], unit_table [max_rounds][32][8][2] (descriptor, seq | seq0 << 16).
 * vpl_ba_debug_point_chains replays the commit chains of the solve pass (marg_pass = 0) or of the MARGIN_OLD pass
 * (marg_pass = 1, first rounds0 rounds) wave by wave: 1 = every wave finishes, 0 = a wave would wait forever. */
int vpl_ba_debug_point_units(int n_points, const int* point_start, const int* point_nobs, int max_rounds,
                             int* lane_table, int* unit_table, int* rounds, int* rounds0);
int vpl_ba_debug_point_chains(const int* unit_table, int rounds, int rounds0, int marg_pass);

#ifdef __cplusplus
}
#endif
#endif /* VPLINES_BA_H */
