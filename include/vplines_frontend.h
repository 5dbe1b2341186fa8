/*
 * vplines_frontend.h -- C ABI of the MI355X-native line front-end (EDLines extractor + KLT line matcher).
 *
 * Drop-in boundary for the line detector of multiplefish/VPLines-SLAM:
 *   int EDLineDetector::EDline(cv::Mat& image, std::vector<Line>& lines, bool smoothed)
 *        line_matching/src/edline_detector.h:82-84, edline_detector.cpp:1176-1198
 * as it is called by LineFeatureTracker::readImage (feature_tracker/src/line_feature_tracker.cpp:87)
 * with smoothed = true on 8-bit single-channel, undistorted + CLAHE'd frames.  Images arrive as raw
 * row-major uint8 buffers (cv::Mat::data of a continuous CV_8UC1 matrix); lines leave in the numeric
 * layout of struct Line (line_matching/src/line.h:8-17).
 *
 * smoothed = false -- the reference's default (edline_detector.h:79-81) and the path of its two demo programs -- runs
 * cv::GaussianBlur(image, Size(ksize, ksize), sigma) first (edline_detector.cpp:82-84): vpl_edlines_detect_ex.
 * Functions return 0 or a negative VPL_E_* code (same codes as vplines_ba.h); nothing falls back to a CPU path.
 */
#ifndef VPLINES_FRONTEND_H
#define VPLINES_FRONTEND_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* EDLineParam, edline_detector.h:33-41 (ksize / sigma: the Gaussian pre-blur of smoothed = false; unused with smoothed = true) */
typedef struct vpl_edline_param {
  int ksize;
  float sigma;
  float gradientThreshold;   /* production: 30 (line_feature_tracker_node.cpp:203) */
  float anchorThreshold;     /* 5 */
  int scanIntervals;         /* 2 */
  int minLineLen;            /* min_line_length = 35 */
  double lineFitErrThreshold;/* line_fit_err = 1.8 */
} vpl_edline_param;

/* numeric part of struct Line (line.h:8-17) */
typedef struct vpl_line {
  float line_endpoint[4];    /* x1,y1,x2,y2 */
  double line_equation[3];   /* w1 x + w2 y + w3 = 0, w1^2 + w2^2 = 1 */
  float center[2];
  float length;
} vpl_line;

typedef struct vpl_fe_ctx vpl_fe_ctx;   /* opaque: device buffers for a batch of frames of one size */

void vpl_edline_default_param(vpl_edline_param* p);   /* {5, 1, 30, 5, 2, 35, 1.8} */

int vpl_fe_create(vpl_fe_ctx** out, int device, int max_images, int width, int height, int max_lines_per_image);
void vpl_fe_destroy(vpl_fe_ctx* ctx);
int vpl_fe_set_stream(vpl_fe_ctx* ctx, void* hip_stream);   /* (work in flight on the stream used so far is completed first) */
int vpl_fe_synchronize(vpl_fe_ctx* ctx);
const char* vpl_fe_last_error(const vpl_fe_ctx* ctx);

/* ---- image preparation of LineFeatureTracker::readImage (feature_tracker/src/line_feature_tracker.cpp:62-68) ----
 *   cv::remap(_img, img, undist_map1_, undist_map2_, CV_INTER_LINEAR);  createCLAHE(3.0, Size(8,8))->apply(img, img);
 * vpl_pre_set_maps: the CV_32FC1 maps of PinholeCamera::initUndistortRectifyMap (line_feature_tracker.cpp:32), [H][W]
 *   each; NULL, NULL = no remap.  vpl_pre_upload: raw 8-bit frames [n][H][W].  vpl_pre_run (asynchronous): remap
 *   (INTER_LINEAR, BORDER_CONSTANT 0), then CLAHE when `equalize` (reference: clip 3.0, 8 x 8 tiles); the result
 *   REPLACES the frame batch that vpl_edlines_detect / vpl_match_run work on, so the raw frames never come back to the
 *   host.  vpl_pre_download copies the prepared frames out (tests, display). */
int vpl_pre_set_maps(vpl_fe_ctx* ctx, const float* map_x, const float* map_y);
int vpl_pre_upload(vpl_fe_ctx* ctx, int n_images, const uint8_t* raw_images);
int vpl_pre_run(vpl_fe_ctx* ctx, int equalize, double clip_limit, int tiles_x, int tiles_y);
int vpl_pre_download(vpl_fe_ctx* ctx, int n_images, uint8_t* images);
/* upload + run + (images != NULL ? download : synchronize) */
int vpl_pre_batch(vpl_fe_ctx* ctx, int n_images, const uint8_t* raw_images, int equalize, double clip_limit, int tiles_x,
                  int tiles_y, uint8_t* images);

/* instrumentation (bench.py): device time of every kernel launch of vpl_edlines_detect / vpl_match_run since timing was
 * enabled, measured with hipEvents on the context's stream, in launch order.  Arrays of length *count on input. */
int vpl_fe_enable_kernel_timing(vpl_fe_ctx* ctx, int enable);
int vpl_fe_kernel_times(vpl_fe_ctx* ctx, int* count, const char** names, double* ms);

/* three-phase form (inputs resident in HBM while timing) */
int vpl_edlines_upload(vpl_fe_ctx* ctx, int n_images, const uint8_t* images /* [n][H][W] */);
int vpl_edlines_detect(vpl_fe_ctx* ctx, const vpl_edline_param* param);   /* enqueue; asynchronous */
/* The reference collects lines in a std::vector; here a frame holds max_lines_per_image of them (vpl_fe_create).  A frame in
 * which the detector found more is refused: VPL_E_CAPACITY with the frame and its count in vpl_fe_last_error -- from this call
 * and from vpl_match_download when the match took its lines from the detector (vpl_match_from_detected); `lines` / `counts`
 * are written all the same (the lines that arrived first, in no defined selection). */
int vpl_edlines_download(vpl_fe_ctx* ctx, int n_images, vpl_line* lines /* [n][max_lines] */, int* counts /* [n] */);

/* EDline(image, lines, smoothed) with the flag as an argument.  smoothed == 0: the Gaussian pre-blur of EdgeDrawing
 * (edline_detector.cpp:82-84) runs first, fused with the gradient stage -- OpenCV's bit-exact 8-bit GaussianBlur: taps in 8.8
 * fixed point, rows then columns, BORDER_REFLECT_101.  ksize must be odd and <= 7 (ksize <= 0 with sigma > 0: OpenCV's automatic
 * size); frames of at least 8 x 8.
 * OpenCV published two roundings of the taps; vpl_fe_set_blur_kernel selects one per context:
 *   VPL_BLUR_NORMALISED (default): 3.4.9+ / 4.2+, taps sum to 256 (5, sigma 1: 14 62 104 62 14) -- the one that reproduces the
 *                                  reference's own output line_matching/data/edline_result.png (tests/golden/README.md);
 *   VPL_BLUR_OPENCV_341:           3.4.1 - 3.4.8 / 4.0 - 4.1, cvRound(256 g_i) (14 63 103 63 14; the README's 3.4.2). */
#define VPL_BLUR_NORMALISED 0
#define VPL_BLUR_OPENCV_341 1
int vpl_fe_set_blur_kernel(vpl_fe_ctx* ctx, int mode);
int vpl_edlines_detect_ex(vpl_fe_ctx* ctx, const vpl_edline_param* param, int smoothed);   /* enqueue; asynchronous */
int vpl_edlines_detect_batch_ex(vpl_fe_ctx* ctx, int n_images, const uint8_t* images, const vpl_edline_param* param, int smoothed,
                                vpl_line* lines, int* counts);
/* test access: keep a copy of the blurred frames of smoothed == 0 detections (off by default: the blurred frame otherwise
 * never leaves the work-group's LDS) and read one back */
int vpl_fe_keep_blurred(vpl_fe_ctx* ctx, int enable);
int vpl_edlines_debug_blurred(vpl_fe_ctx* ctx, int img, uint8_t* blurred /* [H*W] */);

/* EDline(image, lines, smoothed=true) for a batch: upload + detect + synchronize + download.
 * Lines of one image are returned in edge-chain order (the reference's order is the nondeterministic
 * arrival order of its worker threads, edline_detector.cpp:1080-1084). */
int vpl_edlines_detect_batch(vpl_fe_ctx* ctx, int n_images, const uint8_t* images, const vpl_edline_param* param,
                             vpl_line* lines, int* counts);

/* LineMatching::LineFilter(lines, distance_threshold, parallel_threshold = sin 3 deg) (line_matching.h:35-37,
 * line_matching.cpp:167-264; the reference's demo calls it between detector and matcher, test_line_matching.cpp:57,74): in
 * order of decreasing length, a line removes every shorter one that is nearly parallel to it and has an end point closer than
 * distance_threshold.  Lines of equal length are taken in index order (the reference's std::sort leaves their order open).
 * _detected: on the lines of the last detect where they lie in HBM (asynchronous; vpl_edlines_download / vpl_match_from_detected
 * then see the filtered lists).  _batch: caller-owned lists [n][max_lines], counts [n], both in / out (uses the context's
 * line table: the last detect's lines are overwritten).  max_lines_per_image <= 8192. */
#define VPL_LINE_FILTER_PARALLEL_DEFAULT 0.0348994967f
int vpl_line_filter_detected(vpl_fe_ctx* ctx, float distance_threshold, float parallel_threshold);
int vpl_line_filter_batch(vpl_fe_ctx* ctx, int n_images, vpl_line* lines, int* counts, float distance_threshold,
                          float parallel_threshold);

/* test access to the intermediate stages of image `img` of the last detect (any pointer may be NULL):
 * dx, dy, gImg [H*W int16]; dirImg [H*W uint8]; anchors [2*cap uint32 x,y] ; chains xC,yC [cap uint32], sId [cap/20+2] */
int vpl_edlines_debug_stage(vpl_fe_ctx* ctx, int img, int16_t* dx, int16_t* dy, int16_t* gImg, uint8_t* dirImg,
                            uint32_t* anchors, int* n_anchors, uint32_t* chain_x, uint32_t* chain_y, uint32_t* sId,
                            int* n_edges);
/* routing counters of image `img` of the last detect: steps walked, LDS tile loads, walks started, shader cycles */
int vpl_edlines_debug_route_stats(vpl_fe_ctx* ctx, int img, unsigned long long* out4);

/* ------------------------------------------------------------------------------------------------------------
 * KLT line matching.  Drop-in boundary:
 *   bool LineMatching::Matching(img_ref, img_cur, lines_ref, lines_cur, line_ref_to_line_cur, K_ref, K_cur, T_cur_ref,
 *                               illumination_adapt, topological_filter, ...)
 *        line_matching/src/line_matching.h:21-33, line_matching.cpp:605-690
 * as called by LineFeatureTracker::match_line_match (feature_tracker/src/line_feature_tracker.cpp:291-314) with
 * K_ref = K_cur = T_cur_ref = NULL, illumination_adapt = true, topological_filter = true.  The KLT inside is
 * KLT::calc2D (klt.cpp:491-628) with a 13x13 window, 4 pyramid levels, 30 iterations / eps 0.001, minEig 1e-4.
 * Images are the frames uploaded with vpl_edlines_upload (a pair names two of them by index); lines are passed by the
 * caller, as in the reference (they may have been filtered on the host after detection).
 * ------------------------------------------------------------------------------------------------------------ */

/* LineMatching ctor arguments (line_matching.h:14-18), the two Matching() flags, TopologicalFilter defaults (:45-47) */
typedef struct vpl_match_param {
  int step;                          /* 10 */
  float closest_line_threshold;      /* 0.5 */
  float line_matching_ratio;         /* 0.4 */
  float line_distance_error_ratio;   /* 3 */
  float klt_error_threshold;         /* 40 */
  int illumination_adapt;            /* 1 in the tracker */
  int topological_filter;            /* 1 in the tracker */
  float topo_distance_threshold;     /* 15 */
  float topo_length_tolerate_ratio;  /* 0.2 */
  float topo_violation_ratio;        /* 0.05 */
} vpl_match_param;

void vpl_match_default_param(vpl_match_param* p);

/* device buffers for up to max_pairs pairs, max_kps key points per pair (Anchors() output, ~ sum(len/step + 2)) */
int vpl_match_reserve(vpl_fe_ctx* ctx, int max_pairs, int max_kps);

/* three-phase form.  lines_ref / lines_cur: [n_pairs][max_lines_per_image] (the stride given to vpl_fe_create). */
int vpl_match_upload(vpl_fe_ctx* ctx, int n_pairs, const int* ref_image, const int* cur_image,
                     const vpl_line* lines_ref, const int* n_ref, const vpl_line* lines_cur, const int* n_cur);
/* The lines of this context's last vpl_edlines_detect as the matcher's input, device to device: what
 * LineFeatureTracker::readImage does between detector and matcher (line_feature_tracker.cpp:110-118, :290-322) without the lines
 * leaving HBM.  At most max_lines lines per frame take part (the first ones in the detector's order).  Asynchronous.
 * vpl_match_counts: how many lines of each pair's frames take part (sizes of the rows vpl_match_download fills). */
int vpl_match_from_detected(vpl_fe_ctx* ctx, int n_pairs, const int* ref_image, const int* cur_image, int max_lines);
int vpl_match_counts(vpl_fe_ctx* ctx, int n_pairs, int* n_ref, int* n_cur);
int vpl_match_run(vpl_fe_ctx* ctx, const vpl_match_param* param);   /* enqueue; asynchronous */
/* line_ref_to_line_cur [n_pairs][max_lines] (-1 = unmatched; rows of pairs with matched == 0 are left untouched, as
 * Matching() leaves its output vector when it returns false); matched [n_pairs] = Matching()'s return value.
 * VPL_E_CAPACITY if a pair produced more key points than max_kps. */
int vpl_match_download(vpl_fe_ctx* ctx, int n_pairs, int* line_ref_to_line_cur, int* matched);

/* Matching() for a batch of pairs: images [n_images][H][W] are uploaded, then upload + run + synchronize + download */
int vpl_line_match_batch(vpl_fe_ctx* ctx, int n_images, const uint8_t* images, int n_pairs, const int* ref_image,
                         const int* cur_image, const vpl_line* lines_ref, const int* n_ref, const vpl_line* lines_cur,
                         const int* n_cur, const vpl_match_param* param, int* line_ref_to_line_cur, int* matched);

/* test access (any pointer may be NULL): key points of pair `pair` of the last run (LineMatching::getPointMatchResult,
 * line_matching.h:60-65): kps_ref, kps_cur [2*n_kps], status, err, kp2line_cur [n_kps] */
int vpl_match_debug_kps(vpl_fe_ctx* ctx, int pair, int cap, float* kps_ref, float* kps_cur, uint8_t* status, float* err,
                        int* kp2line_cur, int* n_kps);
/* pyramid level `level` of image `img` without its border: pixels [h*w], derivative [h*w*2]; *w, *h; returns
 * VPL_E_INVALID for a level that was not built */
int vpl_match_debug_level(vpl_fe_ctx* ctx, int img, int level, uint8_t* pixels, int16_t* deriv, int* w, int* h);

/* ---- vanishing points: vanishing_point_detection::run_vanishing_point_detection
 * (feature_tracker/src/vanishing_point_detection.cpp:37-66, called at line_feature_tracker.cpp:237-266) for a batch of frames.
 *   hyp_lines [n][max_lines]  the lines the 2-line hypotheses and the sphere grid are built from (`lines`: verticalLine or
 *                             all lines, :240-243);  all_lines [n][max_lines]  the lines that are classified
 *   f, cx, cy                 init(K(0,0), K(0,2), K(1,2), 0.5) (line_feature_tracker.cpp:33)
 *   seeds [n]                 the reference calls srand(time(NULL)) in every frame (:107); here the caller supplies the
 *                             seed and the device generates glibc's rand() stream for it
 *   first_frame [n]           frame_count == 0 (no vps[1]/vps[2] swap, :318-339)
 *   vps [n][3][3]             the three unit vectors;  vp_ids [n][max_lines]  per line 0..2 = its VP, 3 = none
 *   status [n]                0, or -1 when no hypothesis can be drawn (fewer than two lines, or only parallel ones -- the
 *                             reference does not return in that case); vps are 0 and every id is 3 then
 * Defined where the reference is not: a query index past the end of lx (:411-416, :428-431) means "no query line".
 * Synchronous. */
int vpl_vp_detect_batch(vpl_fe_ctx* ctx, int n_frames, const vpl_line* hyp_lines, const int* n_hyp, const vpl_line* all_lines,
                        const int* n_all, float f, float cx, float cy, const uint32_t* seeds, const int* first_frame,
                        double* vps, int* vp_ids, int* status);
/* test access after vpl_vp_detect_batch (any pointer may be NULL): smoothed sphere grid [90][360], the drawn line pairs
 * [105][2], index of the winning hypothesis, rand() calls made by the hypothesis stage */
int vpl_vp_debug(vpl_fe_ctx* ctx, int frame, double* grid, int* pairs, int* best_idx, int* drawn);

/* ---- list handling of LineFeatureTracker::readImage after the match (line_feature_tracker.cpp:96-229) ----
 * Host-side integer bookkeeping (no device work, no context): id propagation from the previous frame's lines through
 * line_ref_to_line_cur, fresh ids for the unmatched lines, and the max_h_lines / max_v_lines quota.  The reference's
 * behaviour is kept as written: a match to detection 0 is ignored (`mt > 0`, :121); t_cnt is read with the NEW frame's
 * index (:123) from a vector that keeps the previous frame's detection count (n_tcnt_prev; a read past its end, which
 * the reference leaves undefined, counts as 0); new lines are classed with the 3.14-based angle test of :166.
 *   ends_new   [n_new][4]  end points of the new frame's detections
 *   id_prev    [n_prev]    lineID of the previous frame's (kept) lines;  prev_to_new [n_prev] their match, -1 = none
 *   keep, id_out (capacity n_new): the new frame's kept lines as indices into the detections and their ids;
 *   tcnt_out   [n_new]     t_cnt in detection order.  *allfeature_cnt is advanced by the number of fresh ids.
 *   vertical_new (capacity n_new), n_vertical_new: may be NULL; the reference's `verticalLine` (:151-176), i.e. every
 *              new line of the v class whether the quota keeps it or not (the test on tracked lines at :151 is never true)
 * Returns the number of kept lines, or a negative VPL_E_* code. */
int vpl_line_track_ids(int n_new, const float* ends_new, int n_prev, const int* id_prev, const int* tcnt_prev, int n_tcnt_prev,
                       const int* prev_to_new, int max_h_lines, int max_v_lines, int* allfeature_cnt, int* keep, int* id_out,
                       int* tcnt_out, int* vertical_new, int* n_vertical_new);

#ifdef __cplusplus
}
#endif
#endif /* VPLINES_FRONTEND_H */
