/*
 * vplines_frontend.h -- C ABI of the MI355X-native line front-end (EDLines extractor).
 *
 * Drop-in boundary for the line detector of multiplefish/VPLines-SLAM:
 *   int EDLineDetector::EDline(cv::Mat& image, std::vector<Line>& lines, bool smoothed)
 *        line_matching/src/edline_detector.h:82-84, edline_detector.cpp:1176-1198
 * as it is called by LineFeatureTracker::readImage (feature_tracker/src/line_feature_tracker.cpp:87)
 * with smoothed = true on 8-bit single-channel, undistorted + CLAHE'd frames.  Images arrive as raw
 * row-major uint8 buffers (cv::Mat::data of a continuous CV_8UC1 matrix); lines leave in the numeric
 * layout of struct Line (line_matching/src/line.h:8-17).
 *
 * Only smoothed = true is implemented (the production path); the Gaussian pre-blur of the demo path
 * (edline_detector.cpp:82-84) is not.  Functions return 0 or a negative VPL_E_* code
 * (same codes as vplines_ba.h); nothing falls back to a CPU path.
 */
#ifndef VPLINES_FRONTEND_H
#define VPLINES_FRONTEND_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* EDLineParam, edline_detector.h:33-41 (ksize/sigma are carried for layout compatibility; unused with smoothed=true) */
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
int vpl_fe_set_stream(vpl_fe_ctx* ctx, void* hip_stream);
int vpl_fe_synchronize(vpl_fe_ctx* ctx);
const char* vpl_fe_last_error(const vpl_fe_ctx* ctx);

/* three-phase form (inputs resident in HBM while timing) */
int vpl_edlines_upload(vpl_fe_ctx* ctx, int n_images, const uint8_t* images /* [n][H][W] */);
int vpl_edlines_detect(vpl_fe_ctx* ctx, const vpl_edline_param* param);   /* enqueue; asynchronous */
int vpl_edlines_download(vpl_fe_ctx* ctx, int n_images, vpl_line* lines /* [n][max_lines] */, int* counts /* [n] */);

/* EDline(image, lines, smoothed=true) for a batch: upload + detect + synchronize + download.
 * Lines of one image are returned in edge-chain order (the reference's order is the nondeterministic
 * arrival order of its worker threads, edline_detector.cpp:1080-1084). */
int vpl_edlines_detect_batch(vpl_fe_ctx* ctx, int n_images, const uint8_t* images, const vpl_edline_param* param,
                             vpl_line* lines, int* counts);

/* test access to the intermediate stages of image `img` of the last detect (any pointer may be NULL):
 * dx, dy, gImg [H*W int16]; dirImg [H*W uint8]; anchors [2*cap uint32 x,y] ; chains xC,yC [cap uint32], sId [cap/20+2] */
int vpl_edlines_debug_stage(vpl_fe_ctx* ctx, int img, int16_t* dx, int16_t* dy, int16_t* gImg, uint8_t* dirImg,
                            uint32_t* anchors, int* n_anchors, uint32_t* chain_x, uint32_t* chain_y, uint32_t* sId,
                            int* n_edges);

#ifdef __cplusplus
}
#endif
#endif /* VPLINES_FRONTEND_H */
