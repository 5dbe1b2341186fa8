// TEST INFRASTRUCTURE -- CPU oracle of the image preparation in LineFeatureTracker::readImage
// (feature_tracker/src/line_feature_tracker.cpp:62-68):
//     cv::remap(_img, img, undist_map1_, undist_map2_, CV_INTER_LINEAR);
//     cv::createCLAHE(3.0, cv::Size(8, 8))->apply(img, img);           (when EQUALIZE)
// and of the list handling after the match (:96-229, see track_ids below).
//
// PARITY UNPINNED.  remap and CLAHE live in OpenCV 3.4.2 (README.md:5), which is neither under /root/reference nor in
// this image; what follows restates the published algorithms of modules/imgproc (imgwarp.cpp remap with CV_32FC1 maps,
// clahe.cpp) for 8-bit single-channel images, and the reference holds no fixture for either:
//   remap:  sx = cvRound(mapx*32), sy = cvRound(mapy*32) (INTER_BITS = 5); the four neighbours are weighted with the
//           integer table (32-fx)(32-fy)*32 ... (INTER_REMAP_COEF_BITS = 15; the table's saturated entry 32768 -> 32767,+1
//           of the (0,0) cell cannot change an 8-bit result), rounded by (sum + 2^14) >> 15; BORDER_CONSTANT 0 per tap.
//   CLAHE:  per-tile 256-bin histogram, clipLimit = max(int(clip * tileArea / 256), 1), excess redistributed as
//           clipped/256 to every bin and the remainder one count every max(256/residual, 1) bins, LUT =
//           cvRound(cumsum * 255/tileArea) in float; bilinear interpolation between the four surrounding tile LUTs in
//           float, cvRound.  Images whose size is not a multiple of the grid are padded with BORDER_REFLECT_101 for the
//           histograms.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

inline int cv_round(float v) { return (int)std::lrintf(v); }   // round half to even, as SSE cvtss2si
inline int cv_floor(float v) { int i = (int)v; return i - (i > v); }
inline int sat_short(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
  return p;
}

void remap_linear(const uint8_t* src, int W, int H, const float* mapx, const float* mapy, uint8_t* dst) {
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const size_t i = (size_t)y * W + x;
      const int sx = cv_round(mapx[i] * 32), sy = cv_round(mapy[i] * 32);
      const int ix = sat_short(sx >> 5), iy = sat_short(sy >> 5), fx = sx & 31, fy = sy & 31;
      const int w[4] = {(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32};
      int acc = 0;
      for (int t = 0; t < 4; ++t) {
        const int px = ix + (t & 1), py = iy + (t >> 1);
        const int v = (px >= 0 && px < W && py >= 0 && py < H) ? src[(size_t)py * W + px] : 0;
        acc += v * w[t];
      }
      dst[i] = sat_u8((acc + (1 << 14)) >> 15);
    }
}

void clahe(const uint8_t* src, int W, int H, double clip_limit, int tilesX, int tilesY, uint8_t* dst) {
  int extW = W, extH = H;
  if (W % tilesX != 0 || H % tilesY != 0) { extW = W + (tilesX - W % tilesX); extH = H + (tilesY - H % tilesY); }
  const int tw = extW / tilesX, th = extH / tilesY, area = tw * th;
  const float lutScale = (float)255 / area;
  int clipLimit = 0;
  if (clip_limit > 0.0) clipLimit = std::max((int)(clip_limit * area / 256), 1);
  std::vector<uint8_t> lut((size_t)tilesX * tilesY * 256);
  for (int ty = 0; ty < tilesY; ++ty)
    for (int tx = 0; tx < tilesX; ++tx) {
      int hist[256] = {0};
      for (int y = ty * th; y < (ty + 1) * th; ++y)
        for (int x = tx * tw; x < (tx + 1) * tw; ++x) hist[src[(size_t)reflect101(y, H) * W + reflect101(x, W)]]++;
      if (clipLimit > 0) {
        int clipped = 0;
        for (int i = 0; i < 256; ++i)
          if (hist[i] > clipLimit) { clipped += hist[i] - clipLimit; hist[i] = clipLimit; }
        const int redistBatch = clipped / 256;
        int residual = clipped - redistBatch * 256;
        for (int i = 0; i < 256; ++i) hist[i] += redistBatch;
        if (residual != 0) {
          const int residualStep = std::max(256 / residual, 1);
          for (int i = 0; i < 256 && residual > 0; i += residualStep, residual--) hist[i]++;
        }
      }
      uint8_t* tl = &lut[(size_t)(ty * tilesX + tx) * 256];
      int sum = 0;
      for (int i = 0; i < 256; ++i) { sum += hist[i]; tl[i] = sat_u8(cv_round(sum * lutScale)); }
    }
  const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
  for (int y = 0; y < H; ++y) {
    const float tyf = y * inv_th - 0.5f;
    int ty1 = cv_floor(tyf), ty2 = ty1 + 1;
    const float ya = tyf - ty1, ya1 = 1.0f - ya;
    ty1 = std::max(ty1, 0); ty2 = std::min(ty2, tilesY - 1);
    const uint8_t* p1 = &lut[(size_t)ty1 * tilesX * 256];
    const uint8_t* p2 = &lut[(size_t)ty2 * tilesX * 256];
    for (int x = 0; x < W; ++x) {
      const float txf = x * inv_tw - 0.5f;
      int tx1 = cv_floor(txf), tx2 = tx1 + 1;
      const float xa = txf - tx1, xa1 = 1.0f - xa;
      tx1 = std::max(tx1, 0); tx2 = std::min(tx2, tilesX - 1);
      const int v = src[(size_t)y * W + x];
      const int i1 = tx1 * 256 + v, i2 = tx2 * 256 + v;
      const float res = (p1[i1] * xa1 + p1[i2] * xa) * ya1 + (p2[i1] * xa1 + p2[i2] * xa) * ya;
      dst[(size_t)y * W + x] = sat_u8(cv_round(res));
    }
  }
}

// segAngle (line_feature_tracker.cpp:20-25)
double seg_angle(const float* e) {
  if (e[2] > e[0]) return std::atan2(e[3] - e[1], e[2] - e[0]);
  return std::atan2(e[1] - e[3], e[0] - e[2]);
}
bool is_h(double angle) {   // the "h" class of :166,185 (3.14, not pi)
  return (angle >= 3.14 / 4.0 && angle <= 3 * 3.14 / 4.0) || (angle <= -3.14 / 4.0 && angle >= -3 * 3.14 / 4.0);
}

}  // namespace

extern "C" {

int orc_remap_linear(const uint8_t* src, int W, int H, const float* mapx, const float* mapy, uint8_t* dst) {
  remap_linear(src, W, H, mapx, mapy, dst);
  return 0;
}

// in-place use (dst == src) is what the reference does; the LUTs are complete before the first pixel is rewritten
int orc_clahe(const uint8_t* src, int W, int H, double clip_limit, int tilesX, int tilesY, uint8_t* dst) {
  std::vector<uint8_t> tmp(src, src + (size_t)W * H);
  clahe(tmp.data(), W, H, clip_limit, tilesX, tilesY, dst);
  return 0;
}

// The list handling of LineFeatureTracker::readImage after the match (line_feature_tracker.cpp:96-229), quirks kept:
//   * a match to forw line 0 is dropped (`if (mt > 0)`, :121);
//   * t_cnt is read with the forw index (:123): forw.t_cnt[mt] = cur.t_cnt[mt] + 1.  cur.t_cnt keeps the length of cur's
//     detection list (n_tcnt_cur, it is not swapped at :227-228); past its end the reference reads out of bounds, this
//     restatement defines that read as 0;
//   * unmatched lines get fresh ids in detection order (:137-141), are split into the h / v angle classes (:160-178) and
//     only fill what the tracked lines leave of max_h_lines / max_v_lines (:196-224).
// ends_forw: [n_forw][4] end points.  id_cur: [n_cur], tcnt_cur: [n_tcnt_cur].  cur_to_forw: [n_cur] (LineMatching::Matching's
// line_ref_to_line_cur with ref = previous frame), -1 = no match.  Outputs, capacity n_forw: keep[] = indices into the
// forw detection list in the new order, id_out[], tcnt_out[] (t_cnt keeps the pre-swap detection order, as the
// reference's vector does, so tcnt_out[k] belongs to detection k, not to keep[k]).  Returns the new line count.
int orc_track_ids(int n_forw, const float* ends_forw, int n_cur, const int* id_cur, const int* tcnt_cur, int n_tcnt_cur,
                  const int* cur_to_forw,
                  int max_h_lines, int max_v_lines, int* allfeature_cnt, int* keep, int* id_out, int* tcnt_out, int* vertical_new,
                  int* n_vertical_new) {
  std::vector<int> lineID(n_forw, -1), t_cnt(n_forw, 0);
  for (int k = 0; k < n_cur; ++k) {
    const int mt = cur_to_forw[k];
    if (mt > 0 && mt < n_forw) {
      lineID[mt] = id_cur[k];
      t_cnt[mt] = (mt < n_tcnt_cur ? tcnt_cur[mt] : 0) + 1;
    }
  }
  std::vector<int> tracked, tracked_id, h_new, h_id, v_new, v_id;
  for (int i = 0; i < n_forw; ++i) {
    if (lineID[i] == -1) {
      lineID[i] = (*allfeature_cnt)++;
      (is_h(seg_angle(ends_forw + 4 * i)) ? h_new : v_new).push_back(i);
      (is_h(seg_angle(ends_forw + 4 * i)) ? h_id : v_id).push_back(lineID[i]);
    } else {
      tracked.push_back(i);
      tracked_id.push_back(lineID[i]);
    }
  }
  int h_line = 0, v_line = 0;
  for (int i : tracked) (is_h(seg_angle(ends_forw + 4 * i)) ? h_line : v_line)++;
  int diff_h = max_h_lines - h_line, diff_v = max_v_lines - v_line;
  if (diff_h > 0) {
    diff_h = std::min<int>(diff_h, (int)h_new.size());
    for (int k = 0; k < diff_h; ++k) { tracked.push_back(h_new[k]); tracked_id.push_back(h_id[k]); }
  }
  if (diff_v > 0) {
    diff_v = std::min<int>(diff_v, (int)v_new.size());
    for (int k = 0; k < diff_v; ++k) { tracked.push_back(v_new[k]); tracked_id.push_back(v_id[k]); }
  }
  for (size_t k = 0; k < tracked.size(); ++k) { keep[k] = tracked[k]; id_out[k] = tracked_id[k]; }
  for (int k = 0; k < n_forw; ++k) tcnt_out[k] = t_cnt[k];
  if (vertical_new && n_vertical_new) {   // verticalLine (:151-176): the tracked-line test there can never hold, so new v lines only
    for (size_t k = 0; k < v_new.size(); ++k) vertical_new[k] = v_new[k];
    *n_vertical_new = (int)v_new.size();
  }
  return (int)tracked.size();
}

}  // extern "C"
