// Deterministic synthetic sliding windows for tests and bench.py.
// Host-only C++; no reference code and no oracle code is used here.  It produces
// exactly the inputs Estimator::optimizationwithLine() consumes (vpl_window
// fields) plus the raw IMU samples Estimator::processIMU() would have received.
//
//   RNG  : PCG64 (XSL-RR 128/64), seed = caller-provided 64-bit value
//   path : circle r = 2 m, omega = 0.5 rad/s, z = 1 + 0.3 sin(1.3 t), small roll/pitch/yaw wobble;
//          keyframes at 10 Hz, IMU at 200 Hz (20 samples per keyframe interval), as SURVEY.md 8d
//   body : x = tangent, y = up, z = radially outward, so that the EuRoC camera
//          (optical axis ~ body z, config/euroc/euroc_config.yaml:35-43) looks outward
//   noise: EuRoC values (config/euroc/euroc_config.yaml:59-63)
#include <cmath>
#include <cstdint>
#include <cstring>

namespace {

struct Pcg64 {
  unsigned __int128 state, inc;
  explicit Pcg64(uint64_t seed) {
    inc = (((unsigned __int128)0x5851f42d4c957f2dULL << 64 | 0x14057b7ef767814fULL) << 1) | 1;
    state = 0;
    next();
    state += ((unsigned __int128)seed << 64) | (seed ^ 0x9e3779b97f4a7c15ULL);
    next();
  }
  uint64_t next() {
    const unsigned __int128 mult = ((unsigned __int128)0x2360ed051fc65da4ULL << 64) | 0x4385df649fccf645ULL;
    state = state * mult + inc;
    uint64_t hi = (uint64_t)(state >> 64), lo = (uint64_t)state;
    uint64_t x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((64 - rot) & 63));
  }
  double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double uniform(double a, double b) { return a + (b - a) * uniform(); }
  double normal() {
    double u1 = uniform(), u2 = uniform();
    if (u1 < 1e-300) u1 = 1e-300;
    return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
  }
};

struct V3 { double x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
struct M3 { double m[3][3]; };
inline V3 mul(const M3& A, V3 v) {
  return {A.m[0][0] * v.x + A.m[0][1] * v.y + A.m[0][2] * v.z, A.m[1][0] * v.x + A.m[1][1] * v.y + A.m[1][2] * v.z,
          A.m[2][0] * v.x + A.m[2][1] * v.y + A.m[2][2] * v.z};
}
inline V3 mulT(const M3& A, V3 v) {
  return {A.m[0][0] * v.x + A.m[1][0] * v.y + A.m[2][0] * v.z, A.m[0][1] * v.x + A.m[1][1] * v.y + A.m[2][1] * v.z,
          A.m[0][2] * v.x + A.m[1][2] * v.y + A.m[2][2] * v.z};
}
inline M3 mul(const M3& A, const M3& B) {
  M3 C;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) C.m[i][j] = A.m[i][0] * B.m[0][j] + A.m[i][1] * B.m[1][j] + A.m[i][2] * B.m[2][j];
  return C;
}
inline M3 cols(V3 a, V3 b, V3 c) { return M3{{{a.x, b.x, c.x}, {a.y, b.y, c.y}, {a.z, b.z, c.z}}}; }
// rotation matrix -> unit quaternion (x,y,z,w), w >= 0 branch-complete
void mat2quat(const M3& R, double q[4]) {
  double t = R.m[0][0] + R.m[1][1] + R.m[2][2];
  double w, x, y, z;
  if (t > 0) {
    double s = std::sqrt(t + 1.0) * 2;
    w = 0.25 * s; x = (R.m[2][1] - R.m[1][2]) / s; y = (R.m[0][2] - R.m[2][0]) / s; z = (R.m[1][0] - R.m[0][1]) / s;
  } else if (R.m[0][0] > R.m[1][1] && R.m[0][0] > R.m[2][2]) {
    double s = std::sqrt(1.0 + R.m[0][0] - R.m[1][1] - R.m[2][2]) * 2;
    w = (R.m[2][1] - R.m[1][2]) / s; x = 0.25 * s; y = (R.m[0][1] + R.m[1][0]) / s; z = (R.m[0][2] + R.m[2][0]) / s;
  } else if (R.m[1][1] > R.m[2][2]) {
    double s = std::sqrt(1.0 + R.m[1][1] - R.m[0][0] - R.m[2][2]) * 2;
    w = (R.m[0][2] - R.m[2][0]) / s; x = (R.m[0][1] + R.m[1][0]) / s; y = 0.25 * s; z = (R.m[1][2] + R.m[2][1]) / s;
  } else {
    double s = std::sqrt(1.0 + R.m[2][2] - R.m[0][0] - R.m[1][1]) * 2;
    w = (R.m[1][0] - R.m[0][1]) / s; x = (R.m[0][2] + R.m[2][0]) / s; y = (R.m[1][2] + R.m[2][1]) / s; z = 0.25 * s;
  }
  q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}
M3 quat2mat(const double q[4]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  return M3{{{1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)},
             {2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)},
             {2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)}}};
}
// exp map of a rotation vector
M3 expso3(V3 w) {
  double th = norm(w);
  M3 I{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
  if (th < 1e-12) return I;
  V3 k = w * (1.0 / th);
  double c = std::cos(th), s = std::sin(th), v = 1 - c;
  return M3{{{c + k.x * k.x * v, k.x * k.y * v - k.z * s, k.x * k.z * v + k.y * s},
             {k.y * k.x * v + k.z * s, c + k.y * k.y * v, k.y * k.z * v - k.x * s},
             {k.z * k.x * v - k.y * s, k.z * k.y * v + k.x * s, c + k.z * k.z * v}}};
}

const double kRadius = 2.0, kOmega = 0.5, kG = 9.81007;

struct State { V3 p, v, a; M3 R; V3 w_b; };
// attitude: yaw follows the circle, plus small roll/pitch oscillations so that the
// camera-IMU extrinsic and the biases are observable inside one window
M3 attitude(double t) {
  double th = kOmega * t;
  V3 xb{-std::sin(th), std::cos(th), 0}, zb{std::cos(th), std::sin(th), 0}, yb{0, 0, 1};
  M3 base = cols(xb, yb, zb);
  V3 wob{0.20 * std::sin(1.7 * t + 0.3), 0.15 * std::sin(1.1 * t + 1.0), 0.10 * std::sin(2.3 * t)};
  return mul(base, expso3(wob));
}
State truth(double t) {
  double th = kOmega * t;
  State s;
  s.p = {kRadius * std::cos(th), kRadius * std::sin(th), 1.0 + 0.3 * std::sin(1.3 * t)};
  s.v = {-kRadius * kOmega * std::sin(th), kRadius * kOmega * std::cos(th), 0.3 * 1.3 * std::cos(1.3 * t)};
  s.a = {-kRadius * kOmega * kOmega * std::cos(th), -kRadius * kOmega * kOmega * std::sin(th),
         -0.3 * 1.3 * 1.3 * std::sin(1.3 * t)};
  s.R = attitude(t);
  // body angular velocity by central differences of R(t): vee(R^T dR/dt), error O(h^2) ~ 1e-9
  const double h = 1e-5;
  M3 Rp = attitude(t + h), Rm = attitude(t - h);
  M3 D;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) D.m[i][j] = (Rp.m[i][j] - Rm.m[i][j]) / (2 * h);
  M3 W;  // R^T * D
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) W.m[i][j] = s.R.m[0][i] * D.m[0][j] + s.R.m[1][i] * D.m[1][j] + s.R.m[2][i] * D.m[2][j];
  s.w_b = {0.5 * (W.m[2][1] - W.m[1][2]), 0.5 * (W.m[0][2] - W.m[2][0]), 0.5 * (W.m[1][0] - W.m[0][1])};
  return s;
}

// EuRoC extrinsic, config/euroc/euroc_config.yaml:35-43 (imu^R_cam, imu^T_cam)
const M3 kRic{{{0.0148655429818, -0.999880929698, 0.00414029679422},
               {0.999557249008, 0.0149672133247, 0.025715529948},
               {-0.0257744366974, 0.00375618835797, 0.999660727178}}};
const V3 kTic{-0.0216401454975, -0.064676986768, 0.00981073058949};

// orthonormal (psi1,psi2,psi3,phi) <-> Pluecker; own derivation of the Bartoli-Sturm form
void plk_to_orth(V3 n, V3 v, double o[4]) {
  double nn = norm(n), vn = norm(v);
  V3 u1 = n * (1 / nn), u2 = v * (1 / vn), u3 = cross(u1, u2);
  o[0] = std::atan2(u2.z, u3.z);
  o[1] = std::asin(-u1.z);
  o[2] = std::atan2(u1.y, u1.x);
  o[3] = std::asin(vn / std::sqrt(nn * nn + vn * vn));
}
void orth_to_plk(const double o[4], V3& n, V3& v) {
  double s1 = std::sin(o[0]), c1 = std::cos(o[0]), s2 = std::sin(o[1]), c2 = std::cos(o[1]), s3 = std::sin(o[2]), c3 = std::cos(o[2]);
  V3 u1{c2 * c3, c2 * s3, -s2};
  V3 u2{s1 * s2 * c3 - c1 * s3, s1 * s2 * s3 + c1 * c3, s1 * c2};
  n = u1 * std::cos(o[3]);
  v = u2 * std::sin(o[3]);
}

}  // namespace

extern "C" {

struct vplw_config {
  int n_points;       // P
  int n_lines;        // L
  int track_len;      // observations per track (6)
  int with_vp;        // every line observation carries its VP (flag 1.0)
  int imu_rate_div;   // IMU samples per keyframe interval
  double kf_dt;       // keyframe spacing [s]
  double pose_sigma_p, pose_sigma_theta_deg, vel_sigma;  // initial state perturbation (window drift)
  double frame_sigma_scale;                              // independent per-frame part, as a fraction of the above
  double pix_sigma;   // observation noise in normalised units (1/460)
  double depth_rel_sigma, orth_sigma;
  double acc_n, gyr_n, ba_sigma, bg_sigma;
  double triad_roll_deg;  // rotation of the Manhattan triad about the mid-window direction of travel
  // Sign of the z component of the initial Pluecker direction in the start camera frame.  The observed VP always has
  // z > 0 (the detector's convention, vanishing_point_detection.cpp:154,160).  vpProjectionFactor's literal Jacobian
  // (line_projection_factor.cpp:62-64) equals -d_z/|d| * |d| times the true derivative in its first two columns: a
  // descent direction only for d_z < 0.  -1: d_z < 0 (default), +1: d_z > 0, 0: alternating per line
  int line_dir_sign;
};

void vplw_default_config(vplw_config* c, int n_points, int n_lines, int with_vp) {
  c->n_points = n_points; c->n_lines = n_lines; c->track_len = 6; c->with_vp = with_vp; c->imu_rate_div = 20; c->kf_dt = 0.1;
  c->pose_sigma_p = 0.05; c->pose_sigma_theta_deg = 1.0; c->vel_sigma = 0.05; c->frame_sigma_scale = 0.1;
  c->pix_sigma = 1.0 / 460.0; c->depth_rel_sigma = 0.1; c->orth_sigma = 0.02;
  c->acc_n = 0.08; c->gyr_n = 0.004; c->ba_sigma = 0.02; c->bg_sigma = 0.002;
  c->triad_roll_deg = 90.0;
  c->line_dir_sign = -1;
}

// Output buffers (caller-allocated):
//  pose[11][7], speed_bias[11][9], ex_pose[7]                   initial estimates
//  pose_true[11][7], speed_bias_true[11][9]                      ground truth
//  point_start[P], point_nobs[P], point_obs[P*track_len*3], inv_depth[P]
//  line_start[L], line_nobs[L], line_obs[L*track_len*8], line_plk[L*6]
//  imu_samples[11][imu_rate_div*7] (entry 0 unused), imu_acc0[11][3], imu_gyr0[11][3]
int vplw_generate(uint64_t seed, const vplw_config* cfg, double t_start, double* pose, double* speed_bias,
                  double* ex_pose, double* pose_true, double* speed_bias_true, int* point_start, int* point_nobs,
                  double* point_obs, double* inv_depth, int* line_start, int* line_nobs, double* line_obs,
                  double* line_plk, double* imu_samples, double* imu_acc0, double* imu_gyr0) {
  Pcg64 rng(seed);
  const int NF = 11;
  const double kf_dt = cfg->kf_dt;
  State st[NF];
  M3 Rwc[NF];
  V3 twc[NF];
  for (int i = 0; i < NF; ++i) {
    st[i] = truth(t_start + kf_dt * i);
    Rwc[i] = mul(st[i].R, kRic);
    twc[i] = st[i].p + mul(st[i].R, kTic);
  }
  V3 ba{rng.normal() * cfg->ba_sigma, rng.normal() * cfg->ba_sigma, rng.normal() * cfg->ba_sigma};
  V3 bg{rng.normal() * cfg->bg_sigma, rng.normal() * cfg->bg_sigma, rng.normal() * cfg->bg_sigma};

  // ground truth + perturbed initial states
  M3 Rinit[NF];
  V3 pinit[NF];
  const double sth0 = cfg->pose_sigma_theta_deg * M_PI / 180.0;
  const M3 Edrift = expso3(V3{rng.normal() * sth0, rng.normal() * sth0, rng.normal() * sth0});
  const V3 drift_p{rng.normal() * cfg->pose_sigma_p, rng.normal() * cfg->pose_sigma_p, rng.normal() * cfg->pose_sigma_p};
  const V3 drift_v{rng.normal() * cfg->vel_sigma, rng.normal() * cfg->vel_sigma, rng.normal() * cfg->vel_sigma};
  for (int i = 0; i < NF; ++i) {
    double q[4];
    mat2quat(st[i].R, q);
    double* pt = pose_true + 7 * i;
    pt[0] = st[i].p.x; pt[1] = st[i].p.y; pt[2] = st[i].p.z; pt[3] = q[0]; pt[4] = q[1]; pt[5] = q[2]; pt[6] = q[3];
    double* sb = speed_bias_true + 9 * i;
    sb[0] = st[i].v.x; sb[1] = st[i].v.y; sb[2] = st[i].v.z; sb[3] = ba.x; sb[4] = ba.y; sb[5] = ba.z; sb[6] = bg.x; sb[7] = bg.y; sb[8] = bg.z;
    // estimator error = drift shared by the whole window (a rigid motion about frame 0 plus a velocity offset, drawn
    // once below) + a small independent part per frame (frame_sigma_scale of the same sigmas)
    const double fs = cfg->frame_sigma_scale;
    V3 dp{rng.normal() * cfg->pose_sigma_p * fs, rng.normal() * cfg->pose_sigma_p * fs, rng.normal() * cfg->pose_sigma_p * fs};
    double sth = cfg->pose_sigma_theta_deg * M_PI / 180.0 * fs;
    V3 dth{rng.normal() * sth, rng.normal() * sth, rng.normal() * sth};
    pinit[i] = st[0].p + mul(Edrift, st[i].p - st[0].p) + drift_p + dp;
    Rinit[i] = mul(mul(Edrift, st[i].R), expso3(dth));
    mat2quat(Rinit[i], q);
    double* p = pose + 7 * i;
    p[0] = pinit[i].x; p[1] = pinit[i].y; p[2] = pinit[i].z; p[3] = q[0]; p[4] = q[1]; p[5] = q[2]; p[6] = q[3];
    double* s = speed_bias + 9 * i;
    V3 vi = mul(Edrift, st[i].v) + drift_v;
    s[0] = vi.x + rng.normal() * cfg->vel_sigma * fs;
    s[1] = vi.y + rng.normal() * cfg->vel_sigma * fs;
    s[2] = vi.z + rng.normal() * cfg->vel_sigma * fs;
    for (int k = 3; k < 9; ++k) s[k] = 0.0;  // bias estimates start at zero
  }
  {
    double q[4];
    mat2quat(kRic, q);
    ex_pose[0] = kTic.x; ex_pose[1] = kTic.y; ex_pose[2] = kTic.z; ex_pose[3] = q[0]; ex_pose[4] = q[1]; ex_pose[5] = q[2]; ex_pose[6] = q[3];
  }

  // IMU samples: interval j covers (t_{j-1}, t_j]; sample k at t_{j-1} + (k+1) dt
  const int ns = cfg->imu_rate_div;
  const double dt = kf_dt / ns;
  auto meas = [&](double t, double* acc, double* gyr, bool noisy) {
    State s = truth(t);
    V3 a = mulT(s.R, s.a + V3{0, 0, kG}) + ba;
    V3 w = s.w_b + bg;
    if (noisy) {
      a = a + V3{rng.normal(), rng.normal(), rng.normal()} * cfg->acc_n;
      w = w + V3{rng.normal(), rng.normal(), rng.normal()} * cfg->gyr_n;
    }
    acc[0] = a.x; acc[1] = a.y; acc[2] = a.z; gyr[0] = w.x; gyr[1] = w.y; gyr[2] = w.z;
  };
  for (int j = 1; j < NF; ++j) {
    double t0 = t_start + kf_dt * (j - 1);
    meas(t0, imu_acc0 + 3 * j, imu_gyr0 + 3 * j, true);
    for (int k = 0; k < ns; ++k) {
      double* s = imu_samples + ((size_t)j * ns + k) * 7;
      s[0] = dt;
      meas(t0 + dt * (k + 1), s + 1, s + 4, true);
    }
  }
  for (int k = 0; k < 3; ++k) { imu_acc0[k] = 0; imu_gyr0[k] = 0; }

  const double fov_x = 1.0, fov_y = std::tan(30.0 * M_PI / 180.0);
  auto project = [&](int f, V3 pw, double& x, double& y) -> bool {
    V3 pc = mulT(Rwc[f], pw - twc[f]);
    if (pc.z < 0.2) return false;
    x = pc.x / pc.z; y = pc.y / pc.z;
    return std::fabs(x) <= fov_x && std::fabs(y) <= fov_y;
  };

  // points
  const int TL = cfg->track_len;
  for (int k = 0; k < cfg->n_points; ++k) {
    int s = k % (11 - TL + 1);   // track_len 6: start frames 0..5
    V3 pw{0, 0, 0};
    double d = 0;
    for (int tries = 0; tries < 10000; ++tries) {
      d = rng.uniform(2.0, 8.0);
      double x = rng.uniform(-fov_x, fov_x), y = rng.uniform(-fov_y, fov_y);
      pw = mul(Rwc[s], V3{x * d, y * d, d}) + twc[s];
      bool ok = true;
      for (int f = s; f < s + TL && ok; ++f) { double u, v; ok = project(f, pw, u, v); }
      if (ok) break;
    }
    point_start[k] = s;
    point_nobs[k] = TL;
    for (int f = 0; f < TL; ++f) {
      double u, v;
      project(s + f, pw, u, v);
      double* o = point_obs + ((size_t)k * TL + f) * 3;
      o[0] = u + rng.normal() * cfg->pix_sigma;
      o[1] = v + rng.normal() * cfg->pix_sigma;
      o[2] = 1.0;
    }
    // true depth along the first (noisy) observation ray, perturbed
    V3 pc = mulT(Rwc[s], pw - twc[s]);
    inv_depth[k] = (1.0 / pc.z) * (1.0 + rng.normal() * cfg->depth_rel_sigma);
  }

  // Manhattan triad.  A line whose interpretation plane contains the camera motion cannot be triangulated, so the three
  // families make equal angles (54.7 deg) with the mid-window direction of travel instead of with the optical axis; the
  // roll about that direction (triad_roll_deg, default 90) keeps every family away from the image plane so that all
  // vanishing points stay finite (camera-frame z components 0.82, 0.41, 0.41 at mid-window).
  M3 triad;
  {
    V3 tc = mulT(Rwc[5], st[5].v);           // direction of travel in the mid-window camera frame
    tc = tc * (1.0 / norm(tc));
    V3 zc{0, 0, 1};
    V3 xo = zc - tc * dot(zc, tc);            // optical axis made orthogonal to the travel direction
    xo = xo * (1.0 / norm(xo));
    V3 yo = cross(tc, xo);
    const double rr = cfg->triad_roll_deg * M_PI / 180.0;
    M3 Q0;
    for (int i = 0; i < 3; ++i) {
      double a = rr + i * 2.0 * M_PI / 3.0;
      V3 e = tc * (1 / std::sqrt(3.0)) + (yo * std::cos(a) + xo * std::sin(a)) * std::sqrt(2.0 / 3.0);
      Q0.m[0][i] = e.x; Q0.m[1][i] = e.y; Q0.m[2][i] = e.z;
    }
    triad = mul(Rwc[5], Q0);  // columns = world directions of the 3 line families
  }

  for (int k = 0; k < cfg->n_lines; ++k) {
    int s = k % (11 - TL + 1);   // track_len 6: start frames 0..5
    int ax = k % 3;
    V3 dir{triad.m[0][ax], triad.m[1][ax], triad.m[2][ax]};
    V3 e1{0, 0, 0}, e2{0, 0, 0};
    for (int tries = 0; tries < 10000; ++tries) {
      double d = rng.uniform(2.0, 8.0), len = rng.uniform(1.0, 3.0);
      double x = rng.uniform(-fov_x, fov_x), y = rng.uniform(-fov_y, fov_y);
      V3 mid = mul(Rwc[s], V3{x * d, y * d, d}) + twc[s];
      e1 = mid - dir * (len / 2);
      e2 = mid + dir * (len / 2);
      bool ok = true;
      for (int f = s; f < s + TL && ok; ++f) { double u, v; ok = project(f, e1, u, v) && project(f, e2, u, v); }
      if (ok) break;
    }
    line_start[k] = s;
    line_nobs[k] = TL;
    for (int f = 0; f < TL; ++f) {
      double* o = line_obs + ((size_t)k * TL + f) * 8;
      double u, v;
      project(s + f, e1, u, v);
      o[0] = u + rng.normal() * cfg->pix_sigma; o[1] = v + rng.normal() * cfg->pix_sigma;
      project(s + f, e2, u, v);
      o[2] = u + rng.normal() * cfg->pix_sigma; o[3] = v + rng.normal() * cfg->pix_sigma;
      V3 vp = mulT(Rwc[s + f], dir);
      if (vp.z < 0) vp = vp * -1.0;
      o[4] = vp.x; o[5] = vp.y; o[6] = vp.z; o[7] = cfg->with_vp ? 1.0 : 0.0;
    }
    // world Pluecker (n = p x d, v = d) -> orthonormal -> perturb angles -> start camera frame of the INITIAL pose
    V3 nw = cross(e1, dir), vw = dir;
    {
      const double want = cfg->line_dir_sign ? (double)cfg->line_dir_sign : ((k / 3) % 2 ? -1.0 : 1.0);
      const double dz = mulT(Rwc[s], dir).z;
      const double sgn = (dz * want >= 0) ? 1.0 : -1.0;
      nw = nw * sgn; vw = vw * sgn;
    }
    double o4[4];
    plk_to_orth(nw, vw, o4);
    for (int c = 0; c < 4; ++c) o4[c] += rng.normal() * cfg->orth_sigma;
    orth_to_plk(o4, nw, vw);
    M3 Rwc0 = mul(Rinit[s], kRic);
    V3 twc0 = pinit[s] + mul(Rinit[s], kTic);
    // L_c = [R^T (n - t x v) ; R^T v]
    V3 nc = mulT(Rwc0, nw - cross(twc0, vw));
    V3 vc = mulT(Rwc0, vw);
    double* pl = line_plk + 6 * k;
    pl[0] = nc.x; pl[1] = nc.y; pl[2] = nc.z; pl[3] = vc.x; pl[4] = vc.y; pl[5] = vc.z;
  }
  return 0;
}
}
