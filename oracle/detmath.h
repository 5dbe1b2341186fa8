// TEST INFRASTRUCTURE -- sin / cos / atan / atan2 / acos in plain IEEE double arithmetic (+ - * / sqrt only).
// The vanishing-point stage of the reference is decided by values that sit exactly on cell borders of its sphere grid
// (vanishing_point_detection.cpp:137-147,298-311: longitude of vp2 = lambda = j degrees up to rounding), so its result
// depends on the last bit of the libm it is linked with.  To make "the same result" a meaningful statement the oracle
// fixes the elementary functions to the classic fdlibm kernels (argument reduction by pi/2 in two parts, the __kernel_sin /
// __kernel_cos / atan polynomials), every operation an IEEE double operation in the order written here; the device
// carries the same formulas (csrc/vp_kernels.h) and so reproduces them bit for bit.  Accuracy: a few ulp (checked
// against NumPy in tests/test_vpdetect.py), which is what a libm gives as well.
#pragma once
#include <cmath>

namespace det {

constexpr double kPio2Hi = 1.57079632673412561417e+00;   // first 33 bits of pi/2
constexpr double kPio2Lo = 6.07710050650619224932e-11;   // pi/2 - kPio2Hi
constexpr double kInvPio2 = 6.36619772367581382433e-01;
constexpr double kPi = 3.14159265358979311600e+00;
constexpr double kPiLo = 1.2246467991473531772e-16;

inline double ksin(double x) {   // |x| <= pi/4
  const double z = x * x;
  const double r = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                   z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
  return x + x * z * (-1.66666666666666324348e-01 + z * r);
}
inline double kcos(double x) {   // |x| <= pi/4
  const double z = x * x;
  const double r = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                   z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
  return 1.0 - (0.5 * z - z * r);
}
// x = n * pi/2 + r, |r| <= pi/4 (+ rounding); good for |x| < ~1e5
inline int reduce(double x, double* r) {
  const double fn = std::floor(x * kInvPio2 + 0.5);
  *r = (x - fn * kPio2Hi) - fn * kPio2Lo;
  return (int)fn;
}
inline double sin(double x) {
  double r;
  const int n = reduce(x, &r) & 3;
  return n == 0 ? ksin(r) : n == 1 ? kcos(r) : n == 2 ? -ksin(r) : -kcos(r);
}
inline double cos(double x) {
  double r;
  const int n = reduce(x, &r) & 3;
  return n == 0 ? kcos(r) : n == 1 ? -ksin(r) : n == 2 ? -kcos(r) : ksin(r);
}
inline double atan(double x) {
  const double ax = x < 0 ? -x : x;
  if (ax != ax) return x;
  double t, hi, lo;
  int id;
  if (ax < 0.4375) { t = ax; id = -1; hi = 0; lo = 0; }
  else if (ax < 0.6875) { t = (2.0 * ax - 1.0) / (2.0 + ax); id = 0; hi = 4.63647609000806093515e-01; lo = 2.26987774529616870924e-17; }
  else if (ax < 1.1875) { t = (ax - 1.0) / (ax + 1.0); id = 1; hi = 7.85398163397448278999e-01; lo = 3.06161699786838301793e-17; }
  else if (ax < 2.4375) { t = (ax - 1.5) / (1.0 + 1.5 * ax); id = 2; hi = 9.82793723247329054082e-01; lo = 1.39033110312309984516e-17; }
  else { t = -1.0 / ax; id = 3; hi = 1.57079632679489655800e+00; lo = 6.12323399573676603587e-17; }
  const double z = t * t, w = z * z;
  const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 +
                    w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
  const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 +
                    w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
  const double res = id < 0 ? t - t * (s1 + s2) : hi - ((t * (s1 + s2) - lo) - t);
  return x < 0 ? -res : res;
}
inline double atan2(double y, double x) {
  if (x != x || y != y) return x + y;
  if (y == 0.0) return x < 0 || (x == 0.0 && std::signbit(x)) ? (std::signbit(y) ? -kPi : kPi) : y;
  if (x == 0.0) return y < 0 ? -kPio2Hi - kPio2Lo : kPio2Hi + kPio2Lo;
  const double a = atan((y < 0 ? -y : y) / (x < 0 ? -x : x));   // first quadrant angle
  const double q = x > 0 ? a : kPi - (a - kPiLo);
  return y < 0 ? -q : q;
}
inline double acos(double x) {
  if (x >= 1.0) return 0.0;
  if (x <= -1.0) return kPi;
  return 2.0 * atan(std::sqrt((1.0 - x) / (1.0 + x)));
}

}  // namespace det
