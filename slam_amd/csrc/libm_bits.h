// exp(x) for non-positive x in fewer instructions than the device library's, and division by a fixed divisor; shared by the sector-query kernels' histogram-filter
// mode and tools/exp_check.hip, which compares it with the device library's exp.
#pragma once
#include <hip/hip_runtime.h>

namespace mcl {

// d = a * b + c as one v_fma_f64 whatever the compiler would rather do: with the addend a constant that lives in
// vector registers across a loop it otherwise picks the two-operand form, which overwrites its addend, and copies
// the constant with a 64-bit move before every step of a polynomial.
__device__ __forceinline__ double fma3(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// ... with the addend in scalar registers: a constant is then two scalar moves where it is used (they issue beside the
// vector instructions) and takes no vector registers at all
__device__ __forceinline__ double fma3s(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
  return d;
}

// exp(x) for x <= 0 (or NaN), bit for bit what the device library's exp returns there: the same reduction
// (n = rint(x log2 e), r = x - n ln2 in two parts), the same degree-11 polynomial, ldexp, zero below -1075 -- without
// the moves (see fma3) and without the overflow test that a non-positive argument cannot need.  SCALAR: the
// coefficients as scalar operands (fma3s) -- the histogram filter's update measured 8 % faster with them, the float64
// sector query, short of scalar registers as it is, 2 % slower.
// (tools/exp_check.hip compares the two over the whole range.)
template <bool SCALAR = false>
__device__ __forceinline__ double exp_nonpos(double x) {
  auto step = [](double a, double b, double c) { return SCALAR ? fma3s(a, b, c) : fma3(a, b, c); };
  const double n = __builtin_rint(x * __longlong_as_double(0x3ff71547652b82fell));
  double r = __builtin_fma(n, __longlong_as_double((long long)0xbfe62e42fefa39efull), x);
  r = __builtin_fma(n, __longlong_as_double((long long)0xbc7abc9e3b39803full), r);
  double p = step(r, __longlong_as_double(0x3e5ade156a5dcb37ll), __longlong_as_double(0x3e928af3fca7ab0cll));
  p = step(r, p, __longlong_as_double(0x3ec71dee623fde64ll));
  p = step(r, p, __longlong_as_double(0x3efa01997c89e6b0ll));
  p = step(r, p, __longlong_as_double(0x3f2a01a014761f6ell));
  p = step(r, p, __longlong_as_double(0x3f56c16c1852b7b0ll));
  p = step(r, p, __longlong_as_double(0x3f81111111122322ll));
  p = step(r, p, __longlong_as_double(0x3fa55555555502a1ll));
  p = step(r, p, __longlong_as_double(0x3fc5555555555511ll));
  p = step(r, p, __longlong_as_double(0x3fe000000000000bll));
  p = __builtin_fma(r, p, 1.0);
  p = __builtin_fma(r, p, 1.0);
  const double v = ldexp(p, (int)n);
  return (x < -1075.0) ? 0.0 : v;
}

// atan2(y, x), bit for bit the device library's: the quotient of the smaller by the larger magnitude, a degree-19
// polynomial in its square, the fix-ups by octant and the special values -- with three-operand fused multiply-adds
// (see fma3: the library's nineteen coefficients cost a 64-bit move each where they live in registers across a loop).
__device__ __forceinline__ double atan2_bits(double y, double x) {
  const double ax = __builtin_fabs(x), ay = __builtin_fabs(y);
  const double v = __builtin_fmax(ax, ay), u = __builtin_fmin(ax, ay);
  const double q = u / v;
  const double s = q * q;
  double p = fma3s(s, __longlong_as_double(0x3eeba404b5e68a13ll), __longlong_as_double((long long)0xbf23e260bd3237f4ull));
  p = fma3s(s, p, __longlong_as_double(0x3f4b2bb069efb384ll));
  p = fma3s(s, p, __longlong_as_double((long long)0xbf67952daf56de9bull));
  p = fma3s(s, p, __longlong_as_double(0x3f7d6d43a595c56fll));
  p = fma3s(s, p, __longlong_as_double((long long)0xbf8c6ea4a57d9582ull));
  p = fma3s(s, p, __longlong_as_double(0x3f967e295f08b19fll));
  p = fma3s(s, p, __longlong_as_double((long long)0xbf9e9ae6fc27006aull));
  p = fma3s(s, p, __longlong_as_double(0x3fa2c15b5711927all));
  p = fma3s(s, p, __longlong_as_double((long long)0xbfa59976e82d3ff0ull));
  p = fma3s(s, p, __longlong_as_double(0x3fa82d5d6ef28734ll));
  p = fma3s(s, p, __longlong_as_double((long long)0xbfaae5ce6a214619ull));
  p = fma3s(s, p, __longlong_as_double(0x3fae1bb48427b883ll));
  p = fma3s(s, p, __longlong_as_double((long long)0xbfb110e48b207f05ull));
  p = fma3s(s, p, __longlong_as_double(0x3fb3b13657b87036ll));
  p = fma3s(s, p, __longlong_as_double((long long)0xbfb745d119378e4full));
  p = fma3s(s, p, __longlong_as_double(0x3fbc71c717e1913cll));
  p = fma3s(s, p, __longlong_as_double((long long)0xbfc2492492376b7dull));
  p = fma3s(s, p, __longlong_as_double(0x3fc99999999952ccll));
  p = fma3s(s, p, __longlong_as_double((long long)0xbfd5555555555523ull));
  const double t = s * p;
  double a = __builtin_fma(q, t, q);
  const double half_pi = __longlong_as_double(0x3ff921fb54442d18ll), pi = __longlong_as_double(0x400921fb54442d18ll);
  const bool x_neg = __double_as_longlong(x) < 0;
  if (ax < ay) a = half_pi - a;
  if (x_neg) a = pi - a;
  if (y == 0.0) a = x_neg ? pi : 0.0;
  if (__builtin_isinf(x) && __builtin_isinf(y)) a = x_neg ? __longlong_as_double(0x4002d97c7f3321d2ll) : __longlong_as_double(0x3fe921fb54442d18ll);
  if (x != x || y != y) a = __longlong_as_double(0x7ff8000000000000ll);
  return __builtin_copysign(a, y);
}

// num / d for one divisor and many numerators, bit for bit what the compiler's division sequence gives for numerators of
// ordinary size: that sequence scales its operands only at the ends of the exponent range (v_div_scale_f64), and
// otherwise is a reciprocal refined twice -- which depends on the divisor alone and is formed once here --, a product,
// a remainder and one correction.  `usable` says whether the divisor is of ordinary size; the numerator's part: below
// 2^-969 the sequence would scale it, above 2^600 nothing here has been checked (callers divide small numbers).
struct DivBy {
  double d, r;
  bool usable;
};
__device__ __forceinline__ DivBy make_div_by(double d) {
  DivBy D;
  D.d = d;
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  D.r = r;
  const double ad = __builtin_fabs(d);
  D.usable = ad >= 0x1p-100 && ad <= 0x1p100;
  return D;
}
__device__ __forceinline__ double div_by(double num, const DivBy& D) {
  const double q0 = num * D.r;
  const double rem = __builtin_fma(-D.d, q0, num);
  return __builtin_fma(rem, D.r, q0);
}

// ... and the compiler's division where the divisor is not of ordinary size (zero, infinite, NaN included)
__device__ __forceinline__ double div_by_checked(double num, const DivBy& D) {
  return D.usable ? div_by(num, D) : num / D.d;
}

}  // namespace mcl
