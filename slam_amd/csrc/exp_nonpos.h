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

// exp(x) for x <= 0 (or NaN), bit for bit what the device library's exp returns there: the same reduction
// (n = rint(x log2 e), r = x - n ln2 in two parts), the same degree-11 polynomial, ldexp, zero below -1075 -- without
// the moves (see fma3) and without the overflow test that a non-positive argument cannot need.
// (tools/exp_check.hip compares the two over the whole range.)
__device__ __forceinline__ double exp_nonpos(double x) {
  const double n = __builtin_rint(x * __longlong_as_double(0x3ff71547652b82fell));
  double r = __builtin_fma(n, __longlong_as_double((long long)0xbfe62e42fefa39efull), x);
  r = __builtin_fma(n, __longlong_as_double((long long)0xbc7abc9e3b39803full), r);
  double p = fma3(r, __longlong_as_double(0x3e5ade156a5dcb37ll), __longlong_as_double(0x3e928af3fca7ab0cll));
  p = fma3(r, p, __longlong_as_double(0x3ec71dee623fde64ll));
  p = fma3(r, p, __longlong_as_double(0x3efa01997c89e6b0ll));
  p = fma3(r, p, __longlong_as_double(0x3f2a01a014761f6ell));
  p = fma3(r, p, __longlong_as_double(0x3f56c16c1852b7b0ll));
  p = fma3(r, p, __longlong_as_double(0x3f81111111122322ll));
  p = fma3(r, p, __longlong_as_double(0x3fa55555555502a1ll));
  p = fma3(r, p, __longlong_as_double(0x3fc5555555555511ll));
  p = fma3(r, p, __longlong_as_double(0x3fe000000000000bll));
  p = __builtin_fma(r, p, 1.0);
  p = __builtin_fma(r, p, 1.0);
  const double v = ldexp(p, (int)n);
  return (x < -1075.0) ? 0.0 : v;
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
  D.usable = d >= 0x1p-100 && d <= 0x1p100;
  return D;
}
__device__ __forceinline__ double div_by(double num, const DivBy& D) {
  const double q0 = num * D.r;
  const double rem = __builtin_fma(-D.d, q0, num);
  return __builtin_fma(rem, D.r, q0);
}

}  // namespace mcl
