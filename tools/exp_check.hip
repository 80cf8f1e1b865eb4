// Compares exp_nonpos (slam_amd/csrc/libm_bits.h) with the device library's exp, bit for bit, over non-positive
// arguments: a dense sweep of [-1100, 0], every binade down to the subnormals, and the special values; and div_by
// with the compiler's division for nine divisors and numerators -t^2/2 from 2^-800 to 2^80.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I slam_amd/csrc tools/exp_check.hip -o /tmp/exp_check && /tmp/exp_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cmath>
#include "libm_bits.h"

__global__ void check(uint64_t n, double lo, double step, unsigned long long* bad, double* first_bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = lo + (double)i * step;
  const double a = mcl::exp_nonpos(x), a2 = mcl::exp_nonpos<true>(x), b = exp(x);
  if (__double_as_longlong(a) != __double_as_longlong(b) || __double_as_longlong(a2) != __double_as_longlong(b)) {
    if (atomicAdd(bad, 1ull) == 0ull) *first_bad = x;
  }
}

__global__ void check_list(const double* xs, int n, unsigned long long* bad, double* first_bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = xs[i];
  const double a = mcl::exp_nonpos(x), a2 = mcl::exp_nonpos<true>(x), b = exp(x);
  const bool same = (__double_as_longlong(a) == __double_as_longlong(b) && __double_as_longlong(a2) == __double_as_longlong(b)) || (a != a && a2 != a2 && b != b);
  if (!same) {
    if (atomicAdd(bad, 1ull) == 0ull) *first_bad = x;
  }
}

__global__ void check_atan2(uint64_t n, uint64_t seed, unsigned long long* bad, double* first_bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // two hashes of the index: mantissas all over, exponents within +-40 of each other (and now and then far apart),
  // every combination of signs; a share of the pairs on the axes and the diagonals
  uint64_t h = (i + seed) * 0x9e3779b97f4a7c15ull; h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ull; h ^= h >> 32;
  uint64_t g = (h + 0x632be59bd9b4e019ull) * 0x94d049bb133111ebull; g ^= g >> 31;
  const int ex = (int)(h & 63) - 32, ey = (i % 97 == 0) ? (int)(g & 1023) - 512 : ex + (int)((g >> 10) & 15) - 8;
  double x = ldexp(1.0 + (double)(h >> 12) * 0x1p-52, ex), y = ldexp(1.0 + (double)(g >> 12) * 0x1p-52, ey);
  if (h & 0x100) x = -x;
  if (g & 0x200) y = -y;
  if (i % 1013 == 0) y = x;
  if (i % 1019 == 0) y = 0.0;
  if (i % 1021 == 0) x = 0.0;
  if (i % 1031 == 0) y = -y;
  const double a = mcl::atan2_bits(y, x), b = atan2(y, x);
  if (__double_as_longlong(a) != __double_as_longlong(b)) {
    if (atomicAdd(bad, 1ull) == 0ull) { first_bad[0] = y; first_bad[1] = x; }
  }
}

__global__ void check_atan2_list(const double* v, int n, unsigned long long* bad, double* first_bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * n) return;
  const double y = v[i / n], x = v[i % n];
  const double a = mcl::atan2_bits(y, x), b = atan2(y, x);
  const bool same = __double_as_longlong(a) == __double_as_longlong(b) || (a != a && b != b);
  if (!same) {
    if (atomicAdd(bad, 1ull) == 0ull) { first_bad[0] = y; first_bad[1] = x; }
  }
}

__global__ void check_div(uint64_t n, double d, unsigned long long* bad, double* first_bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // numerators -0.5 * t^2 as the mixture forms them, t from 2^-400 to 2^40 (geometric), and a linear sweep near zero
  const double t = (i & 1) ? ldexp(1.0 + (double)((i >> 1) & 0xfffff) / 1048576.0, (int)((i >> 21) % 440) - 400) : (double)i * 1e-9;
  const double num = -0.5 * (t * t);
  const mcl::DivBy D = mcl::make_div_by(d);
  const double a = mcl::div_by(num, D), b = num / d;
  // (a zero numerator gives +0 here and -0 there; what is asked is the exponential of it)
  const bool same = __double_as_longlong(a) == __double_as_longlong(b) || (a == 0.0 && b == 0.0) ||
                    (fabs(num) < 0x1p-969 && mcl::exp_nonpos(a) == exp(b));
  if (!same && D.usable) {
    if (atomicAdd(bad, 1ull) == 0ull) *first_bad = num;
  }
}

int main() {
  unsigned long long* bad; double* fb;
  hipMalloc(&bad, 8); hipMalloc(&fb, 8);
  hipMemset(bad, 0, 8); hipMemset(fb, 0, 8);
  const uint64_t n = 1ull << 31;
  check<<<(unsigned)(n / 256), 256>>>(n, -1100.0, 1100.0 / (double)n, bad, fb);
  // every binade: +-x for mantissas spread over the binade
  const int per = 64;
  static double xs[2200 * per];
  int m = 0;
  for (int e = -1074; e <= 10; ++e)
    for (int k = 0; k < per; ++k) xs[m++] = -std::ldexp(1.0 + (double)k / per, e);
  xs[m++] = 0.0; xs[m++] = -0.0; xs[m++] = -INFINITY; xs[m++] = NAN; xs[m++] = -1075.0; xs[m++] = -1074.999; xs[m++] = -745.2; xs[m++] = -708.4;
  double* dx; hipMalloc(&dx, sizeof(double) * m);
  hipMemcpy(dx, xs, sizeof(double) * m, hipMemcpyHostToDevice);
  check_list<<<(m + 255) / 256, 256>>>(dx, m, bad, fb);
  for (double d : {0.0225, 1.0, 0.01, 3.7e-5, 123.456, 2.5e-9, 9.9e12, -0.37, -4099.5})
    check_div<<<(unsigned)((1ull << 28) / 256), 256>>>(1ull << 28, d, bad, fb);
  unsigned long long hb0 = 0; hipMemcpy(&hb0, bad, 8, hipMemcpyDeviceToHost);
  double* fb2; hipMalloc(&fb2, 16); hipMemset(fb2, 0, 16);
  unsigned long long* bad2; hipMalloc(&bad2, 8); hipMemset(bad2, 0, 8);
  check_atan2<<<(unsigned)((1ull << 31) / 256), 256>>>(1ull << 31, 12345ull, bad2, fb2);
  const double sp[] = {0.0, -0.0, 1.0, -1.0, INFINITY, -INFINITY, NAN, 5e-324, -5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
                       -1.7976931348623157e308, 1e-300, -1e-300, 1e300, 3.0, -4.0, 0.5, 1e-17, -1e17};
  const int nsp = (int)(sizeof(sp) / sizeof(sp[0]));
  double* dsp; hipMalloc(&dsp, sizeof(sp)); hipMemcpy(dsp, sp, sizeof(sp), hipMemcpyHostToDevice);
  check_atan2_list<<<(nsp * nsp + 255) / 256, 256>>>(dsp, nsp, bad2, fb2);
  unsigned long long hb2 = 0; double hf2[2] = {0, 0};
  hipMemcpy(&hb2, bad2, 8, hipMemcpyDeviceToHost); hipMemcpy(hf2, fb2, 16, hipMemcpyDeviceToHost);
  printf("atan2: pairs checked: %llu + %d, differing: %llu%s\n", 1ull << 31, nsp * nsp, hb2, hb2 ? "" : " (identical)");
  if (hb2) printf("first differing pair: y = %.17g, x = %.17g\n", hf2[0], hf2[1]);
  unsigned long long hb; double hf;
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, fb, 8, hipMemcpyDeviceToHost);
  if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 2; }
  printf("exp and division: arguments checked: %llu + %d (+ 9 x 2^28 quotients), differing: %llu%s\n", (unsigned long long)n, m, hb, hb ? "" : " (identical)");
  if (hb) printf("first differing argument: %.17g\n", hf);
  return (hb || hb2) ? 1 : 0;
}
