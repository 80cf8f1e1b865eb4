// Device-side random numbers and the twelve-integer normal sampler shared by the motion kernels.
#ifndef MCL_DEVICE_RNG_H
#define MCL_DEVICE_RNG_H
#include <cstdint>

#include <hip/hip_runtime.h>

namespace mcl {

// ---------------------------------------------------------------------------
// Counter-based generator (Philox 4x32, 10 rounds) for the device-side noise
// stream: 36 integers uniform on [-1000, 1000] per particle.
// ---------------------------------------------------------------------------
__device__ inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  uint64_t p0 = (uint64_t)M0 * c[0];
  uint64_t p1 = (uint64_t)M1 * c[2];
  uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  uint32_t n1 = (uint32_t)p1;
  uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ inline void philox4x32(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// Two independent standard normals from two 32-bit words: Box-Muller in float32 on the hardware's log2 / sqrt /
// sin / cos (sin and cos take their argument in turns, so the uniform number goes in as it is).  For streams
// that are the device's own (not numpy's), where the contract is the distribution.
__device__ inline void box_muller_f32(uint32_t a, uint32_t b, double* n0, double* n1) {
  const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0, 1)
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);            // [0, 1), in turns
  const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1)
  *n0 = (double)(rad * __builtin_amdgcn_cosf(u2));
  *n1 = (double)(rad * __builtin_amdgcn_sinf(u2));
}

// Sum of twelve scaled integers in the order numpy's pairwise reduction uses
// for n = 12 (mc_localization.py:23-24: `r.sum()`): eight accumulators
// combined as a tree, then the four remaining terms one by one.
__device__ inline double sum12(const int* k) {
  double r[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) r[i] = (double)k[i] * 0.001;
  double a = (r[0] + r[1]) + (r[2] + r[3]);
  double b = (r[4] + r[5]) + (r[6] + r[7]);
  double s = a + b;
  s = s + r[8];
  s = s + r[9];
  s = s + r[10];
  s = s + r[11];
  return s;
}

}  // namespace mcl
#endif
