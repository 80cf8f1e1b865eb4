// Sector range query + beam likelihood: one 64-lane wavefront per particle.
//
// What is computed (mc_localization.py:116-171, 235-240): for every beam k the
// squared distance to the nearest obstacle-cell centre with r2 <= z_max^2 whose
// bearing a = atan2(dy, dx) (wrapped to [0, 2pi)) satisfies
//     |beta_k - a| <= t   or   2pi - t <= |beta_k - a| <= 2pi + t,
// beta_k = deg_k*pi/180 + theta (never wrapped), then z_k = sqrt(min r2) and the
// weight prod_k p(z_k | z*_k) * pm.
//
// How.  For every 8x8-cell tile the library keeps the "neighbourhood list": the
// obstacle cells (row << 16 | col) of all tiles that can hold a cell within z_max
// of some point of that tile, nearest tile first, contiguous in memory.  A wave
// reads the slice of its particle's tile 64 cells at a time (next batch in
// flight), drops cells outside the disk with a float32 test and queues the rest
// in a 128-entry LDS ring; whenever 64 are queued they are matched together, so
// the expensive part runs with (almost) all lanes busy.  Per matched cell:
// exact float64 dx, dy, r2; a float32 bearing picks the window of the sorted
// beam table the cell can fall into (the table is extended by images of the
// top beams shifted by -2pi, so one window also covers the seam); the window's
// entries are read from LDS four at a time; a cell that is not closer than an
// entry's current minimum is dropped before any angle test; only where the
// float32 bearing is within 1e-5 rad of a sector edge is the float64 atan2
// evaluated and the reference predicate applied literally.  Each (cell, beam)
// match is a 64-bit LDS atomic min on the bit pattern of r2 (non-negative
// doubles order like unsigned integers).  The beam tables are staged in LDS
// once per block.  The kernel is bound by vector-instruction issue; its HBM
// traffic is the neighbourhood slices, ~1 KB per particle.
//
// Sector intervals (beam sets that allow them; not the staged query, not the
// histogram filter): the sector edges cut the circle into intervals inside
// which the set of sectors that hold a bearing does not change, a cell files
// ONE minimum under its interval, a beam reads the few intervals its sector
// spans, and a cell within rounding of an edge is decided in float64 against
// the one sector that ends and the one that starts there.  The float32
// throughput path runs as two launches: an instantiation that holds nothing
// but the interval path (IVONLY), and the general one over the particles the
// first leaves (headings for which the reference's one-turn predicate is not
// the circular one).
//
// Compiled with -ffp-contract=off.
#include <algorithm>
#include <type_traits>
#include <cstdlib>

#include "device_rng.h"
#include "mclhip_internal.h"
#include "libm_bits.h"

namespace mcl {

constexpr double kTwoPi = 6.283185307179586;  // == np.pi*2.0
#ifndef MCL_WPB
#define MCL_WPB 4
#endif
constexpr int kWavesPerBlock = MCL_WPB;
#ifndef MCL_GROUP
#define MCL_GROUP 4
#endif
constexpr int kGroup = MCL_GROUP;   // beams examined per step of the matching loop
constexpr int kCapSpan = 16;
constexpr int cap_buckets(int cap) { int n = 1; while (n < 2 * cap) n <<= 1; return n; }
constexpr int kEdgeQueue = 32;        // cells on a sector edge that a particle can park (interval path); more: the particle is done entry by entry
constexpr int kEdgeQueueWide = 64;    // ... in the interval-only kernel, whose queue also takes the unused part of the ring (see edgeq)
// histogram filter: (heading, beam) entries a lane takes per round of the mixtures, in straight-line code so that their
// float64 chains overlap.  936 entries are five rounds of 192 (2: eight rounds of 128; 4: four of 256): measured at
// 2048 x 2048 x 72, 24.3 ms with two, 23.65 with three, 23.7 with four.
#ifndef MCL_GRID_WAYS
#define MCL_GRID_WAYS 3
#endif
constexpr int kSubLists = 64;         // the list of left-over particles is kept as this many lists with a counter each (blocks b, b + 64, ... share one)
constexpr int kFiled = 16;            // particles an interval-only wave notes before it appends them to the list in one go
constexpr int kSeamSlots = 64;        // beams near the seam of the bearings whose cells there are asked one by one (interval-only kernel, headings outside the circular range)
constexpr int kWaveScratch = 128 + 2 * kEdgeQueue + 2 + kFiled + 2;   // u32 words of per-wave LDS scratch: the ring of cells waiting to be matched, the parked cells (u64 each), their count, the noted particles, their count

// ---- cross-lane helpers on the DPP path (no LDS round trip) ----------------
#define MCL_DPP_I32(old, v, ctrl, rmask) __builtin_amdgcn_update_dpp((old), (v), (ctrl), (rmask), 0xf, false)

// inclusive prefix sum over the 64 lanes
__device__ inline int wave_incl_scan(int v) {
  v += MCL_DPP_I32(0, v, 0x111, 0xf);  // row_shr:1
  v += MCL_DPP_I32(0, v, 0x112, 0xf);  // row_shr:2
  v += MCL_DPP_I32(0, v, 0x114, 0xf);  // row_shr:4
  v += MCL_DPP_I32(0, v, 0x118, 0xf);  // row_shr:8
  v += MCL_DPP_I32(0, v, 0x142, 0xa);  // row_bcast:15 into rows 1 and 3
  v += MCL_DPP_I32(0, v, 0x143, 0xc);  // row_bcast:31 into rows 2 and 3
  return v;
}

template <int CTRL, int RMASK>
__device__ inline double dpp_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = MCL_DPP_I32(0, (int)(b & 0xffffffffll), CTRL, RMASK);
  const int hi = MCL_DPP_I32(0, (int)(b >> 32), CTRL, RMASK);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// sum over the 64 lanes, valid in every lane (lanes that receive nothing add +0.0)
__device__ inline double wave_sum(double v) {
  v += dpp_f64<0x111, 0xf>(v);
  v += dpp_f64<0x112, 0xf>(v);
  v += dpp_f64<0x114, 0xf>(v);
  v += dpp_f64<0x118, 0xf>(v);
  v += dpp_f64<0x142, 0xa>(v);
  v += dpp_f64<0x143, 0xc>(v);
  return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(__double_as_longlong(v) >> 32), 63) << 32) |
                              (unsigned int)__builtin_amdgcn_readlane((int)__double_as_longlong(v), 63));
}

// 32-bit integer sum over the 64 lanes, valid in every lane
__device__ inline int wave_sum_i32(int v) {
  v += MCL_DPP_I32(0, v, 0x111, 0xf);
  v += MCL_DPP_I32(0, v, 0x112, 0xf);
  v += MCL_DPP_I32(0, v, 0x114, 0xf);
  v += MCL_DPP_I32(0, v, 0x118, 0xf);
  v += MCL_DPP_I32(0, v, 0x142, 0xa);
  v += MCL_DPP_I32(0, v, 0x143, 0xc);
  return __builtin_amdgcn_readlane(v, 63);
}

// Sum over the 64 lanes of the lanes' log2 totals, valid in every lane, in FIXED POINT: the whole part of every
// term in one 32-bit integer, its fraction in units of 2^-25 in another, both added exactly.  A float32 sum of 64
// terms of a few tens each is good to 1e-6 of the terms, which shows where terms of either sign cancel to a
// log-weight near zero -- the best particles; this one is good to 64 * 2^-26 = 1e-6 at worst (7e-8 on average)
// whatever the terms' size, costs what a float64 sum over the lanes does, and does not depend on the order in
// which lanes (or, with atomics, waves) are added.  Terms that are not finite, or beyond 2^20 in size, take the
// float64 sum.
constexpr double kFixedUnit = 33554432.0;   // 2^25
__device__ inline void to_fixed(double v, int* whole, unsigned int* frac) {
  const double fl = floor(v);
  *whole = (int)fl;
  *frac = (unsigned int)rint((v - fl) * kFixedUnit);   // [0, 2^25]: 64 of them fit 32 bits
}
__device__ inline double from_fixed(int whole, unsigned int frac) { return (double)whole + (double)frac * (1.0 / kFixedUnit); }
__device__ inline double wave_sum_fixed(double v) {
  if (__ballot(!(fabs(v) < 1048576.0)) != 0ull) return wave_sum(v);   // (also NaN)
  int whole;
  unsigned int frac;
  to_fixed(v, &whole, &frac);
  return from_fixed(wave_sum_i32(whole), (unsigned int)wave_sum_i32((int)frac));
}

// Bearing of (dx, dy) in [0, 2pi) (SIGNED: in (-pi, pi], one reflection less), float32, |error| < 1e-6 rad for finite
// inputs (0 for dx = dy = 0, as atan2 gives).  Only used to pick candidate beams: decisions within 1e-5 rad of a sector edge
// are taken again with the float64 atan2.  Odd minimax polynomial of degree 15 for atan on [0, 1]
// (1.4e-7), v_rcp_f32 for the ratio, three reflections.
template <bool SIGNED = false>
__device__ inline float bearing_f32(float dy, float dx) {
  const float ax = fabsf(dx), ay = fabsf(dy);
  const float mx = fmaxf(fmaxf(ax, ay), 1.0e-30f), mn = fminf(ax, ay);   // (dx = dy = 0: the ratio is 0, and so is the result)
  const float t = mn * __builtin_amdgcn_rcpf(mx);
  const float s = t * t;
  float p = -0.004054124001413584f;
  p = fmaf(p, s, 0.021861203014850616f);
  p = fmaf(p, s, -0.05590951070189476f);
  p = fmaf(p, s, 0.09641963243484497f);
  p = fmaf(p, s, -0.13908521831035614f);
  p = fmaf(p, s, 0.19946539402008057f);
  p = fmaf(p, s, -0.33329856395721436f);
  p = fmaf(p, s, 0.9999993443489075f);
  float r = t * p;
  r = ay > ax ? 1.5707964f - r : r;
  r = dx < 0.0f ? 3.1415927f - r : r;
  if constexpr (SIGNED) r = copysignf(r, dy);     // (-pi, pi]: for callers that reduce the angle themselves
  else r = dy < 0.0f ? 6.2831855f - r : r;
  return r;
}

// The same in TURNS, in (-1/2, 1/2]: the polynomial's coefficients divided by 2 pi, reflections about 1/4 and 1/2.
// For the interval-only kernel, which wants the bearing's place on a circle cut into sector intervals: the fraction
// of (bearing - heading - first edge), all in turns, is that place -- one subtraction and v_fract_f32 instead of two
// subtractions, a product, a floor and a fused multiply-add.  |error| < 2e-7 turns.
__device__ inline float bearing_turns_f32(float dy, float dx) {
  const float ax = fabsf(dx), ay = fabsf(dy);
  const float mx = fmaxf(fmaxf(ax, ay), 1.0e-30f), mn = fminf(ax, ay);
  const float t = mn * __builtin_amdgcn_rcpf(mx);
  const float s = t * t;
  float p = (float)(-0.004054124001413584 * 0.15915494309189535);
  p = fmaf(p, s, (float)(0.021861203014850616 * 0.15915494309189535));
  p = fmaf(p, s, (float)(-0.05590951070189476 * 0.15915494309189535));
  p = fmaf(p, s, (float)(0.09641963243484497 * 0.15915494309189535));
  p = fmaf(p, s, (float)(-0.13908521831035614 * 0.15915494309189535));
  p = fmaf(p, s, (float)(0.19946539402008057 * 0.15915494309189535));
  p = fmaf(p, s, (float)(-0.33329856395721436 * 0.15915494309189535));
  p = fmaf(p, s, (float)(0.9999993443489075 * 0.15915494309189535));
  float r = t * p;
  r = ay > ax ? 0.25f - r : r;
  r = dx < 0.0f ? 0.5f - r : r;
  return copysignf(r, dy);
}

// float64 bearing in [0, 2pi) exactly as the reference forms it (mc_localization.py:133-135).
// Kept out of line: it runs for a few cells in a thousand, and inlined its two dozen polynomial
// constants would sit in registers of the hot loop.
__device__ __attribute__((noinline)) double exact_bearing(double dy, double dx) {
  double a = atan2_bits(dy, dx);   // (the device library's atan2, bit for bit)
  if (a < 0.0) a += kTwoPi;
  return a;
}

// float64 logarithm for the arguments the fast path does not take (zero, subnormal, infinite, NaN);
// out of line for the same reason
__device__ __attribute__((noinline)) double slow_log(double v) { return log(v); }

// Profiling build (-DMCL_STAMPS): cycles spent per section, summed over waves, in P.stamps.
#ifdef MCL_STAMPS
#define STAMP(i) do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); acc[i] += t__ - t_last; t_last = t__; } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// Counting build (-DMCL_COUNTS): loop trip counts in P.stamps[8..15]; COUNT = once per wave, COUNTL = once per active lane.
#ifdef MCL_COUNTS
#define COUNT(i) do { if (lane == (int)__ffsll((unsigned long long)__ballot(1)) - 1) cnt[i] += 1; } while (0)
#define COUNTL(i) do { cnt[i] += 1; } while (0)
#else
#define COUNT(i) do { } while (0)
#define COUNTL(i) do { } while (0)
#endif

// Occupancy target: six waves per SIMD (80 VGPRs, nine of them spilled).  Measured at 2^20 x 180:
// 3 waves 2.32 ms, 4 waves 1.88 ms, 5 waves 1.61 ms, 6 waves 1.58 ms, 8 waves (34 spills) 2.27 ms.
#ifndef MCL_RAY_WAVES
#define MCL_RAY_WAVES 6
#endif
#ifndef MCL_FAST_LIKELIHOOD
#define MCL_FAST_LIKELIHOOD 1
#endif
// The float32-mixture instantiation needs fewer registers: seven waves per SIMD (72 VGPRs).  Measured at 2^20 x 180:
// 5 waves 1.51 ms, 6 waves 1.43 ms, 7 waves 1.40 ms, 8 waves 1.74 ms.
#ifndef MCL_FAST_INTERVALS
#define MCL_FAST_INTERVALS 1      // sector intervals in every instantiation but the staged query and the histogram filter
#endif
#ifndef MCL_FAST_WAVES
#define MCL_FAST_WAVES 7
#endif
// The interval-only instantiation.  Measured at 2^20 x 180: 7 waves 0.88 ms, 8 waves 0.856 ms, 9 waves 0.93 ms (its code
// compiled for nine waves and held to eight by its LDS request: 0.92 ms -- the eight-wave code is the better code).
#ifndef MCL_IVONLY_WAVES
#define MCL_IVONLY_WAVES 8
#endif
// Blocks of the particle-filter launch (each wave then loops over particles).  Seven blocks of four waves fit a
// compute unit, 1792 the chip; the launch wants several rounds of that, so that blocks which draw expensive particles
// do not leave the chip idle at the end, but not so many that the beam tables are staged too often.  Measured at
// 2^20 x 180 (ms): 2048 blocks 1.84, 4096 1.51, 8192 1.40, 16384 1.38, 32768 1.40, 65536 1.47, 262144 1.97.
#ifndef MCL_MAX_BLOCKS
#define MCL_MAX_BLOCKS 16384
#endif
// WPB: wavefronts per block.  Four for the particle filter; the histogram filter's table (all headings x
// beams) makes the per-wave LDS array large, and more waves behind one copy of the block-wide tables is
// what lifts its occupancy.
// RNOISE: the predicted ranges get device-drawn noise (augmented MCL at particle counts the host cannot draw for).
// FAST: float32 state, log-weights only, measured ranges finite and >= 0 (the throughput configuration): the beam
// mixture runs in float32 on the full-rate instructions, with the one quantity that needs more -- the difference
// between predicted and measured range, which the Gaussian term amplifies by d / sigma^2 -- formed from the exact
// float64 squared distance as (r2 - z*^2) / (z + z*), so that it carries float32 RELATIVE accuracy however close
// the two ranges are.
// IVONLY (with FAST): the kernel serves only the particles that take the sector-interval path and files the others --
// headings outside the interval path's range, and the rare particle with a cell on a sector edge that intervals cannot
// express -- in P.redo_list; nothing of the entry-by-entry test is compiled in (51 instead of 72 vector registers,
// hardly any scalar-register spill).  LISTED: the particles are those of P.list (what an IVONLY launch left over).
// CAP (with IVONLY): the LDS arrays are laid out for CAP beams, CAP intervals, sectors of at most kCapSpan intervals
// and a look-up of cap_buckets(CAP) entries, whatever the table at hand needs: every address is then a compile-time
// offset from the block's or the wave's base instead of a value kept in a scalar register.
// UNIF (with IVONLY), non-zero: the intervals are equally wide (evenly spaced beams): a bearing's interval and its edges
// are arithmetic, not two dependent LDS reads; positive: every sector spans exactly UNIF intervals (the loop over them
// has a compile-time length: 2 % of the kernel).
template <typename T, bool STAGE, bool PRODUCT, bool GRID = false, int WPB = kWavesPerBlock, bool RNOISE = false,
          bool FAST = false, bool IVONLY = false, bool LISTED = false, int CAP = 0, int UNIF = 0>
// (waves per SIMD as a closed range: with only a lower bound the compiler budgets scalar registers for the ten waves
// a small vector-register count would allow -- 80 instead of 96 -- and spills the rest into vector lanes)
#define MCL_RAY_OCC(T, GRID, WPB, FAST, IVONLY) (GRID ? WPB / 4 : (IVONLY ? MCL_IVONLY_WAVES : (FAST ? MCL_FAST_WAVES : (sizeof(T) == 4 ? MCL_RAY_WAVES : 4))))
__global__ __attribute__((amdgpu_flat_work_group_size(64 * WPB, 64 * WPB),
                          amdgpu_waves_per_eu(MCL_RAY_OCC(T, GRID, WPB, FAST, IVONLY), MCL_RAY_OCC(T, GRID, WPB, FAST, IVONLY)))) void raycast_kernel(
    RayParams P, const T* __restrict__ xs, const T* __restrict__ ys, const T* __restrict__ ths,
    const uint8_t* __restrict__ pm, int64_t n, double* __restrict__ w_out, double* __restrict__ logw_out,
    double* __restrict__ ranges_out, int32_t* __restrict__ hits_out, const QueryRecord* __restrict__ qrec) {
  // (P must stay the first argument and the list up to logw_out as it is: the interval-only instantiation reads both
  // through the argument block's address, see Pk and KArgs below)
  extern __shared__ __align__(16) unsigned char smem[];
  // LISTED: the launch is sized for every particle, the list is usually short: a block beyond it leaves at once
  if constexpr (LISTED) {
    // (the entry-by-entry launch, the last of a query: the lists' counts go where the host reads them, see hint_out --
    // before block 0 finds out whether it has anything else to do)
    if constexpr (FAST && !IVONLY) {
      if (blockIdx.x == 0 && P.hint_out != nullptr && threadIdx.x < 2 * kSubLists) P.hint_out[threadIdx.x] = P.hint_src[threadIdx.x];
    }
    if ((int64_t)(blockIdx.x / kSubLists) * WPB >= (int64_t)P.list_count[blockIdx.x % kSubLists]) return;
  }
  const int B = P.B, NB = P.NB;
  // GRID: the beam table holds the classes of (heading, beam) entries that share a bearing; NE entries
  const int NE = GRID ? P.g_NE : 0;
  const int NT = GRID ? P.g_B0 : B;                               // rows of the per-beam mixture tables
  // block-wide tables.  L*: the dimensions the arrays are laid out for -- the table's own, or the instantiation's capacity
  constexpr bool STATIC_L = IVONLY && CAP > 0;
  const int LB = STATIC_L ? CAP : B;
  const int LNT = GRID ? P.g_B0 : LB;
  double* s_ts = reinterpret_cast<double*>(smem);                 // [B] sorted beam angles
  double* s_zt = s_ts + LB;                                       // [NT] measured range (sorted order; GRID: per beam of one heading)
  double* s_el = s_zt + LNT;                                      // [NT] eta*lambda (GRID: exp(-lambda * z_meas))
  unsigned long long* s_rmin_all = reinterpret_cast<unsigned long long*>(s_el + LNT);
  const int NXP = IVONLY ? 0 : P.G + B + kGroup;                 // extended table + one group of padding
  // per wave: minima per table entry (GRID: or per interval, whichever array is longer), then (GRID) per entry, and a double per class
  const int NXQ = GRID ? (NXP > P.iv_NE + P.iv_cmax ? NXP : P.iv_NE + P.iv_cmax) : NXP;
  const int RS = NXQ + NE + (GRID ? B : 0);
  // With sector intervals (see RayParams::iv_*; not for the staged query, which also wants the hit cell, nor for the
  // histogram filter's classes) the wave's table holds the smallest squared distance per INTERVAL -- iv_NE entries and
  // the first iv_cmax again behind them, so that a sector that wraps reads on -- for a particle that takes the interval
  // path, and per table entry for one that does not: one array, sized for the larger of the two.
  constexpr bool IVK = !STAGE;   // (GRID: the class table's intervals, when every sector spans at most two -- the host says so by leaving iv_NE set)
  const int IVN = IVK ? P.iv_NE : 0;
  const int IVS = IVK ? P.iv_NE + P.iv_cmax : 0;
  const int LIVN = STATIC_L ? CAP : IVN;
  const int LNBI = STATIC_L ? cap_buckets(CAP) : P.iv_NB;
  // the interval-only kernel's arithmetic layouts, run over the list of what the first launch left, also serve the
  // headings outside the circular range (`quirky` below).  (An instantiation of its own: the same code inside the first
  // launch's kernel cost every particle 36 instructions of 550 -- tests, branches and reloads of spilled registers.)
  constexpr bool QK = IVONLY && LISTED && CAP > 0 && UNIF > 0;
  const int TS = STATIC_L ? CAP + kCapSpan + (QK ? kSeamSlots : 0) : (RS > IVS ? RS : IVS);   // entries of the per-wave table
  unsigned long long* s_aux_all = s_rmin_all + (size_t)WPB * TS;
  unsigned int* s_walk_all = reinterpret_cast<unsigned int*>(s_aux_all + (size_t)((STAGE || PRODUCT) ? WPB * RS : 0));
  float* s_relx = reinterpret_cast<float*>(s_walk_all + (size_t)WPB * kWaveScratch);  // [G + B] extended angle table
  unsigned short* s_perm = reinterpret_cast<unsigned short*>(s_relx + NXP);
  unsigned short* s_lut = s_perm + (IVONLY ? 0 : B);
  unsigned int* s_cls = reinterpret_cast<unsigned int*>(s_lut + (IVONLY ? 0 : ((NB + B + 1) & ~1) - B));   // [NE] class << 20 | heading << 10 | beam of entry e (GRID), 4-byte aligned
  // GRID: what the entry-by-entry test at a sector edge reads -- the entries of every class and the two terms of an entry's own angle
  double* s_ang = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(s_cls + NE) + 7) & ~(uintptr_t)7);   // [B0] beam angles, [D] heading angles (radians)
  // (behind the angles, per beam of the measurement: p_max * w_max and p_rand * w_rand of the mixture, which depend on
  // the measured range alone -- two reads per entry instead of eight instructions)
  double* s_tail = s_ang + (GRID ? P.g_B0 + P.g_D : 0);         // [2 B0]
  unsigned short* s_member = reinterpret_cast<unsigned short*>(s_tail + (GRID ? 2 * P.g_B0 : 0));   // [NE] entries by class
  unsigned short* s_first = s_member + NE;                                                     // [B + 1] start of every class in s_member
  // the interval tables stand where the class tables would (GRID: behind them)
  float* s_ivedge = GRID ? reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(s_first + B + 1) + 3) & ~(uintptr_t)3)
                         : reinterpret_cast<float*>(s_cls);                                    // [IVN + 1]
  unsigned short* s_ivbeam = reinterpret_cast<unsigned short*>(s_ivedge + LIVN + 1);           // [B]
  unsigned short* s_ivlut = s_ivbeam + LB;                                                      // [iv_NB]
  unsigned int* s_ivends = reinterpret_cast<unsigned int*>(s_ivlut + ((LNBI + LB + 1) & ~1) - LB);   // [IVN], 4-byte aligned

  // z*^2 as the FAST mixture wants it -- the largest float64 whose correctly rounded root does not exceed z*: the
  // short-reading term applies where sqrt(r2) <= z* (range_sensor.py:47), decided below on r2 itself, and fl(z* z*) can
  // fall an ulp short of the r2 whose root z* is (a particle that stands where the measurement was taken has z == z*
  // in every beam).  The difference r2 - z*^2 that the Gaussian term takes from this entry is the same to 2e-16.
  auto short_threshold = [](double zs) {
    double c2 = zs * zs;
    for (int it = 0; it < 4 && sqrt(__longlong_as_double(__double_as_longlong(c2) + 1)) <= zs; ++it) c2 = __longlong_as_double(__double_as_longlong(c2) + 1);
    for (int it = 0; it < 4 && c2 > 0.0 && sqrt(c2) > zs; ++it) c2 = __longlong_as_double(__double_as_longlong(c2) - 1);
    return c2;
  };
  constexpr bool ONE_PASS = IVONLY && CAP > 0 && UNIF > 0 && FAST && 64 * WPB >= CAP;
  if constexpr (ONE_PASS) {
    // the arithmetic layouts: a thread per table row, every load of the row asked for before the first is used (one
    // trip to memory per block, not one per table: what a short list's launch spends its time on), and none of the
    // tables that find an interval or a sector's start by lookup
    const int k = threadIdx.x;
    double ts = 0.0, zs = 0.0, el = 0.0;
    unsigned int ends = 0u;
    if (k < B) { ts = P.tab_sorted[k]; zs = P.zt[k]; el = P.etalam[k]; }
    if (k < IVN) ends = P.iv_ends[k];
    if (k < B) {
      s_ts[k] = ts;
      s_zt[k] = short_threshold(zs);
      reinterpret_cast<float2*>(s_el)[k] = make_float2((float)zs, (float)(el * P.w_short));
    }
    if (k < IVN) s_ivends[k] = ends;
  } else if constexpr (LISTED && FAST && !IVONLY) {
    // the launch over a short list spends a good part of its time on these tables: every load of a row asked for
    // before the first is used, rows of all tables side by side
    const int rows = NXP > NB ? NXP : NB;   // (NXP > B)
    for (int k = threadIdx.x; k < rows; k += blockDim.x) {
      double ts = 0.0, zs = 0.0, el = 0.0;
      unsigned short pk = 0, lk = 0;
      float rx = 1.0e30f;                    // padding: beyond any window
      if (k < B) { ts = P.tab_sorted[k]; pk = P.perm[k]; zs = P.zt[k]; el = P.etalam[k]; }
      if (k < P.G + B) rx = P.relx[k];
      if (k < NB) lk = P.lut[k];
      if (k < B) {
        s_ts[k] = ts; s_perm[k] = pk;
        s_zt[k] = short_threshold(zs);
        reinterpret_cast<float2*>(s_el)[k] = make_float2((float)zs, (float)(el * P.w_short));
      }
      if (k < NXP) s_relx[k] = rx;
      if (k < NB) s_lut[k] = lk;
    }
  } else {
  for (int k = threadIdx.x; k < B; k += blockDim.x) {
    s_ts[k] = P.tab_sorted[k];
    if (!IVONLY) s_perm[k] = P.perm[k];
  }
  if (!STAGE)
    for (int k = threadIdx.x; k < NT; k += blockDim.x) {
      if constexpr (FAST) {
        const double zs = P.zt[k];
        s_zt[k] = short_threshold(zs);
        reinterpret_cast<float2*>(s_el)[k] = make_float2((float)zs, (float)(P.etalam[k] * P.w_short));
      } else {
        s_zt[k] = P.zt[k];
        s_el[k] = P.etalam[k];
      }
    }
  }
  if (GRID) {
    for (int k = threadIdx.x; k < NE; k += blockDim.x) { s_cls[k] = P.g_cls[k]; s_member[k] = P.g_member[k]; }
    for (int k = threadIdx.x; k <= B; k += blockDim.x) s_first[k] = P.g_cls_first[k];
    for (int k = threadIdx.x; k < P.g_B0 + P.g_D; k += blockDim.x) s_ang[k] = P.g_ar[k];   // one array: beams, then headings
    for (int k = threadIdx.x; k < P.g_B0; k += blockDim.x) {
      const double xm = P.zt[k];                                  // range_sensor.py:55-67 for the measured range
      const double p_max = (fabs(xm - P.z_max) < 1e-6) ? 1.0 : 0.0;
      const double p_rand = ((0.0 <= xm) && (xm <= P.z_max)) ? P.p_rand : 0.0;
      s_tail[k] = p_max * P.w_max;
      s_tail[P.g_B0 + k] = p_rand * P.w_rand;
    }
  }
  if (!IVONLY && !(LISTED && FAST)) {
    for (int k = threadIdx.x; k < NXP; k += blockDim.x) s_relx[k] = (k < P.G + B) ? P.relx[k] : 1.0e30f;  // padding: beyond any window
    for (int k = threadIdx.x; k < NB; k += blockDim.x) s_lut[k] = P.lut[k];
  }
  if (!ONE_PASS && IVK && IVN > 0) {
    for (int k = threadIdx.x; k <= IVN; k += blockDim.x) s_ivedge[k] = P.iv_edge[k];
    for (int k = threadIdx.x; k < B; k += blockDim.x) s_ivbeam[k] = P.iv_beam[k];
    for (int k = threadIdx.x; k < P.iv_NB; k += blockDim.x) s_ivlut[k] = P.iv_lut[k];
    for (int k = threadIdx.x; k < IVN; k += blockDim.x) s_ivends[k] = P.iv_ends[k];
  }
  __syncthreads();

  // The map fields arrive as one eight-register load; copied out one by one (opaque to the compiler) they are separate
  // values to the register allocator, which otherwise spills and refills the eight as a block for every use of one.
  const uint64_t* nbr_off = P.nbr_off;
  const uint32_t* nbr_cells = P.nbr_cells;
  int ETW = P.ETW, ETH = P.ETH, rtx = P.rtx, rty = P.rty;
  if constexpr (!GRID) asm volatile("" : "+s"(nbr_off), "+s"(nbr_cells), "+s"(ETW), "+s"(ETH), "+s"(rtx), "+s"(rty));
  // likewise these arguments.  (Not the pose pointers: one that has gone through this is no longer known to point to
  // read-only memory, its loads become vector loads and the pose lives in vector registers -- measured slower.)
  asm volatile("" : "+s"(logw_out), "+s"(n));
  const int lane0 = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: particle index, pose and slice bounds live in scalar registers
  unsigned long long* const rmin = s_rmin_all + (size_t)wv * TS;   // per extended-table entry (or per interval): bits of the smallest r2
  unsigned long long* const emin = rmin + NXQ;                     // GRID: per entry, what only its own angle let in
  unsigned long long* const aux = s_aux_all + (size_t)wv * RS;     // STAGE: hit cell per entry; PRODUCT: p_k in caller order
  unsigned int* const ring = s_walk_all + (size_t)wv * kWaveScratch;   // [128] cells waiting to be matched
  // [EQ] edge << 32 | cell.  The interval-only kernel keeps no ring of cells: of the ring's 128 words it uses 48 for its
  // results (outbuf, outidx), and its queue -- which the second pass fills with the cells at the seam of the bearings
  // as well -- starts behind them and runs on through the other kernels' queue: 72 entries' room for 64.
  constexpr int EQ = IVONLY ? kEdgeQueueWide : kEdgeQueue;
  static_assert(48 + 2 * kEdgeQueueWide <= 128 + 2 * kEdgeQueue, "the wide queue ends where the narrow one does, at the latest");
  unsigned long long* const edgeq = reinterpret_cast<unsigned long long*>(ring + (IVONLY ? 48 : 128));
  unsigned int* const edge_cnt = ring + 128 + 2 * kEdgeQueue;
  // IVONLY: the particles this wave leaves to the launch that follows are noted here and appended to the list kFiled
  // at a time -- one atomic per wave and batch, not per particle: a cloud whose headings are all outside the interval
  // path's range would otherwise queue a million atomics on one counter.
  unsigned int* const filed = ring + 128 + 2 * kEdgeQueue + 2;
  unsigned int* const filed_cnt = filed + kFiled;
  auto flush_filed = [&](int lane_) {
    const int nf = __builtin_amdgcn_readfirstlane((int)filed_cnt[0]);
    if (nf > 0) {
      // (the list's pointers from the argument block, here where they are needed: pointers the compiler can trace to
      // the kernel's arguments are global ones -- a generic pointer's atomic would count against every counter)
      typedef const __attribute__((address_space(4))) RayParams* KP;
      KP Pk = (KP)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(Pk));
      int32_t* const redo_list_ = Pk->redo_list + (size_t)(blockIdx.x % kSubLists) * Pk->sub_cap;
      uint32_t* const redo_count_ = Pk->redo_count + (blockIdx.x % kSubLists);
      unsigned int base = 0u;
      if (lane_ == 0) base = atomicAdd(redo_count_, (unsigned int)nf);
      base = (unsigned int)__builtin_amdgcn_readfirstlane((int)base);
      if (lane_ < nf) redo_list_[base + lane_] = (int32_t)filed[lane_];
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): no store is left outstanding (see flush_out)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (lane_ == 0) filed_cnt[0] = 0u;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  };
  if (IVONLY && lane0 == 0) { filed_cnt[0] = 0u; edge_cnt[0] = 0u; }
  // IVONLY: the log-weights wait in LDS (where the other instantiations keep their ring of cells) and go to memory
  // kOutBatch at a time.  A store per particle would be outstanding beside the loads that are fetched ahead -- the next
  // particle's record and first batch of cells -- and the vector-memory counter only counts in order among accesses of
  // ONE kind: with a store in flight every wait for one of those loads becomes a wait for all of them.
  constexpr int kOutBatch = 16;
  double* const outbuf = reinterpret_cast<double*>(ring);   // [kOutBatch]
  unsigned int* const outidx = ring + 2 * kOutBatch;         // [kOutBatch] LISTED: whose result it is
  int out_slot = 0;                  // wave-uniform
  unsigned int out_mask = 0u;        // slots that hold a result (a particle left to the next launch has none)
  int64_t out_base = 0;              // the particle behind slot 0
  auto flush_out = [&](int lane_, int64_t step_) {
    if (out_mask != 0u) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      struct KArgs { RayParams P; const void* xs; const void* ys; const void* ths; const void* pm; int64_t n; double* w_out; double* logw_out; };
      typedef const __attribute__((address_space(4))) KArgs* KA;   // (repeats the head of this kernel's argument list, in order)
      KA ka = (KA)__builtin_amdgcn_kernarg_segment_ptr();
      asm volatile("" : "+s"(ka));
      double* const lo = ka->logw_out;
      if (lane_ < kOutBatch && ((out_mask >> lane_) & 1u)) lo[LISTED ? (int64_t)outidx[lane_] : out_base + (int64_t)lane_ * step_] = outbuf[lane_];
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): no store is left outstanding
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    out_mask = 0u;
    out_slot = 0;
  };
  unsigned long long* const imin = rmin;                               // the same array read as minima per sector interval
  // QK: per beam near the seam of the bearings, what the cells at the seam gave it (see `quirky`)
  unsigned long long* const smin = rmin + (STATIC_L ? CAP + kCapSpan : 0);
  const float seam_wf = (float)((2.0 * P.half + 1e-4) / kTwoPi);     // QK: bearings this close to the seam (turns) are asked beam by beam
  const int seam_wi = (int)ceil((2.0 * P.half + 1e-4) / kTwoPi * (double)P.iv_NE) + 2;   // ... in intervals, rounded up, and two for rounding
  const float iv_originf = (float)P.iv_origin, iv_inv_bwf = (float)((double)P.iv_NB / kTwoPi), iv_th_hif = (float)P.iv_th_hi, iv_th_midf = (float)P.iv_th_mid;
  const float iv_nearf = (float)kEdgeEps + (UNIF != 0 ? 4e-6f : 3e-6f);     // closer than this to a sector edge, float32 does not decide
  const float iv_near_uf = 0.5f - iv_nearf * (float)((double)P.iv_NE / kTwoPi);   // UNIF: the same band in interval widths, from the middle
  const float iv_inv_wf = (float)((double)P.iv_NE / kTwoPi);   // UNIF: intervals per radian
  const float iv_nf = (float)P.iv_NE;                          // ... and per turn
  const float iv_nf_below = __uint_as_float(__float_as_uint(iv_nf) - 1u);   // the largest float32 below the interval count
  const int lane_run = (int)(threadIdx.x & 63) * P.iv_stride;   // UNIF > 0: where the sector of the lane's first beam starts

  const unsigned long long r2max_bits = (unsigned long long)__double_as_longlong(P.r2max);
  const float disk_r = (float)(P.z_max * (1.0 + 1e-5) + 0.02 * (P.res_x < P.res_y ? P.res_x : P.res_y));
  const float disk_r2 = disk_r * disk_r;
  const double neg_half_inv_s2 = -0.5 / P.sigma2;
  const double neg_half_inv_s2_log2e = neg_half_inv_s2 * 1.4426950408889634;
  const double neg_lambda_log2e = -P.lambda * 1.4426950408889634;
  const double span = P.ts_max - P.ts_min;
  const double reach = P.half + 2.0 * kEdgeEps;   // candidate window half-width
  const double inv_two_pi = 1.0 / kTwoPi;
  const double inv_res_x = 1.0 / P.res_x, inv_res_y = 1.0 / P.res_y;
  const float reachf = (float)reach, inv_bwf = (float)P.inv_bw, ext_minf = (float)P.ext_min;
  const float ts_minf = (float)P.ts_min, spanf = (float)span, inv_two_pif = (float)inv_two_pi;
  const int G = P.G, NX = P.G + B;
  const bool seam = span < kTwoPi - 1e-9;   // images of the top beams exist: one window covers both sides of the seam
  const float sure_in = (float)(P.half - kEdgeEps), sure_out = (float)(P.half + kEdgeEps);

#ifdef MCL_STAMPS
  unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
#ifdef MCL_COUNTS
  unsigned long long cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const int64_t stride = (int64_t)gridDim.x * WPB;
  // The wave's next particle: every stride-th one, or (LISTED) the one every stride-th place of the list names.
  // (LISTED: block b serves list b % kSubLists together with blocks b + kSubLists, b + 2 kSubLists, ...)
  const int32_t* const my_list = LISTED ? P.list + (size_t)(blockIdx.x % kSubLists) * P.sub_cap : nullptr;
  if constexpr (LISTED) n = (int64_t)P.list_count[blockIdx.x % kSubLists];
  const int64_t step = LISTED ? (int64_t)(gridDim.x / kSubLists) * WPB : stride;
  int64_t place = (LISTED ? (int64_t)(blockIdx.x / kSubLists) : (int64_t)blockIdx.x) * WPB + wv - step;
  int64_t p = 0, p_next = 0;
  auto next_particle = [&]() -> bool {
    place += step;
    if (place >= n) return false;
    p_next = LISTED ? (int64_t)my_list[place] : place;
    return true;
  };
  T nx = (T)0, ny = (T)0, nth = (T)0;
  unsigned char npm = 0;
  // IVONLY: everything about the next particle comes in the motion kernel's record (one eight-dword scalar load)
  // (ONE vector load: lane l reads dword l & 7 of the record, and the particle's turn starts with seven v_readlane that
  // spread them into scalar registers.  A vector load waits behind the in-order vector-memory counter until its
  // register is used; scalar loads share a counter with LDS, so the first LDS read after one would wait for it.)
  unsigned int n_rec = 0u;
  // (the launch over the left-over list: the slice of cells the motion kernel noted in the particle's record -- start,
  // length of the prefix of its tile's list that the pose can reach, flags -- comes with the pose, a particle ahead:
  // no read of the tile's bounds behind the pose, and the prefix instead of the whole list)
  uint4 n_slice = make_uint4(0u, 0u, 0u, 0u);
  auto fetch_next = [&]() {
    if constexpr (IVONLY) {
      n_rec = reinterpret_cast<const unsigned int*>(qrec + p_next)[lane0 & 7];
    } else {
      nx = xs[p_next]; ny = ys[p_next]; nth = ths[p_next];
      if (!STAGE && !GRID) npm = pm[p_next];
      if constexpr (LISTED && FAST) {
        if (qrec != nullptr) n_slice = reinterpret_cast<const uint4*>(qrec + p_next)[1];
      }
    }
  };
  bool have = next_particle();
  if (have) fetch_next();
  // IVONLY: the first batch of the next particle's list, fetched while the particle before it is still at its beams.
  // (Four batches ahead -- a particle's whole list in flight at once -- measured -0.8 % in the global regime and
  // +1.5 % in the tracking regime: the kernel does not wait for memory.)
  unsigned int e_pref = 0u;
  if constexpr (IVONLY) {
    if (have && __builtin_amdgcn_readlane((int)n_rec, 6) != 0)
      e_pref = P.nbr_cells[((uint64_t)(unsigned int)__builtin_amdgcn_readlane((int)n_rec, 4) | ((uint64_t)(unsigned int)__builtin_amdgcn_readlane((int)n_rec, 5) << 32)) + lane0];
    // (nothing outstanding when the loop is entered: what the compiler must assume at the head of the loop is then what
    // an iteration leaves behind -- the prefetched batch and the result's store -- and it waits where those are used)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  }
  bool entry_by_entry = false;   // the particle at hand is being done again without sector intervals
  while (have) {
    STAMP(7);
    p = p_next;
    if constexpr (IVONLY) { if (out_slot == 0) out_base = p; }
    // the lane index is made opaque once per particle: comparisons against it are then formed where they are used
    // (one instruction) instead of being kept as 64-bit masks in scalar registers that spill
    int lane = lane0;
    if constexpr (!GRID) asm volatile("" : "+v"(lane));
    auto rec_word = [&](int k) { return IVONLY ? (unsigned int)__builtin_amdgcn_readlane((int)n_rec, k) : 0u; };
    if constexpr (IVONLY) { nx = (T)__uint_as_float(rec_word(0)); ny = (T)__uint_as_float(rec_word(1)); nth = (T)__uint_as_float(rec_word(2)); }
    const float tho = __uint_as_float(rec_word(3));
    const uint64_t c_first = (uint64_t)rec_word(4) | ((uint64_t)rec_word(5) << 32);
    const unsigned int c_ncells = rec_word(6), c_flags = rec_word(7);
    const double x = (double)nx, y = (double)ny, th = (double)nth;
    const bool on_free = IVONLY ? (c_flags & 1u) != 0u : npm != 0;
    // headings for which the reference's predicate is the circular one take the interval path (wave-uniform); a
    // particle is done again entry by entry when a cell on a sector edge cannot be filed under intervals (see below)
    // QK: a heading outside the range in which the reference's predicate is the circular one.  What differs there
    // (mc_localization.py:143-149 recognises beam angle + heading - bearing within a turn of zero, no further):
    //   a beam whose sector does not touch the seam of the bearings (bearing 0 = 2 pi) sees every cell of its sector
    //   the same number of turns away, floor((beam angle + heading) / 2 pi): the beam keeps what the intervals hold if
    //   that is -1, 0 or 1 and sees nothing otherwise;
    //   a cell within two half-angles of the seam is not filed under an interval: it is parked, and asked with the
    //   reference's own predicate for every beam whose sector may hold it; what it gives goes to a minimum per beam
    //   (smin, for the kSeamSlots beams from seam_k0 on).  Sectors that touch the seam hold only such cells.
    const bool quirky = QK && (c_flags & 4u) != 0u;
    const float seam_w = quirky ? seam_wf : -1.0f;
    int seam_k0 = 0, seam_turns = 0;      // quirky: the first beam with a slot in smin; the heading's whole turns ...
    float seam_thf = 0.0f;                // ... and what is left of it
    const float seam_lof = (float)P.half + 1e-5f, seam_hif = (float)(kTwoPi - P.half) - 1e-5f;
    const bool use_iv = IVONLY ? (c_flags & (QK ? 6u : 2u)) != 0u
                               : (IVK && IVN > 0 && !entry_by_entry && fabsf((float)th - iv_th_midf) <= iv_th_hif);
    entry_by_entry = false;
    bool lane_redo = false;
    bool lane_parked = false;   // this lane has parked a cell
    // tile that holds the particle; poses far outside any map (or NaN) visit no tile
    // (a product instead of a division: an index off by one at an exact tile edge is covered by the
    // ring table's two cells of slack)
    const double fxc = x * inv_res_x, fyc = y * inv_res_y;
    const bool pose_ok = (fabs(fxc) < 1.0e6) && (fabs(fyc) < 1.0e6);
    const int tcx = pose_ok ? ((int)floor(fxc)) >> (3 - kQueryShift) : 0;   // query tile
    const int tcy = pose_ok ? ((int)floor(fyc)) >> (3 - kQueryShift) : 0;
    // the particle's slice of neighbourhood cells; NaN / infinite headings match no beam
    uint64_t first_cell = 0ull;
    unsigned int n_cells = 0u;
    if constexpr (IVONLY) {
      // (the record holds the slice of the particle's tile; no cells for a particle this kernel does not serve)
      first_cell = c_first;
      n_cells = c_ncells;
    } else {
      bool noted = false;
      if constexpr (LISTED && FAST) {
        // (bits 1 and 2 of the flags: the record holds the slice -- see QueryRecord)
        noted = qrec != nullptr && (n_slice.w & 6u) != 0u;
        if (noted) { first_cell = (uint64_t)n_slice.x | ((uint64_t)n_slice.y << 32); n_cells = n_slice.z; }
      }
      const int ex = tcx + rtx, ey = tcy + rty;
      if (!noted && pose_ok && fabs(th) < INFINITY && ex >= 0 && ex < ETW && ey >= 0 && ey < ETH && (!IVONLY || use_iv)) {
        const uint64_t* no = nbr_off + ((size_t)ey * ETW + ex);
        first_cell = no[0];
        n_cells = (unsigned int)(no[1] - first_cell);
      }
    }
    const float xf = (float)x, yf = (float)y;
    const float thf = (float)th;
    if constexpr (IVONLY) {
      // the record after this one: asked for now (this particle's has been taken apart above), it arrives while this
      // particle's cells are matched.  Unconditionally -- after the last particle its own record once more -- so that
      // the number of loads in flight is the same on every path and the compiler can wait for the older one alone.
      have = next_particle();
      if (!have) p_next = p;
      fetch_next();
    }

    // one cell against the beam table
    auto match_cell = [&](unsigned int e, bool valid, int phase) {
      const int col = (int)(e & 0xffffu), row = (int)(e >> 16);
      const double ox = (double)((float)col * P.res_xf);  // mc_localization.py:68-71: float32 products
      const double oy = (double)((float)row * P.res_yf);
      const double dx = ox - x, dy = oy - y;              // :124-125, float64
      // :126.  The float32 likelihood takes the fused form (one rounding less than the reference's sum of two rounded
      // products: a last-bit difference in float64, nine orders of magnitude below that path's tolerance)
      const double r2 = FAST ? fma(dx, dx, dy * dy) : dx * dx + dy * dy;
      COUNT(1);
      bool hit = valid && r2 <= P.r2max;       // :127
      // (the interval path reduces the angle itself.  FAST: pose and cell coordinates are float32 numbers, so their
      // float32 difference is the float64 one rounded -- the same operand without two conversions)
      const float aaf = IVONLY ? bearing_turns_f32((float)row * P.res_yf - yf, (float)col * P.res_xf - xf)
                               : (FAST ? bearing_f32<false>((float)row * P.res_yf - yf, (float)col * P.res_xf - xf)
                                       : bearing_f32<false>((float)dy, (float)dx));
      const unsigned long long r2bits = (unsigned long long)__double_as_longlong(r2);
      if constexpr (IVK) {
        if (use_iv) {
          // the interval that holds the bearing: one minimum serves every beam whose sector spans it.  Only a
          // bearing within rounding of a sector edge goes through the beam-by-beam test below.
          // the bearing's place on the circle that starts at interval edge 0: in turns, [0, 1), for the interval-only
          // kernel; in radians, [0, 2 pi) up to rounding (the ends are edges), for the others
          float a, place = 0.0f;
          if constexpr (IVONLY) {
            place = __builtin_amdgcn_fractf(aaf - tho);
            a = place * 6.2831853071795865f;
          } else {
            a = (aaf - thf) - iv_originf;                             // in (-4 pi, 5 pi) for the headings of this path
            a = fmaf(-6.2831853071795865f, floorf(a * 0.15915494309189535f), a);
          }
          int m;
          float e0 = 0.0f, e1 = 0.0f, uf = 0.0f;
          if constexpr (UNIF != 0) {
            // (the place is not negative and the conversion truncates towards zero: only the upper end needs a clamp; NaN
            // converts to 0.  The product is within 4e-7 rad of the true place: the band below allows for it, and a
            // bearing that lands one interval off is within the band of the edge between the two, which is what counts)
            if constexpr (IVONLY) {
              const float au = fminf(place * iv_nf, iv_nf_below);   // [0, N): the conversion needs no clamp, and
              m = (int)au;
              uf = __builtin_amdgcn_fractf(au);                      // the place inside the interval is its fraction
            } else {
              const float au = a * iv_inv_wf;
              m = (int)au;
              m = m > IVN - 1 ? IVN - 1 : m;
              uf = au - (float)m;      // place inside the interval, in interval widths
            }
          } else {
            int bkt = (int)(a * iv_inv_bwf);
            bkt = bkt < 0 ? 0 : (bkt > P.iv_NB - 1 ? P.iv_NB - 1 : bkt);
            m = s_ivlut[bkt];                   // a bucket is half the average interval: at most a step or two
            e0 = s_ivedge[m]; e1 = s_ivedge[m + 1];
            if (e1 <= a && m + 1 < IVN) {
              ++m; e0 = e1; e1 = s_ivedge[m + 1];
              while (e1 <= a && m + 1 < IVN) { ++m; e0 = e1; e1 = s_ivedge[m + 1]; }
            }
          }
          bool at_lower, near_edge;
          if constexpr (UNIF != 0) {
            const bool inside = fabsf(uf - 0.5f) <= iv_near_uf;
            near_edge = !inside;                                                    // also NaN
            if constexpr (QK) near_edge = near_edge || fabsf(aaf) < seam_w;         // at the seam of the bearings (never, unless `quirky`)
            at_lower = !(uf >= 0.5f);
          } else {
            at_lower = !(a - e0 >= iv_nearf);                                       // also NaN
            near_edge = at_lower || !(e1 - a >= iv_nearf);
          }
          // (nested, not two conditions side by side: the mask of the second branch is then the first one's
          // complement, formed by a scalar instruction instead of a second vector comparison)
          if (hit) {
          if (!near_edge) atomicMin(&imin[m], r2bits);
          else {
            // within rounding of edge E: parked, decided in float64 after the walk (rare; keeps the call out of this loop)
            COUNTL(2);
            unsigned int E = (unsigned int)(at_lower ? m : (m + 1 == IVN ? 0 : m + 1));
            if constexpr (QK) { if (fabsf(aaf) < seam_w) E = 0x80000000u | (unsigned int)m; }   // a cell at the seam: its interval
            const unsigned int pos = atomicAdd(edge_cnt, 1u);
            if (pos < (unsigned int)EQ) edgeq[pos] = ((unsigned long long)E << 32) | e;
            else lane_redo = true;
            lane_parked = true;
          }
          }
          hit = false;
        }
      }
      if (!IVONLY && hit) {
        COUNTL(2);
        const unsigned long long cellbits = (unsigned long long)((unsigned int)row * (unsigned int)P.W + (unsigned int)col);
        // beams whose table angle lies within `reach` of aa - th + c*2pi, c in {-1, 0, 1}
        double psi = 0.0;
        float psif = 0.0f;
        int c_lo, c_hi;
        if constexpr (FAST) {
          // float32 throughout: the heading of a float32 state is exact in float32, the bearing is a float32 already,
          // and which windows are looked at has 1e-5 rad of slack (reach against the edges that are tested)
          psif = (aaf - thf) - ts_minf;
          c_lo = (int)ceilf((-reachf - psif) * inv_two_pif);
          c_hi = (int)floorf((spanf + reachf - psif) * inv_two_pif);
        } else {
          psi = ((double)aaf - th) - P.ts_min;
          c_lo = (int)ceil((-reach - psi) * inv_two_pi);
          c_hi = (int)floor((span + reach - psi) * inv_two_pi);
        }
        c_lo = c_lo < -1 ? -1 : c_lo;
        c_hi = c_hi > 1 ? 1 : c_hi;
        double a_exact = 0.0;
        bool have_exact = false;
        STAMP(4);
        // With the seam images in the table every lane needs one pass (its lowest window; the next
        // window's beams appear as images inside it), so the lanes of a batch stay together.
#pragma unroll 1
        for (int c = c_lo; c <= c_hi; c += (seam ? 2 : 1)) {
          // window centre relative to the smallest table angle; small whenever the window can
          // hold a beam, so float32 is enough for everything but the edge cases
          COUNT(3);
          const float centref = FAST ? psif + (float)c * 6.2831853071795865f : (float)(psi + (double)c * kTwoPi);
          const float lof = centref - reachf;
          int i0 = 0;
          if (lof > ext_minf) {
            int bkt = (int)((lof - ext_minf) * inv_bwf) - 1;
            bkt = bkt < 0 ? 0 : (bkt > NB - 1 ? NB - 1 : bkt);
            i0 = s_lut[bkt];
          }
          if (!(seam && c + 1 <= c_hi) && i0 < G) i0 = G;   // images stand for window c + 1: skip them if it is not allowed
          // beams are visited kGroup at a time: all LDS reads of a group are issued
          // before the first use, so their latency is paid once per group, not once per beam
#pragma unroll 1
          for (int kb = i0; kb < NX; kb += kGroup) {
            COUNT(4);
            COUNTL(5);
            float eo[kGroup];
            unsigned long long rm[kGroup];
#pragma unroll
            for (int j = 0; j < kGroup; ++j) {               // tables are padded: kb + j needs no bound check
              eo[j] = s_relx[kb + j] - centref;               // bearing offset from the beam's axis
              rm[j] = rmin[kb + j];
            }
            bool any_edge = false;
#pragma unroll
            for (int j = 0; j < kGroup; ++j) {
              const float ae = fabsf(eo[j]);
              const bool live = (STAGE && phase == 1) ? (rm[j] == r2bits) : (r2bits < rm[j]);
              const bool inside = ae <= sure_in, near = ae < sure_out;
              any_edge = any_edge | (live & (near != inside));   // too close to a sector edge for float32
              if (live & inside) {
                COUNTL(7);
                if (!STAGE || phase == 0) atomicMin(&rmin[kb + j], r2bits);
                else atomicMin(&aux[kb + j], cellbits);       // lowest row-major index among equally near cells
              }
            }
            if (any_edge) {                                 // rare: exact float64 bearing
              COUNT(6);
              if (!have_exact) {
                a_exact = exact_bearing(dy, dx);            // :133-135
                have_exact = true;
              }
#pragma unroll 1
              for (int j = 0; j < kGroup; ++j) {
                const int ie = kb + j;                      // entry of the extended table
                const float ae = fabsf(s_relx[ie] - centref);
                const unsigned long long rmj = rmin[ie];
                const bool live = (STAGE && phase == 1) ? (rmj == r2bits) : (r2bits < rmj);
                if (live && ae > sure_in && ae < sure_out) {
                  const int ke = ie - G + (ie < G ? B : 0);   // its beam, sorted order
                  if constexpr (GRID) {
                    // every entry of the class with its own angle (robot.py:252-258)
                    for (int m = s_first[ke]; m < s_first[ke + 1]; ++m) {
                      const unsigned int me = s_member[m], ck = s_cls[me];
                      const double beta = s_ang[ck & 0x3ffu] + s_ang[P.g_B0 + ((ck >> 10) & 0x3ffu)];   // robot.py:252
                      const double de = fabs((beta + th) - a_exact);
                      if ((de <= P.half) || (de >= P.two_pi_lo && de <= P.two_pi_hi)) atomicMin(&emin[me], r2bits);
                    }
                  } else {
                    const double beta = s_ts[ke] + th;          // :143
                    const double de = fabs(beta - a_exact);     // :148
                    if ((de <= P.half) || (de >= P.two_pi_lo && de <= P.two_pi_hi)) {  // :149
                      if (!STAGE || phase == 0) atomicMin(&rmin[ie], r2bits);
                      else atomicMin(&aux[ie], cellbits);
                    }
                  }
                }
              }
            }
            const bool beyond = eo[kGroup - 1] > reachf;    // the table ascends: nothing further can be in reach
            if (beyond) break;
          }
        }
      }
    };

    STAMP(0);
    {
      if constexpr (STATIC_L) {
        // the fixed layout's table holds CAP + kCapSpan entries: the whole of CAP is cleared, whatever part is in use
#pragma unroll
        for (int j = 0; j < CAP; j += 64) rmin[lane + j] = r2max_bits;
      } else {
        const int n_clear = (use_iv && !GRID) ? IVN : RS;   // (GRID: the minima per entry behind the intervals as well)
        for (int k = lane; k < n_clear; k += 64) {
          rmin[k] = r2max_bits;
          if (STAGE) aux[k] = 0xffffffffull;
        }
      }
      if constexpr (!IVONLY) { if (IVK && lane == 0) edge_cnt[0] = 0u; }   // (IVONLY: cleared once, and again by whoever used it)
    }
    for (int phase = 0; phase < (STAGE ? 2 : 1); ++phase) {
      int q_head = 0, q_count = 0;   // wave-uniform state of the ring of cells waiting to be matched
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // Cells of a fetched batch that a float32 distance test (a little generous; the exact
      // float64 test follows in match_cell) puts within z_max are queued in a 128-entry LDS
      // ring; whenever 64 are waiting they are matched together, so that the expensive part
      // runs with (almost) every lane busy although the slice holds many cells out of range.
      unsigned int e_next = IVONLY ? e_pref : ((lane < (int)n_cells) ? nbr_cells[first_cell + lane] : 0u);
      STAMP(2);
      if constexpr (IVONLY) {
        // The interval path matches the fetched batches as they come: a list holds about 200 cells for the 130 within
        // z_max, and without the entry-by-entry test a cell outside the disc costs less in a half-empty batch than the
        // compaction of every batch does (0.884 -> 0.856 ms at 2^20 x 180).
        // (a list is whole batches of 64, padded with a cell out of every pose's reach: no per-lane bounds)
        // (none for a particle that is left to the launch that follows)
        const unsigned int n_list = use_iv ? (unsigned int)__builtin_amdgcn_readfirstlane((int)n_cells) : 0u;
        // (the loads go through the argument block's own pointer: one that has been made opaque is a generic pointer to
        // the compiler, and loads through those count against the LDS counter as well)
        // (the loads go through the argument block's own pointer: one that has been made opaque is a generic pointer to
        // the compiler, and loads through those count against the LDS counter as well)
        if constexpr (QK) {
          // the pass over what the first launch left is bound by the latency of a particle's chain of loads, not by
          // issue: three batches in flight, asked for whether the list has them or not (the array is allocated that
          // far beyond its last list; the count of outstanding loads is then the same on every path)
          unsigned int e_b = P.nbr_cells[first_cell + 64 + lane], e_c = P.nbr_cells[first_cell + 128 + lane];
          for (unsigned int sb = 0; sb < n_list; sb += 64) {
            const unsigned int e = e_next;
            COUNT(0);
            e_next = e_b; e_b = e_c;
            e_c = P.nbr_cells[first_cell + sb + 192 + lane];
            match_cell(e, true, phase);
          }
        } else {
        for (unsigned int sb = 0; sb < n_list; sb += 64) {
          const unsigned int e = e_next;
          COUNT(0);
          if (sb + 64 < n_list) e_next = P.nbr_cells[first_cell + sb + 64 + lane];   // next batch in flight
          match_cell(e, true, phase);
        }
        }
        n_cells = 0u;
        // Every batch has been consumed and the next record is needed now: an explicit vmcnt(0), on every path.  Without
        // it the compiler cannot tell that no batch load is left outstanding and waits for ALL vector memory -- the
        // prefetch below included -- wherever a register of the cell loop is written again.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        // (the next record has arrived by now)
        if (have && __builtin_amdgcn_readlane((int)n_rec, 6) != 0)
          e_pref = P.nbr_cells[((uint64_t)(unsigned int)__builtin_amdgcn_readlane((int)n_rec, 4) | ((uint64_t)(unsigned int)__builtin_amdgcn_readlane((int)n_rec, 5) << 32)) + lane];
      }
      for (unsigned int sb = 0; sb < n_cells; sb += 64) {
        const unsigned int e = e_next;
        COUNT(0);
        if (sb + 64 + lane < n_cells) e_next = nbr_cells[first_cell + sb + 64 + lane];   // next batch in flight
        const float cdx = (float)(e & 0xffffu) * P.res_xf - xf;
        const float cdy = (float)(e >> 16) * P.res_yf - yf;
        const bool keep = (sb + lane < n_cells) && (cdx * cdx + cdy * cdy <= disk_r2);
        const unsigned long long km = __ballot(keep);
        if (keep) {
          const unsigned int pos = __builtin_amdgcn_mbcnt_hi((unsigned int)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)km, 0u));
          ring[(q_head + q_count + (int)pos) & 127] = e;
        }
        q_count += __popcll(km);
        STAMP(3);
        if (q_count >= 64) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          match_cell(ring[(q_head + lane) & 127], true, phase);
          STAMP(5);
          q_head = (q_head + 64) & 127;
          q_count -= 64;
        }
      }
      if (q_count > 0) {                                     // what is left in the ring
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        match_cell(lane < q_count ? ring[(q_head + lane) & 127] : 0u, lane < q_count, phase);
        STAMP(5);
        q_count = 0;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // a beam that also has a seam image keeps the smaller of the two minima in both entries
      // (entry G + k is beam k; entry g < G is the image of beam B - G + g)
      if (!use_iv) {
#pragma unroll 1
        for (int g = lane; g < G; g += 64) {
          const int real = B + g;   // == G + (B - G + g)
          if (!STAGE || phase == 0) {
            const unsigned long long a = rmin[g], b = rmin[real];
            const unsigned long long m = a < b ? a : b;
            rmin[g] = m;
            rmin[real] = m;
          } else {
            const unsigned long long a = aux[g], b = aux[real];
            aux[real] = a < b ? a : b;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if constexpr (IVK) {
      if (use_iv) {
        // The parked cells.  Sectors that hold edge E in their interior hold the cell; the one sector that ends at E
        // and the one that starts there are asked with the reference's predicate in float64
        // (mc_localization.py:148-149).  The interval below E serves the interior sectors and the ending one, the
        // interval above it the interior ones and the starting one: every outcome but one is a choice of these two
        // minima, and that one -- both refuse while interior sectors exist -- sends the particle through the
        // entry-by-entry test.
        // (most particles park nothing: a flag kept in a register spares them the read of the count and its wait)
        if constexpr (QK) {
          if (quirky) {
            // the interval that holds the seam, and the first beam whose sector can reach a cell parked there
            const int m_s = (int)(__builtin_amdgcn_fractf(-tho) * iv_nf);
            seam_k0 = ((m_s - seam_wi - UNIF - 1 + P.iv_NE) / P.iv_stride) % B;
            const double whole = rint(th * inv_two_pi);
            seam_thf = (float)fma(-whole, kTwoPi, th);
            seam_turns = (int)fmax(-1.0e6, fmin(1.0e6, whole));
            smin[lane0] = r2max_bits;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          }
        }
        const int nq = __ballot(lane_parked) != 0ull ? __builtin_amdgcn_readfirstlane((int)edge_cnt[0]) : 0;
        if constexpr (IVONLY) {
          if (nq > 0) {   // most particles park nothing: the count is cleared by the ones that did
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (lane == 0) edge_cnt[0] = 0u;
          }
        }
        if (nq > 0) {
          // (histogram filter: what a parked cell's lane works out -- squared distance, exact bearing, the classes at its
          // edge -- for the pass over the classes' entries below)
          unsigned long long g_r2bits = 0ull;
          double g_exact = 0.0;
          unsigned int g_ends = 0u;
          if (lane < nq && lane < EQ) {
            const unsigned long long q = edgeq[lane];
            const unsigned int e = (unsigned int)q;
            const int E = (int)(q >> 32);
            const int col = (int)(e & 0xffffu), row = (int)(e >> 16);
            const double dx = (double)((float)col * P.res_xf) - x, dy = (double)((float)row * P.res_yf) - y;   // as in match_cell
            const unsigned long long r2bits = (unsigned long long)__double_as_longlong(FAST ? fma(dx, dx, dy * dy) : dx * dx + dy * dy);
            const int left = E <= 0 ? IVN - 1 : E - 1, right = E < 0 ? 0 : E;
            const unsigned int ends = s_ivends[right];
            // a cell at the seam of the bearings (QK): every beam whose sector spans its interval or a neighbour of it
            // (the float32 interval may be one off) is asked with the reference's predicate -- sectors that start at
            // interval m + 1 down to m - UNIF.  Most of those beams hold the cell clearly inside or outside: the float32
            // bearing (within 1e-6 rad) decides them, and the float64 arctangent is taken only where a sector edge -- or
            // the seam itself, where the bearing jumps by a turn -- is within 1e-5 rad.
            const bool at_seam = QK && E < 0;
            unsigned int unsure = 0u;
            const int seam_m = E & 0x7fffffff, seam_st = QK ? P.iv_stride : 1;
            const int seam_kh = seam_st == 1 ? seam_m + 1 : (seam_m + 1) / seam_st;
            auto seam_give = [&](int kk) {
              int slot = kk - seam_k0;
              slot += slot < 0 ? B : 0;
              if (slot < kSeamSlots) atomicMin(&smin[slot], r2bits);
              else lane_redo = true;                           // (cannot happen: the host has checked the span)
            };
            if constexpr (QK) {
              if (at_seam) {
                const float at = bearing_turns_f32((float)row * P.res_yf - yf, (float)col * P.res_xf - xf);
                const double ar = (double)(at < 0.0f ? at + 1.0f : at) * kTwoPi;     // :133-135, approximately
                const bool seam_sure = fabsf(at) > 4e-6f;
                const float halff = (float)P.half, lof = (float)P.two_pi_lo, hif = (float)P.two_pi_hi;
                int j = 0;
                for (int k = seam_kh; k * seam_st >= seam_m - UNIF; --k, ++j) {
                  const int kk = k < 0 ? k + B : (k >= B ? k - B : k);
                  const float def = (float)fabs((s_ts[kk] + th) - ar);   // :143, :148 (where it matters, below 7: float32 holds it to 5e-7)
                  const bool clear = fabsf(def - halff) > 1e-5f && fabsf(def - lof) > 1e-5f && fabsf(def - hif) > 1e-5f;
                  if (!seam_sure || !clear) unsure |= 1u << j;
                  else if (def <= halff || (def >= lof && def <= hif)) seam_give(kk);   // :149
                }
              }
            }
            double a_exact = 0.0;
            if (!at_seam || unsure != 0u) a_exact = exact_bearing(dy, dx);            // :133-135
            if constexpr (QK) {
              if (at_seam && unsure != 0u) {
                int j = 0;
                for (int k = seam_kh; k * seam_st >= seam_m - UNIF; --k, ++j) {
                  if (!((unsure >> j) & 1u)) continue;
                  const int kk = k < 0 ? k + B : (k >= B ? k - B : k);
                  const double de = fabs((s_ts[kk] + th) - a_exact);                  // :143, :148
                  if ((de <= P.half) || (de >= P.two_pi_lo && de <= P.two_pi_hi)) seam_give(kk);   // :149
                }
              }
            }
            auto refuses = [&](unsigned int k) {
              const double beta = s_ts[k] + th;                      // :143
              if constexpr (QK) {
                // a beam two or more turns away sees nothing whatever is filed (see beam_r2bits): not a refusal that
                // would send the particle through the entry-by-entry test
                if (quirky && fabs(floor(beta * inv_two_pi)) > 1.0) return false;
              }
              const double de = fabs(beta - a_exact);                // :148
              return !((de <= P.half) || (de >= P.two_pi_lo && de <= P.two_pi_hi));   // :149
            };
            if constexpr (GRID) {
              // Histogram filter: the entries of a class are asked one by one, each with its own unreduced angle
              // (robot.py:252-258 -- where the last bits of an angle decide, entries of one class may differ), and what
              // an entry accepts goes to the entry's own minimum.  The class that ends at the edge, the one that
              // starts there, and the one that holds the edge in its interior (sectors span two intervals at most:
              // it is the class whose sector starts an interval earlier), which accepts without being asked.
              // Done below, a parked cell at a time with the lanes on the entries: sixteen lanes for each of the three
              // classes (a lane per parked cell walking its 39 entries one after the other cost three times as much).
              g_r2bits = r2bits; g_exact = a_exact; g_ends = ends;
              // (the class that holds the edge inside accepts without being asked -- every entry of it --: the cell goes
              // to a minimum of that class, kept where the class's normaliser will stand, instead of to its entries')
              if ((ends >> 22) & 1u) {
                const unsigned int prev = s_ivends[left];
                if ((prev >> 21) & 1u) atomicMin(emin + NE + ((prev >> 11) & 0x3ffu), r2bits);
              }
            } else if (!at_seam) {
            const bool no_end = ((ends >> 10) & 1u) && refuses(ends & 0x3ffu);
            const bool no_start = ((ends >> 21) & 1u) && refuses((ends >> 11) & 0x3ffu);
            if (!no_end) atomicMin(&imin[left], r2bits);
            else if (no_start && ((ends >> 22) & 1u)) lane_redo = true;
            if (!no_start) atomicMin(&imin[right], r2bits);
            }
          }
          if constexpr (GRID) {
            // sixteen lanes on the entries of the class that ends at a parked cell's edge, sixteen on the class that
            // starts there; each half of the wave has a parked cell of its own, and two such pairs go through a round in
            // straight-line code (each is a chain of four dependent LDS reads: two chains overlap)
            const int role = (lane >> 4) & 1, slot = lane & 15;
            const bool upper = lane >= 32;
            const int n_parked = nq < EQ ? nq : EQ;
            const long long eb = __double_as_longlong(g_exact);
            auto pick = [&](int v, int qa, int qb) -> unsigned int {      // what lane qa (lower half) / qb (upper half) holds
              const int a = __builtin_amdgcn_readlane(v, qa), b = __builtin_amdgcn_readlane(v, qb);
              return (unsigned int)(upper ? b : a);
            };
            auto cell_of = [&](int qa, int qb, unsigned long long& rb, double& ax, unsigned int& cc) -> bool {
              const unsigned int ends = pick((int)g_ends, qa, qb);
              rb = ((unsigned long long)pick((int)(g_r2bits >> 32), qa, qb) << 32) | (unsigned long long)pick((int)g_r2bits, qa, qb);
              ax = __longlong_as_double((long long)(((unsigned long long)pick((int)(eb >> 32), qa, qb) << 32) |
                                                    (unsigned long long)pick((int)eb, qa, qb)));
              cc = role == 0 ? (ends & 0x3ffu) : ((ends >> 11) & 0x3ffu);
              return (upper ? qb : qa) < n_parked && (role == 0 ? ((ends >> 10) & 1u) != 0u : ((ends >> 21) & 1u) != 0u);
            };
            auto accepts = [&](unsigned int me, double ax) -> bool {
              const unsigned int ck = s_cls[me];
              const double beta = s_ang[ck & 0x3ffu] + s_ang[P.g_B0 + ((ck >> 10) & 0x3ffu)];   // robot.py:252
              const double de = fabs((beta + th) - ax);
              return (de <= P.half) || (de >= P.two_pi_lo && de <= P.two_pi_hi);
            };
            const int q_last = n_parked - 1;
#pragma unroll 1
            for (int q = 0; q < n_parked; q += 4) {
              unsigned long long rb_a, rb_b;
              double ax_a, ax_b;
              unsigned int cc_a, cc_b;
              // (a lane index past the parked cells is read as the last one's and disabled)
              const bool en_a = cell_of(q, q + 1 < n_parked ? q + 1 : q_last, rb_a, ax_a, cc_a) && (!upper || q + 1 < n_parked);
              const bool en_b = cell_of(q + 2 < n_parked ? q + 2 : q_last, q + 3 < n_parked ? q + 3 : q_last, rb_b, ax_b, cc_b) &&
                                (upper ? q + 3 < n_parked : q + 2 < n_parked);
              const int mm_a = (int)s_first[cc_a] + slot, end_a = (int)s_first[cc_a + 1];
              const int mm_b = (int)s_first[cc_b] + slot, end_b = (int)s_first[cc_b + 1];
              const bool v_a = en_a && mm_a < end_a, v_b = en_b && mm_b < end_b;
              const unsigned int me_a = s_member[v_a ? mm_a : 0], me_b = s_member[v_b ? mm_b : 0];
              const bool ok_a = accepts(me_a, ax_a), ok_b = accepts(me_b, ax_b);
              if (v_a && ok_a) atomicMin(&emin[me_a], rb_a);
              if (v_b && ok_b) atomicMin(&emin[me_b], rb_b);
              // (classes of more than sixteen entries: the rest, one after the other)
              if (__ballot((en_a && mm_a + 16 < end_a) || (en_b && mm_b + 16 < end_b)) != 0ull) {
                if (en_a)
                  for (int mm = mm_a + 16; mm < end_a; mm += 16) {
                    const unsigned int me = s_member[mm];
                    if (accepts(me, ax_a)) atomicMin(&emin[me], rb_a);
                  }
                if (en_b)
                  for (int mm = mm_b + 16; mm < end_b; mm += 16) {
                    const unsigned int me = s_member[mm];
                    if (accepts(me, ax_b)) atomicMin(&emin[me], rb_b);
                  }
              }
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
      if (!IVONLY && use_iv && __ballot(lane_redo) != 0ull) {   // (rare) the same particle again, entry by entry; its pose is still at hand
        entry_by_entry = true;
        continue;
      }
    }

    // ranges -> beam mixture -> weight
    if constexpr (GRID) STAMP(1);   // (histogram filter: the parked cells, apart from the mixtures)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (use_iv) {
      for (int k = lane; k < P.iv_cmax; k += 64) imin[IVN + k] = imin[k];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if constexpr (!IVONLY) {
      have = next_particle();
      if (have) fetch_next();   // next pose in flight while this particle's weight is computed
    }
    if constexpr (IVONLY) {
      // not for this kernel: the heading is outside the interval path's range, or a cell on a sector edge asked for
      // the entry-by-entry test; the particle is filed for the launch that follows
      if (!use_iv || __ballot(lane_redo) != 0ull) {
        const int nf = __builtin_amdgcn_readfirstlane((int)filed_cnt[0]);
        if (lane == 0) { filed[nf] = (unsigned int)p; filed_cnt[0] = (unsigned int)(nf + 1); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (nf + 1 == kFiled) flush_filed(lane);
        if (++out_slot == kOutBatch) flush_out(lane, step);
        continue;
      }
    }
    // smallest squared distance in the sector of the beam at sorted position k: the per-beam table, and with sector
    // intervals also the intervals its sector spans (squared distances are not negative: their order as float64 is
    // their order as bit patterns)
    auto beam_r2bits = [&](int k) -> unsigned long long {
      unsigned long long rb = r2max_bits;
      if (!use_iv) rb = rmin[G + k];
      if constexpr (IVK) {
        if (use_iv) {
          // (UNIF > 0: launched only for tables whose sectors start at k * iv_stride and span UNIF intervals each)
          const unsigned int span_k = UNIF > 0 ? 0u : s_ivbeam[k];
          const unsigned long long* run = imin + (UNIF > 0 ? (unsigned int)(lane_run + (k - lane) * P.iv_stride) : (span_k & 0x3ffu));
          const int cnt = UNIF > 0 ? UNIF : (int)(span_k >> 10);
          double best = P.r2max;
          // (the plain instruction: nothing here is a NaN, so the library minimum's quieting of operands is not needed)
          auto lower = [](double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
          // five intervals at a time, their reads in flight together (a read-minimum chain per interval: 0.845 -> 0.826 ms
          // at 180 beams, 1.43 -> 1.29 ms at 360; a branch of its own for exactly five measured 0.807 alone and 0.877
          // beside this loop -- the register allocation of this kernel turns on such things), the rest one by one; `uniform`: every sector spans as many intervals (evenly spaced beams)
          const bool uniform = UNIF > 0 || P.iv_cmin == P.iv_cmax;
          const int cmax_ = UNIF > 0 ? UNIF : P.iv_cmax;
          int j = 0;
          for (; j + 5 <= cmax_; j += 5) {
            double v0 = __longlong_as_double((long long)run[j]), v1 = __longlong_as_double((long long)run[j + 1]),
                   v2 = __longlong_as_double((long long)run[j + 2]), v3 = __longlong_as_double((long long)run[j + 3]),
                   v4 = __longlong_as_double((long long)run[j + 4]);
            if (!uniform) {
              v0 = j < cnt ? v0 : best; v1 = j + 1 < cnt ? v1 : best; v2 = j + 2 < cnt ? v2 : best;
              v3 = j + 3 < cnt ? v3 : best; v4 = j + 4 < cnt ? v4 : best;
            }
            // (every minimum starts at z_max^2: the first group of a uniform run needs no cap of its own)
            const double g5 = lower(lower(lower(v0, v1), lower(v2, v3)), v4);
            best = (UNIF > 0 && j == 0) ? g5 : lower(best, g5);
          }
          for (; j < cmax_; ++j) {
            const double v = __longlong_as_double((long long)run[j]);
            best = lower(best, (uniform || j < cnt) ? v : best);
          }
          rb = (unsigned long long)__double_as_longlong(best);
          if constexpr (QK) {
            if (quirky) {
              // (float32: the heading less its whole turns, taken once per particle; within 2e-6 rad, which the 1e-5 of
              // margin covers -- a beam that is wrongly taken to touch the seam holds only cells that were parked)
              const float bf = (float)s_ts[k] + seam_thf;                           // :143, less seam_turns turns
              const float adj = floorf(bf * 0.15915494309189535f);
              const float d = fmaf(-adj, 6.2831853071795865f, bf);                  // where the beam points, in [0, 2 pi)
              const int turns = seam_turns + (int)adj;
              const bool keeps = turns >= -1 && turns <= 1 && d > seam_lof && d < seam_hif;
              int slot = k - seam_k0;
              slot += slot < 0 ? B : 0;
              const unsigned long long at_seam = slot < kSeamSlots ? smin[slot] : r2max_bits;
              rb = keeps ? rb : r2max_bits;
              rb = at_seam < rb ? at_seam : rb;
            }
          }
        }
      }
      return rb;
    };
    if (STAGE) {
      for (int k = lane; k < B; k += 64) {
        const double r2 = __longlong_as_double((long long)rmin[G + k]);
        const int ko = s_perm[k];
        ranges_out[p * B + ko] = sqrt(r2);  // :160
        const unsigned int h = (unsigned int)aux[G + k];
        hits_out[p * B + ko] = (h == 0xffffffffu) ? -1 : (int32_t)h;
      }
    } else if (GRID) {
      // Histogram filter (robot.py:289-300): the pose is a grid cell, the beam table holds every
      // (heading, beam) pair of that cell, and the measured range is the FIRST argument of
      // compute_px here (range_sensor.py:95-112 called from robot.py:297), the predicted one the second.
      // what depends on the predicted range alone -- the root and the normaliser of the short-reading
      // term (range_sensor.py:48) -- once per class; an entry that its own angle gave a nearer cell redoes it
      double* const etac = reinterpret_cast<double*>(emin + NE);
      // (with sector intervals a class's squared range is the smaller of its intervals', and its root goes where the
      // intervals stood: every class reads first -- the roots wait in the other array meanwhile)
      const int zoff = use_iv ? 0 : G;
      if (use_iv && B <= 128) {
        // up to two classes per lane, both in straight-line code (their chains of a root, an exponential and a division
        // overlap): the ranges first -- every class reads the interval minima before any class's range is written over
        // them --, then normalisers and stores
        const int ca = lane < B ? lane : B - 1, cb = lane + 64 < B ? lane + 64 : B - 1;
        // (the class's own minimum: parked cells that its sector holds inside, see above)
        const unsigned long long held_a = reinterpret_cast<const unsigned long long*>(etac)[ca], spans_a = beam_r2bits(ca);
        const unsigned long long held_b = reinterpret_cast<const unsigned long long*>(etac)[cb], spans_b = beam_r2bits(cb);
        const double za = sqrt(__longlong_as_double((long long)(held_a < spans_a ? held_a : spans_a)));   // robot.py:269
        const double zb = sqrt(__longlong_as_double((long long)(held_b < spans_b ? held_b : spans_b)));
        const double na = 1.0 / (1.0 - exp_nonpos<true>(-P.lambda * za)), nb = 1.0 / (1.0 - exp_nonpos<true>(-P.lambda * zb));
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane < B) { etac[ca] = na; rmin[zoff + ca] = (unsigned long long)__double_as_longlong(za); }
        if (lane + 64 < B) { etac[cb] = nb; rmin[zoff + cb] = (unsigned long long)__double_as_longlong(zb); }
      } else {
      if (use_iv) {
        for (int cc = lane; cc < B; cc += 64) {
          const unsigned long long held = reinterpret_cast<const unsigned long long*>(etac)[cc], spans = beam_r2bits(cc);
          etac[cc] = sqrt(__longlong_as_double((long long)(held < spans ? held : spans)));   // robot.py:269
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      }
      for (int cc = lane; cc < B; cc += 64) {
        const double z = use_iv ? etac[cc] : sqrt(__longlong_as_double((long long)rmin[G + cc]));   // robot.py:269
        etac[cc] = 1.0 / (1.0 - exp_nonpos<true>(-P.lambda * z));
        rmin[zoff + cc] = (unsigned long long)__double_as_longlong(z);
      }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      // p(z_meas | z_pred) of one entry (range_sensor.py:95-112 called from robot.py:297)
      // (without branches, so that two entries' chains can be interleaved: the exponential is taken whatever the range
      // test says -- exp_nonpos and div_by give the bits of exp and of the division, with fewer instructions)
      const DivBy by_sigma2 = make_div_by(P.sigma2);
      // (the mixtures once for a divisor of ordinary size and once for any other: a test inside the loop, uniform as it
      // is, would cut the loop's body into blocks and keep the two entries' chains apart)
      auto likelihoods = [&](auto fast_div_tag) {
        constexpr bool kFastDiv = decltype(fast_div_tag)::value;
        auto entry_px = [&](int kb, double z, double eta) -> double {
          const double xm = s_zt[kb];
          const double el = s_el[kb];                              // exp(-lambda * z_meas) tabulated per beam
          const bool in_range = (0.0 <= xm) && (xm <= P.z_max);
          const double d = xm - z;
          const double num = -0.5 * (d * d);
          double e_hit = exp_nonpos<true>(kFastDiv ? div_by(num, by_sigma2) : num / P.sigma2);
          asm volatile("" : "+v"(e_hit));   // (taken here: not moved under a branch on the range test, where the second entry's chain could not join it)
          const double p_hit = in_range ? P.c_hit * e_hit : 0.0;
          const double p_short = ((0.0 <= xm) && (xm <= z)) ? eta * P.lambda * el : 0.0;
          return p_hit * P.w_hit + p_short * P.w_short + s_tail[kb] + s_tail[P.g_B0 + kb];   // (+ p_max * w_max + p_rand * w_rand)
        };
        // entries whose own minimum may undercut their class wait in the ring and are done together, so the
        // common case stays free of the root / exponential / division they need
        auto own_range = [&](unsigned int k) {
          const unsigned int ck = s_cls[k];
          const int cc = (int)(ck >> 20);
          double z = __longlong_as_double((long long)rmin[zoff + cc]);
          double eta = etac[cc];
          const double zo = sqrt(__longlong_as_double((long long)emin[k]));
          if (zo < z) { z = zo; eta = 1.0 / (1.0 - exp_nonpos<true>(-P.lambda * z)); }
          emin[k] = (unsigned long long)__double_as_longlong(entry_px((int)(ck & 0x3ffu), z, eta));
        };
        STAMP(4);   // (histogram filter: class ranges and normalisers)
        int waiting = 0;
        // several entries per lane and round (MCL_GRID_WAYS), in straight-line code (an entry beyond the table computes the
        // last one's value and drops it, one that waits for its own range computes its class's and drops it): their
        // chains of float64 arithmetic -- an exponential and a division each -- overlap, which four waves per SIMD
        // alone do not hide
        constexpr int kWays = MCL_GRID_WAYS;
        for (int k0 = 0; k0 < NE; k0 += 64 * kWays) {
          int kk[kWays];
          unsigned int ck[kWays];
          double zz[kWays], ee[kWays], px[kWays];
          bool own[kWays];
  #pragma unroll
          for (int u = 0; u < kWays; ++u) {
            kk[u] = k0 + 64 * u + lane;
            const int j = kk[u] < NE ? kk[u] : NE - 1;
            ck[u] = s_cls[j];
            const int cc = (int)(ck[u] >> 20);
            zz[u] = __longlong_as_double((long long)rmin[zoff + cc]);
            ee[u] = etac[cc];
            const unsigned long long me = emin[j];
            own[u] = kk[u] < NE && (me != r2max_bits) && (__longlong_as_double((long long)me) <= zz[u] * zz[u] * (1.0 + 1e-15));
          }
  #pragma unroll
          for (int u = 0; u < kWays; ++u) px[u] = entry_px((int)(ck[u] & 0x3ffu), zz[u], ee[u]);
  #pragma unroll
          for (int u = 0; u < kWays; ++u)
            if (kk[u] < NE && !own[u]) emin[kk[u]] = (unsigned long long)__double_as_longlong(px[u]);   // in place: the minimum is spent
          bool any = false;
  #pragma unroll
          for (int u = 0; u < kWays; ++u) any = any || own[u];
          if (__ballot(any) != 0ull) {
  #pragma unroll
            for (int u = 0; u < kWays; ++u) {
              const unsigned long long om = __ballot(own[u]);
              if (own[u]) {
                const unsigned int pos = __builtin_amdgcn_mbcnt_hi((unsigned int)(om >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)om, 0u));
                ring[waiting + (int)pos] = (unsigned int)kk[u];
              }
              waiting += __popcll(om);
              if (waiting >= 64) {          // (the ring holds 128)
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                own_range(ring[waiting - 64 + lane]);
                waiting -= 64;
              }
            }
          }
        }
        if (waiting > 0) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (lane < waiting) own_range(ring[lane]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int64_t cell = P.g_cell[p];
        for (int d = lane; d < P.g_D; d += 64) {
          double q = 1.0;   // q *= p_k in beam order (robot.py:295-298)
          for (int k = 0; k < P.g_B0; ++k) q = q * __longlong_as_double((long long)emin[d * P.g_B0 + k]);
          P.g_pk[cell * P.g_D + d] = (float)(P.g_pk_pred[cell * P.g_D + d] * q);   // robot.py:300, float32 store
        }
      };
      if (by_sigma2.usable) likelihoods(std::true_type{}); else likelihoods(std::false_type{});
    } else if constexpr (FAST) {
      // range_sensor.py:31-112 per beam, float32 (see the comment at the template head).  What is exact:
      //   0 <= z <= z_max holds for every entry (r2 <= z_max^2 by construction), so p_hit and p_rand always apply;
      //   z <= z*  <=>  r2 <= (the largest float64 whose root rounds to z* or below: the staged "z*^2");
      //   |z - z_max| < 1e-6  <=>  r2 > (z_max - 1e-6)^2 (float64 bit patterns).
      const float k_hit = (float)neg_half_inv_s2_log2e, k_short = (float)neg_lambda_log2e;
      const double chw = P.c_hit * P.w_hit;
      const float prw = (float)(P.p_rand * P.w_rand), wmax = (float)P.w_max;
      const double zm1 = P.z_max - 1e-6;
      const unsigned long long near_max_bits = (unsigned long long)__double_as_longlong(zm1 > 0.0 ? zm1 * zm1 : 0.0);
      double l2s = 0.0;          // sum of log2 of the lane's partial products
      double prod = 1.0;
      int nfac = 0;
      // log2 of a partial product: exponent and mantissa split in integer arithmetic, v_log_f32 of the mantissa
      // (in [1, 2): absolute error ~1e-7), joined in float64
      auto log2_of_product = [&](double v) -> double {
        const long long bits = __double_as_longlong(v);
        const int ex = (int)((bits >> 52) & 0x7ff);
        if (ex == 0 || ex == 0x7ff || bits < 0) return slow_log(v) * 1.4426950408889634;   // zero, subnormal, infinite, NaN
        const double mant = __longlong_as_double((bits & 0x000fffffffffffffll) | 0x3ff0000000000000ll);
        return (double)(ex - 1023) + (double)__builtin_amdgcn_logf((float)mant);
      };
      auto beam_factor = [&](int k) {
        const unsigned long long rb = beam_r2bits(k);
        const double r2 = __longlong_as_double((long long)rb);
        if constexpr (RNOISE) {
          // augmented MCL with device-drawn range noise (mc_localization2.py:166-173): r + sigma n on the predicted
          // range -- min(z_max, z_max + sigma n) on a map without obstacles --, n from one counter block per
          // (particle, beam in sorted order).  The noisy range can leave [0, z_max], so every indicator of
          // range_sensor.py:31-67 is taken on it; the cancellation-free difference of the noiseless path has nothing to
          // offer here (the noise is of the size of sigma_hit), the terms are plain float32.
          const uint64_t gp = (uint64_t)(P.rn_index0 + p);
          uint32_t ctr[4] = {(uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)k, 0x524fu};
          philox4x32(ctr, (uint32_t)P.rn_seed, (uint32_t)(P.rn_seed >> 32));
          double n0, n1;
          box_muller_f32(ctr[0], ctr[1], &n0, &n1);
          const float term = (float)n0 * (float)P.rn_sigma;
          const float zmaxf = (float)P.z_max;
          const float2 zf = reinterpret_cast<const float2*>(s_el)[k];   // z*, eta * lambda * w_short
          const float zn = P.no_obstacles ? fminf(zmaxf, zmaxf + term) : __builtin_amdgcn_sqrtf((float)r2) + term;
          const bool in_range = (0.0f <= zn) && (zn <= zmaxf);
          const float d = zn - zf.x;
          const float e_hit = __builtin_amdgcn_exp2f((d * d) * k_hit);
          const float e_short = __builtin_amdgcn_exp2f(zn * k_short);
          float small = in_range ? prw : 0.0f;
          small = small + ((0.0f <= zn && zn <= zf.x) ? e_short * zf.y : 0.0f);
          small = small + (fabsf(zn - zmaxf) < 1e-6f ? wmax : 0.0f);
          return (in_range ? (double)e_hit * chw : 0.0) + (double)small;
        }
        const float num = (float)(r2 - s_zt[k]);                      // z^2 - z*^2, cancellation exact in float64
        const float2 zf = reinterpret_cast<const float2*>(s_el)[k];   // z*, eta * lambda * w_short
        const float z0 = __builtin_amdgcn_sqrtf((float)r2);
        const float den = (z0 + 1e-30f) + zf.x;                       // > 0 (both ranges 0: d = 0 * 1e30 = 0)
        const float d = num * __builtin_amdgcn_rcpf(den);             // z - z*
        const float e_hit = __builtin_amdgcn_exp2f((d * d) * k_hit);
        const float e_short = __builtin_amdgcn_exp2f(z0 * k_short);
        // the small terms in float32; the dominant one and the sum in float64: beams in the same state (typically
        // many at z = z* = z_max) round alike, and a float32 sum would add that rounding up coherently
        float small = prw;                                            // p_rand * w_rand
        small = small + (num <= 0.0f ? e_short * zf.y : 0.0f);        // p_short * w_short where z <= z*
        small = small + (rb > near_max_bits ? wmax : 0.0f);           // p_max * w_max
        return (double)e_hit * chw + (double)small;                   // + p_hit * w_hit
      };
      if constexpr (STATIC_L && CAP / 64 <= 16) {
        // the fixed layout's beam count is at most CAP: a lane has CAP / 64 beams at most, one after the other without
        // a loop (its counter, its bound and the partial product's count are gone: 12 instructions of 45 per round)
#pragma unroll
        for (int it = 0; it < CAP / 64; ++it) {
          const int k = lane + 64 * it;
          if (k < B) prod *= beam_factor(k);
        }
        l2s = log2_of_product(prod);
      } else {
        for (int k = lane; k < B; k += 64) {
          prod *= beam_factor(k);
          if (++nfac == 16) {                                         // sixteen factors >= 1e-3 cannot underflow
            l2s += log2_of_product(prod);
            prod = 1.0;
            nfac = 0;
          }
        }
        if (nfac) l2s += log2_of_product(prod);
      }
      const double lsd = wave_sum_fixed(l2s) * 0.6931471805599453;
      if constexpr (IVONLY) {
        if (lane == 0) { outbuf[out_slot] = on_free ? lsd : -INFINITY; if constexpr (LISTED) outidx[out_slot] = (unsigned int)p; }
        out_mask |= 1u << out_slot;
        if (++out_slot == kOutBatch) flush_out(lane, step);
      } else if (lane == 0) logw_out[p] = on_free ? lsd : -INFINITY;
    } else {
      typedef T R;  // likelihood arithmetic follows the state dtype
      // log-weight: each lane multiplies its few p_k in float64 and takes one logarithm of the
      // partial product (sum of logs == log of product; 16 factors >= 1e-3 cannot underflow)
      double ls = 0.0, prod = 1.0;
      int nfac = 0;
      // float64 state: the library logarithm.  float32 state: exponent and mantissa are split in
      // integer arithmetic, the mantissa (in [1, 2)) goes through v_log_f32 and the two parts are
      // joined in float64 -- absolute error ~1e-7 per call whatever the size of the product
      auto log_of_product = [&](double v) -> double {
        if constexpr (sizeof(R) == 4) {
          const long long bits = __double_as_longlong(v);
          const int ex = (int)((bits >> 52) & 0x7ff);
          if (ex == 0 || ex == 0x7ff || bits < 0) return slow_log(v);   // zero, subnormal, infinite, NaN, negative
          const double mant = __longlong_as_double((bits & 0x000fffffffffffffll) | 0x3ff0000000000000ll);
          return ((double)(ex - 1023) + (double)__builtin_amdgcn_logf((float)mant)) * 0.6931471805599453;
        } else {
          return log(v);
        }
      };
      for (int k = lane; k < B; k += 64) {
        const double r2 = __longlong_as_double((long long)beam_r2bits(k));
        double zd;
        // (the correctly rounded root in float32 state too: the indicators below compare it with the measured range,
        // which for a particle that stands where the measurement was taken is this very root, bit for bit)
        zd = sqrt(r2);
        if constexpr (RNOISE) {
          // one counter block per (particle, beam in caller order); mc_localization2.py:166-173 as below
          const uint64_t gp = (uint64_t)(P.rn_index0 + p);
          uint32_t ctr[4] = {(uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)s_perm[k], 0x524eu};
          philox4x32(ctr, (uint32_t)P.rn_seed, (uint32_t)(P.rn_seed >> 32));
          double n0, n1;
          box_muller_f32(ctr[0], ctr[1], &n0, &n1);
          const double term = n0 * P.rn_sigma;
          zd = P.no_obstacles ? fmin(P.z_max, P.z_max + term) : zd + term;
        } else if (P.range_noise) {
          // augmented MCL: r += randn()*0.15 on every predicted range (mc_localization2.py:166-168);
          // without any obstacle the reading is min(z_max, z_max + noise) (:170-173)
          const double term = P.range_noise[p * B + s_perm[k]];
          zd = P.no_obstacles ? fmin(P.z_max, P.z_max + term) : zd + term;
        }
        const double ztd = s_zt[k];
        const bool in_range = (0.0 <= zd) && (zd <= P.z_max);       // range_sensor.py:35, :63
        const bool in_short = (0.0 <= zd) && (zd <= ztd);           // range_sensor.py:47
        double pkd;
        if constexpr (sizeof(R) == 4) {
          // float32 mode: the two exponentials in float32, exponents and the mixture itself in
          // float64 (float32 constants would bias every beam's log the same way)
          const double d = zd - ztd;
          // v_exp_f32 on the exponent scaled to base 2 (about 1e-7 relative; underflow to zero is what
          // the mixture's constant floor absorbs)
          const float e_hit = __builtin_amdgcn_exp2f((float)((d * d) * neg_half_inv_s2_log2e));
          const float e_short = __builtin_amdgcn_exp2f((float)(neg_lambda_log2e * zd));
          pkd = (in_range ? (double)e_hit * (P.c_hit * P.w_hit) : 0.0) +
                (in_short ? (double)e_short * (s_el[k] * P.w_short) : 0.0) +
                ((fabs(zd - P.z_max) < 1e-6) ? P.w_max : 0.0) + (in_range ? P.p_rand * P.w_rand : 0.0);
        } else {
          // range_sensor.py:31-67, operation by operation
          const double d = zd - ztd;
          // (exp_nonpos: the bits of exp for the arguments that count here, none of which is positive)
          const double p_hit = in_range ? P.c_hit * exp_nonpos(-0.5 * (d * d) / P.sigma2) : 0.0;
          const double p_short = in_short ? s_el[k] * exp_nonpos(-P.lambda * zd) : 0.0;
          const double p_max = (fabs(zd - P.z_max) < 1e-6) ? 1.0 : 0.0;
          const double p_rand = in_range ? P.p_rand : 0.0;
          pkd = p_hit * P.w_hit + p_short * P.w_short + p_max * P.w_max + p_rand * P.w_rand;
        }
        prod *= pkd;
        if (++nfac == 16) {
          ls += log_of_product(prod);
          prod = 1.0;
          nfac = 0;
        }
        if (PRODUCT) aux[s_perm[k]] = (unsigned long long)__double_as_longlong(pkd);
      }
      if (nfac) ls += log_of_product(prod);
      const double lsd = wave_sum(ls);
      double q = 1.0;
      if (PRODUCT) {
        // q *= p_k strictly in beam order (mc_localization.py:235-238): a running product that
        // underflows on the way stays zero, as it does in the reference
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int k = 0; k < B; ++k) q = q * __longlong_as_double((long long)aux[k]);
      }
      if (lane == 0) {
        const bool fr = on_free;
        if (PRODUCT) w_out[p] = fr ? q : q * 0.0;  // q * pm (mc_localization.py:240)
        logw_out[p] = fr ? lsd : -INFINITY;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    STAMP(6);
  }
  if constexpr (IVONLY) { flush_filed(lane0); flush_out(lane0, step); }
#ifdef MCL_STAMPS
  if (lane0 == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&P.stamps[i], acc[i]);
#endif
#ifdef MCL_COUNTS
  for (int i = 0; i < 8; ++i)
    if (cnt[i]) atomicAdd(&P.stamps[8 + i], cnt[i]);
#endif
}

static size_t ray_lds_bytes(int B, int G, int NB, bool aux, int waves = kWavesPerBlock, int NE = 0, int B0 = 0, int iv_NE = 0,
                            int iv_NB = 0, int iv_cmax = 0, bool ivonly = false, int cap = 0) {
  if (ivonly && cap > 0) {   // the interval-only kernel's fixed layout
    B = cap; iv_NE = cap; iv_cmax = kCapSpan; iv_NB = cap_buckets(cap);
  }
  const int NXP = ivonly ? 0 : B + G + kGroup;
  const int NT = NE ? B0 : B;
  const int NXQ = NE ? std::max(NXP, iv_NE + iv_cmax) : NXP;                   // (histogram filter: class minima or interval minima, then per entry)
  const int RS = NXQ + NE + (NE ? B : 0);
  const int TS = std::max(RS, iv_NE ? iv_NE + iv_cmax : 0) + (ivonly && cap > 0 ? kSeamSlots : 0);   // per-wave table: per entry or per interval (+ the beams at the seam)
  size_t s = (size_t)(B + 2 * NT) * sizeof(double);
  s += (size_t)waves * (TS + (aux ? RS : 0)) * sizeof(unsigned long long);
  s += (size_t)waves * kWaveScratch * sizeof(unsigned int);
  s += (size_t)NXP * sizeof(float);
  if (!ivonly) s += ((size_t)B + NB + 4) * sizeof(unsigned short);
  if (iv_NE)                                                                   // sector-interval tables
    s += (size_t)(iv_NE + 1) * sizeof(float) + ((size_t)B + iv_NB + 2) * sizeof(unsigned short) + (size_t)iv_NE * sizeof(unsigned int) + 8;
  if (NE) {
    s += (size_t)NE * sizeof(unsigned int) + 8;                            // class / heading / beam of every entry (+ alignment of what follows)
    s += (size_t)(3 * B0 + NE / B0) * sizeof(double);                      // beam and heading angles, the mixture's per-beam terms
    s += ((size_t)NE + B + 2) * sizeof(unsigned short);                    // entries by class, class starts
  }
  return (s + 15) & ~(size_t)15;
}

// Measurement update of the histogram filter: one wave per free cell, beam table = all (heading,
// beam) pairs.  d_x/d_y/d_th: float64 poses of the free cells (th = 0), d_cell their flat index.
int launch_grid_update(mclhip_ctx* c, const double* d_x, const double* d_y, const double* d_th, const int32_t* d_cell,
                       int64_t n, int n_dir, int beams_per_dir, const double* pk_pred, float* pk) {
  if (n <= 0) return MCLHIP_OK;
  RayParams P = make_ray_params(c);
  P.g_cell = d_cell; P.g_D = n_dir; P.g_B0 = beams_per_dir; P.g_pk_pred = pk_pred; P.g_pk = pk;
  P.g_NE = c->gl_NE; P.g_cls = c->gl_cls; P.g_cls_first = c->gl_cls_first; P.g_member = c->gl_member; P.g_ar = c->gl_mbeta; P.g_thd = c->gl_mbeta + beams_per_dir;
  P.zt = c->gl_zt; P.etalam = c->gl_el;   // per beam of one heading
  // Sector intervals of the class table, where every sector spans at most two of them (the cell on a sector edge is
  // then decided against the one class that ends there, the one that starts there and the one that holds the edge
  // in its interior -- see the kernel); other tables run window by window.
  if (!MCL_FAST_INTERVALS || P.iv_NE <= 0 || P.iv_cmax > 2) { P.iv_NE = 0; P.iv_cmax = 0; P.iv_NB = 0; }
  // as many waves per block as the LDS holds: 16, 12, 8 or 4
  const size_t lds_max = 160 * 1024;
  int waves = 16;
  while (waves > 4 && ray_lds_bytes(P.B, P.G, P.NB, false, waves, P.g_NE, P.g_B0, P.iv_NE, P.iv_NB, P.iv_cmax) > lds_max) waves -= 4;
  const size_t lds = ray_lds_bytes(P.B, P.G, P.NB, false, waves, P.g_NE, P.g_B0, P.iv_NE, P.iv_NB, P.iv_cmax);
  if (lds > lds_max) return fail(c, MCLHIP_E_LIMIT, "heading x beam table does not fit the kernel's LDS budget");
  int64_t blocks = (n + waves - 1) / waves;
  const int64_t max_blocks = 256 * 8 * 4 * 4 / waves;
  if (blocks > max_blocks) blocks = max_blocks;
#define MCL_GRID(W) do { if (P.iv_NE > 0 && c->iv_uniform) MCL_GRID_U(W, -1); else MCL_GRID_U(W, 0); } while (0)
#define MCL_GRID_U(W, U)                                                                                              \
  do {                                                                                                                \
    auto kern = raycast_kernel<double, false, false, true, W, false, false, false, false, 0, U>;                      \
    if (lds > 48 * 1024)                                                                                              \
      MCL_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                     (int)lds));                                                                      \
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * W), lds, c->stream, P, d_x, d_y, d_th,                 \
                       (const uint8_t*)nullptr, n, (double*)nullptr, (double*)nullptr, (double*)nullptr,              \
                       (int32_t*)nullptr, (const QueryRecord*)nullptr);                                               \
  } while (0)
  if (waves == 16) MCL_GRID(16);
  else if (waves == 12) MCL_GRID(12);
  else if (waves == 8) MCL_GRID(8);
  else MCL_GRID(4);
#undef MCL_GRID
#undef MCL_GRID_U
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// the instantiation's name as the profiler prints it, blanks left out
static std::string ray_kernel_name(bool f32, bool stage, bool product, bool grid, int wpb, bool rnoise, bool fast, bool ivonly,
                                   bool listed, int cap, int unif) {
  auto b = [](bool v) { return v ? "true" : "false"; };
  return std::string("raycast_kernel<") + (f32 ? "float" : "double") + "," + b(stage) + "," + b(product) + "," + b(grid) + "," +
         std::to_string(wpb) + "," + b(rnoise) + "," + b(fast) + "," + b(ivonly) + "," + b(listed) + "," + std::to_string(cap) + "," +
         std::to_string(unif) + ">";
}

int ray_arith_span(const mclhip_ctx* c) {
  if (!MCL_FAST_INTERVALS || c->iv_NE <= 0 || c->iv_cmax > kCapSpan) return 0;
  if (!(c->iv_uniform && c->iv_cmin == c->iv_cmax && c->iv_stride > 0)) return 0;
  auto fits = [&](int cc) { return c->B <= cc && c->iv_NE <= cc && c->iv_NB <= cap_buckets(cc); };
  if (fits(192)) return c->iv_cmax == 5 ? 5 : 0;
  if (fits(384) && 2 * kWavesPerBlock <= 16) return c->iv_cmax == 10 ? 10 : 0;
  return 0;
}

bool ray_serves_every_heading(const mclhip_ctx* c) {
  const int span = ray_arith_span(c);
  if (span == 0 || !c->every_heading) return false;
  // the beams that can receive a cell parked at the seam of the bearings are kSeamSlots at most (the kernel's seam_wi,
  // seam_k0 and slots)
  const int wi = (int)std::ceil((2.0 * c->half + 1e-4) / (2.0 * M_PI) * (double)c->iv_NE) + 2;
  return (2 * wi + span + 2) / c->iv_stride + 2 <= kSeamSlots && c->iv_NE > wi + span + 2;
}

int launch_raycast(mclhip_ctx* c, const void* x, const void* y, const void* th, const uint8_t* pm, int64_t n,
                   int dtype, double* w_out, double* logw_out, double* ranges_out, int32_t* hits_out,
                   const double* range_noise, double rn_sigma, uint64_t rn_seed, const QueryRecord* qrec) {
  const bool product = w_out != nullptr;
  // (whether the motion launch before this one has cleared the left-over counts: said once, for the launch that follows it)
  const bool counts_cleared = c->redo_cleared;
  c->redo_cleared = false;
  if (n <= 0) return MCLHIP_OK;
  RayParams P = make_ray_params(c);
  P.range_noise = range_noise;
  P.rn_sigma = rn_sigma; P.rn_seed = rn_seed;
  const bool device_noise = rn_sigma > 0.0;
  const bool stage = ranges_out != nullptr;
  // log-weights only, float32 state, no range noise, measured ranges finite and >= 0: the float32 mixture
  // (device-drawn range noise as well, where the query runs as the two launches over a fixed layout: decided below)
  bool fast = !stage && dtype == MCLHIP_F32 && !product && !range_noise && c->zmeas_ok && MCL_FAST_LIKELIHOOD;
  if (stage || !MCL_FAST_INTERVALS) { P.iv_NE = 0; P.iv_cmax = 0; }
  // The float32-mixture path with sector intervals runs as two launches: the interval-only kernel, which files the
  // particles it does not serve (headings outside the interval path's range: 5 % of a uniform cloud), and the general
  // kernel over that list.  Together 0.9x ms instead of 1.05 ms at 2^20 x 180.
  // (the interval-only kernel takes its particles from the records the motion kernel left)
  bool split = fast && P.iv_NE > 0 && c->redo_list && n <= c->redo_cap && pm != nullptr && qrec != nullptr && c->split_mode != 0;
  // last time's left-over counts (a hint, see mclhip_ctx::redo_hint): of the first launch, and of the second pass
  uint64_t left_a = 0, left_b = 0;
  if (c->redo_hint)
    for (int k = 0; k < kSubLists; ++k) {
      left_a += reinterpret_cast<volatile uint32_t*>(c->redo_hint)[k];
      left_b += reinterpret_cast<volatile uint32_t*>(c->redo_hint)[kSubLists + k];
    }
  const bool every = split && !device_noise && ray_serves_every_heading(c) && c->redo_list2 != nullptr &&
                     (c->every_heading == 1 || (c->every_heading < 0 && left_a * 16 > (uint64_t)n && left_a * 2 < (uint64_t)n));
  if (split && c->split_mode < 0 && c->redo_hint) {
    const uint64_t left = every ? left_b : left_a;
    if (left * 4 > (uint64_t)n && (++c->split_skipped % 8) != 0) split = false;
  }
  // Wavefronts per block.  The block-wide tables grow with the beam count; when they keep a compute unit from holding
  // the waves the instantiation is compiled for, twice the waves behind one copy of the tables hold more (360 beams
  // with sector intervals: 16 waves per unit in blocks of four, 24 in blocks of eight).
  int wpb = kWavesPerBlock;
  // the interval-only kernel's fixed LDS layouts: 192 beams / intervals in blocks of four waves, 384 in blocks of eight
  int cap = 0;
  if (split && P.iv_cmax <= kCapSpan) {
    auto fits = [&](int cc) { return P.B <= cc && P.iv_NE <= cc && P.iv_NB <= cap_buckets(cc); };
    if (fits(192)) cap = 192;
    else if (fits(384) && 2 * kWavesPerBlock <= 16) { cap = 384; wpb = 2 * kWavesPerBlock; }
  }
  if (fast && device_noise && !(split && cap != 0)) {   // no fixed layout for this table: the general kernel draws the noise
    fast = false; split = false; cap = 0; wpb = kWavesPerBlock;
  }
  auto lds_of = [&](int w) { return ray_lds_bytes(P.B, P.G, P.NB, stage || product, w, 0, 0, P.iv_NE, P.iv_NB, P.iv_cmax, split, cap); };
  if (fast && cap == 0 && 2 * kWavesPerBlock <= 16) {
    const int want = 4 * (split ? MCL_IVONLY_WAVES : MCL_FAST_WAVES);
    auto waves_per_unit = [&](int w) { const size_t per = lds_of(w); return per > 160 * 1024 ? 0 : std::min<int>(want, (int)(160 * 1024 / per) * w); };
    if (waves_per_unit(2 * kWavesPerBlock) > waves_per_unit(kWavesPerBlock)) wpb = 2 * kWavesPerBlock;
  }
  const size_t lds = lds_of(wpb);
  if (lds > 160 * 1024) return fail(c, MCLHIP_E_LIMIT, "beam table does not fit the kernel's LDS budget");
  int64_t blocks = (n + wpb - 1) / wpb;
  const int64_t max_blocks = (int64_t)MCL_MAX_BLOCKS * kWavesPerBlock / wpb;
  if (blocks > max_blocks) blocks = max_blocks;
  dim3 grid((unsigned)blocks), block(64 * wpb);
  // large beam tables need more dynamic LDS than a kernel gets by default
#define MCL_RAY(TT, SS, PP) MCL_RAY_N(TT, SS, PP, false)
#define MCL_RAY_N(TT, SS, PP, NN) MCL_RAY_F(TT, SS, PP, NN, false)
#define MCL_RAY_F(TT, SS, PP, NN, FF) MCL_RAY_W(TT, SS, PP, NN, FF, kWavesPerBlock, false, false, 0)
#define MCL_RAY_W(TT, SS, PP, NN, FF, WW, II, LL, CC) MCL_RAY_U(TT, SS, PP, NN, FF, WW, II, LL, CC, 0)
#define MCL_RAY_U(TT, SS, PP, NN, FF, WW, II, LL, CC, UU)                                                              \
  do {                                                                                                               \
    auto kern = raycast_kernel<TT, SS, PP, false, WW, NN, FF, II, LL, CC, UU>;                                         \
    if (!(LL)) c->query_kernel = ray_kernel_name(sizeof(TT) == 4, SS, PP, false, WW, NN, FF, II, LL, CC, UU);        \
    if (lds > 48 * 1024)                                                                                             \
      MCL_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                     (int)lds));                                                                     \
    hipLaunchKernelGGL(kern, grid, block, lds, c->stream, P, (const TT*)x, (const TT*)y, (const TT*)th, pm, n, w_out, \
                       logw_out, ranges_out, hits_out, II ? qrec : (const QueryRecord*)nullptr);                     \
  } while (0)
  if (!stage && !split) timing_begin(c, 0);
  if (stage) {
    if (dtype != MCLHIP_F64) return fail(c, MCLHIP_E_ARG, "stage query takes float64 poses");
    MCL_RAY(double, true, false);
  } else if (device_noise && !fast) {
    if (dtype == MCLHIP_F32) {
      if (product) MCL_RAY_N(float, false, true, true); else MCL_RAY_N(float, false, false, true);
    } else {
      if (product) MCL_RAY_N(double, false, true, true); else MCL_RAY_N(double, false, false, true);
    }
  } else if (dtype == MCLHIP_F32) {
    if (product) MCL_RAY(float, false, true);
    else if (split) {
      // blocks b, b + 64, ... of the first launch append to list b % 64: what one list can receive is bounded by the
      // particles those blocks serve
      P.sub_cap = ((n + wpb - 1) / wpb + kSubLists - 1) / kSubLists * wpb;
      if ((int64_t)P.sub_cap * kSubLists > c->redo_cap + mclhip_ctx::kRedoSlack) return fail(c, MCLHIP_E_LIMIT, "left-over list too small");
      const size_t lds_b = ray_lds_bytes(P.B, P.G, P.NB, false, kWavesPerBlock, 0, 0, 0, 0, 0);   // the second launch's
      if (lds_b > 160 * 1024) return fail(c, MCLHIP_E_LIMIT, "beam table does not fit the kernel's LDS budget");
      if (!counts_cleared) MCL_HIP(c, hipMemsetAsync(c->redo_count, 0, 2 * kSubLists * sizeof(uint32_t), c->stream));   // (both lists' counts)
      P.redo_list = c->redo_list; P.redo_count = c->redo_count;
      timing_begin(c, 0);
      const bool even = c->iv_uniform;                                       // equally wide intervals
      const int arith = ray_arith_span(c);                                   // ... as many per sector, sectors start at k * stride: the UNIF > 0 variants
      if (device_noise && cap == 192 && even) MCL_RAY_U(float, false, false, true, true, kWavesPerBlock, true, false, 192, -1);
      else if (device_noise && cap == 192) MCL_RAY_U(float, false, false, true, true, kWavesPerBlock, true, false, 192, 0);
      else if (device_noise && even) MCL_RAY_U(float, false, false, true, true, 2 * kWavesPerBlock, true, false, 384, -1);
      else if (device_noise) MCL_RAY_U(float, false, false, true, true, 2 * kWavesPerBlock, true, false, 384, 0);
      else if (cap == 192 && arith == 5) MCL_RAY_U(float, false, false, false, true, kWavesPerBlock, true, false, 192, 5);
      else if (cap == 192 && even) MCL_RAY_U(float, false, false, false, true, kWavesPerBlock, true, false, 192, -1);
      else if (cap == 384 && arith == 10) MCL_RAY_U(float, false, false, false, true, 2 * kWavesPerBlock, true, false, 384, 10);
      else if (cap == 384 && even) MCL_RAY_U(float, false, false, false, true, 2 * kWavesPerBlock, true, false, 384, -1);
      else if (cap == 192) MCL_RAY_W(float, false, false, false, true, kWavesPerBlock, true, false, 192);
      else if (cap == 384) MCL_RAY_W(float, false, false, false, true, 2 * kWavesPerBlock, true, false, 384);
      else if (wpb != kWavesPerBlock) MCL_RAY_W(float, false, false, false, true, 2 * kWavesPerBlock, true, false, 0);
      else MCL_RAY_W(float, false, false, false, true, kWavesPerBlock, true, false, 0);
      MCL_HIP(c, hipGetLastError());
      timing_end(c);
      timing_begin(c, 5);
      // what is left over, entry by entry (no interval tables; the list's length is known to the device only: the
      // grid is sized for every particle and the blocks beyond the list leave at once -- a short list is bound by
      // the latency of one particle per wave, so it wants many waves rather than long strides)
      int64_t blocks_q = (n + wpb - 1) / wpb;
      const int64_t cap_q = 3584 * kWavesPerBlock / wpb;                    // (as below; 896 … 14336 blocks measured: 0.083, 0.066, 0.063, 0.066, 0.069 ms for the two launches)
      if (blocks_q > cap_q) blocks_q = cap_q;
      blocks_q = (blocks_q + kSubLists - 1) / kSubLists * kSubLists;
      if (every) {
        // first the arithmetic interval kernel once more, over the list: it serves the headings outside the circular
        // range as well (the records say which particles those are) and files what is then left -- particles with a
        // cell on a sector edge that intervals cannot take, headings that are not finite -- in a second list
        P.list = c->redo_list; P.list_count = c->redo_count;
        P.redo_list = c->redo_list2; P.redo_count = c->redo_count + kSubLists;
        const dim3 grid_q((unsigned)blocks_q);
#define MCL_RAY_Q(WW, CC, UU)                                                                                          \
  do {                                                                                                               \
    auto kern = raycast_kernel<float, false, false, false, WW, false, true, true, true, CC, UU>;                      \
    if (lds > 48 * 1024)                                                                                             \
      MCL_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                     (int)lds));                                                                     \
    hipLaunchKernelGGL(kern, grid_q, block, lds, c->stream, P, (const float*)x, (const float*)y, (const float*)th,   \
                       pm, n, w_out, logw_out, ranges_out, hits_out, qrec);                                          \
  } while (0)
        if (cap == 192) MCL_RAY_Q(kWavesPerBlock, 192, 5); else MCL_RAY_Q(2 * kWavesPerBlock, 384, 10);
#undef MCL_RAY_Q
        MCL_HIP(c, hipGetLastError());
      }
      P.redo_list = nullptr; P.redo_count = nullptr;
      P.list = every ? c->redo_list2 : c->redo_list; P.list_count = every ? c->redo_count + kSubLists : c->redo_count;
      P.iv_NE = 0; P.iv_cmax = 0;
      P.hint_out = c->redo_hint; P.hint_src = c->redo_count;    // (both lists' counts are final by now)
      int64_t blocks_b = (n + kWavesPerBlock - 1) / kWavesPerBlock;
      // (two rounds of the blocks a chip holds: measured at 2^20 particles with 33000 left over, 16384 blocks 0.070 ms,
      // 7168 0.067, 3584 0.063, 1792 0.062, 896 0.083)
      // (after the arithmetic kernel's second pass the list holds a handful: a small grid, whose fixed cost is small)
      if (blocks_b > (every ? 512 : 3584)) blocks_b = every ? 512 : 3584;
      blocks_b = (blocks_b + kSubLists - 1) / kSubLists * kSubLists;      // the same number of blocks behind every list
      {
        auto kern = device_noise ? raycast_kernel<float, false, false, false, kWavesPerBlock, true, true, false, true>
                                 : raycast_kernel<float, false, false, false, kWavesPerBlock, false, true, false, true>;
        if (lds_b > 48 * 1024)
          MCL_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks_b), block.x == 64u * kWavesPerBlock ? block : dim3(64 * kWavesPerBlock), lds_b, c->stream, P,
                           (const float*)x, (const float*)y, (const float*)th, pm, n, w_out, logw_out, ranges_out, hits_out,
                           qrec);
      }
      timing_end(c);
      c->every_last = every;
    }
    else if (fast && wpb != kWavesPerBlock) MCL_RAY_W(float, false, false, false, true, 2 * kWavesPerBlock, false, false, 0);
    else if (fast) MCL_RAY_F(float, false, false, false, true);
    else MCL_RAY(float, false, false);
  } else {
    if (product) MCL_RAY(double, false, true); else MCL_RAY(double, false, false);
  }
#undef MCL_RAY
#undef MCL_RAY_N
#undef MCL_RAY_F
#undef MCL_RAY_W
#undef MCL_RAY_U
  if (!stage && !split) timing_end(c);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

}  // namespace mcl
