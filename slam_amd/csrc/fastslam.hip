// FastSLAM 1.0 with unknown correspondence -- the reference's 13_fast_slam/fast_slam1.py
// (fast_slam(), :240-433) on the GPU.
//
// One thread per particle.  Kernels of a filter step: pose sample (Table 5.3 with the twelve-integer
// normal sampler, :160-189); measurement draws; maximum-likelihood association of every sensed landmark
// with a 2-sigma Mahalanobis gate (:286-315); a sweep over the features that forms the particle's weight the way
// the weight / feature-decay pass does (:377-404, it removes from the list it iterates); resampling; and then,
// only for the particles that were drawn, the sweep that applies the EKF updates (:348-375), the births (:318-347)
// and the decay, before the slots of the drawn ancestors are copied to their descendants.  The per-feature measurement
// prediction / Jacobian / innovation covariance of :259-282 is never stored: it is a function of the
// feature as it stood before the step and of the sampled pose, so the association sweep and the update
// each form it from the 48 bytes of the feature they read anyway (same expressions, same results), which
// spares a 96-byte record per feature slot written and read back every step.  Resampling reuses the
// order-exact sequential sum / CDF of the particle filter (weights.hip), then one kernel gathers the
// feature slots of the drawn ancestors.
//
// Storage is tiled by wavefront: the 64 particles of a tile keep all their feature slots next to each
// other -- word (field f of slot s of particle p) = (((p / 64) * F + s) * NF + (f & ~1)) * 64 + 2 (p % 64) + (f & 1),
// fields in pairs -- so a wavefront's accesses are 1-KB runs of 16 bytes per lane, the fields of one slot
// are adjacent runs, and everything a wavefront touches in a step lies in a few megabytes of one region.
//
// Every measurement is drawn per PARTICLE (the reference calls measure() inside the particle
// loop, :288): the caller passes the noise-free (r, phi) of each landmark from the true pose and
// either pre-drawn standard normals (stream parity) or a seed (device generator).
//
// Compiled with -ffp-contract=off.
#include <algorithm>
#include <cmath>
#include <numeric>
#include <cstring>
#include <vector>

#include "device_rng.h"
#include "mclhip_internal.h"
#include "libm_bits.h"

namespace mcl {

// fields of a feature slot (64-bit words; the last holds the two 32-bit counters)
enum { F_M0 = 0, F_M1, F_C00, F_C01, F_C10, F_C11, F_W, F_INT, F_WORDS };
constexpr int kSectors = 64;   // bearing sectors of the association sweep's landmark index (a power of two)
// per-step measurement: (r, phi), best likelihood, best feature (-1 none, -2 out of range)
enum { M_ZR = 0, M_ZPHI, M_BW, M_BC, M_WORDS };

struct FsParams {
  int64_t M;
  int F, L, mcap;           // feature capacity per particle, landmarks in sensing range of the true pose, rows of `meas`
  double v, w, dt, c1, c2, c3;
  double qt_r, qt_phi;      // std_r**2, std_phi**2
  double std_r, std_phi, r_max, r_max2, p0, gate2;
  const int16_t* ints;      // [M][3][12] or null
  const double* normals;    // [M][n_norm] or null
  int n_norm;
  uint64_t seed;
  int64_t index0;           // global index of local particle 0 (keys the device generators)
  double *x, *y, *th, *wout;
  int32_t *nfeat, *new_id;
  double* feat;             // tiles x F x F_WORDS x 64
  const uint8_t* shared_tile;   // per tile of `feat`: 1 = the slots of all its particles stand once, in lane 0's region
  double* meas;             // tiles x mcap x M_WORDS x 64
  int use_q;                // the association sweep keeps every measurement's noise in steps of sigma/32 in LDS (it fits)
  double far_r;             // largest noise-free range among the landmarks in range
  const unsigned long long* sectors;   // [blocks of 64 landmarks][kSectors] which landmarks' bearings lie in a sector; then one mask of all per block
  int32_t* overflow;
};

// this lane's word 0 of a record.  Fields are stored in pairs -- (f0, f1) of lane 0, (f0, f1) of lane 1, ...,
// then (f2, f3) of lane 0, ... -- so that neighbouring fields travel as one 16-byte access per lane.
__device__ __host__ inline int64_t tile_word(int64_t p, int64_t rows, int64_t row, int words) {
  return ((((p >> 6) * rows + row) * words) << 6) + ((p & 63) << 1);
}
#define FLD(rec, f) (rec)[((f) >> 1) * 128 + ((f) & 1)]
// `pf`: the particle whose region of the tile holds this particle's slots (itself, or lane 0 of a shared tile);
// `pm`: the particle whose measurements are meant
#define FREC(slot) (P.feat + tile_word(pf, P.F, (slot), F_WORDS))
#define MREC(u) (P.meas + tile_word(pm, P.mcap, (u), M_WORDS))
// After resampling, the 64 particles of a tile often descend from ONE ancestor (with hundreds of landmarks a few
// hundred ancestors feed a million particles): their feature slots are then identical, the tile keeps one copy -- in
// the region of its lane 0 -- and says so in a byte per tile.  Every lane of such a tile reads the same addresses.
__device__ inline int64_t slot_owner(const uint8_t* __restrict__ shared_tile, int64_t p) {
  return shared_tile[p >> 6] ? (p & ~(int64_t)63) : p;
}
#define FLD_I(rec, f, half) ((int32_t*)&FLD(rec, f))[half]   // 32-bit halves of a word

struct M2 { double a, b, c, d; };   // [[a, b], [c, d]]

__device__ inline M2 mul(const M2& p, const M2& q) {   // np.matmul of 2x2 matrices
  return M2{p.a * q.a + p.b * q.c, p.a * q.b + p.b * q.d, p.c * q.a + p.d * q.c, p.c * q.b + p.d * q.d};
}
__device__ inline M2 tr(const M2& p) { return M2{p.a, p.c, p.b, p.d}; }

// np.linalg.inv / det of a 2x2 matrix the way LAPACK's LU with partial pivoting goes about it
__device__ inline M2 inv_lu(const M2& m, double* det) {
  const bool swap = fabs(m.c) > fabs(m.a);
  const double p0 = swap ? m.c : m.a, p1 = swap ? m.d : m.b;   // pivot row
  const double r0 = swap ? m.a : m.c, r1 = swap ? m.b : m.d;
  // (three quotients by the pivot and two by u11: the reciprocal of each formed once, see DivBy -- the same bits)
  const DivBy by_p0 = make_div_by(p0);
  const double l = div_by_checked(r0, by_p0);
  const double u11 = r1 - l * p1;
  const DivBy by_u11 = make_div_by(u11);
  if (det) *det = swap ? -(p0 * u11) : (p0 * u11);
  // solve for the two columns of the (row-permuted) identity
  const double b0a = swap ? 0.0 : 1.0, b1a = swap ? 1.0 : 0.0;   // column 0
  const double b0b = swap ? 1.0 : 0.0, b1b = swap ? 0.0 : 1.0;   // column 1
  const double y1a = b1a - l * b0a, y1b = b1b - l * b0b;
  const double x1a = div_by_checked(y1a, by_u11), x1b = div_by_checked(y1b, by_u11);
  const double x0a = div_by_checked(b0a - p1 * x1a, by_p0), x0b = div_by_checked(b0b - p1 * x1b, by_p0);
  return M2{x0a, x0b, x1a, x1b};
}

__device__ inline void predict_measurement(double m0, double m1, double x, double y, double th, double* r, double* phi,
                                           M2* H) {
  const double dx = m0 - x, dy = m1 - y;
  const double q = dx * dx + dy * dy;
  const double rr = sqrt(q);
  double ph = atan2_bits(dy, dx) - th;        // :213
  if (ph > M_PI) ph -= M_PI * 2;
  else if (ph < -M_PI) ph += M_PI * 2;
  *r = rr; *phi = ph;
  const DivBy by_rr = make_div_by(rr), by_q = make_div_by(q);
  *H = M2{div_by_checked(dx, by_rr), div_by_checked(dy, by_rr), div_by_checked(-dy, by_q), div_by_checked(dx, by_q)}; // :233-236
}

__device__ inline double wrap_residual(double d) {
  if (d > M_PI) d -= M_PI * 2;                // :301-304
  else if (d < -M_PI) d += M_PI * 2;
  return d;
}

// four independent standard normals from one block of the counter-based generator (two float32 Box-Muller
// pairs).  The draws of a device-noise run are its own stream, not numpy's: what matters is their distribution,
// and a 24-bit uniform with single-precision transcendentals gives a normal whose error is far below the sensor
// noise it models.
__device__ inline void device_normal_quad(uint64_t particle, uint32_t draw, uint64_t seed, double out[4]) {
  uint32_t ctr[4] = {(uint32_t)particle, (uint32_t)(particle >> 32), draw, 0x4653u};
  philox4x32(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  box_muller_f32(ctr[0], ctr[1], &out[0], &out[1]);
  box_muller_f32(ctr[2], ctr[3], &out[2], &out[3]);
}

#ifndef FS_WAVES
#define FS_WAVES 3
#endif
#ifndef FS_BATCH
#define FS_BATCH 3
#endif

// ---- pose sample (:160-189) ----
__global__ __launch_bounds__(128) void fs_pose_kernel(const FsParams P) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.M) return;
  int k[36];
  if (P.ints) {
    const int16_t* src = P.ints + p * 36;
#pragma unroll
    for (int j = 0; j < 36; ++j) k[j] = (int)src[j];
  } else {
#pragma unroll
    for (int b = 0; b < 9; ++b) {
      const uint64_t gp = (uint64_t)(P.index0 + p);
      uint32_t ctr[4] = {(uint32_t)gp, (uint32_t)(gp >> 32), (uint32_t)b, 0x4d4fu};
      philox4x32(ctr, (uint32_t)P.seed, (uint32_t)(P.seed >> 32));
#pragma unroll
      for (int j = 0; j < 4; ++j) k[b * 4 + j] = (int)(((uint64_t)ctr[j] * 2000ull) >> 32) - 1000;   // randint(-1000, 1000)
    }
  }
  const double v_hat = P.v + P.c1 * sum12(k);
  const double w_hat = P.w + P.c2 * sum12(k + 12);
  const double g_hat = P.c3 * sum12(k + 24);
  const double x0 = P.x[p], y0 = P.y[p], th0 = P.th[p];
  const double q = v_hat / w_hat;
  P.x[p] = x0 - q * sin(th0) + q * sin(th0 + w_hat * P.dt);
  P.y[p] = y0 + q * cos(th0) - q * cos(th0 + w_hat * P.dt);
  P.th[p] = th0 + w_hat * P.dt + g_hat * P.dt;
}

// ---- per-feature prediction (:259-282): z_pred, H, inv Q, det Q from the feature as it stands before
// the step's updates and the sampled pose ----
struct RawFeat { double m0, m1, c00, c01, c10, c11; };
__device__ inline RawFeat load_raw(const double* f) {
  return RawFeat{FLD(f, F_M0), FLD(f, F_M1), FLD(f, F_C00), FLD(f, F_C01), FLD(f, F_C10), FLD(f, F_C11)};
}
struct Predicted { double fr, fphi, q00, q01, q10, q11, dq; };
__device__ inline Predicted predict_feature(const RawFeat& r, double x, double y, double th, double qt_r, double qt_phi,
                                            M2* H_out) {
  double zr, zphi;
  M2 H;
  predict_measurement(r.m0, r.m1, x, y, th, &zr, &zphi, &H);
  const M2 C{r.c00, r.c01, r.c10, r.c11};
  M2 Q = mul(mul(H, C), tr(H));              // :276
  Q.a = Q.a + qt_r; Q.b = Q.b + 0.0; Q.c = Q.c + 0.0; Q.d = Q.d + qt_phi;
  double det;
  const M2 iq = inv_lu(Q, &det);
  if (H_out) *H_out = H;
  return Predicted{zr, zphi, iq.a, iq.b, iq.c, iq.d, det};
}

// The out-of-line helpers below see plain values and this lane's record pointers only -- nothing they
// would have to read again after a store -- and their constants stay out of the registers of the loops.

// the likelihood of a pair that passed the gate (:308-309): eta depends on the feature alone and is formed once per
// feature, beside the reads of its candidates, not behind them; the exponential with the library's bits in fewer
// instructions (a positive argument -- an innovation covariance that is not positive definite -- takes the library's)
__device__ __attribute__((noinline)) double library_exp(double x) { return exp(x); }   // (out of line: its constants stay out of the loops' registers)
__device__ __forceinline__ double gate_eta(double dq) { return 1.0 / (2.0 * M_PI * sqrt(dq)); }
__device__ __forceinline__ double gate_weight(double d2, double eta) {
  const double arg = -0.5 * d2;
  return eta * (arg <= 0.0 ? exp_nonpos<true>(arg) : library_exp(arg));
}

// new feature (:318-347)
__device__ __attribute__((noinline)) void create_feature(double* f, int32_t id, double zr, double zphi, double x,
                                                         double y, double th, double qt_r, double qt_phi, double p0) {
  const double m0 = zr * cos(zphi + th) + x, m1 = zr * sin(zphi + th) + y;   // :201-202
  double r_, phi_;
  M2 H;
  predict_measurement(m0, m1, x, y, th, &r_, &phi_, &H);
  const M2 iH = inv_lu(H, nullptr);
  const M2 Qt{qt_r, 0.0, 0.0, qt_phi};
  const M2 C = mul(mul(iH, Qt), tr(iH));   // :336
  FLD(f, F_M0) = m0; FLD(f, F_M1) = m1;
  FLD(f, F_C00) = C.a; FLD(f, F_C01) = C.b; FLD(f, F_C10) = C.c; FLD(f, F_C11) = C.d;
  FLD(f, F_W) = p0;
  FLD_I(f, F_INT, 0) = 1; FLD_I(f, F_INT, 1) = id;
}

// ---- every landmark: draw the measurement, associate, create or update (:286-375) ----
// The association of a measurement reads only the prediction of the features that existed before this
// step, never what an earlier measurement of the same step changed.  So all measurements of the step are
// drawn first and parked in device memory (draw_measurements, at the head of the association kernel: the stores
// drain while it computes), ONE sweep over the stored features associates
// them all (fs_associate_kernel), and then they are applied (fs_update_kernel): updates feature by feature
// -- two measurements may pick the same feature: they are applied one after the other, in landmark order,
// with the one prediction -- and births in landmark order; an update touches a feature that existed
// before the step, a birth a new slot, so the two do not meet.
//
// A wavefront's time goes into waiting for memory, so each kernel is arranged to have few dependent round
// trips: loads a step ahead of their use, stores that nothing waits for, out-of-line calls (which drain
// the memory queue) only where nothing is in flight.

// one noisy measurement per landmark in range (:145-152), parked in device memory for the sweeps that follow; returns
// the largest range / bearing noise this particle drew.  s_q (nullable): this thread's column of the noise in steps
// of sigma / 32, two signed bytes per measurement.
__device__ inline void draw_measurements(const FsParams& P, int64_t p, const double2* __restrict__ truth, uint16_t* s_q,
                                         double* mr_out, double* mphi_out) {
  const int nv = P.L;
  const int64_t pm = p;
  const double* normals = P.normals ? P.normals + p * P.n_norm : nullptr;
  double mr = 0.0, mphi = 0.0;
  for (int u0 = 0; u0 < nv; u0 += 2) {
    double nn[4] = {0.0, 0.0, 0.0, 0.0};   // (range, bearing) noise of landmarks u0 and u0 + 1
    if (normals) {
      nn[0] = normals[2 * u0]; nn[1] = normals[2 * u0 + 1];
      if (u0 + 1 < nv) { nn[2] = normals[2 * u0 + 2]; nn[3] = normals[2 * u0 + 3]; }
    } else {
      device_normal_quad((uint64_t)(P.index0 + p), (uint32_t)(u0 >> 1), P.seed, nn);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int u = u0 + h;
      if (u < nv) {
        const double2 t = truth[u];
        const double n_r = nn[2 * h], n_phi = nn[2 * h + 1];
        const double zr = t.x + n_r * P.std_r;
        double zphi = t.y + n_phi * P.std_phi;
        if (zphi > M_PI) zphi -= 2 * M_PI;
        else if (zphi < -M_PI) zphi += 2 * M_PI;
        mr = fmax(mr, fabs(n_r * P.std_r)); mphi = fmax(mphi, fabs(n_phi * P.std_phi));
        const bool in_range = zr <= P.r_max;        // :291
        if (s_q) {
          // the noise again in a sixteenth of the space: steps of sigma / 32, so that the association sweep can bound
          // this reading to within sigma / 64 without fetching it.  Bytes: (-128, *) = beyond +-3.9 sigma, no bound;
          // (*, -128) = out of range, never a candidate.
          const double s_r = rint(n_r * 32.0), s_phi = rint(n_phi * 32.0);
          const bool wide = !(fabs(s_r) <= 127.0) || !(fabs(s_phi) <= 127.0);
          const int b_r = wide ? -128 : (int)s_r, b_phi = !in_range ? -128 : (wide ? 0 : (int)s_phi);
          s_q[u * 128] = (uint16_t)((b_r & 0xff) | ((b_phi & 0xff) << 8));
        }
        double* m = MREC(u);
        FLD(m, M_ZR) = in_range ? zr : __longlong_as_double(0x7ff8000000000000ll);   // NaN never passes a gate
        FLD(m, M_ZPHI) = zphi;
        FLD(m, M_BW) = 0.0;
        FLD_I(m, M_BC, 0) = in_range ? -1 : -2;
      }
    }
  }
  *mr_out = mr;
  *mphi_out = mphi;
}

// The sweep looks at a (feature, measurement) pair closely only if it can pass the gate at all.  For a
// positive definite Q, r' invQ r >= r_k^2 / Q_kk: each residual alone must be within the gate of its own
// variance (Q_00 = invQ_11 det Q, Q_11 = invQ_00 det Q).  A measurement differs from the landmark's
// noise-free reading -- the same for every particle -- by at most the largest noise this particle drew,
// and the angular residual is at least the distance on the circle.  The test uses twice the gate, far
// beyond any rounding of the quantities involved.  Of the landmarks only those are tested whose noise-free
// bearing lies in a sector the feature's window touches: the host files the landmarks of a step by
// bearing sector (one 64-bit set per sector and block of 64 landmarks), a lane joins the sets of its
// window and walks the few bits.  The table and the sets are staged in LDS; the candidates of a feature
// are then read together.
__global__ __launch_bounds__(128, FS_WAVES) void fs_associate_kernel(const FsParams P, const double2* __restrict__ truth) {
  extern __shared__ __align__(16) unsigned char fs_assoc_lds[];
  const int nv = P.L;
  const int nv8 = (nv + 7) & ~7;                // the table is padded to a multiple of eight with far-away readings
  const int nblk = (nv + 63) >> 6;
  unsigned long long* const s_sect = reinterpret_cast<unsigned long long*>(fs_assoc_lds);   // [nblk][kSectors] + [nblk]
  float2* const s_tf = reinterpret_cast<float2*>(s_sect + nblk * (kSectors + 1));            // [nv8] noise-free (r, phi) of the landmarks, float32
  uint16_t* const s_q = reinterpret_cast<uint16_t*>(s_tf + nv8);                            // [nv][128], this thread's column
  for (int k = threadIdx.x; k < nv8; k += blockDim.x) {
    const double2 t = truth[k];
    s_tf[k] = make_float2((float)t.x, (float)t.y);
  }
  for (int k = threadIdx.x; k < nblk * (kSectors + 1); k += blockDim.x) s_sect[k] = P.sectors[k];
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.M) return;
  const bool use_q = P.use_q != 0;
  // the measurements first: their stores drain while the sweep below computes (a thread reads back only its own)
  double mr, mphi;
  draw_measurements(P, p, truth, use_q ? s_q + threadIdx.x : nullptr, &mr, &mphi);
  // The cheap test runs in float32: it only has to keep every pair that can pass the gate, it uses twice the gate,
  // and what float32 rounding can move (a few units in the seventh digit of a range, 1e-6 rad) is added to the
  // widths below; the close test that follows is the reference's float64 arithmetic.
  const float q_r = (float)(P.std_r * (1.0 / 32.0)), q_phi = (float)(P.std_phi * (1.0 / 32.0));
  const float h_phi = (float)(P.std_phi * (1.0 / 64.0)) + 1e-5f;
  const float h_r0 = (float)(P.std_r * (1.0 / 64.0)), far_rf = (float)P.far_r * 1.00001f + 1e-4f;
  const int n_old = P.nfeat[p];
  if (n_old == 0) return;
  const int64_t pm = p, pf = slot_owner(P.shared_tile, p);
  const double x = P.x[p], y = P.y[p], th = P.th[p];
  const float mrf = (float)mr * 1.00001f + 1e-6f, mphif = (float)mphi * 1.00001f + 1e-5f;
  const double two_gate = 2.0 * P.gate2;
  const float two_gatef = (float)two_gate * 1.00001f;
  const double inv_sector = (double)kSectors / (2.0 * M_PI);
  constexpr int kBatch = FS_BATCH;
  RawFeat next = load_raw(FREC(0));
  for (int j = 0; j < n_old; ++j) {
    const RawFeat raw = next;
    if (j + 1 < n_old) next = load_raw(FREC(j + 1));   // in flight during this feature's prediction and tests
    {
      // no reading can pass the range gate of a feature that lies further out than the farthest landmark in
      // range + this particle's largest range noise + the gate of the largest range variance the feature can
      // have (|h' C h| <= sum |C_ij| for a unit h): such a feature needs no prediction at all
      // (float32 with every bound rounded outwards: a feature skipped here is far beyond the gate)
      const float dx = (float)(raw.m0 - x), dy = (float)(raw.m1 - y);
      const float q_cap = (fabsf((float)raw.c00) + fabsf((float)raw.c01)) + (fabsf((float)raw.c10) + fabsf((float)raw.c11)) + (float)P.qt_r;
      const float lim = (far_rf + mrf) + sqrtf(two_gatef * q_cap) * 1.00001f;
      if ((dx * dx + dy * dy) * 0.9999f > lim * lim) continue;
    }
    const Predicted f = predict_feature(raw, x, y, th, P.qt_r, P.qt_phi, nullptr);
    const double eta_j = gate_eta(f.dq);
    const double lim0 = two_gate * (f.q11 * f.dq), lim1 = two_gate * (f.q00 * f.dq);
    const float lim0f = (float)lim0 * 1.0001f, lim1f = (float)lim1 * 1.0001f;      // NaN: nothing is out, the close test decides
    const float ffr = (float)f.fr, ffphi = (float)f.fphi;
    const float h_r = h_r0 + 2e-6f * (fabsf(ffr) + far_rf + 1.0f);                 // float32 rounding of ranges this large
    // a reading can pass only if its noise-free bearing is within `reach` of the predicted one on the circle
    const double reach = mphi + sqrt(lim1) + 1e-6;
    const bool everything = !(fabs(f.fphi) < 1.0e6) || !(reach < 3.0);   // also NaN
    const int b_lo = everything ? 0 : (int)floor((f.fphi - reach + M_PI) * inv_sector);
    const int b_hi = everything ? -1 : (int)floor((f.fphi + reach + M_PI) * inv_sector);
    for (int blk = 0; blk < nblk; ++blk) {
      const int base = blk << 6;
      unsigned long long look = everything ? s_sect[nblk * kSectors + blk] : 0ull;
      for (int b = b_lo; b <= b_hi; ++b) look |= s_sect[blk * kSectors + (b & (kSectors - 1))];
      unsigned long long cand = 0;                // measurements that may pass this feature's gate
      while (look) {
        const int k = __ffsll((long long)look) - 1;
        look &= look - 1;
        const float2 t = s_tf[base + k];
        // this particle's reading of the landmark: known to sigma / 64 from the staged bytes, else only to within
        // the largest noise the particle drew
        float n0 = 0.0f, n1 = 0.0f, w0 = mrf, w1 = mphif;
        bool never = false;
        if (use_q) {
          const int qq = s_q[(base + k) * 128 + threadIdx.x];
          const int b0 = (int)(signed char)(qq & 0xff), b1 = (int)(signed char)(qq >> 8);
          never = b1 == -128;
          if (b0 != -128) { n0 = (float)b0 * q_r; n1 = (float)b1 * q_phi; w0 = h_r; w1 = h_phi; }
        }
        const float g0 = fmaxf(fabsf((t.x + n0) - ffr) - w0, 0.0f);
        const float d = (t.y + n1) - ffphi;
        const float g1 = fmaxf(fabsf(d - 6.2831853f * rintf(d * 0.15915494f)) - w1, 0.0f);
        const bool out = never || (g0 * g0 > lim0f) || (g1 * g1 > lim1f);
        if (!out) cand |= 1ull << k;
      }
      while (cand) {
        int uu[kBatch];
        double bw[kBatch], zr[kBatch], zp[kBatch], wj[kBatch];
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
          uu[k] = -1;
          if (cand) {
            uu[k] = base + __ffsll((long long)cand) - 1;
            cand &= cand - 1;
            const double* m = MREC(uu[k]);
            bw[k] = FLD(m, M_BW); zr[k] = FLD(m, M_ZR); zp[k] = FLD(m, M_ZPHI);
          }
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
          wj[k] = -1.0;
          if (uu[k] >= 0) {
            const double e0 = zr[k] - f.fr;
            const double e1 = wrap_residual(zp[k] - f.fphi);
            const double t0 = e0 * f.q00 + e1 * f.q10;   // r.T @ invQ
            const double t1 = e0 * f.q01 + e1 * f.q11;
            const double d2 = t0 * e0 + t1 * e1;
            if (d2 <= P.gate2) {                          // :307
              const double w = gate_weight(d2, eta_j);
              if (bw[k] < w) wj[k] = w;                   // :311 (a NaN likelihood never wins)
            }
          }
        }
#pragma unroll
        for (int k = 0; k < kBatch; ++k) {
          if (wj[k] >= 0.0) { double* m = MREC(uu[k]); FLD(m, M_BW) = wj[k]; FLD_I(m, M_BC, 0) = j; }
        }
      }
    }
  }
}

// a feature slot in registers
struct Slot { double m0, m1; M2 C; double w; int cnt, id; };
__device__ inline Slot load_slot(const double* f) {
  Slot d;
  d.m0 = FLD(f, F_M0); d.m1 = FLD(f, F_M1);
  d.C = M2{FLD(f, F_C00), FLD(f, F_C01), FLD(f, F_C10), FLD(f, F_C11)};
  d.w = FLD(f, F_W);
  d.cnt = FLD_I(f, F_INT, 0); d.id = FLD_I(f, F_INT, 1);
  return d;
}
__device__ inline void store_slot(double* f, const Slot& d) {   // whole 16-byte pairs: no partial lines
  FLD(f, F_M0) = d.m0; FLD(f, F_M1) = d.m1;
  FLD(f, F_C00) = d.C.a; FLD(f, F_C01) = d.C.b; FLD(f, F_C10) = d.C.c; FLD(f, F_C11) = d.C.d;
  FLD(f, F_W) = d.w;
  FLD_I(f, F_INT, 0) = d.cnt; FLD_I(f, F_INT, 1) = d.id;
}

// EKF update (:348-375) with the step's prediction pr / H of the feature and its current mean and covariance
__device__ inline void ekf_update(Slot& d, const Predicted& pr, const M2& H, double zr, double zphi, double w_max) {
  const M2 iq{pr.q00, pr.q01, pr.q10, pr.q11};
  const M2 K = mul(mul(d.C, tr(H)), iq);       // :360
  const double e0 = zr - pr.fr;
  const double e1 = wrap_residual(zphi - pr.fphi);
  d.m0 = d.m0 + (K.a * e0 + K.b * e1);         // :369
  d.m1 = d.m1 + (K.c * e0 + K.d * e1);
  const M2 KH = mul(K, H);
  const M2 IKH{1.0 - KH.a, 0.0 - KH.b, 0.0 - KH.c, 1.0 - KH.d};
  d.C = mul(IKH, d.C);                         // :370
  d.w = w_max;
  d.cnt = d.cnt + 1;
}

#ifndef FS_UPDATE_WAVES
#define FS_UPDATE_WAVES 3
#endif
// ---- EKF updates (:348-375), births (:318-347) and the weight / feature-decay pass (:377-404) as ONE sweep
// over the particle's features in list order.  What a feature needs to know -- which measurement picked it --
// is the inverse of the association, built per thread in LDS (per live slot the first measurement that picked
// it, per measurement the next one that picked the same feature).  The reference removes from the list its
// for-loop walks, so the feature that slides into a vacated slot is not visited by the pass (`slid`); births
// stand behind the old features in the list and are visited last.
//
// The sweep runs twice per filter step.  FS_WEIGH, for every particle, forms the particle's weight and changes
// nothing: a weight needs, per feature, whether it was picked (then the likelihood of its last picker) or else
// its position and counter (the decay visit decides whether the NEXT feature is visited) -- 32 of a slot's 64
// bytes, and no arithmetic of the update.  FS_APPLY, after resampling, for the particles that were drawn at least
// once (`state[p] == 1`): reads every slot once, a step ahead of its use, applies the updates and the decay,
// writes the slot once at its compacted position, appends the births.  A particle nobody descends from is never
// updated: its slots are dead once the survivors' slots have been copied.
// IT: unsigned char while fewer than 255 measurements are in range, else unsigned short.
// Dynamic LDS: births [blocks of 64 measurements][128] u64, then the inverse [rows][128] IT, rows >= every
// particle's feature count, then the chains [measurements][128] IT.
enum { FS_WEIGH = 0, FS_APPLY = 1, FS_COPY = 2 };
// Where FS_COPY writes: the other buffer of the resampled set.
struct FsDest {
  const int64_t* idx;       // ancestor of every position
  double* feat;
  int32_t *nfeat, *new_id;
  uint8_t* shared_tile;
};
template <typename IT, int MODE>
__global__ __launch_bounds__(128, FS_UPDATE_WAVES) void fs_update_kernel(const FsParams P, int rows, uint8_t* __restrict__ state,
                                                                         const FsDest D) {
  extern __shared__ __align__(16) unsigned char fs_update_lds[];
  constexpr unsigned kNone = (IT)~(IT)0;
  const int tid = threadIdx.x;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + tid;
  if (p >= P.M) return;
  if (MODE == FS_APPLY && state[p] != 1) return;
  // q: the particle whose step is meant -- this one, or (FS_COPY) the ancestor of position p.  When the 64 positions
  // of a tile have one ancestor, the tile keeps one copy and its 64 lanes share the work: lane 0 files the picks, then
  // every lane takes a feature.
  int64_t q = p;
  bool one_ancestor = false;
  if constexpr (MODE == FS_COPY) {
    q = D.idx[p];
    const int64_t q0 = __shfl(q, 0);
    one_ancestor = (__ballot(1) == ~0ull) && (__ballot(q != q0) == 0ull);
  }
  const bool worker = !one_ancestor || (tid & 63) == 0;
  int kept = 0;            // slots written so far: survivors close ranks as the pass goes
  int32_t new_id = 0;
  const int64_t pm = q, pf = slot_owner(P.shared_tile, q);
  const int nv = P.L, nblk = (nv + 63) >> 6;
  unsigned long long* const s_birth = reinterpret_cast<unsigned long long*>(fs_update_lds);   // [nblk][128]
  IT* const s_inv = reinterpret_cast<IT*>(s_birth + (size_t)nblk * 128);                     // [rows][128]
  IT* const s_next = s_inv + (size_t)rows * 128;                                              // [nv][128]
  const double x = P.x[q], y = P.y[q], th = P.th[q];
  const int n_old = min(P.nfeat[q], rows);
  if (worker) {
  for (int j = 0; j < n_old; ++j) s_inv[j * 128 + tid] = (IT)kNone;
  // the association, inverted: sixteen picks are read together, then filed -- last measurement first, so that a
  // slot ends up with its first picker and every picker points to the next one of the same feature
  for (int blk = nblk - 1; blk >= 0; --blk) {
    unsigned long long birth = 0;
    const int k_end = min(nv, (blk + 1) << 6);
    for (int k0 = (blk << 6) + ((k_end - 1 - (blk << 6)) & ~15); k0 >= (blk << 6); k0 -= 16) {
      int c[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) c[k] = FLD_I(MREC(min(k0 + k, nv - 1)), M_BC, 0);
#pragma unroll
      for (int k = 15; k >= 0; --k) {
        if (k0 + k < nv) {
          if (c[k] >= 0 && c[k] < n_old) {
            s_next[(k0 + k) * 128 + tid] = s_inv[c[k] * 128 + tid];
            s_inv[c[k] * 128 + tid] = (IT)(k0 + k);
          } else if (c[k] == -1) {
            birth |= 1ull << ((k0 + k) & 63);
          }
        }
      }
    }
    s_birth[blk * 128 + tid] = birth;
  }
  }  // worker: the picks are filed
  bool slid = false;       // the previous feature was discarded, this one takes its place unvisited
  if (MODE == FS_COPY && one_ancestor) {
    // ---- one ancestor for the whole tile: lane l works on features l, l + 64, ... and on the l-th birth; what the
    // pass over the list decides (:377-404 -- a discarded feature makes the pass skip its successor) is a short
    // scalar loop over three wave masks; the one copy goes to lane 0's region of the destination tile
    const int lane = tid & 63, col = tid & ~63;   // lane 0's column of the tables
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int64_t p0 = p & ~(int64_t)63;
    new_id = P.new_id[q];
    for (int j0 = 0; j0 < n_old; j0 += 64) {
      const int j = j0 + lane;
      const bool valid = j < n_old;
      Slot cur{1.0, 0.0, M2{1.0, 0.0, 0.0, 1.0}, 0.0, 0, 0};
      bool upd = false, in_range = false, dflag = false;
      if (valid) {
        cur = load_slot(FREC(j));
        const unsigned e = s_inv[j * 128 + col];
        upd = e != kNone;
        if (upd) {
          const double* m = MREC(e);
          M2 H;
          const Predicted pr = predict_feature(RawFeat{cur.m0, cur.m1, cur.C.a, cur.C.b, cur.C.c, cur.C.d}, x, y, th, P.qt_r, P.qt_phi, &H);
          ekf_update(cur, pr, H, FLD(m, M_ZR), FLD(m, M_ZPHI), FLD(m, M_BW));
          for (unsigned u2 = s_next[e * 128 + col]; u2 != kNone; u2 = s_next[u2 * 128 + col]) {
            const double* m2 = MREC(u2);
            ekf_update(cur, pr, H, FLD(m2, M_ZR), FLD(m2, M_ZPHI), FLD(m2, M_BW));
          }
        } else {
          const double dx = cur.m0 - x, dy = cur.m1 - y;
          in_range = dx * dx + dy * dy <= P.r_max2;
          dflag = in_range && (cur.cnt - 1 < 0);      // discarded if the pass visits it
        }
      }
      const unsigned long long umask = __ballot(upd), dmask = __ballot(dflag);
      const int n_here = min(n_old - j0, 64);
      unsigned long long visited = 0, discard = 0;
      for (int b = 0; b < n_here; ++b) {
        if (slid) { slid = false; continue; }
        visited |= 1ull << b;
        if (!((umask >> b) & 1) && ((dmask >> b) & 1)) { discard |= 1ull << b; slid = true; }
      }
      if (valid && !upd && ((visited >> lane) & 1)) {
        cur.w = P.p0;
        if (in_range) cur.cnt = cur.cnt - 1;
      }
      const bool keep = valid && !((discard >> lane) & 1);
      const unsigned long long kmask = __ballot(keep);
      if (keep) store_slot(D.feat + tile_word(p0, P.F, kept + __popcll(kmask & ((1ull << lane) - 1ull)), F_WORDS), cur);
      kept += __popcll(kmask);
    }
    int rank0 = 0;           // births before this block of measurements
    for (int blk = 0; blk < nblk; ++blk) {
      const unsigned long long births = s_birth[blk * 128 + col];
      const int nb = __popcll(births);
      const int r = rank0 + lane;
      if (lane < nb && n_old + r < P.F) {          // (more than the capacity: reported by the weighing sweep)
        unsigned long long mk = births;
        for (int k = 0; k < lane; ++k) mk &= mk - 1;
        const int u = (blk << 6) + __ffsll((long long)mk) - 1;
        const double* m = MREC(u);
        create_feature(D.feat + tile_word(p0, P.F, kept + r, F_WORDS), new_id + r, FLD(m, M_ZR), FLD(m, M_ZPHI), x, y, th, P.qt_r, P.qt_phi, P.p0);
      }
      rank0 += nb;
    }
    const int created = max(0, min(rank0, P.F - n_old));
    kept += created;
    new_id += created;
  } else if (worker) {
  if constexpr (MODE == FS_WEIGH) {
    double wgt = 1.0;
    int n_assoc = 0;
    // slot j + 1's position and counter, and the likelihood of the last measurement that picked it, are on their
    // way while slot j is looked at
    struct Ahead { double m0, m1, bw; int cnt; bool upd; };
    auto fetch = [&](int jj) {
      Ahead a{0.0, 0.0, 0.0, 0, false};
      if (jj < n_old) {
        const double* f = FREC(jj);
        a.m0 = FLD(f, F_M0); a.m1 = FLD(f, F_M1); a.cnt = FLD_I(f, F_INT, 0);
        unsigned e = s_inv[jj * 128 + tid];
        if (e != kNone) {
          for (unsigned u2 = s_next[e * 128 + tid]; u2 != kNone; u2 = s_next[u2 * 128 + tid]) e = u2;
          a.upd = true;
          a.bw = FLD(MREC(e), M_BW);
        }
      }
      return a;
    };
    Ahead a1 = fetch(0);
    for (int j = 0; j < n_old; ++j) {
      const Ahead cur = a1;
      a1 = fetch(j + 1);
      if (slid) {
        slid = false;
      } else if (cur.upd) {
        wgt = wgt * cur.bw;
        ++n_assoc;
      } else {
        const double dx = cur.m0 - x, dy = cur.m1 - y;
        if (dx * dx + dy * dy <= P.r_max2 && cur.cnt - 1 < 0) slid = true;   // discarded
      }
    }
    int n_list = n_old;
    for (int blk = 0; blk < nblk; ++blk) {
      int births = __popcll(s_birth[blk * 128 + tid]);
      for (; births > 0; --births) {
        if (n_list >= P.F) { atomicMax(P.overflow, 1); continue; }
        ++n_list;
        if (slid) slid = false;
        else { wgt = wgt * P.p0; ++n_assoc; }
      }
    }
    if (n_assoc == 0) wgt = P.p0;
    P.wout[p] = wgt;
  } else {
    // FS_APPLY writes the particle's own region in place (the host has made every tile private before);
    // FS_COPY writes position p's region of the other buffer
    double* const out = MODE == FS_COPY ? D.feat : P.feat;
    new_id = P.new_id[q];
    // slot j + 1 and the measurement that picked it are on their way while slot j is worked on (two ahead: no gain)
    struct Ahead { Slot sl; unsigned e; double zr, zp, bw; };
    auto fetch = [&](int jj) {
      Ahead a;
      a.sl = Slot{1.0, 0.0, M2{1.0, 0.0, 0.0, 1.0}, 0.0, 0, 0};
      a.e = kNone; a.zr = a.zp = a.bw = 0.0;
      if (jj < n_old) {
        a.sl = load_slot(FREC(jj));
        a.e = s_inv[jj * 128 + tid];
        if (a.e != kNone) { const double* m = MREC(a.e); a.zr = FLD(m, M_ZR); a.zp = FLD(m, M_ZPHI); a.bw = FLD(m, M_BW); }
      }
      return a;
    };
    Ahead a1 = fetch(0);
    for (int j = 0; j < n_old; ++j) {
      Slot cur = a1.sl;
      const unsigned e = a1.e;
      const double zr = a1.zr, zp = a1.zp, bw = a1.bw;
      a1 = fetch(j + 1);
      const bool upd = e != kNone;
      if (upd) {
        M2 H;
        const Predicted pr = predict_feature(RawFeat{cur.m0, cur.m1, cur.C.a, cur.C.b, cur.C.c, cur.C.d}, x, y, th, P.qt_r, P.qt_phi, &H);
        ekf_update(cur, pr, H, zr, zp, bw);
        // rare: the other measurements that picked this feature, in landmark order, same prediction
        for (unsigned u2 = s_next[e * 128 + tid]; u2 != kNone; u2 = s_next[u2 * 128 + tid]) {
          const double* m = MREC(u2);
          ekf_update(cur, pr, H, FLD(m, M_ZR), FLD(m, M_ZPHI), FLD(m, M_BW));
        }
      }
      bool keep = true, touched = upd;
      if (slid) {
        slid = false;
      } else if (!upd) {
        const double dx = cur.m0 - x, dy = cur.m1 - y;
        cur.w = P.p0;
        touched = true;
        if (dx * dx + dy * dy <= P.r_max2) {
          cur.cnt = cur.cnt - 1;
          if (cur.cnt < 0) { keep = false; slid = true; }   // discard
        }
      }
      if (keep) {
        double* g = out + tile_word(p, P.F, kept, F_WORDS);
        if (MODE == FS_COPY || upd || kept != j) {
          store_slot(g, cur);
        } else if (touched) {                    // only the weight / counter pair changed
          FLD(g, F_W) = cur.w;
          FLD_I(g, F_INT, 0) = cur.cnt; FLD_I(g, F_INT, 1) = cur.id;
        }
        ++kept;
      }
    }
    // births, in landmark order; n_list: length of the reference's list before anything is discarded
    int n_list = n_old;
    for (int blk = 0; blk < nblk; ++blk) {
      unsigned long long births = s_birth[blk * 128 + tid];
      while (births) {
        const int u = (blk << 6) + __ffsll((long long)births) - 1;
        births &= births - 1;
        if (n_list >= P.F) continue;             // reported by the weighing sweep
        const double* m = MREC(u);
        create_feature(out + tile_word(p, P.F, kept, F_WORDS), new_id++, FLD(m, M_ZR), FLD(m, M_ZPHI), x, y, th, P.qt_r, P.qt_phi, P.p0);
        ++n_list;
        ++kept;
      }
    }
  }
  }  // one particle's sweep
  if constexpr (MODE == FS_APPLY) {
    P.nfeat[p] = kept;
    P.new_id[p] = new_id;
    state[p] = 2;            // applied: a second request for this particle in the same step finds nothing to do
  }
  if constexpr (MODE == FS_COPY) {
    D.nfeat[p] = kept;
    D.new_id[p] = new_id;
    if ((tid & 63) == 0) D.shared_tile[p >> 6] = one_ancestor ? 1 : 0;
  }
}

// every particle of a shared tile gets its own copy of the tile's slots (block row y = slot); the caller then clears the
// tile flags.  For the paths that address a particle's own region: in-place updates, row packing, state read-out.
__global__ void fs_unshare_kernel(const uint8_t* __restrict__ shared_tile, int64_t M, int F, const int32_t* __restrict__ nfeat,
                                  double* __restrict__ feat) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = blockIdx.y;
  if (p >= M || (p & 63) == 0 || !shared_tile[p >> 6] || slot >= nfeat[p]) return;
  const double* a = feat + tile_word(p & ~(int64_t)63, F, slot, F_WORDS);
  double* b = feat + tile_word(p, F, slot, F_WORDS);
#pragma unroll
  for (int w = 0; w < F_WORDS; ++w) FLD(b, w) = FLD(a, w);
}

// marks the particles of a list (ancestor indices, selected rows) whose update has not been applied yet
__global__ void fs_mark_kernel(const int64_t* __restrict__ list, int64_t n, int64_t M, uint8_t* __restrict__ state) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t p = list[i];
  if (p >= 0 && p < M && state[p] == 0) state[p] = 1;
}
__global__ void fs_mark_range_kernel(int64_t lo, int64_t hi, uint8_t* __restrict__ state) {
  const int64_t p = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < hi && state[p] == 0) state[p] = 1;
}

// largest feature count over the particles (sizes the update sweep's LDS table of the next step)
__global__ void fs_max_count_kernel(const int32_t* __restrict__ nfeat, int64_t M, int32_t* __restrict__ out) {
  int v = 0;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < M; p += (int64_t)gridDim.x * blockDim.x) v = max(v, nfeat[p]);
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, v);
}

// largest feature count among the particles of a list
__global__ void fs_max_sel_kernel(const int64_t* __restrict__ sel, int64_t n, const int32_t* __restrict__ nfeat, int32_t* __restrict__ out) {
  int v = 0;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) v = max(v, nfeat[sel[j]]);
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) atomicMax(out, v);
}

__global__ void fs_total_kernel(const double* __restrict__ cdf, int64_t n, double floor_value, double* __restrict__ scal) {
  if (threadIdx.x == 0 && blockIdx.x == 0) scal[3] = fmax(floor_value, cdf[n - 1]);   // max(1e-8, sum_w) (:415)
}

__global__ void fs_divide_kernel(double* __restrict__ w, int64_t n, const double* __restrict__ scal) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = w[i] / scal[3];   // :418
}

// Feature rows for the exchange between ranks: per particle 2 + 9 F 64-bit words --
// [nfeat, new_id, then per slot: m0 m1 c00 c01 c10 c11 w (doubles), i, id].  Block row y = 0 handles the
// two counters, y = 1 + slot the slots.
__global__ void fs_pack_rows_kernel(int64_t lo, int64_t n, int F, int R, const int32_t* __restrict__ nfeat,
                                    const int32_t* __restrict__ newid, const double* __restrict__ feat,
                                    unsigned long long* __restrict__ out, int32_t* __restrict__ overflow) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int64_t p = lo + j;
  unsigned long long* row = out + j * (2 + (int64_t)9 * R);   // R slots per row
  const int y = blockIdx.y;
  if (y == 0) {
    row[0] = (unsigned long long)(long long)nfeat[p]; row[1] = (unsigned long long)(long long)newid[p];
    if (nfeat[p] > R) atomicMax(overflow, 2);                   // the row cannot hold this particle's features
    return;
  }
  const int slot = y - 1;
  if (slot >= nfeat[p]) return;
  unsigned long long* dst = row + 2 + (int64_t)9 * slot;
  const double* f = feat + tile_word(p, F, slot, F_WORDS);
#pragma unroll
  for (int w = 0; w < 7; ++w) dst[w] = (unsigned long long)__double_as_longlong(FLD(f, w));
  dst[7] = (unsigned long long)(long long)FLD_I(f, F_INT, 0);
  dst[8] = (unsigned long long)(long long)FLD_I(f, F_INT, 1);
}

// the same for a list of local particles: row j of the output holds particle sel[j]
__global__ void fs_pack_sel_kernel(const int64_t* __restrict__ sel, int64_t n, int F, int R, const int32_t* __restrict__ nfeat,
                                   const int32_t* __restrict__ newid, const double* __restrict__ feat,
                                   unsigned long long* __restrict__ out, int32_t* __restrict__ overflow) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int64_t p = sel[j];
  unsigned long long* row = out + j * (2 + (int64_t)9 * R);
  const int y = blockIdx.y;
  if (y == 0) {
    row[0] = (unsigned long long)(long long)nfeat[p]; row[1] = (unsigned long long)(long long)newid[p];
    if (nfeat[p] > R) atomicMax(overflow, 2);
    return;
  }
  const int slot = y - 1;
  if (slot >= nfeat[p]) return;
  unsigned long long* dst = row + 2 + (int64_t)9 * slot;
  const double* f = feat + tile_word(p, F, slot, F_WORDS);
#pragma unroll
  for (int w = 0; w < 7; ++w) dst[w] = (unsigned long long)__double_as_longlong(FLD(f, w));
  dst[7] = (unsigned long long)(long long)FLD_I(f, F_INT, 0);
  dst[8] = (unsigned long long)(long long)FLD_I(f, F_INT, 1);
}

__global__ void fs_unpack_rows_kernel(const int64_t* __restrict__ idx, int64_t row0_global, int64_t L, int64_t M, int F, int R,
                                      const unsigned long long* __restrict__ rows, int32_t* __restrict__ nfeat,
                                      int32_t* __restrict__ newid, double* __restrict__ feat) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  int64_t rr = idx[p] - row0_global;   // row0_global == 0 with idx = the received-row positions themselves
  rr = rr < 0 ? 0 : (rr > L - 1 ? L - 1 : rr);
  const unsigned long long* row = rows + rr * (2 + (int64_t)9 * R);
  const int n = min((int)(long long)row[0], R);
  const int y = blockIdx.y;
  if (y == 0) { nfeat[p] = n; newid[p] = (int)(long long)row[1]; return; }
  const int slot = y - 1;
  if (slot >= n) return;
  const unsigned long long* src = row + 2 + (int64_t)9 * slot;
  double* f = feat + tile_word(p, F, slot, F_WORDS);
#pragma unroll
  for (int w = 0; w < 7; ++w) FLD(f, w) = __longlong_as_double((long long)src[w]);
  FLD_I(f, F_INT, 0) = (int)(long long)src[7];
  FLD_I(f, F_INT, 1) = (int)(long long)src[8];
}

}  // namespace mcl

using namespace mcl;

// per-context FastSLAM storage
struct FsState {
  int F = 0;
  int64_t M = 0, tiles = 0;
  int cur = 0;
  double* feat[2] = {nullptr, nullptr};   // feature slots, double-buffered for the resampling
  uint8_t* shared[2] = {nullptr, nullptr};  // per tile of feat[b]: its 64 particles share the copy in lane 0's region
  bool maybe_shared = false;              // some tile of the current buffer may be shared (set by a resampling, cleared by fs_make_private)
  int32_t* nfeat[2] = {nullptr, nullptr};
  int32_t* newid[2] = {nullptr, nullptr};
  double* meas = nullptr; int mcap = 0;   // per-step measurements, mcap rows per tile
  double* d_truth = nullptr; int truth_cap = 0;   // (r, phi) of the landmarks in range, padded to a multiple of eight
  unsigned long long* d_sectors = nullptr;        // landmark sets per bearing sector and block of 64, then one set of all per block
  int32_t* overflow = nullptr;
  int32_t* d_maxfeat = nullptr;                   // largest feature count of a particle
  int row_slots = 0;                              // feature slots per exchange row (<= F)
  int32_t max_feat = 0; bool max_feat_valid = false;   // its host copy; valid between the end of a step and the next step's kernels
  std::vector<double> h_truth;                    // host sides of the per-step tables (alive while their uploads are in flight)
  std::vector<unsigned long long> h_sect;
  int16_t* d_ints = nullptr; double* d_normals = nullptr; int64_t normals_cap = 0;
  unsigned long long* stage = nullptr; int64_t stage_words = 0;   // feature rows on their way to / from another rank
  // the step whose weights have been formed and whose updates are applied to particles on request
  uint8_t* state = nullptr;                       // per particle: 0 update not asked for, 1 asked for, 2 applied
  FsParams step_params; int step_rows = 0; int step_entry = 1; size_t step_lds = 0; bool step_open = false;
  double std_r = 0.5, std_phi = 3 * M_PI / 180.0, r_max = 10.0, p0 = 0.01, thres_sig = 2.0;
};

static void fs_free(FsState* s) {
  if (!s) return;
  void* all[] = {s->feat[0], s->feat[1], s->nfeat[0], s->nfeat[1], s->newid[0], s->newid[1], s->meas, s->d_truth, s->d_sectors,
                 s->overflow, s->d_maxfeat, s->d_ints, s->d_normals, s->stage, s->state, s->shared[0], s->shared[1]};
  for (void* q : all) if (q) (void)hipFree(q);
  delete s;
}

void mcl::fs_release(mclhip_ctx* c) {
  fs_free((FsState*)c->fs);
  c->fs = nullptr;
}

static int fs_stage(mclhip_ctx* c, FsState* s, int64_t words) {
  if (words <= s->stage_words) return MCLHIP_OK;
  if (s->stage) { (void)hipStreamSynchronize(c->stream); (void)hipFree(s->stage); s->stage = nullptr; }
  const int64_t cap = words + words / 4;
  MCL_HIP(c, hipMalloc((void**)&s->stage, (size_t)cap * 8));
  s->stage_words = cap;
  return MCLHIP_OK;
}

#define F_TRY(expr) do { int rc__ = (expr); if (rc__ != MCLHIP_OK) return rc__; } while (0)
#define F_HIP(expr) MCL_HIP(c, expr)

// enqueues the count of the largest feature list; s->max_feat holds it once the stream has been synchronised
static int fs_count_features(mclhip_ctx* c, FsState* s, const int32_t* nfeat) {
  F_HIP(hipMemsetAsync(s->d_maxfeat, 0, sizeof(int32_t), c->stream));
  hipLaunchKernelGGL(fs_max_count_kernel, dim3((unsigned)std::min<int64_t>((s->M + 255) / 256, 256)), dim3(256), 0, c->stream,
                     nfeat, s->M, s->d_maxfeat);
  F_HIP(hipGetLastError());
  F_HIP(hipMemcpyAsync(&s->max_feat, s->d_maxfeat, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  s->max_feat_valid = true;
  return MCLHIP_OK;
}

// one launch of the feature sweep of the open step: FS_WEIGH for every particle, FS_APPLY for those marked in s->state
template <typename IT>
static int fs_launch_sweep_typed(mclhip_ctx* c, FsState* s, int mode, const FsDest& dest) {
  const dim3 grid((unsigned)((s->M + 127) / 128));
  auto* const weigh = fs_update_kernel<IT, FS_WEIGH>;
  auto* const apply = fs_update_kernel<IT, FS_APPLY>;
  auto* const copy = fs_update_kernel<IT, FS_COPY>;
  auto* const k = mode == FS_WEIGH ? weigh : (mode == FS_APPLY ? apply : copy);
  if (s->step_lds > 48 * 1024)
    F_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->step_lds));
  hipLaunchKernelGGL(k, grid, dim3(128), s->step_lds, c->stream, s->step_params, s->step_rows, s->state, dest);
  F_HIP(hipGetLastError());
  return MCLHIP_OK;
}
static int fs_launch_sweep(mclhip_ctx* c, FsState* s, int mode, const FsDest& dest = FsDest{nullptr, nullptr, nullptr, nullptr, nullptr}) {
  return s->step_entry == 1 ? fs_launch_sweep_typed<unsigned char>(c, s, mode, dest) : fs_launch_sweep_typed<unsigned short>(c, s, mode, dest);
}

// gives every particle of the current buffer its own copy of its slots (tiles whose particles descend from one ancestor
// keep a single copy after a resampling): needed before anything addresses a particle's own region
static int fs_make_private(mclhip_ctx* c, FsState* s) {
  if (!s->maybe_shared) return MCLHIP_OK;
  const int a = s->cur;
  hipLaunchKernelGGL(fs_unshare_kernel, dim3((unsigned)((s->M + 255) / 256), s->F), dim3(256), 0, c->stream,
                     (const uint8_t*)s->shared[a], s->M, s->F, (const int32_t*)s->nfeat[a], s->feat[a]);
  F_HIP(hipGetLastError());
  F_HIP(hipMemsetAsync(s->shared[a], 0, (size_t)s->tiles, c->stream));
  s->maybe_shared = false;
  return MCLHIP_OK;
}

// applies the open step's updates to the particles of a list (device pointer to n indices) that have not had them yet
static int fs_apply_list(mclhip_ctx* c, FsState* s, const int64_t* d_list, int64_t n) {
  if (!s->step_open || n <= 0) return MCLHIP_OK;
  F_TRY(fs_make_private(c, s));
  hipLaunchKernelGGL(fs_mark_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_list, n, s->M, s->state);
  F_HIP(hipGetLastError());
  return fs_launch_sweep(c, s, FS_APPLY);
}

// uploads the step's inputs and enqueues the kernels that every particle runs (pose sample .. weight)
static int fs_enqueue_kernel(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6], const double* truth_r,
                             const double* truth_phi, const uint8_t* truth_valid, int L, const int16_t* ints,
                             const double* normals, uint64_t seed) {
  if (!c) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  if (!alpha || (L > 0 && (!truth_r || !truth_phi || !truth_valid)) || L < 0) return fail(c, MCLHIP_E_ARG, "alpha / truth arrays NULL");
  FsState* s = (FsState*)c->fs;
  const int64_t M = s->M;
  // the capacity-overflow word belongs to the step that starts here: an earlier step's report is not repeated
  F_HIP(hipMemsetAsync(s->overflow, 0, sizeof(int32_t), c->stream));
  // landmarks in range of the true pose, in landmark order, as (r, phi) pairs: measure() returns
  // r_max + 1 for the others, which fast_slam() skips (:291).  Padded to a multiple of eight with readings
  // no feature is near.
  std::vector<double>& near = s->h_truth;
  near.clear();
  for (int l = 0; l < L; ++l)
    if (truth_valid[l]) { near.push_back(truth_r[l]); near.push_back(truth_phi[l]); }
  const int n_valid = (int)(near.size() / 2);
  while ((near.size() / 2) % 8) { near.push_back(HUGE_VAL); near.push_back(0.0); }
  if (n_valid > s->mcap || (int)(near.size() / 2) > s->truth_cap) {
    F_HIP(hipStreamSynchronize(c->stream));
    if (s->d_truth) (void)hipFree(s->d_truth);
    if (s->meas) (void)hipFree(s->meas);
    if (s->d_sectors) (void)hipFree(s->d_sectors);
    s->d_truth = s->meas = nullptr; s->d_sectors = nullptr; s->truth_cap = s->mcap = 0;
    const int cap = ((n_valid + n_valid / 4 + 8) + 7) & ~7;
    F_HIP(hipMalloc((void**)&s->d_truth, (size_t)2 * cap * sizeof(double)));
    F_HIP(hipMalloc((void**)&s->d_sectors, (size_t)((cap + 63) / 64) * (kSectors + 1) * sizeof(unsigned long long)));
    F_HIP(hipMalloc((void**)&s->meas, (size_t)s->tiles * cap * M_WORDS * 64 * sizeof(double)));
    s->truth_cap = s->mcap = cap;
  }
  const int n_blk = (n_valid + 63) / 64;
  if (!near.empty()) {
    // the association sweep's index: which landmarks of every block of 64 have their bearing in which sector of the circle
    std::vector<unsigned long long>& sect = s->h_sect;
    sect.assign((size_t)n_blk * (kSectors + 1), 0ull);
    for (int u = 0; u < n_valid; ++u) {
      const double phi = near[2 * u + 1];
      const unsigned long long bit = 1ull << (u & 63);
      sect[(size_t)n_blk * kSectors + (u >> 6)] |= bit;
      if (std::fabs(phi) < 1.0e6) {
        const int b = (int)std::floor((phi + M_PI) * ((double)kSectors / (2.0 * M_PI)));   // the kernel's own expression
        sect[(size_t)(u >> 6) * kSectors + (b & (kSectors - 1))] |= bit;
      } else {
        for (int b = 0; b < kSectors; ++b) sect[(size_t)(u >> 6) * kSectors + b] |= bit;    // no bearing to file it under
      }
    }
    F_HIP(hipMemcpyAsync(s->d_truth, near.data(), near.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    F_HIP(hipMemcpyAsync(s->d_sectors, sect.data(), sect.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream));
  }
  if (ints) {
    if (!s->d_ints) F_HIP(hipMalloc((void**)&s->d_ints, (size_t)M * 36 * sizeof(int16_t)));
    F_HIP(hipMemcpyAsync(s->d_ints, ints, (size_t)M * 36 * sizeof(int16_t), hipMemcpyHostToDevice, c->stream));
  }
  if (normals && n_valid > 0) {
    const int64_t need = M * 2 * n_valid;
    if (need > s->normals_cap) {
      if (s->d_normals) (void)hipFree(s->d_normals);
      F_HIP(hipMalloc((void**)&s->d_normals, (size_t)need * sizeof(double)));
      s->normals_cap = need;
    }
    F_HIP(hipMemcpyAsync(s->d_normals, normals, (size_t)need * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  const int a = s->cur;
  // rows of the update sweep's LDS table: the largest feature count any particle holds (left by the end of the
  // previous step; counted now if this is not the call that follows one)
  if (!s->max_feat_valid) {
    F_TRY(fs_count_features(c, s, s->nfeat[a]));
    F_HIP(hipStreamSynchronize(c->stream));
  }
  const int32_t max_feat = s->max_feat;
  s->max_feat_valid = false;                       // births of this step may raise it
  FsParams P;
  P.M = M; P.F = s->F; P.L = n_valid; P.mcap = s->mcap;
  P.v = v; P.w = omega; P.dt = dt;
  P.c1 = (alpha[0] * std::fabs(v) + alpha[1] * std::fabs(omega)) / 6.0;   // b/6.0 (:164, :181-183)
  P.c2 = (alpha[2] * std::fabs(v) + alpha[3] * std::fabs(omega)) / 6.0;
  P.c3 = (alpha[4] * std::fabs(v) + alpha[5] * std::fabs(omega)) / 6.0;
  P.qt_r = std::pow(s->std_r, 2.0); P.qt_phi = std::pow(s->std_phi, 2.0);   // std_r**2, std_phi**2 (:267-268)
  P.std_r = s->std_r; P.std_phi = s->std_phi; P.r_max = s->r_max; P.r_max2 = std::pow(s->r_max, 2.0);
  P.p0 = s->p0; P.gate2 = std::pow(s->thres_sig, 2.0);
  P.ints = ints ? s->d_ints : nullptr;
  P.normals = (normals && n_valid > 0) ? s->d_normals : nullptr;
  P.n_norm = 2 * n_valid;
  P.seed = seed;
  P.index0 = c->first_index;
  P.x = (double*)c->px[c->cur]; P.y = (double*)c->py[c->cur]; P.th = (double*)c->pth[c->cur]; P.wout = c->w[c->cur];
  P.nfeat = s->nfeat[a]; P.new_id = s->newid[a];
  P.feat = s->feat[a]; P.meas = s->meas; P.shared_tile = s->shared[a];
  P.overflow = s->overflow;
  P.sectors = s->d_sectors;
  P.far_r = 0.0;
  for (int u = 0; u < n_valid; ++u) P.far_r = std::max(P.far_r, near[2 * u]);
  const dim3 fs_grid((unsigned)((M + 127) / 128));
  const size_t assoc_tables = (size_t)n_blk * (kSectors + 1) * sizeof(unsigned long long) + near.size() / 2 * sizeof(float2);
  const size_t assoc_q = (size_t)n_valid * 128 * sizeof(uint16_t);
  P.use_q = assoc_tables + assoc_q <= 40 * 1024 ? 1 : 0;      // four blocks per CU keep their columns of quantised noise in LDS
  hipLaunchKernelGGL(fs_pose_kernel, fs_grid, dim3(128), 0, c->stream, P);
  if (n_valid > 0) {
    const size_t assoc_lds = assoc_tables + (P.use_q ? assoc_q : 0);
    if (assoc_lds > 60 * 1024) return fail(c, MCLHIP_E_LIMIT, "more landmarks in range than the association kernel's table holds");
    hipLaunchKernelGGL(fs_associate_kernel, fs_grid, dim3(128), assoc_lds, c->stream, P, (const double2*)s->d_truth);
  }
  {
    // the sweep's inverse table has one row per feature of the particle that has most
    const size_t entry = n_valid < 255 ? 1 : 2;
    const int rows = std::max(max_feat, 1);
    const size_t upd_lds = (size_t)((n_valid + 63) / 64) * 128 * sizeof(unsigned long long) + (size_t)(rows + n_valid) * 128 * entry;
    if (n_valid >= 65535 || upd_lds > 160 * 1024)
      return fail(c, MCLHIP_E_LIMIT, "more features per particle (or landmarks in range) than the update kernel's LDS table holds");
    s->step_params = P; s->step_rows = rows; s->step_entry = (int)entry; s->step_lds = upd_lds;
    F_HIP(hipMemsetAsync(s->state, 0, (size_t)M, c->stream));
    s->step_open = true;
    F_TRY(fs_launch_sweep(c, s, FS_WEIGH));       // every particle's weight; the updates follow for those that are drawn
  }
  F_HIP(hipGetLastError());
  return MCLHIP_OK;
}

extern "C" {

int mclhip_fs_init(mclhip_ctx* c, int64_t M, int capacity) {
  if (!c) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (c->dtype != MCLHIP_F64) return fail(c, MCLHIP_E_ARG, "FastSLAM needs a float64 context");
  if (M <= 0 || capacity <= 0 || capacity > 4096) return fail(c, MCLHIP_E_ARG, "M <= 0 or capacity outside 1..4096");
  fs_release(c);
  std::vector<double> X((size_t)M * 3, 0.0);     // every particle starts at the origin (:104-108)
  F_TRY(mclhip_set_particles(c, X.data(), nullptr, M));
  FsState* s = new FsState();
  c->fs = s;
  s->M = M; s->F = capacity; s->row_slots = capacity;
  s->tiles = (M + 63) / 64;
  const size_t slots = (size_t)s->tiles * capacity * 64;   // feature slots, padded to whole wavefront tiles
  for (int b = 0; b < 2; ++b) {
    F_HIP(hipMalloc((void**)&s->feat[b], slots * F_WORDS * sizeof(double)));
    F_HIP(hipMalloc((void**)&s->nfeat[b], M * sizeof(int32_t)));
    F_HIP(hipMalloc((void**)&s->newid[b], M * sizeof(int32_t)));
    F_HIP(hipMemset(s->nfeat[b], 0, M * sizeof(int32_t)));
    F_HIP(hipMemset(s->newid[b], 0, M * sizeof(int32_t)));
    F_HIP(hipMalloc((void**)&s->shared[b], (size_t)s->tiles));
    F_HIP(hipMemset(s->shared[b], 0, (size_t)s->tiles));
  }
  F_HIP(hipMalloc((void**)&s->overflow, sizeof(int32_t)));
  F_HIP(hipMemset(s->overflow, 0, sizeof(int32_t)));
  F_HIP(hipMalloc((void**)&s->d_maxfeat, sizeof(int32_t)));
  F_HIP(hipMalloc((void**)&s->state, (size_t)M));
  return MCLHIP_OK;
}

int mclhip_fs_set_model(mclhip_ctx* c, double std_r, double std_phi, double r_max, double p0, double thres_sig) {
  if (!c || !c->fs) return c ? fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first") : MCLHIP_E_ARG;
  FsState* s = (FsState*)c->fs;
  s->std_r = std_r; s->std_phi = std_phi; s->r_max = r_max; s->p0 = p0; s->thres_sig = thres_sig;
  return MCLHIP_OK;
}

int mclhip_fs_step(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6], const double* truth_r,
                   const double* truth_phi, const uint8_t* truth_valid, int L, const int16_t* ints,
                   const double* normals, uint64_t seed, double r, int64_t* idx_out) {
  F_TRY(fs_enqueue_kernel(c, v, omega, dt, alpha, truth_r, truth_phi, truth_valid, L, ints, normals, seed));
  FsState* s = (FsState*)c->fs;
  const int64_t M = s->M;
  const int b = 1 - s->cur;
  // normalise by the Python-order sum with its 1e-8 floor, then the guarded low-variance walk (:411-431)
  F_TRY(launch_sequential_total(c));                     // c->cdf_exact[M-1] = sum_w, added one by one
  const unsigned g = (unsigned)((M + 255) / 256);
  hipLaunchKernelGGL(fs_total_kernel, dim3(1), dim3(64), 0, c->stream, c->cdf_exact, M, 1e-8, c->scal);
  hipLaunchKernelGGL(fs_divide_kernel, dim3(g), dim3(256), 0, c->stream, c->w[c->cur], M, c->scal);
  F_TRY(launch_resample(c, r, MCLHIP_RESAMPLE_PARALLEL)); // flips c->cur; c->idx = ancestors
  // updates, births and decay of the drawn ancestors, written straight into their descendants' places: one sweep per
  // position, or one per tile where the whole tile has one ancestor
  F_TRY(fs_launch_sweep(c, s, FS_COPY, FsDest{c->idx, s->feat[b], s->nfeat[b], s->newid[b], s->shared[b]}));
  s->step_open = false;
  s->maybe_shared = true;
  s->cur = b;
  F_TRY(fs_count_features(c, s, s->nfeat[b]));
  int32_t ov = 0;
  F_HIP(hipMemcpyAsync(&ov, s->overflow, sizeof(ov), hipMemcpyDeviceToHost, c->stream));
  F_HIP(hipStreamSynchronize(c->stream));
  if (idx_out) F_HIP(hipMemcpy(idx_out, c->idx, M * sizeof(int64_t), hipMemcpyDeviceToHost));
  if (ov) return fail(c, MCLHIP_E_LIMIT, "a particle ran out of feature slots: raise the capacity given to mclhip_fs_init");
  return MCLHIP_OK;
}

int mclhip_fs_get_state(mclhip_ctx* c, double* pose, double* w, int32_t* nfeat, int32_t* new_id, double* f_m,
                        double* f_cov, int32_t* f_i, double* f_w, int32_t* f_id) {
  if (!c) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  FsState* s = (FsState*)c->fs;
  const int64_t M = s->M;
  const int F = s->F, a = s->cur;
  F_TRY(fs_make_private(c, s));
  if (s->step_open) {                            // between the per-particle kernels and the exchange: every particle's update
    hipLaunchKernelGGL(fs_mark_range_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, c->stream, (int64_t)0, M, s->state);
    F_HIP(hipGetLastError());
    F_TRY(fs_launch_sweep(c, s, FS_APPLY));
  }
  F_HIP(hipStreamSynchronize(c->stream));
  if (pose || w) F_TRY(mclhip_get_particles(c, pose, w, nullptr));
  std::vector<int32_t> nf(M);
  F_HIP(hipMemcpy(nf.data(), s->nfeat[a], M * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (nfeat) std::copy(nf.begin(), nf.end(), nfeat);
  if (new_id) F_HIP(hipMemcpy(new_id, s->newid[a], M * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (f_m || f_cov || f_w || f_i || f_id) {
    const size_t words = (size_t)s->tiles * F * F_WORDS * 64;
    std::vector<double> tmp(words);
    F_HIP(hipMemcpy(tmp.data(), s->feat[a], words * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t p = 0; p < M; ++p)
      for (int j = 0; j < F; ++j) {
        const bool live = j < nf[p];
        const double* f = tmp.data() + tile_word(p, F, j, F_WORDS);
        const size_t o = (size_t)p * F + j;
        int32_t halves[2];
        std::memcpy(halves, &FLD(f, F_INT), sizeof(halves));
        if (f_m) { f_m[o * 2] = live ? FLD(f, F_M0) : 0.0; f_m[o * 2 + 1] = live ? FLD(f, F_M1) : 0.0; }
        if (f_cov) for (int q = 0; q < 4; ++q) f_cov[o * 4 + q] = live ? FLD(f, F_C00 + q) : 0.0;
        if (f_w) f_w[o] = live ? FLD(f, F_W) : 0.0;
        if (f_i) f_i[o] = live ? halves[0] : 0;
        if (f_id) f_id[o] = live ? halves[1] : -1;
      }
  }
  return MCLHIP_OK;
}

/* ---- one FastSLAM filter over several GPUs: the per-rank halves ---- */

int mclhip_fs_shard_kernel(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6], const double* truth_r,
                           const double* truth_phi, const uint8_t* truth_valid, int L, const int16_t* ints,
                           const double* normals, uint64_t seed) {
  F_TRY(fs_enqueue_kernel(c, v, omega, dt, alpha, truth_r, truth_phi, truth_valid, L, ints, normals, seed));
  FsState* s = (FsState*)c->fs;
  int32_t ov = 0;
  F_HIP(hipMemcpyAsync(&ov, s->overflow, sizeof(ov), hipMemcpyDeviceToHost, c->stream));
  F_HIP(hipStreamSynchronize(c->stream));
  if (ov) return fail(c, MCLHIP_E_LIMIT, "a particle ran out of feature slots: raise the capacity given to mclhip_fs_init");
  return MCLHIP_OK;
}

int mclhip_fs_shard_last_sum(mclhip_ctx* c, double* last_out) {
  if (!c || !last_out) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (c->M <= 0) return fail(c, MCLHIP_E_STATE, "no particles");
  F_HIP(hipStreamSynchronize(c->stream));
  F_HIP(hipMemcpy(last_out, c->cdf_exact + (c->M - 1), sizeof(double), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

int64_t mclhip_fs_row_words(mclhip_ctx* c) {
  if (!c || !c->fs) return 0;
  return 2 + (int64_t)9 * ((FsState*)c->fs)->row_slots;
}

int mclhip_fs_set_row_slots(mclhip_ctx* c, int slots) {
  if (!c) return MCLHIP_E_ARG;
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  FsState* s = (FsState*)c->fs;
  if (slots < 1 || slots > s->F) return fail(c, MCLHIP_E_ARG, "row slots outside 1..capacity");
  s->row_slots = slots;
  return MCLHIP_OK;
}

int mclhip_fs_max_features(mclhip_ctx* c, int32_t* out) {
  if (!c || !out) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  FsState* s = (FsState*)c->fs;
  if (!s->max_feat_valid) F_TRY(fs_count_features(c, s, s->nfeat[s->cur]));
  F_HIP(hipStreamSynchronize(c->stream));
  *out = s->max_feat;
  return MCLHIP_OK;
}

int mclhip_fs_shard_pack(mclhip_ctx* c, const int64_t* lo, const int64_t* hi, int nslices, void* rows) {
  if (!c) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  if (!lo || !hi || nslices <= 0) return fail(c, MCLHIP_E_ARG, "lo/hi NULL or no slices");
  FsState* s = (FsState*)c->fs;
  const int64_t words = 2 + (int64_t)9 * s->row_slots;
  int64_t total = 0;
  for (int k = 0; k < nslices; ++k) {
    if (lo[k] < 0 || hi[k] > s->M || hi[k] < lo[k]) return fail(c, MCLHIP_E_ARG, "slice outside the local range");
    total += hi[k] - lo[k];
  }
  if (total == 0) return MCLHIP_OK;
  if (!rows) return fail(c, MCLHIP_E_ARG, "rows is NULL");
  F_TRY(fs_stage(c, s, total * words));
  F_TRY(fs_make_private(c, s));
  if (s->step_open) {                           // rows travel with their step's updates applied
    for (int k = 0; k < nslices; ++k)
      if (hi[k] > lo[k])
        hipLaunchKernelGGL(fs_mark_range_kernel, dim3((unsigned)((hi[k] - lo[k] + 255) / 256)), dim3(256), 0, c->stream, lo[k], hi[k], s->state);
    F_HIP(hipGetLastError());
    F_TRY(fs_launch_sweep(c, s, FS_APPLY));
  }
  int64_t off = 0;
  for (int k = 0; k < nslices; ++k) {
    const int64_t n = hi[k] - lo[k];
    if (n == 0) continue;
    hipLaunchKernelGGL(fs_pack_rows_kernel, dim3((unsigned)((n + 63) / 64), s->row_slots + 1), dim3(64), 0, c->stream, lo[k], n,
                       s->F, s->row_slots, s->nfeat[s->cur], s->newid[s->cur], (const double*)s->feat[s->cur],
                       s->stage + off * words, s->overflow);
    off += n;
  }
  F_HIP(hipGetLastError());
  F_HIP(hipMemcpyAsync(rows, s->stage, (size_t)total * words * 8, hipMemcpyDefault, c->stream));
  int32_t ov = 0;
  F_HIP(hipMemcpyAsync(&ov, s->overflow, sizeof(ov), hipMemcpyDeviceToHost, c->stream));
  F_HIP(hipStreamSynchronize(c->stream));
  if (ov >= 2) {
    F_HIP(hipMemset(s->overflow, 0, sizeof(int32_t)));
    return fail(c, MCLHIP_E_LIMIT, "a particle holds more features than an exchange row has slots (mclhip_fs_set_row_slots)");
  }
  return MCLHIP_OK;
}

int mclhip_fs_shard_apply_drawn(mclhip_ctx* c, int32_t* max_out) {
  if (!c || !max_out) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  FsState* s = (FsState*)c->fs;
  const int64_t total = c->drawn_total;
  *max_out = 0;
  if (total == 0) return MCLHIP_OK;
  if (!c->sel || c->sel_cap < total) return fail(c, MCLHIP_E_STATE, "call mclhip_shard_pack_drawn first (it selects the rows)");
  F_TRY(fs_apply_list(c, s, (const int64_t*)c->sel, total));
  F_HIP(hipMemsetAsync(s->d_maxfeat, 0, sizeof(int32_t), c->stream));
  hipLaunchKernelGGL(fs_max_sel_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 256)), dim3(256), 0, c->stream,
                     (const int64_t*)c->sel, total, s->nfeat[s->cur], s->d_maxfeat);
  F_HIP(hipGetLastError());
  F_HIP(hipMemcpyAsync(max_out, s->d_maxfeat, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  F_HIP(hipStreamSynchronize(c->stream));
  s->max_feat_valid = false;                     // d_maxfeat no longer holds the count of the whole set
  return MCLHIP_OK;
}

int mclhip_fs_shard_pack_drawn(mclhip_ctx* c, void* rows) {
  if (!c) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  FsState* s = (FsState*)c->fs;
  const int64_t words = 2 + (int64_t)9 * s->row_slots;
  const int64_t total = c->drawn_total;
  if (total == 0) return MCLHIP_OK;
  if (!rows) return fail(c, MCLHIP_E_ARG, "rows is NULL");
  if (!c->sel || c->sel_cap < total) return fail(c, MCLHIP_E_STATE, "call mclhip_shard_pack_drawn first (it selects the rows)");
  F_TRY(fs_stage(c, s, total * words));
  F_TRY(fs_make_private(c, s));
  F_TRY(fs_apply_list(c, s, (const int64_t*)c->sel, total));
  hipLaunchKernelGGL(fs_pack_sel_kernel, dim3((unsigned)((total + 63) / 64), s->row_slots + 1), dim3(64), 0, c->stream,
                     (const int64_t*)c->sel, total, s->F, s->row_slots, s->nfeat[s->cur], s->newid[s->cur],
                     (const double*)s->feat[s->cur], s->stage, s->overflow);
  F_HIP(hipGetLastError());
  F_HIP(hipMemcpyAsync(rows, s->stage, (size_t)total * words * 8, hipMemcpyDefault, c->stream));
  int32_t ov = 0;
  F_HIP(hipMemcpyAsync(&ov, s->overflow, sizeof(ov), hipMemcpyDeviceToHost, c->stream));
  F_HIP(hipStreamSynchronize(c->stream));
  if (ov >= 2) {
    F_HIP(hipMemset(s->overflow, 0, sizeof(int32_t)));
    return fail(c, MCLHIP_E_LIMIT, "a particle holds more features than an exchange row has slots (mclhip_fs_set_row_slots)");
  }
  return MCLHIP_OK;
}

}  // extern "C"

/* ---- the same over several GPUs without the host in the loop: every call below only enqueues ---- */

// marks the first n = counts[0] + ... + counts[world - 1] particles of a list, n on the device
__global__ void fs_mark_counted_kernel(const int64_t* __restrict__ list, const int64_t* __restrict__ counts, int world, int64_t M,
                                       uint8_t* __restrict__ state) {
  int64_t n = 0;
  for (int d = 0; d < world; ++d) n += counts[d];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t p = list[i];
  if (p >= 0 && p < M && state[p] == 0) state[p] = 1;
}

// info[0] = largest feature count among the listed particles, info[1] = a particle ran out of slots,
// info[2] = the first record exchange overflowed
__global__ void fs_select_info_kernel(const int64_t* __restrict__ list, const int64_t* __restrict__ counts, int world,
                                      const int32_t* __restrict__ nfeat, const int32_t* __restrict__ overflow,
                                      const double* __restrict__ first_round, int64_t* __restrict__ info) {
  __shared__ int s_max[256];
  int64_t n = 0;
  for (int d = 0; d < world; ++d) n += counts[d];
  int v = 0;
  for (int64_t j = threadIdx.x; j < n; j += blockDim.x) v = max(v, nfeat[list[j]]);
  s_max[threadIdx.x] = v;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) s_max[threadIdx.x] = max(s_max[threadIdx.x], s_max[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) { info[0] = s_max[0]; info[1] = *overflow ? 1 : 0; info[2] = first_round[1] != 0.0 ? 1 : 0; }
}

extern "C" {

int mclhip_fs_dshard_kernel(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6], const double* truth_r,
                            const double* truth_phi, const uint8_t* truth_valid, int L, const int16_t* ints,
                            const double* normals, uint64_t seed, double* d_chunks_out) {
  if (!c || !d_chunks_out) return MCLHIP_E_ARG;
  F_TRY(fs_enqueue_kernel(c, v, omega, dt, alpha, truth_r, truth_phi, truth_valid, L, ints, normals, seed));
  F_TRY(launch_shard_weights(c, MCLHIP_W_PRODUCT, 0.0));      // chunk sums of the raw weights: they only place the scans
  const int64_t nch = (c->M + kSumChunk - 1) / kSumChunk;
  F_HIP(hipMemcpyAsync(d_chunks_out, c->chunk_sums, nch * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  return MCLHIP_OK;
}

int mclhip_fs_dshard_select(mclhip_ctx* c, int world, const int64_t* d_counts, const double* d_first_round, void* d_rows,
                            int64_t* d_info_out) {
  if (!c) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  if (!d_counts || !d_first_round || !d_rows || !d_info_out || world < 1 || world > 64)
    return fail(c, MCLHIP_E_ARG, "device pointer NULL or world outside 1..64");
  FsState* s = (FsState*)c->fs;
  const int64_t most = s->M + world;             // every local row once, the rows at the rank boundaries twice
  if (c->sel_cap < most) {
    F_HIP(hipStreamSynchronize(c->stream));
    if (c->sel) (void)hipFree(c->sel);
    c->sel = nullptr; c->sel_cap = 0;
    F_HIP(hipMalloc((void**)&c->sel, (size_t)most * sizeof(int64_t)));
    c->sel_cap = most;
  }
  F_TRY(launch_pack_drawn(c, world, (double*)d_rows, (int64_t*)c->sel));
  if (s->step_open) {
    F_TRY(fs_make_private(c, s));
    hipLaunchKernelGGL(fs_mark_counted_kernel, dim3((unsigned)((most + 255) / 256)), dim3(256), 0, c->stream,
                       (const int64_t*)c->sel, d_counts, world, s->M, s->state);
    F_HIP(hipGetLastError());
    F_TRY(fs_launch_sweep(c, s, FS_APPLY));
  }
  hipLaunchKernelGGL(fs_select_info_kernel, dim3(1), dim3(256), 0, c->stream, (const int64_t*)c->sel, d_counts, world,
                     (const int32_t*)s->nfeat[s->cur], (const int32_t*)s->overflow, d_first_round, d_info_out);
  F_HIP(hipGetLastError());
  return MCLHIP_OK;
}

int mclhip_fs_dshard_pack(mclhip_ctx* c, int64_t n_rows, void* d_feature_rows) {
  if (!c) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  FsState* s = (FsState*)c->fs;
  if (n_rows == 0) return MCLHIP_OK;
  if (!d_feature_rows || n_rows < 0 || n_rows > c->sel_cap) return fail(c, MCLHIP_E_ARG, "rows NULL or more rows than were selected");
  hipLaunchKernelGGL(fs_pack_sel_kernel, dim3((unsigned)((n_rows + 63) / 64), s->row_slots + 1), dim3(64), 0, c->stream,
                     (const int64_t*)c->sel, n_rows, s->F, s->row_slots, s->nfeat[s->cur], s->newid[s->cur],
                     (const double*)s->feat[s->cur], (unsigned long long*)d_feature_rows, s->overflow);
  F_HIP(hipGetLastError());
  return MCLHIP_OK;
}

int mclhip_fs_dshard_unpack(mclhip_ctx* c, const void* d_rows, const void* d_feature_rows, int64_t L, double r, int max_features) {
  if (!c) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  if (!d_rows || !d_feature_rows || L <= 0) return fail(c, MCLHIP_E_ARG, "rows NULL or empty");
  FsState* s = (FsState*)c->fs;
  F_TRY(launch_unpack_drawn(c, (const double*)d_rows, L, r, true));   // poses and weights; the row every position took -> c->kscan
  const int b = 1 - s->cur;
  F_HIP(hipMemsetAsync(s->shared[b], 0, (size_t)s->tiles, c->stream));
  hipLaunchKernelGGL(fs_unpack_rows_kernel, dim3((unsigned)((s->M + 63) / 64), s->row_slots + 1), dim3(64), 0, c->stream,
                     reinterpret_cast<const int64_t*>(c->kscan), (int64_t)0, L, s->M, s->F, s->row_slots,
                     (const unsigned long long*)d_feature_rows, s->nfeat[b], s->newid[b], s->feat[b]);
  F_HIP(hipGetLastError());
  s->cur = b;
  s->step_open = false;
  s->maybe_shared = false;
  // no particle of any rank holds more features than the longest list that travelled: the next step's tables are sized by it
  s->max_feat = max_features;
  s->max_feat_valid = true;
  return MCLHIP_OK;
}

}  // extern "C"

extern "C" {

int mclhip_fs_shard_unpack(mclhip_ctx* c, const void* rows, int64_t L) {
  if (!c) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  if (!rows || L <= 0) return fail(c, MCLHIP_E_ARG, "rows NULL or empty");
  FsState* s = (FsState*)c->fs;
  const int64_t words = 2 + (int64_t)9 * s->row_slots;
  F_TRY(fs_stage(c, s, L * words));
  F_HIP(hipMemcpyAsync(s->stage, rows, (size_t)L * words * 8, hipMemcpyDefault, c->stream));
  const int b = 1 - s->cur;
  F_HIP(hipMemsetAsync(s->shared[b], 0, (size_t)s->tiles, c->stream));   // every particle gets its own slots from its row
  // the received row every local position took was left in c->kscan by mclhip_shard_unpack_drawn
  hipLaunchKernelGGL(fs_unpack_rows_kernel, dim3((unsigned)((s->M + 63) / 64), s->row_slots + 1), dim3(64), 0, c->stream,
                     reinterpret_cast<const int64_t*>(c->kscan), (int64_t)0, L, s->M, s->F, s->row_slots,
                     (const unsigned long long*)s->stage, s->nfeat[b], s->newid[b], s->feat[b]);
  F_HIP(hipGetLastError());
  F_TRY(fs_count_features(c, s, s->nfeat[b]));
  F_HIP(hipStreamSynchronize(c->stream));
  s->cur = b;
  s->step_open = false;
  s->maybe_shared = false;
  return MCLHIP_OK;
}

}  // extern "C"
