// FastSLAM 1.0 with unknown correspondence -- the reference's 13_fast_slam/fast_slam1.py
// (fast_slam(), :240-433) on the GPU.
//
// One thread per particle, three kernels per filter step: pose sample (Table 5.3 with the twelve-integer
// normal sampler, :160-189) and per-feature measurement prediction / Jacobian / innovation covariance
// (:259-282); measurement draws, maximum-likelihood association of every sensed landmark with a 2-sigma
// Mahalanobis gate (:286-315) and feature birth (:318-347) or EKF update (:348-375); the weight /
// feature-decay pass that removes from the list it iterates (:377-404).  Resampling reuses the
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
#include <cmath>
#include <cstring>
#include <vector>

#include "device_rng.h"
#include "mclhip_internal.h"

namespace mcl {

// fields of a feature slot (64-bit words; the last holds the two 32-bit counters)
enum { F_M0 = 0, F_M1, F_C00, F_C01, F_C10, F_C11, F_W, F_INT, F_WORDS };
// per-step prediction of a slot: z_pred, H, inv Q, det Q, "seen this step"
enum { S_ZR = 0, S_ZPHI, S_H00, S_H01, S_H10, S_H11, S_Q00, S_Q01, S_Q10, S_Q11, S_DET, S_UPD, S_WORDS };
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
  double* scr;              // tiles x F x S_WORDS x 64
  double* meas;             // tiles x mcap x M_WORDS x 64
  double* noise;            // [2][M]: largest range / bearing noise a particle drew this step
  int32_t* overflow;
};

// this lane's word 0 of a record.  Fields are stored in pairs -- (f0, f1) of lane 0, (f0, f1) of lane 1, ...,
// then (f2, f3) of lane 0, ... -- so that neighbouring fields travel as one 16-byte access per lane.
__device__ __host__ inline int64_t tile_word(int64_t p, int64_t rows, int64_t row, int words) {
  return ((((p >> 6) * rows + row) * words) << 6) + ((p & 63) << 1);
}
#define FREC(slot) (P.feat + tile_word(p, P.F, (slot), F_WORDS))
#define SREC(slot) (P.scr + tile_word(p, P.F, (slot), S_WORDS))
#define MREC(u) (P.meas + tile_word(p, P.mcap, (u), M_WORDS))
#define FLD(rec, f) (rec)[((f) >> 1) * 128 + ((f) & 1)]
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
  const double l = r0 / p0;
  const double u11 = r1 - l * p1;
  if (det) *det = swap ? -(p0 * u11) : (p0 * u11);
  // solve for the two columns of the (row-permuted) identity
  const double b0a = swap ? 0.0 : 1.0, b1a = swap ? 1.0 : 0.0;   // column 0
  const double b0b = swap ? 1.0 : 0.0, b1b = swap ? 0.0 : 1.0;   // column 1
  const double y1a = b1a - l * b0a, y1b = b1b - l * b0b;
  const double x1a = y1a / u11, x1b = y1b / u11;
  const double x0a = (b0a - p1 * x1a) / p0, x0b = (b0b - p1 * x1b) / p0;
  return M2{x0a, x0b, x1a, x1b};
}

__device__ inline void predict_measurement(double m0, double m1, double x, double y, double th, double* r, double* phi,
                                           M2* H) {
  const double dx = m0 - x, dy = m1 - y;
  const double q = dx * dx + dy * dy;
  const double rr = sqrt(q);
  double ph = atan2(dy, dx) - th;             // :213
  if (ph > M_PI) ph -= M_PI * 2;
  else if (ph < -M_PI) ph += M_PI * 2;
  *r = rr; *phi = ph;
  *H = M2{dx / rr, dy / rr, -dy / q, dx / q}; // :233-236
}

__device__ inline double wrap_residual(double d) {
  if (d > M_PI) d -= M_PI * 2;                // :301-304
  else if (d < -M_PI) d += M_PI * 2;
  return d;
}

// two independent standard normals from the counter-based generator (both Box-Muller outputs of two
// 32-bit words)
__device__ inline void device_normal_pair(uint64_t particle, uint32_t draw, uint64_t seed, double* a, double* b) {
  uint32_t ctr[4] = {(uint32_t)particle, (uint32_t)(particle >> 32), draw, 0x4653u};
  philox4x32(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  const double u1 = ((double)ctr[0] + 1.0) * (1.0 / 4294967296.0);   // (0, 1]
  const double u2 = (double)ctr[1] * (1.0 / 4294967296.0);
  const double rad = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincos(2.0 * M_PI * u2, &sn, &cs);
  *a = rad * cs; *b = rad * sn;
}

#ifndef FS_WAVES
#define FS_WAVES 3
#endif
#ifndef FS_BATCH
#define FS_BATCH 6
#endif

// ---- pose sample (:160-189) and per-feature prediction (:259-282) ----
__global__ __launch_bounds__(128) void fs_predict_kernel(const FsParams P) {
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
  const double x = x0 - q * sin(th0) + q * sin(th0 + w_hat * P.dt);
  const double y = y0 + q * cos(th0) - q * cos(th0 + w_hat * P.dt);
  const double th = th0 + w_hat * P.dt + g_hat * P.dt;
  P.x[p] = x; P.y[p] = y; P.th[p] = th;

  const int n_old = P.nfeat[p];
  for (int j = 0; j < n_old; ++j) {
    const double* f = FREC(j);
    double* sc = SREC(j);
    double r, phi;
    M2 H;
    predict_measurement(FLD(f, F_M0), FLD(f, F_M1), x, y, th, &r, &phi, &H);
    const M2 C{FLD(f, F_C00), FLD(f, F_C01), FLD(f, F_C10), FLD(f, F_C11)};
    M2 Q = mul(mul(H, C), tr(H));              // :276
    Q.a = Q.a + P.qt_r; Q.b = Q.b + 0.0; Q.c = Q.c + 0.0; Q.d = Q.d + P.qt_phi;
    double det;
    const M2 iq = inv_lu(Q, &det);
    FLD(sc, S_ZR) = r; FLD(sc, S_ZPHI) = phi;
    FLD(sc, S_H00) = H.a; FLD(sc, S_H01) = H.b; FLD(sc, S_H10) = H.c; FLD(sc, S_H11) = H.d;
    FLD(sc, S_Q00) = iq.a; FLD(sc, S_Q01) = iq.b; FLD(sc, S_Q10) = iq.c; FLD(sc, S_Q11) = iq.d;
    FLD(sc, S_DET) = det;
    FLD_I(sc, S_UPD, 0) = 0;
  }
}

// The out-of-line helpers below see plain values and this lane's record pointers only -- nothing they
// would have to read again after a store -- and their constants stay out of the registers of the loops.

// the likelihood of a pair that passed the gate (:308-309)
__device__ __attribute__((noinline)) double gate_weight(double d2, double dq) {
  const double eta = 1.0 / (2.0 * M_PI * sqrt(dq));
  return eta * exp(-0.5 * d2);
}

// new feature (:318-347)
__device__ __attribute__((noinline)) void create_feature(double* f, double* sc, int32_t id, double zr, double zphi, double x,
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
  FLD_I(sc, S_UPD, 0) = 1;
}

// ---- every landmark: draw the measurement, associate, create or update (:286-375) ----
// The association of a measurement reads only what the prediction pass stored (z_pred, invQ, det Q of the
// features that existed before this step), never what an earlier measurement of the same step changed.  So
// all measurements of the step are drawn first and parked in device memory (fs_draw_kernel), ONE sweep
// over the stored features associates them all (fs_associate_kernel), and then they are applied
// (fs_apply_kernel): updates in landmark order -- two measurements may pick the same feature -- and
// births in landmark order; an update touches a feature that existed before the step, a birth a new
// slot, so the two do not meet.
//
// A wavefront's time goes into waiting for memory, so each kernel is arranged to have few dependent round
// trips: loads a step ahead of their use, stores that nothing waits for, out-of-line calls (which drain
// the memory queue) only where nothing is in flight.

// one noisy measurement per landmark in range (:145-152) and the largest noise of this particle
__global__ __launch_bounds__(128) void fs_draw_kernel(const FsParams P, const double2* __restrict__ truth) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.M) return;
  const int nv = P.L;
  const double* normals = P.normals ? P.normals + p * P.n_norm : nullptr;
  double mr = 0.0, mphi = 0.0;
  for (int u = 0; u < nv; ++u) {
    const double2 t = truth[u];
    double n_r, n_phi;
    if (normals) {
      n_r = normals[2 * u];
      n_phi = normals[2 * u + 1];
    } else {
      device_normal_pair((uint64_t)(P.index0 + p), (uint32_t)u, P.seed, &n_r, &n_phi);
    }
    const double zr = t.x + n_r * P.std_r;
    double zphi = t.y + n_phi * P.std_phi;
    if (zphi > M_PI) zphi -= 2 * M_PI;
    else if (zphi < -M_PI) zphi += 2 * M_PI;
    mr = fmax(mr, fabs(n_r * P.std_r)); mphi = fmax(mphi, fabs(n_phi * P.std_phi));
    const bool in_range = zr <= P.r_max;        // :291
    double* m = MREC(u);
    FLD(m, M_ZR) = in_range ? zr : __longlong_as_double(0x7ff8000000000000ll);   // NaN never passes a gate
    FLD(m, M_ZPHI) = zphi;
    FLD(m, M_BW) = 0.0;
    FLD_I(m, M_BC, 0) = in_range ? -1 : -2;
  }
  P.noise[p] = mr;
  P.noise[P.M + p] = mphi;
}

// The sweep looks at a (feature, measurement) pair closely only if it can pass the gate at all.  For a
// positive definite Q, r' invQ r >= r_k^2 / Q_kk: each residual alone must be within the gate of its own
// variance (Q_00 = invQ_11 det Q, Q_11 = invQ_00 det Q).  A measurement differs from the landmark's
// noise-free reading -- the same for every particle, so it sits in scalar registers -- by at most the
// largest noise this particle drew, and the angular residual is at least the distance on the circle.  The
// test uses twice the gate, far beyond any rounding of the quantities involved; neighbouring particles
// share ancestors and hence feature order, so whole wavefronts skip nearly all pairs.  Eight landmarks are
// tested per round without a branch; the candidates of a feature are then read together.
struct Predicted { double fr, fphi, q00, q01, q10, q11, dq; };
__device__ inline Predicted load_predicted(const double* sc) {
  return Predicted{FLD(sc, S_ZR), FLD(sc, S_ZPHI), FLD(sc, S_Q00), FLD(sc, S_Q01), FLD(sc, S_Q10), FLD(sc, S_Q11), FLD(sc, S_DET)};
}

__global__ __launch_bounds__(128, FS_WAVES) void fs_associate_kernel(const FsParams P, const double2* __restrict__ truth) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.M) return;
  const int nv = P.L;
  const int n_old = P.nfeat[p];
  if (n_old == 0) return;
  const double mr = P.noise[p], mphi = P.noise[P.M + p];
  const double two_gate = 2.0 * P.gate2;
  constexpr int kBatch = FS_BATCH;
  const int nv8 = (nv + 7) & ~7;                // the table is padded to a multiple of eight with far-away readings
  Predicted next = load_predicted(SREC(0));
  for (int j = 0; j < n_old; ++j) {
    const Predicted f = next;
    if (j + 1 < n_old) next = load_predicted(SREC(j + 1));   // in flight during this feature's tests
    const double lim0 = two_gate * (f.q11 * f.dq), lim1 = two_gate * (f.q00 * f.dq);
    unsigned long long cand = 0;                // measurements that may pass this feature's gate
    for (int u0 = 0; u0 < nv8; u0 += 8) {
      unsigned pass = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double2 t = truth[u0 + k];
        const double g0 = fmax(fabs(t.x - f.fr) - mr, 0.0);
        const double d = t.y - f.fphi;
        const double g1 = fmax(fabs(d - (2.0 * M_PI) * rint(d * (0.5 / M_PI))) - mphi, 0.0);
        const bool out = (g0 * g0 > lim0) || (g1 * g1 > lim1);
        pass |= (out || u0 + k >= nv) ? 0u : (1u << k);
      }
      cand |= (unsigned long long)pass << (u0 & 63);
      if ((u0 & 63) == 56 || u0 + 8 >= nv8) {   // 64 measurements tested: look at the candidates
        const int base = u0 & ~63;
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
                const double w = gate_weight(d2, f.dq);
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
}

// what an EKF update reads of its feature
struct SlotData { M2 H, iq, C; double zr, zphi, m0, m1, det; int cnt, id; };
__device__ inline SlotData load_slot(const double* f, const double* sc) {
  SlotData d;
  d.H = M2{FLD(sc, S_H00), FLD(sc, S_H01), FLD(sc, S_H10), FLD(sc, S_H11)};
  d.iq = M2{FLD(sc, S_Q00), FLD(sc, S_Q01), FLD(sc, S_Q10), FLD(sc, S_Q11)};
  d.C = M2{FLD(f, F_C00), FLD(f, F_C01), FLD(f, F_C10), FLD(f, F_C11)};
  d.zr = FLD(sc, S_ZR); d.zphi = FLD(sc, S_ZPHI);
  d.m0 = FLD(f, F_M0); d.m1 = FLD(f, F_M1);
  d.cnt = FLD_I(f, F_INT, 0); d.id = FLD_I(f, F_INT, 1);
  d.det = FLD(sc, S_DET);
  return d;
}
struct MeasRec { int c; double zr, zphi, bw; };
__device__ inline MeasRec load_meas(const double* m) {
  return MeasRec{FLD_I(m, M_BC, 0), FLD(m, M_ZR), FLD(m, M_ZPHI), FLD(m, M_BW)};
}

// EKF update with the prediction-pass H, invQ, z_pred (:348-375)
__device__ inline void update_feature(double* f, double* sc, const SlotData& d, double zr, double zphi, double w_max) {
  const M2 K = mul(mul(d.C, tr(d.H)), d.iq);   // :360
  const double e0 = zr - d.zr;
  const double e1 = wrap_residual(zphi - d.zphi);
  FLD(f, F_M0) = d.m0 + (K.a * e0 + K.b * e1);   // :369
  FLD(f, F_M1) = d.m1 + (K.c * e0 + K.d * e1);
  const M2 KH = mul(K, d.H);
  const M2 IKH{1.0 - KH.a, 0.0 - KH.b, 0.0 - KH.c, 1.0 - KH.d};
  const M2 Cn = mul(IKH, d.C);                 // :370
  FLD(f, F_C00) = Cn.a; FLD(f, F_C01) = Cn.b; FLD(f, F_C10) = Cn.c; FLD(f, F_C11) = Cn.d;
  // whole 16-byte pairs are written (the untouched halves with what was read): no partial lines
  FLD(f, F_W) = w_max;
  FLD_I(f, F_INT, 0) = d.cnt + 1; FLD_I(f, F_INT, 1) = d.id;
  FLD(sc, S_DET) = d.det;
  FLD_I(sc, S_UPD, 0) = 1; FLD_I(sc, S_UPD, 1) = 0;
}

#ifndef FS_APPLY_WAVES
#define FS_APPLY_WAVES 3
#endif
__global__ __launch_bounds__(128, FS_APPLY_WAVES) void fs_apply_kernel(const FsParams P) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.M) return;
  const int nv = P.L;
  const double x = P.x[p], y = P.y[p], th = P.th[p];
  int n = P.nfeat[p];
  int32_t new_id = P.new_id[p];
  for (int base = 0; base < nv; base += 64) {
    const int top = min(nv, base + 64);
    // updates: the measurement two steps ahead and the feature of the next step are on their way while
    // this step computes; a feature picked twice in a row is read again after its stores
    unsigned long long births = 0;
    MeasRec m1 = load_meas(MREC(base));
    MeasRec m2 = base + 1 < top ? load_meas(MREC(base + 1)) : MeasRec{-2, 0.0, 0.0, 0.0};
    SlotData s1 = {};
    if (m1.c >= 0) s1 = load_slot(FREC(m1.c), SREC(m1.c));
    for (int u = base; u < top; ++u) {
      const MeasRec cur = m1;
      const SlotData cs = s1;
      m1 = m2;
      m2 = u + 2 < top ? load_meas(MREC(u + 2)) : MeasRec{-2, 0.0, 0.0, 0.0};
      const bool again = m1.c == cur.c;
      if (m1.c >= 0 && !again) s1 = load_slot(FREC(m1.c), SREC(m1.c));
      if (cur.c >= 0) update_feature(FREC(cur.c), SREC(cur.c), cs, cur.zr, cur.zphi, cur.bw);
      if (cur.c == -1) births |= 1ull << (u - base);
      if (m1.c >= 0 && again) s1 = load_slot(FREC(m1.c), SREC(m1.c));
    }
    // births (:318-347), in landmark order
    while (births) {
      const int u = base + __ffsll((long long)births) - 1;
      births &= births - 1;
      if (n >= P.F) { atomicMax(P.overflow, 1); continue; }
      const double* m = MREC(u);
      create_feature(FREC(n), SREC(n), new_id++, FLD(m, M_ZR), FLD(m, M_ZPHI), x, y, th, P.qt_r, P.qt_phi, P.p0);
      ++n;
    }
  }
  P.nfeat[p] = n;
  P.new_id[p] = new_id;
}

// ---- weight and feature decay (:377-404).  The reference removes from the list its for-loop
// walks, so the feature that slides into a vacated slot is not visited in this pass. ----
__global__ __launch_bounds__(128) void fs_decay_kernel(const FsParams P) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.M) return;
  const double x = P.x[p], y = P.y[p];
  const int n = P.nfeat[p];
  double wgt = 1.0;
  int n_assoc = 0;
  int kept = 0;            // slots written so far: survivors close ranks as the pass goes
  bool slid = false;       // the previous feature was discarded, this one takes its place unvisited
  for (int j = 0; j < n; ++j) {
    double* f = FREC(j);
    bool keep = true;
    if (slid) {
      slid = false;
    } else if (FLD_I(SREC(j), S_UPD, 0)) {
      wgt = wgt * FLD(f, F_W);
      ++n_assoc;
    } else {
      const double dx = FLD(f, F_M0) - x, dy = FLD(f, F_M1) - y;
      FLD(f, F_W) = P.p0;
      if (dx * dx + dy * dy <= P.r_max2) {
        const int cnt = FLD_I(f, F_INT, 0) - 1;
        FLD_I(f, F_INT, 0) = cnt;
        if (cnt < 0) { keep = false; slid = true; }   // discard
      }
    }
    if (keep) {
      if (kept != j) {
        double* g = FREC(kept);
#pragma unroll
        for (int w = 0; w < F_WORDS; ++w) FLD(g, w) = FLD(f, w);
      }
      ++kept;
    }
  }
  if (n_assoc == 0) wgt = P.p0;
  P.nfeat[p] = kept;
  P.wout[p] = wgt;
}

// feature slots of the drawn ancestors: block row y = slot
__global__ void fs_gather_kernel(const int64_t* __restrict__ idx, int64_t M, int F, const int32_t* __restrict__ nfeat_in,
                                 const int32_t* __restrict__ newid_in, int32_t* __restrict__ nfeat_out,
                                 int32_t* __restrict__ newid_out, const double* __restrict__ in, double* __restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = blockIdx.y;
  if (p >= M) return;
  const int64_t src = idx[p];
  const int n = nfeat_in[src];
  if (slot == 0) { nfeat_out[p] = n; newid_out[p] = newid_in[src]; }
  if (slot >= n) return;
  const double* a = in + tile_word(src, F, slot, F_WORDS);
  double* b = out + tile_word(p, F, slot, F_WORDS);
#pragma unroll
  for (int w = 0; w < F_WORDS; ++w) FLD(b, w) = FLD(a, w);
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
__global__ void fs_pack_rows_kernel(int64_t lo, int64_t n, int F, const int32_t* __restrict__ nfeat,
                                    const int32_t* __restrict__ newid, const double* __restrict__ feat,
                                    unsigned long long* __restrict__ out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int64_t p = lo + j;
  unsigned long long* row = out + j * (2 + (int64_t)9 * F);
  const int y = blockIdx.y;
  if (y == 0) { row[0] = (unsigned long long)(long long)nfeat[p]; row[1] = (unsigned long long)(long long)newid[p]; return; }
  const int slot = y - 1;
  if (slot >= nfeat[p]) return;
  unsigned long long* dst = row + 2 + (int64_t)9 * slot;
  const double* f = feat + tile_word(p, F, slot, F_WORDS);
#pragma unroll
  for (int w = 0; w < 7; ++w) dst[w] = (unsigned long long)__double_as_longlong(FLD(f, w));
  dst[7] = (unsigned long long)(long long)FLD_I(f, F_INT, 0);
  dst[8] = (unsigned long long)(long long)FLD_I(f, F_INT, 1);
}

__global__ void fs_unpack_rows_kernel(const int64_t* __restrict__ idx, int64_t row0_global, int64_t L, int64_t M, int F,
                                      const unsigned long long* __restrict__ rows, int32_t* __restrict__ nfeat,
                                      int32_t* __restrict__ newid, double* __restrict__ feat) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  int64_t rr = idx[p] - row0_global;
  rr = rr < 0 ? 0 : (rr > L - 1 ? L - 1 : rr);
  const unsigned long long* row = rows + rr * (2 + (int64_t)9 * F);
  const int n = (int)(long long)row[0];
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
  double* feat[2] = {nullptr, nullptr};   // feature slots, double-buffered for the gather
  int32_t* nfeat[2] = {nullptr, nullptr};
  int32_t* newid[2] = {nullptr, nullptr};
  double* scr = nullptr;                  // per-step predictions
  double* meas = nullptr; int mcap = 0;   // per-step measurements, mcap rows per tile
  double* noise = nullptr;
  double* d_truth = nullptr; int truth_cap = 0;   // (r, phi) of the landmarks in range, padded to a multiple of eight
  int32_t* overflow = nullptr;
  int16_t* d_ints = nullptr; double* d_normals = nullptr; int64_t normals_cap = 0;
  unsigned long long* stage = nullptr; int64_t stage_words = 0;   // feature rows on their way to / from another rank
  double std_r = 0.5, std_phi = 3 * M_PI / 180.0, r_max = 10.0, p0 = 0.01, thres_sig = 2.0;
};

static void fs_free(FsState* s) {
  if (!s) return;
  void* all[] = {s->feat[0], s->feat[1], s->nfeat[0], s->nfeat[1], s->newid[0], s->newid[1], s->scr, s->meas, s->noise, s->d_truth,
                 s->overflow, s->d_ints, s->d_normals, s->stage};
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

// uploads the step's inputs and enqueues the three per-particle kernels (pose sample .. weight / decay pass)
static int fs_enqueue_kernel(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6], const double* truth_r,
                             const double* truth_phi, const uint8_t* truth_valid, int L, const int16_t* ints,
                             const double* normals, uint64_t seed) {
  if (!c) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  if (!alpha || (L > 0 && (!truth_r || !truth_phi || !truth_valid)) || L < 0) return fail(c, MCLHIP_E_ARG, "alpha / truth arrays NULL");
  FsState* s = (FsState*)c->fs;
  const int64_t M = s->M;
  // landmarks in range of the true pose, in landmark order, as (r, phi) pairs: measure() returns
  // r_max + 1 for the others, which fast_slam() skips (:291).  Padded to a multiple of eight with readings
  // no feature is near.
  std::vector<double> near;
  for (int l = 0; l < L; ++l)
    if (truth_valid[l]) { near.push_back(truth_r[l]); near.push_back(truth_phi[l]); }
  const int n_valid = (int)(near.size() / 2);
  while ((near.size() / 2) % 8) { near.push_back(HUGE_VAL); near.push_back(0.0); }
  if (n_valid > s->mcap || (int)(near.size() / 2) > s->truth_cap) {
    F_HIP(hipStreamSynchronize(c->stream));
    if (s->d_truth) (void)hipFree(s->d_truth);
    if (s->meas) (void)hipFree(s->meas);
    s->d_truth = s->meas = nullptr; s->truth_cap = s->mcap = 0;
    const int cap = ((n_valid + n_valid / 4 + 8) + 7) & ~7;
    F_HIP(hipMalloc((void**)&s->d_truth, (size_t)2 * cap * sizeof(double)));
    F_HIP(hipMalloc((void**)&s->meas, (size_t)s->tiles * cap * M_WORDS * 64 * sizeof(double)));
    s->truth_cap = s->mcap = cap;
  }
  if (!near.empty()) {
    F_HIP(hipMemcpyAsync(s->d_truth, near.data(), near.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    F_HIP(hipStreamSynchronize(c->stream));   // `near` is a local
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
  P.feat = s->feat[a]; P.scr = s->scr; P.meas = s->meas; P.noise = s->noise;
  P.overflow = s->overflow;
  const dim3 fs_grid((unsigned)((M + 127) / 128));
  hipLaunchKernelGGL(fs_predict_kernel, fs_grid, dim3(128), 0, c->stream, P);
  if (n_valid > 0) {
    hipLaunchKernelGGL(fs_draw_kernel, fs_grid, dim3(128), 0, c->stream, P, (const double2*)s->d_truth);
    hipLaunchKernelGGL(fs_associate_kernel, fs_grid, dim3(128), 0, c->stream, P, (const double2*)s->d_truth);
    hipLaunchKernelGGL(fs_apply_kernel, fs_grid, dim3(128), 0, c->stream, P);
  }
  hipLaunchKernelGGL(fs_decay_kernel, fs_grid, dim3(128), 0, c->stream, P);
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
  s->M = M; s->F = capacity;
  s->tiles = (M + 63) / 64;
  const size_t slots = (size_t)s->tiles * capacity * 64;   // feature slots, padded to whole wavefront tiles
  for (int b = 0; b < 2; ++b) {
    F_HIP(hipMalloc((void**)&s->feat[b], slots * F_WORDS * sizeof(double)));
    F_HIP(hipMalloc((void**)&s->nfeat[b], M * sizeof(int32_t)));
    F_HIP(hipMalloc((void**)&s->newid[b], M * sizeof(int32_t)));
    F_HIP(hipMemset(s->nfeat[b], 0, M * sizeof(int32_t)));
    F_HIP(hipMemset(s->newid[b], 0, M * sizeof(int32_t)));
  }
  F_HIP(hipMalloc((void**)&s->scr, slots * S_WORDS * sizeof(double)));
  F_HIP(hipMalloc((void**)&s->noise, (size_t)2 * M * sizeof(double)));
  F_HIP(hipMalloc((void**)&s->overflow, sizeof(int32_t)));
  F_HIP(hipMemset(s->overflow, 0, sizeof(int32_t)));
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
  const int a = s->cur, b = 1 - s->cur;
  // normalise by the Python-order sum with its 1e-8 floor, then the guarded low-variance walk (:411-431)
  F_TRY(launch_sequential_total(c));                     // c->cdf_exact[M-1] = sum_w, added one by one
  const unsigned g = (unsigned)((M + 255) / 256);
  hipLaunchKernelGGL(fs_total_kernel, dim3(1), dim3(64), 0, c->stream, c->cdf_exact, M, 1e-8, c->scal);
  hipLaunchKernelGGL(fs_divide_kernel, dim3(g), dim3(256), 0, c->stream, c->w[c->cur], M, c->scal);
  F_TRY(launch_resample(c, r, MCLHIP_RESAMPLE_PARALLEL)); // flips c->cur; c->idx = ancestors
  // feature slots follow their ancestors
  hipLaunchKernelGGL(fs_gather_kernel, dim3(g, s->F), dim3(256), 0, c->stream, c->idx, M, s->F, s->nfeat[a], s->newid[a],
                     s->nfeat[b], s->newid[b], (const double*)s->feat[a], s->feat[b]);
  F_HIP(hipGetLastError());
  s->cur = b;
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
  return 2 + (int64_t)9 * ((FsState*)c->fs)->F;
}

int mclhip_fs_shard_pack(mclhip_ctx* c, const int64_t* lo, const int64_t* hi, int nslices, void* rows) {
  if (!c) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  if (!lo || !hi || nslices <= 0) return fail(c, MCLHIP_E_ARG, "lo/hi NULL or no slices");
  FsState* s = (FsState*)c->fs;
  const int64_t words = 2 + (int64_t)9 * s->F;
  int64_t total = 0;
  for (int k = 0; k < nslices; ++k) {
    if (lo[k] < 0 || hi[k] > s->M || hi[k] < lo[k]) return fail(c, MCLHIP_E_ARG, "slice outside the local range");
    total += hi[k] - lo[k];
  }
  if (total == 0) return MCLHIP_OK;
  if (!rows) return fail(c, MCLHIP_E_ARG, "rows is NULL");
  F_TRY(fs_stage(c, s, total * words));
  int64_t off = 0;
  for (int k = 0; k < nslices; ++k) {
    const int64_t n = hi[k] - lo[k];
    if (n == 0) continue;
    hipLaunchKernelGGL(fs_pack_rows_kernel, dim3((unsigned)((n + 63) / 64), s->F + 1), dim3(64), 0, c->stream, lo[k], n,
                       s->F, s->nfeat[s->cur], s->newid[s->cur], (const double*)s->feat[s->cur], s->stage + off * words);
    off += n;
  }
  F_HIP(hipGetLastError());
  F_HIP(hipMemcpyAsync(rows, s->stage, (size_t)total * words * 8, hipMemcpyDefault, c->stream));
  F_HIP(hipStreamSynchronize(c->stream));
  return MCLHIP_OK;
}

int mclhip_fs_shard_unpack(mclhip_ctx* c, const void* rows, int64_t L, int64_t row0_global) {
  if (!c) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  if (!rows || L <= 0) return fail(c, MCLHIP_E_ARG, "rows NULL or empty");
  FsState* s = (FsState*)c->fs;
  const int64_t words = 2 + (int64_t)9 * s->F;
  F_TRY(fs_stage(c, s, L * words));
  F_HIP(hipMemcpyAsync(s->stage, rows, (size_t)L * words * 8, hipMemcpyDefault, c->stream));
  const int b = 1 - s->cur;
  hipLaunchKernelGGL(fs_unpack_rows_kernel, dim3((unsigned)((s->M + 63) / 64), s->F + 1), dim3(64), 0, c->stream, c->idx,
                     row0_global, L, s->M, s->F, (const unsigned long long*)s->stage, s->nfeat[b], s->newid[b], s->feat[b]);
  F_HIP(hipGetLastError());
  F_HIP(hipStreamSynchronize(c->stream));
  s->cur = b;
  return MCLHIP_OK;
}

}  // extern "C"
