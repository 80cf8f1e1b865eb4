// FastSLAM 1.0 with unknown correspondence -- the reference's 13_fast_slam/fast_slam1.py
// (fast_slam(), :240-433) on the GPU.
//
// One thread per particle for the filter step: pose sample (Table 5.3 with the twelve-integer
// normal sampler, :160-189), per-feature measurement prediction / Jacobian / innovation
// covariance (:259-282), maximum-likelihood association of every sensed landmark with a 2-sigma
// Mahalanobis gate (:286-315), feature birth (:318-347) or EKF update (:348-375), and the weight /
// feature-decay pass that removes from the list it iterates (:377-404).  Features live in
// fixed-capacity slots, structure of arrays [slot][particle], so that neighbouring threads touch
// neighbouring addresses.  Resampling reuses the order-exact sequential sum / CDF of the particle
// filter (weights.hip), then one kernel gathers the feature slots of the drawn ancestors.
//
// Every measurement is drawn per PARTICLE (the reference calls measure() inside the particle
// loop, :288): the caller passes the noise-free (r, phi) of each landmark from the true pose and
// either pre-drawn standard normals (stream parity) or a seed (device generator).
//
// Compiled with -ffp-contract=off.
#include <cmath>

#include "device_rng.h"
#include "mclhip_internal.h"

namespace mcl {

struct FsParams {
  int64_t M;
  int F, L;                 // feature capacity per particle, landmarks
  double v, w, dt, c1, c2, c3;
  double qt_r, qt_phi;      // std_r**2, std_phi**2
  double std_r, std_phi, r_max, r_max2, p0, gate2;
  const int16_t* ints;      // [M][3][12] or null
  const double* normals;    // [M][n_norm] or null
  int n_norm;
  uint64_t seed;
  int64_t index0;           // global index of local particle 0 (keys the device generators)
  const double* tr; const double* tphi; const uint8_t* tvalid;   // [L] noise-free measurements
  double *x, *y, *th, *wout;
  int32_t *nfeat, *new_id, *fi, *fid;
  double *fm0, *fm1, *c00, *c01, *c10, *c11, *fw;
  // per-step scratch, [slot][particle]
  double *zr, *zphi, *h00, *h01, *h10, *h11, *iq00, *iq01, *iq10, *iq11, *detq;
  uint8_t* upd;
  int32_t* overflow;
};

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

// standard normal from the counter-based generator (Box-Muller on two 32-bit words)
__device__ inline double device_normal(uint64_t particle, uint32_t draw, uint64_t seed) {
  uint32_t ctr[4] = {(uint32_t)particle, (uint32_t)(particle >> 32), draw, 0x4653u};
  philox4x32(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  const double u1 = ((double)ctr[0] + 1.0) * (1.0 / 4294967296.0);   // (0, 1]
  const double u2 = (double)ctr[1] * (1.0 / 4294967296.0);
  return sqrt(-2.0 * log(u1)) * cos(2.0 * M_PI * u2);
}

__global__ __launch_bounds__(128) void fs_step_kernel(FsParams P) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.M) return;
  const int64_t M = P.M;
#define AT(arr, slot) arr[(int64_t)(slot) * M + p]
  // ---- pose sample (:160-189) ----
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

  // ---- per-feature prediction (:259-282) ----
  int n = P.nfeat[p];
  const int n_old = n;
  for (int j = 0; j < n_old; ++j) {
    double r, phi;
    M2 H;
    predict_measurement(AT(P.fm0, j), AT(P.fm1, j), x, y, th, &r, &phi, &H);
    const M2 C{AT(P.c00, j), AT(P.c01, j), AT(P.c10, j), AT(P.c11, j)};
    M2 Q = mul(mul(H, C), tr(H));              // :276
    Q.a = Q.a + P.qt_r; Q.b = Q.b + 0.0; Q.c = Q.c + 0.0; Q.d = Q.d + P.qt_phi;
    double det;
    const M2 iq = inv_lu(Q, &det);
    AT(P.zr, j) = r; AT(P.zphi, j) = phi;
    AT(P.h00, j) = H.a; AT(P.h01, j) = H.b; AT(P.h10, j) = H.c; AT(P.h11, j) = H.d;
    AT(P.iq00, j) = iq.a; AT(P.iq01, j) = iq.b; AT(P.iq10, j) = iq.c; AT(P.iq11, j) = iq.d;
    AT(P.detq, j) = det;
    AT(P.upd, j) = 0;
  }

  // ---- every landmark: draw the measurement, associate, create or update (:286-375) ----
  // The association of a measurement reads only what the prediction pass stored (z_pred, invQ,
  // det Q of the features that existed before this step), never what an earlier measurement of
  // the same step changed.  So sixteen measurements are associated together -- one sweep over the
  // stored features serves all sixteen (8: 29 ms, 16: 26 ms at 2^20 x 256) -- and then applied one by one in landmark order.
  constexpr int kChunk = 16;
  int used = 0;
  int l = 0;
  while (l < P.L) {
    double mzr[kChunk], mzphi[kChunk], best_w[kChunk];
    int best_c[kChunk];
    int nm = 0;
    for (; l < P.L && nm < kChunk; ++l) {
      if (!P.tvalid[l]) continue;               // measure() returns r_max + 1: skipped at :291
      double n_r, n_phi;
      if (P.normals) {
        n_r = P.normals[p * P.n_norm + used];
        n_phi = P.normals[p * P.n_norm + used + 1];
      } else {
        n_r = device_normal((uint64_t)(P.index0 + p), (uint32_t)used, P.seed);
        n_phi = device_normal((uint64_t)(P.index0 + p), (uint32_t)used + 1u, P.seed);
      }
      used += 2;
      const double zr = P.tr[l] + n_r * P.std_r;   // :145-152
      double zphi = P.tphi[l] + n_phi * P.std_phi;
      if (zphi > M_PI) zphi -= 2 * M_PI;
      else if (zphi < -M_PI) zphi += 2 * M_PI;
      if (!(zr <= P.r_max)) continue;           // :291
      mzr[nm] = zr; mzphi[nm] = zphi; best_w[nm] = 0.0; best_c[nm] = -1;
      ++nm;
    }
    if (nm == 0) continue;
    for (int j = 0; j < n_old; ++j) {
      const double fr = AT(P.zr, j), fphi = AT(P.zphi, j);
      const double q00 = AT(P.iq00, j), q01 = AT(P.iq01, j), q10 = AT(P.iq10, j), q11 = AT(P.iq11, j);
      const double dq = AT(P.detq, j);
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        if (u < nm) {
          const double e0 = mzr[u] - fr;
          const double e1 = wrap_residual(mzphi[u] - fphi);
          const double t0 = e0 * q00 + e1 * q10;   // r.T @ invQ
          const double t1 = e0 * q01 + e1 * q11;
          const double d2 = t0 * e0 + t1 * e1;
          if (d2 <= P.gate2) {                  // :307
            const double eta = 1.0 / (2.0 * M_PI * sqrt(dq));
            const double wj = eta * exp(-0.5 * d2);
            if (best_w[u] < wj) { best_w[u] = wj; best_c[u] = j; }
          }
        }
      }
    }
#pragma unroll 1
    for (int u = 0; u < nm; ++u) {
      const double zr = mzr[u], zphi = mzphi[u], w_max = best_w[u];
      const int c = best_c[u];
      if (c == -1) {                              // new feature (:318-347)
        if (n >= P.F) { atomicMax(P.overflow, 1); continue; }
        const double m0 = zr * cos(zphi + th) + x, m1 = zr * sin(zphi + th) + y;   // :201-202
        double r_, phi_;
        M2 H;
        predict_measurement(m0, m1, x, y, th, &r_, &phi_, &H);
        const M2 iH = inv_lu(H, nullptr);
        const M2 Qt{P.qt_r, 0.0, 0.0, P.qt_phi};
        const M2 C = mul(mul(iH, Qt), tr(iH));   // :336
        AT(P.fm0, n) = m0; AT(P.fm1, n) = m1;
        AT(P.c00, n) = C.a; AT(P.c01, n) = C.b; AT(P.c10, n) = C.c; AT(P.c11, n) = C.d;
        AT(P.fi, n) = 1; AT(P.fw, n) = P.p0; AT(P.fid, n) = P.new_id[p];
        AT(P.upd, n) = 1;
        P.new_id[p] += 1;
        ++n;
      } else {                                    // EKF update (:348-375)
        const M2 H{AT(P.h00, c), AT(P.h01, c), AT(P.h10, c), AT(P.h11, c)};
        const M2 iq{AT(P.iq00, c), AT(P.iq01, c), AT(P.iq10, c), AT(P.iq11, c)};
        const M2 C{AT(P.c00, c), AT(P.c01, c), AT(P.c10, c), AT(P.c11, c)};
        const M2 K = mul(mul(C, tr(H)), iq);     // :360
        const double e0 = zr - AT(P.zr, c);
        const double e1 = wrap_residual(zphi - AT(P.zphi, c));
        AT(P.fm0, c) = AT(P.fm0, c) + (K.a * e0 + K.b * e1);   // :369
        AT(P.fm1, c) = AT(P.fm1, c) + (K.c * e0 + K.d * e1);
        const M2 KH = mul(K, H);
        const M2 IKH{1.0 - KH.a, 0.0 - KH.b, 0.0 - KH.c, 1.0 - KH.d};
        const M2 Cn = mul(IKH, C);               // :370
        AT(P.c00, c) = Cn.a; AT(P.c01, c) = Cn.b; AT(P.c10, c) = Cn.c; AT(P.c11, c) = Cn.d;
        AT(P.fi, c) = AT(P.fi, c) + 1;
        AT(P.upd, c) = 1;
        AT(P.fw, c) = w_max;
      }
    }
  }

  // ---- weight and feature decay (:377-404).  The reference removes from the list its for-loop
  // walks, so the feature that slides into a vacated slot is not visited in this pass. ----
  double wgt = 1.0;
  int n_assoc = 0;
  int it = 0;
  while (it < n) {
    const int j = it;
    ++it;
    if (AT(P.upd, j)) {
      wgt = wgt * AT(P.fw, j);
      ++n_assoc;
    } else {
      const double dx = AT(P.fm0, j) - x, dy = AT(P.fm1, j) - y;
      AT(P.fw, j) = P.p0;
      if (dx * dx + dy * dy <= P.r_max2) {
        const int cnt = AT(P.fi, j) - 1;
        AT(P.fi, j) = cnt;
        if (cnt < 0) {                          // discard: later features move down one slot
          for (int s = j; s + 1 < n; ++s) {
            AT(P.fm0, s) = AT(P.fm0, s + 1); AT(P.fm1, s) = AT(P.fm1, s + 1);
            AT(P.c00, s) = AT(P.c00, s + 1); AT(P.c01, s) = AT(P.c01, s + 1);
            AT(P.c10, s) = AT(P.c10, s + 1); AT(P.c11, s) = AT(P.c11, s + 1);
            AT(P.fi, s) = AT(P.fi, s + 1); AT(P.fw, s) = AT(P.fw, s + 1); AT(P.fid, s) = AT(P.fid, s + 1);
            AT(P.upd, s) = AT(P.upd, s + 1);
          }
          --n;
        }
      }
    }
  }
  if (n_assoc == 0) wgt = P.p0;
  P.nfeat[p] = n;
  P.wout[p] = wgt;
#undef AT
}

// feature slots of the drawn ancestors
__global__ void fs_gather_kernel(const int64_t* __restrict__ idx, int64_t M, int F, const int32_t* __restrict__ nfeat_in,
                                 const int32_t* __restrict__ newid_in, int32_t* __restrict__ nfeat_out,
                                 int32_t* __restrict__ newid_out, const double* const* in_d, double* const* out_d,
                                 const int32_t* const* in_i, int32_t* const* out_i) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int slot = blockIdx.y;
  if (p >= M) return;
  const int64_t src = idx[p];
  const int n = nfeat_in[src];
  if (slot == 0) { nfeat_out[p] = n; newid_out[p] = newid_in[src]; }
  if (slot >= n) return;
  const int64_t a = (int64_t)slot * M + src, b = (int64_t)slot * M + p;
#pragma unroll
  for (int f = 0; f < 7; ++f) out_d[f][b] = in_d[f][a];
#pragma unroll
  for (int f = 0; f < 2; ++f) out_i[f][b] = in_i[f][a];
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
__global__ void fs_pack_rows_kernel(int64_t lo, int64_t n, int64_t M, int F, const int32_t* __restrict__ nfeat,
                                    const int32_t* __restrict__ newid, const double* const* fd, const int32_t* const* fi,
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
  const int64_t a = (int64_t)slot * M + p;
#pragma unroll
  for (int f = 0; f < 7; ++f) dst[f] = (unsigned long long)__double_as_longlong(fd[f][a]);
  dst[7] = (unsigned long long)(long long)fi[0][a];
  dst[8] = (unsigned long long)(long long)fi[1][a];
}

__global__ void fs_unpack_rows_kernel(const int64_t* __restrict__ idx, int64_t row0_global, int64_t L, int64_t M, int F,
                                      const unsigned long long* __restrict__ rows, int32_t* __restrict__ nfeat,
                                      int32_t* __restrict__ newid, double* const* fd, int32_t* const* fi) {
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
  const int64_t b = (int64_t)slot * M + p;
#pragma unroll
  for (int f = 0; f < 7; ++f) fd[f][b] = __longlong_as_double((long long)src[f]);
  fi[0][b] = (int)(long long)src[7];
  fi[1][b] = (int)(long long)src[8];
}

}  // namespace mcl

using namespace mcl;

// per-context FastSLAM storage
struct FsState {
  int F = 0;
  int64_t M = 0;
  int cur = 0;
  // double fields: m0 m1 c00 c01 c10 c11 w ; int fields: i id
  double* fd[2][7] = {{nullptr}};
  int32_t* fint[2][2] = {{nullptr}};
  int32_t* nfeat[2] = {nullptr, nullptr};
  int32_t* newid[2] = {nullptr, nullptr};
  double* scratch[11] = {nullptr};
  uint8_t* upd = nullptr;
  int32_t* overflow = nullptr;
  double* d_truth = nullptr; uint8_t* d_valid = nullptr; int truth_cap = 0;
  int16_t* d_ints = nullptr; double* d_normals = nullptr; int64_t normals_cap = 0;
  void** d_ptrs = nullptr;   // pointer tables: [0..6] doubles of buffer 0, [7..13] of buffer 1, [14..15] / [16..17] ints
  unsigned long long* stage = nullptr; int64_t stage_words = 0;   // feature rows on their way to / from another rank
  double std_r = 0.5, std_phi = 3 * M_PI / 180.0, r_max = 10.0, p0 = 0.01, thres_sig = 2.0;
};

static void fs_free(FsState* s) {
  if (!s) return;
  for (int b = 0; b < 2; ++b) {
    for (auto& q : s->fd[b]) if (q) (void)hipFree(q);
    for (auto& q : s->fint[b]) if (q) (void)hipFree(q);
    if (s->nfeat[b]) (void)hipFree(s->nfeat[b]);
    if (s->newid[b]) (void)hipFree(s->newid[b]);
  }
  for (auto& q : s->scratch) if (q) (void)hipFree(q);
  void* rest[] = {s->upd, s->overflow, s->d_truth, s->d_valid, s->d_ints, s->d_normals, s->d_ptrs, s->stage};
  for (void* q : rest) if (q) (void)hipFree(q);
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

static int fs_upload_tables(mclhip_ctx* c, FsState* s) {
  void* tab[18];
  for (int f = 0; f < 7; ++f) { tab[f] = s->fd[0][f]; tab[7 + f] = s->fd[1][f]; }
  for (int f = 0; f < 2; ++f) { tab[14 + f] = s->fint[0][f]; tab[16 + f] = s->fint[1][f]; }
  MCL_HIP(c, hipMemcpyAsync(s->d_ptrs, tab, sizeof(tab), hipMemcpyHostToDevice, c->stream));
  MCL_HIP(c, hipStreamSynchronize(c->stream));   // tab is a stack array
  return MCLHIP_OK;
}

#define F_TRY(expr) do { int rc__ = (expr); if (rc__ != MCLHIP_OK) return rc__; } while (0)
#define F_HIP(expr) MCL_HIP(c, expr)

// uploads the step's inputs and enqueues the per-particle kernel (pose sample .. weight / decay pass)
static int fs_enqueue_kernel(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6], const double* truth_r,
                             const double* truth_phi, const uint8_t* truth_valid, int L, const int16_t* ints,
                             const double* normals, uint64_t seed) {
  if (!c) return MCLHIP_E_ARG;
  F_HIP(hipSetDevice(c->device));
  if (!c->fs) return fail(c, MCLHIP_E_STATE, "call mclhip_fs_init first");
  if (!alpha || (L > 0 && (!truth_r || !truth_phi || !truth_valid)) || L < 0) return fail(c, MCLHIP_E_ARG, "alpha / truth arrays NULL");
  FsState* s = (FsState*)c->fs;
  const int64_t M = s->M;
  int n_valid = 0;
  for (int l = 0; l < L; ++l) n_valid += truth_valid[l] ? 1 : 0;
  if (L > s->truth_cap) {
    if (s->d_truth) (void)hipFree(s->d_truth);
    if (s->d_valid) (void)hipFree(s->d_valid);
    F_HIP(hipMalloc((void**)&s->d_truth, (size_t)2 * L * sizeof(double)));
    F_HIP(hipMalloc((void**)&s->d_valid, (size_t)L));
    s->truth_cap = L;
  }
  if (L > 0) {
    F_HIP(hipMemcpyAsync(s->d_truth, truth_r, L * sizeof(double), hipMemcpyHostToDevice, c->stream));
    F_HIP(hipMemcpyAsync(s->d_truth + L, truth_phi, L * sizeof(double), hipMemcpyHostToDevice, c->stream));
    F_HIP(hipMemcpyAsync(s->d_valid, truth_valid, L, hipMemcpyHostToDevice, c->stream));
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
  P.M = M; P.F = s->F; P.L = L;
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
  P.tr = s->d_truth; P.tphi = s->d_truth ? s->d_truth + L : nullptr; P.tvalid = s->d_valid;
  P.x = (double*)c->px[c->cur]; P.y = (double*)c->py[c->cur]; P.th = (double*)c->pth[c->cur]; P.wout = c->w[c->cur];
  P.nfeat = s->nfeat[a]; P.new_id = s->newid[a];
  P.fm0 = s->fd[a][0]; P.fm1 = s->fd[a][1]; P.c00 = s->fd[a][2]; P.c01 = s->fd[a][3]; P.c10 = s->fd[a][4];
  P.c11 = s->fd[a][5]; P.fw = s->fd[a][6];
  P.fi = s->fint[a][0]; P.fid = s->fint[a][1];
  P.zr = s->scratch[0]; P.zphi = s->scratch[1]; P.h00 = s->scratch[2]; P.h01 = s->scratch[3]; P.h10 = s->scratch[4];
  P.h11 = s->scratch[5]; P.iq00 = s->scratch[6]; P.iq01 = s->scratch[7]; P.iq10 = s->scratch[8]; P.iq11 = s->scratch[9];
  P.detq = s->scratch[10];
  P.upd = s->upd; P.overflow = s->overflow;
  if (!ints && !normals) { /* device generator for both streams */ }
  hipLaunchKernelGGL(fs_step_kernel, dim3((unsigned)((M + 127) / 128)), dim3(128), 0, c->stream, P);
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
  const size_t n = (size_t)M * capacity;
  for (int b = 0; b < 2; ++b) {
    for (auto& q : s->fd[b]) F_HIP(hipMalloc((void**)&q, n * sizeof(double)));
    for (auto& q : s->fint[b]) F_HIP(hipMalloc((void**)&q, n * sizeof(int32_t)));
    F_HIP(hipMalloc((void**)&s->nfeat[b], M * sizeof(int32_t)));
    F_HIP(hipMalloc((void**)&s->newid[b], M * sizeof(int32_t)));
    F_HIP(hipMemset(s->nfeat[b], 0, M * sizeof(int32_t)));
    F_HIP(hipMemset(s->newid[b], 0, M * sizeof(int32_t)));
  }
  for (auto& q : s->scratch) F_HIP(hipMalloc((void**)&q, n * sizeof(double)));
  F_HIP(hipMalloc((void**)&s->upd, n));
  F_HIP(hipMalloc((void**)&s->overflow, sizeof(int32_t)));
  F_HIP(hipMemset(s->overflow, 0, sizeof(int32_t)));
  F_HIP(hipMalloc((void**)&s->d_ptrs, 36 * sizeof(void*)));
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
  void* tab[36];
  for (int f = 0; f < 7; ++f) { tab[f] = s->fd[a][f]; tab[7 + f] = s->fd[b][f]; }
  for (int f = 0; f < 2; ++f) { tab[14 + f] = s->fint[a][f]; tab[16 + f] = s->fint[b][f]; }
  F_HIP(hipMemcpyAsync(s->d_ptrs, tab, 18 * sizeof(void*), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(fs_gather_kernel, dim3(g, s->F), dim3(256), 0, c->stream, c->idx, M, s->F, s->nfeat[a], s->newid[a],
                     s->nfeat[b], s->newid[b], (const double* const*)s->d_ptrs, (double* const*)(s->d_ptrs + 7),
                     (const int32_t* const*)(s->d_ptrs + 14), (int32_t* const*)(s->d_ptrs + 16));
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
  const size_t n = (size_t)M * F;
  std::vector<double> tmp(n);
  auto pull_d = [&](int f, double* dst, int stride, int offset) -> int {   // [slot][particle] -> [particle][slot][stride]
    F_HIP(hipMemcpy(tmp.data(), s->fd[a][f], n * sizeof(double), hipMemcpyDeviceToHost));
    for (int64_t p = 0; p < M; ++p)
      for (int j = 0; j < F; ++j) dst[((size_t)p * F + j) * stride + offset] = j < nf[p] ? tmp[(size_t)j * M + p] : 0.0;
    return MCLHIP_OK;
  };
  if (f_m) { F_TRY(pull_d(0, f_m, 2, 0)); F_TRY(pull_d(1, f_m, 2, 1)); }
  if (f_cov) for (int q = 0; q < 4; ++q) F_TRY(pull_d(2 + q, f_cov, 4, q));
  if (f_w) F_TRY(pull_d(6, f_w, 1, 0));
  std::vector<int32_t> ti(n);
  auto pull_i = [&](int f, int32_t* dst, int32_t fill) -> int {
    F_HIP(hipMemcpy(ti.data(), s->fint[a][f], n * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int64_t p = 0; p < M; ++p)
      for (int j = 0; j < F; ++j) dst[(size_t)p * F + j] = j < nf[p] ? ti[(size_t)j * M + p] : fill;
    return MCLHIP_OK;
  };
  if (f_i) F_TRY(pull_i(0, f_i, 0));
  if (f_id) F_TRY(pull_i(1, f_id, -1));
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
  F_TRY(fs_upload_tables(c, s));
  int64_t off = 0;
  for (int k = 0; k < nslices; ++k) {
    const int64_t n = hi[k] - lo[k];
    if (n == 0) continue;
    hipLaunchKernelGGL(fs_pack_rows_kernel, dim3((unsigned)((n + 63) / 64), s->F + 1), dim3(64), 0, c->stream, lo[k], n,
                       s->M, s->F, s->nfeat[s->cur], s->newid[s->cur], (const double* const*)(s->d_ptrs + 7 * s->cur),
                       (const int32_t* const*)(s->d_ptrs + 14 + 2 * s->cur), s->stage + off * words);
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
  F_TRY(fs_upload_tables(c, s));
  F_HIP(hipMemcpyAsync(s->stage, rows, (size_t)L * words * 8, hipMemcpyDefault, c->stream));
  const int b = 1 - s->cur;
  hipLaunchKernelGGL(fs_unpack_rows_kernel, dim3((unsigned)((s->M + 63) / 64), s->F + 1), dim3(64), 0, c->stream, c->idx,
                     row0_global, L, s->M, s->F, (const unsigned long long*)s->stage, s->nfeat[b], s->newid[b],
                     (double* const*)(s->d_ptrs + 7 * b), (int32_t* const*)(s->d_ptrs + 14 + 2 * b));
  F_HIP(hipGetLastError());
  F_HIP(hipStreamSynchronize(c->stream));
  s->cur = b;
  return MCLHIP_OK;
}

}  // extern "C"
