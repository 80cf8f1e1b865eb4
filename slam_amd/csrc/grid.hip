// Grid (histogram) localisation, Table 8.1 -- the reference's robot.py on the GPU.
//
//   predict (robot.py:125-190): pk_pred[i,j,d] = sum over the 7x7xD neighbourhood of
//       pk[ni,nj,nd] * p(x_t = cell(i,j,d) | u_t, x_t-1 = cell(ni,nj,nd))
//   for interior cells at or above the obstacle threshold, 0 elsewhere.  The density
//   (robot.py:76-122) depends on the two cells only through their offset and the two headings,
//   up to the rounding of absolute coordinates (measured: <= 3e-13 relative on the reference's
//   own map), so it is evaluated once per control into a coefficient table
//   K[d][row offset][col offset][nd] and the prediction becomes a 7x7 stencil with a DxD matrix
//   per offset: LDS-tiled, coefficients broadcast through scalar loads, float64 FMAs.
//
//   update (robot.py:277-310): per free cell and heading, the sector range query of the particle
//   filter and the product of beam likelihoods; pk = float32(pk_pred * q); pk /= pk.sum() in
//   float32 (numpy's pairwise order).  One wavefront per CELL: the beam table holds all
//   (heading, beam) pairs, so the geometry of a cell's obstacle neighbourhood is evaluated once
//   for all headings (raycast.hip, GRID mode).
//
// Compiled with -ffp-contract=off; the stencil asks for FMAs explicitly.
#include <cmath>
#include <numeric>

#include "mclhip_internal.h"

namespace mcl {

constexpr int kReach = 3;              // neighbourhood half-width (robot.py:161)
constexpr int kSide = 2 * kReach + 1;  // 7
constexpr int kOffsets = kSide * kSide;

struct GridMotion {
  double v, w, dt;
  double b1, b2, b3;   // variances alpha0|v|+alpha1|w|, ... (robot.py:118-120)
  double res_x, res_y, res_theta;
};

__device__ inline double heading_of(int d, double res_theta) {
  return (double)d * res_theta * M_PI / 180.0;   // d*res_theta*np.pi/180.0 (robot.py:159)
}

__device__ inline double normal_density(double a, double b) {
  return 1.0 / sqrt(2.0 * M_PI * b) * exp(-0.5 * (a * a) / b);   // robot.py:65
}

// One thread per table entry.  Layout [d / DT][offset][nd / NC][d % DT][nd % NC] so that the
// DT x NC block a stencil step needs is contiguous.
__global__ void grid_coeff_kernel(GridMotion m, int D, int DT, int NC, double* __restrict__ K) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= D * kOffsets * D) return;
  const int nd = t % D, off = (t / D) % kOffsets, d = t / (D * kOffsets);
  const int a = off / kSide, b = off % kSide;
  // target cell at the origin, previous cell at the offset (robot.py:159, :178)
  const double x1 = 0.0, y1 = 0.0, ha1 = heading_of(d, m.res_theta);
  const double x0 = (double)(b - kReach) * m.res_x, y0 = (double)(a - kReach) * m.res_y;
  const double ha0 = heading_of(nd, m.res_theta);
  const double c = cos(ha0), s = sin(ha0);
  const double t1 = (x0 - x1) * c + (y0 - y1) * s;            // robot.py:94
  double t2 = (y0 - y1) * c - (x0 - x1) * s;                  // :95
  if (fabs(t2) < 1e-6) t2 = 1e-6;                             // :97-98
  const double mu = 1.0 / 2.0 * t1 / t2;                      // :100
  const double mx = (x0 + x1) / 2, my = (y0 + y1) / 2;
  const double cx = mx + mu * (y0 - y1);                      // :106
  const double cy = my + mu * (x1 - x0);                      // :107
  const double ex = x0 - cx, ey = y0 - cy;
  const double cr = sqrt(ex * ex + ey * ey);                  // :108
  const double dha = atan2(y1 - cy, x1 - cx) - atan2(y0 - cy, x0 - cx);   // :109
  const double v_hat = dha / m.dt * cr;                       // :110
  const double w_hat = dha / m.dt;                            // :111
  const double g_hat = (ha1 - ha0) / m.dt - w_hat;            // :112
  const double p = normal_density(m.v - v_hat, m.b1) * normal_density(m.w - w_hat, m.b2) * normal_density(g_hat, m.b3);
  const int nchunk = D / NC;
  const size_t idx = ((((size_t)(d / DT) * kOffsets + off) * nchunk + nd / NC) * DT + d % DT) * NC + nd % NC;
  K[idx] = p;
}

// (16 CX) x 16 output cells per block, CX cells (16 columns apart) and DT headings per thread, neighbour
// headings staged NC at a time.  The coefficients are uniform over the block and arrive through scalar loads;
// every one of them feeds CX multiply-adds per lane, which divides the scalar traffic per unit of arithmetic.
#ifndef GRID_CX
#define GRID_CX 2
#endif
template <int DT, int NC, int CX = GRID_CX>
__global__ __launch_bounds__(256) void grid_predict_kernel(const float* __restrict__ pk, const double* __restrict__ K,
                                                           const uint8_t* __restrict__ mask, double* __restrict__ out,
                                                           int H, int W, int D) {
  constexpr int TSX = 16 * CX, TSY = 16, TWX = TSX + 2 * kReach, TWY = TSY + 2 * kReach;
  __shared__ float tile[TWX * TWY * NC];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int j0 = blockIdx.x * TSX, i0 = blockIdx.y * TSY;
  const int dg = blockIdx.z;
  const int nchunk = D / NC;
  double acc[CX][DT];
#pragma unroll
  for (int h = 0; h < CX; ++h)
#pragma unroll
    for (int q = 0; q < DT; ++q) acc[h][q] = 0.0;
  for (int ch = 0; ch < nchunk; ++ch) {
    __syncthreads();
    for (int t = threadIdx.x; t < TWX * TWY * NC; t += 256) {
      const int n = t % NC, c = (t / NC) % TWX, r = t / (NC * TWX);
      const int gi = i0 + r - kReach, gj = j0 + c - kReach;
      float v = 0.0f;
      if (gi >= 0 && gi < H && gj >= 0 && gj < W) v = pk[((size_t)gi * W + gj) * D + ch * NC + n];
      tile[t] = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int off = 0; off < kOffsets; ++off) {
      const int a = off / kSide, b = off % kSide;
      const float* tp = &tile[((ty + a) * TWX + tx + b) * NC];
      const double* Ko = K + (((size_t)dg * kOffsets + off) * nchunk + ch) * (DT * NC);   // uniform over the block
#pragma unroll
      for (int n = 0; n < NC; ++n) {
        double v[CX];
#pragma unroll
        for (int h = 0; h < CX; ++h) v[h] = (double)tp[h * 16 * NC + n];
#pragma unroll
        for (int q = 0; q < DT; ++q) {
          const double k = Ko[q * NC + n];
#pragma unroll
          for (int h = 0; h < CX; ++h) acc[h][q] = fma(k, v[h], acc[h][q]);
        }
      }
    }
  }
  const int i = i0 + ty;
#pragma unroll
  for (int h = 0; h < CX; ++h) {
    const int j = j0 + tx + 16 * h;
    if (i < H && j < W) {
      const bool live = i >= kReach && i < H - kReach && j >= kReach && j < W - kReach && (mask[(size_t)i * W + j] & 1);   // robot.py:150-157
#pragma unroll
      for (int q = 0; q < DT; ++q) out[((size_t)i * W + j) * D + dg * DT + q] = live ? acc[h][q] : 0.0;
    }
  }
}

__global__ void grid_prior_kernel(const uint8_t* __restrict__ mask, int64_t ncell, int D, float value, float* __restrict__ pk) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ncell * D) return;
  pk[t] = (mask[t / D] & 2) ? value : 0.0f;   // robot.py:49-59
}

// np.sum of a contiguous float32 vector: 8192-element chunks, pairwise inside with float32
// accumulators, chunk results added left to right.
template <int DEPTH>
__device__ float pairwise_f32(const float* a, int n) {
  if (n < 8) {
    float s = 0.0f;
    for (int i = 0; i < n; ++i) s = s + a[i];
    return s;
  }
  if (n <= 128) {
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
    }
    float s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) s = s + a[i];
    return s;
  }
  int h = n / 2;
  h -= h % 8;
  if constexpr (DEPTH > 0) {
    return pairwise_f32<DEPTH - 1>(a, h) + pairwise_f32<DEPTH - 1>(a + h, n - h);
  } else {
    return 0.0f;  // unreachable for n <= 8192
  }
}

// thread (leaf, j) owns accumulator j of a 128-element leaf; xor butterflies rebuild numpy's tree
__global__ __launch_bounds__(512) void chunk_sum_f32_kernel(const float* __restrict__ w, int64_t n, float* __restrict__ chunk_sums) {
  __shared__ float wave_sum[8];
  const int64_t base = (int64_t)blockIdx.x * kSumChunk;
  const int cnt = (int)((n - base) < kSumChunk ? (n - base) : kSumChunk);
  if (cnt == kSumChunk) {
    const int leaf = threadIdx.x >> 3, j = threadIdx.x & 7;
    const float* a = w + base + leaf * 128 + j;
    float r = a[0];
#pragma unroll
    for (int g = 1; g < 16; ++g) r = r + a[8 * g];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) r = r + __shfl_xor(r, d, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0)
      chunk_sums[blockIdx.x] = ((wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3])) +
                               ((wave_sum[4] + wave_sum[5]) + (wave_sum[6] + wave_sum[7]));
  } else {
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = pairwise_f32<7>(w + base, cnt);
  }
}

// the chunk sums added strictly left to right by one lane -- out of LDS, where the whole block has staged them (read
// from memory sixteen at a time the chain waited for a memory round trip per batch: 0.79 ms for the 36864 chunks of a
// 2048 x 2048 x 72 belief)
constexpr int kTotalStage = 8192;
__global__ __launch_bounds__(1024) void total_sum_f32_kernel(const float* __restrict__ chunk_sums, int64_t nchunks, float* __restrict__ total) {
  __shared__ __align__(16) float s_v[kTotalStage];
  float t = 0.0f;
  for (int64_t base = 0; base < nchunks; base += kTotalStage) {
    const int cnt = (int)((nchunks - base) < kTotalStage ? (nchunks - base) : kTotalStage);
    __syncthreads();
    for (int k = threadIdx.x; k < cnt; k += blockDim.x) s_v[k] = chunk_sums[base + k];
    __syncthreads();
    if (threadIdx.x == 0) {
      int i = 0;
      if (base == 0) { t = s_v[0]; i = 1; }
      for (; i + 16 <= cnt; i += 16) {   // reads issued together, additions strictly left to right
        float v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = s_v[i + j];
#pragma unroll
        for (int j = 0; j < 16; ++j) t = t + v[j];
      }
      for (; i < cnt; ++i) t = t + s_v[i];
    }
  }
  if (threadIdx.x == 0) total[0] = t;
}

__global__ void divide_f32_kernel(float* __restrict__ w, int64_t n, const float* __restrict__ total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = w[i] / total[0];   // robot.py:308
}

template <typename P>
static int galloc(mclhip_ctx* c, P** p, size_t bytes) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (bytes == 0) bytes = 16;
  MCL_HIP(c, hipMalloc((void**)p, bytes));
  return MCLHIP_OK;
}

static int guse(mclhip_ctx* c) {
  if (!c) return MCLHIP_E_ARG;
  MCL_HIP(c, hipSetDevice(c->device));
  return MCLHIP_OK;
}

static int gready(mclhip_ctx* c) {
  if (!c->gl_mask) return fail(c, MCLHIP_E_STATE, "no grid map: call mclhip_grid_set_map first");
  if (!c->gl_pk) return fail(c, MCLHIP_E_STATE, "no grid sensor: call mclhip_grid_set_sensor first");
  return MCLHIP_OK;
}

struct Timer {   // HIP events around one kernel group on the context's stream
  mclhip_ctx* c; int which; hipEvent_t a = nullptr, b = nullptr;
  Timer(mclhip_ctx* c_, int w) : c(c_), which(w) {
    if (!c->timing) return;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b); (void)hipEventRecord(a, c->stream);
  }
  void stop() { if (a) (void)hipEventRecord(b, c->stream); }
  void collect() {
    if (!a) return;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, a, b) == hipSuccess) { c->t_ms[which] += ms; c->t_launch[which] += 1; }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    a = b = nullptr;
  }
};

}  // namespace mcl

using namespace mcl;

#define G_TRY(expr) do { int rc__ = (expr); if (rc__ != MCLHIP_OK) return rc__; } while (0)

extern "C" {

int mclhip_grid_set_map(mclhip_ctx* c, const uint8_t* map, int H, int W, double res_x, double res_y, int obstacle_below) {
  G_TRY(guse(c));
  if (!map || H <= 0 || W <= 0) return fail(c, MCLHIP_E_ARG, "map is NULL or empty");
  if ((int64_t)H * W > ((int64_t)1 << 31) - 1) return fail(c, MCLHIP_E_LIMIT, "grid is limited to 2^31 cells");
  const size_t n = (size_t)H * W;
  // the sector query treats value 0 as obstacle and 255 as free (mc_localization.py:66, :110); here
  // a cell is an obstacle below the threshold and free at or above it (robot.py:157, :199)
  std::vector<uint8_t> remap(n), mask(n);
  std::vector<double> hx, hy;
  std::vector<int32_t> hc;
  int64_t nprior = 0;
  for (int i = 0; i < H; ++i)
    for (int j = 0; j < W; ++j) {
      const size_t k = (size_t)i * W + j;
      const bool fr = map[k] >= obstacle_below;
      remap[k] = fr ? 255 : 0;
      mask[k] = (uint8_t)((fr ? 1 : 0) | (map[k] != 0 ? 2 : 0));
      if (map[k] != 0) ++nprior;
      if (fr) {
        hx.push_back((double)j * res_x);   // j*res_x, i*res_y (robot.py:289)
        hy.push_back((double)i * res_y);
        hc.push_back((int32_t)k);
      }
    }
  G_TRY(mclhip_set_map(c, remap.data(), H, W, res_x, res_y));
  c->gl_nfree = (int64_t)hc.size();
  c->gl_nprior = nprior;
  G_TRY(galloc(c, &c->gl_mask, n));
  MCL_HIP(c, hipMemcpy(c->gl_mask, mask.data(), n, hipMemcpyHostToDevice));
  const size_t nf = hc.size();
  G_TRY(galloc(c, &c->gl_x, nf * sizeof(double)));
  G_TRY(galloc(c, &c->gl_y, nf * sizeof(double)));
  G_TRY(galloc(c, &c->gl_th, nf * sizeof(double)));
  G_TRY(galloc(c, &c->gl_cell, nf * sizeof(int32_t)));
  if (nf) {
    MCL_HIP(c, hipMemcpy(c->gl_x, hx.data(), nf * sizeof(double), hipMemcpyHostToDevice));
    MCL_HIP(c, hipMemcpy(c->gl_y, hy.data(), nf * sizeof(double), hipMemcpyHostToDevice));
    MCL_HIP(c, hipMemset(c->gl_th, 0, nf * sizeof(double)));
    MCL_HIP(c, hipMemcpy(c->gl_cell, hc.data(), nf * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  if (c->gl_pk) { (void)hipFree(c->gl_pk); c->gl_pk = nullptr; }   // belief belongs to the old map
  return MCLHIP_OK;
}

int mclhip_grid_set_sensor(mclhip_ctx* c, const double* angles_deg, int B0, int n_dir, double res_theta_deg,
                           double z_max, double sigma_hit, double lambda_short, const double w4[4], double half) {
  G_TRY(guse(c));
  if (!c->gl_mask) return fail(c, MCLHIP_E_STATE, "no grid map: call mclhip_grid_set_map first");
  if (!angles_deg || !w4 || B0 <= 0 || n_dir <= 0) return fail(c, MCLHIP_E_ARG, "angles/w4 NULL, no beams or no headings");
  if ((int64_t)B0 * n_dir > kMaxBeams) return fail(c, MCLHIP_E_LIMIT, "headings x beams is limited to 1024");
  const int NE = B0 * n_dir;
  std::vector<double> tab((size_t)NE);
  for (int d = 0; d < n_dir; ++d) {
    const double theta = (double)d * res_theta_deg * M_PI / 180.0;   // robot.py:289
    for (int k = 0; k < B0; ++k) {
      if (!std::isfinite(angles_deg[k])) return fail(c, MCLHIP_E_ARG, "beam angles must be finite");
      tab[(size_t)d * B0 + k] = angles_deg[k] * M_PI / 180.0 + theta;   // robot.py:252
    }
  }
  // Many (heading, beam) pairs point the same way: with headings every 5 degrees and beams every 15 there are
  // 72 absolute bearings for 936 pairs.  Entries whose angles agree modulo one turn to within 1e-9 rad form a
  // class; the sector query runs on the class table, and only a cell within the float32 uncertainty of a
  // sector edge (where the last bits of an entry's own angle decide) is tested entry by entry.  The
  // reference's test recognises one turn of difference and no more (robot.py:256-258): the sector of an angle
  // in [0, 2pi) is the full window around it, and so is that of an angle one turn higher or lower as long as
  // its window does not reach past 4pi or -2pi (there the reference's own window is cut off); angles beyond
  // that are refused.
  const double two_pi = M_PI * 2;
  std::vector<double> red(NE);
  for (int e = 0; e < NE; ++e) {
    if (!(tab[e] > -two_pi + half + 1e-6 && tab[e] < 2 * two_pi - half - 1e-6))
      return fail(c, MCLHIP_E_LIMIT, "heading + beam angle (and its sector) must lie within (-2pi, 4pi)");
    double u = tab[e] - two_pi * std::floor(tab[e] / two_pi);
    if (u >= two_pi) u -= two_pi;
    if (u < 0) u = 0;
    red[e] = u;
  }
  std::vector<uint16_t> order(NE);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](uint16_t a, uint16_t b) { return red[a] < red[b]; });
  std::vector<double> centre, mbeta((size_t)B0 + n_dir);
  std::vector<uint32_t> cls(NE);
  std::vector<uint16_t> first, member(NE);
  for (int m = 0; m < NE; ++m) {
    const int e = order[m];
    if (centre.empty() || red[e] - centre.back() > 1e-9) { centre.push_back(red[e]); first.push_back((uint16_t)m); }
    cls[e] = ((uint32_t)(centre.size() - 1) << 20) | ((uint32_t)(e / B0) << 10) | (uint32_t)(e % B0);
    member[m] = (uint16_t)e;
  }
  first.push_back((uint16_t)NE);
  for (int k = 0; k < B0; ++k) mbeta[k] = angles_deg[k] * M_PI / 180.0;                       // the two terms of robot.py:252
  for (int d = 0; d < n_dir; ++d) mbeta[(size_t)B0 + d] = (double)d * res_theta_deg * M_PI / 180.0;
  G_TRY(set_sensor_table(c, centre.data(), (int)centre.size(), z_max, sigma_hit, lambda_short, w4, half));
  c->gl_NE = NE;
  G_TRY(galloc(c, &c->gl_cls, NE * sizeof(uint32_t)));
  G_TRY(galloc(c, &c->gl_cls_first, first.size() * sizeof(uint16_t)));
  G_TRY(galloc(c, &c->gl_member, NE * sizeof(uint16_t)));
  G_TRY(galloc(c, &c->gl_mbeta, mbeta.size() * sizeof(double)));
  G_TRY(galloc(c, &c->gl_zt, B0 * sizeof(double)));
  G_TRY(galloc(c, &c->gl_el, B0 * sizeof(double)));
  G_TRY(galloc(c, &c->gl_zm, B0 * sizeof(double)));
  MCL_HIP(c, hipMemcpy(c->gl_cls, cls.data(), NE * sizeof(uint32_t), hipMemcpyHostToDevice));
  MCL_HIP(c, hipMemcpy(c->gl_cls_first, first.data(), first.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  MCL_HIP(c, hipMemcpy(c->gl_member, member.data(), NE * sizeof(uint16_t), hipMemcpyHostToDevice));
  MCL_HIP(c, hipMemcpy(c->gl_mbeta, mbeta.data(), mbeta.size() * sizeof(double), hipMemcpyHostToDevice));
  c->gl_D = n_dir; c->gl_B0 = B0; c->gl_res_theta = res_theta_deg;
  c->gl_angles_deg.assign(angles_deg, angles_deg + B0);
  const size_t n = (size_t)c->H * c->W * n_dir;
  G_TRY(galloc(c, &c->gl_pk, n * sizeof(float)));
  G_TRY(galloc(c, &c->gl_pk_pred, n * sizeof(double)));
  G_TRY(galloc(c, &c->gl_K, (size_t)n_dir * kOffsets * n_dir * sizeof(double)));
  G_TRY(galloc(c, &c->gl_part, ((n + kSumChunk - 1) / kSumChunk + 1) * sizeof(float)));
  MCL_HIP(c, hipMemset(c->gl_pk, 0, n * sizeof(float)));
  MCL_HIP(c, hipMemset(c->gl_pk_pred, 0, n * sizeof(double)));
  return MCLHIP_OK;
}

int mclhip_grid_init_belief(mclhip_ctx* c) {
  G_TRY(guse(c));
  G_TRY(gready(c));
  const int64_t ncell = (int64_t)c->H * c->W;
  const int64_t n = ncell * c->gl_D;
  const float value = 1.0f / (float)c->gl_nprior;   // ones(float32) /= num_free_cells (robot.py:49-59)
  hipLaunchKernelGGL(grid_prior_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->gl_mask, ncell,
                     c->gl_D, value, c->gl_pk);
  MCL_HIP(c, hipGetLastError());
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  return MCLHIP_OK;
}

int mclhip_grid_set_belief(mclhip_ctx* c, const float* pk) {
  G_TRY(guse(c));
  G_TRY(gready(c));
  if (!pk) return fail(c, MCLHIP_E_ARG, "pk is NULL");
  MCL_HIP(c, hipMemcpy(c->gl_pk, pk, (size_t)c->H * c->W * c->gl_D * sizeof(float), hipMemcpyHostToDevice));
  return MCLHIP_OK;
}

int mclhip_grid_set_prediction(mclhip_ctx* c, const double* pk_pred) {
  G_TRY(guse(c));
  G_TRY(gready(c));
  if (!pk_pred) return fail(c, MCLHIP_E_ARG, "pk_pred is NULL");
  MCL_HIP(c, hipMemcpy(c->gl_pk_pred, pk_pred, (size_t)c->H * c->W * c->gl_D * sizeof(double), hipMemcpyHostToDevice));
  return MCLHIP_OK;
}

int mclhip_grid_get_belief(mclhip_ctx* c, float* pk, double* pk_pred) {
  G_TRY(guse(c));
  G_TRY(gready(c));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  const size_t n = (size_t)c->H * c->W * c->gl_D;
  if (pk) MCL_HIP(c, hipMemcpy(pk, c->gl_pk, n * sizeof(float), hipMemcpyDeviceToHost));
  if (pk_pred) MCL_HIP(c, hipMemcpy(pk_pred, c->gl_pk_pred, n * sizeof(double), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

int mclhip_grid_predict(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6]) {
  G_TRY(guse(c));
  G_TRY(gready(c));
  if (!alpha || !(dt > 0)) return fail(c, MCLHIP_E_ARG, "alpha NULL or dt <= 0");
  const int D = c->gl_D;
  GridMotion m;
  m.v = v; m.w = omega; m.dt = dt;
  m.b1 = alpha[0] * std::fabs(v) + alpha[1] * std::fabs(omega);   // robot.py:118-120
  m.b2 = alpha[2] * std::fabs(v) + alpha[3] * std::fabs(omega);
  m.b3 = alpha[4] * std::fabs(v) + alpha[5] * std::fabs(omega);
  m.res_x = c->res_x; m.res_y = c->res_y; m.res_theta = c->gl_res_theta;
  const int DT = (D % 8 == 0) ? 8 : (D % 4 == 0) ? 4 : 1;
  Timer tm(c, 3);
  const int nk = D * kOffsets * D;
  hipLaunchKernelGGL(grid_coeff_kernel, dim3((nk + 255) / 256), dim3(256), 0, c->stream, m, D, DT, DT, c->gl_K);
  dim3 grid((c->W + 16 * GRID_CX - 1) / (16 * GRID_CX), (c->H + 15) / 16, D / DT), block(256);
  if (DT == 8)
    hipLaunchKernelGGL((grid_predict_kernel<8, 8>), grid, block, 0, c->stream, c->gl_pk, c->gl_K, c->gl_mask, c->gl_pk_pred, c->H, c->W, D);
  else if (DT == 4)
    hipLaunchKernelGGL((grid_predict_kernel<4, 4>), grid, block, 0, c->stream, c->gl_pk, c->gl_K, c->gl_mask, c->gl_pk_pred, c->H, c->W, D);
  else
    hipLaunchKernelGGL((grid_predict_kernel<1, 1>), grid, block, 0, c->stream, c->gl_pk, c->gl_K, c->gl_mask, c->gl_pk_pred, c->H, c->W, D);
  tm.stop();
  MCL_HIP(c, hipGetLastError());
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  tm.collect();
  return MCLHIP_OK;
}

int mclhip_grid_update(mclhip_ctx* c, const double* z_meas, float* sum_out) {
  G_TRY(guse(c));
  G_TRY(gready(c));
  if (!z_meas) return fail(c, MCLHIP_E_ARG, "z_meas is NULL");
  G_TRY(build_ring_table(c));
  const int D = c->gl_D, B0 = c->gl_B0;
  MCL_HIP(c, hipMemcpy(c->gl_zm, z_meas, B0 * sizeof(double), hipMemcpyHostToDevice));
  const int64_t n = (int64_t)c->H * c->W * D;
  Timer tm(c, 4);
  G_TRY(launch_grid_beam_table(c, c->gl_zm, B0, B0, c->gl_zt, c->gl_el));
  MCL_HIP(c, hipMemsetAsync(c->gl_pk, 0, n * sizeof(float), c->stream));   // cells below the threshold (robot.py:302)
  G_TRY(launch_grid_update(c, c->gl_x, c->gl_y, c->gl_th, c->gl_cell, c->gl_nfree, D, B0, c->gl_pk_pred, c->gl_pk));
  const int64_t nchunks = (n + kSumChunk - 1) / kSumChunk;
  hipLaunchKernelGGL(chunk_sum_f32_kernel, dim3((unsigned)nchunks), dim3(512), 0, c->stream, c->gl_pk, n, c->gl_part);
  hipLaunchKernelGGL(total_sum_f32_kernel, dim3(1), dim3(1024), 0, c->stream, c->gl_part, nchunks, c->gl_part + nchunks);
  hipLaunchKernelGGL(divide_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->gl_pk, n,
                     c->gl_part + nchunks);
  tm.stop();
  MCL_HIP(c, hipGetLastError());
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  tm.collect();
  if (sum_out) MCL_HIP(c, hipMemcpy(sum_out, c->gl_part + nchunks, sizeof(float), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

}  // extern "C"
