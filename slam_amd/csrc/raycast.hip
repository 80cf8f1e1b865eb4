// Sector range query + beam likelihood: one 64-lane wavefront per particle.
//
// What is computed (mc_localization.py:116-171, 235-240): for every beam k the
// squared distance to the nearest obstacle-cell centre with r2 <= z_max^2 whose
// bearing a = atan2(dy, dx) (wrapped to [0, 2pi)) satisfies
//     |beta_k - a| <= t   or   2pi - t <= |beta_k - a| <= 2pi + t,
// beta_k = deg_k*pi/180 + theta (never wrapped), then z_k = sqrt(min r2) and the
// weight prod_k p(z_k | z*_k) * pm.
//
// How: the wave scans the obstacle bit plane over the particle's window
// (one 8x8 tile per lane per pass), compacts the set bits into a per-wave LDS
// list, and evaluates 64 obstacle cells at a time: exact float64 dx, dy, r2;
// a float32 bearing decides which beams' sectors the cell falls into, and only
// where that bearing is within 1e-5 rad of a sector edge is the float64 atan2
// evaluated.  Each (cell, beam) match is a 64-bit LDS atomic min on the bit
// pattern of r2 (non-negative doubles order like unsigned integers).  The beam
// table (sorted angles, bucket index, measured ranges, short-reading constants)
// is staged in LDS once per block.
//
// Compiled with -ffp-contract=off.
#include "mclhip_internal.h"

namespace mcl {

constexpr int kListCap = 576;       // per-wave obstacle list: 512 per round + a carried remainder < 64
constexpr int kTakeMax = 8;         // set bits a lane contributes per round
constexpr double kEdgeEps = 1e-5;   // float32 bearing is trusted farther than this from a sector edge
constexpr double kTwoPi = 6.283185307179586;  // == np.pi*2.0
constexpr int kWavesPerBlock = 4;

struct WaveLds {
  unsigned long long* rmin;  // [B] bit pattern of the smallest r2 per beam
  unsigned int* hit;         // [B] row-major cell index of that cell (stage mode)
  unsigned int* list;        // [kListCap] packed (row << 16 | col)
};

__device__ inline int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int o = __shfl_up(v, d, 64);
    if (lane >= d) v += o;
  }
  return v;
}

// 1 = inside a sector, 0 = outside, 2 = too close to an edge for the float32 bearing
__device__ inline int classify(double d, double half, double lo2, double hi2) {
  if (d <= half - kEdgeEps) return 1;
  if (d < half + kEdgeEps) return 2;
  if (d < lo2 - kEdgeEps) return 0;
  if (d <= lo2 + kEdgeEps) return 2;
  if (d < hi2 - kEdgeEps) return 1;
  if (d <= hi2 + kEdgeEps) return 2;
  return 0;
}

template <typename T, bool STAGE, bool PRODUCT>
__global__ __launch_bounds__(256) void raycast_kernel(RayParams P, const T* __restrict__ xs, const T* __restrict__ ys,
                                                      const T* __restrict__ ths, const uint8_t* __restrict__ pm,
                                                      int64_t n, double* __restrict__ w_out,
                                                      double* __restrict__ logw_out, double* __restrict__ ranges_out,
                                                      int32_t* __restrict__ hits_out) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int B = P.B, NB = P.NB;
  // block-wide beam table
  double* s_ts = reinterpret_cast<double*>(smem);
  double* s_to = s_ts + B;
  double* s_zt = s_to + B;
  double* s_el = s_zt + B;
  unsigned long long* s_rmin_all = reinterpret_cast<unsigned long long*>(s_el + B);
  unsigned int* s_hit_all = reinterpret_cast<unsigned int*>(s_rmin_all + (size_t)kWavesPerBlock * B);
  unsigned int* s_list_all = s_hit_all + (size_t)kWavesPerBlock * B;
  unsigned short* s_perm = reinterpret_cast<unsigned short*>(s_list_all + (size_t)kWavesPerBlock * kListCap);
  unsigned short* s_lut = s_perm + B;

  for (int k = threadIdx.x; k < B; k += blockDim.x) {
    s_ts[k] = P.tab_sorted[k];
    s_to[k] = P.tab_orig[k];
    s_perm[k] = P.perm[k];
    if (!STAGE) {
      s_zt[k] = P.zt[k];
      s_el[k] = P.etalam[k];
    }
  }
  for (int k = threadIdx.x; k < NB; k += blockDim.x) s_lut[k] = P.lut[k];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wv = threadIdx.x >> 6;
  WaveLds L;
  L.rmin = s_rmin_all + (size_t)wv * B;
  L.hit = s_hit_all + (size_t)wv * B;
  L.list = s_list_all + (size_t)wv * kListCap;

  const unsigned long long r2max_bits = (unsigned long long)__double_as_longlong(P.r2max);
  const float cull_r = (float)(P.z_max + 2.0 * (P.res_x + P.res_y));
  const float cull_r2 = cull_r * cull_r;
  const double neg_half_inv_s2 = -0.5 / P.sigma2;

  for (int64_t p = (int64_t)blockIdx.x * kWavesPerBlock + wv; p < n; p += (int64_t)gridDim.x * kWavesPerBlock) {
    const double x = (double)xs[p], y = (double)ys[p], th = (double)ths[p];
    for (int k = lane; k < B; k += 64) {
      L.rmin[k] = r2max_bits;
      if (STAGE) L.hit[k] = 0xffffffffu;
    }
    // window of tiles that can hold a cell within z_max (two cells of slack: cell
    // coordinates are float32 products and differ from col*res by < 1e-3 cell)
    int tx0 = 0, ty0 = 0, ntx = 0, nty = 0;
    {
      double fx0 = (x - P.z_max) / P.res_x - 2.0, fx1 = (x + P.z_max) / P.res_x + 2.0;
      double fy0 = (y - P.z_max) / P.res_y - 2.0, fy1 = (y + P.z_max) / P.res_y + 2.0;
      // comparisons are false for NaN, leaving the window empty
      if (fx1 >= 0.0 && fx0 <= (double)(P.W - 1) && fy1 >= 0.0 && fy0 <= (double)(P.H - 1)) {
        int j0 = fx0 < 0.0 ? 0 : (int)fx0;
        int j1 = fx1 > (double)(P.W - 1) ? P.W - 1 : (int)fx1;
        int i0 = fy0 < 0.0 ? 0 : (int)fy0;
        int i1 = fy1 > (double)(P.H - 1) ? P.H - 1 : (int)fy1;
        tx0 = j0 >> 3;
        ty0 = i0 >> 3;
        ntx = (j1 >> 3) - tx0 + 1;
        nty = (i1 >> 3) - ty0 + 1;
      }
    }
    const int nt = ntx * nty;
    const float xf = (float)x, yf = (float)y;

    for (int phase = 0; phase < (STAGE ? 2 : 1); ++phase) {
      int count = 0;  // wave-uniform number of list entries
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

      // evaluates the first `upto` list entries that form whole batches (or all of them when final)
      auto drain = [&](bool final_pass) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int b = 0;
        while (final_pass ? (b < count) : (b + 64 <= count)) {
          const bool valid = (b + lane) < count;
          unsigned int e = valid ? L.list[b + lane] : 0u;
          const int col = (int)(e & 0xffffu), row = (int)(e >> 16);
          const double ox = (double)((float)col * P.res_xf);  // mc_localization.py:68-71: float32 products
          const double oy = (double)((float)row * P.res_yf);
          const double dx = ox - x, dy = oy - y;              // :124-125, float64
          const double r2 = dx * dx + dy * dy;                // :126
          if (valid && r2 <= P.r2max) {                       // :127
            float af = atan2f((float)dy, (float)dx);
            if (af < 0.0f) af += 6.2831855f;
            const double aa = (double)af;
            const double phi = aa - th;
            double a_exact = 0.0;
            bool have_exact = false;
            const unsigned long long r2bits = (unsigned long long)__double_as_longlong(r2);
            const unsigned int cell = (unsigned int)row * (unsigned int)P.W + (unsigned int)col;
#pragma unroll 1
            for (int c = -1; c <= 1; ++c) {
              const double centre = phi + (double)c * kTwoPi;
              const double lo = centre - P.half - 2.0 * kEdgeEps;
              const double hi = centre + P.half + 2.0 * kEdgeEps;
              if (!(hi >= P.ts_min && lo <= P.ts_max)) continue;
              int k = 0;
              if (lo > P.ts_min) {
                int bkt = (int)((lo - P.ts_min) * P.inv_bw) - 1;
                bkt = bkt < 0 ? 0 : (bkt > NB - 1 ? NB - 1 : bkt);
                k = s_lut[bkt];
              }
#pragma unroll 1
              for (; k < B; ++k) {
                const double tk = s_ts[k];
                if (tk < lo) continue;
                if (tk > hi) break;
                const int ko = s_perm[k];
                const double beta = s_to[ko] + th;  // :143
                int verdict = classify(fabs(beta - aa), P.half, P.two_pi_lo, P.two_pi_hi);
                if (verdict == 2) {
                  if (!have_exact) {
                    a_exact = atan2(dy, dx);                 // :133
                    if (a_exact < 0.0) a_exact += kTwoPi;    // :134-135
                    have_exact = true;
                  }
                  const double de = fabs(beta - a_exact);    // :148
                  verdict = (de <= P.half) || (de >= P.two_pi_lo && de <= P.two_pi_hi);  // :149
                }
                if (verdict == 1) {
                  if (!STAGE || phase == 0) {
                    atomicMin(&L.rmin[ko], r2bits);
                  } else if (L.rmin[ko] == r2bits) {
                    atomicMin(&L.hit[ko], cell);  // lowest row-major index on equal r2
                  }
                }
              }
            }
          }
          b += 64;
        }
        // carry the incomplete batch to the front of the list
        const int rem = count - b;
        if (rem > 0) {
          unsigned int e = (lane < rem) ? L.list[b + lane] : 0u;
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          if (lane < rem) L.list[lane] = e;
          count = rem;
        } else {
          count = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      };

      for (int base = 0; base < nt; base += 64) {
        const int t = base + lane;
        unsigned long long bits = 0ull;
        int ty = 0, tx = 0;
        if (t < nt) {
          ty = ty0 + t / ntx;
          tx = tx0 + t % ntx;
          // nearest point of the tile's cell-centre rectangle to the particle
          const float cx0 = (float)(tx * 8) * P.res_xf, cx1 = (float)(tx * 8 + 7) * P.res_xf;
          const float cy0 = (float)(ty * 8) * P.res_yf, cy1 = (float)(ty * 8 + 7) * P.res_yf;
          const float ddx = xf < cx0 ? cx0 - xf : (xf > cx1 ? xf - cx1 : 0.0f);
          const float ddy = yf < cy0 ? cy0 - yf : (yf > cy1 ? yf - cy1 : 0.0f);
          if (ddx * ddx + ddy * ddy <= cull_r2) bits = P.obst[(size_t)ty * P.TW + tx];
        }
        while (__ballot(bits != 0ull) != 0ull) {
          const int pc = __popcll(bits);
          const int take = pc < kTakeMax ? pc : kTakeMax;
          const int incl = wave_incl_scan(take, lane);
          const int tot = __shfl(incl, 63, 64);
          if (count + tot > kListCap) drain(false);
          int pos = count + incl - take;
          for (int q = 0; q < take; ++q) {
            const int bit = __ffsll((long long)bits) - 1;
            bits &= bits - 1ull;
            const unsigned int row = (unsigned int)(ty * 8 + (bit >> 3));
            const unsigned int col = (unsigned int)(tx * 8 + (bit & 7));
            L.list[pos++] = (row << 16) | col;
          }
          count += tot;
        }
      }
      drain(true);
    }

    // ranges -> beam mixture -> weight
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (STAGE) {
      for (int k = lane; k < B; k += 64) {
        const double r2 = __longlong_as_double((long long)L.rmin[k]);
        ranges_out[p * B + k] = sqrt(r2);  // :160
        const unsigned int h = L.hit[k];
        hits_out[p * B + k] = (h == 0xffffffffu) ? -1 : (int32_t)h;
      }
    } else {
      typedef T R;  // likelihood arithmetic follows the state dtype
      double ls = 0.0;
      for (int k = lane; k < B; k += 64) {
        const double r2 = __longlong_as_double((long long)L.rmin[k]);
        const double zd = sqrt(r2);
        const double ztd = s_zt[k];
        double pkd;
        if constexpr (sizeof(R) == 4) {
          // float32 mode: the two exponentials in float32, the mixture itself with float64
          // constants (rounding them to float32 would bias every beam's log the same way)
          // exponents formed in float64 (a float32 sigma^2 would scale every exponent the same way)
          const double d = zd - ztd;
          const float e_hit = exp((float)((d * d) * neg_half_inv_s2));
          const float e_short = exp((float)(-P.lambda * zd));
          pkd = (double)e_hit * (P.c_hit * P.w_hit) + ((zd <= ztd) ? (double)e_short * (s_el[k] * P.w_short) : 0.0) +
                ((fabs(zd - P.z_max) < 1e-6) ? P.w_max : 0.0) + P.p_rand * P.w_rand;
          ls += (double)log((float)pkd);
        } else {
          // range_sensor.py:31-67, operation by operation; 0 <= z <= z_max holds by construction
          const double d = zd - ztd;
          const double p_hit = P.c_hit * exp(-0.5 * (d * d) / P.sigma2);
          const double p_short = (zd <= ztd) ? s_el[k] * exp(-P.lambda * zd) : 0.0;
          const double p_max = (fabs(zd - P.z_max) < 1e-6) ? 1.0 : 0.0;
          pkd = p_hit * P.w_hit + p_short * P.w_short + p_max * P.w_max + P.p_rand * P.w_rand;
          ls += log(pkd);
        }
        if (PRODUCT) L.rmin[k] = (unsigned long long)__double_as_longlong(pkd);
      }
      double lsd = ls;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) lsd += __shfl_xor(lsd, d, 64);
      double q = 1.0;
      if (PRODUCT) {
        // q *= p_k strictly in beam order (mc_localization.py:235-238): a running product that
        // underflows on the way stays zero, as it does in the reference
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        for (int k = 0; k < B; ++k) q = q * __longlong_as_double((long long)L.rmin[k]);
      }
      if (lane == 0) {
        const bool fr = pm[p] != 0;
        if (PRODUCT) w_out[p] = fr ? q : q * 0.0;  // q * pm (mc_localization.py:240)
        logw_out[p] = fr ? lsd : -INFINITY;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

static size_t ray_lds_bytes(int B, int NB) {
  size_t s = (size_t)4 * B * sizeof(double);
  s += (size_t)kWavesPerBlock * B * sizeof(unsigned long long);
  s += (size_t)kWavesPerBlock * B * sizeof(unsigned int);
  s += (size_t)kWavesPerBlock * kListCap * sizeof(unsigned int);
  s += (size_t)B * sizeof(unsigned short) + (size_t)NB * sizeof(unsigned short);
  return (s + 15) & ~(size_t)15;
}

int launch_raycast(mclhip_ctx* c, const void* x, const void* y, const void* th, const uint8_t* pm, int64_t n,
                   int dtype, double* w_out, double* logw_out, double* ranges_out, int32_t* hits_out) {
  const bool product = w_out != nullptr;
  if (n <= 0) return MCLHIP_OK;
  RayParams P = make_ray_params(c);
  const size_t lds = ray_lds_bytes(P.B, P.NB);
  if (lds > 64 * 1024) return fail(c, MCLHIP_E_LIMIT, "beam table does not fit the kernel's LDS budget");
  int64_t blocks = (n + kWavesPerBlock - 1) / kWavesPerBlock;
  const int64_t max_blocks = 256 * 8 * 4;  // a few waves per SIMD; each wave loops over particles
  if (blocks > max_blocks) blocks = max_blocks;
  dim3 grid((unsigned)blocks), block(64 * kWavesPerBlock);
  const bool stage = ranges_out != nullptr;
  if (stage) {
    if (dtype != MCLHIP_F64) return fail(c, MCLHIP_E_ARG, "stage query takes float64 poses");
    hipLaunchKernelGGL((raycast_kernel<double, true, false>), grid, block, lds, c->stream, P, (const double*)x,
                       (const double*)y, (const double*)th, pm, n, w_out, logw_out, ranges_out, hits_out);
  } else {
#define MCL_RAY(TT, PP)                                                                                         \
  hipLaunchKernelGGL((raycast_kernel<TT, false, PP>), grid, block, lds, c->stream, P, (const TT*)x, (const TT*)y, \
                     (const TT*)th, pm, n, w_out, logw_out, ranges_out, hits_out)
    if (dtype == MCLHIP_F32) {
      if (product) MCL_RAY(float, true); else MCL_RAY(float, false);
    } else {
      if (product) MCL_RAY(double, true); else MCL_RAY(double, false);
    }
#undef MCL_RAY
  }
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

}  // namespace mcl
