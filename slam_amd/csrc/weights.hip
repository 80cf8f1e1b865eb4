// Weight normalisation in numpy's summation order, and low-variance resampling
// whose indices equal those of the reference's strictly sequential float64
// walk (mc_localization.py:242-262) while being computed in parallel.
//
// Compiled with -ffp-contract=off.
#include "mclhip_internal.h"
#include "libm_bits.h"

namespace mcl {

// ===========================================================================
// np.sum order.  numpy reduces a contiguous float64 vector in chunks of 8192;
// inside a chunk it sums pairwise: <8 serial; <=128: eight strided accumulators
// combined ((0+1)+(2+3))+((4+5)+(6+7)) plus a serial tail; else split at n/2
// rounded down to a multiple of 8.  Chunk results are added left to right.
// ===========================================================================
template <int DEPTH>
__device__ double pairwise(const double* a, int n) {
  if (n < 8) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s = s + a[i];
    return s;
  }
  if (n <= 128) {
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
    }
    double s = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) s = s + a[i];
    return s;
  }
  int h = n / 2;
  h -= h % 8;
  if constexpr (DEPTH > 0) {
    return pairwise<DEPTH - 1>(a, h) + pairwise<DEPTH - 1>(a + h, n - h);
  } else {
    return 0.0;  // unreachable for n <= 8192
  }
}

// One block of 512 threads per 8192-element chunk.  Thread (leaf, j) owns
// accumulator j of a 128-element leaf; xor-butterflies over lane bits 0..2
// combine the eight accumulators as ((0+1)+(2+3))+((4+5)+(6+7)), bits 3..5
// combine the eight leaves a wave holds, and the eight wave results are
// combined in the same left/right order -- numpy's recursion tree for n = 8192.
// FROM_LOG: the weights are formed on the way in, w = exp(logw - max logw) (0 when every log-weight is -inf),
// and written out for the kernels that follow.
template <bool FROM_LOG>
__global__ __launch_bounds__(512) void chunk_sum_kernel(double* __restrict__ w, int64_t n, double* __restrict__ chunk_sums,
                                                        const double* __restrict__ logw, const double* __restrict__ scal) {
  __shared__ double wave_sum[8];
  const int64_t base = (int64_t)blockIdx.x * kSumChunk;
  const int cnt = (int)((n - base) < kSumChunk ? (n - base) : kSumChunk);
  const double mx = FROM_LOG ? scal[1] : 0.0;
  auto weight = [&](int64_t i) -> double {
    if constexpr (FROM_LOG) {
      const double v = (mx == -INFINITY) ? 0.0 : exp_nonpos(logw[i] - mx);
      w[i] = v;
      return v;
    } else {
      return w[i];
    }
  };
  if (cnt == kSumChunk) {
    const int leaf = threadIdx.x >> 3, j = threadIdx.x & 7;
    const int64_t a = base + leaf * 128 + j;
    double r = weight(a);
#pragma unroll
    for (int g = 1; g < 16; ++g) r = r + weight(a + 8 * g);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) r = r + __shfl_xor(r, d, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0)
      chunk_sums[blockIdx.x] = ((wave_sum[0] + wave_sum[1]) + (wave_sum[2] + wave_sum[3])) +
                               ((wave_sum[4] + wave_sum[5]) + (wave_sum[6] + wave_sum[7]));
  } else {
    if constexpr (FROM_LOG) {
      for (int i = threadIdx.x; i < cnt; i += blockDim.x) (void)weight(base + i);
      __syncthreads();                       // the block's own stores are visible to its thread 0
    }
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = pairwise<7>(w + base, cnt);
  }
}

__global__ void total_sum_kernel(const double* __restrict__ chunk_sums, int nchunks, double* __restrict__ scal) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double t = 0.0;
  for (int i = 0; i < nchunks; ++i) t = t + chunk_sums[i];
  scal[0] = t;
}

__global__ void divide_kernel(double* __restrict__ w, int64_t n, const double* __restrict__ scal) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = w[i] / scal[0];  // mc_localization.py:248
}

// the same with the total formed on the spot: every block adds the chunk sums left to right (numpy's order across
// chunks; a few hundred additions at most), block 0 leaves the total in scal[0] for the caller
__global__ __launch_bounds__(256) void total_divide_kernel(double* __restrict__ w, int64_t n,
                                                           const double* __restrict__ chunk_sums, int nchunks,
                                                           double* __restrict__ scal) {
  __shared__ double s_chunk[1024];
  __shared__ double s_total;
  for (int i = threadIdx.x; i < nchunks; i += blockDim.x) s_chunk[i] = chunk_sums[i];   // one parallel read ...
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < nchunks; ++i) t = t + s_chunk[i];                                // ... one serial chain of additions
    s_total = t;
    if (blockIdx.x == 0) scal[0] = t;
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = w[i] / s_total;  // mc_localization.py:248
}

// log-domain weights: w = exp(logw - max logw)
__global__ __launch_bounds__(256) void max_partial_kernel(const double* __restrict__ v, int64_t n,
                                                          double* __restrict__ part) {
  __shared__ double sm[4];
  double m = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmax(m, v[i]);
  for (int d = 32; d >= 1; d >>= 1) m = fmax(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
}

__global__ __launch_bounds__(256) void max_final_kernel(const double* __restrict__ part, int np,
                                                        double* __restrict__ scal, double* __restrict__ also = nullptr) {
  __shared__ double sm[4];
  double m = -INFINITY;
  for (int i = threadIdx.x; i < np; i += blockDim.x) m = fmax(m, part[i]);
  for (int d = 32; d >= 1; d >>= 1) m = fmax(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
    scal[1] = t;
    if (also) also[0] = t;       // (the caller's own place for it)
  }
}


int launch_normalise(mclhip_ctx* c, int mode) {
  const int64_t n = c->M;
  double* w = c->w[c->cur];
  const unsigned g = (unsigned)((n + 255) / 256);
  const int nchunks = (int)((n + kSumChunk - 1) / kSumChunk);
  if (mode == MCLHIP_W_LOG) {
    const int np = 1024;
    hipLaunchKernelGGL(max_partial_kernel, dim3(np), dim3(256), 0, c->stream, c->logw, n, (double*)c->blk_a);
    hipLaunchKernelGGL(max_final_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->blk_a, np, c->scal);
    hipLaunchKernelGGL(chunk_sum_kernel<true>, dim3(nchunks), dim3(512), 0, c->stream, w, n, c->chunk_sums,
                       (const double*)c->logw, (const double*)c->scal);
  } else {
    hipLaunchKernelGGL(chunk_sum_kernel<false>, dim3(nchunks), dim3(512), 0, c->stream, w, n, c->chunk_sums,
                       (const double*)nullptr, (const double*)nullptr);
  }
  if (nchunks <= 1024) {
    hipLaunchKernelGGL(total_divide_kernel, dim3(g), dim3(256), 0, c->stream, w, n, (const double*)c->chunk_sums, nchunks,
                       c->scal);
  } else {   // very many chunks: one pass over them, not one per block
    hipLaunchKernelGGL(total_sum_kernel, dim3(1), dim3(64), 0, c->stream, c->chunk_sums, nchunks, c->scal);
    hipLaunchKernelGGL(divide_kernel, dim3(g), dim3(256), 0, c->stream, w, n, c->scal);
  }
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// ===========================================================================
// Device-wide inclusive scans (three kernels: tile reduce, scan of the tile
// sums by one block, tile scan + offset).  Tiles are 2048 elements.
// ===========================================================================
struct KF {  // integer increment and irregular-element flag, scanned together
  unsigned long long k;
  unsigned int f;
};
__device__ inline KF operator+(const KF& a, const KF& b) { return KF{a.k + b.k, a.f + b.f}; }
__device__ inline double shfl_up_v(double v, int d) { return __shfl_up(v, d, 64); }
__device__ inline KF shfl_up_v(const KF& v, int d) {
  return KF{(unsigned long long)__shfl_up((long long)v.k, d, 64), (unsigned int)__shfl_up((int)v.f, d, 64)};
}
__device__ inline unsigned int shfl_up_v(unsigned int v, int d) { return (unsigned int)__shfl_up((int)v, d, 64); }
__device__ inline double zero_of(double*) { return 0.0; }
__device__ inline KF zero_of(KF*) { return KF{0ull, 0u}; }
__device__ inline unsigned int zero_of(unsigned int*) { return 0u; }

// inclusive scan of one value per thread across a 256-thread block; returns the block total
template <typename V>
__device__ inline V block_incl_scan(V v, V* total) {
  __shared__ V wave_tot[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    V o = shfl_up_v(v, d);
    if (lane >= d) v = o + v;
  }
  if (lane == 63) wave_tot[wv] = v;
  __syncthreads();
  V off = zero_of((V*)nullptr);
  for (int i = 0; i < wv; ++i) off = off + wave_tot[i];
  V tot = ((wave_tot[0] + wave_tot[1]) + wave_tot[2]) + wave_tot[3];
  __syncthreads();
  *total = tot;
  return off + v;
}

template <typename V, typename Loader>
__global__ __launch_bounds__(kScanBlock) void scan_reduce_kernel(Loader ld, int64_t n, V* __restrict__ blk) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  V acc = zero_of((V*)nullptr);
#pragma unroll
  for (int j = 0; j < kScanItems; ++j)
    if (base + j < n) acc = acc + ld(base + j);
  V tot;
  block_incl_scan<V>(acc, &tot);
  if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}

// exclusive scan of the tile sums, in place, by a single block
template <typename V>
__global__ __launch_bounds__(kScanBlock) void scan_blocks_kernel(V* __restrict__ blk, int nblk) {
  __shared__ V incl_all[kScanBlock];
  V carry = zero_of((V*)nullptr);
  for (int base = 0; base < nblk; base += kScanBlock) {
    const int i = base + threadIdx.x;
    V v = (i < nblk) ? blk[i] : zero_of((V*)nullptr);
    V tot;
    V inc = block_incl_scan<V>(v, &tot);
    incl_all[threadIdx.x] = inc;
    __syncthreads();
    if (i < nblk) blk[i] = (threadIdx.x == 0) ? carry : carry + incl_all[threadIdx.x - 1];
    carry = carry + tot;
    __syncthreads();
  }
}

template <typename V, typename Loader, typename Storer>
__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(Loader ld, Storer st, int64_t n,
                                                                const V* __restrict__ blk) {
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  V item[kScanItems];
  V acc = zero_of((V*)nullptr);
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    item[j] = (base + j < n) ? ld(base + j) : zero_of((V*)nullptr);
    acc = acc + item[j];
  }
  V tot;
  V inc = block_incl_scan<V>(acc, &tot);
  // exclusive prefix of this thread inside the tile: previous thread's inclusive value
  __shared__ V incl_all[kScanBlock];
  incl_all[threadIdx.x] = inc;
  __syncthreads();
  V run = blk[blockIdx.x];
  if (threadIdx.x > 0) run = run + incl_all[threadIdx.x - 1];
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    run = run + item[j];
    if (base + j < n) st(base + j, run, item[j]);
  }
}

// ---------------------------------------------------------------------------
// Loaders / storers
// ---------------------------------------------------------------------------
struct LoadF64 {
  const double* p;
  __device__ double operator()(int64_t i) const { return p[i]; }
};
struct StoreF64 {
  double* p;
  __device__ void operator()(int64_t i, double incl, double) const { p[i] = incl; }
};

// Order-exact CDF.  The reference's running sum is the strictly sequential
// chain c_i = fl(c_{i-1} + w_i).  While c stays inside one binade [2^e, 2^(e+1))
// every step rounds to the same grid u = 2^(e-52), so fl(c + w) = c + u*rint(w/u)
// whenever w/u is not exactly half-way between two integers: the chain is an
// exact INTEGER prefix sum there.  An element is "regular" when an approximate
// scan, with a rigorous error bound, proves that c before and after it lies in
// one binade, and its w/u is not a tie; all other elements ("irregular": binade
// crossings, ties, the first element, denormal-range sums) are replayed one by
// one with true float64 additions, in order, between the integer runs.
struct ClassifyLoader {
  const double* ca;  // approximate inclusive scan of the local wn
  const double* wn;
  int64_t i0;        // global index of local element 0 (0 on a single GPU)
  double off_value;  // approximate running sum of everything before local element 0 ...
  const double* off_ptr;   // ... or where to read it on the device (when not NULL)
  __device__ KF operator()(int64_t il) const {
    const double off = off_ptr ? *off_ptr : off_value;
    const double w = wn[il];
    const int64_t i = i0 + il;
    if (i == 0) return KF{0ull, 1u};
    if (w == 0.0) return KF{0ull, 0u};  // c unchanged in any binade
    const double prev = (il == 0) ? off : ca[il - 1] + off, cur = ca[il] + off;
    // |approximate - sequential| <= (i + 64) * 2^-53 * cur; twice that is used
    const double delta = (double)(i + 64) * 0x1p-52 * cur;
    const double lo = prev - delta, hi = cur + delta;
    if (!(lo >= 0x1p-1000) || !(hi < INFINITY)) return KF{0ull, 1u};
    const int e = ilogb(lo);
    // below half a grid step of the lower candidate binade the addition leaves c unchanged,
    // whichever side of a binade boundary c is on (a dominant particle parks c next to 2^0)
    if (w < ldexp(1.0, e - 53)) return KF{0ull, 0u};
    if (ilogb(hi) != e) return KF{0ull, 1u};
    const double s = ldexp(w, 52 - e);  // exact, < 2^53
    if (s - floor(s) == 0.5) return KF{0ull, 1u};
    return KF{(unsigned long long)rint(s), 0u};
  }
};

struct IrrRecord {
  int64_t i;
  double w;
  unsigned long long P;
};
struct RunBase {
  double c;               // exact running sum right after the irregular element
  unsigned long long P;   // integer prefix at that element
  int e;                  // binade of c
  int pad;
};

struct StoreKF {
  unsigned long long* ks;
  unsigned int* fs;
  IrrRecord* irr;
  const double* wn;
  int64_t i0;
  __device__ void operator()(int64_t i, const KF& incl, const KF& own) const {
    ks[i] = incl.k;
    fs[i] = incl.f;
    if (own.f) irr[incl.f - 1] = IrrRecord{i0 + i, wn[i], incl.k};
  }
};

// One wavefront walks the irregular elements in order.
__device__ inline void replay_body(const IrrRecord* __restrict__ irr, unsigned int nirr, RunBase* __restrict__ bases, int lane) {
  double c = 0.0;
  unsigned long long Pprev = 0ull;
  int e_run = -1000;
  for (unsigned int base = 0; base < nirr; base += 64) {
    const unsigned int cnt = (nirr - base) < 64u ? (nirr - base) : 64u;
    IrrRecord rec = IrrRecord{0, 0.0, 0ull};
    if ((unsigned)lane < cnt) rec = irr[base + lane];
    RunBase mine = RunBase{0.0, 0ull, -1000, 0};
    for (unsigned int j = 0; j < cnt; ++j) {
      const int64_t ii = __shfl((long long)rec.i, (int)j, 64);
      const double wj = __shfl(rec.w, (int)j, 64);
      const unsigned long long Pj = (unsigned long long)__shfl((long long)rec.P, (int)j, 64);
      const double before = c + (double)(Pj - Pprev) * ldexp(1.0, e_run - 52);  // exact
      c = (ii == 0) ? wj : before + wj;  // the one true float64 addition
      Pprev = Pj;
      e_run = (c >= 0x1p-1000 && c < INFINITY) ? ilogb(c) : -1000;
      if ((unsigned)lane == j) mine = RunBase{c, Pj, e_run, 0};
    }
    if ((unsigned)lane < cnt) bases[base + lane] = mine;
  }
}

__global__ __launch_bounds__(64) void replay_kernel(const IrrRecord* __restrict__ irr,
                                                    const unsigned int* __restrict__ fs, int64_t n,
                                                    int64_t nirr_given, RunBase* __restrict__ bases,
                                                    const unsigned long long* __restrict__ nirr_ptr) {
  const unsigned int nirr = nirr_ptr ? (unsigned int)nirr_ptr[0] : (fs ? fs[n - 1] : (unsigned int)nirr_given);
  replay_body(irr, nirr, bases, threadIdx.x);
}

__global__ void finalize_cdf_kernel(const unsigned long long* __restrict__ ks, const unsigned int* __restrict__ fs,
                                    const RunBase* __restrict__ bases, int64_t n, unsigned long long P_off,
                                    unsigned int F_off, double* __restrict__ cdf,
                                    const unsigned long long* __restrict__ off_ptr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (off_ptr) { P_off = off_ptr[1]; F_off = (unsigned int)off_ptr[2]; }   // offsets computed on the device
  const RunBase b = bases[F_off + fs[i] - 1];
  cdf[i] = b.c + (double)(P_off + ks[i] - b.P) * ldexp(1.0, b.e - 52);  // exact
}

// index(m) = number of c_i that are < U_m, U_m = r + m/M  (mc_localization.py:254-257)
// A wavefront serves 64 consecutive positions.  U ascends with m and the running sums ascend with i, so the answers of
// the wave lie between the answer for its first position and the answer for its last.  The wave finds those two
// together -- half of it probes for each, 32 places at a time: four rounds of dependent reads for a million sums
// instead of twenty per lane --, stages the sums between them in LDS and every lane finishes there.  A stretch longer
// than kSearchSpan (many particles of next to no weight between two positions) is cut in 64 by one more round of
// probes, and every lane searches its piece nine ways at a time.  Same comparisons as a plain binary search, hence
// the same indices.  (At 2^20 particles that search took 21 us, a chain of twenty dependent reads.)
constexpr int kSearchSpan = 256;
template <typename T>
__global__ __launch_bounds__(256) void search_gather_kernel(const double* __restrict__ cdf, int64_t n, double r,
                                                            const T* __restrict__ x, const T* __restrict__ y,
                                                            const T* __restrict__ th, const double* __restrict__ w,
                                                            T* __restrict__ x2, T* __restrict__ y2,
                                                            T* __restrict__ th2, double* __restrict__ w2,
                                                            int64_t* __restrict__ idx) {
  __shared__ double s_c[4][kSearchSpan];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * blockDim.x + (int64_t)wv * 64;
  if (m0 >= n) return;                                   // (the whole wave)
  const int64_t m = m0 + lane;
  const int64_t m_last = m0 + 63 < n ? m0 + 63 : n - 1;
  const double U = r + ((double)m / (double)n);
  // the two ends: first i with !(u > cdf[i]), n if none -- lanes 0..31 for the first position, 32..63 for the last
  int64_t L, H;
  {
    const int half = lane >> 5, hl = lane & 31;
    const double u = r + ((double)(half ? m_last : m0) / (double)n);
    int64_t lo = 0, hi = n;                              // the same in all lanes of a half
    while (__ballot(hi - lo > 32) != 0ull) {
      const bool open = hi - lo > 32;
      const int64_t step = (hi - lo + 31) / 32;
      const int64_t p = lo + (int64_t)hl * step;
      const bool in = open && p < hi;
      const unsigned int hit = (unsigned int)(__ballot(in && !(u > cdf[p])) >> (32 * half));
      const unsigned int valid = (unsigned int)(__ballot(in) >> (32 * half));
      if (open) {
        if (hit == 0u) {
          lo = lo + (int64_t)(31 - __clz((int)valid)) * step + 1;     // beyond the last place probed
        } else {
          const int f = __ffs((int)hit) - 1;
          hi = lo + (int64_t)f * step;
          if (f > 0) lo = lo + (int64_t)(f - 1) * step + 1;
        }
      }
    }
    const int64_t p = lo + hl;
    const unsigned int hit = (unsigned int)(__ballot(p < hi && !(u > cdf[p])) >> (32 * half));
    const int64_t ans = hit == 0u ? hi : lo + (__ffs((int)hit) - 1);
    L = __shfl((long long)ans, 0, 64);
    H = __shfl((long long)ans, 32, 64);
  }
  // every lane's answer lies in [L, H]
  int64_t lo = L, hi = H;
  if (H - L <= kSearchSpan) {
    for (int64_t k = lane; k < H - L; k += 64) s_c[wv][k] = cdf[L + k];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    int a = 0, b = (int)(H - L);
    while (a < b) {
      const int mid = (a + b) >> 1;
      if (U > s_c[wv][mid]) a = mid + 1; else b = mid;
    }
    lo = L + a;
  } else {
    {
      // 64 places across the stretch, one per lane; every lane then knows the piece between two of them that holds its answer
      const int64_t step = (H - L + 63) / 64;
      const int64_t p = L + (int64_t)lane * step;
      s_c[wv][lane] = p < H ? cdf[p] : INFINITY;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      int a = 0, b = 64;                                 // first j with !(U > s[j])
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (U > s_c[wv][mid]) a = mid + 1; else b = mid;
      }
      if (a > 0) lo = L + (int64_t)(a - 1) * step + 1;
      if (a < 64 && L + (int64_t)a * step < H) hi = L + (int64_t)a * step;
    }
    // nine ways at a time: eight reads in flight per round
    while (hi - lo > 8) {
      const int64_t step = (hi - lo + 8) / 9;
      int below = 0;
      int64_t q[8];
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        q[k] = lo + (int64_t)(k + 1) * step - 1;
        v[k] = q[k] < hi ? cdf[q[k]] : INFINITY;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) below += (U > v[k]) ? 1 : 0;       // (ascending: the places below U come first)
      int64_t nlo = lo, nhi = hi;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k == below - 1) nlo = q[k] + 1;
        if (k == below && q[k] < hi) nhi = q[k];
      }
      lo = nlo; hi = nhi;
    }
    {
      int below = 0;
      double v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = lo + k < hi ? cdf[lo + k] : INFINITY;
#pragma unroll
      for (int k = 0; k < 8; ++k) below += (U > v[k]) ? 1 : 0;
      lo = lo + below;
    }
  }
  if (m >= n) return;
  if (lo > n - 1) lo = n - 1;
  idx[m] = lo;
  x2[m] = x[lo];
  y2[m] = y[lo];
  th2[m] = th[lo];
  w2[m] = w[lo];
}

// The reference's loop by a single lane (verifier, and the path for weight
// vectors the parallel scan does not accept: NaN / infinite sums).
__global__ void serial_walk_kernel(const double* __restrict__ wn, int64_t n, double r, int64_t* __restrict__ idx) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double c = wn[0];
  int64_t i = 0;
  for (int64_t m = 0; m < n; ++m) {
    const double U = r + ((double)m / (double)n);
    while (U > c) {
      if (i == n - 1) break;
      i = i + 1;
      c = c + wn[i];
    }
    idx[m] = i;
  }
}

template <typename T>
__global__ void gather_kernel(const int64_t* __restrict__ idx, int64_t n, const T* __restrict__ x,
                              const T* __restrict__ y, const T* __restrict__ th, const double* __restrict__ w,
                              T* __restrict__ x2, T* __restrict__ y2, T* __restrict__ th2, double* __restrict__ w2) {
  int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= n) return;
  const int64_t s = idx[m];
  x2[m] = x[s];
  y2[m] = y[s];
  th2[m] = th[s];
  w2[m] = w[s];
}


#define MCL_TRY_W(expr) do { int rc__ = (expr); if (rc__ != MCLHIP_OK) return rc__; } while (0)

// c->cdf_exact[i] = the strictly sequential float64 running sum of v[0..i] (== np.cumsum), in parallel
static int build_exact_cdf(mclhip_ctx* c, const double* wn, int64_t n) {
  const unsigned g = (unsigned)((n + 255) / 256);
  const int ntile = (int)((n + kScanTile - 1) / kScanTile);
  // approximate inclusive scan of the normalised weights
  LoadF64 lf{wn};
  StoreF64 sf{c->cdf_approx};
  hipLaunchKernelGGL((scan_reduce_kernel<double, LoadF64>), dim3(ntile), dim3(kScanBlock), 0, c->stream, lf, n,
                     (double*)c->blk_a);
  hipLaunchKernelGGL((scan_blocks_kernel<double>), dim3(1), dim3(kScanBlock), 0, c->stream, (double*)c->blk_a,
                     ntile);
  hipLaunchKernelGGL((scan_apply_kernel<double, LoadF64, StoreF64>), dim3(ntile), dim3(kScanBlock), 0, c->stream,
                     lf, sf, n, (const double*)c->blk_a);
  // integer increments + irregular flags, scanned together
  ClassifyLoader cl{c->cdf_approx, wn, 0, 0.0, nullptr};
  StoreKF sk{(unsigned long long*)c->kscan, c->fscan, (IrrRecord*)c->irr, wn, 0};
  hipLaunchKernelGGL((scan_reduce_kernel<KF, ClassifyLoader>), dim3(ntile), dim3(kScanBlock), 0, c->stream, cl, n,
                     (KF*)c->blk_b);
  hipLaunchKernelGGL((scan_blocks_kernel<KF>), dim3(1), dim3(kScanBlock), 0, c->stream, (KF*)c->blk_b, ntile);
  hipLaunchKernelGGL((scan_apply_kernel<KF, ClassifyLoader, StoreKF>), dim3(ntile), dim3(kScanBlock), 0, c->stream,
                     cl, sk, n, (const KF*)c->blk_b);
  hipLaunchKernelGGL(replay_kernel, dim3(1), dim3(64), 0, c->stream, (const IrrRecord*)c->irr, c->fscan, n,
                     (int64_t)0, (RunBase*)c->bases, (const unsigned long long*)nullptr);
  hipLaunchKernelGGL(finalize_cdf_kernel, dim3(g), dim3(256), 0, c->stream, (const unsigned long long*)c->kscan,
                     c->fscan, (const RunBase*)c->bases, n, 0ull, 0u, c->cdf_exact, (const unsigned long long*)nullptr);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

template <typename T>
static int resample_typed(mclhip_ctx* c, double r, int algo) {
  const int64_t n = c->M;
  const int a = c->cur, b = 1 - c->cur;
  const double* wn = c->w[a];
  const unsigned g = (unsigned)((n + 255) / 256);
  if (algo == MCLHIP_RESAMPLE_SERIAL) {
    hipLaunchKernelGGL(serial_walk_kernel, dim3(1), dim3(64), 0, c->stream, wn, n, r, c->idx);
    hipLaunchKernelGGL(gather_kernel<T>, dim3(g), dim3(256), 0, c->stream, c->idx, n, (const T*)c->px[a],
                       (const T*)c->py[a], (const T*)c->pth[a], wn, (T*)c->px[b], (T*)c->py[b], (T*)c->pth[b],
                       c->w[b]);
  } else {
    MCL_TRY_W(build_exact_cdf(c, wn, n));
    hipLaunchKernelGGL(search_gather_kernel<T>, dim3(g), dim3(256), 0, c->stream, c->cdf_exact, n, r,
                       (const T*)c->px[a], (const T*)c->py[a], (const T*)c->pth[a], wn, (T*)c->px[b], (T*)c->py[b],
                       (T*)c->pth[b], c->w[b], c->idx);
  }
  MCL_HIP(c, hipGetLastError());
  c->cur = b;
  return MCLHIP_OK;
}

int launch_resample(mclhip_ctx* c, double r, int algo) {
  if (c->dtype == MCLHIP_F32) return resample_typed<float>(c, r, algo);
  return resample_typed<double>(c, r, algo);
}


// ===========================================================================
// The same normalisation + order-exact CDF + search as above in six launches instead of thirteen, for the fused
// update (mclhip_step*): at 2^20 particles the thirteen kernels are microseconds each and a third of their time is
// the gaps between them.  Same arithmetic where it is observable: the sum in numpy's order, w / sum, the strictly
// sequential running sum -- so weights and indices are bit-equal to those of mclhip_normalise + mclhip_resample.
//   1  max_partial_kernel            (log-weights only)
//   2  chunk_tile_sum_kernel         w = exp(logw - max); exact sum per 8192-chunk, approximate sum per 2048-tile
//   3  normalise_classify_kernel     total, w /= total, approximate scan inside the tile, classification, tile aggregates
//   4  records_kernel                integer prefix + record count per element, irregular records; the block that
//                                    finishes last replays the records (the one serial step)
//   5  finalize_cdf_kernel
//   6  search_gather_kernel
// ===========================================================================
constexpr int kMaxFusedChunks = 1024;   // chunk sums are added one by one by every block: beyond this the unfused path is used

template <bool FROM_LOG>
__global__ __launch_bounds__(512) void chunk_tile_sum_kernel(double* __restrict__ w, int64_t n,
                                                             double* __restrict__ chunk_sums,
                                                             double* __restrict__ tile_sums,
                                                             const double* __restrict__ logw,
                                                             const double* __restrict__ max_part, int np,
                                                             double* __restrict__ scal) {
  __shared__ double wave_sum[8];
  __shared__ double s_max;
  const int64_t base = (int64_t)blockIdx.x * kSumChunk;
  const int cnt = (int)((n - base) < kSumChunk ? (n - base) : kSumChunk);
  double mx = 0.0;
  if constexpr (FROM_LOG) {
    double m = -INFINITY;
    for (int i = threadIdx.x; i < np; i += blockDim.x) m = fmax(m, max_part[i]);
    for (int d = 32; d >= 1; d >>= 1) m = fmax(m, __shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = wave_sum[0];
      for (int k = 1; k < 8; ++k) t = fmax(t, wave_sum[k]);
      s_max = t;
      if (blockIdx.x == 0) scal[1] = t;
    }
    __syncthreads();
    mx = s_max;
    __syncthreads();
  }
  auto weight = [&](int64_t i) -> double {
    if constexpr (FROM_LOG) {
      const double v = (mx == -INFINITY) ? 0.0 : exp_nonpos(logw[i] - mx);
      w[i] = v;
      return v;
    } else {
      return w[i];
    }
  };
  if (cnt == kSumChunk) {
    // numpy's pairwise tree for n = 8192 (see chunk_sum_kernel); its nodes of 2048 elements are the tile sums
    const int leaf = threadIdx.x >> 3, j = threadIdx.x & 7;
    const int64_t a = base + leaf * 128 + j;
    double r = weight(a);
#pragma unroll
    for (int g = 1; g < 16; ++g) r = r + weight(a + 8 * g);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) r = r + __shfl_xor(r, d, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
      const double t0 = wave_sum[0] + wave_sum[1], t1 = wave_sum[2] + wave_sum[3];
      const double t2 = wave_sum[4] + wave_sum[5], t3 = wave_sum[6] + wave_sum[7];
      chunk_sums[blockIdx.x] = (t0 + t1) + (t2 + t3);
      double* ts = tile_sums + (size_t)blockIdx.x * 4;
      ts[0] = t0; ts[1] = t1; ts[2] = t2; ts[3] = t3;
    }
  } else {
    if constexpr (FROM_LOG) {
      for (int i = threadIdx.x; i < cnt; i += blockDim.x) (void)weight(base + i);
      __syncthreads();                       // the block's own stores are visible to its threads
    }
    if (threadIdx.x == 0) chunk_sums[blockIdx.x] = pairwise<7>(w + base, cnt);
    // tile sums of the ragged chunk: any order will do (they only feed the approximate scan)
    const int q = threadIdx.x >> 7, t = threadIdx.x & 127;   // four tiles, 128 threads each (two waves)
    double r = 0.0;
    for (int i = q * kScanTile + t; i < (q + 1) * kScanTile && i < cnt; i += 128) r = r + w[base + i];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) r = r + __shfl_xor(r, d, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = r;
    __syncthreads();
    if (threadIdx.x < 4 && (int64_t)threadIdx.x * kScanTile < cnt)
      tile_sums[(size_t)blockIdx.x * 4 + threadIdx.x] = wave_sum[2 * threadIdx.x] + wave_sum[2 * threadIdx.x + 1];
  }
}

// classification of one element of the sequential chain (see ClassifyLoader): prev / cur = approximate running sum
// before / after it
__device__ inline KF classify_element(int64_t i, double w, double prev, double cur) {
  if (i == 0) return KF{0ull, 1u};
  if (w == 0.0) return KF{0ull, 0u};
  const double delta = (double)(i + 64) * 0x1p-52 * cur;
  const double lo = prev - delta, hi = cur + delta;
  if (!(lo >= 0x1p-1000) || !(hi < INFINITY)) return KF{0ull, 1u};
  const int e = ilogb(lo);
  if (w < ldexp(1.0, e - 53)) return KF{0ull, 0u};
  if (ilogb(hi) != e) return KF{0ull, 1u};
  const double s = ldexp(w, 52 - e);
  if (s - floor(s) == 0.5) return KF{0ull, 1u};
  return KF{(unsigned long long)rint(s), 0u};
}

// What kernels 3 and 4 share: the tile's normalised weights, the approximate running sum at every element (offset =
// sum of the earlier tiles' sums / total), the classification.  Both kernels run exactly this code on exactly these
// values, so they classify alike.
struct TileScan {
  double wn[kScanItems];
  KF kf[kScanItems];
  KF local;        // inclusive scan of the thread's KF sum across the block
  KF tile_total;
};

// (shard_off, i0: what stands before local element 0 on a rank that holds a later part of the vector -- its approximate
// share of the total, and its global index)
template <bool DIVIDE>
__device__ inline void scan_tile(TileScan& T, double* __restrict__ w, int64_t n, const double* __restrict__ tile_sums,
                                 double total, double shard_off = 0.0, int64_t i0 = 0) {
  __shared__ double s_red[4];
  __shared__ double s_off;
  // un-normalised weight before this tile: every thread adds a strided share of the earlier tiles' sums
  double part = 0.0;
  for (int j = threadIdx.x; j < (int)blockIdx.x; j += kScanBlock) part = part + tile_sums[j];
  for (int d = 32; d >= 1; d >>= 1) part = part + __shfl_xor(part, d, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) s_off = shard_off + ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) / total;
  __syncthreads();
  const double off = s_off;
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
  double acc = 0.0;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    double v = 0.0;
    if (base + j < n) {
      v = w[base + j];
      if (DIVIDE) { v = v / total; w[base + j] = v; }   // mc_localization.py:248
    }
    T.wn[j] = v;
    acc = acc + v;
  }
  double tot;
  const double incl = block_incl_scan<double>(acc, &tot);
  double run = off + (incl - acc);   // approximate running sum before this thread's first element
  KF ksum = KF{0ull, 0u};
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    const double prev = run;
    run = run + T.wn[j];
    T.kf[j] = (base + j < n) ? classify_element(i0 + base + j, T.wn[j], prev, run) : KF{0ull, 0u};
    ksum = ksum + T.kf[j];
  }
  T.local = block_incl_scan<KF>(ksum, &T.tile_total);
}

// total of the chunk sums in numpy's order (left to right), by every block for itself
__device__ inline double total_of_chunks(const double* __restrict__ chunk_sums, int nchunks) {
  __shared__ double s_chunk[kMaxFusedChunks];
  __shared__ double s_total;
  for (int i = threadIdx.x; i < nchunks; i += blockDim.x) s_chunk[i] = chunk_sums[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    int i = 0;
    for (; i + 8 <= nchunks; i += 8) {        // eight reads in flight, then the eight additions in order
      double v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = s_chunk[i + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) t = t + v[j];
    }
    for (; i < nchunks; ++i) t = t + s_chunk[i];
    s_total = t;
  }
  __syncthreads();
  return s_total;
}

__global__ __launch_bounds__(kScanBlock) void normalise_classify_kernel(double* __restrict__ w, int64_t n,
                                                                        const double* __restrict__ chunk_sums, int nchunks,
                                                                        const double* __restrict__ tile_sums,
                                                                        KF* __restrict__ tile_kf, double* __restrict__ scal,
                                                                        unsigned int* __restrict__ done_counter) {
  const double total = total_of_chunks(chunk_sums, nchunks);
  if (blockIdx.x == 0 && threadIdx.x == 0) { scal[0] = total; *done_counter = 0u; }
  TileScan T;
  scan_tile<true>(T, w, n, tile_sums, total);
  if (threadIdx.x == 0) tile_kf[blockIdx.x] = T.tile_total;
}

// the same for one rank's part of the vector: the total is the sum of EVERY rank's chunk sums, added left to right in
// rank order (== np.sum of the whole vector); what the ranks before this one hold, over the total, is where this
// rank's approximate running sum starts (-> scal[5], for records_kernel<true>)
__global__ __launch_bounds__(kScanBlock) void normalise_classify_shard_kernel(double* __restrict__ w, int64_t n,
                                                                              const double* __restrict__ all_chunks, ShardLayout lay,
                                                                              const double* __restrict__ tile_sums,
                                                                              KF* __restrict__ tile_kf, double* __restrict__ scal,
                                                                              unsigned int* __restrict__ done_counter) {
  __shared__ double s_chunk[kMaxFusedChunks];
  __shared__ double s_tb[2];
  int at = 0;
  for (int r = 0; r < lay.world; ++r) {
    const double* src = all_chunks + (size_t)r * lay.cmax;
    for (int j = threadIdx.x; j < lay.nchunks[r]; j += blockDim.x) s_chunk[at + j] = src[j];
    at += lay.nchunks[r];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // strictly left to right, eight LDS reads in flight ahead of their eight additions (at eight ranks there are a
    // thousand chunk sums: one read's latency per addition would cost every block tens of microseconds)
    auto run_on = [&](double t, int from, int to) {
      int i = from;
      for (; i + 8 <= to; i += 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = s_chunk[i + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) t = t + v[j];
      }
      for (; i < to; ++i) t = t + s_chunk[i];
      return t;
    };
    int mine = 0;
    for (int r = 0; r < lay.rank; ++r) mine += lay.nchunks[r];
    const double below = run_on(0.0, 0, mine);
    const double t = run_on(below, mine, at);
    s_tb[0] = t;
    s_tb[1] = below / t;
  }
  __syncthreads();
  const double total = s_tb[0], shard_off = s_tb[1];
  if (blockIdx.x == 0 && threadIdx.x == 0) { scal[0] = total; scal[5] = shard_off; *done_counter = 0u; }
  TileScan T;
  scan_tile<true>(T, w, n, tile_sums, total, shard_off, lay.first[lay.rank]);
  if (threadIdx.x == 0) tile_kf[blockIdx.x] = T.tile_total;
}

// SHARD: one rank's part of the vector (global index of its first element i0, its approximate offset in scal[5]); the
// block that finishes last does not replay the records -- the list is not whole before every rank's part of it has
// been gathered -- but exports them: the integer total, their number, and the first `cap` of them -> out
template <bool SHARD>
__global__ __launch_bounds__(kScanBlock) void records_kernel(double* __restrict__ wn, int64_t n,
                                                             const double* __restrict__ tile_sums,
                                                             const KF* __restrict__ tile_kf, const double* __restrict__ scal,
                                                             unsigned long long* __restrict__ ks, unsigned int* __restrict__ fs,
                                                             IrrRecord* __restrict__ irr, RunBase* __restrict__ bases,
                                                             unsigned int* __restrict__ done_counter, int64_t i0,
                                                             int cap, unsigned long long* __restrict__ out) {
  __shared__ KF s_kred[4];
  __shared__ int s_last;
  const double total = scal[0];
  TileScan T;
  scan_tile<false>(T, wn, n, tile_sums, total, SHARD ? scal[5] : 0.0, SHARD ? i0 : 0);
  unsigned long long* const ktot_slot = reinterpret_cast<unsigned long long*>(const_cast<double*>(scal) + 7);   // SHARD: the integer total
  // integer prefix and record count of the earlier tiles (exact, any order)
  KF part = KF{0ull, 0u};
  for (int j = threadIdx.x; j < (int)blockIdx.x; j += kScanBlock) part = part + tile_kf[j];
  for (int d = 32; d >= 1; d >>= 1) part = part + KF{(unsigned long long)__shfl_xor((long long)part.k, d, 64), (unsigned int)__shfl_xor((int)part.f, d, 64)};
  if ((threadIdx.x & 63) == 0) s_kred[threadIdx.x >> 6] = part;
  __syncthreads();
  const KF tile_off = (s_kred[0] + s_kred[1]) + (s_kred[2] + s_kred[3]);
  KF own_sum = KF{0ull, 0u};
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) own_sum = own_sum + T.kf[j];
  KF run = tile_off + KF{T.local.k - own_sum.k, T.local.f - own_sum.f};   // exclusive prefix of this thread
  const int64_t base = (int64_t)blockIdx.x * kScanTile + (int64_t)threadIdx.x * kScanItems;
#pragma unroll
  for (int j = 0; j < kScanItems; ++j) {
    run = run + T.kf[j];
    if (base + j < n) {
      ks[base + j] = run.k;
      fs[base + j] = run.f;
      if (T.kf[j].f) {
        // the records (and their number, below) are read by another block of this launch: agent-scope stores,
        // which go through to memory, and agent-scope loads on the reading side
        IrrRecord* dst = irr + (run.f - 1);
        __hip_atomic_store(&dst->i, (SHARD ? i0 : 0) + base + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<long long*>(&dst->w), __double_as_longlong(T.wn[j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&dst->P, run.k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (base + j == n - 1) {
        __hip_atomic_store(done_counter + 1, run.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // number of records
        if constexpr (SHARD) __hip_atomic_store(ktot_slot, run.k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  // The block that finishes last walks the irregular records (the serial step of the method).  Hand-over between
  // blocks: every wave waits until its stores have left, the block meets, one lane counts the block in; the block
  // whose count comes last reads what the others stored, with loads that go past the compute unit's own cache.
  //
  // Why relaxed atomics and a wait are enough here (and an agent-scope release / acquire pair is not needed).  The
  // eight XCDs of the GPU have an L2 each; a plain store may stay dirty in the writing XCD's L2, and a release fence at
  // agent scope writes ALL of that L2's dirty lines back -- this kernel has just written 12 bytes per particle that
  // nobody shares.  But everything one block hands to another -- the records, their number, the counter -- is written
  // and read with agent-scope ATOMIC accesses, which are performed at the point all XCDs share (stores write through,
  // loads bypass lines the local L2 does not own); a store is acknowledged, and counted off vmcnt, only once it has been
  // performed there.  So: records stored (agent-scope atomic stores) -> s_waitcnt vmcnt(0) in every wave: performed ->
  // barrier: every wave of the block is past its wait -> one atomic increment, itself totally ordered with the other
  // blocks' increments at the same point.  The block that reads grid - 1 therefore counts after every other block's
  // records were performed, and its agent-scope atomic loads read them where they were performed.  The formal pair
  // (MCL_TAIL_RELACQ: the increment with acquire-release semantics at agent scope) gives the same indices and costs the
  // write-back and invalidation of an L2 per block: measured 0.02 ms slower per step at 2^20 particles.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
#ifdef MCL_TAIL_RELACQ
    const unsigned int before = __hip_atomic_fetch_add(done_counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
    const unsigned int before = __hip_atomic_fetch_add(done_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    s_last = before == gridDim.x - 1;
  }
  __syncthreads();
  if constexpr (SHARD) {
    if (!s_last) return;
    const unsigned int nrec = __hip_atomic_load(done_counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 0) {
      out[0] = __hip_atomic_load(ktot_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      out[1] = nrec;
    }
    IrrRecord* dst = reinterpret_cast<IrrRecord*>(out + 2);
    const unsigned int m = nrec < (unsigned int)cap ? nrec : (unsigned int)cap;
    for (unsigned int j = threadIdx.x; j < m; j += blockDim.x) {
      const IrrRecord* src = irr + j;
      IrrRecord rec;
      rec.i = __hip_atomic_load(&src->i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      rec.w = __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long*>(&src->w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      rec.P = __hip_atomic_load(&src->P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      dst[j] = rec;
    }
    return;
  }
  if (!s_last || threadIdx.x >= 64) return;
  // replay (see replay_kernel), by the first wavefront of the last block
  const int lane = threadIdx.x;
  const unsigned int nirr = __hip_atomic_load(done_counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  double c = 0.0;
  unsigned long long Pprev = 0ull;
  int e_run = -1000;
  for (unsigned int b0 = 0; b0 < nirr; b0 += 64) {
    const unsigned int cnt = (nirr - b0) < 64u ? (nirr - b0) : 64u;
    IrrRecord rec = IrrRecord{0, 0.0, 0ull};
    if ((unsigned)lane < cnt) {
      const IrrRecord* src = irr + b0 + lane;
      rec.i = __hip_atomic_load(&src->i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      rec.w = __longlong_as_double(__hip_atomic_load(reinterpret_cast<const long long*>(&src->w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      rec.P = __hip_atomic_load(&src->P, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    RunBase mine = RunBase{0.0, 0ull, -1000, 0};
    for (unsigned int j = 0; j < cnt; ++j) {
      const int64_t ii = __shfl((long long)rec.i, (int)j, 64);
      const double wj = __shfl(rec.w, (int)j, 64);
      const unsigned long long Pj = (unsigned long long)__shfl((long long)rec.P, (int)j, 64);
      const double before = c + (double)(Pj - Pprev) * ldexp(1.0, e_run - 52);  // exact
      c = (ii == 0) ? wj : before + wj;  // the one true float64 addition
      Pprev = Pj;
      e_run = (c >= 0x1p-1000 && c < INFINITY) ? ilogb(c) : -1000;
      if ((unsigned)lane == j) mine = RunBase{c, Pj, e_run, 0};
    }
    if ((unsigned)lane < cnt) bases[b0 + lane] = mine;
  }
}

template <typename T>
static int normalise_resample_typed(mclhip_ctx* c, int mode, double r) {
  const int64_t n = c->M;
  const int a = c->cur, b = 1 - c->cur;
  double* w = c->w[a];
  const unsigned g = (unsigned)((n + 255) / 256);
  const int nchunks = (int)((n + kSumChunk - 1) / kSumChunk);
  const int ntile = (int)((n + kScanTile - 1) / kScanTile);
  double* tile_sums = c->cdf_approx;                              // [ntile] of the [M] array: free in this path
  unsigned int* done_counter = reinterpret_cast<unsigned int*>(c->scal + 6);
  if (mode == MCLHIP_W_LOG) {
    const int np = 1024;
    hipLaunchKernelGGL(max_partial_kernel, dim3(np), dim3(256), 0, c->stream, c->logw, n, (double*)c->blk_a);
    hipLaunchKernelGGL(chunk_tile_sum_kernel<true>, dim3(nchunks), dim3(512), 0, c->stream, w, n, c->chunk_sums, tile_sums,
                       (const double*)c->logw, (const double*)c->blk_a, np, c->scal);
  } else {
    hipLaunchKernelGGL(chunk_tile_sum_kernel<false>, dim3(nchunks), dim3(512), 0, c->stream, w, n, c->chunk_sums, tile_sums,
                       (const double*)nullptr, (const double*)nullptr, 0, c->scal);
  }
  hipLaunchKernelGGL(normalise_classify_kernel, dim3(ntile), dim3(kScanBlock), 0, c->stream, w, n,
                     (const double*)c->chunk_sums, nchunks, (const double*)tile_sums, (KF*)c->blk_b, c->scal, done_counter);
  hipLaunchKernelGGL(records_kernel<false>, dim3(ntile), dim3(kScanBlock), 0, c->stream, w, n, (const double*)tile_sums,
                     (const KF*)c->blk_b, (const double*)c->scal, (unsigned long long*)c->kscan, c->fscan,
                     (IrrRecord*)c->irr, (RunBase*)c->bases, done_counter, (int64_t)0, 0, (unsigned long long*)nullptr);
  hipLaunchKernelGGL(finalize_cdf_kernel, dim3(g), dim3(256), 0, c->stream, (const unsigned long long*)c->kscan,
                     c->fscan, (const RunBase*)c->bases, n, 0ull, 0u, c->cdf_exact, (const unsigned long long*)nullptr);
  hipLaunchKernelGGL(search_gather_kernel<T>, dim3(g), dim3(256), 0, c->stream, c->cdf_exact, n, r, (const T*)c->px[a],
                     (const T*)c->py[a], (const T*)c->pth[a], (const double*)w, (T*)c->px[b], (T*)c->py[b], (T*)c->pth[b],
                     c->w[b], c->idx);
  MCL_HIP(c, hipGetLastError());
  c->cur = b;
  return MCLHIP_OK;
}

// normalise + resample for the fused update; falls back to the two separate sequences for very large sets
int launch_normalise_resample(mclhip_ctx* c, int mode, double r) {
  const int nchunks = (int)((c->M + kSumChunk - 1) / kSumChunk);
  if (nchunks > kMaxFusedChunks) {
    MCL_TRY_W(launch_normalise(c, mode));
    return launch_resample(c, r, MCLHIP_RESAMPLE_PARALLEL);
  }
  if (c->dtype == MCLHIP_F32) return normalise_resample_typed<float>(c, mode, r);
  return normalise_resample_typed<double>(c, mode, r);
}


// ===========================================================================
// One filter over several GPUs.  Every rank owns a contiguous range of the
// global particle vector; the pieces below are the per-rank halves of the
// normalisation and of the order-exact resampling, the host combines them.
// ===========================================================================
__global__ void divide_by_kernel(double* __restrict__ w, int64_t n, double total) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = w[i] / total;  // mc_localization.py:248 with the global sum
}

__global__ void exp_shift_by_kernel(const double* __restrict__ logw, double* __restrict__ w, int64_t n, double mx) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  w[i] = (mx == -INFINITY) ? 0.0 : exp_nonpos(logw[i] - mx);
}

// raw weights from log-weights: w = exp(logw) (a particle off the free cells has -inf and gets 0)
__global__ void exp_weights_kernel(const double* __restrict__ logw, double* __restrict__ w, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) w[i] = exp(logw[i]);
}

int launch_exp_weights(mclhip_ctx* c) {
  const int64_t n = c->M;
  if (n <= 0) return MCLHIP_OK;
  hipLaunchKernelGGL(exp_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->logw, c->w[c->cur], n);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

int launch_shard_max_logw(mclhip_ctx* c, double* d_also) {
  const int np = 1024;
  hipLaunchKernelGGL(max_partial_kernel, dim3(np), dim3(256), 0, c->stream, c->logw, c->M, (double*)c->blk_a);
  hipLaunchKernelGGL(max_final_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)c->blk_a, np, c->scal, d_also);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// form the raw weights (log mode: exp(logw - gmax)) and this rank's 8192-element chunk sums
int launch_shard_weights(mclhip_ctx* c, int mode, double gmax) {
  const int64_t n = c->M;
  double* w = c->w[c->cur];
  const unsigned g = (unsigned)((n + 255) / 256);
  if (mode == MCLHIP_W_LOG)
    hipLaunchKernelGGL(exp_shift_by_kernel, dim3(g), dim3(256), 0, c->stream, c->logw, w, n, gmax);
  const int nchunks = (int)((n + kSumChunk - 1) / kSumChunk);
  hipLaunchKernelGGL(chunk_sum_kernel<false>, dim3(nchunks), dim3(512), 0, c->stream, w, n, c->chunk_sums,
                     (const double*)nullptr, (const double*)nullptr);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// log-mode weights with the global maximum already in scal[1] (device-driven path)
// (the chunk sums go straight to where the caller gathers them from; the sums per 2048-tile, for the approximate scan
// of launch_dshard_scan, to c->cdf_approx; scal[1] = the global maximum)
int launch_dshard_weights(mclhip_ctx* c, const double* d_gmax, double* d_chunks_out) {
  const int64_t n = c->M;
  double* w = c->w[c->cur];
  const int nchunks = (int)((n + kSumChunk - 1) / kSumChunk);
  hipLaunchKernelGGL(chunk_tile_sum_kernel<true>, dim3(nchunks), dim3(512), 0, c->stream, w, n, d_chunks_out, c->cdf_approx,
                     (const double*)c->logw, d_gmax, 1, c->scal);
  c->dshard_tiles_ready = true;
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// divide by the global sum, approximate scan, classification, integer scan; leaves this rank's
// irregular records (global index, weight, LOCAL integer prefix) in c->irr
int launch_shard_scan(mclhip_ctx* c, double total, double approx_offset, int64_t first_index, const double* d_offset) {
  const int64_t n = c->M;
  double* wn = c->w[c->cur];
  const unsigned g = (unsigned)((n + 255) / 256);
  if (d_offset)   // device-driven path: the global sum is in scal[0], the offset behind d_offset
    hipLaunchKernelGGL(divide_kernel, dim3(g), dim3(256), 0, c->stream, wn, n, (const double*)c->scal);
  else
    hipLaunchKernelGGL(divide_by_kernel, dim3(g), dim3(256), 0, c->stream, wn, n, total);
  const int ntile = (int)((n + kScanTile - 1) / kScanTile);
  LoadF64 lf{wn};
  StoreF64 sf{c->cdf_approx};
  hipLaunchKernelGGL((scan_reduce_kernel<double, LoadF64>), dim3(ntile), dim3(kScanBlock), 0, c->stream, lf, n,
                     (double*)c->blk_a);
  hipLaunchKernelGGL((scan_blocks_kernel<double>), dim3(1), dim3(kScanBlock), 0, c->stream, (double*)c->blk_a, ntile);
  hipLaunchKernelGGL((scan_apply_kernel<double, LoadF64, StoreF64>), dim3(ntile), dim3(kScanBlock), 0, c->stream, lf,
                     sf, n, (const double*)c->blk_a);
  ClassifyLoader cl{c->cdf_approx, wn, first_index, approx_offset, d_offset};
  StoreKF sk{(unsigned long long*)c->kscan, c->fscan, (IrrRecord*)c->irr, wn, first_index};
  hipLaunchKernelGGL((scan_reduce_kernel<KF, ClassifyLoader>), dim3(ntile), dim3(kScanBlock), 0, c->stream, cl, n,
                     (KF*)c->blk_b);
  hipLaunchKernelGGL((scan_blocks_kernel<KF>), dim3(1), dim3(kScanBlock), 0, c->stream, (KF*)c->blk_b, ntile);
  hipLaunchKernelGGL((scan_apply_kernel<KF, ClassifyLoader, StoreKF>), dim3(ntile), dim3(kScanBlock), 0, c->stream, cl,
                     sk, n, (const KF*)c->blk_b);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// replay the GLOBAL record list (already on the device in d_records, global integer prefixes)
// and turn this rank's integer prefixes into exact running sums
int launch_shard_finalize(mclhip_ctx* c, const void* d_records, int64_t nrec, void* d_bases, uint64_t P_off,
                          uint32_t F_off, const unsigned long long* d_meta) {
  const int64_t n = c->M;
  const unsigned g = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(replay_kernel, dim3(1), dim3(64), 0, c->stream, (const IrrRecord*)d_records,
                     (const unsigned int*)nullptr, n, nrec, (RunBase*)d_bases, d_meta);
  hipLaunchKernelGGL(finalize_cdf_kernel, dim3(g), dim3(256), 0, c->stream, (const unsigned long long*)c->kscan,
                     c->fscan, (const RunBase*)d_bases, n, (unsigned long long)P_off, (unsigned int)F_off,
                     c->cdf_exact, d_meta);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// ---------------------------------------------------------------------------
// Device-driven variant of the cross-rank steps: what the host computes from gathered values in the
// functions above is computed by small kernels here, so that one filter step needs a single
// device-to-host read (the boundary counts that size the row exchange).
// ---------------------------------------------------------------------------
// scal[0] = sum of every rank's chunk sums, added left to right in rank order (== np.sum of the
// whole weight vector); scal[5] = (sum of the chunks of the ranks before this one) / scal[0]
__global__ void shard_totals_kernel(const double* __restrict__ all_chunks, ShardLayout lay, double* __restrict__ scal) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double t = 0.0, below = 0.0;
  for (int r = 0; r < lay.world; ++r) {
    if (r == lay.rank) below = t;
    const double* src = all_chunks + (size_t)r * lay.cmax;
    for (int j = 0; j < lay.nchunks[r]; ++j) t = t + src[j];
  }
  scal[0] = t;
  scal[5] = below / t;
}

// header (integer total, record count) + the first `cap` irregular records of this rank
__global__ void export_records_kernel(const unsigned long long* __restrict__ ks, const unsigned int* __restrict__ fs,
                                      const IrrRecord* __restrict__ irr, int64_t n, int cap,
                                      unsigned long long* __restrict__ out) {
  const unsigned int nrec = fs[n - 1];
  if (threadIdx.x == 0) { out[0] = ks[n - 1]; out[1] = nrec; }
  IrrRecord* dst = reinterpret_cast<IrrRecord*>(out + 2);
  const unsigned int m = nrec < (unsigned int)cap ? nrec : (unsigned int)cap;
  for (unsigned int j = threadIdx.x; j < m; j += blockDim.x) dst[j] = irr[j];
}

// every rank's exported block -> one global record list with global integer prefixes;
// meta[0] = records in the list, meta[1] / meta[2] = this rank's integer / record offset,
// meta[3] = 1 when some rank had more records than fit (the caller then takes the host path)
__device__ inline void merge_records_body(const unsigned long long* __restrict__ all, int world, int rank, int cap,
                                          IrrRecord* __restrict__ g_records, unsigned long long* __restrict__ meta) {
  __shared__ unsigned long long p_off[65];
  __shared__ unsigned long long f_off[65];
  const size_t stride = 2 + (size_t)cap * (sizeof(IrrRecord) / sizeof(unsigned long long));
  if (threadIdx.x == 0) {
    unsigned long long p = 0ull, f = 0ull, over = 0ull;
    for (int s = 0; s < world; ++s) {
      p_off[s] = p; f_off[s] = f;
      const unsigned long long* blk = all + (size_t)s * stride;
      p += blk[0];                  // wraps modulo 2^64 like the device prefix
      if (blk[1] > (unsigned long long)cap) over = 1ull;
      f += blk[1] > (unsigned long long)cap ? (unsigned long long)cap : blk[1];
    }
    meta[0] = f; meta[1] = p_off[rank]; meta[2] = f_off[rank]; meta[3] = over;
  }
  __syncthreads();
  for (int s = 0; s < world; ++s) {
    const unsigned long long* blk = all + (size_t)s * stride;
    const unsigned long long nrec = blk[1] > (unsigned long long)cap ? (unsigned long long)cap : blk[1];
    const IrrRecord* src = reinterpret_cast<const IrrRecord*>(blk + 2);
    for (unsigned long long j = threadIdx.x; j < nrec; j += blockDim.x) {
      IrrRecord rec = src[j];
      rec.P = rec.P + p_off[s];
      g_records[f_off[s] + j] = rec;
    }
  }
}

__global__ void merge_records_kernel(const unsigned long long* __restrict__ all, int world, int rank, int cap,
                                     IrrRecord* __restrict__ g_records, unsigned long long* __restrict__ meta) {
  merge_records_body(all, world, rank, cap, g_records, meta);
}

// ---- which rows travel: only those that some position draws --------------------------------------------------
// Position m draws the row i(m) = number of running sums below U_m, U_m = r + m / M (mc_localization.py:254-257).
// Seen from a row with running sum c: the positions that draw it end where U_m first exceeds c.  first_above(c) is
// that position, found with the very comparison the walk uses, so that source and destination agree to the bit.
__device__ inline double position_u(int64_t m, int64_t m_total, double r) { return r + ((double)m / (double)m_total); }

__device__ inline int64_t first_above(double c, int64_t m_total, double r) {
  // first m in [0, m_total] with U_m > c (m_total if none; also for NaN, which nothing exceeds)
  int64_t lo = 0, hi = m_total;
  const double est = (c - r) * (double)m_total;            // U is monotone in m: a bracket around the estimate, checked
  if (est > -4.0 && est < (double)m_total + 4.0) {
    int64_t a = (int64_t)est - 2, b = (int64_t)est + 3;
    a = a < 0 ? 0 : (a > m_total ? m_total : a);
    b = b < 0 ? 0 : (b > m_total ? m_total : b);
    const bool lo_ok = (a == 0) || !(position_u(a - 1, m_total, r) > c);
    const bool hi_ok = (b == m_total) || (position_u(b, m_total, r) > c);
    if (lo_ok && hi_ok) { lo = a; hi = b; }
  }
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (position_u(mid, m_total, r) > c) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// G[j] = first position beyond those that can draw local row j or an earlier one; the globally last row keeps every
// position to the end (indices past the end are clamped to it)
__global__ void draw_ends_kernel(const double* __restrict__ cdf, int64_t n, int64_t m_total, double r, int last_rank,
                                 int64_t* __restrict__ G) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  G[j] = (last_rank && j == n - 1) ? m_total : first_above(cdf[j], m_total, r);
}

// local row j is drawn iff some position lies in [G[j-1], G[j]); whether the first local row is drawn depends on the
// previous rank's last running sum, which is not known here: it always counts as drawn (one row too many at worst)
struct DrawnFlagLoader {
  const int64_t* G;
  __device__ unsigned int operator()(int64_t j) const { return (j == 0 || G[j] > G[j - 1]) ? 1u : 0u; }
};
struct StoreU32 {
  unsigned int* p;
  __device__ void operator()(int64_t i, unsigned int incl, unsigned int) const { p[i] = incl; }
};

// slices[4d .. 4d+3] = first row, last row (local; first > last: none), number of drawn rows among them, and where
// they start in the send buffer, for destination rank d; counts[d] = that number, counts[world] = 1 when nothing
// was planned (the record merge overflowed, or the plan would not fit n + world rows).  Destination d owns the positions [first[d], first[d+1]); it needs the rows j with
// G[j] > first[d] (some position of d at or after the row's first drawer... its range ends beyond d's start) and
// G[j-1] <= last position of d.
__global__ void plan_slices_kernel(const int64_t* __restrict__ G, const unsigned int* __restrict__ S, int64_t n,
                                   ShardLayout lay, const unsigned long long* __restrict__ meta,
                                   int64_t* __restrict__ slices, int64_t* __restrict__ counts) {
  __shared__ int64_t s_cnt[64];
  // a wavefront per destination: the searches probe 64 places at a time (four rounds of dependent reads for a million
  // rows instead of twenty)
  const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  auto first_row_above = [&](int64_t x) -> int64_t {        // first j with G[j] > x (n if none); wave-uniform
    int64_t lo = 0, hi = n;                                  // the answer lies in [lo, hi]
    while (hi - lo > 64) {
      const int64_t step = (hi - lo + 63) / 64;
      const int64_t p = lo + (int64_t)lane * step;
      const bool in = p < hi;
      const unsigned long long m = __ballot(in && G[p] > x);
      const unsigned long long valid = __ballot(in);
      if (m == 0ull) {
        lo = lo + (int64_t)(63 - __clzll(valid)) * step + 1;  // beyond the last place probed
      } else {
        const int f = __ffsll((long long)m) - 1;
        hi = lo + (int64_t)f * step;
        if (f > 0) lo = lo + (int64_t)(f - 1) * step + 1;
      }
    }
    const int64_t p = lo + lane;
    const unsigned long long m = __ballot(p < hi && G[p] > x);
    return m == 0ull ? hi : lo + (__ffsll((long long)m) - 1);
  };
  for (int d = threadIdx.x >> 6; d < lay.world; d += nw) {
    const int64_t jl = first_row_above(lay.first[d]);
    int64_t jh = first_row_above(lay.first[d + 1] - 1);      // G[j-1] <= last position  <=>  j <= that row
    if (jh > n - 1) jh = n - 1;
    int64_t cnt = 0;
    if (lay.first[d + 1] > lay.first[d] && jl <= jh) cnt = (int64_t)S[jh] - (jl > 0 ? (int64_t)S[jl - 1] : 0);
    if (lane == 0) {
      slices[4 * d] = jl; slices[4 * d + 1] = jh; slices[4 * d + 2] = cnt;
      s_cnt[d] = cnt;
      counts[d] = cnt;
    }
  }
  __syncthreads();
  const int d = threadIdx.x;
  if (d == 0) {
    // With exact running sums the draw ends ascend and the slices hold every local row once and the rows at the rank
    // boundaries twice: at most n + world rows, which is what the callers' send buffers hold.  After a record merge
    // that ran out of room the running sums are not the exact ones and the draw ends need not ascend: then, or
    // whenever the plan would exceed that bound, nothing is planned (no row is packed) and the overflow word is set,
    // so that the caller finishes the step another way.
    int64_t total = 0;
    for (int k = 0; k < lay.world; ++k) total += s_cnt[k];
    const bool refuse = (meta && meta[3]) || total > n + (int64_t)lay.world;
    int64_t off = 0;
    for (int k = 0; k < lay.world; ++k) {
      if (refuse) { slices[4 * k + 2] = 0; counts[k] = 0; s_cnt[k] = 0; }
      slices[4 * k + 3] = off;
      off += s_cnt[k];
    }
    counts[lay.world] = refuse ? 1 : 0;
  }
}

// rows of six doubles (x, y, theta, w, running sum, global index) of the drawn local rows, destination after
// destination; a row that positions of two ranks draw goes to both
template <typename T>
__global__ __launch_bounds__(256) void pack_drawn_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ th,
                                                         const double* __restrict__ w, const double* __restrict__ cdf,
                                                         const int64_t* __restrict__ G, const unsigned int* __restrict__ S,
                                                         int64_t n, int64_t first_index, int world,
                                                         const int64_t* __restrict__ slices, double* __restrict__ dst,
                                                         int64_t* __restrict__ sel) {
  __shared__ int64_t s_sl[4 * 64];
  for (int k = threadIdx.x; k < 4 * world; k += blockDim.x) s_sl[k] = slices[k];
  __syncthreads();
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if (!(j == 0 || G[j] > G[j - 1])) return;
  const int64_t sj = (int64_t)S[j];
  for (int d = 0; d < world; ++d) {                          // a handful of ranks; the slices ascend with d
    const int64_t jl = s_sl[4 * d], jh = s_sl[4 * d + 1];
    if (s_sl[4 * d + 2] == 0 || j < jl || j > jh) continue;
    const int64_t pos = s_sl[4 * d + 3] + (sj - (jl > 0 ? (int64_t)S[jl - 1] : 0)) - 1;
    double* row = dst + pos * 6;
    row[0] = (double)x[j];
    row[1] = (double)y[j];
    row[2] = (double)th[j];
    row[3] = w[j];
    row[4] = cdf[j];
    row[5] = __longlong_as_double((long long)(first_index + j));
    if (sel) sel[pos] = j;                                   // which local row stands at this place of the send buffer
  }
}

// resample this rank's positions from the received rows (all sources, ascending global index): the ancestor is the
// first row whose running sum is not below U_m, the last row if there is none (the clamp of the single-GPU walk)
template <typename T>
__global__ __launch_bounds__(256) void unpack_drawn_kernel(const double* __restrict__ rows, int64_t L, int64_t first_index,
                                                           int64_t m_total, int64_t n, double r, T* __restrict__ x2,
                                                           T* __restrict__ y2, T* __restrict__ th2, double* __restrict__ w2,
                                                           int64_t* __restrict__ idx, int64_t* __restrict__ pos_out) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const double U = position_u(first_index + j, m_total, r);
  int64_t lo = 0, hi = L;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (U > rows[mid * 6 + 4]) lo = mid + 1; else hi = mid;
  }
  if (lo > L - 1) lo = L - 1;
  const double* s = rows + lo * 6;
  if (pos_out) pos_out[j] = lo;                              // which received row this position took
  idx[j] = (int64_t)__double_as_longlong(s[5]);
  x2[j] = (T)s[0];
  y2[j] = (T)s[1];
  th2[j] = (T)s[2];
  w2[j] = s[3];
}

// ---- the same exchange without a host read: every pair of ranks exchanges a block of `cap` rows, whatever it holds.
// The rows a rank draws from itself -- all of them when the weights are flat -- never enter the exchange: they are
// packed into a buffer of the context's own.  What a block holds is known from the gathered counts, on the device;
// whether every block was large enough is known there too, and the host looks at that flag only after the step has
// been enqueued to its end (nothing is resampled when it is set: the step is then finished with exact counts).
// 0 fine, 1 a rank's record merge overflowed (it planned nothing), 2 some pair's rows exceed `cap`: from the whole
// gathered matrix (all_counts[s][d]: rows rank s has for rank d; all_counts[s][world]: rank s planned nothing), so the
// same on every rank.  By a whole block of 256 threads; s_flag: 256 ints of LDS.
__device__ inline int exchange_flag(const int64_t* __restrict__ all_counts, int world, int64_t cap, int* s_flag) {
  int f1 = 0, f2 = 0;
  for (int k = threadIdx.x; k < world * (world + 1); k += blockDim.x) {
    const int s_ = k / (world + 1), d = k - s_ * (world + 1);
    const int64_t v = all_counts[k];
    if (d == world) { if (v != 0) f1 = 1; }
    else if (s_ != d && v > cap) f2 = 1;
  }
  (void)s_flag;
  const int any1 = __syncthreads_or(f1), any2 = __syncthreads_or(f2);
  return any1 ? 1 : (any2 ? 2 : 0);
}

template <typename T>
__global__ __launch_bounds__(256) void pack_fixed_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ th,
                                                         const double* __restrict__ w, const double* __restrict__ cdf,
                                                         const int64_t* __restrict__ G, const unsigned int* __restrict__ S,
                                                         int64_t n, int64_t first_index, int world, int rank, int64_t cap,
                                                         const int64_t* __restrict__ slices, double* __restrict__ dst,
                                                         double* __restrict__ self_dst,
                                                         const int64_t* __restrict__ all_counts, int64_t* __restrict__ flag_out,
                                                         int64_t* __restrict__ host_late) {
  __shared__ int64_t s_sl[4 * 64];
  __shared__ int s_flag[256];
  if (blockIdx.x == 0) {   // the exchange's flag, as early as it can be known: the host waits for this, not for the step's end
    const int f = exchange_flag(all_counts, world, cap, s_flag);
    if (threadIdx.x == 0) flag_out[0] = f;
    if (host_late) {       // ... and for this: counts and flag stored straight into the host's page-locked memory
      for (int k = threadIdx.x; k < world * (world + 1); k += blockDim.x) host_late[k] = all_counts[k];
      if (threadIdx.x == 0) host_late[world * (world + 1)] = f;
    }
  }
  for (int k = threadIdx.x; k < 4 * world; k += blockDim.x) s_sl[k] = slices[k];
  __syncthreads();
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  if (!(j == 0 || G[j] > G[j - 1])) return;
  const int64_t sj = (int64_t)S[j];
  for (int d = 0; d < world; ++d) {
    const int64_t jl = s_sl[4 * d], jh = s_sl[4 * d + 1];
    if (s_sl[4 * d + 2] == 0 || j < jl || j > jh) continue;
    const int64_t local = (sj - (jl > 0 ? (int64_t)S[jl - 1] : 0)) - 1;     // place among the rows for rank d
    double* row;
    if (d == rank) row = self_dst + local * 6;
    else if (local < cap) row = dst + ((int64_t)d * cap + local) * 6;
    else continue;                                                          // does not fit its block: the flag says so
    row[0] = (double)x[j];
    row[1] = (double)y[j];
    row[2] = (double)th[j];
    row[3] = w[j];
    row[4] = cdf[j];
    row[5] = __longlong_as_double((long long)(first_index + j));
  }
}

// unpack_drawn_kernel over rows that stand in one block per source (the rank's own in its own buffer)
// Every block reads the gathered counts for itself (all_counts[s][d]: rows rank s has for rank d; all_counts[s][world]:
// rank s planned nothing because a record merge overflowed): how many rows come from every source, and the flag --
// 0 fine, 1 a record merge overflowed, 2 some pair's rows exceed `cap` -- which is the same on every rank (it is formed
// from the whole matrix) and goes to flag_out.
template <typename T>
__global__ __launch_bounds__(256) void unpack_fixed_kernel(const double* __restrict__ recv, const double* __restrict__ self_rows,
                                                           const int64_t* __restrict__ all_counts, int world, int rank, int64_t cap,
                                                           int64_t* __restrict__ flag_out,
                                                           int64_t first_index, int64_t m_total, int64_t n, double r,
                                                           T* __restrict__ x2, T* __restrict__ y2, T* __restrict__ th2,
                                                           double* __restrict__ w2, int64_t* __restrict__ idx) {
  __shared__ int64_t s_off[66];
  __shared__ int s_flag[256];
  const int flag = exchange_flag(all_counts, world, cap, s_flag);
  if (threadIdx.x == 0) {
    int64_t off = 0;
    for (int s_ = 0; s_ < world; ++s_) {
      s_off[s_] = off;
      off += all_counts[(size_t)s_ * (world + 1) + rank];
    }
    s_off[world] = off;
    if (blockIdx.x == 0) flag_out[0] = flag;
  }
  __syncthreads();
  if (flag != 0) return;
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const double U = position_u(first_index + j, m_total, r);
  auto block_of = [&](int s) { return s == rank ? self_rows : recv + (int64_t)s * cap * 6; };
  // the source whose rows hold the first running sum that is not below U: sources ascend, so it is the first one whose
  // last row qualifies; if there is none, the last row of all (the clamp of the single-GPU walk)
  int src = -1, last_src = -1;
  for (int s = 0; s < world; ++s) {
    const int64_t len = s_off[s + 1] - s_off[s];
    if (len <= 0) continue;
    last_src = s;
    if (!(U > block_of(s)[(len - 1) * 6 + 4])) { src = s; break; }
  }
  if (last_src < 0) return;                                  // (no rows at all: cannot happen, every position has an ancestor)
  const double* rows;
  int64_t lo = 0;
  if (src < 0) {
    rows = block_of(last_src);
    lo = s_off[last_src + 1] - s_off[last_src] - 1;
  } else {
    rows = block_of(src);
    int64_t hi = s_off[src + 1] - s_off[src];
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (U > rows[mid * 6 + 4]) lo = mid + 1; else hi = mid;
    }
  }
  const double* sr = rows + lo * 6;
  idx[j] = (int64_t)__double_as_longlong(sr[5]);
  x2[j] = (T)sr[0];
  y2[j] = (T)sr[1];
  th2[j] = (T)sr[2];
  w2[j] = sr[3];
}

int launch_pack_fixed(mclhip_ctx* c, int world, int rank, int64_t cap, double* d_dst, double* d_self,
                      const int64_t* d_all_counts, int64_t* d_flag_out, int64_t* h_late) {
  const int64_t n = c->M;
  const int a = c->cur;
  const unsigned g = (unsigned)((n + 255) / 256);
  const int64_t* G = c->draw_ends;
  const int64_t* slices = c->small_i + 8;
  if (c->dtype == MCLHIP_F32)
    hipLaunchKernelGGL(pack_fixed_kernel<float>, dim3(g), dim3(256), 0, c->stream, (const float*)c->px[a], (const float*)c->py[a],
                       (const float*)c->pth[a], (const double*)c->w[a], (const double*)c->cdf_exact, G, (const unsigned int*)c->fscan,
                       n, c->first_index, world, rank, cap, slices, d_dst, d_self, d_all_counts, d_flag_out, h_late);
  else
    hipLaunchKernelGGL(pack_fixed_kernel<double>, dim3(g), dim3(256), 0, c->stream, (const double*)c->px[a], (const double*)c->py[a],
                       (const double*)c->pth[a], (const double*)c->w[a], (const double*)c->cdf_exact, G, (const unsigned int*)c->fscan,
                       n, c->first_index, world, rank, cap, slices, d_dst, d_self, d_all_counts, d_flag_out, h_late);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// (the arrays are swapped here, on the host, whether or not the kernel resamples: see mclhip_dshard_unpack_fixed)
int launch_unpack_fixed(mclhip_ctx* c, const double* d_recv, const double* d_self, const int64_t* d_all_counts, int world,
                        int rank, int64_t cap, double r, int64_t* d_flag_out) {
  const int64_t n = c->M;
  const int b = 1 - c->cur;
  const unsigned g = (unsigned)((n + 255) / 256);
  if (c->dtype == MCLHIP_F32)
    hipLaunchKernelGGL(unpack_fixed_kernel<float>, dim3(g), dim3(256), 0, c->stream, d_recv, d_self, d_all_counts, world,
                       rank, cap, d_flag_out, c->first_index, c->m_total, n, r, (float*)c->px[b], (float*)c->py[b], (float*)c->pth[b],
                       c->w[b], c->idx);
  else
    hipLaunchKernelGGL(unpack_fixed_kernel<double>, dim3(g), dim3(256), 0, c->stream, d_recv, d_self, d_all_counts, world,
                       rank, cap, d_flag_out, c->first_index, c->m_total, n, r, (double*)c->px[b], (double*)c->py[b], (double*)c->pth[b],
                       c->w[b], c->idx);
  MCL_HIP(c, hipGetLastError());
  c->cur = b;
  return MCLHIP_OK;
}

int launch_dshard_scan(mclhip_ctx* c, const double* d_all_chunks, const ShardLayout& lay, void* d_rec_out, int cap) {
  int all_chunks = 0;
  for (int r = 0; r < lay.world; ++r) all_chunks += lay.nchunks[r];
  const bool tiles = c->dshard_tiles_ready;   // (the tile sums of launch_dshard_weights; spent below either way)
  c->dshard_tiles_ready = false;
  if (all_chunks <= kMaxFusedChunks && tiles) {
    // two launches (see normalise_resample_typed): total and offset by every block for itself, division, approximate
    // scan and classification; then integer prefixes and records, exported by the block that finishes last
    const int64_t n = c->M;
    double* w = c->w[c->cur];
    const int ntile = (int)((n + kScanTile - 1) / kScanTile);
    unsigned int* done_counter = reinterpret_cast<unsigned int*>(c->scal + 6);
    hipLaunchKernelGGL(normalise_classify_shard_kernel, dim3(ntile), dim3(kScanBlock), 0, c->stream, w, n, d_all_chunks, lay,
                       (const double*)c->cdf_approx, (KF*)c->blk_b, c->scal, done_counter);
    hipLaunchKernelGGL(records_kernel<true>, dim3(ntile), dim3(kScanBlock), 0, c->stream, w, n, (const double*)c->cdf_approx,
                       (const KF*)c->blk_b, (const double*)c->scal, (unsigned long long*)c->kscan, c->fscan,
                       (IrrRecord*)c->irr, (RunBase*)c->bases, done_counter, c->first_index, cap,
                       (unsigned long long*)d_rec_out);
    MCL_HIP(c, hipGetLastError());
    return MCLHIP_OK;
  }
  hipLaunchKernelGGL(shard_totals_kernel, dim3(1), dim3(64), 0, c->stream, d_all_chunks, lay, c->scal);
  MCL_TRY_W(launch_shard_scan(c, 0.0, 0.0, c->first_index, c->scal + 5));
  hipLaunchKernelGGL(export_records_kernel, dim3(1), dim3(256), 0, c->stream, (const unsigned long long*)c->kscan,
                     c->fscan, (const IrrRecord*)c->irr, c->M, cap, (unsigned long long*)d_rec_out);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}



// ---- FastSLAM over several GPUs, device-driven: the weight vector is normalised by the Python-order sum of
// fast_slam1.py:411-418 (the last running sum of the raw weights, floored at 1e-8), not by np.sum.  Two rounds of the
// record exchange: raw weights (lasts == nullptr: nothing is divided, scal[0] = 1), then the weights over that sum.
__global__ void fs_shard_totals_kernel(const double* __restrict__ all_chunks, const double* __restrict__ lasts, ShardLayout lay,
                                       double* __restrict__ scal) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double below = 0.0;                                        // approximate: only places the local scan's error bound
  for (int r = 0; r < lay.rank; ++r) {
    const double* src = all_chunks + (size_t)r * lay.cmax;
    for (int j = 0; j < lay.nchunks[r]; ++j) below = below + src[j];
  }
  const double total = lasts ? fmax(1e-8, lasts[2 * (lay.world - 1)]) : 1.0;   // max(1e-8, sum_w) (:415); lasts: (sum, flag) per rank
  scal[0] = total;
  scal[5] = below / total;
}

int launch_fs_dshard_scan(mclhip_ctx* c, const double* d_all_chunks, const double* d_lasts, const ShardLayout& lay,
                          void* d_rec_out, int cap) {
  hipLaunchKernelGGL(fs_shard_totals_kernel, dim3(1), dim3(64), 0, c->stream, d_all_chunks, d_lasts, lay, c->scal);
  MCL_TRY_W(launch_shard_scan(c, 0.0, 0.0, c->first_index, c->scal + 5));
  hipLaunchKernelGGL(export_records_kernel, dim3(1), dim3(256), 0, c->stream, (const unsigned long long*)c->kscan,
                     c->fscan, (const IrrRecord*)c->irr, c->M, cap, (unsigned long long*)d_rec_out);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// out[0] = this rank's last running sum, out[1] = 1.0 if some rank had more records than the exchange holds
__global__ void fs_last_sum_kernel(const double* __restrict__ cdf, int64_t n, const unsigned long long* __restrict__ meta,
                                   double* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = cdf[n - 1]; out[1] = meta[3] ? 1.0 : 0.0; }
}

// first round: the global record list of the raw weights, this rank's exact running sums, its last one -> d_last_out
int launch_fs_dshard_finalize_raw(mclhip_ctx* c, const void* d_all_rec, const ShardLayout& lay, int cap, double* d_last_out) {
  unsigned long long* meta = reinterpret_cast<unsigned long long*>(c->small_i);
  hipLaunchKernelGGL(merge_records_kernel, dim3(1), dim3(256), 0, c->stream, (const unsigned long long*)d_all_rec,
                     lay.world, lay.rank, cap, (IrrRecord*)c->g_records, meta);
  MCL_TRY_W(launch_shard_finalize(c, c->g_records, 0, c->g_bases, 0, 0, meta));
  hipLaunchKernelGGL(fs_last_sum_kernel, dim3(1), dim3(64), 0, c->stream, (const double*)c->cdf_exact, c->M, meta, d_last_out);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// merge_records_kernel and replay_kernel in one launch: the block merges, its first wavefront replays
__global__ __launch_bounds__(256) void merge_replay_kernel(const unsigned long long* __restrict__ all, int world, int rank, int cap,
                                                           IrrRecord* __restrict__ g_records, unsigned long long* __restrict__ meta,
                                                           RunBase* __restrict__ bases) {
  merge_records_body(all, world, rank, cap, g_records, meta);
  __syncthreads();                       // (one block: its stores are visible to its own threads behind the barrier)
  if (threadIdx.x < 64) replay_body(g_records, (unsigned int)meta[0], bases, threadIdx.x);
}

// finalize_cdf_kernel, draw_ends_kernel and the first pass of the scan of the drawn flags in one launch: a thread per
// element (coalesced), a block per 256; exact running sums -> cdf, draw ends -> G (an array of its own: a block's first
// flag needs the draw end of the element before it, which this block forms once more from that element's integer
// prefix), the number of drawn rows of the block -> blk_count
constexpr int kDrawBlock = 256;
__global__ __launch_bounds__(kDrawBlock) void finalize_draw_count_kernel(const unsigned long long* __restrict__ ks,
                                                                         const unsigned int* __restrict__ fs,
                                                                         const RunBase* __restrict__ bases, int64_t n,
                                                                         double* __restrict__ cdf,
                                                                         const unsigned long long* __restrict__ off_ptr,
                                                                         int64_t m_total, double r, int last_rank,
                                                                         int64_t* __restrict__ G, unsigned int* __restrict__ blk_count) {
  __shared__ int64_t s_g[kDrawBlock];
  __shared__ unsigned int s_cnt[kDrawBlock / 64];
  const unsigned long long P_off = off_ptr[1];
  const unsigned int F_off = (unsigned int)off_ptr[2];
  auto end_of = [&](int64_t i, bool store) -> int64_t {
    const RunBase b = bases[F_off + fs[i] - 1];
    const double ci = b.c + (double)(P_off + ks[i] - b.P) * ldexp(1.0, b.e - 52);  // exact (see finalize_cdf_kernel)
    if (store) cdf[i] = ci;
    return (last_rank && i == n - 1) ? m_total : first_above(ci, m_total, r);
  };
  const int64_t i = (int64_t)blockIdx.x * kDrawBlock + threadIdx.x;
  int64_t g = 0;
  if (i < n) { g = end_of(i, true); G[i] = g; }
  s_g[threadIdx.x] = g;
  __syncthreads();
  unsigned int flag = 0u;
  if (i < n) {
    int64_t before = 0;
    if (threadIdx.x > 0) before = s_g[threadIdx.x - 1];
    else if (i > 0) before = end_of(i - 1, false);
    flag = (i == 0 || g > before) ? 1u : 0u;                 // DrawnFlagLoader
  }
  const unsigned int wave_cnt = (unsigned int)__popcll(__ballot(flag != 0u));
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = wave_cnt;
  __syncthreads();
  if (threadIdx.x == 0) blk_count[blockIdx.x] = (s_cnt[0] + s_cnt[1]) + (s_cnt[2] + s_cnt[3]);
}

// second pass: S[i] = number of drawn rows among local rows 0..i; every block adds the earlier blocks' counts for itself
__global__ __launch_bounds__(kDrawBlock) void drawn_scan_kernel(const int64_t* __restrict__ G, int64_t n,
                                                                const unsigned int* __restrict__ blk_count,
                                                                unsigned int* __restrict__ S) {
  __shared__ unsigned int s_red[kDrawBlock / 64];
  __shared__ unsigned int s_wave[kDrawBlock / 64];
  unsigned int part = 0u;
  for (int j = threadIdx.x; j < (int)blockIdx.x; j += kDrawBlock) part += blk_count[j];
  for (int d = 32; d >= 1; d >>= 1) part += (unsigned int)__shfl_xor((int)part, d, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = part;
  const int64_t i = (int64_t)blockIdx.x * kDrawBlock + threadIdx.x;
  DrawnFlagLoader fl{G};
  const unsigned int f = (i < n) ? fl(i) : 0u;
  const unsigned long long m = __ballot(f != 0u);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) s_wave[wv] = (unsigned int)__popcll(m);
  __syncthreads();
  unsigned int off = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  for (int k = 0; k < wv; ++k) off += s_wave[k];
  const unsigned int incl = off + (unsigned int)__popcll(m & ((lane == 63) ? ~0ull : ((1ull << (lane + 1)) - 1ull)));
  if (i < n) S[i] = incl;
}

int launch_dshard_finalize(mclhip_ctx* c, const void* d_all_rec, const ShardLayout& lay, int cap, double r,
                           int64_t* d_counts) {
  unsigned long long* meta = reinterpret_cast<unsigned long long*>(c->small_i);   // [4] of the 1024-entry staging
  const int64_t n = c->M;
  const unsigned g = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(merge_replay_kernel, dim3(1), dim3(256), 0, c->stream, (const unsigned long long*)d_all_rec,
                     lay.world, lay.rank, cap, (IrrRecord*)c->g_records, meta, (RunBase*)c->g_bases);
  // (the draw ends in an array of their own -- the tile sums' of launch_dshard_weights, spent by now --, the scan of the
  // drawn flags over the record counts, which the running sums have consumed)
  int64_t* G = reinterpret_cast<int64_t*>(c->cdf_approx);
  unsigned int* blk_count = reinterpret_cast<unsigned int*>(c->bases);   // (one per 256 rows; the single-GPU replay's bases are not in use on this path)
  hipLaunchKernelGGL(finalize_draw_count_kernel, dim3(g), dim3(kDrawBlock), 0, c->stream, (const unsigned long long*)c->kscan,
                     (const unsigned int*)c->fscan, (const RunBase*)c->g_bases, n, c->cdf_exact, (const unsigned long long*)meta,
                     c->m_total, r, lay.rank == lay.world - 1 ? 1 : 0, G, blk_count);
  hipLaunchKernelGGL(drawn_scan_kernel, dim3(g), dim3(kDrawBlock), 0, c->stream, (const int64_t*)G, n,
                     (const unsigned int*)blk_count, c->fscan);
  c->draw_ends = G;
  int64_t* slices = c->small_i + 8;
  hipLaunchKernelGGL(plan_slices_kernel, dim3(1), dim3(64 * (lay.world < 16 ? lay.world : 16)), 0, c->stream, (const int64_t*)G, (const unsigned int*)c->fscan, n, lay,
                     meta, slices, d_counts);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// which local rows are drawn at all, and by the positions of which rank: only those travel.  Needs the exact running
// sums in c->cdf_exact; leaves the draw ends in c->kscan, the scan of the drawn flags in c->fscan, the slices behind
// c->small_i + 8 and the row counts per destination (+ the overflow word) in d_counts.
int launch_plan_drawn(mclhip_ctx* c, const ShardLayout& lay, double r, int64_t* d_counts, const unsigned long long* meta,
                      bool ends_done) {
  const int64_t n = c->M;
  const unsigned g = (unsigned)((n + 255) / 256);
  const int ntile = (int)((n + kScanTile - 1) / kScanTile);
  int64_t* G = reinterpret_cast<int64_t*>(c->kscan);           // the integer prefixes are spent once the running sums exist
  c->draw_ends = G;
  unsigned int* S = c->fscan;                                  // likewise the record counts
  if (!ends_done)
    hipLaunchKernelGGL(draw_ends_kernel, dim3(g), dim3(256), 0, c->stream, (const double*)c->cdf_exact, n, c->m_total, r,
                       lay.rank == lay.world - 1 ? 1 : 0, G);
  DrawnFlagLoader fl{G};
  StoreU32 su{S};
  hipLaunchKernelGGL((scan_reduce_kernel<unsigned int, DrawnFlagLoader>), dim3(ntile), dim3(kScanBlock), 0, c->stream, fl, n,
                     (unsigned int*)c->blk_a);
  hipLaunchKernelGGL((scan_blocks_kernel<unsigned int>), dim3(1), dim3(kScanBlock), 0, c->stream, (unsigned int*)c->blk_a, ntile);
  hipLaunchKernelGGL((scan_apply_kernel<unsigned int, DrawnFlagLoader, StoreU32>), dim3(ntile), dim3(kScanBlock), 0, c->stream, fl,
                     su, n, (const unsigned int*)c->blk_a);
  int64_t* slices = c->small_i + 8;                            // [4 * world] behind the merge's four words
  hipLaunchKernelGGL(plan_slices_kernel, dim3(1), dim3(64 * (lay.world < 16 ? lay.world : 16)), 0, c->stream, (const int64_t*)G, (const unsigned int*)S, n, lay,
                     meta, slices, d_counts);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// the drawn rows of every destination, one after the other, into d_dst (six doubles per row)
int launch_pack_drawn(mclhip_ctx* c, int world, double* d_dst, int64_t* d_sel) {
  const int64_t n = c->M;
  const int a = c->cur;
  const unsigned g = (unsigned)((n + 255) / 256);
  const int64_t* G = c->draw_ends;
  const int64_t* slices = c->small_i + 8;
  if (c->dtype == MCLHIP_F32)
    hipLaunchKernelGGL(pack_drawn_kernel<float>, dim3(g), dim3(256), 0, c->stream, (const float*)c->px[a], (const float*)c->py[a],
                       (const float*)c->pth[a], (const double*)c->w[a], (const double*)c->cdf_exact, G, (const unsigned int*)c->fscan,
                       n, c->first_index, world, slices, d_dst, d_sel);
  else
    hipLaunchKernelGGL(pack_drawn_kernel<double>, dim3(g), dim3(256), 0, c->stream, (const double*)c->px[a], (const double*)c->py[a],
                       (const double*)c->pth[a], (const double*)c->w[a], (const double*)c->cdf_exact, G, (const unsigned int*)c->fscan,
                       n, c->first_index, world, slices, d_dst, d_sel);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// keep_pos: the received row every position took is left in c->kscan (as int64) for what travels beside the poses
int launch_unpack_drawn(mclhip_ctx* c, const double* d_rows, int64_t L, double r, bool keep_pos) {
  int64_t* pos = keep_pos ? reinterpret_cast<int64_t*>(c->kscan) : nullptr;
  const int64_t n = c->M;
  const int b = 1 - c->cur;
  const unsigned g = (unsigned)((n + 255) / 256);
  if (c->dtype == MCLHIP_F32)
    hipLaunchKernelGGL(unpack_drawn_kernel<float>, dim3(g), dim3(256), 0, c->stream, d_rows, L, c->first_index, c->m_total, n, r,
                       (float*)c->px[b], (float*)c->py[b], (float*)c->pth[b], c->w[b], c->idx, pos);
  else
    hipLaunchKernelGGL(unpack_drawn_kernel<double>, dim3(g), dim3(256), 0, c->stream, d_rows, L, c->first_index, c->m_total, n, r,
                       (double*)c->px[b], (double*)c->py[b], (double*)c->pth[b], c->w[b], c->idx, pos);
  MCL_HIP(c, hipGetLastError());
  c->cur = b;
  return MCLHIP_OK;
}

// counts[j] = number of local running sums that are < U[j]
__global__ void count_below_kernel(const double* __restrict__ cdf, int64_t n, const double* __restrict__ U, int nu,
                                   int64_t* __restrict__ counts) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nu) return;
  const double u = U[j];
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (u > cdf[mid]) lo = mid + 1; else hi = mid;
  }
  counts[j] = lo;
}

int launch_count_below(mclhip_ctx* c, const double* d_U, int nu, int64_t* d_counts) {
  hipLaunchKernelGGL(count_below_kernel, dim3((nu + 63) / 64), dim3(64), 0, c->stream, c->cdf_exact, c->M, d_U, nu,
                     d_counts);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// rows of five doubles (x, y, theta, w, running sum) for local particles [lo, hi)
template <typename T>
__global__ void pack_rows_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ th,
                                 const double* __restrict__ w, const double* __restrict__ cdf, int64_t lo, int64_t hi,
                                 double* __restrict__ dst) {
  int64_t i = lo + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hi) return;
  double* r = dst + (i - lo) * 5;
  r[0] = (double)x[i];
  r[1] = (double)y[i];
  r[2] = (double)th[i];
  r[3] = w[i];
  r[4] = cdf[i];
}

int launch_pack_rows(mclhip_ctx* c, int64_t lo, int64_t hi, double* d_dst) {
  if (hi <= lo) return MCLHIP_OK;
  const int a = c->cur;
  const unsigned g = (unsigned)((hi - lo + 255) / 256);
  if (c->dtype == MCLHIP_F32)
    hipLaunchKernelGGL(pack_rows_kernel<float>, dim3(g), dim3(256), 0, c->stream, (const float*)c->px[a],
                       (const float*)c->py[a], (const float*)c->pth[a], c->w[a], c->cdf_exact, lo, hi, d_dst);
  else
    hipLaunchKernelGGL(pack_rows_kernel<double>, dim3(g), dim3(256), 0, c->stream, (const double*)c->px[a],
                       (const double*)c->py[a], (const double*)c->pth[a], c->w[a], c->cdf_exact, lo, hi, d_dst);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// resample this rank's global positions m = first_index + j from the received rows:
// index = number of received running sums that are < U_m, clamped to the last row
template <typename T>
__global__ __launch_bounds__(256) void unpack_resample_kernel(const double* __restrict__ rows, int64_t L,
                                                              int64_t first_index, int64_t m_total, int64_t n,
                                                              double r, int64_t row0_global, T* __restrict__ x2,
                                                              T* __restrict__ y2, T* __restrict__ th2,
                                                              double* __restrict__ w2, int64_t* __restrict__ idx) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int64_t m = first_index + j;
  const double U = r + ((double)m / (double)m_total);  // mc_localization.py:254
  int64_t lo = 0, hi = L;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (U > rows[mid * 5 + 4]) lo = mid + 1; else hi = mid;
  }
  if (lo > L - 1) lo = L - 1;
  const double* s = rows + lo * 5;
  idx[j] = row0_global + lo;
  x2[j] = (T)s[0];
  y2[j] = (T)s[1];
  th2[j] = (T)s[2];
  w2[j] = s[3];
}

int launch_unpack_resample(mclhip_ctx* c, const double* d_rows, int64_t L, int64_t first_index, int64_t m_total,
                           double r, int64_t row0_global) {
  const int64_t n = c->M;
  const int b = 1 - c->cur;
  const unsigned g = (unsigned)((n + 255) / 256);
  if (c->dtype == MCLHIP_F32)
    hipLaunchKernelGGL(unpack_resample_kernel<float>, dim3(g), dim3(256), 0, c->stream, d_rows, L, first_index,
                       m_total, n, r, row0_global, (float*)c->px[b], (float*)c->py[b], (float*)c->pth[b], c->w[b],
                       c->idx);
  else
    hipLaunchKernelGGL(unpack_resample_kernel<double>, dim3(g), dim3(256), 0, c->stream, d_rows, L, first_index,
                       m_total, n, r, row0_global, (double*)c->px[b], (double*)c->py[b], (double*)c->pth[b], c->w[b],
                       c->idx);
  MCL_HIP(c, hipGetLastError());
  c->cur = b;
  return MCLHIP_OK;
}


// ===========================================================================
// Augmented MCL (mc_localization2.py): running mean of the raw weights and
// random-pose injection.
// ===========================================================================
__global__ void scale_by_kernel(const double* __restrict__ w, double* __restrict__ out, int64_t n, double m) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = w[i] / m;  // w[m] / M (mc_localization2.py:246)
}

__global__ void copy_last_kernel(const double* __restrict__ cdf, int64_t n, double* __restrict__ scal) {
  if (threadIdx.x == 0 && blockIdx.x == 0) scal[2] = cdf[n - 1];
}

// scal[2] = w_avg: `w_avg += w[m] / M` over m = 0..M-1, the same strictly sequential chain the
// order-exact scan reproduces, so the result is bit-equal to the reference's loop
int launch_mean_weight(mclhip_ctx* c) {
  const int64_t n = c->M;
  const unsigned g = (unsigned)((n + 255) / 256);
  double* tmp = c->w[1 - c->cur];  // the gather target is free between sample and resample
  hipLaunchKernelGGL(scale_by_kernel, dim3(g), dim3(256), 0, c->stream, c->w[c->cur], tmp, n, (double)n);
  MCL_TRY_W(build_exact_cdf(c, tmp, n));
  hipLaunchKernelGGL(copy_last_kernel, dim3(1), dim3(64), 0, c->stream, c->cdf_exact, n, c->scal);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// running sum of the live weights exactly as `s = 0; for w in weights: s += w` produces it
int launch_sequential_total(mclhip_ctx* c) { return build_exact_cdf(c, c->w[c->cur], c->M); }

template <typename T>
__global__ void inject_kernel(const int64_t* __restrict__ rows, const double* __restrict__ poses, int64_t n_inj,
                              T* __restrict__ x, T* __restrict__ y, T* __restrict__ th, double* __restrict__ w,
                              double w_value, int set_weight) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n_inj) return;
  const int64_t m = rows[j];
  x[m] = (T)poses[3 * j];
  y[m] = (T)poses[3 * j + 1];
  th[m] = (T)poses[3 * j + 2];
  if (set_weight) w[m] = w_value;  // Wt.append(1.0/self.M)  (mc_localization2.py:288)
}

int launch_inject(mclhip_ctx* c, const int64_t* d_rows, const double* d_poses, int64_t n_inj, int which_buffer,
                  bool set_weight) {
  if (n_inj <= 0) return MCLHIP_OK;
  const unsigned g = (unsigned)((n_inj + 255) / 256);
  const int b = which_buffer;
  const double wv = 1.0 / (double)c->M;
  if (c->dtype == MCLHIP_F32)
    hipLaunchKernelGGL(inject_kernel<float>, dim3(g), dim3(256), 0, c->stream, d_rows, d_poses, n_inj,
                       (float*)c->px[b], (float*)c->py[b], (float*)c->pth[b], c->w[b], wv, set_weight ? 1 : 0);
  else
    hipLaunchKernelGGL(inject_kernel<double>, dim3(g), dim3(256), 0, c->stream, d_rows, d_poses, n_inj,
                       (double*)c->px[b], (double*)c->py[b], (double*)c->pth[b], c->w[b], wv, set_weight ? 1 : 0);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

size_t irr_record_bytes() { return sizeof(IrrRecord); }
size_t run_base_bytes() { return sizeof(RunBase); }
size_t kf_bytes() { return sizeof(KF); }

}  // namespace mcl
