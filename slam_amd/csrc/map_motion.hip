// Map ingestion (bit planes), beam-constant table, velocity motion sampling and
// the stand-alone beam-mixture kernel.  Compiled with -ffp-contract=off: every
// float64 expression below is meant to round exactly where the reference's
// numpy/Python expression rounds.
#include <algorithm>

#include "mclhip_internal.h"
#include "device_rng.h"

namespace mcl {

// ---------------------------------------------------------------------------
// Map -> two bit planes in 8x8 tiles.  Bit (r*8+c) of tile (ty,tx) describes
// cell (8*ty+r, 8*tx+c).  Obstacle == 0 (mc_localization.py:65), free == 255
// (mc_localization.py:109); anything else is neither.
// ---------------------------------------------------------------------------
__global__ void build_tiles_kernel(const uint8_t* __restrict__ map, int H, int W, int TH, int TW,
                                   uint64_t* __restrict__ obst, uint64_t* __restrict__ freeb) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= TH * TW) return;
  int ty = t / TW, tx = t % TW;
  uint64_t ob = 0, fr = 0;
  for (int r = 0; r < 8; ++r) {
    int row = ty * 8 + r;
    if (row >= H) break;
    for (int c = 0; c < 8; ++c) {
      int col = tx * 8 + c;
      if (col >= W) break;
      uint8_t v = map[(size_t)row * W + col];
      uint64_t bit = 1ull << (r * 8 + c);
      if (v == 0) ob |= bit;
      if (v == 255) fr |= bit;
    }
  }
  obst[t] = ob;
  freeb[t] = fr;
}

int launch_build_tiles(mclhip_ctx* c, const uint8_t* d_map) {
  int n = c->TH * c->TW;
  hipLaunchKernelGGL(build_tiles_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_map, c->H, c->W, c->TH,
                     c->TW, c->obst, c->freeb);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// ---------------------------------------------------------------------------
// Neighbourhood lists.  For every query tile of the grid (kQueryCells^2 cells; the grid extended by the ring
// radius on each side) the obstacle cells that can be within reach of some point of the query tile, nearest
// obstacle tile first: what a particle standing anywhere in that query tile has to look at is then one
// contiguous slice (a multiple of 64 entries long: see the padding at the end of nbr_list_kernel).  The ring table names the 8x8 obstacle tiles around the 8x8 tile the query tile lies in; of
// those, the ones whose nearest cell is out of reach of the query tile are left out as a whole, and of the others
// every cell is tested by itself (a list of whole tiles is half as long again: it covers a disc whose radius is
// larger by a tile's diagonal).
// ---------------------------------------------------------------------------
struct NbrGeom {
  int TW, TH, n_ring, rtx, rty, ETW, ETH;   // obstacle tiles, ring, extended 8x8 tile grid
  double res_x, res_y, reach2, cell_reach2;
  double class_r0, class_step;              // see cell_class
};

// A list is kept in kNeedClasses classes by the cells' distance from the CENTRE of the query tile: class 0 within
// class_r0 (z_max and a little), class j within class_r0 + j * class_step, the last class everything else.  A pose at
// distance rho from that centre can only reach cells within z_max + rho of it (triangle inequality), so it needs the
// classes up to ceil(rho / class_step) and no more: the prefix of the list that holds them.  (Poses near the middle of
// their tile skip the rim that the tile's corners need: 14 % of the batches on the benchmark's map.)
__device__ inline int cell_class(const NbrGeom& g, int qrow0, int qcol0, uint32_t e) {
  const double dx = ((double)(int)(e & 0xffffu) - ((double)qcol0 + 0.5 * kQueryCells)) * g.res_x;
  const double dy = ((double)(int)(e >> 16) - ((double)qrow0 + 0.5 * kQueryCells)) * g.res_y;
  const double over = sqrt(dx * dx + dy * dy) - g.class_r0;
  const int k = over <= 0.0 ? 0 : (int)ceil(over / g.class_step);
  return k > kNeedClasses - 1 ? kNeedClasses - 1 : k;
}

// fewest whole cells between the query rows / columns [q0, q0 + kQueryCells) and the obstacle tile's [8 o, 8 o + 8)
__device__ inline int cell_gap(int q0, int o) {
  const int q1 = q0 + kQueryCells - 1, o0 = o * 8, o1 = o0 + 7;
  return o0 > q1 ? o0 - q1 - 1 : (q0 > o1 ? q0 - o1 : 0);
}

__device__ inline bool in_reach(const NbrGeom& g, int qrow0, int qcol0, int y, int x) {
  const double gx = (double)cell_gap(qcol0, x) * g.res_x, gy = (double)cell_gap(qrow0, y) * g.res_y;
  return gx * gx + gy * gy <= g.reach2;
}

// One obstacle cell (row << 16 | col) against the query tile: a pose of the tile has its coordinate, in cells, in
// [q0, q0 + kQueryCells), the cell stands at its own index; cell_reach2 holds a cell of slack for the rounding of
// both.
__device__ inline bool cell_in_reach(const NbrGeom& g, int qrow0, int qcol0, uint32_t e) {
  const int col = (int)(e & 0xffffu), row = (int)(e >> 16);
  const int dc = col < qcol0 ? qcol0 - col : (col > qcol0 + kQueryCells ? col - (qcol0 + kQueryCells) : 0);
  const int dr = row < qrow0 ? qrow0 - row : (row > qrow0 + kQueryCells ? row - (qrow0 + kQueryCells) : 0);
  const double gx = (double)dc * g.res_x, gy = (double)dr * g.res_y;
  return gx * gx + gy * gy <= g.cell_reach2;
}

// One wavefront per query tile.  FILL == false: the length of its list into cnt, the number of cells of every class
// into cls_cnt; true: the list itself, from nbr_off on, class by class, and how many batches of 64 hold the classes
// up to k into need[k].
template <bool FILL>
__global__ __launch_bounds__(256) void nbr_list_kernel(const uint32_t* __restrict__ tile_off,
                                                       const uint32_t* __restrict__ cells, const uint16_t* __restrict__ ring,
                                                       NbrGeom g, const uint64_t* __restrict__ nbr_off,
                                                       uint32_t* __restrict__ out_buf, uint32_t* __restrict__ cls_cnt,
                                                       QueryTile* __restrict__ need) {
  const int SW = g.ETW << kQueryShift, SH = g.ETH << kQueryShift;
  const int lane = threadIdx.x & 63;
  const int64_t t = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= (int64_t)SW * SH) return;
  const int sy = (int)(t / SW), sx = (int)(t % SW);
  const int ty = (sy >> kQueryShift) - g.rty, tx = (sx >> kQueryShift) - g.rtx;   // the 8x8 tile it lies in
  const int qrow0 = ty * 8 + (sy & ((1 << kQueryShift) - 1)) * kQueryCells;
  const int qcol0 = tx * 8 + (sx & ((1 << kQueryShift) - 1)) * kQueryCells;
  uint64_t out = FILL ? nbr_off[t] : 0ull;   // (FILL: place in the lists; else: the count, which fits 32 bits)
  uint32_t at[kNeedClasses];                  // wave-uniform.  FILL: where the next cell of class k goes; else: their number
#pragma unroll
  for (int k = 0; k < kNeedClasses; ++k) at[k] = 0u;
  if (FILL) {
    uint32_t run = 0u;
#pragma unroll
    for (int k = 0; k < kNeedClasses; ++k) {
      const uint32_t nk = cls_cnt[(size_t)t * kNeedClasses + k];
      at[k] = run;
      run += nk;
      if (lane == 0) { const uint32_t b = (run + 63u) / 64u; need[t].need[k] = (uint16_t)(b > 65535u ? 65535u : b); }
    }
  }
  for (int r = 0; r < g.n_ring; ++r) {
    const int y = ty + (int)(ring[r] >> 8) - 128, x = tx + (int)(ring[r] & 0xffu) - 128;
    if (y < 0 || y >= g.TH || x < 0 || x >= g.TW) continue;
    if (!in_reach(g, qrow0, qcol0, y, x)) continue;
    const uint32_t* to = tile_off + ((size_t)y * g.TW + x);
    const uint32_t o = to[0], n = to[1] - o;   // n <= 64
    if (n == 0) continue;
    const uint32_t e = (uint32_t)lane < n ? cells[o + lane] : 0u;
    const bool keep = (uint32_t)lane < n && cell_in_reach(g, qrow0, qcol0, e);
    const unsigned long long km = __ballot(keep);
    if (km == 0ull) continue;
    const int cls = keep ? cell_class(g, qrow0, qcol0, e) : -1;
#pragma unroll
    for (int k = 0; k < kNeedClasses; ++k) {
      const unsigned long long cm = __ballot(cls == k);
      if (FILL && cls == k) {
        const uint32_t pos = __builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
        out_buf[out + at[k] + pos] = e;
      }
      at[k] += (uint32_t)__popcll(cm);
    }
    if (!FILL) out += (uint32_t)__popcll(km);
  }
  if (!FILL) {
    if (lane == 0) out_buf[t] = (uint32_t)out;
#pragma unroll
    for (int k = 0; k < kNeedClasses; ++k)
      if (lane == k) cls_cnt[(size_t)t * kNeedClasses + k] = at[k];
  }
  if (FILL) {
    out += at[kNeedClasses - 1];              // (the last class's end: the list's)
    if (lane == 0) { need[t].first = nbr_off[t]; need[t].n_cells = (uint32_t)(nbr_off[t + 1] - nbr_off[t]); need[t].pad = 0u; }
    // A list occupies whole batches of 64 entries; the rest of the last batch is a cell no pose of the query tile can
    // reach (the far end of the column range: reach is at most 127 tiles), so that a walk needs no per-lane bounds.
    const uint32_t pad = qcol0 < 32768 ? 0xffffu : 0u;   // row 0, last or first column
    for (uint64_t i = out + (uint64_t)lane; i < nbr_off[t + 1]; i += 64u) out_buf[i] = pad;
  }
}

static NbrGeom nbr_geom(const mclhip_ctx* c) {
  NbrGeom g;
  g.TW = c->TW; g.TH = c->TH; g.n_ring = c->n_ring; g.rtx = c->rtx; g.rty = c->rty; g.ETW = c->ETW; g.ETH = c->ETH;
  g.res_x = c->res_x; g.res_y = c->res_y;
  const double reach = c->z_max + 2.0 * (c->res_x + c->res_y);   // the ring table's own margin
  g.reach2 = reach * reach;
  const double cell_reach = c->z_max + (c->res_x + c->res_y);
  g.cell_reach2 = cell_reach * cell_reach;
  g.class_r0 = c->z_max + nbr_class_slack(c);
  g.class_step = nbr_class_step(c);
  return g;
}

int launch_nbr_count(mclhip_ctx* c, uint32_t* d_cnt, uint32_t* d_cls_cnt) {
  const int64_t ne = ((int64_t)c->ETW << kQueryShift) * ((int64_t)c->ETH << kQueryShift);
  hipLaunchKernelGGL(nbr_list_kernel<false>, dim3((unsigned)((ne + 3) / 4)), dim3(256), 0, c->stream, c->tile_off, c->cells,
                     c->ring, nbr_geom(c), (const uint64_t*)nullptr, d_cnt, d_cls_cnt, (QueryTile*)nullptr);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

int launch_nbr_fill(mclhip_ctx* c, uint32_t* d_cls_cnt) {
  const int64_t ne = ((int64_t)c->ETW << kQueryShift) * ((int64_t)c->ETH << kQueryShift);
  hipLaunchKernelGGL(nbr_list_kernel<true>, dim3((unsigned)((ne + 3) / 4)), dim3(256), 0, c->stream, c->tile_off, c->cells,
                     c->ring, nbr_geom(c), c->nbr_off, c->nbr_cells, d_cls_cnt, c->nbr_need);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// ---------------------------------------------------------------------------
// Per-beam constants that depend only on the measured range z*_k:
//   etalam_k = 1/(1 - exp(-lambda*z*_k)) * lambda      (range_sensor.py:49-50)
// ---------------------------------------------------------------------------
__global__ void beam_table_kernel(const double* __restrict__ zmeas, const uint16_t* __restrict__ perm, int B,
                                  double lambda, double* __restrict__ zt, double* __restrict__ etalam) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;  // position in sorted-angle order
  if (k >= B) return;
  double z = zmeas[perm[k]];
  double eta = 1.0 / (1.0 - exp(-lambda * z));
  zt[k] = z;
  etalam[k] = eta * lambda;
}

// histogram filter: the measured range is the first argument of compute_px there, so what is
// constant per table entry is exp(-lambda * z_meas) (range_sensor.py:50); it goes where the
// particle filter keeps eta * lambda
__global__ void grid_beam_table_kernel(const double* __restrict__ zmeas, int B0, int NE, double lambda,
                                       double* __restrict__ zt, double* __restrict__ eshort) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;   // entry e = heading * B0 + beam
  if (e >= NE) return;
  const double z = zmeas[e % B0];
  zt[e] = z;
  eshort[e] = exp(-lambda * z);
}

int launch_grid_beam_table(mclhip_ctx* c, const double* zmeas, int B0, int NE, double* zt, double* eshort) {
  hipLaunchKernelGGL(grid_beam_table_kernel, dim3((NE + 63) / 64), dim3(64), 0, c->stream, zmeas, B0, NE, c->lambda, zt,
                     eshort);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

int launch_beam_table(mclhip_ctx* c) {
  hipLaunchKernelGGL(beam_table_kernel, dim3((c->B + 63) / 64), dim3(64), 0, c->stream, c->zmeas_stage, c->perm, c->B,
                     c->lambda, c->zt, c->etalam);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// Free-cell indicator of mc_localization.py:106-112: int(x/res + 0.5) truncates
// toward zero; the cell must hold exactly 255.
__device__ inline bool on_free_cell(double x, double y, const MotionParams& mp) {
  double fu = x / mp.res_x + 0.5;
  double fv = y / mp.res_y + 0.5;
  if (!(fu > -1.0 && fu < (double)mp.W && fv > -1.0 && fv < (double)mp.H)) return false;  // also rejects NaN
  int u = (int)fu, v = (int)fv;  // C conversion truncates toward zero, like Python's int()
  if (u < 0 || u >= mp.W || v < 0 || v >= mp.H) return false;
  uint64_t t = mp.freeb[(size_t)(v >> 3) * mp.TW + (u >> 3)];
  return (t >> (((v & 7) << 3) | (u & 7))) & 1ull;
}

template <typename T>
__global__ __launch_bounds__(256) void motion_kernel(T* __restrict__ xs, T* __restrict__ ys, T* __restrict__ ths,
                                                     uint8_t* __restrict__ pm, int64_t n, MotionParams mp,
                                                     const int16_t* __restrict__ noise) {
  int64_t blk = blockIdx.x;
  if (mp.prep.zt) {
    // block 0: the measurement's per-beam constants, as beam_table_kernel forms them, and the counts the sector query
    // adds to (see StepPrep)
    if (blk == 0) {
      const StepPrep& sp = mp.prep;
      for (int k = threadIdx.x; k < sp.B; k += blockDim.x) {
        const double z = sp.zmeas[sp.perm[k]];
        const double eta = 1.0 / (1.0 - exp(-sp.lambda * z));
        sp.zt[k] = z;
        sp.etalam[k] = eta * sp.lambda;
      }
      if (sp.clear && (int)threadIdx.x < sp.n_clear) sp.clear[threadIdx.x] = 0u;
      return;
    }
    blk -= 1;
  }
  int64_t i = blk * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = (double)xs[i], y = (double)ys[i], th = (double)ths[i];
  double v_hat = mp.v, w_hat = mp.omega, g_hat = 0.0;
  if (mp.noise_mode != 0) {
    int k[36];
    if (mp.noise_mode == 1) {
      const int16_t* src = noise + i * 36;
#pragma unroll
      for (int j = 0; j < 36; ++j) k[j] = (int)src[j];
    } else {
#pragma unroll
      for (int b = 0; b < 9; ++b) {
        const uint64_t gi = (uint64_t)(mp.index0 + i);
        uint32_t ctr[4] = {(uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)b, 0x4d434cu};
        philox4x32(ctr, (uint32_t)mp.seed, (uint32_t)(mp.seed >> 32));
#pragma unroll
        for (int j = 0; j < 4; ++j) k[b * 4 + j] = (int)(((uint64_t)ctr[j] * (uint64_t)mp.ints_span) >> 32) - 1000;
      }
    }
    v_hat = mp.v + mp.c1 * sum12(k);          // mc_localization.py:89
    w_hat = mp.omega + mp.c2 * sum12(k + 12); // :90
    g_hat = mp.c3 * sum12(k + 24);            // :91
  }
  double nx, ny, nt;
  // (sine and cosine of an angle together: one argument reduction for the two, the library's values for each)
  double s0, c0;
  sincos(th, &s0, &c0);
  if (mp.book || fabs(w_hat) > 0.0) {  // mc_localization.py:97-100; the book's form divides by w_hat whatever it is
    double q = v_hat / w_hat;
    double wd = w_hat * mp.dt;
    double s1, c1;
    sincos(th + wd, &s1, &c1);
    nx = x - q * s0 + q * s1;
    ny = y + q * c0 - q * c1;
    nt = th + wd + g_hat * mp.dt;
  } else {  // :101-104
    nx = x + v_hat * c0 * mp.dt;
    ny = y + v_hat * s0 * mp.dt;
    nt = th + g_hat * mp.dt;
  }
  T sx = (T)nx, sy = (T)ny, st = (T)nt;
  xs[i] = sx;
  ys[i] = sy;
  ths[i] = st;
  const bool fr = !mp.book && on_free_cell((double)sx, (double)sy, mp);
  pm[i] = fr ? 1 : 0;
  if (mp.qp.out) {
    // the particle's record for the interval-only sector query (see QueryRecord): its tile's slice of neighbourhood
    // cells -- the tile from a float32 product, off by one at an exact tile edge at worst, which the lists' cell of
    // slack covers -- and whether its heading is one that path serves (also false for NaN / infinite headings)
    const QueryPrep& q = mp.qp;
    QueryRecord rec;
    rec.x = (float)sx; rec.y = (float)sy; rec.th = (float)st;
    // (reduced to [0, 1] in float64: a heading many turns away would leave a float32 quotient no fraction bits)
    const double turns = ((double)rec.th + (double)q.origin) * 0.15915494309189535;
    rec.tho = (float)(turns - floor(turns));
    const bool circular = fabsf(rec.th - q.th_mid) <= q.th_hi;
    const bool served = circular || (q.serve_all && fabsf(rec.th) < INFINITY);
    const float fxc = rec.x * q.inv_res_xf, fyc = rec.y * q.inv_res_yf;
    const bool pose_ok = (fabsf(fxc) < 1.0e6f) && (fabsf(fyc) < 1.0e6f);   // far outside any map, or NaN: no tile
    uint64_t first = 0ull;
    uint32_t n_cells = 0u;
    if (pose_ok && served) {
      const int ex = (((int)floorf(fxc)) >> (3 - kQueryShift)) + q.rtx, ey = (((int)floorf(fyc)) >> (3 - kQueryShift)) + q.rty;
      if (ex >= 0 && ex < q.ETW && ey >= 0 && ey < q.ETH) {
        const size_t tile = (size_t)ey * q.ETW + ex;
        const uint4* src_ = reinterpret_cast<const uint4*>(q.nbr_need + tile);     // (one 32-byte read)
        const uint4 t0 = src_[0], t1 = src_[1];
        first = (uint64_t)t0.x | ((uint64_t)t0.y << 32);
        n_cells = t0.z;
        // of the list only the classes this pose can reach (see cell_class): by its distance from the centre of the tile
        // it was filed under
        const float cx = (float)(((ex - q.rtx) * kQueryCells) + kQueryCells / 2), cy = (float)(((ey - q.rty) * kQueryCells) + kQueryCells / 2);
        const float dxm = (fxc - cx) * q.res_xf, dym = (fyc - cy) * q.res_yf;
        const float rho = sqrtf(dxm * dxm + dym * dym);
        int k = (int)ceilf((rho + q.need_slack) * q.need_inv_step);
        k = k < 0 ? 0 : (k > kNeedClasses - 1 ? kNeedClasses - 1 : k);
        const uint64_t half = k < 4 ? ((uint64_t)t1.x | ((uint64_t)t1.y << 32)) : ((uint64_t)t1.z | ((uint64_t)t1.w << 32));
        const uint32_t part = 64u * (uint32_t)((half >> (16 * (k & 3))) & 0xffffull);
        n_cells = part < n_cells ? part : n_cells;
      }
    }
    rec.first_lo = (uint32_t)first; rec.first_hi = (uint32_t)(first >> 32); rec.n_cells = n_cells;
    rec.flags = (fr ? 1u : 0u) | (circular ? 2u : 0u) | (served && !circular ? 4u : 0u);
    uint4* dst = reinterpret_cast<uint4*>(&q.out[i]);
    const uint4* src = reinterpret_cast<const uint4*>(&rec);
    dst[0] = src[0];
    dst[1] = src[1];
  }
}

// Initial particle cloud drawn on the device (mc_localization.py:173-219): rejection sampling of
// (x, y) until the pose rounds to a free cell -- uniform over [0, ext_x) x [0, ext_y) with heading
// U[0, 2pi), or normal around x0 with sigma 0.1 * extent and heading N(x0.theta, pi).  The draws
// come from the counter-based generator keyed by the global particle index, not from numpy's
// stream, so the cloud has the reference's distribution, not its values.
struct InitParams {
  int gaussian;
  double ext_x, ext_y, x0, y0, th0;
  uint64_t seed;
  int max_tries;
};

__device__ inline double uniform01(uint32_t hi, uint32_t lo) {   // 53-bit uniform in [0, 1)
  return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

template <typename T>
__global__ __launch_bounds__(256) void init_particles_kernel(T* __restrict__ xs, T* __restrict__ ys, T* __restrict__ ths,
                                                             int64_t n, MotionParams mp, InitParams ip,
                                                             uint32_t* __restrict__ failures) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t gi = (uint64_t)(mp.index0 + i);
  double x = 0.0, y = 0.0, a = 0.0;
  bool ok = false;
  for (int t = 0; t < ip.max_tries && !ok; ++t) {
    uint32_t c[4] = {(uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)t, 0x494e4954u};
    philox4x32(c, (uint32_t)ip.seed, (uint32_t)(ip.seed >> 32));
    uint32_t d[4] = {(uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)t, 0x494e4955u};
    philox4x32(d, (uint32_t)ip.seed, (uint32_t)(ip.seed >> 32));
    const double u1 = uniform01(c[0], c[1]), u2 = uniform01(c[2], c[3]);
    const double u3 = uniform01(d[0], d[1]), u4 = uniform01(d[2], d[3]);
    if (ip.gaussian) {   // two Box-Muller pairs: (x, y) and the heading
      const double r1 = sqrt(-2.0 * log(1.0 - u1)), r2 = sqrt(-2.0 * log(1.0 - u3));
      x = r1 * cos(2.0 * M_PI * u2) * ip.ext_x * 0.1 + ip.x0;
      y = r1 * sin(2.0 * M_PI * u2) * ip.ext_y * 0.1 + ip.y0;
      a = r2 * cos(2.0 * M_PI * u4) * M_PI + ip.th0;
    } else {
      x = u1 * ip.ext_x;
      y = u2 * ip.ext_y;
      a = u3 * M_PI * 2.0;
    }
    ok = on_free_cell(x, y, mp);
  }
  if (!ok) atomicAdd(failures, 1u);
  xs[i] = (T)x; ys[i] = (T)y; ths[i] = (T)a;
}

// Random-pose injection of augmented MCL drawn on the device (mc_localization2.py:266-288): position m is
// injected when its uniform draw is <= p_random; the pose is uniform over [0, ext_x) x [0, ext_y) on a free
// cell (value 255), heading U[0, 2pi).  The draws are a function of (global position, seed) alone, so the
// two passes -- poses written over X[m] in place before the walk reads X, and pose + weight 1/M written to
// the resampled set afterwards -- see the same values without storing them.
template <typename T>
__global__ __launch_bounds__(256) void inject_device_kernel(T* __restrict__ xs, T* __restrict__ ys, T* __restrict__ ths,
                                                            double* __restrict__ w, int64_t n, MotionParams mp, InitParams ip,
                                                            double p_random, double w_value, int set_weight,
                                                            uint32_t* __restrict__ counters) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t gi = (uint64_t)(mp.index0 + i);
  uint32_t s[4] = {(uint32_t)gi, (uint32_t)(gi >> 32), 0u, 0x494e4a30u};
  philox4x32(s, (uint32_t)ip.seed, (uint32_t)(ip.seed >> 32));
  if (!(uniform01(s[0], s[1]) <= p_random)) return;          // :267
  double x = 0.0, y = 0.0;
  bool ok = false;
  for (int t = 0; t < ip.max_tries && !ok; ++t) {             // :271-280
    uint32_t c[4] = {(uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)t, 0x494e4a31u};
    philox4x32(c, (uint32_t)ip.seed, (uint32_t)(ip.seed >> 32));
    x = uniform01(c[0], c[1]) * ip.ext_x;
    y = uniform01(c[2], c[3]) * ip.ext_y;
    ok = on_free_cell(x, y, mp);
  }
  if (!ok) { atomicAdd(&counters[1], 1u); return; }
  const double a = uniform01(s[2], s[3]) * M_PI * 2.0;       // :281
  xs[i] = (T)x; ys[i] = (T)y; ths[i] = (T)a;
  if (set_weight) w[i] = w_value;                             // :288
  else atomicAdd(&counters[0], 1u);
}

int launch_inject_device(mclhip_ctx* c, int which_buffer, const MotionParams& mp, double p_random, double ext_x,
                         double ext_y, uint64_t seed, bool set_weight, uint32_t* d_counters) {
  InitParams ip;
  ip.gaussian = 0; ip.ext_x = ext_x; ip.ext_y = ext_y; ip.x0 = ip.y0 = ip.th0 = 0.0;
  ip.seed = seed; ip.max_tries = 4096;
  const int b = which_buffer;
  const int64_t n = c->M;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  const double wv = 1.0 / (double)c->M;
  if (c->dtype == MCLHIP_F32)
    hipLaunchKernelGGL(inject_device_kernel<float>, grid, block, 0, c->stream, (float*)c->px[b], (float*)c->py[b],
                       (float*)c->pth[b], c->w[b], n, mp, ip, p_random, wv, set_weight ? 1 : 0, d_counters);
  else
    hipLaunchKernelGGL(inject_device_kernel<double>, grid, block, 0, c->stream, (double*)c->px[b], (double*)c->py[b],
                       (double*)c->pth[b], c->w[b], n, mp, ip, p_random, wv, set_weight ? 1 : 0, d_counters);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

int launch_init_particles(mclhip_ctx* c, void* x, void* y, void* th, int64_t n, const MotionParams& mp, int gaussian,
                          double ext_x, double ext_y, const double* x0, uint64_t seed, uint32_t* d_failures) {
  InitParams ip;
  ip.gaussian = gaussian; ip.ext_x = ext_x; ip.ext_y = ext_y;
  ip.x0 = x0 ? x0[0] : 0.0; ip.y0 = x0 ? x0[1] : 0.0; ip.th0 = x0 ? x0[2] : 0.0;
  ip.seed = seed; ip.max_tries = 4096;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (c->dtype == MCLHIP_F32)
    hipLaunchKernelGGL(init_particles_kernel<float>, grid, block, 0, c->stream, (float*)x, (float*)y, (float*)th, n, mp, ip, d_failures);
  else
    hipLaunchKernelGGL(init_particles_kernel<double>, grid, block, 0, c->stream, (double*)x, (double*)y, (double*)th, n, mp, ip, d_failures);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// the particle set re-loaded from its snapshot: three state arrays and the weights in one launch
template <typename T>
__global__ __launch_bounds__(256) void restore_kernel(T* __restrict__ x, T* __restrict__ y, T* __restrict__ th,
                                                      double* __restrict__ w, const T* __restrict__ sx,
                                                      const T* __restrict__ sy, const T* __restrict__ sth,
                                                      const double* __restrict__ sw, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    x[i] = sx[i]; y[i] = sy[i]; th[i] = sth[i]; w[i] = sw[i];
  }
}

int launch_restore(mclhip_ctx* c, void* x, void* y, void* th, double* w, const void* sx, const void* sy, const void* sth,
                   const double* sw, int64_t n, int dtype) {
  if (n <= 0) return MCLHIP_OK;
  dim3 grid((unsigned)std::min<int64_t>((n + 255) / 256, 8192)), block(256);
  if (dtype == MCLHIP_F32)
    hipLaunchKernelGGL(restore_kernel<float>, grid, block, 0, c->stream, (float*)x, (float*)y, (float*)th, w, (const float*)sx,
                       (const float*)sy, (const float*)sth, sw, n);
  else
    hipLaunchKernelGGL(restore_kernel<double>, grid, block, 0, c->stream, (double*)x, (double*)y, (double*)th, w,
                       (const double*)sx, (const double*)sy, (const double*)sth, sw, n);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

int launch_motion(mclhip_ctx* c, void* x, void* y, void* th, uint8_t* pm, int64_t n, const MotionParams& mp,
                  const int16_t* d_noise, int dtype) {
  if (n <= 0) return MCLHIP_OK;
  dim3 grid((unsigned)((n + 255) / 256 + (mp.prep.zt ? 1 : 0))), block(256);
  if (dtype == MCLHIP_F32)
    hipLaunchKernelGGL(motion_kernel<float>, grid, block, 0, c->stream, (float*)x, (float*)y, (float*)th, pm, n, mp,
                       d_noise);
  else
    hipLaunchKernelGGL(motion_kernel<double>, grid, block, 0, c->stream, (double*)x, (double*)y, (double*)th, pm, n,
                       mp, d_noise);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

// ---------------------------------------------------------------------------
// Beam mixture p(z | z*) for explicit (z, z*) pairs (range_sensor.py:95-112).
// ---------------------------------------------------------------------------
__global__ void compute_px_kernel(const double* __restrict__ z, const double* __restrict__ zt, int64_t n,
                                  RayParams P, double* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = z[i], t = zt[i];
  bool in = (0.0 <= x) && (x <= P.z_max);
  double d = x - t;
  double p_hit = in ? P.c_hit * exp(-0.5 * (d * d) / P.sigma2) : 0.0;
  double p_short = 0.0;
  if (0.0 <= x && x <= t) {
    double eta = 1.0 / (1.0 - exp(-P.lambda * t));
    p_short = eta * P.lambda * exp(-P.lambda * x);
  }
  double p_max = (fabs(x - P.z_max) < 1e-6) ? 1.0 : 0.0;
  double p_rand = in ? P.p_rand : 0.0;
  out[i] = p_hit * P.w_hit + p_short * P.w_short + p_max * P.w_max + p_rand * P.w_rand;
}

int launch_compute_px(mclhip_ctx* c, const double* z, const double* zt, int64_t n, double* p) {
  if (n <= 0) return MCLHIP_OK;
  RayParams P = make_ray_params(c);
  hipLaunchKernelGGL(compute_px_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, z, zt, n, P, p);
  MCL_HIP(c, hipGetLastError());
  return MCLHIP_OK;
}

}  // namespace mcl
