// C ABI of libmclhip.so (declared in include/mclhip.h): argument checking, host
// <-> device staging and the order in which the stage kernels are enqueued.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <vector>

#include "mclhip_internal.h"

namespace mcl {

static thread_local std::string g_create_err;

int fail(mclhip_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg; else g_create_err = msg;
  return code;
}

size_t irr_record_bytes();
size_t run_base_bytes();
size_t kf_bytes();

// ---- device memory of mclhip_raycast_static ----------------------------------------------------------
// The stage query needs staging on the device beside the caller's host buffers: the poses as passed,
// their structure-of-arrays form, and the ranges / hit cells before they are copied back.  mclhip_raycast
// allocates and frees that per call; mclhip_raycast_static keeps it in the code object instead.  Every GPU
// the library runs on holds its own copy of the array (960 KiB) at its own address until the process ends.
constexpr int64_t kStaticMaxPoses = 4096;      // poses per call
constexpr int64_t kStaticMaxEntries = 65536;   // poses x beams per call
constexpr size_t kStaticOffAos = 0;                                                          // [n][3] float64
constexpr size_t kStaticOffX = kStaticOffAos + (size_t)kStaticMaxPoses * 3 * sizeof(double);   // [n] float64 each
constexpr size_t kStaticOffY = kStaticOffX + (size_t)kStaticMaxPoses * sizeof(double);
constexpr size_t kStaticOffTh = kStaticOffY + (size_t)kStaticMaxPoses * sizeof(double);
constexpr size_t kStaticOffRanges = kStaticOffTh + (size_t)kStaticMaxPoses * sizeof(double);   // [n][B] float64
constexpr size_t kStaticOffHits = kStaticOffRanges + (size_t)kStaticMaxEntries * sizeof(double);   // [n][B] int32
constexpr size_t kStaticBytes = kStaticOffHits + (size_t)kStaticMaxEntries * sizeof(int32_t);
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

// In host code `scratch` is only a handle of the symbol, not an address a kernel or a copy can use: the
// address of each device's copy is looked up once per device (with that device current) and kept here.
constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;                    // guards the two tables below
static unsigned char* g_scratch_ptr[kMaxDevices] = {};
static hipError_t g_scratch_err[kMaxDevices] = {};
// One caller at a time stages through a device's array: contexts on one GPU share it, and a second thread's poses
// would otherwise land between the first one's upload and its read-back.  Held for the whole of a call.
static std::mutex g_scratch_call_mutex[kMaxDevices];

// called from mclhip_create with the context's device current
static void resolve_static_scratch(int device) {
  if (device < 0 || device >= kMaxDevices) return;
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[device]) return;
  void* p = nullptr;
  const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
  g_scratch_err[device] = e;
  if (e == hipSuccess) g_scratch_ptr[device] = static_cast<unsigned char*>(p);
}

__global__ void split_poses_kernel(const double* __restrict__ aos, int64_t n, double* __restrict__ x,
                                   double* __restrict__ y, double* __restrict__ th) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  x[i] = aos[3 * i];
  y[i] = aos[3 * i + 1];
  th[i] = aos[3 * i + 2];
}

RayParams make_ray_params(const mclhip_ctx* c) {
  RayParams P;
  P.nbr_off = c->nbr_off;
  P.nbr_cells = c->nbr_cells;
  P.ETW = c->ETW << kQueryShift; P.ETH = c->ETH << kQueryShift; P.rtx = c->rtx << kQueryShift; P.rty = c->rty << kQueryShift;
  P.W = c->W;
  P.res_xf = (float)c->res_x; P.res_yf = (float)c->res_y;
  P.res_x = c->res_x; P.res_y = c->res_y;
  P.z_max = c->z_max;
  P.r2max = std::pow(c->z_max, 2.0);  // max_sensor_range**2 (mc_localization.py:127)
  P.B = c->B; P.NB = c->NB;
  P.tab_sorted = c->tab_sorted; P.perm = c->perm; P.lut = c->lut;
  P.ts_min = c->ts_min; P.ts_max = c->ts_max; P.inv_bw = c->inv_bw;
  P.relx = c->relx; P.G = c->G; P.ext_min = c->ext_min;
  P.iv_NE = c->iv_NE; P.iv_NB = c->iv_NB; P.iv_cmax = c->iv_cmax; P.iv_cmin = c->iv_cmin; P.iv_stride = c->iv_stride; P.iv_edge = c->iv_edge; P.iv_beam = c->iv_beam; P.iv_lut = c->iv_lut; P.iv_ends = c->iv_ends;
  P.iv_origin = c->iv_origin; P.iv_th_hi = c->iv_th_hi; P.iv_th_mid = c->iv_th_mid;
  P.half = c->half;
  P.two_pi_lo = M_PI * 2 - c->half;  // mc_localization.py:149
  P.two_pi_hi = M_PI * 2 + c->half;
  P.zt = c->zt; P.etalam = c->etalam;
  P.sigma2 = std::pow(c->sigma, 2.0);                    // s_hit**2 (range_sensor.py:37)
  P.c_hit = 1 / std::sqrt(2 * M_PI * P.sigma2);          // 1/np.sqrt(2*np.pi*s_hit**2)
  P.lambda = c->lambda;
  P.w_hit = c->w4[0]; P.w_short = c->w4[1]; P.w_max = c->w4[2]; P.w_rand = c->w4[3];
  P.p_rand = 1 / c->z_max;                               // range_sensor.py:64
  P.redo_list = nullptr; P.redo_count = nullptr; P.list = nullptr; P.list_count = nullptr; P.sub_cap = 0;
  P.hint_out = nullptr; P.hint_src = nullptr;
  P.range_noise = nullptr;
  P.rn_sigma = 0.0; P.rn_seed = 0; P.rn_index0 = c->first_index;
  P.stamps = c->stamps;
  P.g_cell = nullptr; P.g_pk_pred = nullptr; P.g_pk = nullptr; P.g_D = 0; P.g_B0 = 0;
  P.g_NE = 0; P.g_cls = nullptr; P.g_cls_first = nullptr; P.g_member = nullptr; P.g_ar = nullptr; P.g_thd = nullptr;
  P.no_obstacles = c->n_obstacles == 0 ? 1 : 0;
  return P;
}

// Tile offsets (in 8x8-cell tiles, relative to the tile that holds the particle) that can contain a
// cell within z_max of some point of that tile, sorted nearest first.
int build_ring_table(mclhip_ctx* c) {
  if (!c->ring_dirty) return MCLHIP_OK;
  if (!c->obst || c->B == 0) return MCLHIP_OK;  // needs both the cell size and z_max
  const int rcx = (int)std::ceil(c->z_max / c->res_x) + 2, rcy = (int)std::ceil(c->z_max / c->res_y) + 2;
  const int rtx = (rcx + 7) / 8 + 1, rty = (rcy + 7) / 8 + 1;
  if (rtx > 127 || rty > 127) return fail(c, MCLHIP_E_LIMIT, "z_max spans more than 127 tiles of 8 cells");
  const double reach = c->z_max + 2.0 * (c->res_x + c->res_y);
  std::vector<std::pair<double, uint16_t>> ent;
  for (int dy = -rty; dy <= rty; ++dy)
    for (int dx = -rtx; dx <= rtx; ++dx) {
      const int gx = dx == 0 ? 0 : (std::abs(dx) - 1) * 8 + 1;  // fewest cells between the two tiles
      const int gy = dy == 0 ? 0 : (std::abs(dy) - 1) * 8 + 1;
      const double d2 = std::pow(gx * c->res_x, 2.0) + std::pow(gy * c->res_y, 2.0);
      if (d2 <= reach * reach) ent.push_back({d2, (uint16_t)(((dy + 128) << 8) | (dx + 128))});
    }
  std::stable_sort(ent.begin(), ent.end(), [](const std::pair<double, uint16_t>& a, const std::pair<double, uint16_t>& b) { return a.first < b.first; });
  if (ent.size() > 4096) return fail(c, MCLHIP_E_LIMIT, "z_max / cell size too large for the tile table");
  std::vector<uint16_t> ring(ent.size());
  for (size_t i = 0; i < ent.size(); ++i) ring[i] = ent[i].second;
  if (c->ring) { (void)hipFree(c->ring); c->ring = nullptr; }
  MCL_HIP(c, hipMalloc((void**)&c->ring, ring.size() * sizeof(uint16_t)));
  MCL_HIP(c, hipMemcpy(c->ring, ring.data(), ring.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  c->n_ring = (int)ring.size();
  // neighbourhood lists over the tile grid extended by the ring radius
  c->rtx = rtx; c->rty = rty;
  c->ETW = c->TW + 2 * rtx; c->ETH = c->TH + 2 * rty;
  const size_t ne = ((size_t)c->ETW << kQueryShift) * ((size_t)c->ETH << kQueryShift);   // query tiles
  uint32_t* d_cnt = nullptr;
  uint32_t* d_cls = nullptr;     // cells per class of every list (between the counting and the filling pass)
  MCL_HIP(c, hipMalloc((void**)&d_cnt, ne * sizeof(uint32_t)));
  if (hipMalloc((void**)&d_cls, ne * kNeedClasses * sizeof(uint32_t)) != hipSuccess) {
    (void)hipFree(d_cnt);
    return fail(c, MCLHIP_E_HIP, "neighbourhood count: out of device memory");
  }
  int rc = launch_nbr_count(c, d_cnt, d_cls);
  std::vector<uint32_t> cnt(ne);
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(cnt.data(), d_cnt, ne * sizeof(uint32_t), hipMemcpyDeviceToHost);
  (void)hipFree(d_cnt);
  struct FreeCls { uint32_t* p; ~FreeCls() { if (p) (void)hipFree(p); } } free_cls{d_cls};
  if (rc != MCLHIP_OK) return rc;
  if (e != hipSuccess) return fail(c, MCLHIP_E_HIP, std::string("neighbourhood count: ") + hipGetErrorString(e));
  std::vector<uint64_t> off(ne + 1);
  uint64_t acc = 0;
  for (size_t t = 0; t < ne; ++t) { off[t] = acc; acc += (cnt[t] + 63u) & ~63ull; }   // whole batches of 64: the fill pads them
  off[ne] = acc;
  if (c->nbr_off) { (void)hipFree(c->nbr_off); c->nbr_off = nullptr; }
  if (c->nbr_cells) { (void)hipFree(c->nbr_cells); c->nbr_cells = nullptr; }
  if (c->nbr_need) { (void)hipFree(c->nbr_need); c->nbr_need = nullptr; }
  // The lists are what the map costs in device memory: four bytes per entry, offsets of eight per query tile.  They
  // are refused, before anything is allocated, when they would not fit what the device has free (less a tenth: the
  // particles come later) -- or MCLHIP_LIST_BUDGET_GB, if the environment sets one.
  {
    const double need = (double)acc * sizeof(uint32_t) + (double)(ne + 1) * sizeof(uint64_t) + (double)ne * sizeof(QueryTile);
    size_t free_b = 0, total_b = 0;
    double budget = 0.0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) budget = 0.9 * (double)free_b;
    if (const char* e = std::getenv("MCLHIP_LIST_BUDGET_GB")) { const double v = std::atof(e) * 1e9; if (v > 0.0 && (budget == 0.0 || v < budget)) budget = v; }
    if (budget > 0.0 && need > budget) {
      char msg[256];
      std::snprintf(msg, sizeof(msg), "neighbourhood lists of this map need %.1f GB of device memory (%.3g entries), %.1f GB are available: "
                    "a coarser map, a smaller z_max or fewer obstacle cells", need / 1e9, (double)acc, budget / 1e9);
      return fail(c, MCLHIP_E_LIMIT, msg);
    }
  }
  MCL_HIP(c, hipMalloc((void**)&c->nbr_off, (ne + 1) * sizeof(uint64_t)));
  MCL_HIP(c, hipMalloc((void**)&c->nbr_need, ne * sizeof(QueryTile)));
  // (kListReadAhead entries behind the last list: a kernel that asks for a list's batches ahead of knowing where it ends
  // may read that far beyond it)
  MCL_HIP(c, hipMalloc((void**)&c->nbr_cells, (size_t)(acc + kListReadAhead) * sizeof(uint32_t)));
  MCL_HIP(c, hipMemsetAsync(c->nbr_cells + acc, 0, kListReadAhead * sizeof(uint32_t), c->stream));
  MCL_HIP(c, hipMemcpy(c->nbr_off, off.data(), (ne + 1) * sizeof(uint64_t), hipMemcpyHostToDevice));
  rc = launch_nbr_fill(c, d_cls);
  if (rc != MCLHIP_OK) return rc;
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  c->ring_dirty = false;
  return MCLHIP_OK;
}

static size_t esize(const mclhip_ctx* c) { return c->dtype == MCLHIP_F32 ? sizeof(float) : sizeof(double); }

template <typename P>
static int dev_alloc(mclhip_ctx* c, P** p, size_t bytes) {
  if (*p) { (void)hipFree(*p); *p = nullptr; }
  if (bytes == 0) bytes = 16;
  MCL_HIP(c, hipMalloc((void**)p, bytes));
  return MCLHIP_OK;
}

#define MCL_TRY(expr) do { int rc__ = (expr); if (rc__ != MCLHIP_OK) return rc__; } while (0)

static int ensure_capacity(mclhip_ctx* c, int64_t M) {
  if (M <= c->cap) return MCLHIP_OK;
  const size_t es = esize(c);
  for (int b = 0; b < 2; ++b) {
    MCL_TRY(dev_alloc(c, &c->px[b], M * es));
    MCL_TRY(dev_alloc(c, &c->py[b], M * es));
    MCL_TRY(dev_alloc(c, &c->pth[b], M * es));
    MCL_TRY(dev_alloc(c, &c->w[b], M * sizeof(double)));
  }
  MCL_TRY(dev_alloc(c, &c->logw, M * sizeof(double)));
  MCL_TRY(dev_alloc(c, &c->pm, M));
  const int64_t nchunks = (M + kSumChunk - 1) / kSumChunk;
  const int64_t ntile = (M + kScanTile - 1) / kScanTile;
  MCL_TRY(dev_alloc(c, &c->chunk_sums, nchunks * sizeof(double)));
  MCL_TRY(dev_alloc(c, &c->cdf_approx, M * sizeof(double)));
  MCL_TRY(dev_alloc(c, &c->cdf_exact, M * sizeof(double)));
  MCL_TRY(dev_alloc(c, &c->kscan, M * sizeof(uint64_t)));
  MCL_TRY(dev_alloc(c, &c->fscan, M * sizeof(uint32_t)));
  MCL_TRY(dev_alloc(c, &c->blk_a, std::max<int64_t>(ntile, 1024) * sizeof(double)));
  MCL_TRY(dev_alloc(c, &c->blk_b, ntile * kf_bytes()));
  MCL_TRY(dev_alloc(c, &c->irr, M * irr_record_bytes()));
  MCL_TRY(dev_alloc(c, &c->bases, M * run_base_bytes()));
  MCL_TRY(dev_alloc(c, &c->idx, M * sizeof(int64_t)));
  MCL_TRY(dev_alloc(c, &c->redo_list, (M + mclhip_ctx::kRedoSlack) * sizeof(int32_t)));
  MCL_TRY(dev_alloc(c, &c->redo_list2, (M + mclhip_ctx::kRedoSlack) * sizeof(int32_t)));
  MCL_TRY(dev_alloc(c, &c->redo_count, 2 * 64 * sizeof(uint32_t)));   // (one set of counts per list)
  c->redo_cap = M;
  if (c->dtype == MCLHIP_F32) MCL_TRY(dev_alloc(c, &c->qrec, (size_t)M * sizeof(QueryRecord)));
  if (c->noise) { (void)hipFree(c->noise); c->noise = nullptr; }
  c->cap = M;
  return MCLHIP_OK;
}

static int use(mclhip_ctx* c) {
  if (!c) return MCLHIP_E_ARG;
  MCL_HIP(c, hipSetDevice(c->device));
  return MCLHIP_OK;
}

// timing: events are recorded around kernel groups and read back after the
// stream has been synchronised
static void span_begin(mclhip_ctx* c, std::vector<mclhip_ctx::Span>& v, int which) {
  c->span_open = false;
  if (!c->timing || (c->timing == 2 && which != 0)) return;
  c->span_open = true;
  mclhip_ctx::Span s; s.which = which;
  (void)hipEventCreate(&s.a); (void)hipEventCreate(&s.b);
  (void)hipEventRecord(s.a, c->stream);
  v.push_back(s);
}
static void span_end(mclhip_ctx* c, std::vector<mclhip_ctx::Span>& v) {
  if (!c->span_open || v.empty()) return;
  c->span_open = false;
  (void)hipEventRecord(v.back().b, c->stream);
}
static void span_collect(mclhip_ctx* c, std::vector<mclhip_ctx::Span>& v) {
  for (auto& s : v) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { c->t_ms[s.which] += ms; c->t_launch[s.which] += 1; }
    (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b);
  }
  v.clear();
}

static MotionParams make_motion_params(const mclhip_ctx* c, double v, double omega, double dt, const double* alpha,
                                       int noise_mode, uint64_t seed) {
  MotionParams mp;
  mp.v = v; mp.omega = omega; mp.dt = dt;
  // b = alpha_i*|v| + alpha_j*|w| ; sample(b) = b/6.0 * sum  (mc_localization.py:24, :89-91)
  mp.c1 = (alpha[0] * std::fabs(v) + alpha[1] * std::fabs(omega)) / 6.0;
  mp.c2 = (alpha[2] * std::fabs(v) + alpha[3] * std::fabs(omega)) / 6.0;
  mp.c3 = (alpha[4] * std::fabs(v) + alpha[5] * std::fabs(omega)) / 6.0;
  mp.res_x = c->res_x; mp.res_y = c->res_y;
  mp.freeb = c->freeb;
  mp.TW = c->TW; mp.TH = c->TH; mp.W = c->W; mp.H = c->H;
  mp.noise_mode = noise_mode;
  mp.ints_span = 2001;   // np.random.random_integers(-1000, 1000) (mc_localization.py:23)
  mp.book = 0;
  mp.seed = seed;
  mp.index0 = c->first_index;
  mp.qp.out = nullptr;
  mp.prep.zt = nullptr;
  return mp;
}

// The motion kernel can leave, per particle, what the interval-only sector query needs (QueryRecord): float32 state,
// a beam table that has sector intervals, neighbourhood lists built.
void make_query_prep(const mclhip_ctx* c, QueryPrep& qp) {
  qp.out = nullptr;
  if (c->dtype != MCLHIP_F32 || c->iv_NE <= 0 || !c->qrec || !c->nbr_off || c->ring_dirty) return;
  qp.out = c->qrec;
  qp.nbr_off = c->nbr_off;
  qp.ETW = c->ETW << kQueryShift; qp.ETH = c->ETH << kQueryShift; qp.rtx = c->rtx << kQueryShift; qp.rty = c->rty << kQueryShift;
  qp.inv_res_xf = (float)(1.0 / c->res_x); qp.inv_res_yf = (float)(1.0 / c->res_y);
  qp.th_mid = (float)c->iv_th_mid; qp.th_hi = (float)c->iv_th_hi;
  qp.origin = (float)c->iv_origin;
  qp.nbr_need = c->nbr_need;
  qp.res_xf = (float)c->res_x; qp.res_yf = (float)c->res_y;
  qp.need_slack = (float)(0.5 * nbr_class_slack(c));   // (half of what the classes' first radius adds to z_max: the other half is the cells')
  qp.need_inv_step = (float)(1.0 / nbr_class_step(c));
  qp.serve_all = ray_serves_every_heading(c) ? 1 : 0;
}

// the same for code outside this file (the sector query marks its own launches)
void timing_begin(mclhip_ctx* c, int which) { span_begin(c, c->spans, which); }
void timing_end(mclhip_ctx* c) { span_end(c, c->spans); }

static int need(mclhip_ctx* c, bool map, bool sensor, bool particles) {
  if (map && !c->obst) return fail(c, MCLHIP_E_STATE, "no map: call mclhip_set_map first");
  if (sensor && c->B == 0) return fail(c, MCLHIP_E_STATE, "no sensor: call mclhip_set_sensor first");
  if (particles && c->M == 0) return fail(c, MCLHIP_E_STATE, "no particles: call mclhip_set_particles first");
  if (map && sensor) return build_ring_table(c);
  return MCLHIP_OK;
}

// enqueue motion + sector query + likelihood for the live particle set
static int enqueue_sample(mclhip_ctx* c, double v, double omega, double dt, const double* alpha, int noise_mode,
                          uint64_t seed, std::vector<mclhip_ctx::Span>& spans, bool want_product,
                          const double* d_range_noise = nullptr, double rn_sigma = 0.0, uint32_t ints_span = 2001) {
  MotionParams mp = make_motion_params(c, v, omega, dt, alpha, noise_mode, seed);
  mp.ints_span = ints_span;
  const int a = c->cur;
  // log-weights from the float32 mixture: the query may run as the interval-only kernel, which reads the motion
  // kernel's records (launch_raycast decides; records it does not use cost 32 bytes per particle of writes)
  if (!want_product && !d_range_noise && c->zmeas_ok) make_query_prep(c, mp.qp);
  // (the measurement's table and the left-over counts are the query's business: one more block of the motion kernel
  // sees to both, see StepPrep)
  if (!c->zsrc || c->zsrc_slot < 0) return fail(c, MCLHIP_E_STATE, "no measurement at hand");
  const int zk = c->zsrc_slot;
  if (c->M <= 0) MCL_HIP(c, hipMemcpyAsync(c->zmeas_stage, c->zsrc, c->B * sizeof(double), hipMemcpyHostToDevice, c->stream));
  mp.prep.zt = c->zt; mp.prep.etalam = c->etalam; mp.prep.zmeas = c->zsrc; mp.prep.perm = c->perm;
  mp.prep.B = c->B; mp.prep.lambda = c->lambda;
  mp.prep.clear = c->redo_count; mp.prep.n_clear = c->redo_count ? 2 * 64 : 0;
  span_begin(c, spans, 1);
  MCL_TRY(launch_motion(c, c->px[a], c->py[a], c->pth[a], c->pm, c->M, mp, c->noise, c->dtype));
  span_end(c, spans);
  if (c->M <= 0) MCL_TRY(launch_beam_table(c));   // (no launch, no extra block)
  else c->redo_cleared = c->redo_count != nullptr;
  MCL_HIP(c, hipEventRecord(c->zev[zk], c->stream));   // the measurement's slot has been read once this has passed
  c->zev_set[zk] = true;
  // (timed inside: group 0 around its kernel, or around the first of its two kernels and group 5 around the second)
  // the raw product is only formed when somebody will read it: it costs one dependent multiply per beam
  MCL_TRY(launch_raycast(c, c->px[a], c->py[a], c->pth[a], c->pm, c->M, c->dtype, want_product ? c->w[a] : nullptr,
                         c->logw, nullptr, nullptr, d_range_noise, rn_sigma, seed, mp.qp.out));
  return MCLHIP_OK;
}

static int upload_zmeas(mclhip_ctx* c, const double* z_meas) {
  if (!z_meas) return fail(c, MCLHIP_E_ARG, "z_meas is NULL");
  const int k = c->zslot;
  c->zslot = (c->zslot + 1) % mclhip_ctx::kZSlots;
  if (!c->zev[k]) MCL_HIP(c, hipEventCreateWithFlags(&c->zev[k], hipEventDisableTiming));
  if (c->zev_set[k]) MCL_HIP(c, hipEventSynchronize(c->zev[k]));   // the slot's previous upload
  double* host = c->zpin + (size_t)k * c->B;
  std::memcpy(host, z_meas, c->B * sizeof(double));
  bool ok = true;
  for (int b = 0; b < c->B; ++b) ok = ok && (z_meas[b] >= 0.0) && (z_meas[b] <= 1.0e9);   // false for NaN as well
  c->zmeas_ok = ok;
  // (no copy: the sample step that follows reads the slot in place -- one block of its motion launch -- and marks the
  // slot as read, see enqueue_sample)
  c->zsrc = host; c->zsrc_slot = k;
  return MCLHIP_OK;
}

static int upload_noise(mclhip_ctx* c, const int16_t* noise_ints) {
  const size_t bytes = (size_t)c->M * 36 * sizeof(int16_t);
  if (!c->noise) MCL_TRY(dev_alloc(c, &c->noise, (size_t)c->cap * 36 * sizeof(int16_t)));
  MCL_HIP(c, hipMemcpyAsync(c->noise, noise_ints, bytes, hipMemcpyHostToDevice, c->stream));
  return MCLHIP_OK;
}

}  // namespace mcl

using namespace mcl;

extern "C" {

int mclhip_abi_version(void) { return 2; }

int mclhip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mclhip_create(mclhip_ctx** out, int device, int dtype) {
  if (!out) return fail(nullptr, MCLHIP_E_ARG, "out is NULL");
  *out = nullptr;
  if (dtype != MCLHIP_F32 && dtype != MCLHIP_F64) return fail(nullptr, MCLHIP_E_ARG, "dtype must be 0 (f32) or 1 (f64)");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0)
    return fail(nullptr, MCLHIP_E_HIP, std::string("no HIP device available: ") + hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(nullptr, MCLHIP_E_ARG, "device index out of range");
  mclhip_ctx* c = new mclhip_ctx();
  c->device = device;
  c->dtype = dtype;
  e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  c->own_stream = c->stream;
  if (e == hipSuccess) e = hipMalloc((void**)&c->scal, 8 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&c->stamps, 16 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemset(c->stamps, 0, 16 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->redo_hint, 128 * sizeof(uint32_t), hipHostMallocDefault);
  if (e == hipSuccess) std::memset(c->redo_hint, 0, 128 * sizeof(uint32_t));
  if (const char* m = std::getenv("MCLHIP_EVERY_HEADING")) c->every_heading = (m[0] == '0') ? 0 : (m[0] == '1' ? 1 : -1);
  if (const char* m = std::getenv("MCLHIP_SPLIT")) c->split_mode = (m[0] == '0') ? 0 : (m[0] == '1' ? 1 : -1);
  if (e == hipSuccess) resolve_static_scratch(device);
  if (e != hipSuccess) {
    fail(nullptr, MCLHIP_E_HIP, std::string("context set-up failed: ") + hipGetErrorString(e));
    delete c;
    return MCLHIP_E_HIP;
  }
  *out = c;
  return MCLHIP_OK;
}

void mclhip_destroy(mclhip_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  fs_release(c);
  void* ptrs[] = {c->obst, c->freeb, c->tile_off, c->cells, c->nbr_off, c->nbr_need, c->nbr_cells, c->tab_sorted, c->tab_orig, c->perm, c->inv_perm, c->lut, c->relx, c->iv_edge, c->iv_beam, c->iv_lut, c->iv_ends, c->redo_list, c->redo_list2, c->redo_count, c->qrec, c->zt, c->etalam, c->zmeas_stage, c->ring,
                  c->px[0], c->px[1], c->py[0], c->py[1], c->pth[0], c->pth[1], c->w[0], c->w[1], c->logw, c->pm,
                  c->sx, c->sy, c->sth, c->sw, c->chunk_sums, c->scal, c->cdf_approx, c->cdf_exact, c->kscan,
                  c->fscan, c->blk_a, c->blk_b, c->irr, c->bases, c->idx, c->noise, c->counters, c->range_noise, c->inj_rows, c->stamps,
                  c->inj_poses, c->g_records, c->g_bases, c->gl_mask, c->gl_x, c->gl_y, c->gl_th, c->gl_cell, c->gl_pk, c->gl_pk_pred, c->gl_K, c->gl_part, c->gl_cls, c->gl_cls_first, c->gl_member, c->gl_mbeta, c->gl_zt, c->gl_el, c->gl_zm,
                  c->rows_send, c->rows_recv, c->rows_self, c->small_d, c->small_i, c->sel};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (c->zpin) (void)hipHostFree(c->zpin);
  if (c->redo_hint) (void)hipHostFree(c->redo_hint);
  for (int k = 0; k < mclhip_ctx::kZSlots; ++k)
    if (c->zev[k]) (void)hipEventDestroy(c->zev[k]);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char* mclhip_last_error(mclhip_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int mclhip_set_map(mclhip_ctx* c, const uint8_t* map, int H, int W, double res_x, double res_y) {
  MCL_TRY(use(c));
  if (!map || H <= 0 || W <= 0) return fail(c, MCLHIP_E_ARG, "map is NULL or empty");
  if (H > 65535 || W > 65535) return fail(c, MCLHIP_E_LIMIT, "map sides are limited to 65535 cells");
  if (!(res_x > 0) || !(res_y > 0)) return fail(c, MCLHIP_E_ARG, "cell size must be positive");
  MCL_HIP(c, hipStreamSynchronize(c->stream));   // the map tables are replaced: nothing enqueued earlier may still read them
  c->H = H; c->W = W; c->TH = (H + 7) / 8; c->TW = (W + 7) / 8;
  c->res_x = res_x; c->res_y = res_y;
  c->ring_dirty = true;
  const size_t nt = (size_t)c->TH * c->TW;
  MCL_TRY(dev_alloc(c, &c->obst, nt * sizeof(uint64_t)));
  MCL_TRY(dev_alloc(c, &c->freeb, nt * sizeof(uint64_t)));
  // per-tile obstacle lists (mc_localization.py:61-71 keeps one flat list; the kernels want the
  // cells grouped by 8x8 tile so that a particle only touches the tiles around it)
  {
    std::vector<uint32_t> off(nt + 1, 0), cells;
    cells.reserve((size_t)H * W / 16);
    for (int ty = 0; ty < c->TH; ++ty)
      for (int tx = 0; tx < c->TW; ++tx) {
        off[(size_t)ty * c->TW + tx] = (uint32_t)cells.size();
        const int r1 = std::min(ty * 8 + 8, H), c1 = std::min(tx * 8 + 8, W);
        for (int r = ty * 8; r < r1; ++r)
          for (int q = tx * 8; q < c1; ++q)
            if (map[(size_t)r * W + q] == 0) cells.push_back(((uint32_t)r << 16) | (uint32_t)q);
      }
    off[nt] = (uint32_t)cells.size();
    c->n_obstacles = (int64_t)cells.size();
    MCL_TRY(dev_alloc(c, &c->tile_off, (nt + 1) * sizeof(uint32_t)));
    MCL_TRY(dev_alloc(c, &c->cells, cells.size() * sizeof(uint32_t)));
    MCL_HIP(c, hipMemcpy(c->tile_off, off.data(), (nt + 1) * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (!cells.empty())
      MCL_HIP(c, hipMemcpy(c->cells, cells.data(), cells.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  uint8_t* d_map = nullptr;
  MCL_HIP(c, hipMalloc((void**)&d_map, (size_t)H * W));
  hipError_t e = hipMemcpyAsync(d_map, map, (size_t)H * W, hipMemcpyHostToDevice, c->stream);
  int rc = MCLHIP_OK;
  if (e == hipSuccess) rc = launch_build_tiles(c, d_map);
  hipError_t e2 = hipStreamSynchronize(c->stream);
  (void)hipFree(d_map);
  if (e != hipSuccess) return fail(c, MCLHIP_E_HIP, std::string("map upload: ") + hipGetErrorString(e));
  if (rc != MCLHIP_OK) return rc;
  if (e2 != hipSuccess) return fail(c, MCLHIP_E_HIP, std::string("map tiling: ") + hipGetErrorString(e2));
  // with map and sensor both known the neighbourhood lists are built here, so that the calls that use them
  // allocate nothing (mclhip_raycast_static relies on it)
  if (c->B > 0) MCL_TRY(build_ring_table(c));
  return MCLHIP_OK;
}

int mclhip_set_sensor(mclhip_ctx* c, const double* angles_deg, int B, double z_max, double sigma_hit,
                      double lambda_short, const double w4[4], double half) {
  MCL_TRY(use(c));
  if (!angles_deg || !w4 || B <= 0) return fail(c, MCLHIP_E_ARG, "angles/w4 NULL or B <= 0");
  if (B > kMaxBeams) return fail(c, MCLHIP_E_LIMIT, "at most 1024 beams");
  for (int k = 0; k < B; ++k)
    if (!std::isfinite(angles_deg[k])) return fail(c, MCLHIP_E_ARG, "beam angles must be finite");
  std::vector<double> tab(B);
  for (int k = 0; k < B; ++k) tab[k] = angles_deg[k] * M_PI / 180.0;  // angle*np.pi/180.0 (mc_localization.py:143)
  MCL_TRY(mcl::set_sensor_table(c, tab.data(), B, z_max, sigma_hit, lambda_short, w4, half));
  if (c->obst) MCL_TRY(build_ring_table(c));   // see mclhip_set_map
  return MCLHIP_OK;
}

}  // extern "C"

// Beam table from angles in radians (the sector query adds the pose heading to them).
int mcl::set_sensor_table(mclhip_ctx* c, const double* tab_rad, int B, double z_max, double sigma_hit,
                          double lambda_short, const double w4[4], double half) {
  if (B > kMaxBeams) return fail(c, MCLHIP_E_LIMIT, "at most 1024 beams");
  if (!(z_max > 0) || !(sigma_hit > 0) || !(half >= 0)) return fail(c, MCLHIP_E_ARG, "bad sensor constants");
  // the tables are replaced with blocking copies: work enqueued by the calls that do not wait (mclhip_step_async,
  // mclhip_dshard_*) must have finished with the old ones
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  c->B = B; c->z_max = z_max; c->sigma = sigma_hit; c->lambda = lambda_short; c->half = half;
  c->ring_dirty = true;
  for (int i = 0; i < 4; ++i) c->w4[i] = w4[i];
  std::vector<double> tab(tab_rad, tab_rad + B), ts(B);
  std::vector<uint16_t> perm(B);
  std::iota(perm.begin(), perm.end(), 0);
  std::stable_sort(perm.begin(), perm.end(), [&](uint16_t a, uint16_t b) { return tab[a] < tab[b]; });
  for (int k = 0; k < B; ++k) ts[k] = tab[perm[k]];
  c->ts_min = ts.front(); c->ts_max = ts.back();
  const double span = c->ts_max - c->ts_min;
  // Extended table for the sector query: beams within two window half-widths of one turn above the
  // smallest angle also appear one turn lower, so that a window that straddles the 0 / 2pi seam of
  // the bearing is one contiguous run of entries.
  const double reach = half + 2.0 * kEdgeEps;
  const double two_pi = M_PI * 2;
  std::vector<float> relx;
  int G = 0;
  if (span < two_pi - 1e-9)   // a table that spans a full turn or more is matched window by window, without images
    for (int k = 0; k < B; ++k)
      if (ts[k] - c->ts_min >= two_pi - 2.0 * reach - 1e-6) ++G;
  if (G > B - 1) G = B > 1 ? B - 1 : 0;   // never an image of every beam
  for (int k = B - G; k < B; ++k) relx.push_back((float)((ts[k] - c->ts_min) - two_pi));
  for (int k = 0; k < B; ++k) relx.push_back((float)(ts[k] - c->ts_min));
  c->G = G;
  c->ext_min = (double)relx.front();
  int NB = 1;
  while (NB < 2 * (B + G)) NB <<= 1;
  c->NB = NB;
  const double ext_span = span - c->ext_min;
  const double bw = ext_span / NB;
  c->inv_bw = (ext_span > 0) ? 1.0 / bw : 0.0;
  std::vector<uint16_t> lut(NB);
  for (int b = 0; b < NB; ++b) {
    const float edge = (float)(c->ext_min + b * bw);
    lut[b] = (uint16_t)(std::lower_bound(relx.begin(), relx.end(), edge) - relx.begin());
  }
  MCL_TRY(dev_alloc(c, &c->relx, relx.size() * sizeof(float)));
  MCL_HIP(c, hipMemcpy(c->relx, relx.data(), relx.size() * sizeof(float), hipMemcpyHostToDevice));
  MCL_TRY(dev_alloc(c, &c->tab_sorted, B * sizeof(double)));
  MCL_TRY(dev_alloc(c, &c->tab_orig, B * sizeof(double)));
  MCL_TRY(dev_alloc(c, &c->perm, B * sizeof(uint16_t)));
  MCL_TRY(dev_alloc(c, &c->inv_perm, B * sizeof(uint16_t)));
  MCL_TRY(dev_alloc(c, &c->lut, NB * sizeof(uint16_t)));
  MCL_TRY(dev_alloc(c, &c->zt, B * sizeof(double)));
  MCL_TRY(dev_alloc(c, &c->etalam, B * sizeof(double)));
  MCL_TRY(dev_alloc(c, &c->zmeas_stage, B * sizeof(double)));
  for (int k = 0; k < mclhip_ctx::kZSlots; ++k)
    if (c->zev_set[k]) { MCL_HIP(c, hipEventSynchronize(c->zev[k])); c->zev_set[k] = false; }
  if (c->zpin) { (void)hipHostFree(c->zpin); c->zpin = nullptr; }
  c->zsrc = nullptr; c->zsrc_slot = -1;
  MCL_HIP(c, hipHostMalloc((void**)&c->zpin, (size_t)mclhip_ctx::kZSlots * B * sizeof(double), hipHostMallocDefault));
  MCL_HIP(c, hipMemcpy(c->tab_sorted, ts.data(), B * sizeof(double), hipMemcpyHostToDevice));
  MCL_HIP(c, hipMemcpy(c->tab_orig, tab.data(), B * sizeof(double), hipMemcpyHostToDevice));
  MCL_HIP(c, hipMemcpy(c->perm, perm.data(), B * sizeof(uint16_t), hipMemcpyHostToDevice));
  {
    std::vector<uint16_t> inv(B);
    for (int k = 0; k < B; ++k) inv[perm[k]] = (uint16_t)k;
    MCL_HIP(c, hipMemcpy(c->inv_perm, inv.data(), B * sizeof(uint16_t), hipMemcpyHostToDevice));
  }
  MCL_HIP(c, hipMemcpy(c->lut, lut.data(), NB * sizeof(uint16_t), hipMemcpyHostToDevice));
  {
    // Sector intervals.  Available when every beam angle lies in [0, 2 pi) and the sectors are proper arcs; edges that
    // agree to 1e-9 rad count as one (a bearing that close to either is decided by the exact test anyway).
    c->iv_NE = 0; c->iv_NB = 0; c->iv_cmax = 0; c->iv_cmin = 0; c->iv_uniform = false; c->iv_stride = 0;
    const bool usable = c->ts_min >= 0.0 && c->ts_max < two_pi && half > 16.0 * kEdgeEps && half < 0.5 * M_PI;
    if (usable) {
      const double origin = c->ts_min - half;
      std::vector<double> all;
      auto reduced = [&](double a) { double v = a - origin; while (v >= two_pi) v -= two_pi; while (v < 0.0) v += two_pi; return v; };
      for (int k = 0; k < B; ++k) { all.push_back(reduced(ts[k] - half)); all.push_back(reduced(ts[k] + half)); }
      std::sort(all.begin(), all.end());
      std::vector<double> edge;
      for (double v : all)
        if (edge.empty() || v - edge.back() > 1e-9) edge.push_back(v);
      if (edge.size() > 1 && two_pi - edge.back() <= 1e-9) edge.pop_back();        // the same edge as 0 from the other side
      // Edges that all fall on multiples of one width (the reference's own beams: every 15 degrees, sectors of +-5,
      // so every edge is a multiple of 5 degrees): the circle is cut into that many equal intervals instead, some of
      // which no sector spans -- a bearing's interval is then arithmetic, not a search (the kernels' "even" layouts)
      if (edge.size() >= 2) {
        double gmin = two_pi - edge.back() + edge.front();
        for (size_t m = 0; m + 1 < edge.size(); ++m) gmin = std::min(gmin, edge[m + 1] - edge[m]);
        const long nn = gmin > 1e-4 ? std::lround(two_pi / gmin) : 0;
        if (nn >= 2 && nn <= 384 && (size_t)nn > edge.size()) {
          bool lattice = true;
          for (double v : edge) {
            const double q = v * (double)nn / two_pi;
            if (std::fabs(q - std::round(q)) * two_pi / (double)nn > 1e-9) lattice = false;
          }
          if (lattice) {
            std::vector<double> even((size_t)nn);
            for (long m = 0; m < nn; ++m) even[(size_t)m] = two_pi * (double)m / (double)nn;
            for (double v : edge) even[(size_t)(std::lround(v * (double)nn / two_pi) % nn)] = v;   // (the sectors' own edges as they are)
            edge.swap(even);
          }
        }
      }
      const int NE = (int)edge.size();
      auto index_of = [&](double v) {            // the merged edge nearest to v on the circle
        int best = 0; double bd = 1e9;
        for (int m = 0; m < NE; ++m) {
          double d = std::fabs(edge[m] - v); d = std::min(d, two_pi - d);
          if (d < bd) { bd = d; best = m; }
        }
        return best;
      };
      bool ok = NE >= 2 && NE <= 1023 && edge[0] <= 1e-9;
      // every interval is much wider than the band around an edge in which float32 does not decide: a bearing is
      // then near one edge at most, and a sector that holds an edge in its interior holds that whole band
      for (int m = 0; m < NE && ok; ++m)
        if ((m + 1 < NE ? edge[m + 1] : two_pi) - edge[m] < 1e-4) ok = false;
      std::vector<uint16_t> beam(B);
      std::vector<uint32_t> ends(NE > 0 ? NE : 1, 0u);
      int cmax = 0, cmin = 1 << 30;
      for (int k = 0; k < B && ok; ++k) {
        const int m0 = index_of(reduced(ts[k] - half)), m1 = index_of(reduced(ts[k] + half));
        const int cnt = (m1 - m0 + NE) % NE;
        if (cnt < 1 || cnt > 63) { ok = false; break; }
        cmax = std::max(cmax, cnt); cmin = std::min(cmin, cnt);
        beam[k] = (uint16_t)(m0 | (cnt << 10));
        // one sector may start and one may end at an edge (repeated beams, or different beams whose edges agree, leave
        // the table to the beam-by-beam test)
        if ((ends[m0] >> 21) & 1u) { ok = false; break; }
        ends[m0] |= ((uint32_t)k << 11) | (1u << 21);
        if ((ends[m1] >> 10) & 1u) { ok = false; break; }
        ends[m1] |= (uint32_t)k | (1u << 10);
        for (int j = 1; j < cnt; ++j) ends[(m0 + j) % NE] |= 1u << 22;
      }
      if (ok) {
        int NBI = 1;
        while (NBI < 2 * NE) NBI <<= 1;   // two-byte entries: the table costs what the four-per-interval one-byte table did
        std::vector<float> ef(NE + 1);
        for (int m = 0; m < NE; ++m) ef[m] = (float)edge[m];
        ef[0] = 0.0f;
        ef[NE] = (float)two_pi;
        std::vector<uint16_t> ilut(NBI);
        for (int b = 0; b < NBI; ++b) {
          const float lower = (float)(two_pi * b / NBI);
          int m = 0;
          while (m + 1 < NE && ef[m + 1] <= lower) ++m;
          ilut[b] = (uint16_t)m;
        }
        MCL_TRY(dev_alloc(c, &c->iv_edge, ef.size() * sizeof(float)));
        MCL_TRY(dev_alloc(c, &c->iv_beam, B * sizeof(uint16_t)));
        MCL_TRY(dev_alloc(c, &c->iv_lut, NBI * sizeof(uint16_t)));
        MCL_TRY(dev_alloc(c, &c->iv_ends, NE * sizeof(uint32_t)));
        MCL_HIP(c, hipMemcpy(c->iv_ends, ends.data(), NE * sizeof(uint32_t), hipMemcpyHostToDevice));
        MCL_HIP(c, hipMemcpy(c->iv_edge, ef.data(), ef.size() * sizeof(float), hipMemcpyHostToDevice));
        MCL_HIP(c, hipMemcpy(c->iv_beam, beam.data(), B * sizeof(uint16_t), hipMemcpyHostToDevice));
        MCL_HIP(c, hipMemcpy(c->iv_lut, ilut.data(), NBI * sizeof(uint16_t), hipMemcpyHostToDevice));
        c->iv_NE = NE; c->iv_NB = NBI; c->iv_cmax = cmax; c->iv_cmin = cmin; c->iv_origin = origin;
        // equally wide intervals (evenly spaced beams): a kernel variant finds a bearing's interval by arithmetic
        c->iv_uniform = true;
        for (int m = 0; m < NE; ++m)
          if (std::fabs(edge[m] - two_pi * m / NE) > 1e-9) c->iv_uniform = false;
        // ... and whose sectors start at interval k * stride (sorted position k): the start needs no table then
        if (c->iv_uniform && NE % B == 0) {
          const int stride = NE / B;
          bool arith = true;
          for (int k = 0; k < B; ++k)
            if ((beam[k] & 0x3ffu) != (unsigned)(k * stride)) arith = false;
          if (arith) c->iv_stride = stride;
        }
        // headings for which no sector leaves the range the reference recognises: the highest beam angle + heading +
        // half-angle stays below two turns, the lowest beam angle + heading - half-angle above minus one turn
        // (kept as centre and half-width: one subtraction and one comparison in the kernel)
        {
          const double hi = 2.0 * two_pi - half - c->ts_max - 1e-3, lo = -two_pi + half - c->ts_min + 1e-3;
          c->iv_th_mid = 0.5 * (hi + lo);
          c->iv_th_hi = 0.5 * (hi - lo) - 1e-6;      // |heading - mid| <= this
        }
      }
    }
  }
  MCL_HIP(c, hipMemset(c->zt, 0, B * sizeof(double)));
  MCL_HIP(c, hipMemset(c->etalam, 0, B * sizeof(double)));
  return MCLHIP_OK;
}

extern "C" {

int mclhip_set_particles(mclhip_ctx* c, const double* X, const double* w, int64_t M) {
  MCL_TRY(use(c));
  if (!X || M <= 0) return fail(c, MCLHIP_E_ARG, "X is NULL or M <= 0");
  if (M > ((int64_t)1 << 31)) return fail(c, MCLHIP_E_LIMIT, "at most 2^31 particles per context");
  MCL_HIP(c, hipStreamSynchronize(c->stream));   // blocking copies below: nothing enqueued earlier may still use the arrays
  MCL_TRY(ensure_capacity(c, M));
  c->M = M;
  c->cur = 0;
  const size_t es = esize(c);
  std::vector<unsigned char> hx(M * es), hy(M * es), ht(M * es);
  if (c->dtype == MCLHIP_F32) {
    float* fx = (float*)hx.data(); float* fy = (float*)hy.data(); float* ft = (float*)ht.data();
    for (int64_t i = 0; i < M; ++i) { fx[i] = (float)X[3 * i]; fy[i] = (float)X[3 * i + 1]; ft[i] = (float)X[3 * i + 2]; }
  } else {
    double* fx = (double*)hx.data(); double* fy = (double*)hy.data(); double* ft = (double*)ht.data();
    for (int64_t i = 0; i < M; ++i) { fx[i] = X[3 * i]; fy[i] = X[3 * i + 1]; ft[i] = X[3 * i + 2]; }
  }
  MCL_HIP(c, hipMemcpy(c->px[0], hx.data(), M * es, hipMemcpyHostToDevice));
  MCL_HIP(c, hipMemcpy(c->py[0], hy.data(), M * es, hipMemcpyHostToDevice));
  MCL_HIP(c, hipMemcpy(c->pth[0], ht.data(), M * es, hipMemcpyHostToDevice));
  std::vector<double> hw;
  if (!w) { hw.assign(M, 1.0 / (double)M); w = hw.data(); }  // np.ones(M)/M (mc_localization.py:41)
  MCL_HIP(c, hipMemcpy(c->w[0], w, M * sizeof(double), hipMemcpyHostToDevice));
  std::vector<double> z(M, 0.0);
  MCL_HIP(c, hipMemcpy(c->logw, z.data(), M * sizeof(double), hipMemcpyHostToDevice));
  return MCLHIP_OK;
}

int mclhip_init_particles(mclhip_ctx* c, int64_t M, const double* x0, double extent_x, double extent_y, uint64_t seed) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, true, false, false));
  if (M <= 0 || !(extent_x > 0) || !(extent_y > 0)) return fail(c, MCLHIP_E_ARG, "M <= 0 or empty extent");
  if (M > ((int64_t)1 << 31)) return fail(c, MCLHIP_E_LIMIT, "at most 2^31 particles per context");
  MCL_HIP(c, hipStreamSynchronize(c->stream));   // the particle arrays may be reallocated
  MCL_TRY(ensure_capacity(c, M));
  c->M = M;
  c->cur = 0;
  const double al[6] = {0, 0, 0, 0, 0, 0};
  MotionParams mp = make_motion_params(c, 0.0, 0.0, 0.0, al, 0, seed);
  if (!c->counters) MCL_TRY(dev_alloc(c, &c->counters, 16 * sizeof(uint32_t)));
  MCL_HIP(c, hipMemsetAsync(c->counters, 0, sizeof(uint32_t), c->stream));
  MCL_TRY(launch_init_particles(c, c->px[0], c->py[0], c->pth[0], M, mp, x0 ? 1 : 0, extent_x, extent_y, x0, seed, c->counters));
  std::vector<double> hw(M, 1.0 / (double)M), z(M, 0.0);   // np.ones(M)/M (mc_localization.py:41)
  MCL_HIP(c, hipMemcpyAsync(c->w[0], hw.data(), M * sizeof(double), hipMemcpyHostToDevice, c->stream));
  MCL_HIP(c, hipMemcpyAsync(c->logw, z.data(), M * sizeof(double), hipMemcpyHostToDevice, c->stream));
  uint32_t failed = 0;
  MCL_HIP(c, hipMemcpyAsync(&failed, c->counters, sizeof(failed), hipMemcpyDeviceToHost, c->stream));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  if (failed) return fail(c, MCLHIP_E_STATE, "no free cell found for some particles (is any cell of the map 255 inside the extent?)");
  return MCLHIP_OK;
}

int mclhip_get_particles(mclhip_ctx* c, double* X, double* w, double* logw) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  const int64_t M = c->M;
  const int a = c->cur;
  if (X) {
    const size_t es = esize(c);
    std::vector<unsigned char> hx(M * es), hy(M * es), ht(M * es);
    MCL_HIP(c, hipMemcpy(hx.data(), c->px[a], M * es, hipMemcpyDeviceToHost));
    MCL_HIP(c, hipMemcpy(hy.data(), c->py[a], M * es, hipMemcpyDeviceToHost));
    MCL_HIP(c, hipMemcpy(ht.data(), c->pth[a], M * es, hipMemcpyDeviceToHost));
    if (c->dtype == MCLHIP_F32) {
      const float* fx = (const float*)hx.data(); const float* fy = (const float*)hy.data(); const float* ft = (const float*)ht.data();
      for (int64_t i = 0; i < M; ++i) { X[3 * i] = fx[i]; X[3 * i + 1] = fy[i]; X[3 * i + 2] = ft[i]; }
    } else {
      const double* fx = (const double*)hx.data(); const double* fy = (const double*)hy.data(); const double* ft = (const double*)ht.data();
      for (int64_t i = 0; i < M; ++i) { X[3 * i] = fx[i]; X[3 * i + 1] = fy[i]; X[3 * i + 2] = ft[i]; }
    }
  }
  if (w) MCL_HIP(c, hipMemcpy(w, c->w[a], M * sizeof(double), hipMemcpyDeviceToHost));
  if (logw) MCL_HIP(c, hipMemcpy(logw, c->logw, M * sizeof(double), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

int64_t mclhip_num_particles(mclhip_ctx* c) { return c ? c->M : 0; }

int mclhip_raycast(mclhip_ctx* c, const double* poses, int64_t n, double* ranges, int32_t* hit_cell) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, true, true, false));
  if (!poses || !ranges || n <= 0) return fail(c, MCLHIP_E_ARG, "poses/ranges NULL or n <= 0");
  const int B = c->B;
  std::vector<double> hx(n), hy(n), ht(n);
  for (int64_t i = 0; i < n; ++i) { hx[i] = poses[3 * i]; hy[i] = poses[3 * i + 1]; ht[i] = poses[3 * i + 2]; }
  double *dx = nullptr, *dy = nullptr, *dt = nullptr, *dr = nullptr;
  int32_t* dh = nullptr;
  int rc = MCLHIP_OK;
  hipError_t e = hipMalloc((void**)&dx, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&dy, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&dt, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&dr, (size_t)n * B * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&dh, (size_t)n * B * sizeof(int32_t));
  if (e == hipSuccess) e = hipMemcpyAsync(dx, hx.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dy, hy.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dt, ht.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) rc = launch_raycast(c, dx, dy, dt, nullptr, n, MCLHIP_F64, nullptr, nullptr, dr, dh);
  if (e == hipSuccess && rc == MCLHIP_OK) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess && rc == MCLHIP_OK) e = hipMemcpy(ranges, dr, (size_t)n * B * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == MCLHIP_OK && hit_cell)
    e = hipMemcpy(hit_cell, dh, (size_t)n * B * sizeof(int32_t), hipMemcpyDeviceToHost);
  (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dt); (void)hipFree(dr); (void)hipFree(dh);
  if (rc != MCLHIP_OK) return rc;
  if (e != hipSuccess) return fail(c, MCLHIP_E_HIP, std::string("raycast: ") + hipGetErrorString(e));
  return MCLHIP_OK;
}

int mclhip_raycast_static(mclhip_ctx* c, const double* poses, int64_t n, double* ranges, int32_t* hit_cell) {
  MCL_TRY(use(c));
  if (!c->obst) return fail(c, MCLHIP_E_STATE, "no map: call mclhip_set_map first");
  if (c->B == 0) return fail(c, MCLHIP_E_STATE, "no sensor: call mclhip_set_sensor first");
  if (c->ring_dirty)   // building them allocates; mclhip_set_map / mclhip_set_sensor do it once both are known
    return fail(c, MCLHIP_E_STATE, "map tables are not built: set the map and the sensor with mclhip_set_map / mclhip_set_sensor");
  if (!poses || !ranges || n <= 0) return fail(c, MCLHIP_E_ARG, "poses/ranges NULL or n <= 0");
  const int B = c->B;
  if (n > kStaticMaxPoses || n * (int64_t)B > kStaticMaxEntries)
    return fail(c, MCLHIP_E_LIMIT, "mclhip_raycast_static takes at most 4096 poses and 65536 (pose, beam) pairs per call; "
                                   "use mclhip_raycast for more");
  if (c->device < 0 || c->device >= kMaxDevices) return fail(c, MCLHIP_E_LIMIT, "device index beyond the static array's table");
  unsigned char* base = nullptr;
  hipError_t base_err = hipSuccess;
  {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    base = g_scratch_ptr[c->device];
    base_err = g_scratch_err[c->device];
  }
  if (!base)
    return fail(c, MCLHIP_E_HIP, std::string("address of the static device array is not known: ") + hipGetErrorString(base_err));
  std::lock_guard<std::mutex> call_lock(g_scratch_call_mutex[c->device]);   // until the results are back on the host
  double* const d_aos = reinterpret_cast<double*>(base + kStaticOffAos);
  double* const dx = reinterpret_cast<double*>(base + kStaticOffX);
  double* const dy = reinterpret_cast<double*>(base + kStaticOffY);
  double* const dt = reinterpret_cast<double*>(base + kStaticOffTh);
  double* const dr = reinterpret_cast<double*>(base + kStaticOffRanges);
  int32_t* const dh = reinterpret_cast<int32_t*>(base + kStaticOffHits);
  MCL_HIP(c, hipMemcpyAsync(d_aos, poses, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(split_poses_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                     (const double*)d_aos, n, dx, dy, dt);
  MCL_HIP(c, hipGetLastError());
  // the kernel writes every (pose, beam) entry of ranges and hit cells: nothing of the last call's contents is read
  MCL_TRY(launch_raycast(c, dx, dy, dt, nullptr, n, MCLHIP_F64, nullptr, nullptr, dr, dh));
  MCL_HIP(c, hipMemcpyAsync(ranges, dr, (size_t)n * B * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (hit_cell) MCL_HIP(c, hipMemcpyAsync(hit_cell, dh, (size_t)n * B * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  return MCLHIP_OK;
}

static int motion_stage(mclhip_ctx* c, const double* poses, int64_t n, double v, double omega, double dt,
                        const double alpha[6], const int16_t* noise_ints, double* poses_out, double* pm_out, bool book) {
  MCL_TRY(use(c));
  if (!book) MCL_TRY(need(c, true, false, false));
  if (!poses || !alpha || !poses_out || n <= 0) return fail(c, MCLHIP_E_ARG, "poses/alpha/poses_out NULL or n <= 0");
  std::vector<double> hx(n), hy(n), ht(n);
  for (int64_t i = 0; i < n; ++i) { hx[i] = poses[3 * i]; hy[i] = poses[3 * i + 1]; ht[i] = poses[3 * i + 2]; }
  double *dx = nullptr, *dy = nullptr, *dth = nullptr;
  uint8_t* dpm = nullptr;
  int16_t* dn = nullptr;
  int rc = MCLHIP_OK;
  hipError_t e = hipMalloc((void**)&dx, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&dy, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&dth, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&dpm, n);
  if (e == hipSuccess && noise_ints) e = hipMalloc((void**)&dn, (size_t)n * 36 * sizeof(int16_t));
  if (e == hipSuccess) e = hipMemcpyAsync(dx, hx.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dy, hy.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dth, ht.data(), n * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess && noise_ints)
    e = hipMemcpyAsync(dn, noise_ints, (size_t)n * 36 * sizeof(int16_t), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    MotionParams mp = make_motion_params(c, v, omega, dt, alpha, noise_ints ? 1 : 0, 0);
    mp.book = book ? 1 : 0;
    rc = launch_motion(c, dx, dy, dth, dpm, n, mp, dn, MCLHIP_F64);
  }
  std::vector<uint8_t> hpm(n);
  if (e == hipSuccess && rc == MCLHIP_OK) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess && rc == MCLHIP_OK) e = hipMemcpy(hx.data(), dx, n * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == MCLHIP_OK) e = hipMemcpy(hy.data(), dy, n * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == MCLHIP_OK) e = hipMemcpy(ht.data(), dth, n * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && rc == MCLHIP_OK) e = hipMemcpy(hpm.data(), dpm, n, hipMemcpyDeviceToHost);
  (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dth); (void)hipFree(dpm); (void)hipFree(dn);
  if (rc != MCLHIP_OK) return rc;
  if (e != hipSuccess) return fail(c, MCLHIP_E_HIP, std::string("motion: ") + hipGetErrorString(e));
  for (int64_t i = 0; i < n; ++i) {
    poses_out[3 * i] = hx[i]; poses_out[3 * i + 1] = hy[i]; poses_out[3 * i + 2] = ht[i];
    if (pm_out) pm_out[i] = hpm[i] ? 1.0 : 0.0;
  }
  return MCLHIP_OK;
}

int mclhip_motion(mclhip_ctx* c, const double* poses, int64_t n, double v, double omega, double dt,
                  const double alpha[6], const int16_t* noise_ints, double* poses_out, double* pm_out) {
  return motion_stage(c, poses, n, v, omega, dt, alpha, noise_ints, poses_out, pm_out, false);
}

int mclhip_motion_book(mclhip_ctx* c, const double* poses, int64_t n, double v, double omega, double dt,
                       const double alpha[6], const int16_t* noise_ints, double* poses_out) {
  return motion_stage(c, poses, n, v, omega, dt, alpha, noise_ints, poses_out, nullptr, true);
}

int mclhip_compute_px(mclhip_ctx* c, const double* z, const double* z_true, int64_t n, double* p_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, true, false));
  if (!z || !z_true || !p_out || n <= 0) return fail(c, MCLHIP_E_ARG, "z/z_true/p_out NULL or n <= 0");
  double *dz = nullptr, *dt = nullptr, *dp = nullptr;
  int rc = MCLHIP_OK;
  hipError_t e = hipMalloc((void**)&dz, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&dt, n * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void**)&dp, n * sizeof(double));
  if (e == hipSuccess) e = hipMemcpyAsync(dz, z, n * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dt, z_true, n * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) rc = launch_compute_px(c, dz, dt, n, dp);
  if (e == hipSuccess && rc == MCLHIP_OK) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess && rc == MCLHIP_OK) e = hipMemcpy(p_out, dp, n * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(dz); (void)hipFree(dt); (void)hipFree(dp);
  if (rc != MCLHIP_OK) return rc;
  if (e != hipSuccess) return fail(c, MCLHIP_E_HIP, std::string("compute_px: ") + hipGetErrorString(e));
  return MCLHIP_OK;
}

int mclhip_sample(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6], const int16_t* noise_ints,
                  uint64_t seed, const double* z_meas) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, true, true, true));
  if (!alpha) return fail(c, MCLHIP_E_ARG, "alpha is NULL");
  MCL_TRY(upload_zmeas(c, z_meas));
  if (noise_ints) MCL_TRY(upload_noise(c, noise_ints));
  std::vector<mclhip_ctx::Span>& spans = c->spans;
  MCL_TRY(enqueue_sample(c, v, omega, dt, alpha, noise_ints ? 1 : 2, seed, spans, true));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, spans);
  return MCLHIP_OK;
}

int mclhip_sample_augmented(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6],
                            const int16_t* noise_ints, uint64_t seed, const double* z_meas,
                            const double* range_noise, double* w_avg_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, true, true, true));
  if (!alpha) return fail(c, MCLHIP_E_ARG, "alpha is NULL");
  MCL_TRY(upload_zmeas(c, z_meas));
  if (noise_ints) MCL_TRY(upload_noise(c, noise_ints));
  if (range_noise) {
    const size_t bytes = (size_t)c->M * c->B * sizeof(double);
    MCL_TRY(dev_alloc(c, &c->range_noise, bytes));
    MCL_HIP(c, hipMemcpyAsync(c->range_noise, range_noise, bytes, hipMemcpyHostToDevice, c->stream));
  }
  MCL_TRY(enqueue_sample(c, v, omega, dt, alpha, noise_ints ? 1 : 2, seed, c->spans, true,
                         range_noise ? c->range_noise : nullptr));
  MCL_TRY(launch_mean_weight(c));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, c->spans);
  if (w_avg_out) MCL_HIP(c, hipMemcpy(w_avg_out, c->scal + 2, sizeof(double), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

int mclhip_resample_augmented(mclhip_ctx* c, double r, const int64_t* inj_rows, const double* inj_poses,
                              int64_t n_inj, int64_t* idx_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (n_inj < 0 || n_inj > c->M || (n_inj && (!inj_rows || !inj_poses)))
    return fail(c, MCLHIP_E_ARG, "bad injection list");
  for (int64_t j = 0; j < n_inj; ++j)
    if (inj_rows[j] < 0 || inj_rows[j] >= c->M) return fail(c, MCLHIP_E_ARG, "injected row outside [0, M)");
  if (n_inj > c->inj_cap) {
    MCL_TRY(dev_alloc(c, &c->inj_rows, (size_t)n_inj * sizeof(int64_t)));
    MCL_TRY(dev_alloc(c, &c->inj_poses, (size_t)n_inj * 3 * sizeof(double)));
    c->inj_cap = n_inj;
  }
  if (n_inj) {
    MCL_HIP(c, hipMemcpyAsync(c->inj_rows, inj_rows, n_inj * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    MCL_HIP(c, hipMemcpyAsync(c->inj_poses, inj_poses, n_inj * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  // self.X[m] = (x, y, a) in place, before the walk reads X (mc_localization2.py:284-286)
  MCL_TRY(launch_inject(c, c->inj_rows, c->inj_poses, n_inj, c->cur, false));
  MCL_TRY(launch_resample(c, r, MCLHIP_RESAMPLE_PARALLEL));
  // the injected positions themselves: pose and weight 1/M (:287-288)
  MCL_TRY(launch_inject(c, c->inj_rows, c->inj_poses, n_inj, c->cur, true));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  if (idx_out) MCL_HIP(c, hipMemcpy(idx_out, c->idx, c->M * sizeof(int64_t), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

int mclhip_sample_augmented_device(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6], uint64_t seed,
                                   const double* z_meas, double range_sigma, double* w_avg_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, true, true, true));
  if (!alpha) return fail(c, MCLHIP_E_ARG, "alpha is NULL");
  if (!(range_sigma > 0.0)) return fail(c, MCLHIP_E_ARG, "range_sigma must be positive");
  MCL_TRY(upload_zmeas(c, z_meas));
  // integers of the motion sampler uniform on [-1000, 1000): np.random.randint(-1000, 1000, 12) (mc_localization2.py:23)
  // float32 state: log-weights from the float32 mixture (the interval-only sector query with the noise drawn in its
  // mixture, three times as fast as the general kernel), the raw weights of mc_localization2.py:235-240 as their
  // exponentials -- the noise is the device's own, so what is kept is the distribution, not a sequence of roundings.
  // float64 state: the product itself, factor by factor.
  const bool from_log = c->dtype == MCLHIP_F32;
  MCL_TRY(enqueue_sample(c, v, omega, dt, alpha, 2, seed, c->spans, !from_log, nullptr, range_sigma, 2000));
  if (from_log) MCL_TRY(launch_exp_weights(c));
  MCL_TRY(launch_mean_weight(c));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, c->spans);
  if (w_avg_out) MCL_HIP(c, hipMemcpy(w_avg_out, c->scal + 2, sizeof(double), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

int mclhip_resample_augmented_device(mclhip_ctx* c, double r, double p_random, double extent_x, double extent_y,
                                     uint64_t seed, int64_t* n_injected_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, true, false, true));
  if (!(extent_x > 0) || !(extent_y > 0)) return fail(c, MCLHIP_E_ARG, "empty extent");
  if (!c->counters) MCL_TRY(dev_alloc(c, &c->counters, 16 * sizeof(uint32_t)));
  MCL_HIP(c, hipMemsetAsync(c->counters, 0, 2 * sizeof(uint32_t), c->stream));
  const bool any = p_random > 0.0;        // also false for NaN, as `rand() <= nan` is
  const double al[6] = {0, 0, 0, 0, 0, 0};
  MotionParams mp = make_motion_params(c, 0.0, 0.0, 0.0, al, 0, seed);
  // self.X[m] = (x, y, a) in place, before the walk reads X (mc_localization2.py:284-286)
  if (any) MCL_TRY(launch_inject_device(c, c->cur, mp, p_random, extent_x, extent_y, seed, false, c->counters));
  MCL_TRY(launch_resample(c, r, MCLHIP_RESAMPLE_PARALLEL));
  // the injected positions themselves: pose and weight 1/M (:287-288)
  if (any) MCL_TRY(launch_inject_device(c, c->cur, mp, p_random, extent_x, extent_y, seed, true, c->counters));
  uint32_t cnt[2] = {0, 0};
  MCL_HIP(c, hipMemcpyAsync(cnt, c->counters, sizeof(cnt), hipMemcpyDeviceToHost, c->stream));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  if (n_injected_out) *n_injected_out = (int64_t)cnt[0];
  if (cnt[1]) return fail(c, MCLHIP_E_STATE, "no free cell found for an injected pose (is any cell of the map 255 inside the extent?)");
  return MCLHIP_OK;
}

int mclhip_normalise(mclhip_ctx* c, int mode, double* sum_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (mode != MCLHIP_W_PRODUCT && mode != MCLHIP_W_LOG) return fail(c, MCLHIP_E_ARG, "bad weight mode");
  std::vector<mclhip_ctx::Span>& spans = c->spans;
  span_begin(c, spans, 2);
  MCL_TRY(launch_normalise(c, mode));
  span_end(c, spans);
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, spans);
  if (sum_out) MCL_HIP(c, hipMemcpy(sum_out, c->scal, sizeof(double), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

int mclhip_resample(mclhip_ctx* c, double r, int algo, int64_t* idx_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (algo != MCLHIP_RESAMPLE_SERIAL && algo != MCLHIP_RESAMPLE_PARALLEL) return fail(c, MCLHIP_E_ARG, "bad algo");
  std::vector<mclhip_ctx::Span>& spans = c->spans;
  span_begin(c, spans, 2);
  MCL_TRY(launch_resample(c, r, algo));
  span_end(c, spans);
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, spans);
  if (idx_out) MCL_HIP(c, hipMemcpy(idx_out, c->idx, c->M * sizeof(int64_t), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

int mclhip_normalise_resample(mclhip_ctx* c, int mode, double r, double* sum_out, int64_t* idx_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (mode != MCLHIP_W_PRODUCT && mode != MCLHIP_W_LOG) return fail(c, MCLHIP_E_ARG, "bad weight mode");
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_normalise_resample(c, mode, r));
  span_end(c, c->spans);
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, c->spans);
  if (sum_out) MCL_HIP(c, hipMemcpy(sum_out, c->scal, sizeof(double), hipMemcpyDeviceToHost));
  if (idx_out) MCL_HIP(c, hipMemcpy(idx_out, c->idx, c->M * sizeof(int64_t), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

static int step_common(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6],
                       const int16_t* noise_ints, uint64_t seed, const double* z_meas, int weight_mode, double r,
                       bool wait) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, true, true, true));
  if (!alpha) return fail(c, MCLHIP_E_ARG, "alpha is NULL");
  if (weight_mode != MCLHIP_W_PRODUCT && weight_mode != MCLHIP_W_LOG) return fail(c, MCLHIP_E_ARG, "bad weight mode");
  MCL_TRY(upload_zmeas(c, z_meas));
  if (noise_ints) MCL_TRY(upload_noise(c, noise_ints));
  std::vector<mclhip_ctx::Span>& spans = c->spans;
  MCL_TRY(enqueue_sample(c, v, omega, dt, alpha, noise_ints ? 1 : 2, seed, spans, weight_mode == MCLHIP_W_PRODUCT));
  span_begin(c, spans, 2);
  MCL_TRY(launch_normalise_resample(c, weight_mode, r));
  span_end(c, spans);
  if (wait) {
    MCL_HIP(c, hipStreamSynchronize(c->stream));
    span_collect(c, spans);
  }
  return MCLHIP_OK;
}

int mclhip_step(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6], const int16_t* noise_ints,
                uint64_t seed, const double* z_meas, int weight_mode, double r) {
  return step_common(c, v, omega, dt, alpha, noise_ints, seed, z_meas, weight_mode, r, true);
}

int mclhip_step_async(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6], uint64_t seed,
                      const double* z_meas, int weight_mode, double r) {
  return step_common(c, v, omega, dt, alpha, nullptr, seed, z_meas, weight_mode, r, false);
}

int mclhip_sync(mclhip_ctx* c) {
  MCL_TRY(use(c));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_snapshot_particles(mclhip_ctx* c) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  const size_t es = esize(c);
  const int64_t M = c->M;
  MCL_TRY(dev_alloc(c, &c->sx, M * es));
  MCL_TRY(dev_alloc(c, &c->sy, M * es));
  MCL_TRY(dev_alloc(c, &c->sth, M * es));
  MCL_TRY(dev_alloc(c, &c->sw, M * sizeof(double)));
  const int a = c->cur;
  MCL_HIP(c, hipMemcpyAsync(c->sx, c->px[a], M * es, hipMemcpyDeviceToDevice, c->stream));
  MCL_HIP(c, hipMemcpyAsync(c->sy, c->py[a], M * es, hipMemcpyDeviceToDevice, c->stream));
  MCL_HIP(c, hipMemcpyAsync(c->sth, c->pth[a], M * es, hipMemcpyDeviceToDevice, c->stream));
  MCL_HIP(c, hipMemcpyAsync(c->sw, c->w[a], M * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  c->sM = M;
  return MCLHIP_OK;
}

int mclhip_restore_particles_async(mclhip_ctx* c) {
  MCL_TRY(use(c));
  if (c->sM == 0 || c->sM != c->M) return fail(c, MCLHIP_E_STATE, "no snapshot of the current particle count");
  const int64_t M = c->M;
  const int a = c->cur;
  return launch_restore(c, c->px[a], c->py[a], c->pth[a], c->w[a], c->sx, c->sy, c->sth, c->sw, M, c->dtype);
}

// ---- one filter over several GPUs ----------------------------------------------------------
int mclhip_shard_set(mclhip_ctx* c, int64_t first_index, int64_t m_total) {
  if (!c) return MCLHIP_E_ARG;
  if (first_index < 0 || m_total <= 0 || first_index >= m_total) return fail(c, MCLHIP_E_ARG, "bad shard range");
  c->first_index = first_index;
  c->m_total = m_total;
  return MCLHIP_OK;
}

int mclhip_shard_sample(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6],
                        const int16_t* noise_ints, uint64_t seed, const double* z_meas, int weight_mode) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, true, true, true));
  if (!alpha) return fail(c, MCLHIP_E_ARG, "alpha is NULL");
  MCL_TRY(upload_zmeas(c, z_meas));
  if (noise_ints) MCL_TRY(upload_noise(c, noise_ints));
  MCL_TRY(enqueue_sample(c, v, omega, dt, alpha, noise_ints ? 1 : 2, seed, c->spans, weight_mode == MCLHIP_W_PRODUCT));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_shard_max_logw(mclhip_ctx* c, double* max_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!max_out) return fail(c, MCLHIP_E_ARG, "max_out is NULL");
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_shard_max_logw(c));
  span_end(c, c->spans);
  MCL_HIP(c, hipMemcpyAsync(max_out, c->scal + 1, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_shard_weights(mclhip_ctx* c, int mode, double gmax, double* chunk_sums_out, int64_t n_chunks) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  const int64_t nch = (c->M + kSumChunk - 1) / kSumChunk;
  if (!chunk_sums_out || n_chunks != nch) return fail(c, MCLHIP_E_ARG, "chunk_sums_out NULL or wrong chunk count");
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_shard_weights(c, mode, gmax));
  span_end(c, c->spans);
  MCL_HIP(c, hipMemcpyAsync(chunk_sums_out, c->chunk_sums, nch * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_shard_scan(mclhip_ctx* c, double total, double approx_offset, uint64_t* k_total, uint32_t* n_records) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!k_total || !n_records) return fail(c, MCLHIP_E_ARG, "output pointer is NULL");
  if (c->m_total <= 0) return fail(c, MCLHIP_E_STATE, "call mclhip_shard_set first");
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_shard_scan(c, total, approx_offset, c->first_index));
  span_end(c, c->spans);
  MCL_HIP(c, hipMemcpyAsync(k_total, c->kscan + (c->M - 1), sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  MCL_HIP(c, hipMemcpyAsync(n_records, c->fscan + (c->M - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_shard_get_records(mclhip_ctx* c, void* records_out, int64_t cap_records) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  uint32_t nrec = 0;
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  MCL_HIP(c, hipMemcpy(&nrec, c->fscan + (c->M - 1), sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (cap_records < (int64_t)nrec || (!records_out && nrec)) return fail(c, MCLHIP_E_ARG, "records_out too small");
  if (nrec) MCL_HIP(c, hipMemcpy(records_out, c->irr, (size_t)nrec * irr_record_bytes(), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

int mclhip_shard_finalize(mclhip_ctx* c, const void* all_records, int64_t n_all, uint64_t P_off, uint32_t F_off) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!all_records || n_all <= 0) return fail(c, MCLHIP_E_ARG, "the global record list is empty (element 0 is always a record)");
  if (n_all > c->g_records_cap) {
    const int64_t cap = n_all + n_all / 2 + 1024;
    MCL_TRY(dev_alloc(c, &c->g_records, (size_t)cap * irr_record_bytes()));
    MCL_TRY(dev_alloc(c, &c->g_bases, (size_t)cap * run_base_bytes()));
    c->g_records_cap = cap;
  }
  MCL_HIP(c, hipMemcpyAsync(c->g_records, all_records, (size_t)n_all * irr_record_bytes(), hipMemcpyHostToDevice, c->stream));
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_shard_finalize(c, c->g_records, n_all, c->g_bases, P_off, F_off));
  span_end(c, c->spans);
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_shard_count_below(mclhip_ctx* c, const double* U, int n, int64_t* counts_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!U || !counts_out || n <= 0 || n > 1024) return fail(c, MCLHIP_E_ARG, "U/counts NULL or n outside 1..1024");
  if (!c->small_d) MCL_TRY(dev_alloc(c, &c->small_d, 1024 * sizeof(double)));
  if (!c->small_i) MCL_TRY(dev_alloc(c, &c->small_i, 1024 * sizeof(int64_t)));
  MCL_HIP(c, hipMemcpyAsync(c->small_d, U, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  MCL_TRY(launch_count_below(c, c->small_d, n, c->small_i));
  MCL_HIP(c, hipMemcpyAsync(counts_out, c->small_i, n * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  return MCLHIP_OK;
}

// ---- device-driven cross-rank path: every call below only enqueues on the context's stream ----

int mclhip_set_stream(mclhip_ctx* c, void* stream) {
  MCL_TRY(use(c));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  c->stream = stream ? (hipStream_t)stream : c->own_stream;
  return MCLHIP_OK;
}

static int make_layout(mclhip_ctx* c, ShardLayout& lay, const int64_t* firsts, int world, int rank) {
  if (!firsts || world < 1 || world > 64 || rank < 0 || rank >= world) return fail(c, MCLHIP_E_ARG, "firsts NULL or world/rank outside 1..64");
  lay.world = world; lay.rank = rank; lay.cmax = 0;
  for (int r = 0; r < world; ++r) {
    if (firsts[r + 1] < firsts[r]) return fail(c, MCLHIP_E_ARG, "firsts must ascend");
    lay.first[r] = firsts[r];
    lay.nchunks[r] = (int)((firsts[r + 1] - firsts[r] + kSumChunk - 1) / kSumChunk);
    lay.cmax = std::max(lay.cmax, lay.nchunks[r]);
  }
  lay.first[world] = firsts[world];
  if (firsts[rank + 1] - firsts[rank] != c->M || firsts[rank] != c->first_index || firsts[world] != c->m_total)
    return fail(c, MCLHIP_E_STATE, "firsts disagree with mclhip_shard_set / the local particle count");
  return MCLHIP_OK;
}

int mclhip_dshard_sample_max(mclhip_ctx* c, double v, double omega, double dt, const double alpha[6],
                             const int16_t* noise_ints, uint64_t seed, const double* z_meas, int weight_mode,
                             double* d_max_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, true, true, true));
  if (!alpha || !d_max_out) return fail(c, MCLHIP_E_ARG, "alpha or d_max_out is NULL");
  MCL_TRY(upload_zmeas(c, z_meas));
  if (noise_ints) {
    // pre-drawn integers (parity runs) are read from the caller's buffer: this call returns only when that is done
    MCL_TRY(upload_noise(c, noise_ints));
    MCL_HIP(c, hipStreamSynchronize(c->stream));
  }
  MCL_TRY(enqueue_sample(c, v, omega, dt, alpha, noise_ints ? 1 : 2, seed, c->spans, weight_mode == MCLHIP_W_PRODUCT));
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_shard_max_logw(c, d_max_out));   // (written where the caller reduces it: no copy in between)
  span_end(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_dshard_weights(mclhip_ctx* c, const double* d_gmax, double* d_chunks_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!d_gmax || !d_chunks_out) return fail(c, MCLHIP_E_ARG, "device pointer is NULL");
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_dshard_weights(c, d_gmax, d_chunks_out));   // (reads the maximum, leaves the chunk sums, where the caller keeps them)
  span_end(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_dshard_scan(mclhip_ctx* c, const double* d_all_chunks, const int64_t* firsts, int world, int rank,
                       void* d_records_out, int cap) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!d_all_chunks || !d_records_out || cap < 1) return fail(c, MCLHIP_E_ARG, "device pointer NULL or cap < 1");
  if (c->m_total <= 0) return fail(c, MCLHIP_E_STATE, "call mclhip_shard_set first");
  ShardLayout lay;
  MCL_TRY(make_layout(c, lay, firsts, world, rank));
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_dshard_scan(c, d_all_chunks, lay, d_records_out, cap));
  span_end(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_dshard_finalize(mclhip_ctx* c, const void* d_all_records, const int64_t* firsts, int world, int rank,
                           int cap, double r, int64_t* d_counts_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!d_all_records || !d_counts_out || cap < 1) return fail(c, MCLHIP_E_ARG, "device pointer NULL or cap < 1");
  ShardLayout lay;
  MCL_TRY(make_layout(c, lay, firsts, world, rank));
  const int64_t n_all = (int64_t)world * cap;
  if (n_all > c->g_records_cap) {
    MCL_HIP(c, hipStreamSynchronize(c->stream));
    MCL_TRY(dev_alloc(c, &c->g_records, (size_t)n_all * irr_record_bytes()));
    MCL_TRY(dev_alloc(c, &c->g_bases, (size_t)n_all * run_base_bytes()));
    c->g_records_cap = n_all;
  }
  if (!c->small_i) MCL_TRY(dev_alloc(c, &c->small_i, 1024 * sizeof(int64_t)));
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_dshard_finalize(c, d_all_records, lay, cap, r, d_counts_out));
  span_end(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_fs_dshard_scan(mclhip_ctx* c, const double* d_all_chunks, const double* d_all_lasts, const int64_t* firsts, int world,
                          int rank, void* d_records_out, int cap) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!d_all_chunks || !d_records_out || cap < 1) return fail(c, MCLHIP_E_ARG, "device pointer NULL or cap < 1");
  if (c->m_total <= 0) return fail(c, MCLHIP_E_STATE, "call mclhip_shard_set first");
  ShardLayout lay;
  MCL_TRY(make_layout(c, lay, firsts, world, rank));
  MCL_TRY(launch_fs_dshard_scan(c, d_all_chunks, d_all_lasts, lay, d_records_out, cap));
  return MCLHIP_OK;
}

int mclhip_fs_dshard_finalize_raw(mclhip_ctx* c, const void* d_all_records, const int64_t* firsts, int world, int rank, int cap,
                                  double* d_last_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!d_all_records || !d_last_out || cap < 1) return fail(c, MCLHIP_E_ARG, "device pointer NULL or cap < 1");
  ShardLayout lay;
  MCL_TRY(make_layout(c, lay, firsts, world, rank));
  const int64_t n_all = (int64_t)world * cap;
  if (n_all > c->g_records_cap) {
    MCL_HIP(c, hipStreamSynchronize(c->stream));
    MCL_TRY(dev_alloc(c, &c->g_records, (size_t)n_all * irr_record_bytes()));
    MCL_TRY(dev_alloc(c, &c->g_bases, (size_t)n_all * run_base_bytes()));
    c->g_records_cap = n_all;
  }
  if (!c->small_i) MCL_TRY(dev_alloc(c, &c->small_i, 1024 * sizeof(int64_t)));
  MCL_TRY(launch_fs_dshard_finalize_raw(c, d_all_records, lay, cap, d_last_out));
  return MCLHIP_OK;
}

int mclhip_dshard_pack(mclhip_ctx* c, int world, void* d_rows) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!d_rows || world < 1 || world > 64) return fail(c, MCLHIP_E_ARG, "d_rows NULL or world outside 1..64");
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_pack_drawn(c, world, (double*)d_rows, nullptr));
  span_end(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_dshard_unpack_resample(mclhip_ctx* c, const void* d_rows, int64_t L, double r) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!d_rows || L <= 0) return fail(c, MCLHIP_E_ARG, "d_rows NULL or empty");
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_unpack_drawn(c, (const double*)d_rows, L, r, false));
  span_end(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_dshard_pack_fixed(mclhip_ctx* c, int world, int rank, int64_t cap_rows, void* d_blocks,
                             const int64_t* d_all_counts, int64_t* d_flag_out, int64_t* h_late_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!d_blocks || !d_all_counts || !d_flag_out || world < 1 || world > 64 || rank < 0 || rank >= world || cap_rows < 1)
    return fail(c, MCLHIP_E_ARG, "NULL argument, world outside 1..64, rank outside the world or cap_rows < 1");
  if (c->M + 64 > c->rows_self_cap) {
    MCL_HIP(c, hipStreamSynchronize(c->stream));
    MCL_TRY(dev_alloc(c, &c->rows_self, (size_t)(c->M + 64) * 6 * sizeof(double)));
    c->rows_self_cap = c->M + 64;
  }
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_pack_fixed(c, world, rank, cap_rows, (double*)d_blocks, (double*)c->rows_self, d_all_counts, d_flag_out, h_late_out));
  span_end(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_dshard_unpack_fixed(mclhip_ctx* c, const void* d_blocks, const int64_t* d_all_counts, int world, int rank,
                               int64_t cap_rows, double r, int64_t* d_flag_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!d_blocks || !d_all_counts || !d_flag_out || world < 1 || world > 64 || rank < 0 || rank >= world || cap_rows < 1)
    return fail(c, MCLHIP_E_ARG, "NULL argument, world outside 1..64, rank outside the world or cap_rows < 1");
  if (!c->rows_self) return fail(c, MCLHIP_E_STATE, "mclhip_dshard_pack_fixed comes first");
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_unpack_fixed(c, (const double*)d_blocks, (const double*)c->rows_self, d_all_counts, world, rank, cap_rows, r,
                              d_flag_out));
  span_end(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_dshard_unpack_undo(mclhip_ctx* c) {
  MCL_TRY(use(c));
  c->cur = 1 - c->cur;
  return MCLHIP_OK;
}

int mclhip_get_indices(mclhip_ctx* c, int64_t* idx_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!idx_out) return fail(c, MCLHIP_E_ARG, "idx_out is NULL");
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  MCL_HIP(c, hipMemcpy(idx_out, c->idx, c->M * sizeof(int64_t), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

int mclhip_shard_pack(mclhip_ctx* c, const int64_t* lo, const int64_t* hi, int nslices, void* rows) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!lo || !hi || nslices <= 0) return fail(c, MCLHIP_E_ARG, "lo/hi NULL or no slices");
  int64_t total = 0;
  for (int s = 0; s < nslices; ++s) {
    if (lo[s] < 0 || hi[s] > c->M || hi[s] < lo[s]) return fail(c, MCLHIP_E_ARG, "slice outside the local range");
    total += hi[s] - lo[s];
  }
  if (total == 0) return MCLHIP_OK;
  if (!rows) return fail(c, MCLHIP_E_ARG, "rows is NULL");
  if (total > c->rows_send_cap) {
    MCL_TRY(dev_alloc(c, &c->rows_send, (size_t)(total + total / 4) * 6 * sizeof(double)));   // capacity in rows of up to six doubles
    c->rows_send_cap = total + total / 4;
  }
  span_begin(c, c->spans, 2);
  int64_t off = 0;
  for (int s = 0; s < nslices; ++s) {
    MCL_TRY(launch_pack_rows(c, lo[s], hi[s], (double*)c->rows_send + off * 5));
    off += hi[s] - lo[s];
  }
  span_end(c, c->spans);
  MCL_HIP(c, hipMemcpyAsync(rows, c->rows_send, (size_t)total * 5 * sizeof(double), hipMemcpyDefault, c->stream));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, c->spans);
  return MCLHIP_OK;
}

int mclhip_shard_unpack_resample(mclhip_ctx* c, const void* rows, int64_t L, double r, int64_t row0_global,
                                 int64_t* idx_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!rows || L <= 0) return fail(c, MCLHIP_E_ARG, "rows NULL or empty");
  if (c->m_total <= 0) return fail(c, MCLHIP_E_STATE, "call mclhip_shard_set first");
  if (L > c->rows_recv_cap) {
    MCL_TRY(dev_alloc(c, &c->rows_recv, (size_t)(L + L / 4) * 6 * sizeof(double)));
    c->rows_recv_cap = L + L / 4;
  }
  MCL_HIP(c, hipMemcpyAsync(c->rows_recv, rows, (size_t)L * 5 * sizeof(double), hipMemcpyDefault, c->stream));
  span_begin(c, c->spans, 2);
  MCL_TRY(launch_unpack_resample(c, (const double*)c->rows_recv, L, c->first_index, c->m_total, r, row0_global));
  span_end(c, c->spans);
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  span_collect(c, c->spans);
  if (idx_out) MCL_HIP(c, hipMemcpy(idx_out, c->idx, c->M * sizeof(int64_t), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

// ---- the drawn-row exchange for the host-combined path (rows may be host or device memory) ----

int mclhip_shard_plan_drawn(mclhip_ctx* c, const int64_t* firsts, int world, int rank, double r, int64_t* counts_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!counts_out) return fail(c, MCLHIP_E_ARG, "counts_out is NULL");
  if (c->m_total <= 0) return fail(c, MCLHIP_E_STATE, "call mclhip_shard_set first");
  ShardLayout lay;
  MCL_TRY(make_layout(c, lay, firsts, world, rank));
  if (!c->small_i) MCL_TRY(dev_alloc(c, &c->small_i, 1024 * sizeof(int64_t)));
  int64_t* d_counts = c->small_i + 8 + 4 * 64;
  MCL_TRY(launch_plan_drawn(c, lay, r, d_counts, nullptr));
  MCL_HIP(c, hipMemcpyAsync(counts_out, d_counts, world * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  c->drawn_total = 0;
  for (int d = 0; d < world; ++d) c->drawn_total += counts_out[d];
  return MCLHIP_OK;
}

int mclhip_shard_pack_drawn(mclhip_ctx* c, int world, void* rows) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (world < 1 || world > 64) return fail(c, MCLHIP_E_ARG, "world outside 1..64");
  const int64_t total = c->drawn_total;
  if (total == 0) return MCLHIP_OK;
  if (!rows) return fail(c, MCLHIP_E_ARG, "rows is NULL");
  if (total > c->rows_send_cap) {
    MCL_TRY(dev_alloc(c, &c->rows_send, (size_t)(total + total / 4) * 6 * sizeof(double)));
    c->rows_send_cap = total + total / 4;
  }
  if (total > c->sel_cap) {
    MCL_TRY(dev_alloc(c, &c->sel, (size_t)(total + total / 4) * sizeof(int64_t)));
    c->sel_cap = total + total / 4;
  }
  MCL_TRY(launch_pack_drawn(c, world, (double*)c->rows_send, c->sel));
  MCL_HIP(c, hipMemcpyAsync(rows, c->rows_send, (size_t)total * 6 * sizeof(double), hipMemcpyDefault, c->stream));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  return MCLHIP_OK;
}

int mclhip_shard_unpack_drawn(mclhip_ctx* c, const void* rows, int64_t L, double r, int64_t* idx_out) {
  MCL_TRY(use(c));
  MCL_TRY(need(c, false, false, true));
  if (!rows || L <= 0) return fail(c, MCLHIP_E_ARG, "rows NULL or empty");
  if (c->m_total <= 0) return fail(c, MCLHIP_E_STATE, "call mclhip_shard_set first");
  if (L > c->rows_recv_cap) {
    MCL_TRY(dev_alloc(c, &c->rows_recv, (size_t)(L + L / 4) * 6 * sizeof(double)));
    c->rows_recv_cap = L + L / 4;
  }
  MCL_HIP(c, hipMemcpyAsync(c->rows_recv, rows, (size_t)L * 6 * sizeof(double), hipMemcpyDefault, c->stream));
  MCL_TRY(launch_unpack_drawn(c, (const double*)c->rows_recv, L, r, true));
  MCL_HIP(c, hipStreamSynchronize(c->stream));
  if (idx_out) MCL_HIP(c, hipMemcpy(idx_out, c->idx, c->M * sizeof(int64_t), hipMemcpyDeviceToHost));
  return MCLHIP_OK;
}

const char* mclhip_query_kernel_name(mclhip_ctx* c) { return c ? c->query_kernel.c_str() : ""; }

int mclhip_timing_enable(mclhip_ctx* c, int on) {
  if (!c) return MCLHIP_E_ARG;
  c->timing = on == 2 ? 2 : (on != 0 ? 1 : 0);
  for (int i = 0; i < 6; ++i) { c->t_launch[i] = 0; c->t_ms[i] = 0; }
  return MCLHIP_OK;
}

int mclhip_timing_get(mclhip_ctx* c, int which, int64_t* launches, double* total_ms) {
  if (c && which >= 100 && which < 116 && c->stamps) {   // profiling build: cycles of kernel section which-100
    unsigned long long v = 0;
    (void)hipMemcpy(&v, c->stamps + (which - 100), sizeof(v), hipMemcpyDeviceToHost);
    if (launches) *launches = (int64_t)v;
    if (total_ms) *total_ms = (double)v;
    return MCLHIP_OK;
  }
  if (c && (which == 6 || which == 7)) {
    // particles the last two-launch sector query left over, as the host last saw it: 6 -- to the entry-by-entry launch;
    // 7 -- by the first launch (the same, unless the arithmetic interval kernel made its second pass in between)
    uint64_t left = 0;
    const int from = (which == 6 && c->every_last) ? 64 : 0;
    if (c->redo_hint)
      for (int k = 0; k < 64; ++k) left += reinterpret_cast<volatile uint32_t*>(c->redo_hint)[from + k];
    if (launches) *launches = (int64_t)left;
    if (total_ms) *total_ms = 0.0;
    return MCLHIP_OK;
  }
  if (!c || which < 0 || which > 5) return MCLHIP_E_ARG;
  if (launches) *launches = c->t_launch[which];
  if (total_ms) *total_ms = c->t_ms[which];
  return MCLHIP_OK;
}

}  // extern "C"
