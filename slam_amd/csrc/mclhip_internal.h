// Internal declarations shared by the translation units of libmclhip.so.
// gfx950 (MI355X) only: 64-lane wavefronts are assumed throughout.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <string>
#include <vector>

#include "../../include/mclhip.h"

namespace mcl {

constexpr int kWave = 64;
constexpr int kMaxBeams = 1024;
constexpr double kEdgeEps = 1e-5;  // the float32 bearing of the sector query is trusted farther than this from a sector edge
constexpr int kSumChunk = 8192;   // numpy reduces float64 vectors chunk by chunk of this size
constexpr int kScanBlock = 256;   // threads per scan block
constexpr int kScanItems = 8;     // elements per thread in the scans
constexpr int kScanTile = kScanBlock * kScanItems;

// Neighbourhood lists are kept per QUERY tile of (8 >> kQueryShift)^2 cells: the smaller the query tile, the
// fewer cells of its list lie outside a particle's disk (and the more lists there are).  With every cell of a list
// tested against the query tile (reach z_max plus a cell) and a pose walking only the classes of the list it can
// reach (kNeedClasses), an 8x8 query tile on a map with 4 % of its cells occupied has a pose walk 3.08 batches of 64
// for the 132 cells within z_max, a 4x4 query tile fewer: measured on one GPU, 2^20 x 180, the interval-only launch
// 0.577 -> 0.562 ms (global regime) and 0.543 -> 0.529 ms (tracking) at three and a half times the list memory (0.7 GB
// for a 4096^2 map, 14 GB for 16384^2) -- memory this part has to spare; hence 4x4.
#ifndef MCL_QUERY_SHIFT
#define MCL_QUERY_SHIFT 1
#endif
constexpr int kQueryShift = MCL_QUERY_SHIFT;        // 0: 8x8 cells, 1: 4x4, 2: 2x2
constexpr int kQueryCells = 8 >> kQueryShift;
constexpr int kNeedClasses = 8;   // a query tile's list is kept in this many classes by distance from the tile's centre (map_motion.hip: cell_class)

// Everything the sector-query kernel needs about the map and the sensor.
struct RayParams {
  const uint64_t* nbr_off;   // [ETH*ETW + 1] start of each query tile's neighbourhood list in nbr_cells[] (64-bit: a
                             // 16384^2 map with a fifth of its cells occupied lists 4.6e9 entries)
  const uint32_t* nbr_cells; // per tile: the obstacle cells (row << 16 | col) of every tile that can hold a cell
                             // within z_max of a point of that tile, nearest tile first
  int ETW, ETH, rtx, rty;    // the query-tile grid, extended by rtx / rty query tiles on each side (poses just outside the map)
  int W;                     // map width in cells (hit cells are reported as row * W + col)
  float res_xf, res_yf;  // float32 cell size: cell-centre coordinates are float32 products
  double res_x, res_y;
  double z_max, r2max;
  int B, NB;
  const double* tab_sorted;  // beam angles in radians (deg*pi/180), ascending
  const uint16_t* perm;      // sorted position -> caller's beam index
  const uint16_t* lut;       // bucket -> first position of the extended table at or after the bucket's lower edge
  const float* relx;         // [G + B] extended table: float32(angle - smallest angle); the first G entries are
                             // the last G beams moved down one turn (their images across the 0 / 2pi seam)
  int G;                     // number of such images
  double ext_min;            // relx[0] as float64 (<= 0): lower end of the bucket index
  double ts_min, ts_max, inv_bw;
  double half, two_pi_lo, two_pi_hi;
  // Sector intervals (float32-mixture instantiation): the 2B sector edges, sorted around the circle, cut it into
  // intervals inside each of which the set of sectors that hold a bearing does not change -- a contiguous run of beams.
  // A cell that is not within rounding of an edge then costs ONE minimum (its interval's) instead of one per beam, and
  // a beam's range is the smallest of the few intervals its sector spans.  iv_NE == 0: not available for this table.
  int iv_NE, iv_NB;            // intervals (<= 1023), buckets of the look-up (a power of two)
  int iv_cmax, iv_cmin;        // most / fewest intervals a sector spans
  int iv_stride;               // > 0: the sector of the beam at sorted position k starts at interval k * iv_stride (evenly
                               // spaced beams whose spacing divides the sector's width): no table read to find it
  const float* iv_edge;        // [NE + 1] lower edge of every interval, measured from iv_origin; iv_edge[NE] = 2 pi
  const uint16_t* iv_beam;     // [B] first interval | number of intervals (<= 63) << 10 of the beam at sorted position k
  const uint16_t* iv_lut;      // [NB] bucket -> last interval that starts at or below the bucket's lower edge
  const uint32_t* iv_ends;     // [NE] per edge m (the lower edge of interval m): the one beam whose sector ENDS there (bits 0-9,
                               // bit 10 = there is one), the one whose sector STARTS there (bits 11-20, bit 21), and bit 22 = some
                               // sector holds the edge in its interior.  A bearing within rounding of the edge is decided for
                               // exactly these beams with the reference's float64 predicate.
  double iv_origin;            // angle of edge 0 (the lowest sector edge)
  double iv_th_mid, iv_th_hi;  // headings with |heading - iv_th_mid| <= iv_th_hi take this path: there the reference's predicate
                               // (one turn of slack either way, mc_localization.py:148-149) is the circular one -- every beam
                               // angle + heading stays within (-2 pi + half-angle, 4 pi - half-angle), and a bearing in
                               // [0, 2 pi) is then never two turns away from it.  For beams over [0, 358] degrees with
                               // 5-degree sectors: headings from -355 to 357 degrees.
  // Two launches of the float32-mixture path: the interval-only kernel files the particles it does not serve ...
  int32_t* redo_list;          // 64 lists of sub_cap places each (blocks b, b + 64, ... append to list b % 64)
  uint32_t* redo_count;        // [64], zero before the launch
  int64_t sub_cap;
  // ... and the launch that follows takes its particles from that list
  const int32_t* list;
  const uint32_t* list_count;
  // beam mixture
  const double* zt;      // measured range per beam, in sorted-angle order
  const double* etalam;  // eta_short * lambda_short per beam, in sorted-angle order
  double c_hit;          // 1/sqrt(2*pi*sigma^2)
  double sigma2;         // sigma^2
  double lambda;
  double w_hit, w_short, w_max, w_rand;
  double p_rand;         // 1/z_max
  const double* range_noise;  // [n][B] additive noise on the predicted ranges, caller beam order (augmented MCL) or null
  int no_obstacles;      // the map holds no obstacle cell
  // augmented MCL with device-drawn range noise: sigma * N(0, 1) per (particle, beam), keyed by the global particle index
  double rn_sigma;
  uint64_t rn_seed;
  int64_t rn_index0;
  unsigned long long* stamps;  // profiling build only: cycles per kernel section
  // histogram-filter update (one pose per free grid cell)
  const int32_t* g_cell;       // flat cell index of pose p
  const double* g_pk_pred;     // [H*W*D]
  float* g_pk;                 // [H*W*D]
  int g_D, g_B0;               // headings, beams per heading
  int g_NE;                    // entries (headings x beams); the beam table proper holds their classes
  const uint32_t* g_cls;       // [NE] class << 20 | heading << 10 | beam of entry e
  const uint16_t* g_cls_first; // [B + 1]
  const uint16_t* g_member;    // [NE] entries by class
  const double* g_ar;          // [B0] beam angles in radians, [D] heading angles: an entry's own angle is their sum (robot.py:252)
  const double* g_thd;
  // the launch over the left-over list: block 0 stores the lists' counts (64 + 64) where the host reads them as a hint
  // for the next query (page-locked host memory; no copy of their own)
  uint32_t* hint_out;
  const uint32_t* hint_src;
};

// What the interval-only sector query needs of a particle, left by the motion kernel in one 32-byte record: the
// wave that serves the particle fetches it a particle ahead -- with two VECTOR loads in which every lane reads the
// same 16 bytes, so that the values wait in vector registers behind the in-order vector-memory counter until they
// are used (scalar loads share their counter with LDS: the first LDS read after one waits for it) -- and starts on
// the particle's list of cells without computing a tile or waiting for the tile's bounds (a dependent read from HBM
// per particle).
constexpr int kListReadAhead = 256;   // entries of RayParams::nbr_cells that may be read beyond the end of a list (allocated behind the last)
struct QueryRecord {
  float x, y, th;        // the pose as stored (float32 state)
  float tho;             // (heading + angle of interval edge 0) in turns: a bearing in turns minus this, its fraction
                         // taken, is the bearing's place on the circle of sector intervals
  uint32_t first_lo, first_hi;   // the slice of RayParams::nbr_cells the particle has to look at starts here (64 bits) ...
  uint32_t n_cells;      // ... and is this long (whole batches of 64)
  uint32_t flags;        // bit 0: the pose rounds to a free cell; bit 1: the heading is in the range in which the
                         // reference's predicate is the circular one (the interval path's first launch serves it);
                         // bit 2: a finite heading outside that range, for the second pass of the arithmetic interval
                         // kernel, which restores what the reference does at the seam of the bearings and for beam
                         // angles two or more turns away.  The slice is filled in for either.
};
static_assert(sizeof(QueryRecord) == 32, "two 16-byte loads");
// What the motion kernel reads of a query tile, in one 32-byte load: where the tile's list starts, how long it is, and
// how many batches of it hold the classes up to k.
struct QueryTile {
  uint64_t first;
  uint32_t n_cells, pad;
  uint16_t need[kNeedClasses];
};
static_assert(sizeof(QueryTile) == 32, "one 32-byte load");
struct QueryPrep {
  QueryRecord* out;      // null: no records are written
  const uint64_t* nbr_off;
  int ETW, ETH, rtx, rty;
  float inv_res_xf, inv_res_yf;
  float th_mid, th_hi;   // |heading - th_mid| <= th_hi: served by the interval path (RayParams::iv_th_*)
  float origin;          // RayParams::iv_origin
  // the list's classes (map_motion.hip: cell_class): need[k] = batches of 64 that hold the classes up to k; a pose
  // at distance rho from its tile's centre takes class ceil((rho + need_slack) * need_inv_step)
  const QueryTile* nbr_need;
  float res_xf, res_yf, need_slack, need_inv_step;
  int serve_all;         // every finite heading is served (the interval-only kernel at hand handles the ones outside the range)
};

// What the sector query that follows a motion launch needs beside the poses: the measurement's per-beam constants
// (map_motion.hip: beam_table_kernel) and the left-over lists' counts at zero.  One more block of the motion kernel
// does both, instead of a kernel and a fill of their own between the two big launches.
struct StepPrep {
  double* zt;            // null: nothing to prepare
  double* etalam;
  const double* zmeas;
  const uint16_t* perm;
  uint32_t* clear;       // [n_clear] words set to zero (n_clear <= 256), or null
  int B, n_clear;
  double lambda;
};

struct MotionParams {
  double v, omega, dt;
  double c1, c2, c3;     // b/6 for the three draws
  double res_x, res_y;
  const uint64_t* freeb; // free-cell bit plane, tiled like the obstacle plane
  int TW, TH, W, H;
  int noise_mode;        // 0 none, 1 pre-drawn integers, 2 device generator
  uint32_t ints_span;    // device generator: integers uniform on [-1000, -1000 + ints_span): 2001 (random_integers) or 2000 (randint)
  int book;              // 1: Table 5.3 as printed (05_robot_motion/sample_motion_model_velocity.py:90-108): no straight-line branch, no map
  uint64_t seed;
  int64_t index0;        // global index of particle 0 (keys the device generator)
  QueryPrep qp;          // records for the sector query that follows (float32 state, sector intervals), or out == null
  StepPrep prep;         // the query's per-step tables, prepared by one extra block (zt == null: not)
};

}  // namespace mcl

struct mclhip_ctx {
  int device = 0;
  int dtype = MCLHIP_F64;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;   // the stream the context created (stream may be a caller's)
  std::string err;

  // map
  int H = 0, W = 0, TH = 0, TW = 0;
  double res_x = 0, res_y = 0;
  uint64_t* obst = nullptr;     // obstacle bit plane (8x8-cell tiles)
  uint64_t* freeb = nullptr;    // free-cell bit plane
  uint32_t* tile_off = nullptr; // per-tile obstacle lists: offsets ...
  uint32_t* cells = nullptr;    // ... and entries
  uint64_t* nbr_off = nullptr;  // per-tile neighbourhood lists (built from the two above and the ring table)
  mcl::QueryTile* nbr_need = nullptr; // [query tiles] start, length and class prefixes of every list (QueryPrep::nbr_need)
  uint32_t* nbr_cells = nullptr;
  int ETW = 0, ETH = 0, rtx = 0, rty = 0;
  int64_t n_obstacles = 0;

  // sensor
  int B = 0, NB = 0;
  double z_max = 0, sigma = 0, lambda = 0, w4[4] = {0, 0, 0, 0}, half = 0;
  double ts_min = 0, ts_max = 0, inv_bw = 0;
  double* tab_sorted = nullptr;
  double* tab_orig = nullptr;
  uint16_t* perm = nullptr;
  uint16_t* inv_perm = nullptr;
  uint16_t* lut = nullptr;
  float* iv_edge = nullptr; uint16_t* iv_beam = nullptr; uint16_t* iv_lut = nullptr; uint32_t* iv_ends = nullptr;   // sector intervals (see RayParams)
  int iv_NE = 0, iv_NB = 0, iv_cmax = 0, iv_cmin = 0, iv_stride = 0; double iv_origin = 0.0, iv_th_hi = 0.0, iv_th_mid = 0.0; bool iv_uniform = false;   // (uniform: the intervals are equally wide)
  float* relx = nullptr;
  int G = 0;
  double ext_min = 0;
  double* zt = nullptr;      // [B]
  double* etalam = nullptr;  // [B]
  double* zmeas_stage = nullptr;  // [B] staging for z_meas uploads
  // page-locked copies of the caller's z_meas: a step that only enqueues must not leave the device reading the
  // caller's buffer after the call has returned.  kZSlots steps may be in flight before a slot is waited for.
  static constexpr int kZSlots = 8;
  double* zpin = nullptr;         // [kZSlots][B], host
  const double* zsrc = nullptr;   // the measurement the next sample step reads: a slot of zpin (the step's motion launch reads it in place)
  int zsrc_slot = -1;
  hipEvent_t zev[kZSlots] = {};   // upload of the slot has finished
  bool zev_set[kZSlots] = {};
  int zslot = 0;
  bool zmeas_ok = false;          // the last uploaded measurement is finite and >= 0 everywhere (what the float32 mixture assumes)
  uint16_t* ring = nullptr;       // tile visiting order of the sector query
  int n_ring = 0;
  bool ring_dirty = true;         // map resolution or z_max changed since the table was built

  // particles (structure of arrays; pose arrays are float or double by dtype)
  int64_t M = 0, cap = 0;
  int32_t* redo_list2 = nullptr;   // ... and what the second pass over that list left over in turn (its counts: redo_count + 64)
  int32_t* redo_list = nullptr; uint32_t* redo_count = nullptr; int64_t redo_cap = 0;   // particles the interval-only sector query left over
  bool redo_cleared = false;       // the motion launch just enqueued has set redo_count[] to zero (launch_raycast then need not)
  mcl::QueryRecord* qrec = nullptr;   // [cap] the motion kernel's records for the interval-only sector query
  static constexpr int64_t kRedoSlack = 64 * 16 * 2;   // the 64 lists are rounded up to whole blocks of waves
  // How many particles the interval-only launch left over last time (page-locked host copy of the 64 counters, written
  // by an asynchronous copy behind the second launch and read, without waiting, before the next query is enqueued --
  // a hint, a step old): when most of the cloud is left over (headings outside the interval path's range) the query
  // runs as one kernel, which serves such a cloud faster than two launches do, and probes again every eighth query.
  uint32_t* redo_hint = nullptr;
  std::string query_kernel;     // the sector-query instantiation the last query ran (its first launch, if it ran as two)
  // Headings outside the circular range (QueryRecord::flags bit 2): the arithmetic interval kernel can serve them in a
  // second pass, over the list of what the first launch left, before the entry-by-entry launch takes what is then
  // still left.  -1: when the first launch last left between a sixteenth and half of the cloud (a shorter list is served
  // as fast entry by entry -- the extra launch has a fixed cost; a longer one faster by the one kernel that holds both
  // paths, see split_mode.  Measured at 2^20 x 180: 3 % left over 0.749 against 0.750 ms a step, 17 % 0.870 against
  // 0.900, all of them 1.61 against 1.36); 0 never; 1 always (environment MCLHIP_EVERY_HEADING).
  int every_heading = -1;
  bool every_last = false;      // the last query made that second pass
  int split_mode = -1;          // -1 by the hint, 0 never two launches, 1 always (environment MCLHIP_SPLIT at creation)
  int64_t split_skipped = 0;
  void* px[2] = {nullptr, nullptr};
  void* py[2] = {nullptr, nullptr};
  void* pth[2] = {nullptr, nullptr};
  double* w[2] = {nullptr, nullptr};
  int cur = 0;               // which of the two buffers holds the live set
  double* logw = nullptr;
  uint8_t* pm = nullptr;
  // snapshot (benchmark helper)
  void* sx = nullptr; void* sy = nullptr; void* sth = nullptr; double* sw = nullptr; int64_t sM = 0;

  // scratch for normalise / resample
  double* chunk_sums = nullptr;  // [ceil(M/8192)]
  double* scal = nullptr;        // [8] device scalars: 0 sum, 1 max logw
  double* cdf_approx = nullptr;  // [M]
  double* cdf_exact = nullptr;   // [M]
  uint64_t* kscan = nullptr;     // [M] integer prefix sums
  uint32_t* fscan = nullptr;     // [M] irregular-element counts
  void* blk_a = nullptr;         // block partials (double / u64+u32)
  void* blk_b = nullptr;
  void* irr = nullptr;           // irregular-element records
  void* bases = nullptr;         // run bases
  int64_t* idx = nullptr;        // [M]
  int16_t* noise = nullptr;      // [M][3][12] staging for pre-drawn integers
  double* range_noise = nullptr; // [M][B] staging for the additive range noise of augmented MCL
  int64_t* inj_rows = nullptr;   // injected-particle staging (augmented MCL)
  double* inj_poses = nullptr;
  int64_t inj_cap = 0;
  uint32_t* counters = nullptr;  // small device counters
  unsigned long long* stamps = nullptr;  // [16] cycles per section (profiling build of the sector query)

  // multi-GPU: this context's slice of one global particle vector
  int64_t first_index = 0, m_total = 0;
  void* g_records = nullptr; int64_t g_records_cap = 0;   // global irregular-record list
  void* g_bases = nullptr;
  void* rows_send = nullptr; int64_t rows_send_cap = 0;   // packed rows handed to / received from the exchange
  void* rows_recv = nullptr; int64_t rows_recv_cap = 0;
  int64_t* draw_ends = nullptr;      // where the last draw plan left the draw ends (c->kscan or c->cdf_approx)
  bool dshard_tiles_ready = false;   // c->cdf_approx holds the tile sums of launch_dshard_weights
  void* rows_self = nullptr; int64_t rows_self_cap = 0;   // the fixed-block exchange: the rows a rank draws from itself
  int64_t* sel = nullptr; int64_t sel_cap = 0;            // local row behind every place of the send buffer (drawn-row exchange)
  int64_t drawn_total = 0;                                // rows the last plan sends in all
  double* small_d = nullptr; int64_t* small_i = nullptr;  // [1024] staging for boundary queries

  // histogram filter (grid localisation): belief over (row, col, heading)
  int gl_D = 0, gl_B0 = 0;          // headings, beams per heading
  double gl_res_theta = 0;          // degrees per heading bin
  int64_t gl_nfree = 0;             // cells at or above the obstacle threshold
  int64_t gl_nprior = 0;            // cells with a non-zero map value (support of the initial belief)
  uint8_t* gl_mask = nullptr;       // [H*W] 1 where the cell is free, 2 where only the raw map value is non-zero
  double* gl_x = nullptr; double* gl_y = nullptr; double* gl_th = nullptr;   // poses of the free cells
  int32_t* gl_cell = nullptr;       // their flat indices
  float* gl_pk = nullptr;           // [H*W*D] belief, float32 as the reference keeps it
  double* gl_pk_pred = nullptr;     // [H*W*D] prediction
  double* gl_K = nullptr;           // motion-model coefficients of the current control
  float* gl_part = nullptr;         // float32 chunk sums + total
  // (heading, beam) entries of the update and the classes of entries that share an absolute bearing
  int gl_NE = 0;                    // entries = headings x beams
  uint32_t* gl_cls = nullptr;       // [NE] class (position in the sorted class table) << 16 | beam k of entry e = d * B0 + k
  uint16_t* gl_cls_first = nullptr; // [classes + 1] start of each class in gl_member
  uint16_t* gl_member = nullptr;    // [NE] entries grouped by class
  double* gl_mbeta = nullptr;       // [B0] beam angles, then [D] heading angles, in radians: an entry's table angle is the sum of its two, as robot.py:252 forms it
  double* gl_zt = nullptr;          // [B0] measured range per beam
  double* gl_el = nullptr;          // [B0] exp(-lambda * z_meas)
  double* gl_zm = nullptr;          // [B0] staged measurement
  std::vector<double> gl_angles_deg;

  void* fs = nullptr;               // FastSLAM storage (fastslam.hip owns the type)

  // timing
  int timing = 0;            // 0 off, 1 every kernel group, 2 the sector-query kernel only
  bool span_open = false;    // the last span_begin recorded an event
  int64_t t_launch[6] = {0, 0, 0, 0, 0, 0};
  double t_ms[6] = {0, 0, 0, 0, 0, 0};
  struct Span { hipEvent_t a, b; int which; };
  std::vector<Span> spans;   // event pairs recorded but not yet read back
};

namespace mcl {

// error helpers (api translation unit owns the definitions)
int fail(mclhip_ctx* c, int code, const std::string& msg);
#define MCL_HIP(c, expr)                                                              \
  do {                                                                                \
    hipError_t e__ = (expr);                                                          \
    if (e__ != hipSuccess)                                                            \
      return mcl::fail((c), MCLHIP_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
  } while (0)

// kernels launchers, one per stage; all enqueue on c->stream
int launch_build_tiles(mclhip_ctx* c, const uint8_t* d_map);
int launch_beam_table(mclhip_ctx* c);
int launch_grid_beam_table(mclhip_ctx* c, const double* zmeas, int B0, int NE, double* zt, double* eshort);
int set_sensor_table(mclhip_ctx* c, const double* tab_rad, int B, double z_max, double sigma_hit, double lambda_short,
                     const double w4[4], double half);
int launch_grid_update(mclhip_ctx* c, const double* d_x, const double* d_y, const double* d_th, const int32_t* d_cell,
                       int64_t n, int n_dir, int beams_per_dir, const double* pk_pred, float* pk);
int build_ring_table(mclhip_ctx* c);
int launch_nbr_count(mclhip_ctx* c, uint32_t* d_cnt, uint32_t* d_cls_cnt);
int launch_nbr_fill(mclhip_ctx* c, uint32_t* d_cls_cnt);
int launch_motion(mclhip_ctx* c, void* x, void* y, void* th, uint8_t* pm, int64_t n, const MotionParams& mp,
                  const int16_t* d_noise, int dtype);
int launch_inject_device(mclhip_ctx* c, int which_buffer, const mcl::MotionParams& mp, double p_random, double ext_x,
                         double ext_y, uint64_t seed, bool set_weight, uint32_t* d_counters);
int launch_restore(mclhip_ctx* c, void* x, void* y, void* th, double* w, const void* sx, const void* sy, const void* sth,
                   const double* sw, int64_t n, int dtype);
int launch_init_particles(mclhip_ctx* c, void* x, void* y, void* th, int64_t n, const MotionParams& mp, int gaussian,
                          double ext_x, double ext_y, const double* x0, uint64_t seed, uint32_t* d_failures);
// timing events of group `which` around what is launched between the two calls (no-ops unless timing is enabled)
void timing_begin(mclhip_ctx* c, int which);
void timing_end(mclhip_ctx* c);
int launch_raycast(mclhip_ctx* c, const void* x, const void* y, const void* th, const uint8_t* pm, int64_t n,
                   int dtype, double* w_out, double* logw_out, double* ranges_out, int32_t* hits_out,
                   const double* range_noise = nullptr, double rn_sigma = 0.0, uint64_t rn_seed = 0,
                   const QueryRecord* qrec = nullptr);
// fills mp.qp so that the motion kernel leaves the records of the sector query that follows (or out = null: none)
void make_query_prep(const mclhip_ctx* c, QueryPrep& qp);
// the classes of a neighbourhood list (map_motion.hip: cell_class): a little beyond z_max for what rounding can move a
// cell or a pose; the step is a seventh of the farthest a pose can stand from its query tile's centre
inline double nbr_class_slack(const mclhip_ctx* c) { return 0.02 * (c->res_x < c->res_y ? c->res_x : c->res_y); }
inline double nbr_class_step(const mclhip_ctx* c) {
  const double hx = 0.5 * kQueryCells * c->res_x, hy = 0.5 * kQueryCells * c->res_y;
  return std::sqrt(hx * hx + hy * hy) / (kNeedClasses - 1);
}
// > 0: the beam table at hand is one of the interval-only kernel's arithmetic layouts (that many intervals per sector)
int ray_arith_span(const mclhip_ctx* c);
// ... and that kernel also serves the headings outside the circular range
bool ray_serves_every_heading(const mclhip_ctx* c);
int launch_mean_weight(mclhip_ctx* c);
int launch_exp_weights(mclhip_ctx* c);   // w = exp(logw) for the live set
int launch_sequential_total(mclhip_ctx* c);   // c->cdf_exact = strictly sequential running sum of the live weights
void fs_release(mclhip_ctx* c);
int launch_inject(mclhip_ctx* c, const int64_t* d_rows, const double* d_poses, int64_t n_inj, int which_buffer,
                  bool set_weight);
int launch_compute_px(mclhip_ctx* c, const double* z, const double* zt, int64_t n, double* p);
int launch_normalise(mclhip_ctx* c, int mode);
int launch_resample(mclhip_ctx* c, double r, int algo);
int launch_normalise_resample(mclhip_ctx* c, int mode, double r);   // the two above in six launches (fused update)
int launch_shard_max_logw(mclhip_ctx* c, double* d_also = nullptr);   // (d_also: a second place for the maximum, on the device)
int launch_shard_weights(mclhip_ctx* c, int mode, double gmax);
int launch_dshard_weights(mclhip_ctx* c, const double* d_gmax, double* d_chunks_out);
int launch_shard_scan(mclhip_ctx* c, double total, double approx_offset, int64_t first_index, const double* d_offset = nullptr);
struct ShardLayout {   // who holds what of the global particle vector (at most 64 ranks)
  int world, rank, cmax;
  int nchunks[64];       // 8192-element chunks per rank
  int64_t first[65];     // global index of every rank's first particle; first[world] = total
};
int launch_dshard_scan(mclhip_ctx* c, const double* d_all_chunks, const ShardLayout& lay, void* d_rec_out, int cap);
int launch_dshard_finalize(mclhip_ctx* c, const void* d_all_rec, const ShardLayout& lay, int cap, double r, int64_t* d_counts);
int launch_fs_dshard_scan(mclhip_ctx* c, const double* d_all_chunks, const double* d_lasts, const ShardLayout& lay,
                          void* d_rec_out, int cap);
int launch_fs_dshard_finalize_raw(mclhip_ctx* c, const void* d_all_rec, const ShardLayout& lay, int cap, double* d_last_out);
int launch_shard_finalize(mclhip_ctx* c, const void* d_records, int64_t nrec, void* d_bases, uint64_t P_off,
                          uint32_t F_off, const unsigned long long* d_meta = nullptr);
int launch_count_below(mclhip_ctx* c, const double* d_U, int nu, int64_t* d_counts);
int launch_plan_drawn(mclhip_ctx* c, const ShardLayout& lay, double r, int64_t* d_counts, const unsigned long long* meta,
                      bool ends_done = false);   // (ends_done: the draw ends are in c->kscan already)
int launch_pack_drawn(mclhip_ctx* c, int world, double* d_dst, int64_t* d_sel);
int launch_unpack_drawn(mclhip_ctx* c, const double* d_rows, int64_t L, double r, bool keep_pos);
int launch_pack_fixed(mclhip_ctx* c, int world, int rank, int64_t cap, double* d_dst, double* d_self,
                      const int64_t* d_all_counts, int64_t* d_flag_out, int64_t* h_late);
int launch_unpack_fixed(mclhip_ctx* c, const double* d_recv, const double* d_self, const int64_t* d_all_counts, int world,
                        int rank, int64_t cap, double r, int64_t* d_flag_out);
int launch_pack_rows(mclhip_ctx* c, int64_t lo, int64_t hi, double* d_dst);
int launch_unpack_resample(mclhip_ctx* c, const double* d_rows, int64_t L, int64_t first_index, int64_t m_total,
                           double r, int64_t row0_global);
RayParams make_ray_params(const mclhip_ctx* c);

}  // namespace mcl
