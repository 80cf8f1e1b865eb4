/*
 * mclhip.h -- C ABI of libmclhip.so: the Monte Carlo localization update
 * (motion sampling, sector range query, beam likelihood, weight normalisation,
 * low-variance resampling) as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (trip2eee/slam) is pure Python and has no FFI of its own; the
 * interface each entry point stands in for is the Python method or expression
 * cited beside it (paths relative to the reference checkout,
 * 08_grid_and_montecarlo_localization/ unless another directory is named).
 *
 * Conventions
 *   - plain C, no C++ types, no exceptions across the boundary;
 *   - every call returns 0 on success and a negative MCLHIP_E_* code on error;
 *     the message is available from mclhip_last_error();
 *   - host buffers are borrowed for the duration of the call, contiguous, C order;
 *   - the library owns all device memory; one context drives one GPU (one
 *     process per GPU; ranks are combined by the host, see the shard_* calls);
 *   - a context is not thread-safe: one caller thread per context;
 *   - calls are synchronous (they return after the device work has finished)
 *     except mclhip_step_async(), mclhip_restore_particles_async() and the
 *     mclhip_dshard_*() calls, which only enqueue on the context's stream; every
 *     synchronous call first waits for what those have enqueued.
 */
#ifndef MCLHIP_H
#define MCLHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mclhip_ctx mclhip_ctx;

enum {
  MCLHIP_OK = 0,
  MCLHIP_E_ARG = -1,     /* bad argument / call order */
  MCLHIP_E_HIP = -2,     /* a HIP runtime call failed */
  MCLHIP_E_STATE = -3,   /* map / sensor / particles not set yet */
  MCLHIP_E_LIMIT = -4    /* size beyond what the kernels support */
};

/* dtype of the particle state kept on the device */
enum { MCLHIP_F32 = 0, MCLHIP_F64 = 1 };

/* how mclhip_normalise forms the weights it normalises */
enum {
  MCLHIP_W_PRODUCT = 0,  /* w = prod_k p_k * pm, the reference's raw product (mc_localization.py:235-240) */
  MCLHIP_W_LOG = 1       /* w = exp(logw - max logw): same weights after normalisation, no underflow */
};

/* which CDF walk mclhip_resample uses; both give identical indices */
enum {
  MCLHIP_RESAMPLE_SERIAL = 0,   /* one lane replays the reference's loop (mc_localization.py:253-259) */
  MCLHIP_RESAMPLE_PARALLEL = 1  /* order-exact parallel scan (default) */
};

/* Library / context ------------------------------------------------------ */

/* ABI version of this header (bumped on incompatible change). */
int mclhip_abi_version(void);

/* Number of HIP devices visible to the process (no context needed). */
int mclhip_device_count(void);

/* Replaces MCLocalization.__init__ (mc_localization.py:36-56) for the device
 * side: creates the context on GPU `device`.  dtype: MCLHIP_F32 / MCLHIP_F64. */
int mclhip_create(mclhip_ctx** out, int device, int dtype);
void mclhip_destroy(mclhip_ctx* ctx);

/* Text of the last error on this context (ctx == NULL: last create error). */
const char* mclhip_last_error(mclhip_ctx* ctx);

/* Map and sensor --------------------------------------------------------- */

/* Replaces make_obstacle_lut (mc_localization.py:61-71) and the free-cell
 * test's map (mc_localization.py:109).  map: uint8 [H][W]; obstacle == 0,
 * free == 255.  Kept on the device as two bit planes in 8x8-cell tiles. */
int mclhip_set_map(mclhip_ctx* ctx, const uint8_t* map, int H, int W, double res_x, double res_y);

/* Replaces the sensor_angles list (mc_localization.py:45-48), the sector
 * half-angle (mc_localization.py:139) and RangeSensor's constants
 * (range_sensor.py:5-28).  w4 = {w_hit, w_short, w_max, w_rand}, already
 * normalised.  angles_deg in degrees, any order, B <= 1024. */
int mclhip_set_sensor(mclhip_ctx* ctx, const double* angles_deg, int B, double z_max, double sigma_hit,
                      double lambda_short, const double w4[4], double sector_half_angle_rad);

/* Particle state --------------------------------------------------------- */

/* Replaces assignment of self.X / self.w (mc_localization.py:40-41).
 * X_aos: [M][3] (x, y, theta); w: [M] or NULL (-> 1/M). */
int mclhip_set_particles(mclhip_ctx* ctx, const double* X_aos, const double* w, int64_t M);

/* Replaces MCLocalization.initialize (mc_localization.py:173-219) for particle
 * counts at which the reference's Python rejection loop is not an option: M poses
 * drawn on the device on free cells (value 255 after the reference's rounding) --
 * x0 == NULL: uniform over [0, extent_x) x [0, extent_y), heading U[0, 2pi)
 * (:176-196); else normal around x0[3] with sigma 0.1 * extent, heading
 * N(x0[2], pi) (:197-219).  Same distribution as the reference, not numpy's
 * stream; weights 1/M. */
int mclhip_init_particles(mclhip_ctx* ctx, int64_t M, const double* x0, double extent_x, double extent_y,
                          uint64_t seed);

/* Replaces reads of self.X / self.w.  Any pointer may be NULL.
 * logw: sum_k log p_k of the last sample (-inf where the motion left the free space). */
int mclhip_get_particles(mclhip_ctx* ctx, double* X_aos, double* w, double* logw);

int64_t mclhip_num_particles(mclhip_ctx* ctx);

/* Stage entry points (each is one stage of the update, usable on its own) - */

/* Replaces MCLocalization.measure(xt, noise=False) (mc_localization.py:116-171)
 * over n poses.  ranges: [n][B]; hit_cell (nullable): [n][B] row-major index
 * of the nearest obstacle cell inside each beam's sector, -1 = none. */
int mclhip_raycast(mclhip_ctx* ctx, const double* poses_aos, int64_t n, double* ranges, int32_t* hit_cell);

/* mclhip_raycast for callers that cannot allocate device memory (a serving loop
 * that measures one true pose per step, harnesses that forbid hipMalloc).  No
 * counterpart in the reference.  Same arguments and results as mclhip_raycast, but
 * the staging the kernels need (poses, ranges and hit cells on the device) lives in
 * a 960-KiB array inside the library's code object instead of being allocated and
 * freed per call: the call performs no hipMalloc / hipFree.  It is synchronous
 * (copies from and to the caller's host buffers, then waits for the stream), so it
 * cannot itself be recorded into a HIP graph.  Limits: n <= 4096 and n * B <= 65536,
 * more is refused with MCLHIP_E_LIMIT.  Map and sensor must both be set (the map
 * tables are built by mclhip_set_map / mclhip_set_sensor, whichever comes second).
 * Every context on one GPU shares that GPU's array; the library serialises the calls
 * of one process on it (a mutex per device, held from the upload of the poses to the
 * read-back of the results), so contexts driven from different threads do not mix
 * their data -- they wait for each other.  Other processes have their own copy. */
int mclhip_raycast_static(mclhip_ctx* ctx, const double* poses_aos, int64_t n, double* ranges, int32_t* hit_cell);

/* Replaces MCLocalization.sample_motion_model_velocity (mc_localization.py:74-114)
 * over n poses.  noise_ints: [n][3][12] integers in [-1000, 1000], the values
 * np.random.random_integers(-1000, 1000, 12) would return for the three draws
 * of each pose (mc_localization.py:20-24, :89-91), or NULL for noise=False.
 * poses_out [n][3], pm_out [n] (1.0 on a free cell, else 0.0). */
int mclhip_motion(mclhip_ctx* ctx, const double* poses_aos, int64_t n, double v, double omega, double dt,
                  const double alpha[6], const int16_t* noise_ints, double* poses_out, double* pm_out);

/* Replaces sample_motion_model_velocity(ut, xt_1, dt) of the chapter-5 script
 * (05_robot_motion/sample_motion_model_velocity.py:90-108), Table 5.3 as printed:
 * the arc update divides by w_hat whatever its value (w_hat == 0 gives inf / NaN
 * poses, as numpy's float64 does) and no map is involved, so the context needs
 * none.  noise_ints as in mclhip_motion (:73-77 draws them with
 * random_integers(-1000, 1000, 12)); NULL = no noise.  poses_out [n][3]. */
int mclhip_motion_book(mclhip_ctx* ctx, const double* poses_aos, int64_t n, double v, double omega, double dt,
                       const double alpha[6], const int16_t* noise_ints, double* poses_out);

/* Replaces RangeSensor.compute_px(z, z_true) (range_sensor.py:95-112) over n pairs. */
int mclhip_compute_px(mclhip_ctx* ctx, const double* z, const double* z_true, int64_t n, double* p_out);

/* Replaces the particle loop of MCLocalization.sample (mc_localization.py:226-240):
 * motion sample, sector query and beam likelihood for every particle; leaves
 * the moved poses, w = prod p * pm and logw on the device.
 * noise_ints: [M][3][12] pre-drawn integers as in mclhip_motion, or NULL to
 * draw them on the device (counter-based generator keyed by `seed`).
 * z_meas: [B] measured ranges (self.sensor_ranges, mc_localization.py:224). */
int mclhip_sample(mclhip_ctx* ctx, double v, double omega, double dt, const double alpha[6],
                  const int16_t* noise_ints, uint64_t seed, const double* z_meas);

/* Replaces `self.w = self.w / np.sum(self.w)` (mc_localization.py:248); the sum
 * is accumulated in the order np.sum uses, so it is bit-equal to numpy's.
 * mode: MCLHIP_W_PRODUCT / MCLHIP_W_LOG.  sum_out nullable. */
int mclhip_normalise(mclhip_ctx* ctx, int mode, double* sum_out);

/* Replaces the walk of MCLocalization.resample (mc_localization.py:250-262)
 * after normalisation: r = np.random.rand() * 1/M is drawn by the caller.
 * Gathers poses and weights (weights are carried, not reset).  An index that
 * would run past the last particle is clamped to M-1 (the reference raises
 * IndexError there; 13_fast_slam/fast_slam1.py:426 clamps).
 * idx_out (nullable): [M] source index of every resampled particle. */
int mclhip_resample(mclhip_ctx* ctx, double r, int algo, int64_t* idx_out);

/* mclhip_normalise + mclhip_resample (the whole of MCLocalization.resample,
 * mc_localization.py:242-262) in one call and in the six kernels the fused update
 * mclhip_step uses for them, on the weights the context holds: the same sum, the
 * same normalised weights, the same indices, bit for bit.  sum_out, idx_out nullable. */
int mclhip_normalise_resample(mclhip_ctx* ctx, int mode, double r, double* sum_out, int64_t* idx_out);

/* One whole update = mclhip_sample + mclhip_normalise + mclhip_resample
 * (mc_localization.py:329-331) without intermediate host synchronisation.
 * mclhip_step_async returns after enqueueing; mclhip_sync waits. */
int mclhip_step(mclhip_ctx* ctx, double v, double omega, double dt, const double alpha[6],
                const int16_t* noise_ints, uint64_t seed, const double* z_meas, int weight_mode, double r);
int mclhip_step_async(mclhip_ctx* ctx, double v, double omega, double dt, const double alpha[6],
                      uint64_t seed, const double* z_meas, int weight_mode, double r);
int mclhip_sync(mclhip_ctx* ctx);

/* Augmented MCL (mc_localization2.py, Table 8.3) -----------------------------
 * Replaces the particle loop of its sample() (mc_localization2.py:231-246): as
 * mclhip_sample, plus range_noise [M][B] (caller beam order, nullable): the
 * term randn()*0.15 added to every predicted range (:166-168), and the running
 * mean w_avg = sum_m w[m]/M accumulated strictly left to right (:246), which is
 * returned in w_avg_out.  noise_ints here come from np.random.randint(-1000,
 * 1000, 12) (:23). */
int mclhip_sample_augmented(mclhip_ctx* ctx, double v, double omega, double dt, const double alpha[6],
                            const int16_t* noise_ints, uint64_t seed, const double* z_meas,
                            const double* range_noise, double* w_avg_out);

/* Replaces the walk of its resample() (mc_localization2.py:266-296) after
 * mclhip_normalise: positions inj_rows[j] (ascending) receive the pose
 * inj_poses[j] and the weight 1/M instead of a resampled particle.  As in the
 * reference, the injected poses overwrite X in place first, so a position that
 * resamples source row inj_rows[j] gets the injected pose too. */
int mclhip_resample_augmented(mclhip_ctx* ctx, double r, const int64_t* inj_rows, const double* inj_poses,
                              int64_t n_inj, int64_t* idx_out);

/* The two calls above for particle counts at which the host cannot draw the numbers (one normal per particle
 * and beam): the motion integers (uniform on [-1000, 1000), as np.random.randint(-1000, 1000, 12) gives them,
 * mc_localization2.py:23), the range noise range_sigma * N(0, 1) (0.15 in the reference, :166) and the
 * injection draws (:266-283: one uniform per position, injected when <= p_random; pose uniform over
 * [0, extent_x) x [0, extent_y) on a cell of value 255, heading U[0, 2pi)) come from the counter-based
 * generator keyed by the global particle index and `seed`.  Same distributions as the reference, not numpy's
 * stream.  n_injected_out (nullable): number of injected positions.
 * With float32 state the noise is drawn inside the float32 beam mixture of the interval-only sector query and the raw
 * weights are the exponentials of the log-weights (relative error about 1e-7, and a noise stream of its own); with
 * float64 state the product is formed factor by factor as the reference does (mc_localization2.py:235-240). */
int mclhip_sample_augmented_device(mclhip_ctx* ctx, double v, double omega, double dt, const double alpha[6],
                                   uint64_t seed, const double* z_meas, double range_sigma, double* w_avg_out);
int mclhip_resample_augmented_device(mclhip_ctx* ctx, double r, double p_random, double extent_x, double extent_y,
                                     uint64_t seed, int64_t* n_injected_out);

/* Re-load the particle set from a device-resident snapshot taken by
 * mclhip_snapshot_particles (benchmark helper: every timed step starts from
 * the same synthetic cloud; a device-to-device copy, no host traffic). */
int mclhip_snapshot_particles(mclhip_ctx* ctx);
int mclhip_restore_particles_async(mclhip_ctx* ctx);

/* One filter over several GPUs ---------------------------------------------
 * One process and one context per GPU; rank g owns the contiguous global
 * particle range [first_index, first_index + M_local).  The reference is a
 * single-process loop, so these calls have no counterpart there: they are the
 * per-rank halves of mc_localization.py:248 (the sum) and :250-262 (the walk);
 * the host combines the small per-rank results (sums, record lists, counts)
 * across ranks and moves the packed rows.  Pointers named `rows` may be host
 * or device memory (the library copies with hipMemcpyDefault). */

/* Declares which slice of the global particle vector this context holds. */
int mclhip_shard_set(mclhip_ctx* ctx, int64_t first_index, int64_t m_total);

/* mclhip_sample for the local slice; the device generator is keyed by the
 * GLOBAL particle index, so the draws do not depend on the sharding.
 * weight_mode MCLHIP_W_LOG skips the raw product. */
int mclhip_shard_sample(mclhip_ctx* ctx, double v, double omega, double dt, const double alpha[6],
                        const int16_t* noise_ints, uint64_t seed, const double* z_meas, int weight_mode);

/* Largest log-weight of the local slice (MCLHIP_W_LOG only). */
int mclhip_shard_max_logw(mclhip_ctx* ctx, double* max_out);

/* Forms the local raw weights (MCLHIP_W_LOG: exp(logw - gmax)) and returns the
 * sums of the local 8192-element chunks, each in numpy's pairwise order.
 * M_local must be a multiple of 8192 on every rank but the last, so that the
 * chunks are those of the global vector. */
int mclhip_shard_weights(mclhip_ctx* ctx, int mode, double gmax, double* chunk_sums_out, int64_t n_chunks);

/* Divides by the global sum and runs the local half of the order-exact scan.
 * approx_offset: approximate sum of the normalised weights of all lower ranks.
 * Returns the local integer total and the number of local irregular records. */
int mclhip_shard_scan(mclhip_ctx* ctx, double total, double approx_offset, uint64_t* k_total, uint32_t* n_records);

/* Copies the local irregular records out: n_records x {int64 global index,
 * double weight, uint64 local integer prefix} (24 bytes each). */
int mclhip_shard_get_records(mclhip_ctx* ctx, void* records_out, int64_t cap_records);

/* Takes the records of ALL ranks in rank order, integer prefixes already made
 * global by the caller, replays them and finishes the local running sums.
 * P_off / F_off: integer total / record count of all lower ranks. */
int mclhip_shard_finalize(mclhip_ctx* ctx, const void* all_records, int64_t n_all, uint64_t P_off, uint32_t F_off);

/* counts_out[j] = number of local running sums that are < U[j]. */
int mclhip_shard_count_below(mclhip_ctx* ctx, const double* U, int n, int64_t* counts_out);

/* Packs the local particles [lo[s], hi[s]) of every slice s, one after the
 * other, as rows of five doubles (x, y, theta, w, running sum) into `rows`. */
int mclhip_shard_pack(mclhip_ctx* ctx, const int64_t* lo, const int64_t* hi, int nslices, void* rows);

/* Resamples the local positions from L received rows (global source indices
 * row0_global ...): index = number of received running sums < r + m/M_total,
 * clamped to the last row; gathers pose and weight.  idx_out nullable. */
int mclhip_shard_unpack_resample(mclhip_ctx* ctx, const void* rows, int64_t L, double r, int64_t row0_global,
                                 int64_t* idx_out);

/* The same exchange with only the rows that some position draws (mc_localization.py:253-259
 * and fast_slam1.py:421-431 read X[i] / Y[i] only for the drawn i): after mclhip_shard_finalize,
 * counts_out[d] (host, [world]) = rows this rank sends to rank d; mclhip_shard_pack_drawn writes
 * them destination after destination as rows of six doubles (x, y, theta, w, running sum, global
 * index as int64 bits); mclhip_shard_unpack_drawn resamples the local positions from the L rows
 * received from all ranks (ascending global index) and remembers which received row every
 * position took (for mclhip_fs_shard_unpack). */
int mclhip_shard_plan_drawn(mclhip_ctx* ctx, const int64_t* firsts, int world, int rank, double r, int64_t* counts_out);
int mclhip_shard_pack_drawn(mclhip_ctx* ctx, int world, void* rows);
int mclhip_shard_unpack_drawn(mclhip_ctx* ctx, const void* rows, int64_t L, double r, int64_t* idx_out);

/* Device-driven variant of the multi-GPU step.  No counterpart in the reference
 * (it is single-process).  Same arithmetic as the mclhip_shard_* calls above, but
 * every call only enqueues on the context's stream and exchanges data through
 * DEVICE pointers (e.g. torch tensors that RCCL collectives read and write), so a
 * whole step needs one device-to-host read (the row counts that size the exchange).
 * Only rows that some position draws travel.  Log-weight mode only.
 * firsts [world + 1]: global index of every rank's first particle, firsts[world]
 * = total.  cap: irregular records per rank that fit the exchange block of
 * 16 + 24 * cap bytes. */

/* Run the context on a caller's HIP stream (NULL: back to its own). */
int mclhip_set_stream(mclhip_ctx* ctx, void* hip_stream);
/* motion + sector query + likelihood, then the local maximum log-weight -> *d_max_out. */
int mclhip_dshard_sample_max(mclhip_ctx* ctx, double v, double omega, double dt, const double alpha[6],
                             const int16_t* noise_ints, uint64_t seed, const double* z_meas, int weight_mode,
                             double* d_max_out);
/* w = exp(logw - *d_gmax); this rank's 8192-element chunk sums -> d_chunks_out. */
int mclhip_dshard_weights(mclhip_ctx* ctx, const double* d_gmax, double* d_chunks_out);
/* d_all_chunks [world][cmax] (cmax = largest chunk count of a rank): global sum in
 * rank order, normalisation, local scan; header + records -> d_records_out. */
int mclhip_dshard_scan(mclhip_ctx* ctx, const double* d_all_chunks, const int64_t* firsts, int world, int rank,
                       void* d_records_out, int cap);
/* d_all_records [world][16 + 24*cap bytes]: global record list, exact local running
 * sums; then which local rows are drawn at all (mc_localization.py:253-259 reads X[i]
 * only for the drawn i) and by the positions of which rank.  d_counts_out [world + 1]:
 * number of rows this rank sends to every rank, last entry = 1 if a rank overflowed cap. */
int mclhip_dshard_finalize(mclhip_ctx* ctx, const void* d_all_records, const int64_t* firsts, int world, int rank,
                           int cap, double r, int64_t* d_counts_out);
/* The drawn rows for rank 0, then for rank 1, ... (d_counts_out[d] rows each) into d_rows:
 * six doubles per row -- x, y, theta, w, running sum, global index (as int64 bits). */
int mclhip_dshard_pack(mclhip_ctx* ctx, int world, void* d_rows);
/* Resamples the local positions from the L received rows (all sources, ascending global
 * index): ancestor = first row whose running sum is not below r + m/M_total, the last row
 * if there is none; gathers pose and weight.  No wait. */
int mclhip_dshard_unpack_resample(mclhip_ctx* ctx, const void* d_rows, int64_t L, double r);
/* The same exchange without a host read in the step (no counterpart in the reference, which is one process).
 * Every pair of ranks exchanges a block of cap_rows rows whatever it holds -- d_blocks [world][cap_rows][6] doubles,
 * sent and received with equal splits -- and the rows a rank draws from itself stay in a buffer of the context.
 *   mclhip_dshard_pack_fixed    after mclhip_dshard_finalize and the gathering of every rank's d_counts_out into
 *                               d_all_counts [world][world + 1] (still on the device): the drawn rows for rank d into
 *                               block d (what exceeds cap_rows is left out), the rank's own into the context's buffer;
 *                               *d_flag_out (device memory) as below -- known here already: with h_late_out (page-locked
 *                               host memory the device can write, [world * (world + 1) + 1] words, or NULL) the kernel
 *                               stores the gathered counts and the flag there itself, and a caller that records an
 *                               event behind this call waits for that event instead of for the step's end
 *   mclhip_dshard_unpack_fixed  d_blocks: the received blocks (block s from rank s).  *d_flag_out
 *                               = 0: the local positions have been resampled as by mclhip_dshard_unpack_resample;
 *                               1: some rank's record exchange overflowed; 2: some pair has more rows than cap_rows --
 *                               then NOTHING was resampled.  The flag is the same on every rank (it is formed from the
 *                               whole matrix); the caller reads it once the step has been enqueued, and on 1 or 2
 *                               calls mclhip_dshard_unpack_undo (the call has swapped the context's particle arrays
 *                               on the host as every resampling call does) and finishes the step with exact counts
 *                               (mclhip_dshard_pack / _unpack_resample) or on the host path.
 * Neither call waits. */
int mclhip_dshard_pack_fixed(mclhip_ctx* ctx, int world, int rank, int64_t cap_rows, void* d_blocks,
                             const int64_t* d_all_counts, int64_t* d_flag_out, int64_t* h_late_out);
int mclhip_dshard_unpack_fixed(mclhip_ctx* ctx, const void* d_blocks, const int64_t* d_all_counts, int world, int rank,
                               int64_t cap_rows, double r, int64_t* d_flag_out);
int mclhip_dshard_unpack_undo(mclhip_ctx* ctx);
/* Ancestor indices of the last resampling (waits for the stream). */
int mclhip_get_indices(mclhip_ctx* ctx, int64_t* idx_out);

/* Grid (histogram) localisation -- robot.py / grid_localization.py (Table 8.1) --
 * on the same context.  The belief lives on the device as float32 [H][W][D]
 * (the reference keeps float32, robot.py:49-60), the prediction as float64. */

/* Replaces Robot.__init__'s map handling and make_obstacle_lut (robot.py:40-60,
 * :195-205): cells with map < obstacle_below are obstacles, cells with
 * map >= obstacle_below are free (robot.py:20, :157, :199 use 200); the initial
 * belief is supported on cells with map != 0 (robot.py:53). */
int mclhip_grid_set_map(mclhip_ctx* ctx, const uint8_t* map, int H, int W, double res_x, double res_y,
                        int obstacle_below);

/* Replaces the sensor set-up of Robot.__init__ (robot.py:28-36) for n_dir
 * headings of res_theta_deg degrees (robot.py:8, :43): beam k of heading d points
 * along angles_deg[k]*pi/180 + d*res_theta_deg*pi/180 (robot.py:252, :289).
 * n_dir * B0 <= 1024.  Allocates the belief (zero). */
int mclhip_grid_set_sensor(mclhip_ctx* ctx, const double* angles_deg, int B0, int n_dir, double res_theta_deg,
                           double z_max, double sigma_hit, double lambda_short, const double w4[4],
                           double sector_half_angle);

/* Uniform belief over the cells with map != 0 (robot.py:49-60). */
int mclhip_grid_init_belief(mclhip_ctx* ctx);
/* Host <-> device copies of the belief [H][W][D] float32 and the prediction
 * [H][W][D] float64 (either output pointer may be NULL). */
int mclhip_grid_set_belief(mclhip_ctx* ctx, const float* pk);
int mclhip_grid_set_prediction(mclhip_ctx* ctx, const double* pk_pred);
int mclhip_grid_get_belief(mclhip_ctx* ctx, float* pk, double* pk_pred);

/* Replaces the belief part of Robot.predict (robot.py:148-187): pk_pred =
 * 7x7xD neighbourhood sum of pk * motion_model_velocity (robot.py:76-122) on
 * interior free cells, 0 elsewhere.  All D headings are neighbours (the
 * reference hard-codes its four, robot.py:168). */
int mclhip_grid_predict(mclhip_ctx* ctx, double v, double omega, double dt, const double alpha[6]);

/* Replaces Robot.update (robot.py:277-310) given the measured ranges
 * z_meas[B0]: pk = float32(pk_pred * prod_k compute_px(z_meas[k], z_pred[k]))
 * on free cells, 0 elsewhere, then pk /= pk.sum() in float32.  sum_out
 * (nullable) receives that float32 sum. */
int mclhip_grid_update(mclhip_ctx* ctx, const double* z_meas, float* sum_out);

/* FastSLAM 1.0 with unknown correspondence -- 13_fast_slam/fast_slam1.py -- on a
 * float64 context.  Particle poses and weights share the particle-filter state
 * (mclhip_get_particles); landmark features live in `capacity` slots per particle. */

/* Replaces FastSLAM.__init__ (fast_slam1.py:98-108): M particles at the origin,
 * no features. */
int mclhip_fs_init(mclhip_ctx* ctx, int64_t M, int capacity);

/* Module constants of fast_slam1.py:27-39 (STD_R, STD_PHI, r_max, P0, thres_sig);
 * the defaults are the reference's values. */
int mclhip_fs_set_model(mclhip_ctx* ctx, double std_r, double std_phi, double r_max, double p0, double thres_sig);

/* Replaces FastSLAM.fast_slam (fast_slam1.py:240-433) without the plotting.
 * truth_r/truth_phi/truth_valid [L]: range and bearing of every landmark from the
 * TRUE pose before noise and whether it is sensed (measure(), :131-142); each
 * particle adds its own noise to them (:145-146 inside the particle loop, :288).
 * ints [M][3][12] int16: the integers of the pose sampler (:163), NULL = device
 * generator.  normals [M][2 * #valid]: the standard normals of the measurements
 * in consumption order, NULL = device generator keyed by `seed`.
 * r = np.random.rand() * 1/M (:423).  idx_out [M] nullable: ancestor indices.
 * Returns MCLHIP_E_LIMIT when a particle needs more feature slots than capacity.
 * Order on the device: the weights of all particles first (they need the association
 * and the decay bookkeeping, not the updated means), the walk of :421-431, and then the
 * updates of :318-404 for the drawn ancestors only, whose slots their descendants copy. */
int mclhip_fs_step(mclhip_ctx* ctx, double v, double omega, double dt, const double alpha[6], const double* truth_r,
                   const double* truth_phi, const uint8_t* truth_valid, int L, const int16_t* ints,
                   const double* normals, uint64_t seed, double r, int64_t* idx_out);

/* Particle and feature state, all pointers nullable: pose [M][3], w [M],
 * nfeat [M], new_id [M] (Particle.new_feature_id), f_m [M][cap][2],
 * f_cov [M][cap][2][2], f_i [M][cap] (observation counters), f_w [M][cap],
 * f_id [M][cap]; unused slots read 0 (f_id: -1). */
int mclhip_fs_get_state(mclhip_ctx* ctx, double* pose, double* w, int32_t* nfeat, int32_t* new_id, double* f_m,
                        double* f_cov, int32_t* f_i, double* f_w, int32_t* f_id);

/* One FastSLAM filter over several GPUs (no counterpart in the reference, which is
 * single-process): contiguous particle ranges per rank (mclhip_shard_set), the
 * weight sum and the running sums in the reference's strictly sequential order
 * through the mclhip_shard_scan / _finalize exchange, poses and weights in the
 * 5-double rows of mclhip_shard_pack, feature slots in rows of
 * mclhip_fs_row_words() 64-bit words. */

/* The per-particle part of fast_slam1.py:246-404 on this rank's particles (device
 * generators are keyed by the global particle index): pose, measurements,
 * association and the raw weights of all of them.  The EKF updates, births and the
 * feature decay (:318-404) are applied to a particle when its row is packed
 * (mclhip_fs_shard_pack, mclhip_fs_shard_pack_drawn) or by mclhip_fs_get_state: a
 * particle that no position draws is replaced by the resampling and is never updated. */
int mclhip_fs_shard_kernel(mclhip_ctx* ctx, double v, double omega, double dt, const double alpha[6],
                           const double* truth_r, const double* truth_phi, const uint8_t* truth_valid, int L,
                           const int16_t* ints, const double* normals, uint64_t seed);
/* Last running sum of this rank after mclhip_shard_finalize (the last rank's value
 * is the sum of all weights, added one by one as fast_slam1.py:411-413 does). */
int mclhip_fs_shard_last_sum(mclhip_ctx* ctx, double* last_out);
/* 64-bit words per particle in a feature row: 2 + 9 * slots, slots = the capacity unless
 * mclhip_fs_set_row_slots has made the rows shorter. */
int64_t mclhip_fs_row_words(mclhip_ctx* ctx);
/* Largest feature count among this rank's particles; and the number of slots a feature row carries from now
 * on (1 .. capacity, the same on every rank: at least the largest count over all ranks, so that the rows move
 * the live features and not the unused capacity).  Packing a particle with more features than that fails. */
int mclhip_fs_max_features(mclhip_ctx* ctx, int32_t* max_out);
int mclhip_fs_set_row_slots(mclhip_ctx* ctx, int slots);
/* Feature rows of the local particles [lo[k], hi[k]) -> rows (host or device memory);
 * between mclhip_fs_shard_kernel and mclhip_fs_shard_unpack with that step's updates applied. */
int mclhip_fs_shard_pack(mclhip_ctx* ctx, const int64_t* lo, const int64_t* hi, int nslices, void* rows);
/* After mclhip_shard_pack_drawn: applies the step's updates to the particles it selected and
 * returns the largest feature count among them (what the ranks agree the row length on). */
int mclhip_fs_shard_apply_drawn(mclhip_ctx* ctx, int32_t* max_out);
/* Feature rows of the particles mclhip_shard_pack_drawn selected, in the same order. */
int mclhip_fs_shard_pack_drawn(mclhip_ctx* ctx, void* rows);
/* After mclhip_shard_unpack_drawn: every local particle takes the feature slots of its
 * ancestor from the L received rows (the row that call chose for it). */
int mclhip_fs_shard_unpack(mclhip_ctx* ctx, const void* rows, int64_t L);

/* The same step with the host out of the loop (no counterpart in the reference): every call only
 * enqueues on the context's stream (mclhip_set_stream) and exchanges data through DEVICE pointers
 * that RCCL collectives read and write; one device-to-host read per step (row counts, longest
 * feature list, overflow flags: all in one gathered block).  Order of a step:
 *   mclhip_fs_dshard_kernel           per-particle kernels; chunk sums of the raw weights -> d_chunks_out
 *   [all-gather chunks]
 *   mclhip_fs_dshard_scan(lasts NULL) local scan of the raw weights; header + records -> d_records_out
 *   [all-gather records]
 *   mclhip_fs_dshard_finalize_raw     exact running sums of the raw weights; (last sum, overflow flag) -> d_last_out[2]
 *   [all-gather lasts]                the last rank's last sum is sum_w of fast_slam1.py:411-413
 *   mclhip_fs_dshard_scan(lasts)      w /= max(1e-8, sum_w) (:415-418), local scan of the normalised weights
 *   [all-gather records]
 *   mclhip_dshard_finalize            running sums c_i (:421-428), which rows are drawn -> d_counts_out[world + 1]
 *   mclhip_fs_dshard_select           pose rows of the drawn particles -> d_rows (room for M_local + world rows of six
 *                                     doubles); the step's updates (:318-404) for exactly those; d_info_out[3] =
 *                                     longest feature list among them, capacity overflow, record overflow of round one
 *   [all-gather counts + info, read on the host; mclhip_fs_set_row_slots(longest list over all ranks)]
 *   mclhip_fs_dshard_pack             feature rows of the selected particles -> d_feature_rows
 *   [all-to-all of the pose rows, all-to-all of the feature rows]
 *   mclhip_fs_dshard_unpack           the local set from the received rows (max_features: the agreed longest list) */
int mclhip_fs_dshard_kernel(mclhip_ctx* ctx, double v, double omega, double dt, const double alpha[6],
                            const double* truth_r, const double* truth_phi, const uint8_t* truth_valid, int L,
                            const int16_t* ints, const double* normals, uint64_t seed, double* d_chunks_out);
int mclhip_fs_dshard_scan(mclhip_ctx* ctx, const double* d_all_chunks, const double* d_all_lasts, const int64_t* firsts,
                          int world, int rank, void* d_records_out, int cap);
int mclhip_fs_dshard_finalize_raw(mclhip_ctx* ctx, const void* d_all_records, const int64_t* firsts, int world, int rank,
                                  int cap, double* d_last_out);
int mclhip_fs_dshard_select(mclhip_ctx* ctx, int world, const int64_t* d_counts, const double* d_first_round, void* d_rows,
                            int64_t* d_info_out);
int mclhip_fs_dshard_pack(mclhip_ctx* ctx, int64_t n_rows, void* d_feature_rows);
int mclhip_fs_dshard_unpack(mclhip_ctx* ctx, const void* d_rows, const void* d_feature_rows, int64_t L, double r,
                            int max_features);

/* Measurement ------------------------------------------------------------ */

/* Per-kernel timing with HIP events on the context's own stream.
 * which: 0 = sector-query + likelihood kernel (where the query runs as two launches: the first, which serves the
 *            particles of the sector-interval path), 1 = motion kernel,
 *        2 = normalise + resample + gather kernels together,
 *        3 = grid prediction, 4 = grid update,
 *        5 = the sector query's launches after the first (the particles the first left over), only with on = 1;
 *        6 = no timing: *launches = how many particles the last two-launch sector query left to its entry-by-entry
 *            launch (the count the host last saw: written behind that launch, never waited for);
 *        7 = likewise, what the first launch left over: the same number, unless the interval kernel made a second
 *            pass over that list in between (evenly spaced beams; headings outside the range in which the
 *            reference's predicate is the circular one; the environment variable MCLHIP_EVERY_HEADING, read when a
 *            context is created, forces that pass with 1 and forbids it with 0).
 * on: 0 off, 1 every group, 2 the sector-query kernel only (two events per step instead of six: an event between
 * two kernels costs a few microseconds of idle GPU).  Enabling resets the accumulators. */
int mclhip_timing_enable(mclhip_ctx* ctx, int on);
/* Name of the sector-query kernel instantiation the last query of this context ran (where it ran as two launches: the
 * first), as a profiler prints it with the blanks left out -- "raycast_kernel<float,false,...>"; "" before the
 * first query.  Measurement only: lets a benchmark tie its figures to the profile of the very kernel it timed.  No
 * counterpart in the reference.  The string belongs to the context and is valid until its next query. */
const char* mclhip_query_kernel_name(mclhip_ctx* ctx);
int mclhip_timing_get(mclhip_ctx* ctx, int which, int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* MCLHIP_H */
