"""ctypes binding of libmclhip.so (the C ABI declared in include/mclhip.h).

The library is the only compute path of this package: if it cannot be loaded,
or no GPU is visible when a context is created, the call raises -- there is no
CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("MCLHIP_LIBNAME", "libmclhip.so"))

F32, F64 = 0, 1
W_PRODUCT, W_LOG = 0, 1
RESAMPLE_SERIAL, RESAMPLE_PARALLEL = 0, 1

_lib = None

_dbl_p = C.POINTER(C.c_double)
_i16_p = C.POINTER(C.c_int16)
_i32_p = C.POINTER(C.c_int32)
_i64_p = C.POINTER(C.c_int64)
_u8_p = C.POINTER(C.c_uint8)
_f32_p = C.POINTER(C.c_float)
_ctx_p = C.c_void_p

# name -> (restype, argtypes); every symbol include/mclhip.h declares
PROTOTYPES = {
    "mclhip_abi_version": (C.c_int, []),
    "mclhip_device_count": (C.c_int, []),
    "mclhip_create": (C.c_int, [C.POINTER(_ctx_p), C.c_int, C.c_int]),
    "mclhip_destroy": (None, [_ctx_p]),
    "mclhip_last_error": (C.c_char_p, [_ctx_p]),
    "mclhip_set_map": (C.c_int, [_ctx_p, _u8_p, C.c_int, C.c_int, C.c_double, C.c_double]),
    "mclhip_set_sensor": (C.c_int, [_ctx_p, _dbl_p, C.c_int, C.c_double, C.c_double, C.c_double, _dbl_p, C.c_double]),
    "mclhip_set_particles": (C.c_int, [_ctx_p, _dbl_p, _dbl_p, C.c_int64]),
    "mclhip_init_particles": (C.c_int, [_ctx_p, C.c_int64, _dbl_p, C.c_double, C.c_double, C.c_uint64]),
    "mclhip_get_particles": (C.c_int, [_ctx_p, _dbl_p, _dbl_p, _dbl_p]),
    "mclhip_num_particles": (C.c_int64, [_ctx_p]),
    "mclhip_raycast": (C.c_int, [_ctx_p, _dbl_p, C.c_int64, _dbl_p, _i32_p]),
    "mclhip_raycast_static": (C.c_int, [_ctx_p, _dbl_p, C.c_int64, _dbl_p, _i32_p]),
    "mclhip_motion": (C.c_int, [_ctx_p, _dbl_p, C.c_int64, C.c_double, C.c_double, C.c_double, _dbl_p, _i16_p,
                                _dbl_p, _dbl_p]),
    "mclhip_motion_book": (C.c_int, [_ctx_p, _dbl_p, C.c_int64, C.c_double, C.c_double, C.c_double, _dbl_p, _i16_p,
                                     _dbl_p]),
    "mclhip_compute_px": (C.c_int, [_ctx_p, _dbl_p, _dbl_p, C.c_int64, _dbl_p]),
    "mclhip_sample": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.c_double, _dbl_p, _i16_p, C.c_uint64, _dbl_p]),
    "mclhip_normalise": (C.c_int, [_ctx_p, C.c_int, _dbl_p]),
    "mclhip_resample": (C.c_int, [_ctx_p, C.c_double, C.c_int, _i64_p]),
    "mclhip_normalise_resample": (C.c_int, [_ctx_p, C.c_int, C.c_double, _dbl_p, _i64_p]),
    "mclhip_step": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.c_double, _dbl_p, _i16_p, C.c_uint64, _dbl_p,
                              C.c_int, C.c_double]),
    "mclhip_step_async": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.c_double, _dbl_p, C.c_uint64, _dbl_p, C.c_int,
                                    C.c_double]),
    "mclhip_sample_augmented": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.c_double, _dbl_p, _i16_p, C.c_uint64,
                                          _dbl_p, _dbl_p, _dbl_p]),
    "mclhip_resample_augmented": (C.c_int, [_ctx_p, C.c_double, _i64_p, _dbl_p, C.c_int64, _i64_p]),
    "mclhip_sample_augmented_device": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.c_double, _dbl_p, C.c_uint64, _dbl_p,
                                                 C.c_double, _dbl_p]),
    "mclhip_resample_augmented_device": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint64,
                                                   _i64_p]),
    "mclhip_sync": (C.c_int, [_ctx_p]),
    "mclhip_snapshot_particles": (C.c_int, [_ctx_p]),
    "mclhip_restore_particles_async": (C.c_int, [_ctx_p]),
    "mclhip_shard_set": (C.c_int, [_ctx_p, C.c_int64, C.c_int64]),
    "mclhip_shard_sample": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.c_double, _dbl_p, _i16_p, C.c_uint64, _dbl_p,
                                      C.c_int]),
    "mclhip_shard_max_logw": (C.c_int, [_ctx_p, _dbl_p]),
    "mclhip_shard_weights": (C.c_int, [_ctx_p, C.c_int, C.c_double, _dbl_p, C.c_int64]),
    "mclhip_shard_scan": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "mclhip_shard_get_records": (C.c_int, [_ctx_p, C.c_void_p, C.c_int64]),
    "mclhip_shard_finalize": (C.c_int, [_ctx_p, C.c_void_p, C.c_int64, C.c_uint64, C.c_uint32]),
    "mclhip_shard_count_below": (C.c_int, [_ctx_p, _dbl_p, C.c_int, _i64_p]),
    "mclhip_shard_pack": (C.c_int, [_ctx_p, _i64_p, _i64_p, C.c_int, C.c_void_p]),
    "mclhip_shard_unpack_resample": (C.c_int, [_ctx_p, C.c_void_p, C.c_int64, C.c_double, C.c_int64, _i64_p]),
    "mclhip_set_stream": (C.c_int, [_ctx_p, C.c_void_p]),
    "mclhip_dshard_sample_max": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.c_double, _dbl_p, _i16_p, C.c_uint64,
                                           _dbl_p, C.c_int, C.c_void_p]),
    "mclhip_dshard_weights": (C.c_int, [_ctx_p, C.c_void_p, C.c_void_p]),
    "mclhip_dshard_scan": (C.c_int, [_ctx_p, C.c_void_p, _i64_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "mclhip_dshard_finalize": (C.c_int, [_ctx_p, C.c_void_p, _i64_p, C.c_int, C.c_int, C.c_int, C.c_double,
                                         C.c_void_p]),
    "mclhip_dshard_pack": (C.c_int, [_ctx_p, C.c_int, C.c_void_p]),
    "mclhip_dshard_unpack_resample": (C.c_int, [_ctx_p, C.c_void_p, C.c_int64, C.c_double]),
    "mclhip_dshard_pack_fixed": (C.c_int, [_ctx_p, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mclhip_dshard_unpack_fixed": (C.c_int, [_ctx_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_void_p]),
    "mclhip_dshard_unpack_undo": (C.c_int, [_ctx_p]),
    "mclhip_get_indices": (C.c_int, [_ctx_p, _i64_p]),
    "mclhip_grid_set_map": (C.c_int, [_ctx_p, _u8_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]),
    "mclhip_grid_set_sensor": (C.c_int, [_ctx_p, _dbl_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                         C.c_double, _dbl_p, C.c_double]),
    "mclhip_grid_init_belief": (C.c_int, [_ctx_p]),
    "mclhip_grid_set_belief": (C.c_int, [_ctx_p, _f32_p]),
    "mclhip_grid_set_prediction": (C.c_int, [_ctx_p, _dbl_p]),
    "mclhip_grid_get_belief": (C.c_int, [_ctx_p, _f32_p, _dbl_p]),
    "mclhip_grid_predict": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.c_double, _dbl_p]),
    "mclhip_grid_update": (C.c_int, [_ctx_p, _dbl_p, _f32_p]),
    "mclhip_fs_init": (C.c_int, [_ctx_p, C.c_int64, C.c_int]),
    "mclhip_fs_set_model": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "mclhip_fs_step": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.c_double, _dbl_p, _dbl_p, _dbl_p, _u8_p, C.c_int,
                                 _i16_p, _dbl_p, C.c_uint64, C.c_double, _i64_p]),
    "mclhip_fs_get_state": (C.c_int, [_ctx_p, _dbl_p, _dbl_p, _i32_p, _i32_p, _dbl_p, _dbl_p, _i32_p, _dbl_p, _i32_p]),
    "mclhip_fs_shard_kernel": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.c_double, _dbl_p, _dbl_p, _dbl_p, _u8_p,
                                         C.c_int, _i16_p, _dbl_p, C.c_uint64]),
    "mclhip_fs_shard_last_sum": (C.c_int, [_ctx_p, _dbl_p]),
    "mclhip_fs_row_words": (C.c_int64, [_ctx_p]),
    "mclhip_fs_max_features": (C.c_int, [_ctx_p, C.POINTER(C.c_int32)]),
    "mclhip_fs_set_row_slots": (C.c_int, [_ctx_p, C.c_int]),
    "mclhip_fs_shard_pack": (C.c_int, [_ctx_p, _i64_p, _i64_p, C.c_int, C.c_void_p]),
    "mclhip_fs_shard_apply_drawn": (C.c_int, [_ctx_p, C.POINTER(C.c_int32)]),
    "mclhip_fs_dshard_kernel": (C.c_int, [_ctx_p, C.c_double, C.c_double, C.c_double, _dbl_p, _dbl_p, _dbl_p, _u8_p,
                                          C.c_int, _i16_p, _dbl_p, C.c_uint64, C.c_void_p]),
    "mclhip_fs_dshard_scan": (C.c_int, [_ctx_p, C.c_void_p, C.c_void_p, _i64_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "mclhip_fs_dshard_finalize_raw": (C.c_int, [_ctx_p, C.c_void_p, _i64_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "mclhip_fs_dshard_select": (C.c_int, [_ctx_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mclhip_fs_dshard_pack": (C.c_int, [_ctx_p, C.c_int64, C.c_void_p]),
    "mclhip_fs_dshard_unpack": (C.c_int, [_ctx_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_int]),
    "mclhip_fs_shard_pack_drawn": (C.c_int, [_ctx_p, C.c_void_p]),
    "mclhip_fs_shard_unpack": (C.c_int, [_ctx_p, C.c_void_p, C.c_int64]),
    "mclhip_shard_plan_drawn": (C.c_int, [_ctx_p, _i64_p, C.c_int, C.c_int, C.c_double, _i64_p]),
    "mclhip_shard_pack_drawn": (C.c_int, [_ctx_p, C.c_int, C.c_void_p]),
    "mclhip_shard_unpack_drawn": (C.c_int, [_ctx_p, C.c_void_p, C.c_int64, C.c_double, _i64_p]),
    "mclhip_timing_enable": (C.c_int, [_ctx_p, C.c_int]),
    "mclhip_timing_get": (C.c_int, [_ctx_p, C.c_int, _i64_p, _dbl_p]),
    "mclhip_query_kernel_name": (C.c_char_p, [_ctx_p]),
}


class MclHipError(RuntimeError):
    pass


def load():
    """Load libmclhip.so and bind every prototype.  Raises if the file or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MclHipError("libmclhip.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C slam_amd/csrc`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError names the missing symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError("expected shape %r, got %r" % (shape, a.shape))
    return a


class Context:
    """One GPU context (wraps mclhip_ctx*)."""

    def __init__(self, device=0, dtype=F64):
        self.lib = load()
        h = _ctx_p()
        rc = self.lib.mclhip_create(C.byref(h), int(device), int(dtype))
        if rc != 0:
            raise MclHipError("mclhip_create failed (%d): %s" % (rc, self.lib.mclhip_last_error(None).decode()))
        self.h = h
        self.dtype = dtype
        self.B = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.mclhip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise MclHipError("%s failed (%d): %s" % (what, rc, self.lib.mclhip_last_error(self.h).decode()))

    # map / sensor ---------------------------------------------------------
    def set_map(self, grid, res_x, res_y):
        g = np.ascontiguousarray(grid, dtype=np.uint8)
        if g.ndim != 2:
            raise ValueError("map must be 2-D uint8")
        self._ck(self.lib.mclhip_set_map(self.h, _p(g, _u8_p), g.shape[0], g.shape[1], float(res_x), float(res_y)),
                 "mclhip_set_map")

    def set_sensor(self, angles_deg, z_max, sigma_hit, lambda_short, w4, half_angle):
        a = _f64(angles_deg)
        w = _f64(w4, (4,))
        self._ck(self.lib.mclhip_set_sensor(self.h, _p(a, _dbl_p), len(a), float(z_max), float(sigma_hit),
                                            float(lambda_short), _p(w, _dbl_p), float(half_angle)),
                 "mclhip_set_sensor")
        self.B = len(a)

    # particles ------------------------------------------------------------
    def set_particles(self, X, w=None):
        X = _f64(X)
        if X.ndim != 2 or X.shape[1] != 3:
            raise ValueError("X must be [M,3]")
        wv = None if w is None else _f64(w, (X.shape[0],))
        self._ck(self.lib.mclhip_set_particles(self.h, _p(X, _dbl_p), _p(wv, _dbl_p), X.shape[0]),
                 "mclhip_set_particles")

    def num_particles(self):
        return int(self.lib.mclhip_num_particles(self.h))

    def get_particles(self, want_X=True, want_w=True, want_logw=False):
        M = self.num_particles()
        X = np.empty((M, 3)) if want_X else None
        w = np.empty(M) if want_w else None
        lw = np.empty(M) if want_logw else None
        self._ck(self.lib.mclhip_get_particles(self.h, _p(X, _dbl_p), _p(w, _dbl_p), _p(lw, _dbl_p)),
                 "mclhip_get_particles")
        return X, w, lw

    # stages -----------------------------------------------------------------
    def raycast(self, poses, want_hits=True, static=False):
        """static: through mclhip_raycast_static (no device allocation; at most 4096 poses / 65536 pose-beam pairs)."""
        P = _f64(poses).reshape(-1, 3)
        n = len(P)
        R = np.empty((n, self.B))
        Hc = np.empty((n, self.B), np.int32) if want_hits else None
        if static:
            self._ck(self.lib.mclhip_raycast_static(self.h, _p(P, _dbl_p), n, _p(R, _dbl_p), _p(Hc, _i32_p)),
                     "mclhip_raycast_static")
        else:
            self._ck(self.lib.mclhip_raycast(self.h, _p(P, _dbl_p), n, _p(R, _dbl_p), _p(Hc, _i32_p)), "mclhip_raycast")
        return R, Hc

    def motion(self, poses, v, omega, dt, alpha, noise_ints=None):
        P = _f64(poses).reshape(-1, 3)
        n = len(P)
        al = _f64(alpha, (6,))
        nz = None
        if noise_ints is not None:
            nz = np.ascontiguousarray(noise_ints, dtype=np.int16)
            if nz.shape != (n, 3, 12):
                raise ValueError("noise_ints must be [n,3,12]")
        out = np.empty((n, 3))
        pm = np.empty(n)
        self._ck(self.lib.mclhip_motion(self.h, _p(P, _dbl_p), n, float(v), float(omega), float(dt), _p(al, _dbl_p),
                                        _p(nz, _i16_p), _p(out, _dbl_p), _p(pm, _dbl_p)), "mclhip_motion")
        return out, pm

    def motion_book(self, poses, v, omega, dt, alpha, noise_ints=None):
        """Table 5.3 without the straight-line branch and without a map (mclhip_motion_book)."""
        P = _f64(poses).reshape(-1, 3)
        n = len(P)
        al = _f64(alpha, (6,))
        nz = None
        if noise_ints is not None:
            nz = np.ascontiguousarray(noise_ints, dtype=np.int16)
            if nz.shape != (n, 3, 12):
                raise ValueError("noise_ints must be [n,3,12]")
        out = np.empty((n, 3))
        self._ck(self.lib.mclhip_motion_book(self.h, _p(P, _dbl_p), n, float(v), float(omega), float(dt),
                                             _p(al, _dbl_p), _p(nz, _i16_p), _p(out, _dbl_p)), "mclhip_motion_book")
        return out

    def compute_px(self, z, z_true):
        z, zt = np.broadcast_arrays(np.asarray(z, np.float64), np.asarray(z_true, np.float64))
        shape = z.shape
        z = np.ascontiguousarray(z).ravel()
        zt = np.ascontiguousarray(zt).ravel()
        out = np.empty(len(z))
        self._ck(self.lib.mclhip_compute_px(self.h, _p(z, _dbl_p), _p(zt, _dbl_p), len(z), _p(out, _dbl_p)),
                 "mclhip_compute_px")
        return out.reshape(shape)

    def _noise(self, noise_ints):
        if noise_ints is None:
            return None
        nz = np.ascontiguousarray(noise_ints, dtype=np.int16)
        if nz.shape != (self.num_particles(), 3, 12):
            raise ValueError("noise_ints must be [M,3,12]")
        return nz

    def sample(self, v, omega, dt, alpha, z_meas, noise_ints=None, seed=0):
        al = _f64(alpha, (6,))
        zm = _f64(z_meas, (self.B,))
        nz = self._noise(noise_ints)
        self._ck(self.lib.mclhip_sample(self.h, float(v), float(omega), float(dt), _p(al, _dbl_p), _p(nz, _i16_p),
                                        int(seed), _p(zm, _dbl_p)), "mclhip_sample")

    def sample_augmented(self, v, omega, dt, alpha, z_meas, noise_ints=None, range_noise=None, seed=0):
        """-> w_avg, the strictly sequential sum of w[m]/M."""
        al = _f64(alpha, (6,))
        zm = _f64(z_meas, (self.B,))
        nz = self._noise(noise_ints)
        rn = None if range_noise is None else _f64(range_noise, (self.num_particles(), self.B))
        wavg = C.c_double(0.0)
        self._ck(self.lib.mclhip_sample_augmented(self.h, float(v), float(omega), float(dt), _p(al, _dbl_p),
                                                  _p(nz, _i16_p), int(seed), _p(zm, _dbl_p), _p(rn, _dbl_p),
                                                  C.byref(wavg)), "mclhip_sample_augmented")
        return wavg.value

    def resample_augmented(self, r, inj_rows, inj_poses, want_idx=True):
        rows = np.ascontiguousarray(inj_rows, dtype=np.int64).ravel()
        poses = _f64(inj_poses).reshape(-1, 3)
        if len(rows) != len(poses):
            raise ValueError("one pose per injected row")
        idx = np.empty(self.num_particles(), np.int64) if want_idx else None
        self._ck(self.lib.mclhip_resample_augmented(self.h, float(r), _p(rows, _i64_p) if len(rows) else None,
                                                    _p(poses, _dbl_p) if len(rows) else None, len(rows),
                                                    _p(idx, _i64_p)), "mclhip_resample_augmented")
        return idx

    def sample_augmented_device(self, v, omega, dt, alpha, z_meas, seed, range_sigma=0.15):
        """Augmented-MCL sample with motion integers and range noise drawn on the device; returns w_avg."""
        al = _f64(alpha, (6,))
        zm = _f64(z_meas, (self.B,))
        w_avg = C.c_double(0.0)
        self._ck(self.lib.mclhip_sample_augmented_device(self.h, float(v), float(omega), float(dt), _p(al, _dbl_p),
                                                         int(seed), _p(zm, _dbl_p), float(range_sigma), C.byref(w_avg)),
                 "mclhip_sample_augmented_device")
        return w_avg.value

    def resample_augmented_device(self, r, p_random, extent_x, extent_y, seed):
        """Augmented-MCL resampling with the injections drawn on the device; returns their number."""
        n = C.c_int64(0)
        self._ck(self.lib.mclhip_resample_augmented_device(self.h, float(r), float(p_random), float(extent_x),
                                                           float(extent_y), int(seed), C.byref(n)),
                 "mclhip_resample_augmented_device")
        return n.value

    def normalise(self, mode=W_PRODUCT):
        s = C.c_double(0.0)
        self._ck(self.lib.mclhip_normalise(self.h, int(mode), C.byref(s)), "mclhip_normalise")
        return s.value

    def resample(self, r, algo=RESAMPLE_PARALLEL, want_idx=True):
        idx = np.empty(self.num_particles(), np.int64) if want_idx else None
        self._ck(self.lib.mclhip_resample(self.h, float(r), int(algo), _p(idx, _i64_p)), "mclhip_resample")
        return idx

    def normalise_resample(self, r, mode=W_PRODUCT, want_idx=True):
        """mclhip_normalise + mclhip_resample as the fused update runs them; returns (sum of the weights, indices)."""
        s = C.c_double(0.0)
        idx = np.empty(self.num_particles(), np.int64) if want_idx else None
        self._ck(self.lib.mclhip_normalise_resample(self.h, int(mode), float(r), C.byref(s), _p(idx, _i64_p)),
                 "mclhip_normalise_resample")
        return s.value, idx

    def step(self, v, omega, dt, alpha, z_meas, r, noise_ints=None, seed=0, weight_mode=W_PRODUCT, wait=True):
        al = _f64(alpha, (6,))
        zm = _f64(z_meas, (self.B,))
        if wait:
            nz = self._noise(noise_ints)
            self._ck(self.lib.mclhip_step(self.h, float(v), float(omega), float(dt), _p(al, _dbl_p), _p(nz, _i16_p),
                                          int(seed), _p(zm, _dbl_p), int(weight_mode), float(r)), "mclhip_step")
        else:
            if noise_ints is not None:
                raise ValueError("the asynchronous step draws its noise on the device")
            self._ck(self.lib.mclhip_step_async(self.h, float(v), float(omega), float(dt), _p(al, _dbl_p), int(seed),
                                                _p(zm, _dbl_p), int(weight_mode), float(r)), "mclhip_step_async")

    def sync(self):
        self._ck(self.lib.mclhip_sync(self.h), "mclhip_sync")

    def snapshot_particles(self):
        self._ck(self.lib.mclhip_snapshot_particles(self.h), "mclhip_snapshot_particles")

    def restore_particles_async(self):
        self._ck(self.lib.mclhip_restore_particles_async(self.h), "mclhip_restore_particles_async")

    # one filter over several GPUs (per-rank halves; see slam_amd/sharded.py) -------------
    RECORD_DTYPE = np.dtype([("i", "<i8"), ("w", "<f8"), ("P", "<u8")])

    def shard_set(self, first_index, m_total):
        self._ck(self.lib.mclhip_shard_set(self.h, int(first_index), int(m_total)), "mclhip_shard_set")

    def shard_sample(self, v, omega, dt, alpha, z_meas, noise_ints, seed, weight_mode):
        al = _f64(alpha, (6,))
        zm = _f64(z_meas, (self.B,))
        nz = self._noise(noise_ints)
        self._ck(self.lib.mclhip_shard_sample(self.h, float(v), float(omega), float(dt), _p(al, _dbl_p),
                                              _p(nz, _i16_p), int(seed), _p(zm, _dbl_p), int(weight_mode)),
                 "mclhip_shard_sample")

    def shard_max_logw(self):
        m = C.c_double(0.0)
        self._ck(self.lib.mclhip_shard_max_logw(self.h, C.byref(m)), "mclhip_shard_max_logw")
        return m.value

    def shard_weights(self, mode, gmax):
        n = (self.num_particles() + 8191) // 8192
        out = np.empty(n)
        self._ck(self.lib.mclhip_shard_weights(self.h, int(mode), float(gmax), _p(out, _dbl_p), n),
                 "mclhip_shard_weights")
        return out

    def shard_scan(self, total, approx_offset):
        k = C.c_uint64(0)
        n = C.c_uint32(0)
        self._ck(self.lib.mclhip_shard_scan(self.h, float(total), float(approx_offset), C.byref(k), C.byref(n)),
                 "mclhip_shard_scan")
        return int(k.value), int(n.value)

    def shard_get_records(self, n):
        out = np.zeros(n, self.RECORD_DTYPE)
        self._ck(self.lib.mclhip_shard_get_records(self.h, out.ctypes.data_as(C.c_void_p), n),
                 "mclhip_shard_get_records")
        return out

    def shard_finalize(self, all_records, p_off, f_off):
        rec = np.ascontiguousarray(all_records, dtype=self.RECORD_DTYPE)
        self._ck(self.lib.mclhip_shard_finalize(self.h, rec.ctypes.data_as(C.c_void_p), len(rec),
                                                int(p_off) & 0xFFFFFFFFFFFFFFFF, int(f_off)), "mclhip_shard_finalize")

    def shard_count_below(self, U):
        U = _f64(U)
        out = np.empty(len(U), np.int64)
        self._ck(self.lib.mclhip_shard_count_below(self.h, _p(U, _dbl_p), len(U), _p(out, _i64_p)),
                 "mclhip_shard_count_below")
        return out

    def shard_pack(self, lo, hi, rows_ptr):
        lo = np.ascontiguousarray(lo, dtype=np.int64)
        hi = np.ascontiguousarray(hi, dtype=np.int64)
        self._ck(self.lib.mclhip_shard_pack(self.h, _p(lo, _i64_p), _p(hi, _i64_p), len(lo), C.c_void_p(rows_ptr)),
                 "mclhip_shard_pack")

    def shard_unpack_resample(self, rows_ptr, L, r, row0_global, want_idx=False):
        idx = np.empty(self.num_particles(), np.int64) if want_idx else None
        self._ck(self.lib.mclhip_shard_unpack_resample(self.h, C.c_void_p(rows_ptr), int(L), float(r),
                                                       int(row0_global), _p(idx, _i64_p)),
                 "mclhip_shard_unpack_resample")
        return idx

    # device-driven multi-GPU step (device pointers in, nothing waits) ---------
    def set_stream(self, hip_stream):
        self._ck(self.lib.mclhip_set_stream(self.h, C.c_void_p(hip_stream or 0)), "mclhip_set_stream")

    def dshard_sample_max(self, v, omega, dt, alpha, z_meas, noise_ints, seed, mode, d_max):
        al = _f64(alpha, (6,))
        z = _f64(z_meas)
        if noise_ints is not None:
            noise_ints = np.ascontiguousarray(noise_ints, dtype=np.int16)
        self._ck(self.lib.mclhip_dshard_sample_max(self.h, float(v), float(omega), float(dt), _p(al, _dbl_p),
                                                   _p(noise_ints, _i16_p), int(seed), _p(z, _dbl_p), int(mode),
                                                   C.c_void_p(d_max)), "mclhip_dshard_sample_max")

    def dshard_weights(self, d_gmax, d_chunks):
        self._ck(self.lib.mclhip_dshard_weights(self.h, C.c_void_p(d_gmax), C.c_void_p(d_chunks)), "mclhip_dshard_weights")

    def dshard_scan(self, d_all_chunks, firsts, rank, d_records, cap):
        f = np.ascontiguousarray(firsts, np.int64)
        self._ck(self.lib.mclhip_dshard_scan(self.h, C.c_void_p(d_all_chunks), _p(f, _i64_p), len(f) - 1, int(rank),
                                             C.c_void_p(d_records), int(cap)), "mclhip_dshard_scan")

    def dshard_finalize(self, d_all_records, firsts, rank, cap, r, d_counts):
        f = np.ascontiguousarray(firsts, np.int64)
        self._ck(self.lib.mclhip_dshard_finalize(self.h, C.c_void_p(d_all_records), _p(f, _i64_p), len(f) - 1, int(rank),
                                                 int(cap), float(r), C.c_void_p(d_counts)), "mclhip_dshard_finalize")

    def dshard_pack(self, world, d_rows):
        self._ck(self.lib.mclhip_dshard_pack(self.h, int(world), C.c_void_p(d_rows)), "mclhip_dshard_pack")

    def dshard_unpack_resample(self, d_rows, L, r):
        self._ck(self.lib.mclhip_dshard_unpack_resample(self.h, C.c_void_p(d_rows), int(L), float(r)),
                 "mclhip_dshard_unpack_resample")

    def dshard_pack_fixed(self, world, rank, cap_rows, d_blocks, d_all_counts, d_flag, h_late=None):
        self._ck(self.lib.mclhip_dshard_pack_fixed(self.h, int(world), int(rank), int(cap_rows), C.c_void_p(d_blocks),
                                                   C.c_void_p(d_all_counts), C.c_void_p(d_flag), C.c_void_p(h_late)),
                 "mclhip_dshard_pack_fixed")

    def dshard_unpack_fixed(self, d_blocks, d_all_counts, world, rank, cap_rows, r, d_flag):
        self._ck(self.lib.mclhip_dshard_unpack_fixed(self.h, C.c_void_p(d_blocks), C.c_void_p(d_all_counts), int(world),
                                                     int(rank), int(cap_rows), float(r), C.c_void_p(d_flag)),
                 "mclhip_dshard_unpack_fixed")

    def dshard_unpack_undo(self):
        self._ck(self.lib.mclhip_dshard_unpack_undo(self.h), "mclhip_dshard_unpack_undo")

    def get_indices(self):
        idx = np.empty(self.num_particles(), np.int64)
        self._ck(self.lib.mclhip_get_indices(self.h, _p(idx, _i64_p)), "mclhip_get_indices")
        return idx

    def init_particles(self, M, x0, extent_x, extent_y, seed=0):
        x = None if x0 is None else _f64(np.asarray(x0, np.float64).ravel(), (3,))
        self._ck(self.lib.mclhip_init_particles(self.h, int(M), _p(x, _dbl_p), float(extent_x), float(extent_y),
                                                int(seed)), "mclhip_init_particles")

    # grid (histogram) localisation ------------------------------------------
    def grid_set_map(self, grid, res_x, res_y, obstacle_below=200):
        g = np.ascontiguousarray(grid, dtype=np.uint8)
        if g.ndim != 2:
            raise ValueError("map must be 2-D")
        self._ck(self.lib.mclhip_grid_set_map(self.h, _p(g, _u8_p), g.shape[0], g.shape[1], float(res_x), float(res_y),
                                              int(obstacle_below)), "mclhip_grid_set_map")
        self.grid_shape = g.shape

    def grid_set_sensor(self, angles_deg, n_dir, res_theta_deg, z_max, sigma_hit, lambda_short, w4, half_angle):
        a = _f64(angles_deg)
        w = _f64(w4, (4,))
        self._ck(self.lib.mclhip_grid_set_sensor(self.h, _p(a, _dbl_p), len(a), int(n_dir), float(res_theta_deg),
                                                 float(z_max), float(sigma_hit), float(lambda_short), _p(w, _dbl_p),
                                                 float(half_angle)), "mclhip_grid_set_sensor")
        self.grid_dirs = int(n_dir)
        self.grid_beams = len(a)

    def _grid_dims(self):
        return (self.grid_shape[0], self.grid_shape[1], self.grid_dirs)

    def grid_init_belief(self):
        self._ck(self.lib.mclhip_grid_init_belief(self.h), "mclhip_grid_init_belief")

    def grid_set_belief(self, pk):
        a = np.ascontiguousarray(pk, dtype=np.float32)
        if a.shape != self._grid_dims():
            raise ValueError("belief must have shape %r" % (self._grid_dims(),))
        self._ck(self.lib.mclhip_grid_set_belief(self.h, _p(a, _f32_p)), "mclhip_grid_set_belief")

    def grid_set_prediction(self, pk_pred):
        a = _f64(pk_pred, self._grid_dims())
        self._ck(self.lib.mclhip_grid_set_prediction(self.h, _p(a, _dbl_p)), "mclhip_grid_set_prediction")

    def grid_get_belief(self, want_pk=True, want_pred=False):
        pk = np.empty(self._grid_dims(), np.float32) if want_pk else None
        pp = np.empty(self._grid_dims(), np.float64) if want_pred else None
        self._ck(self.lib.mclhip_grid_get_belief(self.h, _p(pk, _f32_p), _p(pp, _dbl_p)), "mclhip_grid_get_belief")
        return pk, pp

    def grid_predict(self, v, omega, dt, alpha):
        al = _f64(alpha, (6,))
        self._ck(self.lib.mclhip_grid_predict(self.h, float(v), float(omega), float(dt), _p(al, _dbl_p)),
                 "mclhip_grid_predict")

    def grid_update(self, z_meas):
        z = _f64(z_meas, (self.grid_beams,))
        s = C.c_float(0.0)
        self._ck(self.lib.mclhip_grid_update(self.h, _p(z, _dbl_p), C.byref(s)), "mclhip_grid_update")
        return np.float32(s.value)

    # FastSLAM 1.0 ------------------------------------------------------------
    def fs_init(self, M, capacity):
        self._ck(self.lib.mclhip_fs_init(self.h, int(M), int(capacity)), "mclhip_fs_init")
        self.fs_M, self.fs_cap = int(M), int(capacity)

    def fs_set_model(self, std_r, std_phi, r_max, p0, thres_sig):
        self._ck(self.lib.mclhip_fs_set_model(self.h, float(std_r), float(std_phi), float(r_max), float(p0),
                                              float(thres_sig)), "mclhip_fs_set_model")

    def fs_step(self, v, omega, dt, alpha, truth_r, truth_phi, truth_valid, ints, normals, r, seed=0, want_idx=False):
        al = _f64(alpha, (6,))
        tr_, tp = _f64(truth_r), _f64(truth_phi)
        tv = np.ascontiguousarray(truth_valid, dtype=np.uint8)
        L = len(tr_)
        if tp.shape != (L,) or tv.shape != (L,):
            raise ValueError("truth arrays must have one entry per landmark")
        nv = int(np.count_nonzero(tv))
        if ints is not None:
            ints = np.ascontiguousarray(ints, dtype=np.int16)
            if ints.shape != (self.fs_M, 3, 12):
                raise ValueError("ints must have shape (M, 3, 12)")
        if normals is not None:
            normals = _f64(normals, (self.fs_M, 2 * nv))
        idx = np.empty(self.fs_M, np.int64) if want_idx else None
        self._ck(self.lib.mclhip_fs_step(self.h, float(v), float(omega), float(dt), _p(al, _dbl_p), _p(tr_, _dbl_p),
                                         _p(tp, _dbl_p), _p(tv, _u8_p), L, _p(ints, _i16_p), _p(normals, _dbl_p),
                                         int(seed), float(r), _p(idx, _i64_p)), "mclhip_fs_step")
        return idx

    def fs_shard_kernel(self, v, omega, dt, alpha, truth_r, truth_phi, truth_valid, ints, normals, seed=0):
        al = _f64(alpha, (6,))
        tr_, tp = _f64(truth_r), _f64(truth_phi)
        tv = np.ascontiguousarray(truth_valid, dtype=np.uint8)
        nv = int(np.count_nonzero(tv))
        if ints is not None:
            ints = np.ascontiguousarray(ints, dtype=np.int16)
            if ints.shape != (self.fs_M, 3, 12):
                raise ValueError("ints must have shape (M, 3, 12)")
        if normals is not None:
            normals = _f64(normals, (self.fs_M, 2 * nv))
        self._ck(self.lib.mclhip_fs_shard_kernel(self.h, float(v), float(omega), float(dt), _p(al, _dbl_p),
                                                 _p(tr_, _dbl_p), _p(tp, _dbl_p), _p(tv, _u8_p), len(tr_),
                                                 _p(ints, _i16_p), _p(normals, _dbl_p), int(seed)), "mclhip_fs_shard_kernel")

    def fs_dshard_kernel(self, v, omega, dt, alpha, truth_r, truth_phi, truth_valid, ints, normals, seed, d_chunks):
        """The per-particle kernels of one step, enqueued only; chunk sums of the raw weights -> device pointer."""
        al = _f64(alpha, (6,))
        tr_, tp = _f64(truth_r), _f64(truth_phi)
        tv = np.ascontiguousarray(truth_valid, dtype=np.uint8)
        nv = int(np.count_nonzero(tv))
        if ints is not None:
            ints = np.ascontiguousarray(ints, dtype=np.int16)
            if ints.shape != (self.fs_M, 3, 12):
                raise ValueError("ints must have shape (M, 3, 12)")
        if normals is not None:
            normals = _f64(normals, (self.fs_M, 2 * nv))
        self._fs_inputs = (ints, normals)         # read by copies that may still be in flight when this returns
        self._ck(self.lib.mclhip_fs_dshard_kernel(self.h, float(v), float(omega), float(dt), _p(al, _dbl_p),
                                                  _p(tr_, _dbl_p), _p(tp, _dbl_p), _p(tv, _u8_p), len(tr_),
                                                  _p(ints, _i16_p), _p(normals, _dbl_p), int(seed), C.c_void_p(d_chunks)),
                 "mclhip_fs_dshard_kernel")

    def fs_dshard_scan(self, d_all_chunks, d_all_lasts, firsts, rank, d_records, cap):
        f = np.ascontiguousarray(firsts, np.int64)
        self._ck(self.lib.mclhip_fs_dshard_scan(self.h, C.c_void_p(d_all_chunks), C.c_void_p(d_all_lasts), _p(f, _i64_p),
                                                len(f) - 1, int(rank), C.c_void_p(d_records), int(cap)), "mclhip_fs_dshard_scan")

    def fs_dshard_finalize_raw(self, d_all_records, firsts, rank, cap, d_last):
        f = np.ascontiguousarray(firsts, np.int64)
        self._ck(self.lib.mclhip_fs_dshard_finalize_raw(self.h, C.c_void_p(d_all_records), _p(f, _i64_p), len(f) - 1, int(rank),
                                                        int(cap), C.c_void_p(d_last)), "mclhip_fs_dshard_finalize_raw")

    def fs_dshard_select(self, world, d_counts, d_first_round, d_rows, d_info):
        self._ck(self.lib.mclhip_fs_dshard_select(self.h, int(world), C.c_void_p(d_counts), C.c_void_p(d_first_round),
                                                  C.c_void_p(d_rows), C.c_void_p(d_info)), "mclhip_fs_dshard_select")

    def fs_dshard_pack(self, n_rows, d_feature_rows):
        self._ck(self.lib.mclhip_fs_dshard_pack(self.h, int(n_rows), C.c_void_p(d_feature_rows)), "mclhip_fs_dshard_pack")

    def fs_dshard_unpack(self, d_rows, d_feature_rows, L, r, max_features):
        self._ck(self.lib.mclhip_fs_dshard_unpack(self.h, C.c_void_p(d_rows), C.c_void_p(d_feature_rows), int(L), float(r),
                                                  int(max_features)), "mclhip_fs_dshard_unpack")

    def fs_shard_last_sum(self):
        v = C.c_double(0.0)
        self._ck(self.lib.mclhip_fs_shard_last_sum(self.h, C.byref(v)), "mclhip_fs_shard_last_sum")
        return v.value

    def fs_row_words(self):
        return int(self.lib.mclhip_fs_row_words(self.h))

    def fs_max_features(self):
        n = C.c_int32(0)
        self._ck(self.lib.mclhip_fs_max_features(self.h, C.byref(n)), "mclhip_fs_max_features")
        return n.value

    def fs_set_row_slots(self, slots):
        self._ck(self.lib.mclhip_fs_set_row_slots(self.h, int(slots)), "mclhip_fs_set_row_slots")

    def fs_shard_pack(self, lo, hi, rows_ptr):
        lo = np.ascontiguousarray(lo, np.int64)
        hi = np.ascontiguousarray(hi, np.int64)
        self._ck(self.lib.mclhip_fs_shard_pack(self.h, _p(lo, _i64_p), _p(hi, _i64_p), len(lo), C.c_void_p(rows_ptr)),
                 "mclhip_fs_shard_pack")

    def fs_shard_apply_drawn(self):
        """Applies the open step's updates to the rows `shard_pack_drawn` selected; largest feature count among them."""
        v = C.c_int32()
        self._ck(self.lib.mclhip_fs_shard_apply_drawn(self.h, C.byref(v)), "mclhip_fs_shard_apply_drawn")
        return int(v.value)

    def fs_shard_pack_drawn(self, rows_ptr):
        self._ck(self.lib.mclhip_fs_shard_pack_drawn(self.h, C.c_void_p(rows_ptr)), "mclhip_fs_shard_pack_drawn")

    def fs_shard_unpack(self, rows_ptr, L):
        self._ck(self.lib.mclhip_fs_shard_unpack(self.h, C.c_void_p(rows_ptr), int(L)), "mclhip_fs_shard_unpack")

    def shard_plan_drawn(self, firsts, rank, r):
        f = np.ascontiguousarray(firsts, np.int64)
        out = np.zeros(len(f) - 1, np.int64)
        self._ck(self.lib.mclhip_shard_plan_drawn(self.h, _p(f, _i64_p), len(f) - 1, int(rank), float(r), _p(out, _i64_p)),
                 "mclhip_shard_plan_drawn")
        return out

    def shard_pack_drawn(self, world, rows_ptr):
        self._ck(self.lib.mclhip_shard_pack_drawn(self.h, int(world), C.c_void_p(rows_ptr)), "mclhip_shard_pack_drawn")

    def shard_unpack_drawn(self, rows_ptr, L, r, want_idx=False):
        idx = np.empty(self.num_particles(), np.int64) if want_idx else None
        self._ck(self.lib.mclhip_shard_unpack_drawn(self.h, C.c_void_p(rows_ptr), int(L), float(r), _p(idx, _i64_p)),
                 "mclhip_shard_unpack_drawn")
        return idx

    def fs_feature_counts(self):
        """Features per particle and the next feature id of every particle (no feature data is copied)."""
        M = self.fs_M
        nfeat, new_id = np.empty(M, np.int32), np.empty(M, np.int32)
        self._ck(self.lib.mclhip_fs_get_state(self.h, None, None, _p(nfeat, _i32_p), _p(new_id, _i32_p), None, None, None,
                                              None, None), "mclhip_fs_get_state")
        return nfeat, new_id

    def fs_get_state(self):
        M, F = self.fs_M, self.fs_cap
        out = {"pose": np.empty((M, 3)), "w": np.empty(M), "nfeat": np.empty(M, np.int32), "new_id": np.empty(M, np.int32),
               "f_m": np.empty((M, F, 2)), "f_cov": np.empty((M, F, 2, 2)), "f_i": np.empty((M, F), np.int32),
               "f_w": np.empty((M, F)), "f_id": np.empty((M, F), np.int32)}
        self._ck(self.lib.mclhip_fs_get_state(self.h, _p(out["pose"], _dbl_p), _p(out["w"], _dbl_p),
                                              _p(out["nfeat"], _i32_p), _p(out["new_id"], _i32_p),
                                              _p(out["f_m"], _dbl_p), _p(out["f_cov"], _dbl_p), _p(out["f_i"], _i32_p),
                                              _p(out["f_w"], _dbl_p), _p(out["f_id"], _i32_p)), "mclhip_fs_get_state")
        return out

    def timing_enable(self, on=True):
        """on: False / True (every kernel group) / 2 (the sector-query kernel only)."""
        self._ck(self.lib.mclhip_timing_enable(self.h, 2 if on == 2 else (1 if on else 0)), "mclhip_timing_enable")

    def query_kernel_name(self):
        """The sector-query instantiation the last query ran, e.g. 'raycast_kernel<float,false,...>'."""
        return (self.lib.mclhip_query_kernel_name(self.h) or b"").decode()

    def timing_get(self, which):
        n = C.c_int64(0)
        ms = C.c_double(0.0)
        self._ck(self.lib.mclhip_timing_get(self.h, int(which), C.byref(n), C.byref(ms)), "mclhip_timing_get")
        return n.value, ms.value
