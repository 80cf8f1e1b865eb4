"""Debug helper: largest relative error of the float32-mode log-weights against the float64 checker
on the device's own poses (golden cases; the checker is imported here as a checker only).  Both passes:
the one that also forms the raw product (general mixture) and the log-weight-only one (float32 mixture)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import mcl_oracle as O
from slam_amd import _lib as L
import test_gpu_parity as T
for name in ["demo_seed0", "synth256_b180", "synth256_b360", "synth128_b13"]:
    g = T.load(name)
    ctx = T._ctx(name, g, L.F32)
    grid = T._grid(name, g)
    lut, cell = O.obstacle_lut(grid)
    s = T.steps(g)[0]
    for mode in ("product", "log-only"):
        ctx.set_particles(s["X_in"])
        if mode == "product":
            ctx.sample(s["ut"][0], s["ut"][1], float(s["dt"]), T.ALPHA, s["z_meas"], s["noise"])
        else:
            ctx.shard_sample(s["ut"][0], s["ut"][1], float(s["dt"]), T.ALPHA, s["z_meas"], s["noise"], 0, L.W_LOG)
        X, w, logw = ctx.get_particles(True, True, True)
        Ro, Ho = O.sector_query_batch(X, lut, cell, list(g["sensor_angles"]))
        sp = O.SensorParams()
        wo, lwo, _ = O.weights_from_ranges(Ro, s["z_meas"], O.free_cell(X[:, 0], X[:, 1], grid), sp)
        fin = np.isfinite(lwo)
        d = logw[fin] - lwo[fin]
        rel = np.abs(d) / np.maximum(1.0, np.abs(lwo[fin]))
        j = int(np.argmax(rel))
        print("%-16s %-9s max rel %.3e  mean signed rel %.3e  max abs %.3e | worst: logw %.4f abs err %.3e" % (
            name, mode, rel.max(), np.mean(d / np.maximum(1.0, np.abs(lwo[fin]))), np.max(np.abs(d)), lwo[fin][j], d[j]))
        if mode == "log-only":
            # per-beam view of the worst particle: which beams carry the error budget
            m = np.nonzero(fin)[0][j]
            z, zs = Ro[m], np.asarray(s["z_meas"], np.float64)
            p = O.beam_px(z, zs, sp)
            dd = z - zs
            amp = np.abs(dd) / sp.sigma_hit ** 2
            print("      worst particle: %d beams with |d|<0.7; sum |d|/s^2 over them %.1f; beams at z_max %d" % (
                int(np.sum(np.abs(dd) < 0.7)), float(np.sum(amp[np.abs(dd) < 0.7])), int(np.sum(z >= 5.0))))
