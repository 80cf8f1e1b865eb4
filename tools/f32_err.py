"""Debug helper: largest relative error of the float32-mode log-weights against the float64 checker
on the device's own poses (golden cases; the checker is imported here as a checker only)."""
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
    ctx.set_particles(s["X_in"])
    ctx.sample(s["ut"][0], s["ut"][1], float(s["dt"]), T.ALPHA, s["z_meas"], s["noise"])
    X, w, logw = ctx.get_particles(True, True, True)
    Ro, Ho = O.sector_query_batch(X, lut, cell, list(g["sensor_angles"]))
    wo, lwo, _ = O.weights_from_ranges(Ro, s["z_meas"], O.free_cell(X[:, 0], X[:, 1], grid), O.SensorParams())
    fin = np.isfinite(lwo)
    d = logw[fin] - lwo[fin]
    print("%-16s max rel %.3e  mean signed rel %.3e  max abs %.3e" % (
        name, np.max(np.abs(d) / np.maximum(1.0, np.abs(lwo[fin]))), np.mean(d / np.maximum(1.0, np.abs(lwo[fin]))), np.max(np.abs(d))))
