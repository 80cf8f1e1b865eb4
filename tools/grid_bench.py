"""Timing of the histogram filter on the large configuration (2048^2 cells x 72 headings by default)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from slam_amd import _lib as L, workload as W
N = int(os.environ.get("GRID_N", 2048)); D = int(os.environ.get("GRID_D", 72)); steps = int(os.environ.get("GRID_STEPS", 2))
angles = [0, 15, 30, 45, 60, 75, 90, 345, 330, 315, 300, 285, 270]
grid = W.synthetic_map(N)
ctx = L.Context(0, L.F64)
t0 = time.time()
ctx.grid_set_map(grid, W.RES, W.RES, 200)
w4 = np.array([0.97, 0.005, 0.020, 0.005]); w4 /= w4.sum()
ctx.grid_set_sensor(angles, D, 360.0 / D, W.Z_MAX, 0.15, 1.0, w4, 5 * np.pi / 180.0)
ctx.grid_init_belief()
print("set-up %.1f s" % (time.time() - t0), flush=True)
z = np.array([2.0, 3.1, 5.0, 1.2, 0.8, 4.4, 5.0, 2.2, 2.9, 3.3, 5.0, 1.0, 0.6])
ctx.timing_enable(True)
for t in range(steps):
    ctx.grid_predict(1.5, 0.0 if t % 2 == 0 else 0.5, 0.1, [0.2, 0.1, 0.2, 0.2, 0.1, 0.1])
    s = ctx.grid_update(z)
    print("step", t, "sum", s, flush=True)
np_, msp = ctx.timing_get(3); nu, msu = ctx.timing_get(4)
cells = N * N
pred_ms, upd_ms = msp / max(np_, 1), msu / max(nu, 1)
free = int(np.count_nonzero(grid >= 200))
evals = free * D * 13                      # (cell, heading, beam) sector evaluations of one update
# the update is bound by the rate at which the SIMDs issue its vector instructions: the instruction mix of one launch,
# priced with the measured issue cost of every opcode (profiles/r*_grid_valu_roofline.json, collected at this
# configuration), against 1024 SIMDs x 2.4 GHz
from profile_facts import latest
src, prof = latest("grid_valu_roofline")
same = prof is not None and (N, D) == (2048, 72)
roof = {"kernel": "raycast_kernel GRID mode (update)", "bound": "valu", "unit": "G SIMD issue-cycles/s", "peak": 1024 * 2.4,
        "achieved": None, "frac": None, "traffic": None, "source": src,
        "evaluations_per_second": evals / (upd_ms * 1e-3)}
if same:
    roof["achieved"] = prof["priced_issue_cycles_per_launch"] / (upd_ms * 1e-3) / 1e9
    roof["frac"] = roof["achieved"] / roof["peak"]
    roof["frac_hw_valu_active"] = prof["hardware_valu_active_cycles_per_launch"] / (upd_ms * 1e-3) / 1e9 / roof["peak"]
    roof["traffic"] = prof["hbm_bytes_per_launch"]
    roof["hbm"] = {"achieved_GBps": prof["hbm_bytes_per_launch"] / (upd_ms * 1e-3) / 1e9, "peak_GBps": 8000.0,
                   "frac": prof["hbm_bytes_per_launch"] / (upd_ms * 1e-3) / 1e9 / 8000.0}
out = {"metric": "grid localisation: (cell, heading) poses per second, predict + update", "grid": [N, N, D],
       "value": cells * D / ((pred_ms + upd_ms) * 1e-3), "unit": "poses/s", "predict_ms": pred_ms, "update_ms": upd_ms,
       "roofline": roof,
       "predict_roofline": {"kernel": "grid_predict_kernel", "bound": "fp64 vector FMA (dense 7x7xDxD contraction)",
                            "unit": "TFLOP/s", "peak": 78.6, "achieved": 2.0 * cells * D * D * 49 / (pred_ms * 1e-3) / 1e12,
                            "frac": 2.0 * cells * D * D * 49 / (pred_ms * 1e-3) / 1e12 / 78.6}}
if os.environ.get("GRID_CPU", "1") == "1":
    # CPU baseline: the numpy restatement on the reference's own 100x100x4 problem (one core)
    from oracle import grid_oracle as G
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "grid_seed0.npz"))
    t0 = time.perf_counter()
    pp = G.predict(g["s0_pk_in"], g["map"], (1.5, 0), 4)
    G.update(pp, g["map"], [float(v) for v in g["s0_z_meas"]], 4)
    dtc = time.perf_counter() - t0
    out["cpu_baseline"] = {"value": 100 * 100 * 4 / dtc, "unit": "poses/s", "cores": 1, "kind": "port",
                           "sample": "reference configuration 100x100 cells x 4 headings, one predict + update in %.1f s" % dtc}
import json
print(json.dumps(out))
