#!/usr/bin/env python3
"""Benchmark of the MCL update (BASELINE.json: particle-beam evaluations per second).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...                      (starts N ranks itself, one per GPU)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One step = one whole update of the particle set: restore the synthetic cloud
(device-to-device), velocity-motion sampling with device-generated noise, sector
range query + beam likelihood, weight normalisation (numpy summation order) and
order-exact low-variance resampling + gather.  Workload at N=1: configs[1] of
BASELINE.json -- 2^20 particles x 180 beams on a synthetic 4096^2 occupancy
grid, float32 particle state.  With N > 1 every rank holds 2^20 particles of one
N * 2^20 filter (weak scaling).

Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_PER_EVAL = 97.2   # 96.96 one-byte grid cells per +-5 deg sector of radius z_max/res + 0.22 B particle state
HBM_PEAK_GBS = 8000.0
SIMDS = 1024                 # 256 compute units x 4 SIMD-32
PEAK_CLOCK_GHZ = 2.4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", type=int, default=1 << 20, help="particles per GPU")
    ap.add_argument("--beams", type=int, default=180)
    ap.add_argument("--map", type=int, default=4096, help="side of the synthetic occupancy grid in cells")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--regime", default="global", choices=["global", "tracking"],
                    help="global: every step starts from the uniform synthetic cloud; tracking: the filter runs on")
    ap.add_argument("--cpu-particles", type=int, default=8192, help="particles of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--host-collectives", action="store_true",
                    help="multi-rank step with the host combining the ranks' small results (default: device-driven)")
    ap.add_argument("--direct-rccl", action="store_true",
                    help="the device-driven step's collectives through librccl on the step's own stream (slam_amd/rccl_direct.py)")
    ap.add_argument("--rehearse-collectives", action="store_true",
                    help="run the multi-rank code path (RCCL group, cross-rank resampling) even with one rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of a multi-rank run (gloo: host-combined exchange, for rehearsals)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="every rank uses GPU 0 (rehearsal of the multi-rank path on a one-GPU box; needs --backend gloo)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher check without a GPU: the ranks rendezvous over gloo and rank 0 prints the line")
    return ap.parse_args()


def visible_gpus(topology="/sys/class/kfd/kfd/topology/nodes"):
    """GPUs this process would see, counted WITHOUT loading torch or the HIP runtime (the launcher forks the ranks and
    must not have initialised a GPU itself): the kernel driver's topology nodes that have SIMDs, narrowed by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES when set.  None when the topology cannot be read
    (the ranks then report a missing device themselves)."""
    import glob
    nodes = 0
    files = glob.glob(os.path.join(topology, "*", "properties"))
    if not files:
        return None
    for f in files:
        try:
            for line in open(f):
                k, _, v = line.partition(" ")
                if k == "simd_count" and int(v) > 0:
                    nodes += 1
        except (OSError, ValueError):
            return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            nodes = min(nodes, len([x for x in v.split(",") if x.strip() != ""]))
    return nodes


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher around it: start N fresh rank processes (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment) and wait for them.  The parent touches no GPU.  Rank 0 writes the
    JSON line to this process's stdout; the exit status is non-zero if any rank fails."""
    import socket
    import subprocess
    n = a.gpus
    if not (a.dry_run or a.share_gpu):
        have = visible_gpus()
        if have is not None and have < n:
            sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible\n" % (n, have))
            return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        pending = list(range(n))
        while pending:
            for r in list(pending):
                code = procs[r].poll()
                if code is None:
                    continue
                pending.remove(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    sys.stderr.write("bench.py: rank %d exited with status %d\n" % (r, code))
            if rc:
                break
            time.sleep(0.05)
    finally:
        for p in procs:              # a failed rank leaves the others waiting in a collective
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    return rc


def cpu_baseline(grid, angles, X, z_meas, ut, n_particles):
    """The oracle (numpy restatement of the reference, one core) on a bounded sample of the
    same workload: one whole update of `n_particles` particles against the full obstacle list."""
    from oracle import mcl_oracle as O
    rng = np.random.RandomState(7)
    Xs = X[:n_particles].copy()
    noise = rng.randint(-1000, 1001, size=(n_particles, 3, 12)).astype(np.int16)
    orc = O.OracleMCL(grid, angles, Xs[0].copy())
    orc.X = Xs
    t0 = time.perf_counter()
    orc.z_meas = z_meas
    orc.X, orc.pm = O.motion(orc.X, ut, 0.1, grid, noise)
    orc.R, _ = O.sector_query_batch(orc.X, orc.lut, orc.cell, angles)
    orc.w, orc.logw, _ = O.weights_from_ranges(orc.R, z_meas, orc.pm, orc.sp)
    lw = orc.logw - np.max(orc.logw)
    orc.w = np.exp(lw)
    orc.resample(rng.rand() / n_particles, clamp=True)
    dt = time.perf_counter() - t0
    return {"value": n_particles * len(angles) / dt, "unit": "particle-beam evals/s", "cores": 1, "host_cores": os.cpu_count(),
            "host_cores_usable": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "kind": "port",
            "sample": "%d particles x %d beams, full obstacle list (%d cells), one update in %.1f s; "
                      "cost is linear in the particle count" % (n_particles, len(angles), len(orc.lut), dt)}


def stream_copy_ceiling(torch):
    """Measured device-to-device copy rate (bytes read + bytes written per second) of a 1-GiB buffer: what this
    GPU's HBM delivers to the simplest streaming kernel; GB/s."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device="cuda")
    b = torch.empty(n, dtype=torch.float32, device="cuda")
    a.fill_(1.0)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    return 2.0 * n * 4 / (ms * 1e-3) / 1e9


def profiled_mix(args, kernel_name):
    """What the committed counter passes say about one launch of the sector-query kernel (profiles/rNN_valu_roofline*.json,
    written by tools/valu_roofline.py from rocprofv3 --pmc passes): vector wave-instructions priced with their measured
    issue cost, the hardware's own count of VALU-active cycles, and the HBM bytes (FETCH_SIZE doubled, as gfx950 tallies
    128-byte requests at 64 bytes, + WRITE_SIZE; separate passes).  A profile is quoted only for the workload its passes
    were run on, and only while it describes the code in the tree: it carries the sha256 of the kernel's source files and
    the name of the kernel, and one whose sources have changed since, or that is of another instantiation than the run
    just timed, comes back with `stale` set -- the caller then reports no fraction.  None: no profile of this workload."""
    import glob
    import hashlib
    want = {"particles": args.particles, "beams": args.beams, "map": args.map, "dtype": args.dtype, "regime": args.regime}
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_valu_roofline*.json"))):
        d = json.load(open(f))
        wl = d.get("workload")
        if wl is None and "grid" not in f:     # profiles of earlier rounds: the default workload, no stamps
            wl = {"particles": 1 << 20, "beams": 180, "map": 4096, "dtype": "f32", "regime": "global"}
        if wl == want:
            best = (f, d)
    if best is None:
        return None
    f, d = best
    d["file"] = os.path.relpath(f, ROOT)
    why = []
    stamps = d.get("source_sha256")
    if not stamps:
        why.append("the profile carries no source stamps")
    else:
        for rel, sha in stamps.items():
            path = os.path.join(ROOT, rel)
            if not os.path.exists(path) or hashlib.sha256(open(path, "rb").read()).hexdigest() != sha:
                why.append("%s has changed since the profile was taken" % rel)
    prof_kernel = d.get("kernel", "").replace("void mcl::", "").replace(" ", "").split("(")[0]
    if kernel_name and prof_kernel != kernel_name.replace(" ", ""):
        why.append("the profile is of %s, the run was %s" % (prof_kernel, kernel_name))
    d["stale"] = why
    return d


def roofline(args, M, B, ray_ms, n_ray, copy_gbs, kernel_name=""):
    """The sector-query kernel against the ceiling that binds it.  It is bound by vector-instruction issue: per
    particle it issues several hundred vector instructions and reads about 1.5 KB, so the figure a launch is held to
    is (vector wave-instructions x issue cycles each) / (SIMDs x clock), with the kernel time measured live; the
    measured HBM rate stands beside it."""
    mix = profiled_mix(args, kernel_name)
    peak = SIMDS * PEAK_CLOCK_GHZ                       # G SIMD issue cycles per second
    out = {"bound": "valu", "achieved": None, "peak": peak, "unit": "G SIMD issue-cycles/s", "frac": None, "traffic": None,
           "kernel": "raycast_kernel (sector query + beam likelihood; where the query runs as two launches, the first: "
                     "the interval-only instantiation, which serves all particles but the few per cent whose heading lies "
                     "outside the sector-interval path's range)", "kernel_ms": ray_ms, "launches": n_ray,
           "peak_is": "%d SIMD-32 x %.1f GHz" % (SIMDS, PEAK_CLOCK_GHZ),
           "algorithmic_bytes_per_eval": ALGO_BYTES_PER_EVAL,
           "algorithmic_bytes_per_launch": ALGO_BYTES_PER_EVAL * float(M) * B,
           "stream_copy_GBps": copy_gbs, "kernel_instantiation": kernel_name}
    if mix and mix["stale"]:
        out["stale"] = True
        out["stale_because"] = mix["stale"]
        out["source"] = mix["file"]
        return out
    if mix and n_ray:
        sec = ray_ms * 1e-3
        cyc = mix["priced_issue_cycles_per_launch"]
        out["achieved"] = cyc / sec / 1e9
        out["frac"] = out["achieved"] / peak
        out["issue_cycles_per_launch"] = cyc
        out["valu_wave_instructions_per_launch"] = mix["valu_wave_instructions_per_launch"]
        out["valu_active_cycles_per_launch_hw"] = mix["hardware_valu_active_cycles_per_launch"]
        out["frac_hw_valu_active"] = mix["hardware_valu_active_cycles_per_launch"] / sec / 1e9 / peak
        out["frac_hw_valu_active_is"] = ("SQ_ACTIVE_INST_VALU x 4 over SIMD cycles (rocprof's VALUBusy): counted per wave in "
                                         "quad-cycles, it passes 1 when the vector pipeline is saturated with two-cycle "
                                         "instructions; the priced fraction above is the one to hold the kernel to")
        out["traffic"] = mix["hbm_bytes_per_launch"]
        out["traffic_unit"] = "HBM bytes per launch (rocprofv3 FETCH_SIZE*2 + WRITE_SIZE, %s)" % mix["file"]
        if mix["hbm_bytes_per_launch"]:
            gbs = mix["hbm_bytes_per_launch"] / sec / 1e9
            out["hbm"] = {"achieved_GBps": gbs, "peak_GBps": HBM_PEAK_GBS, "frac": gbs / HBM_PEAK_GBS,
                          "l2_hit_rate": mix.get("l2_hit_rate")}
        out["source"] = mix["file"]
        out["stale"] = False
        if mix.get("lane_utilisation_hw") is not None:
            out["lane_utilisation_hw"] = mix["lane_utilisation_hw"]
        if mix.get("cell_loop"):
            out["lane_utilisation_of_the_cell_loop"] = mix["cell_loop"].get("lane_utilisation_of_the_cell_loop")
        if mix.get("algorithmic_floor"):
            af = mix["algorithmic_floor"]
            out["algorithmic_floor_ms"] = af["floor_ms_at_peak_clock"]
            out["frac_of_algorithmic_floor"] = af["floor_ms_at_peak_clock"] / ray_ms
            out["algorithmic_floor_is"] = ("fewest vector wave-instructions per particle with every lane doing needed work "
                                           "(%d + %d x cells within z_max / 64 + %d x beams / 64 = %.0f against %.0f issued), at the "
                                           "kernel's average issue cost" % (af["lane_instructions"]["per_particle"],
                                                                            af["lane_instructions"]["per_cell_in_range"],
                                                                            af["lane_instructions"]["per_beam"],
                                                                            af["wave_instructions_per_particle"],
                                                                            af["wave_instructions_per_particle_issued"]))
    return out


def dry_run(a, rank, world, json_fd):
    """Launcher check: rendezvous over gloo, one collective, rank 0 prints a line that says how many ranks it saw."""
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    dist.barrier()
    # what the real run would set up, checked without a GPU: the layout of the global particle vector over the ranks
    # (every rank but the last holds whole 8192-element chunks: the sum is formed chunk by chunk in rank order), and the
    # device memory one rank's particle state takes
    M = a.particles
    problems = []
    if M % 8192 and world > 1:
        problems.append("particles per GPU must be a multiple of 8192 on every rank but the last")
    if M > (1 << 31):
        problems.append("at most 2^31 particles per context")
    if a.beams > 1024:
        problems.append("at most 1024 beams")
    if a.map > 65535:
        problems.append("map sides are limited to 65535 cells")
    state = 4 if a.dtype == "f32" else 8
    per_particle = 2 * (3 * state + 8) + 8 + 1 + 8 * 3 + 8 + 4 + 24 + 24 + 8 + 4 + (32 if a.dtype == "f32" else 0)
    if rank == 0:
        line = json.dumps({"metric": "particle-beam evals/sec (node) + ms per MCL step, 1M particles x 180 beams",
                           "value": None, "unit": "particle-beam evals/s", "n_gpus": world, "dry_run": True,
                           "world_size_seen": int(t.item()), "steps": a.steps, "warmup": a.warmup,
                           "plan": {"particles_total": M * world, "particles_per_gpu": M, "beams": a.beams, "map": a.map,
                                    "first_index_per_rank": [r * M for r in range(world)],
                                    "chunks_of_8192_per_rank": (M + 8191) // 8192,
                                    "fused_tail": (M + 8191) // 8192 <= 1024,
                                    "particle_state_bytes_per_gpu": per_particle * M,
                                    "evaluations_per_step": float(M) * world * a.beams,
                                    "problems": problems}})
        os.write(json_fd, (line + "\n").encode())
    dist.destroy_process_group()
    if problems:
        sys.exit(2)


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(launch_ranks(a))          # this process starts the ranks and touches no GPU itself
    # stdout carries exactly one JSON line: whatever libraries print on the way (RCCL's version banner
    # goes to stdout) is sent to stderr instead
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if a.dry_run:
        return dry_run(a, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), json_fd)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; running %d rank(s)\n" % (a.gpus, world, world))
    if a.share_gpu:
        if world > 1 and a.backend != "gloo":
            sys.exit("bench.py: --share-gpu needs --backend gloo (RCCL refuses two ranks on one device)")
        local_rank = 0
    import torch
    dist = None
    if world > 1 or a.rehearse_collectives:
        import datetime
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        limit = datetime.timedelta(seconds=300)     # a rank that died leaves the others in a collective: give up, do not hang
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank), timeout=limit)
        elif a.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=limit)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=limit)
    from slam_amd import _lib, workload as W
    from slam_amd.sharded import ShardedFilter

    M, B = a.particles, a.beams
    grid = W.synthetic_map(a.map)
    angles = W.beam_angles(B)
    ctx = _lib.Context(local_rank, _lib.F32 if a.dtype == "f32" else _lib.F64)
    ctx.set_map(grid, W.RES, W.RES)
    w4 = np.array([0.97, 0.005, 0.020, 0.005])
    w4 = w4 / w4.sum()
    ctx.set_sensor(angles, W.Z_MAX, 0.15, 1.0, w4, 5 * np.pi / 180.0)
    # The global cloud is the ranks' slices one after the other; every rank draws only its own (seeded by its rank:
    # rank 0's slice is the one-GPU cloud), not the whole set -- 2^26 poses at configs[2] -- to keep one eighth of it.
    X = W.uniform_particles(grid, M, seed=W.SEED_PARTICLES + 7919 * rank)
    ctx.set_particles(X)
    ctx.snapshot_particles()

    # the true pose follows the controls; its measured ranges are an input of every step
    n_total = a.warmup + a.steps
    ctrl = W.controls(n_total)
    from slam_amd.mc_localization import advance_true_pose
    x_gt = X[0].copy()                                      # the true pose starts as particle 0 of rank 0
    if dist is not None and world > 1:
        t_gt = torch.from_numpy(x_gt).to("cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.broadcast(t_gt, 0)
        x_gt = t_gt.cpu().numpy().copy()
    z_meas = []
    for (v, om) in ctrl:
        x_gt = advance_true_pose(x_gt, (v, om), 0.1)
        z_meas.append(ctx.raycast(x_gt, want_hits=False)[0][0])
    rs = np.random.RandomState(W.SEED_STREAM)
    r_draw = rs.rand(n_total) / (M * world)
    alpha = [4, 2, 4, 4, 4, 2]
    filt = ShardedFilter(ctx, dist, rank, world, M, force_collectives=a.rehearse_collectives,
                         device_path=False if a.host_collectives else (True if (dist is not None and a.backend == "gloo"
                                                                                    and world > 1) else None),
                         direct_rccl=True if a.direct_rccl else None)

    def one_step(t):
        if a.regime == "global":
            ctx.restore_particles_async()
        v, om = ctrl[t]
        filt.step(v, om, 0.1, alpha, z_meas[t], r_draw[t], seed=W.SEED_STREAM * 1000 + t)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for t in range(a.warmup):
        one_step(t)
    fence()
    ctx.timing_enable(2)          # events around the sector-query kernel only: every event is idle GPU time
    t0 = time.perf_counter()
    for t in range(a.warmup, n_total):
        one_step(t)
    ctx.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    # rows of the exchange: the largest per-rank figures (the slowest rank sets the step time)
    ex_rows = getattr(filt, "rows_sent_total", 0) / max(getattr(filt, "steps_done", 0), 1)
    ex_off = getattr(filt, "rows_sent_off_rank", 0) / max(getattr(filt, "steps_done", 0), 1)
    if dist is not None and world > 1:
        te = torch.tensor([ex_rows, ex_off], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        ex_rows, ex_off = float(te[0].item()), float(te[1].item())
    n_ray, ms_ray = ctx.timing_get(0)
    try:
        kernel_name = ctx.query_kernel_name()
    except AttributeError:                   # a library build without the name
        kernel_name = ""
    # the other kernel groups, for the record: a few more steps outside the timed region with every group timed
    ctx.timing_enable(True)
    for t in range(a.warmup, min(a.warmup + 5, n_total)):
        one_step(t)
    fence()
    n_mot, ms_mot = ctx.timing_get(1)
    n_res, ms_res = ctx.timing_get(2)
    n_left, ms_left = ctx.timing_get(5)      # the sector query's second launch (particles the first left over), if any
    try:
        left_over, _ = ctx.timing_get(6)
    except _lib.MclHipError:                 # a library build without the count
        left_over = None
    ctx.timing_enable(False)

    if rank == 0:
        copy_gbs = stream_copy_ceiling(torch)
        evals_step = float(M) * world * B
        ms_step = elapsed / a.steps * 1e3
        ray_ms = ms_ray / max(n_ray, 1)
        out = {
            "metric": "particle-beam evals/sec (node) + ms per MCL step, 1M particles x 180 beams",
            "value": evals_step / (elapsed / a.steps),
            "unit": "particle-beam evals/s",
            "n_gpus": world, "world_size_seen": (dist.get_world_size() if dist is not None else 1),
            "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64 geometry predicates + %s state/likelihood" % a.dtype,
            "data": "synthetic",
            "config": {"workload": "MCL %d particles/GPU x %d beams on synthetic %dx%d occupancy grid (%.1f%% occupied), "
                                   "%s state, device noise, log-weights, regime=%s"
                                   % (M, B, a.map, a.map, 100.0 * float(np.mean(grid == 0)), a.dtype, a.regime),
                       "particles_per_gpu": M, "beams": B, "map": a.map, "regime": a.regime,
                       "sharding": ("contiguous particle ranges, map replicated; %s" %
                                    ("device-driven exchange" if filt.device_path else "host-combined exchange"))
                       if filt.collective else "single GPU"},
            "roofline": roofline(a, M, B, ray_ms, n_ray, copy_gbs, kernel_name),
            "exchange": ({"rows_per_step_max_over_ranks": ex_rows,
                          "bytes_between_ranks_per_step_max_over_ranks": 48.0 * ex_off,
                          "host_reads_per_step": 1 if filt.device_path else "host-combined: about 13",
                          "host_read_is": (("after the step's last kernel has been enqueued (rows travel in blocks of %d per pair "
                                            "of ranks; the device says whether they were large enough)" % filt.fixed_rows)
                                           if getattr(filt, "fixed_rows", 0) else "in mid-step: the row counts of the all-to-all")
                          if filt.device_path else None,
                          "steps_finished_with_exact_counts": getattr(filt, "exact_exchanges", None),
                          "collectives": ("librccl on the step's own stream" if getattr(filt, "direct", None) is not None
                                          else "torch.distributed"),
                          "host_ms_per_step": ({k: 1e3 * v / max(getattr(filt, "steps_done", 0), 1)
                                                for k, v in getattr(filt, "clock", {}).items()} or None),
                          "note": "only rows that some position draws travel (48 B each)"}
                         if filt.collective else None),
            "kernel_ms": {"raycast": ray_ms, "raycast_leftover_launch": (ms_left / n_left) if n_left else None,
                          "raycast_leftover_particles": left_over,
                          "motion": ms_mot / max(n_mot, 1),
                          "normalise+resample+gather": ms_res / max(n_res, 1)},
        }
        if world == 1 and a.cpu_particles > 0:
            out["cpu_baseline"] = cpu_baseline(grid, angles, X, z_meas[0], ctrl[0], a.cpu_particles)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
