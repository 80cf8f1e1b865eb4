"""What the benchmark scripts quote from the committed profiles: the newest profiles/r*_<name>.json and per-kernel
HBM traffic (rocprofv3 FETCH_SIZE and WRITE_SIZE count kilobytes; on gfx950 a fetch request stands for two of them)."""
import glob, json, os

PROFILES = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def latest(name):
    """(path, parsed json) of the newest profiles/r*_<name>.json, or (None, None)."""
    found = sorted(glob.glob(os.path.join(PROFILES, "r[0-9]*_%s.json" % name)))
    if not found:
        return None, None
    return os.path.relpath(found[-1], os.path.dirname(PROFILES)), json.load(open(found[-1]))


def hbm_bytes(counters):
    """HBM bytes per launch of one kernel's counter set, or None if the passes are missing."""
    if "FETCH_SIZE" not in counters or "WRITE_SIZE" not in counters:
        return None
    return (2.0 * counters["FETCH_SIZE"]["per_launch_mean"] + counters["WRITE_SIZE"]["per_launch_mean"]) * 1024.0


def traffic_per_step(pmc, steps, keep=lambda name: True):
    """Sum over kernels of HBM bytes per launch x launches, divided by the number of filter steps the profiled run made."""
    total, per_kernel = 0.0, {}
    for name, c in pmc.items():
        b = hbm_bytes(c)
        if b is None or not keep(name):
            continue
        n = c["FETCH_SIZE"]["launches"]
        per_kernel[name.split("(")[0].replace("void ", "")] = b * n / steps
        total += b * n
    return total / steps, per_kernel
