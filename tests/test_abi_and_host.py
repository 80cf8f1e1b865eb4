"""CPU-side checks: the C-ABI library loads and exports every symbol include/mclhip.h declares,
the ctypes table covers the header, and the host-side helpers behave like the reference's."""
import ctypes
import os
import re
import subprocess
import warnings

import numpy as np
import pytest

from oracle import mcl_oracle as O
from util import GOLDEN, load, steps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mclhip.h")
LIB = os.path.join(ROOT, "slam_amd", "libmclhip.so")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mclhip_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(LIB)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    out = subprocess.check_output(["nm", "-D", "--defined-only", LIB]).decode()
    exported = set(re.findall(r"\bT (mclhip_\w+)", out))
    assert set(names) <= exported


def test_ctypes_table_matches_header():
    from slam_amd import _lib
    assert sorted(_lib.PROTOTYPES) == _declared()
    lib = _lib.load()
    assert lib.mclhip_abi_version() == 2


def test_no_device_is_a_loud_error():
    from slam_amd import _lib
    lib = _lib.load()
    if lib.mclhip_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.MclHipError):
        _lib.Context(0, _lib.F64)
    from slam_amd.mc_localization import MCLocalization
    with pytest.raises(_lib.MclHipError):
        MCLocalization(np.zeros(3), np.full((8, 8), 255, np.uint8))


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "slam_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            assert not re.search(r"^\s*(from|import)\s+.*oracle", open(os.path.join(pkg, f)).read(), re.M), f
    assert "oracle" not in open(os.path.join(ROOT, "slam_amd", "csrc", "api.hip")).read()


def test_host_noise_sampler_consumes_the_stream_like_the_reference():
    """sample_normal_distribution: 12 integers from np.random per call, b/6 * sum(k*0.001)."""
    from slam_amd import mc_localization as M
    np.random.seed(3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ints = np.random.random_integers(-1000, 1000, size=(5, 12))
    np.random.seed(3)
    for i in range(5):
        v = M.sample_normal_distribution(6.0)
        assert v == 6.0 / 6.0 * O.sum12(ints[i])


def test_true_pose_update_matches_golden_including_float32_first_step():
    from slam_amd.mc_localization import advance_true_pose
    g = load("demo_seed0")
    for s in steps(g):
        x_in = s["x_gt_in"].astype(np.float32) if bool(s["x_gt_in_is_f32"]) else s["x_gt_in"]
        assert np.array_equal(advance_true_pose(x_in, [1.5, 0], 0.1), s["x_gt_out"])
    g = load("demo_seed2")
    for s in steps(g):
        x_in = s["x_gt_in"].astype(np.float32) if bool(s["x_gt_in_is_f32"]) else s["x_gt_in"]
        assert np.array_equal(advance_true_pose(x_in, [float(s["ut"][0]), float(s["ut"][1])], 0.1), s["x_gt_out"])


def test_workload_generators_are_deterministic():
    from slam_amd import workload as W
    a = W.synthetic_map(256)
    b = W.synthetic_map(256)
    assert np.array_equal(a, b) and set(np.unique(a)) == {0, 255}
    assert 0.035 < np.mean(a == 0) < 0.06
    X = W.uniform_particles(a, 1000)
    assert np.array_equal(X, W.uniform_particles(a, 1000))
    u = np.trunc(X[:, 0] / W.RES + 0.5).astype(int)
    v = np.trunc(X[:, 1] / W.RES + 0.5).astype(int)
    assert np.all(a[v, u] == 255)


def test_particle_filter_histogram_matches_reference():
    """ParticleFilter.update_pdf (host part of the chapter-4 drop-in) on the resampled sets the reference produced:
    weights accumulated per rounded position in particle order, then normalised -- bit for bit."""
    from slam_amd import particle_filter as pf
    g = load("particle_filter04")
    for n in range(int(g["n_cases"])):
        flt = pf.ParticleFilter.__new__(pf.ParticleFilter)      # no device needed for this method
        flt.M, flt.x_max = len(g["c%d_w" % n]), int(g["x_max"])
        flt.px = np.zeros(flt.x_max)
        flt.X, flt.w = list(g["c%d_X_out" % n]), list(g["c%d_w_out" % n])
        flt.update_pdf()
        assert np.array_equal(flt.px, g["c%d_px" % n]), n
