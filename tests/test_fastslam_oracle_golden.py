"""The FastSLAM oracle against the state the reference's FastSLAM object reached, step by step, under
the same random numbers (tools/gen_golden.py `fastslam`).  Bit-exact."""
import numpy as np
import pytest

from oracle import fastslam_oracle as F
from util import load, steps


@pytest.mark.parametrize("name", ["fastslam_seed0", "fastslam_seed3"])
def test_every_step_bit_exact(name):
    g = load(name)
    assert np.array_equal(g["landmarks"], F.LANDMARKS.astype(np.float64))
    orc = F.OracleFastSLAM(int(g["M"]))
    for t, s in enumerate(steps(g)):
        ut = [float(s["ut"][0]), float(s["ut"][1])]
        orc.step(ut, s["ints"], s["normals"], float(s["r_uniform"][0]))
        assert np.array_equal(np.array(orc.x_gt, np.float64), s["x_gt"]), t
        st = orc.state(s["f_m"].shape[1])
        for k in ("pose", "w", "nfeat", "new_id", "f_m", "f_cov", "f_i", "f_w", "f_id"):
            assert np.array_equal(st[k], s[k]), (t, k)


def test_feature_list_shrinks_under_its_own_loop():
    """fast_slam1.py:380-401 removes from the list it iterates: the element after a removed one is
    skipped in that pass.  Build a particle where two adjacent features run out together."""
    y = F.Part()
    y.x = np.array([[0.0, 0.0, 0.0]]).T
    for j in range(3):
        f = F.Feat(np.array([[1.0 + j, 0.5]]).T, np.eye(2) * 0.1)
        f.i, f.id = 0, j
        y.feature.append(f)
    truth = [(20.0, 0.0, False)]     # nothing sensed: every feature is "not updated" and in range
    F.particle_step(y, [25, 0.001], np.zeros((3, 12), np.int16), np.zeros(0), truth)
    assert [f.id for f in y.feature] == [1]          # 0 removed, 1 skipped, 2 removed
    assert y.feature[0].i == 0 and y.w == F.P0
