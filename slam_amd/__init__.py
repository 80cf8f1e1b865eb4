"""slam_amd: the Monte Carlo localization update of trip2eee/slam on MI355X.

`slam_amd.mc_localization.MCLocalization` and `slam_amd.range_sensor.RangeSensor`
keep the reference's Python call surface
(08_grid_and_montecarlo_localization/mc_localization.py, range_sensor.py); the
arithmetic runs in hand-written HIP kernels behind the C ABI of libmclhip.so
(include/mclhip.h), bound with ctypes in `slam_amd._lib`.
"""
from . import _lib  # noqa: F401
from .mc_localization import MCLocalization  # noqa: F401
from .range_sensor import RangeSensor  # noqa: F401

__all__ = ["MCLocalization", "RangeSensor"]
