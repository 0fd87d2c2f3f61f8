"""calico_amd - MI355X-native batch calibration optimizer (hot path of yangjames/Calico).

The compute path is libcalico_hip.so (hand-written HIP for gfx950) behind the C
ABI in include/calico_hip.h; this package is the thin host side: the ctypes
binding (`_capi`), a Python mirror of the reference's BatchOptimizer / Sensor
API (`api`), and synthetic problem generators (`synthetic`).
"""
from . import _capi  # noqa: F401
