"""`from calico_amd import calico` — the reference's Python package surface (calico/calico.cpp:18-437 bindings plus
the helpers of calico/utils.py) over the MI355X HIP backend. The extension module is built by
`__graft_entry__.build()` next to libcalico_hip.so; there is no other backend."""
from .._calico import *  # noqa: F401,F403
from .._calico import __doc__ as _backend_doc  # noqa: F401
from .utils import (ComputeRmseHeatmapAndFeatureCount, DetectionsToCameraMeasurements,  # noqa: F401
                    InitializePinholeAndPoses)
