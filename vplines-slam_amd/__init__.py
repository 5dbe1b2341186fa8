"""vplines-slam_amd: MI355X-native sliding-window bundle adjustment (and line
front-end) of VPLines-SLAM behind a C ABI (include/vplines_ba.h).

Python here is plumbing only: ctypes bindings to the HIP shared library, the
synthetic workload generator and torch.distributed sharding of window batches.
The HIP library is mandatory -- nothing in this package falls back to a CPU
implementation."""
from . import _build  # noqa: F401
from .capi import (  # noqa: F401
    BaOptions, Preintegration, Prior, Window, SolveReport, Context, SlideTracks, default_options, load_hip_library,
    MARGIN_OLD, MARGIN_SECOND_NEW, MARGIN_NONE,
)
from . import workload  # noqa: F401
from . import shard  # noqa: F401
from . import frontend  # noqa: F401
