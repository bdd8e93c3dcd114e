"""scanner_amd -- MI355X-native spectrum-scan DSP path (drop-in for wpats/scanner's
process.cpp / fft.cpp / utility.cpp per-buffer loop).

The compute lives in hand-written HIP kernels behind the C-ABI of include/scanner_hip.h
(scanner_amd/libscanner_hip.so).  This package is host plumbing around that library:
``capi`` binds it with ctypes, ``plan`` wraps a plan handle, ``synth`` generates
deterministic synthetic IQ, ``sweep`` shards a frequency table over ranks.  There is no
CPU fallback: importing ``capi`` without the built library raises.
"""
from . import capi  # noqa: F401
from .capi import (  # noqa: F401
    KIND_BYTE_COMPLEX, KIND_SHORT, KIND_SHORT_COMPLEX, KIND_FLOAT_COMPLEX, OUT_HITS, OUT_SPECTRUM, PLAN_OVERLAP_SLOTS,
    HIT_DTYPE, ScannerError)
from .plan import Plan, WelchPlan  # noqa: F401
from . import synth, sweep  # noqa: F401,E402
