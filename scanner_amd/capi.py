"""ctypes binding of include/scanner_hip.h (libscanner_hip.so).

This is the Python twin of the stub a C++ maintainer would write against the header
(INTEGRATION.md).  It fails loudly when the library is missing: the product has no CPU path.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCN_LIB selects an experiment build (scripts/build_variants.py); the product library otherwise
LIB_PATH = os.environ.get("SCN_LIB") or os.path.join(_HERE, "libscanner_hip.so")

OK, E_INVALID, E_HIP, E_NOMEM, E_STATE, E_TRUNCATED, E_NO_DEVICE, E_COMM = range(8)
KIND_BYTE_COMPLEX, KIND_SHORT, KIND_SHORT_COMPLEX, KIND_FLOAT_COMPLEX = 1, 2, 3, 4
MODE_TIME_DOMAIN, MODE_FREQUENCY_DOMAIN = 1, 2
WIN_RECTANGULAR, WIN_BLACKMAN_HARRIS = 3, 5
WIN_HANN, WIN_BLACKMAN, WIN_KAISER, WIN_BARTLETT, WIN_FLATTOP, WIN_HAMMING = 1, 2, 4, 6, 7, 8   # (0, GNU Radio's WIN_HAMMING, is "the default" here)
OUT_SPECTRUM, OUT_HITS = 1, 2
PLAN_OVERLAP_SLOTS = 4  # each slot on its own compute stream (scanner_hip.h)
DC_IGNORE_NONE = 0xFFFFFFFF
NUM_SLOTS = 4
ABI_VERSION = 5
PATH_UNSUPPORTED, PATH_FUSED, PATH_FOUR_STEP, PATH_STAGED, PATH_BLUESTEIN = range(5)
COMM_ID_BYTES = 128
GATHER_TICKETS = 4

BYTES_PER_SAMPLE = {KIND_BYTE_COMPLEX: 2, KIND_SHORT: 4, KIND_SHORT_COMPLEX: 4, KIND_FLOAT_COMPLEX: 8}

# struct scn_hit
HIT_DTYPE = np.dtype([("seq_id", "<u8"), ("i", "<u4"), ("power_db", "<f4"), ("freq_hz", "<u8")], align=True)
assert HIT_DTYPE.itemsize == 24


class PlanDesc(C.Structure):
    """struct scn_plan_desc"""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n", C.c_uint32),
        ("sample_rate", C.c_uint32),
        ("sample_kind", C.c_uint32),
        ("enob", C.c_uint32),
        ("correct_dc", C.c_uint32),
        ("window_type", C.c_uint32),
        ("mode", C.c_uint32),
        ("threshold", C.c_float),
        ("dc_ignore_bins", C.c_uint32),
        ("use_bandwidth", C.c_double),
        ("trigger_count", C.c_uint32),
        ("max_batch", C.c_uint32),
        ("max_hits", C.c_uint32),
        ("flags", C.c_uint32),
        ("device_id", C.c_int32),
        ("reserved", C.c_uint32 * 5),
    ]


class WelchDesc(C.Structure):
    """struct scn_welch_desc"""
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n", C.c_uint32),
        ("segments_per_psd", C.c_uint32),
        ("window_type", C.c_uint32),
        ("max_psd", C.c_uint32),
        ("device_id", C.c_int32),
        ("sample_kind", C.c_uint32),
        ("enob", C.c_uint32),
        ("correct_dc", C.c_uint32),
        ("reserved", C.c_uint32 * 1),
    ]


# every symbol include/scanner_hip.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
SYMBOLS = {
    "scn_error_name": (C.c_char_p, [C.c_int]),
    "scn_last_error": (C.c_char_p, []),
    "scn_abi_version": (C.c_uint32, []),
    "scn_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "scn_size_path": (C.c_int, [C.c_uint32, C.POINTER(C.c_uint32)]),
    "scn_plan_create": (C.c_int, [C.POINTER(PlanDesc), C.POINTER(_vp)]),
    "scn_plan_destroy": (C.c_int, [_vp]),
    "scn_buffer_bytes": (C.c_int, [_vp, C.POINTER(C.c_size_t)]),
    "scn_host_buffer": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "scn_submit": (C.c_int, [_vp, C.c_int, C.c_uint32, _vp, _vp]),
    "scn_submit_device": (C.c_int, [_vp, C.c_int, _vp, C.c_uint32, _vp, _vp, _vp]),
    "scn_plan_set_table": (C.c_int, [_vp, _vp, C.c_uint32]),
    "scn_submit_indexed": (C.c_int, [_vp, C.c_int, C.c_uint32, C.c_uint32, _vp]),
    "scn_submit_device_indexed": (C.c_int, [_vp, C.c_int, _vp, C.c_uint32, C.c_uint32, _vp, _vp]),
    "scn_collect": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_uint32, C.POINTER(C.c_uint32), _vp]),
    "scn_collect_more": (C.c_int, [_vp, C.c_int, C.c_uint32, _vp, C.c_uint32, C.POINTER(C.c_uint32)]),
    "scn_hits_view": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(C.c_uint32)]),
    "scn_collect_time_domain": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp]),
    "scn_convert_raw": (C.c_int, [_vp, _vp, C.c_uint32, _vp]),
    "scn_wait": (C.c_int, [_vp, C.c_int]),
    "scn_plan_stream": (C.c_int, [_vp, C.POINTER(_vp)]),
    "scn_slot_stream": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "scn_device_spectrum": (C.c_int, [_vp, C.c_int, C.POINTER(_vp)]),
    "scn_plan_window": (C.c_int, [_vp, _vp, C.c_uint32]),
    "scn_gather_post": (C.c_int, [_vp, _vp, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "scn_gather_wait": (C.c_int, [_vp, C.c_uint32, C.POINTER(_vp), C.POINTER(C.c_uint64), _vp]),
    "scn_welch_create": (C.c_int, [C.POINTER(WelchDesc), C.POINTER(_vp)]),
    "scn_welch_destroy": (C.c_int, [_vp]),
    "scn_welch_samples": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_size_t)]),
    "scn_welch_partition": (C.c_int, [_vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "scn_welch_host_buffer": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(C.c_size_t)]),
    "scn_welch_submit": (C.c_int, [_vp, C.c_int, C.c_uint32]),
    "scn_welch_submit_device": (C.c_int, [_vp, C.c_int, _vp, C.c_uint32, _vp]),
    "scn_welch_collect": (C.c_int, [_vp, C.c_int, _vp]),
    "scn_frequency_table": (C.c_int, [C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint32,
                                      C.c_uint32, _vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "scn_comm_unique_id": (C.c_int, [_vp]),
    "scn_comm_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(_vp)]),
    "scn_comm_destroy": (C.c_int, [_vp]),
    "scn_gather_hits": (C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_uint64), _vp]),
    "scn_gather_hits_device": (C.c_int, [_vp, _vp, C.c_int, C.c_uint32, _vp, C.c_uint64, C.POINTER(C.c_uint64), _vp]),
    "scn_gather_fetch": (C.c_int, [_vp, C.c_uint64, _vp, C.c_uint64, C.POINTER(C.c_uint64)]),
    "scn_gather_layout": (C.c_int, [_vp, C.c_uint32, _vp]),
    "scn_hackrf_sweep_fixup": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_double),
                                         C.POINTER(C.c_uint32)]),
}


class ScannerError(RuntimeError):
    def __init__(self, status, where, detail):
        self.status = status
        super().__init__(f"{where}: {detail} (status {status})")


_lib = None


def lib():
    """Load libscanner_hip.so (once).  Raises if it has not been built -- there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m scanner_amd.build` "
                "(hipcc --offload-arch=gfx950); scanner_amd has no CPU fallback")
        try:
            # share the process's HIP runtime with torch when torch is in use
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is optional for the pure C-ABI
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            if os.environ.get("SCN_LIB") and not hasattr(L, name):
                continue  # an experiment build (SCN_LIB) of an older tree may lack newer entry points; the product library may not
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.scn_abi_version() != ABI_VERSION:
            raise ImportError("libscanner_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(status, where):
    if status != OK:
        L = lib()
        raise ScannerError(status, where, f"{L.scn_error_name(status).decode()}: {L.scn_last_error().decode()}")


def size_path(n):
    """PATH_* of a frequency-domain plan of n points (scn_size_path; needs no device)."""
    out = C.c_uint32()
    check(lib().scn_size_path(int(n), C.byref(out)), "scn_size_path")
    return out.value


def frequency_table(sample_rate, start, stop, use_bandwidth=0.75, dc_ignore_width=0.0, shard=0, n_shards=1):
    """Centre frequencies of frequencyTable.cpp:9-37 (optionally one contiguous shard).
    Returns (first_index, float64 array)."""
    L = lib()
    cnt, first = C.c_uint32(), C.c_uint32()
    check(L.scn_frequency_table(int(sample_rate), start, stop, use_bandwidth, dc_ignore_width, shard, n_shards,
                                None, 0, C.byref(cnt), C.byref(first)), "scn_frequency_table")
    out = np.empty(cnt.value, np.float64)
    check(L.scn_frequency_table(int(sample_rate), start, stop, use_bandwidth, dc_ignore_width, shard, n_shards,
                                out.ctypes.data_as(_vp), cnt.value, C.byref(cnt), C.byref(first)),
          "scn_frequency_table")
    return first.value, out


def gather_layout(per_rank):
    """offsets[r] of rank r's records in the rank-major gathered hit list, total last (scn_gather_layout)."""
    per_rank = np.ascontiguousarray(per_rank, np.uint32)
    off = np.empty(per_rank.size + 1, np.uint64)
    check(lib().scn_gather_layout(per_rank.ctypes.data_as(_vp), per_rank.size, off.ctypes.data_as(_vp)), "scn_gather_layout")
    return off


def hackrf_sweep_fixup(transfer, scan_offset_hz=0):
    """HackRFSource::interpolateSamples (hackRFSource.cpp:186-222) on a uint8/int8 numpy transfer,
    IN PLACE.  Returns (centre frequency, mismatch count)."""
    assert transfer.dtype.itemsize == 1 and transfer.flags.c_contiguous and transfer.flags.writeable
    fc, mism = C.c_double(), C.c_uint32()
    check(lib().scn_hackrf_sweep_fixup(transfer.ctypes.data_as(_vp), transfer.size, int(scan_offset_hz),
                                       C.byref(fc), C.byref(mism)), "scn_hackrf_sweep_fixup")
    return fc.value, mism.value
