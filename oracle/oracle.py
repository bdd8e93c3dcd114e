"""ctypes binding for the CPU oracle (oracle/libscn_oracle.so) + float64 goldens.

TEST INFRASTRUCTURE ONLY.  PARITY PARTLY PINNED: the converters (utility.cpp:9-84), the dB map
(utility.cpp:86-98) and the frequency table are held bit for bit to the reference's own sources compiled into
oracle/_ref (ref_dsp_lib / ref_lib below); the window, the multiply, the FFT and process_fft are not -- the
reference has no golden vectors and those sources cannot be built here (see oracle/scn_oracle.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module;
the product package scanner_amd never does.

Two layers:
  * ``Oracle`` -- the float32 C restatement of utility.cpp / process.cpp /
    fft.cpp (the thing parity is checked against).
  * ``ref64_*`` -- float64 numpy restatements of the third-party arithmetic
    (FFTW DFT definition, GNU Radio Blackman-Harris, VOLK multiply) used to
    pin BOTH the oracle and the HIP path to the mathematics.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libscn_oracle.so")

KIND_BYTE_COMPLEX = 1
KIND_SHORT = 2
KIND_SHORT_COMPLEX = 3
KIND_FLOAT_COMPLEX = 4

HIT_DTYPE = np.dtype(
    [("seq_id", "<u8"), ("i", "<u4"), ("power_db", "<f4"), ("freq_hz", "<u8")], align=True
)
assert HIT_DTYPE.itemsize == 24


class Params(C.Structure):
    _fields_ = [
        ("n", C.c_uint32),
        ("sample_rate", C.c_uint32),
        ("threshold", C.c_float),
        ("use_bandwidth", C.c_double),
        ("dc_ignore_bins", C.c_uint32),
        ("trigger_count", C.c_uint32),
    ]


def build(force=False):
    """Compile oracle/libscn_oracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "scn_oracle.c")
    if (
        force
        or not os.path.exists(_LIB)
        or os.path.getmtime(_LIB) < max(os.path.getmtime(src), os.path.getmtime(src[:-2] + ".h"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB


_REF_LIB = os.path.join(_HERE, "_ref", "libref_frequency_table.so")
_ref = None


def ref_available():
    """Whether oracle/_ref has been built -- without loading it (collection must not map reference code into a GPU run)."""
    return os.path.exists(_REF_LIB)


def ref_lib():
    """The reference's own frequencyTable.cpp, compiled from /root/reference by `make -C oracle ref` (the only
    reference source that builds in this image -- see ref_binding.cpp).  None where it has not been built."""
    global _ref
    if _ref is None and os.path.exists(_REF_LIB):
        L = C.CDLL(_REF_LIB)
        u32, dbl, vp = C.c_uint32, C.c_double, C.c_void_p
        L.ref_frequency_table.restype = u32
        L.ref_frequency_table.argtypes = [u32, dbl, dbl, dbl, dbl, vp, u32]
        L.ref_frequency_walk.argtypes = [u32, dbl, dbl, dbl, dbl, u32, vp, vp, vp]
        L.ref_frequency_start.restype = dbl
        L.ref_frequency_start.argtypes = [u32, dbl, dbl, dbl, dbl]
        L.ref_frequency_stop.restype = dbl
        L.ref_frequency_stop.argtypes = [u32, dbl, dbl, dbl, dbl]
        _ref = L
    return _ref


def ref_frequency_table(sample_rate, start, stop, use_bandwidth=0.75, dc_ignore_width=0.0):
    """frequencyTable.cpp:9-37 as the reference itself computes it (prints its "Frequency i: f" lines to stdout)."""
    L = ref_lib()
    assert L is not None, "oracle/_ref has not been built (needs /root/reference)"
    n = L.ref_frequency_table(int(sample_rate), start, stop, use_bandwidth, dc_ignore_width, None, 0)
    out = np.empty(n, np.float64)
    L.ref_frequency_table(int(sample_rate), start, stop, use_bandwidth, dc_ignore_width, _p(out), n)
    return out


def ref_frequency_walk(sample_rate, start, stop, use_bandwidth, dc_ignore_width, steps):
    """(frequency, iteration count, scan-start flag) of `steps` consecutive tunes, reference code."""
    L = ref_lib()
    assert L is not None
    f = np.empty(steps, np.float64)
    it = np.empty(steps, np.uint32)
    ss = np.empty(steps, np.uint8)
    L.ref_frequency_walk(int(sample_rate), start, stop, use_bandwidth, dc_ignore_width, steps, _p(f), _p(it), _p(ss))
    return f, it, ss


_REF_DSP_LIB = os.path.join(_HERE, "_ref", "libref_utility.so")
_ref_dsp = None


def ref_dsp_available():
    return os.path.exists(_REF_DSP_LIB)


def ref_dsp_lib():
    """The reference's own utility.cpp -- the three integer->complex-float converters (utility.cpp:9-84) and the dB map
    (utility.cpp:86-98) -- compiled from /root/reference by `make -C oracle ref` (see ref_dsp_binding.cpp for how its
    one foreign include is satisfied from this image).  None where it has not been built."""
    global _ref_dsp
    if _ref_dsp is None and os.path.exists(_REF_DSP_LIB):
        L = C.CDLL(_REF_DSP_LIB)
        u32, vp, i = C.c_uint32, C.c_void_p, C.c_int
        L.ref_short_planar_to_float.argtypes = [vp, vp, vp, u32, u32, i]
        L.ref_short_complex_to_float.argtypes = [vp, vp, u32, u32, i]
        L.ref_byte_complex_to_float.argtypes = [vp, vp, u32, u32, i]
        L.ref_complex_to_magnitude.argtypes = [vp, vp, u32]
        _ref_dsp = L
    return _ref_dsp


def ref_convert(kind, raw, n, enob, correct_dc):
    """One buffer through the REFERENCE's converter for `kind` (same argument shapes as Oracle.convert)."""
    L = ref_dsp_lib()
    assert L is not None, "oracle/_ref/libref_utility.so has not been built (needs /root/reference)"
    raw = np.ascontiguousarray(raw).copy()  # (the reference takes non-const pointers)
    out = np.empty(n, np.complex64)
    if kind == KIND_SHORT_COMPLEX:
        L.ref_short_complex_to_float(_p(raw), _p(out), n, enob, int(correct_dc))
    elif kind == KIND_SHORT:
        flat = raw.reshape(-1)
        L.ref_short_planar_to_float(_p(flat), C.c_void_p(flat.ctypes.data + 2 * n), _p(out), n, enob, int(correct_dc))
    elif kind == KIND_BYTE_COMPLEX:
        L.ref_byte_complex_to_float(_p(raw), _p(out), n, enob, int(correct_dc))
    else:
        raise ValueError(kind)
    return out


def ref_magnitude(X):
    """utility.cpp:86-98 as the reference's own object code computes it."""
    L = ref_dsp_lib()
    assert L is not None
    X = np.ascontiguousarray(X, np.complex64).copy()
    out = np.empty(X.size, np.float32)
    L.ref_complex_to_magnitude(_p(X), _p(out), X.size)
    return out


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        vp, u32, f32p = C.c_void_p, C.c_uint32, C.POINTER(C.c_float)
        L.scn_oracle_short_complex_to_float_complex.argtypes = [vp, vp, u32, u32, C.c_int]
        L.scn_oracle_short_planar_to_float_complex.argtypes = [vp, vp, vp, u32, u32, C.c_int]
        L.scn_oracle_byte_complex_to_float_complex.argtypes = [vp, vp, u32, u32, C.c_int]
        L.scn_oracle_window_blackman_harris.argtypes = [vp, u32]
        L.scn_oracle_window_apply.argtypes = [vp, vp, u32]
        L.scn_oracle_window.restype = C.c_int
        L.scn_oracle_window.argtypes = [u32, vp, u32]
        L.scn_oracle_set_window_type.restype = C.c_int
        L.scn_oracle_set_window_type.argtypes = [u32]
        L.scn_oracle_fft_create.restype = vp
        L.scn_oracle_fft_create.argtypes = [u32]
        L.scn_oracle_fft_destroy.argtypes = [vp]
        L.scn_oracle_fft_process.argtypes = [vp, vp, vp]
        L.scn_oracle_complex_to_magnitude.argtypes = [vp, vp, u32, C.c_int]
        L.scn_oracle_process_fft.restype = u32
        L.scn_oracle_process_fft.argtypes = [
            C.POINTER(Params), vp, C.c_double, C.c_uint64, vp, vp, u32, C.POINTER(C.c_int)]
        L.scn_oracle_time_domain.restype = C.c_int
        L.scn_oracle_time_domain.argtypes = [vp, u32, C.c_float, f32p, f32p]
        L.scn_oracle_frequency_table.restype = u32
        L.scn_oracle_frequency_table.argtypes = [
            u32, C.c_double, C.c_double, C.c_double, C.c_double, vp, u32]
        L.scn_oracle_run_batch.restype = C.c_uint64
        L.scn_oracle_run_batch.argtypes = [
            C.POINTER(Params), C.c_int, u32, C.c_int, vp, u32, vp, vp, vp, vp, C.c_uint64, vp, u32]
        L.scn_oracle_welch.argtypes = [vp, u32, u32, u32, vp]
        L.scn_oracle_set_fft_mode.argtypes = [C.c_int]
        L.scn_oracle_set_direct_dft.argtypes = [C.c_int]
        L.scn_oracle_get_fft_mode.restype = C.c_int
        L.scn_oracle_hackrf_interpolate.restype = C.c_double
        L.scn_oracle_hackrf_interpolate.argtypes = [vp, u32, u32, C.POINTER(u32)]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def raw_bytes_per_sample(kind):
    return {KIND_BYTE_COMPLEX: 2, KIND_SHORT: 4, KIND_SHORT_COMPLEX: 4, KIND_FLOAT_COMPLEX: 8}[kind]


class Oracle:
    """The reference's ProcessSamples+SampleQueue arithmetic for one configuration.

    Ctor arguments mirror ProcessSamples (process.h:74-85) and SampleQueue
    (messageQueue.h:141-146)."""

    def __init__(self, n, sample_rate=8000000, threshold=10.0, kind=KIND_FLOAT_COMPLEX, enob=12,
                 correct_dc=False, use_bandwidth=0.75, dc_ignore_bins=4, trigger_count=1047):
        self.n = int(n)
        self.kind = kind
        self.enob = enob
        self.correct_dc = bool(correct_dc)
        self.params = Params(self.n, int(sample_rate), float(threshold), float(use_bandwidth),
                             int(dc_ignore_bins), int(trigger_count))

    # -- single stages ------------------------------------------------------
    def convert(self, raw):
        """raw: int16 [n,2] / int16 [2,n] planar / int8 [n,2] / complex64 [n] -> complex64 [n]."""
        n = self.n
        out = np.empty(n, np.complex64)
        L = lib()
        if self.kind == KIND_FLOAT_COMPLEX:
            return np.ascontiguousarray(raw, np.complex64).copy()
        raw = np.ascontiguousarray(raw)
        if self.kind == KIND_SHORT_COMPLEX:
            assert raw.dtype == np.int16 and raw.size == 2 * n
            L.scn_oracle_short_complex_to_float_complex(_p(raw), _p(out), n, self.enob, self.correct_dc)
        elif self.kind == KIND_SHORT:
            assert raw.dtype == np.int16 and raw.size == 2 * n
            flat = raw.reshape(-1)
            L.scn_oracle_short_planar_to_float_complex(
                _p(flat), C.c_void_p(flat.ctypes.data + 2 * n), _p(out), n, self.enob, self.correct_dc)
        elif self.kind == KIND_BYTE_COMPLEX:
            assert raw.dtype == np.int8 and raw.size == 2 * n
            L.scn_oracle_byte_complex_to_float_complex(_p(raw), _p(out), n, self.enob, self.correct_dc)
        else:
            raise ValueError(self.kind)
        return out

    def window(self):
        w = np.empty(self.n, np.float32)
        lib().scn_oracle_window_blackman_harris(_p(w), self.n)
        return w

    def fft(self, x):
        x = np.ascontiguousarray(x, np.complex64)
        out = np.empty(self.n, np.complex64)
        L = lib()
        f = L.scn_oracle_fft_create(self.n)
        assert f, "oracle FFT needs n >= 2"
        L.scn_oracle_fft_process(f, _p(out), _p(x))
        L.scn_oracle_fft_destroy(f)
        return out

    def magnitude(self, X, use_log2f=False):
        X = np.ascontiguousarray(X, np.complex64)
        out = np.empty(self.n, np.float32)
        lib().scn_oracle_complex_to_magnitude(_p(X), _p(out), self.n, int(use_log2f))
        return out

    def time_domain(self, x, threshold=None):
        x = np.ascontiguousarray(x, np.complex64)
        mx, mn = C.c_float(), C.c_float()
        thr = self.params.threshold if threshold is None else threshold
        r = lib().scn_oracle_time_domain(_p(x), self.n, thr, C.byref(mx), C.byref(mn))
        return bool(r), mx.value, mn.value

    # -- whole path ---------------------------------------------------------
    def run(self, raw, center_freqs=None, seq_ids=None, want_power=True, want_hits=True, threads=1):
        """raw: array holding n_buffers buffers back to back in the kind's wire format.
        Returns (power_db [B,n] or None, hits (HIT_DTYPE) or None, trigger uint8[B])."""
        raw = np.ascontiguousarray(raw)
        per = raw_bytes_per_sample(self.kind) * self.n
        assert raw.nbytes % per == 0, (raw.nbytes, per)
        nb = raw.nbytes // per
        fc = np.zeros(nb, np.float64) if center_freqs is None else np.ascontiguousarray(center_freqs, np.float64)
        seq = np.arange(nb, dtype=np.uint64) if seq_ids is None else np.ascontiguousarray(seq_ids, np.uint64)
        assert fc.size == nb and seq.size == nb
        power = np.empty((nb, self.n), np.float32) if want_power else None
        trig = np.zeros(nb, np.uint8)
        cap = nb * self.n if want_hits else 0
        hits = np.zeros(cap, HIT_DTYPE) if want_hits else None
        total = lib().scn_oracle_run_batch(
            C.byref(self.params), self.kind, self.enob, int(self.correct_dc), _p(raw), nb, _p(fc), _p(seq),
            _p(power), _p(hits), cap, _p(trig), threads)
        if want_hits:
            hits = hits[:total].copy()
        return power, hits, trig


def frequency_table(sample_rate, start, stop, use_bandwidth=0.75, dc_ignore_width=0.0):
    """frequencyTable.cpp:9-37"""
    L = lib()
    cnt = L.scn_oracle_frequency_table(sample_rate, start, stop, use_bandwidth, dc_ignore_width, None, 0)
    out = np.empty(cnt, np.float64)
    L.scn_oracle_frequency_table(sample_rate, start, stop, use_bandwidth, dc_ignore_width, _p(out), cnt)
    return out


def set_direct_dft(direct):
    """Lengths that are not powers of two: False (default) = the DFT sum factored over the prime factors of n, in double;
    True = the sum as written, O(n^2), in double.  Applies to Oracle objects created afterwards."""
    lib().scn_oracle_set_direct_dft(1 if direct else 0)


WIN_HAMMING, WIN_HANN, WIN_BLACKMAN, WIN_RECTANGULAR, WIN_KAISER, WIN_BLACKMAN_HARRIS, WIN_BARTLETT, WIN_FLATTOP = range(8)


def window(win_type, n):
    """gr::fft::window::build(win_type, n, 0.0) as process.cpp:18 calls it ([3P], published definitions), float32 [n]."""
    w = np.empty(n, np.float32)
    assert lib().scn_oracle_window(int(win_type), _p(w), n) == 0, win_type
    return w


def ref64_window_of(win_type, n):
    """float64 evaluation of the same published definitions (independent of the C code: numpy)."""
    x = np.arange(n, dtype=np.float64) / (n - 1)
    cs = {WIN_HAMMING: (0.54, 0.46), WIN_HANN: (0.5, 0.5), WIN_BLACKMAN: (0.42, 0.5, 0.08),
          WIN_BLACKMAN_HARRIS: (0.35875, 0.48829, 0.14128, 0.01168),
          WIN_FLATTOP: tuple(c / 4.63867 for c in (1.0, 1.93, 1.29, 0.388, 0.028))}
    if win_type in (WIN_RECTANGULAR, WIN_KAISER):   # Kaiser with beta = 0.0: I0(0) / I0(0)
        return np.ones(n)
    if win_type == WIN_BARTLETT:
        i = np.arange(n, dtype=np.float64)
        return np.where(i < n // 2, 2 * i / (n - 1), 2 - 2 * i / (n - 1))
    return sum(((-1) ** k) * c * np.cos(2 * np.pi * k * x) for k, c in enumerate(cs[win_type]))


class window_type:
    """`with oracle.window_type(t):` -- run_batch / welch build window t inside (process-wide switch, reset on the way out)."""

    def __init__(self, t):
        self.t = int(t)

    def __enter__(self):
        assert lib().scn_oracle_set_window_type(self.t) == 0, self.t
        return self

    def __exit__(self, *exc):
        lib().scn_oracle_set_window_type(WIN_BLACKMAN_HARRIS)
        return False


class direct_dft:
    """`with oracle.direct_dft():` -- plans created inside evaluate the DFT sum as written, O(n^2); the process-wide switch is back
    at the factored form on the way out, also when the body raises (a test that failed half way used to leave it on)."""

    def __enter__(self):
        set_direct_dft(True)
        return self

    def __exit__(self, *exc):
        set_direct_dft(False)
        return False


def set_fft_mode(accurate):
    """True (default): double-internal FFT rounded to float (FFTW's accuracy class, the parity reference);
    False: textbook float radix-2 (the timed cpu_baseline, a second opinion)."""
    lib().scn_oracle_set_fft_mode(1 if accurate else 0)


def hackrf_interpolate(transfer_u8, scan_offset=0):
    """hackRFSource.cpp:186-222.  Returns (patched copy of the transfer, centre frequency, mismatches)."""
    buf = np.array(transfer_u8, dtype=np.uint8, copy=True)
    mism = C.c_uint32()
    fc = lib().scn_oracle_hackrf_interpolate(_p(buf), buf.size, int(scan_offset), C.byref(mism))
    return buf, fc, mism.value


# ---------------------------------------------------------------------------
# float64 mathematics (pins the third-party arithmetic)
# ---------------------------------------------------------------------------

def ref64_window(n):
    """4-term Blackman-Harris, symmetric, float64 ([3P] gr::fft::window::build)."""
    k = np.arange(n, dtype=np.float64) / (n - 1)
    return (0.35875 - 0.48829 * np.cos(2 * np.pi * k) + 0.14128 * np.cos(4 * np.pi * k)
            - 0.01168 * np.cos(6 * np.pi * k))


def ref64_spectrum(x_c64, window_f32):
    """float64 evaluation of window -> DFT -> power for float32 inputs.
    x_c64: [B,n] complex64 (already converted), window_f32: [n] float32.
    Returns (X complex128 [B,n], linear power [B,n], dB = 5*log10(power))."""
    x = np.asarray(x_c64).astype(np.complex128) * np.asarray(window_f32).astype(np.float64)
    X = np.fft.fft(x, axis=-1)
    P = X.real ** 2 + X.imag ** 2
    with np.errstate(divide="ignore"):
        dB = 5.0 * np.log10(P)
    return X, P, dB


def welch(x_c64, n=65536, k=16, n_psd=1):
    """BASELINE C5: float32 Welch PSD (dB) of a complex64 stream; see scn_oracle_welch."""
    x = np.ascontiguousarray(x_c64, np.complex64)
    assert x.size >= (n_psd * k + 1) * (n // 2)
    out = np.empty((n_psd, n), np.float32)
    lib().scn_oracle_welch(_p(x), n, k, n_psd, _p(out))
    return out


def welch_convert(raw, kind, enob, correct_dc, hop):
    """K1 of a Welch stream: the raw stream is a run of DELIVERY BLOCKS of `hop` samples (one AppendSamples call each,
    messageQueue.h:190-237), each converted by the reference's converter for `kind` (utility.cpp:9-84; with correct_dc the
    integer mean removed is the block's own).  Returns the complex64 stream."""
    if kind == KIND_FLOAT_COMPLEX:
        return np.ascontiguousarray(raw).view(np.complex64).reshape(-1).copy()
    raw = np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
    per = raw_bytes_per_sample(kind) * hop
    assert raw.size % per == 0, (raw.size, per)
    conv = Oracle(hop, kind=kind, enob=enob, correct_dc=correct_dc)
    dt = np.int8 if kind == KIND_BYTE_COMPLEX else np.int16
    return np.concatenate([conv.convert(raw[b * per:(b + 1) * per].view(dt)) for b in range(raw.size // per)])


def welch_raw(raw, kind, enob, correct_dc, n=65536, k=16, n_psd=1):
    """BASELINE C5 on a wire-format stream: welch_convert (K1 per delivery block of n/2 samples) then welch."""
    return welch(welch_convert(raw, kind, enob, correct_dc, n // 2), n, k, n_psd)


def ref64_welch(x_c64, window_f32, n=65536, k=16, n_psd=1):
    """float64 evaluation of the same definition."""
    x = np.asarray(x_c64).astype(np.complex128)
    w = np.asarray(window_f32).astype(np.float64)
    hop = n // 2
    out = np.empty((n_psd, n))
    for p in range(n_psd):
        acc = np.zeros(n)
        for s in range(k):
            seg = x[(p * k + s) * hop:(p * k + s) * hop + n] * w
            X = np.fft.fft(seg)
            acc += X.real ** 2 + X.imag ** 2
        with np.errstate(divide="ignore"):
            out[p] = 5.0 * np.log10(acc / k)
    return out
