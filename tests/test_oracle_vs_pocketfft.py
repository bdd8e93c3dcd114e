"""The oracle's window -> FFT -> dB chain against an INDEPENDENT single-precision FFT that is in this image: pocketfft, through
scipy.fft on complex64 input (scipy keeps complex64 in single precision).  The reference's own FFT is FFTW's float transform
(fft.cpp:4-25), which cannot be built here; pocketfft stands where any correct float32 FFT stands relative to the float64
truth.  Two things are checked on the bench's own synthetic input:
  * a float32 chain built from third-party pieces only (numpy float32 multiply, pocketfft complex64, float32 dB map) meets the
    bar of tests/tolerances.py against the oracle -- the bar the GPU path is held to is one a production float32 FFT passes;
  * the hit lists the two chains produce agree wherever no bin sits inside the guard band of the threshold.
This is a consistency check between independent implementations, not a pin to the reference's output."""
import numpy as np
import pytest
import scipy.fft

from scanner_amd import synth
from tests import tolerances as tol


def float32_chain(x, window):
    """numpy / pocketfft only: window multiply in float32 (process.cpp:28-34), forward unnormalised c2c FFT in single precision
    (fft.cpp:20-25), 10*log2(sqrt(re^2 + im^2))/log2(10) in float32 (utility.cpp:86-98)."""
    xw = (x * window[None, :].astype(np.float32)).astype(np.complex64)
    X = scipy.fft.fft(xw, axis=-1)
    assert X.dtype == np.complex64            # single precision all the way
    re, im = X.real.astype(np.float32), X.imag.astype(np.float32)
    mag = np.sqrt(re * re + im * im, dtype=np.float32)
    with np.errstate(divide="ignore"):
        return (np.float32(10.0) * np.log2(mag.astype(np.float64)) / np.log2(10.0)).astype(np.float32)


@pytest.mark.parametrize("n", [1024, 4096, 8192, 16384])
def test_independent_float32_fft_meets_the_same_bar(oracle_mod, n):
    O = oracle_mod
    nb = 24
    x = synth.cfloat_batch(n, nb, seed=11 + n)
    o = O.Oracle(n, 8000000, 1e9)
    p_ref, _, _ = o.run(x)
    p_pf = float32_chain(x, o.window())
    fig = tol.compare_spectra(p_pf, p_ref)    # raises outside the bar
    assert fig["max_rel_power_vs_max_bin_mean"] < tol.REL_POWER
    # detections: identical wherever the threshold is not within the guard band of some evaluated bin
    thr = tol.pick_threshold(p_ref, n, start=10.0)
    _, h_ref, _ = O.Oracle(n, 8000000, thr).run(x, 3e6 + 6e6 * np.arange(nb))
    keep = tol.evaluated_mask(n)
    i_ref = sorted((int(s), int(i)) for s, i in zip(h_ref["seq_id"], h_ref["i"]))
    jj = (np.arange(n) + n // 2) % n          # i -> j (process.cpp:47)
    i_pf = sorted((b, int(i)) for b in range(nb) for i in np.nonzero(keep[jj] & (p_pf[b][jj] > np.float32(thr)))[0])
    assert i_pf == i_ref and len(i_ref) > 0
