"""The parity bar, written down once (BASELINE.json north_star: "power spectra within 1e-5
relative float tolerance, bin indices of detected peaks bit-exact").

float32 FFTs of different butterfly order differ by ~1e-7 of the buffer's RMS level in every
bin, so a bin far below the mean (a Rayleigh-small bin) has an unbounded *per-bin* relative
error even between two correct implementations (complex64 vs
complex128 FFT already shows per-bin max 1.9e-4).  "1e-5 relative" is therefore taken
relative to max(P_bin, mean P of the buffer) on linear power for EVERY bin, and on the dB
output as 1e-5*|dB| + 1e-4 for every bin at or above the buffer's mean power -- the bins a
threshold detector can report (a 1e-4 dB step is a 4.6e-5 relative power step; the ~1e-6
of the buffer's RMS level that two float32 FFTs differ by exceeds that on bins 20 dB down).
The strict per-bin figure is computed and reported by the tests too.

SAID PLAINLY, because a green suite could be misread otherwise: the 1e-5 of north_star is met in the sense "relative to
max(bin, buffer mean)".  The dB criterion is only ASSERTED on bins at or above the buffer's mean power; over ALL bins the
same runs show max dB errors of ~1.3e-3 dB and strict per-bin relative power errors of ~6e-4 (the keys
max_db_err_all_bins / strict_per_bin_rel_power_max of compare_spectra's result, printed by smoke()): that is the float32
noise floor on Rayleigh-small bins, present between any two float32 FFTs (the reference's FFTW included), not an
implementation error -- but it is not "1e-5 per bin" either.  And the oracle these figures are taken against is itself pinned
by float64 mathematics only, not by the reference's own output (parity unpinned).
"""
import numpy as np

REL_POWER = 1e-5
DB_REL = 1e-5
DB_ABS = 1e-4
DB_MIN_POWER_RATIO = 1.0     # dB criterion applies to bins with P >= this * mean(P)
GUARD_DB = 1e-3              # no evaluated bin may sit this close to the threshold


def db_to_power(db):
    """inverse of dB = 5*log10(P)  (utility.cpp:86-98 computes 10*log10 |X|)"""
    return np.power(10.0, np.asarray(db, np.float64) / 5.0)


def compare_spectra(db_test, db_ref):
    """Returns dict of error figures; raises AssertionError when outside the bar."""
    db_test = np.asarray(db_test, np.float64)
    db_ref = np.asarray(db_ref, np.float64)
    assert db_test.shape == db_ref.shape
    # NaN and +inf must sit in the same places.  -inf is an ordinary value of the map (a bin of exactly zero
    # power, utility.cpp:95): it enters the linear-power criterion as P = 0, so "-inf here, finite there" passes
    # only where the finite side is itself below 1e-5 of the buffer's mean power (cancellation residue -- e.g.
    # the constant 2e6-sized samples the negative-sum DC quirk produces leave bins of 0 vs 0.25 beside a 1e14
    # mean), and fails as a dB error wherever the reference bin is at or above the mean.
    bad_r = np.isnan(db_ref) | (db_ref == np.inf)
    bad_t = np.isnan(db_test) | (db_test == np.inf)
    assert np.array_equal(bad_r, bad_t), "nan / +inf pattern differs"
    ok_r, ok_t = np.isfinite(db_ref), np.isfinite(db_test)
    finite = ok_r & ok_t
    P_t = np.where(ok_t, db_to_power(np.where(ok_t, db_test, 0)), 0.0)
    P_r = np.where(ok_r, db_to_power(np.where(ok_r, db_ref, 0)), 0.0)
    mean = P_r.mean(axis=-1, keepdims=True)
    scale = np.maximum(P_r, mean)
    lin = np.where(bad_r, 0.0, np.abs(P_t - P_r) / np.where(scale > 0, scale, 1.0))
    big = ok_r & (P_r >= DB_MIN_POWER_RATIO * mean) & (mean > 0)
    with np.errstate(invalid="ignore"):
        db_err = np.where(ok_r & ok_t, np.abs(db_test - db_ref), np.where(ok_r == ok_t, 0.0, np.inf))
    db_bar = DB_REL * np.abs(db_ref) + DB_ABS
    with np.errstate(divide="ignore", invalid="ignore"):
        strict = np.where(P_r > 0, np.abs(P_t - P_r) / P_r, 0.0)
    out = {
        "max_rel_power_vs_max_bin_mean": float(lin.max()),
        "max_db_err_big_bins": float(db_err[big].max()) if big.any() else 0.0,
        "max_db_err_all_bins": float(db_err[finite].max()) if finite.any() else 0.0,
        "strict_per_bin_rel_power_max": float(strict.max()),
        "strict_per_bin_rel_power_p9999": float(np.quantile(strict, 0.9999)),
    }
    assert out["max_rel_power_vs_max_bin_mean"] <= REL_POWER, out
    assert np.all(db_err[big] <= db_bar[big]), out
    return out


def flip_unsafe(db_ref, threshold, margin=4.0):
    """Boolean mask (shape of db_ref): bins whose side of `threshold` the spectrum tolerance itself could change --
    |P_ref - P_thr| <= margin * REL_POWER * max(P_ref, mean P), or within GUARD_DB of the threshold in dB.  Hit
    lists are demanded bit-exact everywhere else.  (A buffer dominated by one huge component -- e.g. the 2e6-sized
    offset the negative-sum DC quirk of utility.cpp:77-78 adds -- leaves its other bins as cancellation residue far
    below the mean: two correct float32 FFTs then disagree by whole dB there.)"""
    db_ref = np.asarray(db_ref, np.float64)
    ok = np.isfinite(db_ref)
    P = np.where(ok, db_to_power(np.where(ok, db_ref, 0)), 0.0)
    mean = P.mean(axis=-1, keepdims=True)
    p_thr = db_to_power(threshold)
    lin = np.abs(P - p_thr) <= margin * REL_POWER * np.maximum(P, mean)
    with np.errstate(invalid="ignore"):
        near = np.abs(db_ref - threshold) < GUARD_DB
    return lin | near


def evaluated_mask(n, use_bandwidth=0.75, dc_ignore_bins=4):
    """Boolean [n] over natural bin j: True where process.cpp:46-52 evaluates the bin."""
    half = n // 2
    use_window = int(use_bandwidth * n / 2.0)
    j = np.arange(n)
    i = (j + half) % n
    skip = (j < dc_ignore_bins) | ((n - j) < dc_ignore_bins)
    skip |= (i < (half - use_window)) | (i > (half + use_window))
    return ~skip


def pick_threshold(db_ref, n, start, use_bandwidth=0.75, dc_ignore_bins=4):
    """Smallest threshold >= start (steps of 0.01 dB) with an empty guard band on the
    reference spectrum, so bit-exact hit indices are a fair demand."""
    m = evaluated_mask(n, use_bandwidth, dc_ignore_bins)
    vals = np.asarray(db_ref, np.float64)[..., m].ravel()
    vals = np.sort(vals[np.isfinite(vals)])   # (sorted once: a step is two binary searches, also on 5 M bins)
    thr = np.float32(start)
    for _ in range(10000):
        lo = np.searchsorted(vals, float(thr) - GUARD_DB, side="right")    # first value > thr - GUARD
        hi = np.searchsorted(vals, float(thr) + GUARD_DB, side="left")     # first value >= thr + GUARD
        if lo >= hi:
            return float(thr)
        thr = np.float32(thr + np.float32(0.01))
    raise AssertionError("no guard-band-free threshold found")
