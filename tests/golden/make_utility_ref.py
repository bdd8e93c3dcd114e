"""Generates tests/golden/utility_ref.npz from the REFERENCE's own utility.cpp (oracle/_ref/libref_utility.so, built by
`make -C oracle ref` in the container that has /root/reference): inputs and the reference's outputs for the three
integer -> complex-float converters (utility.cpp:9-84) with every ENOB the front-ends use and both DC settings --
including the two quirks (the scale's sign flip when ENOB equals the integer width, the int32 /= uint32 "mean" of a
negative sum) -- and for the dB map (utility.cpp:86-98).  Data only: inputs and expected outputs.
    python tests/golden/make_utility_ref.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def cases():
    """(name, kind, n, enob, correct_dc, raw)"""
    rng = np.random.default_rng(20260101)
    out = []
    for n in (64, 1000):
        for enob in (10, 12, 14, 16):
            for dc in (False, True):
                full = 1 << (min(enob, 15) - 1)
                iq = rng.integers(-full, full, size=(n, 2)).astype(np.int16)
                out.append((f"s16c_n{n}_e{enob}_dc{int(dc)}", O.KIND_SHORT_COMPLEX, n, enob, dc, iq))
                out.append((f"s16p_n{n}_e{enob}_dc{int(dc)}", O.KIND_SHORT, n, enob, dc, np.ascontiguousarray(iq.T)))
        for dc in (False, True):
            b = rng.integers(-128, 128, size=(n, 2)).astype(np.int8)
            out.append((f"s8c_n{n}_e8_dc{int(dc)}", O.KIND_BYTE_COMPLEX, n, 8, dc, b))
    # means of either sign, extremes, a constant buffer
    neg = (rng.integers(-300, 100, size=(256, 2)) - 50).astype(np.int16)
    out.append(("s16c_negmean_dc1", O.KIND_SHORT_COMPLEX, 256, 12, True, neg))
    out.append(("s16p_negmean_dc1", O.KIND_SHORT, 256, 12, True, np.ascontiguousarray(neg.T)))
    out.append(("s8c_negmean_dc1", O.KIND_BYTE_COMPLEX, 256, 8, True, (neg // 4).astype(np.int8)))
    ext = np.array([[32767, -32768], [-32768, 32767], [0, 0], [1, -1]] * 16, np.int16)
    out.append(("s16c_extremes_e16_dc0", O.KIND_SHORT_COMPLEX, 64, 16, False, ext))
    out.append(("s16c_extremes_e12_dc1", O.KIND_SHORT_COMPLEX, 64, 12, True, ext))
    out.append(("s8c_extremes_dc1", O.KIND_BYTE_COMPLEX, 64, 8, True, np.array([[127, -128], [-128, 127], [0, 0], [1, -1]] * 16, np.int8)))
    out.append(("s16c_const_dc1", O.KIND_SHORT_COMPLEX, 32, 12, True, np.full((32, 2), 77, np.int16)))
    return out


def spectra():
    rng = np.random.default_rng(7)
    x = (rng.standard_normal(4096) + 1j * rng.standard_normal(4096)).astype(np.complex64)
    scales = np.float32(10.0) ** rng.integers(-12, 9, size=4096).astype(np.float32)
    x = (x * scales).astype(np.complex64)
    x[:6] = [0, 1, 1j, -1, 3e-39 + 0j, 1e19 + 1e19j]  # zero (-> -inf), units, a denormal, near-overflow of re*re
    return x


def main():
    assert O.ref_dsp_available(), "build oracle/_ref first: make -C oracle ref (needs /root/reference)"
    d = {}
    names = []
    for name, kind, n, enob, dc, raw in cases():
        names.append(name)
        d[name + "_in"] = raw
        d[name + "_meta"] = np.array([kind, n, enob, int(dc)], np.int64)
        d[name + "_out"] = O.ref_convert(kind, raw, n, enob, dc)
    d["names"] = np.array(names)
    X = spectra()
    d["mag_in"] = X
    with np.errstate(all="ignore"):
        d["mag_out"] = O.ref_magnitude(X)
    np.savez_compressed(os.path.join(HERE, "utility_ref.npz"), **d)
    print("wrote", os.path.join(HERE, "utility_ref.npz"), len(names), "converter cases")


if __name__ == "__main__":
    main()
