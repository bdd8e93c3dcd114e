"""Golden vectors for the frequency table, produced by the REFERENCE's own frequencyTable.cpp (built from
/root/reference by `make -C oracle ref`; run this only in the container that has the reference):
    python tests/golden/make_frequency_table_ref.py  ->  tests/golden/frequency_table_ref.npz
Inputs and expected outputs only -- no reference source is stored."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

# (sample_rate, start, stop, use_bandwidth, dc_ignore_width): CLI defaults of scan.cpp:60-106 and the shapes the tests use
CASES = [
    (8000000, 88e6, 108e6, 0.75, 0.0),
    (8000000, 88e6, 130e6, 0.75, 0.0),
    (8000000, 2400e6, 2440e6, 0.75, 0.0),
    (8000000, 100e6, 6100e6, 0.75, 0.0),        # C4-sized sweep: 1000 centres
    (8000000, 3e6, 98307e6, 0.75, 0.0),         # 16384 centres
    (20000000, 70e6, 6000e6, 0.75, 0.0),
    (2400000, 24e6, 1766e6, 0.5, 0.0),
    (8000000, 88e6, 108e6, 0.75, 0.25),         # DC-ignore stepping: step = (use - dc)/2
    (10000000, 400e6, 470e6, 0.8, 0.1),
    (8000000, 433.92e6, 0.0, 0.75, 0.0),        # stop == 0: a single centre
    (8000000, 88e6, 88.5e6, 0.75, 0.0),         # stop inside the first step: empty table? (count 0 or 1 -- whatever the reference says)
    (12500000, 1e9, 1.1e9, 0.75, 0.0),
]


def main():
    assert oracle.ref_lib() is not None, "build oracle/_ref first: make -C oracle ref"
    out = {"cases": np.array(CASES, dtype=np.float64)}
    for k, (fs, start, stop, bw, dc) in enumerate(CASES):
        tab = oracle.ref_frequency_table(fs, start, stop, bw, dc)
        out[f"table_{k}"] = tab
        if len(tab):
            steps = min(3 * len(tab) + 2, 300)
            f, it, ss = oracle.ref_frequency_walk(fs, start, stop, bw, dc, steps)
            out[f"walk_f_{k}"], out[f"walk_it_{k}"], out[f"walk_ss_{k}"] = f, it, ss
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frequency_table_ref.npz"), **out)
    print("wrote", len(CASES), "cases")


if __name__ == "__main__":
    main()
