#!/usr/bin/env python3
"""Writes tests/golden/<case>.npz: the CPU oracle's outputs for every case of tests/golden_cases.py (VERDICT r2 next #2a).

Run in the BUILD container (no GPU needed):   python tools/gen_golden.py [case ...]
The vectors are made by oracle/ (this repository's CPU restatement of the reference's algorithm -- the reference itself cannot be
built here: no OpenCV, no Eigen), so they FREEZE the restatement; they do not pin it to a run of the reference.
Regenerating a fixture is a reviewed act: a change in any *.npz must come with the reason in the commit message."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_cases as GC  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    O.build()
    api = GC.OracleApi(O)
    names = sys.argv[1:] or list(GC.CASES)
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    total = 0
    for name in names:
        res = GC.CASES[name](api)
        path = os.path.join(out_dir, name + ".npz")
        np.savez_compressed(path, **{k: np.asarray(v) for k, v in res.items()})
        sz = os.path.getsize(path)
        total += sz
        print("%-24s %3d arrays %8d bytes" % (name, len(res), sz))
    print("total %d bytes" % total)


if __name__ == "__main__":
    main()
