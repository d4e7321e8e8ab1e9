"""Writes tests/golden/resize_np.npz: outputs of oracle/resize_np.py (the independent numpy restatement of torchvision 0.10's tensor
Resize) for the three cases VERDICT r4 #9 names -- 128 -> 224 (BASELINE configs[0]: up-scale), 900 -> 224 (the reference's default
crops: 4x down-scale without antialias) and 64 x 150 -> 204 x 480 (the max_size cap, non-square).  Inputs are regenerated from
``splice_amd.synth.uniform`` seeds by the tests (not stored); one channel each keeps the file small.
    python oracle/make_resize_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import resize_np  # noqa: E402
from splice_amd import synth  # noqa: E402

CASES = {"up128": ((1, 128, 128), 224), "down900": ((1, 900, 900), 224), "cap64x150": ((1, 64, 150), 224)}


def case_input(name):
    return synth.uniform(2024, "resize/" + name, CASES[name][0])


if __name__ == "__main__":
    out = {}
    for name, (shape, size) in CASES.items():
        y = resize_np.resize_shorter_edge(case_input(name), size, 480)
        out[name] = y
        print(name, shape, "->", y.shape, float(y.mean()))
    assert out["cap64x150"].shape == (1, 204, 480) and out["down900"].shape == (1, 224, 224)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "resize_np.npz"), **out)
