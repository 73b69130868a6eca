#!/usr/bin/env python3
"""Generate tests/golden/reference_pointwise_outputs.npz: outputs of the COMPILED REFERENCE
(oracle/_ref/libqnnpack_ref.so) for a spread of the add / global-average-pooling cases of tests/_pointwise.py.
Run in the build container:  python tests/golden/generate_golden_pointwise.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _pointwise as pw  # noqa: E402
from oracle import ref  # noqa: E402


def slack(buf):
    p = np.concatenate([np.zeros(32, np.uint8), buf, np.zeros(32, np.uint8)])
    return p[32:32 + buf.size]


def main():
    lib = ref.lib()
    blobs = {}
    adds = [c for c in pw.add_cases() if c.batch > 0]
    adds = adds[::9] + [c for c in adds if c.name.startswith("ax_")]
    for case in adds:
        a, b, _ = pw.add_tensors(case)
        out, _ = pw.add_run(lib, case, slack(a), slack(b))
        blobs[f"add/{case.name}/a"], blobs[f"add/{case.name}/b"], blobs[f"add/{case.name}/output"] = a, b, out
    gaps = [c for c in pw.gap_cases() if c.batch > 0]
    gaps = gaps[::37] + [c for c in gaps if c.name.startswith("gx_")]
    for case in gaps:
        inp = pw.gap_tensors(case)
        out, _ = pw.gap_run(lib, case, slack(inp))
        blobs[f"gap/{case.name}/input"], blobs[f"gap/{case.name}/output"] = inp, out
    path = os.path.join(HERE, "reference_pointwise_outputs.npz")
    np.savez_compressed(path, **blobs)
    print(f"{len(adds)} add + {len(gaps)} global-average-pooling cases ->", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
