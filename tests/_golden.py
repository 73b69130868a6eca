"""Loader for tests/golden/reference_outputs.npz (outputs of the compiled reference, see generate_golden.py)."""
import os

import numpy as np

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs.npz")
_cache = None


def golden():
    global _cache
    if _cache is None:
        _cache = dict(np.load(_PATH))
    return _cache


def names(kind: str):
    return sorted({k.split("/")[1] for k in golden() if k.startswith(kind + "/")})


def entry(kind: str, name: str):
    g = golden()
    key = f"{kind}/{name}"
    scale, zp = g[key + "/quant"]
    return g[key + "/input"], g[key + "/kernel"], g[key + "/bias"], (np.float32(scale), int(zp)), g[key + "/output"]
