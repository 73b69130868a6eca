"""qnnpack_amd -- MI355X (gfx950) native build of QNNPACK's uint8 conv/GEMM hot path.

The product is the C-ABI shared library ``libqnnpack_gfx950.so`` (plain-C host
code + hand-written HIP kernels, sources under ``qnnpack_amd/csrc``) exporting
the reference's ``include/qnnpack.h`` entry points. This Python package is only
the thin ctypes loader used by tests, ``bench.py`` and ``__graft_entry__.py``.

There is no CPU fallback anywhere in this package: if the library has not been
built, or no gfx950 device is usable, loading / ``qnnp_initialize`` fails loudly.
"""
from __future__ import annotations

import os
import subprocess
import sys

from .binding import Gfx950Library, QnnpackError, QnnpackLibrary, Status, address_of

__all__ = [
    "Gfx950Library", "QnnpackLibrary", "QnnpackError", "Status", "address_of",
    "library_path", "debug_library_path", "build", "load", "load_debug",
]

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libqnnpack_gfx950.so"
_DBG_NAME = "libqnnpack_gfx950_dbg.so"
_loaded = None
_loaded_debug = None


def library_path() -> str:
    # QNNP_GFX950_LIBRARY: measurement aid -- load another build of the library (an `ABLATION=1` build with in-kernel
    # cycle stamps, an older build for a same-box A/B) in place of the product
    return os.environ.get("QNNP_GFX950_LIBRARY") or os.path.join(_PKG_DIR, _LIB_NAME)


def debug_library_path() -> str:
    """The measurement companion: debug hooks (create-time host logic for the CPU test tier) and the bare-MFMA
    probe. Never linked by the product; tests and bench.py load it beside it."""
    return os.path.join(_PKG_DIR, _DBG_NAME)


def build(verbose: bool = False) -> str:
    """Compile the C host code and the gfx950 HIP kernels in-tree (needs hipcc, no GPU)."""
    cmd = ["make", "-C", os.path.join(_PKG_DIR, "csrc"), "-j8"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError(f"building {_LIB_NAME} failed (exit {res.returncode})")
    return library_path()


def load() -> Gfx950Library:
    """Load the built product library (once per process).

    If torch is already imported its bundled HIP runtime (same SONAME) is
    resident and the library binds to it, so torch device tensors and this
    library share one HIP context. Otherwise the ROCm runtime from the library's
    rpath (/opt/rocm/lib) is used.
    """
    global _loaded
    if _loaded is None:
        try:
            # Load torch's bundled HIP runtime FIRST when torch is installed: both it and
            # /opt/rocm's copy have SONAME libamdhip64.so.7, and two HIP runtimes in one
            # process cannot see each other's allocations or streams.
            import torch  # noqa: F401
        except Exception:
            pass
        path = library_path()
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: the HIP extension has not been built. "
                "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C qnnpack_amd/csrc`. "
                "There is no CPU fallback.")
        _loaded = Gfx950Library(path)
    return _loaded


def load_debug():
    """ctypes handle of libqnnpack_gfx950_dbg.so (once per process). Wrapped so that callers written against
    a library object (`.lib.<symbol>`) work unchanged."""
    global _loaded_debug
    if _loaded_debug is None:
        import ctypes
        try:
            import torch  # noqa: F401  -- same HIP runtime as the product (see load())
        except Exception:
            pass
        path = debug_library_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C qnnpack_amd/csrc`.")

        class _Debug:
            pass
        dbg = _Debug()
        dbg.path = path
        dbg.lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        dbg.lib.qnnp_gfx950_mfma_probe.restype = ctypes.c_int
        dbg.lib.qnnp_gfx950_mfma_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]

        def mfma_probe(random_operands: bool, iters: int = 12800) -> float:
            tops = ctypes.c_float(0.0)
            rc = dbg.lib.qnnp_gfx950_mfma_probe(1 if random_operands else 0, iters, ctypes.byref(tops))
            if rc != 0:
                raise RuntimeError(f"qnnp_gfx950_mfma_probe -> {rc}")
            return float(tops.value)
        dbg.mfma_probe = mfma_probe
        dbg.lib.qnnp_gfx950_copy_probe.restype = ctypes.c_int
        dbg.lib.qnnp_gfx950_copy_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]

        def copy_probe(read_only: bool = False, mbytes: int = 1024, reps: int = 5, streaming: bool = False) -> float:
            """GB/s of a 16-byte-per-lane streaming kernel on this chip (copy: bytes read + written); streaming = the nt
            policy on its loads and stores, the form the product's whole-line writers use."""
            gbs = ctypes.c_float(0.0)
            rc = dbg.lib.qnnp_gfx950_copy_probe(1 if read_only else (2 if streaming else 0), mbytes, reps, ctypes.byref(gbs))
            if rc != 0:
                raise RuntimeError(f"qnnp_gfx950_copy_probe -> {rc}")
            return float(gbs.value)
        dbg.copy_probe = copy_probe
        dbg.lib.qnnp_gfx950_launch_floor_probe.restype = ctypes.c_int
        dbg.lib.qnnp_gfx950_launch_floor_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]

        def launch_floor_probe(kernels: int = 31, blocks: int = 256, replays: int = 20) -> float:
            """microseconds per launch of a hipGraph of `kernels` dependent EMPTY kernels (hip/mfma_probe.hip)"""
            us = ctypes.c_float(0.0)
            rc = dbg.lib.qnnp_gfx950_launch_floor_probe(kernels, blocks, replays, ctypes.byref(us))
            if rc != 0:
                raise RuntimeError(f"qnnp_gfx950_launch_floor_probe -> {rc}")
            return float(us.value)
        dbg.launch_floor_probe = launch_floor_probe
        _loaded_debug = dbg
    return _loaded_debug
