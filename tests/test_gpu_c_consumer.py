"""GPU tier: a C translation unit compiled against the REFERENCE's own include/qnnpack.h (tests/c_consumer/consumer.c,
built by oracle/Makefile into oracle/_ref/ref_header_consumer while /root/reference was present) and linked to the
product library runs the reference benchmark driver's flow (bench/convolution.cc:59-98) with host tensors; its outputs
must equal the scalar oracle byte for byte. ctypes proves the ABI; this proves the header half of the drop-in claim
(INTEGRATION.md section 1)."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import o1

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_header_consumer")
CASES = ["conv3x3", "conv3x3s2_rgb", "dw3x3", "grouped1x1", "fc_1x1024x1000", "fc_37x200x96"]


def test_consumer_was_built_against_the_reference_header():
    assert os.path.exists(EXE), (f"{EXE} is missing: build it where /root/reference exists "
                                 "(`make -C oracle consumer`, part of __graft_entry__.build())")
    # linked to the product, not to the compiled reference
    needed = subprocess.run(["readelf", "-d", EXE], capture_output=True, text=True).stdout
    assert "libqnnpack_gfx950.so" in needed and "libqnnpack_ref" not in needed


@pytest.mark.parametrize("name", CASES)
def test_reference_header_caller_matches_oracle(name, tmp_path):
    assert os.path.exists(EXE), f"{EXE} is missing (see test_consumer_was_built_against_the_reference_header)"
    run = subprocess.run([EXE, name, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    desc = dict(re.findall(r"(\w+)=([\w.x+-]+)", run.stdout))
    fc = int(desc["fc"])
    groups, gic, goc, batch = int(desc["groups"]), int(desc["gic"]), int(desc["goc"]), int(desc["batch"])
    kh, kw = (int(v) for v in desc["k"].split("x"))
    h, w = (int(v) for v in desc["in"].split("x"))
    pad, stride, dilation = int(desc["pad"]), int(desc["stride"]), int(desc["dilation"])
    izp, kzp, ozp = int(desc["izp"]), int(desc["kzp"]), int(desc["ozp"])
    qmin, qmax = int(desc["qmin"]), int(desc["qmax"])
    scale = np.float32(float(desc["scale"]))
    inp = np.fromfile(tmp_path / f"{name}.in", dtype=np.uint8)
    kernel = np.fromfile(tmp_path / f"{name}.kernel", dtype=np.uint8)
    bias = np.fromfile(tmp_path / f"{name}.bias", dtype=np.int32)
    got = np.fromfile(tmp_path / f"{name}.out", dtype=np.uint8)
    cout = groups * goc
    if fc:
        acc = o1.gemm_acc(inp.reshape(batch, gic), kernel.reshape(goc, gic), bias, izp, kzp)
        rows = batch
    else:
        shape = o1.conv_shape(batch, h, w, (pad, pad, pad, pad), (kh, kw), (stride, stride), (dilation, dilation),
                              groups, gic, goc, groups * gic)
        oh, ow = o1.conv_output_hw(shape)
        acc = o1.conv2d_acc(shape, inp, kernel.reshape(groups, goc, kh, kw, gic), bias, izp, kzp)
        rows = batch * oh * ow
    expected = np.zeros(rows * cout, np.uint8)
    o1.requantize_rows(acc.reshape(rows, cout), scale, ozp, qmin, qmax, expected, cout)
    assert got.size == expected.size
    bad = np.nonzero(got != expected)[0]
    assert bad.size == 0, f"{name}: {bad.size} of {got.size} bytes differ, first at {int(bad[0])}"
