"""GPU tier: grouped 1x1 convolutions run as ONE dense GEMM (round 6; qnnpack_amd/csrc/convolution.c builds a [groups * GOC][groups * GIC]
weight matrix with the groups' blocks on its diagonal and the KERNEL ZERO POINT everywhere else, operator-run.c takes it from 65536 rows up;
"gemm_kernel" 31 forces it) against the scalar oracle of the GROUPED operator: every off-diagonal product is (a - izp) * (kzp - kzp) = 0, so
the bytes must be the grouped ones whatever the zero points. Reference path: q8gemm once per group under qnnp_run_operator
(src/operator-run.c:770-804); shapes: ShuffleNet v1's grouped 1x1 layers (bench/convolution.cc:108-330)."""
import numpy as np
import pytest

from _cases import ConvCase, conv_tensors
from _gpu import from_device, to_device
from qnnpack_amd.binding import QnnpackError
from _runner import FILL, assert_bytes_equal, conv_expected

pytestmark = pytest.mark.gpu

CASES = [
    ConvCase("d_g2_25_88", (7, 9), groups=2, gic=25, goc=88, batch=3),
    ConvCase("d_g3_20_72", (6, 5), groups=3, gic=20, goc=72, batch=2),
    ConvCase("d_g4_17_62", (9, 9), groups=4, gic=17, goc=62, batch=2),
    ConvCase("d_g4_68_17", (9, 9), groups=4, gic=68, goc=17, batch=2),
    ConvCase("d_g8_12_45", (11, 7), groups=8, gic=12, goc=45, batch=2),
    ConvCase("d_g8_48_12", (5, 5), groups=8, gic=48, goc=12, batch=4),
    ConvCase("d_g8_96_48_long_k", (4, 4), groups=8, gic=96, goc=48, batch=2),          # 768 -> 384
    ConvCase("d_g2_1_3", (5, 5), groups=2, gic=1, goc=3, batch=2),
    ConvCase("d_g7_3_5", (6, 6), groups=7, gic=3, goc=5, batch=3),
    ConvCase("d_g2_16_16_aligned", (8, 8), groups=2, gic=16, goc=16, batch=4),          # dense 32 -> 32: the streaming kernel's shape
    ConvCase("d_g4_16_64_aligned", (8, 8), groups=4, gic=16, goc=64, batch=4),          # dense 64 -> 256
    ConvCase("d_g4_strides", (6, 6), groups=4, gic=17, goc=62, batch=2, input_pixel_stride=75, output_pixel_stride=251),
    ConvCase("d_g8_zp_0_255", (6, 6), groups=8, gic=12, goc=45, batch=2, izp=0, kzp=255),
    ConvCase("d_g8_zp_255_0", (6, 6), groups=8, gic=12, goc=45, batch=2, izp=255, kzp=0),
    ConvCase("d_g3_zp_9_200", (6, 6), groups=3, gic=20, goc=72, batch=2, izp=9, kzp=200),
    ConvCase("d_g3_kzp128", (6, 6), groups=3, gic=20, goc=72, batch=2, kzp=128),
    ConvCase("d_g4_clamp", (6, 6), groups=4, gic=17, goc=62, batch=2, qmin=90, qmax=160),
    ConvCase("d_g2_512_512_widest", (3, 3), groups=2, gic=512, goc=512, batch=2),        # 1024 -> 1024: the widest image that is built
]


def _run(lib, case, code):
    expected, quant, out_hw = conv_expected(case)
    inp, kernel, bias = conv_tensors(case)
    oscale, ozp = quant
    oh, ow = out_hw
    cout = case.groups * case.goc
    rows = case.batch * oh * ow
    lib.set_option("gemm_kernel", code)          # (an operator keeps the code that is set when it is SET UP)
    try:
        op = lib.create_convolution2d_nhwc_q8(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, case.groups, case.gic, case.goc,
                                              case.izp, 1.0, case.kzp, 1.0, kernel, bias, ozp, float(oscale), case.qmin, case.qmax, 0)
        try:
            d_in = to_device(inp)
            d_out = to_device(np.full((rows - 1) * case.out_stride + cout, FILL, np.uint8))
            lib.setup_convolution2d_nhwc_q8(op, case.batch, case.input_size[0], case.input_size[1], d_in, case.in_stride, d_out, case.out_stride)
            lib.run_operator(op)
            return from_device(d_out), expected, lib.operator_ran_dense(op), lib.operator_kernel(op)
        finally:
            lib.delete_operator(op)
    finally:
        lib.set_option("gemm_kernel", 0)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_dense_image_gives_the_grouped_bytes(qnnp, case):
    out, expected, dense, kname = _run(qnnp, case, 31)
    assert dense, kname
    assert_bytes_equal(out, expected, f"gfx950 dense image on {kname} vs oracle of the grouped operator [{case.name}]")


def test_few_rows_keep_the_grouped_image_and_many_rows_take_the_dense_one(qnnp):
    small = ConvCase("d_auto_small", (14, 14), groups=8, gic=12, goc=45, batch=4)            # 784 rows
    out, expected, dense, kname = _run(qnnp, small, 0)
    assert not dense, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} (automatic, grouped) vs oracle")
    big = ConvCase("d_auto_big", (28, 28), groups=8, gic=12, goc=45, batch=84)                 # 65856 rows
    out, expected, dense, kname = _run(qnnp, big, 0)
    assert dense, kname
    assert_bytes_equal(out, expected, f"gfx950 dense image on {kname} (automatic) vs oracle")


def test_other_forced_kernels_keep_the_grouped_image(qnnp):
    big = ConvCase("d_forced_29", (28, 28), groups=4, gic=17, goc=62, batch=84)
    out, expected, dense, kname = _run(qnnp, big, 29)
    assert not dense and kname.endswith("_u16"), kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} (forced, grouped) vs oracle")


def test_forcing_the_dense_image_where_none_exists_is_refused(qnnp):
    # more than 1024 channels on one side: no dense image is built
    case = ConvCase("d_none", (3, 3), groups=2, gic=520, goc=16, batch=1)
    with pytest.raises(QnnpackError):
        _run(qnnp, case, 31)
    # a single group has none either
    with pytest.raises(QnnpackError):
        _run(qnnp, ConvCase("d_none_g1", (3, 3), groups=1, gic=32, goc=32, batch=1), 31)
