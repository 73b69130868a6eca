"""GPU tier: the register sliding-window depthwise kernel (q8_dwconv_row3x3_kernel in
qnnpack_amd/csrc/hip/q8dwconv.hip), forced with "dwconv_kernel" = 3, against the scalar oracle: strides 1 and
2, every padding combination the reference's tests use, images narrower than a window / a segment, odd sizes,
channel counts with a ragged last wave, pixel strides, batch, zero points and clamps."""
import pytest

from _cases import CONV_CASES, EXTRA_CONV_CASES, ConvCase, conv_tensors

CONV_BY_NAME = {c.name: c for c in list(CONV_CASES) + list(EXTRA_CONV_CASES)}
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run

pytestmark = pytest.mark.gpu


def _dw(name, hw, c, **kw):
    kw.setdefault("padding", (1, 1, 1, 1))
    return ConvCase(name, hw, (3, 3), kw.pop("padding"), groups=c, gic=1, goc=1, **kw)


CASES = [
    _dw("r_c32_14", (14, 14), 32, batch=3),
    _dw("r_c4_1x1img", (1, 1), 4),
    _dw("r_c8_2x3img", (2, 3), 8, batch=2),
    _dw("r_c16_3x3img_nopad", (3, 3), 16, padding=(0, 0, 0, 0)),
    _dw("r_c24_9x40_wide", (9, 40), 24),
    _dw("r_c20_40x9_tall", (40, 9), 20, batch=2),
    _dw("r_c96_s2", (29, 31), 96, subsampling=(2, 2)),
    _dw("r_c144_s2_even", (28, 28), 144, subsampling=(2, 2), batch=2),
    _dw("r_c32_s2_pad_tl_only", (15, 15), 32, subsampling=(2, 2), padding=(1, 0, 0, 1)),
    _dw("r_c32_pad_asym", (12, 13), 32, padding=(1, 0, 1, 0)),
    _dw("r_c32_pad2", (10, 11), 32, padding=(2, 2, 2, 2)),
    _dw("r_c260_ragged_lanes", (7, 7), 260, batch=5),
    _dw("r_c960_7x7", (7, 7), 960, batch=2),
    _dw("r_c32_strided_pixels", (11, 12), 32, input_pixel_stride=40, output_pixel_stride=36),
    _dw("r_c64_zp", (9, 9), 64, izp=255, kzp=0),
    _dw("r_c64_zp2", (9, 9), 64, izp=0, kzp=255),
    _dw("r_c32_qmin_qmax", (9, 9), 32, qmin=100, qmax=150),
    _dw("r_c32_112", (112, 112), 32),
]


@pytest.fixture()
def row(qnnp):
    qnnp.set_option("dwconv_kernel", 3)
    yield qnnp
    qnnp.set_option("dwconv_kernel", 0)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_row_kernel_matches_oracle(row, case):
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(row, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname == "q8_dwconv_row_3x3", kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("name", ["x_dw5x5_c64", "x_dw3x3_c64_d2"])
def test_unsupported_shapes_are_reported_not_silently_rerouted(row, name):
    from qnnpack_amd import QnnpackError
    case = CONV_BY_NAME[name]
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    with pytest.raises(QnnpackError):
        conv_run(row, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
