"""GPU tier: the barrier-free streaming kernel for short-K pointwise / fully-connected layers
(qnnpack_amd/csrc/hip/q8pwconv.hip), forced with "gemm_kernel" = 5, against the scalar oracle: row-block
edges, every K-block count and both load widths (16- and 8-byte aligned rows), channel-count edges and the
three store modes, strides, zero-point / clamp variants, and the 1x1 convolution form. MobileNetV2's
pointwise layers (BASELINE configs[3]) take this kernel automatically; test_gpu_fullsize.py asserts that."""
import numpy as np
import pytest

from _cases import ConvCase, FcCase
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run, fc_expected, fc_run

pytestmark = pytest.mark.gpu

KERNEL = "q8_pw_stream_mfma"


@pytest.fixture()
def pw(qnnp):
    qnnp.set_option("gemm_kernel", 5)
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


def _fc(pw, case):
    expected, quant = fc_expected(case)
    out, kname = fc_run(pw, case, quant, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("m", [1, 2, 31, 32, 33, 63, 64, 65, 100, 1000, 4097])
def test_row_edges(pw, m):
    _fc(pw, FcCase(f"pw_m{m}", m, 32, 48))


@pytest.mark.parametrize("k", [16, 32, 48, 64, 80, 96, 112, 128, 144, 160, 176, 192, 208, 224, 240, 256])
def test_k_blocks_16_byte_rows(pw, k):
    _fc(pw, FcCase(f"pw_k{k}", 131, k, 40))


@pytest.mark.parametrize("k", [8, 24, 40, 56, 72, 104, 136, 200, 248])
def test_k_blocks_8_byte_rows(pw, k):
    _fc(pw, FcCase(f"pw_k{k}", 77, k, 36))


@pytest.mark.parametrize("n", [1, 3, 4, 15, 16, 17, 31, 32, 33, 48, 96, 100, 144, 255, 256, 576])
def test_channel_edges_and_store_modes(pw, n):
    _fc(pw, FcCase(f"pw_n{n}", 70, 32, n))


@pytest.mark.parametrize("kw", [dict(izp=0, kzp=0), dict(izp=255, kzp=255), dict(izp=128, kzp=128),
                                dict(izp=3, kzp=250), dict(qmin=128), dict(qmax=128)],
                         ids=lambda d: "_".join(f"{k}{v}" for k, v in d.items()))
def test_quantization_variants(pw, kw):
    _fc(pw, FcCase("pw_q_" + "_".join(f"{k}{v}" for k, v in kw.items()), 90, 48, 52, **kw))


def test_strided_rows(pw):
    _fc(pw, FcCase("pw_strided", 130, 48, 80, input_stride=64, output_stride=96))


@pytest.mark.parametrize("n", [8, 24, 40, 56, 72, 104, 120, 248])
@pytest.mark.parametrize("m", [32, 33, 63, 64, 65, 97, 1000, 4097])
def test_dense_rows_of_8_mod_16_bytes(pw, m, n):
    """round 4: 24-channel MobileNet layers (56x56x96 -> 24, 144 -> 24) and their kin -- dense output rows whose length is a
    multiple of 8 but not of 16 bytes leave the staged kernel as ONE contiguous 16-byte aligned run per 32-row block
    (stream_copy_out's third mode; an odd number of rows ends in an 8-byte store)"""
    _fc(pw, FcCase(f"pw_dense8_m{m}_n{n}", m, 96, n))


@pytest.mark.parametrize("kw", [dict(output_stride=40), dict(input_stride=160), dict(izp=3, kzp=250), dict(qmin=100, qmax=150)],
                         ids=lambda d: "_".join(f"{k}{v}" for k, v in d.items()))
def test_dense8_neighbours(pw, kw):
    """strided output rows keep the direct-store flavour; strided input rows, zero points and clamps do not matter"""
    _fc(pw, FcCase("pw_dense8_" + "_".join(f"{k}{v}" for k, v in kw.items()), 333, 144, 24, **kw))


@pytest.mark.parametrize("k", [16, 48, 96, 144, 200, 256])
def test_dense8_k_blocks(pw, k):
    _fc(pw, FcCase(f"pw_dense8_k{k}", 201, k, 24))


def test_strided_rows_8_byte(pw):
    _fc(pw, FcCase("pw_strided8", 130, 24, 20, input_stride=40, output_stride=28))


def test_unaligned_output_uses_byte_stores(pw):
    _fc(pw, FcCase("pw_out_unaligned", 67, 32, 18, output_stride=19))


@pytest.mark.parametrize("case", [
    ConvCase("pw_1x1_16_96", (28, 30), gic=16, goc=96, batch=3),
    ConvCase("pw_1x1_24_144", (14, 14), gic=24, goc=144, batch=2),
    ConvCase("pw_1x1_144_24", (13, 11), gic=144, goc=24, batch=2),
    ConvCase("pw_1x1_strided_pixels", (9, 9), gic=32, goc=36, input_pixel_stride=48, output_pixel_stride=40),
], ids=lambda c: c.name)
def test_pointwise_convolution(pw, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(pw, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


# ---- strided 1x1 convolutions (round 5): the staged flavour reads a table row per output pixel (ResNet's downsampling
#      shortcuts; reference: the same indirection + q8conv path as any convolution, src/convolution.c:414-421) ----
@pytest.mark.parametrize("case", [
    ConvCase("pw_1x1_s2_64_128", (28, 28), subsampling=(2, 2), gic=64, goc=128, batch=3),
    ConvCase("pw_1x1_s2_odd_sizes", (27, 29), subsampling=(2, 2), gic=32, goc=48, batch=2),
    ConvCase("pw_1x1_s3x2_rect", (20, 31), subsampling=(3, 2), gic=16, goc=64, batch=5),
    ConvCase("pw_1x1_s2_strided_pixels", (15, 15), subsampling=(2, 2), gic=64, goc=96, batch=2, input_pixel_stride=80, output_pixel_stride=112),
    ConvCase("pw_1x1_s2_8_byte_rows", (16, 18), subsampling=(2, 2), gic=24, goc=64, batch=2),
    ConvCase("pw_1x1_s2_256_512", (14, 14), subsampling=(2, 2), gic=256, goc=512, batch=4),
    ConvCase("pw_1x1_s2_kzp126", (12, 12), subsampling=(2, 2), gic=128, goc=256, batch=2, kzp=126, izp=3),
    ConvCase("pw_1x1_s4_one_column", (13, 3), subsampling=(4, 4), gic=48, goc=80, batch=7),
    # ONE output pixel per image (rows_per_image == 1: the 32-bit division magic does not exist for a divisor of 1)
    ConvCase("pw_1x1_s2_one_pixel_per_image", (2, 2), subsampling=(2, 2), gic=64, goc=96, batch=70),
    ConvCase("pw_1x1_s2_one_pixel_image", (1, 1), subsampling=(2, 2), gic=32, goc=64, batch=2100),
], ids=lambda c: c.name)
def test_strided_pointwise_convolution(pw, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(pw, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("case", [
    ConvCase("pw_1x1_s2_padded", (14, 14), padding=(1, 1, 1, 1), subsampling=(2, 2), gic=64, goc=128),   # taps on padding
    ConvCase("pw_1x1_s2_n32", (14, 14), subsampling=(2, 2), gic=64, goc=32),          # one channel block: the first flavour, no table
    ConvCase("pw_1x1_s2_grouped", (14, 14), subsampling=(2, 2), groups=2, gic=32, goc=64),
], ids=lambda c: c.name)
def test_strided_pointwise_shapes_it_does_not_take(qnnp, case):
    from qnnpack_amd import QnnpackError
    expected, quant, out_hw = conv_expected(case)
    qnnp.set_option("gemm_kernel", 5)
    try:
        with pytest.raises(QnnpackError):
            conv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
    finally:
        qnnp.set_option("gemm_kernel", 0)
    out, kname = conv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)   # auto: another kernel, same bytes
    assert kname != KERNEL
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("case", [FcCase("pw_bad_k", 64, 260, 16),       # K > 256
                                  FcCase("pw_bad_align", 64, 20, 16),    # rows only 4-byte aligned
                                  FcCase("pw_bad_lds", 64, 256, 500)],   # weights exceed the LDS budget and the rows are
                                                                         # not 16-byte aligned (no channel columns)
                         ids=lambda c: c.name)
def test_unsupported_shapes_are_reported_not_silently_rerouted(pw, case):
    from qnnpack_amd import QnnpackError
    _, quant = fc_expected(case)
    with pytest.raises(QnnpackError):
        fc_run(pw, case, quant, to_device=to_device, from_device=from_device)


# ---- channel columns (16-byte aligned rows): every workgroup column owns 2 / 4 / 8 channel blocks, any N fits ----
@pytest.mark.parametrize("case", [
    FcCase("pw_cols_n512_k256", 64, 256, 512),                       # the old LDS limit: 16 blocks x 8 KiB of weights
    FcCase("pw_cols_n576_k96", 200, 96, 576),                        # MobileNetV2 14x14 expand shape, few rows
    FcCase("pw_cols_n960_k160", 300, 160, 960),
    FcCase("pw_cols_n272_ragged", 99, 48, 272),                      # 8.5 blocks: the last column is one half block
    FcCase("pw_cols_n144_k24", 130, 24, 144),                        # 8-byte aligned input rows, 4.5 blocks
    FcCase("pw_cols_strided", 130, 64, 320, input_stride=80, output_stride=352),
    FcCase("pw_cols_clamp_zp", 70, 32, 192, izp=3, kzp=250, qmin=20, qmax=230),
    FcCase("pw_cols_1row", 1, 32, 256),
], ids=lambda c: c.name)
def test_channel_columns(pw, case):
    _fc(pw, case)


@pytest.mark.parametrize("case", [
    FcCase("pw_cols8_rows90000", 90000, 32, 544),                    # 17 blocks in columns of 8 (the last holds one)
    FcCase("pw_cols4_rows100000", 100000, 32, 272),                  # 9 blocks in columns of 4
    FcCase("pw_whole_rows140000", 140000, 16, 96),                   # enough rows: whole rows per unit, dense image
], ids=lambda c: c.name)
def test_channel_columns_chosen_by_row_count(pw, case):
    _fc(pw, case)


def test_auto_selection_takes_it_for_many_rows(qnnp):
    case = FcCase("pw_auto", 4096, 32, 64)
    expected, quant = fc_expected(case)
    out, kname = fc_run(qnnp, case, quant, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


# ---- the global-operand flavour ("gemm_kernel" = 6): one wave per 32x32 block, any K; from K = 256 on the reduction is
#      split over the four waves of a workgroup (one workgroup per block) ----
GW_KERNEL = "q8_pw_stream_gw_mfma"
GWK_KERNEL = "q8_pw_stream_gwk_mfma"


def _gw_name(k):
    """(these cases have a handful of 32x32 blocks: from K = 256 on the reduction is split; with more than two blocks
    per CU it is not -- test_gw_many_blocks_keep_one_wave_per_block)"""
    return GWK_KERNEL if k >= 256 else GW_KERNEL


@pytest.fixture()
def gw(qnnp):
    qnnp.set_option("gemm_kernel", 6)
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


def _fc_gw(gw, case):
    expected, quant = fc_expected(case)
    out, kname = fc_run(gw, case, quant, to_device=to_device, from_device=from_device)
    assert kname == _gw_name(case.input_channels), kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("m", [1, 31, 32, 33, 100, 1000])
def test_gw_row_edges(gw, m):
    _fc_gw(gw, FcCase(f"gw_m{m}", m, 384, 48))


@pytest.mark.parametrize("k", [16, 32, 48, 112, 128, 144, 240, 256, 272, 384, 496, 512, 528, 576, 960, 1024, 1040, 1280, 2064])
def test_gw_k_blocks_and_unroll_tails(gw, k):
    _fc_gw(gw, FcCase(f"gw_k{k}", 70, k, 40))


@pytest.mark.parametrize("n", [1, 4, 16, 31, 33, 96, 160, 1000])
def test_gw_channel_edges_and_store_modes(gw, n):
    _fc_gw(gw, FcCase(f"gw_n{n}", 50, 320, n))


@pytest.mark.parametrize("kw", [dict(izp=0, kzp=0), dict(izp=255, kzp=255), dict(izp=3, kzp=250), dict(qmin=128)],
                         ids=lambda d: "_".join(f"{k}{v}" for k, v in d.items()))
def test_gw_quantization_variants(gw, kw):
    _fc_gw(gw, FcCase("gw_q_" + "_".join(f"{k}{v}" for k, v in kw.items()), 90, 400, 52, **kw))


def test_gw_strided_rows(gw):
    _fc_gw(gw, FcCase("gw_strided", 130, 304, 80, input_stride=320, output_stride=96))


def test_gw_pointwise_convolution(gw):
    case = ConvCase("gw_1x1_960_160", (7, 7), gic=960, goc=160, batch=4)
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(gw, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == GWK_KERNEL, kname          # 35 blocks of K = 960: split over the waves
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


def test_gw_many_blocks_keep_one_wave_per_block(gw):
    case = FcCase("gw_many_blocks", 6272, 576, 160)          # 196 x 5 blocks > 2 per CU
    expected, quant = fc_expected(case)
    out, kname = fc_run(gw, case, quant, to_device=to_device, from_device=from_device)
    assert kname == GW_KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


def test_gw_unsupported_alignment_is_reported(gw):
    from qnnpack_amd import QnnpackError
    case = FcCase("gw_bad_align", 64, 24, 16)         # rows only 8-byte aligned
    _, quant = fc_expected(case)
    with pytest.raises(QnnpackError):
        fc_run(gw, case, quant, to_device=to_device, from_device=from_device)


# ---- the long-K staged flavour ("gemm_kernel" = 9): 256 < K <= 1024, 16-byte aligned rows on both sides ----
LONGK_KERNEL = "q8_pw_stream_longk_mfma"


@pytest.fixture()
def longk(qnnp):
    qnnp.set_option("gemm_kernel", 9)
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


def _fc_longk(lib, case):
    expected, quant = fc_expected(case)
    out, kname = fc_run(lib, case, quant, to_device=to_device, from_device=from_device)
    assert kname == LONGK_KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("k", [272, 288, 304, 320, 384, 400, 576, 624, 640, 656, 960, 1008, 1024])
def test_longk_k_blocks_across_the_three_register_budgets(longk, k):
    """9..12 K blocks run in the 12-block kernel, 13..20 in the 20-block one, 21..32 in the 32-block one; K % 32 = 16
    leaves half of the last block to the a' == 0 line."""
    _fc_longk(longk, FcCase(f"lk_k{k}", 70, k, 48))


@pytest.mark.parametrize("m", [1, 31, 32, 33, 100, 1000])
def test_longk_row_edges(longk, m):
    _fc_longk(longk, FcCase(f"lk_m{m}", m, 384, 96))


@pytest.mark.parametrize("case", [
    FcCase("lk_n16", 50, 320, 16),
    FcCase("lk_n96_whole", 90, 384, 96),                              # three blocks: whole rows per unit
    FcCase("lk_n160_columns", 90, 576, 160),                          # five blocks in columns of two (the last of one)
    FcCase("lk_n320_columns", 64, 960, 320),
    FcCase("lk_n1280_columns", 40, 320, 1280),                        # MobileNetV2 layer 30's shape
    FcCase("lk_n272_ragged", 45, 400, 272),                           # 8.5 blocks
    FcCase("lk_strided", 130, 304, 80, input_stride=320, output_stride=96),
    FcCase("lk_clamp_zp", 70, 512, 64, izp=3, kzp=250, qmin=20, qmax=230),
    FcCase("lk_zp_extremes", 70, 448, 64, izp=255, kzp=0),
], ids=lambda c: c.name)
def test_longk_channel_layouts_and_quantization(longk, case):
    _fc_longk(longk, case)


@pytest.mark.parametrize("case", [
    FcCase("lk_pf_k512_n128", 90000, 512, 128),                       # more units than waves: the next unit's rows in flight (K budget 20)
    FcCase("lk_pf_k384_n96_odd_rows", 100001, 384, 96),               # K budget 12, a ragged last unit
    FcCase("lk_pf_k640_n64_zp", 81000, 640, 64, izp=3, kzp=126),
    FcCase("lk_pf_strided", 85000, 512, 48, input_stride=528, output_stride=64),
], ids=lambda c: c.name)
def test_longk_many_rows_prefetching_flavour(longk, case):
    _fc_longk(longk, case)


@pytest.mark.parametrize("kzp,kernel", [(126, LONGK_KERNEL), (127, "q8_gemm_mfma_128x128_c16")], ids=["kzp126", "kzp127"])
def test_longk_takes_resnet50_512_to_128_automatically(qnnp, kzp, kernel):
    """28x28x512 -> 128 at batch 128 (100352 rows, four channel blocks): the long-K flavour with the next unit in flight -- round 6:
    where the operator has a zero-point-centred image (kernel zero point 127 / 128) the 128-row centred GEMM takes it (32.6 -> 21.2 us)."""
    case = ConvCase("lk_1x1_512_128", (28, 28), gic=512, goc=128, batch=128, kzp=kzp)
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == kernel, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("kzp,kernel", [(126, LONGK_KERNEL), (127, "q8_gemm_mfma_128x128_c16")], ids=["kzp126", "kzp127"])
def test_longk_pointwise_convolution_and_auto_selection(qnnp, kzp, kernel):
    """14x14x384 -> 96 at a batch that gives 784 row blocks: chosen automatically (MobileNetV2 layer 20) -- round 6: with a centred
    image the 128-row centred GEMM is (8.1 -> 6.2 us); other kernel zero points keep the long-K flavour."""
    case = ConvCase("lk_1x1_384_96", (14, 14), gic=384, goc=96, batch=128, kzp=kzp)
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == kernel, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("case", [FcCase("lk_bad_short_k", 64, 256, 64),      # K <= 256 belongs to the short-K kernels
                                  FcCase("lk_bad_k", 64, 1040, 64),           # K > 1024
                                  FcCase("lk_bad_out_align", 64, 384, 72)],   # output rows not 16-byte aligned
                         ids=lambda c: c.name)
def test_longk_unsupported_shapes_are_reported(longk, case):
    from qnnpack_amd import QnnpackError
    _, quant = fc_expected(case)
    with pytest.raises(QnnpackError):
        fc_run(longk, case, quant, to_device=to_device, from_device=from_device)
