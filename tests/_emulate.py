"""numpy replay of the DEVICE kernels' documented index arithmetic on the host-built
images (packed weights, folded bias, offset table), used by the CPU test tier to
check the create/setup-time host logic without a GPU. The formulas below are the
ones the HIP kernels implement (qnnpack_amd/csrc/hip/q8igemm.hip, q8dwconv.hip,
pack.h); the GPU tier then checks the kernels themselves against the oracle.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, Structure, c_float, c_int32, c_size_t, c_uint8, c_uint32, c_void_p

import numpy as np


class Requant(Structure):
    """struct qnnp_hip_requant (hip/qnnp_hip.h)"""

    _fields_ = [("multiplier", c_int32), ("remainder_mask", c_int32), ("remainder_threshold", c_int32),
                ("shift", c_uint32), ("output_min_less_zero_point", c_int32),
                ("output_max_less_zero_point", c_int32), ("output_zero_point", c_int32)]


def bind_debug_hooks(lib):
    L = lib.lib
    L.qnnp_debug_pack_igemm_w.restype = None
    L.qnnp_debug_pack_igemm_w.argtypes = [c_uint32] * 5 + [c_uint8, c_uint8, c_void_p, c_void_p, c_void_p, c_void_p]
    L.qnnp_debug_pack_igemm_w_slots.restype = None
    L.qnnp_debug_pack_igemm_w_slots.argtypes = [c_uint32] * 7 + [c_uint8, c_uint8, c_void_p, c_void_p, c_void_p, c_void_p]
    L.qnnp_debug_pack_dwconv_w.restype = None
    L.qnnp_debug_pack_dwconv_w.argtypes = [c_uint32] * 4 + [c_uint8, c_uint8, c_void_p, c_void_p, c_void_p, c_void_p]
    L.qnnp_debug_conv2d_offsets.restype = None
    L.qnnp_debug_conv2d_offsets.argtypes = [c_size_t] * 5 + [c_uint32] * 8 + [c_void_p]
    L.qnnp_debug_compute_requant.restype = None
    L.qnnp_debug_compute_requant.argtypes = [c_float, c_uint8, c_uint8, c_uint8, POINTER(Requant)]
    return L


def round_up(x, q):
    return (x + q - 1) // q * q


def host_requant(L, scale, zp, qmin, qmax) -> Requant:
    rq = Requant()
    L.qnnp_debug_compute_requant(np.float32(scale), zp, qmin, qmax, ctypes.byref(rq))
    return rq


def q31_requantize_np(acc: np.ndarray, rq: Requant) -> np.ndarray:
    """hip/requant.hip.h, vectorised in int64."""
    n = acc.astype(np.int64)
    p = n * int(rq.multiplier) + (1 << 30)
    q = ((p >> 31) + (1 << 31)) % (1 << 32) - (1 << 31)          # truncate to int32
    rem = (q & int(rq.remainder_mask)) - (n < 0)
    y = (q >> int(rq.shift)) + (rem > int(rq.remainder_threshold))
    y = np.maximum(y, int(rq.output_min_less_zero_point))
    y = np.minimum(y, int(rq.output_max_less_zero_point))
    return (y + int(rq.output_zero_point)).astype(np.uint8)


def wrap32(x: np.ndarray) -> np.ndarray:
    return ((x.astype(np.int64) + (1 << 31)) % (1 << 32) - (1 << 31)).astype(np.int64)


def host_pack_igemm(L, groups, n, k_total, izp, kzp, kernel, bias):
    n_pad, k_pad = round_up(n, 32), round_up(k_total, 64)
    kernel = np.ascontiguousarray(kernel, dtype=np.uint8)
    bias = np.ascontiguousarray(bias, dtype=np.int32)
    packed = np.empty(groups * n_pad * k_pad, dtype=np.int8)
    bias2 = np.empty(groups * n_pad, dtype=np.int32)
    L.qnnp_debug_pack_igemm_w(groups, n, k_total, n_pad, k_pad, izp, kzp, kernel.ctypes.data, bias.ctypes.data,
                              packed.ctypes.data, bias2.ctypes.data)
    return packed, bias2, n_pad, k_pad


def host_pack_igemm_slots(L, groups, n, ks, kc, kc_slot, izp, kzp, kernel, bias):
    """pack.h "channel slots": a tap occupies kc_slot K positions (4 for 3-channel inputs)."""
    n_pad, k_pad = round_up(n, 32), round_up(ks * kc_slot, 64)
    kernel = np.ascontiguousarray(kernel, dtype=np.uint8)
    bias = np.ascontiguousarray(bias, dtype=np.int32)
    packed = np.empty(groups * n_pad * k_pad, dtype=np.int8)
    bias2 = np.empty(groups * n_pad, dtype=np.int32)
    L.qnnp_debug_pack_igemm_w_slots(groups, n, ks, kc, kc_slot, n_pad, k_pad, izp, kzp, kernel.ctypes.data,
                                    bias.ctypes.data, packed.ctypes.data, bias2.ctypes.data)
    return packed, bias2, n_pad, k_pad


def emulate_igemm_c3(L, n, ks, izp, kzp, kernel, bias, a_rows, rows, rq, out_stride, fill=0xA5):
    """3-channel slot mode of q8igemm.hip: each tap is fetched as one dword whose 4th byte is replaced by
    K padding (raw 0x80 -> a' = 0); padding taps read {izp, izp, izp, 0x80}. a_rows: uint8 [rows, ks*3]."""
    packed, bias2, n_pad, k_pad = host_pack_igemm_slots(L, 1, n, ks, 3, 4, izp, kzp, kernel, bias)
    w = unpack_fragments(packed, 1, n_pad, k_pad)[0]
    a4 = np.full((rows, ks, 4), 0x80, dtype=np.int64)
    a4[:, :, :3] = a_rows.reshape(rows, ks, 3)
    a4 = a4.reshape(rows, ks * 4)
    a_s = (a4 ^ 0x80) - 256 * ((a4 ^ 0x80) >= 128)
    a_pad = np.zeros((rows, k_pad), dtype=np.int64)
    a_pad[:, :ks * 4] = a_s
    rowsum = a_pad.sum(axis=1)
    acc = wrap32(a_pad @ w.T + (128 - kzp) * rowsum[:, None] + bias2[None, :].astype(np.int64))
    q = q31_requantize_np(acc[:, :n], rq)
    out = np.full((rows - 1) * out_stride + n, fill, dtype=np.uint8)
    for r in range(rows):
        out[r * out_stride: r * out_stride + n] = q[r]
    return out


def unpack_fragments(packed, groups, n_pad, k_pad) -> np.ndarray:
    """What the MFMA sees: W'[g][col][kk] from the fragment panels, using the KERNEL's read rule:
    the wave's fragment for (column block nb, K block kb) is the 1 KiB at ((g*NB + nb)*KB + kb)*1024,
    lane l bytes j=0..15 <-> (col = nb*32 + (l & 31), kk = kb*32 + (l >> 5)*16 + j)."""
    NB, KB = n_pad // 32, k_pad // 32
    frag = packed.reshape(groups, NB, KB, 64, 16).astype(np.int64)
    w = np.zeros((groups, n_pad, k_pad), dtype=np.int64)
    lane = np.arange(64)
    for nb in range(NB):
        for kb in range(KB):
            cols = nb * 32 + (lane & 31)
            for j in range(16):
                kk = kb * 32 + (lane >> 5) * 16 + j
                w[:, cols, kk] = frag[:, nb, kb, lane, j]
    return w


def emulate_igemm(L, case_groups, n, kc, ks, izp, kzp, kernel, bias, a_rows_fn, rows, rq, out_stride, fill=0xA5):
    """a_rows_fn(g) -> uint8 [rows, ks*kc] activation matrix as the kernel gathers it
    (padding taps already substituted by the input zero point)."""
    k_total = ks * kc
    packed, bias2, n_pad, k_pad = host_pack_igemm(L, case_groups, n, k_total, izp, kzp, kernel, bias)
    w = unpack_fragments(packed, case_groups, n_pad, k_pad)
    out = np.full((rows - 1) * out_stride + case_groups * n, fill, dtype=np.uint8)
    row_coeff = 128 - kzp
    for g in range(case_groups):
        a = a_rows_fn(g).astype(np.int64)
        a_s = (a ^ 0x80) - 256 * ((a ^ 0x80) >= 128)                       # a' = int8(a ^ 0x80) = a - 128
        a_pad = np.zeros((rows, k_pad), dtype=np.int64)
        a_pad[:, :k_total] = a_s                                            # K padding: a' = 0
        rowsum = a_pad.sum(axis=1)
        acc = a_pad @ w[g].T                                                # [rows, n_pad]
        acc = wrap32(acc + row_coeff * rowsum[:, None] + bias2.reshape(case_groups, n_pad)[g][None, :].astype(np.int64))
        q = q31_requantize_np(acc[:, :n], rq)
        for r in range(rows):
            out[r * out_stride + g * n: r * out_stride + g * n + n] = q[r]
    return out


def host_offsets(L, case, oh, ow) -> np.ndarray:
    H, W = case.input_size
    taps = case.kernel_size[0] * case.kernel_size[1]
    table = np.empty(oh * ow * taps, dtype=np.int32)
    L.qnnp_debug_conv2d_offsets(H, W, case.in_stride, oh, ow, case.kernel_size[0], case.kernel_size[1],
                                case.subsampling[0], case.subsampling[1], case.dilation[0], case.dilation[1],
                                case.padding[0], case.padding[3], table.ctypes.data)
    return table.reshape(oh * ow, taps)


def gather_conv_rows(case, inp, offsets, g, oh, ow):
    """Kernel gather rule: a(m, tap*kc + ch) = input[img*image_stride + offsets[pix][tap] + g*kc + ch], zp if offset < 0."""
    H, W = case.input_size
    kc = case.gic
    taps = offsets.shape[1]
    image_stride = H * W * case.in_stride
    rows = case.batch * oh * ow
    a = np.empty((rows, taps * kc), dtype=np.uint8)
    ch = np.arange(kc)
    for m in range(rows):
        img, pix = divmod(m, oh * ow)
        for t in range(taps):
            off = int(offsets[pix, t])
            if off < 0:
                a[m, t * kc:(t + 1) * kc] = case.izp
            else:
                base = img * image_stride + off + g * kc
                a[m, t * kc:(t + 1) * kc] = inp[base + ch]
    return a


def emulate_dwconv(L, case, inp, kernel, bias, rq, oh, ow, fill=0xA5):
    C = case.groups
    KH, KW = case.kernel_size
    taps = KH * KW
    c_pad = round_up(C, 16)
    kernel = np.ascontiguousarray(kernel.reshape(C, taps), dtype=np.uint8)
    bias = np.ascontiguousarray(bias, dtype=np.int32)
    wadj = np.empty(taps * c_pad, dtype=np.int16)
    bias1 = np.empty(c_pad, dtype=np.int32)
    L.qnnp_debug_pack_dwconv_w(C, c_pad, KH, KW, case.izp, case.kzp, kernel.ctypes.data, bias.ctypes.data,
                               wadj.ctypes.data, bias1.ctypes.data)
    wadj = wadj.reshape(taps, c_pad).astype(np.int64)
    H, W = case.input_size
    rows = case.batch * oh * ow
    out = np.full((rows - 1) * case.out_stride + C, fill, dtype=np.uint8)
    ch = np.arange(C)
    for n in range(case.batch):
        for oy in range(oh):
            for ox in range(ow):
                acc = bias1[:C].astype(np.int64).copy()
                for ky in range(KH):
                    iy = oy * case.subsampling[0] + ky * case.dilation[0] - case.padding[0]
                    for kx in range(KW):
                        ix = ox * case.subsampling[1] + kx * case.dilation[1] - case.padding[3]
                        if 0 <= iy < H and 0 <= ix < W:
                            a = inp[((n * H + iy) * W + ix) * case.in_stride + ch].astype(np.int64)
                        else:
                            a = np.full(C, case.izp, dtype=np.int64)
                        acc += a * wadj[ky * KW + kx, :C]
                m = (n * oh + oy) * ow + ox
                out[m * case.out_stride: m * case.out_stride + C] = q31_requantize_np(wrap32(acc), rq)
    return out


def emulate_dwconv_dot4(L, case, inp, kernel, bias, rq, oh, ow, fill=0xA5):
    """Replay of the int8 dot-product walk of the 3x3 stride-1 column kernel (q8dwconv.hip, QUAD flavour) from the HOST
    image it consumes (pack.h qnnp_pack_dwconv_dot4): per input row the quad T = (col0, col1, col2, 0) of bytes
    re-centred with kx (padding pixels read the input zero point and are re-centred like any other), an output is
    image[3] + T[t] . W4[0] + T[t+1] . W4[1] + T[t+2] . W4[2] in int8 x int8 products. Returns None when the weights'
    range class is 0 (the kernel then takes the int16 pair walk)."""
    import ctypes
    C = case.groups
    assert case.kernel_size == (3, 3) and case.subsampling == (1, 1) and case.dilation == (1, 1)
    c_pad = round_up(C, 16)
    kernel = np.ascontiguousarray(kernel.reshape(C, 9), dtype=np.uint8)
    bias = np.ascontiguousarray(bias, dtype=np.int32)
    wadj = np.empty(9 * c_pad, dtype=np.int16)
    bias1 = np.empty(c_pad, dtype=np.int32)
    image = np.zeros(4 * c_pad, dtype=np.uint32)
    fn = L.qnnp_debug_pack_dwconv_dot4
    fn.restype = ctypes.c_uint32
    fn.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint8, ctypes.c_uint8] + [ctypes.c_void_p] * 5
    rng_class = fn(C, c_pad, case.izp, case.kzp, kernel.ctypes.data, bias.ctypes.data, wadj.ctypes.data,
                   bias1.ctypes.data, image.ctypes.data)
    if rng_class == 0:
        return None
    image = image.reshape(4, c_pad)
    w4 = image[:3].view(np.int8).reshape(3, c_pad, 4).astype(np.int64)[:, :C, :]      # [row][channel][col]
    bias4 = image[3, :C].astype(np.int64)
    kx = 0x80 if rng_class == 1 else 0x7F
    H, W = case.input_size
    rows = case.batch * oh * ow
    out = np.full((rows - 1) * case.out_stride + C, fill, dtype=np.uint8)
    ch = np.arange(C)

    def quad(n, iy, ox):            # [channel][4] int8 values of input row iy, columns ox - pad_left .. + 2, and a zero
        t = np.zeros((C, 4), dtype=np.int64)
        for k in range(3):
            ix = ox + k - case.padding[3]
            if 0 <= iy < H and 0 <= ix < W:
                a = inp[((n * H + iy) * W + ix) * case.in_stride + ch]
            else:
                a = np.full(C, case.izp, dtype=np.uint8)
            t[:, k] = (a ^ kx).astype(np.uint8).view(np.int8)
        return t

    for n in range(case.batch):
        for ox in range(ow):
            for oy in range(oh):
                acc = bias4.copy()
                for r in range(3):
                    acc += (quad(n, oy + r - case.padding[0], ox) * w4[r]).sum(axis=1)
                m = (n * oh + oy) * ow + ox
                out[m * case.out_stride: m * case.out_stride + C] = q31_requantize_np(wrap32(acc), rq)
    return out
