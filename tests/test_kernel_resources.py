"""CPU tier: what the compiler made of the shipped kernels. The gfx950 code objects are taken out of
libqnnpack_gfx950.so (clang offload bundles in the .hip_fatbin section) and their AMDGPU metadata is read with
llvm-readelf: no kernel of the default dispatch paths may touch scratch memory (a spill inside a streaming loop is a
round trip to memory per trip -- three flavours of the staged pointwise kernel had acquired some unnoticed), and the
register need of the occupancy-critical ones must stay within the wave count they are launched for."""
import os
import re
import shutil
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "qnnpack_amd", "libqnnpack_gfx950.so")
READELF = shutil.which("llvm-readelf") or "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _code_objects(blob):
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0:
            return
        (count,) = struct.unpack_from("<Q", blob, i + 24)
        off = i + 32
        for _ in range(count):
            o, size, tsize = struct.unpack_from("<QQQ", blob, off)
            off += 24
            triple = blob[off:off + tsize].decode()
            off += tsize
            if "gfx950" in triple and size:
                yield blob[i + o:i + o + size]
        pos = i + len(MAGIC)


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    if not os.path.exists(READELF):
        pytest.skip("llvm-readelf not available")
    out = {}
    tmp = tmp_path_factory.mktemp("codeobj")
    blob = open(LIB, "rb").read()
    for k, elf in enumerate(_code_objects(blob)):
        path = tmp / f"co{k}.elf"
        path.write_bytes(elf)
        notes = subprocess.run([READELF, "--notes", str(path)], capture_output=True, text=True, check=True).stdout
        for entry in notes.split("  - .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", entry).group(1)
            demangled = name
            out[demangled] = {
                "vgpr": int(re.search(r"\.vgpr_count:\s+(\d+)", entry).group(1)),
                "scratch": int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", entry).group(1)),
                "spill": int(re.search(r"\.vgpr_spill_count:\s+(\d+)", entry).group(1)),
            }
    assert len(out) > 100, "expected the library's kernel instantiations"
    return out


# mangled-name fragments of the kernels the automatic dispatch picks for aligned tensors; the
# generic implicit-GEMM fallback for unaligned / odd-channel tensors (q8_igemm_mfma_kernel, byte gathers) is known to
# spill in its 1-byte flavours and is reported by test_report_of_spilling_kernels below, not asserted
DEFAULT_PATH = ["q8_gemm_mfma_256x256_c16_kernel", "q8_gemm_mfma_128xN_c16_kernel", "q8_gemm_mfma_128xN_u16_kernel", "q8_conv_wave_ws16_kernel", "q8_gemm_mfma_256x256_c_kernel", "q8_gemm_mfma_256x256_kernelILb0ELi4E", "q8_gemm_mfma_256x256_kernelILb1ELi4E", "q8_pw_stream_staged_kernel", "q8_pw_stream_longk_kernel",
                "q8_pw_stream_gw_kernel", "q8_pw_stream_gwk_kernel", "q8_conv_stream_c3s_kernel",
                "q8_dwconv_col3x3_kernel", "q8_conv_wave_reg_kernel", "q8_conv_wave_ws_kernel", "q8_conv_lds_mfma", "q8_vadd", "q8_gavgpool",
                "q8_conv_patch_kernel", "q8_conv_c3rows32_kernel", "q8_conv_c3rows32_lds_kernel"]


def test_default_path_kernels_do_not_spill(kernels):
    checked = 0
    for name, k in kernels.items():
        if any(f in name for f in DEFAULT_PATH):
            checked += 1
            assert k["scratch"] == 0 and k["spill"] == 0, (name, k)
    assert checked >= 60, checked


@pytest.mark.parametrize("fragment,max_vgpr,why", [
    # 512 VGPRs per SIMD lane, allocated in blocks of 8: n waves per SIMD need <= floor(512 / n / 8) * 8 each
    ("23q8_dwconv_col3x3_kernelILi1ELi3ELb1ELb1ELb1E", 72, "dot-product depthwise walk, six row buffers: 7 waves per SIMD"),
    ("23q8_dwconv_col3x3_kernelILi2ELi3ELb1ELb0ELb0E", 80, "stride-2 depthwise walk: the 6 waves per SIMD its plan counts on"),
    ("26q8_pw_stream_staged_kernelILi1ELi16ELi3ELb1E", 72, "staged pointwise kernel, 1 K block: 7 workgroups per CU"),
    ("26q8_pw_stream_staged_kernelILi2ELi16ELi3ELb1E", 80, "2 K blocks: 6 per CU"),
    ("26q8_pw_stream_staged_kernelILi4ELi16ELi3ELb1E", 96, "3-4 K blocks: 5 per CU"),
    ("26q8_pw_stream_staged_kernelILi7ELi16ELi3ELb1E", 128, "5-7 K blocks: 4 per CU"),
    ("25q8_conv_stream_c3s_kernelILi3ELb1E", 96, "first-layer kernel: 5 per CU"),
    ("27q8_gemm_mfma_256x256_kernelILb0ELi4ELi256ELi0ELb0E", 256, "256x256 GEMM: 2 waves per SIMD"),
    ("29q8_gemm_mfma_256x256_c_kernel", 256, "256x256 GEMM, zero-point-centred flavour: 2 waves per SIMD"),
    ("31q8_gemm_mfma_256x256_c16_kernel", 256, "256x256 GEMM on 16x16x64 MFMAs (centred and row-sum flavours): 2 waves per SIMD"),
    ("29q8_gemm_mfma_128xN_c16_kernelILi3ELi0ELi4E", 128, "128x128 centred GEMM: 64 KiB of LDS allow two workgroups per CU, the registers must too"),
    ("29q8_gemm_mfma_128xN_c16_kernelILi3ELi0ELi2E", 80, "128x64 centred GEMM: three workgroups per CU by LDS"),
    ("24q8_conv_wave_ws16_kernel", 256, "weight-stationary 3x3 convolution on 16x16x64 MFMAs: 2 waves per SIMD"),
    ("20q8_conv_patch_kernelILi8ELi4E", 128, "patch kernel, 256 positions x 128 channels: TWO 8-wave workgroups per CU (4 waves per SIMD)"),
    ("20q8_conv_patch_kernelILi4ELi8E", 256, "patch kernel, 128 positions x 256 channels: 2 waves per SIMD"),
    ("23q8_conv_c3rows32_kernel", 256, "7x7 / 5x5 first-layer kernel: 2 waves per SIMD"),
    ("27q8_conv_c3rows32_lds_kernelILi2E", 168, "its LDS-staged flavour, 64 channels: 3 waves per SIMD (one workgroup's staging runs under two others' units)"),
    ("27q8_conv_c3rows32_lds_kernelILi1E", 168, "... 32 channels"),
])
def test_register_budgets_of_the_occupancy_critical_kernels(kernels, fragment, max_vgpr, why):
    hits = {n: k for n, k in kernels.items() if fragment in n}
    if "q8_conv_patch_kernelILi8ELi4E" in fragment:
        # (the channel-chunked flavour -- last template argument true -- is alone on its CU by its LDS plan: 256 registers)
        hits = {n: k for n, k in hits.items() if "Lb0EEEvNS_11IgemmParams" in n}
    assert hits, f"kernel {fragment} not found"
    for name, k in hits.items():
        assert k["vgpr"] <= max_vgpr, (name, k, why)


def test_report_of_spilling_kernels(kernels, capsys):
    """not an assertion: lists what spills, so that a change that adds to the list shows up in the test log"""
    spilling = sorted((k["scratch"], n) for n, k in kernels.items() if k["scratch"])
    with capsys.disabled():
        print(f"\n{len(spilling)} of {len(kernels)} kernels use scratch:", ", ".join(f"{b} B {n[:70]}" for b, n in spilling[-6:]))
    assert all("q8_igemm_mfma_kernel" in n or "conv_wave_mfma_kernelILi2E" in n or "row3x3" in n or
               "q8_gemm_mfma_256x256_kernelILb0ELi2E" in n or "q8_gemm_mfma_256x256_kernelILb1ELi2E" in n
               for _, n in spilling), "a kernel outside the known fallback / A-B flavours started to spill"


OBJDUMP = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"


def test_lean_gemm_is_the_only_writer_of_m0_in_its_kernels(tmp_path):
    """The lean and the zero-point-centred GEMM flavours issue their LDS-DMA as inline assembly and write m0 themselves,
    in the main loop one MFMA AHEAD of the load that uses it (q8gemm256.hip / q8gemm256c.hip, dma16_set_m0). That is only sound while nothing the compiler emits
    in those kernels touches m0: every m0 reference in their disassembly must be one of the kernel's own
    `s_mov_b32 m0, sN`, and every LDS-DMA must be the saddr form the inline assembly spells out."""
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    blob = open(LIB, "rb").read()
    seen = 0
    for k, elf in enumerate(_code_objects(blob)):
        path = tmp_path / f"co{k}.elf"
        path.write_bytes(elf)
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(path)], capture_output=True, text=True, check=True).stdout
        for m in re.finditer(r"^[0-9a-f]+ <(\S*q8_gemm_mfma_256x256_(?:c_)?kernel\S*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M):
            name, body = m.group(1), m.group(2)
            if "_c_kernel" not in name and not name.endswith("ELb0ELb1EEEvNS_11IgemmParamsE"):      # <..., PP = false, LEAN = true>
                continue
            seen += 1
            m0_lines = [ln.strip() for ln in body.split("\n") if re.search(r"\bm0\b", ln)]
            assert m0_lines, name
            for ln in m0_lines:
                assert re.match(r"s_mov_b32 m0, s\d+\b", ln), (name, ln)
            dma = [ln.strip() for ln in body.split("\n") if "global_load_lds" in ln]
            assert dma and all(re.match(r"global_load_lds_dwordx4 v\d+, s\[\d+:\d+\]", ln) for ln in dma), (name, dma[:3])
    assert seen == 1 + 18, seen     # the lean flavour + the centred flavour's 9 requantization / clamp classes (round 5: + the bounded sequence under an explicit clamp) x aligned or not (its burst-read A/B structure is in measurement builds only)


def test_streaming_store_flavours_survive_the_compiler(tmp_path):
    """"streaming_stores" (include/qnnpack_gfx950.h): the kernels that write whole lines once choose between a plain and
    an `nt` 16-byte store at run time. Written with the builtin in one arm of the branch, hipcc merged the two stores
    and dropped the hint without a word (every measurement then showed "no effect"); they are inline-asm instructions
    now, and this test looks for both flavours in the disassembly of each kernel family that has them."""
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    blob = open(LIB, "rb").read()
    want = {"q8_pw_stream_staged_kernel": 0, "q8_pw_stream_longk_kernel": 0, "q8_vadd_flat_kernel": 0,
            "q8_gemm_mfma_256x256_kernelILb0ELi4ELi256ELi0ELb0ELb1E": 0,
            "q8_dwconv_col3x3_kernel": 0, "q8_dwconv_col5x5_kernel": 0, "q8_conv_c3rows_kernel": 0, "q8_conv_c3rows32_kernel": 0,
            "q8_conv_c3rows32_lds_kernel": 0,
            "q8_conv_wave_ws_kernelILi2ELi2E": 0,        # (round 5: whole-line stores of the 64 -> 64 weight-stationary 3x3 kernel)
            "q8_gemm_mfma_256x256_c_kernel": 0}
    for k, elf in enumerate(_code_objects(blob)):
        path = tmp_path / f"co{k}.elf"
        path.write_bytes(elf)
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(path)], capture_output=True, text=True, check=True).stdout
        for m in re.finditer(r"^[0-9a-f]+ <(\S+)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M):
            name, body = m.group(1), m.group(2)
            for frag in want:
                if frag in name:
                    op = "buffer_store_dword " if "dwconv" in frag else ("buffer_store_dwordx4" if ("c3rows" in frag or "conv_wave_ws" in frag) else "global_store_dwordx4")
                    stores = [ln for ln in body.split("\n") if op in ln]
                    hinted = [ln for ln in stores if re.search(r"\bnt\b", ln)]
                    assert hinted and len(hinted) < len(stores), (name, len(hinted), len(stores))
                    want[frag] += 1
    assert all(v > 0 for v in want.values()), want


def test_patch_kernel_is_the_only_writer_of_m0_in_its_code(tmp_path):
    """q8_conv_patch_kernel (hip/q8convpatch.hip) issues every LDS-DMA as inline assembly and leaves m0 as it set it (a save /
    restore pair per piece would be half of the ring's scalar instructions). Sound only while nothing the compiler emits in
    that kernel reads or writes m0: every m0 reference in its disassembly must be one of its own `s_mov_b32 m0, sN`, and every
    LDS-DMA one of the two forms the assembly spells out (flat per-lane address: the patch; scalar base + lane offset: the ring)."""
    if not os.path.exists(LIB):
        pytest.skip("library not built")
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    blob = open(LIB, "rb").read()
    seen = 0
    for k, elf in enumerate(_code_objects(blob)):
        path = tmp_path / f"co{k}.elf"
        path.write_bytes(elf)
        dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(path)], capture_output=True, text=True, check=True).stdout
        for m in re.finditer(r"^[0-9a-f]+ <(\S*q8_conv_patch_kernel\S*)>:\n(.*?)(?=^[0-9a-f]+ <|\Z)", dis, re.S | re.M):
            name, body = m.group(1), m.group(2)
            seen += 1
            m0_lines = [ln.strip() for ln in body.split("\n") if re.search(r"\bm0\b", ln)]
            assert m0_lines, name
            for ln in m0_lines:
                assert re.match(r"s_mov_b32 m0, s\d+\b", ln), (name, ln)
            dma = [ln.strip() for ln in body.split("\n") if "global_load_lds" in ln]
            assert dma and all(re.match(r"global_load_lds_dwordx4 v(\d+|\[\d+:\d+\]), (s\[\d+:\d+\]|off)", ln) for ln in dma), (name, dma[:3])
    assert seen == 3 * 3 * 8, seen      # three tile flavours x (two step widths + the channel-chunked 128-byte one) x eight requantization classes
