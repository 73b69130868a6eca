#!/usr/bin/env python3
"""Print VGPR / scratch / LDS of the gfx950 kernels in an object or library whose (demangled) name matches a pattern.
usage: tools/kres.py <file.o|.so> [regex]"""
import re, struct, subprocess, sys, tempfile, os
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
def code_objects(blob):
    pos = 0
    while True:
        i = blob.find(MAGIC, pos)
        if i < 0: return
        (count,) = struct.unpack_from("<Q", blob, i + 24); off = i + 32
        for _ in range(count):
            o, size, tsize = struct.unpack_from("<QQQ", blob, off); off += 24
            triple = blob[off:off + tsize].decode(); off += tsize
            if "gfx950" in triple and size: yield blob[i + o:i + o + size]
        pos = i + len(MAGIC)
blob = open(sys.argv[1], "rb").read()
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for k, elf in enumerate(code_objects(blob)):
    with tempfile.NamedTemporaryFile(suffix=".elf", delete=False) as f:
        f.write(elf); path = f.name
    notes = subprocess.run([READELF, "--notes", path], capture_output=True, text=True).stdout
    os.unlink(path)
    for entry in notes.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", entry).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if pat and not pat.search(dem): continue
        g = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", entry).group(1))
        print(f"vgpr {g('vgpr_count'):4d} agpr {int(entry.split()[0]):4d} sgpr {g('sgpr_count'):4d} scratch {g('private_segment_fixed_size'):5d} spill {g('vgpr_spill_count'):4d} lds {g('group_segment_fixed_size'):7d}  {dem[:150]}")
