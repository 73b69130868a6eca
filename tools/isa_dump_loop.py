#!/usr/bin/env python3
"""Print the instructions of the loops of one kernel from hipcc -S output.
usage: tools/isa_dump_loop.py file.s <mangled-name regex> [min_mfma]  -- prints every backward-branch loop with >= min_mfma MFMAs"""
import re, sys
text = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for m in re.finditer(r"^(\S+):\s*; @\S+\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if not pat.search(name): continue
    print("==", name)
    labels, ins = {}, []
    for l in body.split("\n"):
        s = l.strip()
        lm = re.match(r"(\.LBB\d+_\d+):", s)
        if lm: labels[lm.group(1)] = len(ins); continue
        if not s or s.startswith(";") or s.startswith("."): continue
        ins.append(s.split(";")[0].rstrip())
    for i, x in enumerate(ins):
        bm = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", x) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", x)
        if bm and bm.group(1) in labels and labels[bm.group(1)] < i:
            st = labels[bm.group(1)]
            n = sum(1 for y in ins[st:i] if "mfma" in y)
            if n >= min_mfma:
                print(f"-- loop [{st},{i}] {i-st} instructions, {n} mfma")
                for y in ins[st:i + 1]: print("   ", y[:100])
