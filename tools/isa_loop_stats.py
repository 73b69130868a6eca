#!/usr/bin/env python3
"""Static instruction mix of the loops of one kernel, from hipcc's -S output.

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only ... file.hip -o file.s
    python tools/isa_loop_stats.py file.s <kernel name substring> [min VALU per loop]

For every backward branch (a loop) prints how many instructions of each class one trip issues. Used to budget VALU
issue slots per output for the VALU-bound kernels (tools/ubench_valu.hip gives the cycles per instruction class)."""
import collections
import re
import sys


def main():
    path, needle = sys.argv[1], sys.argv[2]
    min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    text = open(path).read()
    for m in re.finditer(r"^(\S+):\s*; @\S+\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if needle not in name:
            continue
        print("==", name)
        labels, instrs = {}, []
        for line in body.split("\n"):
            line = line.strip()
            lm = re.match(r"(\.LBB\d+_\d+):", line)
            if lm:
                labels[lm.group(1)] = len(instrs)
                continue
            if not line or line.startswith(";") or line.startswith("."):
                continue
            instrs.append(line)
        for i, ins in enumerate(instrs):
            bm = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", ins) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", ins)
            if not bm or bm.group(1) not in labels:
                continue
            start = labels[bm.group(1)]
            if start > i:
                continue
            ops = [x.split()[0] for x in instrs[start:i + 1]]
            c = collections.Counter(ops)
            valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
            if valu < min_valu:
                continue
            groups = collections.OrderedDict()
            groups["total"] = len(ops)
            groups["valu"] = valu
            groups["mfma"] = sum(v for k, v in c.items() if k.startswith("v_mfma"))
            groups["salu"] = sum(v for k, v in c.items() if k.startswith("s_") and not k.startswith("s_waitcnt") and not k.startswith("s_nop"))
            groups["waitcnt"] = c.get("s_waitcnt", 0)
            groups["vmem_ld"] = sum(v for k, v in c.items() if k.startswith("global_load") or k.startswith("buffer_load"))
            groups["vmem_st"] = sum(v for k, v in c.items() if k.startswith("global_store") or k.startswith("buffer_store"))
            groups["ds"] = sum(v for k, v in c.items() if k.startswith("ds_"))
            print(f"loop {bm.group(1)} [{start}..{i}]:", ", ".join(f"{k} {v}" for k, v in groups.items()))
            print("   ", ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1]) if k.startswith("v_")))


if __name__ == "__main__":
    main()
