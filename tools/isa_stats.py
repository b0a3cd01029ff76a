#!/usr/bin/env python3
"""Static instruction statistics of a kernel from hipcc's -S output: per basic block, how many packed / other
VALU, LDS, vector-memory, s_nop and scalar instructions.  tools/isa_stats.py file.s kernel-name-substring
(build the .s with: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -S
--cuda-device-only -I include -o /tmp/sd.s gr4-packet-modem_amd/csrc/syncword_detection.hip)"""
import re
import sys
from collections import Counter


def kernels(path):
    name, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name and line.startswith(".Lfunc_end"):
            yield name, body
            name = None
            continue
        if name is not None:
            body.append(line.rstrip("\n"))


def classify(op):
    if op.startswith("v_pk_"):
        return "v_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_nop":
        return "s_nop"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, want = sys.argv[1], sys.argv[2]
    for name, body in kernels(path):
        if want not in name:
            continue
        blocks, cur, label = [], Counter(), "entry"
        ops = Counter()
        for line in body:
            t = line.strip()
            if not t or t.startswith((";", ".", "//")) and not t.startswith(".LBB"):
                continue
            m = re.match(r"^(\.LBB\w+):", t)
            if m:
                blocks.append((label, cur))
                cur, label = Counter(), m.group(1)
                continue
            op = t.split()[0]
            cur[classify(op)] += 1
            ops[op] += 1
        blocks.append((label, cur))
        total = Counter()
        for _, c in blocks:
            total.update(c)
        print(name)
        print("  total:", dict(total))
        for lab, c in blocks:
            if sum(c.values()) >= 60:
                print(f"  {lab:>12}: " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
        if len(sys.argv) > 3:
            print("  ops:", ops.most_common(25))


if __name__ == "__main__":
    main()
