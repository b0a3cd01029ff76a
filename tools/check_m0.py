#!/usr/bin/env python3
"""Build-time guard for the M0 assumption of k_correlate_w64 (csrc/correlate_w64.hpp): the kernel's exchange stores
(ds_write_addtid_b32: LDS address = M0 + offset + 4 lane) and, in the LDS-DMA variants, the template copies rely on M0
holding what the kernel's own inline asm wrote.  Nothing else may write M0 inside those kernels: hipcc would do so for
LDS-DMA builtins, GWS / sendmsg, v_movrel or v_interp.  This compiles csrc/syncword_detection.hip to assembly with the
library's flags and fails if, inside any k_correlate_w64 instantiation, an instruction OUTSIDE an inline-asm block
(;;#ASMSTART .. ;;#ASMEND) writes m0, or an asm block writes it in any form but `s_mov_b32 m0, s<N>`.
Usage: tools/check_m0.py [file.s]   (without an argument it compiles to a temporary file)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gr4-packet-modem_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-fno-slp-vectorize"]


def compile_to_asm(path):
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", "-o", path,
                                                             os.path.join(CSRC, "syncword_detection.hip")],
                          stderr=subprocess.DEVNULL)


def check(path, kernel="k_correlate_w64"):
    name, in_asm, bad, seen, sites = None, False, [], 0, 0
    writes_m0 = re.compile(r"^\s*(s_\w+)\s+m0\b")
    for no, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1) if kernel in m.group(1) else None
            seen += name is not None
            in_asm = False
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            name = None
            continue
        if ";;#ASMSTART" in line:
            in_asm = True
        elif ";;#ASMEND" in line:
            in_asm = False
        w = writes_m0.match(line)
        if not w:
            continue
        if not in_asm:
            bad.append(f"{path}:{no}: compiler-generated write of m0 in {name}: {line.strip()}")
        elif not re.match(r"^\s*s_mov_b32\s+m0,\s*s\d+\s*$", line):
            bad.append(f"{path}:{no}: unexpected form of the m0 write in {name}: {line.strip()}")
        else:
            sites += 1
    return seen, sites, bad


def main():
    if len(sys.argv) > 1:
        seen, sites, bad = check(sys.argv[1])
    else:
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "syncword_detection.s")
            compile_to_asm(path)
            seen, sites, bad = check(path)
    if seen == 0:
        bad.append("no k_correlate_w64 instantiation found in the assembly")
    for b in bad:
        print(b, file=sys.stderr)
    print(f"check_m0: {seen} k_correlate_w64 instantiations, {sites} m0 writes, all inside the kernel's own asm"
          if not bad else "check_m0: FAILED")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
