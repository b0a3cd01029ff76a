#!/usr/bin/env python3
"""Build-time ISA guards for k_correlate_w64 (csrc/correlate_w64.hpp).  The kernel relies on three things the compiler
does not know about; this compiles csrc/syncword_detection.hip to assembly with the library's own flags (`make -C csrc
syncword_detection.s`: HIPCC / ARCH / ABL / EXTRA are the Makefile's, so an A/B build is checked against ITS compile)
and fails the build when, inside any k_correlate_w64 instantiation,

1. M0.  The exchange stores (ds_write_addtid_b32: LDS address = M0 + offset + 4 lane) and, in the LDS-DMA variants,
   the template copies rely on M0 holding what the kernel's own inline asm wrote.  Nothing else may write M0: any
   instruction whose FIRST operand is m0 (s_mov / s_add ..., but also v_readfirstlane_b32 m0, vN and v_readlane_b32
   m0, ...) outside an inline-asm block (;;#ASMSTART .. ;;#ASMEND), or inside one in any form but `s_mov_b32 m0, s<N>`.
2. Scratch.  The kernel sits at 240 VGPRs; a spill would land between an asm load group and its wait (3.) and is a
   performance cliff anyway: .private_segment_fixed_size must be 0.
3. Template loads.  load_template_half issues eight global_load_dwordx4 from inline asm; their results arrive only
   after a later, separate `s_waitcnt vmcnt(n)` asm.  The compiler does not know the registers are still being written:
   between the load group and the wait that covers it (vmcnt counts loads AND stores in issue order on gfx9) no
   instruction may read or write a destination register of the group (a copy, a split or a spill would silently use
   stale templates).  Checked along every path of the kernel's control-flow graph, inside the loop that issued the
   loads (a group is only issued when another iteration of its loop follows, so the loop's exit edge is not a path
   for it -- the one thing this tool takes from the source instead of the assembly).

Usage: tools/check_m0.py [file.s]   (without an argument it builds csrc/syncword_detection.s)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gr4-packet-modem_amd", "csrc")
VMEM = re.compile(r"^\s*(global_|buffer_|flat_|scratch_)(load|store|atomic)")
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def compile_to_asm():
    subprocess.check_call(["make", "-s", "-C", CSRC, "syncword_detection.s"])
    return os.path.join(CSRC, "syncword_detection.s")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def parse_kernels(path, kernel):
    """-> {kernel name: [(line no, text, in_asm)]} for the instantiations of `kernel`, {name: scratch bytes}"""
    bodies, scratch, name, in_asm, cur, depth = {}, {}, None, False, None, 0
    for no, raw in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", raw)
        if m:
            name = m.group(1) if kernel in m.group(1) else None
            if name:
                bodies[name] = []
            in_asm = False
            continue
        ms = re.match(r"^\s*\.amdhsa_kernel\s+(\S+)", raw)
        if ms:
            cur = ms.group(1)
        mp = re.match(r"^\s*\.amdhsa_private_segment_fixed_size\s+(\d+)", raw)
        if mp and cur and kernel in cur:
            scratch[cur] = int(mp.group(1))
        if name is None:
            continue
        if raw.startswith(".Lfunc_end"):
            name = None
            continue
        if ";;#ASMSTART" in raw:
            in_asm = True
            continue
        if ";;#ASMEND" in raw:
            in_asm = False
            continue
        # hipcc annotates every basic block with its innermost loop: "; %bb.25:  ; in Loop: Header=BB31_20 Depth=2"
        # (fall-through blocks) or ".LBB31_26:  ; in Loop: Header=BB31_11 Depth=1"
        if re.match(r"^(; %bb\.\d+:|\.LBB\d+_\d+:)", raw):
            md = re.search(r"Depth=(\d+)", raw)
            depth = int(md.group(1)) if md else 0
        line = raw.split(";")[0].rstrip()
        if not line.strip() or line.lstrip().startswith((".", "#")) and not line.strip().endswith(":"):
            continue
        bodies[name].append((no, line.strip(), in_asm, depth))
    return bodies, scratch


def check_loads(path, name, body, bad):
    """3.: walks the kernel's control-flow graph (both ways at conditional branches, loops until no new state shows
    up) with the queue of vector-memory operations in flight; returns the number of asm loads seen"""
    labels = {text[:-1]: i for i, (_, text, _, _) in enumerate(body) if text.endswith(":")}
    seen_states, work, groups, reported = set(), [(0, ())], set(), set()
    while work:
        pc, pending = work.pop()
        while pc < len(body):
            no, text, in_asm, depth = body[pc]
            # a template load is only issued when another iteration of its loop follows (correlate_w64.hpp: `if (bin + 1
            # < n_bins) load_template_half(...)`); the walk cannot know that the exit edge of that loop is not taken
            # then, so loads issued at loop depth d are dropped when the walk leaves to a shallower depth
            if pending and any(d > depth for _, _, d in pending):
                pending = tuple(e for e in pending if e[2] <= depth)
                while pending and not pending[0][0]:
                    pending = pending[1:]
            if text.endswith(":"):
                key = (pc, pending)
                if key in seen_states:
                    break
                seen_states.add(key)
                pc += 1
                continue
            op = text.split()[0]
            if op == "s_endpgm":
                break
            if op == "s_branch":
                pc = labels[text.split()[1]]
                continue
            if op.startswith("s_cbranch"):
                work.append((labels[text.split()[1]], pending))
                pc += 1
                continue
            if op == "s_waitcnt":
                mv = re.search(r"vmcnt\((\d+)\)", text)
                if mv:
                    keep = int(mv.group(1))
                    pending = pending[len(pending) - keep:] if keep < len(pending) else pending
                    if keep == 0:
                        pending = ()
                    while pending and not pending[0][0]:
                        pending = pending[1:]
                pc += 1
                continue
            if VMEM.match(text):
                dst = frozenset()
                if in_asm and text.startswith("global_load_dwordx4"):
                    dst = frozenset(regs_of(text.split(",")[0]))
                    groups.add(no)
                    used = regs_of(",".join(text.split(",")[1:]))
                else:
                    used = regs_of(text)
            else:
                dst, used = None, regs_of(text)
            for regs, at, _ in pending:
                hit = regs & used
                if hit and (no, at) not in reported:
                    reported.add((no, at))
                    bad.append(f"{path}:{no}: {name}: v{min(hit)} is still being loaded (asm load at line {at}, no "
                               f"s_waitcnt vmcnt covers it on this path): {text}")
            if dst is not None:
                pending = (pending + ((dst, no if dst else 0, depth if dst else 0),))[-63:]
            # operations older than the oldest asm load still in flight do not matter: without this the walk would
            # tell paths apart by their stores
            while pending and not pending[0][0]:
                pending = pending[1:]
            pc += 1
    return len(groups)


def check(path, kernel="k_correlate_w64"):
    bodies, scratch = parse_kernels(path, kernel)
    bad, sites, groups = [], 0, 0
    writes_m0 = re.compile(r"^(\w+)\s+m0\b")
    for name, body in bodies.items():
        for no, text, in_asm, _ in body:
            if not writes_m0.match(text):
                continue
            if not in_asm:
                bad.append(f"{path}:{no}: compiler-generated write of m0 in {name}: {text}")
            elif not re.match(r"^s_mov_b32\s+m0,\s*s\d+$", text):
                bad.append(f"{path}:{no}: unexpected form of the m0 write in {name}: {text}")
            else:
                sites += 1
        groups += check_loads(path, name, body, bad)
    for k, v in scratch.items():
        if v != 0:
            bad.append(f"{path}: {k} uses {v} bytes of scratch (spills): the kernel must stay in registers")
    return len(bodies), sites, groups, len(scratch), bad


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else compile_to_asm()
    seen, sites, groups, n_scr, bad = check(path)
    if seen == 0:
        bad.append("no k_correlate_w64 instantiation found in the assembly")
    if n_scr == 0:
        bad.append("no .amdhsa_private_segment_fixed_size found for k_correlate_w64")
    for b in bad[:40]:
        print(b, file=sys.stderr)
    print(f"check_m0: {seen} k_correlate_w64 instantiations: {sites} m0 writes, all inside the kernel's own asm; "
          f"{n_scr} kernels without scratch; {groups} asm template loads, none touched before its s_waitcnt"
          if not bad else f"check_m0: FAILED ({len(bad)} findings)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
