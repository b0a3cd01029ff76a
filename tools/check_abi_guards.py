#!/usr/bin/env python3
"""Build-time guard for "no exceptions cross the ABI" (include/gr4pm_hip.h): every extern "C" function DEFINITION in
gr4-packet-modem_amd/csrc/*.hip must either be a function-try-block that ends in one of the GR4PM_ABI_CATCH* handlers
(csrc/common.hpp) or be on the short list of entry points that cannot throw (one-line accessors, `delete h`).

  tools/check_abi_guards.py            check (exit 1 and a list when an entry point is unguarded); run by build()
  tools/check_abi_guards.py --apply    rewrite unguarded multi-line definitions into function-try-blocks in place

The sources keep one layout for these definitions -- the signature ends in `)` on its own line(s), `{` and the closing
`}` stand alone in column 0 -- which is what makes both the check and the rewrite line-based."""
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gr4-packet-modem_amd", "csrc")
SIG = re.compile(r'^(extern "C" )?([A-Za-z_][\w \*]*?[\s\*])(gr4pm_\w+)\(')
# cannot throw: plain loads / stores / arithmetic on the arguments, no allocation, no C++ library call
NOTHROW_ONE_LINERS = {
    "gr4pm_last_error", "gr4pm_set_deferred_sync", "gr4pm_version", "gr4pm_multichannel_receiver_in_flight",
    "gr4pm_packet_receiver_inflight", "gr4pm_syncword_detection_syncword_samples_size",
    "gr4pm_syncword_detection_self_corr", "gr4pm_test_allocation_count",
}


def catch_for(ret):
    ret = ret.strip()
    if ret == "gr4pm_status":
        return "GR4PM_ABI_CATCH"
    if ret == "void":
        return "GR4PM_ABI_CATCH_VOID"
    if ret.endswith("*"):
        return "GR4PM_ABI_CATCH_RET(nullptr)"
    return "GR4PM_ABI_CATCH_RET(0)"


def scan(path, apply=False):
    src = open(path).read().split("\n")
    out, bad, i, in_ext = [], [], 0, False
    changed = False
    while i < len(src):
        line = src[i]
        if line.startswith('extern "C" {'):
            in_ext = True
        elif line.startswith('} // extern "C"'):
            in_ext = False
        m = SIG.match(line)
        if not (m and (in_ext or m.group(1))):
            out.append(line)
            i += 1
            continue
        ret, name = m.group(2), m.group(3)
        j = i
        while not src[j].rstrip().endswith((")", "{", "}", ";", "try")):
            j += 1
        tail = src[j].rstrip()
        if tail.endswith(";"):  # declaration
            out.extend(src[i:j + 1])
            i = j + 1
            continue
        if tail.endswith("}"):  # one-line definition
            if name not in NOTHROW_ONE_LINERS and "GR4PM_ABI_CATCH" not in tail:
                bad.append((name, i + 1, "one-line definition not on the no-throw list"))
            out.extend(src[i:j + 1])
            i = j + 1
            continue
        # multi-line definition: `{` or `try {` on the next line, body up to the first `}` in column 0
        k = j + 1
        if tail.endswith("try"):
            opener = "try"
        else:
            opener = src[k].rstrip()
        end = k
        while src[end] != "}":
            end += 1
        guarded = (opener.startswith("try") or tail.endswith("try")) and src[end + 1].startswith("GR4PM_ABI_CATCH")
        if guarded:
            out.extend(src[i:end + 2])
            i = end + 2
            continue
        if apply and opener == "{":
            out.extend(src[i:k])
            out.append("try {")
            out.extend(src[k + 1:end + 1])
            out.append(catch_for(ret))
            changed = True
        else:
            bad.append((name, i + 1, "no function-try-block / GR4PM_ABI_CATCH handler"))
            out.extend(src[i:end + 1])
        i = end + 1
    if apply and changed:
        open(path, "w").write("\n".join(out))
    return bad


def main():
    apply = "--apply" in sys.argv
    bad = []
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        for name, line, why in scan(path, apply):
            bad.append(f"{os.path.relpath(path, ROOT)}:{line}: {name}: {why}")
    if bad:
        print("extern \"C\" entry points without an exception guard:\n  " + "\n  ".join(bad))
        sys.exit(1)
    print("check_abi_guards: every extern \"C\" definition is guarded")


if __name__ == "__main__":
    main()
