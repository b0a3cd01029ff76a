#!/usr/bin/env python3
"""Build-time guard of the register / LDS budgets the pipelined receiver's co-residency rests on.

The chain's rate depends on which kernels FIT BESIDE a correlator workgroup on a CU: k_correlate_w64 takes 2 x 240 of a
SIMD's 512 VGPRs and 151 552 of the CU's 163 840 bytes of LDS, which leaves 32 VGPRs and 12 288 bytes -- exactly what the
two serial kernels (k_costas_cap, k_rot_checkpoints) are held to.  One more __shared__ word in the correlator, or a
compiler that needs 34 registers for the PLL, returns the chain to 50 Gsps with every test green.  So the budgets are
read back from the code objects inside the built library (the .hip_fatbin bundles -> the gfx950 ELF -> its
NT_AMDGPU_METADATA note, msgpack) and the build fails when one is exceeded.

Exit status: 0 = every budget kept, 1 = a budget exceeded (build() fails), 2 = the library could not be inspected
(unknown bundle layout; build() prints a warning and goes on: that is not an occupancy finding).

Usage: tools/check_occupancy.py [--lib path/to/lib.so] [--json out.json]
`--json` writes the table of every kernel of the library (profiles/r5_occupancy.json is that file)."""
import argparse
import json
import os
import re
import struct
import sys

try:  # not a declared dependency of anything else in the repository: a reader of its own stands in when it is absent
    import msgpack
    unpack_metadata = lambda desc: msgpack.unpackb(desc, raw=False, strict_map_key=False)
except ImportError:
    msgpack = None

    def unpack_metadata(desc):
        """the subset of MessagePack the AMDGPU metadata note uses (maps, arrays, strings, integers, booleans, nil, floats)"""
        def rd(p):
            b = desc[p]
            if b <= 0x7F:
                return b, p + 1
            if b >= 0xE0:
                return b - 256, p + 1
            if 0x80 <= b <= 0x8F:
                return rd_map(b & 15, p + 1)
            if 0x90 <= b <= 0x9F:
                return rd_arr(b & 15, p + 1)
            if 0xA0 <= b <= 0xBF:
                n = b & 31
                return desc[p + 1:p + 1 + n].decode(), p + 1 + n
            if b == 0xC0:
                return None, p + 1
            if b in (0xC2, 0xC3):
                return b == 0xC3, p + 1
            if b in (0xC4, 0xC5, 0xC6, 0xD9, 0xDA, 0xDB):
                w = {0xC4: 1, 0xC5: 2, 0xC6: 4, 0xD9: 1, 0xDA: 2, 0xDB: 4}[b]
                n = int.from_bytes(desc[p + 1:p + 1 + w], "big")
                raw = desc[p + 1 + w:p + 1 + w + n]
                return (raw.decode() if b >= 0xD9 else bytes(raw)), p + 1 + w + n
            if b == 0xCA:
                return struct.unpack_from(">f", desc, p + 1)[0], p + 5
            if b == 0xCB:
                return struct.unpack_from(">d", desc, p + 1)[0], p + 9
            if 0xCC <= b <= 0xCF:
                w = 1 << (b - 0xCC)
                return int.from_bytes(desc[p + 1:p + 1 + w], "big"), p + 1 + w
            if 0xD0 <= b <= 0xD3:
                w = 1 << (b - 0xD0)
                return int.from_bytes(desc[p + 1:p + 1 + w], "big", signed=True), p + 1 + w
            if b in (0xDC, 0xDD):
                w = 2 if b == 0xDC else 4
                return rd_arr(int.from_bytes(desc[p + 1:p + 1 + w], "big"), p + 1 + w)
            if b in (0xDE, 0xDF):
                w = 2 if b == 0xDE else 4
                return rd_map(int.from_bytes(desc[p + 1:p + 1 + w], "big"), p + 1 + w)
            raise ValueError(f"MessagePack type byte {b:#x} not handled")

        def rd_arr(n, p):
            out = []
            for _ in range(n):
                v, p = rd(p)
                out.append(v)
            return out, p

        def rd_map(n, p):
            out = {}
            for _ in range(n):
                k, p = rd(p)
                v, p = rd(p)
                out[k] = v
            return out, p

        return rd(0)[0]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "gr4-packet-modem_amd", "libgr4pm_hip.so")

# gfx950: 512 VGPRs per SIMD lane, 160 KiB of LDS per CU; VGPRs are allocated in granules of 8
SIMD_VGPRS, CU_LDS = 512, 160 * 1024

# (regular expression over the demangled-ish kernel name, budget).  Budgets: vgpr = .vgpr_count + .agpr_count upper bound,
# lds = .group_segment_fixed_size (exact when `lds_exact`), scratch = .private_segment_fixed_size upper bound.
BUDGETS = [
    # the production correlator (VAR = 98304 + 16384 = 114688; the A/B form with fixed shares has the same budget)
    (r"k_correlate_w64ILi(114688|245760)E", dict(vgpr=240, lds=151552, lds_exact=True, scratch=0)),
    # every other instantiation (strides above 1793: 98304; round 2's bin loops; timing-only variants of an EXPERIMENTS
    # build) still has to leave room for a second wave per SIMD (granule of 8: 248) and use no scratch
    (r"k_correlate_w64ILi", dict(vgpr=248, lds=151552, lds_exact=True, scratch=0)),
    # what runs BESIDE two correlator waves of a SIMD and the correlator's LDS
    (r"k_costas_capILi\dELi2E", dict(vgpr=32, lds=CU_LDS - 151552)),
    (r"k_rot_checkpoints", dict(vgpr=32, lds=CU_LDS - 151552)),
    # (round 6: the detector tail's workgroups of four waves -- one wave per SIMD, so 32 registers each -- that went
    # from one wave to several for the sake of one atomic per workgroup and stay beside the correlator as before)
    (r"k_(compact_pending|resolve_visited|scan_entries)", dict(vgpr=32, lds=CU_LDS - 151552, scratch=0)),
    # (the detector tail's streaming kernel: eight waves per SIMD since round 5)
    (r"k_candidates_waveILi12ELb1E", dict(vgpr=64, scratch=0)),
    # time-sliced against the correlator a CU at a time: their own occupancy targets
    (r"k_symbol_filter_fast", dict(vgpr=64)),
    (r"k_correlate_w64_oneILi0E", dict(vgpr=168, scratch=0)),  # three waves per SIMD
    (r"k_correlate_4096ILi1E", dict(vgpr=128, scratch=0)),     # four waves per SIMD
]


def code_objects(path):
    """every gfx950 ELF inside the library's .hip_fatbin section (one clang offload bundle per translation unit)"""
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    for m in re.finditer(re.escape(magic), data):
        p = m.start() + len(magic)
        (n,) = struct.unpack_from("<Q", data, p)
        p += 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tlen].decode()
            p += tlen
            if "amdgcn" in triple and size:
                out.append(data[m.start() + off:m.start() + off + size])
    return out


def kernels_of(elf):
    """amdhsa.kernels of one code object: walk the section headers to the SHT_NOTE sections, find the AMDGPU metadata note"""
    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "not a 64-bit ELF"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    found = []
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        if sh_type != 7:  # SHT_NOTE
            continue
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        p, end = off, off + size
        while p + 12 <= end:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) & ~3
            desc = elf[p:p + descsz]
            p += (descsz + 3) & ~3
            if name == b"AMDGPU" and ntype == 32:  # NT_AMDGPU_METADATA
                meta = unpack_metadata(desc)
                found.extend(meta.get("amdhsa.kernels", []))
    return found


def table(path):
    rows = []
    for elf in code_objects(path):
        for k in kernels_of(elf):
            vg, ag = int(k.get(".vgpr_count", 0)), int(k.get(".agpr_count", 0))
            alloc = (max(vg + ag, 1) + 7) // 8 * 8
            rows.append({
                "kernel": k[".name"], "vgpr": vg, "agpr": ag, "sgpr": int(k.get(".sgpr_count", 0)),
                "lds_bytes": int(k.get(".group_segment_fixed_size", 0)),
                "scratch_bytes": int(k.get(".private_segment_fixed_size", 0)),
                "spilled_vgprs": int(k.get(".vgpr_spill_count", 0)),
                "max_flat_workgroup_size": int(k.get(".max_flat_workgroup_size", 0)),
                "waves_per_simd_by_vgpr": min(8, SIMD_VGPRS // alloc),
            })
    rows.sort(key=lambda r: r["kernel"])
    return rows


def check(rows):
    errors, seen = [], set()
    for r in rows:
        for i, (pat, b) in enumerate(BUDGETS):
            if not re.search(pat, r["kernel"]):
                continue
            seen.add(i)
            regs = r["vgpr"] + r["agpr"]
            if "vgpr" in b and regs > b["vgpr"]:
                errors.append(f"{r['kernel']}: {regs} VGPRs, budget {b['vgpr']}")
            if "lds" in b:
                if b.get("lds_exact") and r["lds_bytes"] != b["lds"]:
                    errors.append(f"{r['kernel']}: {r['lds_bytes']} bytes of LDS, must be exactly {b['lds']}")
                elif r["lds_bytes"] > b["lds"]:
                    errors.append(f"{r['kernel']}: {r['lds_bytes']} bytes of LDS, budget {b['lds']}")
            if "scratch" in b and r["scratch_bytes"] > b["scratch"]:
                errors.append(f"{r['kernel']}: {r['scratch_bytes']} bytes of scratch, budget {b['scratch']}")
            break  # the first pattern that matches is the kernel's budget
    for i, (pat, _) in enumerate(BUDGETS):
        if i not in seen:
            errors.append(f"no kernel matches /{pat}/: the guard is looking for a kernel that was renamed or removed")
    return errors


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=LIB)
    ap.add_argument("--json")
    a = ap.parse_args()
    try:
        rows = table(a.lib)
    except Exception as e:  # a layout this reader does not know: "could not inspect" is not "budget exceeded"
        print(f"check_occupancy: WARNING: could not read the code objects of {a.lib}: {e!r}")
        return 2
    if not rows:
        compressed = b"CCOB" in open(a.lib, "rb").read()
        print("check_occupancy: WARNING: no code object found in", a.lib,
              "(compressed offload bundles: link with --no-offload-compress)" if compressed else "")
        return 2
    errors = check(rows)
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"library": os.path.relpath(a.lib, ROOT), "arch": "gfx950",
                       "budgets": [{"pattern": p, **b} for p, b in BUDGETS], "kernels": rows}, f, indent=1)
            f.write("\n")
    for e in errors:
        print("check_occupancy:", e)
    if not errors:
        print(f"check_occupancy: {len(rows)} kernels, every budget kept")
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
