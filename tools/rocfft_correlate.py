#!/usr/bin/env python3
"""Builds and runs tools/rocfft_correlate.hip (the rocFFT multi-kernel correlator: baseline only)
beside k_correlate_w64 on the same sizes, on the GPU box:

    tools/rocfft_correlate.py [items=2^26] [bins=4]        timing of both (HIP events), one JSON line
    tools/rocfft_correlate.py --pmc <out-dir> [items] [bins]   additionally FETCH_SIZE / WRITE_SIZE of both under
        rocprofv3 (separate --pmc passes, kernel-trace only), summed per call -> bytes per input sample

Writes nothing itself; `> profiles/r4_rocfft_baseline.json` keeps the line."""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:]]
pmc_dir = None
if args and args[0] == "--pmc":
    pmc_dir = args[1]
    args = args[2:]
items = int(args[0]) if args else 1 << 26
bins = int(args[1]) if len(args) > 1 else 4
exe = os.path.join(ROOT, "tools", "rocfft_correlate.bin")
src = os.path.join(ROOT, "tools", "rocfft_correlate.hip")
if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-o", exe, src, "-L/opt/rocm/lib", "-lrocfft",
                           "-Wl,-rpath,/opt/rocm/lib"])
base = json.loads(subprocess.check_output([exe, str(items), str(bins), "5"], text=True).strip().splitlines()[-1])
ours = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "bench_correlate.py"), str(items), "5", str(bins)],
                               text=True, stderr=subprocess.DEVNULL).strip().splitlines()[-1]
out = {"rocfft_multi_kernel": base, "k_correlate_w64": ours}


def pmc_bytes(cmd, counter, out_dir):
    """sum of a counter over every kernel dispatch of one process run (rocprofv3 --pmc, its own pass)"""
    out_dir = os.path.abspath(out_dir)
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "--"] + cmd,
                   cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    per_kernel = {}
    for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                k = r["Kernel_Name"].split("(")[0][:48]
                per_kernel.setdefault(k, [0.0, 0])
                per_kernel[k][0] += float(r["Counter_Value"])
                per_kernel[k][1] += 1
    return per_kernel


if pmc_dir:
    # the guide's recipe (MI355X_MICROARCH.md, HBM / rocprofv3): separate passes; FETCH_SIZE and WRITE_SIZE are KiB, FETCH_SIZE
    # doubled on gfx950 (as tools/pmc_other_kernels.py and profiles/r3_k_correlate_hbm_traffic.json do)
    n_blocks = (items - 2048) // 1752 + 1
    samples = n_blocks * 1752
    calls = 1 + 1  # the binary runs one warm-up call + `reps` calls; reps = 1 here
    traffic = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        pk = pmc_bytes([exe, str(items), str(bins), "1"], counter, os.path.join(pmc_dir, "rocfft_" + counter))
        scale = 2048.0 if counter == "FETCH_SIZE" else 1024.0
        traffic[counter] = {k: round(v[0] * scale / calls / samples, 2) for k, v in pk.items()}
        traffic[counter]["total_bytes_per_sample"] = round(sum(v[0] for v in pk.values()) * scale / calls / samples, 2)
    out["rocfft_multi_kernel"]["hbm_bytes_per_sample"] = traffic
    out["k_correlate_w64_hbm_bytes_per_sample"] = "12.29 (profiles/r3_k_correlate_hbm_traffic.json: 8.28 read + 4.01 written)"
print(json.dumps(out))
