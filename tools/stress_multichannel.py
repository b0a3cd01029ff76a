"""soak test of gr4pm_multichannel_receiver: many batches of random sizes through the pipelined form (one launch for
all channels, input read in place, up to four batches in flight) and through the synchronous per-channel form; every
symbol and every tag must be identical.  python tools/stress_multichannel.py [channels] [batches] [seed]"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
pkg = importlib.import_module("gr4-packet-modem_amd")
import _signals as sig  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 200
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rng = np.random.default_rng(seed)
n_max = 1 << 16
total = n_max * 8
locs = sorted(rng.choice(np.arange(500, total // 4 - 2500, 1700), size=40, replace=False).tolist())
base, _ = sig.qa_syncword_stream(total // 4, locs, 0.0, seed=seed)
base = (0.7 * base[:total] + sig.awgn(total, 0.05, seed + 1)).astype(np.complex64)
k = np.arange(total, dtype=np.float64)
xs = np.stack([np.roll(base, 997 * c) * np.exp(1j * ((-0.03 + 0.06 * c / max(C - 1, 1)) * k)) for c in range(C)])
xd = torch.from_numpy(xs.astype(np.complex64)).cuda()


def same_tags(a, b):
    return a.size == b.size and all(a[f].tobytes() == b[f].tobytes() for f in a.dtype.names)


os.environ["GR4PM_MC_PER_CHANNEL"] = "1"
sync = pkg.NativeMultiChannelReceiver(C, max_items=n_max, tags_cap=256, workers=4)
del os.environ["GR4PM_MC_PER_CHANNEL"]
if os.environ.get("STRESS_PIPE_PER_CHANNEL") == "1":
    os.environ["GR4PM_MC_PER_CHANNEL"] = "1"
pipe = pkg.NativeMultiChannelReceiver(C, max_items=n_max, tags_cap=256, workers=4,
                                      output_ring=os.environ.get("STRESS_RING", "1") == "1")
os.environ.pop("GR4PM_MC_PER_CHANNEL", None)
if os.environ.get("STRESS_INPLACE", "1") == "1":
    pipe.set_input_in_place(True)
depth = int(os.environ.get("STRESS_DEPTH", "4"))
pos, want, got, parts = 0, [], [], []
for b in range(n_batches):
    n = int(os.environ["STRESS_FIXED"]) if os.environ.get("STRESS_FIXED") else int(rng.integers(4096, n_max))
    if pos + n > total:
        pos = 0  # wrap: both receivers see the same discontinuity
    w = xd[:, pos:pos + n]
    parts.append(w)
    r = sync.process_bulk(w, 200)  # the same strided rows for both (a copy made on torch's stream would race the library's)
    want.append([(x["consumed"], x["symbols"].clone(), x["tags"], x["detector_tags"]) for x in r])
    pos += r[0]["consumed"]
    if pipe.in_flight() >= depth:
        got.append([(x["consumed"], x["symbols"].clone(), x["tags"], x["detector_tags"]) for x in pipe.collect()])
    pipe.submit(w, 200)
while pipe.in_flight():
    got.append([(x["consumed"], x["symbols"].clone(), x["tags"], x["detector_tags"]) for x in pipe.collect()])
assert len(got) == len(want) == n_batches
n_tags, bad = 0, 0
for b in range(n_batches):
    for c in range(C):
        g, w_ = got[b][c], want[b][c]
        what = []
        if g[0] != w_[0]:
            what.append(f"consumed {g[0]} != {w_[0]}")
        if g[1].numel() != w_[1].numel():
            what.append(f"symbols {g[1].numel()} != {w_[1].numel()}")
        elif not torch.equal(g[1].view(torch.int64), w_[1].view(torch.int64)):
            d = (g[1].view(torch.int64) != w_[1].view(torch.int64)).nonzero().flatten()
            what.append(f"{d.numel()} of {g[1].numel()} symbols differ, first {int(d[0])} last {int(d[-1])}")
        if not same_tags(g[2], w_[2]):
            what.append(f"tags {g[2].size} vs {w_[2].size}")
        if not same_tags(g[3], w_[3]):
            what.append(f"detector tags {g[3].size} vs {w_[3].size}")
        if what:
            bad += 1
            if bad <= 12:
                print(f"batch {b} (n = {parts[b].shape[1]}) channel {c}: " + "; ".join(what))
        n_tags += g[2].size
if bad:
    print(f"stress FAILED: {bad} (batch, channel) pairs differ")
    sys.exit(1)
print(f"stress ok: {C} channels, {n_batches} batches, {n_tags} tags")
