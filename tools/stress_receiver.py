"""soak test of gr4pm_packet_receiver: many batches of random sizes (announced up to two ahead) through the pipelined
native receiver and through the sequential one; symbols, tags and, with a mode argument, LLRs / packets must be
identical.  python tools/stress_receiver.py [batches] [seed] [plain|soft|decode|lean]
lean (round 6): the pipelined packets_only receiver against the sequential FULL form of the decoding receiver -- packets,
packet lengths, header messages and tags must be identical (the streams the packets_only form does not write are skipped)."""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("gr4-packet-modem_amd")
import bench  # noqa: E402

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else "plain"
rng = np.random.default_rng(seed)
dev = torch.device("cuda")
n_max = 1 << 19
total = n_max * 6
lean = mode == "lean"
if lean:
    mode = "decode"
if mode == "decode":
    x, _ = bench.packet_stream(pkg, total, seed=seed, device=dev)
else:
    x, _ = bench.burst_stream(pkg, total, bench.unit_norm_rrc(pkg), seed=seed, device=dev)
torch.cuda.synchronize()
kw = dict(max_items=n_max, tags_cap=1024, soft_bits=mode != "plain", decode_headers=mode == "decode")
seq = pkg.NativePacketReceiver(pipelined=False, **kw)
pipe = pkg.NativePacketReceiver(pipelined=True, packets_only=lean, **kw)
lo = 40000 if mode == "decode" else 4096
chunks, pos = [], 0
for b in range(n_batches):
    n = int(rng.integers(lo, n_max))
    if pos + n > total:
        pos = 0
    chunks.append(x[pos:pos + n])
    pos += ((n - 2048) // 1752 + 1) * 1752


def keep(r):
    out = {"consumed": r["consumed"], "symbols": r["symbols"].clone(), "tags": r["tags"].copy()}
    for k in ("llr", "packets"):
        if k in r and r[k] is not None:
            out[k] = r[k].clone()
    if "packet_lengths" in r:
        out["packet_lengths"] = np.array(r["packet_lengths"]).copy()
    if "header_messages" in r:
        out["header_messages"] = r["header_messages"].copy()
    return out


want = []
for c in chunks:
    r = seq.process_bulk(c, 1500)
    want += [keep(q) for q in ([r] if r is not None else [])]
want += [keep(q) for q in seq.flush()]
got, announced = [], 0
for k, c in enumerate(chunks):
    while announced < min(k + 2, len(chunks) - 1):
        announced += 1
        pipe.announce(chunks[announced])
    r = pipe.process_bulk(c, 1500)
    if r is not None:
        got.append(keep(r))
got += [keep(q) for q in pipe.flush()]
assert len(got) == len(want) == n_batches, (len(got), len(want))
n_tags = 0
for b in range(n_batches):
    g, w = got[b], want[b]
    assert g["consumed"] == w["consumed"], (b, "consumed")
    assert torch.equal(g["symbols"].view(torch.int64), w["symbols"].view(torch.int64)), (b, "symbols")
    assert g["tags"].size == w["tags"].size and all(  # field by field: the records carry padding bytes
        g["tags"][f].tobytes() == w["tags"][f].tobytes() for f in w["tags"].dtype.names), (b, "tags")
    for k in ("llr", "packets"):
        if k in w and k in g:
            assert torch.equal(g[k].view(torch.uint8), w[k].view(torch.uint8)), (b, k)
    if "header_messages" in w:
        assert all(g["header_messages"][f].tobytes() == w["header_messages"][f].tobytes()
                   for f in w["header_messages"].dtype.names), (b, "header_messages")
    if "packet_lengths" in w:
        assert np.array_equal(g["packet_lengths"], w["packet_lengths"]), (b, "packet_lengths")
    n_tags += g["tags"].size
print(f"stress ok ({'lean' if lean else mode}): {n_batches} batches, {n_tags} tags")
