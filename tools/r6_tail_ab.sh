#!/bin/bash
# Same-box A/B of round 6's detector-tail work: the library built from the commit in front of it (3b8143c: `git worktree
# add /tmp/pretail 3b8143c && make -C /tmp/pretail/gr4-packet-modem_amd/csrc && cp
# /tmp/pretail/gr4-packet-modem_amd/libgr4pm_hip.so tools/ab/libgr4pm_hip_before_tail.so`; the C-ABI is the same) against
# the library in the tree: headline (2^28 samples per step, pipelined), config 2 (64 channels x 2^22) with its latency
# table, and the headline chain on ONE stream (--no-pipeline --no-lookahead: the sum of the kernels), two rounds each.
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in $R/tools/ab/libgr4pm_hip_before_tail.so ""; do
  name="${lib:+before the tail work}"; name="${name:-round 6 final       }"
  GR4PM_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-config5-leg --no-host-stream-leg --no-per-bins --no-sparse-leg --no-pmc-traffic --repeats 3 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config2']
print('$name', 'headline', d['value'], 'Msps', d['ms_per_step'], 'ms | config 2', c['value'], 'Msps', c['ms_per_step'], 'ms | latency of a batch of 4096 / 65536 / 262144 samples per channel:', ' / '.join(str(c['latency']['batches'][k]['latency_ms']) for k in ('4096','65536','262144') if k in c['latency']['batches']), 'ms')"
  GR4PM_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --no-sparse-leg --no-pmc-traffic --repeats 3 --no-pipeline --no-lookahead 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', 'headline chain on one stream', d['value'], 'Msps', d['ms_per_step'], 'ms')"
done; done
