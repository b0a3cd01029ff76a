#!/bin/bash
# the headline step with the correlator's bit-identical forms (GR4PM_W64_VARIANT): default (templates from global
# memory into registers, 240 VGPRs) | 65536 (planar loop, templates by LDS-DMA, 210 VGPRs: leaves 80 registers per SIMD
# to the kernels of the other stages) | 0 (round 2's loop, 214 VGPRs)
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for v in -1 65536 0; do
  GR4PM_W64_VARIANT=$v python3 $R/bench.py --no-cpu-baseline --no-channels-leg --no-config5-leg --no-host-stream-leg --no-per-bins --no-sparse-leg --no-pmc-traffic --repeats 3 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant $v', d['value'], d['ms_per_step'], d['roofline']['frac'])"
done; done
