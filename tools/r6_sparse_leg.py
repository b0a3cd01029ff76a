#!/usr/bin/env python3
"""bench.py's `sparse` leg alone, in a process of its own (round 6: is the leg's rate the tool's rate?)"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
pkg = ge.load_package()
device = torch.device("cuda", 0)
rrc = bench.unit_norm_rrc(pkg)
r = bench.sparse_leg(pkg, device, rrc, passes=int(sys.argv[1]) if len(sys.argv) > 1 else 48)
print(json.dumps(r["streams"]["dense_packets"]))
