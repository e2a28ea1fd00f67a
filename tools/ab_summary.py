#!/usr/bin/env python3
"""Summary table of a tools/ab.sh result file (the JSON lines tools/kbench.py printed): per
configuration and library the smallest minimum and the median of the last round, registers, frame hash.
usage: tools/ab_summary.py gpurun_out/<file> [...]"""
import collections
import json
import sys

for path in sys.argv[1:]:
    best = collections.OrderedDict()
    for line in open(path):
        if line.startswith("{"):
            d = json.loads(line)
            k = (d["config"], d["lib"])
            best[k] = min(best.get(k, (1e30,))[0], d["min_ms"]), d["median_ms"], d["sha"], d["vgprs"]
    for (cfg, lib), (mn, med, sha, vg) in best.items():
        print(f"{cfg:24s} {lib:40s} min {mn:10.4f} ms  median {med:10.4f}  vgprs {vg}  sha {sha}")
