#!/usr/bin/env python3
"""Where does a kernel spill?  Histogram of scratch_load / scratch_store (and v_accvgpr_*, v_writelane / v_readlane) instructions by
source line, from an assembly listing compiled with -gline-tables-only -S --cuda-device-only.
   python tools/spill_lines.py file.s [min_count]"""
import collections
import re
import sys

files = {}
hist = collections.Counter()
acc = collections.Counter()
lane = collections.Counter()
cur = None
for line in open(sys.argv[1]):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
    if m:
        cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        continue
    t = line.strip()
    if t.startswith(("scratch_store", "scratch_load")):
        hist[cur] += 1
    elif t.startswith("v_accvgpr"):
        acc[cur] += 1
    elif t.startswith(("v_writelane", "v_readlane")):
        lane[cur] += 1
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 3
print("scratch ops: %d   accvgpr moves: %d   lane moves: %d" % (sum(hist.values()), sum(acc.values()), sum(lane.values())))
for name, h in (("scratch", hist), ("accvgpr", acc), ("lane", lane)):
    print("--", name)
    for (k, v) in h.most_common(40):
        if v >= mn:
            print("  %5d  %s:%s" % (v, k[0] if k else "?", k[1] if k else "?"))
