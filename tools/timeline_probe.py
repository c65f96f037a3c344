#!/usr/bin/env python3
"""usage: timeline_probe.py <kernel_trace.csv> [n_last]  -- the last n_last kernel launches of a rocprofv3 --kernel-trace run as
a timeline: start offset, duration and the gap to the kernel before (us); then per kernel name the mean duration and mean
gap in front of it.  For the streaming regimes, where launch gaps rival the kernels."""
import collections
import csv
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = rows[-n_last:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:30]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    if i < 60:
        print(f"{(s - t0) / 1e3:10.1f} us  {name:28s} dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  grid {r.get('Grid_Size', '?')} wg {r.get('Workgroup_Size', '?')}")
    a = agg[name]
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2] += gap
    prev_end = max(prev_end or e, e)
span = (int(rows[-1]["End_Timestamp"]) - t0) / 1e3
print(f"--- {len(rows)} launches over {span:.1f} us")
busy = sum(a[1] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:28s} n {a[0]:5d}  mean dur {a[1] / a[0]:7.2f} us  mean gap before {a[2] / a[0]:6.2f} us  share of span {a[1] / span:5.3f}")
print(f"kernels busy {busy:.1f} us = {busy / span:.3f} of the span; gaps {span - busy:.1f} us")
