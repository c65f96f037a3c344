#!/usr/bin/env python3
"""Mean per launch of every SQ counter per kernel from a rocprofv3 --pmc counter_collection.csv (stdout: JSON)."""
import collections
import csv
import json
import re
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0][:40]


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k.startswith("k_"):
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in sorted(cs.items())}
           for k, cs in sorted(agg.items())}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
