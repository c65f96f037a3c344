#!/usr/bin/env python3
"""Mean per launch of every SQ counter per kernel from a rocprofv3 --pmc counter_collection.csv (stdout: JSON).
usage: summarize_sq.py <counter_collection.csv> [bench_log_with_the_json_line]
The optional bench log (stdout of the profiled bench.py run) tags the summary with the workload and the hash of the kernel
sources (`_config`, as tools/summarize_profiles.py does): bench.py reports roofline.valu_issue_util only from a summary whose
tag matches the run."""
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
    if len(sys.argv) > 2:
        import os
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from summarize_profiles import config_tag
        tag = config_tag(sys.argv[2])
        if tag:
            out["_config"] = tag
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
