#!/usr/bin/env python3
"""Summarise rocprofv3 output (gpurun_out/...) into profiles/: copies the --stats kernel summary and converts the
separate FETCH_SIZE / WRITE_SIZE PMC passes into per-launch HBM bytes per kernel.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read (16 B/lane), so it is doubled; WRITE_SIZE is exact
for 16-B-per-lane streaming stores.

usage: summarize_profiles.py <tag> <stats_csv> <fetch_counter_csv> <write_counter_csv> [bench_log_with_the_json_line]

The optional bench log (stdout of the profiled bench.py run) tags the summary with the workload and the hash of the
kernel sources it was measured on (`_config`); bench.py only reports roofline.traffic from a summary whose tag matches
the run.
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0][:40]


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def bench_line(bench_log):
    line = None
    with open(bench_log) as f:
        for ln in f:
            if ln.startswith("{") and '"metric"' in ln:
                line = json.loads(ln)
    return line


def config_tag(bench_log):
    """the workload and the hash of the kernel sources a profiled bench.py run was measured on"""
    import hashlib, glob
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "convopeq_amd", "csrc", "*.hip"))):
        with open(p, "rb") as f:
            h.update(f.read())
    line = bench_line(bench_log)
    if not line:
        return None
    c = line["config"]
    return {"streams": c["streams_per_gpu"], "ir_len": c["ir_taps"], "block": c["block"],
            "blocks_per_call": c["blocks_per_call"], "partition": c["partition"],
            "schedule": "uniform" if c["schedule"].startswith("uniform") else "nuc", "eq": c["eq"],
            "kernel_sources": h.hexdigest()[:16]}


def check_fractions(line, where=""):
    """A roofline fraction above 1 means the line charges a kernel more bytes or flops than it can have moved: an accounting
    error in bench.py, not a fast kernel.  Returns the offending (name, value) pairs."""
    r = line.get("roofline") or {}
    cands = {"roofline.frac": r.get("frac"), "roofline.frac_arithmetic": r.get("frac_arithmetic"),
             "roofline.hbm_kernel.frac": (r.get("hbm_kernel") or {}).get("frac"),
             "roofline.hbm_regime.frac": (r.get("hbm_regime") or {}).get("frac"),
             "roofline.frac_of_pmc_traffic": r.get("frac_of_pmc_traffic")}
    for k, v in (line.get("kernels") or {}).items():
        if v.get("achieved_gbs") is not None:
            cands[f"kernels.{k}.achieved_gbs/8000"] = v["achieved_gbs"] / 8000.0
    return [(where + k, v) for k, v in cands.items() if v is not None and v > 1.0]


def main():
    if sys.argv[1] == "--check":        # summarize_profiles.py --check <file.jsonl | bench log> ...: exit 1 on any fraction > 1
        bad = []
        for path in sys.argv[2:]:
            with open(path) as f:
                for i, ln in enumerate(f):
                    if ln.startswith("{") and '"metric"' in ln:
                        bad += check_fractions(json.loads(ln), f"{path}:{i + 1}: ")
        for k, v in bad:
            print(f"FRACTION > 1: {k} = {v}", file=sys.stderr)
        sys.exit(1 if bad else 0)
    tag, stats, fetch, write = sys.argv[1:5]
    bench_log = sys.argv[5] if len(sys.argv) > 5 else None
    out_dir = os.environ.get("CPQ_PROFILES_OUT", os.path.join(ROOT, "profiles"))
    os.makedirs(out_dir, exist_ok=True)
    shutil.copy(stats, os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
    fs = per_kernel(fetch, "FETCH_SIZE")
    ws = per_kernel(write, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fs) | set(ws)):
        if not k.startswith("k_"):
            continue
        f_kib, nf = fs.get(k, (0.0, 0))
        w_kib, nw = ws.get(k, (0.0, 0))
        res[k] = {
            "fetch_size_kib_raw": f_kib, "write_size_kib_raw": w_kib, "launches_sampled": [nf, nw],
            "hbm_read_bytes_per_launch": 2.0 * f_kib * 1024.0,      # x2: gfx950 FETCH_SIZE half-count correction
            "hbm_write_bytes_per_launch": w_kib * 1024.0,
            "hbm_bytes_per_launch": 2.0 * f_kib * 1024.0 + w_kib * 1024.0,
        }
    res["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB units; FETCH_SIZE doubled "
                    "per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B). Infinity-Cache hits are included.")
    if bench_log:
        t = config_tag(bench_log)
        if t:
            res["_config"] = t
        bad = check_fractions(bench_line(bench_log) or {}, bench_log + ": ")
        if bad:
            for k, v in bad:
                print(f"FRACTION > 1: {k} = {v}", file=sys.stderr)
            sys.exit(1)
    with open(os.path.join(out_dir, f"{tag}_pmc_traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
