#!/usr/bin/env python3
"""usage (GPU box, library built from tools/variants/make_wg_trace.py's variant): wg_trace_probe.py <label> [bench.py args]
Runs bench.py's timed loop in this process, then reads the per-workgroup trace of the LAST k_svf_cascade_tpv<8> launch:
where each workgroup ran (XCD, CU), when it started and ended (100 MHz real-time counter), its shader clock (s_memtime
ticks over that interval) and when it finished each span.  Prints the balance of the launch: occupancy of the workgroup
slots, spread of the end times, per-XCD clock and duration."""
import collections
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
label = sys.argv[1]
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-parity"] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
from convopeq_amd import _capi  # noqa: E402

lib = _capi.load() if hasattr(_capi, "load") else _capi._lib
NW = 4096
rec = np.dtype([("t0", "<u8"), ("t1", "<u8"), ("c0", "<u8"), ("c1", "<u8"), ("hw", "<u4"), ("xcc", "<u4"), ("block", "<u4"),
                ("n", "<u4"), ("spanT", "<u8", (96,)), ("spanId", "<u4", (96,)), ("spanC", "<u8", (96,)), ("last", "<u8"), ("phase", "<u8", (8,))])
assert rec.itemsize == 2040
buf = np.zeros(NW, dtype=rec)
fn = lib.cpq_diag_wg_trace
fn.argtypes = [C.c_void_p, C.c_size_t]
fn.restype = C.c_int
assert fn(buf.ctypes.data, buf.nbytes) == 0
w = buf[buf["t1"] > 0]
print(f"== {label}: {len(w)} workgroups traced")
t0 = w["t0"].min()
T = (w["t1"].max() - t0) / 100.0        # us
dur = (w["t1"] - w["t0"]) / 100.0
clk = (w["c1"] - w["c0"]) / np.maximum(dur, 1e-9) / 1e3     # GHz if s_memtime ticks at the shader clock
start = (w["t0"] - t0) / 100.0
end = (w["t1"] - t0) / 100.0
print(f"launch {T:.1f} us; workgroup duration mean {dur.mean():.1f} min {dur.min():.1f} max {dur.max():.1f} us; "
      f"slot occupancy sum(dur) / (n x launch) = {dur.sum() / (len(w) * T):.3f}")
print(f"start: max {start.max():.1f} us, {np.sum(start > 5)} workgroups later than 5 us; end percentiles (us): "
      + " ".join(f"p{p}={np.percentile(end, p):.0f}" for p in (0, 5, 25, 50, 75, 95, 100)))
print(f"s_memtime ticks / us: mean {clk.mean() * 1e3:.1f} min {clk.min() * 1e3:.1f} max {clk.max() * 1e3:.1f}")
xcc = w["xcc"] & 0xF
cu = (w["hw"] >> 8) & 0xF
sh = (w["hw"] >> 12) & 0x1
se = (w["hw"] >> 13) & 0x7
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"  XCD {x}: {m.sum():4d} wgs  dur mean {dur[m].mean():8.1f} min {dur[m].min():8.1f} max {dur[m].max():8.1f} us  end mean {end[m].mean():8.1f}  "
          f"ticks/us {clk[m].mean() * 1e3:7.1f}  tasks/wg {w['n'][m].mean():.1f}")
place = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
print(f"distinct (XCD, SE, SH, CU) places {len(place)}; workgroups per place: {sorted(collections.Counter(place.values()).items())}")
# per-span pace of the workgroups: time per span in the first and in the last quarter of each workgroup's spans
n = np.minimum(w["n"], 96)
if n.min() >= 8:
    sp = w["spanT"].astype(np.int64)
    first = np.array([(sp[i, n[i] // 4] - sp[i, 0]) / max(n[i] // 4, 1) for i in range(len(w))]) / 100.0
    last = np.array([(sp[i, n[i] - 1] - sp[i, n[i] - 1 - n[i] // 4]) / max(n[i] // 4, 1) for i in range(len(w))]) / 100.0
    print(f"us per span: first quarter mean {first.mean():.2f} (min {first.min():.2f} max {first.max():.2f}); last quarter mean {last.mean():.2f} "
          f"(min {last.min():.2f} max {last.max():.2f})")
    # shader clock over the launch: s_memtime ticks per us between consecutive span ends, mean over the workgroups
    k = int(n.min())
    dT = np.diff(w["spanT"][:, :k].astype(np.int64), axis=1) / 100.0
    dC = np.diff(w["spanC"][:, :k].astype(np.int64), axis=1)
    print("us per span by span index (every 4th):", np.round(dT.mean(axis=0)[::4], 1).tolist())
    print("s_memtime ticks per us by span index (every 4th):", np.round((dC / np.maximum(dT, 1e-9)).mean(axis=0)[::4], 0).tolist())
    # pairs sharing a CU: does the partner's end speed the other one up?
    order = np.argsort(end)
    print("slowest 5 workgroups:", [(int(w['block'][i]), int(xcc[i]), int(se[i]), int(cu[i]), round(float(end[i]), 1)) for i in order[-5:]])
    print("fastest 5 workgroups:", [(int(w['block'][i]), int(xcc[i]), int(se[i]), int(cu[i]), round(float(end[i]), 1)) for i in order[:5]])
ph = w["phase"].astype(np.float64) / 100.0
tasks = np.maximum(w["n"], 1)[:, None]
print("us per task by phase (0 between tasks, 1 ticket, 2 tables / masks / until the span load is issued, 3 span load, 4 band loop, 5 store): "
      + " ".join(f"{v:.2f}" for v in (ph / tasks).mean(axis=0)[:6]) + f"   sum {(ph / tasks).sum(axis=1).mean():.2f}")
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", f"wg_trace_{label}.npz")
np.savez_compressed(out, t0=w["t0"], t1=w["t1"], c0=w["c0"], c1=w["c1"], hw=w["hw"], xcc=w["xcc"], block=w["block"], n=w["n"],
                    spanT=w["spanT"], spanId=w["spanId"], spanC=w["spanC"])
