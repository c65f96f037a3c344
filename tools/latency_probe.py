#!/usr/bin/env python3
"""Per-call latency of the streaming regime as a real-time host sees it: one callback per call, the host waits for every call
(enqueue + device synchronise), 256 streams x 131072 taps, conv + EQ.  Prints mean / median / p99 / max of the wall time per
call in microseconds.  usage: python tools/latency_probe.py [--block 512] [--calls 400] [--any] [--streams 256]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import convopeq_amd as amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--block", type=int, default=512)
ap.add_argument("--calls", type=int, default=400)
ap.add_argument("--streams", type=int, default=256)
ap.add_argument("--any", action="store_true")
a = ap.parse_args()
S, B, L = a.streams, a.block, 131072
eng = amd.BatchedEngine(S, block_size=B, max_ir_len=L, max_blocks_per_call=1,
                        schedule=amd.CPQ_SCHED_REFERENCE_NUC, call_mode=amd.CPQ_CALLS_ANY if a.any else amd.CPQ_CALLS_WHOLE_BLOCKS)
eng.set_stream(torch.cuda.current_stream().cuda_stream)
for s in range(S):
    eng.set_impulse(s, bench.gen_ir(L, s, 0), bench.gen_ir(L, s, 1))
eng.set_eq_params(amd.CPQ_ALL_STREAMS, bench.bench_eq_params(amd, 0.2))
x = torch.from_numpy(np.stack([bench.gen_pcm(B, c // 2, c % 2) for c in range(2 * S)])).cuda()
y = torch.empty_like(x)
for _ in range(40):
    eng.process_device(x.data_ptr(), y.data_ptr(), B)
torch.cuda.synchronize()
t = np.empty(a.calls)
for k in range(a.calls):
    t0 = time.perf_counter()
    eng.process_device(x.data_ptr(), y.data_ptr(), B)
    eng.synchronize()                 # the engine's stream only (cpq_engine_synchronize), as a host would wait for its output
    t[k] = (time.perf_counter() - t0) * 1e6
print(f"block {B} {'any' if a.any else 'whole blocks'}: per-call wall time us: mean {t.mean():.1f} median {np.median(t):.1f} "
      f"p90 {np.percentile(t, 90):.1f} p99 {np.percentile(t, 99):.1f} max {t.max():.1f}  (budget of a {B}-sample callback at 48 kHz: {B / 48.0 * 1e3 / 1e3 * 1e3:.0f} us)")
eng.close()
