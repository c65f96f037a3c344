#!/usr/bin/env python3
"""Probe: do the FMA-bound MAC path and the VALU-bound SVF cascade overlap when they run on two HIP streams?
Two engines (conv only / EQ only, independent data) timed alone and together.  Decides whether a call-pipelined
entry point (EQ of call i concurrent with the convolution of call i+1) would pay."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import convopeq_amd as amd
import bench

S, T, B, L = 256, 64, 512, 131072
n = T * B
conv = amd.BatchedEngine(S, max_ir_len=L, max_blocks_per_call=T)
eq = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
conv.set_stream(sa.cuda_stream)
eq.set_stream(sb.cuda_stream)
ir = bench.gen_ir(L, 0, 0)
conv.set_impulse(amd.CPQ_ALL_STREAMS, ir, ir)        # shared IR: same FMA work, quick set-up
eq.set_eq_params(amd.CPQ_ALL_STREAMS, bench.bench_eq_params(amd, 0.2))
x = torch.from_numpy(np.tile(bench.gen_pcm(n, 0, 0), (2 * S, 1))).cuda()
y1, y2 = torch.empty_like(x), torch.empty_like(x)

def run(do_conv, do_eq, steps=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if do_conv: conv.conv_process_device(x.data_ptr(), y1.data_ptr(), n)
        if do_eq: eq.eq_process_device(x.data_ptr(), y2.data_ptr(), n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

for _ in range(2):
    a, b, c = run(True, False), run(False, True), run(True, True)
print(f"conv alone {a:.3f} ms, eq alone {b:.3f} ms, sum {a + b:.3f} ms, concurrent {c:.3f} ms per step")
