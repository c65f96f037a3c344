#!/usr/bin/env python3
"""Experiment: do the HBM-bound convolver kernels of one half of the streams run beside the fp64-bound EQ kernel of the
other half when the two halves are enqueued on two HIP streams?  Two engines of S/2 streams each (independent streams
shard trivially), each on its own HIP stream, calls enqueued alternately; aggregate throughput against ONE engine of S
streams on one stream.  usage: python tools/overlap_two_engines.py [S] [blocks_per_call] [partition]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import convopeq_amd as amd  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
P = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
L, B = 131072, 512
n = T * B


def make(streams, first_id, hip_stream):
    eng = amd.BatchedEngine(streams, block_size=B, max_ir_len=L, max_blocks_per_call=T, partition_size=P)
    eng.set_stream(hip_stream.cuda_stream)
    for s in range(streams):
        eng.set_impulse(s, bench.gen_ir(L, first_id + s, 0), bench.gen_ir(L, first_id + s, 1))
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, bench.bench_eq_params(amd, 0.2))
    host = np.empty((2 * streams, n))
    for s in range(streams):
        for ch in range(2):
            host[2 * s + ch] = bench.gen_pcm(n, first_id + s, ch)
    with torch.cuda.stream(hip_stream):
        d_in = torch.from_numpy(host).cuda()
        d_out = torch.empty_like(d_in)
    return eng, d_in, d_out


def run(engs, steps=12, warmup=3, stagger=False):
    def one(i):
        e, a, b = engs[i]
        e.process_device(a.data_ptr(), b.data_ptr(), n)
    for _ in range(warmup):
        for i in range(len(engs)):
            one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if stagger and len(engs) == 2:
        # half a call out of phase: engine 1 starts its first call with the convolver only while engine 0 is in its EQ
        engs[0][0].conv_process_device(engs[0][1].data_ptr(), engs[0][2].data_ptr(), n)
    for _ in range(steps):
        for i in range(len(engs)):
            one(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return sum(e.n_streams for e, _, _ in engs) * n * steps / dt / 1e6


s_a, s_b = torch.cuda.Stream(), torch.cuda.Stream()
whole = make(S, 0, s_a)
print(f"one engine, {S} streams, P={P}, T={T}: {run([whole]):.0f} M stereo-samples/s")
whole[0].close()
del whole
torch.cuda.empty_cache()
halves = [make(S // 2, 0, s_a), make(S // 2, S // 2, s_b)]
print(f"two engines of {S // 2} streams on two HIP streams: {run(halves):.0f} M stereo-samples/s")
print(f"same, second engine half a call out of phase: {run(halves, stagger=True):.0f} M stereo-samples/s")
one_half = run([halves[0]])
print(f"one of the halves alone ({S // 2} streams): {one_half:.0f} M stereo-samples/s")
