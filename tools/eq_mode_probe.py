#!/usr/bin/env python3
"""Which EQ kernel for short calls?  Times cpq_eq_process_device for calls of 1 ... 23 blocks in the automatic
(time-parallel) and the sequential (lane-skewed) mode; 256 streams, bench EQ preset."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import convopeq_amd as amd

S, B = 256, 512
for T in (1, 2, 3, 4, 6, 7, 8, 12, 15, 16, 23):
    n = T * B
    x = torch.from_numpy(np.tile(bench.gen_pcm(n, 0, 0), (2 * S, 1))).cuda()
    y = torch.empty_like(x)
    row = []
    for mode in (amd.CPQ_EQ_MODE_AUTO, amd.CPQ_EQ_MODE_SEQUENTIAL):
        eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
        eng.set_eq_params(amd.CPQ_ALL_STREAMS, bench.bench_eq_params(amd, 0.2))
        eng.set_eq_mode(mode)
        for _ in range(20):
            eng.eq_process_device(x.data_ptr(), y.data_ptr(), n)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 200
        for _ in range(reps):
            eng.eq_process_device(x.data_ptr(), y.data_ptr(), n)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / reps * 1e3)
        eng.close()
    print(f"T = {T:2d}: time-parallel {row[0]:.4f} ms, sequential {row[1]:.4f} ms per call")
