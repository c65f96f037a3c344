"""world_size-2 gloo test (CPU) of the N>1 path: streams shard across ranks with no data-path collective; the
end-of-run counter reduction is the only collective.  The per-rank 'engine' here is the CPU oracle acting as a
stand-in for the GPU engine so that the sharding arithmetic and the reduction can be checked without GPUs."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, total_streams, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from convopeq_amd.sharding import reduce_counters, streams_of_rank
    import oracle_lib as O
    mine = streams_of_rank(total_streams, world, rank)
    B, nb, L = 512, 6, 3000
    samples, err2, emax = 0, 0.0, 0.0
    for s in mine:
        for ch in range(2):
            h = O.gen_ir(L, stream=s, channel=ch)
            x = O.gen_pcm(B * nb, stream=s, channel=ch)
            c = O.Nuc()
            c.set_impulse(h, B)
            y = c.run(x, B)
            ref = np.convolve(x, h)[:len(x)]
            err2 += float(np.sum((y - ref) ** 2))
            emax = max(emax, float(np.abs(y - ref).max()))
        samples += B * nb
    tot, tmax, e2, em = reduce_counters(samples, 1.0 + rank, err2, emax)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([tot, tmax, e2, em, len(mine)]))
    dist.destroy_process_group()


def test_streams_shard_and_counters_reduce(tmp_path):
    world, total = 2, 5
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert r0[4] + r1[4] == total and {r0[4], r1[4]} == {2.0, 3.0}        # every stream owned exactly once
    assert r0[0] == r1[0] == total * 512 * 6                               # SUM of samples
    assert r0[1] == r1[1] == 2.0                                           # MAX of elapsed
    assert r0[2] == r1[2] and r0[3] == r1[3] and r0[3] < 1e-14


def test_sharding_helpers():
    sys.path.insert(0, ROOT)
    from convopeq_amd.sharding import streams_of_rank, weak_scaling_streams
    owned = sorted(s for r in range(8) for s in streams_of_rank(8192, 8, r))
    assert owned == list(range(8192)) and len(streams_of_rank(8192, 8, 3)) == 1024
    assert weak_scaling_streams(256, 4, 2) == list(range(512, 768))
