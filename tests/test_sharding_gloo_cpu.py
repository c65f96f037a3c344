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


# ---------------------------------------------------------------- bench.py's own launcher, rank layout and reduction
def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_launcher_command_and_defaults():
    b = _bench()
    seen = {}

    def fake_run(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    assert b.launch_ranks(4, ["--gpus", "4", "--steps", "3"], run=fake_run) == 7           # the children's exit code comes back
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # configs[1] on one GPU, the per-GPU share of configs[4] on several
    assert b.parse_args([]).streams == 256 and b.parse_args(["--gpus", "8"]).streams == 1024
    assert b.parse_args(["--gpus", "8", "--streams", "64"]).streams == 64


def test_bench_rank_layout_refuses_a_mismatched_launch():
    b = _bench()
    a = b.parse_args(["--gpus", "2"])
    assert b.rank_layout(a, {"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2"}, 8) == (1, 1, 2, 1)
    with pytest.raises(SystemExit):                     # WORLD_SIZE != --gpus: an error, not a warning
        b.rank_layout(a, {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "4"}, 8)
    with pytest.raises(SystemExit):                     # more ranks than GPUs over RCCL: no silent device sharing
        b.rank_layout(a, {"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2"}, 1)
    g = b.parse_args(["--gpus", "2", "--dist-backend", "gloo"])
    assert b.rank_layout(g, {"RANK": "1", "LOCAL_RANK": "1", "WORLD_SIZE": "2"}, 1) == (1, 1, 2, 0)     # rehearsal


def test_bench_self_launches_two_ranks_and_reduces(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: two child ranks, one JSON line, whole-job counters."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--stub-step",
                        "--steps", "4", "--warmup", "1", "--streams", "8", "--blocks-per-call", "64"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                              # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1
    assert d["config"]["first_stream_of_rank"] == [0, 8]            # rank-major global stream ids, disjoint shards
    total = 2 * 8 * 64 * 512 * 4                                    # SUM over ranks of streams x samples x steps
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 * 4 - total) / total < 1e-3      # value = total / MAX elapsed
    assert len(d["per_rank"]["samples_per_s_mega"]) == 2
    # rank 1 sleeps twice as long per step: the job's time is the slower rank's
    assert d["ms_per_step"] >= 2.0 and d["per_rank"]["samples_per_s_mega"][0] > d["per_rank"]["samples_per_s_mega"][1]
    # a launch whose world size disagrees with --gpus fails instead of running one rank
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub-step"], env=env2,
                        capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (r2.stderr + r2.stdout)


def test_bench_under_the_drivers_launcher_at_eight_ranks():
    """The launch the driver uses for the scaling run -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 ... bench.py --gpus 8` -- with --stub-step (gloo, no GPU): eight ranks, one JSON line from rank
    0, rank-major disjoint stream shards, whole-job counters over the MAX elapsed time."""
    import json
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dist-backend", "gloo", "--stub-step",
                        "--steps", "3", "--warmup", "1", "--streams", "4", "--blocks-per-call", "16"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stderr[-3000:], r.stdout[-1000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["first_stream_of_rank"] == [0, 4, 8, 12, 16, 20, 24, 28]
    total = 8 * 4 * 16 * 512 * 3
    assert abs(d["value"] * 1e6 * d["ms_per_step"] * 1e-3 * 3 - total) / total < 1e-3
    assert len(d["per_rank"]["samples_per_s_mega"]) == 8
    assert d["ms_per_step"] >= 8.0                       # rank 7 sleeps 8 ms per step: the job's time is the slowest rank's
