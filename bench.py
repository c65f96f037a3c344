#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native ConvoPeq hot path.

Metric (BASELINE.json): Mega stereo-samples/s convolved (131072-tap IR, blk=512), fp64.
One *step* = one pass of the hot path over one batch: every one of S stereo streams pushes T blocks of 512
samples through the convolver (131072-tap private IR per channel, reference semantics) and, unless --no-eq,
the 20-band SVF EQ -- one cpq_engine_process_block_device() call with inputs already resident in HBM.

N = 1 runs BASELINE.json configs[1] (256 stereo streams).  N > 1 runs configs[4]: 8192 / 8 = 1024 streams per GPU,
one rank per GPU, streams sharded across ranks (convopeq_amd/sharding.py), no data-path collective; the only
collective is the end-of-run reduction of the counters (RCCL).  `python bench.py --gpus N` with no WORLD_SIZE in the
environment starts the N ranks itself (torch.distributed.run, from a parent that never touches the GPU); under a
launcher (WORLD_SIZE set) the process is one of the ranks and WORLD_SIZE must equal --gpus.

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

B = 512
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6  # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz (fp64 vector FMA)
STREAMS_CONFIG2 = 256           # BASELINE.json configs[1]
STREAMS_CONFIG5_SHARE = 1024    # BASELINE.json configs[4]: 8192 streams over 8 GPUs


# ----------------------------------------------------------------------------- synthetic data
def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def rand_pm1(seed, stream, channel, idx):
    with np.errstate(over="ignore"):
        key = np.uint64(seed) ^ (np.uint64(stream) << np.uint64(40)) ^ (np.uint64(channel) << np.uint64(32))
        u = splitmix64(key ^ idx.astype(np.uint64))
    return (u >> np.uint64(11)).astype(np.float64) * (2.0 / 9007199254740992.0) - 1.0


def gen_pcm(n, stream, channel, start=0, seed=0xC0FFEE):
    return 0.25 * rand_pm1(seed, stream, channel, np.arange(start, start + n, dtype=np.uint64))


def gen_ir(length, stream, channel, seed=0x1257):
    i = np.arange(length, dtype=np.uint64)
    return 0.05 * rand_pm1(seed, stream, channel, i) * np.exp(-6.9 * i.astype(np.float64) / length)


BENCH_FREQS = [25.0, 40.0, 63.0, 100.0, 160.0, 250.0, 400.0, 630.0, 1000.0, 1600.0, 2500.0, 4000.0, 6300.0,
               10000.0, 11000.0, 12500.0, 14000.0, 16500.0, 18000.0, 19500.0]
BENCH_GAINS = [3, -2, 4, -3, 2, -4, 3, -2, 1.5, -1.5, 2, -2, 3, -3, 1, -1, 2, -2, 1, -1]


def load_autoeq_preset():
    """tests/golden/autoeq_he400se.json: the values of the reference's sample AutoEq preset."""
    with open(os.path.join(ROOT, "tests", "golden", "autoeq_he400se.json")) as f:
        return json.load(f)


def fill_eq_params(p, preset, saturation, attr):
    """Fills an EqParams-like ctypes struct (product or oracle flavour; `attr` maps the field names that differ)."""
    if preset == "autoeq":
        d = load_autoeq_preset()
        for i in range(20):
            b = p.bands[i]
            if i < len(d["filters"]):
                f = d["filters"][i]
                b.frequency, b.gain, b.q = f["fc"], f["gain"], f["q"]
                b.type = {"LSC": 0, "PK": 1, "HSC": 2}[f["type"]]
                b.enabled = 1
            else:                   # loadFromTextFile first disables every band and zeroes its gain (Core.cpp:305-310)
                b.enabled, b.gain = 0, 0.0
            setattr(b, attr["channel_mode"], 0)
        setattr(p, attr["total_gain_db"], d["preamp_db"])
    else:
        for i in range(20):
            b = p.bands[i]
            b.frequency, b.gain, b.q = BENCH_FREQS[i], BENCH_GAINS[i], 1.41
            b.enabled = 1
            setattr(b, attr["channel_mode"], 0)
            b.type = 0 if i == 0 else (2 if i == 19 else 1)
    setattr(p, attr["saturation"], saturation)
    return p


PRODUCT_ATTR = {"channel_mode": "channel_mode", "total_gain_db": "total_gain_db", "saturation": "nonlinear_saturation"}
ORACLE_ATTR = {"channel_mode": "channelMode", "total_gain_db": "totalGainDb", "saturation": "nonlinearSaturation"}


def bench_eq_params(amd, saturation, preset="bench"):
    """SURVEY.md 8(d) EQ bench preset (or the AutoEq one)."""
    return fill_eq_params(amd.eq_params_default(), preset, saturation, PRODUCT_ATTR)


# ----------------------------------------------------------------------------- CPU baseline (oracle, "port")
def _pin_to(cpu):
    try:
        os.sched_setaffinity(0, {cpu})          # pid 0 = the calling thread
        return True
    except (AttributeError, OSError):
        return False


def cpu_share():
    """CPUs this process may actually use at once: CPQ_CPU_THREADS, else the cgroup CPU quota (v2 cpu.max, v1
    cfs_quota / cfs_period) when one is set.  (threads, where the number came from) or (None, ...)."""
    if os.environ.get("CPQ_CPU_THREADS"):
        return int(os.environ["CPQ_CPU_THREADS"]), "CPQ_CPU_THREADS"
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            return max(1, int(int(quota) / int(period) + 0.5)), "cgroup cpu.max"
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0:
            return max(1, int(quota / period + 0.5)), "cgroup cfs_quota"
    except (OSError, ValueError):
        pass
    return None, "no quota: every CPU in the affinity mask"


def cpu_baseline(ir_len, use_eq, saturation, preset, target_seconds=6.0):
    """Times the CPU oracle (restatement of the reference NUC + SVF EQ) on the host cores of this box: one stereo stream
    per thread, each with its own IRs, blocks of 512, every thread pinned to one of the CPUs this process may run on.
    Then the single-thread config-1 leg (1 stream, 4096 taps, EQ bypassed: the reference's plumbing case)."""
    import oracle_lib as O
    O.lib()
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    share, share_src = cpu_share()
    cores = max(1, min(len(cpus), share) if share else len(cpus))
    cpus = cpus[:cores]
    n_blocks = 1024                       # per thread and pass: 524288 samples per channel
    po = fill_eq_params(O.eq_params_default(), preset, saturation, ORACLE_ATTR)
    done = [0] * cores
    pinned = [False] * cores
    stop_at = [0.0]

    def work(tid):
        pinned[tid] = _pin_to(cpus[tid])
        irs = [O.gen_ir(ir_len, stream=tid, channel=ch) for ch in range(2)]
        nucs = [O.Nuc(), O.Nuc()]
        for ch in range(2):
            nucs[ch].set_impulse(irs[ch], B)
        x = [O.gen_pcm(n_blocks * B, stream=tid, channel=ch) for ch in range(2)]
        state = np.zeros(168)
        barrier.wait()
        while True:
            y = [nucs[ch].run(x[ch], B) for ch in range(2)]
            if use_eq:
                O.eq_process_stereo(y[0], y[1], po, state=state)
            done[tid] += n_blocks * B
            if time.perf_counter() >= stop_at[0]:
                break

    barrier = threading.Barrier(cores + 1)
    threads = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    for t in threads:
        t.start()
    stop_at[0] = time.perf_counter() + 3600.0
    barrier.wait()
    t0 = time.perf_counter()
    stop_at[0] = t0 + target_seconds
    for t in threads:
        t.join()
    dt = time.perf_counter() - t0
    total = sum(done)

    # config 1, single thread: SetImpulse(ir, 4096, 512) -> Add/Get per block
    h = O.gen_ir(4096, stream=0, channel=0)
    hr = O.gen_ir(4096, stream=0, channel=1)
    c1 = [O.Nuc(), O.Nuc()]
    c1[0].set_impulse(h, B)
    c1[1].set_impulse(hr, B)
    x1 = [O.gen_pcm(4096 * B, stream=0, channel=ch) for ch in range(2)]
    t1 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t1 < 1.5:
        for ch in range(2):
            c1[ch].run(x1[ch], B)
        n1 += 4096 * B
    dt1 = time.perf_counter() - t1
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(total / dt / 1e6, 3), "unit": "Mega stereo-samples/s", "cores": cores, "kind": "port",
            "host_cpu_count": os.cpu_count(), "cpu_share": share, "cpu_share_source": share_src, "cpu_model": model,
            "threads_pinned": all(pinned),
            "sample": f"{cores} stereo streams (one per pinned thread; the box reports os.cpu_count() = {os.cpu_count()}, "
                      f"{cores} threads used: {share_src}), {ir_len}-tap IR each, blk {B}, "
                      f"{total // cores} samples per stream, conv{'+EQ' if use_eq else ''}, {dt:.1f} s wall; "
                      "oracle = C restatement of the reference NUC schedule (own radix-2 FFT, not IPP)",
            "config1_single_thread": {"value": round(n1 / dt1 / 1e6, 3), "unit": "Mega stereo-samples/s", "cores": 1,
                                      "sample": f"1 stereo stream, 4096-tap IR, blk {B}, EQ bypassed, {n1} samples, {dt1:.1f} s"}}


def parity_of_timed_engine(amd, eng, d_in, d_out, torch, S, n, ir_len, use_eq, saturation, preset, stream_ids, exact,
                           shared_ir):
    """Parity of the TIMED engine itself: its state is reset, two calls of the step's input run from reset, and the output
    rows of a few streams (first, second, middle, last) are read back and compared with the oracle fed the same IR / PCM.
    The oracle is the checker here; nothing of it is timed or shipped."""
    import oracle_lib as O
    O.lib()
    eng.conv_reset()
    eng.eq_reset()
    outs = []
    for _ in range(2):
        if use_eq:
            eng.process_device(d_in.data_ptr(), d_out.data_ptr(), n)
        else:
            eng.conv_process_device(d_in.data_ptr(), d_out.data_ptr(), n)
        torch.cuda.synchronize()
        outs.append(d_out.clone())
    picks = sorted({0, min(1, S - 1), S // 2, S - 1})
    po = fill_eq_params(O.eq_params_default(), preset, saturation, ORACLE_ATTR)
    err2, cnt, emax, sig2 = 0.0, 0, 0.0, 0.0
    for s in picks:
        gid = stream_ids[s]
        ref = []
        for ch in range(2):
            h = gen_ir(ir_len, 0 if shared_ir else gid, ch)
            x = gen_pcm(n, gid, ch)
            xx = np.concatenate([x, x])
            if exact:
                import scipy.signal
                ref.append(scipy.signal.fftconvolve(xx, h)[:2 * n])
            else:
                nuc = O.Nuc()
                nuc.set_impulse(h, B)
                ref.append(nuc.run(xx, B))
        if use_eq:
            ref[0], ref[1], _ = O.eq_process_stereo(ref[0], ref[1], po)
        for ch in range(2):
            y = np.concatenate([o[2 * s + ch].cpu().numpy() for o in outs])
            d = y - ref[ch]
            err2 += float(np.dot(d, d))
            sig2 += float(np.dot(ref[ch], ref[ch]))
            emax = max(emax, float(np.abs(d).max()))
            cnt += d.size
    return {"rms_err": float(np.sqrt(err2 / cnt)), "max_abs_err": emax, "signal_rms": float(np.sqrt(sig2 / cnt)),
            "target_rms_err": 1e-12, "err_sq_sum": err2, "count": cnt,
            "sample": f"the timed engine ({S} streams), reset, two calls of {n} samples; streams {picks} of this rank "
                      f"(global ids {[stream_ids[s] for s in picks]}), both channels, conv{'+EQ' if use_eq else ''}, "
                      f"HIP output vs the oracle ({'scipy fftconvolve' if exact else 'stateful NUC emulation'}) on the same IR + PCM"}


# ----------------------------------------------------------------------------- PMC traffic (committed profiles)
def kernel_source_hash():
    """Identifies the kernels a PMC summary was measured on: sha256 over the .hip sources."""
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "convopeq_amd", "csrc", "*.hip"))):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def load_pmc_traffic(path, kernel, running):
    """HBM bytes per launch of `kernel` from a committed rocprofv3 PMC summary, with its provenance.  The number is only
    returned when the summary says which workload and kernel sources it was measured on and both match this run;
    otherwise traffic is null (a stale number is worse than none)."""
    src = {"file": os.path.relpath(path, ROOT) if path else None, "match": False}
    if not path or not os.path.exists(path):
        src["reason"] = "no PMC summary found"
        return None, src
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception as ex:       # noqa: BLE001
        src["reason"] = f"unreadable: {ex}"
        return None, src
    tag = d.get("_config")
    if not tag:
        src["reason"] = "summary carries no _config tag (pre-round-2 file)"
        return None, src
    src["measured_on"] = tag
    mism = [k for k in ("streams", "ir_len", "block", "blocks_per_call", "partition", "schedule", "eq", "kernel_sources")
            if tag.get(k) != running.get(k)]
    if mism:
        src["reason"] = "differs from this run in: " + ", ".join(mism)
        return None, src
    # the profiler id k_fdl_mac covers the workgroup-cooperative variant of long calls (k_fdl_mac_wg in rocprof names)
    # and the register-tile variants of short ones (k_fdl_mac<TT, PF>, summarised as "k_fdl_mac")
    names = {"k_fdl_mac": ["k_fdl_mac_wg", "k_fdl_mac"], "k_fdl_mac:tile": ["k_fdl_mac"],
             "k_svf_cascade_tp": ["k_svf_cascade_tpv", "k_svf_cascade_tp"]}.get(kernel, [kernel])
    for nme in names:
        if nme in d and isinstance(d[nme], dict):
            src["match"] = True
            src["kernel"] = nme
            return d[nme].get("hbm_bytes_per_launch"), src
    src["reason"] = f"kernel {kernel} not in the summary"
    return None, src


def newest_pmc_summary():
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    return c[-1] if c else None


def load_sq_issue_util(path, kernel, running, waves_per_simd=4):
    """VALU issue-slot utilisation of `kernel` from a committed SQ-counter summary (tools/summarize_sq.py):
    SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = the share of a wave's cycles in which it issues a VALU instruction; times the waves
    a SIMD holds = the share of the SIMD's VALU issue slots in use.  Same provenance rule as load_pmc_traffic."""
    src = {"file": os.path.relpath(path, ROOT) if path else None, "match": False}
    if not path or not os.path.exists(path):
        src["reason"] = "no SQ-counter summary found"
        return None, src
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception as ex:       # noqa: BLE001
        src["reason"] = f"unreadable: {ex}"
        return None, src
    tag = d.get("_config")
    if not tag:
        src["reason"] = "summary carries no _config tag"
        return None, src
    src["measured_on"] = tag
    mism = [k for k in ("streams", "ir_len", "block", "blocks_per_call", "partition", "schedule", "eq", "kernel_sources")
            if tag.get(k) != running.get(k)]
    if mism:
        src["reason"] = "differs from this run in: " + ", ".join(mism)
        return None, src
    for nme in {"k_svf_cascade_tp": ["k_svf_cascade_tpv", "k_svf_cascade_tp"], "k_fdl_mac": ["k_fdl_mac_wg", "k_fdl_mac"]}.get(kernel, [kernel]):
        c = d.get(nme)
        if isinstance(c, dict) and "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c:
            src.update({"match": True, "kernel": nme, "waves_per_simd": waves_per_simd})
            return round(c["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / c["SQ_WAVE_CYCLES"]["mean_per_launch"] * waves_per_simd, 4), src
    src["reason"] = f"kernel {kernel} not in the summary"
    return None, src


def newest_sq_summary():
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters.json")))
    return c[-1] if c else None


# ----------------------------------------------------------------------------- algorithmic bytes / flops
def launch_plan(n, layers):
    """layers: [(partition, partitions of the IR)].  A layer's kernels launch when a partition of it has filled: a call of n
    samples carries n / P partitions -- a whole number of them per launch, one launch per call, when n >= P; one partition
    per launch every P / n calls otherwise (the streaming regimes: a 4096-sample tail layer under 512-sample calls runs
    every eighth call).  -> [(P, K, partitions per launch, launches per call)]"""
    out = []
    for pl, kl in layers:
        nb = max(1, n // pl)
        out.append((pl, kl, nb, (n / pl) / nb))
    return out


def mac_launch_bytes(n_ch, ir_rows_mult, pl, kl, nb):
    """k_fdl_mac, one launch producing nb output rows per channel: every FDL row, IR row and output row once."""
    return (n_ch * (kl + nb - 1) + ir_rows_mult * kl + n_ch * nb) * pl * 16


def algorithmic_bytes_per_step(n_ch, n, ir_rows_mult, fft_layers, mac_layers, layered_tail=None, native_tails=0, native_fused=False):
    """HBM bytes one call (step) of n samples needs per kernel family if every row / sample is moved once.
    fft_layers / mac_layers: launch_plan() lists (they differ in layered mode: one transform grid, one MAC per layer).
    layered_tail: (n_tail, [min(n, output_delay + 2 P) per tail layer]) in layered mode; native_tails: tail layers of a plan
    group (their delay-line read-add passes over the call's output); native_fused: whole 512-sample-block calls of a plan
    group -- layer 0's forward transform also writes the input into the accumulators of the layers whose partition has not
    filled (8 B per sample each), its inverse transform also reads the tail layers' delay lines (8 B per sample each), and no
    pass over the output is left."""
    b = {
        "k_rfft_fwd_ols": sum(lps * n_ch * nb * (pl * 8 + pl * 16) for pl, _, nb, lps in fft_layers),
        "k_fdl_mac": sum(lps * mac_launch_bytes(n_ch, ir_rows_mult, pl, kl, nb) for pl, kl, nb, lps in mac_layers),
        "k_fdl_mac_dcnyq": sum(lps * (n_ch * (kl + nb - 1 + nb) * 16 + ir_rows_mult * kl * 16) for pl, kl, nb, lps in mac_layers),
        "k_rfft_inv_ols": sum(lps * n_ch * nb * (pl * 16 + pl * 8) for pl, _, nb, lps in mac_layers),
        "k_svf_cascade_tp": n_ch * n * 16,
        "k_svf_cascade": n_ch * n * 16,
        "k_convproc_mix": 0,
    }
    if layered_tail:
        n_tail, spans = layered_tail
        # the layer-0 inverse transform also reads what the replayed delay-line reader adds (8 B per sample and tail layer);
        # only what later calls may still read is appended to the rings
        b["k_rfft_inv_ols"] += n_ch * n * 8 * n_tail
        b["k_convproc_mix"] = n_ch * 16 * sum(spans)
    elif native_tails > 0 and native_fused:
        b["k_rfft_inv_ols"] += n_ch * n * 8 * native_tails
        b["k_rfft_fwd_ols"] += n_ch * n * 8 * sum(1 for pl, _, _, _ in fft_layers[1:] if pl > n)
    elif native_tails > 0:
        # the delay-line read-add of the tail layers over the call's output (read + write once, 8 B per sample and tail layer
        # from the rings; the ring writes are inside the inverse transforms)
        b["k_convproc_mix"] = n_ch * n * (16 + 8 * native_tails)
    return b


def svf_flop_model(saturation):
    """fp64 operations per band-sample in the reference's arithmetic (processBandStereo, EQProcessor.Processing.cpp:228-262):
    the linear recurrence is 10 instructions, 7 of them FMAs = 17 flops.  With saturation > 0 the output stage adds the
    fastTanh blend: 9 arithmetic flops (x^2, numerator 2, denominator 2, one division counted as one, blend 3) + 7 slots of
    min / max / compare / select (argument clamp 2, output guard 2, +-100 clamp 2, select 1).  At saturation 0 the blend is
    skipped (`if sat > 0`): the guard and the clamp remain (4 slots, no arithmetic)."""
    if saturation > 0.0:
        return {"slots": 33.0, "arithmetic": 26.0}
    return {"slots": 21.0, "arithmetic": 17.0}


# ----------------------------------------------------------------------------- launcher
def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n_ranks, argv, run=subprocess.call):
    """Parent of `python bench.py --gpus N` (no WORLD_SIZE): starts N ranks of this script as CHILD processes and returns
    their exit code.  The parent has not imported torch, loaded the HIP library or touched the GPU (and never execs)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return run(cmd, env=env)


def rank_layout(args, env, device_count):
    """(rank, local_rank, world, device index) of this process, or SystemExit when the launch does not match --gpus."""
    rank = int(env.get("RANK", "0"))
    local_rank = int(env.get("LOCAL_RANK", "0"))
    world = int(env.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: launch one rank per GPU "
                         f"(or run `python bench.py --gpus {args.gpus}` without a launcher)")
    if local_rank >= device_count:
        if args.dist_backend != "gloo":
            raise SystemExit(f"bench.py: rank {rank} (local {local_rank}) has no GPU of its own: {device_count} visible")
        return rank, local_rank, world, local_rank % max(device_count, 1)      # gloo rehearsal: ranks share the devices
    return rank, local_rank, world, local_rank


def timed_steps(step, sync, dist, steps, warmup, after_warmup=None, extra=None):
    """The timing contract: W untimed warm-up steps, then EXACTLY K steps bracketed by a barrier + device synchronise on
    both sides.  Returns (this rank's own time to its last step, time to the closing barrier)."""
    for _ in range(warmup):
        step()
    sync()
    if after_warmup:
        after_warmup()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    if extra is not None:
        extra["enqueue_s"] = time.perf_counter() - t0      # host time to enqueue the K steps (before the device has finished them)
    sync()
    mine = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    return mine, time.perf_counter() - t0


def run_stub(args):
    """--stub-step (tests/test_sharding_gloo_cpu.py): the rank layout, the timing harness and the counter reduction of
    the real bench with a sleep in place of the engine call -- no GPU, no library; the line says so and is no measurement."""
    import torch  # noqa: F401
    import torch.distributed as dist
    from convopeq_amd import sharding
    rank, _, world, _ = rank_layout(args, os.environ, 0 if args.dist_backend == "gloo" else 1 << 30)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    S, n = args.streams, args.blocks_per_call * args.block
    ids = sharding.weak_scaling_streams(S, world, rank)
    mine, elapsed = timed_steps(lambda: time.sleep(0.001 * (1 + rank)), lambda: None, dist if world > 1 else None,
                                args.steps, args.warmup)
    samples = float(S) * n * args.steps
    rate = samples / mine / 1e6
    samples, elapsed, _, _ = sharding.reduce_counters(samples, elapsed)
    rates = sharding.gather_values(rate)
    firsts = sharding.gather_values(float(ids[0]))
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (no kernels)", "value": round(samples / elapsed / 1e6, 3),
                          "unit": "Mega stereo-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(elapsed / args.steps * 1e3, 4), "data": "stub", "scaling": "weak",
                          "config": {"streams_per_gpu": S, "first_stream_of_rank": [int(f) for f in firsts]},
                          "per_rank": {"samples_per_s_mega": [round(r, 1) for r in rates]}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=0,
                    help=f"stereo streams per GPU (default {STREAMS_CONFIG2} at --gpus 1 = configs[1], "
                         f"{STREAMS_CONFIG5_SHARE} at --gpus N>1 = the per-GPU share of configs[4])")
    ap.add_argument("--ir-len", type=int, default=131072)
    ap.add_argument("--block", type=int, default=512, help="diagnostic: block size")
    ap.add_argument("--partition", type=int, default=-1,
                    help="internal FFT partition size; 0 = the block size; default -1 = CPQ_PARTITION_AUTO, the engine's own "
                         "choice: 4096 when the calls are whole 4096-sample partitions (the throughput path: "
                         "profiles/r02b_sweep_partition_x_blocks_per_call.txt), else 512, else the block")
    ap.add_argument("--blocks-per-call", type=int, default=1024,
                    help="blocks of --block samples per process() call (default 1024 = 524288 samples, the reference's largest "
                         "process() block, ConvolverProcessor.Runtime.cpp:609)")
    ap.add_argument("--mac-tile", type=int, default=0)
    ap.add_argument("--no-eq", action="store_true")
    ap.add_argument("--eq-only", action="store_true", help="diagnostic: time the EQ kernel alone")
    ap.add_argument("--saturation", type=float, default=0.2)
    ap.add_argument("--eq-preset", choices=["bench", "autoeq"], default="bench",
                    help="bench: SURVEY 8(d) preset; autoeq: the reference's sample AutoEq preset (tests/golden fixture)")
    ap.add_argument("--pcm-scale", type=float, default=1.0,
                    help="diagnostic: multiplies the synthetic PCM (e.g. 64 drives the EQ's guarded hot-signal path)")
    ap.add_argument("--shared-ir", action="store_true")
    ap.add_argument("--schedule", choices=["uniform", "nuc"], default="uniform",
                    help="uniform: one partition size for the whole h_eff (headline, HBM-roofline path); nuc: the reference's "
                         "own non-uniform schedule run natively (BASELINE.json configs[3])")
    ap.add_argument("--call-mode", choices=["blocks", "any"], default="blocks",
                    help="blocks: calls of whole power-of-two blocks (CPQ_CALLS_WHOLE_BLOCKS); any: CPQ_CALLS_ANY -- --block is then "
                         "the call quantum (480, 441 ...), the reference's Add / Get are reproduced chunk by chunk (plan groups)")
    ap.add_argument("--exact", action="store_true", help="plain linear convolution instead of reference h_eff")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--host-buffers", action="store_true", help="diagnostic: host-pointer entry point (PCIe-inclusive rate)")
    ap.add_argument("--pinned", action="store_true", help="with --host-buffers: pin the host buffers (cpq_host_register)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (single-GPU rehearsal of N>1)")
    ap.add_argument("--stub-step", action="store_true", help="test-only: launcher / reduction self-test without a GPU")
    ap.add_argument("--pmc-json", default=None, help="PMC summary for roofline.traffic (default: newest profiles/r*_pmc_traffic.json)")
    ap.add_argument("--sq-json", default=None, help="SQ-counter summary for roofline.valu_issue_util (default: newest profiles/r*_sq_counters.json)")
    args = ap.parse_args(argv)
    if args.streams <= 0:
        args.streams = STREAMS_CONFIG2 if args.gpus <= 1 else STREAMS_CONFIG5_SHARE
    if args.partition == args.block:
        args.partition = 0
    return args


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.stub_step:
        return run_stub(args)

    import torch
    import convopeq_amd as amd
    from convopeq_amd import sharding

    rank, local_rank, world, dev_index = rank_layout(args, os.environ, torch.cuda.device_count())
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    red_dev = "cuda" if (dist is not None and args.dist_backend == "nccl") else "cpu"

    global B
    B = args.block
    S, T, L = args.streams, args.blocks_per_call, args.ir_len
    n = T * B
    use_eq = not args.no_eq
    eng = amd.BatchedEngine(S, block_size=B, max_ir_len=L, max_blocks_per_call=T,
                            semantics=amd.CPQ_SEM_EXACT if args.exact else amd.CPQ_SEM_REFERENCE,
                            device=dev_index, mac_tile=args.mac_tile, partition_size=args.partition,
                            schedule=amd.CPQ_SCHED_REFERENCE_NUC if args.schedule == "nuc" else amd.CPQ_SCHED_UNIFORM,
                            call_mode=amd.CPQ_CALLS_ANY if args.call_mode == "any" else amd.CPQ_CALLS_WHOLE_BLOCKS)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)

    # synthetic IRs and PCM; stream ids are global (rank-major, sharding.py) so every rank convolves different streams
    t_setup = time.perf_counter()
    ids = sharding.weak_scaling_streams(S, world, rank)
    if args.shared_ir:
        eng.set_impulse(amd.CPQ_ALL_STREAMS, gen_ir(L, 0, 0), gen_ir(L, 0, 1))
    else:
        for s in range(S):
            eng.set_impulse(s, gen_ir(L, ids[s], 0), gen_ir(L, ids[s], 1))
    if use_eq:
        eng.set_eq_params(amd.CPQ_ALL_STREAMS, bench_eq_params(amd, args.saturation, args.eq_preset))
    host = np.empty((2 * S, n))
    for s in range(S):
        for ch in range(2):
            host[2 * s + ch] = gen_pcm(n, ids[s], ch)
    if args.pcm_scale != 1.0:
        host *= args.pcm_scale
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.empty_like(d_in)
    plan = eng.plan()
    P = eng.partition_size()                             # internal FFT partition size (the engine's choice at --partition -1)
    Tp = n // P                                          # partitions per call
    taps = L if args.exact else plan.heff_len
    k_parts = (taps + P - 1) // P
    setup_s = time.perf_counter() - t_setup

    host_out = np.empty_like(host) if args.host_buffers else None
    if args.host_buffers and args.pinned:
        for a in (host, host_out):
            assert eng._lib.cpq_host_register(a.ctypes.data, a.nbytes) == 0

    def step():
        if args.host_buffers:
            # PCIe-inclusive: pageable host buffers through cpq_engine_process_block (H2D + kernels + D2H + sync)
            rc = eng._lib.cpq_engine_process_block(eng._h, host.ctypes.data_as(amd._capi.c_double_p),
                                                   host_out.ctypes.data_as(amd._capi.c_double_p), n)
            assert rc == 0
        elif args.eq_only:
            eng.eq_process_device(d_in.data_ptr(), d_out.data_ptr(), n)
        elif use_eq:
            eng.process_device(d_in.data_ptr(), d_out.data_ptr(), n)
        else:
            eng.conv_process_device(d_in.data_ptr(), d_out.data_ptr(), n)

    def start_profile():
        eng.profile_enable(True)        # pre-creates the event pool: no hipEventCreate inside the timed region
        eng.profile_reset()

    # Per-kernel times come from HIP events the library records around every kernel scope on the engine's stream.  Around
    # long calls they are free (8 scopes per 10 ms); around 512-sample calls every event costs the stream ~10 us of
    # serialisation -- more than the kernels between them (profiles/r04c_streaming_timeline.txt) -- so for calls below 65536
    # samples the timed region runs WITHOUT them and the same K steps are repeated with them for the per-kernel numbers.
    host_side = {}
    separate_profile = n < 65536
    my_elapsed, elapsed = timed_steps(step, torch.cuda.synchronize, dist, args.steps, args.warmup,
                                      None if separate_profile else start_profile, host_side)
    profiled_elapsed = None
    if separate_profile:
        start_profile()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        profiled_elapsed = time.perf_counter() - t0
    prof = eng.profile_read()

    # The same MAC path with ONE partition per call (the reference's own call pattern): every FDL and IR row is
    # streamed for a single output row, so the kernel is HBM-bound there, whereas at T blocks per call it is past the
    # fp64 ridge.  Measured after the timed region (not part of `value`) to give the roofline object both regimes.
    prof1 = None
    if not (args.eq_only or args.host_buffers or args.schedule == "nuc" or args.call_mode == "any") and rank == 0:
        P1 = P
        eng.profile_reset()
        for _ in range(40):
            eng.conv_process_device(d_in.data_ptr(), d_out.data_ptr(), P1)
        torch.cuda.synchronize()
        prof1 = eng.profile_read()
    eng.profile_enable(False)

    # parity of the timed engine (every rank checks its own shard; squared errors are summed over ranks)
    parity = None
    if not (args.no_parity or args.eq_only or args.host_buffers or args.pcm_scale != 1.0) and B == 512:
        parity = parity_of_timed_engine(amd, eng, d_in, d_out, torch, S, n, L, use_eq, args.saturation, args.eq_preset,
                                        ids, args.exact, args.shared_ir)

    samples = float(S) * n * args.steps          # stereo samples this rank processed
    my_rate = samples / my_elapsed / 1e6
    samples, elapsed, err2, emax = sharding.reduce_counters(samples, elapsed, parity["err_sq_sum"] if parity else 0.0,
                                                            parity["max_abs_err"] if parity else 0.0, device=red_dev)
    rates = sharding.gather_values(my_rate, device=red_dev)
    if parity and world > 1:
        cnt_all = parity["count"] * world
        parity["rms_err"] = float(np.sqrt(err2 / cnt_all))
        parity["max_abs_err"] = emax
        parity["sample"] += f"; squared errors summed over {world} ranks"

    if rank == 0:
        n_ch = 2 * S
        # algorithmic HBM bytes per launch: every row / sample a kernel needs is moved once
        spec_bytes = P * 16
        ir_rows = (2 if args.shared_ir else n_ch) * k_parts
        # layers the convolver actually runs: (partition size, partitions of the IR) -> launch_plan()
        if args.schedule == "nuc" or args.call_mode == "any":
            layer_list = [(plan.part_size[l], plan.num_parts_ir[l]) for l in range(plan.num_layers)]
        else:
            layer_list = [(P, k_parts)]
        layers = launch_plan(n, layer_list)
        ir_mult = 2 if args.shared_ir else n_ch
        # time-varying plans under reference semantics (a tail partition longer than the IR before it: blocks >= 1024 at the
        # defaults) run in layered mode on the uniform grid: one forward FFT, then per layer a MAC over that layer's partitions
        # and an inverse FFT, and the replayed delay-line reader
        n_tail = plan.num_layers - 1
        layered = (args.schedule == "uniform" and not args.exact and args.call_mode == "blocks" and
                   any(plan.part_size[l] > plan.output_delay[l] for l in range(1, plan.num_layers)))
        mac_layers = launch_plan(n, [(P, (plan.len[l] + P - 1) // P) for l in range(plan.num_layers)]) if layered else layers
        alg_bytes = algorithmic_bytes_per_step(
            n_ch, n, ir_mult, layers, mac_layers,
            layered_tail=(n_tail, [min(n, plan.output_delay[l] + 2 * plan.part_size[l]) for l in range(1, plan.num_layers)]) if layered else None,
            native_tails=0 if layered else max(0, len(layers) - 1),
            native_fused=(args.schedule == "nuc" and args.call_mode == "blocks" and B == 512 and not layered))
        # fp64 operations per step (FMA = 2): the cooperative MAC kernel (>= 48 rows per call) spends 3 real FMAs per complex
        # MAC (Gauss), the tile kernels 4; SVF: svf_flop_model()
        mac_flop = lambda nb: 6.0 if nb >= 48 else 8.0
        n_bands = 20 if args.eq_preset == "bench" else len(load_autoeq_preset()["filters"])
        svf_model = svf_flop_model(args.saturation)
        alg_flops = {"k_fdl_mac": sum(lps * mac_flop(nb) * n_ch * nb * kl * pl for pl, kl, nb, lps in mac_layers),
                     "k_svf_cascade_tp": svf_model["slots"] * n_bands * n_ch * n}
        per_kernel = {}
        for name, (cnt, ms) in prof.items():
            if cnt == 0:
                continue
            avg_s = ms / cnt * 1e-3
            step_s = ms / args.steps * 1e-3
            per_launch = int(alg_bytes[name] * args.steps / cnt)
            per_kernel[name] = {"launches": cnt, "avg_launch_ms": round(avg_s * 1e3, 4),
                                "algorithmic_bytes_per_launch": per_launch,
                                "achieved_gbs": round(alg_bytes[name] / step_s / 1e9, 1)}
            if alg_bytes[name] == 0:      # small bookkeeping launches only (no pass over the samples): no bandwidth figure
                per_kernel[name]["achieved_gbs"] = None
            if name in alg_flops:
                per_kernel[name]["fp64_tflops"] = round(alg_flops[name] / step_s / 1e12, 2)
            if name == "k_svf_cascade_tp":
                per_kernel[name]["fp64_tflops_arithmetic"] = round(alg_flops[name] * svf_model["arithmetic"] / svf_model["slots"] / step_s / 1e12, 2)
        tot = {k: v["avg_launch_ms"] * v["launches"] for k, v in per_kernel.items()}
        dominant = max(tot, key=tot.get)
        co_dominant = sorted(k for k in tot if tot[k] >= 0.9 * tot[dominant])
        # the FDL MAC and the SVF cascade can tie within a few percent; when they do, the roofline object describes the
        # HBM-streaming one (the kernel north_star defines the roofline on), both are listed
        if "k_fdl_mac" in co_dominant:
            dominant = "k_fdl_mac"
        dk = per_kernel[dominant]
        running = {"streams": S, "ir_len": L, "block": B, "blocks_per_call": T, "partition": P,
                   "schedule": args.schedule, "eq": use_eq, "kernel_sources": kernel_source_hash()}
        pmc_path = args.pmc_json or newest_pmc_summary()
        hbm_regime = None
        if prof1 and prof1.get("k_fdl_mac", (0, 0.0))[0] > 0:
            cnt1, ms1 = prof1["k_fdl_mac"]
            lp1 = launch_plan(P, [(P, (plan.len[l] + P - 1) // P) for l in range(plan.num_layers)] if layered else [(P, k_parts)])
            # (40 calls of one partition each: bytes of one call / MAC launches per call)
            b1 = int(algorithmic_bytes_per_step(n_ch, P, ir_mult, lp1, lp1)["k_fdl_mac"] * 40 / cnt1)
            gbs1 = b1 / (ms1 / cnt1 * 1e-3) / 1e9
            tr1, src1 = load_pmc_traffic(pmc_path, "k_fdl_mac:tile", running)
            hbm_regime = {"kernel": "k_fdl_mac", "blocks_per_call": P // B, "algorithmic_bytes_per_launch": b1,
                          "avg_launch_ms": round(ms1 / cnt1, 4), "achieved": round(gbs1, 1),
                          "frac": round(gbs1 / HBM_PEAK_GBS, 4), "launches": cnt1, "traffic": tr1}
        flop_per_byte = alg_flops["k_fdl_mac"] / alg_bytes["k_fdl_mac"] if alg_bytes["k_fdl_mac"] else 0.0
        ridge = FP64_VECTOR_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS
        traffic, traffic_src = load_pmc_traffic(pmc_path, dominant, running)
        if dominant.startswith("k_svf"):
            # 16 B of HBM per sample against ~35 fp64 instructions per band-sample: the fp64 vector issue rate bounds it
            issue_util, issue_src = load_sq_issue_util(args.sq_json or newest_sq_summary(), dominant, running)
            roof = {"kernel": dominant, "bound": "fp64_vector", "achieved": dk.get("fp64_tflops"),
                    "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round((dk.get("fp64_tflops") or 0.0) / FP64_VECTOR_PEAK_TFLOPS, 4),
                    # the same with the arithmetic operations only (no min / max / compare / select slots)
                    "flops_per_band_sample": svf_model,
                    "achieved_arithmetic": dk.get("fp64_tflops_arithmetic"),
                    "frac_arithmetic": round((dk.get("fp64_tflops_arithmetic") or 0.0) / FP64_VECTOR_PEAK_TFLOPS, 4),
                    # what the SQ counters say about the same kernel: share of the SIMDs' VALU issue slots in use
                    "valu_issue_util": issue_util, "valu_issue_util_source": issue_src,
                    "note": "k_svf_cascade_tp is fp64-issue bound, not HBM bound: 20 sequential nonlinear bands per sample, "
                            f"{svf_model['slots']:.0f} operation slots ({svf_model['arithmetic']:.0f} of them arithmetic) per band-sample in the "
                            "reference's arithmetic against 16 B of HBM traffic per sample; fp64 "
                            "MFMA and VALU share one datapath on gfx950 (profiles/r02a_ubench_fp64_valu_mfma_coexec.txt)",
                    "hbm_kernel": {"kernel": "k_fdl_mac", "achieved": per_kernel["k_fdl_mac"]["achieved_gbs"] if "k_fdl_mac" in per_kernel else None,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(per_kernel["k_fdl_mac"]["achieved_gbs"] / HBM_PEAK_GBS, 4) if "k_fdl_mac" in per_kernel else None,
                                   "note": "60 % reads / 40 % writes; MI355X_MICROARCH.md measures 6.0-6.2 TB/s for swept streams "
                                           "(the attainable part of the 8 TB/s peak that `frac` is quoted against)"}}
        else:
            roof = {"kernel": dominant, "bound": "hbm" if flop_per_byte < ridge else "fp64_vector",
                    "achieved": dk["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(dk["achieved_gbs"] / HBM_PEAK_GBS, 4),
                    "note": "HBM stream of FDL and IR spectra; `bound` from the kernel's flop/byte against the fp64 ridge"}
        roof.update({
            "traffic": traffic, "traffic_source": traffic_src,
            "frac_of_pmc_traffic": (round(traffic / (dk["avg_launch_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None),
            "algorithmic_bytes_per_launch": dk["algorithmic_bytes_per_launch"],
            "avg_launch_ms": dk["avg_launch_ms"], "launches": dk["launches"],
            "co_dominant_kernels": co_dominant,
            "fp64_vector": {"achieved_tflops": dk.get("fp64_tflops"), "peak_tflops": FP64_VECTOR_PEAK_TFLOPS},
            # k_fdl_mac at T partitions per call executes 6 K T flop (3 FMAs per complex MAC) per 16 (2K + T) bytes
            "mac_flop_per_byte": round(flop_per_byte, 2), "ridge_flop_per_byte": round(ridge, 2),
            "hbm_regime": hbm_regime,
        })
        # a fraction above 1 is an accounting error (more algorithmic bytes or flops charged than the kernel can have moved)
        fracs = [roof.get("frac"), (roof.get("hbm_kernel") or {}).get("frac"), (hbm_regime or {}).get("frac"), roof.get("frac_arithmetic")]
        roof["accounting_ok"] = all(f is None or f <= 1.0 for f in fracs)
        config_name = ("configs[1]" if (world == 1 and S == STREAMS_CONFIG2 and L == 131072 and B == 512) else
                       "configs[4] per-GPU share (8192 streams / 8 GPUs)" if (S == STREAMS_CONFIG5_SHARE and L == 131072 and B == 512) else
                       "modified")
        out = {
            "metric": "Mega stereo-samples/s convolved (131072-tap IR, blk=512)",
            "value": round(samples / elapsed / 1e6, 3),
            "unit": "Mega stereo-samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{S} stereo streams per GPU x {world} GPU(s) = {S * world} streams, {L}-tap IR "
                            f"({'one shared stereo IR' if args.shared_ir else 'private IR per channel'}), blk {B}, "
                            f"fp64 overlap-save conv"
                            f"{' + 20-band SVF EQ (%s preset, sat %.1f)' % (args.eq_preset, args.saturation) if use_eq else ''}"
                            f" [BASELINE.json {config_name}]",
                "streams_per_gpu": S, "ir_taps": L, "block": B, "blocks_per_call": T,
                "schedule": (f"uniform overlap-save, FFT partition P={P}, K={k_parts} partitions of "
                             f"{'h' if args.exact else 'h_eff (reference NUC semantics at blk 512)'}, "
                             f"{T} blocks ({n} samples) per call = {Tp} partitions per FDL pass") if (args.schedule == "uniform" and args.call_mode == "blocks")
                            else ("non-uniform (the reference's own layer plan run natively): " +
                                  " + ".join(f"{kl} x {pl}" for pl, kl, _, _ in layers) + f" partitions, {T} blocks per call"),
                "partition": P,
                "eq": use_eq, "eq_preset": args.eq_preset if use_eq else None, "saturation": args.saturation if use_eq else None,
                "pcm_scale": args.pcm_scale,
                "parallelism": f"streams sharded rank-major, {world} rank(s), one per GPU"
                               + (" (gloo rehearsal: ranks share the visible devices)" if args.dist_backend == "gloo" and world > 1 else ""),
                "gb_per_s_of_samples": round(samples / elapsed * 16 / 1e9, 3),
            },
            "per_rank": {"samples_per_s_mega": [round(r, 1) for r in rates], "min": round(min(rates), 1),
                         "max": round(max(rates), 1)},
            "parity": ({k: v for k, v in parity.items() if k not in ("err_sq_sum", "count")} if parity else None),
            "roofline": roof,
            "kernels": per_kernel,
            "kernels_ms_per_step": {k: round(v[1] / max(args.steps, 1), 4) for k, v in prof.items()},
            # streaming diagnostics: kernel scopes the engine enqueued per step (a scope = one kernel family of one stage; the
            # plan groups launch several kernels per scope) and the host time of one enqueue (includes the host replay of the
            # reference's Add / Get bookkeeping under CPQ_CALLS_ANY / the native schedule)
            "kernel_scopes_per_step": round(sum(v[0] for v in prof.values()) / max(args.steps, 1), 2),
            # calls below 65536 samples: `value` / `ms_per_step` are timed without the library's event profiling, the
            # per-kernel numbers come from a second pass of the same K steps with it (ms per step of that pass here)
            "kernel_times_from": ("second pass of the same steps with event profiling on" if separate_profile else "the timed region"),
            "ms_per_step_profiled_pass": round(profiled_elapsed / args.steps * 1e3, 4) if profiled_elapsed else None,
            "host_enqueue_us_per_step": round(host_side.get("enqueue_s", 0.0) / max(args.steps, 1) * 1e6, 1),
            "call_mode": args.call_mode,
            "setup_s": round(setup_s, 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(L, use_eq, args.saturation, args.eq_preset)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
