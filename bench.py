#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native ConvoPeq hot path.

Metric (BASELINE.json): Mega stereo-samples/s convolved (131072-tap IR, blk=512), fp64.
One *step* = one pass of the hot path over one batch: every one of S stereo streams pushes T blocks of 512
samples through the convolver (131072-tap private IR per channel, reference semantics) and, unless --no-eq,
the 20-band SVF EQ -- one cpq_engine_process_block_device() call with inputs already resident in HBM.

N = 1 runs BASELINE.json configs[1] (256 stereo streams).  N > 1 (launched by torch.distributed.run, one rank
per GPU) keeps the per-GPU workload fixed (weak scaling, streams sharded across ranks, no data-path
collective); the only collective is the final RCCL all-reduce of the counters.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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


def bench_eq_params(amd, saturation):
    """SURVEY.md 8(d) EQ bench preset."""
    freqs = [25.0, 40.0, 63.0, 100.0, 160.0, 250.0, 400.0, 630.0, 1000.0, 1600.0, 2500.0, 4000.0, 6300.0,
             10000.0, 11000.0, 12500.0, 14000.0, 16500.0, 18000.0, 19500.0]
    gains = [3, -2, 4, -3, 2, -4, 3, -2, 1.5, -1.5, 2, -2, 3, -3, 1, -1, 2, -2, 1, -1]
    p = amd.eq_params_default()
    for i in range(20):
        b = p.bands[i]
        b.frequency, b.gain, b.q = freqs[i], gains[i], 1.41
        b.enabled, b.channel_mode = 1, 0
        b.type = 0 if i == 0 else (2 if i == 19 else 1)
    p.nonlinear_saturation = saturation
    return p


# ----------------------------------------------------------------------------- CPU baseline (oracle, "port")
def cpu_baseline(ir_len, use_eq, saturation, target_seconds=6.0):
    """Times the CPU oracle (restatement of the reference NUC + SVF EQ) on the host cores of this box.
    Bounded sample: one stereo stream per thread, each with its own 131072-tap IRs, blocks of 512."""
    import oracle_lib as O
    O.lib()
    cores = max(1, min(os.cpu_count() or 1, 16))
    n_blocks = 1024                       # per thread and pass: 524288 samples per channel
    po = O.eq_params_bench(saturation)
    done = [0] * cores
    stop_at = [0.0]
    first = {}                            # thread 0's first 64 blocks of output: the parity sample

    def work(tid):
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
                y[0], y[1], _ = O.eq_process_stereo(y[0], y[1], po, state=state)
            if tid == 0 and not first:
                first["y"] = np.stack([y[0][:64 * B].copy(), y[1][:64 * B].copy()])
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
    return {"value": round(total / dt / 1e6, 3), "unit": "Mega stereo-samples/s", "cores": cores, "kind": "port",
            "sample": f"{cores} stereo streams (one per thread), {ir_len}-tap IR each, blk {B}, "
                      f"{total // cores} samples per stream, conv{'+EQ' if use_eq else ''}, {dt:.1f} s wall; "
                      "oracle = C restatement of the reference NUC schedule (own radix-2 FFT, not IPP)"}, first.get("y")


def parity_against(amd, ref, ir_len, use_eq, saturation, device):
    """fp64 RMS / max-abs difference between the HIP path and the CPU baseline's own output for stream 0 (first 64 blocks
    from reset, same IR / PCM / EQ preset): the oracle here is the checker, nothing of it is timed or shipped."""
    n = ref.shape[1]
    eng = amd.BatchedEngine(1, block_size=B, max_ir_len=ir_len, max_blocks_per_call=n // B, device=device)
    eng.set_impulse(0, gen_ir(ir_len, 0, 0), gen_ir(ir_len, 0, 1))
    x = np.stack([gen_pcm(n, 0, 0), gen_pcm(n, 0, 1)])
    if use_eq:
        eng.set_eq_params(amd.CPQ_ALL_STREAMS, bench_eq_params(amd, saturation))
        y = eng.process(x)
    else:
        y = eng.conv_process(x)
    eng.close()
    d = y - ref
    return {"rms_err": float(np.sqrt(np.mean(d * d))), "max_abs_err": float(np.abs(d).max()),
            "signal_rms": float(np.sqrt(np.mean(ref * ref))), "target_rms_err": 1e-12,
            "sample": f"stream 0, {n} samples per channel from reset, conv{'+EQ' if use_eq else ''}, GPU vs the CPU baseline's output"}


def load_pmc_traffic(path, kernel):
    """HBM bytes per launch of `kernel` from a committed rocprofv3 PMC summary (profiles/*.json), or None."""
    try:
        with open(path) as f:
            d = json.load(f)
        # the profiler id k_fdl_mac covers the workgroup-cooperative variant of long calls (k_fdl_mac_wg in rocprof
        # names) and the register-tile variants of short ones (k_fdl_mac<TT, PF>, summarised as "k_fdl_mac")
        if kernel == "k_fdl_mac" and "k_fdl_mac_wg" in d:
            return d["k_fdl_mac_wg"].get("hbm_bytes_per_launch")
        if kernel == "k_fdl_mac:tile":
            kernel = "k_fdl_mac"
        if kernel in d:
            return d[kernel].get("hbm_bytes_per_launch")
        cands = [v for k, v in d.items() if k.startswith(kernel) and isinstance(v, dict)]
        return max((v.get("hbm_bytes_per_launch") for v in cands), default=None)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--streams", type=int, default=256, help="stereo streams per GPU")
    ap.add_argument("--ir-len", type=int, default=131072)
    ap.add_argument("--block", type=int, default=512, help="diagnostic: block size")
    ap.add_argument("--partition", type=int, default=0, help="internal FFT partition size (0 = block size)")
    ap.add_argument("--blocks-per-call", type=int, default=64)
    ap.add_argument("--mac-tile", type=int, default=0)
    ap.add_argument("--no-eq", action="store_true")
    ap.add_argument("--eq-only", action="store_true", help="diagnostic: time the EQ kernel alone")
    ap.add_argument("--saturation", type=float, default=0.2)
    ap.add_argument("--shared-ir", action="store_true")
    ap.add_argument("--schedule", choices=["uniform", "nuc"], default="uniform",
                    help="uniform: one partition size for the whole h_eff (headline, HBM-roofline path); nuc: the reference's "
                         "own non-uniform schedule run natively (BASELINE.json configs[3])")
    ap.add_argument("--exact", action="store_true", help="plain linear convolution instead of reference h_eff")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-buffers", action="store_true", help="diagnostic: host-pointer entry point (PCIe-inclusive rate)")
    ap.add_argument("--pinned", action="store_true", help="with --host-buffers: pin the host buffers (cpq_host_register)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (single-GPU rehearsal of N>1)")
    ap.add_argument("--pmc-json", default=os.path.join(ROOT, "profiles", "r01k_pmc_traffic.json"))
    args = ap.parse_args()

    import torch
    import convopeq_amd as amd

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()      # (== local_rank on a full node)
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    global B
    B = args.block
    S, T, L = args.streams, args.blocks_per_call, args.ir_len
    n = T * B
    use_eq = not args.no_eq
    eng = amd.BatchedEngine(S, block_size=B, max_ir_len=L, max_blocks_per_call=T,
                            semantics=amd.CPQ_SEM_EXACT if args.exact else amd.CPQ_SEM_REFERENCE,
                            device=dev_index, mac_tile=args.mac_tile, partition_size=args.partition,
                            schedule=amd.CPQ_SCHED_REFERENCE_NUC if args.schedule == "nuc" else amd.CPQ_SCHED_UNIFORM)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)

    # synthetic IRs and PCM; stream ids are global so every rank convolves different streams
    t_setup = time.perf_counter()
    g0 = rank * S
    if args.shared_ir:
        eng.set_impulse(amd.CPQ_ALL_STREAMS, gen_ir(L, 0, 0), gen_ir(L, 0, 1))
    else:
        for s in range(S):
            eng.set_impulse(s, gen_ir(L, g0 + s, 0), gen_ir(L, g0 + s, 1))
    if use_eq:
        eng.set_eq_params(amd.CPQ_ALL_STREAMS, bench_eq_params(amd, args.saturation))
    host = np.empty((2 * S, n))
    for s in range(S):
        for ch in range(2):
            host[2 * s + ch] = gen_pcm(n, g0 + s, ch)
    d_in = torch.from_numpy(host).cuda()
    d_out = torch.empty_like(d_in)
    plan = eng.plan()
    P = args.partition if args.partition else B          # internal FFT partition size
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

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    eng.profile_enable(True)
    eng.profile_reset()

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read()

    # The same MAC path with ONE partition per call (the reference's own call pattern): every FDL and IR row is
    # streamed for a single output row, so the kernel is HBM-bound there, whereas at T blocks per call it is past the
    # fp64 ridge.  Measured after the timed region (not part of `value`) to give the roofline object both regimes.
    prof1 = None
    if not (args.eq_only or args.host_buffers or args.schedule == "nuc") and rank == 0:
        P1 = args.partition if args.partition else B
        eng.profile_reset()
        for _ in range(40):
            eng.conv_process_device(d_in.data_ptr(), d_out.data_ptr(), P1)
        torch.cuda.synchronize()
        prof1 = eng.profile_read()
    eng.profile_enable(False)

    samples = float(S) * n * args.steps          # stereo samples this rank processed
    if dist is not None:
        red_dev = "cuda" if args.dist_backend == "nccl" else "cpu"
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        ss = torch.tensor([samples], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(ss, op=dist.ReduceOp.SUM)
        elapsed, samples = tt.item(), ss.item()

    if rank == 0:
        n_ch = 2 * S
        # algorithmic HBM bytes per launch: every row / sample a kernel needs is moved once
        spec_bytes = P * 16
        ir_rows = (2 if args.shared_ir else n_ch) * k_parts
        # layers the convolver actually runs: (partition size, partitions, partitions per step)
        if args.schedule == "nuc":
            layers = [(plan.part_size[l], plan.num_parts_ir[l], n / plan.part_size[l]) for l in range(plan.num_layers)]
        else:
            layers = [(P, k_parts, Tp)]
        ir_mult = 2 if args.shared_ir else n_ch
        alg_bytes = {      # per STEP; one launch per step and kernel under the uniform schedule
            "k_rfft_fwd_ols": sum(n_ch * nb * (pl * 8 + pl * 16) for pl, _, nb in layers),
            "k_fdl_mac": sum((n_ch * (kl + nb - 1) + ir_mult * kl + n_ch * nb) * pl * 16 for pl, kl, nb in layers),
            "k_fdl_mac_dcnyq": sum(n_ch * (kl + nb - 1 + nb) * 16 + ir_mult * kl * 16 for pl, kl, nb in layers),
            "k_rfft_inv_ols": sum(n_ch * nb * (pl * 16 + pl * 8) for pl, _, nb in layers),
            "k_svf_cascade_tp": n_ch * n * 16,
            "k_svf_cascade": n_ch * n * 16,
            "k_convproc_mix": n_ch * n * 16 * max(0, len(layers) - 1),      # delay-line write / read-add of the tail layers
        }
        # fp64 operations per step (FMA = 2): MAC 8 per complex MAC; SVF ~35 fp64 instructions per band-sample
        # the cooperative MAC kernel (>= 48 rows per call) spends 3 real FMAs per complex MAC (Gauss), the tile kernels 4
        mac_flop = lambda nb: 6.0 if nb >= 48 else 8.0
        alg_flops = {"k_fdl_mac": sum(mac_flop(nb) * n_ch * nb * kl * pl for pl, kl, nb in layers),
                     "k_svf_cascade_tp": 2.0 * 35 * 20 * n_ch * n}
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
            if name in alg_flops:
                per_kernel[name]["fp64_tflops"] = round(alg_flops[name] / step_s / 1e12, 2)
        tot = {k: v["avg_launch_ms"] * v["launches"] for k, v in per_kernel.items()}
        dominant = max(tot, key=tot.get)
        co_dominant = sorted(k for k in tot if tot[k] >= 0.9 * tot[dominant])
        # the FDL MAC and the SVF cascade tie within a few percent at the default config; when they do, the roofline
        # object describes the HBM-streaming one (the kernel north_star defines the roofline on), both are listed
        if "k_fdl_mac" in co_dominant:
            dominant = "k_fdl_mac"
        dk = per_kernel[dominant]
        hbm_regime = None
        if prof1 and prof1.get("k_fdl_mac", (0, 0.0))[0] > 0:
            cnt1, ms1 = prof1["k_fdl_mac"]
            b1 = (n_ch * k_parts + ir_rows + n_ch) * spec_bytes
            gbs1 = b1 / (ms1 / cnt1 * 1e-3) / 1e9
            hbm_regime = {"kernel": "k_fdl_mac", "blocks_per_call": P // B, "algorithmic_bytes_per_launch": b1,
                          "avg_launch_ms": round(ms1 / cnt1, 4), "achieved": round(gbs1, 1),
                          "frac": round(gbs1 / HBM_PEAK_GBS, 4), "launches": cnt1,
                          "traffic": load_pmc_traffic(args.pmc_json, "k_fdl_mac:tile")}
        flop_per_byte = alg_flops["k_fdl_mac"] / alg_bytes["k_fdl_mac"]
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
                "workload": f"{S} stereo streams per GPU, {L}-tap IR "
                            f"({'one shared stereo IR' if args.shared_ir else 'private IR per channel'}), blk {B}, "
                            f"fp64 overlap-save conv{' + 20-band SVF EQ (sat %.1f)' % args.saturation if use_eq else ''}"
                            f" [BASELINE.json configs[1]{'' if (S == 256 and L == 131072) else ' (modified)'}]",
                "streams_per_gpu": S, "ir_taps": L, "block": B, "blocks_per_call": T,
                "schedule": (f"uniform overlap-save, FFT partition P={P}, K={k_parts} partitions of "
                             f"{'h' if args.exact else 'h_eff (reference NUC semantics at blk 512)'}, "
                             f"{T} blocks ({n} samples) per call = {Tp} partitions per FDL pass") if args.schedule == "uniform"
                            else ("non-uniform (the reference's own layer plan run natively): " +
                                  " + ".join(f"{kl} x {pl}" for pl, kl, _ in layers) + f" partitions, {T} blocks per call"),
                "partition": P,
                "eq": use_eq, "parallelism": f"streams sharded, {world} rank(s)",
                "gb_per_s_of_samples": round(samples / elapsed * 16 / 1e9, 3),
            },
            "roofline": {
                "kernel": dominant, "bound": "hbm", "achieved": dk["achieved_gbs"], "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(dk["achieved_gbs"] / HBM_PEAK_GBS, 4),
                "traffic": load_pmc_traffic(args.pmc_json, dominant),
                "algorithmic_bytes_per_launch": dk["algorithmic_bytes_per_launch"],
                "avg_launch_ms": dk["avg_launch_ms"], "launches": dk["launches"],
                "co_dominant_kernels": co_dominant,
                "note": ("k_svf_cascade_tp is fp64-VALU issue bound, not HBM bound: 20 sequential nonlinear bands per "
                         "sample (~35 fp64 instructions per band-sample) against 16 B of HBM traffic per sample"
                         if dominant.startswith("k_svf") else "HBM stream of FDL and IR spectra"),
                "fp64_vector": {"achieved_tflops": dk.get("fp64_tflops"), "peak_tflops": FP64_VECTOR_PEAK_TFLOPS},
                # k_fdl_mac at T partitions per call executes 6 K T flop (3 FMAs per complex MAC) per 16 (2K + T) bytes:
                # at the fp64 ridge (78.6 TFLOP/s / 8 TB/s = 9.8 flop/B) around T = 64 at K = 259
                "mac_flop_per_byte": round(flop_per_byte, 2),
                "ridge_flop_per_byte": round(FP64_VECTOR_PEAK_TFLOPS * 1e3 / HBM_PEAK_GBS, 2),
                "hbm_regime": hbm_regime,
            },
            "kernels": per_kernel,
            "kernels_ms_per_step": {k: round(v[1] / max(args.steps, 1), 4) for k, v in prof.items()},
            "setup_s": round(setup_s, 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ref = cpu_baseline(L, use_eq, args.saturation)
            if ref is not None and B == 512 and not (args.exact or args.partition):
                out["cpu_baseline"]["parity"] = parity_against(amd, ref, L, use_eq, args.saturation, dev_index)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
