#!/usr/bin/env python3
"""Seeded soak of the random GPU parity sweeps beyond the seeds tests/ runs: the EQ parameter space, the convolver's
configuration space, random transition sequences through the whole chain, ragged calls at random quanta (CPQ_CALLS_ANY) with mid-run IR
reloads and the EQ on the same callbacks, internal FFT partitions larger than the block, transition sequences at arbitrary quanta.  Prints one line per failing seed and a summary;
exit code 1 on any failure.  usage: python tools/soak_gpu.py [first_seed] [n_seeds]   (on the GPU box)"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import convopeq_amd as amd          # noqa: E402
import oracle_lib                   # noqa: E402
import numpy as np                  # noqa: E402
import test_gpu_parity as T         # noqa: E402
import test_gpu_ragged as R         # noqa: E402


def ragged_case(seed):
    """CPQ_CALLS_ANY: random quantum (any integer 1..2048), IR length, scale, direct head, two plans in one engine, calls of
    random ragged sizes (incl. single samples and the engine's maximum) vs the oracle's Add / Get emulation."""
    rng = np.random.default_rng(seed)
    quantum = int(rng.choice([int(rng.integers(1, 2049)), 441, 480, 96, 64, 512, 1000, 37]))
    max_blocks = int(rng.integers(1, 6))
    max_call = quantum * max_blocks
    taps = [int(rng.choice([int(rng.integers(1, 3000)), int(rng.integers(3000, 40000)), 131072])) for _ in range(2)]
    direct = bool(rng.random() < 0.3)
    scale = float(rng.choice([1.0, 0.37]))
    total = int(rng.integers(6000, 30000)) + 3 * max(taps) // 2
    sizes = []
    while sum(sizes) < total:
        r = rng.random()
        sizes.append(int(max_call if r < 0.15 else (min(max_call, rng.integers(1, 8)) if r < 0.3 else rng.integers(1, max_call + 1))))
    n = sum(sizes)
    S = 2
    irs = [R_ORACLE.gen_ir(taps[c // 2], stream=seed % 1000 + c // 2, channel=c % 2) for c in range(2 * S)]
    x = R.make_inputs(R_ORACLE, [seed % 1000 + s for s in range(S)], n)
    with_eq = bool(rng.random() < 0.5)                 # the whole chain: the EQ runs on the same callbacks (AGC, a gain ramp)
    eng = amd.BatchedEngine(S, block_size=quantum, max_ir_len=max(taps), max_blocks_per_call=max_blocks, call_mode=amd.CPQ_CALLS_ANY)
    try:
        for s in range(S):
            eng.set_impulse(s, irs[2 * s], irs[2 * s + 1], scale=scale, direct_head=direct)
        if not with_eq:
            # now and then stream 1 gets a new IR in mid-run (a new NUC from silence: SetImpulse) while stream 0 plays on
            reload_at = int(rng.integers(1, len(sizes))) if (rng.random() < 0.4 and len(sizes) > 1) else None
            new_len = int(rng.integers(1, max(taps) + 1))
            new_irs = [R_ORACLE.gen_ir(new_len, stream=seed % 1000 + 7, channel=ch) for ch in range(2)]
            ys, pos = [], 0
            for k, m in enumerate(sizes):
                if k == reload_at:
                    eng.set_impulse(1, new_irs[0], new_irs[1], scale=scale, direct_head=direct)
                ys.append(eng.conv_process(np.ascontiguousarray(x[:, pos:pos + m])))
                pos += m
            y = np.concatenate(ys, axis=1)
            for c in range(2 * S):
                if c >= 2 and reload_at is not None:
                    p0 = sum(sizes[:reload_at])
                    ref0, _ = R.oracle_calls(R_ORACLE, irs[c], x[c], quantum, sizes[:reload_at], direct=direct, scale=scale)
                    ref1, _ = R.oracle_calls(R_ORACLE, new_irs[c - 2], x[c][p0:], quantum, sizes[reload_at:], direct=direct, scale=scale)
                    ref = np.concatenate([ref0, ref1])
                else:
                    ref, _ = R.oracle_calls(R_ORACLE, irs[c], x[c], quantum, sizes, direct=direct, scale=scale)
                R.check(y[c], ref)
            return
        pos_p = [R_ORACLE.eq_params_bench(float(rng.choice([0.0, 0.2, 0.7]))) for _ in range(S)]
        pos_p[0].agcEnabled = int(rng.random() < 0.5)
        gain0 = float(rng.uniform(-6, 6))
        pos_p[1].totalGainDb = gain0
        if rng.random() < 0.5:
            eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL)
        for s in range(S):
            eng.set_eq_params(s, R._copy_eq(pos_p[s], amd.eq_params_default()))
        change_at = int(rng.integers(0, len(sizes)))
        new_gain = float(rng.uniform(-9, 3))
        outs, pos = [], 0
        for k, m in enumerate(sizes):
            if k == change_at:
                pos_p[1].totalGainDb = new_gain
                eng.set_eq_params(1, R._copy_eq(pos_p[1], amd.eq_params_default()))
            outs.append(eng.process(np.ascontiguousarray(x[:, pos:pos + m])))
            pos += m
        y = np.concatenate(outs, axis=1)
        for s in range(S):
            conv = [R.oracle_calls(R_ORACLE, irs[2 * s + ch], x[2 * s + ch], quantum, sizes, direct=direct, scale=scale)[0] for ch in range(2)]
            po = R_ORACLE.EqParams.from_buffer_copy(pos_p[s])
            if s == 1:
                po.totalGainDb = gain0
            state, pos = np.zeros(168), 0
            for k, m in enumerate(sizes):
                if k == change_at and s == 1:
                    po.totalGainDb = new_gain
                yl, yr, state = R_ORACLE.eq_process_stereo(conv[0][pos:pos + m], conv[1][pos:pos + m], po, block=quantum, state=state)
                err = max(R.rms(y[2 * s, pos:pos + m] - yl), R.rms(y[2 * s + 1, pos:pos + m] - yr))
                assert err <= 1e-12, (s, k, err)
                pos += m
    finally:
        eng.close()

oracle_lib.lib()
R_ORACLE = oracle_lib


def partition_case(seed):
    """Whole-block engines with an internal FFT partition larger than the block (uniform schedule, plain IR): random block,
    partition, IR length (single-layer, multi-layer LTI and time-varying plans), scale, direct head; calls of whole
    partitions.  The output is the reference's at the caller's block size whatever the partition."""
    rng = np.random.default_rng(seed)
    block = int(rng.choice([64, 128, 256, 512, 1024, 2048]))
    cands = [p for p in (128, 256, 512, 1024, 2048, 4096) if p > block]
    part = int(rng.choice(cands)) if rng.random() < 0.8 else -1
    ir_len = int(np.exp(rng.uniform(np.log(100), np.log(200000))))
    eff_part = part if part > 0 else 4096
    calls_parts = int(rng.integers(1, 5)) if part > 0 else int(rng.integers(8, 12))
    max_blocks = calls_parts * eff_part // block
    scale = float(rng.choice([1.0, rng.uniform(0.2, 1.5)]))
    direct = bool(rng.random() < 0.3)
    irs = [R_ORACLE.gen_ir(ir_len, seed=0x3333 + seed, channel=ch) for ch in range(2)]
    eng = amd.BatchedEngine(1, block_size=block, max_ir_len=ir_len, max_blocks_per_call=max_blocks, partition_size=part)
    try:
        p_used = eng.partition_size()
        eng.set_impulse(0, irs[0], irs[1], scale=scale, direct_head=direct)
        total = ir_len + 3 * 4096 + int(rng.integers(0, 30000))
        sizes = []
        while sum(sizes) < total:
            sizes.append(int(rng.integers(1, max_blocks * block // p_used + 1)) * p_used)
        n = sum(sizes)
        x = R.make_inputs(R_ORACLE, [seed % 1000], n)
        outs, pos = [], 0
        for m in sizes:
            outs.append(eng.conv_process(np.ascontiguousarray(x[:, pos:pos + m])))
            pos += m
        y = np.concatenate(outs, axis=1)
    finally:
        eng.close()
    for c in range(2):
        nuc = R_ORACLE.Nuc()
        assert nuc.set_impulse(irs[c], block, scale=scale, direct=direct)
        ref = nuc.run(x[c], block)
        nuc.close()
        err = R.rms(y[c] - ref)
        assert err <= 1e-12, (seed, block, part, p_used, ir_len, direct, err)
def eq_chained_case(seed):
    """Long EQ calls on few channels (chained spans, svf_kernels.hip): 1 ... 6 streams, 2 ... 10 whole spans + a ragged remainder
    of any length per call, random per-stream parameters, now and then a NaN / Inf / huge sample or a hot stretch; vs the oracle."""
    rng = np.random.default_rng(seed)
    S = int(rng.integers(1, 7))
    spans = int(rng.integers(2, 11))
    rem = int(rng.choice([0, int(rng.integers(1, 8192)), 512, 1024, 441]))
    n = spans * 8192 + rem
    calls = int(rng.integers(1, 4))
    x = np.stack([oracle_lib.gen_pcm(calls * n, stream=seed * 16 + c // 2, channel=c % 2) for c in range(2 * S)])
    dirty = set()
    if rng.random() < 0.4:
        for _ in range(int(rng.integers(1, 4))):
            c, i = int(rng.integers(0, 2 * S)), int(rng.integers(0, calls * n))
            x[c, i] = rng.choice([np.nan, np.inf, -np.inf, 1e300, 3e10, -5e12])
            dirty.add(c // 2)
    if rng.random() < 0.3:
        a = int(rng.integers(0, calls * n - 100))
        x[:, a:a + int(rng.integers(100, 30000))] *= float(rng.choice([20.0, 64.0, 300.0]))
    eng = amd.BatchedEngine(S, block_size=512, max_ir_len=512, max_blocks_per_call=(n + 511) // 512, call_mode=amd.CPQ_CALLS_ANY)
    pos = []
    for s in range(S):
        po = T._random_eq_params(oracle_lib, rng, allow_ms=False)
        po.filterStructure = 0
        po.agcEnabled = 0
        for i in range(20):         # keep the bands inside what the guard proof accepts, so that the time-parallel kernels run
            po.bands[i].gain = float(np.clip(po.bands[i].gain, -12.0, 12.0))
            po.bands[i].q = float(np.clip(po.bands[i].q, 0.3, 8.0))
        pos.append(po)
        eng.set_eq_params(s, T._copy_params(po, amd.eq_params_default()))
    y = np.concatenate([eng.eq_process(x[:, k * n:(k + 1) * n]) for k in range(calls)], axis=1)
    launches, gave_up = eng.eq_chain_status()
    eng.close()
    assert gave_up == 0, (seed, "a hand-over gave up")
    assert np.all(np.isfinite(y)), seed
    for s in range(S):
        yl, yr, _ = oracle_lib.eq_process_stereo(x[2 * s], x[2 * s + 1], pos[s], block=512)
        err = max(np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max())
        # behind a sample of 1e10 ... 1e300 the states decay from up to 1e15 through the fast path, whose rounding is relative to them
        assert err <= (1e-3 if s in dirty else 1e-11), (seed, s, S, spans, rem, calls, err, launches)


first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fails = []
t0 = time.time()
for seed in range(first, first + count):
    cases = [("eq", lambda: T.test_eq_random_parameter_sweep(amd, oracle_lib, seed, 48000.0 if seed % 3 else 96000.0, 512 if seed % 2 else 256)),
             ("conv", lambda: T.test_convolver_random_configuration_sweep(amd, oracle_lib, seed)),
             ("chain", lambda: T.test_whole_chain_random_transition_sequence(amd, oracle_lib, seed)),
             ("ragged", lambda: ragged_case(seed)),
             ("partition", lambda: partition_case(seed)),
             ("ragged-chain", lambda: R.test_whole_chain_random_transitions_at_arbitrary_quantum(amd, oracle_lib, seed)),
             ("eq-chained", lambda: eq_chained_case(seed))]
    for name, fn in cases:
        try:
            fn()
        except Exception as e:          # noqa: BLE001
            fails.append((name, seed))
            print(f"FAIL {name} seed {seed}: {type(e).__name__}: {str(e)[:300]}", flush=True)
            traceback.print_exc(limit=3)
    if (seed - first) % 10 == 9:
        print(f"... {seed - first + 1} seeds, {len(fails)} failures, {time.time() - t0:.0f} s", flush=True)
print(f"soak: seeds {first}..{first + count - 1}, {7 * count} cases, {len(fails)} failures: {fails}")
sys.exit(1 if fails else 0)
