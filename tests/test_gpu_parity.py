"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances: north_star asks <= 1e-12 RMS per sample in fp64 versus the reference CPU path; the tests hold
the convolver to 1e-13 RMS (absolute, signal RMS ~0.1-1) and the SVF cascade to bit equality or 1e-15.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

B = 512


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


def make_inputs(O, n_streams, n_samples, start=0):
    x = np.empty((2 * n_streams, n_samples))
    for s in range(n_streams):
        for ch in range(2):
            x[2 * s + ch] = O.gen_pcm(n_samples, stream=s, channel=ch, start=start)
    return x


def oracle_conv(O, irs, x, block=B):
    y = np.empty_like(x)
    for c in range(x.shape[0]):
        nuc = O.Nuc()
        assert nuc.set_impulse(irs[c], block)
        y[c] = nuc.run(x[c], block)
        nuc.close()
    return y


@pytest.fixture(scope="module")
def amd():
    import convopeq_amd
    return convopeq_amd


def test_config1_single_stream_4096_taps(amd, oracle):
    """BASELINE.json configs[0]: 1 stereo stream, 4096-tap IR, 512-sample blocks, EQ bypassed, one block per call."""
    O = oracle
    irs = [O.gen_ir(4096, stream=0, channel=ch) for ch in range(2)]
    x = make_inputs(O, 1, 64 * B)
    ref = oracle_conv(O, irs, x)
    eng = amd.BatchedEngine(1, max_ir_len=4096, max_blocks_per_call=1)
    eng.set_impulse(0, irs[0], irs[1])
    assert eng.is_ready() and eng.latency() == 512
    y = np.concatenate([eng.conv_process(x[:, i * B:(i + 1) * B]) for i in range(64)], axis=1)
    err = rms(y - ref)
    print("config1 rms err", err, "signal rms", rms(ref))
    assert err <= 1e-13
    eng.close()


@pytest.mark.parametrize("blocks_per_call,tile", [(1, 16), (5, 4), (16, 8), (24, 16), (48, 32), (32, 0), (40, 0), (112, 0)])
def test_long_ir_reference_semantics(amd, oracle, blocks_per_call, tile):
    """131072-tap IRs, private per channel: reference semantics (tail gain 1.4375, L1 lag +1408) via h_eff,
    time-batched T blocks per call, compared with the oracle's Add/Get schedule emulation."""
    O = oracle
    S = 2
    irs = [O.gen_ir(131072, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    n_blocks = {5: 335, 32: 320, 40: 320, 112: 336}.get(blocks_per_call, 336)
    x = make_inputs(O, S, n_blocks * B)
    ref = oracle_conv(O, irs, x)
    eng = amd.BatchedEngine(S, max_ir_len=131072, max_blocks_per_call=blocks_per_call, mac_tile=tile)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    step = blocks_per_call * B
    ys = [eng.conv_process(x[:, o:o + step]) for o in range(0, n_blocks * B, step)]
    y = np.concatenate(ys, axis=1)
    err = rms(y - ref)
    print(f"T={blocks_per_call} tile={tile} rms err {err:.3e} signal rms {rms(ref):.3f}")
    assert err <= 1e-13
    # ragged last call + reset
    eng.conv_reset()
    y2 = eng.conv_process(x[:, :step])
    assert np.array_equal(y2, ys[0])
    eng.close()


@pytest.mark.parametrize("partition,blocks_per_call,tile", [(1024, 16, 0), (2048, 32, 0), (4096, 8, 0), (4096, 64, 0), (4096, 512, 0),
                                                           (4096, 128, 16), (4096, 64, 4)])
def test_larger_internal_partition(amd, oracle, partition, blocks_per_call, tile):
    """cpq_engine_desc.partition_size: the FFT partition P exceeds the caller's block (time-batched calls of whole
    partitions); h_eff is still the one the reference derives for the 512-sample block, so the output is the oracle's
    Add/Get emulation at block 512.  P = 4096 is the throughput path of bench.py (persistent 4096-point FFT kernels)."""
    O = oracle
    S = 2
    irs = [O.gen_ir(131072, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    n_call = blocks_per_call * B
    calls = max(2, (336 * B) // n_call)
    x = make_inputs(O, S, calls * n_call)
    ref = oracle_conv(O, irs, x)
    eng = amd.BatchedEngine(S, max_ir_len=131072, max_blocks_per_call=blocks_per_call, partition_size=partition, mac_tile=tile)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    y = np.concatenate([eng.conv_process(x[:, o:o + n_call]) for o in range(0, calls * n_call, n_call)], axis=1)
    err = rms(y - ref)
    print(f"P={partition} T={blocks_per_call} tile={tile}: rms err {err:.3e}")
    assert err <= 1e-13
    with pytest.raises(amd.CpqError):            # calls must be whole partitions
        eng.conv_process(x[:, :B])
    eng.close()


@pytest.mark.parametrize("block,blocks_per_call,semantics,schedule,expect", [
    (512, 64, "ref", "uniform", 4096), (512, 3, "ref", "uniform", 512), (128, 256, "ref", "uniform", 4096),
    (128, 64, "ref", "uniform", 512), (128, 12, "ref", "uniform", 512), (256, 3, "ref", "uniform", 256),
    (1024, 32, "ref", "uniform", 4096), (1024, 32, "exact", "uniform", 4096), (1024, 8, "exact", "uniform", 1024),
    (512, 64, "ref", "nuc", 512)])
def test_automatic_partition_choice(amd, oracle, block, blocks_per_call, semantics, schedule, expect):
    """CPQ_PARTITION_AUTO: 4096 where the calls are eight or more whole 4096-sample partitions, else 512, else the block; never for
    the reference's own schedule.  The output is the oracle's Add/Get
    emulation at the caller's block size whatever the engine picks."""
    O = oracle
    L = 20000
    irs = [O.gen_ir(L, stream=0, channel=c) for c in range(2)]
    n_call = blocks_per_call * block
    calls = max(2, (3 * L) // n_call)
    x = make_inputs(O, 1, calls * n_call)
    eng = amd.BatchedEngine(1, block_size=block, max_ir_len=L, max_blocks_per_call=blocks_per_call,
                            partition_size=amd.CPQ_PARTITION_AUTO,
                            semantics=amd.CPQ_SEM_EXACT if semantics == "exact" else amd.CPQ_SEM_REFERENCE,
                            schedule=amd.CPQ_SCHED_REFERENCE_NUC if schedule == "nuc" else amd.CPQ_SCHED_UNIFORM)
    assert eng.partition_size() == expect
    eng.set_impulse(0, irs[0], irs[1])
    y = np.concatenate([eng.conv_process(x[:, o:o + n_call]) for o in range(0, calls * n_call, n_call)], axis=1)
    if semantics == "exact":
        ref = np.stack([np.convolve(x[c], irs[c])[:x.shape[1]] for c in range(2)])
        tol = 1e-12
    else:
        ref = oracle_conv(O, irs, x, block=block)
        tol = 1e-13
    assert rms(y - ref) <= tol
    eng.close()


def test_shared_ir_and_exact_semantics(amd, oracle):
    O = oracle
    from scipy.signal import fftconvolve
    S = 3
    irl, irr = O.gen_ir(20000, channel=0), O.gen_ir(20000, channel=1)
    x = make_inputs(O, S, 80 * B)
    eng = amd.BatchedEngine(S, max_ir_len=20000, max_blocks_per_call=8, semantics=amd.CPQ_SEM_EXACT)
    eng.set_impulse(amd.CPQ_ALL_STREAMS, irl, irr, scale=0.5)
    y = np.concatenate([eng.conv_process(x[:, o:o + 8 * B]) for o in range(0, 80 * B, 8 * B)], axis=1)
    for c in range(2 * S):
        ref = fftconvolve(x[c], 0.5 * (irl if c % 2 == 0 else irr))[:x.shape[1]]
        assert rms(y[c] - ref) <= 1e-13
    eng.close()


@pytest.mark.parametrize("mode", ["sequential", "auto"])
@pytest.mark.parametrize("sat", [0.0, 0.2])
def test_eq_cascade_matches_oracle(amd, oracle, sat, mode):
    """20-band SVF cascade, bench preset, stereo mode.  The sequential kernel keeps the
    reference's op order -> bit equality with the oracle restatement; the time-parallel kernel (auto) differs
    by rounding only (tolerance 1e-13 abs, north_star: 1e-12 RMS)."""
    O = oracle
    S = 4
    n = 16 * B
    x = make_inputs(O, S, 2 * n)
    po = O.eq_params_bench(sat)
    pa = amd.eq_params_default()
    for i in range(20):
        for f, g in (("frequency", "frequency"), ("gain", "gain"), ("q", "q"), ("enabled", "enabled"),
                     ("type", "type"), ("channelMode", "channel_mode")):
            setattr(pa.bands[i], g, getattr(po.bands[i], f))
    pa.nonlinear_saturation = sat
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=16)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, pa)
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL if mode == "sequential" else amd.CPQ_EQ_MODE_AUTO)
    eng.profile_enable(True)
    y = np.concatenate([eng.eq_process(x[:, :n]), eng.eq_process(x[:, n:])], axis=1)
    prof = eng.profile_read()
    used = "k_svf_cascade" if mode == "sequential" else "k_svf_cascade_tp"
    assert prof[used][0] == 2, prof          # the kernel under test is the one that ran
    worst = 0.0
    for s in range(S):
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], po)
        worst = max(worst, np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max())
    print("eq sat", sat, mode, "max abs diff", worst)
    assert worst <= (0.0 if mode == "sequential" else 1e-13)
    eng.close()


@pytest.mark.parametrize("mode", ["sequential", "auto"])
def test_svf_band_matches_the_reference_display_biquad(amd, oracle, mode):
    """External anchor for the band kernel (A11 / A12): the reference's own svfToDisplayBiquad
    (src/tests/EQProcessorMaxGainTests.cpp:67-87, compiled unmodified; fixture tests/golden/svf_display_biquad_ref.json) says
    which biquad an SVF band with given coefficients is.  With saturation 0 one enabled band of the HIP EQ must filter like
    scipy.signal.lfilter with that biquad -- for every band type, on both kernels (lane-skewed sequential, time-parallel)."""
    import json
    from scipy.signal import lfilter
    O = oracle
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "svf_display_biquad_ref.json")) as f:
        cases = json.load(f)["cases"]
    n = 32 * B
    x = make_inputs(O, 1, n)
    eng = amd.BatchedEngine(1, max_ir_len=512, max_blocks_per_call=32)
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL if mode == "sequential" else amd.CPQ_EQ_MODE_AUTO)
    worst = 0.0
    for case in cases:
        p = amd.eq_params_default()
        for i in range(20):
            p.bands[i].enabled = 0
        b0 = p.bands[0]
        b0.enabled, b0.type, b0.frequency, b0.gain, b0.q, b0.channel_mode = 1, case["type"], case["freq"], case["gain_db"], case["q"], 0
        p.nonlinear_saturation = 0.0
        p.total_gain_db = 0.0
        eng.set_eq_params(amd.CPQ_ALL_STREAMS, p)
        eng.eq_reset()
        y = eng.eq_process(x)
        bq = np.array([float.fromhex(v) for v in case["biquad"]])
        for c in range(2):
            ref = lfilter(bq[:3] / bq[3], bq[3:] / bq[3], x[c])
            err = np.abs(y[c] - ref).max() / max(1.0, np.abs(ref).max())
            worst = max(worst, err)
            assert err <= 2e-12, (case["type"], case["freq"], err)
    print("SVF band vs the reference's display biquad,", mode, "kernel: worst relative error", worst)
    eng.close()


@pytest.mark.parametrize("mode", ["sequential", "auto"])
def test_svf_band_matches_the_reference_cookbook_biquad(amd, oracle, mode):
    """External anchor for calcSVFCoeffs' peaking / shelving branches (A14): the reference's own cookbook designers
    (src/tests/EQBoundExcessBenchmark.cpp:188-245, compiled unmodified; fixture tests/golden/rbj_biquad_ref.json) give the
    z-domain biquad of a low shelf / peaking / high shelf band.  With saturation 0 one enabled band of the HIP EQ -- designed
    by the product from the same (float) parameters -- must filter like scipy.signal.lfilter with the reference's biquad.
    The two are the same transfer function in two forms; what is left is the direct form's own coefficient cancellation
    (tests/test_ref_eq_math_cpu.py states the measured bounds): <= 2e-9 from 200 Hz up, <= 1e-6 below, at 48 kHz."""
    import json
    from scipy.signal import lfilter
    O = oracle
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rbj_biquad_ref.json")) as f:
        cases = [c for c in json.load(f)["cases"] if c["sr"] == 48000.0]
    assert len(cases) >= 30
    n = 32 * B
    x = make_inputs(O, 1, n)
    eng = amd.BatchedEngine(1, max_ir_len=512, max_blocks_per_call=32)
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL if mode == "sequential" else amd.CPQ_EQ_MODE_AUTO)
    worst = 0.0
    for case in cases:
        p = amd.eq_params_default()
        for i in range(20):
            p.bands[i].enabled = 0
        b0 = p.bands[0]
        b0.enabled, b0.type, b0.frequency, b0.gain, b0.q, b0.channel_mode = 1, case["type"], case["freq"], case["gain_db"], case["q"], 0
        p.nonlinear_saturation = 0.0
        p.total_gain_db = 0.0
        eng.set_eq_params(amd.CPQ_ALL_STREAMS, p)
        eng.eq_reset()
        y = eng.eq_process(x)
        bq = np.array([float.fromhex(v) for v in case["biquad"]])
        tol = 2e-9 if case["freq"] >= 200.0 else 1e-6
        for c in range(2):
            ref = lfilter(bq[:3] / bq[3], bq[3:] / bq[3], x[c])
            err = np.abs(y[c] - ref).max() / max(1.0, np.abs(ref).max())
            worst = max(worst, err)
            assert err <= tol, (case["type"], case["freq"], case["gain_db"], case["q"], err)
    print("SVF band vs the reference's cookbook biquad,", mode, "kernel: worst relative error", worst)
    eng.close()


def test_conv_then_eq_whole_path(amd, oracle):
    O = oracle
    S = 2
    irs = [O.gen_ir(8192, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    x = make_inputs(O, S, 32 * B)
    ref = oracle_conv(O, irs, x)
    po = O.eq_params_bench(0.2)
    pa = amd.eq_params_default()
    for i in range(20):
        b = pa.bands[i]
        b.frequency, b.gain, b.q = po.bands[i].frequency, po.bands[i].gain, po.bands[i].q
        b.enabled, b.type, b.channel_mode = po.bands[i].enabled, po.bands[i].type, po.bands[i].channelMode
    for s in range(S):
        ref[2 * s], ref[2 * s + 1], _ = O.eq_process_stereo(ref[2 * s], ref[2 * s + 1], po)
    eng = amd.BatchedEngine(S, max_ir_len=8192, max_blocks_per_call=8)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, pa)
    y = np.concatenate([eng.process(x[:, o:o + 8 * B]) for o in range(0, 32 * B, 8 * B)], axis=1)
    err = rms(y - ref)
    print("conv+eq rms err", err)
    assert err <= 1e-13
    eng.close()


def _copy_params(po, pa):
    for i in range(20):
        b, o = pa.bands[i], po.bands[i]
        b.frequency, b.gain, b.q, b.enabled, b.type, b.channel_mode = o.frequency, o.gain, o.q, o.enabled, o.type, o.channelMode
    pa.nonlinear_saturation = po.nonlinearSaturation
    pa.total_gain_db = po.totalGainDb
    pa.filter_structure = po.filterStructure
    pa.agc_enabled = po.agcEnabled
    return pa


def test_config4_long_reverb_524288_taps(amd, oracle):
    """BASELINE.json configs[3] shape (524288-tap IR, blk 512; 2 of the 64 streams so the oracle finishes in
    seconds): three reference layers (512/4096/32768), L2 lag -232064, h_eff semantics."""
    O = oracle
    S, L, T = 2, 524288, 64
    irs = [O.gen_ir(L, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    n_blocks = 1280
    x = make_inputs(O, S, n_blocks * B)
    ref = oracle_conv(O, irs, x)
    eng = amd.BatchedEngine(S, max_ir_len=L, max_blocks_per_call=T)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    p = eng.plan()
    assert p.num_layers == 3 and p.lag[2] == -232064 and p.lti_valid == 1
    y = np.concatenate([eng.conv_process(x[:, o:o + T * B]) for o in range(0, n_blocks * B, T * B)], axis=1)
    err = rms(y - ref)
    print(f"config4 rms err {err:.3e} signal rms {rms(ref):.3f}")
    assert err <= 1e-12
    eng.close()


@pytest.mark.parametrize("mode", ["sequential", "auto"])
def test_eq_left_right_channel_modes_and_disabled_bands(amd, oracle, mode):
    """Left/Right channel modes run the scalar processBand arithmetic (no FMA, hard-clipped fastTanh); disabled
    bands are skipped; total gain applied at the end."""
    O = oracle
    S, n = 2, 8 * B
    x = make_inputs(O, S, n)
    po = O.eq_params_bench(0.2)
    for i in (1, 5, 9):
        po.bands[i].channelMode = 1
    for i in (2, 6, 10):
        po.bands[i].channelMode = 2
    for i in (3, 7):
        po.bands[i].enabled = 0
    po.totalGainDb = -6.5
    pa = _copy_params(po, amd.eq_params_default())
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=8)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, pa)
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL if mode == "sequential" else amd.CPQ_EQ_MODE_AUTO)
    y = eng.eq_process(x)
    worst = 0.0
    for s in range(S):
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], po)
        worst = max(worst, np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max())
    print("eq modes", mode, "max abs diff", worst)
    assert worst <= (0.0 if mode == "sequential" else 1e-13)
    eng.close()


def test_eq_per_stream_params_and_hot_signal(amd, oracle):
    """Different EQ parameters per stream; input scaled x40 so the saturation blend and the +-100 clamp act."""
    O = oracle
    S, n = 3, 16 * B
    x = 40.0 * make_inputs(O, S, n)
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=16)
    pos = []
    for s in range(S):
        po = O.eq_params_bench(0.1 * s)                # stream 0: linear (saturation 0) with large boosts
        for i in range(20):
            po.bands[i].gain = abs(po.bands[i].gain) * 4.0 if s == 0 else po.bands[i].gain * (1.0 + s)
        po.bands[4].type = 3 if s == 1 else 1          # a low-pass band on stream 1
        po.bands[12].type = 4 if s == 2 else 1         # a high-pass band on stream 2
        pos.append(po)
        eng.set_eq_params(s, _copy_params(po, amd.eq_params_default()))
    y = eng.eq_process(x)
    for s in range(S):
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], pos[s])
        if s == 0:
            assert np.abs(yl).max() == 100.0            # the +-100 clamp is exercised
        assert max(np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max()) <= 1e-11
    eng.close()


def test_eq_nonfinite_input_takes_guarded_path(amd, oracle):
    """NaN / Inf / huge samples: the time-parallel kernel must fall back to the guarded recurrence for that span
    and reproduce the reference's sanitising behaviour exactly like the sequential kernel."""
    O = oracle
    S, n = 1, 16 * B
    x = make_inputs(O, S, n)
    x[0, 100] = np.nan
    x[1, 5000] = np.inf
    x[0, 6000] = 1e300
    po = O.eq_params_bench(0.2)
    pa = _copy_params(po, amd.eq_params_default())
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=16)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, pa)
    y = eng.eq_process(x)
    yl, yr, _ = O.eq_process_stereo(x[0], x[1], po)
    assert np.all(np.isfinite(y))
    assert max(np.abs(y[0] - yl).max(), np.abs(y[1] - yr).max()) <= 1e-12
    eng.close()


def test_eq_then_conv_order(amd, oracle):
    O = oracle
    S = 1
    irs = [O.gen_ir(6000, stream=0, channel=ch) for ch in range(2)]
    x = make_inputs(O, S, 24 * B)
    po = O.eq_params_bench(0.2)
    el, er, _ = O.eq_process_stereo(x[0], x[1], po)
    ref = oracle_conv(O, irs, np.stack([el, er]))
    eng = amd.BatchedEngine(S, max_ir_len=6000, max_blocks_per_call=8)
    eng.set_impulse(0, irs[0], irs[1])
    eng.set_eq_params(0, _copy_params(po, amd.eq_params_default()))
    eng.set_order(amd.CPQ_ORDER_EQ_THEN_CONV)
    y = np.concatenate([eng.process(x[:, o:o + 8 * B]) for o in range(0, 24 * B, 8 * B)], axis=1)
    assert rms(y - ref) <= 1e-13
    eng.close()


def test_prepare_resets_state_and_unsupported_features_fail_loudly(amd, oracle):
    O = oracle
    eng = amd.BatchedEngine(1, max_ir_len=4096, max_blocks_per_call=4)
    ir = O.gen_ir(4096)
    eng.set_impulse(0, ir, ir)
    x = make_inputs(O, 1, 4 * B)
    y1 = eng.conv_process(x)
    eng.conv_process(x)
    eng.prepare_to_play(48000.0, 4 * B)
    assert np.array_equal(eng.conv_process(x), y1)
    with pytest.raises(amd.CpqError):
        eng.conv_process(x[:, :100])                    # not a multiple of the block size
    p = amd.eq_params_default()
    p.bands[3].channel_mode = 5          # outside Stereo / Left / Right / Mid / Side
    with pytest.raises(amd.CpqError):
        eng.set_eq_params(0, p)
    eng.close()
    long_ir = O.gen_ir(131072)
    # FilterSpec whose tail partition is not a power of two (air-absorption mode with multiplier 6: 3072 samples)
    eng = amd.BatchedEngine(1, max_ir_len=131072, max_blocks_per_call=2)
    with pytest.raises(amd.CpqError) as e4:
        eng.set_impulse(0, long_ir, long_ir, spec=amd.FilterSpec.defaults(tail_mode=0, tail_l1l2_multiplier=6))
    assert e4.value.status == -5
    eng.close()
    with pytest.raises(amd.CpqError):
        amd.BatchedEngine(1, block_size=500, max_ir_len=4096)       # not a power of two


@pytest.mark.parametrize("block,ir_len,blocks_per_call", [(64, 20000, 40), (128, 131072, 64), (256, 131072, 8),
                                                           (256, 131072, 48), (1024, 5000, 4), (2048, 5760, 3), (4096, 5760, 2)])
def test_block_size_sweep_reference_semantics(amd, oracle, block, ir_len, blocks_per_call):
    """BASELINE.json configs[2] block sizes (generic FFT kernels): reference semantics wherever the reference
    itself is LTI (B <= 256 with long IRs: negative layer lags; B >= 1024 with IR inside layer 0)."""
    O = oracle
    S = 1
    irs = [O.gen_ir(ir_len, stream=0, channel=ch) for ch in range(2)]
    total_blocks = blocks_per_call * max(2, (ir_len // block + 200) // blocks_per_call + 1)
    total_blocks = min(total_blocks, 6 * blocks_per_call if block >= 1024 else total_blocks)
    x = make_inputs(O, S, total_blocks * block)
    ref = oracle_conv(O, irs, x, block=block)
    eng = amd.BatchedEngine(S, block_size=block, max_ir_len=ir_len, max_blocks_per_call=blocks_per_call)
    eng.set_impulse(0, irs[0], irs[1])
    assert eng.latency() == block
    step = blocks_per_call * block
    y = np.concatenate([eng.conv_process(x[:, o:o + step]) for o in range(0, x.shape[1], step)], axis=1)
    err = rms(y - ref)
    print(f"B={block} L={ir_len} T={blocks_per_call} rms err {err:.3e} signal rms {rms(ref):.3f}")
    assert err <= 1e-13
    eng.close()


@pytest.mark.parametrize("block", [1024, 2048, 4096])
def test_large_blocks_exact_semantics_long_ir(amd, oracle, block):
    from scipy.signal import fftconvolve
    O = oracle
    ir = [O.gen_ir(131072, channel=ch) for ch in range(2)]
    T = 4
    x = make_inputs(O, 1, 40 * T * block // (block // 512))
    eng = amd.BatchedEngine(1, block_size=block, max_ir_len=131072, max_blocks_per_call=T, semantics=amd.CPQ_SEM_EXACT)
    eng.set_impulse(0, ir[0], ir[1])
    step = T * block
    n = (x.shape[1] // step) * step
    y = np.concatenate([eng.conv_process(x[:, o:o + step]) for o in range(0, n, step)], axis=1)
    for c in range(2):
        assert rms(y[c] - fftconvolve(x[c, :n], ir[c])[:n]) <= 1e-13
    eng.close()


@pytest.mark.parametrize("blk,T", [(128, 7), (127, 7), (480, 1), (441, 3), (441, 17), (1000, 7)])
def test_eq_small_blocks_use_both_kernels(amd, oracle, blk, T):
    """Calls that are no whole number of spans.  Below 1024 samples: k_svf_cascade_tp over the even part (blk 128 x 7 = 896: one
    whole 512-sample span + one whose tail is padding; a 480-sample callback: one padded span), a last odd sample on the
    lane-skewed kernel (127 x 7 = 889) -- all inside the one time-parallel scope.  From 1024 samples up ONE span of 1 ... 7
    waves x 1024 samples whose tail is padding, any count: 441 x 3 = 1323 (odd: the state is taken behind sample 10 of chunk
    82), 441 x 17 = 7497 (two launches: 4096 + 3401), 1000 x 7 = 7000."""
    O = oracle
    S = 2
    x = make_inputs(O, S, 3 * T * blk)
    po = O.eq_params_bench(0.2)
    eng = amd.BatchedEngine(S, block_size=blk, max_ir_len=512, max_blocks_per_call=T,
                            call_mode=amd.CPQ_CALLS_WHOLE_BLOCKS if blk == 128 else amd.CPQ_CALLS_ANY)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    eng.profile_enable(True)
    y = np.concatenate([eng.eq_process(x[:, o:o + T * blk]) for o in range(0, x.shape[1], T * blk)], axis=1)
    prof = eng.profile_read()
    assert prof["k_svf_cascade_tp"][0] == 3 and prof["k_svf_cascade"][0] == 0
    for s in range(S):
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], po, block=blk)
        assert max(np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max()) <= 1e-13
    eng.close()


@pytest.mark.parametrize("mix,peak", [(1.0, 0), (0.35, 777), (0.9995, 100), (0.0, 64)])
def test_processor_level_wrapper_steady_state(amd, oracle, mix, peak):
    """ConvolverProcessor::process steady state around the kernel-level convolver."""
    O = oracle
    S, T = 2, 8
    irs = [O.gen_ir(9000, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    x = make_inputs(O, S, 5 * T * B)
    eng = amd.BatchedEngine(S, max_ir_len=9000, max_blocks_per_call=T)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=mix, ir_peak_latency=peak)
    assert eng.convproc_delay(0) == 512 + peak
    y = np.concatenate([eng.convproc_process(x[:, o:o + T * B]) for o in range(0, x.shape[1], T * B)], axis=1)
    for c in range(2 * S):
        ref = O.convproc_steady(irs[c], x[c], B, mix=mix, ir_peak_latency=peak)
        ok = np.isfinite(ref)
        assert np.array_equal(np.isfinite(y[c]), ok)
        assert rms(y[c][ok] - ref[ok]) <= 1e-13
    if mix == 1.0:
        # finding 5: wetG = equalPowerSin(1) = 1.0000035... != 1
        eng.conv_reset()
        plain = np.concatenate([eng.conv_process(x[:, o:o + T * B]) for o in range(0, x.shape[1], T * B)], axis=1)
        ok = np.isfinite(plain[0])
        ratio = y[0][ok][np.abs(plain[0][ok]) > 1e-3] / plain[0][ok][np.abs(plain[0][ok]) > 1e-3]
        assert abs(ratio.mean() - 1.0000035) < 1e-7
    eng.close()


def test_processor_level_bypass_and_whole_path(amd, oracle):
    O = oracle
    S, T = 1, 4
    irs = [O.gen_ir(3000, stream=0, channel=ch) for ch in range(2)]
    x = make_inputs(O, S, 6 * T * B)
    eng = amd.BatchedEngine(S, max_ir_len=3000, max_blocks_per_call=T)
    eng.set_impulse(0, irs[0], irs[1])
    eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=1.0, bypassed=True, ir_peak_latency=40)
    y = np.concatenate([eng.convproc_process(x[:, o:o + T * B]) for o in range(0, x.shape[1], T * B)], axis=1)
    for c in range(2):
        assert np.array_equal(y[c], O.convproc_steady(irs[c], x[c], B, bypassed=True, ir_peak_latency=40))
    # whole path at processor level: conv wrapper (mix 0.8) then EQ, in place
    eng.prepare_to_play(48000.0, T * B)
    eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=0.8, ir_peak_latency=40)
    eng.set_conv_level(amd.CPQ_LEVEL_PROCESSOR)
    po = O.eq_params_bench(0.2)
    eng.set_eq_params(0, _copy_params(po, amd.eq_params_default()))
    y = np.concatenate([eng.process(x[:, o:o + T * B]) for o in range(0, x.shape[1], T * B)], axis=1)
    w = [O.convproc_steady(irs[c], x[c], B, mix=0.8, ir_peak_latency=40) for c in range(2)]
    rl, rr, _ = O.eq_process_stereo(w[0], w[1], po)
    assert rms(y[0] - rl) <= 1e-13 and rms(y[1] - rr) <= 1e-13
    eng.close()


@pytest.mark.parametrize("mode", ["sequential", "auto"])
@pytest.mark.parametrize("conv_is_last,hc,lc,lp", [(1, 1, 0, 1), (1, 0, 1, 1), (1, 2, 0, 1), (0, 1, 0, 0), (0, 1, 0, 2)])
def test_output_filter_df2t_cascade(amd, oracle, mode, conv_is_last, hc, lc, lp):
    """OutputFilter three-section DF-II-T cascade (stereo FMA path)."""
    O = oracle
    S, T = 3, 12
    x = make_inputs(O, S, 3 * T * B)
    q = O.outfilter_design(conv_is_last, hc, lc, lp, 48000.0)
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
    eng.set_outfilter_params(amd.CPQ_ALL_STREAMS, conv_is_last, hc, lc, lp)
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL if mode == "sequential" else amd.CPQ_EQ_MODE_AUTO)
    y = np.concatenate([eng.outfilter_process(x[:, o:o + T * B]) for o in range(0, x.shape[1], T * B)], axis=1)
    worst, err2 = 0.0, 0.0
    for s in range(S):
        yl, yr, _ = O.outfilter_process_stereo(x[2 * s], x[2 * s + 1], q)
        worst = max(worst, np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max())
        err2 += np.sum((y[2 * s] - yl) ** 2) + np.sum((y[2 * s + 1] - yr) ** 2)
    r = float(np.sqrt(err2 / y.size))
    print("outfilter", mode, (conv_is_last, hc, lc, lp), "max abs diff", worst, "rms", r)
    # sequential kernel: reference op order -> bit equality.  Time-parallel kernel: the 18/20 Hz high-pass has poles
    # at |z| = 0.998, where ANY fp64 evaluation order is ~2e-13 (max) from the exact result (see the CPU test
    # against a long-double evaluation); bar = 1e-12 RMS of BASELINE.json.
    assert worst <= (0.0 if mode == "sequential" else 5e-12) and r <= (0.0 if mode == "sequential" else 1e-12)
    eng.outfilter_reset()
    y2 = eng.outfilter_process(x[:, :T * B])
    assert np.abs(y2 - y[:, :T * B]).max() <= (0.0 if mode == "sequential" else 5e-12)
    eng.close()


def test_whole_chain_conv_eq_output_filter(amd, oracle):
    """DSPCore order Conv -> EQ -> OutputFilter (EQ last: 20 Hz HPF + 2 x LPF), processor-level convolver."""
    O = oracle
    S, T = 2, 8
    irs = [O.gen_ir(7000, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    x = make_inputs(O, S, 4 * T * B)
    po = O.eq_params_bench(0.2)
    q = O.outfilter_design(0, 1, 0, 1, 48000.0)
    eng = amd.BatchedEngine(S, max_ir_len=7000, max_blocks_per_call=T)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=1.0)
    eng.set_conv_level(amd.CPQ_LEVEL_PROCESSOR)
    eng.set_outfilter_params(amd.CPQ_ALL_STREAMS, 0, 1, 0, 1)
    eng.enable_output_filter(True)
    y = np.concatenate([eng.process(x[:, o:o + T * B]) for o in range(0, x.shape[1], T * B)], axis=1)
    for s in range(S):
        w = [O.convproc_steady(irs[2 * s + ch], x[2 * s + ch], B) for ch in range(2)]
        el, er, _ = O.eq_process_stereo(w[0], w[1], po)
        fl, fr, _ = O.outfilter_process_stereo(el, er, q)
        assert rms(y[2 * s] - fl) <= 1e-12 and rms(y[2 * s + 1] - fr) <= 1e-12
    eng.close()


def test_direct_head_is_accepted(amd, oracle):
    """enableDirectHead moves the first <= 32 taps to a time-domain FIR (processDirectBlock); without a FilterSpec the
    sum is the same convolution."""
    O = oracle
    irs = [O.gen_ir(3000, channel=ch) for ch in range(2)]
    x = make_inputs(O, 1, 16 * B)
    ref = np.empty_like(x)
    for c in range(2):
        nuc = O.Nuc()
        assert nuc.set_impulse(irs[c], B, scale=0.7, direct=True)
        ref[c] = nuc.run(x[c], B)
    eng = amd.BatchedEngine(1, max_ir_len=3000, max_blocks_per_call=4)
    eng.set_impulse(0, irs[0], irs[1], scale=0.7, direct_head=True)
    assert eng.plan().direct_taps == 32
    y = np.concatenate([eng.conv_process(x[:, o:o + 4 * B]) for o in range(0, 16 * B, 4 * B)], axis=1)
    assert rms(y - ref) <= 1e-13
    eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=1.0, ir_peak_latency=5)
    assert eng.convproc_delay(0) == 5          # algorithmLatency = 0 with the direct head (Runtime.cpp:266)
    eng.close()


@pytest.mark.parametrize("kw,ir_len", [(dict(), 4096), (dict(hc_mode=0, lc_mode=1), 5000), (dict(hc_mode=2), 3000),
                                       (dict(tail_enabled=0), 40000), (dict(tail_mode=2, hc_mode=0), 131072),
                                       (dict(sample_rate=96000.0, tail_start_seconds=0.2), 9000)])
def test_filter_spec_single_layer(amd, oracle, kw, ir_len):
    """non-NULL FilterSpec whose plan has one layer -- HC/LC spectral gains on every partition
    spectrum (a circular operation inside each 2P frame, finding 5), tail disabled = IR truncated to 32 partitions."""
    O = oracle
    names = {"hc_mode": "hcMode", "lc_mode": "lcMode", "tail_enabled": "tailEnabled", "tail_mode": "tailMode",
             "sample_rate": "sampleRate", "tail_start_seconds": "tailStartSeconds"}
    sa = amd.FilterSpec.defaults(**kw)
    so = O.FilterSpec.defaults(applySpectrumFilter=1, **{names[k]: v for k, v in kw.items()})
    irs = [O.gen_ir(ir_len, channel=ch) for ch in range(2)]
    T = 8
    x = make_inputs(O, 1, 6 * T * B)
    ref = np.empty_like(x)
    for c in range(2):
        nuc = O.Nuc()
        assert nuc.set_impulse(irs[c], B, spec=so)
        assert nuc.plan().numLayers == 1
        ref[c] = nuc.run(x[c], B)
    eng = amd.BatchedEngine(1, max_ir_len=ir_len, max_blocks_per_call=T)
    eng.set_impulse(0, irs[0], irs[1], spec=sa)
    y = np.concatenate([eng.conv_process(x[:, o:o + T * B]) for o in range(0, x.shape[1], T * B)], axis=1)
    err = rms(y - ref)
    print("filterspec", kw, ir_len, "rms err", err, "signal", rms(ref))
    assert err <= 1e-13 and rms(ref) > 1e-3
    eng.close()


@pytest.mark.parametrize("sat", [0.0, 0.2])
def test_eq_parallel_structure(amd, oracle, sat):
    """FilterStructure::Parallel -- out = src + sum_b (band_b(src) - src), in band order."""
    O = oracle
    S, T = 2, 6
    x = make_inputs(O, S, 3 * T * B)
    po = O.eq_params_bench(sat)
    po.filterStructure = 1
    po.bands[4].channelMode = 1
    po.bands[9].channelMode = 2
    po.bands[13].enabled = 0
    po.totalGainDb = 1.5
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    eng.profile_enable(True)
    y = np.concatenate([eng.eq_process(x[:, o:o + T * B]) for o in range(0, x.shape[1], T * B)], axis=1)
    assert eng.profile_read()["k_svf_cascade"][0] == 3
    for s in range(S):
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], po)
        assert np.array_equal(y[2 * s], yl) and np.array_equal(y[2 * s + 1], yr)
    eng.close()


@pytest.mark.parametrize("block,ir_len,blocks_per_call,n_calls,partition", [
    (1024, 131072, 4, 90, 0), (1024, 131072, 32, 10, 0), (2048, 131072, 3, 60, 0), (1024, 524288, 16, 50, 0),
    (1024, 40000, 5, 30, 0), (1024, 131072, 32, 10, 4096), (2048, 131072, 16, 12, -1), (1024, 131072, 8, 40, 2048),
    (1024, 600000, 32, 23, -1)])        # three layers 1024 / 8192 / 65536: the third one starts at tap 530048
def test_time_varying_reference_semantics_large_blocks(amd, oracle, block, ir_len, blocks_per_call, n_calls, partition):
    """BASELINE.json configs[2], B >= 1024 with tail layers: partSize_L > outputDelaySamples_L, the reference's
    delay-line reader drops tail samples (SURVEY A6 'model invalid').  The engine runs one convolution per layer and
    replays the reader; compared with the oracle's stateful Add/Get emulation."""
    O = oracle
    S = 2
    irs = [O.gen_ir(ir_len, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    x = make_inputs(O, S, n_calls * blocks_per_call * block)
    ref = oracle_conv(O, irs, x, block=block)
    assert O.plan(ir_len, block).ltiValid == 0
    # partition: the FFT partition of the per-layer convolutions (0 = the block, -1 = the engine's choice); the reader that
    # makes the plan time-varying runs per callback of `block` on the layer outputs either way
    eng = amd.BatchedEngine(S, block_size=block, max_ir_len=ir_len, max_blocks_per_call=blocks_per_call, partition_size=partition)
    if partition == -1:
        assert eng.partition_size() == 4096
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    step = blocks_per_call * block
    y = np.concatenate([eng.conv_process(x[:, o:o + step]) for o in range(0, x.shape[1], step)], axis=1)
    err = rms(y - ref)
    # the closed-form h_eff must NOT describe this case
    he = O.heff(irs[0], block)
    from scipy.signal import fftconvolve
    lti = fftconvolve(x[0], he)[:x.shape[1]]
    print(f"time-varying B={block} L={ir_len} T={blocks_per_call}: rms err {err:.3e}, signal {rms(ref):.3f}, "
          f"distance of h_eff model {rms(lti - ref[0]):.3e}")
    assert err <= 1e-13 and rms(lti - ref[0]) > 1e-6
    eng.conv_reset()
    y2 = eng.conv_process(x[:, :step])
    assert np.array_equal(y2, y[:, :step])
    eng.close()


@pytest.mark.parametrize("mode,structure", [("sequential", 0), ("auto", 0), ("sequential", 1)])
def test_eq_agc_block_rate(amd, oracle, mode, structure):
    """AGC -- per-callback RMS envelopes (attack 0.2 s / release 2 s), gain smoothing and the
    linear gain ramp, replayed in the reference's accumulation orders."""
    O = oracle
    S, T = 3, 16
    x = make_inputs(O, S, 6 * T * B)
    x[:, 20000:30000] *= 6.0                     # level step so that the gain moves
    x[2:4] *= 0.2
    po = O.eq_params_bench(0.2)
    po.agcEnabled = 1
    po.filterStructure = structure
    po.totalGainDb = -3.0                        # ignored while AGC is on
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
    pa = _copy_params(po, amd.eq_params_default())
    eng.set_eq_params(0, pa)
    eng.set_eq_params(1, pa)
    po_off = O.eq_params_bench(0.2)
    po_off.filterStructure = structure
    eng.set_eq_params(2, _copy_params(po_off, amd.eq_params_default()))      # stream 2 without AGC
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL if mode == "sequential" else amd.CPQ_EQ_MODE_AUTO)
    y = np.concatenate([eng.eq_process(x[:, o:o + T * B]) for o in range(0, x.shape[1], T * B)], axis=1)
    worst = 0.0
    for s in range(S):
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], po if s < 2 else po_off)
        worst = max(worst, np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max())
        if s == 0:
            plain, _, _ = O.eq_process_stereo(x[0], x[1], po_off)
            assert np.abs(yl - plain).max() > 1e-3          # the AGC really changes the signal
    print("agc", mode, structure, "max abs diff", worst)
    assert worst <= (0.0 if mode == "sequential" else 1e-12)
    eng.close()


@pytest.mark.parametrize("mode", ["sequential", "auto"])
def test_eq_total_gain_ramp_on_parameter_change(amd, oracle, mode):
    """A16: a new total gain reaches the output through a 50 ms LinearRamp (2400 samples at 48 kHz) applied with
    applyGainRamp_AVX2's incremental-add pattern; a second change in mid-ramp restarts from the current value."""
    O = oracle
    S, T = 2, 2
    n_calls = 12
    x = make_inputs(O, S, n_calls * T * B)
    gains_db = {0: -2.0, 3: 4.5, 4: -12.0, 9: 0.0}            # change points (call index -> new total gain)
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL if mode == "sequential" else amd.CPQ_EQ_MODE_AUTO)
    po = O.eq_params_bench(0.2)
    state = [np.zeros(168), np.zeros(168)]
    worst = 0.0
    for k in range(n_calls):
        if k in gains_db:
            po.totalGainDb = gains_db[k]
            eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
        seg = x[:, k * T * B:(k + 1) * T * B]
        y = eng.eq_process(seg)
        for s in range(S):
            yl, yr, state[s] = O.eq_process_stereo(seg[2 * s], seg[2 * s + 1], po, state=state[s])
            worst = max(worst, np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max())
    print("gain ramp", mode, "max abs diff", worst)
    assert worst <= (0.0 if mode == "sequential" else 1e-13)
    eng.close()


@pytest.mark.parametrize("structure", [0, 1])
@pytest.mark.parametrize("sat", [0.0, 0.2])
def test_eq_mid_side_channel_modes(amd, oracle, sat, structure):
    """Mid / Side bands.  The reference sends the whole call to the basic process(block)
    (Processing.cpp:1036-1044): M = (L+R)/2, S = (L-R)/2, scalar processBand on one component with its own state,
    L = M+S, R = M-S (:690-739 serial, :792-836 parallel); bands within 0.01 dB of flat are inactive on that path
    (createBandNode, Coefficients.cpp:48-53).  Second stream keeps Stereo/Left/Right bands only (cache path)."""
    O = oracle
    S, T = 2, 5
    x = make_inputs(O, S, 4 * T * B)
    x[1] = 0.7 * x[1] + 0.5 * x[0]                      # correlated L/R so Mid and Side differ in level
    po = [O.eq_params_bench(sat), O.eq_params_bench(sat)]
    for q in po:
        q.filterStructure = structure
        q.totalGainDb = -1.25
        q.bands[2].channelMode = 1
        q.bands[7].channelMode = 2
    for i in (0, 5, 11, 19):
        po[0].bands[i].channelMode = 3
    for i in (3, 8, 14):
        po[0].bands[i].channelMode = 4
    po[0].bands[6].gain = 0.005                          # flat: inactive on the basic path only
    po[1].bands[6].gain = 0.005                          # ... but active (and saturating) on the cache path
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
    for s in range(S):
        eng.set_eq_params(s, _copy_params(po[s], amd.eq_params_default()))
    y = np.concatenate([eng.eq_process(x[:, o:o + T * B]) for o in range(0, x.shape[1], T * B)], axis=1)
    for s in range(S):
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], po[s])
        assert np.array_equal(y[2 * s], yl) and np.array_equal(y[2 * s + 1], yr), (s, np.abs(y[2 * s] - yl).max())
    assert rms(y[0] - x[0]) > 1e-3
    eng.close()


@pytest.mark.parametrize("kw,ir_len,block,T", [(dict(), 131072, 512, 8), (dict(), 131072, 512, 3),
                                               (dict(tail_mode=0, tail_strength=1.6, hc_mode=0, tail_start_seconds=0.1), 131072, 512, 20),
                                               (dict(tail_mode=0, lc_mode=1), 100000, 256, 5),
                                               (dict(hc_mode=2, tail_strength=0.4), 50000, 64, 40),
                                               (dict(), 100000, 256, 16),
                                               # time-varying in the reference (tail partition 4096 > layer-0 length 4080 / 2640):
                                               # the delay-line reader skips blocks; replayed by k_tail_schedule
                                               (dict(tail_mode=0), 131072, 512, 8),
                                               (dict(tail_mode=0, tail_strength=0.7, hc_mode=1), 131072, 512, 3),
                                               (dict(tail_mode=0, tail_start_seconds=0.03), 60000, 512, 5),
                                               # tail partitions above 4096: four-step FFT, permuted spectra
                                               (dict(), 524288, 512, 8),                    # 512 / 4096 / 32768
                                               (dict(hc_mode=0), 131072, 1024, 3),          # 1024 / 8192, time-varying
                                               (dict(tail_mode=0, lc_mode=1), 200000, 2048, 2),     # 2048 / 16384
                                               (dict(), 300000, 256, 32)])                  # 256 / 2048 / 16384
def test_filter_spec_with_tail_layers(amd, oracle, kw, ir_len, block, T):
    """non-NULL FilterSpec on a multi-layer plan.  The HC/LC gains (and, in tail mode 0, the air-absorption
    damping) multiply every partition spectrum at that LAYER's FFT size (NUC.cpp:336-443, :1060-1097), so each tail
    layer runs on the reference's own partition grid (4096 for B = 512) and reaches the output through the delay line
    at done_callback * B.  Checked against the stateful NUC emulation with the same spec."""
    O = oracle
    names = {"hc_mode": "hcMode", "lc_mode": "lcMode", "tail_enabled": "tailEnabled", "tail_mode": "tailMode",
             "sample_rate": "sampleRate", "tail_start_seconds": "tailStartSeconds", "tail_strength": "tailStrength"}
    sa = amd.FilterSpec.defaults(**kw)
    so = O.FilterSpec.defaults(applySpectrumFilter=1, **{names[k]: v for k, v in kw.items()})
    irs = [O.gen_ir(ir_len, channel=ch) for ch in range(2)]
    n_calls = max(4, (ir_len + 40000) // (T * block))
    x = make_inputs(O, 1, n_calls * T * block)
    ref = np.empty_like(x)
    for c in range(2):
        nuc = O.Nuc()
        assert nuc.set_impulse(irs[c], block, spec=so)
        pl = nuc.plan()
        assert pl.numLayers >= 2
        ref[c] = nuc.run(x[c], block)
    eng = amd.BatchedEngine(1, block_size=block, max_ir_len=ir_len, max_blocks_per_call=T)
    eng.set_impulse(0, irs[0], irs[1], spec=sa)
    y = np.concatenate([eng.conv_process(x[:, o:o + T * block]) for o in range(0, x.shape[1], T * block)], axis=1)
    err = rms(y - ref)
    # the plain h_eff convolution (no spectral shaping) must differ visibly: the test would not see a missing filter otherwise
    print("filterspec+tails", kw, ir_len, block, T, "layers", pl.numLayers, "lti", pl.ltiValid, "rms err", err, "signal", rms(ref))
    assert err <= 1e-13 and rms(ref) > 1e-3
    eng.conv_reset()
    y2 = eng.conv_process(x[:, :T * block])
    assert np.array_equal(y2, y[:, :T * block])            # Reset() restores the initial state of every layer
    eng.close()


def test_filter_spec_tails_shared_ir_in_place_whole_path(amd, oracle):
    """FilterSpec tail layers with one shared stereo IR (CPQ_ALL_STREAMS), the device entry point called IN PLACE
    (d_in == d_out: the tail layers must take their input before layer 0 overwrites it) and the EQ behind it."""
    import torch
    O = oracle
    S, T, ir_len = 3, 6, 70000
    sa = amd.FilterSpec.defaults(hc_mode=0)
    so = O.FilterSpec.defaults(applySpectrumFilter=1, hcMode=0)
    irs = [O.gen_ir(ir_len, channel=ch) for ch in range(2)]
    x = make_inputs(O, S, 30 * T * B)
    po = O.eq_params_bench(0.2)
    eng = amd.BatchedEngine(S, max_ir_len=ir_len, max_blocks_per_call=T)
    eng.set_impulse(amd.CPQ_ALL_STREAMS, irs[0], irs[1], spec=sa)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL)
    outs = []
    for o in range(0, x.shape[1], T * B):
        d = torch.from_numpy(np.ascontiguousarray(x[:, o:o + T * B])).cuda()
        eng.process_device(d.data_ptr(), d.data_ptr(), T * B)
        torch.cuda.synchronize()
        outs.append(d.cpu().numpy())
    y = np.concatenate(outs, axis=1)
    for s in range(S):
        w = []
        for ch in range(2):
            nuc = O.Nuc()
            assert nuc.set_impulse(irs[ch], B, spec=so) and nuc.plan().numLayers == 2
            w.append(nuc.run(x[2 * s + ch], B))
        el, er, _ = O.eq_process_stereo(w[0], w[1], po)
        assert rms(y[2 * s] - el) <= 1e-13 and rms(y[2 * s + 1] - er) <= 1e-13, (s, rms(y[2 * s] - el))
    eng.close()


@pytest.mark.parametrize("ir_len,block,T,S", [(131072, 512, 8, 2), (131072, 512, 64, 2), (524288, 512, 16, 1),
                                              (131072, 1024, 4, 1), (20000, 128, 12, 2), (3000, 512, 4, 1),
                                              # three layers above B = 512: 1024 / 8192 / 65536 and 2048 / 16384 / 131072
                                              (600000, 1024, 32, 1), (1150000, 2048, 32, 1)])
def test_native_non_uniform_schedule(amd, oracle, ir_len, block, T, S):
    """CPQ_SCHED_REFERENCE_NUC (BASELINE.json configs[3]): the reference's own non-uniform partition schedule run
    natively -- layer 0 at the block size, tail layers at 8x / 64x on their own FFT grids, merged through the replayed
    delay-line reader -- against the stateful NUC emulation (filterSpec = nullptr), LTI and time-varying plans, and
    against the uniform-schedule engine (same reference semantics, different partitioning) where that one applies."""
    O = oracle
    irs = [O.gen_ir(ir_len, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    n_calls = max(3, (ir_len + 30000) // (T * block))
    x = make_inputs(O, S, n_calls * T * block)
    ref = np.empty_like(x)
    for c in range(2 * S):
        nuc = O.Nuc()
        assert nuc.set_impulse(irs[c], block)
        ref[c] = nuc.run(x[c], block)
    eng = amd.BatchedEngine(S, block_size=block, max_ir_len=ir_len, max_blocks_per_call=T,
                            schedule=amd.CPQ_SCHED_REFERENCE_NUC)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    y = np.concatenate([eng.conv_process(x[:, o:o + T * block]) for o in range(0, x.shape[1], T * block)], axis=1)
    err = rms(y - ref)
    print("native nuc", ir_len, block, T, "layers", eng.plan().num_layers, "rms err", err, "signal", rms(ref))
    assert err <= 1e-13 and rms(ref) > 1e-3
    eng.close()


@pytest.mark.parametrize("ir_len,block", [(4096, 512), (8192, 512), (16384, 512), (8192, 1024), (4096, 256)] +
                         [(n, 512) for n in (1024, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193)])
def test_reference_mt_nupc_measurement_scenarios(amd, oracle, ir_len, block):
    """The scenarios of the reference's own NUC test (src/tests/MT-NUPC-Measurement.cpp:116-122 MT-NUPC-01 configs,
    :196-199 MT-NUPC-03 partition-boundary lengths; IR = sign(sin(0.1 i)) :49-51, Dirac in the first block then
    silence, one block per Add/Get, 2 x irLen output samples :73-85).  The reference asserts SetImpulse success and
    output energy > 1e-20 (:183); here additionally the whole response must equal the stateful emulation's."""
    O = oracle
    i = np.arange(ir_len)
    ir = np.where(np.sin(i * 0.1) > 0.0, 1.0, -1.0)
    total = ((ir_len * 2 + block - 1) // block) * block
    x = np.zeros((2, total))
    x[:, 0] = 1.0
    eng = amd.BatchedEngine(1, block_size=block, max_ir_len=ir_len, max_blocks_per_call=1)
    eng.set_impulse(0, ir, ir)                                # SetImpulse(ir, irLen, blockSize, 1.0, false, nullptr)
    assert eng.is_ready() and eng.latency() == max(block, 64)
    y = np.concatenate([eng.conv_process(x[:, o:o + block]) for o in range(0, total, block)], axis=1)
    assert float(np.sum(y[0] ** 2)) > 1e-20                   # the reference's own assertion
    nuc = O.Nuc()
    assert nuc.set_impulse(ir, block)
    ref = nuc.run(x[0], block)
    assert np.abs(y[0] - ref).max() <= 1e-12 and np.array_equal(y[0], y[1])
    eng.close()


def test_reference_mt_nupc_02_dirac_response(amd, oracle):
    """MT-NUPC-02 (src/tests/MT-NUPC-Measurement.cpp:143-190): IR = sin(0.5 i), 8192 taps, blk 512, Dirac then
    Add(nullptr) = silence for 16384 output samples; assertion: energy > 1e-20."""
    O = oracle
    ir = np.sin(np.arange(8192) * 0.5)
    x = np.zeros((2, 16384))
    x[:, 0] = 1.0
    eng = amd.BatchedEngine(1, block_size=512, max_ir_len=8192, max_blocks_per_call=1)
    eng.set_impulse(0, ir, ir)
    y = np.concatenate([eng.conv_process(x[:, o:o + 512]) for o in range(0, 16384, 512)], axis=1)
    energy = float(np.sum(y[0] ** 2))
    assert energy > 1e-20
    nuc = O.Nuc()
    assert nuc.set_impulse(ir, 512)
    assert np.abs(y[0] - nuc.run(x[0], 512)).max() <= 1e-12
    eng.close()


@pytest.mark.parametrize("schedule", ["uniform", "nuc"])
def test_ragged_call_lengths_across_ring_wraps(amd, oracle, schedule):
    """Calls of varying length (1 ... T_max blocks, every MAC kernel variant) for long enough that the FDL ring
    (512 slots at 131072 taps), the tail-layer rings and the delay lines wrap several times."""
    O = oracle
    rng = np.random.default_rng(7)
    T_max, ir_len = 112, 131072
    sizes = []
    while sum(sizes) < 1500:
        sizes.append(int(rng.choice([1, 2, 3, 5, 8, 13, 31, 40, 47, 48, 64, 100, 112])))
    total = sum(sizes) * B
    irs = [O.gen_ir(ir_len, channel=ch) for ch in range(2)]
    x = make_inputs(O, 1, total)
    ref = np.empty_like(x)
    for c in range(2):
        nuc = O.Nuc()
        assert nuc.set_impulse(irs[c], B)
        ref[c] = nuc.run(x[c], B)
    eng = amd.BatchedEngine(1, max_ir_len=ir_len, max_blocks_per_call=T_max,
                            schedule=amd.CPQ_SCHED_REFERENCE_NUC if schedule == "nuc" else amd.CPQ_SCHED_UNIFORM)
    eng.set_impulse(0, irs[0], irs[1])
    outs, o = [], 0
    for t in sizes:
        outs.append(eng.conv_process(x[:, o:o + t * B]))
        o += t * B
    y = np.concatenate(outs, axis=1)
    err = rms(y - ref)
    tail_err = rms(y[:, -100 * B:] - ref[:, -100 * B:])
    print("ragged", schedule, len(sizes), "calls", sum(sizes), "blocks, rms err", err, "last 100 blocks", tail_err)
    assert err <= 1e-13 and tail_err <= 1e-13
    eng.close()


@pytest.mark.parametrize("sat", [0.0, 0.2])
def test_eq_time_parallel_kernels_hand_over_state(amd, oracle, sat):
    """One call of 59 blocks = three 8192-sample spans on the eight-wave vector-form kernel (k_svf_cascade_tpv<8>; with 4
    channels they run as chained spans: three workgroups per channel, band states handed over inside the launch) + one span
    of 5632 samples on six waves whose tail is padding (k_svf_cascade_tpv<0, false, PARTIAL>); band states pass between the
    two kernels and from call to call through the state array.  The second call carries an Inf in its third span, the third
    a NaN in its second span (guarded path in staged pieces of 2048 samples)."""
    O = oracle
    S, T = 2, 59
    x = make_inputs(O, S, 3 * T * B)
    x[1, 2 * T * B + 8192 + 5000] = np.nan
    x[2, T * B + 16384 + 3000] = np.inf
    po = O.eq_params_bench(sat)
    po.bands[6].channelMode = 1
    po.totalGainDb = 0.75
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    eng.profile_enable(True)
    y = np.concatenate([eng.eq_process(x[:, o:o + T * B]) for o in range(0, x.shape[1], T * B)], axis=1)
    assert eng.profile_read()["k_svf_cascade_tp"][0] == 3 and eng.profile_read()["k_svf_cascade"][0] == 0
    worst = 0.0
    for s in range(S):
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], po)
        worst = max(worst, np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max())
    print("tp hand-over sat", sat, "max abs diff", worst)
    assert np.all(np.isfinite(y)) and worst <= 1e-13
    eng.close()


@pytest.mark.parametrize("sat", [0.0, 0.2])
def test_eq_short_calls_one_span_of_one_to_seven_waves(amd, oracle, sat):
    """Calls of 2 ... 15 blocks (and 23 = one 8192-sample span + 7 blocks): the remainder after the whole 8192-sample spans
    runs as ONE span of 1 ... 7 waves x 1024 samples in matrix form, an odd block count leaves one 512-sample span in VALU
    form.  A NaN in a 7-block call and an Inf in an 11-block call send those spans through the guarded path (pieces of
    3072 and of 4096 + 1024 samples)."""
    O = oracle
    S = 2
    calls = [2, 3, 5, 7, 9, 11, 15, 1, 23, 4, 6, 7, 11, 2]
    x = make_inputs(O, S, sum(calls) * B)
    starts = np.cumsum([0] + calls) * B
    x[0, starts[11] + 2500] = np.nan              # the second 7-block call
    x[3, starts[12] + 4500] = np.inf              # the second 11-block call
    po = O.eq_params_bench(sat)
    po.bands[3].channelMode = 2
    po.bands[9].type = 3
    po.totalGainDb = -1.25
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=max(calls))
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    y = np.concatenate([eng.eq_process(x[:, starts[i]:starts[i + 1]]) for i in range(len(calls))], axis=1)
    worst = 0.0
    for s in range(S):
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], po)
        worst = max(worst, np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max())
    print("short calls sat", sat, "max abs diff", worst)
    assert np.all(np.isfinite(y)) and worst <= 1e-13
    eng.close()


@pytest.mark.parametrize("kw,ir_len,block,T,schedule", [(dict(), 3000, 512, 4, "uniform"), (dict(hc_mode=0), 131072, 512, 8, "uniform"),
                                                       (dict(tail_mode=0), 100000, 256, 12, "uniform"),
                                                       (None, 131072, 512, 16, "nuc"), (None, 20, 64, 3, "uniform"),
                                                       (None, 131072, 1024, 4, "uniform")])
def test_direct_head_time_domain_path(amd, oracle, kw, ir_len, block, T, schedule):
    """The direct head as the reference runs it (src/MKLNonUniformConvolver.cpp:689-731, 1169-1232): the first
    min(irLen, 32) taps leave the FFT path BEFORE the partition spectra and any FilterSpec gains are formed, and run as a
    time-domain FIR over [history | block] whose output is flushed below 1e-20.  With a FilterSpec the head is therefore
    NOT spectrally shaped -- the case the 'taps stay in the FFT path' shortcut could not represent.  kw = None: no spec
    (incl. the native non-uniform schedule, an IR shorter than 32 taps, and a time-varying plan at block 1024)."""
    O = oracle
    names = {"hc_mode": "hcMode", "lc_mode": "lcMode", "tail_mode": "tailMode"}
    sa = amd.FilterSpec.defaults(**kw) if kw is not None else None
    so = O.FilterSpec.defaults(applySpectrumFilter=1, **{names[k]: v for k, v in kw.items()}) if kw is not None else None
    irs = [O.gen_ir(ir_len, channel=ch) for ch in range(2)]
    n_calls = max(4, (ir_len + 20000) // (T * block))
    x = make_inputs(O, 1, n_calls * T * block)
    ref = np.empty_like(x)
    for c in range(2):
        nuc = O.Nuc()
        assert nuc.set_impulse(irs[c], block, scale=0.9, direct=True, spec=so)
        assert nuc.plan().directTaps == min(ir_len, 32)
        ref[c] = nuc.run(x[c], block)
    eng = amd.BatchedEngine(1, block_size=block, max_ir_len=ir_len, max_blocks_per_call=T,
                            schedule=amd.CPQ_SCHED_REFERENCE_NUC if schedule == "nuc" else amd.CPQ_SCHED_UNIFORM)
    eng.set_impulse(0, irs[0], irs[1], scale=0.9, direct_head=True, spec=sa)
    y = np.concatenate([eng.conv_process(x[:, o:o + T * block]) for o in range(0, x.shape[1], T * block)], axis=1)
    err = rms(y - ref)
    print("direct head", kw, ir_len, block, T, schedule, "rms err", err, "signal", rms(ref))
    assert err <= 1e-13 and rms(ref) > 1e-4
    # a later IR without the direct head switches the FIR off again for that stream
    eng.conv_reset()
    eng.set_impulse(0, irs[0], irs[1], scale=0.9, spec=sa)
    nuc = O.Nuc()
    assert nuc.set_impulse(irs[0], block, scale=0.9, spec=so)
    y2 = eng.conv_process(x[:, :T * block])
    assert rms(y2[0] - nuc.run(x[0, :T * block], block)) <= 1e-13
    eng.close()


def test_prepare_with_a_new_sample_rate_redesigns_the_filters(amd, oracle):
    """EQProcessor::prepareToPlay / OutputFilter::prepare rebuild their coefficients when the rate changes
    (src/eqprocessor/EQProcessor.Core.cpp:679-826): parameters set at 48 kHz, then prepare(96 kHz)."""
    O = oracle
    S, T = 2, 16
    x = make_inputs(O, S, T * B)
    po = [O.eq_params_bench(0.2), O.eq_params_bench(0.0)]
    po[1].bands[3].gain = 5.5
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
    for s in range(S):
        eng.set_eq_params(s, _copy_params(po[s], amd.eq_params_default()))
    eng.set_outfilter_params(amd.CPQ_ALL_STREAMS, 0, 1, 0, 1)
    eng.prepare_to_play(96000.0, T * B)
    y = eng.eq_process(x)
    z = eng.outfilter_process(x)
    q = O.outfilter_design(0, 1, 0, 1, 96000.0)
    for s in range(S):
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], po[s], sr=96000.0)
        assert max(np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max()) <= 1e-13
        fl, fr, _ = O.outfilter_process_stereo(x[2 * s], x[2 * s + 1], q)
        assert rms(z[2 * s] - fl) <= 1e-12 and rms(z[2 * s + 1] - fr) <= 1e-12
    eng.close()


def test_host_pointer_path_pipelines_pinned_buffers(amd, oracle):
    """cpq_engine_process_block on PINNED caller buffers (cpq_host_register) runs upload / kernels / download pipelined
    over four time chunks; the result must equal the plain single-shot path used for pageable buffers (in place too)."""
    O = oracle
    S, T = 3, 64
    irs = [O.gen_ir(9000, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    x = make_inputs(O, S, 2 * T * B)
    po = O.eq_params_bench(0.2)

    def run(pin):
        eng = amd.BatchedEngine(S, max_ir_len=9000, max_blocks_per_call=T)
        for s in range(S):
            eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
        eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
        outs = []
        for o in range(0, x.shape[1], T * B):
            buf = np.ascontiguousarray(x[:, o:o + T * B])
            if pin:
                assert eng._lib.cpq_host_register(buf.ctypes.data, buf.nbytes) == 0
            rc = eng._lib.cpq_engine_process_block(eng._h, buf.ctypes.data_as(amd._capi.c_double_p),
                                                   buf.ctypes.data_as(amd._capi.c_double_p), T * B)      # in place
            assert rc == 0
            if pin:
                assert eng._lib.cpq_host_unregister(buf.ctypes.data) == 0
            outs.append(buf)
        eng.close()
        return np.concatenate(outs, axis=1)

    y_pageable, y_pinned = run(False), run(True)
    assert np.abs(y_pinned - y_pageable).max() <= 1e-13
    ref = np.empty_like(x)
    for s in range(S):
        w = []
        for ch in range(2):
            nuc = O.Nuc()
            assert nuc.set_impulse(irs[2 * s + ch], B)
            w.append(nuc.run(x[2 * s + ch], B))
        ref[2 * s], ref[2 * s + 1], _ = O.eq_process_stereo(w[0], w[1], po)
    assert rms(y_pinned - ref) <= 1e-13


def test_processor_level_mix_ramp(amd, oracle):
    """SURVEY N1 transitions: a mix change after processing has started is smoothed by the LinearRamp mixSmoother
    (4800 samples at 48 kHz): per-sample equal-power gains for every callback that starts while it runs, a retarget in
    the middle of a ramp keeps the remaining step count, a different ramp time on the second stream, and a final change
    to dry-only lets the convolver run until the ramp has finished and then copies the delayed dry signal."""
    O = oracle
    S, T = 2, 6
    irs = [O.gen_ir(6000, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    calls = 12
    x = make_inputs(O, S, calls * T * B)
    # mix per call and stream (a call = 6 callbacks); stream 1 uses a 30 ms ramp
    sched = [[1.0, 1.0, 0.35, 0.8, 0.8, 0.8, 0.2, 0.2, 0.9, 0.9, 0.9, 0.9],
             [0.5, 0.5, 0.5, 0.95, 0.95, 0.1, 0.1, 0.1, 0.1, 0.6, 0.6, 0.6]]
    times = [0.0, 0.03]
    eng = amd.BatchedEngine(S, max_ir_len=6000, max_blocks_per_call=T)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    outs = []
    for k in range(calls):
        for s in range(S):
            eng.set_convproc_params(s, mix=sched[s][k], smoothing_time_sec=times[s])
        outs.append(eng.convproc_process(x[:, k * T * B:(k + 1) * T * B]))
    y = np.concatenate(outs, axis=1)
    for s in range(S):
        per_cb = [m for m in sched[s] for _ in range(T)]
        for ch in range(2):
            ref = O.convproc_mix_schedule(irs[2 * s + ch], x[2 * s + ch], B, per_cb, smoothing_time=times[s] or 0.1)
            err = np.abs(y[2 * s + ch] - ref).max()
            assert err <= 1e-13, (s, ch, err)
    eng.close()
    # all streams to dry-only: the ramp still needs the wet signal, afterwards the output is the delayed input
    eng = amd.BatchedEngine(1, max_ir_len=6000, max_blocks_per_call=T)
    eng.set_impulse(0, irs[0], irs[1])
    mixes = [0.7, 0.7, 0.0, 0.0, 0.0, 0.0]
    outs = []
    for k, m in enumerate(mixes):
        eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=m)
        outs.append(eng.convproc_process(x[:2, k * T * B:(k + 1) * T * B]))
    y = np.concatenate(outs, axis=1)
    per_cb = [m for m in mixes for _ in range(T)]
    ref = O.convproc_mix_schedule(irs[0], x[0, :len(per_cb) * B], B, per_cb)
    assert np.abs(y[0] - ref).max() <= 1e-13
    assert np.array_equal(y[0, 5 * T * B:], ref[5 * T * B:])           # plain copy of the delayed dry signal
    eng.close()


def test_ingested_ir_file_end_to_end(amd, oracle):
    """SURVEY N3 + N1: the reference's own sample IR file through cpq_ir_load_wav -> cpq_ir_prepare (DC blocker, window,
    1 s target length, scale factor, peak latency) -> set_impulse(scale, FilterSpec) -> processor-level process, against
    the oracle run on the same prepared IR (the ingest itself is compared with its numpy restatement in
    tests/test_ir_ingest_cpu.py)."""
    O = oracle
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "impulse_room_correction_hpf_lpf.wav")
    ir, rate = amd.ir_load_wav(path)
    prep = amd.ir_prepare(ir, rate, 48000.0, 1.0)
    h, sf, lat = prep["ir"], prep["scale"]["scale_factor"], prep["ir_peak_latency"]
    assert h.shape == (2, 48000) and 0.0 < sf < 10.0 and 0 <= lat < 48000
    S, T = 2, 8
    n = 6 * T * B
    x = make_inputs(O, S, n)
    eng = amd.BatchedEngine(S, max_ir_len=48000, max_blocks_per_call=T)
    spec_a, spec_o = amd.FilterSpec.defaults(), O.FilterSpec.defaults(applySpectrumFilter=1)
    eng.set_impulse(amd.CPQ_ALL_STREAMS, h[0], h[1], scale=sf, spec=spec_a)
    eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=0.8, ir_peak_latency=lat)
    assert eng.convproc_delay(0) == B + lat
    y = np.concatenate([eng.convproc_process(x[:, k * T * B:(k + 1) * T * B]) for k in range(6)], axis=1)
    worst = 0.0
    for c in range(2 * S):
        ref = O.convproc_steady(h[c % 2], x[c], B, mix=0.8, ir_peak_latency=lat, scale=sf, spec=spec_o)
        worst = max(worst, rms(y[c] - ref) / max(rms(ref), 1e-30))
    assert worst <= 1e-12, worst
    eng.close()


def test_long_ingested_ir_three_layers_end_to_end(amd, oracle):
    """The reference's 20 s sample IR (sampledata/synthetic_long_ir_20s.wav, fixture tests/golden/) through cpq_ir_load_wav ->
    cpq_ir_prepare (10 s target: 480000 taps) -> set_impulse(scale factor) at 512-sample blocks -- the reference's
    three-layer plan (512 / 4096 / 32768, src/MKLNonUniformConvolver.cpp:738-758) with its tail gains and lags -> kernel-level
    process of 300000 samples (past the start of the third layer), against the oracle's stateful emulation on the same
    prepared IR."""
    O = oracle
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "synthetic_long_ir_20s.wav")
    ir, rate = amd.ir_load_wav(path)
    prep = amd.ir_prepare(ir, rate, 48000.0, 10.0)
    h, sf = prep["ir"], prep["scale"]["scale_factor"]
    assert h.shape == (2, 480000) and 0.0 < sf < 10.0
    plan = amd.nuc_plan(480000, B)
    assert plan.num_layers == 3 and [plan.part_size[l] for l in range(3)] == [512, 4096, 32768]
    S, T = 1, 64
    calls = 10                                           # 327680 samples
    x = make_inputs(O, S, calls * T * B)
    eng = amd.BatchedEngine(S, max_ir_len=480000, max_blocks_per_call=T)
    eng.set_impulse(0, h[0], h[1], scale=sf)
    y = np.concatenate([eng.conv_process(x[:, k * T * B:(k + 1) * T * B]) for k in range(calls)], axis=1)
    eng.close()
    for c in range(2):
        nuc = O.Nuc()
        assert nuc.set_impulse(h[c], B, scale=sf)
        ref = nuc.run(x[c], B)
        nuc.close()
        assert rms(y[c] - ref) <= 1e-12 * max(1.0, rms(ref)), (c, rms(y[c] - ref), rms(ref))
        assert rms(ref[300000:]) > 0.0


@pytest.mark.parametrize("block,agc,eq_mode", [(512, False, "seq"), (512, False, "auto"), (64, False, "seq"), (512, True, "seq")])
def test_eq_bypass_fade_state_machine(amd, oracle, block, agc, eq_mode):
    """SURVEY N4 (bypass fades): cpq_eq_set_bypass per stream, changed between calls while processing -- the 5 ms fade
    through the basic path (flat bands drop out of the cascade there, visible with saturation 0.2), the frozen state
    while bypassed, the state clear + fade-in on release, a release in the middle of a fade-out (block 64: the fade spans
    four callbacks), a moving total gain and the AGC across the bypass.  Against the state machine restated in
    tests/oracle_lib.py (EqWithBypass) around the oracle's EQ."""
    O = oracle
    S, T, calls = 3, 6, 10
    po = O.eq_params_bench(0.2)
    for i in (3, 8, 12):
        po.bands[i].gain = 0.0                      # flat bands: active on the parameter path, inactive as band nodes
    po.totalGainDb = -3.0
    po.agcEnabled = int(agc)
    n = T * block
    x = make_inputs(O, S, calls * n)
    # bypass request per call and stream; stream 2 is never bypassed
    req = [[0, 0, 1, 1, 1, 0, 0, 1, 0, 0],
           [0, 1, 0, 1, 1, 1, 0, 0, 0, 1],
           [0] * 10]
    eng = amd.BatchedEngine(S, block_size=block, max_ir_len=block, max_blocks_per_call=T)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    if eq_mode == "seq":
        eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL)
    outs = []
    for k in range(calls):
        for s in range(S):
            eng.set_eq_bypass(s, req[s][k])
        if k == 6:                                  # a total-gain change while stream 0 has just been released
            pa = _copy_params(po, amd.eq_params_default())
            pa.total_gain_db = 2.0
            eng.set_eq_params(0, pa)
        outs.append(eng.eq_process(x[:, k * n:(k + 1) * n]))
    y = np.concatenate(outs, axis=1)
    eng.close()
    tol = 0.0 if eq_mode == "seq" else 2e-14
    for s in range(S):
        ps = O.EqParams.from_buffer_copy(po)
        ref = O.EqWithBypass(ps, 48000.0, block)
        rl, rr = [], []
        for k in range(calls):
            if k == 6 and s == 0:
                ref.p.totalGainDb = 2.0
            for t in range(T):
                o = (k * T + t) * block
                a, b = ref.callback(x[2 * s, o:o + block].copy(), x[2 * s + 1, o:o + block].copy(), bool(req[s][k]))
                rl.append(a)
                rr.append(b)
        rl, rr = np.concatenate(rl), np.concatenate(rr)
        err = max(np.abs(y[2 * s] - rl).max(), np.abs(y[2 * s + 1] - rr).max())
        assert err <= tol, (s, err)
    # while bypassed (fade complete) the output is the input, bit for bit
    k = 3
    assert np.array_equal(y[0, k * n + block:(k + 1) * n], x[0, k * n + block:(k + 1) * n])
    assert not np.array_equal(y[4, k * n:(k + 1) * n], x[4, k * n:(k + 1) * n])


@pytest.mark.parametrize("block,T", [(512, 1), (64, 4), (512, 2)])       # the 240-sample fade ends in the LAST callback of a call in the first two
def test_eq_bypass_request_set_once_then_left_alone(amd, oracle, block, T):
    """A host sets the bypass request when it changes, not before every block: after the release fade has ended inside a
    call, the following calls -- with no cpq_eq_set_bypass in between -- must be back on the parameter path (a flat band
    is active there and applies its saturation stage; on the basic path's band nodes, used while fading, it is not)."""
    O = oracle
    po = O.eq_params_bench(0.2)
    for i in (3, 8, 12):
        po.bands[i].gain = 0.0
    n, calls = T * block, 9
    x = make_inputs(O, 1, calls * n)
    eng = amd.BatchedEngine(1, block_size=block, max_ir_len=block, max_blocks_per_call=T)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL)
    changes = {2: True, 5: False}                      # call index -> new request; nothing is set at the other calls
    req, outs, reqs = False, [], []
    for k in range(calls):
        if k in changes:
            req = changes[k]
            eng.set_eq_bypass(0, req)
        if k == 5:                                     # new parameters arrive with the release (they put the stream's
            pa = _copy_params(po, amd.eq_params_default())     # tables back on the parameter path, the fade moves them again)
            pa.total_gain_db = 1.5
            eng.set_eq_params(0, pa)
        reqs.append(req)
        outs.append(eng.eq_process(x[:, k * n:(k + 1) * n]))
    y = np.concatenate(outs, axis=1)
    eng.close()
    ref = O.EqWithBypass(O.EqParams.from_buffer_copy(po), 48000.0, block)
    rl, rr = [], []
    for k in range(calls):
        if k == 5:
            ref.set_total_gain_db(1.5)
        for t in range(T):
            o = (k * T + t) * block
            a, b = ref.callback(x[0, o:o + block].copy(), x[1, o:o + block].copy(), reqs[k])
            rl.append(a)
            rr.append(b)
    assert np.array_equal(y[0], np.concatenate(rl)) and np.array_equal(y[1], np.concatenate(rr))


@pytest.mark.parametrize("direct", [False, True])
@pytest.mark.parametrize("any_calls", [False, True, "nuc"])
def test_ir_reload_on_a_live_stream_starts_from_silence(amd, oracle, direct, any_calls):
    """SetImpulse leaves a convolver that has seen no input (every buffer allocated anew and zeroed,
    src/MKLNonUniformConvolver.cpp:697-714): a stream given a new IR in mid-run plays like a new NUC from that call on --
    no old input through the new IR's frequency-domain delay line or its direct head -- while its neighbour plays on."""
    O = oracle
    S, T, calls, reload_at = 2, 4, 8, 3
    n = T * B
    irs = [O.gen_ir(9000, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    new = [O.gen_ir(5000, stream=9, channel=ch) for ch in range(2)]
    x = make_inputs(O, S, calls * n)
    eng = amd.BatchedEngine(S, max_ir_len=9000, max_blocks_per_call=T,
                            call_mode=amd.CPQ_CALLS_ANY if any_calls is True else amd.CPQ_CALLS_WHOLE_BLOCKS,
                            schedule=amd.CPQ_SCHED_REFERENCE_NUC if any_calls == "nuc" else amd.CPQ_SCHED_UNIFORM)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1], direct_head=direct)
    outs = []
    for k in range(calls):
        if k == reload_at:
            eng.set_impulse(1, new[0], new[1], direct_head=direct)
        outs.append(eng.conv_process(x[:, k * n:(k + 1) * n]))
    y = np.concatenate(outs, axis=1)
    eng.close()

    def nuc_run(ir, sig):
        nuc = O.Nuc()
        assert nuc.set_impulse(ir, B, direct=direct)
        out = nuc.run(sig, B)
        nuc.close()
        return out
    cut = reload_at * n
    for ch in range(2):
        assert rms(y[ch] - nuc_run(irs[ch], x[ch])) <= 1e-13                          # stream 0 does not notice
        assert rms(y[2 + ch][:cut] - nuc_run(irs[2 + ch], x[2 + ch][:cut])) <= 1e-13
        assert rms(y[2 + ch][cut:] - nuc_run(new[ch], x[2 + ch][cut:])) <= 1e-13      # a new NUC from the reload on


@pytest.mark.parametrize("mode", ["sequential", "auto"])
def test_eq_band_parameters_change_on_a_live_stream(amd, oracle, mode):
    """New band parameters (type, frequency, gain, Q, channel mode, saturation) between calls: the coefficient cache is
    swapped, the filter states run on (src/eqprocessor/EQProcessor.ProcessingCache.cpp:71-90 builds a new cache, filterState
    is untouched) -- per stream, while the neighbour keeps its parameters."""
    O = oracle
    S, T, calls = 2, 17, 6                      # 17 blocks: an 8192-sample span + a one-wave span of the matrix form
    n = T * B
    x = make_inputs(O, S, calls * n)
    rng = np.random.default_rng(5)
    po = [O.eq_params_bench(0.2), O.eq_params_bench(0.2)]
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL if mode == "sequential" else amd.CPQ_EQ_MODE_AUTO)
    for s in range(S):
        eng.set_eq_params(s, _copy_params(po[s], amd.eq_params_default()))
    state = [np.zeros(168), np.zeros(168)]
    worst = 0.0
    for k in range(calls):
        if k in (2, 3, 5):                       # stream 1 gets new bands; stream 0 never does
            for i in rng.choice(20, size=6, replace=False):
                b = po[1].bands[int(i)]
                b.gain = float(rng.uniform(-9, 9))
                b.frequency = float(np.clip(b.frequency * rng.uniform(0.5, 2.0), 20.0, 20000.0))
                b.q = float(rng.uniform(0.3, 6.0))
                b.type = int(rng.choice([0, 1, 2, 3, 4])) if rng.random() < 0.3 else b.type
                b.channelMode = int(rng.choice([0, 1, 2])) if rng.random() < 0.3 else b.channelMode
            po[1].nonlinearSaturation = float(rng.choice([0.0, 0.2, 0.6]))
            eng.set_eq_params(1, _copy_params(po[1], amd.eq_params_default()))
        seg = x[:, k * n:(k + 1) * n]
        y = eng.eq_process(seg)
        for s in range(S):
            yl, yr, state[s] = O.eq_process_stereo(seg[2 * s], seg[2 * s + 1], po[s], state=state[s])
            worst = max(worst, np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max())
    print("live EQ parameter change", mode, "max abs diff", worst)
    assert worst <= (0.0 if mode == "sequential" else 1e-12)
    eng.close()


def test_dspcore_routing_gains_and_bypasses(amd, oracle):
    """The rest of DSPCore's block routing (DSPCoreDouble.cpp:384-470): EQ -> conv order with convolverInputTrimGain,
    outputMakeupGain after the output filter, convBypassed (the convolver stage is skipped, state untouched), and with
    the convolver and a stream's EQ both bypassed the output filter is not run for that stream either."""
    O = oracle
    S, T = 2, 4
    n = T * B
    irs = [O.gen_ir(3000, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    x = make_inputs(O, S, 6 * n)
    po = O.eq_params_bench(0.2)
    q = O.outfilter_design(1, 1, 0, 1, 48000.0)
    trim, makeup = [0.5, 1.0 + 1e-13], [1.25, 0.8]            # stream 1's trim is within 1e-12 of 1: not applied
    eng = amd.BatchedEngine(S, max_ir_len=3000, max_blocks_per_call=T)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
        eng.set_gains(s, trim[s], makeup[s])
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL)
    eng.set_order(amd.CPQ_ORDER_EQ_THEN_CONV)
    eng.set_outfilter_params(amd.CPQ_ALL_STREAMS, 1, 1, 0, 1)
    eng.enable_output_filter(True)
    # calls 0-1: everything active; 2-3: convolver bypassed; 4: also stream 0's EQ bypass requested before the call
    # (its fade callback still filters, the output filter does not run for it); 5: all back
    outs = []
    for k in range(6):
        eng.set_conv_bypass(k in (2, 3, 4))
        eng.set_eq_bypass(0, k == 4)
        outs.append(eng.process(x[:, k * n:(k + 1) * n]))
    y = np.concatenate(outs, axis=1)
    eng.close()
    for s in range(S):
        eq = O.EqWithBypass(O.EqParams.from_buffer_copy(po), 48000.0, B)
        nucs = [O.Nuc(), O.Nuc()]
        for ch in range(2):
            assert nucs[ch].set_impulse(irs[2 * s + ch], B)
        of_state = None
        ref = np.empty((2, 6 * n))
        for k in range(6):
            conv_bypassed, eq_req = k in (2, 3, 4), (k == 4 and s == 0)
            blk = [np.empty(n), np.empty(n)]
            for t in range(T):
                o = k * n + t * B
                a, b = eq.callback(x[2 * s, o:o + B].copy(), x[2 * s + 1, o:o + B].copy(), eq_req)
                blk[0][t * B:(t + 1) * B], blk[1][t * B:(t + 1) * B] = a, b
            if not conv_bypassed:
                for ch in range(2):
                    v = blk[ch] * trim[s] if abs(trim[s] - 1.0) > 1e-12 else blk[ch]
                    blk[ch] = nucs[ch].run(v, B)
            if not (conv_bypassed and eq_req):
                blk[0], blk[1], of_state = O.outfilter_process_stereo(blk[0], blk[1], q, of_state)
            ref[0, k * n:(k + 1) * n], ref[1, k * n:(k + 1) * n] = blk[0] * makeup[s], blk[1] * makeup[s]
        for ch in range(2):
            err = np.abs(y[2 * s + ch] - ref[ch]).max()
            assert err <= 2e-13, (s, ch, err)


def test_processor_level_latency_crossfade(amd, oracle):
    """SURVEY N1 transitions: irPeakLatency changing on a live stream.  The dry path cross-fades over 20 ms from the
    delay in use to the new one; a change of one sample is below the reference's 2-sample threshold and is not followed;
    a change that arrives while a fade runs waits for its end (here: in the middle of a call); in-place device calls
    (the delay ring has the input before the convolver overwrites it)."""
    import torch
    O = oracle
    S, T = 2, 3
    n = T * B
    irs = [O.gen_ir(2000, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    # ir_peak_latency per call; stream 1 moves while stream 0 rests.  Calls are 3 callbacks long except where noted: a
    # one-callback call starts a fade and the next call already carries another latency -- it has to wait for the end of
    # the running fade, which comes inside that call's first callback, so the new fade starts at its second callback.
    peaks = [[100, 100, 700, 700, 701, 701, 40, 40, 40, 40, 40],
             [0, 0, 0, 1500, 300, 300, 300, 300, 301, 900, 120]]
    blocks = [3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 3]
    calls = len(peaks[0])
    x = make_inputs(O, S, sum(blocks) * B)
    eng = amd.BatchedEngine(S, max_ir_len=2000, max_blocks_per_call=T)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    eng.set_conv_level(amd.CPQ_LEVEL_PROCESSOR)
    outs = []
    o = 0
    for k in range(calls):
        for s in range(S):
            eng.set_convproc_params(s, mix=0.6, ir_peak_latency=peaks[s][k])
        m = blocks[k] * B
        d = torch.from_numpy(np.ascontiguousarray(x[:, o:o + m])).cuda()
        eng._ck(eng._lib.cpq_convproc_process_device(eng._h, d.data_ptr(), d.data_ptr(), m))
        torch.cuda.synchronize()
        outs.append(d.cpu().numpy())
        o += m
    y = np.concatenate(outs, axis=1)
    eng.close()
    for s in range(S):
        per_cb = [p for p, nb in zip(peaks[s], blocks) for _ in range(nb)]
        for ch in range(2):
            ref = O.convproc_latency_schedule(irs[2 * s + ch], x[2 * s + ch], B, per_cb, mix=0.6)
            err = np.abs(y[2 * s + ch] - ref).max()
            assert err <= 1e-13, (s, ch, err)


def test_processor_level_direct_head_latency_start(amd, oracle):
    """With the direct head the processor's algorithm latency is 0 (Runtime.cpp:266) while prepareToPlay starts the
    latency compensation at latency + irLatency (Lifecycle.cpp:380-383): the first callbacks cross-fade the dry read
    from block + irPeakLatency to irPeakLatency."""
    O = oracle
    T = 4
    irs = [O.gen_ir(3000, channel=ch) for ch in range(2)]
    x = make_inputs(O, 1, 3 * T * B)
    eng = amd.BatchedEngine(1, max_ir_len=3000, max_blocks_per_call=T)
    eng.set_impulse(0, irs[0], irs[1], direct_head=True)
    eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=0.5, ir_peak_latency=77)
    eng.set_conv_level(amd.CPQ_LEVEL_PROCESSOR)
    assert eng.convproc_delay(0) == 77
    y = np.concatenate([eng.convproc_process(x[:, k * T * B:(k + 1) * T * B]) for k in range(3)], axis=1)
    eng.close()
    for ch in range(2):
        ref = O.convproc_latency_schedule(irs[ch], x[ch], B, [77] * (3 * T), mix=0.5, direct_head=True)
        assert np.abs(y[ch] - ref).max() <= 1e-13


@pytest.mark.parametrize("eq_mode", ["seq", "auto"])
def test_eq_band_reset_waits_for_silence(amd, oracle, eq_mode):
    """SURVEY N4 (band-reset handshake): requestBandReset is deferred until a callback whose input block is silent
    (every sample <= 1e-8 in magnitude; a block of 1e-9 noise counts, one sample of 2e-8 does not), single bands and all
    bands, a request that stays pending across calls, and a request consumed by a bypass fade instead."""
    O = oracle
    S, T, calls = 2, 4, 8
    n = T * B
    po = O.eq_params_bench(0.2)
    x = make_inputs(O, S, calls * n)
    rng = np.random.default_rng(2)
    # stream 0: callback 9 (call 2, t = 1) almost silent but one sample at 2e-8; callback 14 (call 3, t = 2) 1e-9 noise
    x[0:2, 9 * B:10 * B] = rng.standard_normal((2, B)) * 1e-9
    x[1, 9 * B + 17] = 2e-8
    x[0:2, 14 * B:15 * B] = rng.standard_normal((2, B)) * 1e-9
    # stream 1: callback 5 (call 1, t = 1) exactly zero
    x[2:4, 5 * B:6 * B] = 0.0
    eng = amd.BatchedEngine(S, max_ir_len=B, max_blocks_per_call=T)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    if eq_mode == "seq":
        eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL)
    req = [[0] * calls, [0, 0, 0, 0, 0, 1, 0, 0]]               # stream 1: bypass engaged in call 5, released in call 6
    outs = []
    for k in range(calls):
        if k == 1:
            eng.request_band_reset(0, (1 << 3) | (1 << 17))
            eng.request_band_reset(1)
        if k == 5:
            eng.request_band_reset(1, 1 << 7)                   # consumed by the fade-out callback
        eng.set_eq_bypass(1, req[1][k])
        outs.append(eng.eq_process(x[:, k * n:(k + 1) * n]))
    y = np.concatenate(outs, axis=1)
    eng.close()
    tol = 0.0 if eq_mode == "seq" else 2e-14
    for s in range(S):
        rl, rr = [], []
        # the same requests at the same calls
        ref = O.EqWithBypass(O.EqParams.from_buffer_copy(po), 48000.0, B)
        for k in range(calls):
            if k == 1:
                ref.request_band_reset((1 << 3) | (1 << 17) if s == 0 else 0xFFFFFFFF)
            if k == 5 and s == 1:
                ref.request_band_reset(1 << 7)
            for t in range(T):
                o = (k * T + t) * B
                a, b = ref.callback(x[2 * s, o:o + B].copy(), x[2 * s + 1, o:o + B].copy(), bool(req[s][k]))
                rl.append(a)
                rr.append(b)
            if s == 0:
                assert (ref.pending != 0) == (1 <= k <= 2), k          # fires at callback 14, not at 9
            else:
                assert ref.pending == 0, k                             # zero block in call 1; the fade in call 5
        rl, rr = np.concatenate(rl), np.concatenate(rr)
        err = max(np.abs(y[2 * s] - rl).max(), np.abs(y[2 * s + 1] - rr).max())
        assert err <= tol, (s, err)
    # the reset is audible: without it the output after callback 14 differs
    plain = O.EqWithBypass(O.EqParams.from_buffer_copy(po), 48000.0, B)
    pl = np.concatenate([plain.callback(x[0, c * B:(c + 1) * B].copy(), x[1, c * B:(c + 1) * B].copy(), False)[0] for c in range(16)])
    assert np.abs(pl[15 * B:16 * B] - y[0, 15 * B:16 * B]).max() > 1e-9


def _random_eq_params(O, rng, allow_ms=True):
    p = O.eq_params_default()
    freqs = np.exp(rng.uniform(np.log(15.0), np.log(26000.0), 20))          # beyond the clamps [20, min(20000, 0.95 nyq)] on purpose
    for i in range(20):
        b = p.bands[i]
        b.frequency = float(freqs[i])
        b.gain = float(rng.choice([0.0, 0.005, rng.uniform(-50, 50), rng.uniform(-6, 6), rng.uniform(-6, 6)]))
        b.q = float(np.exp(rng.uniform(np.log(0.005), np.log(25.0))))
        b.enabled = int(rng.random() < 0.8)
        b.type = int(rng.integers(0, 5))
        b.channelMode = int(rng.choice([0, 0, 0, 1, 2] + ([3, 4] if allow_ms else [])))
    p.nonlinearSaturation = float(rng.choice([0.0, 0.2, 0.2, 1.0, rng.uniform(0, 1)]))
    p.totalGainDb = float(rng.choice([0.0, rng.uniform(-24, 12)]))
    p.filterStructure = int(rng.random() < 0.3)
    p.agcEnabled = int(rng.random() < 0.2)
    return p


@pytest.mark.parametrize("seed,sr,blk", [(1, 48000.0, 512), (2, 48000.0, 512), (3, 48000.0, 512), (4, 96000.0, 512),
                                         (5, 44100.0, 256), (6, 192000.0, 1024), (7, 48000.0, 64)])
def test_eq_random_parameter_sweep(amd, oracle, seed, sr, blk):
    """Seeded sweep of the EQ parameter space: eight streams per engine, every stream its own random parameter set (all
    five band types, frequencies / gains / Q beyond the clamps, disabled and flat bands, Left / Right / Mid / Side
    modes, serial and parallel structure, saturation 0..1, total gain, AGC), in the automatic kernel choice and on the
    sequential kernel, levels from -40 dBFS to clipping, several sample rates and callback sizes.  GPU vs oracle."""
    O = oracle
    rng = np.random.default_rng(seed)
    S, T, calls = 8, 2048 // blk, 3
    n = T * blk
    params = [_random_eq_params(O, rng) for _ in range(S)]
    x = make_inputs(O, S, calls * n)
    for s in range(S):
        x[2 * s:2 * s + 2] *= float(rng.choice([0.04, 1.0, 4.0, 12.0]))
    refs = []
    for s in range(S):
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], params[s], sr=sr, block=blk)
        refs.append((yl, yr))
    for mode in ("auto", "seq"):
        eng = amd.BatchedEngine(S, block_size=blk, max_ir_len=blk, max_blocks_per_call=T, sample_rate=sr)
        for s in range(S):
            eng.set_eq_params(s, _copy_params(params[s], amd.eq_params_default()))
        if mode == "seq":
            eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL)
        y = np.concatenate([eng.eq_process(x[:, k * n:(k + 1) * n]) for k in range(calls)], axis=1)
        eng.close()
        for s in range(S):
            scale = max(1.0, float(np.abs(refs[s][0]).max()), float(np.abs(refs[s][1]).max()))
            err = max(np.abs(y[2 * s] - refs[s][0]).max(), np.abs(y[2 * s + 1] - refs[s][1]).max()) / scale
            assert err <= (0.0 if mode == "seq" else 1e-12), (seed, sr, blk, mode, s, err)


@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_convolver_random_configuration_sweep(amd, oracle, seed):
    """Seeded sweep of the convolver's configuration space against the stateful emulation: IR length, block size,
    FilterSpec on / off with random HC / LC / tail modes, tail start, strength and layer multiplier (powers of two),
    scale, direct head, uniform and native non-uniform schedule, calls of varying length."""
    O = oracle
    rng = np.random.default_rng(1000 + seed)
    block = int(rng.choice([64, 128, 256, 512, 512, 1024, 2048]))
    ir_len = int(np.exp(rng.uniform(np.log(40), np.log(90000))))
    T = int(rng.choice([1, 2, 3, 5, 8]))
    use_spec = rng.random() < 0.6
    kw = {}
    if use_spec:
        kw = dict(hc_mode=int(rng.integers(0, 3)), lc_mode=int(rng.integers(0, 2)), tail_mode=int(rng.integers(0, 3)),
                  tail_start_seconds=float(rng.choice([0.02, 0.085, 0.2, 0.5])), tail_strength=float(rng.uniform(0.0, 2.5)),
                  tail_l1l2_multiplier=int(rng.choice([2, 4, 8, 16])), sample_rate=float(rng.choice([44100.0, 48000.0, 96000.0])))
        kw["tail_enabled"] = int(kw["tail_mode"] != 2)
        if kw["tail_mode"] == 0:          # air absorption raises the multiplier to at least 6 (NUC.cpp:652): 6 x block is not a
            kw["tail_l1l2_multiplier"] = int(rng.choice([8, 16]))      # power of two and the reference's FFT plan breaks there
    names = {"hc_mode": "hcMode", "lc_mode": "lcMode", "tail_enabled": "tailEnabled", "tail_mode": "tailMode",
             "sample_rate": "sampleRate", "tail_start_seconds": "tailStartSeconds", "tail_strength": "tailStrength",
             "tail_l1l2_multiplier": "tailL1L2Multiplier"}
    sa = amd.FilterSpec.defaults(**kw) if use_spec else None
    so = O.FilterSpec.defaults(applySpectrumFilter=1, **{names[k]: v for k, v in kw.items()}) if use_spec else None
    scale = float(rng.choice([1.0, rng.uniform(0.1, 2.0)]))
    direct = bool(rng.random() < 0.3)
    sched = amd.CPQ_SCHED_REFERENCE_NUC if rng.random() < 0.3 else amd.CPQ_SCHED_UNIFORM
    irs = [O.gen_ir(ir_len, seed=0x1257 + seed, channel=ch) for ch in range(2)]
    total_blocks = max(3 * T, (ir_len + 3 * 4096) // block + 2 * T)
    total_blocks = min(total_blocks, 40000 // block * 8 + 3 * T)
    cfg = (block, ir_len, T, kw, scale, direct, sched)
    eng = amd.BatchedEngine(1, block_size=block, max_ir_len=ir_len, max_blocks_per_call=T, schedule=sched,
                            sample_rate=kw.get("sample_rate", 48000.0))
    eng.set_impulse(0, irs[0], irs[1], scale=scale, direct_head=direct, spec=sa)
    # calls of 1..T blocks
    sizes = []
    left = total_blocks
    while left > 0:
        k = int(min(left, rng.integers(1, T + 1)))
        sizes.append(k)
        left -= k
    x = make_inputs(O, 1, total_blocks * block)
    ref = np.empty_like(x)
    for c in range(2):
        nuc = O.Nuc()
        assert nuc.set_impulse(irs[c], block, scale=scale, direct=direct, spec=so), cfg
        ref[c] = nuc.run(x[c], block)
        nuc.close()
    outs, o = [], 0
    for k in sizes:
        outs.append(eng.conv_process(x[:, o:o + k * block]))
        o += k * block
    y = np.concatenate(outs, axis=1)
    eng.close()
    err = rms(y - ref) / max(rms(ref), 1e-30)
    assert err <= 1e-12, (cfg, err)


def test_non_power_of_two_tail_partition_is_refused(amd, oracle):
    """Air absorption with a multiplier below 6 makes the tail partition 6 x the block (NUC.cpp:652): the reference's FFT
    plan rounds such a size down to a power of two (FFTBackend.cpp:27-30) and its output is no convolution, so the engine
    refuses the plan instead of imitating it."""
    O = oracle
    eng = amd.BatchedEngine(1, block_size=256, max_ir_len=21259, max_blocks_per_call=1)
    ir = O.gen_ir(21259)
    with pytest.raises(amd.CpqError) as e:
        eng.set_impulse(0, ir, ir, spec=amd.FilterSpec.defaults(tail_mode=0, tail_start_seconds=0.5, tail_l1l2_multiplier=4))
    assert e.value.status == -5 and "1536" in str(e.value)
    eng.set_impulse(0, ir, ir, spec=amd.FilterSpec.defaults(tail_mode=0, tail_start_seconds=0.5, tail_l1l2_multiplier=8))
    eng.close()


@pytest.mark.parametrize("seed", list(range(11, 31)))
def test_whole_chain_random_transition_sequence(amd, oracle, seed):
    """Everything that can move on a live stream, at once and at random: per call and stream the mix, the IR peak
    latency, the EQ bypass request, band-reset requests (single bands / all) and the total gain change with some
    probability, DSPCore's convolver bypass toggles now and then, every third seed runs EQ -> conv with trim gains; silent callbacks are sprinkled in so that pending resets fire.  Chain: processor-level convolver ->
    EQ -> output filter -> make-up gain through cpq_engine_process_block, against the per-callback restatements
    (ConvProcStream, EqWithBypass, the output-filter oracle) chained the same way."""
    O = oracle
    rng = np.random.default_rng(seed)
    S, T, calls = 3, 3, 14
    n = T * B
    irs = [O.gen_ir(int(rng.integers(600, 4000)), stream=s, channel=ch) for s in range(S) for ch in range(2)]
    ir_len = max(len(h) for h in irs)
    irs = [np.concatenate([h, np.zeros(ir_len - len(h))]) for h in irs]
    x = make_inputs(O, S, calls * n)
    for _ in range(6):                                   # silent callbacks (exact zeros or 1e-9 noise)
        s, cb = int(rng.integers(0, S)), int(rng.integers(0, calls * T))
        x[2 * s:2 * s + 2, cb * B:(cb + 1) * B] = 0.0 if rng.random() < 0.5 else rng.standard_normal((2, B)) * 1e-9
    po = O.eq_params_bench(0.2)
    po.bands[5].gain = 0.0
    q = O.outfilter_design(0, 1, 0, 1, 48000.0)
    makeup = [1.0, 0.7, 1.3]
    order = amd.CPQ_ORDER_EQ_THEN_CONV if seed % 3 == 0 else amd.CPQ_ORDER_CONV_THEN_EQ
    trim = [0.5, 1.0, 1.7]                               # convolverInputTrimGain, used in EQ -> conv order
    conv_byp = False
    mix = [float(rng.uniform(0.2, 1.0)) for _ in range(S)]
    peak = [int(rng.integers(0, 1500)) for _ in range(S)]
    byp = [False] * S
    gain_db = [0.0] * S
    eng = amd.BatchedEngine(S, max_ir_len=ir_len, max_blocks_per_call=T)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
        eng.set_convproc_params(s, mix=mix[s], ir_peak_latency=peak[s])
        eng.set_gains(s, trim[s], makeup[s])
    eng.set_order(order)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    if seed % 2:                                          # odd seeds: sequential EQ kernel, even seeds: time-parallel
        eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL)
    eng.set_conv_level(amd.CPQ_LEVEL_PROCESSOR)
    eng.set_outfilter_params(amd.CPQ_ALL_STREAMS, 0, 1, 0, 1)
    eng.enable_output_filter(True)
    conv = [None] * S        # created at the first block the convolver stage sees: what is set before that applies at once
    eqs = [O.EqWithBypass(O.EqParams.from_buffer_copy(po), 48000.0, B) for _ in range(S)]
    ofs = [None] * S
    ref = np.empty_like(x)
    outs = []
    lazy = seed % 4 == 0
    last_proc, last_byp = [None] * S, [None] * S
    for k in range(calls):
        if rng.random() < 0.12:                          # DSPCore's convBypassed: the convolver stage is not called
            conv_byp = not conv_byp
        eng.set_conv_bypass(conv_byp)
        for s in range(S):
            if k > 0 and rng.random() < 0.3:
                mix[s] = float(rng.uniform(0.05, 1.0))
            if k > 0 and rng.random() < 0.3:
                peak[s] = int(rng.integers(0, 1500)) if rng.random() < 0.8 else peak[s] + 1
            if rng.random() < 0.25:
                byp[s] = not byp[s]
            # a host sets a parameter when it changes, not before every block (seeds divisible by 4; the others set all
            # of them before every call)
            if not lazy or k == 0 or (mix[s], peak[s]) != last_proc[s]:
                eng.set_convproc_params(s, mix=mix[s], ir_peak_latency=peak[s])
                last_proc[s] = (mix[s], peak[s])
            if not lazy or k == 0 or byp[s] != last_byp[s]:
                eng.set_eq_bypass(s, byp[s])
                last_byp[s] = byp[s]
            if k == 0:
                eqs[s].sync(byp[s])                     # requested before the first block: the state follows at once
            if rng.random() < 0.2:
                mask = 0xFFFFFFFF if rng.random() < 0.4 else int(rng.integers(1, 1 << 20))
                eng.request_band_reset(s, mask)
                eqs[s].request_band_reset(mask)
            if rng.random() < 0.15:
                gain_db[s] = float(rng.uniform(-9.0, 3.0))
                pa = _copy_params(po, amd.eq_params_default())
                pa.total_gain_db = gain_db[s]
                eng.set_eq_params(s, pa)
                eqs[s].set_total_gain_db(gain_db[s], before_first_block=(k == 0))
        outs.append(eng.process(x[:, k * n:(k + 1) * n]))
        for s in range(S):
            for t in range(T):
                o = (k * T + t) * B
                a, b = x[2 * s, o:o + B].copy(), x[2 * s + 1, o:o + B].copy()
                if not conv_byp and conv[s] is None:
                    conv[s] = O.ConvProcStream(irs[2 * s], irs[2 * s + 1], B, mix[s], peak[s])
                if order == amd.CPQ_ORDER_CONV_THEN_EQ:
                    if not conv_byp:
                        a, b = conv[s].callback(a, b, mix[s], peak[s])
                    a, b = eqs[s].callback(np.ascontiguousarray(a), np.ascontiguousarray(b), byp[s])
                else:
                    a, b = eqs[s].callback(a, b, byp[s])
                    if not conv_byp:
                        if abs(trim[s] - 1.0) > 1e-12:
                            a, b = a * trim[s], b * trim[s]
                        a, b = conv[s].callback(a, b, mix[s], peak[s])
                if not (conv_byp and byp[s]):            # the output filter runs when the convolver or the EQ is active
                    a, b, ofs[s] = O.outfilter_process_stereo(np.ascontiguousarray(a), np.ascontiguousarray(b), q, ofs[s])
                ref[2 * s, o:o + B], ref[2 * s + 1, o:o + B] = a * makeup[s], b * makeup[s]
    y = np.concatenate(outs, axis=1)
    eng.close()
    for s in range(S):
        err = np.abs(y[2 * s:2 * s + 2] - ref[2 * s:2 * s + 2]).max()
        assert err <= 1e-12, (seed, s, err)


def test_eq_agc_reset_request(amd, oracle):
    """requestAgcReset: the AGC envelopes and gain of the stream restart at its next processed block -- also when that
    block comes only after a bypass has been released."""
    O = oracle
    S, T, calls = 2, 4, 8
    n = T * B
    po = O.eq_params_bench(0.2)
    po.agcEnabled = 1
    x = make_inputs(O, S, calls * n)
    x[0:2] *= 3.0                                        # loud: the AGC gain settles well below 1
    eng = amd.BatchedEngine(S, max_ir_len=B, max_blocks_per_call=T)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL)
    refs = [O.EqWithBypass(O.EqParams.from_buffer_copy(po), 48000.0, B) for _ in range(S)]
    byp1 = [0, 0, 0, 1, 1, 0, 0, 0]                      # stream 1: request while bypassed, consumed at the release
    outs, ref = [], np.empty_like(x)
    for k in range(calls):
        if k == 3:
            eng.request_agc_reset(0); refs[0].request_agc_reset()
        if k == 4:
            eng.request_agc_reset(1); refs[1].request_agc_reset()
        eng.set_eq_bypass(1, byp1[k])
        outs.append(eng.eq_process(x[:, k * n:(k + 1) * n]))
        for s in range(S):
            for t in range(T):
                o = (k * T + t) * B
                a, b = refs[s].callback(x[2 * s, o:o + B].copy(), x[2 * s + 1, o:o + B].copy(), bool(byp1[k]) if s == 1 else False)
                ref[2 * s, o:o + B], ref[2 * s + 1, o:o + B] = a, b
    y = np.concatenate(outs, axis=1)
    eng.close()
    assert np.array_equal(y, ref)
    plain = O.eq_process_stereo(x[0], x[1], po)[0]
    assert np.abs(plain[3 * n:4 * n] - y[0, 3 * n:4 * n]).max() > 1e-3      # the reset is audible


def test_abi_argument_errors_leave_the_engine_usable(amd, oracle):
    """Every entry point with arguments it must refuse (null handle, null buffers, stream out of range, sizes that are
    not block multiples, calls before the data they need, non-finite gains): a negative status, a message in
    cpq_last_error, no crash -- and the engine computes the same result afterwards."""
    import ctypes as C
    O = oracle
    K = amd._capi
    L = K.load()
    eng = amd.BatchedEngine(2, max_ir_len=2048, max_blocks_per_call=2)
    h = eng._h
    n = 2 * B
    x = make_inputs(O, 2, n)
    buf_in = np.ascontiguousarray(x)
    buf_out = np.empty_like(buf_in)
    dp = lambda a: a.ctypes.data_as(K.c_double_p)
    # before set_impulse / set_eq_params / set_outfilter_params
    assert L.cpq_conv_process(h, dp(buf_in), dp(buf_out), n) == K.CPQ_ERR_NOT_READY
    assert L.cpq_eq_process(h, dp(buf_in), dp(buf_out), n) == K.CPQ_ERR_NOT_READY
    assert L.cpq_outfilter_process(h, dp(buf_in), dp(buf_out), n) == K.CPQ_ERR_NOT_READY
    assert L.cpq_engine_process_block(h, dp(buf_in), dp(buf_out), n) == K.CPQ_ERR_NOT_READY
    assert L.cpq_conv_is_ready(h) == 0 and len(L.cpq_last_error(h)) > 0
    ir = O.gen_ir(2048)
    eng.set_impulse(amd.CPQ_ALL_STREAMS, ir, ir)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(O.eq_params_bench(0.2), amd.eq_params_default()))
    y0 = eng.process(x)
    eng.conv_reset(); eng.eq_reset()
    null = C.c_void_p(None)
    bad = []
    bad.append(L.cpq_conv_process(null, dp(buf_in), dp(buf_out), n))
    bad.append(L.cpq_conv_process(h, None, dp(buf_out), n))
    bad.append(L.cpq_conv_process(h, dp(buf_in), None, n))
    bad.append(L.cpq_conv_process(h, dp(buf_in), dp(buf_out), 0))
    bad.append(L.cpq_conv_process(h, dp(buf_in), dp(buf_out), -B))
    bad.append(L.cpq_conv_process(h, dp(buf_in), dp(buf_out), B + 1))
    bad.append(L.cpq_conv_process(h, dp(buf_in), dp(buf_out), 3 * B))                  # more than max_blocks_per_call
    bad.append(L.cpq_conv_process_device(h, C.c_void_p(8), C.c_void_p(16), n))         # misaligned device pointers
    bad.append(L.cpq_conv_set_impulse(h, 2, dp(ir), dp(ir), 2048, 1.0, 0, None))      # stream out of range
    bad.append(L.cpq_conv_set_impulse(h, 0, None, dp(ir), 2048, 1.0, 0, None))
    bad.append(L.cpq_conv_set_impulse(h, 0, dp(ir), dp(ir), 0, 1.0, 0, None))
    bad.append(L.cpq_conv_set_impulse(h, 0, dp(ir), dp(ir), 4096, 1.0, 0, None))      # longer than max_ir_len
    bad.append(L.cpq_eq_set_params(h, 0, None))
    bad.append(L.cpq_eq_set_params(h, -2, C.byref(amd.eq_params_default())))
    bad.append(L.cpq_eq_set_bypass(h, 7, 1))
    bad.append(L.cpq_eq_request_band_reset(h, 7, 1))
    bad.append(L.cpq_eq_request_agc_reset(h, -3))
    bad.append(L.cpq_eq_set_mode(h, 9))
    bad.append(L.cpq_engine_set_order(h, 5))
    bad.append(L.cpq_engine_set_conv_level(h, 5))
    bad.append(L.cpq_engine_set_gains(h, 0, float("nan"), 1.0))
    bad.append(L.cpq_engine_set_gains(h, 0, 1.0, float("inf")))
    bad.append(L.cpq_engine_set_gains(h, 9, 1.0, 1.0))
    bad.append(L.cpq_convproc_set_params(h, 0, None))
    bad.append(L.cpq_convproc_set_params(h, 0, C.byref(K.ConvProcParams(1.5, 0, 0, 0.0))))      # mix > 1
    bad.append(L.cpq_convproc_set_params(h, 0, C.byref(K.ConvProcParams(0.5, 0, -1, 0.0))))     # negative latency
    bad.append(L.cpq_convproc_set_params(h, 0, C.byref(K.ConvProcParams(0.5, 0, 0, 5.0))))      # smoothing time out of range
    bad.append(L.cpq_outfilter_set_params(h, 0, 0, 7, 0, 0))
    bad.append(L.cpq_engine_prepare(h, -1.0, B))
    bad.append(L.cpq_engine_prepare(h, 48000.0, 64 * B))                               # beyond the engine's call size
    bad.append(L.cpq_profile_read(h, 99, None, None))
    bad.append(L.cpq_convproc_delay(h, 5))
    assert all(b < 0 for b in bad), bad
    created = K._E()
    d = K.EngineDesc(C.sizeof(K.EngineDesc), 0, 0, 512, 4096, 4, 0, 0, 48000.0, 0, 0)            # zero streams
    assert L.cpq_engine_create(C.byref(d), C.byref(created)) < 0 and not created.value
    d = K.EngineDesc(C.sizeof(K.EngineDesc), 99, 1, 512, 4096, 4, 0, 0, 48000.0, 0, 0)           # no such device
    assert L.cpq_engine_create(C.byref(d), C.byref(created)) < 0 and not created.value
    assert L.cpq_engine_create(None, C.byref(created)) < 0
    L.cpq_engine_destroy(None)
    # nothing above has touched the state
    assert np.array_equal(eng.process(x), y0)
    eng.close()


def test_two_engines_interleaved_and_no_device_memory_leak(amd, oracle):
    """Handles are independent: two engines with different configurations, called alternately on the same device, give
    what each gives alone.  Creating and destroying engines that have exercised every lazily allocated buffer (FilterSpec
    tail layers, processor-level ring, ramps, bypass cross-fade, AGC, silence flags, profiling events) returns the
    device memory."""
    import torch
    O = oracle
    x = make_inputs(O, 2, 8 * B)

    def build(kind):
        if kind == 0:
            e = amd.BatchedEngine(2, max_ir_len=9000, max_blocks_per_call=4)
            irs = [O.gen_ir(9000, stream=7, channel=c) for c in range(2)]
            e.set_impulse(amd.CPQ_ALL_STREAMS, irs[0], irs[1], spec=amd.FilterSpec.defaults(hc_mode=0))
            po = O.eq_params_bench(0.2); po.agcEnabled = 1
            e.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
            e.set_conv_level(amd.CPQ_LEVEL_PROCESSOR)
            e.set_outfilter_params(amd.CPQ_ALL_STREAMS, 0, 1, 0, 1)
            e.enable_output_filter(True)
        else:
            e = amd.BatchedEngine(2, block_size=256, max_ir_len=3000, max_blocks_per_call=8)
            irs = [O.gen_ir(3000, stream=3, channel=c) for c in range(2)]
            e.set_impulse(amd.CPQ_ALL_STREAMS, irs[0], irs[1], direct_head=True)
            e.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(O.eq_params_bench(0.0), amd.eq_params_default()))
            e.set_order(amd.CPQ_ORDER_EQ_THEN_CONV)
        return e

    def run(e, k, step):
        if step == 1:                                    # move everything that allocates lazily
            e.set_eq_bypass(0, True)
            e.request_band_reset(1, 0xFFFFFFFF)
            e.request_agc_reset(1)
            e.set_gains(0, 0.5, 1.5)
            if k == 0:
                e.set_convproc_params(0, mix=0.4, ir_peak_latency=300)
        if step == 2:
            e.set_eq_bypass(0, False)
        n = 4 * B if k == 0 else 8 * 256
        return e.process(x[:, :n])

    alone = []
    for k in range(2):
        e = build(k)
        e.profile_enable(True)
        alone.append([run(e, k, s) for s in range(3)])
        e.profile_read()
        e.close()
    a, b = build(0), build(1)
    both = [[], []]
    for s in range(3):
        both[0].append(run(a, 0, s))
        both[1].append(run(b, 1, s))
    a.close(); b.close()
    for k in range(2):
        for s in range(3):
            assert np.array_equal(both[k][s], alone[k][s]), (k, s)
    # leak check with engines large enough that every lazily allocated buffer is >= 1 MB: memory after 4 and after 16
    # create / use / destroy cycles must agree (allocator slack does not grow, a leak does: >= 12 MB per forgotten buffer)
    def big(kind):
        S2 = 32
        if kind == 0:
            e = amd.BatchedEngine(S2, max_ir_len=9000, max_blocks_per_call=8)
            ir = O.gen_ir(9000, stream=7)
            e.set_impulse(amd.CPQ_ALL_STREAMS, ir, ir, spec=amd.FilterSpec.defaults(hc_mode=0))
            po = O.eq_params_bench(0.2); po.agcEnabled = 1
            e.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
            e.set_conv_level(amd.CPQ_LEVEL_PROCESSOR)
            e.set_outfilter_params(amd.CPQ_ALL_STREAMS, 0, 1, 0, 1)
            e.enable_output_filter(True)
        else:
            e = amd.BatchedEngine(S2, block_size=256, max_ir_len=3000, max_blocks_per_call=16)
            ir = O.gen_ir(3000, stream=3)
            e.set_impulse(amd.CPQ_ALL_STREAMS, ir, ir, direct_head=True)
            e.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(O.eq_params_bench(0.0), amd.eq_params_default()))
            e.set_order(amd.CPQ_ORDER_EQ_THEN_CONV)
        return e
    xb = np.tile(x[:2, :8 * 512], (32, 1))
    free = []
    for i in range(16):
        e = big(i % 2)
        e.profile_enable(True)
        for st in range(3):
            if st == 1:
                e.set_eq_bypass(0, True); e.request_band_reset(1, 0xFFFFFFFF); e.request_agc_reset(1); e.set_gains(0, 0.5, 1.5)
                if i % 2 == 0:
                    e.set_convproc_params(0, mix=0.4, ir_peak_latency=300)
            if st == 2:
                e.set_eq_bypass(0, False)
            e.process(xb[:, :4096])
        e.close()
        torch.cuda.synchronize()
        free.append(torch.cuda.mem_get_info()[0])
    assert abs(free[3] - free[15]) < 4 << 20, free


def test_processor_level_dry_only_with_unequal_ramps_fails_before_any_state_moves(amd, oracle):
    """Two streams on the uniform path are set dry-only together while one of them is in the middle of a mix ramp (it keeps
    its remaining step count): when its ramp has run out and the other one's has not, one convolver would rest and the other
    not -- which the uniform path cannot do (a per-stream rest needs a plan group).  The call must be refused BEFORE the ramp replay has consumed anything: refused twice with the same
    status, and after stream 0 is given a mix again the engine continues exactly like a twin that never made the refused calls
    (the reference keeps convolving while a ramp runs: isSmoothing || mix > 0.001, ConvolverProcessor.Runtime.cpp:373-375)."""
    O = oracle
    S, T = 2, 4
    irs = [O.gen_ir(3000, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    x = make_inputs(O, S, 6 * T * B)

    def run(with_refused_calls):
        eng = amd.BatchedEngine(S, max_ir_len=3000, max_blocks_per_call=T)
        for s in range(S):
            eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
        call = lambda k: eng.convproc_process(x[:, k * T * B:(k + 1) * T * B])
        out = [call(0)]                                          # both wet (the first call snaps)
        eng.set_convproc_params(0, mix=0.5)
        out.append(call(1))                                      # stream 0 ramps (4800 samples = 2.3 calls): 2752 left
        eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=0.0)    # both dry-only: stream 0 keeps its 2752 steps, stream 1 starts 4800
        out.append(call(2))
        out.append(call(3))                                      # stream 0's ramp ends inside this call
        if with_refused_calls:
            for _ in range(2):                                   # stream 0 would rest, stream 1 still ramps
                with pytest.raises(amd.CpqError) as ei:
                    call(4)
                assert ei.value.status == -5          # CPQ_ERR_UNSUPPORTED
        eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=0.6)
        out.append(call(4))
        out.append(call(5))
        eng.close()
        return np.concatenate(out, axis=1)

    a, b = run(True), run(False)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("blk", [300, 441, 512, 1000])
def test_eq_short_spans_guarded_path_and_state_hand_over(amd, oracle, blk):
    """Calls below 1024 samples run on k_svf_cascade_short (one wave up to 512 samples, two above; the last chunk of 8 partly
    padding at 300 / 441 / 1000).  Twelve one-callback calls: the band states pass from call to call through the state array; a
    NaN, an Inf and a 1e12 sample send three of the calls through the kernel's guarded recurrence (bit-faithful), the call
    behind the 1e12 sample starts from out-of-range states (guarded by its states, not its input)."""
    O = oracle
    S, calls = 2, 12
    x = make_inputs(O, S, calls * blk)
    x[0, 2 * blk + 17] = np.nan
    x[3, 5 * blk + blk - 1] = np.inf
    x[1, 8 * blk + blk - 3] = 1.0e12
    po = O.eq_params_bench(0.2)
    po.bands[2].channelMode = 1
    po.bands[11].type = 4
    eng = amd.BatchedEngine(S, block_size=blk, max_ir_len=512, max_blocks_per_call=1, call_mode=amd.CPQ_CALLS_ANY)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    y = np.concatenate([eng.eq_process(x[:, k * blk:(k + 1) * blk]) for k in range(calls)], axis=1)
    assert np.all(np.isfinite(y))
    for s in range(S):
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], po, block=blk)
        d = np.maximum(np.abs(y[2 * s] - yl), np.abs(y[2 * s + 1] - yr))
        big = 8 * blk + blk - 3            # behind the 1e12 sample the states decay from 1e11 through fast-path rounding
        assert d[:big].max() <= 1e-12 and d.max() <= (1e-4 if s == 0 else 1e-12), (s, d[:big].max(), d.max())
    eng.close()
