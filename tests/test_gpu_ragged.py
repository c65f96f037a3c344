"""GPU parity for CPQ_CALLS_ANY: any call quantum and ragged calls.

The reference accepts any blockSize and any n: `Add` accumulates input per layer until a partition of nextPow2(max(bs, 64))
(x m, x m^2) samples is full (src/MKLNonUniformConvolver.cpp:1431-1446), `Get` reads layer 0 through the output ring and
zero-fills a short read at the end (:1376-1402), the tail layers follow the distributed MAC (:1497-1545) and the delay-line
reader per call (:1653-1688); StereoConvolver::process is Add + Get per callQuantum chunk (Runtime.cpp:659-682, 1159-1184).
The oracle restates exactly that (orc_nuc_add / orc_nuc_get); the HIP path must reproduce it chunk for chunk -- including
the start-up zeros and the zero gaps the reference produces at quanta like 441.  Tolerance 1e-13 RMS (north_star 1e-12);
samples the reference zero-fills must be exactly zero.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


@pytest.fixture(scope="module")
def amd():
    import convopeq_amd
    return convopeq_amd


def make_inputs(O, stream_ids, n):
    x = np.empty((2 * len(stream_ids), n))
    for i, s in enumerate(stream_ids):
        for ch in range(2):
            x[2 * i + ch] = O.gen_pcm(n, stream=s, channel=ch)
    return x


def oracle_calls(O, ir, x, quantum, call_sizes, direct=False, spec=None, scale=1.0):
    """Add + Get per chunk of `quantum` samples inside every call, as StereoConvolver::process is driven."""
    nuc = O.Nuc()
    assert nuc.set_impulse(ir, quantum, scale=scale, direct=direct, spec=spec)
    y = np.empty_like(x)
    got_all = []
    pos = 0
    for n in call_sizes:
        o = 0
        while o < n:
            m = min(quantum, n - o)
            nuc.add(x[pos + o:pos + o + m])
            out, got = nuc.get(m)
            y[pos + o:pos + o + m] = out
            got_all.append(got)
            o += m
        pos += n
    nuc.close()
    return y[:pos], got_all


def run_engine(eng, x, call_sizes):
    ys, pos = [], 0
    for n in call_sizes:
        ys.append(eng.conv_process(np.ascontiguousarray(x[:, pos:pos + n])))
        pos += n
    return np.concatenate(ys, axis=1)


def check(y, ref, tol=1e-13):
    err = rms(y - ref)
    assert err <= tol, err
    zeros = ref == 0.0
    assert np.array_equal(y[zeros], ref[zeros])           # zero-filled samples stay exactly zero
    return err


@pytest.mark.parametrize("taps", [4096, 131072])
@pytest.mark.parametrize("quantum", [480, 441, 96, 1000])
def test_arbitrary_call_quantum(amd, oracle, quantum, taps):
    """Device block sizes that are not powers of two: 480 / 441 (10 ms at 48 / 44.1 kHz), 96, 1000."""
    O = oracle
    S = 2
    blocks_per_call = 16 if quantum >= 400 else 64
    calls = 6 if taps == 131072 else 3
    call_sizes = [blocks_per_call * quantum] * calls
    n = sum(call_sizes)
    irs = [O.gen_ir(taps, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    x = make_inputs(O, range(S), n)
    eng = amd.BatchedEngine(S, block_size=quantum, max_ir_len=taps, max_blocks_per_call=blocks_per_call, call_mode=amd.CPQ_CALLS_ANY)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    assert eng.is_ready()
    assert eng.latency() == max(64, 1 << (quantum - 1).bit_length())       # getLatency() = layer-0 partition
    y = run_engine(eng, x, call_sizes)
    worst = 0.0
    for c in range(2 * S):
        ref, _ = oracle_calls(O, irs[c], x[c], quantum, call_sizes)
        worst = max(worst, check(y[c], ref))
        assert rms(ref) > 1e-3
    print(f"quantum {quantum}, {taps} taps: worst rms err {worst:.3e}")
    # Reset() starts over: the same calls give the same output
    eng.conv_reset()
    y2 = run_engine(eng, x, call_sizes[:1])
    assert np.array_equal(y2, y[:, :call_sizes[0]])
    eng.close()


@pytest.mark.parametrize("quantum,taps", [(512, 20000), (256, 131072), (64, 6000)])
def test_ragged_call_sequence(amd, oracle, quantum, taps):
    """Calls of 1 ... 700 samples in a seeded random order (cut into quantum-sized chunks, the last one shorter)."""
    O = oracle
    rng = np.random.default_rng(1234 + quantum)
    call_sizes = [int(v) for v in rng.integers(1, 701, size=90)] + [1, 2, 700, 699, quantum, quantum - 1, quantum + 1]
    n = sum(call_sizes)
    S = 2
    irs = [O.gen_ir(taps, stream=5 + c // 2, channel=c % 2) for c in range(2 * S)]
    x = make_inputs(O, range(5, 5 + S), n)
    eng = amd.BatchedEngine(S, block_size=quantum, max_ir_len=taps, max_blocks_per_call=(700 + quantum - 1) // quantum,
                            call_mode=amd.CPQ_CALLS_ANY)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    y = run_engine(eng, x, call_sizes)
    worst = 0.0
    for c in range(2 * S):
        ref, _ = oracle_calls(O, irs[c], x[c], quantum, call_sizes)
        worst = max(worst, check(y[c], ref))
    print(f"ragged calls, quantum {quantum}, {taps} taps: worst rms err {worst:.3e}")
    eng.close()


def test_single_sample_calls_and_direct_head(amd, oracle):
    """n = 1 ... 5 samples per call with the direct head on (the head runs per chunk, :1169-1232, added before the tails)."""
    O = oracle
    quantum, taps = 128, 9000
    call_sizes = [1, 2, 3, 4, 5] * 120
    n = sum(call_sizes)
    irs = [O.gen_ir(taps, stream=9, channel=ch) for ch in range(2)]
    x = make_inputs(O, [9], n)
    eng = amd.BatchedEngine(1, block_size=quantum, max_ir_len=taps, max_blocks_per_call=1, call_mode=amd.CPQ_CALLS_ANY)
    eng.set_impulse(0, irs[0], irs[1], scale=0.7, direct_head=True)
    y = run_engine(eng, x, call_sizes)
    for c in range(2):
        ref, _ = oracle_calls(O, irs[c], x[c], quantum, call_sizes, direct=True, scale=0.7)
        assert rms(y[c] - ref) <= 1e-13
    eng.close()


def test_streams_with_their_own_plans(amd, oracle):
    """One NUC per channel with its own IR length and FilterSpec (src/ConvolverProcessor.h:741-814): four streams, four
    layer plans -- short single-layer IR, two-layer, three-layer, and a FilterSpec plan whose tail reader is time-varying
    (air absorption mode: layer 0 ends below the tail partition) -- in one engine, ragged calls, shared call sizes."""
    O = oracle
    quantum = 512
    lens = [3000, 40000, 131072, 60000]
    specs = [None, None, None, O.FilterSpec.defaults(tailMode=0, applySpectrumFilter=1, hcMode=0, lcMode=1)]
    S = len(lens)
    call_sizes = [512 * 8, 700, 512 * 16, 333, 512 * 12] * 3
    n = sum(call_sizes)
    x = make_inputs(O, range(20, 20 + S), n)
    eng = amd.BatchedEngine(S, block_size=quantum, max_ir_len=max(lens), max_blocks_per_call=16, call_mode=amd.CPQ_CALLS_ANY)
    irs = []
    for s in range(S):
        irs.append([O.gen_ir(lens[s], stream=20 + s, channel=ch) for ch in range(2)])
        spec = None
        if specs[s] is not None:
            spec = amd.FilterSpec.defaults(tail_mode=0, hc_mode=0, lc_mode=1)
        eng.set_impulse(s, irs[s][0], irs[s][1], spec=spec)
    y = run_engine(eng, x, call_sizes)
    for s in range(S):
        for ch in range(2):
            ref, _ = oracle_calls(O, irs[s][ch], x[2 * s + ch], quantum, call_sizes, spec=specs[s])
            err = check(y[2 * s + ch], ref)
            print(f"stream {s} ({lens[s]} taps{', FilterSpec' if specs[s] is not None else ''}) ch {ch}: rms err {err:.3e}")
    # a stream reloaded while the others keep playing starts from silence like a new NUC; the others do not notice
    new_ir = [O.gen_ir(10000, stream=77, channel=ch) for ch in range(2)]
    eng.set_impulse(1, new_ir[0], new_ir[1])
    tail_calls = [512 * 4, 123, 512 * 6]
    x2 = make_inputs(O, range(20, 20 + S), n + sum(tail_calls))[:, n:]
    y2 = run_engine(eng, x2, tail_calls)
    for ch in range(2):
        ref, _ = oracle_calls(O, new_ir[ch], x2[2 + ch], quantum, tail_calls)
        assert rms(y2[2 + ch] - ref) <= 1e-13
    for s in (0, 2, 3):
        for ch in range(2):
            xx = np.concatenate([x[2 * s + ch], x2[2 * s + ch]])
            ref, _ = oracle_calls(O, irs[s][ch], xx, quantum, call_sizes + tail_calls, spec=specs[s])
            assert rms(y2[2 * s + ch] - ref[n:]) <= 1e-13
    eng.close()


def test_whole_block_engine_refuses_ragged_calls_loudly(amd, oracle):
    O = oracle
    eng = amd.BatchedEngine(1, block_size=512, max_ir_len=4096, max_blocks_per_call=4)
    eng.set_impulse(0, O.gen_ir(4096), O.gen_ir(4096, channel=1))
    with pytest.raises(amd.CpqError) as ei:
        eng.conv_process(np.zeros((2, 480)))
    assert "CPQ_CALLS_ANY" in str(ei.value)
    with pytest.raises(amd.CpqError):
        amd.BatchedEngine(1, block_size=480, max_ir_len=4096, max_blocks_per_call=4)        # non-power-of-two needs CPQ_CALLS_ANY
    eng.close()


@pytest.mark.parametrize("direct", [False, True])
def test_per_stream_bypass_rests_one_convolver(amd, oracle, direct):
    """ConvolverProcessor bypass of ONE stream (src/convolver/ConvolverProcessor.Runtime.cpp:123-186): its output is the
    delayed dry signal and its convolver is not called -- the NUC keeps the state it had and resumes with it on release --
    while the other streams go on.  Processor level, three streams with their own IRs, the middle one bypassed for two
    calls and released again.  (Dry-only, mix <= 0.001, rests the convolver through the same mechanism once its mix ramp
    has ended.)  direct: with the direct head, whose history must rest too (processDirectBlock sits inside Add(),
    src/MKLNonUniformConvolver.cpp:1407-1430): the first headTaps - 1 samples after a release show it."""
    O = oracle
    L = O.lib()
    quantum, T, taps, S = 512, 8, 20000, 3
    n_call = T * quantum
    mode = ["on", "on", "bypass", "bypass", "on", "on", "bypass", "on"]      # of stream 1, per call
    n = n_call * len(mode)
    irs = [[O.gen_ir(taps, stream=30 + s, channel=ch) for ch in range(2)] for s in range(S)]
    x = make_inputs(O, range(30, 30 + S), n)
    eng = amd.BatchedEngine(S, block_size=quantum, max_ir_len=taps, max_blocks_per_call=T, call_mode=amd.CPQ_CALLS_ANY)
    for s in range(S):
        eng.set_impulse(s, irs[s][0], irs[s][1], direct_head=direct)
    eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=1.0)
    ys = []
    for c, m in enumerate(mode):
        eng.set_convproc_params(1, mix=1.0, bypassed=(m == "bypass"))
        ys.append(eng.convproc_process(np.ascontiguousarray(x[:, c * n_call:(c + 1) * n_call])))
    y = np.concatenate(ys, axis=1)
    wet_g = L.orc_equal_power_sin(1.0)
    for s in range(S):
        for ch in range(2):
            xc = x[2 * s + ch]
            # the delayed dry signal: algorithm latency = layer-0 partition, or 0 with the direct head (Runtime.cpp:266; the
            # start-up cross-fade from block + irLatency is over long before the first bypassed call)
            dry = xc if direct else np.concatenate([np.zeros(quantum), xc[:-quantum]])
            nuc = O.Nuc()
            assert nuc.set_impulse(irs[s][ch], quantum, direct=direct)
            ref = np.empty(n)
            for c, m in enumerate(mode):
                seg = slice(c * n_call, (c + 1) * n_call)
                if s == 1 and m != "on":
                    ref[seg] = dry[seg]                  # the convolver rests: no Add / Get
                else:
                    ref[seg] = nuc.run(xc[seg], quantum) * wet_g
            nuc.close()
            assert rms(y[2 * s + ch] - ref) <= 1e-13, (s, ch)
    eng.close()


def test_nuc_level_call_runs_a_stream_that_rested_at_processor_level(amd, oracle):
    """Resting a convolver is processor-level state (ConvolverProcessor::process decides not to call its NUC).  A call at the
    kernel level -- the NUC's own Add / Get -- after a per-stream bypass runs every stream, the formerly resting one from
    the state it kept."""
    O = oracle
    quantum, T, taps, S = 512, 4, 9000, 2
    n_call = T * quantum
    irs = [[O.gen_ir(taps, stream=60 + s, channel=ch) for ch in range(2)] for s in range(S)]
    x = make_inputs(O, range(60, 60 + S), 3 * n_call)
    eng = amd.BatchedEngine(S, block_size=quantum, max_ir_len=taps, max_blocks_per_call=T, call_mode=amd.CPQ_CALLS_ANY)
    for s in range(S):
        eng.set_impulse(s, irs[s][0], irs[s][1])
    eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=1.0)
    y0 = eng.convproc_process(np.ascontiguousarray(x[:, :n_call]))                 # both streams run
    eng.set_convproc_params(1, mix=1.0, bypassed=True)
    eng.convproc_process(np.ascontiguousarray(x[:, n_call:2 * n_call]))             # stream 1 rests (its NUC sees nothing)
    y2 = eng.conv_process(np.ascontiguousarray(x[:, 2 * n_call:]))                  # kernel level: everybody runs
    assert y0.shape == y2.shape
    for s in range(S):
        for ch in range(2):
            nuc = O.Nuc()
            assert nuc.set_impulse(irs[s][ch], quantum)
            nuc.run(x[2 * s + ch, :n_call], quantum)
            if s == 0:
                nuc.run(x[2 * s + ch, n_call:2 * n_call], quantum)
            ref = nuc.run(x[2 * s + ch, 2 * n_call:], quantum)
            nuc.close()
            assert rms(y2[2 * s + ch] - ref) <= 1e-13, (s, ch)
    eng.close()


def _copy_eq(po, pa):
    for i in range(20):
        b, o = pa.bands[i], po.bands[i]
        b.frequency, b.gain, b.q, b.enabled, b.type, b.channel_mode = o.frequency, o.gain, o.q, o.enabled, o.type, o.channelMode
    pa.total_gain_db, pa.agc_enabled = po.totalGainDb, po.agcEnabled
    pa.nonlinear_saturation, pa.filter_structure = po.nonlinearSaturation, po.filterStructure
    return pa


@pytest.mark.parametrize("mode", ["sequential", "auto"])
@pytest.mark.parametrize("agc", [False, True])
def test_eq_short_last_callback_with_ramp_and_agc(amd, oracle, agc, mode):
    """CPQ_CALLS_ANY with the EQ's per-callback state in play: calls of k x 480 + r samples.  The short last callback is a
    callback of r samples to the total-gain ramp (skip(r)), to the block-rate AGC (RMS and table[numSamples] coefficients
    over r samples, src/eqprocessor/EQProcessor.Processing.cpp:376-427) and to the cascade, as in the reference's
    process(block); the oracle cuts the call the same way (orc_eq_process_stereo: blocks of `quantum`, the last shorter)."""
    O = oracle
    quantum, S = 480, 2
    call_sizes = [480 * 3 + 200, 480 * 2, 137, 480 + 479, 1, 480 * 4, 959, 480 * 4 + 1, 333, 480 * 4]
    gains_db = {0: -1.5, 2: 3.0, 3: -6.0, 6: 0.5}             # change points (call index -> new total gain): ramps run through
    n = sum(call_sizes)                                       # the ragged calls
    x = make_inputs(O, range(S), n)
    x[:, 3000:5000] *= 5.0                                    # a level step so that the AGC gain moves
    po = O.eq_params_bench(0.2)
    po.agcEnabled = 1 if agc else 0
    eng = amd.BatchedEngine(S, block_size=quantum, max_ir_len=512, max_blocks_per_call=5, call_mode=amd.CPQ_CALLS_ANY)
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL if mode == "sequential" else amd.CPQ_EQ_MODE_AUTO)
    state = [np.zeros(168) for _ in range(S)]
    worst, pos = 0.0, 0
    for k, m in enumerate(call_sizes):
        if k in gains_db or k == 0:
            po.totalGainDb = gains_db.get(k, po.totalGainDb)
            eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_eq(po, amd.eq_params_default()))
        seg = np.ascontiguousarray(x[:, pos:pos + m])
        y = eng.eq_process(seg)
        for s in range(S):
            yl, yr, state[s] = O.eq_process_stereo(seg[2 * s], seg[2 * s + 1], po, block=quantum, state=state[s])
            worst = max(worst, np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max())
        pos += m
    print(f"ragged EQ agc={agc} {mode}: max abs diff {worst:.3e}")
    assert worst <= 1e-12
    eng.close()


@pytest.mark.parametrize("quantum", [480, 441])
@pytest.mark.parametrize("agc", [False, True])
def test_eq_bypass_fade_and_band_reset_across_short_callbacks(amd, oracle, quantum, agc):
    """The EQ bypass state machine (5 ms fade through the basic path, frozen state, state clear + fade-in on release) and a
    band reset with calls of k x quantum + r samples: the short last callback draws r values of the fade ramp, like any
    other callback (EqWithBypass in tests/oracle_lib.py, driven callback by callback with the same cuts)."""
    O = oracle
    S = 2
    call_sizes = [quantum * 2 + 100, quantum, 77, quantum * 3, quantum + 1, quantum * 2 + quantum // 2, 5, quantum * 3 - 1,
                  quantum * 2, quantum * 3 + 17]
    req = [[0, 1, 1, 1, 0, 0, 1, 0, 0, 0],
           [0, 0, 1, 0, 1, 1, 0, 0, 1, 0]]
    po = O.eq_params_bench(0.2)
    po.bands[5].gain = 0.0                                 # a flat band: drops out of the basic path's band nodes
    po.totalGainDb = -2.0
    po.agcEnabled = int(agc)
    n = sum(call_sizes)
    x = make_inputs(O, range(S), n)
    eng = amd.BatchedEngine(S, block_size=quantum, max_ir_len=512, max_blocks_per_call=4, call_mode=amd.CPQ_CALLS_ANY)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_eq(po, amd.eq_params_default()))
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL)
    outs, pos = [], 0
    for k, m in enumerate(call_sizes):
        for s in range(S):
            eng.set_eq_bypass(s, req[s][k])
        if k == 7:
            eng.request_band_reset(1, 0x00000F0F)
        outs.append(eng.eq_process(np.ascontiguousarray(x[:, pos:pos + m])))
        pos += m
    y = np.concatenate(outs, axis=1)
    eng.close()
    for s in range(S):
        ref = O.EqWithBypass(O.EqParams.from_buffer_copy(po), 48000.0, quantum)
        rl, rr, pos = [], [], 0
        for k, m in enumerate(call_sizes):
            if k == 7 and s == 1:
                ref.request_band_reset(0x00000F0F)
            o = 0
            while o < m:
                ln = min(quantum, m - o)
                a, b = ref.callback(x[2 * s, pos + o:pos + o + ln].copy(), x[2 * s + 1, pos + o:pos + o + ln].copy(), bool(req[s][k]))
                rl.append(a)
                rr.append(b)
                o += ln
            pos += m
        err = max(np.abs(y[2 * s] - np.concatenate(rl)).max(), np.abs(y[2 * s + 1] - np.concatenate(rr)).max())
        print(f"ragged bypass quantum {quantum} agc={agc} stream {s}: max abs diff {err:.3e}")
        assert err <= 1e-13, (s, err)


@pytest.mark.parametrize("quantum,taps", [(441, 20000), (480, 131072)])
def test_whole_chain_at_arbitrary_quantum(amd, oracle, quantum, taps):
    """cpq_engine_process_block with CPQ_CALLS_ANY: convolver (Add + Get per callback of `quantum`, the last one of a call
    shorter) followed by the EQ on the same callbacks, with the AGC on one stream and a total-gain ramp on the other."""
    O = oracle
    S = 2
    call_sizes = [quantum * 3 + 123, quantum * 2, 59, quantum * 4, quantum + quantum // 3, quantum * 4 - 1, 1, quantum * 3]
    n = sum(call_sizes)
    irs = [O.gen_ir(taps, stream=20 + c // 2, channel=c % 2) for c in range(2 * S)]
    x = make_inputs(O, range(20, 20 + S), n)
    pos_params = [O.eq_params_bench(0.2), O.eq_params_bench(0.2)]
    pos_params[0].agcEnabled = 1
    pos_params[1].totalGainDb = -4.0
    eng = amd.BatchedEngine(S, block_size=quantum, max_ir_len=taps, max_blocks_per_call=4, call_mode=amd.CPQ_CALLS_ANY)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
        eng.set_eq_params(s, _copy_eq(pos_params[s], amd.eq_params_default()))
    outs, pos = [], 0
    for k, m in enumerate(call_sizes):
        if k == 3:                                          # a gain change: the ramp runs through ragged calls
            pos_params[1].totalGainDb = 2.5
            eng.set_eq_params(1, _copy_eq(pos_params[1], amd.eq_params_default()))
        outs.append(eng.process(np.ascontiguousarray(x[:, pos:pos + m])))
        pos += m
    y = np.concatenate(outs, axis=1)
    eng.close()
    worst = 0.0
    for s in range(S):
        conv = [oracle_calls(O, irs[2 * s + ch], x[2 * s + ch], quantum, call_sizes)[0] for ch in range(2)]
        state = np.zeros(168)
        po = O.eq_params_bench(0.2)
        po.agcEnabled = 1 if s == 0 else 0
        po.totalGainDb = 0.0 if s == 0 else -4.0
        pos = 0
        for k, m in enumerate(call_sizes):
            if k == 3 and s == 1:
                po.totalGainDb = 2.5
            yl, yr, state = O.eq_process_stereo(conv[0][pos:pos + m], conv[1][pos:pos + m], po, block=quantum, state=state)
            worst = max(worst, rms(y[2 * s, pos:pos + m] - yl), rms(y[2 * s + 1, pos:pos + m] - yr))
            pos += m
    print(f"whole chain quantum {quantum}, {taps} taps: worst rms err per call {worst:.3e}")
    assert worst <= 1e-12


@pytest.mark.parametrize("quantum,mix,peak", [(480, 0.35, 100), (441, 1.0, 0), (96, 0.0, 33)])
def test_processor_level_at_arbitrary_quantum(amd, oracle, quantum, mix, peak):
    """ConvolverProcessor::process steady state (dry delay line of getLatency() + irPeakLatency, equal-power mix, wet
    sanitise) around the convolver at a call quantum that is not a power of two and with ragged calls: the algorithm
    latency is the layer-0 partition nextPow2(max(quantum, 64)), not the quantum."""
    O = oracle
    L = O.lib()
    S, taps = 2, 9000
    call_sizes = [quantum * 3 + 11, quantum, 7, quantum * 4, quantum * 2 + quantum // 2, 1, quantum * 4 - 3, quantum * 3]
    n = sum(call_sizes)
    irs = [O.gen_ir(taps, stream=30 + c // 2, channel=c % 2) for c in range(2 * S)]
    x = make_inputs(O, range(30, 30 + S), n)
    eng = amd.BatchedEngine(S, block_size=quantum, max_ir_len=taps, max_blocks_per_call=4, call_mode=amd.CPQ_CALLS_ANY)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=mix, ir_peak_latency=peak)
    p0 = max(64, 1 << (quantum - 1).bit_length())
    assert eng.convproc_delay(0) == p0 + peak
    outs, pos = [], 0
    for m in call_sizes:
        outs.append(eng.convproc_process(np.ascontiguousarray(x[:, pos:pos + m])))
        pos += m
    y = np.concatenate(outs, axis=1)
    eng.close()
    mixd = float(np.float32(mix))
    delay = p0 + peak
    for c in range(2 * S):
        dry = np.zeros(n)
        dry[delay:] = x[c][:n - delay]
        if not (mixd > 0.001):
            ref = dry
        else:
            wet, _ = oracle_calls(O, irs[c], x[c], quantum, call_sizes)
            wet_g = L.orc_equal_power_sin(mixd) * 1.0
            dry_g = L.orc_equal_power_sin(1.0 - mixd) if mixd < 0.999 else 0.0
            ref = (wet * wet_g) + (dry * dry_g)
        err = rms(y[c] - ref)
        assert err <= 1e-13, (c, err)


@pytest.mark.parametrize("mode", ["sequential", "auto"])
def test_output_filter_with_ragged_calls(amd, oracle, mode):
    """OutputFilter (three DF-II-T sections, a stateless-per-callback cascade) over calls of any length: the cascade kernels
    take any sample count; the filter state carries across the cuts."""
    O = oracle
    S, quantum = 2, 441
    call_sizes = [441 * 3 + 17, 441, 5, 441 * 4, 1000, 1, 441 * 2 + 63, 441 * 4 - 1]
    n = sum(call_sizes)
    x = make_inputs(O, range(S), n)
    q = O.outfilter_design(1, 1, 1, 2, 48000.0)
    eng = amd.BatchedEngine(S, block_size=quantum, max_ir_len=512, max_blocks_per_call=4, call_mode=amd.CPQ_CALLS_ANY)
    eng.set_outfilter_params(amd.CPQ_ALL_STREAMS, 1, 1, 1, 2)
    eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL if mode == "sequential" else amd.CPQ_EQ_MODE_AUTO)
    outs, pos = [], 0
    for m in call_sizes:
        outs.append(eng.outfilter_process(np.ascontiguousarray(x[:, pos:pos + m])))
        pos += m
    y = np.concatenate(outs, axis=1)
    eng.close()
    for s in range(S):
        yl, yr, _ = O.outfilter_process_stereo(x[2 * s], x[2 * s + 1], q)
        worst = max(np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max())
        assert worst <= (0.0 if mode == "sequential" else 5e-12), worst


def test_dspcore_chain_at_arbitrary_quantum(amd, oracle):
    """DSPCore's order Conv (processor level: dry delay + equal-power mix) -> EQ (AGC on) -> OutputFilter through
    cpq_engine_process_block at a 480-sample quantum with ragged calls."""
    O = oracle
    L = O.lib()
    S, quantum, taps, mix = 2, 480, 7000, 0.6
    call_sizes = [480 * 3 + 250, 480, 3, 480 * 4, 480 * 2 + 479, 480 * 4 - 7, 480 * 3]
    n = sum(call_sizes)
    irs = [O.gen_ir(taps, stream=40 + c // 2, channel=c % 2) for c in range(2 * S)]
    x = make_inputs(O, range(40, 40 + S), n)
    po = O.eq_params_bench(0.2)
    po.agcEnabled = 1
    q = O.outfilter_design(0, 1, 0, 1, 48000.0)
    eng = amd.BatchedEngine(S, block_size=quantum, max_ir_len=taps, max_blocks_per_call=4, call_mode=amd.CPQ_CALLS_ANY)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_eq(po, amd.eq_params_default()))
    eng.set_convproc_params(amd.CPQ_ALL_STREAMS, mix=mix)
    eng.set_conv_level(amd.CPQ_LEVEL_PROCESSOR)
    eng.set_outfilter_params(amd.CPQ_ALL_STREAMS, 0, 1, 0, 1)
    eng.enable_output_filter(True)
    outs, pos = [], 0
    for m in call_sizes:
        outs.append(eng.process(np.ascontiguousarray(x[:, pos:pos + m])))
        pos += m
    y = np.concatenate(outs, axis=1)
    eng.close()
    mixd = float(np.float32(mix))
    wet_g, dry_g = L.orc_equal_power_sin(mixd) * 1.0, L.orc_equal_power_sin(1.0 - mixd)
    for s in range(S):
        w = []
        for ch in range(2):
            wet, _ = oracle_calls(O, irs[2 * s + ch], x[2 * s + ch], quantum, call_sizes)
            dry = np.zeros(n)
            dry[512:] = x[2 * s + ch][:n - 512]
            w.append((wet * wet_g) + (dry * dry_g))
        state, pos, el, er = np.zeros(168), 0, [], []
        for m in call_sizes:
            a, b, state = O.eq_process_stereo(w[0][pos:pos + m], w[1][pos:pos + m], po, block=quantum, state=state)
            el.append(a)
            er.append(b)
            pos += m
        fl, fr, _ = O.outfilter_process_stereo(np.concatenate(el), np.concatenate(er), q)
        assert rms(y[2 * s] - fl) <= 1e-12 and rms(y[2 * s + 1] - fr) <= 1e-12


@pytest.mark.parametrize("seed", list(range(101, 113)))
def test_whole_chain_random_transitions_at_arbitrary_quantum(amd, oracle, seed):
    """tests/test_gpu_parity.py::test_whole_chain_random_transition_sequence under CPQ_CALLS_ANY: a call quantum of 480, 441
    or 96 samples, ragged calls, and per call and stream a moving mix, IR peak latency (latency cross-fade of the dry
    path), EQ bypass request, band resets and total-gain changes; every third seed runs EQ -> conv with trim gains.
    Chain: processor-level convolver -> EQ -> output filter -> make-up gain, against the per-callback restatements
    (ConvProcStream with getLatency() = nextPow2(max(quantum, 64)), EqWithBypass, the output-filter oracle) cut into the
    same callbacks: `quantum` samples each, the last one of a call shorter."""
    O = oracle
    rng = np.random.default_rng(seed)
    quantum = int(rng.choice([480, 441, 96] if seed < 1000 else [480, 441, 96, 37, 512, 1000, 64, 250]))      # tools/soak_gpu.py: seeds >= 1000
    p0 = max(64, 1 << (quantum - 1).bit_length())
    S, max_blocks = 2, (3 if seed < 1000 else int(rng.integers(1, 5)))
    call_sizes = []
    for _ in range(14):
        r = rng.random()
        call_sizes.append(int(quantum * max_blocks if r < 0.2 else (rng.integers(1, 20) if r < 0.3 else rng.integers(1, quantum * max_blocks + 1))))
    n = sum(call_sizes)
    irs = [O.gen_ir(int(rng.integers(600, 4000)), stream=50 + s, channel=ch) for s in range(S) for ch in range(2)]
    ir_len = max(len(h) for h in irs)
    irs = [np.concatenate([h, np.zeros(ir_len - len(h))]) for h in irs]
    x = make_inputs(O, range(50, 50 + S), n)
    po = O.eq_params_bench(0.2)
    po.bands[5].gain = 0.0
    q = O.outfilter_design(0, 1, 0, 1, 48000.0)
    makeup, trim = [1.0, 0.7], [0.5, 1.7]
    order = amd.CPQ_ORDER_EQ_THEN_CONV if seed % 3 == 0 else amd.CPQ_ORDER_CONV_THEN_EQ
    mix = [float(rng.uniform(0.2, 1.0)) for _ in range(S)]
    peak = [int(rng.integers(0, 1500)) for _ in range(S)]
    byp = [False] * S
    eng = amd.BatchedEngine(S, block_size=quantum, max_ir_len=ir_len, max_blocks_per_call=max_blocks, call_mode=amd.CPQ_CALLS_ANY)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
        eng.set_convproc_params(s, mix=mix[s], ir_peak_latency=peak[s])
        eng.set_gains(s, trim[s], makeup[s])
    eng.set_order(order)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_eq(po, amd.eq_params_default()))
    if seed % 2:
        eng.set_eq_mode(amd.CPQ_EQ_MODE_SEQUENTIAL)
    eng.set_conv_level(amd.CPQ_LEVEL_PROCESSOR)
    eng.set_outfilter_params(amd.CPQ_ALL_STREAMS, 0, 1, 0, 1)
    eng.enable_output_filter(True)
    conv = [None] * S
    eqs = [O.EqWithBypass(O.EqParams.from_buffer_copy(po), 48000.0, quantum) for _ in range(S)]
    ofs = [None] * S
    ref = np.empty_like(x)
    lazy = seed % 2 == 0
    last_proc, last_byp = [None] * S, [None] * S
    outs, pos = [], 0
    for k, m in enumerate(call_sizes):
        for s in range(S):
            if k > 0 and rng.random() < 0.3:
                mix[s] = float(rng.uniform(0.05, 1.0))
            if k > 0 and rng.random() < 0.3:
                peak[s] = int(rng.integers(0, 1500)) if rng.random() < 0.8 else peak[s] + 1
            if rng.random() < 0.25:
                byp[s] = not byp[s]
            # a host sets a parameter when it changes, not before every block (even seeds; odd seeds set everything every call)
            if not lazy or k == 0 or (mix[s], peak[s]) != last_proc[s]:
                eng.set_convproc_params(s, mix=mix[s], ir_peak_latency=peak[s])
                last_proc[s] = (mix[s], peak[s])
            if not lazy or k == 0 or byp[s] != last_byp[s]:
                eng.set_eq_bypass(s, byp[s])
                last_byp[s] = byp[s]
            if k == 0:
                eqs[s].sync(byp[s])
            if rng.random() < 0.2:
                mask = 0xFFFFFFFF if rng.random() < 0.4 else int(rng.integers(1, 1 << 20))
                eng.request_band_reset(s, mask)
                eqs[s].request_band_reset(mask)
            if rng.random() < 0.15:
                g = float(rng.uniform(-9.0, 3.0))
                pa = _copy_eq(po, amd.eq_params_default())
                pa.total_gain_db = g
                eng.set_eq_params(s, pa)
                eqs[s].set_total_gain_db(g, before_first_block=(k == 0))
        outs.append(eng.process(np.ascontiguousarray(x[:, pos:pos + m])))
        for s in range(S):
            o = 0
            while o < m:
                ln = min(quantum, m - o)
                a, b = x[2 * s, pos + o:pos + o + ln].copy(), x[2 * s + 1, pos + o:pos + o + ln].copy()
                if conv[s] is None:
                    conv[s] = O.ConvProcStream(irs[2 * s], irs[2 * s + 1], quantum, mix[s], peak[s], latency=p0)
                if order == amd.CPQ_ORDER_CONV_THEN_EQ:
                    a, b = conv[s].callback(a, b, mix[s], peak[s])
                    a, b = eqs[s].callback(np.ascontiguousarray(a), np.ascontiguousarray(b), byp[s])
                else:
                    a, b = eqs[s].callback(a, b, byp[s])
                    a, b = a * trim[s], b * trim[s]
                    a, b = conv[s].callback(a, b, mix[s], peak[s])
                a, b, ofs[s] = O.outfilter_process_stereo(np.ascontiguousarray(a), np.ascontiguousarray(b), q, ofs[s])
                ref[2 * s, pos + o:pos + o + ln], ref[2 * s + 1, pos + o:pos + o + ln] = a * makeup[s], b * makeup[s]
                o += ln
        pos += m
    y = np.concatenate(outs, axis=1)
    eng.close()
    for s in range(S):
        err = np.abs(y[2 * s:2 * s + 2] - ref[2 * s:2 * s + 2]).max()
        assert err <= 1e-12, (seed, quantum, s, err)


def test_latency_cross_fade_starting_in_a_one_sample_call(amd, oracle):
    """A change of the IR peak latency starts the 20 ms cross-fade of the dry path; when the call it starts in carries a
    single sample the fade contributes ONE value to that call (the cross-fade buffer must exist for it), and the mix ramp
    that starts with it likewise."""
    O = oracle
    quantum, taps = 96, 3000
    call_sizes = [96 * 2, 1, 96 * 3, 2, 96 * 3 - 1, 96 * 3, 96 * 3, 96 * 3, 96 * 3, 96 * 3]
    peaks = [700, 150, 150, 900, 900, 900, 900, 900, 900, 900]
    mixes = [0.6, 0.25, 0.25, 0.25, 0.8, 0.8, 0.8, 0.8, 0.8, 0.8]
    n = sum(call_sizes)
    irs = [O.gen_ir(taps, stream=60, channel=ch) for ch in range(2)]
    x = make_inputs(O, [60], n)
    eng = amd.BatchedEngine(1, block_size=quantum, max_ir_len=taps, max_blocks_per_call=3, call_mode=amd.CPQ_CALLS_ANY)
    eng.set_impulse(0, irs[0], irs[1])
    eng.set_convproc_params(0, mix=mixes[0], ir_peak_latency=peaks[0])
    ref = O.ConvProcStream(irs[0], irs[1], quantum, mixes[0], peaks[0], latency=128)
    outs, rl, rr, pos = [], [], [], 0
    for k, m in enumerate(call_sizes):
        eng.set_convproc_params(0, mix=mixes[k], ir_peak_latency=peaks[k])
        outs.append(eng.convproc_process(np.ascontiguousarray(x[:, pos:pos + m])))
        o = 0
        while o < m:
            ln = min(quantum, m - o)
            a, b = ref.callback(x[0, pos + o:pos + o + ln].copy(), x[1, pos + o:pos + o + ln].copy(), mixes[k], peaks[k])
            rl.append(a)
            rr.append(b)
            o += ln
        pos += m
    y = np.concatenate(outs, axis=1)
    eng.close()
    assert np.abs(y[0] - np.concatenate(rl)).max() <= 1e-13 and np.abs(y[1] - np.concatenate(rr)).max() <= 1e-13
