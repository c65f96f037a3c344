"""Chained spans of the time-parallel EQ cascade (svf_kernels.hip: k_svf_cascade_tpv<8, true>): engines with fewer channels
than the device holds workgroups of the span kernel deal the (span, channel) pairs of a call to the workgroups; a band's state
is handed from span to span inside the launch.  Against the oracle's EQProcessor restatement (processBandStereo /
processBand, src/eqprocessor/EQProcessor.Processing.cpp:128-276, serial structure :1231-1253)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

B = 512
SPAN = 8192


@pytest.fixture(scope="module")
def amd():
    import convopeq_amd
    return convopeq_amd


def _inputs(O, S, n, start=0):
    return np.stack([O.gen_pcm(n, stream=c // 2, channel=c % 2, start=start) for c in range(2 * S)])


def _copy_params(po, pa):
    for i in range(20):
        b, o = pa.bands[i], po.bands[i]
        b.frequency, b.gain, b.q, b.enabled, b.type, b.channel_mode = o.frequency, o.gain, o.q, o.enabled, o.type, o.channelMode
    pa.total_gain_db, pa.agc_enabled, pa.nonlinear_saturation, pa.filter_structure = po.totalGainDb, po.agcEnabled, po.nonlinearSaturation, po.filterStructure
    return pa


def _oracle(O, x, po, S):
    ref = np.empty_like(x)
    for s in range(S):
        ref[2 * s], ref[2 * s + 1], _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], po)
    return ref


@pytest.mark.parametrize("sat", [0.0, 0.2])
def test_chained_spans_long_calls_with_ragged_remainder(amd, oracle, sat):
    """Three calls of 259 blocks (16 spans of 8192 samples, dealt to 64 workgroups as 4 x 16 tasks, + a remainder of one wave
    x 1024 samples + one 512-sample span) on 2 streams: the states pass from span to span through the hand-over granules,
    from call to call through the state array; clean signal."""
    O = oracle
    S, T = 2, 259
    x = _inputs(O, S, 3 * T * B)
    po = O.eq_params_bench(sat)
    po.bands[6].channelMode = 1
    po.bands[9].type = 3
    po.totalGainDb = 0.75
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    y = np.concatenate([eng.eq_process(x[:, o:o + T * B]) for o in range(0, x.shape[1], T * B)], axis=1)
    launches, gave_up = eng.eq_chain_status()
    assert launches == 3 and gave_up == 0
    worst = np.abs(y - _oracle(O, x, po, S)).max()
    print("chained spans sat", sat, "max abs diff", worst)
    assert np.all(np.isfinite(y)) and worst <= 1e-13
    eng.close()


def test_chained_spans_guarded_spans_and_out_of_range_states(amd, oracle):
    """Spans that leave the fast path inside a chained launch: a NaN and an Inf in different spans and channels (input out of
    range: the span runs on the guarded path, its neighbours on the fast path), a stretch of hot signal (general output stage),
    and a 3e10 sample 20 samples before the end of a span -- that span is guarded by its input and leaves band states of
    1e8 (low bands: in range) ... 4e9 (mid bands: out of range), so the NEXT span receives an out-of-range start state in the
    middle of its band loop, has published the bands before it from good states, and is run again on the guarded path."""
    O = oracle
    S, T = 2, 128 + 3            # 8 spans + a remainder
    n = T * B
    x = _inputs(O, S, 3 * n)
    x[:, 20000:45000] *= 64.0                       # hot: |y| >= 4.5 in many waves
    x[0, n + 2 * SPAN + 100] = np.nan               # call 2, span 2
    x[3, n + 5 * SPAN + 4000] = np.inf              # call 2, span 5
    x[1, 2 * n + 3 * SPAN + SPAN - 20] = 3.0e10     # call 3, end of span 3
    x[2, 2 * n + 6 * SPAN + 17] = -1.0e300          # call 3, span 6: sanitised to silence by the output guard
    po = O.eq_params_bench(0.2)
    po.bands[3].channelMode = 2
    po.totalGainDb = -1.25
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    y = np.concatenate([eng.eq_process(x[:, o:o + n]) for o in range(0, 3 * n, n)], axis=1)
    launches, gave_up = eng.eq_chain_status()
    assert launches == 3 and gave_up == 0
    ref = _oracle(O, x, po, S)
    assert np.all(np.isfinite(y))
    # channel 1 behind the 3e10 sample: states of 1e9 ... 1e10 decay through the fast path of the later spans, whose rounding
    # is relative to them
    big = 2 * n + 3 * SPAN + SPAN - 20
    d = np.abs(y - ref)
    print("guarded / out-of-range: max abs diff", d[[0, 2, 3]].max(), d[1, :big].max(), "behind the 3e10 sample", d[1, big:].max())
    assert d[[0, 2, 3]].max() <= 1e-12 and d[1, :big].max() <= 1e-12
    assert d[1, big:].max() <= 1e-6 and d[1, big + 3 * SPAN:].max() <= 1e-9
    eng.close()


@pytest.mark.parametrize("S", [1, 5, 24])
def test_chained_spans_stream_counts(amd, oracle, S):
    """2, 10 and 48 channels x 6 spans per call: more tasks than one round of workgroups only at the last; per-stream
    parameters, so that a workgroup that changes channel between tasks must reload its tables."""
    O = oracle
    T = 96
    x = _inputs(O, S, 2 * T * B)
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
    pos = []
    for s in range(S):
        po = O.eq_params_bench(0.2 if s % 2 == 0 else 0.0)
        for i in range(20):
            po.bands[i].gain = po.bands[i].gain * (1.0 + 0.1 * (s % 7))
        po.bands[(3 * s) % 20].type = 3 + (s % 2)
        pos.append(po)
        eng.set_eq_params(s, _copy_params(po, amd.eq_params_default()))
    y = np.concatenate([eng.eq_process(x[:, o:o + T * B]) for o in range(0, x.shape[1], T * B)], axis=1)
    launches, gave_up = eng.eq_chain_status()
    assert launches == 2 and gave_up == 0
    worst = 0.0
    for s in range(S):
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], pos[s])
        worst = max(worst, np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max())
    print("chained spans, streams", S, "max abs diff", worst)
    assert worst <= 1e-13
    eng.close()


def test_chained_spans_more_channels_than_workgroup_slots(amd, oracle):
    """300 streams = 600 channels on a device that holds 512 workgroups of the span kernel: one workgroup per channel would run
    a second round of 88 -- the engine chains instead (engine_core.cpp: the rule), tasks outnumber the workgroups from the first
    span on and a workgroup's next task is always another channel.  Four of the streams against the oracle."""
    O = oracle
    S, T = 300, 64
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
    check = [0, 1, 255, 256, 299]
    rng = np.random.default_rng(7)
    x = 0.25 * (2.0 * rng.random((2 * S, T * B)) - 1.0)
    for s in check:
        x[2 * s], x[2 * s + 1] = O.gen_pcm(T * B, stream=s, channel=0), O.gen_pcm(T * B, stream=s, channel=1)
    pos = {}
    for s in range(S):
        po = O.eq_params_bench(0.2)
        po.bands[s % 20].gain = 0.5 * (s % 9) - 2.0
        if s in check:
            pos[s] = po
        eng.set_eq_params(s, _copy_params(po, amd.eq_params_default()))
    y = eng.eq_process(x)
    launches, gave_up = eng.eq_chain_status()
    import torch
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    assert gave_up == 0
    if n_cu == 256:
        assert launches == 1        # chained (600 channels on 512 slots): the launch advanced the generation
    worst = 0.0
    for s in check:
        yl, yr, _ = O.eq_process_stereo(x[2 * s], x[2 * s + 1], pos[s])
        worst = max(worst, np.abs(y[2 * s] - yl).max(), np.abs(y[2 * s + 1] - yr).max())
    print("chained spans, 300 streams: max abs diff", worst, "launches", launches)
    assert worst <= 1e-13
    eng.close()


def test_two_engines_on_two_host_threads_share_the_device(amd, oracle):
    """Row (e) of SURVEY.md section 8 in one process: two engines on device 0, each driven by its own host thread (the loop of
    tests/examples/multi_device_host.cpp with both handles on one device), real IRs + PCM through convolver and EQ, several
    calls in flight on two streams at once -- both EQ launches chain their spans and compete for the same CUs.  Each engine's
    output must be bit-equal to the same engine run alone (streams are independent: src/ConvolverProcessor.h:669)."""
    O = oracle
    S, T, L, calls = 3, 64, 20000, 4
    n = T * B

    def make(seed):
        irs = [O.gen_ir(L, stream=seed + c // 2, channel=c % 2) for c in range(2 * S)]
        x = np.stack([O.gen_pcm(calls * n, stream=seed + c // 2, channel=c % 2) for c in range(2 * S)])
        po = O.eq_params_bench(0.2)
        po.bands[5].gain = 1.0 + seed
        eng = amd.BatchedEngine(S, block_size=B, max_ir_len=L, max_blocks_per_call=T)
        eng.prepare_to_play(48000.0, n)
        for s in range(S):
            eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
        eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
        return eng, irs, x, po

    def run(eng, x, out, idx):
        out[idx] = np.concatenate([eng.process(x[:, o:o + n]) for o in range(0, calls * n, n)], axis=1)

    alone = [None, None]
    for i, seed in enumerate((0, 7)):
        eng, irs, x, po = make(seed)
        run(eng, x, alone, i)
        if i == 0:          # the first one also against the oracle
            ref = np.empty_like(x)
            for c in range(2 * S):
                nuc = O.Nuc()
                nuc.set_impulse(irs[c], B)
                ref[c] = nuc.run(x[c], B)
            ref = _oracle(O, ref, po, S)
            assert np.sqrt(np.mean((alone[0] - ref) ** 2)) <= 1e-13
        eng.close()
    both = [None, None]
    engines = [make(0), make(7)]
    threads = [threading.Thread(target=run, args=(engines[i][0], engines[i][2], both, i)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(2):
        launches, gave_up = engines[i][0].eq_chain_status()
        assert launches == calls and gave_up == 0
        assert both[i] is not None and np.array_equal(both[i], alone[i])
        engines[i][0].close()


def test_chained_spans_output_filter_and_whole_chain(amd, oracle):
    """The OutputFilter's three DF-II-T sections (src/OutputFilter.cpp:143-165) run on the same cascade kernels: calls of
    five spans + a ragged remainder on 2 streams chain their spans too (band classes: linear sections only); then conv -> EQ ->
    output filter in one call (DSPCore order), EQ and filter both chained, against the oracle's chain."""
    O = oracle
    S, T = 2, 85                     # 43520 samples = 5 spans + 2560
    n = T * B
    x = _inputs(O, S, 3 * n)
    q = O.outfilter_design(0, 1, 0, 1, 48000.0)
    eng = amd.BatchedEngine(S, max_ir_len=512, max_blocks_per_call=T)
    eng.set_outfilter_params(amd.CPQ_ALL_STREAMS, 0, 1, 0, 1)
    y = np.concatenate([eng.outfilter_process(x[:, o:o + n]) for o in range(0, 3 * n, n)], axis=1)
    launches, gave_up = eng.eq_chain_status()
    assert launches == 3 and gave_up == 0
    err2 = 0.0
    for s in range(S):
        yl, yr, _ = O.outfilter_process_stereo(x[2 * s], x[2 * s + 1], q)
        err2 += np.sum((y[2 * s] - yl) ** 2) + np.sum((y[2 * s + 1] - yr) ** 2)
    assert np.sqrt(err2 / y.size) <= 1e-12          # (its 20 Hz high-pass: any fp64 evaluation order is ~2e-13 from exact)
    eng.close()

    L = 6000
    irs = [O.gen_ir(L, stream=c // 2, channel=c % 2) for c in range(2 * S)]
    po = O.eq_params_bench(0.2)
    eng = amd.BatchedEngine(S, max_ir_len=L, max_blocks_per_call=T)
    for s in range(S):
        eng.set_impulse(s, irs[2 * s], irs[2 * s + 1])
    eng.set_eq_params(amd.CPQ_ALL_STREAMS, _copy_params(po, amd.eq_params_default()))
    eng.set_outfilter_params(amd.CPQ_ALL_STREAMS, 0, 1, 0, 1)
    eng.enable_output_filter(True)
    y = np.concatenate([eng.process(x[:, o:o + n]) for o in range(0, 2 * n, n)], axis=1)
    launches, gave_up = eng.eq_chain_status()
    assert launches == 4 and gave_up == 0           # two calls x (EQ + output filter)
    for s in range(S):
        w = []
        for ch in range(2):
            nuc = O.Nuc()
            nuc.set_impulse(irs[2 * s + ch], B)
            w.append(nuc.run(x[2 * s + ch, :2 * n], B))
        el, er, _ = O.eq_process_stereo(w[0], w[1], po)
        fl, fr, _ = O.outfilter_process_stereo(el, er, q)
        assert np.sqrt(np.mean((y[2 * s] - fl) ** 2)) <= 1e-12 and np.sqrt(np.mean((y[2 * s + 1] - fr) ** 2)) <= 1e-12
    eng.close()
