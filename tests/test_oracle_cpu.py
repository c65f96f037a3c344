"""CPU tests of the oracle (no GPU): pins against the reference's stand-alone headers (oracle/_ref or the committed
fixtures generated from it), against the observations SURVEY.md recorded from the running reference, and against
independent mathematics (long-double direct form, scipy)."""
import json
import os

import numpy as np
import pytest
from scipy.signal import fftconvolve

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def rms(a):
    return float(np.sqrt(np.mean(np.square(a))))


def test_fast_tanh_matches_reference_header_fixture(oracle):
    """A15: oracle restatement == reference FastTanhApprox.h outputs (bit-exact), scalar and SSE2 semantics."""
    g = load("fasttanh_ref.json")
    L = oracle.lib()
    for xs, s, v in zip(g["x"], g["scalar"], g["v128"]):
        x = float(xs)
        assert L.orc_fast_tanh_scalar(x) == float(s), x
        got = L.orc_fast_tanh_v128(x)
        assert got == float(v) or (np.isnan(got) and np.isnan(float(v))), x
    # the vector path clamps the argument instead of clipping the result
    assert abs(L.orc_fast_tanh_v128(10.0) - load("survey_observations.json")["fast_tanh_v128_at_clip"]) < 1e-5
    assert L.orc_fast_tanh_scalar(10.0) == 1.0


def test_fast_tanh_against_live_reference_probe(oracle):
    R = oracle.ref_probe()
    if R is None:
        pytest.skip("oracle/_ref not built and reference tree absent")
    L = oracle.lib()
    rng = np.random.default_rng(7)
    for x in np.concatenate([rng.uniform(-8, 8, 2000), rng.normal(0, 1e-3, 200)]):
        assert L.orc_fast_tanh_scalar(float(x)) == R.ref_fast_tanh_scalar(float(x))
        assert L.orc_fast_tanh_v128(float(x)) == R.ref_fast_tanh_v128(float(x))


def test_eq_params_default_matches_reference_header(oracle):
    g = load("eq_params_default_ref.json")
    p = oracle.eq_params_default()
    for b, ref in zip(p.bands, g["bands"]):
        assert [b.frequency, b.gain, b.q, b.enabled, b.type, b.channelMode] == ref
    assert p.totalGainDb == g["totalGainDb"] and p.agcEnabled == g["agcEnabled"]
    assert p.nonlinearSaturation == np.float32(g["nonlinearSaturation"]) and p.filterStructure == g["filterStructure"]


def test_svf_design_known_answer_from_survey(oracle):
    k = load("survey_observations.json")["svf_known_answer"]
    c = oracle.svf_design(k["type"], k["freq"], k["gain_db"], k["q"], k["sr"])
    for name in ("a1", "a2", "a3", "m0", "m1", "m2"):
        assert getattr(c, name) == float(k[name]), name


def test_layer_plan_matches_survey_observations(oracle):
    s = load("survey_observations.json")
    lp = s["layer_plan_48k_blk512_131072"]
    p = oracle.plan(131072, 512)
    assert p.numLayers == 2 and p.len[0] == lp["l0_taps"] and p.numPartsIR[0] == lp["l0_parts"]
    assert p.partSize[1] == lp["l1_part"] and p.numPartsIR[1] == lp["l1_parts"]
    assert p.partsPerCallback[1] == lp["l1_parts_per_callback"]
    assert abs(p.gain[1] - s["tail_gains_default"]["g1"]) < 1e-12
    p3 = oracle.plan(524288, 512)
    assert p3.numLayers == 3 and abs(p3.gain[2] - s["tail_gains_default"]["g2"]) < 1e-12
    for row in s["lags"]:
        p = oracle.plan(row["ir_len"], row["block"])
        assert p.ltiValid == 1
        if "lag" in row:
            assert [p.lag[i] for i in range(1, p.numLayers)] == row["lag"]
        else:
            assert p.lag[p.numLayers - 1] == row["lag_last"]
    for blk in s["lti_invalid_blocks"]:
        assert oracle.plan(131072, blk).ltiValid == 0


def test_config1_is_exact_zero_latency_convolution(oracle):
    """SURVEY finding 3: irLen <= L0 and block == partSize -> exact linear convolution, lag 0, getLatency 512."""
    h = oracle.gen_ir(4096)
    x = oracle.gen_pcm(512 * 40)
    c = oracle.Nuc()
    assert c.set_impulse(h, 512)
    assert c.latency() == 512
    y = c.run(x, 512)
    assert rms(y - fftconvolve(x, h)[:len(x)]) < 2e-16
    idx = np.array([0, 1, 511, 512, 4095, 4096, 9999, len(x) - 1])
    assert np.abs(y[idx] - oracle.direct_conv_at(x, h, idx)).max() < 1e-15


@pytest.mark.parametrize("ir_len,block,nblocks", [(131072, 512, 560), (131072, 128, 1600), (131072, 256, 900),
                                                   (40000, 512, 200), (524288, 512, 1300)])
def test_schedule_emulation_equals_heff_convolution(oracle, ir_len, block, nblocks):
    """A6: the Add/Get schedule emulation is LTI-equivalent to x * h_eff whenever ltiValid."""
    h = oracle.gen_ir(ir_len)
    x = oracle.gen_pcm(block * nblocks)
    c = oracle.Nuc()
    assert c.set_impulse(h, block)
    y = c.run(x, block)
    he = oracle.heff(h, block)
    assert rms(y - fftconvolve(x, he)[:len(x)]) < 5e-15
    assert rms(y - fftconvolve(x, h)[:len(x)]) > 1e-3 or ir_len <= 5760   # and it is NOT the plain convolution


def test_non_lti_block_sizes_differ_from_heff(oracle):
    """blk 1024: the reference reader skips tail blocks (time-varying); the closed form must NOT match."""
    h = oracle.gen_ir(131072)
    x = oracle.gen_pcm(1024 * 200)
    c = oracle.Nuc()
    assert c.set_impulse(h, 1024)
    y = c.run(x, 1024)
    he = oracle.heff(h, 1024)
    assert rms(y - fftconvolve(x, he)[:len(x)]) > 1e-6


def test_reset_and_silence_and_short_ir(oracle):
    h = oracle.gen_ir(100)
    c = oracle.Nuc()
    assert c.set_impulse(h, 512)
    x = oracle.gen_pcm(512 * 4)
    y1 = c.run(x, 512)
    c.reset()
    y2 = c.run(x, 512)
    assert np.array_equal(y1, y2)
    c.reset()
    oracle.lib().orc_nuc_add(c._h, None, 512)       # nullptr input = silence (NUC.h:205)
    y, got = c.get(512)
    assert got == 512 and not y.any()
    assert not oracle.Nuc().set_impulse(np.zeros(0), 512)


def test_direct_head_and_scale(oracle):
    h = oracle.gen_ir(3000)
    x = oracle.gen_pcm(512 * 16)
    c = oracle.Nuc()
    assert c.set_impulse(h, 512, scale=0.5, direct=True)
    assert c.plan().directTaps == 32
    y = c.run(x, 512)
    assert rms(y - fftconvolve(x, 0.5 * h)[:len(x)]) < 1e-15


def test_ragged_calls_accumulate_partitions(oracle):
    """Add() with n < partSize accumulates input; output arrives with one partition of latency."""
    h = oracle.gen_ir(2048)
    x = oracle.gen_pcm(512 * 8)
    c = oracle.Nuc()
    c.set_impulse(h, 512)
    out = []
    for o in range(0, len(x), 128):
        c.add(x[o:o + 128])
        y, _ = c.get(128)
        out.append(y)
    y = np.concatenate(out)
    ref = fftconvolve(x, h)[:len(x)]
    assert rms(y[384:] - ref[:len(x) - 384]) < 1e-15     # first full partition completes at sample 512-128


def test_oracle_regression_vectors(oracle):
    g = np.load(os.path.join(GOLD, "nuc_oracle_vectors.npz"))
    for name, (L, B, nb) in {"c4096_b512": (4096, 512, 12), "c20000_b128": (20000, 128, 200),
                              "c131072_b512": (131072, 512, 300)}.items():
        c = oracle.Nuc()
        c.set_impulse(oracle.gen_ir(L), B)
        y = c.run(oracle.gen_pcm(B * nb), B)
        assert np.array_equal(y[g[name + "_idx"]], g[name + "_y"])
    x, xr = oracle.gen_pcm(4096, channel=0), oracle.gen_pcm(4096, channel=1)
    for sat in (0.0, 0.2):
        yl, yr, _ = oracle.eq_process_stereo(x, xr, oracle.eq_params_bench(sat))
        assert np.array_equal(yl[::64], g[f"eq_sat{sat}_l"]) and np.array_equal(yr[::64], g[f"eq_sat{sat}_r"])


def test_svf_equals_rbj_transfer_function_when_linear(oracle):
    """With saturation 0 one peaking band has the magnitude response A-peaking at f0 (sanity of the SVF form)."""
    c = oracle.svf_design(1, 1000.0, 6.0, 1.0, 48000.0)
    n = 1 << 15
    x = np.zeros(n)
    x[0] = 1.0
    st = np.zeros(2)
    oracle.lib().orc_svf_band_stereo_lane(oracle.dp(x), n, c, oracle.dp(st), 0.0)
    H = np.abs(np.fft.rfft(x))
    k0 = int(round(1000.0 / 48000.0 * n))
    assert abs(20 * np.log10(H[k0]) - 6.0) < 0.01 and abs(20 * np.log10(H[3])) < 0.01


def test_mono_and_stereo_kernels_agree_when_linear(oracle):
    c = oracle.svf_design(0, 120.0, -4.0, 0.9, 48000.0)
    x = oracle.gen_pcm(4096)
    a, b = x.copy(), x.copy()
    oracle.lib().orc_svf_band_stereo_lane(oracle.dp(a), len(a), c, oracle.dp(np.zeros(2)), 0.2)
    oracle.lib().orc_svf_band_mono(oracle.dp(b), len(b), c, oracle.dp(np.zeros(2)), 0.2)
    assert np.abs(a - b).max() < 1e-14          # FMA placement differs, values agree to rounding


def test_guards_zero_nonfinite(oracle):
    c = oracle.svf_design(1, 1000.0, 6.0, 1.0, 48000.0)
    x = np.array([1.0, np.nan, 1.0, np.inf, 0.5, 1e300, 0.25, 0.1])
    st = np.zeros(2)
    oracle.lib().orc_svf_band_stereo_lane(oracle.dp(x), len(x), c, oracle.dp(st), 0.2)
    assert np.all(np.isfinite(x)) and np.all(np.abs(x) <= 100.0) and np.all(np.isfinite(st))


def test_equal_power_sin_quirk(oracle):
    v = oracle.lib().orc_equal_power_sin(1.0)
    assert abs(v - load("survey_observations.json")["equal_power_sin_1"]) < 1e-7 and v != 1.0


def test_mid_side_bands_properties(oracle):
    """Mid/Side bands (basic process(block), Processing.cpp:690-739): with L == R the Side component is zero, so Side
    bands are the identity and Mid bands equal the scalar band kernel on the common signal; a flat (< 0.01 dB) band is
    inactive on that path (createBandNode, Coefficients.cpp:48-53) although it is enabled."""
    O = oracle
    x = O.gen_pcm(2048)
    p = O.eq_params_bench(0.2)
    for b in range(20):
        p.bands[b].enabled = 1 if b in (4, 9) else 0
    p.bands[4].channelMode = 4            # Side
    p.bands[9].channelMode = 3            # Mid
    p.totalGainDb = 0.0
    yl, yr, st = O.eq_process_stereo(x, x, p)
    c = O.svf_design(p.bands[9].type, p.bands[9].frequency, p.bands[9].gain, p.bands[9].q, 48000.0)
    # the Mid band sees M = (x + x) * 0.5 = x exactly and S = 0: L = M' + 0, R = M' - 0
    m = x.copy()
    stm = np.zeros(2)
    O.lib().orc_svf_band_mono(O.dp(m), len(m), c, O.dp(stm), float(np.float32(0.2)))   # saturation is a float parameter
    assert np.array_equal(yl, m) and np.array_equal(yr, m)
    assert np.array_equal(st[88 + 2 * 9:88 + 2 * 9 + 2], stm) and not st[:80].any() and not st[128:].any()
    # flat band: enabled, Peaking, |gain| < 0.01 dB -> skipped on the basic path, so the output does not change
    p.bands[10].enabled = 1
    p.bands[10].gain = 0.004
    yl2, yr2, _ = O.eq_process_stereo(x, x, p)
    assert np.array_equal(yl2, yl)
    # without a Mid/Side band the same flat band is active (cache path) and the saturation blend alters the signal
    p.bands[4].channelMode = 0
    p.bands[9].channelMode = 0
    a, _, _ = O.eq_process_stereo(x, x, p)
    p.bands[10].enabled = 0
    b, _, _ = O.eq_process_stereo(x, x, p)
    assert np.abs(a - b).max() > 1e-6


@pytest.mark.parametrize("ir_len,block", [(4096, 512), (8192, 512), (16384, 512), (8192, 1024), (4096, 256)] +
                         [(n, 512) for n in (1024, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193)])
def test_oracle_passes_the_references_own_nuc_test(oracle, ir_len, block):
    """The only checks the reference's tests make on convolver output (src/tests/MT-NUPC-Measurement.cpp:116-122,
    :183, :196-199): SetImpulse succeeds and the Dirac response has energy > 1e-20, for these (irLen, blockSize) pairs
    with IR = sign(sin(0.1 i)).  They hold no values, so they cannot pin the oracle's numbers (parity unpinned); on top of
    them the response of an LTI plan must equal the closed-form h_eff of SURVEY.md A6."""
    i = np.arange(ir_len)
    ir = np.where(np.sin(i * 0.1) > 0.0, 1.0, -1.0)
    total = ((ir_len * 2 + block - 1) // block) * block
    x = np.zeros(total)
    x[0] = 1.0
    nuc = oracle.Nuc()
    assert nuc.set_impulse(ir, block)
    y = nuc.run(x, block)
    assert float(np.sum(y * y)) > 1e-20
    pl = nuc.plan()
    if pl.ltiValid:
        h = oracle.heff(ir, block)
        m = min(len(h), total)
        assert np.abs(y[:m] - h[:m]).max() <= 1e-12
