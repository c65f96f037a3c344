"""Pins against the reference's OWN stand-alone EQ math (src/tests/EQProcessorMaxGainTests.cpp, compiled unmodified into
oracle/_ref by oracle/ref_probe_eqmath.cpp; only standard headers):

 * its main() passes here (188 checks of the reference) -- the build is the reference's code;
 * svfToDisplayBiquad ("実装は EQProcessor.Coefficients.cpp:347-368 と同一") states which biquad an SVF band with
   coefficients (a1, a2, a3, m0, m1, m2) IS.  The oracle's band recurrence (restated from processBand /
   processBandStereo, EQProcessor.Processing.cpp:128-276) must therefore filter like scipy.signal.lfilter with that
   biquad: an anchor for the band kernel's linear part that does not come from this repo's reading of the kernel;
 * calcLPFSVF: the low-pass coefficients calcSVFCoeffs derives -- bit-equal to the product's designSvf and the oracle's.

The fixture tests/golden/svf_display_biquad_ref.json (made by tests/golden/make_golden.py from the probe) carries the
same data to the GPU box, where tests/test_gpu_parity.py::test_svf_band_matches_the_reference_display_biquad runs the HIP
kernels against it.
"""
import json
import os

import numpy as np
import pytest
from scipy.signal import lfilter

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [(0, 120.0, 6.0, 0.7), (0, 80.0, -9.0, 1.2), (1, 1000.0, 6.0, 0.707), (1, 3500.0, -12.0, 4.0), (1, 250.0, 18.0, 0.3),
         (2, 8000.0, 4.5, 0.9), (2, 12000.0, -7.0, 0.6), (3, 1000.0, 0.0, 0.707), (3, 15000.0, 0.0, 2.0), (4, 40.0, 0.0, 0.707),
         (4, 2000.0, 0.0, 1.5), (1, 19000.0, 10.0, 8.0)]


def _need_probe(oracle):
    R = oracle.ref_probe()
    if R is None or not hasattr(R, "ref_svf_to_display_biquad"):
        pytest.skip("oracle/_ref probe of the reference's EQ math test not available")
    return R


def test_reference_eq_math_selftest_passes(oracle):
    R = _need_probe(oracle)
    assert R.ref_eq_math_selftest() == 0            # the reference's own 188 assertions, compiled from its tree


def test_lowpass_design_is_the_references_formula(oracle):
    """calcLPFSVF (reference) == the oracle's orc_svf_design == the product's designSvf for LowPass, bit for bit."""
    _need_probe(oracle)
    import ctypes as C
    lib = None
    try:
        from convopeq_amd import _capi
        lib = _capi.load()
    except Exception:                # noqa: BLE001  (no product library in this environment: the oracle half still runs)
        pass
    for f, q in ((1000.0, 0.707), (60.0, 1.0), (15000.0, 2.5), (333.0, 0.5), (4800.0, 12.0)):
        ref = oracle.ref_calc_lpf_svf(float(np.float32(f)), float(np.float32(q)), 48000.0)
        o = oracle.svf_design(3, f, 0.0, q, 48000.0)
        assert [o.a1, o.a2, o.a3, o.m0, o.m1, o.m2] == list(ref)
        if lib is not None:
            c = _capi.SvfCoeffs()
            assert lib.cpq_eq_design_svf(3, f, 0.0, q, 48000.0, C.byref(c)) == 0
            assert [c.a1, c.a2, c.a3, c.m0, c.m1, c.m2] == list(ref)


@pytest.mark.parametrize("btype,freq,gain,q", CASES)
def test_oracle_band_recurrence_is_the_references_biquad(oracle, btype, freq, gain, q):
    """Oracle recurrence (saturation 0, both arithmetic flavours) == lfilter with the reference's own equivalent biquad."""
    _need_probe(oracle)
    O = oracle
    c = O.svf_design(btype, freq, gain, q, 48000.0)
    b, a = O.ref_svf_to_display_biquad([c.a1, c.a2, c.a3, c.m0, c.m1, c.m2])
    x = O.gen_pcm(16384, stream=3, channel=0)
    ref = lfilter(b / a[0], a / a[0], x)
    for kernel in (O.lib().orc_svf_band_mono, O.lib().orc_svf_band_stereo_lane):
        y = x.copy()
        state = np.zeros(2)
        kernel(O.dp(y), len(y), c, O.dp(state), 0.0)
        err = np.abs(y - ref).max() / max(1.0, np.abs(ref).max())
        assert err <= 2e-12, (btype, freq, err)          # a direct-form biquad at 40 Hz / 48 kHz is itself only ~1e-12 accurate


def test_fixture_matches_the_probe(oracle):
    """tests/golden/svf_display_biquad_ref.json is what the probe says today (regenerate with make_golden.py if not)."""
    _need_probe(oracle)
    with open(os.path.join(HERE, "golden", "svf_display_biquad_ref.json")) as f:
        fx = json.load(f)
    for case in fx["cases"]:
        b, a = oracle.ref_svf_to_display_biquad([float.fromhex(v) for v in case["svf"]])
        assert [v.hex() for v in np.concatenate([b, a])] == case["biquad"]


def test_fixture_cases_are_the_products_designs():
    """The SVF coefficients in the fixture are designSvf's (so the GPU test filters with what the engine would design)."""
    import ctypes as C
    from convopeq_amd import _capi
    lib = _capi.load()
    with open(os.path.join(HERE, "golden", "svf_display_biquad_ref.json")) as f:
        fx = json.load(f)
    for case in fx["cases"]:
        c = _capi.SvfCoeffs()
        assert lib.cpq_eq_design_svf(case["type"], case["freq"], case["gain_db"], case["q"], 48000.0, C.byref(c)) == 0
        assert [float(v).hex() for v in (c.a1, c.a2, c.a3, c.m0, c.m1, c.m2)] == case["svf"]


# ---------------------------------------------------------------------------------------------------------------------
# The reference's cookbook designers (src/tests/EQBoundExcessBenchmark.cpp:188-245, compiled unmodified via
# oracle/ref_probe_eqbound.cpp; fixture tests/golden/rbj_biquad_ref.json): peaking / low shelf / high shelf as z-domain
# biquads.  calcSVFCoeffs (EQProcessor.Coefficients.cpp:431-560) designs the SAME transfer functions in TPT-SVF form -- both
# are the bilinear transform of the cookbook's analog prototypes, pre-warped at the band frequency -- so the product's
# designSvf, read back through the band's state-space form, must have the response of the reference's biquad.  Measured over
# the grid (20 Hz ... 19 kHz, +-24 dB, Q 0.1 ... 20, 44.1 ... 384 kHz): relative response deviation <= 2.5e-7, all of it the
# cookbook form's own cancellation at 20 ... 55 Hz against 192 / 384 kHz (1 - cos w0); <= 2e-10 from 200 Hz up at <= 96 kHz.
def _svf_response(c, w):
    """H(e^jw) of the TPT-SVF band with coefficients c = (a1, a2, a3, m0, m1, m2): state (ic1, ic2), input v0."""
    a1, a2, a3, m0, m1, m2 = c
    A = np.array([[2 * a1 - 1, -2 * a2], [2 * a2, 1 - 2 * a3]])
    Bv = np.array([2 * a2, 2 * a3])
    Cv = np.array([m1 * a1 + m2 * a2, -m1 * a2 + m2 * (1 - a3)])
    D = m0 + m1 * a2 + m2 * a3
    z = np.exp(1j * w)
    det = (z - A[0, 0]) * (z - A[1, 1]) - A[0, 1] * A[1, 0]
    inv_b0 = ((z - A[1, 1]) * Bv[0] + A[0, 1] * Bv[1]) / det
    inv_b1 = (A[1, 0] * Bv[0] + (z - A[0, 0]) * Bv[1]) / det
    return D + Cv[0] * inv_b0 + Cv[1] * inv_b1


def _load_rbj_cases():
    with open(os.path.join(HERE, "golden", "rbj_biquad_ref.json")) as f:
        return json.load(f)["cases"]


def test_rbj_fixture_matches_the_probe(oracle):
    if oracle.ref_rbj_biquad(1, 1000.0, 6.0, 0.707, 48000.0) is None:
        pytest.skip("oracle/_ref probe of the reference's EQ bound benchmark not available")
    for case in _load_rbj_cases():
        b, a = oracle.ref_rbj_biquad(case["type"], case["freq"], case["gain_db"], case["q"], case["sr"])
        assert [v.hex() for v in np.concatenate([b, a])] == case["biquad"]


def test_shelf_and_peaking_designs_have_the_response_of_the_references_cookbook_biquads(oracle):
    """designSvf (product) and orc_svf_design (oracle), low shelf / peaking / high shelf, against the reference's designers."""
    import ctypes as C
    from convopeq_amd import _capi
    lib = _capi.load()
    worst, worst_mid = 0.0, 0.0
    for case in _load_rbj_cases():
        bq = np.array([float.fromhex(v) for v in case["biquad"]])
        sr = case["sr"]
        w = np.geomspace(2 * np.pi * 10.0 / sr, np.pi * 0.999, 300)
        zi = np.exp(-1j * w)
        h_ref = (bq[0] + bq[1] * zi + bq[2] * zi * zi) / (bq[3] + bq[4] * zi + bq[5] * zi * zi)
        c = _capi.SvfCoeffs()
        assert lib.cpq_eq_design_svf(case["type"], case["freq"], case["gain_db"], case["q"], sr, C.byref(c)) == 0
        o = oracle.svf_design(case["type"], case["freq"], case["gain_db"], case["q"], sr)
        assert [c.a1, c.a2, c.a3, c.m0, c.m1, c.m2] == [o.a1, o.a2, o.a3, o.m0, o.m1, o.m2]      # product == oracle, bit for bit
        h = _svf_response((c.a1, c.a2, c.a3, c.m0, c.m1, c.m2), w)
        dev = float(np.max(np.abs(h - h_ref) / np.abs(h_ref)))
        worst = max(worst, dev)
        if case["freq"] >= 200.0 and sr <= 96000.0:
            worst_mid = max(worst_mid, dev)
    assert worst <= 1e-6, worst
    assert worst_mid <= 1e-9, worst_mid
