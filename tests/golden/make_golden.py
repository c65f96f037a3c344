#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.  Run in the build container (the reference tree
and oracle/_ref are only available there).  Fixtures are DATA (inputs + expected outputs), never reference text.

  fasttanh_ref.json        outputs of the reference's own src/dsp/math/FastTanhApprox.h (scalar and SSE2 paths),
                           obtained by compiling that stand-alone header where it lies (oracle/ref_probe.cpp)
  eq_params_default_ref.json  a default-constructed convo::EQParameters (src/core/EQParameters.h) via the same probe
  svf_display_biquad_ref.json  the reference's own svfToDisplayBiquad (src/tests/EQProcessorMaxGainTests.cpp:67-87, compiled
                           unmodified via oracle/ref_probe_eqmath.cpp) applied to SVF coefficient sets: the biquad each band IS
  rbj_biquad_ref.json      the reference's own cookbook designers calcPeakingBiquad / calcLowShelfBiquad / calcHighShelfBiquad
                           (src/tests/EQBoundExcessBenchmark.cpp:188-245, compiled unmodified via oracle/ref_probe_eqbound.cpp)
                           over a grid of frequency x gain x Q x sample rate: the responses calcSVFCoeffs aims at
  survey_observations.json hand-transcribed observations of the RUNNING reference recorded in SURVEY.md
                           (section 0 findings 2-4, section 8(a) row A6, section 8(c)); provenance: survey session
  nuc_oracle_vectors.npz   outputs of this repo's oracle for fixed seeded inputs (regression pin of the oracle
                           itself, NOT reference output)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402


def main():
    R = O.ref_probe()
    if R is None:
        raise SystemExit("oracle/_ref/libcpq_ref_probe.so missing and /root/reference absent")
    xs = np.concatenate([np.linspace(-6.0, 6.0, 97), [4.5, -4.5, 4.499999999999999, -4.499999999999999, 0.0, -0.0,
                                                       1e-300, 1e300, -1e300, 3.0, np.inf, -np.inf]])
    ft = {"x": [repr(float(v)) for v in xs],
          "scalar": [repr(float(R.ref_fast_tanh_scalar(float(v)))) for v in xs],
          "v128": [repr(float(R.ref_fast_tanh_v128(float(v)))) for v in xs],
          "source": "reference src/dsp/math/FastTanhApprox.h compiled unmodified via oracle/ref_probe.cpp"}
    with open(os.path.join(HERE, "fasttanh_ref.json"), "w") as f:
        json.dump(ft, f, indent=0)

    p = O.EqParams()
    R.ref_eq_params_default(p)
    eq = {"bands": [[b.frequency, b.gain, b.q, b.enabled, b.type, b.channelMode] for b in p.bands],
          "totalGainDb": p.totalGainDb, "agcEnabled": p.agcEnabled,
          "nonlinearSaturation": p.nonlinearSaturation, "filterStructure": p.filterStructure,
          "sizeof_EQParameters": R.ref_sizeof_eq_parameters(), "sizeof_EQBandParams": R.ref_sizeof_eq_band_params(),
          "source": "reference src/core/EQParameters.h compiled unmodified via oracle/ref_probe.cpp"}
    with open(os.path.join(HERE, "eq_params_default_ref.json"), "w") as f:
        json.dump(eq, f, indent=0)

    import importlib.util
    spec = importlib.util.spec_from_file_location("t_ref_eq", os.path.join(os.path.dirname(HERE), "test_ref_eq_math_cpu.py"))
    t = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(t)
    cases = []
    for btype, freq, gain, q in t.CASES:
        c = O.svf_design(btype, freq, gain, q, 48000.0)
        svf = [c.a1, c.a2, c.a3, c.m0, c.m1, c.m2]
        b, a = O.ref_svf_to_display_biquad(svf)
        cases.append({"type": btype, "freq": freq, "gain_db": gain, "q": q, "svf": [float(v).hex() for v in svf],
                      "biquad": [float(v).hex() for v in np.concatenate([b, a])]})
    with open(os.path.join(HERE, "svf_display_biquad_ref.json"), "w") as f:
        json.dump({"source": "reference src/tests/EQProcessorMaxGainTests.cpp svfToDisplayBiquad compiled unmodified via "
                             "oracle/ref_probe_eqmath.cpp; svf = a1 a2 a3 m0 m1 m2 (this repo's design, hex doubles), "
                             "biquad = b0 b1 b2 a0 a1 a2 (the reference's equivalent biquad, hex doubles); sample rate 48000",
                   "cases": cases}, f, indent=0)

    # the reference's cookbook designers over a grid (float-representable parameters: the EQ's parameters are floats)
    rb = []
    for sr in (44100.0, 48000.0, 96000.0, 192000.0, 384000.0):
        for btype in (0, 1, 2):
            for freq in (20.0, 55.0, 200.0, 1000.0, 4000.0, 12000.0, 19000.0):
                for gain in (-24.0, -9.0, -1.0, 0.5, 6.0, 24.0):
                    for q in (0.1, 0.5, 0.7071, 1.41, 5.0, 20.0):
                        if (len(rb) * 7 + int(freq)) % 5:          # a fifth of the grid keeps the fixture small
                            rb_skip = True
                        else:
                            rb_skip = False
                        f32, g32, q32 = float(np.float32(freq)), float(np.float32(gain)), float(np.float32(q))
                        if rb_skip and not (sr == 48000.0 and q == 0.7071):
                            continue
                        b, a = O.ref_rbj_biquad(btype, f32, g32, q32, sr)
                        rb.append({"type": btype, "freq": f32, "gain_db": g32, "q": q32, "sr": sr,
                                   "biquad": [float(v).hex() for v in np.concatenate([b, a])]})
    with open(os.path.join(HERE, "rbj_biquad_ref.json"), "w") as f:
        json.dump({"source": "reference src/tests/EQBoundExcessBenchmark.cpp calcLowShelfBiquad (type 0) / calcPeakingBiquad (1) / "
                             "calcHighShelfBiquad (2) compiled unmodified via oracle/ref_probe_eqbound.cpp; "
                             "biquad = b0 b1 b2 a0 a1 a2 (hex doubles)", "cases": rb}, f, indent=0)

    survey = {
        "source": "SURVEY.md: values observed from the unmodified reference sources running in the survey session",
        "svf_known_answer": {   # Peaking 1 kHz +6 dB Q 0.707 @48 kHz (glibc libm)
            "type": 1, "freq": 1000.0, "gain_db": 6.0, "q": 0.707, "sr": 48000.0,
            "a1": "0.93464312858157605", "a2": "0.061259747143704447", "a3": "0.004015175958984288",
            "m0": "1", "m1": "0.99659369608338311", "m2": "0"},
        "layer_plan_48k_blk512_131072": {   # finding 2 + section 3.3
            "l0_taps": 5760, "l0_parts": 12, "l1_part": 4096, "l1_parts": 31, "l1_parts_per_callback": 4},
        "tail_gains_default": {"g1": 1.4375, "g2": 1.10, "tail_start_sec": 0.12},   # finding 4 / A6
        "lags": [   # A6 verified residuals: (irLen, block) -> lag per tail layer
            {"ir_len": 131072, "block": 512, "lag": [1408]},
            {"ir_len": 131072, "block": 128, "lag": [-2304, -60672]},
            {"ir_len": 131072, "block": 256, "lag": [-2176]},
            {"ir_len": 524288, "block": 512, "lag_last": -232064}],
        "lti_invalid_blocks": [1024, 2048],     # A6: model invalid (reference drops tail blocks)
        "config1": {"ir_len": 4096, "block": 512, "latency_reported": 512, "lag_observed": 0,
                    "rms_vs_fftconvolve_max": 1.4e-16},
        "equal_power_sin_1": 1.0000035,        # finding 5: wetG at mix = 1 (9th-order Taylor)
        "fast_tanh_v128_at_clip": 1.01613,     # A15: 4.5*47.25/209.25
    }
    with open(os.path.join(HERE, "survey_observations.json"), "w") as f:
        json.dump(survey, f, indent=1)

    # regression pin of the oracle itself on small seeded cases
    out = {}
    for name, (L, B, nb) in {"c4096_b512": (4096, 512, 12), "c20000_b128": (20000, 128, 200),
                              "c131072_b512": (131072, 512, 300)}.items():
        h = O.gen_ir(L)
        x = O.gen_pcm(B * nb)
        c = O.Nuc()
        c.set_impulse(h, B)
        y = c.run(x, B)
        idx = np.linspace(0, len(y) - 1, 64).astype(np.int64)
        out[name + "_idx"] = idx
        out[name + "_y"] = y[idx]
    x = O.gen_pcm(4096, channel=0)
    xr = O.gen_pcm(4096, channel=1)
    for sat in (0.0, 0.2):
        yl, yr, _ = O.eq_process_stereo(x, xr, O.eq_params_bench(sat))
        out[f"eq_sat{sat}_l"] = yl[::64]
        out[f"eq_sat{sat}_r"] = yr[::64]
    np.savez(os.path.join(HERE, "nuc_oracle_vectors.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
