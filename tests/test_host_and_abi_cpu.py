"""CPU tests of the product's host logic through the C ABI (no GPU): the shared library loads, exports every
symbol include/convopeq_mi355x.h declares, its host-only design functions agree with the oracle and with the
fixtures, and device entry points fail loudly without a GPU (no CPU fallback)."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.fixture(scope="module")
def amd():
    import convopeq_amd
    return convopeq_amd


def header_symbols():
    text = open(os.path.join(ROOT, "include", "convopeq_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cpq_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(amd):
    from convopeq_amd import _capi
    lib = C.CDLL(_capi.LIB_PATH)
    declared = header_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(_capi.SYMBOLS) == declared, "python binding table out of sync with the header"
    assert lib.cpq_abi_version() == 2


def test_library_contains_gfx950_code_object():
    so = os.path.join(ROOT, "convopeq_amd", "libconvopeq_mi355x.so")
    out = subprocess.run(["strings", "-a", so], capture_output=True, text=True).stdout
    assert "gfx950" in out
    for k in ("k_rfft_fwd_ols", "k_fdl_mac", "k_rfft_inv_ols", "k_svf_cascade"):
        assert k in out


def test_product_never_references_the_oracle():
    """The product path must not link, load or import anything under oracle/."""
    so = os.path.join(ROOT, "convopeq_amd", "libconvopeq_mi355x.so")
    needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
    assert "oracle" not in needed
    for dirpath, _, files in os.walk(os.path.join(ROOT, "convopeq_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in src and "cpq_oracle" not in src and "libcpq_oracle" not in src, f
                assert "ir_ingest_oracle" not in src, f
                code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith(("//", "*", "/*", "#", '"""')))
                assert "oracle/" not in code and "import oracle" not in code, f


def test_no_gpu_means_loud_failure(amd):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(amd.CpqError) as ei:
        amd.BatchedEngine(1)
    assert ei.value.status == -2 and "no CPU fallback" in str(ei.value)


@pytest.mark.parametrize("ir_len,block", [(4096, 512), (131072, 512), (131072, 128), (131072, 256), (524288, 512),
                                           (131072, 1024), (131072, 2048), (100, 64), (5761, 512), (300000, 512)])
def test_plan_matches_oracle(amd, oracle, ir_len, block):
    a = amd.nuc_plan(ir_len, block)
    o = oracle.plan(ir_len, block)
    assert a.num_layers == o.numLayers and a.latency == o.latency and a.lti_valid == o.ltiValid
    for i in range(a.num_layers):
        assert (a.part_size[i], a.offset[i], a.len[i], a.num_parts_ir[i], a.num_parts[i], a.parts_per_callback[i],
                a.output_delay[i], a.lag[i], a.done_callback[i]) == \
               (o.partSize[i], o.offset[i], o.len[i], o.numPartsIR[i], o.numParts[i], o.partsPerCallback[i],
                o.outputDelay[i], o.lag[i], o.doneCallback[i])
        assert a.gain[i] == o.gain[i]


def test_plan_with_filter_spec_variants(amd, oracle):
    for kw in (dict(tail_mode=0, tail_strength=1.5), dict(tail_mode=2), dict(tail_enabled=0),
               dict(tail_start_seconds=0.3, tail_l1l2_multiplier=4, tail_strength=0.2), dict(sample_rate=96000.0)):
        sa = amd.FilterSpec.defaults(**kw)
        so = oracle.FilterSpec.defaults(**{{"tail_mode": "tailMode", "tail_strength": "tailStrength",
                                            "tail_enabled": "tailEnabled", "tail_start_seconds": "tailStartSeconds",
                                            "tail_l1l2_multiplier": "tailL1L2Multiplier",
                                            "sample_rate": "sampleRate"}[k]: v for k, v in kw.items()})
        a = amd.nuc_plan(200000, 512, spec=sa)
        o = oracle.plan(200000, 512, spec=so)
        assert a.num_layers == o.numLayers
        for i in range(a.num_layers):
            assert (a.len[i], a.part_size[i], a.gain[i], a.lag[i]) == (o.len[i], o.partSize[i], o.gain[i], o.lag[i])


def test_heff_matches_oracle_and_survey_form(amd, oracle):
    h = oracle.gen_ir(131072)
    a = amd.nuc_heff(h, 512)
    assert np.array_equal(a, oracle.heff(h, 512))
    # SURVEY finding 4: y = x*h[0:5760] + 1.4375 * delay_1408(x*h[5760:])
    ref = np.zeros(131072 + 1408)
    ref[:5760] = h[:5760]
    ref[5760 + 1408:] += 1.4375000000000002 * h[5760:]
    assert np.array_equal(a, ref)
    assert np.array_equal(amd.nuc_heff(h[:4096], 512, scale=0.5), 0.5 * h[:4096])


def test_svf_design_matches_oracle_and_survey(amd, oracle):
    k = json.load(open(os.path.join(HERE, "golden", "survey_observations.json")))["svf_known_answer"]
    c = amd.design_svf(k["type"], k["freq"], k["gain_db"], k["q"], k["sr"])
    for n in ("a1", "a2", "a3", "m0", "m1", "m2"):
        assert getattr(c, n) == float(k[n])
    rng = np.random.default_rng(3)
    for _ in range(300):
        t = int(rng.integers(0, 5))
        f, g, q = float(rng.uniform(5, 30000)), float(rng.uniform(-60, 60)), float(rng.uniform(0.001, 30))
        sr = float(rng.choice([44100.0, 48000.0, 96000.0, 192000.0]))
        a, o = amd.design_svf(t, f, g, q, sr), oracle.svf_design(t, f, g, q, sr)
        for n in ("a1", "a2", "a3", "m0", "m1", "m2"):
            assert getattr(a, n) == getattr(o, n), (t, f, g, q, sr, n)
    c = amd.design_svf(1, 1000.0, 0.0, 1.0, -1.0)           # invalid rate -> bypass coefficients
    assert (c.a1, c.a2, c.a3, c.m0, c.m1, c.m2) == (1.0, 0.0, 0.0, 1.0, 0.0, 0.0)


def test_eq_params_default_matches_reference_fixture(amd):
    g = json.load(open(os.path.join(HERE, "golden", "eq_params_default_ref.json")))
    p = amd.eq_params_default()
    for b, ref in zip(p.bands, g["bands"]):
        assert [b.frequency, b.gain, b.q, b.enabled, b.type, b.channel_mode] == ref
    assert p.nonlinear_saturation == np.float32(g["nonlinearSaturation"]) and p.total_gain_db == g["totalGainDb"]


def test_invalid_arguments_are_rejected_without_gpu(amd):
    from convopeq_amd import _capi
    lib = _capi.load()
    p = _capi.NucPlan()
    assert lib.cpq_nuc_plan_compute(0, 512, 0, None, C.byref(p)) == _capi.CPQ_ERR_INVALID_ARG
    assert lib.cpq_nuc_plan_compute(100, 0, 0, None, C.byref(p)) == _capi.CPQ_ERR_INVALID_ARG
    assert lib.cpq_engine_create(None, None) == _capi.CPQ_ERR_INVALID_ARG
    h = _capi._E()
    d = _capi.EngineDesc(C.sizeof(_capi.EngineDesc), 0, 1, 500, 4096, 1, 0, 0, 48000.0, 0, 0)   # block not a power of two
    assert lib.cpq_engine_create(C.byref(d), C.byref(h)) == _capi.CPQ_ERR_INVALID_ARG
    assert b"power of two" in lib.cpq_last_error(None)
    assert lib.cpq_status_string(-5) == b"not supported by this engine version"
    assert lib.cpq_kernel_name(1) == b"k_fdl_mac"


def test_outfilter_design_matches_oracle_and_rbj(amd, oracle):
    """N2: OutputFilter::prepare coefficient design through the C ABI == oracle restatement; sanity vs scipy."""
    from scipy.signal import freqz
    for sr in (44100.0, 48000.0, 96000.0):
        for cil in (0, 1):
            for hc in range(3):
                for lc in range(2):
                    for lp in range(3):
                        a = amd.outfilter_design(cil, hc, lc, lp, sr)
                        o = oracle.outfilter_design(cil, hc, lc, lp, sr)
                        for x, y in zip(a, o):
                            assert (x.b0, x.b1, x.b2, x.a1, x.a2) == (y.b0, y.b1, y.b2, y.a1, y.a2)
    q = amd.outfilter_design(0, 1, 0, 1, 48000.0)        # EQ last: 20 Hz HPF, 2 x 19 kHz LPF (Q 0.7071)
    w, hh = freqz([q[1].b0, q[1].b1, q[1].b2], [1.0, q[1].a1, q[1].a2], worN=[19000.0], fs=48000.0)
    assert abs(20 * np.log10(abs(hh[0])) + 3.01) < 0.02
    w, hh = freqz([q[0].b0, q[0].b1, q[0].b2], [1.0, q[0].a1, q[0].a2], worN=[20.0], fs=48000.0)
    assert abs(20 * np.log10(abs(hh[0])) + 3.01) < 0.02
    s = amd.outfilter_design(1, 2, 0, 1, 48000.0)        # Soft high cut: second stage is the identity
    assert (s[2].b0, s[2].b1, s[2].b2, s[2].a1, s[2].a2) == (1.0, 0.0, 0.0, 0.0, 0.0)


def test_oracle_df2t_matches_scipy_lfilter(oracle):
    from scipy.signal import lfilter
    q = oracle.outfilter_design(1, 0, 0, 1, 48000.0)
    x = oracle.gen_pcm(4096)
    yl, yr, _ = oracle.outfilter_process_stereo(x, x, q)
    ref = x
    for s in q:
        ref = lfilter([s.b0, s.b1, s.b2], [1.0, s.a1, s.a2], ref)
    # the 18 Hz high-pass has poles at |z| = 0.998: any fp64 evaluation order sits ~2e-13 from the exact result
    assert np.abs(yl - ref).max() < 2e-12 and np.array_equal(yl, yr)


def test_multi_device_host_example_builds_and_fails_loudly_without_a_gpu(tmp_path):
    """INTEGRATION.md's multi-GPU host loop (one engine per device, one worker thread each) compiles against the C header
    alone, links against the library's exports, and -- without a GPU -- reports CPQ_ERR_NO_DEVICE per device instead of
    computing anything."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "convopeq_amd")
    exe = tmp_path / "multi_device_host"
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "examples", "multi_device_host.cpp"), "-o", str(exe),
                           "-L", lib_dir, "-lconvopeq_mi355x", "-Wl,-rpath," + lib_dir, "-pthread"])
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the example would run for real")
    r = subprocess.run([str(exe), "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("status -2") == 2 and "engines created: 0 of 2" in r.stdout, r.stdout
