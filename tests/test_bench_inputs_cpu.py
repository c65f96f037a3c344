"""bench.py generates its synthetic PCM / IRs with numpy; they must be the SURVEY.md 8(d) counter-based streams that the
oracle (and therefore the parity tests) use."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_bench_generators_match_oracle(oracle):
    b = _bench()
    for stream, ch in ((0, 0), (3, 1), (255, 0), (8191, 1)):
        assert np.array_equal(b.gen_pcm(4096, stream, ch, start=1000), oracle.gen_pcm(4096, stream=stream, channel=ch, start=1000))
        ir_b, ir_o = b.gen_ir(5000, stream, ch), oracle.gen_ir(5000, stream=stream, channel=ch)
        assert np.abs(ir_b - ir_o).max() <= 1e-17       # numpy exp vs libm exp: last-ulp differences only
    x = b.gen_pcm(100000, 1, 0)
    assert abs(x.mean()) < 2e-3 and abs(x.std() - 0.25 / np.sqrt(3)) < 2e-3 and np.abs(x).max() <= 0.25


def test_bench_algorithmic_bytes_formula():
    """(2S (K+T-1) + nIR K + 2S T) * P * 16 bytes per k_fdl_mac launch at config 2."""
    n_ch, k, t, p = 512, 259, 64, 512
    assert (n_ch * (k + t - 1) + n_ch * k + n_ch * t) * p * 16 == 2705326080


def test_bench_launch_plan_and_streaming_bytes():
    """The native schedule under one 512-sample block per call (the reference's own call pattern,
    src/convolver/ConvolverProcessor.Runtime.cpp:659-682): layer 0 (12 x 512) launches every call, layer 1 (31 x 4096) once
    every 8 calls -- a call is charged an eighth of that launch (= the reference's 4 partitions per callback,
    src/MKLNonUniformConvolver.cpp:988-994), not all 31 partitions (round 3 charged them all: roofline.frac 3.7)."""
    b = _bench()
    lp = b.launch_plan(512, [(512, 12), (4096, 31)])
    assert lp == [(512, 12, 1, 1.0), (4096, 31, 1, 0.125)]
    n_ch = 512
    mac = b.algorithmic_bytes_per_step(n_ch, 512, n_ch, lp, lp, native_tails=1)["k_fdl_mac"]
    l0 = (n_ch * 12 + n_ch * 12 + n_ch) * 512 * 16            # 12 FDL rows + 12 IR rows + 1 output row per channel
    l1 = (n_ch * 31 + n_ch * 31 + n_ch) * 4096 * 16
    assert mac == l0 + l1 / 8 and 3.5e8 < mac < 3.8e8         # ~ 360 MB per call, not 2.2 GB
    # whole-call launches: 1024 blocks per call on the uniform schedule at P = 4096, K = 33: one launch of 128 partitions
    lp = b.launch_plan(524288, [(4096, 33)])
    assert lp == [(4096, 33, 128, 1.0)]
    assert b.algorithmic_bytes_per_step(n_ch, 524288, n_ch, lp, lp)["k_fdl_mac"] == (n_ch * (33 + 127) + n_ch * 33 + n_ch * 128) * 4096 * 16
    # a 480-sample quantum against 512-sample layer-0 partitions: 15 launches per 16 calls
    lp = b.launch_plan(480, [(512, 12)])
    assert lp[0][2] == 1 and abs(lp[0][3] - 0.9375) < 1e-12
    # config 2 at T = 64, P = 512 (round-1 schedule): the figure DESIGN section 4 quotes
    lp = b.launch_plan(64 * 512, [(512, 259)])
    assert b.algorithmic_bytes_per_step(n_ch, 64 * 512, n_ch, lp, lp)["k_fdl_mac"] == 2705326080


def test_bench_svf_flop_model_follows_the_saturation_setting():
    b = _bench()
    assert b.svf_flop_model(0.2) == {"slots": 33.0, "arithmetic": 26.0}
    assert b.svf_flop_model(0.0) == {"slots": 21.0, "arithmetic": 17.0}      # no fastTanh blend at saturation 0 (`if sat > 0`)


def test_fraction_check_flags_accounting_errors():
    import importlib.util
    spec = importlib.util.spec_from_file_location("sp", os.path.join(ROOT, "tools", "summarize_profiles.py"))
    sp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sp)
    ok = {"roofline": {"frac": 0.41, "hbm_kernel": {"frac": 0.68}}, "kernels": {"k_fdl_mac": {"achieved_gbs": 5400.0}}}
    assert sp.check_fractions(ok) == []
    bad = {"roofline": {"frac": 3.68}, "kernels": {"k_fdl_mac": {"achieved_gbs": 29451.7}}}
    assert len(sp.check_fractions(bad)) == 2


def test_bench_bytes_of_the_fused_native_path():
    """Whole 512-sample-block calls of a plan group: layer 0's forward transform also fills the accumulators of the layers whose
    partition has not filled, its inverse transform also reads the tail layers' delay lines -- no pass over the output is left
    (profiles/r04z_sweep_configs.jsonl line 16 charged the removed pass: 18 x the HBM peak)."""
    b = _bench()
    n_ch = 128
    lp = b.launch_plan(524288, [(512, 12), (4096, 64), (32768, 8)])
    fused = b.algorithmic_bytes_per_step(n_ch, 524288, n_ch, lp, lp, native_tails=2, native_fused=True)
    plain = b.algorithmic_bytes_per_step(n_ch, 524288, n_ch, lp, lp, native_tails=2)
    assert fused["k_convproc_mix"] == 0 and plain["k_convproc_mix"] == n_ch * 524288 * 32
    assert fused["k_rfft_inv_ols"] - plain["k_rfft_inv_ols"] == n_ch * 524288 * 16
    assert fused["k_rfft_fwd_ols"] == plain["k_rfft_fwd_ols"]                 # every layer's partition fills: nothing to accumulate
    lp1 = b.launch_plan(512, [(512, 12), (4096, 31)])
    f1 = b.algorithmic_bytes_per_step(512, 512, 512, lp1, lp1, native_tails=1, native_fused=True)
    p1 = b.algorithmic_bytes_per_step(512, 512, 512, lp1, lp1, native_tails=1)
    assert f1["k_rfft_fwd_ols"] - p1["k_rfft_fwd_ols"] == 512 * 512 * 8       # the block into layer 1's accumulator
