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
