"""IR ingest (WAV -> double, conditioning, computeScaleFactor): host-only entry points of the C ABI against
the numpy restatement in oracle/ir_ingest_oracle.py, the restatement against scipy's WAV reader and closed forms.
No GPU needed: cpq_ir_* never touch the device."""
import importlib.util
import math
import os
import struct

import numpy as np
import pytest
from scipy.io import wavfile

HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLE = os.path.join(HERE, "golden", "impulse_room_correction_hpf_lpf.wav")   # the reference's own sample IR


@pytest.fixture(scope="module")
def O():
    spec = importlib.util.spec_from_file_location(
        "ir_ingest_oracle", os.path.join(os.path.dirname(HERE), "oracle", "ir_ingest_oracle.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def amd():
    import convopeq_amd
    return convopeq_amd


def write_wav(path, frames, rate, bits, fmt, extensible=False, extra_chunks=True, cut=0, riff_len=None):
    """frames: [n][channels] int or float32 array already in the file's sample type."""
    n, ch = frames.shape
    if fmt == "float":
        payload = frames.astype("<f4").tobytes()
    elif bits == 8:
        payload = frames.astype(np.uint8).tobytes()
    elif bits == 16:
        payload = frames.astype("<i2").tobytes()
    elif bits == 24:
        v = frames.astype(np.int64) & 0xFFFFFF
        payload = b"".join(int(x).to_bytes(3, "little") for x in v.reshape(-1))
    else:
        payload = frames.astype("<i4").tobytes()
    tag = 3 if fmt == "float" else 1
    bpf = ch * bits // 8
    if extensible:
        guid = struct.pack("<IHH", tag, 0, 0x10) + bytes([0x80, 0, 0, 0xAA, 0, 0x38, 0x9B, 0x71])
        fmt_body = struct.pack("<HHIIHHHHI", 0xFFFE, ch, rate, rate * bpf, bpf, bits, 22, bits, 0) + guid
    else:
        fmt_body = struct.pack("<HHIIHH", tag, ch, rate, rate * bpf, bpf, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body
    if extra_chunks:
        chunks += b"fact" + struct.pack("<II", 4, n) + b"odd " + struct.pack("<I", 3) + b"abc\0"   # odd-length chunk is padded
    data = b"data" + struct.pack("<I", len(payload)) + payload
    if cut:
        data = data[:-cut]
    body = b"WAVE" + chunks + data + (b"LIST" + struct.pack("<I", 4) + b"INFO" if extra_chunks else b"")
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body) + cut if riff_len is None else riff_len) + body)


def wav_cases(tmp_path):
    rng = np.random.default_rng(5)
    n = 1003
    out = []
    for name, bits, fmt, ext, ch in [("u8", 8, "int", False, 2), ("s16", 16, "int", False, 2), ("s24", 24, "int", False, 1),
                                     ("s32", 32, "int", False, 2), ("f32", 32, "float", False, 2), ("s24x", 24, "int", True, 3),
                                     ("f32x", 32, "float", True, 2)]:
        if fmt == "float":
            fr = (rng.standard_normal((n, ch)) * 0.4).astype(np.float32)
            fr[5, 0], fr[6, 0], fr[7, 0], fr[n - 1, 0], fr[8, 0] = np.nan, np.inf, 1e-30, -np.inf, 3.5
        elif bits == 8:
            fr = rng.integers(0, 256, (n, ch))
        else:
            lim = 1 << (bits - 1)
            fr = rng.integers(-lim, lim, (n, ch))
            fr[0, 0], fr[1, 0] = -lim, lim - 1
        p = str(tmp_path / f"{name}.wav")
        write_wav(p, fr, 44100 if name == "s16" else 48000, bits, fmt, ext)
        out.append((name, p, bits, fmt))
    return out


def test_oracle_wav_decoding_against_scipy(O, tmp_path):
    """pins the restatement's reader: scipy.io.wavfile is an independent implementation of the container."""
    rate, data = wavfile.read(SAMPLE)
    planes, r = O.load_wav(SAMPLE)
    assert r == rate == 48000 and planes.shape == (2, data.shape[0]) and data.dtype == np.float32
    assert np.array_equal(planes, np.clip(data.T.astype(np.float64), -1, 1))
    for name, p, bits, fmt in wav_cases(tmp_path):
        rate, data = wavfile.read(p)
        data = data.reshape(len(data), -1)
        planes, r = O.load_wav(p)
        assert r == rate and planes.shape == data.T.shape, name
        if fmt == "float":
            ref = data.T.astype(np.float64)
            ref[np.isnan(ref) | (np.abs(ref) < 1e-20)] = 0.0
            body = ref.shape[1] // 4 * 4
            ref[:, body:][np.isinf(ref[:, body:])] = 0.0
            assert np.array_equal(planes, np.clip(ref, -1, 1)), name
        else:
            # scipy: u8 as is, s16 as is, s24 left-justified in int32, s32 as is
            fixed = {8: (data.astype(np.int64) - 128) << 24, 16: data.astype(np.int64) << 16,
                     24: data.astype(np.int64), 32: data.astype(np.int64)}[bits]
            ref = fixed.astype(np.int32).astype(np.float32) * (np.float32(1.0) / np.float32(0x7FFFFFFF))
            assert np.array_equal(planes, np.clip(ref.T.astype(np.float64), -1, 1)), name
            assert np.abs(planes).max() <= 1.0


def test_load_wav_matches_oracle_bit_for_bit(amd, O, tmp_path):
    for name, p, bits, fmt in [("sample", SAMPLE, 32, "float")] + wav_cases(tmp_path):
        got, rate = amd.ir_load_wav(p)
        ref, r = O.load_wav(p)
        assert rate == r and got.shape == ref.shape and np.array_equal(got, ref), name


def test_load_wav_edge_cases(amd, O, tmp_path):
    fr = np.arange(40, dtype=np.int64).reshape(20, 2) * 1000
    # payload shorter than the data chunk says: the missing bytes read as zero, the frame count stays
    p = str(tmp_path / "cut.wav")
    write_wav(p, fr, 48000, 16, "int", extra_chunks=False, cut=13)
    got, _ = amd.ir_load_wav(p)
    ref, _ = O.load_wav(p)
    assert got.shape == (2, 20) and np.array_equal(got, ref) and got[0, -1] == 0.0 and got[1, -3] == 0.0 and got[1, 0] > 0
    # chunk walk stops at the RIFF length: a data chunk behind it is not seen
    p = str(tmp_path / "short_riff.wav")
    write_wav(p, fr, 48000, 16, "int", extra_chunks=False, riff_len=4 + 8 + 16)
    assert O.load_wav(p) is None
    with pytest.raises(amd.CpqError) as e:
        amd.ir_load_wav(p)
    assert e.value.status == -1                  # a valid header without frames
    for name, blob in [("empty.wav", b""), ("junk.wav", b"RIFX" + b"\0" * 64), ("notwave.wav", b"RIFF" + struct.pack("<I", 36) + b"AVI " + b"\0" * 32)]:
        p = str(tmp_path / name)
        open(p, "wb").write(blob)
        with pytest.raises(amd.CpqError) as e:
            amd.ir_load_wav(p)
        assert e.value.status == -5, name
    # 64-bit float and ADPCM payloads: the reference's reader rejects them
    p = str(tmp_path / "f64.wav")
    body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 3, 1, 48000, 48000 * 8, 8, 64) + b"data" + struct.pack("<I", 80) + b"\0" * 80
    open(p, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
    with pytest.raises(amd.CpqError):
        amd.ir_load_wav(p)
    p = str(tmp_path / "adpcm.wav")
    body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 2, 1, 48000, 24000, 1, 4) + b"data" + struct.pack("<I", 80) + b"\0" * 80
    open(p, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
    with pytest.raises(amd.CpqError):
        amd.ir_load_wav(p)
    with pytest.raises(amd.CpqError) as e:
        amd.ir_load_wav(str(tmp_path / "missing.wav"))
    assert e.value.status == -1
    # no frames
    p = str(tmp_path / "noframes.wav")
    write_wav(p, fr[:0], 48000, 16, "int")
    with pytest.raises(amd.CpqError) as e:
        amd.ir_load_wav(p)
    assert e.value.status == -1


def test_frequency_response_gain_closed_forms(amd, O):
    """the estimate against mathematics: a unit impulse in the flat part of the window has |H| = 1 at every bin, a
    bin-centred sinusoid of amplitude A has a peak of A N / 2 (coherent-gain corrected)."""
    n = 1024
    win = np.ones(n)
    taper = 0.5 * (n - 1) * 0.5
    i = np.arange(n, dtype=np.float64)
    lo, hi = i < taper, i > (n - 1) - taper
    win[lo] = 0.5 * (1 + np.cos(2 * np.pi * i[lo] / (0.5 * (n - 1)) - np.pi))
    win[hi] = 0.5 * (1 + np.cos(2 * np.pi * (i[hi] - ((n - 1) - taper)) / (0.5 * (n - 1))))
    h = np.zeros((1, n))
    h[0, 500] = 1.0
    want = 1.0 / win.mean()
    assert abs(O.estimate_max_frequency_response_gain(h) - want) < 1e-12
    assert abs(amd.ir_estimate_max_frequency_response_gain(h) - want) < 1e-12
    a = 0.3
    s = a * np.cos(2 * np.pi * 64 * i / n)[None, :]
    got_o, got_p = O.estimate_max_frequency_response_gain(s), amd.ir_estimate_max_frequency_response_gain(s)
    exact = np.abs(np.fft.rfft(s[0] * win)).max() / win.mean()
    assert exact <= got_o < exact * 1.01 and abs(got_o - a * n / 2) / (a * n / 2) < 0.01      # refined peak >= bin peak
    assert abs(got_p - got_o) <= 1e-12 * got_o
    # degenerate inputs
    assert amd.ir_estimate_max_frequency_response_gain(np.zeros((2, 64))) == 1.0 == O.estimate_max_frequency_response_gain(np.zeros((2, 64)))
    assert amd.ir_estimate_max_frequency_response_gain(np.ones((1, 1))) == 1.0 == O.estimate_max_frequency_response_gain(np.ones((1, 1)))


def scale_close(got, ref):
    assert got["has_scale_factor"] == ref["has_scale_factor"]
    for k in ("scale_factor", "peak_value", "rms_value", "frequency_peak_gain"):
        assert abs(got[k] - ref[k]) <= 1e-12 * max(abs(ref[k]), 1e-300), (k, got[k], ref[k])
    assert abs(got["additional_attenuation_db"] - ref["additional_attenuation_db"]) <= 1e-5


def test_compute_scale_factor_matches_oracle(amd, O):
    rng = np.random.default_rng(11)
    t = np.arange(70000, dtype=np.float64)
    decay = rng.standard_normal((2, 70000)) * np.exp(-t / 9000.0) * 0.05            # > 65536: analysis window capped
    spike = np.zeros((2, 4096)); spike[0, 10] = 0.9; spike[1, 11] = -0.2; spike += rng.standard_normal((2, 4096)) * 1e-4
    tone = (0.2 * np.sin(2 * np.pi * 1000.3 * t[:30000] / 48000.0))[None, :]          # frequency clamp + RMS clamp
    mono_short = rng.standard_normal((1, 37)) * 0.3
    cases = {"decay": decay, "spike": spike, "tone": tone, "mono_short": mono_short, "one": np.array([[0.25]]),
             "silent": np.zeros((2, 100)), "tiny": np.full((1, 8), 1e-12)}
    seen_clamps = set()
    for name, ir in cases.items():
        got, ref = amd.ir_compute_scale_factor(ir), O.compute_scale_factor(ir)
        scale_close(got, ref)
        if ref["additional_attenuation_db"] > 0:
            seen_clamps.add(name)
    assert {"spike", "tone"} <= seen_clamps
    assert amd.ir_compute_scale_factor(cases["silent"])["scale_factor"] == 1.0
    e = float(np.dot(decay[0], decay[0]))
    base = 0.5011872336272722 / math.sqrt(max(e, float(np.dot(decay[1], decay[1]))))
    assert amd.ir_compute_scale_factor(decay)["scale_factor"] <= base * (1 + 1e-12)
    # jump protection: the IR playing now is 40 dB quieter than the new one would be
    quiet = decay * 1.0
    got = amd.ir_compute_scale_factor(spike, current_ir=quiet, current_scale=0.01)
    ref = O.compute_scale_factor(spike, quiet, 0.01)
    scale_close(got, ref)
    assert got["scale_factor"] < amd.ir_compute_scale_factor(spike)["scale_factor"] * 0.5
    # ... and no protection when the levels are comparable
    got2 = amd.ir_compute_scale_factor(spike, current_ir=spike, current_scale=1.0)
    assert got2["scale_factor"] == amd.ir_compute_scale_factor(spike)["scale_factor"]


def test_peak_latency_matches_oracle(amd, O):
    rng = np.random.default_rng(3)
    t = np.arange(20000, dtype=np.float64)
    ir = rng.standard_normal((2, 20000)) * np.exp(-np.abs(t - 700) / 400.0)
    ir[1] = np.roll(ir[1], 900)
    assert amd.ir_estimate_peak_latency(ir) == O.estimate_peak_latency(ir, ir.shape[1])
    assert 600 < amd.ir_estimate_peak_latency(ir[:1]) < 800 < amd.ir_estimate_peak_latency(ir)
    d = np.zeros((1, 512)); d[0, 77] = 1.0
    assert amd.ir_estimate_peak_latency(d) == 77 == O.estimate_peak_latency(d, 512)
    assert amd.ir_estimate_peak_latency(np.zeros((2, 64))) == 0


def test_prepare_matches_oracle_on_the_sample_ir(amd, O):
    ir, rate = amd.ir_load_wav(SAMPLE)
    for secs in (1.0, 0.5, 3.0, 0.1):              # 0.1 s = 4800 samples: the file (8253 frames) is cut and faded
        got = amd.ir_prepare(ir, rate, 48000.0, secs)
        ref = O.prepare(ir, rate, 48000.0, secs)
        assert got["ir"].shape == ref["ir"].shape == (2, int(48000 * float(np.float32(secs))))
        assert np.abs(got["ir"] - ref["ir"]).max() <= 1e-15
        scale_close(got["scale"], ref["scale"])
        assert got["ir_peak_latency"] == ref["ir_peak_latency"]
    # the conditioned IR: starts in the pre-taper (window 0 at sample 0), ends faded to silence, zero-padded
    got = amd.ir_prepare(ir, rate, 48000.0, 1.0)
    kept = O.trimmed_length(ir)
    assert np.all(got["ir"][:, kept:] == 0.0) and abs(got["ir"][0, kept - 1]) < 1e-6
    assert 0 < got["scale"]["scale_factor"] and got["scale"]["has_scale_factor"]


LONG_IR = os.path.join(HERE, "golden", "synthetic_long_ir_20s.wav")   # the reference's 20 s sample IR (stereo, 16-bit, 48 kHz)


def test_long_sample_ir_decodes_and_prepares_like_the_restatement(amd, O):
    """The reference's sampledata/synthetic_long_ir_20s.wav (960000 frames): 16-bit PCM decoding bit-identical to the
    restatement and to scipy's reader; conditioned to 10 s (480000 taps: a three-layer plan at 512-sample blocks) like the
    restatement."""
    rate, data = wavfile.read(LONG_IR)
    got, r = amd.ir_load_wav(LONG_IR)
    ref, r2 = O.load_wav(LONG_IR)
    assert r == r2 == rate == 48000 and got.shape == ref.shape == (2, 960000) and data.dtype == np.int16
    assert np.array_equal(got, ref)
    fixed = (data.astype(np.int64) << 16).astype(np.int32).astype(np.float32) * (np.float32(1.0) / np.float32(0x7FFFFFFF))
    assert np.array_equal(got, np.clip(fixed.T.astype(np.float64), -1, 1))
    p, q = amd.ir_prepare(got, r, 48000.0, 10.0), O.prepare(ref, r2, 48000.0, 10.0)
    assert p["ir"].shape == q["ir"].shape == (2, 480000)
    assert np.abs(p["ir"] - q["ir"]).max() <= 1e-15
    scale_close(p["scale"], q["scale"])
    assert p["ir_peak_latency"] == q["ir_peak_latency"]


def test_prepare_variants_and_errors(amd, O):
    rng = np.random.default_rng(8)
    t = np.arange(30000, dtype=np.float64)
    mono = (rng.standard_normal(30000) * np.exp(-t / 3000.0) * 0.2 + 0.01)[None, :]       # DC offset for the blocker
    mono[0, 25000:] = 0.0                                                                  # trailing silence is trimmed
    got, ref = amd.ir_prepare(mono, 44100.0, 44100.0, 0.5), O.prepare(mono, 44100.0, 44100.0, 0.5)
    assert got["ir"].shape == (1, 22050) and np.abs(got["ir"] - ref["ir"]).max() <= 1e-15
    scale_close(got["scale"], ref["scale"])
    assert got["ir_peak_latency"] == ref["ir_peak_latency"]
    assert abs(got["ir"][0, 5000:20000].mean()) < abs(mono[0, 5000:20000].mean())          # the blocker removed DC
    # against the IR playing now
    cur = rng.standard_normal((2, 2000)) * 1e-4
    got = amd.ir_prepare(mono, 44100.0, 44100.0, 0.5, current_ir=cur, current_scale=1.0)
    ref = O.prepare(mono, 44100.0, 44100.0, 0.5, cur, 1.0)
    scale_close(got["scale"], ref["scale"])
    # all-silent file: one sample is kept, scale factor 1
    got = amd.ir_prepare(np.zeros((2, 100)), 48000.0, 48000.0, 1.0)
    assert got["scale"]["scale_factor"] == 1.0 and not np.any(got["ir"])
    # another rate needs the resampler
    with pytest.raises(amd.CpqError) as e:
        amd.ir_prepare(mono, 44100.0, 48000.0, 1.0)
    assert e.value.status == -5
    with pytest.raises(amd.CpqError):
        amd.ir_prepare(mono, 48000.0, 48000.0, 0.0)
    # length cap of the reference (2^21 samples)
    got = amd.ir_prepare(mono, 768000.0, 768000.0, 3.0)
    assert got["ir"].shape == (1, 2097152)


def test_minimum_phase_conversion(amd, O):
    """PhaseMode::Minimum: the product's own radix-2 FFT against numpy's in the restatement (the cepstral construction
    amplifies FFT rounding near spectral nulls: 1e-9 of the peak is the bar here), and against the mathematics: same
    magnitude response, all energy moved to the front, a minimum-phase input comes back unchanged."""
    ir, rate = amd.ir_load_wav(SAMPLE)
    prep = amd.ir_prepare(ir, rate, 48000.0, 0.25)["ir"]                 # 12000 samples -> 65536-point transforms
    got = amd.ir_convert_to_minimum_phase(prep)
    ref = O.convert_to_minimum_phase(prep)
    assert got.shape == ref.shape == prep.shape
    assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max()
    n = prep.shape[1]
    for ch in range(2):
        a, b = np.abs(np.fft.rfft(prep[ch], 4 * n)), np.abs(np.fft.rfft(got[ch], 4 * n))
        strong = a > 1e-3 * a.max()
        assert np.abs(b[strong] / a[strong] - 1.0).max() < 1e-3          # same magnitude where there is any (time aliasing of the cepstrum aside)
        ea, eb = np.cumsum(prep[ch] ** 2), np.cumsum(got[ch] ** 2)
        assert np.all(eb[:n // 2] >= ea[:n // 2] * (1 - 1e-9) - 1e-15)   # energy arrives no later than in any other phase
    # a decaying exponential is minimum phase already
    t = np.arange(4096, dtype=np.float64)
    e = (0.5 * 0.99 ** t)[None, :]
    assert np.abs(amd.ir_convert_to_minimum_phase(e) - e).max() < 1e-9
    # through cpq_ir_prepare
    a = amd.ir_prepare(ir, rate, 48000.0, 0.25, phase_mode=2)
    b = O.prepare(ir, rate, 48000.0, 0.25, minimum_phase=True)
    assert np.abs(a["ir"] - b["ir"]).max() <= 1e-9 * np.abs(b["ir"]).max()
    assert abs(a["scale"]["scale_factor"] / b["scale"]["scale_factor"] - 1.0) < 1e-8
    assert abs(a["ir_peak_latency"] - b["ir_peak_latency"]) <= 1
    assert a["ir_peak_latency"] < amd.ir_prepare(ir, rate, 48000.0, 0.25)["ir_peak_latency"]
    # a silent IR: the conversion does not validate (peak <= 1e-12) and the IR stays as it is
    z = amd.ir_prepare(np.zeros((1, 300)), 48000.0, 48000.0, 0.01, phase_mode=2)
    assert not np.any(z["ir"])
    with pytest.raises(amd.CpqError) as ex:
        amd.ir_prepare(ir, rate, 48000.0, 0.25, phase_mode=1)            # Mixed: not built
    assert ex.value.status == -5
    with pytest.raises(amd.CpqError) as ex:
        amd.ir_convert_to_minimum_phase(np.ones((1, 2097153)))           # 4 n above the reference's limit
    assert ex.value.status == -5


@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_prepare_random_sweep(amd, O, seed):
    """Seeded sweep of cpq_ir_prepare against the restatement: 1-3 channels, 50..60000 samples, decays, peak positions,
    DC offsets, trailing silence, sample rates, target lengths from 10 ms to 1.2 s, as-is and minimum phase, with and
    without an IR playing now."""
    rng = np.random.default_rng(seed)
    ch = int(rng.integers(1, 4))
    n = int(np.exp(rng.uniform(np.log(50), np.log(60000))))
    rate = float(rng.choice([44100.0, 48000.0, 96000.0]))
    t = np.arange(n, dtype=np.float64)
    peak_at = int(rng.integers(0, max(1, n // 3)))
    ir = rng.standard_normal((ch, n)) * np.exp(-np.abs(t - peak_at) / rng.uniform(5.0, n / 2.0)) * rng.uniform(0.01, 0.9)
    ir += rng.choice([0.0, 0.0, 0.02])                      # DC offset
    ir = np.clip(ir, -1.0, 1.0)
    if rng.random() < 0.4:
        ir[:, int(n * rng.uniform(0.5, 0.95)):] = 0.0       # trailing silence
    secs = float(rng.choice([0.01, 0.05, 0.3, 1.0, 1.2]))
    minimum = bool(rng.random() < 0.4)
    cur = None
    cur_scale = 1.0
    if rng.random() < 0.4:
        cur = rng.standard_normal((int(rng.integers(1, 3)), int(rng.integers(100, 5000)))) * float(rng.choice([1e-4, 0.05, 0.5]))
        cur_scale = float(rng.uniform(0.05, 2.0))
    got = amd.ir_prepare(ir, rate, rate, secs, current_ir=cur, current_scale=cur_scale, phase_mode=2 if minimum else 0)
    ref = O.prepare(ir, rate, rate, secs, cur, cur_scale, minimum_phase=minimum)
    assert got["ir"].shape == ref["ir"].shape
    tol_ir = 1e-9 if minimum else 1e-15
    assert np.abs(got["ir"] - ref["ir"]).max() <= tol_ir * max(1.0, np.abs(ref["ir"]).max()), (seed, minimum)
    assert got["scale"]["has_scale_factor"] == ref["scale"]["has_scale_factor"]
    rel = 1e-8 if minimum else 1e-12
    for k in ("scale_factor", "peak_value", "rms_value", "frequency_peak_gain"):
        assert abs(got["scale"][k] - ref["scale"][k]) <= rel * max(abs(ref["scale"][k]), 1e-300), (seed, k)
    assert abs(got["ir_peak_latency"] - ref["ir_peak_latency"]) <= (1 if minimum else 0)
