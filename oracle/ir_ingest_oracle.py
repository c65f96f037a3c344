"""CPU restatement (numpy) of the reference's IR ingest path -- TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module; the product (convopeq_amd/csrc/ir_ingest.cpp) never does.

PARITY UNPINNED: the reference holds no golden vectors or tests for this path, and its sources for it need JUCE's
generated JuceHeader.h (and MKL / r8brain), so they cannot be built here.  What pins this file: the WAV decoding is
checked against scipy.io.wavfile (an independent reader) on the reference's own sample IR
(sampledata/impulse_room_correction_hpf_lpf.wav, committed as tests/golden/impulse_room_correction_hpf_lpf.wav) and on
synthetic files of every sample format; the analysis functions against closed forms (unit impulse, sinusoid).

Each function follows the reference file:line it names (paths relative to the reference tree), in the reference's own
operation order, so that the product -- written independently, with a different FFT and loop structure -- is compared
against a second reading of the same code.
"""
import math
import struct

import numpy as np

F32 = np.float32


# ------------------------------------------------------------------------------------------------ WAV -> double
def parse_wav(raw):
    """JUCE WavAudioFormatReader constructor (JUCE/modules/juce_audio_formats/codecs/juce_WavAudioFormat.cpp:1209-1346)
    restricted to what decides the sample data: fmt and data chunks of RIFF files."""
    def u(fmt, at):
        size = struct.calcsize(fmt)
        return struct.unpack(fmt, raw[at:at + size].ljust(size, b"\0"))[0]
    if raw[0:4] != b"RIFF" or raw[8:12] != b"WAVE":
        return None
    end = u("<I", 4) + 8
    pos = 12
    info = {"float": False, "bpf": 0, "channels": 0, "bits": 0, "rate": 0, "data_start": 0, "data_len": 0, "frames": 0}
    while pos < end and pos < len(raw):
        ctype, length = raw[pos:pos + 4], u("<I", pos + 4)
        pos += 8
        chunk_end = pos + length + (length & 1)
        if ctype == b"fmt ":
            fmt = u("<H", pos)
            info["channels"], info["rate"], info["bits"] = u("<H", pos + 2), u("<I", pos + 4), u("<H", pos + 14)
            info["bpf"] = info["channels"] * info["bits"] // 8
            if fmt == 3:
                info["float"] = True
            elif fmt == 0xFFFE:
                if length < 40:
                    info["bpf"] = 0
                else:
                    guid = raw[pos + 24:pos + 40]
                    tail = bytes([0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71])
                    if guid == struct.pack("<I", 3) + tail:
                        info["float"] = True
                    elif guid != struct.pack("<I", 1) + tail and guid != bytes(
                            [1, 0, 0, 0, 0x21, 0x07, 0xD3, 0x11, 0x86, 0x44, 0xC8, 0xC1, 0xCA, 0, 0, 0]):
                        info["bpf"] = 0
            elif fmt != 1:
                info["bpf"] = 0
        elif ctype == b"data":
            info["data_start"], info["data_len"] = pos, length
            info["frames"] = length // info["bpf"] if info["bpf"] > 0 else 0
        elif chunk_end <= pos:
            break
        pos = chunk_end
    ok = info["rate"] > 0 and info["channels"] > 0 and info["bpf"] > 0 and info["bits"] <= 32
    return info if ok else None


def sanitize_and_limit(x):
    """applyHighQuality64BitTransform, gain 1 (src/InputBitDepthTransform.h:31-100): 4-wide body, scalar tail."""
    x = np.array(x, dtype=np.float64)
    body = len(x) // 4 * 4
    bad = np.isnan(x) | (np.abs(x) < 1.0e-20)
    bad[body:] |= np.isinf(x[body:])
    x[bad] = 0.0
    return np.clip(x, -1.0, 1.0)


def load_wav(path):
    """LoaderThread::doLoadStep (src/convolver/ConvolverProcessor.LoaderThread.cpp:431-486): planes, sample rate."""
    raw = open(path, "rb").read()
    w = parse_wav(raw)
    if w is None or w["frames"] <= 0:
        return None
    n, ch, bits = w["frames"], w["channels"], w["bits"]
    payload = raw[w["data_start"]:w["data_start"] + n * w["bpf"]].ljust(n * w["bpf"], b"\0")
    b = np.frombuffer(payload, dtype=np.uint8).reshape(n, ch, bits // 8).astype(np.uint32)
    if bits == 32 and w["float"]:
        f = np.frombuffer(payload, dtype="<f4").reshape(n, ch).astype(F32)
    else:
        # ReadHelper<Int32, ...>: left-justified 32-bit integers (juce_WavAudioFormat.cpp:1517-1530), then
        # convertFixedToFloat: float(int) * (1.0f / 0x7fffffff) in float (format/juce_AudioFormatReader.cpp:48-55)
        if bits == 8:
            fixed = ((b[..., 0] - 128) & 0xFF) << 24
        elif bits == 16:
            fixed = (b[..., 0] << 16) | (b[..., 1] << 24)
        elif bits == 24:
            fixed = (b[..., 0] << 8) | (b[..., 1] << 16) | (b[..., 2] << 24)
        else:
            fixed = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16) | (b[..., 3] << 24)
        fixed = fixed.astype(np.uint32).view(np.int32)
        f = fixed.astype(F32) * (F32(1.0) / F32(0x7FFFFFFF))
    planes = np.stack([sanitize_and_limit(f[:, c].astype(np.float64)) for c in range(ch)])
    return planes, float(w["rate"])


# ------------------------------------------------------------------------------------------------ conditioning
def trimmed_length(ir):
    """doTrimStep (LoaderThread.cpp:497-551): last sample above 1e-15 on channel 0 / 1, at least 1."""
    m = np.abs(ir[0]) > 1.0e-15
    if ir.shape[0] > 1:
        m |= np.abs(ir[1]) > 1.0e-15
    idx = np.nonzero(m)[0]
    return max(1, int(idx[-1]) + 1 if len(idx) else 0)


def dc_block(x, rate, cutoff=1.0):
    """UltraHighRateDCBlocker::init + process (src/UltraHighRateDCBlocker.h:78-187)."""
    alpha = []
    for ratio in (1.0 - 0.1, 1.0 + 0.1):
        omega = 2.0 * math.pi * (cutoff * ratio) / rate
        a = -math.expm1(-omega)
        alpha.append(1.0e-6 if (not math.isfinite(a) or a <= 0.0 or a >= 1.0) else a)
    s0 = s1 = 0.0
    a0, a1 = alpha
    out = np.empty(len(x))
    for i, v in enumerate(x.tolist()):
        s0 = s0 + a0 * (v - s0)
        v = v - s0
        s1 = s1 + a1 * (v - s1)
        v = v - s1
        out[i] = v
    return out


def asymmetric_tukey(x):
    """applyAsymmetricTukey (src/convolver/ConvolverProcessor.ResampleAndFallback.cpp:109-197)."""
    n = len(x)
    peak = int(np.argmax(np.abs(x)))
    alpha_pre = 0.05
    alpha_post = max(0.05, min(0.25, 0.05 + 0.033 * (math.log2(float(n)) - 10.0)))
    win = np.ones(n)
    if peak > 0:
        pre = int(math.floor(peak * alpha_pre))
        if pre > 0:
            scale = math.pi / (peak * alpha_pre)
            args = scale * np.arange(pre, dtype=np.float64) + (-math.pi)
            win[:pre] = 0.5 * (1.0 + np.cos(args))
    dist = float(n - 1 - peak)
    if dist > 1.0e-9:
        start = peak + int(math.ceil(dist * (1.0 - alpha_post)))
        length = n - start
        if length > 0:
            scale = (math.pi / alpha_post) / dist
            offset = (math.pi / alpha_post) * ((float(start) - float(peak)) / dist - (1.0 - alpha_post))
            win[start:] = 0.5 * (1.0 + np.cos(scale * np.arange(length, dtype=np.float64) + offset))
    return x * win


# ------------------------------------------------------------------------------------------------ analysis
def simple_real_fft(data, n):
    """simpleRealFFT (src/IRAnalyzer.cpp:13-58): radix-2 DIT with the running-product twiddle; returns bins 0..n/2."""
    buf = np.array(data[:n], dtype=np.complex128)
    j = 0
    perm = np.arange(n)
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j ^= bit
        if i < j:
            perm[i], perm[j] = perm[j], perm[i]
    buf = buf[perm]
    length = 2
    while length <= n:
        ang = -2.0 * math.pi / float(length)
        w = complex(math.cos(ang), math.sin(ang))
        tw = np.empty(length // 2, dtype=np.complex128)
        t = complex(1.0, 0.0)
        for k in range(length // 2):
            tw[k] = t
            t = complex(t.real * w.real - t.imag * w.imag, t.real * w.imag + t.imag * w.real)
        v = buf.reshape(n // length, length)
        hi = v[:, length // 2:]
        prod = (tw.real * hi.real - tw.imag * hi.imag) + 1j * (tw.real * hi.imag + tw.imag * hi.real)
        lo = v[:, :length // 2].copy()
        v[:, :length // 2] = lo + prod
        v[:, length // 2:] = lo - prod
        length <<= 1
    return buf[:n // 2 + 1]


def estimate_max_frequency_response_gain(ir):
    """IRAnalyzer::estimateMaxFrequencyResponseGain (src/IRAnalyzer.cpp:63-155)."""
    ir = np.atleast_2d(ir)
    channels, samples = ir.shape
    if samples <= 0 or channels <= 0:
        return 1.0
    copy_len = min(samples, 65536)
    n = 1
    while n < copy_len:
        n <<= 1
    if n < 2:
        return 1.0
    alpha = 0.5
    taper = alpha * float(n - 1) * 0.5
    win = np.ones(n)
    for i in range(n):
        t = float(i)
        if t < taper:
            win[i] = 0.5 * (1.0 + math.cos((2.0 * math.pi * t) / (alpha * float(n - 1)) - math.pi))
        elif t > float(n - 1) - taper:
            win[i] = 0.5 * (1.0 + math.cos((2.0 * math.pi * (t - (float(n - 1) - taper))) / (alpha * float(n - 1))))
    win_sum = 0.0
    for i in range(copy_len):
        win_sum += win[i]
    win_mean = win_sum / float(copy_len)
    if win_mean < 1e-18:
        return 1.0
    best = 0.0
    for ch in range(channels):
        frame = np.zeros(n)
        frame[:copy_len] = ir[ch, :copy_len] * win[:copy_len]
        spec = simple_real_fft(frame, n)
        bins = n // 2
        mags = np.sqrt(spec.real * spec.real + spec.imag * spec.imag)
        mags[0] = abs(spec[0].real)
        mags[bins] = abs(spec[bins].real)
        best = max(best, float(mags.max()))
        for b in range(1, bins - 1):
            lo, mid, hi = mags[b - 1], mags[b], mags[b + 1]
            if mid > lo and mid > hi and mid > 1e-18 and lo > 1e-18 and hi > 1e-18:
                l_lo, l_mid, l_hi = math.log(lo), math.log(mid), math.log(hi)
                denom = l_lo - 2.0 * l_mid + l_hi
                if abs(denom) > 1e-18:
                    delta = 0.5 * (l_lo - l_hi) / denom
                    best = max(best, mid * math.exp(-delta * (l_mid - l_lo)))
    best /= win_mean
    return best if best > 1e-18 else 1.0


def _peak_rms(ir, scale):
    v = np.atleast_2d(ir) * scale
    if v.size == 0:
        return 0.0, 0.0
    energy = 0.0
    for ch in range(v.shape[0]):              # one running sum over channel-major samples, as the reference accumulates
        energy = float(np.cumsum(np.concatenate([[energy], v[ch] * v[ch]]))[-1])
    return float(np.max(np.abs(v))), math.sqrt(energy / float(v.shape[0] * v.shape[1]))


def compute_scale_factor(ir, current_ir=None, current_scale=1.0):
    """IRConverter::computeScaleFactor (src/IRConverter.cpp:17-196).  Returns dict like cpq_ir_scale."""
    ir = np.atleast_2d(np.asarray(ir, dtype=np.float64))
    res = {"scale_factor": 1.0, "has_scale_factor": False, "additional_attenuation_db": 0.0, "peak_value": 0.0,
           "rms_value": 0.0, "frequency_peak_gain": 1.0}
    channels, samples = ir.shape
    scale = 1.0
    if samples > 0 and channels > 0:
        max_energy = 0.0
        for ch in range(channels):
            e = float(np.dot(ir[ch], ir[ch]))
            if math.isfinite(e) and e > 1.0e-18:
                max_energy = max(max_energy, e)
        if max_energy > 1.0e-18 and math.isfinite(max_energy):
            scale = (1.0 / math.sqrt(max_energy)) * 0.5011872336272722
    if scale <= 0.0 or not math.isfinite(scale):
        return res
    res["scale_factor"], res["has_scale_factor"] = scale, True
    peak, rms = _peak_rms(ir, 1.0)
    fpg = estimate_max_frequency_response_gain(ir)
    res["peak_value"], res["rms_value"], res["frequency_peak_gain"] = peak, rms, fpg
    p_db = r_db = f_db = 0.0
    if peak * scale > 0.5:
        c = 0.5 / (peak * scale)
        res["scale_factor"] *= c
        scale *= c
        p_db = -20.0 * math.log10(c)
    if rms * scale > 0.25:
        c = 0.25 / (rms * scale)
        res["scale_factor"] *= c
        r_db = -20.0 * math.log10(c)
    if fpg > 1.41:
        c = 1.41 / fpg
        res["scale_factor"] *= c
        f_db = -20.0 * math.log10(c)
    res["additional_attenuation_db"] = float(F32(p_db + r_db + f_db))
    if current_ir is not None and np.atleast_2d(current_ir).size > 0:
        c_peak, c_rms = _peak_rms(current_ir, current_scale)
        n_peak, n_rms = _peak_rms(ir, res["scale_factor"])
        peak_jump = c_peak > 1.0e-9 and n_peak > c_peak * 4.0 and n_peak > 0.5
        rms_jump = c_rms > 1.0e-9 and n_rms > c_rms * 4.0 and n_rms > 0.25
        if peak_jump or rms_jump:
            by_peak = by_rms = math.inf
            if n_peak > 1.0e-12 and c_peak > 1.0e-12:
                by_peak = (c_peak * 4.0) / n_peak
            if n_rms > 1.0e-12 and c_rms > 1.0e-12:
                by_rms = (c_rms * 4.0) / n_rms
            ratio = min(by_peak, by_rms)
            if math.isfinite(ratio) and 0.0 < ratio < 1.0:
                res["scale_factor"] *= ratio
    return res


def estimate_peak_latency(ir, length):
    """LoaderThread::estimatePeakLatencySamples (LoaderThread.cpp:149-209)."""
    ir = np.atleast_2d(ir)
    if length <= 0:
        return 0
    max_centroid = 0.0
    for ch in range(ir.shape[0]):
        e = ir[ch, :length] * ir[ch, :length]
        cum = np.cumsum(e)                      # sequential running sum, as the reference accumulates
        total = float(cum[-1])
        if total < 1e-12:
            continue
        hit = np.nonzero(cum >= total * 0.999)[0]
        cutoff = int(hit[0]) if len(hit) else length - 1
        sum_e = float(np.cumsum(e[:cutoff + 1])[-1])
        sum_w = float(np.cumsum(np.arange(cutoff + 1, dtype=np.float64) * e[:cutoff + 1])[-1])
        centroid = sum_w / sum_e if sum_e > 0.0 else 0.0
        max_centroid = max(max_centroid, centroid)
    lat = int(math.floor(max_centroid + 0.5))
    return min(max(lat, 0), length - 1)


def convert_to_minimum_phase(ir):
    """ConvolverProcessorInternal::convertToMinimumPhase (src/convolver/ConvolverProcessor.ResampleAndFallback.cpp:333-469),
    with numpy's FFT in place of MKL's.  None where the reference gives up."""
    ir = np.atleast_2d(np.asarray(ir, dtype=np.float64))
    channels, n = ir.shape
    if n <= 0 or channels < 1:
        return None
    size = 1
    while size < n * 4:
        size <<= 1
    if size > 8388608:
        return None
    out = np.empty((channels, n))
    half = size // 2
    for ch in range(channels):
        z = np.fft.fft(ir[ch], size)
        mag = np.maximum(np.abs(z), 1.0e-300)
        c = np.fft.ifft(np.log(mag).astype(np.complex128))
        c[0] = c[0].real
        c[1:half] = c[1:half].real * 2.0
        c[half] = c[half].real
        c[half + 1:] = 0.0
        z = np.fft.fft(c)
        re, im = np.clip(z.real, -50.0, 50.0), np.clip(z.imag, -50.0, 50.0)
        m = np.exp(re)
        e = m * np.cos(im) + 1j * (m * np.sin(im))
        if not (np.all(np.isfinite(e.real)) and np.all(np.isfinite(e.imag))):
            return None
        y = np.fft.ifft(e).real[:n].copy()
        if not np.all(np.isfinite(y)):
            return None
        y[np.abs(y) < 1.0e-18] = 0.0
        out[ch] = y
    return out


def prepare(ir, ir_rate, sample_rate, target_ir_length_sec=1.0, current_ir=None, current_scale=1.0, minimum_phase=False):
    """doTrimStep + doTransformStep in PhaseMode::AsIs + buildConvolverFromTrimmed's latency
    (LoaderThread.cpp:490-641, 696-709, 220)."""
    ir = np.atleast_2d(np.asarray(ir, dtype=np.float64))
    if abs(ir_rate - sample_rate) > 1e-6:
        raise ValueError("resampling (r8brain) is not restated")
    kept = trimmed_length(ir)
    work = []
    for ch in range(ir.shape[0]):
        w = ir[ch, :kept].copy()
        if ir_rate > 0.0:
            w = dc_block(w, ir_rate)
        work.append(asymmetric_tukey(w))
    target = int(ir_rate * float(F32(target_ir_length_sec)))
    target = max(1, min(target, 2097152))
    out = np.zeros((ir.shape[0], target))
    copy = min(target, kept)
    max_fade = max(256, int(math.floor(sample_rate * 0.080 + 0.5)))
    fade = int(math.floor(float(copy) * 0.02 + 0.5))
    fade = max(256, min(max_fade, fade))
    fade = max(0, min(fade, copy - 1))
    for ch in range(ir.shape[0]):
        out[ch, :copy] = work[ch][:copy]
        if fade > 0:                            # AudioBuffer::applyGainRamp: running gain, increment (0 - 1) / fade
            g, inc = 1.0, (0.0 - 1.0) / float(fade)
            for i in range(copy - fade, copy):
                out[ch, i] *= g
                g += inc
    if minimum_phase:                           # doTransformStep (LoaderThread.cpp:652-680): kept only when it validates
        mp = convert_to_minimum_phase(out)
        if mp is not None and np.all(np.isfinite(mp)) and np.abs(mp).max() > 1.0e-12:
            out = mp
    scale = compute_scale_factor(out, current_ir, current_scale)
    if not scale["has_scale_factor"]:
        scale["scale_factor"] = 1.0
    return {"ir": out, "sample_rate": ir_rate, "scale": scale, "ir_peak_latency": estimate_peak_latency(out, target)}
