// ir_ingest.cpp -- host side of the IR ingest path: WAV file -> fp64 planes -> conditioned, trimmed IR +
// scale factor + peak latency, i.e. everything between an IR file and cpq_engine_set_impulse().
//
// One-off set-up work on a few MB: it stays on the host (the reference runs it on a loader thread), nothing here is on
// the audio path.  Reference behaviour restated (paths relative to the reference tree):
//   * file -> float -> double                src/convolver/ConvolverProcessor.LoaderThread.cpp:431-486 (JUCE WAV reader:
//                                            JUCE/modules/juce_audio_formats/codecs/juce_WavAudioFormat.cpp:1209-1346,
//                                            1517-1530; fixed -> float: format/juce_AudioFormatReader.cpp:48-55),
//                                            sanitise + clamp src/InputBitDepthTransform.h:31-100
//   * trailing-silence trim                  LoaderThread.cpp:497-551
//   * DC blocker                             src/UltraHighRateDCBlocker.h:78-187 (1 Hz, two one-pole sections at -/+10 %)
//   * asymmetric Tukey window                src/convolver/ConvolverProcessor.ResampleAndFallback.cpp:109-197
//   * target length + fade-out               LoaderThread.cpp:619-636, StateAndUI.cpp:942-957
//   * scale factor                           src/IRConverter.cpp:17-196, src/IRAnalyzer.cpp:63-155
//   * peak latency (energy centroid)         LoaderThread.cpp:149-209
// Not restated: sample-rate conversion (third-party r8brain, not in the reference tree) -- an IR at another rate is
// refused with CPQ_ERR_UNSUPPORTED -- and the mixed phase transform (PhaseMode::AsIs and Minimum only).
//   * minimum phase                          src/convolver/ConvolverProcessor.ResampleAndFallback.cpp:333-469
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <vector>

#include "convopeq_mi355x.h"

namespace {

constexpr double kPi = 3.14159265358979323846;

// ---------------------------------------------------------------------------------------------- buffers
bool allocBuffer(cpq_ir_buffer* b, int channels, int samples, double rate)
{
    b->n_channels = channels;
    b->n_samples = samples;
    b->sample_rate = rate;
    b->data = static_cast<double*>(std::calloc((size_t)channels * (size_t)samples, sizeof(double)));
    return b->data != nullptr;
}

inline const double* plane(const cpq_ir_buffer* b, int ch) { return b->data + (size_t)ch * (size_t)b->n_samples; }
inline double* plane(cpq_ir_buffer* b, int ch) { return b->data + (size_t)ch * (size_t)b->n_samples; }

bool validBuffer(const cpq_ir_buffer* b)
{
    return b && b->data && b->n_channels > 0 && b->n_samples > 0;
}

// ---------------------------------------------------------------------------------------------- WAV
struct ByteReader {
    const std::vector<unsigned char>& bytes;
    uint64_t pos = 0;
    bool exhausted() const { return pos >= bytes.size(); }
    uint64_t take(int n)          // little-endian, bytes past the end read as zero (as a short InputStream::read does)
    {
        uint64_t v = 0;
        for (int i = 0; i < n; ++i) {
            const uint64_t p = pos + (uint64_t)i;
            if (p < bytes.size()) v |= (uint64_t)bytes[p] << (8 * i);
        }
        pos += (uint64_t)n;
        return v;
    }
};

constexpr uint32_t fourcc(const char (&s)[5])
{
    return (uint32_t)(unsigned char)s[0] | ((uint32_t)(unsigned char)s[1] << 8) | ((uint32_t)(unsigned char)s[2] << 16) |
           ((uint32_t)(unsigned char)s[3] << 24);
}

struct WavInfo {
    uint32_t channels = 0, bits = 0;
    double rate = 0.0;
    int bytesPerFrame = 0;
    bool isFloat = false;
    uint64_t dataStart = 0;
    int64_t dataLength = 0, frames = 0;
};

// GUID tails of WAVE_FORMAT_EXTENSIBLE sub-formats: {type}-0000-0010-8000-00aa00389b71 (PCM 1, IEEE float 3) and
// the ambisonic B-format PCM GUID {1}-0721-11d3-8644-c8c1ca000000
bool parseWav(const std::vector<unsigned char>& bytes, WavInfo& w)
{
    ByteReader in { bytes };
    const uint32_t first = (uint32_t)in.take(4);
    uint64_t end = 0;
    bool rf64 = false;
    if (first == fourcc("RF64")) { in.take(4); rf64 = true; }
    else if (first == fourcc("RIFF")) { const uint64_t len = in.take(4); end = len + in.pos; }
    else return false;
    const uint64_t riffStart = in.pos;
    if ((uint32_t)in.take(4) != fourcc("WAVE")) return false;
    if (rf64) {
        if ((uint32_t)in.take(4) == fourcc("ds64")) {
            const uint32_t length = (uint32_t)in.take(4);
            if (length < 28) return false;
            const uint64_t chunkEnd = in.pos + length + (length & 1u);
            end = in.take(8) + riffStart;
            w.dataLength = (int64_t)in.take(8);
            in.pos = chunkEnd;
        }
    }
    while (in.pos < end && !in.exhausted()) {
        const uint32_t type = (uint32_t)in.take(4);
        const uint32_t length = (uint32_t)in.take(4);
        uint64_t chunkEnd = in.pos + length + (length & 1u);
        if (type == fourcc("fmt ")) {
            const unsigned format = (unsigned)in.take(2);
            w.channels = (uint32_t)in.take(2);
            const uint32_t intRate = (uint32_t)in.take(4);
            w.rate = (double)intRate;
            const uint32_t bytesPerSec = (uint32_t)in.take(4);
            in.take(2);
            w.bits = (uint32_t)(int)(int16_t)in.take(2);
            if (w.bits > 64 && intRate > 0) {
                w.bytesPerFrame = (int)(bytesPerSec / intRate);
                if (w.channels > 0) w.bits = 8u * (unsigned)w.bytesPerFrame / w.channels;
            } else {
                w.bytesPerFrame = (int)(w.channels * w.bits / 8);
            }
            if (format == 3) {
                w.isFloat = true;
            } else if (format == 0xfffe) {
                if (length < 40) {
                    w.bytesPerFrame = 0;
                } else {
                    in.take(4);          // cbSize + valid bits
                    in.take(4);          // channel mask
                    const uint32_t d1 = (uint32_t)in.take(4);
                    const uint32_t d2 = (uint32_t)in.take(2), d3 = (uint32_t)in.take(2);
                    unsigned char d4[8];
                    for (auto& b : d4) b = (unsigned char)in.take(1);
                    static const unsigned char kWaveTail[8] = { 0x80, 0x00, 0x00, 0xaa, 0x00, 0x38, 0x9b, 0x71 };
                    static const unsigned char kAmbiTail[8] = { 0x86, 0x44, 0xc8, 0xc1, 0xca, 0x00, 0x00, 0x00 };
                    const bool waveGuid = d2 == 0x0000 && d3 == 0x0010 && std::memcmp(d4, kWaveTail, 8) == 0;
                    const bool ambiGuid = d1 == 1 && d2 == 0x0721 && d3 == 0x11d3 && std::memcmp(d4, kAmbiTail, 8) == 0;
                    if (waveGuid && d1 == 3) w.isFloat = true;
                    else if (!(waveGuid && d1 == 1) && !ambiGuid) w.bytesPerFrame = 0;
                }
            } else if (format != 1) {
                w.bytesPerFrame = 0;     // compressed payloads (incl. Ogg-in-WAV) are not IR material
            }
        } else if (type == fourcc("data")) {
            if (rf64) {
                if (w.dataLength > 0) chunkEnd = in.pos + (uint64_t)w.dataLength + ((uint64_t)w.dataLength & 1u);
            } else {
                w.dataLength = (int64_t)length;
            }
            w.dataStart = in.pos;
            w.frames = w.bytesPerFrame > 0 ? w.dataLength / w.bytesPerFrame : 0;
        } else if (chunkEnd <= in.pos) {
            break;
        }
        in.pos = chunkEnd;
    }
    return w.rate > 0.0 && w.channels > 0 && w.bytesPerFrame > 0 && w.bits <= 32;
}

// one sample of the JUCE reader's AudioBuffer<float>: integers are left-justified to 32 bits, converted to float
// (round to nearest) and multiplied by 1 / 0x7fffffff in float arithmetic; float32 payloads pass through.
float decodeSample(const unsigned char* p, uint32_t bits, bool isFloat)
{
    int32_t fixed;
    switch (bits) {
    case 8:  fixed = (int32_t)(((uint32_t)p[0] - 128u) << 24); break;
    case 16: fixed = (int32_t)(((uint32_t)p[0] << 16) | ((uint32_t)p[1] << 24)); break;
    case 24: fixed = (int32_t)(((uint32_t)p[0] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 24)); break;
    case 32: {
        const uint32_t u = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
        if (isFloat) { float f; std::memcpy(&f, &u, 4); return f; }
        fixed = (int32_t)u;
        break;
    }
    default: return 0.0f;                 // the reader leaves other widths untouched (zero-initialised)
    }
    constexpr float kFixedToFloat = 1.0f / static_cast<float>(0x7fffffff);
    return static_cast<float>(fixed) * kFixedToFloat;
}

// applyHighQuality64BitTransform with gain 1: NaN and |v| < 1e-20 -> 0, clamp to [-1, 1].  An infinity clamps to +-1
// in the reference's 4-wide body and becomes 0 in its scalar tail (the last n % 4 samples).
inline double sanitizeAndLimit(double v, bool scalarTail)
{
    const bool inf = std::isinf(v);
    if (v != v || std::fabs(v) < 1.0e-20 || (inf && scalarTail)) v = 0.0;
    return std::min(1.0, std::max(-1.0, v));
}

// ---------------------------------------------------------------------------------------------- conditioning
int trimmedLength(const cpq_ir_buffer* b)
{
    // last sample above 1e-15 on channel 0 or 1 (further channels are not looked at), at least one sample stays
    const double* c0 = plane(b, 0);
    const double* c1 = b->n_channels > 1 ? plane(b, 1) : nullptr;
    int keep = 0;
    for (int i = b->n_samples - 1; i >= 0; --i)
        if (std::fabs(c0[i]) > 1.0e-15 || (c1 && std::fabs(c1[i]) > 1.0e-15)) { keep = i + 1; break; }
    return std::max(1, keep);
}

void dcBlock(double* x, int n, double rate, double cutoffHz)
{
    double alpha[2] = { 1.0e-6, 1.0e-6 };
    if (std::isfinite(rate) && rate > 0.0 && std::isfinite(cutoffHz) && cutoffHz > 0.0) {
        const double ratio[2] = { 1.0 - 0.1, 1.0 + 0.1 };
        for (int i = 0; i < 2; ++i) {
            const double omega = 2.0 * kPi * (cutoffHz * ratio[i]) / rate;
            const double a = -std::expm1(-omega);
            alpha[i] = (!std::isfinite(a) || a <= 0.0 || a >= 1.0) ? 1.0e-6 : a;
        }
    }
    double s0 = 0.0, s1 = 0.0;
    for (int i = 0; i < n; ++i) {
        double v = x[i];
        s0 = s0 + alpha[0] * (v - s0);
        v = v - s0;
        s1 = s1 + alpha[1] * (v - s1);
        v = v - s1;
        x[i] = v;
    }
}

void asymmetricTukey(double* x, int n)
{
    if (n <= 0) return;
    int peak = 0;                                    // first sample of largest magnitude
    for (int i = 1; i < n; ++i)
        if (std::fabs(x[peak]) < std::fabs(x[i])) peak = i;
    const double alphaPre = 0.05;
    const double alphaPost = std::max(0.05, std::min(0.25, 0.05 + 0.033 * (std::log2((double)n) - 10.0)));
    if (peak > 0) {
        const int len = (int)std::floor(peak * alphaPre);
        const double scale = kPi / (peak * alphaPre);
        for (int i = 0; i < len; ++i) x[i] *= 0.5 * (1.0 + std::cos(scale * (double)i + (-kPi)));
    }
    const double toEnd = (double)(n - 1 - peak);
    if (toEnd > 1.0e-9) {
        const int start = peak + (int)std::ceil(toEnd * (1.0 - alphaPost));
        const double scale = (kPi / alphaPost) / toEnd;
        const double offset = (kPi / alphaPost) * (((double)start - (double)peak) / toEnd - (1.0 - alphaPost));
        for (int i = start; i < n; ++i) x[i] *= 0.5 * (1.0 + std::cos(scale * (double)(i - start) + offset));
    }
}

// ---------------------------------------------------------------------------------------------- analysis
// in-place complex FFT, N a power of two: iterative radix-2 with exact per-index twiddles; inverse scales by 1 / N
void fftInPlace(std::vector<std::complex<double>>& a, bool inverse)
{
    const size_t N = a.size();
    int lg = 0;
    while (((size_t)1 << lg) < N) ++lg;
    for (size_t i = 0; i < N; ++i) {
        size_t r = 0;
        for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1u) << (lg - 1 - b);
        if (r > i) std::swap(a[i], a[r]);
    }
    std::vector<std::complex<double>> tw(std::max<size_t>(1, N / 2));
    for (size_t k = 0; k < N / 2; ++k) {
        const double ang = (inverse ? 2.0 : -2.0) * kPi * (double)k / (double)N;
        tw[k] = std::complex<double>(std::cos(ang), std::sin(ang));
    }
    for (size_t half = 1; half < N; half <<= 1) {
        const size_t stride = N / (2 * half);
        for (size_t base = 0; base < N; base += 2 * half)
            for (size_t j = 0; j < half; ++j) {
                const auto w = tw[j * stride];
                const auto& hi = a[base + j + half];
                const std::complex<double> t(w.real() * hi.real() - w.imag() * hi.imag(),
                                             w.real() * hi.imag() + w.imag() * hi.real());
                const auto u = a[base + j];
                a[base + j] = u + t;
                a[base + j + half] = u - t;
            }
    }
    if (inverse) {
        const double inv = 1.0 / (double)N;
        for (auto& v : a) v = std::complex<double>(v.real() * inv, v.imag() * inv);
    }
}

// magnitudes |X[0..N/2]| of a real sequence (N a power of two)
void magnitudeSpectrum(const std::vector<double>& x, int N, std::vector<double>& mags)
{
    std::vector<std::complex<double>> a((size_t)N);
    for (int i = 0; i < N; ++i) a[(size_t)i] = std::complex<double>(x[(size_t)i], 0.0);
    fftInPlace(a, false);
    mags.resize((size_t)N / 2 + 1);
    mags[0] = std::fabs(a[0].real());
    mags[(size_t)N / 2] = std::fabs(a[(size_t)N / 2].real());
    for (int k = 1; k < N / 2; ++k)
        mags[(size_t)k] = std::sqrt(a[(size_t)k].real() * a[(size_t)k].real() + a[(size_t)k].imag() * a[(size_t)k].imag());
}

// convertToMinimumPhase (src/convolver/ConvolverProcessor.ResampleAndFallback.cpp:333-469) for one channel: the
// homomorphic construction -- log magnitude -> real cepstrum -> folded onto the causal side -> exponentiated spectrum --
// at 4x zero padding.  false: the reference gives up (non-finite intermediate values).
bool minimumPhase(const double* src, int n, int fftSize, double* dst)
{
    std::vector<std::complex<double>> z((size_t)fftSize);
    for (int i = 0; i < n; ++i) z[(size_t)i] = std::complex<double>(src[i], 0.0);
    fftInPlace(z, false);
    for (auto& v : z) v = std::complex<double>(std::log(std::max(std::hypot(v.real(), v.imag()), 1.0e-300)), 0.0);
    fftInPlace(z, true);
    const size_t half = (size_t)fftSize / 2;
    z[0] = std::complex<double>(z[0].real(), 0.0);
    for (size_t i = 1; i < half; ++i) z[i] = std::complex<double>(z[i].real() * 2.0, 0.0);
    z[half] = std::complex<double>(z[half].real(), 0.0);
    for (size_t i = half + 1; i < (size_t)fftSize; ++i) z[i] = std::complex<double>(0.0, 0.0);
    fftInPlace(z, false);
    for (auto& v : z) {
        const double re = std::min(50.0, std::max(-50.0, v.real())), im = std::min(50.0, std::max(-50.0, v.imag()));
        const double m = std::exp(re);
        v = std::complex<double>(m * std::cos(im), m * std::sin(im));
        if (!std::isfinite(v.real()) || !std::isfinite(v.imag())) return false;
    }
    fftInPlace(z, true);
    for (int i = 0; i < n; ++i) {
        double v = z[(size_t)i].real();
        if (!std::isfinite(v)) return false;
        if (std::fabs(v) < 1.0e-18) v = 0.0;
        dst[i] = v;
    }
    return true;
}

int nextPow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

double maxFrequencyResponseGain(const double* const* ir, int channels, int samples, double gain)
{
    if (samples <= 0 || channels <= 0) return 1.0;
    const int copyLen = std::min(samples, 65536);
    const int N = nextPow2(copyLen);
    if (N < 2) return 1.0;
    // Tukey window (alpha 0.5) over the FFT size; coherent gain = mean over the copied part
    const double alpha = 0.5;
    const double taper = alpha * (double)(N - 1) * 0.5;
    std::vector<double> win((size_t)N);
    for (int i = 0; i < N; ++i) {
        const double t = (double)i;
        if (t < taper) win[(size_t)i] = 0.5 * (1.0 + std::cos((2.0 * kPi * t) / (alpha * (double)(N - 1)) - kPi));
        else if (t > (double)(N - 1) - taper)
            win[(size_t)i] = 0.5 * (1.0 + std::cos((2.0 * kPi * (t - ((double)(N - 1) - taper))) / (alpha * (double)(N - 1))));
        else win[(size_t)i] = 1.0;
    }
    double winSum = 0.0;
    for (int i = 0; i < copyLen; ++i) winSum += win[(size_t)i];
    const double winMean = winSum / (double)copyLen;
    if (winMean < 1e-18) return 1.0;

    double best = 0.0;
    std::vector<double> frame((size_t)N), mags;
    for (int ch = 0; ch < channels; ++ch) {
        std::fill(frame.begin(), frame.end(), 0.0);
        for (int i = 0; i < copyLen; ++i) frame[(size_t)i] = (ir[ch][i] * gain) * win[(size_t)i];
        magnitudeSpectrum(frame, N, mags);
        const int bins = N / 2;
        for (int b = 0; b <= bins; ++b) best = std::max(best, mags[(size_t)b]);
        // log-parabolic (Gaussian) refinement of interior local maxima
        for (int b = 1; b < bins - 1; ++b) {
            const double lo = mags[(size_t)b - 1], mid = mags[(size_t)b], hi = mags[(size_t)b + 1];
            if (mid > lo && mid > hi && mid > 1e-18 && lo > 1e-18 && hi > 1e-18) {
                const double lLo = std::log(lo), lMid = std::log(mid), lHi = std::log(hi);
                const double denom = lLo - 2.0 * lMid + lHi;
                if (std::fabs(denom) > 1e-18) {
                    const double delta = 0.5 * (lLo - lHi) / denom;
                    best = std::max(best, mid * std::exp(-delta * (lMid - lLo)));
                }
            }
        }
    }
    best /= winMean;
    return best > 1e-18 ? best : 1.0;
}

struct PeakRms { double peak = 0.0, rms = 0.0; };
PeakRms peakAndRms(const double* const* ir, int channels, int samples, double scale)
{
    PeakRms r;
    if (channels <= 0 || samples <= 0) return r;
    double energy = 0.0;
    for (int ch = 0; ch < channels; ++ch)
        for (int i = 0; i < samples; ++i) {
            const double v = ir[ch][i] * scale;
            r.peak = std::max(r.peak, std::fabs(v));
            energy += v * v;
        }
    r.rms = std::sqrt(energy / (double)(channels * samples));
    return r;
}

int peakLatency(const double* const* ir, int channels, int length)
{
    if (length <= 0 || channels <= 0) return 0;
    double maxCentroid = 0.0;
    for (int ch = 0; ch < channels; ++ch) {
        const double* d = ir[ch];
        double total = 0.0;
        for (int i = 0; i < length; ++i) total += d[i] * d[i];
        if (total < 1e-12) continue;
        double run = 0.0;
        int cutoff = length - 1;
        for (int i = 0; i < length; ++i) {
            run += d[i] * d[i];
            if (run >= total * 0.999) { cutoff = i; break; }
        }
        double sumE = 0.0, sumW = 0.0;
        for (int i = 0; i <= cutoff; ++i) {
            const double e = d[i] * d[i];
            sumE += e;
            sumW += (double)i * e;
        }
        const double centroid = sumE > 0.0 ? sumW / sumE : 0.0;
        maxCentroid = std::max(maxCentroid, centroid);
    }
    const int lat = (int)std::floor(maxCentroid + 0.5);
    return std::min(std::max(lat, 0), length - 1);
}

std::vector<const double*> planes(const cpq_ir_buffer* b)
{
    std::vector<const double*> p((size_t)b->n_channels);
    for (int c = 0; c < b->n_channels; ++c) p[(size_t)c] = plane(b, c);
    return p;
}

}  // namespace

extern "C" {

void cpq_ir_buffer_free(cpq_ir_buffer* b)
{
    if (!b) return;
    std::free(b->data);
    b->data = nullptr;
    b->n_channels = b->n_samples = 0;
}

int32_t cpq_ir_load_wav(const char* path, cpq_ir_buffer* out)
{
    if (!path || !out) return CPQ_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof(*out));
    std::FILE* f = std::fopen(path, "rb");
    if (!f) return CPQ_ERR_INVALID_ARG;
    std::vector<unsigned char> bytes;
    {
        unsigned char chunk[1 << 16];
        size_t got;
        try {
            while ((got = std::fread(chunk, 1, sizeof(chunk), f)) > 0) bytes.insert(bytes.end(), chunk, chunk + got);
        } catch (const std::bad_alloc&) { std::fclose(f); return CPQ_ERR_OOM; }
    }
    std::fclose(f);
    WavInfo w;
    if (!parseWav(bytes, w)) return CPQ_ERR_UNSUPPORTED;          // "Unsupported audio format or corrupted file"
    if (w.frames <= 0 || w.frames > (int64_t)std::numeric_limits<int>::max()) return CPQ_ERR_INVALID_ARG;
    // The reference zero-fills frames the file does not hold.  Kept for short files, but a data chunk that claims far
    // more than the file contains (a corrupted length field) is refused rather than allocated.
    const uint64_t claimedEnd = w.dataStart + (uint64_t)w.frames * (uint64_t)w.bytesPerFrame;
    if (claimedEnd > 2 * (uint64_t)bytes.size() + (1u << 20)) return CPQ_ERR_UNSUPPORTED;
    if (!allocBuffer(out, (int)w.channels, (int)w.frames, w.rate)) return CPQ_ERR_OOM;
    const int bytesPerSample = (int)w.bits / 8;
    unsigned char pad[4];
    for (int64_t i = 0; i < w.frames; ++i)
        for (uint32_t ch = 0; ch < w.channels; ++ch) {
            const uint64_t at = w.dataStart + (uint64_t)i * (uint64_t)w.bytesPerFrame + (uint64_t)ch * (uint64_t)bytesPerSample;
            const unsigned char* p;
            if (at + 4 <= bytes.size()) p = bytes.data() + at;
            else {                                              // short file: missing bytes read as zero
                for (int k = 0; k < 4; ++k) pad[k] = at + (uint64_t)k < bytes.size() ? bytes[at + (uint64_t)k] : 0;
                p = pad;
            }
            plane(out, (int)ch)[i] = sanitizeAndLimit((double)decodeSample(p, w.bits, w.isFloat), i >= w.frames / 4 * 4);
        }
    return CPQ_OK;
}

double cpq_ir_estimate_max_frequency_response_gain(const double* const* ir, int32_t n_channels, int32_t n_samples)
{
    if (!ir) return 1.0;
    return maxFrequencyResponseGain(ir, n_channels, n_samples, 1.0);
}

int32_t cpq_ir_estimate_peak_latency(const double* const* ir, int32_t n_channels, int32_t n_samples)
{
    if (!ir) return 0;
    return peakLatency(ir, n_channels, n_samples);
}

int32_t cpq_ir_compute_scale_factor(const double* const* ir, int32_t n_channels, int32_t n_samples,
                                    const double* const* current_ir, int32_t current_channels, int32_t current_samples,
                                    double current_scale, cpq_ir_scale* out)
{
    if (!out) return CPQ_ERR_INVALID_ARG;
    *out = cpq_ir_scale { 1.0, 0, 0.0f, 0.0, 0.0, 1.0 };
    if (!ir || n_channels < 0 || n_samples < 0) return CPQ_ERR_INVALID_ARG;

    // stage 1: -6 dB below unit energy of the loudest channel
    double scale = 1.0;
    if (n_samples > 0 && n_channels > 0) {
        double maxEnergy = 0.0;
        for (int ch = 0; ch < n_channels; ++ch) {
            double e = 0.0;
            for (int i = 0; i < n_samples; ++i) e += ir[ch][i] * ir[ch][i];
            if (std::isfinite(e) && e > 1.0e-18) maxEnergy = std::max(maxEnergy, e);
        }
        if (maxEnergy > 1.0e-18 && std::isfinite(maxEnergy)) scale = (1.0 / std::sqrt(maxEnergy)) * 0.5011872336272722;
    }
    if (scale <= 0.0 || !std::isfinite(scale)) return CPQ_OK;
    out->scale_factor = scale;
    out->has_scale_factor = 1;

    // stage 2: peak, RMS, frequency-response peak of the unscaled IR
    const PeakRms raw = peakAndRms(ir, n_channels, n_samples, 1.0);
    out->peak_value = raw.peak;
    out->rms_value = raw.rms;
    out->frequency_peak_gain = maxFrequencyResponseGain(ir, n_channels, n_samples, 1.0);

    // stage 3: protective clamps (peak 0.5, RMS 0.25 after the peak clamp, frequency response +3 dB)
    double peakDb = 0.0, rmsDb = 0.0, freqDb = 0.0;
    if (raw.peak * scale > 0.5) {
        const double c = 0.5 / (raw.peak * scale);
        out->scale_factor *= c;
        scale *= c;
        peakDb = -20.0 * std::log10(c);
    }
    if (raw.rms * scale > 0.25) {
        const double c = 0.25 / (raw.rms * scale);
        out->scale_factor *= c;
        rmsDb = -20.0 * std::log10(c);
    }
    if (out->frequency_peak_gain > 1.41) {
        const double c = 1.41 / out->frequency_peak_gain;
        out->scale_factor *= c;
        freqDb = -20.0 * std::log10(c);
    }
    out->additional_attenuation_db = (float)(peakDb + rmsDb + freqDb);

    // jump protection against the IR that is playing now
    if (current_ir && current_channels > 0 && current_samples > 0) {
        const PeakRms cur = peakAndRms(current_ir, current_channels, current_samples, current_scale);
        const PeakRms neu = peakAndRms(ir, n_channels, n_samples, out->scale_factor);
        const bool peakJump = cur.peak > 1.0e-9 && neu.peak > cur.peak * 4.0 && neu.peak > 0.5;
        const bool rmsJump = cur.rms > 1.0e-9 && neu.rms > cur.rms * 4.0 && neu.rms > 0.25;
        if (peakJump || rmsJump) {
            double byPeak = std::numeric_limits<double>::infinity(), byRms = byPeak;
            if (neu.peak > 1.0e-12 && cur.peak > 1.0e-12) byPeak = (cur.peak * 4.0) / neu.peak;
            if (neu.rms > 1.0e-12 && cur.rms > 1.0e-12) byRms = (cur.rms * 4.0) / neu.rms;
            const double ratio = std::min(byPeak, byRms);
            if (std::isfinite(ratio) && ratio > 0.0 && ratio < 1.0) out->scale_factor *= ratio;
        }
    }
    return CPQ_OK;
}

int32_t cpq_ir_convert_to_minimum_phase(const cpq_ir_buffer* in, cpq_ir_buffer* out)
{
    if (!out) return CPQ_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof(*out));
    if (!validBuffer(in)) return CPQ_ERR_INVALID_ARG;
    if ((int64_t)in->n_samples * 4 > 8388608) return CPQ_ERR_UNSUPPORTED;      // MAX_MINPHASE_FFT_SIZE: the reference skips it
    const int fftSize = nextPow2(in->n_samples * 4);
    if (!allocBuffer(out, in->n_channels, in->n_samples, in->sample_rate)) return CPQ_ERR_OOM;
    try {
        for (int ch = 0; ch < in->n_channels; ++ch)
            if (!minimumPhase(plane(in, ch), in->n_samples, fftSize, plane(out, ch))) {
                cpq_ir_buffer_free(out);
                return CPQ_ERR_UNSUPPORTED;
            }
    } catch (const std::bad_alloc&) { cpq_ir_buffer_free(out); return CPQ_ERR_OOM; }
    return CPQ_OK;
}

int32_t cpq_ir_prepare(const cpq_ir_buffer* loaded, double sample_rate, float target_ir_length_sec, int32_t phase_mode,
                       const cpq_ir_buffer* current_ir, double current_scale, cpq_ir_prepared* out)
{
    if (!out) return CPQ_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof(*out));
    out->scale.scale_factor = 1.0;
    out->scale.frequency_peak_gain = 1.0;
    if (!validBuffer(loaded) || !(sample_rate > 0.0) || !std::isfinite(sample_rate) || !(target_ir_length_sec > 0.0f))
        return CPQ_ERR_INVALID_ARG;
    if (current_ir && !validBuffer(current_ir)) return CPQ_ERR_INVALID_ARG;
    if (phase_mode != CPQ_PHASE_AS_IS && phase_mode != CPQ_PHASE_MINIMUM)
        return phase_mode == CPQ_PHASE_MIXED ? CPQ_ERR_UNSUPPORTED : CPQ_ERR_INVALID_ARG;
    if (loaded->sample_rate > 0.0 && std::fabs(loaded->sample_rate - sample_rate) > 1e-6)
        return CPQ_ERR_UNSUPPORTED;                               // needs the r8brain resampler
    const double rate = loaded->sample_rate > 0.0 ? loaded->sample_rate : 0.0;

    // trailing silence, DC blocker, window: on a working copy of the kept part
    const int kept = trimmedLength(loaded);
    const int channels = loaded->n_channels;
    std::vector<double> work;
    try { work.resize((size_t)channels * (size_t)kept); } catch (const std::bad_alloc&) { return CPQ_ERR_OOM; }
    for (int ch = 0; ch < channels; ++ch) {
        double* w = work.data() + (size_t)ch * (size_t)kept;
        std::memcpy(w, plane(loaded, ch), sizeof(double) * (size_t)kept);
        if (rate > 0.0) dcBlock(w, kept, rate, 1.0);
        asymmetricTukey(w, kept);
    }

    // target length (seconds of the engine rate, capped at 2^21) and the fade-out of the copied part
    int target = (int)(rate * (double)target_ir_length_sec);
    target = std::max(1, std::min(target, 2097152));
    if (!allocBuffer(&out->ir, channels, target, rate)) return CPQ_ERR_OOM;
    const int copy = std::min(target, kept);
    const int maxFade = std::max(256, (int)std::round(sample_rate * 0.080));
    int fade = (int)std::round((double)copy * 0.02);
    fade = std::max(256, std::min(maxFade, fade));
    fade = std::max(0, std::min(fade, copy - 1));
    for (int ch = 0; ch < channels; ++ch) {
        double* d = plane(&out->ir, ch);
        std::memcpy(d, work.data() + (size_t)ch * (size_t)kept, sizeof(double) * (size_t)copy);
        if (fade > 0) {
            double g = 1.0;
            const double inc = (0.0 - 1.0) / (float)fade;
            for (int i = copy - fade; i < copy; ++i) { d[i] *= g; g += inc; }
        }
    }

    if (phase_mode == CPQ_PHASE_MINIMUM) {
        // doTransformStep: the converted IR replaces the trimmed one only when it validates -- finite, peak above 1e-12
        // (LoaderThread.cpp:652-680); a conversion the reference gives up on leaves the IR as it is
        cpq_ir_buffer mp;
        const int32_t rcMp = cpq_ir_convert_to_minimum_phase(&out->ir, &mp);
        if (rcMp == CPQ_ERR_OOM) { cpq_ir_buffer_free(&out->ir); return rcMp; }
        if (rcMp == CPQ_OK) {
            double peak = 0.0;
            bool finite = true;
            for (size_t i = 0; i < (size_t)channels * (size_t)target; ++i) {
                finite = finite && std::isfinite(mp.data[i]);
                peak = std::max(peak, std::fabs(mp.data[i]));
            }
            if (finite && peak > 1.0e-12) std::memcpy(out->ir.data, mp.data, sizeof(double) * (size_t)channels * (size_t)target);
            cpq_ir_buffer_free(&mp);
        }
    }

    const auto p = planes(&out->ir);
    const auto cur = current_ir ? planes(current_ir) : std::vector<const double*>();
    const int32_t rc = cpq_ir_compute_scale_factor(p.data(), channels, target, current_ir ? cur.data() : nullptr,
                                                   current_ir ? current_ir->n_channels : 0,
                                                   current_ir ? current_ir->n_samples : 0, current_scale, &out->scale);
    if (rc != CPQ_OK) { cpq_ir_buffer_free(&out->ir); return rc; }
    if (!out->scale.has_scale_factor) out->scale.scale_factor = 1.0;
    out->ir_peak_latency = peakLatency(p.data(), channels, target);
    return CPQ_OK;
}

void cpq_ir_prepared_free(cpq_ir_prepared* p)
{
    if (p) cpq_ir_buffer_free(&p->ir);
}

}  // extern "C"
