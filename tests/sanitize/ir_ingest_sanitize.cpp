// ASan / UBSan driver for the IR ingest host code (convopeq_amd/csrc/ir_ingest.cpp): a valid WAV file, then the same
// file truncated at every length and with every header byte corrupted in turn, plus seeded random garbage -- the reader
// must reject or decode each without reading out of bounds -- and the conditioning / analysis functions on degenerate
// buffers.  Built by tests/test_host_sanitizers_cpu.py.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "convopeq_mi355x.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

static void put32(std::vector<unsigned char>& v, uint32_t x) { for (int i = 0; i < 4; ++i) v.push_back((unsigned char)(x >> (8 * i))); }
static void put16(std::vector<unsigned char>& v, uint32_t x) { for (int i = 0; i < 2; ++i) v.push_back((unsigned char)(x >> (8 * i))); }
static void tag(std::vector<unsigned char>& v, const char* t) { v.insert(v.end(), t, t + 4); }

static std::vector<unsigned char> makeWav(int channels, int bits, bool isFloat, bool extensible, int frames)
{
    std::vector<unsigned char> f;
    const int bpf = channels * bits / 8;
    tag(f, "RIFF"); put32(f, 0); tag(f, "WAVE");
    tag(f, "fmt "); put32(f, extensible ? 40 : 16);
    put16(f, extensible ? 0xfffe : (isFloat ? 3 : 1)); put16(f, (uint32_t)channels); put32(f, 48000); put32(f, 48000u * (uint32_t)bpf);
    put16(f, (uint32_t)bpf); put16(f, (uint32_t)bits);
    if (extensible) {
        put16(f, 22); put16(f, (uint32_t)bits); put32(f, 3);
        put32(f, isFloat ? 3 : 1); put16(f, 0); put16(f, 0x10);
        const unsigned char tail[8] = { 0x80, 0, 0, 0xaa, 0, 0x38, 0x9b, 0x71 };
        f.insert(f.end(), tail, tail + 8);
    }
    tag(f, "data"); put32(f, (uint32_t)(frames * bpf));
    uint32_t s = 12345;
    for (int i = 0; i < frames * bpf; ++i) { s = s * 1664525u + 1013904223u; f.push_back((unsigned char)(s >> 24)); }
    const uint32_t len = (uint32_t)f.size() - 8;
    for (int i = 0; i < 4; ++i) f[4 + (size_t)i] = (unsigned char)(len >> (8 * i));
    return f;
}

static int loadBytes(const std::string& path, const std::vector<unsigned char>& b, cpq_ir_buffer* out)
{
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return -100;
    if (!b.empty()) std::fwrite(b.data(), 1, b.size(), f);
    std::fclose(f);
    return cpq_ir_load_wav(path.c_str(), out);
}

int main(int argc, char** argv)
{
    const std::string path = std::string(argc > 1 ? argv[1] : "/tmp") + "/fuzz.wav";
    int decoded = 0, rejected = 0;
    const int formats[][4] = { { 2, 16, 0, 0 }, { 1, 24, 0, 1 }, { 2, 32, 1, 0 }, { 3, 8, 0, 0 }, { 2, 32, 1, 1 } };
    for (const auto& fm : formats) {
        const auto good = makeWav(fm[0], fm[1], fm[2] != 0, fm[3] != 0, 37);
        cpq_ir_buffer b;
        CHECK(loadBytes(path, good, &b) == CPQ_OK && b.n_channels == fm[0] && b.n_samples == 37 && b.sample_rate == 48000.0);
        for (int c = 0; c < b.n_channels * b.n_samples; ++c) CHECK(std::fabs(b.data[c]) <= 1.0);
        cpq_ir_buffer_free(&b);
        cpq_ir_buffer_free(&b);                                  // idempotent
        for (size_t cut = 0; cut < good.size(); ++cut) {         // every truncation
            std::vector<unsigned char> t(good.begin(), good.begin() + (long)cut);
            const int rc = loadBytes(path, t, &b);
            if (rc == CPQ_OK) { ++decoded; CHECK(b.data != nullptr && b.n_samples > 0); cpq_ir_buffer_free(&b); }
            else { ++rejected; CHECK(b.data == nullptr); }
        }
        const size_t header = good.size() - 37u * (size_t)(fm[0] * fm[1] / 8);
        for (size_t at = 0; at < header; ++at)                   // every header byte, three corruptions
            for (unsigned char v : { (unsigned char)0x00, (unsigned char)0xff, (unsigned char)(good[at] ^ 0x40) }) {
                auto t = good;
                t[at] = v;
                const int rc = loadBytes(path, t, &b);
                if (rc == CPQ_OK) { ++decoded; cpq_ir_buffer_free(&b); } else ++rejected;
            }
    }
    uint32_t s = 99;
    for (int trial = 0; trial < 300; ++trial) {                  // random garbage behind a plausible start
        std::vector<unsigned char> t;
        tag(t, trial % 3 ? "RIFF" : "RF64"); put32(t, trial % 5 ? 0xffffffffu : 64u); tag(t, "WAVE");
        if (trial % 3 == 0) { tag(t, "ds64"); put32(t, 28); }
        const int n = 16 + trial;
        for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; t.push_back((unsigned char)(s >> 24)); }
        cpq_ir_buffer b;
        if (loadBytes(path, t, &b) == CPQ_OK) { ++decoded; cpq_ir_buffer_free(&b); } else ++rejected;
    }
    std::remove(path.c_str());
    CHECK(decoded > 0 && rejected > 0);
    cpq_ir_buffer nb;
    CHECK(cpq_ir_load_wav(nullptr, &nb) == CPQ_ERR_INVALID_ARG && cpq_ir_load_wav(path.c_str(), nullptr) == CPQ_ERR_INVALID_ARG);
    CHECK(cpq_ir_load_wav(path.c_str(), &nb) == CPQ_ERR_INVALID_ARG);

    // conditioning and analysis on degenerate and ordinary buffers
    const int lens[] = { 1, 2, 3, 5, 257, 1000, 70000 };
    for (int n : lens)
        for (int ch = 1; ch <= 3; ++ch)
            for (int kind = 0; kind < 4; ++kind) {
                std::vector<double> data((size_t)n * (size_t)ch);
                for (size_t i = 0; i < data.size(); ++i) {
                    s = s * 1664525u + 1013904223u;
                    const double r = (double)(s >> 8) / 8388608.0 - 1.0;
                    data[i] = kind == 0 ? 0.0 : kind == 1 ? r * std::exp(-(double)(i % (size_t)n) / (0.2 * n + 1)) : kind == 2 ? 1.0 : (i % (size_t)n == (size_t)n - 1 ? -1.0 : 0.0);
                }
                cpq_ir_buffer in { ch, n, 48000.0, data.data() };
                for (float secs : { 0.001f, 0.2f }) {
                    cpq_ir_prepared p;
                    CHECK(cpq_ir_prepare(&in, 48000.0, secs, (n % 2) ? CPQ_PHASE_MINIMUM : CPQ_PHASE_AS_IS, nullptr, 1.0, &p) == CPQ_OK);
                    CHECK(p.ir.n_channels == ch && p.ir.n_samples == (int)(48000.0 * (double)secs));
                    CHECK(std::isfinite(p.scale.scale_factor) && p.scale.scale_factor > 0.0);
                    CHECK(p.ir_peak_latency >= 0 && p.ir_peak_latency < p.ir.n_samples);
                    for (int i = 0; i < p.ir.n_channels * p.ir.n_samples; ++i) CHECK(std::isfinite(p.ir.data[i]));
                    cpq_ir_prepared q;
                    CHECK(cpq_ir_prepare(&in, 48000.0, secs, CPQ_PHASE_AS_IS, &p.ir, 0.5, &q) == CPQ_OK);     // against itself as the current IR
                    cpq_ir_prepared_free(&q);
                    cpq_ir_prepared_free(&p);
                }
                std::vector<const double*> planes;
                for (int c = 0; c < ch; ++c) planes.push_back(data.data() + (size_t)c * (size_t)n);
                const double g = cpq_ir_estimate_max_frequency_response_gain(planes.data(), ch, n);
                CHECK(std::isfinite(g) && g > 0.0);
                const int lat = cpq_ir_estimate_peak_latency(planes.data(), ch, n);
                CHECK(lat >= 0 && lat < n);
                cpq_ir_scale sc;
                CHECK(cpq_ir_compute_scale_factor(planes.data(), ch, n, planes.data(), ch, n, 1e-3, &sc) == CPQ_OK);
                CHECK(std::isfinite(sc.scale_factor) && sc.scale_factor > 0.0);
            }
    cpq_ir_prepared p;
    double one = 1.0;
    cpq_ir_buffer bad { 1, 1, 44100.0, &one };
    CHECK(cpq_ir_prepare(&bad, 48000.0, 1.0f, CPQ_PHASE_AS_IS, nullptr, 1.0, &p) == CPQ_ERR_UNSUPPORTED && p.ir.data == nullptr);
    CHECK(cpq_ir_prepare(nullptr, 48000.0, 1.0f, CPQ_PHASE_AS_IS, nullptr, 1.0, &p) == CPQ_ERR_INVALID_ARG);
    CHECK(cpq_ir_prepare(&bad, 0.0, 1.0f, CPQ_PHASE_AS_IS, nullptr, 1.0, &p) == CPQ_ERR_INVALID_ARG);
    CHECK(cpq_ir_prepare(&bad, 44100.0, -1.0f, CPQ_PHASE_AS_IS, nullptr, 1.0, &p) == CPQ_ERR_INVALID_ARG);
    CHECK(cpq_ir_prepare(&bad, 44100.0, 1.0f, CPQ_PHASE_MIXED, nullptr, 1.0, &p) == CPQ_ERR_UNSUPPORTED);
    CHECK(cpq_ir_prepare(&bad, 44100.0, 1.0f, 7, nullptr, 1.0, &p) == CPQ_ERR_INVALID_ARG);
    cpq_ir_scale sc;
    CHECK(cpq_ir_compute_scale_factor(nullptr, 1, 1, nullptr, 0, 0, 1.0, &sc) == CPQ_ERR_INVALID_ARG);
    std::printf("ir ingest: %d decoded, %d rejected, %d failed checks\n", decoded, rejected, fails);
    return fails ? 1 : 0;
}
