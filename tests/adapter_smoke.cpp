// Compiles and runs the C++ adapter on a GPU box: reference-style Add/Get and process() calls, checked against
// the long-double direct form (built by tests/test_gpu_adapter.py).
#include <cmath>
#include <cstdio>
#include <vector>

#include "convopeq_mi355x.hpp"

int main(int argc, char** argv)
{
    const int S = 2, B = 512, L = 3000, T = 4, N = 3 * T * B;
    try {
        cpq::Engine eng(S, B, L, T);
        cpq::BatchedConvolver conv(eng);
        std::vector<double> ir(L), x(static_cast<size_t>(2 * S) * N), y(x.size());
        for (int i = 0; i < L; ++i) ir[i] = std::sin(0.37 * i) * std::exp(-i / 600.0) * 0.05;
        for (size_t i = 0; i < x.size(); ++i) x[i] = std::sin(0.001 * i * (1 + i % 7)) * 0.25;
        if (!conv.SetImpulse(CPQ_ALL_STREAMS, ir.data(), ir.data(), L, B)) return 2;
        if (!conv.isReady() || conv.getLatency() != 512) return 3;
        // planar [channel][numSamples] per call
        std::vector<double> in(static_cast<size_t>(2 * S) * T * B), out(in.size());
        for (int call = 0; call < 3; ++call) {
            for (int c = 0; c < 2 * S; ++c)
                for (int i = 0; i < T * B; ++i) in[c * T * B + i] = x[static_cast<size_t>(c) * N + call * T * B + i];
            conv.Add(in.data(), T * B);
            if (conv.Get(out.data(), T * B) != T * B) return 4;
            for (int c = 0; c < 2 * S; ++c)
                for (int i = 0; i < T * B; ++i) y[static_cast<size_t>(c) * N + call * T * B + i] = out[c * T * B + i];
        }
        double worst = 0.0;
        for (int c = 0; c < 2 * S; ++c)
            for (int n = 0; n < N; n += 97) {
                long double acc = 0;
                for (int j = 0; j < L && j <= n; ++j) acc += (long double)ir[j] * x[static_cast<size_t>(c) * N + n - j];
                worst = std::fmax(worst, std::fabs((double)acc - y[static_cast<size_t>(c) * N + n]));
            }
        std::printf("adapter max abs err %.3e\n", worst);
        if (!(worst < 1e-13)) return 5;
        // a call the whole-block engine cannot take is refused out loud, not answered with silent zeros
        if (conv.Add(in.data(), 480) || conv.lastStatus() == CPQ_OK || conv.Get(out.data(), 480) != 0) return 10;
        {
            // a 480-sample device block: the adapter picks CPQ_CALLS_ANY; the first Get has nothing yet (the reference's
            // ring is empty until 512 samples have come in), the second delivers the convolution's first 480 samples
            cpq::Engine eng480(S, 480, L, 1);
            cpq::BatchedConvolver c480(eng480);
            if (!eng480.acceptsAnyCallSize() || !c480.SetImpulse(CPQ_ALL_STREAMS, ir.data(), ir.data(), L, 480)) return 11;
            if (c480.getLatency() != 512) return 12;
            std::vector<double> i2(static_cast<size_t>(2 * S) * 480), o2(i2.size());
            double w2 = 0.0;
            for (int call = 0; call < 6; ++call) {
                for (int c = 0; c < 2 * S; ++c)
                    for (int i = 0; i < 480; ++i) i2[c * 480 + i] = x[static_cast<size_t>(c) * N + call * 480 + i];
                if (!c480.Add(i2.data(), 480)) return 13;
                const int got = c480.Get(o2.data(), 480);
                if (got != (call == 0 ? 0 : 480)) return 14;
                for (int c = 0; c < 2 * S && call > 0; ++c)
                    for (int i = 0; i < 480; i += 37) {
                        const int nn = (call - 1) * 480 + i;        // one call of latency: output sample nn of the convolution
                        long double acc = 0;
                        for (int j = 0; j < L && j <= nn; ++j) acc += (long double)ir[j] * x[static_cast<size_t>(c) * N + nn - j];
                        w2 = std::fmax(w2, std::fabs((double)acc - o2[c * 480 + i]));
                    }
            }
            std::printf("adapter, 480-sample calls: max abs err %.3e\n", w2);
            if (!(w2 < 1e-13)) return 15;
        }
        if (argc > 1) {
            // processor-level adapter: IR file -> loader steps -> engine; EQ bypassed before the first block = pass-through
            // of the EQ stage, so the block is wet * wetG + delayed dry * dryG of the file's IR: finite, not silent
            cpq::Engine eng2(S, B, 48000, T);
            cpq::BatchedProcessor proc(eng2);
            proc.prepareToPlay(48000.0, T * B);
            if (!proc.loadImpulseFile(CPQ_ALL_STREAMS, argv[1], 48000.0, 1.0f, CPQ_PHASE_MINIMUM, nullptr, 0.7f)) return 6;
            cpq_eq_params ep;
            cpq_eq_params_default(&ep);
            if (!proc.setEqParameters(CPQ_ALL_STREAMS, ep)) return 7;
            cpq_engine_set_conv_level(eng2.get(), 1);
            proc.setBypassFromRT(CPQ_ALL_STREAMS, true);
            proc.requestBandReset(0, 0xFFFFFFFFu);
            proc.setGains(1, 1.0, 0.5);
            std::vector<std::vector<double>> planes(2 * S, std::vector<double>(T * B));
            std::vector<double*> ptrs;
            for (int c = 0; c < 2 * S; ++c) {
                for (int i = 0; i < T * B; ++i) planes[c][i] = x[static_cast<size_t>(c) * N + i];
                ptrs.push_back(planes[c].data());
            }
            cpq::AudioBlockBatch blk{ ptrs.data(), 2 * S, T * B };
            proc.process(blk);
            double energy = 0.0;
            for (int c = 0; c < 2 * S; ++c)
                for (int i = 0; i < T * B; ++i) { if (!std::isfinite(planes[c][i])) return 8; energy += planes[c][i] * planes[c][i]; }
            std::printf("processor adapter block energy %.6e\n", energy);
            if (!(energy > 1e-6)) return 9;
        }
        return 0;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 1;
    }
}
