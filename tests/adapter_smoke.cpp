// Compiles and runs the C++ adapter on a GPU box: reference-style Add/Get and process() calls, checked against
// the long-double direct form (built by tests/test_gpu_adapter.py).
#include <cmath>
#include <cstdio>
#include <vector>

#include "convopeq_mi355x.hpp"

int main()
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
        return worst < 1e-13 ? 0 : 5;
    } catch (const std::exception& e) {
        std::printf("exception: %s\n", e.what());
        return 1;
    }
}
