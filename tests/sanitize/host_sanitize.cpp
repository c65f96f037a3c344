// host_sanitize.cpp -- drives the product's host-only design code (convopeq_amd/csrc/host_design.cpp: layer plan, h_eff,
// spectral gains, SVF / biquad design and the time-parallel tables) and the oracle's C restatement over a sweep of
// arguments under AddressSanitizer + UndefinedBehaviorSanitizer.  CPU build only (GPU ASan is not available on the
// pool); compiled and run by tests/test_host_sanitizers_cpu.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "host_design.hpp"
extern "C" {
#include "cpq_oracle.h"
}

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)

int main()
{
    const int irLens[] = { 1, 31, 512, 4096, 5760, 20000, 131072, 524288 };
    const int blocks[] = { 64, 128, 512, 1024, 4096 };
    for (int irLen : irLens)
        for (int b : blocks)
            for (int mode = -1; mode < 3; ++mode) {
                cpq_filter_spec sp{ 48000.0, 1, 0, mode < 0 ? 1 : mode, 1, 0.085, 1.0, 8, 0 };
                orc_filter_spec so{};
                so.sampleRate = 48000.0; so.hcMode = 1; so.lcMode = 0; so.tailMode = sp.tail_mode; so.tailEnabled = 1;
                so.tailStartSeconds = 0.085; so.tailStrength = 1.0; so.tailL1L2Multiplier = 8; so.applySpectrumFilter = 0;
                cpq_nuc_plan p;
                orc_nuc_plan q;
                CHECK(cpq::computeNucPlan(irLen, b, (irLen & 1) != 0, mode < 0 ? nullptr : &sp, &p) == CPQ_OK);
                CHECK(orc_nuc_plan_compute(irLen, b, (irLen & 1) != 0, mode < 0 ? nullptr : &so, &q) == 0);
                CHECK(p.num_layers == q.numLayers && p.lti_valid == q.ltiValid && p.direct_taps == q.directTaps);
                for (int l = 0; l < p.num_layers; ++l)
                    CHECK(p.part_size[l] == q.partSize[l] && p.offset[l] == q.offset[l] && p.len[l] == q.len[l] &&
                          p.num_parts_ir[l] == q.numPartsIR[l] && p.parts_per_callback[l] == q.partsPerCallback[l] &&
                          p.output_delay[l] == q.outputDelay[l] && p.gain[l] == q.gain[l]);
                if (irLen <= 131072 && p.lti_valid) {
                    std::vector<double> ir((size_t)irLen), h;
                    orc_gen_ir(ir.data(), irLen, 0x1257, 0, 0);
                    CHECK(cpq::buildHeff(ir.data(), irLen, b, 0.5, mode < 0 ? nullptr : &sp, h, nullptr) == CPQ_OK);
                    CHECK((int)h.size() == p.heff_len || p.heff_len == 0);
                }
                if (mode >= 0) {
                    std::vector<double> g;
                    cpq::spectrumFilterGains(sp, 2 * b, g);
                    CHECK((int)g.size() == b + 1);
                    for (double v : g) CHECK(std::isfinite(v) && v >= 0.0 && v <= 1.0 + 1e-12);
                    const bool air = cpq::airAbsorptionGains(sp, 1, b + 1, g);
                    CHECK(air == (mode == 0));
                }
            }
    // SVF design + time-parallel tables over the parameter box (and beyond it: the clamps must hold)
    const float freqs[] = { -5.0f, 997.0f, 1e9f };
    const float gains[] = { -100.0f, 0.005f, 48.0f };
    const float qs[] = { 0.0f, 0.707f, 1e6f };
    std::vector<double> tab((size_t)cpq::kSvfTpTableDoubles);
    for (int type = -1; type <= 5; ++type)
        for (float f : freqs) for (float g : gains) for (float q : qs)
            for (double sr : { 0.0, 44100.0 }) {
                cpq_svf_coeffs c;
                orc_svf_coeffs o;
                cpq::designSvf(type, f, g, q, sr, &c);
                orc_svf_design(type, f, g, q, sr, &o);
                CHECK(c.a1 == o.a1 && c.a2 == o.a2 && c.a3 == o.a3 && c.m0 == o.m0 && c.m1 == o.m1 && c.m2 == o.m2);
                (void)cpq::buildSvfTpTables(c, tab.data());
            }
    for (int conv = 0; conv < 2; ++conv) for (int hc = 0; hc < 3; ++hc) for (int lc = 0; lc < 2; ++lc) for (int lp = 0; lp < 3; ++lp) {
        cpq_biquad_coeffs q[3];
        cpq::designOutputFilter(conv, hc, lc, lp, 48000.0, q);
        for (const auto& s : q) { CHECK(std::isfinite(s.b0) && std::isfinite(s.a2)); (void)cpq::buildBiquadTpTables(s, tab.data()); }
    }
    cpq_eq_params d;
    cpq::defaultEqParams(&d);
    CHECK(d.bands[19].frequency == 24000.0f && cpq::totalGainLinear(-120.0f) == 0.0);
    std::printf("host sanitize sweep: %d failed checks\n", fails);
    return fails ? 1 : 0;
}
