// host_design.cpp -- host-side (non-GPU) design math of libconvopeq_mi355x:
//   * the layer plan MKLNonUniformConvolver::SetImpulse derives and the closed-form effective
//     impulse response h_eff that makes the reference's observable output one linear convolution
//   * EQProcessor::calcSVFCoeffs
// Product code: never calls into oracle/.  Reference citations are relative to the reference tree.
#include "host_design.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace cpq {

namespace {

template <typename T>
T limit(T lo, T hi, T v) { return v < lo ? lo : (hi < v ? hi : v); }   // juce::jlimit

int nextPow2(int v)
{
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

constexpr int kL0MaxParts = 32;   // src/MKLNonUniformConvolver.h:392
constexpr int kL1MaxParts = 64;   // src/MKLNonUniformConvolver.h:393
constexpr int kMaxDirectTaps = 32;  // src/MKLNonUniformConvolver.cpp:689

}  // namespace

// src/MKLNonUniformConvolver.cpp:626-684 (tail profile), :689-695 (direct head), :738-758 (layers),
// :784-786 (slots), :988-994 (partsPerCallback), :1005-1024 (B13 output delay)
int computeNucPlan(int irLen, int blockSize, bool enableDirectHead, const cpq_filter_spec* spec,
                   cpq_nuc_plan* out)
{
    if (irLen <= 0 || blockSize <= 0 || out == nullptr) return CPQ_ERR_INVALID_ARG;
    cpq_nuc_plan p;
    std::memset(&p, 0, sizeof(p));

    const int tailMode = spec ? limit(0, 2, static_cast<int>(spec->tail_mode)) : 1;
    const bool tailEnabled = (tailMode != 2) && (spec ? spec->tail_enabled != 0 : true);
    const double fsTail = spec ? spec->sample_rate : 48000.0;
    double tailStart = spec ? limit(0.01, 0.80, spec->tail_start_seconds) : 0.085;
    const double userStrength = spec ? limit(0.0, 2.0, spec->tail_strength) : 1.0;
    double strength = userStrength;
    int mult = spec ? limit(2, 16, static_cast<int>(spec->tail_l1l2_multiplier)) : 8;
    const double s01 = limit(0.0, 1.0, userStrength * 0.5);
    double g1 = 1.0, g2 = 1.0;

    if (!tailEnabled || tailMode == 2) {
        g1 = g2 = 0.0;
    } else if (tailMode == 0) {
        tailStart = limit(0.01, 0.80, std::max(tailStart, 0.055));
        mult = limit(2, 16, std::max(mult, 6));
        strength = limit(0.0, 2.0, userStrength);
        g1 = limit(0.0, 2.0, strength * (0.95 - 0.25 * s01));
        g2 = limit(0.0, 2.0, strength * (0.80 - 0.45 * s01));
    } else {
        tailStart = limit(0.01, 0.80, std::max(tailStart, 0.12));
        strength = limit(0.0, 2.0, std::max(strength, 1.25));
        mult = limit(2, 16, std::max(mult, 8));
        g1 = limit(0.0, 2.0, strength * (1.05 + 0.20 * s01));
        g2 = limit(0.0, 2.0, strength * (0.82 + 0.12 * s01));
    }

    const int p0 = nextPow2(std::max(blockSize, 64));
    p.direct_taps = enableDirectHead ? std::min(irLen, std::min(p0, kMaxDirectTaps)) : 0;
    const int partOf[3] = { p0, p0 * mult, p0 * mult * mult };

    const int l0Cap = kL0MaxParts * p0;
    const int l0Wanted = limit(p0, l0Cap, static_cast<int>(std::llround(tailStart * fsTail)));
    const int l0 = std::min(irLen, tailEnabled ? l0Wanted : l0Cap);
    const int l1 = tailEnabled ? std::max(0, std::min(irLen - l0, kL1MaxParts * partOf[1])) : 0;
    const int l2 = tailEnabled ? std::max(0, irLen - l0 - l1) : 0;
    const int lenOf[3] = { l0, l1, l2 };
    const int offOf[3] = { 0, l0, l0 + l1 };
    const double gainOf[3] = { 1.0, g1, g2 };

    int n = 0, before = 0;
    p.lti_valid = (blockSize == p0) ? 1 : 0;
    int heffLen = 0;
    for (int li = 0; li < 3; ++li) {
        if (lenOf[li] <= 0) continue;
        p.part_size[n] = partOf[li];
        p.offset[n] = offOf[li];
        p.len[n] = lenOf[li];
        p.num_parts_ir[n] = (lenOf[li] + partOf[li] - 1) / partOf[li];
        p.num_parts[n] = nextPow2(p.num_parts_ir[n]);
        p.gain[n] = gainOf[li];
        p.output_delay[n] = before;
        if (li > 0) {
            const int blocksPerPart = (partOf[li] + blockSize - 1) / blockSize;
            int ppc = std::max(1, (p.num_parts_ir[n] + blocksPerPart - 1) / blocksPerPart);
            ppc = std::min(ppc, p.num_parts_ir[n]);
            p.parts_per_callback[n] = ppc;
            // first tail block: input complete at callback blocksPerPart-1, MAC spread over
            // ceil(numPartsIR/ppc) callbacks starting there, written + read in the last of them
            p.done_callback[n] = (blocksPerPart - 1) + ((p.num_parts_ir[n] + ppc - 1) / ppc - 1);
            p.lag[n] = p.done_callback[n] * blockSize - p.offset[n];
            if (p.part_size[n] > p.output_delay[n]) p.lti_valid = 0;   // reader skips blocks (:1658-1666)
        }
        heffLen = std::max(heffLen, p.offset[n] + p.lag[n] + p.len[n]);
        before += lenOf[li];
        ++n;
    }
    if (n == 0) return CPQ_ERR_INVALID_ARG;
    p.num_layers = n;
    p.latency = p0;
    p.heff_len = heffLen;
    *out = p;
    return CPQ_OK;
}

int buildHeff(const double* ir, int irLen, int blockSize, double scale, const cpq_filter_spec* spec,
              std::vector<double>& heff, cpq_nuc_plan* planOut)
{
    cpq_nuc_plan p;
    const int rc = computeNucPlan(irLen, blockSize, false, spec, &p);
    if (rc != CPQ_OK) return rc;
    heff.assign(static_cast<size_t>(p.heff_len), 0.0);
    // cblas_dscal of every partition spectrum (:939-940) == scaling the taps
    const bool scaled = std::abs(scale - 1.0) > 1e-12;
    for (int l = 0; l < p.num_layers; ++l) {
        const int at = p.offset[l] + p.lag[l];
        for (int i = 0; i < p.len[l]; ++i) {
            const int d = at + i;
            if (d < 0) continue;
            const double tap = scaled ? ir[p.offset[l] + i] * scale : ir[p.offset[l] + i];
            heff[static_cast<size_t>(d)] += p.gain[l] * tap;
        }
    }
    if (planOut) *planOut = p;
    return CPQ_OK;
}

// MKLNonUniformConvolver::applySpectrumFilter (src/MKLNonUniformConvolver.cpp:336-443): real per-bin gains of the
// HC (high cut) and LC (low cut) output filters for an N-point frame; gains[0..N/2].
void spectrumFilterGains(const cpq_filter_spec& spec, int N, std::vector<double>& gains)
{
    const double pi = 3.141592653589793238462643383279502884;
    const double fs = spec.sample_rate;
    const double nyquist = fs * 0.5;
    const int halfN = N / 2, cSize = halfN + 1;
    const double hcStart = (fs <= 48000.0) ? 18000.0 : 22000.0;
    const double hcEnd = nyquist;
    const double lcEnd = (spec.lc_mode == 1) ? 6.0 : 8.0;
    const double lcStart = (spec.lc_mode == 1) ? 15.0 : 18.0;
    gains.assign((size_t)cSize, 1.0);
    {
        const int kS = (int)std::round(hcStart * N / fs);
        const int kE = std::min(halfN, (int)std::round(hcEnd * N / fs));
        for (int k = 0; k < cSize; ++k) {
            if (k <= kS || k > kE) continue;
            const double x = (double)(k - kS) / (double)(kE - kS);
            switch (spec.hc_mode) {
                case 0: gains[k] = 1.0 / std::sqrt(1.0 + std::pow(x, 8.0)); break;
                case 1: gains[k] = 0.5 * (1.0 + std::cos(pi * x)); break;
                case 2: gains[k] = std::exp(-4.60517 * x * x); break;
                default: break;
            }
        }
    }
    {
        const int kE = (int)std::round(lcEnd * N / fs);
        const int kS = (int)std::round(lcStart * N / fs);
        for (int k = 0; k < cSize; ++k) {
            if (k <= kE) gains[k] = 0.0;
            else if (k < kS) {
                const double x = (double)(k - kE) / (double)std::max(1, kS - kE);
                gains[k] *= 0.5 * (1.0 - std::cos(pi * x));
            }
        }
    }
}

// Air-absorption HF damping of tail layer `layer` (1 or 2) in tail mode 0 (src/MKLNonUniformConvolver.cpp:1060-1097):
// gains[k] = exp(-coeff (k / (complexSize - 1))^2), the same curve on every partition of the layer.
// Returns false when the spec does not select it (tail disabled or tail mode != 0).
bool airAbsorptionGains(const cpq_filter_spec& spec, int layer, int complexSize, std::vector<double>& gains)
{
    const int tailMode = limit(0, 2, static_cast<int>(spec.tail_mode));
    const bool tailEnabled = (tailMode != 2) && spec.tail_enabled != 0;
    if (!tailEnabled || tailMode != 0 || layer < 1) return false;
    const double userStrength = limit(0.0, 2.0, spec.tail_strength);
    const double s01 = limit(0.0, 1.0, userStrength * 0.5);
    const double tailStart = limit(0.01, 0.80, std::max(limit(0.01, 0.80, spec.tail_start_seconds), 0.055));   // :648-650
    const double startNorm = limit(0.65, 1.55, tailStart / 0.085);
    const double coeff = (0.35 + 1.10 * s01) * startNorm * ((layer == 1) ? 1.0 : 1.6);
    const double denom = static_cast<double>(std::max(1, complexSize - 1));
    gains.resize(static_cast<size_t>(complexSize));
    for (int k = 0; k < complexSize; ++k) {
        const double fn = static_cast<double>(k) / denom;
        gains[static_cast<size_t>(k)] = std::exp(-coeff * fn * fn);
    }
    return true;
}

// src/eqprocessor/EQProcessor.Coefficients.cpp:84-96 (clamps, float), :101-130, :431-618
void designSvf(int type, float freq, float gainDb, float q, double sr, cpq_svf_coeffs* c)
{
    auto bypass = [c]() { c->a1 = 1.0; c->a2 = 0.0; c->a3 = 0.0; c->m0 = 1.0; c->m1 = 0.0; c->m2 = 0.0; };
    *c = cpq_svf_coeffs{ 0.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0 };
    if (sr <= 0.0) { bypass(); return; }

    const float nyquist = static_cast<float>(sr * 0.5);
    const float fMax = std::min(20000.0f, nyquist * 0.95f);
    freq = limit(20.0f, fMax, freq);
    q = limit(0.01f, 20.0f, q);
    gainDb = limit(-48.0f, 48.0f, gainDb);

    const double f = freq, gdb = gainDb, Q = q;
    const double pi = 3.141592653589793238462643383279502884;   // juce::MathConstants<double>::pi
    double A = 1.0, g = 0.0, k = 0.0;
    switch (type) {
        case 0: A = std::pow(10.0, gdb / 40.0); g = std::tan(pi * f / sr) / std::sqrt(A); k = 1.0 / Q; break;
        case 1: A = std::pow(10.0, gdb / 40.0); g = std::tan(pi * f / sr); k = 1.0 / (Q * A); break;
        case 2: A = std::pow(10.0, gdb / 40.0); g = std::tan(pi * f / sr) * std::sqrt(A); k = 1.0 / Q; break;
        case 3: case 4: g = std::tan(pi * f / sr); k = 1.0 / Q; break;
        default: return;   // `return {}`
    }
    if (!std::isfinite(g) || !std::isfinite(k)) { bypass(); return; }
    const double den = 1.0 + g * (g + k);
    if (std::abs(den) < 1.0e-15) { bypass(); return; }
    c->a1 = 1.0 / den;
    c->a2 = g * c->a1;
    c->a3 = g * c->a2;
    switch (type) {
        case 0: c->m0 = 1.0;   c->m1 = k * (A - 1.0);     c->m2 = A * A - 1.0; break;
        case 1: c->m0 = 1.0;   c->m1 = (A - 1.0 / A) / Q; c->m2 = 0.0; break;
        case 2: c->m0 = A * A; c->m1 = k * (1.0 - A) * A; c->m2 = 1.0 - A * A; break;
        case 3: c->m0 = 0.0;   c->m1 = 0.0;               c->m2 = 1.0; break;
        case 4: c->m0 = 1.0;   c->m1 = -k;                c->m2 = -1.0; break;
    }
}

// src/core/EQParameters.h:31-46
void defaultEqParams(cpq_eq_params* p)
{
    static const float f[CPQ_NUM_BANDS] = { 20.0f, 32.0f, 50.0f, 80.0f, 125.0f, 200.0f, 315.0f, 500.0f,
                                            800.0f, 1250.0f, 2000.0f, 3150.0f, 5000.0f, 8000.0f, 12500.0f,
                                            16000.0f, 19000.0f, 20000.0f, 22000.0f, 24000.0f };
    std::memset(p, 0, sizeof(*p));
    for (int i = 0; i < CPQ_NUM_BANDS; ++i)
        p->bands[i] = cpq_eq_band{ f[i], 0.0f, 0.707f, 1, 1, 0 };
    p->total_gain_db = 0.0f;
    p->agc_enabled = 0;
    p->nonlinear_saturation = 0.2f;
    p->filter_structure = 0;
}

// juce::Decibels::decibelsToGain<double> as used by storeTotalGainDb (src/eqprocessor/EQProcessor.h:447-451)
double totalGainLinear(float db)
{
    const double d = db;
    return d > -100.0 ? std::pow(10.0, d * 0.05) : 0.0;
}

// State-space form of one TPT-SVF band (src/eqprocessor/EQProcessor.Processing.cpp:228-241):
//   ic' = A ic + Bv v0,   y_lin = C ic + D v0
//   A = [[2 a1 - 1, -2 a2], [2 a2, 1 - 2 a3]],  Bv = [2 a2, 2 a3],  C = [m1 a1 + m2 a2, -m1 a2 + m2 (1 - a3)]
namespace {
typedef long double ld;
bool buildTpTablesFromStateSpace(const ld* A, const ld* C, const ld* Bv, ld D, double* out);
}  // namespace

bool buildSvfTpTables(const cpq_svf_coeffs& c, double* out)
{
    const ld a1 = c.a1, a2 = c.a2, a3 = c.a3, m1 = c.m1, m2 = c.m2;
    const ld A[4] = { 2 * a1 - 1, -2 * a2, 2 * a2, 1 - 2 * a3 };
    const ld C[2] = { m1 * a1 + m2 * a2, -m1 * a2 + m2 * (1 - a3) };
    const ld Bv[2] = { 2 * a2, 2 * a3 };
    const ld D = (ld)c.m0 + m1 * a2 + m2 * a3;          // coefficient of v0 in y = m0 v0 + m1 v1 + m2 v2
    return buildTpTablesFromStateSpace(A, C, Bv, D, out);
}

// DF-II-T biquad (src/OutputFilter.h:33-68): y = b0 x + w1; w1' = b1 x - a1 y + w2; w2' = b2 x - a2 y
//   => state (w1, w2): A = [[-a1, 1], [-a2, 0]], Bv = [b1 - a1 b0, b2 - a2 b0], C = [1, 0], D = b0
bool buildBiquadTpTables(const cpq_biquad_coeffs& q, double* out)
{
    const ld A[4] = { -(ld)q.a1, 1, -(ld)q.a2, 0 };
    const ld C[2] = { 1, 0 };
    const ld Bv[2] = { (ld)q.b1 - (ld)q.a1 * q.b0, (ld)q.b2 - (ld)q.a2 * q.b0 };
    return buildTpTablesFromStateSpace(A, C, Bv, (ld)q.b0, out);
}

namespace {
bool buildTpTablesFromStateSpace(const ld* A, const ld* C, const ld* Bv, ld D, double* out)
{
    auto mul = [](const ld* x, const ld* y, ld* z) {
        const ld r[4] = { x[0] * y[0] + x[1] * y[2], x[0] * y[1] + x[1] * y[3],
                          x[2] * y[0] + x[3] * y[2], x[2] * y[1] + x[3] * y[3] };
        for (int i = 0; i < 4; ++i) z[i] = r[i];
    };
    auto power = [&](long n, ld* r) {   // r = A^n by binary powering
        ld acc[4] = { 1, 0, 0, 1 }, base[4] = { A[0], A[1], A[2], A[3] };
        while (n > 0) {
            if (n & 1) mul(base, acc, acc);
            mul(base, base, base);
            n >>= 1;
        }
        for (int q = 0; q < 4; ++q) r[q] = acc[q];
    };
    // layout: struct TpLcTables { Mk[6][4]; Mw[4]; P[64][4]; G[16][2]; } for LC = 16 then LC = 2 (svf_kernels.hip)
    const int* lcs = kSvfTpLc;
    constexpr int kPerLc = 6 * 4 + 4 + 64 * 4 + 16 * 2;
    for (int li = 0; li < kSvfTpLcCount; ++li) {
        double* o = out + li * kPerLc;
        const int lc = lcs[li];
        ld M[4];
        for (int k = 0; k < 6; ++k) {
            power((long)lc << k, M);
            for (int q = 0; q < 4; ++q) o[k * 4 + q] = (double)M[q];
        }
        power((long)lc * 64, M);
        for (int q = 0; q < 4; ++q) o[24 + q] = (double)M[q];
        for (int c = 0; c < 64; ++c) {
            power((long)lc * (c + 1), M);
            for (int q = 0; q < 4; ++q) o[28 + c * 4 + q] = (double)M[q];
        }
        ld P[4] = { 1, 0, 0, 1 };
        for (int i = 0; i < 16; ++i) {
            o[28 + 256 + 2 * i] = (double)(C[0] * P[0] + C[1] * P[2]);
            o[28 + 256 + 2 * i + 1] = (double)(C[0] * P[1] + C[1] * P[3]);
            mul(A, P, P);
        }
    }
    // matrix form of a 16-sample chunk (MFMA path of the time-parallel kernel): zero-state impulse response
    // h[0] = D, h[n] = C A^(n-1) B behind 15 zeros, and the chunk's end-state map e[:, k] = A^(15-k) B
    {
        double* o = out + kSvfTpLcCount * kPerLc;
        for (int i = 0; i < kSvfTpMfmaDoubles; ++i) o[i] = 0.0;
        ld v[2] = { Bv[0], Bv[1] };                     // A^n B
        o[15] = (double)D;
        for (int n = 1; n < 16; ++n) {
            o[15 + n] = (double)(C[0] * v[0] + C[1] * v[1]);
            o[32 + (16 - n)] = (double)v[0];            // e[0][15 - (n-1)] = (A^(n-1) B)_0
            o[48 + (16 - n)] = (double)v[1];
            const ld t0 = A[0] * v[0] + A[1] * v[1], t1 = A[2] * v[0] + A[3] * v[1];
            v[0] = t0; v[1] = t1;
        }
        o[32 + 0] = (double)v[0];                       // A^15 B
        o[48 + 0] = (double)v[1];
    }
    for (int i = 0; i < kSvfTpTableDoubles; ++i) if (!std::isfinite(out[i])) return false;

    // guard-freedom proof: sup_n |A^n|_inf (carried state) and the l1 gain input -> state must keep every
    // state below 1e15 for |input|, |carried state| < 1e9 (one decade of margin)
    // spectral radius first: a section that does not decay (bypass coefficients: A = I) or decays too slowly for the
    // bounded iteration below is rejected without running it
    {
        const ld tr = A[0] + A[3], det = A[0] * A[3] - A[1] * A[2], disc = tr * tr - 4 * det;
        const ld rho = disc < 0 ? std::sqrt(std::fabs(det))
                                : std::max(std::fabs(tr + std::sqrt(disc)), std::fabs(tr - std::sqrt(disc))) / 2;
        if (!(rho < 1.0L - 6.0e-6L)) return false;
    }
    ld Q[4] = { 1, 0, 0, 1 };
    ld s[2] = { Bv[0], Bv[1] };
    ld kappa = 1, l1 = 0;
    const long maxIter = 1L << 23;
    long n = 0;
    for (; n < maxIter; ++n) {
        l1 += std::max(std::fabs(s[0]), std::fabs(s[1]));
        const ld t0 = A[0] * s[0] + A[1] * s[1], t1 = A[2] * s[0] + A[3] * s[1];
        s[0] = t0; s[1] = t1;
        mul(A, Q, Q);
        const ld nq = std::max(std::fabs(Q[0]) + std::fabs(Q[1]), std::fabs(Q[2]) + std::fabs(Q[3]));
        kappa = std::max(kappa, nq);
        if (!(nq < 1e30L)) return false;
        if (nq < 1e-22L && std::max(std::fabs(s[0]), std::fabs(s[1])) < 1e-22L) break;
    }
    if (n >= maxIter) return false;
    // states: |s| <= (kappa + l1) * 1e9 must stay far below the 1e15 guard;
    // band output: |y_lin| <= (|C|_1 (kappa + l1) + |D|) * 1e9 must stay below it as well (the kernel omits the
    // output guard on its fast path); |D| <= |C|_1-scale coefficients, bounded here by 1e3 (A^2 at +48 dB = 251)
    const ld c1 = std::fabs(C[0]) + std::fabs(C[1]);
    return (kappa + l1) < 1.0e5L && (c1 * (kappa + l1) + 1.0e3L) < 1.0e5L;
}
}  // namespace

// OutputFilter::makeLPF / makeHPF / makeIdentity (src/OutputFilter.cpp:23-72), RBJ cookbook forms
static cpq_biquad_coeffs biquadIdentity() { return cpq_biquad_coeffs{ 1.0, 0.0, 0.0, 0.0, 0.0 }; }

static cpq_biquad_coeffs makeLpf(double fc, double Q, double fs)
{
    const double nyq = fs * 0.4999;
    if (fc >= nyq || Q <= 0.0 || fs <= 0.0) return biquadIdentity();
    const double w0 = 2.0 * 3.141592653589793238462643383279502884 * fc / fs;
    const double sn = std::sin(w0), cs = std::cos(w0);
    const double alpha = sn / (2.0 * Q);
    const double a0inv = 1.0 / (1.0 + alpha);
    cpq_biquad_coeffs c;
    c.b0 = (1.0 - cs) * 0.5 * a0inv;
    c.b1 = (1.0 - cs) * a0inv;
    c.b2 = (1.0 - cs) * 0.5 * a0inv;
    c.a1 = (-2.0 * cs) * a0inv;
    c.a2 = (1.0 - alpha) * a0inv;
    return c;
}

static cpq_biquad_coeffs makeHpf(double fc, double Q, double fs)
{
    const double nyq = fs * 0.4999;
    if (fc <= 0.0 || fc >= nyq || Q <= 0.0 || fs <= 0.0) return biquadIdentity();
    const double w0 = 2.0 * 3.141592653589793238462643383279502884 * fc / fs;
    const double sn = std::sin(w0), cs = std::cos(w0);
    const double alpha = sn / (2.0 * Q);
    const double a0inv = 1.0 / (1.0 + alpha);
    cpq_biquad_coeffs c;
    c.b0 = (1.0 + cs) * 0.5 * a0inv;
    c.b1 = -(1.0 + cs) * a0inv;
    c.b2 = (1.0 + cs) * 0.5 * a0inv;
    c.a1 = (-2.0 * cs) * a0inv;
    c.a2 = (1.0 - alpha) * a0inv;
    return c;
}

// the three sections OutputFilter::process runs, in order (src/OutputFilter.cpp:78-121, 214-222, 318-324):
//   convIsLast:  LC (HPF 18/15 Hz) -> HC stage 0 -> HC stage 1        (fc 19 kHz, or 22 kHz above 48 kHz)
//   EQ is last:  HPF 20 Hz         -> LP stage 0 -> LP stage 1        (fc 19 kHz, or 24 kHz above 48 kHz)
void designOutputFilter(int convIsLast, int hcMode, int lcMode, int lpMode, double fs, cpq_biquad_coeffs out[3])
{
    const double fcHc = (fs <= 48000.0) ? 19000.0 : 22000.0;
    const double fcLp = (fs <= 48000.0) ? 19000.0 : 24000.0;
    if (convIsLast) {
        out[0] = (lcMode == 1) ? makeHpf(15.0, 0.5, fs) : makeHpf(18.0, 0.70711, fs);
        switch (hcMode) {
            case 0: out[1] = makeLpf(fcHc, 0.54120, fs); out[2] = makeLpf(fcHc, 1.30656, fs); break;
            case 2: out[1] = makeLpf(fcHc, 0.5, fs); out[2] = biquadIdentity(); break;
            default: out[1] = makeLpf(fcHc, 0.70711, fs); out[2] = makeLpf(fcHc, 0.70711, fs); break;
        }
    } else {
        out[0] = makeHpf(20.0, 0.70711, fs);
        const double q = (lpMode == 0) ? 1.0 : (lpMode == 2 ? 0.5 : 0.70711);
        out[1] = makeLpf(fcLp, q, fs);
        out[2] = makeLpf(fcLp, q, fs);
    }
}

}  // namespace cpq
