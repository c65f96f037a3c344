// svf_kernels.hip -- 20-band TPT state-variable-filter cascade for gfx950.
//
// Replaces processBandStereo / processBand (src/eqprocessor/EQProcessor.Processing.cpp:191-276, :128-186)
// as driven by EQProcessor::process(block, params, cache) in its serial structure (:1231-1253) followed by
// the steady total gain (:1262-1274).
//
// The recurrence is serial in time per (channel, band) and, because every band output passes through the
// fastTanh saturation blend, serial across bands too.  The only parallelism is channel x band with the bands
// skewed in time: lane = (channel, band); at step s band b processes sample s-b and hands its output to
// band b+1 of the same channel through a one-lane wave shift.  One wave carries 3 channels x 20 bands, or the
// 2 channels of one stream when a band works on the Mid or Side component (the two lanes of such a band swap
// their inputs, both run the same mono recurrence on the encoded component and each decodes its own channel).
// Samples enter and leave through LDS in 64-sample coalesced chunks.
//
// Arithmetic follows the reference operation for operation (same FMA sites, IEEE division, same guards),
// so with identical coefficients the output is expected to be bit-identical to the SSE2+FMA path.
// This file is compiled with -ffp-contract=off: fused operations appear only where written as fma().
#include "kernels.hpp"


namespace cpq {

namespace {


// sanitizeFiniteInRangeV(v, 0, 1e15): non-finite or |v| >= 1e15 -> 0  (Processing.cpp:90-101)
__device__ __forceinline__ double sanitize(double v)
{
    // |v| < 1e15 is false for NaN and for +-Inf, so the reference's separate finiteness test is implied
    return (fabs(v) < 1.0e15) ? v : 0.0;
}


// in and out may alias (in-place processing like the reference): no __restrict__ on them.
// kChPerWave = 3 packs channels densely; kChPerWave = 2 keeps the L/R pair of a stream in one wave (Mid/Side bands).
template <int kChPerWave>
__global__ __launch_bounds__(64) void k_svf_cascade(const double* in, double* out,
                                                    int64_t chStride, int nCh, int nSamples,
                                                    const double* __restrict__ coef, const int* __restrict__ flags,
                                                    const double* __restrict__ satGain, double* __restrict__ state)
{
    __shared__ double xin[kChPerWave][64];
    __shared__ double yout[kChPerWave][128];

    const int lane = threadIdx.x;
    const int chl = lane / kBands;                 // 0..3 (3 = idle lanes 60..63)
    const int band = lane - chl * kBands;
    const int c0 = blockIdx.x * kChPerWave;
    const int c = c0 + chl;
    const bool live = (chl < kChPerWave) && (c < nCh);

    double a1 = 0, a2 = 0, a3 = 0, m0 = 1, m1 = 0, m2 = 0, ic1 = 0, ic2 = 0, sat = 0, gain = 1;
    int flag = 0;
    if (live) {
        const double* cf = coef + ((int64_t)c * kBands + band) * 6;
        a1 = cf[0]; a2 = cf[1]; a3 = cf[2]; m0 = cf[3]; m1 = cf[4]; m2 = cf[5];
        flag = flags[c * kBands + band];
        ic1 = state[((int64_t)c * kBands + band) * 2];
        ic2 = state[((int64_t)c * kBands + band) * 2 + 1];
        sat = satGain[c * 2];
        gain = satGain[c * 2 + 1];
    }
    const bool active = (flag & 1) != 0;
    const bool mono = (flag & 2) != 0;      // Left/Right channel mode -> scalar processBand arithmetic
    const bool df2t = (flag & 4) != 0;      // OutputFilter section: Direct-Form-II-transposed biquad (coef = b0 b1 b2 a1 a2)
    // FilterStructure::Parallel (Processing.cpp:1164-1226): every band filters the block INPUT; out = src + accum with
    // accum = (((0 + y_0) - src) + y_1) - src ... in band order.  Bit 3 is set on all 20 band slots of the channel.
    const bool parallel = (flag & 8) != 0;
    // Mid (1) / Side (2) band of the basic process(block) path (Processing.cpp:690-739, :792-836); bit 1 is set too
    // (processBand arithmetic).  The component state is the L slot's; both lanes carry it.
    const int msMode = (kChPerWave == 2) ? ((flag >> 4) & 3) : 0;
    const int partner = (lane < kBands) ? lane + kBands : lane - kBands;
    if (kChPerWave == 2) {
        const double l1 = __shfl(ic1, band), l2 = __shfl(ic2, band);
        if (msMode) { ic1 = l1; ic2 = l2; }
    }
    const double oneMinusSat = 1.0 - sat;

    double ylast = 0.0, xlast = 0.0;
    const int nChunks = (nSamples + 63) / 64;      // any nSamples (CPQ_CALLS_ANY: 480-sample callbacks, ragged calls): the last chunk may be short

    for (int chunk = 0; chunk <= nChunks; ++chunk) {
        // stage the next 64 input samples of the wave's channels (last iteration only drains the skew)
        if (chunk < nChunks) {
#pragma unroll
            for (int q = 0; q < kChPerWave; ++q)
                if (c0 + q < nCh)
                    xin[q][lane] = (chunk * 64 + lane < nSamples) ? in[(int64_t)(c0 + q) * chStride + (int64_t)chunk * 64 + lane] : 0.0;
        }
        __syncthreads();
        const int steps = (chunk < nChunks) ? 64 : (kBands - 1);
        for (int i = 0; i < steps; ++i) {
            const int n = chunk * 64 + i - band;                  // sample this lane handles at this step
            const double fromPrev = __shfl_up(ylast, 1);
            const double xPrev = __shfl_up(xlast, 1);
            const double xRaw = xin[chl < kChPerWave ? chl : 0][i & 63];
            // serial: v0 = previous band's output.  parallel: v0 = the raw input sample, ylast carries the accumulator
            const double xv = (band == 0) ? xRaw : xPrev;
            const double accIn = (band == 0) ? 0.0 : fromPrev;
            const double vOwn = parallel ? xv : ((band == 0) ? xRaw : fromPrev);
            const double vOther = (kChPerWave == 2) ? __shfl(vOwn, partner) : 0.0;
            if (live && n >= 0 && n < nSamples) {
                double v0 = vOwn, mid = 0.0, side = 0.0;
                if (active && msMode) {
                    // M = (L + R) * 0.5, S = (L - R) * 0.5  (copy / add|subtract / multiply, :699-704)
                    const double l = (chl == 0) ? vOwn : vOther, r = (chl == 0) ? vOther : vOwn;
                    mid = (l + r) * 0.5;
                    side = (l - r) * 0.5;
                    v0 = (msMode == 1) ? mid : side;
                }
                double y = v0;
                if (active && df2t) {
                    // biquadStep128_FMA (src/OutputFilter.cpp:143-165): state (w1, w2) in (ic1, ic2);
                    // explicit flush of |w| < 1e-20 like the reference
                    const double yy = fma(a1, v0, ic1);                       // b0 x + w1
                    double n1 = fma(a2, v0, fma(-m0, yy, ic2));               // b1 x - a1 y + w2
                    double n2 = fma(-m1, yy, a3 * v0);                        // b2 x - a2 y
                    ic1 = (fabs(n1) < 1.0e-20) ? 0.0 : n1;
                    ic2 = (fabs(n2) < 1.0e-20) ? 0.0 : n2;
                    y = yy;
                } else if (active) {
                    if (!mono) {
                        const double v3 = v0 - ic2;
                        const double v1 = fma(a1, ic1, a2 * v3);
                        const double v2 = fma(a2, ic1, fma(a3, v3, ic2));
                        ic1 = fma(2.0, v1, -ic1);
                        ic2 = fma(2.0, v2, -ic2);
                        y = fma(m0, v0, fma(m1, v1, m2 * v2));
                        if (sat > 0.0) {
                            // fastTanhV128: clamp the argument, then 27/9 Pade (FastTanhApprox.h:49-55,112-119)
                            const double xc = fmin(fmax(y, -4.5), 4.5);
                            const double x2 = xc * xc;
                            const double th = (xc * (27.0 + x2)) / (27.0 + 9.0 * x2);
                            y = (y * oneMinusSat) + (th * sat);
                        }
                        y = sanitize(y);
                        ic1 = sanitize(ic1);
                        ic2 = sanitize(ic2);
                        y = fmin(fmax(y, -100.0), 100.0);
                    } else {
                        const double v3 = v0 - ic2;
                        const double v1 = a1 * ic1 + a2 * v3;
                        const double v2 = ic2 + a2 * ic1 + a3 * v3;
                        ic1 = 2.0 * v1 - ic1;
                        ic2 = 2.0 * v2 - ic2;
                        y = m0 * v0 + m1 * v1 + m2 * v2;
                        if (sat > 0.0) {
                            // scalar fastTanh: hard +-1 beyond the clip threshold (FastTanhApprox.h:101-107)
                            double th;
                            if (y >= 4.5) th = 1.0;
                            else if (y <= -4.5) th = -1.0;
                            else { const double x2 = y * y; th = y * (27.0 + x2) / (27.0 + 9.0 * x2); }
                            y = y * oneMinusSat + th * sat;
                        }
                        y = sanitize(y);
                        y = y < -100.0 ? -100.0 : (y > 100.0 ? 100.0 : y);
                        ic1 = sanitize(ic1);
                        ic2 = sanitize(ic2);
                    }
                }
                if (active && msMode) {
                    // decode: L = M + S, R = M - S with the filtered component replaced (:711-714)
                    const double mo = (msMode == 1) ? y : mid, so = (msMode == 1) ? side : y;
                    y = (chl == 0) ? (mo + so) : (mo - so);
                }
                if (parallel) {
                    // accum += work; accum -= src  (juce::FloatVectorOperations::add / subtract, :1195-1198);
                    // Mid/Side bands: accum += work - src (:831-835)
                    const double acc = active ? (msMode ? (accIn + (y - xv)) : ((accIn + y) - xv)) : accIn;
                    ylast = acc;
                    xlast = xv;
                    if (band == kBands - 1) yout[chl][n & 127] = (xv + acc) * gain;     // block = src + accum (:1220-1221)
                } else {
                    ylast = y;
                    if (band == kBands - 1) yout[chl][n & 127] = y * gain;
                }
            }
        }
        __syncthreads();
        // block chunk-1 is complete once this chunk's steps ran (band 19 lags 19 steps)
        if (chunk >= 1) {
#pragma unroll
            for (int q = 0; q < kChPerWave; ++q)
                if (c0 + q < nCh && (chunk - 1) * 64 + lane < nSamples)
                    out[(int64_t)(c0 + q) * chStride + (int64_t)(chunk - 1) * 64 + lane] =
                        yout[q][((chunk - 1) * 64 + lane) & 127];
        }
        __syncthreads();
    }
    if (live) {
        state[((int64_t)c * kBands + band) * 2] = ic1;
        state[((int64_t)c * kBands + band) * 2 + 1] = ic2;
    }
}


// ---------------------------------------------------------------------------------------------------------
// Time-parallel kernels.
//
// Inside one band the state update is LINEAR in (v0, ic1eq, ic2eq): the fastTanh blend, the +-100 clamp and
// the output guard act on the band OUTPUT only and never feed back into the state (Processing.cpp:228-262).
// So one band over a span of chunks of LC samples (one chunk per lane, held in registers) is run as
//   1. the chunks' zero-state end states e_c = E x_c (two FMAs per sample, accumulated by the pass of the band before),
//   2. chunk start states by a scan of S_c = M S_(c-1) + e_c, M = A^LC: DPP shift-and-combine steps inside each wave,
//      then the wave totals are chained (2x2 products) and folded in with per-lane powers A^(LC (c+1));
//      all matrix powers are precomputed on the host in extended precision,
//   3. ONE pass of the reference recurrence over the chunk from its true start state, then saturation blend / guard / clamp.
// Bands remain sequential (the nonlinearity sits between them); HBM sees one read and one write per sample.
// k_svf_cascade_tpv: chunks of 16, eight waves per span of 8192 samples (or 1 ... 7 waves for what a call leaves);
// k_svf_cascade_short: chunks of 8, one or two waves, spans below 1024 samples.
//
// The state guards of the reference (non-finite or >= 1e15 -> 0) cannot trip when the span input is finite and
// below kTpInputBound and the incoming state is below it too (the host proves state gain * bound < 1e15 per
// band before enabling these kernels); otherwise the span is run by the guarded sequential path.
// Result differs from the sequential recurrence by rounding only (measured <= 3e-15 abs over 20 bands).

constexpr double kTpInputBound = 1.0e9;
constexpr int kTpChunks = 256;               // (default thread count of tp_scan)
constexpr int kTpLcMain = kSvfTpLc[0];       // samples per chunk of the span kernels
constexpr int kTpStride = kTpLcMain + 2;     // LDS row stride in doubles: rows 16-byte aligned for b128 access, 36 dwords
                                             // apart so that 16 consecutive rows cover all 64 banks

// per (stream, band); one block per chunk length (host_design.hpp: kSvfTpLc = {16, 8}); must match host buildSvfTpTables()
struct TpLcTables {
    double Mk[6][4];     // A^(LC*2^k), row-major 2x2: in-wave scan steps
    double Mw[4];        // A^(LC*64): one whole wave of chunks
    double P[64][4];     // A^(LC*(c+1)): carries the wave's start state to the end of chunk c
    double G[16][2];     // C*A^i, i < LC (state-to-output response; not read by the kernels of this file any more)
};
// e[:, k] = A^(15-k) B: end state of a 16-sample chunk's zero-state run (its last LC columns serve a chunk of LC);
// ht: the chunk's zero-state impulse response in matrix form (round 2's MFMA path, tools/variants/; not read here)
struct TpMfmaTables {
    double ht[32];
    double e[2][16];
};
struct TpBandTables { TpLcTables t[kSvfTpLcCount]; TpMfmaTables mm; };

// num / den for the fastTanh Pade: den in [27, 209.25], |num| <= 212.7, so the range scaling and special-case
// fix-up of the generic fp64 division (v_div_scale / v_div_fmas / v_div_fixup, which serialise on VCC) are
// no-ops and are omitted; what remains is the same Newton + residual sequence, so the quotient equals the IEEE
// result for every normal-range quotient and independent divisions can be interleaved.
// ONE_STEP (fast path of the time-parallel kernel): a single Newton step.  v_rcp_f64 is good to 4.6e-8 here, one step
// leaves r within 2.2e-15 and the residual correction absorbs that: 0 mismatches against the IEEE quotient in 2.1e9
// operand pairs of this range (tools/ubench/pade_div_check.hip); a miss would be a 1-ulp difference in fastTanh.
template <bool ONE_STEP = false>
__device__ __forceinline__ double pade_div(double num, double den)
{
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    if (!ONE_STEP) r = fma(fma(-den, r, 1.0), r, r);
    const double q = num * r;
    return fma(fma(-den, q, num), r, q);
}

// Output stage of one band for N independent samples (blend with fastTanh, output guard, clamp), written
// stage by stage so that independent operations are adjacent in program order.
template <bool MONO, bool SAT, int N, bool GUARD = true>
__device__ __forceinline__ void tp_nonlinear(double (&y)[N], double sat, double oneMinusSat)
{
    if (SAT) {
        double xc[N], num[N], den[N];
#pragma unroll
        for (int j = 0; j < N; ++j) xc[j] = MONO ? y[j] : fmin(fmax(y[j], -4.5), 4.5);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double x2 = xc[j] * xc[j];
            num[j] = xc[j] * (27.0 + x2);
            // fast path: one rounding less in the denominator and in the blend (rounding-level, like the rest of the
            // time-parallel evaluation); the guarded path keeps the reference's operation order
            den[j] = GUARD ? (27.0 + 9.0 * x2) : fma(9.0, x2, 27.0);
        }
#pragma unroll
        for (int j = 0; j < N; ++j) num[j] = pade_div<!GUARD>(num[j], den[j]);
        if (MONO) {
            // scalar fastTanh: +-1 beyond the clip threshold (FastTanhApprox.h:101-107)
#pragma unroll
            for (int j = 0; j < N; ++j) { num[j] = (y[j] >= 4.5) ? 1.0 : num[j]; num[j] = (y[j] <= -4.5) ? -1.0 : num[j]; }
        }
#pragma unroll
        for (int j = 0; j < N; ++j) y[j] = GUARD ? ((y[j] * oneMinusSat) + (num[j] * sat)) : fma(num[j], sat, y[j] * oneMinusSat);
    }
    // output guard (non-finite or |y| >= 1e15 -> 0): the host only enables the time-parallel kernel when it has
    // proven |y| stays below 1e15 for every span this path accepts (inputs and carried states below kTpInputBound),
    // so on the fast path (GUARD = false) the guard is the identity and is omitted
    if (GUARD) {
#pragma unroll
        for (int j = 0; j < N; ++j) y[j] = sanitize(y[j]);
    }
    if (MONO) {
#pragma unroll
        for (int j = 0; j < N; ++j) { y[j] = (y[j] < -100.0) ? -100.0 : y[j]; y[j] = (y[j] > 100.0) ? 100.0 : y[j]; }
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) y[j] = fmin(fmax(y[j], -100.0), 100.0);
    }
}

// Output stage when every |y| of the group is below the fastTanh clip threshold (4.5): neither the argument clamp, nor
// the scalar path's hard +-1, nor the +-100 clamp can act (|out| <= |y| for 0 <= sat <= 1), and the blend folds into
// one rational function:  y (1 - s) + s y (27 + y^2) / (27 + 9 y^2)  =  y (27 + c1 y^2) / (27 + 9 y^2),  c1 = 9 - 8 s,
// which in partial fractions is  y (c1 / 9 + (3 - c1 / 3) / (3 + y^2)):  one reciprocal of den in [3, 23.25] and a
// multiply-add instead of a full division.  The reciprocal is refined with one third-order step (r (1 + e + e^2), e = 1 -
// den r: v_rcp_f64 is good to 4.6e-8 here, e^3 ~ 1e-22), so r is the correctly rounded reciprocal up to 1 ulp and the
// result is within ~1 ulp of the reference expression (rounding-level, like the rest of the time-parallel evaluation;
// 7 operations per sample instead of 10).  Both band kinds share it.
// ORDER 2: one second-order step instead (r (1 + e): relative error e^2 <= 2.2e-15 in r, <= 0.4 ... 2 e-15 in the result
// for sat = 0.2 ... 1 -- the size of the other rounding errors of a band; one operation less).
// (the callers form the two constants once per channel and say that they are wave-uniform: computed here they sat in vector
// registers across the band loop, at the 128-register limit of the span kernel)
struct TpSmallConsts { double ca, cb; };
__device__ __forceinline__ TpSmallConsts tp_small_consts(double sat)
{
    const double c1 = 9.0 - 8.0 * sat;
    return { c1 * (1.0 / 9.0), 3.0 - c1 * (1.0 / 3.0) };
}
template <int N, int ORDER = 3>
__device__ __forceinline__ void tp_nonlinear_small(double (&y)[N], const TpSmallConsts& k)
{
    const double ca = k.ca, cb = k.cb;
    double den[N], r[N];
#pragma unroll
    for (int j = 0; j < N; ++j) den[j] = fma(y[j], y[j], 3.0);
    if (N == 4 && ORDER == 2) {
        // one reciprocal for the four denominators (v_rcp_f64 issues at a quarter of the FMA rate): 1 / (d0 d1 d2 d3),
        // refined once, then multiplied back apart -- 27 issue slots for four samples instead of 36.  The denominators
        // lie in [3, 3 + bound^2]: no over- or underflow in the products.
        const double p01 = den[0] * den[1], p23 = den[2 % N] * den[3 % N];
        const double pp = p01 * p23;
        double q = __builtin_amdgcn_rcp(pp);
        q = fma(fma(-pp, q, 1.0), q, q);
        q *= cb;                         // (cb folded into the shared reciprocal: every instruction below has ONE scalar operand)
        const double q01 = q * p23, q23 = q * p01;
        y[0] *= fma(q01, den[1], ca);
        y[1] *= fma(q01, den[0], ca);
        y[2 % N] *= fma(q23, den[3 % N], ca);
        y[3 % N] *= fma(q23, den[2 % N], ca);
        return;
    }
    if (N == 2 && ORDER == 2) {          // the same for a pair: 14 slots for two samples instead of 18
        const double pp = den[0] * den[1 % N];
        double q = __builtin_amdgcn_rcp(pp);
        q = fma(fma(-pp, q, 1.0), q, q);
        q *= cb;                         // cb / (den0 den1): one product for the pair, and ONE scalar operand per instruction
        y[0] *= fma(q, den[1 % N], ca);
        y[1 % N] *= fma(q, den[0], ca);
        return;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) r[j] = __builtin_amdgcn_rcp(den[j]);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double e = fma(-den[j], r[j], 1.0);
        r[j] = (ORDER == 2) ? fma(e, r[j], r[j]) : fma(fma(e, e, e), r[j], r[j]);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) y[j] *= fma(cb, r[j], ca);
}

// a * b + c with a in scalar registers, as the three-address instruction: for the output of a peaking band (m1 v1 + v0, v0 the
// sample's own register) the compiler's choice was a register copy plus the two-address v_fmac_f64 -- one more issue slot per sample
__device__ __forceinline__ double fma_sgpr(double a, double b, double c)
{
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "s"(a), "v"(b), "v"(c));
    return d;
}

// zero-state (or continuing) recurrence of one band over N samples held in registers: v[j] <- y_lin[j]
// KIND: 0 = SVF, packed stereo arithmetic (FMA), 1 = SVF scalar arithmetic (Left/Right modes), 2 = DF-II-T biquad
// of the OutputFilter (coefficients b0 b1 b2 a1 a2 in a1 a2 a3 m0 m1; state w1 w2 in ic1 ic2)
// CAP: the state behind sample capAt is copied to (c1, c2) -- the end state of a span whose last chunk is partly padding
// UNI: the coefficients are wave-uniform (scalar registers)
template <int KIND, int N, bool CAP = false, bool UNI = false>
__device__ __forceinline__ void tp_recur(double (&v)[N], double& ic1, double& ic2, double a1, double a2, double a3,
                                         double m0, double m1, double m2, int capAt = -1, double* c1 = nullptr, double* c2 = nullptr)
{
#pragma unroll
    for (int j = 0; j < N; ++j) {
        if (CAP && j > 0) { *c1 = (capAt == j - 1) ? ic1 : *c1; *c2 = (capAt == j - 1) ? ic2 : *c2; }
        const double v0 = v[j];
        if (KIND == 2) {
            const double yy = fma(a1, v0, ic1);
            const double n1 = fma(a2, v0, fma(-m0, yy, ic2));
            ic2 = fma(-m1, yy, a3 * v0);
            ic1 = n1;
            v[j] = yy;
            continue;
        }
        const double v3 = v0 - ic2;
        if (KIND == 3) {
            // SVF band with m0 == 1 and m2 == 0 (every peaking band): the output needs v1 only, and the second state follows
            // without v2:  ic2' = 2 v2 - ic2 = ic2 + 2 a2 ic1 + 2 a3 v3  (a1 = 2 a2, a2 = 2 a3 passed in: exact doublings).
            // Seven operations per sample instead of ten; same quantities, rounded in a different order.
            const double v1 = fma(m2, ic1, m0 * v3);             // m2 = a1, m0 = a2 of the band here
            ic2 = fma(a1, ic1, fma(a2, v3, ic2));
            ic1 = fma(2.0, v1, -ic1);
            v[j] = UNI ? fma_sgpr(m1, v1, v0) : fma(m1, v1, v0);
            continue;
        }
        if (KIND == 1) {
            const double v1 = a1 * ic1 + a2 * v3;
            const double v2 = ic2 + a2 * ic1 + a3 * v3;
            ic1 = 2.0 * v1 - ic1;
            ic2 = 2.0 * v2 - ic2;
            v[j] = m0 * v0 + m1 * v1 + m2 * v2;
        } else {
            const double v1 = fma(a1, ic1, a2 * v3);
            const double v2 = fma(a2, ic1, fma(a3, v3, ic2));
            ic1 = fma(2.0, v1, -ic1);
            ic2 = fma(2.0, v2, -ic2);
            v[j] = fma(m0, v0, fma(m1, v1, m2 * v2));
        }
    }
    if (CAP) { *c1 = (capAt == N - 1) ? ic1 : *c1; *c2 = (capAt == N - 1) ? ic2 : *c2; }
}

// cross-lane move of a double through the DPP network (2 x v_mov_b32_dpp, no LDS traffic); lanes whose source is
// outside the row / wave, and rows disabled by ROWMASK, receive +0.0
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int kDppRowShr = 0x110;      // + n: lane i <- lane i-n inside its row of 16
constexpr int kDppWaveShr1 = 0x138;    // lane i <- lane i-1 across the whole wave
constexpr int kDppRowBcast15 = 0x142;  // lane 15 of each row -> every lane of the next row
constexpr int kDppRowBcast31 = 0x143;  // lane 31 -> every lane of rows 2 and 3

// chunk start states of band b from the chunk end states (ic1, ic2) of the zero-state runs: inclusive scan of
// S_c = M S_(c-1) + e_c.  Inside a wave: 4 shift-and-combine steps within each row of 16 lanes (row_shr 1/2/4/8 with
// the powers A^(LC 2^k)), then lane 15 of a row carries into the next row and lane 31 into the upper half
// (row_bcast 15/31) with per-lane powers A^(LC (n+1)), n = lane mod 16 / mod 32; across the W waves the totals are
// chained through LDS as before.
// per-lane powers A^(LC (n+1)) for n = lane mod 16, lane mod 32, lane: global (L2) loads, to be issued well before the scan
struct TpLanePowers { double2 pa01, pa23, pb01, pb23, pc01, pc23; };
__device__ __forceinline__ TpLanePowers tp_load_powers(const double* __restrict__ Pglob, int lane)
{
    const double2* Pv = reinterpret_cast<const double2*>(Pglob);
    TpLanePowers p;
    p.pa01 = Pv[(lane & 15) * 2]; p.pa23 = Pv[(lane & 15) * 2 + 1];
    p.pb01 = Pv[(lane & 31) * 2]; p.pb23 = Pv[(lane & 31) * 2 + 1];
    p.pc01 = Pv[lane * 2];        p.pc23 = Pv[lane * 2 + 1];
    return p;
}

// ---- geometry and hand-over of the vector-form kernel (k_svf_cascade_tpv, below); tp_scan carries the hand-over
constexpr int kTpvWaves = 8;                    // waves of the span kernel
constexpr int kTpvSpan = kTpvWaves * 1024;      // samples per span
constexpr unsigned kTpvApplyGain = 1u << 31;    // bandFilter bit: this launch applies the channel's output gain
constexpr int kTpvQStride = 6;          // doubles per row of the quarter-chunk transposition buffer: 48 B, conflict-free b128 rows
constexpr int kTpvGuardPiece = 2048;    // samples per staged piece of the guarded path
constexpr int kTpvU = 4;                // samples per group of the small-signal output stage (one v_rcp_f64 per group; 4 fits
                                        // the registers since the E rows are scalar operands: 26 slots per 4 samples against 2 x 14)
constexpr unsigned kTpvSpinLimit = 1u << 22;    // polls of a hand-over before the launch gives up (seconds; a guarded span takes milliseconds)

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define CPQ_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// header of the chain buffer (64 bytes), the arrival counters of the CUs, then the granules [channel][chainSpans][band][4]
struct TpvChainHeader { unsigned gen, done, ticket, error, pad[12]; };
static_assert(sizeof(TpvChainHeader) == 64, "chain header");
constexpr int kTpvPlaces = 8 * 256;             // (XCD, shader engine / array / CU of HW_ID): one arrival counter each
constexpr int kTpvChainPrefix = (int)((sizeof(TpvChainHeader) + kTpvPlaces * sizeof(unsigned)) / sizeof(unsigned long long));
constexpr int kTpvSliceBit = 12;                // time slices of 2^12 ticks of the 100 MHz real-time counter: 41 us

// hand-over of one task (all wave-uniform)
struct TpvLink {
    const unsigned long long* poll;   // granules [band][4] of the same channel's span before this one; nullptr: first span
    unsigned long long* pub;          // granules of this span; nullptr: last span of the call
    unsigned epoch;
    int* flag;                        // LDS: != 0 once a start state outside the proven range has arrived (1 + band)
    unsigned* error;                  // header word: a poll gave up
    int slice;                        // 0 / 1: the time slices in which this workgroup runs at raised priority; -1: none
    const struct TpBandTables* tb;    // the stream's tables in device memory (the pass reads the next band's E rows through the scalar cache)
};

// Two workgroups of the span kernel share a CU, and the instruction arbiter serves the older wave first: left alone, the
// workgroup that arrived first runs at the pace it would have alone (55 us per span) and the other one at half of it (108 us),
// so with one workgroup per channel and as many channels as the chip has room for, half the workgroups finish at 2 / 3 of the
// launch and the CUs run the last third on two waves per SIMD (profiles/r04w_eq_workgroup_trace.txt).  Each workgroup
// therefore raises its priority in alternate slices of the (chip-wide) real-time counter, the first arrival on a CU in the even
// slices, the second in the odd ones: equal shares, both finish together.  A hint only: results do not depend on it.
// (the counter is read in front of a band's scan and used behind it: a scalar memory read, its latency under the scan)
__device__ __forceinline__ unsigned tpv_slice_clock(int slice)
{
    return slice < 0 ? 0u : (unsigned)__builtin_amdgcn_s_memrealtime();
}
__device__ __forceinline__ void tpv_time_slice(int slice, unsigned clock)
{
    if (slice < 0) return;
    asm volatile("" : "+s"(clock));          // (keeps the compiler from testing -- and waiting for -- the counter where it was read)
    if (((clock >> kTpvSliceBit) & 1u) == (unsigned)slice) __builtin_amdgcn_s_setprio(2);
    else                                                    __builtin_amdgcn_s_setprio(0);
}
// order of arrival of this workgroup on its CU (one thread): counters that are never reset -- two arrivals per CU and launch
// keep their parity, and any other sequence still alternates
__device__ __forceinline__ int tpv_arrival(unsigned* arrivals)
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_ID, XCC_ID
    const unsigned place = ((xcc & 7u) << 8) | ((hw >> 8) & 0xFFu);          // HW_ID[15:8]: shader engine, array, CU
    return (int)(__hip_atomic_fetch_add((gu32*)(arrivals + place), 1u, CPQ_RLX_AGENT) & 1u);
}

// lanes 0 ... 3 of the calling wave read one granule each until all four carry this launch's epoch
// (requesting the granules a band ahead of the poll was measured: 30 spilled registers in the chained kernel, 1.53 against
// 1.50 ms at 64 streams -- not kept)
__device__ __forceinline__ void tpv_poll_state(const unsigned long long* g, unsigned epoch, int lane, double& sx, double& sy, unsigned* error)
{
    unsigned long long v = 0;
    bool ok = lane >= 4;
    for (unsigned spins = 0;; ++spins) {
        if (!ok) {
            v = __hip_atomic_load((const gu64*)(g + lane), CPQ_RLX_AGENT);
            ok = (unsigned)(v >> 32) == epoch;
        }
        if (__all(ok)) break;
        // bounded: a launch that cannot make progress (a bug) sets the error word and runs out on whatever it reads
        if ((spins & 1023u) == 1023u &&
            (spins >= kTpvSpinLimit || __hip_atomic_load((const gu32*)error, CPQ_RLX_AGENT) != 0u)) {
            if (lane == 0) __hip_atomic_store((gu32*)error, 1u, CPQ_RLX_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    const int h = (int)(unsigned)v;
    sx = __hiloint2double(__builtin_amdgcn_readlane(h, 1), __builtin_amdgcn_readlane(h, 0));
    sy = __hiloint2double(__builtin_amdgcn_readlane(h, 3), __builtin_amdgcn_readlane(h, 2));
}
// one lane: the four granules of a band's end state
__device__ __forceinline__ void tpv_publish_state(unsigned long long* g, unsigned epoch, double sx, double sy)
{
    const unsigned long long e = (unsigned long long)epoch << 32;
    __hip_atomic_store((gu64*)(g + 0), e | (unsigned)__double2loint(sx), CPQ_RLX_AGENT);
    __hip_atomic_store((gu64*)(g + 1), e | (unsigned)__double2hiint(sx), CPQ_RLX_AGENT);
    __hip_atomic_store((gu64*)(g + 2), e | (unsigned)__double2loint(sy), CPQ_RLX_AGENT);
    __hip_atomic_store((gu64*)(g + 3), e | (unsigned)__double2hiint(sy), CPQ_RLX_AGENT);
}

// ONE workgroup barrier per band: the wave totals go through wtot[parity] (the caller flips the parity per band, so a
// wave that is already in the next band writes the other half while slow waves still read this one) and the span's
// end state goes to sNext while every wave reads the start state from sCur (the caller swaps the two per span).
// Plate != nullptr: the per-lane powers are loaded from there right where they are used (register-tight callers: the
// loads then wait on L2 behind the other waves of the SIMD) and pw is ignored.
// CHAINED (link != nullptr): the span's start state of band b arrives from the workgroup that has the span before (four
// lanes of wave 0 poll for it in front of the barrier), the end state is published to the workgroup that has the span behind.
template <int NTHREADS = kTpChunks, bool CHAINED = false>
__device__ __forceinline__ void tp_scan(double ic1, double ic2, double& s0x, double& s0y, const double* Mall, int b,
                                        const TpLanePowers& pw, double* wtot, double* sCur, double* sNext, int tid,
                                        const double* __restrict__ Plate = nullptr, int endTid = -1, const TpvLink* link = nullptr)
{
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: the chain below branches on it
    const double2* Pl = reinterpret_cast<const double2*>(Plate);
    double2 pa01 = pw.pa01, pa23 = pw.pa23, pb01 = pw.pb01, pb23 = pw.pb23, pc01 = pw.pc01, pc23 = pw.pc23;
    // band row of the LDS tables through a VGPR base, so that every read below is base + immediate offset
    uint32_t mOff = (uint32_t)b * (uint32_t)(28 * sizeof(double));      // Mall = [band][28]
    asm volatile("" : "+v"(mOff));
    const double* Mb = reinterpret_cast<const double*>(reinterpret_cast<const char*>(Mall) + mOff);
    const double2* Mb2 = reinterpret_cast<const double2*>(Mb);          // rows of 28 doubles: 16-byte aligned (TpLds / TpLdsM)
    double sx = ic1, sy = ic2;
#define CPQ_ROW_STEP(k)                                                                                       \
    {                                                                                                         \
        const double2 k01 = Mb2[(k) * 2], k23 = Mb2[(k) * 2 + 1];   /* 16-byte reads: one 16-bit immediate offset each */ \
        const double k0 = k01.x, k1 = k01.y, k2 = k23.x, k3 = k23.y;                                          \
        const double px = dpp_f64<kDppRowShr + (1 << (k)), 0xF>(sx);                                          \
        const double py = dpp_f64<kDppRowShr + (1 << (k)), 0xF>(sy);                                          \
        const double nx = fma(k1, py, fma(k0, px, sx));                                                       \
        const double ny = fma(k3, py, fma(k2, px, sy));                                                       \
        sx = nx;                                                                                              \
        sy = ny;                                                                                              \
    }
    CPQ_ROW_STEP(0)
    CPQ_ROW_STEP(1)
    CPQ_ROW_STEP(2)
    CPQ_ROW_STEP(3)
#undef CPQ_ROW_STEP
    if (Plate) { pa01 = Pl[(lane & 15) * 2]; pa23 = Pl[(lane & 15) * 2 + 1]; pb01 = Pl[(lane & 31) * 2]; pb23 = Pl[(lane & 31) * 2 + 1]; }
    {   // rows 1 and 3 <- total of the row below
        const double px = dpp_f64<kDppRowBcast15, 0xA>(sx);
        const double py = dpp_f64<kDppRowBcast15, 0xA>(sy);
        const double nx = fma(pa01.y, py, fma(pa01.x, px, sx));
        const double ny = fma(pa23.y, py, fma(pa23.x, px, sy));
        sx = nx;
        sy = ny;
    }
    {   // rows 2 and 3 <- total of the lower half
        const double px = dpp_f64<kDppRowBcast31, 0xC>(sx);
        const double py = dpp_f64<kDppRowBcast31, 0xC>(sy);
        const double nx = fma(pb01.y, py, fma(pb01.x, px, sx));
        const double ny = fma(pb23.y, py, fma(pb23.x, px, sy));
        sx = nx;
        sy = ny;
    }
    if (lane == 63) { wtot[2 * wave] = sx; wtot[2 * wave + 1] = sy; }
    if (Plate) { pc01 = Pl[lane * 2]; pc23 = Pl[lane * 2 + 1]; }
    if (CHAINED && link->poll && wave == 0) {
        double px, py;
        tpv_poll_state(link->poll + b * 4, link->epoch, lane, px, py, link->error);
        if (lane == 0) {
            sCur[2 * b] = px;
            sCur[2 * b + 1] = py;
            // outside the range the host proved guard-free: the span is run again on the guarded path (tpv_fast_span)
            if (!(fabs(px) < kTpInputBound && fabs(py) < kTpInputBound) && *link->flag == 0) *link->flag = 1 + b;
        }
    }
    __syncthreads();
    // state at the start of this wave's segment: the span's start state carried through the totals of the waves before it
    double bx = sCur[2 * b], by = sCur[2 * b + 1];
    const double2 mw01 = Mb2[12], mw23 = Mb2[13];
    const double mw0 = mw01.x, mw1 = mw01.y, mw2 = mw23.x, mw3 = mw23.y;
    for (int w = 0; w < wave; ++w) {
        const double tx = wtot[2 * w], ty = wtot[2 * w + 1];
        const double nx = fma(mw1, by, fma(mw0, bx, tx));
        const double ny = fma(mw3, by, fma(mw2, bx, ty));
        bx = nx;
        by = ny;
    }
    sx = fma(pc01.y, by, fma(pc01.x, bx, sx));
    sy = fma(pc23.y, by, fma(pc23.x, bx, sy));
    s0x = dpp_f64<kDppWaveShr1, 0xF>(sx);
    s0y = dpp_f64<kDppWaveShr1, 0xF>(sy);
    if (lane == 0) { s0x = bx; s0y = by; }
    // end of the span: the state behind the last chunk (endTid >= 0: behind chunk endTid -- a span whose tail is padding)
    if (tid == (endTid >= 0 ? endTid : (NTHREADS ? NTHREADS : (int)blockDim.x) - 1)) {
        sNext[2 * b] = sx;
        sNext[2 * b + 1] = sy;
        if (CHAINED && link->pub && *link->flag == 0) tpv_publish_state(link->pub + b * 4, link->epoch, sx, sy);
    }
}

// guarded sequential fallback for one band over the span held in LDS (one thread): the reference recurrence
// with every guard, used when the span input or the carried state is outside the proven-safe range.
template <int KIND>
__device__ void tp_band_guarded(double* buf, int lc, const double* cf, double sat, double* sState, int nSamples)
{
    const double a1 = cf[0], a2 = cf[1], a3 = cf[2], m0 = cf[3], m1 = cf[4], m2 = cf[5];
    const double oneMinusSat = 1.0 - sat;
    double ic1 = sState[0], ic2 = sState[1];
    for (int c = 0; c * lc < nSamples; ++c)
        for (int i = 0; i < lc && c * lc + i < nSamples; ++i) {
            double y[1] = { buf[c * kTpStride + i] };
            tp_recur<KIND, 1>(y, ic1, ic2, a1, a2, a3, m0, m1, m2);
            if (KIND == 2) {          // OutputFilter: no output stage, denormal flush of the state (OutputFilter.cpp:154-162)
                ic1 = (fabs(ic1) < 1.0e-20) ? 0.0 : ic1;
                ic2 = (fabs(ic2) < 1.0e-20) ? 0.0 : ic2;
            } else {
                if (sat > 0.0) tp_nonlinear<KIND == 1, true, 1>(y, sat, oneMinusSat);
                else           tp_nonlinear<KIND == 1, false, 1>(y, sat, oneMinusSat);
                ic1 = sanitize(ic1);
                ic2 = sanitize(ic2);
            }
            buf[c * kTpStride + i] = y[0];
        }
    sState[0] = ic1;
    sState[1] = ic2;
}

// LDS traffic between lanes of ONE wave: the hardware keeps a wave's DS operations in order; this keeps the compiler
// from moving them across each other
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------------------
// Vector form with the chunk in registers ("tpv").  fp64 MFMA and fp64 VALU share one datapath on gfx950 at the same
// rate (tools/ubench/coexec_f64.hip), so the dense form of a band's linear part -- T x on ten 4x4x4 block products, G s0
// on a 16x16x4, E x as 32 FMAs and a cross-lane reduction: ~16 FMA slots per sample -- costs more issue slots than the
// recurrence it replaces (10 operations per sample, end state included).  Here lane = one chunk of 16 consecutive
// samples held in 32 VGPRs through all bands, and per band
//   (1) the chunk start states come from the scan of the zero-state end states e = E x, which the PREVIOUS band's pass
//       accumulated from its outputs as it produced them (2 FMAs per sample),
//   (2) ONE pass over the 16 samples runs the reference recurrence from the true start state, applies the output stage
//       and feeds the next band's E x: no zero-state run, no state-response fix-up, no cross-lane traffic inside a band.
// Band coefficients are wave-uniform SGPR operands, the E rows come from LDS as broadcast reads, the per-lane scan powers
// live in LDS.  Span I/O: coalesced 16-byte accesses, transposed to chunk-per-lane through the wave's own padded LDS
// buffer (four quarters of four samples).
// WAVES > 0: whole spans of WAVES x 1024 samples; one workgroup per channel walks the spans of its channel, or (CHAINED)
// the spans of all channels are dealt to the workgroups through a ticket, below.  WAVES = 0: ONE span of blockDim.x / 64
// (1 ... 7) waves x 1024 samples (what a call leaves after its whole spans).
// A span whose input or start state is outside the range for which the host proved the reference's guards idle goes
// through the one-thread guarded recurrence (cold code, kept out of the band loop's register allocation).
//
// Chained spans (CHAINED; engines with fewer channels than the chip has room for workgroups).  A band's state at the
// end of span s is its state at the start of span s + 1: with the bands of a span run one behind the other, span s + 1
// can follow span s ONE BAND behind, on another workgroup.  The (span, channel) pairs are tasks in span-major order,
// taken by the workgroups through an atomic ticket: a task only ever waits for the task of the same channel one span
// earlier, whose ticket is lower and therefore already held by a running (or finished) workgroup -- no deadlock whatever
// the grid, the residency or the dispatch order.  The hand-over per (channel, span, band) is the band's end state, two
// doubles, as four 8-byte granules {epoch : 32 | half a double : 32}, each written by ONE write-through (sc1) store and
// polled with sc1 loads by four lanes of the consumer's first wave: the data is the flag, no fence on either side
// (a release / acquire pair per band writes back / invalidates the XCD's whole L2 and cost more than the chaining gained,
// profiles/r03b_eq_chained_spans.txt).  The epoch is the launch's generation, kept in device memory (header of the chain
// buffer: the last workgroup of a launch to finish advances it), so the granules are never cleared and a captured launch
// replays correctly.
// what the kernel keeps in LDS (8 waves: 76 KB, two workgroups per CU)
template <int MAXW>
struct TpvShared {
    // per wave 64 rows x kTpvQStride (span I/O); as a whole: kTpvGuardPiece / 16 rows x kTpStride of the guarded path
    alignas(16) double scratch[(MAXW * 64 * kTpvQStride > kTpvGuardPiece / 16 * kTpStride) ? MAXW * 64 * kTpvQStride
                                                                                           : kTpvGuardPiece / 16 * kTpStride];
    alignas(16) double P[kBands][64][4];                       // A^(16 (n + 1)): per-lane powers of the scan
    alignas(16) double M[kBands][28];                          // Mk[6][4], Mw[4] of every band (scan)
    alignas(16) double E[kBands][16][2];                       // (A^(15-k) B)_x, _y: end state of a chunk's zero-state run
    double stateA[kBands * 2], stateB[kBands * 2];
    alignas(16) double wtot[2 * 2 * MAXW];
    int flag;
    int task;                                                  // chained spans: the task the workgroup took; the launch's epoch
    unsigned epoch;
    int role;                                                  // chained spans: order of arrival on the CU (tpv_time_slice)
};

// max(|a|, |b|, |c|) of the HIGH words of three doubles, taken as floats: the bit pattern of a double's high word grows with
// |x| like a float's does, and the fastTanh clip threshold 4.5 has a zero low word, so |x| < 4.5 <=> high word (sign cleared) <
// 0x40120000 exactly.  One 32-bit instruction for three samples where v_cmp_lt_f64 takes one per sample.  (A high word that is
// a float NaN -- |x| >= 2^1017, infinities, NaNs -- is passed over by the maximum: the fast path only runs on spans whose inputs
// and states the host has proven to stay below 1e15, and a span that produced one anyway is met by the state check of the next.)
__device__ __forceinline__ float tpv_max3_hi(double a, double b, double c)
{
    float m;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(m) : "v"(__double2hiint(a)), "v"(__double2hiint(b)), "v"(__double2hiint(c)));
    return m;
}
__device__ __forceinline__ float tpv_max3_hi(float m, double b, double c)      // running maximum: one register
{
    float r;
    asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(r) : "v"(m), "v"(__double2hiint(b)), "v"(__double2hiint(c)));
    return r;
}

// One band over the lane's 16 samples from its true start state.  KIND 0 / 3: SVF band, general / with m0 == 1 and m2 == 0
// (seven operations per sample in the recurrence instead of ten) (the packed-stereo FMA arithmetic for
// both arithmetic flavours of the reference -- the time-parallel evaluation is rounding-level anyway; `mono` selects the
// scalar fastTanh's hard +-1 on the rare large-signal output stage), KIND 2: DF-II-T biquad of the OutputFilter.
// The recurrence runs over all 16 samples first: the output stage does not feed back into the state, so it follows as
// independent evaluations behind ONE wave-uniform test for the small-signal form.  En = the NEXT band's E rows in LDS.
// CAP (spans whose last chunk is partly padding): the lane with capAt >= 0 also returns the band's state behind its sample
// capAt in (c1, c2).
template <int KIND, bool SAT, bool CAP = false>
__device__ __forceinline__ void tpv_pass(double (&x)[16], double ic1, double ic2, const double* __restrict__ cfb, bool mono,
                                         const double* En, const double* Eglob, double& e0o, double& e1o,
                                         double sat, bool smallOk, const TpSmallConsts& smallK,
                                         int capAt = -1, double* c1 = nullptr, double* c2 = nullptr)
{
    {
        const double a1 = cfb[0], a2 = cfb[1], a3 = cfb[2], m0 = cfb[3], m1 = cfb[4], m2 = cfb[5];
        if (KIND == 3)          // m0 == 1 and m2 == 0 (peaking bands)
            tp_recur<3, 16, CAP, true>(x, ic1, ic2, 2.0 * a2, 2.0 * a3, 0.0, a2, m1, a1, capAt, c1, c2);
        else
        tp_recur<KIND, 16, CAP>(x, ic1, ic2, a1, a2, a3, m0, m1, m2, capAt, c1, c2);
    }
    double e0 = 0.0, e1 = 0.0;
    bool done = false;
    if (KIND != 2) {          // kind 2: linear section, no output stage
        // every |x[j]| below the clip threshold?  (tpv_max3_hi: nine instructions for the sixteen samples, one register)
        float mx = tpv_max3_hi(x[0], x[1], x[2]);
#pragma unroll
        for (int j = 3; j < 15; j += 2) mx = tpv_max3_hi(mx, x[j], x[j + 1]);
        mx = tpv_max3_hi(mx, x[15], x[15]);
        const int small = (int)(__float_as_uint(mx) < 0x40120000u);
        if (smallOk && __all(small)) {
            // kTpvU at a time, kept apart in the schedule: sixteen evaluations in flight at once do not fit the registers.
            // The next band's E rows are wave-uniform: read from the table in device memory through the scalar cache into scalar
            // registers, a group ahead of their use (from LDS they were 16 vector reads per band and a wait in every group)
            typedef const __attribute__((address_space(4))) double* ScalarPtr;
            const ScalarPtr g0 = (ScalarPtr)Eglob, g1 = g0 + 16;       // e[0][k], e[1][k]
            double ea[kTpvU], eb[kTpvU];
#pragma unroll
            for (int j = 0; j < kTpvU; ++j) { ea[j] = g0[j]; eb[j] = g1[j]; }
#pragma unroll
            for (int h = 0; h < 16 / kTpvU; ++h) {
                double na[kTpvU], nb[kTpvU];
#pragma unroll
                for (int j = 0; j < kTpvU; ++j) {
                    na[j] = (h + 1 < 16 / kTpvU) ? g0[kTpvU * (h + 1) + j] : 0.0;
                    nb[j] = (h + 1 < 16 / kTpvU) ? g1[kTpvU * (h + 1) + j] : 0.0;
                }
                double v[kTpvU];
#pragma unroll
                for (int j = 0; j < kTpvU; ++j) v[j] = x[kTpvU * h + j];
                if (SAT) tp_nonlinear_small<kTpvU, 2>(v, smallK);
#pragma unroll
                for (int j = 0; j < kTpvU; ++j) {
                    e0 = fma(ea[j], v[j], e0);
                    e1 = fma(eb[j], v[j], e1);
                    x[kTpvU * h + j] = v[j];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < kTpvU; ++j) { ea[j] = na[j]; eb[j] = nb[j]; }
            }
            done = true;
        } else {
            // rare: a sample at or above the fastTanh clip threshold somewhere in the wave
            // (1 - sat formed here, behind a value the compiler cannot see through: hoisted out of the band loop it held two vector
            // registers through the hot path, which sits at the register limit)
            double satHere = sat;
            asm volatile("" : "+s"(satHere));
            const double oneMinusSat = 1.0 - satHere;
#pragma unroll 1
            for (int h = 0; h < 4; ++h) {
                // rotate instead of indexing: x stays in registers
                double v[4] = { x[0], x[1], x[2], x[3] };
                if (mono) tp_nonlinear<true, SAT, 4, false>(v, sat, oneMinusSat);
                else      tp_nonlinear<false, SAT, 4, false>(v, sat, oneMinusSat);
#pragma unroll
                for (int j = 0; j < 12; ++j) x[j] = x[j + 4];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[12 + j] = v[j];
            }
        }
    }
    if (!done) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double2 ee = *reinterpret_cast<const double2*>(En + 2 * j);
            e0 = fma(ee.x, x[j], e0);
            e1 = fma(ee.y, x[j], e1);
        }
    }
    e0o = e0;
    e1o = e1;
}

// span <-> registers: 8 coalesced 16-byte accesses per lane, transposed through the wave's LDS buffer in four quarters
// (samples 4 h ... 4 h + 3 of every chunk: 64 rows of kTpvQStride).  Lane l of access k holds samples 2 (l & 7), + 1 of
// chunk 8 k + (l >> 3): the lanes with ((l >> 1) & 3) == h belong to quarter h.
__device__ __forceinline__ void tpv_span_load(const double* src, double* buf, int lane, double (&x)[16])
{
    typedef double v2 __attribute__((ext_vector_type(2)));
    v2 t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = __builtin_nontemporal_load(reinterpret_cast<const v2*>(src + k * 128 + lane * 2));
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        if (((lane >> 1) & 3) == h) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                *reinterpret_cast<double2*>(buf + (8 * k + (lane >> 3)) * kTpvQStride + 2 * (lane & 1)) = make_double2(t[k].x, t[k].y);
        }
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double2 v = *reinterpret_cast<const double2*>(buf + lane * kTpvQStride + 2 * j);
            x[4 * h + 2 * j] = v.x;
            x[4 * h + 2 * j + 1] = v.y;
        }
        wave_lds_sync();
    }
}
// The same for a span with nValid (< 1024 possible, any count) samples in this wave's part: 8-byte accesses, nothing read or
// written behind sample nValid - 1 (the padding reads as silence); rows of any alignment (a 441-sample call has odd rows).
__device__ __forceinline__ void tpv_span_load_partial(const double* src, double* buf, int lane, double (&x)[16], int nValid)
{
    double t[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int i = k * 128 + lane * 2;
        t[2 * k] = (i < nValid) ? src[i] : 0.0;
        t[2 * k + 1] = (i + 1 < nValid) ? src[i + 1] : 0.0;
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        if (((lane >> 1) & 3) == h) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                *reinterpret_cast<double2*>(buf + (8 * k + (lane >> 3)) * kTpvQStride + 2 * (lane & 1)) = make_double2(t[2 * k], t[2 * k + 1]);
        }
        wave_lds_sync();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const double2 v = *reinterpret_cast<const double2*>(buf + lane * kTpvQStride + 2 * j);
            x[4 * h + 2 * j] = v.x;
            x[4 * h + 2 * j + 1] = v.y;
        }
        wave_lds_sync();
    }
}
__device__ __forceinline__ void tpv_span_store_partial(double* dst, double* buf, int lane, const double (&x)[16], double gain, int nValid)
{
#pragma unroll
    for (int h = 0; h < 4; ++h) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<double2*>(buf + lane * kTpvQStride + 2 * j) = make_double2(x[4 * h + 2 * j] * gain, x[4 * h + 2 * j + 1] * gain);
        wave_lds_sync();
        if (((lane >> 1) & 3) == h) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double2 v = *reinterpret_cast<const double2*>(buf + (8 * k + (lane >> 3)) * kTpvQStride + 2 * (lane & 1));
                const int i = k * 128 + lane * 2;
                if (i < nValid) dst[i] = v.x;
                if (i + 1 < nValid) dst[i + 1] = v.y;
            }
        }
        wave_lds_sync();
    }
}
// The lanes of quarter h pick their eight 16-byte pieces out of the exchange buffer into t[8], assigned under `if (quarter
// == mine)`.  t[] would enter the quarter loop undefined, and the compiler then keeps a 32-register "don't care" tuple alive
// across the whole SPAN loop, spilled and reloaded per span (as much HBM traffic again as the span itself:
// profiles/r03n_pmc_traffic.json, r03o_ab_eq_span_store.txt); an empty asm defines the registers at the top of the function.
__device__ __forceinline__ void tpv_span_store(double* dst, double* buf, int lane, const double (&x)[16], double gain)
{
    typedef double v2 __attribute__((ext_vector_type(2)));
    v2 t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "=v"(t[k]));
#pragma unroll
    for (int h = 0; h < 4; ++h) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            *reinterpret_cast<double2*>(buf + lane * kTpvQStride + 2 * j) = make_double2(x[4 * h + 2 * j] * gain, x[4 * h + 2 * j + 1] * gain);
        wave_lds_sync();
        if (((lane >> 1) & 3) == h) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double2 v = *reinterpret_cast<const double2*>(buf + (8 * k + (lane >> 3)) * kTpvQStride + 2 * (lane & 1));
                t[k] = v2{ v.x, v.y };
            }
        }
        wave_lds_sync();
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(t[k], reinterpret_cast<v2*>(dst + k * 128 + lane * 2));
}

// Guarded run of ONE span: pieces of kTpvGuardPiece samples staged in the scratch area as [chunk][sample], one thread runs
// the reference recurrence with every guard, band by band; the states advance in sState.
template <class SH>
__device__ void tpv_guarded_span(SH& sh, int spanLen, const double* src, double* dst, double* sState, const double* cf,
                                 unsigned activeMask, unsigned long long kinds, double sat, double gain, int tid, int nThreads)
{
    for (int base = 0; base < spanLen; base += kTpvGuardPiece) {
        const int cnt = (spanLen - base < kTpvGuardPiece) ? spanLen - base : kTpvGuardPiece;
        __syncthreads();
        for (int j = tid; j < cnt; j += nThreads) sh.scratch[(j >> 4) * kTpStride + (j & 15)] = src[base + j];
        __syncthreads();
        if (tid == 0) {
            for (unsigned m = activeMask; m; m &= m - 1) {
                const int b = __builtin_ctz(m);
                const int kind = (int)((kinds >> (2 * b)) & 3);
                if (kind == 2)      tp_band_guarded<2>(sh.scratch, 16, cf + b * 6, sat, sState + 2 * b, cnt);
                else if (kind == 1) tp_band_guarded<1>(sh.scratch, 16, cf + b * 6, sat, sState + 2 * b, cnt);
                else                tp_band_guarded<0>(sh.scratch, 16, cf + b * 6, sat, sState + 2 * b, cnt);
            }
        }
        __syncthreads();
        for (int j = tid; j < cnt; j += nThreads) dst[base + j] = sh.scratch[(j >> 4) * kTpStride + (j & 15)] * gain;
    }
    __syncthreads();
}

// a wave-uniform double, said to be one (two scalar registers instead of two vector registers)
__device__ __forceinline__ double tpv_uniform(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

// the lane's index in its wave / the thread index, rebuilt from the lane count behind a value the compiler cannot see
// through: kept in a register across the band loops they end up in scratch (the kernel sits at the 128-register limit)
__device__ __forceinline__ int tpv_lane_id()
{
    int z = 0;
    asm volatile("" : "+s"(z));
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)z));
}

// The bands of `run` (one class: CLS 0 = SVF, 3 = SVF with output v0 + m1 v1, 2 = DF-II-T) over the span held in x: per band the scan of the chunk end
// states, then the pass.  e0 / e1: E x of the run's first band on entry, of the first band of `rest` (the bands behind
// the run) on exit.
// endTid >= 0 (PARTIAL spans only): the span ends behind sample capAt of chunk endTid (its tail is padding).
template <int CLS, bool SAT, int NT, bool CHAINED, bool PARTIAL, class SH>
__device__ __forceinline__ void tpv_band_run(double (&x)[16], double& e0, double& e1, int& par, unsigned run, unsigned rest,
                                             unsigned monoMask, SH& sh, double* sState, double* sNext, const double* __restrict__ cf,
                                             double sat, int waveU, int nThreads, const TpvLink& link, int endTid = -1, int capAt = 15)
{
    const bool smallOk = (sat >= 0.0) && (sat <= 1.0);
    TpSmallConsts smallK = tp_small_consts(sat);
    smallK.ca = tpv_uniform(smallK.ca);
    smallK.cb = tpv_uniform(smallK.cb);
    const TpLanePowers pw = {};                       // the per-lane powers come from LDS where tp_scan uses them
#pragma unroll 1
    while (run) {
        const int b = __builtin_ctz(run);
        run &= run - 1;
        const int nb = run ? __builtin_ctz(run) : (rest ? __builtin_ctz(rest) : b);     // last band: its E x result is not used
        double s0x, s0y;
        // (the thread index and the LDS addresses derived from it are rebuilt per band: tpv_lane_id)
        const int tidL = (waveU << 6) + tpv_lane_id();
        const unsigned clock = tpv_slice_clock(link.slice);
        // the scan is the band's chain of dependent cross-lane steps and LDS round trips, with the workgroup's barrier in it: run
        // ahead of everything else on the SIMD it is over sooner (1 ... 1.5 % on the kernel); the pass behind it runs at the slice's priority
        if (link.slice >= 0) __builtin_amdgcn_s_setprio(3);
        tp_scan<NT, CHAINED>(e0, e1, s0x, s0y, &sh.M[0][0], b, pw, sh.wtot + par * 2 * (nThreads >> 6), sState, sNext, tidL, &sh.P[b][0][0], PARTIAL ? endTid : -1, &link);
        tpv_time_slice(link.slice, clock);
        par ^= 1;
        if (PARTIAL) {
            // the band's end state is the one behind the span's last valid sample, which the pass meets inside chunk endTid
            // (the scan above left there the state behind that chunk's padding)
            double c1 = 0.0, c2 = 0.0;
            tpv_pass<CLS, SAT, true>(x, s0x, s0y, cf + b * 6, (monoMask >> b) & 1, &sh.E[nb][0][0], &link.tb[nb].mm.e[0][0], e0, e1, sat, smallOk, smallK,
                                     tidL == endTid ? capAt : -1, &c1, &c2);
            if (tidL == endTid) { sNext[2 * b] = c1; sNext[2 * b + 1] = c2; }
        } else {
            tpv_pass<CLS, SAT>(x, s0x, s0y, cf + b * 6, (monoMask >> b) & 1, &sh.E[nb][0][0], &link.tb[nb].mm.e[0][0], e0, e1, sat, smallOk, smallK);
        }
    }
}

// band masks of one channel: active bands of this launch; DF-II-T sections; SVF bands with the scalar fastTanh; SVF bands
// whose output is v0 + m1 v1 (every peaking band).  Wave-uniform by construction; said so, they live in scalar registers
// (left in vector registers they were spilled and reloaded inside the band loop).
struct TpvBands { unsigned active, df, mono, peak; unsigned long long kinds; };
// (lane b of every wave reads band b's flag word and the two coefficients of the peaking test; the masks are ballots.  A chained
// workgroup changes channel with every task: read band by band by every thread, this cost 60 dependent uniform loads per task)
__device__ __forceinline__ TpvBands tpv_band_masks(const int* __restrict__ flags, const double* __restrict__ cf, int c, unsigned bandFilter)
{
    const int lane = tpv_lane_id();
    int f = 0;
    double c3 = 0.0, c5 = 1.0;
    if (lane < kBands) {
        f = flags[c * kBands + lane];
        c3 = cf[lane * 6 + 3];
        c5 = cf[lane * 6 + 5];
    }
    const int kind = (f >> 1) & 3;          // 0 SVF stereo, 1 SVF scalar, 2 DF-II-T
    const unsigned k0 = (unsigned)__ballot(kind & 1), k1 = (unsigned)__ballot(kind & 2);
    TpvBands m;
    m.active = (unsigned)__ballot(f & 1) & bandFilter;      // bandFilter: the bands of this launch
    m.df = k1 & ~k0;
    m.mono = k0 & ~k1;
    m.peak = (unsigned)__ballot(lane < kBands && kind != 2 && c3 == 1.0 && c5 == 0.0);
    m.kinds = 0;                             // 2 bits per band (the guarded path reads them)
    for (int b = 0; b < kBands; ++b)
        m.kinds |= (unsigned long long)(((k0 >> b) & 1u) | (((k1 >> b) & 1u) << 1)) << (2 * b);
    m.active = __builtin_amdgcn_readfirstlane(m.active);
    m.df = __builtin_amdgcn_readfirstlane(m.df);
    m.mono = __builtin_amdgcn_readfirstlane(m.mono);
    m.peak = __builtin_amdgcn_readfirstlane(m.peak);
    return m;
}
// the tables of one stream into LDS: 16-byte pieces (Mk and Mw are contiguous in TpLcTables: 14 pieces per band; the E rows are
// paired up from the two rows of the host's e[2][16]), every load independent of the others
template <class SH>
__device__ __forceinline__ void tpv_load_tables(SH& sh, const TpBandTables* __restrict__ tb, int tid, int nThreads)
{
    for (int i = tid; i < kBands * 14; i += nThreads) {
        const int b = i / 14, q = i % 14;
        reinterpret_cast<double2*>(&sh.M[b][0])[q] = reinterpret_cast<const double2*>(&tb[b].t[0].Mk[0][0])[q];
    }
    for (int i = tid; i < kBands * 16; i += nThreads) {
        const int b = i >> 4, k = i & 15;
        *reinterpret_cast<double2*>(&sh.E[b][k][0]) = make_double2(tb[b].mm.e[0][k], tb[b].mm.e[1][k]);
    }
    for (int i = tid; i < kBands * 128; i += nThreads) {       // 16-byte pieces of P[64][4]
        const int b = i >> 7, q = i & 127;
        reinterpret_cast<double2*>(&sh.P[b][0][0])[q] = reinterpret_cast<const double2*>(&tb[b].t[0].P[0][0])[q];
    }
}

// One span on the fast path: load, range check, the band loop, store.  Returns 0 when done; 1 when the span's input (or,
// checkStates, a start state in sState) is outside the proven range: nothing has been stored or published; CHAINED: 2 + b
// when the start state of band b arrived out of range: the bands from b
// on have not been published, nothing has been stored -- the caller runs the span again on the guarded path.
// PARTIAL: the span holds nValid samples (1 ... 1024 x waves, any count): nValidW of them in this wave's part, the rest of the
// span is padding (read as silence, not written); the band states end behind sample nValid - 1.
template <int NT, bool CHAINED, bool PARTIAL, class SH>
__device__ __forceinline__ int tpv_fast_span(SH& sh, const double* srcW, double* dstW, const TpvBands& bm, const double* __restrict__ cf,
                                             double sat, double gain, double* sState, double* sNext, bool checkStates, int waveU,
                                             int nThreads, const TpvLink& link, int nValid = 0)
{
    const int nValidW = PARTIAL ? max(0, min(1024, nValid - waveU * 1024)) : 1024;
    const int endTid = PARTIAL ? (nValid - 1) >> 4 : -1, capAt = PARTIAL ? ((nValid - 1) & 15) : 15;
    double* buf = sh.scratch + waveU * 64 * kTpvQStride;
    double x[16];
    // (the per-lane addresses of the span I/O are rebuilt per span, like the thread index inside the band loop: kept across
    // the band loops they were the kernel's last spilled registers)
    int laneIo = tpv_lane_id();
    const int tidS = (waveU << 6) + laneIo;           // (= threadIdx.x, for the same reason)
    if (PARTIAL) tpv_span_load_partial(srcW, buf, laneIo, x, nValidW);
    else         tpv_span_load(srcW, buf, laneIo, x);
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 16; ++j) bad |= !(fabs(x[j]) < kTpInputBound);
    if (checkStates && tidS < kBands * 2) bad |= !(fabs(sState[tidS]) < kTpInputBound);
    if (tidS == 0) sh.flag = 0;
    __syncthreads();
    if (__any(bad) && laneIo == 0) atomicOr(&sh.flag, 1);
    __syncthreads();
    if (sh.flag != 0) return 1;

    // ---- the band loop: runs of one band class (an EQ channel is all SVF bands, an OutputFilter channel all DF-II-T
    // sections: one run), each in a loop body of its own
    if (bm.active) {
        double e0 = 0.0, e1 = 0.0;
        {
            const double* E = &sh.E[__builtin_ctz(bm.active)][0][0];     // the first band's E x on the raw input
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const double2 ee = *reinterpret_cast<const double2*>(E + 2 * j);
                e0 = fma(ee.x, x[j], e0);
                e1 = fma(ee.y, x[j], e1);
            }
        }
        int par = 0;
        unsigned mask = bm.active;
#pragma unroll 1
        while (mask) {
            const int bFirst = __builtin_ctz(mask);
            const int cls = ((bm.df >> bFirst) & 1) ? 2 : (((bm.peak >> bFirst) & 1) ? 3 : 0);
            const unsigned same = mask & (cls == 2 ? bm.df : (cls == 3 ? bm.peak : ~(bm.df | bm.peak)));
            const unsigned other = mask & ~same;
            const unsigned run = other ? (same & ((other & (0u - other)) - 1u)) : same;     // bands below the first one of another class
            const unsigned rest = mask & ~run;
#define CPQ_RUN(CLS, SAT) tpv_band_run<CLS, SAT, NT, CHAINED, PARTIAL>(x, e0, e1, par, run, rest, bm.mono, sh, sState, sNext, cf, sat, waveU, nThreads, link, endTid, capAt)
            if (cls == 2)        CPQ_RUN(2, false);
            else if (sat > 0.0) { if (cls == 3) CPQ_RUN(3, true); else CPQ_RUN(0, true); }
            else                { if (cls == 3) CPQ_RUN(3, false); else CPQ_RUN(0, false); }
#undef CPQ_RUN
            mask = rest;
        }
    }
    if (CHAINED) {
        // (set by the poller in front of a band's barrier: every thread is past that barrier here.  Nothing has been stored:
        // the caller's second run may read the input in place)
        const int f = sh.flag;
        if (f != 0) { __syncthreads(); return 1 + f; }
    }
    laneIo = tpv_lane_id();
    if (PARTIAL) tpv_span_store_partial(dstW, buf, laneIo, x, gain, nValidW);
    else         tpv_span_store(dstW, buf, laneIo, x, gain);
    __syncthreads();                      // the last thread's end states are in sNext; every wave's stores are out
    return 0;
}

template <int WAVES, bool CHAINED, bool PARTIAL = false>
__global__ __launch_bounds__((WAVES ? WAVES : 7) * 64, PARTIAL ? 2 : 4) void k_svf_cascade_tpv(const double* in, double* out, int64_t chStride,
                                                                              int nSpans, int nCh, const double* __restrict__ coef,
                                                                              const int* __restrict__ flags,
                                                                              const double* __restrict__ satGain,
                                                                              double* __restrict__ state,
                                                                              const TpBandTables* __restrict__ tables,
                                                                              unsigned long long* chain, int chainSpans,
                                                                              unsigned bandFilter, int nValid)
{
    // WAVES == 0: ONE span of blockDim.x / 64 (1 ... 7) waves x 1024 samples -- what a call leaves behind its whole 8192-sample
    // spans; PARTIAL: of nValid samples, any count from 1 to that (the span's tail is padding)
    constexpr bool kPartial = PARTIAL;
    constexpr int kMaxWaves = WAVES ? WAVES : 7;
    constexpr int kNT = WAVES * 64;                           // 0: blockDim.x
    __shared__ TpvShared<kMaxWaves> sh;
    const int tid = threadIdx.x, nThreads = WAVES ? kNT : (int)blockDim.x;
    const int waveU = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int spanLen = nThreads * 16;

    if (!CHAINED) {
        const int c = (int)blockIdx.x;
        const double* __restrict__ cf = coef + (int64_t)c * kBands * 6;
        const TpBandTables* __restrict__ tb = tables + (int64_t)(c >> 1) * kBands;
        const double sat = tpv_uniform(satGain[c * 2]), gain = tpv_uniform((bandFilter & kTpvApplyGain) ? satGain[c * 2 + 1] : 1.0);
        double* outCh = out + (int64_t)c * chStride;
        const double* inCh = in + (int64_t)c * chStride;
        const TpvBands bm = tpv_band_masks(flags, cf, c, bandFilter);
        double* sState = sh.stateA;
        double* sNext = sh.stateB;
        if (tid < kBands * 2) { sh.stateA[tid] = state[(int64_t)c * kBands * 2 + tid]; sh.stateB[tid] = sh.stateA[tid]; }
        // whole spans, one workgroup per channel: the two workgroups of a CU take turns at raised priority (tpv_time_slice)
        if (WAVES != 0 && tid == 0) sh.task = chain ? tpv_arrival(reinterpret_cast<unsigned*>(chain) + sizeof(TpvChainHeader) / sizeof(unsigned)) : -1;
        tpv_load_tables(sh, tb, tid, nThreads);
        __syncthreads();
        const TpvLink link = { nullptr, nullptr, 0u, nullptr, nullptr, WAVES != 0 ? __builtin_amdgcn_readfirstlane(sh.task) : -1, tb };

        int sp = 0;
#pragma unroll 1
        for (; sp < nSpans; ++sp) {
            if (tpv_fast_span<kNT, false, kPartial>(sh, inCh + (int64_t)sp * spanLen + waveU * 1024, outCh + (int64_t)sp * spanLen + waveU * 1024,
                                                    bm, cf, sat, gain, sState, sNext, true, waveU, nThreads, link, nValid) != 0)
                break;
            { double* t = sState; sState = sNext; sNext = t; }
        }
        const int tidE = (waveU << 6) + tpv_lane_id();            // (= tid, rebuilt: see tpv_lane_id)
        // cold: this span and the later ones through the guarded recurrence (states advance in sState)
        for (; sp < nSpans; ++sp)
            tpv_guarded_span(sh, kPartial ? nValid : spanLen, inCh + (int64_t)sp * spanLen, outCh + (int64_t)sp * spanLen, sState, cf, bm.active, bm.kinds,
                             sat, gain, tidE, nThreads);
        __syncthreads();
        // the call's end states (only this launch's bands are written)
        if (nSpans > 0 && tidE < kBands * 2 && ((bm.active >> (tidE >> 1)) & 1)) state[(int64_t)c * kBands * 2 + tidE] = sState[tidE];
    } else {
        TpvChainHeader* hdr = reinterpret_cast<TpvChainHeader*>(chain);
        unsigned long long* gran = chain + kTpvChainPrefix;
        if (tid == 0) sh.epoch = __hip_atomic_load((const gu32*)&hdr->gen, CPQ_RLX_AGENT) + 1u;
        // (the tickets balance the work whatever the arbiter does; the slices still gain 2.6 % at 64 streams: the scan's raised priority)
        if (tid == 0) sh.role = tpv_arrival(reinterpret_cast<unsigned*>(chain) + sizeof(TpvChainHeader) / sizeof(unsigned));
        const int nTasks = nSpans * nCh;
        int cur = -1;                     // channel whose tables are in LDS
        TpvBands bm = { 0, 0, 0, 0, 0 };
#pragma unroll 1
        for (;;) {
            __syncthreads();              // the task before is done with sh (tables, states, task word)
            if (tid == 0) sh.task = (int)__hip_atomic_fetch_add((gu32*)&hdr->ticket, 1u, CPQ_RLX_AGENT);
            __syncthreads();
            const int q = __builtin_amdgcn_readfirstlane(sh.task);
            if (q >= nTasks) break;
            const unsigned epoch = __builtin_amdgcn_readfirstlane(sh.epoch);
            const int sp = q / nCh, c = q - sp * nCh;
            const double* __restrict__ cf = coef + (int64_t)c * kBands * 6;
            const double sat = tpv_uniform(satGain[c * 2]), gain = tpv_uniform((bandFilter & kTpvApplyGain) ? satGain[c * 2 + 1] : 1.0);
            const int tidT = (waveU << 6) + tpv_lane_id();
            if (c != cur) {
                bm = tpv_band_masks(flags, cf, c, bandFilter);
                tpv_load_tables(sh, tables + (int64_t)(c >> 1) * kBands, tidT, nThreads);
                cur = c;
            }
            // first span of the call: the start states are the channel's stored states; later spans receive them band by band
            if (sp == 0 && tidT < kBands * 2) sh.stateA[tidT] = state[(int64_t)c * kBands * 2 + tidT];
            __syncthreads();
            unsigned long long* g = gran + ((int64_t)c * chainSpans + sp) * (kBands * 4);
            const TpvLink link = { sp > 0 ? g - kBands * 4 : nullptr, sp + 1 < nSpans ? g : nullptr, epoch, &sh.flag, &hdr->error, __builtin_amdgcn_readfirstlane(sh.role),
                                   tables + (int64_t)(c >> 1) * kBands };
            const double* src = in + (int64_t)c * chStride + (int64_t)sp * spanLen;
            double* dst = out + (int64_t)c * chStride + (int64_t)sp * spanLen;
            const int r = tpv_fast_span<kNT, true, false>(sh, src + waveU * 1024, dst + waveU * 1024, bm, cf, sat, gain, sh.stateA, sh.stateB,
                                                          sp == 0, waveU, nThreads, link);
            const int tidE = (waveU << 6) + tpv_lane_id();
            if (r != 0) {
                // cold: the span on the guarded path.  Its start states: all of them from the span before (that workgroup has
                // then finished the span), or the stored ones; its end states are published for the bands the fast path has not
                // published (r >= 2: the bands below r - 2 went out before the out-of-range state arrived, from good states).
                if (link.poll && waveU == 0) {
                    for (unsigned m = bm.active; m; m &= m - 1) {
                        const int b = __builtin_ctz(m);
                        double sx, sy;
                        tpv_poll_state(link.poll + b * 4, epoch, tidE, sx, sy, link.error);
                        if (tidE == 0) { sh.stateA[2 * b] = sx; sh.stateA[2 * b + 1] = sy; }
                    }
                } else if (!link.poll && tidE < kBands * 2) {
                    sh.stateA[tidE] = state[(int64_t)c * kBands * 2 + tidE];
                }
                __syncthreads();
                tpv_guarded_span(sh, spanLen, src, dst, sh.stateA, cf, bm.active, bm.kinds, sat, gain, tidE, nThreads);
                if (tidE < kBands * 2) sh.stateB[tidE] = sh.stateA[tidE];
                __syncthreads();
                if (link.pub && tidE == 0) {
                    for (unsigned m = bm.active; m; m &= m - 1) {
                        const int b = __builtin_ctz(m);
                        if (r >= 2 && b < r - 2) continue;
                        tpv_publish_state(link.pub + b * 4, epoch, sh.stateB[2 * b], sh.stateB[2 * b + 1]);
                    }
                }
            }
            // last span of the call: the end states (in stateB, band by band) are the channel's stored states
            if (sp + 1 == nSpans && tidE < kBands * 2 && ((bm.active >> (tidE >> 1)) & 1)) state[(int64_t)c * kBands * 2 + tidE] = sh.stateB[tidE];
        }
        // the last workgroup to get here resets the ticket and advances the generation for the next launch
        if (tid == 0) {
            const unsigned epoch = sh.epoch;
            if (__hip_atomic_fetch_add((gu32*)&hdr->done, 1u, CPQ_RLX_AGENT) == gridDim.x - 1u) {
                __hip_atomic_store((gu32*)&hdr->ticket, 0u, CPQ_RLX_AGENT);
                __hip_atomic_store((gu32*)&hdr->done, 0u, CPQ_RLX_AGENT);
                __hip_atomic_store((gu32*)&hdr->gen, epoch, CPQ_RLX_AGENT);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// Spans below 1024 samples: one 512- / 480- / 441-sample callback per call -- the reference's own call pattern
// (src/convolver/ConvolverProcessor.Runtime.cpp:659-682) -- and what a ragged call leaves.  A launch of this size is nothing but
// the latency of 20 dependent bands, so the kernel is built for latency: ONE wave per channel up to 512 samples (two up to 1023)
// with the lane's chunk of 8 samples in registers, vector form as above (scan of the E x sums, then one pass from the true start
// state), no barrier and no cross-wave chain on one wave, no occupancy target (the whole register file is the wave's), every
// per-band constant in LDS or requested a band ahead.  Any sample count: the tail of the last chunk is padding, the band
// states are taken behind the last valid sample inside the pass.
template <int LC>
struct ShortShared {
    alignas(16) double M[kBands][28];
    alignas(16) double E[kBands][LC][2];
    alignas(16) double cf[kBands][6];
    alignas(16) double stateA[kBands * 2];
    alignas(16) double stateB[kBands * 2];
    alignas(16) double wtot[2 * 2 * 4];
    alignas(16) double buf[(1024 / 16) * kTpStride];      // guarded path: the span as [chunk of 16][sample]
    int flag;
};

// A band's constants in registers: the scan's matrix powers, the band's coefficients, the NEXT active band's E rows.  One
// batch of LDS reads per band, issued a band ahead of its use: with one wave per SIMD nothing else hides an LDS round trip
// (~130 cycles), and read where they are used the ~40 reads of a band were most of its time (SQ_WAIT_ANY 53 %).
template <int LC>
struct ShortConsts { double2 m[14]; double2 cf[3]; double2 e[LC]; double2 s0; };      // s0: the band's state at the start of the span
template <int LC>
__device__ __forceinline__ ShortConsts<LC> short_load_consts(const ShortShared<LC>& sh, int b, int nb)
{
    ShortConsts<LC> k;
#pragma unroll
    for (int i = 0; i < 14; ++i) k.m[i] = *reinterpret_cast<const double2*>(&sh.M[b][2 * i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) k.cf[i] = *reinterpret_cast<const double2*>(&sh.cf[b][2 * i]);
#pragma unroll
    for (int i = 0; i < LC; ++i) k.e[i] = *reinterpret_cast<const double2*>(&sh.E[nb][i][0]);
    k.s0 = *reinterpret_cast<const double2*>(&sh.stateA[2 * b]);
    return k;
}

// tp_scan for one to four waves with the matrix powers in registers (k.m: Mk[6][4], Mw[4] as 14 pairs)
template <int LC>
__device__ __forceinline__ void short_scan(double ic1, double ic2, double& s0x, double& s0y, const ShortConsts<LC>& k, const TpLanePowers& pw,
                                           double* wtot, double* sNext, int b, int tid, int nWaves, int endTid)
{
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double sx = ic1, sy = ic2;
#define CPQ_ROW_STEP(K)                                                                                       \
    {                                                                                                         \
        const double2 k01 = k.m[2 * (K)], k23 = k.m[2 * (K) + 1];                                             \
        const double px = dpp_f64<kDppRowShr + (1 << (K)), 0xF>(sx);                                          \
        const double py = dpp_f64<kDppRowShr + (1 << (K)), 0xF>(sy);                                          \
        const double nx = fma(k01.y, py, fma(k01.x, px, sx));                                                 \
        const double ny = fma(k23.y, py, fma(k23.x, px, sy));                                                 \
        sx = nx;                                                                                              \
        sy = ny;                                                                                              \
    }
    CPQ_ROW_STEP(0)
    CPQ_ROW_STEP(1)
    CPQ_ROW_STEP(2)
    CPQ_ROW_STEP(3)
#undef CPQ_ROW_STEP
    {   // rows 1 and 3 <- total of the row below
        const double px = dpp_f64<kDppRowBcast15, 0xA>(sx);
        const double py = dpp_f64<kDppRowBcast15, 0xA>(sy);
        const double nx = fma(pw.pa01.y, py, fma(pw.pa01.x, px, sx));
        const double ny = fma(pw.pa23.y, py, fma(pw.pa23.x, px, sy));
        sx = nx;
        sy = ny;
    }
    {   // rows 2 and 3 <- total of the lower half
        const double px = dpp_f64<kDppRowBcast31, 0xC>(sx);
        const double py = dpp_f64<kDppRowBcast31, 0xC>(sy);
        const double nx = fma(pw.pb01.y, py, fma(pw.pb01.x, px, sx));
        const double ny = fma(pw.pb23.y, py, fma(pw.pb23.x, px, sy));
        sx = nx;
        sy = ny;
    }
    double bx = k.s0.x, by = k.s0.y;          // the span's start state
    if (nWaves > 1) {
        if (lane == 63) { wtot[2 * wave] = sx; wtot[2 * wave + 1] = sy; }
        __syncthreads();
        for (int w = 0; w < wave; ++w) {          // (wave-uniform)
            const double tx = wtot[2 * w], ty = wtot[2 * w + 1];
            const double nx = fma(k.m[12].y, by, fma(k.m[12].x, bx, tx));
            const double ny = fma(k.m[13].y, by, fma(k.m[13].x, bx, ty));
            bx = nx;
            by = ny;
        }
    }
    sx = fma(pw.pc01.y, by, fma(pw.pc01.x, bx, sx));
    sy = fma(pw.pc23.y, by, fma(pw.pc23.x, bx, sy));
    s0x = dpp_f64<kDppWaveShr1, 0xF>(sx);
    s0y = dpp_f64<kDppWaveShr1, 0xF>(sy);
    if (lane == 0) { s0x = bx; s0y = by; }
    if (tid == endTid) { sNext[2 * b] = sx; sNext[2 * b + 1] = sy; }
}

template <int CLS, bool SAT, int LC>
__device__ __forceinline__ void short_pass(double (&x)[LC], double ic1, double ic2, const ShortConsts<LC>& k, bool mono,
                                           double& e0, double& e1, double sat, int capAt, double& c1, double& c2)
{
    const double a1 = k.cf[0].x, a2 = k.cf[0].y, a3 = k.cf[1].x, m0 = k.cf[1].y, m1 = k.cf[2].x, m2 = k.cf[2].y;
    if (CLS == 3) tp_recur<3, LC, true>(x, ic1, ic2, 2.0 * a2, 2.0 * a3, 0.0, a2, m1, a1, capAt, &c1, &c2);
    else          tp_recur<CLS, LC, true>(x, ic1, ic2, a1, a2, a3, m0, m1, m2, capAt, &c1, &c2);
    if (CLS != 2) {           // class 2: linear section (OutputFilter), no output stage
        int small = 1;
#pragma unroll
        for (int j = 0; j < LC; ++j) small &= (int)(fabs(x[j]) < 4.5);
        if (sat >= 0.0 && sat <= 1.0 && __all(small)) {
            if (SAT) {
                const TpSmallConsts ks = tp_small_consts(sat);
#pragma unroll
                for (int h = 0; h < LC / 2; ++h) {
                    double v[2] = { x[2 * h], x[2 * h + 1] };
                    tp_nonlinear_small<2, 2>(v, ks);
                    x[2 * h] = v[0];
                    x[2 * h + 1] = v[1];
                }
            }
        } else {
            const double oneMinusSat = 1.0 - sat;
#pragma unroll
            for (int h = 0; h < LC / 4; ++h) {
                double v[4] = { x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3] };
                if (mono) tp_nonlinear<true, SAT, 4, false>(v, sat, oneMinusSat);
                else      tp_nonlinear<false, SAT, 4, false>(v, sat, oneMinusSat);
#pragma unroll
                for (int j = 0; j < 4; ++j) x[4 * h + j] = v[j];
            }
        }
    }
    double f0 = 0.0, f1 = 0.0;
#pragma unroll
    for (int j = 0; j < LC; ++j) {
        f0 = fma(k.e[j].x, x[j], f0);
        f1 = fma(k.e[j].y, x[j], f1);
    }
    e0 = f0;
    e1 = f1;
}

template <int LC>
__global__ __launch_bounds__(256, 1) void k_svf_cascade_short(const double* in, double* out, int64_t chStride, int n,
                                                             const double* __restrict__ coef, const int* __restrict__ flags,
                                                             const double* __restrict__ satGain, double* __restrict__ state,
                                                             const TpBandTables* __restrict__ tables)
{
    constexpr int LI = 1;                          // table block of the chunk length (host_design.hpp: kSvfTpLc)
    static_assert(kSvfTpLc[LI] == LC, "chunk length without a table block");
    __shared__ ShortShared<LC> sh;
    const int tid = threadIdx.x, nThreads = blockDim.x, lane = tid & 63;
    const int c = blockIdx.x;
    const double* __restrict__ cfg = coef + (int64_t)c * kBands * 6;
    const TpBandTables* __restrict__ tb = tables + (int64_t)(c >> 1) * kBands;
    const double sat = satGain[c * 2], gain = satGain[c * 2 + 1];
    const double* src = in + (int64_t)c * chStride;
    double* dst = out + (int64_t)c * chStride;
    // the lane's chunk first (the longest latency), the tables behind it
    double x[LC];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < LC; ++j) {
        const int i = tid * LC + j;
        x[j] = (i < n) ? src[i] : 0.0;
    }
    for (int i = tid; i < kBands * 28; i += nThreads) {
        const int b = i / 28, q = i % 28;
        sh.M[b][q] = (q < 24) ? tb[b].t[LI].Mk[q / 4][q % 4] : tb[b].t[LI].Mw[q - 24];
    }
    for (int i = tid; i < kBands * LC * 2; i += nThreads) {
        const int b = i / (LC * 2), k = (i % (LC * 2)) >> 1, r = i & 1;
        sh.E[b][k][r] = tb[b].mm.e[r][16 - LC + k];          // A^(LC - 1 - k) B: the last LC columns of the 16-sample map
    }
    for (int i = tid; i < kBands * 6; i += nThreads) sh.cf[i / 6][i % 6] = cfg[i];
    if (tid < kBands * 2) { sh.stateA[tid] = state[(int64_t)c * kBands * 2 + tid]; sh.stateB[tid] = sh.stateA[tid]; }
    if (tid == 0) sh.flag = 0;
    unsigned active = 0, dfMask = 0, monoMask = 0;
    for (int b = 0; b < kBands; ++b) {
        const int f = flags[c * kBands + b];
        active |= (unsigned)(f & 1) << b;
        dfMask |= (unsigned)(((f >> 1) & 3) == 2) << b;
        monoMask |= (unsigned)(((f >> 1) & 3) == 1) << b;
    }
    active = __builtin_amdgcn_readfirstlane(active);
    dfMask = __builtin_amdgcn_readfirstlane(dfMask);
    monoMask = __builtin_amdgcn_readfirstlane(monoMask);
#pragma unroll
    for (int j = 0; j < LC; ++j) bad |= !(fabs(x[j]) < kTpInputBound);
    __syncthreads();
    if (tid < kBands * 2) bad |= !(fabs(sh.stateA[tid]) < kTpInputBound);
    if (__any(bad) && lane == 0) atomicOr(&sh.flag, 1);
    __syncthreads();
    if (sh.flag != 0) {
        // cold: input or a start state outside the range the host proved guard-free -- the reference recurrence with every guard
        for (int j = tid; j < n; j += nThreads) sh.buf[(j >> 4) * kTpStride + (j & 15)] = src[j];
        __syncthreads();
        if (tid == 0) {
            for (unsigned m = active; m; m &= m - 1) {
                const int b = __builtin_ctz(m);
                if ((dfMask >> b) & 1)        tp_band_guarded<2>(sh.buf, 16, sh.cf[b], sat, sh.stateA + 2 * b, n);
                else if ((monoMask >> b) & 1) tp_band_guarded<1>(sh.buf, 16, sh.cf[b], sat, sh.stateA + 2 * b, n);
                else                          tp_band_guarded<0>(sh.buf, 16, sh.cf[b], sat, sh.stateA + 2 * b, n);
            }
        }
        __syncthreads();
        for (int j = tid; j < n; j += nThreads) dst[j] = sh.buf[(j >> 4) * kTpStride + (j & 15)] * gain;
        if (tid < kBands * 2 && ((active >> (tid >> 1)) & 1)) state[(int64_t)c * kBands * 2 + tid] = sh.stateA[tid];
        return;
    }
    if (active) {
        const int endTid = (n - 1) / LC, capAt = (n - 1) % LC;
        double e0 = 0.0, e1 = 0.0;
        {
            const double* E = &sh.E[__builtin_ctz(active)][0][0];
#pragma unroll
            for (int j = 0; j < LC; ++j) {
                const double2 ee = *reinterpret_cast<const double2*>(E + 2 * j);
                e0 = fma(ee.x, x[j], e0);
                e1 = fma(ee.y, x[j], e1);
            }
        }
        int par = 0;
        const int nWaves = nThreads >> 6;
        const int first = __builtin_ctz(active);
        const unsigned rest0 = active & (active - 1);
        TpLanePowers pw = tp_load_powers(&tb[first].t[LI].P[0][0], lane);
        ShortConsts<LC> kc = short_load_consts<LC>(sh, first, rest0 ? __builtin_ctz(rest0) : first);
#pragma unroll 1
        for (unsigned m = active; m; m &= m - 1) {
            const int b = __builtin_ctz(m);
            const unsigned rest = m & (m - 1);
            const int nb = rest ? __builtin_ctz(rest) : b;
            const unsigned rest2 = rest & (rest - 1);
            // the next band's constants and per-lane powers, requested now (used a band from here)
            const TpLanePowers pwNext = tp_load_powers(&tb[nb].t[LI].P[0][0], lane);
            const ShortConsts<LC> kn = short_load_consts<LC>(sh, nb, rest2 ? __builtin_ctz(rest2) : nb);
            double s0x, s0y;
            short_scan(e0, e1, s0x, s0y, kc, pw, sh.wtot + par * 2 * 4, sh.stateB, b, tid, nWaves, endTid);
            par ^= 1;
            const bool peak = !((dfMask >> b) & 1) && kc.cf[1].y == 1.0 && kc.cf[2].y == 0.0;
            const bool mono = (monoMask >> b) & 1;
            const int cap = (tid == endTid) ? capAt : -1;
            double c1 = 0.0, c2 = 0.0;
            if ((dfMask >> b) & 1)   short_pass<2, false, LC>(x, s0x, s0y, kc, mono, e0, e1, sat, cap, c1, c2);
            else if (sat > 0.0) { if (peak) short_pass<3, true, LC>(x, s0x, s0y, kc, mono, e0, e1, sat, cap, c1, c2);
                                  else      short_pass<0, true, LC>(x, s0x, s0y, kc, mono, e0, e1, sat, cap, c1, c2); }
            else                { if (peak) short_pass<3, false, LC>(x, s0x, s0y, kc, mono, e0, e1, sat, cap, c1, c2);
                                  else      short_pass<0, false, LC>(x, s0x, s0y, kc, mono, e0, e1, sat, cap, c1, c2); }
            pw = pwNext;
            kc = kn;
            // the band's end state: behind the last valid sample (the scan left there the state behind the chunk's padding)
            if (tid == endTid) { sh.stateB[2 * b] = c1; sh.stateB[2 * b + 1] = c2; }
        }
    }
#pragma unroll
    for (int j = 0; j < LC; ++j) {
        const int i = tid * LC + j;
        if (i < n) dst[i] = x[j] * gain;
    }
    __syncthreads();
    if (tid < kBands * 2 && ((active >> (tid >> 1)) & 1)) state[(int64_t)c * kBands * 2 + tid] = sh.stateB[tid];
}
}  // namespace

void launch_svf_cascade(hipStream_t stream, const double* in, double* out, int64_t chStride, int nCh, int nSamples,
                        const double* coef, const int* flags, const double* satGain, double* state, bool streamPairs)
{
    if (streamPairs)
        hipLaunchKernelGGL(k_svf_cascade<2>, dim3((nCh + 1) / 2), dim3(64), 0, stream, in, out, chStride, nCh, nSamples,
                           coef, flags, satGain, state);
    else
        hipLaunchKernelGGL(k_svf_cascade<3>, dim3((nCh + 2) / 3), dim3(64), 0, stream, in, out, chStride, nCh, nSamples,
                           coef, flags, satGain, state);
}

int svf_chain_spans(int maxSamples) { return maxSamples / kTpvSpan + 1; }
size_t svf_chain_bytes(int nCh, int maxSamples)
{
    // header + arrival counters + granules [channel][span][band][4] of the chained spans (maxSamples = 0: no chained spans)
    const size_t granules = maxSamples > 0 ? (size_t)nCh * (size_t)svf_chain_spans(maxSamples) * kBands * 4 : 0;
    return (kTpvChainPrefix + granules) * sizeof(unsigned long long);
}

void launch_svf_cascade_tp(hipStream_t stream, const double* in, double* out, int64_t chStride, int nCh, int nSamples,
                           const double* coef, const int* flags, const double* satGain, double* state,
                           const void* tables, void* chain, int chainSpans, int chainGrid)
{
    static_assert(sizeof(TpBandTables) == kSvfTpTableDoubles * sizeof(double), "host/device table layout");
    // whole 8192-sample spans on the eight-wave kernel, what is left as one span of 1 ... 7 waves x 1024 samples, and a
    // last block of 512 on the chunk-length-2 path of the four-wave kernel
    const TpBandTables* tb = reinterpret_cast<const TpBandTables*>(tables);
    constexpr unsigned kAllBands = 0xFFFFFu | kTpvApplyGain;
    int done = 0;
    const int nSpans8 = nSamples / kTpvSpan;
    if (nSpans8 > 0) {
        // chainSpans > 0 (the engine's choice: channel counts that do not fill whole rounds of chainGrid = 2 per CU workgroups):
        // chained spans, the (span, channel) tasks dealt to chainGrid workgroups.  Otherwise one workgroup per channel.
        if (chain && chainSpans > 0 && nSpans8 >= 2 && nSpans8 <= chainSpans) {
            const int nTasks = nSpans8 * nCh;
            hipLaunchKernelGGL((k_svf_cascade_tpv<kTpvWaves, true>), dim3(nTasks < chainGrid ? nTasks : chainGrid), dim3(kTpvWaves * 64), 0, stream,
                               in, out, chStride, nSpans8, nCh, coef, flags, satGain, state, tb, reinterpret_cast<unsigned long long*>(chain),
                               chainSpans, kAllBands, 0);
        } else {
            hipLaunchKernelGGL((k_svf_cascade_tpv<kTpvWaves, false>), dim3(nCh), dim3(kTpvWaves * 64), 0, stream, in, out, chStride, nSpans8, nCh,
                               coef, flags, satGain, state, tb, reinterpret_cast<unsigned long long*>(chain), 0, kAllBands, 0);
        }
        done = nSpans8 * kTpvSpan;
    }
    // What is left behind the whole spans (and calls shorter than one).  From 1024 samples up: ONE span of up to seven waves x
    // 1024 samples on the same kernel, its tail padding where the count is not a multiple of 1024 (two launches for 7169 ... 8191
    // samples).  Below 1024 samples the chain of 20 dependent bands is all a launch costs, and 64 chunks of eight samples on one
    // wave walk it faster than 32 chunks of sixteen (0.034 against 0.067 ms per 512-sample callback,
    // profiles/r04f_eq_short_calls.txt): k_svf_cascade_short, any sample count in one launch.
    while (nSamples - done >= 1024) {
        int cnt = nSamples - done;
        if (cnt > 7 * 1024) cnt = 4 * 1024;
        const dim3 block(((cnt + 1023) / 1024) * 64);
        if (cnt % 1024 == 0)
            hipLaunchKernelGGL((k_svf_cascade_tpv<0, false, false>), dim3(nCh), block, 0, stream, in + done, out + done, chStride, 1, nCh,
                               coef, flags, satGain, state, tb, (unsigned long long*)nullptr, 0, kAllBands, cnt);
        else
            hipLaunchKernelGGL((k_svf_cascade_tpv<0, false, true>), dim3(nCh), block, 0, stream, in + done, out + done, chStride, 1, nCh,
                               coef, flags, satGain, state, tb, (unsigned long long*)nullptr, 0, kAllBands, cnt);
        done += cnt;
    }
    if (nSamples > done) {
        // (a chunk of 4 samples on two to four waves measured the same 0.034 ms per 512-sample callback as this chunk of 8 on
        // one or two: profiles/r04f_eq_short_calls.txt)
        const int cnt = nSamples - done;
        hipLaunchKernelGGL(k_svf_cascade_short<8>, dim3(nCh), dim3(((cnt + 511) / 512) * 64), 0, stream, in + done, out + done, chStride, cnt,
                           coef, flags, satGain, state, tb);
    }
}
}  // namespace cpq
