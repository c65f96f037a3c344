// tools/variants/svf_wave_pipeline.hip -- NOT built.  Round-3 experiment: the vector-form EQ kernel with the waves of a channel
// pipelined along the time axis (one 1024-sample piece per wave, band states handed from wave to wave through LDS slots,
// no workgroup barrier; poison + sequential fix-up pass for out-of-range pieces).  Parity-green (79 EQ tests) but its
// time swung between 6.2 and 20 ms with code placement (profiles/r03b_ab_eq_forms.txt); the barrier version was kept.

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
// kChPerWave = 1 with `redo`: the fix-up pass behind the time-parallel kernel (below) -- channel c is processed from sample
// redo[c].piece * 1024 on if redo[c].ticket names this launch, else not at all.
struct SvfRedo { unsigned long long ticket; int piece; int pad; };

template <int kChPerWave>
__global__ __launch_bounds__(64) void k_svf_cascade(const double* in, double* out,
                                                    int64_t chStride, int nCh, int nSamples,
                                                    const double* __restrict__ coef, const int* __restrict__ flags,
                                                    const double* __restrict__ satGain, double* __restrict__ state,
                                                    const SvfRedo* __restrict__ redo = nullptr, unsigned long long ticket = 0)
{
    if (kChPerWave == 1 && redo) {
        const SvfRedo r = redo[blockIdx.x];
        if (r.ticket != ticket || (int64_t)r.piece * 1024 >= nSamples) return;      // wave-uniform
        in += (int64_t)r.piece * 1024;
        out += (int64_t)r.piece * 1024;
        nSamples -= r.piece * 1024;
    }
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
// Time-parallel variant.
//
// Inside one band the state update is LINEAR in (v0, ic1eq, ic2eq): the fastTanh blend, the +-100 clamp and
// the output guard act on the band OUTPUT only and never feed back into the state (Processing.cpp:228-262).
// So one band over a span of 64*W chunks of LC samples can be run as
//   1. every lane runs the reference recurrence over its own chunk from ZERO state  -> y_zs, end state e
//   2. chunk start states by a scan of S_c = M S_(c-1) + e_c, M = A^LC: 6 Kogge-Stone steps inside each wave,
//      then the W wave totals are chained (W-1 2x2 products) and folded in with per-lane powers A^(LC (c+1));
//      all matrix powers are precomputed on the host in extended precision
//   3. y_lin[i] = y_zs[i] + (C A^i) . s0_chunk, then saturation blend / guard / clamp exactly as the reference.
// Bands remain sequential (the nonlinearity sits between them); the span lives in LDS for all 20 bands, so
// HBM sees one read and one write per sample.  One workgroup of W waves per channel: 64*W-way time
// parallelism per band, W*channels waves in flight (2 per SIMD at 256 streams, W = 4).
//
// The state guards of the reference (non-finite or >= 1e15 -> 0) cannot trip when the span input is finite and
// below kTpInputBound and the incoming state is below it too (the host proves state gain * bound < 1e15 per
// band before enabling this kernel); otherwise the span is run by the guarded sequential path below.
// Result differs from the sequential recurrence by rounding only (measured <= 3e-15 abs over 20 bands).

constexpr double kTpInputBound = 1.0e9;
constexpr int kTpWaves = kSvfTpWaves;        // waves per channel
constexpr int kTpChunks = 64 * kTpWaves;     // chunks (= threads) per span
constexpr int kTpLcMain = kSvfTpLc[0];       // samples per chunk, main spans (4096 samples)
constexpr int kTpLcTail = kSvfTpLc[1];       // samples per chunk, 512-sample remainder spans
constexpr int kTpStride = kTpLcMain + 2;     // LDS row stride in doubles: rows 16-byte aligned for b128 access, 36 dwords
                                             // apart so that 16 consecutive rows cover all 64 banks

// per (stream, band); one block per chunk length (kTpLcMain, kTpLcTail); must match host buildSvfTpTables()
struct TpLcTables {
    double Mk[6][4];     // A^(LC*2^k), row-major 2x2: in-wave scan steps
    double Mw[4];        // A^(LC*64): one whole wave of chunks
    double P[64][4];     // A^(LC*(c+1)): carries the wave's start state to the end of chunk c
    double G[16][2];     // C*A^i, i < LC
};
// matrix form of one 16-sample chunk for the MFMA path: T[m][k] = ht[15 + m - k] (zero-state response, lower triangular
// Toeplitz), e[:, k] = A^(15-k) B (end state of the chunk)
struct TpMfmaTables {
    double ht[32];
    double e[2][16];
};
struct TpBandTables { TpLcTables t[2]; TpMfmaTables mm; };

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
template <int N, int ORDER = 3>
__device__ __forceinline__ void tp_nonlinear_small(double (&y)[N], double c1)
{
    const double ca = c1 * (1.0 / 9.0), cb = 3.0 - c1 * (1.0 / 3.0);
    double den[N], r[N];
#pragma unroll
    for (int j = 0; j < N; ++j) den[j] = fma(y[j], y[j], 3.0);
#pragma unroll
#if defined(CPQ_ABLV) && (CPQ_ABLV & 1)
    for (int j = 0; j < N; ++j) r[j] = den[j] * 0.3;
#else
    for (int j = 0; j < N; ++j) r[j] = __builtin_amdgcn_rcp(den[j]);
#endif
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double e = fma(-den[j], r[j], 1.0);
        r[j] = (ORDER == 2) ? fma(e, r[j], r[j]) : fma(fma(e, e, e), r[j], r[j]);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) y[j] *= fma(cb, r[j], ca);
}

// zero-state (or continuing) recurrence of one band over N samples held in registers: v[j] <- y_lin[j]
// KIND: 0 = SVF, packed stereo arithmetic (FMA), 1 = SVF scalar arithmetic (Left/Right modes), 2 = DF-II-T biquad
// of the OutputFilter (coefficients b0 b1 b2 a1 a2 in a1 a2 a3 m0 m1; state w1 w2 in ic1 ic2)
template <int KIND, int N>
__device__ __forceinline__ void tp_recur(double (&v)[N], double& ic1, double& ic2, double a1, double a2, double a3,
                                         double m0, double m1, double m2)
{
#pragma unroll
    for (int j = 0; j < N; ++j) {
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
            v[j] = fma(m1, v1, v0);
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
}

// per-workgroup LDS copy of the channel's per-band constants (one chunk length at a time)
struct alignas(16) TpLds {
    double cf[kBands][6];        // a1 a2 a3 m0 m1 m2
    double M[kBands][28];        // Mk[6][4], Mw[4]
    double G[kBands][32];        // G[16][2]
};

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

// ONE workgroup barrier per band: the wave totals go through wtot[parity] (the caller flips the parity per band, so a
// wave that is already in the next band writes the other half while slow waves still read this one) and the span's
// end state goes to sNext while every wave reads the start state from sCur (the caller swaps the two per span).
// Plate != nullptr: the per-lane powers are loaded from there right where they are used (register-tight callers: the
// loads then wait on L2 behind the other waves of the SIMD) and pw is ignored.
template <int NTHREADS = kTpChunks>
__device__ __forceinline__ void tp_scan(double ic1, double ic2, double& s0x, double& s0y, const double* Mall, int b,
                                        const TpLanePowers& pw, double* wtot, const double* sCur, double* sNext, int tid,
                                        const double* __restrict__ Plate = nullptr)
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
    if (tid == (NTHREADS ? NTHREADS : (int)blockDim.x) - 1) { sNext[2 * b] = sx; sNext[2 * b + 1] = sy; }     // end of the span
}

// guarded sequential fallback for one band over the span held in LDS (one thread): the reference recurrence
// with every guard, used when the span input or the carried state is outside the proven-safe range.
template <int KIND>
__device__ void tp_band_guarded(double* buf, int lc, const double* cf, double sat, double* sState, int nChunks = kTpChunks)
{
    const double a1 = cf[0], a2 = cf[1], a3 = cf[2], m0 = cf[3], m1 = cf[4], m2 = cf[5];
    const double oneMinusSat = 1.0 - sat;
    double ic1 = sState[0], ic2 = sState[1];
    for (int c = 0; c < nChunks; ++c)
        for (int i = 0; i < lc; ++i) {
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

// U samples of a chunk row <-> registers; rows are 16-byte aligned (kTpStride even), so pairs move as one b128 access
template <int U>
__device__ __forceinline__ void tp_row_load(double (&v)[U], const double* p)
{
    if (U % 2 == 0) {
#pragma unroll
        for (int j = 0; j < U; j += 2) { const double2 t = *reinterpret_cast<const double2*>(p + j); v[j] = t.x; v[j + (U > 1)] = t.y; }
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = p[j];
    }
}
template <int U>
__device__ __forceinline__ void tp_row_store(const double (&v)[U], double* p)
{
    if (U % 2 == 0) {
#pragma unroll
        for (int j = 0; j < U; j += 2) *reinterpret_cast<double2*>(p + j) = make_double2(v[j], v[j + (U > 1)]);
    } else {
#pragma unroll
        for (int j = 0; j < U; ++j) p[j] = v[j];
    }
}

// One span (kTpChunks chunks of LC samples) through all active bands.  Every thread owns one chunk = one LDS
// row, so between bands no barrier is needed for the sample data; per band the only exchange is the 4 wave
// totals of the state scan.  The output stage of band b and the zero-state run of the next active band are
// fused over the same registers (the next band consumes what the output stage just produced).
template <int LC, bool SAT>
__device__ __forceinline__ void tp_span(const double* in, double* out, double* buf, double* wtot, double*& sState,
                                        double*& sNext, int* sFlag, const TpLds* L, int tid, const int* __restrict__ fl,
                                        const TpBandTables* __restrict__ tb, double sat, double gain)
{
    constexpr int LCI = (LC == kTpLcMain) ? 0 : 1;
    constexpr int U = (LC < 8) ? LC : 8;
    const double oneMinusSat = 1.0 - sat;
    const bool smallOk = (sat >= 0.0) && (sat <= 1.0);      // |out| <= |y| on the small-signal output stage
    const double smallC1 = 9.0 - 8.0 * sat;
    // span -> LDS, coalesced; sample j of the span sits at row j / LC, column j % LC
    bool bad = false;
#pragma unroll 4
    for (int it = 0; it < LC; ++it) {
        const int j = it * kTpChunks + tid;
        const double x = in[j];
        bad |= !(fabs(x) < kTpInputBound);
        buf[(j / LC) * kTpStride + (j % LC)] = x;
    }
    if (tid < kBands * 2) bad |= !(fabs(sState[tid]) < kTpInputBound);
    if (tid == 0) *sFlag = 0;
    __syncthreads();
    if (__any(bad) && (tid & 63) == 0) atomicOr(sFlag, 1);
    __syncthreads();
    const bool unsafe = (*sFlag != 0);

    if (unsafe) {
        for (int b = 0; b < kBands; ++b) {
            const int flag = fl[b];
            if (!(flag & 1)) continue;
            if (tid == 0) {
                if (flag & 4)      tp_band_guarded<2>(buf, LC, L->cf[b], sat, sState + 2 * b);
                else if (flag & 2) tp_band_guarded<1>(buf, LC, L->cf[b], sat, sState + 2 * b);
                else               tp_band_guarded<0>(buf, LC, L->cf[b], sat, sState + 2 * b);
            }
            __syncthreads();
        }
    } else {
        double* row = buf + tid * kTpStride;
        int par = 0;
        int b = 0;
        while (b < kBands && !(fl[b] & 1)) ++b;              // first active band (uniform)
        if (b < kBands) {
            // zero-state run of the first active band on the raw input
            double ic1 = 0.0, ic2 = 0.0;
            {
                const double a1 = L->cf[b][0], a2 = L->cf[b][1], a3 = L->cf[b][2];
                const double m0 = L->cf[b][3], m1 = L->cf[b][4], m2 = L->cf[b][5];
                const int kind = (fl[b] >> 1) & 3;      // 0 SVF stereo, 1 SVF scalar, 2 DF-II-T
#pragma unroll 1
                for (int i0 = 0; i0 < LC; i0 += U) {
                    double v[U];
                    tp_row_load<U>(v, row + i0);
                    if (kind == 2)      tp_recur<2, U>(v, ic1, ic2, a1, a2, a3, m0, m1, m2);
                    else if (kind == 1) tp_recur<1, U>(v, ic1, ic2, a1, a2, a3, m0, m1, m2);
                    else                tp_recur<0, U>(v, ic1, ic2, a1, a2, a3, m0, m1, m2);
                    tp_row_store<U>(v, row + i0);
                }
            }
            while (b < kBands) {
                int nb = b + 1;
                while (nb < kBands && !(fl[nb] & 1)) ++nb;     // next active band (uniform)
                double s0x, s0y;
                tp_scan(ic1, ic2, s0x, s0y, &L->M[0][0], b, tp_load_powers(&tb[b].t[LCI].P[0][0], tid & 63),
                        wtot + par * 2 * kTpWaves, sState, sNext, tid);
                par ^= 1;
                // response table row of band b through a VGPR base: reads below are base + immediate offset
                uint32_t gOff = (uint32_t)b * (uint32_t)sizeof(L->G[0]);
                asm volatile("" : "+v"(gOff));
                const double* Gb = reinterpret_cast<const double*>(reinterpret_cast<const char*>(&L->G[0][0]) + gOff);
                const int kindB = (fl[b] >> 1) & 3;
                const bool hasNext = nb < kBands;
                const int kindN = hasNext ? ((fl[nb] >> 1) & 3) : 0;
                double a1 = 0, a2 = 0, a3 = 0, m0 = 1, m1 = 0, m2 = 0;
                if (hasNext) {
                    a1 = L->cf[nb][0]; a2 = L->cf[nb][1]; a3 = L->cf[nb][2];
                    m0 = L->cf[nb][3]; m1 = L->cf[nb][4]; m2 = L->cf[nb][5];
                }
                ic1 = 0.0; ic2 = 0.0;
#pragma unroll 1
                for (int i0 = 0; i0 < LC; i0 += U) {
                    double v[U];
                    tp_row_load<U>(v, row + i0);
#pragma unroll
                    for (int j = 0; j < U; ++j)
                        v[j] = fma(Gb[2 * (i0 + j) + 1], s0y, fma(Gb[2 * (i0 + j)], s0x, v[j]));
                    if (kindB != 2) {     // kindB == 2 (OutputFilter biquad): linear section, no output stage
                        double big = fabs(v[0]);
#pragma unroll
                        for (int j = 1; j < U; ++j) big = fmax(big, fabs(v[j]));
                        if (smallOk && __all(big < 4.5)) {          // wave-uniform: the usual case at audio levels
                            if (SAT) tp_nonlinear_small<U>(v, smallC1);
                        } else if (kindB == 1) tp_nonlinear<true, SAT, U, false>(v, sat, oneMinusSat);
                        else                   tp_nonlinear<false, SAT, U, false>(v, sat, oneMinusSat);
                    }
                    if (hasNext) {
                        if (kindN == 2)      tp_recur<2, U>(v, ic1, ic2, a1, a2, a3, m0, m1, m2);
                        else if (kindN == 1) tp_recur<1, U>(v, ic1, ic2, a1, a2, a3, m0, m1, m2);
                        else                 tp_recur<0, U>(v, ic1, ic2, a1, a2, a3, m0, m1, m2);
                    }
                    tp_row_store<U>(v, row + i0);
                }
                b = nb;
            }
        }
        __syncthreads();
        { double* t = sState; sState = sNext; sNext = t; }      // the span's end states become the next span's start states
    }
#pragma unroll 4
    for (int it = 0; it < LC; ++it) {
        const int j = it * kTpChunks + tid;
        out[j] = buf[(j / LC) * kTpStride + (j % LC)] * gain;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// MFMA form of the main path (chunk length 16).  Inside one band everything between the input and the output stage
// is linear, and over a chunk of 16 samples it is a small dense product shared by all chunks of the span:
//     y_lin (16 x chunks) = [ T | G ] (16 x 18)  .  [ x ; s0 ] (18 x chunks),     e (2 x chunks) = E (2 x 16) . x
// with T the lower-triangular Toeplitz matrix of the band's zero-state impulse response, G = C A^i the state
// response and s0 the chunk start states from the scan of the end states e.  That product runs on the matrix cores
// (v_mfma_f64_16x16x4_f64: 5 per tile of 16 chunks), which are idle otherwise and issue beside the VALU.
// Register layout = the instruction's own: lane (m = lane & 15, g = lane >> 4) holds of tile tau (16 chunks) the
// samples g + 4 j (j = register) of chunk 16 tau + m.  The D registers of one band ARE the B operands of the next
// (k-step s = register s), so the span stays in registers across the 20 bands; the output stage is element-wise.
typedef double v4d __attribute__((ext_vector_type(4)));

struct alignas(16) TpLdsM {
    double cf[kBands][6];        // a1 a2 a3 m0 m1 m2 (guarded fallback)
    double M[kBands][28];        // Mk[6][4], Mw[4] (scan)
    double Gq[kBands][16][4];    // (C A^i)_x, (C A^i)_y, 0, 0: A-operand rows of the state response
    double ht[kBands][32];
    double e[kBands][2][16];
};

// LDS traffic between lanes of ONE wave: the hardware keeps a wave's DS operations in order; this keeps the compiler
// from moving them across each other
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// tables of the matrix form -> LDS (nThreads = threads of the workgroup); the caller synchronises
__device__ __forceinline__ void tp_load_tables_m(TpLdsM* L, const double* __restrict__ cf,
                                                 const TpBandTables* __restrict__ tb, int tid, int nThreads)
{
    for (int i = tid; i < kBands * 6; i += nThreads) L->cf[i / 6][i % 6] = cf[i];
    for (int i = tid; i < kBands * 28; i += nThreads) {
        const int b = i / 28, q = i % 28;
        L->M[b][q] = (q < 24) ? tb[b].t[0].Mk[q / 4][q % 4] : tb[b].t[0].Mw[q - 24];
    }
    for (int i = tid; i < kBands * 64; i += nThreads) {
        const int b = i / 64, r = (i % 64) / 4, q = i % 4;
        L->Gq[b][r][q] = (q < 2) ? tb[b].t[0].G[r][q] : 0.0;
    }
    for (int i = tid; i < kBands * 32; i += nThreads) {
        L->ht[i / 32][i % 32] = tb[i / 32].mm.ht[i % 32];
        L->e[i / 32][(i % 32) / 16][i % 16] = tb[i / 32].mm.e[(i % 32) / 16][i % 16];
    }
}

// The band loop of the matrix form: x = the wave's 64 chunks in the MFMA layout (in and out); red / s0q = the wave's LDS
// scratch (256 double2 / 256 doubles); NTHREADS = threads of the workgroup (64 per wave of the span; 0 = blockDim.x).
// wtot: two parities of [2 * waves] wave totals (one workgroup barrier per band, see tp_scan).  A barrier-free variant
// (totals published with per-band flags, waves polling only their predecessors) measured slower: 0.59 vs 0.58 ms.
template <bool SAT, int NTHREADS>
__device__ __forceinline__ void tp_bands_mfma(v4d (&x)[4], double2* red, double* s0q, double* wtot, const double* sState,
                                              double* sNext, const TpLdsM* L, int tid, const int* __restrict__ fl,
                                              const TpBandTables* __restrict__ tb, double sat)
{
    const int lane = tid & 63;
    const int m = lane & 15, g = lane >> 4;
    const double oneMinusSat = 1.0 - sat;
    const bool smallOk = (sat >= 0.0) && (sat <= 1.0);
    const double smallC1 = 9.0 - 8.0 * sat;
        int par = 0;
        for (int b = 0; b < kBands; ++b) {
            const int flag = fl[b];
            if (!(flag & 1)) continue;                    // uniform
            const int kind = (flag >> 1) & 3;
            // tables of the band
            const TpLanePowers pw = {};                       // register-tight: tp_scan loads the per-lane powers where it uses them
            double a[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) a[s4] = L->ht[b][15 + m - 4 * s4 - g];
            // (1) end state of every chunk's zero-state run: e = E x, partial over this lane's four samples per tile ...
            double e0[4], e1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { e0[j] = L->e[b][0][g + 4 * j]; e1[j] = L->e[b][1][g + 4 * j]; }
#pragma unroll
            for (int tau = 0; tau < 4; ++tau) {
                double px = e0[0] * x[tau][0], py = e1[0] * x[tau][0];
#pragma unroll
                for (int j = 1; j < 4; ++j) { px = fma(e0[j], x[tau][j], px); py = fma(e1[j], x[tau][j], py); }
                red[(tau * 4 + g) * 16 + m] = make_double2(px, py);
            }
            // (2) the zero-state part of the product, T x, does not wait for the start states: its 16 MFMAs go to the
            // matrix pipe now and run beside the reduction and the scan below (x is dead from here: acc takes its place)
#if !(defined(CPQ_ABL) && (CPQ_ABL & 4))
#ifndef CPQ_TX_16X16
            {
                // T is lower-triangular Toeplitz: of its sixteen 4 x 4 blocks only the ten on and below the diagonal are
                // non-zero, and block (i, j) depends on i - j alone.  v_mfma_f64_4x4x4_4b_f64 multiplies one such block
                // into four batches of four chunks; its operand layout (B[k][n] at lane 16 k + n, D[i][n] at lane
                // 16 i + n, A[i][k] at lane 16 k + 4 batch + i: tools/ubench/mfma_f64_4x4x4.hip) is register s of the
                // 16x16x4 layout = block row s, so the two instructions mix freely.  10 small MFMAs (~17-20 cycles each)
                // instead of 4 large ones (64 cycles each) per tile.
                double a4[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) a4[d] = L->ht[b][15 + 4 * d + (m & 3) - g];
                double dacc[4][4];
                // block-column major: consecutive MFMAs write different accumulators
#ifdef CPQ_TX_TILE_MAJOR
#pragma unroll
                for (int tau = 0; tau < 4; ++tau)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = j; i < 4; ++i)
                            dacc[tau][i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[i - j], x[tau][j], j == 0 ? 0.0 : dacc[tau][i], 0, 0, 0);
#else
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tau = 0; tau < 4; ++tau)
#pragma unroll
                        for (int i = j; i < 4; ++i)
                            dacc[tau][i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[i - j], x[tau][j], j == 0 ? 0.0 : dacc[tau][i], 0, 0, 0);
#endif
#pragma unroll
                for (int tau = 0; tau < 4; ++tau) x[tau] = v4d{ dacc[tau][0], dacc[tau][1], dacc[tau][2], dacc[tau][3] };
            }
#else
            {
                v4d acc[4];
#pragma unroll
                for (int tau = 0; tau < 4; ++tau) acc[tau] = v4d{ 0.0, 0.0, 0.0, 0.0 };
                // k-step major: consecutive MFMAs belong to different tiles, so none waits for its own accumulator
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int tau = 0; tau < 4; ++tau)
                        acc[tau] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], x[tau][s4], acc[tau], 0, 0, 0);
#pragma unroll
                for (int tau = 0; tau < 4; ++tau) x[tau] = acc[tau];
            }
#endif
#endif
            wave_lds_sync();
            // ... the partial end states summed over the four lane groups; lane l ends up with chunk l of the wave
            double ic1, ic2;
#if defined(CPQ_ABL) && (CPQ_ABL & 1)
            ic1 = e0[0]; ic2 = e1[1];
#else
            {
                const double2 p0 = red[(g * 4 + 0) * 16 + m], p1 = red[(g * 4 + 1) * 16 + m];
                const double2 p2 = red[(g * 4 + 2) * 16 + m], p3 = red[(g * 4 + 3) * 16 + m];
                ic1 = (p0.x + p1.x) + (p2.x + p3.x);
                ic2 = (p0.y + p1.y) + (p2.y + p3.y);
            }
#endif
            // (3) chunk start states
            double s0x, s0y;
#if defined(CPQ_ABL) && (CPQ_ABL & 2)
            s0x = ic1 * e0[1]; s0y = ic2 * e1[0];
#else
            tp_scan<NTHREADS>(ic1, ic2, s0x, s0y, &L->M[0][0], b, pw, wtot + par * 2 * ((NTHREADS ? NTHREADS : (int)blockDim.x) / 64), sState, sNext, tid,
                              &tb[b].t[0].P[0][0]);
            par ^= 1;
#endif
            // (4) the state response G s0 completes the product (k-step 4: rows 16 + g of [x ; s0], staged through the
            // wave's LDS scratch), tile by tile, followed by (5) the element-wise output stage of that tile
            *reinterpret_cast<double2*>(s0q + lane * 4) = make_double2(s0x, s0y);
            *reinterpret_cast<double2*>(s0q + lane * 4 + 2) = make_double2(0.0, 0.0);
            wave_lds_sync();
            const double ag = L->Gq[b][m][g];
            double sb[4];
#pragma unroll
            for (int tau = 0; tau < 4; ++tau) sb[tau] = s0q[(tau * 16 + m) * 4 + g];
#pragma unroll
            for (int tau = 0; tau < 4; ++tau) x[tau] = __builtin_amdgcn_mfma_f64_16x16x4f64(ag, sb[tau], x[tau], 0, 0, 0);
#if defined(CPQ_ABL) && (CPQ_ABL & 8)
            if (false) {
#else
            if (kind != 2) {          // kind 2 (OutputFilter biquad): linear section, no output stage
#endif
#pragma unroll
                for (int tau = 0; tau < 4; ++tau) {
                    double v[4] = { x[tau][0], x[tau][1], x[tau][2], x[tau][3] };
                    // four compares with the |.| modifier (a NaN fails them and takes the guarded code below)
                    const bool small = (int)(fabs(v[0]) < 4.5) & (int)(fabs(v[1]) < 4.5) & (int)(fabs(v[2]) < 4.5) & (int)(fabs(v[3]) < 4.5);
                    if (smallOk && __all(small)) {
                        if (SAT) tp_nonlinear_small<4>(v, smallC1);
                    } else if (kind == 1) tp_nonlinear<true, SAT, 4, false>(v, sat, oneMinusSat);
                    else                  tp_nonlinear<false, SAT, 4, false>(v, sat, oneMinusSat);
                    x[tau] = v4d{ v[0], v[1], v[2], v[3] };
                }
            }
        }
}

__device__ __forceinline__ void tp_load_tables(TpLds* L, const double* __restrict__ cf,
                                               const TpBandTables* __restrict__ tb, int lci, int tid)
{
    for (int i = tid; i < kBands * 6; i += kTpChunks) L->cf[i / 6][i % 6] = cf[i];
    for (int i = tid; i < kBands * 28; i += kTpChunks) {
        const int b = i / 28, q = i % 28;
        L->M[b][q] = (q < 24) ? tb[b].t[lci].Mk[q / 4][q % 4] : tb[b].t[lci].Mw[q - 24];
    }
    for (int i = tid; i < kBands * 32; i += kTpChunks) L->G[i / 32][i % 32] = tb[i / 32].t[lci].G[(i % 32) / 2][i % 2];
    __syncthreads();
}

// Spans of 512 samples (256 chunks of 2) in VALU form: the last block of a call with an odd block count, and calls of
// one block.
__global__ __launch_bounds__(kTpChunks) void k_svf_cascade_tp(const double* in, double* out, int64_t chStride,
                                                              int nSamples, const double* __restrict__ coef,
                                                              const int* __restrict__ flags,
                                                              const double* __restrict__ satGain,
                                                              double* __restrict__ state,
                                                              const TpBandTables* __restrict__ tables)
{
    __shared__ double buf[kTpChunks * kTpStride];
    __shared__ TpLds L;
    __shared__ double sStateA[kBands * 2], sStateB[kBands * 2];     // start / end states of the current span (swapped per span)
    __shared__ double wtot[2 * 2 * kTpWaves];                        // wave totals, two parities
    double* sState = sStateA;
    double* sNext = sStateB;
    __shared__ int sFlag;
    const int tid = threadIdx.x;
    const int c = blockIdx.x;
    const double* cf = coef + (int64_t)c * kBands * 6;
    const int* fl = flags + c * kBands;
    const TpBandTables* tb = tables + (int64_t)(c >> 1) * kBands;     // tables are per stream
    const double sat = satGain[c * 2], gain = satGain[c * 2 + 1];
    if (tid < kBands * 2) { sStateA[tid] = state[(int64_t)c * kBands * 2 + tid]; sStateB[tid] = sStateA[tid]; }

    const double* src = in + (int64_t)c * chStride;
    double* dst = out + (int64_t)c * chStride;
    tp_load_tables(&L, cf, tb, 1, tid);
    for (int done = 0; nSamples - done >= kTpChunks * kTpLcTail; done += kTpChunks * kTpLcTail) {
        if (sat > 0.0) tp_span<kTpLcTail, true>(src + done, dst + done, buf, wtot, sState, sNext, &sFlag, &L, tid, fl, tb, sat, gain);
        else           tp_span<kTpLcTail, false>(src + done, dst + done, buf, wtot, sState, sNext, &sFlag, &L, tid, fl, tb, sat, gain);
    }
    __syncthreads();
    if (tid < kBands * 2) state[(int64_t)c * kBands * 2 + tid] = sState[tid];
}

// Span I/O of the matrix-form kernels.  The MFMA layout wants lane (m, g) to hold samples g + 4 j (register j) of chunk m;
// loaded as such, one instruction touches an 8-byte word in 16 different 128-byte lines and every 32-byte sector is fetched
// (and written) in pieces: PMC traffic 1.7x the algorithmic bytes.  Instead lane (m, g) moves the whole sector, samples
// 4 g ... 4 g + 3, with two 16-byte accesses and a 4 x 4 transpose across the four 16-lane rows of the wave puts them in
// place (v_permlane32_swap / v_permlane16_swap, tools/ubench/permlane_transpose.hip: 8 VALU instructions per tile).
__device__ __forceinline__ void tp_swap32(double& a, double& b)      // rows 2,3 of a <-> rows 0,1 of b
{
    const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double(h[0], l[0]);
    b = __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ void tp_swap16(double& a, double& b)      // rows 1,3 of a <-> rows 0,2 of b
{
    const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double(h[0], l[0]);
    b = __hiloint2double(h[1], l[1]);
}
// s[i] at row g = element (g, i)  ->  s[j] at row g = element (j, g); its own inverse
__device__ __forceinline__ void tp_transpose4(double (&s)[4])
{
    tp_swap32(s[0], s[2]);
    tp_swap32(s[1], s[3]);
    tp_swap16(s[0], s[1]);
    tp_swap16(s[2], s[3]);
}
// chunk = the 16 samples of chunk m (128-byte aligned); g = lane >> 4
__device__ __forceinline__ v4d tp_tile_load(const double* chunk, int g)
{
    // streaming accesses: a span is read once and written once per call; the L2 is left to the per-stream scan tables
    typedef double v2 __attribute__((ext_vector_type(2)));
    const v2 a = __builtin_nontemporal_load(reinterpret_cast<const v2*>(chunk + 4 * g));
    const v2 b = __builtin_nontemporal_load(reinterpret_cast<const v2*>(chunk + 4 * g + 2));
    double s[4] = { a.x, a.y, b.x, b.y };
    tp_transpose4(s);
    return v4d{ s[0], s[1], s[2], s[3] };
}
// Stores go through the wave's LDS scratch instead (tile = the 16 chunks x 16 samples = 2 KB at `tile`, buf = 16 rows of
// kTpStride doubles): every store instruction then writes 1 KB of whole 128-byte lines.  Partial-line stores made the L2
// fetch the rest of each line from memory first (PMC: reads 2.3x, writes 1.4x the algorithmic bytes).
__device__ __forceinline__ void tp_tile_store(double* tile, double* buf, int lane, v4d x, double gain)
{
    const int m = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) buf[m * kTpStride + g + 4 * j] = x[j] * gain;
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = lane + 64 * i;                          // pair of samples: chunk p >> 3, samples 2 (p & 7), + 1
        typedef double v2 __attribute__((ext_vector_type(2)));
        const double2 v = *reinterpret_cast<const double2*>(buf + (p >> 3) * kTpStride + 2 * (p & 7));
        __builtin_nontemporal_store(v2{ v.x, v.y }, reinterpret_cast<v2*>(tile + 2 * p));
    }
    wave_lds_sync();                                          // the next tile reuses buf
}

// ---------------------------------------------------------------------------------------------------------
// Eight waves per channel: spans of 8192 samples (512 chunks of 16).  The band loop is a dependent chain (end states ->
// reduction -> scan -> product -> output stage) that two waves per SIMD do not hide; with twice the waves per channel
// four waves share a SIMD.  No LDS staging of the span (it would not fit twice per CU beside the tables): every lane
// loads and stores its 16 samples straight in the MFMA layout (4 x 8 B per 128-byte line and instruction, the four
// registers of a tile cover the line).  Spans with non-finite / out-of-range input go through the guarded sequential
// code in two 4096-sample halves staged in the scratch area.  Handles whole 8192-sample spans only; the launcher runs
// k_svf_cascade_tp on what is left.
// WAVES = 16 (spans of 16384 samples, one workgroup per CU) is for engines with fewer channels than the chip has CUs:
// at 128 channels the eight-wave kernel leaves half the CUs idle and the other half at two waves per SIMD.
constexpr int kTp8Threads = 512;
constexpr int kTp8Span = kTp8Threads * 16;

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 4) void k_svf_cascade_tp8(const double* in, double* out, int64_t chStride,
                                                                int nSpans, const double* __restrict__ coef,
                                                                const int* __restrict__ flags,
                                                                const double* __restrict__ satGain,
                                                                double* __restrict__ state,
                                                                const TpBandTables* __restrict__ tables)
{
    constexpr int kThreads = WAVES * 64, kSpan = kThreads * 16;
    constexpr int kScratchDoubles = WAVES * (512 + 256);     // per wave: red (256 double2) + s0q (256 doubles); 8 waves: 48 KB
    static_assert(kScratchDoubles >= 256 * kTpStride, "the guarded path stages 4096 samples in the scratch area");
    __shared__ __align__(16) double scratch[kScratchDoubles];
    __shared__ TpLdsM LM;
    __shared__ double sStateA[kBands * 2], sStateB[kBands * 2];
    __shared__ double wtot[2 * 2 * WAVES];
    __shared__ int sFlag;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int c = blockIdx.x;
    const double* cf = coef + (int64_t)c * kBands * 6;
    const int* fl = flags + c * kBands;
    const TpBandTables* tb = tables + (int64_t)(c >> 1) * kBands;
    const double sat = satGain[c * 2], gain = satGain[c * 2 + 1];
    double* sState = sStateA;
    double* sNext = sStateB;
    if (tid < kBands * 2) { sStateA[tid] = state[(int64_t)c * kBands * 2 + tid]; sStateB[tid] = sStateA[tid]; }
    tp_load_tables_m(&LM, cf, tb, tid, kThreads);
    __syncthreads();

    for (int sp = 0; sp < nSpans; ++sp) {
        const double* src = in + (int64_t)c * chStride + (int64_t)sp * kSpan;
        double* dst = out + (int64_t)c * chStride + (int64_t)sp * kSpan;
        v4d x[4];
        bool bad = false;
#pragma unroll
        for (int tau = 0; tau < 4; ++tau) {
            x[tau] = tp_tile_load(src + (wave * 64 + tau * 16 + m) * 16, g);
#pragma unroll
            for (int j = 0; j < 4; ++j) bad |= !(fabs(x[tau][j]) < kTpInputBound);
        }
        if (tid < kBands * 2) bad |= !(fabs(sState[tid]) < kTpInputBound);
        if (tid == 0) sFlag = 0;
        __syncthreads();
        if (__any(bad) && lane == 0) atomicOr(&sFlag, 1);
        __syncthreads();
        if (sFlag != 0) {
            // guarded path: pieces of 4096 samples through the one-thread reference recurrence, staged in the scratch
            // area as [chunk][sample]; in and out may alias, every sample of a piece is read before the piece is written
            for (int half = 0; half < kSpan / 4096; ++half) {
                for (int j = tid; j < 4096; j += kThreads)
                    scratch[(j / 16) * kTpStride + (j % 16)] = src[half * 4096 + j];
                __syncthreads();
                for (int b = 0; b < kBands; ++b) {
                    const int flag = fl[b];
                    if (!(flag & 1)) continue;
                    if (tid == 0) {
                        if (flag & 4)      tp_band_guarded<2>(scratch, 16, LM.cf[b], sat, sState + 2 * b);
                        else if (flag & 2) tp_band_guarded<1>(scratch, 16, LM.cf[b], sat, sState + 2 * b);
                        else               tp_band_guarded<0>(scratch, 16, LM.cf[b], sat, sState + 2 * b);
                    }
                    __syncthreads();
                }
                for (int j = tid; j < 4096; j += kThreads)
                    dst[half * 4096 + j] = scratch[(j / 16) * kTpStride + (j % 16)] * gain;
                __syncthreads();
            }
            continue;
        }
        double2* red = reinterpret_cast<double2*>(scratch) + wave * 256;
        double* s0q = scratch + WAVES * 512 + wave * 256;
        if (sat > 0.0) tp_bands_mfma<true, kThreads>(x, red, s0q, wtot, sState, sNext, &LM, tid, fl, tb, sat);
        else           tp_bands_mfma<false, kThreads>(x, red, s0q, wtot, sState, sNext, &LM, tid, fl, tb, sat);
#pragma unroll
        for (int tau = 0; tau < 4; ++tau)
            tp_tile_store(dst + (wave * 64 + tau * 16) * 16, reinterpret_cast<double*>(red), lane, x[tau], gain);
        __syncthreads();                      // the last thread's end states are in sNext
        { double* t = sState; sState = sNext; sNext = t; }
    }
    __syncthreads();
    if (tid < kBands * 2) state[(int64_t)c * kBands * 2 + tid] = sState[tid];
}

// ---------------------------------------------------------------------------------------------------------
// One to seven waves per channel, ONE span of waves x 1024 samples in the same matrix form: what a call leaves after
// its whole 8192-sample spans (and all of a call of 2 ... 15 blocks of 512).  The cost of a span is the latency of the
// 20-band chain, whatever its length, so the remainder is split over as many waves as it has 1024-sample pieces rather
// than walked through span by span; before, it ran as 4096-sample spans on four waves and then as 512-sample spans in
// VALU form (chunk length 2) at three times the cost per block.
constexpr int kTpwMaxWaves = 7;
constexpr int kTpwScratchDoubles = kTpwMaxWaves * (512 + 256);      // per wave: red (256 double2) + s0q (256 doubles)

__global__ __launch_bounds__(kTpwMaxWaves * 64, 4) void k_svf_cascade_tpw(const double* in, double* out, int64_t chStride,
                                                                       const double* __restrict__ coef,
                                                                       const int* __restrict__ flags,
                                                                       const double* __restrict__ satGain,
                                                                       double* __restrict__ state,
                                                                       const TpBandTables* __restrict__ tables)
{
    static_assert(kTpwScratchDoubles >= 256 * kTpStride, "the guarded path stages up to 4096 samples in the scratch area");
    __shared__ __align__(16) double scratch[kTpwScratchDoubles];
    __shared__ TpLdsM LM;
    __shared__ double sStateA[kBands * 2], sStateB[kBands * 2];
    __shared__ double wtot[2 * 2 * kTpwMaxWaves];
    __shared__ int sFlag;
    const int tid = threadIdx.x, nThreads = blockDim.x;
    const int nWaves = nThreads >> 6;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int c = blockIdx.x;
    const double* cf = coef + (int64_t)c * kBands * 6;
    const int* fl = flags + c * kBands;
    const TpBandTables* tb = tables + (int64_t)(c >> 1) * kBands;
    const double sat = satGain[c * 2], gain = satGain[c * 2 + 1];
    if (tid < kBands * 2) { sStateA[tid] = state[(int64_t)c * kBands * 2 + tid]; sStateB[tid] = sStateA[tid]; }
    tp_load_tables_m(&LM, cf, tb, tid, nThreads);
    if (tid == 0) sFlag = 0;
    __syncthreads();

    const double* src = in + (int64_t)c * chStride;
    double* dst = out + (int64_t)c * chStride;
    v4d x[4];
    bool bad = false;
#pragma unroll
    for (int tau = 0; tau < 4; ++tau) {
        x[tau] = tp_tile_load(src + (wave * 64 + tau * 16 + m) * 16, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) bad |= !(fabs(x[tau][j]) < kTpInputBound);
    }
    if (tid < kBands * 2) bad |= !(fabs(sStateA[tid]) < kTpInputBound);
    if (__any(bad) && lane == 0) atomicOr(&sFlag, 1);
    __syncthreads();
    if (sFlag != 0) {
        // guarded path: pieces of up to 4096 samples through the one-thread reference recurrence, staged in the scratch
        // area as [chunk][sample]; in and out may alias, every sample of a piece is read before the piece is written
        const int nSamples = nWaves * 1024;
        for (int base = 0; base < nSamples; base += 4096) {
            const int cnt = (nSamples - base < 4096) ? nSamples - base : 4096;
            for (int j = tid; j < cnt; j += nThreads) scratch[(j / 16) * kTpStride + (j % 16)] = src[base + j];
            __syncthreads();
            for (int b = 0; b < kBands; ++b) {
                const int flag = fl[b];
                if (!(flag & 1)) continue;
                if (tid == 0) {
                    if (flag & 4)      tp_band_guarded<2>(scratch, 16, LM.cf[b], sat, sStateA + 2 * b, cnt / 16);
                    else if (flag & 2) tp_band_guarded<1>(scratch, 16, LM.cf[b], sat, sStateA + 2 * b, cnt / 16);
                    else               tp_band_guarded<0>(scratch, 16, LM.cf[b], sat, sStateA + 2 * b, cnt / 16);
                }
                __syncthreads();
            }
            for (int j = tid; j < cnt; j += nThreads) dst[base + j] = scratch[(j / 16) * kTpStride + (j % 16)] * gain;
            __syncthreads();
        }
        if (tid < kBands * 2) state[(int64_t)c * kBands * 2 + tid] = sStateA[tid];
        return;
    }
    double2* red = reinterpret_cast<double2*>(scratch) + wave * 256;
    double* s0q = scratch + kTpwMaxWaves * 512 + wave * 256;
    if (sat > 0.0) tp_bands_mfma<true, 0>(x, red, s0q, wtot, sStateA, sStateB, &LM, tid, fl, tb, sat);
    else           tp_bands_mfma<false, 0>(x, red, s0q, wtot, sStateA, sStateB, &LM, tid, fl, tb, sat);
#pragma unroll
    for (int tau = 0; tau < 4; ++tau)
        tp_tile_store(dst + (wave * 64 + tau * 16) * 16, reinterpret_cast<double*>(red), lane, x[tau], gain);
    __syncthreads();                          // the span's end states are in sStateB
    if (tid < kBands * 2) state[(int64_t)c * kBands * 2 + tid] = sStateB[tid];
}

// ---------------------------------------------------------------------------------------------------------
// Vector form with the chunk in registers ("tpv"), waves pipelined along the time axis.
//
// fp64 MFMA and fp64 VALU share one datapath on gfx950 at the same rate (tools/ubench/coexec_f64.hip), so the dense form
// of a band's linear part -- T x on ten 4x4x4 block products, G s0 on a 16x16x4, E x as 32 FMAs and a cross-lane
// reduction: ~16 FMA slots per sample -- costs more issue slots than the recurrence it replaces (7 ... 10 operations per
// sample, end state included).  Here lane = one chunk of 16 consecutive samples held in 32 VGPRs through all bands, a
// WAVE = one piece of 1024 consecutive samples, and per band
//   (1) the chunk start states come from the wave-level scan (DPP) of the zero-state end states e = E x, which the
//       PREVIOUS band's pass accumulated from its outputs as it produced them (2 FMAs per sample), plus the state W at
//       the start of the piece,
//   (2) ONE pass over the 16 samples runs the reference recurrence from the true start state, applies the output stage
//       and feeds the next band's E x: no zero-state run, no state-response fix-up, no cross-lane traffic.
// The pieces of a channel are dealt round-robin to the NW = nGroups x (waves per workgroup) waves that work on it; piece m
// needs, band by band, the state at the end of piece m - 1 and nothing else, so the waves run as a software pipeline
// along the time axis, each a fraction of a band behind its predecessor, and hand the two doubles over through a slot
// per (wave, band): in LDS inside a workgroup, in global memory between the workgroups of a channel (nGroups > 1:
// engines with fewer channels than the chip has room for workgroups).  There is NO workgroup barrier in the loops: the
// waves of a SIMD drift into different phases of the band (scan / recurrence / output stage) and fill each other's
// dependency stalls, which a barrier per band prevented (the barrier version ran the SIMDs 77 % busy).
// A slot is rewritten by its owner only after the owner consumed the state of the piece before its next one, which
// (transitively, through the ring of waves) the consumer of the old value has produced: no back-pressure needed.
// Band coefficients are wave-uniform SGPR operands, the E rows and the scan powers come from LDS.  Piece I/O: coalesced
// 16-byte accesses, transposed to chunk-per-lane through the wave's own padded LDS buffer in four quarters; the wave's
// next piece is requested into L2 while the bands run.
// A piece whose input or start state is outside the range for which the host proved the reference's guards idle is not
// processed here: its wave records the piece and the piece's start states (SvfRedo entry + the channel's state row) and
// from then on only drains the ring, publishing NaN states -- poison -- so that every later piece of the channel is left
// alone too; the bit-faithful sequential kernel, launched behind this one, redoes the channel from the recorded piece
// on (it returns at once for every other channel).  None of that code is inside the band loop.

struct TpvChainSlot { double sx, sy; unsigned long long ticket; unsigned long long pad; };
constexpr int kTpvMaxGroups = 16;       // workgroups per channel at most (slots per channel in the global ring)

__device__ __forceinline__ void tpv_chain_put(TpvChainSlot* s, double sx, double sy, unsigned long long ticket)
{
    __hip_atomic_store(&s->sx, sx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&s->sy, sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&s->ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tpv_chain_get(const TpvChainSlot* s, double& sx, double& sy, unsigned long long ticket)
{
    while (__hip_atomic_load(&s->ticket, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != ticket) __builtin_amdgcn_s_sleep(2);
    sx = __hip_atomic_load(&s->sx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sy = __hip_atomic_load(&s->sy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef CPQ_TPV_PEAK
#define CPQ_TPV_PEAK 1
#endif
#ifndef CPQ_TPV_ASM
#define CPQ_TPV_ASM 1
#endif
#ifndef CPQ_TPV_OPQLANE
#define CPQ_TPV_OPQLANE 1
#endif
#ifndef CPQ_TPV_RAREUNROLL
#define CPQ_TPV_RAREUNROLL 1
#endif
#ifndef CPQ_TPV_U
#define CPQ_TPV_U 4
#endif
#ifndef CPQ_TPV_ORDER
#define CPQ_TPV_ORDER 2
#endif
constexpr int kTpvWaves = 8;            // waves per workgroup at most
constexpr int kTpvQStride = 6;          // doubles per row of the quarter-chunk transposition buffer: 48 B, conflict-free b128 rows

// what the kernel keeps in LDS (75 KB: two workgroups per CU)
struct TpvShared {
    alignas(16) double scratch[kTpvWaves * 64 * kTpvQStride];  // per wave 64 rows x kTpvQStride (piece I/O)
    alignas(16) double P[kBands][64][4];                       // A^(16 (n + 1)): per-lane powers of the scan
    alignas(16) double M[kBands][20];                          // A^(16 2^k), k < 4, and A^(16 64) (scan)
    alignas(16) double E[kBands][16][2];                       // (A^(15-k) B)_x, _y: end state of a chunk's zero-state run
    alignas(16) double slotS[kTpvWaves][kBands][2];            // hand-over: state at the end of the wave's latest piece ...
    unsigned slotSeq[kTpvWaves][kBands];                       // ... and which piece that was (index + 1)
    alignas(16) double ownW[kTpvWaves][kBands][2];             // start states the wave consumed for its current piece
};

// The lane id, recomputed where it is used (two instructions) and hidden from loop-invariant code motion: addresses that
// depend on it are formed inside the loops instead of being hoisted into VGPRs that stay occupied (or get spilled and
// reloaded) across the band loop.
__device__ __forceinline__ int tpv_lane()
{
#if CPQ_TPV_OPQLANE
    int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
#else
    return threadIdx.x & 63;
#endif
}

// where a wave gets the start states of its piece from and where it leaves the end states
struct TpvLink {
    int wPrev;                      // LDS slot (wave) to read, or -1: the global slot gPrev
    int wOwn;                       // LDS slot to write, or -1: nobody in this workgroup reads it
    const TpvChainSlot* gPrev;      // global slot of the workgroup before (wave 0 of a chained workgroup)
    TpvChainSlot* gOwn;             // global slot of this workgroup (its last wave, chained) or null
    double* st;                     // the call's start states (piece 0) and end states (last piece) [band][2]
    unsigned long long ticketBase;  // launch serial << 32
};
__device__ __forceinline__ double tpv_uniform(double v)
{
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// start state of piece m, band b (wave-uniform, in SGPRs)
__device__ __forceinline__ void tpv_get_state(TpvShared& sh, const TpvLink& L, int m, int b, double& wx, double& wy)
{
    double sx = 0.0, sy = 0.0;
    if (m == 0) {
        sx = L.st[2 * b];
        sy = L.st[2 * b + 1];
    } else if (L.wPrev < 0) {
        if (tpv_lane() == 0) tpv_chain_get(L.gPrev + b, sx, sy, L.ticketBase | (unsigned)m);
    } else {
        // all lanes poll the same word (broadcast read); the data was written before the sequence number
        while (__hip_atomic_load(&sh.slotSeq[L.wPrev][b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != (unsigned)m)
            __builtin_amdgcn_s_sleep(1);
        const double2 v = *reinterpret_cast<const double2*>(&sh.slotS[L.wPrev][b][0]);      // ordered behind the acquire above
        sx = v.x;
        sy = v.y;
    }
    wx = tpv_uniform(sx);
    wy = tpv_uniform(sy);
}
// end state of piece m, band b (wave-uniform values)
__device__ __forceinline__ void tpv_put_state(TpvShared& sh, const TpvLink& L, int m, int nPieces, int b, double sx, double sy)
{
    if (tpv_lane() != 0) return;
    if (L.wOwn >= 0) {
        *reinterpret_cast<double2*>(&sh.slotS[L.wOwn][b][0]) = make_double2(sx, sy);
        __hip_atomic_store(&sh.slotSeq[L.wOwn][b], (unsigned)(m + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (L.gOwn) tpv_chain_put(L.gOwn + b, sx, sy, L.ticketBase | (unsigned)(m + 1));
    if (m == nPieces - 1) { L.st[2 * b] = sx; L.st[2 * b + 1] = sy; }
}

__device__ __forceinline__ double tpv_readlane(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// Chunk start states of one band inside a wave.  (e0, e1) = the lane's zero-state chunk end state; in-wave inclusive scan
// of S_c = M S_(c-1) + e_c from a zero start (4 shift-and-combine steps inside each row of 16 lanes with the powers
// A^(16 2^k), then lane 15 of a row carries into the next row and lane 31 into the upper half with the per-lane powers);
// the wave total + the piece's start state W give the piece's end state, which is published at once (the next wave waits
// for it); s0 = the scan shifted by one lane + A^(16 lane) W.  Returns false when W is outside the proven range or
// poison (W is then in wx, wy and nothing has been published).
__device__ __forceinline__ bool tpv_wave_scan(double e0, double e1, double& s0x, double& s0y, TpvShared& sh, int b,
                                              const TpvLink& L, int m, int nPieces, double& wx, double& wy)
{
    const int lane = tpv_lane();
    const double2* Mb = reinterpret_cast<const double2*>(&sh.M[b][0]);
    const double2* Pl = reinterpret_cast<const double2*>(&sh.P[b][0][0]);
    double sx = e0, sy = e1;
#define CPQ_ROW_STEP(k)                                                                                       \
    {                                                                                                         \
        const double2 k01 = Mb[(k) * 2], k23 = Mb[(k) * 2 + 1];                                               \
        const double px = dpp_f64<kDppRowShr + (1 << (k)), 0xF>(sx);                                          \
        const double py = dpp_f64<kDppRowShr + (1 << (k)), 0xF>(sy);                                          \
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
        const double2 pa01 = Pl[(lane & 15) * 2], pa23 = Pl[(lane & 15) * 2 + 1];
        const double px = dpp_f64<kDppRowBcast15, 0xA>(sx);
        const double py = dpp_f64<kDppRowBcast15, 0xA>(sy);
        const double nx = fma(pa01.y, py, fma(pa01.x, px, sx));
        const double ny = fma(pa23.y, py, fma(pa23.x, px, sy));
        sx = nx;
        sy = ny;
    }
    {   // rows 2 and 3 <- total of the lower half
        const double2 pb01 = Pl[(lane & 31) * 2], pb23 = Pl[(lane & 31) * 2 + 1];
        const double px = dpp_f64<kDppRowBcast31, 0xC>(sx);
        const double py = dpp_f64<kDppRowBcast31, 0xC>(sy);
        const double nx = fma(pb01.y, py, fma(pb01.x, px, sx));
        const double ny = fma(pb23.y, py, fma(pb23.x, px, sy));
        sx = nx;
        sy = ny;
    }
    const double tx = tpv_readlane(sx, 63), ty = tpv_readlane(sy, 63);      // zero-start end state of the piece
    tpv_get_state(sh, L, m, b, wx, wy);
    if (!(fabs(wx) < kTpInputBound) || !(fabs(wy) < kTpInputBound)) return false;
    if (lane == 0) *reinterpret_cast<double2*>(&sh.ownW[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)][b][0]) = make_double2(wx, wy);
    {
        const double2 mw01 = Mb[8], mw23 = Mb[9];
        tpv_put_state(sh, L, m, nPieces, b, fma(mw01.y, wy, fma(mw01.x, wx, tx)), fma(mw23.y, wy, fma(mw23.x, wx, ty)));
    }
    const double2 pc01 = Pl[lane * 2], pc23 = Pl[lane * 2 + 1];
    sx = fma(pc01.y, wy, fma(pc01.x, wx, sx));
    sy = fma(pc23.y, wy, fma(pc23.x, wx, sy));
    s0x = dpp_f64<kDppWaveShr1, 0xF>(sx);
    s0y = dpp_f64<kDppWaveShr1, 0xF>(sy);
    if (lane == 0) { s0x = wx; s0y = wy; }
    return true;
}

// The recurrence of one SVF band over the lane's 16 samples as ONE block of hand-placed instructions: every sample's result
// lands in the register that held its input (tied operands), so the band loop carries x in fixed registers whatever the
// band's class -- with the C++ forms the compiler renamed the sixteen values per class and paid for it in copies and
// scratch traffic at every merge.  Coefficients are SGPR operands (one per instruction: the constant-bus limit of gfx9).
// Peak form (m0 == 1, m2 == 0; seven operations per sample):
//   v3 = x - ic2;  t = a2 v3;  ic2 = k3 v3 + ic2;  t = a1 ic1 + t (= v1);  ic2 = k2 ic1 + ic2;  ic1 = 2 t - ic1;  x = m1 t + x
// with k2 = 2 a2, k3 = 2 a3.
#define CPQ_SVF_PEAK_STEP(X)                                   \
    "v_add_f64 %[v3], " X ", -%[ic2]\n\t"                      \
    "v_mul_f64 %[t], %[a2], %[v3]\n\t"                         \
    "v_fma_f64 %[ic2], %[k3], %[v3], %[ic2]\n\t"               \
    "v_fma_f64 %[t], %[a1], %[ic1], %[t]\n\t"                  \
    "v_fma_f64 %[ic2], %[k2], %[ic1], %[ic2]\n\t"              \
    "v_fma_f64 %[ic1], 2.0, %[t], -%[ic1]\n\t"                 \
    "v_fma_f64 " X ", %[m1], %[t], " X "\n\t"
__device__ __forceinline__ void tpv_recur_peak8(double& x0, double& x1, double& x2, double& x3, double& x4, double& x5, double& x6,
                                                double& x7, double& ic1, double& ic2, double a1, double a2, double k2, double k3,
                                                double m1)
{
    double v3, t;
    asm volatile(CPQ_SVF_PEAK_STEP("%[x0]") CPQ_SVF_PEAK_STEP("%[x1]") CPQ_SVF_PEAK_STEP("%[x2]") CPQ_SVF_PEAK_STEP("%[x3]")
                 CPQ_SVF_PEAK_STEP("%[x4]") CPQ_SVF_PEAK_STEP("%[x5]") CPQ_SVF_PEAK_STEP("%[x6]") CPQ_SVF_PEAK_STEP("%[x7]")
                 : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5), [x6] "+v"(x6),
                   [x7] "+v"(x7), [ic1] "+v"(ic1), [ic2] "+v"(ic2), [v3] "=&v"(v3), [t] "=&v"(t)
                 : [a1] "s"(a1), [a2] "s"(a2), [k2] "s"(k2), [k3] "s"(k3), [m1] "s"(m1));
}
#undef CPQ_SVF_PEAK_STEP
// General form (ten operations per sample), the reference's own operation order:
//   v3 = x - ic2;  t = a2 v3;  t = a1 ic1 + t (= v1);  u = a3 v3 + ic2;  u = a2 ic1 + u (= v2);  ic1 = 2 t - ic1;
//   ic2 = 2 u - ic2;  u = m2 u;  u = m1 t + u;  x = m0 x + u
#define CPQ_SVF_GEN_STEP(X)                                    \
    "v_add_f64 %[v3], " X ", -%[ic2]\n\t"                      \
    "v_mul_f64 %[t], %[a2], %[v3]\n\t"                         \
    "v_fma_f64 %[u], %[a3], %[v3], %[ic2]\n\t"                 \
    "v_fma_f64 %[t], %[a1], %[ic1], %[t]\n\t"                  \
    "v_fma_f64 %[u], %[a2], %[ic1], %[u]\n\t"                  \
    "v_fma_f64 %[ic1], 2.0, %[t], -%[ic1]\n\t"                 \
    "v_fma_f64 %[ic2], 2.0, %[u], -%[ic2]\n\t"                 \
    "v_mul_f64 %[u], %[m2], %[u]\n\t"                          \
    "v_fma_f64 %[u], %[m1], %[t], %[u]\n\t"                    \
    "v_fma_f64 " X ", %[m0], " X ", %[u]\n\t"
__device__ __forceinline__ void tpv_recur_gen8(double& x0, double& x1, double& x2, double& x3, double& x4, double& x5, double& x6,
                                               double& x7, double& ic1, double& ic2, double a1, double a2, double a3, double m0,
                                               double m1, double m2)
{
    double v3, t, u;
    asm volatile(CPQ_SVF_GEN_STEP("%[x0]") CPQ_SVF_GEN_STEP("%[x1]") CPQ_SVF_GEN_STEP("%[x2]") CPQ_SVF_GEN_STEP("%[x3]")
                 CPQ_SVF_GEN_STEP("%[x4]") CPQ_SVF_GEN_STEP("%[x5]") CPQ_SVF_GEN_STEP("%[x6]") CPQ_SVF_GEN_STEP("%[x7]")
                 : [x0] "+v"(x0), [x1] "+v"(x1), [x2] "+v"(x2), [x3] "+v"(x3), [x4] "+v"(x4), [x5] "+v"(x5), [x6] "+v"(x6),
                   [x7] "+v"(x7), [ic1] "+v"(ic1), [ic2] "+v"(ic2), [v3] "=&v"(v3), [t] "=&v"(t), [u] "=&v"(u)
                 : [a1] "s"(a1), [a2] "s"(a2), [a3] "s"(a3), [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2));
}
#undef CPQ_SVF_GEN_STEP

// wave-uniform value the optimiser cannot see through: a constant wrapped in it is materialised where it is used instead
// of being hoisted out of the band loop into registers that stay occupied for the whole kernel
__device__ __forceinline__ double tpv_opaque(double v)
{
    asm volatile("" : "+s"(v));
    return v;
}

// Output stage of N samples when some sample of the wave is at or above the fastTanh clip threshold (rare).  Below the
// threshold: the small-signal form.  At or above it the tanh argument is clamped (packed-stereo flavour: th = the Pade at
// 4.5) or the result is +-1 (scalar flavour), so the blend is y (1 - sat) +- sat th_c; then the +-100 clamp, which can
// only act there.  Same values as tp_nonlinear<MONO, SAT, N, false> at rounding level.
template <bool SAT, int N>
__device__ __forceinline__ void tpv_nonlinear_rare(double (&y)[N], bool mono, const double* __restrict__ satPtr, double smallC1)
{
    const double lim = tpv_opaque(100.0);
    if (SAT) {
        const double sat = tpv_opaque(*satPtr);
        const double thc = tpv_opaque(mono ? 1.0 : (4.5 * (27.0 + 20.25)) / (27.0 + 9.0 * 20.25));
        const double sThc = sat * thc, oneMinusSat = 1.0 - sat, clip = tpv_opaque(4.5);
        double sm[N];
#pragma unroll
        for (int j = 0; j < N; ++j) sm[j] = y[j];
        tp_nonlinear_small<N, 3>(sm, smallC1);
#pragma unroll
        for (int j = 0; j < N; ++j) {
            const double lin = fma(y[j], oneMinusSat, copysign(sThc, y[j]));
            y[j] = (fabs(y[j]) < clip) ? sm[j] : lin;
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) y[j] = fmin(fmax(y[j], -lim), lim);
}

// One band over the lane's 16 samples from its true start state.  KIND 0 / 3: SVF band, general / with m0 == 1 and m2 == 0
// (every peaking band: seven operations per sample in the recurrence instead of ten), in the packed-stereo FMA arithmetic
// for both arithmetic flavours of the reference -- the time-parallel evaluation is rounding-level anyway; `mono` selects
// the scalar fastTanh's hard +-1 on the rare large-signal output stage; KIND 2: DF-II-T biquad of the OutputFilter.
// The recurrence runs over all 16 samples first: the output stage does not feed back into the state, so it follows as
// independent evaluations behind ONE wave-uniform test for the small-signal form.  En = the NEXT band's E rows in LDS.
template <bool SAT>
__device__ __forceinline__ void tpv_pass(double (&x)[16], double ic1, double ic2, const double* __restrict__ cfb, int KIND, bool mono,
                                         const double* En, double& e0o, double& e1o,
                                         const double* __restrict__ satPtr, bool smallOk, double smallC1)
{
    {
        // only the recurrence differs between the band classes (wave-uniform branch); what follows is common code
        const double a1 = cfb[0], a2 = cfb[1], a3 = cfb[2], m0 = cfb[3], m1 = cfb[4], m2 = cfb[5];
#if CPQ_TPV_ASM
        if (KIND == 3) {
            const double k2 = tpv_uniform(2.0 * a2), k3 = tpv_uniform(2.0 * a3);     // doubled: wave-uniform, back to SGPRs
            tpv_recur_peak8(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], ic1, ic2, a1, a2, k2, k3, m1);
            tpv_recur_peak8(x[8], x[9], x[10], x[11], x[12], x[13], x[14], x[15], ic1, ic2, a1, a2, k2, k3, m1);
        } else if (KIND == 0) {
            tpv_recur_gen8(x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7], ic1, ic2, a1, a2, a3, m0, m1, m2);
            tpv_recur_gen8(x[8], x[9], x[10], x[11], x[12], x[13], x[14], x[15], ic1, ic2, a1, a2, a3, m0, m1, m2);
        } else {
#else
        if (KIND == 3) {
            tp_recur<3, 16>(x, ic1, ic2, tpv_uniform(2.0 * a2), tpv_uniform(2.0 * a3), 0.0, a2, m1, a1);
        } else if (KIND == 0) {
            tp_recur<0, 16>(x, ic1, ic2, a1, a2, a3, m0, m1, m2);
        } else {
#endif
            tp_recur<2, 16>(x, ic1, ic2, a1, a2, a3, m0, m1, m2);
        }
    }
    double e0 = 0.0, e1 = 0.0;
    bool done = false;
    if (KIND != 2) {          // kind 2: linear section, no output stage
        int small = 1;
#pragma unroll
        for (int j = 0; j < 16; ++j) small &= (int)(fabs(x[j]) < 4.5);      // a NaN fails and takes the general code
        if (smallOk && __all(small)) {
            // CPQ_TPV_U at a time, kept apart in the schedule: sixteen evaluations in flight at once do not fit the registers
#pragma unroll
            for (int h = 0; h < 16 / CPQ_TPV_U; ++h) {
                double v[CPQ_TPV_U];
#pragma unroll
                for (int j = 0; j < CPQ_TPV_U; ++j) v[j] = x[CPQ_TPV_U * h + j];
                if (SAT) tp_nonlinear_small<CPQ_TPV_U, CPQ_TPV_ORDER>(v, smallC1);
#pragma unroll
                for (int j = 0; j < CPQ_TPV_U; ++j) {
                    const double2 ee = *reinterpret_cast<const double2*>(En + 2 * (CPQ_TPV_U * h + j));
                    e0 = fma(ee.x, v[j], e0);
                    e1 = fma(ee.y, v[j], e1);
                    x[CPQ_TPV_U * h + j] = v[j];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            done = true;
        } else {
            // rare: a sample at or above the fastTanh clip threshold somewhere in the wave (straight-line code: a loop over
            // the groups would carry x through loop registers of its own and cost the hot path copies at every merge)
#if CPQ_TPV_RAREUNROLL
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                double v[4] = { x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3] };
                tpv_nonlinear_rare<SAT, 4>(v, mono, satPtr, smallC1);
#pragma unroll
                for (int j = 0; j < 4; ++j) x[4 * h + j] = v[j];
                __builtin_amdgcn_sched_barrier(0);
            }
#else
#pragma unroll 1
            for (int h = 0; h < 4; ++h) {
                double v[4] = { x[0], x[1], x[2], x[3] };
                tpv_nonlinear_rare<SAT, 4>(v, mono, satPtr, smallC1);
#pragma unroll
                for (int j = 0; j < 12; ++j) x[j] = x[j + 4];
#pragma unroll
                for (int j = 0; j < 4; ++j) x[12 + j] = v[j];
            }
#endif
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

// piece <-> registers: 8 coalesced 16-byte accesses per lane, transposed through the wave's LDS buffer in four quarters
// (samples 4 h ... 4 h + 3 of every chunk: 64 rows of kTpvQStride).  Lane l of access k holds samples 2 (l & 7), + 1 of
// chunk 8 k + (l >> 3): the lanes with ((l >> 1) & 3) == h belong to quarter h.
__device__ __forceinline__ void tpv_piece_load(const double* src, double* buf, double (&x)[16])
{
    const int lane = tpv_lane();
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
__device__ __forceinline__ void tpv_piece_store(double* dst, double* buf, const double (&x)[16], double gain)
{
    const int lane = tpv_lane();
    typedef double v2 __attribute__((ext_vector_type(2)));
    v2 t[8];
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

// The active bands over the piece held in x: per band the scan of the chunk end states, then the pass in the form of the
// band's class (wave-uniform branch: DF-II-T section / SVF band with output v0 + m1 v1 / SVF band).  e0 / e1: E x of the
// first band on entry.  Returns the band at which a start state was out of the proven range or poison (it is in wx, wy
// then), or -1.
template <bool SAT>
__device__ __forceinline__ int tpv_band_loop(double (&x)[16], double e0, double e1, unsigned mask, unsigned dfMask, unsigned peakMask,
                                             unsigned monoMask, TpvShared& sh, const double* __restrict__ cf, const double* __restrict__ satPtr, double sat,
                                             const TpvLink& L, int m, int nPieces, double& wx, double& wy)
{
    const bool smallOk = (sat >= 0.0) && (sat <= 1.0);
    const double smallC1 = 9.0 - 8.0 * sat;
#pragma unroll 1
    while (mask) {
        const int b = __builtin_ctz(mask);
        mask &= mask - 1;
        const int nb = mask ? __builtin_ctz(mask) : b;     // last band: its E x result is not used
        double s0x, s0y;
        if (!tpv_wave_scan(e0, e1, s0x, s0y, sh, b, L, m, nPieces, wx, wy)) return b;
        const double* En = &sh.E[nb][0][0];
        const bool mono = (monoMask >> b) & 1;
        const int kind = ((dfMask >> b) & 1) ? 2 : (((peakMask >> b) & 1) ? 3 : 0);
        tpv_pass<SAT>(x, s0x, s0y, cf + b * 6, kind, mono, En, e0, e1, satPtr, smallOk, smallC1);
    }
    return -1;
}

__global__ __launch_bounds__(kTpvWaves * 64, 4) void k_svf_cascade_tpv(const double* in, double* out, int64_t chStride,
                                                                     int nPieces, int nGroups, const double* __restrict__ coef,
                                                                     const int* __restrict__ flags,
                                                                     const double* __restrict__ satGain,
                                                                     double* __restrict__ state,
                                                                     const TpBandTables* __restrict__ tables,
                                                                     TpvChainSlot* chain, unsigned long long ticketBase,
                                                                     SvfRedo* redo)
{
    __shared__ TpvShared sh;
    const int tid = threadIdx.x, nThreads = blockDim.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nW = nThreads >> 6;
    const int c = blockIdx.x / nGroups, grp = blockIdx.x - c * nGroups;
    const double* __restrict__ cf = coef + (int64_t)c * kBands * 6;
    const TpBandTables* __restrict__ tb = tables + (int64_t)(c >> 1) * kBands;
    const double sat = satGain[c * 2], gain = satGain[c * 2 + 1];
    const double* inCh = in + (int64_t)c * chStride;
    double* outCh = out + (int64_t)c * chStride;
    unsigned activeMask = 0, dfMask = 0, monoMask = 0, peakMask = 0;
    // active bands; DF-II-T sections; SVF bands with the scalar fastTanh; SVF bands whose output is v0 + m1 v1
    for (int b = 0; b < kBands; ++b) {
        const int f = flags[c * kBands + b];
        const int kind = (f >> 1) & 3;
        activeMask |= (unsigned)(f & 1) << b;
        dfMask |= (unsigned)(kind == 2) << b;
        monoMask |= (unsigned)(kind == 1) << b;
        peakMask |= (unsigned)(CPQ_TPV_PEAK && kind != 2 && cf[b * 6 + 3] == 1.0 && cf[b * 6 + 5] == 0.0) << b;
    }
    for (int i = tid; i < kBands * 20; i += nThreads) {
        const int b = i / 20, q = i % 20;
        sh.M[b][q] = (q < 16) ? tb[b].t[0].Mk[q / 4][q % 4] : tb[b].t[0].Mw[q - 16];
    }
    for (int i = tid; i < kBands * 32; i += nThreads) {
        const int b = i / 32, k = (i % 32) >> 1, r = i & 1;
        sh.E[b][k][r] = tb[b].mm.e[r][k];
    }
    for (int i = tid; i < kBands * 128; i += nThreads) {       // 16-byte pieces of P[64][4]
        const int b = i >> 7, q = i & 127;
        reinterpret_cast<double2*>(&sh.P[b][0][0])[q] = reinterpret_cast<const double2*>(&tb[b].t[0].P[0][0])[q];
    }
    for (int i = tid; i < kTpvWaves * kBands; i += nThreads) (&sh.slotSeq[0][0])[i] = 0;
    __syncthreads();                                           // the only workgroup barrier

    const int NW = nGroups * nW;                               // waves on this channel
    const int u = grp * nW + wave;                             // this wave's place in the ring
    TpvLink L;
    const bool chainedIn = (wave == 0) && (nGroups > 1), lastOfChained = (wave == nW - 1) && (nGroups > 1);
    TpvChainSlot* chainCh = chain + (int64_t)c * kTpvMaxGroups * kBands;
    L.wPrev = chainedIn ? -1 : ((wave == 0) ? nW - 1 : wave - 1);
    L.wOwn = lastOfChained ? -1 : wave;
    L.gPrev = chainCh + (int64_t)((grp + nGroups - 1) % nGroups) * kBands;
    L.gOwn = lastOfChained ? chainCh + (int64_t)grp * kBands : nullptr;
    L.st = state + (int64_t)c * kBands * 2;
    L.ticketBase = ticketBase;
    double* buf = sh.scratch + wave * 64 * kTpvQStride;

    double x[16];
    double wx = 0.0, wy = 0.0;
    int m = u, bCold = -1;
    bool haveW = false;
#pragma unroll 1
    for (; m < nPieces; m += NW) {
        const double* src = inCh + (int64_t)m * 1024;
        tpv_piece_load(src, buf, x);
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 16; ++j) bad |= !(fabs(x[j]) < kTpInputBound);
        if (__any(bad)) { bCold = 0; break; }
        if (activeMask) {
            double e0 = 0.0, e1 = 0.0;
            {
                const double* E = &sh.E[__builtin_ctz(activeMask)][0][0];     // the first band's E x on the raw input
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const double2 ee = *reinterpret_cast<const double2*>(E + 2 * j);
                    e0 = fma(ee.x, x[j], e0);
                    e1 = fma(ee.y, x[j], e1);
                }
            }
            // this wave's next piece: one 4-byte load per 128-byte line pulls it into L2 while the bands run
            if (m + NW < nPieces) (void)*reinterpret_cast<const volatile int*>(src + (int64_t)NW * 1024 + tpv_lane() * 16);
            const int bs = (sat > 0.0) ? tpv_band_loop<true>(x, e0, e1, activeMask, dfMask, peakMask, monoMask, sh, cf, satGain + c * 2, sat, L, m, nPieces, wx, wy)
                                       : tpv_band_loop<false>(x, e0, e1, activeMask, dfMask, peakMask, monoMask, sh, cf, satGain + c * 2, sat, L, m, nPieces, wx, wy);
            if (bs >= 0) { bCold = bs; haveW = true; }
            if (bCold >= 0) break;
        }
        tpv_piece_store(outCh + (int64_t)m * 1024, buf, x, gain);
    }
    if (bCold < 0) return;

    // ---- this wave found piece m out of range at band bCold (haveW: by the start state in wx, wy; else by its input)
    // or received poison there.  Detector (no poison received): record the piece and its start states -- the ones of the
    // bands before bCold are in sh.ownW, the others arrive now.  Then drain: consume and poison every remaining slot.
    bool first = !(haveW && (wx != wx));
#pragma unroll 1
    for (; m < nPieces; m += NW) {
#pragma unroll 1
        for (unsigned bm = activeMask; bm; bm &= bm - 1) {
            const int b = __builtin_ctz(bm);
            if (b < bCold) {
                if (first && lane == 0) { L.st[2 * b] = sh.ownW[wave][b][0]; L.st[2 * b + 1] = sh.ownW[wave][b][1]; }
                continue;                                      // consumed and published before the piece went bad
            }
            if (!haveW) tpv_get_state(sh, L, m, b, wx, wy);
            haveW = false;
            if (wx != wx) first = false;                       // poison: an earlier piece is the one to redo from
            if (first && lane == 0) { L.st[2 * b] = wx; L.st[2 * b + 1] = wy; }
            const double nan = __builtin_nan("");
            // (never the call's end state: tpv_put_state writes that for m == nPieces - 1 only on the fast path)
            tpv_put_state(sh, L, m, nPieces + 1, b, nan, nan);
        }
        if (first && lane == 0) {
            redo[c].piece = m;
            __hip_atomic_store(&redo[c].ticket, ticketBase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        first = false;
        bCold = 0;
    }
}

}  // namespace

void launch_svf_cascade(hipStream_t stream, const double* in, double* out, int64_t chStride, int nCh, int nSamples,
                        const double* coef, const int* flags, const double* satGain, double* state, bool streamPairs)
{
    if (streamPairs)
        hipLaunchKernelGGL(k_svf_cascade<2>, dim3((nCh + 1) / 2), dim3(64), 0, stream, in, out, chStride, nCh, nSamples,
                           coef, flags, satGain, state, (const SvfRedo*)nullptr, 0ull);
    else
        hipLaunchKernelGGL(k_svf_cascade<3>, dim3((nCh + 2) / 3), dim3(64), 0, stream, in, out, chStride, nCh, nSamples,
                           coef, flags, satGain, state, (const SvfRedo*)nullptr, 0ull);
}

}  // namespace cpq

namespace cpq {
size_t svf_chain_bytes(int nCh)
{
    return (size_t)nCh * kTpvMaxGroups * kBands * sizeof(TpvChainSlot);
}
size_t svf_redo_bytes(int nCh) { return (size_t)nCh * sizeof(SvfRedo); }

void launch_svf_cascade_tp(hipStream_t stream, const double* in, double* out, int64_t chStride, int nCh, int nSamples,
                           const double* coef, const int* flags, const double* satGain, double* state,
                           const void* tables, void* redo, unsigned long long* ticket, void* chain)
{
    static_assert(sizeof(TpBandTables) == kSvfTpTableDoubles * sizeof(double), "host/device table layout");
    // whole 1024-sample pieces on the wave-pipelined kernel (+ its fix-up pass), a last block of 512 on the
    // chunk-length-2 kernel
    const TpBandTables* tb = reinterpret_cast<const TpBandTables*>(tables);
    int done = 0;
    static int nCu = 0;
    if (nCu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        nCu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const int nPieces = nSamples / 1024;
    if (nPieces > 0) {
        const int nW = nPieces < kTpvWaves ? nPieces : kTpvWaves;
        // two workgroups per CU (LDS); several workgroups per channel when the channels alone do not fill the chip: every
        // workgroup must be resident at once (they wait for each other), so channels x groups stays within 2 x CUs
        int nGroups = 1;
        if (chain && nW == kTpvWaves) {
            nGroups = (2 * nCu) / nCh;
            const int useful = (nPieces + kTpvWaves - 1) / kTpvWaves;
            if (nGroups > useful) nGroups = useful;
            if (nGroups > kTpvMaxGroups) nGroups = kTpvMaxGroups;
            if (nGroups < 2) nGroups = 1;
        }
        const unsigned long long tk = (++*ticket) << 32;
        hipLaunchKernelGGL(k_svf_cascade_tpv, dim3(nCh * nGroups), dim3(64 * nW), 0, stream, in, out, chStride, nPieces, nGroups,
                           coef, flags, satGain, state, tb, reinterpret_cast<TpvChainSlot*>(chain), tk,
                           reinterpret_cast<SvfRedo*>(redo));
        done = nPieces * 1024;
        // channels that met input or states outside the proven range: redone from the recorded piece on by the sequential
        // kernel (one wave per channel; returns at once everywhere else)
        hipLaunchKernelGGL(k_svf_cascade<1>, dim3(nCh), dim3(64), 0, stream, in, out, chStride, nCh, done, coef, flags, satGain,
                           state, reinterpret_cast<const SvfRedo*>(redo), tk);
    }
    if (nSamples > done)
        hipLaunchKernelGGL(k_svf_cascade_tp, dim3(nCh), dim3(kTpChunks), 0, stream, in + done, out + done, chStride,
                           nSamples - done, coef, flags, satGain, state, tb);
}
}  // namespace cpq
