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
template <int N>
__device__ __forceinline__ void tp_nonlinear_small(double (&y)[N], double c1)
{
    const double ca = c1 * (1.0 / 9.0), cb = 3.0 - c1 * (1.0 / 3.0);
    double den[N], r[N];
#pragma unroll
    for (int j = 0; j < N; ++j) den[j] = fma(y[j], y[j], 3.0);
#pragma unroll
    for (int j = 0; j < N; ++j) r[j] = __builtin_amdgcn_rcp(den[j]);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double e = fma(-den[j], r[j], 1.0);
        r[j] = fma(fma(e, e, e), r[j], r[j]);
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

}  // namespace cpq

namespace cpq {
void launch_svf_cascade_tp(hipStream_t stream, const double* in, double* out, int64_t chStride, int nCh, int nSamples,
                           const double* coef, const int* flags, const double* satGain, double* state,
                           const void* tables)
{
    static_assert(sizeof(TpBandTables) == kSvfTpTableDoubles * sizeof(double), "host/device table layout");
    // whole 8192-sample spans on the eight-wave kernel, what is left as one span of 1 ... 7 waves x 1024 samples, and a
    // last block of 512 on the chunk-length-2 path of the four-wave kernel
    const TpBandTables* tb = reinterpret_cast<const TpBandTables*>(tables);
    int done = 0;
    // fewer channels than CUs: sixteen waves per channel on the whole 16384-sample spans
    static int nCu = 0;
    if (nCu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        nCu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    if (nCh <= nCu && nSamples >= 2 * kTp8Span) {
        const int nSpans16 = nSamples / (2 * kTp8Span);
        hipLaunchKernelGGL(k_svf_cascade_tp8<16>, dim3(nCh), dim3(2 * kTp8Threads), 0, stream, in, out, chStride, nSpans16, coef,
                           flags, satGain, state, tb);
        done = nSpans16 * 2 * kTp8Span;
    }
    const int nSpans8 = (nSamples - done) / kTp8Span;
    if (nSpans8 > 0)
        hipLaunchKernelGGL(k_svf_cascade_tp8<8>, dim3(nCh), dim3(kTp8Threads), 0, stream, in + done, out + done, chStride, nSpans8,
                           coef, flags, satGain, state, tb);
    done += nSpans8 * kTp8Span;
    const int nWaves = (nSamples - done) / 1024;
    if (nWaves > 0) {
        hipLaunchKernelGGL(k_svf_cascade_tpw, dim3(nCh), dim3(nWaves * 64), 0, stream, in + done, out + done, chStride, coef,
                           flags, satGain, state, tb);
        done += nWaves * 1024;
    }
    if (nSamples > done)
        hipLaunchKernelGGL(k_svf_cascade_tp, dim3(nCh), dim3(kTpChunks), 0, stream, in + done, out + done, chStride,
                           nSamples - done, coef, flags, satGain, state, tb);
}
}  // namespace cpq
