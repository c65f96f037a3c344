// svf_kernels.hip -- 20-band TPT state-variable-filter cascade for gfx950.
//
// Replaces processBandStereo / processBand (src/eqprocessor/EQProcessor.Processing.cpp:191-276, :128-186)
// as driven by EQProcessor::process(block, params, cache) in its serial structure (:1231-1253) followed by
// the steady total gain (:1262-1274).
//
// The recurrence is serial in time per (channel, band) and, because every band output passes through the
// fastTanh saturation blend, serial across bands too.  The only parallelism is channel x band with the bands
// skewed in time: lane = (channel, band); at step s band b processes sample s-b and hands its output to
// band b+1 of the same channel through a one-lane wave shift.  One wave carries 3 channels x 20 bands.
// Samples enter and leave through LDS in 64-sample coalesced chunks.
//
// Arithmetic follows the reference operation for operation (same FMA sites, IEEE division, same guards),
// so with identical coefficients the output is expected to be bit-identical to the SSE2+FMA path.
// This file is compiled with -ffp-contract=off: fused operations appear only where written as fma().
#include "kernels.hpp"

namespace cpq {

namespace {

constexpr int kChPerWave = 3;

// sanitizeFiniteInRangeV(v, 0, 1e15): non-finite or |v| >= 1e15 -> 0  (Processing.cpp:90-101)
__device__ __forceinline__ double sanitize(double v)
{
    const bool ok = ((v - v) == 0.0) && (fabs(v) < 1.0e15);
    return ok ? v : 0.0;
}

// in and out may alias (in-place processing like the reference): no __restrict__ on them.
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
    const double oneMinusSat = 1.0 - sat;

    double ylast = 0.0;
    const int nChunks = nSamples / 64;      // nSamples is a multiple of the 64-sample minimum block

    for (int chunk = 0; chunk <= nChunks; ++chunk) {
        // stage the next 64 input samples of the wave's channels (last iteration only drains the skew)
        if (chunk < nChunks) {
#pragma unroll
            for (int q = 0; q < kChPerWave; ++q)
                if (c0 + q < nCh) xin[q][lane] = in[(int64_t)(c0 + q) * chStride + (int64_t)chunk * 64 + lane];
        }
        __syncthreads();
        const int steps = (chunk < nChunks) ? 64 : (kBands - 1);
        for (int i = 0; i < steps; ++i) {
            const int n = chunk * 64 + i - band;                  // sample this lane handles at this step
            const double fromPrev = __shfl_up(ylast, 1);
            const double v0 = (band == 0) ? xin[chl < kChPerWave ? chl : 0][i & 63] : fromPrev;
            if (live && n >= 0 && n < nSamples) {
                double y = v0;
                if (active) {
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
                ylast = y;
                if (band == kBands - 1) yout[chl][n & 127] = y * gain;
            }
        }
        __syncthreads();
        // block chunk-1 is complete once this chunk's steps ran (band 19 lags 19 steps)
        if (chunk >= 1) {
#pragma unroll
            for (int q = 0; q < kChPerWave; ++q)
                if (c0 + q < nCh)
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

}  // namespace

void launch_svf_cascade(hipStream_t stream, const double* in, double* out, int64_t chStride, int nCh, int nSamples,
                        const double* coef, const int* flags, const double* satGain, double* state)
{
    const int grid = (nCh + kChPerWave - 1) / kChPerWave;
    hipLaunchKernelGGL(k_svf_cascade, dim3(grid), dim3(64), 0, stream, in, out, chStride, nCh, nSamples, coef, flags,
                       satGain, state);
}

}  // namespace cpq
