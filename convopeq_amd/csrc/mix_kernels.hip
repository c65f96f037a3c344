// mix_kernels.hip -- processor-level dry/wet stage of the convolver.
//
// Steady state of ConvolverProcessor::process (src/convolver/ConvolverProcessor.Runtime.cpp): the wet chunk is
// sanitised (:50-60, :722), the dry signal is read from a delay line (:549-567) and both are mixed with the
// equal-power gains (mixSteadySmall, :635-657: mul, mul, add -- not fused).  HBM-trivial elementwise work.
#include "kernels.hpp"

namespace cpq {

namespace {

// One range of a processor-level call.  The dry signal comes from the delay ring (the call's input has been written at
// absolute position pos0 .. before this kernel runs, so in-place calls need no copy): delayed by dNew[s]; during a
// latency cross-fade (:394-540) the first xLen[s] samples of the range are new * g + old * (1 - g) with the old delay
// dOld[s] and the LinearRamp values xGains[s][i] formed on the host (integer delays: the Catmull-Rom branch of the
// reference's reader is only reached by a fractional delay, which nothing in it produces).
__global__ __launch_bounds__(256) void k_convproc_mix(const double* wet, double* out, int64_t chStride, int nSamples,
                                                      const double* __restrict__ gains,
                                                      const double* __restrict__ ring, int ringMask, long long pos0,
                                                      const int* __restrict__ dNew, const int* __restrict__ dOld,
                                                      const int* __restrict__ xLen, const double* __restrict__ xGains,
                                                      int xCap, int wetValid, const int* __restrict__ rampLen,
                                                      const double* __restrict__ rampGains, int rampCap, int rampOff,
                                                      const int* __restrict__ wetOn)
{
    const int c = blockIdx.y;
    const int s = c >> 1;
    if (wetOn && !wetOn[s]) wetValid = 0;         // this stream's convolver rests (bypass / dry-only): delayed dry signal only
    const double wetG = gains[2 * s], dryG = gains[2 * s + 1];
    // mix smoothing (:591-607, mixSmoothingSmall :611-632): the first rampLen[s] samples of the call carry per-sample
    // gains (equalPowerSin of the LinearRamp's values, formed on the host); rampOff = first sample of this range
    const int nRamp = rampLen ? rampLen[s] - rampOff : 0;
    const double* rg = rampGains + ((int64_t)s * rampCap + rampOff) * 2;
    const int dn = dNew[s], dold = dOld[s];
    const int nx = xLen ? xLen[s] : 0;
    const double* xg = xGains + (int64_t)s * xCap;
    const double* w = wet + (int64_t)c * chStride;
    double* o = out + (int64_t)c * chStride;
    const double* r = ring + (int64_t)c * (ringMask + 1);
    const int stride = gridDim.x * blockDim.x;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < nSamples; n += stride) {
        double dry = r[(pos0 + n - dn) & ringMask];
        if (n < nx) {
            const double old = r[(pos0 + n - dold) & ringMask];
            const double g = xg[n];
            dry = (dry * g) + (old * (1.0 - g));
        }
        if (wetValid) {
            double wv = w[n];
            // isFiniteAndAbsBelowNoLibm(x, 1e300): false for NaN / Inf
            wv = (fabs(wv) < 1.0e300) ? wv : 0.0;
            if (n < nRamp) o[n] = (wv * rg[2 * n]) + (dry * rg[2 * n + 1]);
            else           o[n] = (wv * wetG) + (dry * dryG);
        } else {
            o[n] = dry;           // dry-only fast path (:573-585) and bypass (:123-186): plain copy of the delayed input
        }
    }
}

// a larger delay ring takes over the retained samples of the old one (absolute positions end - oldCap .. end - 1)
__global__ __launch_bounds__(256) void k_ring_regrow(const double* __restrict__ oldRing, int oldMask, double* __restrict__ newRing,
                                                     int newMask, long long end)
{
    const double* a = oldRing + (int64_t)blockIdx.y * (oldMask + 1);
    double* b = newRing + (int64_t)blockIdx.y * (newMask + 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= oldMask; i += gridDim.x * blockDim.x) {
        const long long pos = end - 1 - i;
        b[pos & newMask] = a[pos & oldMask];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Layered (time-varying) reference semantics.  When a tail layer's partition is longer than the IR that precedes it
// (partSize_L > outputDelaySamples_L, e.g. block 1024 with the default tail start), the reference's delay-line
// reader (delayLineReadAdd, src/MKLNonUniformConvolver.cpp:1653-1688) skips ahead after every tail block and then
// runs dry until the next one: the output is NOT one linear convolution.  The engine then convolves every layer
// segment separately (natural time, no offset) and replays the reader on those streams:
//   out[cB + j] = (x*h0)[cB + j] + sum_L g_L * ynat_L[start_L(c) + j]   when callback c reads, else no tail term.

struct TailState {            // device-resident, zeroed by reset
    long long cb;             // callbacks (blocks of B) processed since reset
    long long R[2];           // delayReadCursor of tail layers 1, 2
    long long g0;             // global index of the first sample of the call the schedule was last made for
};

// one thread per tail layer replays the reader for the T callbacks of this call
__global__ void k_tail_schedule(TailState* st, long long* __restrict__ sched, int T, int B, int nTail, int2 l1, int2 l2,
                                int d1, int d2)
{
    const int l = threadIdx.x;
    const long long cb0 = st->cb;
    if (l < nTail) {
        const int PL = l == 0 ? l1.x : l2.x;       // layer partition size
        const int oL = l == 0 ? l1.y : l2.y;       // outputDelaySamples
        const int D = l == 0 ? d1 : d2;            // callbacks between partition fill and delay-line write
        const int bpp = PL / B;
        long long R = st->R[l];
        for (int i = 0; i < T; ++i) {
            const long long c = cb0 + i;
            const long long done = (c - D + 1 > 0) ? (c - D + 1) / bpp : 0;
            const long long W = done * PL;                               // delayWriteCursor after Add() of callback c
            const long long maxRead = W - oL > 0 ? W - oL : 0;
            const long long start = R > maxRead ? R : maxRead;
            if (start + B > W) sched[(long long)l * T + i] = -1;         // "writer not far enough ahead": skip
            else { sched[(long long)l * T + i] = start; R = start + B; }
        }
        st->R[l] = R;
    }
    __syncthreads();
    if (l == 0) { st->cb = cb0 + T; st->g0 = cb0 * (long long)B; }
}

// what later calls may still read of this call's tail outputs goes into the per-layer rings at (global sample index &
// mask): everything from the layer's read cursor on (the reader never steps back), at most one ring's worth.  Runs behind
// the layer-0 inverse transform, whose stores add the reader's samples (fft_kernels.hip: store_block2<2>) and still read
// the older contents.
__global__ __launch_bounds__(256) void k_tail_append(const double* __restrict__ layerOut, double* __restrict__ ring,
                                                     const TailState* __restrict__ st, int nCh, int nSamples, int ringMask)
{
    const int v = blockIdx.y;                      // (tail layer, channel)
    const int l = v / nCh;
    const double* src = layerOut + (long long)v * nSamples;
    double* dst = ring + (long long)v * (ringMask + 1);
    const long long g0 = st->g0;
    long long from = st->R[l] - g0;
    if (from < nSamples - (ringMask + 1)) from = nSamples - (ringMask + 1);
    if (from < 0) from = 0;
    const int stride = gridDim.x * blockDim.x;
    for (long long i = from + blockIdx.x * blockDim.x + threadIdx.x; i < nSamples; i += stride)
        dst[(g0 + i) & ringMask] = src[i];
}

// ---------------------------------------------------------------------------------------------------------
// AGC of the EQ (EQProcessor::processAGC, src/eqprocessor/EQProcessor.Processing.cpp:367-445): block-rate RMS
// envelopes, one gain per callback block, applied as a linear ramp.  The arithmetic replays the reference's
// accumulation orders so the result is bit-identical given the same inputs.

// calculateRMS (:21-52): four FMA accumulator lanes over i%4, summed left to right, sqrt(sum / n).
// 4 threads per (channel, block); 16 such groups per wave.
__global__ __launch_bounds__(64) void k_agc_block_rms(const double* __restrict__ x, int64_t chStride, int nCh, int B, int T,
                                                      double* __restrict__ rms)
{
    const int grp = blockIdx.x * 16 + (threadIdx.x >> 2);
    const int j = threadIdx.x & 3;
    const bool live = grp < nCh * T;
    const int c = live ? grp / T : 0;
    const int t = live ? grp - c * T : 0;
    const double* d = x + (int64_t)c * chStride + (int64_t)t * B;
    double acc = 0.0;
    const int vEnd = B & ~3;              // four accumulator lanes over the whole groups of 4, then the scalar remainder
    if (live)
        for (int i = j; i < vEnd; i += 4) acc = fma(d[i], d[i], acc);
    const int base = threadIdx.x & ~3;
    const double a0 = __shfl(acc, base), a1 = __shfl(acc, base + 1), a2 = __shfl(acc, base + 2), a3 = __shfl(acc, base + 3);
    if (live && j == 0) {
        double sumSq = ((a0 + a1) + a2) + a3;
        for (int i = vEnd; i < B; ++i) sumSq += d[i] * d[i];
        rms[grp] = sqrt(sumSq / (double)B);
    }
}

// one thread per stream walks the callbacks of the call: envelopes, target gain, smoothed gain (:411-438)
__global__ __launch_bounds__(64) void k_agc_gains(const double* __restrict__ rmsIn, const double* __restrict__ rmsOut,
                                                  double* __restrict__ state, const int* __restrict__ agcOn,
                                                  double* __restrict__ gains, int S, int T, int B, double bAtt, double bRel,
                                                  double bSm)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S || !agcOn[s]) return;
    double envIn = state[3 * s], envOut = state[3 * s + 1], cur = state[3 * s + 2] + 1.0;
    for (int t = 0; t < T; ++t) {
        double inR = fmax(rmsIn[(2 * s) * T + t], rmsIn[(2 * s + 1) * T + t]);        // max over channels, starting from 0
        double outR = fmax(rmsOut[(2 * s) * T + t], rmsOut[(2 * s + 1) * T + t]);
        inR = fmax(inR, 0.0); outR = fmax(outR, 0.0);
        if (!((inR - inR) == 0.0) || inR > 1000.0) inR = 1000.0;
        if (!((outR - outR) == 0.0) || outR > 1000.0) outR = 1000.0;
        const double inA = (inR > envIn) ? bAtt : bRel;
        const double outA = (outR > envOut) ? bAtt : bRel;
        envIn = envIn * (1.0 - inA) + inR * inA;
        envOut = envOut * (1.0 - outA) + outR * outA;
        if (envIn < 1.0e-20) envIn = 0.0;
        if (envOut < 1.0e-20) envOut = 0.0;
        double target = 1.0;                                   // calculateAGCGain (:343-358)
        if (!(envOut < 1.0e-6)) {
            const double ratio = envIn / envOut;
            if (!(ratio > 1.0 / 1.059 && ratio < 1.059)) {
                const double lo = (double)0.06f, hi = (double)16.0f;
                target = ratio < lo ? lo : (ratio > hi ? hi : ratio);
            }
        }
        const double next = cur * (1.0 - bSm) + target * bSm;
        gains[((int64_t)s * T + t) * 2] = cur;
        gains[((int64_t)s * T + t) * 2 + 1] = (next - cur) / (double)B;
        cur = next;
    }
    state[3 * s] = envIn; state[3 * s + 1] = envOut; state[3 * s + 2] = cur - 1.0;
}

// applyGainRamp_AVX2 (:279-337): gain of sample i = 16 m + 4 q + j is lane j's start value advanced m times by
// 16*inc and then q times by 4*inc, each advance a separate rounded addition
__global__ __launch_bounds__(256) void k_agc_ramp(double* data, int64_t chStride, const double* __restrict__ gains,
                                                  const int* __restrict__ agcOn, int B, int T)
{
    const int c = blockIdx.y;
    const int s = c >> 1;
    if (!agcOn[s]) return;
    double* d = data + (int64_t)c * chStride;
    const int n = B * T;
    const int stride = gridDim.x * blockDim.x;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
        const int t = idx / B, i = idx - t * B;
        const double start = gains[((int64_t)s * T + t) * 2], inc = gains[((int64_t)s * T + t) * 2 + 1];
        const int j = i & 3, q = (i >> 2) & 3, mm = i >> 4;
        const int vEnd4 = B & ~3;
        double g;
        if (i >= vEnd4) {             // scalar remainder of a callback whose length is not a multiple of 4 (:331-332)
            g = start + (double)vEnd4 * inc;
            for (int k = vEnd4; k < i; ++k) g = g + inc;
        } else {
            g = (j == 0) ? start : (j == 1 ? start + inc : (j == 2 ? start + 2.0 * inc : start + 3.0 * inc));
            const double inc4 = 4.0 * inc, inc16 = 16.0 * inc;
            for (int k = 0; k < mm; ++k) g = g + inc16;
            for (int k = 0; k < q; ++k) g = g + inc4;
        }
        d[idx] *= g;
    }
}


// ---- FilterSpec tail layers run at the reference's own partition size (engine_conv.cpp, SpecTail) -------------------
// dst[c][dstOff + i] = src[c][srcOff + i], i < n: input accumulation of a tail layer (inputAccBuf, NUC.cpp:1433-1452)
__global__ __launch_bounds__(256) void k_rows_copy(const double* __restrict__ src, int64_t srcStride, int64_t srcOff,
                                                   double* __restrict__ dst, int64_t dstStride, int64_t dstOff, int n)
{
    const double* s = src + (int64_t)blockIdx.y * srcStride + srcOff;
    double* d = dst + (int64_t)blockIdx.y * dstStride + dstOff;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = s[i];
}

// ring[c][(pos + i) & mask] = z[c][i]: completed tail blocks into the delay line (delayLineWrite, NUC.cpp:1639-1648)
__global__ __launch_bounds__(256) void k_ring_put(const double* __restrict__ z, int64_t zStride, int n,
                                                  double* __restrict__ ring, int mask, long long pos)
{
    const double* s = z + (int64_t)blockIdx.y * zStride;
    double* r = ring + (int64_t)blockIdx.y * (mask + 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        r[(pos + i) & mask] = s[i];
}

// delayLineReadAdd (NUC.cpp:1653-1688) for the callbacks of one call: callback cb reads B samples at delay-line position
// sched[cb] (k_tail_schedule replays the reader; -1 = "writer not far enough ahead", nothing is added):
// out[c][cb B + j] += ring[c][(sched[cb] + j) & mask] * gain
__global__ __launch_bounds__(256) void k_ring_add(double* out, int64_t outStride, int n, int B, const double* __restrict__ ring,
                                                  int mask, const long long* __restrict__ sched, double gain)
{
    double* o = out + (int64_t)blockIdx.y * outStride;
    const double* r = ring + (int64_t)blockIdx.y * (mask + 1);
    const bool unity = fabs(gain - 1.0) < 1.0e-12;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int cb = i / B, j = i - cb * B;
        const long long s0 = sched[cb];
        if (s0 < 0) continue;
        const double v = r[(s0 + j) & mask];
        o[i] = unity ? (o[i] + v) : (o[i] + v * gain);
    }
}


// ---- plan groups (engine_native.cpp): a group's channels are rows chMap[local] of the call's buffers (-1 = unused slot);
// every Add / Get pair of the reference is one CHUNK of the call (q samples, the last one possibly shorter), whose read
// positions the host replays (ringRead :1376-1402, delayLineReadAdd :1653-1688) and uploads per call.
// acc[local][dstOff + i] = in[chMap[local]][i]: input accumulation of one layer (inputAccBuf, NUC.cpp:1431-1446)
__global__ __launch_bounds__(256) void k_rows_gather(const double* __restrict__ src, int64_t srcStride,
                                                     const int* __restrict__ chMap, double* __restrict__ dst,
                                                     int64_t dstStride, int64_t dstOff, int n)
{
    const int g = chMap[blockIdx.y];
    if (g < 0) return;
    const double* s = src + (int64_t)g * srcStride;
    double* d = dst + (int64_t)blockIdx.y * dstStride + dstOff;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] = s[i];
}

// ring[c][(pos[j] + i) & mask] = z[c][j P + i]: the blocks a layer finished in this call go to its output ring / delay
// line at the position each one has in the reference (ringWrite :1341-1371, delayLineWrite :1639-1648)
__global__ __launch_bounds__(256) void k_ring_put_blocks(const double* __restrict__ z, int64_t zStride, int P, int nb,
                                                         double* __restrict__ ring, int mask,
                                                         const long long* __restrict__ pos)
{
    const double* s = z + (int64_t)blockIdx.y * zStride;
    double* r = ring + (int64_t)blockIdx.y * (mask + 1);
    const int n = nb * P;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = i / P;
        const long long p = pos[j];
        if (p >= 0) r[(p + (i - j * P)) & mask] = s[i];
    }
}

// Get() of layer 0 per chunk: cnt[cb] samples from ring position pos[cb], zero-filled to the chunk's end (:1376-1402)
// the call's input into the accumulators of up to three layers of a plan group at once (Add(): every layer accumulates the
// same input, src/MKLNonUniformConvolver.cpp:1431-1446): the input row is read once
struct GatherDst { double* dst[3]; long long stride[3]; long long off[3]; int n; };
struct GatherTab { long long* dst; int n; long long v[kGatherTabMax]; };
__global__ __launch_bounds__(256) void k_rows_gather_multi(const double* __restrict__ src, int64_t srcStride,
                                                           const int* __restrict__ chMap, GatherDst d, int n, GatherTab tab)
{
    // the call's chunk tables (kernel arguments) into device memory for the kernels behind this one
    if (blockIdx.x == 0 && blockIdx.y == 0 && (int)threadIdx.x < tab.n) tab.dst[threadIdx.x] = tab.v[threadIdx.x];
    const int g = chMap[blockIdx.y];
    if (g < 0) return;
    const double* s = src + (int64_t)g * srcStride;
    double* d0 = d.dst[0] + (int64_t)blockIdx.y * d.stride[0] + d.off[0];
    double* d1 = d.n > 1 ? d.dst[1] + (int64_t)blockIdx.y * d.stride[1] + d.off[1] : nullptr;
    double* d2 = d.n > 2 ? d.dst[2] + (int64_t)blockIdx.y * d.stride[2] + d.off[2] : nullptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double v = s[i];
        d0[i] = v;
        if (d1) d1[i] = v;
        if (d2) d2[i] = v;
    }
}

// delay-line read-add of both tail layers in one pass over the output (Get(), :1620-1633: layer 1 first, then layer 2)
__global__ __launch_bounds__(256) void k_ring_add_chunks2(double* __restrict__ out, int64_t outStride,
                                                          const int* __restrict__ chMap, int n, int q,
                                                          const double* __restrict__ ringA, int maskA,
                                                          const long long* __restrict__ schedA, double gainA,
                                                          const double* __restrict__ ringB, int maskB,
                                                          const long long* __restrict__ schedB, double gainB)
{
    const int g = chMap[blockIdx.y];
    if (g < 0) return;
    double* o = out + (int64_t)g * outStride;
    const double* ra = ringA + (int64_t)blockIdx.y * (maskA + 1);
    const double* rb = ringB + (int64_t)blockIdx.y * (maskB + 1);
    const bool unityA = fabs(gainA - 1.0) < 1.0e-12, unityB = fabs(gainB - 1.0) < 1.0e-12;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int cb = i / q, j = i - cb * q;
        const long long sa = schedA[cb], sb = schedB[cb];
        if (sa < 0 && sb < 0) continue;
        double v = o[i];
        if (sa >= 0) { const double a = ra[(sa + j) & maskA]; v = unityA ? (v + a) : (v + a * gainA); }
        if (sb >= 0) { const double b = rb[(sb + j) & maskB]; v = unityB ? (v + b) : (v + b * gainB); }
        o[i] = v;
    }
}

__global__ __launch_bounds__(256) void k_ring_get_chunks(double* __restrict__ out, int64_t outStride,
                                                         const int* __restrict__ chMap, int n, int q,
                                                         const double* __restrict__ ring, int mask,
                                                         const long long* __restrict__ pos, const long long* __restrict__ cnt)
{
    const int g = chMap[blockIdx.y];
    if (g < 0) return;
    double* o = out + (int64_t)g * outStride;
    const double* r = ring + (int64_t)blockIdx.y * (mask + 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int cb = i / q, j = i - cb * q;
        o[i] = (j < cnt[cb]) ? r[(pos[cb] + j) & mask] : 0.0;
    }
}

// Get() of a whole call in ONE pass over the members' output rows: the chunk-wise read of layer 0's ring with its zero-fill
// (k_ring_get_chunks) and behind it the delay-line read-add of one or two tail layers (k_ring_add_chunks[2]); ringB may be null
__global__ __launch_bounds__(256) void k_ring_get_add_chunks(double* __restrict__ out, int64_t outStride,
                                                             const int* __restrict__ chMap, int n, int q,
                                                             const double* __restrict__ ring0, int mask0,
                                                             const long long* __restrict__ pos, const long long* __restrict__ cnt,
                                                             const double* __restrict__ ringA, int maskA,
                                                             const long long* __restrict__ schedA, double gainA,
                                                             const double* __restrict__ ringB, int maskB,
                                                             const long long* __restrict__ schedB, double gainB)
{
    const int g = chMap[blockIdx.y];
    if (g < 0) return;
    double* o = out + (int64_t)g * outStride;
    const double* r0 = ring0 + (int64_t)blockIdx.y * (mask0 + 1);
    const double* ra = ringA + (int64_t)blockIdx.y * (maskA + 1);
    const double* rb = ringB ? ringB + (int64_t)blockIdx.y * (maskB + 1) : nullptr;
    const bool unityA = fabs(gainA - 1.0) < 1.0e-12, unityB = fabs(gainB - 1.0) < 1.0e-12;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int cb = i / q, j = i - cb * q;
        double v = (j < cnt[cb]) ? r0[(pos[cb] + j) & mask0] : 0.0;
        const long long sa = schedA[cb], sb = rb ? schedB[cb] : -1;
        if (sa >= 0) { const double a = ra[(sa + j) & maskA]; v = unityA ? (v + a) : (v + a * gainA); }
        if (sb >= 0) { const double b = rb[(sb + j) & maskB]; v = unityB ? (v + b) : (v + b * gainB); }
        o[i] = v;
    }
}

// delayLineReadAdd per chunk (sched[cb] < 0: the writer is not far enough ahead, nothing is added)
__global__ __launch_bounds__(256) void k_ring_add_chunks(double* __restrict__ out, int64_t outStride,
                                                         const int* __restrict__ chMap, int n, int q,
                                                         const double* __restrict__ ring, int mask,
                                                         const long long* __restrict__ sched, double gain)
{
    const int g = chMap[blockIdx.y];
    if (g < 0) return;
    double* o = out + (int64_t)g * outStride;
    const double* r = ring + (int64_t)blockIdx.y * (mask + 1);
    const bool unity = fabs(gain - 1.0) < 1.0e-12;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int cb = i / q, j = i - cb * q;
        const long long s0 = sched[cb];
        if (s0 < 0) continue;
        const double v = r[(s0 + j) & mask];
        o[i] = unity ? (o[i] + v) : (o[i] + v * gain);
    }
}


// ---- direct head (processDirectBlock, src/MKLNonUniformConvolver.cpp:1169-1232): the first <= 32 taps run as a
// time-domain FIR over [history | block]; same accumulation pattern as the AVX2 loop (two 4-lane FMA accumulators over
// blocks of 8 taps, lanes summed as (0+2)+(1+3), scalar tail), result flushed to 0 when non-finite or below 1e-20.
// hist: the last 32 input samples of the previous call per channel (ping-pong: histOld read, histNew written).
__global__ __launch_bounds__(256) void k_direct_head(const double* __restrict__ in, int64_t inStride, int n,
                                                     const double* __restrict__ irRev, const int* __restrict__ taps,
                                                     const int* __restrict__ irSlot, const double* __restrict__ histOld,
                                                     double* __restrict__ histNew, double* __restrict__ dout,
                                                     const int* __restrict__ wetOn)
{
    const int c = blockIdx.y;
    if (wetOn && wetOn[c >> 1] == 0) {         // the stream's convolver rests: no output, the history stays as it is
        for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) dout[(int64_t)c * n + s] = 0.0;
        if (blockIdx.x == 0 && threadIdx.x < 32) histNew[c * 32 + threadIdx.x] = histOld[c * 32 + threadIdx.x];
        return;
    }
    const int slot = irSlot[c];
    const int nt = taps[slot];
    const double* h = irRev + slot * 32;
    const double* x = in + (int64_t)c * inStride;
    const double* ho = histOld + c * 32;
    for (int s = blockIdx.x * blockDim.x + threadIdx.x; s < n; s += gridDim.x * blockDim.x) {
        double y = 0.0;
        if (nt > 0) {
            auto win = [&](int k) {                       // window[s + k], window = [last nt-1 samples | block]
                const int p = s + k - (nt - 1);
                return p >= 0 ? x[p] : ho[32 + p];
            };
            double a0[4] = { 0, 0, 0, 0 }, a1[4] = { 0, 0, 0, 0 };
            const int v8 = (nt / 8) * 8;
            int k = 0;
            for (; k < v8; k += 8) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a0[j] = fma(h[k + j], win(k + j), a0[j]);
                    a1[j] = fma(h[k + 4 + j], win(k + 4 + j), a1[j]);
                }
            }
            double v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = a0[j] + a1[j];
            y = (v[0] + v[2]) + (v[1] + v[3]);
            for (; k < nt; ++k) y += h[k] * win(k);
            if (!(fabs(y) >= 1.0e-20 && fabs(y) <= 1.79769313486231570815e308)) y = 0.0;    // non-finite or below the threshold
        }
        dout[(int64_t)c * n + s] = y;
    }
    // next history = last 32 samples of (history ++ block)
    if (blockIdx.x == 0 && threadIdx.x < 32) {
        const int p = n - 32 + (int)threadIdx.x;
        histNew[c * 32 + threadIdx.x] = p >= 0 ? x[p] : ho[32 + p];
    }
}

// out[c][i] += add[c][i]  (Get(): direct output added to the ring output, :1609-1618)
__global__ __launch_bounds__(256) void k_rows_add(double* out, int64_t outStride, const double* __restrict__ add, int n)
{
    double* o = out + (int64_t)blockIdx.y * outStride;
    const double* a = add + (int64_t)blockIdx.y * n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) o[i] += a[i];
}

// EQ bypass cross-fade (EQProcessor.Processing.cpp:977-1003): out = out * g + dry * (1 - g) for the streams flagged in
// `on`; g = gains[s][i] for the first len[s] samples (the LinearRamp's values, formed on the host), gEnd[s] after.
__global__ __launch_bounds__(256) void k_bypass_blend(double* out, int64_t outStride, const double* __restrict__ dry,
                                                      int64_t dryStride, int n, const int* __restrict__ on,
                                                      const int* __restrict__ len, const double* __restrict__ gEnd,
                                                      const double* __restrict__ gains, int cap)
{
    const int c = blockIdx.y;
    const int s = c >> 1;
    if (!on[s]) return;
    double* o = out + (int64_t)c * outStride;
    const double* d = dry + (int64_t)c * dryStride;
    const int nr = len[s];
    const double ge = gEnd[s];
    const double* g = gains + (int64_t)s * cap;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double wg = i < nr ? g[i] : ge;
        const double dg = 1.0 - wg;
        o[i] = (o[i] * wg) + (d[i] * dg);
    }
}

// isAudioBlockSilent (src/eqprocessor/EQProcessor.Processing.cpp:460-475) per (stream, callback): 1 when no sample of
// either channel exceeds 1e-8 in magnitude.  One wave per (stream, callback).
__global__ __launch_bounds__(64) void k_block_silence(const double* __restrict__ x, int64_t chStride, int B, int T, int S,
                                                      int* __restrict__ silent)
{
    const int idx = blockIdx.x;
    if (idx >= S * T) return;
    const int s = idx / T, t = idx - s * T;
    int loud = 0;
    for (int ch = 0; ch < 2; ++ch) {
        const double* p = x + (int64_t)(2 * s + ch) * chStride + (int64_t)t * B;
        for (int i = threadIdx.x; i < B; i += 64) loud |= (fabs(p[i]) > 1.0e-8) ? 1 : 0;
    }
    const unsigned long long any = __ballot(loud);
    if (threadIdx.x == 0) silent[idx] = any == 0ull ? 1 : 0;
}

// data[c][i] *= gain[stream] for the streams whose gain is not exactly 1 (scaleBlockFallback,
// src/audioengine/AudioEngine.Processing.DSPCoreDouble.cpp:93-105)
__global__ __launch_bounds__(256) void k_rows_scale(double* data, int64_t stride, int n, const double* __restrict__ gain)
{
    const double g = gain[blockIdx.y >> 1];
    if (g == 1.0) return;
    double* d = data + (int64_t)blockIdx.y * stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) d[i] *= g;
}

}  // namespace

void launch_agc_block_rms(hipStream_t stream, const double* x, int64_t chStride, int nCh, int B, int T, double* rms)
{
    hipLaunchKernelGGL(k_agc_block_rms, dim3((nCh * T + 15) / 16), dim3(64), 0, stream, x, chStride, nCh, B, T, rms);
}

void launch_agc_apply(hipStream_t stream, double* data, int64_t chStride, int S, int B, int T, const double* rmsIn,
                      const double* rmsOut, double* state, const int* agcOn, double* gains, double bAtt, double bRel,
                      double bSm)
{
    hipLaunchKernelGGL(k_agc_gains, dim3((S + 63) / 64), dim3(64), 0, stream, rmsIn, rmsOut, state, agcOn, gains, S, T, B,
                       bAtt, bRel, bSm);
    int bx = (B * T + 255) / 256;
    if (bx > 32) bx = 32;
    hipLaunchKernelGGL(k_agc_ramp, dim3(bx, 2 * S), dim3(256), 0, stream, data, chStride, gains, agcOn, B, T);
}

void launch_gain_ramp(hipStream_t stream, double* data, int64_t chStride, int S, int B, int T, const double* gains,
                      const int* on)
{
    int bx = (B * T + 255) / 256;
    if (bx > 32) bx = 32;
    hipLaunchKernelGGL(k_agc_ramp, dim3(bx, 2 * S), dim3(256), 0, stream, data, chStride, gains, on, B, T);
}

void launch_convproc_mix(hipStream_t stream, const double* wet, double* out, int64_t chStride, int nCh, int nSamples,
                         const double* gains, const double* ring, int ringSize, long long pos0, const int* dNew,
                         const int* dOld, const int* xLen, const double* xGains, int xCap, int wetValid,
                         const int* rampLen, const double* rampGains, int rampCap, int rampOff, const int* wetOn)
{
    if (nSamples <= 0) return;
    int bx = (nSamples + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_convproc_mix, dim3(bx, nCh), dim3(256), 0, stream, wet, out, chStride, nSamples, gains, ring,
                       ringSize - 1, pos0, dNew, dOld, xLen, xGains, xCap, wetValid, rampLen, rampGains, rampCap, rampOff, wetOn);
}

void launch_ring_regrow(hipStream_t stream, const double* oldRing, int oldSize, double* newRing, int newSize, long long end,
                        int nCh)
{
    int bx = (oldSize + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_ring_regrow, dim3(bx, nCh), dim3(256), 0, stream, oldRing, oldSize - 1, newRing, newSize - 1, end);
}

static dim3 rowsGrid(int n, int nCh) { int bx = (n + 255) / 256; if (bx > 64) bx = 64; if (bx < 1) bx = 1; return dim3(bx, nCh); }

void launch_rows_copy(hipStream_t stream, const double* src, int64_t srcStride, int64_t srcOff, double* dst,
                      int64_t dstStride, int64_t dstOff, int n, int nCh)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_rows_copy, rowsGrid(n, nCh), dim3(256), 0, stream, src, srcStride, srcOff, dst, dstStride, dstOff, n);
}

void launch_bypass_blend(hipStream_t stream, double* out, int64_t outStride, const double* dry, int64_t dryStride, int n,
                         int nCh, const int* on, const int* len, const double* gEnd, const double* gains, int cap)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_bypass_blend, rowsGrid(n, nCh), dim3(256), 0, stream, out, outStride, dry, dryStride, n, on, len,
                       gEnd, gains, cap);
}

void launch_block_silence(hipStream_t stream, const double* x, int64_t chStride, int B, int T, int S, int* silent)
{
    if (S * T <= 0) return;
    hipLaunchKernelGGL(k_block_silence, dim3(S * T), dim3(64), 0, stream, x, chStride, B, T, S, silent);
}

void launch_rows_scale(hipStream_t stream, double* data, int64_t stride, int n, int nCh, const double* gain)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_rows_scale, rowsGrid(n, nCh), dim3(256), 0, stream, data, stride, n, gain);
}

void launch_ring_put(hipStream_t stream, const double* z, int64_t zStride, int n, double* ring, int ringSize,
                     long long pos, int nCh)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_ring_put, rowsGrid(n, nCh), dim3(256), 0, stream, z, zStride, n, ring, ringSize - 1, pos);
}

void launch_ring_add(hipStream_t stream, double* out, int64_t outStride, int n, int B, const double* ring, int ringSize,
                     const long long* sched, double gain, int nCh)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_ring_add, rowsGrid(n, nCh), dim3(256), 0, stream, out, outStride, n, B, ring, ringSize - 1, sched, gain);
}

void launch_rows_gather(hipStream_t stream, const double* src, int64_t srcStride, const int* chMap, double* dst,
                        int64_t dstStride, int64_t dstOff, int n, int nCh)
{
    if (n <= 0 || nCh <= 0) return;
    hipLaunchKernelGGL(k_rows_gather, rowsGrid(n, nCh), dim3(256), 0, stream, src, srcStride, chMap, dst, dstStride, dstOff, n);
}

void launch_rows_gather_multi(hipStream_t stream, const double* src, int64_t srcStride, const int* chMap, int nLayers,
                              double* const* dst, const int64_t* dstStride, const int64_t* dstOff, int n, int nCh,
                              long long* tabDst, const long long* tab, int nTab)
{
    if (n <= 0 || nCh <= 0 || nLayers <= 0) return;
    GatherDst d{};
    d.n = nLayers > 3 ? 3 : nLayers;
    for (int l = 0; l < d.n; ++l) { d.dst[l] = dst[l]; d.stride[l] = dstStride[l]; d.off[l] = dstOff[l]; }
    GatherTab gt{};
    if (tabDst && tab && nTab > 0 && nTab <= kGatherTabMax) { gt.dst = tabDst; gt.n = nTab; for (int i = 0; i < nTab; ++i) gt.v[i] = tab[i]; }
    hipLaunchKernelGGL(k_rows_gather_multi, rowsGrid(n, nCh), dim3(256), 0, stream, src, srcStride, chMap, d, n, gt);
}

void launch_ring_add_chunks2(hipStream_t stream, double* out, int64_t outStride, const int* chMap, int n, int q,
                             const double* ringA, int ringSizeA, const long long* schedA, double gainA,
                             const double* ringB, int ringSizeB, const long long* schedB, double gainB, int nCh)
{
    if (n <= 0 || nCh <= 0) return;
    hipLaunchKernelGGL(k_ring_add_chunks2, rowsGrid(n, nCh), dim3(256), 0, stream, out, outStride, chMap, n, q, ringA,
                       ringSizeA - 1, schedA, gainA, ringB, ringSizeB - 1, schedB, gainB);
}

void launch_ring_get_add_chunks(hipStream_t stream, double* out, int64_t outStride, const int* chMap, int n, int q,
                                const double* ring0, int ringSize0, const long long* pos, const long long* cnt,
                                const double* ringA, int ringSizeA, const long long* schedA, double gainA,
                                const double* ringB, int ringSizeB, const long long* schedB, double gainB, int nCh)
{
    if (n <= 0 || nCh <= 0) return;
    hipLaunchKernelGGL(k_ring_get_add_chunks, rowsGrid(n, nCh), dim3(256), 0, stream, out, outStride, chMap, n, q, ring0, ringSize0 - 1, pos, cnt,
                       ringA, ringSizeA - 1, schedA, gainA, ringB, ringSizeB - 1, schedB, gainB);
}

void launch_ring_put_blocks(hipStream_t stream, const double* z, int64_t zStride, int P, int nb, double* ring, int ringSize,
                            const long long* pos, int nCh)
{
    if (nb <= 0 || nCh <= 0) return;
    hipLaunchKernelGGL(k_ring_put_blocks, rowsGrid(nb * P, nCh), dim3(256), 0, stream, z, zStride, P, nb, ring, ringSize - 1, pos);
}

void launch_ring_get_chunks(hipStream_t stream, double* out, int64_t outStride, const int* chMap, int n, int q,
                            const double* ring, int ringSize, const long long* pos, const long long* cnt, int nCh)
{
    if (n <= 0 || nCh <= 0) return;
    hipLaunchKernelGGL(k_ring_get_chunks, rowsGrid(n, nCh), dim3(256), 0, stream, out, outStride, chMap, n, q, ring,
                       ringSize - 1, pos, cnt);
}

void launch_ring_add_chunks(hipStream_t stream, double* out, int64_t outStride, const int* chMap, int n, int q,
                            const double* ring, int ringSize, const long long* sched, double gain, int nCh)
{
    if (n <= 0 || nCh <= 0) return;
    hipLaunchKernelGGL(k_ring_add_chunks, rowsGrid(n, nCh), dim3(256), 0, stream, out, outStride, chMap, n, q, ring,
                       ringSize - 1, sched, gain);
}

void launch_tail_append(hipStream_t stream, const void* state, const double* layerOut, double* ring, int nCh, int nSamples,
                        int ringSlots, int nTail)
{
    hipLaunchKernelGGL(k_tail_append, dim3(8, nCh * nTail), dim3(256), 0, stream, layerOut, ring, reinterpret_cast<const TailState*>(state),
                       nCh, nSamples, ringSlots - 1);
}

void launch_tail_schedule(hipStream_t stream, void* state, long long* sched, int T, int B, int nTail, int pl1, int ol1, int d1,
                          int pl2, int ol2, int d2)
{
    hipLaunchKernelGGL(k_tail_schedule, dim3(1), dim3(64), 0, stream, reinterpret_cast<TailState*>(state), sched, T, B, nTail,
                       make_int2(pl1, ol1), make_int2(pl2, ol2), d1, d2);
}

void launch_direct_head(hipStream_t stream, const double* in, int64_t inStride, int n, const double* irRev, const int* taps,
                        const int* irSlot, const double* histOld, double* histNew, double* dout, int nCh, const int* wetOn)
{
    hipLaunchKernelGGL(k_direct_head, rowsGrid(n, nCh), dim3(256), 0, stream, in, inStride, n, irRev, taps, irSlot, histOld,
                       histNew, dout, wetOn);
}

void launch_rows_add(hipStream_t stream, double* out, int64_t outStride, const double* add, int n, int nCh)
{
    hipLaunchKernelGGL(k_rows_add, rowsGrid(n, nCh), dim3(256), 0, stream, out, outStride, add, n);
}

}  // namespace cpq
