// mix_kernels.hip -- processor-level dry/wet stage of the convolver.
//
// Steady state of ConvolverProcessor::process (src/convolver/ConvolverProcessor.Runtime.cpp): the wet chunk is
// sanitised (:50-60, :722), the dry signal is read from a delay line (:549-567) and both are mixed with the
// equal-power gains (mixSteadySmall, :635-657: mul, mul, add -- not fused).  HBM-trivial elementwise work.
#include "kernels.hpp"

namespace cpq {

namespace {

__device__ __forceinline__ double dry_at(const double* __restrict__ hist, const double* dryIn, int histCap, int n)
{
    // sample index n of the current call, n may be negative (history)
    return n >= 0 ? dryIn[n] : hist[histCap + n];
}

__global__ __launch_bounds__(256) void k_convproc_mix(const double* wet, const double* dryIn, double* out,
                                                      int64_t chStride, int nSamples,
                                                      const double* __restrict__ gains,
                                                      const int* __restrict__ delay,
                                                      const double* __restrict__ histOld,
                                                      double* __restrict__ histNew, int histCap, int wetValid)
{
    const int c = blockIdx.y;
    const int s = c >> 1;
    const double wetG = gains[2 * s], dryG = gains[2 * s + 1];
    const int d = delay[s];
    const double* w = wet + (int64_t)c * chStride;
    const double* x = dryIn + (int64_t)c * chStride;
    double* o = out + (int64_t)c * chStride;
    const double* ho = histOld + (int64_t)c * histCap;
    double* hn = histNew + (int64_t)c * histCap;
    const int stride = gridDim.x * blockDim.x;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < nSamples; n += stride) {
        const double dry = dry_at(ho, x, histCap, n - d);
        if (wetValid) {
            double wv = w[n];
            // isFiniteAndAbsBelowNoLibm(x, 1e300): false for NaN / Inf
            wv = (fabs(wv) < 1.0e300) ? wv : 0.0;
            o[n] = (wv * wetG) + (dry * dryG);
        } else {
            o[n] = dry;           // dry-only fast path (:573-585) and bypass (:123-186): plain copy of the delayed input
        }
    }
    // next history = last histCap samples of (history ++ input)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < histCap; i += stride)
        hn[i] = dry_at(ho, x, histCap, nSamples - histCap + i);
}

}  // namespace

void launch_convproc_mix(hipStream_t stream, const double* wet, const double* dryIn, double* out, int64_t chStride,
                         int nCh, int nSamples, const double* gains, const int* delay, const double* histOld,
                         double* histNew, int histCap, int wetValid)
{
    const int work = nSamples > histCap ? nSamples : histCap;
    int bx = (work + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_convproc_mix, dim3(bx, nCh), dim3(256), 0, stream, wet, dryIn, out, chStride, nSamples, gains,
                       delay, histOld, histNew, histCap, wetValid);
}

}  // namespace cpq
