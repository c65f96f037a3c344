// microbenchmark: fp64 FMA issue under dependency on gfx950 -- NACC independent chains per wave (1 = fully dependent),
// 1/2/4 waves per SIMD; reports cycles per FMA per wave from s_memtime and the sustained TFLOP/s.
// What it answers: how many independent fp64 operations a wave needs between two dependent ones.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters, double a, double b)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
void run(int wavesPerSimd)
{
    const int blocks = 256 * wavesPerSimd;   // 256 CUs x (wavesPerSimd blocks of 4 waves)
    double* d; long long* c;
    hipMalloc(&d, sizeof(double) * blocks * 256);
    hipMalloc(&c, sizeof(long long) * blocks);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(d, c, 100, 0.999, 1e-3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(d, c, iters, 0.999, 1e-3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), c, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= blocks;
    const double nf = 8.0 * NACC * iters;
    const double flops = 2.0 * nf * (double)blocks * 256;
    printf("chains %2d waves/SIMD %d: %.3f ms  %.2f TFLOP/s  %.2f cycles per FMA per wave (memtime ticks)\n", NACC, wavesPerSimd, ms,
           flops / ms / 1e9, mean / nf);
    hipFree(d); hipFree(c);
}
int main()
{
    for (int w : {1, 2, 4}) { run<1>(w); run<2>(w); run<3>(w); run<4>(w); run<8>(w); }
    return 0;
}
