// microbenchmark: sustained fp64 FMA rate on gfx950 for 1/2/4 waves per SIMD and 8/16/32 independent chains
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wavesPerSimd)
{
    const int blocks = 256 * wavesPerSimd;   // 256 CUs x (wavesPerSimd blocks of 4 waves)
    double* d;
    hipMalloc(&d, sizeof(double) * blocks * 256);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(d, 100, 0.999, 1e-3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(d, iters, 0.999, 1e-3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * NACC * iters * (double)blocks * 256;
    printf("nacc %2d waves/SIMD %d: %.3f ms  %.2f TFLOP/s\n", NACC, wavesPerSimd, ms, flops / ms / 1e9);
    hipFree(d);
}
int main()
{
    for (int w : {1, 2, 4}) { run<4>(w); run<8>(w); run<16>(w); run<32>(w); }
    return 0;
}
