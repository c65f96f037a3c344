// clock_under_load.hip -- what shader clock does the part hold while every SIMD issues back-to-back fp64 FMAs?
// s_memtime (clock64) counts shader-clock cycles, s_memrealtime (wall_clock64) counts a constant 100 MHz reference:
// their ratio over a long DFMA loop is the sustained frequency; cycles per FMA follows from the known instruction
// count.  Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 clock_under_load.hip -o /tmp/cul && /tmp/cul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, long long* clk, int iters, double a, double b)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main()
{
    for (int wavesPerSimd : { 1, 2, 4 }) {
        const int blocks = 256 * wavesPerSimd, iters = 40000;
        constexpr int NACC = 16;
        double* d; long long* c;
        hipMalloc(&d, sizeof(double) * blocks * 256);
        hipMalloc(&c, sizeof(long long) * 2 * blocks);
        k<NACC><<<blocks, 256>>>(d, c, 100, 0.999, 1e-3);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<NACC><<<blocks, 256>>>(d, c, iters, 0.999, 1e-3);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(2 * blocks);
        hipMemcpy(h.data(), c, sizeof(long long) * 2 * blocks, hipMemcpyDeviceToHost);
        double sc = 0, sw = 0;
        for (int i = 0; i < blocks; ++i) { sc += h[2 * i]; sw += h[2 * i + 1]; }
        const double mhz = sc / sw * 100.0;                         // shader cycles per 100 MHz tick
        const double cyclesPerFma = (sc / blocks) / ((double)iters * NACC * wavesPerSimd);   // per SIMD, waves interleaved
        printf("waves/SIMD %d: %.3f ms, %.2f TFLOP/s, sustained shader clock %.0f MHz, %.2f cycles per wave-FMA per SIMD\n",
               wavesPerSimd, ms, 2.0 * NACC * iters * (double)blocks * 256 / ms / 1e9, mhz, cyclesPerFma);
        hipFree(d); hipFree(c);
    }
    return 0;
}
