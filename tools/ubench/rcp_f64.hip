// Issue cost of v_rcp_f64 against v_fma_f64 (and the f32 route cvt -> v_rcp_f32 -> cvt) on gfx950: independent chains,
// one to four waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/rcp tools/ubench/rcp_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a, double b)
{
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = 3.0 + threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) acc[i] = fma(acc[i], a, b);
            else if (MODE == 1) acc[i] = __builtin_amdgcn_rcp(acc[i]);
            else if (MODE == 2) acc[i] = (double)__builtin_amdgcn_rcpf((float)acc[i]);
            else { acc[i] = __builtin_amdgcn_rcp(acc[i]); acc[i] = fma(acc[i], a, b); acc[i] = fma(acc[i], a, b); acc[i] = fma(acc[i], a, b); }
        }
    }
    double s = 0.0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(double* d, int blocks, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(d, iters, 0.999, 1e-3);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    double* d; hipMalloc(&d, 256 * 4 * 256 * 8 * sizeof(double));
    const int iters = 20000;
    for (int wavesPerSimd = 1; wavesPerSimd <= 4; wavesPerSimd *= 2) {
        const int blocks = 256 * wavesPerSimd;      // 4 waves per block = one per SIMD
        const float tf = run<0>(d, blocks, iters), tr = run<1>(d, blocks, iters), t32 = run<2>(d, blocks, iters), tm = run<3>(d, blocks, iters);
        printf("%d wave(s) per SIMD: 16 x %d ops per lane | fma %.3f ms | v_rcp_f64 %.3f ms (%.2f fma) | cvt+v_rcp_f32+cvt %.3f ms (%.2f fma) | rcp + 3 fma %.3f ms (%.2f fma)\n",
               wavesPerSimd, iters, tf, tr, tr / tf, t32, t32 / tf, tm, tm / tf);
    }
    return 0;
}
