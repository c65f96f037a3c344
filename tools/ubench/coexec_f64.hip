// microbenchmark: do fp64 VALU FMAs and fp64 MFMAs of two waves on ONE SIMD execute side by side on gfx950, or do they
// share the fp64 datapath?  512-thread workgroups, one per CU: waves w and w + 4 land on the same SIMD.  Role A = a DFMA
// loop (independent chains), role B = a v_mfma_f64_16x16x4 / 4x4x4 loop (independent accumulators).  Timed: A alone
// (waves 4-7 exit), B alone (waves 0-3 exit), A beside B, and for reference A beside A / B beside B.
//   separate pipes:  t(A|B) ~ max(tA, tB);   shared datapath:  t(A|B) ~ tA + tB
// build: hipcc --offload-arch=gfx950 -O3 -o coexec_f64 coexec_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

template <int MF>   // MF: 0 = 16x16x4, 1 = 4x4x4
__global__ __launch_bounds__(512) void k(double* out, int itersA, int itersB, int roleLo, int roleHi, double a, double b)
{
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? roleLo : roleHi;      // 0 = exit, 1 = VALU, 2 = MFMA
    double s = 0.0;
    if (role == 1) {
        double acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 1e-3 + i;
        for (int it = 0; it < itersA; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i];
    } else if (role == 2) {
        if (MF == 0) {
            v4d acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = v4d{ a, b, a, b };
            for (int it = 0; it < itersB; ++it) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        } else {
            double acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = a + i;
            for (int it = 0; it < itersB; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[i];
        }
    } else {
        return;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MF>
float run(double* d, int itersA, int itersB, int lo, int hi)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MF><<<256, 512>>>(d, 10, 10, lo, hi, 0.999, 1e-3);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        k<MF><<<256, 512>>>(d, itersA, itersB, lo, hi, 0.999, 1e-3);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

int main()
{
    double* d;
    hipMalloc(&d, sizeof(double) * 256 * 512);
    for (int mf = 0; mf < 2; ++mf) {
        // 16 DFMA = 64 issue cycles per iteration; 4 x 16x16x4 = 256 pipe cycles, 16 x 4x4x4 ~ 256-320: balance the two roles
        const int itersA = 40000, itersB = 10000;
        auto r = [&](int lo, int hi) { return mf == 0 ? run<0>(d, itersA, itersB, lo, hi) : run<1>(d, itersA, itersB, lo, hi); };
        const float tA = r(1, 0), tB = r(0, 2), tAB = r(1, 2), tAA = r(1, 1), tBB = r(2, 2);
        printf("%s: VALU alone %.3f ms | MFMA alone %.3f ms | VALU beside MFMA %.3f ms (max %.3f, sum %.3f) | VALU beside VALU %.3f | MFMA beside MFMA %.3f\n",
               mf == 0 ? "mfma_f64_16x16x4" : "mfma_f64_4x4x4  ", tA, tB, tAB, tA > tB ? tA : tB, tA + tB, tAA, tBB);
    }
    hipFree(d);
    return 0;
}
