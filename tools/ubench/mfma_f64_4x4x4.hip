// mfma_f64_4x4x4.hip -- v_mfma_f64_4x4x4_4b_f64 on gfx950: (1) operand / result layout, found by brute force over the
// candidate index maps; (2) issue cost against v_mfma_f64_16x16x4_f64 (wave cycles per instruction from s_memtime with
// four independent accumulator chains per wave, one wave per SIMD and four waves per SIMD).
// Question behind it: a lower-triangular 16 x 16 Toeplitz product needs 10 of the 16 4 x 4 blocks -- would 10 small
// MFMAs beat 4 large ones?  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 mfma_f64_4x4x4.hip -o /tmp/m444 && /tmp/m444
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k_layout(const double* A, const double* B, double* D)
{
    // every lane supplies A[lane], B[lane] and stores D[lane]: the host then searches the index maps
    const int l = threadIdx.x;
    double d = __builtin_amdgcn_mfma_f64_4x4x4f64(A[l], B[l], 0.0, 0, 0, 0);
    D[l] = d;
}

template <int KIND>
__global__ void k_rate(long long* cycles, double* sink, int iters)
{
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 - threadIdx.x * 1e-4;
    long long t0, t1;
    if (KIND == 0) {
        double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        sink[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
    } else {
        v4d c0 = { 0, 0, 0, 0 }, c1 = c0, c2 = c0, c3 = c0;
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        sink[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int KIND>
static double timeRate(int wavesPerBlock, int iters)
{
    long long* dc; double* ds;
    hipMalloc(&dc, 8); hipMalloc(&ds, sizeof(double) * 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(1024), dim3(64 * wavesPerBlock), 0, 0, dc, ds, 16);          // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_rate<KIND>, dim3(1024), dim3(64 * wavesPerBlock), 0, 0, dc, ds, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long cyc; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    const double flopsPer = KIND == 0 ? 4.0 * 4 * 4 * 4 * 2 : 16.0 * 16 * 4 * 2;
    const double total = 1024.0 * wavesPerBlock * iters * 4.0 * flopsPer;
    std::printf("%s  %d waves/block: %.3f ms, %.1f TFLOP/s, wave-0 counter ticks per instruction %.1f\n",
                KIND == 0 ? "4x4x4_4b " : "16x16x4  ", wavesPerBlock, ms, total / ms * 1e-9, (double)cyc / (iters * 4.0));
    hipFree(dc); hipFree(ds);
    return ms;
}

int main()
{
    // ---- layout
    double hA[64], hB[64], hD[64];
    for (int i = 0; i < 64; ++i) { hA[i] = 1.0 + 0.01 * i + 0.3 * std::sin(i); hB[i] = 2.0 - 0.02 * i + 0.2 * std::cos(1.7 * i); }
    double *dA, *dB, *dD;
    hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 512);
    hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, 512, hipMemcpyDeviceToHost);
    // candidate maps: lane = 16 x + 4 y + z with (x, y, z) any permutation of (block, row, column) for each of A[i][k],
    // B[k][j], D[i][j]
    const int perm[6][3] = { {0,1,2}, {0,2,1}, {1,0,2}, {1,2,0}, {2,0,1}, {2,1,0} };
    const char* names[3] = { "b", "r", "c" };
    auto lane = [&](int p, int b, int r, int c) { const int v[3] = { b, r, c }; return 16 * v[perm[p][0]] + 4 * v[perm[p][1]] + v[perm[p][2]]; };
    int found = 0;
    for (int ma = 0; ma < 6; ++ma) for (int mb = 0; mb < 6; ++mb) for (int md = 0; md < 6; ++md) {
        double err = 0;
        for (int b = 0; b < 4; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
            double sum = 0;
            for (int k = 0; k < 4; ++k) sum += hA[lane(ma, b, i, k)] * hB[lane(mb, b, k, j)];
            err = std::fmax(err, std::fabs(sum - hD[lane(md, b, i, j)]));
        }
        if (err < 1e-12) {
            std::printf("layout (lane = 16 x + 4 y + z; b block, r row, c column of the operand): A[i][k] (%s,%s,%s)  B[k][j] (%s,%s,%s)  D[i][j] (%s,%s,%s)\n",
                        names[perm[ma][0]], names[perm[ma][1]], names[perm[ma][2]], names[perm[mb][0]], names[perm[mb][1]], names[perm[mb][2]],
                        names[perm[md][0]], names[perm[md][1]], names[perm[md][2]]);
            ++found;
        }
    }
    if (!found) {
        std::printf("layout: none of the 216 candidate maps matches; D:");
        for (int i = 0; i < 64; ++i) std::printf(" %.4f", hD[i]);
        std::printf("\n");
    }
    // ---- rate
    for (int w : { 4, 16 }) { timeRate<0>(w, 4096); timeRate<1>(w, 1024); }
    return 0;
}
