// mfma_f64_layout.hip -- checks the operand / result layout of v_mfma_f64_16x16x4_f64 that the time-parallel SVF kernel
// relies on:  A[m = lane & 15][k = lane >> 4],  B[k = lane >> 4][n = lane & 15],  D reg j = D[(lane >> 4) + 4 j][lane & 15]
// -- so the D registers of one product ARE the B operands (k-step s = register s) of the next one.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 mfma_f64_layout.hip -o /tmp/mfl && /tmp/mfl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void k(const double* T, const double* X, double* Y, double* Y2)
{
    const int l = threadIdx.x, m = l & 15, g = l >> 4;
    // Y = T (16x16) * X (16x16): X held as "register s = row 4 s + g, column m"
    double xr[4];
    for (int s = 0; s < 4; ++s) xr[s] = X[(4 * s + g) * 16 + m];
    v4d acc = { 0, 0, 0, 0 };
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T[m * 16 + 4 * s + g], xr[s], acc, 0, 0, 0);
    for (int j = 0; j < 4; ++j) Y[(g + 4 * j) * 16 + m] = acc[j];
    // second product straight from the accumulators: Y2 = T * Y
    v4d acc2 = { 0, 0, 0, 0 };
    for (int s = 0; s < 4; ++s) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(T[m * 16 + 4 * s + g], acc[s], acc2, 0, 0, 0);
    for (int j = 0; j < 4; ++j) Y2[(g + 4 * j) * 16 + m] = acc2[j];
}

int main()
{
    double hT[256], hX[256], hY[256], hY2[256], rY[256], rY2[256];
    for (int i = 0; i < 256; ++i) { hT[i] = std::sin(0.37 * i) + 0.1 * (i % 7); hX[i] = std::cos(0.11 * i * i); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += hT[i * 16 + k] * hX[k * 16 + j]; rY[i * 16 + j] = s; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += hT[i * 16 + k] * rY[k * 16 + j]; rY2[i * 16 + j] = s; }
    double *dT, *dX, *dY, *dY2;
    hipMalloc(&dT, sizeof(hT)); hipMalloc(&dX, sizeof(hX)); hipMalloc(&dY, sizeof(hY)); hipMalloc(&dY2, sizeof(hY2));
    hipMemcpy(dT, hT, sizeof(hT), hipMemcpyHostToDevice); hipMemcpy(dX, hX, sizeof(hX), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dT, dX, dY, dY2);
    hipMemcpy(hY, dY, sizeof(hY), hipMemcpyDeviceToHost); hipMemcpy(hY2, dY2, sizeof(hY2), hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 256; ++i) { e1 = std::fmax(e1, std::fabs(hY[i] - rY[i])); e2 = std::fmax(e2, std::fabs(hY2[i] - rY2[i])); }
    std::printf("max |Y - ref| = %.3e, max |Y2 - ref| = %.3e (chained product from the accumulators)\n", e1, e2);
    return (e1 < 1e-12 && e2 < 1e-10) ? 0 : 1;
}
