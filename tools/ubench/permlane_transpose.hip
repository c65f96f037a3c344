// permlane_transpose.hip -- 4 x 4 transpose across the four 16-lane rows of a wave with v_permlane32_swap /
// v_permlane16_swap (gfx950), as the eight-wave SVF kernel uses it to load a tile of 16 chunks x 16 samples with two
// 16-byte loads per lane (lane (m, g) reads samples 4g .. 4g+3 of chunk m) and still end up in the MFMA layout
// (register j = sample g + 4j).  Checks the helper against the direct 8-byte loads.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 permlane_transpose.hip -o /tmp/plt && /tmp/plt
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ void swap32(double& a, double& b)      // rows 2,3 of a <-> rows 0,1 of b
{
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double(h[0], l[0]);
    b = __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ void swap16(double& a, double& b)      // rows 1,3 of a <-> rows 0,2 of b
{
    unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double(h[0], l[0]);
    b = __hiloint2double(h[1], l[1]);
}
// s[i] at row g = element (g, i)  ->  s[j] at row g = element (j, g)
__device__ __forceinline__ void transpose4(double (&s)[4])
{
    swap32(s[0], s[2]);
    swap32(s[1], s[3]);
    swap16(s[0], s[1]);
    swap16(s[2], s[3]);
}

__global__ void k(const double* x, double* direct, double* viaT, double* back)
{
    const int lane = threadIdx.x, m = lane & 15, g = lane >> 4;
    double d[4], s[4];
    for (int j = 0; j < 4; ++j) d[j] = x[m * 16 + g + 4 * j];             // MFMA layout, 8-byte loads
    const double2 a = *reinterpret_cast<const double2*>(x + m * 16 + 4 * g);
    const double2 b = *reinterpret_cast<const double2*>(x + m * 16 + 4 * g + 2);
    s[0] = a.x; s[1] = a.y; s[2] = b.x; s[3] = b.y;
    transpose4(s);
    for (int j = 0; j < 4; ++j) { direct[lane * 4 + j] = d[j]; viaT[lane * 4 + j] = s[j]; }
    transpose4(s);                                                          // an involution: back to the load order
    for (int j = 0; j < 4; ++j) back[lane * 4 + j] = s[j];
}

int main()
{
    double hx[256], hd[256], ht[256], hb[256];
    for (int i = 0; i < 256; ++i) hx[i] = 1000.0 + i + 1e-9 * i;
    double *dx, *dd, *dt, *db;
    (void)hipMalloc(&dx, 2048); (void)hipMalloc(&dd, 2048); (void)hipMalloc(&dt, 2048); (void)hipMalloc(&db, 2048);
    (void)hipMemcpy(dx, hx, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dd, dt, db);
    (void)hipMemcpy(hd, dd, 2048, hipMemcpyDeviceToHost); (void)hipMemcpy(ht, dt, 2048, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hb, db, 2048, hipMemcpyDeviceToHost);
    int bad = 0, badBack = 0;
    for (int i = 0; i < 256; ++i) bad += hd[i] != ht[i];
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) badBack += hb[l * 4 + i] != hx[(l & 15) * 16 + 4 * (l >> 4) + i];
    std::printf("transpose vs direct loads: %d mismatches; transposed back vs load order: %d mismatches\n", bad, badBack);
    if (bad) for (int l = 0; l < 64; l += 16) std::printf(" lane %2d direct %.0f %.0f %.0f %.0f  via %.0f %.0f %.0f %.0f\n", l, hd[l*4], hd[l*4+1], hd[l*4+2], hd[l*4+3], ht[l*4], ht[l*4+1], ht[l*4+2], ht[l*4+3]);
    return bad || badBack;
}
