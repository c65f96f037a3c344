// pade_div_check.hip -- how many Newton steps does the fastTanh Pade division need after v_rcp_f64 to equal the
// IEEE quotient?  den in [27, 209.25], |num| <= 212.7 (svf_kernels.hip: pade_div).  Counts mismatches against the
// compiler's correctly rounded division over random operands and over operands generated the way the kernel does.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off pade_div_check.hip -o /tmp/pdc && /tmp/pdc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ double div2(double num, double den)
{
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    r = fma(fma(-den, r, 1.0), r, r);
    const double q = num * r;
    return fma(fma(-den, q, num), r, q);
}
__device__ __forceinline__ double div1(double num, double den)
{
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.0), r, r);
    const double q = num * r;
    return fma(fma(-den, q, num), r, q);
}
__device__ __forceinline__ uint64_t mix(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void k(unsigned long long* bad, double* maxRcpErr, int iters)
{
    const uint64_t gid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    unsigned long long b1 = 0, b2 = 0, b1k = 0, b2k = 0;
    double worst = 0.0, w0 = 0.0, w1 = 0.0;
    for (int i = 0; i < iters; ++i) {
        const uint64_t u = mix(gid * 1000003ull + i), w = mix(u);
        const double den = 27.0 + (double)(u >> 11) * (1.0 / 9007199254740992.0) * 182.25;
        const double num = ((double)(w >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0) * 212.7;
        const double q = num / den;
        b2 += (div2(num, den) != q);
        b1 += (div1(num, den) != q);
        // operands as the kernel forms them: x in [-4.5, 4.5]
        const double x = ((double)(w >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0) * 4.5;
        const double x2 = x * x, n2 = x * (27.0 + x2), d2 = 27.0 + 9.0 * x2;
        const double qk = n2 / d2;
        b2k += (div2(n2, d2) != qk);
        b1k += (div1(n2, d2) != qk);
        const double r = __builtin_amdgcn_rcp(den);
        const double e = fabs(fma(-den, r, 1.0));
        worst = e > worst ? e : worst;
        // cheaper forms, relative error against the IEEE quotient: q0 = num * rcp(den); q1 = num * (one Newton step)
        const double r1 = fma(fma(-den, r, 1.0), r, r);
        const double e0 = qk != 0.0 ? fabs((n2 * __builtin_amdgcn_rcp(d2) - qk) / qk) : 0.0;
        const double rk = __builtin_amdgcn_rcp(d2);
        const double e1 = qk != 0.0 ? fabs((n2 * fma(fma(-d2, rk, 1.0), rk, rk) - qk) / qk) : 0.0;
        w0 = e0 > w0 ? e0 : w0;
        w1 = e1 > w1 ? e1 : w1;
        (void)r1;
    }
    atomicMax(reinterpret_cast<unsigned long long*>(maxRcpErr + 1), (unsigned long long)__double_as_longlong(w0));
    atomicMax(reinterpret_cast<unsigned long long*>(maxRcpErr + 2), (unsigned long long)__double_as_longlong(w1));
    atomicAdd(&bad[0], b2); atomicAdd(&bad[1], b1); atomicAdd(&bad[2], b2k); atomicAdd(&bad[3], b1k);
    // max over threads (values are non-negative: integer compare is monotone)
    atomicMax(reinterpret_cast<unsigned long long*>(maxRcpErr), (unsigned long long)__double_as_longlong(worst));
}
int main()
{
    unsigned long long* bad; double* err;
    hipMalloc(&bad, 4 * sizeof(unsigned long long)); hipMemset(bad, 0, 4 * sizeof(unsigned long long));
    hipMalloc(&err, 3 * sizeof(double)); hipMemset(err, 0, 3 * sizeof(double));
    const int blocks = 4096, threads = 256, iters = 2048;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, bad, err, iters);
    unsigned long long h[4]; double e[3];
    hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(e, err, sizeof(e), hipMemcpyDeviceToHost);
    printf("samples %.3g  mismatches: two-step %llu, one-step %llu (random); two-step %llu, one-step %llu (kernel operands); max |1 - den*rcp(den)| = %.3g; max relative error of num*rcp(den) %.3g, of num*(rcp + one Newton step) %.3g (kernel operands)\n",
           (double)blocks * threads * iters, h[0], h[1], h[2], h[3], e[0], e[1], e[2]);
    return 0;
}
