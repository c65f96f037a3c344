// hbm_read.hip -- what a plain two-stream read reaches on this part: the practical ceiling for the one-block-per-call MAC,
// which streams an FDL row and an IR row per partition step (2.18 GB per launch at the bench configuration).
// Every lane reads 16 B from each of two arrays per iteration and folds them into one FMA chain; block-strided so that a
// wave reads 1 KB contiguous per array and iteration, like the kernels do.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 hbm_read.hip -o /tmp/hbr && /tmp/hbr
#include <hip/hip_runtime.h>
#include <cstdio>

template <int UNROLL>
__global__ __launch_bounds__(256) void k_read2(const double2* __restrict__ a, const double2* __restrict__ b, size_t n,
                                               double* __restrict__ sink)
{
    double acc0 = 0.0, acc1 = 0.0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        double2 x[UNROLL], y[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { x[u] = a[i + u * stride]; y[u] = b[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { acc0 = fma(x[u].x, y[u].x, acc0); acc1 = fma(x[u].y, y[u].y, acc1); }
    }
    for (; i < n; i += stride) { acc0 = fma(a[i].x, b[i].x, acc0); acc1 = fma(a[i].y, b[i].y, acc1); }
    if (acc0 + acc1 == 1.2345e300) sink[0] = acc0;      // never true: keeps the loads alive
}

template <int UNROLL>
static void run(const double2* a, const double2* b, size_t n, double* sink, int blocks)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_read2<UNROLL>, dim3(blocks), dim3(256), 0, 0, a, b, n, sink);
    (void)hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read2<UNROLL>, dim3(blocks), dim3(256), 0, 0, a, b, n, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 2.0 * n * sizeof(double2);
    std::printf("unroll %d, %5d blocks: %.3f ms per launch, %.0f GB/s\n", UNROLL, blocks, ms / reps, bytes / (ms / reps) * 1e-6);
}

int main()
{
    const size_t n = (size_t)68 * 1024 * 1024;          // 2 x 1.09 GB
    double2 *a, *b; double* sink;
    if (hipMalloc(&a, n * sizeof(double2)) != hipSuccess || hipMalloc(&b, n * sizeof(double2)) != hipSuccess) return 1;
    (void)hipMalloc(&sink, 8);
    (void)hipMemset(a, 0, n * sizeof(double2)); (void)hipMemset(b, 0, n * sizeof(double2));
    for (int blocks : { 2048, 4096, 16384 }) { run<4>(a, b, n, sink, blocks); run<8>(a, b, n, sink, blocks); }
    return 0;
}
