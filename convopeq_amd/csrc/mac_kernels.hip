// mac_kernels.hip -- frequency-domain delay line (FDL) multiply-accumulate for gfx950.
//
//   Y[c][t][bin] = sum_{k<K} X[c][t-k][bin] * H[ir(c)][k][bin]
//
// Replaces the per-partition accumulateSplitComplex loop of the reference (src/MKLNonUniformConvolver.cpp:
// 150-195 called from :1293-1308 and :1505-1520), batched over channels and over the T blocks of one
// process call.  One lane owns one frequency bin of one channel and keeps TT consecutive output blocks
// in registers: per partition step it loads ONE FDL row element and ONE IR row element (16 B each,
// 1 KB coalesced per wave) and performs TT complex MACs against a sliding register window of the FDL,
// i.e. the HBM stream per output block shrinks by TT versus the one-block-per-pass schedule
// (SURVEY.md finding 10 / section 8(d)).  No MFMA: per bin this is a Toeplitz matrix-vector product, not a
// dense contraction; fp64 vector FMA only.
//
// Accumulation order: ascending IR partition index k (the reference walks its reversed partition array
// over the same pairs, :959-985,:1291-1308); FMA instead of the reference's mul/add.
#include "kernels.hpp"

namespace cpq {

namespace {

template <int TT, int PF>
__global__ __launch_bounds__(256) void k_fdl_mac(const double2* __restrict__ X, const double2* __restrict__ H,
                                                 const int* __restrict__ irSlot, double2* __restrict__ Y,
                                                 int nPairs, int kPad, int ringMask, int head, int T, int nTiles,
                                                 int64_t hSlotStride)
{
    static_assert(TT % PF == 0, "prefetch depth must divide the tile");
    // XCD-aware decomposition: blocks b and b+8 share an XCD (and its L2) under round-robin dispatch, so the
    // nTiles time tiles of one (channel, half-spectrum) pair -- which re-read the same IR rows and
    // overlapping FDL rows -- are made consecutive members of one residue class mod 8.
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    const int q = bid >> 3;
    const int tile = q % nTiles;
    const int pair = (q / nTiles) * 8 + xcd;
    if (pair >= nPairs) return;
    const int c = pair >> 1;
    const int bin = (pair & 1) * 256 + threadIdx.x;
    const int t0 = tile * TT;

    const double2* __restrict__ Xc = X + (int64_t)c * (ringMask + 1) * kP + bin;
    const double2* __restrict__ Hc = H + (int64_t)irSlot[c] * hSlotStride + bin;
    const int base = head + t0;

    double2 acc[TT], xw[TT], xn[PF], hn[PF];
#pragma unroll
    for (int u = 0; u < TT; ++u) {
        acc[u] = make_double2(0.0, 0.0);
        xw[u] = Xc[(int64_t)((base + u) & ringMask) * kP];          // window: X[t0 + u - k] at (u - k) mod TT
    }
#pragma unroll
    for (int r = 0; r < PF; ++r) {
        xn[r] = Xc[(int64_t)((base - r - 1) & ringMask) * kP];
        hn[r] = Hc[(int64_t)r * kP];
    }

    for (int k0 = 0; k0 < kPad; k0 += TT) {
#pragma unroll
        for (int r = 0; r < TT; ++r) {
            const int k = k0 + r;
            const double2 h = hn[r % PF];
            const double2 xnew = xn[r % PF];
            hn[r % PF] = Hc[(int64_t)(k + PF) * kP];                                   // IR row k+PF (zero rows past K)
            xn[r % PF] = Xc[(int64_t)((base - (k + PF) - 1) & ringMask) * kP];          // FDL row entering at step k+PF
#pragma unroll
            for (int i = 0; i < TT; ++i) {
                const double2 x = xw[(i - r + TT) % TT];
                acc[i].x = fma(x.x, h.x, fma(-x.y, h.y, acc[i].x));
                acc[i].y = fma(x.x, h.y, fma(x.y, h.x, acc[i].y));
            }
            xw[TT - 1 - r] = xnew;    // X[t0 + TT-1 - k] retires, X[t0 - k - 1] takes its place
        }
    }
#pragma unroll
    for (int i = 0; i < TT; ++i)
        if (t0 + i < T) Y[((int64_t)c * T + t0 + i) * kP + bin] = acc[i];
}

// Packed bin 0 holds (DC, Nyquist): two independent real MACs per (channel, block).
__global__ __launch_bounds__(256) void k_fdl_mac_dcnyq(const double2* __restrict__ XDN,
                                                       const double2* __restrict__ HDN,
                                                       const int* __restrict__ irSlot, double2* __restrict__ Y,
                                                       int nCh, int K, int ringMask, int head, int T, int hdnStride)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nCh * T) return;
    const int c = idx / T;
    const int t = idx - c * T;
    const double2* __restrict__ x = XDN + (int64_t)c * (ringMask + 1);
    const double2* __restrict__ h = HDN + (int64_t)irSlot[c] * hdnStride;
    double dc = 0.0, ny = 0.0;
    for (int k = 0; k < K; ++k) {
        const double2 xv = x[(head + t - k) & ringMask];
        const double2 hv = h[k];
        dc = fma(xv.x, hv.x, dc);
        ny = fma(xv.y, hv.y, ny);
    }
    Y[(int64_t)idx * kP] = make_double2(dc, ny);
}

template <int TT>
void launch_mac_t(hipStream_t stream, const double2* X, const double2* H, const int* irSlot, double2* Y, int nCh,
                  int kPad, int ringSlots, int head, int T, int64_t hSlotStride)
{
    constexpr int PF = (TT < kMacPrefetch) ? TT : kMacPrefetch;
    const int nPairs = nCh * 2;
    const int nTiles = (T + TT - 1) / TT;
    const int groups = (nPairs + 7) / 8;
    const int grid = groups * nTiles * 8;
    hipLaunchKernelGGL((k_fdl_mac<TT, PF>), dim3(grid), dim3(256), 0, stream, X, H, irSlot, Y, nPairs, kPad,
                       ringSlots - 1, head, T, nTiles, hSlotStride);
}

}  // namespace

void launch_fdl_mac(hipStream_t stream, int tile, const double2* X, const double2* H, const int* irSlot, double2* Y,
                    int nCh, int kPad, int ringSlots, int head, int T, int64_t hSlotStride)
{
    switch (tile) {
        case 4:  launch_mac_t<4>(stream, X, H, irSlot, Y, nCh, kPad, ringSlots, head, T, hSlotStride); break;
        case 8:  launch_mac_t<8>(stream, X, H, irSlot, Y, nCh, kPad, ringSlots, head, T, hSlotStride); break;
        default: launch_mac_t<16>(stream, X, H, irSlot, Y, nCh, kPad, ringSlots, head, T, hSlotStride); break;
    }
}

void launch_fdl_mac_dcnyq(hipStream_t stream, const double2* XDN, const double2* HDN, const int* irSlot, double2* Y,
                          int nCh, int K, int ringSlots, int head, int T, int hdnStride)
{
    const int total = nCh * T;
    hipLaunchKernelGGL(k_fdl_mac_dcnyq, dim3((total + 255) / 256), dim3(256), 0, stream, XDN, HDN, irSlot, Y, nCh, K,
                       ringSlots - 1, head, T, hdnStride);
}

}  // namespace cpq
