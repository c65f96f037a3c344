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

// NT: streaming (non-temporal) loads -- 1: FDL rows, 2: FDL and IR rows.  Only for calls of a single time tile, where no
// other workgroup reads the same rows (IR rows: only when every channel has its own IR): one-block calls 0.360 -> 0.344 ms.
template <int TT, int PF, int NT = 0>
__global__ __launch_bounds__(256, (TT >= 32 || (TT == 16 && PF >= 16)) ? 1 : 2) void k_fdl_mac(const double2* __restrict__ X, const double2* __restrict__ H,
                                                 const int* __restrict__ irSlot, double2* __restrict__ Y,
                                                 int nPairs, int kPad, int ringMask, int head, int T, int nTiles,
                                                 int64_t hSlotStride, int P, int segShift, int K)
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
    // a block covers one segment of blockDim.x (<= 256) bins; P >> segShift... segments per spectrum = 1 << segShift
    const int c = pair >> segShift;
    const int bin = (pair & ((1 << segShift) - 1)) * blockDim.x + threadIdx.x;
    const int t0 = tile * TT;

    const double2* __restrict__ Xc = X + (int64_t)c * (ringMask + 1) * P + bin;
    const double2* __restrict__ Hc = H + (int64_t)irSlot[c] * hSlotStride + bin;
    const int base = head + t0;

    // Small tiles (short, HBM-bound calls) keep the two products of the real part apart: Re = p - q with p = sum a c,
    // q = sum b d.  Element 0 of a packed spectrum is (DC, Nyquist), two independent real MACs -- exactly (p, q) -- so
    // these variants need no separate DC/Nyquist kernel.  Larger tiles keep the fused form (registers).
    constexpr bool kSplitRe = TT <= 8;
    double2 acc[TT], xw[TT], xn[PF], hn[PF];
    double accq[kSplitRe ? TT : 1];
#pragma unroll
    for (int u = 0; u < TT; ++u) {
        acc[u] = make_double2(0.0, 0.0);
        if (kSplitRe) accq[u] = 0.0;
        xw[u] = Xc[(int64_t)((base + u) & ringMask) * P];          // window: X[t0 + u - k] at (u - k) mod TT
    }
#pragma unroll
    for (int r = 0; r < PF; ++r) {
        xn[r] = Xc[(int64_t)((base - r - 1) & ringMask) * P];
        hn[r] = Hc[(int64_t)r * P];               // rows >= K of a slot are zero
    }

    for (int k0 = 0; k0 < kPad; k0 += TT) {
#pragma unroll
        for (int r = 0; r < TT; ++r) {
            const int k = k0 + r;
            const double2 h = hn[r % PF];
            const double2 xnew = xn[r % PF];
            // rows that only steps >= K would consume are not fetched (at K = 33, tile 4: 33 + 36 instead of 40 + 43 rows
            // per output tile): IR row k + PF serves step k + PF, the FDL row entering at step k + PF serves k + PF + 1 on
            if (k + PF < K) {
                if (NT >= 2) {
                    const double* hp = reinterpret_cast<const double*>(Hc + (int64_t)(k + PF) * P);
                    hn[r % PF] = make_double2(__builtin_nontemporal_load(hp), __builtin_nontemporal_load(hp + 1));
                } else {
                    hn[r % PF] = Hc[(int64_t)(k + PF) * P];                               // IR row k+PF
                }
            }
            if (k + PF + 1 < K) {
                if (NT >= 1) {
                    const double* xp = reinterpret_cast<const double*>(Xc + (int64_t)((base - (k + PF) - 1) & ringMask) * P);
                    xn[r % PF] = make_double2(__builtin_nontemporal_load(xp), __builtin_nontemporal_load(xp + 1));
                } else {
                    xn[r % PF] = Xc[(int64_t)((base - (k + PF) - 1) & ringMask) * P];      // FDL row entering at step k+PF
                }
            }
            // keep the two loads HERE: without the fence the scheduler sinks them next to their use PF steps
            // later (to save registers) and the kernel runs with <= 3 loads in flight per wave
            __builtin_amdgcn_sched_barrier(0);
            if (k < K)                        // uniform; false only in the last pass (kPad - K < TT steps)
#pragma unroll
            for (int i = 0; i < TT; ++i) {
                const double2 x = xw[(i - r + TT) % TT];
                if (kSplitRe) {
                    acc[i].x = fma(x.x, h.x, acc[i].x);
                    accq[i] = fma(x.y, h.y, accq[i]);
                } else {
                    acc[i].x = fma(x.x, h.x, fma(-x.y, h.y, acc[i].x));
                }
                acc[i].y = fma(x.x, h.y, fma(x.y, h.x, acc[i].y));
            }
            xw[TT - 1 - r] = xnew;    // X[t0 + TT-1 - k] retires, X[t0 - k - 1] takes its place
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int i = 0; i < TT; ++i)
        if (t0 + i < T) {
            double2 y = acc[i];
            if (kSplitRe) y = (bin == 0) ? make_double2(acc[i].x, accq[i]) : make_double2(acc[i].x - accq[i], acc[i].y);
            Y[((int64_t)c * T + t0 + i) * P + bin] = y;
        }
}

// ---------------------------------------------------------------------------------------------------------
// Workgroup-cooperative variant for calls of >= 48 blocks.
//
// At that batching the kernel is bound by the fp64 FMA rate, not by HBM (8 K T flop per 16 (2K + T) bytes), so the
// complex MAC is done with THREE real FMAs (Gauss): M1 += a c, M2 += b d, M3 += (a + b)(c + d), Re = M1 - M2,
// Im = M3 - M1 - M2 at the end.  The X-side sum a + b is formed once when a row enters the register window, the
// H-side sum c + d once per IR row for the lane's 8 outputs: 26 VALU operations per partition step instead of 32.
// Rounding differs from the 4-multiply form by the usual Gauss bound (normwise the same order; measured RMS error
// against the oracle unchanged at 4.9e-16).  The register-tile kernel above (short, HBM-bound calls) keeps 4 FMAs.
//
// fp64 FMA needs >= 4 waves per SIMD to approach its issue rate on gfx950 (measured: 1 wave 25-45, 2 waves
// ~50, 4 waves ~60 TFLOP/s), i.e. <= 128 VGPRs per lane, which caps the register tile at 8 outputs per lane;
// a tile that small re-reads every FDL and IR row 8x more often than a 64-output tile.  So eight waves of one
// workgroup take the eight consecutive 8-block tiles of the SAME (channel, 64-bin column) and share the FDL rows
// through LDS: wave w needs FDL row (t0 + 8w - k - 1) at partition step k, which wave w-1 needs eight steps
// later, so each row is fetched from HBM once, parked in a 9-block LDS ring (72 KB, lane-linear rows: every
// ds_read/ds_write_b128 is conflict-free) and read by the eight waves at eight different times.  One barrier per
// 8 steps (plus one for the IR rows, which are staged the same way: each wave fetches one IR row of the next chunk).
// HBM traffic per launch = every needed FDL row, IR row and output row exactly once.
constexpr int kWgWaves = 8;
constexpr int kWgTile = 8;                       // outputs per lane
constexpr int kWgRingBlocks = kWgWaves + 1;      // blocks of 8 rows
static_assert(kWgWaves == kWgTile, "k_fdl_mac_wg fetches block -1 one row per wave: waves per workgroup == rows per block");

// The IR rows of a chunk are staged through LDS one chunk ahead (one row per wave, like the FDL rows): a prefetch distance
// of 8+ partition steps without holding them in registers (a 4-deep register prefetch measured 7 % slower, 2-deep 35 %).
__global__ __launch_bounds__(64 * kWgWaves, 4) void k_fdl_mac_wg(const double2* __restrict__ X,
                                                                  const double2* __restrict__ H,
                                                                  const int* __restrict__ irSlot,
                                                                  double2* __restrict__ Y, int kPad, int ringMask,
                                                                  int head, int T, int nGroups, int64_t hSlotStride,
                                                                  int P, int nCols, int nWork, int K)
{
    __shared__ double2 ring[kWgRingBlocks * kWgTile * 64];
    __shared__ double2 hst[kWgTile * 64];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // wave-uniform: the loop below branches on it
    // workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8): give each XCD a contiguous range of logical ids so
    // that the nGroups workgroups of one (channel, column) -- same IR rows, overlapping FDL rows -- share one L2
    const int perXcd = gridDim.x >> 3;                       // the grid is padded to a multiple of 8
    const int logical = (blockIdx.x & 7) * perXcd + (blockIdx.x >> 3);
    if (logical >= nWork) return;                            // uniform per workgroup: no barrier is skipped by part of it
    const int grp = logical % nGroups;
    const int cg = logical / nGroups;
    const int c = cg / nCols;
    const int bin = (cg - c * nCols) * 64 + lane;
    const int base = head + grp * (kWgWaves * kWgTile);          // FDL slot of the group's first output block
    const int t0w = grp * (kWgWaves * kWgTile) + w * kWgTile;    // this wave's first output block

    // uniform row bases (SGPR) + per-lane bin offset: the loads use the scalar-base addressing form
    const double2* __restrict__ Xu = X + (int64_t)c * (ringMask + 1) * P;
    const double2* __restrict__ Hu = H + (int64_t)irSlot[c] * hSlotStride;
    auto slotOf = [](int b) { return ((b % kWgRingBlocks) + kWgRingBlocks) % kWgRingBlocks; };
    auto xrow = [&](int slot) { return Xu + (int64_t)(slot & ringMask) * P; };
    auto hrow = [&](int k) { return Hu + (int64_t)k * P; };

    // three real products per complex MAC (Gauss): M1 += a c, M2 += b d, M3 += (a + b)(c + d);
    // Re = M1 - M2, Im = M3 - M1 - M2.  The X-side sums live next to the register window, the H-side sum is formed
    // once per IR row and serves the 8 outputs of the lane.
    double m1[kWgTile], m2[kWgTile], m3[kWgTile], xsum[kWgTile];
    double2 xw[kWgTile];
#pragma unroll
    for (int u = 0; u < kWgTile; ++u) {
        m1[u] = 0.0; m2[u] = 0.0; m3[u] = 0.0;
        xw[u] = xrow(base + kWgTile * w + u)[bin];
        xsum[u] = xw[u].x + xw[u].y;
    }
    // ring prologue: block b (rows base+8b .. base+8b+7) for b = 0..6 is the register window of wave b;
    // block -1 is fetched one row per wave
    if (w < kWgWaves - 1) {
#pragma unroll
        for (int u = 0; u < kWgTile; ++u) ring[(slotOf(w) * kWgTile + u) * 64 + lane] = xw[u];
    }
    {
        const double2 xs = xrow(base - kWgTile + w)[bin];
        ring[(slotOf(-1) * kWgTile + w) * 64 + lane] = xs;
    }
    hst[w * 64 + lane] = hrow(w)[bin];            // IR rows of chunk 0
    __syncthreads();

    // The last chunk holds rem = K - 8 (nChunks - 1) partition steps (1 ... 8): the steps past K are skipped, and so
    // are the loads that only they would consume -- IR rows >= K, and the FDL rows that enter the register windows
    // after step rem - 1 (row 7 - r of a ring block refills a window slot at step r for use from step r + 1 on).  At
    // K = 33 (131072 taps at P = 4096) that is 7 of 40 steps and 15 of 144 row reads per 64 output rows.
    const int nChunks = kPad / kWgTile;
    const int rem = K - (nChunks - 1) * kWgTile;
    const bool active = t0w < T;                  // wave-uniform
    for (int j = 0; j < nChunks; ++j) {
        const bool lastChunk = (j == nChunks - 1), nextIsLast = (j == nChunks - 2);
        const int steps = lastChunk ? rem : kWgTile;
        // row w of block (-j-2): needed by wave 0 in the next chunk; staged in a register, parked at chunk end
        // the FDL rows of a channel are read by this workgroup alone (one group at T <= 64): streaming hint; the IR rows
        // may be shared by every channel (CPQ_ALL_STREAMS) and stay cacheable
        double2 xs = make_double2(0.0, 0.0), hsn = make_double2(0.0, 0.0);
        if (!lastChunk && (!nextIsLast || w > kWgTile - rem)) {
            const double* xp = reinterpret_cast<const double*>(xrow(base + kWgTile * (-j - 2) + w) + bin);
            xs = make_double2(__builtin_nontemporal_load(xp), __builtin_nontemporal_load(xp + 1));
        }
        if (!lastChunk && (!nextIsLast || w < rem))
            hsn = hrow((j + 1) * kWgTile + w)[bin];               // IR row w of the next chunk
        const double2* blk = ring + slotOf(w - j - 1) * kWgTile * 64 + lane;
        // a wave whose 8 outputs lie beyond T only feeds the ring (partial last group: T mod 64 != 0)
        if (active)
#pragma unroll
        for (int r = 0; r < kWgTile; ++r) {
            if (r < steps) {                      // uniform; false only in the last chunk
            const double2 h = hst[r * 64 + lane];
            const double hs = h.x + h.y;
            // the slot refilled in the previous step (window slot 8-r, first used now): its sum
            if (r > 0) xsum[kWgTile - r] = xw[kWgTile - r].x + xw[kWgTile - r].y;
            // output 7 is the last user of window slot 7-r: retire it first, then refill the slot straight from
            // the LDS ring (X[t0w - k - 1]); the read's latency hides behind the other seven MACs
            {
                const double2 x = xw[kWgTile - 1 - r];
                m1[kWgTile - 1] = fma(x.x, h.x, m1[kWgTile - 1]);
                m2[kWgTile - 1] = fma(x.y, h.y, m2[kWgTile - 1]);
                m3[kWgTile - 1] = fma(xsum[kWgTile - 1 - r], hs, m3[kWgTile - 1]);
            }
            xw[kWgTile - 1 - r] = blk[(kWgTile - 1 - r) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < kWgTile - 1; ++i) {
                const int sl = (i - r + kWgTile) % kWgTile;
                const double2 x = xw[sl];
                m1[i] = fma(x.x, h.x, m1[i]);
                m2[i] = fma(x.y, h.y, m2[i]);
                m3[i] = fma(xsum[sl], hs, m3[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (lastChunk) break;
        // slot of block (-j-2) == slot of block (7-j), last read by wave 7 in chunk j-1: free since the barrier
        // that ended that chunk
        ring[(slotOf(-j - 2) * kWgTile + w) * 64 + lane] = xs;
        // slot 0 was refilled in the last step of this chunk
        xsum[0] = xw[0].x + xw[0].y;
        __syncthreads();                      // every wave is done with this chunk's IR rows
        hst[w * 64 + lane] = hsn;
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < kWgTile; ++i)
        if (t0w + i < T) {
            // element 0 packs (DC, Nyquist), two independent REAL MACs: exactly M1 and M2 of the Gauss form, so this
            // kernel needs no separate DC/Nyquist pass
            const double2 y = (bin == 0) ? make_double2(m1[i], m2[i]) : make_double2(m1[i] - m2[i], (m3[i] - m1[i]) - m2[i]);
            // streaming store: the row is not touched again before the inverse FFT (0.724 -> 0.717 ms)
            double* yp = reinterpret_cast<double*>(Y + ((int64_t)c * T + t0w + i) * P + bin);
            __builtin_nontemporal_store(y.x, yp);
            __builtin_nontemporal_store(y.y, yp + 1);
        }
}

// Packed bin 0 holds (DC, Nyquist): two independent real MACs per (channel, block).  One wave per (channel,
// block): lanes stride over the partitions (coalesced reads of the compact XDN / HDN rows), then a wave reduction.
__global__ __launch_bounds__(256) void k_fdl_mac_dcnyq(const double2* __restrict__ XDN,
                                                       const double2* __restrict__ HDN,
                                                       const int* __restrict__ irSlot, double2* __restrict__ Y,
                                                       int nCh, int K, int ringMask, int head, int T, int hdnStride, int P)
{
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // (channel, block) of this wave
    if (idx >= nCh * T) return;
    const int c = idx / T;
    const int t = idx - c * T;
    const double2* __restrict__ x = XDN + (int64_t)c * (ringMask + 1);
    const double2* __restrict__ h = HDN + (int64_t)irSlot[c] * hdnStride;
    double dc = 0.0, ny = 0.0;
    for (int k = lane; k < K; k += 64) {
        const double2 xv = x[(head + t - k) & ringMask];
        const double2 hv = h[k];
        dc = fma(xv.x, hv.x, dc);
        ny = fma(xv.y, hv.y, ny);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        dc += __shfl_down(dc, off);
        ny += __shfl_down(ny, off);
    }
    if (lane == 0) Y[(int64_t)idx * P] = make_double2(dc, ny);
}

template <int TT, int PF>
void launch_mac_t(hipStream_t stream, const double2* X, const double2* H, const int* irSlot, double2* Y, int P,
                  int nCh, int kPad, int K, int ringSlots, int head, int T, int64_t hSlotStride, bool hPrivate = false)
{
    const int threads = P < 256 ? P : 256;
    int segShift = 0;
    while ((threads << segShift) < P) ++segShift;
    const int nPairs = nCh << segShift;
    const int nTiles = (T + TT - 1) / TT;
    const int groups = (nPairs + 7) / 8;
    const int grid = groups * nTiles * 8;
    if constexpr (TT <= 8) {
        if (nTiles == 1) {              // every row is read by exactly one workgroup: streaming loads
            if (hPrivate)
                hipLaunchKernelGGL((k_fdl_mac<TT, PF, 2>), dim3(grid), dim3(threads), 0, stream, X, H, irSlot, Y, nPairs, kPad,
                                   ringSlots - 1, head, T, nTiles, hSlotStride, P, segShift, K);
            else
                hipLaunchKernelGGL((k_fdl_mac<TT, PF, 1>), dim3(grid), dim3(threads), 0, stream, X, H, irSlot, Y, nPairs, kPad,
                                   ringSlots - 1, head, T, nTiles, hSlotStride, P, segShift, K);
            return;
        }
    }
    hipLaunchKernelGGL((k_fdl_mac<TT, PF>), dim3(grid), dim3(threads), 0, stream, X, H, irSlot, Y, nPairs, kPad,
                       ringSlots - 1, head, T, nTiles, hSlotStride, P, segShift, K);
}

}  // namespace

// Variant for `tile` (0 = automatic) and T output rows per channel: 0 = the workgroup-cooperative kernel (64 outputs
// per workgroup: waves past the last row only feed the LDS ring; worth it from ~48 rows on (below, the register tiles measured faster)), else the register tile -- sized so that short calls do not spend most
// of their FMAs on rows that are not there (T = 1, the reference's own call pattern, is HBM-bound at any tile).
int fdl_mac_variant(int tile, int T)
{
    if (tile) return tile;
    if (T >= 48) return 0;
    if (T >= 12) return 16;
    if (T >= 6) return 8;
    return 4;
}

// the variants that leave packed bin 0 to launch_fdl_mac_dcnyq
bool fdl_mac_needs_dcnyq(int tile, int T) { return fdl_mac_variant(tile, T) > 8; }

int fdl_mac_kpad_align(int tile, int T)
{
    const int v = fdl_mac_variant(tile, T);
    return v == 0 ? kWgTile : v;
}

void launch_fdl_mac(hipStream_t stream, int tile, const double2* X, const double2* H, const int* irSlot, double2* Y,
                    int P, int nCh, int K, int ringSlots, int head, int T, int64_t hSlotStride, bool hPrivate)
{
    // K = partitions in use; the kernels walk it in steps of their tile (rows K ... kPad - 1 of every IR slot are zero)
    const int kAlign = fdl_mac_kpad_align(tile, T);
    const int kPad = (K + kAlign - 1) / kAlign * kAlign;
    tile = fdl_mac_variant(tile, T);
    if (tile == 0) {      // long calls: workgroup-cooperative kernel
        const int nGroups = (T + kWgWaves * kWgTile - 1) / (kWgWaves * kWgTile);
        const int nCols = P / 64;
        const int nWork = nCh * nCols * nGroups;
        hipLaunchKernelGGL(k_fdl_mac_wg, dim3((nWork + 7) / 8 * 8), dim3(64 * kWgWaves), 0, stream, X, H,
                           irSlot, Y, kPad, ringSlots - 1, head, T, nGroups, hSlotStride, P, nCols, nWork, K);
        return;
    }
    const int pf = 4;     // prefetch depth in partition steps (deeper measured slower: register pressure)
#define CPQ_MAC_CASE(TT_, PF_) launch_mac_t<TT_, PF_>(stream, X, H, irSlot, Y, P, nCh, kPad, K, ringSlots, head, T, hSlotStride, hPrivate)
    switch (tile) {
        case 4:  CPQ_MAC_CASE(4, 4); break;
        case 8:  if (pf >= 8) CPQ_MAC_CASE(8, 8); else CPQ_MAC_CASE(8, 4); break;
        case 32: if (pf >= 16) CPQ_MAC_CASE(32, 16); else if (pf >= 8) CPQ_MAC_CASE(32, 8); else CPQ_MAC_CASE(32, 4); break;
        default: if (pf >= 16) CPQ_MAC_CASE(16, 16); else if (pf >= 8) CPQ_MAC_CASE(16, 8); else CPQ_MAC_CASE(16, 4); break;
    }
#undef CPQ_MAC_CASE
}

void launch_fdl_mac_dcnyq(hipStream_t stream, const double2* XDN, const double2* HDN, const int* irSlot, double2* Y,
                          int P, int nCh, int K, int ringSlots, int head, int T, int hdnStride)
{
    const int total = nCh * T;          // one wave each, 4 waves per block
    hipLaunchKernelGGL(k_fdl_mac_dcnyq, dim3((total + 3) / 4), dim3(256), 0, stream, XDN, HDN, irSlot, Y, nCh, K,
                       ringSlots - 1, head, T, hdnStride, P);
}

}  // namespace cpq
