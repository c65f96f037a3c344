// fft_kernels.hip -- wave-level fp64 real FFT kernels for gfx950 (CDNA4, wave64).
//
// One wavefront computes one 1024-point real transform as a 512-point complex FFT: 8 complex points
// per lane in registers, three radix-8 passes, two exchanges through 9 KB of LDS whose strides (72 and
// 66 complex) keep every ds_read_b128/ds_write_b128 lane group on distinct banks, then the real-FFT
// split.  The overlap-save framing of the reference (copy [prev|cur], src/MKLNonUniformConvolver.cpp:
// 1256-1258), its CCS de-interleave (:132-139, :1270-1283) and the output half selection (:1332) are
// fused into the loads/stores, so a frame is read once from HBM and a spectrum written once.
//
// Replaces ippsFFTFwd_RToCCS_64f / ippsFFTInv_CCSToR_64f (src/FFTBackend.cpp:123-150) with the same
// scaling convention (IPP_FFT_DIV_INV_BY_N: forward unscaled, inverse 1/N, :33-35).
//
// Spectrum layout ("packed"): 512 complex per transform; bin k (1..511) = X[k]; bin 0 = (X[0], X[512])
// (DC and Nyquist are both real), so a spectrum is exactly 8 KB and a wave stores 1 KB per instruction.
#include "kernels.hpp"

namespace cpq {

namespace {

typedef double v2d __attribute__((ext_vector_type(2)));      // 16-byte streaming (non-temporal) loads / stores

__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }

// a * w (INV = false) or a * conj(w) (INV = true)
__device__ __forceinline__ double2 cmul(double2 a, double2 b)
{
    return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}

// exp(-2 pi i q / 16), q < 8 (correctly rounded constants)
__device__ __forceinline__ double2 w16(int q)
{
    constexpr double c1 = 0.92387953251128673848, s1 = 0.38268343236508978178, r = 0.70710678118654752440;
    switch (q) {
        case 0: return make_double2(1.0, 0.0);
        case 1: return make_double2(c1, -s1);
        case 2: return make_double2(r, -r);
        case 3: return make_double2(s1, -c1);
        case 4: return make_double2(0.0, -1.0);
        case 5: return make_double2(-s1, -c1);
        case 6: return make_double2(-r, -r);
        default: return make_double2(-c1, -s1);
    }
}

template <bool INV>
__device__ __forceinline__ double2 cmulw(double2 a, double2 w)
{
    if (INV) return make_double2(fma(a.x, w.x, a.y * w.y), fma(a.y, w.x, -(a.x * w.y)));
    return make_double2(fma(a.x, w.x, -(a.y * w.y)), fma(a.x, w.y, a.y * w.x));
}

// 4-point DFT of (t0..t3) -> (t0..t3), natural order
template <bool INV>
__device__ __forceinline__ void dft4(double2& t0, double2& t1, double2& t2, double2& t3)
{
    const double2 e0 = cadd(t0, t2), e1 = csub(t0, t2);
    const double2 o0 = cadd(t1, t3), d = csub(t1, t3);
    const double2 o1 = INV ? make_double2(-d.y, d.x) : make_double2(d.y, -d.x);   // * (+i) / * (-i)
    t0 = cadd(e0, o0);
    t1 = cadd(e1, o1);
    t2 = csub(e0, o0);
    t3 = csub(e1, o1);
}

// 8-point DFT in registers: v[p] <- sum_a W8^(a p) v[a], natural order in and out
template <bool INV>
__device__ __forceinline__ void dft8(double2 (&v)[8])
{
    constexpr double h = 0.70710678118654752440;
    double2 s0 = cadd(v[0], v[4]), d0 = csub(v[0], v[4]);
    double2 s1 = cadd(v[1], v[5]), d1 = csub(v[1], v[5]);
    double2 s2 = cadd(v[2], v[6]), d2 = csub(v[2], v[6]);
    double2 s3 = cadd(v[3], v[7]), d3 = csub(v[3], v[7]);
    if (INV) {
        d1 = make_double2(h * (d1.x - d1.y), h * (d1.x + d1.y));
        d2 = make_double2(-d2.y, d2.x);
        d3 = make_double2(-h * (d3.x + d3.y), h * (d3.x - d3.y));
    } else {
        d1 = make_double2(h * (d1.x + d1.y), h * (d1.y - d1.x));
        d2 = make_double2(d2.y, -d2.x);
        d3 = make_double2(h * (d3.y - d3.x), -h * (d3.x + d3.y));
    }
    dft4<INV>(s0, s1, s2, s3);
    dft4<INV>(d0, d1, d2, d3);
    v[0] = s0; v[2] = s1; v[4] = s2; v[6] = s3;
    v[1] = d0; v[3] = d1; v[5] = d2; v[7] = d3;
}

constexpr int kLdsPerWave = 72 * 8;   // complex elements

// 512-point complex FFT of one wave.  In: v[j] = z[lane + 64 j].  Out: v[r] = Z[lane + 64 r].
// n = 64a + 8b + c, k = p + 8q + 64r:
//   pass 1 over a (registers), twiddle W512^((8b+c) p); exchange -> lane (p, c), registers over b
//   pass 2 over b, twiddle W64^(c q);                   exchange -> lane (p + 8q), registers over c
//   pass 3 over c.
// LDS traffic between the lanes of ONE wave: the hardware keeps a wave's DS operations in order; this keeps the compiler
// from moving them across each other
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// v[p] *= w^p (INV: conj(w)^p), p = 1..7, the powers formed by multiplication (depth <= 3 products: a few ulp) rather than
// read from a table: the transforms below are bound by LDS bandwidth (a frame of the P = 4096 kernels moved 106 16-byte
// LDS accesses per lane, 42 of them twiddle-table reads), not by the VALU
template <bool INV>
__device__ __forceinline__ void mul_powers(double2 (&v)[8], double2 w1)
{
    const double2 w2 = cmul(w1, w1), w3 = cmul(w2, w1), w4 = cmul(w2, w2);
    v[1] = cmulw<INV>(v[1], w1);
    v[2] = cmulw<INV>(v[2], w2);
    v[3] = cmulw<INV>(v[3], w3);
    v[4] = cmulw<INV>(v[4], w4);
    v[5] = cmulw<INV>(v[5], cmul(w4, w1));
    v[6] = cmulw<INV>(v[6], cmul(w3, w3));
    v[7] = cmulw<INV>(v[7], cmul(w4, w3));
}

// 512-point complex FFT of one wave: v[p] = z[lane + 64 p] in, Z[lane + 64 c] out (three radix-8 passes, two exchanges
// through `lds`).  wa = W512^lane, wb = W64^(lane & 7) (W_N = exp(-2 pi i / N)).
// WAVE: the wave is part of a larger workgroup and `lds` is its private slice: wave-level synchronisation instead of
// workgroup barriers
template <bool INV, bool WAVE = false>
__device__ __forceinline__ void wave_cfft512(double2 (&v)[8], double2* lds, int lane, double2 wa, double2 wb)
{
    auto sync = [] { if (WAVE) wave_sync(); else __syncthreads(); };
    dft8<INV>(v);
    mul_powers<INV>(v, wa);
#pragma unroll
    for (int p = 0; p < 8; ++p) lds[72 * p + lane] = v[p];
    sync();
    const int pp = lane >> 3, cc = lane & 7;
#pragma unroll
    for (int b = 0; b < 8; ++b) v[b] = lds[72 * pp + 8 * b + cc];
    sync();
    dft8<INV>(v);
    mul_powers<INV>(v, wb);
#pragma unroll
    for (int q = 0; q < 8; ++q) lds[66 * cc + pp + 8 * q] = v[q];
    sync();
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = lds[66 * c + lane];
    dft8<INV>(v);
}

// real-FFT split: Z (512-pt FFT of even+i*odd samples) -> packed spectrum of the 1024-pt real frame
__device__ __forceinline__ void wave_split_store(double2 (&v)[8], double2* lds, double2 wl,
                                                 int lane, double2* __restrict__ spec, double2* __restrict__ dcnyq)
{
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) lds[lane + 64 * r] = v[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k = lane + 64 * r;
        const double2 zk = v[r];
        const double2 zm = lds[(512 - k) & 511];
        const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));   // (Z[k] + conj Z[512-k]) / 2
        const double2 d = make_double2(0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y));   // (Z[k] - conj Z[512-k]) / 2
        const double2 o = make_double2(d.y, -d.x);                                  // -i d
        const double2 w = cmul(wl, w16(r));                                         // exp(-2 pi i (lane + 64 r) / 1024)
        double2 xk = make_double2(e.x + fma(o.x, w.x, -(o.y * w.y)), e.y + fma(o.x, w.y, o.y * w.x));
        if (k == 0) {
            xk = make_double2(zk.x + zk.y, zk.x - zk.y);   // (DC, Nyquist)
            *dcnyq = xk;
        }
        // streaming store: an FDL row is next read by the MAC of this or a later call, never from cache (the MAC of the
        // same call runs 1 % faster with its L2 left alone)
        __builtin_nontemporal_store(v2d{ xk.x, xk.y }, reinterpret_cast<v2d*>(spec + k));      // one 16-byte store per lane
    }
}

// side: a plan group's call on its first launch -- the block also goes into the input accumulators of the group's other layers
// (Add()'s per-layer accumulation, src/MKLNonUniformConvolver.cpp:1431-1446: what k_rows_gather_multi would do in a launch of
// its own) and the call's chunk tables, riding along as kernel arguments, are stored for the kernels behind this one.
// tail*: the samples behind the last whole block of the source rows (a layer's input accumulator) go to the front of the other
// accumulator buffer -- the remainder move that would be a k_rows_copy launch of its own.
struct FwdSide {
    double* dst[2]; long long stride[2], off[2]; int n;
    long long* tabDst; int nTab; long long tab[kGatherTabMax];
    double* tailDst; long long tailStride; int tailLen;
};
template <bool SIDE>
__global__ __launch_bounds__(64) void k_rfft_fwd_ols(const double* __restrict__ in, int64_t chStride,
                                                     const double* __restrict__ histOld,
                                                     double* __restrict__ histNew, double2* __restrict__ X,
                                                     double2* __restrict__ XDN, FftTables tw, int T, int head,
                                                     int ringMask, FwdSide side)
{
    __shared__ double2 lds[kLdsPerWave];
    const int lane = threadIdx.x;
    const int c = blockIdx.x / T;
    const int t = blockIdx.x - c * T;
    const double* cur = in + (int64_t)c * chStride + (int64_t)t * kP;
    const double* prev = (t > 0) ? (cur - kP) : (histOld + (int64_t)c * kP);
    const double2 wl = tw.tw1024[lane], wa = tw.tw512[lane], wb = tw.tw512[8 * (lane & 7)];

    double2 v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const double2*>(prev + 2 * (lane + 64 * j));
#pragma unroll
    for (int j = 0; j < 4; ++j) v[4 + j] = *reinterpret_cast<const double2*>(cur + 2 * (lane + 64 * j));
    if (t == T - 1) {   // overlap history for the next call (prevInputBuf, NUC.cpp:1258)
        double* hn = histNew + (int64_t)c * kP;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<double2*>(hn + 2 * (lane + 64 * j)) = v[4 + j];
    }
    if (SIDE) {
        if (blockIdx.x == 0 && lane < side.nTab) side.tabDst[lane] = side.tab[lane];
        if (t == T - 1 && side.tailLen > 0) {
            const double* ts = in + (int64_t)c * chStride + (int64_t)T * kP;
            double* td = side.tailDst + (int64_t)c * side.tailStride;
            for (int i = lane; i < side.tailLen; i += 64) td[i] = ts[i];
        }
        for (int a = 0; a < side.n; ++a) {          // (offsets even: 16-byte stores)
            double* d = side.dst[a] + (int64_t)c * side.stride[a] + side.off[a] + (int64_t)t * kP;
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<double2*>(d + 2 * (lane + 64 * j)) = v[4 + j];
        }
    }
    wave_cfft512<false>(v, lds, lane, wa, wb);
    const int slot = (head + t) & ringMask;
    const int64_t row = (int64_t)c * (ringMask + 1) + slot;
    wave_split_store(v, lds, wl, lane, X + row * kP, XDN + row);
}

__global__ __launch_bounds__(64) void k_ir_spectra(const double* __restrict__ heff, int heffLen,
                                                   double2* __restrict__ H, double2* __restrict__ HDN, FftTables tw)
{
    __shared__ double2 lds[kLdsPerWave];
    const int lane = threadIdx.x;
    const int k = blockIdx.x;
    const double2 wl = tw.tw1024[lane], wa = tw.tw512[lane], wb = tw.tw512[8 * (lane & 7)];
    double2 v[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = k * kP + 2 * (lane + 64 * j);
        v[j] = make_double2(i < heffLen ? heff[i] : 0.0, (i + 1) < heffLen ? heff[i + 1] : 0.0);
    }
#pragma unroll
    for (int j = 4; j < 8; ++j) v[j] = make_double2(0.0, 0.0);   // zero-padded second half (NUC.cpp:921-928)
    wave_cfft512<false>(v, lds, lane, wa, wb);
    wave_split_store(v, lds, wl, lane, H + (int64_t)k * kP, HDN + k);
}

// Where an inverse transform puts block t of channel c (i = even sample index inside the block):
//   MODE 0: the call's output rows;
//   MODE 1: a per-channel ring at the position the block has there -- pos[t], or pos0 + t P when pos is null; a negative
//           position drops the block (ringWrite / delayLineWrite of the plan groups, src/MKLNonUniformConvolver.cpp:1341-1371,
//           :1639-1648, fused into the transform's stores);
//   MODE 2: the output rows PLUS what the replayed delay-line reader of the tail layers adds at those samples (layered
//           mode, mix_kernels.hip: k_tail_schedule has filled sched for the call; the additions come from this call's
//           natural-time tail outputs where the sample lies inside the call, from the layer's ring where it is older) --
//           the read-modify-write pass over the call's output that k_layer_combine would make is gone.
//   MODE 3: the output rows PLUS the delay-line read-add of a plan group's tail layers (Get(), :1620-1633: layer 1 first, then
//           layer 2, `dst += src` or `dst += src * gain`) where every block is one chunk of the call: what k_ring_add_chunks[2]
//           would do in a pass of its own over the output.
struct OutSpec {
    double* ring; int mask; const long long* pos; long long pos0;                                  // MODE 1; MODE 3: ring A, its mask, its schedule
    const double* ringB; int maskB; const long long* schedB;                                      // MODE 3: ring B (or null)
    const double* layerOut; const double* tailRing; const long long* tailState; const long long* sched;      // MODE 2
    int nCb, B, log2B, tailMask, nTail, nChAll, nSamples; double g1, g2;
};
template <int MODE>
__device__ __forceinline__ void store_block2(double* out, int64_t chStride, const OutSpec& ro, int c, int t, int P, int i, double2 v)
{
    if (MODE == 0) { *reinterpret_cast<double2*>(out + (int64_t)c * chStride + (int64_t)t * P + i) = v; return; }
    if (MODE == 2) {
        const long long g0 = ro.tailState[3];          // TailState::g0: global index of the call's first sample
        auto tail = [&](int n, double y) {
            const int cbk = n >> ro.log2B, j = n & (ro.B - 1);          // (whole-block engines: B is a power of two)
            for (int l = 0; l < ro.nTail; ++l) {
                const long long s = ro.sched[(long long)l * ro.nCb + cbk];
                if (s >= 0) {
                    const double g = l == 0 ? ro.g1 : ro.g2;
                    const long long idx = s + j;
                    const double x = idx >= g0 ? ro.layerOut[((long long)l * ro.nChAll + c) * ro.nSamples + (idx - g0)]
                                               : ro.tailRing[((long long)l * ro.nChAll + c) * (ro.tailMask + 1) + (idx & ro.tailMask)];
                    // delayLineReadAdd: dst += src (gain within 1e-12 of 1) else dst += src * gain (:1673-1676)
                    y = (fabs(g - 1.0) < 1.0e-12) ? (y + x) : (y + x * g);
                }
            }
            return y;
        };
        const int n0 = t * P + i;
        v.x = tail(n0, v.x);
        v.y = tail(n0 + 1, v.y);
        *reinterpret_cast<double2*>(out + (int64_t)c * chStride + (int64_t)t * P + i) = v;
        return;
    }
    if (MODE == 3) {
        const long long sa = ro.pos[t], sb = ro.ringB ? ro.schedB[t] : -1;
        if (sa >= 0) {
            const double* ra = ro.ring + (int64_t)c * (ro.mask + 1);
            const double a0 = ra[(sa + i) & ro.mask], a1 = ra[(sa + i + 1) & ro.mask];
            const bool unity = fabs(ro.g1 - 1.0) < 1.0e-12;
            v.x = unity ? (v.x + a0) : (v.x + a0 * ro.g1);
            v.y = unity ? (v.y + a1) : (v.y + a1 * ro.g1);
        }
        if (sb >= 0) {
            const double* rb = ro.ringB + (int64_t)c * (ro.maskB + 1);
            const double b0 = rb[(sb + i) & ro.maskB], b1 = rb[(sb + i + 1) & ro.maskB];
            const bool unity = fabs(ro.g2 - 1.0) < 1.0e-12;
            v.x = unity ? (v.x + b0) : (v.x + b0 * ro.g2);
            v.y = unity ? (v.y + b1) : (v.y + b1 * ro.g2);
        }
        *reinterpret_cast<double2*>(out + (int64_t)c * chStride + (int64_t)t * P + i) = v;
        return;
    }
    const long long p = ro.pos ? ro.pos[t] : ro.pos0 + (long long)t * P;
    if (p < 0) return;
    double* r = ro.ring + (int64_t)c * (ro.mask + 1);
    const long long a = p + i;
    if ((p & 1) == 0) *reinterpret_cast<double2*>(r + (a & ro.mask)) = v;
    else { r[a & ro.mask] = v.x; r[(a + 1) & ro.mask] = v.y; }
}

template <int MODE>
__global__ __launch_bounds__(64) void k_rfft_inv_ols(const double2* __restrict__ Y, double* __restrict__ out,
                                                     int64_t chStride, FftTables tw, int T, OutSpec ro)
{
    __shared__ double2 lds[kLdsPerWave];
    const int lane = threadIdx.x;
    const int c = blockIdx.x / T;
    const int t = blockIdx.x - c * T;
    const double2* y = Y + (int64_t)blockIdx.x * kP;
    const double2 wl = tw.tw1024[lane], wa = tw.tw512[lane], wb = tw.tw512[8 * (lane & 7)];

    double2 v[8];
    const double2 y0 = y[0];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = lane + 64 * j;
        const double2 yk = y[k];
        const double2 ym = y[(512 - k) & 511];
        const double2 e = make_double2(0.5 * (yk.x + ym.x), 0.5 * (yk.y - ym.y));
        const double2 d = make_double2(0.5 * (yk.x - ym.x), 0.5 * (yk.y + ym.y));
        const double2 w = cmul(wl, w16(j));                                                     // exp(-2 pi i (lane + 64 j) / 1024)
        const double2 o = make_double2(fma(d.x, w.x, d.y * w.y), fma(d.y, w.x, -(d.x * w.y)));   // d * conj(w)
        double2 z = make_double2(e.x - o.y, e.y + o.x);                                          // E + i O
        if (k == 0) z = make_double2(0.5 * (y0.x + y0.y), 0.5 * (y0.x - y0.y));
        v[j] = z;
    }
    wave_cfft512<true>(v, lds, lane, wa, wb);
    // second half of the 1024-sample frame: z[n], n = lane + 64 r, r = 4..7  (NUC.cpp:1332)
    constexpr double s = 1.0 / 512.0;
#pragma unroll
    for (int r = 4; r < 8; ++r)
        store_block2<MODE>(out, chStride, ro, c, t, kP, 2 * (lane + 64 * (r - 4)), make_double2(v[r].x * s, v[r].y * s));
}


// ---------------------------------------------------------------------------------------------------------
// Generic partition sizes (P = 64 ... 2048 except 512): one workgroup per transform, radix-2 Stockham
// autosort between two LDS buffers (2 x P complex), then the same real-FFT split.  Slower per byte than the
// wave-level 512-point kernels above; covers the block-size sweep of BASELINE.json configs[2].
// tw.tw512 holds exp(-2 pi i m / P) (m < P), tw.tw1024 holds exp(-2 pi i k / 2P) (k < P) for the engine's P.

template <bool INV>
__device__ __forceinline__ double2* stockham(double2* a, double2* b, int M, const double2* __restrict__ twM)
{
    const int half = M >> 1;
    for (int ns = 1; ns < M; ns <<= 1) {
        const int tstep = half / ns;                 // twiddle index scale: exp(-2 pi i k / (2 ns)) = twM[k * tstep]
        for (int j = threadIdx.x; j < half; j += blockDim.x) {
            const int k = j & (ns - 1);
            const double2 w = twM[k * tstep];
            const double2 u = a[j];
            const double2 v = cmulw<INV>(a[j + half], w);
            const int o = ((j - k) << 1) + k;
            b[o] = cadd(u, v);
            b[o + ns] = csub(u, v);
        }
        __syncthreads();
        double2* t = a; a = b; b = t;
    }
    return a;      // buffer holding the result, natural order
}

__device__ __forceinline__ void split_store_generic(const double2* Z, int M, const double2* __restrict__ tw2M,
                                                    double2* __restrict__ spec, double2* __restrict__ dcnyq)
{
    for (int k = threadIdx.x; k < M; k += blockDim.x) {
        const double2 zk = Z[k];
        const double2 zm = Z[(M - k) & (M - 1)];
        const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
        const double2 d = make_double2(0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y));
        const double2 o = make_double2(d.y, -d.x);
        const double2 w = tw2M[k];
        double2 xk = make_double2(e.x + fma(o.x, w.x, -(o.y * w.y)), e.y + fma(o.x, w.y, o.y * w.x));
        if (k == 0) {
            xk = make_double2(zk.x + zk.y, zk.x - zk.y);
            *dcnyq = xk;
        }
        spec[k] = xk;
    }
}

__global__ __launch_bounds__(256) void k_rfft_fwd_ols_generic(const double* __restrict__ in, int64_t chStride,
                                                              const double* __restrict__ histOld,
                                                              double* __restrict__ histNew, double2* __restrict__ X,
                                                              double2* __restrict__ XDN, FftTables tw, int P, int T,
                                                              int head, int ringMask)
{
    extern __shared__ double2 dyn[];
    double2* a = dyn;
    double2* b = dyn + P;
    const int c = blockIdx.x / T;
    const int t = blockIdx.x - c * T;
    const double* cur = in + (int64_t)c * chStride + (int64_t)t * P;
    const double* prev = (t > 0) ? (cur - P) : (histOld + (int64_t)c * P);
    const int halfP = P >> 1;
    for (int n = threadIdx.x; n < P; n += blockDim.x) {
        const double* src = (n < halfP) ? (prev + 2 * n) : (cur + 2 * (n - halfP));
        const double2 v = *reinterpret_cast<const double2*>(src);
        a[n] = v;
        if (t == T - 1 && n >= halfP) *reinterpret_cast<double2*>(histNew + (int64_t)c * P + 2 * (n - halfP)) = v;
    }
    __syncthreads();
    const double2* Z = stockham<false>(a, b, P, tw.tw512);
    const int slot = (head + t) & ringMask;
    const int64_t row = (int64_t)c * (ringMask + 1) + slot;
    split_store_generic(Z, P, tw.tw1024, X + row * P, XDN + row);
}

__global__ __launch_bounds__(256) void k_ir_spectra_generic(const double* __restrict__ heff, int heffLen,
                                                            double2* __restrict__ H, double2* __restrict__ HDN,
                                                            FftTables tw, int P)
{
    extern __shared__ double2 dyn[];
    double2* a = dyn;
    double2* b = dyn + P;
    const int k = blockIdx.x;
    const int halfP = P >> 1;
    for (int n = threadIdx.x; n < P; n += blockDim.x) {
        double2 v = make_double2(0.0, 0.0);
        if (n < halfP) {
            const int i = k * P + 2 * n;
            v = make_double2(i < heffLen ? heff[i] : 0.0, (i + 1) < heffLen ? heff[i + 1] : 0.0);
        }
        a[n] = v;
    }
    __syncthreads();
    const double2* Z = stockham<false>(a, b, P, tw.tw512);
    split_store_generic(Z, P, tw.tw1024, H + (int64_t)k * P, HDN + k);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_rfft_inv_ols_generic(const double2* __restrict__ Y, double* __restrict__ out,
                                                              int64_t chStride, FftTables tw, int P, int T, OutSpec ro)
{
    extern __shared__ double2 dyn[];
    double2* a = dyn;
    double2* b = dyn + P;
    const int c = blockIdx.x / T;
    const int t = blockIdx.x - c * T;
    const double2* y = Y + (int64_t)blockIdx.x * P;
    const double2 y0 = y[0];
    for (int k = threadIdx.x; k < P; k += blockDim.x) {
        const double2 yk = y[k];
        const double2 ym = y[(P - k) & (P - 1)];
        const double2 e = make_double2(0.5 * (yk.x + ym.x), 0.5 * (yk.y - ym.y));
        const double2 d = make_double2(0.5 * (yk.x - ym.x), 0.5 * (yk.y + ym.y));
        const double2 w = tw.tw1024[k];
        const double2 o = make_double2(fma(d.x, w.x, d.y * w.y), fma(d.y, w.x, -(d.x * w.y)));
        double2 z = make_double2(e.x - o.y, e.y + o.x);
        if (k == 0) z = make_double2(0.5 * (y0.x + y0.y), 0.5 * (y0.x - y0.y));
        a[k] = z;
    }
    __syncthreads();
    const double2* z = stockham<true>(a, b, P, tw.tw512);
    const double s = 1.0 / (double)P;
    const int halfP = P >> 1;
    for (int n = halfP + threadIdx.x; n < P; n += blockDim.x)      // second half of the 2P-sample frame
        store_block2<MODE>(out, chStride, ro, c, t, P, 2 * (n - halfP), make_double2(z[n].x * s, z[n].y * s));
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------------------------
// Large partitions (P = 1024, 2048, 4096): one workgroup of P/8 threads per transform, 8 points per thread,
// mixed-radix Stockham (radix-8 stages, then one radix-4 or radix-2 stage when P is not a power of 8) through
// ONE P-complex LDS buffer: every thread reads its 8 inputs, barrier, butterflies in registers, writes its 8
// outputs, barrier.  Stage with radix R and Ns = product of earlier radices, butterfly j in [0, P/R):
//   k = j mod Ns,  in[j + q P/R] * exp(-2 pi i q k / (R Ns))  ->  out[(j - k) R + k + q Ns].
// LDS element index with one pad element per 8: the first radix-8 stage writes elements 8 j + q from consecutive lanes j
// (stride 128 B: an 8-way bank conflict unpadded, conflict-free at stride 9), the second one 64 a + b
__device__ __forceinline__ int wgp(int i) { return i + (i >> 3); }

template <bool INV>
__device__ __forceinline__ void wg_stage8(double2* lds, int M, int ns, const double2* __restrict__ twM)
{
    const int j = threadIdx.x;
    const int stride = M >> 3;
    const int k = j & (ns - 1);
    double2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = lds[wgp(j + q * stride)];
    if (ns > 1) mul_powers<INV>(v, twM[k * (M / (8 * ns))]);      // one root per lane and stage, its powers by multiplication
    dft8<INV>(v);
    __syncthreads();
    const int o = ((j - k) << 3) + k;
#pragma unroll
    for (int q = 0; q < 8; ++q) lds[wgp(o + q * ns)] = v[q];
    __syncthreads();
}

// final radix-R stage (R = 2 or 4) over all P points with P/8 threads: each thread does 8/R butterflies
template <bool INV, int R>
__device__ __forceinline__ void wg_stage_small(double2* lds, int M, int ns, const double2* __restrict__ twM)
{
    constexpr int NB = 8 / R;
    const int stride = M / R;
    const int tstep = M / (R * ns);
    double2 v[NB][R];
    int outBase[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = threadIdx.x + b * blockDim.x;
        const int k = j & (ns - 1);
        outBase[b] = (j - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            v[b][q] = lds[wgp(j + q * stride)];
            if (q > 0) v[b][q] = cmulw<INV>(v[b][q], twM[q * k * tstep]);
        }
        if (R == 4) dft4<INV>(v[b][0], v[b][1], v[b][2], v[b][3]);
        else { const double2 a = v[b][0], c = v[b][1]; v[b][0] = cadd(a, c); v[b][1] = csub(a, c); }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < R; ++q) lds[wgp(outBase[b] + q * ns)] = v[b][q];
    __syncthreads();
}

template <bool INV>
__device__ __forceinline__ void wg_cfft(double2* lds, int M, const double2* __restrict__ twM)
{
    int ns = 1;
    while (ns * 8 <= M) { wg_stage8<INV>(lds, M, ns, twM); ns *= 8; }
    if (ns * 4 == M) wg_stage_small<INV, 4>(lds, M, ns, twM);
    else if (ns * 2 == M) wg_stage_small<INV, 2>(lds, M, ns, twM);
}

// real-FFT split of the padded LDS buffer (same arithmetic as split_store_generic)
__device__ __forceinline__ void split_store_wg(const double2* Z, int M, const double2* __restrict__ tw2M,
                                               double2* __restrict__ spec, double2* __restrict__ dcnyq)
{
    for (int k = threadIdx.x; k < M; k += blockDim.x) {
        const double2 zk = Z[wgp(k)];
        const double2 zm = Z[wgp((M - k) & (M - 1))];
        const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
        const double2 d = make_double2(0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y));
        const double2 o = make_double2(d.y, -d.x);
        const double2 w = tw2M[k];
        double2 xk = make_double2(e.x + fma(o.x, w.x, -(o.y * w.y)), e.y + fma(o.x, w.y, o.y * w.x));
        if (k == 0) {
            xk = make_double2(zk.x + zk.y, zk.x - zk.y);
            *dcnyq = xk;
        }
        spec[k] = xk;
    }
}

__global__ __launch_bounds__(512) void k_rfft_fwd_ols_wg(const double* __restrict__ in, int64_t chStride,
                                                         const double* __restrict__ histOld,
                                                         double* __restrict__ histNew, double2* __restrict__ X,
                                                         double2* __restrict__ XDN, FftTables tw, int P, int T,
                                                         int head, int ringMask)
{
    extern __shared__ double2 dyn[];
    const int c = blockIdx.x / T;
    const int t = blockIdx.x - c * T;
    const double* cur = in + (int64_t)c * chStride + (int64_t)t * P;
    const double* prev = (t > 0) ? (cur - P) : (histOld + (int64_t)c * P);
    const int halfP = P >> 1;
    for (int n = threadIdx.x; n < P; n += blockDim.x) {
        const double* src = (n < halfP) ? (prev + 2 * n) : (cur + 2 * (n - halfP));
        const double2 v = *reinterpret_cast<const double2*>(src);
        dyn[wgp(n)] = v;
        if (t == T - 1 && n >= halfP) *reinterpret_cast<double2*>(histNew + (int64_t)c * P + 2 * (n - halfP)) = v;
    }
    __syncthreads();
    wg_cfft<false>(dyn, P, tw.tw512);
    const int slot = (head + t) & ringMask;
    const int64_t row = (int64_t)c * (ringMask + 1) + slot;
    split_store_wg(dyn, P, tw.tw1024, X + row * P, XDN + row);
}

__global__ __launch_bounds__(512) void k_ir_spectra_wg(const double* __restrict__ heff, int heffLen,
                                                       double2* __restrict__ H, double2* __restrict__ HDN,
                                                       FftTables tw, int P)
{
    extern __shared__ double2 dyn[];
    const int k = blockIdx.x;
    const int halfP = P >> 1;
    for (int n = threadIdx.x; n < P; n += blockDim.x) {
        double2 v = make_double2(0.0, 0.0);
        if (n < halfP) {
            const int i = k * P + 2 * n;
            v = make_double2(i < heffLen ? heff[i] : 0.0, (i + 1) < heffLen ? heff[i + 1] : 0.0);
        }
        dyn[wgp(n)] = v;
    }
    __syncthreads();
    wg_cfft<false>(dyn, P, tw.tw512);
    split_store_wg(dyn, P, tw.tw1024, H + (int64_t)k * P, HDN + k);
}

template <int MODE>
__global__ __launch_bounds__(512) void k_rfft_inv_ols_wg(const double2* __restrict__ Y, double* __restrict__ out,
                                                         int64_t chStride, FftTables tw, int P, int T, OutSpec ro)
{
    extern __shared__ double2 dyn[];
    const int c = blockIdx.x / T;
    const int t = blockIdx.x - c * T;
    const double2* y = Y + (int64_t)blockIdx.x * P;
    const double2 y0 = y[0];
    for (int k = threadIdx.x; k < P; k += blockDim.x) {
        const double2 yk = y[k];
        const double2 ym = y[(P - k) & (P - 1)];
        const double2 e = make_double2(0.5 * (yk.x + ym.x), 0.5 * (yk.y - ym.y));
        const double2 d = make_double2(0.5 * (yk.x - ym.x), 0.5 * (yk.y + ym.y));
        const double2 w = tw.tw1024[k];
        const double2 o = make_double2(fma(d.x, w.x, d.y * w.y), fma(d.y, w.x, -(d.x * w.y)));
        double2 z = make_double2(e.x - o.y, e.y + o.x);
        if (k == 0) z = make_double2(0.5 * (y0.x + y0.y), 0.5 * (y0.x - y0.y));
        dyn[wgp(k)] = z;
    }
    __syncthreads();
    wg_cfft<true>(dyn, P, tw.tw512);
    const double s = 1.0 / (double)P;
    const int halfP = P >> 1;
    for (int n = halfP + threadIdx.x; n < P; n += blockDim.x)
        store_block2<MODE>(out, chStride, ro, c, t, P, 2 * (n - halfP), make_double2(dyn[wgp(n)].x * s, dyn[wgp(n)].y * s));
}

// ---------------------------------------------------------------------------------------------------------
// P = 4096 (8192-point real frames), the partition of the time-batched throughput path.  Four-step transform of the
// 4096 complex points z[n], n = n2 + 512 n1, inside ONE workgroup of eight waves, built on the wave-level 512-point
// transform above (which streams at 5-6 TB/s, twice what a Stockham pipeline with a workgroup barrier behind every stage
// reached at this size):
//   1. thread j = n2 holds z[j + 512 q], q < 8 (coalesced loads): 8-point DFT over n1 in registers, times W4096^(j k1)
//   2. ONE workgroup-level exchange hands row k1 (512 values) to wave k1
//   3. wave k1: 512-point transform over n2 in its own 9 KB slice of the exchange buffer (wave-local exchanges only)
//      -> Z[k1 + 8 k2]
// The spectrum is stored PERMUTED, element k1 * 512 + k2 = bin k1 + 8 k2 (contiguous per wave: coalesced stores), like
// the larger partitions below: the MAC is element-wise and the forward, IR and inverse transforms agree on it; element 0
// is still the packed (DC, Nyquist).  The real-FFT split pairs (k1, k2) with (8 - k1, 511 - k2) for k1 > 0 and with
// (0, (512 - k2) mod 512) for k1 = 0: a second exchange between partner waves.  Three workgroup barriers per frame
// (four in the inverse) instead of eight: forward 1.38 -> 1.33 ms, inverse 1.88 -> 1.70 ms per 1024-block call of 256 streams
// (starting the CU's second workgroup half a frame late changed nothing).  A workgroup walks consecutive frames of one channel; the forward transform
// keeps the half frame that frames t and t + 1 share in registers (every input block is read once).
// Twiddles never come from global memory inside the loop: W4096^(j k1) factors over the octal digits of j into entries of
// three 8 x 8 tables in LDS (3 KB, filled once per workgroup from the engine's extended-precision table; the first two
// are also the tables of the wave-level transform), the split twiddle is one per-thread constant times W16^r.
constexpr int kP4 = 4096;
constexpr int kP4Row = kLdsPerWave;     // 576: row stride of the exchange buffer = one wave's scratch slice

// per-lane first powers; the rest of every twiddle set is formed by multiplication (mul_powers)
struct P4Tables { double2 w512[64], w4096[64], w64[8]; };      // W512^lane, W4096^lane, W64^k

__device__ __forceinline__ void p4_load_tables(P4Tables* T, const double2* __restrict__ twM)     // twM[m] = W4096^m
{
    const int i = threadIdx.x;
    if (i < 64) { T->w512[i] = twM[8 * i]; T->w4096[i] = twM[i]; }
    if (i < 8) T->w64[i] = twM[64 * i];
    __syncthreads();
}

// v[k1] *= W4096^(+-j k1), j = lane + 64 w
template <bool INV>
__device__ __forceinline__ void p4_twiddle(double2 (&v)[8], const P4Tables* T, int j)
{
    mul_powers<INV>(v, cmul(T->w4096[j & 63], T->w64[j >> 6]));
}

// partner of element (w, k2) in the real-FFT split, as an index into the exchange buffer
__device__ __forceinline__ int p4_partner(int w, int k2)
{
    return w == 0 ? ((512 - k2) & 511) : (8 - w) * kP4Row + (511 - k2);
}

// one forward frame: in v = z[j + 512 q]; out: the packed spectrum row (permuted) and its (DC, Nyquist) element
__device__ __forceinline__ void p4_forward_frame(double2 (&v)[8], double2* dyn, const P4Tables* tabs, double2 wk,
                                                 double2* __restrict__ spec, double2* __restrict__ dcnyq)
{
    const int j = threadIdx.x, w = j >> 6, lane = j & 63;
    dft8<false>(v);
    p4_twiddle<false>(v, tabs, j);
#pragma unroll
    for (int k1 = 0; k1 < 8; ++k1) dyn[k1 * kP4Row + j] = v[k1];
    __syncthreads();
    double2* mine = dyn + w * kP4Row;
#pragma unroll
    for (int m = 0; m < 8; ++m) v[m] = mine[lane + 64 * m];
    wave_sync();                            // this wave's reads before it reuses its slice as scratch
    wave_cfft512<false, true>(v, mine, lane, tabs->w512[lane], tabs->w64[lane & 7]);
    // v[r] = Z[w + 8 (lane + 64 r)]; the split needs the partner wave's row
    wave_sync();
#pragma unroll
    for (int r = 0; r < 8; ++r) mine[lane + 64 * r] = v[r];
    __syncthreads();
    // (five of the eight partner addresses are spilled and reloaded per frame -- the kernel sits at its 128 registers; rebuilding
    // them per frame instead removes the scratch traffic and measured SLOWER, 1.15 -> 1.20-1.26 ms: profiles/r03q_ab_fft_partner.txt)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k2 = lane + 64 * r;
        const double2 zk = v[r];
        const double2 zm = dyn[p4_partner(w, k2)];
        const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));   // (Z[k] + conj Z[4096-k]) / 2
        const double2 d = make_double2(0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y));   // (Z[k] - conj Z[4096-k]) / 2
        const double2 o = make_double2(d.y, -d.x);                                  // -i d
        const double2 tw = cmul(wk, w16(r));                                        // exp(-2 pi i (w + 8 lane + 512 r) / 8192)
        double2 xk = make_double2(e.x + fma(o.x, tw.x, -(o.y * tw.y)), e.y + fma(o.x, tw.y, o.y * tw.x));
        if (w == 0 && k2 == 0) {
            xk = make_double2(zk.x + zk.y, zk.x - zk.y);   // (DC, Nyquist)
            *dcnyq = xk;
        }
        // one 16-byte streaming store per lane: an FDL row is next read by the MAC, never from cache
        __builtin_nontemporal_store(v2d{ xk.x, xk.y }, reinterpret_cast<v2d*>(spec + w * 512 + k2));
    }
    __syncthreads();                        // the partners are done with this wave's slice before the next frame writes
}

__global__ __launch_bounds__(512, 4) void k_rfft_fwd_ols_p4(const double* __restrict__ in, int64_t chStride,
                                                          const double* __restrict__ histOld,
                                                          double* __restrict__ histNew, double2* __restrict__ X,
                                                          double2* __restrict__ XDN, FftTables tw, int T, int split,
                                                          int head, int ringMask)
{
    extern __shared__ double2 dyn[];
    __shared__ P4Tables tabs;
    const int j = threadIdx.x;
    const int c = blockIdx.x / split;
    const int part = blockIdx.x - c * split;
    const int per = (T + split - 1) / split;
    const int t0 = part * per, t1 = min(T, t0 + per);
    if (t0 >= t1) return;
    p4_load_tables(&tabs, tw.tw512);
    const double2 wk = tw.tw1024[(j >> 6) + 8 * (j & 63)];
    const double* base = in + (int64_t)c * chStride;
    // frame t = [block t-1 | block t]; thread j holds complex points n = j + 512 q: n < 2048 from the previous block
    double2 keep[4];
    {
        const double* prev = (t0 > 0) ? (base + (int64_t)(t0 - 1) * kP4) : (histOld + (int64_t)c * kP4);
#pragma unroll
        for (int q = 0; q < 4; ++q) keep[q] = *reinterpret_cast<const double2*>(prev + 2 * (j + 512 * q));
    }
    for (int t = t0; t < t1; ++t) {
        double2 v[8];
        const double* cur = base + (int64_t)t * kP4;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = keep[q];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[4 + q] = *reinterpret_cast<const double2*>(cur + 2 * (j + 512 * q));
#pragma unroll
        for (int q = 0; q < 4; ++q) keep[q] = v[4 + q];
        if (t == T - 1) {       // overlap history for the next call (prevInputBuf, NUC.cpp:1258)
            double* hn = histNew + (int64_t)c * kP4;
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<double2*>(hn + 2 * (j + 512 * q)) = v[4 + q];
        }
        const int slot = (head + t) & ringMask;
        const int64_t row = (int64_t)c * (ringMask + 1) + slot;
        p4_forward_frame(v, dyn, &tabs, wk, X + row * kP4, XDN + row);
    }
}

// IR partition spectra at P = 4096: frames [h[k P .. (k+1) P) | 0], same permuted layout
__global__ __launch_bounds__(512, 4) void k_ir_spectra_p4(const double* __restrict__ heff, int heffLen,
                                                        double2* __restrict__ H, double2* __restrict__ HDN, FftTables tw)
{
    extern __shared__ double2 dyn[];
    __shared__ P4Tables tabs;
    const int j = threadIdx.x;
    const int k = blockIdx.x;
    p4_load_tables(&tabs, tw.tw512);
    const double2 wk = tw.tw1024[(j >> 6) + 8 * (j & 63)];
    double2 v[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = k * kP4 + 2 * (j + 512 * q);
        v[q] = make_double2(i < heffLen ? heff[i] : 0.0, (i + 1) < heffLen ? heff[i + 1] : 0.0);
        v[4 + q] = make_double2(0.0, 0.0);          // zero-padded second half (NUC.cpp:921-928)
    }
    p4_forward_frame(v, dyn, &tabs, wk, H + (int64_t)k * kP4, HDN + k);
}

template <int MODE>
__global__ __launch_bounds__(512, 4) void k_rfft_inv_ols_p4(const double2* __restrict__ Y, double* __restrict__ out,
                                                          int64_t chStride, FftTables tw, int T, int split, OutSpec ro)
{
    extern __shared__ double2 dyn[];
    __shared__ P4Tables tabs;
    const int j = threadIdx.x, w = j >> 6, lane = j & 63;
    const int c = blockIdx.x / split;
    const int part = blockIdx.x - c * split;
    const int per = (T + split - 1) / split;
    const int t0 = part * per, t1 = min(T, t0 + per);
    if (t0 >= t1) return;
    p4_load_tables(&tabs, tw.tw512);
    const double2 wk = tw.tw1024[w + 8 * lane];
    const double2* ybase = Y + (int64_t)c * T * kP4;
    double2* mine = dyn + w * kP4Row;
    for (int t = t0; t < t1; ++t) {
        double2 v[8];
        const double2* yrow = ybase + (int64_t)t * kP4;
        const double2* y = yrow + w * 512;
        // wave w reads its row of the permuted spectrum and, a second time, the row of its partner wave for the element
        // Y[4096 - k] of the real-FFT split: the second read is served by the L2 (the workgroup reads each row twice within
        // microseconds) and spares the LDS -- the bound of this kernel -- 16 accesses per lane and two workgroup barriers
        double2 pa[8], pb[8];
        // (the partner offsets are rebuilt per frame from a value the compiler cannot see through: kept across the frame loop
        // they cost 16 registers, which were spilled and reloaded per frame)
        int laneP = lane;
        asm volatile("" : "+v"(laneP));
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k2 = lane + 64 * r;
            const int k2p = laneP + 64 * r;
            pa[r] = y[k2];
            pb[r] = yrow[w == 0 ? ((512 - k2p) & 511) : (8 - w) * 512 + (511 - k2p)];
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int k2 = lane + 64 * r;
            const double2 a = pa[r], b = pb[r];
            const double2 e = make_double2(0.5 * (a.x + b.x), 0.5 * (a.y - b.y));
            const double2 d = make_double2(0.5 * (a.x - b.x), 0.5 * (a.y + b.y));
            const double2 tk = cmul(wk, w16(r));
            const double2 o = make_double2(fma(d.x, tk.x, d.y * tk.y), fma(d.y, tk.x, -(d.x * tk.y)));   // d * conj(w)
            double2 z = make_double2(e.x - o.y, e.y + o.x);                                              // E + i O
            if (w == 0 && k2 == 0) z = make_double2(0.5 * (a.x + a.y), 0.5 * (a.x - a.y));
            v[r] = z;
        }
        wave_cfft512<true, true>(v, mine, lane, tabs.w512[lane], tabs.w64[lane & 7]);
        // v[m] = C[w][lane + 64 m]: hand column n2 = j to thread j
        wave_sync();
#pragma unroll
        for (int m = 0; m < 8; ++m) mine[lane + 64 * m] = v[m];
        __syncthreads();
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) v[k1] = dyn[k1 * kP4Row + j];
        __syncthreads();                    // the next frame's row writes reuse the buffer
        p4_twiddle<true>(v, &tabs, j);
        dft8<true>(v);
        // second half of the 8192-sample frame: x[n], n = j + 512 n1, n1 = 4..7 (NUC.cpp:1332)
        constexpr double s = 1.0 / (double)kP4;
        if (MODE != 0) {
#pragma unroll
            for (int q = 4; q < 8; ++q)
                store_block2<MODE>(out, chStride, ro, c, t, kP4, 2 * (j + 512 * (q - 4)), make_double2(v[q].x * s, v[q].y * s));
        } else {
            double* o = out + (int64_t)c * chStride + (int64_t)t * kP4;
#pragma unroll
            for (int q = 4; q < 8; ++q)
                *reinterpret_cast<double2*>(o + 2 * (j + 512 * (q - 4))) = make_double2(v[q].x * s, v[q].y * s);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Partitions above 4096 samples (P = 8192, 16384, 32768: tail layers of a FilterSpec plan run at the reference's own
// partition size): four-step FFT of the M = P complex points z[n] = x[2n] + i x[2n+1], M = M1 * 512, n = n1 * 512 + n2:
//   column pass: for every column n2 the M1-point FFT over n1, times W_M^(n2 k1)          -> scratch A[k1][n2]
//   row pass   : for every row k1 the 512-point FFT over n2 (the wave-level kernel above) -> Z[k1 + M1 k2]
// The spectrum stays in that PERMUTED order, element k1 * 512 + k2 <-> bin k1 + M1 k2: the MAC is element-wise and
// the forward, IR and inverse transforms agree on it; element 0 is still the packed (DC, Nyquist).  The real-FFT split
// pairs (k1, k2) with (M1 - k1, 511 - k2) for k1 > 0 and with (0, (512 - k2) mod 512) for k1 = 0, so one workgroup of
// two waves transforms the row pair (k1, M1 - k1).  tools/fft_fourstep_proto.py is the numpy model of this scheme.
// tw.tw512 = exp(-2 pi i m / P) and tw.tw1024 = exp(-2 pi i k / 2P), P entries each.

// columns per workgroup in the column pass: a wave-width of them (coalesced 1 KB rows) up to 64-point columns, fewer for the
// 128- / 256-point columns of 65536- / 131072-sample partitions so that the tile stays at 64 KB and the workgroup at 512 threads
__host__ __device__ constexpr int bigCols(int M1) { return M1 <= 64 ? 64 : (M1 == 128 ? 32 : 16); }

// Stockham stages over the element axis of a [element][column] LDS tile; thread = (column, j), j < M1 / 8
template <bool INV>
__device__ __forceinline__ void col_stage8(double2* lds, int M1, int ns, const double2* __restrict__ twP, int j, int col, int kBigCols)
{
    const int stride = M1 >> 3;
    const int k = j & (ns - 1);
    double2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = lds[(j + q * stride) * kBigCols + col];
    const int tstep = (M1 / (8 * ns)) * 512;             // exp(-2 pi i m / M1) = twP[m * 512]
    mul_powers<INV>(v, twP[k * tstep]);
    dft8<INV>(v);
    __syncthreads();
    const int o = ((j - k) << 3) + k;
#pragma unroll
    for (int q = 0; q < 8; ++q) lds[(o + q * ns) * kBigCols + col] = v[q];
    __syncthreads();
}

template <bool INV, int R>
__device__ __forceinline__ void col_stage_small(double2* lds, int M1, int ns, const double2* __restrict__ twP, int j, int col, int kBigCols)
{
    constexpr int NB = 8 / R;
    const int nthr = M1 >> 3;
    const int stride = M1 / R;
    const int tstep = (M1 / (R * ns)) * 512;
    double2 v[NB][R];
    int outBase[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int jj = j + b * nthr;
        const int k = jj & (ns - 1);
        outBase[b] = (jj - k) * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) {
            v[b][q] = lds[(jj + q * stride) * kBigCols + col];
            if (q > 0) v[b][q] = cmulw<INV>(v[b][q], twP[q * k * tstep]);
        }
        if (R == 4) dft4<INV>(v[b][0], v[b][1], v[b][2], v[b][3]);
        else { const double2 a = v[b][0], c = v[b][1]; v[b][0] = cadd(a, c); v[b][1] = csub(a, c); }
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int q = 0; q < R; ++q) lds[(outBase[b] + q * ns) * kBigCols + col] = v[b][q];
    __syncthreads();
}

// M1-point FFT of the thread's column: the first radix-8 stage takes its inputs from registers (v[q] = element
// j + q M1/8), the result ends in LDS in natural order
template <bool INV>
__device__ __forceinline__ void col_fft(double2 (&v)[8], double2* lds, int M1, const double2* __restrict__ twP, int j, int col, int kBigCols)
{
    dft8<INV>(v);
#pragma unroll
    for (int q = 0; q < 8; ++q) lds[(8 * j + q) * kBigCols + col] = v[q];
    __syncthreads();
    if (M1 >= 64) {                  // 64 = 8 x 8, 128 = 8 x 8 x 2, 256 = 8 x 8 x 4
        col_stage8<INV>(lds, M1, 8, twP, j, col, kBigCols);
        if (M1 == 128) col_stage_small<INV, 2>(lds, M1, 64, twP, j, col, kBigCols);
        else if (M1 == 256) col_stage_small<INV, 4>(lds, M1, 64, twP, j, col, kBigCols);
    } else if (M1 == 32) col_stage_small<INV, 4>(lds, M1, 8, twP, j, col, kBigCols);
    else col_stage_small<INV, 2>(lds, M1, 8, twP, j, col, kBigCols);
}

// column pass, forward.  FRAME = true: overlap-save frame [previous P | current P] of (channel c, block t);
// FRAME = false: zero-padded IR partition blockIdx / 8 of heff.  grid = transforms * 8 column tiles, 8 M1 threads.
template <bool FRAME>
__global__ __launch_bounds__(512) void k_big_cols_fwd(const double* __restrict__ in, int64_t chStride,
                                                      const double* __restrict__ histOld, double* __restrict__ histNew,
                                                      int heffLen, double2* __restrict__ A, FftTables tw, int P, int T)
{
    extern __shared__ double2 dyn[];
    const int M1 = P >> 9;
    const int kBigCols = bigCols(M1), nTiles = 512 / kBigCols;
    const int tr = blockIdx.x / nTiles, tile = blockIdx.x - tr * nTiles;
    const int col = threadIdx.x & (kBigCols - 1), j = threadIdx.x / kBigCols;
    const int n2 = tile * kBigCols + col;
    const int halfM = P >> 1;
    const int stride = M1 >> 3;
    double2 v[8];
    if (FRAME) {
        const int c = tr / T, t = tr - c * T;
        const double* cur = in + (int64_t)c * chStride + (int64_t)t * P;
        const double* prev = (t > 0) ? (cur - P) : (histOld + (int64_t)c * P);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = (j + q * stride) * 512 + n2;
            const double* src = (n < halfM) ? (prev + 2 * n) : (cur + 2 * (n - halfM));
            v[q] = *reinterpret_cast<const double2*>(src);
            if (t == T - 1 && n >= halfM)      // overlap history for the next call (prevInputBuf, NUC.cpp:1258)
                *reinterpret_cast<double2*>(histNew + (int64_t)c * P + 2 * (n - halfM)) = v[q];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int n = (j + q * stride) * 512 + n2;
            double2 x = make_double2(0.0, 0.0);
            if (n < halfM) {
                const int64_t i = (int64_t)tr * P + 2 * n;
                x = make_double2(i < heffLen ? in[i] : 0.0, (i + 1) < heffLen ? in[i + 1] : 0.0);
            }
            v[q] = x;
        }
    }
    col_fft<false>(v, dyn, M1, tw.tw512, j, col, kBigCols);
    double2* a = A + (int64_t)tr * P;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k1 = j + q * stride;
        a[k1 * 512 + n2] = cmulw<false>(dyn[k1 * kBigCols + col], tw.twCol[k1 * 512 + n2]);       // W_M^(n2 k1)
    }
}

// row pass + real-FFT split, forward: workgroup = rows (k1, M1 - k1) of one transform, one wave each
__global__ __launch_bounds__(128) void k_big_rows_fwd(const double2* __restrict__ A, double2* __restrict__ X,
                                                      double2* __restrict__ XDN, FftTables tw, int P, int T, int head,
                                                      int ringMask, int ringRows)
{
    __shared__ double2 lds[2][kLdsPerWave];
    __shared__ double2 rows[2][512];
    const int M1 = P >> 9;
    const int nPairs = (M1 >> 1) + 1;
    const int tr = blockIdx.x / nPairs, k1 = blockIdx.x - tr * nPairs;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int krow = (w == 0) ? k1 : ((M1 - k1) & (M1 - 1));
    const double2* a = A + (int64_t)tr * P + krow * 512;
    double2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = a[lane + 64 * r];
    wave_cfft512<false>(v, lds[w], lane, tw.tw512[lane * M1], tw.tw512[8 * (lane & 7) * M1]);
#pragma unroll
    for (int r = 0; r < 8; ++r) rows[w][lane + 64 * r] = v[r];
    __syncthreads();
    // ringRows > 0: FDL ring of (channel, block); ringRows == 0: IR partition rows (tr = partition)
    int64_t row = tr;
    if (ringRows > 0) {
        const int c = tr / T, t = tr - c * T;
        row = (int64_t)c * ringRows + ((head + t) & ringMask);
    }
    double2* x = X + row * P + krow * 512;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k2 = lane + 64 * r;
        const double2 zk = v[r];
        const double2 zm = rows[w ^ 1][krow == 0 ? ((512 - k2) & 511) : (511 - k2)];
        const double2 e = make_double2(0.5 * (zk.x + zm.x), 0.5 * (zk.y - zm.y));
        const double2 d = make_double2(0.5 * (zk.x - zm.x), 0.5 * (zk.y + zm.y));
        const double2 o = make_double2(d.y, -d.x);
        const double2 wk = tw.twSplit[krow * 512 + k2];
        double2 xk = make_double2(e.x + fma(o.x, wk.x, -(o.y * wk.y)), e.y + fma(o.x, wk.y, o.y * wk.x));
        if (krow == 0 && k2 == 0) {
            xk = make_double2(zk.x + zk.y, zk.x - zk.y);   // (DC, Nyquist)
            XDN[row] = xk;
        }
        x[k2] = xk;
    }
}

// inverse: undo the split, inverse 512-point FFT over k2, conj twiddle -> scratch A'[k1][n2]
__global__ __launch_bounds__(128) void k_big_rows_inv(const double2* __restrict__ Y, double2* __restrict__ A, FftTables tw,
                                                      int P)
{
    __shared__ double2 lds[2][kLdsPerWave];
    __shared__ double2 rows[2][512];
    const int M1 = P >> 9;
    const int nPairs = (M1 >> 1) + 1;
    const int tr = blockIdx.x / nPairs, k1 = blockIdx.x - tr * nPairs;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int krow = (w == 0) ? k1 : ((M1 - k1) & (M1 - 1));
    const double2* y = Y + (int64_t)tr * P + krow * 512;
    const double2 y0 = Y[(int64_t)tr * P];
    // the partner row (M1 - krow) is the other wave's own row (k1 = 0 and M1 / 2: both waves hold the same row): exchanged
    // through LDS like in the forward pass instead of a second read of it from L2
    double2 own[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) { own[r] = y[lane + 64 * r]; rows[w][lane + 64 * r] = own[r]; }
    __syncthreads();
    double2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int k2 = lane + 64 * r;
        const double2 yk = own[r];
        const double2 yp = rows[w ^ 1][krow == 0 ? ((512 - k2) & 511) : (511 - k2)];
        const double2 e = make_double2(0.5 * (yk.x + yp.x), 0.5 * (yk.y - yp.y));
        const double2 d = make_double2(0.5 * (yk.x - yp.x), 0.5 * (yk.y + yp.y));
        const double2 wk = tw.twSplit[krow * 512 + k2];
        const double2 o = make_double2(fma(d.x, wk.x, d.y * wk.y), fma(d.y, wk.x, -(d.x * wk.y)));   // d * conj(w)
        double2 z = make_double2(e.x - o.y, e.y + o.x);
        if (krow == 0 && k2 == 0) z = make_double2(0.5 * (y0.x + y0.y), 0.5 * (y0.x - y0.y));
        v[r] = z;
    }
    wave_cfft512<true>(v, lds[w], lane, tw.tw512[lane * M1], tw.tw512[8 * (lane & 7) * M1]);
    double2* a = A + (int64_t)tr * P + krow * 512;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int n2 = lane + 64 * r;
        a[n2] = cmulw<true>(v[r], tw.twCol[krow * 512 + n2]);
    }
}

// inverse column pass: M1-point inverse FFT over k1 per column, second half of the frame (n1 >= M1 / 2) to out, 1/M
template <int MODE>
__global__ __launch_bounds__(512) void k_big_cols_inv(const double2* __restrict__ A, double* __restrict__ out,
                                                      int64_t chStride, FftTables tw, int P, int T, OutSpec ro)
{
    extern __shared__ double2 dyn[];
    const int M1 = P >> 9;
    const int kBigCols = bigCols(M1), nTiles = 512 / kBigCols;
    const int tr = blockIdx.x / nTiles, tile = blockIdx.x - tr * nTiles;
    const int col = threadIdx.x & (kBigCols - 1), j = threadIdx.x / kBigCols;
    const int n2 = tile * kBigCols + col;
    const int stride = M1 >> 3;
    const double2* a = A + (int64_t)tr * P;
    double2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = a[(j + q * stride) * 512 + n2];
    col_fft<true>(v, dyn, M1, tw.tw512, j, col, kBigCols);
    const int c = tr / T, t = tr - c * T;
    const double s = 1.0 / (double)P;
    const int halfM = P >> 1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int n1 = j + q * stride;
        if (n1 >= (M1 >> 1)) {
            const double2 z = dyn[n1 * kBigCols + col];
            const int n = n1 * 512 + n2;
            store_block2<MODE>(out, chStride, ro, c, t, P, 2 * (n - halfM), make_double2(z.x * s, z.y * s));
        }
    }
}

}  // namespace
void fill_big_twiddles(const double2* tw512, const double2* tw1024, int P, double2* twCol, double2* twSplit)
{
    const int M1 = P >> 9;
    for (int k1 = 0; k1 < M1; ++k1)
        for (int n = 0; n < 512; ++n) {
            twCol[k1 * 512 + n] = tw512[n * k1];
            twSplit[k1 * 512 + n] = tw1024[k1 + M1 * n];
        }
}
namespace {

// H[k][bin] *= gain[bin] for the IR partition spectra of one IR slot (the HC/LC spectral shaping of a non-NULL
// FilterSpec, src/MKLNonUniformConvolver.cpp:433-441); packed bin 0 = (DC * gain[0], Nyquist * gain[P])
// From P = 4096 up the spectra are stored permuted (element k1 * 512 + k2 holds bin k1 + (P / 512) k2, see k_*_p4 / k_big_*).
__global__ __launch_bounds__(256) void k_spectrum_gain(double2* __restrict__ H, double2* __restrict__ HDN,
                                                       const double* __restrict__ gain, int P)
{
    const int k = blockIdx.x;
    double2* row = H + (int64_t)k * P;
    const int M1 = (P >= 4096) ? (P >> 9) : 0;
    for (int e = threadIdx.x; e < P; e += blockDim.x) {
        const int b = M1 ? ((e >> 9) + M1 * (e & 511)) : e;      // bin of element e
        double2 v = row[e];
        if (e == 0) {
            v = make_double2(v.x * gain[0], v.y * gain[P]);
            HDN[k] = v;
        } else {
            v = make_double2(v.x * gain[b], v.y * gain[b]);
        }
        row[e] = v;
    }
}

template <typename K>
void allowLargeLds(K kernel, size_t bytes)
{
    if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace

static size_t wgLdsBytes(int P) { return (size_t)(P + (P >> 3)) * sizeof(double2); }   // padded buffer of the *_wg kernels
// frames of one channel are walked by `split` workgroups (contiguous ranges): one per channel when the channels alone
// fill the chip (2 workgroups per CU), more for few channels
static int p4Split(int nCh, int T)
{
    int split = 1;
    while (nCh * split < 2048 && split * 2 <= T) split *= 2;
    return split;
}
static int genericThreads(int P) { return P / 2 < 64 ? 64 : (P / 2 > 256 ? 256 : P / 2); }

void launch_rfft_fwd_ols(hipStream_t stream, const double* in, int64_t chStride, const double* histOld,
                         double* histNew, double2* X, double2* XDN, FftTables tw, int P, int nCh, int T, int head,
                         int ringSlots, double2* scratch)
{
    if (P > 4096) {         // four-step: column pass into scratch [nCh * T][P], then row pass + split into the FDL ring
        const int M1 = P >> 9;
        hipLaunchKernelGGL(k_big_cols_fwd<true>, dim3(nCh * T * (512 / bigCols(M1))), dim3(bigCols(M1) * M1 / 8), (size_t)M1 * bigCols(M1) * sizeof(double2),
                           stream, in, chStride, histOld, histNew, 0, scratch, tw, P, T);
        hipLaunchKernelGGL(k_big_rows_fwd, dim3(nCh * T * ((M1 >> 1) + 1)), dim3(128), 0, stream, scratch, X, XDN, tw, P,
                           T, head, ringSlots - 1, ringSlots);
        return;
    }
    if (P == kP)
        hipLaunchKernelGGL(k_rfft_fwd_ols<false>, dim3(nCh * T), dim3(64), 0, stream, in, chStride, histOld, histNew, X, XDN,
                           tw, T, head, ringSlots - 1, FwdSide{});
    else if (P == kP4) {
        const int split = p4Split(nCh, T);
        allowLargeLds(k_rfft_fwd_ols_p4, wgLdsBytes(P));
        hipLaunchKernelGGL(k_rfft_fwd_ols_p4, dim3(nCh * split), dim3(512), wgLdsBytes(P), stream, in, chStride, histOld,
                           histNew, X, XDN, tw, T, split, head, ringSlots - 1);
    } else if (P >= 1024) {
        allowLargeLds(k_rfft_fwd_ols_wg, wgLdsBytes(P));
        hipLaunchKernelGGL(k_rfft_fwd_ols_wg, dim3(nCh * T), dim3(P / 8), wgLdsBytes(P), stream, in, chStride,
                           histOld, histNew, X, XDN, tw, P, T, head, ringSlots - 1);
    } else
        hipLaunchKernelGGL(k_rfft_fwd_ols_generic, dim3(nCh * T), dim3(genericThreads(P)), 2 * P * sizeof(double2),
                           stream, in, chStride, histOld, histNew, X, XDN, tw, P, T, head, ringSlots - 1);
}

bool rfft_fwd_can_carry_side(int P, int nSide, const int64_t* stride, const int64_t* off, int nTab)
{
    if (P != kP || nSide < 0 || nSide > 2 || nTab < 0 || nTab > kGatherTabMax || nTab > 64) return false;
    for (int a = 0; a < nSide; ++a) if ((off[a] | stride[a]) & 1) return false;      // 16-byte stores into every channel's row
    return true;
}

void launch_rfft_fwd_ols_side(hipStream_t stream, const double* in, int64_t chStride, const double* histOld, double* histNew, double2* X,
                              double2* XDN, FftTables tw, int nCh, int T, int head, int ringSlots, int nSide, double* const* dst,
                              const int64_t* dstStride, const int64_t* dstOff, long long* tabDst, const long long* tab, int nTab,
                              double* tailDst, int64_t tailStride, int tailLen)
{
    FwdSide s{};
    s.tailDst = tailDst; s.tailStride = tailStride; s.tailLen = tailDst ? tailLen : 0;
    s.n = nSide;
    for (int a = 0; a < nSide; ++a) { s.dst[a] = dst[a]; s.stride[a] = dstStride[a]; s.off[a] = dstOff[a]; }
    if (tabDst && tab && nTab > 0) { s.tabDst = tabDst; s.nTab = nTab; for (int i = 0; i < nTab; ++i) s.tab[i] = tab[i]; }
    hipLaunchKernelGGL(k_rfft_fwd_ols<true>, dim3(nCh * T), dim3(64), 0, stream, in, chStride, histOld, histNew, X, XDN, tw, T, head,
                       ringSlots - 1, s);
}

void launch_ir_spectra(hipStream_t stream, const double* heff, int heffLen, double2* H, double2* HDN, FftTables tw,
                       int P, int nParts, double2* scratch)
{
    if (P > 4096) {         // scratch [nParts][P]
        const int M1 = P >> 9;
        hipLaunchKernelGGL(k_big_cols_fwd<false>, dim3(nParts * (512 / bigCols(M1))), dim3(bigCols(M1) * M1 / 8), (size_t)M1 * bigCols(M1) * sizeof(double2),
                           stream, heff, 0, nullptr, nullptr, heffLen, scratch, tw, P, 1);
        hipLaunchKernelGGL(k_big_rows_fwd, dim3(nParts * ((M1 >> 1) + 1)), dim3(128), 0, stream, scratch, H, HDN, tw, P, 1, 0,
                           0, 0);
        return;
    }
    if (P == kP)
        hipLaunchKernelGGL(k_ir_spectra, dim3(nParts), dim3(64), 0, stream, heff, heffLen, H, HDN, tw);
    else if (P == kP4) {
        allowLargeLds(k_ir_spectra_p4, wgLdsBytes(P));
        hipLaunchKernelGGL(k_ir_spectra_p4, dim3(nParts), dim3(512), wgLdsBytes(P), stream, heff, heffLen, H, HDN, tw);
    } else if (P >= 1024) {
        allowLargeLds(k_ir_spectra_wg, wgLdsBytes(P));
        hipLaunchKernelGGL(k_ir_spectra_wg, dim3(nParts), dim3(P / 8), wgLdsBytes(P), stream, heff, heffLen, H, HDN,
                           tw, P);
    } else
        hipLaunchKernelGGL(k_ir_spectra_generic, dim3(nParts), dim3(genericThreads(P)), 2 * P * sizeof(double2), stream,
                           heff, heffLen, H, HDN, tw, P);
}

void launch_spectrum_gain(hipStream_t stream, double2* H, double2* HDN, const double* gain, int P, int nParts)
{
    hipLaunchKernelGGL(k_spectrum_gain, dim3(nParts), dim3(256), 0, stream, H, HDN, gain, P);
}

namespace {
template <int MODE>
void launch_inv(hipStream_t stream, const double2* Y, double* out, int64_t chStride, FftTables tw, int P, int nCh, int T,
                double2* scratch, OutSpec ro)
{
    if (P > 4096) {         // scratch [nCh * T][P]
        const int M1 = P >> 9;
        hipLaunchKernelGGL(k_big_rows_inv, dim3(nCh * T * ((M1 >> 1) + 1)), dim3(128), 0, stream, Y, scratch, tw, P);
        hipLaunchKernelGGL(k_big_cols_inv<MODE>, dim3(nCh * T * (512 / bigCols(M1))), dim3(bigCols(M1) * M1 / 8), (size_t)M1 * bigCols(M1) * sizeof(double2), stream,
                           scratch, out, chStride, tw, P, T, ro);
        return;
    }
    if (P == kP)
        hipLaunchKernelGGL(k_rfft_inv_ols<MODE>, dim3(nCh * T), dim3(64), 0, stream, Y, out, chStride, tw, T, ro);
    else if (P == kP4) {
        const int split = p4Split(nCh, T);
        allowLargeLds(k_rfft_inv_ols_p4<MODE>, wgLdsBytes(P));
        hipLaunchKernelGGL(k_rfft_inv_ols_p4<MODE>, dim3(nCh * split), dim3(512), wgLdsBytes(P), stream, Y, out, chStride, tw, T,
                           split, ro);
    } else if (P >= 1024) {
        allowLargeLds(k_rfft_inv_ols_wg<MODE>, wgLdsBytes(P));
        hipLaunchKernelGGL(k_rfft_inv_ols_wg<MODE>, dim3(nCh * T), dim3(P / 8), wgLdsBytes(P), stream, Y, out, chStride,
                           tw, P, T, ro);
    } else
        hipLaunchKernelGGL(k_rfft_inv_ols_generic<MODE>, dim3(nCh * T), dim3(genericThreads(P)), 2 * P * sizeof(double2),
                           stream, Y, out, chStride, tw, P, T, ro);
}
}  // namespace

void launch_rfft_inv_ols(hipStream_t stream, const double2* Y, double* out, int64_t chStride, FftTables tw, int P,
                         int nCh, int T, double2* scratch)
{
    launch_inv<0>(stream, Y, out, chStride, tw, P, nCh, T, scratch, OutSpec{});
}

void launch_rfft_inv_ols_add(hipStream_t stream, const double2* Y, double* out, int64_t chStride, FftTables tw, int nCh, int T,
                             const double* ringA, int ringSizeA, const long long* schedA, double gainA,
                             const double* ringB, int ringSizeB, const long long* schedB, double gainB)
{
    OutSpec os{};
    os.ring = const_cast<double*>(ringA); os.mask = ringSizeA - 1; os.pos = schedA; os.g1 = gainA;
    os.ringB = ringB; os.maskB = ringSizeB - 1; os.schedB = schedB; os.g2 = gainB;
    hipLaunchKernelGGL(k_rfft_inv_ols<3>, dim3(nCh * T), dim3(64), 0, stream, Y, out, chStride, tw, T, os);
}

void launch_rfft_inv_ols_ring(hipStream_t stream, const double2* Y, double* ring, int ringSize, const long long* pos,
                              long long pos0, FftTables tw, int P, int nCh, int T, double2* scratch)
{
    OutSpec os{};
    os.ring = ring; os.mask = ringSize - 1; os.pos = pos; os.pos0 = pos0;
    launch_inv<1>(stream, Y, nullptr, 0, tw, P, nCh, T, scratch, os);
}

void launch_rfft_inv_ols_tail(hipStream_t stream, const double2* Y, double* out, int64_t chStride, FftTables tw, int P, int nCh,
                              int T, double2* scratch, const double* layerOut, const double* tailRing, int tailRingSize,
                              const void* tailState, const long long* sched, int nCallbacks, int B, int nTail, double g1, double g2)
{
    OutSpec os{};
    os.layerOut = layerOut; os.tailRing = tailRing; os.tailState = reinterpret_cast<const long long*>(tailState); os.sched = sched;
    os.log2B = 0;
    while ((1 << os.log2B) < B) ++os.log2B;
    os.nCb = nCallbacks; os.B = B; os.tailMask = tailRingSize - 1; os.nTail = nTail; os.nChAll = nCh; os.nSamples = (int)chStride;
    os.g1 = g1; os.g2 = g2;
    launch_inv<2>(stream, Y, out, chStride, tw, P, nCh, T, scratch, os);
}

}  // namespace cpq
