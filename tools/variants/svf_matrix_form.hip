// tools/variants/svf_matrix_form.hip -- NOT built.  The matrix-form ("MFMA") band loop and kernels of rounds 1-2
// (k_svf_cascade_tp8<8|16>, k_svf_cascade_tpw), kept for the record after the vector-form kernel replaced them in round 3
// (EQ alone, 256 streams x 524288 samples: 6.29-6.33 ms against 5.60-5.64 ms, profiles/r03b_ab_eq_forms.txt).  The code
// is the part cut out of convopeq_amd/csrc/svf_kernels.hip at commit "vector-form EQ kernel"; it refers to helpers that
// stay there (TpBandTables, tp_scan, tp_nonlinear*, tp_band_guarded, wave_lds_sync, dpp_f64, kTpStride ...).
// The CPQ_ABL / CPQ_TX_* macros are the timing ablations and product-order variants of round 2 (tools/ablate_svf.sh).

// ---------------------------------------------------------------------------------------------------------
// MFMA form of the main path (chunk length 16).  Inside one band everything between the input and the output stage
// is linear, and over a chunk of 16 samples it is a small dense product shared by all chunks of the span:
//     y_lin (16 x chunks) = [ T | G ] (16 x 18)  .  [ x ; s0 ] (18 x chunks),     e (2 x chunks) = E (2 x 16) . x
// with T the lower-triangular Toeplitz matrix of the band's zero-state impulse response, G = C A^i the state
// response and s0 the chunk start states from the scan of the end states e.  That product runs on the matrix cores
// (v_mfma_f64_16x16x4_f64: 5 per tile of 16 chunks), which are idle otherwise and issue beside the VALU.
// Register layout = the instruction's own: lane (m = lane & 15, g = lane >> 4) holds of tile tau (16 chunks) the
// samples g + 4 j (j = register) of chunk 16 tau + m.  The D registers of one band ARE the B operands of the next
// (k-step s = register s), so the span stays in registers across the 20 bands; the output stage is element-wise.
typedef double v4d __attribute__((ext_vector_type(4)));

struct alignas(16) TpLdsM {
    double cf[kBands][6];        // a1 a2 a3 m0 m1 m2 (guarded fallback)
    double M[kBands][28];        // Mk[6][4], Mw[4] (scan)
    double Gq[kBands][16][4];    // (C A^i)_x, (C A^i)_y, 0, 0: A-operand rows of the state response
    double ht[kBands][32];
    double e[kBands][2][16];
};

// tables of the matrix form -> LDS (nThreads = threads of the workgroup); the caller synchronises
__device__ __forceinline__ void tp_load_tables_m(TpLdsM* L, const double* __restrict__ cf,
                                                 const TpBandTables* __restrict__ tb, int tid, int nThreads)
{
    for (int i = tid; i < kBands * 6; i += nThreads) L->cf[i / 6][i % 6] = cf[i];
    for (int i = tid; i < kBands * 28; i += nThreads) {
        const int b = i / 28, q = i % 28;
        L->M[b][q] = (q < 24) ? tb[b].t[0].Mk[q / 4][q % 4] : tb[b].t[0].Mw[q - 24];
    }
    for (int i = tid; i < kBands * 64; i += nThreads) {
        const int b = i / 64, r = (i % 64) / 4, q = i % 4;
        L->Gq[b][r][q] = (q < 2) ? tb[b].t[0].G[r][q] : 0.0;
    }
    for (int i = tid; i < kBands * 32; i += nThreads) {
        L->ht[i / 32][i % 32] = tb[i / 32].mm.ht[i % 32];
        L->e[i / 32][(i % 32) / 16][i % 16] = tb[i / 32].mm.e[(i % 32) / 16][i % 16];
    }
}

// The band loop of the matrix form: x = the wave's 64 chunks in the MFMA layout (in and out); red / s0q = the wave's LDS
// scratch (256 double2 / 256 doubles); NTHREADS = threads of the workgroup (64 per wave of the span; 0 = blockDim.x).
// wtot: two parities of [2 * waves] wave totals (one workgroup barrier per band, see tp_scan).  A barrier-free variant
// (totals published with per-band flags, waves polling only their predecessors) measured slower: 0.59 vs 0.58 ms.
template <bool SAT, int NTHREADS>
__device__ __forceinline__ void tp_bands_mfma(v4d (&x)[4], double2* red, double* s0q, double* wtot, const double* sState,
                                              double* sNext, const TpLdsM* L, int tid, const int* __restrict__ fl,
                                              const TpBandTables* __restrict__ tb, double sat)
{
    const int lane = tid & 63;
    const int m = lane & 15, g = lane >> 4;
    const double oneMinusSat = 1.0 - sat;
    const bool smallOk = (sat >= 0.0) && (sat <= 1.0);
    const double smallC1 = 9.0 - 8.0 * sat;
        int par = 0;
        for (int b = 0; b < kBands; ++b) {
            const int flag = fl[b];
            if (!(flag & 1)) continue;                    // uniform
            const int kind = (flag >> 1) & 3;
            // tables of the band
            const TpLanePowers pw = {};                       // register-tight: tp_scan loads the per-lane powers where it uses them
            double a[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) a[s4] = L->ht[b][15 + m - 4 * s4 - g];
            // (1) end state of every chunk's zero-state run: e = E x, partial over this lane's four samples per tile ...
            double e0[4], e1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { e0[j] = L->e[b][0][g + 4 * j]; e1[j] = L->e[b][1][g + 4 * j]; }
#pragma unroll
            for (int tau = 0; tau < 4; ++tau) {
                double px = e0[0] * x[tau][0], py = e1[0] * x[tau][0];
#pragma unroll
                for (int j = 1; j < 4; ++j) { px = fma(e0[j], x[tau][j], px); py = fma(e1[j], x[tau][j], py); }
                red[(tau * 4 + g) * 16 + m] = make_double2(px, py);
            }
            // (2) the zero-state part of the product, T x, does not wait for the start states: its 16 MFMAs go to the
            // matrix pipe now and run beside the reduction and the scan below (x is dead from here: acc takes its place)
#if !(defined(CPQ_ABL) && (CPQ_ABL & 4))
#ifndef CPQ_TX_16X16
            {
                // T is lower-triangular Toeplitz: of its sixteen 4 x 4 blocks only the ten on and below the diagonal are
                // non-zero, and block (i, j) depends on i - j alone.  v_mfma_f64_4x4x4_4b_f64 multiplies one such block
                // into four batches of four chunks; its operand layout (B[k][n] at lane 16 k + n, D[i][n] at lane
                // 16 i + n, A[i][k] at lane 16 k + 4 batch + i: tools/ubench/mfma_f64_4x4x4.hip) is register s of the
                // 16x16x4 layout = block row s, so the two instructions mix freely.  10 small MFMAs (~17-20 cycles each)
                // instead of 4 large ones (64 cycles each) per tile.
                double a4[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) a4[d] = L->ht[b][15 + 4 * d + (m & 3) - g];
                double dacc[4][4];
                // block-column major: consecutive MFMAs write different accumulators
#ifdef CPQ_TX_TILE_MAJOR
#pragma unroll
                for (int tau = 0; tau < 4; ++tau)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int i = j; i < 4; ++i)
                            dacc[tau][i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[i - j], x[tau][j], j == 0 ? 0.0 : dacc[tau][i], 0, 0, 0);
#else
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int tau = 0; tau < 4; ++tau)
#pragma unroll
                        for (int i = j; i < 4; ++i)
                            dacc[tau][i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a4[i - j], x[tau][j], j == 0 ? 0.0 : dacc[tau][i], 0, 0, 0);
#endif
#pragma unroll
                for (int tau = 0; tau < 4; ++tau) x[tau] = v4d{ dacc[tau][0], dacc[tau][1], dacc[tau][2], dacc[tau][3] };
            }
#else
            {
                v4d acc[4];
#pragma unroll
                for (int tau = 0; tau < 4; ++tau) acc[tau] = v4d{ 0.0, 0.0, 0.0, 0.0 };
                // k-step major: consecutive MFMAs belong to different tiles, so none waits for its own accumulator
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int tau = 0; tau < 4; ++tau)
                        acc[tau] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], x[tau][s4], acc[tau], 0, 0, 0);
#pragma unroll
                for (int tau = 0; tau < 4; ++tau) x[tau] = acc[tau];
            }
#endif
#endif
            wave_lds_sync();
            // ... the partial end states summed over the four lane groups; lane l ends up with chunk l of the wave
            double ic1, ic2;
#if defined(CPQ_ABL) && (CPQ_ABL & 1)
            ic1 = e0[0]; ic2 = e1[1];
#else
            {
                const double2 p0 = red[(g * 4 + 0) * 16 + m], p1 = red[(g * 4 + 1) * 16 + m];
                const double2 p2 = red[(g * 4 + 2) * 16 + m], p3 = red[(g * 4 + 3) * 16 + m];
                ic1 = (p0.x + p1.x) + (p2.x + p3.x);
                ic2 = (p0.y + p1.y) + (p2.y + p3.y);
            }
#endif
            // (3) chunk start states
            double s0x, s0y;
#if defined(CPQ_ABL) && (CPQ_ABL & 2)
            s0x = ic1 * e0[1]; s0y = ic2 * e1[0];
#else
            tp_scan<NTHREADS>(ic1, ic2, s0x, s0y, &L->M[0][0], b, pw, wtot + par * 2 * ((NTHREADS ? NTHREADS : (int)blockDim.x) / 64), sState, sNext, tid,
                              &tb[b].t[0].P[0][0]);
            par ^= 1;
#endif
            // (4) the state response G s0 completes the product (k-step 4: rows 16 + g of [x ; s0], staged through the
            // wave's LDS scratch), tile by tile, followed by (5) the element-wise output stage of that tile
            *reinterpret_cast<double2*>(s0q + lane * 4) = make_double2(s0x, s0y);
            *reinterpret_cast<double2*>(s0q + lane * 4 + 2) = make_double2(0.0, 0.0);
            wave_lds_sync();
            const double ag = L->Gq[b][m][g];
            double sb[4];
#pragma unroll
            for (int tau = 0; tau < 4; ++tau) sb[tau] = s0q[(tau * 16 + m) * 4 + g];
#pragma unroll
            for (int tau = 0; tau < 4; ++tau) x[tau] = __builtin_amdgcn_mfma_f64_16x16x4f64(ag, sb[tau], x[tau], 0, 0, 0);
#if defined(CPQ_ABL) && (CPQ_ABL & 8)
            if (false) {
#else
            if (kind != 2) {          // kind 2 (OutputFilter biquad): linear section, no output stage
#endif
#pragma unroll
                for (int tau = 0; tau < 4; ++tau) {
                    double v[4] = { x[tau][0], x[tau][1], x[tau][2], x[tau][3] };
                    // four compares with the |.| modifier (a NaN fails them and takes the guarded code below)
                    const bool small = (int)(fabs(v[0]) < 4.5) & (int)(fabs(v[1]) < 4.5) & (int)(fabs(v[2]) < 4.5) & (int)(fabs(v[3]) < 4.5);
                    if (smallOk && __all(small)) {
                        if (SAT) tp_nonlinear_small<4>(v, smallC1);
                    } else if (kind == 1) tp_nonlinear<true, SAT, 4, false>(v, sat, oneMinusSat);
                    else                  tp_nonlinear<false, SAT, 4, false>(v, sat, oneMinusSat);
                    x[tau] = v4d{ v[0], v[1], v[2], v[3] };
                }
            }
        }
}

// Span I/O of the matrix-form kernels.  The MFMA layout wants lane (m, g) to hold samples g + 4 j (register j) of chunk m;
// loaded as such, one instruction touches an 8-byte word in 16 different 128-byte lines and every 32-byte sector is fetched
// (and written) in pieces: PMC traffic 1.7x the algorithmic bytes.  Instead lane (m, g) moves the whole sector, samples
// 4 g ... 4 g + 3, with two 16-byte accesses and a 4 x 4 transpose across the four 16-lane rows of the wave puts them in
// place (v_permlane32_swap / v_permlane16_swap, tools/ubench/permlane_transpose.hip: 8 VALU instructions per tile).
__device__ __forceinline__ void tp_swap32(double& a, double& b)      // rows 2,3 of a <-> rows 0,1 of b
{
    const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    const auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double(h[0], l[0]);
    b = __hiloint2double(h[1], l[1]);
}
__device__ __forceinline__ void tp_swap16(double& a, double& b)      // rows 1,3 of a <-> rows 0,2 of b
{
    const unsigned alo = __double2loint(a), ahi = __double2hiint(a), blo = __double2loint(b), bhi = __double2hiint(b);
    const auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double(h[0], l[0]);
    b = __hiloint2double(h[1], l[1]);
}
// s[i] at row g = element (g, i)  ->  s[j] at row g = element (j, g); its own inverse
__device__ __forceinline__ void tp_transpose4(double (&s)[4])
{
    tp_swap32(s[0], s[2]);
    tp_swap32(s[1], s[3]);
    tp_swap16(s[0], s[1]);
    tp_swap16(s[2], s[3]);
}
// chunk = the 16 samples of chunk m (128-byte aligned); g = lane >> 4
__device__ __forceinline__ v4d tp_tile_load(const double* chunk, int g)
{
    // streaming accesses: a span is read once and written once per call; the L2 is left to the per-stream scan tables
    typedef double v2 __attribute__((ext_vector_type(2)));
    const v2 a = __builtin_nontemporal_load(reinterpret_cast<const v2*>(chunk + 4 * g));
    const v2 b = __builtin_nontemporal_load(reinterpret_cast<const v2*>(chunk + 4 * g + 2));
    double s[4] = { a.x, a.y, b.x, b.y };
    tp_transpose4(s);
    return v4d{ s[0], s[1], s[2], s[3] };
}
// Stores go through the wave's LDS scratch instead (tile = the 16 chunks x 16 samples = 2 KB at `tile`, buf = 16 rows of
// kTpStride doubles): every store instruction then writes 1 KB of whole 128-byte lines.  Partial-line stores made the L2
// fetch the rest of each line from memory first (PMC: reads 2.3x, writes 1.4x the algorithmic bytes).
__device__ __forceinline__ void tp_tile_store(double* tile, double* buf, int lane, v4d x, double gain)
{
    const int m = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) buf[m * kTpStride + g + 4 * j] = x[j] * gain;
    wave_lds_sync();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = lane + 64 * i;                          // pair of samples: chunk p >> 3, samples 2 (p & 7), + 1
        typedef double v2 __attribute__((ext_vector_type(2)));
        const double2 v = *reinterpret_cast<const double2*>(buf + (p >> 3) * kTpStride + 2 * (p & 7));
        __builtin_nontemporal_store(v2{ v.x, v.y }, reinterpret_cast<v2*>(tile + 2 * p));
    }
    wave_lds_sync();                                          // the next tile reuses buf
}

// ---------------------------------------------------------------------------------------------------------
// Eight waves per channel: spans of 8192 samples (512 chunks of 16).  The band loop is a dependent chain (end states ->
// reduction -> scan -> product -> output stage) that two waves per SIMD do not hide; with twice the waves per channel
// four waves share a SIMD.  No LDS staging of the span (it would not fit twice per CU beside the tables): every lane
// loads and stores its 16 samples straight in the MFMA layout (4 x 8 B per 128-byte line and instruction, the four
// registers of a tile cover the line).  Spans with non-finite / out-of-range input go through the guarded sequential
// code in two 4096-sample halves staged in the scratch area.  Handles whole 8192-sample spans only; the launcher runs
// k_svf_cascade_tp on what is left.
// WAVES = 16 (spans of 16384 samples, one workgroup per CU) is for engines with fewer channels than the chip has CUs:
// at 128 channels the eight-wave kernel leaves half the CUs idle and the other half at two waves per SIMD.
constexpr int kTp8Threads = 512;
constexpr int kTp8Span = kTp8Threads * 16;

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 4) void k_svf_cascade_tp8(const double* in, double* out, int64_t chStride,
                                                                int nSpans, const double* __restrict__ coef,
                                                                const int* __restrict__ flags,
                                                                const double* __restrict__ satGain,
                                                                double* __restrict__ state,
                                                                const TpBandTables* __restrict__ tables)
{
    constexpr int kThreads = WAVES * 64, kSpan = kThreads * 16;
    constexpr int kScratchDoubles = WAVES * (512 + 256);     // per wave: red (256 double2) + s0q (256 doubles); 8 waves: 48 KB
    static_assert(kScratchDoubles >= 256 * kTpStride, "the guarded path stages 4096 samples in the scratch area");
    __shared__ __align__(16) double scratch[kScratchDoubles];
    __shared__ TpLdsM LM;
    __shared__ double sStateA[kBands * 2], sStateB[kBands * 2];
    __shared__ double wtot[2 * 2 * WAVES];
    __shared__ int sFlag;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int c = blockIdx.x;
    const double* cf = coef + (int64_t)c * kBands * 6;
    const int* fl = flags + c * kBands;
    const TpBandTables* tb = tables + (int64_t)(c >> 1) * kBands;
    const double sat = satGain[c * 2], gain = satGain[c * 2 + 1];
    double* sState = sStateA;
    double* sNext = sStateB;
    if (tid < kBands * 2) { sStateA[tid] = state[(int64_t)c * kBands * 2 + tid]; sStateB[tid] = sStateA[tid]; }
    tp_load_tables_m(&LM, cf, tb, tid, kThreads);
    __syncthreads();

    for (int sp = 0; sp < nSpans; ++sp) {
        const double* src = in + (int64_t)c * chStride + (int64_t)sp * kSpan;
        double* dst = out + (int64_t)c * chStride + (int64_t)sp * kSpan;
        v4d x[4];
        bool bad = false;
#pragma unroll
        for (int tau = 0; tau < 4; ++tau) {
            x[tau] = tp_tile_load(src + (wave * 64 + tau * 16 + m) * 16, g);
#pragma unroll
            for (int j = 0; j < 4; ++j) bad |= !(fabs(x[tau][j]) < kTpInputBound);
        }
        if (tid < kBands * 2) bad |= !(fabs(sState[tid]) < kTpInputBound);
        if (tid == 0) sFlag = 0;
        __syncthreads();
        if (__any(bad) && lane == 0) atomicOr(&sFlag, 1);
        __syncthreads();
        if (sFlag != 0) {
            // guarded path: pieces of 4096 samples through the one-thread reference recurrence, staged in the scratch
            // area as [chunk][sample]; in and out may alias, every sample of a piece is read before the piece is written
            for (int half = 0; half < kSpan / 4096; ++half) {
                for (int j = tid; j < 4096; j += kThreads)
                    scratch[(j / 16) * kTpStride + (j % 16)] = src[half * 4096 + j];
                __syncthreads();
                for (int b = 0; b < kBands; ++b) {
                    const int flag = fl[b];
                    if (!(flag & 1)) continue;
                    if (tid == 0) {
                        if (flag & 4)      tp_band_guarded<2>(scratch, 16, LM.cf[b], sat, sState + 2 * b);
                        else if (flag & 2) tp_band_guarded<1>(scratch, 16, LM.cf[b], sat, sState + 2 * b);
                        else               tp_band_guarded<0>(scratch, 16, LM.cf[b], sat, sState + 2 * b);
                    }
                    __syncthreads();
                }
                for (int j = tid; j < 4096; j += kThreads)
                    dst[half * 4096 + j] = scratch[(j / 16) * kTpStride + (j % 16)] * gain;
                __syncthreads();
            }
            continue;
        }
        double2* red = reinterpret_cast<double2*>(scratch) + wave * 256;
        double* s0q = scratch + WAVES * 512 + wave * 256;
        if (sat > 0.0) tp_bands_mfma<true, kThreads>(x, red, s0q, wtot, sState, sNext, &LM, tid, fl, tb, sat);
        else           tp_bands_mfma<false, kThreads>(x, red, s0q, wtot, sState, sNext, &LM, tid, fl, tb, sat);
#pragma unroll
        for (int tau = 0; tau < 4; ++tau)
            tp_tile_store(dst + (wave * 64 + tau * 16) * 16, reinterpret_cast<double*>(red), lane, x[tau], gain);
        __syncthreads();                      // the last thread's end states are in sNext
        { double* t = sState; sState = sNext; sNext = t; }
    }
    __syncthreads();
    if (tid < kBands * 2) state[(int64_t)c * kBands * 2 + tid] = sState[tid];
}

// ---------------------------------------------------------------------------------------------------------
// One to seven waves per channel, ONE span of waves x 1024 samples in the same matrix form: what a call leaves after
// its whole 8192-sample spans (and all of a call of 2 ... 15 blocks of 512).  The cost of a span is the latency of the
// 20-band chain, whatever its length, so the remainder is split over as many waves as it has 1024-sample pieces rather
// than walked through span by span; before, it ran as 4096-sample spans on four waves and then as 512-sample spans in
// VALU form (chunk length 2) at three times the cost per block.
constexpr int kTpwMaxWaves = 7;
constexpr int kTpwScratchDoubles = kTpwMaxWaves * (512 + 256);      // per wave: red (256 double2) + s0q (256 doubles)

__global__ __launch_bounds__(kTpwMaxWaves * 64, 4) void k_svf_cascade_tpw(const double* in, double* out, int64_t chStride,
                                                                       const double* __restrict__ coef,
                                                                       const int* __restrict__ flags,
                                                                       const double* __restrict__ satGain,
                                                                       double* __restrict__ state,
                                                                       const TpBandTables* __restrict__ tables)
{
    static_assert(kTpwScratchDoubles >= 256 * kTpStride, "the guarded path stages up to 4096 samples in the scratch area");
    __shared__ __align__(16) double scratch[kTpwScratchDoubles];
    __shared__ TpLdsM LM;
    __shared__ double sStateA[kBands * 2], sStateB[kBands * 2];
    __shared__ double wtot[2 * 2 * kTpwMaxWaves];
    __shared__ int sFlag;
    const int tid = threadIdx.x, nThreads = blockDim.x;
    const int nWaves = nThreads >> 6;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    const int c = blockIdx.x;
    const double* cf = coef + (int64_t)c * kBands * 6;
    const int* fl = flags + c * kBands;
    const TpBandTables* tb = tables + (int64_t)(c >> 1) * kBands;
    const double sat = satGain[c * 2], gain = satGain[c * 2 + 1];
    if (tid < kBands * 2) { sStateA[tid] = state[(int64_t)c * kBands * 2 + tid]; sStateB[tid] = sStateA[tid]; }
    tp_load_tables_m(&LM, cf, tb, tid, nThreads);
    if (tid == 0) sFlag = 0;
    __syncthreads();

    const double* src = in + (int64_t)c * chStride;
    double* dst = out + (int64_t)c * chStride;
    v4d x[4];
    bool bad = false;
#pragma unroll
    for (int tau = 0; tau < 4; ++tau) {
        x[tau] = tp_tile_load(src + (wave * 64 + tau * 16 + m) * 16, g);
#pragma unroll
        for (int j = 0; j < 4; ++j) bad |= !(fabs(x[tau][j]) < kTpInputBound);
    }
    if (tid < kBands * 2) bad |= !(fabs(sStateA[tid]) < kTpInputBound);
    if (__any(bad) && lane == 0) atomicOr(&sFlag, 1);
    __syncthreads();
    if (sFlag != 0) {
        // guarded path: pieces of up to 4096 samples through the one-thread reference recurrence, staged in the scratch
        // area as [chunk][sample]; in and out may alias, every sample of a piece is read before the piece is written
        const int nSamples = nWaves * 1024;
        for (int base = 0; base < nSamples; base += 4096) {
            const int cnt = (nSamples - base < 4096) ? nSamples - base : 4096;
            for (int j = tid; j < cnt; j += nThreads) scratch[(j / 16) * kTpStride + (j % 16)] = src[base + j];
            __syncthreads();
            for (int b = 0; b < kBands; ++b) {
                const int flag = fl[b];
                if (!(flag & 1)) continue;
                if (tid == 0) {
                    if (flag & 4)      tp_band_guarded<2>(scratch, 16, LM.cf[b], sat, sStateA + 2 * b, cnt / 16);
                    else if (flag & 2) tp_band_guarded<1>(scratch, 16, LM.cf[b], sat, sStateA + 2 * b, cnt / 16);
                    else               tp_band_guarded<0>(scratch, 16, LM.cf[b], sat, sStateA + 2 * b, cnt / 16);
                }
                __syncthreads();
            }
            for (int j = tid; j < cnt; j += nThreads) dst[base + j] = scratch[(j / 16) * kTpStride + (j % 16)] * gain;
            __syncthreads();
        }
        if (tid < kBands * 2) state[(int64_t)c * kBands * 2 + tid] = sStateA[tid];
        return;
    }
    double2* red = reinterpret_cast<double2*>(scratch) + wave * 256;
    double* s0q = scratch + kTpwMaxWaves * 512 + wave * 256;
    if (sat > 0.0) tp_bands_mfma<true, 0>(x, red, s0q, wtot, sStateA, sStateB, &LM, tid, fl, tb, sat);
    else           tp_bands_mfma<false, 0>(x, red, s0q, wtot, sStateA, sStateB, &LM, tid, fl, tb, sat);
#pragma unroll
    for (int tau = 0; tau < 4; ++tau)
        tp_tile_store(dst + (wave * 64 + tau * 16) * 16, reinterpret_cast<double*>(red), lane, x[tau], gain);
    __syncthreads();                          // the span's end states are in sStateB
    if (tid < kBands * 2) state[(int64_t)c * kBands * 2 + tid] = sStateB[tid];
}

