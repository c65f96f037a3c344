// engine_conv.cpp -- kernel-level convolver: FilterSpec tail layers, the per-call kernel sequence, cpq_conv_* (see engine_internal.hpp, include/convopeq_mi355x.h).
#include "engine_internal.hpp"

using namespace cpqi;

namespace cpqi {

// --- FilterSpec tail layers ---------------------------------------------------------------------------
void freeSpecTails(cpq_engine* e)
{
    for (SpecTail& t : e->specTails) if (t.mem) (void)hipFree(t.mem);
    e->specTails.clear();
    if (e->specState) (void)hipFree(e->specState);
    if (e->specSched) (void)hipFree(e->specSched);
    e->specState = nullptr;
    e->specSched = nullptr;
}

int resetSpecTails(cpq_engine* e)
{
    for (SpecTail& t : e->specTails) {
        CPQ_HIP(e, hipMemsetAsync(t.X, 0, (size_t)e->nCh * t.ringSlots * t.P * sizeof(double2), e->stream));
        CPQ_HIP(e, hipMemsetAsync(t.XDN, 0, (size_t)e->nCh * t.ringSlots * sizeof(double2), e->stream));
        for (int i = 0; i < 2; ++i) {
            CPQ_HIP(e, hipMemsetAsync(t.hist[i], 0, (size_t)e->nCh * t.P * sizeof(double), e->stream));
            CPQ_HIP(e, hipMemsetAsync(t.acc[i], 0, (size_t)e->nCh * t.accCap * sizeof(double), e->stream));
        }
        CPQ_HIP(e, hipMemsetAsync(t.ring, 0, (size_t)e->nCh * t.outRing * sizeof(double), e->stream));
        t.head = t.histSel = t.accSel = t.fill = 0;
        t.blocksDone = 0;
    }
    if (e->specState) CPQ_HIP(e, hipMemsetAsync(e->specState, 0, 3 * sizeof(long long), e->stream));
    return CPQ_OK;
}

int allocSpecTails(cpq_engine* e, const cpq_nuc_plan& pl)
{
    freeSpecTails(e);
    const int64_t nCh = e->nCh;
    const int nMax = e->tMax * e->P;
    for (int l = 1; l < pl.num_layers; ++l) {
        SpecTail t;
        t.P = pl.part_size[l];
        t.K = pl.num_parts_ir[l];
        t.kPad = (int)alignUp(t.K, e->macTile == 32 ? 32 : 16);      // a multiple of every tile the MAC launcher may pick
        t.hRows = t.kPad + 16;                                      // zero rows for the kernels' 4-row read-ahead
        t.nbMax = (t.P - 1 + nMax) / t.P;
        t.ringSlots = nextPow2(t.kPad + cpq::kMacMaxTile + t.nbMax);
        t.accCap = t.P + nMax;
        t.gain = pl.gain[l];
        // the reader is at most outputDelay + one partition behind the writer
        t.outRing = nextPow2(pl.output_delay[l] + 3 * t.P + nMax + e->B);
        struct Item { void** ptr; int64_t bytes; };
        Item items[] = {
            { (void**)&t.X, nCh * t.ringSlots * t.P * (int64_t)sizeof(double2) },
            { (void**)&t.XDN, nCh * t.ringSlots * (int64_t)sizeof(double2) },
            { (void**)&t.H, nCh * t.hRows * t.P * (int64_t)sizeof(double2) },
            { (void**)&t.HDN, nCh * t.hRows * (int64_t)sizeof(double2) },
            { (void**)&t.Y, nCh * t.nbMax * t.P * (int64_t)sizeof(double2) },
            { (void**)&t.tw, t.P * (int64_t)sizeof(double2) },
            { (void**)&t.tw2, t.P * (int64_t)sizeof(double2) },
            { (void**)&t.hist[0], nCh * t.P * (int64_t)sizeof(double) },
            { (void**)&t.hist[1], nCh * t.P * (int64_t)sizeof(double) },
            { (void**)&t.acc[0], nCh * t.accCap * (int64_t)sizeof(double) },
            { (void**)&t.acc[1], nCh * t.accCap * (int64_t)sizeof(double) },
            { (void**)&t.z, nCh * t.nbMax * t.P * (int64_t)sizeof(double) },
            { (void**)&t.ring, nCh * t.outRing * (int64_t)sizeof(double) },
            { (void**)&t.gainDev, (t.P + 1) * (int64_t)sizeof(double) },
            { (void**)&t.scratch, (t.P > 4096 ? std::max<int64_t>(nCh * t.nbMax, t.K) * t.P * (int64_t)sizeof(double2) : 256) },
        };
        int64_t total = 0;
        for (const Item& it : items) total += alignUp(it.bytes, 256);
        if (hipMalloc((void**)&t.mem, (size_t)total) != hipSuccess) {
            (void)hipGetLastError();
            freeSpecTails(e);
            return fail(e, CPQ_ERR_OOM, "FilterSpec tail layer %d: %lld bytes could not be allocated", l, (long long)total);
        }
        int64_t off = 0;
        for (const Item& it : items) { *it.ptr = t.mem + off; off += alignUp(it.bytes, 256); }
        e->specTails.push_back(t);
        CPQ_HIP(e, hipMemsetAsync(t.mem, 0, (size_t)total, e->stream));
        std::vector<double2> w(t.P), w2(t.P);
        const long double twoPi = 6.283185307179586476925286766559005768L;
        for (int m = 0; m < t.P; ++m) {
            const long double a = -twoPi * m / (long double)t.P, b = -twoPi * m / (long double)(2 * t.P);
            w[m] = make_double2((double)cosl(a), (double)sinl(a));
            w2[m] = make_double2((double)cosl(b), (double)sinl(b));
        }
        CPQ_HIP(e, hipStreamSynchronize(e->stream));
        CPQ_HIP(e, hipMemcpy(t.tw, w.data(), t.P * sizeof(double2), hipMemcpyHostToDevice));
        CPQ_HIP(e, hipMemcpy(t.tw2, w2.data(), t.P * sizeof(double2), hipMemcpyHostToDevice));
    }
    if (hipMalloc(&e->specState, 3 * sizeof(long long)) != hipSuccess ||
        hipMalloc((void**)&e->specSched, sizeof(long long) * 2 * (size_t)(nMax / e->B)) != hipSuccess) {
        (void)hipGetLastError();
        freeSpecTails(e);
        return fail(e, CPQ_ERR_OOM, "FilterSpec tail schedule buffers could not be allocated");
    }
    CPQ_HIP(e, hipMemset(e->specState, 0, 3 * sizeof(long long)));
    e->specPlan = pl;
    return CPQ_OK;
}

// the call's input joins every tail layer's accumulator (inputAccBuf, NUC.cpp:1433-1452); must run before the main
// path writes dOut, which may alias dIn
void specTailsAppend(cpq_engine* e, const double* dIn, int n)
{
    for (SpecTail& t : e->specTails)
        cpq::launch_rows_copy(e->stream, dIn, n, 0, t.acc[t.accSel], t.accCap, t.fill, n, e->nCh);
}

// every partition that filled up is convolved (FFT, FDL push, MAC over the layer's partitions, IFFT; NUC.cpp:1456-1544)
// and appended to the layer's delay line; the reference's reader (:1653-1688: readStart = max(readCursor, writeCursor -
// outputDelay), skip when the writer is not far enough ahead) is replayed per callback by k_tail_schedule, so both the
// constant-lag (LTI) plans and the block-skipping ones come out as in the reference
int specTailsRun(cpq_engine* e, double* dOut, int n)
{
    const cpq_nuc_plan& pl = e->specPlan;
    const int nTail = (int)e->specTails.size();
    const int T = n / e->B;
    {
        ProfScope p(e, CPQ_K_MIX);
        const int ppc1 = pl.parts_per_callback[1], ppc2 = nTail > 1 ? pl.parts_per_callback[2] : 1;
        const int d1 = (pl.num_parts_ir[1] + ppc1 - 1) / ppc1 - 1;
        const int d2 = nTail > 1 ? (pl.num_parts_ir[2] + ppc2 - 1) / ppc2 - 1 : 0;
        cpq::launch_tail_schedule(e->stream, e->specState, e->specSched, T, e->B, nTail, pl.part_size[1], pl.output_delay[1], d1,
                                  nTail > 1 ? pl.part_size[2] : e->B, nTail > 1 ? pl.output_delay[2] : 0, d2);
    }
    int li = 0;
    for (SpecTail& t : e->specTails) {
        const int total = t.fill + n;
        const int nb = total / t.P;
        const int rem = total - nb * t.P;
        if (nb > 0) {
            const cpq::FftTables tw{ t.tw, t.tw2 };
            {
                ProfScope p(e, CPQ_K_RFFT_FWD);
                cpq::launch_rfft_fwd_ols(e->stream, t.acc[t.accSel], t.accCap, t.hist[t.histSel], t.hist[t.histSel ^ 1], t.X,
                                         t.XDN, tw, t.P, e->nCh, nb, t.head, t.ringSlots, t.scratch);
            }
            {
                ProfScope p(e, CPQ_K_FDL_MAC);
                cpq::launch_fdl_mac(e->stream, e->macTile, t.X, t.H, e->irSlot, t.Y, t.P, e->nCh, t.kPad, t.ringSlots, t.head,
                                    nb, (int64_t)t.hRows * t.P, e->irPrivate);
            }
            if (cpq::fdl_mac_needs_dcnyq(e->macTile, nb)) {      // the cooperative kernel produces the packed bin itself
                ProfScope p(e, CPQ_K_DCNYQ);
                cpq::launch_fdl_mac_dcnyq(e->stream, t.XDN, t.HDN, e->irSlot, t.Y, t.P, e->nCh, t.K, t.ringSlots, t.head, nb,
                                          t.hRows);
            }
            {
                ProfScope p(e, CPQ_K_RFFT_INV);
                cpq::launch_rfft_inv_ols(e->stream, t.Y, t.z, (int64_t)t.nbMax * t.P, tw, t.P, e->nCh, nb, t.scratch);
            }
            ProfScope p(e, CPQ_K_MIX);
            cpq::launch_ring_put(e->stream, t.z, (int64_t)t.nbMax * t.P, nb * t.P, t.ring, t.outRing, t.blocksDone * t.P,
                                 e->nCh);
            cpq::launch_rows_copy(e->stream, t.acc[t.accSel], t.accCap, (int64_t)nb * t.P, t.acc[t.accSel ^ 1], t.accCap, 0, rem,
                                  e->nCh);
            t.blocksDone += nb;
            t.head = (t.head + nb) & (t.ringSlots - 1);
            t.histSel ^= 1;
            t.accSel ^= 1;
        }
        t.fill = rem;
        ProfScope p(e, CPQ_K_MIX);
        cpq::launch_ring_add(e->stream, dOut, n, n, e->B, t.ring, t.outRing, e->specSched + (size_t)li * T, t.gain, e->nCh);
        ++li;
    }
    CPQ_HIP(e, hipGetLastError());
    return CPQ_OK;
}

// --- enqueue helpers (device pointers, no sync) -----------------------------------------------------
int enqueueConv(cpq_engine* e, const double* dIn, double* dOut, int T)
{
    if (!cpq_conv_is_ready(e)) return fail(e, CPQ_ERR_NOT_READY, "set_impulse has not covered every stream");
    const int64_t stride = (int64_t)T * e->P;
    if (!e->specTails.empty()) specTailsAppend(e, dIn, (int)stride);
    if (e->anyDirect) {       // before anything writes dOut, which may alias dIn
        ProfScope p(e, CPQ_K_MIX);
        cpq::launch_direct_head(e->stream, dIn, stride, (int)stride, e->directIr, e->directTaps, e->irSlot,
                                e->directHist[e->directSel], e->directHist[e->directSel ^ 1], e->directOut, e->nCh);
        e->directSel ^= 1;
    }
    auto addDirect = [&]() {
        if (!e->anyDirect) return;
        ProfScope p(e, CPQ_K_MIX);
        cpq::launch_rows_add(e->stream, dOut, stride, e->directOut, (int)stride, e->nCh);
    };
    if (e->layered) {
        const cpq_nuc_plan& pl = e->layerPlan;
        const int nTail = pl.num_layers - 1;
        {
            ProfScope p(e, CPQ_K_RFFT_FWD);
            cpq::launch_rfft_fwd_ols(e->stream, dIn, stride, e->hist[e->histSel], e->hist[e->histSel ^ 1], e->X, e->XDN,
                                     tables(e), e->P, e->nCh, T, e->head, e->ringSlots);
        }
        for (int l = 0; l < pl.num_layers; ++l) {
            const int kTile = cpq::fdl_mac_kpad_align(e->macTile, T);
            const int kPad = (int)alignUp(e->layerK[l], kTile);
            double* dst = (l == 0) ? dOut : e->layerOut + (int64_t)(l - 1) * e->nCh * stride;
            {
                ProfScope p(e, CPQ_K_FDL_MAC);
                cpq::launch_fdl_mac(e->stream, e->macTile, e->X, e->H + (int64_t)e->layerRow[l] * e->P, e->irSlot, e->Y,
                                    e->P, e->nCh, kPad, e->ringSlots, e->head, T, (int64_t)e->hRows * e->P, e->irPrivate);
            }
            if (cpq::fdl_mac_needs_dcnyq(e->macTile, T)) {
                ProfScope p(e, CPQ_K_DCNYQ);
                cpq::launch_fdl_mac_dcnyq(e->stream, e->XDN, e->HDN + e->layerRow[l], e->irSlot, e->Y, e->P, e->nCh,
                                          e->layerK[l], e->ringSlots, e->head, T, e->hRows);
            }
            {
                ProfScope p(e, CPQ_K_RFFT_INV);
                cpq::launch_rfft_inv_ols(e->stream, e->Y, dst, stride, tables(e), e->P, e->nCh, T);
            }
        }
        {
            ProfScope p(e, CPQ_K_MIX);
            const int ppc1 = pl.parts_per_callback[1], ppc2 = nTail > 1 ? pl.parts_per_callback[2] : 1;
            const int d1 = (pl.num_parts_ir[1] + ppc1 - 1) / ppc1 - 1;
            const int d2 = nTail > 1 ? (pl.num_parts_ir[2] + ppc2 - 1) / ppc2 - 1 : 0;
            cpq::launch_tail_layers(e->stream, e->tailState, e->tailSched, e->layerOut, e->tailRing, dOut, e->nCh,
                                    (int)stride, e->B, e->tailRingSlots, nTail, pl.part_size[1], pl.output_delay[1], d1,
                                    nTail > 1 ? pl.part_size[2] : e->B, nTail > 1 ? pl.output_delay[2] : 0, d2,
                                    pl.gain[1], nTail > 1 ? pl.gain[2] : 0.0);
        }
        addDirect();
        CPQ_HIP(e, hipGetLastError());
        e->head = (e->head + T) & (e->ringSlots - 1);
        e->histSel ^= 1;
        return CPQ_OK;
    }
    {
        ProfScope p(e, CPQ_K_RFFT_FWD);
        cpq::launch_rfft_fwd_ols(e->stream, dIn, stride, e->hist[e->histSel], e->hist[e->histSel ^ 1], e->X, e->XDN,
                                 tables(e), e->P, e->nCh, T, e->head, e->ringSlots);
    }
    {
        ProfScope p(e, CPQ_K_FDL_MAC);
        cpq::launch_fdl_mac(e->stream, e->macTile, e->X, e->H, e->irSlot, e->Y, e->P, e->nCh,
                            (int)alignUp(e->kMaxReal, cpq::fdl_mac_kpad_align(e->macTile, T)), e->ringSlots,
                            e->head, T, (int64_t)e->hRows * e->P, e->irPrivate);
    }
    if (cpq::fdl_mac_needs_dcnyq(e->macTile, T)) {      // the cooperative kernel produces the packed (DC, Nyquist) bin itself
        ProfScope p(e, CPQ_K_DCNYQ);
        cpq::launch_fdl_mac_dcnyq(e->stream, e->XDN, e->HDN, e->irSlot, e->Y, e->P, e->nCh, e->kMaxReal, e->ringSlots,
                                  e->head, T, e->hRows);
    }
    {
        ProfScope p(e, CPQ_K_RFFT_INV);
        cpq::launch_rfft_inv_ols(e->stream, e->Y, dOut, stride, tables(e), e->P, e->nCh, T);
    }
    CPQ_HIP(e, hipGetLastError());
    e->head = (e->head + T) & (e->ringSlots - 1);
    e->histSel ^= 1;
    addDirect();          // Get(): direct output first, then the tail layers (src/MKLNonUniformConvolver.cpp:1606-1633)
    if (!e->specTails.empty()) return specTailsRun(e, dOut, (int)stride);
    CPQ_HIP(e, hipGetLastError());
    return CPQ_OK;
}

}  // namespace cpqi

extern "C" {

// --------------------------------------------------------------------------- convolver
int32_t cpq_conv_set_impulse(cpq_engine* e, int32_t stream, const double* irL, const double* irR, int32_t irLen,
                             double scale, int32_t direct, const cpq_filter_spec* spec)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    if (!irL || !irR || irLen <= 0) return fail(e, CPQ_ERR_INVALID_ARG, "null impulse or non-positive length");
    if (stream != CPQ_ALL_STREAMS && (stream < 0 || stream >= e->desc.n_streams))
        return fail(e, CPQ_ERR_INVALID_ARG, "stream %d out of range", stream);
    if (irLen > e->desc.max_ir_len) return fail(e, CPQ_ERR_INVALID_ARG, "ir_len %d > max_ir_len %d", irLen, e->desc.max_ir_len);
    // FilterSpec: the HC/LC gains (and the air-absorption damping) multiply every partition spectrum of every layer at
    // that layer's FFT size (:336-443, :1060-1097), so every layer keeps the reference's own partition size: layer 0 in
    // the main path, each tail layer in a SpecTail (partitions up to 32768; all such IRs of an engine share one plan).
    // CPQ_SCHED_REFERENCE_NUC runs every IR that way (spec or not): the reference's own partition schedule.
    std::vector<double> gains;
    cpq_nuc_plan sp{};
    bool specTails = false;
    const bool nativeNuc = e->desc.schedule == CPQ_SCHED_REFERENCE_NUC;
    const int slotFirst = (stream == CPQ_ALL_STREAMS) ? 0 : 2 * stream;
    if (spec || nativeNuc) {
        if (cpq::computeNucPlan(irLen, e->desc.block_size, direct != 0, spec, &sp) != CPQ_OK)
            return fail(e, CPQ_ERR_INVALID_ARG, "layer plan failed");
        if (e->P != sp.part_size[0])
            return fail(e, CPQ_ERR_UNSUPPORTED, "FilterSpec needs partition_size == the reference layer-0 partition (%d)", sp.part_size[0]);
        if (e->desc.semantics != CPQ_SEM_REFERENCE) return fail(e, CPQ_ERR_INVALID_ARG, "FilterSpec requires reference semantics");
        if (sp.num_layers > 1) {
            if (e->layered) return fail(e, CPQ_ERR_UNSUPPORTED, "engine is in time-varying (layered) mode");
            for (int l = 1; l < sp.num_layers; ++l)
                if (sp.part_size[l] > 32768 || (sp.part_size[l] & (sp.part_size[l] - 1)))
                    return fail(e, CPQ_ERR_UNSUPPORTED, "FilterSpec tail layer %d has partition size %d; supported: powers of two up to 32768",
                                l, sp.part_size[l]);
            if (e->specTails.empty() || std::memcmp(&sp, &e->specPlan, sizeof(sp)) != 0) {
                for (int slot = 0; slot < e->nCh; ++slot)
                    if (e->slotSpecTail[slot] && !(slot == slotFirst || slot == slotFirst + 1) && stream != CPQ_ALL_STREAMS)
                        return fail(e, CPQ_ERR_UNSUPPORTED, "FilterSpec IRs with tail layers must share one layer plan (IR length and spec)");
                CPQ_HIP(e, hipSetDevice(e->device));
                const int rc = allocSpecTails(e, sp);
                if (rc != CPQ_OK) return rc;
                std::fill(e->slotSpecTail.begin(), e->slotSpecTail.end(), 0);
            }
            specTails = true;
        }
        if (spec) cpq::spectrumFilterGains(*spec, 2 * e->P, gains);
    }

    CPQ_HIP(e, hipSetDevice(e->device));
    // Direct head (src/MKLNonUniformConvolver.cpp:689-731): the first min(irLen, partSize0, 32) taps leave the FFT path
    // (zeroed there, :730-731, before the spectra and any FilterSpec gains are formed) and run as a time-domain FIR.
    const int headTaps = direct ? std::min(irLen, std::min(nextPow2(std::max(e->desc.block_size, 64)), 32)) : 0;
    if (direct && !e->directIr) {
        const size_t callSamples = (size_t)e->tMax * e->P;
        if (hipMalloc((void**)&e->directIr, sizeof(double) * 32 * e->nCh) != hipSuccess ||
            hipMalloc((void**)&e->directTaps, sizeof(int) * e->nCh) != hipSuccess ||
            hipMalloc((void**)&e->directHist[0], sizeof(double) * 32 * e->nCh) != hipSuccess ||
            hipMalloc((void**)&e->directHist[1], sizeof(double) * 32 * e->nCh) != hipSuccess ||
            hipMalloc((void**)&e->directOut, sizeof(double) * e->nCh * callSamples) != hipSuccess)
            return fail(e, CPQ_ERR_OOM, "direct-head buffers could not be allocated");
        CPQ_HIP(e, hipMemset(e->directIr, 0, sizeof(double) * 32 * e->nCh));
        CPQ_HIP(e, hipMemset(e->directTaps, 0, sizeof(int) * e->nCh));
        CPQ_HIP(e, hipMemset(e->directHist[0], 0, sizeof(double) * 32 * e->nCh));
        CPQ_HIP(e, hipMemset(e->directHist[1], 0, sizeof(double) * 32 * e->nCh));
        e->directTapsHost.assign(e->nCh, 0);
    }
    const double* irs[2] = { irL, irR };
    // IR slots: stream s owns slots 2s, 2s+1; CPQ_ALL_STREAMS shares slots 0 and 1 between all streams
    const int slotBase = (stream == CPQ_ALL_STREAMS) ? 0 : 2 * stream;
    std::vector<double> heff;
    // does the reference stay LTI for this IR length / block size?  If not (tail partition longer than the IR that
    // precedes it), switch the engine to layered mode: one convolution per layer + replay of the delay-line reader.
    bool wantLayered = false;
    if (e->desc.semantics == CPQ_SEM_REFERENCE && !spec && !nativeNuc) {
        cpq_nuc_plan probe;
        if (cpq::computeNucPlan(irLen, e->desc.block_size, false, nullptr, &probe) != CPQ_OK)
            return fail(e, CPQ_ERR_INVALID_ARG, "layer plan failed");
        wantLayered = !probe.lti_valid && probe.num_layers > 1;
        if (wantLayered) {
            if (e->P != e->B)
                return fail(e, CPQ_ERR_UNSUPPORTED, "time-varying reference semantics need partition_size == block_size");
            bool anyLoaded = false;
            for (char l : e->irLoaded) anyLoaded = anyLoaded || l;
            if (anyLoaded && (!e->layered || std::memcmp(&probe, &e->layerPlan, sizeof(probe)) != 0))
                return fail(e, CPQ_ERR_UNSUPPORTED, "time-varying reference semantics need the same IR length on every stream");
            if (!e->layered) {
                // lazily allocate the per-layer buffers
                const int nTail = probe.num_layers - 1;
                int span = 0;
                for (int l = 1; l < probe.num_layers; ++l) span = std::max(span, probe.output_delay[l] + 2 * probe.part_size[l]);
                e->tailRingSlots = nextPow2(span + 2 * e->B + e->tMax * e->P);
                const size_t callSamples = (size_t)e->tMax * e->P;
                if (hipMalloc((void**)&e->layerOut, sizeof(double) * nTail * e->nCh * callSamples) != hipSuccess ||
                    hipMalloc((void**)&e->tailRing, sizeof(double) * (size_t)nTail * e->nCh * e->tailRingSlots) != hipSuccess ||
                    hipMalloc(&e->tailState, 3 * sizeof(long long)) != hipSuccess ||
                    hipMalloc((void**)&e->tailSched, sizeof(long long) * 2 * (size_t)e->tMax) != hipSuccess)
                    return fail(e, CPQ_ERR_OOM, "layered-mode buffers could not be allocated");
                CPQ_HIP(e, hipMemset(e->tailRing, 0, sizeof(double) * (size_t)nTail * e->nCh * e->tailRingSlots));
                CPQ_HIP(e, hipMemset(e->tailState, 0, 3 * sizeof(long long)));
                e->layerPlan = probe;
                int row = 0;
                for (int l = 0; l < probe.num_layers; ++l) {
                    e->layerRow[l] = row;
                    e->layerK[l] = (probe.len[l] + e->P - 1) / e->P;
                    row += (int)alignUp(e->layerK[l], cpq::kMacMaxTile) + cpq::kMacMaxTile;
                }
                if (row > e->hRows) return fail(e, CPQ_ERR_INVALID_ARG, "layered IR needs %d rows, capacity %d", row, e->hRows);
                e->layered = true;
            }
        } else if (e->layered) {
            return fail(e, CPQ_ERR_UNSUPPORTED, "engine is in time-varying (layered) mode: every IR must share that plan");
        }
    }

    for (int ch = 0; ch < 2; ++ch) {
        cpq_nuc_plan pl;
        int rc;
        const int slot = slotBase + ch;
        double2* Hs = e->H + (int64_t)slot * e->hRows * e->P;
        double2* HDNs = e->HDN + (int64_t)slot * e->hRows;
        if (wantLayered) {
            pl = e->layerPlan;
            CPQ_HIP(e, hipMemsetAsync(Hs, 0, (size_t)e->hRows * e->P * sizeof(double2), e->stream));
            CPQ_HIP(e, hipMemsetAsync(HDNs, 0, (size_t)e->hRows * sizeof(double2), e->stream));
            const bool scaled = std::abs(scale - 1.0) > 1e-12;
            for (int l = 0; l < pl.num_layers; ++l) {
                heff.assign(irs[ch] + pl.offset[l], irs[ch] + pl.offset[l] + pl.len[l]);
                if (scaled) for (double& v : heff) v *= scale;
                if (l == 0 && e->directIr) {
                    double rev[32] = { 0 };
                    for (int i = 0; i < headTaps; ++i) {
                        rev[i] = irs[ch][headTaps - 1 - i] * scale;
                        if (i < (int)heff.size()) heff[(size_t)i] = 0.0;
                    }
                    CPQ_HIP(e, hipMemcpyAsync(e->directIr + slot * 32, rev, sizeof(rev), hipMemcpyHostToDevice, e->stream));
                    CPQ_HIP(e, hipMemcpyAsync(e->directTaps + slot, &headTaps, sizeof(int), hipMemcpyHostToDevice, e->stream));
                    CPQ_HIP(e, hipStreamSynchronize(e->stream));
                    e->directTapsHost[slot] = headTaps;
                }
                CPQ_HIP(e, hipMemcpyAsync(e->heffDev, heff.data(), heff.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
                cpq::launch_ir_spectra(e->stream, e->heffDev, (int)heff.size(), Hs + (int64_t)e->layerRow[l] * e->P,
                                       HDNs + e->layerRow[l], tables(e), e->P, e->layerK[l]);
                CPQ_HIP(e, hipGetLastError());
                CPQ_HIP(e, hipStreamSynchronize(e->stream));
            }
            e->irParts[slot] = e->hRows;
            e->plan = pl;
            e->plan.direct_taps = direct ? std::min(irLen, std::min(pl.part_size[0], 32)) : 0;
            e->planValid = true;
            e->directHead = direct != 0;
            continue;
        }
        // tail rows of this slot left by an earlier FilterSpec IR
        for (SpecTail& t : e->specTails) {
            CPQ_HIP(e, hipMemsetAsync(t.H + (int64_t)slot * t.hRows * t.P, 0, (size_t)t.hRows * t.P * sizeof(double2), e->stream));
            CPQ_HIP(e, hipMemsetAsync(t.HDN + (int64_t)slot * t.hRows, 0, (size_t)t.hRows * sizeof(double2), e->stream));
        }
        if (!e->specTails.empty()) e->slotSpecTail[slot] = specTails ? 1 : 0;
        if (specTails) {
            // layer 0 here, the tail layers below: each on its own partition grid
            pl = sp;
            heff.assign(irs[ch], irs[ch] + sp.len[0]);
            if (std::abs(scale - 1.0) > 1e-12) for (double& v : heff) v *= scale;
        } else if (e->desc.semantics == CPQ_SEM_REFERENCE) {
            rc = cpq::buildHeff(irs[ch], irLen, e->desc.block_size, scale, spec, heff, &pl);
            if (rc != CPQ_OK) return fail(e, rc, "layer plan failed");
            if (!pl.lti_valid)
                return fail(e, CPQ_ERR_UNSUPPORTED,
                            "the reference drops tail blocks for this IR length / block size (time-varying output)");
        } else {
            rc = cpq::computeNucPlan(irLen, e->desc.block_size, false, nullptr, &pl);
            if (rc != CPQ_OK) return fail(e, rc, "layer plan failed");
            heff.assign(irs[ch], irs[ch] + irLen);
            if (std::abs(scale - 1.0) > 1e-12) for (double& v : heff) v *= scale;
        }
        if (e->directIr) {
            double rev[32] = { 0 };
            for (int i = 0; i < headTaps; ++i) {
                rev[i] = irs[ch][headTaps - 1 - i] * scale;                  // m_directIRRev (:716-718)
                if (i < (int)heff.size()) heff[(size_t)i] = 0.0;            // the head leaves the FFT path
            }
            CPQ_HIP(e, hipMemcpyAsync(e->directIr + slot * 32, rev, sizeof(rev), hipMemcpyHostToDevice, e->stream));
            CPQ_HIP(e, hipMemcpyAsync(e->directTaps + slot, &headTaps, sizeof(int), hipMemcpyHostToDevice, e->stream));
            CPQ_HIP(e, hipStreamSynchronize(e->stream));                     // rev / headTaps are stack storage
            e->directTapsHost[slot] = headTaps;
        }
        const int parts = ((int)heff.size() + e->P - 1) / e->P;
        if (parts > e->kCap) return fail(e, CPQ_ERR_INVALID_ARG, "h_eff needs %d partitions, capacity %d", parts, e->kCap);
        CPQ_HIP(e, hipMemcpyAsync(e->heffDev, heff.data(), heff.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
        // stale partitions of a longer previous IR in this slot become zero rows
        if (e->irParts[slot] > parts) {
            CPQ_HIP(e, hipMemsetAsync(Hs + (int64_t)parts * e->P, 0, (size_t)(e->irParts[slot] - parts) * e->P * sizeof(double2), e->stream));
            CPQ_HIP(e, hipMemsetAsync(HDNs + parts, 0, (size_t)(e->irParts[slot] - parts) * sizeof(double2), e->stream));
        }
        cpq::launch_ir_spectra(e->stream, e->heffDev, (int)heff.size(), Hs, HDNs, tables(e), e->P, parts);
        if (!gains.empty()) {
            CPQ_HIP(e, hipMemcpyAsync(e->gainDev, gains.data(), gains.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
            cpq::launch_spectrum_gain(e->stream, Hs, HDNs, e->gainDev, e->P, parts);
        }
        CPQ_HIP(e, hipGetLastError());
        CPQ_HIP(e, hipStreamSynchronize(e->stream));   // heffDev is reused for the next channel
        if (specTails) {
            std::vector<double> g;
            int l = 1;
            for (SpecTail& t : e->specTails) {
                heff.assign(irs[ch] + sp.offset[l], irs[ch] + sp.offset[l] + sp.len[l]);
                if (std::abs(scale - 1.0) > 1e-12) for (double& v : heff) v *= scale;
                double2* Ht = t.H + (int64_t)slot * t.hRows * t.P;
                double2* HDNt = t.HDN + (int64_t)slot * t.hRows;
                CPQ_HIP(e, hipMemcpyAsync(e->heffDev, heff.data(), heff.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
                cpq::launch_ir_spectra(e->stream, e->heffDev, (int)heff.size(), Ht, HDNt, cpq::FftTables{ t.tw, t.tw2 }, t.P, t.K,
                                       t.scratch);
                if (spec) {
                    cpq::spectrumFilterGains(*spec, 2 * t.P, g);        // applySpectrumFilter at this layer's FFT size
                    CPQ_HIP(e, hipMemcpyAsync(t.gainDev, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
                    cpq::launch_spectrum_gain(e->stream, Ht, HDNt, t.gainDev, t.P, t.K);
                }
                CPQ_HIP(e, hipStreamSynchronize(e->stream));
                if (spec && cpq::airAbsorptionGains(*spec, l, t.P + 1, g)) {     // tail mode 0 (:1060-1097)
                    CPQ_HIP(e, hipMemcpyAsync(t.gainDev, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
                    cpq::launch_spectrum_gain(e->stream, Ht, HDNt, t.gainDev, t.P, t.K);
                    CPQ_HIP(e, hipStreamSynchronize(e->stream));
                }
                CPQ_HIP(e, hipGetLastError());
                ++l;
            }
        }
        e->irParts[slot] = parts;
        e->plan = pl;
        e->plan.direct_taps = direct ? std::min(irLen, std::min(pl.part_size[0], 32)) : 0;
        e->planValid = true;
        e->directHead = direct != 0;
    }
    if (stream == CPQ_ALL_STREAMS) {
        for (int c = 0; c < e->nCh; ++c) { e->irSlotHost[c] = c & 1; e->irLoaded[c] = 1; }
    } else {
        for (int ch = 0; ch < 2; ++ch) { e->irSlotHost[2 * stream + ch] = 2 * stream + ch; e->irLoaded[2 * stream + ch] = 1; }
    }
    CPQ_HIP(e, hipMemcpy(e->irSlot, e->irSlotHost.data(), sizeof(int) * e->nCh, hipMemcpyHostToDevice));
    e->irPrivate = true;
    for (int c = 0; c < e->nCh; ++c) e->irPrivate = e->irPrivate && e->irSlotHost[c] == c;
    e->anyDirect = false;
    if (e->directIr)
        for (int c = 0; c < e->nCh; ++c) if (e->irLoaded[c] && e->directTapsHost[e->irSlotHost[c]] > 0) e->anyDirect = true;
    int kMax = 0;
    for (int c = 0; c < e->nCh; ++c) if (e->irLoaded[c]) kMax = std::max(kMax, e->irParts[e->irSlotHost[c]]);
    e->kMaxReal = kMax;
    e->kActive = (int)alignUp(kMax, cpq::kMacMaxTile);
    return CPQ_OK;
}

int32_t cpq_conv_is_ready(const cpq_engine* e)
{
    if (!e) return 0;
    for (char l : e->irLoaded) if (!l) return 0;
    return 1;
}

int32_t cpq_conv_latency(const cpq_engine* e) { return (e && e->planValid) ? e->plan.latency : 0; }

int32_t cpq_conv_get_plan(const cpq_engine* e, cpq_nuc_plan* plan)
{
    if (!e || !plan || !e->planValid) return CPQ_ERR_NOT_READY;
    *plan = e->plan;
    return CPQ_OK;
}

int32_t cpq_conv_reset(cpq_engine* e) { return e ? zeroRuntimeState(e, true, false) : CPQ_ERR_INVALID_ARG; }

int32_t cpq_conv_process_device(cpq_engine* e, const double* dIn, double* dOut, int32_t nSamples)
{
    int T = 0;
    const int rc = checkCall(e, dIn, dOut, nSamples, &T);
    if (rc != CPQ_OK) return rc;
    CPQ_HIP(e, hipSetDevice(e->device));
    return enqueueConv(e, dIn, dOut, T);
}

int32_t cpq_conv_process(cpq_engine* e, const double* in, double* out, int32_t nSamples)
{
    return viaStaging(e, in, out, nSamples, [e](const double* a, double* b, int T) { return enqueueConv(e, a, b, T); });
}

}  // extern "C"
