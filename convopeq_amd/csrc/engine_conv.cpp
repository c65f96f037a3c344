// engine_conv.cpp -- kernel-level convolver: FilterSpec tail layers, the per-call kernel sequence, cpq_conv_* (see engine_internal.hpp, include/convopeq_mi355x.h).
#include "engine_internal.hpp"

using namespace cpqi;

namespace cpqi {

// --- enqueue helpers (device pointers, no sync) -----------------------------------------------------
// n samples per channel; the main (uniform) path needs n to be a whole number of partitions, plan groups take any n
int enqueueConv(cpq_engine* e, const double* dIn, double* dOut, int n)
{
    if (!cpq_conv_is_ready(e)) return fail(e, CPQ_ERR_NOT_READY, "set_impulse has not covered every stream");
    const int64_t stride = n;
    const int T = n / e->P;
    e->lastCallSamples = n;
    if (e->mainActive && (int64_t)T * e->P != n)
        return fail(e, CPQ_ERR_INVALID_ARG, "n_samples=%d is not a multiple of the partition size %d (streams on the uniform path)", n, e->P);
    // before anything writes dOut, which may alias dIn: the call's input joins the plan groups' accumulators and the direct head runs
    if (!e->groups.empty()) { const int rc = groupsAppend(e, dIn, n); if (rc != CPQ_OK) return rc; }
    if (e->anyDirect) {
        ProfScope p(e, CPQ_K_MIX);
        // a stream whose convolver rests (per-stream bypass / dry-only at the processor level) keeps its head history: the
        // reference's processDirectBlock sits inside Add(), which ConvolverProcessor does not call then
        cpq::launch_direct_head(e->stream, dIn, stride, (int)stride, e->directIr, e->directTaps, e->irSlot,
                                e->directHist[e->directSel], e->directHist[e->directSel ^ 1], e->directOut, e->nCh,
                                e->honourFrozen ? e->procWetOn : nullptr);
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
        // the tail layers first (natural time, into layerOut), then the reader's schedule for the call's callbacks, then layer
        // 0, whose inverse transform adds what the reader reads as it stores the output (no separate pass over the output)
        for (int l = pl.num_layers - 1; l >= 0; --l) {
            {
                ProfScope p(e, CPQ_K_FDL_MAC);
                cpq::launch_fdl_mac(e->stream, e->macTile, e->X, e->H + (int64_t)e->layerRow[l] * e->P, e->irSlot, e->Y,
                                    e->P, e->nCh, e->layerK[l], e->ringSlots, e->head, T, (int64_t)e->hRows * e->P, e->irPrivate);
            }
            if (cpq::fdl_mac_needs_dcnyq(e->macTile, T)) {
                ProfScope p(e, CPQ_K_DCNYQ);
                cpq::launch_fdl_mac_dcnyq(e->stream, e->XDN, e->HDN + e->layerRow[l], e->irSlot, e->Y, e->P, e->nCh,
                                          e->layerK[l], e->ringSlots, e->head, T, e->hRows);
            }
            if (l > 0) {
                ProfScope p(e, CPQ_K_RFFT_INV);
                cpq::launch_rfft_inv_ols(e->stream, e->Y, e->layerOut + (int64_t)(l - 1) * e->nCh * stride, stride, tables(e), e->P, e->nCh, T);
                continue;
            }
            {
                ProfScope p(e, CPQ_K_MIX);
                const int ppc1 = pl.parts_per_callback[1], ppc2 = nTail > 1 ? pl.parts_per_callback[2] : 1;
                const int d1 = (pl.num_parts_ir[1] + ppc1 - 1) / ppc1 - 1;
                const int d2 = nTail > 1 ? (pl.num_parts_ir[2] + ppc2 - 1) / ppc2 - 1 : 0;
                cpq::launch_tail_schedule(e->stream, e->tailState, e->tailSched, n / e->B, e->B, nTail, pl.part_size[1], pl.output_delay[1], d1,
                                          nTail > 1 ? pl.part_size[2] : e->B, nTail > 1 ? pl.output_delay[2] : 0, d2);
            }
            {
                ProfScope p(e, CPQ_K_RFFT_INV);
                cpq::launch_rfft_inv_ols_tail(e->stream, e->Y, dOut, stride, tables(e), e->P, e->nCh, T, nullptr, e->layerOut, e->tailRing,
                                              e->tailRingSlots, e->tailState, e->tailSched, n / e->B, e->B, nTail, pl.gain[1],
                                              nTail > 1 ? pl.gain[2] : 0.0);
            }
            {
                ProfScope p(e, CPQ_K_MIX);
                cpq::launch_tail_append(e->stream, e->tailState, e->layerOut, e->tailRing, e->nCh, (int)stride, e->tailRingSlots, nTail);
            }
        }
        addDirect();
        CPQ_HIP(e, hipGetLastError());
        e->head = (e->head + T) & (e->ringSlots - 1);
        e->histSel ^= 1;
        return CPQ_OK;
    }
    if (e->mainActive) {
        // the channels [c0, c0 + cnt): every buffer of the path is [channel][...]
        auto window = [&](int c0, int cnt) {
            const int64_t ring = (int64_t)c0 * e->ringSlots;
            {
                ProfScope p(e, CPQ_K_RFFT_FWD);
                cpq::launch_rfft_fwd_ols(e->stream, dIn + c0 * stride, stride, e->hist[e->histSel] + (int64_t)c0 * e->P,
                                         e->hist[e->histSel ^ 1] + (int64_t)c0 * e->P, e->X + ring * e->P, e->XDN + ring,
                                         tables(e), e->P, cnt, T, e->head, e->ringSlots);
            }
            double2* y = e->Y + (int64_t)c0 * T * e->P;
            {
                ProfScope p(e, CPQ_K_FDL_MAC);
                cpq::launch_fdl_mac(e->stream, e->macTile, e->X + ring * e->P, e->H, e->irSlot + c0, y, e->P, cnt,
                                    e->kMaxReal, e->ringSlots, e->head, T, (int64_t)e->hRows * e->P, e->irPrivate);
            }
            if (cpq::fdl_mac_needs_dcnyq(e->macTile, T)) {      // the cooperative kernel produces the packed (DC, Nyquist) bin itself
                ProfScope p(e, CPQ_K_DCNYQ);
                cpq::launch_fdl_mac_dcnyq(e->stream, e->XDN + ring, e->HDN, e->irSlot + c0, y, e->P, cnt, e->kMaxReal, e->ringSlots,
                                          e->head, T, e->hRows);
            }
            {
                ProfScope p(e, CPQ_K_RFFT_INV);
                cpq::launch_rfft_inv_ols(e->stream, y, dOut + c0 * stride, stride, tables(e), e->P, cnt, T);
            }
        };
        window(0, e->nCh);
        CPQ_HIP(e, hipGetLastError());
        e->head = (e->head + T) & (e->ringSlots - 1);
        e->histSel ^= 1;
    }
    // Get(): ring output of layer 0, then the direct output, then the tail layers (src/MKLNonUniformConvolver.cpp:1606-1633)
    if (!e->groups.empty()) { const int rc = groupsRunLayer0(e, dOut, n); if (rc != CPQ_OK) return rc; }
    addDirect();
    if (!e->groups.empty()) return groupsRunTails(e, dOut, n);
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
    // Which path runs this IR?  A plan group (engine_native.cpp: every layer at the reference's own partition size, the
    // Add / Get bookkeeping replayed per chunk) takes it when the engine accepts any call quantum (CPQ_CALLS_ANY), when it
    // runs the reference's schedule natively (CPQ_SCHED_REFERENCE_NUC), or when a FilterSpec plan has tail layers -- the
    // HC/LC gains (and the air-absorption damping) multiply every partition spectrum at that LAYER's FFT size (:336-443,
    // :1060-1097).  Everything else runs on the main path: one uniform partition over h_eff.
    std::vector<double> gains;
    cpq_nuc_plan sp{};
    const bool nativeNuc = e->desc.schedule == CPQ_SCHED_REFERENCE_NUC;
    bool native = e->anyCalls || nativeNuc;
    if (spec || native) {
        if (cpq::computeNucPlan(irLen, e->desc.block_size, direct != 0, spec, &sp) != CPQ_OK)
            return fail(e, CPQ_ERR_INVALID_ARG, "layer plan failed");
        if (e->desc.semantics != CPQ_SEM_REFERENCE) return fail(e, CPQ_ERR_INVALID_ARG, "FilterSpec and the native schedule require reference semantics");
        if (spec && sp.num_layers > 1) native = true;
        if (!native && e->P != sp.part_size[0])
            return fail(e, CPQ_ERR_UNSUPPORTED, "FilterSpec needs partition_size == the reference layer-0 partition (%d)", sp.part_size[0]);
        if (native && e->layered) return fail(e, CPQ_ERR_UNSUPPORTED, "engine is in time-varying (layered) mode");
        if (spec && !native) cpq::spectrumFilterGains(*spec, 2 * e->P, gains);
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
    if (e->desc.semantics == CPQ_SEM_REFERENCE && !spec && !native) {
        cpq_nuc_plan probe;
        if (cpq::computeNucPlan(irLen, e->desc.block_size, false, nullptr, &probe) != CPQ_OK)
            return fail(e, CPQ_ERR_INVALID_ARG, "layer plan failed");
        wantLayered = !probe.lti_valid && probe.num_layers > 1;
        if (wantLayered) {
            // each layer is ONE linear convolution of the call with its segment of the IR, whatever the FFT partition;
            // the reader that makes the plan time-varying is replayed per callback of block_size on the layer outputs
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
                    hipMalloc(&e->tailState, 4 * sizeof(long long)) != hipSuccess ||
                    hipMalloc((void**)&e->tailSched, sizeof(long long) * 2 * ((size_t)e->tMax * e->P / e->B)) != hipSuccess)      // per tail layer and callback
                    return fail(e, CPQ_ERR_OOM, "layered-mode buffers could not be allocated");
                CPQ_HIP(e, hipMemset(e->tailRing, 0, sizeof(double) * (size_t)nTail * e->nCh * e->tailRingSlots));
                CPQ_HIP(e, hipMemset(e->tailState, 0, 4 * sizeof(long long)));
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

    auto setDirect = [&](int slot, const double* ir) -> int {         // m_directIRRev (:716-718)
        if (!e->directIr) return CPQ_OK;
        double rev[32] = { 0 };
        for (int i = 0; i < headTaps; ++i) rev[i] = ir[headTaps - 1 - i] * scale;
        CPQ_HIP(e, hipMemcpyAsync(e->directIr + slot * 32, rev, sizeof(rev), hipMemcpyHostToDevice, e->stream));
        CPQ_HIP(e, hipMemcpyAsync(e->directTaps + slot, &headTaps, sizeof(int), hipMemcpyHostToDevice, e->stream));
        CPQ_HIP(e, hipStreamSynchronize(e->stream));                     // rev / headTaps are stack storage
        e->directTapsHost[slot] = headTaps;
        return CPQ_OK;
    };
    const int sFirst = (stream == CPQ_ALL_STREAMS) ? 0 : stream, sEnd = (stream == CPQ_ALL_STREAMS) ? e->desc.n_streams : stream + 1;
    // SetImpulse leaves a convolver that has seen no input (every buffer is allocated anew and zeroed,
    // src/MKLNonUniformConvolver.cpp:697-714, :880-935): the stream's input history goes -- the direct head's last samples, and on
    // the main path its rows of the frequency-domain delay line and the overlap block; the other streams play on
    for (int s = sFirst; s < sEnd; ++s) {
        for (double* p : { e->directHist[0], e->directHist[1] })
            if (p) CPQ_HIP(e, hipMemsetAsync(p + (size_t)2 * s * 32, 0, sizeof(double) * 2 * 32, e->stream));
        if (!native && e->X) {
            CPQ_HIP(e, hipMemsetAsync(e->X + (int64_t)2 * s * e->ringSlots * e->P, 0, (size_t)2 * e->ringSlots * e->P * sizeof(double2), e->stream));
            CPQ_HIP(e, hipMemsetAsync(e->XDN + (int64_t)2 * s * e->ringSlots, 0, (size_t)2 * e->ringSlots * sizeof(double2), e->stream));
            for (double* p : { e->hist[0], e->hist[1] })
                if (p) CPQ_HIP(e, hipMemsetAsync(p + (int64_t)2 * s * e->P, 0, (size_t)2 * e->P * sizeof(double), e->stream));
            if (e->tailRing)            // layered mode: the stream's delay lines (the reader's phase is the engine's and runs on)
                for (int l = 0; l + 1 < e->layerPlan.num_layers; ++l)
                    CPQ_HIP(e, hipMemsetAsync(e->tailRing + ((size_t)l * e->nCh + 2 * s) * e->tailRingSlots, 0,
                                              sizeof(double) * 2 * e->tailRingSlots, e->stream));
        }
    }
    if (native) {
        // the stream(s) leave the main path: their rows there become zero (the main path then contributes silence)
        for (int ch = 0; ch < 2; ++ch) {
            const int slot = slotBase + ch;
            if (e->irParts[slot] > 0) {
                CPQ_HIP(e, hipMemsetAsync(e->H + (int64_t)slot * e->hRows * e->P, 0, (size_t)e->irParts[slot] * e->P * sizeof(double2), e->stream));
                CPQ_HIP(e, hipMemsetAsync(e->HDN + (int64_t)slot * e->hRows, 0, (size_t)e->irParts[slot] * sizeof(double2), e->stream));
                e->irParts[slot] = 0;
            }
            const int rc = setDirect(slot, irs[ch]);
            if (rc != CPQ_OK) return rc;
        }
        const int rc = nativeSetImpulse(e, stream, irL, irR, irLen, scale, headTaps, spec, sp);
        if (rc != CPQ_OK) return rc;
        e->plan = sp;
        e->planValid = true;
        e->directHead = direct != 0;
    } else {
        for (int s = sFirst; s < sEnd; ++s)
            if (e->groupOf[(size_t)s] >= 0) {        // back from a plan group to the main path
                const int rc = leaveNativeGroup(e, s);
                if (rc != CPQ_OK) return rc;
            }
    }
    for (int ch = 0; ch < 2 && !native; ++ch) {
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
                    for (int i = 0; i < headTaps && i < (int)heff.size(); ++i) heff[(size_t)i] = 0.0;
                    rc = setDirect(slot, irs[ch]);
                    if (rc != CPQ_OK) return rc;
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
        if (e->desc.semantics == CPQ_SEM_REFERENCE) {
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
            for (int i = 0; i < headTaps && i < (int)heff.size(); ++i) heff[(size_t)i] = 0.0;     // the head leaves the FFT path
            rc = setDirect(slot, irs[ch]);
            if (rc != CPQ_OK) return rc;
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
    e->mainActive = false;
    for (int s = 0; s < e->desc.n_streams; ++s) e->mainActive = e->mainActive || (e->irLoaded[2 * s] && e->groupOf[(size_t)s] < 0);
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

int32_t cpq_conv_last_got(const cpq_engine* e, int32_t stream)
{
    if (!e || stream < 0 || stream >= e->desc.n_streams) return CPQ_ERR_INVALID_ARG;
    const int gi = e->groupOf[(size_t)stream];
    return gi < 0 ? e->lastCallSamples : e->groups[(size_t)gi]->lastGot;
}

int32_t cpq_conv_reset(cpq_engine* e) { return e ? zeroRuntimeState(e, true, false) : CPQ_ERR_INVALID_ARG; }

int32_t cpq_conv_process_device(cpq_engine* e, const double* dIn, double* dOut, int32_t nSamples)
{
    const int rc = checkCall(e, dIn, dOut, nSamples);
    if (rc != CPQ_OK) return rc;
    CPQ_HIP(e, hipSetDevice(e->device));
    return enqueueConv(e, dIn, dOut, nSamples);
}

int32_t cpq_conv_process(cpq_engine* e, const double* in, double* out, int32_t nSamples)
{
    return viaStaging(e, in, out, nSamples, [e](const double* a, double* b, int n) { return enqueueConv(e, a, b, n); });
}

}  // extern "C"
