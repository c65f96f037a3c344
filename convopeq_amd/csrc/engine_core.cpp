// engine_core.cpp -- C ABI of libconvopeq_mi355x.so (see include/convopeq_mi355x.h): engine life cycle, routing, profiling.
//
// Host side of the engine: the device arena (the reference's per-buffer mkl_malloc manager,
// src/AlignedAllocation.h:22-163 + src/MKLNonUniformConvolver.h:288-365, collapsed into one HBM
// allocation laid out from (streams, partitions, ring slots, blocks per call)), the per-call kernel
// sequence, and the prepare/set_impulse/set_params control surface.  No CPU fallback exists: without a
// HIP device cpq_engine_create fails with CPQ_ERR_NO_DEVICE and nothing else can be called.
#include "engine_internal.hpp"

namespace {

std::string g_createError;
std::mutex g_createErrorMutex;

}  // namespace

namespace cpqi {

int fail(cpq_engine* e, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (e) e->lastError = buf;
    else { std::lock_guard<std::mutex> lk(g_createErrorMutex); g_createError = buf; }
    return code;
}


int nextPow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
int64_t alignUp(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

cpq::FftTables tables(const cpq_engine* e) { return cpq::FftTables{ e->tw512, e->tw1024 }; }


int checkCall(cpq_engine* e, const void* in, const void* out, int nSamples)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    if (!in || !out) return fail(e, CPQ_ERR_INVALID_ARG, "null buffer");
    if (e->anyCalls) {
        if (nSamples <= 0 || nSamples > e->maxCall)
            return fail(e, CPQ_ERR_INVALID_ARG, "n_samples=%d outside 1..%d (max_blocks_per_call * block_size)", nSamples, e->maxCall);
    } else {
        if (nSamples <= 0 || nSamples % e->P != 0)
            return fail(e, CPQ_ERR_INVALID_ARG, "n_samples=%d is not a positive multiple of the partition size %d "
                        "(create the engine with CPQ_CALLS_ANY for other call sizes)", nSamples, e->P);
        if (nSamples / e->P > e->tMax)
            return fail(e, CPQ_ERR_INVALID_ARG, "n_samples=%d exceeds max_blocks_per_call=%d blocks of %d", nSamples,
                        e->desc.max_blocks_per_call, e->B);
    }
    if ((reinterpret_cast<uintptr_t>(in) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u))
        return fail(e, CPQ_ERR_INVALID_ARG, "buffers must be 16-byte aligned");
    return CPQ_OK;
}


// Call-sized buffers that only some entry points need are allocated when first used: the staging pair of the host-pointer
// entry points, the hand-off buffer of the EQ -> convolver order (3 x 2 GB at 256 streams and 524288-sample calls).
int ensureCallBuffer(cpq_engine* e, double** buf, const char* what)
{
    if (*buf) return CPQ_OK;
    const size_t bytes = (size_t)e->nCh * e->tMax * e->P * sizeof(double);
    if (hipMalloc((void**)buf, bytes) != hipSuccess) {
        (void)hipGetLastError();
        *buf = nullptr;
        return fail(e, CPQ_ERR_OOM, "%s buffer of %zu bytes could not be allocated", what, bytes);
    }
    return CPQ_OK;
}

int zeroRuntimeState(cpq_engine* e, bool conv, bool eq)
{
    CPQ_HIP(e, hipSetDevice(e->device));
    if (conv) {
        CPQ_HIP(e, hipMemsetAsync(e->X, 0, (size_t)e->nCh * e->ringSlots * e->P * sizeof(double2), e->stream));
        CPQ_HIP(e, hipMemsetAsync(e->XDN, 0, (size_t)e->nCh * e->ringSlots * sizeof(double2), e->stream));
        CPQ_HIP(e, hipMemsetAsync(e->hist[0], 0, (size_t)e->nCh * e->P * sizeof(double), e->stream));
        CPQ_HIP(e, hipMemsetAsync(e->hist[1], 0, (size_t)e->nCh * e->P * sizeof(double), e->stream));
        e->head = 0;
        e->histSel = 0;
        { const int rc = resetGroups(e); if (rc != CPQ_OK) return rc; }
        for (double* p : { e->directHist[0], e->directHist[1] })
            if (p) CPQ_HIP(e, hipMemsetAsync(p, 0, sizeof(double) * 32 * e->nCh, e->stream));
        if (e->tailState) CPQ_HIP(e, hipMemsetAsync(e->tailState, 0, 4 * sizeof(long long), e->stream));
        if (e->tailRing) CPQ_HIP(e, hipMemsetAsync(e->tailRing, 0, sizeof(double) * (size_t)(e->layerPlan.num_layers - 1) * e->nCh * e->tailRingSlots, e->stream));
        if (e->dryRing) CPQ_HIP(e, hipMemsetAsync(e->dryRing, 0, (size_t)e->nCh * e->dryRingSize * sizeof(double), e->stream));
        e->dryPos = 0;
        for (auto& f : e->latFade) f = cpq_engine::LatencyFade{};
    }
    if (eq) {
        CPQ_HIP(e, hipMemsetAsync(e->svfState, 0, (size_t)e->nCh * kBands * 2 * sizeof(double), e->stream));
        CPQ_HIP(e, hipMemsetAsync(e->ofState, 0, (size_t)e->nCh * kBands * 2 * sizeof(double), e->stream));
        if (e->agcState) CPQ_HIP(e, hipMemsetAsync(e->agcState, 0, (size_t)e->desc.n_streams * 3 * sizeof(double), e->stream));
    }
    CPQ_HIP(e, hipStreamSynchronize(e->stream));
    return CPQ_OK;
}


}  // namespace cpqi

using namespace cpqi;

extern "C" {

int32_t cpq_abi_version(void) { return CPQ_ABI_VERSION; }

const char* cpq_status_string(int32_t s)
{
    switch (s) {
        case CPQ_OK: return "ok";
        case CPQ_ERR_INVALID_ARG: return "invalid argument";
        case CPQ_ERR_NO_DEVICE: return "no usable HIP device";
        case CPQ_ERR_OOM: return "out of device memory";
        case CPQ_ERR_DEVICE: return "HIP runtime error";
        case CPQ_ERR_UNSUPPORTED: return "not supported by this engine version";
        case CPQ_ERR_NOT_READY: return "engine not ready";
        default: return "unknown status";
    }
}

const char* cpq_last_error(const cpq_engine* e)
{
    if (e) return e->lastError.c_str();
    std::lock_guard<std::mutex> lk(g_createErrorMutex);
    static thread_local std::string copy;
    copy = g_createError;
    return copy.c_str();
}

const char* cpq_kernel_name(int32_t id)
{
    switch (id) {
        case CPQ_K_RFFT_FWD: return "k_rfft_fwd_ols";
        case CPQ_K_FDL_MAC: return "k_fdl_mac";
        case CPQ_K_DCNYQ: return "k_fdl_mac_dcnyq";
        case CPQ_K_RFFT_INV: return "k_rfft_inv_ols";
        case CPQ_K_SVF: return "k_svf_cascade";
        case CPQ_K_SVF_TP: return "k_svf_cascade_tp";
        case CPQ_K_MIX: return "k_convproc_mix";
        case CPQ_K_OUTFILT: return "k_outfilter_cascade";
        default: return "?";
    }
}

// ------------------------------------------------------------------ diagnostics
// The partition FFT kernels in isolation (tests/test_gpu_fft.py): forward of every overlap-save frame from a silent history,
// inverse of the same spectra.  Own device buffers and twiddles, the null stream; no engine.
int32_t cpq_diag_eq_chain_status(cpq_engine* e, uint32_t* launches, uint32_t* gaveUp)
{
    if (!e || !launches || !gaveUp) return CPQ_ERR_INVALID_ARG;
    *launches = 0;
    *gaveUp = 0;
    if (!e->svfChain || e->svfChainSpans <= 0) return CPQ_OK;
    (void)hipSetDevice(e->device);
    uint32_t hdr[4] = { 0, 0, 0, 0 };          // generation, finished workgroups, ticket, error (svf_kernels.hip: TpvChainHeader)
    CPQ_HIP(e, hipStreamSynchronize(e->stream));
    CPQ_HIP(e, hipMemcpy(hdr, e->svfChain, sizeof(hdr), hipMemcpyDeviceToHost));
    *launches = hdr[0];
    *gaveUp = hdr[3];
    return CPQ_OK;
}

int32_t cpq_diag_partition_fft(int32_t P, int32_t nCh, int32_t T, const double* in, double* spectra, double* out)
{
    if (P < 64 || P > 131072 || (P & (P - 1)) || nCh <= 0 || T <= 0 || !in || !spectra || !out) return CPQ_ERR_INVALID_ARG;
    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0) { (void)hipGetLastError(); return CPQ_ERR_NO_DEVICE; }
    int ringSlots = 1;
    while (ringSlots < T) ringSlots <<= 1;
    const size_t nTime = (size_t)nCh * T * P, nSpec = (size_t)nCh * ringSlots * P;
    double *dIn = nullptr, *dOut = nullptr, *dHist = nullptr;
    double2 *dX = nullptr, *dXdn = nullptr, *dTw = nullptr, *dTw2 = nullptr, *dScratch = nullptr, *dY = nullptr;
    std::vector<double2> w1((size_t)P), w2((size_t)P);
    const long double twoPi = 6.283185307179586476925286766559005768L;
    for (int m = 0; m < P; ++m) {
        const long double a = -twoPi * m / (long double)P, b = -twoPi * m / (long double)(2 * P);
        w1[(size_t)m] = make_double2((double)cosl(a), (double)sinl(a));
        w2[(size_t)m] = make_double2((double)cosl(b), (double)sinl(b));
    }
    int32_t rc = CPQ_OK;
    auto ok = [&](hipError_t err) { if (err != hipSuccess && rc == CPQ_OK) { (void)hipGetLastError(); rc = CPQ_ERR_DEVICE; } return err == hipSuccess; };
    ok(hipMalloc((void**)&dIn, nTime * sizeof(double))) && ok(hipMalloc((void**)&dOut, nTime * sizeof(double))) &&
        ok(hipMalloc((void**)&dHist, (size_t)2 * nCh * P * sizeof(double))) && ok(hipMalloc((void**)&dX, nSpec * sizeof(double2))) &&
        ok(hipMalloc((void**)&dXdn, (size_t)nCh * ringSlots * sizeof(double2))) && ok(hipMalloc((void**)&dTw, (size_t)2 * P * sizeof(double2))) &&
        ok(hipMalloc((void**)&dTw2, (size_t)2 * P * sizeof(double2))) &&       // (second halves: the reordered tables of the four-step transforms)
        ok(hipMalloc((void**)&dScratch, (P > 4096 ? (size_t)nCh * T * P : 1) * sizeof(double2))) &&
        ok(hipMalloc((void**)&dY, (size_t)nCh * T * P * sizeof(double2)));
    if (rc == CPQ_OK) {
        ok(hipMemset(dHist, 0, (size_t)2 * nCh * P * sizeof(double)));
        ok(hipMemset(dX, 0, nSpec * sizeof(double2)));
        ok(hipMemcpy(dIn, in, nTime * sizeof(double), hipMemcpyHostToDevice));
        ok(hipMemcpy(dTw, w1.data(), (size_t)P * sizeof(double2), hipMemcpyHostToDevice));
        ok(hipMemcpy(dTw2, w2.data(), (size_t)P * sizeof(double2), hipMemcpyHostToDevice));
        if (P > 4096) {
            std::vector<double2> wc((size_t)P), ws((size_t)P);
            cpq::fill_big_twiddles(w1.data(), w2.data(), P, wc.data(), ws.data());
            ok(hipMemcpy(dTw + P, wc.data(), (size_t)P * sizeof(double2), hipMemcpyHostToDevice));
            ok(hipMemcpy(dTw2 + P, ws.data(), (size_t)P * sizeof(double2), hipMemcpyHostToDevice));
        }
    }
    if (rc == CPQ_OK) {
        const cpq::FftTables tw{ dTw, dTw2, P > 4096 ? dTw + P : nullptr, P > 4096 ? dTw2 + P : nullptr };
        cpq::launch_rfft_fwd_ols(nullptr, dIn, (int64_t)T * P, dHist, dHist + (size_t)nCh * P, dX, dXdn, tw, P, nCh, T, 0, ringSlots, dScratch);
        // the ring holds block t of channel c at [c][t] of ringSlots slots: [c][t] of T slots for the inverse and the caller
        for (int c = 0; c < nCh && rc == CPQ_OK; ++c)
            ok(hipMemcpyAsync(dY + (size_t)c * T * P, dX + (size_t)c * ringSlots * P, (size_t)T * P * sizeof(double2), hipMemcpyDeviceToDevice, nullptr));
        cpq::launch_rfft_inv_ols(nullptr, dY, dOut, (int64_t)T * P, tw, P, nCh, T, dScratch);
        ok(hipGetLastError());
        ok(hipDeviceSynchronize());
    }
    if (rc == CPQ_OK) {
        ok(hipMemcpy(spectra, dY, (size_t)nCh * T * P * sizeof(double2), hipMemcpyDeviceToHost));
        ok(hipMemcpy(out, dOut, nTime * sizeof(double), hipMemcpyDeviceToHost));
    }
    for (void* p : { (void*)dIn, (void*)dOut, (void*)dHist, (void*)dX, (void*)dXdn, (void*)dTw, (void*)dTw2, (void*)dScratch, (void*)dY })
        if (p) (void)hipFree(p);
    return rc;
}

// ------------------------------------------------------------------ host-only helpers
int32_t cpq_nuc_plan_compute(int32_t irLen, int32_t blockSize, int32_t direct, const cpq_filter_spec* spec,
                             cpq_nuc_plan* plan)
{
    return cpq::computeNucPlan(irLen, blockSize, direct != 0, spec, plan);
}

int32_t cpq_nuc_heff(const double* ir, int32_t irLen, int32_t blockSize, double scale, const cpq_filter_spec* spec,
                     double* heff, int32_t cap)
{
    if (!ir) return CPQ_ERR_INVALID_ARG;
    std::vector<double> h;
    cpq_nuc_plan p;
    const int rc = cpq::buildHeff(ir, irLen, blockSize, scale, spec, h, &p);
    if (rc != CPQ_OK) return rc;
    if (heff && cap > 0) std::memcpy(heff, h.data(), sizeof(double) * (size_t)std::min<int>(cap, (int)h.size()));
    return (int32_t)h.size();
}

int32_t cpq_eq_design_svf(int32_t type, float freq, float gainDb, float q, double sr, cpq_svf_coeffs* out)
{
    if (!out) return CPQ_ERR_INVALID_ARG;
    cpq::designSvf(type, freq, gainDb, q, sr, out);
    return CPQ_OK;
}

void cpq_eq_params_default(cpq_eq_params* p) { if (p) cpq::defaultEqParams(p); }

// ------------------------------------------------------------------------------ engine
int32_t cpq_engine_create(const cpq_engine_desc* d, cpq_engine** out)
{
    if (!d || !out) return fail(nullptr, CPQ_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (d->struct_size != (int32_t)sizeof(cpq_engine_desc))
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "struct_size %d != %zu", d->struct_size, sizeof(cpq_engine_desc));
    if (d->n_streams <= 0 || d->max_ir_len <= 0 || d->max_blocks_per_call <= 0)
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "n_streams, max_ir_len and max_blocks_per_call must be positive");
    if (d->call_mode != CPQ_CALLS_WHOLE_BLOCKS && d->call_mode != CPQ_CALLS_ANY)
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "call_mode must be CPQ_CALLS_WHOLE_BLOCKS or CPQ_CALLS_ANY");
    const bool anyCalls = d->call_mode == CPQ_CALLS_ANY;
    if (anyCalls) {
        if (d->block_size < 1 || d->block_size > 4096)
            return fail(nullptr, CPQ_ERR_INVALID_ARG, "block_size (the call quantum) must be in [1, 4096]");
        if (d->semantics != CPQ_SEM_REFERENCE || (d->partition_size != 0 && d->partition_size != CPQ_PARTITION_AUTO))
            return fail(nullptr, CPQ_ERR_INVALID_ARG, "CPQ_CALLS_ANY runs the reference's own layer plan: reference semantics, partition_size 0");
    } else if (d->block_size < 64 || d->block_size > 4096 || (d->block_size & (d->block_size - 1)))
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "block_size must be a power of two in [64, 4096] (any quantum from 1 to 4096: call_mode = CPQ_CALLS_ANY)");
    if (d->semantics != CPQ_SEM_REFERENCE && d->semantics != CPQ_SEM_EXACT)
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "bad semantics");
    if (d->mac_tile != 0 && d->mac_tile != 4 && d->mac_tile != 8 && d->mac_tile != 16 && d->mac_tile != 32)
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "mac_tile must be 0, 4, 8, 16 or 32");
    if (d->schedule != CPQ_SCHED_UNIFORM && d->schedule != CPQ_SCHED_REFERENCE_NUC)
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "schedule must be CPQ_SCHED_UNIFORM or CPQ_SCHED_REFERENCE_NUC");
    if (d->schedule == CPQ_SCHED_REFERENCE_NUC &&
        (d->semantics != CPQ_SEM_REFERENCE ||
         (d->partition_size != 0 && d->partition_size != CPQ_PARTITION_AUTO && d->partition_size != d->block_size)))
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "the non-uniform schedule needs reference semantics and partition_size == block_size");

    int nDev = 0;
    if (hipGetDeviceCount(&nDev) != hipSuccess || nDev <= 0)
        return fail(nullptr, CPQ_ERR_NO_DEVICE, "no HIP device visible: the gfx950 kernels cannot run (no CPU fallback)");
    if (d->device < 0 || d->device >= nDev)
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "device %d out of range (%d visible)", d->device, nDev);
    if (hipSetDevice(d->device) != hipSuccess) return fail(nullptr, CPQ_ERR_NO_DEVICE, "hipSetDevice failed");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d->device) != hipSuccess)
        return fail(nullptr, CPQ_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, CPQ_ERR_NO_DEVICE, "device %d is %s; this library ships gfx950 code only", d->device,
                    prop.gcnArchName);

    cpq_engine* e = new (std::nothrow) cpq_engine();
    if (!e) return fail(nullptr, CPQ_ERR_OOM, "host allocation failed");
    e->desc = *d;
    e->device = d->device;
    e->sampleRate = d->sample_rate > 0.0 ? d->sample_rate : 48000.0;
    e->nCh = 2 * d->n_streams;
    e->B = d->block_size;
    e->anyCalls = anyCalls;
    e->P0 = nextPow2(std::max(d->block_size, 64));
    e->maxCall = (int)std::min<int64_t>((int64_t)d->max_blocks_per_call * d->block_size, (int64_t)1 << 30);
    int32_t partition = d->partition_size;
    if (partition == CPQ_PARTITION_AUTO) {
        // the larger partition wins wherever the calls allow it (profiles/r02b_sweep_partition_x_blocks_per_call.txt:
        // K shrinks by P / B, the FFT cost per sample stays): 4096 for calls of at least eight such partitions (below
        // that the MAC streams the same IR + FDL bytes per call at either size and the 512-point kernels do it faster:
        // profiles/r02e_small_calls.txt), else 512, else the block itself.  The reference's own
        // schedule and ragged calls keep the reference's layer-0 partition.  (Blocks of 1024 and more, whose reference plan
        // is time-varying, run one convolution per layer: those take the larger partition as well.)
        partition = 0;
        if (!anyCalls && d->schedule == CPQ_SCHED_UNIFORM)
            for (int32_t cand : { 4096, 512 })
                if (cand > d->block_size && e->maxCall % cand == 0 && (cand == 512 || e->maxCall >= 8 * cand)) { partition = cand; break; }
    }
    e->desc.partition_size = partition;
    e->P = partition ? partition : (anyCalls ? e->P0 : d->block_size);
    if (!anyCalls && (e->P < e->B || e->P > 4096 || (e->P & (e->P - 1)) || ((int64_t)d->max_blocks_per_call * e->B) % e->P != 0)) {
        const int p = e->P;
        delete e;
        return fail(nullptr, CPQ_ERR_INVALID_ARG,
                    "partition_size %d must be a power of two in [block_size, 4096] dividing block_size*max_blocks_per_call", p);
    }
    e->tMax = (int)(((int64_t)e->maxCall + e->P - 1) / e->P);     // partitions per call (rounded up for ragged calls)
    e->macTile = d->mac_tile;     // 0 = automatic (workgroup-cooperative kernel for calls of >= 32 blocks)

    // partition capacity from the longest h_eff the plan can produce for max_ir_len
    cpq_nuc_plan pl;
    if (cpq::computeNucPlan(d->max_ir_len, d->block_size, false, nullptr, &pl) != CPQ_OK) {
        delete e;
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "cannot plan max_ir_len=%d", d->max_ir_len);
    }
    const int taps = (d->semantics == CPQ_SEM_REFERENCE) ? std::max(pl.heff_len, d->max_ir_len) : d->max_ir_len;
    // engines whose every stream runs in a plan group (engine_native.cpp) keep only a token main-path arena
    const bool allNative = anyCalls || d->schedule == CPQ_SCHED_REFERENCE_NUC;
    const int kReal = allNative ? 1 : (taps + e->P - 1) / e->P;
    e->kCap = (int)alignUp(kReal, cpq::kMacMaxTile);
    e->hRows = e->kCap + 4 * cpq::kMacMaxTile;   // zero rows read by the prefetch past the last partition (per layer in layered mode)
    e->ringSlots = nextPow2(e->kCap + cpq::kMacMaxTile + e->tMax);
    e->heffCap = std::max<int64_t>((int64_t)e->kCap * e->P, d->max_ir_len);

    // ---- arena layout
    struct Item { void** ptr; int64_t bytes; };
    const int64_t nCh = e->nCh;
    {
        int nCu = 0;
        if (hipDeviceGetAttribute(&nCu, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess) { (void)hipGetLastError(); nCu = 0; }
        e->svfChainGrid = 2 * nCu;          // the span kernel keeps 76 KB of LDS: two workgroups per CU
    }
    // One workgroup per channel runs whole rounds of svfChainGrid workgroups: a last round of r workgroups costs 0.70 of a full
    // one up to half the slots (the workgroups are alone on their CUs) and a whole one beyond.  Chained spans deal (span,
    // channel) tasks to the slots whatever the count, at 6 % more per task (hand-over polls, tables per task).  Measured
    // (profiles/r04x_eq_stream_counts.txt): 300 streams 7.63 -> chained; 255 / 256 / 512 / 1024 streams stay as they are.
    bool chained = false;
    if (e->svfChainGrid > 0 && e->maxCall >= 2 * 8192) {
        const int64_t full = nCh / e->svfChainGrid, r = nCh % e->svfChainGrid;
        const double perChannel = (double)full + (r == 0 ? 0.0 : (2 * r <= e->svfChainGrid ? 0.70 : 1.0));
        chained = 1.06 * (double)nCh / (double)e->svfChainGrid < perChannel;
    }
    Item items[] = {
        { (void**)&e->X, nCh * e->ringSlots * e->P * (int64_t)sizeof(double2) },
        { (void**)&e->XDN, nCh * e->ringSlots * (int64_t)sizeof(double2) },
        { (void**)&e->H, nCh * e->hRows * e->P * (int64_t)sizeof(double2) },
        { (void**)&e->HDN, nCh * e->hRows * (int64_t)sizeof(double2) },
        { (void**)&e->Y, nCh * e->tMax * e->P * (int64_t)sizeof(double2) },
        { (void**)&e->hist[0], nCh * e->P * (int64_t)sizeof(double) },
        { (void**)&e->hist[1], nCh * e->P * (int64_t)sizeof(double) },
        { (void**)&e->heffDev, e->heffCap * (int64_t)sizeof(double) },
        { (void**)&e->gainDev, (e->P + 1) * (int64_t)sizeof(double) },
        { (void**)&e->tw512, e->P * (int64_t)sizeof(double2) },
        { (void**)&e->tw1024, e->P * (int64_t)sizeof(double2) },
        { (void**)&e->irSlot, nCh * (int64_t)sizeof(int) },
        { (void**)&e->svfCoef, nCh * kBands * 6 * (int64_t)sizeof(double) },
        { (void**)&e->svfFlags, nCh * kBands * (int64_t)sizeof(int) },
        { (void**)&e->svfSatGain, nCh * 2 * (int64_t)sizeof(double) },
        { (void**)&e->svfState, nCh * kBands * 2 * (int64_t)sizeof(double) },
        { (void**)&e->svfTp, (nCh / 2) * kBands * cpq::kSvfTpTableDoubles * (int64_t)sizeof(double) },
        { (void**)&e->ofCoef, nCh * kBands * 6 * (int64_t)sizeof(double) },
        { (void**)&e->ofFlags, nCh * kBands * (int64_t)sizeof(int) },
        { (void**)&e->ofSatGain, nCh * 2 * (int64_t)sizeof(double) },
        { (void**)&e->ofState, nCh * kBands * 2 * (int64_t)sizeof(double) },
        { (void**)&e->ofTp, (nCh / 2) * kBands * cpq::kSvfTpTableDoubles * (int64_t)sizeof(double) },
        // scheduling words of the time-parallel cascade (svf_kernels.hip): header, the arrival counters of the CUs, and -- only
        // for engines that chain their spans (the rule above) -- the band states handed from span to span
        { (void**)&e->svfChain, (int64_t)cpq::svf_chain_bytes((int)nCh, chained ? e->maxCall : 0) },
    };
    e->svfChainSpans = chained ? cpq::svf_chain_spans(e->maxCall) : 0;
    int64_t total = 0;
    for (const Item& it : items) total += alignUp(it.bytes, 256);
    if (hipMalloc((void**)&e->arena, (size_t)total) != hipSuccess) {
        (void)hipGetLastError();
        delete e;
        return fail(nullptr, CPQ_ERR_OOM, "device arena of %lld bytes could not be allocated", (long long)total);
    }
    e->arenaBytes = total;
    int64_t off = 0;
    for (const Item& it : items) { *it.ptr = e->arena + off; off += alignUp(it.bytes, 256); }

    // everything starts zero: FDL, history, IR spectra (incl. padding rows), SVF state
    if (hipMemset(e->arena, 0, (size_t)total) != hipSuccess) {
        cpq_engine_destroy(e);
        return fail(nullptr, CPQ_ERR_DEVICE, "hipMemset of the arena failed");
    }
    // twiddles in extended precision on the host, rounded once (SURVEY.md section 7 "hard parts")
    std::vector<double2> w512(e->P), w1024(e->P);
    const long double twoPi = 6.283185307179586476925286766559005768L;
    for (int m = 0; m < e->P; ++m) {
        const long double a = -twoPi * m / (long double)e->P, b = -twoPi * m / (long double)(2 * e->P);
        w512[m] = make_double2((double)cosl(a), (double)sinl(a));
        w1024[m] = make_double2((double)cosl(b), (double)sinl(b));
    }
    if (hipMemcpy(e->tw512, w512.data(), e->P * sizeof(double2), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(e->tw1024, w1024.data(), e->P * sizeof(double2), hipMemcpyHostToDevice) != hipSuccess) {
        cpq_engine_destroy(e);
        return fail(nullptr, CPQ_ERR_DEVICE, "twiddle upload failed");
    }
    e->irSlotHost.assign(e->nCh, 0);
    for (int c = 0; c < e->nCh; ++c) e->irSlotHost[c] = c;
    e->irLoaded.assign(e->nCh, 0);
    e->irParts.assign(e->nCh, 0);
    e->groupOf.assign(d->n_streams, -1);
    e->eqTpSafe.assign(d->n_streams, 1);   // no active band yet: trivially guard-free
    e->eqMidSide.assign(d->n_streams, 0);
    e->eqParamsHost.assign(d->n_streams, cpq_eq_params{});
    e->eqParamsSet.assign(d->n_streams, 0);
    e->eqBypass.assign(d->n_streams, cpq_engine::EqBypass{});
    e->eqResetPending.assign(d->n_streams, 0u);
    e->agcResetPending.assign(d->n_streams, 0);
    e->latFade.assign(d->n_streams, cpq_engine::LatencyFade{});
    e->trimHost.assign(d->n_streams, 1.0);
    e->makeupHost.assign(d->n_streams, 1.0);
    e->ofPass.assign(d->n_streams, 0);
    e->ofModesHost.assign(d->n_streams, cpq_engine::OfModes{ 0, 1, 0, 1 });
    e->ofModesSet.assign(d->n_streams, 0);
    e->procParams.assign(d->n_streams, cpq_convproc_params{ 1.0f, 0, 0, 0.0f });
    e->procBypass.assign(d->n_streams, 0);
    e->procDryOnly.assign(d->n_streams, 0);
    e->mixRamp.assign(d->n_streams, cpq_engine::MixRamp{});
    e->agcOnHost.assign(d->n_streams, 0);
    e->gainRamp.assign(d->n_streams, cpq_engine::GainRamp{});
    if (hipMemcpy(e->irSlot, e->irSlotHost.data(), sizeof(int) * e->nCh, hipMemcpyHostToDevice) != hipSuccess) {
        cpq_engine_destroy(e);
        return fail(nullptr, CPQ_ERR_DEVICE, "irSlot upload failed");
    }
    *out = e;
    return CPQ_OK;
}

void cpq_engine_destroy(cpq_engine* e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    for (auto& s : e->prof) {
        for (auto& ev : s.pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
        for (auto& ev : s.freeList) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    }
    freeGroups(e);
    freePinnedRing(e);
    if (e->copyIn) {
        (void)hipStreamDestroy(e->copyIn);
        (void)hipStreamDestroy(e->copyOut);
        for (int i = 0; i < 4; ++i) { (void)hipEventDestroy(e->evIn[i]); (void)hipEventDestroy(e->evDone[i]); }
    }
    if (e->arena) (void)hipFree(e->arena);
    for (double* p : { e->stageIn, e->stageOut, e->mid, e->dryRing, e->latGains, e->layerOut, e->tailRing, e->agcState, e->agcRmsIn, e->agcRmsOut, e->agcGains }) if (p) (void)hipFree(p);
    if (e->agcOn) (void)hipFree(e->agcOn);
    if (e->rampOn) (void)hipFree(e->rampOn);
    if (e->rampGains) (void)hipFree(e->rampGains);
    if (e->tailState) (void)hipFree(e->tailState);
    if (e->tailSched) (void)hipFree(e->tailSched);
    if (e->procGains) (void)hipFree(e->procGains);
    if (e->procDelay) (void)hipFree(e->procDelay);
    if (e->procWetOn) (void)hipFree(e->procWetOn);
    for (int* p : { e->latNew, e->latOld, e->latLen }) if (p) (void)hipFree(p);
    if (e->eqDry) (void)hipFree(e->eqDry);
    if (e->silentDev) (void)hipFree(e->silentDev);
    if (e->silentHost) (void)hipHostFree(e->silentHost);
    if (e->trimDev) (void)hipFree(e->trimDev);
    if (e->makeupDev) (void)hipFree(e->makeupDev);
    if (e->blendOn) (void)hipFree(e->blendOn);
    if (e->blendLen) (void)hipFree(e->blendLen);
    if (e->blendEnd) (void)hipFree(e->blendEnd);
    if (e->blendGains) (void)hipFree(e->blendGains);
    if (e->mixRampLen) (void)hipFree(e->mixRampLen);
    if (e->mixRampGains) (void)hipFree(e->mixRampGains);
    for (double* p : { e->directIr, e->directHist[0], e->directHist[1], e->directOut }) if (p) (void)hipFree(p);
    if (e->directTaps) (void)hipFree(e->directTaps);
    delete e;
}

int32_t cpq_engine_set_stream(cpq_engine* e, void* s)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    e->stream = reinterpret_cast<hipStream_t>(s);
    return CPQ_OK;
}

int32_t cpq_engine_synchronize(cpq_engine* e)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    CPQ_HIP(e, hipSetDevice(e->device));
    CPQ_HIP(e, hipStreamSynchronize(e->stream));
    return CPQ_OK;
}

int64_t cpq_engine_arena_bytes(const cpq_engine* e) { return e ? e->arenaBytes : 0; }
int32_t cpq_engine_partition_size(const cpq_engine* e) { return e ? e->P : 0; }

int32_t cpq_engine_prepare(cpq_engine* e, double sampleRate, int32_t maxBlock)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    if (sampleRate <= 0.0) return fail(e, CPQ_ERR_INVALID_ARG, "sample rate must be positive");
    if (maxBlock <= 0 || maxBlock > e->maxCall)
        return fail(e, CPQ_ERR_INVALID_ARG, "max_block %d exceeds block_size*max_blocks_per_call", maxBlock);
    const bool rateChanged = sampleRate != e->sampleRate;
    e->sampleRate = sampleRate;
    if (rateChanged) {
        // coefficients follow the rate: re-design what was set (one call when every stream shares the parameters)
        const int S = e->desc.n_streams;
        bool same = S > 0 && e->eqParamsSet[0];
        for (int s = 1; s < S && same; ++s)
            same = e->eqParamsSet[s] && std::memcmp(&e->eqParamsHost[s], &e->eqParamsHost[0], sizeof(cpq_eq_params)) == 0;
        if (same) {
            const cpq_eq_params p = e->eqParamsHost[0];
            const int rc = cpq_eq_set_params(e, CPQ_ALL_STREAMS, &p);
            if (rc != CPQ_OK) return rc;
        } else {
            for (int s = 0; s < S; ++s)
                if (e->eqParamsSet[s]) {
                    const cpq_eq_params p = e->eqParamsHost[s];
                    const int rc = cpq_eq_set_params(e, s, &p);
                    if (rc != CPQ_OK) return rc;
                }
        }
        for (int s = 0; s < S; ++s)
            if (e->ofModesSet[s]) {
                const auto m = e->ofModesHost[s];
                const int rc = cpq_outfilter_set_params(e, s, m.convIsLast, m.hc, m.lc, m.lp);
                if (rc != CPQ_OK) return rc;
            }
    }
    e->eqProcessed = false;
    e->procProcessed = false;
    syncEqBypass(e);
    for (size_t s = 0; s < e->mixRamp.size(); ++s) {      // mixSmoother.setCurrentAndTargetValue(mix) (Lifecycle.cpp:370-371)
        auto& r = e->mixRamp[s];
        r.current = r.target = (double)e->procParams[s].mix;
        r.step = 0.0;
        r.remaining = 0;
    }
    for (auto& r : e->gainRamp) { r.current = r.target = r.wanted; r.step = 0.0; r.remaining = 0; }   // setCurrentAndTargetValue (Core.cpp:765)
    return zeroRuntimeState(e, true, true);
}

int32_t cpq_host_register(void* ptr, size_t bytes)
{
    if (!ptr || bytes == 0) return CPQ_ERR_INVALID_ARG;
    return hipHostRegister(ptr, bytes, hipHostRegisterDefault) == hipSuccess ? CPQ_OK : CPQ_ERR_DEVICE;
}

int32_t cpq_host_unregister(void* ptr)
{
    if (!ptr) return CPQ_ERR_INVALID_ARG;
    return hipHostUnregister(ptr) == hipSuccess ? CPQ_OK : CPQ_ERR_DEVICE;
}

int32_t cpq_engine_set_order(cpq_engine* e, int32_t order)
{
    if (!e || (order != CPQ_ORDER_CONV_THEN_EQ && order != CPQ_ORDER_EQ_THEN_CONV)) return CPQ_ERR_INVALID_ARG;
    e->order = order;
    return CPQ_OK;
}


// ------------------------------------------------------------------------ whole path
static int enqueueBoth(cpq_engine* e, const double* a, double* b, int n)
{
    int rc = CPQ_OK;
    auto conv = [e](const double* x, double* y, int t) {
        return e->convLevel == CPQ_LEVEL_PROCESSOR ? enqueueConvProc(e, x, y, t) : enqueueConv(e, x, y, t);
    };
    if (e->order == CPQ_ORDER_CONV_THEN_EQ) {
        if (!e->convBypassed) rc = conv(a, b, n);
        else if (a != b) cpq::launch_rows_copy(e->stream, a, n, 0, b, n, 0, n, e->nCh);
        if (rc == CPQ_OK) rc = enqueueEq(e, b, b, n);
    } else if (e->convBypassed) {
        rc = enqueueEq(e, a, b, n);
    } else {
        rc = ensureCallBuffer(e, &e->mid, "EQ -> convolver hand-off");
        if (rc == CPQ_OK) rc = enqueueEq(e, a, e->mid, n);
        if (rc == CPQ_OK && e->anyTrim) {       // scaleBlockFallback(block, convolverInputTrimGain) (:440-447)
            ProfScope p(e, CPQ_K_MIX);
            cpq::launch_rows_scale(e->stream, e->mid, n, n, e->nCh, e->trimDev);
        }
        if (rc == CPQ_OK) rc = conv(e->mid, b, n);
    }
    if (rc == CPQ_OK && e->ofInPath) {
        // outputFilter.process runs when the convolver or the EQ is active (:453-463); a stream with both bypassed
        // keeps its filter state untouched
        bool anyActive = false;
        for (int s = 0; s < e->desc.n_streams; ++s) {
            const char pass = (e->convBypassed && e->eqBypass[s].requested) ? 1 : 0;
            anyActive = anyActive || !pass;
            if (pass == e->ofPass[s] || !e->ofModesSet[s]) continue;
            int flags[2 * kBands] = {};
            if (!pass) for (int ch = 0; ch < 2; ++ch) for (int k = 0; k < 3; ++k) flags[ch * kBands + k] = 1 | 4;
            { const int rcUp = stageUpload(e, e->ofFlags + (size_t)2 * s * kBands, flags, sizeof(flags)); if (rcUp != CPQ_OK) return rcUp; }
            e->ofPass[s] = pass;
        }
        if (anyActive) rc = enqueueOutFilter(e, b, b, n);
    }
    if (rc == CPQ_OK && e->anyMakeup) {         // scaleBlockFallback(block, outputMakeupGain) (:465-469)
        ProfScope p(e, CPQ_K_MIX);
        cpq::launch_rows_scale(e->stream, b, n, n, e->nCh, e->makeupDev);
        CPQ_HIP(e, hipGetLastError());
    }
    return rc;
}

int32_t cpq_engine_set_gains(cpq_engine* e, int32_t stream, double convInputTrimGain, double outputMakeupGain)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    if (stream != CPQ_ALL_STREAMS && (stream < 0 || stream >= e->desc.n_streams))
        return fail(e, CPQ_ERR_INVALID_ARG, "stream %d out of range", stream);
    if (!std::isfinite(convInputTrimGain) || !std::isfinite(outputMakeupGain))
        return fail(e, CPQ_ERR_INVALID_ARG, "gains must be finite");
    CPQ_HIP(e, hipSetDevice(e->device));
    const int S = e->desc.n_streams;
    if (!e->trimDev) {
        if (hipMalloc((void**)&e->trimDev, sizeof(double) * S) != hipSuccess ||
            hipMalloc((void**)&e->makeupDev, sizeof(double) * S) != hipSuccess)
            return fail(e, CPQ_ERR_OOM, "gain buffers could not be allocated");
    }
    const int s0 = (stream == CPQ_ALL_STREAMS) ? 0 : stream;
    const int s1 = (stream == CPQ_ALL_STREAMS) ? S : stream + 1;
    for (int s = s0; s < s1; ++s) {
        // the trim is applied only when it differs from 1 by more than 1e-12 (:440)
        e->trimHost[s] = std::fabs(convInputTrimGain - 1.0) > 1e-12 ? convInputTrimGain : 1.0;
        e->makeupHost[s] = outputMakeupGain;
    }
    e->anyTrim = e->anyMakeup = false;
    for (int s = 0; s < S; ++s) { e->anyTrim = e->anyTrim || e->trimHost[s] != 1.0; e->anyMakeup = e->anyMakeup || e->makeupHost[s] != 1.0; }
    CPQ_HIP(e, hipStreamSynchronize(e->stream));
    CPQ_HIP(e, hipMemcpy(e->trimDev, e->trimHost.data(), sizeof(double) * S, hipMemcpyHostToDevice));
    CPQ_HIP(e, hipMemcpy(e->makeupDev, e->makeupHost.data(), sizeof(double) * S, hipMemcpyHostToDevice));
    return CPQ_OK;
}

int32_t cpq_engine_set_conv_bypass(cpq_engine* e, int32_t bypassed)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    e->convBypassed = bypassed != 0;
    return CPQ_OK;
}

int32_t cpq_engine_process_block_device(cpq_engine* e, const double* dIn, double* dOut, int32_t nSamples)
{
    const int rc = checkCall(e, dIn, dOut, nSamples);
    if (rc != CPQ_OK) return rc;
    CPQ_HIP(e, hipSetDevice(e->device));
    return enqueueBoth(e, dIn, dOut, nSamples);
}

int32_t cpq_engine_process_block(cpq_engine* e, const double* in, double* out, int32_t nSamples)
{
    return viaStaging(e, in, out, nSamples, [e](const double* a, double* b, int n) { return enqueueBoth(e, a, b, n); });
}

// -------------------------------------------------------------------------- profiling
int32_t cpq_profile_enable(cpq_engine* e, int32_t on)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    e->profiling = on != 0;
    if (e->profiling) {
        // event pool up front, so that no hipEventCreate runs inside a timed region (ProfScope only creates on exhaustion)
        CPQ_HIP(e, hipSetDevice(e->device));
        constexpr size_t kPool = 96;
        for (auto& s : e->prof)
            while (s.freeList.size() + s.pending.size() < kPool) {
                std::pair<hipEvent_t, hipEvent_t> ev;
                CPQ_HIP(e, hipEventCreate(&ev.first));
                CPQ_HIP(e, hipEventCreate(&ev.second));
                s.freeList.push_back(ev);
            }
    }
    return CPQ_OK;
}

static int drainProfile(cpq_engine* e)
{
    CPQ_HIP(e, hipSetDevice(e->device));
    CPQ_HIP(e, hipStreamSynchronize(e->stream));
    for (auto& s : e->prof) {
        for (auto& ev : s.pending) {
            float ms = 0.0f;
            CPQ_HIP(e, hipEventElapsedTime(&ms, ev.first, ev.second));
            s.totalMs += ms;
            s.launches += 1;
            s.freeList.push_back(ev);
        }
        s.pending.clear();
    }
    return CPQ_OK;
}

int32_t cpq_profile_reset(cpq_engine* e)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    const int rc = drainProfile(e);
    for (auto& s : e->prof) { s.launches = 0; s.totalMs = 0.0; }
    return rc;
}

int32_t cpq_profile_read(cpq_engine* e, int32_t id, int64_t* launches, double* totalMs)
{
    if (!e || id < 0 || id >= CPQ_K_COUNT) return CPQ_ERR_INVALID_ARG;
    const int rc = drainProfile(e);
    if (rc != CPQ_OK) return rc;
    if (launches) *launches = e->prof[id].launches;
    if (totalMs) *totalMs = e->prof[id].totalMs;
    return CPQ_OK;
}

}  // extern "C"
