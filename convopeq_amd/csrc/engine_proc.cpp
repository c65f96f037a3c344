// engine_proc.cpp -- processor-level convolver stage: dry delay ring, mix ramp, latency cross-fade, cpq_convproc_* (see engine_internal.hpp, include/convopeq_mi355x.h).
#include "engine_internal.hpp"

using namespace cpqi;

// ----------------------------------------------------------------- convolver, processor level (N1)
namespace cpqi {

// equalPowerSin, src/convolver/ConvolverProcessor.Runtime.cpp:26-31 (9th-order Taylor of sin(pi x / 2))
double equalPowerSin(double x)
{
    const double t = x * (3.141592653589793238462643383279502884 * 0.5);
    const double t2 = t * t;
    return t * (1.0 + t2 * (-1.0 / 6.0 + t2 * (1.0 / 120.0 + t2 * (-1.0 / 5040.0 + t2 * (1.0 / 362880.0)))));
}

int procDelayOf(const cpq_engine* e, int s)
{
    // algorithmLatency = conv->latency (layer-0 partSize == block size; direct head unsupported),
    // irPeakLatency clamped like :266-277 (MAX_BLOCK_SIZE 524288, MAX_IR_LATENCY 2^21)
    const int alg = e->directHead ? 0 : std::min(e->P0, 524288);    // storedDirectHeadEnabled ? 0 : latency (:266); latency = layer-0 partSize
    const int peak = std::min(std::max(0, (int)e->procParams[s].ir_peak_latency), 2097152);
    return alg + peak;
}

int uploadProcParams(cpq_engine* e)
{
    const int S = e->desc.n_streams;
    std::vector<double> g((size_t)S * 2);
    std::vector<int> d(S);
    int maxDelay = 0;
    for (int s = 0; s < S; ++s) {
        const double mix = (double)e->procParams[s].mix;                 // targetMixValue (float widened, :366)
        g[2 * s] = equalPowerSin(mix) * 1.0;                             // * CONVOLUTION_HEADROOM_GAIN
        g[2 * s + 1] = (mix < 0.999) ? equalPowerSin(1.0 - mix) : 0.0;   // needsDrySignal, :375, :676
        // !needsConvolution (:374, :573-585): the delayed dry signal is copied as it is -- matters when the convolver
        // still runs because a mix ramp is finishing in the same call
        if (!(mix > 0.001)) { g[2 * s] = 0.0; g[2 * s + 1] = 1.0; }
        d[s] = procDelayOf(e, s);
        maxDelay = std::max(maxDelay, d[s]);
    }
    CPQ_HIP(e, hipSetDevice(e->device));
    CPQ_HIP(e, hipStreamSynchronize(e->stream));
    if (!e->procGains) {
        CPQ_HIP(e, hipMalloc((void**)&e->procGains, sizeof(double) * 2 * S));
        CPQ_HIP(e, hipMalloc((void**)&e->procDelay, sizeof(int) * S));
    }
    // delay ring: the longest delay in sight (any IR that fits the engine: irPeakLatency < irLen) plus one call; a larger
    // request later grows it, keeping what it holds
    const int64_t need = (int64_t)std::max(maxDelay, e->P0 + e->desc.max_ir_len) + (int64_t)e->tMax * e->P + 1;
    if (need > e->dryRingSize) {
        const int size = nextPow2((int)std::min<int64_t>(need, (int64_t)1 << 30));
        double* ring = nullptr;
        if (hipMalloc((void**)&ring, sizeof(double) * (size_t)e->nCh * size) != hipSuccess)
            return fail(e, CPQ_ERR_OOM, "dry delay line of %d samples per channel could not be allocated", size);
        CPQ_HIP(e, hipMemset(ring, 0, sizeof(double) * (size_t)e->nCh * size));
        if (e->dryRing) {
            cpq::launch_ring_regrow(e->stream, e->dryRing, e->dryRingSize, ring, size, e->dryPos, e->nCh);
            CPQ_HIP(e, hipStreamSynchronize(e->stream));
            (void)hipFree(e->dryRing);
        }
        e->dryRing = ring;
        e->dryRingSize = size;
    }
    if (!e->latNew) {
        if (hipMalloc((void**)&e->latNew, sizeof(int) * S) != hipSuccess || hipMalloc((void**)&e->latOld, sizeof(int) * S) != hipSuccess ||
            hipMalloc((void**)&e->latLen, sizeof(int) * S) != hipSuccess)
            return fail(e, CPQ_ERR_OOM, "latency buffers could not be allocated");
    }
    CPQ_HIP(e, hipMemcpy(e->procGains, g.data(), sizeof(double) * g.size(), hipMemcpyHostToDevice));
    CPQ_HIP(e, hipMemcpy(e->procDelay, d.data(), sizeof(int) * d.size(), hipMemcpyHostToDevice));
    return CPQ_OK;
}

int enqueueConvProc(cpq_engine* e, const double* dIn, double* dOut, int n)
{
    if (!e->procGains) { const int rc = uploadProcParams(e); if (rc != CPQ_OK) return rc; }
    {
        // feasibility first, before the ramp replay below consumes samples: resting ONE convolver needs that stream in a plan
        // group.  Which convolvers rest in THIS call is predicted from the ramps exactly as the replay below will decide it (a
        // dry-only stream keeps convolving while its mix ramp runs: streams set dry-only together can still differ for a call or
        // two when their ramps have different lengths left) -- nothing is mutated here.
        const int Sn = e->desc.n_streams;
        bool allWant = true, anyWant = false;
        std::vector<char> want((size_t)Sn, 0);
        for (int s = 0; s < Sn; ++s) {
            bool willRamp = false;
            if (!e->procBypass[s]) {
                const auto& r = e->mixRamp[s];
                const double tgt = (double)e->procParams[s].mix;
                const bool retarget = std::fabs(r.target - tgt) > 1.0e-5 && tgt != r.target;
                willRamp = (retarget ? (r.remaining > 0 ? r.remaining : r.totalSteps) : r.remaining) > 0;
            }
            want[(size_t)s] = e->procBypass[s] || (e->procDryOnly[s] && !willRamp);
            allWant = allWant && want[(size_t)s];
            anyWant = anyWant || want[(size_t)s];
        }
        if (anyWant && !allWant)
            for (int s = 0; s < Sn; ++s)
                if (want[(size_t)s] && e->groupOf[(size_t)s] < 0)
                    return fail(e, CPQ_ERR_UNSUPPORTED, "stream %d: a per-stream bypass / dry-only needs the stream on the reference's own "
                                "layer plan (CPQ_CALLS_ANY, CPQ_SCHED_REFERENCE_NUC or a FilterSpec plan with tail layers); on the "
                                "uniform path set it for CPQ_ALL_STREAMS", s);
    }
    // mix smoothing: per callback the reference moves the ramp's target to the current mix (:366-371) and, while the ramp
    // is running at the START of a callback, mixes that whole callback with per-sample gains equalPowerSin(getNextValue())
    // (:591-607).  Parameters only change between calls, so the smoothed region is a prefix of the call.
    const int S = e->desc.n_streams;
    std::vector<int> mixRampLenHost;
    std::vector<double> rampHost;
    bool anyRamp = false;
    int rampStride = 0;                 // samples per stream in rampHost / on the device: the longest smoothed prefix
    {
        for (int s = 0; s < S; ++s) {
            if (e->procBypass[s]) continue;             // the bypass path does not touch the smoother (:123-186)
            auto& r = e->mixRamp[s];
            const double tgt = (double)e->procParams[s].mix;
            if (std::fabs(r.target - tgt) > 1.0e-5 && tgt != r.target) {                 // setTargetValue
                r.target = tgt;
                const int steps = r.remaining > 0 ? r.remaining : r.totalSteps;
                r.step = (r.target - r.current) / (double)steps;
                r.remaining = steps;
            }
            if (r.remaining <= 0) continue;
            if (!anyRamp) { mixRampLenHost.assign(S, 0); anyRamp = true; }
            mixRampLenHost[s] = (int)std::min<int64_t>(n, ((int64_t)r.remaining + e->B - 1) / e->B * e->B);
            rampStride = std::max(rampStride, mixRampLenHost[s]);
        }
        if (anyRamp) rampHost.assign((size_t)S * rampStride * 2, 0.0);
        for (int s = 0; s < S && anyRamp; ++s) {
            auto& r = e->mixRamp[s];
            for (int i = 0; i < mixRampLenHost[s]; ++i) {
                if (r.remaining > 0) {                                                  // getNextValue
                    r.current += r.step;
                    if (--r.remaining <= 0) r.current = r.target;
                }
                rampHost[((size_t)s * rampStride + i) * 2] = equalPowerSin(r.current) * 1.0;
                rampHost[((size_t)s * rampStride + i) * 2 + 1] = equalPowerSin(1.0 - r.current);
            }
        }
    }
    if (anyRamp) {
        if (!e->mixRampLen && hipMalloc((void**)&e->mixRampLen, sizeof(int) * S) != hipSuccess)
            return fail(e, CPQ_ERR_OOM, "mix-ramp buffers could not be allocated");
        if (rampStride > e->mixRampCap) {
            if (e->mixRampGains) (void)hipFree(e->mixRampGains);
            e->mixRampGains = nullptr;
            e->mixRampCap = 0;
            if (hipMalloc((void**)&e->mixRampGains, sizeof(double) * 2 * (size_t)S * rampStride) != hipSuccess)
                return fail(e, CPQ_ERR_OOM, "mix-ramp buffers could not be allocated");
            e->mixRampCap = rampStride;
        }
        { const int rcUp = stageUpload(e, e->mixRampLen, mixRampLenHost.data(), sizeof(int) * S); if (rcUp != CPQ_OK) return rcUp; }
        { const int rcUp = stageUpload(e, e->mixRampGains, rampHost.data(), sizeof(double) * rampHost.size()); if (rcUp != CPQ_OK) return rcUp; }
    }
    const bool firstCall = !e->procProcessed;
    e->procProcessed = true;
    // needsConvolution = isSmoothing || mix > 0.001 (:374); bypassed: the convolver is not called (:123-186).  A stream whose
    // convolver rests while others run is moved to a plan group of its own and that group is not processed.
    std::vector<char> rest((size_t)S, 0);
    bool allRest = true, anyRest = false;
    for (int s = 0; s < S; ++s) {
        const bool ramping = anyRamp && mixRampLenHost[s] > 0;
        rest[s] = e->procBypass[s] || (e->procDryOnly[s] && !ramping);
        allRest = allRest && rest[s];
        anyRest = anyRest || rest[s];
    }
    const bool skipConv = allRest;
    if (!allRest && (anyRest || !e->groups.empty()))
        for (int s = 0; s < S; ++s) {
            if (e->groupOf[(size_t)s] < 0) { if (rest[s]) return setStreamFrozen(e, s, true); continue; }     // reports why not
            const int rc = setStreamFrozen(e, s, rest[s] != 0);
            if (rc != CPQ_OK) return rc;
        }
    {
        std::vector<int> wetOn((size_t)S);
        for (int s = 0; s < S; ++s) wetOn[s] = rest[s] ? 0 : 1;
        if (wetOn != e->procWetOnHost) {
            if (!e->procWetOn && hipMalloc((void**)&e->procWetOn, sizeof(int) * S) != hipSuccess)
                return fail(e, CPQ_ERR_OOM, "processor-level flags could not be allocated");
            const int rc = stageUpload(e, e->procWetOn, wetOn.data(), sizeof(int) * S);
            if (rc != CPQ_OK) return rc;
            e->procWetOnHost = wetOn;
        }
    }
    // the call's input goes into the delay ring before the convolver may overwrite it (in-place calls)
    const long long pos0 = e->dryPos;
    {
        ProfScope p(e, CPQ_K_MIX);
        cpq::launch_ring_put(e->stream, dIn, (int64_t)n, n, e->dryRing, e->dryRingSize, pos0, e->nCh);
    }
    e->dryPos += n;
    if (!skipConv) {
        e->honourFrozen = true;               // resting streams: their plan groups and direct heads are not run
        const int rc = enqueueConv(e, dIn, dOut, n);
        e->honourFrozen = false;
        if (rc != CPQ_OK) return rc;
    }
    // Latency compensation per callback and stream (:263-290): a total latency that moved by >= 2 samples starts, unless
    // one is running, a 20 ms cross-fade of the dry read from the delay in use to the new one; the callbacks that start
    // while it runs blend sample by sample until the ramp ends (:394-540).  The bypass reads at the present latency
    // (:141-145).  Ranges of the call between the callbacks where some stream starts a fade go to one launch each.
    const int cbs = (n + e->B - 1) / e->B;                  // callbacks (chunks of the call quantum; the last one may be shorter)
    auto lenOf = [&](int t) { return std::min(e->B, n - t * e->B); };
    const int xTotal = std::max(1, (int)(e->sampleRate * 0.02 + 0.5));
    struct Range { int c0, c1; };
    std::vector<Range> ranges;
    std::vector<int> dNew, dOld, xLen;
    std::vector<std::vector<double>> xg;          // per range: [S][len]
    {
        std::vector<char> starts((size_t)cbs, 0);
        // pass 1: where do fades start (needs the per-stream replay, so replay on copies)
        {
            for (int s = 0; s < S; ++s) {
                if (e->procBypass[s]) continue;
                auto f = e->latFade[s];
                const double total = (double)procDelayOf(e, s);
                if (!f.primed || firstCall) {             // prepareToPlay: latency + irLatency, fade gain at 1 (Lifecycle.cpp:377-388)
                    f.latCurrent = f.latTarget = f.oldDelay = (double)std::min(e->P0 + std::min(std::max(0, (int)e->procParams[s].ir_peak_latency), 2097152), 2097152 + 524288);
                    f.current = f.target = 1.0; f.remaining = 0; f.primed = true;
                }
                for (int t = 0; t < cbs; ++t) {
                    if (std::fabs(f.latTarget - total) >= 2.0 && f.remaining <= 0) {
                        f.oldDelay = f.latCurrent; f.current = 0.0; f.target = 1.0; f.step = 1.0 / (double)xTotal; f.remaining = xTotal;
                        f.latTarget = total;
                        if (t > 0) starts[t] = 1;
                    }
                    if (f.remaining > 0) {
                        f.remaining = std::max(0, f.remaining - lenOf(t));
                        if (f.remaining <= 0) { f.latCurrent = f.latTarget; f.oldDelay = f.latCurrent; }
                    }
                }
            }
        }
        int c0 = 0;
        for (int t = 1; t <= cbs; ++t)
            if (t == cbs || starts[t]) { ranges.push_back(Range{ c0, t }); c0 = t; }
    }
    const int R = (int)ranges.size();
    dNew.assign((size_t)R * S, 0); dOld.assign((size_t)R * S, 0); xLen.assign((size_t)R * S, 0);
    xg.assign(R, std::vector<double>());
    for (int s = 0; s < S; ++s) {
        auto& f = e->latFade[s];
        const int totalI = procDelayOf(e, s);
        if (e->procBypass[s]) {
            for (int r = 0; r < R; ++r) dNew[(size_t)r * S + s] = dOld[(size_t)r * S + s] = totalI;
            continue;
        }
        if (!f.primed || firstCall) {
            f.latCurrent = f.latTarget = f.oldDelay = (double)std::min(e->P0 + std::min(std::max(0, (int)e->procParams[s].ir_peak_latency), 2097152), 2097152 + 524288);
            f.current = f.target = 1.0; f.step = 0.0; f.remaining = 0; f.primed = true;
        }
        for (int r = 0; r < R; ++r) {
            std::vector<double> vals;
            bool fading = false;
            for (int t = ranges[r].c0; t < ranges[r].c1; ++t) {
                if (std::fabs(f.latTarget - (double)totalI) >= 2.0 && f.remaining <= 0) {
                    f.oldDelay = f.latCurrent;
                    f.current = 0.0; f.target = 1.0;                     // applyImmediateValueRT(0), setTargetValue(1)
                    f.step = (f.target - f.current) / (double)xTotal;
                    f.remaining = xTotal;
                    f.latTarget = (double)totalI;
                }
                if (t == ranges[r].c0) {
                    fading = f.remaining > 0;
                    dNew[(size_t)r * S + s] = fading ? (int)f.latTarget : (int)(f.latCurrent + 0.5);
                    dOld[(size_t)r * S + s] = (int)f.oldDelay;
                }
                if (f.remaining > 0) {
                    for (int i = 0; i < lenOf(t); ++i) {                 // getNextValue until the ramp has ended
                        f.current += f.step;
                        if (--f.remaining <= 0) f.current = f.target;
                        vals.push_back(f.current);
                        if (f.remaining <= 0) break;
                    }
                    if (f.remaining <= 0) { f.latCurrent = f.latTarget; f.oldDelay = f.latCurrent; }
                }
            }
            xLen[(size_t)r * S + s] = (int)vals.size();
            if (!vals.empty()) {
                if (xg[r].empty()) xg[r].assign((size_t)S * (xTotal + e->B), 0.0);
                std::memcpy(&xg[r][(size_t)s * (xTotal + e->B)], vals.data(), sizeof(double) * vals.size());
            }
        }
    }
    bool anyFadeValues = false;            // a fade of ONE value (a one-sample call) needs the buffer as well
    for (int r = 0; r < R; ++r) anyFadeValues = anyFadeValues || !xg[r].empty();
    if (anyFadeValues && e->latCap < xTotal + e->B) {
        if (e->latGains) (void)hipFree(e->latGains);
        e->latGains = nullptr;
        e->latCap = 0;
        if (hipMalloc((void**)&e->latGains, sizeof(double) * (size_t)S * (xTotal + e->B)) != hipSuccess)
            return fail(e, CPQ_ERR_OOM, "latency cross-fade buffer could not be allocated");
        e->latCap = xTotal + e->B;
    }
    for (int r = 0; r < R; ++r) {
        const int off = ranges[r].c0 * e->B, len = std::min(ranges[r].c1 * e->B, n) - off;
        const bool fade = !xg[r].empty();
        const std::vector<int> rn(dNew.begin() + (size_t)r * S, dNew.begin() + (size_t)(r + 1) * S);
        const std::vector<int> ro(dOld.begin() + (size_t)r * S, dOld.begin() + (size_t)(r + 1) * S);
        if (rn != e->latNewHost) {
            { const int rcUp = stageUpload(e, e->latNew, rn.data(), sizeof(int) * S); if (rcUp != CPQ_OK) return rcUp; }
            e->latNewHost = rn;
        }
        if (ro != e->latOldHost) {
            { const int rcUp = stageUpload(e, e->latOld, ro.data(), sizeof(int) * S); if (rcUp != CPQ_OK) return rcUp; }
            e->latOldHost = ro;
        }
        if (fade) {
            { const int rcUp = stageUpload(e, e->latLen, &xLen[(size_t)r * S], sizeof(int) * S); if (rcUp != CPQ_OK) return rcUp; }
            { const int rcUp = stageUpload(e, e->latGains, xg[r].data(), sizeof(double) * xg[r].size()); if (rcUp != CPQ_OK) return rcUp; }
        }
        ProfScope p(e, CPQ_K_MIX);
        cpq::launch_convproc_mix(e->stream, dOut + off, dOut + off, (int64_t)n, e->nCh, len, e->procGains, e->dryRing,
                                 e->dryRingSize, pos0 + off, e->latNew, e->latOld, fade ? e->latLen : nullptr, e->latGains,
                                 e->latCap, skipConv ? 0 : 1, anyRamp ? e->mixRampLen : nullptr, e->mixRampGains, rampStride, off,
                                 e->procWetOn);
    }
    CPQ_HIP(e, hipGetLastError());
    return CPQ_OK;
}

}  // namespace cpqi

extern "C" {

int32_t cpq_convproc_set_params(cpq_engine* e, int32_t stream, const cpq_convproc_params* p)
{
    if (!e || !p) return CPQ_ERR_INVALID_ARG;
    if (stream != CPQ_ALL_STREAMS && (stream < 0 || stream >= e->desc.n_streams))
        return fail(e, CPQ_ERR_INVALID_ARG, "stream %d out of range", stream);
    if (!(p->mix >= 0.0f && p->mix <= 1.0f)) return fail(e, CPQ_ERR_INVALID_ARG, "mix must be in [0, 1]");
    if (p->ir_peak_latency < 0) return fail(e, CPQ_ERR_INVALID_ARG, "ir_peak_latency must be >= 0");
    const bool dryOnly = !((double)p->mix > 0.001);        // needsConvolution, :374
    if (stream != CPQ_ALL_STREAMS && (p->bypassed || dryOnly) && e->groupOf[(size_t)stream] < 0 && e->irLoaded[2 * (size_t)stream])
        return fail(e, CPQ_ERR_UNSUPPORTED, "a per-stream bypass / dry-only rests ONE convolver: the stream must run on the reference's own "
                    "layer plan (CPQ_CALLS_ANY, CPQ_SCHED_REFERENCE_NUC or a FilterSpec plan with tail layers); on the uniform path set it "
                    "for CPQ_ALL_STREAMS");
    const int s0 = (stream == CPQ_ALL_STREAMS) ? 0 : stream;
    const int s1 = (stream == CPQ_ALL_STREAMS) ? e->desc.n_streams : stream + 1;
    if (p->smoothing_time_sec != 0.0f && !(p->smoothing_time_sec >= 0.01f && p->smoothing_time_sec <= 0.5f))
        return fail(e, CPQ_ERR_INVALID_ARG, "smoothing_time_sec must be 0 (default 0.1 s) or in [0.01, 0.5]");
    for (int s = s0; s < s1; ++s) {
        e->procParams[s] = *p;
        auto& r = e->mixRamp[s];
        const double t = p->smoothing_time_sec != 0.0f ? (double)p->smoothing_time_sec : 0.1;     // SMOOTHING_TIME_DEFAULT_SEC
        const int steps = (int)(e->sampleRate * t + 0.5);
        r.totalSteps = steps > 0 ? steps : 1;
        // before the first processor-level call (the reference's prepareToPlay: setCurrentAndTargetValue, Lifecycle.cpp:370)
        // the mix applies at once; afterwards it is the ramp's new target
        if (!e->procProcessed) { r.current = r.target = (double)p->mix; r.step = 0.0; r.remaining = 0; }
    }
    for (int s = s0; s < s1; ++s) { e->procBypass[s] = p->bypassed != 0; e->procDryOnly[s] = dryOnly; }
    return uploadProcParams(e);
}

int32_t cpq_convproc_delay(const cpq_engine* e, int32_t stream)
{
    if (!e || stream < 0 || stream >= e->desc.n_streams) return CPQ_ERR_INVALID_ARG;
    return procDelayOf(e, stream);
}

int32_t cpq_engine_set_conv_level(cpq_engine* e, int32_t level)
{
    if (!e || (level != CPQ_LEVEL_NUC && level != CPQ_LEVEL_PROCESSOR)) return CPQ_ERR_INVALID_ARG;
    e->convLevel = level;
    if (level == CPQ_LEVEL_NUC) clearFrozen(e);      // resting a stream is processor-level state
    return CPQ_OK;
}

int32_t cpq_convproc_process_device(cpq_engine* e, const double* dIn, double* dOut, int32_t nSamples)
{
    const int rc = checkCall(e, dIn, dOut, nSamples);
    if (rc != CPQ_OK) return rc;
    CPQ_HIP(e, hipSetDevice(e->device));
    return enqueueConvProc(e, dIn, dOut, nSamples);
}

int32_t cpq_convproc_process(cpq_engine* e, const double* in, double* out, int32_t nSamples)
{
    return viaStaging(e, in, out, nSamples, [e](const double* a, double* b, int n) { return enqueueConvProc(e, a, b, n); });
}

}  // extern "C"
