// engine_eq.cpp -- EQ and output filter: design, device tables, bypass / band-reset state machine, cpq_eq_* / cpq_outfilter_* (see engine_internal.hpp, include/convopeq_mi355x.h).
#include "engine_internal.hpp"

using namespace cpqi;

namespace cpqi {

// one cascade launch: the time-parallel kernels over every even number of samples, the lane-skewed one over a last odd sample (or everything)
int enqueueCascade(cpq_engine* e, const double* dIn, double* dOut, int64_t stride, int n, bool tp, int idTp, int idSeq,
                   const double* coef, const int* flags, const double* satGain, double* state, const double* tables,
                   bool streamPairs = false)
{
    const int nTp = tp ? n : 0;                   // the time-parallel kernels take any number of samples
    if (nTp > 0) {
        ProfScope p(e, idTp);
        cpq::launch_svf_cascade_tp(e->stream, dIn, dOut, stride, e->nCh, nTp, coef, flags, satGain, state, tables,
                                   e->svfChain, e->svfChainSpans, e->svfChainGrid);
    }
    if (n > nTp) {
        ProfScope p(e, idSeq);
        cpq::launch_svf_cascade(e->stream, dIn + nTp, dOut + nTp, stride, e->nCh, n - nTp, coef, flags, satGain,
                                state, streamPairs);
    }
    CPQ_HIP(e, hipGetLastError());
    return CPQ_OK;
}



// prepareToPlay / reset: bypassFadeGain.setCurrentAndTargetValue(requested ? 0 : 1), the effective flag follows the
// request (src/eqprocessor/EQProcessor.Core.cpp:596, 656, 802)
void syncEqBypass(cpq_engine* e)
{
    e->anyEqBypass = false;
    std::fill(e->eqResetPending.begin(), e->eqResetPending.end(), 0u);      // the caller zeroes every state anyway
    std::fill(e->agcResetPending.begin(), e->agcResetPending.end(), 0);
    e->anyEqReset = false;
    for (auto& b : e->eqBypass) {
        b.effective = b.requested;
        b.current = b.target = b.requested ? 0.0 : 1.0;
        b.step = 0.0;
        b.remaining = 0;
        e->anyEqBypass = e->anyEqBypass || b.requested || b.mode != 0;
    }
}

// Device tables of one stream's EQ as the reference would run it: createCoeffCache (bandActive = enabled && sr > 0,
// src/eqprocessor/EQProcessor.ProcessingCache.cpp:71-90) or, for basicPath, the band nodes of the basic process(block)
// (inactive for non-LP/HP bands within 0.01 dB of flat: createBandNode, Coefficients.cpp:48-53).  An active Mid/Side
// band sends the whole call through the basic path (Processing.cpp:1036-1044).
struct EqDesign {
    double coef[2][kBands][6];
    int flags[2][kBands];
    std::vector<double> tp;
    double satGain[2];
    bool tpSafe = true, midSide = false;
};

void designEqStream(const cpq_engine* e, const cpq_eq_params& p, bool basicPath, EqDesign& d)
{
    d.tp.assign((size_t)kBands * cpq::kSvfTpTableDoubles, 0.0);
    d.tpSafe = true;
    d.midSide = false;
    for (int b = 0; b < kBands; ++b)
        d.midSide = d.midSide || (p.bands[b].enabled && e->sampleRate > 0.0 && p.bands[b].channel_mode >= 3);
    const bool nodes = basicPath || d.midSide;
    for (int b = 0; b < kBands; ++b) {
        const cpq_eq_band& bp = p.bands[b];
        bool active = bp.enabled && e->sampleRate > 0.0;
        if (nodes && bp.type != 3 && bp.type != 4 && std::fabs(bp.gain) < 0.01f) active = false;
        cpq_svf_coeffs c{ 0, 0, 0, 0, 0, 1, 0, 0 };
        if (active) {
            cpq::designSvf(bp.type, bp.frequency, bp.gain, bp.q, e->sampleRate, &c);
            d.tpSafe = cpq::buildSvfTpTables(c, &d.tp[(size_t)b * cpq::kSvfTpTableDoubles]) && d.tpSafe;
        }
        for (int ch = 0; ch < 2; ++ch) {
            const double v[6] = { c.a1, c.a2, c.a3, c.m0, c.m1, c.m2 };
            std::memcpy(d.coef[ch][b], v, sizeof(v));
            // Stereo -> both channels through the packed SSE2+FMA kernel; Left/Right -> one channel, scalar kernel
            // Mid/Side -> both channel lanes run the scalar kernel on the encoded component (flag bit 4 / 5)
            const bool on = active && (bp.channel_mode == 0 || bp.channel_mode == 1 + ch || bp.channel_mode >= 3);
            d.flags[ch][b] = (on ? 1 : 0) | ((bp.channel_mode != 0) ? 2 : 0) | (p.filter_structure == 1 ? 8 : 0) |
                             (bp.channel_mode == 3 ? 16 : 0) | (bp.channel_mode == 4 ? 32 : 0);
        }
    }
    // with AGC the total-gain ramp is replaced by processAGC (Processing.cpp:1256-1259): unity gain in the cascade kernel
    d.satGain[0] = (double)p.nonlinear_saturation;
    d.satGain[1] = p.agc_enabled ? 1.0 : cpq::totalGainLinear(p.total_gain_db);
    if (p.filter_structure == 1 || d.midSide) d.tpSafe = false;   // parallel structure and Mid/Side bands: lane-skewed kernel
}

// Switches what the device tables of one stream hold (on the engine's stream, in order with the kernels around it):
// 0 = the parameters as set, 1 = the basic path's band nodes, 2 = pass-through (EQ bypass in effect: nothing runs,
// not even the total gain, the AGC or the gain ramp).
int setEqStreamMode(cpq_engine* e, int s, int mode)
{
    auto& bp = e->eqBypass[s];
    if (bp.mode == mode) return CPQ_OK;
    const size_t c0 = (size_t)s * 2;
    if (mode == 2) {
        int zeros[2 * kBands] = {};
        const double sg[4] = { 0.0, 1.0, 0.0, 1.0 };
        { const int rcUp = stageUpload(e, e->svfFlags + c0 * kBands, zeros, sizeof(zeros)); if (rcUp != CPQ_OK) return rcUp; }
        { const int rcUp = stageUpload(e, e->svfSatGain + c0 * 2, sg, sizeof(sg)); if (rcUp != CPQ_OK) return rcUp; }
        e->eqTpSafe[s] = 1;
        e->eqMidSide[s] = 0;
        e->gainRamp[s].devUnity = true;
    } else {
        if (!e->eqParamsSet[s]) { bp.mode = mode; return CPQ_OK; }      // no parameters: every band inactive anyway
        EqDesign d;
        designEqStream(e, e->eqParamsHost[s], mode == 1, d);
        double sg[4] = { d.satGain[0], d.satGain[1], d.satGain[0], d.satGain[1] };
        { const int rcUp = stageUpload(e, e->svfCoef + c0 * kBands * 6, d.coef, sizeof(d.coef)); if (rcUp != CPQ_OK) return rcUp; }
        { const int rcUp = stageUpload(e, e->svfFlags + c0 * kBands, d.flags, sizeof(d.flags)); if (rcUp != CPQ_OK) return rcUp; }
        { const int rcUp = stageUpload(e, e->svfSatGain + c0 * 2, sg, sizeof(sg)); if (rcUp != CPQ_OK) return rcUp; }
        { const int rcUp = stageUpload(e, e->svfTp + (size_t)s * d.tp.size(), d.tp.data(), d.tp.size() * sizeof(double)); if (rcUp != CPQ_OK) return rcUp; }
        e->eqTpSafe[s] = d.tpSafe ? 1 : 0;
        e->eqMidSide[s] = d.midSide ? 1 : 0;
        e->gainRamp[s].devUnity = false;          // the constant gain (or 1.0 with AGC) is on the device again
    }
    if (e->agcOn) {
        const int on = (mode != 2 && e->agcOnHost[s]) ? 1 : 0;
        { const int rcUp = stageUpload(e, e->agcOn + s, &on, sizeof(int)); if (rcUp != CPQ_OK) return rcUp; }
    }
    bp.mode = mode;
    return CPQ_OK;
}

// EQ over n samples (a multiple of the block) of rows `stride` apart.  pass: streams (or nullptr) that are bypassed for
// the whole range -- their device tables hold pass-through flags, and the host-side gain ramp must not move either.
int enqueueEqCore(cpq_engine* e, const double* dIn, double* dOut, int64_t stride, int n, const char* pass)
{
    bool tp = (e->eqMode == CPQ_EQ_MODE_AUTO);
    for (char s : e->eqTpSafe) tp = tp && s;
    const int cbs = n / e->B;            // callback blocks in this range (AGC is block-rate)
    const bool raggedTail = n % e->B != 0;     // CPQ_CALLS_ANY: the last callback of the call is shorter than the quantum
    // total-gain ramp (Processing.cpp:1262-1274): per callback setTargetValue / skip on a LinearRamp (50 ms);
    // evaluated on the host (scalar per-stream state), applied by the ramp kernel only while some stream is moving
    std::vector<int> rampOnHost;
    std::vector<double> rampHost;
    bool anyRamp = false;
    {
        const int S = e->desc.n_streams;
        const int total = std::max(1, (int)(e->sampleRate * 0.05 + 0.5));
        for (int s = 0; s < S; ++s) {
            auto& r = e->gainRamp[s];
            if (e->agcOnHost[s] || (pass && pass[s])) continue;
            const bool moving = r.remaining > 0 || std::fabs(r.target - r.wanted) > 1e-6 || r.current != r.wanted;
            if (!moving) continue;
            if (raggedTail)
                return fail(e, CPQ_ERR_UNSUPPORTED, "a total-gain ramp is running: the call must be whole callbacks of %d samples until it ends", e->B);
            if (!anyRamp) { rampOnHost.assign(S, 0); rampHost.assign((size_t)S * cbs * 2, 0.0); anyRamp = true; }
            rampOnHost[s] = 1;
            for (int t = 0; t < cbs; ++t) {
                if (std::fabs(r.target - r.wanted) > 1e-6) {           // setTargetValue
                    if (r.wanted != r.target) {
                        r.target = r.wanted;
                        const int steps = r.remaining > 0 ? r.remaining : total;
                        r.step = (r.target - r.current) / (double)steps;
                        r.remaining = steps;
                    }
                }
                const double start = r.current;
                if (r.remaining > 0) {                                  // skip(numSamples)
                    if (e->B >= r.remaining) { r.current = r.target; r.remaining = 0; }
                    else { r.current += r.step * (double)e->B; r.remaining -= e->B; }
                }
                rampHost[((size_t)s * cbs + t) * 2] = start;
                rampHost[((size_t)s * cbs + t) * 2 + 1] = (r.current - start) / (double)e->B;
            }
        }
        // streams whose gain is applied by the ramp kernel need unity gain in the cascade kernel, and back again
        for (int s = 0; s < S; ++s) {
            auto& r = e->gainRamp[s];
            if (e->agcOnHost[s] || (pass && pass[s])) continue;
            const bool needUnity = anyRamp && rampOnHost[s];
            if (needUnity != r.devUnity) {
                const double g = needUnity ? 1.0 : r.wanted;
                for (int ch = 0; ch < 2; ++ch)
                    { const int rcUp = stageUpload(e, e->svfSatGain + (size_t)(2 * s + ch) * 2 + 1, &g, sizeof(double)); if (rcUp != CPQ_OK) return rcUp; }
                r.devUnity = needUnity;
            }
        }
        if (anyRamp) {
            const size_t cbMax = (size_t)e->tMax * e->P / e->B;
            if (!e->rampOn) {
                if (hipMalloc((void**)&e->rampOn, sizeof(int) * S) != hipSuccess ||
                    hipMalloc((void**)&e->rampGains, sizeof(double) * 2 * S * cbMax) != hipSuccess)
                    return fail(e, CPQ_ERR_OOM, "gain ramp buffers could not be allocated");
            }
            { const int rcUp = stageUpload(e, e->rampOn, rampOnHost.data(), sizeof(int) * S); if (rcUp != CPQ_OK) return rcUp; }
            { const int rcUp = stageUpload(e, e->rampGains, rampHost.data(), sizeof(double) * rampHost.size()); if (rcUp != CPQ_OK) return rcUp; }
        }
    }
    e->eqProcessed = true;
    if (e->anyAgc && raggedTail) return fail(e, CPQ_ERR_UNSUPPORTED, "the block-rate AGC needs calls of whole callbacks of %d samples", e->B);
    if (e->anyAgc) {
        ProfScope p(e, CPQ_K_MIX);
        cpq::launch_agc_block_rms(e->stream, dIn, stride, e->nCh, e->B, cbs, e->agcRmsIn);     // cachedInputRMS (:1116-1127)
    }
    bool midSide = false;
    for (char m : e->eqMidSide) midSide = midSide || m;
    const int rc = enqueueCascade(e, dIn, dOut, stride, n, tp, CPQ_K_SVF_TP, CPQ_K_SVF, e->svfCoef, e->svfFlags, e->svfSatGain,
                                  e->svfState, e->svfTp, midSide);
    if (rc != CPQ_OK) return rc;
    if (anyRamp) {
        ProfScope p(e, CPQ_K_MIX);
        cpq::launch_gain_ramp(e->stream, dOut, stride, e->desc.n_streams, e->B, cbs, e->rampGains, e->rampOn);
        CPQ_HIP(e, hipGetLastError());
    }
    if (!e->anyAgc) return rc;
    {
        ProfScope p(e, CPQ_K_MIX);
        // block coefficients of the tables prepareToPlay builds (src/eqprocessor/EQProcessor.Core.cpp:776-784)
        const double nn = (double)e->B, sr = e->sampleRate;
        const double bAtt = 1.0 - std::exp(-nn / (sr * 0.2)), bRel = 1.0 - std::exp(-nn / (sr * 2.0)),
                     bSm = 1.0 - std::exp(-nn / (sr * 0.2));
        cpq::launch_agc_block_rms(e->stream, dOut, stride, e->nCh, e->B, cbs, e->agcRmsOut);
        cpq::launch_agc_apply(e->stream, dOut, stride, e->desc.n_streams, e->B, cbs, e->agcRmsIn, e->agcRmsOut,
                              e->agcState, e->agcOn, e->agcGains, bAtt, bRel, bSm);
    }
    CPQ_HIP(e, hipGetLastError());
    return CPQ_OK;
}

// EQProcessor as DSPCore drives it with a per-stream bypass request (setBypassFromRT + process,
// src/audioengine/AudioEngine.Processing.DSPCoreDouble.cpp:384-413).  Per callback and stream the host replays the
// bypass state machine of the basic process(block) (Processing.cpp:499-526, 565-624, 977-1015): a change of the request
// starts the 5 ms LinearRamp bypassFadeGain; callbacks that start while it runs are processed through the basic path
// (band nodes) and cross-faded with the dry block sample by sample; once the fade-out is complete the EQ returns early
// (states, gain ramp and AGC frozen); releasing the bypass clears every filter state and fades back in.  The call is cut
// where some stream changes class, each piece runs the ordinary kernels with the streams' tables switched accordingly.
// n samples of rows `stride` apart: whole callbacks of e->B samples, or (no per-callback state in play) any n
static int enqueueEqRange(cpq_engine* e, const double* dIn, double* dOut, int64_t stride, int n)
{
    if (!e->eqSet) return fail(e, CPQ_ERR_NOT_READY, "cpq_eq_set_params has not been called");
    const int S = e->desc.n_streams;
    auto agcReset = [e](int s) -> int {        // rtAgcCurrentGainShadow = 1, envelopes = 0 (Processing.cpp:586-593, 1070-1077)
        if (e->agcState) CPQ_HIP(e, hipMemsetAsync(e->agcState + (size_t)s * 3, 0, sizeof(double) * 3, e->stream));
        e->agcResetPending[s] = 0;
        return CPQ_OK;
    };
    if (!e->anyEqBypass && !e->anyEqReset) {
        for (int s = 0; s < S; ++s)
            if (e->agcResetPending[s]) { const int rc = agcReset(s); if (rc != CPQ_OK) return rc; }
        e->eqProcessed = true;
        return enqueueEqCore(e, dIn, dOut, stride, n, nullptr);
    }
    if (n % e->B != 0)
        return fail(e, CPQ_ERR_UNSUPPORTED, "an EQ bypass transition or band reset is pending: the call must be whole callbacks of %d samples", e->B);
    const int cbs = n / e->B;
    const int total = std::max(1, (int)(e->sampleRate * 0.005 + 0.5));       // BYPASS_FADE_TIME_SEC (EQProcessor.h:564)
    enum : char { kNormal = 0, kFade = 1, kPass = 2 };
    std::vector<char> cls((size_t)S * cbs, kNormal);
    std::vector<uint32_t> reset((size_t)S * cbs, 0u);           // bands cleared at the start of the callback
    std::vector<char> released((size_t)S * cbs, 0);             // the bypass is released here: every band is to be cleared
    std::vector<std::vector<double>> gains(S);          // fade values of a stream's kFade callbacks, in order
    bool stillActive = false;
    for (int s = 0; s < S; ++s) {
        auto& b = e->eqBypass[s];
        for (int t = 0; t < cbs; ++t) {
            const double want = b.requested ? 0.0 : 1.0;
            if (std::fabs(b.target - want) > 1.0e-12) {
                if (!b.requested && b.effective) { released[(size_t)s * cbs + t] = 1; b.effective = false; }
                if (want != b.target) {                                       // LinearRamp::setTargetValue
                    b.target = want;
                    const int steps = b.remaining > 0 ? b.remaining : total;
                    b.step = (b.target - b.current) / (double)steps;
                    b.remaining = steps;
                }
            }
            const bool transition = b.remaining > 0;
            if (b.requested && !b.effective && !transition) b.effective = true;
            if (b.requested && b.effective && !transition) { cls[(size_t)s * cbs + t] = kPass; continue; }
            if (!transition) continue;
            cls[(size_t)s * cbs + t] = kFade;
            for (int i = 0; i < e->B && b.remaining > 0; ++i) {              // getNextValue while the ramp runs
                b.current += b.step;
                if (--b.remaining <= 0) b.current = b.target;
                gains[s].push_back(b.current);
            }
            if (b.remaining <= 0) b.effective = b.requested;
        }
    }
    // pending band resets: at the first callback that is fading (canSafelyResetState, :565-568) or whose input block is
    // silent; a fully bypassed callback returns before it gets there.  Silence is only known on the device: one small
    // kernel, one read-back and ONE stream synchronisation per call while a reset waits on a stream that is playing.
    {
        bool needSilence = false;
        for (int s = 0; s < S && !needSilence; ++s) {
            if (!e->eqResetPending[s]) continue;
            for (int t = 0; t < cbs; ++t) {
                if (cls[(size_t)s * cbs + t] == kFade) break;
                if (cls[(size_t)s * cbs + t] == kNormal) { needSilence = true; break; }
            }
        }
        if (needSilence) {
            const size_t cbMax = (size_t)e->tMax * e->P / e->B;
            if (!e->silentDev) {
                if (hipMalloc((void**)&e->silentDev, sizeof(int) * S * cbMax) != hipSuccess ||
                    hipHostMalloc((void**)&e->silentHost, sizeof(int) * S * cbMax) != hipSuccess)
                    return fail(e, CPQ_ERR_OOM, "silence flags could not be allocated");
            }
            cpq::launch_block_silence(e->stream, dIn, stride, e->B, cbs, S, e->silentDev);
            CPQ_HIP(e, hipMemcpyAsync(e->silentHost, e->silentDev, sizeof(int) * (size_t)S * cbs, hipMemcpyDeviceToHost, e->stream));
            CPQ_HIP(e, hipStreamSynchronize(e->stream));
        }
        e->anyEqReset = false;
        for (int s = 0; s < S; ++s) {
            uint32_t pending = e->eqResetPending[s];
            for (int t = 0; t < cbs; ++t) {
                if (released[(size_t)s * cbs + t]) pending = 0xFFFFFFFFu;
                if (!pending) continue;
                const char k = cls[(size_t)s * cbs + t];
                if (k == kFade || (k == kNormal && needSilence && e->silentHost[(size_t)s * cbs + t])) {
                    reset[(size_t)s * cbs + t] = pending;
                    pending = 0u;
                }
            }
            e->eqResetPending[s] = pending;
            e->anyEqReset = e->anyEqReset || pending != 0u;
        }
    }
    e->eqProcessed = true;
    std::vector<char> pass(S);
    std::vector<int> onHost(S), lenHost(S);
    std::vector<double> endHost(S), gainsHost;
    std::vector<size_t> used(S, 0);                     // fade values of the stream consumed by earlier pieces
    int rc = CPQ_OK;
    for (int c0 = 0; c0 < cbs && rc == CPQ_OK;) {
        int c1 = c0 + 1;
        auto sameClass = [&](int t) {
            for (int s = 0; s < S; ++s)
                if (cls[(size_t)s * cbs + t] != cls[(size_t)s * cbs + c0] || reset[(size_t)s * cbs + t]) return false;
            return true;
        };
        while (c1 < cbs && sameClass(c1)) ++c1;
        const int64_t off = (int64_t)c0 * e->B;
        const int nSeg = (c1 - c0) * e->B;
        bool anyFade = false;
        int cap = 1;
        for (int s = 0; s < S && rc == CPQ_OK; ++s) {
            const char k = cls[(size_t)s * cbs + c0];
            const uint32_t mask = reset[(size_t)s * cbs + c0];
            if (mask == 0xFFFFFFFFu) {             // every band of the stream, Mid / Side states included
                CPQ_HIP(e, hipMemsetAsync(e->svfState + (size_t)s * 2 * kBands * 2, 0, sizeof(double) * 2 * kBands * 2, e->stream));
            } else if (mask) {
                for (int b = 0; b < kBands; ++b)
                    if (mask & (1u << b))
                        for (int ch = 0; ch < 2; ++ch)
                            CPQ_HIP(e, hipMemsetAsync(e->svfState + ((size_t)(2 * s + ch) * kBands + b) * 2, 0, sizeof(double) * 2, e->stream));
            }
            rc = setEqStreamMode(e, s, k == kPass ? 2 : k == kFade ? 1 : 0);
            if (rc == CPQ_OK && e->agcResetPending[s] && k != kPass) rc = agcReset(s);      // a bypassed block returns before it
            pass[s] = k == kPass;
            onHost[s] = k == kFade;
            lenHost[s] = 0;
            endHost[s] = 1.0;
            if (k == kFade) {
                anyFade = true;
                lenHost[s] = (int)std::min<size_t>(gains[s].size() - used[s], (size_t)nSeg);
                // past the end of the ramp getNextValue keeps returning its final value
                endHost[s] = gains[s].empty() ? e->eqBypass[s].current : gains[s][std::min(gains[s].size(), used[s] + (size_t)nSeg) - 1];
                cap = std::max(cap, lenHost[s]);
            }
        }
        if (rc != CPQ_OK) break;
        if (anyFade) {
            if (!e->eqDry) {
                if (hipMalloc((void**)&e->eqDry, sizeof(double) * (size_t)e->nCh * e->tMax * e->P) != hipSuccess ||
                    hipMalloc((void**)&e->blendOn, sizeof(int) * S) != hipSuccess ||
                    hipMalloc((void**)&e->blendLen, sizeof(int) * S) != hipSuccess ||
                    hipMalloc((void**)&e->blendEnd, sizeof(double) * S) != hipSuccess)
                    return fail(e, CPQ_ERR_OOM, "EQ bypass cross-fade buffers could not be allocated");
            }
            if (cap > e->blendCap) {
                if (e->blendGains) (void)hipFree(e->blendGains);
                e->blendGains = nullptr;
                e->blendCap = 0;
                if (hipMalloc((void**)&e->blendGains, sizeof(double) * (size_t)S * cap) != hipSuccess)
                    return fail(e, CPQ_ERR_OOM, "EQ bypass cross-fade buffers could not be allocated");
                e->blendCap = cap;
            }
            gainsHost.assign((size_t)S * e->blendCap, 0.0);
            for (int s = 0; s < S; ++s) {
                if (!onHost[s]) continue;
                std::memcpy(&gainsHost[(size_t)s * e->blendCap], gains[s].data() + used[s], sizeof(double) * (size_t)lenHost[s]);
                used[s] += (size_t)lenHost[s];
            }
            { const int rcUp = stageUpload(e, e->blendOn, onHost.data(), sizeof(int) * S); if (rcUp != CPQ_OK) return rcUp; }
            { const int rcUp = stageUpload(e, e->blendLen, lenHost.data(), sizeof(int) * S); if (rcUp != CPQ_OK) return rcUp; }
            { const int rcUp = stageUpload(e, e->blendEnd, endHost.data(), sizeof(double) * S); if (rcUp != CPQ_OK) return rcUp; }
            { const int rcUp = stageUpload(e, e->blendGains, gainsHost.data(), sizeof(double) * gainsHost.size()); if (rcUp != CPQ_OK) return rcUp; }
            ProfScope p(e, CPQ_K_MIX);
            cpq::launch_rows_copy(e->stream, dIn, stride, off, e->eqDry, (int64_t)nSeg, 0, nSeg, e->nCh);
        }
        rc = enqueueEqCore(e, dIn + off, dOut + off, stride, nSeg, pass.data());
        if (rc == CPQ_OK && anyFade) {
            ProfScope p(e, CPQ_K_MIX);
            cpq::launch_bypass_blend(e->stream, dOut + off, stride, e->eqDry, (int64_t)nSeg, nSeg, e->nCh, e->blendOn,
                                     e->blendLen, e->blendEnd, e->blendGains, e->blendCap);
            CPQ_HIP(e, hipGetLastError());
        }
        c0 = c1;
    }
    // after the pieces ran: a stream whose tables were left on the basic path's band nodes (its fade ended inside this call)
    // still needs the next call to come through here, which puts the parameter tables back
    for (const auto& b : e->eqBypass) stillActive = stillActive || b.requested || b.effective || b.remaining > 0 || b.mode != 0;
    e->anyEqBypass = stillActive;
    return rc;
}

// A call whose last callback is shorter than the quantum (CPQ_CALLS_ANY: n = k B + r) runs as the whole callbacks followed
// by ONE callback of r samples, as the reference's process(block) sees them (src/convolver/ConvolverProcessor.Runtime.cpp:
// 659-682 cuts the host block the same way): the total-gain ramp skips r samples, the AGC takes its RMS and its block
// coefficients (table[numSamples], src/eqprocessor/EQProcessor.Processing.cpp:383-400) over r samples, the bypass fade
// draws r values.  Without per-callback state in play the cascade simply runs over all n samples.
int enqueueEq(cpq_engine* e, const double* dIn, double* dOut, int n)
{
    const int r = n % e->B;
    bool perCallback = e->anyAgc || e->anyEqBypass || e->anyEqReset;
    for (size_t s = 0; s < e->gainRamp.size() && !perCallback; ++s) {
        const auto& g = e->gainRamp[s];
        perCallback = !e->agcOnHost[s] && (g.remaining > 0 || std::fabs(g.target - g.wanted) > 1e-6 || g.current != g.wanted);
    }
    if (r == 0 || !perCallback) return enqueueEqRange(e, dIn, dOut, (int64_t)n, n);
    int rc = CPQ_OK;
    if (n - r > 0) rc = enqueueEqRange(e, dIn, dOut, (int64_t)n, n - r);
    if (rc != CPQ_OK) return rc;
    struct QuantumOf {            // the short callback is a callback of r samples to everything below
        cpq_engine* e; int saved;
        QuantumOf(cpq_engine* e_, int b) : e(e_), saved(e_->B) { e->B = b; }
        ~QuantumOf() { e->B = saved; }
    } tail(e, r);
    return enqueueEqRange(e, dIn + (n - r), dOut + (n - r), (int64_t)n, r);
}

int enqueueOutFilter(cpq_engine* e, const double* dIn, double* dOut, int n)
{
    if (!e->ofSet) return fail(e, CPQ_ERR_NOT_READY, "cpq_outfilter_set_params has not been called");
    const bool tp = (e->eqMode == CPQ_EQ_MODE_AUTO) && e->ofTpSafe;
    return enqueueCascade(e, dIn, dOut, (int64_t)n, n, tp, CPQ_K_OUTFILT, CPQ_K_OUTFILT, e->ofCoef, e->ofFlags,
                          e->ofSatGain, e->ofState, e->ofTp);
}


}  // namespace cpqi

extern "C" {

// ---------------------------------------------------------------------------------- EQ
int32_t cpq_eq_set_params(cpq_engine* e, int32_t stream, const cpq_eq_params* p)
{
    if (!e || !p) return CPQ_ERR_INVALID_ARG;
    if (stream != CPQ_ALL_STREAMS && (stream < 0 || stream >= e->desc.n_streams))
        return fail(e, CPQ_ERR_INVALID_ARG, "stream %d out of range", stream);
    if (p->filter_structure != 0 && p->filter_structure != 1) return fail(e, CPQ_ERR_INVALID_ARG, "filter_structure must be 0 (serial) or 1 (parallel)");
    for (int b = 0; b < kBands; ++b)
        if (p->bands[b].enabled && (p->bands[b].channel_mode < 0 || p->bands[b].channel_mode > 4))
            return fail(e, CPQ_ERR_INVALID_ARG, "band %d: channel_mode must be 0..4 (Stereo, Left, Right, Mid, Side)", b);

    EqDesign d;
    designEqStream(e, *p, false, d);
    const auto& coef = d.coef;
    const auto& flags = d.flags;
    const std::vector<double>& tp = d.tp;
    bool tpSafe = d.tpSafe;
    const bool midSide = d.midSide;
    const double satGain[2] = { d.satGain[0], d.satGain[1] };

    CPQ_HIP(e, hipSetDevice(e->device));
    const int s0 = (stream == CPQ_ALL_STREAMS) ? 0 : stream;
    const int s1 = (stream == CPQ_ALL_STREAMS) ? e->desc.n_streams : stream + 1;
    std::vector<double> hc((size_t)(s1 - s0) * 2 * kBands * 6), hs((size_t)(s1 - s0) * 2 * 2);
    std::vector<int> hf((size_t)(s1 - s0) * 2 * kBands);
    std::vector<double> ht((size_t)(s1 - s0) * tp.size());
    for (int s = s0; s < s1; ++s)
        for (int ch = 0; ch < 2; ++ch) {
            const size_t ci = (size_t)(s - s0) * 2 + ch;
            std::memcpy(&hc[ci * kBands * 6], coef[ch], sizeof(coef[ch]));
            std::memcpy(&hf[ci * kBands], flags[ch], sizeof(flags[ch]));
            hs[ci * 2] = satGain[0];
            hs[ci * 2 + 1] = satGain[1];
        }
    for (int s = s0; s < s1; ++s) std::memcpy(&ht[(size_t)(s - s0) * tp.size()], tp.data(), tp.size() * sizeof(double));
    for (int s = s0; s < s1; ++s) { e->eqTpSafe[s] = tpSafe ? 1 : 0; e->eqMidSide[s] = midSide ? 1 : 0; }
    const size_t c0 = (size_t)s0 * 2;
    CPQ_HIP(e, hipStreamSynchronize(e->stream));
    CPQ_HIP(e, hipMemcpy(e->svfCoef + c0 * kBands * 6, hc.data(), hc.size() * sizeof(double), hipMemcpyHostToDevice));
    CPQ_HIP(e, hipMemcpy(e->svfFlags + c0 * kBands, hf.data(), hf.size() * sizeof(int), hipMemcpyHostToDevice));
    CPQ_HIP(e, hipMemcpy(e->svfSatGain + c0 * 2, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice));
    CPQ_HIP(e, hipMemcpy(e->svfTp + (size_t)s0 * tp.size(), ht.data(), ht.size() * sizeof(double), hipMemcpyHostToDevice));
    e->eqSet = true;   // streams never given parameters keep all bands inactive (pass-through)
    for (int s = s0; s < s1; ++s) { e->eqParamsHost[s] = *p; e->eqParamsSet[s] = 1; e->eqBypass[s].mode = 0; }
    for (int s = s0; s < s1; ++s) {
        auto& r = e->gainRamp[s];
        r.wanted = cpq::totalGainLinear(p->total_gain_db);
        r.devUnity = false;               // the upload above put the constant gain (or 1.0 with AGC) on the device
        if (!e->eqProcessed || p->agc_enabled) { r.current = r.target = r.wanted; r.step = 0.0; r.remaining = 0; }
    }
    for (int s = s0; s < s1; ++s) e->agcOnHost[s] = p->agc_enabled ? 1 : 0;
    e->anyAgc = false;
    for (int v : e->agcOnHost) e->anyAgc = e->anyAgc || v;
    if (e->anyAgc) {
        const int S = e->desc.n_streams;
        const size_t cbMax = (size_t)e->tMax * e->P / e->B;
        if (!e->agcOn) {
            if (hipMalloc((void**)&e->agcOn, sizeof(int) * S) != hipSuccess ||
                hipMalloc((void**)&e->agcState, sizeof(double) * 3 * S) != hipSuccess ||
                hipMalloc((void**)&e->agcRmsIn, sizeof(double) * e->nCh * cbMax) != hipSuccess ||
                hipMalloc((void**)&e->agcRmsOut, sizeof(double) * e->nCh * cbMax) != hipSuccess ||
                hipMalloc((void**)&e->agcGains, sizeof(double) * 2 * S * cbMax) != hipSuccess)
                return fail(e, CPQ_ERR_OOM, "AGC buffers could not be allocated");
            CPQ_HIP(e, hipMemset(e->agcState, 0, sizeof(double) * 3 * S));
        }
        CPQ_HIP(e, hipMemcpy(e->agcOn, e->agcOnHost.data(), sizeof(int) * S, hipMemcpyHostToDevice));
    } else if (e->agcOn) {
        CPQ_HIP(e, hipMemcpy(e->agcOn, e->agcOnHost.data(), sizeof(int) * e->desc.n_streams, hipMemcpyHostToDevice));
    }
    return CPQ_OK;
}

int32_t cpq_eq_set_bypass(cpq_engine* e, int32_t stream, int32_t bypassed)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    if (stream != CPQ_ALL_STREAMS && (stream < 0 || stream >= e->desc.n_streams))
        return fail(e, CPQ_ERR_INVALID_ARG, "stream %d out of range", stream);
    const int s0 = (stream == CPQ_ALL_STREAMS) ? 0 : stream;
    const int s1 = (stream == CPQ_ALL_STREAMS) ? e->desc.n_streams : stream + 1;
    for (int s = s0; s < s1; ++s) {
        auto& b = e->eqBypass[s];
        b.requested = bypassed != 0;
        if (!e->eqProcessed) {            // before the first callback the fade is synchronised, not run (Core.cpp:288, 802)
            b.effective = b.requested;
            b.current = b.target = b.requested ? 0.0 : 1.0;
            b.step = 0.0;
            b.remaining = 0;
        }
    }
    e->anyEqBypass = false;
    for (const auto& b : e->eqBypass) e->anyEqBypass = e->anyEqBypass || b.requested || b.effective || b.remaining > 0 || b.mode != 0;
    return CPQ_OK;
}

int32_t cpq_eq_request_band_reset(cpq_engine* e, int32_t stream, uint32_t bandMask)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    if (stream != CPQ_ALL_STREAMS && (stream < 0 || stream >= e->desc.n_streams))
        return fail(e, CPQ_ERR_INVALID_ARG, "stream %d out of range", stream);
    const int s0 = (stream == CPQ_ALL_STREAMS) ? 0 : stream;
    const int s1 = (stream == CPQ_ALL_STREAMS) ? e->desc.n_streams : stream + 1;
    // requestBandReset(-1) asks for every band (mask 0xFFFFFFFF); single bands keep only the 20 real bits
    const uint32_t m = bandMask == 0xFFFFFFFFu ? bandMask : (bandMask & ((1u << kBands) - 1u));
    for (int s = s0; s < s1; ++s) {
        e->eqResetPending[s] |= m;
        e->anyEqReset = e->anyEqReset || e->eqResetPending[s] != 0u;
    }
    return CPQ_OK;
}

int32_t cpq_eq_request_agc_reset(cpq_engine* e, int32_t stream)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    if (stream != CPQ_ALL_STREAMS && (stream < 0 || stream >= e->desc.n_streams))
        return fail(e, CPQ_ERR_INVALID_ARG, "stream %d out of range", stream);
    const int s0 = (stream == CPQ_ALL_STREAMS) ? 0 : stream;
    const int s1 = (stream == CPQ_ALL_STREAMS) ? e->desc.n_streams : stream + 1;
    for (int s = s0; s < s1; ++s) e->agcResetPending[s] = 1;
    return CPQ_OK;
}

int32_t cpq_eq_set_mode(cpq_engine* e, int32_t mode)
{
    if (!e || (mode != CPQ_EQ_MODE_AUTO && mode != CPQ_EQ_MODE_SEQUENTIAL)) return CPQ_ERR_INVALID_ARG;
    e->eqMode = mode;
    return CPQ_OK;
}

int32_t cpq_eq_reset(cpq_engine* e)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    syncEqBypass(e);
    return zeroRuntimeState(e, false, true);
}

int32_t cpq_eq_process_device(cpq_engine* e, const double* dIn, double* dOut, int32_t nSamples)
{
    const int rc = checkCall(e, dIn, dOut, nSamples);
    if (rc != CPQ_OK) return rc;
    CPQ_HIP(e, hipSetDevice(e->device));
    return enqueueEq(e, dIn, dOut, nSamples);
}

int32_t cpq_eq_process(cpq_engine* e, const double* in, double* out, int32_t nSamples)
{
    return viaStaging(e, in, out, nSamples, [e](const double* a, double* b, int n) { return enqueueEq(e, a, b, n); });
}

// ------------------------------------------------------------------- output filter (N2)
int32_t cpq_outfilter_design(int32_t convIsLast, int32_t hcMode, int32_t lcMode, int32_t lpMode, double sampleRate,
                             cpq_biquad_coeffs out[3])
{
    if (!out || hcMode < 0 || hcMode > 2 || lcMode < 0 || lcMode > 1 || lpMode < 0 || lpMode > 2) return CPQ_ERR_INVALID_ARG;
    cpq::designOutputFilter(convIsLast, hcMode, lcMode, lpMode, sampleRate, out);
    return CPQ_OK;
}

int32_t cpq_outfilter_set_params(cpq_engine* e, int32_t stream, int32_t convIsLast, int32_t hcMode, int32_t lcMode,
                                 int32_t lpMode)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    if (stream != CPQ_ALL_STREAMS && (stream < 0 || stream >= e->desc.n_streams))
        return fail(e, CPQ_ERR_INVALID_ARG, "stream %d out of range", stream);
    cpq_biquad_coeffs q[3];
    const int rc = cpq_outfilter_design(convIsLast, hcMode, lcMode, lpMode, e->sampleRate, q);
    if (rc != CPQ_OK) return fail(e, rc, "bad output filter mode");
    std::vector<double> coef((size_t)kBands * 6, 0.0), tp((size_t)kBands * cpq::kSvfTpTableDoubles, 0.0);
    std::vector<int> flags(kBands, 0);
    bool safe = true;
    for (int b = 0; b < 3; ++b) {
        const double v[6] = { q[b].b0, q[b].b1, q[b].b2, q[b].a1, q[b].a2, 0.0 };
        std::memcpy(&coef[(size_t)b * 6], v, sizeof(v));
        flags[b] = 1 | 4;       // active, DF-II-T section (an identity section is run like the reference runs it)
        safe = cpq::buildBiquadTpTables(q[b], &tp[(size_t)b * cpq::kSvfTpTableDoubles]) && safe;
    }
    CPQ_HIP(e, hipSetDevice(e->device));
    CPQ_HIP(e, hipStreamSynchronize(e->stream));
    const int s0 = (stream == CPQ_ALL_STREAMS) ? 0 : stream;
    const int s1 = (stream == CPQ_ALL_STREAMS) ? e->desc.n_streams : stream + 1;
    const double sg[2] = { 0.0, 1.0 };
    for (int s = s0; s < s1; ++s) { e->ofModesHost[s] = cpq_engine::OfModes{ convIsLast, hcMode, lcMode, lpMode }; e->ofModesSet[s] = 1; }
    for (int s = s0; s < s1; ++s) {
        for (int ch = 0; ch < 2; ++ch) {
            const size_t c = (size_t)2 * s + ch;
            CPQ_HIP(e, hipMemcpy(e->ofCoef + c * kBands * 6, coef.data(), coef.size() * sizeof(double), hipMemcpyHostToDevice));
            CPQ_HIP(e, hipMemcpy(e->ofFlags + c * kBands, flags.data(), flags.size() * sizeof(int), hipMemcpyHostToDevice));
            CPQ_HIP(e, hipMemcpy(e->ofSatGain + c * 2, sg, sizeof(sg), hipMemcpyHostToDevice));
        }
        CPQ_HIP(e, hipMemcpy(e->ofTp + (size_t)s * tp.size(), tp.data(), tp.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    e->ofTpSafe = e->ofTpSafe && safe;
    e->ofSet = true;
    return CPQ_OK;
}

int32_t cpq_engine_enable_output_filter(cpq_engine* e, int32_t on)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    e->ofInPath = on != 0;
    return CPQ_OK;
}

int32_t cpq_outfilter_reset(cpq_engine* e)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    CPQ_HIP(e, hipSetDevice(e->device));
    CPQ_HIP(e, hipMemsetAsync(e->ofState, 0, (size_t)e->nCh * kBands * 2 * sizeof(double), e->stream));
    CPQ_HIP(e, hipStreamSynchronize(e->stream));
    return CPQ_OK;
}

int32_t cpq_outfilter_process_device(cpq_engine* e, const double* dIn, double* dOut, int32_t nSamples)
{
    const int rc = checkCall(e, dIn, dOut, nSamples);
    if (rc != CPQ_OK) return rc;
    CPQ_HIP(e, hipSetDevice(e->device));
    return enqueueOutFilter(e, dIn, dOut, nSamples);
}

int32_t cpq_outfilter_process(cpq_engine* e, const double* in, double* out, int32_t nSamples)
{
    return viaStaging(e, in, out, nSamples, [e](const double* a, double* b, int n) { return enqueueOutFilter(e, a, b, n); });
}

}  // extern "C"
