// cpq_api.cpp -- C ABI of libconvopeq_mi355x.so (see include/convopeq_mi355x.h).
//
// Host side of the engine: the device arena (the reference's per-buffer mkl_malloc manager,
// src/AlignedAllocation.h:22-163 + src/MKLNonUniformConvolver.h:288-365, collapsed into one HBM
// allocation laid out from (streams, partitions, ring slots, blocks per call)), the per-call kernel
// sequence, and the prepare/set_impulse/set_params control surface.  No CPU fallback exists: without a
// HIP device cpq_engine_create fails with CPQ_ERR_NO_DEVICE and nothing else can be called.
#include "convopeq_mi355x.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "host_design.hpp"
#include "kernels.hpp"

using cpq::kBands;

namespace {

std::string g_createError;
std::mutex g_createErrorMutex;

struct ProfileSlot {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> freeList;
    int64_t launches = 0;
    double totalMs = 0.0;
};

}  // namespace

// One tail layer of a FilterSpec plan, run at the reference's OWN partition size: the HC/LC gains (and the air-absorption
// damping) multiply every partition spectrum of the layer at that layer's FFT size, which folds time-aliased energy
// into the frame (src/MKLNonUniformConvolver.cpp:336-443) -- only reproducible with the same partitioning.
struct SpecTail {
    int P = 0, K = 0, kPad = 0, hRows = 0, ringSlots = 0, nbMax = 0, accCap = 0, outRing = 0;
    int head = 0, histSel = 0, accSel = 0, fill = 0;
    long long blocksDone = 0;
    double gain = 1.0;          // tail-layer gain applied by the delay-line reader
    char* mem = nullptr;
    double2 *X = nullptr, *XDN = nullptr, *H = nullptr, *HDN = nullptr, *Y = nullptr, *tw = nullptr, *tw2 = nullptr;
    double *hist[2] = { nullptr, nullptr }, *acc[2] = { nullptr, nullptr }, *z = nullptr, *ring = nullptr, *gainDev = nullptr;
    double2* scratch = nullptr;   // four-step FFT workspace (P > 4096): [max(nCh * nbMax, K)][P]
};

struct cpq_engine {
    cpq_engine_desc desc{};
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copyIn = nullptr, copyOut = nullptr;      // host-pointer entry points: upload / download beside the kernels
    hipEvent_t evIn[4] = {}, evDone[4] = {};
    std::string lastError;

    // geometry
    int nCh = 0;          // 2 * streams
    int B = 0;            // caller's block size (the reference's blockSize: layer plan, latency)
    int P = 0;            // internal partition size (samples) == complex bins per packed spectrum; multiple of B
    int kCap = 0;         // partition capacity per IR slot (multiple of kMacMaxTile)
    int hRows = 0;        // kCap + prefetch rows allocated per IR slot
    int ringSlots = 0;    // FDL ring slots per channel (power of two)
    int tMax = 0;
    int macTile = 16;

    // device arena
    char* arena = nullptr;
    int64_t arenaBytes = 0;
    double2* X = nullptr;       // [nCh][ringSlots][e->P]       FDL ring of packed spectra
    double2* XDN = nullptr;     // [nCh][ringSlots]           (DC, Nyquist) of every FDL slot
    double2* H = nullptr;       // [nCh][hRows][e->P]           IR partition spectra per IR slot
    double2* HDN = nullptr;     // [nCh][hRows]
    double2* Y = nullptr;       // [nCh][tMax][e->P]            accumulated output spectra of the call
    double* hist[2] = { nullptr, nullptr };   // [nCh][e->P]    overlap history, ping-pong
    double* stageIn = nullptr;  // [nCh][tMax*e->P]             staging for the host-pointer entry points
    double* stageOut = nullptr;
    double* mid = nullptr;      // [nCh][tMax*e->P]             conv <-> EQ hand-off (not used when in place)
    double* heffDev = nullptr;  // staging for one h_eff upload
    double* gainDev = nullptr;  // [P+1] spectral gains of a FilterSpec
    bool directHead = false;    // last set_impulse enabled the direct head (affects the processor-level dry delay)
    // direct head (allocated on first use): reversed, scaled head taps and tap count per IR slot, input history, output
    double* directIr = nullptr;         // [nCh slots][32]
    int* directTaps = nullptr;          // [nCh slots]
    double* directHist[2] = { nullptr, nullptr };   // [nCh][32] last input samples, ping-pong
    double* directOut = nullptr;        // [nCh][tMax * P]
    int directSel = 0;
    std::vector<int> directTapsHost;    // per IR slot
    bool anyDirect = false;
    int64_t heffCap = 0;
    double2* tw512 = nullptr;
    double2* tw1024 = nullptr;
    int* irSlot = nullptr;      // [nCh] device
    double* svfCoef = nullptr;  // [nCh][20][6]
    int* svfFlags = nullptr;    // [nCh][20]
    double* svfSatGain = nullptr;   // [nCh][2]
    double* svfState = nullptr; // [nCh][20][2]
    double* svfTp = nullptr;    // [streams][20][kSvfTpTableDoubles]  time-parallel kernel tables
    // output filter (N2): the same cascade kernels running DF-II-T sections in band slots 0..2
    double* ofCoef = nullptr;   // [nCh][20][6]  b0 b1 b2 a1 a2 -
    int* ofFlags = nullptr;     // [nCh][20]
    double* ofSatGain = nullptr;
    double* ofState = nullptr;  // [nCh][20][2]  w1 w2
    double* ofTp = nullptr;     // [streams][20][kSvfTpTableDoubles]
    bool ofSet = false, ofTpSafe = true, ofInPath = false;

    // run-time state
    int head = 0;               // ring slot of the next block
    int histSel = 0;
    int kActive = 0;            // max partitions over the loaded IRs (multiple of kMacMaxTile)
    int kMaxReal = 0;           // max real partition count (DC/Nyquist loop bound)
    std::vector<int> irSlotHost;
    std::vector<char> irLoaded; // per channel
    std::vector<int> irParts;   // per IR slot: partitions in use
    std::vector<char> slotSpecTail;   // per IR slot: loaded with a FilterSpec plan that has tail layers
    cpq_nuc_plan plan{};        // plan of the most recent set_impulse
    bool planValid = false;
    bool eqSet = false;
    std::vector<char> eqTpSafe; // per stream: time-parallel kernel proven guard-free
    std::vector<char> eqMidSide; // per stream: some active band filters the Mid or Side component
    // last parameters per stream, re-designed when prepare() changes the sample rate (EQProcessor::prepareToPlay rebuilds
    // its band nodes on a rate change, src/eqprocessor/EQProcessor.Core.cpp:679-826; OutputFilter::prepare likewise)
    std::vector<cpq_eq_params> eqParamsHost;
    std::vector<char> eqParamsSet;
    struct OfModes { int convIsLast, hc, lc, lp; };
    std::vector<OfModes> ofModesHost;
    std::vector<char> ofModesSet;
    int eqMode = CPQ_EQ_MODE_AUTO;
    int order = CPQ_ORDER_CONV_THEN_EQ;
    double sampleRate = 48000.0;

    // total-gain LinearRamp per stream (src/DspNumericPolicy.h:319-421; 50 ms, EQProcessor.h SMOOTHING_TIME_SEC)
    struct GainRamp { double current = 1.0, target = 1.0, step = 0.0; int remaining = 0; double wanted = 1.0; bool devUnity = false; };
    std::vector<GainRamp> gainRamp;     // per stream
    bool eqProcessed = false;           // a process call has consumed EQ parameters since prepare
    // EQ bypass per stream (EQProcessor::setBypassFromRT + the fade of the basic process(block),
    // src/eqprocessor/EQProcessor.Processing.cpp:499-526, 977-1015): LinearRamp bypassFadeGain over 5 ms
    struct EqBypass {
        bool requested = false, effective = false;
        double current = 1.0, target = 1.0, step = 0.0;
        int remaining = 0;
        int mode = 0;                   // what the device tables of the stream hold now: 0 parameters as set,
    };                                  // 1 band nodes of the basic path, 2 pass-through
    std::vector<EqBypass> eqBypass;
    // requestBandReset (EQProcessor.h; Processing.cpp:595-624): bands whose state is cleared at the first callback where
    // that is safe -- the block is silent, or a bypass fade is running
    std::vector<uint32_t> eqResetPending;
    bool anyEqReset = false;
    int* silentDev = nullptr;           // [streams][callbacks]
    int* silentHost = nullptr;          // pinned
    bool anyEqBypass = false;           // some stream is not in the plain "never bypassed" state
    // DSPCore block routing (src/audioengine/AudioEngine.Processing.DSPCoreDouble.cpp:384-470)
    std::vector<double> trimHost, makeupHost;   // per stream: convolverInputTrimGain (EQ -> conv order), outputMakeupGain
    double* trimDev = nullptr;
    double* makeupDev = nullptr;
    bool anyTrim = false, anyMakeup = false;
    bool convBypassed = false;          // state.convBypassed: the convolver stage is not called
    std::vector<char> ofPass;           // per stream: output filter tables hold pass-through flags (nothing active)
    double* eqDry = nullptr;            // [nCh][tMax * P] dry copy for the bypass cross-fade (allocated on first use)
    int* blendOn = nullptr;             // [streams]
    int* blendLen = nullptr;            // [streams] samples with their own fade gain
    double* blendEnd = nullptr;         // [streams] gain after those
    double* blendGains = nullptr;       // [streams][fade steps]
    int blendCap = 0;
    int* rampOn = nullptr;              // [streams] device
    double* rampGains = nullptr;        // [streams][callbacks][2] device

    // EQ AGC (allocated on first use)
    std::vector<int> agcOnHost;
    bool anyAgc = false;
    int* agcOn = nullptr;           // [streams]
    double* agcState = nullptr;     // [streams][3]
    double* agcRmsIn = nullptr;     // [nCh][callbacks]
    double* agcRmsOut = nullptr;
    double* agcGains = nullptr;     // [streams][callbacks][2]

    // layered (time-varying) reference semantics: per-layer convolutions + replay of the tail delay-line reader
    bool layered = false;
    cpq_nuc_plan layerPlan{};
    int layerRow[3] = { 0, 0, 0 };      // first IR row of each layer inside a channel's slot
    int layerK[3] = { 0, 0, 0 };        // partitions per layer
    double* layerOut = nullptr;         // [nTail][nCh][tMax*P]
    double* tailRing = nullptr;         // [nTail][nCh][tailRingSlots]
    int tailRingSlots = 0;
    void* tailState = nullptr;          // device: callback counter + read cursors
    long long* tailSched = nullptr;     // device: [nTail][tMax]

    // FilterSpec plans with tail layers (LTI-valid ones): layer 0 runs in the main path, each tail layer in a SpecTail
    std::vector<SpecTail> specTails;
    cpq_nuc_plan specPlan{};
    void* specState = nullptr;          // device: callback counter + delay-line read cursors of the tail layers
    long long* specSched = nullptr;     // device: [2][callbacks per call] read positions (-1 = the reader skips)

    // processor-level wrapper (N1)
    int convLevel = CPQ_LEVEL_NUC;
    std::vector<cpq_convproc_params> procParams;   // per stream
    bool procBypassed = false, procDryOnly = false;
    // mix smoothing (LinearRamp mixSmoother, src/ConvolverProcessor.h:945; Runtime.cpp:340-375, 591-607): per stream
    struct MixRamp { double current = 1.0, target = 1.0, step = 0.0; int remaining = 0, totalSteps = 4800; };
    std::vector<MixRamp> mixRamp;
    bool procProcessed = false;         // a processor-level call has run since create / prepare: parameter changes ramp
    int* mixRampLen = nullptr;             // [streams] device: leading samples of the call with per-sample gains
    double* mixRampGains = nullptr;     // [streams][mixRampCap][2] device (allocated when a ramp first runs)
    int mixRampCap = 0;
    double* procGains = nullptr;    // [streams][2] device
    int* procDelay = nullptr;       // [streams] device
    // dry delay line: a ring per channel (the reference's 4 Mi-sample delayBuffer, Runtime.cpp:378-391), sized for the
    // longest delay an IR of max_ir_len can ask for plus one call; every call's input is written before anything reads
    double* dryRing = nullptr;      // [nCh][dryRingSize] device, allocated on first use
    int dryRingSize = 0;
    long long dryPos = 0;           // absolute position of the next input sample
    // latency compensation (Runtime.cpp:263-290, 394-540): latencySmoother is only ever snapped, crossfadeGain runs 20 ms
    struct LatencyFade {
        double latCurrent = 0.0, latTarget = 0.0, oldDelay = 0.0;
        double current = 1.0, target = 1.0, step = 0.0;
        int remaining = 0;
        bool primed = false;        // latCurrent holds the prepareToPlay value (Lifecycle.cpp:380-388)
    };
    std::vector<LatencyFade> latFade;
    int* latNew = nullptr;          // [streams] device: delay of the dry read
    int* latOld = nullptr;          // [streams] delay faded out
    std::vector<int> latNewHost, latOldHost;    // what the two device arrays hold
    int* latLen = nullptr;          // [streams] samples of the range that are cross-faded
    double* latGains = nullptr;     // [streams][latCap]
    int latCap = 0;

    // profiling
    bool profiling = false;
    ProfileSlot prof[CPQ_K_COUNT];
};

namespace {

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

#define CPQ_HIP(e, call)                                                                             \
    do {                                                                                             \
        hipError_t err__ = (call);                                                                   \
        if (err__ != hipSuccess)                                                                     \
            return fail((e), CPQ_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(err__));      \
    } while (0)

int nextPow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }
int64_t alignUp(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

cpq::FftTables tables(const cpq_engine* e) { return cpq::FftTables{ e->tw512, e->tw1024 }; }

struct ProfScope {
    cpq_engine* e;
    int id;
    hipEvent_t stop = nullptr;
    ProfScope(cpq_engine* eng, int kid) : e(eng), id(kid)
    {
        if (!e->profiling) return;
        ProfileSlot& s = e->prof[id];
        std::pair<hipEvent_t, hipEvent_t> ev;
        if (!s.freeList.empty()) { ev = s.freeList.back(); s.freeList.pop_back(); }
        else { (void)hipEventCreate(&ev.first); (void)hipEventCreate(&ev.second); }
        (void)hipEventRecord(ev.first, e->stream);
        stop = ev.second;
        s.pending.push_back(ev);
    }
    ~ProfScope() { if (stop) (void)hipEventRecord(stop, e->stream); }
};

int checkCall(cpq_engine* e, const void* in, const void* out, int nSamples, int* T)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    if (!in || !out) return fail(e, CPQ_ERR_INVALID_ARG, "null buffer");
    if (nSamples <= 0 || nSamples % e->P != 0)
        return fail(e, CPQ_ERR_INVALID_ARG, "n_samples=%d is not a positive multiple of the partition size %d", nSamples,
                    e->P);
    const int t = nSamples / e->P;
    if (t > e->tMax)
        return fail(e, CPQ_ERR_INVALID_ARG, "n_samples=%d exceeds max_blocks_per_call=%d blocks of %d", nSamples,
                    e->desc.max_blocks_per_call, e->B);
    if ((reinterpret_cast<uintptr_t>(in) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u))
        return fail(e, CPQ_ERR_INVALID_ARG, "buffers must be 16-byte aligned");
    *T = t;
    return CPQ_OK;
}

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
                                    nb, (int64_t)t.hRows * t.P);
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
                                    e->P, e->nCh, kPad, e->ringSlots, e->head, T, (int64_t)e->hRows * e->P);
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
                            e->head, T, (int64_t)e->hRows * e->P);
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

int enqueueCascade(cpq_engine* e, const double* dIn, double* dOut, int64_t stride, int n, bool tp, int idTp, int idSeq,
                   const double* coef, const int* flags, const double* satGain, double* state, const double* tables,
                   bool streamPairs = false)
{
    const int nTp = tp ? (n / 512) * 512 : 0;     // the time-parallel kernel works in 512-sample spans
    if (nTp > 0) {
        ProfScope p(e, idTp);
        cpq::launch_svf_cascade_tp(e->stream, dIn, dOut, stride, e->nCh, nTp, coef, flags, satGain, state, tables);
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
        CPQ_HIP(e, hipMemcpyAsync(e->svfFlags + c0 * kBands, zeros, sizeof(zeros), hipMemcpyHostToDevice, e->stream));
        CPQ_HIP(e, hipMemcpyAsync(e->svfSatGain + c0 * 2, sg, sizeof(sg), hipMemcpyHostToDevice, e->stream));
        e->eqTpSafe[s] = 1;
        e->eqMidSide[s] = 0;
        e->gainRamp[s].devUnity = true;
    } else {
        if (!e->eqParamsSet[s]) { bp.mode = mode; return CPQ_OK; }      // no parameters: every band inactive anyway
        EqDesign d;
        designEqStream(e, e->eqParamsHost[s], mode == 1, d);
        double sg[4] = { d.satGain[0], d.satGain[1], d.satGain[0], d.satGain[1] };
        CPQ_HIP(e, hipMemcpyAsync(e->svfCoef + c0 * kBands * 6, d.coef, sizeof(d.coef), hipMemcpyHostToDevice, e->stream));
        CPQ_HIP(e, hipMemcpyAsync(e->svfFlags + c0 * kBands, d.flags, sizeof(d.flags), hipMemcpyHostToDevice, e->stream));
        CPQ_HIP(e, hipMemcpyAsync(e->svfSatGain + c0 * 2, sg, sizeof(sg), hipMemcpyHostToDevice, e->stream));
        CPQ_HIP(e, hipMemcpyAsync(e->svfTp + (size_t)s * d.tp.size(), d.tp.data(), d.tp.size() * sizeof(double),
                                  hipMemcpyHostToDevice, e->stream));
        e->eqTpSafe[s] = d.tpSafe ? 1 : 0;
        e->eqMidSide[s] = d.midSide ? 1 : 0;
        e->gainRamp[s].devUnity = false;          // the constant gain (or 1.0 with AGC) is on the device again
    }
    if (e->agcOn) {
        const int on = (mode != 2 && e->agcOnHost[s]) ? 1 : 0;
        CPQ_HIP(e, hipMemcpyAsync(e->agcOn + s, &on, sizeof(int), hipMemcpyHostToDevice, e->stream));
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
                    CPQ_HIP(e, hipMemcpyAsync(e->svfSatGain + (size_t)(2 * s + ch) * 2 + 1, &g, sizeof(double), hipMemcpyHostToDevice, e->stream));
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
            CPQ_HIP(e, hipMemcpyAsync(e->rampOn, rampOnHost.data(), sizeof(int) * S, hipMemcpyHostToDevice, e->stream));
            CPQ_HIP(e, hipMemcpyAsync(e->rampGains, rampHost.data(), sizeof(double) * rampHost.size(), hipMemcpyHostToDevice, e->stream));
        }
    }
    e->eqProcessed = true;
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
int enqueueEq(cpq_engine* e, const double* dIn, double* dOut, int T)
{
    if (!e->eqSet) return fail(e, CPQ_ERR_NOT_READY, "cpq_eq_set_params has not been called");
    const int n = T * e->P;
    const int S = e->desc.n_streams;
    if (!e->anyEqBypass && !e->anyEqReset) { e->eqProcessed = true; return enqueueEqCore(e, dIn, dOut, (int64_t)n, n, nullptr); }
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
        stillActive = stillActive || b.requested || b.effective || b.remaining > 0 || b.mode != 0;
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
            cpq::launch_block_silence(e->stream, dIn, (int64_t)n, e->B, cbs, S, e->silentDev);
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
            CPQ_HIP(e, hipMemcpyAsync(e->blendOn, onHost.data(), sizeof(int) * S, hipMemcpyHostToDevice, e->stream));
            CPQ_HIP(e, hipMemcpyAsync(e->blendLen, lenHost.data(), sizeof(int) * S, hipMemcpyHostToDevice, e->stream));
            CPQ_HIP(e, hipMemcpyAsync(e->blendEnd, endHost.data(), sizeof(double) * S, hipMemcpyHostToDevice, e->stream));
            CPQ_HIP(e, hipMemcpyAsync(e->blendGains, gainsHost.data(), sizeof(double) * gainsHost.size(), hipMemcpyHostToDevice, e->stream));
            ProfScope p(e, CPQ_K_MIX);
            cpq::launch_rows_copy(e->stream, dIn, (int64_t)n, off, e->eqDry, (int64_t)nSeg, 0, nSeg, e->nCh);
        }
        rc = enqueueEqCore(e, dIn + off, dOut + off, (int64_t)n, nSeg, pass.data());
        if (rc == CPQ_OK && anyFade) {
            ProfScope p(e, CPQ_K_MIX);
            cpq::launch_bypass_blend(e->stream, dOut + off, (int64_t)n, e->eqDry, (int64_t)nSeg, nSeg, e->nCh, e->blendOn,
                                     e->blendLen, e->blendEnd, e->blendGains, e->blendCap);
            CPQ_HIP(e, hipGetLastError());
        }
        c0 = c1;
    }
    e->anyEqBypass = stillActive;
    return rc;
}

int enqueueOutFilter(cpq_engine* e, const double* dIn, double* dOut, int T)
{
    if (!e->ofSet) return fail(e, CPQ_ERR_NOT_READY, "cpq_outfilter_set_params has not been called");
    const bool tp = (e->eqMode == CPQ_EQ_MODE_AUTO) && e->ofTpSafe;
    return enqueueCascade(e, dIn, dOut, (int64_t)T * e->P, T * e->P, tp, CPQ_K_OUTFILT, CPQ_K_OUTFILT, e->ofCoef, e->ofFlags,
                          e->ofSatGain, e->ofState, e->ofTp);
}

// Host-pointer entry points: H2D, the kernel sequence and D2H.  Long calls are cut into four time chunks (each a complete
// engine call: the state carries over on the engine's stream) so that the upload of chunk i+1 and the download of chunk
// i-1 run on two copy streams beside the kernels of chunk i.  With pinned caller buffers (cpq_host_register) the three
// overlap; pageable buffers take the plain upload / kernels / download sequence.
template <typename F>
int viaStaging(cpq_engine* e, const double* in, double* out, int nSamples, F&& body)
{
    int T = 0;
    int rc = checkCall(e, in, out, nSamples, &T);
    if (rc != CPQ_OK) return rc;
    CPQ_HIP(e, hipSetDevice(e->device));
    constexpr int kChunks = 4;
    auto pinned = [](const void* p) {
        hipPointerAttribute_t a{};
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain malloc'd memory
        return a.type == hipMemoryTypeHost;
    };
    // pageable buffers: the runtime stages every copy and blocks the host, so chunking only adds strided copies
    // (measured 1157 vs 1230 M samples/s); one upload, one download
    if (T < 32 || T % kChunks != 0 || !pinned(in) || !pinned(out)) {
        const size_t bytes = (size_t)e->nCh * nSamples * sizeof(double);
        CPQ_HIP(e, hipMemcpyAsync(e->stageIn, in, bytes, hipMemcpyHostToDevice, e->stream));
        rc = body(e->stageIn, e->stageOut, T);
        if (rc != CPQ_OK) return rc;
        CPQ_HIP(e, hipMemcpyAsync(out, e->stageOut, bytes, hipMemcpyDeviceToHost, e->stream));
        CPQ_HIP(e, hipStreamSynchronize(e->stream));
        return CPQ_OK;
    }
    if (!e->copyIn) {
        CPQ_HIP(e, hipStreamCreateWithFlags(&e->copyIn, hipStreamNonBlocking));
        CPQ_HIP(e, hipStreamCreateWithFlags(&e->copyOut, hipStreamNonBlocking));
        for (int i = 0; i < kChunks; ++i) {
            CPQ_HIP(e, hipEventCreateWithFlags(&e->evIn[i], hipEventDisableTiming));
            CPQ_HIP(e, hipEventCreateWithFlags(&e->evDone[i], hipEventDisableTiming));
        }
    }
    const int chunkT = T / kChunks;
    const size_t chunkLen = (size_t)chunkT * e->P;                       // samples per channel and chunk
    const size_t hostPitch = (size_t)nSamples * sizeof(double), devPitch = chunkLen * sizeof(double);
    auto download = [&](int i) -> int {
        CPQ_HIP(e, hipStreamWaitEvent(e->copyOut, e->evDone[i], 0));
        CPQ_HIP(e, hipMemcpy2DAsync(out + i * chunkLen, hostPitch, e->stageOut + (size_t)i * e->nCh * chunkLen, devPitch, devPitch,
                                    (size_t)e->nCh, hipMemcpyDeviceToHost, e->copyOut));
        return CPQ_OK;
    };
    for (int i = 0; i < kChunks; ++i) {
        double* dIn = e->stageIn + (size_t)i * e->nCh * chunkLen;
        double* dOut = e->stageOut + (size_t)i * e->nCh * chunkLen;
        CPQ_HIP(e, hipMemcpy2DAsync(dIn, devPitch, in + i * chunkLen, hostPitch, devPitch, (size_t)e->nCh, hipMemcpyHostToDevice,
                                    e->copyIn));
        CPQ_HIP(e, hipEventRecord(e->evIn[i], e->copyIn));
        CPQ_HIP(e, hipStreamWaitEvent(e->stream, e->evIn[i], 0));
        rc = body(dIn, dOut, chunkT);
        if (rc != CPQ_OK) { (void)hipDeviceSynchronize(); return rc; }
        CPQ_HIP(e, hipEventRecord(e->evDone[i], e->stream));
        if (i > 0) { rc = download(i - 1); if (rc != CPQ_OK) return rc; }
    }
    rc = download(kChunks - 1);
    if (rc != CPQ_OK) return rc;
    CPQ_HIP(e, hipStreamSynchronize(e->copyOut));
    CPQ_HIP(e, hipStreamSynchronize(e->stream));
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
        { const int rc = resetSpecTails(e); if (rc != CPQ_OK) return rc; }
        for (double* p : { e->directHist[0], e->directHist[1] })
            if (p) CPQ_HIP(e, hipMemsetAsync(p, 0, sizeof(double) * 32 * e->nCh, e->stream));
        if (e->tailState) CPQ_HIP(e, hipMemsetAsync(e->tailState, 0, 3 * sizeof(long long), e->stream));
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

}  // namespace

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
    if (d->block_size < 64 || d->block_size > 4096 || (d->block_size & (d->block_size - 1)))
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "block_size must be a power of two in [64, 4096]");
    if (d->semantics != CPQ_SEM_REFERENCE && d->semantics != CPQ_SEM_EXACT)
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "bad semantics");
    if (d->mac_tile != 0 && d->mac_tile != 4 && d->mac_tile != 8 && d->mac_tile != 16 && d->mac_tile != 32)
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "mac_tile must be 0, 4, 8, 16 or 32");
    if (d->schedule != CPQ_SCHED_UNIFORM && d->schedule != CPQ_SCHED_REFERENCE_NUC)
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "schedule must be CPQ_SCHED_UNIFORM or CPQ_SCHED_REFERENCE_NUC");
    if (d->schedule == CPQ_SCHED_REFERENCE_NUC &&
        (d->semantics != CPQ_SEM_REFERENCE || (d->partition_size != 0 && d->partition_size != d->block_size)))
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
    e->P = d->partition_size ? d->partition_size : d->block_size;
    if (e->P < e->B || e->P > 4096 || (e->P & (e->P - 1)) || ((int64_t)d->max_blocks_per_call * e->B) % e->P != 0) {
        const int p = e->P;
        delete e;
        return fail(nullptr, CPQ_ERR_INVALID_ARG,
                    "partition_size %d must be a power of two in [block_size, 4096] dividing block_size*max_blocks_per_call", p);
    }
    e->tMax = (int)(((int64_t)d->max_blocks_per_call * e->B) / e->P);     // partitions per call
    e->macTile = d->mac_tile;     // 0 = automatic (workgroup-cooperative kernel for calls of >= 32 blocks)

    // partition capacity from the longest h_eff the plan can produce for max_ir_len
    cpq_nuc_plan pl;
    if (cpq::computeNucPlan(d->max_ir_len, d->block_size, false, nullptr, &pl) != CPQ_OK) {
        delete e;
        return fail(nullptr, CPQ_ERR_INVALID_ARG, "cannot plan max_ir_len=%d", d->max_ir_len);
    }
    const int taps = (d->semantics == CPQ_SEM_REFERENCE) ? std::max(pl.heff_len, d->max_ir_len) : d->max_ir_len;
    const int kReal = (taps + e->P - 1) / e->P;
    e->kCap = (int)alignUp(kReal, cpq::kMacMaxTile);
    e->hRows = e->kCap + 4 * cpq::kMacMaxTile;   // zero rows read by the prefetch past the last partition (per layer in layered mode)
    e->ringSlots = nextPow2(e->kCap + cpq::kMacMaxTile + e->tMax);
    e->heffCap = (int64_t)e->kCap * e->P;

    // ---- arena layout
    struct Item { void** ptr; int64_t bytes; };
    const int64_t nCh = e->nCh;
    const int64_t callSamples = (int64_t)e->tMax * e->P;
    Item items[] = {
        { (void**)&e->X, nCh * e->ringSlots * e->P * (int64_t)sizeof(double2) },
        { (void**)&e->XDN, nCh * e->ringSlots * (int64_t)sizeof(double2) },
        { (void**)&e->H, nCh * e->hRows * e->P * (int64_t)sizeof(double2) },
        { (void**)&e->HDN, nCh * e->hRows * (int64_t)sizeof(double2) },
        { (void**)&e->Y, nCh * e->tMax * e->P * (int64_t)sizeof(double2) },
        { (void**)&e->hist[0], nCh * e->P * (int64_t)sizeof(double) },
        { (void**)&e->hist[1], nCh * e->P * (int64_t)sizeof(double) },
        { (void**)&e->stageIn, nCh * callSamples * (int64_t)sizeof(double) },
        { (void**)&e->stageOut, nCh * callSamples * (int64_t)sizeof(double) },
        { (void**)&e->mid, nCh * callSamples * (int64_t)sizeof(double) },
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
    };
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
    e->slotSpecTail.assign(e->nCh, 0);
    e->eqTpSafe.assign(d->n_streams, 1);   // no active band yet: trivially guard-free
    e->eqMidSide.assign(d->n_streams, 0);
    e->eqParamsHost.assign(d->n_streams, cpq_eq_params{});
    e->eqParamsSet.assign(d->n_streams, 0);
    e->eqBypass.assign(d->n_streams, cpq_engine::EqBypass{});
    e->eqResetPending.assign(d->n_streams, 0u);
    e->latFade.assign(d->n_streams, cpq_engine::LatencyFade{});
    e->trimHost.assign(d->n_streams, 1.0);
    e->makeupHost.assign(d->n_streams, 1.0);
    e->ofPass.assign(d->n_streams, 0);
    e->ofModesHost.assign(d->n_streams, cpq_engine::OfModes{ 0, 1, 0, 1 });
    e->ofModesSet.assign(d->n_streams, 0);
    e->procParams.assign(d->n_streams, cpq_convproc_params{ 1.0f, 0, 0, 0.0f });
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
    freeSpecTails(e);
    if (e->copyIn) {
        (void)hipStreamDestroy(e->copyIn);
        (void)hipStreamDestroy(e->copyOut);
        for (int i = 0; i < 4; ++i) { (void)hipEventDestroy(e->evIn[i]); (void)hipEventDestroy(e->evDone[i]); }
    }
    if (e->arena) (void)hipFree(e->arena);
    for (double* p : { e->dryRing, e->latGains, e->layerOut, e->tailRing, e->agcState, e->agcRmsIn, e->agcRmsOut, e->agcGains }) if (p) (void)hipFree(p);
    if (e->agcOn) (void)hipFree(e->agcOn);
    if (e->rampOn) (void)hipFree(e->rampOn);
    if (e->rampGains) (void)hipFree(e->rampGains);
    if (e->tailState) (void)hipFree(e->tailState);
    if (e->tailSched) (void)hipFree(e->tailSched);
    if (e->procGains) (void)hipFree(e->procGains);
    if (e->procDelay) (void)hipFree(e->procDelay);
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

int32_t cpq_engine_prepare(cpq_engine* e, double sampleRate, int32_t maxBlock)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    if (sampleRate <= 0.0) return fail(e, CPQ_ERR_INVALID_ARG, "sample rate must be positive");
    if (maxBlock <= 0 || maxBlock > e->P * e->tMax)
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

// ----------------------------------------------------------------- convolver, processor level (N1)
namespace {

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
    const int alg = e->directHead ? 0 : std::min(e->B, 524288);     // storedDirectHeadEnabled ? 0 : latency (:266)
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
    const int64_t need = (int64_t)std::max(maxDelay, e->B + e->desc.max_ir_len) + (int64_t)e->tMax * e->P + 1;
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

int enqueueConvProc(cpq_engine* e, const double* dIn, double* dOut, int T)
{
    if (!e->procGains) { const int rc = uploadProcParams(e); if (rc != CPQ_OK) return rc; }
    const int n = T * e->P;
    // mix smoothing: per callback the reference moves the ramp's target to the current mix (:366-371) and, while the ramp
    // is running at the START of a callback, mixes that whole callback with per-sample gains equalPowerSin(getNextValue())
    // (:591-607).  Parameters only change between calls, so the smoothed region is a prefix of the call.
    const int S = e->desc.n_streams;
    std::vector<int> mixRampLenHost;
    std::vector<double> rampHost;
    bool anyRamp = false;
    int rampStride = 0;                 // samples per stream in rampHost / on the device: the longest smoothed prefix
    if (!e->procBypassed) {
        for (int s = 0; s < S; ++s) {
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
        CPQ_HIP(e, hipMemcpyAsync(e->mixRampLen, mixRampLenHost.data(), sizeof(int) * S, hipMemcpyHostToDevice, e->stream));
        CPQ_HIP(e, hipMemcpyAsync(e->mixRampGains, rampHost.data(), sizeof(double) * rampHost.size(), hipMemcpyHostToDevice, e->stream));
        CPQ_HIP(e, hipStreamSynchronize(e->stream));         // the host vectors go out of scope
    }
    const bool firstCall = !e->procProcessed;
    e->procProcessed = true;
    const bool skipConv = e->procBypassed || (e->procDryOnly && !anyRamp);      // needsConvolution = isSmoothing || mix > 0.001
    // the call's input goes into the delay ring before the convolver may overwrite it (in-place calls)
    const long long pos0 = e->dryPos;
    {
        ProfScope p(e, CPQ_K_MIX);
        cpq::launch_ring_put(e->stream, dIn, (int64_t)n, n, e->dryRing, e->dryRingSize, pos0, e->nCh);
    }
    e->dryPos += n;
    if (!skipConv) {
        const int rc = enqueueConv(e, dIn, dOut, T);
        if (rc != CPQ_OK) return rc;
    }
    // Latency compensation per callback and stream (:263-290): a total latency that moved by >= 2 samples starts, unless
    // one is running, a 20 ms cross-fade of the dry read from the delay in use to the new one; the callbacks that start
    // while it runs blend sample by sample until the ramp ends (:394-540).  The bypass reads at the present latency
    // (:141-145).  Ranges of the call between the callbacks where some stream starts a fade go to one launch each.
    const int cbs = n / e->B;
    const int xTotal = std::max(1, (int)(e->sampleRate * 0.02 + 0.5));
    struct Range { int c0, c1; };
    std::vector<Range> ranges;
    std::vector<int> dNew, dOld, xLen;
    std::vector<std::vector<double>> xg;          // per range: [S][len]
    {
        std::vector<char> starts((size_t)cbs, 0);
        // pass 1: where do fades start (needs the per-stream replay, so replay on copies)
        if (!e->procBypassed) {
            for (int s = 0; s < S; ++s) {
                auto f = e->latFade[s];
                const double total = (double)procDelayOf(e, s);
                if (!f.primed || firstCall) {             // prepareToPlay: latency + irLatency, fade gain at 1 (Lifecycle.cpp:377-388)
                    f.latCurrent = f.latTarget = f.oldDelay = (double)std::min(e->B + std::min(std::max(0, (int)e->procParams[s].ir_peak_latency), 2097152), 2097152 + 524288);
                    f.current = f.target = 1.0; f.remaining = 0; f.primed = true;
                }
                for (int t = 0; t < cbs; ++t) {
                    if (std::fabs(f.latTarget - total) >= 2.0 && f.remaining <= 0) {
                        f.oldDelay = f.latCurrent; f.current = 0.0; f.target = 1.0; f.step = 1.0 / (double)xTotal; f.remaining = xTotal;
                        f.latTarget = total;
                        if (t > 0) starts[t] = 1;
                    }
                    if (f.remaining > 0) {
                        f.remaining = std::max(0, f.remaining - e->B);
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
    int cap = 1;
    for (int s = 0; s < S; ++s) {
        auto& f = e->latFade[s];
        const int totalI = procDelayOf(e, s);
        if (e->procBypassed) {
            for (int r = 0; r < R; ++r) dNew[(size_t)r * S + s] = dOld[(size_t)r * S + s] = totalI;
            continue;
        }
        if (!f.primed || firstCall) {
            f.latCurrent = f.latTarget = f.oldDelay = (double)std::min(e->B + std::min(std::max(0, (int)e->procParams[s].ir_peak_latency), 2097152), 2097152 + 524288);
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
                    for (int i = 0; i < e->B; ++i) {                     // getNextValue until the ramp has ended
                        f.current += f.step;
                        if (--f.remaining <= 0) f.current = f.target;
                        vals.push_back(f.current);
                        if (f.remaining <= 0) break;
                    }
                    if (f.remaining <= 0) { f.latCurrent = f.latTarget; f.oldDelay = f.latCurrent; }
                }
            }
            xLen[(size_t)r * S + s] = (int)vals.size();
            cap = std::max(cap, (int)vals.size());
            if (!vals.empty()) {
                if (xg[r].empty()) xg[r].assign((size_t)S * (xTotal + e->B), 0.0);
                std::memcpy(&xg[r][(size_t)s * (xTotal + e->B)], vals.data(), sizeof(double) * vals.size());
            }
        }
    }
    if (cap > 1 && e->latCap < xTotal + e->B) {
        if (e->latGains) (void)hipFree(e->latGains);
        e->latGains = nullptr;
        e->latCap = 0;
        if (hipMalloc((void**)&e->latGains, sizeof(double) * (size_t)S * (xTotal + e->B)) != hipSuccess)
            return fail(e, CPQ_ERR_OOM, "latency cross-fade buffer could not be allocated");
        e->latCap = xTotal + e->B;
    }
    for (int r = 0; r < R; ++r) {
        const int off = ranges[r].c0 * e->B, len = (ranges[r].c1 - ranges[r].c0) * e->B;
        const bool fade = !xg[r].empty();
        const std::vector<int> rn(dNew.begin() + (size_t)r * S, dNew.begin() + (size_t)(r + 1) * S);
        const std::vector<int> ro(dOld.begin() + (size_t)r * S, dOld.begin() + (size_t)(r + 1) * S);
        if (rn != e->latNewHost) {
            CPQ_HIP(e, hipMemcpyAsync(e->latNew, rn.data(), sizeof(int) * S, hipMemcpyHostToDevice, e->stream));
            e->latNewHost = rn;
        }
        if (ro != e->latOldHost) {
            CPQ_HIP(e, hipMemcpyAsync(e->latOld, ro.data(), sizeof(int) * S, hipMemcpyHostToDevice, e->stream));
            e->latOldHost = ro;
        }
        if (fade) {
            CPQ_HIP(e, hipMemcpyAsync(e->latLen, &xLen[(size_t)r * S], sizeof(int) * S, hipMemcpyHostToDevice, e->stream));
            CPQ_HIP(e, hipMemcpyAsync(e->latGains, xg[r].data(), sizeof(double) * xg[r].size(), hipMemcpyHostToDevice, e->stream));
        }
        ProfScope p(e, CPQ_K_MIX);
        cpq::launch_convproc_mix(e->stream, dOut + off, dOut + off, (int64_t)n, e->nCh, len, e->procGains, e->dryRing,
                                 e->dryRingSize, pos0 + off, e->latNew, e->latOld, fade ? e->latLen : nullptr, e->latGains,
                                 e->latCap, skipConv ? 0 : 1, anyRamp ? e->mixRampLen : nullptr, e->mixRampGains, rampStride, off);
        if (fade || R > 1) CPQ_HIP(e, hipStreamSynchronize(e->stream));      // the host vectors are reused / go out of scope
    }
    CPQ_HIP(e, hipGetLastError());
    return CPQ_OK;
}

}  // namespace

extern "C" {

int32_t cpq_convproc_set_params(cpq_engine* e, int32_t stream, const cpq_convproc_params* p)
{
    if (!e || !p) return CPQ_ERR_INVALID_ARG;
    if (stream != CPQ_ALL_STREAMS && (stream < 0 || stream >= e->desc.n_streams))
        return fail(e, CPQ_ERR_INVALID_ARG, "stream %d out of range", stream);
    if (!(p->mix >= 0.0f && p->mix <= 1.0f)) return fail(e, CPQ_ERR_INVALID_ARG, "mix must be in [0, 1]");
    if (p->ir_peak_latency < 0) return fail(e, CPQ_ERR_INVALID_ARG, "ir_peak_latency must be >= 0");
    const bool dryOnly = !((double)p->mix > 0.001);        // needsConvolution, :374
    if (stream != CPQ_ALL_STREAMS && (p->bypassed || dryOnly || e->procBypassed || e->procDryOnly))
        return fail(e, CPQ_ERR_UNSUPPORTED, "bypass / dry-only freeze the convolver state and must be set for CPQ_ALL_STREAMS");
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
    if (stream == CPQ_ALL_STREAMS) { e->procBypassed = p->bypassed != 0; e->procDryOnly = dryOnly; }
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
    return CPQ_OK;
}

int32_t cpq_convproc_process_device(cpq_engine* e, const double* dIn, double* dOut, int32_t nSamples)
{
    int T = 0;
    const int rc = checkCall(e, dIn, dOut, nSamples, &T);
    if (rc != CPQ_OK) return rc;
    CPQ_HIP(e, hipSetDevice(e->device));
    return enqueueConvProc(e, dIn, dOut, T);
}

int32_t cpq_convproc_process(cpq_engine* e, const double* in, double* out, int32_t nSamples)
{
    return viaStaging(e, in, out, nSamples, [e](const double* a, double* b, int T) { return enqueueConvProc(e, a, b, T); });
}

}  // extern "C"

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
    int T = 0;
    const int rc = checkCall(e, dIn, dOut, nSamples, &T);
    if (rc != CPQ_OK) return rc;
    CPQ_HIP(e, hipSetDevice(e->device));
    return enqueueEq(e, dIn, dOut, T);
}

int32_t cpq_eq_process(cpq_engine* e, const double* in, double* out, int32_t nSamples)
{
    return viaStaging(e, in, out, nSamples, [e](const double* a, double* b, int T) { return enqueueEq(e, a, b, T); });
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
    int T = 0;
    const int rc = checkCall(e, dIn, dOut, nSamples, &T);
    if (rc != CPQ_OK) return rc;
    CPQ_HIP(e, hipSetDevice(e->device));
    return enqueueOutFilter(e, dIn, dOut, T);
}

int32_t cpq_outfilter_process(cpq_engine* e, const double* in, double* out, int32_t nSamples)
{
    return viaStaging(e, in, out, nSamples, [e](const double* a, double* b, int T) { return enqueueOutFilter(e, a, b, T); });
}

// ------------------------------------------------------------------------ whole path
static int enqueueBoth(cpq_engine* e, const double* a, double* b, int T)
{
    int rc = CPQ_OK;
    const int n = T * e->P;
    auto conv = [e](const double* x, double* y, int t) {
        return e->convLevel == CPQ_LEVEL_PROCESSOR ? enqueueConvProc(e, x, y, t) : enqueueConv(e, x, y, t);
    };
    if (e->order == CPQ_ORDER_CONV_THEN_EQ) {
        if (!e->convBypassed) rc = conv(a, b, T);
        else if (a != b) cpq::launch_rows_copy(e->stream, a, n, 0, b, n, 0, n, e->nCh);
        if (rc == CPQ_OK) rc = enqueueEq(e, b, b, T);
    } else if (e->convBypassed) {
        rc = enqueueEq(e, a, b, T);
    } else {
        rc = enqueueEq(e, a, e->mid, T);
        if (rc == CPQ_OK && e->anyTrim) {       // scaleBlockFallback(block, convolverInputTrimGain) (:440-447)
            ProfScope p(e, CPQ_K_MIX);
            cpq::launch_rows_scale(e->stream, e->mid, n, n, e->nCh, e->trimDev);
        }
        if (rc == CPQ_OK) rc = conv(e->mid, b, T);
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
            CPQ_HIP(e, hipMemcpyAsync(e->ofFlags + (size_t)2 * s * kBands, flags, sizeof(flags), hipMemcpyHostToDevice, e->stream));
            e->ofPass[s] = pass;
        }
        if (anyActive) rc = enqueueOutFilter(e, b, b, T);
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
    int T = 0;
    const int rc = checkCall(e, dIn, dOut, nSamples, &T);
    if (rc != CPQ_OK) return rc;
    CPQ_HIP(e, hipSetDevice(e->device));
    return enqueueBoth(e, dIn, dOut, T);
}

int32_t cpq_engine_process_block(cpq_engine* e, const double* in, double* out, int32_t nSamples)
{
    return viaStaging(e, in, out, nSamples, [e](const double* a, double* b, int T) { return enqueueBoth(e, a, b, T); });
}

// -------------------------------------------------------------------------- profiling
int32_t cpq_profile_enable(cpq_engine* e, int32_t on)
{
    if (!e) return CPQ_ERR_INVALID_ARG;
    e->profiling = on != 0;
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
