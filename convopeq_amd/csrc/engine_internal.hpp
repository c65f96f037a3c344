// engine_internal.hpp -- the engine object behind the C ABI and the helpers its translation units share:
//   engine_core.cpp  create / destroy / prepare, status and profiling entry points, DSPCore routing (whole path)
//   engine_conv.cpp  kernel-level convolver: set_impulse, FilterSpec tail layers, the per-call kernel sequence
//   engine_proc.cpp  processor-level stage: dry delay ring, mix ramp, latency cross-fade
//   engine_eq.cpp    EQ and output filter: design, device tables, bypass / band-reset state machine
#pragma once

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

struct ProfileSlot {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> freeList;
    int64_t launches = 0;
    double totalMs = 0.0;
};


// One layer of a plan group (engine_native.cpp), run at the reference's OWN partition size: the HC/LC gains (and the
// air-absorption damping) multiply every partition spectrum of a layer at that layer's FFT size, which folds time-aliased
// energy into the frame (src/MKLNonUniformConvolver.cpp:336-443) -- only reproducible with the same partitioning; and the
// reference's Add / Get bookkeeping (input fill, distributed tail MAC, delay-line reader) runs per layer.
struct NativeLayer {
    int P = 0, K = 0, kPad = 0, hRows = 0, ringSlots = 0, nbMax = 0, accCap = 0, outRing = 0;
    int head = 0, histSel = 0, accSel = 0, fill = 0;     // FDL ring head, ping-pong selectors, input fill (inputPos)
    int fftAhead = 0;           // blocks of this call already transformed straight from the call's input (groupsAppend), 0 = none
    double gain = 1.0;          // tail-layer gain applied by the delay-line reader
    int ppc = 1, outputDelay = 0;       // partsPerCallback (:988-994), outputDelaySamples (:1005-1024)
    // host replay of the reference's integer state: layer 0 -- samples written to / read from the output ring
    // (m_ringAvail = wPos - rPos); tail layers -- delayWriteCursor / delayReadCursor, the distributed MAC's progress
    bool distributing = false;
    int nextPart = 0;
    long long wPos = 0, rPos = 0;
    char* mem = nullptr;
    double2 *X = nullptr, *XDN = nullptr, *H = nullptr, *HDN = nullptr, *Y = nullptr, *tw = nullptr, *tw2 = nullptr, *twCol = nullptr, *twSplit = nullptr;
    double *hist[2] = { nullptr, nullptr }, *acc[2] = { nullptr, nullptr }, *ring = nullptr, *gainDev = nullptr;
    double2* scratch = nullptr;   // four-step FFT workspace (P > 4096): [max(nCh * nbMax, K)][P]
};

// Streams that share one layer plan and one phase (loaded while the group was fresh); see engine_native.cpp
struct PlanGroup {
    cpq_nuc_plan plan{};
    bool hasSpec = false, shared = false;       // shared: CPQ_ALL_STREAMS (one stereo IR for every member)
    bool frozen = false;                        // processor-level bypass / dry-only of its (single) member: not processed
    cpq_filter_spec spec{};
    int capCh = 0, usedCh = 0;                  // allocated / launched local channels (2 per pair slot)
    bool identityMap = false;                   // chMap[i] == i for every launched channel (layer 0 may then write the output rows itself)
    std::vector<int> streamOfPair;              // pair slot -> stream, -1 = free
    std::vector<NativeLayer> layers;
    int* chMapDev = nullptr;                    // [capCh] local channel -> row of the call's buffers (-1 = free slot)
    int* irSlotDev = nullptr;                   // [capCh] local channel -> IR slot
    long long* tabDev = nullptr;                // the call's chunk tables
    int tabCap = 0;
    std::vector<long long> tabHost;
    std::vector<size_t> tabOffs;
    std::vector<int> nbOf;
    long long callW0 = 0, callR0 = 0;           // layer 0's write / read positions before the call in flight was replayed
    bool tabOnDevice = false;                   // the call's tables went out with its first launch (kernel arguments)
    bool tailsDone = false;                     // the call's tail layers ran ahead of layer 0, whose transform adds their blocks
    bool getDeferred = false;                   // layer 0's chunk-wise ring read waits for the tails' read-add: one pass does both
    long long samplesSinceReset = 0;
    int lastGot = 0, lastCall = 0;              // Get()'s return value summed over the chunks of the last call
};

// pinned staging for the small host -> device tables of a call (chunk schedules, per-stream gains and flags): a byte ring;
// a region is reused only after the copy that read it has run (one event per upload)
struct PinnedRing {
    struct Pending { size_t begin, end; hipEvent_t ev; };
    char* host = nullptr;
    size_t cap = 0, head = 0;
    std::vector<Pending> pending;       // in issue order
    std::vector<hipEvent_t> freeEvents;
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
    int B = 0;            // caller's block size (the reference's blockSize / callQuantum: layer plan, chunking)
    int P0 = 0;           // the reference's layer-0 partition = getLatency(): nextPow2(max(B, 64))
    int P = 0;            // internal partition size (samples) == complex bins per packed spectrum; multiple of B
    int kCap = 0;         // partition capacity per IR slot (multiple of kMacMaxTile)
    int hRows = 0;        // kCap + prefetch rows allocated per IR slot
    int ringSlots = 0;    // FDL ring slots per channel (power of two)
    int tMax = 0;
    int macTile = 16;
    bool anyCalls = false;  // CPQ_CALLS_ANY: any call quantum, ragged calls; every stream runs in a plan group
    int maxCall = 0;        // samples per call the engine is sized for (max_blocks_per_call * block_size)

    // device arena
    char* arena = nullptr;
    int64_t arenaBytes = 0;
    double2* X = nullptr;       // [nCh][ringSlots][e->P]       FDL ring of packed spectra
    double2* XDN = nullptr;     // [nCh][ringSlots]           (DC, Nyquist) of every FDL slot
    double2* H = nullptr;       // [nCh][hRows][e->P]           IR partition spectra per IR slot
    double2* HDN = nullptr;     // [nCh][hRows]
    double2* Y = nullptr;       // [nCh][tMax][e->P]            accumulated output spectra of the call
    double* hist[2] = { nullptr, nullptr };   // [nCh][e->P]    overlap history, ping-pong
    double* stageIn = nullptr;  // [nCh][tMax*e->P]             staging for the host-pointer entry points (allocated on first use)
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
    void* svfChain = nullptr;   // scheduling words of the time-parallel cascade: header, arrival counters of the CUs, and (chained spans
                                // only: svfChainSpans > 0) the [channels][svfChainSpans][20][4] hand-over granules
    int svfChainSpans = 0;
    int svfChainGrid = 0;       // workgroups of the span kernel the device holds at once (2 per CU)
    unsigned long long uploadSeq = 0;   // staged uploads so far (they are ordered on the engine's stream only)
    bool ofSet = false, ofTpSafe = true, ofInPath = false;

    // run-time state
    int head = 0;               // ring slot of the next block
    int histSel = 0;
    int kActive = 0;            // max partitions over the loaded IRs (multiple of kMacMaxTile)
    int kMaxReal = 0;           // max real partition count (DC/Nyquist loop bound)
    std::vector<int> irSlotHost;
    bool irPrivate = true;      // every channel reads its own IR rows (no shared slot)
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
    std::vector<char> agcResetPending;  // requestAgcReset: envelopes and gain back to their initial values at the next processed block
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

    // plan groups (engine_native.cpp): streams run on the reference's own layer plan with its Add / Get bookkeeping
    std::vector<PlanGroup*> groups;
    std::vector<int> groupOf;           // per stream: index into groups, -1 = main (uniform) path
    bool mainActive = false;            // some loaded stream runs on the main path
    int lastCallSamples = 0;
    PinnedRing pinned;                  // small per-call host -> device tables

    // processor-level wrapper (N1)
    int convLevel = CPQ_LEVEL_NUC;
    std::vector<cpq_convproc_params> procParams;   // per stream
    std::vector<char> procBypass, procDryOnly;      // per stream: bypassed / mix <= 0.001 (the convolver is not called)
    std::vector<int> procWetOnHost;                 // per stream: what the device flags hold
    int* procWetOn = nullptr;                       // [streams] device: 0 = the stream's output is the delayed dry signal
    bool honourFrozen = false;                      // set around the convolver call of enqueueConvProc: frozen plan groups (and their direct heads) rest
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

namespace cpqi {

int fail(cpq_engine* e, int code, const char* fmt, ...);

#define CPQ_HIP(e, call)                                                                             \
    do {                                                                                             \
        hipError_t err__ = (call);                                                                   \
        if (err__ != hipSuccess)                                                                     \
            return fail((e), CPQ_ERR_DEVICE, "%s failed: %s", #call, hipGetErrorString(err__));      \
    } while (0)


int nextPow2(int v);
int64_t alignUp(int64_t v, int64_t a);
cpq::FftTables tables(const cpq_engine* e);

// the engine's launches go to another stream while one of these is alive (declare it BEFORE the ProfScope it should cover)
struct StreamOverride {
    cpq_engine* e; hipStream_t saved;
    StreamOverride(cpq_engine* eng, hipStream_t s) : e(eng), saved(eng->stream) { e->stream = s; }
    ~StreamOverride() { e->stream = saved; }
};

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


int checkCall(cpq_engine* e, const void* in, const void* out, int nSamples);
int ensureCallBuffer(cpq_engine* e, double** buf, const char* what);
int zeroRuntimeState(cpq_engine* e, bool conv, bool eq);

// engine_native.cpp
int stageUpload(cpq_engine* e, void* dst, const void* src, size_t bytes);
void freePinnedRing(cpq_engine* e);
void freeGroups(cpq_engine* e);
int resetGroups(cpq_engine* e);
int nativeSetImpulse(cpq_engine* e, int stream, const double* irL, const double* irR, int irLen, double scale, int headTaps,
                     const cpq_filter_spec* spec, const cpq_nuc_plan& pl);
int leaveNativeGroup(cpq_engine* e, int stream);
int setStreamFrozen(cpq_engine* e, int stream, bool frozen);
void clearFrozen(cpq_engine* e);           // no plan group rests any more (conv level change, reset, prepare)
int groupsAppend(cpq_engine* e, const double* dIn, int n);
int groupsRunLayer0(cpq_engine* e, double* dOut, int n);
int groupsRunTails(cpq_engine* e, double* dOut, int n);
// engine_conv.cpp
int enqueueConv(cpq_engine* e, const double* dIn, double* dOut, int n);
// engine_proc.cpp
int uploadProcParams(cpq_engine* e);
int enqueueConvProc(cpq_engine* e, const double* dIn, double* dOut, int n);
// engine_eq.cpp
void syncEqBypass(cpq_engine* e);
int enqueueEq(cpq_engine* e, const double* dIn, double* dOut, int n);
int enqueueOutFilter(cpq_engine* e, const double* dIn, double* dOut, int n);

// Host-pointer entry points: H2D, the kernel sequence and D2H.  Long calls are cut into four time chunks (each a complete
// engine call: the state carries over on the engine's stream) so that the upload of chunk i+1 and the download of chunk
// i-1 run on two copy streams beside the kernels of chunk i.  With pinned caller buffers (cpq_host_register) the three
// overlap; pageable buffers take the plain upload / kernels / download sequence.
template <typename F>
int viaStaging(cpq_engine* e, const double* in, double* out, int nSamples, F&& body)
{
    int rc = checkCall(e, in, out, nSamples);
    if (rc != CPQ_OK) return rc;
    CPQ_HIP(e, hipSetDevice(e->device));
    rc = ensureCallBuffer(e, &e->stageIn, "upload staging");
    if (rc == CPQ_OK) rc = ensureCallBuffer(e, &e->stageOut, "download staging");
    if (rc != CPQ_OK) return rc;
    constexpr int kChunks = 4;
    const int T = nSamples / e->P;                                      // partitions in the call (whole ones outside CPQ_CALLS_ANY)
    auto pinned = [](const void* p) {
        hipPointerAttribute_t a{};
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }   // plain malloc'd memory
        return a.type == hipMemoryTypeHost;
    };
    // pageable buffers: the runtime stages every copy and blocks the host, so chunking only adds strided copies
    // (measured 1157 vs 1230 M samples/s); one upload, one download
    if (e->anyCalls || T < 32 || T % kChunks != 0 || !pinned(in) || !pinned(out)) {
        const size_t bytes = (size_t)e->nCh * nSamples * sizeof(double);
        CPQ_HIP(e, hipMemcpyAsync(e->stageIn, in, bytes, hipMemcpyHostToDevice, e->stream));
        rc = body(e->stageIn, e->stageOut, nSamples);
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
        rc = body(dIn, dOut, (int)chunkLen);
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


}  // namespace cpqi
