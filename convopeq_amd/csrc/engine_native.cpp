// engine_native.cpp -- plan groups: streams whose convolver runs on the reference's OWN layer plan (every layer at its own
// partition size, src/MKLNonUniformConvolver.cpp:738-758) with the reference's Add / Get bookkeeping replayed per chunk.
//
// Used for (engine_conv.cpp decides per set_impulse): CPQ_CALLS_ANY engines (any call quantum, ragged calls:
// inputPos accumulation :1431-1446, output ring with zero-fill :1376-1402), CPQ_SCHED_REFERENCE_NUC, and FilterSpec plans with
// tail layers (the spectral gains are partition-size dependent, :336-443).  A group = the streams that share one layer
// plan AND one phase (they were loaded while the group was fresh): the integer state of the reference's Add / Get --
// input fill of every layer, the distributed tail MAC's progress (:1497-1545), delay-line cursors (:1653-1688), ring
// read / write counts -- is then the same for all of them and is replayed ONCE on the host per call; the GPU moves the
// data (per layer: input accumulator -> FFT -> FDL MAC -> IFFT -> output ring / delay line -> chunk-wise read).  Streams
// with different IR lengths, FilterSpecs or load times live in different groups, as one StereoConvolver per stream does
// in the reference (src/ConvolverProcessor.h:741-814).
#include "engine_internal.hpp"

using namespace cpqi;

namespace cpqi {

namespace {

struct BufItem { void** ptr; int64_t rowBytes; bool perSlot; };    // perSlot: [hSlots] rows instead of [capCh]

// the device buffers of one layer, as (pointer, bytes per channel row) so that growing a group copies row prefixes
std::vector<BufItem> layerItems(NativeLayer& t)
{
    return {
        { (void**)&t.X, (int64_t)t.ringSlots * t.P * (int64_t)sizeof(double2), false },
        { (void**)&t.XDN, (int64_t)t.ringSlots * (int64_t)sizeof(double2), false },
        { (void**)&t.H, (int64_t)t.hRows * t.P * (int64_t)sizeof(double2), true },
        { (void**)&t.HDN, (int64_t)t.hRows * (int64_t)sizeof(double2), true },
        { (void**)&t.Y, (int64_t)t.nbMax * t.P * (int64_t)sizeof(double2), false },
        { (void**)&t.hist[0], (int64_t)t.P * (int64_t)sizeof(double), false },
        { (void**)&t.hist[1], (int64_t)t.P * (int64_t)sizeof(double), false },
        { (void**)&t.acc[0], (int64_t)t.accCap * (int64_t)sizeof(double), false },
        { (void**)&t.acc[1], (int64_t)t.accCap * (int64_t)sizeof(double), false },
        { (void**)&t.ring, (int64_t)t.outRing * (int64_t)sizeof(double), false },
    };
}

int allocLayer(cpq_engine* e, PlanGroup& g, NativeLayer& t, int capCh, int hSlots)
{
    auto items = layerItems(t);
    int64_t total = 0;
    for (const BufItem& it : items) total += alignUp(it.rowBytes * (it.perSlot ? hSlots : capCh), 256);
    const int64_t twBytes = alignUp(t.P * (int64_t)sizeof(double2), 256);
    const int64_t gainBytes = alignUp((t.P + 1) * (int64_t)sizeof(double), 256);
    const int64_t scratchBytes = t.P > 4096 ? alignUp(std::max<int64_t>((int64_t)capCh * t.nbMax, t.K) * t.P * (int64_t)sizeof(double2), 256) : 256;
    const bool big = t.P > 4096;          // four-step transforms: + the two reordered tables (kernels.hpp: FftTables)
    total += (big ? 4 : 2) * twBytes + gainBytes + scratchBytes;
    char* mem = nullptr;
    if (hipMalloc((void**)&mem, (size_t)total) != hipSuccess) {
        (void)hipGetLastError();
        return fail(e, CPQ_ERR_OOM, "plan group layer (partition %d, %d channels): %lld bytes could not be allocated", t.P, capCh,
                    (long long)total);
    }
    CPQ_HIP(e, hipMemsetAsync(mem, 0, (size_t)total, e->stream));
    int64_t off = 0;
    for (const BufItem& it : items) { *it.ptr = mem + off; off += alignUp(it.rowBytes * (it.perSlot ? hSlots : capCh), 256); }
    t.tw = (double2*)(mem + off); off += twBytes;
    t.tw2 = (double2*)(mem + off); off += twBytes;
    t.twCol = t.twSplit = nullptr;
    if (big) { t.twCol = (double2*)(mem + off); off += twBytes; t.twSplit = (double2*)(mem + off); off += twBytes; }
    t.gainDev = (double*)(mem + off); off += gainBytes;
    t.scratch = (double2*)(mem + off);
    t.mem = mem;
    // twiddles in extended precision, rounded once
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
    if (big) {
        std::vector<double2> wc(t.P), ws(t.P);
        cpq::fill_big_twiddles(w.data(), w2.data(), t.P, wc.data(), ws.data());
        CPQ_HIP(e, hipMemcpy(t.twCol, wc.data(), t.P * sizeof(double2), hipMemcpyHostToDevice));
        CPQ_HIP(e, hipMemcpy(t.twSplit, ws.data(), t.P * sizeof(double2), hipMemcpyHostToDevice));
    }
    (void)g;
    return CPQ_OK;
}

void layerGeometry(const cpq_engine* e, const cpq_nuc_plan& pl, int l, NativeLayer& t)
{
    const int nMax = e->maxCall;
    t.P = pl.part_size[l];
    t.K = pl.num_parts_ir[l];
    t.kPad = (int)alignUp(t.K, cpq::kMacMaxTile);                  // a multiple of every tile the MAC launcher may pick
    t.hRows = t.kPad + 16;                                         // zero rows for the kernels' 4-row read-ahead
    t.nbMax = (t.P - 1 + nMax) / t.P;
    t.ringSlots = nextPow2(t.kPad + cpq::kMacMaxTile + t.nbMax);
    t.accCap = t.P + nMax;
    t.gain = pl.gain[l];
    t.ppc = std::max(1, (int)pl.parts_per_callback[l]);
    t.outputDelay = pl.output_delay[l];
    // layer 0: the output ring holds what Get() has not read yet (< P0 + one chunk) plus one call of new blocks;
    // tail layers: the reader is at most outputDelay behind the writer, blocks are stored when their partition fills
    t.outRing = (l == 0) ? nextPow2(2 * t.P + nMax + t.P) : nextPow2(pl.output_delay[l] + 3 * t.P + nMax + e->B);
}

void freeGroupBuffers(PlanGroup& g)
{
    for (NativeLayer& t : g.layers) if (t.mem) (void)hipFree(t.mem);
    g.layers.clear();
    if (g.chMapDev) (void)hipFree(g.chMapDev);
    if (g.irSlotDev) (void)hipFree(g.irSlotDev);
    if (g.tabDev) (void)hipFree(g.tabDev);
    g.chMapDev = g.irSlotDev = nullptr;
    g.tabDev = nullptr;
}

void resetGroupHost(PlanGroup& g)
{
    for (NativeLayer& t : g.layers) {
        t.head = t.histSel = t.accSel = t.fill = 0;
        t.distributing = false;
        t.nextPart = 0;
        t.wPos = t.rPos = 0;
    }
    g.samplesSinceReset = 0;
}

int uploadGroupMaps(cpq_engine* e, PlanGroup& g)
{
    std::vector<int> chMap((size_t)g.capCh, -1), irSlot((size_t)g.capCh, 0);
    g.usedCh = 0;
    for (int p = 0; p < g.capCh / 2; ++p) {
        const int s = g.streamOfPair[p];
        for (int ch = 0; ch < 2; ++ch) {
            chMap[2 * p + ch] = s >= 0 ? 2 * s + ch : -1;
            irSlot[2 * p + ch] = g.shared ? ch : 2 * p + ch;
        }
        if (s >= 0) g.usedCh = 2 * p + 2;
    }
    g.identityMap = true;                      // local channel i = row i of the call's buffers, no free slot in between
    for (int i = 0; i < g.usedCh; ++i) g.identityMap = g.identityMap && chMap[(size_t)i] == i;
    CPQ_HIP(e, hipStreamSynchronize(e->stream));
    CPQ_HIP(e, hipMemcpy(g.chMapDev, chMap.data(), sizeof(int) * chMap.size(), hipMemcpyHostToDevice));
    CPQ_HIP(e, hipMemcpy(g.irSlotDev, irSlot.data(), sizeof(int) * irSlot.size(), hipMemcpyHostToDevice));
    return CPQ_OK;
}

int tabEntries(const cpq_engine* e, const PlanGroup& g)
{
    const int chunks = (e->maxCall + e->B - 1) / e->B + 1;
    int n = 2 * chunks;                                             // layer 0: pos, cnt per chunk
    for (size_t l = 1; l < g.layers.size(); ++l) n += chunks + g.layers[l].nbMax + 1;     // sched per chunk, put position per block
    return n;
}

// a group with room for capPairs streams: allocates every layer (the old buffers' row prefixes are kept when growing).
// The new layers are built aside and swapped in on success: on any failure the group is as it was.
int sizeGroup(cpq_engine* e, PlanGroup& g, int capPairs)
{
    const int newCapCh = 2 * capPairs;
    const int hSlots = g.shared ? 2 : newCapCh;
    const std::vector<NativeLayer>& old = g.layers;
    const int oldCapCh = g.capCh, oldHSlots = g.shared ? 2 : oldCapCh;
    std::vector<NativeLayer> fresh;
    int* chMapNew = nullptr;
    int* irSlotNew = nullptr;
    long long* tabNew = nullptr;
    auto undo = [&](int rc) {
        (void)hipStreamSynchronize(e->stream);      // row copies into the fresh buffers may still be in flight
        for (NativeLayer& t : fresh) if (t.mem) (void)hipFree(t.mem);
        if (chMapNew) (void)hipFree(chMapNew);
        if (irSlotNew) (void)hipFree(irSlotNew);
        if (tabNew) (void)hipFree(tabNew);
        (void)hipGetLastError();
        return rc;
    };
    for (int l = 0; l < g.plan.num_layers; ++l) {
        NativeLayer t;
        if (!old.empty()) { t = old[(size_t)l]; t.mem = nullptr; }
        layerGeometry(e, g.plan, l, t);
        const int rc = allocLayer(e, g, t, newCapCh, hSlots);
        if (rc != CPQ_OK) return undo(rc);
        fresh.push_back(t);
        if (!old.empty()) {
            const NativeLayer& o = old[(size_t)l];
            NativeLayer oc = o;
            auto src = layerItems(oc), dst = layerItems(fresh.back());
            for (size_t i = 0; i < src.size(); ++i) {
                const int64_t rows = src[i].perSlot ? oldHSlots : oldCapCh;
                if (hipMemcpyAsync(*dst[i].ptr, *src[i].ptr, (size_t)(rows * src[i].rowBytes), hipMemcpyDeviceToDevice, e->stream) != hipSuccess)
                    return undo(fail(e, CPQ_ERR_DEVICE, "plan group rows could not be copied"));
            }
        }
    }
    if (hipStreamSynchronize(e->stream) != hipSuccess) return undo(fail(e, CPQ_ERR_DEVICE, "plan group resize: stream error"));
    PlanGroup probe = g;                      // table size of the new geometry
    probe.layers = fresh;
    const int tabCap = tabEntries(e, probe);
    if (hipMalloc((void**)&chMapNew, sizeof(int) * newCapCh) != hipSuccess || hipMalloc((void**)&irSlotNew, sizeof(int) * newCapCh) != hipSuccess ||
        hipMalloc((void**)&tabNew, sizeof(long long) * (size_t)tabCap) != hipSuccess)
        return undo(fail(e, CPQ_ERR_OOM, "plan group tables could not be allocated"));
    // success: swap
    for (NativeLayer& o : g.layers) if (o.mem) (void)hipFree(o.mem);
    if (g.chMapDev) (void)hipFree(g.chMapDev);
    if (g.irSlotDev) (void)hipFree(g.irSlotDev);
    if (g.tabDev) (void)hipFree(g.tabDev);
    g.layers = fresh;
    g.chMapDev = chMapNew;
    g.irSlotDev = irSlotNew;
    g.tabDev = tabNew;
    g.capCh = newCapCh;
    g.streamOfPair.resize((size_t)capPairs, -1);
    g.tabCap = tabCap;
    return uploadGroupMaps(e, g);
}

// zero the run-time rows of one stereo pair (a stream that joins, or leaves, a group starts from silence like a new NUC)
int zeroPairState(cpq_engine* e, PlanGroup& g, int pair)
{
    for (NativeLayer& t : g.layers) {
        auto items = layerItems(t);
        for (const BufItem& it : items) {
            if (it.perSlot) continue;
            CPQ_HIP(e, hipMemsetAsync((char*)*it.ptr + (int64_t)2 * pair * it.rowBytes, 0, (size_t)(2 * it.rowBytes), e->stream));
        }
    }
    return CPQ_OK;
}

}  // namespace

// Small host -> device uploads on the processing path (chunk schedules, per-stream gains / flags / fade tables): staged
// through a ring of pinned memory so that the copy is truly stream-ordered, never blocks the host, and never reads host
// storage that has gone out of scope (a pageable hipMemcpyAsync is only safe because the runtime happens to stage it
// synchronously).  Uploads larger than a quarter of the ring take the plain copy and wait for it.
int stageUpload(cpq_engine* e, void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return CPQ_OK;
    ++e->uploadSeq;
    PinnedRing& r = e->pinned;
    if (!r.host) {
        r.cap = (size_t)8 << 20;
        if (hipHostMalloc((void**)&r.host, r.cap) != hipSuccess) {
            (void)hipGetLastError();
            r.host = nullptr;
            return fail(e, CPQ_ERR_OOM, "pinned staging ring could not be allocated");
        }
    }
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (need > r.cap / 4) {
        CPQ_HIP(e, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream));
        CPQ_HIP(e, hipStreamSynchronize(e->stream));
        return CPQ_OK;
    }
    if (r.head + need > r.cap) r.head = 0;
    const size_t begin = r.head, end = r.head + need;
    // uploads still in flight that own bytes of [begin, end) must have run; they retire in issue order
    size_t done = 0;
    for (; done < r.pending.size(); ++done) {
        const PinnedRing::Pending& p = r.pending[done];
        bool overlapsLater = false;
        for (size_t k = done; k < r.pending.size() && !overlapsLater; ++k)
            overlapsLater = r.pending[k].begin < end && begin < r.pending[k].end;
        if (!overlapsLater) break;
        CPQ_HIP(e, hipEventSynchronize(p.ev));
        r.freeEvents.push_back(p.ev);
    }
    if (done) r.pending.erase(r.pending.begin(), r.pending.begin() + (long)done);
    // retire whatever has completed anyway (keeps the list short)
    while (!r.pending.empty() && hipEventQuery(r.pending.front().ev) == hipSuccess) {
        r.freeEvents.push_back(r.pending.front().ev);
        r.pending.erase(r.pending.begin());
    }
    (void)hipGetLastError();            // hipEventQuery reports "not ready" as an error code
    std::memcpy(r.host + begin, src, bytes);
    CPQ_HIP(e, hipMemcpyAsync(dst, r.host + begin, bytes, hipMemcpyHostToDevice, e->stream));
    hipEvent_t ev;
    if (!r.freeEvents.empty()) { ev = r.freeEvents.back(); r.freeEvents.pop_back(); }
    else CPQ_HIP(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    CPQ_HIP(e, hipEventRecord(ev, e->stream));
    r.pending.push_back(PinnedRing::Pending{ begin, end, ev });
    r.head = end;
    return CPQ_OK;
}

void freePinnedRing(cpq_engine* e)
{
    PinnedRing& r = e->pinned;
    if (!r.host) return;
    for (auto& p : r.pending) (void)hipEventDestroy(p.ev);
    for (auto& ev : r.freeEvents) (void)hipEventDestroy(ev);
    r.pending.clear();
    r.freeEvents.clear();
    (void)hipHostFree(r.host);
    r.host = nullptr;
}

void freeGroups(cpq_engine* e)
{
    for (PlanGroup* g : e->groups) { freeGroupBuffers(*g); delete g; }
    e->groups.clear();
    std::fill(e->groupOf.begin(), e->groupOf.end(), -1);
}

int resetGroups(cpq_engine* e)
{
    for (PlanGroup* gp : e->groups) {
        PlanGroup& g = *gp;
        for (NativeLayer& t : g.layers) {
            auto items = layerItems(t);
            for (const BufItem& it : items)
                if (!it.perSlot) CPQ_HIP(e, hipMemsetAsync(*it.ptr, 0, (size_t)(it.rowBytes * g.capCh), e->stream));
        }
        resetGroupHost(g);
        g.frozen = false;       // resting is processor-level state: the next processor-level call re-establishes it (enqueueConvProc)
    }
    return CPQ_OK;
}

int leaveNativeGroup(cpq_engine* e, int stream)
{
    const int gi = e->groupOf[(size_t)stream];
    if (gi < 0) return CPQ_OK;
    PlanGroup& g = *e->groups[(size_t)gi];
    for (size_t p = 0; p < g.streamOfPair.size(); ++p)
        if (g.streamOfPair[p] == stream) {
            g.streamOfPair[p] = -1;
            const int rc = zeroPairState(e, g, (int)p);
            if (rc != CPQ_OK) return rc;
        }
    e->groupOf[(size_t)stream] = -1;
    bool empty = true;
    for (int s : g.streamOfPair) empty = empty && s < 0;
    if (empty) {
        CPQ_HIP(e, hipStreamSynchronize(e->stream));
        freeGroupBuffers(g);
        delete e->groups[(size_t)gi];
        e->groups.erase(e->groups.begin() + gi);
        for (int& go : e->groupOf) if (go > gi) --go;
        return CPQ_OK;
    }
    return uploadGroupMaps(e, g);
}

// The stream gets a plan group of its own with its state and phase as they are (rows copied, host replay state copied):
// what a processor-level bypass needs, which stops ONE stream's Add / Get while the others go on (the reference's
// ConvolverProcessor does not call its convolver while bypassed or dry-only, Runtime.cpp:123-186, 573-585; on release the
// NUC resumes with the history it had).
static int isolateStream(cpq_engine* e, int stream)
{
    const int gi = e->groupOf[(size_t)stream];
    if (gi < 0) return CPQ_ERR_INVALID_ARG;
    PlanGroup& g = *e->groups[(size_t)gi];
    int members = 0, pair = -1;
    for (size_t p = 0; p < g.streamOfPair.size(); ++p) {
        if (g.streamOfPair[p] >= 0) ++members;
        if (g.streamOfPair[p] == stream) pair = (int)p;
    }
    if (members <= 1 || pair < 0) return CPQ_OK;
    PlanGroup* n = new (std::nothrow) PlanGroup();
    if (!n) return fail(e, CPQ_ERR_OOM, "host allocation failed");
    n->plan = g.plan;
    n->hasSpec = g.hasSpec;
    n->spec = g.spec;
    n->shared = g.shared;
    n->samplesSinceReset = std::max<long long>(g.samplesSinceReset, 1);        // not fresh: nobody else may join its phase
    n->lastGot = g.lastGot;
    n->lastCall = g.lastCall;
    int rc = sizeGroup(e, *n, 1);
    if (rc != CPQ_OK) { freeGroupBuffers(*n); delete n; return rc; }
    for (size_t l = 0; l < g.layers.size(); ++l) {
        NativeLayer& src = g.layers[l];
        NativeLayer& dst = n->layers[l];
        auto si = layerItems(src), di = layerItems(dst);
        for (size_t i = 0; i < si.size(); ++i) {
            const int64_t rows = 2, from = si[i].perSlot ? (g.shared ? 0 : 2 * pair) : 2 * pair;
            if (hipMemcpyAsync(*di[i].ptr, (char*)*si[i].ptr + from * si[i].rowBytes, (size_t)(rows * si[i].rowBytes),
                               hipMemcpyDeviceToDevice, e->stream) != hipSuccess) {
                (void)hipStreamSynchronize(e->stream);
                freeGroupBuffers(*n);
                delete n;
                return fail(e, CPQ_ERR_DEVICE, "plan group rows could not be copied");
            }
        }
        dst.head = src.head; dst.histSel = src.histSel; dst.accSel = src.accSel; dst.fill = src.fill;
        dst.distributing = src.distributing; dst.nextPart = src.nextPart; dst.wPos = src.wPos; dst.rPos = src.rPos;
    }
    n->streamOfPair[0] = stream;
    // the stream is the new group's from here on, whatever happens below: the engine stays consistent
    g.streamOfPair[(size_t)pair] = -1;
    e->groups.push_back(n);
    e->groupOf[(size_t)stream] = (int)e->groups.size() - 1;
    rc = zeroPairState(e, g, pair);
    if (rc != CPQ_OK) return rc;
    rc = uploadGroupMaps(e, g);
    if (rc == CPQ_OK) rc = uploadGroupMaps(e, *n);
    return rc;
}

// frozen: the stream's convolver is not called (bypass / dry-only at the processor level); its state waits as it is
int setStreamFrozen(cpq_engine* e, int stream, bool frozen)
{
    int gi = e->groupOf[(size_t)stream];
    if (gi < 0) return fail(e, CPQ_ERR_UNSUPPORTED, "a per-stream bypass needs the stream on the reference's own layer plan "
                                                   "(CPQ_CALLS_ANY, CPQ_SCHED_REFERENCE_NUC or a FilterSpec plan with tail layers); "
                                                   "on the uniform path set it for CPQ_ALL_STREAMS");
    if (e->groups[(size_t)gi]->frozen == frozen) return CPQ_OK;      // (a frozen group has one member: it was isolated first)
    const int rc = isolateStream(e, stream);
    if (rc != CPQ_OK) return rc;
    gi = e->groupOf[(size_t)stream];
    e->groups[(size_t)gi]->frozen = frozen;
    return CPQ_OK;
}

void clearFrozen(cpq_engine* e)
{
    for (PlanGroup* g : e->groups) g->frozen = false;
}

// SetImpulse of one stream (or of all streams with one shared stereo IR) on the reference's own layer plan
int nativeSetImpulse(cpq_engine* e, int stream, const double* irL, const double* irR, int irLen, double scale, int headTaps,
                     const cpq_filter_spec* spec, const cpq_nuc_plan& pl)
{
    if (pl.part_size[0] > 4096)
        return fail(e, CPQ_ERR_UNSUPPORTED, "layer-0 partition %d > 4096", pl.part_size[0]);
    for (int l = 1; l < pl.num_layers; ++l)
        if (pl.part_size[l] > 131072 || (pl.part_size[l] & (pl.part_size[l] - 1)))
            return fail(e, CPQ_ERR_UNSUPPORTED, "tail layer %d has partition size %d; supported: powers of two up to 131072", l, pl.part_size[l]);
    const int S = e->desc.n_streams;
    const bool shared = stream == CPQ_ALL_STREAMS;
    const int s0 = shared ? 0 : stream, s1 = shared ? S : stream + 1;
    for (int s = s0; s < s1; ++s) { const int rc = leaveNativeGroup(e, s); if (rc != CPQ_OK) return rc; }

    // a fresh group with the same plan takes the stream(s); anything else would put them on another group's phase
    PlanGroup* g = nullptr;
    int gi = -1;
    if (!shared)
        for (size_t i = 0; i < e->groups.size() && !g; ++i) {
            PlanGroup& c = *e->groups[i];
            const bool sameSpec = c.hasSpec == (spec != nullptr) && (!spec || std::memcmp(&c.spec, spec, sizeof(*spec)) == 0);
            if (!c.shared && !c.frozen && c.samplesSinceReset == 0 && sameSpec && std::memcmp(&c.plan, &pl, sizeof(pl)) == 0) { g = &c; gi = (int)i; }     // (a resting group holds one isolated stream: nobody joins it)
        }
    if (!g) {
        g = new (std::nothrow) PlanGroup();
        if (!g) return fail(e, CPQ_ERR_OOM, "host allocation failed");
        g->plan = pl;
        g->hasSpec = spec != nullptr;
        if (spec) g->spec = *spec;
        g->shared = shared;
        const int rc = sizeGroup(e, *g, shared ? S : 1);       // one pair, doubled as members join: 256 one-member groups (256 IR lengths) cost what they use
        if (rc != CPQ_OK) { freeGroupBuffers(*g); delete g; return rc; }
        e->groups.push_back(g);
        gi = (int)e->groups.size() - 1;
    }
    // pair slots
    std::vector<int> pairOf;
    for (int s = s0; s < s1; ++s) {
        int pair = -1;
        for (size_t p = 0; p < g->streamOfPair.size() && pair < 0; ++p) if (g->streamOfPair[p] < 0) pair = (int)p;
        if (pair < 0) {
            pair = (int)g->streamOfPair.size();
            const int rc = sizeGroup(e, *g, std::min(S, std::max(2 * pair, 2)));
            if (rc != CPQ_OK) return rc;
        }
        g->streamOfPair[(size_t)pair] = s;
        e->groupOf[(size_t)s] = gi;
        pairOf.push_back(pair);
    }
    { const int rc = uploadGroupMaps(e, *g); if (rc != CPQ_OK) return rc; }

    // partition spectra of every layer (:919-946), the direct head's taps left out of the FFT path (:730-731), scaled
    // (:939-940), FilterSpec gains at the layer's own FFT size (:336-443), air absorption on the tail layers (:1060-1097)
    const double* irs[2] = { irL, irR };
    const bool scaled = std::abs(scale - 1.0) > 1e-12;
    std::vector<double> seg, gains;
    const int hPair = shared ? 0 : pairOf[0];
    for (int ch = 0; ch < 2; ++ch) {
        const int slot = shared ? ch : 2 * hPair + ch;
        for (int l = 0; l < pl.num_layers; ++l) {
            NativeLayer& t = g->layers[(size_t)l];
            seg.assign(irs[ch] + pl.offset[l], irs[ch] + pl.offset[l] + pl.len[l]);
            if (l == 0) for (int i = 0; i < headTaps && i < (int)seg.size(); ++i) seg[(size_t)i] = 0.0;
            if (scaled) for (double& v : seg) v *= scale;
            if ((int64_t)seg.size() > e->heffCap) return fail(e, CPQ_ERR_INVALID_ARG, "layer %d has %zu taps, staging capacity %lld", l, seg.size(), (long long)e->heffCap);
            double2* Ht = t.H + (int64_t)slot * t.hRows * t.P;
            double2* HDNt = t.HDN + (int64_t)slot * t.hRows;
            CPQ_HIP(e, hipMemsetAsync(Ht, 0, (size_t)t.hRows * t.P * sizeof(double2), e->stream));
            CPQ_HIP(e, hipMemsetAsync(HDNt, 0, (size_t)t.hRows * sizeof(double2), e->stream));
            CPQ_HIP(e, hipMemcpyAsync(e->heffDev, seg.data(), seg.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
            cpq::launch_ir_spectra(e->stream, e->heffDev, (int)seg.size(), Ht, HDNt, cpq::FftTables{ t.tw, t.tw2, t.twCol, t.twSplit }, t.P, t.K, t.scratch);
            if (spec) {
                cpq::spectrumFilterGains(*spec, 2 * t.P, gains);
                CPQ_HIP(e, hipMemcpyAsync(t.gainDev, gains.data(), gains.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
                cpq::launch_spectrum_gain(e->stream, Ht, HDNt, t.gainDev, t.P, t.K);
                CPQ_HIP(e, hipStreamSynchronize(e->stream));
                if (l > 0 && cpq::airAbsorptionGains(*spec, l, t.P + 1, gains)) {
                    CPQ_HIP(e, hipMemcpyAsync(t.gainDev, gains.data(), gains.size() * sizeof(double), hipMemcpyHostToDevice, e->stream));
                    cpq::launch_spectrum_gain(e->stream, Ht, HDNt, t.gainDev, t.P, t.K);
                }
            }
            CPQ_HIP(e, hipGetLastError());
            CPQ_HIP(e, hipStreamSynchronize(e->stream));        // heffDev / gainDev / seg are reused
        }
    }
    return CPQ_OK;
}

// ---------------------------------------------------------------------------------------------------- per call
// Replays Add() / Get() of the reference for the chunks of one call (integers only) and fills the call's tables:
// layer 0 read position and count per chunk; per tail layer the delay-line read position per chunk (-1 = skip) and the
// delay-line position of every block whose partition fills in this call.
static void replayCall(const cpq_engine* e, PlanGroup& g, int n, std::vector<long long>& tab, std::vector<size_t>& offs,
                       std::vector<int>& nbOf)
{
    const int q = e->B;
    const int chunks = (n + q - 1) / q;
    const size_t L = g.layers.size();
    offs.assign(2 * L + 1, 0);
    nbOf.assign(L, 0);
    size_t at = 0;
    for (size_t l = 0; l < L; ++l) {
        offs[2 * l] = at; at += (size_t)chunks;                     // layer 0: pos / tails: sched
        offs[2 * l + 1] = at;                                       // layer 0: cnt   / tails: put positions
        at += (l == 0) ? (size_t)chunks : (size_t)((g.layers[l].fill + n) / g.layers[l].P);
    }
    offs[2 * L] = at;
    tab.assign(at, -1);
    std::vector<int> fillOf(L), pendingIdx(L, -1);      // pendingIdx: this call's block whose MAC is still being distributed
    for (size_t l = 0; l < L; ++l) fillOf[l] = g.layers[l].fill;
    for (int c = 0; c < chunks; ++c) {
        const int len = std::min(q, n - c * q);
        for (size_t l = 0; l < L; ++l) {
            NativeLayer& t = g.layers[l];
            // Add(): input accumulation; a partition that fills is transformed at once (:1448-1494)
            fillOf[l] += len;
            while (fillOf[l] >= t.P) {
                fillOf[l] -= t.P;
                if (l == 0) t.wPos += t.P;                          // processLayerBlock -> ringWrite
                else {
                    // a block still distributing is abandoned (:1487-1490): it never reaches the delay line (one of an
                    // earlier call already sits at wPos and is overwritten by this one)
                    if (t.distributing && pendingIdx[l] >= 0) tab[offs[2 * l + 1] + (size_t)pendingIdx[l]] = -1;
                    tab[offs[2 * l + 1] + (size_t)nbOf[l]] = t.wPos;        // where delayLineWrite will put it -- if its MAC completes
                    pendingIdx[l] = nbOf[l];
                    t.distributing = true;
                    t.nextPart = 0;
                }
                ++nbOf[l];
            }
            if (l > 0 && t.distributing) {                          // distributed MAC, once per Add() (:1497-1545)
                t.nextPart = std::min(t.nextPart + t.ppc, t.K);
                if (t.nextPart >= t.K) { t.wPos += t.P; t.distributing = false; t.nextPart = 0; }
            }
            // Get()
            if (l == 0) {
                const long long avail = t.wPos - t.rPos;
                const long long toRead = std::min<long long>(len, avail);
                tab[offs[0] + (size_t)c] = t.rPos;
                tab[offs[1] + (size_t)c] = toRead;
                t.rPos += toRead;
            } else {
                const long long maxRead = t.wPos >= t.outputDelay ? t.wPos - t.outputDelay : 0;
                const long long start = std::max(t.rPos, maxRead);
                if (start + len <= t.wPos) { tab[offs[2 * l] + (size_t)c] = start; t.rPos = start + len; }
            }
        }
    }
}

// layer 0 had no input waiting and nothing in its output ring before this call (w0 = its write position then)
static bool layer0WasEmpty(const PlanGroup& g, long long w0, long long r0)
{
    return g.layers[0].fill == 0 && r0 == w0;
}

int groupsAppend(cpq_engine* e, const double* dIn, int n)
{
    for (PlanGroup* gp : e->groups) {
        PlanGroup& g = *gp;
        if (g.frozen && e->honourFrozen) continue;      // only the processor-level call rests a stream (ConvolverProcessor does not call its NUC then); a NUC-level call runs every stream
        // every layer accumulates the same input (Add(), :1431-1446): one pass over it.  A layer whose accumulator is
        // empty and for which the call is whole partitions needs no accumulation at all when the group's rows are the
        // call's rows: its forward transforms run right here, straight from the call's input (before anything can write
        // dOut, which may alias dIn), and runLayerBlocks continues behind them.
        const bool rowsAreCallRows = g.identityMap && g.usedCh == e->nCh;
        // the reference's Add / Get bookkeeping of this call, replayed here so that a small table can leave with the gather
        // launch below (kernel arguments) instead of a host -> device copy of its own in front of layer 0
        g.callW0 = g.layers[0].wPos;
        g.callR0 = g.layers[0].rPos;
        replayCall(e, g, n, g.tabHost, g.tabOffs, g.nbOf);
        g.tabOnDevice = false;
        if ((int)g.tabHost.size() > g.tabCap) return fail(e, CPQ_ERR_INVALID_ARG, "call of %d samples exceeds the engine's call capacity", n);
        double* dst[3];
        int64_t stride[3], off[3];
        int nl = 0, li = 0;
        auto straight = [&](const NativeLayer& t) { return rowsAreCallRows && t.fill == 0 && n % t.P == 0 && n / t.P <= t.nbMax; };
        // layer 0 at 512 samples, transformed straight from the input, is held back: its launch can carry the other layers'
        // accumulation and the call's tables (one launch instead of two and a copy)
        const bool hold0 = straight(g.layers[0]) && g.layers[0].P == cpq::kP;
        for (NativeLayer& t : g.layers) {
            if (li++ == 3) break;
            t.fftAhead = 0;
            if (straight(t)) {
                t.fftAhead = n / t.P;
                if (hold0 && &t == &g.layers[0]) continue;
                ProfScope p(e, CPQ_K_RFFT_FWD);
                cpq::launch_rfft_fwd_ols(e->stream, dIn, (int64_t)n, t.hist[t.histSel], t.hist[t.histSel ^ 1], t.X, t.XDN,
                                         cpq::FftTables{ t.tw, t.tw2, t.twCol, t.twSplit }, t.P, g.usedCh, n / t.P, t.head, t.ringSlots, t.scratch);
                continue;
            }
            dst[nl] = t.acc[t.accSel];
            stride[nl] = t.accCap;
            off[nl] = t.fill;
            ++nl;
        }
        const bool ride = (int)g.tabHost.size() <= cpq::kGatherTabMax && g.usedCh > 0 && n > 0;
        bool gathered = nl == 0;
        if (hold0) {
            NativeLayer& t = g.layers[0];
            ProfScope p(e, CPQ_K_RFFT_FWD);
            if (cpq::rfft_fwd_can_carry_side(t.P, nl, stride, off, ride ? (int)g.tabHost.size() : 0)) {
                cpq::launch_rfft_fwd_ols_side(e->stream, dIn, (int64_t)n, t.hist[t.histSel], t.hist[t.histSel ^ 1], t.X, t.XDN,
                                              cpq::FftTables{ t.tw, t.tw2, t.twCol, t.twSplit }, g.usedCh, n / t.P, t.head, t.ringSlots, nl, dst,
                                              stride, off, ride ? g.tabDev : nullptr, g.tabHost.data(), ride ? (int)g.tabHost.size() : 0);
                gathered = true;
                g.tabOnDevice = ride;
            } else {
                cpq::launch_rfft_fwd_ols(e->stream, dIn, (int64_t)n, t.hist[t.histSel], t.hist[t.histSel ^ 1], t.X, t.XDN,
                                         cpq::FftTables{ t.tw, t.tw2, t.twCol, t.twSplit }, t.P, g.usedCh, n / t.P, t.head, t.ringSlots, t.scratch);
            }
        }
        if (!gathered) {
            ProfScope p(e, CPQ_K_MIX);
            cpq::launch_rows_gather_multi(e->stream, dIn, n, g.chMapDev, nl, dst, stride, off, n, g.usedCh,
                                          ride ? g.tabDev : nullptr, g.tabHost.data(), ride ? (int)g.tabHost.size() : 0);
            g.tabOnDevice = ride;
        }
    }
    CPQ_HIP(e, hipGetLastError());
    return CPQ_OK;
}

// one layer: every partition that filled up in this call is convolved (FFT, FDL push, MAC over the layer's partitions,
// IFFT; NUC.cpp:1245-1336 / :1456-1544) and stored in the layer's output ring / delay line
// directOut != nullptr (layer 0 only): the inverse transform writes the members' output rows themselves and the output ring is
// passed by -- the caller has checked that this call's Get() reads exactly what this call's blocks produce
// addTails (with directOut, layer 0 at 512 samples): the transform also adds the delay-line blocks of the tail layers, which
// the caller has run ahead of layer 0 (groupsRunLayer0)
static int runLayerBlocks(cpq_engine* e, PlanGroup& g, NativeLayer& t, int n, const long long* putPos, long long ringPos0,
                          double* directOut = nullptr, bool addTails = false)
{
    const int total = t.fill + n;
    const int nb = total / t.P;
    const int rem = total - nb * t.P;
    const int nCh = g.usedCh;
    if (nb > 0) {
        const cpq::FftTables tw{ t.tw, t.tw2, t.twCol, t.twSplit };
        bool remMoved = false;
        if (t.fftAhead != nb) {        // (else: transformed straight from the call's input in groupsAppend)
            ProfScope p(e, CPQ_K_RFFT_FWD);
            if (t.P == cpq::kP) {      // the 512-sample transform also moves the accumulator's remainder (no copy launch behind it)
                cpq::launch_rfft_fwd_ols_side(e->stream, t.acc[t.accSel], t.accCap, t.hist[t.histSel], t.hist[t.histSel ^ 1], t.X, t.XDN, tw, nCh, nb,
                                              t.head, t.ringSlots, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, t.acc[t.accSel ^ 1], t.accCap, rem);
                remMoved = true;
            } else {
                cpq::launch_rfft_fwd_ols(e->stream, t.acc[t.accSel], t.accCap, t.hist[t.histSel], t.hist[t.histSel ^ 1], t.X, t.XDN, tw,
                                         t.P, nCh, nb, t.head, t.ringSlots, t.scratch);
            }
        }
        t.fftAhead = 0;
        {
            ProfScope p(e, CPQ_K_FDL_MAC);
            cpq::launch_fdl_mac(e->stream, e->macTile, t.X, t.H, g.irSlotDev, t.Y, t.P, nCh, t.K, t.ringSlots, t.head, nb,
                                (int64_t)t.hRows * t.P, !g.shared);
        }
        if (cpq::fdl_mac_needs_dcnyq(e->macTile, nb)) {
            ProfScope p(e, CPQ_K_DCNYQ);
            cpq::launch_fdl_mac_dcnyq(e->stream, t.XDN, t.HDN, g.irSlotDev, t.Y, t.P, nCh, t.K, t.ringSlots, t.head, nb, t.hRows);
        }
        {
            // the finished blocks go where the reference puts them: layer 0 straight to the members' output rows (directOut:
            // the output ring is passed by -- read and write positions advanced together on the host), otherwise into the
            // layer's output ring / delay line at the replayed positions, written by the transform itself
            ProfScope p(e, CPQ_K_RFFT_INV);
            if (directOut && addTails) {
                NativeLayer& a = g.layers[1];
                NativeLayer* b = g.layers.size() == 3 ? &g.layers[2] : nullptr;
                cpq::launch_rfft_inv_ols_add(e->stream, t.Y, directOut, (int64_t)n, tw, nCh, nb, a.ring, a.outRing, g.tabDev + g.tabOffs[2], a.gain,
                                             b ? b->ring : nullptr, b ? b->outRing : 2, b ? g.tabDev + g.tabOffs[4] : nullptr, b ? b->gain : 0.0);
            }
            else if (directOut) cpq::launch_rfft_inv_ols(e->stream, t.Y, directOut, (int64_t)n, tw, t.P, nCh, nb, t.scratch);
            else           cpq::launch_rfft_inv_ols_ring(e->stream, t.Y, t.ring, t.outRing, putPos, ringPos0, tw, t.P, nCh, nb, t.scratch);
        }
        if (!remMoved && rem > 0) {
            ProfScope p(e, CPQ_K_MIX);
            cpq::launch_rows_copy(e->stream, t.acc[t.accSel], t.accCap, (int64_t)nb * t.P, t.acc[t.accSel ^ 1], t.accCap, 0, rem, nCh);
        }
        t.head = (t.head + nb) & (t.ringSlots - 1);
        t.histSel ^= 1;
        t.accSel ^= 1;
    }
    t.fill = rem;
    CPQ_HIP(e, hipGetLastError());
    return CPQ_OK;
}

// layer 0 of every group: blocks, then the chunk-wise ring read into the members' output rows (which it overwrites)
int groupsRunLayer0(cpq_engine* e, double* dOut, int n)
{
    for (PlanGroup* gp : e->groups) {
        PlanGroup& g = *gp;
        if (g.frozen && e->honourFrozen) continue;      // only the processor-level call rests a stream (ConvolverProcessor does not call its NUC then); a NUC-level call runs every stream
        const long long w0 = g.callW0, r0 = g.callR0;        // (replayed in groupsAppend)
        if (!g.tabOnDevice) { const int rc = stageUpload(e, g.tabDev, g.tabHost.data(), g.tabHost.size() * sizeof(long long)); if (rc != CPQ_OK) return rc; }
        // Whole-block call on an empty ring, every chunk's Get() taking exactly the chunk this call's blocks produce, members =
        // all channels in order: the inverse FFT writes the output rows directly, no ring put / get (two passes and two
        // launches less; the ring stays empty, its positions advanced on the host by the replay above)
        bool direct = g.identityMap && g.usedCh == e->nCh && layer0WasEmpty(g, w0, r0) && n % g.layers[0].P == 0;
        if (direct) {
            const size_t chunks = (size_t)((n + e->B - 1) / e->B);
            for (size_t c = 0; c < chunks && direct; ++c) {
                const long long want = std::min<long long>(e->B, n - (long long)c * e->B);
                direct = g.tabHost[g.tabOffs[0] + c] == w0 + (long long)c * e->B && g.tabHost[g.tabOffs[1] + c] == want;
            }
        }
        // Layer 0's transform can add the tail layers' delay-line blocks as it stores the output rows (no read-modify-write pass
        // over the output behind it) when every 512-sample block is one chunk of the call and nothing sits between layer 0 and
        // the tails in Get() (no direct head): the tail layers then run FIRST.  Same additions in the same order: bit-identical.
        g.tailsDone = false;
        const bool addTails = direct && g.layers.size() >= 2 && g.layers.size() <= 3 && g.layers[0].P == cpq::kP && e->B == cpq::kP && !e->anyDirect;
        if (addTails) {
            for (size_t l = 1; l < g.layers.size(); ++l) {
                const int rc = runLayerBlocks(e, g, g.layers[l], n, g.tabDev + g.tabOffs[2 * l + 1], 0);
                if (rc != CPQ_OK) return rc;
            }
            g.tailsDone = true;
        }
        { const int rc = runLayerBlocks(e, g, g.layers[0], n, nullptr, w0, direct ? dOut : nullptr, addTails); if (rc != CPQ_OK) return rc; }
        // Get() reads layer 0's ring chunk by chunk and then adds the tail layers' delay-line blocks: with nothing in between (no
        // direct head) the read waits for groupsRunTails, whose pass over the output does both
        g.getDeferred = !direct && !e->anyDirect && g.layers.size() >= 2 && g.layers.size() <= 3;
        if (!direct && !g.getDeferred) {
            ProfScope p(e, CPQ_K_MIX);
            cpq::launch_ring_get_chunks(e->stream, dOut, n, g.chMapDev, n, e->B, g.layers[0].ring, g.layers[0].outRing,
                                        g.tabDev + g.tabOffs[0], g.tabDev + g.tabOffs[1], g.usedCh);
        }
        g.samplesSinceReset += n;
        g.lastCall = n;
        g.lastGot = 0;
        for (size_t c = 0; c < g.tabOffs[2] - g.tabOffs[1] && g.tabOffs[1] + c < g.tabHost.size(); ++c) g.lastGot += (int)g.tabHost[g.tabOffs[1] + c];
    }
    CPQ_HIP(e, hipGetLastError());
    return CPQ_OK;
}

// tail layers of every group: blocks into the delay lines, then the chunk-wise read-add with the layer gain (:1620-1633)
int groupsRunTails(cpq_engine* e, double* dOut, int n)
{
    for (PlanGroup* gp : e->groups) {
        PlanGroup& g = *gp;
        if (g.frozen && e->honourFrozen) continue;      // only the processor-level call rests a stream (ConvolverProcessor does not call its NUC then); a NUC-level call runs every stream
        if (g.tailsDone) { g.tailsDone = false; continue; }     // (ran ahead of layer 0, whose transform added their blocks: groupsRunLayer0)
        for (size_t l = 1; l < g.layers.size(); ++l) {
            const int rc = runLayerBlocks(e, g, g.layers[l], n, g.tabDev + g.tabOffs[2 * l + 1], 0);
            if (rc != CPQ_OK) return rc;
        }
        ProfScope p(e, CPQ_K_MIX);
        if (g.getDeferred) {
            g.getDeferred = false;
            NativeLayer& z = g.layers[0];
            NativeLayer& a = g.layers[1];
            NativeLayer* b = g.layers.size() == 3 ? &g.layers[2] : nullptr;
            cpq::launch_ring_get_add_chunks(e->stream, dOut, n, g.chMapDev, n, e->B, z.ring, z.outRing, g.tabDev + g.tabOffs[0], g.tabDev + g.tabOffs[1],
                                            a.ring, a.outRing, g.tabDev + g.tabOffs[2], a.gain,
                                            b ? b->ring : nullptr, b ? b->outRing : 2, b ? g.tabDev + g.tabOffs[4] : nullptr, b ? b->gain : 0.0, g.usedCh);
            continue;
        }
        if (g.layers.size() == 3) {          // both delay lines in one pass over the output (layer 1 first, as Get() adds them)
            NativeLayer& a = g.layers[1];
            NativeLayer& b = g.layers[2];
            cpq::launch_ring_add_chunks2(e->stream, dOut, n, g.chMapDev, n, e->B, a.ring, a.outRing, g.tabDev + g.tabOffs[2], a.gain,
                                         b.ring, b.outRing, g.tabDev + g.tabOffs[4], b.gain, g.usedCh);
        } else if (g.layers.size() == 2) {
            NativeLayer& t = g.layers[1];
            cpq::launch_ring_add_chunks(e->stream, dOut, n, g.chMapDev, n, e->B, t.ring, t.outRing, g.tabDev + g.tabOffs[2], t.gain, g.usedCh);
        }
    }
    CPQ_HIP(e, hipGetLastError());
    return CPQ_OK;
}

}  // namespace cpqi
