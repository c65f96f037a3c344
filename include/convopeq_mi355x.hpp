// convopeq_mi355x.hpp -- header-only C++20 adapter over the C ABI (convopeq_mi355x.h).
//
// Gives reference-style callers the member-function names of the reference's hot-path surface, batched over
// S stereo streams:
//   cpq::BatchedConvolver  ::SetImpulse / Add / Get / Reset / isReady / getLatency
//        <- convo::MKLNonUniformConvolver (src/MKLNonUniformConvolver.h:197-242), one instance per mono channel
//           in the reference, here one object for every channel of every stream
//   cpq::BatchedProcessor  ::prepareToPlay / process / setEqParameters / loadImpulse
//        <- ConvolverProcessor::{prepareToPlay,process} (src/ConvolverProcessor.h:226,259) and
//           EQProcessor::{prepareToPlay,process(block, params, cache)} (src/eqprocessor/EQProcessor.h:189-205)
// Same argument meaning and error behaviour as the reference: bool / sample-count returns, never throws on the
// processing path, a failed call leaves the output zeroed (the reference's fail-closed FFT policy,
// src/MKLNonUniformConvolver.cpp:75-82) -- and, unlike the reference's void returns, every call that can fail also reports
// it: Add / process / prepareToPlay return false, lastStatus() / lastError() say why.
// Call sizes: a power-of-two block (64..4096) runs the throughput path (calls of whole blocks); any other block size
// (480, 441, 96 ...) or CallMode::Any selects CPQ_CALLS_ANY, where every call length is accepted and the reference's
// inputPos accumulation / ring zero-fill are reproduced chunk by chunk.
#pragma once

#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "convopeq_mi355x.h"

namespace cpq {

// planar block of all streams: channels()[c] points at numSamples doubles (juce::dsp::AudioBlock<double> is
// `double* const* channels, numChannels, numSamples`; here numChannels = 2 * streams)
struct AudioBlockBatch {
    double* const* channels;
    int numChannels;
    int numSamples;
};

// RAII pin of a caller-owned PCM buffer (cpq_host_register): the host-pointer entry points pipeline upload, kernels and
// download only for pinned memory.
class PinnedRegion {
public:
    PinnedRegion(void* ptr, size_t bytes) : ptr_(cpq_host_register(ptr, bytes) == CPQ_OK ? ptr : nullptr) {}
    ~PinnedRegion() { if (ptr_) (void)cpq_host_unregister(ptr_); }
    PinnedRegion(const PinnedRegion&) = delete;
    PinnedRegion& operator=(const PinnedRegion&) = delete;
    bool ok() const { return ptr_ != nullptr; }
private:
    void* ptr_;
};

enum class CallMode { Auto, WholeBlocks, Any };      // Auto: whole blocks for a power-of-two block size, else any

class Engine {
public:
    Engine(int streams, int blockSize, int maxIrLen, int maxBlocksPerCall, double sampleRate = 48000.0,
           cpq_semantics semantics = CPQ_SEM_REFERENCE, int device = 0, int partitionSize = 0,
           cpq_schedule schedule = CPQ_SCHED_UNIFORM, CallMode calls = CallMode::Auto)
    {
        const bool pow2 = blockSize >= 64 && blockSize <= 4096 && (blockSize & (blockSize - 1)) == 0;
        anyCalls_ = calls == CallMode::Any || (calls == CallMode::Auto && !pow2);
        cpq_engine_desc d{};
        d.struct_size = static_cast<int32_t>(sizeof(d));
        d.device = device;
        d.n_streams = streams;
        d.block_size = blockSize;
        d.max_ir_len = maxIrLen;
        d.max_blocks_per_call = maxBlocksPerCall;
        d.semantics = semantics;
        d.mac_tile = 0;
        d.sample_rate = sampleRate;
        d.partition_size = partitionSize;
        d.schedule = schedule;
        d.call_mode = anyCalls_ ? CPQ_CALLS_ANY : CPQ_CALLS_WHOLE_BLOCKS;
        cpq_engine* raw = nullptr;
        const int rc = cpq_engine_create(&d, &raw);
        if (rc != CPQ_OK) throw std::runtime_error(std::string("cpq_engine_create: ") + cpq_last_error(nullptr));
        h_.reset(raw);
        streams_ = streams;
        block_ = blockSize;
        maxCall_ = blockSize * maxBlocksPerCall;
    }
    cpq_engine* get() const noexcept { return h_.get(); }
    int streams() const noexcept { return streams_; }
    int channels() const noexcept { return 2 * streams_; }
    int blockSize() const noexcept { return block_; }
    int maxSamplesPerCall() const noexcept { return maxCall_; }
    bool acceptsAnyCallSize() const noexcept { return anyCalls_; }
    const char* lastError() const noexcept { return cpq_last_error(h_.get()); }

private:
    struct Deleter { void operator()(cpq_engine* e) const noexcept { cpq_engine_destroy(e); } };
    std::unique_ptr<cpq_engine, Deleter> h_;
    int streams_ = 0, block_ = 0, maxCall_ = 0;
    bool anyCalls_ = false;
};

// ---------------------------------------------------------------------------------------------------------------
class BatchedConvolver {
public:
    explicit BatchedConvolver(Engine& e) : e_(e), pending_(static_cast<size_t>(e.channels()) * e.maxSamplesPerCall()),
                                           result_(pending_.size()) {}

    // one shared stereo IR for every stream (CPQ_ALL_STREAMS) or one stream's IR; arguments as
    // MKLNonUniformConvolver::SetImpulse(impulse, irLen, blockSize, scale, enableDirectHead, filterSpec)
    bool SetImpulse(int stream, const double* impulseL, const double* impulseR, int irLen, int blockSize,
                    double scale = 1.0, bool enableDirectHead = false, const cpq_filter_spec* filterSpec = nullptr)
    {
        if (blockSize != e_.blockSize()) return false;
        return cpq_conv_set_impulse(e_.get(), stream, impulseL, impulseR, irLen, scale, enableDirectHead ? 1 : 0,
                                    filterSpec) == CPQ_OK;
    }

    // Add: input planar [channel][numSamples]; nullptr = silence (reference :205-208).  false (and lastStatus() != CPQ_OK)
    // when the engine refused the call -- e.g. a call that is not whole blocks on a CallMode::WholeBlocks engine
    bool Add(const double* input, int numSamples)
    {
        have_ = 0;
        status_ = CPQ_ERR_INVALID_ARG;
        if (numSamples <= 0 || numSamples > e_.maxSamplesPerCall()) return false;
        const size_t n = static_cast<size_t>(e_.channels()) * numSamples;
        if (input) std::memcpy(pending_.data(), input, n * sizeof(double));
        else std::memset(pending_.data(), 0, n * sizeof(double));
        status_ = cpq_conv_process(e_.get(), pending_.data(), result_.data(), numSamples);
        if (status_ == CPQ_OK) have_ = numSamples;
        return status_ == CPQ_OK;
    }

    // Get: the block of the preceding Add.  Returns the samples layer 0's ring delivered (the smallest count over the
    // streams; short reads are zero-filled at the end of every chunk exactly as ringRead does, :1376-1402), 0 and a zeroed
    // output when nothing is available (reference :1560-1562)
    int Get(double* output, int numSamples)
    {
        if (numSamples <= 0) return 0;
        const size_t n = static_cast<size_t>(e_.channels()) * numSamples;
        if (have_ != numSamples) {
            if (output) std::memset(output, 0, n * sizeof(double));
            return 0;
        }
        if (output) std::memcpy(output, result_.data(), n * sizeof(double));
        have_ = 0;
        int got = numSamples;
        for (int s = 0; s < e_.streams(); ++s) {
            const int g = cpq_conv_last_got(e_.get(), s);
            if (g >= 0 && g < got) got = g;
        }
        return got;
    }

    int lastStatus() const noexcept { return status_; }
    const char* lastError() const noexcept { return e_.lastError(); }

    void Reset() { cpq_conv_reset(e_.get()); }
    bool isReady() const noexcept { return cpq_conv_is_ready(e_.get()) != 0; }
    int getLatency() const noexcept { return cpq_conv_latency(e_.get()); }

private:
    Engine& e_;
    std::vector<double> pending_, result_;
    int have_ = 0, status_ = CPQ_OK;
};

// ---------------------------------------------------------------------------------------------------------------
class BatchedProcessor {
public:
    explicit BatchedProcessor(Engine& e) : e_(e), scratch_(static_cast<size_t>(e.channels()) * e.maxSamplesPerCall()) {}

    bool prepareToPlay(double sampleRate, int samplesPerBlock)
    {
        status_ = cpq_engine_prepare(e_.get(), sampleRate, samplesPerBlock);
        return status_ == CPQ_OK;
    }

    bool loadImpulse(int stream, const double* irL, const double* irR, int irLen, double scale = 1.0)
    {
        return cpq_conv_set_impulse(e_.get(), stream, irL, irR, irLen, scale, 0, nullptr) == CPQ_OK;
    }

    // IR file -> conditioned IR -> SetImpulse + peak latency, as the reference's loader thread does for one processor
    // (LoaderThread::doLoadStep .. buildConvolverFromTrimmed); a mono file feeds both channels
    bool loadImpulseFile(int stream, const char* wavPath, double sampleRate, float targetIrLengthSec = 1.0f,
                         cpq_phase_mode phase = CPQ_PHASE_AS_IS, const cpq_filter_spec* spec = nullptr, float mix = 1.0f)
    {
        cpq_ir_buffer file{};
        if (cpq_ir_load_wav(wavPath, &file) != CPQ_OK) return false;
        cpq_ir_prepared prep{};
        const bool ok = cpq_ir_prepare(&file, sampleRate, targetIrLengthSec, phase, nullptr, 1.0, &prep) == CPQ_OK;
        cpq_ir_buffer_free(&file);
        if (!ok) return false;
        const double* l = prep.ir.data;
        const double* r = prep.ir.n_channels > 1 ? prep.ir.data + prep.ir.n_samples : l;
        bool done = cpq_conv_set_impulse(e_.get(), stream, l, r, prep.ir.n_samples, prep.scale.scale_factor, 0, spec) == CPQ_OK;
        const cpq_convproc_params pp{ mix, 0, prep.ir_peak_latency, 0.0f };
        done = done && cpq_convproc_set_params(e_.get(), stream, &pp) == CPQ_OK;
        cpq_ir_prepared_free(&prep);
        return done;
    }

    bool setEqParameters(int stream, const cpq_eq_params& p) { return cpq_eq_set_params(e_.get(), stream, &p) == CPQ_OK; }
    void setProcessingOrder(cpq_order o) { cpq_engine_set_order(e_.get(), o); }
    // EQProcessor::setBypassFromRT / requestBandReset, per stream
    void setBypassFromRT(int stream, bool bypassed) { cpq_eq_set_bypass(e_.get(), stream, bypassed ? 1 : 0); }
    void requestBandReset(int stream, uint32_t mask) { cpq_eq_request_band_reset(e_.get(), stream, mask); }
    void requestAgcReset(int stream) { cpq_eq_request_agc_reset(e_.get(), stream); }
    // DSPCore's remaining per-block routing values
    void setConvolverBypassed(bool bypassed) { cpq_engine_set_conv_bypass(e_.get(), bypassed ? 1 : 0); }
    void setGains(int stream, double convolverInputTrimGain, double outputMakeupGain)
    {
        cpq_engine_set_gains(e_.get(), stream, convolverInputTrimGain, outputMakeupGain);
    }

    // in-place on the planar block, like ConvolverProcessor::process / EQProcessor::process; a refused or failed call
    // clears the block (fail closed) AND returns false with the reason in lastStatus() / lastError()
    bool process(AudioBlockBatch& block)
    {
        const int n = block.numSamples;
        status_ = CPQ_ERR_INVALID_ARG;
        if (n <= 0 || block.numChannels != e_.channels() || n > e_.maxSamplesPerCall()) { clear(block); return false; }
        for (int c = 0; c < block.numChannels; ++c)
            std::memcpy(scratch_.data() + static_cast<size_t>(c) * n, block.channels[c], sizeof(double) * n);
        status_ = cpq_engine_process_block(e_.get(), scratch_.data(), scratch_.data(), n);
        if (status_ != CPQ_OK) { clear(block); return false; }
        for (int c = 0; c < block.numChannels; ++c)
            std::memcpy(block.channels[c], scratch_.data() + static_cast<size_t>(c) * n, sizeof(double) * n);
        return true;
    }

    int lastStatus() const noexcept { return status_; }
    const char* lastError() const noexcept { return e_.lastError(); }

private:
    static void clear(AudioBlockBatch& b)
    {
        for (int c = 0; c < b.numChannels; ++c) std::memset(b.channels[c], 0, sizeof(double) * b.numSamples);
    }
    Engine& e_;
    std::vector<double> scratch_;
    int status_ = CPQ_OK;
};

}  // namespace cpq
