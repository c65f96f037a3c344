// kernels.hpp -- launchers of the gfx950 kernels of libconvopeq_mi355x (implemented in *.hip).
// All launchers enqueue on `stream` and never synchronise.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>

#include "host_design.hpp"

namespace cpq {

// partition size with dedicated wave-level FFT kernels; other powers of two (64..2048) use generic kernels
constexpr int kP = 512;            // samples per partition == complex bins per packed spectrum
constexpr int kMacPrefetch = 4;    // FDL/IR rows kept in flight per lane in k_fdl_mac
constexpr int kMacMaxTile = 32;    // largest outputs-per-lane tile; partition counts are padded to it
constexpr int kBands = 20;

struct FftTables {
    const double2* tw512;    // exp(-2 pi i m / P),  m < P   (P = partition size; 512 in the headline config)
    const double2* tw1024;   // exp(-2 pi i k / 2P), k < P
    // P > 4096 (four-step transforms, P = M1 * 512): the same values in the order the passes walk them, so that a wave reads
    // 64 consecutive entries instead of 64 cache lines (fill_big_twiddles):
    const double2* twCol = nullptr;     // [k1][n2] = tw512[n2 k1]         (column-pass twiddle W_P^(n2 k1))
    const double2* twSplit = nullptr;   // [k1][k2] = tw1024[k1 + M1 k2]   (real-FFT split of bin k1 + M1 k2)
};
// host: the two reordered tables of P entries each from tw512 / tw1024 (P > 4096)
void fill_big_twiddles(const double2* tw512, const double2* tw1024, int P, double2* twCol, double2* twSplit);

// Overlap-save framing + 1024-point real FFT of T blocks per channel into the frequency-domain delay
// line (FDL) ring; also saves the last block as the next call's overlap history.
void launch_rfft_fwd_ols(hipStream_t stream, const double* in, int64_t chStride, const double* histOld,
                         double* histNew, double2* X, double2* XDN, FftTables tw, int P, int nCh, int T,
                         int head, int ringSlots, double2* scratch = nullptr);

// P = 512 only (rfft_fwd_can_carry_side): the same launch also copies every block into up to two other accumulators
// (dst[a] + channel * dstStride[a] + dstOff[a], offsets even) and stores a table of <= kGatherTabMax entries to tabDst -- a plan
// group's input accumulation and chunk tables on the call's first launch instead of a k_rows_gather_multi launch of their own.
bool rfft_fwd_can_carry_side(int P, int nSide, const int64_t* dstStride, const int64_t* dstOff, int nTab);
void launch_rfft_fwd_ols_side(hipStream_t stream, const double* in, int64_t chStride, const double* histOld, double* histNew, double2* X,
                              double2* XDN, FftTables tw, int nCh, int T, int head, int ringSlots, int nSide, double* const* dst,
                              const int64_t* dstStride, const int64_t* dstOff, long long* tabDst, const long long* tab, int nTab,
                              double* tailDst = nullptr, int64_t tailStride = 0, int tailLen = 0);      // tail*: the tailLen samples behind the T blocks of every source row -> tailDst rows
// P = 512 inverse transform that also adds the delay-line blocks of up to two tail layers to the rows it stores: block t reads
// ring X at schedX[t] (negative: nothing to add), out = out + ring (gain within 1e-12 of 1) or out + ring * gain, A before B
void launch_rfft_inv_ols_add(hipStream_t stream, const double2* Y, double* out, int64_t chStride, FftTables tw, int nCh, int T,
                             const double* ringA, int ringSizeA, const long long* schedA, double gainA,
                             const double* ringB, int ringSizeB, const long long* schedB, double gainB);

// IR partition spectra: frames [h[k*P .. (k+1)*P) | 0] for k < nParts.
void launch_ir_spectra(hipStream_t stream, const double* heff, int heffLen, double2* H, double2* HDN,
                       FftTables tw, int P, int nParts, double2* scratch = nullptr);

// H[k][bin] *= gain[bin] (gain has P+1 entries: bins 0..P) for nParts partition spectra
void launch_spectrum_gain(hipStream_t stream, double2* H, double2* HDN, const double* gain, int P, int nParts);

// Y[c][t][bin] = sum_k X[c][slot(head+t-k)][bin] * H[ir(c)][k][bin], bins 1..511 (bin 0 is written but
// is overwritten by launch_fdl_mac_dcnyq).
// hPrivate: every channel has its own IR rows (no CPQ_ALL_STREAMS sharing): single-tile calls may then stream them past the cache
void launch_fdl_mac(hipStream_t stream, int tile, const double2* X, const double2* H, const int* irSlot,
                    double2* Y, int P, int nCh, int K, int ringSlots, int head, int T, int64_t hSlotStride,
                    bool hPrivate = false);      // K = partitions in use (walked in steps of the variant's tile)

// packed bin 0: DC and Nyquist are two independent real MACs.
// kernel variant launch_fdl_mac uses for (tile, T) (0 = workgroup-cooperative) and the multiple kPad must be padded to
int fdl_mac_variant(int tile, int T);
int fdl_mac_kpad_align(int tile, int T);
bool fdl_mac_needs_dcnyq(int tile, int T);    // only the 16- and 32-row register tiles leave packed bin 0 to launch_fdl_mac_dcnyq
void launch_fdl_mac_dcnyq(hipStream_t stream, const double2* XDN, const double2* HDN, const int* irSlot,
                          double2* Y, int P, int nCh, int K, int ringSlots, int head, int T, int hdnStride);

// inverse 1024-point real FFT of Y, scaled 1/N, second half (P samples) to out.
// P > 4096 (8192 ... 131072): four-step transforms through `scratch` ([transforms][P] double2); the spectra are
// then stored permuted (element k1 * 512 + k2 = bin k1 + (P / 512) k2), consistently in all three launchers.
void launch_rfft_inv_ols(hipStream_t stream, const double2* Y, double* out, int64_t chStride, FftTables tw,
                         int P, int nCh, int T, double2* scratch = nullptr);
// The same transform with its result stored straight into per-channel rings: block t of channel c at ring[c][(p_t + i) &
// (ringSize - 1)], p_t = pos[t] (device table; negative: the block is dropped) or pos0 + t P when pos is null -- the
// output-ring / delay-line write of a plan-group layer without the pass over the blocks in between.
void launch_rfft_inv_ols_ring(hipStream_t stream, const double2* Y, double* ring, int ringSize, const long long* pos,
                              long long pos0, FftTables tw, int P, int nCh, int T, double2* scratch = nullptr);

// 20-band TPT-SVF cascade, lane = (channel, band), bands skewed in time across lanes.  streamPairs: one wave per
// stream (its L and R channel) instead of 3 channels per wave -- required when a band has flag bits 4/5
// (Mid / Side component band).
void launch_svf_cascade(hipStream_t stream, const double* in, double* out, int64_t chStride, int nCh,
                        int nSamples, const double* coef, const int* flags, const double* satGain,
                        double* state, bool streamPairs);

// Time-parallel SVF cascade (svf_kernels.hip): whole 8192-sample spans on k_svf_cascade_tpv<8> (one workgroup per channel, or
// chained spans, below), what is left from 1024 samples up as one span of 1 ... 7 waves x 1024 samples whose tail may be
// padding, the rest (and calls below 1024 samples) on k_svf_cascade_short.  nSamples: any number.
// tables: kSvfTpTableDoubles doubles per (stream, band): for each chunk length LC in kSvfTpLc ({16, 8}): Mk[6][4] = A^(LC 2^k),
// Mw[4] = A^(64 LC), P[64][4] = A^(LC (c+1)), G[16][2] = C A^i; then the chunk's end-state map e[2][16] = A^(15-k) B.
// geometry constants kSvfTpLc / kSvfTpTableDoubles: host_design.hpp (shared with the table builder)
// chain / chainSpans / chainGrid: chained spans for engines whose channels do not fill whole rounds of the workgroups the chip
// holds of the span kernel (chainGrid = 2 per CU): the (span, channel) pairs of a call are dealt to chainGrid workgroups, a band's state is
// handed from span to span through `chain` (svf_chain_bytes(channels, largest call) bytes, zero-initialised once,
// chainSpans = svf_chain_spans(largest call); used by one launch at a time).  chainSpans = 0: one workgroup per channel;
// `chain` (svf_chain_bytes(channels, 0) bytes, zero-initialised once; may be nullptr) then only holds the arrival counters by
// which the two workgroups of a CU take turns at raised priority.
void launch_svf_cascade_tp(hipStream_t stream, const double* in, double* out, int64_t chStride, int nCh,
                           int nSamples, const double* coef, const int* flags, const double* satGain,
                           double* state, const void* tables, void* chain = nullptr, int chainSpans = 0,
                           int chainGrid = 0);
size_t svf_chain_bytes(int nCh, int maxSamples);
int svf_chain_spans(int maxSamples);

// Processor-level dry/wet stage (ConvolverProcessor::process): out = sanitize(wet) * wetG + dryDelayed * dryG over one
// range of a call.  The dry signal is read from the delay ring ([nCh][ringSize], the call's input already written at
// absolute position pos0 = position of the range's first sample) dNew[stream] samples back; the first xLen[stream]
// samples are cross-faded from the delay dOld[stream] with the gains xGains[stream][i] (latency change); gains:
// [streams][2] = {wetG, dryG}; rampLen / rampGains: per-sample mix gains of the first samples of the CALL (rampOff =
// offset of this range in the call).
void launch_convproc_mix(hipStream_t stream, const double* wet, double* out, int64_t chStride, int nCh, int nSamples,
                         const double* gains, const double* ring, int ringSize, long long pos0, const int* dNew,
                         const int* dOld, const int* xLen, const double* xGains, int xCap, int wetValid,
                         const int* rampLen = nullptr, const double* rampGains = nullptr, int rampCap = 0, int rampOff = 0,
                         const int* wetOn = nullptr);       // wetOn[stream] == 0: that stream's convolver rests (delayed dry only)
void launch_ring_regrow(hipStream_t stream, const double* oldRing, int oldSize, double* newRing, int newSize, long long end,
                        int nCh);

// EQ AGC (EQProcessor::processAGC): per-callback-block RMS in the reference's accumulation order, then envelopes /
// gain per block (one thread per stream) and the linear gain ramp with the reference's incremental-add pattern.
// rms: [nCh][T]; state: [S][3] = envIn, envOut, gain-1; gains: [S][T][2] = start gain, per-sample increment.
void launch_agc_block_rms(hipStream_t stream, const double* x, int64_t chStride, int nCh, int B, int T, double* rms);
void launch_agc_apply(hipStream_t stream, double* data, int64_t chStride, int S, int B, int T, const double* rmsIn,
                      const double* rmsOut, double* state, const int* agcOn, double* gains, double bAtt, double bRel,
                      double bSm);

// applyGainRamp_AVX2 for the streams flagged in `on`: gains [S][T][2] = start gain, per-sample increment per callback
void launch_gain_ramp(hipStream_t stream, double* data, int64_t chStride, int S, int B, int T, const double* gains,
                      const int* on);

// FilterSpec tail layers at the reference's partition size: input accumulation, delay-line write, delay-line read-add
void launch_rows_copy(hipStream_t stream, const double* src, int64_t srcStride, int64_t srcOff, double* dst,
                      int64_t dstStride, int64_t dstOff, int n, int nCh);
void launch_ring_put(hipStream_t stream, const double* z, int64_t zStride, int n, double* ring, int ringSize,
                     long long pos, int nCh);
void launch_ring_add(hipStream_t stream, double* out, int64_t outStride, int n, int B, const double* ring, int ringSize,
                     const long long* sched, double gain, int nCh);
// Layered mode with the reader's additions inside the layer-0 inverse transform: the schedule first (launch_tail_schedule),
// then launch_rfft_inv_ols_tail writes out = layer-0 convolution + what the reader adds (from layerOut / the rings), then
// launch_tail_append stores what later calls may still read into the rings.
void launch_rfft_inv_ols_tail(hipStream_t stream, const double2* Y, double* out, int64_t chStride, FftTables tw, int P, int nCh,
                              int T, double2* scratch, const double* layerOut, const double* tailRing, int tailRingSize,
                              const void* tailState, const long long* sched, int nCallbacks, int B, int nTail, double g1, double g2);
void launch_tail_append(hipStream_t stream, const void* state, const double* layerOut, double* ring, int nCh, int nSamples,
                        int ringSlots, int nTail);
// replay of the delay-line reader for the T callbacks of a call (state: 4 long long; sched: [nTail][T], -1 = skip)
void launch_tail_schedule(hipStream_t stream, void* state, long long* sched, int T, int B, int nTail, int pl1, int ol1, int d1,
                          int pl2, int ol2, int d2);

// Plan groups (engine_native.cpp): chMap[local channel] = row of the call's buffers (-1 = unused slot); q = chunk length
// (one Add / Get pair of the reference per chunk); pos / cnt / sched: per chunk, replayed on the host.
void launch_rows_gather(hipStream_t stream, const double* src, int64_t srcStride, const int* chMap, double* dst,
                        int64_t dstStride, int64_t dstOff, int n, int nCh);
// up to three layers' accumulators in one pass over the input; both tail layers' read-add in one pass over the output
// tabDst / tab / nTab: a small table (<= kGatherTabMax entries) that rides along as kernel arguments and is stored to tabDst by
// the launch -- the call's chunk schedule of a plan group without a host -> device copy of its own
constexpr int kGatherTabMax = 64;
void launch_rows_gather_multi(hipStream_t stream, const double* src, int64_t srcStride, const int* chMap, int nLayers,
                              double* const* dst, const int64_t* dstStride, const int64_t* dstOff, int n, int nCh,
                              long long* tabDst = nullptr, const long long* tab = nullptr, int nTab = 0);
void launch_ring_add_chunks2(hipStream_t stream, double* out, int64_t outStride, const int* chMap, int n, int q,
                             const double* ringA, int ringSizeA, const long long* schedA, double gainA,
                             const double* ringB, int ringSizeB, const long long* schedB, double gainB, int nCh);
void launch_ring_put_blocks(hipStream_t stream, const double* z, int64_t zStride, int P, int nb, double* ring, int ringSize,
                            const long long* pos, int nCh);
// launch_ring_get_chunks and launch_ring_add_chunks[2] in one pass over the output (ringB may be null)
void launch_ring_get_add_chunks(hipStream_t stream, double* out, int64_t outStride, const int* chMap, int n, int q,
                                const double* ring0, int ringSize0, const long long* pos, const long long* cnt,
                                const double* ringA, int ringSizeA, const long long* schedA, double gainA,
                                const double* ringB, int ringSizeB, const long long* schedB, double gainB, int nCh);
void launch_ring_get_chunks(hipStream_t stream, double* out, int64_t outStride, const int* chMap, int n, int q,
                            const double* ring, int ringSize, const long long* pos, const long long* cnt, int nCh);
void launch_ring_add_chunks(hipStream_t stream, double* out, int64_t outStride, const int* chMap, int n, int q,
                            const double* ring, int ringSize, const long long* sched, double gain, int nCh);

// EQ bypass cross-fade for the streams flagged in `on`: out = out * g + dry * (1 - g); g = gains[s][i] for i < len[s], gEnd[s] after
void launch_bypass_blend(hipStream_t stream, double* out, int64_t outStride, const double* dry, int64_t dryStride, int n,
                         int nCh, const int* on, const int* len, const double* gEnd, const double* gains, int cap);
// silent[stream][callback] = no sample of the stream's two channels above 1e-8 in that callback block
void launch_block_silence(hipStream_t stream, const double* x, int64_t chStride, int B, int T, int S, int* silent);
// data[c][i] *= gain[c / 2] (streams with gain exactly 1 are left alone)
void launch_rows_scale(hipStream_t stream, double* data, int64_t stride, int n, int nCh, const double* gain);

// direct head: time-domain FIR of the first <= 32 taps over [history | block] into dout ([nCh][n]); then out += dout
// wetOn[stream] == 0: that stream's head rests (zero output, history kept)
void launch_direct_head(hipStream_t stream, const double* in, int64_t inStride, int n, const double* irRev, const int* taps,
                        const int* irSlot, const double* histOld, double* histNew, double* dout, int nCh,
                        const int* wetOn = nullptr);
void launch_rows_add(hipStream_t stream, double* out, int64_t outStride, const double* add, int n, int nCh);

}  // namespace cpq
