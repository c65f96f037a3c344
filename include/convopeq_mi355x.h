/*
 * convopeq_mi355x.h -- C ABI of libconvopeq_mi355x.so
 *
 * MI355X-native (gfx950) batched drop-in for ONE hot path of lonewolf-jp/ConvoPeq:
 * the fp64 impulse-response convolver plus the 20-band TPT-SVF parametric EQ,
 * batched over S independent stereo streams (channel c = 2*stream + {0:L,1:R}).
 *
 * Boundary: this header is what the reference's FFI for the path would bind.
 * Each entry point cites the reference interface it replaces (paths relative to
 * the reference tree).  Plain pointers and sizes only; no C++/torch types; no
 * exceptions cross the ABI; every call returns a cpq_status (0 = ok, <0 = error)
 * unless noted.  One host thread per engine handle; handles are independent
 * (one per GPU / per process).
 *
 * PCM layout: planar [stream][channel][sample] fp64, i.e. channel c starts at
 * c * nSamples doubles.  The *_device entry points take device (HBM) pointers
 * and enqueue on the engine's stream without synchronising; the host-pointer
 * twins stage through the engine's device arena and return when the result is
 * in host memory.
 */
#ifndef CONVOPEQ_MI355X_H
#define CONVOPEQ_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPQ_ABI_VERSION 2

typedef enum {
    CPQ_OK               =  0,
    CPQ_ERR_INVALID_ARG  = -1,
    CPQ_ERR_NO_DEVICE    = -2,   /* no gfx950 device / HIP runtime failure at create */
    CPQ_ERR_OOM          = -3,
    CPQ_ERR_DEVICE       = -4,   /* a HIP call failed; see cpq_last_error() */
    CPQ_ERR_UNSUPPORTED  = -5,   /* valid in the reference, not implemented by this engine yet */
    CPQ_ERR_NOT_READY    = -6    /* process before set_impulse / prepare */
} cpq_status;

#define CPQ_ALL_STREAMS (-1)
#define CPQ_NUM_BANDS   20       /* EQProcessor::NUM_BANDS, src/eqprocessor/EQProcessor.h:153 */

/* what "the convolution" means for IRs longer than the reference's layer 0 */
typedef enum {
    /* y = x * h_eff: the closed form of MKLNonUniformConvolver's observable output (tail-layer
     * contouring gains and the constant layer lags of the distributed-MAC schedule and the B13 delay
     * line, src/MKLNonUniformConvolver.cpp:626-684,988-994,1497-1545,1653-1688).
     * Valid iff the reference itself is LTI for the configuration (cpq_nuc_plan.lti_valid). */
    CPQ_SEM_REFERENCE = 0,
    /* y = x * h: the mathematically exact linear convolution (== reference whenever irLen <= layer 0) */
    CPQ_SEM_EXACT = 1
} cpq_semantics;

typedef enum { CPQ_ORDER_CONV_THEN_EQ = 0, CPQ_ORDER_EQ_THEN_CONV = 1 } cpq_order;

/* how the IR is partitioned on the GPU (reference semantics only) */
typedef enum {
    /* one partition size for the whole h_eff (block_size or partition_size): every FDL / IR row is streamed by one
     * MAC kernel -- the HBM-roofline path of BASELINE.json configs[1] */
    CPQ_SCHED_UNIFORM = 0,
    /* the reference's own non-uniform (Gardner) schedule: layer 0 at block_size, tail layers at block_size * m and
     * block_size * m^2 (src/MKLNonUniformConvolver.cpp:738-758), each on its own FFT grid, outputs merged through
     * the replayed delay-line reader.  ~K0 + K1/m + K2/m^2 partition MACs per block instead of irLen / block_size
     * (BASELINE.json configs[3]).  Every IR of the engine must have the same layer plan. */
    CPQ_SCHED_REFERENCE_NUC = 1
} cpq_schedule;

/* which call sizes the process entry points accept */
typedef enum {
    /* n_samples = T * block_size with block_size a power of two (64..4096): every B-sample quantum is one Add + Get
     * pair of the reference with a full layer-0 partition, the regime in which its output is one linear convolution
     * (h_eff) and the time-batched uniform schedule applies -- the throughput path */
    CPQ_CALLS_WHOLE_BLOCKS = 0,
    /* any block_size from 1 to 4096 (the reference's blockSize / callQuantum: 480, 441, 96 ...) and any n_samples >= 1.
     * The call is cut into chunks of block_size samples (the last one may be shorter), each chunk is one Add(chunk) +
     * Get(chunk) of every MKLNonUniformConvolver (StereoConvolver::process, src/convolver/ConvolverProcessor.Runtime.cpp:
     * 1159-1184; chunking :659-682): input accumulates per layer until a partition of nextPow2(max(block_size, 64)) *
     * {1, m, m^2} samples is full (src/MKLNonUniformConvolver.cpp:1431-1446), layer 0 goes through the output ring and a
     * short read is zero-filled at the END of the chunk (:1376-1402), the tail layers follow the distributed MAC's
     * completion schedule and the delay-line reader per chunk (:1497-1545, :1653-1688) -- including the reference's
     * start-up gaps and dropped tail blocks at awkward quanta.  Every stream runs on the reference's own layer plan. */
    CPQ_CALLS_ANY = 1
} cpq_call_mode;

/* POD mirror of convo::FilterSpec, src/MKLNonUniformConvolver.h:123-133 */
typedef struct {
    double  sample_rate;
    int32_t hc_mode;                 /* HCMode: 0 Sharp, 1 Natural, 2 Soft (src/OutputFilter.h:75-80) */
    int32_t lc_mode;                 /* LCMode: 0 Natural, 1 Soft         (src/OutputFilter.h:85-89) */
    int32_t tail_mode;               /* 0 air absorption, 1 layer tail contouring, 2 bypass */
    int32_t tail_enabled;
    double  tail_start_seconds;
    double  tail_strength;
    int32_t tail_l1l2_multiplier;
    int32_t reserved;
} cpq_filter_spec;

/* The layer plan SetImpulse derives (src/MKLNonUniformConvolver.cpp:738-758,784-786,988-994,
 * 1005-1024) plus the A6 closed-form lags.  Host-only computation, no GPU needed. */
typedef struct {
    int32_t num_layers;
    int32_t part_size[3];
    int32_t offset[3];
    int32_t len[3];
    int32_t num_parts_ir[3];
    int32_t num_parts[3];
    int32_t parts_per_callback[3];
    int32_t output_delay[3];
    double  gain[3];
    int32_t direct_taps;
    int32_t latency;                 /* MKLNonUniformConvolver::getLatency(), src/MKLNonUniformConvolver.h:242 */
    int32_t lti_valid;
    int32_t done_callback[3];
    int32_t lag[3];
    int32_t heff_len;                /* taps of h_eff */
} cpq_nuc_plan;

/* POD mirror of EQCoeffsSVF, src/eqprocessor/EQProcessor.h:91-96 */
typedef struct { double g, k, a1, a2, a3, m0, m1, m2; } cpq_svf_coeffs;

/* POD mirror of convo::EQBandParams / convo::EQParameters, src/core/EQParameters.h:13-47 */
typedef struct {
    float   frequency;
    float   gain;
    float   q;
    int32_t enabled;
    int32_t type;                    /* 0 LowShelf, 1 Peaking, 2 HighShelf, 3 LowPass, 4 HighPass */
    int32_t channel_mode;            /* 0 Stereo, 1 Left, 2 Right, 3 Mid, 4 Side (Mid/Side: basic process(block) path) */
} cpq_eq_band;

typedef struct {
    cpq_eq_band bands[CPQ_NUM_BANDS];
    float   total_gain_db;
    int32_t agc_enabled;             /* block-rate AGC (processAGC): replaces the total-gain stage */
    float   nonlinear_saturation;    /* default 0.2 */
    int32_t filter_structure;        /* 0 Serial, 1 Parallel (parallel runs on the lane-skewed kernel) */
} cpq_eq_params;

typedef struct {
    int32_t struct_size;             /* sizeof(cpq_engine_desc) */
    int32_t device;                  /* HIP device ordinal */
    int32_t n_streams;               /* S stereo streams -> 2*S channels */
    int32_t block_size;              /* B: the caller's block / callQuantum.  CPQ_CALLS_WHOLE_BLOCKS: a power of two,
                                        64..4096, partition size P == B (layer-0 partSize of the reference);
                                        CPQ_CALLS_ANY: 1..4096, layer-0 partition nextPow2(max(B, 64)).  512 has
                                        dedicated wave-level FFT kernels, other sizes use generic ones */
    int32_t max_ir_len;              /* longest IR (taps) any stream will be given */
    int32_t max_blocks_per_call;     /* T_max: a process call carries 1..T_max blocks of B samples (CPQ_CALLS_ANY:
                                        1..T_max * B samples)
                                        (reference: up to 524288 samples per process(),
                                        src/convolver/ConvolverProcessor.Runtime.cpp:609,667-682) */
    int32_t semantics;               /* cpq_semantics */
    int32_t mac_tile;                /* 0 = default; else outputs per lane in the FDL MAC kernel (4/8/16/32) */
    double  sample_rate;
    int32_t partition_size;          /* internal FFT partition P: 0 = block_size; CPQ_PARTITION_AUTO = the fastest P
                                        for calls of max_blocks_per_call blocks (4096 when block_size *
                                        max_blocks_per_call is a multiple of it and at least 32768, else 512 when
                                        a multiple of that, else block_size;
                                        uniform schedule, whole-block calls, plain IRs - a FilterSpec with tail
                                        layers needs P == block_size;
                                        read it back with cpq_engine_partition_size); else a power of two with
                                        block_size <= P <= 4096.  The result is the same convolution (with the
                                        h_eff the reference derives for block_size); larger P trades call
                                        granularity for fewer partitions: every call must then carry a multiple
                                        of P samples (offline / batched use).  The reference itself runs its tail
                                        layers at 8x and 64x the block size (src/MKLNonUniformConvolver.cpp:738-740). */
    int32_t schedule;                /* cpq_schedule (was reserved: 0 = uniform) */
    int32_t call_mode;               /* cpq_call_mode: 0 = whole power-of-two blocks (default), 1 = any quantum / ragged calls */
    int32_t reserved;
} cpq_engine_desc;

#define CPQ_PARTITION_AUTO (-1)

typedef struct cpq_engine cpq_engine;

/* ------------------------------------------------------------------ library */
int32_t     cpq_abi_version(void);
const char* cpq_status_string(int32_t status);
/* last error text of this handle (or of the last failed create when e == NULL) */
const char* cpq_last_error(const cpq_engine* e);

/* ------------------------------------------------- host-only design helpers */
/* MKLNonUniformConvolver::SetImpulse layer plan (src/MKLNonUniformConvolver.cpp:626-684,738-758).
 * spec may be NULL (reference default: tail mode 1, start 0.085 s, strength 1, multiplier 8). */
int32_t cpq_nuc_plan_compute(int32_t ir_len, int32_t block_size, int32_t enable_direct_head,
                             const cpq_filter_spec* spec, cpq_nuc_plan* plan);
/* h_eff for one mono IR; writes min(cap, plan.heff_len) taps, returns heff_len or <0. */
int32_t cpq_nuc_heff(const double* ir, int32_t ir_len, int32_t block_size, double scale,
                     const cpq_filter_spec* spec, double* heff, int32_t cap);
/* EQProcessor::calcSVFCoeffs (src/eqprocessor/EQProcessor.Coefficients.cpp:101-130,431-618) */
int32_t cpq_eq_design_svf(int32_t type, float freq, float gain_db, float q, double sample_rate,
                          cpq_svf_coeffs* out);
/* convo::EQParameters::EQParameters() defaults (src/core/EQParameters.h:31-46) */
void    cpq_eq_params_default(cpq_eq_params* p);

/* ------------------------------------------------------------------- engine */
/* replaces: construction of StereoConvolver + 2 MKLNonUniformConvolver + EQProcessor per stream
 * and every per-buffer aligned allocation under them (src/AlignedAllocation.h:22-163,
 * src/MKLNonUniformConvolver.h:288-365): one device arena sized from the descriptor. */
int32_t cpq_engine_create(const cpq_engine_desc* desc, cpq_engine** out);
void    cpq_engine_destroy(cpq_engine* e);
/* hipStream_t the engine enqueues on (NULL = default stream). */
int32_t cpq_engine_set_stream(cpq_engine* e, void* hip_stream);
int32_t cpq_engine_synchronize(cpq_engine* e);
/* bytes of the device arena */
int64_t cpq_engine_arena_bytes(const cpq_engine* e);
/* the internal FFT partition size in use (what CPQ_PARTITION_AUTO resolved to); calls carry multiples of it */
int32_t cpq_engine_partition_size(const cpq_engine* e);

/* replaces ConvolverProcessor::prepareToPlay(double,int) (src/convolver/ConvolverProcessor.Lifecycle.cpp:211-402)
 * and EQProcessor::prepareToPlay(double,int) (src/eqprocessor/EQProcessor.Core.cpp:679-826):
 * publishes the rate, zeroes all run-time state (FDL, overlap history, SVF state) and, when the rate changed, re-designs
 * the EQ and OutputFilter coefficients from the parameters set so far (the reference rebuilds its band nodes on a rate
 * change).  The smoothers take their present targets at once (mix, total gain, EQ bypass fade, latency), pending band /
 * AGC reset requests are dropped with the state they would have cleared.  IR spectra are not touched: an IR belongs to
 * a rate, load it again if needed.  max_block must not exceed block_size * max_blocks_per_call. */
int32_t cpq_engine_prepare(cpq_engine* e, double sample_rate, int32_t max_block);
int32_t cpq_engine_set_order(cpq_engine* e, int32_t order);
/* Pin / unpin a caller buffer that is passed to the host-pointer entry points (cpq_*_process, cpq_engine_process_block):
 * those calls pipeline upload, kernels and download over time chunks, which overlaps fully only for pinned memory. */
int32_t cpq_host_register(void* ptr, size_t bytes);
int32_t cpq_host_unregister(void* ptr);

/* ---------------------------------------------------------------- convolver */
/* replaces StereoConvolver::init (src/ConvolverProcessor.h:741-814) -> 2 x
 * MKLNonUniformConvolver::SetImpulse(impulse, irLen, blockSize, scale, enableDirectHead, filterSpec)
 * (src/MKLNonUniformConvolver.h:197-200).  stream = index or CPQ_ALL_STREAMS (one shared stereo IR).
 * The caller keeps ownership of ir_l/ir_r (copied).
 * On a stream that is already playing the call leaves, like SetImpulse, a convolver that has seen no input (its input
 * history -- delay-line spectra, overlap block, the direct head's last samples -- is cleared); the other streams play on.
 * spec: NULL (the primary parity surface), or a FilterSpec: its HC/LC spectral gains (:336-443) and, in tail mode 0,
 * the air-absorption damping (:1060-1097) are applied to every partition spectrum at that LAYER's FFT size, exactly
 * as the reference does.  Layer 0 runs in the main path; every tail layer of the plan runs on the reference's own
 * partition grid (block_size * multiplier, ...) and reaches the output through the delay-line lag done_callback * B.
 * Tail partitions up to 131072 samples are supported (above 4096 through a four-step FFT); plans that are time-varying
 * in the reference (cpq_nuc_plan.lti_valid == 0) follow the replayed delay-line reader.
 * Limits (CPQ_ERR_UNSUPPORTED): tail partitions that are not a power of two or exceed 131072 (no plan of the reference does), FilterSpec IRs with
 * different layer plans in one engine, partition_size != block_size.
 * enable_direct_head: the first min(ir_len, 32) taps leave the FFT path before the spectra (and any FilterSpec gains)
 * are formed and run as a time-domain FIR over [history | block], flushed below 1e-20, added before the tail layers
 * (src/MKLNonUniformConvolver.cpp:689-731, 1169-1232, 1606-1618). */
int32_t cpq_conv_set_impulse(cpq_engine* e, int32_t stream, const double* ir_l, const double* ir_r,
                             int32_t ir_len, double scale, int32_t enable_direct_head,
                             const cpq_filter_spec* spec);
/* replaces the per-quantum MKLNonUniformConvolver::Add + Get pair (src/MKLNonUniformConvolver.h:208,218;
 * call site StereoConvolver::process, src/convolver/ConvolverProcessor.Runtime.cpp:1159-1184) for every
 * channel of every stream and every B-sample quantum in the call.  n_samples = T*B, 1 <= T <= T_max.
 * in == out is allowed. */
int32_t cpq_conv_process(cpq_engine* e, const double* in, double* out, int32_t n_samples);
int32_t cpq_conv_process_device(cpq_engine* e, const double* d_in, double* d_out, int32_t n_samples);
/* MKLNonUniformConvolver::Reset (src/MKLNonUniformConvolver.h:224) for all channels */
int32_t cpq_conv_reset(cpq_engine* e);
/* MKLNonUniformConvolver::isReady / getLatency (src/MKLNonUniformConvolver.h:229,242) */
int32_t cpq_conv_is_ready(const cpq_engine* e);
int32_t cpq_conv_latency(const cpq_engine* e);
int32_t cpq_conv_get_plan(const cpq_engine* e, cpq_nuc_plan* plan);
/* the return value of Get (src/MKLNonUniformConvolver.cpp:1553-1634) for the last process call of one stream: samples
 * layer 0's output ring delivered, summed over the call's chunks (n_samples unless the ring ran short: start-up, awkward
 * quanta -- the missing samples are zero-filled in the output as ringRead does).  Main-path streams always deliver
 * every sample.  < 0: bad argument. */
int32_t cpq_conv_last_got(const cpq_engine* e, int32_t stream);

/* ------------------------------------------- convolver, processor level (N1) */
/* Restatement of ConvolverProcessor::process(AudioBlock<double>&), steady state and the transitions of a live stream,
 * (src/convolver/ConvolverProcessor.Runtime.cpp:209-810) around the kernel-level convolver:
 *   dry signal through a delay line of (algorithmLatency + irPeakLatency) samples (:266-288, :549-567),
 *   wet = convolver output with NaN / Inf / |x| >= 1e300 replaced by 0 (:50-60, :722),
 *   out = wet * wetG + dry * dryG, wetG = equalPowerSin(mix) * CONVOLUTION_HEADROOM_GAIN (= 1.0),
 *   dryG = mix < 0.999 ? equalPowerSin(1 - mix) : 0, equalPowerSin = the 9th-order Taylor form (:26-31),
 *   so mix = 1 scales the wet signal by 1.0000035... (:373-375, :675-676, :611-657);
 *   mix <= 0.001: dry only, the convolver is not run (:573-585); bypassed: pure delay, convolver not run (:123-186).
 * A mix change after the first processor-level call is smoothed like the reference's mixSmoother (LinearRamp over
 * smoothing_time_sec, per-sample gains equalPowerSin(mix_i) / equalPowerSin(1 - mix_i) for every callback that starts
 * while the ramp runs, :340-375, :591-607); before it (and after cpq_engine_prepare) the mix applies at once, as
 * prepareToPlay sets it.  The latency compensation follows :263-290 and :394-540: the dry delay line is a ring that
 *   remembers B + max_ir_len samples (more when a larger ir_peak_latency is set); a total latency that moves by >= 2
 *   samples on a live stream is cross-faded over 20 ms from the delay in use (a move of one sample is not followed, as in
 *   the reference), a move during a running fade waits for its end; processing starts from latency + irLatency
 *   (Lifecycle.cpp:377-388), so with the direct head the first 20 ms fade from B + ir_peak_latency to ir_peak_latency.
 *   The reader's Catmull-Rom branch needs a fractional delay, which nothing in the reference produces: not built.
 * mix / ir_peak_latency may differ per stream; bypassed and mix <= 0.001 must be set for CPQ_ALL_STREAMS. */
typedef struct {
    float   mix;                 /* 0..1, default 1 (src/ConvolverProcessor.h:950) */
    int32_t bypassed;
    int32_t ir_peak_latency;     /* StereoConvolver::irLatency (peak delay of the loaded IR), samples */
    float   smoothing_time_sec;  /* mix ramp length; 0 = default 0.1 s (SMOOTHING_TIME_DEFAULT_SEC), else 0.01 .. 0.5 */
} cpq_convproc_params;
int32_t cpq_convproc_set_params(cpq_engine* e, int32_t stream, const cpq_convproc_params* p);
int32_t cpq_convproc_process(cpq_engine* e, const double* in, double* out, int32_t n_samples);
int32_t cpq_convproc_process_device(cpq_engine* e, const double* d_in, double* d_out, int32_t n_samples);
/* total dry-path delay in samples of one stream (algorithmLatency + irPeakLatency) */
int32_t cpq_convproc_delay(const cpq_engine* e, int32_t stream);
/* CPQ_LEVEL_NUC (default): cpq_engine_process_block uses the kernel-level convolver (primary parity surface);
 * CPQ_LEVEL_PROCESSOR: it uses cpq_convproc_process, as DSPCore does with ConvolverProcessor::process. */
typedef enum { CPQ_LEVEL_NUC = 0, CPQ_LEVEL_PROCESSOR = 1 } cpq_conv_level;
int32_t cpq_engine_set_conv_level(cpq_engine* e, int32_t level);

/* ----------------------------------------------------------------------- EQ */
/* replaces EQProcessor::createCoeffCache(params, sr, maxBlock, gen) (src/eqprocessor/
 * EQProcessor.ProcessingCache.cpp:56-93) + the (EQParameters, EQCoeffCache*) arguments of process(). */
int32_t cpq_eq_set_params(cpq_engine* e, int32_t stream, const cpq_eq_params* params);
/* replaces EQProcessor::process(AudioBlock<double>&, const EQParameters&, const EQCoeffCache*)
 * (src/eqprocessor/EQProcessor.Processing.cpp:1019-1276), serial structure, steady total gain. */
int32_t cpq_eq_process(cpq_engine* e, const double* in, double* out, int32_t n_samples);
int32_t cpq_eq_process_device(cpq_engine* e, const double* d_in, double* d_out, int32_t n_samples);
/* replaces EQProcessor::setBypassFromRT(bool) as DSPCore calls it before every block
 * (src/audioengine/AudioEngine.Processing.DSPCoreDouble.cpp:384; src/eqprocessor/EQProcessor.Processing.cpp:499-526,
 * 977-1015), per stream.  After the first processed call a change runs the reference's 5 ms bypass fade
 * (BYPASS_FADE_TIME_SEC): the callbacks that start while it runs go through the basic process(block) -- whose band nodes
 * leave flat non-LP/HP bands out -- and are cross-faded with the dry block per sample; once faded out the EQ of the
 * stream does nothing (filter states, total-gain ramp and AGC frozen); releasing the bypass clears the stream's filter
 * states and fades back in.  Before the first call (and at prepare / reset) the state follows the request at once. */
int32_t cpq_eq_set_bypass(cpq_engine* e, int32_t stream, int32_t bypassed);
/* replaces EQProcessor::requestBandReset (src/eqprocessor/EQProcessor.h; consumed in process(),
 * src/eqprocessor/EQProcessor.Processing.cpp:1083-1112 / :595-624): the filter states of the bands in band_mask (bit b =
 * band b; 0xFFFFFFFF = all, Mid / Side states included) are cleared at the start of the first callback whose input
 * block is silent (no sample above 1e-8, isAudioBlockSilent :460-475) or that runs a bypass fade; until then the request
 * stays pending.  While a request is pending on a playing stream each EQ call synchronises the engine's stream once
 * (the silence flags are read back); without a pending request nothing changes. */
int32_t cpq_eq_request_band_reset(cpq_engine* e, int32_t stream, uint32_t band_mask);
/* replaces EQProcessor::requestAgcReset (src/eqprocessor/EQProcessor.h:538-541): at the stream's next processed block the AGC
 * envelopes return to 0 and its gain to 1 (src/eqprocessor/EQProcessor.Processing.cpp:586-593, 1070-1077). */
int32_t cpq_eq_request_agc_reset(cpq_engine* e, int32_t stream);
/* EQ kernel choice.  AUTO: time-parallel kernel (per band: zero-state chunk runs + state scan; equal to the
 * sequential recurrence up to rounding, measured <= 3e-15) whenever the host can prove the reference's state
 * guards cannot trip, else the sequential kernel.  SEQUENTIAL: lane-skewed kernel that reproduces the
 * reference recurrence operation for operation (bit-identical to the SSE2+FMA path given equal coefficients). */
typedef enum { CPQ_EQ_MODE_AUTO = 0, CPQ_EQ_MODE_SEQUENTIAL = 1 } cpq_eq_mode;
int32_t cpq_eq_set_mode(cpq_engine* e, int32_t mode);
/* zero filterState (EQProcessor::prepareToPlay, src/eqprocessor/EQProcessor.Core.cpp:769) */
int32_t cpq_eq_reset(cpq_engine* e);

/* ------------------------------------------------- output filter (N2, adjacent) */
/* POD mirror of convo::BiquadCoeff (src/OutputFilter.h:40-44), a0-normalised Direct Form II Transposed */
typedef struct { double b0, b1, b2, a1, a2; } cpq_biquad_coeffs;
/* OutputFilter::prepare coefficient design (src/OutputFilter.cpp:23-121) for the three sections process() runs:
 * conv_is_last: low cut, high cut stage 0, stage 1; else: 20 Hz high-pass, low-pass stage 0, stage 1.  Host only. */
int32_t cpq_outfilter_design(int32_t conv_is_last, int32_t hc_mode, int32_t lc_mode, int32_t lp_mode,
                             double sample_rate, cpq_biquad_coeffs out[3]);
/* replaces OutputFilter::prepare + the mode arguments of process() (src/OutputFilter.h:106-131) */
int32_t cpq_outfilter_set_params(cpq_engine* e, int32_t stream, int32_t conv_is_last, int32_t hc_mode,
                                 int32_t lc_mode, int32_t lp_mode);
/* replaces OutputFilter::process(block, convIsLast, hcMode, lcMode, lpMode) stereo path: three DF-II-T biquads per
 * sample (biquadStep128_FMA, src/OutputFilter.cpp:143-165, :214-392) */
int32_t cpq_outfilter_process(cpq_engine* e, const double* in, double* out, int32_t n_samples);
int32_t cpq_outfilter_process_device(cpq_engine* e, const double* d_in, double* d_out, int32_t n_samples);
/* OutputFilter::reset (src/OutputFilter.cpp:126-137) */
int32_t cpq_outfilter_reset(cpq_engine* e);
/* on != 0: cpq_engine_process_block also runs the output filter after the conv/EQ pair, as DSPCore does
 * (src/audioengine/AudioEngine.Processing.DSPCoreDouble.cpp:453 ff.); default off */
int32_t cpq_engine_enable_output_filter(cpq_engine* e, int32_t on);

/* ------------------------------------------------------- whole path per call */
/* The remaining per-block values of DSPCore's routing (RuntimeSnapshot fields,
 * src/audioengine/AudioEngine.Processing.DSPCoreDouble.cpp:384-470), used by cpq_engine_process_block:
 * conv_input_trim_gain: convolverInputTrimGain, multiplied into the EQ output before the convolver in EQ -> conv order
 *   when it differs from 1 by more than 1e-12 (:438-447); output_makeup_gain: outputMakeupGain, multiplied into the
 *   block after the output filter (:465-469).
 * cpq_engine_set_conv_bypass: state.convBypassed -- the convolver stage is not called at all (no latency compensation;
 *   the processor-level bypass with its delay line is cpq_convproc_params.bypassed).  The EQ's bypass is
 *   cpq_eq_set_bypass.  With the output filter enabled it runs for the streams whose convolver or EQ is active (:453-463);
 *   conv_is_last of cpq_outfilter_set_params stays the caller's to pass, as DSPCore derives it (:458-459). */
int32_t cpq_engine_set_gains(cpq_engine* e, int32_t stream, double conv_input_trim_gain, double output_makeup_gain);
int32_t cpq_engine_set_conv_bypass(cpq_engine* e, int32_t bypassed);
/* replaces the DSPCore routing of convolverRt().process(block) and eqRt().process(block, params, cache)
 * (src/audioengine/AudioEngine.Processing.DSPCoreDouble.cpp:386-451) in the configured order. */
int32_t cpq_engine_process_block(cpq_engine* e, const double* in, double* out, int32_t n_samples);
int32_t cpq_engine_process_block_device(cpq_engine* e, const double* d_in, double* d_out, int32_t n_samples);

/* ---------------------------------------------------------------- IR ingest (host only, no GPU needed) */
/* Everything between an IR file and cpq_engine_set_impulse(): SURVEY.md N3.  One-off loader-thread work in the
 * reference; plain host code here. */
typedef struct {
    int32_t n_channels;
    int32_t n_samples;
    double  sample_rate;
    double* data;                    /* planar [channel][sample], owned by the library: cpq_ir_buffer_free() */
} cpq_ir_buffer;

/* IRConverter::ScaleFactorResult (src/IRConverter.h) plus the stage-2 analysis values it is derived from */
typedef struct {
    double  scale_factor;
    int32_t has_scale_factor;
    float   additional_attenuation_db;
    double  peak_value;              /* of the unscaled IR */
    double  rms_value;
    double  frequency_peak_gain;     /* IRAnalyzer::estimateMaxFrequencyResponseGain of the unscaled IR */
} cpq_ir_scale;

typedef struct {
    cpq_ir_buffer ir;                /* conditioned IR of the target length (stepTrimmed) */
    cpq_ir_scale  scale;             /* scale.scale_factor is what StereoConvolver::init / SetImpulse receive as `scale` */
    int32_t       ir_peak_latency;   /* estimatePeakLatencySamples: the processor-level dry delay on top of the block */
    int32_t       reserved;
} cpq_ir_prepared;

/* LoaderThread::doLoadStep for a WAV file (src/convolver/ConvolverProcessor.LoaderThread.cpp:431-486): RIFF / RF64,
 * PCM 8 / 16 / 24 / 32 bit and IEEE float 32 (plain or WAVE_FORMAT_EXTENSIBLE) -> float as JUCE's reader produces it
 * -> double, NaN and |v| < 1e-20 to 0, clamped to [-1, 1] (src/InputBitDepthTransform.h:31-100).
 * Frames a short file does not hold read as zero (as the reference's reader does); a data chunk claiming more than
 * twice the file size + 1 MiB is refused as corrupted instead.
 * CPQ_ERR_INVALID_ARG: missing file / no frames; CPQ_ERR_UNSUPPORTED: not a WAV the reference's reader would accept. */
int32_t cpq_ir_load_wav(const char* path, cpq_ir_buffer* out);
void    cpq_ir_buffer_free(cpq_ir_buffer* b);

/* ConvolverProcessor::PhaseMode (src/ConvolverProcessor.h:117-122) */
typedef enum { CPQ_PHASE_AS_IS = 0, CPQ_PHASE_MIXED = 1 /* not built: CPQ_ERR_UNSUPPORTED */, CPQ_PHASE_MINIMUM = 2 } cpq_phase_mode;

/* doTrimStep + doTransformStep (LoaderThread.cpp:490-641, 644-709): trailing-silence trim, 1 Hz DC
 * blocker, asymmetric Tukey window about the peak, zero-padded / cut to int(rate * target_ir_length_sec) samples
 * (IR_LENGTH 0.5..3 s, default 1 s; cap 2^21) with a linear fade-out, computeScaleFactor against the IR playing now
 * (current_ir may be NULL), peak latency.  An IR whose rate differs from sample_rate needs the reference's third-party
 * resampler (r8brain): CPQ_ERR_UNSUPPORTED. */
int32_t cpq_ir_prepare(const cpq_ir_buffer* loaded, double sample_rate, float target_ir_length_sec, int32_t phase_mode,
                       const cpq_ir_buffer* current_ir, double current_scale, cpq_ir_prepared* out);
/* ConvolverProcessorInternal::convertToMinimumPhase (src/convolver/ConvolverProcessor.ResampleAndFallback.cpp:333-469):
 * homomorphic minimum-phase reconstruction at 4x zero padding, every channel on its own.  CPQ_ERR_UNSUPPORTED where the
 * reference gives up (4 n above 8388608 samples, non-finite intermediate values). */
int32_t cpq_ir_convert_to_minimum_phase(const cpq_ir_buffer* in, cpq_ir_buffer* out);
void    cpq_ir_prepared_free(cpq_ir_prepared* p);

/* IRConverter::computeScaleFactor(ir, currentIr, currentScale) (src/IRConverter.cpp:175-196): energy normalisation to
 * -6 dB, then peak (0.5) / RMS (0.25) / frequency-response (+3 dB) clamps and the 4x jump protection. */
int32_t cpq_ir_compute_scale_factor(const double* const* ir, int32_t n_channels, int32_t n_samples,
                                    const double* const* current_ir, int32_t current_channels, int32_t current_samples,
                                    double current_scale, cpq_ir_scale* out);
/* IRAnalyzer::estimateMaxFrequencyResponseGain (src/IRAnalyzer.cpp:63-155) */
double  cpq_ir_estimate_max_frequency_response_gain(const double* const* ir, int32_t n_channels, int32_t n_samples);
/* LoaderThread::estimatePeakLatencySamples (LoaderThread.cpp:149-209) */
int32_t cpq_ir_estimate_peak_latency(const double* const* ir, int32_t n_channels, int32_t n_samples);

/* ---------------------------------------------------------------- profiling */
/* Per-kernel HIP-event timing on the engine's stream (counterpart of the reference's CONV_TIME /
 * EQ_TIME diagnostics, src/convolver/ConvolverProcessor.Runtime.cpp:679-721). */
typedef enum {
    CPQ_K_RFFT_FWD = 0,   /* k_rfft_fwd_ols */
    CPQ_K_FDL_MAC  = 1,   /* k_fdl_mac      */
    CPQ_K_DCNYQ    = 2,   /* k_fdl_mac_dcnyq */
    CPQ_K_RFFT_INV = 3,   /* k_rfft_inv_ols */
    CPQ_K_SVF      = 4,   /* k_svf_cascade (lane-skewed sequential recurrence) */
    CPQ_K_SVF_TP   = 5,   /* the time-parallel cascade kernels (k_svf_cascade_tpv / _short; default) */
    CPQ_K_MIX      = 6,   /* k_convproc_mix (processor-level dry/wet stage) */
    CPQ_K_OUTFILT  = 7,   /* output-filter biquad cascade (k_svf_cascade_tp / k_svf_cascade running DF-II-T sections) */
    CPQ_K_COUNT    = 8
} cpq_kernel_id;
int32_t     cpq_profile_enable(cpq_engine* e, int32_t on);
int32_t     cpq_profile_reset(cpq_engine* e);
/* synchronises the stream, then returns launches and summed milliseconds of one kernel */
int32_t     cpq_profile_read(cpq_engine* e, int32_t kernel_id, int64_t* launches, double* total_ms);
const char* cpq_kernel_name(int32_t kernel_id);

/* ---------------------------------------------------------------- diagnostics */
/* The partition FFT on its own (what replaces ProductionFft::forwardRealToCCS / inverseCCSToR, src/FFTBackend.cpp:123-150,
 * inside processLayerBlock): for n_channels x n_blocks blocks of `partition` samples (host, [channel][block][sample]; the
 * history before block 0 is silence) the forward transform of every overlap-save frame [previous block | block]
 * (2 * partition real points, unscaled) and the inverse transform of those spectra (scaled 1 / (2 * partition)), second half.
 * spectra: [channel][block][partition][2] in the kernels' own storage order -- element 0 = (DC, Nyquist), both real; element
 * e = bin e for partition <= 2048; for larger partitions element k1 * 512 + k2 = bin k1 + (partition / 512) * k2.
 * out: [channel][block][sample], equals the input up to rounding.  partition: a power of two in 64 ... 131072.
 * Needs a gfx950 device; no engine.  For tests of the FFT kernel families in isolation. */
int32_t     cpq_diag_partition_fft(int32_t partition, int32_t n_channels, int32_t n_blocks, const double* in,
                                   double* spectra, double* out);
/* Chained spans of the EQ / output-filter cascade (engines with fewer channels than the device holds workgroups of the span
 * kernel: the spans of a call are dealt to the workgroups and a band's state is handed from span to span inside the launch).
 * Synchronises the engine's stream; *launches = chained launches so far (0: this engine never chains), *gave_up != 0 when a
 * hand-over ever ran into its poll bound (a defect: results of that launch are not valid).  For tests. */
int32_t     cpq_diag_eq_chain_status(cpq_engine* e, uint32_t* launches, uint32_t* gave_up);

#ifdef __cplusplus
}
#endif
#endif /* CONVOPEQ_MI355X_H */
