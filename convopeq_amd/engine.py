"""Host-side harness over the C ABI (tests / bench plumbing).

`BatchedEngine` mirrors the reference's processor surface for the hot path, batched over S stereo streams:
  prepare_to_play(sample_rate, max_block)   <- ConvolverProcessor/EQProcessor::prepareToPlay
  set_impulse(stream, ir_l, ir_r, ...)      <- StereoConvolver::init -> MKLNonUniformConvolver::SetImpulse
  set_eq_params(stream, params)             <- EQProcessor::createCoeffCache + process(block, params, cache)
  process(...) / conv_process / eq_process  <- ConvolverProcessor::process / EQProcessor::process
All numerical work happens in libconvopeq_mi355x.so on the GPU; numpy is only the host buffer type.
"""
import ctypes as C
import os

import numpy as np

from . import _capi as K


class CpqError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"cpq status {status} ({K.load().cpq_status_string(status).decode()}): {msg}")
        self.status = status


def _dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(K.c_double_p)


def nuc_plan(ir_len, block, direct=False, spec=None):
    p = K.NucPlan()
    rc = K.load().cpq_nuc_plan_compute(ir_len, block, int(direct), C.byref(spec) if spec is not None else None,
                                       C.byref(p))
    if rc != 0:
        raise CpqError(rc, "cpq_nuc_plan_compute")
    return p


def nuc_heff(ir, block, scale=1.0, spec=None):
    ir = np.ascontiguousarray(ir, dtype=np.float64)
    sp = C.byref(spec) if spec is not None else None
    n = K.load().cpq_nuc_heff(_dp(ir), len(ir), block, scale, sp, None, 0)
    if n < 0:
        raise CpqError(n, "cpq_nuc_heff")
    out = np.zeros(n, dtype=np.float64)
    K.load().cpq_nuc_heff(_dp(ir), len(ir), block, scale, sp, _dp(out), n)
    return out


def design_svf(btype, freq, gain_db, q, sr):
    c = K.SvfCoeffs()
    rc = K.load().cpq_eq_design_svf(btype, freq, gain_db, q, sr, C.byref(c))
    if rc != 0:
        raise CpqError(rc, "cpq_eq_design_svf")
    return c


def outfilter_design(conv_is_last, hc_mode, lc_mode, lp_mode, sr):
    out = (K.BiquadCoeffs * 3)()
    rc = K.load().cpq_outfilter_design(int(conv_is_last), hc_mode, lc_mode, lp_mode, sr, out)
    if rc != 0:
        raise CpqError(rc, "cpq_outfilter_design")
    return list(out)


def eq_params_default():
    p = K.EqParams()
    K.load().cpq_eq_params_default(C.byref(p))
    return p


def _planes(a):
    a = np.ascontiguousarray(np.atleast_2d(a), dtype=np.float64)
    ptrs = (K.c_double_p * a.shape[0])(*[a[c].ctypes.data_as(K.c_double_p) for c in range(a.shape[0])])
    return a, ptrs


def _ir_buffer(a, rate):
    a = np.ascontiguousarray(np.atleast_2d(a), dtype=np.float64)
    return a, K.IrBuffer(a.shape[0], a.shape[1], float(rate), a.ctypes.data_as(K.c_double_p))


def _scale_dict(s):
    return {"scale_factor": s.scale_factor, "has_scale_factor": bool(s.has_scale_factor),
            "additional_attenuation_db": s.additional_attenuation_db, "peak_value": s.peak_value,
            "rms_value": s.rms_value, "frequency_peak_gain": s.frequency_peak_gain}


def ir_load_wav(path):
    """cpq_ir_load_wav: (planes [channels][samples] float64, sample rate)."""
    b = K.IrBuffer()
    rc = K.load().cpq_ir_load_wav(os.fsencode(path), C.byref(b))
    if rc != 0:
        raise CpqError(rc, f"cpq_ir_load_wav({path})")
    try:
        out = np.ctypeslib.as_array(b.data, shape=(b.n_channels, b.n_samples)).copy()
        return out, b.sample_rate
    finally:
        K.load().cpq_ir_buffer_free(C.byref(b))


def ir_prepare(ir, ir_rate, sample_rate, target_ir_length_sec=1.0, current_ir=None, current_scale=1.0, phase_mode=0):
    """cpq_ir_prepare: dict(ir=[channels][target] float64, scale=..., ir_peak_latency=...)."""
    keep, b = _ir_buffer(ir, ir_rate)
    cur = None
    if current_ir is not None:
        keep2, cur = _ir_buffer(current_ir, sample_rate)
    out = K.IrPrepared()
    rc = K.load().cpq_ir_prepare(C.byref(b), sample_rate, target_ir_length_sec, phase_mode, C.byref(cur) if cur is not None else None,
                                 current_scale, C.byref(out))
    if rc != 0:
        raise CpqError(rc, "cpq_ir_prepare")
    try:
        data = np.ctypeslib.as_array(out.ir.data, shape=(out.ir.n_channels, out.ir.n_samples)).copy()
        return {"ir": data, "sample_rate": out.ir.sample_rate, "scale": _scale_dict(out.scale),
                "ir_peak_latency": out.ir_peak_latency}
    finally:
        K.load().cpq_ir_prepared_free(C.byref(out))


def ir_convert_to_minimum_phase(ir, rate=48000.0):
    keep, b = _ir_buffer(ir, rate)
    out = K.IrBuffer()
    rc = K.load().cpq_ir_convert_to_minimum_phase(C.byref(b), C.byref(out))
    if rc != 0:
        raise CpqError(rc, "cpq_ir_convert_to_minimum_phase")
    try:
        return np.ctypeslib.as_array(out.data, shape=(out.n_channels, out.n_samples)).copy()
    finally:
        K.load().cpq_ir_buffer_free(C.byref(out))


def ir_compute_scale_factor(ir, current_ir=None, current_scale=1.0):
    a, pa = _planes(ir)
    cur_ptrs, cc, cn = None, 0, 0
    if current_ir is not None:
        c, cur_ptrs = _planes(current_ir)
        cc, cn = c.shape
    out = K.IrScale()
    rc = K.load().cpq_ir_compute_scale_factor(pa, a.shape[0], a.shape[1], cur_ptrs, cc, cn, current_scale, C.byref(out))
    if rc != 0:
        raise CpqError(rc, "cpq_ir_compute_scale_factor")
    return _scale_dict(out)


def ir_estimate_max_frequency_response_gain(ir):
    a, pa = _planes(ir)
    return K.load().cpq_ir_estimate_max_frequency_response_gain(pa, a.shape[0], a.shape[1])


def ir_estimate_peak_latency(ir):
    a, pa = _planes(ir)
    return K.load().cpq_ir_estimate_peak_latency(pa, a.shape[0], a.shape[1])


class BatchedEngine:
    def __init__(self, n_streams, block_size=512, max_ir_len=131072, max_blocks_per_call=64,
                 semantics=K.CPQ_SEM_REFERENCE, device=0, sample_rate=48000.0, mac_tile=0, partition_size=0,
                 schedule=K.CPQ_SCHED_UNIFORM, call_mode=K.CPQ_CALLS_WHOLE_BLOCKS):
        self._lib = K.load()
        self._h = K._E()
        d = K.EngineDesc(C.sizeof(K.EngineDesc), device, n_streams, block_size, max_ir_len, max_blocks_per_call,
                         semantics, mac_tile, sample_rate, partition_size, schedule, call_mode, 0)
        rc = self._lib.cpq_engine_create(C.byref(d), C.byref(self._h))
        if rc != 0:
            raise CpqError(rc, self._lib.cpq_last_error(None).decode())
        self.n_streams = n_streams
        self.n_channels = 2 * n_streams
        self.block_size = block_size

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cpq_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _ck(self, rc):
        if rc != 0:
            raise CpqError(rc, self._lib.cpq_last_error(self._h).decode())

    # ---- control surface
    def set_stream(self, hip_stream_ptr):
        self._ck(self._lib.cpq_engine_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    def synchronize(self):
        self._ck(self._lib.cpq_engine_synchronize(self._h))

    def arena_bytes(self):
        return self._lib.cpq_engine_arena_bytes(self._h)

    def partition_size(self):
        """The internal FFT partition in use (what CPQ_PARTITION_AUTO resolved to)."""
        return self._lib.cpq_engine_partition_size(self._h)

    def prepare_to_play(self, sample_rate, max_block):
        self._ck(self._lib.cpq_engine_prepare(self._h, sample_rate, max_block))

    def set_order(self, order):
        self._ck(self._lib.cpq_engine_set_order(self._h, order))

    def set_impulse(self, stream, ir_l, ir_r, scale=1.0, direct_head=False, spec=None):
        ir_l = np.ascontiguousarray(ir_l, dtype=np.float64)
        ir_r = np.ascontiguousarray(ir_r, dtype=np.float64)
        assert len(ir_l) == len(ir_r)
        self._ck(self._lib.cpq_conv_set_impulse(self._h, stream, _dp(ir_l), _dp(ir_r), len(ir_l), scale,
                                                int(direct_head), C.byref(spec) if spec is not None else None))

    def set_eq_params(self, stream, params):
        self._ck(self._lib.cpq_eq_set_params(self._h, stream, C.byref(params)))

    def set_gains(self, stream, conv_input_trim_gain=1.0, output_makeup_gain=1.0):
        self._ck(self._lib.cpq_engine_set_gains(self._h, stream, conv_input_trim_gain, output_makeup_gain))

    def set_conv_bypass(self, bypassed):
        self._ck(self._lib.cpq_engine_set_conv_bypass(self._h, int(bypassed)))

    def request_band_reset(self, stream, band_mask=0xFFFFFFFF):
        self._ck(self._lib.cpq_eq_request_band_reset(self._h, stream, band_mask))

    def request_agc_reset(self, stream):
        self._ck(self._lib.cpq_eq_request_agc_reset(self._h, stream))

    def set_eq_bypass(self, stream, bypassed):
        self._ck(self._lib.cpq_eq_set_bypass(self._h, stream, int(bypassed)))

    def set_convproc_params(self, stream, mix=1.0, bypassed=False, ir_peak_latency=0, smoothing_time_sec=0.0):
        p = K.ConvProcParams(mix, int(bypassed), ir_peak_latency, smoothing_time_sec)
        self._ck(self._lib.cpq_convproc_set_params(self._h, stream, C.byref(p)))

    def convproc_delay(self, stream):
        return self._lib.cpq_convproc_delay(self._h, stream)

    def set_conv_level(self, level):
        self._ck(self._lib.cpq_engine_set_conv_level(self._h, level))

    def convproc_process(self, x):
        return self._host(self._lib.cpq_convproc_process, x)

    def set_outfilter_params(self, stream, conv_is_last, hc_mode=1, lc_mode=0, lp_mode=1):
        self._ck(self._lib.cpq_outfilter_set_params(self._h, stream, int(conv_is_last), hc_mode, lc_mode, lp_mode))

    def outfilter_process(self, x):
        return self._host(self._lib.cpq_outfilter_process, x)

    def enable_output_filter(self, on=True):
        self._ck(self._lib.cpq_engine_enable_output_filter(self._h, int(on)))

    def outfilter_reset(self):
        self._ck(self._lib.cpq_outfilter_reset(self._h))

    def set_eq_mode(self, mode):
        self._ck(self._lib.cpq_eq_set_mode(self._h, mode))

    def conv_reset(self):
        self._ck(self._lib.cpq_conv_reset(self._h))

    def eq_reset(self):
        self._ck(self._lib.cpq_eq_reset(self._h))

    def is_ready(self):
        return bool(self._lib.cpq_conv_is_ready(self._h))

    def latency(self):
        return self._lib.cpq_conv_latency(self._h)

    def last_got(self, stream=0):
        return self._lib.cpq_conv_last_got(self._h, stream)

    def plan(self):
        p = K.NucPlan()
        self._ck(self._lib.cpq_conv_get_plan(self._h, C.byref(p)))
        return p

    # ---- host-buffer processing: x is [n_channels, n_samples] float64
    def _host(self, fn, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.ndim == 2 and x.shape[0] == self.n_channels
        y = np.empty_like(x)
        self._ck(fn(self._h, _dp(x), _dp(y), x.shape[1]))
        return y

    def conv_process(self, x):
        return self._host(self._lib.cpq_conv_process, x)

    def eq_process(self, x):
        return self._host(self._lib.cpq_eq_process, x)

    def process(self, x):
        return self._host(self._lib.cpq_engine_process_block, x)

    # ---- device-pointer processing (no sync): raw HBM addresses
    def conv_process_device(self, d_in, d_out, n_samples):
        self._ck(self._lib.cpq_conv_process_device(self._h, C.c_void_p(d_in), C.c_void_p(d_out), n_samples))

    def eq_process_device(self, d_in, d_out, n_samples):
        self._ck(self._lib.cpq_eq_process_device(self._h, C.c_void_p(d_in), C.c_void_p(d_out), n_samples))

    def process_device(self, d_in, d_out, n_samples):
        self._ck(self._lib.cpq_engine_process_block_device(self._h, C.c_void_p(d_in), C.c_void_p(d_out), n_samples))

    # ---- profiling
    def profile_enable(self, on=True):
        self._ck(self._lib.cpq_profile_enable(self._h, int(on)))

    def profile_reset(self):
        self._ck(self._lib.cpq_profile_reset(self._h))

    def profile_read(self):
        out = {}
        for name, kid in K.KERNEL_IDS.items():
            n = C.c_int64()
            ms = C.c_double()
            self._ck(self._lib.cpq_profile_read(self._h, kid, C.byref(n), C.byref(ms)))
            out[name] = (n.value, ms.value)
        return out

    def eq_chain_status(self):
        """(chained launches of the EQ / output-filter cascade so far, hand-over gave up flag) -- diagnostics for tests"""
        n = C.c_uint32()
        bad = C.c_uint32()
        self._ck(self._lib.cpq_diag_eq_chain_status(self._h, C.byref(n), C.byref(bad)))
        return n.value, bad.value
