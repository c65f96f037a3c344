"""ctypes view of oracle/libcpq_oracle.so (CPU oracle: test infrastructure only).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = None
_REF = None

c_double_p = C.POINTER(C.c_double)


class FilterSpec(C.Structure):
    _fields_ = [("sampleRate", C.c_double), ("hcMode", C.c_int), ("lcMode", C.c_int),
                ("tailMode", C.c_int), ("tailEnabled", C.c_int), ("tailStartSeconds", C.c_double),
                ("tailStrength", C.c_double), ("tailL1L2Multiplier", C.c_int),
                ("applySpectrumFilter", C.c_int)]

    @classmethod
    def defaults(cls, **kw):
        s = cls(48000.0, 1, 0, 1, 1, 0.085, 1.0, 8, 0)
        for k, v in kw.items():
            setattr(s, k, v)
        return s


class NucPlan(C.Structure):
    _fields_ = [("numLayers", C.c_int), ("partSize", C.c_int * 3), ("offset", C.c_int * 3),
                ("len", C.c_int * 3), ("numPartsIR", C.c_int * 3), ("numParts", C.c_int * 3),
                ("partsPerCallback", C.c_int * 3), ("outputDelay", C.c_int * 3),
                ("gain", C.c_double * 3), ("directTaps", C.c_int), ("latency", C.c_int),
                ("ltiValid", C.c_int), ("doneCallback", C.c_int * 3), ("lag", C.c_int * 3)]


class SvfCoeffs(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("g", "k", "a1", "a2", "a3", "m0", "m1", "m2")]


class Biquad(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("b0", "b1", "b2", "a1", "a2")]


class EqBand(C.Structure):
    _fields_ = [("frequency", C.c_float), ("gain", C.c_float), ("q", C.c_float),
                ("enabled", C.c_int), ("type", C.c_int), ("channelMode", C.c_int)]


class EqParams(C.Structure):
    _fields_ = [("bands", EqBand * 20), ("totalGainDb", C.c_float), ("agcEnabled", C.c_int),
                ("nonlinearSaturation", C.c_float), ("filterStructure", C.c_int)]


def build(force=False):
    so = os.path.join(ORACLE_DIR, "libcpq_oracle.so")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(
            os.path.join(ORACLE_DIR, "cpq_oracle.c")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libcpq_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    L = C.CDLL(build())
    L.orc_splitmix64.restype = C.c_uint64
    L.orc_splitmix64.argtypes = [C.c_uint64]
    L.orc_rand_pm1.restype = C.c_double
    L.orc_rand_pm1.argtypes = [C.c_uint64] * 4
    L.orc_gen_pcm.argtypes = [c_double_p, C.c_int64, C.c_uint64, C.c_int, C.c_int, C.c_int64]
    L.orc_gen_ir.argtypes = [c_double_p, C.c_int, C.c_uint64, C.c_int, C.c_int]
    L.orc_fft_create.restype = C.c_void_p
    L.orc_fft_create.argtypes = [C.c_int]
    L.orc_fft_destroy.argtypes = [C.c_void_p]
    L.orc_fft_fwd_ccs.argtypes = [C.c_void_p, c_double_p, c_double_p]
    L.orc_fft_inv_ccs.argtypes = [C.c_void_p, c_double_p, c_double_p]
    L.orc_nuc_plan_compute.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(FilterSpec), C.POINTER(NucPlan)]
    L.orc_nuc_heff.argtypes = [c_double_p, C.c_int, C.c_int, C.c_double, C.POINTER(FilterSpec), c_double_p, C.c_int]
    L.orc_nuc_create.restype = C.c_void_p
    L.orc_nuc_destroy.argtypes = [C.c_void_p]
    L.orc_nuc_set_impulse.argtypes = [C.c_void_p, c_double_p, C.c_int, C.c_int, C.c_double, C.c_int,
                                      C.POINTER(FilterSpec)]
    L.orc_nuc_add.argtypes = [C.c_void_p, c_double_p, C.c_int]
    L.orc_nuc_get.argtypes = [C.c_void_p, c_double_p, C.c_int]
    L.orc_nuc_reset.argtypes = [C.c_void_p]
    L.orc_nuc_latency.argtypes = [C.c_void_p]
    L.orc_nuc_get_plan.argtypes = [C.c_void_p, C.POINTER(NucPlan)]
    L.orc_nuc_run.argtypes = [C.c_void_p, c_double_p, c_double_p, C.c_int, C.c_int]
    L.orc_direct_conv_at.argtypes = [c_double_p, C.c_int64, c_double_p, C.c_int,
                                     C.POINTER(C.c_int64), C.c_int, c_double_p]
    L.orc_eq_params_default.argtypes = [C.POINTER(EqParams)]
    L.orc_svf_design.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_double, C.POINTER(SvfCoeffs)]
    L.orc_fast_tanh_scalar.restype = C.c_double
    L.orc_fast_tanh_scalar.argtypes = [C.c_double]
    L.orc_fast_tanh_v128.restype = C.c_double
    L.orc_fast_tanh_v128.argtypes = [C.c_double]
    L.orc_svf_band_stereo_lane.argtypes = [c_double_p, C.c_int64, C.POINTER(SvfCoeffs), c_double_p, C.c_double]
    L.orc_svf_band_mono.argtypes = [c_double_p, C.c_int64, C.POINTER(SvfCoeffs), c_double_p, C.c_double]
    L.orc_eq_process_stereo.argtypes = [c_double_p, c_double_p, C.c_int64, C.c_int, C.POINTER(EqParams),
                                        C.c_double, c_double_p]
    L.orc_eq_process_stereo_ex.argtypes = [c_double_p, c_double_p, C.c_int64, C.c_int, C.POINTER(EqParams),
                                           C.c_double, c_double_p, C.c_int]
    L.orc_outfilter_design.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(Biquad)]
    L.orc_biquad_df2t_lane.argtypes = [c_double_p, C.c_int64, C.POINTER(Biquad), c_double_p]
    L.orc_outfilter_process_stereo.argtypes = [c_double_p, c_double_p, C.c_int64, C.POINTER(Biquad), c_double_p]
    L.orc_equal_power_sin.restype = C.c_double
    L.orc_equal_power_sin.argtypes = [C.c_double]
    _LIB = L
    return L


def ref_probe():
    """oracle/_ref/libcpq_ref_probe.so: the reference's own stand-alone headers, or None."""
    global _REF
    if _REF is not None:
        return _REF
    so = os.path.join(ORACLE_DIR, "_ref", "libcpq_ref_probe.so")
    if not os.path.exists(so):
        if os.path.isdir("/root/reference/src"):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
        if not os.path.exists(so):
            return None
    R = C.CDLL(so)
    for n in ("ref_fast_tanh_scalar", "ref_fast_tanh_v128", "ref_fast_tanh_softclip_scalar"):
        getattr(R, n).restype = C.c_double
        getattr(R, n).argtypes = [C.c_double]
    R.ref_eq_params_default.argtypes = [C.POINTER(EqParams)]
    if hasattr(R, "ref_svf_to_display_biquad"):       # the reference's pure-math EQ test (oracle/ref_probe_eqmath.cpp)
        R.ref_eq_math_selftest.restype = C.c_int
        R.ref_svf_to_display_biquad.argtypes = [c_double_p, c_double_p]
        R.ref_calc_lpf_svf.argtypes = [C.c_double, C.c_double, C.c_double, c_double_p]
        R.ref_biquad_magnitude_squared.restype = C.c_double
        R.ref_biquad_magnitude_squared.argtypes = [c_double_p, C.c_double, C.c_double]
    _REF = R
    return R


def dp(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


def ref_svf_to_display_biquad(svf6):
    """(b, a) of the biquad the reference's own svfToDisplayBiquad assigns to SVF coefficients (a1 a2 a3 m0 m1 m2)."""
    s = np.ascontiguousarray(svf6, dtype=np.float64)
    out = np.empty(6)
    ref_probe().ref_svf_to_display_biquad(dp(s), dp(out))
    return out[:3].copy(), out[3:].copy()


def ref_rbj_biquad(btype, freq, gain_db, q, sr):
    """(b, a) of the reference's own cookbook designers (src/tests/EQBoundExcessBenchmark.cpp:188-245, compiled unmodified via
    oracle/ref_probe_eqbound.cpp): btype 0 low shelf, 1 peaking, 2 high shelf; or None when the probe lacks them."""
    R = ref_probe()
    if R is None or not hasattr(R, "ref_rbj_biquad"):
        return None
    R.ref_rbj_biquad.restype = C.c_int
    R.ref_rbj_biquad.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]
    out = np.empty(6)
    if R.ref_rbj_biquad(int(btype), float(freq), float(gain_db), float(q), float(sr), dp(out)) != 0:
        return None
    return out[:3].copy(), out[3:].copy()


def ref_calc_lpf_svf(freq, q, sr):
    out = np.empty(6)
    ref_probe().ref_calc_lpf_svf(freq, q, sr, dp(out))
    return out


# ------------------------------------------------------------------ helpers
def gen_pcm(n, seed=0xC0FFEE, stream=0, channel=0, start=0):
    x = np.empty(n, dtype=np.float64)
    lib().orc_gen_pcm(dp(x), n, seed, stream, channel, start)
    return x


def gen_ir(length, seed=0x1257, stream=0, channel=0):
    h = np.empty(length, dtype=np.float64)
    lib().orc_gen_ir(dp(h), length, seed, stream, channel)
    return h


def plan(ir_len, block, direct=False, spec=None):
    p = NucPlan()
    rc = lib().orc_nuc_plan_compute(ir_len, block, int(direct), C.byref(spec) if spec is not None else None,
                                    C.byref(p))
    if rc != 0:
        raise ValueError("plan failed")
    return p


def heff(ir, block, scale=1.0, spec=None):
    ir = np.ascontiguousarray(ir, dtype=np.float64)
    sp = C.byref(spec) if spec is not None else None
    need = lib().orc_nuc_heff(dp(ir), len(ir), block, scale, sp, None, 0)
    out = np.zeros(need, dtype=np.float64)
    lib().orc_nuc_heff(dp(ir), len(ir), block, scale, sp, dp(out), need)
    return out


class Nuc:
    """Stateful restatement of MKLNonUniformConvolver (one mono channel)."""

    def __init__(self):
        self._h = lib().orc_nuc_create()

    def close(self):
        if self._h:
            lib().orc_nuc_destroy(self._h)
            self._h = None

    __del__ = close

    def set_impulse(self, ir, block, scale=1.0, direct=False, spec=None):
        ir = np.ascontiguousarray(ir, dtype=np.float64)
        return bool(lib().orc_nuc_set_impulse(self._h, dp(ir), len(ir), block, scale, int(direct),
                                              C.byref(spec) if spec is not None else None))

    def add(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        lib().orc_nuc_add(self._h, dp(x), len(x))

    def get(self, n):
        y = np.empty(n, dtype=np.float64)
        got = lib().orc_nuc_get(self._h, dp(y), n)
        return y, got

    def run(self, x, block):
        x = np.ascontiguousarray(x, dtype=np.float64)
        nb = len(x) // block
        y = np.empty(nb * block, dtype=np.float64)
        lib().orc_nuc_run(self._h, dp(x), dp(y), block, nb)
        return y

    def reset(self):
        lib().orc_nuc_reset(self._h)

    def latency(self):
        return lib().orc_nuc_latency(self._h)

    def plan(self):
        p = NucPlan()
        lib().orc_nuc_get_plan(self._h, C.byref(p))
        return p


def direct_conv_at(x, h, idx):
    x = np.ascontiguousarray(x, dtype=np.float64)
    h = np.ascontiguousarray(h, dtype=np.float64)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    y = np.empty(len(idx), dtype=np.float64)
    lib().orc_direct_conv_at(dp(x), len(x), dp(h), len(h), idx.ctypes.data_as(C.POINTER(C.c_int64)), len(idx), dp(y))
    return y


def svf_design(btype, freq, gain_db, q, sr):
    c = SvfCoeffs()
    lib().orc_svf_design(btype, freq, gain_db, q, sr, C.byref(c))
    return c


def eq_params_default():
    p = EqParams()
    lib().orc_eq_params_default(C.byref(p))
    return p


BENCH_GAINS_DB = [3, -2, 4, -3, 2, -4, 3, -2, 1.5, -1.5, 2, -2, 3, -3, 1, -1, 2, -2, 1, -1]
DEFAULT_FREQS = [25.0, 40.0, 63.0, 100.0, 160.0, 250.0, 400.0, 630.0, 1000.0, 1600.0,
                 2500.0, 4000.0, 6300.0, 10000.0, 11000.0, 12500.0, 14000.0, 16500.0, 18000.0, 19500.0]


def eq_params_bench(saturation=0.2):
    """SURVEY.md 8(d) EQ bench preset: EQProcessor::DEFAULT_FREQS, Q 1.41, alternating gains,
    band 0 LowShelf, band 19 HighShelf, rest Peaking, Stereo, Serial, AGC off, 0 dB total."""
    p = eq_params_default()
    for i in range(20):
        b = p.bands[i]
        b.frequency = DEFAULT_FREQS[i]
        b.gain = BENCH_GAINS_DB[i]
        b.q = 1.41
        b.enabled = 1
        b.type = 0 if i == 0 else (2 if i == 19 else 1)
        b.channelMode = 0
    p.nonlinearSaturation = saturation
    return p


def eq_process_stereo(xl, xr, params, sr=48000.0, block=512, state=None):
    yl = np.array(xl, dtype=np.float64, copy=True)
    yr = np.array(xr, dtype=np.float64, copy=True)
    if state is None:
        state = np.zeros(2 * 20 * 2 + 3 + 5 + 2 * 20 * 2, dtype=np.float64)
    lib().orc_eq_process_stereo(dp(yl), dp(yr), len(yl), block, C.byref(params), sr, dp(state))
    return yl, yr, state


class LinearRamp:
    """convo::LinearRamp (src/DspNumericPolicy.h:319-421)."""

    def __init__(self, value, sr, time_sec):
        self.current = self.target = float(value)
        self.step, self.remaining = 0.0, 0
        self.total = max(1, int(sr * time_sec + 0.5))

    def set_target(self, v):
        if v == self.target:
            return
        self.target = v
        steps = self.remaining if self.remaining > 0 else self.total
        self.step = (self.target - self.current) / float(steps)
        self.remaining = steps

    def next(self):
        if self.remaining <= 0:
            return self.current
        self.current += self.step
        self.remaining -= 1
        if self.remaining <= 0:
            self.current = self.target
        return self.current


class EqWithBypass:
    """EQProcessor as DSPCore drives it (src/audioengine/AudioEngine.Processing.DSPCoreDouble.cpp:384-413): per callback
    setBypassFromRT(requested), then process(block, params, cache) -- which falls back to the basic process(block) while
    the bypass is in effect or fading (src/eqprocessor/EQProcessor.Processing.cpp:1023-1034) -- or process(block)
    directly while the bypass is requested.  The basic path (:486-1016) fades between the processed and the dry block
    over BYPASS_FADE_TIME_SEC = 5 ms, returns early once the fade-out is complete (states frozen) and clears every
    filter state when the bypass is released."""

    def __init__(self, params, sr=48000.0, block=512):
        self.p, self.sr, self.block = params, sr, block
        self.state = np.zeros(168)
        # prepareToPlay: smoothTotalGain.setCurrentAndTargetValue(gain at that time) (src/eqprocessor/EQProcessor.Core.cpp:765)
        # -- set here rather than at the first processed block, which a bypass may put off
        gdb = float(params.totalGainDb)
        g0 = 10.0 ** (gdb * 0.05) if gdb > -100.0 else 0.0
        self.state[83:88] = [1.0, g0, g0, 0.0, 0.0]
        self.fade = LinearRamp(1.0, sr, 0.005)
        self.effective = False
        self.pending = 0                                    # rtDeferredBandResetMask
        self.agc_reset = False                              # agcResetSerial moved

    def request_band_reset(self, mask=0xFFFFFFFF):
        self.pending |= mask

    def request_agc_reset(self):
        self.agc_reset = True

    def set_total_gain_db(self, db, before_first_block=False):
        """storeTotalGainDb; before the first block (prepareToPlay) the ramp is set to the value, afterwards it runs."""
        self.p.totalGainDb = db
        if before_first_block:
            gdb = float(self.p.totalGainDb)
            g0 = 10.0 ** (gdb * 0.05) if gdb > -100.0 else 0.0
            self.state[83:88] = [1.0, g0, g0, 0.0, 0.0]

    def sync(self, requested):
        """prepareToPlay / reset with the bypass already requested (setBypass on the message thread): the fade is set, not
        run (src/eqprocessor/EQProcessor.Core.cpp:285-288, 800-802)."""
        self.effective = bool(requested)
        self.fade.current = self.fade.target = 0.0 if requested else 1.0
        self.fade.step, self.fade.remaining = 0.0, 0

    def callback(self, xl, xr, requested):
        L = lib()
        target = 0.0 if requested else 1.0
        if abs(self.fade.target - target) > 1.0e-12:
            if not requested and self.effective:
                self.pending |= 0xFFFFFFFF                  # deferred reset of all bands (:507-511)
                self.effective = False
            self.fade.set_target(target)
        transition = self.fade.remaining > 0
        if requested and not self.effective and not transition:
            self.effective = True
        if requested and self.effective and not transition:
            return xl.copy(), xr.copy()
        basic = requested or self.effective or transition
        if self.agc_reset:                                  # :586-593 / :1070-1077: envelopes 0, gain 1
            self.state[80:83] = 0.0
            self.agc_reset = False
        if self.pending:                                    # :565-568, :603-624 (parameter path: silence only, :1083-1112)
            silent = not (np.any(np.abs(xl) > 1.0e-8) or np.any(np.abs(xr) > 1.0e-8))
            if basic or silent:
                st = self.state
                if self.pending == 0xFFFFFFFF:
                    st[0:80] = 0.0
                    st[88:168] = 0.0
                else:
                    for b in range(20):
                        if self.pending & (1 << b):
                            for base in (0, 40, 88, 128):   # L, R, Mid, Side
                                st[base + 2 * b:base + 2 * b + 2] = 0.0
                self.pending = 0
        yl, yr = xl.copy(), xr.copy()
        L.orc_eq_process_stereo_ex(dp(yl), dp(yr), len(yl), self.block, C.byref(self.p), self.sr, dp(self.state),
                                   int(basic))
        if transition:
            for i in range(len(yl)):
                g = self.fade.next()
                d = 1.0 - g
                yl[i] = yl[i] * g + xl[i] * d
                yr[i] = yr[i] * g + xr[i] * d
            if self.fade.remaining <= 0:
                self.effective = bool(requested)
        return yl, yr


def convproc_steady(ir, x, block, mix=1.0, bypassed=False, ir_peak_latency=0, scale=1.0, spec=None):
    """Steady-state restatement of ConvolverProcessor::process for ONE channel over the whole signal x
    (src/convolver/ConvolverProcessor.Runtime.cpp:209-810): dry delay line of algorithmLatency + irPeakLatency,
    wet sanitise (NaN / Inf / |x| >= 1e300 -> 0, :50-60), out = wet*wetG + dry*dryG with the Taylor
    equalPowerSin (:26-31, :675-676); mix <= 0.001 -> delayed dry only, bypass -> delayed dry only (convolver not run)."""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = len(x)
    mixd = float(np.float32(mix))
    delay = min(block, 524288) + min(max(0, ir_peak_latency), 2097152)
    dry = np.zeros(n)
    if delay < n:
        dry[delay:] = x[:n - delay]
    if bypassed or not (mixd > 0.001):
        return dry
    nuc = Nuc()
    assert nuc.set_impulse(ir, block, scale=scale, spec=spec)
    wet = nuc.run(x, block)
    nuc.close()
    bad = ~(np.abs(wet) < 1.0e300)
    wet = np.where(bad, 0.0, wet)
    wet_g = L.orc_equal_power_sin(mixd) * 1.0
    dry_g = L.orc_equal_power_sin(1.0 - mixd) if mixd < 0.999 else 0.0
    return (wet * wet_g) + (dry * dry_g)


def convproc_mix_schedule(ir, x, block, mix_per_callback, sr=48000.0, smoothing_time=0.1, ir_peak_latency=0,
                          initial_mix=None):
    """ConvolverProcessor::process for ONE channel with a mix value per callback (block): the LinearRamp mixSmoother
    (src/DspNumericPolicy.h:319-421; 100 ms default) moves to a new mix over smoothing_time, and every callback that
    STARTS while it runs is mixed with per-sample gains equalPowerSin(mix_i) / equalPowerSin(1 - mix_i)
    (src/convolver/ConvolverProcessor.Runtime.cpp:366-375, 591-607, 727-735); other callbacks use the steady gains, or
    copy the delayed dry signal when mix <= 0.001 (:573-585).  The convolver is run on every callback here (callers keep
    mix > 0.001 until a final dry-only stretch whose wet signal is not used)."""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = len(x)
    assert n == block * len(mix_per_callback)
    delay = min(block, 524288) + min(max(0, ir_peak_latency), 2097152)
    dry = np.zeros(n)
    if delay < n:
        dry[delay:] = x[:n - delay]
    nuc = Nuc()
    assert nuc.set_impulse(ir, block)
    wet = nuc.run(x, block)
    nuc.close()
    wet = np.where(~(np.abs(wet) < 1.0e300), 0.0, wet)
    total = max(1, int(sr * smoothing_time + 0.5))
    m0 = float(np.float32(mix_per_callback[0] if initial_mix is None else initial_mix))
    cur, tgt, step, rem = m0, m0, 0.0, 0
    out = np.empty(n)
    eps = L.orc_equal_power_sin
    for cb, mix in enumerate(mix_per_callback):
        mixd = float(np.float32(mix))
        if abs(tgt - mixd) > 1.0e-5 and mixd != tgt:          # setTargetValue
            tgt = mixd
            steps = rem if rem > 0 else total
            step = (tgt - cur) / float(steps)
            rem = steps
        lo, hi = cb * block, (cb + 1) * block
        if rem > 0:                                           # isSmoothing at the start of the callback
            for i in range(lo, hi):
                if rem > 0:
                    cur += step
                    rem -= 1
                    if rem <= 0:
                        cur = tgt
                out[i] = (wet[i] * (eps(cur) * 1.0)) + (dry[i] * eps(1.0 - cur))
        elif not (mixd > 0.001):
            out[lo:hi] = dry[lo:hi]
        else:
            wg = eps(mixd) * 1.0
            dg = eps(1.0 - mixd) if mixd < 0.999 else 0.0
            out[lo:hi] = (wet[lo:hi] * wg) + (dry[lo:hi] * dg)
    return out


def convproc_latency_schedule(ir, x, block, peak_per_callback, mix=0.6, sr=48000.0, direct_head=False):
    """ConvolverProcessor::process for ONE channel while irPeakLatency changes between callbacks: the latency compensation
    of src/convolver/ConvolverProcessor.Runtime.cpp:263-290 (a total latency that moved by >= 2 samples starts, unless
    one is running, a 20 ms cross-fade from the delay in use to the new one) and :394-540 (callbacks that start while
    crossfadeGain runs read the delay line at both delays and blend new * g + old * (1 - g) until the ramp ends).
    prepareToPlay starts from latency + irLatency (Lifecycle.cpp:377-388)."""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    n = len(x)
    assert n == block * len(peak_per_callback)
    nuc = Nuc()
    assert nuc.set_impulse(ir, block, direct=direct_head)
    wet = nuc.run(x, block)
    nuc.close()
    wet = np.where(~(np.abs(wet) < 1.0e300), 0.0, wet)
    mixd = float(np.float32(mix))
    wg = L.orc_equal_power_sin(mixd) * 1.0
    dg = L.orc_equal_power_sin(1.0 - mixd) if mixd < 0.999 else 0.0

    def dry_at(i, d):
        j = i - int(d)
        return x[j] if j >= 0 else 0.0

    lat_cur = lat_tgt = old = float(block + peak_per_callback[0])
    fade = LinearRamp(1.0, sr, 0.02)
    out = np.empty(n)
    for cb, peak in enumerate(peak_per_callback):
        total = float((0 if direct_head else block) + peak)
        if abs(lat_tgt - total) >= 2.0 and fade.remaining <= 0:
            old = lat_cur
            fade.current = fade.target = 0.0
            fade.set_target(1.0)
            lat_tgt = total
        lo = cb * block
        dry = np.empty(block)
        if fade.remaining > 0:
            k = 0
            while k < block:
                g = fade.next()
                dry[k] = dry_at(lo + k, lat_tgt) * g + dry_at(lo + k, old) * (1.0 - g)
                k += 1
                if fade.remaining <= 0:
                    break
            for i in range(k, block):
                dry[i] = dry_at(lo + i, lat_tgt)
            if fade.remaining <= 0:
                lat_cur = lat_tgt
                old = lat_cur
        else:
            d = int(lat_cur + 0.5)
            for i in range(block):
                dry[i] = dry_at(lo + i, d)
        out[lo:lo + block] = (wet[lo:lo + block] * wg) + (dry * dg)
    return out


class ConvProcStream:
    """ConvolverProcessor::process for one stereo stream, callback by callback, with everything that can move on a live
    stream at once: the mix ramp (src/convolver/ConvolverProcessor.Runtime.cpp:340-375, 591-607), the latency
    compensation with its cross-fade (:263-290, 394-540) and the steady paths (:573-585, :611-676).  The two channels
    share the smoothers, as in the reference."""

    def __init__(self, ir_l, ir_r, block, mix, peak, sr=48000.0, smoothing_time=0.1, scale=1.0, spec=None, latency=None):
        # latency: getLatency() of the convolver = its layer-0 partition; equals the block for power-of-two blocks (default),
        # nextPow2(max(block, 64)) for any other call quantum.  A callback may be shorter than the block (the last one of a
        # ragged host call): its length is that of the arrays passed to callback().
        self.block, self.L = block, lib()
        self.base = block if latency is None else latency
        self.nucs = [Nuc(), Nuc()]
        assert self.nucs[0].set_impulse(ir_l, block, scale=scale, spec=spec)
        assert self.nucs[1].set_impulse(ir_r, block, scale=scale, spec=spec)
        self.mix = LinearRamp(float(np.float32(mix)), sr, smoothing_time)
        self.fade = LinearRamp(1.0, sr, 0.02)
        self.lat_cur = self.lat_tgt = self.old = float(self.base + peak)
        self.hist = [np.zeros(0), np.zeros(0)]

    def _dry(self, ch, i, d):
        j = i - int(d)
        return self.hist[ch][j] if j >= 0 else 0.0

    def callback(self, xl, xr, mix, peak):
        B, eps = len(xl), self.L.orc_equal_power_sin
        x = [np.ascontiguousarray(xl, dtype=np.float64), np.ascontiguousarray(xr, dtype=np.float64)]
        lo = len(self.hist[0])
        for ch in range(2):
            self.hist[ch] = np.concatenate([self.hist[ch], x[ch]])
        total = float(self.base + peak)
        if abs(self.lat_tgt - total) >= 2.0 and self.fade.remaining <= 0:
            self.old = self.lat_cur
            self.fade.current = self.fade.target = 0.0
            self.fade.set_target(1.0)
            self.lat_tgt = total
        mixd = float(np.float32(mix))
        if abs(self.mix.target - mixd) > 1.0e-5:
            self.mix.set_target(mixd)
        smoothing = self.mix.remaining > 0
        needs_conv = smoothing or mixd > 0.001
        # dry block
        dry = [np.empty(B), np.empty(B)]
        if self.fade.remaining > 0:
            gains = []
            while len(gains) < B:
                gains.append(self.fade.next())
                if self.fade.remaining <= 0:
                    break
            for ch in range(2):
                for i in range(B):
                    new = self._dry(ch, lo + i, self.lat_tgt)
                    if i < len(gains):
                        new = new * gains[i] + self._dry(ch, lo + i, self.old) * (1.0 - gains[i])
                    dry[ch][i] = new
            if self.fade.remaining <= 0:
                self.lat_cur = self.lat_tgt
                self.old = self.lat_cur
        else:
            d = int(self.lat_cur + 0.5)
            for ch in range(2):
                for i in range(B):
                    dry[ch][i] = self._dry(ch, lo + i, d)
        if not needs_conv:
            return dry[0], dry[1]
        wet = []
        for ch in range(2):                                 # Add + Get of one callback (StereoConvolver::process)
            self.nucs[ch].add(x[ch])
            wet.append(self.nucs[ch].get(B)[0])
        wet = [np.where(~(np.abs(w) < 1.0e300), 0.0, w) for w in wet]
        out = [np.empty(B), np.empty(B)]
        if smoothing:
            for i in range(B):
                m = self.mix.next()
                wg, dg = eps(m) * 1.0, eps(1.0 - m)
                for ch in range(2):
                    out[ch][i] = (wet[ch][i] * wg) + (dry[ch][i] * dg)
        else:
            wg = eps(mixd) * 1.0
            dg = eps(1.0 - mixd) if mixd < 0.999 else 0.0
            for ch in range(2):
                out[ch] = (wet[ch] * wg) + (dry[ch] * dg)
        return out[0], out[1]


def outfilter_design(conv_is_last, hc_mode=1, lc_mode=0, lp_mode=1, sr=48000.0):
    out = (Biquad * 3)()
    lib().orc_outfilter_design(int(conv_is_last), hc_mode, lc_mode, lp_mode, sr, out)
    return out


def outfilter_process_stereo(xl, xr, coeffs, state=None):
    yl = np.array(xl, dtype=np.float64, copy=True)
    yr = np.array(xr, dtype=np.float64, copy=True)
    if state is None:
        state = np.zeros(12, dtype=np.float64)
    lib().orc_outfilter_process_stereo(dp(yl), dp(yr), len(yl), coeffs, dp(state))
    return yl, yr, state
