"""ctypes binding of libconvopeq_mi355x.so (the C ABI declared in include/convopeq_mi355x.h).

The library is the product; this module only loads it.  There is no Python or CPU fallback: if the
shared object is missing or a symbol is absent, import fails loudly.
"""
import ctypes as C
import importlib.util
import os
import sys

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libconvopeq_mi355x.so")

CPQ_OK = 0
CPQ_ERR_INVALID_ARG = -1
CPQ_ERR_NO_DEVICE = -2
CPQ_ERR_OOM = -3
CPQ_ERR_DEVICE = -4
CPQ_ERR_UNSUPPORTED = -5
CPQ_ERR_NOT_READY = -6
CPQ_ALL_STREAMS = -1
CPQ_SEM_REFERENCE = 0
CPQ_SEM_EXACT = 1
CPQ_SCHED_UNIFORM = 0
CPQ_SCHED_REFERENCE_NUC = 1
CPQ_CALLS_WHOLE_BLOCKS = 0
CPQ_CALLS_ANY = 1
CPQ_PARTITION_AUTO = -1
CPQ_ORDER_CONV_THEN_EQ = 0
CPQ_ORDER_EQ_THEN_CONV = 1
KERNEL_IDS = {"k_rfft_fwd_ols": 0, "k_fdl_mac": 1, "k_fdl_mac_dcnyq": 2, "k_rfft_inv_ols": 3, "k_svf_cascade": 4,
              "k_svf_cascade_tp": 5, "k_convproc_mix": 6, "k_outfilter_cascade": 7}
CPQ_LEVEL_NUC = 0
CPQ_LEVEL_PROCESSOR = 1
CPQ_EQ_MODE_AUTO = 0
CPQ_EQ_MODE_SEQUENTIAL = 1
CPQ_PHASE_AS_IS = 0
CPQ_PHASE_MIXED = 1
CPQ_PHASE_MINIMUM = 2

c_double_p = C.POINTER(C.c_double)


class FilterSpec(C.Structure):
    _fields_ = [("sample_rate", C.c_double), ("hc_mode", C.c_int32), ("lc_mode", C.c_int32),
                ("tail_mode", C.c_int32), ("tail_enabled", C.c_int32), ("tail_start_seconds", C.c_double),
                ("tail_strength", C.c_double), ("tail_l1l2_multiplier", C.c_int32), ("reserved", C.c_int32)]

    @classmethod
    def defaults(cls, **kw):
        s = cls(48000.0, 1, 0, 1, 1, 0.085, 1.0, 8, 0)
        for k, v in kw.items():
            setattr(s, k, v)
        return s


class NucPlan(C.Structure):
    _fields_ = [("num_layers", C.c_int32), ("part_size", C.c_int32 * 3), ("offset", C.c_int32 * 3),
                ("len", C.c_int32 * 3), ("num_parts_ir", C.c_int32 * 3), ("num_parts", C.c_int32 * 3),
                ("parts_per_callback", C.c_int32 * 3), ("output_delay", C.c_int32 * 3),
                ("gain", C.c_double * 3), ("direct_taps", C.c_int32), ("latency", C.c_int32),
                ("lti_valid", C.c_int32), ("done_callback", C.c_int32 * 3), ("lag", C.c_int32 * 3),
                ("heff_len", C.c_int32)]


class SvfCoeffs(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("g", "k", "a1", "a2", "a3", "m0", "m1", "m2")]


class EqBand(C.Structure):
    _fields_ = [("frequency", C.c_float), ("gain", C.c_float), ("q", C.c_float),
                ("enabled", C.c_int32), ("type", C.c_int32), ("channel_mode", C.c_int32)]


class EqParams(C.Structure):
    _fields_ = [("bands", EqBand * 20), ("total_gain_db", C.c_float), ("agc_enabled", C.c_int32),
                ("nonlinear_saturation", C.c_float), ("filter_structure", C.c_int32)]


class BiquadCoeffs(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("b0", "b1", "b2", "a1", "a2")]


class ConvProcParams(C.Structure):
    _fields_ = [("mix", C.c_float), ("bypassed", C.c_int32), ("ir_peak_latency", C.c_int32), ("smoothing_time_sec", C.c_float)]


class IrBuffer(C.Structure):
    _fields_ = [("n_channels", C.c_int32), ("n_samples", C.c_int32), ("sample_rate", C.c_double), ("data", c_double_p)]


class IrScale(C.Structure):
    _fields_ = [("scale_factor", C.c_double), ("has_scale_factor", C.c_int32), ("additional_attenuation_db", C.c_float),
                ("peak_value", C.c_double), ("rms_value", C.c_double), ("frequency_peak_gain", C.c_double)]


class IrPrepared(C.Structure):
    _fields_ = [("ir", IrBuffer), ("scale", IrScale), ("ir_peak_latency", C.c_int32), ("reserved", C.c_int32)]


class EngineDesc(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("device", C.c_int32), ("n_streams", C.c_int32),
                ("block_size", C.c_int32), ("max_ir_len", C.c_int32), ("max_blocks_per_call", C.c_int32),
                ("semantics", C.c_int32), ("mac_tile", C.c_int32), ("sample_rate", C.c_double),
                ("partition_size", C.c_int32), ("schedule", C.c_int32), ("call_mode", C.c_int32), ("reserved", C.c_int32)]


# every symbol include/convopeq_mi355x.h declares: (restype, argtypes)
_E = C.c_void_p
SYMBOLS = {
    "cpq_abi_version": (C.c_int32, []),
    "cpq_status_string": (C.c_char_p, [C.c_int32]),
    "cpq_last_error": (C.c_char_p, [_E]),
    "cpq_nuc_plan_compute": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(FilterSpec), C.POINTER(NucPlan)]),
    "cpq_nuc_heff": (C.c_int32, [c_double_p, C.c_int32, C.c_int32, C.c_double, C.POINTER(FilterSpec), c_double_p, C.c_int32]),
    "cpq_eq_design_svf": (C.c_int32, [C.c_int32, C.c_float, C.c_float, C.c_float, C.c_double, C.POINTER(SvfCoeffs)]),
    "cpq_eq_params_default": (None, [C.POINTER(EqParams)]),
    "cpq_engine_create": (C.c_int32, [C.POINTER(EngineDesc), C.POINTER(_E)]),
    "cpq_engine_destroy": (None, [_E]),
    "cpq_engine_set_stream": (C.c_int32, [_E, C.c_void_p]),
    "cpq_engine_synchronize": (C.c_int32, [_E]),
    "cpq_engine_arena_bytes": (C.c_int64, [_E]),
    "cpq_engine_partition_size": (C.c_int32, [_E]),
    "cpq_engine_prepare": (C.c_int32, [_E, C.c_double, C.c_int32]),
    "cpq_engine_set_order": (C.c_int32, [_E, C.c_int32]),
    "cpq_host_register": (C.c_int32, [C.c_void_p, C.c_size_t]),
    "cpq_host_unregister": (C.c_int32, [C.c_void_p]),
    "cpq_conv_set_impulse": (C.c_int32, [_E, C.c_int32, c_double_p, c_double_p, C.c_int32, C.c_double, C.c_int32, C.POINTER(FilterSpec)]),
    "cpq_conv_process": (C.c_int32, [_E, c_double_p, c_double_p, C.c_int32]),
    "cpq_conv_process_device": (C.c_int32, [_E, C.c_void_p, C.c_void_p, C.c_int32]),
    "cpq_conv_reset": (C.c_int32, [_E]),
    "cpq_conv_is_ready": (C.c_int32, [_E]),
    "cpq_conv_latency": (C.c_int32, [_E]),
    "cpq_conv_get_plan": (C.c_int32, [_E, C.POINTER(NucPlan)]),
    "cpq_conv_last_got": (C.c_int32, [_E, C.c_int32]),
    "cpq_convproc_set_params": (C.c_int32, [_E, C.c_int32, C.POINTER(ConvProcParams)]),
    "cpq_convproc_process": (C.c_int32, [_E, c_double_p, c_double_p, C.c_int32]),
    "cpq_convproc_process_device": (C.c_int32, [_E, C.c_void_p, C.c_void_p, C.c_int32]),
    "cpq_convproc_delay": (C.c_int32, [_E, C.c_int32]),
    "cpq_engine_set_conv_level": (C.c_int32, [_E, C.c_int32]),
    "cpq_eq_set_params": (C.c_int32, [_E, C.c_int32, C.POINTER(EqParams)]),
    "cpq_eq_process": (C.c_int32, [_E, c_double_p, c_double_p, C.c_int32]),
    "cpq_eq_process_device": (C.c_int32, [_E, C.c_void_p, C.c_void_p, C.c_int32]),
    "cpq_eq_set_bypass": (C.c_int32, [_E, C.c_int32, C.c_int32]),
    "cpq_eq_request_band_reset": (C.c_int32, [_E, C.c_int32, C.c_uint32]),
    "cpq_eq_request_agc_reset": (C.c_int32, [_E, C.c_int32]),
    "cpq_eq_set_mode": (C.c_int32, [_E, C.c_int32]),
    "cpq_eq_reset": (C.c_int32, [_E]),
    "cpq_outfilter_design": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.POINTER(BiquadCoeffs)]),
    "cpq_outfilter_set_params": (C.c_int32, [_E, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "cpq_outfilter_process": (C.c_int32, [_E, c_double_p, c_double_p, C.c_int32]),
    "cpq_outfilter_process_device": (C.c_int32, [_E, C.c_void_p, C.c_void_p, C.c_int32]),
    "cpq_outfilter_reset": (C.c_int32, [_E]),
    "cpq_engine_enable_output_filter": (C.c_int32, [_E, C.c_int32]),
    "cpq_engine_set_gains": (C.c_int32, [_E, C.c_int32, C.c_double, C.c_double]),
    "cpq_engine_set_conv_bypass": (C.c_int32, [_E, C.c_int32]),
    "cpq_engine_process_block": (C.c_int32, [_E, c_double_p, c_double_p, C.c_int32]),
    "cpq_engine_process_block_device": (C.c_int32, [_E, C.c_void_p, C.c_void_p, C.c_int32]),
    "cpq_ir_load_wav": (C.c_int32, [C.c_char_p, C.POINTER(IrBuffer)]),
    "cpq_ir_buffer_free": (None, [C.POINTER(IrBuffer)]),
    "cpq_ir_prepare": (C.c_int32, [C.POINTER(IrBuffer), C.c_double, C.c_float, C.c_int32, C.POINTER(IrBuffer), C.c_double, C.POINTER(IrPrepared)]),
    "cpq_ir_convert_to_minimum_phase": (C.c_int32, [C.POINTER(IrBuffer), C.POINTER(IrBuffer)]),
    "cpq_ir_prepared_free": (None, [C.POINTER(IrPrepared)]),
    "cpq_ir_compute_scale_factor": (C.c_int32, [C.POINTER(c_double_p), C.c_int32, C.c_int32, C.POINTER(c_double_p), C.c_int32, C.c_int32, C.c_double, C.POINTER(IrScale)]),
    "cpq_ir_estimate_max_frequency_response_gain": (C.c_double, [C.POINTER(c_double_p), C.c_int32, C.c_int32]),
    "cpq_ir_estimate_peak_latency": (C.c_int32, [C.POINTER(c_double_p), C.c_int32, C.c_int32]),
    "cpq_profile_enable": (C.c_int32, [_E, C.c_int32]),
    "cpq_profile_reset": (C.c_int32, [_E]),
    "cpq_profile_read": (C.c_int32, [_E, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "cpq_diag_partition_fft": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, c_double_p, c_double_p, c_double_p]),
    "cpq_diag_eq_chain_status": (C.c_int32, [_E, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "cpq_kernel_name": (C.c_char_p, [C.c_int32]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """One process, one HIP runtime.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as the system one
    this library links).  If this library is loaded first it pulls in /opt/rocm's copy, a later `import torch` maps the
    bundled copy as a second runtime, and torch.cuda then reports "No HIP GPUs are available".  When a torch installation
    is present (found WITHOUT importing it) its runtime is mapped first, so both sides resolve libamdhip64.so.7 to the
    same object whatever the import order.  No torch: nothing happens and the system runtime is used."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    for root in spec.submodule_search_locations:
        cand = os.path.join(root, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load():
    """Load the shared library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C convopeq_amd/csrc` (or __graft_entry__.build()). "
            "convopeq_amd has no CPU fallback.")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)        # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
