"""convopeq_amd -- MI355X-native (gfx950) batched drop-in for ConvoPeq's convolver + 20-band EQ hot path.

The product is convopeq_amd/libconvopeq_mi355x.so (hand-written HIP kernels behind the C ABI in
include/convopeq_mi355x.h).  Importing this package loads that library and fails loudly if it is missing.
"""
from . import _capi
from ._capi import (CPQ_ALL_STREAMS, CPQ_EQ_MODE_AUTO, CPQ_EQ_MODE_SEQUENTIAL, CPQ_LEVEL_NUC, CPQ_LEVEL_PROCESSOR, CPQ_ORDER_CONV_THEN_EQ, CPQ_ORDER_EQ_THEN_CONV, CPQ_SCHED_REFERENCE_NUC, CPQ_SCHED_UNIFORM, CPQ_SEM_EXACT, CPQ_CALLS_ANY, CPQ_CALLS_WHOLE_BLOCKS, CPQ_PARTITION_AUTO,
                    CPQ_SEM_REFERENCE, EqParams, FilterSpec, NucPlan, SvfCoeffs)
from .engine import (BatchedEngine, CpqError, design_svf, eq_params_default, ir_compute_scale_factor, ir_convert_to_minimum_phase,
                     ir_estimate_max_frequency_response_gain, ir_estimate_peak_latency, ir_load_wav, ir_prepare, nuc_heff,
                     nuc_plan, outfilter_design)

_capi.load()   # no fallback: ImportError if the HIP library is absent

__all__ = ["BatchedEngine", "CpqError", "design_svf", "eq_params_default", "nuc_heff", "nuc_plan", "outfilter_design", "ir_load_wav", "ir_prepare", "ir_compute_scale_factor", "ir_convert_to_minimum_phase",
           "ir_estimate_max_frequency_response_gain", "ir_estimate_peak_latency",
           "EqParams", "FilterSpec", "NucPlan", "SvfCoeffs", "CPQ_ALL_STREAMS", "CPQ_SEM_REFERENCE",
           "CPQ_SEM_EXACT", "CPQ_CALLS_ANY", "CPQ_CALLS_WHOLE_BLOCKS", "CPQ_PARTITION_AUTO", "CPQ_SCHED_UNIFORM", "CPQ_SCHED_REFERENCE_NUC", "CPQ_ORDER_CONV_THEN_EQ", "CPQ_ORDER_EQ_THEN_CONV", "CPQ_EQ_MODE_AUTO",
           "CPQ_EQ_MODE_SEQUENTIAL", "CPQ_LEVEL_NUC", "CPQ_LEVEL_PROCESSOR"]
