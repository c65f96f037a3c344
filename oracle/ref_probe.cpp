// ref_probe.cpp -- ORACLE support (test infrastructure, NOT product code).
//
// Compiles the two stand-alone headers of the reference hot path from where they
// lie under /root/reference (nothing is copied into this repo):
//   src/dsp/math/FastTanhApprox.h   (only <immintrin.h>, <cstdint>, <type_traits>)
//   src/core/EQParameters.h         (only <array>)
// and exports them through a C ABI so tests can pin the oracle's restatement of
// fastTanh (A15) and the EQParameters defaults against the real reference code.
// Every other translation unit on the path needs Intel IPP, oneMKL or the
// generated JuceHeader.h, which this image lacks => unbuildable here.
//
// Output goes to oracle/_ref/ only (git-ignored, travels with gpurun).
#include "dsp/math/FastTanhApprox.h"
#include "core/EQParameters.h"

#include <cstring>

extern "C" {

double ref_fast_tanh_scalar(double x) { return convo::dsp::fastTanh<>(x); }

double ref_fast_tanh_v128(double x)
{
    const __m128d r = convo::dsp::fastTanhV128<>(_mm_set1_pd(x));
    return _mm_cvtsd_f64(r);
}

double ref_fast_tanh_softclip_scalar(double x) { return convo::dsp::fastTanh<convo::dsp::SoftClipPadéPolicy>(x); }

// flattened view of a default-constructed convo::EQParameters
struct ref_eq_band { float frequency, gain, q; int enabled; int type; int channelMode; };
struct ref_eq_params { ref_eq_band bands[20]; float totalGainDb; int agcEnabled; float nonlinearSaturation; int filterStructure; };

void ref_eq_params_default(ref_eq_params* out)
{
    const convo::EQParameters p;
    for (int i = 0; i < 20; ++i) {
        out->bands[i].frequency = p.bands[i].frequency;
        out->bands[i].gain = p.bands[i].gain;
        out->bands[i].q = p.bands[i].q;
        out->bands[i].enabled = p.bands[i].enabled ? 1 : 0;
        out->bands[i].type = p.bands[i].type;
        out->bands[i].channelMode = p.bands[i].channelMode;
    }
    out->totalGainDb = p.totalGainDb;
    out->agcEnabled = p.agcEnabled ? 1 : 0;
    out->nonlinearSaturation = p.nonlinearSaturation;
    out->filterStructure = p.filterStructure;
}

int ref_sizeof_eq_parameters() { return static_cast<int>(sizeof(convo::EQParameters)); }
int ref_sizeof_eq_band_params() { return static_cast<int>(sizeof(convo::EQBandParams)); }

}  // extern "C"
