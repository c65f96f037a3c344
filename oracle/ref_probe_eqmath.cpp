// ref_probe_eqmath.cpp -- ORACLE support (test infrastructure, NOT product code).
//
// Compiles the reference's own stand-alone EQ math test, src/tests/EQProcessorMaxGainTests.cpp (only standard headers:
// "JUCE/AudioEngine に依存しない純粋数学テスト"), unmodified from where it lies under /root/reference, and exports three
// things through a C ABI:
//   * its main() -- the reference's own checks, run here as a self-test of the build;
//   * svfToDisplayBiquad (:67-87, "実装は EQProcessor.Coefficients.cpp:347-368 と同一"): the z-domain biquad that is
//     equivalent to an SVF band with coefficients (a1, a2, a3, m0, m1, m2) -- an independent statement BY THE REFERENCE of
//     what the band recurrence of processBand / processBandStereo computes in its linear part;
//   * calcLPFSVF (:89-100): the low-pass coefficient formulas, as calcSVFCoeffs derives them.
// The functions live in an anonymous namespace of that file, so it is included textually (nothing is copied into this
// repo) with its main renamed.  Output goes to oracle/_ref/ only (git-ignored, travels with gpurun).
#define main ref_eq_math_tests_main
#include "tests/EQProcessorMaxGainTests.cpp"
#undef main

extern "C" {

// runs the reference's own assertions; 0 = all passed
int ref_eq_math_selftest() { return ref_eq_math_tests_main(); }

// svf = a1 a2 a3 m0 m1 m2  ->  bq = b0 b1 b2 a0 a1 a2
void ref_svf_to_display_biquad(const double* svf, double* bq)
{
    EQCoeffsSVF s;
    s.a1 = svf[0]; s.a2 = svf[1]; s.a3 = svf[2]; s.m0 = svf[3]; s.m1 = svf[4]; s.m2 = svf[5];
    const EQCoeffsBiquad b = svfToDisplayBiquad(s);
    bq[0] = b.b0; bq[1] = b.b1; bq[2] = b.b2; bq[3] = b.a0; bq[4] = b.a1; bq[5] = b.a2;
}

void ref_calc_lpf_svf(double freqHz, double q, double sr, double* svf)
{
    const EQCoeffsSVF s = calcLPFSVF(freqHz, q, sr);
    svf[0] = s.a1; svf[1] = s.a2; svf[2] = s.a3; svf[3] = s.m0; svf[4] = s.m1; svf[5] = s.m2;
}

double ref_biquad_magnitude_squared(const double* bq, double freqHz, double sr)
{
    EQCoeffsBiquad b;
    b.b0 = bq[0]; b.b1 = bq[1]; b.b2 = bq[2]; b.a0 = bq[3]; b.a1 = bq[4]; b.a2 = bq[5];
    return getMagnitudeSquared(b, freqHz, sr);
}

}  // extern "C"
