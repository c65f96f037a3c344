// ref_probe_eqbound.cpp -- ORACLE support (test infrastructure, NOT product code).
//
// Compiles the reference's stand-alone EQ bound benchmark, src/tests/EQBoundExcessBenchmark.cpp (standard headers only),
// unmodified from where it lies under /root/reference, and exports its Audio-EQ-Cookbook designers through a C ABI:
//   calcPeakingBiquad / calcLowShelfBiquad / calcHighShelfBiquad (:188-245)
// -- the reference's own statement of the peaking / shelving responses its EQ aims at, independent of this repo's reading
// of calcSVFCoeffs (src/eqprocessor/EQProcessor.Coefficients.cpp:431-560).  The functions live in an anonymous namespace
// of that file, so it is included textually (nothing is copied into this repo) with its main renamed; the benchmark's
// main() itself is not run.  Output goes to oracle/_ref/ only (git-ignored, travels with gpurun).
#define main ref_eq_bound_benchmark_main
#include "tests/EQBoundExcessBenchmark.cpp"
#undef main

extern "C" {

// type: 0 low shelf, 1 peaking, 2 high shelf (this repo's band type codes);  bq = b0 b1 b2 a0 a1 a2
int ref_rbj_biquad(int type, double freqHz, double gainDb, double q, double sr, double* bq)
{
    EQCoeffsBiquad b;
    if (type == 0)      b = calcLowShelfBiquad(freqHz, gainDb, q, sr);
    else if (type == 1) b = calcPeakingBiquad(freqHz, gainDb, q, sr);
    else if (type == 2) b = calcHighShelfBiquad(freqHz, gainDb, q, sr);
    else return -1;
    bq[0] = b.b0; bq[1] = b.b1; bq[2] = b.b2; bq[3] = b.a0; bq[4] = b.a1; bq[5] = b.a2;
    return 0;
}

}  // extern "C"
