// host_design.hpp -- host-side design math (layer plan, h_eff, SVF coefficients). See host_design.cpp.
#pragma once

#include <vector>

#include "convopeq_mi355x.h"

namespace cpq {

// geometry of the time-parallel SVF kernel (svf_kernels.hip), shared with buildSvfTpTables()
#ifndef CPQ_TP_WAVES
#define CPQ_TP_WAVES 4          // waves per channel in k_svf_cascade_tp (512-sample spans of 64*W chunks; table slot 1)
#endif
constexpr int kSvfTpWaves = CPQ_TP_WAVES;
constexpr int kSvfTpLc[2] = { 4096 / (64 * kSvfTpWaves), 512 / (64 * kSvfTpWaves) };
constexpr int kSvfTpLcDoubles = 6 * 4 + 4 + 64 * 4 + 16 * 2;
// after the two per-chunk-length blocks: the matrix form of one 16-sample chunk for the MFMA path (chunk length 16 only):
// ht[32] = 15 zeros, h[0..15], 0 (zero-state impulse response; T[m][k] = ht[15 + m - k]),  e[2][16] = A^(15-k) B
constexpr int kSvfTpMfmaDoubles = 32 + 2 * 16;
constexpr int kSvfTpTableDoubles = 2 * kSvfTpLcDoubles + kSvfTpMfmaDoubles;

int    computeNucPlan(int irLen, int blockSize, bool enableDirectHead, const cpq_filter_spec* spec,
                      cpq_nuc_plan* out);
int    buildHeff(const double* ir, int irLen, int blockSize, double scale, const cpq_filter_spec* spec,
                 std::vector<double>& heff, cpq_nuc_plan* planOut);
void   spectrumFilterGains(const cpq_filter_spec& spec, int N, std::vector<double>& gains);
bool   airAbsorptionGains(const cpq_filter_spec& spec, int layer, int complexSize, std::vector<double>& gains);
void   designSvf(int type, float freq, float gainDb, float q, double sr, cpq_svf_coeffs* c);
void   defaultEqParams(cpq_eq_params* p);
double totalGainLinear(float db);
// Tables of the time-parallel SVF kernel for one band (kSvfTpTableDoubles doubles, layout TpBandTables in
// svf_kernels.hip: 2x2 powers of the state matrix and the state-to-output response).  Returns false when the state guards of the reference could
// trip for inputs / carried states below 1e9 (or the filter does not decay), i.e. the kernel must not be used.
bool   buildSvfTpTables(const cpq_svf_coeffs& c, double* out);
bool   buildBiquadTpTables(const cpq_biquad_coeffs& q, double* out);
void   designOutputFilter(int convIsLast, int hcMode, int lcMode, int lpMode, double fs, cpq_biquad_coeffs out[3]);

}  // namespace cpq
