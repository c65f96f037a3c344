// host_design.hpp -- host-side design math (layer plan, h_eff, SVF coefficients). See host_design.cpp.
#pragma once

#include <vector>

#include "convopeq_mi355x.h"

namespace cpq {

// geometry of the time-parallel SVF kernels (svf_kernels.hip), shared with buildSvfTpTables()
// chunk lengths with a table block each: 16 (spans of 8192 / 1024 x waves samples: k_svf_cascade_tpv), 8 (spans below 1024
// samples on one or two waves: k_svf_cascade_short)
constexpr int kSvfTpLcCount = 2;
constexpr int kSvfTpLc[kSvfTpLcCount] = { 16, 8 };
constexpr int kSvfTpLcDoubles = 6 * 4 + 4 + 64 * 4 + 16 * 2;
// after the two per-chunk-length blocks: the matrix form of one 16-sample chunk for the MFMA path (chunk length 16 only):
// ht[32] = 15 zeros, h[0..15], 0 (zero-state impulse response; T[m][k] = ht[15 + m - k]),  e[2][16] = A^(15-k) B
constexpr int kSvfTpMfmaDoubles = 32 + 2 * 16;
constexpr int kSvfTpTableDoubles = kSvfTpLcCount * kSvfTpLcDoubles + kSvfTpMfmaDoubles;

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
