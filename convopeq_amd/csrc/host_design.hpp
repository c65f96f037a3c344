// host_design.hpp -- host-side design math (layer plan, h_eff, SVF coefficients). See host_design.cpp.
#pragma once

#include <vector>

#include "convopeq_mi355x.h"

namespace cpq {

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
