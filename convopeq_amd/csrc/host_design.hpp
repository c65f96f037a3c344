// host_design.hpp -- host-side design math (layer plan, h_eff, SVF coefficients). See host_design.cpp.
#pragma once

#include <vector>

#include "convopeq_mi355x.h"

namespace cpq {

int    computeNucPlan(int irLen, int blockSize, bool enableDirectHead, const cpq_filter_spec* spec,
                      cpq_nuc_plan* out);
int    buildHeff(const double* ir, int irLen, int blockSize, double scale, const cpq_filter_spec* spec,
                 std::vector<double>& heff, cpq_nuc_plan* planOut);
void   designSvf(int type, float freq, float gainDb, float q, double sr, cpq_svf_coeffs* c);
void   defaultEqParams(cpq_eq_params* p);
double totalGainLinear(float db);

}  // namespace cpq
