#!/usr/bin/env python3
"""usage: make_wg_trace.py <svf_kernels.hip> <out.hip>
Diagnostic variant of the EQ span kernel (NOT part of the product build): every workgroup of k_svf_cascade_tpv records its
start / end time (s_memrealtime, constant 100 MHz), its shader-clock count over the same interval (s_memtime), the XCD and
CU it ran on, and the time it finished each span (task), into a device table that `cpq_diag_wg_trace` copies out.
tools/wg_trace.sh builds the library from the variant, runs tools/wg_trace_probe.py and puts the tracked source back."""
import sys

src = open(sys.argv[1], encoding="utf-8").read()


def put(anchor, text, after=True, count=1):
    global src
    assert src.count(anchor) == count, (anchor, src.count(anchor))
    src = src.replace(anchor, anchor + text if after else text + anchor)


put("template <int NT, bool CHAINED, bool PARTIAL, class SH>\n__device__ __forceinline__ int tpv_fast_span", """
struct WgTrace { unsigned long long t0, t1, c0, c1; unsigned hw, xcc, block, nSpans; unsigned long long spanT[96]; unsigned spanId[96]; unsigned long long spanC[96]; unsigned long long last; unsigned long long phase[8]; };
__device__ WgTrace g_wgTrace[4096];
__device__ __forceinline__ void wg_trace_begin(int tid)
{
    if (tid == 0 && blockIdx.x < 4096) {
        WgTrace& w = g_wgTrace[blockIdx.x];
        w.t0 = __builtin_amdgcn_s_memrealtime();
        w.c0 = __builtin_amdgcn_s_memtime();
        w.hw = __builtin_amdgcn_s_getreg(63492);        // HW_REG_HW_ID
        w.xcc = __builtin_amdgcn_s_getreg(63508);       // HW_REG_XCC_ID
        w.block = blockIdx.x;
        w.nSpans = 0;
        w.last = w.t0;
        for (int k = 0; k < 8; ++k) w.phase[k] = 0;
    }
}
__device__ __forceinline__ void wg_trace_span(int tid, int id)
{
    if (tid == 0 && blockIdx.x < 4096) {
        WgTrace& w = g_wgTrace[blockIdx.x];
        const unsigned k = w.nSpans;
        if (k < 96) { w.spanT[k] = __builtin_amdgcn_s_memrealtime(); w.spanId[k] = (unsigned)id; w.spanC[k] = __builtin_amdgcn_s_memtime(); }
        w.nSpans = k + 1;
    }
}
// time since the workgroup's last stamp, added to phase k (0: between tasks, 1: ticket, 2: tables and masks, 3: span load and
// range check, 4: band loop, 5: span store)
__device__ __forceinline__ void wg_trace_phase(int tid, int k)
{
    if (tid == 0 && blockIdx.x < 4096) {
        WgTrace& w = g_wgTrace[blockIdx.x];
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        w.phase[k] += now - w.last;
        w.last = now;
    }
}
__device__ __forceinline__ void wg_trace_end(int tid)
{
    if (tid == 0 && blockIdx.x < 4096) {
        WgTrace& w = g_wgTrace[blockIdx.x];
        w.t1 = __builtin_amdgcn_s_memrealtime();
        w.c1 = __builtin_amdgcn_s_memtime();
    }
}

""", after=False)
put("        const int c = (int)blockIdx.x;\n        const double* __restrict__ cf = coef + (int64_t)c * kBands * 6;\n        const TpBandTables* __restrict__ tb",
    "        if (WAVES == 8) wg_trace_begin(tid);\n", after=False)
put("            { double* t = sState; sState = sNext; sNext = t; }\n", "            if (WAVES == 8) wg_trace_span(tid, sp);\n")
put("state[(int64_t)c * kBands * 2 + tidE] = sState[tidE];\n", "        if (WAVES == 8) wg_trace_end(tid);\n")
put("        int cur = -1;                     // channel whose tables are in LDS\n", "        wg_trace_begin(tid);\n")
put("            if (sp + 1 == nSpans && tidE < kBands * 2 && ((bm.active >> (tidE >> 1)) & 1)) state[(int64_t)c * kBands * 2 + tidE] = sh.stateB[tidE];\n",
    "            wg_trace_span(tid, q);\n")
put("        // the last workgroup to get here resets the ticket and advances the generation for the next launch\n", "        wg_trace_end(tid);\n")
put("    else         tpv_span_load(srcW, buf, laneIo, x);\n", "    if (NT == 512) wg_trace_phase(tidS, 2);\n", after=False)
put("    bool bad = false;\n#pragma unroll\n    for (int j = 0; j < 16; ++j) bad |= !(fabs(x[j]) < kTpInputBound);\n", "    if (NT == 512) wg_trace_phase(tidS, 3);\n", after=False)
put("    laneIo = tpv_lane_id();\n    if (PARTIAL) tpv_span_store_partial(dstW, buf, laneIo, x, gain, nValidW);\n",
    "    if (NT == 512) wg_trace_phase((waveU << 6) + tpv_lane_id(), 4);\n", after=False)
put("    __syncthreads();                      // the last thread's end states are in sNext; every wave's stores are out\n",
    "    if (NT == 512) wg_trace_phase((waveU << 6) + tpv_lane_id(), 5);\n")
put("            if (q >= nTasks) break;\n", "            wg_trace_phase(tid, 1);\n")
put("            __syncthreads();              // the task before is done with sh (tables, states, task word)\n", "            wg_trace_phase(tid, 0);\n", after=False)
src += """
extern "C" int cpq_diag_wg_trace(void* dst, size_t bytes)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cpq::g_wgTrace), bytes) == hipSuccess ? 0 : -1;
}
"""
open(sys.argv[2], "w", encoding="utf-8").write(src)
