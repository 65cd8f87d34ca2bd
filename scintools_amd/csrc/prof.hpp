// prof.hpp -- optional hipEvent bracketing of the two hot kernels (gather, mat-vec) so that
// bench.py can report per-kernel average launch time from the timed region itself.
#pragma once
#include <algorithm>
#include <vector>

#include "common.hpp"

namespace scint {

// (2: the complex64 mat-vec of the mixed sweep; 3, 4: the model step of the chi^2 sweep -- rank-1 back-map with its bound kernel,
//  complex-to-real model transform with the chi^2 sink and the final sum)
//  5, 6, 7: the three kernels of the two-trip calc_sspec -- input copy + sums, strided axis, row transforms + |.|^2 / dB)
enum ProfKernel { kProfGather = 0, kProfMatvec = 1, kProfMatvec32 = 2, kProfRevmap = 3, kProfModel = 4,
                  kProfSspecPrep = 5, kProfSspecCols = 6, kProfSspecRows = 7, kProfCount = 8 };

struct Profiler {
    bool enabled = false;
    std::vector<hipEvent_t> pool;          // recycled events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> open[kProfCount];
    // ms[k] is the length of the UNION of kernel k's launch intervals: the sweep drives two
    // streams, so launches of one kernel may overlap in time and must not be counted twice.
    double ms[kProfCount] = {};
    double ms_sum[kProfCount] = {};    // plain sum of the individual launch spans (what rocprofv3 averages)
    int64_t launches[kProfCount] = {};
    hipEvent_t base = nullptr;             // time origin of the current begin/end window
    std::vector<std::pair<float, float>> spans[kProfCount];

    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
    // record the start event; returns a slot index or -1
    int begin(ProfKernel k, hipStream_t s) {
        if (!enabled) return -1;
        hipEvent_t a = get(), b = get();
        if (!a || !b) return -1;
        (void)hipEventRecord(a, s);
        open[k].push_back({a, b});
        return (int)open[k].size() - 1;
    }
    void end(ProfKernel k, int slot, hipStream_t s) {
        if (slot >= 0) (void)hipEventRecord(open[k][(size_t)slot].second, s);
    }
    // harvest every bracket whose stop event has completed; the rest stay open
    void collect() {
        if (!base) return;
        for (int k = 0; k < kProfCount; ++k) {
            std::vector<std::pair<hipEvent_t, hipEvent_t>> keep;
            for (auto& pr : open[k]) {
                if (hipEventQuery(pr.second) != hipSuccess) { keep.push_back(pr); continue; }
                float t0 = 0.f, t1 = 0.f;
                if (hipEventElapsedTime(&t0, base, pr.first) == hipSuccess &&
                    hipEventElapsedTime(&t1, base, pr.second) == hipSuccess) {
                    spans[k].push_back({t0, t1});
                    ms_sum[k] += (double)(t1 - t0);
                    launches[k] += 1;
                }
                pool.push_back(pr.first);
                pool.push_back(pr.second);
            }
            open[k].swap(keep);
        }
    }
    // union length of the harvested intervals (call after a device synchronise + collect)
    void finish() {
        for (int k = 0; k < kProfCount; ++k) {
            std::sort(spans[k].begin(), spans[k].end());
            double total = 0.0;
            float lo = 0.f, hi = -1.f;
            for (auto& sp : spans[k]) {
                if (hi < lo || sp.first > hi) {
                    if (hi >= lo) total += (double)(hi - lo);
                    lo = sp.first; hi = sp.second;
                } else if (sp.second > hi) {
                    hi = sp.second;
                }
            }
            if (hi >= lo) total += (double)(hi - lo);
            ms[k] = total;
            spans[k].clear();
        }
    }
    void reset(hipStream_t s) {
        (void)hipDeviceSynchronize();
        collect();
        for (int k = 0; k < kProfCount; ++k) { ms[k] = 0; ms_sum[k] = 0; launches[k] = 0; spans[k].clear(); }
        if (!base) (void)hipEventCreate(&base);
        if (base) (void)hipEventRecord(base, s);
    }
};

Profiler& profiler();

}  // namespace scint
