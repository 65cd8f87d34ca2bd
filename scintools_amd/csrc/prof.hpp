// prof.hpp -- optional hipEvent bracketing of the two hot kernels (gather, mat-vec) so that
// bench.py can report per-kernel average launch time from the timed region itself.
#pragma once
#include <vector>

#include "common.hpp"

namespace scint {

enum ProfKernel { kProfGather = 0, kProfMatvec = 1, kProfCount = 2 };

struct Profiler {
    bool enabled = false;
    std::vector<hipEvent_t> pool;          // recycled events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> open[kProfCount];
    double ms[kProfCount] = {0, 0};
    int64_t launches[kProfCount] = {0, 0};

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
    // call after the stream has been synchronised
    void collect() {
        for (int k = 0; k < kProfCount; ++k) {
            for (auto& pr : open[k]) {
                float t = 0.f;
                if (hipEventElapsedTime(&t, pr.first, pr.second) == hipSuccess) {
                    ms[k] += t;
                    launches[k] += 1;
                }
                pool.push_back(pr.first);
                pool.push_back(pr.second);
            }
            open[k].clear();
        }
    }
    void reset() {
        collect();
        for (int k = 0; k < kProfCount; ++k) { ms[k] = 0; launches[k] = 0; }
    }
};

Profiler& profiler();

}  // namespace scint
