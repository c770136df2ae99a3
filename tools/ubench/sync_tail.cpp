// Developer probe: the tail of the synchronous ocean_compute_waves -- the reference's call shape, one blocking
// ComputeWaves per frame (WaterSurfaceMesh.cpp:145-154) -- from a plain C++ host, no Python in the way.  Per call: host time inside the
// enqueue (three launches) and inside the wait (poll of the completion records), 10 000 calls; percentiles of both and of their sum, the
// spacing of the slow calls, and the same with the calling thread pinned to one CPU.
//   g++ -O2 -std=c++17 tools/ubench/sync_tail.cpp -Iinclude -Lwatersurfacerendering_amd -locean_hip -Wl,-rpath,$PWD/watersurfacerendering_amd -o /tmp/sync_tail
//   /tmp/sync_tail N [calls] [pin_cpu or -1]
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ocean.h"

static double pct(std::vector<double> v, double p)
{
    std::sort(v.begin(), v.end());
    return v[(size_t)(p * (v.size() - 1))];
}

int main(int argc, char** argv)
{
    const unsigned n = argc > 1 ? (unsigned)std::atoi(argv[1]) : 2048;
    const int calls = argc > 2 ? std::atoi(argv[2]) : 10000;
    const int pin = argc > 3 ? std::atoi(argv[3]) : -1;
    if (pin >= 0) {
        cpu_set_t set;
        CPU_ZERO(&set);
        CPU_SET(pin, &set);
        if (sched_setaffinity(0, sizeof set, &set) != 0) std::perror("sched_setaffinity");
    }
    ocean_t* c = nullptr;
    if (ocean_create(&c, n, 1, 0) != OCEAN_OK || ocean_prepare(c, 1, nullptr) != OCEAN_OK) { std::fprintf(stderr, "create / prepare failed\n"); return 1; }
    ocean_set_frame_tracking(c, 1);
    float amp = 0.f;
    for (int j = 0; j < 200; ++j) ocean_compute_waves(c, 0.05f * j, &amp);
    using clk = std::chrono::steady_clock;
    std::vector<double> enq(calls), wait(calls), tot(calls);
    for (int j = 0; j < calls; ++j) {
        const auto t0 = clk::now();
        if (ocean_compute_waves_async(c, 0.05f * j) != OCEAN_OK) return 2;
        const auto t1 = clk::now();
        if (ocean_wait_frame(c, &amp) != OCEAN_OK) return 3;
        const auto t2 = clk::now();
        enq[j] = std::chrono::duration<double, std::micro>(t1 - t0).count();
        wait[j] = std::chrono::duration<double, std::micro>(t2 - t1).count();
        tot[j] = enq[j] + wait[j];
    }
    const double med = pct(tot, 0.5);
    std::printf("N=%u calls=%d pin=%d   total  p50 %.1f  p90 %.1f  p95 %.1f  p99 %.1f  max %.1f us   (p95 / p50 = %.2f)\n", n, calls, pin, med, pct(tot, 0.9),
                pct(tot, 0.95), pct(tot, 0.99), pct(tot, 1.0), pct(tot, 0.95) / med);
    std::printf("                        enqueue p50 %.1f  p90 %.1f  p95 %.1f  p99 %.1f  max %.1f us\n", pct(enq, 0.5), pct(enq, 0.9), pct(enq, 0.95), pct(enq, 0.99), pct(enq, 1.0));
    std::printf("                        wait    p50 %.1f  p90 %.1f  p95 %.1f  p99 %.1f  max %.1f us\n", pct(wait, 0.5), pct(wait, 0.9), pct(wait, 0.95), pct(wait, 0.99), pct(wait, 1.0));
    // which calls are slow, and in which half of the call
    int slow = 0, slow_enq = 0, last = -1;
    std::vector<int> gaps;
    for (int j = 0; j < calls; ++j)
        if (tot[j] > 1.15 * med) {
            ++slow;
            if (enq[j] > pct(enq, 0.5) + 0.1 * med) ++slow_enq;
            if (last >= 0) gaps.push_back(j - last);
            last = j;
        }
    std::sort(gaps.begin(), gaps.end());
    std::printf("                        calls above 1.15 x median: %d (%.1f %%), of which slow in the enqueue: %d; median spacing %d calls\n", slow, 100.0 * slow / calls,
                slow_enq, gaps.empty() ? 0 : gaps[gaps.size() / 2]);
    ocean_destroy(c);
    return 0;
}
