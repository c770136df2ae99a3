// Skeleton of VERDICT r05 #6's concurrent split of a serial 512^2 frame (developer tool): kernels that just take the time the real ones take
// (z pass 6.4 us, HEIGHT-only ~3.0, NORMAL-only ~4.8, k_xpass_b 5.2, displacement pass 4.5: serial kernel times of round 6), launched
//   serial      z -> xb -> xd on one stream (today's frame)
//   side        main: z -> HEIGHT -> xd;  side stream: wait(event behind z) -> NORMAL;  host waits for both streams' last events
//   halves      main: z{height, pair 0} (4.4) -> HEIGHT -> xd;  side: z{pair 1, pair 2} (4.4) -> NORMAL  (no cross-stream wait at all)
// host wall time per frame over 2000 frames, event polls (no blocking synchronisation).  What the launch machinery allows, whatever the kernels do.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_busy(unsigned ticks10ns)      // 64 workgroups of 64 threads spin for the given time (100 MHz counter)
{
    const unsigned long long t0 = wall_clock64();
    while ((unsigned)(wall_clock64() - t0) < ticks10ns) __builtin_amdgcn_s_sleep(2);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static void poll(hipEvent_t e) { while (hipEventQuery(e) == hipErrorNotReady) __builtin_ia32_pause(); }
int main()
{
    hipStream_t m, s; CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t ez, em, es; CK(hipEventCreateWithFlags(&ez, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&em, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&es, hipEventDisableTiming));
    auto K = [&](hipStream_t st, double us) { k_busy<<<64, 64, 0, st>>>((unsigned)(us * 100.0)); };
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            const int frames = 2000;
            double total = 0;
            for (int f = -100; f < frames; ++f) {
                const auto t0 = std::chrono::steady_clock::now();
                if (mode == 0) { K(m, 6.4); K(m, 5.2); K(m, 4.5); CK(hipEventRecord(em, m)); poll(em); }
                else if (mode == 1) {
                    K(m, 6.4); CK(hipEventRecord(ez, m)); K(m, 3.0); K(m, 4.5); CK(hipEventRecord(em, m));
                    CK(hipStreamWaitEvent(s, ez, 0)); K(s, 4.8); CK(hipEventRecord(es, s));
                    poll(em); poll(es);
                } else {
                    K(m, 4.4); K(m, 3.0); K(m, 4.5); CK(hipEventRecord(em, m));
                    K(s, 4.4); K(s, 4.8); CK(hipEventRecord(es, s));
                    poll(em); poll(es);
                }
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                if (f >= 0) total += us;
            }
            printf("%-7s  %6.2f us per frame (sum of its kernels on the critical path: %.1f)\n", mode == 0 ? "serial" : mode == 1 ? "side" : "halves",
                   total / frames, mode == 0 ? 16.1 : mode == 1 ? 13.9 : 11.9);
        }
    }
    return 0;
}
