// Device-to-host paths for the maps of one frame (developer tool, round 6: what should ocean_compute_waves_read use?).
// For 4 / 16 / 64 MiB per map, two maps: wall time from "both maps final in HBM" to "both maps in host memory", three ways --
//   sdma1   two hipMemcpyAsync on ONE stream (the runtime's DMA engines), event poll
//   sdma2   the same on TWO streams
//   kern    one copy kernel per map (grid-stride, 16 B per lane) writing the page-locked destination through its device address, on two streams
//   kern_nt the same with non-temporal stores
// against pinned (hipHostMalloc) and registered (malloc + hipHostRegister) destinations.  Prints us and GB/s.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f4 __attribute__((ext_vector_type(4)));
template <bool NT> __global__ void k_copy(const f4* __restrict__ s, f4* __restrict__ d, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { const f4 v = s[i]; if (NT) __builtin_nontemporal_store(v, d + i); else d[i] = v; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static void poll(hipEvent_t e) { while (hipEventQuery(e) == hipErrorNotReady) __builtin_ia32_pause(); }
int main()
{
    hipStream_t s[2]; hipEvent_t ev[2];
    for (int i = 0; i < 2; ++i) { CK(hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); }
    for (size_t mib : {4, 16, 64}) {
        const size_t bytes = mib << 20, n = bytes / 16;
        f4* d[2]; for (auto& p : d) { CK(hipMalloc(&p, bytes)); CK(hipMemset(p, 1, bytes)); }
        for (int reg = 0; reg < 2; ++reg) {
            void* h[2]; void* hd[2];
            for (int i = 0; i < 2; ++i) {
                if (reg) { h[i] = aligned_alloc(4096, bytes); memset(h[i], 0, bytes); CK(hipHostRegister(h[i], bytes, hipHostRegisterDefault)); }
                else CK(hipHostMalloc(&h[i], bytes, hipHostMallocDefault));
                CK(hipHostGetDevicePointer(&hd[i], h[i], 0));
            }
            auto run = [&](int mode) -> double {
                double best = 1e30, sum = 0; const int reps = 30;
                for (int r = 0; r < reps + 3; ++r) {
                    CK(hipDeviceSynchronize());
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int i = 0; i < 2; ++i) {
                        hipStream_t st = s[mode == 0 ? 0 : i];
                        if (mode <= 1) CK(hipMemcpyAsync(h[i], d[i], bytes, hipMemcpyDeviceToHost, st));
                        else if (mode == 2) k_copy<false><<<1024, 256, 0, st>>>(d[i], (f4*)hd[i], n);
                        else k_copy<true><<<1024, 256, 0, st>>>(d[i], (f4*)hd[i], n);
                        CK(hipEventRecord(ev[i], st));
                    }
                    poll(ev[0]); poll(ev[1]);
                    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                    if (r >= 3) { sum += us; if (us < best) best = us; }
                }
                printf("  %2zu MiB x 2  %-10s %-8s  mean %8.1f us  best %8.1f us  %5.1f GB/s\n", mib, reg ? "registered" : "pinned",
                       mode == 0 ? "sdma1" : mode == 1 ? "sdma2" : mode == 2 ? "kern" : "kern_nt", sum / reps, best, 2.0 * bytes / (sum / reps) * 1e-3);
                return 0;
            };
            for (int mode = 0; mode < 4; ++mode) run(mode);
            // the copies landed?
            if (((const unsigned char*)h[1])[bytes - 1] != 1) printf("  MISMATCH\n");
            for (int i = 0; i < 2; ++i) { if (reg) { CK(hipHostUnregister(h[i])); free(h[i]); } else CK(hipHostFree(h[i])); }
        }
        for (auto p : d) CK(hipFree(p));
    }
    return 0;
}
