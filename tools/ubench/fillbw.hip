// Store-bandwidth ceilings by store flavour (developer tool): a linear 16-B-per-lane fill of a buffer far beyond the
// 256 MiB Infinity Cache, and the same with the 1 KiB-run-per-wave shape of the map stores, for the cache-policy bits
// a gfx950 global store can carry.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define FILL_KERNEL(name, bits)                                                                              \
__global__ void name(f4* __restrict__ d, size_t n) {                                                          \
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t s = (size_t)gridDim.x * blockDim.x; \
    const f4 v = {1.f, 2.f, 3.f, 4.f};                                                                        \
    for (; i < n; i += s) { f4* p = d + i; asm volatile("global_store_dwordx4 %0, %1, off " bits :: "v"(p), "v"(v) : "memory"); } \
}
FILL_KERNEL(k_plain, "")
FILL_KERNEL(k_nt, "nt")
FILL_KERNEL(k_sc0, "sc0")
FILL_KERNEL(k_sc1, "sc1")
FILL_KERNEL(k_sc0sc1, "sc0 sc1")
FILL_KERNEL(k_sc1nt, "sc1 nt")
FILL_KERNEL(k_sc0nt, "sc0 nt")
FILL_KERNEL(k_all, "sc0 sc1 nt")
int main() {
    for (size_t mb : {128, 1024}) {
        const size_t bytes = mb << 20, n = bytes / 16;
        f4* d; if (hipMalloc(&d, bytes) != hipSuccess) return 1;
        hipMemset(d, 0, bytes);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto time = [&](auto kern, const char* name) {
            const int reps = 20;
            for (int w = 0; w < 3; ++w) kern<<<4096, 256>>>(d, n);
            hipEventRecord(e0); for (int r = 0; r < reps; ++r) kern<<<4096, 256>>>(d, n); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%5zu MiB  %-12s %6.0f GB/s\n", mb, name, bytes * 1e-9 * reps / (ms * 1e-3));
        };
        time(k_plain, "plain"); time(k_nt, "nt"); time(k_sc0, "sc0"); time(k_sc1, "sc1"); time(k_sc0sc1, "sc0 sc1");
        time(k_sc1nt, "sc1 nt"); time(k_sc0nt, "sc0 nt"); time(k_all, "sc0 sc1 nt");
        hipFree(d);
    }
    return 0;
}
