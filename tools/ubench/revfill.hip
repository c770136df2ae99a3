// Does the lane ORDER of a fully covered store matter?  The mirrored map rows are written with descending addresses over the
// lanes of a wave (lane l -> texel base - l).  Same 1 KiB per wave instruction either way; plain and nt flavours; buffer far
// beyond the Infinity Cache.  Also: ascending, but each lane 16 B at a 32-byte stride (half-covered lines) for scale.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int MODE, bool NT>
__global__ void k_fill(f4* __restrict__ d, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; const size_t s = (size_t)gridDim.x * blockDim.x;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (; i < n; i += s) {
        size_t j = i;
        if (MODE == 1) j = (i & ~(size_t)63) + (63 - (i & 63));                 // reversed within the wave
        if (MODE == 2) j = (i & ~(size_t)63) + (((i & 63) * 17) & 63);          // a permutation within the wave's 1 KiB
        if (MODE == 3) j = (i & ~(size_t)3) + (3 - (i & 3));                    // reversed within each 64-byte quad
        f4* p = d + j;
        if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    }
}
int main()
{
    const size_t bytes = (size_t)1024 << 20, n = bytes / 16;
    f4* d; if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    (void)hipMemset(d, 0, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto time = [&](auto kern, const char* name) {
        const int reps = 20;
        for (int w = 0; w < 3; ++w) kern<<<4096, 256>>>(d, n);
        (void)hipEventRecord(e0); for (int r = 0; r < reps; ++r) kern<<<4096, 256>>>(d, n); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %6.0f GB/s\n", name, bytes * 1e-9 * reps / (ms * 1e-3));
    };
    for (int rep = 0; rep < 2; ++rep) {
        time(k_fill<0, false>, "ascending plain"); time(k_fill<1, false>, "descending plain"); time(k_fill<2, false>, "permuted plain"); time(k_fill<3, false>, "quad-reversed plain");
        time(k_fill<0, true>, "ascending nt"); time(k_fill<1, true>, "descending nt"); time(k_fill<2, true>, "permuted nt"); time(k_fill<3, true>, "quad-reversed nt");
    }
    return 0;
}
