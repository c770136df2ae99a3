// How the 256 MB memory-side cache treats a resident read set next to a streaming write set
// (developer tool): per iteration read all of A (ra MiB) and write all of D (wd MiB).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ void k_rw(const v4* __restrict__ a, size_t na, v4* __restrict__ d, size_t nd, float* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    const v4 v = {1.f, 2.f, 3.f, 4.f};
    size_t m = na > nd ? na : nd;
    for (; i < m; i += s) {
        if (i < na) { v4 x = a[i]; acc += x.x + x.w; }
        if (i < nd) { if (NT) __builtin_nontemporal_store(v, &d[i]); else d[i] = v; }
    }
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float* o; hipMalloc(&o, 4);
    for (int uncached = 0; uncached < 2; ++uncached)
    for (size_t ra : {0, 64, 128})
    for (size_t wd : {64, 128, 192, 256, 384, 768}) {
        v4 *a = nullptr, *d = nullptr;
        if (ra) { hipMalloc(&a, ra << 20); hipMemset(a, 1, ra << 20); }
        if (uncached) { if (hipExtMallocWithFlags((void**)&d, wd << 20, hipDeviceMallocUncached) != hipSuccess) { printf("uncached alloc failed\n"); return 0; } }
        else hipMalloc(&d, wd << 20);
        hipMemset(d, 0, wd << 20);
        for (int nt = 0; nt < 2; ++nt) {
            auto launch = [&] { if (nt) k_rw<1><<<4096, 256>>>(a, (ra << 20) / 16, d, (wd << 20) / 16, o); else k_rw<0><<<4096, 256>>>(a, (ra << 20) / 16, d, (wd << 20) / 16, o); };
            const int reps = 20;
            for (int w = 0; w < 5; ++w) launch();
            hipEventRecord(e0); for (int r = 0; r < reps; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double us = ms * 1e3 / reps;
            printf("%s read %3zu MiB + write %3zu MiB %s: %6.1f us  %5.0f GB/s\n", uncached ? "UC" : "  ", ra, wd, nt ? "nt" : "  ", us, (ra + wd) * 1.048576 / us * 1e3);
        }
        if (a) hipFree(a); hipFree(d);
    }
    return 0;
}
