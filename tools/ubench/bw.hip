// Streaming bandwidth ceilings of the device this runs on (developer tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_fill(float4* __restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (; i < n; i += s) d[i] = v;
}
__global__ void k_read(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += s) { float4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) d[i] = a[i];
}
// 1 read : 1.5 write mix, like one ocean frame (133 MB in, 201 MB out)
__global__ void k_mix(const float4* __restrict__ a, float4* __restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, s = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += s) { float4 v = a[2 * i]; float4 w = a[2 * i + 1]; d[3 * i] = v; d[3 * i + 1] = w; d[3 * i + 2] = make_float4(v.x + w.x, v.y, w.z, 1.f); }
}
int main() {
  for (size_t mb : {16, 32, 64, 128, 256, 1024}) {
    const size_t bytes = mb << 20; const size_t n = bytes / 16;
    printf("== buffer %zu MiB (dst 1.5x)\n", mb);
    float4 *a, *d; float* o;
    hipMalloc(&a, bytes); hipMalloc(&d, bytes * 3 / 2); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(d, 0, bytes * 3 / 2);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](auto launch, double gb, const char* name) {
        const int reps = 20;
        for (int w = 0; w < 3; ++w) launch();
        hipEventRecord(e0); for (int r = 0; r < reps; ++r) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-6s %.0f GB/s\n", name, gb * reps / (ms * 1e-3));
    };
    for (int blocks : {4096}) {
        time([&] { k_fill<<<blocks, 256>>>(d, n); }, bytes * 1e-9, "fill");
        time([&] { k_read<<<blocks, 256>>>(a, o, n); }, bytes * 1e-9, "read");
        time([&] { k_copy<<<blocks, 256>>>(a, d, n); }, 2 * bytes * 1e-9, "copy");
        time([&] { k_mix<<<blocks, 256>>>(a, d, n / 2); }, 2.5 * bytes * 1e-9, "mix");
    }
    hipFree(a); hipFree(d); hipFree(o);
  }
    return 0;
}
