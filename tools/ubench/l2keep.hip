// Does data written by one kernel stay in the writing XCD's L2 for the next kernel?  Developer tool.
// Kernel W: workgroup g writes chunk (g) of a buffer.  Kernel R reads chunk (g + shift*?) ... with shift 0 the
// reader of a chunk runs on the XCD that wrote it (workgroups are dealt round-robin over the 8 XCDs), with
// shift 1 on the neighbouring XCD.  Total footprint is varied around the 8 x 4 MiB of L2.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_xcc(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | 20); }
__global__ void k_write(float4* buf, size_t per_wg) {
    float4* p = buf + (size_t)blockIdx.x * per_wg;
    for (size_t i = threadIdx.x; i < per_wg; i += blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void k_read(const float4* buf, size_t per_wg, int shift, int nwg, float* sink) {
    const int g = (blockIdx.x + shift) % nwg;
    const float4* p = buf + (size_t)g * per_wg;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < per_wg; i += blockDim.x) { float4 v = p[i]; acc += v.x + v.w; }
    if (acc == 123.f) sink[0] = acc;
}
int main() {
    const int nwg = 1024;
    int* dx; CK(hipMalloc(&dx, nwg * 4)); int hx[64];
    k_xcc<<<64, 64>>>(dx); CK(hipMemcpy(hx, dx, 64 * 4, hipMemcpyDeviceToHost));
    printf("xcc id of workgroups 0..15:"); for (int i = 0; i < 16; ++i) printf(" %d", hx[i]); printf("\n");
    float* sink; CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : {8, 16, 24, 32, 64, 128}) {
        const size_t bytes = mb << 20, per_wg = bytes / 16 / nwg;
        float4* buf; CK(hipMalloc(&buf, bytes));
        for (int shift : {0, 1, 8}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                k_write<<<nwg, 256>>>(buf, per_wg);
                hipEventRecord(e0);
                k_read<<<nwg, 256>>>(buf, per_wg, shift, nwg, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("%4zu MiB  reader shift %d (%s XCD): read %.1f us  %.0f GB/s\n", mb, shift, shift % 8 ? "other" : "same", best * 1e3, bytes / (best * 1e-3) * 1e-9);
        }
        hipFree(buf);
    }
    return 0;
}
