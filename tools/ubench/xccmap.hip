// Which XCD does workgroup i of a launch run on, and does it depend on the stream (hardware queue) the launch goes through?
// Eight streams created one after the other; on each, a grid of G workgroups records HW_REG_XCC_ID and the CU / SE fields of HW_REG_HW_ID.
// build: hipcc -O2 --offload-arch=gfx950 -o xccmap xccmap.hip ; usage: xccmap [G] [threads] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_where(unsigned* out)
{
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hwid;
        smem[0] = (unsigned char)xcc;
    }
}
int main(int argc, char** argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 64, T = argc > 2 ? atoi(argv[2]) : 512, lds = argc > 3 ? atoi(argv[3]) : 4096;
    unsigned* d; hipMalloc(&d, 2 * G * sizeof(unsigned));
    hipFuncSetAttribute((const void*)k_where, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<unsigned> h(2 * G);
    for (int s = 0; s < 8; ++s) {
        hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_where, dim3(G), dim3(T), lds, st, d);
            hipStreamSynchronize(st);
            hipMemcpy(h.data(), d, 2 * G * sizeof(unsigned), hipMemcpyDeviceToHost);
            printf("stream %d rep %d  xcc of workgroups 0..%d:", s, rep, G < 32 ? G - 1 : 31);
            for (int i = 0; i < G && i < 32; ++i) printf(" %u", h[2 * i] & 15u);
            int per[16] = {0};
            for (int i = 0; i < G; ++i) per[h[2 * i] & 15u]++;
            printf("   per xcc:");
            for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
            int rr = 0;
            for (int i = 0; i < G; ++i) rr += ((h[2 * i] & 15u) == ((h[0] & 15u) + i) % 8);
            printf("   round-robin from xcc %u: %d of %d\n", h[0] & 15u, rr, G);
        }
    }
    return 0;
}
