// Throughput of the block-level FFT engine alone (fft_engine.h), in the geometries the z pass could use at N = 2048 (developer tool):
// inputs come from registers, outputs are folded into one dummy store, so what is timed is butterflies + LDS exchanges + barriers --
// the "skeleton" that the ablation of round 2 found to be 60 % of the z pass.  One workgroup = the four transforms of a column.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I watersurfacerendering_amd/csrc tools/ubench/fftskel.hip -o tools/ubench/fftskel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "fft_engine.h"
using namespace ocean;

template <int N, int C, int T, class P, int BATCHES, int MINW>
__global__ void __launch_bounds__(T, MINW) k_skel(const c32* __restrict__ tw, float* __restrict__ sink, float seed)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* fbuf = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    TwiddleRegs<N, C, T, P> twr;
    twr.load(tw, tid);
    float acc = 0.0f;
    const float s0 = seed + 1e-3f * (float)blockIdx.x;
#pragma unroll 1
    for (int b = 0; b < BATCHES; ++b) {
        const float sb = s0 + (float)b;
        auto in = [&](int e, int c, int, int) -> c32 { return make_float2(sb + 1e-4f * (float)e, (float)c - 1e-4f * (float)e); };
        auto out = [&](int, int, c32 v, int, int) { acc += v.x - v.y; };
        batch_fft<N, C, T, P>(fbuf, twr, tid, in, out);
    }
    if (acc == 1234.5678f) sink[blockIdx.x] = acc;
}

template <int N, int C, int T, class P, int BATCHES, int MINW>
static void run(const char* name, const c32* tw, float* sink, size_t extra_lds, int grid)
{
    const size_t lds = sizeof(c32) * fft_lds_elems<N, C>() + extra_lds;
    auto kern = k_skel<N, C, T, P, BATCHES, MINW>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, 0, tw, sink, 0.5f);
    const int reps = 50;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(T), lds, 0, tw, sink, 0.5f + r);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, T, lds);
    printf("%-58s grid %5d  LDS %6zu B  %d WG/CU  %7.2f us per launch  (%d transforms)\n", name, grid, lds, occ, ms / reps * 1e3, grid * C * BATCHES);
}

int main()
{
    constexpr int N = 2048;
    std::vector<c32> tw(N);
    for (int k = 0; k < N; ++k) { const double a = 2.0 * M_PI * k / N; tw[k] = make_float2((float)cos(a), (float)sin(a)); }
    c32* dtw; float* sink;
    hipMalloc(&dtw, N * sizeof(c32)); hipMalloc(&sink, 1 << 20);
    hipMemcpy(dtw, tw.data(), N * sizeof(c32), hipMemcpyHostToDevice);
    for (int grid : {768, 1025}) {
        run<N, 2, 512, Radices<8, 8, 8, 4>, 2, 6>("A  512 thr, 2 x (2 transforms), radix 8.8.8.4 [z pass]", dtw, sink, 16384, grid);
        run<N, 2, 512, Radices<8, 8, 8, 4>, 2, 6>("A' same, no table space (4 WG/CU by LDS)", dtw, sink, 0, grid);
        run<N, 2, 256, Plan<2048>, 2, 3>("D  256 thr, 2 x (2 transforms), radix 16.16.8", dtw, sink, 16384, grid);
        run<N, 2, 512, Plan<2048>, 2, 4>("H  512 thr (half idle in the radix-16 stages), 16.16.8, 128 VGPRs", dtw, sink, 16384, grid);
        run<N, 1, 256, Radices<8, 8, 8, 4>, 4, 4>("E  256 thr, 4 x (1 transform), radix 8.8.8.4", dtw, sink, 16384, grid);
        run<N, 1, 128, Plan<2048>, 4, 3>("F  128 thr, 4 x (1 transform), radix 16.16.8", dtw, sink, 16384, grid);
        run<N, 1, 64, Plan<2048>, 4, 2>("B  64 thr (one wave), 4 x (1 transform), radix 16.16.8", dtw, sink, 16384, grid);
        run<N, 1, 64, Plan<2048>, 4, 2>("B' same, no table space", dtw, sink, 0, grid);
        run<N, 4, 1024, Radices<8, 8, 8, 4>, 1, 2>("G  1024 thr, 1 x (4 transforms), radix 8.8.8.4", dtw, sink, 16384, grid);
    }
    return 0;
}
