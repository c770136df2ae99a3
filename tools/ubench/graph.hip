// Host cost of three dependent small kernels per "frame": stream launches vs one hipGraphLaunch
// (with and without updating one node's parameters per frame).  Developer tool.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Args { float* p; float t; int pad[24]; };
__global__ void k_a(Args a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[0] = a.t; }
__global__ void k_b(Args a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[1] = a.p[0] + 1.f; }
__global__ void k_c(Args a) { if (threadIdx.x == 0 && blockIdx.x == 0) a.p[2] = a.p[1] + 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    float* d; CK(hipMalloc(&d, 64));
    const int D = 4, frames = 20000;
    hipStream_t st[D]; for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Args a{d, 0.f, {}};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto t0, auto t1) { return std::chrono::duration<double, std::micro>(t1 - t0).count(); };
    for (int depth : {1, 4}) {
        // (a) plain launches
        for (int w = 0; w < 2; ++w) {
            auto t0 = now();
            for (int f = 0; f < frames; ++f) {
                a.t = (float)f; hipStream_t s = st[f % depth];
                hipLaunchKernelGGL(k_a, dim3(256), dim3(256), 0, s, a);
                hipLaunchKernelGGL(k_b, dim3(256), dim3(256), 0, s, a);
                hipLaunchKernelGGL(k_c, dim3(256), dim3(256), 0, s, a);
            }
            auto t1 = now();
            for (int i = 0; i < depth; ++i) CK(hipStreamSynchronize(st[i]));
            auto t2 = now();
            if (w) printf("depth %d  launches: enqueue %.2f us/frame, total %.2f us/frame\n", depth, us(t0, t1) / frames, us(t0, t2) / frames);
        }
        // (b) graphs, one per chain, node 0 updated per frame
        hipGraph_t g[D]; hipGraphExec_t ge[D]; hipGraphNode_t n0[D];
        for (int i = 0; i < depth; ++i) {
            CK(hipGraphCreate(&g[i], 0));
            void* params[] = {&a};
            hipKernelNodeParams kp{}; kp.blockDim = dim3(256); kp.gridDim = dim3(256); kp.kernelParams = params; kp.sharedMemBytes = 0;
            hipGraphNode_t na, nb, nc;
            kp.func = (void*)k_a; CK(hipGraphAddKernelNode(&na, g[i], nullptr, 0, &kp));
            kp.func = (void*)k_b; CK(hipGraphAddKernelNode(&nb, g[i], &na, 1, &kp));
            kp.func = (void*)k_c; CK(hipGraphAddKernelNode(&nc, g[i], &nb, 1, &kp));
            n0[i] = na;
            CK(hipGraphInstantiate(&ge[i], g[i], nullptr, nullptr, 0));
        }
        for (int upd = 0; upd < 2; ++upd)
            for (int w = 0; w < 2; ++w) {
                auto t0 = now();
                for (int f = 0; f < frames; ++f) {
                    const int c = f % depth;
                    if (upd) {
                        a.t = (float)f; void* params[] = {&a};
                        hipKernelNodeParams kp{}; kp.func = (void*)k_a; kp.blockDim = dim3(256); kp.gridDim = dim3(256); kp.kernelParams = params;
                        hipGraphExecKernelNodeSetParams(ge[c], n0[c], &kp);
                    }
                    hipGraphLaunch(ge[c], st[c]);
                }
                auto t1 = now();
                for (int i = 0; i < depth; ++i) CK(hipStreamSynchronize(st[i]));
                auto t2 = now();
                if (w) printf("depth %d  graph%s: enqueue %.2f us/frame, total %.2f us/frame\n", depth, upd ? "+update" : "       ", us(t0, t1) / frames, us(t0, t2) / frames);
            }
    }
    return 0;
}
