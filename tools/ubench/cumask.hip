// Round 6 (late): how do the bits of a stream's compute-unit mask (hipExtStreamCreateWithCUMask) map onto the part's 8 XCDs x 32 CUs?
// For a few mask patterns: a grid of 4096 small workgroups that each stay ~20 us, recording HW_REG_XCC_ID and the CU / SH / SE fields of
// HW_REG_HW_ID; prints how many distinct CUs were used and how they spread over the XCDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
__global__ void k_where(unsigned* out)
{
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid;
    }
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(8);      // 20 us of 100 MHz ticks
}
static int b_all(int) { return 1; }
static int b_lo128(int i) { return i < 128; }
static int b_lo32(int i) { return i < 32; }
static int b_even(int i) { return (i & 1) == 0; }
static int b_mod8(int i) { return i % 8 < 3; }
static int b_div8(int i) { return (i / 8) % 32 < 12; }
static int b_mod32(int i) { return i % 32 < 12; }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    const int G = 4096;
    unsigned* d; CK(hipMalloc(&d, 2 * G * sizeof(unsigned)));
    std::vector<unsigned> h(2 * G);
    struct Pat { const char* name; int (*bit)(int); };
    const Pat pats[] = {{"all 256", b_all}, {"bits 0..127", b_lo128}, {"bits 0..31", b_lo32}, {"even bits", b_even}, {"i % 8 < 3", b_mod8},
                        {"(i / 8) % 32 < 12", b_div8}, {"i % 32 < 12", b_mod32}};
    for (const Pat& p : pats) {
        uint32_t mask[8] = {};
        int set = 0;
        for (int i = 0; i < 256; ++i) if (p.bit(i)) { mask[i / 32] |= 1u << (i % 32); ++set; }
        hipStream_t st; CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
        hipLaunchKernelGGL(k_where, dim3(G), dim3(64), 0, st, d);
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h.data(), d, 2 * G * sizeof(unsigned), hipMemcpyDeviceToHost));
        std::set<unsigned> cus; int per[8] = {}; std::set<unsigned> perx[8];
        for (int i = 0; i < G; ++i) { const unsigned x = h[2 * i] & 7u, key = (h[2 * i + 1] >> 8) & 0xFFu; cus.insert(x << 8 | key); perx[x].insert(key); per[x]++; }
        printf("%-20s bits set %3d -> distinct CUs used %3zu; CUs per XCC:", p.name, set, cus.size());
        for (int x = 0; x < 8; ++x) printf(" %zu", perx[x].size());
        printf("   workgroups per XCC:");
        for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
        printf("\n");
        CK(hipStreamDestroy(st));
    }
    return 0;
}
