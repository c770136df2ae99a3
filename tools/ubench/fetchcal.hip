// FETCH_SIZE calibration (developer tool; profiles/README.md): kernels that read a KNOWN number of bytes, each byte
// exactly once, from a buffer far larger than the 256 MiB Infinity Cache, in the access shapes of the ocean kernels.
// Run under `rocprofv3 --pmc FETCH_SIZE` (and `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum`): the ratio
// known bytes / (FETCH_SIZE KB * 1024) per kernel is the correction factor for that access width.
// MI355X_MICROARCH.md (HBM section) gives 2.0 for 16 B/lane coalesced reads and calls other widths uncalibrated.
//   k_read16        16 B per lane, fully coalesced (the z pass's h0 loads)
//   k_read8         8 B per lane, fully coalesced
//   k_read4         4 B per lane, fully coalesced (raw-height rows)
//   k_read2         2 B per lane pairs: 4-byte loads of two 16-bit values (the z pass's 16-bit dispersion)
//   k_gather32x8    8 B per lane, 4 consecutive lanes = one 32-byte piece, pieces `stride` bytes apart
//                   (the x pass reading 4 neighbouring columns of the half-size intermediates)
//   k_gather32x16   16 B per lane, 2 consecutive lanes = one 32-byte piece (the height role's float4 loads)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <class T> __global__ void k_read(const T* __restrict__ a, float* __restrict__ out, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t s = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += s) { const T v = a[i]; acc += reinterpret_cast<const float*>(&v)[0]; }
    if (acc == 123.456f) out[0] = acc;
}
// pieces of 32 bytes: piece p of "row" r sits at byte r * stride + p * 32; a wave reads 16 (8 B lanes) or 32 (16 B
// lanes) pieces of DIFFERENT rows per instruction; every piece of the buffer is read exactly once overall
template <class T> __global__ void k_gather32(const char* __restrict__ a, float* __restrict__ out, size_t rows, size_t stride)
{
    constexpr int LP = 32 / sizeof(T);                     // lanes per piece
    const size_t pieces_per_row = stride / 32;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    const size_t lane_in_piece = tid % LP, slot = tid / LP, nslots = nthreads / LP;
    float acc = 0.f;
    // slot walks over (piece column, row) with the ROW fastest: neighbouring slots hit different rows
    for (size_t w = slot; w < rows * pieces_per_row; w += nslots) {
        const size_t r = w % rows, p = w / rows;
        const T v = *reinterpret_cast<const T*>(a + r * stride + p * 32 + lane_in_piece * sizeof(T));
        acc += reinterpret_cast<const float*>(&v)[0];
    }
    if (acc == 123.456f) out[0] = acc;
}
int main()
{
    const size_t bytes = (size_t)1 << 30;                  // 1 GiB, 4x the Infinity Cache
    char* a; float* o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&o, 4) != hipSuccess) return 1;
    hipMemset(a, 1, bytes);
    hipDeviceSynchronize();
    const int blocks = 4096;
    for (int rep = 0; rep < 3; ++rep) {
        k_read<float4><<<blocks, 256>>>(reinterpret_cast<const float4*>(a), o, bytes / 16);
        k_read<float2><<<blocks, 256>>>(reinterpret_cast<const float2*>(a), o, bytes / 8);
        k_read<float><<<blocks, 256>>>(reinterpret_cast<const float*>(a), o, bytes / 4);
        k_read<unsigned><<<blocks, 256>>>(reinterpret_cast<const unsigned*>(a), o, bytes / 4);
        const size_t stride = 8256;                        // = 2 * NUP * 8 bytes at N = 2048: one spectrum row of a pair, both sides
        k_gather32<float2><<<blocks, 256>>>(a, o, bytes / stride, stride);
        k_gather32<float4><<<blocks, 256>>>(a, o, bytes / stride, stride);
        hipDeviceSynchronize();
    }
    printf("known bytes per kernel launch: %zu (gathers: %zu)\n", bytes, (bytes / 8256) * 8256);
    return 0;
}
