// ocean_kernels.h -- device kernels of the MI355X ocean synthesiser (gfx950).
//
// Frame pipeline (one ComputeWaves(t), reference WSTessendorf.cpp:284-455),
// three launches, 108 algorithmic HBM bytes per texel:
//
//   k_rows        animate h~(k,t) (.h:265-275), build the Hermitian-symmetrised
//                 spectra of the seven real output fields packed into three
//                 complex pairs + the height alone, row (x-axis) inverse FFT.
//                 reads 12 B/texel (h0 8 + omega 4), writes 28 B/texel.
//   k_cols_height column (z-axis) inverse FFT of the height (two real columns
//                 per complex transform), (-1)^(m+n) sign, global min/max by
//                 atomics, raw signed height out.   reads 4, writes 4 B/texel.
//   k_cols_maps   column inverse FFT of the three pairs, sign, lambda, height
//                 normalisation, packs both RGBA32F maps.
//                 reads 24 + 4, writes 32 B/texel.
//
// The seven 2-D FFTs of the reference (.cpp:338-367) collapse to 3.5 complex
// ones because only real parts are consumed (.cpp:380-437):
//   Re B[X] = B[X_h],  X_h(k) = (X(k) + conj X(-k)) / 2,
// and two real-output fields p, q share one transform B[P_h + i Q_h] = p + i q.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fft_engine.h"

namespace ocean {

struct TileParams {          // device copy of one tile's properties
    float wind_x, wind_y;    // unit vector (SetWindDirection, .cpp:476-479)
    float wind_speed;        // (.cpp:481-484)
    float phillips_a;        // (.cpp:492-495)
    float damping;           // (.cpp:502-505)
    float base_freq;         // 2 pi / T as float (.cpp:486-490)
    float length;            // tile length L
    float pad_;
    uint64_t seed;
};

struct FrameArgs {
    const float2* h0;        // [tiles][N][N]   base amplitudes h0(k)
    const float* omega;      // [tiles][N][N]   quantised dispersion
    const float* k1d;        // [tiles][N]      k(i) = float(pi*(2i-N)/L)
    const float2* tw;        // [N]             exp(+2 pi i k / N)
    float2* z;               // [tiles][3][N][N] row-transformed pairs
    float2* zh;              // [tiles][N/2][N]  row-transformed height (rows 0..N/2-1)
    float* hraw;             // [tiles][N][N]    signed, un-normalised height
    unsigned* minmax;        // [tiles][2]       ordered-int keys of min, max
    float4* disp;            // [tiles][N][N]
    float4* nrm;             // [tiles][N][N]
    const float* toff;       // [tiles] or null
    const float* lambda;     // [tiles]
    float t;
    unsigned long long* stamps;   // diagnostic builds only (-DOCEAN_STAMPS), else null
};

#ifdef OCEAN_STAMPS
#define OCEAN_STAMP(k) do { if (threadIdx.x == 0 && a.stamps) a.stamps[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 16 + (k)] = clock64(); } while (0)
#else
#define OCEAN_STAMP(k) do {} while (0)
#endif

// ---- float <-> order-preserving unsigned key (for atomicMin/atomicMax) -----
__host__ __device__ inline unsigned float_key(float f)
{
    union { float f; unsigned u; } v; v.f = f;
    return (v.u & 0x80000000u) ? ~v.u : (v.u | 0x80000000u);
}
__host__ __device__ inline float key_float(unsigned k)
{
    union { float f; unsigned u; } v;
    v.u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return v.f;
}

// ============================================================================
// Prepare(): wave vectors (.cpp:60-85), gaussian draws (.cpp:87-103, RNG
// replaced by a counter-based one), base spectrum + dispersion (.cpp:105-148).
// No FMA contraction here: omega goes through floor() and must match the fp32
// evaluation order of the reference.
// ============================================================================
__device__ inline uint64_t splitmix64(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed + (idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ inline float2 gauss_pair(uint64_t seed, uint64_t idx)
{
    const uint64_t z = splitmix64(seed, idx);
    const double u1 = ((double)(z >> 40) + 1.0) * (1.0 / 16777216.0);
    const double u2 = (double)((z >> 8) & 0xFFFFFFull) * (1.0 / 16777216.0);
    const double r = sqrt(-2.0 * log(u1));
    const double a = 6.283185307179586476925286766559 * u2;
    double s, c;
    sincos(a, &s, &c);
    return make_float2((float)(r * c), (float)(r * s));
}

__global__ void k_init_k1d(float* __restrict__ k1d, const TileParams* __restrict__ tp, int n)
{
    const int tile = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // M_PI * (2.0f*i - kSize) / kLength : float numerator, double product/quotient (.cpp:76-79)
    const float num = 2.0f * (float)i - (float)n;
    k1d[(size_t)tile * n + i] =
        (float)(3.14159265358979323846 * (double)num / (double)tp[tile].length);
}

__device__ inline float phillips_nc(const TileParams& p, float ux, float uz, float k)
{
#pragma clang fp contract(off)
    // WSTessendorf.h:249-263
    const float k2 = k * k;
    const float k4 = k2 * k2;
    float cf = ux * p.wind_x + uz * p.wind_y;
    cf = cf * cf;
    const float lw = p.wind_speed * p.wind_speed / 9.81f;
    const float l2 = lw * lw;
    return p.phillips_a * expf(-1.0f / (k2 * l2)) / k4 * cf * expf(-k2 * p.damping * p.damping);
}

__global__ void k_init_spectrum(float2* __restrict__ h0, float* __restrict__ omega, float2* __restrict__ xi_out,
                                const float2* __restrict__ xi_in, const float* __restrict__ k1d,
                                const TileParams* __restrict__ tp, int n)
{
#pragma clang fp contract(off)
    const int tile = blockIdx.y;
    const size_t n2 = (size_t)n * n;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const TileParams p = tp[tile];
    const int m = (int)(i / n), q = (int)(i % n);
    const float kx = k1d[(size_t)tile * n + q], kz = k1d[(size_t)tile * n + m];
    const float d = kx * kx + kz * kz;
    const float k = sqrtf(d);
    const float2 g = xi_in ? xi_in[tile * n2 + i] : gauss_pair(p.seed, i);
    if (xi_out) xi_out[tile * n2 + i] = g;
    float2 a = make_float2(0.f, 0.f);
    float w = 0.f;
    if (k > 0.00001f) {
        const float inv = 1.0f / sqrtf(d);            // glm::normalize (.h:133-136)
        const float ux = kx * inv, uz = kz * inv;
        const float sp = sqrtf(phillips_nc(p, ux, uz, k));
        const float s = 1.0f / sqrtf(2.0f);
        a.x = (s * g.x) * sp;                         // .h:237-243
        a.y = (s * g.y) * sp;
        w = floorf(sqrtf(9.81f * k) / p.base_freq) * p.base_freq;   // .h:284-297
    }
    h0[tile * n2 + i] = a;
    omega[tile * n2 + i] = w;
}

// ============================================================================
// h~(k, t): WaveHeightFT (.h:265-275).  conj(h0(-k)) of the reference equals
// conj(h0(k)) (same gaussian draw, Phillips even in k: .cpp:131-135), so
//   h~ = h0 e^{i wt} + conj(h0) e^{-i wt} = 2 (Re h0 cos wt - Im h0 sin wt)  exactly real.
// omega*t is ONE fp32 multiply like the reference; sincosf is the accurate
// (Payne-Hanek backed) one since wt reaches 1e4 rad.
// ============================================================================
__device__ __forceinline__ float animate(float h0r, float h0i, float w, float t)
{
#pragma clang fp contract(off)
    const float wt = w * t;
    float s, c;
#ifdef OCEAN_ABL_SINCOS
    s = wt * 1e-4f; c = 1.0f - s;
#else
    sincosf(wt, &s, &c);
#endif
    const float re = h0r * c - h0i * s;
    return re + re;
}

// ============================================================================
// k_rows: RP row pairs (r, N-r) per workgroup (pair 0 = the two self-mirrored
// rows 0 and N/2).  Slot s = 2*rr + side holds row r (side 0) or its mirror.
// ============================================================================
#ifndef OCEAN_ROWS_MINW
#define OCEAN_ROWS_MINW 1
#endif
template <int N, int RP, int T, class P = Plan<N>>
__global__ void __launch_bounds__(T, OCEAN_ROWS_MINW) k_rows(const FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int C = 2 * RP;
    c32* fbuf = reinterpret_cast<c32*>(smem);
    constexpr int HS = N + 16;   // row stride of hs: +16 banks between the two rows of a pair
    float* hs = reinterpret_cast<float*>(fbuf + fft_lds_elems<N, C>());   // [C][HS]

    const int tid = threadIdx.x;
    const int tile = blockIdx.y;
    const int r0 = blockIdx.x * RP;
    const size_t n2 = (size_t)N * N;
    const float2* __restrict__ h0 = a.h0 + tile * n2;
    const float* __restrict__ om = a.omega + tile * n2;
    const float t = a.t + (a.toff ? a.toff[tile] : 0.0f);
    const float* __restrict__ k1 = a.k1d + (size_t)tile * N;              // [N], L1-resident table

    auto row_of = [&](int s) {
        const int r = r0 + (s >> 1);
        return (s & 1) ? (r == 0 ? N / 2 : N - r) : r;
    };

    OCEAN_STAMP(0);
    // -- phase 1: animate both rows of every pair into LDS -------------------
    // all global loads of the workgroup's input are issued before the first
    // sincos: one HBM round trip per workgroup instead of one per iteration
    {
        constexpr int P1 = (C * (N / 2)) / T;
        static_assert((C * (N / 2)) % T == 0, "phase-1 tiling");
        constexpr int PB = P1 > 4 ? 4 : P1;          // loads in flight per thread per batch (6 VGPRs each)
        static_assert(P1 % PB == 0, "phase-1 batches");
#pragma unroll 1
        for (int ub = 0; ub < P1; ub += PB) {
            float4 hv[PB];
            float2 wv[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int e = tid + (ub + u) * T;
                const int s = e / (N / 2);
                const int n = (e % (N / 2)) * 2;
                const size_t g = (size_t)row_of(s) * N + n;
#ifdef OCEAN_ABL_NOLOAD
                hv[u] = make_float4(1.f + g, 2.f, 3.f, 4.f); wv[u] = make_float2(0.5f, 0.25f);
#else
                hv[u] = *reinterpret_cast<const float4*>(h0 + g);
                wv[u] = *reinterpret_cast<const float2*>(om + g);
#endif
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int e = tid + (ub + u) * T;
                const int s = e / (N / 2);
                const int n = (e % (N / 2)) * 2;
                float2 v;
                v.x = animate(hv[u].x, hv[u].y, wv[u].x, t);
                v.y = animate(hv[u].z, hv[u].w, wv[u].y, t);
                *reinterpret_cast<float2*>(hs + s * HS + n) = v;
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        // min starts at FLT_MAX, max at FLT_MIN (> 0): WSTessendorf.cpp:289-290
        a.minmax[2 * tile + 0] = float_key(3.402823466e+38f);
        a.minmax[2 * tile + 1] = float_key(1.175494351e-38f);
    }
    __syncthreads();
    OCEAN_STAMP(1);

    // -- phase 2: three packed pairs, C row transforms each ------------------
    // per element: a = h~(k), b = h~(-k);  -k <-> ((N-m)%N, (N-n)%N), and
    // k(-idx) = -k(idx) except on the Nyquist row/column 0 (self-mirrored).
#pragma unroll 1
    for (int g = 0; g < 3; ++g) {
        float2* __restrict__ zg = a.z + ((size_t)tile * 3 + g) * n2;
        auto in = [&](int n, int s) -> c32 {
            const int r = r0 + (s >> 1);
            const int ms = (r == 0) ? s : (s ^ 1);
            const float av = hs[s * HS + n];
            const float bv = hs[ms * HS + ((N - n) & (N - 1))];
#ifdef OCEAN_ABL_NOIN
            return make_float2(av, bv);
#endif
            const float kxa = k1[n];
            const float kza = k1[row_of(s)];
            const float kxb = (n == 0) ? kxa : -kxa;
            const float kzb = (r == 0) ? kza : -kza;
            if (g == 1) {
                // slopes: c = i k  ->  X_h = i * (k a - kbar b)/2 ; pair = sx_h + i sz_h
                const float sx = 0.5f * (kxa * av - kxb * bv);
                const float sz = 0.5f * (kza * av - kzb * bv);
                return make_float2(-sz, sx);
            }
            const float d = kxa * kxa + kza * kza;
            const float inv = d > 1e-10f ? rsqrtf(d) : 0.0f;     // |k| > 1e-5 (.h:135)
            const float uxa = kxa * inv, uza = kza * inv;
            const float uxb = (n == 0) ? uxa : -uxa;
            const float uzb = (r == 0) ? uza : -uza;
            if (g == 0) {
                // displacements: c = -i u -> X_h = i * (ubar b - u a)/2 ; pair = Dx_h + i Dz_h
                const float dx = 0.5f * (uxb * bv - uxa * av);
                const float dz = 0.5f * (uzb * bv - uza * av);
                return make_float2(-dz, dx);
            }
            // derivatives: c = k u (real) ; pair = dxDx_h + i dzDz_h
            const float e = 0.5f * (kxa * uxa * av + kxb * uxb * bv);
            const float f = 0.5f * (kza * uza * av + kzb * uzb * bv);
            return make_float2(e, f);
        };
#ifdef OCEAN_ABL_NOSTORE
        auto out = [&](int q, int s, c32 v, int, int) { asm volatile("" ::"v"(v.x), "v"(v.y)); if (q < 0) zg[q] = v; };
#else
        auto out = [&](int q, int s, c32 v, int, int) { zg[(size_t)row_of(s) * N + q] = v; };
#endif
        batch_fft<N, C, T, P>(fbuf, a.tw, tid, in, out);
        OCEAN_STAMP(2 + g);
    }

    // -- height alone: one transform per pair, rows 0..N/2-1 ------------------
    {
        float2* __restrict__ zh = a.zh + (size_t)tile * (N / 2) * N;
        auto in = [&](int n, int rr) -> c32 {
            const int r = r0 + rr;
            const int nm = (N - n) & (N - 1);
            if (r == 0)   // rows 0 and N/2: both real-symmetric, packed as re + i im
                return make_float2(0.5f * (hs[n] + hs[nm]), 0.5f * (hs[HS + n] + hs[HS + nm]));
            return make_float2(0.5f * (hs[(2 * rr) * HS + n] + hs[(2 * rr + 1) * HS + nm]), 0.0f);
        };
        auto out = [&](int q, int rr, c32 v, int, int) { zh[(size_t)(r0 + rr) * N + q] = v; };
        batch_fft<N, RP, T, P>(fbuf, a.tw, tid, in, out);
        OCEAN_STAMP(5);
    }
}

template <int N, int RP> constexpr size_t rows_lds_bytes()
{
    return sizeof(c32) * fft_lds_elems<N, 2 * RP>() + sizeof(float) * (2 * RP * (N + 16));
}

// XCD-aware panel order: workgroups are dealt round-robin over the 8 XCDs, so
// give the workgroups that share an XCD (same id % 8) consecutive panels: the
// 128-byte lines two neighbouring panels share are then fetched into one L2.
__device__ __forceinline__ int xcd_swizzle(int id, int n)
{
    if (n % 8 != 0) return id;
    return (id % 8) * (n / 8) + id / 8;
}

// ============================================================================
// k_cols_height: 2*CP columns per workgroup, two adjacent real columns per
// complex transform.  Column input Y_q(m) = Zh(m, q) for m < N/2 and its
// conjugate mirror above; rows 0 and N/2 (real) are packed in Zh row 0.
// ============================================================================
template <int N, int CP, int T, class P = Plan<N>>
__global__ void __launch_bounds__(T) k_cols_height(const FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* fbuf = reinterpret_cast<c32*>(smem);
    constexpr int NW = (T + 63) / 64;
    float* red = reinterpret_cast<float*>(fbuf + fft_lds_elems<N, CP>());   // [2][NW], behind the FFT image
    const int tid = threadIdx.x;
    const int tile = blockIdx.y;
    const int q0 = xcd_swizzle(blockIdx.x, gridDim.x) * (2 * CP);
    const float2* __restrict__ zh = a.zh + (size_t)tile * (N / 2) * N;
    float* __restrict__ hraw = a.hraw + (size_t)tile * N * N;

    float vmin = 3.402823466e+38f, vmax = -3.402823466e+38f;
    auto in = [&](int m, int c) -> c32 {
        const int row = (m == 0 || m == N / 2) ? 0 : (m < N / 2 ? m : N - m);
        const float4 z = *reinterpret_cast<const float4*>(zh + (size_t)row * N + q0 + 2 * c);
        if (m == 0) return make_float2(z.x, z.z);
        if (m == N / 2) return make_float2(z.y, z.w);
        if (m < N / 2) return make_float2(z.x - z.w, z.y + z.z);     // Z(q) + i Z(q+1)
        return make_float2(z.x + z.w, z.z - z.y);                      // conj Z(q) + i conj Z(q+1)
    };
    auto out = [&](int p, int c, c32 v, int, int) {
        const int q = q0 + 2 * c;
        const float s = ((p + q) & 1) ? -1.0f : 1.0f;                  // .cpp:388-390
        const float h0v = s * v.x, h1v = -s * v.y;
        vmin = fminf(vmin, fminf(h0v, h1v));
        vmax = fmaxf(vmax, fmaxf(h0v, h1v));
        *reinterpret_cast<float2*>(hraw + (size_t)p * N + q) = make_float2(h0v, h1v);
    };
    batch_fft<N, CP, T, P>(fbuf, a.tw, tid, in, out);

    // workgroup reduction -> one atomic pair (.cpp:391-392, 407-411)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        vmin = fminf(vmin, __shfl_xor(vmin, o));
        vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    }
    if ((tid & 63) == 0) { red[tid >> 6] = vmin; red[NW + (tid >> 6)] = vmax; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < NW; ++w) { vmin = fminf(vmin, red[w]); vmax = fmaxf(vmax, red[NW + w]); }
        atomicMin(a.minmax + 2 * tile + 0, float_key(vmin));
        atomicMax(a.minmax + 2 * tile + 1, float_key(vmax));
    }
}

// ============================================================================
// k_cols_maps: C columns per workgroup.  blockIdx.z = 0: displacement map
// (pair 0 + height), 1: normal map (pairs 1 and 2).
// ============================================================================
template <int N, int C, int T, class P = Plan<N>>
__global__ void __launch_bounds__(T) k_cols_maps(const FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* fbuf = reinterpret_cast<c32*>(smem);
    using LS = LastStage<N, C, T, P>;
    const int tid = threadIdx.x;
    const int tile = blockIdx.y;
    const int q0 = xcd_swizzle(blockIdx.x, gridDim.x) * C;
    const size_t n2 = (size_t)N * N;
    const float2* __restrict__ zt = a.z + (size_t)tile * 3 * n2;

    if (blockIdx.z == 0) {
        const float* __restrict__ hraw = a.hraw + tile * n2;
        float4* __restrict__ disp = a.disp + tile * n2;
        // raw heights of the texels this thread will finish, fetched up front
        float hv[LS::IT][LS::RL];
#pragma unroll
        for (int u = 0; u < LS::IT; ++u) {
            const int w = tid + u * T;
            if (!LS::GUARD || w < LS::ITEMS) {
                const int c = w % C, j = w / C;
#pragma unroll
                for (int i = 0; i < LS::RL; ++i) hv[u][i] = hraw[(size_t)(j + i * LS::STRIDE) * N + q0 + c];
            }
        }
        // NormalizeHeights (.cpp:443-455): A = max(|min|, |max|), y *= 1/A
        const float mn = key_float(a.minmax[2 * tile + 0]);
        const float mx = key_float(a.minmax[2 * tile + 1]);
        const float inv_a = 1.0f / fmaxf(fabsf(mn), fabsf(mx));
        const float lambda = a.lambda[tile];
        const float2* __restrict__ z0 = zt;
        auto in = [&](int m, int c) -> c32 { return z0[(size_t)m * N + q0 + c]; };
        auto out = [&](int p, int c, c32 v, int u, int i) {
            const int q = q0 + c;
            const float s = ((p + q) & 1) ? -1.0f : 1.0f;
            // (sign*lambda)*Re Dx, h/A, (sign*lambda)*Re Dz, 1   (.cpp:394-403)
            disp[(size_t)p * N + q] = make_float4(s * lambda * v.x, hv[u][i] * inv_a, s * lambda * v.y, 1.0f);
        };
        batch_fft<N, C, T, P>(fbuf, a.tw, tid, in, out);
    } else {
        float4* __restrict__ nrm = a.nrm + tile * n2;
        c32 held[LS::IT][LS::RL];
        {
            const float2* __restrict__ z1 = zt + n2;
            auto in = [&](int m, int c) -> c32 { return z1[(size_t)m * N + q0 + c]; };
            auto out = [&](int, int, c32 v, int u, int i) { held[u][i] = v; };
            batch_fft<N, C, T, P>(fbuf, a.tw, tid, in, out);
        }
        {
            const float2* __restrict__ z2 = zt + 2 * n2;
            auto in = [&](int m, int c) -> c32 { return z2[(size_t)m * N + q0 + c]; };
            auto out = [&](int p, int c, c32 v, int u, int i) {
                const int q = q0 + c;
                const float s = ((p + q) & 1) ? -1.0f : 1.0f;
                // (slope x, slope z, dDx/dx, dDz/dz) * sign   (.cpp:430-435)
                nrm[(size_t)p * N + q] = make_float4(s * held[u][i].x, s * held[u][i].y, s * v.x, s * v.y);
            };
            batch_fft<N, C, T, P>(fbuf, a.tw, tid, in, out);
        }
    }
}

// ---- per-size launch geometry ---------------------------------------------------
template <int N> struct Geo;
#define OCEAN_GEO(n, rp, tr, pr, cp, th, ph, cm, tm, pm)                                       \
    template <> struct Geo<n> {                                                                \
        static constexpr int RP = rp, T_ROWS = tr;      /* row pairs per workgroup, threads */ \
        static constexpr int CP = cp, T_H = th;         /* column pairs (height), threads   */ \
        static constexpr int CM = cm, T_M = tm;         /* columns (maps), threads          */ \
        using PR = pr; using PH = ph; using PM = pm;    /* radix plans                      */ \
    };
#define OCEAN_R(...) Radices<__VA_ARGS__>
OCEAN_GEO(16, 8, 64, Plan<16>, 8, 64, Plan<16>, 16, 64, Plan<16>)
OCEAN_GEO(32, 8, 64, Plan<32>, 8, 64, Plan<32>, 16, 64, Plan<32>)
OCEAN_GEO(64, 4, 64, Plan<64>, 8, 64, Plan<64>, 16, 128, Plan<64>)
OCEAN_GEO(128, 4, 64, Plan<128>, 8, 64, Plan<128>, 16, 128, Plan<128>)
OCEAN_GEO(256, 4, 128, Plan<256>, 8, 128, Plan<256>, 16, 256, Plan<256>)
OCEAN_GEO(512, 1, 128, Plan<512>, 2, 128, Plan<512>, 8, 256, Plan<512>)
OCEAN_GEO(1024, 1, 128, Plan<1024>, 4, 256, Plan<1024>, 8, 512, Plan<1024>)
#ifdef OCEAN_ROWS_R8
OCEAN_GEO(2048, 1, 512, OCEAN_R(8, 8, 8, 4), 4, 512, Plan<2048>, 8, 1024, Plan<2048>)
#else
OCEAN_GEO(2048, 1, 256, Plan<2048>, 4, 512, Plan<2048>, 8, 1024, Plan<2048>)
#endif
OCEAN_GEO(4096, 1, 512, Plan<4096>, 2, 512, Plan<4096>, 4, 1024, Plan<4096>)
#undef OCEAN_R
#undef OCEAN_GEO

}  // namespace ocean
