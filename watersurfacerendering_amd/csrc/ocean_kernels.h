// ocean_kernels.h -- device kernels of the MI355X ocean synthesiser (gfx950).
//
// One ComputeWaves(t) (reference WSTessendorf.cpp:284-455) = three launches.
//
// Structure the reference's arithmetic has and this pipeline exploits:
//  (1) h~(k,t) is exactly REAL for every k (the reference conjugates h0(k), not
//      h0(-k): .cpp:131-135, .h:265-275), and only real parts of the seven
//      inverse FFTs are consumed (.cpp:380-437).  So  Re B[c_f * h~] = B[X_f]
//      with the Hermitian part X_f(k) = (c_f(k) h~(k) + conj(c_f(-k)) h~(-k))/2,
//      and two real-output fields share one complex transform.
//  (2) every X_f is even or odd under index inversion (m,n) -> (-m,-n) mod N,
//      hence so is every output field: out(-p,-q) = eps_f * out(p,q)
//      (height, dDx/dx, dDz/dz even; Dx, Dz, slopes odd).  Only rows 0..N/2 of
//      the row pass and columns 0..N/2 of the column pass are transformed; the
//      other half is the mirror image.  (Checked bit-exactly against the
//      oracle: tests/test_oracle.py::test_reference_output_point_symmetry.)
//
//   k_rows      one spectrum row m in [0, N/2] per workgroup: animate rows m and
//               -m, S+ = (a+b)/2, S- = (a-b)/2, build the three packed pairs and
//               the height from S+/S- and the wave-vector coefficients, four
//               row (x-axis) inverse FFTs.
//   k_cols_b    column (z-axis) pass, part 1, one launch, two kinds of
//               workgroup: HEIGHT (two real columns per complex transform,
//               sign, raw heights out, global min/max by atomics) and NORMAL
//               (pairs 1 and 2 -> finished normal map, both mirror halves).
//   k_cols_disp part 2, needs the min/max: pair 0 + raw height -> displacement map.
//
// HBM bytes per texel actually moved (this pipeline): 12 (h0, omega) + 14 + 14
// (half-size intermediates out and in) + 2 + 2 (raw height) + 32 (maps) = 76,
// against 108 for the straightforward 3.5-transform two-pass scheme the
// roofline accounting of SURVEY.md section 8d assumes.
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
    float2* z;               // [tiles][3][N/2+1][2][NUP] row-transformed pairs: row m, side 0 = columns
                             //   u = 0..N/2, side 1 = columns (N-u)%N, NUP = N/2 + 8 (padded)
    float2* zh;              // [tiles][N/2+1][NUP]      row-transformed height, columns 0..N/2
    float* hraw;             // [tiles][NUP/8][N][8]     signed raw height, columns 0..N/2, 8-column tiles
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

// ---- half-spectrum storage geometry -------------------------------------------
template <int N> struct Half {
    static constexpr int NU = N / 2 + 1;          // columns (units) / rows kept: 0..N/2
    static constexpr int NUP = N / 2 + 8;         // padded to a multiple of 8
    static constexpr size_t Z_GROUP = (size_t)NU * 2 * NUP;       // float2 per packed pair
    static constexpr size_t Z_TILE = 3 * Z_GROUP;
    static constexpr size_t ZH_TILE = (size_t)NU * NUP;
    static constexpr size_t HRAW_TILE = (size_t)NUP * N;          // floats
};
__device__ __forceinline__ size_t hraw_index(int n, int p, int u) { return ((size_t)(u >> 3) * n + p) * 8 + (u & 7); }

// XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so give the
// workgroups that share an XCD (same id % 8) consecutive column blocks: lines
// shared by neighbouring blocks (input sectors, mirrored map rows) meet in one L2.
__device__ __forceinline__ int xcd_swizzle(int id, int n)
{
    const int q = n / 8, r = n % 8, x = id % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + id / 8;
}

// ============================================================================
// k_rows: workgroup = spectrum row m (blockIdx.x in [0, N/2]); mirror row
// mb = (N-m)%N.  a(n) = h~(m,n), b(n) = h~(mb,(N-n)%N).
//   S+ = (a+b)/2, S- = (a-b)/2;   Tx = (n==0 ? S- : S+), Tz = (m==0 ? S- : S+)
//   (k(-idx) = -k(idx) except on the self-mirrored Nyquist row/column 0)
//   pair 0: Dx_h + i Dz_h     = ( uz*Tz, -ux*Tx)      odd   (c = -i u, .cpp:323-326)
//   pair 1: sx_h + i sz_h     = (-kz*Tz,  kx*Tx)      odd   (c =  i k, .cpp:309-310)
//   pair 2: dxDx_h + i dzDz_h = (kx*ux*S+, kz*uz*S+)  even  (.cpp:327-330)
//   height: S+                                         even
// Two batches of two interleaved transforms: {pair 0, pair 1}, {pair 2, height}.
// ============================================================================
template <int N, int T, class P = Plan<N>>
__global__ void __launch_bounds__(T) k_rows(const FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using HF = Half<N>;
    c32* fbuf = reinterpret_cast<c32*>(smem);                              // 2 interleaved transforms
    float* sp = reinterpret_cast<float*>(fbuf + fft_lds_elems<N, 2>());    // S+ [N]
    float* sm = sp + N;                                                    // S- [N]
    float* raw = reinterpret_cast<float*>(fbuf);                           // h~ rows m, mb (before the FFTs)

    const int tid = threadIdx.x;
    const int tile = blockIdx.y;
    const int m = blockIdx.x;
    const int mb = (N - m) & (N - 1);
    const size_t n2 = (size_t)N * N;
    const float2* __restrict__ h0 = a.h0 + tile * n2;
    const float* __restrict__ om = a.omega + tile * n2;
    const float t = a.t + (a.toff ? a.toff[tile] : 0.0f);
    const float* __restrict__ k1 = a.k1d + (size_t)tile * N;               // [N], cache-resident table

    OCEAN_STAMP(0);
    // -- phase 1: animate rows m and mb; all loads issued before the first sincos
    {
        constexpr int ELEMS = N;                      // 2 rows * N/2 texel pairs
        constexpr int P1 = (ELEMS + T - 1) / T;
        constexpr int PB = P1 > 4 ? 4 : P1;
        static_assert(P1 % PB == 0, "phase-1 batches");
#pragma unroll 1
        for (int ub = 0; ub < P1; ub += PB) {
            float4 hv[PB];
            float2 wv[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int e = tid + (ub + u) * T;
                if (ELEMS % T == 0 || e < ELEMS) {
                    const int s = e / (N / 2);
                    const int n = (e % (N / 2)) * 2;
                    const size_t g = (size_t)(s ? mb : m) * N + n;
#ifdef OCEAN_ABL_NOLOAD
                    hv[u] = make_float4(1.f + g, 2.f, 3.f, 4.f); wv[u] = make_float2(0.5f, 0.25f);
#else
                    hv[u] = *reinterpret_cast<const float4*>(h0 + g);
                    wv[u] = *reinterpret_cast<const float2*>(om + g);
#endif
                }
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int e = tid + (ub + u) * T;
                if (ELEMS % T == 0 || e < ELEMS) {
                    const int s = e / (N / 2);
                    const int n = (e % (N / 2)) * 2;
                    float2 v;
                    v.x = animate(hv[u].x, hv[u].y, wv[u].x, t);
                    v.y = animate(hv[u].z, hv[u].w, wv[u].y, t);
                    *reinterpret_cast<float2*>(raw + s * N + n) = v;
                }
            }
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        // min starts at FLT_MAX, max at FLT_MIN (> 0): WSTessendorf.cpp:289-290
        a.minmax[2 * tile + 0] = float_key(3.402823466e+38f);
        a.minmax[2 * tile + 1] = float_key(1.175494351e-38f);
    }
    __syncthreads();
    for (int n = tid; n < N; n += T) {
        const float av = raw[n], bv = raw[N + ((N - n) & (N - 1))];
        sp[n] = 0.5f * (av + bv);
        sm[n] = 0.5f * (av - bv);
    }
    __syncthreads();
    OCEAN_STAMP(1);

    const float kz = k1[m];
    const float kz2 = kz * kz;
    const bool row0 = (m == 0);
    float2* __restrict__ zt = a.z + (size_t)tile * HF::Z_TILE + (size_t)m * 2 * HF::NUP;
    float2* __restrict__ zh = a.zh + (size_t)tile * HF::ZH_TILE + (size_t)m * HF::NUP;

    // -- batch A: slot 0 = pair 0 (Dx, Dz), slot 1 = pair 1 (sx, sz) ------------
    {
        auto in = [&](int n, int c) -> c32 {
            const float sv = sp[n];
#ifdef OCEAN_ABL_NOIN
            return make_float2(sv, sv);
#endif
            const float dv = sm[n];
            const float tx = (n == 0) ? dv : sv;
            const float tz = row0 ? dv : sv;
            const float kx = k1[n];
            const float d = kx * kx + kz2;
            const float inv = d > 1e-10f ? rsqrtf(d) : 0.0f;      // |k| > 1e-5 (.h:135)
            const float cz = c ? -kz : kz * inv;                   // -kz | uz
            const float cx = c ? kx : -kx * inv;                   //  kx | -ux
            return make_float2(cz * tz, cx * tx);
        };
        auto out = [&](int q, int c, c32 v, int, int) {
#ifdef OCEAN_ABL_NOSTORE
            asm volatile("" ::"v"(v.x), "v"(v.y)); if (q >= 0) return;
#endif
            float2* __restrict__ zg = zt + (size_t)c * HF::Z_GROUP;
            if (q <= N / 2) {
                zg[q] = v;
                if (q == 0 || q == N / 2) zg[HF::NUP + q] = v;     // self-mirrored columns: both sides
            } else {
                zg[HF::NUP + (N - q)] = v;
            }
        };
        batch_fft<N, 2, T, P>(fbuf, a.tw, tid, in, out);
        OCEAN_STAMP(2);
    }
    // -- batch B: slot 0 = pair 2 (dDx/dx, dDz/dz), slot 1 = height ----------------
    {
        auto in = [&](int n, int c) -> c32 {
            const float sv = sp[n];
#ifdef OCEAN_ABL_NOIN
            return make_float2(sv, sv);
#endif
            if (c) return make_float2(sv, 0.0f);
            const float kx = k1[n];
            const float d = kx * kx + kz2;
            const float inv = d > 1e-10f ? rsqrtf(d) : 0.0f;
            return make_float2(kx * kx * inv * sv, kz2 * inv * sv);
        };
        auto out = [&](int q, int c, c32 v, int, int) {
#ifdef OCEAN_ABL_NOSTORE
            asm volatile("" ::"v"(v.x), "v"(v.y)); if (q >= 0) return;
#endif
            if (c) {
                if (q <= N / 2) zh[q] = v;                          // real input: other half is the conjugate
                return;
            }
            float2* __restrict__ zg = zt + (size_t)2 * HF::Z_GROUP;
            if (q <= N / 2) {
                zg[q] = v;
                if (q == 0 || q == N / 2) zg[HF::NUP + q] = v;
            } else {
                zg[HF::NUP + (N - q)] = v;
            }
        };
        batch_fft<N, 2, T, P>(fbuf, a.tw, tid, in, out);
        OCEAN_STAMP(3);
    }
}

template <int N> constexpr size_t rows_lds_bytes()
{
    return sizeof(c32) * fft_lds_elems<N, 2>() + sizeof(float) * 2 * N;
}

// ---- column pass helpers -----------------------------------------------------------
// Column u of a packed pair: rows 0..N/2 come from side 0; row mf > N/2 is the
// mirror image eps * Z(N-mf, N-u) = eps * side 1 of row N-mf.
template <int N>
__device__ __forceinline__ c32 load_pair_column(const float2* __restrict__ zg, int mf, int u, float eps)
{
    using HF = Half<N>;
    if (mf <= N / 2) return zg[(size_t)mf * 2 * HF::NUP + u];
    const c32 v = zg[((size_t)(N - mf) * 2 + 1) * HF::NUP + u];
    return make_float2(eps * v.x, eps * v.y);
}

// ============================================================================
// k_cols_b: blockIdx.x < HB  -> HEIGHT workgroup: 2*C columns (C transforms of two
//                                real columns each), raw signed height + min/max
//           otherwise        -> NORMAL workgroup: C columns of pairs 1 and 2 ->
//                                normal map texels (p,u) and mirror (-p,-u)
// ============================================================================
template <int N, int C, int T, class P = Plan<N>>
__global__ void __launch_bounds__(T) k_cols_b(const FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using HF = Half<N>;
    using LS = LastStage<N, C, T, P>;
    c32* fbuf = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int tile = blockIdx.y;
    constexpr int HB = HF::NUP / (2 * C);                 // height workgroups
    constexpr int NB = (HF::NU + C - 1) / C;              // normal workgroups
    static_assert(HF::NUP % (2 * C) == 0, "height column blocks");

    if (blockIdx.x < HB) {
        constexpr int NW = (T + 63) / 64;
        float* red = reinterpret_cast<float*>(fbuf + fft_lds_elems<N, C>());
        const int u0 = blockIdx.x * 2 * C;
        const float2* __restrict__ zh = a.zh + (size_t)tile * HF::ZH_TILE;
        float* __restrict__ hraw = a.hraw + (size_t)tile * HF::HRAW_TILE;
        float vmin = 3.402823466e+38f, vmax = -3.402823466e+38f;
        // Y_u(mf) = Zh(mf,u), mf <= N/2; conj Zh(N-mf,u) above (real, even spectrum);
        // rows 0 and N/2 are real.  Two columns u, u+1 per transform: Y_u + i Y_{u+1}.
        auto in = [&](int mf, int c) -> c32 {
            const int row = mf <= N / 2 ? mf : N - mf;
            const float4 z = *reinterpret_cast<const float4*>(zh + (size_t)row * HF::NUP + u0 + 2 * c);
            if (mf == 0 || mf == N / 2) return make_float2(z.x, z.z);
            if (mf < N / 2) return make_float2(z.x - z.w, z.y + z.z);
            return make_float2(z.x + z.w, z.z - z.y);
        };
        auto out = [&](int p, int c, c32 v, int, int) {
            const int u = u0 + 2 * c;
            const float s = ((p + u) & 1) ? -1.0f : 1.0f;              // .cpp:388-390
            const float ha = s * v.x, hb = -s * v.y;
            if (u <= N / 2) { vmin = fminf(vmin, ha); vmax = fmaxf(vmax, ha); }
            if (u + 1 <= N / 2) { vmin = fminf(vmin, hb); vmax = fmaxf(vmax, hb); }
            *reinterpret_cast<float2*>(hraw + hraw_index(N, p, u)) = make_float2(ha, hb);
        };
        batch_fft<N, C, T, P>(fbuf, a.tw, tid, in, out);
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
        return;
    }

    // ---- NORMAL workgroup ------------------------------------------------------------
    const int u0 = xcd_swizzle(blockIdx.x - HB, NB) * C;
    const float2* __restrict__ z1 = a.z + (size_t)tile * HF::Z_TILE + HF::Z_GROUP;
    const float2* __restrict__ z2 = z1 + HF::Z_GROUP;
    float4* __restrict__ nrm = a.nrm + (size_t)tile * N * N;
    c32 held[LS::IT][LS::RL];
    {
        auto in = [&](int mf, int c) -> c32 { return load_pair_column<N>(z1, mf, u0 + c, -1.0f); };
        auto out = [&](int, int, c32 v, int u, int i) { held[u][i] = v; };
        batch_fft<N, C, T, P>(fbuf, a.tw, tid, in, out);
    }
    {
        auto in = [&](int mf, int c) -> c32 { return load_pair_column<N>(z2, mf, u0 + c, 1.0f); };
        auto out = [&](int p, int c, c32 v, int u, int i) {
            const int q = u0 + c;
            if (q > N / 2) return;                                      // padding column
            const float s = ((p + q) & 1) ? -1.0f : 1.0f;
            // (slope x, slope z, dDx/dx, dDz/dz) * sign   (.cpp:430-435)
            const float4 o = make_float4(s * held[u][i].x, s * held[u][i].y, s * v.x, s * v.y);
            nrm[(size_t)p * N + q] = o;
            if (q != 0 && q != N / 2)                                    // mirror texel: slopes odd, derivatives even
                nrm[(size_t)((N - p) & (N - 1)) * N + (N - q)] = make_float4(-o.x, -o.y, o.z, o.w);
        };
        batch_fft<N, C, T, P>(fbuf, a.tw, tid, in, out);
    }
}

// ============================================================================
// k_cols_disp: C columns of pair 0 + raw height -> displacement map texels (p,u)
// and mirror (-p,-u).  NormalizeHeights (.cpp:443-455) folded in.
// ============================================================================
template <int N, int C, int T, class P = Plan<N>>
__global__ void __launch_bounds__(T) k_cols_disp(const FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using HF = Half<N>;
    using LS = LastStage<N, C, T, P>;
    c32* fbuf = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int tile = blockIdx.y;
    constexpr int NB = (HF::NU + C - 1) / C;
    const int u0 = xcd_swizzle(blockIdx.x, NB) * C;
    const float2* __restrict__ z0 = a.z + (size_t)tile * HF::Z_TILE;
    const float* __restrict__ hraw = a.hraw + (size_t)tile * HF::HRAW_TILE;
    float4* __restrict__ disp = a.disp + (size_t)tile * N * N;

    // raw heights of the texels this thread will finish, fetched up front
    float hv[LS::IT][LS::RL];
#pragma unroll
    for (int u = 0; u < LS::IT; ++u) {
        const int w = tid + u * T;
        if (!LS::GUARD || w < LS::ITEMS) {
            const int c = w % C, j = w / C;
#pragma unroll
            for (int i = 0; i < LS::RL; ++i) hv[u][i] = hraw[hraw_index(N, j + i * LS::STRIDE, u0 + c)];
        }
    }
    // A = max(|min|, |max|), y *= 1/A
    const float mn = key_float(a.minmax[2 * tile + 0]);
    const float mx = key_float(a.minmax[2 * tile + 1]);
    const float inv_a = 1.0f / fmaxf(fabsf(mn), fabsf(mx));
    const float lambda = a.lambda[tile];
    auto in = [&](int mf, int c) -> c32 { return load_pair_column<N>(z0, mf, u0 + c, -1.0f); };
    auto out = [&](int p, int c, c32 v, int u, int i) {
        const int q = u0 + c;
        if (q > N / 2) return;
        const float s = ((p + q) & 1) ? -1.0f : 1.0f;
        // (sign*lambda)*Re Dx, h/A, (sign*lambda)*Re Dz, 1   (.cpp:394-403)
        const float4 o = make_float4(s * lambda * v.x, hv[u][i] * inv_a, s * lambda * v.y, 1.0f);
        disp[(size_t)p * N + q] = o;
        if (q != 0 && q != N / 2)                                        // mirror: Dx, Dz odd, height even
            disp[(size_t)((N - p) & (N - 1)) * N + (N - q)] = make_float4(-o.x, o.y, -o.z, 1.0f);
    };
    batch_fft<N, C, T, P>(fbuf, a.tw, tid, in, out);
}

// ---- per-size launch geometry ---------------------------------------------------
template <int N> struct Geo;
#define OCEAN_GEO(n, tr, pr, cc, tc, pc)                                                        \
    template <> struct Geo<n> {                                                                \
        static constexpr int T_ROWS = tr;               /* threads of k_rows               */  \
        static constexpr int CC = cc, T_C = tc;         /* transforms per column workgroup */  \
        using PR = pr; using PC = pc;                   /* radix plans                     */  \
    };
OCEAN_GEO(16, 64, Plan<16>, 4, 64, Plan<16>)
OCEAN_GEO(32, 64, Plan<32>, 4, 64, Plan<32>)
OCEAN_GEO(64, 64, Plan<64>, 4, 64, Plan<64>)
OCEAN_GEO(128, 64, Plan<128>, 4, 64, Plan<128>)
OCEAN_GEO(256, 64, Plan<256>, 4, 64, Plan<256>)
OCEAN_GEO(512, 128, Plan<512>, 4, 256, Plan<512>)
OCEAN_GEO(1024, 128, Plan<1024>, 4, 256, Plan<1024>)
OCEAN_GEO(2048, 256, Plan<2048>, 4, 512, Plan<2048>)
OCEAN_GEO(4096, 512, Plan<4096>, 4, 1024, Plan<4096>)
#undef OCEAN_GEO

}  // namespace ocean
