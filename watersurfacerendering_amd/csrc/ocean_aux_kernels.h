// ocean_aux_kernels.h -- the kernels around the frame path, compiled into ocean_api.hip only: Prepare() (wave vectors, gaussian draws, base
// spectrum and quantised dispersion: WSTessendorf.cpp:36-148), the fp16 copy of the spectrum and the bounds of the half2 intermediates, the
// half-precision pack and the copy kernel of the read-out, and the consumers of SURVEY.md 8f ranks 3-4 (vertex stage, cascades, mip chain).
// The frame kernels themselves are in ocean_kernels.h.
#pragma once
#include "ocean_kernels.h"

namespace ocean {

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

__global__ void k_init_spectrum(float2* __restrict__ h0, float* __restrict__ omega, uint16_t* __restrict__ omega_q,
                                float* __restrict__ base_freq, unsigned* __restrict__ omega_q_overflow, float2* __restrict__ xi_out,
                                const float2* __restrict__ xi_in, const float* __restrict__ k1d,
                                const TileParams* __restrict__ tp, int n)
{
#pragma clang fp contract(off)
    const int tile = blockIdx.y;
    const size_t n2 = (size_t)n * n;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const TileParams p = tp[tile];
    // the spectrum is stored TRANSPOSED: element i holds wave index (m, q) = (i % n, i / n),
    // so that a spectrum column (fixed kx) is one contiguous run for the first pass.
    // The gaussian draw of texel (m, q) keeps the reference's row-major index m*n + q.
    const int q = (int)(i / n), m = (int)(i % n);
    const size_t ref = (size_t)m * n + q;
    const float kx = k1d[(size_t)tile * n + q], kz = k1d[(size_t)tile * n + m];
    const float d = kx * kx + kz * kz;
    const float k = sqrtf(d);
    const float2 g = xi_in ? xi_in[tile * n2 + ref] : gauss_pair(p.seed, ref);
    if (xi_out) xi_out[tile * n2 + ref] = g;
    float2 a = make_float2(0.f, 0.f);
    float w = 0.f, steps = 0.f;
    if (i == 0) base_freq[tile] = p.base_freq;
    if (k > 0.00001f) {
        const float inv = 1.0f / sqrtf(d);            // glm::normalize (.h:133-136)
        const float ux = kx * inv, uz = kz * inv;
        const float sp = sqrtf(phillips_nc(p, ux, uz, k));
        const float s = 1.0f / sqrtf(2.0f);
        a.x = (s * g.x) * sp;                         // .h:237-243
        a.y = (s * g.y) * sp;
        float disp;                                   // the relation the reference calls, or one of the two it only defines
        if (p.dispersion == 1)        // sqrt(g k tanh(k D)): in double, rounded once (tanhf differs between libms)
            disp = (float)sqrt((double)(9.81f * k) * tanh((double)k * (double)p.dispersion_param));
        else if (p.dispersion == 2)   // sqrt(g k (1 + k^2 L^2))
            disp = sqrtf(9.81f * k * (1.0f + k * k * p.dispersion_param * p.dispersion_param));
        else
            disp = sqrtf(9.81f * k);
        steps = floorf(disp / p.base_freq);
        w = steps * p.base_freq;                         // .h:284-287
    }
    h0[tile * n2 + i] = a;
    omega[tile * n2 + i] = w;
    // omega is an integer multiple of base_freq: the frame kernels read that integer (2 bytes instead of
    // 4 per texel) and rebuild the same float, float(steps) * base_freq, unless some multiple needs more bits
    omega_q[tile * n2 + i] = (uint16_t)(steps < 65536.0f ? (unsigned)steps : 0u);
    if (!(steps < 65536.0f)) atomicOr(omega_q_overflow, 1u);
}

// fp16 spectrum variant (BASELINE config 4): h0 stored as half2 scaled per tile so
// that max|component| maps to 2^14 (keeps the small amplitudes normal numbers).
__global__ void k_h0_absmax(const float2* __restrict__ h0, unsigned* __restrict__ maxbits, size_t n2)
{
    const int tile = blockIdx.y;
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        const float2 v = h0[tile * n2 + i];
        m = fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(maxbits + tile, __float_as_uint(m));   // non-negative floats order like uints
}

__global__ void k_h0_to_half(const float2* __restrict__ h0, __half2* __restrict__ h0h, const unsigned* __restrict__ maxbits,
                             float* __restrict__ inv_scale, size_t n2)
{
    const int tile = blockIdx.y;
    const float m = __uint_as_float(maxbits[tile]);
    // power-of-two scale: exact to apply and to undo
    int e = 0;
    if (m > 0.0f) (void)frexpf(m, &e);                 // m = f * 2^e, f in [0.5, 1)
    const float scale = ldexpf(1.0f, 14 - e);
    if (blockIdx.x == 0 && threadIdx.x == 0) inv_scale[tile] = ldexpf(1.0f, e - 14);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
        const float2 v = h0[tile * n2 + i];
        h0h[tile * n2 + i] = __floats2half2_rn(v.x * scale, v.y * scale);
    }
}

// Bounds for the half2 intermediates (ocean_set_intermediate_precision(16)), per tile, time independent:
//   |h~(k, t)| <= 2 |h0(k)|, so every component of a z-pass output of spectrum column n is at most
//   2 * sum_e |h0(e, n)| + 2 * sum_e |h0(e, -n)|   for the fields weighted by unit vectors (pair 0, height), and the same
//   with |k| |h0| for the fields weighted by k (pairs 1 and 2).  One workgroup per spectrum column (contiguous in the
//   transposed layout) sums |h0| and |k| |h0|; the maxima over the columns go to bounds[tile][0..1] as float bits.
__global__ void k_inter_bounds(const float2* __restrict__ h0, const float* __restrict__ k1d, unsigned* __restrict__ bounds, int n)
{
    const int tile = blockIdx.y, col = blockIdx.x;
    const float2* __restrict__ c = h0 + ((size_t)tile * n + col) * n;
    const float* __restrict__ k1 = k1d + (size_t)tile * n;
    const float kx = k1[col];
    float su = 0.0f, sk = 0.0f;
    for (int e = threadIdx.x; e < n; e += blockDim.x) {
        const float2 v = c[e];
        const float m = sqrtf(v.x * v.x + v.y * v.y), kz = k1[e];
        su += m;
        sk += m * sqrtf(kx * kx + kz * kz);
    }
    __shared__ float ru[16], rk[16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { su += __shfl_xor(su, o); sk += __shfl_xor(sk, o); }
    if ((threadIdx.x & 63) == 0) { ru[threadIdx.x >> 6] = su; rk[threadIdx.x >> 6] = sk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tu = 0.0f, tk = 0.0f;
        for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) { tu += ru[w]; tk += rk[w]; }
        atomicMax(bounds + 2 * tile + 0, __float_as_uint(tu));      // non-negative floats order like uints
        atomicMax(bounds + 2 * tile + 1, __float_as_uint(tk));
    }
}


// Packed-map gather at half the bytes (the gather is xGMI-bound): one RGBA32F texel -> four halves
// (round to nearest even; |values| of both maps are far below the largest half, 65504, for any sea the reference
// parameters can describe -- larger values saturate to +-inf like any float -> half conversion).
__global__ void k_pack_half(const float4* __restrict__ src, uint2* __restrict__ dst, size_t texels)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < texels; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        const __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
        uint2 o;
        __builtin_memcpy(&o.x, &a, 4); __builtin_memcpy(&o.y, &b, 4);
        dst[i] = o;
    }
}

// ============================================================================
// Vertex-stage consumer: what the reference's vertex shader does with
// the two maps (src/shaders/WaterSurfaceMesh.vert:24-41) for the grid its mesh generator
// builds (WaterSurfaceMesh.cpp:500-533), as a kernel -- displaced positions and normals for a
// consumer that is not the Vulkan renderer.  Sampling is the sampler the reference creates
// (vulkan/Sampler.cpp:60-66): LINEAR filter, REPEAT addressing, unnormalised coordinate
// s = u*W - 0.5, texels floor(s) and floor(s)+1 (mod W), weights from frac(s), evaluated in
// fp32 in the order written below (no contraction), which oracle/consumer.py repeats.
// One thread per vertex; memory-bound (8 texel reads that mostly hit in cache, 2 writes).
// ============================================================================
struct GridArgs {
    const float4* disp;      // [N][N] of the tile
    const float4* nrm;
    const unsigned* minmax;  // keys of the tile's raw height range (A = max(|min|, |max|) = WSHeightAmp)
    float4* positions;       // [(g+1)^2]  xyz = displaced position, w = displacement.w (jacobian slot)
    float4* normals;         // [(g+1)^2]  xyz = unit normal, w = 0
    int n;                   // map size
    int grid;                // quads per side (kTileSize of CreateGridVertices)
    float vertex_distance;   // kScale
    float uv_scale;          // ubo.scale
    float choppy;            // ubo.WSChoppy = GetDisplacementLambda()
};

__device__ __forceinline__ float4 sample_linear_repeat(const float4* __restrict__ tex, int n, float u, float v)
{
#pragma clang fp contract(off)
    const float s = u * (float)n - 0.5f, t = v * (float)n - 0.5f;
    const float fs = floorf(s), ft = floorf(t);
    const float a = s - fs, b = t - ft;
    const int x0 = (int)fs & (n - 1), y0 = (int)ft & (n - 1);
    const int x1 = (x0 + 1) & (n - 1), y1 = (y0 + 1) & (n - 1);
    const float4 t00 = tex[(unsigned)(y0 * n + x0)], t10 = tex[(unsigned)(y0 * n + x1)];
    const float4 t01 = tex[(unsigned)(y1 * n + x0)], t11 = tex[(unsigned)(y1 * n + x1)];
    const float ia = 1.0f - a, ib = 1.0f - b;
    auto mix = [&](float c00, float c10, float c01, float c11) {
        return (c00 * ia + c10 * a) * ib + (c01 * ia + c11 * a) * b;
    };
    return make_float4(mix(t00.x, t10.x, t01.x, t11.x), mix(t00.y, t10.y, t01.y, t11.y),
                       mix(t00.z, t10.z, t01.z, t11.z), mix(t00.w, t10.w, t01.w, t11.w));
}

__global__ void k_displace_grid(const GridArgs g)
{
#pragma clang fp contract(off)
    const int side = g.grid + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= side * side) return;
    const int half = g.grid / 2;
    const int xi = i % side - half, yi = i / side - half;          // WaterSurfaceMesh.cpp:514-518
    const float px = (float)xi * g.vertex_distance, pz = (float)yi * g.vertex_distance;
    const float u = (float)(xi + half) / (float)g.grid, v = (float)(yi + half) / (float)g.grid;
    const float amp = fmaxf(fabsf(key_float(g.minmax[0])), fabsf(key_float(g.minmax[1])));
    const float us = u * g.uv_scale, vs = v * g.uv_scale;          // .vert:26
    float4 d = sample_linear_repeat(g.disp, g.n, us, vs);
    d.y = d.y * amp;                                               // .vert:27
    g.positions[i] = make_float4(px + d.x, 0.0f + d.y, pz + d.z, d.w);   // .vert:28-29
    const float4 sl = sample_linear_repeat(g.nrm, g.n, us, vs);    // .vert:33
    const float nx = -(sl.x / (1.0f + g.choppy * sl.z));           // .vert:34-38
    const float nz = -(sl.y / (1.0f + g.choppy * sl.w));
    const float len = sqrtf(nx * nx + 1.0f + nz * nz);
    g.normals[i] = make_float4(nx / len, 1.0f / len, nz / len, 0.0f);
}

// Cascades (the reference's own to-do "Endless - solving the tiling artifacts", README.md:37-44): the
// usual cure for the visible repetition of one FFT tile is to add several tiles of different lengths and seeds, each
// sampled at its own rate.  The tiles of a batch already are independent oceans with their own tile length, so the
// consumer only has to sum them: vertex = grid point + sum_c D_c(uv * s_c) (each height times its own amplitude A_c),
// normal from the summed slopes and summed displacement derivatives with the reference's formula (.vert:34-38).
// w carries the smallest Jacobian slot of the cascades (all 1 unless OCEAN_MODE_JACOBIAN).
constexpr int OCEAN_MAX_CASCADES = 8;
struct CascadeArgs {
    GridArgs g;                        // disp / nrm / minmax of the FIRST tile of the cascade; n, grid, vertex_distance, choppy
    int count;
    size_t tile_texels;                // N * N
    float uv_scale[OCEAN_MAX_CASCADES];
};

__global__ void k_displace_grid_cascades(const CascadeArgs a)
{
#pragma clang fp contract(off)
    const GridArgs& g = a.g;
    const int side = g.grid + 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= side * side) return;
    const int half = g.grid / 2;
    const int xi = i % side - half, yi = i / side - half;
    const float px = (float)xi * g.vertex_distance, pz = (float)yi * g.vertex_distance;
    const float u = (float)(xi + half) / (float)g.grid, v = (float)(yi + half) / (float)g.grid;
    float dx = 0.0f, dy = 0.0f, dz = 0.0f, w = 3.402823466e+38f;
    float sx = 0.0f, sz = 0.0f, ddx = 0.0f, ddz = 0.0f;
    for (int c = 0; c < a.count; ++c) {
        const float us = u * a.uv_scale[c], vs = v * a.uv_scale[c];
        const float amp = fmaxf(fabsf(key_float(g.minmax[2 * c + 0])), fabsf(key_float(g.minmax[2 * c + 1])));
        const float4 d = sample_linear_repeat(g.disp + (size_t)c * a.tile_texels, g.n, us, vs);
        const float4 sl = sample_linear_repeat(g.nrm + (size_t)c * a.tile_texels, g.n, us, vs);
        dx = dx + d.x; dy = dy + d.y * amp; dz = dz + d.z;
        w = fminf(w, d.w);
        sx = sx + sl.x; sz = sz + sl.y; ddx = ddx + sl.z; ddz = ddz + sl.w;
    }
    g.positions[i] = make_float4(px + dx, 0.0f + dy, pz + dz, w);
    const float nx = -(sx / (1.0f + g.choppy * ddx));
    const float nz = -(sz / (1.0f + g.choppy * ddz));
    const float len = sqrtf(nx * nx + 1.0f + nz * nz);
    g.normals[i] = make_float4(nx / len, 1.0f / len, nz / len, 0.0f);
}

// Mip chain of the maps (the reference's LOD hook: s_kUseMipMapping, WaterSurfaceMesh.h:216; Texture2D::GenerateMipmaps,
// vulkan/Texture2D.cpp:228-330 -- level i = vkCmdBlitImage(VK_FILTER_LINEAR) of level i-1 into half the extent).  An exact 2:1
// linear blit samples the point shared by four source texels: the bilinear formula of sample_linear_repeat with both weights
// 1/2, evaluated in the same order (oracle/consumer.py::mip_chain repeats it).  One launch per level, both maps per launch
// (blockIdx.y); a level is N^2/4^l texels, so everything after the first two is launch latency.
struct MipArgs {
    const float4* src[2];    // level l-1 of the displacement map, of the normal map
    float4* dst[2];          // level l
    int w;                   // extent of level l (source extent 2w)
};
__global__ void k_mip_level(const MipArgs m)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m.w * m.w) return;
    const int x = i % m.w, y = i / m.w, sw = 2 * m.w;
    const float4* __restrict__ s = m.src[blockIdx.y];
    const float4 t00 = s[(unsigned)((2 * y) * sw + 2 * x)], t10 = s[(unsigned)((2 * y) * sw + 2 * x + 1)];
    const float4 t01 = s[(unsigned)((2 * y + 1) * sw + 2 * x)], t11 = s[(unsigned)((2 * y + 1) * sw + 2 * x + 1)];
    auto mix = [](float c00, float c10, float c01, float c11) { return (c00 * 0.5f + c10 * 0.5f) * 0.5f + (c01 * 0.5f + c11 * 0.5f) * 0.5f; };
    m.dst[blockIdx.y][i] = make_float4(mix(t00.x, t10.x, t01.x, t11.x), mix(t00.y, t10.y, t01.y, t11.y),
                                       mix(t00.z, t10.z, t01.z, t11.z), mix(t00.w, t10.w, t01.w, t11.w));
}


}  // namespace ocean
