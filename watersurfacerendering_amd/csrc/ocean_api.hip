// ocean_api.hip -- host side of libocean_hip.so: the C ABI of include/ocean.h
// over the gfx950 kernels of ocean_kernels.h.  C++17, HIP runtime only (no
// hipFFT/rocFFT, no torch types).  There is no CPU fallback anywhere in this
// file: without a usable device every entry point returns an error.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <vector>

#include "../../include/ocean.h"
#include "ocean_kernels.h"

using namespace ocean;

static thread_local int g_last_hip = 0;

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) {                         \
            g_last_hip = (int)e_;                       \
            return OCEAN_E_HIP;                         \
        }                                               \
    } while (0)

struct ocean_ctx {
    uint32_t n = 0;
    uint32_t tiles = 0;
    int device = 0;
    bool prepared = false;
    bool own_stream = true;
    hipStream_t stream = nullptr;
    hipStream_t own = nullptr;
    std::vector<ocean_params> params;
    uint64_t seed = 0;
    // device state
    float2* h0 = nullptr;
    float* omega = nullptr;
    float* k1d = nullptr;
    float2* tw = nullptr;
    float2* z = nullptr;
    float2* zh = nullptr;
    float* hraw = nullptr;
    unsigned* minmax = nullptr;
    float4* disp = nullptr;
    float4* nrm = nullptr;
    float4* ext_disp = nullptr;
    float4* ext_nrm = nullptr;
    float* toff = nullptr;
    bool use_toff = false;
    float* lambda = nullptr;
    TileParams* tparams = nullptr;
    float2* xi = nullptr;          // injected or generated draws (kept for read-back)
    unsigned* h_minmax = nullptr;  // pinned
    unsigned long long* stamps = nullptr;   // diagnostic builds only
    hipEvent_t ev[8] = {};
};

static void free_device(ocean_ctx* c)
{
    void* bufs[] = {c->h0, c->omega, c->k1d, c->tw, c->z, c->zh, c->hraw, c->minmax, c->disp, c->nrm,
                    c->toff, c->lambda, c->tparams, c->xi};
    for (void* b : bufs) if (b) (void)hipFree(b);
    c->h0 = nullptr; c->omega = nullptr; c->k1d = nullptr; c->tw = nullptr; c->z = nullptr; c->zh = nullptr;
    c->hraw = nullptr; c->minmax = nullptr; c->disp = nullptr; c->nrm = nullptr; c->toff = nullptr;
    c->lambda = nullptr; c->tparams = nullptr; c->xi = nullptr;
    c->prepared = false;
}

static bool size_ok(uint32_t n) { return n >= 16 && n <= 4096 && (n & (n - 1)) == 0; }

static int alloc_device(ocean_ctx* c)
{
    const size_t n = c->n, n2 = n * n, t = c->tiles;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(&c->h0, t * n2 * sizeof(float2)));
    HIP_TRY(hipMalloc(&c->omega, t * n2 * sizeof(float)));
    HIP_TRY(hipMalloc(&c->k1d, t * n * sizeof(float)));
    HIP_TRY(hipMalloc(&c->tw, n * sizeof(float2)));
    // half-spectrum intermediates (ocean_kernels.h, struct Half): padded columns must read as zero
    const size_t nu = n / 2 + 1, nup = n / 2 + 8;
    HIP_TRY(hipMalloc(&c->z, t * 3 * nu * 2 * nup * sizeof(float2)));
    HIP_TRY(hipMalloc(&c->zh, t * nu * nup * sizeof(float2)));
    HIP_TRY(hipMalloc(&c->hraw, t * nup * n * sizeof(float)));
    HIP_TRY(hipMemset(c->z, 0, t * 3 * nu * 2 * nup * sizeof(float2)));
    HIP_TRY(hipMemset(c->zh, 0, t * nu * nup * sizeof(float2)));
    HIP_TRY(hipMemset(c->hraw, 0, t * nup * n * sizeof(float)));
    HIP_TRY(hipMalloc(&c->minmax, t * 2 * sizeof(unsigned)));
    HIP_TRY(hipMalloc(&c->disp, t * n2 * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->nrm, t * n2 * sizeof(float4)));
    HIP_TRY(hipMalloc(&c->toff, t * sizeof(float)));
    HIP_TRY(hipMalloc(&c->lambda, t * sizeof(float)));
    HIP_TRY(hipMalloc(&c->tparams, t * sizeof(TileParams)));
    HIP_TRY(hipMemset(c->toff, 0, t * sizeof(float)));
    // twiddle table exp(+2 pi i k / N), rounded once from double
    std::vector<float2> tw(n);
    for (size_t k = 0; k < n; ++k) {
        const double a = 2.0 * M_PI * (double)k / (double)n;
        tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    HIP_TRY(hipMemcpy(c->tw, tw.data(), n * sizeof(float2), hipMemcpyHostToDevice));
    return OCEAN_OK;
}

extern "C" {

void ocean_default_params(ocean_params* p)
{
    if (!p) return;
    p->tile_length = 1000.0f;
    p->wind_dir_x = 1.0f; p->wind_dir_y = 1.0f;
    p->wind_speed = 30.0f;
    p->anim_period = 200.0f;
    p->phillips_const = 3e-7f;
    p->damping = 0.1f;
    p->lambda = -1.0f;
}

const char* ocean_strerror(int code)
{
    switch (code) {
        case OCEAN_OK: return "ok";
        case OCEAN_E_INVALID: return "invalid argument";
        case OCEAN_E_NO_DEVICE: return "no usable HIP device (gfx950 required, no CPU fallback)";
        case OCEAN_E_HIP: return "HIP runtime error";
        case OCEAN_E_NOT_READY: return "ocean_prepare has not been called";
        case OCEAN_E_NOMEM: return "out of memory";
        case OCEAN_E_UNSUPPORTED: return "tile size must be a power of two in [16, 4096]";
        default: return "unknown error";
    }
}

int ocean_abi_version(void) { return OCEAN_ABI_VERSION; }
int ocean_last_hip_error(void) { return g_last_hip; }

int ocean_create(ocean_t** out, uint32_t tile_size, uint32_t tiles, int device)
{
    if (!out || tiles == 0) return OCEAN_E_INVALID;
    *out = nullptr;
    if (tile_size == 0 || (tile_size & (tile_size - 1))) return OCEAN_E_INVALID;
    if (!size_ok(tile_size)) return OCEAN_E_UNSUPPORTED;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return OCEAN_E_NO_DEVICE;
    if (device < 0 || device >= count) return OCEAN_E_NO_DEVICE;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return OCEAN_E_NO_DEVICE;
    ocean_ctx* c = new (std::nothrow) ocean_ctx();
    if (!c) return OCEAN_E_NOMEM;
    c->n = tile_size; c->tiles = tiles; c->device = device;
    c->params.resize(tiles);
    for (auto& p : c->params) ocean_default_params(&p);
    int rc = OCEAN_OK;
    do {
        if (hipSetDevice(device) != hipSuccess) { rc = OCEAN_E_HIP; break; }
        if (hipStreamCreateWithFlags(&c->own, hipStreamNonBlocking) != hipSuccess) { rc = OCEAN_E_HIP; break; }
        c->stream = c->own;
        if (hipHostMalloc((void**)&c->h_minmax, tiles * 2 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) { rc = OCEAN_E_HIP; break; }
        for (auto& e : c->ev) if (hipEventCreate(&e) != hipSuccess) { rc = OCEAN_E_HIP; break; }
        if (rc) break;
        rc = alloc_device(c);
    } while (0);
    if (rc != OCEAN_OK) { ocean_destroy(c); return rc; }
    *out = c;
    return OCEAN_OK;
}

void ocean_destroy(ocean_t* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_device(c);
    if (c->h_minmax) (void)hipHostFree(c->h_minmax);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    if (c->own) (void)hipStreamDestroy(c->own);
    delete c;
}

int ocean_set_lambda(ocean_t* c, uint32_t tile, float lambda);

int ocean_set_params(ocean_t* c, uint32_t tile, const ocean_params* p)
{
    if (!c || !p) return OCEAN_E_INVALID;
    if (tile != OCEAN_ALL_TILES && tile >= c->tiles) return OCEAN_E_INVALID;
    if (!(p->tile_length > 0.0f) || (p->wind_dir_x == 0.0f && p->wind_dir_y == 0.0f)) return OCEAN_E_INVALID;
    for (uint32_t i = 0; i < c->tiles; ++i)
        if (tile == OCEAN_ALL_TILES || tile == i) {
            c->params[i] = *p;
        }
    // like the reference, the new values are picked up by the next ocean_prepare
    // (WaterSurfaceMesh.cpp:888-900); frames keep using the prepared state until then
    return ocean_set_lambda(c, tile, p->lambda);
}

int ocean_get_params(const ocean_t* c, uint32_t tile, ocean_params* p)
{
    if (!c || !p || tile >= c->tiles) return OCEAN_E_INVALID;
    *p = c->params[tile];
    return OCEAN_OK;
}

int ocean_set_lambda(ocean_t* c, uint32_t tile, float lambda)
{
    if (!c) return OCEAN_E_INVALID;
    if (tile != OCEAN_ALL_TILES && tile >= c->tiles) return OCEAN_E_INVALID;
    std::vector<float> l(c->tiles);
    for (uint32_t i = 0; i < c->tiles; ++i) {
        if (tile == OCEAN_ALL_TILES || tile == i) c->params[i].lambda = lambda;
        l[i] = c->params[i].lambda;
    }
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(c->lambda, l.data(), c->tiles * sizeof(float), hipMemcpyHostToDevice));
    return OCEAN_OK;
}

int ocean_set_tile_size(ocean_t* c, uint32_t tile_size)
{
    if (!c) return OCEAN_E_INVALID;
    if (tile_size == 0 || (tile_size & (tile_size - 1))) return OCEAN_E_INVALID;   // .cpp:461-467
    if (!size_ok(tile_size)) return OCEAN_E_UNSUPPORTED;
    if (tile_size == c->n) return OCEAN_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    free_device(c);
    c->n = tile_size;
    c->ext_disp = nullptr; c->ext_nrm = nullptr;
    int rc = alloc_device(c);
    if (rc) return rc;
    return ocean_set_lambda(c, OCEAN_ALL_TILES, c->params[0].lambda) == OCEAN_OK ? OCEAN_OK : OCEAN_E_HIP;
}

uint32_t ocean_tile_size(const ocean_t* c) { return c ? c->n : 0; }
uint32_t ocean_tiles(const ocean_t* c) { return c ? c->tiles : 0; }

int ocean_prepare(ocean_t* c, uint64_t seed, const float* xi_or_null)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = c->n, n2 = n * n, t = c->tiles;
    std::vector<TileParams> tp(t);
    std::vector<float> lam(t);
    for (size_t i = 0; i < t; ++i) {
        const ocean_params& p = c->params[i];
        // SetWindDirection: w * (1/sqrt(dot(w,w)))  (.cpp:476-479)
        const float d = p.wind_dir_x * p.wind_dir_x + p.wind_dir_y * p.wind_dir_y;
        const float inv = 1.0f / std::sqrt(d);
        tp[i].wind_x = p.wind_dir_x * inv;
        tp[i].wind_y = p.wind_dir_y * inv;
        tp[i].wind_speed = p.wind_speed > 0.0001f ? p.wind_speed : 0.0001f;       // .cpp:481-484
        tp[i].phillips_a = p.phillips_const;
        tp[i].damping = p.damping;
        tp[i].base_freq = (float)((double)2.0f * M_PI / (double)p.anim_period);   // .cpp:486-490
        tp[i].length = p.tile_length;
        tp[i].pad_ = 0.0f;
        tp[i].seed = seed + i;
        lam[i] = p.lambda;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(c->tparams, tp.data(), t * sizeof(TileParams), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->lambda, lam.data(), t * sizeof(float), hipMemcpyHostToDevice));
    if (!c->xi) HIP_TRY(hipMalloc(&c->xi, t * n2 * sizeof(float2)));
    if (xi_or_null) HIP_TRY(hipMemcpy(c->xi, xi_or_null, t * n2 * sizeof(float2), hipMemcpyHostToDevice));
    {
        dim3 g((unsigned)((n + 255) / 256), (unsigned)t);
        hipLaunchKernelGGL(k_init_k1d, g, dim3(256), 0, c->stream, c->k1d, c->tparams, (int)n);
        dim3 g2((unsigned)((n2 + 255) / 256), (unsigned)t);
        hipLaunchKernelGGL(k_init_spectrum, g2, dim3(256), 0, c->stream, c->h0, c->omega,
                           xi_or_null ? (float2*)nullptr : c->xi, xi_or_null ? c->xi : (const float2*)nullptr,
                           c->k1d, c->tparams, (int)n);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->seed = seed;
    c->prepared = true;
    return OCEAN_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------
// frame launch
// ---------------------------------------------------------------------------------
template <class K>
static hipError_t allow_lds(K kernel, size_t bytes)
{
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

template <int N>
static hipError_t launch_frame(ocean_ctx* c, const FrameArgs& a, int which /* bit0 rows, bit1 cols_b, bit2 disp */,
                               hipEvent_t* marks /* 4 events or null */)
{
    using G = Geo<N>;
    using HF = Half<N>;
    const unsigned tiles = c->tiles;
    hipError_t e;
    constexpr int C = G::CC;
    constexpr size_t lds_rows = rows_lds_bytes<N>();
    constexpr size_t lds_b = sizeof(c32) * fft_lds_elems<N, C>() + sizeof(float) * 2 * ((G::T_C + 63) / 64);
    constexpr size_t lds_d = sizeof(c32) * fft_lds_elems<N, C>();
    constexpr unsigned hb = HF::NUP / (2 * C), nb = (HF::NU + C - 1) / C;
    static bool attr_done = false;
    if (!attr_done) {
        if ((e = allow_lds(k_rows<N, G::T_ROWS, typename G::PR>, lds_rows)) != hipSuccess) return e;
        if ((e = allow_lds(k_cols_b<N, C, G::T_C, typename G::PC>, lds_b)) != hipSuccess) return e;
        if ((e = allow_lds(k_cols_disp<N, C, G::T_C, typename G::PC>, lds_d)) != hipSuccess) return e;
        attr_done = true;
    }
    if (marks) (void)hipEventRecord(marks[0], c->stream);
    if (which & 1) {
        unsigned gx = N / 2 + 1;
#ifdef OCEAN_STAMPS
        if (const char* ev = getenv("OCEAN_DEBUG_ROWS_GRID")) gx = (unsigned)atoi(ev);   // diagnostic: partial grid
#endif
        hipLaunchKernelGGL((k_rows<N, G::T_ROWS, typename G::PR>), dim3(gx, tiles), dim3(G::T_ROWS), lds_rows,
                           c->stream, a);
    }
    if (marks) (void)hipEventRecord(marks[1], c->stream);
    if (which & 2)
        hipLaunchKernelGGL((k_cols_b<N, C, G::T_C, typename G::PC>), dim3(hb + nb, tiles), dim3(G::T_C), lds_b,
                           c->stream, a);
    if (marks) (void)hipEventRecord(marks[2], c->stream);
    if (which & 4)
        hipLaunchKernelGGL((k_cols_disp<N, C, G::T_C, typename G::PC>), dim3(nb, tiles), dim3(G::T_C), lds_d,
                           c->stream, a);
    if (marks) (void)hipEventRecord(marks[3], c->stream);
    return hipGetLastError();
}

static int enqueue_frame(ocean_ctx* c, float t, int which, hipEvent_t* marks)
{
    if (!c) return OCEAN_E_INVALID;
    if (!c->prepared) return OCEAN_E_NOT_READY;
    FrameArgs a;
    a.h0 = c->h0; a.omega = c->omega; a.k1d = c->k1d; a.tw = c->tw;
    a.z = c->z; a.zh = c->zh; a.hraw = c->hraw; a.minmax = c->minmax;
    a.disp = c->ext_disp ? c->ext_disp : c->disp;
    a.nrm = c->ext_nrm ? c->ext_nrm : c->nrm;
    a.toff = c->use_toff ? c->toff : nullptr;
    a.lambda = c->lambda;
    a.t = t;
    a.stamps = c->stamps;
    hipError_t e = hipErrorInvalidValue;
    switch (c->n) {
        case 16: e = launch_frame<16>(c, a, which, marks); break;
        case 32: e = launch_frame<32>(c, a, which, marks); break;
        case 64: e = launch_frame<64>(c, a, which, marks); break;
        case 128: e = launch_frame<128>(c, a, which, marks); break;
        case 256: e = launch_frame<256>(c, a, which, marks); break;
        case 512: e = launch_frame<512>(c, a, which, marks); break;
        case 1024: e = launch_frame<1024>(c, a, which, marks); break;
        case 2048: e = launch_frame<2048>(c, a, which, marks); break;
        case 4096: e = launch_frame<4096>(c, a, which, marks); break;
        default: return OCEAN_E_UNSUPPORTED;
    }
    if (e != hipSuccess) { g_last_hip = (int)e; return OCEAN_E_HIP; }
    return OCEAN_OK;
}

extern "C" {

int ocean_compute_waves_async(ocean_t* c, float t)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    return enqueue_frame(c, t, 7, nullptr);
}

int ocean_synchronize(ocean_t* c)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OCEAN_OK;
}

static int fetch_minmax(ocean_ctx* c)
{
    HIP_TRY(hipMemcpyAsync(c->h_minmax, c->minmax, c->tiles * 2 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OCEAN_OK;
}

static float amp_of(const ocean_ctx* c, uint32_t tile, float* mn_out, float* mx_out)
{
    const float mn = key_float(c->h_minmax[2 * tile]), mx = key_float(c->h_minmax[2 * tile + 1]);
    if (mn_out) *mn_out = mn;
    if (mx_out) *mx_out = mx;
    return std::fmax(std::fabs(mn), std::fabs(mx));   // .cpp:448
}

int ocean_compute_waves(ocean_t* c, float t, float* out_amp)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    int rc = enqueue_frame(c, t, 7, nullptr);
    if (rc) return rc;
    rc = fetch_minmax(c);
    if (rc) return rc;
    if (out_amp)
        for (uint32_t i = 0; i < c->tiles; ++i) out_amp[i] = amp_of(c, i, nullptr, nullptr);
    return OCEAN_OK;
}

int ocean_set_time_offsets(ocean_t* c, const float* offsets)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (!offsets) { c->use_toff = false; return OCEAN_OK; }
    HIP_TRY(hipMemcpy(c->toff, offsets, c->tiles * sizeof(float), hipMemcpyHostToDevice));
    c->use_toff = true;
    return OCEAN_OK;
}

int ocean_get_heights(ocean_t* c, uint32_t tile, float* amp, float* min_h, float* max_h)
{
    if (!c || tile >= c->tiles) return OCEAN_E_INVALID;
    if (!c->prepared) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    int rc = fetch_minmax(c);
    if (rc) return rc;
    const float a = amp_of(c, tile, min_h, max_h);
    if (amp) *amp = a;
    return OCEAN_OK;
}

int ocean_read_maps(ocean_t* c, uint32_t first, uint32_t count, float* disp, float* nrm)
{
    if (!c || first >= c->tiles || count == 0 || first + count > c->tiles) return OCEAN_E_INVALID;
    if (!c->prepared) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    const float4* d = (c->ext_disp ? c->ext_disp : c->disp) + first * n2;
    const float4* q = (c->ext_nrm ? c->ext_nrm : c->nrm) + first * n2;
    if (disp) HIP_TRY(hipMemcpyAsync(disp, d, count * n2 * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    if (nrm) HIP_TRY(hipMemcpyAsync(nrm, q, count * n2 * sizeof(float4), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OCEAN_OK;
}

int ocean_device_maps(ocean_t* c, void** d_disp, void** d_nrm)
{
    if (!c) return OCEAN_E_INVALID;
    if (d_disp) *d_disp = c->ext_disp ? (void*)c->ext_disp : (void*)c->disp;
    if (d_nrm) *d_nrm = c->ext_nrm ? (void*)c->ext_nrm : (void*)c->nrm;
    return OCEAN_OK;
}

int ocean_bind_output(ocean_t* c, void* d_disp, void* d_nrm)
{
    if (!c) return OCEAN_E_INVALID;
    if (((uintptr_t)d_disp | (uintptr_t)d_nrm) & 15u) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->ext_disp = (float4*)d_disp;
    c->ext_nrm = (float4*)d_nrm;
    return OCEAN_OK;
}

void* ocean_stream(ocean_t* c) { return c ? (void*)c->stream : nullptr; }

int ocean_set_stream(ocean_t* c, void* s)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own;
    return OCEAN_OK;
}

int ocean_read_spectrum(ocean_t* c, uint32_t tile, float* h0, float* omega)
{
    if (!c || tile >= c->tiles) return OCEAN_E_INVALID;
    if (!c->prepared) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h0) HIP_TRY(hipMemcpy(h0, c->h0 + tile * n2, n2 * sizeof(float2), hipMemcpyDeviceToHost));
    if (omega) HIP_TRY(hipMemcpy(omega, c->omega + tile * n2, n2 * sizeof(float), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}

int ocean_read_xi(ocean_t* c, uint32_t tile, float* xi)
{
    if (!c || tile >= c->tiles || !xi) return OCEAN_E_INVALID;
    if (!c->prepared || !c->xi) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(xi, c->xi + tile * n2, n2 * sizeof(float2), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}

int ocean_time_frames(ocean_t* c, float t0, float dt, int warmup, int frames, float* ms_total, float* ms_kernel)
{
    if (!c || frames <= 0 || warmup < 0) return OCEAN_E_INVALID;
    if (!c->prepared) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    for (int j = 0; j < warmup; ++j)
        if ((rc = enqueue_frame(c, t0 + dt * (float)j, 7, nullptr))) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipEventRecord(c->ev[4], c->stream));
    for (int j = 0; j < frames; ++j)
        if ((rc = enqueue_frame(c, t0 + dt * (float)(warmup + j), 7, nullptr))) return rc;
    HIP_TRY(hipEventRecord(c->ev[5], c->stream));
    HIP_TRY(hipEventSynchronize(c->ev[5]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, c->ev[4], c->ev[5]));
    if (ms_total) *ms_total = ms;
    if (ms_kernel) {
        // second pass: events around every launch (adds event overhead, so it is
        // reported separately from ms_total and never mixed into it)
        double acc[3] = {0, 0, 0};
        for (int j = 0; j < frames; ++j) {
            if ((rc = enqueue_frame(c, t0 + dt * (float)(warmup + j), 7, c->ev))) return rc;
            HIP_TRY(hipEventSynchronize(c->ev[3]));
            for (int k = 0; k < 3; ++k) {
                float m = 0.f;
                HIP_TRY(hipEventElapsedTime(&m, c->ev[k], c->ev[k + 1]));
                acc[k] += m;
            }
        }
        for (int k = 0; k < 3; ++k) ms_kernel[k] = (float)(acc[k] / frames);
    }
    return OCEAN_OK;
}

#ifdef OCEAN_STAMPS
// diagnostic build only: per-workgroup clock stamps of the last frame
int ocean_debug_stamps(ocean_t* c, int enable, unsigned long long* host_out, size_t count)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (enable && !c->stamps) {
        HIP_TRY(hipMalloc(&c->stamps, (size_t)1 << 24));
        HIP_TRY(hipMemset(c->stamps, 0, (size_t)1 << 24));
    }
    if (host_out) HIP_TRY(hipMemcpy(host_out, c->stamps, count * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}
#endif

int ocean_algorithmic_bytes_per_texel(const ocean_t* c)
{
    (void)c;
    return 108;   // SURVEY.md 8d accounting for the 7-field two-pass scheme; this pipeline moves 76 (ocean_kernels.h)
}

}  // extern "C"
