// ocean_api.hip -- host side of libocean_hip.so: the C ABI of include/ocean.h
// over the gfx950 kernels of ocean_kernels.h.  C++17, HIP runtime only (no
// hipFFT/rocFFT, no torch types).  There is no CPU fallback anywhere in this
// file: without a usable device every entry point returns an error.
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <mutex>
#include <vector>

#include "ocean_ctx.h"
#include "ocean_aux_kernels.h"      // this translation unit also holds the Prepare(), read-out and consumer kernels

using namespace ocean;

static thread_local int g_last_hip = 0;

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t e_ = (expr);                         \
        if (e_ != hipSuccess) {                         \
            g_last_hip = (int)e_;                       \
            (void)hipGetLastError();                    \
            return e_ == hipErrorOutOfMemory ? OCEAN_E_NOMEM : OCEAN_E_HIP; \
        }                                               \
    } while (0)

// Ranges page-locked through THIS library, with their device addresses (looked up once, at registration): ocean_compute_waves_read's copy
// kernels store through them without asking the runtime on every call (hipHostGetDevicePointer costs a few microseconds of a 190 us call).
// Memory page-locked by other means is still found, per call, through the runtime.  Process-wide like the registrations themselves; guarded.
namespace {
struct PinnedRange { char* host; size_t bytes; char* dev; };
std::mutex g_pinned_mutex;
std::vector<PinnedRange> g_pinned;
void* pinned_device_address(const void* host_ptr, size_t bytes)
{
    const char* h = static_cast<const char*>(host_ptr);
    {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        for (const PinnedRange& r : g_pinned)
            if (h >= r.host && h + bytes <= r.host + r.bytes) return r.dev + (h - r.host);
    }
    // page-locked by other means: the runtime knows -- and the WHOLE range must be (its last byte one linear mapping away from its first)
    void *dp = nullptr, *dp_last = nullptr;
    if (bytes == 0 || hipHostGetDevicePointer(&dp, const_cast<void*>(host_ptr), 0) != hipSuccess ||
        hipHostGetDevicePointer(&dp_last, const_cast<char*>(h) + (bytes - 1), 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (!dp || static_cast<char*>(dp_last) != static_cast<char*>(dp) + (bytes - 1)) return nullptr;
    return dp;
}
}  // namespace

static void free_set(ocean_ctx* c, int i);
static void comm_release(ocean_ctx* c);
static void release_import(ocean_ctx* c);

static void free_device(ocean_ctx* c)
{
    void* bufs[] = {c->h0, c->omega, c->omega_q, c->base_freq, c->omega_q_overflow, c->k1d, c->tw, c->toff, c->lambda, c->tparams, c->xi,
                    c->h0h, c->h0_inv_scale, c->h0_maxbits, c->zscale, c->zbounds};
    for (void* b : bufs) if (b) (void)hipFree(b);
    for (int i = 0; i < MAXD; ++i) free_set(c, i);
    c->h0 = nullptr; c->omega = nullptr; c->omega_q = nullptr; c->base_freq = nullptr; c->omega_q_overflow = nullptr;
    c->k1d = nullptr; c->tw = nullptr;
    c->toff = nullptr; c->lambda = nullptr; c->tparams = nullptr; c->xi = nullptr;
    c->h0h = nullptr; c->h0_inv_scale = nullptr; c->h0_maxbits = nullptr; c->zscale = nullptr; c->zbounds = nullptr;
    c->prepared = false; c->placement_done = false;
    // nothing of the old buffers may be referred to any more: no frame, no chain to read out, no mips of the old size
    c->have_frame = false; c->last_set = 0; c->frame_ctr = 0; c->mips_ready = false; c->grid_vertices = 0;
    c->maps_shared = false;             // the exported / handed-out maps are gone with the buffers
}

static void free_set(ocean_ctx* c, int i)
{
    void* per[] = {c->z[i], c->zh[i], c->hraw[i], c->z3[i], c->jraw[i], c->jac0[i], c->minmax[i], c->hdone[i], c->zdone[i], c->done_ctr[i], c->dispN[i]};   // (nrmN: same allocation)
    for (void* b : per) if (b) (void)hipFree(b);
    if (c->done_rec[i]) (void)hipHostFree(c->done_rec[i]);
    for (auto& p : c->pack_half[i]) if (p) { (void)hipFree(p); p = nullptr; }
    c->z[i] = nullptr; c->zh[i] = nullptr; c->hraw[i] = nullptr; c->z3[i] = nullptr; c->jraw[i] = nullptr; c->jac0[i] = nullptr;
    c->minmax[i] = nullptr; c->hdone[i] = nullptr; c->zdone[i] = nullptr; c->zgen[i] = 0; c->done_ctr[i] = nullptr; c->done_rec[i] = nullptr; c->seq[i] = 0; c->frame_valid[i] = false;
    c->dispN[i] = nullptr; c->nrmN[i] = nullptr;
}

static int alloc_set_buffers(ocean_ctx* c, int i)
{
    const size_t n = c->n, n2 = n * n, t = c->tiles;
    const size_t nu = n / 2 + 1, nup = (n / 2 + 16) & ~(size_t)15;
    // half-spectrum intermediates (ocean_kernels.h, struct Half): padded columns must read as zero
    HIP_TRY(hipMalloc(&c->z[i], t * 3 * nu * 2 * nup * sizeof(float2)));
    HIP_TRY(hipMalloc(&c->zh[i], t * nu * nup * sizeof(float2)));
    HIP_TRY(hipMalloc(&c->hraw[i], t * nup * n * sizeof(float)));
    // zero-fill ON THE CHAIN'S OWN STREAM: the chain streams are non-blocking (no implicit ordering with the
    // null stream), and the first z pass of the chain is enqueued right behind this
    HIP_TRY(hipMemsetAsync(c->z[i], 0, t * 3 * nu * 2 * nup * sizeof(float2), stream_of(c, i)));
    HIP_TRY(hipMemsetAsync(c->zh[i], 0, t * nu * nup * sizeof(float2), stream_of(c, i)));
    HIP_TRY(hipMemsetAsync(c->hraw[i], 0, t * nup * n * sizeof(float), stream_of(c, i)));
    HIP_TRY(hipMalloc(&c->minmax[i], t * 2 * sizeof(unsigned)));
    HIP_TRY(hipMalloc(&c->hdone[i], t * sizeof(unsigned)));
    HIP_TRY(hipMalloc(&c->zdone[i], t * sizeof(unsigned)));
    HIP_TRY(hipMemsetAsync(c->zdone[i], 0, t * sizeof(unsigned), stream_of(c, i)));
    c->zgen[i] = 0;
    HIP_TRY(hipMalloc(&c->done_ctr[i], (1 + DONE_GROUPS) * DONE_STRIDE * sizeof(unsigned)));
    HIP_TRY(hipMemsetAsync(c->done_ctr[i], 0, (1 + DONE_GROUPS) * DONE_STRIDE * sizeof(unsigned), stream_of(c, i)));
    // the last workgroup of a frame drops (min key, max key, sequence number) per tile into this host-coherent
    // buffer: the synchronous ComputeWaves polls it -- no stream synchronisation, no device-to-host copy
    HIP_TRY(hipHostMalloc((void**)&c->done_rec[i], t * sizeof(uint4), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->done_rec[i], 0, t * sizeof(uint4));
    c->seq[i] = 0; c->frame_valid[i] = false;       // (a fresh, zeroed record buffer: numbering may start over)
    // both maps of the set in ONE allocation [displacement | normal] -- the range ocean_export_maps hands out as one dma-buf,
    // in the order the reference lays its staging buffer out (WaterSurfaceMesh.cpp:736-738); a whole number of 2 MiB pages
    c->maps_bytes[i] = (2 * t * n2 * sizeof(float4) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    HIP_TRY(hipMalloc(&c->dispN[i], c->maps_bytes[i]));
    c->nrmN[i] = c->dispN[i] + t * n2;
    return OCEAN_OK;
}

// The three extra intermediates of OCEAN_MODE_JACOBIAN (+ 50 % of a chain's intermediates): allocated by the first frame
// of that mode on the chain, so that contexts which never use the mode never pay for it.
static int alloc_jacobian(ocean_ctx* c, int i)
{
    if (c->jac0[i]) return OCEAN_OK;               // the last of the three: complete
    const size_t n = c->n, t = c->tiles;
    const size_t nu = n / 2 + 1, nup = (n / 2 + 16) & ~(size_t)15;
    void** bufs[] = {(void**)&c->z3[i], (void**)&c->jraw[i], (void**)&c->jac0[i]};
    const size_t bytes[] = {t * nu * 2 * nup * sizeof(float2), t * nup * n * sizeof(float), t * nup * n * sizeof(float)};
    for (int k = 0; k < 3; ++k) {
        if (*bufs[k]) { (void)hipFree(*bufs[k]); *bufs[k] = nullptr; }     // leftovers of an earlier failed attempt
        if (hipMalloc(bufs[k], bytes[k]) != hipSuccess || hipMemsetAsync(*bufs[k], 0, bytes[k], stream_of(c, i)) != hipSuccess) {
            for (int j = 0; j <= k; ++j) if (*bufs[j]) { (void)hipFree(*bufs[j]); *bufs[j] = nullptr; }
            g_last_hip = (int)hipGetLastError();
            return OCEAN_E_NOMEM;
        }
    }
    return OCEAN_OK;
}

// intermediates + maps of one pipeline chain (allocated on first use; all or nothing)
static int alloc_set(ocean_ctx* c, int i)
{
    if (c->nrmN[i]) return OCEAN_OK;               // the last buffer allocated: the set is complete
    free_set(c, i);                                 // leftovers of an earlier failed attempt
    const int rc = alloc_set_buffers(c, i);
    if (rc != OCEAN_OK) free_set(c, i);
    return rc;
}

static bool size_ok(uint32_t n) { return n >= 16 && n <= 4096 && (n & (n - 1)) == 0; }

static int alloc_device(ocean_ctx* c)
{
    const size_t n = c->n, n2 = n * n, t = c->tiles;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(&c->h0, t * n2 * sizeof(float2)));
    HIP_TRY(hipMalloc(&c->omega, t * n2 * sizeof(float)));
    HIP_TRY(hipMalloc(&c->omega_q, t * n2 * sizeof(uint16_t)));
    HIP_TRY(hipMalloc(&c->base_freq, t * sizeof(float)));
    HIP_TRY(hipMalloc(&c->omega_q_overflow, sizeof(unsigned)));
    HIP_TRY(hipMalloc(&c->k1d, t * n * sizeof(float)));
    HIP_TRY(hipMalloc(&c->tw, n * sizeof(float2)));
    {
        int rc_ = alloc_set(c, 0);
        if (rc_) return rc_;
    }
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

// An in-launch wait of a merged / one-launch frame that gave up (ocean_kernels.h: wait_counter) has produced a wrong frame: never silently -- and
// never fatally either (on a shared or time-sliced device a preempted producer is a slow producer, not a broken one).  The host
// drains the context, clears the word, switches the hand-off forms OFF for this context (launch_frame: c->merged_x) and runs the most recent frame
// of every chain that used one again in the three-launch form -- same time, same chain, same tracking --, so the caller's wait / synchronise /
// read-out returns the frame it asked for.  What cannot be redone is reported once: a stream-ordered consumer (gather, vertex stage, mips)
// enqueued behind such a frame has consumed it (OCEAN_E_HIP / hipErrorLaunchTimeOut from this call; the context is usable, the consumer call is
// the caller's to repeat).
static int enqueue_frame(ocean_ctx* c, float t, bool pipelined, hipEvent_t* marks, bool track, int redo_set);
static int placement_search(ocean_ctx* c);
static bool fault_raised(const ocean_ctx* c) { return c->fault && __atomic_load_n(c->fault, __ATOMIC_ACQUIRE) != 0u; }
static void reset_pipeline_state(ocean_ctx* c)
{
    for (bool& p : c->gather_pending) p = false;
    c->consumer_pending = false;
    c->burst_pos = 0; c->z_last_set = -1;          // the pipeline is empty: the next pipelined frames start staggered (enqueue_frame)
}
static int drain_streams(ocean_ctx* c)
{
    for (int i = 0; i < MAXD; ++i)
        if (c->own[i]) HIP_TRY(hipStreamSynchronize(c->own[i]));
    if (c->user) HIP_TRY(hipStreamSynchronize(c->user));
    if (c->comm_stream) HIP_TRY(hipStreamSynchronize(c->comm_stream));
    if (c->copy_stream) HIP_TRY(hipStreamSynchronize(c->copy_stream));     // (ocean_compute_waves_read's normal-map copies)
    return OCEAN_OK;
}
static int recover_fault(ocean_ctx* c)
{
    if (c->recovering) { g_last_hip = (int)hipErrorLaunchTimeOut; return OCEAN_E_HIP; }
    { int rc_ = drain_streams(c); if (rc_) return rc_; }
    bool consumed = c->consumer_pending;
    for (bool p : c->gather_pending) consumed = consumed || p;
    __atomic_store_n(c->fault, 0u, __ATOMIC_RELEASE);
    c->merged_x = false;                            // this context keeps the three-launch frame from now on (ocean_set_merged_xpass(ctx, 1) allows the forms again)
    c->fault_recoveries++;
    reset_pipeline_state(c);
    c->recovering = true;
    int rc = OCEAN_OK;
    const int last = c->last_set;
    for (int set = 0; set < MAXD && rc == OCEAN_OK; ++set)
        if (c->frame_valid[set] && c->last_handoff[set]) rc = enqueue_frame(c, c->last_t[set], c->last_pipe[set], nullptr, c->tracked[set], set);
    c->last_set = last;                             // the caller's "most recent frame" is still the one it enqueued last
    if (rc == OCEAN_OK) rc = drain_streams(c);
    c->recovering = false;
    reset_pipeline_state(c);
    if (rc) return rc;
    if (fault_raised(c) || consumed) { g_last_hip = (int)hipErrorLaunchTimeOut; return OCEAN_E_HIP; }
    return OCEAN_OK;
}
static int check_fault(ocean_ctx* c) { return fault_raised(c) ? recover_fault(c) : OCEAN_OK; }

static int sync_all(ocean_ctx* c)
{
    { int rc_ = drain_streams(c); if (rc_) return rc_; }
    if (fault_raised(c)) return recover_fault(c);   // (resets the pipeline bookkeeping whatever it returns)
    reset_pipeline_state(c);
    return OCEAN_OK;
}
#define SYNC_ALL(c) do { int rc_ = sync_all(c); if (rc_) return rc_; } while (0)

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
        case OCEAN_E_UNSUPPORTED: return "unsupported (tile size must be a power of two in [16, 4096]; no export of caller-bound maps)";
        case OCEAN_E_COMM: return "RCCL error (or librccl could not be loaded); see ocean_last_rccl_error";
        default: return "unknown error";
    }
}

int ocean_abi_version(void) { return OCEAN_ABI_VERSION; }

// Content hash of the sources (csrc/Makefile computes it and passes -DOCEAN_BUILD_ID); the marker in front lets the loader find the
// string in the file without loading the library (watersurfacerendering_amd/_abi.py: library_build_id).
#ifndef OCEAN_BUILD_ID
#define OCEAN_BUILD_ID "unknown"
#endif
static const char g_build_id[] = "OCEAN_BUILD_ID:" OCEAN_BUILD_ID;
const char* ocean_build_id(void) { return g_build_id + 15; }
int ocean_last_hip_error(void) { return g_last_hip; }
unsigned ocean_fault_recoveries(const ocean_t* c) { return c ? c->fault_recoveries : 0u; }

int ocean_create(ocean_t** out, uint32_t tile_size, uint32_t tiles, int device)
{
    if (!out || tiles == 0) return OCEAN_E_INVALID;
    *out = nullptr;
    if (tiles > 65535) return OCEAN_E_UNSUPPORTED;        // tiles are blockIdx.y of every launch
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
    c->cu_count = prop.multiProcessorCount;
    c->params.resize(tiles);
    for (auto& p : c->params) ocean_default_params(&p);
    int rc = OCEAN_OK;
    do {
        if (hipSetDevice(device) != hipSuccess) { rc = OCEAN_E_HIP; break; }
        for (int i = 0; i < MAXD && !rc; ++i)
            if (hipStreamCreateWithFlags(&c->own[i], hipStreamNonBlocking) != hipSuccess) rc = OCEAN_E_HIP;
        if (rc) break;
        if (hipHostMalloc((void**)&c->h_minmax, tiles * 2 * sizeof(unsigned), hipHostMallocDefault) != hipSuccess) { rc = OCEAN_E_HIP; break; }
        if (hipHostMalloc((void**)&c->fault, sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { rc = OCEAN_E_HIP; break; }
        *c->fault = 0u;
        if (hipEventCreate(&c->start_ev) != hipSuccess) { rc = OCEAN_E_HIP; break; }
        for (auto& e : c->end_ev) if (hipEventCreate(&e) != hipSuccess) { rc = OCEAN_E_HIP; break; }
        for (auto& row : c->mark_ev) for (auto& e : row) if (hipEventCreate(&e) != hipSuccess) { rc = OCEAN_E_HIP; break; }
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
    (void)sync_all(c);
    comm_release(c);
    release_import(c);
    free_device(c);
    if (c->h_minmax) (void)hipHostFree(c->h_minmax);
    if (c->fault) (void)hipHostFree(c->fault);
    if (c->grid_pos) (void)hipFree(c->grid_pos);
    if (c->mips_disp) (void)hipFree(c->mips_disp);
    if (c->mips_nrm) (void)hipFree(c->mips_nrm);
    if (c->grid_nrm) (void)hipFree(c->grid_nrm);
    if (c->consumer_ev) (void)hipEventDestroy(c->consumer_ev);
    if (c->start_ev) (void)hipEventDestroy(c->start_ev);
    for (auto& e : c->end_ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->z_done) if (e) (void)hipEventDestroy(e);
    for (auto& row : c->mark_ev) for (auto& e : row) if (e) (void)hipEventDestroy(e);
    if (c->nrm_final) (void)hipEventDestroy(c->nrm_final);
    for (auto& e : c->copy_done) if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (int i = 0; i < MAXD; ++i)
        if (c->own[i]) (void)hipStreamDestroy(c->own[i]);
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
    // SetLambda (.cpp:497-500): host state only.  The value reaches the device with the next frame: by value in
    // the launch arguments when every tile has the same lambda (always so for a single tile), otherwise through
    // one upload of the per-tile array at that frame -- a GUI "Apply" of several setters never drains the device.
    if (!c) return OCEAN_E_INVALID;
    if (tile != OCEAN_ALL_TILES && tile >= c->tiles) return OCEAN_E_INVALID;
    for (uint32_t i = 0; i < c->tiles; ++i)
        if (tile == OCEAN_ALL_TILES || tile == i) c->params[i].lambda = lambda;
    c->lambda_dirty = true;
    return OCEAN_OK;
}

int ocean_set_tile_size(ocean_t* c, uint32_t tile_size)
{
    if (!c) return OCEAN_E_INVALID;
    if (tile_size == 0 || (tile_size & (tile_size - 1))) return OCEAN_E_INVALID;   // .cpp:461-467
    if (!size_ok(tile_size)) return OCEAN_E_UNSUPPORTED;
    if (tile_size == c->n) return OCEAN_OK;
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    free_device(c);
    c->n = tile_size;
    release_import(c);
    c->ext_disp = nullptr; c->ext_nrm = nullptr;
    int rc = alloc_device(c);
    if (rc) return rc;
    c->lambda_dirty = true;
    if (c->use_toff)        // the per-tile time offsets are a property of the context, not of the buffers just re-created
        HIP_TRY(hipMemcpy(c->toff, c->toff_host.data(), c->tiles * sizeof(float), hipMemcpyHostToDevice));
    return OCEAN_OK;
}

uint32_t ocean_tile_size(const ocean_t* c) { return c ? c->n : 0; }
uint32_t ocean_tiles(const ocean_t* c) { return c ? c->tiles : 0; }

int ocean_prepare(ocean_t* c, uint64_t seed, const float* xi_or_null)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    // drain what is in flight, then start clean: no frame of the old spectrum is recovered (check_fault), the pipeline bookkeeping is empty
    for (int i = 0; i < MAXD; ++i) if (c->own[i]) (void)hipStreamSynchronize(c->own[i]);
    if (c->user) (void)hipStreamSynchronize(c->user);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (c->fault) *c->fault = 0u;
    reset_pipeline_state(c);
    for (bool& v : c->frame_valid) v = false;
    const size_t n = c->n, n2 = n * n, t = c->tiles;
    std::vector<TileParams> tp(t);
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
        tp[i].dispersion = c->dispersion;
        tp[i].dispersion_param = c->dispersion_param;
        tp[i].seed = seed + i;
    }
    SYNC_ALL(c);
    HIP_TRY(hipMemcpy(c->tparams, tp.data(), t * sizeof(TileParams), hipMemcpyHostToDevice));
    if (!c->xi) HIP_TRY(hipMalloc(&c->xi, t * n2 * sizeof(float2)));
    if (xi_or_null) HIP_TRY(hipMemcpy(c->xi, xi_or_null, t * n2 * sizeof(float2), hipMemcpyHostToDevice));
    {
        dim3 g((unsigned)((n + 255) / 256), (unsigned)t);
        hipLaunchKernelGGL(k_init_k1d, g, dim3(256), 0, stream_of(c, 0), c->k1d, c->tparams, (int)n);
        dim3 g2((unsigned)((n2 + 255) / 256), (unsigned)t);
        HIP_TRY(hipMemsetAsync(c->omega_q_overflow, 0, sizeof(unsigned), stream_of(c, 0)));
        hipLaunchKernelGGL(k_init_spectrum, g2, dim3(256), 0, stream_of(c, 0), c->h0, c->omega, c->omega_q, c->base_freq, c->omega_q_overflow,
                           xi_or_null ? (float2*)nullptr : c->xi, xi_or_null ? c->xi : (const float2*)nullptr,
                           c->k1d, c->tparams, (int)n);
    }
    HIP_TRY(hipGetLastError());
    if (c->h0_bits == 16) {
        if (!c->h0h) {
            HIP_TRY(hipMalloc(&c->h0h, t * n2 * sizeof(__half2)));
            HIP_TRY(hipMalloc(&c->h0_inv_scale, t * sizeof(float)));
            HIP_TRY(hipMalloc(&c->h0_maxbits, t * sizeof(unsigned)));
        }
        HIP_TRY(hipMemsetAsync(c->h0_maxbits, 0, t * sizeof(unsigned), stream_of(c, 0)));
        dim3 gh(1024, (unsigned)t);
        hipLaunchKernelGGL(k_h0_absmax, gh, dim3(256), 0, stream_of(c, 0), c->h0, c->h0_maxbits, n2);
        hipLaunchKernelGGL(k_h0_to_half, gh, dim3(256), 0, stream_of(c, 0), c->h0, c->h0h, c->h0_maxbits,
                           c->h0_inv_scale, n2);
        HIP_TRY(hipGetLastError());
    }
    if (c->inter_bits != c->inter_bits_zeroed) {
        // the two precisions lay the same elements out at 8 or 4 bytes each: what one wrote sits in the other's
        // padding columns, which must read as zero -> zero-fill every allocated chain again
        const size_t nu = n / 2 + 1, nup = (n / 2 + 16) & ~(size_t)15;
        SYNC_ALL(c);
        for (int i = 0; i < MAXD; ++i)
            if (c->z[i]) {
                HIP_TRY(hipMemsetAsync(c->z[i], 0, t * 3 * nu * 2 * nup * sizeof(float2), stream_of(c, i)));
                HIP_TRY(hipMemsetAsync(c->zh[i], 0, t * nu * nup * sizeof(float2), stream_of(c, i)));
                if (c->z3[i]) HIP_TRY(hipMemsetAsync(c->z3[i], 0, t * nu * 2 * nup * sizeof(float2), stream_of(c, i)));
            }
        c->inter_bits_zeroed = c->inter_bits;
    }
    {   // scales of the half2 intermediates and the gain of the Jacobian mode's cross derivative (both precisions)
        if (!c->zscale) {
            HIP_TRY(hipMalloc(&c->zscale, 2 * t * sizeof(float4)));
            HIP_TRY(hipMalloc(&c->zbounds, t * 2 * sizeof(unsigned)));
        }
        HIP_TRY(hipMemsetAsync(c->zbounds, 0, t * 2 * sizeof(unsigned), stream_of(c, 0)));
        hipLaunchKernelGGL(k_inter_bounds, dim3((unsigned)n, (unsigned)t), dim3(256), 0, stream_of(c, 0), c->h0, c->k1d, c->zbounds, (int)n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream_of(c, 0)));
        std::vector<unsigned> hb(2 * t);
        HIP_TRY(hipMemcpy(hb.data(), c->zbounds, 2 * t * sizeof(unsigned), hipMemcpyDeviceToHost));
        std::vector<float4> zs(2 * t);
        auto pow2_below = [](float x) {          // largest power of two <= x (x > 0)
            int e = 0;
            (void)std::frexp(x, &e);             // x = f * 2^e, f in [0.5, 1)
            return std::ldexp(1.0f, e - 1);
        };
        for (size_t i = 0; i < t; ++i) {
            float b[2];
            for (int k = 0; k < 2; ++k) std::memcpy(&b[k], &hb[2 * i + k], 4);
            // a component of a z-pass output is bounded by 2 * (column sum + mirrored column sum) <= 4 * max column
            // sum (of |h0| for the pairs weighted by unit vectors, of |k||h0| for those weighted by k); the largest
            // finite half is 65504: scale = the largest power of two with 4 * bound * scale <= 32768
            auto scale_for = [&](float bound4) { return (bound4 > 0.0f && std::isfinite(bound4)) ? pow2_below(32768.0f / bound4) : 1.0f; };
            const float su = scale_for(4.0f * b[0]), sk = scale_for(4.0f * b[1]);
            // pair 3 (Jacobian mode) packs the height with the k-weighted cross derivative; after ONE axis both parts of
            // the transform hold a mixture of the two, so they share one scale, and the cross part is first amplified by
            // a power of two g to the height's magnitude so that it keeps its relative precision
            const float g = (b[0] > 0.0f && b[1] > 0.0f && std::isfinite(b[0] / b[1])) ? pow2_below(b[0] / b[1]) : 1.0f;
            const float s3 = scale_for(4.0f * (b[0] + g * b[1]));
            zs[2 * i] = make_float4(su, sk, 1.0f / su, 1.0f / sk);
            zs[2 * i + 1] = make_float4(s3, g, 1.0f / s3, 1.0f / g);
        }
        HIP_TRY(hipMemcpy(c->zscale, zs.data(), 2 * t * sizeof(float4), hipMemcpyHostToDevice));
    }
    SYNC_ALL(c);
    {
        unsigned overflow = 1;
        HIP_TRY(hipMemcpy(&overflow, c->omega_q_overflow, sizeof(unsigned), hipMemcpyDeviceToHost));
        c->omega16 = overflow == 0;
#ifdef OCEAN_DEVELOPER      // A/B builds only (make variant ... DEFS=-DOCEAN_DEVELOPER): the shipped library reads no environment
        static const char* const w16_env = getenv("OCEAN_OMEGA16");
        if (w16_env && atoi(w16_env) == 0) c->omega16 = false;
#endif
    }
    c->seed = seed;
    c->prepared = true;
    c->have_frame = false;
    if (c->fault) *c->fault = 0u;
    return placement_search(c);
}

}  // extern "C"

// ---------------------------------------------------------------------------------
// frame launch
static const char* kernel_name_of(int idx)
{
    if (idx == 0) return "k_zpass";
    if (idx == 1) return "k_xpass_b";
    if (idx == 2) return "k_xpass_disp";
    return nullptr;
}

// Enqueues one frame.  pipelined = the call may use the pipeline chains (depth > 1).  redo_set >= 0: recover_fault runs chain redo_set's most
// recent frame again (same chain and regime; no burst / rotation bookkeeping, no hand-off form: c->merged_x is off by then).
static int enqueue_frame(ocean_ctx* c, float t, bool pipelined, hipEvent_t* marks, bool track = false, int redo_set = -1)
{
    if (!c) return OCEAN_E_INVALID;
    if (!c->prepared) return OCEAN_E_NOT_READY;
    (void)hipGetLastError();        // an earlier, unrelated error of this thread must not be taken for this frame's (launch_frame ends in hipGetLastError)
    const bool redo = redo_set >= 0;
    // depth D: frames f, f+1, ... run as D independent chains (own stream, own
    // intermediates, own map set) with no cross-stream dependency at all -- the first
    // pass of one frame fills the memory-idle phases of the others' map passes.
    // Caller-bound output buffers or a caller stream force depth 1.
    const bool pipe = pipelined && c->depth > 1 && !c->user && !c->ext_disp && !c->ext_nrm;
    if (!pipe && c->depth > 1 && pipelined && !redo) {  // leaving pipelined mode: drain the other chains first
        int rc_ = sync_all(c);
        if (rc_) return rc_;
    }
    const int set = redo ? redo_set : (pipe ? (int)(c->frame_ctr % (uint64_t)c->depth) : 0);
    {
        int rc_ = alloc_set(c, set);
        if (rc_) return rc_;
    }
    hipStream_t st = stream_of(c, set);
    if (c->gather_pending[set]) {        // this chain's maps are still being sent: the frame may not rewrite them yet
        HIP_TRY(hipStreamWaitEvent(st, c->gather_done[set], 0));
        c->gather_pending[set] = false;
    }
    if (c->mode == OCEAN_MODE_JACOBIAN) {
        int rc_ = alloc_jacobian(c, set);
        if (rc_) return rc_;
    }
    FrameArgs a;
    a.h0 = c->h0; a.omega = c->omega; a.k1d = c->k1d; a.tw = c->tw;
    a.omega_q = c->omega16 ? c->omega_q : nullptr; a.base_freq = c->base_freq;
    a.h0h = (c->h0_bits == 16) ? c->h0h : nullptr; a.h0_inv_scale = c->h0_inv_scale;
    a.zscale = c->zscale;
    a.z = c->z[set]; a.zh = c->zh[set]; a.hraw = c->hraw[set]; a.minmax = c->minmax[set]; a.hdone = c->hdone[set];
    a.zdone = c->zdone[set]; a.zdone_target = 0; a.poll_sleep = 0; a.fault = c->fault; c->cur_set = set;
    a.z3 = c->z3[set]; a.jraw = c->jraw[set]; a.jac0 = c->jac0[set];
    a.done_rec = c->done_rec[set]; a.done_ctr = track ? c->done_ctr[set] : nullptr;
    // The chain's sequence number is committed at once and never reused while the record buffer lives (a frame whose enqueue fails half way
    // must not leave its number to the next one: a stale completion record would match it); tracking and burst state are committed only once
    // the launches have succeeded, and a failed enqueue invalidates the chain's frame (see below: part of its launches may have run).
    unsigned frame_seq = c->seq[set] + 1u;
    if (frame_seq == 0) frame_seq = 1;                  // never 0: a fresh record buffer reads as "no frame"
    c->seq[set] = frame_seq;
    c->frame_valid[set] = false;
    a.frame_seq = frame_seq;
    a.disp = c->ext_disp ? c->ext_disp : c->dispN[set];
    a.nrm = c->ext_nrm ? c->ext_nrm : c->nrmN[set];
    a.toff = c->use_toff ? c->toff : nullptr;
    {   // choppiness: by value when all tiles agree, else the per-tile device array (uploaded when it changed)
        if (c->lambda_dirty) {
            c->lambda_uniform = true;
            for (uint32_t i = 1; i < c->tiles && c->lambda_uniform; ++i) c->lambda_uniform = c->params[i].lambda == c->params[0].lambda;
        }
        const bool uniform = c->lambda_uniform;
        a.lambda_all = c->params[0].lambda;
        a.lambda = uniform ? nullptr : c->lambda;
        if (!uniform && c->lambda_dirty) {
            int rc_ = sync_all(c);                      // frames in flight still read the old array
            if (rc_) return rc_;
            std::vector<float> l(c->tiles);
            for (uint32_t i = 0; i < c->tiles; ++i) l[i] = c->params[i].lambda;
            HIP_TRY(hipMemcpy(c->lambda, l.data(), c->tiles * sizeof(float), hipMemcpyHostToDevice));
        }
        c->lambda_dirty = false;
    }
    a.t = t;
    a.mode = c->mode;
    a.start_ramp = 0;       // (set per launch by the launcher where a staggered start pays: ocean_launch.h)
    a.disp_host = c->host_out[0]; a.nrm_host = c->host_out[1];      // (ocean_compute_waves_read, small maps; null otherwise)
    a.zmask = 15; a.xb_roles = 3;                   // (per launch: the launcher's frame order)
    a.xcd_rot = 0;
#ifdef OCEAN_DEVELOPER
    if (const char* xr = getenv("OCEAN_XCD_ROT")) a.xcd_rot = atoi(xr) % 7;      // (read per frame: tools/xcd_rot.py flips it between windows)
#endif
    a.rec_mode = track ? 2 : 1;                     // what the frame's LAST launch does with the completion records
    // maps beyond the memory-side cache, or several frames in flight: stream the maps past it
    const double texels = (double)c->tiles * (double)c->n * (double)c->n;
    // several frames in flight, or maps that would push everything else out of the 256 MiB memory-side cache anyway
    // (32 B/texel: 4096^2, 8 x 1024^2 ...): stream the maps past it.  A serial 2048^2 frame (134 MB of maps) is
    // faster with plain stores (76 vs 80 us), a serial 8 x 1024^2 batch (268 MB) with streamed ones (124 vs 150 us).
    int stream_maps = (pipe || texels * 32.0 > c->tune.maps_stream_frac * c->tune.cache_bytes) ? 3 : 0;
    if (c->inter_bits == 16) stream_maps |= 8;          // bit 3: half2 intermediates (kernel variant, not a store policy)
    if (pipe) {
        // what every frame re-reads -- spectrum + the intermediates of every chain in flight -- against the memory-side
        // cache: beyond it the intermediates are streamed too (bit 2, see store_z; the z pass then takes two columns per
        // workgroup so that the streamed stores cover whole lines).  Never for a serial frame: its x pass reads them right
        // behind the z pass (4096^2 serial 321-324 either way, 8 x 1024^2 serial 126 vs 136 us, depth 2 124 vs 114 us;
        // 2048^2 depth 3 -- 243 MB -- 57-58 plain vs 59-60 streamed, depth 4 -- 310 MB -- 61 vs 58.5;
        // profiles/r02_layout_experiments.txt).
        // (bytes per texel of a chain's intermediates: 14 of z / zh + 2 of raw height = 16, about 8 in the half2 form -- the
        //  figures the threshold was measured with; the Jacobian mode's pair 3 (16 of z) and its three real planes (6) make
        //  that 22, 14 in the half2 form)
        const bool jac = c->mode == OCEAN_MODE_JACOBIAN;
        const double inter = c->inter_bits == 16 ? (jac ? 14.0 : 8.0) : (jac ? 22.0 : 16.0);
        const double resident = texels * (10.0 + inter * c->depth);
        if (resident > c->tune.inter_stream_frac * c->tune.cache_bytes) stream_maps |= 4;
    }
    if (!pipe) stream_maps |= 16;                                               // this frame has the device to itself (which staggered start: ocean_launch.h)
#ifdef OCEAN_DEVELOPER      // A/B builds only: the shipped library reads no environment
    static const char* const stream_env = getenv("OCEAN_STREAM_MAPS");          // bit mask of the store policies
    if (stream_env) stream_maps = (atoi(stream_env) & 7) | (stream_maps & 24);
#endif
    // The first frames after a drain start STAGGERED: chains that begin together run the same kind of kernel side by side -- three z passes,
    // then three normal-map passes ... -- and stay in that lockstep for dozens of frames (in the steady state one chain runs half a period
    // away from the other two: profiles/r04_zpass_experiments.txt item 9); so frame k = 1 .. depth-1 of a fresh burst starts its z pass
    // behind the z pass of frame k-1 (one event each).
    // Bursts of 20 / 100 / 1000 frames at 2048^2, depth 3: 52.6 / 49.6 / 47.9 -> 51.6 / 48.0 / 47.3 us per frame (tools/burst_probe.py;
    // chaining EVERY frame's z pass instead costs 8 us per frame: profiles/r04_zpass_experiments.txt item 5).
    // (the burst state -- burst_pos, z_last_set -- is committed below, once the launches have succeeded; the stream wait enqueued here is
    //  harmless if they do not: a wait for an event that has been recorded already)
    c->after_z = nullptr;
    const bool stagger = pipe && !redo && c->burst_pos < c->depth;
    if (stagger) {
        if (!c->z_done[set]) HIP_TRY(hipEventCreateWithFlags(&c->z_done[set], hipEventDisableTiming));
        if (c->z_last_set >= 0 && c->z_last_set != set && c->z_done[c->z_last_set]) HIP_TRY(hipStreamWaitEvent(st, c->z_done[c->z_last_set], 0));
        c->after_z = c->z_done[set];
    }
    hipError_t e = hipErrorInvalidValue;
    if (c->n <= 256) e = ocean_launch_frame_small(c, a, stream_maps, st, marks);
    else if (c->n <= 1024) e = ocean_launch_frame_mid(c, a, stream_maps, st, marks);
    else if (c->n == 2048) e = ocean_launch_frame_2048(c, a, stream_maps, st, marks);
    else if (c->n == 4096) e = ocean_launch_frame_4096(c, a, stream_maps, st, marks);
    else { c->after_z = nullptr; return OCEAN_E_UNSUPPORTED; }
    c->after_z = nullptr;
    if (e != hipSuccess) {
        // some of the frame's launches may have run: the chain's records, height keys and maps no longer describe ONE frame -- the chain has
        // no frame until the next successful enqueue (wait / read-out: OCEAN_E_NOT_READY), the other chains are untouched
        g_last_hip = (int)e;
        if (c->last_set == set) c->have_frame = false;
        c->zgen[set] = 0;                                   // (a one-launch frame may have counted itself without running: start over)
        (void)hipMemsetAsync(c->zdone[set], 0, c->tiles * sizeof(unsigned), st);
        return OCEAN_E_HIP;
    }
    if (pipe && !redo) {
        if (stagger) c->z_last_set = set;
        if (c->burst_pos < MAXD + 1) c->burst_pos++;
    }
    c->frame_valid[set] = true;
    c->tracked[set] = track;
    c->last_t[set] = t; c->last_pipe[set] = pipelined; c->last_handoff[set] = c->handoff;
    if (pipe && !redo) c->frame_ctr++;
    c->have_frame = true;
    c->last_set = set;
    return OCEAN_OK;
}

// ---- placement search (OceanTuning::placement_trials; include/ocean_dev.h: ocean_set_placement_search) ---------------------------------------
// The buffer group the frame's first pass reads and writes -- spectrum, dispersion (both forms), the intermediates of chain 0 -- is allocated
// `trials` times, all candidates alive at once (freed memory would come straight back), filled alike, and serial frames are timed on each in
// turn behind a common warm-up (the shader clock needs ~25 ms of load after an idle gap); the context keeps the fastest group and frees the
// rest.  Frames do not depend on where their buffers are: same bits.  Skipped where calibration frames must not run (caller-owned stream or
// output) or the group is not the usual one (fp16 copy of the spectrum).
static int placement_search(ocean_ctx* c)
{
    // once per allocation: a placement keeps its speed for as long as it lives, so a repeated Prepare on the same buffers (a parameter change
    // in the reference's GUI) costs nothing extra and keeps the first search's report; a resize or a new candidate count searches again
    if (c->placement_done) return OCEAN_OK;
    c->placement_tried = 0; c->placement_us_chosen = c->placement_us_worst = 0.0f;
    const size_t n = c->n, n2 = n * n, t = c->tiles;
    const size_t nu = n / 2 + 1, nup = (n / 2 + 16) & ~(size_t)15;
    enum { NB = 7 };                                // spectrum, dispersion (two forms), chain 0's intermediates (three), chain 0's maps
    const size_t bytes[NB] = {t * n2 * sizeof(float2), t * n2 * sizeof(float), t * n2 * sizeof(uint16_t), t * 3 * nu * 2 * nup * sizeof(float2),
                              t * nu * nup * sizeof(float2), t * nup * n * sizeof(float), c->maps_bytes[0]};
    // which of them differ between candidates (developer switch for attribution runs: tools/placement_probe.py)
    unsigned mask = c->tune.placement_mask;
    bool trace = false;
#ifdef OCEAN_DEVELOPER      // A/B builds only (make variant ... DEFS=-DOCEAN_DEVELOPER): the shipped library reads no environment
    if (const char* e = std::getenv("OCEAN_PLACEMENT_MASK")) mask = (unsigned)std::strtoul(e, nullptr, 0);
    trace = std::getenv("OCEAN_PLACEMENT_TRACE") != nullptr;
#endif
    mask &= (1u << NB) - 1;
    size_t group = 0;
    for (int b = 0; b < NB; ++b) if (mask >> b & 1) group += bytes[b];
    int trials = c->placement_override > 0 ? c->placement_override
                                           : ((c->n >= c->tune.placement_min_n && group <= c->tune.placement_max_group_bytes) ? c->tune.placement_trials : 1);
    if (trials > 16) trials = 16;
    if (trials <= 1 || !mask) { c->placement_done = true; return OCEAN_OK; }
    if (c->user || c->ext_disp || c->ext_nrm || c->h0_bits == 16) return OCEAN_OK;          // (not now: a later Prepare may)
    c->placement_done = true;
    { int rc_ = alloc_set(c, 0); if (rc_) return rc_; }
    hipStream_t st = stream_of(c, 0);
    struct Group { void* p[NB]; };
    std::vector<Group> cand((size_t)trials);
    cand[0] = Group{{c->h0, c->omega, c->omega_q, c->z[0], c->zh[0], c->hraw[0], c->dispN[0]}};
    auto varies = [&](int b) { return (mask >> b & 1) != 0; };
    auto release = [&](Group& g) { for (int b = 0; b < NB; ++b) if (varies(b) && g.p[b]) { (void)hipFree(g.p[b]); g.p[b] = nullptr; } };
    for (int k = 1; k < trials; ++k) {
        Group& g = cand[(size_t)k];
        bool ok = true;
        for (int b = 0; b < NB; ++b) g.p[b] = varies(b) ? nullptr : cand[0].p[b];
        for (int b = 0; b < NB && ok; ++b) if (varies(b)) ok = hipMalloc(&g.p[b], bytes[b]) == hipSuccess;
        for (int b = 0; b < 3 && ok; ++b) if (varies(b)) ok = hipMemcpyAsync(g.p[b], cand[0].p[b], bytes[b], hipMemcpyDeviceToDevice, st) == hipSuccess;
        for (int b = 3; b < 6 && ok; ++b) if (varies(b)) ok = hipMemsetAsync(g.p[b], 0, bytes[b], st) == hipSuccess;       // (padded columns must read as zero)
        if (!ok) { (void)hipGetLastError(); release(g); trials = k; break; }       // out of memory: search among what there is
    }
    if (trials <= 1) return OCEAN_OK;
    auto use = [&](const Group& g) {
        c->h0 = static_cast<float2*>(g.p[0]); c->omega = static_cast<float*>(g.p[1]); c->omega_q = static_cast<uint16_t*>(g.p[2]);
        c->z[0] = static_cast<float2*>(g.p[3]); c->zh[0] = static_cast<float2*>(g.p[4]); c->hraw[0] = static_cast<float*>(g.p[5]);
        c->dispN[0] = static_cast<float4*>(g.p[6]); c->nrmN[0] = c->dispN[0] + t * n2;
    };
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = OCEAN_OK;
    auto frames = [&](int count) { for (int j = 0; j < count && rc == OCEAN_OK; ++j) rc = enqueue_frame(c, 0.05f * (float)j, false, nullptr, false); };
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = OCEAN_E_HIP;
    // warm-up on the first candidate: 30 ms of frames (host clock), so that every candidate is timed at the sustained shader clock
    const auto w0 = std::chrono::steady_clock::now();
    while (rc == OCEAN_OK && std::chrono::steady_clock::now() - w0 < std::chrono::milliseconds(30)) {
        frames(32);
        if (rc == OCEAN_OK && hipStreamSynchronize(st) != hipSuccess) rc = OCEAN_E_HIP;
    }
    const int timed = c->n >= 4096 ? 16 : 40;
    std::vector<float> us((size_t)trials, 0.0f);
    for (int k = 0; k < trials && rc == OCEAN_OK; ++k) {
        use(cand[(size_t)k]);
        frames(8);
        if (rc == OCEAN_OK && hipEventRecord(e0, st) != hipSuccess) rc = OCEAN_E_HIP;
        frames(timed);
        if (rc == OCEAN_OK && (hipEventRecord(e1, st) != hipSuccess || hipEventSynchronize(e1) != hipSuccess)) rc = OCEAN_E_HIP;
        float ms = 0.0f;
        if (rc == OCEAN_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = OCEAN_E_HIP;
        us[(size_t)k] = ms * 1000.0f / (float)timed;
    }
    (void)hipStreamSynchronize(st);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    int best = 0;
    if (rc == OCEAN_OK) {
        for (int k = 1; k < trials; ++k) if (us[(size_t)k] < us[(size_t)best]) best = k;
        c->placement_tried = trials;
        c->placement_us_chosen = us[(size_t)best];
        c->placement_us_worst = *std::max_element(us.begin(), us.end());
        if (trace) {
            std::fprintf(stderr, "ocean placement search (mask 0x%x):", mask);
            for (int k = 0; k < trials; ++k) std::fprintf(stderr, " %.2f", (double)us[(size_t)k]);
            std::fprintf(stderr, " us/frame\n");
        }
    } else {
        (void)hipGetLastError();
    }
    use(cand[(size_t)best]);
    for (int k = 0; k < trials; ++k) if (k != best) release(cand[(size_t)k]);
    c->have_frame = false;                      // (the maps hold a calibration frame: nothing to read out until the caller's first frame)
    for (bool& v : c->frame_valid) v = false;
    return rc;
}

extern "C" {

int ocean_set_placement_search(ocean_t* c, int trials)
{
    if (!c || trials < 0) return OCEAN_E_INVALID;
    if (trials != c->placement_override) c->placement_done = false;
    c->placement_override = trials;
    return OCEAN_OK;
}

int ocean_placement_report(const ocean_t* c, int* trials, float* us_chosen, float* us_worst)
{
    if (!c) return OCEAN_E_INVALID;
    if (trials) *trials = c->placement_tried;
    if (us_chosen) *us_chosen = c->placement_us_chosen;
    if (us_worst) *us_worst = c->placement_us_worst;
    return OCEAN_OK;
}

int ocean_compute_waves_async(ocean_t* c, float t)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    return enqueue_frame(c, t, true, nullptr, c->track_async);
}

int ocean_set_frame_tracking(ocean_t* c, int on)
{
    if (!c) return OCEAN_E_INVALID;
    c->track_async = on != 0;
    return OCEAN_OK;
}

int ocean_synchronize(ocean_t* c)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    return OCEAN_OK;
}

// Waits until every workgroup of the chain's most recently enqueued frame has finished and copies its height keys to
// c->h_minmax: a bounded poll of the completion records the frame's last workgroup writes into host-coherent memory
// (ocean_kernels.h: frame_done), then -- the device is busy with something long, or shared -- a stream synchronisation.
static int wait_frame(ocean_ctx* c, int set)
{
    if (!c->have_frame || !c->done_rec[set] || !c->frame_valid[set]) return OCEAN_E_NOT_READY;
    for (int attempt = 0; ; ++attempt) {
    const unsigned want = c->seq[set];
    const volatile uint4* rec = c->done_rec[set];
    bool synced = false;
    // A completion record says that every workgroup of the frame has issued its stores; it is not a memory fence.  Readers on the
    // context's own streams are ordered by the stream.  Wherever somebody ELSE may read the maps right after this call -- caller-bound
    // or imported output (ocean_bind_output / ocean_bind_output_dmabuf), a map set that has been exported (ocean_export_maps) or whose
    // device pointers have been handed out (ocean_device_maps) -- the wait is a stream synchronisation, i.e. the end-of-kernel release
    // and write-back have happened when it returns: the old "synchronous" guarantee, where external readers exist.
    const bool external = c->ext_disp || c->ext_nrm || c->maps_shared;
    if (!c->tracked[set] || external) {           // (untracked: the records arrive early in the frame's last kernel, only the stream tells when it has finished)
        HIP_TRY(hipStreamSynchronize(stream_of(c, set)));
        synced = true;
    }
    using clock = std::chrono::steady_clock;
    clock::time_point t0;
    bool timed = false;
    unsigned spins = 0;
    for (uint32_t i = 0; i < c->tiles; ++i) {
        while (__atomic_load_n(&rec[i].z, __ATOMIC_ACQUIRE) != want) {
            if (synced) { g_last_hip = (int)hipErrorUnknown; return OCEAN_E_HIP; }      // the stream has drained and the record is not there
            __builtin_ia32_pause();
            if ((++spins & 255u) == 0) {
                if (!timed) { t0 = clock::now(); timed = true; }
                else if (clock::now() - t0 > std::chrono::milliseconds(2)) {
                    HIP_TRY(hipStreamSynchronize(stream_of(c, set)));
                    synced = true;
                }
            }
        }
        c->h_minmax[2 * i + 0] = rec[i].x;
        c->h_minmax[2 * i + 1] = rec[i].y;
    }
    if (!fault_raised(c) || attempt > 0) break;
    // an in-launch wait of some frame of this context gave up: recover (this chain's frame is run again if it was one of them -- new
    // sequence number, everything drained) and read the records once more
    { int rc_ = recover_fault(c); if (rc_) return rc_; }
    if (!c->frame_valid[set]) return OCEAN_E_NOT_READY;
    }
    return fault_raised(c) ? OCEAN_E_HIP : OCEAN_OK;
}

static float amp_of(const ocean_ctx* c, uint32_t tile, float* mn_out, float* mx_out)
{
    const float mn = key_float(c->h_minmax[2 * tile]), mx = key_float(c->h_minmax[2 * tile + 1]);
    if (mn_out) *mn_out = mn;
    if (mx_out) *mx_out = mx;
    return std::fmax(std::fabs(mn), std::fabs(mx));   // .cpp:448
}

int ocean_wait_frame(ocean_t* c, float* out_amp)
{
    if (!c) return OCEAN_E_INVALID;
    if (!c->prepared || !c->have_frame) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    const int rc = wait_frame(c, c->last_set);
    if (rc) return rc;
    if (out_amp)
        for (uint32_t i = 0; i < c->tiles; ++i) out_amp[i] = amp_of(c, i, nullptr, nullptr);
    return OCEAN_OK;
}

int ocean_compute_waves(ocean_t* c, float t, float* out_amp)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    int rc = enqueue_frame(c, t, true, nullptr, true);
    if (rc) return rc;
    return ocean_wait_frame(c, out_amp);        // waits for this frame only
}

int ocean_set_time_offsets(ocean_t* c, const float* offsets)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    if (!offsets) { c->use_toff = false; c->toff_host.clear(); return OCEAN_OK; }
    c->toff_host.assign(offsets, offsets + c->tiles);
    HIP_TRY(hipMemcpy(c->toff, c->toff_host.data(), c->tiles * sizeof(float), hipMemcpyHostToDevice));
    c->use_toff = true;
    return OCEAN_OK;
}

int ocean_get_heights(ocean_t* c, uint32_t tile, float* amp, float* min_h, float* max_h)
{
    if (!c || tile >= c->tiles) return OCEAN_E_INVALID;
    if (!c->prepared) return OCEAN_E_NOT_READY;
    if (!c->have_frame) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    int rc = wait_frame(c, c->last_set);
    if (rc) return rc;
    const float a = amp_of(c, tile, min_h, max_h);
    if (amp) *amp = a;
    return OCEAN_OK;
}

int ocean_read_maps(ocean_t* c, uint32_t first, uint32_t count, float* disp, float* nrm)
{
    if (!c || first >= c->tiles || count == 0 || first + count > c->tiles) return OCEAN_E_INVALID;
    if (!c->prepared || !c->have_frame) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    const float4* d = (c->ext_disp ? c->ext_disp : c->dispN[c->last_set]) + first * n2;
    const float4* q = (c->ext_nrm ? c->ext_nrm : c->nrmN[c->last_set]) + first * n2;
    SYNC_ALL(c);
    if (disp) HIP_TRY(hipMemcpy(disp, d, count * n2 * sizeof(float4), hipMemcpyDeviceToHost));
    if (nrm) HIP_TRY(hipMemcpy(nrm, q, count * n2 * sizeof(float4), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}

// ComputeWaves + read-out as ONE blocking call (the reference's call shape: WaterSurfaceMesh.cpp:145-154 calls ComputeWaves, :701-755 copies both
// maps).  The normal map is final behind the frame's second launch: its copy starts there, on a copy stream, and runs beside the displacement
// pass; the displacement map's copy follows its kernel on the frame's stream.  The call returns from a poll of the two copies' events.
int ocean_compute_waves_read(ocean_t* c, float t, float* out_amp, float* disp, float* nrm)
{
    if (!c || !disp || !nrm) return OCEAN_E_INVALID;
    if (!c->prepared) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    if (!c->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    if (!c->nrm_final) HIP_TRY(hipEventCreateWithFlags(&c->nrm_final, hipEventDisableTiming));
    for (auto& ev : c->copy_done) if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const size_t bytes = (size_t)c->tiles * c->n * c->n * sizeof(float4);
    // Small maps into page-locked memory: the x passes store them there themselves, beside the device copy (OceanTuning::host_store_max_bytes;
    // ranges registered through ocean_host_register are found in the library's own list, others through the runtime).
    void* dev_dst[2] = {nullptr, nullptr};
    if (bytes <= c->tune.host_store_max_bytes) {
        dev_dst[0] = pinned_device_address(disp, bytes);
        dev_dst[1] = pinned_device_address(nrm, bytes);
    }
    const bool direct = dev_dst[0] && dev_dst[1];
    c->host_out[0] = direct ? static_cast<float4*>(dev_dst[0]) : nullptr;
    c->host_out[1] = direct ? static_cast<float4*>(dev_dst[1]) : nullptr;
    c->after_b = direct ? nullptr : c->nrm_final; c->after_b_recorded = false;
    int rc = enqueue_frame(c, t, true, nullptr, true);
    c->after_b = nullptr; c->host_out[0] = c->host_out[1] = nullptr;
    if (rc) return rc;
    const int set = c->last_set;
    hipStream_t st = stream_of(c, set);
    const float4* d = c->ext_disp ? c->ext_disp : c->dispN[set];
    const float4* q = c->ext_nrm ? c->ext_nrm : c->nrmN[set];
    hipStream_t nst = st;
    if (!direct) {
        if (c->after_b_recorded) {      // (frames whose x axis is one launch -- small tiles -- have no such point: both copies follow the frame)
            HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->nrm_final, 0));
            nst = c->copy_stream;
        }
        HIP_TRY(hipMemcpyAsync(nrm, q, bytes, hipMemcpyDeviceToHost, nst));
        hipError_t ce = hipEventRecord(c->copy_done[0], nst);
        if (ce == hipSuccess) ce = hipMemcpyAsync(disp, d, bytes, hipMemcpyDeviceToHost, st);
        if (ce != hipSuccess) {         // no copy into the caller's memory may be in flight when the call returns with an error
            g_last_hip = (int)ce; (void)hipGetLastError();
            (void)hipStreamSynchronize(nst); (void)hipStreamSynchronize(st);
            return ce == hipErrorOutOfMemory ? OCEAN_E_NOMEM : OCEAN_E_HIP;
        }
    }
    // (direct stores: the event behind the frame's last kernel says that its stores -- the host's included -- have been released)
    if (hipError_t ee = hipEventRecord(c->copy_done[1], st); ee != hipSuccess) {
        g_last_hip = (int)ee; (void)hipGetLastError(); (void)hipStreamSynchronize(nst); (void)hipStreamSynchronize(st);
        return OCEAN_E_HIP;
    }
    rc = wait_frame(c, set);        // the frame's completion records (a poll): A, min, max -- the maps may still be on their way
    if (rc == OCEAN_OK && c->fault_recoveries_seen != c->fault_recoveries) {
        // the frame was run again (an in-launch wait had given up): what reached the host may be the wrong frame's -- copy again, plainly
        c->fault_recoveries_seen = c->fault_recoveries;
        HIP_TRY(hipStreamSynchronize(nst)); HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(nrm, q, bytes, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(disp, d, bytes, hipMemcpyDeviceToHost));
    }
    // the copies / the frame's release: polled like the records -- a blocking synchronisation's wake-up costs 13-16 us per call, and this call
    // is the caller's whole frame (the reference's ComputeWaves keeps every core busy for its 44 ms) -- for up to 50 ms, then waited for
    using clock = std::chrono::steady_clock;
    const clock::time_point t0 = clock::now();
    for (int k = direct ? 1 : 0; k < 2; ++k) {
        unsigned spins = 0;
        for (;;) {
            const hipError_t e = hipEventQuery(c->copy_done[k]);
            if (e == hipSuccess) break;
            if (e != hipErrorNotReady) { g_last_hip = (int)e; (void)hipGetLastError(); (void)hipStreamSynchronize(nst); (void)hipStreamSynchronize(st); return OCEAN_E_HIP; }
            if ((++spins & 255u) == 0 && clock::now() - t0 > std::chrono::milliseconds(50)) { HIP_TRY(hipEventSynchronize(c->copy_done[k])); break; }
            __builtin_ia32_pause();
        }
    }
    if (rc) return rc;
    if (out_amp)
        for (uint32_t i = 0; i < c->tiles; ++i) out_amp[i] = amp_of(c, i, nullptr, nullptr);
    return OCEAN_OK;
}

int ocean_host_register(void* host_ptr, size_t bytes)
{
    if (!host_ptr || bytes == 0) return OCEAN_E_INVALID;
    HIP_TRY(hipHostRegister(host_ptr, bytes, hipHostRegisterDefault));
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, host_ptr, 0) == hipSuccess && dp) {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        g_pinned.push_back({static_cast<char*>(host_ptr), bytes, static_cast<char*>(dp)});
    } else {
        (void)hipGetLastError();
    }
    return OCEAN_OK;
}

int ocean_host_unregister(void* host_ptr)
{
    if (!host_ptr) return OCEAN_E_INVALID;
    {
        std::lock_guard<std::mutex> lock(g_pinned_mutex);
        for (size_t i = 0; i < g_pinned.size(); ++i)
            if (g_pinned[i].host == host_ptr) { g_pinned.erase(g_pinned.begin() + (long)i); break; }
    }
    HIP_TRY(hipHostUnregister(host_ptr));
    return OCEAN_OK;
}

int ocean_read_maps_async(ocean_t* c, uint32_t first, uint32_t count, float* disp, float* nrm)
{
    if (!c || first >= c->tiles || count == 0 || first + count > c->tiles) return OCEAN_E_INVALID;
    if (!c->prepared || !c->have_frame) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    const float4* d = (c->ext_disp ? c->ext_disp : c->dispN[c->last_set]) + first * n2;
    const float4* q = (c->ext_nrm ? c->ext_nrm : c->nrmN[c->last_set]) + first * n2;
    hipStream_t st = stream_of(c, c->last_set);          // ordered after the frame that wrote these maps
    if (disp) HIP_TRY(hipMemcpyAsync(disp, d, count * n2 * sizeof(float4), hipMemcpyDeviceToHost, st));
    if (nrm) HIP_TRY(hipMemcpyAsync(nrm, q, count * n2 * sizeof(float4), hipMemcpyDeviceToHost, st));
    return OCEAN_OK;
}

size_t ocean_staging_map_offset(size_t vertices_bytes, size_t indices_bytes)
{
    // AlignSizeTo(verticesSize + indicesSize, FormatToBytes(RGBA32F) = 16): WaterSurfaceMesh.cpp:19-22, 721-724
    return (vertices_bytes + indices_bytes + 15u) & ~(size_t)15u;
}

int ocean_read_maps_staging(ocean_t* c, uint32_t tile, void* mapped_base, size_t vertices_bytes, size_t indices_bytes,
                            size_t* bytes_to_flush)
{
    if (!c || tile >= c->tiles || !mapped_base) return OCEAN_E_INVALID;
    if (!c->prepared || !c->have_frame) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    const size_t map_bytes = (size_t)c->n * c->n * sizeof(float4);
    const size_t off = ocean_staging_map_offset(vertices_bytes, indices_bytes);
    const float4* d = (c->ext_disp ? c->ext_disp : c->dispN[c->last_set]) + (size_t)tile * c->n * c->n;
    const float4* q = (c->ext_nrm ? c->ext_nrm : c->nrmN[c->last_set]) + (size_t)tile * c->n * c->n;
    hipStream_t st = stream_of(c, c->last_set);           // ordered behind the frame that wrote these maps
    char* base = static_cast<char*>(mapped_base);
    // [vertices | indices | pad to 16 | displacements | normals]  (WaterSurfaceMesh.cpp:712-744)
    HIP_TRY(hipMemcpyAsync(base + off, d, map_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(base + off + map_bytes, q, map_bytes, hipMemcpyDeviceToHost, st));
    if (bytes_to_flush) *bytes_to_flush = off + 2 * map_bytes;      // the size the reference flushes from offset 0 (.cpp:746-753)
    return OCEAN_OK;
}

int ocean_device_maps(ocean_t* c, void** d_disp, void** d_nrm)
{
    if (!c) return OCEAN_E_INVALID;
    if (d_disp) *d_disp = c->ext_disp ? (void*)c->ext_disp : (void*)c->dispN[c->last_set];
    if (d_nrm) *d_nrm = c->ext_nrm ? (void*)c->ext_nrm : (void*)c->nrmN[c->last_set];
    c->maps_shared = true;              // somebody outside the context's streams may read the maps from now on: see wait_frame
    return OCEAN_OK;
}

int ocean_set_external_readers(ocean_t* c, int on)
{
    if (!c) return OCEAN_E_INVALID;
    c->maps_shared = on != 0;
    return OCEAN_OK;
}

int ocean_export_maps(ocean_t* c, int* dmabuf_fd, size_t* disp_offset, size_t* nrm_offset, size_t* bytes, int* map_set)
{
    if (!c || !dmabuf_fd) return OCEAN_E_INVALID;
    *dmabuf_fd = -1;
    if (c->ext_disp || c->ext_nrm) return OCEAN_E_UNSUPPORTED;          // caller-bound output: the caller owns (and exports) that memory
    HIP_TRY(hipSetDevice(c->device));
    const int set = c->have_frame ? c->last_set : 0;
    {
        int rc_ = alloc_set(c, set);
        if (rc_) return rc_;
    }
    int fd = -1;
    HIP_TRY(hipMemGetHandleForAddressRange(&fd, (hipDeviceptr_t)c->dispN[set], c->maps_bytes[set], hipMemRangeHandleTypeDmaBufFd, 0));
    *dmabuf_fd = fd;
    c->maps_shared = true;              // an importer reads the maps outside the context's streams from now on: see wait_frame
    if (disp_offset) *disp_offset = 0;
    if (nrm_offset) *nrm_offset = (size_t)c->tiles * c->n * c->n * sizeof(float4);
    if (bytes) *bytes = c->maps_bytes[set];
    if (map_set) *map_set = set;
    return OCEAN_OK;
}

static void release_import(ocean_ctx* c)
{
    if (c->import_mem) { (void)hipDestroyExternalMemory(c->import_mem); c->import_mem = nullptr; c->import_base = nullptr; }
}

int ocean_bind_output(ocean_t* c, void* d_disp, void* d_nrm)
{
    if (!c) return OCEAN_E_INVALID;
    if (((uintptr_t)d_disp | (uintptr_t)d_nrm) & 15u) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    release_import(c);                  // a binding made by ocean_bind_output_dmabuf ends here
    c->ext_disp = (float4*)d_disp;
    c->ext_nrm = (float4*)d_nrm;
    return OCEAN_OK;
}

int ocean_bind_output_dmabuf(ocean_t* c, int dmabuf_fd, size_t bytes, size_t disp_offset, size_t nrm_offset)
{
    if (!c || dmabuf_fd < 0) return OCEAN_E_INVALID;
    const size_t map_bytes = (size_t)c->tiles * c->n * c->n * sizeof(float4);
    if ((disp_offset | nrm_offset) & 15u) return OCEAN_E_INVALID;
    if (map_bytes > bytes || disp_offset > bytes - map_bytes || nrm_offset > bytes - map_bytes) return OCEAN_E_INVALID;     // (no sum that could wrap)
    if (disp_offset < nrm_offset + map_bytes && nrm_offset < disp_offset + map_bytes) return OCEAN_E_INVALID;      // the two maps overlap
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    hipExternalMemoryHandleDesc hd = {};
    hd.type = hipExternalMemoryHandleTypeOpaqueFd;        // what amdgpu's dma-buf descriptors are imported as
    hd.handle.fd = dmabuf_fd;
    hd.size = bytes;
    hipExternalMemory_t mem = nullptr;
    HIP_TRY(hipImportExternalMemory(&mem, &hd));
    hipExternalMemoryBufferDesc bd = {};
    bd.offset = 0; bd.size = bytes;
    void* base = nullptr;
    hipError_t e = hipExternalMemoryGetMappedBuffer(&base, mem, &bd);
    if (e != hipSuccess) { (void)hipDestroyExternalMemory(mem); g_last_hip = (int)e; return OCEAN_E_HIP; }
    release_import(c);
    c->import_mem = mem; c->import_base = base;
    c->ext_disp = reinterpret_cast<float4*>(static_cast<char*>(base) + disp_offset);
    c->ext_nrm = reinterpret_cast<float4*>(static_cast<char*>(base) + nrm_offset);
    return OCEAN_OK;
}

}  // extern "C"

// The consumer kernels (vertex stage, mip chain) write context-wide output buffers and run on the stream of the frame they read.
// At pipeline depth > 1 consecutive consumer calls land on different, mutually unordered chain streams: each call first makes
// its stream wait for the previous consumer launch, so that two of them never write those buffers at once.
static int consumer_begin(ocean_ctx* c, hipStream_t st)
{
    // (a frame whose in-launch wait has given up by now is recovered before anything consumes it; one that gives up later is reported
    //  by the next wait / synchronisation: recover_fault)
    { int rc_ = check_fault(c); if (rc_) return rc_; }
    if (c->consumer_pending && c->consumer_stream != st) HIP_TRY(hipStreamWaitEvent(st, c->consumer_ev, 0));
    return OCEAN_OK;
}
static int consumer_end(ocean_ctx* c, hipStream_t st)
{
    if (!c->consumer_ev) HIP_TRY(hipEventCreateWithFlags(&c->consumer_ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->consumer_ev, st));
    c->consumer_stream = st;
    c->consumer_pending = true;
    return OCEAN_OK;
}
#define CONSUMER_BEGIN(c, st) do { int rc_ = consumer_begin(c, st); if (rc_) return rc_; } while (0)
#define CONSUMER_END(c, st) do { int rc_ = consumer_end(c, st); if (rc_) return rc_; } while (0)

extern "C" {

int ocean_displace_grid(ocean_t* c, uint32_t tile, uint32_t grid_size, float vertex_distance, float uv_scale, float choppy)
{
    if (!c || tile >= c->tiles || grid_size == 0 || grid_size > 8192) return OCEAN_E_INVALID;
    if (!c->prepared || !c->have_frame) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t verts = (grid_size + 1) * (grid_size + 1);
    if (verts > c->grid_capacity) {
        SYNC_ALL(c);
        if (c->grid_pos) (void)hipFree(c->grid_pos);
        if (c->grid_nrm) (void)hipFree(c->grid_nrm);
        c->grid_pos = c->grid_nrm = nullptr; c->grid_capacity = 0;
        HIP_TRY(hipMalloc(&c->grid_pos, (size_t)verts * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->grid_nrm, (size_t)verts * sizeof(float4)));
        c->grid_capacity = verts;
    }
    const size_t n2 = (size_t)c->n * c->n;
    GridArgs g;
    g.disp = (c->ext_disp ? c->ext_disp : c->dispN[c->last_set]) + tile * n2;
    g.nrm = (c->ext_nrm ? c->ext_nrm : c->nrmN[c->last_set]) + tile * n2;
    g.minmax = c->minmax[c->last_set] + 2 * tile;
    g.positions = c->grid_pos; g.normals = c->grid_nrm;
    g.n = (int)c->n; g.grid = (int)grid_size;
    g.vertex_distance = vertex_distance; g.uv_scale = uv_scale; g.choppy = choppy;
    // ordered after the frame that wrote these maps
    CONSUMER_BEGIN(c, stream_of(c, c->last_set));
    hipLaunchKernelGGL(k_displace_grid, dim3((verts + 255) / 256), dim3(256), 0, stream_of(c, c->last_set), g);
    HIP_TRY(hipGetLastError());
    CONSUMER_END(c, stream_of(c, c->last_set));
    c->grid_vertices = verts;
    return OCEAN_OK;
}

int ocean_displace_grid_cascades(ocean_t* c, uint32_t first_tile, uint32_t count, uint32_t grid_size, float vertex_distance,
                                 const float* uv_scales, float choppy)
{
    if (!c || !uv_scales || count == 0 || count > (uint32_t)OCEAN_MAX_CASCADES || first_tile >= c->tiles || first_tile + count > c->tiles ||
        grid_size == 0 || grid_size > 8192)
        return OCEAN_E_INVALID;
    if (!c->prepared || !c->have_frame) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t verts = (grid_size + 1) * (grid_size + 1);
    if (verts > c->grid_capacity) {
        SYNC_ALL(c);
        if (c->grid_pos) (void)hipFree(c->grid_pos);
        if (c->grid_nrm) (void)hipFree(c->grid_nrm);
        c->grid_pos = c->grid_nrm = nullptr; c->grid_capacity = 0;
        HIP_TRY(hipMalloc(&c->grid_pos, (size_t)verts * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->grid_nrm, (size_t)verts * sizeof(float4)));
        c->grid_capacity = verts;
    }
    const size_t n2 = (size_t)c->n * c->n;
    CascadeArgs a;
    a.g.disp = (c->ext_disp ? c->ext_disp : c->dispN[c->last_set]) + first_tile * n2;
    a.g.nrm = (c->ext_nrm ? c->ext_nrm : c->nrmN[c->last_set]) + first_tile * n2;
    a.g.minmax = c->minmax[c->last_set] + 2 * first_tile;
    a.g.positions = c->grid_pos; a.g.normals = c->grid_nrm;
    a.g.n = (int)c->n; a.g.grid = (int)grid_size;
    a.g.vertex_distance = vertex_distance; a.g.uv_scale = 1.0f; a.g.choppy = choppy;
    a.count = (int)count; a.tile_texels = n2;
    for (uint32_t i = 0; i < (uint32_t)OCEAN_MAX_CASCADES; ++i) a.uv_scale[i] = i < count ? uv_scales[i] : 0.0f;
    CONSUMER_BEGIN(c, stream_of(c, c->last_set));
    hipLaunchKernelGGL(k_displace_grid_cascades, dim3((verts + 255) / 256), dim3(256), 0, stream_of(c, c->last_set), a);
    HIP_TRY(hipGetLastError());
    CONSUMER_END(c, stream_of(c, c->last_set));
    c->grid_vertices = verts;
    return OCEAN_OK;
}

int ocean_read_grid(ocean_t* c, float* positions, float* normals)
{
    if (!c) return OCEAN_E_INVALID;
    if (!c->grid_vertices) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    if (positions) HIP_TRY(hipMemcpy(positions, c->grid_pos, (size_t)c->grid_vertices * sizeof(float4), hipMemcpyDeviceToHost));
    if (normals) HIP_TRY(hipMemcpy(normals, c->grid_nrm, (size_t)c->grid_vertices * sizeof(float4), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}

int ocean_device_grid(ocean_t* c, void** d_positions, void** d_normals, uint32_t* vertices)
{
    if (!c) return OCEAN_E_INVALID;
    if (d_positions) *d_positions = c->grid_pos;
    if (d_normals) *d_normals = c->grid_nrm;
    if (vertices) *vertices = c->grid_vertices;
    return OCEAN_OK;
}

size_t ocean_mip_texels(uint32_t n) { return ((size_t)n * n - 1) / 3; }      // sum of (n >> l)^2, l = 1 .. log2 n

int ocean_build_mips(ocean_t* c, uint32_t tile)
{
    if (!c || tile >= c->tiles) return OCEAN_E_INVALID;
    if (!c->prepared || !c->have_frame) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    const uint32_t n = c->n;
    if (c->mips_n != n) {
        SYNC_ALL(c);
        if (c->mips_disp) (void)hipFree(c->mips_disp);
        if (c->mips_nrm) (void)hipFree(c->mips_nrm);
        c->mips_disp = c->mips_nrm = nullptr; c->mips_n = 0; c->mips_ready = false;
        HIP_TRY(hipMalloc(&c->mips_disp, ocean_mip_texels(n) * sizeof(float4)));
        HIP_TRY(hipMalloc(&c->mips_nrm, ocean_mip_texels(n) * sizeof(float4)));
        c->mips_n = n;
    }
    const size_t n2 = (size_t)n * n;
    MipArgs m;
    m.src[0] = (c->ext_disp ? c->ext_disp : c->dispN[c->last_set]) + tile * n2;
    m.src[1] = (c->ext_nrm ? c->ext_nrm : c->nrmN[c->last_set]) + tile * n2;
    m.dst[0] = c->mips_disp; m.dst[1] = c->mips_nrm;
    hipStream_t st = stream_of(c, c->last_set);            // ordered after the frame that wrote these maps
    CONSUMER_BEGIN(c, st);
    for (uint32_t w = n / 2; w >= 1; w /= 2) {
        m.w = (int)w;
        hipLaunchKernelGGL(k_mip_level, dim3((w * w + 255) / 256, 2), dim3(256), 0, st, m);
        m.src[0] = m.dst[0]; m.src[1] = m.dst[1];
        m.dst[0] += (size_t)w * w; m.dst[1] += (size_t)w * w;
    }
    HIP_TRY(hipGetLastError());
    CONSUMER_END(c, st);
    c->mips_ready = true;
    return OCEAN_OK;
}

int ocean_read_mips(ocean_t* c, float* disp_mips, float* nrm_mips)
{
    if (!c) return OCEAN_E_INVALID;
    if (!c->mips_ready) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    const size_t bytes = ocean_mip_texels(c->mips_n) * sizeof(float4);
    if (disp_mips) HIP_TRY(hipMemcpy(disp_mips, c->mips_disp, bytes, hipMemcpyDeviceToHost));
    if (nrm_mips) HIP_TRY(hipMemcpy(nrm_mips, c->mips_nrm, bytes, hipMemcpyDeviceToHost));
    return OCEAN_OK;
}

int ocean_device_mips(ocean_t* c, void** d_disp_mips, void** d_nrm_mips, uint32_t* levels)
{
    if (!c) return OCEAN_E_INVALID;
    if (d_disp_mips) *d_disp_mips = c->mips_ready ? c->mips_disp : nullptr;
    if (d_nrm_mips) *d_nrm_mips = c->mips_ready ? c->mips_nrm : nullptr;
    if (levels) { uint32_t l = 0; for (uint32_t w = c->mips_n; c->mips_ready && w > 1; w /= 2) ++l; *levels = l; }
    return OCEAN_OK;
}

void* ocean_stream(ocean_t* c) { return c ? (void*)stream_of(c, c->last_set) : nullptr; }

int ocean_set_stream(ocean_t* c, void* s)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    c->user = (hipStream_t)s;          // (last_set stays: the most recent frame's maps and records remain the ones read out)
    return OCEAN_OK;
}

int ocean_read_spectrum(ocean_t* c, uint32_t tile, float* h0, float* omega)
{
    if (!c || tile >= c->tiles) return OCEAN_E_INVALID;
    if (!c->prepared) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    SYNC_ALL(c);
    // device storage is transposed ([kx index][kz index]); hand back the reference's row-major [m][n]
    const size_t n = c->n;
    if (h0) {
        std::vector<float2> tmp(n2);
        HIP_TRY(hipMemcpy(tmp.data(), c->h0 + tile * n2, n2 * sizeof(float2), hipMemcpyDeviceToHost));
        float2* out = reinterpret_cast<float2*>(h0);
        for (size_t q = 0; q < n; ++q)
            for (size_t m = 0; m < n; ++m) out[m * n + q] = tmp[q * n + m];
    }
    if (omega) {
        std::vector<float> tmp(n2);
        HIP_TRY(hipMemcpy(tmp.data(), c->omega + tile * n2, n2 * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t q = 0; q < n; ++q)
            for (size_t m = 0; m < n; ++m) omega[m * n + q] = tmp[q * n + m];
    }
    return OCEAN_OK;
}

int ocean_read_xi(ocean_t* c, uint32_t tile, float* xi)
{
    if (!c || tile >= c->tiles || !xi) return OCEAN_E_INVALID;
    if (!c->prepared || !c->xi) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n2 = (size_t)c->n * c->n;
    SYNC_ALL(c);
    HIP_TRY(hipMemcpy(xi, c->xi + tile * n2, n2 * sizeof(float2), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}

int ocean_set_dispersion(ocean_t* c, int kind, float param)
{
    if (!c || kind < OCEAN_DISPERSION_DEEP || kind > OCEAN_DISPERSION_CAPILLARY) return OCEAN_E_INVALID;
    if (kind != OCEAN_DISPERSION_DEEP && !(param > 0.0f)) return OCEAN_E_INVALID;
    c->dispersion = kind;
    c->dispersion_param = param;
    return OCEAN_OK;
}

int ocean_set_mode(ocean_t* c, int mode)
{
    if (!c || mode < OCEAN_MODE_FULL7 || mode > OCEAN_MODE_JACOBIAN) return OCEAN_E_INVALID;
    c->mode = mode;                 // takes effect at the next frame, like SetLambda
    return OCEAN_OK;
}

int ocean_set_spectrum_precision(ocean_t* c, int bits)
{
    if (!c || (bits != 16 && bits != 32)) return OCEAN_E_INVALID;
    if (bits != c->h0_bits) c->prepared = false;     // the copy is built by ocean_prepare
    c->h0_bits = bits;
    return OCEAN_OK;
}

int ocean_set_intermediate_precision(ocean_t* c, int bits)
{
    if (!c || (bits != 16 && bits != 32)) return OCEAN_E_INVALID;
    if (bits != c->inter_bits) c->prepared = false;   // the scales are chosen by ocean_prepare
    c->inter_bits = bits;
    return OCEAN_OK;
}

int ocean_set_pipeline_depth(ocean_t* c, int depth)
{
    if (!c || depth < 1 || depth > MAXD) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    c->depth = depth;
    return OCEAN_OK;
}

int ocean_set_merged_xpass(ocean_t* c, int on)
{
    if (!c) return OCEAN_E_INVALID;
    c->merged_x = on != 0;
    return OCEAN_OK;
}

int ocean_set_start_ramp(ocean_t* c, int on)
{
    if (!c) return OCEAN_E_INVALID;
    c->start_ramp = on != 0;
    return OCEAN_OK;
}

int ocean_select_streams(ocean_t* c, uint32_t frames, float* us_per_frame)
{
    // The context's eight streams sit on the process's four hardware queues in turn, and the queues are not alike: every kernel differs by up to
    // 1 us between them (profiles/r03_bimodal_probe.txt section 4a).
    // Nothing tells a stream's queue but timing: serial frames on the first four streams (one per queue), then the streams are re-ordered,
    // fastest first -- the serial path and chain 0 use the fastest queue, chains 1..3 the next ones.  The maps hold a calibration frame afterwards.
    if (!c || frames == 0) return OCEAN_E_INVALID;
    if (!c->prepared) return OCEAN_E_NOT_READY;
    if (c->user) return OCEAN_E_UNSUPPORTED;               // a caller-owned stream: nothing to choose
    // the calibration frames go to whatever output is bound: never into memory the caller or a renderer owns
    if (c->ext_disp || c->ext_nrm) return OCEAN_E_UNSUPPORTED;
    // (assumes the context's first four streams sit on four distinct hardware queues, which holds when they are the process's first
    //  streams under the default GPU_MAX_HW_QUEUES = 4; other streams created earlier -- a framework's -- shift the mapping, and the
    //  ranking then compares whatever queues the four streams did land on: still a valid order of the context's own streams)
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    constexpr int CAND = 4;
    float us[CAND];
    int rc;
    for (int k = 0; k < CAND; ++k) {
        std::swap(c->own[0], c->own[k]);                   // candidate k runs the serial path (chain 0's buffers)
        rc = OCEAN_OK;
        for (int j = 0; j < 5 && !rc; ++j) rc = enqueue_frame(c, 0.25f * (float)j, false, nullptr);
        hipError_t e = rc ? hipSuccess : hipStreamSynchronize(c->own[0]);
        if (!rc && e == hipSuccess) e = hipEventRecord(c->start_ev, c->own[0]);
        for (uint32_t j = 0; j < frames && !rc && e == hipSuccess; ++j) rc = enqueue_frame(c, 0.25f * (float)j, false, nullptr);
        if (!rc && e == hipSuccess) e = hipEventRecord(c->end_ev[0], c->own[0]);
        if (!rc && e == hipSuccess) e = hipEventSynchronize(c->end_ev[0]);
        float ms = 0.f;
        if (!rc && e == hipSuccess) e = hipEventElapsedTime(&ms, c->start_ev, c->end_ev[0]);
        std::swap(c->own[0], c->own[k]);
        if (rc) return rc;
        if (e != hipSuccess) { g_last_hip = (int)e; return OCEAN_E_HIP; }
        us[k] = ms * 1000.0f / (float)frames;
    }
    int order[CAND] = {0, 1, 2, 3};
    std::sort(order, order + CAND, [&](int a, int b) { return us[a] < us[b]; });
    hipStream_t sorted[CAND];
    for (int k = 0; k < CAND; ++k) sorted[k] = c->own[order[k]];
    for (int k = 0; k < CAND; ++k) {
        c->own[k] = sorted[k];
        if (us_per_frame) us_per_frame[k] = us[order[k]];
    }
    c->have_frame = false;                                 // the maps hold a calibration frame: nothing to read out until the next frame
    c->last_set = 0;
    return OCEAN_OK;
}

int ocean_time_frames(ocean_t* c, float t0, float dt, int warmup, int frames, float* ms_total, float* ms_kernel)
{
    if (!c || frames <= 0 || warmup < 0) return OCEAN_E_INVALID;
    if (!c->prepared) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    for (int j = 0; j < warmup; ++j)
        if ((rc = enqueue_frame(c, t0 + dt * (float)j, true, nullptr))) return rc;
    SYNC_ALL(c);
    HIP_TRY(hipEventRecord(c->start_ev, stream_of(c, 0)));
    HIP_TRY(hipEventSynchronize(c->start_ev));
    for (int j = 0; j < frames; ++j)
        if ((rc = enqueue_frame(c, t0 + dt * (float)(warmup + j), true, nullptr))) return rc;
    // the two chains are independent: the timed region ends when the later one does
    const int nstreams = c->user ? 1 : MAXD;
    for (int i = 0; i < nstreams; ++i) HIP_TRY(hipEventRecord(c->end_ev[i], stream_of(c, i)));
    SYNC_ALL(c);
    float ms = 0.f;
    for (int i = 0; i < nstreams; ++i) {
        float m = 0.f;
        HIP_TRY(hipEventElapsedTime(&m, c->start_ev, c->end_ev[i]));
        if (m > ms) ms = m;
    }
    if (ms_total) *ms_total = ms;
    if (ms_kernel) {
        // second pass over the same frames with events around every launch, in the SAME
        // regime as the timed region: at depth D the chains keep running ahead (the host
        // only waits for a chain's previous frame before reusing its events), so the
        // durations include whatever the concurrent frames cost each other.  Event
        // overhead is why this is a separate pass, never mixed into ms_total.
        const bool pipe = c->depth > 1 && !c->user && !c->ext_disp && !c->ext_nrm;
        const int nsets = pipe ? c->depth : 1;
        double acc[3] = {0, 0, 0};
        long counted = 0;
        auto collect = [&](int set) -> int {       // (every frame of the pass has the same launches: c->launch_count / launch_kernel)
            HIP_TRY(hipEventSynchronize(c->mark_ev[set][2 * c->launch_count - 1]));
            for (int l = 0; l < c->launch_count; ++l) {
                float m = 0.f;
                HIP_TRY(hipEventElapsedTime(&m, c->mark_ev[set][2 * l], c->mark_ev[set][2 * l + 1]));
                acc[c->launch_kernel[l]] += m;      // a kernel that runs twice per frame (split order) reports the sum
            }
            ++counted;
            return OCEAN_OK;
        };
        if (!pipe) {
            // serial frames: one frame at a time (least event-queueing overhead, ~2 us per launch)
            for (int j = 0; j < frames; ++j) {
                if ((rc = enqueue_frame(c, t0 + dt * (float)(warmup + j), true, c->mark_ev[0]))) return rc;
                if ((rc = collect(0))) return rc;
            }
        } else {
            for (int j = 0; j < frames; ++j) {
                const int set = (int)(c->frame_ctr % (uint64_t)c->depth);
                if (j >= nsets && (rc = collect(set))) return rc;
                if ((rc = enqueue_frame(c, t0 + dt * (float)(warmup + j), true, c->mark_ev[set]))) return rc;
            }
            // the frames still in flight are the last min(frames, nsets) ones, oldest first
            const int outstanding = frames < nsets ? frames : nsets;
            for (int k = 0; k < outstanding; ++k) {
                const int set = (int)((c->frame_ctr - (uint64_t)outstanding + (uint64_t)k) % (uint64_t)c->depth);
                if ((rc = collect(set))) return rc;
            }
        }
        SYNC_ALL(c);
        for (int k = 0; k < 3; ++k) ms_kernel[k] = (float)(acc[k] / (double)(counted > 0 ? counted : 1));
    }
    return OCEAN_OK;
}

#ifdef OCEAN_XB_TRACE
// diagnostic build only (tools/archive/xb_trace.py): enable = allocate the trace buffer (the next frames' k_xpass_b fill it); host_out = copy it out
extern "C" int ocean_debug_xb_trace(ocean_t* c, int enable, unsigned long long* host_out, size_t count)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    if (enable && !c->stamps) {
        HIP_TRY(hipMalloc(&c->stamps, (size_t)1 << 20));
        HIP_TRY(hipMemset(c->stamps, 0, (size_t)1 << 20));
    }
    if (host_out) HIP_TRY(hipMemcpy(host_out, c->stamps, count * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}
#endif
#ifdef OCEAN_CLOCKPROBE
// diagnostic build only (tools/slow_window.py): enable = allocate the probe buffer (the next single-transform z passes fill it, record
// [frame_seq % 4096][workgroup][4]); host_out = copy `count` 64-bit words out, starting at word `first`
extern "C" int ocean_debug_clockprobe(ocean_t* c, int enable, unsigned long long* host_out, size_t first, size_t count)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    const size_t bytes = (size_t)ocean::CLOCKPROBE_LAUNCHES * ocean::CLOCKPROBE_WGS * 4 * sizeof(unsigned long long);
    if (enable && !c->stamps) {
        HIP_TRY(hipMalloc(&c->stamps, bytes));
        HIP_TRY(hipMemset(c->stamps, 0, bytes));
    }
    if (host_out) {
        if (!c->stamps || (first + count) * sizeof(unsigned long long) > bytes) return OCEAN_E_INVALID;
        HIP_TRY(hipMemcpy(host_out, c->stamps + first, count * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
    return OCEAN_OK;
}
#endif
#ifdef OCEAN_STAMPS
// diagnostic build only: per-workgroup clock stamps of the last frame
int ocean_debug_stamps(ocean_t* c, int enable, unsigned long long* host_out, size_t count)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    if (enable && !c->stamps) {
        HIP_TRY(hipMalloc(&c->stamps, (size_t)1 << 24));
        HIP_TRY(hipMemset(c->stamps, 0, (size_t)1 << 24));
    }
    if (host_out) HIP_TRY(hipMemcpy(host_out, c->stamps, count * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return OCEAN_OK;
}
#endif

int ocean_last_launch(const ocean_t* c, int idx, ocean_launch_info* out)
{
    if (!c || !out || idx < 0 || idx > 2) return OCEAN_E_INVALID;
    if (!c->have_frame) return OCEAN_E_NOT_READY;
    *out = c->last_launch[idx];
    return OCEAN_OK;
}

#ifdef OCEAN_DEVELOPER
// developer builds only: device addresses of chain `set`'s buffers (placement probes, tools/xb_placement.py)
int ocean_debug_buffers(ocean_t* c, int set, void** out /* [8]: h0, omega_q, z, zh, hraw, minmax, disp, nrm */)
{
    if (!c || !out || set < 0 || set >= MAXD) return OCEAN_E_INVALID;
    void* v[8] = {c->h0, c->omega_q, c->z[set], c->zh[set], c->hraw[set], c->minmax[set], c->dispN[set], c->nrmN[set]};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return OCEAN_OK;
}
#endif

const char* ocean_kernel_name(const ocean_t* c, int idx)
{
    return c ? kernel_name_of(idx) : nullptr;
}

int ocean_algorithmic_bytes_per_launch(const ocean_t* c, int idx)
{
    // the split of ocean_algorithmic_bytes_per_texel over the three launches (fp32 FULL7: 23 / 28 / 22):
    //   z pass       spectrum 8 (4 as half2) + dispersion 1 (2 as fp32) in, 14 of intermediates out (7 as half2; 16 / 8 with pair 3)
    //   x pass, b    pairs 1, 2 and the height plane 10 in (5; pair 3: 12 / 6), raw height 2 + normal map 16 out (+ 4: the Jacobian's two planes)
    //   x pass, disp pair 0 and the raw height 4 + 2 in (2 + 2; + 4 of the two planes), displacement map 16 out
    if (idx < 0 || idx > 2) return 0;
    int b[3];
    ocean_launch_bytes_per_texel(c && c->inter_bits == 16, c && c->mode == OCEAN_MODE_JACOBIAN, c && c->h0_bits == 16, !(c && c->prepared && !c->omega16), b);
    return b[idx];
}

int ocean_algorithmic_bytes_per_texel(const ocean_t* c)
{
    // what THIS pipeline has to move per texel in the seven-field fp32 mode (ocean_kernels.h):
    // 8 (h0) + 1 (16-bit dispersion of the columns 0..N/2 only: a column and its point mirror share it; 2 when the fp32
    // array is needed) + 14 + 14 (half-size intermediates out and in) + 2 + 2 (raw height) + 32 (maps).  SURVEY.md 8d's
    // model of a plain 3.5-transform two-pass scheme is 108.
    if (!c) return 73;
    return 73 + (c->prepared && !c->omega16 ? 1 : 0) - (c->h0_bits == 16 ? 4 : 0) - (c->inter_bits == 16 ? 14 : 0) + (c->mode == OCEAN_MODE_JACOBIAN ? (c->inter_bits == 16 ? 10 : 12) : 0);
}

}  // extern "C"

// ---------------------------------------------------------------------------------
// Packed-map gather over RCCL (the north-star's one exchange step).  Tiles are independent, so
// synthesis needs no collective; the only communication is every rank sending its finished maps to a root.
// librccl is loaded on first use (dlopen) so that single-GPU users of the library carry no dependency on it;
// inside a PyTorch process the already-loaded librccl.so.1 is the one found.
namespace {
struct RcclApi {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    bool ok = false;
};
RcclApi& rccl()
{
    static RcclApi api = [] {
        RcclApi a;
        const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so", nullptr};
        for (int i = 0; names[i] && !a.so; ++i) a.so = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (!a.so) return a;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.so, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.so, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.so, "ncclCommDestroy");
        a.GroupStart = (decltype(a.GroupStart))dlsym(a.so, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.so, "ncclGroupEnd");
        a.Gather = (decltype(a.Gather))dlsym(a.so, "ncclGather");
        a.CommCount = (decltype(a.CommCount))dlsym(a.so, "ncclCommCount");
        a.CommUserRank = (decltype(a.CommUserRank))dlsym(a.so, "ncclCommUserRank");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.GroupStart && a.GroupEnd && a.Gather;
        return a;
    }();
    return api;
}
thread_local int g_last_rccl = 0;
}  // namespace
#define RCCL_TRY(expr)                                  \
    do {                                                \
        ncclResult_t r_ = (expr);                       \
        if (r_ != ncclSuccess) {                        \
            g_last_rccl = (int)r_;                      \
            return OCEAN_E_COMM;                        \
        }                                               \
    } while (0)

static void comm_release(ocean_ctx* c)
{
    if (c->comm) { (void)rccl().CommDestroy(c->comm); c->comm = nullptr; }
    if (c->comm_stream) { (void)hipStreamDestroy(c->comm_stream); c->comm_stream = nullptr; }
    for (auto& e : c->frame_done) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (auto& e : c->gather_done) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (bool& p : c->gather_pending) p = false;
    c->comm_ranks = 0; c->comm_rank = -1;
}

extern "C" {

int ocean_last_rccl_error(void) { return g_last_rccl; }

int ocean_comm_unique_id(void* id_out)
{
    if (!id_out) return OCEAN_E_INVALID;
    static_assert(sizeof(ncclUniqueId) == OCEAN_COMM_ID_BYTES, "ncclUniqueId size");
    if (!rccl().ok) return OCEAN_E_COMM;
    ncclUniqueId id;
    RCCL_TRY(rccl().GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return OCEAN_OK;
}

int ocean_comm_init(ocean_t* c, int nranks, int rank, const void* id_in)
{
    if (!c || !id_in || nranks < 1 || rank < 0 || rank >= nranks) return OCEAN_E_INVALID;
    if (!rccl().ok) return OCEAN_E_COMM;
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    comm_release(c);
    ncclUniqueId id;
    std::memcpy(&id, id_in, sizeof(id));
    RCCL_TRY(rccl().CommInitRank(&c->comm, nranks, id, rank));      // collective: every rank of the job calls it
    c->comm_ranks = nranks; c->comm_rank = rank;
    HIP_TRY(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    for (auto& e : c->frame_done) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : c->gather_done) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return OCEAN_OK;
}

int ocean_comm_count(const ocean_t* c, int* ranks, int* rank)
{
    // what the communicator itself reports (ncclCommCount / ncclCommUserRank), not what ocean_comm_init was told
    if (!c) return OCEAN_E_INVALID;
    if (!c->comm) return OCEAN_E_NOT_READY;
    if (!rccl().CommCount || !rccl().CommUserRank) return OCEAN_E_COMM;
    int n = 0, r = -1;
    RCCL_TRY(rccl().CommCount(c->comm, &n));
    RCCL_TRY(rccl().CommUserRank(c->comm, &r));
    if (ranks) *ranks = n;
    if (rank) *rank = r;
    return OCEAN_OK;
}

int ocean_comm_destroy(ocean_t* c)
{
    if (!c) return OCEAN_E_INVALID;
    HIP_TRY(hipSetDevice(c->device));
    SYNC_ALL(c);
    comm_release(c);
    return OCEAN_OK;
}

static int gather_impl(ocean_ctx* c, int root, void* d_recv_disp, void* d_recv_nrm, bool half)
{
    if (!c) return OCEAN_E_INVALID;
    if (!c->comm) return OCEAN_E_NOT_READY;
    if (root < 0 || root >= c->comm_ranks) return OCEAN_E_INVALID;
    if (c->comm_rank == root && (!d_recv_disp || !d_recv_nrm)) return OCEAN_E_INVALID;
    if (!c->prepared || !c->have_frame) return OCEAN_E_NOT_READY;
    HIP_TRY(hipSetDevice(c->device));
    { int rc_ = check_fault(c); if (rc_) return rc_; }                        // (as in consumer_begin)
    const int set = c->last_set;
    const size_t texels = (size_t)c->tiles * c->n * c->n;
    const size_t count = texels * 4;                                          // elements per map array per rank
    const void* d = c->ext_disp ? c->ext_disp : c->dispN[set];
    const void* q = c->ext_nrm ? c->ext_nrm : c->nrmN[set];
    if (half) {     // convert on the frame's own stream, behind the frame: 16 instead of 32 bytes per texel on the wire
        for (auto& p : c->pack_half[set]) if (!p) HIP_TRY(hipMalloc(&p, texels * sizeof(uint2)));
        const unsigned blocks = (unsigned)((texels + 255) / 256 < 65535 ? (texels + 255) / 256 : 65535);
        hipLaunchKernelGGL(k_pack_half, dim3(blocks), dim3(256), 0, stream_of(c, set), (const float4*)d, c->pack_half[set][0], texels);
        hipLaunchKernelGGL(k_pack_half, dim3(blocks), dim3(256), 0, stream_of(c, set), (const float4*)q, c->pack_half[set][1], texels);
        HIP_TRY(hipGetLastError());
        d = c->pack_half[set][0]; q = c->pack_half[set][1];
    }
    // order the gather behind the frame that wrote these maps, on the communication stream: the chains keep
    // synthesising meanwhile (at depth >= 2 the next frames write other map sets), and this chain's next frame
    // waits for gather_done before it rewrites the maps (enqueue_frame)
    HIP_TRY(hipEventRecord(c->frame_done[set], stream_of(c, set)));
    HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->frame_done[set], 0));
    const ncclDataType_t ty = half ? ncclHalf : ncclFloat;
    RCCL_TRY(rccl().GroupStart());
    ncclResult_t r1 = rccl().Gather(d, d_recv_disp, count, ty, root, c->comm, c->comm_stream);   // zero-copy from the map buffers
    ncclResult_t r2 = rccl().Gather(q, d_recv_nrm, count, ty, root, c->comm, c->comm_stream);
    RCCL_TRY(rccl().GroupEnd());
    RCCL_TRY(r1); RCCL_TRY(r2);
    HIP_TRY(hipEventRecord(c->gather_done[set], c->comm_stream));
    c->gather_pending[set] = true;
    return OCEAN_OK;
}

int ocean_gather_maps(ocean_t* c, int root, void* d_recv_disp, void* d_recv_nrm) { return gather_impl(c, root, d_recv_disp, d_recv_nrm, false); }
int ocean_gather_maps_f16(ocean_t* c, int root, void* d_recv_disp, void* d_recv_nrm) { return gather_impl(c, root, d_recv_disp, d_recv_nrm, true); }

}  // extern "C"
