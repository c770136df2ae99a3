/*
 * oracle/ocean_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's Tessendorf ocean synthesis hot path
 * (class WSTessendorf, /root/reference/src/scene/WSTessendorf.{h,cpp}).
 * It is the checker for the HIP path (tests/, __graft_entry__.smoke()) and the
 * timed CPU baseline of bench.py (cpu_baseline.kind = "port").  Nothing in the
 * product path (watersurfacerendering_amd/, include/) may call into this file.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures
 * for this path, and the reference translation unit cannot be
 * built here: it needs FFTW 3.3.10 (fetched from the network by
 * CMakeLists.txt:157-179, not installed on this image) and glm (empty
 * submodule libs/glm).  The arithmetic outside the FFT follows the reference
 * source line by line (citations below); the DFT is pinned by definition
 * against a naive float64 DFT and scipy's pocketfft in tests/test_oracle.py.
 *
 * Every function cites the reference lines it restates.  Built with
 * -ffp-contract=off: the reference's default x86-64 Release build has no FMA.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

#define REAL float
#define SUF _f
#include "fft_impl.inc"
#undef REAL
#undef SUF
#define REAL double
#define SUF _d
#include "fft_impl.inc"
#undef REAL
#undef SUF

#define ORACLE_MODE_FULL7   0   /* all seven fields (reference behaviour)           */
#define ORACLE_MODE_CHOPPY5 1   /* h, Dx, Dz, slope-x, slope-z; normal.zw = 0        */
#define ORACLE_MODE_HEIGHT1 2   /* height only; disp.xz = 0, normal = 0              */
#define ORACLE_MODE_JACOBIAN 3  /* the seven fields + the reference's COMPUTE_JACOBIAN intent (.h:209-224, .cpp:330-335,
                                 * 368-378, 421-428): two more transforms dz(Dx), dx(Dz) and displacement.w = the Jacobian
                                 * of the horizontal displacement.  The reference's block does not compile (it names
                                 * `displacement` outside its scope, .cpp:427); the arithmetic and its order are its own. */

#define ORACLE_FFT_F32 0        /* reference shape: float FFTs, one thread per 2-D transform */
#define ORACLE_FFT_F64 1        /* float front end, FFT + pack in double              */
#define ORACLE_FFT_F32_TEAM 2   /* float FFTs work-shared by every thread (not the reference's shape: the strong CPU baseline) */
#define ORACLE_FFT_EXTERNAL 4    /* stage D done by a callback (a library FFT on every core): strong CPU baseline, not the reference's shape */
#define ORACLE_FFT_FFTW 3       /* libfftw3f loaded at run time, the reference's plans (WSTessendorf.cpp:191-232); only if present */

/* FFTW 3 (float) entry points, resolved with dlopen when the host has the library.  The reference links
 * FFTW 3.3.10 (CMakeLists.txt:157-179); nothing on the build image provides it, so this path is taken only
 * on a host that does.  Constants from FFTW's public API: FFTW_BACKWARD = +1, FFTW_MEASURE = 0. */
#include <dlfcn.h>
typedef void* (*fftwf_plan_dft_2d_fn)(int, int, void*, void*, int, unsigned);
typedef void (*fftwf_execute_fn)(void*);
typedef void (*fftwf_destroy_plan_fn)(void*);
static struct { int tried; void* so; fftwf_plan_dft_2d_fn plan; fftwf_execute_fn exec; fftwf_destroy_plan_fn destroy; } g_fftw;
int oracle_fftw_available(void)
{
    if (!g_fftw.tried) {
        g_fftw.tried = 1;
        const char* names[] = { "libfftw3f.so.3", "libfftw3f.so", NULL };
        for (int i = 0; names[i] && !g_fftw.so; ++i) g_fftw.so = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (g_fftw.so) {
            g_fftw.plan = (fftwf_plan_dft_2d_fn)dlsym(g_fftw.so, "fftwf_plan_dft_2d");
            g_fftw.exec = (fftwf_execute_fn)dlsym(g_fftw.so, "fftwf_execute");
            g_fftw.destroy = (fftwf_destroy_plan_fn)dlsym(g_fftw.so, "fftwf_destroy_plan");
            if (!g_fftw.plan || !g_fftw.exec || !g_fftw.destroy) { dlclose(g_fftw.so); g_fftw.so = NULL; }
        }
    }
    return g_fftw.so != NULL;
}

int oracle_num_threads(void);
typedef void (*oracle_fft_cb)(float* fields, int nfields, int n, void* user);
#include <time.h>
static double wall_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e3 + (double)ts.tv_nsec * 1e-6;
}

typedef struct oracle_ctx {
    /* properties: WSTessendorf.h:181-199 */
    uint32_t n;
    float length;
    float wind_x, wind_y;    /* unit vector */
    float wind_speed;
    float phillips_a;
    float damping;
    float anim_period;
    float base_freq;
    float lambda;
    int dispersion;          /* 0 deep water (the only one the reference calls), 1 finite depth, 2 capillary */
    float dispersion_param;  /* depth D, or wavelength L */
    float min_height, max_height;
    int prepared;
    /* Prepare() products: WSTessendorf.h:211-215 */
    float* kvec;             /* n*n*2 (kx, kz)      */
    float* kunit;            /* n*n*2               */
    float* h0;               /* n*n*2 heightAmp     */
    float* h0c;              /* n*n*2 heightAmp_conj*/
    float* omega;            /* n*n   dispersion    */
    float* xi;               /* n*n*2 gaussian draws (re, im) */
    /* outputs */
    float* disp;             /* n*n*4 */
    float* nrm;              /* n*n*4 */
    /* FFT state */
    plan1d_f pf;
    plan1d_d pd;
    cpx_f* ff;               /* 9*n*n, one block like WSTessendorf.cpp:158-174 (kTotalInputs = 7 + 2 with COMPUTE_JACOBIAN) */
    cpx_d* fd;               /* 7*n*n, only for ORACLE_FFT_F64 */
    cpx_f* work_f;           /* 7 * 17n */
    cpx_d* work_d;
    cpx_f* work_team;        /* work_team_threads * 17n, ORACLE_FFT_F32_TEAM */
    int work_team_threads;
    void* fftw_plans[9];     /* ORACLE_FFT_FFTW: one in-place plan per field, like WSTessendorf.cpp:191-232 */
    oracle_fft_cb ext_fft;   /* ORACLE_FFT_EXTERNAL */
    void* ext_fft_user;
    double stage_ms[4];      /* wall time of the last frame's stages: A-C spectra, D transforms, E-F pack, G normalise */
} oracle_ctx;

/* ------------------------------------------------------------------------ */
/* setters: WSTessendorf.cpp:459-505                                         */

int oracle_set_tile_size(oracle_ctx* c, uint32_t n)
{   /* :459-468 -- non power of two is ignored (assert compiled out) */
    if (n == 0 || (n & (n - 1))) return -1;
    if (n != c->n) c->prepared = 0;
    c->n = n;
    return 0;
}
void oracle_set_tile_length(oracle_ctx* c, float l) { c->length = l; c->prepared = 0; }     /* :470-474 */
void oracle_set_wind_direction(oracle_ctx* c, float x, float y)
{   /* :476-479, glm::normalize(w) = w * (1 / sqrt(dot(w, w))) */
    const float d = x * x + y * y;
    const float inv = 1.0f / sqrtf(d);
    c->wind_x = x * inv; c->wind_y = y * inv; c->prepared = 0;
}
void oracle_set_wind_speed(oracle_ctx* c, float v)
{   /* :481-484 */
    c->wind_speed = v > 0.0001f ? v : 0.0001f; c->prepared = 0;
}
void oracle_set_animation_period(oracle_ctx* c, float t)
{   /* :486-490: 2.0f * M_PI / T evaluated in double, stored as float */
    c->anim_period = t;
    c->base_freq = (float)((double)2.0f * M_PI / (double)t);
    c->prepared = 0;
}
void oracle_set_phillips_const(oracle_ctx* c, float a) { c->phillips_a = a; c->prepared = 0; } /* :492-495 */
void oracle_set_lambda(oracle_ctx* c, float l) { c->lambda = l; }                             /* :497-500 */
void oracle_set_damping(oracle_ctx* c, float d) { c->damping = d; c->prepared = 0; }         /* :502-505 */

/* The two dispersion relations WSTessendorf.h defines but never calls:
 * kind 1 = DispersionTransWaves(k, D) (.h:301-304), kind 2 = DispersionSmallWaves(k, L) (.h:312-315);
 * kind 0 = DispersionDeepWaves (.h:290-293, what QDispersion uses).  The quantisation of .h:284-287 is
 * applied to all of them so that the animation stays periodic. */
void oracle_set_dispersion(oracle_ctx* c, int kind, float param)
{
    c->dispersion = kind; c->dispersion_param = param; c->prepared = 0;
}
static float dispersion_of(const oracle_ctx* c, float k)
{
    if (c->dispersion == 1)      /* sqrt(g k tanh(k D)): tanhf is libm-dependent, so evaluated in double, rounded once */
        return (float)sqrt((double)(9.81f * k) * tanh((double)k * (double)c->dispersion_param));
    if (c->dispersion == 2) {    /* sqrt(g k (1 + k^2 L^2)), float, left to right like the reference expression */
        const float l = c->dispersion_param;
        return sqrtf(9.81f * k * (1.0f + k * k * l * l));
    }
    return sqrtf(9.81f * k);
}

uint32_t oracle_tile_size(const oracle_ctx* c) { return c->n; }
float oracle_min_height(const oracle_ctx* c) { return c->min_height; }
float oracle_max_height(const oracle_ctx* c) { return c->max_height; }
float oracle_base_freq(const oracle_ctx* c) { return c->base_freq; }
const float* oracle_displacements(const oracle_ctx* c) { return c->disp; }
const float* oracle_normals(const oracle_ctx* c) { return c->nrm; }
const float* oracle_h0(const oracle_ctx* c) { return c->h0; }
const float* oracle_h0_conj(const oracle_ctx* c) { return c->h0c; }
const float* oracle_omega(const oracle_ctx* c) { return c->omega; }
const float* oracle_kvec(const oracle_ctx* c) { return c->kvec; }
const float* oracle_kunit(const oracle_ctx* c) { return c->kunit; }
const float* oracle_xi(const oracle_ctx* c) { return c->xi; }
void oracle_wind(const oracle_ctx* c, float* out2) { out2[0] = c->wind_x; out2[1] = c->wind_y; }

static void free_buffers(oracle_ctx* c)
{
    free(c->kvec); free(c->kunit); free(c->h0); free(c->h0c); free(c->omega); free(c->xi);
    free(c->disp); free(c->nrm); free(c->ff); free(c->fd); free(c->work_f); free(c->work_d); free(c->work_team);
    c->work_team = NULL;
    for (int f = 0; f < 9; ++f)
        if (c->fftw_plans[f]) { g_fftw.destroy(c->fftw_plans[f]); c->fftw_plans[f] = NULL; }
    c->kvec = c->kunit = c->h0 = c->h0c = c->omega = c->xi = c->disp = c->nrm = NULL;
    c->ff = NULL; c->fd = NULL; c->work_f = NULL; c->work_d = NULL;
    plan1d_free_f(&c->pf); plan1d_free_d(&c->pd);
}

/* constructor + defaults: WSTessendorf.cpp:13-26, WSTessendorf.h:36-43,181 */
oracle_ctx* oracle_create(uint32_t n, float length)
{
    oracle_ctx* c = (oracle_ctx*)calloc(1, sizeof(*c));
    if (!c) return NULL;
    c->base_freq = 1.0f;
    c->lambda = -1.0f;
    c->min_height = -1.0f; c->max_height = 1.0f;
    c->n = 512;
    oracle_set_tile_size(c, n);
    oracle_set_tile_length(c, length);
    oracle_set_wind_direction(c, 1.0f, 1.0f);
    oracle_set_wind_speed(c, 30.0f);
    oracle_set_animation_period(c, 200.0f);
    oracle_set_phillips_const(c, 3e-7f);
    oracle_set_damping(c, 0.1f);
    return c;
}

void oracle_destroy(oracle_ctx* c)
{
    if (!c) return;
    free_buffers(c);
    free(c);
}

/* ------------------------------------------------------------------------ */
/* Deterministic replacement for glm::gaussRand (WSTessendorf.cpp:98-99).
 * The reference seeds std::rand() from the clock (core/Application.cpp:21),
 * so no particular draw is reproducible; any i.i.d. N(0,1) pair per texel is a
 * faithful input.  Counter-based: splitmix64(seed, texel) -> Box-Muller in
 * double.  The HIP library implements the same generator
 * (csrc/ocean_kernels.hip, k_init_spectrum).                                */
static inline uint64_t splitmix64(uint64_t seed, uint64_t idx)
{
    uint64_t z = seed + (idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void oracle_gauss_pair(uint64_t seed, uint64_t idx, float* re, float* im)
{
    const uint64_t z = splitmix64(seed, idx);
    const double u1 = ((double)(z >> 40) + 1.0) * (1.0 / 16777216.0);        /* (0,1] */
    const double u2 = (double)((z >> 8) & 0xFFFFFFull) * (1.0 / 16777216.0); /* [0,1) */
    const double r = sqrt(-2.0 * log(u1));
    const double a = 2.0 * M_PI * u2;
    *re = (float)(r * cos(a));
    *im = (float)(r * sin(a));
}

/* ------------------------------------------------------------------------ */
/* PhillipsSpectrum: WSTessendorf.h:249-263 (all fp32, left-to-right)        */
static float phillips(const oracle_ctx* c, float ux, float uy, float k)
{
    const float k2 = k * k;
    const float k4 = k2 * k2;
    float cf = ux * c->wind_x + uy * c->wind_y;
    cf = cf * cf;
    const float lw = c->wind_speed * c->wind_speed / 9.81f;
    const float l2 = lw * lw;
    return c->phillips_a * expf(-1.0f / (k2 * l2)) / k4 * cf * expf(-k2 * c->damping * c->damping);
}

/* Prepare(): WSTessendorf.cpp:36-58.  xi_or_null: n*n*2 floats (re, im) row
 * major to inject the gaussian draws, else they come from (seed, texel).    */
int oracle_prepare(oracle_ctx* c, uint64_t seed, const float* xi_or_null)
{
    const uint32_t n = c->n;
    const size_t n2 = (size_t)n * n;
    free_buffers(c);
    c->kvec = (float*)malloc(n2 * 2 * sizeof(float));
    c->kunit = (float*)malloc(n2 * 2 * sizeof(float));
    c->h0 = (float*)malloc(n2 * 2 * sizeof(float));
    c->h0c = (float*)malloc(n2 * 2 * sizeof(float));
    c->omega = (float*)malloc(n2 * sizeof(float));
    c->xi = (float*)malloc(n2 * 2 * sizeof(float));
    c->disp = (float*)malloc(n2 * 4 * sizeof(float));
    c->nrm = (float*)malloc(n2 * 4 * sizeof(float));
    c->ff = (cpx_f*)malloc(n2 * 9 * sizeof(cpx_f));
    c->work_f = (cpx_f*)malloc((size_t)9 * 17 * n * sizeof(cpx_f));
    if (!c->kvec || !c->kunit || !c->h0 || !c->h0c || !c->omega || !c->xi || !c->disp || !c->nrm ||
        !c->ff || !c->work_f)
        return -2;
    if (plan1d_init_f(&c->pf, (int)n) || plan1d_init_d(&c->pd, (int)n)) return -3;

    /* ComputeWaveVectors: .cpp:60-85; WaveVector ctor: .h:133-136.
     * M_PI * (2.0f*n - kSize) / kLength: float numerator, double product and
     * quotient, narrowed to float by the glm::vec2 constructor. */
    const float fsize = (float)(int32_t)n;
    for (uint32_t m = 0; m < n; ++m)
        for (uint32_t q = 0; q < n; ++q) {
            const size_t i = (size_t)m * n + q;
            const float kx = (float)(M_PI * (double)(2.0f * (float)(int32_t)q - fsize) / (double)c->length);
            const float kz = (float)(M_PI * (double)(2.0f * (float)(int32_t)m - fsize) / (double)c->length);
            c->kvec[2 * i] = kx; c->kvec[2 * i + 1] = kz;
            const float d = kx * kx + kz * kz;
            const float len = sqrtf(d);
            if (len > 0.00001f) {
                const float inv = 1.0f / sqrtf(d);
                c->kunit[2 * i] = kx * inv; c->kunit[2 * i + 1] = kz * inv;
            } else {
                c->kunit[2 * i] = 0.0f; c->kunit[2 * i + 1] = 0.0f;
            }
        }

    /* ComputeGaussRandomArray: .cpp:87-103 (RNG replaced, see above) */
    if (xi_or_null) memcpy(c->xi, xi_or_null, n2 * 2 * sizeof(float));
    else
        for (size_t i = 0; i < n2; ++i) oracle_gauss_pair(seed, i, &c->xi[2 * i], &c->xi[2 * i + 1]);

    /* ComputeBaseWaveHeightField: .cpp:105-148; BaseWaveHeightFT .h:237-243;
     * QDispersion/DispersionDeepWaves .h:284-297 */
    const float inv_sqrt2 = 1.0f / sqrtf(2.0f);   /* s_kOneOver2sqrt, .h:231 */
    #pragma omp parallel for schedule(guided)
    for (size_t i = 0; i < n2; ++i) {
        const float kx = c->kvec[2 * i], kz = c->kvec[2 * i + 1];
        const float k = sqrtf(kx * kx + kz * kz);
        if (k > 0.00001f) {
            const float ux = c->kunit[2 * i], uz = c->kunit[2 * i + 1];
            const float gr = c->xi[2 * i], gi = c->xi[2 * i + 1];
            const float sp = sqrtf(phillips(c, ux, uz, k));
            const float sm = sqrtf(phillips(c, -ux, -uz, k));
            /* (s * xi) * sqrt(P): scalar-complex then complex-scalar products */
            c->h0[2 * i] = (inv_sqrt2 * gr) * sp;
            c->h0[2 * i + 1] = (inv_sqrt2 * gi) * sp;
            c->h0c[2 * i] = (inv_sqrt2 * gr) * sm;
            c->h0c[2 * i + 1] = -((inv_sqrt2 * gi) * sm);
            c->omega[i] = floorf(dispersion_of(c, k) / c->base_freq) * c->base_freq;
        } else {
            c->h0[2 * i] = c->h0[2 * i + 1] = 0.0f;
            c->h0c[2 * i] = 0.0f; c->h0c[2 * i + 1] = -0.0f;
            c->omega[i] = 0.0f;
        }
    }
    /* output defaults: .cpp:48-54 */
    for (size_t i = 0; i < n2; ++i) {
        c->disp[4 * i] = c->disp[4 * i + 1] = c->disp[4 * i + 2] = c->disp[4 * i + 3] = 0.0f;
        c->nrm[4 * i] = 0.0f; c->nrm[4 * i + 1] = 1.0f; c->nrm[4 * i + 2] = c->nrm[4 * i + 3] = 0.0f;
    }
    c->prepared = 1;
    return 0;
}

/* complex product as libstdc++ std::complex<float> does it: (ac-bd, ad+bc) */
static inline cpx_f cmul(cpx_f a, cpx_f b)
{
    cpx_f r; r.re = a.re * b.re - a.im * b.im; r.im = a.re * b.im + a.im * b.re; return r;
}

/* ComputeWaves(t): WSTessendorf.cpp:284-441 + NormalizeHeights :443-455.
 * Returns the amplitude A (and fills disp / nrm).                           */
float oracle_compute_waves(oracle_ctx* c, float t, int mode, int fft_kind)
{
    if (!c->prepared) return NAN;
    const uint32_t n = c->n;
    const size_t n2 = (size_t)n * n;
    cpx_f* height = c->ff;
    cpx_f* slope_x = height + n2;
    cpx_f* slope_z = slope_x + n2;
    cpx_f* d_x = slope_z + n2;
    cpx_f* d_z = d_x + n2;
    cpx_f* dxd_x = d_z + n2;
    cpx_f* dzd_z = dxd_x + n2;
    cpx_f* dzd_x = dzd_z + n2;       /* .cpp:171-174 */
    cpx_f* dxd_z = dzd_x + n2;
    const int nfields = mode == ORACLE_MODE_JACOBIAN ? 9 : (mode == ORACLE_MODE_FULL7 ? 7 : (mode == ORACLE_MODE_CHOPPY5 ? 5 : 1));
    if (fft_kind == ORACLE_FFT_F64 && !c->fd) {
        c->fd = (cpx_d*)malloc(n2 * 9 * sizeof(cpx_d));
        c->work_d = (cpx_d*)malloc((size_t)9 * 17 * n * sizeof(cpx_d));
        if (!c->fd || !c->work_d) return NAN;
    }

    if (fft_kind == ORACLE_FFT_F32_TEAM && (!c->work_team || c->work_team_threads < oracle_num_threads())) {
        free(c->work_team);                  /* the team size may have grown since (oracle_set_num_threads) */
        c->work_team_threads = oracle_num_threads();
        c->work_team = (cpx_f*)malloc((size_t)c->work_team_threads * 17 * n * sizeof(cpx_f));
        if (!c->work_team) return NAN;
    }
    if (fft_kind == ORACLE_FFT_FFTW) {
        if (!oracle_fftw_available()) return NAN;
        for (int f = 0; f < 9; ++f)      /* SetupFFTW, .cpp:191-246: in place, FFTW_BACKWARD, FFTW_MEASURE (planning clobbers the arrays: done before they are filled) */
            if (!c->fftw_plans[f]) c->fftw_plans[f] = g_fftw.plan((int)n, (int)n, c->ff + (size_t)f * n2, c->ff + (size_t)f * n2, +1, 0u);
    }

    const double t_begin = wall_ms();
    /* One `omp parallel` region in the reference (.cpp:292-438); three here, split at the two points where the
     * reference's team meets at a barrier anyway (before and after the transforms), so that stage D can also be
     * handed to an external library FFT (ORACLE_FFT_EXTERNAL) without an idle team spinning beside it. */
    #pragma omp parallel
    {
        /* A :294-300, WaveHeightFT .h:265-275 */
        #pragma omp for schedule(static)
        for (size_t i = 0; i < n2; ++i) {
            const float wt = c->omega[i] * t;
            const float pc = cosf(wt), ps = sinf(wt);
            cpx_f a = { c->h0[2 * i], c->h0[2 * i + 1] };
            cpx_f b = { c->h0c[2 * i], c->h0c[2 * i + 1] };
            cpx_f e1 = { pc, ps }, e2 = { pc, -ps };
            cpx_f p = cmul(a, e1), q = cmul(b, e2);
            height[i].re = p.re + q.re; height[i].im = p.im + q.im;
        }
        if (nfields >= 5) {
            /* B :303-312 */
            #pragma omp for schedule(static) nowait
            for (size_t i = 0; i < n2; ++i) {
                cpx_f ikx = { 0.0f, c->kvec[2 * i] }, ikz = { 0.0f, c->kvec[2 * i + 1] };
                slope_x[i] = cmul(ikx, height[i]);
                slope_z[i] = cmul(ikz, height[i]);
            }
            /* C :315-336 */
            #pragma omp for schedule(static)
            for (size_t i = 0; i < n2; ++i) {
                cpx_f mux = { 0.0f, -c->kunit[2 * i] }, muz = { 0.0f, -c->kunit[2 * i + 1] };
                d_x[i] = cmul(mux, height[i]);
                d_z[i] = cmul(muz, height[i]);
                if (nfields >= 7) {
                    cpx_f ikx = { 0.0f, c->kvec[2 * i] }, ikz = { 0.0f, c->kvec[2 * i + 1] };
                    dxd_x[i] = cmul(ikx, d_x[i]);
                    dzd_z[i] = cmul(ikz, d_z[i]);
                    if (nfields == 9) {          /* .cpp:330-335 */
                        dzd_x[i] = cmul(ikz, d_x[i]);
                        dxd_z[i] = cmul(ikx, d_z[i]);
                    }
                }
            }
        }
    }
    const double t_spectra = wall_ms();
    /* D :338-367 -- one single-threaded 2-D backward DFT per field */
    if (fft_kind == ORACLE_FFT_EXTERNAL) {
        if (!c->ext_fft) return NAN;
        c->ext_fft((float*)c->ff, nfields, (int)n, c->ext_fft_user);     /* in place, unnormalised, backward */
    } else {
        #pragma omp parallel
        {
            if (fft_kind == ORACLE_FFT_F32) {
                #pragma omp for schedule(dynamic, 1)
                for (int f = 0; f < nfields; ++f)
                    fft2d_f(&c->pf, c->ff + (size_t)f * n2, c->work_f + (size_t)f * 17 * n);
            } else if (fft_kind == ORACLE_FFT_F32_TEAM) {
                fft2d_team_f(&c->pf, c->ff, nfields, c->work_team + (size_t)omp_get_thread_num() * 17 * n);
            } else if (fft_kind == ORACLE_FFT_FFTW) {
                #pragma omp for schedule(dynamic, 1)     /* omp sections of .cpp:338-367: one fftwf_execute per field */
                for (int f = 0; f < nfields; ++f) g_fftw.exec(c->fftw_plans[f]);
            } else {
                #pragma omp for schedule(dynamic, 1)
                for (int f = 0; f < nfields; ++f) {
                    cpx_d* dst = c->fd + (size_t)f * n2;
                    const cpx_f* src = c->ff + (size_t)f * n2;
                    for (size_t i = 0; i < n2; ++i) { dst[i].re = src[i].re; dst[i].im = src[i].im; }
                    fft2d_d(&c->pd, dst, c->work_d + (size_t)f * 17 * n);
                    cpx_f* back = c->ff + (size_t)f * n2;
                    for (size_t i = 0; i < n2; ++i) { back[i].re = (float)dst[i].re; back[i].im = (float)dst[i].im; }
                }
            }
        }
    }
    const double t_fft = wall_ms();
    /* :289-290 -- max starts at numeric_limits<float>::min() (= FLT_MIN > 0) */
    float master_max = FLT_MIN;
    float master_min = FLT_MAX;
    const float lambda = c->lambda;
    #pragma omp parallel
    {
        /* E :380-412 */
        float tmax = FLT_MIN, tmin = FLT_MAX;
        #pragma omp for schedule(static) nowait
        for (uint32_t m = 0; m < n; ++m)
            for (uint32_t q = 0; q < n; ++q) {
                const size_t i = (size_t)m * n + q;
                const int sgn = ((q + m) & 1) ? -1 : 1;
                const float hft = height[i].re * (float)sgn;
                tmax = hft > tmax ? hft : tmax;
                tmin = hft < tmin ? hft : tmin;
                float* d = c->disp + 4 * i;
                d[1] = hft;
                if (nfields >= 5) {
                    d[0] = (float)sgn * lambda * d_x[i].re;
                    d[2] = (float)sgn * lambda * d_z[i].re;
                } else { d[0] = 0.0f; d[2] = 0.0f; }
                d[3] = 1.0f;
            }
        #pragma omp critical
        {
            master_max = tmax > master_max ? tmax : master_max;
            master_min = tmin < master_min ? tmin : master_min;
        }
        /* F :414-437 */
        #pragma omp for schedule(static) nowait
        for (uint32_t m = 0; m < n; ++m)
            for (uint32_t q = 0; q < n; ++q) {
                const size_t i = (size_t)m * n + q;
                const int sgn = ((q + m) & 1) ? -1 : 1;
                float* o = c->nrm + 4 * i;
                o[0] = nfields >= 5 ? (float)sgn * slope_x[i].re : 0.0f;
                o[1] = nfields >= 5 ? (float)sgn * slope_z[i].re : 0.0f;
                o[2] = nfields >= 7 ? (float)sgn * dxd_x[i].re : 0.0f;
                o[3] = nfields >= 7 ? (float)sgn * dzd_z[i].re : 0.0f;
                if (nfields == 9) {              /* .cpp:421-428, written to displacement.w as the shaders read it (.vert:29, .frag:210-212) */
                    const float jacobian =
                        (1.0f + lambda * (float)sgn * dxd_x[i].re) *
                        (1.0f + lambda * (float)sgn * dzd_z[i].re) -
                        (lambda * (float)sgn * dxd_z[i].re) *
                        (lambda * (float)sgn * dzd_x[i].re);
                    c->disp[4 * i + 3] = jacobian;
                }
            }
    }
    const double t_pack = wall_ms();
    /* G NormalizeHeights :443-455 -- serial in the reference */
    c->min_height = master_min;
    c->max_height = master_max;
    const float amp = fmaxf(fabsf(master_min), fabsf(master_max));
    const float inv = 1.f / amp;
    if (fft_kind == ORACLE_FFT_F32_TEAM || fft_kind == ORACLE_FFT_EXTERNAL) {     /* the strong baselines also share this loop out (the reference's is serial) */
        #pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n2; ++i) c->disp[4 * i + 1] *= inv;
    } else {
        for (size_t i = 0; i < n2; ++i) c->disp[4 * i + 1] *= inv;
    }
    const double t_end = wall_ms();
    c->stage_ms[0] = t_spectra - t_begin; c->stage_ms[1] = t_fft - t_spectra;
    c->stage_ms[2] = t_pack - t_fft; c->stage_ms[3] = t_end - t_pack;
    return amp;
}

/* Stage D through a caller-supplied transform (bench.py's cpu_baseline_strong hands it scipy's pocketfft on every
 * core): cb(fields, nfields, n, user) must replace each of the nfields consecutive n x n complex-float arrays by
 * its unnormalised backward 2-D DFT, in place. */
void oracle_set_external_fft(oracle_ctx* c, oracle_fft_cb cb, void* user) { c->ext_fft = cb; c->ext_fft_user = user; }

/* Wall time (ms) of the stages of the last oracle_compute_waves: [0] spectra A-C, [1] transforms D, [2] pack E-F,
 * [3] normalise G (bench.py reports the split beside the CPU baseline). */
const double* oracle_stage_ms(const oracle_ctx* c) { return c->stage_ms; }

/* Raw (un-normalised, un-signed) complex FFT outputs of the last call, for
 * intermediate checks: field f in [0,7): h, sx, sz, Dx, Dz, dxDx, dzDz.     */
const float* oracle_field(const oracle_ctx* c, int f) { return (const float*)(c->ff + (size_t)f * c->n * c->n); }

/* Team size of the following frames (bench.py looks for the fastest one for its strong CPU baseline: a container
 * may expose more logical CPUs than its quota lets run at once). */
void oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Plain 1-D / 2-D transforms exported for the DFT pinning tests. */
int oracle_fft2d_f32(int n, float* data /* n*n*2, in place */)
{
    plan1d_f p; if (plan1d_init_f(&p, n)) return -1;
    cpx_f* w = (cpx_f*)malloc((size_t)17 * n * sizeof(cpx_f));
    fft2d_f(&p, (cpx_f*)data, w);
    free(w); plan1d_free_f(&p); return 0;
}
int oracle_fft2d_f64(int n, double* data)
{
    plan1d_d p; if (plan1d_init_d(&p, n)) return -1;
    cpx_d* w = (cpx_d*)malloc((size_t)17 * n * sizeof(cpx_d));
    fft2d_d(&p, (cpx_d*)data, w);
    free(w); plan1d_free_d(&p); return 0;
}
