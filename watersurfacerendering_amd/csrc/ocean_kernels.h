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
//      (height, dDx/dx, dDz/dz even; Dx, Dz, slopes odd).  Only spectrum columns 0..N/2
//      go through the first (z-axis) pass and only map rows 0..N/2 through the second
//      (x-axis) pass; the other half is the mirror image.  (Checked bit-exactly against
//      the oracle: tests/test_oracle.py::test_reference_output_point_symmetry.)
//
//   k_zpass      one spectrum column nb (kx index) in [0, N/2] per workgroup: animate
//                columns nb and -nb (contiguous runs of the transposed spectrum; an element and
//                its point mirror share their phase, hence one sincos and one dispersion read),
//                S+ = (a+b)/2, S- = (a-b)/2, build the three packed pairs and the height
//                from S+/S- and the wave-vector coefficients, four z-axis inverse FFTs.
//   k_xpass_b    x-axis pass, part 1, one launch, two kinds of workgroup: HEIGHT (two
//                real rows per complex transform, sign, raw heights out, global
//                min/max by atomics) and NORMAL (pairs 1 and 2 -> finished normal-map
//                rows q and N-q).
//   k_xpass_disp part 2, needs the min/max: pair 0 + raw height -> displacement-map rows.
//   The x axis goes last so that every map row is written as whole contiguous lines.
//
// HBM bytes per texel actually moved (this pipeline): 9 (h0 8, dispersion as a 16-bit multiple
// of the base frequency, read for half of the columns: 1) + 14 + 14 (half-size intermediates out and in)
// + 2 + 2 (raw height) + 32 (maps) = 73, against 108 for the straightforward 3.5-transform two-pass scheme
// SURVEY.md section 8d models.  59 with half2 intermediates (Z16 kernels), 85 in the Jacobian mode (JAC kernels:
// the height plane becomes pair 3 = (height, cross derivative), and two more half-size real planes
// carry the Jacobian's factors to the displacement pass).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <type_traits>
#include "fft_engine.h"

// Diagnostic builds (never shipped; `make -C csrc variant NAME=x DEFS=...`): -DOCEAN_STAMPS records
// per-workgroup phase clocks (tools/archive/stamps.py); -DOCEAN_ABL_NOLOAD / _NOSTORE / _NOMAPSTORE / _NOFFT /
// _NOIN / _SINCOS remove the global loads, the intermediate stores, the map stores, the butterfly
// arithmetic, the z-pass input arithmetic or the sincos -- results are then wrong on purpose; they
// only attribute time.
namespace ocean {

typedef float ocean_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt(float4* p, float4 v)
{
    ocean_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<ocean_f4*>(p));
}
// Map stores.  The maps are written once and never read back by this pipeline, while the
// intermediates (Z, hraw) and the spectrum are re-read every frame and fit the 256 MiB
// memory-side cache: when several frames are in flight a non-temporal map store keeps the
// output stream from evicting that resident set (2048^2, depth 3: 69 -> 54 us per frame).  A lone
// serial frame prefers plain stores (the cache then buffers the write burst: +4 % with
// non-temporal), so the x-pass kernels exist in both forms (template flag NTS) and the host
// picks per launch.  The texel index is turned into a 32-bit byte offset (N <= 4096: < 2^28)
// so the store addresses as scalar base + vector offset.
// Write-through (`sc1`) stores: the line leaves the XCD's L2 as it is written.  Adopted for the fp32 intermediates up to 2048 (store_z, WT);
// measured and rejected for the maps (developer A/B: -DOCEAN_MAP_SC1=1 stores the maps of the plain form that way, =2 those of the
// non-temporal form as well: k_xpass_disp 16.0 -> 17.6-18.2 us at 2048^2, the pipelined and 4096^2 frames +40 %; profiles/r05_store_policy_experiments.txt).
__device__ __forceinline__ void store_f4_sc1(float4* base, unsigned byte_off, float4 v)
{
    const ocean_f4 t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(byte_off), "v"(t), "s"(base) : "memory");
}
__device__ __forceinline__ void store_f2_sc1(float2* base, unsigned byte_off, float2 v)
{
    typedef float f2v __attribute__((ext_vector_type(2)));
    const f2v t = {v.x, v.y};
    asm volatile("global_store_dwordx2 %0, %1, %2 sc1" ::"v"(byte_off), "v"(t), "s"(base) : "memory");
}
template <bool NTS> __device__ __forceinline__ void store_map(float4* base, unsigned texel, float4 v)
{
#ifdef OCEAN_ABL_NOMAPSTORE      // ablation build: the map texels are computed but never written
    asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
#else
#ifdef OCEAN_MAP_SC1
    if constexpr (!NTS || OCEAN_MAP_SC1 == 2) { store_f4_sc1(base, texel * 16u, v); return; }
#endif
    float4* p = reinterpret_cast<float4*>(reinterpret_cast<char*>(base) + (texel * 16u));
    if constexpr (NTS) store_nt(p, v);
    else *p = v;
#endif
}
#define OCEAN_STORE(base, texel, val) store_map<NTS>((base), (unsigned)(texel), (val))
// The same texel to a SECOND destination as well: the caller's page-locked host array, through its device address (ocean_compute_waves_read,
// small maps: FrameArgs::disp_host / nrm_host; null -- a wave-uniform branch -- in every other frame).  The x passes then write the maps
// across PCIe as they produce them, 1 KiB per wave instruction like the device copy, instead of a copy behind the frame: a blocking 512^2
// call 191 -> 17x us (profiles/r06_dropin_call.txt).  The device copy stays complete: every other read-out sees the frame as before.
__device__ __forceinline__ void store_map_host(float4* host_base, unsigned texel, float4 v)
{
    if (host_base) *reinterpret_cast<float4*>(reinterpret_cast<char*>(host_base) + (texel * 16u)) = v;
}

// Element `idx` of a per-tile array through a 32-bit BYTE offset (every per-tile array here is far
// below 4 GiB): the access then addresses as scalar base + 32-bit vector offset instead of a 64-bit
// vector address per access -- one VGPR instead of two per outstanding load and no 64-bit adds.
template <class T> __device__ __forceinline__ const T& at32(const T* base, unsigned idx)
{
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + idx * (unsigned)sizeof(T));
}
// Intermediate (z-pass output) store.  ZNT = the context's intermediates do not fit the memory-side
// cache (4096^2, or large batches of tiles with several frames in flight): written non-temporally they
// at least leave the cache to the spectrum, which is re-read every frame (4096^2: -4 %, 8 x 1024^2 at
// depth 2: -5 %); where they do fit, a plain store keeps them on chip for the x pass (2048^2: 54 vs 60 us).
// Z16 = reduced-precision intermediates (ocean_set_intermediate_precision(16)): the z-pass outputs are stored as
// half2, multiplied first by a per-tile power of two `scale` chosen at ocean_prepare from a time-independent bound
// of the column sums so that nothing can overflow (k_inter_bounds); the same arrays, half the bytes (14 -> 7 B/texel
// out of the z pass and into the x pass).  Stated accuracy of that mode: tests/test_parity_gpu.py.
// WT = WRITE-THROUGH (round 5, profiles/r05_store_policy_experiments.txt): the fp32 intermediates of the plain form are stored `sc1` -- they leave
// the XCD's L2 as they are written instead of sitting there dirty until the end-of-kernel release writes everything back in one burst behind
// the last workgroup; the x pass that reads them runs on whatever XCD its rows land on, so seven reads in eight come through the fabric anyway.
// 2048^2 serial z pass 21.8-21.9 -> 20.4-20.7 us (x passes unchanged), 512^2 6.5-7.0 -> 6.2-6.6, 8 x 1024^2 36.9-37.2 -> 35.6-36.8; at 4096^2
// -- several rounds of workgroups, 235 MB -- the same stores cost 6-8 us (90-95 -> 99-100): not there.
// A SERIAL frame only: pipelined 2048^2 frames lose 0.7 us per frame with it (47.4 -> 48.1 us, eight interleaved repeats: beside other
// chains' launches the L2 is the better write buffer) -- so it is a store policy of its own, instantiated for the single-transform z pass at
// the sizes where it pays (zpass_has_wt<N>()) and picked per launch (ocean_launch.h).
template <int N> constexpr bool zpass_has_wt() { return N == 2048 || N == 1024; }
template <bool ZNT, bool Z16 = false, bool WT = false> __device__ __forceinline__ void store_z(float2* base, unsigned idx, float2 v, float scale = 1.0f, float scale_y = 0.0f)
{
    if constexpr (Z16) {
        const __half2 h = __floats2half2_rn(v.x * scale, v.y * (scale_y != 0.0f ? scale_y : scale));
        unsigned bits;
        __builtin_memcpy(&bits, &h, 4);
        unsigned* p = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(base) + idx * 4u);
        if constexpr (ZNT) __builtin_nontemporal_store(bits, p);
        else *p = bits;
    } else if constexpr (ZNT) {
        typedef float f2nt __attribute__((ext_vector_type(2)));
        const f2nt t = {v.x, v.y};
        __builtin_nontemporal_store(t, reinterpret_cast<f2nt*>(reinterpret_cast<char*>(base) + idx * 8u));
    } else if constexpr (WT) {
        store_f2_sc1(base, idx * 8u, v);
    } else {
        *reinterpret_cast<float2*>(reinterpret_cast<char*>(base) + idx * 8u) = v;
    }
}
// element idx of an intermediate array as float2 (times `unscale` in the half2 form)
template <bool Z16> __device__ __forceinline__ float2 load_z(const float2* base, unsigned idx, float unscale, float unscale_y = 0.0f)
{
    if constexpr (Z16) {
        const unsigned bits = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(base) + idx * 4u);
        __half2 h;
        __builtin_memcpy(&h, &bits, 4);
        const float2 f = __half22float2(h);
        return make_float2(f.x * unscale, f.y * (unscale_y != 0.0f ? unscale_y : unscale));
    } else {
#ifdef OCEAN_XLOAD_NT      // developer A/B: the x passes read the intermediates, each element once, with non-temporal loads
        typedef float f2nt __attribute__((ext_vector_type(2)));
        const f2nt t = __builtin_nontemporal_load(reinterpret_cast<const f2nt*>(reinterpret_cast<const char*>(base) + idx * 8u));
        return make_float2(t.x, t.y);
#else
        return *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(base) + idx * 8u);
#endif
    }
}
template <class T> __device__ __forceinline__ T& at32(T* base, unsigned idx)
{
    return *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + idx * (unsigned)sizeof(T));
}

struct TileParams {          // device copy of one tile's properties
    float wind_x, wind_y;    // unit vector (SetWindDirection, .cpp:476-479)
    float wind_speed;        // (.cpp:481-484)
    float phillips_a;        // (.cpp:492-495)
    float damping;           // (.cpp:502-505)
    float base_freq;         // 2 pi / T as float (.cpp:486-490)
    float length;            // tile length L
    int dispersion;          // 0 deep water (.h:290-293), 1 finite depth D (.h:301-304), 2 capillary L (.h:312-315)
    float dispersion_param;
    float pad_;
    uint64_t seed;
};

struct FrameArgs {
    const float2* h0;        // [tiles][N][N]   base amplitudes h0(k), TRANSPOSED: [n (kx index)][m (kz index)]
    const float* omega;      // [tiles][N][N]   quantised dispersion, same layout
    const uint16_t* omega_q; // [tiles][N][N]   the same as the integer multiple of base_freq (null: use omega); see k_zpass
    const float* base_freq;  // [tiles]
    const __half2* h0h;      // [tiles][N][N]   optional fp16 copy of h0 scaled by 1/h0_inv_scale[tile] (null = fp32)
    const float* h0_inv_scale;   // [tiles]
    const float* k1d;        // [tiles][N]      k(i) = float(pi*(2i-N)/L)
    const float2* tw;        // [N]             exp(+2 pi i k / N)
    float2* z;               // [tiles][3][NUP/ZB][2][N/2+1][ZB] z-transformed pairs 0..2: element (column nb,
                             //   side, row q) at Half<N>::zidx -- side 0 = rows q = 0..N/2, side 1 = rows (N-q)%N; NUP = N/2 + 16 rounded
                             //   down to a multiple of 16; ZB = 8 rows per block (16 in the half2 form)
    float2* z3;              // [tiles][NUP/ZB][2][N/2+1][ZB]   pair 3 = (height, cross derivative), same layout: OCEAN_MODE_JACOBIAN only
                             //   (allocated with jraw / jac0 by the first frame of that mode, null before)
    float2* zh;              // [tiles][NUP/ZB][N/2+1][ZB]   z-transformed height, rows 0..N/2 (Half<N>::zhidx)
    const float4* zscale;    // [tiles][2] powers of two of the half2 intermediates (Z16 kernels): [0] = (s_u, s_k, 1/s_u, 1/s_k) for the pairs
                             //   weighted by unit vectors / by k; [1] = (s_3, g, 1/s_3, 1/g) for pair 3 of the Jacobian mode, whose
                             //   cross-derivative part is first multiplied by g so that both parts have the height's magnitude
    float* hraw;             // [tiles][NUP][N]          signed raw height of map rows 0..N/2 (+ padding rows)
    float* jraw;             // [tiles][NUP][N]          OCEAN_MODE_JACOBIAN: signed d(Dx)/dz = d(Dz)/dx of the same rows
    float* jac0;             // [tiles][NUP][N]          OCEAN_MODE_JACOBIAN: (1 + lambda dDx/dx)(1 + lambda dDz/dz) of the same rows
    unsigned* minmax;        // [tiles][2]       ordered-int keys of min, max
    unsigned* zdone;         // [tiles]          z-pass workgroups that have finished, counted up frame after frame (never reset): what the x-axis
                             //                  workgroups of a ONE-LAUNCH frame (k_frame) wait for -- until it has reached zdone_target
    unsigned zdone_target;   //                  (N/2 + 1) x the number of one-launch frames this chain has run, this one included (mod 2^32)
    unsigned* fault;         // [1] host-coherent: set to 1 by an in-launch wait that gave up after 20 ms (the frame is then wrong: the host reports it)
    int poll_sleep;          //                  in-launch waits: s_sleep 127 (about 3.4 us of a 2.4 GHz clock / 64) repeated this many times between two polls of a counter (0: s_sleep 2)
    unsigned* hdone;         // [tiles]          HEIGHT workgroups of the frame that have finished (reset by the z pass): what the DISP workgroups
                             //                  of a merged x pass wait for (k_xpass_b, xb_roles bit 2)
    uint4* done_rec;         // [tiles]          host-coherent completion records (min key, max key, frame_seq, 0), written by the
                             //                  displacement pass's last workgroup (frame_done)
    unsigned* done_ctr;      // [1]              workgroups of the displacement pass that have finished (the last one resets it)
    unsigned frame_seq;      // sequence number of this frame on its chain (never 0)
    float4* disp;            // [tiles][N][N]
    float4* nrm;             // [tiles][N][N]
    const float* toff;       // [tiles] or null
    const float* lambda;     // [tiles], or null: every tile uses lambda_all
    float lambda_all;
    float t;
    int start_ramp;          // > 0: the workgroups of this launch start spread over start_ramp x 10 ns, in dispatch order (start_ramp_wait;
                             // set per launch by ocean_launch.h, 0 in ocean_api.hip's arguments)
    int mode;                // 0 FULL7 (reference), 1 CHOPPY5 (dDx/dx = dDz/dz = 0), 2 HEIGHT1 (height only),
                             // 3 JACOBIAN (FULL7 + the cross derivative; displacement.w = Jacobian of the horizontal displacement)
    // ---- set per launch by ocean_launch.h (ocean_api.hip leaves the defaults) ----
    int zmask;               // z pass: the transforms this launch runs -- bit 0 pair 0, bit 1 pair 1, bit 2 pair 2, bit 3 the height (pair 3 in the
                             // Jacobian mode).  15 = all (one z pass per frame); the split frame order (launch_frame) runs {height, pair 0} and
                             // {pair 1, pair 2} as two launches, each animating the spectrum for itself
    int xb_roles;            // k_xpass_b: bit 0 the HEIGHT workgroups, bit 1 the NORMAL workgroups (3 = both in one launch), bit 2 the DISP
                             // workgroups as well (7 = the merged x pass: the whole x axis in one launch, no k_xpass_disp)
    int xcd_rot;             // developer builds only (OCEAN_XCD_ROT, tools/xcd_rot.py): the single-transform z pass hands the column groups of XCDs 1..7 round
                             // by this many places -- which XCD writes which part of the intermediates -- 0 in the shipped library
    float4* disp_host;       // [tiles][N][N] or null: the caller's page-locked destinations of ocean_compute_waves_read (device addresses): the x passes
    float4* nrm_host;        //                        store every map texel there as well (store_map_host)
    int rec_mode;            // completion records of this launch: 0 none, 1 block 0 writes them early (untracked frame: the stream tells when it
                             // has finished), 2 the last workgroup to finish writes them (frame_done; tracked frame) -- the frame's LAST launch
};



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
// h~(k, t): WaveHeightFT (.h:265-275).  conj(h0(-k)) of the reference equals
// conj(h0(k)) (same gaussian draw, Phillips even in k: .cpp:131-135), so
//   h~ = h0 e^{i wt} + conj(h0) e^{-i wt} = 2 (Re h0 cos wt - Im h0 sin wt)  exactly real.
// omega*t is ONE fp32 multiply like the reference, then an accurate sincos (wt reaches 1e4 rad).
// ============================================================================
// sin and cos of an fp32 angle, both within ~1.5 ulp (9e-8 abs) for |x| < 1e5 rad:
// three-term Cody-Waite reduction by pi/2 carried by FMAs, then the Cephes
// single-precision minimax polynomials on [-pi/4, pi/4].  About 25 VALU
// instructions for the pair (ocml's sincosf, with its Payne-Hanek branch, costs ~3x
// that and was ~45% of the z pass).  Larger angles take the library path.
__device__ __forceinline__ void sincos_f32(float x, float& s, float& c)
{
    if (__builtin_expect(fabsf(x) >= 1.0e5f, 0)) {
        sincosf(x, &s, &c);
        return;
    }
    const float fn = rintf(x * 0.636619772f);
    float r = __builtin_fmaf(fn, -1.57079601e+00f, x);
    r = __builtin_fmaf(fn, -3.13916473e-07f, r);
    r = __builtin_fmaf(fn, -5.39030253e-15f, r);
    const float z = r * r;
    float ps = __builtin_fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
    const float sn = __builtin_fmaf(ps * z, r, r);
    float pc = __builtin_fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
    const float cs = __builtin_fmaf(pc * z, z, __builtin_fmaf(z, -0.5f, 1.0f));
    const int q = (int)fn;
    float so = (q & 1) ? cs : sn;
    float co = (q & 1) ? sn : cs;
    s = (q & 2) ? -so : so;
    c = ((q + 1) & 2) ? -co : co;
}

__device__ __forceinline__ float mul_nocontract(float a, float b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float height_re(float h0r, float h0i, float c, float s)
{
#pragma clang fp contract(off)
    const float re = h0r * c - h0i * s;
    return re + re;
}

// ---- phase 1 of the z pass: animate a spectrum column and its point mirror -----------------------------------------------
// The quantised dispersion is point-symmetric bit for bit -- omega(m, n) == omega((N-m)%N, (N-n)%N): k(N-i) = -k(i) exactly, so
// |k| and everything computed from it agree (tests/test_oracle.py checks the array) -- and the z pass always needs element
// e of column nb TOGETHER with element (N-e)%N of the mirror column (N-nb)%N (S+- = (a +- b)/2).  One sincos therefore serves
// both: half the sincos and half the dispersion reads of animating the two columns separately, and S+, S-(0) are formed in
// the registers that hold a and b (no staging of h~ in LDS, one barrier less).  Items are element PAIRS (n, n+1), n even.
// H16: the fp16 copy of the spectrum (ocean_set_spectrum_precision(16)); W16: the dispersion as 16-bit multiples of the base
// frequency (false: the fp32 array, the fallback when some multiple needs more bits).  Neither is worth a kernel instantiation
// of its own -- the fp16 spectrum measured no gain, the fallback is rare -- so one kernel holds the four forms of its phase 1
// and picks one by a wave-uniform branch on the launch arguments (spectrum_form); the branch sits OUTSIDE the load loop: with
// it inside zpass_load_pair the loads of a thread's items no longer all issued ahead of the first sincos (z pass +7 %).  The
// usual form (fp32 spectrum, 16-bit dispersion) keeps an instantiation without the other three (FAST): carrying them cost the
// 2048^2 z pass 3 % (26.1 -> 27.0 us) although they never run.
template <bool FAST, class F>
__device__ __forceinline__ void spectrum_form(const FrameArgs& a, F&& f)
{
    if constexpr (FAST) { f(std::false_type{}, std::true_type{}); return; }     // fp32 spectrum, 16-bit dispersion: the usual form, no branch at all
    if (__builtin_expect(a.h0h != nullptr, 0)) {
        if (a.omega_q) f(std::true_type{}, std::true_type{});
        else f(std::true_type{}, std::false_type{});
    } else if (__builtin_expect(a.omega_q != nullptr, 1)) f(std::false_type{}, std::true_type{});
    else f(std::false_type{}, std::false_type{});
}
// tile sizes whose spectrum is read with non-temporal loads (the usual form: fp32 spectrum, 16-bit dispersion)
#ifndef OCEAN_SPEC_NT_MIN
#define OCEAN_SPEC_NT_MIN 4096
#endif
#ifdef OCEAN_SPEC_NT_ALWAYS
template <int N, bool ZNT> constexpr bool spectrum_nt() { return N >= OCEAN_SPEC_NT_MIN; }
#else
template <int N, bool ZNT> constexpr bool spectrum_nt() { return N >= OCEAN_SPEC_NT_MIN && !ZNT; }
#endif
template <int N, bool H16, bool W16, bool ZNT = false>
__device__ __forceinline__ void zpass_load_pair(const FrameArgs& a, int tile, int col, int n, float h16s, float base,
                                                float4& ha, float2& hb0, float2& hb1, float2& w)
{
    const size_t n2 = (size_t)N * N;
    const int mcol = (N - col) & (N - 1);
    const size_t g = (size_t)col * N + n;
    const size_t m0 = (size_t)mcol * N + ((N - n) & (N - 1)), m1 = (size_t)mcol * N + (N - n - 1);
#ifdef OCEAN_ABL_NOLOAD
    ha = make_float4(1.f + g, 2.f, 3.f, 4.f); hb0 = make_float2(0.5f, 1.5f); hb1 = make_float2(2.5f, 3.5f); w = make_float2(0.5f, 0.25f);
    return;
#endif
    if constexpr (spectrum_nt<N, ZNT>() && !H16 && W16) {
        // beyond the memory-side cache (4096^2: 151 MB of spectrum, read once per z pass): streamed past it, so that the intermediates
        // -- written here, re-read by the x pass right behind -- are what stays resident (ocean_launch.h: the split frame order)
        typedef float nt4 __attribute__((ext_vector_type(4)));
        typedef float nt2 __attribute__((ext_vector_type(2)));
        const float2* __restrict__ h0 = a.h0 + tile * n2;
        const nt4 va = __builtin_nontemporal_load(reinterpret_cast<const nt4*>(h0 + g));
        const nt2 v0 = __builtin_nontemporal_load(reinterpret_cast<const nt2*>(h0 + m0)), v1 = __builtin_nontemporal_load(reinterpret_cast<const nt2*>(h0 + m1));
        ha = make_float4(va.x, va.y, va.z, va.w); hb0 = make_float2(v0.x, v0.y); hb1 = make_float2(v1.x, v1.y);
        const unsigned two = __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(a.omega_q + tile * n2 + g));
        w = make_float2(mul_nocontract((float)(two & 0xffffu), base), mul_nocontract((float)(two >> 16), base));
        return;
    }
    if constexpr (H16) {
        const __half2* __restrict__ hh = a.h0h + tile * n2;
        const float2 raw2 = *reinterpret_cast<const float2*>(hh + g);       // two half2
        const __half2 x0 = *reinterpret_cast<const __half2*>(&raw2.x), x1 = *reinterpret_cast<const __half2*>(&raw2.y);
        const float2 f0 = __half22float2(x0), f1 = __half22float2(x1), g0 = __half22float2(hh[m0]), g1 = __half22float2(hh[m1]);
        ha = make_float4(f0.x * h16s, f0.y * h16s, f1.x * h16s, f1.y * h16s);
        hb0 = make_float2(g0.x * h16s, g0.y * h16s); hb1 = make_float2(g1.x * h16s, g1.y * h16s);
    } else {
        const float2* __restrict__ h0 = a.h0 + tile * n2;
        ha = *reinterpret_cast<const float4*>(h0 + g);
        hb0 = h0[m0]; hb1 = h0[m1];
    }
    if constexpr (W16) {      // two 16-bit multiples of base_freq -> the same two floats the fp32 array holds
        const unsigned two = *reinterpret_cast<const unsigned*>(a.omega_q + tile * n2 + g);
        w = make_float2(mul_nocontract((float)(two & 0xffffu), base), mul_nocontract((float)(two >> 16), base));
    } else {
        w = *reinterpret_cast<const float2*>(a.omega + tile * n2 + g);
    }
}
// h~ of the element (h0a) and of its mirror (h0b), which share the phase w t
__device__ __forceinline__ void animate_with_mirror(float2 h0a, float2 h0b, float w, float t, float& av, float& bv)
{
    const float wt = mul_nocontract(w, t);      // ONE fp32 multiply, like the reference (.h:267)
    float s, c;
#ifdef OCEAN_ABL_SINCOS
    s = wt * 1e-4f; c = 1.0f - s;
#else
    sincos_f32(wt, s, c);
#endif
    av = height_re(h0a.x, h0a.y, c, s);
    bv = height_re(h0b.x, h0b.y, c, s);
}

// ---- half-spectrum storage geometry -------------------------------------------
#ifndef OCEAN_ZTILE
#define OCEAN_ZTILE 8          // rows per block of the intermediates (fp32: 64-byte pieces)
#endif
#ifndef OCEAN_ZTILE_H
#define OCEAN_ZTILE_H 16       // the same for half2 intermediates (64-byte pieces again)
#endif
template <int N> struct Half {
    static constexpr int NU = N / 2 + 1;          // columns (units) / rows kept: 0..N/2
    static constexpr int NUP = (N / 2 + 16) & ~15;   // rows 0..N/2 padded to a multiple of 16
    static constexpr size_t Z_GROUP = (size_t)NU * 2 * NUP;       // float2 per packed pair
    static constexpr size_t Z_TILE = 3 * Z_GROUP;                 // pairs 0..2 (pair 3 = (height, cross derivative) of OCEAN_MODE_JACOBIAN: FrameArgs::z3)
    static constexpr size_t ZH_TILE = (size_t)NU * NUP;
    static constexpr size_t HRAW_TILE = (size_t)NUP * N;          // floats
    // element (column nb, side, row q) of a packed pair's group, and (column nb, row q) of the height's half plane:
    // row-blocked, [q / ZB][side][nb][q % ZB], so that an x-pass workgroup -- which owns a few rows q and walks all columns
    // nb -- reads one dense run per side instead of a small piece out of every column's 16.5 KB run, while the z pass still
    // writes ZB units = 64 bytes at a time (OCEAN_ZTILE = 0: the column-major layout [nb][side][q] of round 1;
    // profiles/r02_layout_experiments.txt has the A/B of 4-, 8- and 16-row blocks).
    template <bool Z16> static constexpr int zb() { return OCEAN_ZTILE ? (Z16 ? OCEAN_ZTILE_H : OCEAN_ZTILE) : 8; }
    static_assert(NUP % 16 == 0, "row blocks");
    template <bool Z16> static __device__ __forceinline__ unsigned zidx(int nb, int side, int q)
    {
        constexpr int ZB = zb<Z16>();
        if constexpr (OCEAN_ZTILE != 0) return (unsigned)(((q / ZB) * 2 + side) * (NU * ZB) + nb * ZB + (q % ZB));
        else return (unsigned)((nb * 2 + side) * NUP + q);
    }
    template <bool Z16> static __device__ __forceinline__ unsigned zhidx(int nb, int q)
    {
        constexpr int ZB = zb<Z16>() > 8 ? zb<Z16>() : 8;       // the height role reads eight units per column
        if constexpr (OCEAN_ZTILE != 0) return (unsigned)((q / ZB) * (NU * ZB) + nb * ZB + (q % ZB));
        else return (unsigned)(nb * NUP + q);
    }
};
// raw height of map row u at column p: rows of N floats, so that a wave working on one row reads
// and writes 64 consecutive floats (an 8-row-interleaved layout cost the displacement pass 5 us)
__device__ __forceinline__ unsigned hraw_index(int n, int p, int u) { return (unsigned)(u * n + p); }

// XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so give the
// workgroups that share an XCD (same id % 8) consecutive column blocks: lines
// shared by neighbouring blocks (input sectors, mirrored map rows) meet in one L2.
__device__ __forceinline__ int xcd_swizzle(int id, int n)
{
    const int q = n / 8, r = n % 8, x = id % 8;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + id / 8;
}

// Where a thread's last-stage outputs of a z-axis transform go in the row-blocked intermediates (Half<N>::zidx / zhidx).  The last
// stage hands work item u, leg i the position p = j + i * S (S = N / last radix, j < S, (u, i) unrolled constants): the side is
// known per leg at compile time (but for the self-mirrored row N/2) and the index is one of three per-item bases -- computed once
// per column -- plus a compile-time constant.  The straightforward form (compare p with N/2, divide and modulo by the block
// height, per output, under divergent branches) was 45 % of the z pass's vector instructions per batch.
template <int N, int T, class P, int ZC, bool Z16> struct ZStore {
    using HF = Half<N>;
    using LS = LastStage<N, ZC, T, P>;
    static constexpr int S = LS::STRIDE, NU = HF::NU;
    static constexpr int ZB = HF::template zb<Z16>(), ZBH = ZB > 8 ? ZB : 8;
    // (not for the half2 four-transform form -- 256^2 and 512^2 with ocean_set_intermediate_precision(16): there the compiler
    //  leaves the kernel's argument block and a closure in scratch memory with this path, 344 bytes of stack)
    static constexpr bool FAST = OCEAN_ZTILE != 0 && S >= ZBH && S % ZBH == 0 && (N / 2) % S == 0 && !(Z16 && ZC == 4);
    int b0[LS::IT], b1[LS::IT], bh[LS::IT];
    // A base is read through an empty asm: otherwise a choice between two outputs' positions (c ? height : pair 2, ...) is folded
    // into ONE load with a selected address, which pins the three small arrays in scratch memory instead of registers.
    static __device__ __forceinline__ int reg(int v) { asm("" : "+v"(v)); return v; }
    // nb: the column of this workgroup; two_columns: item u's column is nb + c (the two-column z pass)
    __device__ __forceinline__ void init(int tid, int nb, bool two_columns = false)
    {
        if constexpr (FAST) {
#pragma unroll
            for (int u = 0; u < LS::IT; ++u) {
                const int w = tid + u * T;
                int c = 0, j = 0;
                if (!LS::GUARD || w < LS::ITEMS) LS::map(w, c, j);
                const int col = nb + (two_columns ? c : 0);
                b0[u] = (j / ZB) * (2 * NU * ZB) + (j % ZB) + col * ZB;                           // side 0, row j
                const int cj = (j + ZB - 1) / ZB, rj = (ZB - j % ZB) % ZB;
                b1[u] = NU * ZB + col * ZB + rj - cj * (2 * NU * ZB);                              // side 1, row -j (+ a multiple of S)
                bh[u] = (j / ZBH) * (NU * ZBH) + (j % ZBH) + col * ZBH;                            // height half plane, row j
            }
        }
    }
    // element of a packed pair's group for output position p (any p)
    __device__ __forceinline__ unsigned pos(int col, int p, int u, int i) const
    {
        if constexpr (!FAST) return p <= N / 2 ? HF::template zidx<Z16>(col, 0, p) : HF::template zidx<Z16>(col, 1, N - p);
        else {
            const int lo = i * S;
            if (lo < N / 2) return (unsigned)(reg(b0[u]) + (lo / ZB) * (2 * NU * ZB));
            const int far = reg(b1[u]) + ((N - lo) / ZB) * (2 * NU * ZB);
            if (lo > N / 2) return (unsigned)far;
            // p == N/2 <=> j == 0: the self-mirrored row lives on side 0, one side's worth (NU * ZB) before what the formula of side 1
            // gives for j == 0 (written as a correction of `far`, not as a choice between b0 and b1: the latter turns into a load
            // with a selected address and keeps the bases in scratch memory)
            return (unsigned)(far - (p == N / 2 ? NU * ZB : 0));
        }
    }
    // the height's half plane keeps rows 0 .. N/2 only
    __device__ __forceinline__ bool keeps(int p, int i) const
    {
        if constexpr (!FAST) return p <= N / 2;
        else return i * S < N / 2 || (i * S == N / 2 && p == N / 2);
    }
    __device__ __forceinline__ unsigned hpos(int col, int p, int u, int i) const
    {
        if constexpr (!FAST) return HF::template zhidx<Z16>(col, p);
        else return (unsigned)(reg(bh[u]) + ((i * S) / ZBH) * (NU * ZBH));      // (row N/2: j == 0, so bh[u] is the column's offset alone)
    }
};

// (Round 5's half-size real-input transform of the height -- built, measured, not adopted -- lives in experimental/zpass_half_height.h and is
//  compiled only into developer builds that define OCEAN_HALF_HEIGHT_MIN.)
#ifdef OCEAN_HALF_HEIGHT_MIN
#include "experimental/zpass_half_height.h"
#else
template <int N> constexpr bool zpass_half_height() { return false; }
#endif

// The four z-axis transforms of one spectrum column (see k_zpass).  COL0 = Nyquist
// column nb == 0, the only one where Tx = S- along the whole column.
// Interleaved transforms per z-pass batch: 2 (two batches: {pair 0, pair 1}, {pair 2, height}) or 4 (all of a
// column's transforms in ONE batch with twice the threads).  Four at once halves the dependent chain of a
// workgroup -- stages, barriers, LDS round trips -- which is what a single small tile waits for: 512^2 z pass
// 8.0 -> 6.9 us, 256^2 (one wave per column, whose two-transform batches fill half of it) 10.1 -> 8.4 us.
// At 1024^2 it is slower (14.9 -> 16.5 us) and from 2048 up the two-batch form keeps three
// workgroups per CU.  (The same idea for the normal-map role -- pairs 1 and 2 as one batch of 2 C columns -- was
// measured 50-60 % slower at both sizes; profiles/r02_small_tile_experiments.txt.)
#ifndef OCEAN_ZC4
#define OCEAN_ZC4 1
#endif
template <int N> constexpr int zpass_columns() { return (OCEAN_ZC4 && (N == 512 || N == 256)) ? 4 : 2; }
// Spectrum columns per z-pass workgroup.  Two NEIGHBOURING columns (each batch = the same pair of both columns) make every
// store instruction of the last stage cover whole 128-byte lines of the row-blocked intermediates (2 x 64-byte pieces side by
// side) instead of half lines -- what the non-temporal stores of big tiles and batches need.  Costs N more floats of LDS (a
// second S+ table) and doubles the workgroup's chain, so only where the workgroups per CU do not change.
// The launcher picks (k_zpass<..., ZW>): always at 4096^2 (one workgroup per CU either way: z pass 150 -> 132 us), from 1024 up
// whenever the intermediates are streamed; never for a single small tile, whose few, longer workgroups would leave the chip
// emptier (1024^2 z pass 16.0 -> 18.4 us).
template <int N> constexpr bool zpass_has_width2() { return zpass_columns<N>() == 2 && N >= 1024 && N <= 2048; }

// The first-stage input of one z-axis transform at element e of a spectrum column: pair 0 (uz Tz, -ux Tx), pair 1
// (-kz Tz, kx Tx), pair 2 (kx ux S+, kz uz S+), pair 3 (S+, g3 kx uz Tc).  ONE definition with contraction off, so that every
// z-pass variant (two or four transforms per batch, one or two columns per workgroup) feeds bit-identical values to its
// transforms: frames are bit-identical whatever variant the launcher picks (tests/test_parity_bench_regimes.py).
template <int PAIR>
__device__ __forceinline__ c32 zpass_input(float kx, float kx2, float kz, float sv, float tx, float tz, float tc, float gate, bool jac, float g3)
{
#pragma clang fp contract(off)
    const float d = __builtin_fmaf(kz, kz, kx2);
    const float inv = d > 1e-10f ? rsqrtf(d) : 0.0f;              // |k| > 1e-5 (.h:135)
    if constexpr (PAIR == 0) { const float f = inv * gate; return make_float2(kz * f * tz, -kx * f * tx); }
    else if constexpr (PAIR == 1) { const float f = -1.0f * gate; return make_float2(kz * f * tz, -kx * f * tx); }
    else if constexpr (PAIR == 2) { const float g = gate * inv * sv; return make_float2(kx2 * g, (kz * kz) * g); }
    else return make_float2(sv, jac ? g3 * (kx * kz * inv * tc) : 0.0f);
}

template <int N, int T, class P, bool COL0, bool ZNT, bool Z16, int ZC, bool ZWT = false>
__device__ __forceinline__ void zpass_transforms(const FrameArgs& a, c32* fbuf, const float* sp, const float* kzt,
                                                 TwiddleRegs<N, ZC, T, P>& twr, float kx, float sm0, int tid,
                                                 int tile, int nb, int batches /* bit 0: {pair 0, pair 1}, bit 1: {pair 2, height} */)
{
    using HF = Half<N>;
    const float kx2 = kx * kx;
    [[maybe_unused]] float su = 1.0f, sk = 1.0f;              // half2 intermediates: pair 0 and the height scale with su, pairs 1 and 2 with sk
    [[maybe_unused]] float s3 = 1.0f, g3 = 1.0f;              // pair 3 (Jacobian mode): common scale of both parts, gain of the cross derivative
    if constexpr (Z16) { const float4 zs = a.zscale[2 * tile]; su = zs.x; sk = zs.y; s3 = a.zscale[2 * tile + 1].x; }
    // the cross derivative goes in multiplied by a power of two g that brings it to the height's magnitude (in fp32 as
    // well: packed with a height a thousand times its size it would inherit the height's rounding error) and comes out
    // of the x pass divided by it
    if (a.mode == 3) g3 = a.zscale[2 * tile + 1].y;
    // element offsets; the half2 form packs the same elements at 4 bytes each from the same base address
    constexpr size_t ES = Z16 ? 4 : 8;
    float2* __restrict__ zt = reinterpret_cast<float2*>(reinterpret_cast<char*>(a.z) + (size_t)tile * HF::Z_TILE * ES);
    float2* __restrict__ z3 = reinterpret_cast<float2*>(reinterpret_cast<char*>(a.z3) + (size_t)tile * HF::Z_GROUP * ES);
    float2* __restrict__ zh = reinterpret_cast<float2*>(reinterpret_cast<char*>(a.zh) + (size_t)tile * HF::ZH_TILE * ES);
    // side 0 holds p = 0..N/2, side 1 holds N-p for p > N/2 (the self-mirrored positions 0 and N/2 exist on side 0
    // only: the x pass knows)
    ZStore<N, T, P, ZC, Z16> zo;
    zo.init(tid, nb);
    // S+(e) and Tx(e), Tz(e)
    auto fetch = [&](int e, float& sv, float& tx, float& tz) {
        if constexpr (COL0) {
            const float g1 = sp[e], g2 = sp[(N - e) & (N - 1)];
            sv = 0.5f * (g1 + g2);
            tx = 0.5f * (g1 - g2);
            tz = (e == 0) ? tx : sv;
        } else {
            sv = sp[e];
            tx = sv;
            tz = (e == 0) ? sm0 : sv;
        }
    };
    if constexpr (ZC == 4) {
        // -- one batch: slot 0 = pair 0, slot 1 = pair 1, slot 2 = pair 2, slot 3 = height (or pair 3) ----------
        const float full7 = (a.mode == 0 || a.mode == 3) ? 1.0f : 0.0f;
        const float choppy = a.mode != 2 ? 1.0f : 0.0f;           // HEIGHT1 keeps only slot 3
        const bool jac = a.mode == 3;
        auto in = [&](int e, int c, int, int) -> c32 {
            float sv, tx, tz;
            fetch(e, sv, tx, tz);
            const float kz = kzt[e];
            const float d = __builtin_fmaf(kz, kz, kx2);
            const float inv = d > 1e-10f ? rsqrtf(d) : 0.0f;      // |k| > 1e-5 (.h:135)
            if (c < 2) {
                const float f = (c ? -1.0f : inv) * choppy;        // pair 1: (-kz Tz, kx Tx); pair 0: (uz Tz, -ux Tx)
                return make_float2(kz * f * tz, -kx * f * tx);
            }
            const float g = full7 * inv * sv;
            const float tc = COL0 ? (e == 0 ? sv : tx) : tz;       // cross derivative: see the two-batch form below
            return c == 2 ? make_float2(kx2 * g, kz * kz * g) : make_float2(sv, jac ? g3 * (kx * kz * inv * tc) : 0.0f);
        };
        auto out = [&](int p, int c, c32 v, int u, int i) {
            const unsigned pos = zo.pos(nb, p, u, i);
            if (c == 3) {
                if (jac) store_z<ZNT, Z16, ZWT>(z3, pos, v, s3);
                else if (zo.keeps(p, i)) store_z<ZNT, Z16, ZWT>(zh, zo.hpos(nb, p, u, i), v, su);
                return;
            }
            if (a.mode == 2) return;
            store_z<ZNT, Z16, ZWT>(zt, (unsigned)c * (unsigned)HF::Z_GROUP + pos, v, c == 0 ? su : sk);
        };
        batch_fft<N, 4, T, P>(fbuf, twr, tid, in, out);
        OCEAN_STAMP(3);
        return;
    } else {
    // -- batch A: slot 0 = pair 0 (Dx, Dz), slot 1 = pair 1 (sx, sz) ------------
    if (a.mode != 2 && (batches & 1)) {
        auto in = [&](int e, int c, int, int) -> c32 {
#pragma clang fp contract(off)
            float sv, tx, tz;
            fetch(e, sv, tx, tz);
#ifdef OCEAN_ABL_NOIN
            return make_float2(sv, tx);
#endif
            // (same operations, same order as zpass_input<0/1>, no contraction: bit-identical to the two-column variant)
            const float kz = kzt[e];
            const float d = __builtin_fmaf(kz, kz, kx2);
            const float inv = d > 1e-10f ? rsqrtf(d) : 0.0f;      // |k| > 1e-5 (.h:135)
            const float f = c ? -1.0f : inv;                       // pair 1: (-kz Tz, kx Tx); pair 0: (uz Tz, -ux Tx)
            return make_float2(kz * f * tz, -kx * f * tx);
        };
        auto out = [&](int p, int c, c32 v, int u, int i) {
#ifdef OCEAN_ABL_NOSTORE
            asm volatile("" ::"v"(v.x), "v"(v.y)); if (p >= 0) return;
#endif
            store_z<ZNT, Z16, ZWT>(zt, (unsigned)c * (unsigned)HF::Z_GROUP + zo.pos(nb, p, u, i), v, c ? sk : su);
        };
        batch_fft<N, 2, T, P>(fbuf, twr, tid, in, out);
        OCEAN_STAMP(2);
    }
    // -- batch B: slot 0 = pair 2 (dDx/dx, dDz/dz), slot 1 = height ----------------
    // OCEAN_MODE_JACOBIAN: slot 1 becomes pair 3 = (height, dDx/dz): the cross derivative's spectrum
    // i kz * (-i ux) h~ = kz ux h~ = kx uz h~ (.cpp:330-335; both of the reference's extra fields are this one), even
    // like the height, rides in the imaginary part the height transform leaves empty.  The result is then a full
    // complex column (not conjugate-symmetric in p any more): stored like the other pairs, as z group 3.
    const float full7 = (a.mode == 0 || a.mode == 3) ? 1.0f : 0.0f;
    const bool jac = a.mode == 3;
    if (batches & 2) {
        auto in = [&](int e, int c, int, int) -> c32 {
#pragma clang fp contract(off)
            float sv, tx, tz;
            fetch(e, sv, tx, tz);
#ifdef OCEAN_ABL_NOIN
            return make_float2(sv, tx);
#endif
            const float kz = kzt[e];
            // cross derivative: kx kz / |k| is odd in kx and in kz separately, so on the self-mirrored Nyquist column
            // (nb == 0) or row (e == 0) -- where one component of k(-idx) keeps its sign -- its Hermitian part takes
            // S- instead of S+ (both at once: S+ again)
            const float tc = COL0 ? (e == 0 ? sv : tx) : tz;
            // (same operations, same order as zpass_input<2/3>, no contraction)
            const float kz2 = kz * kz;
            const float d = __builtin_fmaf(kz, kz, kx2);
            const float inv = d > 1e-10f ? rsqrtf(d) : 0.0f;
            const float g = full7 * inv * sv;                      // pair 2 only exists in the 7-field modes
            return make_float2(c ? sv : kx2 * g, c ? (jac ? g3 * (kx * kz * inv * tc) : 0.0f) : kz2 * g);
        };
        auto out = [&](int p, int c, c32 v, int u, int i) {
#ifdef OCEAN_ABL_NOSTORE
            asm volatile("" ::"v"(v.x), "v"(v.y)); if (p >= 0) return;
#endif
            if (c) {
                if (jac) store_z<ZNT, Z16, ZWT>(z3, zo.pos(nb, p, u, i), v, s3);
                else if constexpr (zpass_half_height<N>()) return;                               // (the height follows below, as the other forms compute it)
                else if (zo.keeps(p, i)) store_z<ZNT, Z16, ZWT>(zh, zo.hpos(nb, p, u, i), v, su);     // real input: other half is the conjugate
                return;
            }
            store_z<ZNT, Z16, ZWT>(zt, 2u * (unsigned)HF::Z_GROUP + zo.pos(nb, p, u, i), v, sk);
        };
        batch_fft<N, 2, T, P>(fbuf, twr, tid, in, out);
        OCEAN_STAMP(3);
#ifdef OCEAN_HALF_HEIGHT_MIN
        if constexpr (zpass_half_height<N>()) {
            // from 2048 points up every form of the z pass computes the height as a real-input transform (zpass_height_half), so that a
            // column's bits do not depend on the form that happened to run it (here: the lone columns 0, 1 and N/2 of the two-column kernel)
            if (!jac) {
                __syncthreads();                    // the batch's last stage has read the image
                using HT = HalfHeightTwiddles<N, T, 1>;
                typename HT::type twh;
                c32 wk[HT::ITW];
                HT::from_table(a.tw, tid, twh, wk);
                float* const spx[1] = {const_cast<float*>(sp)};
                const int cols[1] = {nb};
                zpass_height_half<N, T, 1, ZNT, Z16>(a, fbuf, spx, twh, wk, tid, zh, cols, su,
                                                      [&](int e, int) { asm("" : "+v"(e)); float sv, tx, tz; fetch(e, sv, tx, tz); return sv; });
            }
        }
#endif
    }
    }
}

// Two neighbouring spectrum columns nb0, nb0 + 1 (neither the Nyquist column 0 nor beyond N/2) in one workgroup: four batches
// of two interleaved transforms, batch g = pair g (g = 3: the height, or pair 3 of the Jacobian mode) of BOTH columns, so that
// lanes 0-31 / 32-63 of a last-stage store hold the same 32 rows of column nb0 / nb0 + 1: 4 x (64 + 64) contiguous bytes.
template <int N, int T, class P, bool ZNT, bool Z16, bool FAST>
__device__ __forceinline__ void zpass_two_columns(const FrameArgs& a, unsigned char* smem, TwiddleRegs<N, 2, T, P>& twr,
                                                  int tid, int tile, int nb0)
{
    using HF = Half<N>;
    c32* fbuf = reinterpret_cast<c32*>(smem);
    float* sp0 = reinterpret_cast<float*>(fbuf + fft_lds_elems<N, 2>());   // S+ of column nb0, of column nb0 + 1
    float* sp1 = sp0 + N;
    float* raw = reinterpret_cast<float*>(fbuf);                           // [0], [1]: S-(0) of the two columns, until the first exchange
    const float t = a.t + (a.toff ? a.toff[tile] : 0.0f);
    const float* __restrict__ k1 = a.k1d + (size_t)tile * N;
    const float h16s = a.h0h ? a.h0_inv_scale[tile] : 1.0f;
    const float base = a.omega_q ? a.base_freq[tile] : 0.0f;
    spectrum_form<FAST>(a, [&](auto h16, auto w16) {   // phase 1: columns nb0 and nb0 + 1, each with its mirror (zpass_load_pair)
        constexpr bool H16 = decltype(h16)::value, W16 = decltype(w16)::value;
        constexpr int ITEMS = N;                      // 2 columns * N/2 element pairs
        constexpr int P1 = ITEMS / T;
        constexpr int PB = P1 > 2 ? 2 : P1;           // items in flight per thread (ten registers each; four spill under 2048's cap)
        static_assert(ITEMS % T == 0 && P1 % PB == 0, "phase-1 batches");
#pragma unroll 1
        for (int ub = 0; ub < P1; ub += PB) {
            float4 ha[PB];
            float2 hb0[PB], hb1[PB], wv[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int it = tid + (ub + u) * T;
                zpass_load_pair<N, H16, W16, ZNT>(a, tile, nb0 + it / (N / 2), 2 * (it % (N / 2)), h16s, base, ha[u], hb0[u], hb1[u], wv[u]);
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int it = tid + (ub + u) * T;
                const int c = it / (N / 2), n = 2 * (it % (N / 2));
                float a0, b0, a1, b1;
                animate_with_mirror(make_float2(ha[u].x, ha[u].y), hb0[u], wv[u].x, t, a0, b0);
                animate_with_mirror(make_float2(ha[u].z, ha[u].w), hb1[u], wv[u].y, t, a1, b1);
                *reinterpret_cast<float2*>((c ? sp1 : sp0) + n) = make_float2(0.5f * (a0 + b0), 0.5f * (a1 + b1));
                if (n == 0) raw[c] = 0.5f * (a0 - b0);
            }
        }
    });
    // kz of this thread's first-stage inputs: elements j + i * (N / R0) in every one of the four batches, whichever column the thread's
    // transform belongs to -- in registers (from the k table, once per workgroup) instead of an LDS table: N floats less LDS (2048: 51
    // instead of 59 KB, three workgroups per CU instead of two) and a third of the first stages' LDS reads
    static_assert(FirstStage<N, 2, T, P>::IT == 1, "one first-stage butterfly per thread");
    float kzr[P::r[0]];
#pragma unroll
    for (int i = 0; i < P::r[0]; ++i) kzr[i] = k1[tid / 2 + i * (N / P::r[0])];
    __syncthreads();
    const float sm00 = raw[0], sm01 = raw[1];      // S-(0) of the two columns

    [[maybe_unused]] float su = 1.0f, sk = 1.0f, s3 = 1.0f;
    float g3 = 1.0f;
    if constexpr (Z16) { const float4 zs = a.zscale[2 * tile]; su = zs.x; sk = zs.y; s3 = a.zscale[2 * tile + 1].x; }
    if (a.mode == 3) g3 = a.zscale[2 * tile + 1].y;
    constexpr size_t ES = Z16 ? 4 : 8;
    float2* __restrict__ zt = reinterpret_cast<float2*>(reinterpret_cast<char*>(a.z) + (size_t)tile * HF::Z_TILE * ES);
    float2* __restrict__ z3 = reinterpret_cast<float2*>(reinterpret_cast<char*>(a.z3) + (size_t)tile * HF::Z_GROUP * ES);
    float2* __restrict__ zh = reinterpret_cast<float2*>(reinterpret_cast<char*>(a.zh) + (size_t)tile * HF::ZH_TILE * ES);
    const float kx0 = k1[nb0], kx1 = k1[nb0 + 1];
    const bool jac = a.mode == 3;
    ZStore<N, T, P, 2, Z16> zo;
    zo.init(tid, nb0, true);
    // S+(e), Tz(e) (Tx = S+ off the Nyquist column), kx of column c
    const float kx20 = kx0 * kx0, kx21 = kx1 * kx1;
    auto fetch = [&](int e, int c, float& sv, float& tz, float& kx, float& kx2) {
        sv = c ? sp1[e] : sp0[e];
        tz = (e == 0) ? (c ? sm01 : sm00) : sv;
        kx = c ? kx1 : kx0;
        kx2 = c ? kx21 : kx20;
    };
    if (a.mode != 2) {
        if (a.zmask & 1) {   // pair 0: (uz Tz, -ux Tx)
            auto in = [&](int e, int c, int, int i) -> c32 {
                float sv, tz, kx, kx2; fetch(e, c, sv, tz, kx, kx2);
                return zpass_input<0>(kx, kx2, kzr[i], sv, sv, tz, tz, 1.0f, false, 1.0f);
            };
            auto out = [&](int p, int c, c32 v, int u, int i) { store_z<ZNT, Z16>(zt, zo.pos(nb0 + c, p, u, i), v, su); };
            batch_fft<N, 2, T, P>(fbuf, twr, tid, in, out);
        }
        if (a.zmask & 2) {   // pair 1: (-kz Tz, kx Tx)
            auto in = [&](int e, int c, int, int i) -> c32 {
                float sv, tz, kx, kx2; fetch(e, c, sv, tz, kx, kx2);
                return zpass_input<1>(kx, kx2, kzr[i], sv, sv, tz, tz, 1.0f, false, 1.0f);
            };
            auto out = [&](int p, int c, c32 v, int u, int i) { store_z<ZNT, Z16>(zt, (unsigned)HF::Z_GROUP + zo.pos(nb0 + c, p, u, i), v, sk); };
            batch_fft<N, 2, T, P>(fbuf, twr, tid, in, out);
        }
    }
    if ((a.mode == 0 || a.mode == 3) && (a.zmask & 4)) {   // pair 2: (kx ux S+, kz uz S+) -- only the 7-field modes read it
        auto in = [&](int e, int c, int, int i) -> c32 {
            float sv, tz, kx, kx2; fetch(e, c, sv, tz, kx, kx2);
            return zpass_input<2>(kx, kx2, kzr[i], sv, sv, tz, tz, 1.0f, false, 1.0f);
        };
        auto out = [&](int p, int c, c32 v, int u, int i) { store_z<ZNT, Z16>(zt, 2u * (unsigned)HF::Z_GROUP + zo.pos(nb0 + c, p, u, i), v, sk); };
        batch_fft<N, 2, T, P>(fbuf, twr, tid, in, out);
    }
#ifdef OCEAN_HALF_HEIGHT_MIN
    if constexpr (zpass_half_height<N>()) {
        if ((a.zmask & 8) && !jac) {      // both columns' heights as real-input transforms (zpass_height_half: the single-transform form's bits)
            using HT = HalfHeightTwiddles<N, T, 2>;
            typename HT::type twh;
            c32 wk[HT::ITW];
            HT::from_full(twr, twh, wk);
            float* const spx[2] = {sp0, sp1};
            const int cols[2] = {nb0, nb0 + 1};
            zpass_height_half<N, T, 2, ZNT, Z16>(a, fbuf, spx, twh, wk, tid, zh, cols, su, [&](int e, int c) { return c ? sp1[e] : sp0[e]; });
            return;
        }
    }
#endif
    if (a.zmask & 8) {   // height (or pair 3 = (height, cross derivative) of the Jacobian mode)
        auto in = [&](int e, int c, int, int i) -> c32 {
            float sv, tz, kx, kx2; fetch(e, c, sv, tz, kx, kx2);
            if (!jac) return make_float2(sv, 0.0f);
            return zpass_input<3>(kx, kx2, kzr[i], sv, sv, tz, tz, 1.0f, true, g3);
        };
        auto out = [&](int p, int c, c32 v, int u, int i) {
            if (jac) store_z<ZNT, Z16>(z3, zo.pos(nb0 + c, p, u, i), v, s3);
            else if (zo.keeps(p, i)) store_z<ZNT, Z16>(zh, zo.hpos(nb0 + c, p, u, i), v, su);     // real input: other half is the conjugate
        };
        batch_fft<N, 2, T, P>(fbuf, twr, tid, in, out);
    }
}

// ============================================================================
// k_zpass (first pass, z axis): workgroup = spectrum COLUMN nb (kx index,
// blockIdx.x in [0, N/2]); mirror column nbb = (N-nb)%N.  Both are contiguous
// runs of the transposed spectrum.  Along the column, e = kz index:
//   a(e) = h~(e, nb), b(e) = h~((N-e)%N, nbb)
//   S+ = (a+b)/2, S- = (a-b)/2;   Tx = (nb==0 ? S- : S+), Tz = (e==0 ? S- : S+)
//   (k(-idx) = -k(idx) except on the self-mirrored Nyquist row/column 0)
//   pair 0: Dx_h + i Dz_h     = ( uz*Tz, -ux*Tx)      odd   (c = -i u, .cpp:323-326)
//   pair 1: sx_h + i sz_h     = (-kz*Tz,  kx*Tx)      odd   (c =  i k, .cpp:309-310)
//   pair 2: dxDx_h + i dzDz_h = (kx*ux*S+, kz*uz*S+)  even  (.cpp:327-330)
//   height: S+                                         even
// Two batches of two interleaved transforms over e: {pair 0, pair 1}, {pair 2, height};
// output index p = z position; stored per column nb as side 0 (p <= N/2) / side 1 (N-p).
// ============================================================================
// minimum waves per SIMD asked of the register allocator: three 512-thread workgroups per CU at 2048 (since round 3 the kernel
// needs 62 VGPRs and would fit four, but its 51 KB of LDS allow three); 1024 and 4096 keep the looser bound
template <int N> constexpr int zpass_min_waves() { return N == 2048 ? 6 : (N >= 1024 ? 3 : 1); }
// (two columns per workgroup: since round 4 its kz live in registers, the LDS footprint is the one-column form's and so is the bound)
#ifndef OCEAN_ZLB
#define OCEAN_ZLB zpass_min_waves<N>()
#endif
// The body of k_zpass as a device function: k_frame (the one-launch frame of pipelined small tiles) runs it for its first N/2 + 1 workgroups.
// bx / gx: the workgroup's index and count among the z-pass workgroups of its tile; ONE: part of a one-launch frame -- the intermediates go out
// write-through and the per-tile words the x-axis workgroups of the SAME launch update by atomics are reset write-through too.
template <int N, int T, class P, bool ZNT, bool Z16, int ZW, bool FAST, bool ONE = false>
__device__ __forceinline__ void zpass_body(const FrameArgs& a, unsigned char* smem, const int bx, const int gx)
{
    constexpr int ZC = zpass_columns<N>();
    c32* fbuf = reinterpret_cast<c32*>(smem);                              // ZC interleaved transforms
    float* sp = reinterpret_cast<float*>(fbuf + fft_lds_elems<N, ZC>());   // S+ [N]
    float* kzt = sp + N;                                                   // kz table [N]
    float* raw = reinterpret_cast<float*>(fbuf);                           // [0]: S-(0) of the column, until the first exchange

    const int tid = threadIdx.x;
    const int tile = blockIdx.y;
    // (Rounds 2-3 split the columns of a partially filled last round of a serial 2048^2 frame -- 1025 columns on 768 resident
    //  workgroups -- over two workgroups each, one per batch, both animating the column; since round 4 that tile size runs the
    //  single-transform form, k_zpass_c1, whose 1025 smaller workgroups are resident at once.)
    static_assert(ZW == 1 || zpass_has_width2<N>(), "two columns per workgroup: two-batch sizes from 1024 up");
    TwiddleRegs<N, ZC, T, P> twr;
    twr.load(a.tw, tid);
    auto one_column = [&](const int nb, const int batches) {
    const float t = a.t + (a.toff ? a.toff[tile] : 0.0f);
    const float* __restrict__ k1 = a.k1d + (size_t)tile * N;               // [N], cache-resident table
    const bool col0 = (nb == 0);
    const float h16s = a.h0h ? a.h0_inv_scale[tile] : 1.0f;
    const float base = a.omega_q ? a.base_freq[tile] : 0.0f;

    OCEAN_STAMP(0);
    // -- phase 1: animate column nb with its mirror nbb (zpass_load_pair); all loads issued before the first sincos.
    // S- is needed along the whole column only for the Nyquist column nb == 0 (Tx = S-); every other column needs just
    // S-(0) (Tz at e == 0) and keeps the kz table in LDS instead.  The Nyquist column pairs with itself (nbb == 0): S+ is
    // even and S- odd along e, so ONE array G(e) = h~(e, 0) carries both; every other column stores S+ directly.
    spectrum_form<FAST>(a, [&](auto h16, auto w16) {
        constexpr bool H16 = decltype(h16)::value, W16 = decltype(w16)::value;
        constexpr int PAIRS = N / 2;
        constexpr int P1 = (PAIRS + T - 1) / T;
        constexpr int PB = P1 > 4 ? 4 : P1;          // items in flight per thread
        static_assert(P1 % PB == 0, "phase-1 batches");
#pragma unroll 1
        for (int ub = 0; ub < P1; ub += PB) {
            float4 ha[PB];
            float2 hb0[PB], hb1[PB], wv[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int it = tid + (ub + u) * T;
                if (PAIRS % T == 0 || it < PAIRS) zpass_load_pair<N, H16, W16, ZNT>(a, tile, nb, 2 * it, h16s, base, ha[u], hb0[u], hb1[u], wv[u]);
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int it = tid + (ub + u) * T;
                if (PAIRS % T == 0 || it < PAIRS) {
                    const int n = 2 * it;
                    float a0, b0, a1, b1;
                    animate_with_mirror(make_float2(ha[u].x, ha[u].y), hb0[u], wv[u].x, t, a0, b0);
                    animate_with_mirror(make_float2(ha[u].z, ha[u].w), hb1[u], wv[u].y, t, a1, b1);
                    *reinterpret_cast<float2*>(sp + n) = col0 ? make_float2(a0, a1) : make_float2(0.5f * (a0 + b0), 0.5f * (a1 + b1));
                    *reinterpret_cast<float2*>(kzt + n) = *reinterpret_cast<const float2*>(k1 + n);
                    if (n == 0) raw[0] = 0.5f * (a0 - b0);          // S-(0), for everybody (the FFT image is not in use yet)
                }
            }
        }
    });
    if (bx == 0 && tid == 0 && (a.zmask & 8)) {   // (the launch that transforms the height: ahead of the HEIGHT workgroups' atomics)
        // min starts at FLT_MAX, max at FLT_MIN (> 0): WSTessendorf.cpp:289-290
        if constexpr (ONE) {        // the atomics of this very launch's HEIGHT workgroups follow (behind zdone): the resets must not sit in this XCD's L2
            __hip_atomic_store(a.minmax + 2 * tile + 0, float_key(3.402823466e+38f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.minmax + 2 * tile + 1, float_key(1.175494351e-38f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.hdone + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            a.minmax[2 * tile + 0] = float_key(3.402823466e+38f);
            a.minmax[2 * tile + 1] = float_key(1.175494351e-38f);
            a.hdone[tile] = 0u;                                   // (merged x pass: its DISP workgroups count the HEIGHT workgroups up from here)
        }
    }
    __syncthreads();
    const float sm0 = raw[0];
    OCEAN_STAMP(1);

    if (col0) zpass_transforms<N, T, P, true, ZNT, Z16, ZC, ONE>(a, fbuf, sp, kzt, twr, k1[nb], sm0, tid, tile, nb, batches);
    else zpass_transforms<N, T, P, false, ZNT, Z16, ZC, ONE>(a, fbuf, sp, kzt, twr, k1[nb], sm0, tid, tile, nb, batches);
    };

    if constexpr (ZW == 2) {
        // N/4 + 1 workgroups: block 0 = the Nyquist column 0 and column 1 one after the other, blocks 1 .. N/4-1 = columns
        // 2b, 2b+1 together, the last one = column N/2 alone (dispatched last: the shortest job closes the grid)
        constexpr int LAST = N / 4;
        const int blk = bx == LAST ? LAST : xcd_swizzle(bx, LAST);
        if (blk != 0 && blk != LAST) {
            zpass_two_columns<N, T, P, ZNT, Z16, FAST>(a, smem, twr, tid, tile, 2 * blk);
            return;
        }
        if (blk == LAST) { one_column(N / 2, 3); return; }
        one_column(0, 3);
        __syncthreads();            // the slowest wave is done with the FFT image before the next column's h~ overwrites it
        one_column(1, 3);
    } else {
        int nb = bx;
        const int batches = 3;
#if OCEAN_ZTILE
        nb = xcd_swizzle(nb, gx);     // neighbouring columns write neighbouring pieces of the same lines: same XCD, same L2
#endif
        one_column(nb, batches);
    }
}

template <int N, int T, class P = Plan<N>, bool ZNT = false, bool Z16 = false, int ZW = 1, bool FAST = true>
__global__ void __launch_bounds__(T, OCEAN_ZLB) k_zpass(const FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    zpass_body<N, T, P, ZNT, Z16, ZW, FAST>(a, smem, (int)blockIdx.x, (int)gridDim.x);
}

// A launch whose workgroups are all resident at once on an otherwise idle device reads everything, then writes everything: the first quarter of
// a serial 2048^2 z pass is a load burst with the write path idle, the rest a store burst with the read path idle (a timing-only build without
// any arithmetic takes 19.3 of the 22.2 us).  Spreading the workgroups' starts over a few microseconds -- workgroup i of G waits i / G of the
// ramp, i in dispatch order -- lets the early ones store while the late ones still load.  Same box, interleaved, us: k_zpass_c1 23.0-23.4 ->
// 21.9-22.4 with 5.0 us, k_xpass_disp 17.4-18.0 -> 16.1-16.4 with 4.5 us, k_xpass_b 20.2-20.4 -> 19.2-19.4 with 4.5 us over its NORMAL
// workgroups alone (the height workgroups, which write next to nothing, all start at once; a ramp over all of them gains nothing).  Pipelined
// frames, whose chains run in lockstep -- three such kernels side by side --, gain 1.2 us per frame with 5.0 / 9.0 / 9.0 us.  Only frames of
// ONE 2048^2 tile gain (512^2, 1024^2: a loss; 4096^2 and batches -- several rounds, which overlap by themselves: nothing), in every mode and
// precision, so only they ask for it (ocean_launch.h).  profiles/r04_zpass_experiments.txt items 10-12.
__device__ __forceinline__ void start_ramp_wait(int ramp, unsigned idx, unsigned count)
{
    if (ramp > 0) {
        const unsigned until = idx * (unsigned)ramp / count;                   // 100 MHz ticks (s_memrealtime)
        const unsigned long long t0 = wall_clock64();
        while ((unsigned)(wall_clock64() - t0) < until) __builtin_amdgcn_s_sleep(4);
    }
}

#ifdef OCEAN_CLOCKPROBE
// diagnostic build only (tools/slow_window.py, profiles/r06_slow_window.txt): what clock did a launch of the single-transform z pass run at?
// Thread 0 of every workgroup reads the shader-clock counter (s_memtime: one tick per shader cycle) and the constant 100 MHz counter
// (s_memrealtime) when the workgroup starts and when its last store has been issued; record [launch % LAUNCHES][workgroup][4] =
// {start, end (100 MHz ticks), shader cycles in between, XCC id | HW id << 32}.  The in-kernel clock of the launch is the median over its
// workgroups of cycles / ticks x 100 MHz (MI355X_MICROARCH.md, DVFS give-back (6)); its duration max(end) - min(start).  No output value
// depends on a stamp; the shipped library has none of this.
static __device__ unsigned long long* g_clockprobe = nullptr;
constexpr unsigned CLOCKPROBE_LAUNCHES = 4096, CLOCKPROBE_WGS = 1032;
struct ClockProbe {
    unsigned long long* p;
    unsigned long long c0, r0;
    __device__ explicit ClockProbe(unsigned launch) : p(nullptr), c0(0), r0(0)
    {
        if (threadIdx.x == 0 && g_clockprobe && blockIdx.y == 0 && blockIdx.x < CLOCKPROBE_WGS) {
            p = g_clockprobe + 4 * ((size_t)(launch % CLOCKPROBE_LAUNCHES) * CLOCKPROBE_WGS + blockIdx.x);
            c0 = __builtin_amdgcn_s_memtime();
            r0 = __builtin_amdgcn_s_memrealtime();
        }
    }
    __device__ void end()
    {
        if (p) {
            const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            p[0] = r0; p[1] = r1; p[2] = c1 - c0; p[3] = (unsigned long long)xcc | ((unsigned long long)hwid << 32);
        }
    }
};
#endif

// ============================================================================
// k_zpass_c1: the z pass with ONE transform per batch (four batches per column: pair 0, pair 1, pair 2, height) and half the threads.
// For 4096^2: the two-transform forms need 102-119 KB of LDS (70 KB of FFT image + the S+ / kz tables) -- ONE 1024-thread workgroup per
// CU, whose load burst, butterflies, exchanges and stores run strictly one after the other (0.43 of peak, the phases of a
// lone workgroup never overlap).  With one transform in the image the workgroup needs 35 + 32 = 67 KB and 512 threads: TWO independent
// workgroups per CU, the same number of waves, and each one's load burst and store tail travel under the other's transforms.  Same radix
// plan, same inputs (zpass_input), same twiddles: bit-identical to the other forms (tests/test_variants_gpu.py).
// ============================================================================
template <int N, int T, class P, bool COL0, bool ZNT, bool Z16, bool ZWT = false>
__device__ __forceinline__ void zpass_single_transforms(const FrameArgs& a, c32* fbuf, const float* sp, const float (&kzr)[P::r[0]],
                                                        TwiddleRegs<N, 1, T, P>& twr, float kx, float sm0, int tid, int tile, int nb)
{
    using HF = Half<N>;
    // kz of the first stage's inputs: thread j reads elements j + i * (N / R0), i = 0 .. R0-1, in every one of the four batches -- the
    // same R0 wave-vector components each time, so they live in registers (kzr[i], fetched once per workgroup from the k table) instead
    // of an LDS table: 16 KB less LDS at 4096 -- three workgroups per CU instead of two -- and a quarter of the first stages' LDS reads.
    static_assert(FirstStage<N, 1, T, P>::IT == 1, "one first-stage butterfly per thread");
    const float kx2 = kx * kx;
    [[maybe_unused]] float su = 1.0f, sk = 1.0f, s3 = 1.0f;
    float g3 = 1.0f;
    if constexpr (Z16) { const float4 zs = a.zscale[2 * tile]; su = zs.x; sk = zs.y; s3 = a.zscale[2 * tile + 1].x; }
    if (a.mode == 3) g3 = a.zscale[2 * tile + 1].y;
    constexpr size_t ES = Z16 ? 4 : 8;
    float2* __restrict__ zt = reinterpret_cast<float2*>(reinterpret_cast<char*>(a.z) + (size_t)tile * HF::Z_TILE * ES);
    float2* __restrict__ z3 = reinterpret_cast<float2*>(reinterpret_cast<char*>(a.z3) + (size_t)tile * HF::Z_GROUP * ES);
    float2* __restrict__ zh = reinterpret_cast<float2*>(reinterpret_cast<char*>(a.zh) + (size_t)tile * HF::ZH_TILE * ES);
    const bool jac = a.mode == 3;
    ZStore<N, T, P, 1, Z16> zo;
    zo.init(tid, nb);
    // S+(e), Tx(e), Tz(e), Tc(e) (see k_zpass: the Nyquist column carries G(e) = h~(e, 0) and forms S+- on the fly)
    auto fetch = [&](int e, float& sv, float& tx, float& tz, float& tc) {
        if constexpr (COL0) {
            const float g1 = sp[e], g2 = sp[(N - e) & (N - 1)];
            sv = 0.5f * (g1 + g2);
            tx = 0.5f * (g1 - g2);
            tz = (e == 0) ? tx : sv;
            tc = (e == 0) ? sv : tx;
        } else {
            sv = sp[e];
            tx = sv;
            tz = (e == 0) ? sm0 : sv;
            tc = tz;
        }
    };
    // (a.zmask: which of the four this launch runs -- all of them, or one half of the split frame order; wave-uniform)
    if (a.mode != 2) {
        if (a.zmask & 1) {   // pair 0: (uz Tz, -ux Tx)
            auto in = [&](int e, int, int, int i) -> c32 {
                float sv, tx, tz, tc; fetch(e, sv, tx, tz, tc);
                return zpass_input<0>(kx, kx2, kzr[i], sv, tx, tz, tc, 1.0f, false, 1.0f);
            };
            auto out = [&](int p, int, c32 v, int u, int i) { store_z<ZNT, Z16, ZWT>(zt, zo.pos(nb, p, u, i), v, su); };
            batch_fft<N, 1, T, P>(fbuf, twr, tid, in, out);
        }
        if (a.zmask & 2) {   // pair 1: (-kz Tz, kx Tx)
            auto in = [&](int e, int, int, int i) -> c32 {
                float sv, tx, tz, tc; fetch(e, sv, tx, tz, tc);
                return zpass_input<1>(kx, kx2, kzr[i], sv, tx, tz, tc, 1.0f, false, 1.0f);
            };
            auto out = [&](int p, int, c32 v, int u, int i) { store_z<ZNT, Z16, ZWT>(zt, (unsigned)HF::Z_GROUP + zo.pos(nb, p, u, i), v, sk); };
            batch_fft<N, 1, T, P>(fbuf, twr, tid, in, out);
        }
    }
    if ((a.mode == 0 || a.mode == 3) && (a.zmask & 4)) {   // pair 2: (kx ux S+, kz uz S+) -- only the 7-field modes read it
        auto in = [&](int e, int, int, int i) -> c32 {
            float sv, tx, tz, tc; fetch(e, sv, tx, tz, tc);
            return zpass_input<2>(kx, kx2, kzr[i], sv, tx, tz, tc, 1.0f, false, 1.0f);
        };
        auto out = [&](int p, int, c32 v, int u, int i) { store_z<ZNT, Z16, ZWT>(zt, 2u * (unsigned)HF::Z_GROUP + zo.pos(nb, p, u, i), v, sk); };
        batch_fft<N, 1, T, P>(fbuf, twr, tid, in, out);
    }
#ifdef OCEAN_HALF_HEIGHT_MIN
    if constexpr (zpass_half_height<N>()) {
        if ((a.zmask & 8) && !jac) {      // the height as a real-input transform: half the size + one split step (zpass_height_half)
            using HT = HalfHeightTwiddles<N, T, 1>;
            typename HT::type twh;
            c32 wk[HT::ITW];
            HT::from_full(twr, twh, wk);
            float* const spx[1] = {const_cast<float*>(sp)};
            const int cols[1] = {nb};
            zpass_height_half<N, T, 1, ZNT, Z16>(a, fbuf, spx, twh, wk, tid, zh, cols, su,
                                                  [&](int e, int) {
                                                      // (opaque index: otherwise the Nyquist column's mirror index (N - e) % N is recognised as phase 1's and kept
                                                      //  alive -- through a spill -- across the whole kernel instead of being recomputed in two instructions)
                                                      asm("" : "+v"(e));
                                                      float sv, tx, tz, tc; fetch(e, sv, tx, tz, tc); return sv; });
            return;
        }
    }
#endif
    if (a.zmask & 8) {   // height (or pair 3 = (height, cross derivative) of the Jacobian mode)
        auto in = [&](int e, int, int, int i) -> c32 {
            float sv, tx, tz, tc; fetch(e, sv, tx, tz, tc);
            if (!jac) return make_float2(sv, 0.0f);
            return zpass_input<3>(kx, kx2, kzr[i], sv, tx, tz, tc, 1.0f, true, g3);
        };
        auto out = [&](int p, int, c32 v, int u, int i) {
            if (jac) store_z<ZNT, Z16, ZWT>(z3, zo.pos(nb, p, u, i), v, s3);
            else if (zo.keeps(p, i)) store_z<ZNT, Z16, ZWT>(zh, zo.hpos(nb, p, u, i), v, su);     // real input: other half is the conjugate
        };
        batch_fft<N, 1, T, P>(fbuf, twr, tid, in, out);
    }
}

// (the instantiations that carry all four forms of the spectrum -- !FAST: fp16 copy, fp32 dispersion -- need a few registers more than the
//  80 of six waves per SIMD and spilled 24-28 bytes per lane under that cap: they ask for five, 96 registers, no scratch)
template <int N> constexpr size_t zpass_c1_lds_bytes() { return sizeof(c32) * fft_lds_elems<N, 1>() + sizeof(float) * N; }
// (a radix-16 plan -- N / 16 threads, two waves per 2048-point workgroup -- may use the registers of three waves per SIMD: six workgroups per CU)
template <int N, int T, bool FAST> constexpr int zpass_c1_min_waves() { return T == N / 16 ? 3 : (FAST ? 6 : 5); }
template <int N, int T, class P, bool ZNT = false, bool Z16 = false, bool FAST = true, bool ZWT = false>
__global__ void __launch_bounds__(T, (zpass_c1_min_waves<N, T, FAST>())) k_zpass_c1(const FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    c32* fbuf = reinterpret_cast<c32*>(smem);                              // one transform
    float* sp = reinterpret_cast<float*>(fbuf + fft_lds_elems<N, 1>());    // S+ [N]
    float* raw = reinterpret_cast<float*>(fbuf);                           // [0]: S-(0) of the column, until the first exchange
    const int tid = threadIdx.x;
    const int tile = blockIdx.y;
#ifdef OCEAN_CLOCKPROBE
    ClockProbe clock_probe_(a.frame_seq);
#endif
    start_ramp_wait(a.start_ramp, blockIdx.x, gridDim.x);         // (one 2048^2 tile: ocean_launch.h) -- ahead of every load: nothing is live across the wait
    TwiddleRegs<N, 1, T, P> twr;
    twr.load(a.tw, tid);
#ifdef OCEAN_DEVELOPER      // (experiment: XCD x > 0 takes the column group of XCD 1 + (x - 1 + rot) % 7; group 0 holds one column more and stays)
    const int bxr = (blockIdx.x % 8 == 0 || a.xcd_rot == 0) ? (int)blockIdx.x : (int)(blockIdx.x / 8 * 8 + 1 + (blockIdx.x % 8 - 1 + a.xcd_rot) % 7);
    const int nb = xcd_swizzle(bxr, N / 2 + 1);
#else
    const int nb = xcd_swizzle((int)blockIdx.x, N / 2 + 1);     // neighbouring columns write neighbouring pieces of the same lines: same XCD, same L2
#endif
    const float t = a.t + (a.toff ? a.toff[tile] : 0.0f);
    const float* __restrict__ k1 = a.k1d + (size_t)tile * N;
    const bool col0 = (nb == 0);
    const float h16s = a.h0h ? a.h0_inv_scale[tile] : 1.0f;
    const float base = a.omega_q ? a.base_freq[tile] : 0.0f;
    // phase 1 exactly as in k_zpass (zpass_load_pair / animate_with_mirror)
    spectrum_form<FAST>(a, [&](auto h16, auto w16) {
        constexpr bool H16 = decltype(h16)::value, W16 = decltype(w16)::value;
        constexpr int PAIRS = N / 2;
        constexpr int P1 = (PAIRS + T - 1) / T;
        constexpr int PB = P1 > 4 ? 4 : P1;          // items in flight per thread
        static_assert(P1 % PB == 0 && PAIRS % T == 0, "phase-1 batches");
#pragma unroll 1
        for (int ub = 0; ub < P1; ub += PB) {
            float4 ha[PB];
            float2 hb0[PB], hb1[PB], wv[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u)
                zpass_load_pair<N, H16, W16, ZNT>(a, tile, nb, 2 * (tid + (ub + u) * T), h16s, base, ha[u], hb0[u], hb1[u], wv[u]);
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int n = 2 * (tid + (ub + u) * T);
                float a0, b0, a1, b1;
                animate_with_mirror(make_float2(ha[u].x, ha[u].y), hb0[u], wv[u].x, t, a0, b0);
                animate_with_mirror(make_float2(ha[u].z, ha[u].w), hb1[u], wv[u].y, t, a1, b1);
                *reinterpret_cast<float2*>(sp + n) = col0 ? make_float2(a0, a1) : make_float2(0.5f * (a0 + b0), 0.5f * (a1 + b1));
                if (n == 0) raw[0] = 0.5f * (a0 - b0);          // S-(0), for everybody (the FFT image is not in use yet)
            }
        }
    });
    if (blockIdx.x == 0 && tid == 0 && (a.zmask & 8)) {   // (the launch that transforms the height: ahead of the HEIGHT workgroups' atomics)
        // min starts at FLT_MAX, max at FLT_MIN (> 0): WSTessendorf.cpp:289-290
        a.minmax[2 * tile + 0] = float_key(3.402823466e+38f);
        a.minmax[2 * tile + 1] = float_key(1.175494351e-38f);
        a.hdone[tile] = 0u;                                   // (merged x pass: its DISP workgroups count the HEIGHT workgroups up from here)
    }
    float kzr[P::r[0]];                                   // kz of this thread's first-stage inputs (behind phase 1: the registers are free by now)
#pragma unroll
    for (int i = 0; i < P::r[0]; ++i) kzr[i] = k1[tid + i * (N / P::r[0])];
    __syncthreads();
    const float sm0 = raw[0];
    static_assert(!ZWT || (!ZNT && !Z16), "write-through is a policy of the plain fp32 intermediates");
    if (col0) zpass_single_transforms<N, T, P, true, ZNT, Z16, ZWT>(a, fbuf, sp, kzr, twr, k1[nb], sm0, tid, tile, nb);
    else zpass_single_transforms<N, T, P, false, ZNT, Z16, ZWT>(a, fbuf, sp, kzr, twr, k1[nb], sm0, tid, tile, nb);
#ifdef OCEAN_CLOCKPROBE
    clock_probe_.end();
#endif
}
// tile sizes whose z pass has the single-transform form (the launcher picks it where it is faster: ocean_launch.h)
template <int N> constexpr bool zpass_has_c1() { return N >= 1024; }
// Where the single-transform form is the faster one (profiles/r04_zpass_experiments.txt; stream_maps: ocean_ctx.h, bit 2 = streamed
// intermediates): 4096^2 always (three workgroups per CU instead of one: z pass 111 -> 95 us); 2048^2 and batches of 1024^2 with plain
// intermediate stores (2048^2 24.4 -> 23.2 us and no split last round, 8 x 1024^2 38.1 -> 33.9); a lone 1024^2 tile keeps the
// two-transform form (half the dependent chain: 14.3 vs 15.2 us), streamed intermediates below 4096 the two-column form (whole-line stores).
template <int N> inline bool zpass_c1_pays(int stream_maps, unsigned tiles)
{
    if (N == 4096) return true;
    if (stream_maps & 4) return false;
    return N == 2048 || tiles >= 2;
}

// resident z-pass workgroups per CU (lower bound from LDS, threads and the register cap of the launch bounds)
template <int N, int T> constexpr int zpass_blocks_per_cu()
{
    constexpr int by_lds = (int)(163840 / (sizeof(c32) * fft_lds_elems<N, zpass_columns<N>()>() + sizeof(float) * 2 * N));
    constexpr int by_threads = 2048 / T;
    constexpr int by_regs = (zpass_min_waves<N>() * 4 * 64) / T;
    constexpr int m = by_lds < by_threads ? by_lds : by_threads;
    return (by_regs >= 1 && by_regs < m) ? by_regs : (m < 1 ? 1 : m);
}

template <int N, int ZW = 1> constexpr size_t zpass_lds_bytes()
{
    // FFT image + two tables of N floats: S+ and kz of one column, or S+ of both columns of the two-column form (whose kz live in registers)
    return sizeof(c32) * fft_lds_elems<N, zpass_columns<N>()>() + sizeof(float) * 2 * N;
}


// ---- hand-off inside one launch (the merged x pass) ------------------------------------------
// The XCDs' L2s are not coherent with each other and a CU's L1 is never refreshed by another CU's stores: what one workgroup hands to
// another INSIDE a launch travels write-through -- every store of the handed-off bytes an agent-scope relaxed atomic store (`sc1`: it
// leaves the XCD's L2 at once, no release of the whole L2 needed), every load of them an agent-scope relaxed atomic load (`sc1`: served
// past the L1) -- and is announced by ONE lane's agent-scope atomic add behind every storing wave's s_waitcnt vmcnt(0) and the
// workgroup's barrier; the consumer polls the counter with such a load from one lane, then joins a workgroup barrier
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility": valid forms, the table's first row).
__device__ __forceinline__ void store_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load_wt(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned load_wt(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one lane waits until *ctr has reached `target` (bounded: 20 ms -- a producer that never arrives must not hang the device; the frame is
// then wrong, the lane says so in FrameArgs::fault and the host reports an error), then the workgroup's barrier
__device__ __forceinline__ void wait_counter(const unsigned* ctr, unsigned target, int tid, int poll_sleep = 0, unsigned* fault = nullptr)
{
    if (tid == 0) {
#ifdef OCEAN_FAULT_INJECT      // (tools/fault_probe.py: a variant build whose DISP workgroups wait for one arrival too many -- the wait must give up and the host must say so)
        target += 1u;
#endif
        const unsigned long long t0 = wall_clock64();
        bool ok;
        while (!(ok = load_wt(ctr) >= target) && (wall_clock64() - t0) < 2000000ull) {
            if (poll_sleep <= 0) __builtin_amdgcn_s_sleep(2);
            else for (int k = 0; k < poll_sleep; ++k) __builtin_amdgcn_s_sleep(127);
        }
        if (!ok && fault) *fault = 1u;              // gave up: the host turns this into an error (ocean_api.hip: check_fault)
    }
    __syncthreads();
}

// the same for a counter that is never reset: until it has REACHED target (mod 2^32: the frames of a chain count it up for ever)
__device__ __forceinline__ void wait_counter_reached(const unsigned* ctr, unsigned target, int tid, int poll_sleep = 0, unsigned* fault = nullptr)
{
    if (tid == 0) {
        const unsigned long long t0 = wall_clock64();
        bool ok;
        while (!(ok = (int)(load_wt(ctr) - target) >= 0) && (wall_clock64() - t0) < 2000000ull) {
            if (poll_sleep <= 0) __builtin_amdgcn_s_sleep(2);
            else for (int k = 0; k < poll_sleep; ++k) __builtin_amdgcn_s_sleep(127);
        }
        if (!ok && fault) *fault = 1u;
    }
    __syncthreads();
}

// ---- x-pass helpers ------------------------------------------------------------------
// Column u of a packed pair: rows 0..N/2 come from side 0; row mf > N/2 is the
// mirror image eps * Z(N-mf, N-u) = eps * side 1 of row N-mf.
template <int N, bool Z16 = false>
__device__ __forceinline__ c32 load_pair_column(const float2* __restrict__ zg, int mf, int u, float eps, float unscale = 1.0f, float unscale_y = 0.0f)
{
    using HF = Half<N>;
#ifdef OCEAN_ABL_NOLOAD
    return make_float2(1.0f + mf, 0.5f * u);
#endif
    if (mf <= N / 2) return load_z<Z16>(zg, HF::template zidx<Z16>(mf, 0, u), unscale, unscale_y);
    // mirror of the self-mirrored units 0 and N/2 is the unit itself (side 0)
    const int side = (u == 0 || u == N / 2) ? 0 : 1;
    if constexpr (Z16) return load_z<true>(zg, HF::template zidx<Z16>(N - mf, side, u), eps * unscale, eps * unscale_y);
    const c32 v = load_z<false>(zg, HF::template zidx<Z16>(N - mf, side, u), 1.0f);
    return make_float2(eps * v.x, eps * v.y);
}

// Visits the outputs (position p, column c) the LAST stage of a transform would hand to this
// thread -- used by the reduced modes that skip a transform but still have to write its texels.
template <class LS, int T, class F>
__device__ __forceinline__ void for_each_output(int tid, F f)
{
#pragma unroll
    for (int u = 0; u < LS::IT; ++u) {
        const int w = tid + u * T;
        if (!LS::GUARD || w < LS::ITEMS) {
            int c, j;
            LS::map(w, c, j);
#pragma unroll
            for (int i = 0; i < LS::RL; ++i) f(j + i * LS::STRIDE, c, u, i);
        }
    }
}

// ============================================================================
// x pass (second pass, x axis), two launches:
//   k_xpass_b    blockIdx.x < HB  -> HEIGHT workgroup: 2*C rows of the height (C transforms of
//                                    two real rows each: Y_u + i Y_{u+1}), sign, raw signed
//                                    height out, global min/max
//                otherwise        -> NORMAL workgroup: pairs 1 and 2 -> normal-map rows
//   k_xpass_disp pair 0 + raw height -> displacement-map rows (needs the min/max);
//                NormalizeHeights (.cpp:443-455) is folded into the store
// The height transforms run beside the normal-map ones instead of alone.  (A merged form --
// one height launch, then all three pairs per workgroup with register prefetch -- was
// measured slower at every size and removed.)
// ============================================================================
#ifdef OCEAN_XB_TRACE
// diagnostic build only (tools/archive/xb_trace.py, profiles/r03_xpass_trace.txt): per workgroup of the two x passes, where and when it ran --
// [record][4] = {start, end (100 MHz wall clock), HW_REG_HW_ID, HW_REG_XCC_ID}; records 0.. = k_xpass_b's workgroups, 512.. = k_xpass_disp's
static __device__ unsigned long long* g_xb_trace = nullptr;
struct XbTrace {
    unsigned long long* p;
    __device__ explicit XbTrace(unsigned base = 0) : p(nullptr)
    {
        if (threadIdx.x == 0 && g_xb_trace && blockIdx.y == 0) {
            p = g_xb_trace + 4 * (size_t)(base + blockIdx.x);
            unsigned hwid, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            p[2] = hwid; p[3] = xcc;
            p[0] = wall_clock64();
        }
    }
    __device__ ~XbTrace() { if (p) p[1] = wall_clock64(); }       // thread 0 leaves the kernel: its last store has been issued (not drained)
};
#endif
// Row groups of the x passes: NB - 1 groups of C map rows and the group of row N/2, which with C - 1 padding rows would run C transforms
// for one useful row and -- as the (NB)th workgroup on 256 CUs -- close the launch (per-workgroup trace, profiles/r03_xpass_trace.txt:
// the 257th workgroup of the displacement pass is the only one that shares a CU).  From 1024 up that group runs ONE transform per
// role (the same engine with a single interleaved column: a quarter of the loads, butterflies and stores), in all three roles.
template <int N, int C> constexpr bool xpass_single_row_group() { return C > 1 && N >= 1024 && (N / 2) % (2 * C) == 0; }
// HEIGHT workgroups (2 C rows each: C transforms of two real rows): the groups that hold a valid row, 0 .. N/2 (round 3 also ran the
// all-padding group behind them)
template <int N, int C> constexpr int xpass_height_groups() { return (N / 2 + 1 + 2 * C - 1) / (2 * C); }

// Completion record of a frame.  The workgroup of the frame's last launch (the displacement pass; the normal-map role in the split frame order) that finishes LAST hands, per tile, one 16-byte record
// (min key, max key, frame sequence number, 0) to host-coherent memory: a synchronous ComputeWaves returns from a short poll of
// those words instead of a stream synchronisation (ocean_compute_waves; 13-16 us of wake-up per call at the reference's call
// shape, WaterSurfaceMesh.cpp:145-154).  A record is one store instruction of one lane -- the host never sees half of one -- and
// carries its own sequence number, so nothing depends on the order in which records arrive.  It tells the host that the frame's
// work is done; it is not a memory fence: whatever reads the maps is ordered by the stream, as before.
// Counting is two-level -- workgroup -> one of up to DONE_GROUPS group counters (a cache line each) -> the top counter -- because
// the workgroups of a round finish together and a single word takes ~88 atomics per microsecond (257 of them: +2 us on the 2048^2
// displacement pass; two-level: see DESIGN.md section 6).
constexpr unsigned DONE_GROUPS = 1024, DONE_STRIDE = 16;      // counter g at done_ctr[(1 + g) * DONE_STRIDE], the top one at [0]
// (total / id: the workgroups that count and this one's index among them -- the whole grid by default; k_frame's z-pass workgroups do not count)
template <int T>
__device__ __forceinline__ void frame_done(const FrameArgs& a, unsigned* lds_flag, int tid, unsigned total = 0, unsigned id = 0)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's map stores have been taken
    __syncthreads();                                        // ... every wave's; nobody reads the FFT image any more
    if (tid == 0) {
        if (total == 0) { total = gridDim.x * gridDim.y; id = blockIdx.y * gridDim.x + blockIdx.x; }
        const unsigned groups = total / 16u < DONE_GROUPS ? (total + 15u) / 16u : DONE_GROUPS;
        const unsigned g = id % groups, members = total / groups + (g < total % groups ? 1u : 0u);
        unsigned* gc = a.done_ctr + (1u + g) * DONE_STRIDE;
        bool last = false;
        if (atomicAdd(gc, 1u) == members - 1u) {             // last of its group: the group counter is free again, one add upstairs
            *gc = 0u;
            last = atomicAdd(a.done_ctr, 1u) == groups - 1u;
        }
        lds_flag[0] = last;
    }
    __syncthreads();
    if (!lds_flag[0]) return;
    if (tid == 0) *a.done_ctr = 0u;                         // for the chain's next frame (stream order)
    for (unsigned i = (unsigned)tid; i < gridDim.y; i += (unsigned)T)
        a.done_rec[i] = make_uint4(load_wt(a.minmax + 2 * i + 0), load_wt(a.minmax + 2 * i + 1), a.frame_seq, 0u);     // one 16-byte store (the keys past the
                                                                                        // L1 / this XCD's L2: in a merged launch this very launch's atomics wrote them)
}


// The records of a launch (FrameArgs::rec_mode): the early form by the first workgroup of each tile (the height keys are final when the
// frame's last launches run), or the counted form.  Called by every thread of every workgroup of the launch, behind its work.
template <int T>
__device__ __forceinline__ void frame_records(const FrameArgs& a, unsigned* lds_flag, int tid, bool first_of_tile, unsigned total = 0, unsigned id = 0)
{
    if (a.rec_mode == 1) {
        if (first_of_tile && tid == 0) a.done_rec[blockIdx.y] = make_uint4(load_wt(a.minmax + 2 * blockIdx.y + 0), load_wt(a.minmax + 2 * blockIdx.y + 1), a.frame_seq, 0u);
    } else if (a.rec_mode == 2) {
        frame_done<T>(a, lds_flag, tid, total, id);
    }
}

// The body of k_xpass_b as a device function (k_frame runs it for the workgroups behind its z-pass ones).  bx_in: the workgroup's index among
// the launch's x-axis workgroups; ONE: part of a one-launch frame -- all three roles (HEIGHT, NORMAL, DISP), every workgroup first waits until
// the tile's z-pass workgroups of the same launch have counted themselves in (FrameArgs::zdone; they are dispatched first and wait for nobody),
// and the records' workgroup count leaves the z-pass workgroups out (rec_total / rec_id).
template <int N, int C, int T, class P, bool NTS, bool Z16, bool JAC, bool ONE = false>
__device__ __forceinline__ void xpass_b_body(const FrameArgs& a, unsigned char* smem, const int bx_in, const unsigned rec_total = 0, const unsigned rec_id = 0)
{
    using HF = Half<N>;
    c32* fbuf = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int tile = blockIdx.y;
#ifdef OCEAN_XB_TRACE
    XbTrace xb_trace_;
#endif
    constexpr int LM = 1;                                 // last-stage lane layout (fft_engine.h): a wave = one map row, 1 KiB bursts
    TwiddleRegs<N, C, T, P, LM> twr;
    twr.load(a.tw, tid);
    constexpr int NB = (HF::NU + C - 1) / C;              // normal workgroups
    constexpr int HB = JAC ? NB : xpass_height_groups<N, C>();      // height workgroups
    static_assert(HF::NUP % (2 * C) == 0, "height row blocks");
    // a launch holds the HEIGHT workgroups, the NORMAL workgroups or (usually) both: a.xb_roles; bx = the index in the full grid
    const int bx = bx_in + (a.xb_roles == 2 ? HB : 0);
    if constexpr (ONE) wait_counter_reached(a.zdone + tile, a.zdone_target, tid, a.poll_sleep, a.fault);      // the intermediates of THIS frame are all written (write-through)

    // (the roles are lambdas: every workgroup of a launch, whatever its role, ends in frame_records)
    auto pair3_role = [&]() {
    if constexpr (JAC) {
        // ---- PAIR-3 workgroup (OCEAN_MODE_JACOBIAN): the height travels as the real part of pair 3 with the cross
        // derivative as its imaginary part, so C rows per workgroup like every pair (not 2 C real rows): raw signed
        // height and cross derivative of rows u0 .. u0+C-1 out (both even: the mirrored rows hold the same values),
        // global min/max of the height.
        {
            constexpr int NW = (T + 63) / 64;
            float* red = reinterpret_cast<float*>(fbuf + fft_lds_elems<N, C>());
            const int u0 = xcd_swizzle(bx, HB) * C;
            const float2* __restrict__ z3 = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(a.z3) + (size_t)tile * HF::Z_GROUP * (Z16 ? 4 : 8));
            float* __restrict__ hraw = a.hraw + (size_t)tile * HF::HRAW_TILE;
            float* __restrict__ jraw = a.jraw + (size_t)tile * HF::HRAW_TILE;
            [[maybe_unused]] float ux = 1.0f, uy = 1.0f;
            float ig = 1.0f;                                    // the cross derivative went in amplified by g (zscale): out comes g times it
            { const float4 zs3 = a.zscale[2 * tile + 1]; ig = zs3.w; if constexpr (Z16) { ux = zs3.z; uy = zs3.z; } }
            float vmin = 3.402823466e+38f, vmax = -3.402823466e+38f;
            auto in = [&](int nf, int c, int, int) -> c32 { return load_pair_column<N, Z16>(z3, nf, u0 + c, 1.0f, ux, uy); };
            auto out = [&](int p, int c, c32 v, int, int) {
                const int q = u0 + c;
                if (q > N / 2) return;                                  // padding row
                const float s = ((p + q) & 1) ? -1.0f : 1.0f;
                const float ha = s * v.x;
                vmin = fminf(vmin, ha); vmax = fmaxf(vmax, ha);
                at32(hraw, hraw_index(N, p, q)) = ha;
                at32(jraw, hraw_index(N, p, q)) = s * v.y * ig;
            };
            batch_fft<N, C, T, P>(fbuf, twr, tid, in, out);
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
    }
    };
    // ---- HEIGHT workgroup: C transforms of two real rows each, raw signed heights out, global min / max by atomics.  In a merged x pass
    // (xb_roles bit 2) the DISP workgroups of the same launch read those rows and the final min / max: the rows are stored write-through
    // and the workgroup announces itself on a.hdone behind them (store_wt / wait_counter above).
    auto height_role = [&]() {
        const bool merged = (a.xb_roles & 4) != 0;
        constexpr int NW = (T + 63) / 64;
        float* red = reinterpret_cast<float*>(fbuf + fft_lds_elems<N, C>());
        const int u0 = xcd_swizzle(bx, HB) * 2 * C;
        constexpr size_t ES = Z16 ? 4 : 8;
        const float2* __restrict__ zh = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(a.zh) + (size_t)tile * HF::ZH_TILE * ES);
        float* __restrict__ hraw = a.hraw + (size_t)tile * HF::HRAW_TILE;
        float vmin = 3.402823466e+38f, vmax = -3.402823466e+38f;
        [[maybe_unused]] const float uu = Z16 ? a.zscale[2 * tile].z : 1.0f;
        auto in = [&](int nf, int c, int, int) -> c32 {
            const int row = nf <= N / 2 ? nf : N - nf;
            float4 z;
            if constexpr (Z16) {      // two half2 units in one 8-byte load
                const float2 raw2 = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(zh) + HF::template zhidx<Z16>(row, u0 + 2 * c) * 4u);
                __half2 ha, hb;
                __builtin_memcpy(&ha, &raw2.x, 4); __builtin_memcpy(&hb, &raw2.y, 4);
                const float2 fa = __half22float2(ha), fb = __half22float2(hb);
                z = make_float4(fa.x * uu, fa.y * uu, fb.x * uu, fb.y * uu);
            } else {
                z = *reinterpret_cast<const float4*>(&at32(zh, HF::template zhidx<Z16>(row, u0 + 2 * c)));
            }
            if (nf == 0 || nf == N / 2) return make_float2(z.x, z.z);
            if (nf < N / 2) return make_float2(z.x - z.w, z.y + z.z);
            return make_float2(z.x + z.w, z.z - z.y);
        };
        auto out = [&](int p, int c, c32 v, int, int) {
            const int u = u0 + 2 * c;
            const float s = ((p + u) & 1) ? -1.0f : 1.0f;
            const float ha = s * v.x, hb = -s * v.y;
            if (u <= N / 2) { vmin = fminf(vmin, ha); vmax = fmaxf(vmax, ha); }
            if (u + 1 <= N / 2) { vmin = fminf(vmin, hb); vmax = fmaxf(vmax, hb); }
            if (merged) {
                store_wt(&at32(hraw, hraw_index(N, p, u)), ha);
                store_wt(&at32(hraw, hraw_index(N, p, u + 1)), hb);
            } else {
                at32(hraw, hraw_index(N, p, u)) = ha;
                at32(hraw, hraw_index(N, p, u + 1)) = hb;
            }
        };
        // the group of row N/2 holds one valid row: ONE transform (rows N/2 and N/2 + 1) instead of C
        if constexpr (xpass_single_row_group<N, C>()) {
            if (u0 == N / 2) {
                TwiddleRegs<N, 1, T, P, LM> tw1;
                tw1.load(a.tw, tid);
                batch_fft<N, 1, T, P>(fbuf, tw1, tid, in, out);
            } else {
                batch_fft<N, C, T, P>(fbuf, twr, tid, in, out);
            }
        } else {
            batch_fft<N, C, T, P>(fbuf, twr, tid, in, out);
        }
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
        if (merged) {
            // every wave's write-through stores (and lane 0's two atomics) have been taken, then ONE lane counts the workgroup in
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(a.hdone + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // ---- NORMAL workgroup ------------------------------------------------------------
    // (writing the two halves of a texel separately -- 8-byte stores, no registers held
    // between the transforms -- was measured 50% slower: the partial lines do not meet in L2)
    using LS = LastStage<N, C, T, P, LM>;
    // Row groups: NB - 1 full ones (C map rows each, XCD-swizzled among themselves) and the group of row N/2 with C - 1 padding rows, which goes
    // to the FIRST NORMAL workgroup.  Where the plain swizzle put it (block 385 of 387 at 2048^2) it ran as the younger of two workgroups on
    // a CU, and on some hardware queues that one workgroup took 18-23 instead of 13 us and the kernel 22.5-30 instead of 21 us (per-workgroup
    // trace, profiles/r03_xpass_trace.txt); dispatched first it has its CU's issue slots to itself (0 slow processes in 50 against 1 in 5 on the
    // same box, interleaved) and costs nothing anywhere else.
    // (From 2048 up, where a tile's workgroups outnumber the CUs; at 512^2 and 1024^2 -- every workgroup alone on a CU -- the plain swizzle is
    // 0.3-0.5 us faster and stays.)
    // (the role's body as a lambda: its reduced modes leave early, and in the split frame order -- where this is the frame's LAST launch --
    //  every thread of the workgroup must still reach frame_records below)
    auto normal_role = [&]() {
    const int nid = bx - HB;
    const int u0 = (N >= 2048 ? (nid == 0 ? NB - 1 : xcd_swizzle(nid - 1, NB - 1)) : xcd_swizzle(nid, NB)) * C;
    start_ramp_wait(a.start_ramp, (unsigned)nid, (unsigned)NB);    // the NORMAL workgroups alone (one 2048^2 tile: ocean_launch.h)
    constexpr size_t ESN = Z16 ? 4 : 8;
    const float2* __restrict__ z1 = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(a.z) + ((size_t)tile * HF::Z_TILE + HF::Z_GROUP) * ESN);
    const float2* __restrict__ z2 = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(z1) + HF::Z_GROUP * ESN);
    [[maybe_unused]] const float uk = Z16 ? a.zscale[2 * tile].w : 1.0f;
    float4* __restrict__ nrm = a.nrm + (size_t)tile * N * N;
    float4* __restrict__ nrm_host = a.nrm_host ? a.nrm_host + (size_t)tile * N * N : nullptr;
    [[maybe_unused]] float* __restrict__ jac0 = JAC ? a.jac0 + (size_t)tile * HF::HRAW_TILE : nullptr;
    [[maybe_unused]] const float lambda = a.lambda ? a.lambda[tile] : a.lambda_all;
    c32 held[LS::IT][LS::RL];
    auto emit = [&](int p, int c, c32 slopes, c32 derivs) {
#pragma clang fp contract(off)          // every instantiation of this kernel rounds alike: frames are bit-identical whichever one a launch picks
        const int q = u0 + c;
        if (q > N / 2) return;                                      // padding row
        const float s = ((p + q) & 1) ? -1.0f : 1.0f;
        // (slope x, slope z, dDx/dx, dDz/dz) * sign   (.cpp:430-435)
        const float4 o = make_float4(s * slopes.x, s * slopes.y, s * derivs.x, s * derivs.y);
        if constexpr (JAC)          // (1 + lambda s dxDx)(1 + lambda s dzDz) of .cpp:423-425, finished by the displacement pass
            at32(jac0, hraw_index(N, p, q)) = (1.0f + lambda * o.z) * (1.0f + lambda * o.w);
        OCEAN_STORE(nrm, q * N + p, o);
        store_map_host(nrm_host, (unsigned)(q * N + p), o);
        if (q != 0 && q != N / 2) {                                  // mirror: slopes odd, derivatives even
            OCEAN_STORE(nrm, (N - q) * N + ((N - p) & (N - 1)), make_float4(-o.x, -o.y, o.z, o.w));
            store_map_host(nrm_host, (unsigned)((N - q) * N + ((N - p) & (N - 1))), make_float4(-o.x, -o.y, o.z, o.w));
        }
    };
    const c32 zero = make_float2(0.0f, 0.0f);
    if constexpr (xpass_single_row_group<N, C>()) {
        if (u0 == N / 2) {           // the group of row N/2: one useful row, one interleaved column per transform
            using LS1 = LastStage<N, 1, T, P, LM>;
            if (a.mode == 2) {
                for_each_output<LS1, T>(tid, [&](int p, int c, int, int) { emit(p, c, zero, zero); });
                return;
            }
            TwiddleRegs<N, 1, T, P, LM> tw1;
            tw1.load(a.tw, tid);
            c32 held1[LS1::IT][LS1::RL];
            {
                auto in = [&](int nf, int c, int, int) -> c32 { return load_pair_column<N, Z16>(z1, nf, u0 + c, -1.0f, uk); };
                auto out = [&](int, int, c32 v, int u, int i) { held1[u][i] = v; };
                batch_fft<N, 1, T, P>(fbuf, tw1, tid, in, out);
            }
            if (a.mode == 1) {
                for_each_output<LS1, T>(tid, [&](int p, int c, int u, int i) { emit(p, c, held1[u][i], zero); });
                return;
            }
            __builtin_amdgcn_sched_barrier(0);
            auto in = [&](int nf, int c, int, int) -> c32 { return load_pair_column<N, Z16>(z2, nf, u0 + c, 1.0f, uk); };
            auto out = [&](int p, int c, c32 v, int u, int i) { emit(p, c, held1[u][i], v); };
            batch_fft<N, 1, T, P>(fbuf, tw1, tid, in, out);
            return;
        }
    }
    if (a.mode == 2) {               // HEIGHT1: the normal map is all zero
        for_each_output<LS, T>(tid, [&](int p, int c, int, int) { emit(p, c, zero, zero); });
        return;
    }
    // From 2048 up this role runs one workgroup per CU whatever it does below 256 VGPRs, so the
    // inputs of the second transform are fetched into registers right behind those of the
    // first and travel while the first is computed.  Below that the registers are worth more
    // as a second workgroup per CU.
    constexpr bool PREFETCH = N >= 2048;
    using FS = FirstStage<N, C, T, P>;
    [[maybe_unused]] c32 xb[PREFETCH ? FS::IT : 1][PREFETCH ? FS::R0 : 1];
    auto fetch = [&](const float2* __restrict__ zg, float eps, auto& dst) {
#pragma unroll
        for (int u = 0; u < FS::IT; ++u) {
            const int w = tid + u * T;
            if (!FS::GUARD || w < FS::ITEMS) {
                const int c = w % C, j = w / C;
#pragma unroll
                for (int i = 0; i < FS::R0; ++i) dst[u][i] = load_pair_column<N, Z16>(zg, j + i * FS::STRIDE, u0 + c, eps, uk);
            }
        }
    };
    if constexpr (PREFETCH) {
        c32 xa[FS::IT][FS::R0];
        fetch(z1, -1.0f, xa);
        if (a.mode != 1) fetch(z2, 1.0f, xb);
        auto in = [&](int, int, int u, int i) -> c32 { return xa[u][i]; };
        auto out = [&](int, int, c32 v, int u, int i) { held[u][i] = v; };
        batch_fft<N, C, T, P>(fbuf, twr, tid, in, out);
    } else {
        auto in = [&](int nf, int c, int, int) -> c32 { return load_pair_column<N, Z16>(z1, nf, u0 + c, -1.0f, uk); };
        auto out = [&](int, int, c32 v, int u, int i) { held[u][i] = v; };
        batch_fft<N, C, T, P>(fbuf, twr, tid, in, out);
    }
    if (a.mode == 1) {               // CHOPPY5: slopes only
        for_each_output<LS, T>(tid, [&](int p, int c, int u, int i) { emit(p, c, held[u][i], zero); });
        return;
    }
    if constexpr (PREFETCH) {
        auto in = [&](int, int, int u, int i) -> c32 { return xb[u][i]; };
        auto out = [&](int p, int c, c32 v, int u, int i) { emit(p, c, held[u][i], v); };
        batch_fft<N, C, T, P>(fbuf, twr, tid, in, out);
    } else {
        // keep the second transform's loads from being hoisted over the first one's last
        // stage: that costs ~45 VGPRs and with them the second workgroup per CU
        __builtin_amdgcn_sched_barrier(0);
        auto in = [&](int nf, int c, int, int) -> c32 { return load_pair_column<N, Z16>(z2, nf, u0 + c, 1.0f, uk); };
        auto out = [&](int p, int c, c32 v, int u, int i) { emit(p, c, held[u][i], v); };
        batch_fft<N, C, T, P>(fbuf, twr, tid, in, out);
    }
    };
    // ---- DISP workgroup of a merged x pass (xb_roles bit 2; never in the Jacobian mode): pair 0 -> displacement-map rows u0 .. u0+C-1, like
    // k_xpass_disp, in the SAME launch as the HEIGHT workgroups whose rows and min / max it needs.  It transforms at once and holds the
    // results; only then does it wait for the tile's HEIGHT workgroups (dispatched before it: lower block indices, and they wait for
    // nobody, so the wait cannot deadlock whatever is resident), reads its raw heights and the final keys write-through, and stores.
    // The arithmetic per texel is k_xpass_disp's: the same bits.
    auto disp_role = [&]() {
        if constexpr (!JAC) {
            using LSD = LastStage<N, C, T, P, LM>;
            const int u0 = xcd_swizzle(bx - HB - NB, NB) * C;
            const float2* __restrict__ z0 = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(a.z) + (size_t)tile * HF::Z_TILE * (Z16 ? 4 : 8));
            [[maybe_unused]] const float uu = Z16 ? a.zscale[2 * tile].z : 1.0f;
            const float* __restrict__ hraw = a.hraw + (size_t)tile * HF::HRAW_TILE;
            float4* __restrict__ disp = a.disp + (size_t)tile * N * N;
            float4* __restrict__ disp_host = a.disp_host ? a.disp_host + (size_t)tile * N * N : nullptr;
            c32 dheld[LSD::IT][LSD::RL];
            if (a.mode != 2) {
                auto in = [&](int nf, int c, int, int) -> c32 { return load_pair_column<N, Z16>(z0, nf, u0 + c, -1.0f, uu); };
                auto out = [&](int, int, c32 v, int u, int i) { dheld[u][i] = v; };
                batch_fft<N, C, T, P>(fbuf, twr, tid, in, out);
            } else {
#pragma unroll
                for (int u = 0; u < LSD::IT; ++u)
#pragma unroll
                    for (int i = 0; i < LSD::RL; ++i) dheld[u][i] = make_float2(0.0f, 0.0f);
            }
            wait_counter(a.hdone + tile, (unsigned)HB, tid, a.poll_sleep, a.fault);
            const unsigned kmn = load_wt(a.minmax + 2 * tile + 0), kmx = load_wt(a.minmax + 2 * tile + 1);      // final: every HEIGHT workgroup has counted itself in
            const float mn = key_float(kmn), mx = key_float(kmx);
            const float inv_a = 1.0f / fmaxf(fabsf(mn), fabsf(mx));
            const float lambda = a.lambda ? a.lambda[tile] : a.lambda_all;
            for_each_output<LSD, T>(tid, [&](int p, int c, int u, int i) {
#pragma clang fp contract(off)          // as in k_xpass_disp
                const int q = u0 + c;
                if (q > N / 2) return;
                const float hv = load_wt(&at32(hraw, hraw_index(N, p, q)));
                const float sg = ((p + q) & 1) ? -1.0f : 1.0f;
                const c32 v = dheld[u][i];
                const float4 o = make_float4(sg * lambda * v.x, hv * inv_a, sg * lambda * v.y, 1.0f);
                store_map<NTS>(disp, (unsigned)(q * N + p), o);
                store_map_host(disp_host, (unsigned)(q * N + p), o);
                if (q != 0 && q != N / 2) {     // mirror: the displacements are odd, the height even
                    store_map<NTS>(disp, (unsigned)((N - q) * N + ((N - p) & (N - 1))), make_float4(-o.x, o.y, -o.z, 1.0f));
                    store_map_host(disp_host, (unsigned)((N - q) * N + ((N - p) & (N - 1))), make_float4(-o.x, o.y, -o.z, 1.0f));
                }
            });
        }
    };
    const bool merged_launch = (a.xb_roles & 4) != 0;
    if (bx < HB) { if constexpr (JAC) pair3_role(); else height_role(); }
    else if (!JAC && merged_launch && bx >= HB + NB) disp_role();
    else normal_role();
    // the launch's records: the early form needs the final height keys -- the first NORMAL workgroup of a launch behind the HEIGHT
    // workgroups' launch (split order), the first DISP workgroup of a merged launch (it has waited for them)
    frame_records<T>(a, reinterpret_cast<unsigned*>(smem), tid, merged_launch ? bx == HB + NB : bx == HB, rec_total, rec_id);
}

template <int N, int C, int T, class P = Plan<N>, bool NTS = false, bool Z16 = false, bool JAC = false>
__global__ void __launch_bounds__(T) k_xpass_b(const FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    xpass_b_body<N, C, T, P, NTS, Z16, JAC>(a, smem, (int)blockIdx.x);
}

// ============================================================================
// k_frame: a whole frame in ONE launch (round 5) -- for PIPELINED frames of one tile up to 128^2, which are bound by the rate at which launches
// get through the process's hardware queues (profiles/r05_small_tile_experiments.txt; from 256^2 up the form loses and is not instantiated).  Grid = [N/2 + 1 z-pass workgroups | HEIGHT | NORMAL | DISP];
// every dependency is one-way and points to LOWER block indices -- the x-axis workgroups wait for the tile's z-pass workgroups (zdone), the DISP
// workgroups for the HEIGHT workgroups (hdone) -- and the producers wait for nobody, so nothing can deadlock whatever is resident.  What crosses
// workgroups inside the launch is stored write-through (`sc1`: the intermediates, the raw heights, the resets of the per-tile words), announced
// by ONE lane's agent-scope atomic add behind every storing wave's s_waitcnt vmcnt(0) and the workgroup's barrier, and awaited by one lane's
// `sc1` poll + a barrier (MI355X_MICROARCH.md, inter-workgroup visibility: the measured recipe); the L1 / L2 of the reading side hold none of
// those lines (invalidated at the launch's start, first touched behind the wait).  The bodies are k_zpass's and k_xpass_b's: the same bits.
// ============================================================================
template <int N, int C, int T, class PZ, class PX, bool NTS, bool Z16, bool FAST>
__global__ void __launch_bounds__(T) k_frame(const FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using HF = Half<N>;
    constexpr int NZ = N / 2 + 1;
    constexpr int NB = (HF::NU + C - 1) / C, HB = xpass_height_groups<N, C>();
    const int bx = (int)blockIdx.x;
    if (bx < NZ) {
        zpass_body<N, T, PZ, false, Z16, 1, FAST, true>(a, smem, bx, NZ);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave's write-through stores have been taken ...
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(a.zdone + blockIdx.y, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ... then the workgroup counts itself in
        return;
    }
    xpass_b_body<N, C, T, PX, NTS, Z16, false, true>(a, smem, bx - NZ, (unsigned)(HB + 2 * NB) * gridDim.y, (unsigned)blockIdx.y * (unsigned)(HB + 2 * NB) + (unsigned)(bx - NZ));
}

template <int N, int C, int T, class P = Plan<N>, bool NTS = false, bool Z16 = false, bool JAC = false>
__global__ void __launch_bounds__(T, (T == 512 ? 4 : 1)) k_xpass_disp(const FrameArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using HF = Half<N>;
    constexpr int LM = 1;                                 // as in k_xpass_b
    c32* fbuf = reinterpret_cast<c32*>(smem);
    const int tid = threadIdx.x;
    const int tile = blockIdx.y;
#ifdef OCEAN_XB_TRACE
    XbTrace xd_trace_(512);
#endif
    constexpr int NB = (HF::NU + C - 1) / C;
    const int u0 = xcd_swizzle(blockIdx.x, NB) * C;
    start_ramp_wait(a.start_ramp, blockIdx.x, gridDim.x);   // (one 2048^2 tile: ocean_launch.h)
    const float2* __restrict__ z0 = reinterpret_cast<const float2*>(reinterpret_cast<const char*>(a.z) + (size_t)tile * HF::Z_TILE * (Z16 ? 4 : 8));
    [[maybe_unused]] const float uu = Z16 ? a.zscale[2 * tile].z : 1.0f;
    const float* __restrict__ hraw = a.hraw + (size_t)tile * HF::HRAW_TILE;
    float4* __restrict__ disp = a.disp + (size_t)tile * N * N;
    float4* __restrict__ disp_host = a.disp_host ? a.disp_host + (size_t)tile * N * N : nullptr;
    // CC = map rows of this workgroup's group (C, or 1 for the group of row N/2)
    auto rows = [&](auto cc_tag) {
        constexpr int CC = decltype(cc_tag)::value;
        using LS = LastStage<N, CC, T, P, LM>;
        TwiddleRegs<N, CC, T, P, LM> twr;
        twr.load(a.tw, tid);
        float hv[LS::IT][LS::RL];
        [[maybe_unused]] float jv[JAC ? LS::IT : 1][JAC ? LS::RL : 1], j0[JAC ? LS::IT : 1][JAC ? LS::RL : 1];
#pragma unroll
        for (int u = 0; u < LS::IT; ++u) {
            const int w = tid + u * T;
            if (!LS::GUARD || w < LS::ITEMS) {
                int c, j;
                LS::map(w, c, j);
#pragma unroll
                for (int i = 0; i < LS::RL; ++i) {
                    hv[u][i] = at32(hraw, hraw_index(N, j + i * LS::STRIDE, u0 + c));
                    if constexpr (JAC) {
                        jv[u][i] = at32(a.jraw + (size_t)tile * HF::HRAW_TILE, hraw_index(N, j + i * LS::STRIDE, u0 + c));
                        j0[u][i] = at32(a.jac0 + (size_t)tile * HF::HRAW_TILE, hraw_index(N, j + i * LS::STRIDE, u0 + c));
                    }
                }
            }
        }
        const unsigned kmn = a.minmax[2 * tile + 0], kmx = a.minmax[2 * tile + 1];      // final by now
        if (a.rec_mode == 1 && blockIdx.x == 0 && tid == 0) a.done_rec[tile] = make_uint4(kmn, kmx, a.frame_seq, 0u);
        const float mn = key_float(kmn);
        const float mx = key_float(kmx);
        const float inv_a = 1.0f / fmaxf(fabsf(mn), fabsf(mx));
        const float lambda = a.lambda ? a.lambda[tile] : a.lambda_all;
        auto in = [&](int nf, int c, int, int) -> c32 { return load_pair_column<N, Z16>(z0, nf, u0 + c, -1.0f, uu); };
        auto out = [&](int p, int c, c32 v, int u, int i) {
#pragma clang fp contract(off)          // as in k_xpass_b
            const int q = u0 + c;
            if (q > N / 2) return;
            const float s = ((p + q) & 1) ? -1.0f : 1.0f;
            float w = 1.0f;
            if constexpr (JAC) {        // (1 + l s dxDx)(1 + l s dzDz) - (l s dxDz)(l s dzDx), .cpp:422-426 (the two cross terms are one field)
                const float cross = lambda * jv[u][i];
                w = j0[u][i] - cross * cross;
            }
            const float4 o = make_float4(s * lambda * v.x, hv[u][i] * inv_a, s * lambda * v.y, w);
            OCEAN_STORE(disp, q * N + p, o);
            store_map_host(disp_host, (unsigned)(q * N + p), o);
            if (q != 0 && q != N / 2) {     // mirror: the displacements are odd, height and Jacobian even
                OCEAN_STORE(disp, (N - q) * N + ((N - p) & (N - 1)), make_float4(-o.x, o.y, -o.z, w));
                store_map_host(disp_host, (unsigned)((N - q) * N + ((N - p) & (N - 1))), make_float4(-o.x, o.y, -o.z, w));
            }
        };
        if (a.mode == 2)                 // HEIGHT1: no horizontal displacement, no transform
            for_each_output<LS, T>(tid, [&](int p, int c, int u, int i) { out(p, c, make_float2(0.0f, 0.0f), u, i); });
        else
            batch_fft<N, CC, T, P>(fbuf, twr, tid, in, out);
    };
    if constexpr (xpass_single_row_group<N, C>()) {
        if (u0 == N / 2) rows(std::integral_constant<int, 1>{});
        else rows(std::integral_constant<int, C>{});
    } else {
        rows(std::integral_constant<int, C>{});
    }
    if (a.rec_mode == 2) frame_done<T>(a, reinterpret_cast<unsigned*>(smem), tid);
}


// ---- per-size launch geometry ---------------------------------------------------
template <int N> struct Geo;
#define OCEAN_R(...) Radices<__VA_ARGS__>
#define OCEAN_GEO(n, tr, pr, cc, tc, pc)                                                       \
    template <> struct Geo<n> {                                                                \
        static constexpr int T_ROWS = tr;               /* threads of k_zpass               */ \
        static constexpr int CC = cc, T_C = tc;         /* x pass: rows per workgroup, threads */ \
        using PR = pr; using PC = pc;                   /* radix plans: z pass, x pass      */ \
    };
OCEAN_GEO(16, 64, Plan<16>, 4, 64, Plan<16>)
OCEAN_GEO(32, 64, Plan<32>, 4, 64, Plan<32>)
OCEAN_GEO(64, 64, Plan<64>, 4, 64, Plan<64>)
OCEAN_GEO(128, 64, Plan<128>, 4, 64, Plan<128>)
OCEAN_GEO(256, 64, Plan<256>, 4, 64, Plan<256>)
// (x pass at 512^2 with twice the threads per transform -- 512 threads, radix 4.4.4.8 -- or with two rows per workgroup and twice the
// workgroups: both within the noise of this form, 16.5-17.5 us per serial frame; profiles/r03_small_tile_experiments.txt)
OCEAN_GEO(512, (zpass_columns<512>() == 4 ? 256 : 128), Plan<512>, 4, 256, Plan<512>)
// from 1024 up the z pass runs radix-8 butterflies with twice the threads (one more LDS
// exchange, about half the VGPRs): -9 % at 1024, -1.5 % at 2048, -4.5 % at 4096
OCEAN_GEO(1024, 256, OCEAN_R(8, 8, 4, 4), 4, 256, Plan<1024>)
// (x pass with two rows per workgroup and 256 threads -- 106 VGPRs, four workgroups per CU instead of one -- is 15 % slower at 2048
// and 1024: 16- instead of 32-byte pieces of the intermediates; profiles/r02_layout_experiments.txt)
// (256 threads with two butterflies per thread and stage -- 124 VGPRs, and with the S+ / kz tables out of LDS four workgroups per
// CU, i.e. 1024 of the 1025 columns in ONE round -- is no faster: 26.4-26.7 us for 1024 workgroups against 25.2-26.5 with 512 threads
// at three per CU; the z pass is bound by what a CU gets through, not by the round structure: profiles/r03_zpass_experiments.txt)
OCEAN_GEO(2048, 512, OCEAN_R(8, 8, 8, 4), 4, 512, Plan<2048>)
OCEAN_GEO(4096, 1024, OCEAN_R(8, 8, 8, 8), 2, 512, Plan<4096>)
#undef OCEAN_GEO

// Threads and radix plan of the single-transform z pass (k_zpass_c1): one first-stage butterfly per thread.  The shipped form runs the tile
// size's radix-8 plan with N / 8 threads; developer builds can run the radix-16 plan of fft_engine.h with N / 16 threads from a tile size up
// (-DOCEAN_C1_R16_MIN=2048: one exchange and two barriers less per transform, a quarter fewer twiddle products, half the waves;
// profiles/r06_zpass_experiments.txt).
#ifndef OCEAN_C1_R16_MIN
#define OCEAN_C1_R16_MIN 8192
#endif
template <int N> constexpr bool zpass_c1_r16() { return N >= OCEAN_C1_R16_MIN && N >= 2048; }
template <int N> constexpr int zpass_c1_threads() { return zpass_c1_r16<N>() ? N / 16 : N / 8; }
template <int N> struct C1Plan { using type = std::conditional_t<zpass_c1_r16<N>(), Plan<N>, typename Geo<N>::PR>; };

}  // namespace ocean
