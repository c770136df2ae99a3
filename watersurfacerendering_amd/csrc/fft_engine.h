// fft_engine.h -- block-level batched complex FFT for gfx950 (CDNA4), fp32.
//
// Computes C interleaved length-N unnormalised BACKWARD DFTs
//     B[X](q) = sum_n X(n) * exp(+2*pi*i*q*n/N)
// (the transform the reference obtains from fftwf_plan_dft_2d(..., FFTW_BACKWARD, ...),
//  /root/reference/src/scene/WSTessendorf.cpp:191-232, one axis at a time)
// with T threads of one workgroup.  Stockham autosort: every stage reads
// element (j + i*N/R) and writes element ((j-k)*R + k + i*Ns), so
//   * the FIRST stage takes its inputs from a functor (global loads or values
//     computed on the fly) -- nothing is staged before the first butterfly,
//   * the LAST stage hands its outputs, already in natural order and
//     lane-contiguous, to a functor (coalesced global stores / registers),
//   * only the (stages-1) exchanges in between go through LDS, in place.
// Butterflies are radix 4/8/16 held in VGPRs (64-wide waves, no cross-lane
// traffic); inter-stage twiddles come from one table-loaded base twiddle per
// butterfly raised to powers in log depth.
//
// LDS image: element (idx, c) of the batch lives at  (idx + idx/16) * C + c
// (float2 units).  The c-fastest interleave keeps every stage's reads
// lane-contiguous; the +idx/16 padding spreads the scattered writes of the
// expanding stages over all banks (ds_write_b64: 16-lane groups, 32 banks).
#pragma once
#include <hip/hip_runtime.h>

namespace ocean {

using c32 = float2;

// diagnostic builds only (-DOCEAN_STAMPS): per-workgroup clock stamps
#ifdef OCEAN_STAMPS
static __device__ unsigned long long* g_stamps = nullptr;
__device__ __forceinline__ void stamp(int k)
{
    if (threadIdx.x == 0 && g_stamps)
        g_stamps[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 32 + k] = clock64();
}
#define OCEAN_STAMP(k) ::ocean::stamp(k)
#else
#define OCEAN_STAMP(k) do {} while (0)
#endif

__device__ __forceinline__ c32 cmul(c32 a, c32 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// ---- packed-fp32 complex arithmetic ------------------------------------------------
// The butterflies run on 64-bit register pairs (re, im) with CDNA's packed fp32 VALU
// (v_pk_add/mul/fma_f32: two floats per lane per instruction).  The operand swizzles
// (op_sel) and per-half negation (neg_lo / neg_hi) of those instructions make a complex
// multiply two instructions and "b +/- i*a" one; written out here because the compiler
// reaches the same arithmetic only with 25-30 % extra v_mov / v_xor around it, and these
// kernels are VALU-issue bound (a wave64 VALU instruction occupies its SIMD for 4 cycles).
typedef float v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2 tov(c32 a) { return (v2){a.x, a.y}; }
__device__ __forceinline__ c32 toc(v2 a) { return make_float2(a.x, a.y); }

// (ONE asm statement for the pair: between two statements the compiler's hazard recogniser, which must assume that an inline-asm result may
//  carry a destination-select forwarding hazard, put an s_nop in front of the dependent second instruction -- 144 of them per wave and z-pass
//  column, 4 issue cycles each, for a hazard that full-register packed-fp32 results do not have; round 6, profiles/r06_zpass_experiments.txt)
__device__ __forceinline__ v2 pk_cmul(v2 a, v2 w)           // a * w
{
    v2 d;
#ifdef OCEAN_CMUL_SPLIT      // developer A/B: the two-statement form of rounds 2-5
    v2 t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));          // (a.x w.x, a.x w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"                 // (t.x - a.y w.y, t.y + a.y w.x)
        : "=v"(d) : "v"(a), "v"(w), "v"(t));
#else
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"                                   // (a.x w.x, a.x w.y)
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"                 // (t.x - a.y w.y, t.y + a.y w.x)
        : "=&v"(d) : "v"(a), "v"(w));
#endif
    return d;
}
__device__ __forceinline__ v2 pk_add_i(v2 b, v2 a)          // b + i a = (b.x - a.y, b.y + a.x)
{
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(b), "v"(a));
    return d;
}
__device__ __forceinline__ v2 pk_sub_i(v2 b, v2 a)          // b - i a = (b.x + a.y, b.y - a.x)
{
    v2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(b), "v"(a));
    return d;
}
__device__ __forceinline__ v2 pk_neg_add_i(v2 a)            // -a + i a = (-a.x - a.y, a.x - a.y)
{
    v2 d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,1] neg_hi:[1,0]" : "=v"(d) : "v"(a));
    return d;
}
__device__ __forceinline__ v2 pk_mul(v2 a, v2 b)            // (a.x b.x, a.y b.y) -- as an instruction of its own: a product the compiler sees
{                                                            // may or may not be fused with the butterfly's next add, instantiation by
    v2 d;                                                    // instantiation, and the variants of a kernel must agree bit for bit
    asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ v2 pk_muli(v2 a)                 // i a = (-a.y, a.x)
{
    v2 d;
    asm("v_pk_add_f32 %0, 0, %1 op_sel:[0,1] op_sel_hi:[0,0] neg_lo:[0,1]" : "=v"(d) : "v"(a));
    return d;
}

// ---- in-register radix-R backward DFTs, natural order in and out ----------
__device__ __forceinline__ void dft4(v2& x0, v2& x1, v2& x2, v2& x3)
{
    const v2 t0 = x0 + x2, t1 = x0 - x2;
    const v2 t2 = x1 + x3, d = x1 - x3;
    x0 = t0 + t2; x2 = t0 - t2;
    x1 = pk_add_i(t1, d); x3 = pk_sub_i(t1, d);
}

template <int R> struct Dft;

template <> struct Dft<2> {
    static __device__ __forceinline__ void run(v2 (&x)[2])
    {
        const v2 a = x[0], b = x[1];
        x[0] = a + b; x[1] = a - b;
    }
};

template <> struct Dft<4> {
    static __device__ __forceinline__ void run(v2 (&x)[4]) { dft4(x[0], x[1], x[2], x[3]); }
};

template <> struct Dft<8> {
    static __device__ __forceinline__ void run(v2 (&x)[8])
    {
        // decimation in time: E = DFT4(even), O = DFT4(odd), y[k] = E[k] + w8^k O[k]
        dft4(x[0], x[2], x[4], x[6]);
        dft4(x[1], x[3], x[5], x[7]);
        constexpr float h = 0.70710678118654752440f;
        const v2 hh = {h, h};
        const v2 o0 = x[1];
        const v2 a1 = pk_add_i(x[3], x[3]);          // (1+i) O1   (times h below)
        const v2 o2 = x[5];                          // times i, folded into the adds
        const v2 a3 = pk_neg_add_i(x[7]);            // (-1+i) O3  (times h below)
        const v2 e0 = x[0], e1 = x[2], e2 = x[4], e3 = x[6];
        x[0] = e0 + o0; x[4] = e0 - o0;
        x[1] = __builtin_elementwise_fma(a1, hh, e1); x[5] = __builtin_elementwise_fma(a1, -hh, e1);
        x[2] = pk_add_i(e2, o2); x[6] = pk_sub_i(e2, o2);
        x[3] = __builtin_elementwise_fma(a3, hh, e3); x[7] = __builtin_elementwise_fma(a3, -hh, e3);
    }
};

template <> struct Dft<16> {
    static __device__ __forceinline__ void run(v2 (&x)[16])
    {
        // n = 4a + b, k = k1 + 4 k2:
        //   Y[k1 + 4 k2] = sum_b w4^(b k2) * [ w16^(b k1) * sum_a x[4a + b] w4^(a k1) ]
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4(x[b], x[b + 4], x[b + 8], x[b + 12]);
        // now x[b + 4*k1] = u_b[k1]; multiply by w16^(b*k1)
        constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;   // w16^1
        constexpr float h = 0.70710678118654752440f;                                   // w16^2 = (h, h)
        const v2 hh = {h, h};
        const v2 w1 = {c1, s1}, w3 = {s1, c1}, w9 = {-c1, -s1};
        x[1 + 4] = pk_cmul(x[1 + 4], w1);                 // b=1,k1=1 : w^1
        x[1 + 8] = pk_mul(pk_add_i(x[1 + 8], x[1 + 8]), hh);     // b=1,k1=2 : w^2 = (1+i) h
        x[1 + 12] = pk_cmul(x[1 + 12], w3);               // b=1,k1=3 : w^3
        x[2 + 4] = pk_mul(pk_add_i(x[2 + 4], x[2 + 4]), hh);     // b=2,k1=1 : w^2
        x[2 + 8] = pk_muli(x[2 + 8]);                     // b=2,k1=2 : w^4 = i
        x[2 + 12] = pk_mul(pk_neg_add_i(x[2 + 12]), hh);         // b=2,k1=3 : w^6 = (-1+i) h
        x[3 + 4] = pk_cmul(x[3 + 4], w3);                 // b=3,k1=1 : w^3
        x[3 + 8] = pk_mul(pk_neg_add_i(x[3 + 8]), hh);           // b=3,k1=2 : w^6
        x[3 + 12] = pk_cmul(x[3 + 12], w9);               // b=3,k1=3 : w^9
        // outer DFT4 over b for each k1; result k2 lands at index k1 + 4*k2
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) dft4(x[4 * k1 + 0], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
        // x[4*k1 + k2] holds Y[k1 + 4*k2]  -> transpose the 4x4 index grid
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                const v2 t = x[4 * p + q]; x[4 * p + q] = x[4 * q + p]; x[4 * q + p] = t;
            }
    }
};

// x[i] *= w1^i.  Powers are built by squaring / pairwise products (at most four
// products deep for R = 16, ~2.4e-7 relative) and consumed as soon as they exist, so
// only w1..w(R/2) stay live (not all R-1 powers).
template <int R>
__device__ __forceinline__ void apply_twiddles(v2 (&x)[R], v2 w1)
{
    if constexpr (R == 2) {
        x[1] = pk_cmul(x[1], w1);
    } else if constexpr (R == 4) {
        const v2 w2 = pk_cmul(w1, w1);
        x[1] = pk_cmul(x[1], w1);
        x[2] = pk_cmul(x[2], w2);
        x[3] = pk_cmul(x[3], pk_cmul(w1, w2));
    } else {
        constexpr int H = R / 2;
        v2 pw[H + 1];
        pw[1] = w1;
        x[1] = pk_cmul(x[1], w1);
#pragma unroll
        for (int i = 2; i <= H; ++i) {
            pw[i] = pk_cmul(pw[i / 2], pw[i - i / 2]);
            x[i] = pk_cmul(x[i], pw[i]);
        }
#pragma unroll
        for (int i = 1; i < H; ++i) x[H + i] = pk_cmul(x[H + i], pk_cmul(pw[H], pw[i]));
    }
}

// ---- radix plans -------------------------------------------------------------
template <int... Rs> struct Radices {
    static constexpr int S = sizeof...(Rs);
    static constexpr int r[sizeof...(Rs)] = {Rs...};
    static constexpr int last = r[S - 1];
    static constexpr int product()
    {
        int p = 1;
        for (int i = 0; i < S; ++i) p *= r[i];
        return p;
    }
};

// default plan per transform length (large radices: fewest LDS exchanges)
template <int N> struct Plan;
template <> struct Plan<16> : Radices<16> {};
template <> struct Plan<32> : Radices<8, 4> {};
template <> struct Plan<64> : Radices<8, 8> {};
template <> struct Plan<128> : Radices<16, 8> {};
template <> struct Plan<256> : Radices<16, 16> {};
template <> struct Plan<512> : Radices<8, 8, 8> {};
template <> struct Plan<1024> : Radices<16, 8, 8> {};
template <> struct Plan<2048> : Radices<16, 16, 8> {};
template <> struct Plan<4096> : Radices<16, 16, 16> {};

template <int N> constexpr int lds_padded() { return N + N / 16; }
// (a single-column batch lays one of its exchanges out with 1/8 of padding: lds_index_x)
template <int N, int C> constexpr int fft_lds_elems() { return C == 1 ? N + N / 8 : lds_padded<N>() * C; }

template <int C>
__device__ __forceinline__ int lds_index(int idx, int c) { return (idx + (idx >> 4)) * C + c; }
// lds_index(idx0 + d, c) - lds_index(idx0, c) for the element distances d the stages use -- a compile-time constant, so that the
// R accesses of a butterfly share ONE address register and differ in the instruction's immediate offset (three to four VALU
// instructions less per LDS access: about 40 % of a middle stage's vector instructions were address arithmetic).  Exact for every
// access of a Stockham stage with power-of-two radices: reads touch j + i * (N / R) with j < N / R, writes j0 + i * NS with
// j0 = (j - k) * R + k, k = j % NS < NS; in both cases (idx0 mod 16) + (d mod 16) < 16, so the padding term idx / 16 splits into
// idx0 / 16 + d / 16 (checked exhaustively for every plan of this file: tools/check_lds_offsets.py).
template <int C> constexpr int lds_delta(int d) { return (d + d / 16) * C; }

// Per-exchange layouts of a SINGLE-COLUMN batch (C = 1: k_zpass_c1, the row-N/2 groups of the x passes).  With one column the lanes of a
// wave hold consecutive butterflies j, so every stage READS 64 consecutive elements per instruction -- conflict-free exactly when the
// image has no padding inside aligned runs of 32 -- while the WRITES of the stage before scatter with a stride that depends on that
// stage's NS.  No single padding serves all of them (bank model of ds_read_b64: 2 x 32 lanes on 64 banks, ds_write_b64: 4 x 16 lanes on
// 32 banks; PMC of round 4: 33 % of the single-transform z pass's LDS cycles were conflicts with the 1/16 padding everywhere), but the
// exchanges are independent -- all reads, barrier, all writes -- so each gets its own, named by the NS of the stage that WRITES it:
//     NSW = 1    idx + idx / 16          first-stage writes (stride R) conflict-free; the reads behind them stay 2-way
//     NSW = 8    idx + 8 * (idx / 64)    writes of 8 lanes x 8 elements land 16 banks apart: both sides conflict-free
//     NSW >= 64  idx                     writes are contiguous runs of >= 64 elements already: both sides conflict-free
// (radix-16 first stages -- NSW = 16 -- keep the 1/16 padding).  Weighted by the instruction costs that is 8.5 instead of 12 cycles per
// element and transform.  The offsets of a butterfly's accesses stay compile-time constants (tools/check_lds_offsets.py checks every
// plan, stage, butterfly and leg of these layouts too).  Batches of several columns keep the c-interleaved 1/16 image, and so do
// transforms below 2048 points: at 1024 (128 threads, two waves per workgroup) the new layouts measured 3 % SLOWER (8 x 1024^2 z pass
// 33.9-34.2 -> 34.9-35.3 us) against -3.5 % at 2048 and -2 % at 4096 (profiles/r04_zpass_experiments.txt).
template <int C, int NSW, int N> __device__ __forceinline__ int lds_index_x(int idx, int c)
{
    if constexpr (C == 1 && N >= 2048 && NSW == 8) return idx + 8 * (idx >> 6);
    else if constexpr (C == 1 && N >= 2048 && NSW >= 64) return idx;
    else return lds_index<C>(idx, c);
}
template <int C, int NSW, int N> constexpr int lds_delta_x(int d)
{
    return (C == 1 && N >= 2048 && NSW == 8) ? d + 8 * (d / 64) : ((C == 1 && N >= 2048 && NSW >= 64) ? d : lds_delta<C>(d));
}

// Row-major image of the LAST exchange of an LM = 1 batch (the x passes): element (idx, c) at c * lds_row<N>() + idx.  The last stage of
// that layout reads 64 consecutive idx of ONE column per wave -- in the c-interleaved image those are 32 bytes apart, eight lanes per bank
// pair: 4-way conflicts, a third of the x passes' LDS cycles (PMC, profiles/r03z_pmc_*) -- here they are 512 contiguous bytes.  The stage
// before it writes 16 lanes = (16 / C) idx x C columns per pass, 128 / C contiguous bytes in each of C rows: conflict-free when the rows
// start 32 / C banks apart (mod 32), i.e. lds_row = N + 16 / C float2.  In place like every exchange (all reads, barrier, all writes), so
// changing the layout between two stages costs nothing.
template <int N, int C> constexpr int lds_row() { return N + 16 / C; }
template <int N, int C> __device__ __forceinline__ int lds_index_rm(int idx, int c) { return c * lds_row<N, C>() + idx; }
// (only where the LM = 1 lane layout is in effect -- last_stage_map: whole waves per column -- and the rows fit the image's allocation)
template <int N, int C, int LM, int RLAST> constexpr bool lds_row_major_last()
{
    return LM == 1 && C > 1 && 16 % C == 0 && ((N / RLAST) * C) % (64 * C) == 0 && C * lds_row<N, C>() <= fft_lds_elems<N, C>();
}

// Work-item -> (column c, butterfly j) of the LAST stage.  Two lane layouts for a batch whose
// items fill whole waves:
//   LM = 0  the 64 lanes of a wave cover 64/C consecutive j of EVERY column, so a store
//           instruction writes 64/C consecutive outputs per column (whole 128-byte lines per
//           16-lane group) instead of C-interleaved 64-byte pieces;
//   LM = 1  a wave covers 64 consecutive j of ONE column: one contiguous run per store
//           instruction (1 KiB of float4 texels, 256 B of floats).  The x pass uses this one:
//           3-6 % of a pipelined frame over LM = 0, and the raw-height rows stream.
// Otherwise c is fastest, as in the other stages.
template <int N, int C, int R, int LM = 0>
__device__ __forceinline__ void last_stage_map(int w, int& c, int& j)
{
    constexpr int ITEMS = (N / R) * C;
    if constexpr (LM == 1 && ITEMS % (64 * C) == 0 && C > 1) {
        c = (w / 64) % C;
        j = (w % 64) + 64 * (w / (64 * C));
        return;
    }
    if constexpr (ITEMS % 64 == 0 && 64 % C == 0 && C > 1) {
        constexpr int JB = 64 / C;
        c = (w / JB) % C;
        j = (w % JB) + JB * (w / 64);
        return;
    }
    c = w % C;
    j = w / C;
}

// Base twiddles of every (stage, work item) of one thread, fetched ONCE per kernel
// into registers (a handful of VGPRs): the table loads then overlap the kernel's
// first global loads instead of sitting on the critical path of every transform.
template <int N, int C, int T, class P, int LM = 0> struct TwiddleRegs {
    static constexpr int it_of(int stage) { return ((N / P::r[stage]) * C + T - 1) / T; }
    static constexpr int itmax()
    {
        int m = 1;
        for (int s = 0; s < P::S; ++s) m = it_of(s) > m ? it_of(s) : m;
        return m;
    }
    c32 w[P::S][itmax()];

    // TS: stride in the table -- a table of exp(+2 pi i k / (TS N)) serves a transform of length N at every TS-th entry (the
    // half-size transform of the real height column reads the tile size's table with TS = 2: ocean_kernels.h, zpass_height_half)
    template <int STAGE, int NS, int TS = 1>
    __device__ __forceinline__ void load_from(const c32* __restrict__ tw, int tid)
    {
        constexpr int R = P::r[STAGE];
        constexpr int ITEMS = (N / R) * C;
        constexpr int IT = (ITEMS + T - 1) / T;
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int wi = tid + u * T;
            int j = (wi < ITEMS ? wi : 0) / C;
            if constexpr (STAGE == P::S - 1) {
                int c_unused;
                last_stage_map<N, C, R, LM>(wi < ITEMS ? wi : 0, c_unused, j);
            }
            if constexpr (NS > 1) w[STAGE][u] = tw[TS * ((j % NS) * (N / (NS * R)))];
            else w[STAGE][u] = make_float2(1.f, 0.f);
        }
        if constexpr (STAGE + 1 < P::S) load_from<STAGE + 1, NS * R, TS>(tw, tid);
    }
    __device__ __forceinline__ void load(const c32* __restrict__ tw, int tid) { load_from<0, 1>(tw, tid); }
    template <int TS> __device__ __forceinline__ void load_strided(const c32* __restrict__ tw, int tid) { load_from<0, 1, TS>(tw, tid); }
};

// One Stockham stage over the whole batch.
//   FIRST: inputs from in(idx, c, u, i) (u, i are unrolled constants: the functor may
//          serve values it prefetched into registers);  otherwise from LDS
//   LAST : outputs to out(idx, c, value, u, i) (u, i are unrolled constants);
//          otherwise to LDS, in place (reads complete -> barrier -> writes)
template <int N, int R, int NS, int C, int T, bool FIRST, bool LAST, int STAGE, int LM, bool WRM, int NSR, class TW, class In, class Out>
__device__ __forceinline__ void fft_stage(c32* lds, const TW& twr, int tid, In& in, Out& out)
{
    // NSR: the NS of the stage that wrote the image this stage reads (lds_index_x: per-exchange layouts of single-column batches)
    // WRM: this stage's outputs feed the last stage of a batch whose last exchange is row-major (lds_row_major_last)
    constexpr bool RRM = LAST && !FIRST && lds_row_major_last<N, C, LM, R>();      // ... and the last stage reads that image
    [[maybe_unused]] constexpr int STAMP_BASE = STAGE;
    constexpr int ITEMS = (N / R) * C;
    constexpr int IT = (ITEMS + T - 1) / T;
    constexpr bool GUARD = (ITEMS % T) != 0;
    // volatile: keeps every exchange access a ds_read_b64 / ds_write_b64 of its own.  With a shared base register and immediate
    // offsets the compiler otherwise pairs them into ds_read2(st64)_b64 / ds_write2_b64, which run at HALF the LDS rate and map to 32
    // instead of 64 banks -- the padding of the image is laid out for the latter: bank-conflict cycles doubled in all three kernels
    // (PMC: profiles/r03_lds_experiments.txt).
    typedef __attribute__((address_space(3))) volatile v2 lds_v2;      // (explicitly LDS: a volatile generic pointer would turn into flat accesses)
    lds_v2* ldsv = (lds_v2*)lds;
    if constexpr (LAST) {
        // nothing is written back to LDS: finish one work item at a time (R complex live, not IT*R)
#pragma unroll
        for (int u = 0; u < IT; ++u) {
            const int w = tid + u * T;
            if (!GUARD || w < ITEMS) {
                int c, j;
                last_stage_map<N, C, R, LM>(w, c, j);
                v2 x[R];
                [[maybe_unused]] const int rb = RRM ? lds_index_rm<N, C>(j, c) : lds_index_x<C, NSR, N>(j, c);
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    if constexpr (FIRST) x[i] = tov(in(j + i * (N / R), c, u, i));
                    else if constexpr (RRM) x[i] = ldsv[rb + i * (N / R)];
                    else x[i] = ldsv[rb + lds_delta_x<C, NSR, N>(i * (N / R))];
                }
#ifndef OCEAN_ABL_NOFFT
                if constexpr (NS > 1) apply_twiddles<R>(x, tov(twr.w[STAGE][u]));
                Dft<R>::run(x);
#endif
                const int k = j % NS;
                const int j0 = (j - k) * R + k;
#pragma unroll
                for (int i = 0; i < R; ++i) out(j0 + i * NS, c, toc(x[i]), u, i);
            }
        }
        OCEAN_STAMP(8 + 3 * STAMP_BASE);
        OCEAN_STAMP(9 + 3 * STAMP_BASE);
        return;
    }
    v2 x[IT][R];
#pragma unroll
    for (int u = 0; u < IT; ++u) {
        const int w = tid + u * T;
        if (!GUARD || w < ITEMS) {
            const int c = w % C, j = w / C;
            [[maybe_unused]] const int rb = lds_index_x<C, NSR, N>(j, c);
#pragma unroll
            for (int i = 0; i < R; ++i) {
                if constexpr (FIRST) x[u][i] = tov(in(j + i * (N / R), c, u, i));
                else x[u][i] = ldsv[rb + lds_delta_x<C, NSR, N>(i * (N / R))];
            }
#ifndef OCEAN_ABL_NOFFT
            if constexpr (NS > 1) apply_twiddles<R>(x[u], tov(twr.w[STAGE][u]));
            Dft<R>::run(x[u]);
#endif
        }
    }
    OCEAN_STAMP(8 + 3 * STAMP_BASE);
    __syncthreads();   // every reader of the old image is done
    OCEAN_STAMP(9 + 3 * STAMP_BASE);
#pragma unroll
    for (int u = 0; u < IT; ++u) {
        const int w = tid + u * T;
        if (!GUARD || w < ITEMS) {
            const int c = w % C, j = w / C;
            const int k = j % NS;
            const int j0 = (j - k) * R + k;
            const int wb = WRM ? lds_index_rm<N, C>(j0, c) : lds_index_x<C, NS, N>(j0, c);
#pragma unroll
            for (int i = 0; i < R; ++i) ldsv[wb + (WRM ? i * NS : lds_delta_x<C, NS, N>(i * NS))] = x[u][i];
        }
    }
}

template <int N, int C, int T, class P, int STAGE, int NS, int LM, class TW, class In, class Out>
__device__ __forceinline__ void run_stages(c32* lds, const TW& twr, int tid, In& in, Out& out)
{
    constexpr int R = P::r[STAGE];
    constexpr int NSR = STAGE == 0 ? 1 : NS / P::r[STAGE == 0 ? 0 : STAGE - 1];      // the NS of the stage before this one
    constexpr bool FIRST = STAGE == 0, LAST = STAGE == P::S - 1;
    constexpr bool WRM = STAGE == P::S - 2 && lds_row_major_last<N, C, LM, P::last>();
    fft_stage<N, R, NS, C, T, FIRST, LAST, STAGE, LM, WRM, NSR>(lds, twr, tid, in, out);
    OCEAN_STAMP(10 + 3 * STAGE);
    if constexpr (!LAST) {
        __syncthreads();
        run_stages<N, C, T, P, STAGE + 1, NS * R, LM>(lds, twr, tid, in, out);
    }
}

// C interleaved length-N transforms by the T threads of the workgroup, radix
// plan P (product of radices = N).  `lds` needs fft_lds_elems<N, C>() float2;
// `twr` = TwiddleRegs<N, C, T, P> loaded from the table tw[k] = exp(+2 pi i k / N).
// The call may start while other waves still read `lds` from a previous call:
// the first LDS write is preceded by a barrier.
template <int N, int C, int T, class P = Plan<N>, int LM = 0, class In, class Out>
__device__ __forceinline__ void batch_fft(c32* lds, TwiddleRegs<N, C, T, P, LM>& twr, int tid, In& in, Out& out)
{
    static_assert(P::product() == N, "radix plan does not match the transform length");
    // Launder the base twiddles: otherwise the compiler hoists the whole power chain
    // (up to 15 complex per stage) out of consecutive transforms and keeps ~90 VGPRs
    // live across them -- recomputing 14 products per butterfly is far cheaper than
    // the occupancy that costs.  In place (round 3 laundered a COPY per transform, which kept two sets of
    // base twiddles alive while a transform ran), and only the stages that have twiddles (the first stage's are the constant 1).
#pragma unroll
    for (int s = 1; s < P::S; ++s)
#pragma unroll
        for (int u = 0; u < TwiddleRegs<N, C, T, P, LM>::it_of(s); ++u)
            asm volatile("" : "+v"(twr.w[s][u].x), "+v"(twr.w[s][u].y));
    run_stages<N, C, T, P, 0, 1, LM>(lds, twr, tid, in, out);
}

// Mapping of the FIRST stage: work item w = tid + u*T reads inputs
// idx = j + i*(N/R0), column c, with j = w / C, c = w % C  (in(idx, c, u, i)).
template <int N, int C, int T, class P = Plan<N>> struct FirstStage {
    static constexpr int R0 = P::r[0];
    static constexpr int ITEMS = (N / R0) * C;
    static constexpr int IT = (ITEMS + T - 1) / T;
    static constexpr bool GUARD = (ITEMS % T) != 0;
    static constexpr int STRIDE = N / R0;
};

// Mapping of the LAST stage: work item w = tid + u*T owns outputs
// idx = j + i*(N/RL), column c, with j = w / C, c = w % C (k == j there).
template <int N, int C, int T, class P = Plan<N>, int LM = 0> struct LastStage {
    static constexpr int RL = P::last;
    static constexpr int ITEMS = (N / RL) * C;
    static constexpr int IT = (ITEMS + T - 1) / T;
    static constexpr bool GUARD = (ITEMS % T) != 0;
    static constexpr int STRIDE = N / RL;
    static __device__ __forceinline__ void map(int w, int& c, int& j) { last_stage_map<N, C, RL, LM>(w, c, j); }
};

}  // namespace ocean
