// ocean_launch.h -- launch_frame<N>: the three launches of one frame at tile size N (included by frames_*.hip).
#pragma once
#include <cstdlib>

#include "ocean_ctx.h"

using namespace ocean;

// ---------------------------------------------------------------------------------
template <class K>
static hipError_t allow_lds(K kernel, size_t bytes)
{
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

// One kernel launch.  With an event pair the launch goes through hipExtLaunchKernelGGL, which attaches the events
// to the dispatch itself: their interval is the kernel's own execution time (what rocprofv3 reports), free of the
// 2.5-3 us of marker/launch processing that events recorded around a launch carry.
template <class K, class... A>
static void launch(K kernel, dim3 grid, dim3 block, size_t lds, hipStream_t st, hipEvent_t* ev, A... args)
{
    if (ev) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, st, ev[0], ev[1], 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, lds, st, args...);
}

// tile sizes whose single-tile frames run the merged x pass (measured: profiles/r05_small_tile_experiments.txt); alone = a serial frame
template <int N> constexpr bool xmerge_pays(bool alone) { return alone ? N <= 128 : N <= 512; }

template <int N>
static hipError_t launch_frame(ocean_ctx* c, const FrameArgs& a, int stream_maps /* bit 0: normal map, bit 1: displacement map, bit 2: intermediates */,
                               hipStream_t st, hipEvent_t* marks /* 2 x OCEAN_MAX_LAUNCHES events (start, stop per launch) or null */)
{
    using G = Geo<N>;
    using HF = Half<N>;
    const unsigned tiles = c->tiles;
    hipError_t e;
    constexpr int C = G::CC;
    constexpr size_t lds_rows = zpass_lds_bytes<N, 1>();
    constexpr bool HAS2 = zpass_has_width2<N>();
    constexpr size_t lds_rows2 = zpass_lds_bytes<N, 2>();
    constexpr size_t lds_m = sizeof(c32) * fft_lds_elems<N, C>();
    constexpr size_t lds_b = lds_m + sizeof(float) * 2 * ((G::T_C + 63) / 64);
    static_assert(HF::NUP % (2 * C) == 0, "height row blocks");
    constexpr unsigned hb_b = xpass_height_groups<N, C>(), nb = (HF::NU + C - 1) / C;
    const bool fast = !a.h0h && a.omega_q;       // the usual form of the spectrum: fp32 h0, 16-bit dispersion
    const bool alone = (stream_maps & 16) != 0;
    const unsigned cus = (unsigned)(c->cu_count > 0 ? c->cu_count : 0);
    // Staggered start (ocean_kernels.h: start_ramp_wait; the rule and its constants: OceanTuning, ocean_ctx.h).  A launch gets one when its WHOLE
    // grid is resident at once on this device -- grid <= compute units x resident workgroups per unit, the latter asked of the runtime for the
    // kernels in question -- and it covers enough texels for the bursts to be worth separating; the spread is a fraction of its expected duration.
    // In practice: the three launches of one 2048^2 tile (every mode and precision), 4.2 / 5.1 / 4.0 us serial, the x passes twice that when
    // pipelined; not 1024^2 (too few texels), not 4096^2 or batches (several rounds, which overlap by themselves).  ocean_set_start_ramp(ctx, 0)
    // switches it off altogether (a device shared with other work: a workgroup's wait is simply lost).
    int ramp_z = 0, ramp_b = 0, ramp_d = 0;          // (computed below, once the kernels' attributes are set)
    // function attributes are per device; a context belongs to one device and one thread, so the flag
    // lives in the context (no process-wide state shared between contexts or threads)
    // Which forms of the z pass a tile size has (ocean_kernels.h).  ZW1: one column, two-transform batches (four-transform batches at 256 /
    // 512) -- every size below 1024 and a single 1024^2 tile; C1: one column, single-transform batches, half the threads -- 4096^2
    // always, 2048^2 and batches of 1024^2 unless the intermediates are streamed; ZW2: two neighbouring columns (whole-line stores) --
    // streamed intermediates at 1024^2 and 2048^2.  Developer builds keep every form for A/B runs (OCEAN_ZW, OCEAN_ZC1).
#ifdef OCEAN_DEVELOPER
    constexpr bool DEV = true;
#else
    constexpr bool DEV = false;
#endif
    constexpr bool HAS1 = DEV || N <= 1024;
    constexpr bool HASC1 = zpass_has_c1<N>();
    constexpr bool HAS2_PLAIN = DEV && HAS2;              // (the two-column form with plain stores: never selected by the shipped rules)
    if (c->attr_n != (uint32_t)N) {
#define OCEAN_ALLOW_Z(znt, z16, fast) \
        if constexpr (HAS1) if ((e = allow_lds(k_zpass<N, G::T_ROWS, typename G::PR, znt, z16, 1, fast>, lds_rows)) != hipSuccess) return e; \
        if constexpr (HAS2 && (znt || HAS2_PLAIN)) if ((e = allow_lds(k_zpass<N, G::T_ROWS, typename G::PR, znt, z16, 2, fast>, lds_rows2)) != hipSuccess) return e; \
        if constexpr (HASC1) if ((e = allow_lds(k_zpass_c1<N, zpass_c1_threads<N>(), typename C1Plan<N>::type, znt, z16, fast>, zpass_c1_lds_bytes<N>())) != hipSuccess) return e; \
        if constexpr (HASC1 && zpass_has_wt<N>() && !znt && !z16) if ((e = allow_lds(k_zpass_c1<N, zpass_c1_threads<N>(), typename C1Plan<N>::type, false, false, fast, true>, zpass_c1_lds_bytes<N>())) != hipSuccess) return e;
#define OCEAN_ALLOW_Z2(fast) OCEAN_ALLOW_Z(false, false, fast) OCEAN_ALLOW_Z(true, false, fast) OCEAN_ALLOW_Z(false, true, fast) OCEAN_ALLOW_Z(true, true, fast)
        OCEAN_ALLOW_Z2(true) OCEAN_ALLOW_Z2(false)
#undef OCEAN_ALLOW_Z2
#undef OCEAN_ALLOW_Z
#define OCEAN_ALLOW_X(kern, lds) \
        if ((e = allow_lds(kern<N, C, G::T_C, typename G::PC, false, false, false>, lds)) != hipSuccess) return e; \
        if ((e = allow_lds(kern<N, C, G::T_C, typename G::PC, true, false, false>, lds)) != hipSuccess) return e;  \
        if ((e = allow_lds(kern<N, C, G::T_C, typename G::PC, false, true, false>, lds)) != hipSuccess) return e;  \
        if ((e = allow_lds(kern<N, C, G::T_C, typename G::PC, true, true, false>, lds)) != hipSuccess) return e;   \
        if ((e = allow_lds(kern<N, C, G::T_C, typename G::PC, false, false, true>, lds)) != hipSuccess) return e;  \
        if ((e = allow_lds(kern<N, C, G::T_C, typename G::PC, true, false, true>, lds)) != hipSuccess) return e;   \
        if ((e = allow_lds(kern<N, C, G::T_C, typename G::PC, false, true, true>, lds)) != hipSuccess) return e;   \
        if ((e = allow_lds(kern<N, C, G::T_C, typename G::PC, true, true, true>, lds)) != hipSuccess) return e;
        OCEAN_ALLOW_X(k_xpass_b, lds_b)
        OCEAN_ALLOW_X(k_xpass_disp, lds_m)
#undef OCEAN_ALLOW_X
        c->attr_n = (uint32_t)N;
    }
    if (c->start_ramp && cus) {
        if (c->occ_n != N) {
            int oz = 0, oza = 0, ob = 0, od = 0;
            if constexpr (zpass_has_c1<N>()) {      // (the usual form of the spectrum and the instantiation that carries every form: different launch bounds)
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&oz, k_zpass_c1<N, zpass_c1_threads<N>(), typename C1Plan<N>::type, false, false, true>,
                                                                   zpass_c1_threads<N>(), zpass_c1_lds_bytes<N>());
                (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&oza, k_zpass_c1<N, zpass_c1_threads<N>(), typename C1Plan<N>::type, false, false, false>,
                                                                   zpass_c1_threads<N>(), zpass_c1_lds_bytes<N>());
            }
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&ob, k_xpass_b<N, C, G::T_C, typename G::PC, false, false, false>, G::T_C, lds_b);
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&od, k_xpass_disp<N, C, G::T_C, typename G::PC, false, false, false>, G::T_C, lds_m);
            (void)hipGetLastError();
            c->occ_z = oz; c->occ_z_all = oza; c->occ_b = ob; c->occ_d = od; c->occ_n = N;
        }
        int bpt[3];
        ocean_launch_bytes_per_texel((stream_maps & 8) != 0, a.mode == 3, a.h0h != nullptr, a.omega_q != nullptr, bpt);
        const OceanTuning& tu = c->tune;
        const double texels = (double)tiles * N * N;
        auto ramp_ticks = [&](int bytes_per_texel, unsigned workgroups, int per_cu, double scale) -> int {
            const double bytes = texels * bytes_per_texel;
            const unsigned slots = (unsigned)per_cu < tu.ramp_max_wg_per_cu ? (unsigned)per_cu : tu.ramp_max_wg_per_cu;
            if (per_cu <= 0 || workgroups > slots * cus || texels < tu.ramp_min_texels) return 0;
            return (int)(tu.ramp_frac * scale * bytes / tu.ramp_rate_bytes_per_s * 1.0e8 + 0.5);      // 10 ns ticks (s_memrealtime)
        };
        ramp_z = ramp_ticks(bpt[0], (unsigned)(N / 2 + 1) * tiles, fast ? c->occ_z : c->occ_z_all, 1.0);
        // k_xpass_b: over its NORMAL workgroups and their bytes alone -- the height workgroups (2 B/texel in, 2 out) write next to nothing and all
        // start at once (item 10); the Jacobian mode's few workgroups beyond one round do not change the picture (item 12: same gain)
        ramp_b = ramp_ticks(bpt[1] - 4, (hb_b + nb) * tiles, c->occ_b, alone ? 1.0 : tu.ramp_pipelined_x);
        ramp_d = ramp_ticks(bpt[2], nb * tiles, c->occ_d, alone ? 1.0 : tu.ramp_pipelined_x);
    }
    [[maybe_unused]] const bool ramp = ramp_z || ramp_b || ramp_d;
#ifdef OCEAN_DEVELOPER      // A/B builds only: the shipped library reads no environment
    {   static const char* const rz = getenv("OCEAN_RAMP_Z"); static const char* const rb = getenv("OCEAN_RAMP_B"); static const char* const rd = getenv("OCEAN_RAMP_D");
        static const char* const ra = getenv("OCEAN_RAMP_ANY");     // 1: every size and batch
        const bool on = ramp || (ra && atoi(ra) == 1);
        if (rz && on) ramp_z = atoi(rz);
        if (rb && on) ramp_b = atoi(rb);
        if (rd && on) ramp_d = atoi(rd); }
#endif
#ifdef OCEAN_STAMPS
    // diagnostic: stamps are recorded for ONE kernel of the frame (env OCEAN_DEBUG_STAMP_KERNEL = 0, 1, 2)
    static unsigned long long* null_ptr = nullptr;
    const int stamp_k = getenv("OCEAN_DEBUG_STAMP_KERNEL") ? atoi(getenv("OCEAN_DEBUG_STAMP_KERNEL")) : 0;
    auto arm = [&](int k) {
        (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(ocean::g_stamps), (k == stamp_k && c->stamps) ? &c->stamps : &null_ptr,
                                     sizeof(c->stamps), 0, hipMemcpyHostToDevice, st);
    };
    arm(0);
#endif
    // the form of the z pass (see above); the usual form of the spectrum (fp32 h0, 16-bit dispersion) has instantiations without the
    // other forms' code (FAST)
    bool c1 = HASC1 && zpass_c1_pays<N>(stream_maps, tiles);
    bool zw2 = HAS2 && !c1 && (stream_maps & 4);
#ifdef OCEAN_DEVELOPER      // A/B builds only: the shipped library reads no environment
    static const char* const c1_env = getenv("OCEAN_ZC1");                  // 0 / 1
    if (c1_env && HASC1) c1 = atoi(c1_env) != 0;
    zw2 = HAS2 && !c1 && (stream_maps & 4);
    static const char* const zw_env = getenv("OCEAN_ZW");                   // 1 or 2
    if (zw_env && HAS2 && !c1) zw2 = atoi(zw_env) == 2;
#endif
    // Frame order: z pass (four transforms per column) -> k_xpass_b (HEIGHT + NORMAL workgroups) -> k_xpass_disp.  Developer builds can run
    // the SPLIT order of profiles/r05_4096_experiments.txt instead (OCEAN_FRAME_ORDER=2: z pass {height, pair 0} -> HEIGHT workgroups ->
    // k_xpass_disp -> z pass {pair 1, pair 2} -> NORMAL workgroups; bit-identical, slower at every size -- the x passes sit on the map write
    // stream, not on their reads, and the second animation of the spectrum is paid in full); the per-launch fields it needs (zmask, xb_roles,
    // rec_mode) are part of every launch.
    bool split = false;
#ifdef OCEAN_DEVELOPER
    {   const char* const fo = getenv("OCEAN_FRAME_ORDER");       // (read per frame: tools/ab_order.py flips it between passes)
        split = fo && atoi(fo) == 2 && (c1 || zw2) && (a.mode == 0 || a.mode == 1); }
#endif
    // Merged x pass (round 5): HEIGHT, NORMAL and DISP workgroups in ONE launch of k_xpass_b (xb_roles = 7) instead of k_xpass_b + k_xpass_disp.
    // The DISP workgroups transform pair 0 at once and wait for the tile's HEIGHT workgroups only before their stores (raw heights handed over
    // write-through, ocean_kernels.h: store_wt / wait_counter): a small single tile is three short DEPENDENT latency chains
    // (profiles/r02_small_tile_experiments.txt item 0), and this runs the third one beside the second.  Where every workgroup of the launch
    // is resident at once, one per compute unit (the regime the hand-off recipe is measured for: MI355X_MICROARCH.md), a single tile, not
    // the Jacobian mode (its displacement pass also needs the NORMAL workgroups' product plane).  Same arithmetic per texel: same bits.
    // Where it pays (profiles/r05_small_tile_experiments.txt): a SERIAL frame only up to 128^2 (64^2: 12.8 -> 11.6 us; from 256^2 up the
    // hand-off's three dependent memory round trips cost more than the kernel boundary they replace: 512^2 16.6 -> 18.2 us); PIPELINED frames
    // of one tile up to 512^2, which are bound by the rate of launches, not by any kernel (512^2, depth 4: 13.2-14.9 -> 8.0 us per frame).
    bool merged_x = xmerge_pays<N>(alone) && tiles == 1 && a.mode != 3 && !split && c->merged_x &&
                    (hb_b + 2u * nb) <= c->tune.handoff_wg_per_cu * cus;
#ifdef OCEAN_DEVELOPER
    {   const char* const xm = getenv("OCEAN_XMERGE");            // 0 / 1: force (any size, any batch; still not the Jacobian mode)
        if (xm) merged_x = atoi(xm) != 0 && a.mode != 3 && !split; }
#endif
    // ONE launch for the whole frame (k_frame; round 5): pipelined frames of one tile up to 128^2 in the usual form (fp32 spectrum with 16-bit
    // dispersion, fp32 intermediates, both maps streamed, not the Jacobian mode) -- they are bound by the rate of launches.  Measured at depth 4,
    // us per frame, three launches / merged x pass / one launch: 64^2 11.5 / 7.6 / 3.9, 128^2 15.0 / 10.3 / 5.2 -- and 256^2 8.2-15.4 / 5.8 / 8.3,
    // 512^2 11.6-12.3 / 7.7 / 18.6: from 256^2 up the z pass's 8-byte write-through stores and the x-axis workgroups' reads past the L2 cost more
    // than the launch they save (profiles/r05_small_tile_experiments.txt), so those sizes keep the merged x pass.
    constexpr bool HAS_ONE = (N <= 128 || (DEV && N <= 512)) && G::T_ROWS == G::T_C;      // (developer builds: up to 512^2, for the A/B of that log)
    // (like the merged x pass only where every workgroup of the grid has a slot at once -- one per compute unit, the regime the hand-off recipe is
    //  measured for: forward progress then never depends on the order in which workgroups are dispatched)
    [[maybe_unused]] bool one_launch = HAS_ONE && !alone && tiles == 1 && a.mode <= 2 && fast && (stream_maps & 15) == 3 && !split && c->merged_x &&
                                       ((unsigned)(N / 2 + 1) + hb_b + 2u * nb) <= c->tune.handoff_wg_per_cu * cus;
#ifdef OCEAN_DEVELOPER
    {   const char* const ol = getenv("OCEAN_ONE_LAUNCH");        // 0 / 1 (the form's own preconditions still hold)
        if (ol) one_launch = one_launch && atoi(ol) != 0; }
#endif
    c->handoff = merged_x || one_launch;      // (what recover_fault re-runs without a hand-off, should one of this frame's in-launch waits give up)
    int launches = 0;
    auto next_marks = [&](int kernel) -> hipEvent_t* {      // the event pair of the frame's next launch; remembers which kernel it times
        c->launch_kernel[launches] = kernel;
        hipEvent_t* m = marks ? marks + 2 * launches : nullptr;
        ++launches;
        return m;
    };
    const int rec_last = a.rec_mode;                         // what the frame's LAST launch does with the completion records (ocean_api.hip)
    auto launch_z = [&](int zmask) -> hipError_t {
        unsigned gx = zw2 ? N / 4 + 1 : N / 2 + 1;
        FrameArgs za = a;
        za.zmask = zmask; za.rec_mode = 0;
        // one 2048^2 tile: the single-transform form's 1025 workgroups are one resident round -- staggered start (start_ramp_wait)
        za.start_ramp = c1 ? ramp_z : 0;
#if defined(OCEAN_STAMPS) || defined(OCEAN_DEVELOPER)
        if (const char* ev = getenv("OCEAN_DEBUG_ROWS_GRID")) gx = (unsigned)atoi(ev);   // diagnostic: partial grid (results wrong)
#endif
        const unsigned threads = c1 ? (unsigned)zpass_c1_threads<N>() : (unsigned)G::T_ROWS;
        const size_t lds = c1 ? zpass_c1_lds_bytes<N>() : (zw2 ? lds_rows2 : lds_rows);
        const dim3 grid(gx, tiles), block(threads);
        if (launches == 0) {
            ocean_launch_info& li = c->last_launch[0];
            li.tile_size = N; li.grid_x = gx; li.grid_y = tiles; li.block = threads; li.mode = (uint32_t)a.mode;
            li.per_workgroup = zw2 ? 2u : 1u;
            li.lds_bytes = (uint32_t)lds;
            li.flags = ((stream_maps & 4) ? OCEAN_LAUNCH_NT_INTER : 0u) | ((stream_maps & 8) ? OCEAN_LAUNCH_HALF_INTER : 0u) |
                       (a.mode == 3 ? OCEAN_LAUNCH_JACOBIAN : 0u) | (a.h0h ? OCEAN_LAUNCH_FP16_SPECTRUM : 0u) |
                       (a.omega_q ? 0u : OCEAN_LAUNCH_FP32_DISPERSION) | (c1 ? OCEAN_LAUNCH_SINGLE_TRANSFORM : 0u) |
                       (za.start_ramp ? OCEAN_LAUNCH_STAGGERED_START : 0u) | (split ? OCEAN_LAUNCH_SPLIT_ORDER : 0u);
        }
        hipEvent_t* mz = next_marks(0);
#ifdef OCEAN_CLOCKPROBE
        {   // diagnostic: this translation unit's copy of the probe pointer, set when the context's buffer changes (not per frame)
            static unsigned long long* armed = nullptr;
            if (armed != c->stamps) { armed = c->stamps; (void)hipMemcpyToSymbol(HIP_SYMBOL(ocean::g_clockprobe), &armed, sizeof(armed)); }
        }
#endif
        bool launched = false;
        // a SERIAL frame's fp32 intermediates go out write-through where that form exists (ocean_kernels.h: store_z, zpass_has_wt)
        if constexpr (HASC1 && zpass_has_wt<N>()) {
            bool wt = c1 && alone && !(stream_maps & (4 | 8)) && c->z_write_through;
#ifdef OCEAN_DEVELOPER
            if (const char* ev = getenv("OCEAN_Z_WT")) wt = wt && atoi(ev) != 0;     // (read per frame: A/B inside one process)
#endif
            if (wt) {
                if (fast) launch(k_zpass_c1<N, zpass_c1_threads<N>(), typename C1Plan<N>::type, false, false, true, true>, grid, block, lds, st, mz, za);
                else launch(k_zpass_c1<N, zpass_c1_threads<N>(), typename C1Plan<N>::type, false, false, false, true>, grid, block, lds, st, mz, za);
                launched = true;
                if (launches == 1) c->last_launch[0].flags |= OCEAN_LAUNCH_WT_INTER;
            }
        }
#define OCEAN_ZPASS3(znt, z16, fast) \
        do { if (launched) break; if constexpr (HASC1) { if (c1) { launch(k_zpass_c1<N, zpass_c1_threads<N>(), typename C1Plan<N>::type, znt, z16, fast>, grid, block, lds, st, mz, za); launched = true; break; } } \
             if constexpr (HAS2 && (znt || HAS2_PLAIN)) { if (zw2) { launch(k_zpass<N, G::T_ROWS, typename G::PR, znt, z16, 2, fast>, grid, block, lds, st, mz, za); launched = true; break; } } \
             if constexpr (HAS1) { if (!c1 && !zw2) { launch(k_zpass<N, G::T_ROWS, typename G::PR, znt, z16, 1, fast>, grid, block, lds, st, mz, za); launched = true; } } } while (0)
#define OCEAN_ZPASS2(znt, z16) \
        do { if (fast) OCEAN_ZPASS3(znt, z16, true); else OCEAN_ZPASS3(znt, z16, false); } while (0)
#define OCEAN_ZPASS(znt) \
        do { if (stream_maps & 8) OCEAN_ZPASS2(znt, true); else OCEAN_ZPASS2(znt, false); } while (0)
        if (stream_maps & 4) OCEAN_ZPASS(true); else OCEAN_ZPASS(false);
#undef OCEAN_ZPASS
#undef OCEAN_ZPASS2
#undef OCEAN_ZPASS3
        if (!launched) return hipErrorInvalidConfiguration;     // (a form this build does not carry: the rules above never ask for one)
        return hipSuccess;
    };
    const bool jac = a.mode == 3;       // OCEAN_MODE_JACOBIAN: the height role works on pair 3, C rows per workgroup
    const unsigned hb = jac ? nb : hb_b;
    const dim3 blk(G::T_C);
#define OCEAN_XPASS2(kern, grid, lds, ev, nts, z16, args)                                                          \
        do { if (jac) launch(kern<N, C, G::T_C, typename G::PC, nts, z16, true>, grid, blk, lds, st, ev, args);       \
             else launch(kern<N, C, G::T_C, typename G::PC, nts, z16, false>, grid, blk, lds, st, ev, args); } while (0)
#define OCEAN_XPASS(kern, grid, lds, ev, nts, args)                                                                 \
        do { if (stream_maps & 8) OCEAN_XPASS2(kern, grid, lds, ev, nts, true, args);                                 \
             else OCEAN_XPASS2(kern, grid, lds, ev, nts, false, args); } while (0)
    auto launch_xb = [&](int roles, int rec_mode) {
        dim3 gb((roles & 1 ? hb : 0u) + (roles & 2 ? nb : 0u) + (roles & 4 ? nb : 0u), tiles);
#ifdef OCEAN_XBGRID
        if (const char* ev = getenv("OCEAN_DEBUG_XB_GRID")) gb.x = (unsigned)atoi(ev);   // diagnostic: partial grid (results wrong)
#endif
        if (roles & 2) {
            ocean_launch_info& li = c->last_launch[1];
            li.tile_size = N; li.grid_x = gb.x; li.grid_y = tiles; li.block = G::T_C; li.mode = (uint32_t)a.mode;
            li.per_workgroup = C;
            li.lds_bytes = (uint32_t)lds_b;
            li.flags = ((stream_maps & 1) ? OCEAN_LAUNCH_NT_MAPS : 0u) | ((stream_maps & 8) ? OCEAN_LAUNCH_HALF_INTER : 0u) |
                       (jac ? OCEAN_LAUNCH_JACOBIAN : 0u) | (ramp_b ? OCEAN_LAUNCH_STAGGERED_START : 0u) | (split ? OCEAN_LAUNCH_SPLIT_ORDER : 0u) |
                       ((roles & 4) ? OCEAN_LAUNCH_MERGED_X : 0u);
            if (roles & 4) {                     // no k_xpass_disp this frame: its record describes the launch that did its work
                c->last_launch[2] = li;
                c->last_launch[2].flags = (li.flags & ~(uint32_t)OCEAN_LAUNCH_NT_MAPS) | ((stream_maps & 2) ? OCEAN_LAUNCH_NT_MAPS : 0u);
            }
        }
        hipEvent_t* mb = next_marks(1);
#ifdef OCEAN_XB_TRACE
        {   // diagnostic: this translation unit's copy of the trace pointer, set when the context's buffer changes (not per frame)
            static unsigned long long* armed = nullptr;
            if (armed != c->stamps) { armed = c->stamps; (void)hipMemcpyToSymbol(HIP_SYMBOL(ocean::g_xb_trace), &armed, sizeof(armed)); }
        }
#endif
        FrameArgs ba = a;
        ba.xb_roles = roles; ba.rec_mode = rec_mode;
        // the DISP workgroups of a merged launch poll the HEIGHT count; beside other chains' launches a poll every ~3 us instead of back to back
        // takes the pollers off the memory system (512^2 depth 4: 7.66-7.97 -> 7.35-7.55 us per frame, 256^2 7.8-9.7 -> 7.2-7.3); a serial frame
        // (<= 128^2) keeps the tight poll -- there the wait IS the frame's latency
        ba.poll_sleep = ((roles & 4) && !alone) ? 1 : 0;
#ifdef OCEAN_DEVELOPER
        if (const char* ps = getenv("OCEAN_POLL_SLEEP")) ba.poll_sleep = atoi(ps);
#endif
        ba.start_ramp = ramp_b;             // over its normal-map workgroups (the height workgroups start at once)
        if (stream_maps & 1) OCEAN_XPASS(k_xpass_b, gb, lds_b, mb, true, ba);
        else OCEAN_XPASS(k_xpass_b, gb, lds_b, mb, false, ba);
    };
    auto launch_xd = [&](int rec_mode) {
        dim3 gd(nb, tiles);
#ifdef OCEAN_XBGRID
        if (const char* ev = getenv("OCEAN_DEBUG_XD_GRID")) gd.x = (unsigned)atoi(ev);
#endif
        {
            ocean_launch_info& li = c->last_launch[2];
            li.tile_size = N; li.grid_x = gd.x; li.grid_y = tiles; li.block = G::T_C; li.mode = (uint32_t)a.mode;
            li.per_workgroup = C;
            li.lds_bytes = (uint32_t)lds_m;
            li.flags = ((stream_maps & 2) ? OCEAN_LAUNCH_NT_MAPS : 0u) | ((stream_maps & 8) ? OCEAN_LAUNCH_HALF_INTER : 0u) |
                       (jac ? OCEAN_LAUNCH_JACOBIAN : 0u) | (ramp_d ? OCEAN_LAUNCH_STAGGERED_START : 0u) | (split ? OCEAN_LAUNCH_SPLIT_ORDER : 0u);
        }
        hipEvent_t* md = next_marks(2);
        FrameArgs da = a;
        da.rec_mode = rec_mode;
        da.start_ramp = ramp_d;             // its 257 workgroups are one per CU: the same read-then-write burst
        if (stream_maps & 2) OCEAN_XPASS(k_xpass_disp, gd, lds_m, md, true, da);
        else OCEAN_XPASS(k_xpass_disp, gd, lds_m, md, false, da);
    };
    if constexpr (HAS_ONE) {
        if (one_launch) {
            constexpr size_t lds_one = lds_rows > lds_b ? lds_rows : lds_b;
            auto kern = k_frame<N, C, G::T_C, typename G::PR, typename G::PC, true, false, true>;
            if (c->attr_one_n != (uint32_t)N) {
                if ((e = allow_lds(kern, lds_one)) != hipSuccess) return e;
                c->attr_one_n = (uint32_t)N;
            }
            FrameArgs fa = a;
            fa.zmask = 15; fa.xb_roles = 7; fa.rec_mode = rec_last; fa.start_ramp = 0;
            fa.zdone_target = (++c->zgen[c->cur_set]) * (unsigned)(N / 2 + 1);
            fa.poll_sleep = 0;
#ifdef OCEAN_DEVELOPER
            if (const char* ps = getenv("OCEAN_POLL_SLEEP")) fa.poll_sleep = atoi(ps);
#endif
            const dim3 grid((N / 2 + 1) + hb_b + 2 * nb, tiles), block(G::T_C);
            for (int k = 0; k < 3; ++k) {
                ocean_launch_info& li = c->last_launch[k];
                li.tile_size = N; li.grid_x = grid.x; li.grid_y = tiles; li.block = G::T_C; li.mode = (uint32_t)a.mode;
                li.per_workgroup = k == 0 ? 1u : (uint32_t)C;
                li.lds_bytes = (uint32_t)lds_one;
                li.flags = OCEAN_LAUNCH_ONE_LAUNCH | (k == 0 ? 0u : (OCEAN_LAUNCH_NT_MAPS | OCEAN_LAUNCH_MERGED_X));
            }
            launch(kern, grid, block, lds_one, st, next_marks(0), fa);
            if (c->after_z && (e = hipEventRecord(c->after_z, st)) != hipSuccess) return e;
            c->launch_count = launches;
            return hipGetLastError();
        }
    }
    if (!split) {
        if ((e = launch_z(15)) != hipSuccess) return e;
        if (c->after_z && (e = hipEventRecord(c->after_z, st)) != hipSuccess) return e;   // the first frames after a drain: the next chain's z pass starts behind this one (ocean_api.hip)
#ifdef OCEAN_STAMPS
        if (getenv("OCEAN_DEBUG_ONLY_ZPASS")) return hipGetLastError();
        arm(1);
#endif
        if (merged_x) {
            launch_xb(7, rec_last);
        } else {
            launch_xb(3, 0);
            // the normal map is final behind this launch: ocean_compute_waves_read starts its device-to-host copy here, beside the displacement pass
            if (c->after_b) { if ((e = hipEventRecord(c->after_b, st)) != hipSuccess) return e; c->after_b_recorded = true; }
#ifdef OCEAN_STAMPS
            arm(2);
#endif
            launch_xd(rec_last);
        }
    } else {
        if ((e = launch_z(8 | 1)) != hipSuccess) return e;
        if (c->after_z && (e = hipEventRecord(c->after_z, st)) != hipSuccess) return e;
        launch_xb(1, 0);
        launch_xd(rec_last == 1 ? 1 : 0);                   // (early records may go out here: the height keys are final; counted ones come from the last launch)
        if ((e = launch_z(2 | 4)) != hipSuccess) return e;
        launch_xb(2, rec_last == 2 ? 2 : 0);
    }
#undef OCEAN_XPASS2
#undef OCEAN_XPASS
    c->launch_count = launches;
    return hipGetLastError();
}

