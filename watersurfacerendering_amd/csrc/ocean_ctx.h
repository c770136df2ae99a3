// ocean_ctx.h -- the context behind the opaque ocean_t of include/ocean.h, shared by the translation units of
// libocean_hip.so (ocean_api.hip: the C ABI; frames_*.hip: the per-size frame launchers).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>      // types and prototypes only: the library is loaded with dlopen when a communicator is asked for

#include <vector>

#include "../../include/ocean.h"
#include "../../include/ocean_consumers.h"
#include "../../include/ocean_dev.h"
#include "ocean_kernels.h"

constexpr int MAXD = 8;     // maximum pipeline depth (independent frame chains)
constexpr int OCEAN_MAX_LAUNCHES = 5;   // launches per frame: three (five in the developer-only split order: ocean_launch.h)

// ---- launch and store-policy heuristics: every measured constant in ONE place -----------------------------------------
// The rules are written in quantities of the device and of the launch -- compute units x resident workgroups per unit (asked of the runtime:
// hipOccupancyMaxActiveBlocksPerMultiprocessor), the memory-side cache, the bytes a launch moves -- not in tile sizes; rounds 2-5 had them as
// `N == 2048 && tiles == 1`, 500 / 450 / 900 ticks and literal 200 / 300 MB.  Each constant cites the measurement that fitted it.
struct OceanTuning {
    // Memory-side (Infinity) cache of the device.  HIP has no query for it; MI355X_MICROARCH.md: 256 MiB.
    double cache_bytes = 268435456.0;
    // The maps are written once and never re-read by the pipeline: streamed past the cache (non-temporal stores) whenever frames are pipelined,
    // and for a serial frame once the two maps of the frame alone exceed this fraction of the cache -- 200 MB: a serial 2048^2 frame (134 MB)
    // is faster with plain stores (76 vs 80 us), a serial 8 x 1024^2 batch (268 MB) with streamed ones (124 vs 150 us); ocean_api.hip.
    double maps_stream_frac = 0.745;
    // Pipelined frames: what every frame re-reads -- spectrum + the intermediates of every chain in flight -- against the cache; beyond this
    // fraction (300 MB) the intermediates are streamed too (2048^2 depth 3, 243 MB: 57-58 plain vs 59-60 streamed; depth 4, 310 MB: 61 vs
    // 58.5; profiles/r02_layout_experiments.txt).
    double inter_stream_frac = 1.118;
    // Staggered start (ocean_kernels.h: start_ramp_wait).  A launch whose whole grid is resident at once on an otherwise idle device is a load
    // burst followed by a store burst; spreading the workgroups' starts lets the early ones store while the late ones load.  The spread is a
    // fraction of the launch's expected duration -- its algorithmic bytes at a nominal rate: 0.27 x bytes / 5.5 TB/s = 4.7 / 4.9 / 4.5 us for
    // the three launches of a 2048^2 frame (k_xpass_b: its normal-map workgroups and their 24 B/texel), inside the flat optimum of 4-5 us
    // measured for each (profiles/r04_zpass_experiments.txt item 10; round 5 re-measured the z pass's with write-through stores: flat from
    // 2 to 5 us; rounds 4-5 shipped 5.0 / 4.5 / 4.5) ...
    double ramp_frac = 0.27;
    double ramp_rate_bytes_per_s = 5.5e12;
    // ... twice that for the x passes of PIPELINED frames, which run beside the same launches of the other chains (item 11: 9 us; now 9.9 / 9.1) ...
    double ramp_pipelined_x = 2.0;
    // ... and only for launches over at least this many texels: below it the bursts are too short to be worth a wait (1024^2, 1 Mi texels:
    // z pass 14.5-15.3 -> 14.8-15.3, displacement pass 7.4-8.5 -> 8.2-9.8 us: a loss; 2048^2, 4 Mi: -1 us per launch in every mode and precision,
    // r04 item 12 -- which is why the bound is on texels, not bytes: the half2 / fp16-spectrum launches of a 2048^2 tile move 50 MB and gain too).
    double ramp_min_texels = 4.0e6;
    // ... and whose resident round is at most this many workgroups per compute unit: a unit that is handed ten small workgroups staggers them by
    // itself, and a rule that trusted the occupancy query alone lost 12 % on serial 5 x 1024^2 frames (z pass 33 -> 39-43 us; 3 and 4 tiles:
    // -1 ... +1 %; round 6, profiles/r06_tuning_rules.txt).  The launches that gain hold 1-5 per unit.
    unsigned ramp_max_wg_per_cu = 6;
    // ocean_compute_waves_read: maps up to this size are stored to the page-locked destination by the x passes themselves (through its device
    // address, beside the device copy), larger ones go through the runtime's DMA engines behind their kernels.  Stores from kernels cross PCIe
    // at 52.5 GB/s, the DMA engines reach 45.5-47.9 at 2 x 4 MiB, 53.7-54.4 at 2 x 16 MiB and 56.3 at 2 x 64 MiB (kernel stores: 53.6, 54.3):
    // tools/ubench/d2h.hip, round 6.
    size_t host_store_max_bytes = (size_t)8 << 20;
    // Placement search at Prepare (round 6, profiles/r06_slow_window.txt item 7): how fast the z pass runs is a property of WHERE its spectrum
    // and intermediates were allocated -- contexts created one after the other in one process, same code, same second: 19.8 / 21.6 / 23.1 /
    // 26.7 / 28.6 us per 2048^2 z pass, each stable for the context's life.  Like the reference's own Prepare, which lets FFTW_MEASURE time
    // candidate plans (WSTessendorf.cpp:191-232), ocean_prepare therefore times `placement_trials` candidate allocations of that buffer group
    // (all held at once, so that they ARE different memory) on serial frames and keeps the fastest.  From this tile size up, while one group
    // stays below the byte bound; ocean_set_placement_search(ctx, n) overrides (1 = off).
    int placement_trials = 6;
    unsigned placement_min_n = 2048;            // (1024^2 and batches of smaller tiles: contexts differ by 2-7 %, the search buys 0.5-1.3 %: profiles/r06_slow_window.txt item 10)
    size_t placement_max_group_bytes = (size_t)512 << 20;
    unsigned placement_mask = 0x1f;               // which buffers differ between candidates: bit 0 spectrum, 1-2 dispersion, 3-5 chain 0's intermediates, 6 its maps
    // Merged x pass / one-launch frame (in-launch hand-offs): only where every workgroup of the grid has a compute unit to itself
    // (MI355X_MICROARCH.md, inter-workgroup visibility: the regime the recipe is measured for).
    unsigned handoff_wg_per_cu = 1;
};
// bytes per texel the three launches of a frame move (fp32 FULL7: 23 / 28 / 22; ocean_algorithmic_bytes_per_launch has the derivation)
inline void ocean_launch_bytes_per_texel(bool half_inter, bool jacobian, bool h0_half, bool omega16, int out[3])
{
    const int h0 = h0_half ? 4 : 8, w = omega16 ? 1 : 2;
    const int z_out = jacobian ? (half_inter ? 8 : 16) : (half_inter ? 7 : 14);
    const int pair0 = half_inter ? 2 : 4;
    out[0] = h0 + w + z_out;
    out[1] = (z_out - pair0) + 2 + 16 + (jacobian ? 4 : 0);
    out[2] = pair0 + 2 + 16 + (jacobian ? 4 : 0);
}

struct ocean_ctx {
    OceanTuning tune;               // the heuristics' constants (defaults above; nothing changes them at run time)
    int occ_n = 0;                  // tile size occ_* were asked for: resident workgroups per compute unit of the single-transform z pass and of
    int occ_z = 0, occ_z_all = 0, occ_b = 0, occ_d = 0;    //   (both instantiations) and of the two x passes on this device (hipOccupancyMaxActiveBlocksPerMultiprocessor; ocean_launch.h)
    uint32_t n = 0;
    uint32_t tiles = 0;
    int device = 0;
    bool prepared = false;
    // At pipeline depth D asynchronous frames rotate over D chains, each with its own
    // stream, intermediates and internal map set and no dependency on the others, so the
    // z pass of one frame overlaps the map passes of the others.  A caller-supplied
    // stream, caller-bound output or depth 1 runs everything on one stream / set 0.
    hipStream_t own[MAXD] = {};
    hipStream_t user = nullptr;
    int depth = 1;
    uint64_t frame_ctr = 0;
    bool have_frame = false;        // a frame has been enqueued since the last ocean_prepare
    int dispersion = 0;             // ocean_set_dispersion
    float dispersion_param = 0.0f;
    int last_set = 0;
    int cu_count = 0;               // compute units of the device
    bool start_ramp = true;         // ocean_set_start_ramp: the staggered start may be used where it applies (ocean_launch.h)
    bool z_write_through = true;    // (developer A/B: the serial frames' write-through intermediates may be used where the form exists)
    bool merged_x = true;           // ocean_set_merged_xpass: the one-launch x pass may be used where it applies (ocean_launch.h)
    uint32_t attr_n = 0;            // tile size whose kernels had their LDS attribute set through this context
    bool lambda_uniform = true;
    bool lambda_dirty = true;       // host lambdas newer than the device array (uploaded by the next frame)

    std::vector<ocean_params> params;
    uint64_t seed = 0;
    // device state
    float2* h0 = nullptr;
    float* omega = nullptr;
    uint16_t* omega_q = nullptr;    // omega / base_freq as 16-bit integers (what the frame kernels read)
    float* base_freq = nullptr;     // [tiles]
    unsigned* omega_q_overflow = nullptr;
    bool omega16 = false;           // every multiple fits 16 bits (decided at ocean_prepare)
    float* k1d = nullptr;
    float2* tw = nullptr;
    float2* z[MAXD] = {};
    float2* zh[MAXD] = {};
    float* hraw[MAXD] = {};
    float2* z3[MAXD] = {};         // OCEAN_MODE_JACOBIAN intermediates: pair 3, cross derivative, product of the diagonal factors --
    float* jraw[MAXD] = {};        // allocated by a chain's first frame of that mode (alloc_jacobian), never by the others
    float* jac0[MAXD] = {};
    unsigned* minmax[MAXD] = {};
    unsigned* zdone[MAXD] = {};     // [tiles] z-pass workgroups that have finished, counted up for ever (one-launch frames: FrameArgs::zdone)
    unsigned zgen[MAXD] = {};       // one-launch frames the chain has run (zdone's target = zgen x (N/2 + 1))
    int cur_set = 0;                // the chain of the frame being enqueued (for the launcher)
    uint32_t attr_one_n = 0;        // tile size whose one-launch kernel had its LDS attribute set through this context
    unsigned* hdone[MAXD] = {};     // [tiles] HEIGHT workgroups of the chain's current frame that have finished (merged x pass: FrameArgs::hdone)
    uint4* done_rec[MAXD] = {};     // [tiles] host-coherent completion records (min key, max key, sequence, 0) written by the last
                                    //   workgroup of a frame's last kernel (ocean_kernels.h: frame_done)
    unsigned* done_ctr[MAXD] = {};  // device counters of that kernel's finished workgroups (two levels: ocean_kernels.h, frame_done)
    bool tracked[MAXD] = {};        // the chain's most recent frame counts its finished workgroups (completion records written LAST:
                                    //   ocean_wait_frame polls); otherwise the records only carry the height keys and the wait is a stream synchronisation
    bool track_async = false;       // ocean_set_frame_tracking: asynchronous frames are tracked too (the synchronous call always is)
    unsigned seq[MAXD] = {};        // sequence number of the chain's most recently ISSUED frame: monotonic while the chain's record buffer lives, committed even
                                    //   when the enqueue failed half way (a reused number could match a stale completion record: ADVICE r05)
    bool frame_valid[MAXD] = {};    // the chain's most recent enqueue succeeded: its records, keys and maps describe ONE frame
    float last_t[MAXD] = {};        // time, regime (pipelined or not) and form of the chain's most recent frame: what recover_fault (ocean_api.hip) needs to
    bool last_pipe[MAXD] = {};      //   run it again
    bool last_handoff[MAXD] = {};   // ... it used an in-launch hand-off (merged x pass / one-launch frame: the only forms whose waits can give up)
    bool handoff = false;           // set by launch_frame: the frame it has just enqueued uses an in-launch hand-off
    bool recovering = false;        // recover_fault is re-enqueueing (no recursion)
    int placement_override = 0;     // ocean_set_placement_search: 0 = the library's rule (OceanTuning), n >= 1 = that many candidates (1 = off)
    bool placement_done = false;        // the search has run (or was found unnecessary) for the buffers this context holds now
    int placement_tried = 0;        // what the most recent ocean_prepare did: candidates timed (0: no search), and their serial frame times
    float placement_us_chosen = 0.0f, placement_us_worst = 0.0f;
    unsigned fault_recoveries = 0;  // recoveries after an in-launch wait gave up (ocean_fault_recoveries)
    unsigned fault_recoveries_seen = 0;   // ... that ocean_compute_waves_read has accounted for
    float4* dispN[MAXD] = {};      // internal map sets (set 0 always; others on first use): ONE allocation per set,
    float4* nrmN[MAXD] = {};       //   [displacement maps of all tiles | normal maps of all tiles] (nrmN points into dispN's)
    size_t maps_bytes[MAXD] = {};  // size of that allocation (a whole number of 2 MiB pages: ocean_export_maps)
    float4* ext_disp = nullptr;
    float4* ext_nrm = nullptr;
    bool maps_shared = false;      // the internal maps are visible outside the context's streams (ocean_export_maps, ocean_device_maps) until they
                                   //   are re-allocated: ocean_wait_frame / ocean_compute_waves then synchronise the stream (ocean_api.hip: wait_frame)
    hipExternalMemory_t import_mem = nullptr;   // ocean_bind_output_dmabuf: the imported memory object ext_disp / ext_nrm point into
    void* import_base = nullptr;
    float* toff = nullptr;
    bool use_toff = false;
    std::vector<float> toff_host;   // the per-tile time offsets as set by the caller (re-uploaded when the device buffers are re-created)
    float* lambda = nullptr;
    ocean::TileParams* tparams = nullptr;
    float2* xi = nullptr;          // injected or generated draws (kept for read-back)
    int mode = 0;                  // OCEAN_MODE_*
    int inter_bits_zeroed = 32;    // layout the intermediates' padding was last zero-filled for
    int inter_bits = 32;           // 32, or 16: the z-pass outputs (z, zh) are stored as scaled half2 (ocean_set_intermediate_precision)
    float4* zscale = nullptr;      // [tiles][2]: (s_u, s_k, 1/s_u, 1/s_k), (s_3, g, 1/s_3, 1/g) -- FrameArgs::zscale
    unsigned* zbounds = nullptr;   // [tiles][2] float bits of the column-sum bounds (k_inter_bounds)
    int h0_bits = 32;              // 32, or 16: frames read a scaled half2 copy of h0
    __half2* h0h = nullptr;
    float* h0_inv_scale = nullptr;
    unsigned* h0_maxbits = nullptr;
    unsigned* h_minmax = nullptr;  // pinned
    unsigned* fault = nullptr;     // host-coherent word the kernels' in-launch waits set when they give up (FrameArgs::fault); cleared by recover_fault
    float4* grid_pos = nullptr;     // vertex-stage consumer output (ocean_displace_grid)
    float4* grid_nrm = nullptr;
    uint32_t grid_vertices = 0, grid_capacity = 0;
    float4* mips_disp = nullptr;    // mip chain of one tile's maps (ocean_build_mips): levels 1 .. log2 N, tightly packed
    float4* mips_nrm = nullptr;
    uint32_t mips_n = 0;            // tile size the two buffers were allocated for
    bool mips_ready = false;
    unsigned long long* stamps = nullptr;   // diagnostic builds only
    hipEvent_t start_ev = nullptr;      // ocean_time_frames: start of the timed region
    hipEvent_t end_ev[MAXD] = {};       //                    end of every chain
    hipEvent_t mark_ev[MAXD][2 * OCEAN_MAX_LAUNCHES] = {};   // per-launch timing: (start, stop) of each launch of a frame
    int launch_count = 3;               // launches of the most recent frame (3; 5 in the developer-only split order) and the kernel (0 z pass, 1 k_xpass_b,
    int launch_kernel[OCEAN_MAX_LAUNCHES] = {0, 1, 2, 0, 1};   //   2 k_xpass_disp) each of them ran: ocean_time_frames sums a kernel's launches
    ocean_launch_info last_launch[3] = {};  // what the most recent frame launched (ocean_last_launch)
    hipEvent_t z_done[MAXD] = {};       // pipelined frames right after a drain: recorded behind a chain's z pass (see enqueue_frame)
    hipEvent_t after_z = nullptr;       // what launch_frame records behind the z pass of the frame being enqueued (null: nothing)
    hipEvent_t after_b = nullptr;       // ... and behind k_xpass_b, where the frame has a displacement pass of its own behind it (ocean_compute_waves_read)
    bool after_b_recorded = false;
    hipStream_t copy_stream = nullptr;  // ocean_compute_waves_read: the normal map's device-to-host copy runs here, beside the displacement pass
    hipEvent_t nrm_final = nullptr, copy_done[2] = {};   // its events: normal map final / the two copies have landed
    float4* host_out[2] = {};           // set around its enqueue: device addresses of the caller's page-locked (displacement, normal) arrays when the
                                        //   x passes are to store the maps there themselves (FrameArgs::disp_host / nrm_host), null otherwise
    int burst_pos = 0;                  // pipelined frames enqueued since the context's streams were last drained
    int z_last_set = -1;
    hipEvent_t consumer_ev = nullptr;   // behind the most recent consumer launch (mips, grid): the context-wide output buffers of
    hipStream_t consumer_stream = nullptr;  // those are re-used, so a consumer launch on ANOTHER chain's stream first waits for it
    bool consumer_pending = false;
    // ---- packed-map gather over RCCL (ocean_comm_init / ocean_gather_maps)
    ncclComm_t comm = nullptr;
    int comm_ranks = 0, comm_rank = -1;
    hipStream_t comm_stream = nullptr;
    hipEvent_t frame_done[MAXD] = {};   // recorded on a chain's stream behind the frame whose maps are gathered
    hipEvent_t gather_done[MAXD] = {};  // recorded on the communication stream behind that gather
    bool gather_pending[MAXD] = {};
    uint2* pack_half[MAXD][2] = {};    // half-precision copies of a chain's two maps for ocean_gather_maps_f16 (allocated on first use)     // the chain's next frame must wait for gather_done before rewriting the maps
};


inline hipStream_t stream_of(const ocean_ctx* c, int set) { return c->user ? c->user : c->own[set]; }

// One frame = three launches on `st` (ocean_launch.h); one entry point per group of tile sizes, each compiled in its
// own translation unit (frames_*.hip) so that the library builds in parallel.  stream_maps: bit 0 normal map and
// bit 1 displacement map stored non-temporally, bit 2 intermediates stored non-temporally, bit 3 half2 intermediates,
// bit 4 the frame runs alone on the device (serial frames: the z pass may split its last round of columns).
// marks: 2 x OCEAN_MAX_LAUNCHES events (start, stop per launch) or null.
hipError_t ocean_launch_frame_small(ocean_ctx* c, const ocean::FrameArgs& a, int stream_maps, hipStream_t st, hipEvent_t* marks);   // 16 .. 256
hipError_t ocean_launch_frame_mid(ocean_ctx* c, const ocean::FrameArgs& a, int stream_maps, hipStream_t st, hipEvent_t* marks);     // 512, 1024
hipError_t ocean_launch_frame_2048(ocean_ctx* c, const ocean::FrameArgs& a, int stream_maps, hipStream_t st, hipEvent_t* marks);
hipError_t ocean_launch_frame_4096(ocean_ctx* c, const ocean::FrameArgs& a, int stream_maps, hipStream_t st, hipEvent_t* marks);
