/*
 * include/ocean_dev.h -- developer surface: what tests, bench.py and tools/ use and a renderer never needs (Prepare() read-backs,
 * per-kernel timing, what the last frame launched, switches of the launch heuristics).  Frames are bit-identical whatever these are set to.
 * Part of the C ABI of libocean_hip.so (include/ocean.h is the drop-in boundary; this header declares more of the same library's exports).
 */
#ifndef OCEAN_DEV_H_
#define OCEAN_DEV_H_

#include "ocean.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Optional, once after ocean_prepare: put the context's work on the fastest of the process's hardware queues.  HIP spreads a process's
 * streams over four hardware queues, and on MI355X these are not alike: every kernel of a frame differs by up to 1 us between them
 * (profiles/r03_bimodal_probe.txt; the occasional queue on which the normal-map pass took 1.5-3.5 us longer no longer
 * finds its victim: profiles/r03_xpass_trace.txt).  The call times `frames` serial frames (plus five untimed ones) on each of the context's first four
 * streams -- one per queue -- and re-orders its streams, fastest first: the serial path (the synchronous ocean_compute_waves) and pipeline
 * chain 0 then use the fastest queue, chains 1..3 the next ones.  us_per_frame (NULL or 4 floats) receives the measured frame times in
 * the new order.  50 frames tell the queues apart (4 x 55 frames: 14 ms at 2048^2, 4 ms at 512^2).  Afterwards the maps hold a calibration
 * frame (read-outs return OCEAN_E_NOT_READY until the next frame), and a stream handle fetched earlier with ocean_stream() may no longer be
 * the context's.  OCEAN_E_UNSUPPORTED with a caller-owned stream (ocean_set_stream) and with caller-bound or imported output (the
 * calibration frames must not land in memory somebody else owns).  Results of frames are unaffected: bit-identical.               */
int ocean_select_streams(ocean_t* ctx, uint32_t frames, float* us_per_frame /* [4] or NULL */);
/* Staggered start of a frame's launches (one 2048 x 2048 tile; ocean_kernels.h: start_ramp_wait): workgroup i of a launch whose grid is ONE
 * resident round waits i / G of a few microseconds before its first load, so that the early workgroups store while the late ones still load
 * (-5 % on a serial frame, -2.5 % on pipelined ones).  The library applies it only where every workgroup of the launch is resident at once on
 * THIS device (grid <= compute units x workgroups per unit); on != 0 (default) allows it, 0 switches it off for the context -- for a device
 * shared with other work, where a workgroup's wait is simply lost.  Frames are bit-identical either way.                                    */
int ocean_set_start_ramp(ocean_t* ctx, int on);
/* The x axis in ONE launch (round 5): frames of a single small tile run their height, normal-map and displacement workgroups as one grid --
 * the displacement workgroups transform at once and wait for the tile's height workgroups only before their stores -- two launches per frame
 * instead of three.  Applied where it was measured to pay (pipelined frames up to 512^2, which are bound by the rate of launches: 13-15 -> 8 us
 * per frame at depth 4; serial frames up to 128^2 only -- from 256^2 up the in-launch hand-off costs more than the kernel boundary it replaces)
 * and only where every workgroup of that grid is resident at once, one per compute unit; never in OCEAN_MODE_JACOBIAN.  on != 0 (default)
 * allows it, 0 keeps the three-launch frame.  Bit-identical either way; ocean_last_launch marks such a frame with OCEAN_LAUNCH_MERGED_X on
 * idx 1 and 2 (the same launch).  The same switch governs the step beyond it: pipelined frames of one tile up to 128^2 in the usual form run
 * as ONE launch (OCEAN_LAUNCH_ONE_LAUNCH; 64^2: 11.5 -> 3.9 us per frame at depth 4).                                                                                                          */
int ocean_set_merged_xpass(ocean_t* ctx, int on);

/* Placement search of ocean_prepare (round 6).  The speed of the frame's first pass depends on where the spectrum and the intermediates happen to
 * be allocated -- contexts created back to back in one process run the SAME 2048 x 2048 z pass in 19.8 ... 28.6 us, each stable for the
 * context's life (profiles/r06_slow_window.txt) -- so ocean_prepare, like the reference's Prepare with its FFTW_MEASURE plans
 * (WSTessendorf.cpp:191-232), measures: from 2048 x 2048 up it allocates a few candidate copies of that buffer group, times serial frames on
 * each and keeps the fastest (6 candidates, ~45 ms at 2048 x 2048; frames are bit-identical wherever the buffers are).  trials = 0: the
 * library's rule; 1: off; n: that many candidates at any size.  Takes effect at the next ocean_prepare.  The search runs ONCE per allocation:
 * a repeated ocean_prepare on the same buffers keeps the placement (and the report); a resize, or another `trials`, searches again.
 * ocean_placement_report: what that search did -- candidates timed (0: none) and the serial frame time of the chosen and of the slowest
 * candidate (us).                                                                                                                            */
int ocean_set_placement_search(ocean_t* ctx, int trials);
int ocean_placement_report(const ocean_t* ctx, int* trials, float* us_chosen, float* us_worst);

/* ---- introspection for tests and the bench -------------------------------- */
/* Copies the Prepare() products of one tile to host: h0 (N*N*2), omega (N*N).   */
int ocean_read_spectrum(ocean_t* ctx, uint32_t tile, float* h0, float* omega);
/* Copies the generated gaussian draws of one tile (N*N*2).                      */
int ocean_read_xi(ocean_t* ctx, uint32_t tile, float* xi);

/* Times `frames` back-to-back asynchronous frames (t = t0 + j*dt) with HIP events after
 * `warmup` untimed ones, at the context's pipeline depth.  ms_total = whole timed region;
 * ms_kernel[3] = mean duration per kernel and frame, in ocean_kernel_name order (a kernel that the split frame order
 * launches twice reports the sum of its two launches), from events
 * bracketing every launch on its own stream during a second, separately timed run of
 * the same frames in the same regime (at depth > 1 the launches of different frames
 * overlap, so these are durations under that concurrency).  Any output pointer may be NULL. */
int ocean_time_frames(ocean_t* ctx, float t0, float dt, int warmup, int frames,
                      float* ms_total, float* ms_kernel /* [3] */);

/* Name of the idx-th launch (0..2) of one frame, in the order ocean_time_frames reports
 * them: "k_zpass", "k_xpass_b", "k_xpass_disp" at every tile size.  NULL if idx is out of range. */
const char* ocean_kernel_name(const ocean_t* ctx, int idx);

/* What the most recent frame launched: the idx-th launch's (ocean_kernel_name order) tile size, grid, block and kernel
 * variant.  The host picks a kernel instantiation per frame from the tile size, the precisions, the mode, the pipeline
 * depth and the batch size (store policy, columns per z-pass workgroup, split last round); tests use this to prove that
 * every variant the launcher can select has met the oracle (tests/test_variants_gpu.py).                              */
enum {
    OCEAN_LAUNCH_NT_MAPS         = 1,    /* x passes: maps stored non-temporally (template flag NTS)                    */
    OCEAN_LAUNCH_NT_INTER        = 2,    /* z pass: intermediates stored non-temporally (ZNT)                           */
    OCEAN_LAUNCH_HALF_INTER      = 4,    /* half2 intermediates (Z16), all three                                        */
    OCEAN_LAUNCH_JACOBIAN        = 8,    /* OCEAN_MODE_JACOBIAN: x passes' JAC instantiations, z pass's pair-3 branch   */
    OCEAN_LAUNCH_FP16_SPECTRUM   = 16,   /* z pass reads the half2 copy of h0 (wave-uniform branch, no instantiation)   */
    OCEAN_LAUNCH_FP32_DISPERSION = 32,   /* z pass reads the fp32 dispersion array: some multiple of the base frequency
                                            needs more than 16 bits (wave-uniform branch)                               */
    OCEAN_LAUNCH_SPLIT_LAST_ROUND = 64,  /* (rounds 2-3: the columns of a partially filled last round of z-pass workgroups split
                                            over two workgroups each; never set since round 4 -- the single-transform form
                                            replaced it -- the value stays reserved)                                    */
    OCEAN_LAUNCH_SINGLE_TRANSFORM = 128, /* z pass: one transform per batch, half the threads (k_zpass_c1): two independent
                                            workgroups per CU where the two-transform forms fit only one (4096^2)       */
    OCEAN_LAUNCH_STAGGERED_START = 256,  /* not a variant (same instantiation, same bits): the launch's workgroups, all resident
                                            at once, start spread over a few microseconds so that the early ones store while
                                            the late ones still load -- the three launches of a frame of one 2048^2 tile
                                            (serial and pipelined frames with ramps of their own), nowhere else          */
    OCEAN_LAUNCH_SPLIT_ORDER     = 512,  /* developer builds only (never set by the shipped library): the frame ran in the split order of
                                            profiles/r05_4096_experiments.txt -- z pass and k_xpass_b twice, each time half their work        */
    OCEAN_LAUNCH_WT_INTER        = 2048, /* z pass: fp32 intermediates stored write-through (`sc1`: they leave the XCD's L2 as they are written, no
                                            end-of-kernel write-back burst) -- serial frames at 1024^2 (batches) and 2048^2, single-transform form */
    OCEAN_LAUNCH_ONE_LAUNCH      = 4096, /* the whole frame ran as ONE launch (k_frame: z-pass, height, normal-map and displacement workgroups in one
                                            grid, one-way hand-offs inside it): pipelined frames of one tile up to 128^2 in the usual form; idx 0..2
                                            then describe that one launch                                                                    */
    OCEAN_LAUNCH_MERGED_X        = 1024  /* not a kernel variant: k_xpass_b ran the displacement workgroups as well (one launch for the whole x axis,
                                            no k_xpass_disp); set on idx 1 and idx 2, which then describe that one launch                   */
};
typedef struct ocean_launch_info {
    uint32_t tile_size;
    uint32_t grid_x, grid_y, block;
    uint32_t lds_bytes;
    uint32_t flags;            /* OCEAN_LAUNCH_*                                                                         */
    uint32_t per_workgroup;    /* z pass: spectrum columns per workgroup (1 or 2); x passes: map rows per workgroup     */
    uint32_t mode;             /* OCEAN_MODE_* of the frame                                                              */
} ocean_launch_info;
int ocean_last_launch(const ocean_t* ctx, int idx, ocean_launch_info* out);

/* HBM bytes per texel this pipeline has to move for one seven-field frame at the context's
 * precision settings (73 with the fp32 spectrum: 8 + 1 in -- the 16-bit dispersion is read for half of the columns, a
 * column and its point mirror share it --, 14 + 14 half-size intermediates out and in, 2 + 2 raw height, 32 maps).  SURVEY.md section 8d prices a plain 3.5-transform
 * two-pass scheme at 108; bench.py reports that figure separately, labelled as a model.     */
int ocean_algorithmic_bytes_per_texel(const ocean_t* ctx);
/* The same figure per launch (idx in ocean_kernel_name order; 23 / 28 / 22 for the fp32 seven-field frame): what
 * bench.py's roofline divides by a kernel's duration.  0 if idx is out of range.                                        */
int ocean_algorithmic_bytes_per_launch(const ocean_t* ctx, int idx);

#ifdef __cplusplus
}
#endif
#endif /* OCEAN_DEV_H_ */
