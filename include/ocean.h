/*
 * include/ocean.h -- C ABI of libocean_hip.so, the MI355X (gfx950) Tessendorf
 * FFT ocean synthesiser that replaces the reference's CPU path
 *     class WSTessendorf   (/root/reference/src/scene/WSTessendorf.{h,cpp})
 * behind the same "Prepare / ComputeWaves / read displacement + normal map"
 * contract.  The reference has no FFI layer (a plain C++ class with one caller,
 * src/scene/WaterSurfaceMesh.cpp); this header is the boundary a maintainer
 * binds instead, and include/WSTessendorf.hpp is the source-compatible C++
 * adaptor on top of it.  Each entry point cites the reference interface it
 * replaces.  Plain pointers and sizes only; no exceptions cross the ABI.
 *
 * Conventions
 *   - every function returns OCEAN_OK (0) or a negative OCEAN_E_* code unless
 *     documented otherwise; ocean_strerror() names them.
 *   - one context = T independent tiles of the same size N on one device,
 *     one HIP stream; a context is not re-entrant (like the reference object,
 *     which shares FFTW buffers: WSTessendorf.cpp:164-171).
 *   - maps are tightly packed row-major N x N RGBA32F, texel (m, n) at
 *     float offset 4*(m*N + n): exactly the layout WaterSurfaceMesh.cpp:701-755
 *     memcpy's into its staging buffer.
 *       displacement = (lambda*Dx, height/A, lambda*Dz, 1)   WSTessendorf.cpp:380-412,443-455
 *       normal       = (dh/dx, dh/dz, dDx/dx, dDz/dz)        WSTessendorf.cpp:414-437
 *   - there is NO CPU fallback: every call fails with OCEAN_E_NO_DEVICE /
 *     OCEAN_E_HIP when no gfx950 device is usable.
 *
 * This header is the drop-in boundary (lifetime, properties, Prepare, ComputeWaves, read-out, the upload path of 8f rank 1,
 * the output modes, the RCCL gather of 8e).  Two more headers declare the rest of what libocean_hip.so exports:
 *   include/ocean_consumers.h   SURVEY.md 8f ranks 3-4: the vertex-stage consumer, cascades, the mip chain
 *   include/ocean_dev.h         what tests, bench.py and the A/B tools under tools/ use and a renderer never needs: Prepare() read-backs,
 *                               per-kernel timing, what the last frame launched, the switches of the launch heuristics
 */
#ifndef OCEAN_H_
#define OCEAN_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCEAN_ABI_VERSION 5   /* 5: round 6 -- ocean_compute_waves_read, ocean_fault_recoveries added; the vertex-stage / mip consumers moved to ocean_consumers.h and the bench / A-B plumbing to ocean_dev.h (same symbols, same library); 4: round 5 -- ocean_build_id, ocean_set_external_readers, ocean_set_start_ramp, ocean_set_merged_xpass (additions only); 2: round 2 -- gather, Jacobian mode, half2 intermediates, staging read-out, cascades, mips; 3: round 3 -- ocean_wait_frame,
                                 ocean_set_frame_tracking, ocean_last_launch, ocean_export_maps, ocean_bind_output_dmabuf, ocean_select_streams,
                                 ocean_comm_count, ocean_algorithmic_bytes_per_launch (additions only) */

enum {
    OCEAN_OK            =  0,
    OCEAN_E_INVALID     = -1,   /* bad argument (null, non power-of-two size, ...)     */
    OCEAN_E_NO_DEVICE   = -2,   /* no HIP device / wrong architecture                   */
    OCEAN_E_HIP         = -3,   /* a HIP runtime call failed; see ocean_last_hip_error  */
    OCEAN_E_NOT_READY   = -4,   /* ocean_compute_waves before ocean_prepare             */
    OCEAN_E_NOMEM       = -5,
    OCEAN_E_UNSUPPORTED = -6,   /* tile size outside [16, 4096]; export of caller-bound maps */
    OCEAN_E_COMM        = -7    /* RCCL call failed / librccl missing; see ocean_last_rccl_error */
};

typedef struct ocean_ctx ocean_t;

/* Physical parameters of one tile; defaults = WSTessendorf.h:36-43,181.       */
typedef struct ocean_params {
    float tile_length;      /* L, metres            (ctor arg, WSTessendorf.h:66)  */
    float wind_dir_x;       /* any non-zero vector; normalised like .cpp:476-479   */
    float wind_dir_y;
    float wind_speed;       /* clamped to >= 1e-4   (.cpp:481-484)                  */
    float anim_period;      /* T, seconds           (.cpp:486-490)                  */
    float phillips_const;   /* A                    (.cpp:492-495)                  */
    float damping;          /* l                    (.cpp:502-505)                  */
    float lambda;           /* choppiness, default -1 (.h:181, .cpp:497-500)        */
} ocean_params;

/* Fills *p with the reference defaults (L=1000, wind=(1,1), V=30, T=200,
 * A=3e-7, l=0.1, lambda=-1).                                                     */
void ocean_default_params(ocean_params* p);

const char* ocean_strerror(int code);
int         ocean_abi_version(void);
/* Content hash of the sources this library was built from (first 16 hex digits of the SHA-256 over the Makefile, the headers and the .hip
 * files of csrc/ in name order, include/ocean.h, then the build's extra definitions): the loader compares it with the hash of the sources beside it and
 * rebuilds on a mismatch, whatever the file times say (watersurfacerendering_amd/_abi.py: build); bench.py prints it.                    */
const char* ocean_build_id(void);
/* hipError_t value of the most recent failing HIP call on this thread (0 if none). */
int         ocean_last_hip_error(void);
/* Frames of single small tiles may run several dependent stages in ONE launch, later workgroups waiting for earlier ones inside it (pipelined
 * frames up to 512^2, serial ones up to 128^2; only where the whole grid is resident at once, one workgroup per compute unit).  Such a wait
 * is bounded: a workgroup that has waited 20 ms for its producers -- a device shared or time-sliced with other work can switch them out --
 * gives up, the device never hangs.  The host then RECOVERS: the next ocean_wait_frame / ocean_compute_waves / ocean_synchronize / read-out
 * drains the context, keeps the three-launch frame for this context from then on and runs the affected frames again (same time, same chain),
 * so the call returns the frame it was asked for; this counter says how often that happened (0 on a dedicated device: never observed outside
 * the fault-injection build).  Only a stream-ordered consumer enqueued behind such a frame (ocean_gather_maps, ocean_displace_grid,
 * ocean_build_mips) cannot be redone: the recovering call returns OCEAN_E_HIP (ocean_last_hip_error: hipErrorLaunchTimeOut) ONCE, the context
 * stays usable and the consumer call is the caller's to repeat.                                                                           */
unsigned    ocean_fault_recoveries(const ocean_t* ctx);

/* ---- lifetime: replaces WSTessendorf::WSTessendorf / ~WSTessendorf
 *      (WSTessendorf.cpp:13-34).  tile_size must be a power of two in
 *      [16, 4096] (.cpp:459-468 rejects non powers of two); 1 <= tiles <= 65535.
 *      device = HIP device ordinal.                                              */
int  ocean_create(ocean_t** out, uint32_t tile_size, uint32_t tiles, int device);
void ocean_destroy(ocean_t* ctx);

/* ---- properties: replace the Set.../Get... pairs (WSTessendorf.h:82-122,
 *      .cpp:459-505).  tile = index in [0, tiles) or OCEAN_ALL_TILES.  Like the
 *      reference, everything except lambda takes effect at the next
 *      ocean_prepare; lambda at the next ocean_compute_waves.                    */
#define OCEAN_ALL_TILES 0xFFFFFFFFu
int ocean_set_params(ocean_t* ctx, uint32_t tile, const ocean_params* p);
int ocean_get_params(const ocean_t* ctx, uint32_t tile, ocean_params* p);
int ocean_set_lambda(ocean_t* ctx, uint32_t tile, float lambda);
/* Re-sizes every tile (SetTileSize, .cpp:459-468); invalidates prepare.          */
int ocean_set_tile_size(ocean_t* ctx, uint32_t tile_size);
uint32_t ocean_tile_size(const ocean_t* ctx);
uint32_t ocean_tiles(const ocean_t* ctx);

/* ---- Prepare(): WSTessendorf.cpp:36-58 (wave vectors :60-85, gaussian draws
 *      :87-103, Phillips base spectrum + dispersion :105-148, FFT set-up
 *      :150-247), all on the device.  Tile i uses seed + i.  xi_or_null: host
 *      pointer to tiles*N*N*2 floats (re, im) row-major to inject the N(0,1)
 *      draws instead of generating them (the reference's own draws are
 *      clock-seeded and not reproducible).                                       */
int ocean_prepare(ocean_t* ctx, uint64_t seed, const float* xi_or_null);

/* ---- ComputeWaves(t): WSTessendorf.cpp:284-455.
 *      ocean_compute_waves        : synchronous in the reference's sense: when it
 *                                   returns the frame's work is done and every
 *                                   read-out call of this header sees it; writes
 *                                   the height amplitude A of every tile to
 *                                   out_amp[tiles] (may be NULL) -- the reference's
 *                                   return value.  See ocean_wait_frame for what a
 *                                   reader OUTSIDE the context's streams may assume.
 *      ocean_compute_waves_async  : enqueues the frame on the context's stream
 *                                   and returns; results are valid after
 *                                   ocean_synchronize (or stream order).
 *      t_offsets_or_null: device-resident per-tile time offsets are set with
 *      ocean_set_time_offsets; tile i is evaluated at t + offset[i].             */
int ocean_compute_waves(ocean_t* ctx, float t, float* out_amp);
int ocean_compute_waves_async(ocean_t* ctx, float t);
/* Waits for the most recently enqueued frame -- not for copies or gathers enqueued behind it -- and writes its height
 * amplitudes to out_amp[tiles] (may be NULL).  ocean_compute_waves = ocean_compute_waves_async + ocean_wait_frame.
 * Behind ocean_compute_waves (and behind asynchronous frames after ocean_set_frame_tracking) the wait is a short poll of
 * completion records the frame's last workgroup leaves in host-coherent memory -- no stream synchronisation, whose wake-up
 * costs 13-16 us per call at the reference's call shape, WaterSurfaceMesh.cpp:145-154; after 2 ms of polling it falls back
 * to hipStreamSynchronize.  When it returns, every workgroup of the frame has finished its work; a completion record is NOT a
 * memory fence, so whatever then reads the maps must be ordered by the stream: the read-out calls of this header and work
 * enqueued on ocean_stream() are.  A reader the stream does not order -- another stream, another API, another process -- needs
 * ocean_synchronize() (or an event / semaphore of its own behind the frame).  For the cases the library can see, it does that
 * itself: with caller-bound or imported output (ocean_bind_output, ocean_bind_output_dmabuf), after ocean_export_maps and after
 * ocean_device_maps, ocean_wait_frame and ocean_compute_waves ARE a stream synchronisation (end-of-kernel release and cache
 * write-back included) -- the fast poll is kept for contexts whose maps nobody else can see.
 * With ocean_read_maps_async / ocean_read_maps_staging enqueued BEFORE the wait, the caller gets A while the DMA of the
 * maps is still in flight: the reference's DOUBLE_BUFFERED idea (WaterSurfaceMesh.h:26-34) on the synthesis side;
 * include/WSTessendorf.hpp: ComputeWavesAsync() / Wait().                                                                */
int ocean_wait_frame(ocean_t* ctx, float* out_amp);
/* Completion records cost the frame's last kernel a count of its finished workgroups (about 1 us at 2048^2, nothing when
 * frames are pipelined), so only ocean_compute_waves asks for them by default: behind a plain ocean_compute_waves_async,
 * ocean_wait_frame is a stream synchronisation.  on != 0: asynchronous frames leave them too (what ComputeWavesAsync() of
 * include/WSTessendorf.hpp switches on).                                                                                 */
int ocean_set_frame_tracking(ocean_t* ctx, int on);
int ocean_set_time_offsets(ocean_t* ctx, const float* offsets_or_null /* tiles */);
int ocean_synchronize(ocean_t* ctx);

/* A, min, max of the most recently enqueued frame (GetMinHeight/GetMaxHeight,
 * WSTessendorf.h:91-92; A = ComputeWaves' return).  Waits for that frame
 * (like ocean_wait_frame), not for the other pipeline chains.                    */
int ocean_get_heights(ocean_t* ctx, uint32_t tile, float* amp, float* min_h, float* max_h);

/* ---- read-out: replaces GetDisplacements()/GetNormals() + the two memcpy's
 *      of WaterSurfaceMesh.cpp:701-755.  Host destinations (e.g. the mapped
 *      Vulkan staging pointer); N*N*4 floats per tile, tile-major.  Either may
 *      be NULL.  Synchronous.                                                    */
int ocean_read_maps(ocean_t* ctx, uint32_t first_tile, uint32_t num_tiles, float* disp, float* nrm);
/* ComputeWaves(t) AND the read-out of every tile's maps in one blocking call -- the reference's call shape, one ComputeWaves followed by two
 * memcpy's of the finished maps (WaterSurfaceMesh.cpp:145-154, 701-755): out_amp[tiles] (may be NULL), disp / nrm = tiles*N*N*4 floats each
 * (neither NULL).  Same results as ocean_compute_waves + ocean_read_maps(0, tiles), sooner.  With page-locked destinations
 * (ocean_host_register) and maps up to 8 MiB each -- 512 x 512, the reference's default -- the two x-axis passes store every texel to the
 * host arrays themselves, through their device addresses, beside the device copy: the maps cross PCIe while they are being produced and no
 * copy follows the frame.  Larger maps go through the runtime's DMA engines (56 GB/s at 64 MiB per map against 54 for kernel stores): the
 * normal map, final behind the frame's second launch, on a copy stream beside the displacement pass, the displacement map behind its kernel.
 * The call returns from a poll of events, not from a stream synchronisation.  Pageable destinations still work (staged, blocking copies).   */
int ocean_compute_waves_read(ocean_t* ctx, float t, float* out_amp, float* disp, float* nrm);

/* Asynchronous read-out (the upload path after ComputeWaves,
 * WaterSurfaceMesh.cpp:701-755 + vulkan/Buffer.cpp:133-155).  ocean_host_register pins a
 * caller-owned host range (e.g. the persistently mapped Vulkan staging buffer) so that
 * ocean_read_maps_async can DMA the maps of the most recently enqueued frame straight into
 * it, ordered after that frame on its stream, without blocking the caller; the copy is
 * complete after ocean_synchronize.  With an unregistered (pageable) destination the call
 * still works but degrades to a synchronous copy.                                       */
/* (Registrations are process-wide, like the runtime's own; the library keeps a guarded list of the ranges registered through it so that
 * ocean_compute_waves_read finds their device addresses without asking the runtime on every call -- the one piece of state outside a context.) */
int ocean_host_register(void* host_ptr, size_t bytes);
int ocean_host_unregister(void* host_ptr);
int ocean_read_maps_async(ocean_t* ctx, uint32_t first_tile, uint32_t num_tiles, float* disp, float* nrm);

/* The reference's staging-buffer upload in one call.
 * WaterSurfaceMesh::CopyModelTessDataToStagingBuffer (WaterSurfaceMesh.cpp:701-755) lays its persistently
 * mapped, HOST_VISIBLE staging buffer out as
 *     [ vertices | indices | pad to 16 | displacements | normals ]
 * with the maps at AlignSizeTo(verticesSize + indicesSize, 16) (.cpp:19-22, 721-724) and the normal map
 * immediately behind the displacement map (.cpp:736-738); UpdateFrameMaps then records two
 * vkCmdCopyBufferToImage from those offsets (.cpp:642-699).
 *   ocean_staging_map_offset  that offset, for the caller's VkBufferImageCopy::bufferOffset.
 *   ocean_read_maps_staging   enqueues the two device-to-host copies of `tile`'s maps of the most recently
 *                             enqueued frame to mapped_base + offset, ordered behind that frame on its stream;
 *                             returns at once (register mapped_base's range with ocean_host_register for a true
 *                             DMA; pageable memory degrades to a blocking copy).  Complete after
 *                             ocean_synchronize.  *bytes_to_flush (may be NULL) receives offset + 2 * N*N*16:
 *                             the range the reference then flushes from offset 0 with
 *                             vkFlushMappedMemoryRanges (vulkan/Buffer.cpp:140-155; a no-op for
 *                             HOST_COHERENT memory) before it submits the copies.                       */
size_t ocean_staging_map_offset(size_t vertices_bytes, size_t indices_bytes);
int ocean_read_maps_staging(ocean_t* ctx, uint32_t tile, void* mapped_base, size_t vertices_bytes,
                            size_t indices_bytes, size_t* bytes_to_flush);

/* Device pointers of the maps of tile 0 (tile i at +i*N*N*4 floats): zero-copy
 * hand-off to a device-side consumer (interop, RCCL gather).  A consumer on
 * ocean_stream() is ordered by the stream; any other one orders itself with
 * ocean_synchronize / ocean_wait_frame / ocean_compute_waves, which -- from this
 * call on, until the maps are re-allocated -- synchronise the stream.            */
int ocean_device_maps(ocean_t* ctx, void** d_disp, void** d_nrm);
/* Whether somebody the context's streams do not order may read the internal maps (see ocean_wait_frame): set by ocean_export_maps and
 * ocean_device_maps, cleared when the maps are re-allocated -- and by this call.  on = 0 is the caller's statement that every consumer of
 * the handed-out pointers runs on ocean_stream() (or orders itself with events): ocean_wait_frame / ocean_compute_waves go back to the
 * completion-record poll.  on != 0 forces the stream synchronisation for a context that shares its maps by means the library cannot see. */
int ocean_set_external_readers(ocean_t* ctx, int on);

/* Export of the maps as a dma-buf (the remainder: the reference uploads both maps every frame through a
 * host-visible staging buffer -- CopyModelTessDataToStagingBuffer + UpdateFrameMaps, WaterSurfaceMesh.cpp:642-755,
 * vulkan/Texture2D.cpp:175-226: a device-to-host-to-device round trip of N*N*32 bytes, 40 x the synthesis at 2048^2).  One
 * map set is one device allocation [displacement maps of all tiles | normal maps of all tiles]; this call hands out a
 * dma-buf file descriptor of it (hipMemGetHandleForAddressRange, hipMemRangeHandleTypeDmaBufFd) that a Vulkan renderer
 * imports with VK_EXT_external_memory_dma_buf (VkImportMemoryFdInfoKHR, handle type DMA_BUF_BIT_EXT) and binds to a
 * VkBuffer -- its vkCmdCopyBufferToImage then reads the maps where they were written, bufferOffset = *disp_offset /
 * *nrm_offset (+ tile * N*N*16) in the place of the staging offsets -- or another process / API imports like any dma-buf
 * (HIP: hipImportExternalMemory with hipExternalMemoryHandleTypeOpaqueFd; tests/cpp/import_demo.cpp).  INTEGRATION.md
 * section B has the Vulkan side.
 *   *dmabuf_fd   a new descriptor per call; the caller closes it.  The memory stays owned by the context and valid until
 *                ocean_destroy / ocean_set_tile_size.
 *   *bytes       size of the exported range (the allocation, a whole number of 2 MiB pages >= 2 * tiles * N*N*16).
 *   *map_set     which of the context's map sets this is: the one of the most recently enqueued frame (set 0 before any
 *                frame).  At pipeline depth 1 there is only set 0 and one export serves every frame; at depth D frames
 *                rotate over D sets -- export after each of the first D frames, or keep depth 1 for an importing renderer.
 * Ordering for the importer: ocean_synchronize, ocean_wait_frame or the synchronous ocean_compute_waves on the host (once a map
 * set has been exported the latter two synchronise the frame's stream instead of polling its completion records, so that the
 * maps are written back and visible when they return), or a semaphore of the importer's own API.  Caller-bound output
 * (ocean_bind_output) is not exported: OCEAN_E_UNSUPPORTED.                                                              */
int ocean_export_maps(ocean_t* ctx, int* dmabuf_fd, size_t* disp_offset, size_t* nrm_offset, size_t* bytes, int* map_set);

/* Make the context write its maps into caller-owned device memory
 * (tiles*N*N*4 floats each, 16-byte aligned), e.g. tensors owned by the
 * harness so a collective can send them without a copy.  NULL restores the
 * internal buffers.  While output is bound, ocean_wait_frame and
 * ocean_compute_waves synchronise the stream (the owner of the memory may read
 * it on any stream or API when they return).                                     */
int ocean_bind_output(ocean_t* ctx, void* d_disp, void* d_nrm);
/* The other direction of ocean_export_maps: the RENDERER owns the memory.  It exports a VkDeviceMemory (or any device allocation) as a
 * dma-buf / opaque fd (vkGetMemoryFdKHR), and the context imports it (hipImportExternalMemory) and writes its maps straight into it at
 * the given byte offsets -- tiles*N*N*16 bytes each, 16-byte aligned, not overlapping, inside `bytes`.  Everything said for
 * ocean_bind_output holds (depth 1, no ocean_export_maps of such a context); the descriptor stays the caller's (the import holds its own
 * reference); ocean_bind_output(ctx, NULL, NULL), a resize or ocean_destroy end the binding.                                            */
int ocean_bind_output_dmabuf(ocean_t* ctx, int dmabuf_fd, size_t bytes, size_t disp_offset, size_t nrm_offset);

/* Output mode.  OCEAN_MODE_FULL7 (default) is the reference: all seven fields.  The
 * reduced modes of BASELINE.json / SURVEY.md 8d compute fewer transforms and leave the
 * rest of the texel at the value the reference's maps are initialised to in spirit:
 *   OCEAN_MODE_CHOPPY5  h, Dx, Dz, slope-x, slope-z ("5 iFFTs"); normal.zw = 0
 *   OCEAN_MODE_HEIGHT1  height only; displacement = (0, h/A, 0, 1), normal = 0
 * Takes effect at the next frame (no ocean_prepare needed).                              */
/*   OCEAN_MODE_JACOBIAN the seven fields plus the reference's COMPUTE_JACOBIAN intent (WSTessendorf.h:209-224,
 *                       .cpp:330-335, 368-378, 421-428; dead, non-compiling code there): the cross derivative
 *                       d(Dx)/dz = d(Dz)/dx as an eighth real field, and
 *                         displacement.w = (1 + l dDx/dx)(1 + l dDz/dz) - (l dDx/dz)(l dDz/dx),   l = lambda,
 *                       the Jacobian of the horizontal displacement, instead of the constant 1; the reference's
 *                       shaders already carry it (WaterSurfaceMesh.vert:29) and paint foam where it is negative
 *                       (WaterSurfaceMesh.frag:210-212).  85 instead of 73 bytes per texel.                  */
enum { OCEAN_MODE_FULL7 = 0, OCEAN_MODE_CHOPPY5 = 1, OCEAN_MODE_HEIGHT1 = 2, OCEAN_MODE_JACOBIAN = 3 };
int ocean_set_mode(ocean_t* ctx, int mode);

/* Dispersion relation of the next ocean_prepare (all tiles).  OCEAN_DISPERSION_DEEP (default) is the
 * one the reference uses, sqrt(g k) (WSTessendorf.h:290-293 via QDispersion :284-287).  The other two are
 * the relations the reference defines but never calls: finite depth
 * sqrt(g k tanh(k D)) with param = D in metres (DispersionTransWaves, .h:301-304) and small waves
 * sqrt(g k (1 + k^2 L^2)) with param = L (DispersionSmallWaves, .h:312-315).  The quantisation of
 * .h:284-287 (floor(w / w0) * w0) is applied to all three so the animation stays periodic.   */
enum { OCEAN_DISPERSION_DEEP = 0, OCEAN_DISPERSION_FINITE_DEPTH = 1, OCEAN_DISPERSION_CAPILLARY = 2 };
int ocean_set_dispersion(ocean_t* ctx, int kind, float param);

/* Spectrum storage precision: 32 (default) or 16.  With 16 the per-frame passes read a
 * half2 copy of h0(k), scaled per tile by a power of two, instead of the fp32 one
 * (8 instead of 12 bytes per texel of input; omega stays fp32).  Takes effect at the
 * next ocean_prepare.  Accuracy: tests/test_parity_gpu.py states the measured bound.   */
int ocean_set_spectrum_precision(ocean_t* ctx, int bits);

/* Precision of the intermediates between the two passes: 32 (default) or 16.  With 16 the z-axis pass stores its
 * outputs as half2, scaled per tile by a power of two chosen at ocean_prepare from a time-independent bound of the
 * spectrum's column sums (nothing can overflow), and the x-axis pass reads them back: 7 instead of 14 bytes per
 * texel each way (59 instead of 73 per frame).  This is BASELINE.json config 4's reduced-precision mode: the
 * maps then differ from the fp32 path by up to ~1e-3 of a channel's maximum (tests state the measured bound);
 * the default fp32 path keeps its 1e-5.  Takes effect at the next ocean_prepare.                                  */
int ocean_set_intermediate_precision(ocean_t* ctx, int bits);

/* Frame pipelining.  With depth D consecutive asynchronous frames rotate over D
 * independent chains (own stream, own intermediates, own internal map set): the
 * first pass of one frame fills the memory-idle phases of the other frames' map
 * passes.  ocean_compute_waves, ocean_wait_frame and ocean_get_heights wait for the
 * most recently enqueued frame only; ocean_synchronize and the map read-out calls drain
 * every chain; the read-out functions and ocean_device_maps then refer to the frame
 * enqueued last.  Caller-bound
 * output buffers (ocean_bind_output) or a caller stream force depth 1.  depth 1 (the
 * default) = everything on one stream.  (The reference is strictly serial; its own
 * DOUBLE_BUFFERED switch, WaterSurfaceMesh.h:34, is the same idea on the upload side.)  */
int ocean_set_pipeline_depth(ocean_t* ctx, int depth /* 1 .. 8 */);

/* The hipStream_t the most recent frame was enqueued on (as void*), and a way
 * to make the context use ONE caller-owned stream instead (this also disables
 * pipelining; NULL = back to the context's own streams).                         */
void* ocean_stream(ocean_t* ctx);
int   ocean_set_stream(ocean_t* ctx, void* hip_stream);

/* ---- multi-GPU: one gather of the packed maps over RCCL ---------------------------------
 * Not in the reference (a single-process desktop application); this is the exchange step of the
 * tile-sharded batch mode (BASELINE.json north_star).  Tiles are independent, so the
 * per-frame synthesis needs no collective: one process per GPU owns a contiguous block of tiles
 * (one ocean context), and the only communication is every rank sending its finished maps to a
 * root rank -- ncclGather x 2 (displacement, normal) in one RCCL group, zero-copy from the
 * context's map buffers, 7 concurrent point-to-point xGMI streams into the root on an 8-GPU node.
 *   ocean_comm_unique_id  rank 0 creates the 128-byte id; the host program distributes it to the
 *                         other ranks by any means (MPI, a file, torch.distributed).
 *   ocean_comm_init       collective: every rank calls it with the same id (ncclCommInitRank on
 *                         the context's device).
 *   ocean_gather_maps     enqueues the gather of the most recently enqueued frame's maps on the
 *                         context's communication stream, ordered behind that frame; returns at
 *                         once.  On `root` the maps of rank r, tile i arrive at
 *                         d_recv_*[(r*tiles + i)*N*N*4 floats] (rank-major = global tile order
 *                         for equal shards); other ranks pass NULL.  With pipeline depth >= 2 the
 *                         following frames are synthesised into other map sets meanwhile and a
 *                         chain waits for its gather before rewriting its maps; ocean_synchronize
 *                         also waits for every gather in flight.
 * librccl is loaded at the first of these calls, not at library load.                        */
#define OCEAN_COMM_ID_BYTES 128
int ocean_comm_unique_id(void* id_out /* OCEAN_COMM_ID_BYTES */);
int ocean_comm_init(ocean_t* ctx, int nranks, int rank, const void* id /* OCEAN_COMM_ID_BYTES */);
int ocean_comm_destroy(ocean_t* ctx);
/* Size of the context's communicator and this context's rank in it, as RCCL reports them (ncclCommCount,
 * ncclCommUserRank): what a report of a multi-GPU run should quote instead of the launcher's word.                     */
int ocean_comm_count(const ocean_t* ctx, int* ranks, int* rank);
int ocean_gather_maps(ocean_t* ctx, int root, void* d_recv_disp, void* d_recv_nrm);
/* The same gather with the maps converted to IEEE half on the sending GPU first (round to nearest; 16 instead of 32
 * bytes per texel over xGMI, SURVEY.md 8e's option): the root receives tiles*N*N*4 halves per map and rank.  For
 * consumers that take half textures; the fp32 maps of the context are untouched.                                   */
int ocean_gather_maps_f16(ocean_t* ctx, int root, void* d_recv_disp, void* d_recv_nrm);
/* ncclResult_t of the most recent failing RCCL call on this thread (0 if none).                 */
int ocean_last_rccl_error(void);

#ifdef __cplusplus
}
#endif
#endif /* OCEAN_H_ */
