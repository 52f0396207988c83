/*
 * rrdxr.h -- C ABI of the MI355X-native refraction ray tracer.
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * (bottledspace/refraction-raytracing-dxr) has no FFI of its own: the path sits
 * behind D3D12 command-list calls issued by RefractionDemo.cpp.  Every entry
 * point below cites the reference call it replaces (file:line relative to the
 * reference tree).  Plain C types only; every function returns an rr_status
 * (never aborts, never throws); one caller thread per context, exactly like the
 * reference's single-threaded frame loop (RefractionDemo.cpp:557-612).
 *
 * The library needs a gfx950 device for every rr_* call that takes a context;
 * there is no CPU fallback.  The rr_host_* helpers (OBJ loader, camera math,
 * image decode) are pure host code.
 */
#ifndef RRDXR_H
#define RRDXR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RRDXR_ABI_VERSION 3

typedef enum rr_status {
    RR_OK = 0,
    RR_ERR_INVALID_ARGUMENT = 1,
    RR_ERR_NO_DEVICE = 2,        /* no gfx950 device / HIP runtime failure at create */
    RR_ERR_DEVICE = 3,           /* a HIP call failed; rr_last_error has the text */
    RR_ERR_OUT_OF_MEMORY = 4,
    RR_ERR_STATE = 5,            /* call order violated (e.g. dispatch before build) */
    RR_ERR_IO = 6,
    RR_ERR_UNSUPPORTED = 7,      /* parameter outside what the kernels are built for */
    RR_ERR_TRAVERSAL_OVERFLOW = 8/* sticky device error flag of a dispatch (reserved: the kernels size the traversal
                                  * stack from the built tree and deeper trees are refused at build time) */
} rr_status;

/* Vertex record: Mesh.hpp:6-11 == RayTracing.hlsl:5-9.  32 bytes, stride of t2. */
typedef struct rr_vertex {
    float position[3];
    float norm[3];
    float uv[2];
} rr_vertex;

/* SceneConstants: RayTracing.hlsl:1-4, CPU twin RefractionDemo.cpp:12-15.
 * proj_inv holds the 64 bytes exactly as the CPU writes them (DirectXMath row-major);
 * the shader reads them column-major, which the kernels reproduce. */
typedef struct rr_scene_constants {
    float proj_inv[16];
    float camera_loc[4];
} rr_scene_constants;

/* Mirrors the 64-byte D3D12_RAYTRACING_INSTANCE_DESC filled at RefractionDemo.cpp:324-334. */
typedef struct rr_instance_desc {
    float    transform[12];       /* 3x4 row-major object->world */
    uint32_t instance_id_mask;    /* InstanceID:24 | InstanceMask:8  */
    uint32_t hitgroup_flags;      /* InstanceContributionToHitGroupIndex:24 | Flags:8 */
    uint64_t blas;                /* mesh id from rr_upload_mesh (stands for the BLAS GPU VA) */
} rr_instance_desc;

#define RR_INSTANCE_FLAG_TRIANGLE_CULL_DISABLE            0x1u
#define RR_INSTANCE_FLAG_TRIANGLE_FRONT_COUNTERCLOCKWISE  0x2u

/* The literals of the shader become parameters whose defaults are those literals. */
typedef struct rr_dispatch_params {
    int32_t max_refract;          /* 5      RayTracing.hlsl:82  (payload.count < 5) */
    int32_t max_reflect;          /* 2      RayTracing.hlsl:110 (payload.count < 2); <= 8 */
    float   ior;                  /* 1.3    RayTracing.hlsl:95 */
    float   tmin_primary;         /* 1e-4   RayTracing.hlsl:52 */
    float   tmax_primary;         /* 100    RayTracing.hlsl:53 */
    float   tmin_secondary;       /* 1e-3   RayTracing.hlsl:99,114 */
    float   tmax_secondary;       /* 1000   RayTracing.hlsl:100,115 */
    uint32_t flags;               /* RR_DISPATCH_* */
} rr_dispatch_params;

#define RR_DISPATCH_FLOAT_OUTPUT  0x1u  /* also keep the un-quantised float4 colour per pixel */
#define RR_DISPATCH_COLLECT_STATS 0x2u  /* instrumented kernel: node/triangle/hit/miss counters */
#define RR_DISPATCH_TIME_KERNEL   0x4u  /* bracket the render kernel with a HIP event pair (rr_kernel_time) */
#define RR_DISPATCH_KEEP_COUNTERS 0x8u  /* do not zero the rr_get_stats counters first: keep accumulating */
#define RR_DISPATCH_TILES_RGB8    0x10u /* rr_render_orbit_sharded only: tiles are written as 3 bytes per pixel (alpha is
                                         * always 255): a quarter less to gather; rr_assemble_frames_rgb8 restores RGBA8 */
#define RR_DISPATCH_TONEMAP_REINHARD 0x20u /* SURVEY 8f.2, additive: the R8G8B8A8_UNORM store takes c / (1 + c) per channel (NaN and
                                              negative -> 0, +inf -> 1) instead of the reference's saturating c; the float frame
                                              (RR_DISPATCH_FLOAT_OUTPUT) stays linear.  Off by default = RayTracing.hlsl:62 */
#define RR_DISPATCH_DEBUG_NO_CULL 0x40u /* verification switch: treat the whole frame as the scene's screen rectangle, i.e. trace the
                                           primary ray of EVERY pixel (RayTracing.hlsl:60) instead of shading the blocks the host
                                           projection of the scene bounds rules out as one Miss.  Frames are identical by
                                           contract (tests/test_gpu_parity.py::test_background_culling_...); only slower */

typedef struct rr_stats {
    uint64_t rays;                /* every TraceRay: primary + secondary */
    uint64_t primary;
    uint64_t secondary;
    uint64_t hits;
    uint64_t misses;
    uint64_t terminal_hits;       /* ClosestHit with count >= max_refract */
    uint64_t tir;                 /* RefractRay returned false */
    uint64_t node_visits;         /* internal BVH nodes fetched (32 B each)  */
    uint64_t tri_tests;           /* triangle records fetched (48 B each)    */
    uint64_t pixels;              /* pixels this context rendered (last dispatch, or all frames of rr_render_orbit) */
    uint32_t stats_valid;         /* 1 if the last dispatch ran with RR_DISPATCH_COLLECT_STATS */
    uint32_t traversal_overflow;  /* sticky device error flag of the last dispatch (0 unless a kernel raised it) */
    uint32_t bvh_depth;           /* deepest BLAS / TLAS leaf */
    uint32_t render_kernel;       /* which kernel rendered the last dispatch: 0 k_render_fused (one lane per pixel, nodes through
                                     the L1), 1 k_render_lds (the same with persistent workgroups and the nodes in LDS),
                                     2 k_render_paths (four lanes per pixel), 7 k_stream_* (two-level scenes: one kernel per ray
                                     generation, rays in HBM queues); all of them produce the same bits */
    /* wave-level loop trips of the RR_DISPATCH_COLLECT_STATS kernels: a 64-lane wave issues one internal-node step, one
     * triangle test or one shading pass per trip however many of its lanes take part, so these are what vector-issue
     * time is made of; node_visits / (64 * node_trips) is the lane utilisation of the internal-node phase, and so on */
    uint64_t node_trips;
    uint64_t leaf_trips;
    uint64_t shade_passes;
    uint64_t waves;               /* waves that rendered (one 8x8 pixel block each in the block-per-wave kernels) */
    uint64_t background_waves;    /* of those, the waves of blocks outside the scene's screen rectangle: RayGen + one Miss
                                     (counted in shade_passes too), on a branch of their own in k_render_fused */
    /* ABI 3: the clock the RR_DISPATCH_COLLECT_STATS launches ran at = clock_ticks / clock_ref_ticks * 100 MHz (sums over the
     * waves of their lifetime on the shader clock, s_memtime, and on the constant 100 MHz reference, s_memrealtime) */
    uint64_t clock_ticks;
    uint64_t clock_ref_ticks;
    char     render_kernel_name[96]; /* the instantiation that rendered the last dispatch, as rocprofv3 --kernel-trace names it
                                        (less the argument list), e.g. "k_render_fused<19, 2, false, false, false, unsigned int, 0>" */
} rr_stats;

typedef struct rr_ray {
    float origin[3]; float tmin;
    float dir[3];    float tmax;
    uint32_t flags;               /* RR_RAY_FLAG_* */
    uint32_t pad[3];
} rr_ray;

#define RR_RAY_FLAG_CULL_BACK_FACING_TRIANGLES  0x10u   /* DXR RAY_FLAG values */
#define RR_RAY_FLAG_CULL_FRONT_FACING_TRIANGLES 0x20u

typedef struct rr_hit {
    float    t, u, v;             /* u weights vertex 1, v weights vertex 2 (RayTracing.hlsl:86) */
    uint32_t prim;                /* PrimitiveIndex() */
    uint32_t inst;                /* index into the rr_build_tlas array */
    uint32_t hit;                 /* 0 = miss */
} rr_hit;

typedef struct rr_context rr_context;

/* ---- device / lifetime: createDevice, RefractionDemo.cpp:142-172 ------------------------- */
int  rr_create(int device_ordinal, rr_context** out);
int  rr_destroy(rr_context* ctx);
const char* rr_last_error(const rr_context* ctx);
/* Use a caller-owned hipStream_t (e.g. torch's current stream) instead of the context's own, so that
 * the caller's work on that stream (RCCL collectives, copies) is ordered with the render kernels.
 * NULL means HIP's default stream.  rr_reset_stream returns to the context's own stream.
 * Stands where the command queue of :161-166 stood. */
int  rr_set_stream(rr_context* ctx, void* hip_stream);
int  rr_reset_stream(rr_context* ctx);
/* wait_until_finished, RefractionDemo.cpp:65-71 */
int  rr_wait(rr_context* ctx);

/* ---- assets -> device -------------------------------------------------------------------- */
/* Mesh::upload, Mesh.cpp:55-94 (+ geometry desc Mesh.cpp:39-53): copies; caller keeps ownership.
 * Positions must be finite and no larger than 1e18 in magnitude (rr_host_validate_positions): the BLAS builder works
 * with box areas in fp32 -- a D3D12 driver is free to produce garbage for such input, this library returns
 * RR_ERR_INVALID_ARGUMENT instead.  (Mesh::load itself accepts "v 1e39 0 0": strtof turns it into +inf.) */
int  rr_upload_mesh(rr_context* ctx, const rr_vertex* verts, uint32_t n_verts,
                    const uint32_t* indices, uint32_t n_indices, uint32_t* mesh_id);
/* load_texture, RefractionDemo.cpp:108-140: tightly packed RGB32F, row pitch w*12 (:128). */
int  rr_upload_envmap(rr_context* ctx, const float* rgb, int32_t w, int32_t h);

/* ---- acceleration structures: RefractionDemo.cpp:272-361 ----------------------------------- */
/* BuildRaytracingAccelerationStructure (bottom level), :277-322 */
int  rr_build_blas(rr_context* ctx, uint32_t mesh_id);       /* = PREFER_FAST_TRACE, the flag the reference passes (:286) */
/* D3D12_RAYTRACING_ACCELERATION_STRUCTURE_BUILD_FLAG_PREFER_FAST_TRACE / _FAST_BUILD (same bit values):
 * FAST_TRACE builds a clustered (PLOC) hierarchy for meshes up to 32768 triangles, the Morton LBVH above
 * that; FAST_BUILD always builds the LBVH. */
#define RR_BUILD_PREFER_FAST_TRACE 0x4u
#define RR_BUILD_PREFER_FAST_BUILD 0x8u
int  rr_build_blas_ex(rr_context* ctx, uint32_t mesh_id, uint32_t flags);
/* BuildRaytracingAccelerationStructure (top level), :324-356.  n == 0 is invalid; the
 * reference's scene is one identity instance with mask 1 and flags 0. */
int  rr_build_tlas(rr_context* ctx, const rr_instance_desc* instances, uint32_t n);

/* ---- per frame ----------------------------------------------------------------------------- */
/* copy_to_buffer(cameraConstantBuffer, ...), RefractionDemo.cpp:566 */
int  rr_set_camera(rr_context* ctx, const rr_scene_constants* constants);
/* Multi-GPU sharding (new; the reference is single-adapter): this context renders only the
 * 32x32-pixel tiles t with t % world == rank.  world == 1 (default) renders the whole frame. */
int  rr_set_tile_partition(rr_context* ctx, uint32_t rank, uint32_t world);
/* DispatchRays(desc{W,H,1}), RefractionDemo.cpp:580-594.  Asynchronous on the stream. */
int  rr_dispatch_rays(rr_context* ctx, uint32_t width, uint32_t height, const rr_dispatch_params* params);
/* DispatchRays(desc{W,H,Depth}): the reference always passes Depth = 1 (RefractionDemo.cpp:591); here
 * depth slice f is a whole frame rendered with constants[f] (the constant buffer becomes an array).
 * One launch renders all slices, so the long-running waves of one frame overlap the next frame's work.
 * Slice f is read back with rr_read_frame_slice(ctx, f, ...). */
int  rr_dispatch_rays_batch(rr_context* ctx, uint32_t width, uint32_t height, uint32_t depth,
                            const rr_scene_constants* constants, const rr_dispatch_params* params);
/* CopyResource(backbuffer <- rtTexture) + Present, RefractionDemo.cpp:596-609: blocks, copies the
 * R8G8B8A8_UNORM frame (and the float4 frame if RR_DISPATCH_FLOAT_OUTPUT was set) to host memory.
 * Either pointer may be NULL.  Only valid with world == 1. */
int  rr_read_frame(rr_context* ctx, uint8_t* rgba8, float* rgba32f);
int  rr_read_frame_slice(rr_context* ctx, uint32_t slice, uint8_t* rgba8, float* rgba32f);

/* Sharded output.  A context with world > 1 renders into a compact tile buffer:
 * rr_local_tile_count tiles of 32*32 RGBA8 pixels (4096 B each), tile-major. */
int  rr_local_tile_count(rr_context* ctx, uint32_t width, uint32_t height, uint32_t* n_tiles,
                         uint32_t* max_tiles_any_rank);
/* device -> device copy of this rank's tiles into caller memory (e.g. a torch tensor that RCCL
 * will gather); dst must hold max_tiles_any_rank*4096 bytes, the tail is zero-filled. */
int  rr_export_tiles(rr_context* ctx, void* d_dst);
/* rank 0: de-interleave the gathered [world][max_tiles][4096 B] buffer into a W*H RGBA8 frame.
 * d_frame may be NULL (an internal frame is used; fetch it with rr_read_frame). */
int  rr_assemble_tiles(rr_context* ctx, const void* d_gathered, uint32_t world, void* d_frame);

/* drawFrame loop in C (RefractionDemo.cpp:557-567 + WinMain.cpp:49-59): n_frames times
 * { camera constants for `angle`; rr_set_camera; rr_dispatch_rays; angle += angle_step }, issued
 * as ceil(n_frames / frames_per_dispatch) launches of frames_per_dispatch depth slices each
 * (1 = the reference's one DispatchRays per frame).  The last frames_per_dispatch frames stay
 * readable through rr_read_frame_slice.
 * Asynchronous; *angle is advanced like the reference's `static float angle`.  The counters
 * of rr_get_stats accumulate over the n_frames (they are zeroed once, before the first). */
int  rr_render_orbit(rr_context* ctx, uint32_t width, uint32_t height, const rr_dispatch_params* params,
                     float* angle, float angle_step, uint32_t n_frames, uint32_t frames_per_dispatch,
                     float fov_y, float aspect, float zn, float zf);

/* The same loop with every frame delivered to host memory: frame k lands at host_rgba8 + k*W*H*4.  Launches
 * alternate between two (or rr_set_frames_in_flight) device regions and each finished region is copied out on
 * its own stream while the next launch renders -- the CPU-GPU overlap the reference notes it lacks
 * (RefractionDemo.cpp:519-521).  Blocking: all n_frames are in host memory on return.  Page-lock the buffer
 * (rr_host_register) for full PCIe speed. */
int  rr_render_orbit_to_host(rr_context* ctx, uint32_t width, uint32_t height, const rr_dispatch_params* params,
                             float* angle, float angle_step, uint32_t n_frames, uint32_t frames_per_dispatch,
                             float fov_y, float aspect, float zn, float zf, uint8_t* host_rgba8);

/* How many launches of rr_render_orbit / rr_render_orbit_sharded may be in flight at once (1..4, default 2;
 * 1 = strictly one after the other, like the reference's fence wait per frame, RefractionDemo.cpp:611).  With 2,
 * consecutive launches go to two internal streams and write to two output regions, so the few long-running
 * waves that end one launch overlap the start of the next: monkey.obj 1080p, 137 -> 88 us per frame at
 * frames_per_dispatch = 4, 87 -> 83 at 16, nothing at 64 (ott.obj: 311 -> 172, 124 -> 107).  The call still returns
 * with everything ordered on the context's stream.  Dispatches with RR_DISPATCH_TIME_KERNEL always run one at a
 * time (their durations must be exclusive), and so do launches the persistent k_render_lds was chosen for (its
 * workgroups hold every CU until the launch ends).  rr_dispatch_rays[_batch] are not affected. */
int  rr_set_frames_in_flight(rr_context* ctx, uint32_t n);

/* The same loop for a sharded context (rr_set_tile_partition; world == 1 is allowed): frame f renders this rank's tiles straight into
 * caller device memory at d_tiles + f*frame_stride_bytes (max_tiles_any_rank*4096 B each, tail
 * zero-filled), i.e. into the send buffer of the RCCL gather, with no intermediate copy. */
int  rr_render_orbit_sharded(rr_context* ctx, uint32_t width, uint32_t height, const rr_dispatch_params* params,
                             float* angle, float angle_step, uint32_t n_frames, uint32_t frames_per_dispatch,
                             float fov_y, float aspect, float zn, float zf,
                             void* d_tiles, uint64_t frame_stride_bytes);
/* The same, submitted to internal stream `lane` (0..3) instead of the context's stream.  The lane starts
 * after everything submitted to the context's stream so far; the context's stream does NOT wait for the
 * lane until rr_lane_join (rr_wait / rr_get_stats join every lane).  Keeping two lanes in flight lets the
 * tail of one batch (a few long-running waves) overlap the head of the next, and lets the gather of batch b
 * run while batch b+1 renders.  Lanes own their constant buffers; counters are shared (atomic). */
int  rr_render_orbit_sharded_lane(rr_context* ctx, uint32_t width, uint32_t height, const rr_dispatch_params* params,
                                  float* angle, float angle_step, uint32_t n_frames, uint32_t frames_per_dispatch,
                                  float fov_y, float aspect, float zn, float zf,
                                  void* d_tiles, uint64_t frame_stride_bytes, uint32_t lane);
int  rr_lane_join(rr_context* ctx, uint32_t lane);
/* Multi-GPU tile partition that only moves what has to move (see rr_host_mesh_partition below) */
typedef struct rr_mesh_partition {
    uint32_t tiles_x, n_tiles;
    uint32_t rect_x0, rect_y0, rect_w, rect_h;      /* in tiles */
    uint32_t n_mesh_tiles, n_bg_tiles, max_mesh_tiles_per_rank, world;
    uint32_t rank0_rounds;      /* how the mesh tiles are dealt: rank 0 also renders every background tile, so it takes fewer of them --
                                   ranks 1 .. world-1 take one tile each for rank0_rounds rounds, then rank 0 takes one
                                   (cycle of rank0_rounds * (world - 1) + 1 tiles); 0: plain round robin over all ranks (world == 1,
                                   or no background); 0xffffffff: rank 0 takes none (the background alone is its share) */
    uint32_t pad;
} rr_mesh_partition;
/* The mesh-tile partition for a sharded context, one DispatchRays(W, H, n_frames) per call:
 * rr_mesh_partition_for_orbit says how the n_frames frames starting at `angle` will be dealt (the rectangle is the union over
 * their cameras; the same on every rank), so that the caller can size its buffers: frame f's mesh tiles of this rank go to
 * d_mesh_tiles + f * mesh_stride_bytes (slot s at s * 3072: RGB8, max_mesh_tiles_per_rank slots, unused ones zeroed) -- the send
 * buffer of the gather --, and on rank 0 its background tiles to d_bg_tiles + f * bg_stride_bytes (n_bg_tiles slots; other
 * ranks pass NULL).  rr_assemble_frames_mesh_rgb8 puts gathered mesh tiles and rank 0's background tiles back into rasters. */
int  rr_mesh_partition_for_orbit(rr_context* ctx, uint32_t width, uint32_t height, float angle, float angle_step, uint32_t n_frames,
                                 float fov_y, float aspect, float zn, float zf, rr_mesh_partition* out);
int  rr_render_orbit_mesh_sharded_lane(rr_context* ctx, uint32_t width, uint32_t height, const rr_dispatch_params* params,
                                       float* angle, float angle_step, uint32_t n_frames, float fov_y, float aspect, float zn, float zf,
                                       void* d_mesh_tiles, uint64_t mesh_stride_bytes, void* d_bg_tiles, uint64_t bg_stride_bytes,
                                       uint32_t lane);
int  rr_assemble_frames_mesh_rgb8(rr_context* ctx, const void* d_gathered, uint64_t rank_stride_bytes, uint64_t frame_stride_bytes,
                                  const void* d_bg_tiles, uint64_t bg_stride_bytes, const rr_mesh_partition* part, uint32_t n_frames,
                                  uint32_t width, uint32_t height, void* d_frames, uint64_t out_stride_bytes);
/* ---- the final image gather, natively (north star: "host stays C++ ... final RCCL gather over xGMI") --------------------
 * The reference has one adapter and no collective (RefractionDemo.cpp:163 NodeMask 0); here every rank renders its
 * tiles and ONE gather per batch of frames brings them to the root.  RCCL is looked up at run time (dlopen of
 * librccl.so; nothing links against it), one process per GPU:
 *   rr_comm_unique_id   rank 0 makes the 128-byte id and hands it to the other processes (file, pipe, MPI ...)
 *   rr_comm_init        ncclCommInitRank on the context's device; *comm is the communicator (opaque)
 *   rr_gather_frames    every rank contributes bytes_per_rank bytes at d_send; the root receives rank r's block at
 *                       d_recv + r * bytes_per_rank (d_recv may be NULL elsewhere).  Grouped ncclSend / ncclRecv, so the
 *                       root's seven incoming transfers use its seven xGMI links side by side; enqueued on the context's
 *                       stream (ordered after the renders before it, before the rr_assemble_frames after it).
 *   rr_comm_destroy
 * rr_device_alloc / rr_device_free / rr_device_read give a host without HIP headers the buffers these calls work on. */
int  rr_comm_unique_id(void* id128);
int  rr_comm_init(rr_context* ctx, const void* id128, int rank, int world, void** comm);
int  rr_comm_destroy(void* comm);
int  rr_gather_frames(rr_context* ctx, void* comm, int rank, int world, const void* d_send, void* d_recv,
                      uint64_t bytes_per_rank, int root);
int  rr_device_alloc(rr_context* ctx, uint64_t bytes, void** d_ptr);
int  rr_device_free(rr_context* ctx, void* d_ptr);
int  rr_device_read(rr_context* ctx, const void* d_src, void* host_dst, uint64_t bytes);   /* blocking */

/* rank 0, after gathering n_frames at once: frame f of rank r lies at
 * d_gathered + r*rank_stride_bytes + f*frame_stride_bytes; writes n_frames W*H RGBA8 rasters to
 * d_frames + f*out_stride_bytes.  One launch for all frames. */
int  rr_assemble_frames(rr_context* ctx, const void* d_gathered, uint32_t world, uint64_t rank_stride_bytes,
                        uint64_t frame_stride_bytes, uint32_t n_frames, uint32_t width, uint32_t height,
                        void* d_frames, uint64_t out_stride_bytes);
/* The same for tiles rendered with RR_DISPATCH_TILES_RGB8 (max_tiles_any_rank*3072 B per frame and rank); output RGBA8. */
int  rr_assemble_frames_rgb8(rr_context* ctx, const void* d_gathered, uint32_t world, uint64_t rank_stride_bytes,
                             uint64_t frame_stride_bytes, uint32_t n_frames, uint32_t width, uint32_t height,
                             void* d_frames, uint64_t out_stride_bytes);

/* HIP-event timing on the stream the kernels run on.  rr_timing_begin records an event,
 * rr_timing_end records another, waits for it and returns the elapsed milliseconds. */
int  rr_timing_begin(rr_context* ctx);
int  rr_timing_end(rr_context* ctx, float* elapsed_ms);
/* sum and count of the render-kernel durations of all dispatches issued with
 * RR_DISPATCH_TIME_KERNEL since the last call (blocks until they finished; at most 4096). */
int  rr_kernel_time(rr_context* ctx, float* sum_ms, uint32_t* n_launches);

/* exact counters of the last dispatch (blocks until it finished) */
int  rr_get_stats(rr_context* ctx, rr_stats* out);

/* Page-lock a caller-owned host buffer (e.g. the back buffer rr_read_frame copies into) so that read-backs run at
 * PCIe speed instead of through the runtime's staging copies; unregister before freeing it.  The reference's
 * readback heap plays this role (a CPU-visible committed resource, RefractionDemo.cpp:596-609 present path). */
int  rr_host_register(rr_context* ctx, void* p, size_t bytes);
int  rr_host_unregister(rr_context* ctx, void* p);

/* TraceRay on caller-supplied rays (host arrays), same traversal code as the render path.
 * Stands for RayTracing.hlsl:60,106,121 in isolation; used by the parity tests. */
int  rr_trace_rays(rr_context* ctx, const rr_ray* rays, uint32_t n, rr_hit* hits);

/* Miss on caller-supplied ray directions (host arrays of n x 3 floats in, n x 3 floats out): the equirectangular lookup
 * of RayTracing.hlsl:127-137 in isolation -- atan2 / acos, the division by the literal 3.14159, the float-to-uint texel
 * address and the zero returned outside the texture (reached at atan2 = pi and at r.y = -1); used by the parity tests. */
int  rr_env_lookup(rr_context* ctx, const float* dirs, uint32_t n, float* rgb);

/* Introspection for tests: copies the packed BLAS of a mesh to host.  nodes: n_nodes*64 B
 * (two child boxes + two child refs), tris: n_tris*48 B (v0,e1,e2 with prim id in v0.w). */
int  rr_download_blas(rr_context* ctx, uint32_t mesh_id, void* nodes, uint32_t* n_nodes,
                      void* tris, uint32_t* n_tris);

/* The same hierarchy as the traversal kernels read it: n_nodes*32 B, per node six words holding one slab plane
 * of both children as IEEE half-precision cell counts q on the BLAS's grid (child 0 in the low half, child 1 in the
 * high half; order lox,loy,loz,hix,hiy,hiz), then the two child refs (>= 0: byte offset of an internal node, < 0:
 * leaf ~index).  grid_org_cell: origin xyz (the centre of the bounds) then cell size xyz (plane = org + q*cell).
 * Every stored box contains its fp32 box. */
int  rr_download_qnodes(rr_context* ctx, uint32_t mesh_id, void* qnodes, uint32_t* n_nodes, float grid_org_cell[6]);

/* ---- pure host helpers (no device, no context) --------------------------------------------- */
void rr_default_dispatch_params(rr_dispatch_params* p);
/* Where the box {lo[3], hi[3]} can be seen at all with any of the n constants (GenerateCameraRay, RayTracing.hlsl:27-40):
 * rect = {x0, y0, x1, y1} in pixels, aligned outward to 8 with 8 pixels of margin; the whole frame whenever the projection
 * cannot be trusted (camera inside or beside the box, singular or badly conditioned proj_inv, constants == NULL).  The render
 * kernels shade blocks outside it as one Miss without TraceRay. */
int  rr_host_screen_rect(const float bounds[6], const rr_scene_constants* constants, uint32_t n, uint32_t width, uint32_t height,
                         uint32_t rect[4]);
/* Multi-GPU tile partition that only moves what has to move: the tiles that touch the rectangle ("mesh tiles", raster order
 * inside the rectangle, index i) are dealt to the ranks in turn (rank0_rounds: rank 0 takes fewer, see the struct), each rank's
 * tiles filling the slots of its tile buffer in order; all the other tiles
 * ("background tiles": one Miss per pixel, a tenth of the work and two thirds of the bytes of the reference's views) belong to
 * rank 0, in raster order, and never cross a link.  rect_w == 0: no usable rectangle, every tile is a mesh tile. */
int  rr_host_mesh_partition(const float bounds[6], const rr_scene_constants* constants, uint32_t n, uint32_t width, uint32_t height,
                            uint32_t world, rr_mesh_partition* out);
/* mesh tile `mesh_index` (raster order inside the rectangle) -> the rank it is dealt to and its slot in that rank's buffer;
 * the number of mesh tiles a rank holds */
int  rr_host_mesh_tile_home(const rr_mesh_partition* part, uint32_t mesh_index, uint32_t* rank, uint32_t* slot);
uint32_t rr_host_mesh_tiles_of_rank(const rr_mesh_partition* part, uint32_t rank);
/* RefractionDemo.cpp:559-566: camera constants for an orbit angle.  The reference's literals are
 * fov_y = float(52.0/180.0*3.1415), aspect = 1.333f, zn = 1, zf = 125; frame k uses angle 0.01*(k+1). */
int  rr_host_camera_orbit(float angle, float fov_y, float aspect, float zn, float zf,
                          rr_scene_constants* out);
/* Mesh::load, Mesh.cpp:6-37.  Arrays are malloc'ed, release with rr_host_free.  A file that cannot
 * be opened returns RR_ERR_IO (Mesh::load returns false). */
int  rr_host_mesh_load_obj(const char* filename, rr_vertex** verts, uint32_t* n_verts,
                           uint32_t** indices, uint32_t* n_indices);
/* Additive hardening behind the same output format: with RR_OBJ_HARDENED faces may be
 * polygons (fan-triangulated), corners may be "v", "v/vt", "v//vn" or "v/vt/vn", indices may be
 * negative; a missing uv is (0,0), a face without normals gets its flat winding normal.
 * flags = 0 is exactly rr_host_mesh_load_obj. */
#define RR_OBJ_HARDENED 0x1u
int  rr_host_mesh_load_obj_ex(const char* filename, uint32_t flags, rr_vertex** verts, uint32_t* n_verts,
                              uint32_t** indices, uint32_t* n_indices);
/* stbi_loadf(file,&x,&y,&n,req_comp) as called at RefractionDemo.cpp:111, for the two formats the demo's
 * environment map exists in: Radiance .hdr (flat and RLE scanlines) and .png (1..16-bit gray / RGB / palette,
 * with or without alpha, plain or Adam7-interlaced), LDR expanded with pow(v/255, 2.2) as stb does.
 * NARROWER than stb_image (stb_image.h:1491): JPEG, BMP, TGA, PSD, GIF, PIC and PNM files are refused --
 * NULL, like any other failure (the reference ignores load failures; the C++ mirror reports them). */
float* rr_host_image_loadf(const char* filename, int* x, int* y, int* channels_in_file, int req_comp);
/* Radiance RLE .hdr writer (the reference's envmap.hdr is missing from the mount). */
int  rr_host_image_write_hdr(const char* filename, int w, int h, const float* rgb);
void rr_host_free(void* p);
/* RR_OK if every position of the n vertices is finite and |x|, |y|, |z| <= 1e18 (what rr_upload_mesh requires),
 * RR_ERR_INVALID_ARGUMENT otherwise; *first_bad (may be NULL) receives the index of the first offending vertex. */
int  rr_host_validate_positions(const rr_vertex* verts, uint32_t n_verts, uint32_t* first_bad);
uint32_t rr_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RRDXR_H */
