// RefractionDemo.cpp -- the reference's frame driver (RefractionDemo.cpp:513-612) over the rrdxr
// C ABI.  Same call order as the reference's initialize(): device, env map, mesh load + upload,
// acceleration structures, then per frame: camera constants, DispatchRays, copy out, wait.
#include "RefractionDemo.hpp"

#include <cstring>

namespace RefractionDemo {
namespace {
rr_context* g_ctx = nullptr;
Mesh cubeMesh;                               // the reference's name for its one mesh (RefractionDemo.cpp:17)
Options g_opt;
float g_angle = 0.01f;                       // static float angle (RefractionDemo.cpp:555)
std::vector<uint8_t> g_back;
bool g_back_pinned = false;
std::string g_err;
void* g_comm = nullptr;                      // RCCL communicator of the sharded form
int g_rank = 0, g_world = 1;

int fail(int code, const char* what)
{
    g_err = what;
    if (g_ctx && code != RR_ERR_IO) { g_err += ": "; g_err += rr_last_error(g_ctx); }
    return code;
}
} // namespace

int initialize(const Options& opt)
{
    shutdown();
    g_opt = opt;
    g_angle = opt.angle0;
    int rc = rr_create(opt.device, &g_ctx);                               // createDevice :142-172
    if (rc != RR_OK) return fail(rc, "rr_create");

    int x = 0, y = 0, n = 0;                                              // load_texture :108-140
    float* env = rr_host_image_loadf(opt.env_path.c_str(), &x, &y, &n, 3);
    if (!env) return fail(RR_ERR_IO, "env map could not be loaded");      // the reference does not check (:111)
    rc = rr_upload_envmap(g_ctx, env, x, y);
    rr_host_free(env);
    if (rc != RR_OK) return fail(rc, "rr_upload_envmap");

    cubeMesh = Mesh();
    if (!cubeMesh.load(opt.mesh_path.c_str())) return fail(RR_ERR_IO, "mesh could not be loaded");   // :537
    if ((rc = cubeMesh.upload(g_ctx)) != RR_OK) return fail(rc, "Mesh::upload");                      // :538

    // setupRaytracingAccelerationStructures :272-361: one BLAS, one identity instance, mask 1, flags 0
    const RaytracingGeometry geo = cubeMesh.raytracingGeometry();
    if ((rc = rr_build_blas(g_ctx, geo.mesh_id)) != RR_OK) return fail(rc, "rr_build_blas");
    rr_instance_desc inst;
    std::memset(&inst, 0, sizeof inst);
    inst.transform[0] = inst.transform[5] = inst.transform[10] = 1.0f;    // :325-327
    inst.instance_id_mask = 1u << 24;                                     // InstanceMask = 1, InstanceID = 0
    inst.hitgroup_flags = 0;
    inst.blas = geo.mesh_id;
    if ((rc = rr_build_tlas(g_ctx, &inst, 1)) != RR_OK) return fail(rc, "rr_build_tlas");

    if (g_back_pinned) { rr_host_unregister(g_ctx, g_back.data()); g_back_pinned = false; }
    g_back.assign((size_t)opt.width * opt.height * 4, 0);
    g_back_pinned = rr_host_register(g_ctx, g_back.data(), g_back.size()) == RR_OK;     // best effort: pageable works too
    return RR_OK;
}

int drawFrame()
{
    if (!g_ctx) return fail(RR_ERR_STATE, "initialize first");
    rr_scene_constants sc;
    int rc = rr_host_camera_orbit(g_angle, g_opt.fov_y, g_opt.aspect, g_opt.zn, g_opt.zf, &sc);      // :559-565
    if (rc != RR_OK) return fail(rc, "rr_host_camera_orbit");
    if ((rc = rr_set_camera(g_ctx, &sc)) != RR_OK) return fail(rc, "rr_set_camera");                  // :566
    g_angle += g_opt.angle_step;                                                                      // :567
    if ((rc = rr_dispatch_rays(g_ctx, (uint32_t)g_opt.width, (uint32_t)g_opt.height, &g_opt.dispatch)) != RR_OK)
        return fail(rc, "rr_dispatch_rays");                                                          // :580-594
    if ((rc = rr_read_frame(g_ctx, g_back.data(), nullptr)) != RR_OK) return fail(rc, "rr_read_frame");   // :596-611
    return RR_OK;
}

// `for (;;) drawFrame();` (WinMain.cpp:49-59) without the per-frame fence wait and read-back: n_frames of the
// orbit, frames_per_dispatch depth slices per launch, in_flight launches overlapping.  The last frame lands in
// backBuffer().  The reference notes the missing overlap itself (RefractionDemo.cpp:519-521).
int pump(int n_frames, int frames_per_dispatch, int in_flight, rr_stats* stats)
{
    if (!g_ctx) return fail(RR_ERR_STATE, "initialize first");
    if (n_frames <= 0 || frames_per_dispatch <= 0) return fail(RR_ERR_INVALID_ARGUMENT, "pump: frame counts must be positive");
    int rc = rr_set_frames_in_flight(g_ctx, (uint32_t)in_flight);
    if (rc != RR_OK) return fail(rc, "rr_set_frames_in_flight");
    rc = rr_render_orbit(g_ctx, (uint32_t)g_opt.width, (uint32_t)g_opt.height, &g_opt.dispatch, &g_angle, g_opt.angle_step,
                         (uint32_t)n_frames, (uint32_t)frames_per_dispatch, g_opt.fov_y, g_opt.aspect, g_opt.zn, g_opt.zf);
    if (rc != RR_OK) return fail(rc, "rr_render_orbit");
    const uint32_t last = (uint32_t)((n_frames - 1) % frames_per_dispatch);
    if ((rc = rr_read_frame_slice(g_ctx, last, g_back.data(), nullptr)) != RR_OK) return fail(rc, "rr_read_frame_slice");
    if (stats && (rc = rr_get_stats(g_ctx, stats)) != RR_OK) return fail(rc, "rr_get_stats");
    return RR_OK;
}

// The frame loop with every frame delivered to host memory (frames[k*w*h*4 ...]), read-back overlapped with
// rendering (rr_render_orbit_to_host).  `frames` must hold n_frames frames; page-lock it for full PCIe speed.
int stream(int n_frames, int frames_per_dispatch, int in_flight, uint8_t* frames)
{
    if (!g_ctx) return fail(RR_ERR_STATE, "initialize first");
    if (n_frames <= 0 || frames_per_dispatch <= 0 || !frames) return fail(RR_ERR_INVALID_ARGUMENT, "stream: bad arguments");
    int rc = rr_set_frames_in_flight(g_ctx, (uint32_t)in_flight);
    if (rc != RR_OK) return fail(rc, "rr_set_frames_in_flight");
    rc = rr_render_orbit_to_host(g_ctx, (uint32_t)g_opt.width, (uint32_t)g_opt.height, &g_opt.dispatch, &g_angle, g_opt.angle_step,
                                 (uint32_t)n_frames, (uint32_t)frames_per_dispatch, g_opt.fov_y, g_opt.aspect, g_opt.zn, g_opt.zf, frames);
    if (rc != RR_OK) return fail(rc, "rr_render_orbit_to_host");
    std::memcpy(g_back.data(), frames + (size_t)(n_frames - 1) * g_back.size(), g_back.size());
    return RR_OK;
}

int initializeSharded(const Options& opt, int rank, int world, const void* id128)
{
    int rc = initialize(opt);
    if (rc != RR_OK) return rc;
    if ((rc = rr_set_tile_partition(g_ctx, (uint32_t)rank, (uint32_t)world)) != RR_OK) return fail(rc, "rr_set_tile_partition");
    if ((rc = rr_comm_init(g_ctx, id128, rank, world, &g_comm)) != RR_OK) return fail(rc, "rr_comm_init");
    g_rank = rank; g_world = world;
    return RR_OK;
}

int pumpSharded(int n_frames, int frames_per_gather, rr_stats* stats)
{
    if (!g_ctx || !g_comm) return fail(RR_ERR_STATE, "initializeSharded first");
    if (n_frames <= 0 || frames_per_gather <= 0) return fail(RR_ERR_INVALID_ARGUMENT, "pumpSharded: frame counts must be positive");
    const uint32_t W = (uint32_t)g_opt.width, H = (uint32_t)g_opt.height;
    // The mesh-tile partition (rr_mesh_partition): only the tiles that touch the scene's screen rectangle are dealt to the ranks
    // and gathered; rank 0 renders the background tiles itself.  Buffers are sized for a batch whose rectangle is the whole
    // frame; a batch moves max_mesh_tiles_per_rank tiles per frame and rank.
    const uint32_t n_tiles = ((W + 31u) / 32u) * ((H + 31u) / 32u);
    const uint64_t tile_bytes = 32 * 32 * 3;                                      // RGB8
    const uint64_t rank_stride = (uint64_t)((n_tiles + (uint32_t)g_world - 1u) / (uint32_t)g_world) * tile_bytes * (uint64_t)frames_per_gather;
    const uint64_t bg_bytes = (uint64_t)n_tiles * tile_bytes * (uint64_t)frames_per_gather;
    const uint64_t raster = (uint64_t)W * H * 4;
    // Two buffer sets and two render lanes: batch b is rendered on lane b % 2 into set b % 2 while the gather of batch b - 1
    // (queued on the context's stream behind that lane's join) and its de-interleave run -- the gather travels under the next
    // render instead of after it.  Set b % 2 is free again when batch b + 2 starts: the context's stream, on which gather and
    // de-interleave of batch b were queued, is what lane b % 2 forks from.
    void *d_send[2] = { nullptr, nullptr }, *d_recv[2] = { nullptr, nullptr }, *d_bg[2] = { nullptr, nullptr }, *d_frames = nullptr;
    auto release = [&] { for (int k = 0; k < 2; ++k) { rr_device_free(g_ctx, d_send[k]); rr_device_free(g_ctx, d_recv[k]); rr_device_free(g_ctx, d_bg[k]); } rr_device_free(g_ctx, d_frames); };
    int rc = RR_OK;
    for (int k = 0; k < 2 && rc == RR_OK; ++k) {
        rc = rr_device_alloc(g_ctx, rank_stride, &d_send[k]);
        if (rc == RR_OK && g_rank == 0) rc = rr_device_alloc(g_ctx, rank_stride * (uint64_t)g_world, &d_recv[k]);
        if (rc == RR_OK && g_rank == 0) rc = rr_device_alloc(g_ctx, bg_bytes, &d_bg[k]);
    }
    if (rc == RR_OK && g_rank == 0) rc = rr_device_alloc(g_ctx, raster * (uint64_t)frames_per_gather, &d_frames);
    if (rc != RR_OK) { release(); return fail(rc, "rr_device_alloc"); }
    rr_dispatch_params p = g_opt.dispatch;
    int last_n = 0, pending_n = 0, pending_set = -1;
    rr_mesh_partition pending_part;
    std::memset(&pending_part, 0, sizeof pending_part);
    auto gather_pending = [&]() -> int {            // join the lane of the batch launched before, gather it, de-interleave on rank 0
        if (pending_set < 0) return RR_OK;
        const uint64_t frame_stride = (uint64_t)pending_part.max_mesh_tiles_per_rank * tile_bytes;
        int r = rr_lane_join(g_ctx, (uint32_t)pending_set);
        if (r != RR_OK) return fail(r, "rr_lane_join");
        r = rr_gather_frames(g_ctx, g_comm, g_rank, g_world, d_send[pending_set], d_recv[pending_set], frame_stride * (uint64_t)pending_n, 0);
        if (r != RR_OK) return fail(r, "rr_gather_frames");
        if (g_rank == 0) {
            r = rr_assemble_frames_mesh_rgb8(g_ctx, d_recv[pending_set], frame_stride * (uint64_t)pending_n, frame_stride, d_bg[pending_set],
                                             (uint64_t)pending_part.n_bg_tiles * tile_bytes, &pending_part, (uint32_t)pending_n, W, H, d_frames, raster);
            if (r != RR_OK) return fail(r, "rr_assemble_frames_mesh_rgb8");
        }
        last_n = pending_n; pending_set = -1;
        return RR_OK;
    };
    int b = 0;
    for (int k = 0; k < n_frames && rc == RR_OK; k += frames_per_gather, ++b) {
        const int n = n_frames - k < frames_per_gather ? n_frames - k : frames_per_gather;
        if (k > 0) p.flags |= RR_DISPATCH_KEEP_COUNTERS;
        rr_mesh_partition part;
        rc = rr_mesh_partition_for_orbit(g_ctx, W, H, g_angle, g_opt.angle_step, (uint32_t)n, g_opt.fov_y, g_opt.aspect, g_opt.zn, g_opt.zf, &part);
        if (rc != RR_OK) { fail(rc, "rr_mesh_partition_for_orbit"); break; }
        rc = rr_render_orbit_mesh_sharded_lane(g_ctx, W, H, &p, &g_angle, g_opt.angle_step, (uint32_t)n, g_opt.fov_y, g_opt.aspect, g_opt.zn, g_opt.zf,
                                               d_send[b & 1], (uint64_t)part.max_mesh_tiles_per_rank * tile_bytes, d_bg[b & 1],
                                               (uint64_t)part.n_bg_tiles * tile_bytes, (uint32_t)(b & 1));
        if (rc != RR_OK) { fail(rc, "rr_render_orbit_mesh_sharded_lane"); break; }
        if ((rc = gather_pending()) != RR_OK) break;
        pending_set = b & 1; pending_n = n; pending_part = part;
    }
    if (rc == RR_OK) rc = gather_pending();
    if (rc == RR_OK && g_rank == 0 && last_n > 0) {
        rc = rr_device_read(g_ctx, (const uint8_t*)d_frames + (uint64_t)(last_n - 1) * raster, g_back.data(), raster);
        if (rc != RR_OK) fail(rc, "rr_device_read");
    }
    if (rc == RR_OK) { rc = rr_wait(g_ctx); if (rc != RR_OK) fail(rc, "rr_wait"); }
    if (rc == RR_OK && stats && (rc = rr_get_stats(g_ctx, stats)) != RR_OK) fail(rc, "rr_get_stats");
    if (rc != RR_OK) (void)rr_wait(g_ctx);
    release();
    return rc;
}

const std::vector<uint8_t>& backBuffer() { return g_back; }
rr_context* context() { return g_ctx; }
float currentAngle() { return g_angle; }
const char* lastError() { return g_err.c_str(); }

void shutdown()
{
    if (g_comm) { rr_comm_destroy(g_comm); g_comm = nullptr; }
    if (g_ctx) {
        if (g_back_pinned) { rr_host_unregister(g_ctx, g_back.data()); g_back_pinned = false; }
        rr_destroy(g_ctx); g_ctx = nullptr;
    }
}

} // namespace RefractionDemo
