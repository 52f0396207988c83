// rr_capi.cpp -- the C ABI of include/rrdxr.h over HIP streams.
//
// Stands where RefractionDemo.cpp's D3D12 plumbing stood: createDevice (:142-172), Mesh::upload
// (Mesh.cpp:55-94), load_texture (:108-140), the two BuildRaytracingAccelerationStructure calls
// (:277-356), the per-frame constant-buffer copy (:566), DispatchRays (:580-594), the UAV ->
// backbuffer copy (:596-604) and the fence wait (:65-71).  Everything device-side is a kernel in
// rr_bvh_build.hip / rr_render.hip; this file only owns memory, call order and error reporting.
#include "../../include/rrdxr.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rr_launch.h"

using namespace rr;

static_assert(sizeof(rr_vertex) == 32, "Vertex stride (Mesh.cpp:45)");
static_assert(sizeof(rr_instance_desc) == 64, "D3D12_RAYTRACING_INSTANCE_DESC");
static_assert(sizeof(rr_scene_constants) == 80, "SceneConstants");
static_assert(sizeof(rr_ray) == sizeof(rr_ray_dev) && sizeof(rr_hit) == sizeof(rr_hit_dev), "ray/hit ABI");

// Optional roctx ranges around the coarse steps (build, dispatch, assemble) so that `rocprofv3 --marker-trace`
// shows them next to the kernels.  The marker library is looked up at run time; without it the calls are no-ops.
#include <dlfcn.h>
// RCCL is looked up with dlopen (nothing links against it, and building needs no RCCL header): the two things of its ABI that
// cross this file are declared here -- the 128-byte unique id, passed by value to ncclCommInitRank, and ncclUint8 of ncclDataType_t
struct rr_nccl_unique_id { char internal[128]; };
enum { RR_NCCL_UINT8 = 1 };
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        for (const char* lib : { "librocprofiler-sdk-roctx.so", "libroctx64.so" }) {
            if (void* h = dlopen(lib, RTLD_LAZY | RTLD_LOCAL)) {
                push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
const Roctx& roctx() { static const Roctx r; return r; }
struct Range {
    explicit Range(const char* name) { if (roctx().push) roctx().push(name); }
    ~Range() { if (roctx().pop) roctx().pop(); }
};
} // namespace

namespace {

struct MeshRes {
    float*    d_verts = nullptr;     // n_verts * 8 floats
    uint32_t* d_idx = nullptr;
    uint32_t  n_verts = 0, n_idx = 0, n_tris = 0;
    BvhNode*  nodes = nullptr;       // fp32 hierarchy (builder output, rr_download_blas)
    QNode*    qnodes = nullptr;      // what traversal reads: the same nodes with fp16 planes on the grid of the bounds
    QGrid     grid = { { 0, 0, 0 }, { 1, 1, 1 } };
    TriRec*   tris = nullptr;
    NrmRec*   nrms = nullptr;
    bool      built = false;
    float     bounds[6] = { 0, 0, 0, 0, 0, 0 };
    float     scale = 1.0f;          // max |bounds|
    uint32_t  depth = 0;
};

// device block zeroed before every dispatch: counters, ray shards, error flag
struct CounterBlock {
    unsigned long long counters[16];
    static_assert(C_COUNT <= 16, "CounterBlock::counters holds every rr::Counter");
    uint32_t shards[RAY_SHARDS];
    uint32_t error;
    uint32_t pad[3];
};

} // namespace

struct rr_context {
    int device = 0;
    int n_cus = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string err;

    std::vector<MeshRes> meshes;

    float4* d_env = nullptr;
    int env_w = 0, env_h = 0;

    // TLAS
    std::vector<rr_instance_desc> inst_host;
    InstDev* d_insts = nullptr;
    BvhNode* d_pool_nodes = nullptr;   // fp32 TLAS nodes (builder output)
    QNode*   d_pool_qnodes = nullptr;  // flattened scene as traversal reads it: TLAS nodes, then every BLAS in use
    QGrid    scene_grid = { { 0, 0, 0 }, { 1, 1, 1 } };
    TriRec*  d_pool_tris = nullptr;
    NrmRec*  d_pool_nrms = nullptr;
    uint32_t n_pool_tris = 0;
    uint32_t n_insts = 0, tlas_depth = 0;
    bool tlas_built = false;
    bool single_identity = false;
    float scene_scale = 1.0f;
    float scene_bounds[6] = { 0, 0, 0, 0, 0, 0 };   // world-space box of the whole scene (the TLAS root)

    float*   d_screen = nullptr;     // GenerateCameraRay's screen coordinates for frames of screen_w x screen_h: sx[W], sy[H]
    uint32_t screen_w = 0, screen_h = 0;

    rr_scene_constants cam;
    bool cam_set = false;
    CamDev* d_cams = nullptr;        // device-side constant buffer(s), one per depth slice
    size_t cams_cap = 0;
    // page-locked staging for the constants (a copy from pageable memory makes the runtime stage it itself, a few hundred
    // microseconds in front of every launch): four slots in turn, each guarded by an event recorded behind its copy
    static constexpr int CAM_SLOTS = 4;
    void*      h_cams[CAM_SLOTS] = {};
    size_t     h_cams_cap[CAM_SLOTS] = {};
    hipEvent_t h_cams_ev[CAM_SLOTS] = {};
    bool       h_cams_busy[CAM_SLOTS] = {};
    uint32_t   h_cams_next = 0;

    uint32_t tile_rank = 0, tile_world = 1;

    // lanes: internal streams whose launches may overlap each other (rr_render_orbit_sharded_lane)
    static constexpr uint32_t MAX_LANES = 4;
    hipStream_t lane_stream[MAX_LANES] = {};
    hipEvent_t  lane_fork[MAX_LANES] = {}, lane_done[MAX_LANES] = {};
    CamDev*     lane_cams[MAX_LANES] = {};
    size_t      lane_cams_cap[MAX_LANES] = {};
    bool        lane_busy[MAX_LANES] = {};
    uint32_t    frames_in_flight = 2;    // rr_set_frames_in_flight: launches of rr_render_orbit that may overlap
    size_t      frame_base = 0;          // element offset of the most recent dispatch inside d_rgba8 / d_f32

    // frame
    uint32_t W = 0, H = 0, frame_world = 0, frame_depth = 1;
    uint32_t* d_rgba8 = nullptr;     // world==1: W*H; else local tiles
    float4*   d_f32 = nullptr;
    uint32_t* d_assembled = nullptr; // rank-0 raster after rr_assemble_tiles
    size_t    rgba_elems = 0, f32_elems = 0, assembled_elems = 0;
    bool      have_f32 = false, have_frame = false, have_assembled = false;
    uint64_t  last_pixels = 0;
    uint64_t  accum_pixels = 0;      // pixels of all dispatches since the counters were last zeroed
    bool      last_stats = false;

    CounterBlock* d_cnt = nullptr;
    CounterBlock* d_cnt_trial = nullptr;     // what the two renders of a kernel-choice measurement count into (thrown away)
    uint32_t* d_park[MAX_LANES + 1] = {};   // k_render_lds: parked reflected rays, one slab per stream slot like the tickets
    size_t    park_bytes[MAX_LANES + 1] = {};
    // k_render_lds or k_render_fused for launches of many slices?  Neither wins everywhere (sphere.obj / shell.obj 1080p: the LDS
    // kernel by 6 % all round the orbit; monkey.obj: the L1-fed one by 2 %), and which does depends on how busy the texture
    // path is, not on anything the host can see.  So the first two eligible launches of a scene are timed with HIP events, one
    // on each kernel (adjacent slices of the same orbit), and the faster per slice renders the rest.  Frames are bit-identical.
    char       last_kernel_name[96] = "";
    uint32_t   last_kernel = 0;      // render kernel of the last dispatch: 0 k_render_fused, 1 k_render_lds, 2 k_render_paths, 3 experimental
    uint32_t* d_tickets = nullptr;   // k_render_lds ticket words: one block per stream a launch can be on (lanes, then the context's stream)

    // diagnostics switches, read once at rr_create (never needed for correct results)
    int  dbg_kernel = 0;             // RR_DEBUG_KERNEL: 0 default (measured choice), 1 "fused", 4 "lds", 5 "paths", 10 "stream": that kernel wherever it can render the launch
    int  dbg_stack = 0;              // RR_DEBUG_STACK
    int  dbg_ticket_blocks = 0;      // RR_DEBUG_TICKET: 1 = k_render_lds treats the whole frame as the mesh rectangle, 2 = no rectangle
    bool dbg_group_trace = true;     // RR_DEBUG_GROUP_TRACE=0: k_render_paths never shares a ray between lanes
    bool dbg_async_set = false;
    uint32_t dbg_async[2] = { 2, 2 };    // RR_DEBUG_ASYNC="step,shade": issue thresholds of k_stream_rays in sixteenths of the live lanes
    bool dbg_tile_order = true;      // RR_DEBUG_TILE_ORDER=0: tiles in image order (DispatchDev::rt_*)
    int  dbg_stream_waves = 6;       // (the stream renderer's waves per SIMD are a build-time constant now: -DRR_STREAM_WPS)
    bool dbg_tlas32 = false;         // RR_DEBUG_TLAS32: two-level scenes keep 32-bit stack entries and register-parked rays
    int  dbg_shape = 0;              // RR_DEBUG_SHAPE: first k_render_lds workgroup shape to consider (rr_launch.h)
    std::string dbg_diag;            // RR_DEBUG_DIAG: file that receives per-wave diagnostics of Depth-1 dispatches

    // timing
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> kev;     // pairs
    uint32_t kev_used = 0;

    // k_stream_* (rr_render_stream.hip): ray queues, leaf slots and pixel marks of one pass; grown on demand, never shrunk.
    // One set per stream a launch can be on (the lanes, then the context's stream: launches on one stream are ordered, launches
    // on different lanes overlap), allocated when that stream first renders with the stream renderer.
    StreamDev strm[MAX_LANES + 1] = {};
    size_t    strm_cap[MAX_LANES + 1] = {};          // queue entries allocated (each of the two queues)
    size_t    strm_pixels[MAX_LANES + 1] = {};       // pixel ordinals the slots / marks are allocated for
    size_t    strm_budget = 0;                       // bytes one set may take (stream_budget)
    // Which of two kernels renders a class of launches is MEASURED, once per scene and launch shape: the scene's first dispatch of
    // a class runs on the default kernel (clocks come up), the second is rendered by both candidates, each bracketed by HIP events
    // (the frames are bit-identical, the dispatch just costs three extra launches), and again on the next dispatch of the shape; the default renders every later one unless
    // the alternative took less than 98 % of its time over the two.
    // rr_build_tlas starts every measurement afresh; a launch shape (frame size, bounce limits, launch depth 1 / 2 / 3-7 / 8-23 /
    // 24-47 / 48 and up) has its own choice -- a class remembers its four most recent shapes, so a caller that alternates between two
    // depths does not measure again at every switch -- and a rectangle share that doubles or halves renews a shape's.
    // Classes: two-level scenes (k_render_fused / k_stream_*), launches of many slices of the reference's scene
    // (k_render_fused / k_render_lds), launches of one or two slices (k_render_fused / k_render_paths).
    struct KernelChoice {
        int choice = 0;              // 0 undecided, 1 candidate A (k_render_fused), 2 candidate B
        uint32_t seen = 0;
        unsigned long long key = 0;  // the launch shape the choice was measured for
        double share = 0.0;          // rectangle share of the frame at the measurement
        float ms[2] = { 0.0f, 0.0f };
        bool valid = false;
        unsigned long long stamp = 0;
    };
    struct ChoiceClass {
        KernelChoice e[4];
        unsigned long long clock = 0;
        KernelChoice* find(unsigned long long key)      // the shape's entry; a new shape takes the place of the least recently used
        {
            ++clock;
            KernelChoice* lru = &e[0];
            for (KernelChoice& x : e) {
                if (x.valid && x.key == key) { x.stamp = clock; return &x; }
                if (x.stamp < lru->stamp) lru = &x;
            }
            *lru = KernelChoice();
            lru->valid = true; lru->key = key; lru->stamp = clock;
            return lru;
        }
        const KernelChoice* peek(unsigned long long key) const
        {
            for (const KernelChoice& x : e) if (x.valid && x.key == key) return &x;
            return nullptr;
        }
    };
    static unsigned long long choice_key(uint32_t width, uint32_t height, const rr_dispatch_params& p, uint32_t depth)
    {
        return ((unsigned long long)width << 48) ^ ((unsigned long long)height << 32) ^ ((unsigned long long)(uint32_t)p.max_refract << 8) ^
               ((unsigned long long)(uint32_t)p.max_reflect << 4) ^ (depth <= 2 ? depth : depth < 8 ? 3u : depth < 24 ? 4u : depth < 48 ? 5u : 6u);
    }
    ChoiceClass ch_tlas, ch_many, ch_few;
    hipEvent_t ch_ev[4] = {};

    // trace_rays scratch
    rr_ray_dev* d_rays = nullptr;
    rr_hit_dev* d_hits = nullptr;
    uint32_t ray_cap = 0;
};

namespace {

int fail(rr_context* ctx, int code, const char* what, hipError_t e = hipSuccess)
{
    if (ctx) {
        ctx->err = what;
        if (e != hipSuccess) { ctx->err += ": "; ctx->err += hipGetErrorString(e); }
    }
    return code;
}

#define RR_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) return fail(ctx, e_ == hipErrorOutOfMemory ? RR_ERR_OUT_OF_MEMORY : RR_ERR_DEVICE, #call, e_); \
    } while (0)

template <class T> void dfree(T*& p) { if (p) { (void)hipFree(p); p = nullptr; } }

uint32_t next_pow2(uint32_t v) { uint32_t p = 1; while (p < v) p <<= 1; return p; }

// world -> object inverse of a 3x4 affine (adjugate / det, fixed operation order; mirrored by the oracle)
void affine_inverse(const float t[12], float inv[12])
{
    float a = t[0], b = t[1], c = t[2], d = t[4], e = t[5], f = t[6], g = t[8], h = t[9], i = t[10];
    float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    float det = (a * c00 + b * c01) + c * c02;
    float r = 1.0f / det;
    inv[0] = c00 * r; inv[1] = (c * h - b * i) * r; inv[2] = (b * f - c * e) * r;
    inv[4] = c01 * r; inv[5] = (a * i - c * g) * r; inv[6] = (c * d - a * f) * r;
    inv[8] = c02 * r; inv[9] = (b * g - a * h) * r; inv[10] = (a * e - b * d) * r;
    float tx = t[3], ty = t[7], tz = t[11];
    inv[3] = -((inv[0] * tx + inv[1] * ty) + inv[2] * tz);
    inv[7] = -((inv[4] * tx + inv[5] * ty) + inv[6] * tz);
    inv[11] = -((inv[8] * tx + inv[9] * ty) + inv[10] * tz);
}

float ord2f_host(uint32_t u)
{
    uint32_t v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &v, 4);
    return f;
}

struct BuildScratch {
    BuildBuffers b{};
    void* raw = nullptr;
    ~BuildScratch() { if (raw) (void)hipFree(raw); }
};

// one allocation carved into the builder's scratch arrays (16-byte aligned pieces)
int alloc_build(rr_context* ctx, uint32_t n, BuildScratch& s)
{
    const uint32_t n_pad = next_pow2(n);
    auto al = [](size_t v) { return (v + 255u) & ~(size_t)255u; };
    size_t o_box = 0;
    size_t o_keys = o_box + al((size_t)n * 6 * 4);
    size_t o_parent = o_keys + al((size_t)n_pad * 8);
    size_t o_child = o_parent + al((size_t)(2 * (size_t)n) * 4);
    size_t o_nbox = o_child + al((size_t)(2 * (size_t)n) * 4);
    size_t o_visit = o_nbox + al((size_t)(2 * (size_t)n) * 6 * 4);
    size_t o_scene = o_visit + al((size_t)n * 4);
    size_t o_depth = o_scene + al(6 * 4);
    size_t o_ploc = o_depth + al(4);
    size_t total = o_ploc + al((size_t)n * 8);
    RR_HIP(hipMalloc(&s.raw, total));
    char* base = (char*)s.raw;
    s.b.n = n; s.b.n_pad = n_pad;
    s.b.prim_box = (float*)(base + o_box);
    s.b.keys = (unsigned long long*)(base + o_keys);
    s.b.parent = (int32_t*)(base + o_parent);
    s.b.child = (int32_t*)(base + o_child);
    s.b.node_box = (float*)(base + o_nbox);
    s.b.visit = (uint32_t*)(base + o_visit);
    s.b.scene_box = (uint32_t*)(base + o_scene);
    s.b.depth = (uint32_t*)(base + o_depth);
    s.b.ploc = (uint32_t*)(base + o_ploc);
    return RR_OK;
}

int use_device(rr_context* ctx)
{
    if (!ctx) return RR_ERR_INVALID_ARGUMENT;
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, RR_ERR_DEVICE, "hipSetDevice", e);
    return RR_OK;
}

// grid over a box {lo[3], hi[3]}: 65530 cells span the extent, counted from the centre (planes are stored as fp16
// cell counts, |q| <= 32768); a flat axis gets a tiny positive cell
QGrid make_grid(const float b[6])
{
    QGrid g;
    for (int k = 0; k < 3; ++k) {
        const float ext = b[3 + k] - b[k];
        const float mag = std::max(std::max(std::fabs(b[k]), std::fabs(b[3 + k])), 1e-30f);
        g.cell[k] = std::max(ext, mag * 1e-6f) / 65530.0f;
        g.org[k] = b[k] + 32765.0f * g.cell[k];      // fp16 planes are signed: the grid origin is the centre of the box
    }
    return g;
}

void fill_scene(const rr_context* ctx, SceneDev& sc)
{
    memset(&sc, 0, sizeof sc);
    const MeshRes* m0 = nullptr;
    if (ctx->single_identity) m0 = &ctx->meshes[(size_t)ctx->inst_host[0].blas];
    if (m0) {
        sc.blas0.nodes = m0->qnodes; sc.blas0.grid = m0->grid; sc.blas0.tris = m0->tris; sc.blas0.nrms = m0->nrms;
        sc.blas0.n_tris = m0->n_tris; sc.blas0.depth = m0->depth; sc.blas0.scale = m0->scale;
    }
    sc.pool_nodes = ctx->d_pool_qnodes; sc.grid = ctx->scene_grid; sc.pool_tris = ctx->d_pool_tris; sc.pool_nrms = ctx->d_pool_nrms;
    sc.insts = ctx->d_insts;
    sc.n_insts = ctx->n_insts;
    sc.n_pool_tris = ctx->n_pool_tris;
    sc.single_identity = ctx->single_identity ? 1u : 0u;
    sc.scale = ctx->scene_scale;
    sc.env = ctx->d_env;
    sc.env_w = ctx->env_w; sc.env_h = ctx->env_h;
}

// deepest traversal stack the scene can need (near child followed, far child pushed)
uint32_t scene_stack_need(const rr_context* ctx)
{
    uint32_t blas_max = 0;
    for (uint32_t i = 0; i < ctx->n_insts; ++i) {
        const MeshRes& m = ctx->meshes[(size_t)ctx->inst_host[i].blas];
        if (m.depth > blas_max) blas_max = m.depth;
    }
    return ctx->single_identity ? blas_max : blas_max + ctx->tlas_depth;
}

} // namespace

extern "C" {

uint32_t rr_abi_version(void) { return RRDXR_ABI_VERSION; }

void rr_default_dispatch_params(rr_dispatch_params* p)
{
    if (!p) return;
    p->max_refract = 5;
    p->max_reflect = 2;
    p->ior = 1.3f;
    p->tmin_primary = 0.0001f;
    p->tmax_primary = 100.0f;
    p->tmin_secondary = 0.001f;
    p->tmax_secondary = 1000.0f;
    p->flags = 0;
}

int rr_create(int device_ordinal, rr_context** out)
{
    if (!out) return RR_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RR_ERR_NO_DEVICE;
    if (device_ordinal < 0 || device_ordinal >= n) return RR_ERR_INVALID_ARGUMENT;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return RR_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RR_ERR_NO_DEVICE;   // the code object is gfx950 only
    rr_context* ctx = new (std::nothrow) rr_context();
    if (!ctx) return RR_ERR_OUT_OF_MEMORY;
    ctx->device = device_ordinal;
    ctx->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipSetDevice(device_ordinal) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&ctx->d_cnt, sizeof(CounterBlock)) != hipSuccess ||
        hipMalloc(&ctx->d_tickets, (rr_context::MAX_LANES + 1) * LDS_TICKET_WORDS * sizeof(uint32_t)) != hipSuccess) {
        if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
        dfree(ctx->d_cnt);
        delete ctx;
        return RR_ERR_DEVICE;
    }
    ctx->stream = ctx->own_stream;
    (void)hipMemsetAsync(ctx->d_cnt, 0, sizeof(CounterBlock), ctx->stream);
    (void)hipMemsetAsync(ctx->d_tickets, 0, (rr_context::MAX_LANES + 1) * LDS_TICKET_WORDS * sizeof(uint32_t), ctx->stream);   // the kernel leaves them zero
    if (const char* e = getenv("RR_DEBUG_KERNEL"))
        ctx->dbg_kernel = !strcmp(e, "fused") ? 1 : !strcmp(e, "lds") ? 4 : !strcmp(e, "paths") ? 5 : !strcmp(e, "stream") ? 10 : 0;
    if (const char* e = getenv("RR_DEBUG_STACK")) ctx->dbg_stack = atoi(e);
    if (const char* e = getenv("RR_DEBUG_TICKET")) ctx->dbg_ticket_blocks = atoi(e);
    if (const char* e = getenv("RR_DEBUG_SHAPE")) ctx->dbg_shape = atoi(e);
    if (const char* e = getenv("RR_DEBUG_TLAS32")) ctx->dbg_tlas32 = atoi(e) != 0;
    if (const char* e = getenv("RR_DEBUG_TILE_ORDER")) ctx->dbg_tile_order = atoi(e) != 0;
    if (const char* e = getenv("RR_DEBUG_ASYNC")) { unsigned l = 2, sh = 2; if (sscanf(e, "%u,%u", &l, &sh) == 2 && l >= 1 && sh >= 1) { ctx->dbg_async[0] = l; ctx->dbg_async[1] = sh; ctx->dbg_async_set = true; } }
    if (const char* e = getenv("RR_DEBUG_DIAG")) ctx->dbg_diag = e;
    if (const char* e = getenv("RR_DEBUG_GROUP_TRACE")) ctx->dbg_group_trace = atoi(e) != 0;
    *out = ctx;
    return RR_OK;
}

int rr_destroy(rr_context* ctx)
{
    if (!ctx) return RR_ERR_INVALID_ARGUMENT;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (uint32_t l = 0; l < rr_context::MAX_LANES; ++l) {
        if (!ctx->lane_stream[l]) continue;
        (void)hipStreamSynchronize(ctx->lane_stream[l]);
        (void)hipStreamDestroy(ctx->lane_stream[l]);
        (void)hipEventDestroy(ctx->lane_fork[l]);
        (void)hipEventDestroy(ctx->lane_done[l]);
        dfree(ctx->lane_cams[l]);
    }
    for (MeshRes& m : ctx->meshes) { dfree(m.d_verts); dfree(m.d_idx); dfree(m.nodes); dfree(m.qnodes); dfree(m.tris); dfree(m.nrms); }
    dfree(ctx->d_env); dfree(ctx->d_insts); dfree(ctx->d_pool_nodes); dfree(ctx->d_pool_qnodes); dfree(ctx->d_pool_tris); dfree(ctx->d_pool_nrms); dfree(ctx->d_rgba8); dfree(ctx->d_f32);
    for (StreamDev& sd : ctx->strm) { dfree(sd.q[0]); dfree(sd.q[1]); dfree(sd.fill[0]); dfree(sd.fill[1]); dfree(sd.heads); dfree(sd.slots); dfree(sd.pending); }
    for (hipEvent_t e : ctx->ch_ev) if (e) (void)hipEventDestroy(e);
    dfree(ctx->d_assembled); dfree(ctx->d_cnt); dfree(ctx->d_cnt_trial); dfree(ctx->d_tickets); dfree(ctx->d_screen);
    for (uint32_t l = 0; l <= rr_context::MAX_LANES; ++l) dfree(ctx->d_park[l]); dfree(ctx->d_rays); dfree(ctx->d_hits); dfree(ctx->d_cams);
    for (int k = 0; k < rr_context::CAM_SLOTS; ++k) { if (ctx->h_cams[k]) (void)hipHostFree(ctx->h_cams[k]); if (ctx->h_cams_ev[k]) (void)hipEventDestroy(ctx->h_cams_ev[k]); }
    if (ctx->ev_begin) (void)hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) (void)hipEventDestroy(ctx->ev_end);
    for (hipEvent_t e : ctx->kev) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return RR_OK;
}

const char* rr_last_error(const rr_context* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rr_set_stream(rr_context* ctx, void* hip_stream)
{
    if (int r = use_device(ctx)) return r;
    RR_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;         // NULL is a stream too: HIP's default stream
    return RR_OK;
}

int rr_reset_stream(rr_context* ctx)
{
    if (int r = use_device(ctx)) return r;
    RR_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = ctx->own_stream;
    return RR_OK;
}

namespace {
// the context's stream waits for everything submitted to the lanes
int join_lanes(rr_context* ctx)
{
    for (uint32_t l = 0; l < rr_context::MAX_LANES; ++l)
        if (ctx->lane_busy[l]) {
            RR_HIP(hipStreamWaitEvent(ctx->stream, ctx->lane_done[l], 0));
            ctx->lane_busy[l] = false;
        }
    return RR_OK;
}
} // namespace

int rr_wait(rr_context* ctx)
{
    if (int r = use_device(ctx)) return r;
    if (int r = join_lanes(ctx)) return r;
    RR_HIP(hipStreamSynchronize(ctx->stream));
    return RR_OK;
}

int rr_set_frames_in_flight(rr_context* ctx, uint32_t n)
{
    if (int r = use_device(ctx)) return r;
    if (n == 0 || n > rr_context::MAX_LANES) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_set_frames_in_flight: 1..4");
    ctx->frames_in_flight = n;
    return RR_OK;
}

int rr_lane_join(rr_context* ctx, uint32_t lane)
{
    if (int r = use_device(ctx)) return r;
    if (lane >= rr_context::MAX_LANES) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_lane_join: lane out of range");
    if (ctx->lane_busy[lane]) {
        RR_HIP(hipStreamWaitEvent(ctx->stream, ctx->lane_done[lane], 0));
        ctx->lane_busy[lane] = false;
    }
    return RR_OK;
}

int rr_upload_mesh(rr_context* ctx, const rr_vertex* verts, uint32_t n_verts, const uint32_t* indices, uint32_t n_indices,
                   uint32_t* mesh_id)
{
    if (int r = use_device(ctx)) return r;
    if (!verts || !indices || !mesh_id || n_verts == 0 || n_indices < 3 || n_indices % 3 != 0)
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_upload_mesh: need >= 1 triangle, n_indices % 3 == 0");
    for (uint32_t i = 0; i < n_indices; ++i)
        if (indices[i] >= n_verts) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_upload_mesh: index out of range");
    if (rr_host_validate_positions(verts, n_verts, nullptr) != RR_OK)
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_upload_mesh: non-finite or huge (> 1e18) vertex position");
    MeshRes m;
    m.n_verts = n_verts; m.n_idx = n_indices; m.n_tris = n_indices / 3;
    RR_HIP(hipMalloc(&m.d_verts, (size_t)n_verts * sizeof(rr_vertex)));
    hipError_t e = hipMalloc(&m.d_idx, (size_t)n_indices * 4);
    if (e != hipSuccess) { dfree(m.d_verts); return fail(ctx, RR_ERR_OUT_OF_MEMORY, "hipMalloc(indices)", e); }
    // Mesh.cpp:76-79,88-91: memcpy into the mapped upload buffers
    e = hipMemcpyAsync(m.d_verts, verts, (size_t)n_verts * sizeof(rr_vertex), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(m.d_idx, indices, (size_t)n_indices * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);   // caller keeps ownership of the host arrays
    if (e != hipSuccess) { dfree(m.d_verts); dfree(m.d_idx); return fail(ctx, RR_ERR_DEVICE, "mesh upload", e); }
    ctx->meshes.push_back(m);
    *mesh_id = (uint32_t)ctx->meshes.size() - 1u;
    return RR_OK;
}

int rr_upload_envmap(rr_context* ctx, const float* rgb, int32_t w, int32_t h)
{
    if (int r = use_device(ctx)) return r;
    if (!rgb || w <= 0 || h <= 0) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_upload_envmap: null data or empty size");
    const size_t n = (size_t)w * (size_t)h;
    float* staging = nullptr;
    float4* env = nullptr;
    RR_HIP(hipMalloc(&staging, n * 12));
    hipError_t e = hipMalloc(&env, n * 16);
    if (e == hipSuccess) e = hipMemcpyAsync(staging, rgb, n * 12, hipMemcpyHostToDevice, ctx->stream);   // RowPitch = x*3*4 (:128)
    if (e == hipSuccess) e = launch_env_pad(staging, env, (uint32_t)n, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(staging);
    if (e != hipSuccess) { if (env) (void)hipFree(env); return fail(ctx, RR_ERR_DEVICE, "env upload", e); }
    dfree(ctx->d_env);
    ctx->d_env = env; ctx->env_w = w; ctx->env_h = h;
    return RR_OK;
}

int rr_build_blas(rr_context* ctx, uint32_t mesh_id) { return rr_build_blas_ex(ctx, mesh_id, RR_BUILD_PREFER_FAST_TRACE); }

int rr_build_blas_ex(rr_context* ctx, uint32_t mesh_id, uint32_t flags)
{
    const Range range_("rr_build_blas");
    if (int r = use_device(ctx)) return r;
    if (mesh_id >= ctx->meshes.size()) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_build_blas: unknown mesh id");
    MeshRes& m = ctx->meshes[mesh_id];
    const uint32_t n = m.n_tris;
    if ((uint64_t)n * sizeof(QNode) >= 0x7fffffffull) return fail(ctx, RR_ERR_UNSUPPORTED, "rr_build_blas: mesh too large for 31-bit node refs");
    BuildScratch s;
    if (int r = alloc_build(ctx, n, s)) return r;
    dfree(m.nodes); dfree(m.qnodes); dfree(m.tris); dfree(m.nrms);
    m.built = false;
    RR_HIP(hipMalloc(&m.nodes, (size_t)(n > 1 ? n - 1 : 1) * sizeof(BvhNode)));
    RR_HIP(hipMalloc(&m.qnodes, (size_t)(n > 1 ? n - 1 : 1) * sizeof(QNode)));
    RR_HIP(hipMalloc(&m.tris, (size_t)n * sizeof(TriRec)));
    RR_HIP(hipMalloc(&m.nrms, (size_t)n * sizeof(NrmRec)));
    s.b.nodes = m.nodes;
    RR_HIP(launch_tri_setup(m.d_verts, m.d_idx, n, s.b, ctx->stream));
    if ((flags & RR_BUILD_PREFER_FAST_TRACE) && !(flags & RR_BUILD_PREFER_FAST_BUILD) && n > 1 && n <= PLOC_MAX_PRIMS)
        RR_HIP(launch_ploc(s.b, ctx->stream));          // clustered hierarchy (fewer node visits)
    else
        RR_HIP(launch_lbvh(s.b, ctx->stream));          // Karras radix tree (fastest build, any size)
    RR_HIP(launch_pack_tris(m.d_verts, m.d_idx, s.b, m.tris, m.nrms, ctx->stream));
    uint32_t sb[6], depth = 0;
    RR_HIP(hipMemcpyAsync(sb, s.b.scene_box, sizeof sb, hipMemcpyDeviceToHost, ctx->stream));
    RR_HIP(hipMemcpyAsync(&depth, s.b.depth, 4, hipMemcpyDeviceToHost, ctx->stream));
    RR_HIP(hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 6; ++k) m.bounds[k] = ord2f_host(sb[k]);
    m.scale = 0.0f;
    for (int k = 0; k < 6; ++k) m.scale = std::max(m.scale, std::fabs(m.bounds[k]));
    m.depth = depth;
    m.grid = make_grid(m.bounds);
    RR_HIP(launch_quantize_nodes(m.qnodes, m.nodes, n > 1 ? n - 1 : 1, m.grid, 0, 0, ctx->stream));
    if (depth > 64) return fail(ctx, RR_ERR_UNSUPPORTED, "rr_build_blas: LBVH deeper than the 64-entry traversal stack");
    m.built = true;
    ctx->tlas_built = false;      // any TLAS built before refers to the old BLAS
    return RR_OK;
}

int rr_build_tlas(rr_context* ctx, const rr_instance_desc* instances, uint32_t n)
{
    const Range range_("rr_build_tlas");
    if (int r = use_device(ctx)) return r;
    if (!instances || n == 0) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_build_tlas: need >= 1 instance");
    for (uint32_t i = 0; i < n; ++i) {
        if (instances[i].blas >= ctx->meshes.size()) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_build_tlas: unknown BLAS");
        if (!ctx->meshes[(size_t)instances[i].blas].built) return fail(ctx, RR_ERR_STATE, "rr_build_tlas: BLAS not built");
    }
    static const float ident[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    // pool layout: nodes [0, n_tlas) TLAS, then each distinct BLAS; triangles / normals concatenated
    const uint32_t n_tlas = n > 1 ? n - 1 : 1;
    std::vector<uint32_t> node_off(ctx->meshes.size(), 0xffffffffu), tri_off(ctx->meshes.size(), 0);
    uint32_t n_pool_nodes = n_tlas, n_pool_tris = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const size_t mi = (size_t)instances[i].blas;
        if (node_off[mi] != 0xffffffffu) continue;
        const MeshRes& m = ctx->meshes[mi];
        node_off[mi] = n_pool_nodes; tri_off[mi] = n_pool_tris;
        n_pool_nodes += m.n_tris > 1 ? m.n_tris - 1 : 1;
        n_pool_tris += m.n_tris;
    }
    if ((uint64_t)n_pool_tris + n >= 0x7fffffffull || (uint64_t)n_pool_nodes * sizeof(QNode) >= 0x7fffffffull)
        return fail(ctx, RR_ERR_UNSUPPORTED, "rr_build_tlas: scene too large for 31-bit node / leaf refs");
    std::vector<InstDev> host(n);
    float scene_scale = 0.0f;
    std::vector<float> xb((size_t)n * 18);
    for (uint32_t i = 0; i < n; ++i) {
        const rr_instance_desc& d = instances[i];
        const MeshRes& m = ctx->meshes[(size_t)d.blas];
        InstDev& o = host[i];
        memset(&o, 0, sizeof o);
        o.identity = memcmp(d.transform, ident, sizeof ident) == 0 ? 1u : 0u;
        if (o.identity) memcpy(o.inv, ident, sizeof ident);
        else {
            affine_inverse(d.transform, o.inv);
            for (int k = 0; k < 12; ++k)
                if (!std::isfinite(o.inv[k])) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_build_tlas: singular instance transform");
        }
        o.root = node_off[(size_t)d.blas] * (uint32_t)sizeof(QNode);     // byte offset, like every internal child ref
        o.scale = m.scale;
        o.grid = m.grid;
        for (int c = 0; c < 8; ++c) {       // world-space extent of the instance (for the TLAS box padding)
            const float x = (c & 1) ? m.bounds[3] : m.bounds[0], y = (c & 2) ? m.bounds[4] : m.bounds[1], z = (c & 4) ? m.bounds[5] : m.bounds[2];
            for (int r = 0; r < 3; ++r)
                scene_scale = std::max(scene_scale, std::fabs(d.transform[4 * r] * x + d.transform[4 * r + 1] * y + d.transform[4 * r + 2] * z + d.transform[4 * r + 3]));
        }
        o.flags = d.hitgroup_flags >> 24;
        o.mask = d.instance_id_mask >> 24;
        memcpy(&xb[(size_t)i * 12], d.transform, 48);
        memcpy(&xb[(size_t)n * 12 + (size_t)i * 6], m.bounds, 24);
    }
    ctx->tlas_built = false;
    dfree(ctx->d_insts); dfree(ctx->d_pool_nodes); dfree(ctx->d_pool_qnodes); dfree(ctx->d_pool_tris); dfree(ctx->d_pool_nrms);
    RR_HIP(hipMalloc(&ctx->d_insts, (size_t)n * sizeof(InstDev)));
    RR_HIP(hipMalloc(&ctx->d_pool_nodes, (size_t)n_tlas * sizeof(BvhNode)));
    RR_HIP(hipMalloc(&ctx->d_pool_qnodes, (size_t)n_pool_nodes * sizeof(QNode)));
    RR_HIP(hipMalloc(&ctx->d_pool_tris, (size_t)n_pool_tris * sizeof(TriRec)));
    RR_HIP(hipMalloc(&ctx->d_pool_nrms, (size_t)n_pool_tris * sizeof(NrmRec)));
    float* d_xb = nullptr;
    RR_HIP(hipMalloc(&d_xb, xb.size() * 4));
    BuildScratch s;
    int rc = alloc_build(ctx, n, s);
    hipError_t e = hipSuccess;
    uint32_t depth = 0;
    if (rc == RR_OK) {
        s.b.nodes = ctx->d_pool_nodes;
        s.b.leaf_ref_prim = 1;
        s.b.leaf_base = n_pool_tris;                  // an instance leaf is ~(n_pool_tris + instance index)
        e = hipMemcpyAsync(ctx->d_insts, host.data(), (size_t)n * sizeof(InstDev), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_xb, xb.data(), xb.size() * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = launch_inst_setup(ctx->d_insts, d_xb, n, s.b, ctx->stream);
        // (the top level keeps the Karras hierarchy: the clustered builder, tried on it in round 3, makes the 1 024-instance grid
        // 5 % slower on both renderers -- on a regular lattice every merged-box area ties)
        if (e == hipSuccess) e = launch_lbvh(s.b, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&depth, s.b.depth, 4, hipMemcpyDeviceToHost, ctx->stream);
        // scene grid = the box of the TLAS root (node 0 holds the boxes of its two children)
        BvhNode root;
        if (e == hipSuccess) e = hipMemcpyAsync(&root, ctx->d_pool_nodes, sizeof root, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) {
            float sb[6] = { 3.0e38f, 3.0e38f, 3.0e38f, -3.0e38f, -3.0e38f, -3.0e38f };
            for (int k = 0; k < 2; ++k) {
                if (!(root.lox[k] <= root.hix[k])) continue;           // the empty second child of a one-instance TLAS
                sb[0] = std::min(sb[0], root.lox[k]); sb[1] = std::min(sb[1], root.loy[k]); sb[2] = std::min(sb[2], root.loz[k]);
                sb[3] = std::max(sb[3], root.hix[k]); sb[4] = std::max(sb[4], root.hiy[k]); sb[5] = std::max(sb[5], root.hiz[k]);
            }
            ctx->scene_grid = make_grid(sb);
            memcpy(ctx->scene_bounds, sb, sizeof sb);
            e = launch_quantize_nodes(ctx->d_pool_qnodes, ctx->d_pool_nodes, n_tlas, ctx->scene_grid, 0, 0, ctx->stream);
        }
        for (size_t mi = 0; mi < ctx->meshes.size() && e == hipSuccess; ++mi) {
            if (node_off[mi] == 0xffffffffu) continue;
            const MeshRes& m = ctx->meshes[mi];
            e = launch_quantize_nodes(ctx->d_pool_qnodes + node_off[mi], m.nodes, m.n_tris > 1 ? m.n_tris - 1 : 1, m.grid, node_off[mi], tri_off[mi], ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_pool_tris + tri_off[mi], m.tris, (size_t)m.n_tris * sizeof(TriRec), hipMemcpyDeviceToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(ctx->d_pool_nrms + tri_off[mi], m.nrms, (size_t)m.n_tris * sizeof(NrmRec), hipMemcpyDeviceToDevice, ctx->stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(d_xb);
    if (rc != RR_OK) return rc;
    if (e != hipSuccess) return fail(ctx, RR_ERR_DEVICE, "TLAS build", e);
    ctx->n_pool_tris = n_pool_tris;
    ctx->inst_host.assign(instances, instances + n);
    ctx->n_insts = n;
    ctx->scene_scale = scene_scale;
    ctx->tlas_depth = depth;
    const rr_instance_desc& d0 = instances[0];
    ctx->single_identity = n == 1 && host[0].identity && (d0.hitgroup_flags >> 24) == 0 && ((d0.instance_id_mask >> 24) & 0xffu) != 0;
    if (scene_stack_need(ctx) > 64) return fail(ctx, RR_ERR_UNSUPPORTED, "rr_build_tlas: TLAS+BLAS deeper than the 64-entry stack");
    ctx->tlas_built = true;
    ctx->ch_tlas = rr_context::ChoiceClass(); ctx->ch_many = rr_context::ChoiceClass(); ctx->ch_few = rr_context::ChoiceClass();      // a new scene: the kernels are chosen afresh
    return RR_OK;
}

int rr_set_camera(rr_context* ctx, const rr_scene_constants* constants)
{
    if (!ctx || !constants) return RR_ERR_INVALID_ARGUMENT;
    ctx->cam = *constants;
    ctx->cam_set = true;
    return RR_OK;
}

int rr_set_tile_partition(rr_context* ctx, uint32_t rank, uint32_t world)
{
    if (!ctx) return RR_ERR_INVALID_ARGUMENT;
    if (world == 0 || rank >= world) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_set_tile_partition: need rank < world");
    ctx->tile_rank = rank; ctx->tile_world = world;
    return RR_OK;
}

static void tile_counts(uint32_t W, uint32_t H, uint32_t rank, uint32_t world, uint32_t& tiles_x, uint32_t& n_tiles,
                        uint32_t& local, uint32_t& max_local)
{
    tiles_x = (W + TILE - 1) / TILE;
    n_tiles = tiles_x * ((H + TILE - 1) / TILE);
    local = n_tiles > rank ? (n_tiles - rank + world - 1) / world : 0;
    max_local = (n_tiles + world - 1) / world;
}

int rr_local_tile_count(rr_context* ctx, uint32_t width, uint32_t height, uint32_t* n_tiles, uint32_t* max_tiles_any_rank)
{
    if (!ctx || width == 0 || height == 0) return RR_ERR_INVALID_ARGUMENT;
    uint32_t tx, nt, local, mx;
    tile_counts(width, height, ctx->tile_rank, ctx->tile_world, tx, nt, local, mx);
    if (n_tiles) *n_tiles = local;
    if (max_tiles_any_rank) *max_tiles_any_rank = mx;
    return RR_OK;
}

namespace {

int ensure_cams(rr_context* ctx, size_t n)
{
    if (n <= ctx->cams_cap) return RR_OK;
    RR_HIP(hipStreamSynchronize(ctx->stream));
    dfree(ctx->d_cams);
    ctx->cams_cap = 0;
    size_t cap = n < 64 ? 64 : n;
    RR_HIP(hipMalloc(&ctx->d_cams, cap * sizeof(CamDev)));
    ctx->cams_cap = cap;
    return RR_OK;
}

// DispatchRays(W, H, depth): slice f uses the constants d_cams[f] and writes to out + f*stride.
// ext_tiles != null: compact tile output into caller memory with the given stride (sharded frames).
int ensure_frame_buffers(rr_context* ctx, size_t elems, bool want_f32)
{
    if (elems > ctx->rgba_elems || !ctx->d_rgba8) {
        RR_HIP(hipStreamSynchronize(ctx->stream));
        dfree(ctx->d_rgba8);
        ctx->rgba_elems = 0;
        RR_HIP(hipMalloc(&ctx->d_rgba8, elems * 4));
        ctx->rgba_elems = elems;
    }
    if (want_f32 && (elems > ctx->f32_elems || !ctx->d_f32)) {
        RR_HIP(hipStreamSynchronize(ctx->stream));
        dfree(ctx->d_f32);
        ctx->f32_elems = 0;
        RR_HIP(hipMalloc(&ctx->d_f32, elems * 16));
        ctx->f32_elems = elems;
    }
    return RR_OK;
}

int ensure_lane(rr_context* ctx, uint32_t lane)
{
    if (ctx->lane_stream[lane]) return RR_OK;
    RR_HIP(hipStreamCreateWithFlags(&ctx->lane_stream[lane], hipStreamNonBlocking));
    RR_HIP(hipEventCreateWithFlags(&ctx->lane_fork[lane], hipEventDisableTiming));
    RR_HIP(hipEventCreateWithFlags(&ctx->lane_done[lane], hipEventDisableTiming));
    return RR_OK;
}

// (the screen rectangle of the scene bounds: rr_host_screen_rect, csrc/host/rr_host_partition.cpp)
inline void mesh_screen_rect(const float box[6], const rr_scene_constants* cams, uint32_t n, uint32_t W, uint32_t H, uint32_t r[4])
{
    (void)rr_host_screen_rect(box, cams, n, W, H, r);
}

// ---- k_stream_* : buffers and passes ----------------------------------------------------------------------------------
// One pass renders `fc` consecutive slices of the dispatch.  Worst case per pixel of the ray kernels' blocks: four rays alive in
// one generation (max_reflect <= 2), so a queue holds 4 x pixels entries plus what the waves' 1 024-entry reservations can
// leave unused; slots are 64 B and the mark 1 B per pixel.  A pass is sized to stay inside the budget of its buffer set: a sixth
// of the memory free when the renderer is first used, at most 48 GB -- every kernel of a pass ends in a tail of a few long
// chains, so passes should be few (the 1 024-instance scene at 2160p, Depth 16: 4.04 / 3.68 / 3.49 / 3.39 ms per frame with
// 6 / 12 / 24 / 48 GB, i.e. 2 / 4 / 8 / 16 slices per pass).  The buffers are kept for the life of the context.
constexpr size_t STREAM_BUDGET_MAX = (size_t)48 << 30, STREAM_BUDGET_MIN = (size_t)1 << 30;
constexpr uint32_t STREAM_BLK = 1024;

size_t stream_budget(rr_context* ctx)
{
    if (ctx->strm_budget == 0) {
        size_t free_b = 0, total_b = 0;
        size_t b = STREAM_BUDGET_MAX;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) b = std::min(b, free_b / 6u);
        if (const char* e = getenv("RR_DEBUG_STREAM_BUDGET_GB")) { const long g = atol(e); if (g > 0 && g <= 200) b = (size_t)g << 30; }
        ctx->strm_budget = std::max(b, STREAM_BUDGET_MIN);
    }
    return ctx->strm_budget;
}

// launches of the reference's scene from this many slices on take k_render_lds unless k_render_fused measures faster
inline bool lds_default_depth(uint32_t depth) { return depth >= 24u; }

// the reference's scene (one identity instance) with a node array small enough for LDS beside the traversal stacks
bool scene_fits_lds(const rr_context* ctx)
{
    if (!ctx->single_identity || ctx->dbg_stack != 0 || ctx->inst_host.empty()) return false;
    const MeshRes& m0 = ctx->meshes[(size_t)ctx->inst_host[0].blas];
    const uint32_t node_bytes = (m0.n_tris > 1 ? m0.n_tris - 1 : 1) * (uint32_t)sizeof(QNode);
    return m0.n_tris < 32768u && lds_kernel_shape(node_bytes, scene_stack_need(ctx) + 1, nullptr, ctx->dbg_shape) >= 0;
}

// the buffer set of the stream the dispatch is on
uint32_t stream_slot(const rr_context* ctx)
{
    for (uint32_t l = 0; l < rr_context::MAX_LANES; ++l) if (ctx->lane_stream[l] && ctx->stream == ctx->lane_stream[l]) return l;
    return rr_context::MAX_LANES;
}

struct StreamPlan { uint32_t fc, n_wg; size_t cap, pixels; };

// wave-blocks of `frames` slices that the ray kernels render: the tiles that touch the scene's screen rectangle come first in
// launch order (under the mesh-tile partition: this rank's mesh tiles), in groups of eight tiles
size_t stream_rect_wb(const DispatchDev& a, uint32_t frames)
{
    const size_t all = (size_t)a.blocks_per_frame * frames * 4u;
    if (!a.mesh_part && a.rt_w == 0u) return all;
    const size_t tiles = a.mesh_part ? (size_t)a.n_mesh_local : (size_t)a.rt_w * a.rt_h;
    return std::min(all, ((tiles + 7u) / 8u) * 128u * frames);
}

StreamPlan stream_plan(rr_context* ctx, const DispatchDev& a, uint32_t depth)
{
    auto rect_wb = [&](uint32_t frames) -> size_t { return stream_rect_wb(a, frames); };
    auto bytes = [&](uint32_t frames, StreamPlan& pl) -> size_t {
        const size_t wb = rect_wb(frames);
        pl.pixels = wb * 64u;
        pl.n_wg = (uint32_t)std::min<size_t>((size_t)ctx->n_cus * 6u, std::max<size_t>(1u, (wb + 15u) / 16u));     // six workgroups of the ray kernels fit a CU
        pl.cap = ((4u * pl.pixels + (size_t)pl.n_wg * 4u * STREAM_BLK + STREAM_BLK - 1u) / STREAM_BLK) * STREAM_BLK;
        return 2u * pl.cap * 48u + 2u * (pl.cap / 64u) * 4u + pl.pixels * 65u;
    };
    StreamPlan pl{ 1, 1, 0, 0 };
    uint32_t fc = depth;
    const size_t budget = stream_budget(ctx);
    while (fc > 1u && bytes(fc, pl) > budget) fc = (fc + 1u) / 2u;
    (void)bytes(fc, pl);
    pl.fc = fc;
    return pl;
}

int ensure_stream_buffers(rr_context* ctx, const StreamPlan& pl)
{
    if (pl.cap > 0xffffffffull || pl.pixels > 0xffffffffull) return fail(ctx, RR_ERR_UNSUPPORTED, "stream renderer: pass too large for 32-bit ray indices");
    const uint32_t slot = stream_slot(ctx);
    StreamDev& sd = ctx->strm[slot];
    if (!sd.heads) {      // head counters and, behind them, the chunk ticket counters: one block, zeroed by one memset per pass
        RR_HIP(hipMalloc(&sd.heads, (STREAM_MAX_GEN + STREAM_MAX_GEN * 8u * 16u) * sizeof(uint32_t)));
        sd.next = sd.heads + STREAM_MAX_GEN;
    }
    if (pl.cap > ctx->strm_cap[slot]) {
        RR_HIP(hipStreamSynchronize(ctx->stream));          // (this stream is the set's only user)
        dfree(sd.q[0]); dfree(sd.q[1]); dfree(sd.fill[0]); dfree(sd.fill[1]);
        ctx->strm_cap[slot] = 0;
        for (int k = 0; k < 2; ++k) {
            RR_HIP(hipMalloc(&sd.q[k], pl.cap * 48u));
            RR_HIP(hipMalloc(&sd.fill[k], (pl.cap / 64u) * 4u));
        }
        ctx->strm_cap[slot] = pl.cap;
    }
    if (pl.pixels > ctx->strm_pixels[slot]) {
        RR_HIP(hipStreamSynchronize(ctx->stream));
        dfree(sd.slots); dfree(sd.pending);
        ctx->strm_pixels[slot] = 0;
        RR_HIP(hipMalloc(&sd.slots, pl.pixels * 64u));
        RR_HIP(hipMalloc(&sd.pending, pl.pixels));
        ctx->strm_pixels[slot] = pl.pixels;
    }
    return RR_OK;
}

// the whole dispatch through the generation-per-kernel renderer, `fc` slices per pass
int render_stream(rr_context* ctx, const SceneDev& sc, const DispatchDev& a, uint32_t depth, int need, bool stats, bool rgb8)
{
    const StreamPlan pl = stream_plan(ctx, a, depth);
    if (int r = ensure_stream_buffers(ctx, pl)) return r;
    for (uint32_t f0 = 0; f0 < depth; f0 += pl.fc) {
        const uint32_t fc = std::min(pl.fc, depth - f0);
        DispatchDev b = a;
        // lanes (in sixteenths of the wave's live lanes) a step / a shading pass needs to be issued: measured on the
        // 1 024-instance scene (tools/exp_stream_sweep.sh; RR_DEBUG_ASYNC overrides)
        if (!ctx->dbg_async_set) { b.async_leaf_num = 2u; b.async_shade_num = 8u; }
        b.cams = a.cams + f0;
        b.n_frames = fc;
        b.n_blocks = a.blocks_per_frame * fc;
        b.out_rgba8 = a.out_rgba8 + (size_t)f0 * a.frame_stride;
        if (a.out_f32) b.out_f32 = a.out_f32 + (size_t)f0 * a.frame_stride;
        StreamDev s = ctx->strm[stream_slot(ctx)];
        s.cap = (uint32_t)ctx->strm_cap[stream_slot(ctx)];
        s.n_rect_wb = (uint32_t)stream_rect_wb(a, fc);
        RR_HIP(launch_render_stream(sc, b, s, need, pl.n_wg, stats, ctx->stream, ctx->dbg_stream_waves));
    }
    (void)rgb8;
    return RR_OK;
}

// mesh-tile partition (rr_mesh_partition): where rank 0's background tiles of a dispatch go
struct MeshOut { uint32_t* bg; size_t bg_stride_elems; };

inline bool timed_request(const rr_dispatch_params& p) { return (p.flags & RR_DISPATCH_TIME_KERNEL) != 0; }

// out_slot: which of the frames_in_flight output regions of the internal frame buffer this dispatch writes;
// h_cams: host copy of the depth slices' constants (may be null: no ordering hint)
int dispatch_impl(rr_context* ctx, uint32_t width, uint32_t height, uint32_t depth, const CamDev* d_cams,
                  const rr_scene_constants* h_cams, const rr_dispatch_params& p, uint32_t* ext_tiles, size_t ext_stride_elems,
                  bool keep_counters, uint32_t out_slot = 0, uint32_t out_slot_depth = 0, const MeshOut* mesh = nullptr)
{
    if (width == 0 || height == 0 || width > 32768 || height > 32768 || depth == 0 || depth > 65535)
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "dispatch: bad frame size or depth");
    if (!ctx->tlas_built) return fail(ctx, RR_ERR_STATE, "dispatch: build the BLAS and TLAS first");
    if (p.max_refract < 0 || p.max_refract > 65535 || p.max_reflect < 0)
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "dispatch: negative bounce limit");
    if (p.max_reflect > 8) return fail(ctx, RR_ERR_UNSUPPORTED, "dispatch: max_reflect > 8 (parked-ray registers)");
    if (!(p.ior > 0.0f)) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "dispatch: ior must be > 0");

    uint32_t tiles_x, n_tiles, local, max_local;
    tile_counts(width, height, ctx->tile_rank, ctx->tile_world, tiles_x, n_tiles, local, max_local);
    rr_mesh_partition part;
    memset(&part, 0, sizeof part);
    uint32_t n_mesh_local = 0;
    if (mesh) {         // mesh tiles dealt round robin, background tiles to rank 0: this rank's tiles are its mesh tiles, then those
        if (!ext_tiles || !(p.flags & RR_DISPATCH_TILES_RGB8) || !h_cams) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "mesh partition: RGB8 tile buffers and host constants");
        if (rr_host_mesh_partition(ctx->scene_bounds, (p.flags & RR_DISPATCH_DEBUG_NO_CULL) ? nullptr : h_cams, depth, width, height, ctx->tile_world, &part) != RR_OK)
            return fail(ctx, RR_ERR_INVALID_ARGUMENT, "mesh partition");
        n_mesh_local = rr_host_mesh_tiles_of_rank(&part, ctx->tile_rank);
        if (ctx->tile_rank == 0 && part.n_bg_tiles && !mesh->bg) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "mesh partition: rank 0 needs the background tile buffer");
        local = n_mesh_local + (ctx->tile_rank == 0 ? part.n_bg_tiles : 0);
        max_local = part.max_mesh_tiles_per_rank;
    }
    const bool want_f32 = (p.flags & RR_DISPATCH_FLOAT_OUTPUT) != 0;
    const bool compact = ctx->tile_world > 1 || ext_tiles != nullptr;
    const bool rgb8 = (p.flags & RR_DISPATCH_TILES_RGB8) != 0;
    if (rgb8 && !ext_tiles) return fail(ctx, RR_ERR_UNSUPPORTED, "dispatch: RGB8 tiles only exist in external tile buffers (rr_render_orbit_sharded)");
    // elements are 32-bit words; an RGB8 tile is 3/4 of an RGBA8 tile
    const size_t slice_elems = compact ? (size_t)max_local * TILE * TILE * (rgb8 ? 3 : 4) / 4 : (size_t)width * height;
    const size_t stride = ext_tiles ? ext_stride_elems : slice_elems;
    if (ext_tiles && want_f32) return fail(ctx, RR_ERR_UNSUPPORTED, "dispatch: float output is not available for external tile buffers");
    const size_t out_base = ext_tiles ? 0 : slice_elems * out_slot_depth * out_slot;
    if (!ext_tiles)
        if (int r = ensure_frame_buffers(ctx, out_base + slice_elems * depth, want_f32)) return r;

    if (width != ctx->screen_w || height != ctx->screen_h) {     // new frame size: new tables (nothing in flight may still read the old ones)
        RR_HIP(hipDeviceSynchronize());
        dfree(ctx->d_screen);
        ctx->screen_w = ctx->screen_h = 0;
        RR_HIP(hipMalloc(&ctx->d_screen, ((size_t)width + height) * sizeof(float)));
        RR_HIP(launch_screen_tables(ctx->d_screen, width, height, ctx->stream));
        RR_HIP(hipStreamSynchronize(ctx->stream));
        ctx->screen_w = width; ctx->screen_h = height;
    }
    SceneDev sc;
    fill_scene(ctx, sc);
    DispatchDev a;
    memset(&a, 0, sizeof a);
    a.sx = ctx->d_screen; a.sy = ctx->d_screen + width;
    a.async_leaf_num = ctx->dbg_async[0]; a.async_shade_num = ctx->dbg_async[1];
    a.group_trace = ctx->dbg_group_trace ? 1u : 0u;
    {   // where the scene can be seen at all in these slices
        uint32_t hr[4];
        mesh_screen_rect(ctx->scene_bounds, (p.flags & RR_DISPATCH_DEBUG_NO_CULL) ? nullptr : h_cams, depth, width, height, hr);
        a.hx0 = hr[0]; a.hy0 = hr[1]; a.hx1 = hr[2]; a.hy1 = hr[3];
    }
    if (mesh) {
        a.mesh_part = 1u; a.n_mesh_local = n_mesh_local; a.n_rect_tiles = part.n_mesh_tiles; a.mesh_rounds = part.rank0_rounds;
        a.out_bg = mesh->bg; a.bg_stride = mesh->bg_stride_elems;
        if (part.rect_w) {
            a.rt_x0 = part.rect_x0; a.rt_y0 = part.rect_y0; a.rt_w = part.rect_w; a.rt_h = part.rect_h;
            a.rt_div_w = (uint32_t)(0x100000000ull / a.rt_w) + 1u;
            a.rt_div_o = tiles_x > a.rt_w ? (uint32_t)(0x100000000ull / (tiles_x - a.rt_w)) + 1u : 0u;
        }
    } else if (ctx->dbg_tile_order && ctx->tile_world == 1 && n_tiles < 65536u && a.hx1 > a.hx0 && a.hy1 > a.hy0) {
        // unsharded frames: the tiles that touch the rectangle are rendered first (DispatchDev::rt_*)
        const uint32_t tiles_y = (height + TILE - 1) / TILE;
        const uint32_t x0 = a.hx0 / TILE, y0 = a.hy0 / TILE;
        const uint32_t x1 = std::min(tiles_x, (a.hx1 + TILE - 1) / TILE), y1 = std::min(tiles_y, (a.hy1 + TILE - 1) / TILE);
        if (x1 > x0 && y1 > y0 && (x1 - x0) * (y1 - y0) < n_tiles) {
            a.rt_x0 = x0; a.rt_y0 = y0; a.rt_w = x1 - x0; a.rt_h = y1 - y0;
            a.rt_div_w = (uint32_t)(0x100000000ull / a.rt_w) + 1u;
            a.rt_div_o = tiles_x > a.rt_w ? (uint32_t)(0x100000000ull / (tiles_x - a.rt_w)) + 1u : 0u;
        }
    }
    a.cams = d_cams;
    a.n_frames = depth;
    a.blocks_per_frame = ((local + 7u) & ~7u) * 4u;
    a.frame_stride = stride;
    a.W = width; a.H = height; a.tiles_x = tiles_x; a.n_tiles = n_tiles;
    a.tile_rank = ctx->tile_rank; a.tile_world = ctx->tile_world;
    a.n_local_tiles = local;
    a.n_blocks = a.blocks_per_frame * depth;
    a.compact_out = compact ? (rgb8 ? 2u : 1u) : 0u;
    a.tonemap = (p.flags & RR_DISPATCH_TONEMAP_REINHARD) ? 1u : 0u;
    a.max_refract = p.max_refract; a.max_reflect = p.max_reflect;
    a.ior = p.ior; a.inv_ior = 1.0f / p.ior;
    a.tmin_p = p.tmin_primary; a.tmax_p = p.tmax_primary; a.tmin_s = p.tmin_secondary; a.tmax_s = p.tmax_secondary;
    a.out_rgba8 = ext_tiles ? ext_tiles : ctx->d_rgba8 + out_base;
    a.out_f32 = want_f32 ? ctx->d_f32 + out_base : nullptr;
    a.counters = ctx->d_cnt->counters;
    a.ray_shards = ctx->d_cnt->shards;
    a.error_flag = &ctx->d_cnt->error;
    a.diag = nullptr;
    unsigned long long* d_diag = nullptr;
    const char* diag_path = ctx->dbg_diag.empty() ? nullptr : ctx->dbg_diag.c_str();
    const size_t diag_waves = std::max<size_t>(((size_t)a.n_blocks + (size_t)((a.hx1 - a.hx0) / 8u + 1u) * ((a.hy1 - a.hy0) / 8u + 1u) * depth) * 4 * 2, (size_t)ctx->n_cus * 32);
    if (diag_path && ctx->single_identity) {
        RR_HIP(hipMalloc(&d_diag, diag_waves * 64));
        RR_HIP(hipMemsetAsync(d_diag, 0, diag_waves * 64, ctx->stream));
        a.diag = d_diag;
    }

    const bool stats = (p.flags & RR_DISPATCH_COLLECT_STATS) != 0;
    const uint32_t need = scene_stack_need(ctx);
    const bool keep = keep_counters || (p.flags & RR_DISPATCH_KEEP_COUNTERS) != 0;
    const uint32_t filled = mesh ? n_mesh_local : local;       // slots of the (gathered) tile buffer this rank writes
    if (compact && filled < max_local)           // keep the gathered tail deterministic
        for (uint32_t f = 0; f < depth; ++f)
            RR_HIP(hipMemsetAsync(reinterpret_cast<uint8_t*>(a.out_rgba8 + f * stride) + (size_t)filled * TILE * TILE * (rgb8 ? 3 : 4), 0,
                                  (size_t)(max_local - filled) * TILE * TILE * (rgb8 ? 3 : 4), ctx->stream));
    int stack_sel = need <= 19 ? 19 : need <= 22 ? 22 : need <= 26 ? 26 : need <= 31 ? 31 : need <= 39 ? 39 : 64;     // rr_render.hip: sizes that fill the LDS with 6 / 5 / 4 / 2 workgroups
    if (ctx->dbg_stack >= (int)need) stack_sel = ctx->dbg_stack;   // experiments only; never below the tree depth (the kernels do not check)
    // the reference's scene with a node array small enough for LDS (its own meshes up to shell.obj): persistent workgroups,
    // nodes read from LDS
    const MeshRes* m0 = ctx->single_identity ? &ctx->meshes[(size_t)ctx->inst_host[0].blas] : nullptr;
    const uint32_t node_bytes = m0 ? (m0->n_tris > 1 ? m0->n_tris - 1 : 1) * (uint32_t)sizeof(QNode) : 0;
    // share of the frame in which the scene can be seen at all in these slices
    const double rect_share = (double)(a.hx1 - a.hx0) * (double)(a.hy1 - a.hy0) / ((double)width * (double)height);
    // k_render_lds (persistent workgroups, the BLAS's nodes in LDS) is an alternative for the reference's small meshes that
    // measures within 1-3 % of k_render_fused either way (monkey.obj Depth 64: 5.71 against 5.68 ms per launch); it is kept
    // behind RR_DEBUG_KERNEL=lds, for the parity tests and for experiments, and never chosen by itself.
    const bool lds_fits = scene_fits_lds(ctx) && (uint64_t)a.n_tiles * depth * depth < 0x40000000ull;       // (its ticket arithmetic divides by multiply-high)
    // Launches of one or two slices whose scene is small on screen last as long as their most expensive wave: there the
    // path-parallel kernel (four lanes per pixel inside the scene's screen rectangle: a fifth of the longest chain of
    // dependent rays, four waves per block) wins -- monkey.obj 1080p Depth 1: 268 us against 471, ott.obj 626 against 1 419.
    // It traces the primary ray four times and the count-1 rays twice, so where the mesh fills the frame and the launch is
    // bound by throughput it loses (sphere.obj 483 us against 263, shell.obj 606 against 348): those stay with k_render_fused.
    const bool have_rect = a.hx1 > a.hx0 && a.hy1 > a.hy0;
    uint32_t pool_nodes = 0;
    if (!ctx->single_identity) {
        pool_nodes = ctx->n_insts > 1 ? ctx->n_insts - 1 : 1;
        std::vector<char> seen(ctx->meshes.size(), 0);
        for (uint32_t i = 0; i < ctx->n_insts; ++i) {
            const size_t mi = (size_t)ctx->inst_host[i].blas;
            if (!seen[mi]) { seen[mi] = 1; pool_nodes += ctx->meshes[mi].n_tris > 1 ? ctx->meshes[mi].n_tris - 1 : 1; }
        }
    }
    const bool refill_stack16 = ctx->single_identity ? (m0 && m0->n_tris < 32768u && need > 19)
                                                     : (pool_nodes < 32768u && ctx->n_pool_tris + ctx->n_insts < 32768u);
    // ---- the candidates
    const bool stream_ok = !ctx->single_identity && p.max_reflect <= 2 && p.max_refract <= (int)STREAM_MAX_GEN - 2 && refill_stack16 && need <= 39 &&
                           !a.diag && ctx->dbg_stack == 0 && !ctx->dbg_tlas32;
    const bool paths_ok = !compact && ctx->tile_world == 1 && p.max_reflect <= 2 && need <= 39 && ctx->dbg_stack == 0 && have_rect && depth <= 2;
    auto launch_fused = [&](bool st) -> int {
        // deep trees of small meshes: 16-bit stack entries keep eight waves per SIMD (LDS would otherwise allow 6/5/4)
        bool stack16 = ctx->single_identity && need > 19 && need <= 39 && ctx->dbg_stack == 0 &&
                       ctx->meshes[(size_t)ctx->inst_host[0].blas].n_tris < 32768u;
        // two-level scenes: 16-bit entries wherever every node / leaf reference of the pool fits them (RR_DEBUG_TLAS32=1: never)
        if (!ctx->single_identity && refill_stack16 && (need <= 30 || (need <= 39 && depth > 2)) && p.max_reflect <= 2 && ctx->dbg_stack == 0 && !ctx->dbg_tlas32) stack16 = true;
        // (launches of one or two slices used to take the five-wave build, whose long waves ran faster without the spills of the
        // 6..8-wave builds; since the background branch left those builds with six spilled words the ladder above is the faster one
        // at every depth: sphere.obj Depth 1 250 us against 268, monkey.obj and shell.obj equal)
        if (depth <= 2 && ctx->single_identity) stack16 = false;
        // (the two-level 16-bit-stack builds are sized by the tree itself: 30 entries still leave five workgroups per CU)
        RR_HIP(launch_render_fused(sc, a, !ctx->single_identity && stack16 && ctx->dbg_stack == 0 ? (int)need : stack_sel, p.max_reflect <= 2 ? 2 : 8, st, ctx->stream, stack16));
        return RR_OK;
    };
    // k_render_lds parks reflected rays in a slab per stream slot (allocated at first use: outside anything that is timed)
    auto lds_slot = [&]() -> uint32_t {
        for (uint32_t l = 0; l < rr_context::MAX_LANES; ++l) if (ctx->lane_stream[l] && ctx->stream == ctx->lane_stream[l]) return l;
        return rr_context::MAX_LANES;          // launches on one stream are ordered: one ticket block and one slab per stream
    };
    auto ensure_lds_park = [&](uint32_t slot) -> int {
        const size_t park_need = (size_t)ctx->n_cus * 32 * (p.max_reflect <= 2 ? 2u : 8u) * 8 * 64 * sizeof(uint32_t);     // at most 32 waves per CU
        if (ctx->park_bytes[slot] < park_need) {
            RR_HIP(hipStreamSynchronize(ctx->stream));
            dfree(ctx->d_park[slot]);
            ctx->park_bytes[slot] = 0;
            RR_HIP(hipMalloc(&ctx->d_park[slot], park_need));
            ctx->park_bytes[slot] = park_need;
        }
        return RR_OK;
    };
    auto launch_lds = [&](bool st) -> int {
        LdsDispatch q;
        memset(&q, 0, sizeof q);
        const uint32_t slot = lds_slot();
        q.tickets = ctx->d_tickets + (size_t)slot * LDS_TICKET_WORDS;
        q.park_slots = p.max_reflect <= 2 ? 2u : 8u;
        if (int r = ensure_lds_park(slot)) return r;
        q.park = ctx->d_park[slot];
        uint32_t rect[4];
        mesh_screen_rect(m0->bounds, ((ctx->dbg_ticket_blocks & 3) == 1 || (p.flags & RR_DISPATCH_DEBUG_NO_CULL)) ? nullptr : h_cams, depth, width, height, rect);
        // experiments (RR_DEBUG_TICKET): low bits 1 = whole frame in phase 1, 2 = no phase 1, 3 = phase 1 at every depth;
        // +16: eight queues, a wave starts on its XCD's; +32: parked rays in registers
        const int tk = ctx->dbg_ticket_blocks & 3;
        if (tk == 2) rect[2] = rect[0];
        // eight queues, a wave starts on its XCD's: an XCD then works on every eighth slice, which its L2 rewards
        // (monkey.obj Depth 64: 90 us per frame, 104 with 32 queues entered by wave number)
        q.n_queues = (ctx->dbg_ticket_blocks & 64) ? 64u : (ctx->dbg_ticket_blocks & 128) ? LDS_QUEUES : 8u;   // launch_render_lds caps it at the grid size
        q.home_xcc = (ctx->dbg_ticket_blocks & 16) ? 0u : 1u;
        q.rx0 = rect[0]; q.ry0 = rect[1]; q.rx1 = rect[2]; q.ry1 = rect[3];
        q.node_bytes = node_bytes;
        q.stack_entries = need + 1;                     // the tree's depth bounds the stack; one entry to spare
        RR_HIP(launch_render_lds(sc, a, q, ctx->n_cus, st, ctx->stream, ctx->dbg_shape));
        return RR_OK;
    };
    auto launch_paths = [&](bool st) -> int { RR_HIP(launch_render_paths(sc, a, (int)need, st, ctx->stream)); return RR_OK; };
    auto launch_stream = [&](bool st) -> int { return render_stream(ctx, sc, a, depth, (int)need, st, rgb8); };

    // ---- which kernel: forced by RR_DEBUG_KERNEL, or the class's measured choice
    enum { K_FUSED = 0, K_LDS = 1, K_PATHS = 2, K_STREAM = 7 };
    int kernel = K_FUSED;
    rr_context::ChoiceClass* cls = nullptr;             // the class this launch belongs to, if it has two candidates
    int cand_a = K_FUSED, cand_b = K_FUSED;          // the default and the alternative of the launch's class
    if (ctx->dbg_kernel == 10) { if (stream_ok) kernel = K_STREAM; }
    else if (ctx->dbg_kernel == 5) { if (paths_ok) kernel = K_PATHS; }
    else if (ctx->dbg_kernel == 4) { if (lds_fits && !mesh && !compact) kernel = K_LDS; }      // (k_render_lds renders unsharded dispatches only)
    else if (ctx->dbg_kernel == 0 && !a.diag) {
        if (stream_ok) { cls = &ctx->ch_tlas; cand_b = K_STREAM; }
        else if (paths_ok) { cls = &ctx->ch_few; cand_b = K_PATHS; }
        else if (lds_fits && depth >= 3 && !compact && !mesh) {
            // the persistent kernel pays off from about twenty slices a launch (monkey.obj: 1.43 against 1.38 ms at Depth 16,
            // 1.67 / 1.74 at 20, 2.53 / 2.77 at 32, 4.80 / 5.45 at 64; tools/exp_lds_depths.py): from 24 on it is the default, the
            // L1-fed one the alternative
            cls = &ctx->ch_many;
            if (lds_default_depth(depth)) { cand_a = K_LDS; cand_b = K_FUSED; } else cand_b = K_LDS;
        }
    }
    if (a.diag && paths_ok && rect_share < 0.25 && ctx->dbg_kernel == 0) kernel = K_PATHS;      // (the diagnostic builds keep round 2's rule)
    if (cls) {
        // (k_render_lds gains on k_render_fused with the launch depth: sphere.obj 160 / 160 us per frame at Depth 16, 146 / 157 at 64)
        const unsigned long long key = rr_context::choice_key(width, height, p, depth);
        rr_context::KernelChoice* const ch = cls->find(key);
        if (ch->choice != 0 && (rect_share > 2.0 * ch->share || rect_share * 2.0 < ch->share)) { ch->choice = 0; ch->seen = 0; ch->ms[0] = ch->ms[1] = 0.0f; }
        if (ch->choice == 0 && !(p.flags & RR_DISPATCH_DEBUG_NO_CULL)) {
            if (ch->seen++ >= 1u) {
                // the measurement: both candidates render this dispatch (product builds), one after the other, counting into a
                // block of their own (the dispatch's counters are the caller's: a batch of a sharded pipeline keeps adding to them);
                // launches still running on other lanes would be timed along, so they are waited for first
                if (cand_b == K_STREAM) if (int r = ensure_stream_buffers(ctx, stream_plan(ctx, a, depth))) return r;
                if (cand_a == K_LDS || cand_b == K_LDS) if (int r = ensure_lds_park(lds_slot())) return r;
                for (int k = 0; k < 4; ++k) if (!ctx->ch_ev[k]) RR_HIP(hipEventCreate(&ctx->ch_ev[k]));
                if (!ctx->d_cnt_trial) RR_HIP(hipMalloc(&ctx->d_cnt_trial, sizeof(CounterBlock)));
                RR_HIP(hipDeviceSynchronize());
                RR_HIP(hipMemsetAsync(ctx->d_cnt_trial, 0, sizeof(CounterBlock), ctx->stream));
                struct Restore { DispatchDev& a; unsigned long long* c; uint32_t* s; uint32_t* e; ~Restore() { a.counters = c; a.ray_shards = s; a.error_flag = e; } }
                    restore{ a, a.counters, a.ray_shards, a.error_flag };
                a.counters = ctx->d_cnt_trial->counters; a.ray_shards = ctx->d_cnt_trial->shards; a.error_flag = &ctx->d_cnt_trial->error;
                // A (untimed: the device was just idle, its first launch would pay for the clocks coming back), then A and B timed
                auto launch_k = [&](int k) -> int { return k == K_STREAM ? launch_stream(false) : k == K_PATHS ? launch_paths(false) : k == K_LDS ? launch_lds(false) : launch_fused(false); };
                if (int r = launch_k(cand_a)) return r;
                for (int c = 0; c < 2; ++c) {
                    RR_HIP(hipEventRecord(ctx->ch_ev[2 * c], ctx->stream));
                    if (int r = launch_k(c == 0 ? cand_a : cand_b)) return r;
                    RR_HIP(hipEventRecord(ctx->ch_ev[2 * c + 1], ctx->stream));
                }
                RR_HIP(hipEventSynchronize(ctx->ch_ev[3]));
                float ms_a = 0.0f, ms_b = 0.0f;
                RR_HIP(hipEventElapsedTime(&ms_a, ctx->ch_ev[0], ctx->ch_ev[1]));
                RR_HIP(hipEventElapsedTime(&ms_b, ctx->ch_ev[2], ctx->ch_ev[3]));
                // Two measurements, on consecutive dispatches of the shape, decide together: the alternative renders the shape from
                // then on if it took less than 98 % of the default's time over both (a launch repeats within 1-2 %, and how far apart
                // two kernels are depends on the view: k_render_lds against k_render_fused on monkey.obj at Depth 64 is 2 to 10 %
                // faster launch by launch round the orbit, 6 % over it; tools/exp_lds_vs_fused.py).  An alternative that is clearly
                // slower the first time is not measured again.
                const bool first = !(ch->ms[0] > 0.0f);
                const float sum_a = ch->ms[0] + ms_a, sum_b = ch->ms[1] + ms_b;
                if (first && ms_b >= 1.02f * ms_a) ch->choice = 1;
                else if (!first) ch->choice = sum_b < 0.98f * sum_a ? 2 : 1;
                ch->ms[0] = sum_a; ch->ms[1] = sum_b;
                if (getenv("RR_DEBUG_CHOICE"))
                    fprintf(stderr, "[rr] kernel choice: depth %u candidate %d: default %.3f ms, candidate %.3f ms (%.3f)%s\n", depth, cand_b, ms_a, ms_b,
                            ms_b / ms_a, ch->choice == 2 ? " -> candidate" : ch->choice == 1 ? " -> default" : " (once more)");
                ch->share = rect_share;
            }
        }
        kernel = ch->choice == 2 ? cand_b : cand_a;
        // (until the measurement: the round-2 rule for launches of one or two slices -- the path-parallel kernel where the scene is small on screen)
        if (ch->choice == 0 && cand_b == K_PATHS && rect_share < 0.25) kernel = K_PATHS;
    }
    const bool stream_kernel = kernel == K_STREAM, paths_kernel = kernel == K_PATHS, lds_kernel = kernel == K_LDS;

    if (!keep) RR_HIP(hipMemsetAsync(ctx->d_cnt, 0, sizeof(CounterBlock), ctx->stream));
    const bool timed = (p.flags & RR_DISPATCH_TIME_KERNEL) != 0;
    if (timed) {
        if (ctx->kev_used >= 4096) return fail(ctx, RR_ERR_STATE, "dispatch: 4096 timed dispatches pending, call rr_kernel_time");
        while (ctx->kev.size() < (size_t)(ctx->kev_used + 1) * 2) {
            hipEvent_t e;
            RR_HIP(hipEventCreate(&e));
            ctx->kev.push_back(e);
        }
        RR_HIP(hipEventRecord(ctx->kev[(size_t)ctx->kev_used * 2], ctx->stream));
    }
    if (stream_kernel) { if (int r = launch_stream(stats)) return r; }
    else if (paths_kernel) { if (int r = launch_paths(stats)) return r; }
    else if (lds_kernel) { if (int r = launch_lds(stats)) return r; }
    else { if (int r = launch_fused(stats)) return r; }
    if (timed) {
        RR_HIP(hipEventRecord(ctx->kev[(size_t)ctx->kev_used * 2 + 1], ctx->stream));
        ++ctx->kev_used;
    }
    if (d_diag) {       // experiments only: dump per-wave {start, cycles, max rays per lane, loop trips}
        std::vector<unsigned long long> h(diag_waves * 8);
        RR_HIP(hipStreamSynchronize(ctx->stream));
        RR_HIP(hipMemcpy(h.data(), d_diag, h.size() * 8, hipMemcpyDeviceToHost));
        (void)hipFree(d_diag);
        if (FILE* f = fopen(diag_path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
    snprintf(ctx->last_kernel_name, sizeof ctx->last_kernel_name, "%s", stream_kernel ? last_stream_kernel_name() : last_render_kernel_name());
    ctx->last_kernel = stream_kernel ? 7u : paths_kernel ? 2u : lds_kernel ? 1u : 0u;
    ctx->W = width; ctx->H = height; ctx->frame_world = ctx->tile_world; ctx->frame_depth = depth;
    ctx->have_f32 = want_f32; ctx->have_frame = ext_tiles == nullptr; ctx->have_assembled = false;
    if (!ext_tiles) ctx->frame_base = out_base;
    ctx->last_stats = stats;
    // pixels actually owned by this rank (partial edge tiles counted exactly)
    uint64_t px = 0;
    auto tile_px = [&](uint32_t t) {
        const uint32_t x0 = (t % tiles_x) * TILE, y0 = (t / tiles_x) * TILE;
        const uint32_t w = width - x0 < TILE ? width - x0 : TILE, h = height - y0 < TILE ? height - y0 : TILE;
        return (uint64_t)w * h;
    };
    if (!mesh) for (uint32_t t = ctx->tile_rank; t < n_tiles; t += ctx->tile_world) px += tile_px(t);
    else
        for (uint32_t t = 0; t < n_tiles; ++t) {
            const uint32_t tx = t % tiles_x, ty = t / tiles_x;
            const bool in_rect = part.rect_w == 0 || (tx >= part.rect_x0 && tx < part.rect_x0 + part.rect_w && ty >= part.rect_y0 && ty < part.rect_y0 + part.rect_h);
            const uint32_t i = part.rect_w == 0 ? t : (ty - part.rect_y0) * part.rect_w + (tx - part.rect_x0);
            uint32_t owner = 0, slot = 0;
            if (in_rect) (void)rr_host_mesh_tile_home(&part, i, &owner, &slot);
            if (in_rect ? owner == ctx->tile_rank : ctx->tile_rank == 0) px += tile_px(t);
        }
    px *= depth;
    ctx->last_pixels = px;
    ctx->accum_pixels = (keep ? ctx->accum_pixels : 0) + px;
    return RR_OK;
}

int upload_cams(rr_context* ctx, const rr_scene_constants* c, size_t n)
{
    static_assert(sizeof(CamDev) == sizeof(rr_scene_constants), "constant buffer layout");
    if (int r = ensure_cams(ctx, n)) return r;
    const int slot = (int)(ctx->h_cams_next++ % rr_context::CAM_SLOTS);
    if (ctx->h_cams_busy[slot]) { RR_HIP(hipEventSynchronize(ctx->h_cams_ev[slot])); ctx->h_cams_busy[slot] = false; }
    if (ctx->h_cams_cap[slot] < n) {
        if (ctx->h_cams[slot]) (void)hipHostFree(ctx->h_cams[slot]);
        ctx->h_cams[slot] = nullptr; ctx->h_cams_cap[slot] = 0;
        const size_t cap = n < 64 ? 64 : n;
        RR_HIP(hipHostMalloc(&ctx->h_cams[slot], cap * sizeof(CamDev), hipHostMallocDefault));
        ctx->h_cams_cap[slot] = cap;
    }
    if (!ctx->h_cams_ev[slot]) RR_HIP(hipEventCreateWithFlags(&ctx->h_cams_ev[slot], hipEventDisableTiming));
    memcpy(ctx->h_cams[slot], c, n * sizeof(CamDev));
    RR_HIP(hipMemcpyAsync(ctx->d_cams, ctx->h_cams[slot], n * sizeof(CamDev), hipMemcpyHostToDevice, ctx->stream));   // copy_to_buffer, :566
    RR_HIP(hipEventRecord(ctx->h_cams_ev[slot], ctx->stream));
    ctx->h_cams_busy[slot] = true;
    return RR_OK;
}

} // namespace

int rr_dispatch_rays(rr_context* ctx, uint32_t width, uint32_t height, const rr_dispatch_params* params)
{
    const Range range_("rr_dispatch_rays");
    if (int r = use_device(ctx)) return r;
    if (!ctx->cam_set) return fail(ctx, RR_ERR_STATE, "rr_dispatch_rays: rr_set_camera first");
    rr_dispatch_params p;
    if (params) p = *params; else rr_default_dispatch_params(&p);
    if (int r = upload_cams(ctx, &ctx->cam, 1)) return r;
    return dispatch_impl(ctx, width, height, 1, ctx->d_cams, &ctx->cam, p, nullptr, 0, false);
}

int rr_dispatch_rays_batch(rr_context* ctx, uint32_t width, uint32_t height, uint32_t depth,
                           const rr_scene_constants* constants, const rr_dispatch_params* params)
{
    if (int r = use_device(ctx)) return r;
    if (!constants || depth == 0) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_dispatch_rays_batch: need depth >= 1 constants");
    rr_dispatch_params p;
    if (params) p = *params; else rr_default_dispatch_params(&p);
    if (int r = upload_cams(ctx, constants, depth)) return r;
    return dispatch_impl(ctx, width, height, depth, ctx->d_cams, constants, p, nullptr, 0, false);
}

int rr_read_frame_slice(rr_context* ctx, uint32_t slice, uint8_t* rgba8, float* rgba32f)
{
    if (int r = use_device(ctx)) return r;
    if (!ctx->have_frame) return fail(ctx, RR_ERR_STATE, "rr_read_frame: nothing dispatched");
    const size_t n = (size_t)ctx->W * ctx->H;
    if (ctx->have_assembled) {
        if (rgba32f || slice) return fail(ctx, RR_ERR_STATE, "rr_read_frame: only slice 0 / RGBA8 of an assembled frame");
        if (rgba8) RR_HIP(hipMemcpyAsync(rgba8, ctx->d_assembled, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    } else {
        if (ctx->frame_world != 1) return fail(ctx, RR_ERR_STATE, "rr_read_frame: sharded frame, gather + rr_assemble_tiles first");
        if (slice >= ctx->frame_depth) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_read_frame: slice beyond the dispatch depth");
        if (rgba32f && !ctx->have_f32) return fail(ctx, RR_ERR_STATE, "rr_read_frame: dispatch with RR_DISPATCH_FLOAT_OUTPUT");
        if (rgba8) RR_HIP(hipMemcpyAsync(rgba8, ctx->d_rgba8 + ctx->frame_base + slice * n, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (rgba32f) RR_HIP(hipMemcpyAsync(rgba32f, ctx->d_f32 + ctx->frame_base + slice * n, n * 16, hipMemcpyDeviceToHost, ctx->stream));
    }
    RR_HIP(hipStreamSynchronize(ctx->stream));
    uint32_t err = 0;
    RR_HIP(hipMemcpy(&err, &ctx->d_cnt->error, 4, hipMemcpyDeviceToHost));
    if (err) return fail(ctx, RR_ERR_TRAVERSAL_OVERFLOW, "traversal stack overflow: frame invalid");
    return RR_OK;
}

int rr_read_frame(rr_context* ctx, uint8_t* rgba8, float* rgba32f) { return rr_read_frame_slice(ctx, 0, rgba8, rgba32f); }

int rr_export_tiles(rr_context* ctx, void* d_dst)
{
    if (int r = use_device(ctx)) return r;
    if (!d_dst) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_export_tiles: null destination");
    if (!ctx->have_frame || ctx->frame_world < 2) return fail(ctx, RR_ERR_STATE, "rr_export_tiles: no sharded frame");
    uint32_t tx, nt, local, mx;
    tile_counts(ctx->W, ctx->H, ctx->tile_rank, ctx->frame_world, tx, nt, local, mx);
    RR_HIP(hipMemcpyAsync(d_dst, ctx->d_rgba8 + ctx->frame_base, (size_t)mx * TILE * TILE * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return RR_OK;
}

int rr_assemble_tiles(rr_context* ctx, const void* d_gathered, uint32_t world, void* d_frame)
{
    if (int r = use_device(ctx)) return r;
    if (!d_gathered || world == 0) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_assemble_tiles: bad arguments");
    if (!ctx->have_frame || ctx->W == 0) return fail(ctx, RR_ERR_STATE, "rr_assemble_tiles: dispatch first (frame size)");
    uint32_t tx, nt, local, mx;
    tile_counts(ctx->W, ctx->H, 0, world, tx, nt, local, mx);
    uint32_t* dst = (uint32_t*)d_frame;
    if (!dst) {
        const size_t n = (size_t)ctx->W * ctx->H;
        if (n > ctx->assembled_elems) {
            RR_HIP(hipStreamSynchronize(ctx->stream));
            dfree(ctx->d_assembled);
            ctx->assembled_elems = 0;
            RR_HIP(hipMalloc(&ctx->d_assembled, n * 4));
            ctx->assembled_elems = n;
        }
        dst = ctx->d_assembled;
    }
    RR_HIP(launch_assemble_tiles((const uint32_t*)d_gathered, dst, ctx->W, ctx->H, tx, nt, world, mx, ctx->stream));
    if (!d_frame) ctx->have_assembled = true;
    return RR_OK;
}

namespace {

// drawFrame loop: camera constants for n_frames consecutive orbit angles go to the device constant
// buffer in one copy; the frames are then dispatched in batches of `batch` depth slices.
int orbit_impl(rr_context* ctx, uint32_t width, uint32_t height, const rr_dispatch_params* params, float* angle,
               float angle_step, uint32_t n_frames, uint32_t batch, float fov_y, float aspect, float zn, float zf,
               uint32_t* ext_tiles, size_t ext_stride_elems, uint8_t* host_out = nullptr)
{
    const Range range_("rr_render_orbit");
    if (!angle) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "render_orbit: null angle");
    if (n_frames == 0) return RR_OK;
    if (batch == 0) batch = 1;
    rr_dispatch_params p;
    if (params) p = *params; else rr_default_dispatch_params(&p);
    std::vector<rr_scene_constants> cams(n_frames);
    for (uint32_t k = 0; k < n_frames; ++k) {
        int rc = rr_host_camera_orbit(*angle, fov_y, aspect, zn, zf, &cams[k]);      // RefractionDemo.cpp:559-565
        if (rc != RR_OK) return fail(ctx, rc, "render_orbit: camera");
        *angle += angle_step;                                                       // :567
    }
    ctx->cam = cams.back(); ctx->cam_set = true;
    if (int r = upload_cams(ctx, cams.data(), n_frames)) return r;
    const bool keep_first = (p.flags & RR_DISPATCH_KEEP_COUNTERS) != 0;
    const uint32_t n_batches = (n_frames + batch - 1) / batch;
    // frames in flight: consecutive launches go to alternating lanes so that the long-running waves at the end of
    // one overlap the start of the next.  Not for timed dispatches (their durations must be exclusive).
    uint32_t lanes = ctx->frames_in_flight < n_batches ? ctx->frames_in_flight : n_batches;
    if ((p.flags & RR_DISPATCH_TIME_KERNEL) || !ctx->dbg_diag.empty()) lanes = 1;
    // k_render_lds is persistent -- its workgroups hold every CU until the launch is over --, so two of its launches in flight only
    // get in each other's way (sphere.obj Depth 64: 145 us per frame one at a time, 167 with two in flight)
    bool one_kernel_at_a_time = false;
    if (scene_fits_lds(ctx) && ctx->tile_world == 1) {
        const uint32_t d = batch < n_frames ? batch : n_frames;
        const rr_context::KernelChoice* c = ctx->ch_many.peek(rr_context::choice_key(width, height, p, d));
        const bool alt = c && c->choice == 2;
        const bool lds_renders = ctx->dbg_kernel == 0 ? (lds_default_depth(d) ? !alt : alt) : ctx->dbg_kernel == 4;
        one_kernel_at_a_time = d >= 3u && lds_renders;
        if (lanes > 1 && one_kernel_at_a_time) lanes = 1;
    }
    if (host_out) {          // streaming to host: the copy of one region overlaps the rendering of the other
        if (ext_tiles || ctx->tile_world != 1 || (p.flags & RR_DISPATCH_FLOAT_OUTPUT))
            return fail(ctx, RR_ERR_UNSUPPORTED, "render_orbit_to_host: whole RGBA8 frames of an unsharded context only");
        lanes = ctx->frames_in_flight > 2 ? ctx->frames_in_flight : 2;
    }
    if (lanes <= 1) {
        for (uint32_t k = 0; k < n_frames; k += batch) {
            const uint32_t d = n_frames - k < batch ? n_frames - k : batch;
            uint32_t* ext = ext_tiles ? ext_tiles + (size_t)k * ext_stride_elems : nullptr;
            if (int rc = dispatch_impl(ctx, width, height, d, ctx->d_cams + k, cams.data() + k, p, ext, ext_stride_elems, k > 0 || keep_first)) return rc;
        }
        return RR_OK;
    }
    if (!ext_tiles) {        // all output regions exist before anything overlaps
        uint32_t tiles_x, n_tiles, local, max_local;
        if (width == 0 || height == 0 || width > 32768 || height > 32768) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "dispatch: bad frame size or depth");
        tile_counts(width, height, ctx->tile_rank, ctx->tile_world, tiles_x, n_tiles, local, max_local);
        const size_t slice_elems = ctx->tile_world > 1 ? (size_t)max_local * TILE * TILE : (size_t)width * height;
        if (int r = ensure_frame_buffers(ctx, slice_elems * batch * lanes, (p.flags & RR_DISPATCH_FLOAT_OUTPUT) != 0)) return r;
    }
    if (!keep_first) {
        RR_HIP(hipMemsetAsync(ctx->d_cnt, 0, sizeof(CounterBlock), ctx->stream));
        ctx->accum_pixels = 0;
    }
    for (uint32_t l = 0; l < lanes; ++l) {
        if (int r = ensure_lane(ctx, l)) return r;
        if (ctx->lane_busy[l]) { RR_HIP(hipStreamWaitEvent(ctx->stream, ctx->lane_done[l], 0)); ctx->lane_busy[l] = false; }
    }
    RR_HIP(hipEventRecord(ctx->lane_fork[0], ctx->stream));          // after the constants upload and the counter reset
    for (uint32_t l = 0; l < lanes; ++l) RR_HIP(hipStreamWaitEvent(ctx->lane_stream[l], ctx->lane_fork[0], 0));
    hipStream_t main_stream = ctx->stream;
    int rc = RR_OK;
    for (uint32_t k = 0, b = 0; k < n_frames && rc == RR_OK; k += batch, ++b) {
        const uint32_t d = n_frames - k < batch ? n_frames - k : batch;
        uint32_t* ext = ext_tiles ? ext_tiles + (size_t)k * ext_stride_elems : nullptr;
        ctx->stream = ctx->lane_stream[b % lanes];
        // (streaming to host keeps two regions for the copies' sake; the persistent kernel's launches still go one after the other)
        if (one_kernel_at_a_time && b > 0 && hipStreamWaitEvent(ctx->stream, ctx->lane_fork[(b - 1) % lanes], 0) != hipSuccess)
            rc = fail(ctx, RR_ERR_DEVICE, "render_orbit: lane order");
        if (rc == RR_OK) rc = dispatch_impl(ctx, width, height, d, ctx->d_cams + k, cams.data() + k, p, ext, ext_stride_elems, true, b % lanes, batch);
        if (rc == RR_OK && one_kernel_at_a_time && hipEventRecord(ctx->lane_fork[b % lanes], ctx->stream) != hipSuccess)
            rc = fail(ctx, RR_ERR_DEVICE, "render_orbit: lane order");
        if (rc == RR_OK && host_out) {      // same lane: the region is not rendered into again before this copy is done
            const size_t fb = (size_t)width * height * 4;
            hipError_t e = hipMemcpyAsync(host_out + (size_t)k * fb, ctx->d_rgba8 + ctx->frame_base, (size_t)d * fb, hipMemcpyDeviceToHost, ctx->stream);
            if (e != hipSuccess) rc = fail(ctx, RR_ERR_DEVICE, "render_orbit_to_host: copy", e);
        }
        ctx->stream = main_stream;
    }
    for (uint32_t l = 0; l < lanes; ++l) {                          // join: the caller's stream is ordered after every lane
        hipError_t e = hipEventRecord(ctx->lane_done[l], ctx->lane_stream[l]);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->lane_done[l], 0);
        if (e != hipSuccess && rc == RR_OK) rc = fail(ctx, RR_ERR_DEVICE, "render_orbit: lane join", e);
    }
    return rc;
}

} // namespace

int rr_render_orbit(rr_context* ctx, uint32_t width, uint32_t height, const rr_dispatch_params* params, float* angle,
                    float angle_step, uint32_t n_frames, uint32_t frames_per_dispatch, float fov_y, float aspect, float zn,
                    float zf)
{
    if (int r = use_device(ctx)) return r;
    return orbit_impl(ctx, width, height, params, angle, angle_step, n_frames, frames_per_dispatch, fov_y, aspect, zn, zf,
                      nullptr, 0);
}

int rr_render_orbit_to_host(rr_context* ctx, uint32_t width, uint32_t height, const rr_dispatch_params* params, float* angle,
                            float angle_step, uint32_t n_frames, uint32_t frames_per_dispatch, float fov_y, float aspect, float zn,
                            float zf, uint8_t* host_rgba8)
{
    if (int r = use_device(ctx)) return r;
    if (!host_rgba8) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_render_orbit_to_host: null host buffer");
    if (int r = orbit_impl(ctx, width, height, params, angle, angle_step, n_frames, frames_per_dispatch, fov_y, aspect, zn, zf,
                           nullptr, 0, host_rgba8)) return r;
    RR_HIP(hipStreamSynchronize(ctx->stream));          // every frame is in host memory on return
    uint32_t err = 0;
    RR_HIP(hipMemcpy(&err, &ctx->d_cnt->error, 4, hipMemcpyDeviceToHost));
    if (err) return fail(ctx, RR_ERR_TRAVERSAL_OVERFLOW, "device error flag set: frames invalid");
    return RR_OK;
}

int rr_render_orbit_sharded(rr_context* ctx, uint32_t width, uint32_t height, const rr_dispatch_params* params, float* angle,
                            float angle_step, uint32_t n_frames, uint32_t frames_per_dispatch, float fov_y, float aspect,
                            float zn, float zf, void* d_tiles, uint64_t frame_stride_bytes)
{
    if (int r = use_device(ctx)) return r;
    if (!d_tiles) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_render_orbit_sharded: null tile buffer");
    uint32_t tx, nt, local, mx;
    tile_counts(width ? width : 1, height ? height : 1, ctx->tile_rank, ctx->tile_world, tx, nt, local, mx);
    const uint64_t bpp = (params && (params->flags & RR_DISPATCH_TILES_RGB8)) ? 3 : 4;
    if (frame_stride_bytes < (uint64_t)mx * TILE * TILE * bpp || (frame_stride_bytes & 3u))
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_render_orbit_sharded: frame stride smaller than a tile buffer");
    return orbit_impl(ctx, width, height, params, angle, angle_step, n_frames, frames_per_dispatch, fov_y, aspect, zn, zf,
                      (uint32_t*)d_tiles, (size_t)(frame_stride_bytes / 4));
}

int rr_render_orbit_sharded_lane(rr_context* ctx, uint32_t width, uint32_t height, const rr_dispatch_params* params, float* angle,
                                 float angle_step, uint32_t n_frames, uint32_t frames_per_dispatch, float fov_y, float aspect,
                                 float zn, float zf, void* d_tiles, uint64_t frame_stride_bytes, uint32_t lane)
{
    if (int r = use_device(ctx)) return r;
    if (lane >= rr_context::MAX_LANES) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_render_orbit_sharded_lane: lane out of range");
    if (int r = ensure_lane(ctx, lane)) return r;
    rr_dispatch_params p;
    if (params) p = *params; else rr_default_dispatch_params(&p);
    if (!(p.flags & RR_DISPATCH_KEEP_COUNTERS)) {      // zero the counters where every lane will see it: before the fork
        RR_HIP(hipMemsetAsync(ctx->d_cnt, 0, sizeof(CounterBlock), ctx->stream));
        ctx->accum_pixels = 0;
        p.flags |= RR_DISPATCH_KEEP_COUNTERS;
    }
    // fork: the lane starts after everything submitted to the context's stream so far
    RR_HIP(hipEventRecord(ctx->lane_fork[lane], ctx->stream));
    RR_HIP(hipStreamWaitEvent(ctx->lane_stream[lane], ctx->lane_fork[lane], 0));
    hipStream_t main_stream = ctx->stream;
    CamDev* main_cams = ctx->d_cams;
    size_t main_cap = ctx->cams_cap;
    const uint32_t main_in_flight = ctx->frames_in_flight;
    ctx->stream = ctx->lane_stream[lane];               // the lane has its own constant buffer: no reuse race between lanes
    ctx->d_cams = ctx->lane_cams[lane];
    ctx->cams_cap = ctx->lane_cams_cap[lane];
    ctx->frames_in_flight = 1;                          // a lane is one stream: its launches stay in order
    int rc = rr_render_orbit_sharded(ctx, width, height, &p, angle, angle_step, n_frames, frames_per_dispatch, fov_y, aspect, zn,
                                     zf, d_tiles, frame_stride_bytes);
    hipError_t e = rc == RR_OK ? hipEventRecord(ctx->lane_done[lane], ctx->stream) : hipSuccess;
    ctx->lane_cams[lane] = ctx->d_cams;
    ctx->lane_cams_cap[lane] = ctx->cams_cap;
    ctx->stream = main_stream;
    ctx->d_cams = main_cams;
    ctx->cams_cap = main_cap;
    ctx->frames_in_flight = main_in_flight;
    if (rc != RR_OK) return rc;
    if (e != hipSuccess) return fail(ctx, RR_ERR_DEVICE, "rr_render_orbit_sharded_lane: event", e);
    ctx->lane_busy[lane] = true;
    return RR_OK;
}

int rr_mesh_partition_for_orbit(rr_context* ctx, uint32_t width, uint32_t height, float angle, float angle_step, uint32_t n_frames,
                                float fov_y, float aspect, float zn, float zf, rr_mesh_partition* out)
{
    if (!ctx || !out || n_frames == 0) return RR_ERR_INVALID_ARGUMENT;
    if (!ctx->tlas_built) return fail(ctx, RR_ERR_STATE, "rr_mesh_partition_for_orbit: build the BLAS and TLAS first");
    std::vector<rr_scene_constants> cams(n_frames);
    for (uint32_t k = 0; k < n_frames; ++k) {
        const int rc = rr_host_camera_orbit(angle, fov_y, aspect, zn, zf, &cams[k]);
        if (rc != RR_OK) return fail(ctx, rc, "rr_mesh_partition_for_orbit: camera");
        angle += angle_step;
    }
    return rr_host_mesh_partition(ctx->scene_bounds, cams.data(), n_frames, width, height, ctx->tile_world, out);
}

int rr_render_orbit_mesh_sharded_lane(rr_context* ctx, uint32_t width, uint32_t height, const rr_dispatch_params* params, float* angle,
                                      float angle_step, uint32_t n_frames, float fov_y, float aspect, float zn, float zf, void* d_mesh_tiles,
                                      uint64_t mesh_stride_bytes, void* d_bg_tiles, uint64_t bg_stride_bytes, uint32_t lane)
{
    const Range range_("rr_render_orbit_mesh_sharded");
    if (int r = use_device(ctx)) return r;
    if (lane >= rr_context::MAX_LANES) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_render_orbit_mesh_sharded_lane: lane out of range");
    if (!angle || !d_mesh_tiles || n_frames == 0 || (mesh_stride_bytes & 3u) || (bg_stride_bytes & 3u))
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_render_orbit_mesh_sharded_lane: bad arguments");
    if (int r = ensure_lane(ctx, lane)) return r;
    rr_dispatch_params p;
    if (params) p = *params; else rr_default_dispatch_params(&p);
    p.flags |= RR_DISPATCH_TILES_RGB8;
    std::vector<rr_scene_constants> cams(n_frames);
    for (uint32_t k = 0; k < n_frames; ++k) {
        const int rc = rr_host_camera_orbit(*angle, fov_y, aspect, zn, zf, &cams[k]);
        if (rc != RR_OK) return fail(ctx, rc, "render_orbit: camera");
        *angle += angle_step;
    }
    rr_mesh_partition part;
    if (rr_host_mesh_partition(ctx->scene_bounds, cams.data(), n_frames, width, height, ctx->tile_world, &part) != RR_OK)
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "mesh partition");
    if (mesh_stride_bytes < (uint64_t)part.max_mesh_tiles_per_rank * TILE * TILE * 3 ||
        (ctx->tile_rank == 0 && part.n_bg_tiles && (!d_bg_tiles || bg_stride_bytes < (uint64_t)part.n_bg_tiles * TILE * TILE * 3)))
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_render_orbit_mesh_sharded_lane: tile buffers smaller than rr_mesh_partition_for_orbit says");
    const bool keep = (p.flags & RR_DISPATCH_KEEP_COUNTERS) != 0;
    if (!keep) {                                        // zero the counters where every lane will see it: before the fork
        RR_HIP(hipMemsetAsync(ctx->d_cnt, 0, sizeof(CounterBlock), ctx->stream));
        ctx->accum_pixels = 0;
        p.flags |= RR_DISPATCH_KEEP_COUNTERS;
    }
    ctx->cam = cams.back(); ctx->cam_set = true;
    RR_HIP(hipEventRecord(ctx->lane_fork[lane], ctx->stream));
    RR_HIP(hipStreamWaitEvent(ctx->lane_stream[lane], ctx->lane_fork[lane], 0));
    hipStream_t main_stream = ctx->stream;
    CamDev* main_cams = ctx->d_cams;
    size_t main_cap = ctx->cams_cap;
    ctx->stream = ctx->lane_stream[lane];               // the lane has its own constant buffer: no reuse race between lanes
    ctx->d_cams = ctx->lane_cams[lane];
    ctx->cams_cap = ctx->lane_cams_cap[lane];
    int rc = upload_cams(ctx, cams.data(), n_frames);
    const MeshOut mo = { (uint32_t*)d_bg_tiles, (size_t)(bg_stride_bytes / 4) };
    if (rc == RR_OK) rc = dispatch_impl(ctx, width, height, n_frames, ctx->d_cams, cams.data(), p, (uint32_t*)d_mesh_tiles, (size_t)(mesh_stride_bytes / 4), true, 0, 0, &mo);
    hipError_t e = rc == RR_OK ? hipEventRecord(ctx->lane_done[lane], ctx->stream) : hipSuccess;
    ctx->lane_cams[lane] = ctx->d_cams;
    ctx->lane_cams_cap[lane] = ctx->cams_cap;
    ctx->stream = main_stream;
    ctx->d_cams = main_cams;
    ctx->cams_cap = main_cap;
    if (rc != RR_OK) return rc;
    if (e != hipSuccess) return fail(ctx, RR_ERR_DEVICE, "rr_render_orbit_mesh_sharded_lane: event", e);
    ctx->lane_busy[lane] = true;
    return RR_OK;
}

int rr_assemble_frames_mesh_rgb8(rr_context* ctx, const void* d_gathered, uint64_t rank_stride_bytes, uint64_t frame_stride_bytes,
                                 const void* d_bg_tiles, uint64_t bg_stride_bytes, const rr_mesh_partition* part, uint32_t n_frames,
                                 uint32_t width, uint32_t height, void* d_frames, uint64_t out_stride_bytes)
{
    const Range range_("rr_assemble_frames_mesh_rgb8");
    if (int r = use_device(ctx)) return r;
    if (!d_gathered || !d_frames || !part || part->world == 0 || width == 0 || height == 0 ||
        ((rank_stride_bytes | frame_stride_bytes | bg_stride_bytes | out_stride_bytes | (uint64_t)(uintptr_t)d_gathered | (uint64_t)(uintptr_t)d_bg_tiles) & 3u))
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_assemble_frames_mesh_rgb8: bad arguments (strides and buffers are 4-byte aligned)");
    const uint32_t tiles_x = (width + TILE - 1) / TILE, n_tiles = tiles_x * ((height + TILE - 1) / TILE);
    if (part->tiles_x != tiles_x || part->n_tiles != n_tiles || part->n_mesh_tiles + part->n_bg_tiles != n_tiles ||
        (part->rect_w == 0 ? part->n_bg_tiles != 0 : (part->rect_w * part->rect_h != part->n_mesh_tiles || part->rect_x0 + part->rect_w > tiles_x ||
                                                      (part->rect_y0 + part->rect_h) * tiles_x > n_tiles)))
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_assemble_frames_mesh_rgb8: the partition is not one of this frame size");
    if (frame_stride_bytes < (uint64_t)part->max_mesh_tiles_per_rank * TILE * TILE * 3 || out_stride_bytes < (uint64_t)width * height * 4 ||
        (part->n_bg_tiles && (!d_bg_tiles || bg_stride_bytes < (uint64_t)part->n_bg_tiles * TILE * TILE * 3)))
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_assemble_frames_mesh_rgb8: stride too small");
    const MeshPartDev mp = { part->tiles_x, part->n_tiles, part->rect_x0, part->rect_y0, part->rect_w, part->rect_h, part->world, part->rank0_rounds };
    RR_HIP(launch_assemble_frames_mesh_rgb8((const uint8_t*)d_gathered, (const uint8_t*)d_bg_tiles, (uint32_t*)d_frames, width, height, mp, rank_stride_bytes,
                                            frame_stride_bytes, bg_stride_bytes, out_stride_bytes / 4, n_frames, ctx->stream));
    return RR_OK;
}

int rr_assemble_frames(rr_context* ctx, const void* d_gathered, uint32_t world, uint64_t rank_stride_bytes,
                       uint64_t frame_stride_bytes, uint32_t n_frames, uint32_t width, uint32_t height, void* d_frames,
                       uint64_t out_stride_bytes)
{
    const Range range_("rr_assemble_frames");
    if (int r = use_device(ctx)) return r;
    if (!d_gathered || !d_frames || world == 0 || width == 0 || height == 0 || ((rank_stride_bytes | frame_stride_bytes | out_stride_bytes) & 3u))
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_assemble_frames: bad arguments");
    uint32_t tx, nt, local, mx;
    tile_counts(width, height, 0, world, tx, nt, local, mx);
    if (frame_stride_bytes < (uint64_t)mx * TILE * TILE * 4 || out_stride_bytes < (uint64_t)width * height * 4)
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_assemble_frames: stride too small");
    RR_HIP(launch_assemble_frames((const uint32_t*)d_gathered, (uint32_t*)d_frames, width, height, tx, nt, world,
                                  rank_stride_bytes / 4, frame_stride_bytes / 4, out_stride_bytes / 4, n_frames, ctx->stream));
    return RR_OK;
}

int rr_assemble_frames_rgb8(rr_context* ctx, const void* d_gathered, uint32_t world, uint64_t rank_stride_bytes,
                            uint64_t frame_stride_bytes, uint32_t n_frames, uint32_t width, uint32_t height, void* d_frames,
                            uint64_t out_stride_bytes)
{
    const Range range_("rr_assemble_frames_rgb8");
    if (int r = use_device(ctx)) return r;
    if (!d_gathered || !d_frames || world == 0 || width == 0 || height == 0 ||
        ((rank_stride_bytes | frame_stride_bytes | out_stride_bytes | (uint64_t)(uintptr_t)d_gathered) & 3u))
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_assemble_frames_rgb8: bad arguments (strides and buffers are 4-byte aligned)");
    uint32_t tx, nt, local, mx;
    tile_counts(width, height, 0, world, tx, nt, local, mx);
    if (frame_stride_bytes < (uint64_t)mx * TILE * TILE * 3 || out_stride_bytes < (uint64_t)width * height * 4)
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_assemble_frames_rgb8: stride too small");
    RR_HIP(launch_assemble_frames_rgb8((const uint8_t*)d_gathered, (uint32_t*)d_frames, width, height, tx, nt, world,
                                       rank_stride_bytes, frame_stride_bytes, out_stride_bytes / 4, n_frames, ctx->stream));
    return RR_OK;
}

// ---- RCCL, looked up at run time ---------------------------------------------------------------------------------
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, rr_nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    Rccl()
    {
        // a process that already holds an RCCL (PyTorch's) must use that one: two copies of its globals do not mix
        const char* names[] = { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" };
        for (const char* n : names) if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char* n : names) if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (!lib) return;
        GetUniqueId = (decltype(GetUniqueId))dlsym(lib, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(lib, "ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        Send = (decltype(Send))dlsym(lib, "ncclSend");
        Recv = (decltype(Recv))dlsym(lib, "ncclRecv");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        ok = GetUniqueId && CommInitRank && CommDestroy && GroupStart && GroupEnd && Send && Recv;
    }
};
extern "C++" const Rccl& rccl() { static const Rccl r; return r; }      // (this translation unit's tail is inside extern "C")
static_assert(sizeof(rr_nccl_unique_id) == 128, "rr_comm_unique_id hands out 128 bytes");
} // namespace

int rr_comm_unique_id(void* id128)
{
    if (!id128) return RR_ERR_INVALID_ARGUMENT;
    if (!rccl().ok) return RR_ERR_UNSUPPORTED;                    // no librccl.so on this machine
    return rccl().GetUniqueId(id128) == 0 ? RR_OK : RR_ERR_DEVICE;
}

int rr_comm_init(rr_context* ctx, const void* id128, int rank, int world, void** comm)
{
    if (int r = use_device(ctx)) return r;                        // the communicator belongs to the context's device
    if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_comm_init: bad arguments");
    if (!rccl().ok) return fail(ctx, RR_ERR_UNSUPPORTED, "rr_comm_init: librccl.so not found");
    rr_nccl_unique_id id;
    memcpy(&id, id128, sizeof id);
    *comm = nullptr;
    const int e = rccl().CommInitRank(comm, world, id, rank);
    if (e != 0) { ctx->err = std::string("ncclCommInitRank: ") + (rccl().GetErrorString ? rccl().GetErrorString(e) : "error"); return RR_ERR_DEVICE; }
    return RR_OK;
}

int rr_comm_destroy(void* comm)
{
    if (!comm) return RR_OK;
    if (!rccl().ok) return RR_ERR_UNSUPPORTED;
    return rccl().CommDestroy(comm) == 0 ? RR_OK : RR_ERR_DEVICE;
}

int rr_gather_frames(rr_context* ctx, void* comm, int rank, int world, const void* d_send, void* d_recv, uint64_t bytes_per_rank, int root)
{
    const Range range_("rr_gather_frames");
    if (int r = use_device(ctx)) return r;
    if (!comm || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || !d_send || (rank == root && !d_recv))
        return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_gather_frames: bad arguments");
    if (!rccl().ok) return fail(ctx, RR_ERR_UNSUPPORTED, "rr_gather_frames: librccl.so not found");
    if (bytes_per_rank == 0) return RR_OK;
    const Rccl& R = rccl();
    int e = R.GroupStart();
    if (e == 0) e = R.Send(d_send, (size_t)bytes_per_rank, (int)RR_NCCL_UINT8, root, comm, ctx->stream);
    if (rank == root)
        for (int r = 0; r < world && e == 0; ++r)
            e = R.Recv((char*)d_recv + (size_t)r * bytes_per_rank, (size_t)bytes_per_rank, (int)RR_NCCL_UINT8, r, comm, ctx->stream);
    const int e2 = R.GroupEnd();
    if (e == 0) e = e2;
    if (e != 0) { ctx->err = std::string("rr_gather_frames: ") + (R.GetErrorString ? R.GetErrorString(e) : "RCCL error"); return RR_ERR_DEVICE; }
    return RR_OK;
}

int rr_device_alloc(rr_context* ctx, uint64_t bytes, void** d_ptr)
{
    if (int r = use_device(ctx)) return r;
    if (!d_ptr || bytes == 0) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_device_alloc: bad arguments");
    RR_HIP(hipMalloc(d_ptr, (size_t)bytes));
    return RR_OK;
}

int rr_device_free(rr_context* ctx, void* d_ptr)
{
    if (int r = use_device(ctx)) return r;
    if (d_ptr) { RR_HIP(hipStreamSynchronize(ctx->stream)); RR_HIP(hipFree(d_ptr)); }
    return RR_OK;
}

int rr_device_read(rr_context* ctx, const void* d_src, void* host_dst, uint64_t bytes)
{
    if (int r = use_device(ctx)) return r;
    if (!d_src || !host_dst) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_device_read: null pointer");
    RR_HIP(hipMemcpyAsync(host_dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    RR_HIP(hipStreamSynchronize(ctx->stream));
    return RR_OK;
}

int rr_timing_begin(rr_context* ctx)
{
    if (int r = use_device(ctx)) return r;
    if (!ctx->ev_begin) { RR_HIP(hipEventCreate(&ctx->ev_begin)); RR_HIP(hipEventCreate(&ctx->ev_end)); }
    RR_HIP(hipEventRecord(ctx->ev_begin, ctx->stream));
    return RR_OK;
}

int rr_timing_end(rr_context* ctx, float* elapsed_ms)
{
    if (int r = use_device(ctx)) return r;
    if (!elapsed_ms || !ctx->ev_begin) return fail(ctx, RR_ERR_STATE, "rr_timing_end: rr_timing_begin first");
    RR_HIP(hipEventRecord(ctx->ev_end, ctx->stream));
    RR_HIP(hipEventSynchronize(ctx->ev_end));
    RR_HIP(hipEventElapsedTime(elapsed_ms, ctx->ev_begin, ctx->ev_end));
    return RR_OK;
}

int rr_kernel_time(rr_context* ctx, float* sum_ms, uint32_t* n_launches)
{
    if (int r = use_device(ctx)) return r;
    if (!sum_ms || !n_launches) return RR_ERR_INVALID_ARGUMENT;
    RR_HIP(hipStreamSynchronize(ctx->stream));
    double sum = 0.0;
    for (uint32_t i = 0; i < ctx->kev_used; ++i) {
        float ms = 0.0f;
        RR_HIP(hipEventElapsedTime(&ms, ctx->kev[(size_t)i * 2], ctx->kev[(size_t)i * 2 + 1]));
        sum += ms;
    }
    *sum_ms = (float)sum;
    *n_launches = ctx->kev_used;
    ctx->kev_used = 0;
    return RR_OK;
}

int rr_get_stats(rr_context* ctx, rr_stats* out)
{
    if (int r = use_device(ctx)) return r;
    if (!out) return RR_ERR_INVALID_ARGUMENT;
    if (int r = join_lanes(ctx)) return r;
    CounterBlock* h = (CounterBlock*)malloc(sizeof(CounterBlock));
    if (!h) return RR_ERR_OUT_OF_MEMORY;
    hipError_t e = hipMemcpyAsync(h, ctx->d_cnt, sizeof(CounterBlock), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { free(h); return fail(ctx, RR_ERR_DEVICE, "rr_get_stats", e); }
    memset(out, 0, sizeof *out);
    uint64_t rays = 0;
    for (int i = 0; i < RAY_SHARDS; ++i) rays += h->shards[i];
    out->rays = rays;
    out->pixels = ctx->accum_pixels;
    out->primary = ctx->accum_pixels;
    out->secondary = rays - out->primary;
    out->stats_valid = ctx->last_stats ? 1u : 0u;
    if (ctx->last_stats) {
        out->hits = h->counters[C_HITS]; out->misses = h->counters[C_MISSES]; out->terminal_hits = h->counters[C_TERMINAL];
        out->tir = h->counters[C_TIR]; out->node_visits = h->counters[C_NODES]; out->tri_tests = h->counters[C_TRIS];
        out->node_trips = h->counters[C_NODE_TRIPS]; out->leaf_trips = h->counters[C_LEAF_TRIPS];
        out->shade_passes = h->counters[C_PASSES]; out->waves = h->counters[C_WAVES]; out->background_waves = h->counters[C_BG_WAVES];
        out->clock_ticks = h->counters[C_CLK_TICKS]; out->clock_ref_ticks = h->counters[C_CLK_REAL];
    }
    memcpy(out->render_kernel_name, ctx->last_kernel_name, sizeof out->render_kernel_name);
    out->traversal_overflow = h->error;
    out->bvh_depth = scene_stack_need(ctx);
    out->render_kernel = ctx->last_kernel;
    free(h);
    return RR_OK;
}

int rr_trace_rays(rr_context* ctx, const rr_ray* rays, uint32_t n, rr_hit* hits)
{
    if (int r = use_device(ctx)) return r;
    if (!ctx->tlas_built) return fail(ctx, RR_ERR_STATE, "rr_trace_rays: build the BLAS and TLAS first");
    if (n == 0) return RR_OK;
    if (!rays || !hits) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_trace_rays: null arrays");
    if (n > ctx->ray_cap) {
        RR_HIP(hipStreamSynchronize(ctx->stream));
        dfree(ctx->d_rays); dfree(ctx->d_hits);
        ctx->ray_cap = 0;
        RR_HIP(hipMalloc(&ctx->d_rays, (size_t)n * sizeof(rr_ray_dev)));
        RR_HIP(hipMalloc(&ctx->d_hits, (size_t)n * sizeof(rr_hit_dev)));
        ctx->ray_cap = n;
    }
    SceneDev sc;
    fill_scene(ctx, sc);
    RR_HIP(hipMemsetAsync(&ctx->d_cnt->error, 0, 4, ctx->stream));
    RR_HIP(hipMemcpyAsync(ctx->d_rays, rays, (size_t)n * sizeof(rr_ray_dev), hipMemcpyHostToDevice, ctx->stream));
    RR_HIP(launch_trace_rays(sc, ctx->d_rays, n, ctx->d_hits, &ctx->d_cnt->error, scene_stack_need(ctx) <= 31 ? 31 : 64, ctx->stream));
    RR_HIP(hipMemcpyAsync(hits, ctx->d_hits, (size_t)n * sizeof(rr_hit_dev), hipMemcpyDeviceToHost, ctx->stream));
    uint32_t err = 0;
    RR_HIP(hipMemcpyAsync(&err, &ctx->d_cnt->error, 4, hipMemcpyDeviceToHost, ctx->stream));
    RR_HIP(hipStreamSynchronize(ctx->stream));
    if (err) return fail(ctx, RR_ERR_TRAVERSAL_OVERFLOW, "traversal stack overflow");
    return RR_OK;
}

int rr_env_lookup(rr_context* ctx, const float* dirs, uint32_t n, float* rgb)
{
    if (int r = use_device(ctx)) return r;
    if ((!dirs || !rgb) && n) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_env_lookup: null buffers");
    if (n == 0) return RR_OK;
    float *d_in = nullptr, *d_out = nullptr;
    RR_HIP(hipMalloc(&d_in, (size_t)n * 12));
    hipError_t e = hipMalloc(&d_out, (size_t)n * 12);
    SceneDev sc;
    fill_scene(ctx, sc);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, dirs, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = launch_env_lookup(sc, d_in, n, d_out, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(rgb, d_out, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in); (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ctx, RR_ERR_DEVICE, "rr_env_lookup", e);
    return RR_OK;
}

int rr_download_blas(rr_context* ctx, uint32_t mesh_id, void* nodes, uint32_t* n_nodes, void* tris, uint32_t* n_tris)
{
    if (int r = use_device(ctx)) return r;
    if (mesh_id >= ctx->meshes.size()) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_download_blas: unknown mesh id");
    const MeshRes& m = ctx->meshes[mesh_id];
    if (!m.built) return fail(ctx, RR_ERR_STATE, "rr_download_blas: BLAS not built");
    const uint32_t nn = m.n_tris > 1 ? m.n_tris - 1 : 1;
    if (n_nodes) *n_nodes = nn;
    if (n_tris) *n_tris = m.n_tris;
    RR_HIP(hipStreamSynchronize(ctx->stream));
    if (nodes) RR_HIP(hipMemcpy(nodes, m.nodes, (size_t)nn * sizeof(BvhNode), hipMemcpyDeviceToHost));
    if (tris) RR_HIP(hipMemcpy(tris, m.tris, (size_t)m.n_tris * sizeof(TriRec), hipMemcpyDeviceToHost));
    return RR_OK;
}

int rr_host_register(rr_context* ctx, void* p, size_t bytes)
{
    if (int r = use_device(ctx)) return r;
    if (!p || bytes == 0) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_host_register: null buffer");
    RR_HIP(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return RR_OK;
}

int rr_host_unregister(rr_context* ctx, void* p)
{
    if (int r = use_device(ctx)) return r;
    if (!p) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_host_unregister: null buffer");
    RR_HIP(hipHostUnregister(p));
    return RR_OK;
}

int rr_download_qnodes(rr_context* ctx, uint32_t mesh_id, void* qnodes, uint32_t* n_nodes, float grid_org_cell[6])
{
    if (int r = use_device(ctx)) return r;
    if (mesh_id >= ctx->meshes.size()) return fail(ctx, RR_ERR_INVALID_ARGUMENT, "rr_download_qnodes: unknown mesh id");
    const MeshRes& m = ctx->meshes[mesh_id];
    if (!m.built) return fail(ctx, RR_ERR_STATE, "rr_download_qnodes: BLAS not built");
    const uint32_t nn = m.n_tris > 1 ? m.n_tris - 1 : 1;
    if (n_nodes) *n_nodes = nn;
    if (grid_org_cell) { memcpy(grid_org_cell, m.grid.org, 12); memcpy(grid_org_cell + 3, m.grid.cell, 12); }
    RR_HIP(hipStreamSynchronize(ctx->stream));
    if (qnodes) RR_HIP(hipMemcpy(qnodes, m.qnodes, (size_t)nn * sizeof(QNode), hipMemcpyDeviceToHost));
    return RR_OK;
}

} // extern "C"
