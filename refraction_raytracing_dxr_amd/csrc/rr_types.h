// rr_types.h -- records shared by the host side of the C ABI and the gfx950 kernels.
//
// HBM layout (all arrays 16-byte aligned, hipMalloc'ed once per mesh / scene):
//   QNode    32 B  what traversal reads: both child boxes as fp16 cell counts on the grid of the BLAS bounds + both
//                  child refs, TWO 16-byte requests per visit
//   BvhNode  64 B  the same node in fp32 (builder output, rr_download_blas); QNode is derived from it
//   TriRec   48 B  v0,e1,e2 in LBVH leaf order, original PrimitiveIndex() in v0.w
//   NrmRec   48 B  the three vertex normals of the same triangle (ClosestHit, RayTracing.hlsl:83-85)
//   env      16 B  per texel (RGB32F padded to float4: one dwordx4 load per Miss)
#pragma once
#include <stdint.h>

namespace rr {

// child ref: >= 0 internal node index, < 0 leaf holding primitive ~ref (index into TriRec / instances).
// The two child boxes are stored plane-pair by plane-pair, (child0, child1) adjacent, so that one
// v_pk_fma_f32 evaluates the same slab plane of both children (packed FP32 is the full-rate path on CDNA).
struct alignas(16) BvhNode {
    float lox[2], loy[2];     // 16 B: lower x of child 0/1, lower y of child 0/1
    float loz[2], hix[2];
    float hiy[2], hiz[2];
    int32_t c[2];             // child refs
    uint32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 B");

// Grid over the bounds of one BLAS (object space) or of the scene (TLAS level): plane = org + q*cell, org the centre
// of the bounds, 65530 cells across the extent.
struct QGrid { float org[3]; float cell[3]; };
// Traversal node.  Every word holds one slab plane of both children as fp16 cell counts q: child 0 in the low half,
// child 1 in the high half (v_fma_mix_f32 reads either half directly).  lo planes are rounded down, hi planes up:
// the stored box always contains the fp32 box, which is all the box test needs.
struct alignas(16) QNode {
    uint32_t lox, loy, loz, hix;
    uint32_t hiy, hiz;
    int32_t c[2];             // child refs: >= 0 BYTE offset of an internal node in this array, < 0 leaf ~index
};
static_assert(sizeof(QNode) == 32, "QNode must be 32 B");

struct alignas(16) TriRec {
    float v0[3]; uint32_t prim;
    float e1[3]; uint32_t pad1;
    float e2[3]; uint32_t pad2;
};
static_assert(sizeof(TriRec) == 48, "TriRec must be 48 B");

struct alignas(16) NrmRec {
    float nA[3]; float pad0;
    float nB[3]; float pad1;
    float nC[3]; float pad2;
};
static_assert(sizeof(NrmRec) == 48, "NrmRec must be 48 B");

// one BLAS as the traversal kernels see it
struct BlasDev {
    const QNode* nodes;
    QGrid grid;
    const TriRec*  tris;
    const NrmRec*  nrms;
    uint32_t n_tris;
    uint32_t depth;
    float    scale;           // largest |coordinate| of the mesh bounds (box-test padding)
    uint32_t pad;
};

// instance record for two-level traversal (derived from the 64-byte rr_instance_desc)
struct alignas(16) InstDev {
    float inv[12];            // world -> object 3x4
    uint32_t root;            // byte offset of the root node of this instance's BLAS inside the scene pool
    uint32_t flags;           // RR_INSTANCE_FLAG_*
    uint32_t mask;
    uint32_t identity;
    float    scale;           // largest |coordinate| of the BLAS bounds (box-test padding in object space)
    QGrid    grid;            // of the BLAS this instance refers to
    uint32_t pad;
};
static_assert(sizeof(InstDev) == 96, "InstDev must be 96 B");

// Scenes beyond the reference's single identity instance are FLATTENED at rr_build_tlas: the TLAS nodes
// and the nodes of every BLAS in use live in one node pool (child refs rebased), triangles and normals
// in one pool each.  A leaf ref ~L is a triangle for L < n_pool_tris and instance L - n_pool_tris
// otherwise: one base address and one kind of stack entry for both levels (the lock-step kernels walk the levels nested,
// trace_scene; the stream renderer's lanes are at either level independently).
struct SceneDev {
    BlasDev  blas0;           // used directly when the scene is one identity instance (the reference's case)
    const QNode* pool_nodes;      // [0, n_insts-1): TLAS (on the scene grid), then the BLASes (each on its own grid)
    QGrid grid;                   // scene grid (world space)
    const TriRec*  pool_tris;
    const NrmRec*  pool_nrms;
    const InstDev* insts;
    uint32_t n_insts;
    uint32_t n_pool_tris;
    uint32_t single_identity; // 1: skip the top level entirely
    float scale;              // largest |coordinate| of the world-space scene bounds
    const float4* env;        // w*h float4
    int32_t env_w, env_h;
};

// SceneConstants as they sit in the device-side constant buffer (RefractionDemo.cpp:533-534, :566)
struct CamDev {
    float M[16];              // proj_inv, CPU row-major bytes
    float cam[4];
};

struct DispatchDev {
    const CamDev* cams;       // one per depth slice of the dispatch (DispatchRays(W,H,Depth))
    uint32_t n_frames;        // Depth
    uint32_t blocks_per_frame;
    size_t   frame_stride;    // output elements between consecutive slices
    uint32_t W, H;
    uint32_t tiles_x, n_tiles;      // 32x32 tiles over the frame
    uint32_t tile_rank, tile_world;
    uint32_t n_local_tiles;         // tiles this rank renders
    uint32_t n_blocks;              // 4 blocks (32x8 strips) per local tile, tiles rounded up to a multiple of 8
    uint32_t compact_out;           // 0: W*H raster; 1: [local tile][32*32] RGBA8 (sharded frames); 2: the same as RGB8, 3 B/px
    uint32_t tonemap;               // RR_DISPATCH_TONEMAP_REINHARD: the UNORM8 store takes c / (1 + c)
    int32_t max_refract, max_reflect;
    float ior, inv_ior;
    float tmin_p, tmax_p, tmin_s, tmax_s;
    const float* sx;                // GenerateCameraRay's screen coordinates, one per column / row (k_screen_tables):
    const float* sy;                //   sx[x] = (x + 0.5) / W * 2 - 1,  sy[y] = -((y + 0.5) / H * 2 - 1)
    uint32_t hx0, hy0, hx1, hy1;    // pixels outside this rectangle cannot see the scene: their primary ray is a Miss without a trace
    // Order of the tiles inside a launch (wave_block_pos): the tiles that touch the rectangle first, in raster order, then the
    // rest -- the expensive waves start at once and the launch ends on background waves, a few microseconds each, instead of on
    // the last mesh waves (hundreds).  rt_w == 0: image order.  Tile units; rt_div_w / rt_div_o: 2^32 / d + 1 for the two
    // divisions (exact below 65 536 tiles).
    uint32_t rt_x0, rt_y0, rt_w, rt_h, rt_div_w, rt_div_o;
    // mesh-tile partition (rr_mesh_partition): this rank's launch order is its n_mesh_local mesh tiles (order index k * world +
    // rank among the rectangle's tiles), then -- rank 0 only -- every background tile; mesh tiles are written to out_rgba8
    // (slot k), background tiles to out_bg (slot k - n_mesh_local)
    uint32_t mesh_part, n_mesh_local, n_rect_tiles, mesh_rounds;     // mesh_rounds: rr_mesh_partition::rank0_rounds
    uint32_t* out_bg;
    size_t   bg_stride;
    uint32_t group_trace;           // k_render_paths: 1 = ray levels with few rays left are traced by groups of 2 / 4 lanes per ray (trace_blas_group)
    uint32_t async_leaf_num, async_shade_num;   // k_stream_rays: lanes (in sixteenths of the live lanes) a step / a shading pass needs to be issued
    uint32_t* out_rgba8;            // world==1: W*H raster; else compact tiles
    float4*   out_f32;              // optional, same addressing
    unsigned long long* counters;   // rr::Counter slots
    uint32_t* ray_shards;           // RAY_SHARDS u32 partial ray counts
    uint32_t* error_flag;
    unsigned long long* diag;       // diagnostic builds only: 4 x u64 per wave {start, cycles, rays(max lane), loop trips}
};

// k_render_lds (persistent workgroups, BLAS nodes in LDS): the work queues of one launch.
// Work is handed out in two phases, a ticket at a time; the wave that draws a ticket shares its blocks with its workgroup
// through LDS (rr_render.hip).  Phase 1: the screen rectangle that bounds the mesh in every slice of the launch (the projection
// of the BLAS bounds, computed by the host from the slices' constants, widened to 32-pixel columns) in 32x8 strips of one slice
// each -- these hold every secondary ray, i.e. every expensive block, and start first.  Phase 2: every 32x32 tile of every
// slice in image order, minus the blocks phase 1 rendered: the background, a microsecond per block, sixteen blocks per ticket.
struct LdsDispatch {
    uint32_t* tickets;          // LDS_TICKET_WORDS words, one counter per 64-byte line: phase 1 queues, phase 2 queues, finished workgroups
    uint32_t p1_tickets;        // phase 1: strips of the rectangle * slices; ticket t is slice t % n_frames of strip t / n_frames
    uint32_t pad0;              // (the layout of this block decides which arguments the compiler fetches together, and with that how
                                //  many scalars k_render_lds moves through vector lanes: 49 with these three words in place, 223 without)
    uint32_t rect_bw;           // rectangle width in 8x8 blocks (a multiple of 4)
    uint32_t rx0, ry0, rx1, ry1;// the rectangle in pixels: x multiples of 32, y multiples of 8 (rx1 <= rx0: empty)
    uint32_t p2_tickets;        // phase 2: tiles * slices; ticket t is slice t % n_frames of tile t / n_frames
    uint32_t pad1, pad2;
    uint32_t n_queues;          // ticket queues per phase in use (<= LDS_QUEUES)
    uint32_t home_xcc;          // 1: a wave's first queue is its XCD's number, 0: its own number, modulo n_queues
    uint32_t* park;             // parked reflected rays: [wave of the grid][park_slots][8 words][64 lanes]
    uint32_t park_slots;        // >= max_reflect
    uint32_t node_bytes;        // size of the node array copied to LDS (multiple of 32)
    uint32_t stack_entries;     // per-lane traversal stack entries (> tree depth)
    uint32_t div_frames, div_tiles_x, div_per_row;   // 2^32 / d + 1 for d = n_frames, tiles_x, strips per rectangle row: x / d == umulhi(x, div) for x * d < 2^32
};
// Ticket counters: ticket u belongs to queue u % n_queues; a wave starts on the queue of its XCD's number and, once that
// is empty, goes through the others one by one, each looked at with a plain load before it is drawn from (a counter only ever
// grows, so a queue seen empty stays empty).  Eight queues are the product setting: with a number of slices that is a multiple
// of eight an XCD then keeps to every eighth slice, which its L2 rewards (a word takes about 88 returning atomics per
// microsecond, MI355X_MICROARCH.md "dequeue": ~600 tickets per microsecond over eight).  RR_DEBUG_TICKET +64 / +128 select
// 64 / 512 queues for experiments.
constexpr uint32_t LDS_QUEUES = 512;
constexpr uint32_t LDS_TICKET_WORDS = (2 * LDS_QUEUES + 1) * 16;

// rr_mesh_partition as the de-interleave kernel takes it
struct MeshPartDev { uint32_t tiles_x, n_tiles, rect_x0, rect_y0, rect_w, rect_h, world, rounds; };
// how mesh tile i is dealt (rr_mesh_partition::rank0_rounds = j; the same arithmetic as rr_host_mesh_tile_home)
#if defined(__HIPCC__)
__device__ __forceinline__ void mesh_deal_owner(uint32_t i, uint32_t world, uint32_t j, uint32_t& rank, uint32_t& slot)
{
    if (j == 0u || world == 1u) { rank = i % world; slot = i / world; return; }
    if (j == 0xffffffffu) { rank = 1u + i % (world - 1u); slot = i / (world - 1u); return; }
    const uint32_t cl = j * (world - 1u) + 1u, c = i / cl, pos = i % cl;
    if (pos == cl - 1u) { rank = 0u; slot = c; }
    else { rank = 1u + pos % (world - 1u); slot = c * j + pos / (world - 1u); }
}
// the inverse for one rank: its slot k -> mesh tile index
__device__ __forceinline__ uint32_t mesh_deal_index(uint32_t rank, uint32_t k, uint32_t world, uint32_t j)
{
    if (j == 0u || world == 1u) return k * world + rank;
    if (j == 0xffffffffu) return k * (world - 1u) + (rank - 1u);            // (rank 0 holds no mesh tile)
    const uint32_t cl = j * (world - 1u) + 1u;
    if (rank == 0u) return k * cl + cl - 1u;
    return (k / j) * cl + (k % j) * (world - 1u) + (rank - 1u);
}
#endif

// k_stream_* (rr_render_stream.hip): the ray queues of one pass of the generation-per-kernel renderer
constexpr uint32_t STREAM_MAX_GEN = 64;     // head counters: generations 0 .. max_refract + 1 <= 63
struct StreamDev {
    float4*   q[2];           // ray queues (generation g reads q[g & 1], writes q[(g + 1) & 1]): cap entries of 3 x float4
    uint32_t* fill[2];        // rays in each 64-entry chunk of the queues' reserved blocks
    uint32_t* heads;          // [STREAM_MAX_GEN]: entries reserved in the queue that FEEDS generation g (multiples of 1 024)
    uint32_t* next;           // [STREAM_MAX_GEN][8][16]: chunk ticket counters of generation g's kernel, one per XCD, 64 B apart
    float4*   slots;          // [n_rect_wb * 64][4]: (weight, texel) of a pixel's up to four leaves, in the recursion's order
    uint8_t*  pending;        // [n_rect_wb * 64]: 1 = the pixel's colour is the sum of its slots (resolve), 0 = already stored
    uint32_t  cap;            // entries per queue (a multiple of 1 024)
    uint32_t  n_rect_wb;      // wave-blocks (8x8 pixel blocks, numbered as k_render_fused's) the ray kernels render; the rest
                              // are background blocks (k_stream_background)
};

enum Counter : int {
    C_RAYS = 0, C_PRIMARY, C_SECONDARY, C_HITS, C_MISSES, C_TERMINAL, C_TIR, C_NODES, C_TRIS,
    C_NODE_TRIPS, C_LEAF_TRIPS, C_PASSES, C_WAVES,      // wave-level loop trips of the STATS builds (what the vector unit issues for)
    C_BG_WAVES,                                         // waves of background blocks (k_render_fused's RayGen + Miss branch)
    C_CLK_TICKS, C_CLK_REAL,                            // STATS builds: sum over waves of their lifetime in shader-clock ticks (s_memtime) and on the
                                                        // 100 MHz reference (s_memrealtime): the clock the launch ran at = ticks / real * 100 MHz
    C_COUNT
};
constexpr int RAY_SHARDS = 1024;
constexpr int TILE = 32;

} // namespace rr
