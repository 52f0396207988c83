// rr_launch.h -- host-callable launchers of the gfx950 kernels (implemented in the .hip files)
#pragma once
#include <hip/hip_runtime.h>
#include "rr_types.h"

namespace rr {

// device twins of rr_ray / rr_hit (include/rrdxr.h), same layout
struct alignas(16) rr_ray_dev { float origin[3]; float tmin; float dir[3]; float tmax; uint32_t flags; uint32_t pad[3]; };
struct rr_hit_dev { float t, u, v; uint32_t prim, inst, hit; };
static_assert(sizeof(rr_ray_dev) == 48 && sizeof(rr_hit_dev) == 24, "ABI layout");

// ---- rr_render.hip
hipError_t launch_render_fused(const SceneDev& sc, const DispatchDev& a, int stack, int pend, bool stats, hipStream_t s,
                               bool stack16 = false);
// BLAS nodes in LDS, persistent workgroups (single identity instance whose node array fits: lds_kernel_shape() >= 0)
// shapes (waves per workgroup x workgroups per CU): 0 = 12x2 (the product shape), 1 = 16x2, 2 = 16x1 (experiments:
// RR_DEBUG_SHAPE = first shape to consider); -1: the node array does not fit
int lds_kernel_shape(uint32_t node_bytes, uint32_t stack_entries, size_t* lds_bytes, int min_shape = 0);
hipError_t launch_render_lds(const SceneDev& sc, const DispatchDev& a, LdsDispatch q, int n_cus, bool stats, hipStream_t s, int min_shape = 0);
// launches of one or two slices: four lanes per pixel inside the scene's screen rectangle (k_render_paths)
hipError_t launch_render_paths(const SceneDev& sc, const DispatchDev& a, int stack, bool stats, hipStream_t s);
hipError_t launch_assemble_frames_mesh_rgb8(const uint8_t* gathered, const uint8_t* bg, uint32_t* frames, uint32_t W, uint32_t H, const MeshPartDev& mp,
                                            size_t rank_stride_b, size_t frame_stride_b, size_t bg_stride_b, size_t out_stride, uint32_t n_frames, hipStream_t s);
// ---- rr_render_stream.hip: one kernel per ray generation, rays in HBM queues, lanes refilled as their rays end (two-level scenes)
hipError_t launch_render_stream(const SceneDev& sc, const DispatchDev& a, const StreamDev& s, int stack, uint32_t n_wg, bool stats, hipStream_t st, int waves = 6);
const char* last_stream_kernel_name();
// the instantiation the calling thread's last launch_render_* call launched, e.g. "k_render_fused<19, 2, false, false, false, unsigned int, 0>"
const char* last_render_kernel_name();
hipError_t launch_trace_rays(const SceneDev& sc, const rr_ray_dev* rays, uint32_t n, rr_hit_dev* hits, uint32_t* err,
                             int stack, hipStream_t s);
hipError_t launch_screen_tables(float* out, uint32_t W, uint32_t H, hipStream_t s);
hipError_t launch_env_lookup(const SceneDev& sc, const float* dirs, uint32_t n, float* rgb, hipStream_t s);
hipError_t launch_assemble_tiles(const uint32_t* gathered, uint32_t* frame, uint32_t W, uint32_t H, uint32_t tiles_x,
                                 uint32_t n_tiles, uint32_t world, uint32_t max_tiles, hipStream_t s);
// batched: strides in 32-bit words
hipError_t launch_assemble_frames(const uint32_t* gathered, uint32_t* frames, uint32_t W, uint32_t H, uint32_t tiles_x,
                                  uint32_t n_tiles, uint32_t world, size_t rank_stride, size_t frame_stride,
                                  size_t out_stride, uint32_t n_frames, hipStream_t s);
hipError_t launch_assemble_frames_rgb8(const uint8_t* gathered, uint32_t* frames, uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles,
                                       uint32_t world, size_t rank_stride_b, size_t frame_stride_b, size_t out_stride, uint32_t n_frames,
                                       hipStream_t s);

// ---- rr_bvh_build.hip
// Scratch + outputs of one LBVH build over n primitives (triangles of a mesh, or instances).
struct BuildBuffers {
    uint32_t n;                     // primitives
    uint32_t n_pad;                 // next power of two >= n (sort width)
    float*   prim_box;              // n * 6 (lo.xyz, hi.xyz) in primitive order
    unsigned long long* keys;       // n_pad  (morton30 << 32 | prim)
    int32_t* parent;                // (2n-1): [0,n-1) internal, [n-1, 2n-1) leaves (sorted order)
    int32_t* child;                 // (n-1)*2 child refs
    float*   node_box;              // (2n-1)*6 boxes, same indexing as parent
    uint32_t* visit;                // (n-1) arrival counters
    uint32_t* scene_box;            // 6 ordered-uint encodings of the scene bounds
    uint32_t* depth;                // 1
    uint32_t* ploc;                 // 2*n cluster arrays of the PLOC builder
    BvhNode* nodes;                 // out: max(n-1,1)
    uint32_t leaf_ref_prim;         // 0: leaf ref = ~sorted position (BLAS); 1: ~(leaf_base + primitive index) (TLAS)
    uint32_t leaf_base;
};

hipError_t launch_tri_setup(const void* verts, const uint32_t* idx, uint32_t n_tris, const BuildBuffers& b, hipStream_t s);
hipError_t launch_lbvh(const BuildBuffers& b, hipStream_t s);
// PREFER_FAST_TRACE hierarchy: Morton order + parallel locally-ordered clustering (n <= PLOC_MAX_PRIMS)
constexpr uint32_t PLOC_MAX_PRIMS = 32768;
hipError_t launch_ploc(const BuildBuffers& b, hipStream_t s);
// after launch_lbvh: keys[i] & 0xffffffff is the primitive of sorted leaf i
hipError_t launch_pack_tris(const void* verts, const uint32_t* idx, const BuildBuffers& b, TriRec* tris, NrmRec* nrms,
                            hipStream_t s);
hipError_t launch_inst_setup(const InstDev* insts, const float* blas_bounds, uint32_t n, const BuildBuffers& b, hipStream_t s);
// copy a BLAS into the scene pool: internal refs += node_off, leaf refs ~l -> ~(l + tri_off)
hipError_t launch_quantize_nodes(QNode* dst, const BvhNode* src, uint32_t n_nodes, const QGrid& g, uint32_t node_off, uint32_t tri_off,
                                 hipStream_t s);
hipError_t launch_env_pad(const float* rgb, float4* out, uint32_t n_texels, hipStream_t s);

} // namespace rr
