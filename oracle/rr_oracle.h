/*
 * rr_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, fp32 restatement of the reference's hot path:
 *   RayGen / ClosestHit / Miss            /root/reference/RayTracing.hlsl:27-137
 *   camera constants                      /root/reference/RefractionDemo.cpp:559-567
 *   OBJ loader                            /root/reference/Mesh.cpp:6-37
 *   UNORM8 render target                  /root/reference/RefractionDemo.cpp:428-434
 *   DXR TraceRay semantics (driver side, not in the tree)  SURVEY.md Appendix A.2
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this file.  The shipped library (include/rrdxr.h) never links it.
 *
 * PARITY PIN STATUS: the reference ships no tests, golden images or vectors
 * for the render path and cannot run outside Windows/D3D12, so the *render*
 * is "parity unpinned" (SURVEY.md 8c).  What IS pinned here: the OBJ loader
 * (FNV-1a-64 hashes of SURVEY Appendix B, produced by the reference's own
 * Mesh::load body), the camera (Appendix A.1 known-answer rays), the
 * workload ray counts (Appendix C) and closed-form scenes.
 */
#ifndef RR_ORACLE_H
#define RR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mesh.hpp:6-11 / RayTracing.hlsl:5-9 -- 32 bytes */
typedef struct rro_vertex {
    float position[3];
    float norm[3];
    float uv[2];
} rro_vertex;

/* mirrors the 64-byte D3D12_RAYTRACING_INSTANCE_DESC (RefractionDemo.cpp:324-334) */
typedef struct rro_instance {
    float    transform[12];      /* 3x4 row-major object->world */
    uint32_t id_mask;            /* InstanceID:24 | InstanceMask:8 */
    uint32_t hitgroup_flags;     /* InstanceContributionToHitGroupIndex:24 | Flags:8 */
    uint64_t blas;               /* mesh id returned by rro_scene_add_mesh */
} rro_instance;

/* literals of RayTracing.hlsl:52-53,82,95,99-100,110,114-115 */
typedef struct rro_params {
    int   max_refract;           /* 5    (hlsl:82)  */
    int   max_reflect;           /* 2    (hlsl:110) */
    float ior;                   /* 1.3  (hlsl:95)  */
    float tmin_primary;          /* 1e-4 (hlsl:52)  */
    float tmax_primary;          /* 100  (hlsl:53)  */
    float tmin_secondary;        /* 1e-3 (hlsl:99)  */
    float tmax_secondary;        /* 1000 (hlsl:100) */
    int   use_libm;              /* 0: spec'd rr_atan2f/rr_acosf; 1: libm atan2f/acosf */
    int   accum_mode;            /* 0: literal recursive +=; 1: path-weight sum in DFS order */
    int   use_bvh;               /* 0: brute force over all triangles; 1: CPU median-split BVH */
    int   tonemap;               /* 0: the reference's saturating UNORM8 store (hlsl:62); 1: c / (1 + c) first (SURVEY 8f.2,
                                    the product's RR_DISPATCH_TONEMAP_REINHARD) */
} rro_params;

typedef struct rro_stats {
    uint64_t rays;               /* every TraceRay call, primary + secondary */
    uint64_t primary;
    uint64_t secondary;
    uint64_t hits;
    uint64_t misses;
    uint64_t terminal_hits;      /* ClosestHit invoked with count >= max_refract (A.4) */
    uint64_t tir;                /* RefractRay returned false */
    uint64_t max_rays_per_pixel;
    uint64_t rays_per_level[32];
    uint64_t tri_tests;
    uint64_t node_visits;
} rro_stats;

typedef struct rro_hit {
    float    t, u, v;            /* u weights vertex 1, v weights vertex 2 (hlsl:86) */
    uint32_t prim;
    uint32_t inst;
    int      hit;
} rro_hit;

typedef struct rro_scene rro_scene;

void      rro_default_params(rro_params* p);
uint64_t  rro_fnv1a64(const void* bytes, uint64_t n);

/* Mesh.cpp:6-37.  Returns 0 if the file cannot be opened, else 1 (like Mesh::load).
 * Arrays are malloc'ed; free with rro_free. */
int       rro_mesh_load(const char* filename, rro_vertex** verts, uint32_t* n_verts,
                        uint32_t** indices, uint32_t* n_indices);
void      rro_free(void* p);

/* RefractionDemo.cpp:559-566: proj_inv (64 CPU row-major bytes) + camera_loc. */
void      rro_camera(float angle, float fov_y, float aspect, float zn, float zf,
                     float proj_inv[16], float camera_loc[4]);
/* RayTracing.hlsl:27-40 */
void      rro_generate_camera_ray(const float proj_inv[16], const float camera_loc[4],
                                  uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                                  float origin[3], float dir[3]);

rro_scene* rro_scene_create(void);
void       rro_scene_destroy(rro_scene* s);
/* copies; returns mesh id */
int        rro_scene_add_mesh(rro_scene* s, const rro_vertex* verts, uint32_t n_verts,
                              const uint32_t* indices, uint32_t n_indices);
/* n == 0 / never called: one identity instance of mesh 0 (RefractionDemo.cpp:324-334) */
int        rro_scene_set_instances(rro_scene* s, const rro_instance* inst, uint32_t n);
int        rro_scene_set_envmap(rro_scene* s, const float* rgb, int w, int h);

/* One TraceRay (closest hit, face culling).  flags: 0x10 CULL_BACK, 0x20 CULL_FRONT. */
void       rro_trace(const rro_scene* s, const float origin[3], const float dir[3],
                     float tmin, float tmax, uint32_t flags, int use_bvh, rro_hit* out);
/* Miss shader env lookup (hlsl:127-137) */
void       rro_env_lookup(const rro_scene* s, const float dir[3], int use_libm, float rgb[3]);

/* DispatchRays(W,H,1) restricted to pixel rows/cols [x0,x1) x [y0,y1) and, if
 * tile_world > 1, to the tile_w x tile_h tiles owned by tile_rank (round robin).
 * out_rgb: W*H*3 floats (full frame addressing), out_rgba8: W*H*4 bytes; either may be NULL.
 * out_raycount: W*H uint16 rays per pixel, may be NULL.  Untouched pixels are left as they are. */
int        rro_render(const rro_scene* s, const float proj_inv[16], const float camera_loc[4],
                      uint32_t w, uint32_t h, const rro_params* p,
                      uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                      uint32_t tile_w, uint32_t tile_h, uint32_t tile_rank, uint32_t tile_world,
                      int n_threads,
                      float* out_rgb, uint8_t* out_rgba8, uint16_t* out_raycount,
                      rro_stats* stats);

/* spec'd transcendental approximations (shared *definition* with the HIP path) */
float      rro_atan2f(float y, float x);
float      rro_acosf(float x);
uint8_t    rro_unorm8(float x);

#ifdef __cplusplus
}
#endif
#endif
