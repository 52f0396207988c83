/*
 * rr_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See rr_oracle.h for scope, citations and the parity-pin statement.
 *
 * Build: oracle/Makefile (gcc -O2 -ffp-contract=off -mfma ...).  All arithmetic
 * is fp32; contraction is off so that every fused multiply-add below is an
 * explicit fmaf().  The operation order written here is the arithmetic
 * specification DESIGN.md section "Arithmetic" refers to.
 */
#define _GNU_SOURCE
#include "rr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ vectors */
typedef struct { float x, y, z; } v3;

static inline v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }
/* dot(a,b) = fma(az,bz, fma(ay,by, ax*bx)) */
static inline float v3_dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
/* cross(a,b).x = fma(ay,bz, -(az*by)) ... */
static inline v3 v3_cross(v3 a, v3 b)
{
    return v3_make(fmaf(a.y, b.z, -(a.z * b.y)),
                   fmaf(a.z, b.x, -(a.x * b.z)),
                   fmaf(a.x, b.y, -(a.y * b.x)));
}
/* HLSL normalize(v) = v * rsqrt(dot(v,v)); spec: v * (1 / sqrt(dot)) with IEEE sqrt and divide */
static inline v3 v3_normalize(v3 a)
{
    float inv = 1.0f / sqrtf(v3_dot(a, a));
    return v3_scale(a, inv);
}

/* ------------------------------------------------ spec'd transcendentals
 * HLSL atan2/acos are implementation-defined approximations (DXC lowers them
 * to driver intrinsics).  The oracle and the HIP path both use the Cephes
 * single-precision minimax forms below so that texel selection is bit-stable;
 * tests/test_oracle_goldens.py bounds their distance from libm. */
static float rro_atanf_pos(float x) /* x >= 0 */
{
    float y0;
    if (x > 2.414213562373095f) {          /* tan(3pi/8) */
        y0 = 1.5707963267948966f;
        x = -(1.0f / x);
    } else if (x > 0.4142135623730950f) {  /* tan(pi/8) */
        y0 = 0.7853981633974483f;
        x = (x - 1.0f) / (x + 1.0f);
    } else {
        y0 = 0.0f;
    }
    float z = x * x;
    float p = ((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z
              - 3.33329491539e-1f;
    float r = p * z * x + x;
    return y0 + r;
}

float rro_atan2f(float y, float x)
{
    /* quadrant handling of C atan2f; inputs here are finite direction components */
    if (x != x || y != y) return NAN;
    if (y == 0.0f) {
        if (x > 0.0f || (x == 0.0f && !signbit(x))) return y;          /* +-0 */
        return signbit(y) ? -3.14159265358979323846f : 3.14159265358979323846f;
    }
    if (x == 0.0f) return y > 0.0f ? 1.5707963267948966f : -1.5707963267948966f;
    float a = rro_atanf_pos(fabsf(y) / fabsf(x));
    if (x < 0.0f) a = 3.14159265358979323846f - a;
    return y < 0.0f ? -a : a;
}

static float rro_asinf_core(float a) /* 0 <= a <= 1 */
{
    float z, x;
    int flag = 0;
    if (a > 0.5f) {
        z = 0.5f * (1.0f - a);
        x = sqrtf(z);
        flag = 1;
    } else {
        x = a;
        z = x * x;
    }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z
                + 7.4953002686e-2f) * z + 1.6666752422e-1f);
    float r = p * z * x + x;
    if (flag) {
        r = r + r;
        r = 1.5707963267948966f - r;
    }
    return r;
}

float rro_acosf(float x)
{
    if (!(x >= -1.0f && x <= 1.0f)) return NAN;   /* HLSL acos(|x|>1) = NaN */
    if (x < -0.5f)
        return 3.14159265358979323846f - 2.0f * rro_asinf_core(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f)
        return 2.0f * rro_asinf_core(sqrtf(0.5f * (1.0f - x)));
    float s = rro_asinf_core(fabsf(x));
    if (x < 0.0f) s = -s;
    return 1.5707963267948966f - s;
}

/* D3D ftou: round toward zero, NaN -> 0, negative -> 0, overflow -> 0xffffffff */
static inline uint32_t rro_ftou(float f)
{
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

/* typed UAV store to R8G8B8A8_UNORM (RefractionDemo.cpp:431): NaN -> 0, clamp, round half up */
uint8_t rro_unorm8(float x)
{
    if (!(x > 0.0f)) return 0;
    if (x >= 1.0f) return 255;
    return (uint8_t)floorf(x * 255.0f + 0.5f);
}

/* SURVEY 8f.2 tone map option (not in the reference): c / (1 + c); NaN and c <= 0 -> 0, +inf -> 1 */
static float rro_reinhard(float x)
{
    if (!(x > 0.0f)) return 0.0f;
    float xm = x < 3.4028234663852886e38f ? x : 3.4028234663852886e38f;
    return xm / (1.0f + xm);
}

void rro_default_params(rro_params* p)
{
    p->max_refract = 5;
    p->max_reflect = 2;
    p->ior = 1.3f;
    p->tmin_primary = 0.0001f;
    p->tmax_primary = 100.0f;
    p->tmin_secondary = 0.001f;
    p->tmax_secondary = 1000.0f;
    p->use_libm = 0;
    p->accum_mode = 0;
    p->use_bvh = 0;
    p->tonemap = 0;
}

uint64_t rro_fnv1a64(const void* bytes, uint64_t n)
{
    const uint8_t* b = (const uint8_t*)bytes;
    uint64_t h = 1469598103934665603ull;
    for (uint64_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

void rro_free(void* p) { free(p); }

/* ------------------------------------------------------- Mesh::load (Mesh.cpp:6-37) */
typedef struct { float* d; size_t n, cap; } fvec;
static void fvec_push(fvec* v, const float* src, size_t k)
{
    if (v->n + k > v->cap) {
        v->cap = v->cap ? v->cap * 2 : 1024;
        while (v->cap < v->n + k) v->cap *= 2;
        v->d = (float*)realloc(v->d, v->cap * sizeof(float));
    }
    memcpy(v->d + v->n, src, k * sizeof(float));
    v->n += k;
}

int rro_mesh_load(const char* filename, rro_vertex** verts_out, uint32_t* n_verts,
                  uint32_t** idx_out, uint32_t* n_idx)
{
    FILE* f = fopen(filename, "rb");                       /* Mesh.cpp:8-10 */
    if (!f) return 0;
    fvec locs = { 0 }, uvs = { 0 }, norms = { 0 };
    rro_vertex* verts = NULL;
    size_t nv = 0, capv = 0;
    char* line = NULL;
    size_t cap = 0;
    ssize_t len;
    while ((len = getline(&line, &cap, f)) >= 0) {          /* Mesh.cpp:14 std::getline */
        if (len > 0 && line[len - 1] == '\n') line[len - 1] = 0;
        float x, y, z, u, v;
        int a[3], b[3], c[3];
        if (sscanf(line, "v %f %f %f", &x, &y, &z) == 3) {             /* :15 */
            float t[3] = { x, y, z };
            fvec_push(&locs, t, 3);
        } else if (sscanf(line, "vt %f %f", &u, &v) == 2) {            /* :17 */
            float t[2] = { u, v };
            fvec_push(&uvs, t, 2);
        } else if (sscanf(line, "vn %f %f %f", &x, &y, &z) == 3) {     /* :19 */
            float t[3] = { x, y, z };
            fvec_push(&norms, t, 3);
        } else if (sscanf(line, "f %d/%d/%d %d/%d/%d %d/%d/%d",        /* :21-25 */
                          &a[0], &b[0], &c[0], &a[1], &b[1], &c[1], &a[2], &b[2], &c[2]) == 9) {
            for (int i = 0; i < 3; ++i) {                              /* :26-33 */
                if (nv == capv) {
                    capv = capv ? capv * 2 : 1024;
                    verts = (rro_vertex*)realloc(verts, capv * sizeof(rro_vertex));
                }
                rro_vertex vx;
                memset(&vx, 0, sizeof vx);
                /* the reference does no bounds checks; the oracle refuses instead of reading wild */
                if (a[i] < 1 || (size_t)(3 * a[i]) > locs.n || b[i] < 1 || (size_t)(2 * b[i]) > uvs.n ||
                    c[i] < 1 || (size_t)(3 * c[i]) > norms.n) {
                    free(line); free(locs.d); free(uvs.d); free(norms.d); free(verts); fclose(f);
                    return 0;
                }
                memcpy(vx.position, &locs.d[3 * (a[i] - 1)], sizeof(float) * 3);
                memcpy(vx.uv, &uvs.d[2 * (b[i] - 1)], sizeof(float) * 2);
                memcpy(vx.norm, &norms.d[3 * (c[i] - 1)], sizeof(float) * 3);
                verts[nv++] = vx;
            }
        }
    }
    free(line);
    free(locs.d); free(uvs.d); free(norms.d);
    fclose(f);
    uint32_t* idx = (uint32_t*)malloc((nv ? nv : 1) * sizeof(uint32_t));
    for (size_t i = 0; i < nv; ++i) idx[i] = (uint32_t)i;   /* :31 indices.push_back(verts.size()) */
    if (!verts) verts = (rro_vertex*)malloc(sizeof(rro_vertex));
    *verts_out = verts; *n_verts = (uint32_t)nv;
    *idx_out = idx; *n_idx = (uint32_t)nv;
    return 1;
}

/* ------------------------------------------------------- camera (RefractionDemo.cpp:559-566)
 * DirectXMath is not in the tree; its published (MIT) algorithms are restated:
 * XMScalarSinCos minimax polynomials, XMMatrixPerspectiveFovLH, XMMatrixLookToLH,
 * XMMatrixMultiply (SSE summation order), cofactor inverse. */
static void dx_scalar_sincos(float* s, float* c, float value)
{
    float q = 0.159154943f * value;                         /* XM_1DIV2PI */
    q = (value >= 0.0f) ? (float)((int)(q + 0.5f)) : (float)((int)(q - 0.5f));
    float y = value - 6.283185307f * q;                     /* XM_2PI */
    float sign;
    if (y > 1.570796327f) { y = 3.141592654f - y; sign = -1.0f; }
    else if (y < -1.570796327f) { y = -3.141592654f - y; sign = -1.0f; }
    else sign = 1.0f;
    float y2 = y * y;
    *s = (((((-2.3889859e-08f * y2 + 2.7525562e-06f) * y2 - 0.00019840874f) * y2 + 0.0083333310f) * y2
           - 0.16666667f) * y2 + 1.0f) * y;
    float p = ((((-2.6051615e-07f * y2 + 2.4760495e-05f) * y2 - 0.0013888378f) * y2 + 0.041666638f) * y2
               - 0.5f) * y2 + 1.0f;
    *c = sign * p;
}

typedef struct { float m[4][4]; } m44;

static m44 m44_mul(const m44* a, const m44* b)
{
    m44 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float x = a->m[i][0] * b->m[0][j];
            float y = a->m[i][1] * b->m[1][j];
            float z = a->m[i][2] * b->m[2][j];
            float w = a->m[i][3] * b->m[3][j];
            r.m[i][j] = (x + z) + (y + w);
        }
    return r;
}

static m44 m44_inverse(const m44* M)
{
    const float* m = &M->m[0][0];
    float inv[16];
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    float rdet = 1.0f / det;
    m44 r;
    for (int i = 0; i < 16; ++i) (&r.m[0][0])[i] = inv[i] * rdet;
    return r;
}

void rro_camera(float angle, float fov_y, float aspect, float zn, float zf,
                float proj_inv[16], float camera_loc[4])
{
    /* :559 XMMatrixPerspectiveFovLH */
    float sf, cf;
    dx_scalar_sincos(&sf, &cf, 0.5f * fov_y);
    float hh = cf / sf;
    float ww = hh / aspect;
    float range = zf / (zf - zn);
    m44 proj;
    memset(&proj, 0, sizeof proj);
    proj.m[0][0] = ww; proj.m[1][1] = hh; proj.m[2][2] = range; proj.m[2][3] = 1.0f;
    proj.m[3][2] = -range * zn;
    /* :560 camera_loc = {5cos a, 0, 5 sin a, 1} (CRT cosf/sinf) */
    camera_loc[0] = 5 * cosf(angle); camera_loc[1] = 0.0f; camera_loc[2] = 5 * sinf(angle); camera_loc[3] = 1.0f;
    /* :561 XMMatrixTranslationFromVector */
    m44 world;
    memset(&world, 0, sizeof world);
    world.m[0][0] = world.m[1][1] = world.m[2][2] = world.m[3][3] = 1.0f;
    world.m[3][0] = camera_loc[0]; world.m[3][1] = camera_loc[1]; world.m[3][2] = camera_loc[2];
    /* :562 XMMatrixLookAtLH(eye=(cos(-a),0,sin(-a)), focus 0, up +Y) */
    v3 eye = v3_make(cosf(-angle), 0.0f, sinf(-angle));
    v3 up = v3_make(0.0f, 1.0f, 0.0f);
    v3 eyedir = v3_sub(v3_make(0, 0, 0), eye);
    v3 r2 = v3_normalize(eyedir);
    v3 r0 = v3_normalize(v3_cross(up, r2));
    v3 r1 = v3_cross(r2, r0);
    v3 ne = v3_neg(eye);
    float d0 = v3_dot(r0, ne), d1 = v3_dot(r1, ne), d2 = v3_dot(r2, ne);
    m44 view;
    view.m[0][0] = r0.x; view.m[0][1] = r1.x; view.m[0][2] = r2.x; view.m[0][3] = 0.0f;
    view.m[1][0] = r0.y; view.m[1][1] = r1.y; view.m[1][2] = r2.y; view.m[1][3] = 0.0f;
    view.m[2][0] = r0.z; view.m[2][1] = r1.z; view.m[2][2] = r2.z; view.m[2][3] = 0.0f;
    view.m[3][0] = d0;   view.m[3][1] = d1;   view.m[3][2] = d2;   view.m[3][3] = 1.0f;
    /* :563-565 proj*world*view, inverse */
    m44 pw = m44_mul(&proj, &world);
    m44 pwv = m44_mul(&pw, &view);
    m44 inv = m44_inverse(&pwv);
    memcpy(proj_inv, &inv.m[0][0], 64);
}

/* RayTracing.hlsl:27-40.  The CB bytes are CPU row-major, HLSL reads them
 * column-major, so mul(float4(s,0,1), proj_inv) == M_cpu * (sx,sy,0,1)^T. */
void rro_generate_camera_ray(const float M[16], const float camera_loc[4],
                             uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                             float origin[3], float dir[3])
{
    float px = (float)x + 0.5f, py = (float)y + 0.5f;                 /* :29 */
    float sx = px / (float)w * 2.0f - 1.0f;                           /* :30 */
    float sy = py / (float)h * 2.0f - 1.0f;
    sy = -sy;                                                         /* :33 */
    v3 R;
    R.x = (sx * M[0] + sy * M[1]) + M[3];                             /* :35 */
    R.y = (sx * M[4] + sy * M[5]) + M[7];
    R.z = (sx * M[8] + sy * M[9]) + M[11];
    origin[0] = camera_loc[0]; origin[1] = camera_loc[1]; origin[2] = camera_loc[2];  /* :38 */
    R = v3_normalize(R);                                              /* :39 */
    dir[0] = R.x; dir[1] = R.y; dir[2] = R.z;
}

/* ------------------------------------------------------------------ scene */
typedef struct { v3 lo, hi; } aabb;

typedef struct {
    aabb     box;
    int32_t  left, right;     /* internal: child node ids; leaf: left = -1 */
    uint32_t first, count;    /* leaf: range in prim_order */
} cpu_node;

typedef struct {
    rro_vertex* verts; uint32_t n_verts;
    uint32_t*   idx;   uint32_t n_idx;
    uint32_t    n_tris;
    v3 *v0, *e1, *e2;         /* per triangle, from Vertices[Indices[3p+k]].position */
    cpu_node* nodes; uint32_t n_nodes;
    uint32_t* prim_order;
    aabb bounds;
} cpu_mesh;

typedef struct {
    rro_instance desc;
    float inv[12];            /* world -> object 3x4 */
    int   identity;
    aabb  world_box;
} cpu_inst;

struct rro_scene {
    cpu_mesh* meshes; uint32_t n_meshes;
    cpu_inst* insts;  uint32_t n_insts;
    float* env; int env_w, env_h;
};

static inline aabb aabb_empty(void)
{
    aabb b = { { INFINITY, INFINITY, INFINITY }, { -INFINITY, -INFINITY, -INFINITY } };
    return b;
}
static inline void aabb_grow(aabb* b, v3 p)
{
    b->lo.x = fminf(b->lo.x, p.x); b->lo.y = fminf(b->lo.y, p.y); b->lo.z = fminf(b->lo.z, p.z);
    b->hi.x = fmaxf(b->hi.x, p.x); b->hi.y = fmaxf(b->hi.y, p.y); b->hi.z = fmaxf(b->hi.z, p.z);
}

rro_scene* rro_scene_create(void) { return (rro_scene*)calloc(1, sizeof(rro_scene)); }

static void mesh_free(cpu_mesh* m)
{
    free(m->verts); free(m->idx); free(m->v0); free(m->e1); free(m->e2); free(m->nodes); free(m->prim_order);
}

void rro_scene_destroy(rro_scene* s)
{
    if (!s) return;
    for (uint32_t i = 0; i < s->n_meshes; ++i) mesh_free(&s->meshes[i]);
    free(s->meshes); free(s->insts); free(s->env); free(s);
}

/* --- CPU BVH: top-down object-median split (deliberately NOT the GPU's LBVH) */
typedef struct { const cpu_mesh* m; int axis; } sort_ctx;
static float tri_centroid_axis(const cpu_mesh* m, uint32_t p, int axis)
{
    v3 a = m->v0[p], b = v3_add(m->v0[p], m->e1[p]), c = v3_add(m->v0[p], m->e2[p]);
    float lo, hi;
    if (axis == 0) { lo = fminf(a.x, fminf(b.x, c.x)); hi = fmaxf(a.x, fmaxf(b.x, c.x)); }
    else if (axis == 1) { lo = fminf(a.y, fminf(b.y, c.y)); hi = fmaxf(a.y, fmaxf(b.y, c.y)); }
    else { lo = fminf(a.z, fminf(b.z, c.z)); hi = fmaxf(a.z, fmaxf(b.z, c.z)); }
    return 0.5f * (lo + hi);
}
static int cmp_centroid(const void* pa, const void* pb, void* vctx)
{
    const sort_ctx* c = (const sort_ctx*)vctx;
    uint32_t a = *(const uint32_t*)pa, b = *(const uint32_t*)pb;
    float ca = tri_centroid_axis(c->m, a, c->axis), cb = tri_centroid_axis(c->m, b, c->axis);
    if (ca < cb) return -1;
    if (ca > cb) return 1;
    return a < b ? -1 : (a > b ? 1 : 0);
}

static aabb tri_box(const cpu_mesh* m, uint32_t p)
{
    /* exact vertex positions, not v0+e (which rounds) */
    aabb b = aabb_empty();
    for (int k = 0; k < 3; ++k) {
        const float* q = m->verts[m->idx[3 * p + k]].position;
        aabb_grow(&b, v3_make(q[0], q[1], q[2]));
    }
    return b;
}

static int32_t bvh_build_rec(cpu_mesh* m, uint32_t first, uint32_t count)
{
    int32_t id = (int32_t)m->n_nodes++;
    cpu_node* n = &m->nodes[id];
    aabb box = aabb_empty(), cbox = aabb_empty();
    for (uint32_t i = 0; i < count; ++i) {
        uint32_t p = m->prim_order[first + i];
        aabb tb = tri_box(m, p);
        aabb_grow(&box, tb.lo); aabb_grow(&box, tb.hi);
        aabb_grow(&cbox, v3_make(tri_centroid_axis(m, p, 0), tri_centroid_axis(m, p, 1), tri_centroid_axis(m, p, 2)));
    }
    n->box = box; n->first = first; n->count = count; n->left = -1; n->right = -1;
    if (count <= 4) return id;
    float ex = cbox.hi.x - cbox.lo.x, ey = cbox.hi.y - cbox.lo.y, ez = cbox.hi.z - cbox.lo.z;
    int axis = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);
    sort_ctx ctx = { m, axis };
    qsort_r(m->prim_order + first, count, sizeof(uint32_t), cmp_centroid, &ctx);
    uint32_t half = count / 2;
    int32_t l = bvh_build_rec(m, first, half);
    int32_t r = bvh_build_rec(m, first + half, count - half);
    m->nodes[id].left = l; m->nodes[id].right = r;   /* m->nodes does not move: preallocated */
    return id;
}

int rro_scene_add_mesh(rro_scene* s, const rro_vertex* verts, uint32_t n_verts,
                       const uint32_t* indices, uint32_t n_indices)
{
    for (uint32_t i = 0; i < n_indices; ++i) if (indices[i] >= n_verts) return -1;
    s->meshes = (cpu_mesh*)realloc(s->meshes, (s->n_meshes + 1) * sizeof(cpu_mesh));
    cpu_mesh* m = &s->meshes[s->n_meshes];
    memset(m, 0, sizeof *m);
    m->n_verts = n_verts; m->n_idx = n_indices; m->n_tris = n_indices / 3;
    m->verts = (rro_vertex*)malloc((n_verts ? n_verts : 1) * sizeof(rro_vertex));
    memcpy(m->verts, verts, n_verts * sizeof(rro_vertex));
    m->idx = (uint32_t*)malloc((n_indices ? n_indices : 1) * sizeof(uint32_t));
    memcpy(m->idx, indices, n_indices * sizeof(uint32_t));
    uint32_t T = m->n_tris;
    m->v0 = (v3*)malloc((T ? T : 1) * sizeof(v3));
    m->e1 = (v3*)malloc((T ? T : 1) * sizeof(v3));
    m->e2 = (v3*)malloc((T ? T : 1) * sizeof(v3));
    m->bounds = aabb_empty();
    for (uint32_t p = 0; p < T; ++p) {
        const float* a = m->verts[m->idx[3 * p + 0]].position;
        const float* b = m->verts[m->idx[3 * p + 1]].position;
        const float* c = m->verts[m->idx[3 * p + 2]].position;
        v3 A = v3_make(a[0], a[1], a[2]), B = v3_make(b[0], b[1], b[2]), C = v3_make(c[0], c[1], c[2]);
        m->v0[p] = A; m->e1[p] = v3_sub(B, A); m->e2[p] = v3_sub(C, A);
        aabb_grow(&m->bounds, A); aabb_grow(&m->bounds, B); aabb_grow(&m->bounds, C);
    }
    m->prim_order = (uint32_t*)malloc((T ? T : 1) * sizeof(uint32_t));
    for (uint32_t p = 0; p < T; ++p) m->prim_order[p] = p;
    m->nodes = (cpu_node*)malloc((2 * (size_t)T + 1) * sizeof(cpu_node));
    m->n_nodes = 0;
    if (T) bvh_build_rec(m, 0, T);
    return (int)s->n_meshes++;
}

/* world->object inverse of a 3x4 affine (adjugate / det, fixed operation order) */
static void affine_inverse(const float t[12], float inv[12])
{
    float a = t[0], b = t[1], c = t[2], d = t[4], e = t[5], f = t[6], g = t[8], h = t[9], i = t[10];
    float c00 = e * i - f * h, c01 = f * g - d * i, c02 = d * h - e * g;
    float det = (a * c00 + b * c01) + c * c02;
    float r = 1.0f / det;
    inv[0] = c00 * r;           inv[1] = (c * h - b * i) * r; inv[2] = (b * f - c * e) * r;
    inv[4] = c01 * r;           inv[5] = (a * i - c * g) * r; inv[6] = (c * d - a * f) * r;
    inv[8] = c02 * r;           inv[9] = (b * g - a * h) * r; inv[10] = (a * e - b * d) * r;
    float tx = t[3], ty = t[7], tz = t[11];
    inv[3]  = -((inv[0] * tx + inv[1] * ty) + inv[2] * tz);
    inv[7]  = -((inv[4] * tx + inv[5] * ty) + inv[6] * tz);
    inv[11] = -((inv[8] * tx + inv[9] * ty) + inv[10] * tz);
}

static inline v3 xform_point(const float m[12], v3 p)
{
    return v3_make(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3],
                   ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
                   ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]);
}
static inline v3 xform_dir(const float m[12], v3 p)
{
    return v3_make((m[0] * p.x + m[1] * p.y) + m[2] * p.z,
                   (m[4] * p.x + m[5] * p.y) + m[6] * p.z,
                   (m[8] * p.x + m[9] * p.y) + m[10] * p.z);
}

static void inst_prepare(const rro_scene* s, cpu_inst* ci)
{
    static const float ident[12] = { 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0 };
    ci->identity = memcmp(ci->desc.transform, ident, sizeof ident) == 0;
    if (ci->identity) memcpy(ci->inv, ident, sizeof ident);
    else affine_inverse(ci->desc.transform, ci->inv);
    const cpu_mesh* m = &s->meshes[ci->desc.blas];
    ci->world_box = aabb_empty();
    for (int k = 0; k < 8; ++k) {
        v3 p = v3_make((k & 1) ? m->bounds.hi.x : m->bounds.lo.x,
                       (k & 2) ? m->bounds.hi.y : m->bounds.lo.y,
                       (k & 4) ? m->bounds.hi.z : m->bounds.lo.z);
        aabb_grow(&ci->world_box, ci->identity ? p : xform_point(ci->desc.transform, p));
    }
}

int rro_scene_set_instances(rro_scene* s, const rro_instance* inst, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i) if (inst[i].blas >= s->n_meshes) return -1;
    free(s->insts);
    s->insts = (cpu_inst*)calloc(n ? n : 1, sizeof(cpu_inst));
    s->n_insts = n;
    for (uint32_t i = 0; i < n; ++i) {
        s->insts[i].desc = inst[i];
        inst_prepare(s, &s->insts[i]);
    }
    return 0;
}

static void scene_default_instance(rro_scene* s)
{
    /* RefractionDemo.cpp:324-334: identity transform, mask 1, flags 0, the one BLAS */
    rro_instance d;
    memset(&d, 0, sizeof d);
    d.transform[0] = d.transform[5] = d.transform[10] = 1.0f;
    d.id_mask = 1u << 24;
    d.blas = 0;
    rro_scene_set_instances(s, &d, 1);
}

int rro_scene_set_envmap(rro_scene* s, const float* rgb, int w, int h)
{
    if (w <= 0 || h <= 0) return -1;
    free(s->env);
    s->env = (float*)malloc((size_t)w * h * 3 * sizeof(float));
    memcpy(s->env, rgb, (size_t)w * h * 3 * sizeof(float));
    s->env_w = w; s->env_h = h;
    return 0;
}

/* ------------------------------------------------------------------ TraceRay
 * Closest hit, tmin < t < tmax exclusive, per-ray face culling.
 * Moller-Trumbore in scaled form: barycentric rejections are tested on the
 * det-scaled numerators so that only accepted candidates pay a division.
 *   front-facing  <=>  dot(cross(e1,e2), D) < 0  <=>  det = dot(e1, cross(D,e2)) > 0
 * Equal-t ties resolve to the lower (instance, primitive) pair so the result is
 * independent of test order (brute force == any BVH). */
#define RRO_CULL_BACK  0x10u
#define RRO_CULL_FRONT 0x20u

typedef struct {
    float    t, U, V, ad;
    uint32_t prim, inst;
    int      hit;
} hit_rec;

static inline void tri_test(const cpu_mesh* m, uint32_t p, v3 O, v3 D, float tmin, uint32_t flags,
                            uint32_t inst, hit_rec* best, rro_stats* st)
{
    if (st) st->tri_tests++;
    v3 e1 = m->e1[p], e2 = m->e2[p];
    v3 pv = v3_cross(D, e2);
    float det = v3_dot(e1, pv);
    if (flags & RRO_CULL_BACK) { if (!(det > 0.0f)) return; }
    else if (flags & RRO_CULL_FRONT) { if (!(det < 0.0f)) return; }
    else if (!(det != 0.0f)) return;
    v3 tv = v3_sub(O, m->v0[p]);
    float U = v3_dot(tv, pv);
    v3 qv = v3_cross(tv, e1);
    float V = v3_dot(D, qv);
    float T = v3_dot(e2, qv);
    float ad = det;
    if (det < 0.0f) { U = -U; V = -V; T = -T; ad = -det; }
    if (U < 0.0f || V < 0.0f || U + V > ad) return;
    float t = T / ad;
    if (!(t > tmin)) return;
    if (t < best->t || (t == best->t && best->hit &&
                        (inst < best->inst || (inst == best->inst && p < best->prim)))) {
        best->t = t; best->U = U; best->V = V; best->ad = ad; best->prim = p; best->inst = inst; best->hit = 1;
    }
}

/* conservative slab test; only has to never reject a box whose triangle would be accepted */
/* pad: boxes are grown by 1e-5 of the scene/origin magnitude -- the fp32 triangle test accepts points a
 * few ulps outside a triangle's edge, and a ray running along the symmetry plane of a mirrored mesh
 * hits exactly those edges; without the growth the BVH would cull what brute force accepts. */
static inline int box_test(const aabb* b, v3 O, v3 inv, float tmin, float tmax, float pad)
{
    float t0x = (b->lo.x - pad - O.x) * inv.x, t1x = (b->hi.x + pad - O.x) * inv.x;
    float t0y = (b->lo.y - pad - O.y) * inv.y, t1y = (b->hi.y + pad - O.y) * inv.y;
    float t0z = (b->lo.z - pad - O.z) * inv.z, t1z = (b->hi.z + pad - O.z) * inv.z;
    float tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    return tn <= tf * 1.0000004f;
}

static inline float safe_rcp(float d)
{
    if (fabsf(d) < 1e-20f) d = copysignf(1e-20f, d);
    return 1.0f / d;
}

static void trace_mesh(const cpu_mesh* m, v3 O, v3 D, float tmin, uint32_t flags, uint32_t inst,
                       int use_bvh, hit_rec* best, rro_stats* st)
{
    if (!use_bvh) {
        for (uint32_t p = 0; p < m->n_tris; ++p) tri_test(m, p, O, D, tmin, flags, inst, best, st);
        return;
    }
    if (!m->n_nodes) return;
    v3 inv = v3_make(safe_rcp(D.x), safe_rcp(D.y), safe_rcp(D.z));
    const aabb* rb = &m->nodes[0].box;
    float scale = fmaxf(fmaxf(fmaxf(fabsf(rb->lo.x), fabsf(rb->hi.x)), fmaxf(fabsf(rb->lo.y), fabsf(rb->hi.y))),
                        fmaxf(fabsf(rb->lo.z), fabsf(rb->hi.z)));
    float pad = 1e-5f * fmaxf(fmaxf(fabsf(O.x), fabsf(O.y)), fmaxf(fabsf(O.z), scale));
    int32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    while (sp) {
        const cpu_node* n = &m->nodes[stack[--sp]];
        if (st) st->node_visits++;
        if (!box_test(&n->box, O, inv, tmin, best->t, pad)) continue;
        if (n->left < 0) {
            for (uint32_t i = 0; i < n->count; ++i)
                tri_test(m, m->prim_order[n->first + i], O, D, tmin, flags, inst, best, st);
        } else {
            stack[sp++] = n->left;
            stack[sp++] = n->right;
        }
    }
}

static void trace_scene(const rro_scene* s, v3 O, v3 D, float tmin, float tmax, uint32_t flags,
                        int use_bvh, hit_rec* best, rro_stats* st)
{
    best->t = tmax; best->hit = 0; best->prim = 0; best->inst = 0; best->U = best->V = 0; best->ad = 1;
    for (uint32_t i = 0; i < s->n_insts; ++i) {
        const cpu_inst* ci = &s->insts[i];
        if (((ci->desc.id_mask >> 24) & 0xffu) == 0) continue;      /* InstanceInclusionMask 0xff */
        uint32_t iflags = ci->desc.hitgroup_flags >> 24;
        uint32_t f = flags;
        if (iflags & 0x1u) f &= ~(RRO_CULL_BACK | RRO_CULL_FRONT);  /* TRIANGLE_CULL_DISABLE */
        else if (iflags & 0x2u) {                                    /* TRIANGLE_FRONT_COUNTERCLOCKWISE */
            if (f & RRO_CULL_BACK) f = (f & ~RRO_CULL_BACK) | RRO_CULL_FRONT;
            else if (f & RRO_CULL_FRONT) f = (f & ~RRO_CULL_FRONT) | RRO_CULL_BACK;
        }
        const cpu_mesh* m = &s->meshes[ci->desc.blas];
        if (ci->identity) {
            trace_mesh(m, O, D, tmin, f, i, use_bvh, best, st);
        } else {
            if (use_bvh) {
                v3 inv = v3_make(safe_rcp(D.x), safe_rcp(D.y), safe_rcp(D.z));
                const aabb* wb = &ci->world_box;
                float sc = fmaxf(fmaxf(fmaxf(fabsf(wb->lo.x), fabsf(wb->hi.x)), fmaxf(fabsf(wb->lo.y), fabsf(wb->hi.y))),
                                 fmaxf(fabsf(wb->lo.z), fabsf(wb->hi.z)));
                float pad = 1e-4f * fmaxf(fmaxf(fabsf(O.x), fabsf(O.y)), fmaxf(fabsf(O.z), sc));
                if (!box_test(wb, O, inv, tmin, best->t, pad)) continue;
            }
            v3 Oo = xform_point(ci->inv, O), Do = xform_dir(ci->inv, D);
            trace_mesh(m, Oo, Do, tmin, f, i, use_bvh, best, st);
        }
    }
}

void rro_trace(const rro_scene* cs, const float origin[3], const float dir[3],
               float tmin, float tmax, uint32_t flags, int use_bvh, rro_hit* out)
{
    rro_scene* s = (rro_scene*)cs;
    if (!s->insts) scene_default_instance(s);
    hit_rec b;
    trace_scene(s, v3_make(origin[0], origin[1], origin[2]), v3_make(dir[0], dir[1], dir[2]),
                tmin, tmax, flags, use_bvh, &b, NULL);
    out->hit = b.hit; out->prim = b.prim; out->inst = b.inst;
    out->t = b.hit ? b.t : tmax;
    out->u = b.hit ? b.U / b.ad : 0.0f;
    out->v = b.hit ? b.V / b.ad : 0.0f;
}

/* ------------------------------------------------------------------ Miss (hlsl:127-137) */
static v3 env_lookup(const rro_scene* s, v3 r, int use_libm)
{
    if (!s->env) return v3_make(0, 0, 0);
    float at = use_libm ? atan2f(r.x, r.z) : rro_atan2f(r.x, r.z);
    float ac = use_libm ? acosf(r.y) : rro_acosf(r.y);
    float theta = (float)s->env_w * (at / 3.14159f + 1.0f) / 2;        /* :133 */
    float phi = (float)s->env_h * (ac / 3.14159f);                     /* :134 */
    uint32_t ix = rro_ftou(theta), iy = rro_ftou(phi);                 /* :135 operator[] : ftou, OOB -> 0 */
    if (ix >= (uint32_t)s->env_w || iy >= (uint32_t)s->env_h) return v3_make(0, 0, 0);
    const float* px = &s->env[((size_t)iy * s->env_w + ix) * 3];
    return v3_make(px[0], px[1], px[2]);                               /* mask == (1,1,1): mask*texel == texel */
}

void rro_env_lookup(const rro_scene* s, const float dir[3], int use_libm, float rgb[3])
{
    v3 c = env_lookup(s, v3_make(dir[0], dir[1], dir[2]), use_libm);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* ------------------------------------------------------------------ shading */
typedef struct {
    const rro_scene* s;
    const rro_params* p;
    rro_stats* st;
    uint32_t pixel_rays;
    /* accum_mode 1: path-weight accumulator */
    v3 acc;
} shade_ctx;

static inline v3 reflect_ray(v3 I, v3 N)                               /* hlsl:66-68 */
{
    float k = 2.0f * v3_dot(N, I);
    return v3_make(I.x - k * N.x, I.y - k * N.y, I.z - k * N.z);
}

static inline int refract_ray(v3* R, v3 I, v3 N, float eta)            /* hlsl:70-76 */
{
    float c = v3_dot(N, I);
    float k = 1.0f - (eta * eta) * (1.0f - c * c);
    if (k < 0.0f) return 0;
    float a = eta * c + sqrtf(k);
    v3 r = v3_make(eta * I.x - a * N.x, eta * I.y - a * N.y, eta * I.z - a * N.z);
    *R = v3_normalize(r);
    return 1;
}

/* One TraceRay + the shader it invokes.  Returns payload.color (accum_mode 0);
 * in accum_mode 1 adds weight*texel into ctx->acc instead. */
static v3 trace_and_shade(shade_ctx* c, v3 O, v3 D, float tmin, float tmax, uint32_t flags,
                          int outside, uint32_t count, float weight)
{
    rro_stats* st = c->st;
    st->rays++; c->pixel_rays++;
    if (count < 32) st->rays_per_level[count]++;
    if (count == 0) st->primary++; else st->secondary++;
    hit_rec h;
    trace_scene(c->s, O, D, tmin, tmax, flags, c->p->use_bvh, &h, st);
    if (!h.hit) {                                                      /* Miss */
        st->misses++;
        v3 e = env_lookup(c->s, D, c->p->use_libm);
        if (c->p->accum_mode == 1) {
            c->acc.x = fmaf(weight, e.x, c->acc.x);
            c->acc.y = fmaf(weight, e.y, c->acc.y);
            c->acc.z = fmaf(weight, e.z, c->acc.z);
        }
        return e;
    }
    st->hits++;
    /* ClosestHit, hlsl:79-125.  payload.color of a secondary ray is uninitialised in the
     * reference (hlsl:102,117); the oracle DEFINES it as 0. */
    v3 color = v3_make(0, 0, 0);
    if (!(count < (uint32_t)c->p->max_refract)) {                      /* :82 */
        st->terminal_hits++;
        return color;
    }
    const cpu_inst* ci = &c->s->insts[h.inst];
    const cpu_mesh* m = &c->s->meshes[ci->desc.blas];
    float u = h.U / h.ad, v = h.V / h.ad;
    const float* nA = m->verts[m->idx[3 * h.prim + 0]].norm;           /* :83-85 */
    const float* nB = m->verts[m->idx[3 * h.prim + 1]].norm;
    const float* nC = m->verts[m->idx[3 * h.prim + 2]].norm;
    v3 A = v3_make(nA[0], nA[1], nA[2]), B = v3_make(nB[0], nB[1], nB[2]), C = v3_make(nC[0], nC[1], nC[2]);
    v3 BA = v3_sub(B, A), CA = v3_sub(C, A);
    v3 Nr = v3_make(fmaf(v, CA.x, fmaf(u, BA.x, A.x)),                 /* :86 */
                    fmaf(v, CA.y, fmaf(u, BA.y, A.y)),
                    fmaf(v, CA.z, fmaf(u, BA.z, A.z)));
    if (!ci->identity) {
        /* extension (the reference's only instance is identity): normals go to world space by
         * the inverse transpose of the object->world 3x3 */
        const float* w = ci->inv;
        Nr = v3_make((w[0] * Nr.x + w[4] * Nr.y) + w[8] * Nr.z,
                     (w[1] * Nr.x + w[5] * Nr.y) + w[9] * Nr.z,
                     (w[2] * Nr.x + w[6] * Nr.y) + w[10] * Nr.z);
    }
    v3 N = v3_normalize(Nr);
    v3 X = v3_make(fmaf(h.t, D.x, O.x), fmaf(h.t, D.y, O.y), fmaf(h.t, D.z, O.z));   /* :88 */
    v3 Nf = outside ? N : v3_neg(N);
    const float R0 = (0.2f / 2.2f) * (0.2f / 2.2f);                    /* :92 */
    float b = 1.0f - v3_dot(D, Nf);                                    /* :93 pow(b,5) as b*b*b*b*b */
    float b2 = b * b, b4 = b2 * b2;
    float R = (R0 * (1.0f - R0)) * (b4 * b);
    float eta = outside ? (1.0f / c->p->ior) : c->p->ior;              /* :95 */
    v3 d1;
    if (refract_ray(&d1, D, Nf, eta)) {
        int out2 = !outside;                                           /* :105 */
        float wr = 1.0f - R;
        v3 c2 = trace_and_shade(c, X, d1, c->p->tmin_secondary, c->p->tmax_secondary,
                                out2 ? RRO_CULL_BACK : RRO_CULL_FRONT, out2, count + 1, weight * wr);
        color.x += wr * c2.x; color.y += wr * c2.y; color.z += wr * c2.z;        /* :107 */
    } else {
        st->tir++;
    }
    if (count < (uint32_t)c->p->max_reflect) {                         /* :110 */
        v3 d2 = v3_normalize(reflect_ray(D, Nf));                      /* :113 */
        v3 c2 = trace_and_shade(c, X, d2, c->p->tmin_secondary, c->p->tmax_secondary,
                                outside ? RRO_CULL_BACK : RRO_CULL_FRONT, outside, count + 1, weight * R);
        color.x += R * c2.x; color.y += R * c2.y; color.z += R * c2.z;            /* :122 */
    }
    return color;
}

/* ------------------------------------------------------------------ DispatchRays */
typedef struct {
    const rro_scene* s;
    const float* M; const float* cam;
    uint32_t w, h;
    const rro_params* p;
    uint32_t x0, y0, x1, y1, tile_w, tile_h, tile_rank, tile_world;
    float* out_rgb; uint8_t* out_rgba8; uint16_t* out_raycount;
    uint32_t next_row;
    rro_stats stats;
    pthread_mutex_t mu;
} job;

static void* worker(void* arg)
{
    job* j = (job*)arg;
    rro_stats st;
    memset(&st, 0, sizeof st);
    uint32_t tiles_x = (j->w + j->tile_w - 1) / j->tile_w;
    for (;;) {
        uint32_t y = __atomic_fetch_add(&j->next_row, 1u, __ATOMIC_RELAXED);
        if (y >= j->y1) break;
        for (uint32_t x = j->x0; x < j->x1; ++x) {
            if (j->tile_world > 1) {
                uint32_t tile = (y / j->tile_h) * tiles_x + (x / j->tile_w);
                if (tile % j->tile_world != j->tile_rank) continue;
            }
            float o[3], d[3];
            rro_generate_camera_ray(j->M, j->cam, x, y, j->w, j->h, o, d);
            shade_ctx c = { j->s, j->p, &st, 0, { 0, 0, 0 } };
            /* RayGen hlsl:49-60: payload {0, 1, outside, 0}, CULL_BACK */
            v3 col = trace_and_shade(&c, v3_make(o[0], o[1], o[2]), v3_make(d[0], d[1], d[2]),
                                     j->p->tmin_primary, j->p->tmax_primary, RRO_CULL_BACK, 1, 0, 1.0f);
            if (j->p->accum_mode == 1) col = c.acc;
            if (c.pixel_rays > st.max_rays_per_pixel) st.max_rays_per_pixel = c.pixel_rays;
            size_t pix = (size_t)y * j->w + x;
            if (j->out_rgb) { j->out_rgb[pix * 3 + 0] = col.x; j->out_rgb[pix * 3 + 1] = col.y; j->out_rgb[pix * 3 + 2] = col.z; }
            if (j->out_rgba8) {                                        /* hlsl:62 float4(color,1) -> UNORM8 */
                v3 cq = col;
                if (j->p->tonemap) { cq.x = rro_reinhard(col.x); cq.y = rro_reinhard(col.y); cq.z = rro_reinhard(col.z); }
                j->out_rgba8[pix * 4 + 0] = rro_unorm8(cq.x);
                j->out_rgba8[pix * 4 + 1] = rro_unorm8(cq.y);
                j->out_rgba8[pix * 4 + 2] = rro_unorm8(cq.z);
                j->out_rgba8[pix * 4 + 3] = 255;
            }
            if (j->out_raycount) j->out_raycount[pix] = (uint16_t)(c.pixel_rays > 65535 ? 65535 : c.pixel_rays);
        }
    }
    pthread_mutex_lock(&j->mu);
    rro_stats* t = &j->stats;
    t->rays += st.rays; t->primary += st.primary; t->secondary += st.secondary; t->hits += st.hits;
    t->misses += st.misses; t->terminal_hits += st.terminal_hits; t->tir += st.tir;
    t->tri_tests += st.tri_tests; t->node_visits += st.node_visits;
    if (st.max_rays_per_pixel > t->max_rays_per_pixel) t->max_rays_per_pixel = st.max_rays_per_pixel;
    for (int i = 0; i < 32; ++i) t->rays_per_level[i] += st.rays_per_level[i];
    pthread_mutex_unlock(&j->mu);
    return NULL;
}

int rro_render(const rro_scene* cs, const float proj_inv[16], const float camera_loc[4],
               uint32_t w, uint32_t h, const rro_params* p,
               uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
               uint32_t tile_w, uint32_t tile_h, uint32_t tile_rank, uint32_t tile_world,
               int n_threads, float* out_rgb, uint8_t* out_rgba8, uint16_t* out_raycount,
               rro_stats* stats)
{
    rro_scene* s = (rro_scene*)cs;
    if (!s || !s->n_meshes || !p || !w || !h) return -1;
    if (!s->insts) scene_default_instance(s);
    if (x1 > w) x1 = w;
    if (y1 > h) y1 = h;
    if (tile_w == 0) tile_w = 32;
    if (tile_h == 0) tile_h = 32;
    if (tile_world == 0) tile_world = 1;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    job j;
    memset(&j, 0, sizeof j);
    j.s = s; j.M = proj_inv; j.cam = camera_loc; j.w = w; j.h = h; j.p = p;
    j.x0 = x0; j.y0 = y0; j.x1 = x1; j.y1 = y1;
    j.tile_w = tile_w; j.tile_h = tile_h; j.tile_rank = tile_rank; j.tile_world = tile_world;
    j.out_rgb = out_rgb; j.out_rgba8 = out_rgba8; j.out_raycount = out_raycount;
    j.next_row = y0;
    pthread_mutex_init(&j.mu, NULL);
    if (n_threads == 1) {
        worker(&j);
    } else {
        pthread_t th[256];
        for (int i = 0; i < n_threads; ++i) pthread_create(&th[i], NULL, worker, &j);
        for (int i = 0; i < n_threads; ++i) pthread_join(th[i], NULL);
    }
    pthread_mutex_destroy(&j.mu);
    if (stats) *stats = j.stats;
    return 0;
}
