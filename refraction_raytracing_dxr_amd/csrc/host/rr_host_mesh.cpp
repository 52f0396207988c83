// rr_host_mesh.cpp -- Wavefront OBJ -> Vertex[]/uint32[] with the observable behaviour of
// Mesh::load (Mesh.cpp:6-37): line oriented; "v", "vt", "vn" records feed three pools; every
// "f a/b/c a/b/c a/b/c" corner is un-indexed into a fresh 32-byte vertex {position, norm, uv}
// and indices are simply 0..3T-1 (Mesh.cpp:31).  Anything else on a line is ignored, a fourth
// corner is dropped, and lines that do not START with the keyword are skipped, exactly as the
// reference's sscanf cascade does.  Differences: out-of-range references fail the load
// (RR_ERR_INVALID_ARGUMENT) instead of reading outside the pools.
#include "../../../include/rrdxr.h"

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

// A cursor with scanf-like primitives: literal characters must match exactly, a blank in the
// pattern skips any run of white space (possibly empty), numbers skip leading white space.
struct Scan {
    const char* p;
    bool lit(char c) { if (*p != c) return false; ++p; return true; }
    void blanks() { while (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\v' || *p == '\f' || *p == '\r') ++p; }
    bool real(float& out)
    {
        blanks();
        char* end = nullptr;
        float v = strtof(p, &end);
        if (end == p) return false;
        p = end; out = v;
        return true;
    }
    bool integer(int& out)
    {
        blanks();
        char* end = nullptr;
        long v = strtol(p, &end, 10);
        if (end == p) return false;
        p = end; out = (int)v;
        return true;
    }
};

bool scan_reals(const char* line, const char* key, int n, float* out)
{
    Scan s{ line };
    for (const char* k = key; *k; ++k) if (!s.lit(*k)) return false;
    s.blanks();
    for (int i = 0; i < n; ++i) if (!s.real(out[i])) return false;
    return true;
}

bool scan_face(const char* line, int ref[9])
{
    Scan s{ line };
    if (!s.lit('f')) return false;
    for (int corner = 0; corner < 3; ++corner) {
        if (!s.integer(ref[corner * 3 + 0]) || !s.lit('/') || !s.integer(ref[corner * 3 + 1]) || !s.lit('/') ||
            !s.integer(ref[corner * 3 + 2]))
            return false;
    }
    return true;
}

} // namespace

// Generic face record for the hardened mode: any number of corners, each "v", "v/vt", "v//vn" or
// "v/vt/vn", indices may be negative (relative to the end of the pools read so far).
struct Corner { long v, vt, vn; };

static bool scan_face_any(const char* line, std::vector<Corner>& out)
{
    out.clear();
    const char* p = line;
    if (*p != 'f' || !(p[1] == ' ' || p[1] == '\t')) return false;
    ++p;
    for (;;) {
        while (*p == ' ' || *p == '\t' || *p == '\r') ++p;
        if (!*p) break;
        Corner c{ 0, 0, 0 };
        char* end = nullptr;
        c.v = strtol(p, &end, 10);
        if (end == p) return false;
        p = end;
        if (*p == '/') {
            ++p;
            if (*p != '/') { c.vt = strtol(p, &end, 10); if (end == p) return false; p = end; }
            if (*p == '/') { ++p; c.vn = strtol(p, &end, 10); if (end == p) return false; p = end; }
        }
        out.push_back(c);
    }
    return out.size() >= 3;
}

static int load_obj_impl(const char* filename, uint32_t flags, rr_vertex** verts, uint32_t* n_verts, uint32_t** indices,
                         uint32_t* n_indices);

extern "C" int rr_host_mesh_load_obj(const char* filename, rr_vertex** verts, uint32_t* n_verts, uint32_t** indices,
                                     uint32_t* n_indices)
{
    return load_obj_impl(filename, 0u, verts, n_verts, indices, n_indices);
}

extern "C" int rr_host_mesh_load_obj_ex(const char* filename, uint32_t flags, rr_vertex** verts, uint32_t* n_verts,
                                        uint32_t** indices, uint32_t* n_indices)
{
    return load_obj_impl(filename, flags, verts, n_verts, indices, n_indices);
}

static int load_obj_impl(const char* filename, uint32_t flags, rr_vertex** verts, uint32_t* n_verts, uint32_t** indices,
                         uint32_t* n_indices)
{
    if (!filename || !verts || !n_verts || !indices || !n_indices) return RR_ERR_INVALID_ARGUMENT;
    const bool hardened = (flags & RR_OBJ_HARDENED) != 0;
    std::vector<Corner> poly;
    *verts = nullptr; *indices = nullptr; *n_verts = 0; *n_indices = 0;
    FILE* f = fopen(filename, "rb");
    if (!f) return RR_ERR_IO;                       // Mesh.cpp:9-10: load() returns false

    std::vector<float> pos, tex, nrm;
    std::vector<rr_vertex> out;
    std::string line;
    int ch;
    bool eof = false;
    while (!eof) {
        line.clear();
        while ((ch = fgetc(f)) != EOF && ch != '\n') line.push_back((char)ch);
        if (ch == EOF) { eof = true; if (line.empty()) break; }
        const char* l = line.c_str();
        float v[3];
        int ref[9];
        if (scan_reals(l, "v", 3, v)) pos.insert(pos.end(), v, v + 3);
        else if (scan_reals(l, "vt", 2, v)) tex.insert(tex.end(), v, v + 2);
        else if (scan_reals(l, "vn", 3, v)) nrm.insert(nrm.end(), v, v + 3);
        else if (hardened && scan_face_any(l, poly)) {
            // resolve negative / missing references, fan-triangulate, synthesise what is missing
            const long np = (long)pos.size() / 3, nt = (long)tex.size() / 2, nn = (long)nrm.size() / 3;
            std::vector<rr_vertex> cv(poly.size());
            bool have_n = true;
            for (size_t c = 0; c < poly.size(); ++c) {
                long a = poly[c].v < 0 ? np + poly[c].v + 1 : poly[c].v;
                long b = poly[c].vt < 0 ? nt + poly[c].vt + 1 : poly[c].vt;
                long n = poly[c].vn < 0 ? nn + poly[c].vn + 1 : poly[c].vn;
                if (a < 1 || a > np || b < 0 || b > nt || n < 0 || n > nn) { fclose(f); return RR_ERR_INVALID_ARGUMENT; }
                std::memset(&cv[c], 0, sizeof(rr_vertex));
                std::memcpy(cv[c].position, &pos[(size_t)(a - 1) * 3], 12);
                if (b) std::memcpy(cv[c].uv, &tex[(size_t)(b - 1) * 2], 8);
                if (n) std::memcpy(cv[c].norm, &nrm[(size_t)(n - 1) * 3], 12); else have_n = false;
            }
            for (size_t c = 1; c + 1 < poly.size(); ++c) {
                rr_vertex tri[3] = { cv[0], cv[c], cv[c + 1] };
                if (!have_n) {                 // flat normal on the side the winding faces (cross(e1,e2))
                    float e1[3], e2[3], g[3];
                    for (int k = 0; k < 3; ++k) { e1[k] = tri[1].position[k] - tri[0].position[k]; e2[k] = tri[2].position[k] - tri[0].position[k]; }
                    g[0] = e1[1] * e2[2] - e1[2] * e2[1]; g[1] = e1[2] * e2[0] - e1[0] * e2[2]; g[2] = e1[0] * e2[1] - e1[1] * e2[0];
                    float len = std::sqrt(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
                    if (len > 0.0f) for (int k = 0; k < 3; ++k) g[k] /= len;
                    for (int q = 0; q < 3; ++q) std::memcpy(tri[q].norm, g, 12);
                }
                out.insert(out.end(), tri, tri + 3);
            }
        }
        else if (!hardened && scan_face(l, ref)) {
            for (int c = 0; c < 3; ++c) {
                const long a = ref[c * 3 + 0], b = ref[c * 3 + 1], n = ref[c * 3 + 2];   // 1-based
                if (a < 1 || (size_t)a * 3 > pos.size() || b < 1 || (size_t)b * 2 > tex.size() || n < 1 ||
                    (size_t)n * 3 > nrm.size()) {
                    fclose(f);
                    return RR_ERR_INVALID_ARGUMENT;
                }
                rr_vertex vx;
                std::memset(&vx, 0, sizeof vx);
                std::memcpy(vx.position, &pos[(size_t)(a - 1) * 3], 12);
                std::memcpy(vx.uv, &tex[(size_t)(b - 1) * 2], 8);
                std::memcpy(vx.norm, &nrm[(size_t)(n - 1) * 3], 12);
                out.push_back(vx);
            }
        }
    }
    fclose(f);

    const size_t nv = out.size();
    rr_vertex* vo = (rr_vertex*)malloc((nv ? nv : 1) * sizeof(rr_vertex));
    uint32_t* io = (uint32_t*)malloc((nv ? nv : 1) * sizeof(uint32_t));
    if (!vo || !io) { free(vo); free(io); return RR_ERR_OUT_OF_MEMORY; }
    if (nv) std::memcpy(vo, out.data(), nv * sizeof(rr_vertex));
    for (size_t i = 0; i < nv; ++i) io[i] = (uint32_t)i;
    *verts = vo; *n_verts = (uint32_t)nv; *indices = io; *n_indices = (uint32_t)nv;
    return RR_OK;
}

extern "C" void rr_host_free(void* p) { free(p); }

// what rr_upload_mesh requires of positions: finite, |coordinate| <= 1e18 (box areas stay finite in fp32)
extern "C" int rr_host_validate_positions(const rr_vertex* verts, uint32_t n_verts, uint32_t* first_bad)
{
    if (!verts && n_verts) return RR_ERR_INVALID_ARGUMENT;
    for (uint32_t i = 0; i < n_verts; ++i)
        for (int k = 0; k < 3; ++k) {
            const float v = verts[i].position[k];
            if (!(std::fabs(v) <= 1e18f)) {            // false for NaN and infinities too
                if (first_bad) *first_bad = i;
                return RR_ERR_INVALID_ARGUMENT;
            }
        }
    return RR_OK;
}
