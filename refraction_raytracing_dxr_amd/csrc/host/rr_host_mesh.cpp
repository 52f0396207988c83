// rr_host_mesh.cpp -- Wavefront OBJ -> Vertex[]/uint32[] with the observable behaviour of
// Mesh::load (Mesh.cpp:6-37): line oriented; "v", "vt", "vn" records feed three pools; every
// "f a/b/c a/b/c a/b/c" corner is un-indexed into a fresh 32-byte vertex {position, norm, uv}
// and indices are simply 0..3T-1 (Mesh.cpp:31).  Anything else on a line is ignored, a fourth
// corner is dropped, and lines that do not START with the keyword are skipped, exactly as the
// reference's sscanf cascade does.  Differences: out-of-range references fail the load
// (RR_ERR_INVALID_ARGUMENT) instead of reading outside the pools.
#include "../../../include/rrdxr.h"

#include <cerrno>
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

extern "C" int rr_host_mesh_load_obj(const char* filename, rr_vertex** verts, uint32_t* n_verts, uint32_t** indices,
                                     uint32_t* n_indices)
{
    if (!filename || !verts || !n_verts || !indices || !n_indices) return RR_ERR_INVALID_ARGUMENT;
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
        else if (scan_face(l, ref)) {
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
