// rr_bvh_host.cpp -- EXPERIMENT ONLY (enabled with RR_DEBUG_BVH=sah): full-sweep SAH BVH2 built on the host,
// to measure how much traversal work a better hierarchy than the LBVH saves.  Not part of the product path.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>
#include "rr_types.h"

namespace rr {

struct HBox { float lo[3], hi[3]; };
static inline void grow(HBox& b, const HBox& o) { for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(b.lo[k], o.lo[k]); b.hi[k] = std::max(b.hi[k], o.hi[k]); } }
static inline float area(const HBox& b) { float d[3] = { b.hi[0] - b.lo[0], b.hi[1] - b.lo[1], b.hi[2] - b.lo[2] }; return 2.0f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]); }
static inline HBox empty() { HBox b; for (int k = 0; k < 3; ++k) { b.lo[k] = INFINITY; b.hi[k] = -INFINITY; } return b; }

struct SahBuilder {
    const std::vector<HBox>& pb;
    std::vector<uint32_t>& order;
    std::vector<BvhNode>& nodes;
    std::vector<float> right_area;
    uint32_t depth = 0;
    // returns child ref; box of the subtree in out
    int build(uint32_t a, uint32_t b, HBox& out, uint32_t d)
    {
        if (d > depth) depth = d;
        if (b - a == 1) { out = pb[order[a]]; return ~(int)a; }
        float best = INFINITY; int best_axis = 0; uint32_t best_split = (a + b) / 2;
        for (int axis = 0; axis < 3; ++axis) {
            std::sort(order.begin() + a, order.begin() + b, [&](uint32_t x, uint32_t y) {
                float cx = pb[x].lo[axis] + pb[x].hi[axis], cy = pb[y].lo[axis] + pb[y].hi[axis];
                return cx < cy || (cx == cy && x < y); });
            HBox acc = empty();
            for (uint32_t i = b; i-- > a + 1;) { grow(acc, pb[order[i]]); right_area[i] = area(acc); }
            acc = empty();
            for (uint32_t i = a; i + 1 < b; ++i) {
                grow(acc, pb[order[i]]);
                float c = area(acc) * (float)(i + 1 - a) + right_area[i + 1] * (float)(b - i - 1);
                if (c < best) { best = c; best_axis = axis; best_split = i + 1; }
            }
        }
        std::sort(order.begin() + a, order.begin() + b, [&](uint32_t x, uint32_t y) {
            float cx = pb[x].lo[best_axis] + pb[x].hi[best_axis], cy = pb[y].lo[best_axis] + pb[y].hi[best_axis];
            return cx < cy || (cx == cy && x < y); });
        const int id = (int)nodes.size();
        nodes.emplace_back();
        HBox b0, b1;
        const int c0 = build(a, best_split, b0, d + 1);
        const int c1 = build(best_split, b, b1, d + 1);
        BvhNode& n = nodes[id];
        n.lox[0] = b0.lo[0]; n.loy[0] = b0.lo[1]; n.loz[0] = b0.lo[2]; n.hix[0] = b0.hi[0]; n.hiy[0] = b0.hi[1]; n.hiz[0] = b0.hi[2];
        n.lox[1] = b1.lo[0]; n.loy[1] = b1.lo[1]; n.loz[1] = b1.lo[2]; n.hix[1] = b1.hi[0]; n.hiy[1] = b1.hi[1]; n.hiz[1] = b1.hi[2];
        n.c[0] = c0; n.c[1] = c1; n.pad[0] = n.pad[1] = 0;
        out = b0; grow(out, b1);
        return id;
    }
};

// verts: n_verts*8 floats, idx: 3*n_tris.  Fills nodes (n_tris-1, root = 0) and order (leaf i holds primitive order[i]).
uint32_t host_sah_build(const float* verts, const uint32_t* idx, uint32_t n_tris, std::vector<BvhNode>& nodes,
                        std::vector<uint32_t>& order)
{
    std::vector<HBox> pb(n_tris);
    for (uint32_t p = 0; p < n_tris; ++p) {
        pb[p] = empty();
        for (int c = 0; c < 3; ++c) {
            const float* v = verts + (size_t)idx[3 * p + c] * 8;
            for (int k = 0; k < 3; ++k) { pb[p].lo[k] = std::min(pb[p].lo[k], v[k]); pb[p].hi[k] = std::max(pb[p].hi[k], v[k]); }
        }
    }
    order.resize(n_tris);
    for (uint32_t i = 0; i < n_tris; ++i) order[i] = i;
    nodes.clear();
    nodes.reserve(n_tris);
    SahBuilder sb{ pb, order, nodes, std::vector<float>(n_tris + 1), 0 };
    HBox root;
    if (n_tris > 1) sb.build(0, n_tris, root, 1);
    return sb.depth + 1;
}

} // namespace rr
