// rr_host_partition.cpp -- pure host geometry of a dispatch: where on the screen the scene can be seen at all, and how the
// 32x32 tiles of a frame are dealt to the ranks of a multi-GPU run.  No device, no context: the same answers on every rank.
#include "../../../include/rrdxr.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

// Screen rectangle (pixels, aligned outward to 8x8 blocks, 8 pixels of margin) that contains the projection of the box
// {lo[3], hi[3]} for every one of the n slices' constants.  GenerateCameraRay (RayTracing.hlsl:27-40) sends pixel s to
// the direction A * (sx, sy, 1) with A = columns 0, 1, 3 of proj_inv's upper three rows, so a point X is seen at
// (a/c, b/c) where A * (a, b, c) = X - camera_loc, provided c > 0.
// This rectangle is a CORRECTNESS path, not a hint: k_render_fused and k_render_paths do not trace the primary rays of
// blocks outside it (their pixels are one Miss), and k_render_lds orders its work by it.  So it has to hold for the rays
// the kernels really generate, which are fp32: R = (sx*M0 + sy*M1) + M3 per row carries an absolute error of a few
// 2^-24 * (|M0| + |M1| + |M3|), i.e. the fp32 ray of pixel s is the exact ray of a pixel up to ||A^-1|| * that error away.
// The 8-pixel margin (16 / max(W, H) in screen units, of which a quarter is spent here) therefore only covers matrices
// whose condition number cond_inf(A) = ||A||_inf * ||A^-1||_inf stays below margin / 2^-20; anything worse -- a singular or
// near-singular proj_inv, non-finite constants, a corner at or behind the camera plane (the camera inside or beside the
// box), a projection that overflows -- makes the rectangle the whole frame, which is always right.
// cams == nullptr (RR_DISPATCH_DEBUG_NO_CULL, and callers without a host copy of the constants): the whole frame.
static void screen_rect(const float box[6], const rr_scene_constants* cams, uint32_t n, uint32_t W, uint32_t H, uint32_t r[4])
{
    double x0 = 1e30, y0 = 1e30, x1 = -1e30, y1 = -1e30;
    bool all = cams == nullptr;
    for (int k = 0; k < 6 && !all; ++k) if (!std::isfinite(box[k])) all = true;
    const double margin_s = 16.0 / (double)std::max(W, H);          // 8 pixels in screen units (the frame spans 2)
    for (uint32_t f = 0; f < n && !all; ++f) {
        const float* M = cams[f].proj_inv;
        const double A[3][3] = { { M[0], M[1], M[3] }, { M[4], M[5], M[7] }, { M[8], M[9], M[11] } };
        double norm_a = 0.0;
        for (int i = 0; i < 3; ++i) norm_a = std::max(norm_a, std::fabs(A[i][0]) + std::fabs(A[i][1]) + std::fabs(A[i][2]));
        for (int i = 0; i < 3; ++i) if (!std::isfinite(cams[f].camera_loc[i])) all = true;
        if (all || !std::isfinite(norm_a) || !(norm_a > 0.0)) { all = true; break; }
        // adjugate (cofactors transposed): A^-1 = adj / det
        const double adj[3][3] = {
            { A[1][1] * A[2][2] - A[1][2] * A[2][1], A[0][2] * A[2][1] - A[0][1] * A[2][2], A[0][1] * A[1][2] - A[0][2] * A[1][1] },
            { A[1][2] * A[2][0] - A[1][0] * A[2][2], A[0][0] * A[2][2] - A[0][2] * A[2][0], A[0][2] * A[1][0] - A[0][0] * A[1][2] },
            { A[1][0] * A[2][1] - A[1][1] * A[2][0], A[0][1] * A[2][0] - A[0][0] * A[2][1], A[0][0] * A[1][1] - A[0][1] * A[1][0] } };
        const double det = A[0][0] * adj[0][0] + A[0][1] * adj[1][0] + A[0][2] * adj[2][0];
        double norm_adj = 0.0;
        for (int i = 0; i < 3; ++i) norm_adj = std::max(norm_adj, std::fabs(adj[i][0]) + std::fabs(adj[i][1]) + std::fabs(adj[i][2]));
        // cond = norm_a * norm_adj / |det|; require cond * 2^-20 <= margin_s / 4 (written without the division)
        if (!std::isfinite(det) || !std::isfinite(norm_adj) || !(std::fabs(det) * margin_s * 0.25 >= norm_a * norm_adj * 9.5367431640625e-07)) { all = true; break; }
        for (int c = 0; c < 8 && !all; ++c) {
            const double d[3] = { (double)((c & 1) ? box[3] : box[0]) - cams[f].camera_loc[0],
                                  (double)((c & 2) ? box[4] : box[1]) - cams[f].camera_loc[1],
                                  (double)((c & 4) ? box[5] : box[2]) - cams[f].camera_loc[2] };
            const double a = (adj[0][0] * d[0] + adj[0][1] * d[1] + adj[0][2] * d[2]) / det;
            const double b = (adj[1][0] * d[0] + adj[1][1] * d[1] + adj[1][2] * d[2]) / det;
            const double cc = (adj[2][0] * d[0] + adj[2][1] * d[1] + adj[2][2] * d[2]) / det;
            // in front of the camera plane by a margin relative to the corner's own size in these coordinates (a corner
            // near the plane projects to infinity, and its sign is not to be trusted)
            if (!(cc > 1e-4 * (std::fabs(a) + std::fabs(b) + std::fabs(cc))) || !(cc > 0.0)) { all = true; break; }
            const double sx = a / cc, sy = b / cc;
            const double px = (sx + 1.0) * 0.5 * W - 0.5, py = (1.0 - sy) * 0.5 * H - 0.5;
            if (!(std::fabs(px) < 1e9 && std::fabs(py) < 1e9)) { all = true; break; }
            x0 = std::min(x0, px); x1 = std::max(x1, px); y0 = std::min(y0, py); y1 = std::max(y1, py);
        }
    }
    const uint32_t W8 = (W + 7u) & ~7u, H8 = (H + 7u) & ~7u;
    if (all) { r[0] = 0; r[1] = 0; r[2] = W8; r[3] = H8; return; }
    auto lo8 = [](double v, uint32_t lim) { const double q = std::floor((v - 8.0) / 8.0) * 8.0; return q <= 0.0 ? 0u : q >= lim ? lim : (uint32_t)q; };
    auto hi8 = [](double v, uint32_t lim) { const double q = std::ceil((v + 9.0) / 8.0) * 8.0; return q <= 0.0 ? 0u : q >= lim ? lim : (uint32_t)q; };
    r[0] = lo8(x0, W8); r[1] = lo8(y0, H8); r[2] = hi8(x1, W8); r[3] = hi8(y1, H8);
}


} // namespace

// ---- how mesh tile i of a partition is dealt (shared with the kernels through rr_types.h's MeshDeal: the same arithmetic) ----
static void mesh_owner(uint32_t i, uint32_t world, uint32_t j, uint32_t& rank, uint32_t& slot)
{
    if (j == 0u || world == 1u) { rank = i % world; slot = i / world; return; }
    if (j == 0xffffffffu) { rank = 1u + i % (world - 1u); slot = i / (world - 1u); return; }
    const uint32_t cl = j * (world - 1u) + 1u, c = i / cl, pos = i % cl;
    if (pos == cl - 1u) { rank = 0u; slot = c; }
    else { rank = 1u + pos % (world - 1u); slot = c * j + pos / (world - 1u); }
}

extern "C" {

uint32_t rr_host_mesh_tiles_of_rank(const rr_mesh_partition* part, uint32_t rank)
{
    if (!part || part->world == 0 || rank >= part->world) return 0;
    const uint32_t n = part->n_mesh_tiles, w = part->world, j = part->rank0_rounds;
    if (j == 0u || w == 1u) return n > rank ? (n - rank + w - 1u) / w : 0u;
    if (j == 0xffffffffu) return rank == 0u ? 0u : (n > rank - 1u ? (n - (rank - 1u) + (w - 1u) - 1u) / (w - 1u) : 0u);
    const uint32_t cl = j * (w - 1u) + 1u, full = n / cl, rem = n % cl;
    if (rank == 0u) return full;                                            // (the cycle's last tile: a partial cycle never reaches it)
    uint32_t c = full * j;
    for (uint32_t pos = 0; pos < rem; ++pos) if (1u + pos % (w - 1u) == rank) ++c;
    return c;
}

int rr_host_mesh_tile_home(const rr_mesh_partition* part, uint32_t mesh_index, uint32_t* rank, uint32_t* slot)
{
    if (!part || !rank || !slot || part->world == 0 || mesh_index >= part->n_mesh_tiles) return RR_ERR_INVALID_ARGUMENT;
    mesh_owner(mesh_index, part->world, part->rank0_rounds, *rank, *slot);
    return RR_OK;
}

int rr_host_screen_rect(const float bounds[6], const rr_scene_constants* constants, uint32_t n, uint32_t width, uint32_t height, uint32_t rect[4])
{
    if (!bounds || !rect || width == 0 || height == 0) return RR_ERR_INVALID_ARGUMENT;
    screen_rect(bounds, constants, n, width, height, rect);
    return RR_OK;
}

// The tiles that touch the rectangle ("mesh tiles": every secondary ray of the frame starts in one of them) are dealt round
// robin, in raster order inside the rectangle; the others ("background tiles": one Miss per pixel) all belong to rank 0, which
// needs them where the frame is assembled anyway -- they are a tenth of the work and two thirds of the bytes.
int rr_host_mesh_partition(const float bounds[6], const rr_scene_constants* constants, uint32_t n, uint32_t width, uint32_t height,
                           uint32_t world, rr_mesh_partition* out)
{
    if (!bounds || !out || width == 0 || height == 0 || world == 0) return RR_ERR_INVALID_ARGUMENT;
    uint32_t r[4];
    screen_rect(bounds, constants, n, width, height, r);
    std::memset(out, 0, sizeof *out);
    const uint32_t T = 32;
    out->tiles_x = (width + T - 1) / T;
    const uint32_t tiles_y = (height + T - 1) / T;
    out->n_tiles = out->tiles_x * tiles_y;
    out->world = world;
    const uint32_t x0 = r[0] / T, y0 = r[1] / T;
    const uint32_t x1 = std::min(out->tiles_x, (r[2] + T - 1) / T), y1 = std::min(tiles_y, (r[3] + T - 1) / T);
    if (r[2] > r[0] && r[3] > r[1] && x1 > x0 && y1 > y0 && (x1 - x0) * (y1 - y0) < out->n_tiles && out->n_tiles < 65536u) {
        out->rect_x0 = x0; out->rect_y0 = y0; out->rect_w = x1 - x0; out->rect_h = y1 - y0;
        out->n_mesh_tiles = out->rect_w * out->rect_h;
    } else {
        out->n_mesh_tiles = out->n_tiles;           // no rectangle (rect_w == 0): every tile is a mesh tile, in raster order
    }
    out->n_bg_tiles = out->n_tiles - out->n_mesh_tiles;
    // Rank 0 renders every background tile: it takes fewer mesh tiles, so that its share of the WORK is the others'.  A mesh tile
    // is priced at 16 background tiles (the headline view's issue time: 74 % in 572 mesh tiles, 10 % in 1 468 background tiles).
    out->rank0_rounds = 0;
    if (world > 1 && out->n_bg_tiles > 0) {
        const double cm = 16.0, total = (double)out->n_bg_tiles + cm * out->n_mesh_tiles, share = total / world;
        const double m0 = (share - (double)out->n_bg_tiles) / cm;           // mesh tiles rank 0 should take
        if (m0 < 0.5) out->rank0_rounds = 0xffffffffu;
        else {
            const double f0 = m0 / (double)out->n_mesh_tiles;               // its fraction of them: 1 / (J * (world - 1) + 1)
            const double j = (1.0 / f0 - 1.0) / (double)(world - 1);
            out->rank0_rounds = j < 1.0 ? 1u : j > 1.0e6 ? 0xffffffffu : (uint32_t)(j + 0.5);
        }
    }
    uint32_t mx = 0;
    {   // slots per rank (dense by construction): the largest count is what every rank's gather buffer holds per frame
        uint32_t cnt_max = 0;
        for (uint32_t r = 0; r < world; ++r) cnt_max = std::max(cnt_max, rr_host_mesh_tiles_of_rank(out, r));
        mx = cnt_max;
    }
    out->max_mesh_tiles_per_rank = mx;
    return RR_OK;
}

} // extern "C"
