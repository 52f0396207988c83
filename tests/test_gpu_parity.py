"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bars (stated once, used below):
  * TraceRay (hit / prim / t / u / v): BIT-EXACT -- both sides follow the same arithmetic spec.
  * frame, float colour: |gpu - oracle_literal| <= 1e-4 per channel on EVERY pixel, where
    oracle_literal is the recursive `color += w*child.color` restatement; and bit-exact against
    the oracle's path-weight mode (same summation order as the kernel).
  * frame, RGBA8: <= 1 LSB per channel against the literal oracle, identical against path-weight.
  * exact counters (rays, hits, misses, terminal hits, TIR) equal the oracle's.
"""
import ctypes as C
import os

import numpy as np
import pytest

import oracle as O
import refraction_raytracing_dxr_amd as rr
from conftest import procedural_env

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLOAT_TOL = 1e-4


@pytest.fixture(scope="module")
def gpu():
    r = rr.Renderer(0)
    yield r
    r.close()


def load(name):
    m = rr.Mesh()
    assert m.load(O.asset(name))
    return m


def oracle_scene(meshes, env, instances=None):
    s = O.Scene()
    for m in meshes:
        s.add_mesh(m.verts, m.indices)
    if instances is not None:
        inst = np.zeros(len(instances), O.INSTANCE_DTYPE)
        inst["transform"] = instances["transform"]
        inst["id_mask"] = instances["instance_id_mask"]
        inst["hitgroup_flags"] = instances["hitgroup_flags"]
        inst["blas"] = instances["blas"]
        s.set_instances(inst)
    s.set_envmap(env)
    return s


def gpu_scene(gpu, meshes, env, instances=None):
    ids = []
    for m in meshes:
        mid = gpu.upload_mesh(m.verts, m.indices)
        gpu.build_blas(mid)
        ids.append(mid)
    if instances is None:
        instances = rr.make_instances(meshes=[ids[0]])
    else:
        instances = instances.copy()
        instances["blas"] = [ids[int(b)] for b in instances["blas"]]
    gpu.build_tlas(instances)
    gpu.upload_envmap(env)
    return ids


def render_both(gpu, s, angle, W, H, stats=True, **kw):
    sc = rr.camera_orbit(angle)
    M, cam = np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32)
    flags = rr.DISPATCH_FLOAT_OUTPUT | (rr.DISPATCH_COLLECT_STATS if stats else 0)
    gpu.set_tile_partition(0, 1)
    gpu.set_camera(sc)
    gpu.dispatch_rays(W, H, rr.default_params(flags=flags, **kw))
    rgba, f32 = gpu.read_frame(want_float=True)
    st = gpu.stats()
    lit = s.render(M, cam, W, H, O.default_params(use_bvh=1, **kw))
    pw = s.render(M, cam, W, H, O.default_params(use_bvh=1, accum_mode=1, **kw))
    return rgba, f32, st, lit, pw


def check_frame(rgba, f32, st, lit, pw):
    assert st.traversal_overflow == 0
    o = lit["stats"]
    assert st.rays == o.rays and st.primary == o.primary and st.secondary == o.secondary
    if st.stats_valid:
        assert (st.hits, st.misses, st.terminal_hits, st.tir) == (o.hits, o.misses, o.terminal_hits, o.tir)
    assert np.all(f32[..., 3] == 1.0) and np.all(rgba[..., 3] == 255)
    # literal recursive oracle: stated tolerance on every pixel
    d = np.abs(f32[..., :3] - lit["rgb"])
    assert d.max() <= FLOAT_TOL, "max |d| %.3g at %s" % (d.max(), np.unravel_index(d.argmax(), d.shape))
    assert np.abs(rgba.astype(int) - lit["rgba8"].astype(int)).max() <= 1
    # path-weight oracle (same summation order as the kernel): bit-exact
    assert np.array_equal(f32[..., :3].view(np.uint32), pw["rgb"].view(np.uint32))
    assert np.array_equal(rgba, pw["rgba8"])


# ------------------------------------------------------------------------------- TraceRay
def random_rays(n, seed, radius=4.0):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, rr.RAY_DTYPE)
    o = rng.normal(size=(n, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * rng.uniform(0.0, radius, (n, 1))
    tgt = rng.uniform(-1.2, 1.2, (n, 3))
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["origin"] = o.astype(np.float32)
    rays["dir"] = d.astype(np.float32)
    rays["tmin"] = np.where(rng.random(n) < 0.5, 1e-4, 1e-3).astype(np.float32)
    rays["tmax"] = rng.choice([100.0, 1000.0, 3.0], n).astype(np.float32)
    rays["flags"] = rng.choice([rr.RAY_FLAG_CULL_BACK, rr.RAY_FLAG_CULL_FRONT, 0], n, p=[0.45, 0.45, 0.1])
    return rays


@pytest.mark.parametrize("name,n", [("cube.obj", 4000), ("sphere.obj", 6000), ("monkey.obj", 6000),
                                    ("shell.obj", 6000), ("ott.obj", 3000)])
def test_trace_rays_bit_exact_vs_brute_force(gpu, name, n):
    m = load(name)
    gpu_scene(gpu, [m], procedural_env(32, 16))
    s = oracle_scene([m], procedural_env(32, 16))
    rays = random_rays(n, seed=len(name))
    # include axis-aligned and in-plane rays (zero direction components, origins on box planes)
    rays["dir"][:8] = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 0, 0), (0, 0, 1)]
    rays["origin"][:8] = [(-5, 0, 0), (5, 0.25, 0.125), (0, -5, 0), (0.1, 5, 0.1), (0, 0, -5), (0.2, 0.1, 5),
                          (-5, 1, 1), (1, -1, -5)]
    rays["tmax"][:8] = 100.0
    hits = gpu.trace_rays(rays)
    n_hit = 0
    for k in range(n):
        h = s.trace(rays["origin"][k], rays["dir"][k], float(rays["tmin"][k]), float(rays["tmax"][k]),
                    int(rays["flags"][k]), use_bvh=0)
        g = hits[k]
        assert bool(g["hit"]) == bool(h.hit), "ray %d" % k
        if h.hit:
            n_hit += 1
            assert g["prim"] == h.prim
            assert np.float32(g["t"]).view(np.uint32) == np.float32(h.t).view(np.uint32)
            assert np.float32(g["u"]).view(np.uint32) == np.float32(h.u).view(np.uint32)
            assert np.float32(g["v"]).view(np.uint32) == np.float32(h.v).view(np.uint32)
    assert n_hit > n // 20


def _soup(kind, n, seed):
    """awkward geometry for the builder and the quantised boxes (vertex records, identity indices)"""
    rng = np.random.default_rng(seed)
    if kind == "flat":                     # every triangle in the plane z = 0.25: zero extent on one axis
        P = rng.uniform(-2, 2, (n, 3, 3)); P[..., 2] = 0.25
    elif kind == "far":                    # small mesh far from the origin: coordinates ~1e3, extent ~1
        P = rng.uniform(-0.5, 0.5, (n, 3, 3)) * 0.2 + rng.uniform(-0.5, 0.5, (n, 1, 3)) + np.array([1000.0, -2000.0, 500.0])
    elif kind == "mixed":                  # huge and tiny triangles, slivers, a few degenerate ones
        c = rng.uniform(-3, 3, (n, 1, 3))
        P = c + rng.normal(size=(n, 3, 3)) * rng.choice([1e-4, 1e-2, 0.3, 2.0], (n, 1, 1))
        P[::17, 1] = P[::17, 0]            # zero-area: two equal vertices
        P[5::29, 2] = (P[5::29, 0] + P[5::29, 1]) / 2          # zero-area: collinear
    else:                                  # "line": all centroids on one line (Morton codes collide massively)
        t = rng.uniform(-2, 2, (n, 1, 1))
        P = t * np.array([1.0, 1.0, 1.0]) + rng.normal(size=(n, 3, 3)) * 0.01
    v = np.zeros(n * 3, rr.VERTEX_DTYPE)
    v["position"] = P.reshape(-1, 3).astype(np.float32)
    v["norm"] = (0, 0, 1)
    return v, np.arange(n * 3, dtype=np.uint32)


@pytest.mark.parametrize("kind,n", [("flat", 300), ("far", 400), ("mixed", 700), ("line", 500), ("mixed", 1), ("far", 2), ("flat", 3),
                                    ("line", 5), ("mixed", 9), ("far", 33), ("flat", 64), ("mixed", 65)])
@pytest.mark.parametrize("fast_build", [False, True])
def test_trace_rays_awkward_geometry_vs_brute_force(gpu, kind, n, fast_build):
    """flat, far-away, wildly mixed-size / degenerate and collinear triangle soups: closest hit through the GPU
    hierarchy (both builders, fp16 boxes on the grid of the bounds) == the oracle's brute force, bit for bit"""
    verts, idx = _soup(kind, n, seed=n + len(kind))
    mid = gpu.upload_mesh(verts, idx)
    gpu.build_blas(mid, fast_build=fast_build)
    gpu.build_tlas(rr.make_instances(meshes=[mid]))
    s = O.Scene()
    s.add_mesh(verts, idx)
    rng = np.random.default_rng(7)
    P = verts["position"].astype(np.float64)
    lo, hi = P.min(0), P.max(0)
    ctr, ext = (lo + hi) / 2, max(float((hi - lo).max()), 1e-3)
    m_rays = 1500
    rays = np.zeros(m_rays, rr.RAY_DTYPE)
    o = ctr + rng.normal(size=(m_rays, 3)) * ext
    tgt = P[rng.integers(0, len(P), m_rays)] + rng.normal(size=(m_rays, 3)) * ext * 0.02     # aim near real vertices
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["origin"], rays["dir"] = o.astype(np.float32), d.astype(np.float32)
    rays["origin"][:6] = (ctr + np.array([(-3, 0, 0), (3, 0, 0), (0, -3, 0), (0, 3, 0), (0, 0, -3), (0, 0, 3)]) * ext).astype(np.float32)
    rays["dir"][:6] = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]           # axis-aligned, some in-plane
    rays["tmin"] = 1e-4
    rays["tmax"] = 1e6
    rays["flags"] = rng.choice([rr.RAY_FLAG_CULL_BACK, rr.RAY_FLAG_CULL_FRONT, 0], m_rays)
    hits = gpu.trace_rays(rays)
    n_hit = 0
    for k in range(m_rays):
        h = s.trace(rays["origin"][k], rays["dir"][k], 1e-4, 1e6, int(rays["flags"][k]), use_bvh=0)
        g = hits[k]
        assert bool(g["hit"]) == bool(h.hit), "ray %d" % k
        if h.hit:
            n_hit += 1
            assert g["prim"] == h.prim and np.float32(g["t"]).view(np.uint32) == np.float32(h.t).view(np.uint32)
    assert n_hit >= (20 if n >= 100 else 0)       # (tiny soups offer little to hit; "mixed, 1" is a single degenerate triangle)


def test_miss_shader_bit_exact_incl_out_of_range_texels(gpu):
    """Miss (RayTracing.hlsl:127-137) in isolation, GPU against oracle, bit for bit: the texture edges the shader can
    reach because it divides by the literal 3.14159 (atan2 = pi -> theta >= W; r.y = -1 -> phi >= H: zero texel,
    SURVEY A.2), the axes, signed zeros, every range of the atan / acos range reduction, and 20 000 random directions."""
    m = load("cube.obj")
    env = (np.arange(8 * 4 * 3, dtype=np.float32).reshape(4, 8, 3) + 1.0)
    gpu_scene(gpu, [m], env)
    s = oracle_scene([m], env)
    rng = np.random.default_rng(11)
    d = rng.standard_normal((20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    edge = [(0, 0, -1), (0, -1, 0), (0, 1, 0), (1, 0, 0), (-1, 0, 0), (0, 0, 1), (-0.0, 0, -1), (0.0, -0.0, -1), (-0.0, -1, -0.0),
            (1e-8, 0, -1), (-1e-8, 0, -1), (0, -0.99999994, 0.00034526698), (0, 0.5, 0.8660254), (0, -0.5, 0.8660254),
            (0, 0.50000006, 0.8660254), (0, -0.50000006, -0.8660254)]
    for q in (0.41421354, 0.4142136, 2.4142134, 2.4142137, 1.0, 1e-30, 1e30):           # atan range-reduction boundaries
        for sx in (1.0, -1.0):
            for sz in (1.0, -1.0):
                v = np.array([sx * q, 0.1, sz], np.float64); v /= np.linalg.norm(v)
                edge.append(tuple(v))
    dirs = np.concatenate([np.array(edge, np.float32), d])
    got = gpu.env_lookup(dirs)
    ref = np.stack([s.env_lookup(tuple(float(c) for c in v)) for v in dirs])
    assert np.array_equal(got.view(np.uint32), ref.astype(np.float32).view(np.uint32))
    assert np.array_equal(got[0], (0, 0, 0)) and np.array_equal(got[1], (0, 0, 0))      # the two out-of-range texels
    assert np.array_equal(got[2], env[0, 4]) and np.array_equal(got[3], env[2, 6])


def test_out_of_range_texel_through_dispatch_rays(gpu):
    """The same two edges through rr_dispatch_rays: a 1x1 frame whose only ray leaves along -z / -y and misses."""
    m = load("cube.obj")
    env = np.full((4, 8, 3), 0.75, np.float32)
    gpu_scene(gpu, [m], env)
    for direction, expect in (((0, 0, -1), 0.0), ((0, -1, 0), 0.0), ((0, 1, 0), 0.75)):
        M = np.zeros((4, 4), np.float32)
        M[0, 3], M[1, 3], M[2, 3] = direction                    # R = M * (sx, sy, 0, 1): the constant column is the direction
        gpu.set_camera(rr.scene_constants(M.reshape(16), (50.0, 50.0, 50.0, 1.0)))
        gpu.dispatch_rays(1, 1, rr.default_params(flags=rr.DISPATCH_FLOAT_OUTPUT))
        rgba, f32 = gpu.read_frame(want_float=True)
        assert np.array_equal(f32[0, 0, :3], np.full(3, expect, np.float32)), (direction, f32)
        assert rgba[0, 0, 0] == (0 if expect == 0.0 else 191)


def test_tonemapped_frame_matches_the_oracle(gpu):
    """RR_DISPATCH_TONEMAP_REINHARD on a real frame: monkey.obj under the bench's HDR env map (texels up to 16, most of them
    saturate the plain store), RGBA8 identical to the oracle's tone-mapped store, float frame identical to the plain run."""
    m = load("monkey.obj")
    env = procedural_env(256, 128, seed=0)
    assert env.max() > 4.0
    gpu_scene(gpu, [m], env)
    s = oracle_scene([m], env)
    sc = rr.camera_orbit(0.2)
    gpu.set_camera(sc)
    W, H = 160, 100
    out = {}
    for tm in (0, 1):
        gpu.dispatch_rays(W, H, rr.default_params(max_refract=8, flags=rr.DISPATCH_FLOAT_OUTPUT | (rr.DISPATCH_TONEMAP_REINHARD if tm else 0)))
        out[tm] = gpu.read_frame(want_float=True)
        ref = s.render(np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32), W, H, O.default_params(use_bvh=1, max_refract=8, accum_mode=1, tonemap=tm))
        assert np.array_equal(out[tm][0], ref["rgba8"]), tm
    assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    assert (out[0][0][..., :3] == 255).mean() > 0.3 and (out[1][0][..., :3] == 255).mean() < 0.01      # the plain store saturates, the tone map does not


def test_unorm8_store_edges_through_dispatch_rays(gpu):
    """The typed UAV store to R8G8B8A8_UNORM (RefractionDemo.cpp:431; hlsl:62): NaN -> 0, below 0 -> 0, from 1 up -> 255,
    floor(x * 255 + 0.5) in between -- a frame of Misses over an env map whose texels ARE the edge values (the kernel's
    store is written without branches), RGBA8 against the oracle's and against the rule itself."""
    m = load("cube.obj")
    vals = np.array([np.nan, -np.inf, -1.0, -1e-30, -0.0, 0.0, 1e-30, 0.5 / 255, np.nextafter(np.float32(0.5 / 255), np.float32(1)),
                     np.nextafter(np.float32(0.5 / 255), np.float32(0)), 1.5 / 255, 0.49999997, 0.5, 0.50196078, 254.5 / 255,
                     np.nextafter(np.float32(254.5 / 255), np.float32(0)), np.nextafter(np.float32(1), np.float32(0)), 1.0,
                     np.nextafter(np.float32(1), np.float32(2)), 2.0, 1e30, np.inf, 0.1, 0.2, 0.3, 0.7, 0.9, 0.999, 0.25, 0.75,
                     0.003, 0.996], np.float32)
    ii, jj = np.meshgrid(np.arange(128), np.arange(256), indexing="ij")
    env = np.stack([vals[(ii * 7 + jj * 13) % 32], vals[(ii * 11 + jj * 5 + 3) % 32], vals[(ii * 3 + jj * 17 + 9) % 32]], axis=-1).astype(np.float32)
    gpu_scene(gpu, [m], env)
    s = oracle_scene([m], env)
    sc = rr.camera_orbit(0.01)
    sc.camera_loc[0] += 40.0; sc.camera_loc[1] += 40.0            # far from the cube: every pixel is a Miss
    gpu.set_camera(sc)
    W, H = 96, 64
    gpu.dispatch_rays(W, H, rr.default_params(flags=rr.DISPATCH_FLOAT_OUTPUT | rr.DISPATCH_COLLECT_STATS))
    rgba, f32 = gpu.read_frame(want_float=True)
    assert gpu.stats().hits == 0 and gpu.stats().misses == W * H
    ref = s.render(np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32), W, H, O.default_params(use_bvh=1, accum_mode=1))
    assert np.array_equal(rgba, ref["rgba8"])
    x = f32[..., :3]
    with np.errstate(invalid="ignore"):
        want = np.where(~(x > 0), 0, np.where(x >= 1, 255, np.floor(x * np.float32(255.0) + np.float32(0.5)))).astype(np.uint8)
    assert np.array_equal(rgba[..., :3], want) and (rgba[..., 3] == 255).all()
    seen = set(np.unique(rgba[..., :3]))
    assert {0, 1, 127, 128, 254, 255} <= seen                     # the frame really visits the edges
    # the tone-map option (RR_DISPATCH_TONEMAP_REINHARD): c / (1 + c) in front of the same store; the float frame stays linear
    gpu.dispatch_rays(W, H, rr.default_params(flags=rr.DISPATCH_FLOAT_OUTPUT | rr.DISPATCH_TONEMAP_REINHARD))
    rgba_t, f32_t = gpu.read_frame(want_float=True)
    ref_t = s.render(np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32), W, H, O.default_params(use_bvh=1, accum_mode=1, tonemap=1))
    assert np.array_equal(rgba_t, ref_t["rgba8"])
    assert np.array_equal(f32_t.view(np.uint32), f32.view(np.uint32))
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        xm = np.minimum(x, np.float32(3.4028234663852886e38))
        t = np.where(x > 0, xm / (np.float32(1.0) + xm), np.float32(0.0)).astype(np.float32)
        want_t = np.where(~(t > 0), 0, np.where(t >= 1, 255, np.floor(t * np.float32(255.0) + np.float32(0.5)))).astype(np.uint8)
    assert np.array_equal(rgba_t[..., :3], want_t)
    assert rgba_t[..., :3].max() == 255 and (rgba_t[..., :3][np.isinf(x) & (x > 0)] == 255).all()      # +inf -> 1


def test_lds_and_path_parallel_kernels_render_the_same_frames(tmp_path):
    """k_render_lds (persistent workgroups, BLAS nodes in LDS, tickets, reflected rays parked in memory) and k_render_paths
    (four lanes per pixel, one per root-to-leaf path of the ray tree) against k_render_fused: every frame byte for byte, float colours bit for bit, the same counters --
    at Depth 1 (mesh rectangle first), Depth 5 / 40 (image order), with float output, with stats, sharded tiles (the
    scan form of the ticket order) and for each workgroup shape.  Own processes: the switches are read at rr_create."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import refraction_raytracing_dxr_amd as rr\n"
        "from refraction_raytracing_dxr_amd.synth import asset, procedural_env\n"
        "r = rr.Renderer(0); out = []; cnt = []\n"
        "for name, kw in (('monkey.obj', dict(max_refract=8)), ('sphere.obj', dict(max_refract=4, max_reflect=1)), ('shell.obj', dict(max_reflect=4)), ('cube.obj', dict())):\n"
        "    m = rr.Mesh(); m.load(asset(name))\n"
        "    r.load_scene(m.verts, m.indices, procedural_env(256, 128, seed=9))\n"
        "    for depth, frames in ((1, 2), (5, 5), (40, 40)):\n"
        "        r.render_orbit(323, 181, frames, angle=0.3, params=rr.default_params(flags=rr.DISPATCH_COLLECT_STATS | rr.DISPATCH_FLOAT_OUTPUT, **kw), frames_per_dispatch=depth)\n"
        "        rgba, f32 = r.read_frame(want_float=True, slice=depth - 1)\n"
        "        st = r.stats(); assert st.traversal_overflow == 0\n"
        "        out += [rgba.view(np.uint32)[..., 0].astype(np.float64), f32[..., 0].astype(np.float64), f32[..., 2].astype(np.float64)]\n"
        "        cnt += [st.rays, st.hits, st.misses, st.terminal_hits, st.tir, st.node_visits, st.tri_tests, st.pixels]\n"
        "    r.set_tile_partition(1, 3)\n"
        "    r.render_orbit(323, 181, 3, angle=0.3, params=rr.default_params(**kw), frames_per_dispatch=3)\n"
        "    cnt += [r.stats().rays]\n"
        "    r.set_tile_partition(0, 1)\n"
        "np.save(sys.argv[1], np.stack(out)); print(' '.join(str(c) for c in cnt))\n") % ROOT
    res = {}
    for k, extra in (("fused", {}), ("lds", {}), ("lds", {"RR_DEBUG_SHAPE": "1"}), ("lds", {"RR_DEBUG_SHAPE": "2", "RR_DEBUG_TICKET": "19"}),
                     ("paths", {})):            # k_render_paths: four lanes per pixel, leaves summed in the recursion's order
        env = dict(os.environ, RR_DEBUG_KERNEL=k, **extra)
        tag = k + "".join(extra.values())
        p = subprocess.run([sys.executable, "-c", code, str(tmp_path / (tag + ".npy"))], capture_output=True, text=True, env=env, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[tag] = (np.load(tmp_path / (tag + ".npy")), p.stdout.split())
    for tag in res:
        if not np.array_equal(res["fused"][0], res[tag][0]):
            bad = np.argwhere(res["fused"][0] != res[tag][0])
            raise AssertionError("%s: %d values differ, first at %s: %r / %r" % (tag, len(bad), bad[0], res["fused"][0][tuple(bad[0])], res[tag][0][tuple(bad[0])]))
        a, b = list(res["fused"][1]), list(res[tag][1])
        if tag.startswith("paths"):
            # k_render_paths traces the sparse late ray levels with 2 or 4 lanes per ray (trace_blas_group): the same closest hits
            # in a different visiting order, so node_visits / tri_tests (entries 5, 6 of each group of 8) are its own; the ray,
            # hit, miss, terminal-hit, TIR and pixel counts are the recursion's and must be equal
            keep = [i for i in range(len(a)) if not ((i % 25) < 24 and (i % 25) % 8 in (5, 6))]      # per mesh: 3 x 8 counters + the sharded ray count
            a, b = [a[i] for i in keep], [b[i] for i in keep]
        assert a == b, tag


def test_stream_renderer_renders_the_same_frames(tmp_path):
    """k_stream_* (two-level scenes: one kernel per ray generation, rays in HBM queues, lanes refilled as their rays end, chains
    followed in place) against the lock-step k_render_fused: every frame byte for byte, float colours bit for bit, the recursion's
    counters equal -- instanced scenes with rotations, non-uniform scales, cull flags and a zero mask, a grid of 100 monkeys,
    Depth 1 / 3 / 9, bounce limits from 0/0 to 12/2 (max_reflect decides which generation the chain kernel starts at), float
    output, the tone map, sharded tiles (round-robin and the mesh-tile partition, which must also equal the unsharded frames).
    Own processes: the switch is read at rr_create."""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import refraction_raytracing_dxr_amd as rr\n"
        "from refraction_raytracing_dxr_amd.synth import asset, procedural_env\n"
        "def xf(tx, ty, tz, s=(1, 1, 1), rot=0.0):\n"
        "    c, sn = np.cos(rot), np.sin(rot)\n"
        "    R = np.array([[c, 0, sn], [0, 1, 0], [-sn, 0, c]], np.float32) * np.array(s, np.float32)\n"
        "    return np.concatenate([R, np.array([[tx], [ty], [tz]], np.float32)], axis=1)\n"
        "r = rr.Renderer(0); out = []; cnt = []\n"
        "ids = []\n"
        "for name in ('cube.obj', 'monkey.obj', 'sphere.obj'):\n"
        "    m = rr.Mesh(); m.load(asset(name)); i = r.upload_mesh(m.verts, m.indices); r.build_blas(i); ids.append(i)\n"
        "r.upload_envmap(procedural_env(256, 128, seed=9))\n"
        "scenes = [rr.make_instances(transforms=[xf(0, 0, 0), xf(0, 0, -2.5, (0.5, 0.8, 0.5), 0.4), xf(0.3, 0.2, 2.4, (0.7, 0.7, 0.7), -1.0), xf(0, 1.9, 0, (0.4, 0.4, 0.4), 0.2), xf(0, -1.8, 0.5, (0.5, 0.5, 0.5))],\n"
        "                            meshes=[ids[1], ids[0], ids[1], ids[2], ids[0]], masks=[1, 1, 0xff, 1, 0], flags=[0, 0, 2, 1, 0]),\n"
        "          rr.make_instances(transforms=[xf(1.1 * (i - 4.5), 0.3 * ((i + j) %% 3), 1.1 * (j - 4.5), (0.4, 0.4, 0.4), 0.3 * i) for i in range(10) for j in range(10)], meshes=[ids[1]] * 100),\n"
        "          rr.make_instances(transforms=[xf(0.1, 0.05, 0, (0.12, 0.12, 0.12), 0.5), xf(-0.1, 0, 0.1, (0.08, 0.1, 0.08), 2.0)], meshes=[ids[1], ids[2]])]\n"
        "for si, inst in enumerate(scenes):\n"
        "    r.build_tlas(inst)\n"
        "    for depth, frames, kw, extra in ((1, 2, dict(max_refract=8), 0), (3, 3, dict(max_refract=5, max_reflect=1), 0), (9, 9, dict(max_refract=12), rr.DISPATCH_TONEMAP_REINHARD),\n"
        "                                     (1, 1, dict(max_refract=0, max_reflect=0), 0), (2, 2, dict(max_refract=3, max_reflect=0), 0), (1, 1, dict(max_refract=1, max_reflect=2), 0)):\n"
        "        r.render_orbit(211, 149, frames, angle=0.3 + si, params=rr.default_params(flags=rr.DISPATCH_COLLECT_STATS | rr.DISPATCH_FLOAT_OUTPUT | extra, **kw), frames_per_dispatch=depth)\n"
        "        rgba, f32 = r.read_frame(want_float=True, slice=depth - 1)\n"
        "        st = r.stats(); assert st.traversal_overflow == 0\n"
        "        out += [rgba.view(np.uint32)[..., 0].astype(np.float64), f32[..., 0].astype(np.float64), f32[..., 2].astype(np.float64)]\n"
        "        cnt += [st.rays, st.hits, st.misses, st.terminal_hits, st.tir, st.pixels, st.render_kernel]\n"
        "    import ctypes as C\n"
        "    mx = rr.dist.max_local_tiles(211, 149, 3); gathered = C.c_void_p(); rays = 0\n"
        "    assert r._L.rr_device_alloc(r._h, 3 * mx * 4096, C.byref(gathered)) == 0\n"
        "    for rank in range(3):\n"
        "        r.set_tile_partition(rank, 3)\n"
        "        r.dispatch_rays(211, 149, rr.default_params(max_refract=6))\n"
        "        r.export_tiles(gathered.value + rank * mx * 4096); r.wait(); rays += r.stats().rays\n"
        "    r.assemble_tiles(gathered.value, 3)\n"
        "    out += [r.read_frame().view(np.uint32)[..., 0].astype(np.float64)]\n"
        "    cnt += [rays]\n"
        "    r._L.rr_device_free(r._h, gathered)\n"
        "    r.set_tile_partition(0, 1)\n"
        "    # the mesh-tile partition (rr_mesh_partition), three ranks one after the other on this context, two frames\n"
        "    F, Wm, Hm, ang = 2, 211, 149, 0.3 + si\n"
        "    r.set_tile_partition(0, 3); part = r.mesh_partition_for_orbit(Wm, Hm, F, angle=ang)\n"
        "    fs = max(part.max_mesh_tiles_per_rank, 1) * 3072; bs = max(part.n_bg_tiles, 1) * 3072\n"
        "    gat, bg, frames = C.c_void_p(), C.c_void_p(), C.c_void_p()\n"
        "    assert r._L.rr_device_alloc(r._h, 3 * F * fs, C.byref(gat)) == 0 and r._L.rr_device_alloc(r._h, F * bs, C.byref(bg)) == 0\n"
        "    assert r._L.rr_device_alloc(r._h, F * Wm * Hm * 4, C.byref(frames)) == 0\n"
        "    for rank in range(3):\n"
        "        r.set_tile_partition(rank, 3)\n"
        "        r.render_orbit_mesh_sharded(Wm, Hm, F, C.c_void_p(gat.value + rank * F * fs), fs, bg if rank == 0 else None, bs, angle=ang, params=rr.default_params(max_refract=6))\n"
        "        r.lane_join(0); r.wait()\n"
        "        assert r.stats().render_kernel == (7 if os.environ.get('RR_DEBUG_KERNEL') == 'stream' else 0), r.stats().render_kernel\n"
        "    r.set_tile_partition(0, 3)\n"
        "    r.assemble_frames_mesh(gat, F * fs, fs, bg, bs, part, F, Wm, Hm, frames, Wm * Hm * 4); r.wait()\n"
        "    got = np.empty((F, Hm, Wm), np.uint32)\n"
        "    assert r._L.rr_device_read(r._h, frames, got.ctypes.data_as(C.c_void_p), got.nbytes) == 0\n"
        "    r.set_tile_partition(0, 1)\n"
        "    r.render_orbit(Wm, Hm, F, angle=ang, params=rr.default_params(max_refract=6), frames_per_dispatch=F)\n"
        "    for k in range(F): assert np.array_equal(got[k], r.read_frame(slice=k).view(np.uint32)[..., 0]), ('mesh partition', si, k)\n"
        "    out += [got[F - 1][:149, :211].astype(np.float64)]\n"
        "    for b in (gat, bg, frames): r._L.rr_device_free(r._h, b)\n"
        "    if si == 2:      # a small scene on eight ranks: rank 0 keeps the background and gets no mesh tile at all\n"
        "        r.set_tile_partition(0, 8); p8 = r.mesh_partition_for_orbit(320, 200, 1, angle=ang)\n"
        "        assert p8.rank0_rounds == 0xffffffff and p8.n_mesh_tiles >= 1, (p8.rank0_rounds, p8.n_mesh_tiles, p8.n_bg_tiles)\n"
        "        fs8 = max(p8.max_mesh_tiles_per_rank, 1) * 3072; bs8 = max(p8.n_bg_tiles, 1) * 3072\n"
        "        g8, b8, f8 = C.c_void_p(), C.c_void_p(), C.c_void_p()\n"
        "        assert r._L.rr_device_alloc(r._h, 8 * fs8, C.byref(g8)) == 0 and r._L.rr_device_alloc(r._h, bs8, C.byref(b8)) == 0 and r._L.rr_device_alloc(r._h, 320 * 200 * 4, C.byref(f8)) == 0\n"
        "        for rank in range(8):\n"
        "            r.set_tile_partition(rank, 8)\n"
        "            r.render_orbit_mesh_sharded(320, 200, 1, C.c_void_p(g8.value + rank * fs8), fs8, b8 if rank == 0 else None, bs8, angle=ang, params=rr.default_params(max_refract=6))\n"
        "            r.lane_join(0); r.wait()\n"
        "        r.set_tile_partition(0, 8)\n"
        "        r.assemble_frames_mesh(g8, fs8, fs8, b8, bs8, p8, 1, 320, 200, f8, 320 * 200 * 4); r.wait()\n"
        "        got8 = np.empty((200, 320), np.uint32)\n"
        "        assert r._L.rr_device_read(r._h, f8, got8.ctypes.data_as(C.c_void_p), got8.nbytes) == 0\n"
        "        r.set_tile_partition(0, 1)\n"
        "        r.render_orbit(320, 200, 1, angle=ang, params=rr.default_params(max_refract=6), frames_per_dispatch=1)\n"
        "        assert np.array_equal(got8, r.read_frame().view(np.uint32)[..., 0]), 'eight ranks'\n"
        "        for b in (g8, b8, f8): r._L.rr_device_free(r._h, b)\n"
        "    # launches in flight: every lane has its own queues (seven frames, one per dispatch, three lanes) == one at a time\n"
        "    r.set_frames_in_flight(3)\n"
        "    fl = r.render_orbit_to_host(Wm, Hm, 7, angle=ang, params=rr.default_params(max_refract=7), frames_per_dispatch=1).view(np.uint32)[..., 0]\n"
        "    r.set_frames_in_flight(1)\n"
        "    aa = np.float32(ang)\n"
        "    for k in range(7):\n"
        "        r.render_orbit(Wm, Hm, 1, angle=float(aa), params=rr.default_params(max_refract=7), frames_per_dispatch=1); aa = np.float32(aa + np.float32(0.01))\n"
        "        assert np.array_equal(fl[k], r.read_frame().view(np.uint32)[..., 0]), ('in flight', si, k)\n"
        "    out += [fl[6].astype(np.float64)]\n"
        "np.save(sys.argv[1], np.stack(out)); print(' '.join(str(c) for c in cnt))\n") % ROOT
    res = {}
    for k in ("fused", "stream"):
        env = dict(os.environ, RR_DEBUG_KERNEL=k)
        p = subprocess.run([sys.executable, "-c", code, str(tmp_path / (k + ".npy"))], capture_output=True, text=True, env=env, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[k] = (np.load(tmp_path / (k + ".npy")), p.stdout.split())
    assert np.array_equal(res["fused"][0], res["stream"][0])
    a, b = res["fused"][1], res["stream"][1]
    assert len(a) == len(b)
    kernels = [int(b[i]) for i in range(len(b)) if i % 43 < 42 and (i % 43) % 7 == 6]
    assert kernels and all(k == 7 for k in kernels), kernels          # the stream renderer really rendered every one of them
    assert [x for i, x in enumerate(a) if not (i % 43 < 42 and (i % 43) % 7 == 6)] == [x for i, x in enumerate(b) if not (i % 43 < 42 and (i % 43) % 7 == 6)]


def test_ploc_builder_makes_progress_on_equal_and_overflowing_areas(gpu):
    """The PREFER_FAST_TRACE builder merges mutually nearest clusters by merged-box area.  Identical triangles (every
    candidate area equal), a regular lattice (ties everywhere) and coordinates of 1e18 (areas near the top of fp32) must all
    build and trace like brute force: the neighbour search is total and a round without a mutual pair forces one."""
    rng = np.random.default_rng(5)
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    cases = {
        "identical": np.tile(tri, (40, 1)),
        "lattice": np.concatenate([tri + np.array([i, j, 0], np.float32) * 2 for i in range(7) for j in range(7)]),
        "huge": np.concatenate([tri * np.float32(1e17) + np.array([i, 0, 0], np.float32) * np.float32(8e17) for i in range(2)]),
    }
    for name, pos in cases.items():
        v = np.zeros(len(pos), rr.VERTEX_DTYPE)
        v["position"] = pos
        v["norm"] = (0, 0, 1)
        idx = np.arange(len(pos), dtype=np.uint32)
        mid = gpu.upload_mesh(v, idx)
        gpu.build_blas(mid)
        gpu.build_tlas(rr.make_instances(meshes=[mid]))
        s = O.Scene(); s.add_mesh(v, idx)
        scale = float(np.abs(pos).max())
        rays = np.zeros(300, rr.RAY_DTYPE)
        rays["origin"] = (rng.random((300, 3)).astype(np.float32) * 2 - 0.5) * np.float32(scale * 0.6) + np.array([0, 0, scale], np.float32)
        d = rng.standard_normal((300, 3)).astype(np.float32); d[:, 2] = -np.abs(d[:, 2]) - 1.0
        rays["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True)
        rays["tmin"] = 0.0; rays["tmax"] = np.float32(scale * 100)
        rays["flags"] = 0
        hits = gpu.trace_rays(rays)
        for k in range(300):
            h = s.trace(rays["origin"][k], rays["dir"][k], 0.0, float(rays["tmax"][k]), 0, use_bvh=0)
            assert bool(hits["hit"][k]) == bool(h.hit), (name, k)
            if h.hit:
                assert hits["t"][k] == np.float32(h.t) and hits["prim"][k] == h.prim, (name, k)


def test_builds_are_deterministic(gpu):
    """same mesh, two builds: identical fp32 and quantised hierarchies (the PLOC merge order comes from scans,
    the Karras tree from sorted unique keys; nothing depends on atomics order)"""
    m = load("ott.obj")
    got = []
    for _ in range(2):
        for fast_build in (False, True):
            mid = gpu.upload_mesh(m.verts, m.indices)
            gpu.build_blas(mid, fast_build=fast_build)
            nodes, tris = gpu.download_blas(mid)
            q, org, cell = gpu.download_qnodes(mid)
            got.append((nodes.tobytes(), tris.tobytes(), q.tobytes(), org.tobytes(), cell.tobytes()))
    assert got[0] == got[2] and got[1] == got[3] and got[0] != got[1]


@pytest.mark.parametrize("name", ["cube.obj", "monkey.obj", "ott.obj"])
def test_quantised_nodes_contain_the_fp32_boxes(gpu, name):
    """Traversal reads 32-byte nodes whose planes are fp16 cell counts on a grid over the mesh bounds.  The box test
    only has to be conservative: every stored child box must contain its fp32 box (evaluated in float64 from the same
    float32 grid the kernels use), by no more than the fp16 spacing there (<= 16 cells at the faces of the bounds) plus
    the guard cell; child refs are the same tree with internal refs as byte offsets."""
    m = load(name)
    mid = gpu.upload_mesh(m.verts, m.indices)
    gpu.build_blas(mid)
    nodes, _ = gpu.download_blas(mid)
    q, org, cell = gpu.download_qnodes(mid)
    assert len(q) == len(nodes) and np.all(cell > 0)
    org, cell = org.astype(np.float64), cell.astype(np.float64)
    for ax, (lo, hi) in enumerate((("lox", "hix"), ("loy", "hiy"), ("loz", "hiz"))):
        qlo = org[ax] + q[lo].astype(np.float64) * cell[ax]
        qhi = org[ax] + q[hi].astype(np.float64) * cell[ax]
        flo, fhi = nodes[lo].astype(np.float64), nodes[hi].astype(np.float64)
        real = flo <= fhi                                      # (a one-triangle mesh has an empty second child)
        assert np.all(np.abs(q[lo][real].astype(np.float64)) <= 32768) and np.all(np.abs(q[hi][real].astype(np.float64)) <= 32768)
        assert np.all(qlo[real] <= flo[real]) and np.all(qhi[real] >= fhi[real])
        assert np.all(flo[real] - qlo[real] <= 18 * cell[ax]) and np.all(qhi[real] - fhi[real] <= 18 * cell[ax])
    c, qc = nodes["c"], q["c"]
    assert np.array_equal(qc[c < 0], c[c < 0]) and np.array_equal(qc[c >= 0], c[c >= 0] * 32)
    # the grid spans the mesh bounds, origin at their centre
    P = m.verts["position"][m.indices].astype(np.float64)
    assert np.all(org - 32768 * cell <= P.min(0)) and np.all(org + 32768 * cell >= P.max(0))
    assert np.all(np.abs(org - (P.min(0) + P.max(0)) / 2) <= 8 * cell + 1e-6 * np.abs(org))


def test_fast_build_and_fast_trace_hierarchies_render_the_same_frame(gpu):
    m = load("monkey.obj")
    env = procedural_env(128, 64, seed=31)
    gpu.upload_envmap(env)
    gpu.set_tile_partition(0, 1)
    gpu.set_camera(rr.camera_orbit(0.9))
    frames, visits = [], []
    for fast_build in (False, True):
        mid = gpu.upload_mesh(m.verts, m.indices)
        gpu.build_blas(mid, fast_build=fast_build)
        gpu.build_tlas(rr.make_instances(meshes=[mid]))
        gpu.dispatch_rays(320, 180, rr.default_params(max_refract=8, flags=rr.DISPATCH_FLOAT_OUTPUT | rr.DISPATCH_COLLECT_STATS))
        frames.append(gpu.read_frame(want_float=True)[1].copy())
        visits.append(gpu.stats().node_visits)
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))     # the hierarchy never changes a pixel
    assert visits[0] < visits[1]                                                     # and FAST_TRACE does trace faster


def test_trace_rays_empty_and_single_triangle(gpu):
    # one triangle: the degenerate LBVH (no internal node from Karras)
    v = np.zeros(3, rr.VERTEX_DTYPE)
    v["position"] = [(0, 0, 0), (0, 1, 0), (0, 0, 1)]
    v["norm"] = (1, 0, 0)
    mid = gpu.upload_mesh(v, np.arange(3, dtype=np.uint32))
    gpu.build_blas(mid)
    gpu.build_tlas(rr.make_instances(meshes=[mid]))
    assert len(gpu.trace_rays(np.zeros(0, rr.RAY_DTYPE))) == 0
    rays = np.zeros(3, rr.RAY_DTYPE)
    rays["origin"] = [(2, 0.25, 0.25), (2, 0.25, 0.25), (-2, 0.25, 0.25)]
    rays["dir"] = [(-1, 0, 0), (-1, 0, 0), (1, 0, 0)]
    rays["tmin"], rays["tmax"] = 1e-4, 100.0
    rays["flags"] = [rr.RAY_FLAG_CULL_BACK, rr.RAY_FLAG_CULL_FRONT, rr.RAY_FLAG_CULL_BACK]
    h = gpu.trace_rays(rays)
    # cross(e1,e2) = (1,0,0); front face is seen from +x
    assert list(h["hit"]) == [1, 0, 0] and abs(h["t"][0] - 2.0) < 1e-6
    assert abs(h["u"][0] - 0.25) < 1e-6 and abs(h["v"][0] - 0.25) < 1e-6


# ------------------------------------------------------------------------------- BVH structure
@pytest.mark.parametrize("fast_build", [False, True])
@pytest.mark.parametrize("name", ["cube.obj", "monkey.obj", "ott.obj"])
def test_lbvh_structure(gpu, name, fast_build):
    """both builders (PREFER_FAST_TRACE = clustered PLOC tree, PREFER_FAST_BUILD = Karras radix tree)"""
    m = load(name)
    mid = gpu.upload_mesh(m.verts, m.indices)
    gpu.build_blas(mid, fast_build=fast_build)
    nodes, tris = gpu.download_blas(mid)
    T = len(m.indices) // 3
    assert len(tris) == T and len(nodes) == T - 1
    # leaves are a permutation of the primitives, records hold v0, v1-v0, v2-v0
    assert sorted(tris["prim"].tolist()) == list(range(T))
    P = m.verts["position"][m.indices].reshape(T, 3, 3)
    assert np.array_equal(tris["v0"], P[tris["prim"], 0])
    assert np.array_equal(tris["e1"], P[tris["prim"], 1] - P[tris["prim"], 0])
    assert np.array_equal(tris["e2"], P[tris["prim"], 2] - P[tris["prim"], 0])
    # every node is referenced exactly once, every leaf exactly once; child boxes are exact unions
    seen_nodes, seen_leaves = np.zeros(T - 1, int), np.zeros(T, int)
    seen_nodes[0] = 1
    lo = np.full((T - 1, 3), np.inf, np.float32)
    hi = np.full((T - 1, 3), -np.inf, np.float32)
    order = []
    stack = [0]
    while stack:
        n = stack.pop()
        order.append(n)
        for c in nodes["c"][n]:
            if c >= 0:
                seen_nodes[c] += 1
                stack.append(c)
            else:
                seen_leaves[~c] += 1
    assert np.all(seen_nodes == 1) and np.all(seen_leaves == 1)
    for n in reversed(order):
        for k in (0, 1):
            c = nodes["c"][n][k]
            l = np.array([nodes["lox"][n][k], nodes["loy"][n][k], nodes["loz"][n][k]], np.float32)
            h = np.array([nodes["hix"][n][k], nodes["hiy"][n][k], nodes["hiz"][n][k]], np.float32)
            if c >= 0:
                elo, ehi = lo[c], hi[c]
            else:
                tri = P[tris["prim"][~c]]
                elo, ehi = tri.min(0), tri.max(0)
            assert np.array_equal(l, elo) and np.array_equal(h, ehi), "node %d child %d" % (n, k)
            lo[n] = np.minimum(lo[n], l)
            hi[n] = np.maximum(hi[n], h)
    gpu.build_tlas(rr.make_instances(meshes=[mid]))
    assert 64 >= gpu.stats().bvh_depth >= int(np.ceil(np.log2(T)))


# ------------------------------------------------------------------------------- frames
FRAME_CASES = [
    # (mesh, W, H, angle, params)  -- BASELINE.json configs at sizes the oracle finishes in seconds
    ("sphere.obj", 256, 256, 0.01, dict(max_refract=1)),             # C1 exactly
    ("sphere.obj", 240, 136, 0.01, dict(max_refract=4)),             # C2 at 1/8 scale
    ("monkey.obj", 240, 136, 0.01, dict(max_refract=8)),             # C3 at 1/8 scale
    ("monkey.obj", 256, 192, 0.01, dict()),                          # reference literals (5 / 2)
    ("shell.obj", 256, 192, 0.01, dict()),                           # the mesh the demo loads, 4:3
    ("shell.obj", 200, 150, 2.5, dict(max_refract=8)),
    ("cube.obj", 256, 192, 0.77, dict()),
    ("cube.obj", 97, 61, 0.01, dict(max_refract=3, max_reflect=0)),  # ragged size, no reflections
    ("monkey.obj", 160, 120, 4.0, dict(max_refract=16, max_reflect=3)),   # parked-ray depth > 2
    ("ott.obj", 160, 120, 0.01, dict(max_refract=8)),
    ("monkey.obj", 33, 31, 1.0, dict(max_refract=0)),                # every hit is terminal -> black
    ("cube.obj", 1, 1, 0.01, dict()),                                # a single pixel
    ("monkey.obj", 2049, 3, 0.4, dict(max_refract=6)),               # wider than high: 65 tiles, one partial row and column
    ("shell.obj", 5, 600, 0.01, dict(max_refract=8, ior=1.5)),       # higher than wide, another index of refraction
    ("sphere.obj", 96, 96, 1.2, dict(max_refract=8, ior=0.8)),       # ior < 1: total internal reflection on entry
]


@pytest.mark.parametrize("name,W,H,angle,kw", FRAME_CASES)
def test_frame_parity(gpu, name, W, H, angle, kw):
    m = load(name)
    env = procedural_env(256, 128, seed=3)
    gpu_scene(gpu, [m], env)
    s = oracle_scene([m], env)
    check_frame(*render_both(gpu, s, angle, W, H, **kw))


@pytest.mark.parametrize("name", ["sphere.obj", "shell.obj", "cube.obj", "ott.obj"])
def test_frame_parity_on_the_symmetry_plane(gpu, name):
    """odd height: the middle pixel row has sy == 0, so its rays (and many of their children) run exactly
    in the y = 0 mirror plane of the mesh and hit shared triangle edges.  Checked against BRUTE FORCE."""
    m = load(name)
    env = procedural_env(128, 64, seed=23)
    gpu_scene(gpu, [m], env)
    s = oracle_scene([m], env)
    W, H = (160, 91) if name != "ott.obj" else (96, 55)
    sc = rr.camera_orbit(0.01)
    M, cam = np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32)
    gpu.set_tile_partition(0, 1)
    gpu.set_camera(sc)
    gpu.dispatch_rays(W, H, rr.default_params(max_refract=8, flags=rr.DISPATCH_FLOAT_OUTPUT))
    rgba, f32 = gpu.read_frame(want_float=True)
    ref = s.render(M, cam, W, H, O.default_params(use_bvh=0, max_refract=8, accum_mode=1))
    assert gpu.stats().rays == ref["stats"].rays
    assert np.array_equal(f32[..., :3].view(np.uint32), ref["rgb"].view(np.uint32))


def test_committed_golden_frames(gpu, env_png):
    """the committed fixtures of tests/golden/frames.npz (oracle, brute force, literal recursion)"""
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "frames.npz"))
    cases = [("cube.obj", 64, 48, dict()), ("sphere.obj", 64, 64, dict(max_refract=1)), ("monkey.obj", 96, 54, dict(max_refract=8)),
             ("shell.obj", 64, 48, dict()), ("ott.obj", 48, 36, dict(max_refract=8))]
    for name, w, h, kw in cases:
        m = load(name)
        gpu_scene(gpu, [m], env_png)
        gpu.set_tile_partition(0, 1)
        gpu.set_camera(rr.camera_orbit(0.01))
        gpu.dispatch_rays(w, h, rr.default_params(flags=rr.DISPATCH_FLOAT_OUTPUT | rr.DISPATCH_COLLECT_STATS, **kw))
        rgba, f32 = gpu.read_frame(want_float=True)
        st = gpu.stats()
        key = name.split(".")[0]
        assert [st.rays, st.hits, st.misses] == gold[key + "_counts"].tolist(), name
        assert np.abs(f32[..., :3] - gold[key + "_rgb"]).max() <= FLOAT_TOL, name
        assert np.abs(rgba.astype(int) - gold[key + "_rgba8"].astype(int)).max() <= 1, name


def test_frame_parity_envmap_png(gpu, env_png):
    m = load("shell.obj")
    gpu_scene(gpu, [m], env_png)
    s = oracle_scene([m], env_png)
    rgba, f32, st, lit, pw = render_both(gpu, s, 0.01, 256, 192)
    check_frame(rgba, f32, st, lit, pw)
    # the studio panorama is LDR: nothing saturates, the glass shell is visible
    assert 0.05 < (st.hits / st.rays) < 0.95


def test_stats_and_plain_kernels_agree_and_are_deterministic(gpu):
    m = load("monkey.obj")
    env = procedural_env(128, 64, seed=5)
    gpu_scene(gpu, [m], env)
    sc = rr.camera_orbit(0.3)
    gpu.set_camera(sc)
    frames = []
    for flags in (0, rr.DISPATCH_COLLECT_STATS, 0, rr.DISPATCH_FLOAT_OUTPUT):
        gpu.dispatch_rays(320, 200, rr.default_params(flags=flags, max_refract=8))
        frames.append(gpu.read_frame().copy())
        assert gpu.stats().rays > 320 * 200
    for f in frames[1:]:
        assert np.array_equal(frames[0], f)


def test_instanced_scene_parity(gpu):
    """Extension beyond the reference (one identity instance): TLAS over transformed instances,
    rotation + non-uniform scale + translation, per-instance cull flags."""
    cube, monkey = load("cube.obj"), load("monkey.obj")
    env = procedural_env(128, 64, seed=7)

    def xf(tx, ty, tz, s=(1, 1, 1), rot=0.0):
        c, sn = np.cos(rot), np.sin(rot)
        R = np.array([[c, 0, sn], [0, 1, 0], [-sn, 0, c]], np.float32) * np.array(s, np.float32)
        return np.concatenate([R, np.array([[tx], [ty], [tz]], np.float32)], axis=1)

    inst = rr.make_instances(
        transforms=[xf(0, 0, 0), xf(0, 0, -2.5, (0.5, 0.8, 0.5), 0.4), xf(0.3, 0.2, 2.4, (0.7, 0.7, 0.7), -1.0),
                    xf(0, 1.9, 0, (0.4, 0.4, 0.4), 0.2), xf(0, -1.8, 0.5, (0.5, 0.5, 0.5))],
        meshes=[1, 0, 1, 0, 0], masks=[1, 1, 0xff, 1, 0],
        flags=[0, 0, 0, 1, 0])       # instance 3: TRIANGLE_CULL_DISABLE (0x1); instance 4: InstanceMask 0
    gpu_scene(gpu, [cube, monkey], env, inst)
    s = oracle_scene([cube, monkey], env, inst)
    # TraceRay first: bit-exact incl. instance index
    rays = random_rays(4000, seed=11, radius=5.0)
    hits = gpu.trace_rays(rays)
    seen = set()
    for k in range(len(rays)):
        h = s.trace(rays["origin"][k], rays["dir"][k], float(rays["tmin"][k]), float(rays["tmax"][k]),
                    int(rays["flags"][k]), use_bvh=0)
        assert bool(hits["hit"][k]) == bool(h.hit), k
        if h.hit:
            seen.add(int(h.inst))
            assert hits["inst"][k] == h.inst and hits["prim"][k] == h.prim
            assert np.float32(hits["t"][k]).view(np.uint32) == np.float32(h.t).view(np.uint32)
    assert seen == {0, 1, 2, 3}                   # instance 4 has InstanceMask 0
    check_frame(*render_both(gpu, s, 0.6, 200, 150, max_refract=8))


# ------------------------------------------------------------------------------- background culling
def adversarial_constants(rng, kind, box_lo, box_hi):
    """SceneConstants as rr_set_camera accepts them (any proj_inv, any camera_loc): the orbit camera's constants pushed
    towards everything the host-side screen rectangle of the scene has to survive."""
    fov = np.deg2rad(rng.choice([1.0, 5.0, 30.0, 60.0, 95.0, 140.0, 170.0])) if kind == "fov" else rr.FOV_Y
    aspect = float(rng.choice([0.2, 1.0, 16.0 / 9.0, 5.0])) if kind in ("fov", "skew") else rr.ASPECT
    sc = rr.camera_orbit(float(rng.uniform(0.0, 6.28)), fov_y=float(fov), aspect=aspect)
    M = np.array(sc.proj_inv, np.float32).reshape(4, 4).copy()
    cam = np.array(sc.camera_loc, np.float32).copy()
    ctr, half = 0.5 * (box_lo + box_hi), 0.5 * (box_hi - box_lo)
    if kind == "radius":                    # from deep inside the bounds to far away, through the faces
        cam[:3] = cam[:3] * np.float32(rng.choice([0.0, 0.3, 0.7, 1.0, 1.5, 2.5, 4.0, 20.0]))
    elif kind == "on_bounds":               # on a corner / a face of the bounds, and a hair outside them
        sgn = rng.choice([-1.0, 1.0], 3)
        p = ctr + sgn * half * np.where(rng.random(3) < 0.5, 1.0, rng.uniform(0.0, 1.0, 3))
        cam[:3] = (p + sgn * rng.choice([0.0, 1e-6, 1e-3, 0.05])).astype(np.float32)
    elif kind in ("skew", "singular", "mirror", "random"):
        A = M[:3][:, [0, 1, 3]].astype(np.float64)
        if kind == "random":
            A = rng.normal(size=(3, 3))
        elif kind == "singular":            # singular values spread over 2..9 decades, in a random frame
            U, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            V, _ = np.linalg.qr(rng.normal(size=(3, 3)))
            spread = 10.0 ** rng.uniform(2.0, 9.0)
            A = A @ (U @ np.diag([1.0, spread ** -0.5, 1.0 / spread]) @ V.T)
        else:                               # shear + anisotropic scale of the screen axes (the scene stays in view), optionally mirrored
            S = np.eye(3) + rng.uniform(-0.35, 0.35, (3, 3)) * np.array([[1, 1, 0.3], [1, 1, 0.3], [0.2, 0.2, 0.2]])
            S = S @ np.diag([10.0 ** rng.uniform(-0.7, 0.7), 10.0 ** rng.uniform(-0.7, 0.7), 1.0])
            if kind == "mirror":
                S = S @ np.diag([[1.0, -1.0, 1.0], [-1.0, 1.0, 1.0], [-1.0, -1.0, 1.0]][int(rng.integers(3))])
            A = A @ S
        M[:3][:, [0, 1, 3]] = A.astype(np.float32)
        if rng.random() < 0.3:
            cam[:3] = cam[:3] * np.float32(rng.choice([0.5, 2.0, 8.0]))
    return rr.scene_constants(M, cam), M, cam


CULL_KINDS = ["fov", "radius", "on_bounds", "skew", "singular", "mirror", "random"]
CULL_SIZES = [(640, 360), (8, 8), (257, 131), (2049, 3), (800, 200), (64, 40), (333, 500), (3, 1025), (1200, 96)]


@pytest.mark.parametrize("scene", ["monkey", "sphere_small", "tlas"])
def test_background_culling_equals_tracing_every_primary_ray(gpu, scene):
    """k_render_fused / k_render_paths shade the 8x8 blocks outside the host-side screen rectangle of the scene bounds as one
    Miss, without TraceRay (RayTracing.hlsl:60 traces every pixel).  For a few hundred constants rr_set_camera accepts --
    fov 1..170 degrees, cameras inside / on / just outside the bounds, skewed, mirrored, near-singular and random proj_inv --
    the frame must equal the same dispatch with RR_DISPATCH_DEBUG_NO_CULL (every primary ray traced), Depth 1
    (k_render_paths where the rectangle is small) and Depth 3 (k_render_fused); a subset is checked against the oracle."""
    env = procedural_env(128, 64, seed=21)
    inst = None
    if scene == "monkey":
        meshes = [load("monkey.obj")]
    elif scene == "sphere_small":
        m = load("sphere.obj")
        v = m.verts.copy(); v["position"] = v["position"] * np.float32(0.15) + np.array([0.4, -0.2, 0.1], np.float32)
        m.verts = v
        meshes = [m]
    else:
        def xf(tx, ty, tz, s):
            return np.concatenate([np.eye(3, dtype=np.float32) * np.float32(s), np.array([[tx], [ty], [tz]], np.float32)], axis=1)
        meshes = [load("cube.obj"), load("monkey.obj")]
        inst = rr.make_instances(transforms=[xf(0, 0, 0, 0.6), xf(1.2, 0.3, -0.8, 0.3), xf(-0.9, -0.4, 0.7, 0.4)], meshes=[1, 0, 1])
    gpu_scene(gpu, meshes, env, inst)
    s = oracle_scene(meshes, env, inst)
    lo = np.min([m.verts["position"].min(axis=0) for m in meshes], axis=0).astype(np.float64)
    hi = np.max([m.verts["position"].max(axis=0) for m in meshes], axis=0).astype(np.float64)
    if inst is not None:
        lo, hi = np.array([-1.5, -1.0, -1.2]), np.array([1.6, 1.0, 1.2])
    rng = np.random.default_rng({"monkey": 1, "sphere_small": 2, "tlas": 3}[scene])
    gpu.set_tile_partition(0, 1)
    culled_cases = oracle_cases = 0
    culled_by_kind = dict.fromkeys(CULL_KINDS, 0)
    n_cases = 126
    for k in range(n_cases):
        kind = CULL_KINDS[k % len(CULL_KINDS)]
        W, H = CULL_SIZES[(k // len(CULL_KINDS)) % len(CULL_SIZES)]
        depth = 1 if (k // 3) % 2 == 0 else 3
        cams = [adversarial_constants(rng, kind, lo, hi) for _ in range(depth)]
        kw = dict(max_refract=int(rng.choice([0, 2, 6])), max_reflect=int(rng.choice([0, 2])))
        out = []
        for extra in (0, rr.DISPATCH_DEBUG_NO_CULL):
            p = rr.default_params(flags=rr.DISPATCH_FLOAT_OUTPUT | rr.DISPATCH_COLLECT_STATS | extra, **kw)
            if depth == 1:
                gpu.set_camera(cams[0][0]); gpu.dispatch_rays(W, H, p)
            else:
                gpu.dispatch_rays_batch(W, H, [c[0] for c in cams], p)
            frames = [gpu.read_frame(want_float=True, slice=f) for f in range(depth)]
            out.append((frames, gpu.stats()))
        (fc, stc), (fn, stn) = out
        # (8x8 blocks of the 32x32 tiles that lie beyond the frame's edge count as background blocks in both runs)
        beyond = (((W + 31) // 32) * ((H + 31) // 32) * 16 - ((W + 7) // 8) * ((H + 7) // 8)) * depth
        assert stn.background_waves == beyond or stn.render_kernel == 2
        # the culled branch ran: the kernel counted background blocks inside the frame, or the launch went to k_render_paths
        # (render_kernel 2: it only ever renders the rectangle's blocks with TraceRay)
        culled = stc.background_waves > beyond or (stc.render_kernel == 2 and stn.render_kernel != 2)
        culled_cases += culled
        culled_by_kind[kind] += culled
        tag = "%s case %d: %s %dx%d depth %d %s" % (scene, k, kind, W, H, depth, kw)
        assert stc.rays == stn.rays and (stc.hits, stc.misses, stc.terminal_hits, stc.tir) == (stn.hits, stn.misses, stn.terminal_hits, stn.tir), tag
        for f in range(depth):
            assert np.array_equal(fc[f][0], fn[f][0]), tag
            assert np.array_equal(fc[f][1].view(np.uint32), fn[f][1].view(np.uint32)), tag
        if W * H <= 40000 and k % 4 == 0:       # the oracle on the same constants (path-weight mode: the kernel's summation order)
            M, cam = cams[0][1].reshape(16), cams[0][2]
            pw = s.render(M, cam, W, H, O.default_params(use_bvh=1, accum_mode=1, **kw))
            ok = np.isfinite(pw["rgb"]).all()
            if ok:
                assert np.array_equal(fc[0][1][..., :3].view(np.uint32), pw["rgb"].view(np.uint32)), tag
            assert np.array_equal(fc[0][0], pw["rgba8"]), tag
            oracle_cases += 1
    # the comparison only means something where the culled branch really ran: in a fifth of the cases at least, and for every
    # kind of constants that leaves the scene in front of the camera (fov, distance, skew, mirrored)
    assert culled_cases >= n_cases // 5, "the culled branch ran in only %d of %d cases: %s" % (culled_cases, n_cases, culled_by_kind)
    for kind in ("fov", "radius", "skew", "mirror"):
        assert culled_by_kind[kind] >= 3, culled_by_kind
    assert oracle_cases >= 10


# ------------------------------------------------------------------------------- sharding
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_tiles_reassemble_to_the_single_gpu_frame(gpu, world):
    """Every rank's compact tile buffer, gathered and de-interleaved by rr_assemble_tiles, equals
    the world==1 frame byte for byte (one GPU plays all ranks in turn)."""
    import torch
    m = load("monkey.obj")
    env = procedural_env(128, 64, seed=9)
    gpu_scene(gpu, [m], env)
    W, H = 250, 130                                   # ragged: partial tiles on both edges
    gpu.set_camera(rr.camera_orbit(0.01))
    gpu.set_tile_partition(0, 1)
    gpu.dispatch_rays(W, H, rr.default_params(max_refract=8))
    full = gpu.read_frame().copy()
    full_rays = gpu.stats().rays
    mx = rr.dist.max_local_tiles(W, H, world)
    gathered = torch.zeros(world * mx * rr.dist.TILE_BYTES, dtype=torch.uint8, device="cuda:0")
    rays = 0
    for rank in range(world):
        gpu.set_tile_partition(rank, world)
        n, mx2 = gpu.local_tile_count(W, H)
        assert mx2 == mx and n == len(rr.dist.local_tiles(W, H, rank, world))
        gpu.dispatch_rays(W, H, rr.default_params(max_refract=8))
        gpu.export_tiles(gathered.data_ptr() + rank * mx * rr.dist.TILE_BYTES)
        gpu.wait()
        rays += gpu.stats().rays
    assert rays == full_rays
    gpu.assemble_tiles(gathered.data_ptr(), world)
    assert np.array_equal(gpu.read_frame(), full)
    assert np.array_equal(rr.dist.assemble_host(gathered.cpu().numpy(), W, H, world), full)
    gpu.set_tile_partition(0, 1)


# ------------------------------------------------------------------------------- full size
def test_full_size_properties_1080p_monkey(gpu):
    """BASELINE config C3 at full size: properties that need no oracle run."""
    m = load("monkey.obj")
    env = procedural_env(2048, 1024, seed=0)
    gpu_scene(gpu, [m], env)
    W, H = 1920, 1080
    gpu.set_tile_partition(0, 1)
    gpu.set_camera(rr.camera_orbit(0.01))
    p = rr.default_params(max_refract=8, flags=rr.DISPATCH_COLLECT_STATS | rr.DISPATCH_FLOAT_OUTPUT)
    gpu.dispatch_rays(W, H, p)
    rgba, f32 = gpu.read_frame(want_float=True)
    st = gpu.stats()
    assert st.traversal_overflow == 0 and st.pixels == W * H and st.primary == W * H
    assert st.hits + st.misses == st.rays
    # 1.368 rays/pixel on monkey at limit 8 (resolution independent)
    assert abs(st.rays / (W * H) - 1.368) < 0.01
    # every background pixel equals the env lookup of its primary ray: spot-check corners via the oracle's Miss
    s = oracle_scene([m], env)
    sc = rr.camera_orbit(0.01)
    M, cam = np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32)
    for (x, y) in [(0, 0), (W - 1, 0), (0, H - 1), (W - 1, H - 1), (100, 1000)]:
        _, d = O.camera_ray(M, cam, x, y, W, H)
        assert np.array_equal(f32[y, x, :3], s.env_lookup(d))
    # a 96x64 window through the middle of Suzanne, against the oracle at full-frame addressing
    x0, y0 = 912, 508
    lit = s.render(M, cam, W, H, O.default_params(use_bvh=1, max_refract=8, accum_mode=1),
                   region=(x0, y0, x0 + 96, y0 + 64))
    assert np.array_equal(f32[y0:y0 + 64, x0:x0 + 96, :3].view(np.uint32), lit["rgb"][y0:y0 + 64, x0:x0 + 96].view(np.uint32))
    # idempotence
    gpu.dispatch_rays(W, H, rr.default_params(max_refract=8))
    assert np.array_equal(gpu.read_frame(), rgba)


# ------------------------------------------------------------------------------- error behaviour
def test_error_behaviour():
    r = rr.Renderer(0)
    with pytest.raises(rr.RRError) as e:
        r.dispatch_rays(64, 64)
    assert e.value.status == 5                                   # RR_ERR_STATE: nothing built
    with pytest.raises(rr.RRError):
        r.build_blas(7)
    v = np.zeros(3, rr.VERTEX_DTYPE)
    with pytest.raises(rr.RRError):
        r.upload_mesh(v, np.array([0, 1, 5], np.uint32))         # index out of range
    with pytest.raises(rr.RRError):
        r.upload_mesh(v, np.array([0, 1], np.uint32))            # not a triangle list
    for val in (np.inf, np.nan, 1e20):                           # a position the builder's box areas cannot hold: an error, not a hung build
        bad = np.zeros(6, rr.VERTEX_DTYPE)
        bad["position"] = np.random.default_rng(1).random((6, 3)).astype(np.float32)
        bad["position"][4, 2] = val
        with pytest.raises(rr.RRError) as e:
            r.upload_mesh(bad, np.arange(6, dtype=np.uint32))
        assert e.value.status == 1
    with pytest.raises(rr.RRError):
        r.set_tile_partition(2, 2)
    with pytest.raises(rr.RRError):
        rr.Renderer(99)
    m = load("cube.obj")
    r.load_scene(m.verts, m.indices, procedural_env(16, 8))
    with pytest.raises(rr.RRError):
        r.dispatch_rays(64, 64)                                  # camera not set
    r.set_camera(rr.camera_orbit(0.01))
    with pytest.raises(rr.RRError):
        r.dispatch_rays(64, 64, rr.default_params(max_reflect=9))
    with pytest.raises(rr.RRError):
        r.dispatch_rays(0, 64)
    r.dispatch_rays(64, 64)
    with pytest.raises(rr.RRError):
        r.read_frame(want_float=True)                            # float output was not requested
    assert r.read_frame().shape == (64, 64, 4)
    r.close()


def test_refraction_demo_mirror(tmp_path, env_png):
    """RefractionDemo::initialize / drawFrame with the reference's literals (shell.obj, 1024x768),
    env map through the .hdr path; three frames advance the orbit angle by 0.01 each."""
    hdr = tmp_path / "envMap.hdr"
    rr.write_hdr(hdr, env_png)
    demo = rr.RefractionDemo()
    demo.initialize(1024, 768, mesh_path=O.asset("shell.obj"), env_path=str(hdr))
    f0 = demo.drawFrame().copy()
    f1 = demo.drawFrame().copy()
    assert abs(demo.angle - 0.03) < 1e-6
    assert f0.shape == (768, 1024, 4) and not np.array_equal(f0, f1)
    # frame 0 == a direct render at angle 0.01 against the oracle (window through the shell)
    env_rt, _ = rr.load_texture(str(hdr), 3)
    m = load("shell.obj")
    s = oracle_scene([m], env_rt)
    sc = rr.camera_orbit(0.01)
    M, cam = np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32)
    x0, y0 = 480, 352
    lit = s.render(M, cam, 1024, 768, O.default_params(use_bvh=1), region=(x0, y0, x0 + 64, y0 + 64))
    assert np.abs(f0[y0:y0 + 64, x0:x0 + 64].astype(int) - lit["rgba8"][y0:y0 + 64, x0:x0 + 64].astype(int)).max() <= 1
    demo.renderer.close()


# ------------------------------------------------------------------------------- depth slices
@pytest.mark.parametrize("name", ["monkey.obj", "ott.obj", "sub2"])
def test_batched_dispatch_equals_single_dispatches(gpu, name):
    """DispatchRays(W,H,Depth): every slice is byte-identical to a Depth=1 dispatch with the same constants,
    and the ray counter is the sum.  A Depth-1 launch runs the 5-waves-per-SIMD build; the batch runs the build
    for the tree's depth class: 8 waves with 32-bit stack entries (monkey.obj, 18 levels), 16-bit entries (ott.obj,
    31 levels; the subdivided monkey, 24) -- all three must agree with the oracle-checked single dispatches."""
    if name == "sub2":
        from refraction_raytracing_dxr_amd.synth import subdivide
        m = rr.Mesh()
        m.verts, m.indices = subdivide(load("monkey.obj").verts, 2)
    else:
        m = load(name)
    env = procedural_env(128, 64, seed=13)
    gpu_scene(gpu, [m], env)
    gpu.set_tile_partition(0, 1)
    W, H = 300, 170
    cams = [rr.camera_orbit(0.01 * (k + 1) + 0.5 * k) for k in range(5)]
    p = rr.default_params(max_refract=8, flags=rr.DISPATCH_FLOAT_OUTPUT)
    singles, rays = [], 0
    for c in cams:
        gpu.set_camera(c)
        gpu.dispatch_rays(W, H, p)
        singles.append([x.copy() for x in gpu.read_frame(want_float=True)])
        rays += gpu.stats().rays
    gpu.dispatch_rays_batch(W, H, cams, p)
    assert gpu.stats().rays == rays and gpu.stats().pixels == 5 * W * H
    for k in range(5):
        rgba, f32 = gpu.read_frame(want_float=True, slice=k)
        assert np.array_equal(rgba, singles[k][0])
        assert np.array_equal(f32.view(np.uint32), singles[k][1].view(np.uint32))
    with pytest.raises(rr.RRError):
        gpu.read_frame(slice=5)


def test_orbit_loop_matches_draw_frame_sequence(gpu):
    """rr_render_orbit (the drawFrame loop in C) with 1 and with 4 frames per dispatch == explicit frames."""
    m = load("shell.obj")
    env = procedural_env(128, 64, seed=17)
    gpu_scene(gpu, [m], env)
    W, H = 256, 192
    p = rr.default_params()
    ref = []
    a = np.float32(0.01)
    for k in range(6):
        gpu.set_camera(rr.camera_orbit(a))
        gpu.dispatch_rays(W, H, p)
        ref.append(gpu.read_frame().copy())
        a = np.float32(a + np.float32(0.01))
    nxt = gpu.render_orbit(W, H, 6, params=p, frames_per_dispatch=1)
    assert abs(nxt - 0.07) < 1e-6
    assert np.array_equal(gpu.read_frame(), ref[5])
    gpu.render_orbit(W, H, 6, params=p, frames_per_dispatch=4)          # launches of 4 + 2 slices
    assert np.array_equal(gpu.read_frame(slice=0), ref[4]) and np.array_equal(gpu.read_frame(slice=1), ref[5])
    total = gpu.stats().rays
    gpu.render_orbit(W, H, 6, params=p, frames_per_dispatch=6)
    assert gpu.stats().rays == total
    for k in range(6):
        assert np.array_equal(gpu.read_frame(slice=k), ref[k])
    # overlapping launches (frames in flight): same frames, same counters, every lane and output region reused
    for fl in (2, 3, 4):
        gpu.set_frames_in_flight(fl)
        gpu.render_orbit(W, H, 6, params=p, frames_per_dispatch=1)
        assert np.array_equal(gpu.read_frame(), ref[5]) and gpu.stats().rays == total
        gpu.render_orbit(W, H, 6, params=p, frames_per_dispatch=4)
        assert np.array_equal(gpu.read_frame(slice=0), ref[4]) and np.array_equal(gpu.read_frame(slice=1), ref[5])
        assert gpu.stats().rays == total
        gpu.render_orbit(W, H, 5, params=rr.default_params(flags=rr.DISPATCH_FLOAT_OUTPUT), frames_per_dispatch=2)
        rgba, f32 = gpu.read_frame(want_float=True)
        assert np.array_equal(rgba, ref[4]) and np.all(f32[..., 3] == 1.0)
    gpu.set_frames_in_flight(1)
    with pytest.raises(rr.RRError):
        gpu.set_frames_in_flight(5)


def test_orbit_streamed_to_host_equals_draw_frame_sequence(gpu):
    """rr_render_orbit_to_host: every frame of the loop lands in host memory (copy of one region overlapping the next
    launch); same bytes as dispatch + read_frame per frame, for several batch sizes incl. a ragged last batch."""
    m = load("shell.obj")
    env = procedural_env(128, 64, seed=23)
    gpu_scene(gpu, [m], env)
    W, H, K = 200, 150, 11
    p = rr.default_params()
    ref = []
    a = np.float32(0.01)
    for k in range(K):
        gpu.set_camera(rr.camera_orbit(a))
        gpu.dispatch_rays(W, H, p)
        ref.append(gpu.read_frame().copy())
        a = np.float32(a + np.float32(0.01))
    ref = np.stack(ref)
    for F, fl, pin in ((1, 1, True), (3, 2, True), (4, 3, False), (16, 1, True)):
        gpu.set_frames_in_flight(fl)
        got = gpu.render_orbit_to_host(W, H, K, params=p, frames_per_dispatch=F, pin=pin)
        assert got.shape == ref.shape and np.array_equal(got, ref), (F, fl)
    gpu.set_frames_in_flight(1)
    assert np.array_equal(gpu.read_frame(slice=(K - 1) % 16), ref[-1])          # the last launch is still readable


# ------------------------------------------------------------------------------- N > 1 pipeline
def _sharded_worker(rank, world, port, backend, out, rgb8=True, W=250, mesh=False):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)                       # every rank shares the one card of the test box
    dist.init_process_group(backend, rank=rank, world_size=world)
    m = load("monkey.obj")
    env = procedural_env(128, 64, seed=21)
    r = rr.Renderer(0)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    r.load_scene(m.verts, m.indices, env)
    H, K, F = 130, 13, 2                           # 7 batches: every buffer set and both lanes are reused
    # world 1 under nccl: still issue the RCCL gather (async_op, views of the ring buffers), as the N > 1 ranks do
    sf = rr.dist.ShardedFrames(r, W, H, rank, world, torch.device("cuda", 0), frames_per_gather=F, rgb8=rgb8,
                               always_collective=(backend == "nccl"), mesh_partition=mesh)
    seen = []
    rays = sf.render_orbit(K, angle=0.01, params=rr.default_params(max_refract=8),
                           on_frames=(lambda fr: seen.append(fr.clone())) if rank == 0 else None)
    tot = torch.tensor([rays], dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tot)
    if rank == 0:
        got = sf.frames_host()                     # the last batch: frame 12 (K=13, F=2 -> batches 2,2,2,2,2,2,1)
        allf = torch.cat(seen).cpu().numpy()
        r.set_tile_partition(0, 1)
        a = np.float32(0.01)
        ref_rays = 0
        frames = []
        for k in range(K):
            r.set_camera(rr.camera_orbit(a))
            r.dispatch_rays(W, H, rr.default_params(max_refract=8))
            frames.append(r.read_frame().copy())
            ref_rays += r.stats().rays
            a = np.float32(a + np.float32(0.01))
        ok = (len(got) == 1 and np.array_equal(got[0], frames[12]) and int(tot.item()) == ref_rays
              and len(allf) == K and all(np.array_equal(allf[k], frames[k]) for k in range(K)))
        np.save(out, np.array([int(ok), len(got), int(tot.item()), ref_rays]))
    if world > 1:
        dist.barrier()
    r.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,backend,rgb8,W,mesh", [(1, "nccl", True, 250, False), (2, "gloo", True, 256, False), (2, "gloo", False, 250, False),
                                                       (1, "nccl", True, 256, True), (2, "gloo", True, 250, True), (3, "gloo", True, 256, True)])
def test_sharded_frames_pipeline(tmp_path, world, backend, rgb8, W, mesh):
    """render -> (RCCL | gloo-staged) gather of F frames -> rr_assemble_frames[_rgb8], pipelined over batches,
    equals frame-by-frame single-GPU rendering.  world 2 / 3 run that many processes on the one card; tiles travel as
    RGB8 (the default; W = 256 takes the 16-byte-store path of the de-interleave, W = 250 the ragged one) or RGBA8.
    mesh: the mesh-tile partition (rr_mesh_partition) -- only the tiles that touch the scene's screen rectangle are dealt
    and gathered, rank 0 renders the background tiles itself; the rectangle moves from batch to batch with the orbit."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "ok.npy")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, backend, out, rgb8, W, mesh)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    ok = np.load(out)
    assert ok[0] == 1, ok


def _rotating_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = load("monkey.obj")
    r = rr.Renderer(0)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    r.load_scene(m.verts, m.indices, procedural_env(128, 64, seed=21))
    W, H, K, F = 256, 130, 11, 2                   # batches 0..5: even ones end on rank 0, odd ones on rank 1
    sf = rr.dist.ShardedFrames(r, W, H, rank, world, torch.device("cuda", 0), frames_per_gather=F, rotate_root=True)
    seen = []
    sf.render_orbit(K, angle=0.01, params=rr.default_params(max_refract=8), on_frames=lambda fr: seen.append(fr.clone()))
    mine = torch.cat(seen).cpu().numpy()
    r.set_tile_partition(0, 1)
    a = np.float32(0.01)
    frames = []
    for k in range(K):
        r.set_camera(rr.camera_orbit(a))
        r.dispatch_rays(W, H, rr.default_params(max_refract=8))
        frames.append(r.read_frame().copy())
        a = np.float32(a + np.float32(0.01))
    expect = [k for k in range(K) if (k // F) % world == rank]
    ok = len(mine) == len(expect) and all(np.array_equal(mine[i], frames[k]) for i, k in enumerate(expect))
    np.save(os.path.join(out_dir, "rot%d.npy" % rank), np.array([int(ok), len(mine), len(expect)]))
    dist.barrier()
    r.close()
    dist.destroy_process_group()


def test_sharded_frames_rotating_root(tmp_path):
    """rotate_root: batch b is gathered to and assembled on rank b % world; every rank ends up with exactly its
    batches, bit-identical to single-GPU frames (two processes on the one card, gloo)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rotating_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    for r in range(2):
        ok = np.load(tmp_path / ("rot%d.npy" % r))
        assert ok[0] == 1 and ok[1] == ok[2] > 0, ok


# ------------------------------------------------------------------------------- BASELINE configs 4 and 5
def _xf(tx, ty, tz, s=1.0):
    m = np.eye(4, dtype=np.float32)[:3] * np.float32(s)
    m[:, 3] = (tx, ty, tz)
    return m


def test_config4_multi_blas_scene(gpu):
    """C4: shell (origin) + cube (-4,0,0) + ott (+4,0,0), three BLASes under one TLAS, 8 bounces
    (placement is the builder's choice).  Parity at 1/16 of 3840x2160."""
    meshes = [load("shell.obj"), load("cube.obj"), load("ott.obj")]
    env = procedural_env(256, 128, seed=4)
    inst = rr.make_instances(transforms=[_xf(0, 0, 0), _xf(0, 0, -4.0), _xf(0, 0, 4.0)], meshes=[0, 1, 2])
    gpu_scene(gpu, meshes, env, inst)
    s = oracle_scene(meshes, env, inst)
    check_frame(*render_both(gpu, s, 0.01, 240, 135, max_refract=8))


def test_config5_instanced_grid(gpu):
    """C5 in miniature: monkey.obj instanced on an 8x8 grid in the XZ plane, pitch 3 (TLAS stress), 16 bounces."""
    m = load("monkey.obj")
    env = procedural_env(256, 128, seed=5)
    xs = [_xf(3.0 * (i - 3.5), 0, 3.0 * (j - 3.5), 0.9) for i in range(8) for j in range(8)]
    inst = rr.make_instances(transforms=xs, meshes=[0] * 64)
    gpu_scene(gpu, [m], env, inst)
    s = oracle_scene([m], env, inst)
    # camera pulled back by scaling the constants' translation: use a wider orbit radius via proj_inv of angle 0.7
    sc = rr.camera_orbit(0.7)
    sc.camera_loc[0] *= 5.0
    sc.camera_loc[2] *= 5.0
    M, cam = np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32)
    gpu.set_tile_partition(0, 1)
    gpu.set_camera(sc)
    kw = dict(max_refract=16)
    gpu.dispatch_rays(200, 112, rr.default_params(flags=rr.DISPATCH_FLOAT_OUTPUT | rr.DISPATCH_COLLECT_STATS, **kw))
    rgba, f32 = gpu.read_frame(want_float=True)
    st = gpu.stats()
    lit = s.render(M, cam, 200, 112, O.default_params(use_bvh=1, **kw))
    pw = s.render(M, cam, 200, 112, O.default_params(use_bvh=1, accum_mode=1, **kw))
    check_frame(rgba, f32, st, lit, pw)
    assert st.hits > 2000                      # the grid is in view


def test_full_size_config5_properties(gpu):
    """C5 at full size (monkey x 1024, 3840x2160, 16 bounces): runs, no stack overflow, deterministic,
    counters consistent; a window is checked against the oracle."""
    m = load("monkey.obj")
    env = procedural_env(512, 256, seed=6)
    xs = [_xf(3.0 * (i - 15.5), 0, 3.0 * (j - 15.5)) for i in range(32) for j in range(32)]
    inst = rr.make_instances(transforms=xs, meshes=[0] * 1024)
    gpu_scene(gpu, [m], env, inst)
    sc = rr.camera_orbit(0.4)
    for k in (0, 2):
        sc.camera_loc[k] *= 14.0
    sc.camera_loc[1] = 12.0
    W, H = 3840, 2160
    gpu.set_tile_partition(0, 1)
    gpu.set_camera(sc)
    p = rr.default_params(max_refract=16, flags=rr.DISPATCH_COLLECT_STATS)
    gpu.dispatch_rays(W, H, p)
    a = gpu.read_frame().copy()
    st = gpu.stats()
    assert st.traversal_overflow == 0 and st.pixels == W * H and st.hits + st.misses == st.rays
    gpu.dispatch_rays(W, H, rr.default_params(max_refract=16))
    assert np.array_equal(gpu.read_frame(), a)
    s = oracle_scene([m], env, inst)
    M, cam = np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32)
    x0, y0 = 1900, 1000
    ref = s.render(M, cam, W, H, O.default_params(use_bvh=1, max_refract=16, accum_mode=1), region=(x0, y0, x0 + 48, y0 + 32))
    assert np.array_equal(a[y0:y0 + 32, x0:x0 + 48], ref["rgba8"][y0:y0 + 32, x0:x0 + 48])


def test_full_size_config4_properties(gpu):
    """C4 at full size (shell + cube + ott, three BLASes under one TLAS, 3840x2160, 8 bounces; placement as in
    tools/exp_configs.py and bench.py): runs, no stack overflow, deterministic, counters consistent, the same frame from a
    16-slice DispatchRays and from a single one; windows on each of the three meshes are checked against the oracle."""
    meshes = [load("shell.obj"), load("cube.obj"), load("ott.obj")]
    env = procedural_env(512, 256, seed=7)
    inst = rr.make_instances(transforms=[_xf(0, 0, 0), _xf(0, 0, -4.0), _xf(0, 0, 4.0)], meshes=[0, 1, 2])
    gpu_scene(gpu, meshes, env, inst)
    sc = rr.camera_orbit(0.01)
    for k in (0, 2):
        sc.camera_loc[k] *= 1.6
    W, H = 3840, 2160
    gpu.set_tile_partition(0, 1)
    gpu.set_camera(sc)
    gpu.dispatch_rays(W, H, rr.default_params(max_refract=8, flags=rr.DISPATCH_COLLECT_STATS))
    a = gpu.read_frame().copy()
    st = gpu.stats()
    assert st.traversal_overflow == 0 and st.pixels == W * H and st.hits + st.misses == st.rays and st.rays > 2 * W * H
    gpu.dispatch_rays(W, H, rr.default_params(max_refract=8))
    assert np.array_equal(gpu.read_frame(), a)
    gpu.dispatch_rays_batch(W, H, [sc] * 3, rr.default_params(max_refract=8))             # the batched launch shape
    assert np.array_equal(gpu.read_frame(slice=2), a)
    s = oracle_scene(meshes, env, inst)
    M, cam = np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32)
    covered = a[..., :3].astype(int).sum(axis=2)
    for x0, y0 in ((1900, 1060), (700, 1060), (3000, 1000)):                             # shell, one side mesh, the other
        ref = s.render(M, cam, W, H, O.default_params(use_bvh=1, max_refract=8, accum_mode=1), region=(x0, y0, x0 + 40, y0 + 24))
        assert np.array_equal(a[y0:y0 + 24, x0:x0 + 40], ref["rgba8"][y0:y0 + 24, x0:x0 + 40]), (x0, y0)
        assert ref["stats"].rays >= 40 * 24


def test_rrdemo_cli(tmp_path, env_png):
    """the headless WinMain replacement: C++ host (Mesh, RefractionDemo::initialize/drawFrame) end to end"""
    import subprocess
    exe = os.path.join(os.path.dirname(rr.lib_path()), "rrdemo")
    assert os.path.exists(exe), "rrdemo was not built"
    hdr = tmp_path / "envMap.hdr"
    rr.write_hdr(hdr, env_png)
    out = subprocess.run([exe, "--mesh", O.asset("shell.obj"), "--env", str(hdr), "--size", "256x192", "--frames", "2",
                          "--out", str(tmp_path / "f_%03d.ppm")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    raw = open(tmp_path / "f_000.ppm", "rb").read()
    assert raw.startswith(b"P6\n256 192\n255\n")
    img = np.frombuffer(raw[len(b"P6\n256 192\n255\n"):], np.uint8).reshape(192, 256, 3)
    env_rt, _ = rr.load_texture(str(hdr), 3)
    m = load("shell.obj")
    s = oracle_scene([m], env_rt)
    sc = rr.camera_orbit(0.01)
    ref = s.render(np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32), 256, 192, O.default_params(use_bvh=1))
    assert np.abs(img.astype(int) - ref["rgba8"][..., :3].astype(int)).max() <= 1
    # the loop without per-frame read-back (rr_render_orbit, overlapping launches): last frame == drawFrame's 7th
    seq = subprocess.run([exe, "--mesh", O.asset("shell.obj"), "--env", str(hdr), "--size", "256x192", "--frames", "7",
                          "--out", str(tmp_path / "s_%03d.ppm")], capture_output=True, text=True, timeout=120)
    pump = subprocess.run([exe, "--mesh", O.asset("shell.obj"), "--env", str(hdr), "--size", "256x192", "--frames", "7", "--pump",
                           "--frames-per-dispatch", "2", "--in-flight", "3", "--out", str(tmp_path / "p_%03d.ppm")],
                          capture_output=True, text=True, timeout=120)
    assert seq.returncode == 0 and pump.returncode == 0, pump.stderr
    assert "7 frames of 256x192" in pump.stdout and "3 in flight" in pump.stdout
    assert open(tmp_path / "p_006.ppm", "rb").read() == open(tmp_path / "s_006.ppm", "rb").read()
    # every frame streamed to host memory with overlapped copies: the same seven frames as drawFrame's
    strm = subprocess.run([exe, "--mesh", O.asset("shell.obj"), "--env", str(hdr), "--size", "256x192", "--frames", "7", "--stream",
                           "--frames-per-dispatch", "3", "--in-flight", "2", "--out", str(tmp_path / "t_%03d.ppm")],
                          capture_output=True, text=True, timeout=120)
    assert strm.returncode == 0 and "fps delivered to" in strm.stdout, strm.stderr
    for k in range(7):
        assert open(tmp_path / ("t_%03d.ppm" % k), "rb").read() == open(tmp_path / ("s_%03d.ppm" % k), "rb").read()
    bad = subprocess.run([exe, "--mesh", str(tmp_path / "missing.obj"), "--env", str(hdr)], capture_output=True, text=True)
    assert bad.returncode == 1 and "mesh could not be loaded" in bad.stderr
    # the sharded C++ host, one process per GPU (here: one): tiles -> rr_gather_frames (RCCL, grouped send/recv) ->
    # rr_assemble_frames; the launcher process itself never touches the GPU.  Same seventh frame as drawFrame's.
    sh = subprocess.run([exe, "--mesh", O.asset("shell.obj"), "--env", str(hdr), "--size", "256x192", "--frames", "7", "--gpus", "1",
                         "--frames-per-gather", "3", "--out", str(tmp_path / "g_%03d.ppm")], capture_output=True, text=True, timeout=300)
    assert sh.returncode == 0 and "rank 0 of 1: 7 frames of 256x192" in sh.stdout, sh.stderr + sh.stdout
    assert open(tmp_path / "g_006.ppm", "rb").read() == open(tmp_path / "s_006.ppm", "rb").read()


def test_native_rccl_gather_entry_points(gpu):
    """rr_comm_* / rr_gather_frames / rr_device_* through the C ABI with a communicator of one rank: the tiles of three
    frames are gathered (a grouped ncclSend + ncclRecv to itself), de-interleaved by rr_assemble_frames and read back --
    byte for byte the frames of an unsharded dispatch.  (More ranks need more GPUs: the driver's 8-GPU run.)"""
    L = rr.lib()
    m = load("monkey.obj")
    gpu_scene(gpu, [m], procedural_env(64, 32, seed=3))
    W, H, F = 200, 120, 3
    p = rr.default_params(max_refract=4)
    gpu.set_tile_partition(0, 1)
    gpu.render_orbit(W, H, F, angle=0.2, params=p, frames_per_dispatch=F)
    want = [gpu.read_frame(slice=k).copy() for k in range(F)]
    ident = (C.c_ubyte * 128)()
    assert L.rr_comm_unique_id(ident) == 0
    comm = C.c_void_p()
    gpu._ck(L.rr_comm_init(gpu._h, ident, 0, 1, C.byref(comm)), "rr_comm_init")
    n_local, max_tiles = gpu.local_tile_count(W, H)
    stride = max_tiles * 32 * 32 * 4
    d_send, d_recv, d_frames = C.c_void_p(), C.c_void_p(), C.c_void_p()
    for ptr, nbytes in ((d_send, F * stride), (d_recv, F * stride), (d_frames, F * W * H * 4)):
        gpu._ck(L.rr_device_alloc(gpu._h, nbytes, C.byref(ptr)), "rr_device_alloc")
    gpu.render_orbit_sharded(W, H, F, d_send.value, stride, angle=0.2, params=p, frames_per_dispatch=F)
    gpu._ck(L.rr_gather_frames(gpu._h, comm, 0, 1, d_send, d_recv, F * stride, 0), "rr_gather_frames")
    gpu.assemble_frames(d_recv.value, 1, F * stride, stride, F, W, H, d_frames.value, W * H * 4)
    got = np.empty((F, H, W, 4), np.uint8)
    gpu._ck(L.rr_device_read(gpu._h, d_frames, got.ctypes.data, got.nbytes), "rr_device_read")
    for k in range(F):
        assert np.array_equal(got[k], want[k]), k
    with pytest.raises(rr.RRError):
        gpu._ck(L.rr_gather_frames(gpu._h, comm, 1, 1, d_send, d_recv, 16, 0), "rr_gather_frames")      # rank >= world
    for ptr in (d_send, d_recv, d_frames):
        gpu._ck(L.rr_device_free(gpu._h, ptr), "rr_device_free")
    assert L.rr_comm_destroy(comm) == 0


# ------------------------------------------------------------------------------- large mesh (builder at scale)
def procedural_mesh(n_side, seed=0):
    """bumpy unit-ish sphere patch grid: 2*n_side*n_side triangles with smooth normals"""
    rng = np.random.default_rng(seed)
    u, v = np.meshgrid(np.linspace(0.02, np.pi - 0.02, n_side + 1), np.linspace(0, 2 * np.pi, n_side + 1), indexing="ij")
    rad = 1.0 + 0.08 * np.sin(7 * u) * np.cos(5 * v) + 0.01 * rng.standard_normal(u.shape)
    P = np.stack([rad * np.sin(u) * np.cos(v), rad * np.cos(u), rad * np.sin(u) * np.sin(v)], -1).astype(np.float32)
    N = P / np.linalg.norm(P, axis=-1, keepdims=True)
    idx = np.arange((n_side + 1) * (n_side + 1)).reshape(n_side + 1, n_side + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    tri = np.concatenate([np.stack([a, c, b], 1), np.stack([a, d, c], 1)]).astype(np.int64)    # outward winding
    verts = np.zeros(tri.size, rr.VERTEX_DTYPE)
    verts["position"] = P.reshape(-1, 3)[tri.ravel()]
    verts["norm"] = N.reshape(-1, 3)[tri.ravel()].astype(np.float32)
    return verts, np.arange(tri.size, dtype=np.uint32)


def test_large_mesh_build_and_trace(gpu):
    """131 072 triangles: multi-block bitonic sort stages, Karras tree and the fenced bottom-up refit at a
    size where thousands of workgroups on all XCDs take part; structure checked vectorised, TraceRay
    against the oracle (its own median-split BVH, which was itself checked against brute force)."""
    verts, idx = procedural_mesh(256, seed=3)
    T = len(idx) // 3
    assert T == 131072
    mid = gpu.upload_mesh(verts, idx)
    gpu.build_blas(mid)
    gpu.build_tlas(rr.make_instances(meshes=[mid]))
    nodes, tris = gpu.download_blas(mid)
    assert len(nodes) == T - 1 and np.array_equal(np.sort(tris["prim"]), np.arange(T, dtype=np.uint32))
    c = nodes["c"].ravel()
    internal, leaves = c[c >= 0], ~c[c < 0]
    assert np.array_equal(np.sort(internal), np.arange(1, T - 1)) and np.array_equal(np.sort(leaves), np.arange(T))
    # boxes: every child box equals the union of its own children (leaf: the triangle's box), checked level-free
    P = verts["position"][idx].reshape(T, 3, 3)[tris["prim"]]
    leaf_lo, leaf_hi = P.min(1), P.max(1)
    lo = np.stack([nodes["lox"], nodes["loy"], nodes["loz"]], -1)      # [n, child, xyz]
    hi = np.stack([nodes["hix"], nodes["hiy"], nodes["hiz"]], -1)
    node_lo, node_hi = lo.min(1), hi.max(1)                             # box of each internal node = union of its two child boxes
    for k in (0, 1):
        ck = nodes["c"][:, k]
        is_leaf = ck < 0
        exp_lo = np.where(is_leaf[:, None], leaf_lo[np.where(is_leaf, ~ck, 0)], node_lo[np.where(is_leaf, 0, ck)])
        exp_hi = np.where(is_leaf[:, None], leaf_hi[np.where(is_leaf, ~ck, 0)], node_hi[np.where(is_leaf, 0, ck)])
        assert np.array_equal(lo[:, k], exp_lo) and np.array_equal(hi[:, k], exp_hi)
    s = O.Scene()
    s.add_mesh(verts, idx)
    rays = random_rays(3000, seed=99, radius=3.0)
    hits = gpu.trace_rays(rays)
    nh = 0
    for k in range(len(rays)):
        h = s.trace(rays["origin"][k], rays["dir"][k], float(rays["tmin"][k]), float(rays["tmax"][k]), int(rays["flags"][k]), use_bvh=1)
        assert bool(hits["hit"][k]) == bool(h.hit), k
        if h.hit:
            nh += 1
            assert hits["prim"][k] == h.prim and np.float32(hits["t"][k]).view(np.uint32) == np.float32(h.t).view(np.uint32)
    assert nh > 500
    # and a frame: deterministic across two renders, rays consistent
    s.set_envmap(procedural_env(64, 32, seed=1))
    gpu.upload_envmap(procedural_env(64, 32, seed=1))
    gpu.set_tile_partition(0, 1)
    gpu.set_camera(rr.camera_orbit(0.2))
    gpu.dispatch_rays(640, 360, rr.default_params(max_refract=8, flags=rr.DISPATCH_COLLECT_STATS))
    a = gpu.read_frame().copy()
    st = gpu.stats()
    assert st.traversal_overflow == 0 and st.hits + st.misses == st.rays
    gpu.dispatch_rays(640, 360, rr.default_params(max_refract=8))
    assert np.array_equal(gpu.read_frame(), a)
    sc = rr.camera_orbit(0.2)
    ref = s.render(np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32), 640, 360,
                   O.default_params(use_bvh=1, max_refract=8, accum_mode=1), region=(280, 150, 360, 210))
    assert np.array_equal(a[150:210, 280:360], ref["rgba8"][150:210, 280:360])


@pytest.mark.parametrize("levels,fast_build,depth", [(2, True, 26), (2, False, 24), (3, False, 30)])
def test_stack_size_boundaries(gpu, levels, fast_build, depth):
    """The kernels carry no stack-overflow check: the host picks the 26 / 31 / 39 / 64-entry instantiation from the
    built tree's depth.  Trees whose depth sits exactly on a boundary (26: the radix tree of the 15 472-triangle
    monkey fills the 26-entry stack to the last entry) must still trace like brute force."""
    from refraction_raytracing_dxr_amd.synth import subdivide
    m = load("monkey.obj")
    v, i = subdivide(m.verts, levels)
    mid = gpu.upload_mesh(v, i)
    gpu.build_blas(mid, fast_build=fast_build)
    gpu.build_tlas(rr.make_instances(meshes=[mid]))
    gpu.upload_envmap(procedural_env(32, 16))
    gpu.set_camera(rr.camera_orbit(0.4))
    gpu.dispatch_rays(64, 36, rr.default_params(flags=rr.DISPATCH_COLLECT_STATS))
    assert gpu.stats().bvh_depth == depth
    s = O.Scene()
    s.add_mesh(v, i)
    rays = random_rays(1200, seed=levels * 7 + depth)
    hits = gpu.trace_rays(rays)
    for k in range(len(rays)):
        h = s.trace(rays["origin"][k], rays["dir"][k], float(rays["tmin"][k]), float(rays["tmax"][k]), int(rays["flags"][k]), use_bvh=1)
        assert bool(hits["hit"][k]) == bool(h.hit), k
        if h.hit:
            assert hits["prim"][k] == h.prim and np.float32(hits["t"][k]).view(np.uint32) == np.float32(h.t).view(np.uint32)
    # and a frame through the render kernel of that stack size: deterministic, every ray ends in a hit or a miss
    gpu.set_tile_partition(0, 1)
    gpu.dispatch_rays(320, 180, rr.default_params(max_refract=8, flags=rr.DISPATCH_COLLECT_STATS))
    a = gpu.read_frame().copy()
    st = gpu.stats()
    assert st.hits + st.misses == st.rays and st.traversal_overflow == 0
    gpu.dispatch_rays(320, 180, rr.default_params(max_refract=8))
    assert np.array_equal(gpu.read_frame(), a)


def test_subdivided_monkey_16k_frame_parity(gpu):
    """BASELINE's '~16k tri Suzanne' = monkey.obj midpoint-subdivided twice (15 472 tri): same surface,
    16x the hierarchy.  Float accumulator bit-equal to the oracle's path-weight mode; and since midpoint
    subdivision moves no surface point, the image stays close to the 967-triangle one."""
    from refraction_raytracing_dxr_amd.synth import subdivide
    m = load("monkey.obj")
    v16, i16 = subdivide(m.verts, 2)
    assert len(i16) == 3 * 15472
    env = procedural_env(128, 64, seed=5)
    W, H = 256, 144
    sc = rr.camera_orbit(0.4)
    gpu.load_scene(v16, i16, env)
    gpu.set_tile_partition(0, 1)
    gpu.set_camera(sc)
    gpu.dispatch_rays(W, H, rr.default_params(max_refract=8, flags=rr.DISPATCH_FLOAT_OUTPUT | rr.DISPATCH_COLLECT_STATS))
    rgba, acc = gpu.read_frame(want_float=True)
    st = gpu.stats()
    assert st.traversal_overflow == 0
    s = O.Scene()
    s.add_mesh(v16, i16)
    s.set_envmap(env)
    ref = s.render(np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32), W, H,
                   O.default_params(use_bvh=1, max_refract=8, accum_mode=1))
    assert ref["stats"].rays == st.rays and (ref["stats"].hits, ref["stats"].misses) == (st.hits, st.misses)
    assert np.array_equal(acc[..., :3].view(np.uint32), ref["rgb"].view(np.uint32))
    assert np.array_equal(rgba, ref["rgba8"])
    # the coarse mesh renders nearly the same picture (shading normals differ slightly: normalised midpoints)
    gpu.load_scene(m.verts, m.indices, env)
    gpu.set_camera(sc)
    gpu.dispatch_rays(W, H, rr.default_params(max_refract=8))
    coarse = gpu.read_frame().astype(int)
    assert np.mean(np.abs(coarse - rgba.astype(int)) > 8) < 0.15
