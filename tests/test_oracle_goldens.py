"""Pins the CPU oracle against every golden the survey recorded for the reference (SURVEY.md
Appendix A.1, B, C) and against closed-form scenes.  CPU only."""
import os

import numpy as np
import pytest

import oracle as O

# produced by the reference's own Mesh::load body (g++ 11.4)
LOADER_GOLD = {
    "cube.obj": (12, "62afee94b8d6cb6a", (-1, -1, -1), (1, 1, 1)),
    "sphere.obj": (768, "9b7d5bd5769fd643", (-1.732051,) * 3, (1.732051,) * 3),
    "monkey.obj": (967, "a4734543877c2dd5", (-1.367188, -0.984375, -1.504792), (1.367188, 0.984375, 0.198333)),
    "shell.obj": (1536, "7f2f52b6a1a28e63", (-1.732051,) * 3, (1.732051,) * 3),
    "ott.obj": (12877, "46b040642a0ffe6f", (-0.927691, -1.211907, -1.236633), (0.931792, 1.282290, 0.559067)),
}

# SURVEY Appendix A.1 (float64 evaluation, compare at 1e-5), angle 0.01
CAMERA_KATS = [
    (1024, 768, 512, 384, (-0.999943191, -0.000640168, -0.010639805)),
    (1024, 768, 0, 0, (-0.778907078, 0.379981610, 0.498916566)),
    (1024, 768, 1023, 767, (-0.768773635, -0.379981610, -0.514393889)),
    (1920, 1080, 960, 540, (-0.999946425, -0.000455231, -0.010341152)),
    (1920, 1080, 0, 0, (-0.778775816, 0.380059543, 0.499062092)),
    (1920, 1080, 1919, 1079, (-0.768639490, -0.380059543, -0.514536761)),
]
M_GOLD = np.array([[0.006501146, 0, 0, -0.9919504], [0, 0.487716015, 0, 0],
                   [-0.650092915, 0, 0, -0.009919835], [-2.600371662, 0, 1, 1.059519008]])

# SURVEY Appendix C, 256x192, angle 0.01, envmap.png, max_reflect 2: rays per depth level
RAYCOUNT_GOLD = [
    ("sphere.obj", 1, [49152, 32232], 3),
    ("sphere.obj", 4, [49152, 32232, 32232, 16116], 6),
    ("cube.obj", 5, [49152, 19012, 14476, 6892], 6),
    ("monkey.obj", 5, [49152, 7110, 6942, 3096, 584, 234], 17),
    ("monkey.obj", 8, [49152, 7110, 6942, 3096, 584, 234, 129, 8, 1], 19),
]


@pytest.mark.parametrize("name", sorted(LOADER_GOLD))
def test_loader_matches_reference_hashes(name):
    tris, h, lo, hi = LOADER_GOLD[name]
    verts, idx = O.mesh_load(O.asset(name))
    assert len(verts) == 3 * tris and len(idx) == 3 * tris
    assert "%016x" % O.fnv1a64(verts) == h
    assert np.array_equal(idx, np.arange(3 * tris, dtype=np.uint32))      # Mesh.cpp:31
    assert np.allclose(verts["position"].min(0), lo, atol=1e-6)
    assert np.allclose(verts["position"].max(0), hi, atol=1e-6)


def test_loader_missing_file_returns_false():
    assert O.mesh_load("/nonexistent/file.obj") is None


def test_camera_constants():
    M, cam = O.camera(0.01)
    assert np.allclose(M.reshape(4, 4), M_GOLD, atol=2e-6)
    assert np.allclose(cam, (4.999750002, 0, 0.049999167, 1), atol=1e-6)


@pytest.mark.parametrize("W,H,x,y,d", CAMERA_KATS)
def test_camera_rays(W, H, x, y, d):
    M, cam = O.camera(0.01)
    o, dd = O.camera_ray(M, cam, x, y, W, H)
    assert np.allclose(dd, d, atol=1e-5)
    assert np.allclose(o, cam[:3])


def _scene(name, env):
    v, i = O.mesh_load(O.asset(name))
    s = O.Scene()
    s.add_mesh(v, i)
    s.set_envmap(env)
    return s


@pytest.mark.parametrize("name,limit,levels,maxrays", RAYCOUNT_GOLD)
def test_ray_counts_match_survey(name, limit, levels, maxrays, env_png):
    s = _scene(name, env_png)
    M, cam = O.camera(0.01)
    r = s.render(M, cam, 256, 192, O.default_params(max_refract=limit, use_bvh=1))
    st = r["stats"]
    got = [int(v) for v in st.rays_per_level if v]
    assert got == levels
    assert st.max_rays_per_pixel == maxrays
    assert st.rays == sum(levels) and st.hits + st.misses == st.rays


def test_shell_ray_counts_within_boundary_noise(env_png):
    # Appendix C was produced in float64; the fp32 oracle differs by one silhouette ray on shell
    s = _scene("shell.obj", env_png)
    M, cam = O.camera(0.01)
    st = s.render(M, cam, 256, 192, O.default_params(use_bvh=1))["stats"]
    gold = [49152, 32232, 29864, 29844, 13742]
    got = [int(v) for v in st.rays_per_level if v]
    assert len(got) == len(gold) and all(abs(a - b) <= 2 for a, b in zip(got, gold))


def test_terminal_hits_a4(env_png):
    # limit 1 on sphere -> every refracted child that hits again is terminal (16116 of 32232)
    s = _scene("sphere.obj", env_png)
    M, cam = O.camera(0.01)
    st = s.render(M, cam, 256, 192, O.default_params(max_refract=1, use_bvh=1))["stats"]
    assert st.terminal_hits == 16116
    s = _scene("monkey.obj", env_png)
    st = s.render(M, cam, 256, 192, O.default_params(use_bvh=1))["stats"]
    assert st.terminal_hits == 138


@pytest.mark.parametrize("name", ["cube.obj", "monkey.obj", "shell.obj"])
def test_bvh_equals_brute_force(name, env_png):
    s = _scene(name, env_png)
    M, cam = O.camera(0.37)
    a = s.render(M, cam, 96, 72, O.default_params(use_bvh=0), want_rays=True)
    b = s.render(M, cam, 96, 72, O.default_params(use_bvh=1), want_rays=True)
    assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
    assert np.array_equal(a["rays"], b["rays"])


def test_cube_closed_form_hits():
    v, i = O.mesh_load(O.asset("cube.obj"))
    s = O.Scene()
    s.add_mesh(v, i)
    # straight down the -x axis from the camera side: enters the x=+1 face at t=4
    h = s.trace((5, 0.25, 0.125), (-1, 0, 0), 1e-4, 100.0, O.CULL_BACK)
    assert h.hit and abs(h.t - 4.0) < 1e-6
    assert np.allclose(v["position"][3 * h.prim: 3 * h.prim + 3, 0], 1.0)
    # the same ray with front faces culled sees the inside of the far face at t=6
    h = s.trace((5, 0.25, 0.125), (-1, 0, 0), 1e-4, 100.0, O.CULL_FRONT)
    assert h.hit and abs(h.t - 6.0) < 1e-6
    # tmax is exclusive, tmin is exclusive
    assert not s.trace((5, 0.25, 0.125), (-1, 0, 0), 1e-4, 4.0, O.CULL_BACK).hit
    assert s.trace((5, 0.25, 0.125), (-1, 0, 0), 1e-4, 4.0001, O.CULL_BACK).hit
    assert not s.trace((5, 0.25, 0.125), (-1, 0, 0), 4.0, 5.0, O.CULL_BACK).hit
    # away from the cube: miss
    assert not s.trace((5, 0.25, 0.125), (1, 0, 0), 1e-4, 100.0, O.CULL_BACK).hit
    # barycentrics: u weights vertex 1, v weights vertex 2
    h = s.trace((5, 0.3, -0.2), (-1, 0, 0), 1e-4, 100.0, O.CULL_BACK)
    p = v["position"][3 * h.prim: 3 * h.prim + 3]
    hit_pt = p[0] + h.u * (p[1] - p[0]) + h.v * (p[2] - p[0])
    assert np.allclose(hit_pt, (1.0, 0.3, -0.2), atol=1e-6)


def test_glass_slab_closed_form(env_png):
    """A ray through the cube along a face normal is undeviated: the pixel is
    (1-R)^2 * env(d) + reflection terms, with R = R0(1-R0)*2^5 at normal incidence."""
    v, i = O.mesh_load(O.asset("cube.obj"))
    s = O.Scene()
    s.add_mesh(v, i)
    env = np.zeros((4, 8, 3), np.float32)
    env[:] = (0.5, 0.25, 0.125)                    # constant env: colour = 0.5 * sum of leaf weights
    s.set_envmap(env)
    # camera on the +x axis looking down -x through the cube centre region
    M = np.zeros(16, np.float32)
    M[3] = -1.0                                    # R = (-1, 0, 0) for every pixel
    cam = np.array([5, 0.3, -0.2, 1], np.float32)
    r = s.render(M, cam, 1, 1, O.default_params(), threads=1)
    R0 = np.float32(0.2 / 2.2) ** 2
    R = R0 * (1 - R0) * 32.0                       # pow(1 - dot(D,N), 5) with dot = -1
    # leaves: refract-refract (1-R)^2 ; refract-reflect... all leaves see the same env colour;
    # count<2 reflects: level0 reflect (R), level1 inside reflect (1-R)R then its children etc.
    got = r["rgb"][0, 0] / np.array([0.5, 0.25, 0.125], np.float32)
    assert np.allclose(got, got[0], rtol=1e-6)
    st = r["stats"]
    # tree: primary hit -> refract child (inside) + reflect child (outside, misses)
    #   inside child (count 1) hits the far face -> refract out (miss) + reflect (count 2, inside)
    #     inside reflect (count 2) hits the near face from inside -> refract out (miss) only
    assert st.rays == 1 + 2 + 2 + 1 and st.misses == 3 and st.hits == 3
    expect = (1 - R) * ((1 - R) * 1.0 + R * ((1 - R) * 1.0)) + R * 1.0
    assert abs(got[0] - expect) < 1e-6


def test_env_lookup_edges():
    s = O.Scene()
    v, i = O.mesh_load(O.asset("cube.obj"))
    s.add_mesh(v, i)
    env = np.arange(8 * 4 * 3, dtype=np.float32).reshape(4, 8, 3)
    s.set_envmap(env)
    # straight up: acos(1) = 0 -> row 0; atan2(0, 1) = 0 -> theta = w/2
    assert np.array_equal(s.env_lookup((0, 1, 0)), env[0, 4])
    # -z direction: atan2(0,-1) = pi -> theta = w*(pi/3.14159+1)/2 >= w -> out of range -> 0
    assert np.array_equal(s.env_lookup((0, 0, -1)), (0, 0, 0))
    # +x: atan2(1,0) = pi/2 -> theta = 0.75 w ; y = 0 -> phi = h/2
    assert np.array_equal(s.env_lookup((1, 0, 0)), env[2, 6])
    # straight down: acos(-1)/3.14159 > 1 -> phi >= h -> 0
    assert np.array_equal(s.env_lookup((0, -1, 0)), (0, 0, 0))


def test_unorm8_store():
    L = O.lib()
    assert L.rro_unorm8(float("nan")) == 0 and L.rro_unorm8(-1.0) == 0 and L.rro_unorm8(0.0) == 0
    assert L.rro_unorm8(1.0) == 255 and L.rro_unorm8(7.5) == 255
    assert L.rro_unorm8(0.5) == 128 and L.rro_unorm8(1.0 / 255) == 1 and L.rro_unorm8(0.49 / 255) == 0


def test_specified_transcendentals_close_to_libm():
    """rro_atan2f / rro_acosf are the arithmetic both sides use for texel selection; they must be
    within a few ulp of libm so the 'spec' is not hiding an error."""
    L = O.lib()
    rng = np.random.default_rng(1)
    d = rng.normal(size=(20000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    a = np.array([L.rro_atan2f(float(x), float(z)) for x, _, z in d], np.float32)
    assert np.max(np.abs(a - np.arctan2(d[:, 0].astype(np.float64), d[:, 2].astype(np.float64)))) < 6e-7
    c = np.array([L.rro_acosf(float(y)) for _, y, _ in d], np.float32)
    assert np.max(np.abs(c - np.arccos(d[:, 1].astype(np.float64)))) < 6e-7
    xs = np.linspace(-1, 1, 4001).astype(np.float32)
    c = np.array([L.rro_acosf(float(x)) for x in xs], np.float32)
    assert np.max(np.abs(c - np.arccos(xs.astype(np.float64)))) < 6e-7
    assert np.isnan(L.rro_acosf(1.0000001)) and np.isnan(L.rro_acosf(-1.5))


def test_libm_and_spec_images_agree(env_png):
    s = _scene("monkey.obj", env_png)
    M, cam = O.camera(0.01)
    a = s.render(M, cam, 128, 96, O.default_params(use_bvh=1))
    b = s.render(M, cam, 128, 96, O.default_params(use_bvh=1, use_libm=1))
    diff = np.abs(a["rgb"] - b["rgb"]).max(axis=2)
    assert (diff > 1e-6).mean() < 2e-3          # only texel-boundary flips


def test_path_weight_mode_is_within_rounding_of_recursion(env_hdr):
    s = _scene("shell.obj", env_hdr)
    M, cam = O.camera(0.01)
    a = s.render(M, cam, 128, 96, O.default_params(use_bvh=1, max_refract=8))
    b = s.render(M, cam, 128, 96, O.default_params(use_bvh=1, max_refract=8, accum_mode=1))
    assert np.allclose(a["rgb"], b["rgb"], rtol=2e-6, atol=1e-6)
    assert np.abs(a["rgba8"].astype(int) - b["rgba8"].astype(int)).max() <= 1


@pytest.mark.parametrize("name", ["sphere.obj", "shell.obj", "cube.obj"])
def test_bvh_equals_brute_force_on_the_symmetry_plane(name, env_png):
    """odd frame height: the middle row's rays run exactly in the y = 0 mirror plane of these meshes and
    hit shared edges; the padded box test must not cull what the triangle test accepts"""
    s = _scene(name, env_png)
    M, cam = O.camera(0.01)
    a = s.render(M, cam, 120, 67, O.default_params(use_bvh=0, max_refract=8), want_rays=True)
    b = s.render(M, cam, 120, 67, O.default_params(use_bvh=1, max_refract=8), want_rays=True)
    assert np.array_equal(a["rays"], b["rays"])
    assert np.array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))


def test_committed_golden_frames_are_reproduced():
    """tests/golden/frames.npz (made by tests/golden/make_goldens.py) pins the oracle against drift"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_goldens", os.path.join(os.path.dirname(__file__), "golden", "make_goldens.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "frames.npz"))
    for name, w, h, kw in mg.CASES:
        rgba, rgb, cnt = mg.render(name, w, h, kw)
        key = name.split(".")[0]
        assert np.array_equal(rgba, gold[key + "_rgba8"]), name
        assert np.array_equal(rgb.view(np.uint32), gold[key + "_rgb"].view(np.uint32)), name
        assert np.array_equal(cnt, gold[key + "_counts"]), name


def test_oracle_tonemap_option_is_c_over_one_plus_c():
    """the tone-map option of the checker itself (the product's RR_DISPATCH_TONEMAP_REINHARD is tested against
    it on the GPU): the float image is untouched, RGBA8 = unorm8(c / (1 + c))."""
    import numpy as np
    m_v, m_i = O.mesh_load(O.asset("cube.obj"))
    s = O.Scene()
    s.add_mesh(m_v, m_i)
    rng = np.random.default_rng(3)
    env = (rng.random((16, 32, 3)).astype(np.float32) * np.float32(12.0))
    s.set_envmap(env)
    M, cam = O.camera(0.3)
    a = s.render(M, cam, 48, 32, O.default_params(use_bvh=1))
    b = s.render(M, cam, 48, 32, O.default_params(use_bvh=1, tonemap=1))
    assert np.array_equal(a["rgb"], b["rgb"])
    x = a["rgb"].astype(np.float32)
    t = np.where(x > 0, x / (np.float32(1.0) + x), np.float32(0.0)).astype(np.float32)
    want = np.where(~(t > 0), 0, np.where(t >= 1, 255, np.floor(t * np.float32(255.0) + np.float32(0.5)))).astype(np.uint8)
    assert np.array_equal(b["rgba8"][..., :3], want) and not np.array_equal(a["rgba8"], b["rgba8"])
