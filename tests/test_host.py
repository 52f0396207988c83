"""Host side of the product (no GPU): OBJ loader, camera math, image decode, C-ABI surface."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle as O
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd import _capi
from test_oracle_goldens import CAMERA_KATS, LOADER_GOLD, M_GOLD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------- C ABI surface
def declared_functions():
    """every function prototype in include/rrdxr.h"""
    src = open(os.path.join(ROOT, "include", "rrdxr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rr_[a-z0-9_]+)\s*\(", src)) - {"rr_context"})


def test_library_exports_every_declared_symbol():
    lib = C.CDLL(rr.lib_path())
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "librrdxr.so lacks %s declared in include/rrdxr.h" % n
    # and the binding table covers the header exactly
    assert sorted(_capi.SYMBOLS) == names
    assert rr.lib().rr_abi_version() == 3


def test_header_compiles_as_c_and_structs_have_the_documented_sizes(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "rrdxr.h"\n'
                   '_Static_assert(sizeof(rr_vertex) == 32, "vertex");\n'
                   '_Static_assert(sizeof(rr_instance_desc) == 64, "instance");\n'
                   '_Static_assert(sizeof(rr_scene_constants) == 80, "constants");\n'
                   '_Static_assert(sizeof(rr_ray) == 48 && sizeof(rr_hit) == 24, "ray/hit");\n'
                   'int main(void) { return 0; }\n')
    import subprocess
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o",
                    str(tmp_path / "t.o")], check=True)


def test_no_device_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rr.RRError) as e:
        rr.Renderer(0)
    assert e.value.status == 2          # RR_ERR_NO_DEVICE
    h = C.c_void_p()
    assert rr.lib().rr_create(0, C.byref(h)) == 2 and not h.value


def test_default_params_are_the_shader_literals():
    p = rr.default_params()
    assert (p.max_refract, p.max_reflect) == (5, 2)                       # RayTracing.hlsl:82,110
    assert p.ior == np.float32(1.3)                                       # :95
    assert (p.tmin_primary, p.tmax_primary) == (np.float32(1e-4), 100.0)  # :52-53
    assert (p.tmin_secondary, p.tmax_secondary) == (np.float32(1e-3), 1000.0)   # :99-100
    assert p.flags == 0


# ---------------------------------------------------------------------------------- Mesh::load
@pytest.mark.parametrize("name", sorted(LOADER_GOLD))
def test_mesh_load_matches_reference_hashes(name):
    tris, h, lo, hi = LOADER_GOLD[name]
    m = rr.Mesh()
    assert m.load(O.asset(name)) is True
    assert len(m.verts) == 3 * tris and np.array_equal(m.indices, np.arange(3 * tris, dtype=np.uint32))
    assert "%016x" % O.fnv1a64(m.verts) == h
    ov, oi = O.mesh_load(O.asset(name))
    assert m.verts.tobytes() == ov.tobytes() and np.array_equal(m.indices, oi)


def test_mesh_load_edge_cases(tmp_path):
    m = rr.Mesh()
    assert m.load(str(tmp_path / "missing.obj")) is False                 # Mesh.cpp:9-10
    p = tmp_path / "e.obj"
    p.write_text("")                                                      # empty file: loads, no triangles
    assert m.load(str(p)) is True and len(m.verts) == 0
    # sscanf semantics: leading blanks disqualify a line, comments and unknown records are ignored,
    # a 4th corner is dropped, 'v' wins over 'vt'/'vn' only if three floats follow
    p.write_text("# c\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\n v 9 9 9\nvt 0 0\nvt 1 1\nvn 0 0 1\n"
                 "l 1 2\nf 1/1/1 2/2/1 3/1/1 4/2/1\nf 1//1 2//1 3//1\nf 4/2/1 3/1/1 2/2/1\n")
    m = rr.Mesh()
    assert m.load(str(p)) and len(m.verts) == 6
    assert np.array_equal(m.verts["position"][3], (0, 0, 1)) and np.array_equal(m.verts["uv"][1], (1, 1))
    ov, _ = O.mesh_load(str(p))
    assert m.verts.tobytes() == ov.tobytes()
    # loading twice appends, indices keep counting (Mesh.cpp:31-32)
    assert m.load(str(p)) and len(m.verts) == 12 and m.indices[-1] == 11
    # out-of-range references: the reference reads wild memory; this loader refuses
    p.write_text("v 0 0 0\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 1/1/1\n")
    assert rr.Mesh().load(str(p)) is False


# ---------------------------------------------------------------------------------- camera
def test_position_validation(tmp_path):
    """what rr_upload_mesh demands of positions (finite, |coordinate| <= 1e18), checked on the CPU; Mesh::load itself keeps
    accepting such a file, as the reference's loader does (strtof("1e39") is +inf)"""
    L = rr.lib()
    bad = C.c_uint32(99)
    v = np.zeros(5, rr.VERTEX_DTYPE)
    v["position"] = np.arange(15, dtype=np.float32).reshape(5, 3)
    assert L.rr_host_validate_positions(v.ctypes.data, 5, C.byref(bad)) == 0 and bad.value == 99
    for val in (np.inf, -np.inf, np.nan, 1.1e18, -3e38):
        w = v.copy()
        w["position"][3, 1] = val
        assert L.rr_host_validate_positions(w.ctypes.data, 5, C.byref(bad)) == 1 and bad.value == 3, val
    w = v.copy()
    w["position"][0] = (1e18, -1e18, 0)
    assert L.rr_host_validate_positions(w.ctypes.data, 5, None) == 0
    w["norm"][2] = np.nan                                        # normals and uvs are not positions
    assert L.rr_host_validate_positions(w.ctypes.data, 5, None) == 0
    assert L.rr_host_validate_positions(None, 0, None) == 0
    p = tmp_path / "huge.obj"
    p.write_text("v 1e39 0 0\nv 0 1 0\nv 0 0 1\nvt 0 0\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1\n")
    m = rr.Mesh()
    assert m.load(p) and np.isinf(np.asarray(m.verts)["position"][0, 0])
    assert L.rr_host_validate_positions(np.asarray(m.verts).ctypes.data, 3, C.byref(bad)) == 1 and bad.value == 0


def test_camera_matches_oracle_and_kats():
    for a in (0.01, 0.02, 1.0, 3.14, 6.28, -0.5):
        sc = rr.camera_orbit(a)
        M, cam = O.camera(a)
        assert np.allclose(np.array(sc.proj_inv), M, rtol=0, atol=2e-6)
        assert np.allclose(np.array(sc.camera_loc), cam, rtol=0, atol=1e-6)
    sc = rr.camera_orbit(0.01)
    assert np.allclose(np.array(sc.proj_inv).reshape(4, 4), M_GOLD, atol=2e-6)
    for W, H, x, y, d in CAMERA_KATS:
        _, dd = O.camera_ray(np.array(sc.proj_inv), np.array(sc.camera_loc), x, y, W, H)
        assert np.allclose(dd, d, atol=1e-5)
    bad = _capi.SceneConstants()
    assert rr.lib().rr_host_camera_orbit(0.01, 0.0, 1.333, 1.0, 125.0, C.byref(bad)) == 1


# ---------------------------------------------------------------------------------- stbi_loadf stand-in
REF_STB = os.path.join(ROOT, "oracle", "_ref", "libstb_ref.so")


def have_ref_stb():
    """the reference's stb_image.h compiled in place (oracle/Makefile); only looked for here -- it is loaded by the two tests
    that check against it, when they run, never at collection time (a `-m gpu` run does not map it)"""
    return os.path.exists(REF_STB)


def ref_stb():
    L = C.CDLL(REF_STB)
    L.stbi_loadf.restype = C.POINTER(C.c_float)
    L.stbi_loadf.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    L.stbi_image_free.argtypes = [C.c_void_p]
    return L


def ref_loadf(path, req):
    L = ref_stb()
    x, y, n = C.c_int(), C.c_int(), C.c_int()
    p = L.stbi_loadf(str(path).encode(), C.byref(x), C.byref(y), C.byref(n), req)
    if not p:
        return None, 0
    oc = req if req else n.value
    a = np.ctypeslib.as_array(p, shape=(y.value, x.value, oc)).copy()
    L.stbi_image_free(p)
    return a, n.value


def test_envmap_png_golden():
    env, n = rr.load_texture(O.asset("envmap.png"), 3)
    assert env.shape == (480, 640, 3) and n == 4
    assert "%016x" % O.fnv1a64(env) == "38c5be075155201f"
    assert abs(env.mean() - 0.528603) < 1e-6 and env.max() == 1.0
    assert abs(env[0, 0, 0] - 0.344026) < 1e-6 and abs(env[240, 320, 0] - 0.529523) < 1e-6


def _write_png(path, arr, mode):
    from PIL import Image
    Image.fromarray(arr, mode).save(path)


@pytest.mark.skipif(not have_ref_stb(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("req", [0, 1, 2, 3, 4])
def test_png_decode_bit_exact_vs_reference_stb(tmp_path, req):
    rng = np.random.default_rng(req)
    cases = {
        "rgb.png": (rng.integers(0, 256, (37, 53, 3), dtype=np.uint8), "RGB"),
        "rgba.png": (rng.integers(0, 256, (19, 64, 4), dtype=np.uint8), "RGBA"),
        "gray.png": (rng.integers(0, 256, (40, 31), dtype=np.uint8), "L"),
        "la.png": (rng.integers(0, 256, (16, 16, 2), dtype=np.uint8), "LA"),
        "smooth.png": (np.tile(np.arange(256, dtype=np.uint8), (64, 1)), "L"),     # exercises the PNG filters
    }
    for name, (arr, mode) in cases.items():
        p = tmp_path / name
        _write_png(p, arr, mode)
        mine, n1 = rr.load_texture(p, req)
        ref, n2 = ref_loadf(p, req)
        assert n1 == n2 and mine.shape == ref.shape, name
        assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32)), name
    # palette and 16-bit
    from PIL import Image
    Image.fromarray(cases["rgb.png"][0], "RGB").quantize(17).save(tmp_path / "pal.png")
    Image.fromarray(rng.integers(0, 65536, (9, 11), dtype=np.uint16)).save(tmp_path / "g16.png")
    for name in ("pal.png", "g16.png"):
        mine, n1 = rr.load_texture(tmp_path / name, req)
        ref, n2 = ref_loadf(tmp_path / name, req)
        assert n1 == n2 and np.array_equal(mine.view(np.uint32), ref.view(np.uint32)), name
    mine, _ = rr.load_texture(O.asset("envmap.png"), req)
    ref, _ = ref_loadf(O.asset("envmap.png"), req)
    assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32))


def test_hdr_write_read_roundtrip(tmp_path):
    from conftest import procedural_env
    env = procedural_env(96, 48, seed=2)
    env[0, :9] = 0.0                                   # a run of black pixels (RLE runs, zero exponent)
    env[1, 0] = (1e-40, 0, 0)                          # underflows to the all-zero RGBE pixel
    p = tmp_path / "e.hdr"
    rr.write_hdr(p, env)
    back, n = rr.load_texture(p, 3)
    assert back.shape == env.shape and n == 3
    mx = env.max(axis=2, keepdims=True)
    assert np.all(np.abs(back - env) <= mx / 128.0 + 1e-30)     # 8-bit mantissa shared exponent
    assert np.all(back[0, :9] == 0) and np.all(back[1, 0] == 0)
    # narrow images are stored flat (width < 8): same pixel values
    rr.write_hdr(tmp_path / "n.hdr", env[:, :5].copy())
    nb, _ = rr.load_texture(tmp_path / "n.hdr", 3)
    assert np.array_equal(nb, back[:, :5])
    # idempotent once quantised
    rr.write_hdr(tmp_path / "e2.hdr", back)
    again, _ = rr.load_texture(tmp_path / "e2.hdr", 3)
    assert np.array_equal(again, back)


@pytest.mark.skipif(not have_ref_stb(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("req", [0, 1, 3, 4])
def test_hdr_decode_bit_exact_vs_reference_stb(tmp_path, req):
    from conftest import procedural_env
    for w, h in ((96, 48), (5, 7), (8, 3)):
        p = tmp_path / ("e%dx%d.hdr" % (w, h))
        rr.write_hdr(p, procedural_env(w, h, seed=w))
        mine, n1 = rr.load_texture(p, req)
        ref, n2 = ref_loadf(p, req)
        assert n1 == n2 == 3 and np.array_equal(mine.view(np.uint32), ref.view(np.uint32))


def _write_adam7_png(path, arr, depth=8, palette=None):
    """A PNG with interlace method 1 (Adam7) built by hand: arr is H x W (gray / palette indices, `depth` bits) or
    H x W x C with C = 2 (gray+alpha), 3 (RGB), 4 (RGBA) at 8 bits.  Rows use filter types 0, 1 and 2 in turn."""
    import struct, zlib
    h, w = arr.shape[:2]
    ch = 1 if arr.ndim == 2 else arr.shape[2]
    ctype = {1: 3 if palette is not None else 0, 2: 4, 3: 2, 4: 6}[ch]
    bits_pp = ch * depth
    bpp = max(1, bits_pp // 8)

    def pack_row(px):                       # px: n x ch samples -> bytes
        if depth == 8:
            return bytes(px.astype(np.uint8).reshape(-1))
        if depth == 16:
            return px.astype(">u2").reshape(-1).tobytes()
        bits = "".join(format(int(v), "0%db" % depth) for v in px.reshape(-1))
        bits += "0" * (-len(bits) % 8)
        return bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))

    raw = bytearray()
    for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = arr[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        prev = None
        for j in range(sub.shape[0]):
            cur = pack_row(sub[j].reshape(sub.shape[1], ch))
            ft = j % 3
            if ft == 1:
                out = bytes((cur[i] - (cur[i - bpp] if i >= bpp else 0)) & 255 for i in range(len(cur)))
            elif ft == 2:
                out = bytes((cur[i] - (prev[i] if prev is not None else 0)) & 255 for i in range(len(cur)))
            else:
                out = cur
            raw += bytes([ft]) + out
            prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1))
    if palette is not None:
        png += chunk(b"PLTE", bytes(np.asarray(palette, np.uint8).reshape(-1)))
    png += chunk(b"IDAT", zlib.compress(bytes(raw))) + chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(png)


def _adam7_cases(rng):
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    return {
        "i_rgb.png": (rng.integers(0, 256, (37, 53, 3), dtype=np.uint8), 8, None),
        "i_rgba.png": (rng.integers(0, 256, (9, 5, 4), dtype=np.uint8), 8, None),
        "i_la.png": (rng.integers(0, 256, (16, 17, 2), dtype=np.uint8), 8, None),
        "i_gray.png": (rng.integers(0, 256, (40, 31), dtype=np.uint8), 8, None),
        "i_g16.png": (rng.integers(0, 65536, (11, 9), dtype=np.uint16), 16, None),
        "i_g1.png": (rng.integers(0, 2, (13, 21), dtype=np.uint8), 1, None),
        "i_g2.png": (rng.integers(0, 4, (7, 3), dtype=np.uint8), 2, None),
        "i_pal4.png": (rng.integers(0, 16, (10, 19), dtype=np.uint8), 4, pal),
        "i_tiny.png": (rng.integers(0, 256, (1, 1, 3), dtype=np.uint8), 8, None),      # six of the seven passes are empty
        "i_thin.png": (rng.integers(0, 256, (3, 2, 3), dtype=np.uint8), 8, None),
    }


def test_adam7_png_decodes_to_the_image_it_was_made_from(tmp_path):
    """Interlaced PNGs (stb_image reads them, RefractionDemo.cpp:111 goes through stbi_loadf): the seven passes are put back
    where they belong -- checked against the pixels the file was built from, through the decoder's own LDR -> float rule."""
    rng = np.random.default_rng(7)
    for name, (arr, depth, pal) in _adam7_cases(rng).items():
        _write_adam7_png(tmp_path / name, arr, depth, pal)
        mine, n = rr.load_texture(tmp_path / name, 0)
        if pal is not None:
            want8 = np.asarray(pal)[arr]
        elif depth == 16:
            want8 = (arr >> 8).astype(np.uint8)
        elif depth < 8:
            want8 = (arr * {1: 255, 2: 85, 4: 17}[depth]).astype(np.uint8)
        else:
            want8 = arr
        want8 = want8.reshape(arr.shape[0], arr.shape[1], -1)
        assert mine.shape == want8.shape and n == want8.shape[2], name
        col = want8.shape[2] - (1 if want8.shape[2] in (2, 4) else 0)               # alpha stays linear
        want = np.empty(want8.shape, np.float32)
        want[..., :col] = np.power(want8[..., :col].astype(np.float32) / np.float32(255.0), np.float32(2.2)).astype(np.float32)
        if col < want8.shape[2]:
            want[..., col:] = want8[..., col:].astype(np.float32) / np.float32(255.0)
        assert np.allclose(mine, want, rtol=2e-6, atol=1e-7), name


@pytest.mark.skipif(not have_ref_stb(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("req", [0, 3])
def test_adam7_png_bit_exact_vs_reference_stb(tmp_path, req):
    rng = np.random.default_rng(8)
    for name, (arr, depth, pal) in _adam7_cases(rng).items():
        _write_adam7_png(tmp_path / name, arr, depth, pal)
        mine, n1 = rr.load_texture(tmp_path / name, req)
        ref, n2 = ref_loadf(tmp_path / name, req)
        assert n1 == n2 and mine.shape == ref.shape, name
        assert np.array_equal(mine.view(np.uint32), ref.view(np.uint32)), name


def test_formats_outside_the_asset_surface_are_refused(tmp_path):
    """rr_host_image_loadf decodes Radiance .hdr and PNG only (include/rrdxr.h); the other formats stbi_loadf accepts
    (stb_image.h:1491) come back as a clean failure, never as a crash or a wrong image."""
    from PIL import Image
    img = Image.fromarray(np.arange(48, dtype=np.uint8).reshape(4, 4, 3), "RGB")
    for name, fmt in (("a.jpg", "JPEG"), ("a.bmp", "BMP"), ("a.tga", "TGA"), ("a.gif", "GIF"), ("a.ppm", "PPM")):
        img.save(tmp_path / name, fmt)
        with pytest.raises(rr.RRError):
            rr.load_texture(tmp_path / name)
    (tmp_path / "interlace2.png").write_bytes(b"\x89PNG\r\n\x1a\n" + b"\0\0\0\x0dIHDR" + b"\0\0\0\x02\0\0\0\x02\x08\x02\0\0\x02" + b"\0" * 20)
    with pytest.raises(rr.RRError):
        rr.load_texture(tmp_path / "interlace2.png")      # interlace method 2 does not exist


def test_image_load_failures(tmp_path):
    with pytest.raises(rr.RRError):
        rr.load_texture(tmp_path / "nope.hdr")
    (tmp_path / "junk.png").write_bytes(b"\x89PNG\r\n\x1a\n" + b"\0" * 40)
    with pytest.raises(rr.RRError):
        rr.load_texture(tmp_path / "junk.png")
    (tmp_path / "bad.hdr").write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n-Y 2 +X 2\n" + b"\0" * 16)
    with pytest.raises(rr.RRError):
        rr.load_texture(tmp_path / "bad.hdr")


def test_hardened_obj_loader(tmp_path):
    """quads, v//vn and bare-v corners, negative indices -- additive, default unchanged."""
    for name in LOADER_GOLD:                                   # hardened mode reads the reference's files identically
        a, b = rr.Mesh(), rr.Mesh()
        assert a.load(O.asset(name)) and b.load(O.asset(name), hardened=True)
        assert a.verts.tobytes() == b.verts.tobytes() and np.array_equal(a.indices, b.indices)
    p = tmp_path / "q.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0.5 0.25\n"
                 "f 1//1 2//1 3//1 4//1\n"          # quad, no uv
                 "f -4 -3 -2\n"                      # negative indices, no uv, no normal
                 "f 1/1 2/1 3/1\n")                  # uv but no normal
    strict = rr.Mesh()
    assert strict.load(str(p)) and len(strict.verts) == 0      # the reference's sscanf pattern matches none of these
    m = rr.Mesh()
    assert m.load(str(p), hardened=True)
    assert len(m.verts) == 12 and np.array_equal(m.indices, np.arange(12, dtype=np.uint32))
    P = m.verts["position"]
    assert np.array_equal(P[:6], [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 0, 0), (1, 1, 0), (0, 1, 0)])   # fan
    assert np.array_equal(P[6:9], [(0, 0, 0), (1, 0, 0), (1, 1, 0)])
    assert np.all(m.verts["norm"][:6] == (0, 0, 1)) and np.all(m.verts["uv"][:6] == 0)
    assert np.allclose(m.verts["norm"][6:12], (0, 0, 1))       # flat winding normal where vn is absent
    assert np.all(m.verts["uv"][9:12] == (0.5, 0.25))
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 9\n")
    assert rr.Mesh().load(str(p), hardened=True) is False      # out of range
