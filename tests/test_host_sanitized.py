"""Host code under AddressSanitizer + UBSan (the reference has no sanitizer story; its loaders
read out of bounds on malformed files).  The OBJ loader and the PNG / Radiance decoders of the C ABI are
compiled with g++ -fsanitize=address,undefined and run on the committed assets, on truncations of them and on
seeded bit-flips: every input must end in a decoded asset or a clean refusal, never in a sanitizer report."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "refraction_raytracing_dxr_amd", "csrc", "host")
ASSETS = os.path.join(ROOT, "refraction_raytracing_dxr_amd", "assets")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not available")
    out = str(tmp_path_factory.mktemp("san") / "host_san_driver")
    srcs = [os.path.join(ROOT, "tests", "native", "host_san_driver.cpp")] + \
           [os.path.join(HOST, f) for f in ("rr_host_mesh.cpp", "rr_host_image.cpp", "rr_host_camera.cpp")]
    subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-fno-omit-frame-pointer"] + srcs + ["-o", out], check=True)
    return out


def run(driver, mode, files):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([driver, mode] + files, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    ok, refused = (int(t) for t in r.stdout.split()[1::2])
    return ok, refused


def mutations(path, out_dir, n_trunc, n_flip, seed):
    raw = np.fromfile(path, np.uint8)
    rng = np.random.default_rng(seed)
    files = []
    for k, cut in enumerate(sorted(set(int(c) for c in rng.integers(0, len(raw), n_trunc)) | {0, 1, 8, len(raw) - 1})):
        f = os.path.join(out_dir, "t%d_%s" % (k, os.path.basename(path)))
        raw[:cut].tofile(f)
        files.append(f)
    for k in range(n_flip):
        m = raw.copy()
        for pos in rng.integers(0, len(m), int(rng.integers(1, 6))):
            m[pos] ^= np.uint8(1 << int(rng.integers(0, 8)))
        f = os.path.join(out_dir, "f%d_%s" % (k, os.path.basename(path)))
        m.tofile(f)
        files.append(f)
    return files


def test_valid_assets_decode_cleanly(driver, tmp_path):
    import refraction_raytracing_dxr_amd as rr
    hdr = str(tmp_path / "env.hdr")
    env, _ = rr.load_texture(os.path.join(ASSETS, "envmap.png"), 3)
    rr.write_hdr(hdr, env[:40, :64].copy())
    ok, refused = run(driver, "img", [os.path.join(ASSETS, "envmap.png"), hdr])
    assert ok == 10 and refused == 0
    objs = [os.path.join(ASSETS, f) for f in ("cube.obj", "sphere.obj", "monkey.obj", "shell.obj")]
    assert run(driver, "obj", objs) == (4, 0)
    assert run(driver, "objx", objs) == (4, 0)


def test_malformed_images_never_trip_the_sanitizers(driver, tmp_path):
    import refraction_raytracing_dxr_amd as rr
    env, _ = rr.load_texture(os.path.join(ASSETS, "envmap.png"), 3)
    small_png_src = os.path.join(ASSETS, "envmap.png")
    hdr = str(tmp_path / "env.hdr")
    rr.write_hdr(hdr, env[:24, :40].copy())
    files = mutations(small_png_src, str(tmp_path), 40, 60, seed=1) + mutations(hdr, str(tmp_path), 40, 120, seed=2)
    ok, refused = run(driver, "img", files)
    assert ok + refused == 5 * len(files) and refused > 0


def test_malformed_objs_never_trip_the_sanitizers(driver, tmp_path):
    files = mutations(os.path.join(ASSETS, "cube.obj"), str(tmp_path), 60, 200, seed=3) + \
            mutations(os.path.join(ASSETS, "sphere.obj"), str(tmp_path), 20, 60, seed=4)
    for mode in ("obj", "objx"):
        ok, refused = run(driver, mode, files)
        assert ok + refused == len(files)


def test_interlaced_pngs_never_trip_the_sanitizers(driver, tmp_path):
    """The Adam7 de-interlace path parses untrusted files too: interlaced PNGs of every sample depth (1, 2, 4, 8, 16 bits;
    gray, gray+alpha, RGB, RGBA, palette), 1 x N and N x 1 images (most of the seven passes empty), truncations at every
    kind of offset and seeded bit-flips of all of them, through the ASan + UBSan build: decoded or refused, never a report."""
    from test_host import _adam7_cases, _write_adam7_png
    rng = np.random.default_rng(12)
    cases = dict(_adam7_cases(rng))
    cases.update({
        "i_1xN.png": (rng.integers(0, 256, (23, 1, 3), dtype=np.uint8), 8, None),
        "i_Nx1.png": (rng.integers(0, 256, (1, 29, 3), dtype=np.uint8), 8, None),
        "i_g1_1xN.png": (rng.integers(0, 2, (17, 1), dtype=np.uint8), 1, None),
        "i_g2_Nx1.png": (rng.integers(0, 4, (1, 13), dtype=np.uint8), 2, None),
        "i_g4.png": (rng.integers(0, 16, (9, 10), dtype=np.uint8), 4, None),
    })
    valid, files = [], []
    for k, (name, (arr, depth, pal)) in enumerate(sorted(cases.items())):
        path = str(tmp_path / name)
        _write_adam7_png(path, arr, depth, pal)
        valid.append(path)
        files += mutations(path, str(tmp_path), 12, 25, seed=100 + k)
    ok, refused = run(driver, "img", valid)
    assert ok == 5 * len(valid) and refused == 0
    ok, refused = run(driver, "img", files)
    assert ok + refused == 5 * len(files) and refused > 0
