"""ctypes view of the CPU oracle (oracle/librr_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The shipped package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ASSETS = os.path.join(ROOT, "refraction_raytracing_dxr_amd", "assets")      # the reference's data files, shipped as package data

CULL_BACK = 0x10
CULL_FRONT = 0x20


class Vertex(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("norm", C.c_float * 3), ("uv", C.c_float * 2)]


class Instance(C.Structure):
    _fields_ = [("transform", C.c_float * 12), ("id_mask", C.c_uint32),
                ("hitgroup_flags", C.c_uint32), ("blas", C.c_uint64)]


class Params(C.Structure):
    _fields_ = [("max_refract", C.c_int), ("max_reflect", C.c_int), ("ior", C.c_float),
                ("tmin_primary", C.c_float), ("tmax_primary", C.c_float),
                ("tmin_secondary", C.c_float), ("tmax_secondary", C.c_float),
                ("use_libm", C.c_int), ("accum_mode", C.c_int), ("use_bvh", C.c_int), ("tonemap", C.c_int)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("primary", C.c_uint64), ("secondary", C.c_uint64),
                ("hits", C.c_uint64), ("misses", C.c_uint64), ("terminal_hits", C.c_uint64),
                ("tir", C.c_uint64), ("max_rays_per_pixel", C.c_uint64),
                ("rays_per_level", C.c_uint64 * 32), ("tri_tests", C.c_uint64),
                ("node_visits", C.c_uint64)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("u", C.c_float), ("v", C.c_float),
                ("prim", C.c_uint32), ("inst", C.c_uint32), ("hit", C.c_int)]


VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("norm", "<f4", 3), ("uv", "<f4", 2)])
assert VERTEX_DTYPE.itemsize == 32

_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.path.join(ORACLE_DIR, "librr_oracle.so")
    src = os.path.join(ORACLE_DIR, "rr_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        build()
    L = C.CDLL(so)
    L.rro_default_params.argtypes = [C.POINTER(Params)]
    L.rro_fnv1a64.restype = C.c_uint64
    L.rro_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
    L.rro_mesh_load.restype = C.c_int
    L.rro_mesh_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(Vertex)), C.POINTER(C.c_uint32),
                                C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32)]
    L.rro_free.argtypes = [C.c_void_p]
    L.rro_camera.argtypes = [C.c_float] * 5 + [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.rro_generate_camera_ray.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)] + [C.c_uint32] * 4 + \
        [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.rro_scene_create.restype = C.c_void_p
    L.rro_scene_destroy.argtypes = [C.c_void_p]
    L.rro_scene_add_mesh.restype = C.c_int
    L.rro_scene_add_mesh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.rro_scene_set_instances.restype = C.c_int
    L.rro_scene_set_instances.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.rro_scene_set_envmap.restype = C.c_int
    L.rro_scene_set_envmap.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.rro_trace.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float,
                            C.c_uint32, C.c_int, C.POINTER(Hit)]
    L.rro_env_lookup.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]
    L.rro_render.restype = C.c_int
    L.rro_render.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_uint32,
                             C.POINTER(Params)] + [C.c_uint32] * 8 + [C.c_int, C.c_void_p, C.c_void_p,
                                                                      C.c_void_p, C.POINTER(Stats)]
    L.rro_atan2f.restype = C.c_float
    L.rro_atan2f.argtypes = [C.c_float, C.c_float]
    L.rro_acosf.restype = C.c_float
    L.rro_acosf.argtypes = [C.c_float]
    L.rro_unorm8.restype = C.c_uint8
    L.rro_unorm8.argtypes = [C.c_float]
    _lib = L
    return L


def default_params(**kw):
    p = Params()
    lib().rro_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def fnv1a64(arr):
    a = np.ascontiguousarray(arr)
    return int(lib().rro_fnv1a64(a.ctypes.data, a.nbytes))


def mesh_load(path):
    """Mesh::load restatement -> (verts structured array, indices uint32) or None."""
    v = C.POINTER(Vertex)()
    i = C.POINTER(C.c_uint32)()
    nv, ni = C.c_uint32(), C.c_uint32()
    ok = lib().rro_mesh_load(path.encode(), C.byref(v), C.byref(nv), C.byref(i), C.byref(ni))
    if not ok:
        return None
    verts = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_uint8)), shape=(max(nv.value, 1) * 32,))[:nv.value * 32]
    verts = verts.copy().view(VERTEX_DTYPE)
    idx = np.ctypeslib.as_array(i, shape=(max(ni.value, 1),))[:ni.value].copy()
    lib().rro_free(v)
    lib().rro_free(i)
    return verts, idx


def camera(angle, fov_y=None, aspect=None, zn=1.0, zf=125.0):
    """RefractionDemo.cpp:559-566 -> (proj_inv[16] float32, camera_loc[4] float32)."""
    if fov_y is None:
        fov_y = np.float32(52.0 / 180.0 * 3.1415)
    if aspect is None:
        aspect = np.float32(1.333)
    m = (C.c_float * 16)()
    c = (C.c_float * 4)()
    lib().rro_camera(float(np.float32(angle)), float(fov_y), float(aspect), zn, zf, m, c)
    return np.array(m, dtype=np.float32), np.array(c, dtype=np.float32)


def camera_ray(proj_inv, cam, x, y, w, h):
    o = (C.c_float * 3)()
    d = (C.c_float * 3)()
    M = (C.c_float * 16)(*[float(v) for v in proj_inv])
    cc = (C.c_float * 4)(*[float(v) for v in cam])
    lib().rro_generate_camera_ray(M, cc, x, y, w, h, o, d)
    return np.array(o, dtype=np.float32), np.array(d, dtype=np.float32)


def make_instance(transform3x4=None, mesh=0, mask=1, flags=0, instance_id=0):
    inst = np.zeros(1, dtype=INSTANCE_DTYPE)
    t = np.eye(4, dtype=np.float32)[:3] if transform3x4 is None else np.asarray(transform3x4, np.float32)
    inst["transform"][0] = t.reshape(12)
    inst["id_mask"][0] = (instance_id & 0xffffff) | ((mask & 0xff) << 24)
    inst["hitgroup_flags"][0] = (flags & 0xff) << 24
    inst["blas"][0] = mesh
    return inst


INSTANCE_DTYPE = np.dtype([("transform", "<f4", 12), ("id_mask", "<u4"), ("hitgroup_flags", "<u4"), ("blas", "<u8")])
assert INSTANCE_DTYPE.itemsize == 64


class Scene:
    def __init__(self):
        self.h = lib().rro_scene_create()

    def __del__(self):
        if getattr(self, "h", None):
            lib().rro_scene_destroy(self.h)
            self.h = None

    def add_mesh(self, verts, idx):
        verts = np.ascontiguousarray(verts)
        idx = np.ascontiguousarray(idx, dtype=np.uint32)
        r = lib().rro_scene_add_mesh(self.h, verts.ctypes.data, len(verts), idx.ctypes.data, len(idx))
        if r < 0:
            raise ValueError("rro_scene_add_mesh failed")
        return r

    def set_instances(self, inst):
        inst = np.ascontiguousarray(inst, dtype=INSTANCE_DTYPE)
        if lib().rro_scene_set_instances(self.h, inst.ctypes.data, len(inst)) != 0:
            raise ValueError("rro_scene_set_instances failed")

    def set_envmap(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        h, w, c = rgb.shape
        assert c == 3
        if lib().rro_scene_set_envmap(self.h, rgb.ctypes.data, w, h) != 0:
            raise ValueError("rro_scene_set_envmap failed")

    def trace(self, o, d, tmin, tmax, flags, use_bvh=0):
        hh = Hit()
        oo = (C.c_float * 3)(*[float(v) for v in o])
        dd = (C.c_float * 3)(*[float(v) for v in d])
        lib().rro_trace(self.h, oo, dd, tmin, tmax, flags, use_bvh, C.byref(hh))
        return hh

    def env_lookup(self, d, use_libm=0):
        dd = (C.c_float * 3)(*[float(v) for v in d])
        out = (C.c_float * 3)()
        lib().rro_env_lookup(self.h, dd, use_libm, out)
        return np.array(out, dtype=np.float32)

    def render(self, proj_inv, cam, w, h, params=None, region=None, tile=(32, 32), rank=0, world=1,
               threads=None, want_rays=False):
        """-> dict(rgb float32 [h,w,3], rgba8 uint8 [h,w,4], stats Stats, rays uint16 [h,w])"""
        p = params if params is not None else default_params()
        M = (C.c_float * 16)(*[float(v) for v in proj_inv])
        cc = (C.c_float * 4)(*[float(v) for v in cam])
        x0, y0, x1, y1 = region if region else (0, 0, w, h)
        rgb = np.zeros((h, w, 3), np.float32)
        rgba = np.zeros((h, w, 4), np.uint8)
        rays = np.zeros((h, w), np.uint16) if want_rays else None
        st = Stats()
        if threads is None:
            threads = os.cpu_count() or 1
        r = lib().rro_render(self.h, M, cc, w, h, C.byref(p), x0, y0, x1, y1, tile[0], tile[1], rank, world,
                             threads, rgb.ctypes.data, rgba.ctypes.data,
                             rays.ctypes.data if want_rays else None, C.byref(st))
        if r != 0:
            raise RuntimeError("rro_render failed")
        return dict(rgb=rgb, rgba8=rgba, stats=st, rays=rays)


def asset(name):
    return os.path.join(ASSETS, name)
