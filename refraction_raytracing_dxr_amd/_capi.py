"""ctypes binding of include/rrdxr.h (the C ABI in librrdxr.so).  No torch types cross this line."""
import ctypes as C
import os

import numpy as np

from . import _build

RR_OK = 0
STATUS_NAMES = {0: "RR_OK", 1: "RR_ERR_INVALID_ARGUMENT", 2: "RR_ERR_NO_DEVICE", 3: "RR_ERR_DEVICE",
                4: "RR_ERR_OUT_OF_MEMORY", 5: "RR_ERR_STATE", 6: "RR_ERR_IO", 7: "RR_ERR_UNSUPPORTED",
                8: "RR_ERR_TRAVERSAL_OVERFLOW"}

DISPATCH_FLOAT_OUTPUT = 0x1
DISPATCH_COLLECT_STATS = 0x2
DISPATCH_TIME_KERNEL = 0x4
DISPATCH_KEEP_COUNTERS = 0x8
DISPATCH_TILES_RGB8 = 0x10
DISPATCH_TONEMAP_REINHARD = 0x20
DISPATCH_DEBUG_NO_CULL = 0x40
BUILD_PREFER_FAST_TRACE = 0x4
BUILD_PREFER_FAST_BUILD = 0x8
RAY_FLAG_CULL_BACK = 0x10
RAY_FLAG_CULL_FRONT = 0x20
INSTANCE_FLAG_CULL_DISABLE = 0x1
INSTANCE_FLAG_FRONT_CCW = 0x2

VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("norm", "<f4", 3), ("uv", "<f4", 2)])
INSTANCE_DTYPE = np.dtype([("transform", "<f4", 12), ("instance_id_mask", "<u4"), ("hitgroup_flags", "<u4"),
                           ("blas", "<u8")])
RAY_DTYPE = np.dtype([("origin", "<f4", 3), ("tmin", "<f4"), ("dir", "<f4", 3), ("tmax", "<f4"),
                      ("flags", "<u4"), ("pad", "<u4", 3)])
HIT_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("prim", "<u4"), ("inst", "<u4"), ("hit", "<u4")])
# 64-byte BVH2 node: per plane the pair (child 0, child 1); c = child refs (>= 0 node, < 0 leaf ~ref)
NODE_DTYPE = np.dtype([("lox", "<f4", 2), ("loy", "<f4", 2), ("loz", "<f4", 2), ("hix", "<f4", 2),
                       ("hiy", "<f4", 2), ("hiz", "<f4", 2), ("c", "<i4", 2), ("pad", "<u4", 2)])
TRI_DTYPE = np.dtype([("v0", "<f4", 3), ("prim", "<u4"), ("e1", "<f4", 3), ("pad1", "<u4"),
                      ("e2", "<f4", 3), ("pad2", "<u4")])
assert VERTEX_DTYPE.itemsize == 32 and INSTANCE_DTYPE.itemsize == 64 and RAY_DTYPE.itemsize == 48
QNODE_DTYPE = np.dtype([("lox", "<f2", 2), ("loy", "<f2", 2), ("loz", "<f2", 2), ("hix", "<f2", 2),
                        ("hiy", "<f2", 2), ("hiz", "<f2", 2), ("c", "<i4", 2)])
assert QNODE_DTYPE.itemsize == 32
assert HIT_DTYPE.itemsize == 24 and NODE_DTYPE.itemsize == 64 and TRI_DTYPE.itemsize == 48


class SceneConstants(C.Structure):
    _fields_ = [("proj_inv", C.c_float * 16), ("camera_loc", C.c_float * 4)]


class DispatchParams(C.Structure):
    _fields_ = [("max_refract", C.c_int32), ("max_reflect", C.c_int32), ("ior", C.c_float),
                ("tmin_primary", C.c_float), ("tmax_primary", C.c_float),
                ("tmin_secondary", C.c_float), ("tmax_secondary", C.c_float), ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("primary", C.c_uint64), ("secondary", C.c_uint64), ("hits", C.c_uint64),
                ("misses", C.c_uint64), ("terminal_hits", C.c_uint64), ("tir", C.c_uint64),
                ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64), ("pixels", C.c_uint64),
                ("stats_valid", C.c_uint32), ("traversal_overflow", C.c_uint32), ("bvh_depth", C.c_uint32),
                ("render_kernel", C.c_uint32), ("node_trips", C.c_uint64), ("leaf_trips", C.c_uint64),
                ("shade_passes", C.c_uint64), ("waves", C.c_uint64), ("background_waves", C.c_uint64),
                ("clock_ticks", C.c_uint64), ("clock_ref_ticks", C.c_uint64), ("render_kernel_name", C.c_char * 96)]

    @property
    def clock_ghz(self):
        """shader clock of the COLLECT_STATS launches (s_memtime / s_memrealtime x 100 MHz), 0.0 without them"""
        return self.clock_ticks / self.clock_ref_ticks * 0.1 if self.clock_ref_ticks else 0.0


class MeshPartition(C.Structure):
    """rr_mesh_partition: mesh tiles dealt round robin, background tiles to rank 0"""
    _fields_ = [("tiles_x", C.c_uint32), ("n_tiles", C.c_uint32), ("rect_x0", C.c_uint32), ("rect_y0", C.c_uint32),
                ("rect_w", C.c_uint32), ("rect_h", C.c_uint32), ("n_mesh_tiles", C.c_uint32), ("n_bg_tiles", C.c_uint32),
                ("max_mesh_tiles_per_rank", C.c_uint32), ("world", C.c_uint32), ("rank0_rounds", C.c_uint32), ("pad", C.c_uint32)]


# every symbol include/rrdxr.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "rr_abi_version": (C.c_uint32, []),
    "rr_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "rr_destroy": (C.c_int, [_P]),
    "rr_last_error": (C.c_char_p, [_P]),
    "rr_set_stream": (C.c_int, [_P, _P]),
    "rr_reset_stream": (C.c_int, [_P]),
    "rr_wait": (C.c_int, [_P]),
    "rr_upload_mesh": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "rr_upload_envmap": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "rr_build_blas": (C.c_int, [_P, C.c_uint32]),
    "rr_build_blas_ex": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "rr_build_tlas": (C.c_int, [_P, _P, C.c_uint32]),
    "rr_set_camera": (C.c_int, [_P, C.POINTER(SceneConstants)]),
    "rr_set_tile_partition": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "rr_dispatch_rays": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(DispatchParams)]),
    "rr_dispatch_rays_batch": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, _P, C.POINTER(DispatchParams)]),
    "rr_read_frame": (C.c_int, [_P, _P, _P]),
    "rr_read_frame_slice": (C.c_int, [_P, C.c_uint32, _P, _P]),
    "rr_local_tile_count": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rr_export_tiles": (C.c_int, [_P, _P]),
    "rr_assemble_tiles": (C.c_int, [_P, _P, C.c_uint32, _P]),
    "rr_get_stats": (C.c_int, [_P, C.POINTER(Stats)]),
    "rr_render_orbit": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(DispatchParams), C.POINTER(C.c_float),
                                  C.c_float, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float]),
    "rr_render_orbit_to_host": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(DispatchParams), C.POINTER(C.c_float),
                                          C.c_float, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "rr_render_orbit_sharded": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(DispatchParams), C.POINTER(C.c_float),
                                          C.c_float, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float,
                                          _P, C.c_uint64]),
    "rr_render_orbit_sharded_lane": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(DispatchParams), C.POINTER(C.c_float),
                                               C.c_float, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float,
                                               _P, C.c_uint64, C.c_uint32]),
    "rr_lane_join": (C.c_int, [_P, C.c_uint32]),
    "rr_mesh_partition_for_orbit": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_uint32, C.c_float, C.c_float, C.c_float,
                                              C.c_float, C.POINTER(MeshPartition)]),
    "rr_render_orbit_mesh_sharded_lane": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(DispatchParams), C.POINTER(C.c_float), C.c_float,
                                                    C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float, _P, C.c_uint64, _P, C.c_uint64,
                                                    C.c_uint32]),
    "rr_assemble_frames_mesh_rgb8": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64, _P, C.c_uint64, C.POINTER(MeshPartition), C.c_uint32, C.c_uint32,
                                               C.c_uint32, _P, C.c_uint64]),
    "rr_set_frames_in_flight": (C.c_int, [_P, C.c_uint32]),
    "rr_assemble_frames": (C.c_int, [_P, _P, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                     _P, C.c_uint64]),
    "rr_assemble_frames_rgb8": (C.c_int, [_P, _P, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32,
                                          _P, C.c_uint64]),
    "rr_timing_begin": (C.c_int, [_P]),
    "rr_timing_end": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "rr_kernel_time": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "rr_trace_rays": (C.c_int, [_P, _P, C.c_uint32, _P]),
    "rr_env_lookup": (C.c_int, [_P, _P, C.c_uint32, _P]),
    "rr_comm_unique_id": (C.c_int, [_P]),
    "rr_comm_init": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(_P)]),
    "rr_comm_destroy": (C.c_int, [_P]),
    "rr_gather_frames": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, C.c_uint64, C.c_int]),
    "rr_device_alloc": (C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    "rr_device_free": (C.c_int, [_P, _P]),
    "rr_device_read": (C.c_int, [_P, _P, _P, C.c_uint64]),
    "rr_download_blas": (C.c_int, [_P, C.c_uint32, _P, C.POINTER(C.c_uint32), _P, C.POINTER(C.c_uint32)]),
    "rr_host_register": (C.c_int, [_P, _P, C.c_size_t]),
    "rr_host_unregister": (C.c_int, [_P, _P]),
    "rr_download_qnodes": (C.c_int, [_P, C.c_uint32, _P, C.POINTER(C.c_uint32), C.POINTER(C.c_float)]),
    "rr_default_dispatch_params": (None, [C.POINTER(DispatchParams)]),
    "rr_host_camera_orbit": (C.c_int, [C.c_float] * 5 + [C.POINTER(SceneConstants)]),
    "rr_host_screen_rect": (C.c_int, [C.POINTER(C.c_float), C.POINTER(SceneConstants), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    "rr_host_mesh_tile_home": (C.c_int, [C.POINTER(MeshPartition), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rr_host_mesh_tiles_of_rank": (C.c_uint32, [C.POINTER(MeshPartition), C.c_uint32]),
    "rr_host_mesh_partition": (C.c_int, [C.POINTER(C.c_float), C.POINTER(SceneConstants), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.POINTER(MeshPartition)]),
    "rr_host_mesh_load_obj": (C.c_int, [C.c_char_p, C.POINTER(_P), C.POINTER(C.c_uint32), C.POINTER(_P),
                                        C.POINTER(C.c_uint32)]),
    "rr_host_mesh_load_obj_ex": (C.c_int, [C.c_char_p, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_uint32), C.POINTER(_P),
                                           C.POINTER(C.c_uint32)]),
    "rr_host_image_loadf": (_P, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]),
    "rr_host_image_write_hdr": (C.c_int, [C.c_char_p, C.c_int, C.c_int, _P]),
    "rr_host_free": (None, [_P]),
    "rr_host_validate_positions": (C.c_int, [_P, C.c_uint32, _P]),
}

_lib = None


class RRError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("%s: %s" % (STATUS_NAMES.get(status, status), msg))
        self.status = status


def lib():
    """Loads librrdxr.so, building it first if it is missing or stale.  No fallback: if the HIP
    library cannot be built or loaded this raises."""
    global _lib
    if _lib is None:
        path = _build.build()
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)      # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.rr_abi_version() != 3:
            raise RuntimeError("librrdxr.so ABI version mismatch")
        _lib = L
    return _lib


def lib_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "librrdxr.so")
