"""Host-side mirror of the reference's asset surface and frame driver, over the C ABI.

  Mesh            <-> Mesh.hpp:14-25 (load / verts / indices / upload)
  load_texture    <-> RefractionDemo.cpp:108-140 (stbi_loadf(...,3))
  camera_orbit    <-> RefractionDemo.cpp:559-566
  Renderer        <-> the D3D12 calls of RefractionDemo.cpp:272-361, 566, 580-611
  RefractionDemo  <-> RefractionDemo.hpp:9-10 (initialize / drawFrame)
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (DISPATCH_COLLECT_STATS, DISPATCH_FLOAT_OUTPUT, DISPATCH_KEEP_COUNTERS, DISPATCH_TIME_KERNEL, HIT_DTYPE, INSTANCE_DTYPE, NODE_DTYPE, RAY_DTYPE,
                    TRI_DTYPE, VERTEX_DTYPE, DispatchParams, RRError, SceneConstants, Stats)

FOV_Y = float(np.float32(52.0 / 180.0 * 3.1415))     # RefractionDemo.cpp:559
ASPECT = float(np.float32(1.333))
TILE = 32


def default_params(**kw):
    p = DispatchParams()
    _capi.lib().rr_default_dispatch_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def camera_orbit(angle, fov_y=FOV_Y, aspect=ASPECT, zn=1.0, zf=125.0):
    """-> SceneConstants for the orbit angle (frame k of the reference uses 0.01*(k+1))."""
    sc = SceneConstants()
    rc = _capi.lib().rr_host_camera_orbit(float(np.float32(angle)), fov_y, aspect, zn, zf, C.byref(sc))
    if rc:
        raise RRError(rc, "rr_host_camera_orbit")
    return sc


def scene_constants(proj_inv, camera_loc):
    sc = SceneConstants()
    sc.proj_inv[:] = [float(v) for v in np.asarray(proj_inv, np.float32).reshape(16)]
    sc.camera_loc[:] = [float(v) for v in np.asarray(camera_loc, np.float32).reshape(4)]
    return sc


def load_texture(filename, req_comp=3):
    """stbi_loadf(filename,&x,&y,&n,req_comp) -> (float32 [h,w,req_comp], channels_in_file)."""
    L = _capi.lib()
    x, y, n = C.c_int(), C.c_int(), C.c_int()
    p = L.rr_host_image_loadf(str(filename).encode(), C.byref(x), C.byref(y), C.byref(n), req_comp)
    if not p:
        raise RRError(6, "rr_host_image_loadf(%s)" % filename)
    oc = req_comp if req_comp else n.value
    arr = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(y.value, x.value, oc)).copy()
    L.rr_host_free(p)
    return arr, n.value


def write_hdr(filename, rgb):
    rgb = np.ascontiguousarray(rgb, np.float32)
    h, w, c = rgb.shape
    assert c == 3
    rc = _capi.lib().rr_host_image_write_hdr(str(filename).encode(), w, h, rgb.ctypes.data)
    if rc:
        raise RRError(rc, "rr_host_image_write_hdr")


def make_instances(transforms=None, meshes=None, masks=None, flags=None):
    """64-byte instance records (RefractionDemo.cpp:324-334 fills exactly one: identity, mask 1, flags 0)."""
    if transforms is None:
        transforms = [np.eye(4, dtype=np.float32)[:3]]
    n = len(transforms)
    inst = np.zeros(n, INSTANCE_DTYPE)
    for i, t in enumerate(transforms):
        inst["transform"][i] = np.asarray(t, np.float32).reshape(12)
        inst["instance_id_mask"][i] = (i & 0xffffff) | (((masks[i] if masks is not None else 1) & 0xff) << 24)
        inst["hitgroup_flags"][i] = ((flags[i] if flags is not None else 0) & 0xff) << 24
        inst["blas"][i] = meshes[i] if meshes is not None else 0
    return inst


class Mesh:
    """Mesh.hpp:14-25.  verts: structured array of 32-byte Vertex; indices: uint32."""

    def __init__(self):
        self.verts = np.zeros(0, VERTEX_DTYPE)
        self.indices = np.zeros(0, np.uint32)
        self.mesh_id = None

    def load(self, filename, hardened=False):
        """Mesh::load (Mesh.cpp:6-37): True on success, False if the file cannot be opened.
        hardened=True additionally accepts polygons, v / v/vt / v//vn corners and negative indices."""
        L = _capi.lib()
        v, i = C.c_void_p(), C.c_void_p()
        nv, ni = C.c_uint32(), C.c_uint32()
        rc = L.rr_host_mesh_load_obj_ex(str(filename).encode(), 1 if hardened else 0, C.byref(v), C.byref(nv), C.byref(i),
                                        C.byref(ni))
        if rc:
            return False
        verts = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_uint8)), shape=(max(nv.value, 1) * 32,))
        verts = verts[:nv.value * 32].copy().view(VERTEX_DTYPE)
        idx = np.ctypeslib.as_array(C.cast(i, C.POINTER(C.c_uint32)), shape=(max(ni.value, 1),))[:ni.value].copy()
        L.rr_host_free(v)
        L.rr_host_free(i)
        base = len(self.verts)            # the reference appends (Mesh.cpp:31-32)
        self.verts = np.concatenate([self.verts, verts])
        self.indices = np.concatenate([self.indices, (idx + base).astype(np.uint32)])
        return True

    def upload(self, device):
        """Mesh::upload (Mesh.cpp:55-94); `device` is a Renderer."""
        self.mesh_id = device.upload_mesh(self.verts, self.indices)
        return self.mesh_id

    def raytracingGeometry(self):
        return dict(mesh_id=self.mesh_id, vertex_count=len(self.verts), index_count=len(self.indices),
                    vertex_stride=32)


class Renderer:
    """One context on one GPU (one process per GPU; see dist.py for the sharded frame)."""

    def __init__(self, device=0):
        self._L = _capi.lib()
        h = C.c_void_p()
        rc = self._L.rr_create(int(device), C.byref(h))
        if rc:
            raise RRError(rc, "rr_create(device=%d): no usable gfx950 device" % device)
        self._h = h
        self.device = int(device)
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.rr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc:
            raise RRError(rc, "%s: %s" % (what, self._L.rr_last_error(self._h).decode()))

    def set_stream(self, hip_stream):
        """run on a caller-owned hipStream_t handle (0 / None = HIP's default stream)"""
        self._ck(self._L.rr_set_stream(self._h, C.c_void_p(hip_stream or None)), "rr_set_stream")

    def reset_stream(self):
        self._ck(self._L.rr_reset_stream(self._h), "rr_reset_stream")

    def wait(self):
        self._ck(self._L.rr_wait(self._h), "rr_wait")

    def upload_mesh(self, verts, indices):
        verts = np.ascontiguousarray(verts)
        assert verts.dtype.itemsize == 32
        indices = np.ascontiguousarray(indices, np.uint32)
        mid = C.c_uint32()
        self._ck(self._L.rr_upload_mesh(self._h, verts.ctypes.data, len(verts), indices.ctypes.data, len(indices),
                                        C.byref(mid)), "rr_upload_mesh")
        return mid.value

    def upload_envmap(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.float32)
        h, w, c = rgb.shape
        assert c == 3
        self._ck(self._L.rr_upload_envmap(self._h, rgb.ctypes.data, w, h), "rr_upload_envmap")

    def build_blas(self, mesh_id, fast_build=False):
        """BuildRaytracingAccelerationStructure (bottom level); fast_build=True forces the plain Morton LBVH"""
        self._ck(self._L.rr_build_blas_ex(self._h, mesh_id, _capi.BUILD_PREFER_FAST_BUILD if fast_build
                                          else _capi.BUILD_PREFER_FAST_TRACE), "rr_build_blas")

    def build_tlas(self, instances=None):
        inst = make_instances() if instances is None else np.ascontiguousarray(instances, INSTANCE_DTYPE)
        self._ck(self._L.rr_build_tlas(self._h, inst.ctypes.data, len(inst)), "rr_build_tlas")

    def set_camera(self, sc):
        self._ck(self._L.rr_set_camera(self._h, C.byref(sc)), "rr_set_camera")

    def set_tile_partition(self, rank, world):
        self._ck(self._L.rr_set_tile_partition(self._h, rank, world), "rr_set_tile_partition")

    def dispatch_rays(self, width, height, params=None):
        p = params if params is not None else default_params()
        self._ck(self._L.rr_dispatch_rays(self._h, width, height, C.byref(p)), "rr_dispatch_rays")
        self.width, self.height = width, height

    def dispatch_rays_batch(self, width, height, constants, params=None):
        """DispatchRays(W, H, Depth=len(constants)): one launch, slice f rendered with constants[f]."""
        p = params if params is not None else default_params()
        arr = (SceneConstants * len(constants))(*constants)
        self._ck(self._L.rr_dispatch_rays_batch(self._h, width, height, len(constants), C.cast(arr, C.c_void_p),
                                                C.byref(p)), "rr_dispatch_rays_batch")
        self.width, self.height = width, height

    def read_frame(self, want_float=False, slice=0):
        """-> rgba8 uint8 [h,w,4] (and float32 [h,w,4] if the dispatch kept it)."""
        rgba = np.empty((self.height, self.width, 4), np.uint8)
        f32 = np.empty((self.height, self.width, 4), np.float32) if want_float else None
        self._ck(self._L.rr_read_frame_slice(self._h, slice, rgba.ctypes.data, f32.ctypes.data if want_float else None),
                 "rr_read_frame")
        return (rgba, f32) if want_float else rgba

    def local_tile_count(self, width, height):
        n, mx = C.c_uint32(), C.c_uint32()
        self._ck(self._L.rr_local_tile_count(self._h, width, height, C.byref(n), C.byref(mx)), "rr_local_tile_count")
        return n.value, mx.value

    def export_tiles(self, device_ptr):
        self._ck(self._L.rr_export_tiles(self._h, C.c_void_p(device_ptr)), "rr_export_tiles")

    def assemble_tiles(self, gathered_ptr, world, frame_ptr=None):
        self._ck(self._L.rr_assemble_tiles(self._h, C.c_void_p(gathered_ptr), world,
                                           C.c_void_p(frame_ptr) if frame_ptr else None), "rr_assemble_tiles")

    def render_orbit(self, width, height, n_frames, angle=0.01, angle_step=0.01, params=None, frames_per_dispatch=1,
                     fov_y=FOV_Y, aspect=ASPECT, zn=1.0, zf=125.0):
        """n_frames of the drawFrame loop (camera -> DispatchRays -> angle += step), asynchronous.
        Returns the angle the next frame would use."""
        p = params if params is not None else default_params()
        a = C.c_float(float(np.float32(angle)))
        self._ck(self._L.rr_render_orbit(self._h, width, height, C.byref(p), C.byref(a), float(np.float32(angle_step)),
                                         n_frames, frames_per_dispatch, fov_y, aspect, zn, zf), "rr_render_orbit")
        self.width, self.height = width, height
        return a.value

    def render_orbit_to_host(self, width, height, n_frames, angle=0.01, angle_step=0.01, params=None, frames_per_dispatch=16,
                             fov_y=FOV_Y, aspect=ASPECT, zn=1.0, zf=125.0, pin=True):
        """The drawFrame loop with every frame delivered to host memory (copies overlap rendering).
        -> uint8 [n_frames, h, w, 4]; blocks until all frames have arrived."""
        p = params if params is not None else default_params()
        a = C.c_float(float(np.float32(angle)))
        out = np.empty((n_frames, height, width, 4), np.uint8)
        pinned = pin and self._L.rr_host_register(self._h, out.ctypes.data, out.nbytes) == 0
        try:
            self._ck(self._L.rr_render_orbit_to_host(self._h, width, height, C.byref(p), C.byref(a), float(np.float32(angle_step)),
                                                     n_frames, frames_per_dispatch, fov_y, aspect, zn, zf, out.ctypes.data),
                     "rr_render_orbit_to_host")
        finally:
            if pinned:
                self._L.rr_host_unregister(self._h, out.ctypes.data)
        self.width, self.height = width, height
        return out

    def render_orbit_sharded(self, width, height, n_frames, tiles_ptr, frame_stride_bytes, angle=0.01,
                             angle_step=0.01, params=None, frames_per_dispatch=1, fov_y=FOV_Y, aspect=ASPECT, zn=1.0,
                             zf=125.0, lane=None):
        """lane=None: on the context's stream.  lane=0..3: on that internal stream, not joined until lane_join(lane)."""
        p = params if params is not None else default_params()
        a = C.c_float(float(np.float32(angle)))
        if lane is None:
            self._ck(self._L.rr_render_orbit_sharded(self._h, width, height, C.byref(p), C.byref(a),
                                                     float(np.float32(angle_step)), n_frames, frames_per_dispatch, fov_y,
                                                     aspect, zn, zf, C.c_void_p(tiles_ptr), frame_stride_bytes),
                     "rr_render_orbit_sharded")
        else:
            self._ck(self._L.rr_render_orbit_sharded_lane(self._h, width, height, C.byref(p), C.byref(a),
                                                          float(np.float32(angle_step)), n_frames, frames_per_dispatch,
                                                          fov_y, aspect, zn, zf, C.c_void_p(tiles_ptr), frame_stride_bytes,
                                                          lane), "rr_render_orbit_sharded_lane")
        self.width, self.height = width, height
        return a.value

    def mesh_partition_for_orbit(self, width, height, n_frames, angle=0.01, angle_step=0.01, fov_y=FOV_Y, aspect=ASPECT, zn=1.0, zf=125.0):
        """rr_mesh_partition of the n_frames frames starting at `angle` for this context's (rank, world)."""
        part = _capi.MeshPartition()
        self._ck(self._L.rr_mesh_partition_for_orbit(self._h, width, height, float(np.float32(angle)), float(np.float32(angle_step)), n_frames,
                                                     fov_y, aspect, zn, zf, C.byref(part)), "rr_mesh_partition_for_orbit")
        return part

    def render_orbit_mesh_sharded(self, width, height, n_frames, mesh_ptr, mesh_stride_bytes, bg_ptr, bg_stride_bytes, angle=0.01,
                                  angle_step=0.01, params=None, lane=0, fov_y=FOV_Y, aspect=ASPECT, zn=1.0, zf=125.0):
        """One DispatchRays(W, H, n_frames) of a sharded context under the mesh-tile partition, on render lane `lane`:
        this rank's mesh tiles (RGB8) to mesh_ptr, rank 0's background tiles to bg_ptr.  -> the angle after the last frame."""
        a = C.c_float(np.float32(angle))
        p = params if params is not None else default_params()
        self._ck(self._L.rr_render_orbit_mesh_sharded_lane(self._h, width, height, C.byref(p), C.byref(a), float(np.float32(angle_step)), n_frames,
                                                           fov_y, aspect, zn, zf, mesh_ptr, mesh_stride_bytes, bg_ptr, bg_stride_bytes, lane),
                 "rr_render_orbit_mesh_sharded_lane")
        self.width, self.height = width, height
        return float(a.value)

    def assemble_frames_mesh(self, gathered_ptr, rank_stride_bytes, frame_stride_bytes, bg_ptr, bg_stride_bytes, part, n_frames, width, height,
                             frames_ptr, out_stride_bytes):
        self._ck(self._L.rr_assemble_frames_mesh_rgb8(self._h, gathered_ptr, rank_stride_bytes, frame_stride_bytes, bg_ptr, bg_stride_bytes,
                                                      C.byref(part), n_frames, width, height, frames_ptr, out_stride_bytes),
                 "rr_assemble_frames_mesh_rgb8")

    def set_frames_in_flight(self, n):
        """launches of render_orbit that may overlap (1 = the reference's one-at-a-time frame loop)"""
        self._ck(self._L.rr_set_frames_in_flight(self._h, n), "rr_set_frames_in_flight")

    def lane_join(self, lane):
        self._ck(self._L.rr_lane_join(self._h, lane), "rr_lane_join")

    def assemble_frames(self, gathered_ptr, world, rank_stride_bytes, frame_stride_bytes, n_frames, width, height,
                        frames_ptr, out_stride_bytes, rgb8=False):
        fn = self._L.rr_assemble_frames_rgb8 if rgb8 else self._L.rr_assemble_frames
        self._ck(fn(self._h, C.c_void_p(gathered_ptr), world, rank_stride_bytes,
                                            frame_stride_bytes, n_frames, width, height, C.c_void_p(frames_ptr),
                                            out_stride_bytes), "rr_assemble_frames")

    def timing_begin(self):
        self._ck(self._L.rr_timing_begin(self._h), "rr_timing_begin")

    def timing_end(self):
        ms = C.c_float()
        self._ck(self._L.rr_timing_end(self._h, C.byref(ms)), "rr_timing_end")
        return ms.value

    def kernel_time(self):
        """-> (sum of render-kernel milliseconds, launches) of the dispatches flagged DISPATCH_TIME_KERNEL"""
        ms, n = C.c_float(), C.c_uint32()
        self._ck(self._L.rr_kernel_time(self._h, C.byref(ms), C.byref(n)), "rr_kernel_time")
        return ms.value, n.value

    def stats(self):
        st = Stats()
        self._ck(self._L.rr_get_stats(self._h, C.byref(st)), "rr_get_stats")
        return st

    def trace_rays(self, rays):
        rays = np.ascontiguousarray(rays, RAY_DTYPE)
        hits = np.zeros(len(rays), HIT_DTYPE)
        self._ck(self._L.rr_trace_rays(self._h, rays.ctypes.data, len(rays), hits.ctypes.data), "rr_trace_rays")
        return hits

    def env_lookup(self, dirs):
        """Miss (RayTracing.hlsl:127-137) on an [n,3] array of directions -> [n,3] texels."""
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros_like(dirs)
        self._ck(self._L.rr_env_lookup(self._h, dirs.ctypes.data, len(dirs), out.ctypes.data), "rr_env_lookup")
        return out

    def download_blas(self, mesh_id):
        nn, nt = C.c_uint32(), C.c_uint32()
        self._ck(self._L.rr_download_blas(self._h, mesh_id, None, C.byref(nn), None, C.byref(nt)), "rr_download_blas")
        nodes = np.zeros(nn.value, NODE_DTYPE)
        tris = np.zeros(nt.value, TRI_DTYPE)
        self._ck(self._L.rr_download_blas(self._h, mesh_id, nodes.ctypes.data, C.byref(nn), tris.ctypes.data,
                                          C.byref(nt)), "rr_download_blas")
        return nodes, tris

    def download_qnodes(self, mesh_id):
        """-> (QNODE_DTYPE array, grid origin float32[3], grid cell float32[3]): the nodes as traversal reads them"""
        from ._capi import QNODE_DTYPE
        nn = C.c_uint32()
        g = (C.c_float * 6)()
        self._ck(self._L.rr_download_qnodes(self._h, mesh_id, None, C.byref(nn), g), "rr_download_qnodes")
        q = np.zeros(nn.value, QNODE_DTYPE)
        self._ck(self._L.rr_download_qnodes(self._h, mesh_id, q.ctypes.data, C.byref(nn), g), "rr_download_qnodes")
        ga = np.array(list(g), np.float32)
        return q, ga[:3], ga[3:]

    # convenience: the reference's whole init sequence for one mesh + env map
    def load_scene(self, verts, indices, env_rgb, instances=None):
        mid = self.upload_mesh(verts, indices)
        self.build_blas(mid)
        if instances is None:
            instances = make_instances(meshes=[mid])
        self.build_tlas(instances)
        if env_rgb is not None:
            self.upload_envmap(env_rgb)
        return mid


class RefractionDemo:
    """RefractionDemo.hpp:9-10: initialize(hWnd, w, h) / drawFrame(), headless."""

    def __init__(self):
        self.renderer = None
        self.cubeMesh = Mesh()
        self.angle = 0.01               # static float angle (RefractionDemo.cpp:555)

    def initialize(self, width=1024, height=768, mesh_path="../shell.obj", env_path="../envMap.hdr", device=0,
                   params=None):
        self.width, self.height = width, height
        self.params = params
        self.renderer = Renderer(device)
        env, _ = load_texture(env_path, 3)                       # :527
        self.renderer.upload_envmap(env)
        if not self.cubeMesh.load(mesh_path):                    # :537 (the reference ignores the result)
            raise RRError(6, "Mesh::load(%s)" % mesh_path)
        mid = self.cubeMesh.upload(self.renderer)                # :538
        self.renderer.build_blas(mid)                            # :541
        self.renderer.build_tlas(make_instances(meshes=[mid]))

    def drawFrame(self):
        self.renderer.set_camera(camera_orbit(self.angle))       # :559-566
        self.angle = float(np.float32(self.angle) + np.float32(0.01))   # :567
        self.renderer.dispatch_rays(self.width, self.height, self.params)   # :580-594
        return self.renderer.read_frame()                        # :596-611
