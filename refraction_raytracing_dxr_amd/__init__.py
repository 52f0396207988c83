"""MI355X-native refraction ray tracer: the hot path of bottledspace/refraction-raytracing-dxr
(RayGen / ClosestHit / Miss + the driver's BLAS/TLAS build and TraceRay) as hand-written HIP for
gfx950 behind a C ABI (include/rrdxr.h), with a host-side mirror of the reference's
Mesh / RefractionDemo interface.  There is no CPU fallback: without librrdxr.so and a gfx950
device every render call raises."""
from ._capi import (DISPATCH_COLLECT_STATS, DISPATCH_DEBUG_NO_CULL, DISPATCH_FLOAT_OUTPUT, DISPATCH_KEEP_COUNTERS, DISPATCH_TILES_RGB8, DISPATCH_TIME_KERNEL, DISPATCH_TONEMAP_REINHARD, HIT_DTYPE, INSTANCE_DTYPE, NODE_DTYPE,
                    RAY_DTYPE, RAY_FLAG_CULL_BACK, RAY_FLAG_CULL_FRONT, TRI_DTYPE, VERTEX_DTYPE, RRError, lib,
                    lib_path)
from .host import (ASPECT, FOV_Y, Mesh, RefractionDemo, Renderer, camera_orbit, default_params, load_texture,
                   make_instances, scene_constants, write_hdr)
from . import dist, synth

__all__ = ["Mesh", "RefractionDemo", "Renderer", "camera_orbit", "default_params", "load_texture", "make_instances",
           "scene_constants", "write_hdr", "dist", "lib", "lib_path", "RRError"]
