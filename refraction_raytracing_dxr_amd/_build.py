"""Builds librrdxr.so (HIP kernels + C ABI + host C++) for gfx950 with hipcc, in tree.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box.
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "librrdxr.so")
DEMO = os.path.join(PKG, "rrdemo")

DEVICE_SOURCES = ["rr_bvh_build.hip", "rr_render.hip", "rr_render_stream.hip"]
HOST_SOURCES = ["rr_capi.cpp", "host/rr_host_camera.cpp", "host/rr_host_mesh.cpp", "host/rr_host_image.cpp", "host/rr_host_partition.cpp",
                "host/Mesh.cpp", "host/RefractionDemo.cpp"]
HEADERS = ["rr_types.h", "rr_device.h", "rr_launch.h", "rr_render_common.h", "host/Mesh.hpp", "host/RefractionDemo.hpp",
           "../../include/rrdxr.h"]

# -ffp-contract=off: the arithmetic contract -- FMAs only where fmaf is written
COMMON = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function"]
DEVICE_FLAGS = ["--offload-arch=gfx950", "-fno-gpu-rdc"] + os.environ.get("RR_EXTRA_DEFINES", "").split()


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile everything into librrdxr.so; returns its path."""
    hipcc = _hipcc()
    srcs = [os.path.join(CSRC, s) for s in DEVICE_SOURCES + HOST_SOURCES]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    stamp = os.path.join(objdir, "flags.txt")          # the flags the objects were built with: a change rebuilds everything
    flags = " ".join(COMMON + DEVICE_FLAGS + DEVICE_SOURCES)
    if not os.path.exists(stamp) or open(stamp).read() != flags:
        force = True
    objs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + deps[len(srcs):]):
            cmd = [hipcc, "-c", "-x", "hip", s, "-o", o] + COMMON + DEVICE_FLAGS
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-o", LIB] + objs + ["--offload-arch=gfx950", "-fno-gpu-rdc"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(flags)
    demo_src = os.path.join(CSRC, "tools", "rrdemo.cpp")
    if force or _stale(DEMO, [demo_src, LIB]):
        cmd = [hipcc, demo_src, "-o", DEMO, "-O2", "-std=c++17", "-L" + PKG, "-lrrdxr", "-Wl,-rpath,$ORIGIN"]
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=False, verbose=True))
