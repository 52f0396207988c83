import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected with -m gpu; if someone runs the whole suite on a box without a GPU,
    # skip them loudly instead of failing on rr_create
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (GPU tests run via gpurun)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def env_png():
    """envmap.png decoded by the product loader (640x480 RGB32F, gamma-expanded)"""
    import refraction_raytracing_dxr_amd as rr
    env, _ = rr.load_texture(os.path.join(ROOT, "refraction_raytracing_dxr_amd", "assets", "envmap.png"), 3)
    return env


from refraction_raytracing_dxr_amd.synth import procedural_env  # noqa: E402,F401  (tests import it from here)


@pytest.fixture(scope="session")
def env_hdr():
    return procedural_env()
