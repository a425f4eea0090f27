import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SEED = 0x5EED
# scene id -> options used for the golden vectors (tools/make_golden.py)
SCENE_OPTS = {1: {}, 3: {}, 4: {}, 5: {}, 6: {"image": True}, 7: {}, 8: {}, 9: {"image": True}, 10: {}, 11: {},
              12: {"n_triangles": 20000}}
CONFIG_SCENES = [1, 7, 8, 9, 11, 12]  # the BASELINE configurations


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def scene_kwargs(sid):
    from raytracer_amd import assets
    opt = SCENE_OPTS[sid]
    kw = {}
    if opt.get("image"):
        kw["image"] = assets.earthmap()
    if "n_triangles" in opt:
        kw["n_triangles"] = opt["n_triangles"]
    return kw


@pytest.fixture(scope="session")
def orc():
    import oracle
    return oracle.get()


@pytest.fixture(scope="session")
def rtlib():
    from raytracer_amd import lib
    return lib.get()


@pytest.fixture(scope="session")
def gpu(rtlib):
    """The product library with a GPU behind it; GPU tests fail (not skip) when the device is missing."""
    assert rtlib.device_count() >= 1, "no HIP device visible: GPU tests must run on the MI355X box"
    return rtlib


def load_golden(sid):
    z = np.load(os.path.join(GOLDEN, "trace_scene%02d.npz" % sid))
    return {k: z[k] for k in z.files}


def ulp_diff(a, b):
    """Distance in units in the last place between two float64 arrays of the same sign pattern."""
    ia = np.ascontiguousarray(a).view(np.int64)
    ib = np.ascontiguousarray(b).view(np.int64)
    return np.abs(ia - ib)


def oracle_mode(host_scene):
    """The oracle's box rule that matches what the HIP kernels run on this flattened scene: the carried-interval
    rule, or the reference's per-axis rule where the scene's boxes may not bound their contents."""
    import oracle
    from raytracer_amd import abi
    return oracle.AABB_PER_AXIS_F32 if host_scene.desc.flags & abi.RT_DESC_BOXES_MAY_NOT_BOUND else oracle.AABB_TIGHT


def golden_mode(sid):
    """The same for the committed golden vectors: scenes 1 is the reference's random_scene, whose tree is built over
    t in (0, 0.1) while its spheres move over (0, 1) (src/main.rs:315)."""
    import oracle
    return oracle.AABB_PER_AXIS_F32 if sid in (1, 2) else oracle.AABB_TIGHT
