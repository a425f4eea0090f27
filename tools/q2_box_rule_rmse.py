"""Puts a number on the one place where the GPU's box rule changes a picture of the reference: random_scene builds its
BVH over t in (0, 0.1) (src/main.rs:315) while the shutter is (0, 1), so moving spheres stick out of their boxes and
are clipped, differently by the reference's per-axis test (src/aabb.rs:15-30) and by the carried slab test the GPU
runs. CPU oracle only: same seed, same paths, three renders of BASELINE configs[0] (1200x800):
  loose  = the reference's box test on the reference's boxes (what the Rust binary draws)
  tight  = the carried slab test on the same boxes (what the GPU draws by default)
  right  = boxes built over the shutter interval (RT_SCENE_BVH_SHUTTER_TIMES): no clipping, either test
Prints per-pixel RMSE on resolved [0, 1] values and the share of pixels that differ."""
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), ".."))
import oracle  # noqa: E402
from raytracer_amd import abi  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 16
o = oracle.get()
su = o.api.scene_setup(1)
cam = o.api.camera(su)
W, H = 1200, 800
p = o.make_params(W, H, 0, spp, list(su.background))
sc = o.api.scene_build(1)
loose = o.render(sc, cam, p, mode=oracle.AABB_REFERENCE, threads=8)[0]
tight = o.render(sc, cam, p, mode=oracle.AABB_TIGHT, threads=8)[0]
sc2 = o.api.scene_build(1, flags=abi.RT_SCENE_BVH_SHUTTER_TIMES)
right = o.render(sc2, cam, p, mode=oracle.AABB_TIGHT, threads=8)[0]
right_loose = o.render(sc2, cam, p, mode=oracle.AABB_REFERENCE, threads=8)[0]
seed2 = o.render(sc, cam, o.make_params(W, H, 0, spp, list(su.background), seed=0x5EED + 1), mode=oracle.AABB_REFERENCE, threads=8)[0]
res = lambda a: np.sqrt(a / spp)
rmse = lambda a, b: float(np.sqrt(np.mean((res(a) - res(b)) ** 2)))
differ = lambda a, b: float((np.abs(a - b) > 1e-9).any(axis=2).mean())
print("spp", spp)
print("loose vs tight : rmse %.5f, pixels differing %.4f" % (rmse(loose, tight), differ(loose, tight)))
print("loose vs right : rmse %.5f, pixels differing %.4f" % (rmse(loose, right), differ(loose, right)))
print("tight vs right : rmse %.5f, pixels differing %.4f" % (rmse(tight, right), differ(tight, right)))
print("right: loose vs tight test: rmse %.2g (boxes bound their contents: the two tests agree)" % rmse(right, right_loose))
print("noise floor (loose, two seeds): rmse %.5f" % rmse(loose, seed2))
