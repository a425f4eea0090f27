"""GPU box (host cores only): the CPU oracle on BASELINE configs[4], the kaleidoscope with 10 M synthetic triangles, at a
quarter of the 4K image per side and one sample per pixel; the tree is the linear one the GPU run uses, built by the
oracle on the CPU. Prints as it goes."""
import sys
import time

sys.path.insert(0, ".")
import oracle  # noqa: E402
from raytracer_amd import abi  # noqa: E402

o = oracle.get()
t = time.time()
sc = o.api.scene_build(12, flags=abi.RT_SCENE_BVH_LBVH, n_triangles=10_000_000)
print("oracle scene (10 M triangles, linear tree on the CPU): %.1f s" % (time.time() - t), flush=True)
su = o.api.scene_setup(12)
su.aspect_ratio = 3840 / 2160
cam = o.api.camera(su)
W, H = 1920, 1080
# Only the carried-interval box rule: the reference's per-axis rule (src/aabb.rs:15-30) lets a ray into most boxes of a
# 10 M-leaf tree; 0.5 M samples with it did not finish in three minutes on 8 threads.
for threads in (8, 16):
    for mode, name in ((oracle.AABB_TIGHT, "carried-interval box rule"),):
        p = o.make_params(W, H, 0, 1, list(su.background))
        _, st, secs = o.render(sc, cam, p, mode=mode, threads=threads)
        print("%2d threads, %s: %.3f M samples/s, %.2f M rays/s (%d x %d x 1 spp, %.1f s)" % (threads, name, W * H / secs / 1e6, st["rays"] / secs / 1e6, W, H, secs), flush=True)
