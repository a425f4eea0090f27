import sys, time
sys.path.insert(0, '.')
from raytracer_amd import lib, abi
L = lib.get()
t = time.time()
hs = lib.HostScene.build(12, flags=abi.RT_SCENE_BVH_LBVH, n_triangles=10_000_000)
t1 = time.time()
ds = lib.DeviceScene(hs)
print("HostScene.build %.2f s, DeviceScene %.2f s" % (t1 - t, time.time() - t1))
