"""GPU box: wall time of the first and of later rt_render calls on a scene (what a caller waits for before the first
picture), small and large.     python tools/first_call_time.py [SCENE_ID]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import scene_kwargs
from raytracer_amd import lib


def main(sid):
    gpu = lib.get()
    t = time.time()
    hs = lib.HostScene.build(sid, seed=1, **scene_kwargs(sid))
    print("host scene %.3f s" % (time.time() - t))
    t = time.time()
    ds = lib.DeviceScene(hs)
    print("rt_scene_create %.3f s" % (time.time() - t))
    su = gpu.host.scene_setup(sid)
    cam = gpu.host.camera(su)
    for W, H, spp in ((64, 64, 4), (64, 64, 4), (800, 800, 64), (800, 800, 64), (800, 800, 1024), (800, 800, 1024)):
        t = time.time()
        acc, st = ds.render(cam, lib.make_params(W, H, 0, spp, list(su.background)))
        print("render %dx%d x %d spp: %.3f s wall, %.1f ms in kernels" % (W, H, spp, time.time() - t, st.ms_total))
    ds.close(); hs.close()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 9)
