"""Developer tool (GPU box): BASELINE configs[3] in full (final scene, 800x800, 10 000 spp) through the command-line
renderer's code path, timed, plus the distance of a 256-spp render to it (convergence)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from raytracer_amd import assets, imageio, lib  # noqa: E402

L = lib.get()
hs = lib.HostScene.build(9, image=assets.earthmap())
ds = lib.DeviceScene(hs)
su = L.host.scene_setup(9)
cam = L.host.camera(su)
bg = list(su.background)
t = time.time()
full, st = ds.render(cam, lib.make_params(800, 800, 0, 10000, bg))
secs = time.time() - t
print("10000 spp: %.2f s wall, %.1f M samples/s, %.0f M rays/s, %d batches" % (secs, 6.4e9 / secs / 1e6, st.rays / secs / 1e6, st.batches), flush=True)
part, _ = ds.render(cam, lib.make_params(800, 800, 10000, 10256, bg))
a = np.sqrt(np.clip(full / 10000.0, 0, None))
b = np.sqrt(np.clip(part / 256.0, 0, None))
print("RMSE(256 spp, 10000 spp) on display values: %.4f; mean display value %.4f" % (float(np.sqrt(((a - b) ** 2).mean())), float(a.mean())))
imageio.write_image("gpurun_out/final_10000spp.png", lib.resolve_u8(full, 10000))
