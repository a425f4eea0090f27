"""Developer tool (GPU box): time of rt_bvh_build over n random boxes (device part and whole call with its copies).
python tools/bvh_build_time.py [n]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from raytracer_amd import lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rng = np.random.default_rng(1)
c = rng.uniform(-100.0, 100.0, (n, 3)) * (1.0, 0.01, 1.0)  # a slab, like the synthetic mesh
h = rng.uniform(0.0, 0.05, (n, 3))
boxes = np.concatenate([c - h, c + h], axis=1)
L = lib.get()
for it in range(3):
    t0 = time.time()
    tokens, ms = L.bvh_build(boxes)
    print("n = %d: device %.2f ms, whole call %.1f ms" % (n, ms, (time.time() - t0) * 1e3), flush=True)
