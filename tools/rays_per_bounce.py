"""Developer tool (GPU box): rays entering each bounce of one batch and the extend / shade time that bounce adds, from
renders with increasing max_depth.   python tools/rays_per_bounce.py [workload] [bvh] [spp]"""
import sys

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from raytracer_amd import abi, lib  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "final"]
bvh = sys.argv[2] if len(sys.argv) > 2 else "reference"
S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
L = lib.get()
hs = lib.HostScene.build(wl["scene"], **bench.scene_kwargs(wl, bvh))
ds = lib.DeviceScene(hs)
su = L.host.scene_setup(wl["scene"])
su.aspect_ratio = wl["width"] / wl["height"]
cam = L.host.camera(su)
W, H = wl["width"], wl["height"]
accum = torch.zeros(H, W, 3, dtype=torch.float32, device="cuda")
prev = (0, 0.0, 0.0)
for depth in range(1, 51):
    p = lib.make_params(W, H, 0, S, list(su.background), samples_per_batch=S, max_depth=depth)
    st = abi.RtRenderStats()
    ds.render_device(cam, p, accum.data_ptr(), torch.cuda.current_stream().cuda_stream, stats=st)
    print("bounce %2d rays %10d  extend %8.3f ms  shade %8.3f ms" % (depth - 1, st.rays - prev[0], st.ms_extend - prev[1], st.ms_shade - prev[2]), flush=True)
    prev = (st.rays, st.ms_extend, st.ms_shade)
