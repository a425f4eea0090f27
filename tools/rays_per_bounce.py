"""Developer tool (GPU box): rays entering each bounce of one batch, from renders with increasing max_depth."""
import sys

import torch

sys.path.insert(0, ".")
from raytracer_amd import abi, assets, lib  # noqa: E402

L = lib.get()
hs = lib.HostScene.build(9, image=assets.earthmap())
ds = lib.DeviceScene(hs)
su = L.host.scene_setup(9)
cam = L.host.camera(su)
W = H = 800
S = 64
accum = torch.zeros(H, W, 3, dtype=torch.float32, device="cuda")
prev = 0
for depth in range(1, 51):
    p = lib.make_params(W, H, 0, S, list(su.background), samples_per_batch=S, max_depth=depth)
    st = abi.RtRenderStats()
    ds.render_device(cam, p, accum.data_ptr(), torch.cuda.current_stream().cuda_stream, stats=st)
    print("bounce %2d rays %9d  (extend %.2f ms shade %.2f ms cumulative)" % (depth - 1, st.rays - prev, st.ms_extend, st.ms_shade), flush=True)
    prev = st.rays
