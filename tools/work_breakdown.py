"""Developer tool (GPU box): traversal work per ray by entry kind for the render path of a scene."""
import sys

import torch

sys.path.insert(0, ".")
from raytracer_amd import assets, lib  # noqa: E402


def main():
    L = lib.get()
    for sid, W, H in ((9, 800, 800), (7, 600, 600), (8, 600, 600), (1, 1200, 800)):
        kw = dict(image=assets.earthmap()) if sid == 9 else {}
        hs = lib.HostScene.build(sid, **kw)
        ds = lib.DeviceScene(hs)
        su = L.host.scene_setup(sid)
        cam = L.host.camera(su)
        acc = torch.zeros(H, W, 3, dtype=torch.float32, device="cuda")
        st, work = ds.render_work(cam, lib.make_params(W, H, 0, 4, list(su.background)), acc.data_ptr())
        print("scene", sid, "rays/sample %.2f" % (st.rays / st.paths), " per ray:", {k: round(v / st.rays, 2) for k, v in work.items()})
        ds.close(); hs.close()


if __name__ == "__main__":
    main()
