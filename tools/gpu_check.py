"""Developer check (GPU box): closest-hit parity and small-image agreement between the
HIP path and the CPU oracle for the reference scenes. Prints one line per scene."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import oracle  # noqa: E402
from raytracer_amd import abi, assets, lib  # noqa: E402


def ulp_diff(a, b):
    ia = a.view(np.int64)
    ib = b.view(np.int64)
    return np.abs(ia - ib)


def main():
    L = lib.get()
    o = oracle.get()
    print("devices", L.device_count(), flush=True)
    scenes = [int(x) for x in sys.argv[1:]] or [1, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]
    for sid in scenes:
        kw = dict(image=assets.earthmap()) if sid in (6, 9) else {}
        if sid == 12:
            kw["n_triangles"] = 20000
        hs = lib.HostScene.build(sid, **kw)
        ds = lib.DeviceScene(hs)
        osc = o.api.scene_build(sid, **kw)
        su = o.api.scene_setup(sid)
        cam = o.api.camera(su)
        cam2 = L.host.camera(su)
        assert bytes(cam) == bytes(cam2), "camera mismatch"
        W = 96
        H = int(W / su.aspect_ratio)
        p = o.make_params(W, H, 0, 8, list(su.background))
        rays = o.harvest(osc, cam, p, n_paths=4000, max_rays=16384)
        t0 = time.time()
        oh, orec, ost = o.trace(osc, rays, records=True)
        t1 = time.time()
        gh, ms = ds.trace(rays)
        grec = ds.trace_records(rays)
        gh2, work = ds.trace_work(rays)
        id_mis = int(np.sum((oh["prim_id"] != gh["prim_id"]) | (oh["sub"] != gh["sub"])))
        hit = oh["prim_id"] != abi.RT_MISS
        both = hit & (gh["prim_id"] != abi.RT_MISS)
        tu = ulp_diff(oh["t"][both], gh["t"][both])
        uvd = max(np.abs(oh["u"][both] - gh["u"][both]).max(initial=0), np.abs(oh["v"][both] - gh["v"][both]).max(initial=0))
        pd = np.abs(orec["p"][both] - grec["p"][both]).max(initial=0)
        nd = np.abs(orec["normal"][both] - grec["normal"][both]).max(initial=0)
        ff = int(np.sum(orec["front_face"][both] != grec["front_face"][both]))
        same2 = bool(np.array_equal(gh, gh2))
        print("scene %2d rays %5d hits %5d id_mismatch %d t_ulp_max %d (n>0: %d) uv_maxabs %.2e p_maxabs %.2e n_maxabs %.2e ff_mis %d "
              "work_ok %s gpu_entries/ray %.1f orc_box/ray %.1f gpu_inner/ray %.1f  gpu %.3f ms cpu %.1f ms"
              % (sid, len(rays), int(hit.sum()), id_mis, int(tu.max(initial=0)), int((tu > 0).sum()), uvd, pd, nd, ff, same2,
                 work["entries"] / len(rays), ost["box_tests"] / len(rays), work["inner"] / len(rays), ms, (t1 - t0) * 1e3), flush=True)
        if id_mis:
            bad = np.nonzero((oh["prim_id"] != gh["prim_id"]) | (oh["sub"] != gh["sub"]))[0][:5]
            for b in bad:
                print("   ray", b, rays[b], "oracle", oh[b], "gpu", gh[b])
        # small render
        acc_o, st_o, secs = o.render(osc, cam, p, threads=8)
        acc_g, st = ds.render(cam, p)
        a = np.sqrt(acc_o / 8.0)
        b = np.sqrt(acc_g.astype(np.float64) / 8.0)
        rmse = float(np.sqrt(np.mean((a - b) ** 2)))
        print("          render %dx%dx8: mean oracle %.5f gpu %.5f rmse %.5f rays oracle %d gpu %d gpu_ms %.2f (extend %.2f shade %.2f) cpu_s %.2f"
              % (W, H, a.mean(), b.mean(), rmse, st_o["rays"], st.rays, st.ms_total, st.ms_extend, st.ms_shade, secs), flush=True)
        ds.close()
        hs.close()
        o.api.scene_free(osc)


if __name__ == "__main__":
    main()
