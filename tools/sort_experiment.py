"""Developer experiment (GPU box): how much does ray ordering change closest-hit throughput?
Harvests secondary rays with the oracle, orders them several ways and times rt_trace_closest."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import oracle  # noqa: E402
from raytracer_amd import assets, lib  # noqa: E402


def morton3(ix, iy, iz, bits):
    code = np.zeros(ix.shape, dtype=np.uint64)
    for b in range(bits):
        code |= ((ix >> b) & 1).astype(np.uint64) << np.uint64(3 * b)
        code |= ((iy >> b) & 1).astype(np.uint64) << np.uint64(3 * b + 1)
        code |= ((iz >> b) & 1).astype(np.uint64) << np.uint64(3 * b + 2)
    return code


def main():
    sid = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    L = lib.get()
    o = oracle.get()
    kw = dict(image=assets.earthmap()) if sid in (6, 9) else {}
    hs = lib.HostScene.build(sid, **kw)
    ds = lib.DeviceScene(hs)
    osc = o.api.scene_build(sid, **kw)
    su = o.api.scene_setup(sid)
    cam = o.api.camera(su)
    W = 800
    H = int(W / su.aspect_ratio)
    p = o.make_params(W, H, 0, 8, list(su.background))
    t0 = time.time()
    rays = o.harvest(osc, cam, p, n_paths=W * H * 4, max_rays=14_000_000, max_bounce=6)
    print("harvested", len(rays), "rays in %.1f s" % (time.time() - t0), flush=True)
    for bounce_sel, name in ((rays["bounce"] == 0, "bounce 0"), (rays["bounce"] == 1, "bounce 1"), (rays["bounce"] >= 2, "bounce 2-6")):
        r = rays[bounce_sel]
        n = len(r)
        octant = ((r["d"][:, 0] < 0).astype(np.uint64) | ((r["d"][:, 1] < 0).astype(np.uint64) << np.uint64(1)) | ((r["d"][:, 2] < 0).astype(np.uint64) << np.uint64(2)))
        lo, hi = r["o"].min(0), r["o"].max(0)
        q = np.clip(((r["o"] - lo) / (hi - lo + 1e-9) * 32).astype(np.int64), 0, 31)
        mort = morton3(q[:, 0], q[:, 1], q[:, 2], 5)
        dn = r["d"] / np.linalg.norm(r["d"], axis=1, keepdims=True)
        dq = np.clip(((dn + 1) * 0.5 * 8).astype(np.int64), 0, 7)
        dmort = morton3(dq[:, 0], dq[:, 1], dq[:, 2], 3)
        pixkey = r["pixel"].astype(np.uint64) * np.uint64(64) + r["sample"].astype(np.uint64)
        orders = {
            "harvest order (hashed pixels)": np.arange(n),
            "pixel order": np.argsort(pixkey, kind="stable"),
            "octant, pixel": np.lexsort((pixkey, octant)),
            "octant, morton(o,5b)": np.lexsort((mort, octant)),
            "morton(o,5b), octant": np.lexsort((octant, mort)),
            "morton(o), dir morton(3b)": np.lexsort((dmort, mort)),
            "dir morton(3b), morton(o)": np.lexsort((mort, dmort)),
        }
        base = None
        for k, idx in orders.items():
            rr = np.ascontiguousarray(r[idx])
            best = 1e9
            for _ in range(3):
                _, ms = ds.trace(rr)
                best = min(best, ms)
            base = base or best
            print("%-12s n=%8d  %-32s %8.3f ms  %7.1f Mrays/s  x%.2f" % (name, n, k, best, n / best / 1e3, base / best), flush=True)


if __name__ == "__main__":
    main()
