"""GPU box: closest-hit queries with rays a renderer never makes but a caller of rt_trace can: zero, infinite, NaN and
denormal direction components, origins far away, empty and inverted t ranges, times outside the shutter. The HIP
kernels against the oracle on seeded ad-hoc scenes (ids, sides, bits of t).    python tools/fuzz_evil_rays.py [N_CASES]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import oracle
from conftest import oracle_mode, ulp_diff
from raytracer_amd import abi, lib
from scenegen import random_rays, random_scene_pair


def evil(rng, n):
    rays = random_rays(rng, n)
    k = rng.integers(0, 12, n)
    d = rays["d"]
    o = rays["o"]
    d[k == 0] = 0.0
    d[k == 1, 0] = np.inf
    d[k == 2, 1] = -np.inf
    d[k == 3, 2] = np.nan
    d[k == 4] *= 1e-310
    d[k == 5] *= 1e300
    o[k == 6] *= 1e12
    o[k == 7, 0] = np.nan
    rays["tmin"][k == 8] = 5.0
    rays["tmax"][k == 8] = 1.0
    rays["tmin"][k == 9] = -np.inf
    rays["tm"][k == 10] = rng.uniform(-3, 4, int((k == 10).sum()))
    rays["tmax"][k == 11] = np.nan
    return rays


def main(n_cases):
    gpu, orc = lib.get(), oracle.get()
    bad = 0
    for case in range(n_cases):
        rng = np.random.default_rng(64000 + case)
        builders, (hroot, oroot) = random_scene_pair(rng, gpu, orc, n_obj=int(rng.integers(4, 60)))
        hs = lib.HostScene(gpu, hroot)
        ds = lib.DeviceScene(hs)
        rays = evil(rng, 20000)
        oh, _, ost = orc.trace(oroot, rays, seed=3, mode=oracle_mode(hs))
        gh, work = ds.trace_work(rays, seed=3)
        hit = (oh["prim_id"] != abi.RT_MISS) & (gh["prim_id"] != abi.RT_MISS)
        both_nan = np.isnan(oh["t"]) & np.isnan(gh["t"])
        cmp = hit & ~both_nan
        d = ulp_diff(oh["t"][cmp], gh["t"][cmp]).max(initial=0) if cmp.any() else 0
        ids = int((oh["prim_id"] != gh["prim_id"]).sum())
        subs = int((oh["sub"] != gh["sub"]).sum())
        same_work = work["inner"] == ost["box_tests"] and work["sphere"] == ost["sphere"] and work["medium"] == ost["medium"]
        if ids or subs or d > 4 or not same_work:
            bad += 1
            wrong = np.nonzero(oh["prim_id"] != gh["prim_id"])[0][:3]
            print("case %d: %d ids, %d sides differ, t %d ulp, work equal %s; e.g. rays %s d %s o %s t [%s, %s]" % (
                case, ids, subs, d, same_work, wrong.tolist(), rays["d"][wrong].tolist(), rays["o"][wrong].tolist(),
                rays["tmin"][wrong].tolist(), rays["tmax"][wrong].tolist()))
        ds.close(); hs.close(); orc.api.scene_free(oroot)
    print("%d scenes x 20000 rays: %d with mismatches" % (n_cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 50))
