"""GPU box: the BASELINE scenes built from many scene seeds (another random_scene, another final_scene, another mesh every
time) and with every tree-building rule, rays harvested from the oracle's own paths, closest hits and traversal work of
the HIP kernels against the oracle, then one bounce through the render kernels, then a small picture.
    python tools/fuzz_config_scenes.py [N_SEEDS]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import oracle
from conftest import oracle_mode, scene_kwargs, ulp_diff
from raytracer_amd import abi, lib


def main(n_seeds):
    gpu, orc = lib.get(), oracle.get()
    bad = 0
    runs = 0
    for k in range(n_seeds):
        seed = 0xABC000 + 977 * k
        for sid in (1, 8, 9, 12):
            for flags in (0, abi.RT_SCENE_BVH_WIDEST_AXIS, abi.RT_SCENE_BVH_SHUTTER_TIMES):
                if flags == abi.RT_SCENE_BVH_SHUTTER_TIMES and sid != 1:
                    continue
                kw = scene_kwargs(sid)
                hs = lib.HostScene.build(sid, seed=seed, flags=flags, **kw)
                ds = lib.DeviceScene(hs)
                osc = orc.api.scene_build(sid, seed=seed, flags=flags, **kw)
                su = orc.api.scene_setup(sid)
                cam = orc.api.camera(su)
                W = 128
                H = int(W / su.aspect_ratio)
                p = orc.make_params(W, H, 0, 2, list(su.background), seed=seed)
                rays = orc.harvest(osc, cam, p, n_paths=12000, max_rays=1 << 15, max_bounce=4)
                mode = oracle_mode(hs)
                oh, _, ost = orc.trace(osc, rays, seed=seed, mode=mode)
                gh, work = ds.trace_work(rays, seed=seed)
                hit = oh["prim_id"] != abi.RT_MISS
                d = ulp_diff(oh["t"][hit], gh["t"][hit]).max(initial=0)
                ok = np.array_equal(oh["prim_id"], gh["prim_id"]) and np.array_equal(oh["sub"], gh["sub"]) and d <= (4 if sid in (8, 9) else 0)
                ok_work = (work["inner"] == ost["box_tests"] and work["sphere"] == ost["sphere"] and work["msphere"] == ost["msphere"]
                           and work["box6"] == ost["box6"] and work["triangle"] == ost["triangle"] and work["medium"] == ost["medium"])
                world = rays[(rays["tmin"] == 1e-4) & np.isinf(rays["tmax"])].copy()
                world["pixel"] = np.arange(len(world)) % (W * H)
                world["sample"] = np.arange(len(world)) // (W * H)
                pp = lib.make_params(W, H, 0, int(world["sample"].max()) + 1, list(su.background), seed=seed)
                want = orc.path_step(osc, orc.make_params(W, H, 0, int(world["sample"].max()) + 1, list(su.background), seed=seed), world, mode=mode)
                got = ds.path_step(pp, world)
                hit2 = want["prim_id"] != abi.RT_MISS
                d2 = ulp_diff(want["t"][hit2], got["t"][hit2]).max(initial=0)
                ok_step = (np.array_equal(want["prim_id"], got["prim_id"]) and np.array_equal(want["sub"], got["sub"]) and d2 <= (4 if sid in (8, 9) else 0)
                           and np.array_equal(want["scattered"], got["scattered"]))
                # and whole paths: a small picture, pixel for pixel (1e-6 relative, all but 0.5 % of the pixels: a last bit
                # of log10 / sin / atan2 flips a medium hit, a checker cell or a texel now and then)
                Wi = 64
                Hi = int(Wi / su.aspect_ratio)
                pi = lib.make_params(Wi, Hi, 0, 4, list(su.background), seed=seed)
                acc_g, st = ds.render(cam, pi)
                acc_o, _, _ = orc.render(osc, cam, pi, mode=mode, threads=8)
                off = np.abs(acc_g - acc_o) > 1e-6 * np.abs(acc_o) + 1e-9
                frac = float(off.any(axis=2).mean())
                ok_step = ok_step and frac <= 0.005
                runs += 1
                if frac > 0.0:
                    print("seed %#x scene %d flags %d: %.4f of the pixels differ" % (seed, sid, flags, frac))
                if not (ok and ok_work and ok_step):
                    bad += 1
                    print("seed %#x scene %d flags %d: hits %s (t %d ulp) work %s step %s (t %d ulp); %d ids differ" % (
                        seed, sid, flags, ok, d, ok_work, ok_step, d2, int((oh["prim_id"] != gh["prim_id"]).sum())))
                ds.close(); hs.close(); orc.api.scene_free(osc)
    print("%d scene builds: %d mismatches" % (runs, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 10))
