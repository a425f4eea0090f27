"""GPU box: rt_render against the oracle over odd render parameters: tiny and odd image sizes, sample ranges far from
zero, depth limits from 1, batch sizes that do not divide the sample count, 64-bit seeds, cameras with and without
aperture and shutter.     python tools/fuzz_render_params.py [N_CASES]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import oracle
from conftest import oracle_mode, scene_kwargs
from raytracer_amd import abi, lib


def main(n_cases):
    gpu, orc = lib.get(), oracle.get()
    bad = 0
    scenes = {}
    for sid in (1, 7, 8, 9, 11):
        kw = scene_kwargs(sid)
        hs = lib.HostScene.build(sid, seed=77, **kw)
        scenes[sid] = (hs, lib.DeviceScene(hs), orc.api.scene_build(sid, seed=77, **kw), gpu.host.scene_setup(sid))
    for case in range(n_cases):
        rng = np.random.default_rng(52000 + case)
        sid = int(rng.choice(list(scenes)))
        hs, ds, osc, su = scenes[sid]
        W = int(rng.choice([2, 3, 5, 17, 31, 64, 65, 100]))
        H = int(rng.choice([2, 3, 7, 16, 33, 64]))
        s0 = int(rng.choice([0, 1, 7, 1000, 2 ** 31 - 3, 2 ** 32 - 40]))
        n = int(rng.integers(1, 9))
        depth = int(rng.choice([1, 2, 3, 10, 50]))
        spb = int(rng.choice([0, 1, 2, 3, 5]))
        seed = int(rng.integers(0, 2 ** 63)) * 2 + int(rng.integers(0, 2))
        if rng.uniform() < 0.5:
            cam = gpu.host.camera(su)
        else:
            eye = np.array(su.lookfrom[:]) + rng.normal(size=3) * 5.0
            cam = gpu.host.camera_new(list(eye), list(su.lookat[:]), [0.0, 1.0, 0.0], float(rng.uniform(10, 90)), W / H,
                                      float(rng.choice([0.0, 0.1, 2.0])), float(rng.uniform(1, 20)), 0.0, float(rng.choice([0.0, 1.0])))
        p = lib.make_params(W, H, s0, s0 + n, list(su.background), seed=seed, max_depth=depth, samples_per_batch=spb)
        acc_g, st = ds.render(cam, p)
        acc_o, ost, _ = orc.render(osc, cam, p, mode=oracle_mode(hs), threads=8)
        off = np.abs(acc_g - acc_o) > 1e-6 * np.abs(acc_o) + 1e-9
        frac = float(off.any(axis=2).mean())
        ok = st.paths == W * H * n and abs(int(st.rays) - ost["rays"]) <= 0.01 * ost["rays"] + 4 and frac <= 0.01 and np.isfinite(acc_g).all()
        if not ok:
            bad += 1
            print("case %d scene %d %dx%d samples [%d, %d) depth %d batch %d seed %#x: paths %d rays %d / %d, %.4f of the pixels differ" % (
                case, sid, W, H, s0, s0 + n, depth, spb, seed, st.paths, st.rays, ost["rays"], frac))
    print("%d renders: %d mismatches" % (n_cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 100))
