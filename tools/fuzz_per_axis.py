"""GPU box: seeded campaign over the scenes whose boxes do not bound their contents (trees built over a part of the
shutter interval: tests/scenegen.py), more seeds than the test suite runs. Closest hits of the HIP kernels against the
CPU oracle's per-axis rule on the same f32 boxes: primitive, side / wrapper, bits of t, and the traversal work.
    python tools/fuzz_per_axis.py [N_SEEDS]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import oracle
from raytracer_amd import abi, lib
from scenegen import clipping_scene_pair, nested_cover_scene_pair, nested_interval_scene_pair, random_rays


def main(n_seeds):
    gpu, orc = lib.get(), oracle.get()
    bad = 0
    for seed in range(n_seeds):
        for make in (clipping_scene_pair, nested_cover_scene_pair, nested_interval_scene_pair):
            rng = np.random.default_rng(7000 + seed)
            roots = make(rng, gpu, orc)
            hs = lib.HostScene(gpu, roots[0])
            assert hs.desc.flags & abi.RT_DESC_BOXES_MAY_NOT_BOUND
            ds = lib.DeviceScene(hs)
            rays = random_rays(rng, 20000)
            oh, _, ost = orc.trace(roots[1], rays, seed=5, mode=oracle.AABB_PER_AXIS_F32)
            gh, work = ds.trace_work(rays, seed=5)
            same = (np.array_equal(oh["prim_id"], gh["prim_id"]) and np.array_equal(oh["sub"], gh["sub"])
                    and np.array_equal(oh["t"].view(np.uint64), gh["t"].view(np.uint64)))
            same_work = work["inner"] == ost["box_tests"] and work["msphere"] == ost["msphere"] and work["sphere"] == ost["sphere"]
            rays["tmin"], rays["tmax"] = 1e-4, np.inf
            rays["pixel"] = rng.permutation(256 * 128)[:len(rays)]
            rays["sample"] = 0
            rays["bounce"] = rng.integers(0, 4, len(rays))
            p = lib.make_params(256, 128, 0, 1, [0.3, 0.5, 0.7], seed=5)
            want = orc.path_step(roots[1], p, rays, mode=oracle.AABB_PER_AXIS_F32)
            got = ds.path_step(p, rays)
            same_step = (np.array_equal(want["prim_id"], got["prim_id"]) and np.array_equal(want["t"].view(np.uint64), got["t"].view(np.uint64))
                         and np.array_equal(want["scattered"], got["scattered"]) and np.array_equal(want["o"], got["o"]))
            if not (same and same_work and same_step):
                bad += 1
                print("seed %d %s: hits %s work %s step %s (%d hits differ)" % (seed, make.__name__, same, same_work, same_step,
                                                                             int((oh["prim_id"] != gh["prim_id"]).sum())))
            ds.close(); hs.close(); orc.api.scene_free(roots[1])
    print("%d seeds x 3 scene families: %d mismatches" % (n_seeds, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(int(sys.argv[1]) if len(sys.argv) > 1 else 20))
