"""Generates the committed golden vectors under tests/golden/ with the CPU oracle:
per scene ~4096 rays (primary rays and their bounce 1..3 continuations harvested from
oracle path samples) with the oracle's closest hit (prim_id, sub, raw bits of t), full
hit record, and the outcome of one ray_color iteration on it (emitted, attenuation, scattered ray). The reference itself holds no vectors for this path (its tests cover Vec3
operators only), so these pin the oracle against its own regressions and give the GPU
tests a fixed ray set; they are not outputs of the Rust binary.

Also converts the reference's example render img/kaleidoscope.png (README.md:11, the output of its scene 11 at
800x800) into tests/golden/kaleidoscope_reference_800.npz (the decoded pixels, uint8 [800, 800, 3]) when
/root/reference is present: the one output of the Rust binary the reference holds, against which whole renders of
the oracle and of the GPU path are compared."""
import json
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from raytracer_amd import assets  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 0x5EED
# BASELINE configs: Weekend (1), Cornell (7), Cornell smoke (8), final scene (9), kaleidoscope (11), kaleidoscope mesh (12);
# plus the remaining reference scenes as extra regression inputs
SCENES = {1: {}, 3: {}, 4: {}, 5: {}, 6: {"image": True}, 7: {}, 8: {}, 9: {"image": True}, 10: {}, 11: {}, 12: {"n_triangles": 20000}}


def main():
    o = oracle.get()
    meta = {"seed": SEED, "scenes": {}}
    for sid, opt in SCENES.items():
        kw = {}
        if opt.get("image"):
            kw["image"] = assets.earthmap()
        if "n_triangles" in opt:
            kw["n_triangles"] = opt["n_triangles"]
        sc = o.api.scene_build(sid, seed=SEED, **kw)
        su = o.api.scene_setup(sid)
        cam = o.api.camera(su)
        W = 200
        H = int(W / su.aspect_ratio)
        p = o.make_params(W, H, 0, 4, list(su.background), seed=SEED)
        # the box rule the HIP kernels run on this scene: the reference's per-axis rule (on the f32 boxes) where boxes
        # may not bound their contents (random_scene: tree over t in (0, 0.1), spheres moving over (0, 1)), else the
        # carried-interval rule
        mode = oracle.AABB_PER_AXIS_F32 if sid in (1, 2) else oracle.AABB_TIGHT
        rays = o.harvest(sc, cam, p, n_paths=3000, max_rays=4096, max_bounce=3, mode=mode)
        hits, recs, stats = o.trace(sc, rays, seed=SEED, mode=mode, records=True)
        hits_ref, _, stats_ref = o.trace(sc, rays, seed=SEED, mode=oracle.AABB_REFERENCE)
        hits_tight, _, _ = o.trace(sc, rays, seed=SEED, mode=oracle.AABB_TIGHT)
        steps = o.path_step(sc, p, rays, mode=mode)  # emitted / attenuation / scattered ray of each ray
        np.savez_compressed(os.path.join(OUT, "trace_scene%02d.npz" % sid), rays=rays, hits=hits, records=recs,
                            hits_reference_aabb=hits_ref, steps=steps)
        meta["scenes"][str(sid)] = {
            "options": {k: v for k, v in opt.items()}, "width": W, "height": H, "n_rays": int(len(rays)),
            "n_scattered": int(steps["scattered"].sum()), "n_emitting": int((steps["emitted"] != 0).any(axis=1).sum()),
            "n_hits": int((hits["prim_id"] != 0xFFFFFFFF).sum()),
            "box_rule": "per_axis_f32" if mode == oracle.AABB_PER_AXIS_F32 else "tight",
            "box_tests": int(stats["box_tests"]), "box_tests_reference": int(stats_ref["box_tests"]),
            "ids_differ_from_reference_rule": int((hits["prim_id"] != hits_ref["prim_id"]).sum()),
            "ids_differ_tight_vs_reference_rule": int((hits_tight["prim_id"] != hits_ref["prim_id"]).sum()),
        }
        print(sid, meta["scenes"][str(sid)])
        o.api.scene_free(sc)
    png = "/root/reference/img/kaleidoscope.png"
    if os.path.exists(png):
        from PIL import Image
        img = np.asarray(Image.open(png).convert("RGB")).astype(np.float64)
        meta["kaleidoscope_png"] = {"shape": list(img.shape), "mean_rgb": [float(x) for x in img.reshape(-1, 3).mean(0)],
                                    "min": float(img.min()), "max": float(img.max()),
                                    "corner_rgb": [int(x) for x in img[0, 0]], "centre_rgb": [int(x) for x in img[400, 400]]}
        np.savez_compressed(os.path.join(OUT, "kaleidoscope_reference_800.npz"), rgb=img.astype(np.uint8))
    elif os.path.exists(os.path.join(OUT, "meta.json")):  # keep what an earlier run recorded
        with open(os.path.join(OUT, "meta.json")) as f:
            old = json.load(f)
        if "kaleidoscope_png" in old:
            meta["kaleidoscope_png"] = old["kaleidoscope_png"]
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
