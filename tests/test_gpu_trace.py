"""GPU parity of the closest-hit query (rt_trace_*): the HIP traversal against the committed golden
vectors and against the CPU oracle on seeded ray sets. Bar: primitive ids, box sides and wrapper ids
bit-exact; t within 4 ulp (0 ulp wherever no log10 is involved); the traversal itself must do exactly
the oracle's work (same number of box tests and primitive tests)."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle
from conftest import CONFIG_SCENES, SCENE_OPTS, SEED, load_golden, scene_kwargs, ulp_diff, oracle_mode
from raytracer_amd import abi, lib
from scenegen import aimed_rays, clipping_scene_pair, media_scene_pair, nested_cover_scene_pair, nested_interval_scene_pair, random_rays, random_scene_pair

pytestmark = pytest.mark.gpu

# seeded random scenes per test; RT_FUZZ_CASES widens the campaign for a one-off run (e.g. 100)
FUZZ_CASES = int(os.environ.get("RT_FUZZ_CASES", "12"))

T_ULP = 4  # north-star tolerance on t


def compare_hits(oh, gh, what="", media_ids=None):
    """media_ids: the prim ids of the scene's media; given them, t must agree to the bit on everything else (only a
    medium's t goes through log10, which the device's math library and the C library round differently)."""
    assert np.array_equal(oh["prim_id"], gh["prim_id"]), what + ": primitive ids differ"
    assert np.array_equal(oh["sub"], gh["sub"]), what + ": box side / wrapper ids differ"
    hit = oh["prim_id"] != abi.RT_MISS
    d = ulp_diff(oh["t"][hit], gh["t"][hit])
    assert d.max(initial=0) <= T_ULP, what + ": t off by %d ulp" % d.max(initial=0)
    if media_ids is not None:
        solid = ~np.isin(oh["prim_id"][hit], np.asarray(list(media_ids), dtype=np.uint32))
        assert d[solid].max(initial=0) == 0, what + ": t of a hit that is no medium's off by %d ulp" % d[solid].max(initial=0)
    # u, v go through atan2/asin on spheres (libm vs device math): a few ulp of values in [0, 1]
    # (a NaN on both sides is agreement: a ray lying in the plane of a rect has t = 0 / 0, which the reference's range
    # and bounds checks let through, src/aarect.rs:31-39)
    for c in ("u", "v"):
        a, b = oh[c][hit], gh[c][hit]
        both_nan = np.isnan(a) & np.isnan(b)
        assert np.abs(np.where(both_nan, 0.0, a - b)).max(initial=0) < 1e-14, what + ": " + c + " differs"
    return d


@pytest.mark.parametrize("sid", sorted(SCENE_OPTS))
def test_golden_ray_set(gpu, sid):
    g = load_golden(sid)
    hs = lib.HostScene.build(sid, seed=SEED, **scene_kwargs(sid))
    ds = lib.DeviceScene(hs)
    hits, _ = ds.trace(g["rays"], seed=SEED)
    d = compare_hits(g["hits"], hits, "scene %d" % sid)
    recs = ds.trace_records(g["rays"], seed=SEED)
    gr = g["records"]
    hit = g["hits"]["prim_id"] != abi.RT_MISS
    assert np.array_equal(recs["prim_id"], gr["prim_id"]) and np.array_equal(recs["sub"], gr["sub"])
    assert np.array_equal(recs["front_face"], gr["front_face"])
    exact = hit.copy()
    exact[hit] = d == 0
    # where t is bit-identical so are the point and the normal (only + - * / sqrt are involved)
    assert np.array_equal(recs["p"][exact], gr["p"][exact])
    assert np.array_equal(recs["normal"][exact], gr["normal"][exact])
    rest = hit & ~exact  # ConstantMedium hits: t carries a log10
    assert np.allclose(recs["p"][rest], gr["p"][rest], rtol=1e-13, atol=1e-11)
    assert np.array_equal(recs["normal"][rest], gr["normal"][rest])
    if sid not in (8, 9):
        assert d.max(initial=0) == 0, "t must be bit-identical in scenes without media"
    ds.close()
    hs.close()


@pytest.mark.parametrize("sid", CONFIG_SCENES)
def test_seeded_ray_set_and_work_vs_oracle(gpu, orc, sid):
    seed = 0xC0FFEE + sid
    kw = scene_kwargs(sid)
    hs = lib.HostScene.build(sid, seed=seed, **kw)
    ds = lib.DeviceScene(hs)
    osc = orc.api.scene_build(sid, seed=seed, **kw)
    su = orc.api.scene_setup(sid)
    cam = orc.api.camera(su)
    W = 256
    p = orc.make_params(W, int(W / su.aspect_ratio), 0, 4, list(su.background), seed=seed)
    rays = orc.harvest(osc, cam, p, n_paths=40000, max_rays=1 << 16, max_bounce=3)
    assert len(rays) > 30000
    oh, _, ost = orc.trace(osc, rays, seed=seed, mode=oracle_mode(hs))
    gh, work = ds.trace_work(rays, seed=seed)
    compare_hits(oh, gh, "scene %d" % sid)
    gh2, _ = ds.trace(rays, seed=seed)
    assert np.array_equal(gh, gh2), "instrumented and plain kernels disagree"
    # same traversal: same number of box tests and of primitive tests of every kind
    assert work["inner"] == ost["box_tests"]
    assert work["sphere"] == ost["sphere"] and work["msphere"] == ost["msphere"]
    assert work["box6"] == ost["box6"] and work["triangle"] == ost["triangle"] and work["medium"] == ost["medium"]
    assert work["rect"] == ost["rect"] - 6 * ost["box6"]  # the oracle counts the six sides of a box as rect tests
    ds.close()
    hs.close()
    orc.api.scene_free(osc)


@pytest.mark.parametrize("sid", [9, 12])
def test_widest_axis_tree_parity_and_work(gpu, orc, sid):
    # the optional split rule (widest axis instead of random) on both sides: same tree, same hits, same work
    g = load_golden(sid)
    kw = scene_kwargs(sid)
    hs = lib.HostScene.build(sid, seed=SEED, flags=abi.RT_SCENE_BVH_WIDEST_AXIS, **kw)
    ds = lib.DeviceScene(hs)
    osc = orc.api.scene_build(sid, seed=SEED, flags=abi.RT_SCENE_BVH_WIDEST_AXIS, **kw)
    oh, _, ost = orc.trace(osc, g["rays"], seed=SEED, mode=oracle_mode(hs))
    gh, work = ds.trace_work(g["rays"], seed=SEED)
    compare_hits(oh, gh, "scene %d, widest-axis tree" % sid)
    assert work["inner"] == ost["box_tests"] and work["triangle"] == ost["triangle"] and work["box6"] == ost["box6"]
    # and against the reference-order tree: t is independent of the tree
    assert np.array_equal(gh["t"].view(np.uint64), g["hits"]["t"].view(np.uint64)) or sid == 9
    ds.close()
    hs.close()
    orc.api.scene_free(osc)


def test_forked_host_bvh_build_is_the_oracles_tree(gpu, orc):
    # 200 000 triangles: the product host builds the subtrees of the top levels on separate threads, the oracle builds
    # sequentially; same hits and exactly the same traversal work means the same tree
    kw = dict(n_triangles=200000)
    hs = lib.HostScene.build(12, seed=SEED, **kw)
    ds = lib.DeviceScene(hs)
    osc = orc.api.scene_build(12, seed=SEED, **kw)
    su = orc.api.scene_setup(12)
    cam = orc.api.camera(su)
    p = orc.make_params(128, 128, 0, 2, list(su.background), seed=SEED)
    rays = orc.harvest(osc, cam, p, n_paths=20000, max_rays=1 << 15, max_bounce=3)
    oh, _, ost = orc.trace(osc, rays, seed=SEED, mode=oracle_mode(hs))
    gh, work = ds.trace_work(rays, seed=SEED)
    compare_hits(oh, gh, "200k-triangle mesh")
    assert work["inner"] == ost["box_tests"] and work["triangle"] == ost["triangle"] and work["box6"] == ost["box6"]
    ds.close()
    hs.close()
    orc.api.scene_free(osc)


@pytest.mark.parametrize("sid", [9, 12])
def test_large_ray_pool(gpu, sid):
    # 2.6 M rays (the golden set repeated in a shuffled order): the pool is large enough for the waves to reserve
    # rays in chunks through the shared cursor instead of taking fixed shares; every copy must get the golden hit
    g = load_golden(sid)
    reps = 640
    rng = np.random.default_rng(sid)
    order = rng.permutation(len(g["rays"]) * reps) % len(g["rays"])
    hs = lib.HostScene.build(sid, seed=SEED, **scene_kwargs(sid))
    ds = lib.DeviceScene(hs)
    hits, _ = ds.trace(g["rays"][order], seed=SEED)
    want = g["hits"][order]
    assert np.array_equal(hits["prim_id"], want["prim_id"]) and np.array_equal(hits["sub"], want["sub"])
    hit = want["prim_id"] != abi.RT_MISS
    assert ulp_diff(want["t"][hit], hits["t"][hit]).max(initial=0) <= T_ULP
    ds.close()
    hs.close()


@pytest.mark.parametrize("case", range(FUZZ_CASES))
def test_random_scenes_all_primitive_kinds(gpu, orc, case):
    rng = np.random.default_rng(1234 + case)
    builders, (hroot, oroot) = random_scene_pair(rng, gpu, orc, n_obj=int(rng.integers(4, 40)))
    hs = lib.HostScene(gpu, hroot)
    ds = lib.DeviceScene(hs)
    rays = random_rays(rng, 20000)
    oh, orec, _ = orc.trace(oroot, rays, seed=77, mode=oracle_mode(hs), records=True)
    gh, _ = ds.trace(rays, seed=77)
    media = hs.table("media")
    compare_hits(oh, gh, "random scene %d" % case, media_ids=media["prim_id"] if len(media) else [])
    grec = ds.trace_records(rays, seed=77)
    hit = oh["prim_id"] != abi.RT_MISS
    assert hit.sum() > 40
    assert np.array_equal(orec["front_face"], grec["front_face"])
    assert np.allclose(orec["normal"][hit], grec["normal"][hit], rtol=0, atol=1e-15)
    assert np.allclose(orec["p"][hit], grec["p"][hit], rtol=1e-13, atol=1e-11)
    ds.close()
    hs.close()
    orc.api.scene_free(oroot)


def test_media_asked_before_the_walk(gpu, orc, monkeypatch):
    # World-space media bounded by one sphere leave the library's copy of the entry stream (the first two in walk order)
    # and are asked before the walk. Same hits as the oracle doing likewise (and the same work), as the reference's own
    # order (oracle in its reference mode), and as the library walking them in place.
    hroot, oroot = media_scene_pair(gpu, orc)
    hs = lib.HostScene(gpu, hroot)
    rng = np.random.default_rng(99)
    rays = random_rays(rng, 40000)
    oh, _, ost = orc.trace(oroot, rays, seed=31, mode=oracle_mode(hs))
    assert len(set(oh["prim_id"][oh["prim_id"] != abi.RT_MISS])) >= 10  # every kind of medium above is among the hits
    ref, _, ost_ref = orc.trace(oroot, rays, seed=31, mode=oracle.AABB_REFERENCE)
    assert np.array_equal(ref["prim_id"], oh["prim_id"]) and np.array_equal(ref["t"].view(np.uint64), oh["t"].view(np.uint64))
    ds = lib.DeviceScene(hs)
    gh, work = ds.trace_work(rays, seed=31)
    compare_hits(oh, gh, "media scene")
    assert work["inner"] == ost["box_tests"] and work["sphere"] == ost["sphere"] and work["medium"] == ost["medium"]
    assert ost["medium"] > ost_ref["medium"]  # two media are now asked by every ray, not only by those that reach them
    ds.close()
    monkeypatch.setenv("RT_NO_DEFERRED_MEDIA", "1")  # read when a scene is created: the caller's stream as it stands
    walked = lib.DeviceScene(hs)
    wh, wwork = walked.trace_work(rays, seed=31)
    assert np.array_equal(wh["prim_id"], gh["prim_id"]) and np.array_equal(wh["sub"], gh["sub"])
    assert np.array_equal(wh["t"].view(np.uint64), gh["t"].view(np.uint64))
    assert wwork["medium"] < work["medium"]  # walked in place, a medium is only asked by the rays that reach it
    walked.close()
    # the same through one bounce of the render kernels, where the answer travels with the ray record
    monkeypatch.delenv("RT_NO_DEFERRED_MEDIA")
    ds = lib.DeviceScene(hs)
    world = rays[(rays["tmin"] == 1e-4) & np.isinf(rays["tmax"])].copy()
    world["pixel"] = np.arange(len(world)) % (64 * 64)
    world["sample"] = np.arange(len(world)) // (64 * 64)
    world["bounce"] %= 50
    p = lib.make_params(64, 64, 0, int(world["sample"].max()) + 1, [0.1, 0.2, 0.3], seed=31)
    steps = ds.path_step(p, world)
    osteps = orc.path_step(oroot, orc.make_params(64, 64, 0, int(world["sample"].max()) + 1, [0.1, 0.2, 0.3], seed=31), world, mode=oracle_mode(hs))
    assert np.array_equal(steps["prim_id"], osteps["prim_id"]) and np.array_equal(steps["sub"], osteps["sub"])
    hit = osteps["prim_id"] != abi.RT_MISS
    assert ulp_diff(osteps["t"][hit], steps["t"][hit]).max(initial=0) <= T_ULP
    assert np.array_equal(steps["scattered"], osteps["scattered"])
    ds.close()
    hs.close()
    orc.api.scene_free(oroot)


def test_a_world_of_nothing_but_media_asked_first(gpu, orc):
    # the library's copy of the entry stream is empty: both media are asked before a walk that has nothing to visit
    roots = []
    for api in (gpu.host, orc.api):
        b = api.builder(seed=3)
        lam = b.lambertian([0.5, 0.5, 0.5])
        objs = [b.constant_medium(b.sphere([0.0, 0.0, 0.0], 3.0, lam), 0.3, [1.0, 1.0, 1.0]),
                b.constant_medium(b.sphere([1.0, 0.5, 0.0], 1.5, lam), 0.9, [0.2, 0.4, 0.9])]
        roots.append(b.finish(b.bvh(b.list(objs), 0.0, 1.0)))
    hroot, oroot = roots
    hs = lib.HostScene(gpu, hroot)
    ds = lib.DeviceScene(hs)
    rays = random_rays(np.random.default_rng(5), 5000)
    oh, _, ost = orc.trace(oroot, rays, seed=9, mode=oracle_mode(hs))
    gh, work = ds.trace_work(rays, seed=9)
    compare_hits(oh, gh, "media only")
    assert (oh["prim_id"] != abi.RT_MISS).sum() > 500
    assert work["inner"] == ost["box_tests"] == 0 and work["medium"] == ost["medium"] == 2 * len(rays) and work["sphere"] == ost["sphere"]
    ref, _, _ = orc.trace(oroot, rays, seed=9, mode=oracle.AABB_REFERENCE)
    assert np.array_equal(ref["prim_id"], oh["prim_id"]) and np.array_equal(ref["t"].view(np.uint64), oh["t"].view(np.uint64))
    su = gpu.host.scene_setup(9)
    cam = gpu.host.camera(su)
    acc, st = ds.render(cam, lib.make_params(32, 32, 0, 2, [0.1, 0.2, 0.3]))  # and a render over it does not trip either
    assert np.isfinite(acc).all() and st.rays >= 32 * 32 * 2
    ds.close(); hs.close(); orc.api.scene_free(oroot)


def test_boxes_that_do_not_bound_their_spheres(gpu, orc):
    # A tree built over t in (0, 0.2) around spheres that move over (0, 1), as the reference's random_scene does
    # (src/main.rs:315): the closest hit then depends on the box rule. The flattener flags such a scene and the
    # kernels run the reference's per-axis rule (src/aabb.rs:15-30) on it; the carried-interval rule would clip.
    rng = np.random.default_rng(99)
    roots = clipping_scene_pair(rng, gpu, orc)
    hs = lib.HostScene(gpu, roots[0])
    assert hs.desc.flags & abi.RT_DESC_BOXES_MAY_NOT_BOUND
    ds = lib.DeviceScene(hs)
    assert ds.profile() == 12  # per-axis box rule (8), and 239 entries: vote exit from the box phase (4)
    rays = random_rays(rng, 30000)
    oh, _, ost = orc.trace(roots[1], rays, seed=5, mode=oracle.AABB_PER_AXIS_F32)
    gh, work = ds.trace_work(rays, seed=5)
    d = compare_hits(oh, gh, "boxes that do not bound")
    assert d.max(initial=0) == 0 and work["inner"] == ost["box_tests"] and work["msphere"] == ost["msphere"]
    # the per-axis rule on the f32 boxes against the reference's on its f64 boxes: the same hits but for rays that pass
    # within one f32 rounding of a box face
    ref, _, _ = orc.trace(roots[1], rays, seed=5, mode=oracle.AABB_REFERENCE)
    assert (ref["prim_id"] != oh["prim_id"]).sum() <= 2
    tight, _, _ = orc.trace(roots[1], rays, seed=5, mode=oracle.AABB_TIGHT)
    assert (tight["prim_id"] != oh["prim_id"]).sum() > 20  # the test tells the two rules apart
    # the render kernels' own bounce on the same scene
    rays["tmin"], rays["tmax"] = 1e-4, np.inf
    rays["pixel"] = rng.permutation(256 * 128)[:len(rays)]
    rays["sample"] = 0
    rays["bounce"] = rng.integers(0, 4, len(rays))
    p = lib.make_params(256, 128, 0, 1, [0.3, 0.5, 0.7], seed=5)
    want = orc.path_step(roots[1], p, rays, mode=oracle.AABB_PER_AXIS_F32)
    for pr in (0, 4):
        got = ds.path_step(p, rays, profile=pr)
        assert np.array_equal(want["prim_id"], got["prim_id"]) and np.array_equal(want["t"].view(np.uint64), got["t"].view(np.uint64))
        assert np.array_equal(want["scattered"], got["scattered"]) and np.array_equal(want["o"], got["o"])
    with pytest.raises(lib.RtError):
        ds.path_step(p, rays, profile=1)
    ds.close()
    hs.close()
    orc.api.scene_free(roots[1])


def test_covering_boxes_in_nested_trees(gpu, orc):
    # Covering boxes where the decision is not the same for every node: an inner tree built over the whole motion
    # (0, 1) (its boxes hold) inside an outer tree built over (0, 0.1) (its boxes do not), moving spheres behind a
    # Translate and a RotateY (no bounds known through a wrapper: per-axis test without a covering box), a tree built on
    # the device among them (left as it is), spheres that move over (0.2, 0.7) only, and static objects.
    rng = np.random.default_rng(2024)
    roots = nested_cover_scene_pair(rng, gpu, orc)
    hs = lib.HostScene(gpu, roots[0])
    assert hs.desc.flags & abi.RT_DESC_BOXES_MAY_NOT_BOUND
    nodes = hs.table("nodes")
    inner_entries = nodes[(nodes["meta"] & 15) == abi.RT_N_INNER]
    assert set((inner_entries["meta"] >> 4).tolist()) == {0, 1, 2}  # plain, holding and covering boxes are all there
    ds = lib.DeviceScene(hs)
    rays = random_rays(rng, 30000)
    oh, _, ost = orc.trace(roots[1], rays, seed=9, mode=oracle.AABB_PER_AXIS_F32)
    gh, work = ds.trace_work(rays, seed=9)
    d = compare_hits(oh, gh, "covering boxes, nested trees")
    assert d.max(initial=0) == 0 and work["inner"] == ost["box_tests"] and work["msphere"] == ost["msphere"]
    # the reference as written (f64 boxes, no covering boxes) finds the same hits but for rays within a rounding of a face
    ref, _, rst = orc.trace(roots[1], rays, seed=9, mode=oracle.AABB_REFERENCE)
    assert (ref["prim_id"] != oh["prim_id"]).sum() <= 2
    assert ost["msphere"] < rst["msphere"]  # ... with fewer sphere tests
    ds.close()
    hs.close()
    orc.api.scene_free(roots[1])


def test_outer_tree_over_the_motion_around_an_inner_tree_that_is_not(gpu, orc):
    # an outer tree over (0, 1) around an inner tree over (0, 0.1) whose spheres move over (0, 1), wrappers around moving
    # content: no box above the inner tree holds its content (see tests/test_box_rule_tieback.py for the reference's rule)
    rng = np.random.default_rng(7)
    hroot, oroot = nested_interval_scene_pair(rng, gpu, orc)
    hs = lib.HostScene(gpu, hroot)
    ds = lib.DeviceScene(hs)
    rays = np.concatenate([aimed_rays(rng, 60000), random_rays(rng, 20000)])
    oh, _, ost = orc.trace(oroot, rays, seed=5, mode=oracle.AABB_PER_AXIS_F32)
    gh, work = ds.trace_work(rays, seed=5)
    d = compare_hits(oh, gh, "nested intervals")
    assert d.max(initial=0) == 0 and work["inner"] == ost["box_tests"] and work["msphere"] == ost["msphere"]
    ref, _, _ = orc.trace(oroot, rays, seed=5, mode=oracle.AABB_REFERENCE)
    assert ((ref["prim_id"] != gh["prim_id"]) | (ref["t"].view(np.uint64) != gh["t"].view(np.uint64))).sum() <= 2
    ds.close()
    hs.close()
    orc.api.scene_free(oroot)


def test_edge_cases(gpu, orc):
    b = gpu.host.builder()
    ob = orc.api.builder()
    roots = []
    for bb in (b, ob):
        m = bb.lambertian([0.5, 0.5, 0.5])
        ground = bb.sphere([0, -1000, 0], 1000.0, m)        # the catastrophic-cancellation case of src/hittable.rs:77
        flat = bb.triangle([1, 5, 0], [0, 6, 0], [0, 5, 0], m)  # z-flat triangle: alone in a span-1 node it is invisible
        roots.append(bb.finish(bb.bvh(bb.list([ground, flat, bb.sphere([0, 1, 0], 1.0, m)]))))
    hs = lib.HostScene(gpu, roots[0])
    ds = lib.DeviceScene(hs)
    # no rays at all
    hits, _ = ds.trace(np.zeros(0, dtype=abi.RAY_DTYPE))
    assert len(hits) == 0
    rng = np.random.default_rng(5)
    n = 4096
    rays = np.zeros(n, dtype=abi.RAY_DTYPE)
    rays["tmin"], rays["tmax"] = 1e-4, np.inf
    # grazing rays over the huge ground sphere, rays from inside the unit sphere, rays at the flat triangle
    rays["o"][: n // 2] = np.c_[rng.uniform(-50, 50, n // 2), rng.uniform(1e-6, 0.5, n // 2), rng.uniform(-50, 50, n // 2)]
    rays["d"][: n // 2] = np.c_[rng.uniform(-1, 1, n // 2), rng.uniform(-0.2, 1e-3, n // 2), rng.uniform(-1, 1, n // 2)]
    rays["o"][n // 2: 3 * n // 4] = [0, 1, 0]
    rays["d"][n // 2: 3 * n // 4] = rng.normal(size=(n // 4, 3))
    rays["o"][3 * n // 4:] = np.c_[rng.uniform(0, 1, n // 4), rng.uniform(5, 6, n // 4), np.full(n // 4, 2.0)]
    rays["d"][3 * n // 4:] = [0, 0, -1]
    rays[-1]["tmax"] = 0.5  # empty range
    rays[-2]["tmin"], rays[-2]["tmax"] = 3.0, 1.0  # inverted range
    oh, _, _ = orc.trace(roots[1], rays, mode=oracle_mode(hs))
    gh, _ = ds.trace(rays)
    d = compare_hits(oh, gh, "edge cases")
    assert d.max(initial=0) == 0
    assert (gh["prim_id"][: n // 2] != abi.RT_MISS).sum() > 500
    assert (gh["prim_id"][n // 2: 3 * n // 4] == 2).all()  # from inside: the far root
    ds.close()
    hs.close()
    orc.api.scene_free(roots[1])


def test_box_of_six_sides_edge_cases(gpu, orc):
    # Box6 (src/box6.rs:22-71) hit where the six rect tests interact: rays that start on a face, an edge or a corner,
    # run inside a face plane or along an edge, pass exactly through edges and corners, have zero direction components,
    # t ranges that end exactly on a face. Planes on integers, so that these coincidences are exact in f64. Twelve
    # boxes in one tree (culling boxes in front of them) and three in another (none).
    rng = np.random.default_rng(77)
    for n_box in (12, 3):
        lo = np.array([[3 * (k % 4) - 6, 2 * (k // 4) - 3, (k % 3) - 1] for k in range(n_box)], dtype=np.float64)
        hi = lo + np.array([[1 + k % 2, 1, 2] for k in range(n_box)], dtype=np.float64)
        roots = []
        for api in (gpu.host, orc.api):
            b = api.builder(seed=3)
            m = b.lambertian([0.5, 0.5, 0.5])
            roots.append(b.finish(b.bvh(b.list([b.box6(lo[k], hi[k], m) for k in range(n_box)]))))
        hs = lib.HostScene(gpu, roots[0])
        ds = lib.DeviceScene(hs)
        n = 24000
        rays = np.zeros(n, dtype=abi.RAY_DTYPE)
        rays["tmin"], rays["tmax"] = 1e-4, np.inf
        which = rng.integers(0, n_box, n)
        # a point on the surface lattice of a box: every coordinate at lo, hi or (for two thirds of them) in between
        frac = rng.choice([0.0, 1.0, 0.5, 0.25], size=(n, 3))
        on_box = lo[which] + frac * (hi[which] - lo[which])
        kind = rng.integers(0, 6, n)
        d = rng.normal(size=(n, 3))
        d[kind == 1] = rng.choice([-1.0, 0.0, 1.0], size=((kind == 1).sum(), 3))      # axis and diagonal directions
        d[kind == 2] = np.round(d[kind == 2] * 2) / 2                                  # coarse directions: exact crossings
        d[(d == 0).all(axis=1)] = [0.0, -1.0, 0.0]
        o = on_box.copy()
        back = kind >= 3                                                                # start outside, aimed at the lattice point
        steps = rng.choice([1.0, 2.0, 4.0, 0.5], size=n)
        o[back] = on_box[back] - d[back] * steps[back, None]
        rays["o"], rays["d"] = o, d
        odd = rng.uniform(size=n)
        rays["tmin"][odd < 0.1] = 0.0
        rays["tmin"][(odd >= 0.1) & (odd < 0.2)] = -np.inf
        rays["tmax"][(odd >= 0.2) & (odd < 0.3)] = steps[(odd >= 0.2) & (odd < 0.3)]  # ends exactly where the aimed ray arrives
        rays["tmax"][(odd >= 0.3) & (odd < 0.35)] = 0.25
        oh, _, _ = orc.trace(roots[1], rays, mode=oracle_mode(hs))
        gh, _ = ds.trace(rays)
        d_ulp = compare_hits(oh, gh, "box edge cases, %d boxes" % n_box)
        assert d_ulp.max(initial=0) == 0
        hit = gh["prim_id"] != abi.RT_MISS
        assert 0.3 < hit.mean() < 0.98 and len(np.unique(gh["sub"][hit])) == 6  # every side is hit by some ray
        ds.close()
        hs.close()
        orc.api.scene_free(roots[1])


def test_device_resident_entry_point(gpu):
    import torch
    g = load_golden(9)
    hs = lib.HostScene.build(9, seed=SEED, **scene_kwargs(9))
    ds = lib.DeviceScene(hs)
    rays = torch.from_numpy(g["rays"].view(np.uint8).copy()).cuda()
    out = torch.zeros(len(g["rays"]) * abi.HIT_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        gpu.check(gpu.lib.rt_trace_closest_device(ds.h, rays.data_ptr(), len(g["rays"]), SEED, out.data_ptr(), stream.cuda_stream))
    stream.synchronize()
    dev = out.cpu().numpy().view(abi.HIT_DTYPE)
    host, _ = ds.trace(g["rays"], seed=SEED)
    assert np.array_equal(dev, host)
    ds.close()
    hs.close()
