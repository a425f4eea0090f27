"""The oracle's GPU box modes against the reference's own rule, at scale.

The HIP kernels are compared with the oracle in the mode that mirrors what they run (carried-interval test on f32
boxes, culling boxes, covering boxes, no entries for boxes that contain the world: AABB_TIGHT / AABB_PER_AXIS_F32).
None of that machinery exists in the reference: BVHNode::hit (src/bvh.rs:101-111) tests every node with AABB::hit
as written (src/aabb.rs:15-30) on f64 boxes. These tests tie the mirrored modes back to that rule on the same ray
sets the GPU tests use: the closest hit (primitive, side / wrapper, bits of t) must not depend on the mode. A
mistake shared by the product and the mirrored oracle mode (a culling box that does not cover, a skipped node whose
box mattered, a tie at a padded face) shows up here as a difference from AABB_REFERENCE."""
import numpy as np
import pytest

import oracle
from conftest import CONFIG_SCENES, SEED, oracle_mode, scene_kwargs
from raytracer_amd import abi, lib
from scenegen import aimed_rays, clipping_scene_pair, media_scene_pair, nested_cover_scene_pair, nested_interval_scene_pair, random_rays, random_scene_pair


def same_hits(a, b, what):
    assert np.array_equal(a["prim_id"], b["prim_id"]), what + ": primitive ids differ from the reference rule"
    assert np.array_equal(a["sub"], b["sub"]), what + ": box sides / wrapper ids differ from the reference rule"
    assert np.array_equal(a["t"].view(np.uint64), b["t"].view(np.uint64)), what + ": bits of t differ from the reference rule"
    assert np.array_equal(a["u"].view(np.uint64), b["u"].view(np.uint64)) and np.array_equal(a["v"].view(np.uint64), b["v"].view(np.uint64))


def harvested(orc, osc, sid, seed, n_paths=40000):
    su = orc.api.scene_setup(sid)
    cam = orc.api.camera(su)
    W = 256
    p = orc.make_params(W, int(W / su.aspect_ratio), 0, 4, list(su.background), seed=seed)
    rays = orc.harvest(osc, cam, p, n_paths=n_paths, max_rays=1 << 16, max_bounce=3)
    assert len(rays) > 30000
    return rays, p


@pytest.mark.parametrize("sid", CONFIG_SCENES)
def test_config_scenes_30k_rays(orc, rtlib, sid):
    # the seeded ray sets of tests/test_gpu_trace.py::test_seeded_ray_set_and_work_vs_oracle
    seed = 0xC0FFEE + sid
    kw = scene_kwargs(sid)
    hs = lib.HostScene.build(sid, seed=seed, **kw)
    osc = orc.api.scene_build(sid, seed=seed, **kw)
    rays, p = harvested(orc, osc, sid, seed)
    mode = oracle_mode(hs)
    mine, _, st = orc.trace(osc, rays, seed=seed, mode=mode)
    ref, _, rst = orc.trace(osc, rays, seed=seed, mode=oracle.AABB_REFERENCE)
    if mode == oracle.AABB_PER_AXIS_F32:
        # the reference's per-axis rule on f32 planes against the same rule on its f64 planes: a ray may pass within
        # one f32 rounding of a face of a box that does not bound its sphere
        assert sid == 1
        differ = (mine["prim_id"] != ref["prim_id"]) | (mine["t"].view(np.uint64) != ref["t"].view(np.uint64))
        assert differ.mean() <= 1e-4, differ.sum()
    else:
        same_hits(mine, ref, "scene %d" % sid)
    assert st["box_tests"] < rst["box_tests"]  # ... and the mirrored mode is the cheaper one
    # one whole iteration of ray_color (hit, emitted, scatter) does not depend on the mode either
    a = orc.path_step(osc, p, rays, mode=mode)
    b = orc.path_step(osc, p, rays, mode=oracle.AABB_REFERENCE)
    if mode == oracle.AABB_TIGHT:
        assert a.tobytes() == b.tobytes()
    hs.close()
    orc.api.scene_free(osc)


def test_final_scene_with_1600_ground_boxes(orc, rtlib):
    # boxes_per_side = 40: four times the boxes of src/main.rs:591, tables beyond what the kernels stage in LDS
    seed = SEED
    kw = dict(scene_kwargs(9), boxes_per_side=40)
    hs = lib.HostScene.build(9, seed=seed, **kw)
    osc = orc.api.scene_build(9, seed=seed, **kw)
    rays, _ = harvested(orc, osc, 9, seed)
    mine, _, _ = orc.trace(osc, rays, seed=seed, mode=oracle_mode(hs))
    ref, _, _ = orc.trace(osc, rays, seed=seed, mode=oracle.AABB_REFERENCE)
    same_hits(mine, ref, "final scene, 40 x 40 boxes")
    hs.close()
    orc.api.scene_free(osc)


@pytest.mark.parametrize("tree", ["widest", "lbvh"])
def test_mesh_with_other_tree_policies(orc, rtlib, tree):
    # 200 000 triangles under the two optional tree policies: the rule must not matter whatever the tree
    flags = {"widest": abi.RT_SCENE_BVH_WIDEST_AXIS, "lbvh": abi.RT_SCENE_BVH_LBVH}[tree]
    osc = orc.api.scene_build(12, seed=SEED, flags=flags, n_triangles=200000)
    rays, _ = harvested(orc, osc, 12, SEED, n_paths=20000)
    mine, _, _ = orc.trace(osc, rays, seed=SEED, mode=oracle.AABB_TIGHT)
    ref, _, _ = orc.trace(osc, rays, seed=SEED, mode=oracle.AABB_REFERENCE)
    same_hits(mine, ref, "mesh, %s tree" % tree)
    orc.api.scene_free(osc)


@pytest.mark.parametrize("case", range(6))
def test_random_scenes_all_primitive_kinds(orc, rtlib, case):
    # the scenes and rays of tests/test_gpu_trace.py::test_random_scenes_all_primitive_kinds, and as many rays again:
    # wrappers around several primitives (culling boxes), media of all three boundary kinds, eight and more boxes
    rng = np.random.default_rng(1234 + case)
    _, (hroot, oroot) = random_scene_pair(rng, rtlib, orc, n_obj=int(rng.integers(4, 40)))
    hs = lib.HostScene(rtlib, hroot)
    rays = np.concatenate([random_rays(rng, 20000), random_rays(rng, 20000)])
    mine, _, _ = orc.trace(oroot, rays, seed=77, mode=oracle_mode(hs))
    ref, _, _ = orc.trace(oroot, rays, seed=77, mode=oracle.AABB_REFERENCE)
    assert (ref["prim_id"] != abi.RT_MISS).sum() > 400
    same_hits(mine, ref, "random scene %d" % case)
    hs.close()
    orc.api.scene_free(oroot)


@pytest.mark.parametrize("which", ["clipping", "nested"])
def test_scenes_whose_boxes_do_not_bound(orc, rtlib, which):
    # the two synthetic per-axis scenes of tests/test_gpu_trace.py: covering boxes and the f32 planes must not change a
    # hit but for rays within one f32 rounding of a face
    rng = np.random.default_rng(99 if which == "clipping" else 2024)
    if which == "clipping":
        hroot, oroot = clipping_scene_pair(rng, rtlib, orc)
        hs = lib.HostScene(rtlib, hroot)
        assert hs.desc.flags & abi.RT_DESC_BOXES_MAY_NOT_BOUND
        hs.close()
    else:
        _, oroot = nested_cover_scene_pair(rng, rtlib, orc, oracle_only=True)
    rays = random_rays(rng, 30000)
    mine, _, _ = orc.trace(oroot, rays, seed=5, mode=oracle.AABB_PER_AXIS_F32)
    ref, _, _ = orc.trace(oroot, rays, seed=5, mode=oracle.AABB_REFERENCE)
    differ = (mine["prim_id"] != ref["prim_id"]) | (mine["t"].view(np.uint64) != ref["t"].view(np.uint64))
    assert differ.mean() <= 1e-4, differ.sum()
    orc.api.scene_free(oroot)


@pytest.mark.parametrize("sid,size", [(7, (600, 600)), (8, (600, 600)), (9, (800, 800))])
def test_full_size_pictures_do_not_depend_on_the_mode(orc, sid, size):
    # two samples of every pixel at the BASELINE image sizes: all bounces of every path, bit for bit
    W, H = size
    osc = orc.api.scene_build(sid, seed=SEED, **scene_kwargs(sid))
    su = orc.api.scene_setup(sid)
    cam = orc.api.camera(su)
    p = orc.make_params(W, H, 0, 2, list(su.background), seed=SEED)
    a, sa, _ = orc.render(osc, cam, p, mode=oracle.AABB_TIGHT, threads=8)
    b, sb, _ = orc.render(osc, cam, p, mode=oracle.AABB_REFERENCE, threads=8)
    assert np.array_equal(a, b) and sa["rays"] == sb["rays"]
    orc.api.scene_free(osc)


def test_outer_tree_over_the_motion_around_an_inner_tree_that_is_not(orc, rtlib):
    # A node's box is the union of its children's STORED boxes (src/bvh.rs:80-88, :113-116): an outer tree built over the
    # whole motion (0, 1) around an inner tree built over (0, 0.1) does not hold the inner tree's spheres either, although
    # its own interval covers their motion. Such outer nodes must keep the reference's per-axis test (behind a covering
    # box), and a Translate / RotateY around moving content gets no culling box: the box it presented to BVHNode::new is
    # one the reference never tests.
    rng = np.random.default_rng(7)
    hroot, oroot = nested_interval_scene_pair(rng, rtlib, orc)
    hs = lib.HostScene(rtlib, hroot)
    assert hs.desc.flags & abi.RT_DESC_BOXES_MAY_NOT_BOUND
    nodes = hs.table("nodes")
    kind, mark = nodes["meta"] & 15, nodes["meta"] >> 4
    boxes = np.flatnonzero(kind == abi.RT_N_INNER)
    assert mark[boxes[0]] in (0, abi.RT_INNER_COVER)          # the world's root does not hold its content
    assert {0, abi.RT_INNER_CARRIED, abi.RT_INNER_COVER} <= set(mark[boxes].tolist())
    # box entries = the nodes of the three trees (src/bvh.rs:17-91: one node per span of 1 or 2, else one plus the halves)
    # plus the covering boxes, and nothing else: no culling box in front of the two wrappers
    def n_nodes(n):
        return 1 if n <= 2 else 1 + n_nodes(n // 2) + n_nodes(n - n // 2)
    assert len(boxes) == n_nodes(29) + n_nodes(20) + n_nodes(10) + int((mark[boxes] == abi.RT_INNER_COVER).sum())
    rays = np.concatenate([aimed_rays(rng, 60000), random_rays(rng, 20000)])
    mine, _, st = orc.trace(oroot, rays, seed=5, mode=oracle.AABB_PER_AXIS_F32)
    ref, _, rst = orc.trace(oroot, rays, seed=5, mode=oracle.AABB_REFERENCE)
    assert (ref["prim_id"] != abi.RT_MISS).sum() > 8000
    differ = (mine["prim_id"] != ref["prim_id"]) | (mine["t"].view(np.uint64) != ref["t"].view(np.uint64))
    assert differ.sum() <= 2, differ.sum()  # (12 of the first 60 000 when the outer nodes were marked as holding)
    hs.close()
    orc.api.scene_free(oroot)


def test_media_asked_before_the_walk(orc, rtlib):
    # The product takes world-space media bounded by one sphere out of its entry stream and asks them before the walk; the
    # oracle's mirrored modes do the same (World::hit). Against the reference's order, in which such a medium is asked
    # where it stands in the tree: the same closest hit for rays of every kind of range (tmin below zero, finite tmax).
    hroot, oroot = media_scene_pair(rtlib, orc)
    hs = lib.HostScene(rtlib, hroot)
    rays = random_rays(np.random.default_rng(99), 40000)
    mine, _, st = orc.trace(oroot, rays, seed=31, mode=oracle_mode(hs))
    ref, _, rst = orc.trace(oroot, rays, seed=31, mode=oracle.AABB_REFERENCE)
    same_hits(mine, ref, "media scene")
    assert st["medium"] > rst["medium"]  # the two media asked first are asked by every ray
    hit_ids = set(int(x) for x in mine["prim_id"] if x != abi.RT_MISS)
    assert len(hit_ids) >= 10
    hs.close()
    orc.api.scene_free(oroot)
