"""Host logic: the scene-construction API, BVHNode::new and the flattener (no GPU)."""
import ctypes as C

import numpy as np
import pytest

from conftest import SCENE_OPTS, SEED, scene_kwargs
from raytracer_amd import abi, lib
from raytracer_amd.hostapi import HostApiError

K = {v: k for k, v in abi.NODE_KINDS.items()}


def walk_subtree_end(nodes, i):
    """Index one past the subtree rooted at entry i, derived from the structure alone."""
    kind = int(nodes["meta"][i]) & 15
    if kind == K["inner"] or kind == K["medium"]:
        return int(nodes["link"][i])
    if kind == K["xf_enter"]:
        depth = 1
        j = i + 1
        while depth:
            k = int(nodes["meta"][j]) & 15
            depth += (k == K["xf_enter"]) - (k == K["xf_exit"])
            j += 1
        return j
    return i + 1


@pytest.mark.parametrize("sid", sorted(SCENE_OPTS))
def test_flatten_structure(rtlib, orc, sid):
    hs = lib.HostScene.build(sid, seed=SEED, **scene_kwargs(sid))
    nodes = hs.table("nodes")
    n = len(nodes)
    kinds = nodes["meta"] & 15
    idx = nodes["meta"] >> 4
    # the world is a BVHNode; a node whose box contains the world's (the root first of all) is passed by every ray from
    # inside the scene and gets no entry, so the stream starts with what is below it; scene 1's root does not hold its
    # spheres and keeps its (covered) entry
    if hs.desc.flags & abi.RT_DESC_BOXES_MAY_NOT_BOUND:
        assert kinds[0] == K["inner"] and nodes["link"][0] == n
    # every inner node: 1 or 2 children exactly filling [i+1, link); its f32 box contains inner children's boxes.
    # A box entry directly in front of a wrapper is the culling box the flattener adds for a Translate / RotateY that is
    # a leaf of a tree, or for a box of six sides that is one (not in the reference): one child, padded slightly beyond
    # the child's own box.
    # In a scene whose boxes may not hold their contents (scene 1: tree over (0, 0.1), spheres moving over (0, 1)) a node
    # whose box may not gets a covering box in front of it (marked RT_INNER_COVER, one child: the node, unmarked), over
    # the whole motion of what is below; nodes whose box does hold are marked themselves.
    def plain_cull(i):  # a box entry with exactly one child that is a wrapper or a box of six sides
        return kinds[i + 1] in (K["xf_enter"], K["box6"]) and walk_subtree_end(nodes, i + 1) == int(nodes["link"][i])

    def is_cover(i):
        return idx[i] == 2 and kinds[i + 1] == K["inner"] and idx[i + 1] == 0 and nodes["link"][i] == nodes["link"][i + 1]

    def is_cull(i):
        return is_cover(i) or plain_cull(i)

    inner = np.nonzero(kinds == K["inner"])[0]
    assert set(idx[inner].tolist()) <= {0, 1, 2}
    n_cover = sum(1 for i in inner if i + 1 < n and is_cover(i))
    assert n_cover == int((idx[inner] == 2).sum())
    if hs.desc.flags & abi.RT_DESC_BOXES_MAY_NOT_BOUND:
        assert n_cover > 0
    else:
        assert n_cover == 0  # every box holds its content: no covering boxes

    for i in np.nonzero(kinds == K["inner"])[0]:
        end = int(nodes["link"][i])
        assert i < end <= n
        if is_cull(i):
            continue
        c1 = walk_subtree_end(nodes, i + 1)
        assert c1 <= end
        if c1 < end:
            assert walk_subtree_end(nodes, c1) == end
        for c in (i + 1, c1):
            cull = c < end and kinds[c] == K["inner"] and is_cull(c)
            if c < end and kinds[c] == K["inner"] and not cull and np.isfinite(nodes["lo"][c]).all() and np.isfinite(nodes["lo"][i]).all():
                assert (nodes["lo"][i] <= nodes["lo"][c]).all() and (nodes["hi"][i] >= nodes["hi"][c]).all()
    # table indices in range, one table entry per distinct object, creation-order ids form a permutation
    ids = []
    for kind, tab in (("sphere", "spheres"), ("msphere", "msphere"), ("rect", "rects"), ("box6", "boxes"),
                      ("triangle", "triangles"), ("medium", "media")):
        t = hs.table(tab)
        sel = kinds == K[kind]
        if sel.any():
            assert idx[sel].max() < len(t)
            assert set(idx[sel].tolist()) == set(range(len(t)))
            ids += t["prim_id"].tolist()
    osc = orc.api.scene_build(sid, seed=SEED, **scene_kwargs(sid))
    assert sorted(ids) == list(range(orc.lib.orc_scene_prim_count(osc)))
    orc.api.scene_free(osc)
    # wrappers: enter/exit balanced, depth and ancestors consistent
    xf = hs.table("xforms")
    stack = []
    for i in range(n):
        if kinds[i] == K["xf_enter"]:
            x = xf[idx[i]]
            assert x["depth"] == len(stack) + 1 and list(x["anc"][: len(stack)]) == stack and x["anc"][len(stack)] == idx[i]
            stack.append(int(idx[i]))
        elif kinds[i] == K["xf_exit"]:
            assert stack.pop() == idx[i]
    assert not stack
    hs.close()


def test_cornell_box_layout(rtlib):
    # src/main.rs:459-517: 6 rects, two Translate(RotateY(Box6)), one world BVH
    hs = lib.HostScene.build(7)
    d = hs.desc
    assert (d.n_rects, d.n_boxes, d.n_xforms, d.n_media, d.n_spheres) == (6, 2, 4, 0, 0)
    rects = hs.table("rects")
    light = rects[rects["prim_id"] == 2][0]
    assert (light["a0"], light["a1"], light["b0"], light["b1"], light["k"], light["axis"]) == (213.0, 343.0, 227.0, 332.0, 554.0, 1)
    mats = hs.table("materials")
    tex = hs.table("textures")
    lm = mats[light["mat"]]
    assert lm["kind"] == 3 and tex[lm["tex"]]["a"].tolist() == [15.0, 15.0, 15.0]
    xf = hs.table("xforms")
    rot = xf[xf["kind"] == 1]
    assert np.allclose(np.sort(rot["sin_theta"]), np.sort(np.sin(np.radians([15.0, -18.0]))), rtol=0, atol=1e-15)
    hs.close()


def test_final_scene_layout(rtlib):
    # src/main.rs:587-699: 400 boxes, 1 light rect, 1 moving sphere, 105 spheres (the r=70 boundary appears twice in the
    # stream but once in the table), 2 media, 1 noise texture, 1 image texture
    hs = lib.HostScene.build(9, **scene_kwargs(9))
    d = hs.desc
    assert (d.n_boxes, d.n_rects, d.n_msphere, d.n_spheres, d.n_media, d.n_perlins, d.n_images) == (400, 1, 1, 106, 2, 1, 1)
    nodes = hs.table("nodes")
    assert int(((nodes["meta"] & 15) == K["sphere"]).sum()) == 107
    mats = hs.table("materials")
    metal = mats[mats["kind"] == 1][0]
    assert metal["param"] == 1.0  # fuzz 10.0 clamped (src/material.rs:62)
    med = hs.table("media")
    assert sorted(med["neg_inv_density"].tolist()) == [-1.0 / 0.0001, -1.0 / 0.02]
    assert d.n_texel_bytes == 1024 * 512 * 3
    hs.close()


def test_scene_and_camera_constants_match_oracle(rtlib, orc):
    for sid in sorted(SCENE_OPTS):
        a = rtlib.host.scene_setup(sid)
        b = orc.api.scene_setup(sid)
        assert bytes(a) == bytes(b)
        assert bytes(rtlib.host.camera(a)) == bytes(orc.api.camera(b))
    with pytest.raises(HostApiError):
        rtlib.host.scene_setup(99)
    with pytest.raises(HostApiError):
        rtlib.host.scene_build(99)


def test_camera_known_values(rtlib):
    # Camera::new (src/camera.rs:27-59) for the Cornell view: w = -z, u = -x... with vfov 40, aspect 1
    cam = rtlib.host.camera_new([278, 278, -800], [278, 278, 0], [0, 1, 0], 40.0, 1.0, 0.0, 10.0)
    h = np.tan(np.radians(40.0) / 2)
    assert list(cam.u) == [-1.0, 0.0, 0.0] and list(cam.v) == [0.0, 1.0, 0.0]
    assert np.allclose(list(cam.horizontal), [-20 * h, 0, 0]) and np.allclose(list(cam.vertical), [0, 20 * h, 0])
    assert cam.lens_radius == 0.0
    with pytest.raises(HostApiError):  # lookfrom == lookat: unit() of zero (src/vec3.rs:285-286 panics)
        rtlib.host.camera_new([1, 1, 1], [1, 1, 1], [0, 1, 0], 40.0, 1.0, 0.0, 10.0)


def test_same_seed_same_scene_other_seed_other_scene(rtlib):
    a = lib.HostScene.build(1, seed=1)
    b = lib.HostScene.build(1, seed=1)
    c = lib.HostScene.build(1, seed=2)
    assert a.table("nodes").tobytes() == b.table("nodes").tobytes()
    assert a.table("msphere").tobytes() == b.table("msphere").tobytes()
    assert a.table("msphere").tobytes() != c.table("msphere").tobytes()
    for s in (a, b, c):
        s.close()


def test_builder_span_one_and_shared_objects(rtlib):
    b = rtlib.host.builder()
    m = b.metal([0.8, 0.8, 0.8], 0.01)
    box = b.box6([0, 0, 0], [3, 20, 0.1], m)
    tri = b.triangle([2.5, 22, -0.1], [1.5, 20, -0.86], [1.9, 20, -1.5], b.lambertian([1, 0.45, 0.45]))
    objs = [box, b.translate(b.rotate_y(box, 60.0), [0, 0, 0]), tri]
    hs = lib.HostScene(rtlib, b.finish(b.bvh(b.list(objs))))
    nodes = hs.table("nodes")
    kinds = (nodes["meta"] & 15).tolist()
    # 3 objects: the BVH has a span-1 node (left == right, emitted once) and a span-2 node under a root that gets no
    # entry (its box is the world's: every ray from inside passes it); a wrapper around a single primitive gets no
    # culling box
    assert kinds.count(K["inner"]) == 2 and kinds.count(K["box6"]) == 2 and hs.desc.n_boxes == 1
    assert hs.desc.n_xforms == 2
    xf = hs.table("xforms")
    assert xf["kind"].tolist() == [0, 1] and xf["depth"].tolist() == [1, 2] and xf["inst_id"].tolist() == [1, 0]
    hs.close()


def test_culling_box_in_front_of_a_wrapped_tree(rtlib):
    # Translate(RotateY(BVHNode)) as a leaf of the world tree (the sphere cluster of final_scene, src/main.rs:690-696):
    # a box entry holding the wrapper's own bounding box, slightly padded, comes first and skips the whole wrapper
    b = rtlib.host.builder()
    m = b.lambertian([0.7, 0.7, 0.7])
    cluster = b.bvh(b.list([b.sphere([float(i), 0.5 * i, 1.0], 0.4, m) for i in range(5)]))
    wrapped = b.translate(b.rotate_y(cluster, 15.0), [-10.0, 2.0, 3.0])
    hs = lib.HostScene(rtlib, b.finish(b.bvh(b.list([wrapped, b.sphere([30, 0, 0], 1.0, m)]))))
    nodes = hs.table("nodes")
    kinds = (nodes["meta"] & 15).tolist()
    at = kinds.index(K["xf_enter"])
    assert kinds[at - 1] == K["inner"] and kinds[at + 1] == K["xf_enter"]
    end = int(nodes["link"][at - 1])
    assert kinds[end - 1] == K["xf_exit"] and kinds[end - 2] == K["xf_exit"] and kinds[end] == K["sphere"]
    root_lo, root_hi = nodes["lo"][0], nodes["hi"][0]
    lo, hi = nodes["lo"][at - 1], nodes["hi"][at - 1]
    assert (lo < hi).all() and (lo[0] < root_lo[0] or np.isclose(lo[0], root_lo[0], rtol=1e-5))  # the padded box may reach past the root's
    assert hi[0] < 0 and lo[0] > -20  # around the translated cluster, not around the far sphere
    hs.close()


def test_builder_errors(rtlib):
    b = rtlib.host.builder()
    with pytest.raises(HostApiError):
        b.sphere([0, 0, 0], 1.0, 5)  # bad material handle
    with pytest.raises(HostApiError):
        b.bvh(b.list())  # BVHNode::new on an empty list (the reference would recurse forever / index out of bounds)
    m = b.lambertian([0.5, 0.5, 0.5])
    s = b.sphere([0, 0, 0], 1.0, m)
    inner = b.constant_medium(s, 0.1, [1, 1, 1])
    outer = b.constant_medium(inner, 0.1, [1, 1, 1])
    with pytest.raises(HostApiError):
        b.finish(b.bvh(b.list([outer])))  # medium inside a medium boundary is rejected
    o = s
    for _ in range(5):
        o = b.translate(o, [1, 0, 0])
    with pytest.raises(HostApiError):
        b.finish(b.bvh(b.list([o])))  # more than 4 nested wrappers
    with pytest.raises(HostApiError):
        b.finish(999)


def test_flat_box_is_stored_inverted(rtlib):
    # An axis-aligned triangle alone in a span-1 node has a flat box, which the reference's test never passes
    # (src/aabb.rs:25); it is stored as an inverted box so that the f32 rounding cannot give it thickness.
    b = rtlib.host.builder()
    m = b.lambertian([0.5, 0.5, 0.5])
    tri = b.triangle([1, 0, 0.1], [0, 1, 0.1], [0, 0, 0.1], m)
    hs = lib.HostScene(rtlib, b.finish(b.bvh(b.list([tri]))))
    n0 = hs.table("nodes")[0]
    assert np.isposinf(n0["lo"]).all() and np.isneginf(n0["hi"]).all()
    hs.close()


def test_widest_axis_flag_matches_the_oracle_tree(rtlib, orc):
    # both builders apply the same split rule: same entry count, and (for the mesh) a tree the oracle walks with
    # fewer box tests than the random-axis one
    for sid in (9, 12):
        kw = scene_kwargs(sid)
        ref = lib.HostScene.build(sid, seed=SEED, **kw)
        wide = lib.HostScene.build(sid, seed=SEED, flags=abi.RT_SCENE_BVH_WIDEST_AXIS, **kw)
        assert ref.desc.n_nodes == wide.desc.n_nodes and ref.desc.n_triangles == wide.desc.n_triangles
        a, b = ref.table("nodes"), wide.table("nodes")
        assert not np.array_equal(a["meta"], b["meta"])  # a different tree ...
        assert sorted(a["meta"].tolist()) == sorted(b["meta"].tolist())  # ... over the same objects
        ref.close()
        wide.close()


def test_forked_bvh_build_gives_the_sequential_tree(rtlib, monkeypatch):
    # subtrees over 65536 and more objects are built by separate threads; every node's split-axis draw is addressed by
    # its position in the scene stream, so the tree must be the one a single thread builds
    tables = []
    for levels in ("0", "4"):
        monkeypatch.setenv("RT_BVH_FORK_LEVELS", levels)
        hs = lib.HostScene.build(12, seed=SEED, n_triangles=300000)
        tables.append((hs.table("nodes").tobytes(), hs.table("triangles").tobytes()))
        hs.close()
    assert tables[0] == tables[1]
    # and draws made after the build continue where a sequential build leaves the stream
    def two_bvhs():
        b = rtlib.host.builder(seed=11)
        m = b.lambertian([0.5, 0.5, 0.5])
        first = b.bvh(b.list([b.sphere([float(i), 0.0, 0.0], 0.3, m) for i in range(40)]))
        second = b.bvh(b.list([b.sphere([0.0, float(i), 0.0], 0.3, m) for i in range(40)]))
        hs = lib.HostScene(rtlib, b.finish(b.bvh(b.list([first, second]))))
        out = hs.table("nodes").tobytes()
        hs.close()
        return out
    monkeypatch.setenv("RT_BVH_FORK_LEVELS", "0")
    a = two_bvhs()
    monkeypatch.setenv("RT_BVH_FORK_LEVELS", "4")
    assert a == two_bvhs()


def test_mesh_scene_counts(rtlib):
    hs = lib.HostScene.build(12, n_triangles=5000)
    G = int(np.ceil(np.sqrt(5000 / 8.0)))
    assert hs.desc.n_triangles == 1 + 8 * G * G
    assert hs.desc.n_materials == 2 + 64
    tri = hs.table("triangles")
    assert np.isfinite(tri["e1"]).all() and (np.abs(np.cross(tri["e1"], tri["e2"])).sum(1) > 0).all()
    hs.close()
