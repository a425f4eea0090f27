"""The linear BVH (the tree that rt_bvh_build makes on the device instead of BVHNode::new, src/bvh.rs:17-98): the
oracle's CPU statement of it against a brute-force reading of its definition, the oracle's scenes built with it against
the reference-order tree, and what the product library does without a GPU. The reference has no such builder: these
tests pin the definition, not the reference."""
import ctypes as C

import numpy as np
import pytest

import oracle
from conftest import SEED, load_golden, scene_kwargs
from raytracer_amd import abi, lib
from raytracer_amd.hostapi import HostApiError


def brute_force_tokens(boxes):
    """Pre-order (kind, link) pairs straight from the definition: sort by (Morton code, input position), split every
    range at the highest bit in which its first and last 96-bit keys (code, sorted position) differ."""
    n = len(boxes)
    cen = (boxes[:, :3] + boxes[:, 3:]) * 0.5
    lo, hi = cen.min(0), cen.max(0)
    ext = hi - lo
    q = np.zeros((n, 3), dtype=np.uint64)
    for a in range(3):
        if ext[a] > 0:
            q[:, a] = np.minimum((cen[:, a] - lo[a]) / ext[a] * 2097152.0, 2097151.0).astype(np.uint64)

    def spread(v):
        r = 0
        for i in range(21):
            r |= ((int(v) >> i) & 1) << (3 * i)
        return r

    codes = [(spread(x) << 2) | (spread(y) << 1) | spread(z) for x, y, z in q]
    order = sorted(range(n), key=lambda i: (codes[i], i))
    keys = [(codes[i] << 32) | k for k, i in enumerate(order)]
    out = []
    stack = [("range", 0, n - 1)]
    while stack:  # pre-order without recursion
        item = stack.pop()
        if item[0] == "close":
            out[item[1]] = (abi.RT_N_INNER, len(out))
            continue
        _, f, l = item
        if f == l:
            out.append((abi.RT_BVH_TOKEN_LEAF, order[f]))
            continue
        top = (keys[f] ^ keys[l]).bit_length() - 1
        s = f
        while ((keys[s + 1] >> top) & 1) == ((keys[f] >> top) & 1):
            s += 1
        at = len(out)
        out.append(None)
        stack.append(("close", at))
        stack.append(("range", s + 1, l))
        stack.append(("range", f, s))
    return out


def box_sets():
    rng = np.random.default_rng(7)

    def around(c, h):
        return np.concatenate([c - h, c + h], axis=1)

    c = rng.uniform(-5, 5, (700, 3))
    yield "scattered", around(c, rng.uniform(0, 0.3, (700, 3)))
    yield "two", around(c[:2], np.full((2, 3), 0.1))
    yield "three", around(c[:3], np.full((3, 3), 0.1))
    same = np.repeat(c[:40], 8, axis=0)  # eight objects per centre: equal codes, order decided by position
    yield "ties", around(same, rng.uniform(0, 0.3, same.shape))
    yield "all the same", around(np.zeros((33, 3)), np.full((33, 3), 1.0))
    flat = c.copy()
    flat[:, 1] = 2.5  # no extent on y
    yield "flat axis", around(flat, np.zeros_like(flat))
    grid = np.stack(np.meshgrid(np.arange(8.0), np.arange(8.0), np.arange(8.0)), -1).reshape(-1, 3)
    yield "grid", around(grid, np.full(grid.shape, 0.5))


@pytest.mark.parametrize("name,boxes", list(box_sets()), ids=[n for n, _ in box_sets()])
def test_oracle_tokens_follow_the_definition(orc, name, boxes):
    t = orc.lbvh_tokens(boxes)
    want = brute_force_tokens(boxes)
    assert len(t) == 2 * len(boxes) - 1 == len(want)
    assert [(int(m), int(l)) for m, l in zip(t["meta"], t["link"])] == want
    # every inner box is the union of the leaf boxes under it, rounded outward (or inverted when flat on an axis)
    for k in np.nonzero(t["meta"] == abi.RT_N_INNER)[0][:50]:
        under = t[k + 1:t[k]["link"]]
        objs = under["link"][under["meta"] == abi.RT_BVH_TOKEN_LEAF]
        mn, mx = boxes[objs, :3].min(0), boxes[objs, 3:].max(0)
        if np.any(mn == mx):
            assert np.all(np.isposinf(t[k]["lo"])) and np.all(np.isneginf(t[k]["hi"]))
        else:
            assert np.all(t[k]["lo"].astype(np.float64) <= mn) and np.all(t[k]["hi"].astype(np.float64) >= mx)
            assert np.all(np.nextafter(t[k]["lo"], np.float32(np.inf)).astype(np.float64) > mn)
            assert np.all(np.nextafter(t[k]["hi"], np.float32(-np.inf)).astype(np.float64) < mx)


def test_oracle_refuses_what_the_device_builder_refuses(orc):
    ok = np.array([[0, 0, 0, 1, 1, 1], [2, 2, 2, 3, 3, 3]], dtype=np.float64)
    with pytest.raises(HostApiError):
        orc.lbvh_tokens(ok[:1])
    bad = ok.copy()
    bad[1, 4] = np.nan
    with pytest.raises(HostApiError):
        orc.lbvh_tokens(bad)
    bad = ok.copy()
    bad[0, :3] = -np.inf
    bad[0, 3:] = np.inf  # centre = NaN
    with pytest.raises(HostApiError):
        orc.lbvh_tokens(bad)


@pytest.mark.parametrize("sid", [1, 8, 9, 12])
def test_linear_tree_gives_the_same_hits(orc, sid):
    # the closest hit does not depend on the tree, except where two primitives are hit at exactly the same t
    g = load_golden(sid)
    kw = scene_kwargs(sid)
    flags = abi.RT_SCENE_BVH_LBVH | (abi.RT_SCENE_BVH_SHUTTER_TIMES if sid == 1 else 0)
    osc = orc.api.scene_build(sid, seed=SEED, flags=flags, **kw)
    oh, _, st = orc.trace(osc, g["rays"], seed=SEED, mode=oracle.AABB_TIGHT)
    orc.api.scene_free(osc)
    if sid == 1:  # the golden hits of scene 1 come from boxes over (0, 0.1); compare with a shutter-time reference tree
        ref = orc.api.scene_build(sid, seed=SEED, flags=abi.RT_SCENE_BVH_SHUTTER_TIMES, **kw)
        want, _, _ = orc.trace(ref, g["rays"], seed=SEED, mode=oracle.AABB_TIGHT)
        orc.api.scene_free(ref)
    else:
        want = g["hits"]
    same_t = oh["t"].view(np.uint64) == want["t"].view(np.uint64)
    assert same_t.all(), "scene %d: %d rays with another t" % (sid, (~same_t).sum())
    other = oh["prim_id"] != want["prim_id"]
    assert other.mean() < 0.02  # exact ties (adjacent ground boxes, overlapping mirror boxes) may pick the other object
    assert st["box_tests"] > 0


def test_builder_call_for_a_linear_tree(orc):
    b = orc.api.builder(seed=3)
    m = b.lambertian((0.5, 0.5, 0.5))
    objs = [b.sphere((float(i % 7), float(i // 7), 0.1 * i), 0.3, m) for i in range(40)]
    root = b.bvh_lbvh(b.list(objs), 0.0, 1.0)
    sc = b.finish(root)
    rays = np.zeros(40, dtype=abi.RAY_DTYPE)
    for i in range(40):
        rays[i]["o"] = (i % 7, i // 7, -10.0)
        rays[i]["d"] = (0, 0, 1)
    rays["tmin"], rays["tmax"] = 1e-4, np.inf
    h, _, _ = orc.trace(sc, rays, seed=3)
    assert list(h["prim_id"]) == list(range(40))
    orc.api.scene_free(sc)
    b2 = orc.api.builder(seed=3)
    with pytest.raises(HostApiError):
        b2.bvh_lbvh(b2.list([b2.sphere((0, 0, 0), 1.0, b2.lambertian((1, 1, 1)))]), 0.0, 1.0)


def test_product_checks_arguments_and_has_no_cpu_path(rtlib):
    L = rtlib.lib
    boxes = np.array([[0, 0, 0, 1, 1, 1], [2, 2, 2, 3, 3, 3]], dtype=np.float64)
    tokens = np.zeros(3, dtype=abi.NODE_DTYPE)
    assert L.rt_bvh_build(None, 2, tokens.ctypes.data, None) == abi.RT_E_INVALID
    assert L.rt_bvh_build(boxes.ctypes.data, 2, None, None) == abi.RT_E_INVALID
    assert L.rt_bvh_build(boxes.ctypes.data, 1, tokens.ctypes.data, None) == abi.RT_E_INVALID
    assert L.rt_bvh_build(boxes.ctypes.data, abi.RT_BVH_MAX_OBJECTS + 1, tokens.ctypes.data, None) == abi.RT_E_INVALID
    if rtlib.device_count() >= 1:
        pytest.skip("a GPU is present: the device path is covered by the gpu tests")
    assert L.rt_bvh_build(boxes.ctypes.data, 2, tokens.ctypes.data, None) == abi.RT_E_DEVICE
    assert b"no CPU path" in L.rt_last_error()
    with pytest.raises(HostApiError, match="no CPU path"):
        lib.HostScene.build(7, seed=SEED, flags=abi.RT_SCENE_BVH_LBVH)
