"""rt_bvh_build on the MI355X against the oracle's CPU statement of the same tree: tokens bit for bit, and scenes built
with it traced against the oracle's scenes built the same way (hits and traversal work). The reference has no such
builder (it stands in for BVHNode::new, src/bvh.rs:17-98); what is checked is the GPU against the definition."""
import numpy as np
import pytest

import oracle
from conftest import SEED, load_golden, scene_kwargs, oracle_mode
from raytracer_amd import abi, lib
from test_bvh_tokens import box_sets
from test_gpu_trace import compare_hits

pytestmark = pytest.mark.gpu


def same_tokens(a, b):
    return a.tobytes() == b.tobytes()


@pytest.mark.parametrize("name,boxes", list(box_sets()), ids=[n for n, _ in box_sets()])
def test_tokens_equal_the_oracles_small(gpu, orc, name, boxes):
    got, ms = gpu.bvh_build(boxes)
    assert same_tokens(got, orc.lbvh_tokens(boxes)), name
    assert ms > 0.0


@pytest.mark.parametrize("n", [4, 63, 64, 65, 127, 128, 129, 191, 192, 4095, 4096, 4097, 8193, 12345])
def test_tokens_equal_the_oracles_at_block_boundaries(gpu, orc, n):
    # node boxes come from unions over 64-leaf blocks and a table over whole blocks: sizes around the block size, with
    # clustered centres so that ranges of every length occur
    rng = np.random.default_rng(n)
    c = rng.normal(0.0, 1.0, (n, 3)) + rng.integers(0, 3, (n, 1)) * 10.0
    h = rng.uniform(0.0, 0.2, (n, 3))
    boxes = np.concatenate([c - h, c + h], axis=1)
    got, _ = gpu.bvh_build(boxes)
    assert same_tokens(got, orc.lbvh_tokens(boxes))


@pytest.mark.parametrize("n", [100_000, 2_000_000])
def test_tokens_equal_the_oracles_large(gpu, orc, n):
    rng = np.random.default_rng(n)
    c = rng.normal(0.0, 50.0, (n, 3))
    c[: n // 10] = c[n // 10: 2 * (n // 10)]  # a tenth of the centres twice
    h = rng.uniform(0.0, 0.5, (n, 3))
    h[::7, 1] = 0.0  # flat boxes among them
    boxes = np.concatenate([c - h, c + h], axis=1)
    got, ms = gpu.bvh_build(boxes)
    want = orc.lbvh_tokens(boxes)
    assert same_tokens(got, want)
    leaves = got["link"][got["meta"] == abi.RT_BVH_TOKEN_LEAF]
    assert np.array_equal(np.sort(leaves), np.arange(n, dtype=np.uint32))
    print("rt_bvh_build: %d objects in %.2f ms on the device" % (n, ms))


def test_bad_boxes_are_refused(gpu):
    boxes = np.array([[0, 0, 0, 1, 1, 1], [2, 2, 2, 3, 3, 3], [4, 4, 4, 5, 5, 5]], dtype=np.float64)
    bad = boxes.copy()
    bad[1, 2] = np.nan
    with pytest.raises(lib.RtError, match="NaN"):
        gpu.bvh_build(bad)
    bad = boxes.copy()
    bad[2, 3] = np.inf
    with pytest.raises(lib.RtError, match="finite centre"):
        gpu.bvh_build(bad)
    got, _ = gpu.bvh_build(boxes)  # and the library works afterwards
    assert (got["meta"] == abi.RT_BVH_TOKEN_LEAF).sum() == 3


@pytest.mark.parametrize("sid", [1, 8, 9, 12])
def test_scene_with_device_built_trees(gpu, orc, sid):
    g = load_golden(sid)
    kw = scene_kwargs(sid)
    flags = abi.RT_SCENE_BVH_LBVH | (abi.RT_SCENE_BVH_SHUTTER_TIMES if sid == 1 else 0)
    hs = lib.HostScene.build(sid, seed=SEED, flags=flags, **kw)
    ds = lib.DeviceScene(hs)
    osc = orc.api.scene_build(sid, seed=SEED, flags=flags, **kw)
    oh, _, ost = orc.trace(osc, g["rays"], seed=SEED, mode=oracle_mode(hs))
    gh, work = ds.trace_work(g["rays"], seed=SEED)
    compare_hits(oh, gh, "scene %d, trees built on the device" % sid)
    # the same tree on both sides: exactly the same traversal work
    assert work["inner"] == ost["box_tests"] and work["triangle"] == ost["triangle"] and work["box6"] == ost["box6"]
    assert work["sphere"] == ost["sphere"] and work["msphere"] == ost["msphere"]
    ds.close()
    hs.close()
    orc.api.scene_free(osc)


def test_mesh_of_a_million_triangles(gpu, orc):
    kw = dict(n_triangles=1_000_000)
    hs = lib.HostScene.build(12, seed=SEED, flags=abi.RT_SCENE_BVH_LBVH, **kw)
    ds = lib.DeviceScene(hs)
    osc = orc.api.scene_build(12, seed=SEED, flags=abi.RT_SCENE_BVH_LBVH, **kw)
    su = orc.api.scene_setup(12)
    cam = orc.api.camera(su)
    p = orc.make_params(128, 128, 0, 2, list(su.background), seed=SEED)
    rays = orc.harvest(osc, cam, p, n_paths=20000, max_rays=1 << 15, max_bounce=3)
    oh, _, ost = orc.trace(osc, rays, seed=SEED, mode=oracle_mode(hs))
    gh, work = ds.trace_work(rays, seed=SEED)
    compare_hits(oh, gh, "1M-triangle mesh, tree built on the device")
    assert work["inner"] == ost["box_tests"] and work["triangle"] == ost["triangle"]
    ds.close()
    hs.close()
    orc.api.scene_free(osc)


def test_builder_call(gpu, orc):
    def make(api):
        b = api.builder(seed=3)
        m = b.lambertian((0.5, 0.5, 0.5))
        objs = [b.sphere((float(i % 7), float(i // 7), 0.1 * i), 0.3, m) for i in range(40)]
        objs.append(b.translate(b.rotate_y(b.box6((0, 0, 0), (1, 2, 1), m), 20.0), (3, 3, -4)))
        return b, b.finish(b.bvh_lbvh(b.list(objs), 0.0, 1.0))

    rays = np.zeros(64, dtype=abi.RAY_DTYPE)
    rng = np.random.default_rng(5)
    rays["o"] = rng.uniform(0, 6, (64, 3)) * (1, 1, 0) + (0, 0, -10)
    rays["d"] = (0, 0, 1)
    rays["tmin"], rays["tmax"] = 1e-4, np.inf
    ob, osc = make(orc.api)
    oh, _, _ = orc.trace(osc, rays, seed=3)
    gb, handle = make(gpu.host)
    hs = lib.HostScene(gpu, handle)
    ds = lib.DeviceScene(hs)
    gh = ds.trace(rays, seed=3)
    compare_hits(oh, gh[0] if isinstance(gh, tuple) else gh, "builder scene")
    assert (oh["prim_id"] != abi.RT_MISS).sum() > 5
    ds.close()
    hs.close()
    orc.api.scene_free(osc)


def build_on_device(gpu, boxes, entry="rt_bvh_build_device", stream=None):
    """Calls a device-resident tree build through the C ABI with torch-owned device buffers."""
    import ctypes as C
    import torch
    n = boxes.shape[0]
    d_boxes = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float64)).cuda()
    d_tokens = torch.zeros((2 * n - 1) * abi.NODE_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    d_flag = torch.full((1,), 7, dtype=torch.int32, device="cuda")
    fn = getattr(gpu.lib, entry)
    args = [d_boxes.data_ptr(), n, d_tokens.data_ptr(), d_flag.data_ptr(), stream.cuda_stream if stream is not None else None]
    if entry == "rt_bvh_build_device":
        ms = C.c_double(0.0)
        args.append(C.byref(ms))
    gpu.check(fn(*args))
    if stream is not None:
        stream.synchronize()
    else:
        torch.cuda.synchronize()
    return d_tokens.cpu().numpy().view(abi.NODE_DTYPE), int(d_flag.item())


@pytest.mark.parametrize("n", [2, 3, 64, 4097, 250_000])
def test_device_resident_build_equals_the_host_buffer_build(gpu, orc, n):
    import torch
    rng = np.random.default_rng(n)
    c = rng.normal(0.0, 5.0, (n, 3))
    h = rng.uniform(0.0, 0.3, (n, 3))
    boxes = np.concatenate([c - h, c + h], axis=1)
    want, _ = gpu.bvh_build(boxes)
    got, flag = build_on_device(gpu, boxes)
    assert same_tokens(got, want) and flag == 0
    assert same_tokens(got, orc.lbvh_tokens(boxes))
    s = torch.cuda.Stream()
    got2, flag2 = build_on_device(gpu, boxes, stream=s)
    assert same_tokens(got2, want) and flag2 == 0
    if n >= 3:
        bad = boxes.copy()
        bad[1, 4] = np.nan
        _, flag3 = build_on_device(gpu, bad)
        assert flag3 == 1


def f32_outward(lo, hi):
    """f64 box -> f32 box rounded outward, as the builders store it."""
    l = lo.astype(np.float32)
    l = np.where(l.astype(np.float64) > lo, np.nextafter(l, np.float32(-np.inf)), l)
    h = hi.astype(np.float32)
    h = np.where(h.astype(np.float64) < hi, np.nextafter(h, np.float32(np.inf)), h)
    return l, h


def object_entries(n):
    """Entries of n made-up objects as a flattener would emit them one by one: a primitive (1 entry), a wrapped
    primitive (3), a medium around a sphere (2, the medium entry carrying a link)."""
    entries, offsets = [], [0]
    for i in range(n):
        at = len(entries)
        if i % 3 == 0:
            entries.append((5 | (i << 4), 0))
        elif i % 3 == 1:
            entries += [(6 | (i << 4), 0), (4 | (i << 4), 0), (7 | (i << 4), 0)]
        else:
            entries += [(8 | (i << 4), at + 2), (1 | (i << 4), 0)]
        offsets.append(len(entries))
    arr = np.zeros(len(entries), dtype=abi.NODE_DTYPE)
    arr["meta"] = [e[0] for e in entries]
    arr["link"] = [e[1] for e in entries]
    return arr, np.asarray(offsets, dtype=np.uint32)


@pytest.mark.parametrize("n", [2, 3, 700, 20_000])
def test_tokens_expanded_on_the_device(gpu, n):
    # rt_bvh_build_stream against the expansion a host would do from the tokens of rt_bvh_build
    import ctypes as C
    rng = np.random.default_rng(n)
    c = rng.normal(0.0, 4.0, (n, 3))
    boxes = np.concatenate([c - 0.2, c + 0.2], axis=1)
    tokens, _ = gpu.bvh_build(boxes)
    entries, offsets = object_entries(n)
    want = []
    where = np.zeros(len(tokens) + 1, dtype=np.int64)  # token -> position of its first entry
    for k, t in enumerate(tokens):
        where[k] = len(want)
        if t["meta"] == abi.RT_BVH_TOKEN_LEAF:
            o = int(t["link"])
            for e in entries[offsets[o]:offsets[o + 1]]:
                e = e.copy()
                if (e["meta"] & 15) in (0, 8):
                    e["link"] = int(e["link"]) - int(offsets[o]) + len(want) - (len(want) - where[k])
                want.append(e)
        else:
            want.append(t.copy())
    where[len(tokens)] = len(want)
    want = np.array(want, dtype=abi.NODE_DTYPE)
    inner = (want["meta"] & 15) == 0
    want["link"][inner] = where[want["link"][inner]]
    got = np.zeros(n - 1 + len(entries), dtype=abi.NODE_DTYPE)
    gpu.check(gpu.lib.rt_bvh_build_stream(boxes.ctypes.data, n, entries.ctypes.data, offsets.ctypes.data, got.ctypes.data, None))
    assert got.tobytes() == want.tobytes()
    wrong = np.zeros(5, dtype=abi.NODE_DTYPE)
    assert gpu.lib.rt_bvh_build_stream(boxes.ctypes.data, n, entries.ctypes.data, None, wrong.ctypes.data, None) == abi.RT_E_INVALID


@pytest.mark.parametrize("n", [2, 65, 3000])
def test_refit_gives_every_node_the_box_of_its_leaves(gpu, n):
    import torch
    rng = np.random.default_rng(n + 1)
    c = rng.normal(0.0, 4.0, (n, 3))
    h = rng.uniform(0.05, 0.3, (n, 3))
    a = np.concatenate([c - h, c + h], axis=1)
    tokens, _ = gpu.bvh_build(a)
    moved = a.copy()
    shift = rng.normal(0.0, 0.5, (n, 3))
    moved[:, :3] += shift
    moved[:, 3:] += shift
    moved[::11, 4] = moved[::11, 1]  # some flat boxes

    def refit(boxes):
        d_tok = torch.from_numpy(tokens.view(np.uint8).copy()).cuda()
        d_box = torch.from_numpy(boxes).cuda()
        flag = torch.full((1,), 9, dtype=torch.int32, device="cuda")
        gpu.check(gpu.lib.rt_bvh_refit_device(d_tok.data_ptr(), n, d_box.data_ptr(), flag.data_ptr(), None))
        torch.cuda.synchronize()
        return d_tok.cpu().numpy().view(abi.NODE_DTYPE), int(flag.item())

    same, flag = refit(a)
    assert same.tobytes() == tokens.tobytes() and flag == 0  # the boxes it was built over: nothing changes
    got, flag = refit(moved)
    assert flag == 0
    assert np.array_equal(got["meta"], tokens["meta"]) and np.array_equal(got["link"], tokens["link"])
    leaf = tokens["meta"] == abi.RT_BVH_TOKEN_LEAF
    for p in np.nonzero(~leaf)[0]:
        end = int(tokens["link"][p])
        objs = tokens["link"][p + 1:end][leaf[p + 1:end]]
        lo, hi = moved[objs, :3].min(axis=0), moved[objs, 3:].max(axis=0)
        if (lo == hi).any():
            assert np.isposinf(got["lo"][p]).all() and np.isneginf(got["hi"][p]).all()
        else:
            l, hh = f32_outward(lo, hi)
            assert np.array_equal(got["lo"][p], l) and np.array_equal(got["hi"][p], hh), p
    bad = moved.copy()
    bad[0, 0] = np.nan
    _, flag = refit(bad)
    assert flag == 1
