"""rt_scene_create's check of the scene description (the boundary a foreign host such as the Rust binding hands its
flattened scene over): every description the product's own flattener makes passes it, and a description with one field
out of line is refused with RT_E_INVALID before anything is uploaded, so that no kernel ever indexes outside a table
or walks a stream that does not end. The check runs before the first HIP call, so this runs without a GPU: a
description that passes is then turned down for want of a device (or, on a GPU box, accepted)."""
import ctypes as C

import numpy as np
import pytest

from conftest import SCENE_OPTS, scene_kwargs
from raytracer_amd import abi, lib

M_LAMBERTIAN = 0                       # RtMaterialKind, RtTextureKind (include/rt/abi.h)
T_NOISE, T_IMAGE = 2, 3
TABLES = ("nodes", "spheres", "msphere", "rects", "boxes", "triangles", "xforms", "media", "materials", "textures", "perlins", "images")
COUNTS = {"nodes": "n_nodes", "spheres": "n_spheres", "msphere": "n_msphere", "rects": "n_rects", "boxes": "n_boxes",
          "triangles": "n_triangles", "xforms": "n_xforms", "media": "n_media", "materials": "n_materials",
          "textures": "n_textures", "perlins": "n_perlins", "images": "n_images"}


class Copy:
    """A description whose tables are private copies that a test may change."""

    def __init__(self, hs):
        self.desc = abi.RtSceneDesc()
        C.memmove(C.byref(self.desc), C.byref(hs.desc), C.sizeof(abi.RtSceneDesc))
        self.keep = {}
        for name in TABLES:
            n = getattr(hs.desc, COUNTS[name])
            if n:
                typ = getattr(hs.desc, name)._type_
                arr = (typ * n)()
                C.memmove(arr, getattr(hs.desc, name), C.sizeof(typ) * n)
                self.keep[name] = arr
                setattr(self.desc, name, C.cast(arr, C.POINTER(typ)))

    def __getitem__(self, name):
        return self.keep[name]


def create(rtlib, desc):
    """(return code, message) of rt_scene_create; a scene that was made is destroyed again."""
    h = C.c_void_p()
    rc = rtlib.lib.rt_scene_create(C.byref(desc), C.byref(h))
    msg = rtlib.lib.rt_last_error().decode() if rc else ""
    if rc == 0:
        rtlib.lib.rt_scene_destroy(h)
    return rc, msg


def passes(rc, msg):
    return rc == 0 or (rc == abi.RT_E_DEVICE and "scene description" not in msg)


@pytest.mark.parametrize("sid", sorted(SCENE_OPTS))
def test_every_built_in_scene_passes(rtlib, sid):
    for flags in (0, abi.RT_SCENE_BVH_WIDEST_AXIS):
        hs = lib.HostScene.build(sid, seed=11, flags=flags, **scene_kwargs(sid))
        rc, msg = create(rtlib, hs.desc)
        assert passes(rc, msg), (sid, rc, msg)
        rc, msg = create(rtlib, Copy(hs).desc)  # and so does a copy of it (the helper above)
        assert passes(rc, msg), (sid, rc, msg)
        hs.close()


def test_ad_hoc_scenes_pass(rtlib, orc):
    from scenegen import media_scene_pair, nested_cover_scene_pair, random_scene_pair
    for case in range(20):
        rng = np.random.default_rng(300 + case)
        builders, (hroot, oroot) = random_scene_pair(rng, rtlib, orc, n_obj=int(rng.integers(4, 60)))
        hs = lib.HostScene(rtlib, hroot)
        rc, msg = create(rtlib, hs.desc)
        assert passes(rc, msg), (case, rc, msg)
        hs.close(); orc.api.scene_free(oroot)
    hroot, oroot = media_scene_pair(rtlib, orc)
    hs = lib.HostScene(rtlib, hroot)
    assert passes(*create(rtlib, hs.desc))
    hs.close(); orc.api.scene_free(oroot)


def first(nodes, kind, start=0):
    for i in range(start, len(nodes)):
        if nodes[i].meta & 15 == kind:
            return i
    raise AssertionError("no entry of kind %d" % kind)


def test_one_field_out_of_line_is_refused(rtlib):
    hs = lib.HostScene.build(9, seed=11, **scene_kwargs(9))  # the final scene: every table but triangles is filled

    def refused(change, match):
        c = Copy(hs)
        change(c)
        rc, msg = create(rtlib, c.desc)
        assert rc == abi.RT_E_INVALID and match in msg, (match, rc, msg)

    def field(name, value):
        return lambda c: setattr(c.desc, name, value)

    refused(field("abi_version", abi.RT_ABI_VERSION + 1), "abi_version")
    refused(field("flags", 2), "unknown flag")
    refused(lambda c: setattr(c.desc, "spheres", C.POINTER(abi.RtSphere)()), "null table")
    n = hs.desc.n_nodes
    pristine = Copy(hs)
    base = pristine["nodes"]
    inner = first(base, abi.RT_N_INNER)
    sphere = first(base, abi.RT_N_SPHERE)
    medium = first(base, abi.RT_N_MEDIUM)
    enter = first(base, abi.RT_N_XF_ENTER)
    leave = first(base, abi.RT_N_XF_EXIT)

    def node(i, meta=None, link=None):
        def change(c):
            if meta is not None:
                c["nodes"][i].meta = meta
            if link is not None:
                c["nodes"][i].link = link
        return change

    refused(node(sphere, meta=9), "entry kind")
    refused(node(sphere, meta=abi.RT_N_SPHERE | (hs.desc.n_spheres << 4)), "table index")
    refused(node(inner, meta=abi.RT_N_INNER | (7 << 4)), "unknown marker")
    refused(node(inner, link=inner), "entry link")              # a walk that never advances
    refused(node(inner, link=n + 1), "entry link")              # a walk that leaves the stream
    refused(node(medium, link=medium + 1), "medium without a boundary")
    refused(node(enter, meta=abi.RT_N_SPHERE), "wrapper")
    refused(node(leave, meta=abi.RT_N_SPHERE), "wrapper")
    # a box entry inside the wrapped cluster whose link skips the wrapper's exits: the walk would stay in the cluster's space
    inside = first(base, abi.RT_N_INNER, enter)
    refused(node(inside, link=n), "range")
    # the box entry around the cluster's wrappers, linked into the middle of the cluster: its range would end inside them
    around = [i for i in range(enter) if base[i].meta & 15 == abi.RT_N_INNER and base[i].link > enter]
    if around:
        refused(node(around[-1], link=inside + 1), "range")
    if around and around[-1] == enter - 1:  # the culling box in front of the wrappers: a range that ends just inside them
        refused(node(enter - 1, link=enter + 1), "wrapper entered within")

    def xform(k, **kw):
        def change(c):
            for name, v in kw.items():
                setattr(c["xforms"][k], name, v)
        return change

    refused(xform(0, kind=2), "xform kind/depth")
    refused(xform(0, depth=5), "xform kind/depth")
    refused(xform(1, depth=1), "xform")

    def row(table, k, **kw):
        def change(c):
            for name, v in kw.items():
                setattr(c[table][k], name, v)
        return change

    refused(row("spheres", 0, mat=hs.desc.n_materials), "sphere material")
    refused(row("msphere", 0, mat=hs.desc.n_materials), "moving sphere material")
    refused(row("rects", 0, axis=3), "rect material/axis")
    refused(row("boxes", 0, mat=hs.desc.n_materials), "box material")
    refused(row("media", 0, mat=hs.desc.n_materials), "medium material")
    refused(row("media", 0, rng_id=0xFFF0), "rng id")
    refused(row("materials", 0, kind=9), "material kind")
    lam = next(k for k in range(hs.desc.n_materials) if pristine["materials"][k].kind == M_LAMBERTIAN)
    refused(row("materials", lam, tex=hs.desc.n_textures), "texture index")
    refused(row("textures", 0, kind=9), "texture kind")
    noise = next(k for k in range(hs.desc.n_textures) if pristine["textures"][k].kind == T_NOISE)
    refused(row("textures", noise, aux=hs.desc.n_perlins), "perlin index")
    image = next(k for k in range(hs.desc.n_textures) if pristine["textures"][k].kind == T_IMAGE)
    refused(row("textures", image, aux=hs.desc.n_images), "image index")
    refused(row("images", 0, width=0), "empty image")
    refused(row("images", 0, offset=hs.desc.n_texel_bytes), "outside texel blob")
    refused(lambda c: c["perlins"][0].perm_y.__setitem__(17, 256), "perlin permutation")
    refused(lambda c: c["perlins"][0].perm_x.__setitem__(3, -1), "perlin permutation")
    hs.close()


def test_a_medium_inside_a_boundary_is_refused(rtlib, orc):
    from scenegen import media_scene_pair
    hroot, oroot = media_scene_pair(rtlib, orc)
    hs = lib.HostScene(rtlib, hroot)
    c = Copy(hs)
    nodes = c["nodes"]
    a = first(nodes, abi.RT_N_MEDIUM)
    # the first entry of the boundary becomes a medium entry itself
    nodes[a + 1].meta = abi.RT_N_MEDIUM | (0 << 4)
    nodes[a + 1].link = nodes[a].link
    rc, msg = create(rtlib, c.desc)
    assert rc == abi.RT_E_INVALID and ("medium inside a medium boundary" in msg or "medium without a boundary" in msg), msg
    hs.close(); orc.api.scene_free(oroot)
