"""TEST INFRASTRUCTURE ONLY — ctypes loader of the CPU oracle (oracle/liboracle.so).

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg and by
nothing else; the product package raytracer_amd never imports this module."""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracer_amd import abi
from raytracer_amd.hostapi import SceneApi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

AABB_REFERENCE = 0  # src/aabb.rs:15-30 as written
AABB_TIGHT = 1      # carried slab test on f32 boxes (what the HIP path runs where boxes bound their contents)
AABB_PER_AXIS_F32 = 2  # the reference's per-axis rule on the f32 boxes (what the HIP path runs where they may not)

STAT_NAMES = ("box_tests", "sphere", "msphere", "rect", "box6", "triangle", "medium", "rays")


def build(force=False):
    """Compile liboracle.so with the recipe in oracle/Makefile."""
    if force and os.path.exists(LIB_PATH):
        os.remove(LIB_PATH)
    subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, stdout=subprocess.DEVNULL)


class Oracle:
    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            build()
        self.lib = C.CDLL(path)
        L = self.lib
        L.orc_last_error.restype = C.c_char_p
        self.api = SceneApi(L, "orc_", L.orc_last_error)
        vp, u32, u64, i, d = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double
        P = C.POINTER
        L.orc_philox.argtypes = [P(u32), P(u32), P(u32)]
        L.orc_philox.restype = None
        L.orc_rand_unit.argtypes = [u64, u32, u32, u32, u64]
        L.orc_rand_unit.restype = d
        L.orc_vec3_op.argtypes = [i, P(d), P(d), d, P(d)]
        L.orc_vec3_op.restype = i
        L.orc_resolve_u8.argtypes = [P(d), u32, u32, u64, P(C.c_uint8)]
        L.orc_resolve_u8.restype = i
        L.orc_scene_prim_count.argtypes = [vp]
        L.orc_scene_prim_count.restype = u32
        L.orc_trace.argtypes = [vp, vp, u64, u64, i, vp, vp, P(u64)]
        L.orc_trace.restype = i
        L.orc_render.argtypes = [vp, P(abi.RtCamera), P(abi.RtRenderParams), i, i, P(d), P(u64), P(d)]
        L.orc_render.restype = i
        L.orc_harvest.argtypes = [vp, P(abi.RtCamera), P(abi.RtRenderParams), i, u32, u64, u64, vp, P(u64)]
        L.orc_harvest.restype = i
        L.orc_path_step.argtypes = [vp, P(abi.RtRenderParams), i, vp, u64, vp]
        L.orc_path_step.restype = i
        L.orc_lbvh_tokens.argtypes = [vp, u64, vp]
        L.orc_lbvh_tokens.restype = i

    # ---- known-answer helpers ------------------------------------------
    def philox(self, ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        self.lib.orc_philox(c, k, o)
        return list(o)

    def rand_unit(self, seed, a, b, c, k):
        return self.lib.orc_rand_unit(seed, a, b, c, k)

    def vec3_op(self, op, a, b=None, s=0.0):
        out = (C.c_double * 3)()
        bb = abi.vec3(b) if b is not None else None
        rc = self.lib.orc_vec3_op(op, abi.vec3(a), bb, s, out)
        return rc, list(out)

    def resolve_u8(self, accum, spp):
        a = np.ascontiguousarray(accum, dtype=np.float64)
        h, w = a.shape[0], a.shape[1]
        out = np.zeros((h, w, 3), dtype=np.uint8)
        self.lib.orc_resolve_u8(a.ctypes.data_as(C.POINTER(C.c_double)), w, h, spp,
                                out.ctypes.data_as(C.POINTER(C.c_uint8)))
        return out

    def lbvh_tokens(self, boxes):
        """The checker of rt_bvh_build: (n, 6) f64 boxes -> (2 n - 1) pre-order tokens."""
        boxes = np.ascontiguousarray(boxes, dtype=np.float64)
        assert boxes.ndim == 2 and boxes.shape[1] == 6
        tokens = np.zeros(max(2 * boxes.shape[0] - 1, 1), dtype=abi.NODE_DTYPE)
        self.api.check(self.lib.orc_lbvh_tokens(boxes.ctypes.data, boxes.shape[0], tokens.ctypes.data))
        return tokens

    # ---- queries ---------------------------------------------------------
    def trace(self, scene, rays, seed=0x5EED, mode=AABB_TIGHT, records=False):
        """world.hit for each ray. Returns (hits, records or None, stats dict)."""
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        n = rays.shape[0]
        hits = np.zeros(n, dtype=abi.HIT_DTYPE)
        recs = np.zeros(n, dtype=abi.HITREC_DTYPE) if records else None
        stats = (C.c_uint64 * 8)()
        rc = self.lib.orc_trace(scene, rays.ctypes.data, n, seed, mode, hits.ctypes.data,
                                recs.ctypes.data if records else None, stats)
        self.api.check(rc)
        return hits, recs, dict(zip(STAT_NAMES, stats))

    def path_step(self, scene, params, rays, mode=AABB_TIGHT):
        """One iteration of ray_color per ray (closest hit, emitted, scatter): the checker of rt_path_step."""
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        out = np.zeros(rays.shape[0], dtype=abi.PATHSTEP_DTYPE)
        self.api.check(self.lib.orc_path_step(scene, C.byref(params), mode, rays.ctypes.data, rays.shape[0], out.ctypes.data))
        return out

    def make_params(self, width, height, s0, s1, background, seed=0x5EED, max_depth=50, samples_per_batch=0):
        p = abi.RtRenderParams()
        p.width, p.height, p.sample_begin, p.sample_end = width, height, s0, s1
        p.max_depth, p.samples_per_batch, p.seed = max_depth, samples_per_batch, seed
        p.background = abi.vec3(background)
        return p

    def render(self, scene, cam, params, mode=AABB_TIGHT, threads=8):
        """Returns (accum f64 [H, W, 3] sum over samples, stats dict, seconds)."""
        accum = np.zeros((params.height, params.width, 3), dtype=np.float64)
        stats = (C.c_uint64 * 8)()
        secs = C.c_double()
        rc = self.lib.orc_render(scene, C.byref(cam), C.byref(params), mode, threads,
                                 accum.ctypes.data_as(C.POINTER(C.c_double)), stats, C.byref(secs))
        self.api.check(rc)
        return accum, dict(zip(STAT_NAMES, stats)), secs.value

    def harvest(self, scene, cam, params, n_paths, max_rays, max_bounce=3, mode=AABB_TIGHT):
        """Rays handed to world.hit by n_paths path samples (bounces 0..max_bounce)."""
        rays = np.zeros(max_rays, dtype=abi.RAY_DTYPE)
        n_out = C.c_uint64()
        rc = self.lib.orc_harvest(scene, C.byref(cam), C.byref(params), mode, max_bounce, n_paths, max_rays,
                                  rays.ctypes.data, C.byref(n_out))
        self.api.check(rc)
        return rays[: n_out.value].copy()


_ORACLE = None


def get():
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle()
    return _ORACLE
