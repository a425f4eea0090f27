"""ctypes binding of raytracer_amd/librt_hip.so (include/rt/abi.h, include/rt/host.h).

There is no CPU fallback: if the shared library is missing, or no GPU is present
when a device call is made, the call raises."""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

from . import abi
from .hostapi import SceneApi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")  # RT_HIP_LIB: developer override for A/B builds

ABI_SYMBOLS = (
    "rt_abi_version", "rt_last_error", "rt_device_count", "rt_scene_create", "rt_scene_destroy", "rt_trace_closest",
    "rt_trace_records", "rt_trace_closest_device", "rt_trace_work", "rt_render", "rt_render_device", "rt_render_work",
    "rt_path_step", "rt_scene_profile", "rt_resolve_u8",
    "rt_resolve_u8_device", "rt_bvh_build", "rt_bvh_build_device", "rt_bvh_build_device_noalloc", "rt_bvh_expand_device", "rt_bvh_build_stream",
    "rt_bvh_refit_device",
    "rt_sharded_create", "rt_sharded_destroy", "rt_sharded_device_count", "rt_sharded_render", "rt_shard_samples",
)
HOST_SYMBOLS = (
    "rth_scene_build", "rth_scene_desc", "rth_scene_setup", "rth_scene_free", "rth_camera_new", "rth_builder_new",
    "rth_builder_free", "rth_tex_solid", "rth_tex_checker", "rth_tex_noise", "rth_tex_image", "rth_mat_lambertian",
    "rth_mat_metal", "rth_mat_dielectric", "rth_mat_diffuse_light", "rth_mat_isotropic", "rth_sphere",
    "rth_moving_sphere", "rth_xy_rect", "rth_xz_rect", "rth_yz_rect", "rth_box6", "rth_triangle", "rth_translate",
    "rth_rotate_y", "rth_constant_medium", "rth_list_new", "rth_list_add", "rth_bvh", "rth_bvh_lbvh", "rth_builder_finish",
)


class RtError(RuntimeError):
    pass


def _share_the_hip_runtime_with_torch():
    """PyTorch wheels bundle their own libamdhip64 / libhsa-runtime64. A process that loads the system runtime first
    (through librt_hip.so) and PyTorch afterwards ends up with torch refusing to see the GPU ("No HIP GPUs are
    available"); the other order works, because librt_hip.so then binds to the runtime torch brought. So when torch
    is installed but not imported yet, its runtime is loaded first; device buffers and streams made by torch can
    then be handed to rt_render_device / rt_trace_closest_device whichever module is imported first."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


class Library:
    def __init__(self, path=LIB_PATH):
        if not os.path.exists(path):
            raise RtError("%s is missing: build it with `make -C raytracer_amd/csrc` "
                          "(or __graft_entry__.build()); there is no fallback path" % path)
        _share_the_hip_runtime_with_torch()
        self.lib = C.CDLL(path)
        L = self.lib
        L.rt_last_error.restype = C.c_char_p
        L.rt_abi_version.restype = C.c_uint32
        self.host = SceneApi(L, "rth_", L.rt_last_error)
        vp, u32, u64, i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
        P = C.POINTER
        L.rth_scene_desc.argtypes = [vp]
        L.rth_scene_desc.restype = P(abi.RtSceneDesc)
        L.rt_device_count.restype = i
        L.rt_scene_create.argtypes = [P(abi.RtSceneDesc), P(vp)]
        L.rt_scene_destroy.argtypes = [vp]
        L.rt_scene_destroy.restype = None
        L.rt_trace_closest.argtypes = [vp, vp, u64, u64, vp, P(abi.RtTraceStats)]
        L.rt_trace_records.argtypes = [vp, vp, u64, u64, vp]
        L.rt_trace_closest_device.argtypes = [vp, vp, u64, u64, vp, vp]
        L.rt_trace_work.argtypes = [vp, vp, u64, u64, vp, P(u64)]
        L.rt_render.argtypes = [vp, P(abi.RtCamera), P(abi.RtRenderParams), vp, P(abi.RtRenderStats)]
        L.rt_render_device.argtypes = [vp, P(abi.RtCamera), P(abi.RtRenderParams), vp, vp, P(abi.RtRenderStats)]
        L.rt_render_work.argtypes = [vp, P(abi.RtCamera), P(abi.RtRenderParams), vp, vp, P(abi.RtRenderStats), P(u64)]
        L.rt_path_step.argtypes = [vp, P(abi.RtRenderParams), i, vp, u64, vp, P(u64)]
        L.rt_scene_profile.argtypes = [vp]
        L.rt_resolve_u8.argtypes = [vp, u32, u32, u64, vp]
        L.rt_resolve_u8_device.argtypes = [vp, u32, u32, u64, vp, vp]
        L.rt_bvh_build.argtypes = [vp, u64, vp, P(C.c_double)]
        L.rt_bvh_build_device.argtypes = [vp, u64, vp, vp, vp, P(C.c_double)]
        L.rt_bvh_build_device_noalloc.argtypes = [vp, u64, vp, vp, vp]
        L.rt_bvh_expand_device.argtypes = [vp, u64, vp, vp, u64, vp, vp]
        L.rt_bvh_build_stream.argtypes = [vp, u64, vp, vp, vp, P(C.c_double)]
        L.rt_bvh_refit_device.argtypes = [vp, u64, vp, vp, vp]
        L.rt_sharded_create.argtypes = [P(abi.RtSceneDesc), P(i), i, P(vp)]
        L.rt_sharded_destroy.argtypes = [vp]
        L.rt_sharded_destroy.restype = None
        L.rt_sharded_device_count.argtypes = [vp]
        L.rt_sharded_render.argtypes = [vp, P(abi.RtCamera), P(abi.RtRenderParams), vp, P(abi.RtRenderStats)]
        L.rt_shard_samples.argtypes = [u32, u32, u32, u32, P(u32), P(u32)]
        L.rt_shard_samples.restype = None
        for name in ABI_SYMBOLS:
            f = getattr(L, name)
            if name not in ("rt_last_error", "rt_abi_version", "rt_scene_destroy", "rt_sharded_destroy", "rt_shard_samples"):
                f.restype = i
        if L.rt_abi_version() != abi.RT_ABI_VERSION:
            raise RtError("ABI version mismatch between librt_hip.so and raytracer_amd/abi.py")

    def check(self, rc):
        if rc != 0:
            msg = self.lib.rt_last_error()
            raise RtError("librt_hip call failed (%d): %s" % (rc, msg.decode() if msg else ""))

    def device_count(self):
        return self.lib.rt_device_count()

    def bvh_build(self, boxes):
        """rt_bvh_build: (n, 6) f64 boxes -> (2 n - 1) pre-order tokens (structured array) and the device time in ms."""
        boxes = np.ascontiguousarray(boxes, dtype=np.float64)
        assert boxes.ndim == 2 and boxes.shape[1] == 6
        tokens = np.zeros(max(2 * boxes.shape[0] - 1, 1), dtype=abi.NODE_DTYPE)
        ms = C.c_double(0.0)
        self.check(self.lib.rt_bvh_build(boxes.ctypes.data, boxes.shape[0], tokens.ctypes.data, C.byref(ms)))
        return tokens, ms.value


_LIB = None


def get():
    global _LIB
    if _LIB is None:
        _LIB = Library()
    return _LIB


class HostScene:
    """A flattened scene on the host (RtHostScene) — tables readable as numpy arrays."""

    def __init__(self, lib, handle):
        self.lib = lib
        self.h = handle
        self.desc = lib.lib.rth_scene_desc(handle).contents

    @classmethod
    def build(cls, scene_id, lib=None, **kw):
        lib = lib or get()
        return cls(lib, lib.host.scene_build(scene_id, **kw))

    def table(self, name):
        """Copy of one table of the description as a ctypes-backed numpy structured array."""
        counts = {"nodes": "n_nodes", "spheres": "n_spheres", "msphere": "n_msphere", "rects": "n_rects", "boxes": "n_boxes",
                  "triangles": "n_triangles", "xforms": "n_xforms", "media": "n_media", "materials": "n_materials",
                  "textures": "n_textures", "perlins": "n_perlins", "images": "n_images"}
        n = getattr(self.desc, counts[name])
        ptr = getattr(self.desc, name)
        typ = ptr._type_
        if n == 0:
            return np.zeros(0, dtype=np.dtype(typ))
        arr = (typ * n).from_address(C.addressof(ptr.contents))
        return np.frombuffer(bytes(arr), dtype=np.dtype(typ))

    def close(self):
        if self.h:
            self.lib.host.scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceScene:
    """An RtScene resident in the HBM of the current HIP device."""

    def __init__(self, host_scene, lib=None):
        self.lib = lib or host_scene.lib
        self.h = C.c_void_p()
        self.lib.check(self.lib.lib.rt_scene_create(C.byref(host_scene.desc), C.byref(self.h)))

    def close(self):
        if self.h:
            self.lib.lib.rt_scene_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trace(self, rays, seed=0x5EED):
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=abi.HIT_DTYPE)
        st = abi.RtTraceStats()
        self.lib.check(self.lib.lib.rt_trace_closest(self.h, rays.ctypes.data, rays.shape[0], seed, hits.ctypes.data, C.byref(st)))
        return hits, st.ms_kernel

    def trace_records(self, rays, seed=0x5EED):
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        recs = np.zeros(rays.shape[0], dtype=abi.HITREC_DTYPE)
        self.lib.check(self.lib.lib.rt_trace_records(self.h, rays.ctypes.data, rays.shape[0], seed, recs.ctypes.data))
        return recs

    def trace_work(self, rays, seed=0x5EED):
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=abi.HIT_DTYPE)
        work = (C.c_uint64 * 10)()
        self.lib.check(self.lib.lib.rt_trace_work(self.h, rays.ctypes.data, rays.shape[0], seed, hits.ctypes.data, work))
        w = list(work)
        return hits, {"entries": w[0], **{abi.NODE_KINDS[k]: w[1 + k] for k in range(9)}}

    def profile(self):
        """The closest-hit kernel profile rt_render runs on this scene (bits: 1 wrapped boxes in one step, 2 general
        medium boundary walk, 4 vote exit from the box phase)."""
        rc = self.lib.lib.rt_scene_profile(self.h)
        if rc < 0:
            self.lib.check(rc)
        return rc

    def path_step(self, params, rays, profile=abi.RT_PROFILE_OWN, want_work=False):
        """rt_path_step: one bounce of the render path's own extend + shade launches on the given rays."""
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        out = np.zeros(rays.shape[0], dtype=abi.PATHSTEP_DTYPE)
        work = (C.c_uint64 * 10)()
        self.lib.check(self.lib.lib.rt_path_step(self.h, C.byref(params), profile, rays.ctypes.data, rays.shape[0], out.ctypes.data,
                                                work if want_work else None))
        if not want_work:
            return out
        w = list(work)
        return out, {"entries": w[0], **{abi.NODE_KINDS[k]: w[1 + k] for k in range(9)}}

    def render(self, cam, params, accum=None, want_stats=True):
        """rt_render into a host float32 [H, W, 3] buffer (created zeroed if not given)."""
        if accum is None:
            accum = np.zeros((params.height, params.width, 3), dtype=np.float32)
        assert accum.dtype == np.float32 and accum.flags["C_CONTIGUOUS"]
        st = abi.RtRenderStats()
        self.lib.check(self.lib.lib.rt_render(self.h, C.byref(cam), C.byref(params), accum.ctypes.data,
                                             C.byref(st) if want_stats else None))
        return accum, st

    def render_device(self, cam, params, accum_ptr, stream=None, stats=None):
        """rt_render_device: accum_ptr is a device address (int), stream a hipStream_t handle (int) or None."""
        self.lib.check(self.lib.lib.rt_render_device(self.h, C.byref(cam), C.byref(params), C.c_void_p(accum_ptr),
                                                    C.c_void_p(stream or 0), C.byref(stats) if stats is not None else None))


    def render_work(self, cam, params, accum_ptr, stream=None):
        """rt_render_work: returns (stats, work dict) for an instrumented render into a device buffer."""
        st = abi.RtRenderStats()
        work = (C.c_uint64 * 14)()
        self.lib.check(self.lib.lib.rt_render_work(self.h, C.byref(cam), C.byref(params), C.c_void_p(accum_ptr),
                                                  C.c_void_p(stream or 0), C.byref(st), work))
        w = list(work)
        return st, {"entries": w[0], **{abi.NODE_KINDS[k]: w[1 + k] for k in range(9)}, "lit_slots": w[10],
                    "first_media": w[11], "first_spheres": w[12]}


def kernel_source_hash():
    """sha256 (first 16 hex digits) over the sources of the device code, in a fixed order: identifies the kernels a
    counter measurement under profiles/ was taken with."""
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    names = sorted(os.path.join("device", f) for f in os.listdir(os.path.join(csrc, "device")) if f.endswith(".h")) + ["rng.h", "rt_hip.hip"]
    h = hashlib.sha256()
    for name in names:
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


class ShardedScene:
    """An RtShardedScene: one replica of the scene per GPU of this process, sample ranges sharded over them."""

    def __init__(self, host_scene, devices=None, n_devices=0, lib=None):
        self.lib = lib or host_scene.lib
        self.h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices) if devices is not None else None
        n = len(devices) if devices is not None else n_devices
        self.lib.check(self.lib.lib.rt_sharded_create(C.byref(host_scene.desc), arr, n, C.byref(self.h)))

    def device_count(self):
        return self.lib.lib.rt_sharded_device_count(self.h)

    def render(self, cam, params, accum=None):
        if accum is None:
            accum = np.zeros((params.height, params.width, 3), dtype=np.float32)
        assert accum.dtype == np.float32 and accum.flags["C_CONTIGUOUS"]
        st = abi.RtRenderStats()
        self.lib.check(self.lib.lib.rt_sharded_render(self.h, C.byref(cam), C.byref(params), accum.ctypes.data, C.byref(st)))
        return accum, st

    def close(self):
        if self.h:
            self.lib.lib.rt_sharded_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def shard_samples(lib, begin, end, rank, world):
    """rt_shard_samples: the partition rt_sharded_render uses."""
    lo, hi = C.c_uint32(), C.c_uint32()
    lib.lib.rt_shard_samples(begin, end, rank, world, C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def make_params(width, height, s0, s1, background, seed=0x5EED, max_depth=50, samples_per_batch=0):
    p = abi.RtRenderParams()
    p.width, p.height, p.sample_begin, p.sample_end = width, height, s0, s1
    p.max_depth, p.samples_per_batch, p.seed = max_depth, samples_per_batch, seed
    p.background = abi.vec3(background)
    return p


def resolve_u8(accum, spp, lib=None):
    lib = lib or get()
    a = np.ascontiguousarray(accum, dtype=np.float32)
    out = np.zeros(a.shape, dtype=np.uint8)
    lib.check(lib.lib.rt_resolve_u8(a.ctypes.data, a.shape[1], a.shape[0], spp, out.ctypes.data))
    return out
