"""Thin ctypes front end of the scene-construction entry points declared in
include/rt/host.h. The same call shapes exist with another symbol prefix in the
CPU oracle used by the tests, so the wrapper is parameterised by prefix."""
import ctypes as C

import numpy as np

from . import abi


class HostApiError(RuntimeError):
    pass


class SceneApi:
    """Scene library, camera and builder calls of one shared library."""

    def __init__(self, lib, prefix, last_error):
        self.lib = lib
        self.prefix = prefix
        self._last_error = last_error
        self._declare()

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def check(self, rc):
        if rc < 0:
            msg = self._last_error()
            raise HostApiError("%s call failed (%d): %s" % (self.prefix, rc, msg.decode() if msg else ""))
        return rc

    def _declare(self):
        P, d, i, u32, u64, vp = C.POINTER, C.c_double, C.c_int, C.c_uint32, C.c_uint64, C.c_void_p
        d3p = P(C.c_double)
        sig = {
            "scene_build": [i, u64, P(abi.RtSceneOptions), P(vp)],
            "scene_setup": [i, P(abi.RtSceneSetup)],
            "camera_new": [d3p, d3p, d3p, d, d, d, d, d, d, P(abi.RtCamera)],
            "builder_new": [u64, P(vp)],
            "tex_solid": [vp, d3p], "tex_checker": [vp, d3p, d3p], "tex_noise": [vp, d],
            "tex_image": [vp, P(C.c_uint8), u32, u32],
            "mat_lambertian": [vp, i], "mat_metal": [vp, d3p, d], "mat_dielectric": [vp, d],
            "mat_diffuse_light": [vp, i], "mat_isotropic": [vp, i],
            "sphere": [vp, d3p, d, i], "moving_sphere": [vp, d3p, d3p, d, d, d, i],
            "xy_rect": [vp, d, d, d, d, d, i], "xz_rect": [vp, d, d, d, d, d, i], "yz_rect": [vp, d, d, d, d, d, i],
            "box6": [vp, d3p, d3p, i], "triangle": [vp, d3p, d3p, d3p, i],
            "translate": [vp, i, d3p], "rotate_y": [vp, i, d], "constant_medium": [vp, i, d, i],
            "list_new": [vp], "list_add": [vp, i, i], "bvh": [vp, i, d, d], "bvh_lbvh": [vp, i, d, d], "builder_finish": [vp, i, P(vp)],
        }
        for name, args in sig.items():
            f = self.fn(name)
            f.argtypes = args
            f.restype = C.c_int
        self.fn("scene_free").argtypes = [vp]
        self.fn("scene_free").restype = None
        self.fn("builder_free").argtypes = [vp]
        self.fn("builder_free").restype = None

    # -- scene library ----------------------------------------------------
    def scene_setup(self, scene_id):
        s = abi.RtSceneSetup()
        self.check(self.fn("scene_setup")(scene_id, C.byref(s)))
        return s

    def scene_build(self, scene_id, seed=0x5EED, flags=0, boxes_per_side=0, n_triangles=0, image=None):
        """image: uint8 array of shape (H, W, 3) or None. Returns an opaque scene handle (c_void_p)."""
        opt = abi.RtSceneOptions()
        opt.flags = flags
        opt.boxes_per_side = boxes_per_side
        opt.n_triangles = n_triangles
        keep = None
        if image is not None:
            keep = np.ascontiguousarray(image, dtype=np.uint8)
            opt.image_rgb = keep.ctypes.data_as(C.POINTER(C.c_uint8))
            opt.image_height, opt.image_width = keep.shape[0], keep.shape[1]
        h = C.c_void_p()
        self.check(self.fn("scene_build")(scene_id, seed, C.byref(opt), C.byref(h)))
        return h

    def scene_free(self, h):
        self.fn("scene_free")(h)

    def camera(self, setup, aspect_ratio=None):
        """Camera::new with the per-scene constants of an RtSceneSetup."""
        cam = abi.RtCamera()
        ar = setup.aspect_ratio if aspect_ratio is None else aspect_ratio
        self.check(self.fn("camera_new")(setup.lookfrom, setup.lookat, setup.vup, setup.vfov, ar, setup.aperture,
                                         setup.focus_dist, setup.time0, setup.time1, C.byref(cam)))
        return cam

    def camera_new(self, lookfrom, lookat, vup, vfov, aspect_ratio, aperture, focus_dist, t0=0.0, t1=1.0):
        cam = abi.RtCamera()
        self.check(self.fn("camera_new")(abi.vec3(lookfrom), abi.vec3(lookat), abi.vec3(vup), vfov, aspect_ratio,
                                         aperture, focus_dist, t0, t1, C.byref(cam)))
        return cam

    def builder(self, seed=0x5EED):
        return Builder(self, seed)


class Builder:
    """One method per reference constructor; returns integer handles."""

    def __init__(self, api, seed):
        self.api = api
        self.h = C.c_void_p()
        api.check(api.fn("builder_new")(seed, C.byref(self.h)))

    def close(self):
        if self.h:
            self.api.fn("builder_free")(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        return self.api.check(self.api.fn(name)(self.h, *args))

    def solid(self, c): return self._call("tex_solid", abi.vec3(c))
    def checker(self, c1, c2): return self._call("tex_checker", abi.vec3(c1), abi.vec3(c2))
    def noise(self, scale): return self._call("tex_noise", float(scale))

    def image(self, img):
        a = np.ascontiguousarray(img, dtype=np.uint8)
        return self._call("tex_image", a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[1], a.shape[0])

    def lambertian(self, tex_or_color):
        tex = tex_or_color if isinstance(tex_or_color, int) else self.solid(tex_or_color)
        return self._call("mat_lambertian", tex)

    def metal(self, albedo, fuzz): return self._call("mat_metal", abi.vec3(albedo), float(fuzz))
    def dielectric(self, ri): return self._call("mat_dielectric", float(ri))

    def diffuse_light(self, tex_or_color):
        tex = tex_or_color if isinstance(tex_or_color, int) else self.solid(tex_or_color)
        return self._call("mat_diffuse_light", tex)

    def isotropic(self, tex_or_color):
        tex = tex_or_color if isinstance(tex_or_color, int) else self.solid(tex_or_color)
        return self._call("mat_isotropic", tex)

    def sphere(self, c, r, mat): return self._call("sphere", abi.vec3(c), float(r), mat)

    def moving_sphere(self, c0, c1, t0, t1, r, mat):
        return self._call("moving_sphere", abi.vec3(c0), abi.vec3(c1), float(t0), float(t1), float(r), mat)

    def xy_rect(self, x0, x1, y0, y1, k, mat): return self._call("xy_rect", x0, x1, y0, y1, k, mat)
    def xz_rect(self, x0, x1, z0, z1, k, mat): return self._call("xz_rect", x0, x1, z0, z1, k, mat)
    def yz_rect(self, y0, y1, z0, z1, k, mat): return self._call("yz_rect", y0, y1, z0, z1, k, mat)
    def box6(self, p0, p1, mat): return self._call("box6", abi.vec3(p0), abi.vec3(p1), mat)
    def triangle(self, p1, p2, p0, mat): return self._call("triangle", abi.vec3(p1), abi.vec3(p2), abi.vec3(p0), mat)
    def translate(self, obj, off): return self._call("translate", obj, abi.vec3(off))
    def rotate_y(self, obj, angle): return self._call("rotate_y", obj, float(angle))

    def constant_medium(self, boundary, density, tex_or_color):
        tex = tex_or_color if isinstance(tex_or_color, int) else self.solid(tex_or_color)
        return self._call("constant_medium", boundary, float(density), tex)

    def list(self, objs=()):
        l = self._call("list_new")
        for o in objs:
            self._call("list_add", l, o)
        return l

    def list_add(self, l, obj): return self._call("list_add", l, obj)
    def bvh(self, lst, t0=0.0, t1=1.0): return self._call("bvh", lst, float(t0), float(t1))
    def bvh_lbvh(self, lst, t0=0.0, t1=1.0): return self._call("bvh_lbvh", lst, float(t0), float(t1))

    def finish(self, root):
        """Returns an opaque scene handle; the builder can be closed afterwards."""
        out = C.c_void_p()
        self._call("builder_finish", root, C.byref(out))
        return out
