"""ctypes mirrors of the structs in include/rt/abi.h and include/rt/host.h, plus
matching numpy dtypes for the per-ray arrays. Field order and sizes must match
the headers exactly; tests/test_abi.py checks sizeof against the library."""
import ctypes as C

import numpy as np

RT_MISS = 0xFFFFFFFF
RT_NONE = 0xFFFFFFFF
RT_ABI_VERSION = 2
RT_OK, RT_E_INVALID, RT_E_DEVICE, RT_E_NOMEM, RT_E_UNSUPPORTED = 0, -1, -2, -3, -4
RT_N_INNER, RT_N_SPHERE, RT_N_MSPHERE, RT_N_RECT, RT_N_BOX6, RT_N_TRIANGLE, RT_N_XF_ENTER, RT_N_XF_EXIT, RT_N_MEDIUM = range(9)
RT_INNER_CARRIED, RT_INNER_COVER = 1, 2  # markers of box entries (index field of meta)

d3 = C.c_double * 3


class RtNode(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("hi", C.c_float * 3), ("meta", C.c_uint32), ("link", C.c_uint32)]


class RtSphere(C.Structure):
    _fields_ = [("c", d3), ("r", C.c_double), ("mat", C.c_uint32), ("prim_id", C.c_uint32)]


class RtMovingSphere(C.Structure):
    _fields_ = [("c0", d3), ("c1", d3), ("t0", C.c_double), ("t1", C.c_double), ("r", C.c_double),
                ("mat", C.c_uint32), ("prim_id", C.c_uint32)]


class RtRect(C.Structure):
    _fields_ = [("a0", C.c_double), ("a1", C.c_double), ("b0", C.c_double), ("b1", C.c_double), ("k", C.c_double),
                ("axis", C.c_uint32), ("mat", C.c_uint32), ("prim_id", C.c_uint32), ("_pad", C.c_uint32)]


class RtBox6(C.Structure):
    _fields_ = [("p0", d3), ("p1", d3), ("mat", C.c_uint32), ("prim_id", C.c_uint32)]


class RtTriangle(C.Structure):
    _fields_ = [("p0", d3), ("e1", d3), ("e2", d3), ("mat", C.c_uint32), ("prim_id", C.c_uint32)]


class RtXform(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("depth", C.c_uint32), ("anc", C.c_uint32 * 4), ("inst_id", C.c_uint32),
                ("_pad", C.c_uint32), ("sin_theta", C.c_double), ("cos_theta", C.c_double), ("offset", d3)]


class RtMedium(C.Structure):
    _fields_ = [("neg_inv_density", C.c_double), ("mat", C.c_uint32), ("rng_id", C.c_uint32),
                ("prim_id", C.c_uint32), ("_pad", C.c_uint32)]


class RtMaterial(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("tex", C.c_uint32), ("albedo", d3), ("param", C.c_double)]


class RtTexture(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("aux", C.c_uint32), ("a", d3), ("b", d3)]


class RtPerlin(C.Structure):
    _fields_ = [("ran_vec", (C.c_double * 3) * 256), ("perm_x", C.c_int32 * 256), ("perm_y", C.c_int32 * 256),
                ("perm_z", C.c_int32 * 256)]


class RtImage(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("offset", C.c_uint64)]


class RtSceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("flags", C.c_uint32),
                ("nodes", C.POINTER(RtNode)), ("n_nodes", C.c_uint64),
                ("spheres", C.POINTER(RtSphere)), ("n_spheres", C.c_uint64),
                ("msphere", C.POINTER(RtMovingSphere)), ("n_msphere", C.c_uint64),
                ("rects", C.POINTER(RtRect)), ("n_rects", C.c_uint64),
                ("boxes", C.POINTER(RtBox6)), ("n_boxes", C.c_uint64),
                ("triangles", C.POINTER(RtTriangle)), ("n_triangles", C.c_uint64),
                ("xforms", C.POINTER(RtXform)), ("n_xforms", C.c_uint64),
                ("media", C.POINTER(RtMedium)), ("n_media", C.c_uint64),
                ("materials", C.POINTER(RtMaterial)), ("n_materials", C.c_uint64),
                ("textures", C.POINTER(RtTexture)), ("n_textures", C.c_uint64),
                ("perlins", C.POINTER(RtPerlin)), ("n_perlins", C.c_uint64),
                ("images", C.POINTER(RtImage)), ("n_images", C.c_uint64),
                ("texels", C.POINTER(C.c_uint8)), ("n_texel_bytes", C.c_uint64)]


class RtRay(C.Structure):
    _fields_ = [("o", d3), ("d", d3), ("tm", C.c_double), ("tmin", C.c_double), ("tmax", C.c_double),
                ("pixel", C.c_uint32), ("sample", C.c_uint32), ("bounce", C.c_uint32), ("_pad", C.c_uint32)]


class RtHit(C.Structure):
    _fields_ = [("prim_id", C.c_uint32), ("sub", C.c_uint32), ("t", C.c_double), ("u", C.c_double), ("v", C.c_double)]


class RtHitRecord(C.Structure):
    _fields_ = [("p", d3), ("normal", d3), ("t", C.c_double), ("u", C.c_double), ("v", C.c_double),
                ("prim_id", C.c_uint32), ("sub", C.c_uint32), ("mat", C.c_uint32), ("front_face", C.c_uint32)]


class RtPathStep(C.Structure):
    _fields_ = [("emitted", d3), ("attenuation", d3), ("o", d3), ("d", d3), ("t", C.c_double),
                ("prim_id", C.c_uint32), ("sub", C.c_uint32), ("scattered", C.c_uint32), ("_pad", C.c_uint32)]


class RtCamera(C.Structure):
    _fields_ = [("orig", d3), ("lower_left_corner", d3), ("horizontal", d3), ("vertical", d3), ("u", d3), ("v", d3),
                ("lens_radius", C.c_double), ("time0", C.c_double), ("time1", C.c_double)]


class RtRenderParams(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("sample_begin", C.c_uint32), ("sample_end", C.c_uint32),
                ("max_depth", C.c_uint32), ("samples_per_batch", C.c_uint32), ("seed", C.c_uint64), ("background", d3)]


class RtTraceStats(C.Structure):
    _fields_ = [("ms_kernel", C.c_double)]


class RtRenderStats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("ms_total", C.c_double), ("ms_extend", C.c_double),
                ("ms_shade", C.c_double), ("batches", C.c_uint32), ("_pad", C.c_uint32)]


class RtSceneOptions(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("boxes_per_side", C.c_uint32), ("n_triangles", C.c_uint64),
                ("image_rgb", C.POINTER(C.c_uint8)), ("image_width", C.c_uint32), ("image_height", C.c_uint32)]


class RtSceneSetup(C.Structure):
    _fields_ = [("lookfrom", d3), ("lookat", d3), ("vup", d3), ("vfov", C.c_double), ("aspect_ratio", C.c_double),
                ("aperture", C.c_double), ("focus_dist", C.c_double), ("time0", C.c_double), ("time1", C.c_double),
                ("background", d3), ("samples_per_pixel", C.c_uint32), ("width", C.c_uint32)]


RT_DESC_BOXES_MAY_NOT_BOUND = 1
RT_SCENE_BVH_SHUTTER_TIMES = 1
RT_SCENE_BVH_WIDEST_AXIS = 2
RT_SCENE_BVH_LBVH = 4
RT_BVH_TOKEN_LEAF = 0xF
RT_BVH_MAX_OBJECTS = 1 << 28

# numpy views of the per-ray structs (same layout, no padding surprises: all members are 4/8-byte aligned)
RAY_DTYPE = np.dtype([("o", "<f8", 3), ("d", "<f8", 3), ("tm", "<f8"), ("tmin", "<f8"), ("tmax", "<f8"),
                      ("pixel", "<u4"), ("sample", "<u4"), ("bounce", "<u4"), ("_pad", "<u4")])
HIT_DTYPE = np.dtype([("prim_id", "<u4"), ("sub", "<u4"), ("t", "<f8"), ("u", "<f8"), ("v", "<f8")])
HITREC_DTYPE = np.dtype([("p", "<f8", 3), ("normal", "<f8", 3), ("t", "<f8"), ("u", "<f8"), ("v", "<f8"),
                         ("prim_id", "<u4"), ("sub", "<u4"), ("mat", "<u4"), ("front_face", "<u4")])
PATHSTEP_DTYPE = np.dtype([("emitted", "<f8", 3), ("attenuation", "<f8", 3), ("o", "<f8", 3), ("d", "<f8", 3), ("t", "<f8"),
                           ("prim_id", "<u4"), ("sub", "<u4"), ("scattered", "<u4"), ("_pad", "<u4")])
RT_PROFILE_OWN = -1
NODE_DTYPE = np.dtype([("lo", "<f4", 3), ("hi", "<f4", 3), ("meta", "<u4"), ("link", "<u4")])
assert NODE_DTYPE.itemsize == C.sizeof(RtNode) == 32
assert RAY_DTYPE.itemsize == C.sizeof(RtRay) == 88
assert HIT_DTYPE.itemsize == C.sizeof(RtHit) == 32
assert HITREC_DTYPE.itemsize == C.sizeof(RtHitRecord) == 88
assert PATHSTEP_DTYPE.itemsize == C.sizeof(RtPathStep) == 120

NODE_KINDS = {0: "inner", 1: "sphere", 2: "msphere", 3: "rect", 4: "box6", 5: "triangle", 6: "xf_enter", 7: "xf_exit",
              8: "medium"}


def vec3(v):
    return d3(float(v[0]), float(v[1]), float(v[2]))
