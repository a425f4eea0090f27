/*
 * Host side of `librt_hip.so`: the reference's scene-construction API restated in
 * C++ (raytracer_amd/csrc/host/) and exported with C linkage so that harnesses
 * in any language can build a scene, flatten it and hand the resulting
 * RtSceneDesc to rt_scene_create (include/rt/abi.h).
 *
 * Two ways to get a scene:
 *  1. rth_scene_build: the reference's scene library (src/main.rs:244-765) plus
 *     one build-defined scene (the kaleidoscope with a synthetic triangle soup).
 *  2. the rth_builder_* calls: one call per reference constructor, handles
 *     instead of Arc<dyn ...>, for ad-hoc scenes.
 * Either way the object tree is built exactly as the reference builds it
 * (BVHNode::new, src/bvh.rs:17-98, including its random split axes, drawn from
 * the seeded scene stream) and then flattened depth-first.
 *
 * Identification used in RtHit: every Sphere, MovingSphere, X?Rect, Box6,
 * Triangle and ConstantMedium takes the next prim_id at construction; every
 * Translate and RotateY the next inst_id; every ConstantMedium the next rng_id;
 * all counted from 0 in creation order within one scene.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "abi.h"

#ifdef __cplusplus
extern "C" {
#endif

enum RtSceneId {                 /* the `match scene` of src/main.rs:65-162 */
    RT_SCENE_RANDOM = 1,         /* random_scene, src/main.rs:244-316, aperture 0.1 */
    RT_SCENE_RANDOM_PINHOLE = 2, /* same world, aperture 0 (src/main.rs:75-80) */
    RT_SCENE_SIMPLE_LIGHT = 3,   /* src/main.rs:318-344 */
    RT_SCENE_LIGHT_DEMO = 4,     /* src/main.rs:346-422 */
    RT_SCENE_TWO_PERLIN = 5,     /* src/main.rs:424-442 */
    RT_SCENE_EARTH = 6,          /* src/main.rs:444-457 */
    RT_SCENE_CORNELL = 7,        /* src/main.rs:459-517 */
    RT_SCENE_CORNELL_SMOKE = 8,  /* src/main.rs:519-585 */
    RT_SCENE_FINAL = 9,          /* src/main.rs:587-699 */
    RT_SCENE_TRIANGLE = 10,      /* src/main.rs:701-738 */
    RT_SCENE_KALEIDOSCOPE = 11,  /* src/main.rs:740-765 */
    RT_SCENE_KALEIDOSCOPE_MESH = 12 /* scene 11 + n_triangles synthetic Lambertian triangles (no reference counterpart) */
};

#define RT_SCENE_BVH_SHUTTER_TIMES 1u  /* random_scene: build the BVH over the shutter interval (0,1) instead of the
                                          reference's (0,0.1) (src/main.rs:315), so moving-sphere boxes bound them */

#define RT_SCENE_BVH_WIDEST_AXIS 2u    /* every BVHNode splits along the axis on which its objects' box minima spread the
                                          widest instead of the reference's random axis (src/bvh.rs:24): same closest
                                          hits (up to exact ties in t), far fewer boxes visited on large meshes */

#define RT_SCENE_BVH_LBVH 4u           /* every BVHNode over two or more objects is built by rt_bvh_build on the current
                                          HIP device (Morton order, radix tree) instead of src/bvh.rs:17-98; the scene
                                          stream moves on by the draws of the replaced builder (the scene's geometry
                                          stays the same); same closest hits up to exact ties in t.
                                          Needs a GPU at scene construction; takes precedence over WIDEST_AXIS */

typedef struct RtSceneOptions {
    uint32_t flags;
    uint32_t boxes_per_side;     /* final_scene ground grid (src/main.rs:591); 0 = 20 */
    uint64_t n_triangles;        /* RT_SCENE_KALEIDOSCOPE_MESH: lower bound on the synthetic triangle count */
    const uint8_t* image_rgb;    /* decoded earthmap (src/main.rs:445,665): RGB u8, top row first; NULL = 2x2 grey */
    uint32_t image_width, image_height;
} RtSceneOptions;

typedef struct RtSceneSetup {    /* per-scene constants of src/main.rs:55-179 */
    double lookfrom[3], lookat[3], vup[3];
    double vfov, aspect_ratio, aperture, focus_dist, time0, time1;
    double background[3];
    uint32_t samples_per_pixel;
    uint32_t width;              /* src/main.rs:164 (800); height = width / aspect_ratio */
} RtSceneSetup;

typedef struct RtHostScene RtHostScene;
typedef struct RtBuilder RtBuilder;

int rth_scene_build(int scene_id, uint64_t seed, const RtSceneOptions* options, RtHostScene** out);
const RtSceneDesc* rth_scene_desc(const RtHostScene* scene);
int rth_scene_setup(int scene_id, RtSceneSetup* out);
void rth_scene_free(RtHostScene* scene);

/* Camera::new, src/camera.rs:27-59 */
int rth_camera_new(const double lookfrom[3], const double lookat[3], const double vup[3], double vfov, double aspect_ratio,
                   double aperture, double focus_dist, double t0, double t1, RtCamera* out);

/* Builder: every call returns a handle >= 0 or a negative RtError. */
int rth_builder_new(uint64_t seed, RtBuilder** out);
void rth_builder_free(RtBuilder* b);
int rth_tex_solid(RtBuilder* b, const double c[3]);                            /* SolidColor::new, src/texture.rs:17 */
int rth_tex_checker(RtBuilder* b, const double c1[3], const double c2[3]);     /* CheckerTexture::new, :34 */
int rth_tex_noise(RtBuilder* b, double scale);                                 /* NoiseTexture::new, :59 */
int rth_tex_image(RtBuilder* b, const uint8_t* rgb, uint32_t w, uint32_t h);   /* ImageTexture::new, :82 (decoded) */
int rth_mat_lambertian(RtBuilder* b, int tex);                                 /* Lambertian{albedo}, src/material.rs:22-32 */
int rth_mat_metal(RtBuilder* b, const double albedo[3], double fuzz);          /* Metal::new, :59 */
int rth_mat_dielectric(RtBuilder* b, double ref_idx);                          /* Dielectric::new, :91 */
int rth_mat_diffuse_light(RtBuilder* b, int tex);                              /* DiffuseLight::new, :155 */
int rth_mat_isotropic(RtBuilder* b, int tex);                                  /* Isotropic::new, :186 */
int rth_sphere(RtBuilder* b, const double c[3], double r, int mat);            /* Sphere::new, src/hittable.rs:55 */
int rth_moving_sphere(RtBuilder* b, const double c0[3], const double c1[3], double t0, double t1, double r, int mat); /* :197 */
int rth_xy_rect(RtBuilder* b, double x0, double x1, double y0, double y1, double k, int mat);  /* src/aarect.rs:17 */
int rth_xz_rect(RtBuilder* b, double x0, double x1, double z0, double z1, double k, int mat);  /* :68 */
int rth_yz_rect(RtBuilder* b, double y0, double y1, double z0, double z1, double k, int mat);  /* :119 */
int rth_box6(RtBuilder* b, const double p0[3], const double p1[3], int mat);   /* Box6::new, src/box6.rs:16 */
int rth_triangle(RtBuilder* b, const double p1[3], const double p2[3], const double p0[3], int mat); /* src/triangle.rs:15 */
int rth_translate(RtBuilder* b, int obj, const double offset[3]);              /* Translate::new, src/hittable.rs:268 */
int rth_rotate_y(RtBuilder* b, int obj, double angle_degrees);                 /* RotateY::new, :315 */
int rth_constant_medium(RtBuilder* b, int boundary, double density, int tex);  /* ConstantMedium::new, src/constant_medium.rs:17 */
int rth_list_new(RtBuilder* b);                                                /* HittableList::new, src/hittable.rs:129 */
int rth_list_add(RtBuilder* b, int list, int obj);                             /* HittableList::add, :139 */
int rth_bvh(RtBuilder* b, int list, double time0, double time1);               /* BVHNode::new, src/bvh.rs:94 */
int rth_bvh_lbvh(RtBuilder* b, int list, double time0, double time1);        /* same place in the scene, the tree made by rt_bvh_build (two or more objects) */
int rth_builder_finish(RtBuilder* b, int root, RtHostScene** out);             /* flatten the tree under `root` (a BVHNode) */

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
