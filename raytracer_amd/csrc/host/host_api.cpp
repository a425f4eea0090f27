// C entry points of include/rt/host.h: scene library, camera and the handle-based
// builder. Exceptions thrown by the object model (the reference panics in the same
// places: src/bvh.rs:86,124, src/vec3.rs:285-286) are turned into error codes here.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "errors.hpp"
#include "scenes.hpp"

using namespace rth;

struct RtHostScene {
    Flat flat;
    RtSceneDesc desc{};
    void bind() {
        desc.abi_version = RT_ABI_VERSION;
        desc.flags = flat.boxes_may_not_bound ? RT_DESC_BOXES_MAY_NOT_BOUND : 0u;
        desc.nodes = flat.nodes.data(); desc.n_nodes = flat.nodes.size();
        desc.spheres = flat.spheres.data(); desc.n_spheres = flat.spheres.size();
        desc.msphere = flat.msphere.data(); desc.n_msphere = flat.msphere.size();
        desc.rects = flat.rects.data(); desc.n_rects = flat.rects.size();
        desc.boxes = flat.boxes.data(); desc.n_boxes = flat.boxes.size();
        desc.triangles = flat.triangles.data(); desc.n_triangles = flat.triangles.size();
        desc.xforms = flat.xforms.data(); desc.n_xforms = flat.xforms.size();
        desc.media = flat.media.data(); desc.n_media = flat.media.size();
        desc.materials = flat.materials.data(); desc.n_materials = flat.materials.size();
        desc.textures = flat.textures.data(); desc.n_textures = flat.textures.size();
        desc.perlins = flat.perlins.data(); desc.n_perlins = flat.perlins.size();
        desc.images = flat.images.data(); desc.n_images = flat.images.size();
        desc.texels = flat.texels.data(); desc.n_texel_bytes = flat.texels.size();
    }
};

struct RtBuilder {
    SceneCtx ctx;
    std::vector<TexPtr> tex;
    std::vector<MatPtr> mat;
    std::vector<HitPtr> obj;
    std::vector<std::shared_ptr<HittableList>> lists;
    explicit RtBuilder(uint64_t seed) : ctx(seed) {}
};

namespace {

RtHostScene* flatten_root(const Hittable& root) {
    auto* s = new RtHostScene();
    try {
        root.flatten(s->flat);
        s->flat.seen.clear();
    } catch (...) { delete s; throw; }
    s->bind();
    return s;
}

Vec3 v3(const double* p) { return Vec3(p[0], p[1], p[2]); }

template <class V> bool ok(const V& v, int h) { return h >= 0 && (size_t)h < v.size(); }

template <class F> int guarded(F&& f) {
    try { return f(); }
    catch (const std::bad_alloc&) { return rt::set_error(RT_E_NOMEM, "out of host memory"); }
    catch (const AbiError& e) { return rt::set_error(e.code, e.what()); }
    catch (const std::exception& e) { return rt::set_error(RT_E_INVALID, e.what()); }
    catch (...) { return rt::set_error(RT_E_INVALID, "unknown error"); }
}

int add_obj(RtBuilder* b, HitPtr o) { b->obj.push_back(std::move(o)); return (int)b->obj.size() - 1; }
int add_mat(RtBuilder* b, MatPtr m) { b->mat.push_back(std::move(m)); return (int)b->mat.size() - 1; }
int add_tex(RtBuilder* b, TexPtr t) { b->tex.push_back(std::move(t)); return (int)b->tex.size() - 1; }
int bad_handle() { return rt::set_error(RT_E_INVALID, "bad handle"); }

}  // namespace

extern "C" {

int rth_scene_setup(int scene_id, RtSceneSetup* out) {
    if (!out) return rt::set_error(RT_E_INVALID, "null output");
    if (!scene_setup(scene_id, *out)) return rt::set_error(RT_E_INVALID, "unknown scene id");
    return RT_OK;
}

int rth_scene_build(int scene_id, uint64_t seed, const RtSceneOptions* opt, RtHostScene** out) {
    if (!out) return rt::set_error(RT_E_INVALID, "null output");
    return guarded([&]() -> int {
        SceneOptions o;
        if (opt) {
            o.bvh_shutter_times = (opt->flags & RT_SCENE_BVH_SHUTTER_TIMES) != 0;
            o.bvh_widest_axis = (opt->flags & RT_SCENE_BVH_WIDEST_AXIS) != 0;
            if (opt->boxes_per_side) o.boxes_per_side = opt->boxes_per_side;
            o.n_triangles = opt->n_triangles;
            o.image_rgb = opt->image_rgb; o.image_w = opt->image_width; o.image_h = opt->image_height;
        }
        SceneCtx ctx(seed);
        CtxScope scope(&ctx);
        struct TreePolicy {  // the tree policy applies to this build only, also when it throws
            TreePolicy(bool widest, bool device) { g_bvh_widest_axis = widest; g_bvh_device = device; }
            ~TreePolicy() { g_bvh_widest_axis = false; g_bvh_device = false; }
        } policy(o.bvh_widest_axis, opt && (opt->flags & RT_SCENE_BVH_LBVH) != 0);
        const bool timing = std::getenv("RT_HOST_TIMING") != nullptr;  // developer knob: where scene construction spends its time
        auto t0 = std::chrono::steady_clock::now();
        HitPtr world = build_scene(scene_id, o);
        if (!world) return rt::set_error(RT_E_INVALID, "unknown scene id");
        auto t1 = std::chrono::steady_clock::now();
        *out = flatten_root(*world);
        auto t2 = std::chrono::steady_clock::now();
        world.reset();
        auto t3 = std::chrono::steady_clock::now();
        if (timing)
            std::fprintf(stderr, "rth_scene_build: objects + tree %.3f s, flatten %.3f s, release %.3f s\n", std::chrono::duration<double>(t1 - t0).count(),
                         std::chrono::duration<double>(t2 - t1).count(), std::chrono::duration<double>(t3 - t2).count());
        return RT_OK;
    });
}

const RtSceneDesc* rth_scene_desc(const RtHostScene* s) { return s ? &s->desc : nullptr; }
void rth_scene_free(RtHostScene* s) { delete s; }

int rth_camera_new(const double lookfrom[3], const double lookat[3], const double vup[3], double vfov, double aspect_ratio,
                   double aperture, double focus_dist, double t0, double t1, RtCamera* out) {
    if (!lookfrom || !lookat || !vup || !out) return rt::set_error(RT_E_INVALID, "null argument");
    return guarded([&]() -> int {
        *out = camera_new(v3(lookfrom), v3(lookat), v3(vup), vfov, aspect_ratio, aperture, focus_dist, t0, t1);
        return RT_OK;
    });
}

int rth_builder_new(uint64_t seed, RtBuilder** out) {
    if (!out) return rt::set_error(RT_E_INVALID, "null output");
    return guarded([&]() -> int { *out = new RtBuilder(seed); return RT_OK; });
}
void rth_builder_free(RtBuilder* b) { delete b; }

#define IN_CTX(expr) guarded([&]() -> int { CtxScope scope(&b->ctx); return (expr); })

int rth_tex_solid(RtBuilder* b, const double c[3]) { return IN_CTX(add_tex(b, std::make_shared<SolidColor>(v3(c)))); }
int rth_tex_checker(RtBuilder* b, const double c1[3], const double c2[3]) { return IN_CTX(add_tex(b, std::make_shared<CheckerTexture>(v3(c1), v3(c2)))); }
int rth_tex_noise(RtBuilder* b, double scale) { return IN_CTX(add_tex(b, std::make_shared<NoiseTexture>(scale))); }
int rth_tex_image(RtBuilder* b, const uint8_t* rgb, uint32_t w, uint32_t h) {
    if (!rgb) return rt::set_error(RT_E_INVALID, "null image");
    return IN_CTX(add_tex(b, std::make_shared<ImageTexture>(rgb, w, h)));
}
int rth_mat_lambertian(RtBuilder* b, int tex) { if (!ok(b->tex, tex)) return bad_handle(); return IN_CTX(add_mat(b, std::make_shared<Lambertian>(b->tex[(size_t)tex]))); }
int rth_mat_metal(RtBuilder* b, const double albedo[3], double fuzz) { return IN_CTX(add_mat(b, std::make_shared<Metal>(v3(albedo), fuzz))); }
int rth_mat_dielectric(RtBuilder* b, double ref_idx) { return IN_CTX(add_mat(b, std::make_shared<Dielectric>(ref_idx))); }
int rth_mat_diffuse_light(RtBuilder* b, int tex) { if (!ok(b->tex, tex)) return bad_handle(); return IN_CTX(add_mat(b, std::make_shared<DiffuseLight>(b->tex[(size_t)tex]))); }
int rth_mat_isotropic(RtBuilder* b, int tex) { if (!ok(b->tex, tex)) return bad_handle(); return IN_CTX(add_mat(b, std::make_shared<Isotropic>(b->tex[(size_t)tex]))); }
int rth_sphere(RtBuilder* b, const double c[3], double r, int mat) {
    if (!ok(b->mat, mat)) return bad_handle();
    return IN_CTX(add_obj(b, std::make_shared<Sphere>(v3(c), r, b->mat[(size_t)mat])));
}
int rth_moving_sphere(RtBuilder* b, const double c0[3], const double c1[3], double t0, double t1, double r, int mat) {
    if (!ok(b->mat, mat)) return bad_handle();
    return IN_CTX(add_obj(b, std::make_shared<MovingSphere>(v3(c0), v3(c1), t0, t1, r, b->mat[(size_t)mat])));
}
int rth_xy_rect(RtBuilder* b, double x0, double x1, double y0, double y1, double k, int mat) {
    if (!ok(b->mat, mat)) return bad_handle();
    return IN_CTX(add_obj(b, XYRect(x0, x1, y0, y1, k, b->mat[(size_t)mat])));
}
int rth_xz_rect(RtBuilder* b, double x0, double x1, double z0, double z1, double k, int mat) {
    if (!ok(b->mat, mat)) return bad_handle();
    return IN_CTX(add_obj(b, XZRect(x0, x1, z0, z1, k, b->mat[(size_t)mat])));
}
int rth_yz_rect(RtBuilder* b, double y0, double y1, double z0, double z1, double k, int mat) {
    if (!ok(b->mat, mat)) return bad_handle();
    return IN_CTX(add_obj(b, YZRect(y0, y1, z0, z1, k, b->mat[(size_t)mat])));
}
int rth_box6(RtBuilder* b, const double p0[3], const double p1[3], int mat) {
    if (!ok(b->mat, mat)) return bad_handle();
    return IN_CTX(add_obj(b, std::make_shared<Box6>(v3(p0), v3(p1), b->mat[(size_t)mat])));
}
int rth_triangle(RtBuilder* b, const double p1[3], const double p2[3], const double p0[3], int mat) {
    if (!ok(b->mat, mat)) return bad_handle();
    return IN_CTX(add_obj(b, std::make_shared<Triangle>(v3(p1), v3(p2), v3(p0), b->mat[(size_t)mat])));
}
int rth_translate(RtBuilder* b, int obj, const double offset[3]) {
    if (!ok(b->obj, obj)) return bad_handle();
    return IN_CTX(add_obj(b, std::make_shared<Translate>(b->obj[(size_t)obj], v3(offset))));
}
int rth_rotate_y(RtBuilder* b, int obj, double angle) {
    if (!ok(b->obj, obj)) return bad_handle();
    return IN_CTX(add_obj(b, std::make_shared<RotateY>(b->obj[(size_t)obj], angle)));
}
int rth_constant_medium(RtBuilder* b, int boundary, double density, int tex) {
    if (!ok(b->obj, boundary) || !ok(b->tex, tex)) return bad_handle();
    return IN_CTX(add_obj(b, std::make_shared<ConstantMedium>(b->obj[(size_t)boundary], density, b->tex[(size_t)tex])));
}
int rth_list_new(RtBuilder* b) { return guarded([&]() -> int { b->lists.push_back(std::make_shared<HittableList>()); return (int)b->lists.size() - 1; }); }
int rth_list_add(RtBuilder* b, int list, int obj) {
    if (!ok(b->lists, list) || !ok(b->obj, obj)) return bad_handle();
    return guarded([&]() -> int { b->lists[(size_t)list]->add(b->obj[(size_t)obj]); return RT_OK; });
}
int rth_bvh(RtBuilder* b, int list, double time0, double time1) {
    if (!ok(b->lists, list)) return bad_handle();
    return IN_CTX(add_obj(b, BVHNode::create(*b->lists[(size_t)list], time0, time1)));
}
int rth_bvh_lbvh(RtBuilder* b, int list, double time0, double time1) {
    if (!ok(b->lists, list)) return bad_handle();
    if (b->lists[(size_t)list]->objects.size() < 2) return rt::set_error(RT_E_INVALID, "a tree built on the device needs at least two objects");
    return IN_CTX(add_obj(b, std::make_shared<DeviceBvh>(b->lists[(size_t)list]->objects, time0, time1)));
}
int rth_builder_finish(RtBuilder* b, int root, RtHostScene** out) {
    if (!out) return rt::set_error(RT_E_INVALID, "null output");
    if (!ok(b->obj, root)) return bad_handle();
    return guarded([&]() -> int { *out = flatten_root(*b->obj[(size_t)root]); return RT_OK; });
}

}  // extern "C"
