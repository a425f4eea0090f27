// TEST INFRASTRUCTURE ONLY — C entry points of the CPU oracle (liboracle.so),
// loaded through ctypes by tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg. The product library never links or calls this.
#include <chrono>
#include <cstring>
#include <string>
#include <thread>

#include "../include/rt/host.h"
#include "oracle.hpp"
#include "scenes.hpp"

using namespace orc;

struct OrcScene {
    HitPtr world;
    SceneCtx ctx;
    uint64_t seed = 0;
};

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

Camera to_camera(const RtCamera& c) {
    Camera k;
    k.orig = Point(c.orig[0], c.orig[1], c.orig[2]);
    k.lower_left_corner = Point(c.lower_left_corner[0], c.lower_left_corner[1], c.lower_left_corner[2]);
    k.horizontal = Vec3(c.horizontal[0], c.horizontal[1], c.horizontal[2]);
    k.vertical = Vec3(c.vertical[0], c.vertical[1], c.vertical[2]);
    k.u = Vec3(c.u[0], c.u[1], c.u[2]);
    k.v = Vec3(c.v[0], c.v[1], c.v[2]);
    k.lens_radius = c.lens_radius; k.time0 = c.time0; k.time1 = c.time1;
    return k;
}
void put(double* d, const Vec3& v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

void reset_stats() {
    tls.n_box = 0; tls.n_rays = 0;
    for (auto& c : tls.n_prim) c = 0;
}
void add_stats(uint64_t* s) {  // s[0]=box tests, s[1..6]=primitive tests by kind, s[7]=rays
    s[0] += tls.n_box;
    for (int k = 0; k < K_NKINDS; ++k) s[1 + k] += tls.n_prim[k];
    s[7] += tls.n_rays;
}

}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }

double orc_rand_unit(uint64_t seed, uint32_t a, uint32_t b, uint32_t c, uint64_t k) {
    RngCtx r; r.seed = seed; r.a = a; r.b = b; r.c = c; r.k = k;
    return r.next_unit();
}

// Vec3 operators for the known-answer tests of src/vec3.rs:302-440.
// op: 0 add, 1 add scalar, 2 sub, 3 sub scalar, 4 dot (out[0]), 5 elemul (MulAssign), 6 mul scalar, 7 div scalar,
//     8 elemul, 9 cross, 10 neg, 11 squared_length (out[0]), 12 length (out[0]), 13 unit (returns -1 on zero length)
int orc_vec3_op(int op, const double* a, const double* b, double s, double* out) {
    Vec3 A(a[0], a[1], a[2]), B = b ? Vec3(b[0], b[1], b[2]) : Vec3();
    switch (op) {
        case 0: put(out, A + B); return 0;
        case 1: put(out, A + s); return 0;
        case 2: put(out, A - B); return 0;
        case 3: put(out, A - s); return 0;
        case 4: out[0] = A * B; return 0;
        case 5: { Vec3 t = A; t *= B; put(out, t); return 0; }
        case 6: put(out, A * s); return 0;
        case 7: put(out, A / s); return 0;
        case 8: put(out, Vec3::elemul(A, B)); return 0;
        case 9: put(out, Vec3::cross(A, B)); return 0;
        case 10: put(out, -A); return 0;
        case 11: out[0] = A.squared_length(); return 0;
        case 12: out[0] = A.length(); return 0;
        case 13:
            try { put(out, A.unit()); } catch (const std::exception&) { return -1; }
            return 0;
    }
    return -2;
}

int orc_resolve_u8(const double* accum, uint32_t w, uint32_t h, uint64_t spp, uint8_t* out) {
    for (uint64_t i = 0; i < (uint64_t)w * h; ++i) resolve_pixel(accum + 3 * i, spp, out + 3 * i);
    return 0;
}

int orc_scene_setup(int scene_id, RtSceneSetup* out) {
    SceneSetup s;
    if (!scene_setup(scene_id, s)) return fail(RT_E_INVALID, "unknown scene id");
    put(out->lookfrom, s.lookfrom); put(out->lookat, s.lookat); put(out->vup, s.vup);
    out->vfov = s.vfov; out->aspect_ratio = s.aspect_ratio; out->aperture = s.aperture; out->focus_dist = s.focus_dist;
    out->time0 = s.time0; out->time1 = s.time1; put(out->background, s.background);
    out->samples_per_pixel = s.samples_per_pixel; out->width = s.width;
    return 0;
}

int orc_camera_new(const double lookfrom[3], const double lookat[3], const double vup[3], double vfov, double aspect_ratio,
                   double aperture, double focus_dist, double t0, double t1, RtCamera* out) {
    try {
        Camera c(Point(lookfrom[0], lookfrom[1], lookfrom[2]), Point(lookat[0], lookat[1], lookat[2]),
                 Vec3(vup[0], vup[1], vup[2]), CameraInfo{vfov, aspect_ratio, aperture, focus_dist}, t0, t1);
        put(out->orig, c.orig); put(out->lower_left_corner, c.lower_left_corner); put(out->horizontal, c.horizontal);
        put(out->vertical, c.vertical); put(out->u, c.u); put(out->v, c.v);
        out->lens_radius = c.lens_radius; out->time0 = c.time0; out->time1 = c.time1;
    } catch (const std::exception& e) { return fail(RT_E_INVALID, e.what()); }
    return 0;
}

int orc_scene_build(int scene_id, uint64_t seed, const RtSceneOptions* opt, OrcScene** out) {
    auto* s = new OrcScene();
    s->seed = seed;
    SceneOptions o;
    if (opt) {
        o.bvh_shutter_times = (opt->flags & RT_SCENE_BVH_SHUTTER_TIMES) != 0;
        o.bvh_widest_axis = (opt->flags & RT_SCENE_BVH_WIDEST_AXIS) != 0;
        if (opt->boxes_per_side) o.boxes_per_side = opt->boxes_per_side;
        o.n_triangles = opt->n_triangles;
        o.image_rgb = opt->image_rgb; o.image_w = opt->image_width; o.image_h = opt->image_height;
    }
    cur_ctx = &s->ctx;
    tls.rng.set_scene(seed);
    bvh_widest_axis = o.bvh_widest_axis;
    bvh_lbvh = opt && (opt->flags & RT_SCENE_BVH_LBVH) != 0;
    try {
        s->world = build_scene(scene_id, o);
        if (s->world) s->world = std::make_shared<World>(s->world);
    } catch (const std::exception& e) {
        cur_ctx = nullptr; bvh_widest_axis = false; bvh_lbvh = false; delete s;
        return fail(RT_E_INVALID, e.what());
    }
    bvh_widest_axis = false;
    bvh_lbvh = false;
    cur_ctx = nullptr;
    if (!s->world) { delete s; return fail(RT_E_INVALID, "unknown scene id"); }
    *out = s;
    return 0;
}
void orc_scene_free(OrcScene* s) { delete s; }
uint32_t orc_scene_prim_count(const OrcScene* s) { return s->ctx.next_prim; }

// world.hit for n rays (src/color.rs:41). rec_out may be NULL.
int orc_trace(OrcScene* s, const RtRay* rays, uint64_t n, uint64_t seed, int aabb_mode, RtHit* out, RtHitRecord* rec_out,
              uint64_t* stats) {
    tls.aabb_mode = aabb_mode; tls.carried_t0 = s->ctx.carried_t0; tls.carried_t1 = s->ctx.carried_t1;
    tls.rng.seed = seed;
    reset_stats();
    for (uint64_t i = 0; i < n; ++i) {
        const RtRay& q = rays[i];
        Ray r{Point(q.o[0], q.o[1], q.o[2]), Vec3(q.d[0], q.d[1], q.d[2]), q.tm};
        tls.pixel = q.pixel; tls.sample = q.sample; tls.bounce = q.bounce;
        HitRecord rec;
        bool hit = s->world->hit(r, q.tmin, q.tmax, rec);
        RtHit h{RT_MISS, 0, 0.0, 0.0, 0.0};
        if (hit) {
            h.prim_id = rec.prim_id;
            h.sub = rec.side | ((rec.inst_id + 1u) << 8);
            h.t = rec.t; h.u = rec.u; h.v = rec.v;
        }
        out[i] = h;
        if (rec_out) {
            RtHitRecord& R = rec_out[i];
            std::memset(&R, 0, sizeof R);
            R.prim_id = h.prim_id; R.sub = h.sub;
            if (hit) {
                put(R.p, rec.p); put(R.normal, rec.normal);
                R.t = rec.t; R.u = rec.u; R.v = rec.v;
                R.mat = rec.mat_ptr ? rec.mat_ptr->mat_id : RT_NONE;
                R.front_face = rec.front_face ? 1u : 0u;
            }
        }
    }
    if (stats) { std::memset(stats, 0, 8 * sizeof(uint64_t)); add_stats(stats); }
    return 0;
}

// The sample loop of src/main.rs:193-230: `threads` threads, thread i owns columns [i*W/T, (i+1)*W/T).
// accum_rgb (W*H*3 doubles) receives += the sum of ray_color over samples [sample_begin, sample_end).
int orc_render(OrcScene* s, const RtCamera* cam_in, const RtRenderParams* p, int aabb_mode, int threads, double* accum_rgb,
               uint64_t* stats, double* seconds) {
    if (threads < 1) threads = 1;
    const Camera cam = to_camera(*cam_in);
    const Color background(p->background[0], p->background[1], p->background[2]);
    const uint32_t W = p->width, H = p->height;
    std::vector<std::thread> pool;
    std::vector<std::vector<uint64_t>> tstats((size_t)threads, std::vector<uint64_t>(8, 0));
    auto t_begin = std::chrono::steady_clock::now();
    for (int i = 0; i < threads; ++i) {
        const uint32_t start = (uint32_t)((uint64_t)i * W / (uint64_t)threads);
        const uint32_t end = (uint32_t)((uint64_t)(i + 1) * W / (uint64_t)threads);
        pool.emplace_back([=, &tstats]() {
            tls.aabb_mode = aabb_mode; tls.carried_t0 = s->ctx.carried_t0; tls.carried_t1 = s->ctx.carried_t1;
            tls.rng.seed = p->seed;
            tls.recorder = nullptr;
            reset_stats();
            for (uint32_t x = start; x < end; ++x)
                for (uint32_t y = 0; y < H; ++y) {
                    Color pixel_color(0.0, 0.0, 0.0);
                    for (uint32_t sm = p->sample_begin; sm < p->sample_end; ++sm)
                        pixel_color += sample_pixel(cam, *s->world, background, W, H, x, y, sm, (int)p->max_depth);
                    double* a = accum_rgb + ((size_t)y * W + x) * 3;
                    a[0] += pixel_color.x; a[1] += pixel_color.y; a[2] += pixel_color.z;
                }
            add_stats(tstats[(size_t)i].data());
        });
    }
    for (auto& t : pool) t.join();
    auto t_end = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t_end - t_begin).count();
    if (stats) {
        std::memset(stats, 0, 8 * sizeof(uint64_t));
        for (auto& ts : tstats) for (int k = 0; k < 8; ++k) stats[k] += ts[(size_t)k];
    }
    return 0;
}

// One iteration of ray_color (src/color.rs:41-62) per ray: the checker of rt_path_step. The ray's (pixel, sample,
// bounce) key the medium draw inside world.hit and the scatter stream, as in ray_color above.
int orc_path_step(OrcScene* s, const RtRenderParams* p, int aabb_mode, const RtRay* rays, uint64_t n, RtPathStep* out) {
    tls.aabb_mode = aabb_mode; tls.carried_t0 = s->ctx.carried_t0; tls.carried_t1 = s->ctx.carried_t1;
    tls.rng.seed = p->seed;
    const Color background(p->background[0], p->background[1], p->background[2]);
    for (uint64_t i = 0; i < n; ++i) {
        const RtRay& q = rays[i];
        Ray r{Point(q.o[0], q.o[1], q.o[2]), Vec3(q.d[0], q.d[1], q.d[2]), q.tm};
        tls.pixel = q.pixel; tls.sample = q.sample; tls.bounce = q.bounce;
        RtPathStep& o = out[i];
        std::memset(&o, 0, sizeof o);
        o.prim_id = RT_MISS;
        HitRecord rec;
        if (!s->world->hit(r, 0.0001, INF, rec)) {  // :41-43
            put(o.emitted, background);
            continue;
        }
        o.prim_id = rec.prim_id;
        o.sub = rec.side | ((rec.inst_id + 1u) << 8);
        o.t = rec.t;
        Ray scattered{Vec3::ones(), Vec3::ones(), 0.0};
        Color attenuation(0.0, 0.0, 0.0);
        put(o.emitted, rec.mat_ptr->emitted(rec.u, rec.v, rec.p));  // :50
        tls.rng.set(tls.pixel, tls.sample, tls.bounce, STREAM_SCATTER);
        if (!rec.mat_ptr->scatter(r, rec, attenuation, scattered)) continue;  // :52-57
        if ((int)q.bounce + 1 >= (int)p->max_depth) continue;  // ray_color(depth - 1) with depth - 1 == 0 is black (:37-39)
        o.scattered = 1u;
        put(o.attenuation, attenuation);
        put(o.o, scattered.orig); put(o.d, scattered.dire);
    }
    return 0;
}

// Traces n_paths path samples spread over the image and records every ray handed to world.hit at
// bounces <= max_bounce (primary rays and their continuations), up to max_rays.
int orc_harvest(OrcScene* s, const RtCamera* cam_in, const RtRenderParams* p, int aabb_mode, uint32_t max_bounce,
                uint64_t n_paths, uint64_t max_rays, RtRay* out, uint64_t* n_out) {
    const Camera cam = to_camera(*cam_in);
    const Color background(p->background[0], p->background[1], p->background[2]);
    const uint64_t npix = (uint64_t)p->width * p->height;
    std::vector<RayRecord> rec;
    tls.aabb_mode = aabb_mode; tls.carried_t0 = s->ctx.carried_t0; tls.carried_t1 = s->ctx.carried_t1;
    tls.rng.seed = p->seed;
    tls.recorder = &rec;
    tls.record_max_bounce = max_bounce;
    uint64_t n = 0;
    for (uint64_t i = 0; i < n_paths && n < max_rays; ++i) {
        const uint64_t pix = (i * 7919u + 13u) % npix;
        const uint32_t sm = p->sample_begin + (uint32_t)(i / npix);
        rec.clear();
        sample_pixel(cam, *s->world, background, p->width, p->height, (uint32_t)(pix % p->width), (uint32_t)(pix / p->width), sm,
                     (int)p->max_depth);
        for (const auto& rr : rec) {
            if (n >= max_rays) break;
            RtRay& q = out[n++];
            put(q.o, rr.r.orig); put(q.d, rr.r.dire); q.tm = rr.r.tm; q.tmin = 0.0001; q.tmax = INF;
            q.pixel = rr.pixel; q.sample = rr.sample; q.bounce = rr.bounce; q._pad = 0;
        }
    }
    tls.recorder = nullptr;
    *n_out = n;
    return 0;
}

// ---- builder: one call per reference constructor ---------------------------
struct OrcBuilder {
    SceneCtx ctx;
    uint64_t seed;
    RngCtx rng;  // the scene stream, saved between calls
    std::vector<TexPtr> tex;
    std::vector<MatPtr> mat;
    std::vector<HitPtr> obj;
    std::vector<std::shared_ptr<HittableList>> lists;
};

#define B_ENTER cur_ctx = &b->ctx; tls.rng = b->rng; try {
#define B_LEAVE } catch (const std::exception& e) { b->rng = tls.rng; cur_ctx = nullptr; return fail(RT_E_INVALID, e.what()); } \
    b->rng = tls.rng; cur_ctx = nullptr;
#define CHK(v, h) if ((h) < 0 || (size_t)(h) >= (v).size()) return fail(RT_E_INVALID, "bad handle");

int orc_builder_new(uint64_t seed, OrcBuilder** out) {
    auto* b = new OrcBuilder();
    b->seed = seed; b->rng.set_scene(seed);
    *out = b;
    return 0;
}
void orc_builder_free(OrcBuilder* b) { delete b; }
static Vec3 v3(const double* p) { return Vec3(p[0], p[1], p[2]); }
static int push_mat(OrcBuilder* b, MatPtr m) { m->mat_id = (uint32_t)b->mat.size(); b->mat.push_back(std::move(m)); return (int)b->mat.size() - 1; }
static int push_obj(OrcBuilder* b, HitPtr o) { b->obj.push_back(std::move(o)); return (int)b->obj.size() - 1; }

int orc_tex_solid(OrcBuilder* b, const double c[3]) { b->tex.push_back(mk<SolidColor>(v3(c))); return (int)b->tex.size() - 1; }
int orc_tex_checker(OrcBuilder* b, const double c1[3], const double c2[3]) { b->tex.push_back(mk<CheckerTexture>(v3(c1), v3(c2))); return (int)b->tex.size() - 1; }
int orc_tex_noise(OrcBuilder* b, double scale) {
    B_ENTER b->tex.push_back(mk<NoiseTexture>(scale)); B_LEAVE
    return (int)b->tex.size() - 1;
}
int orc_tex_image(OrcBuilder* b, const uint8_t* rgb, uint32_t w, uint32_t h) { b->tex.push_back(mk<ImageTexture>(w, h, rgb)); return (int)b->tex.size() - 1; }
int orc_mat_lambertian(OrcBuilder* b, int tex) { CHK(b->tex, tex) return push_mat(b, mk<Lambertian>(b->tex[(size_t)tex])); }
int orc_mat_metal(OrcBuilder* b, const double albedo[3], double fuzz) { return push_mat(b, mk<Metal>(v3(albedo), fuzz)); }
int orc_mat_dielectric(OrcBuilder* b, double ref_idx) { return push_mat(b, mk<Dielectric>(ref_idx)); }
int orc_mat_diffuse_light(OrcBuilder* b, int tex) { CHK(b->tex, tex) return push_mat(b, mk<DiffuseLight>(b->tex[(size_t)tex])); }
int orc_mat_isotropic(OrcBuilder* b, int tex) { CHK(b->tex, tex) return push_mat(b, mk<Isotropic>(b->tex[(size_t)tex])); }
int orc_sphere(OrcBuilder* b, const double c[3], double r, int mat) {
    CHK(b->mat, mat) B_ENTER push_obj(b, mk<Sphere>(v3(c), r, b->mat[(size_t)mat])); B_LEAVE return (int)b->obj.size() - 1;
}
int orc_moving_sphere(OrcBuilder* b, const double c0[3], const double c1[3], double t0, double t1, double r, int mat) {
    CHK(b->mat, mat) B_ENTER push_obj(b, mk<MovingSphere>(v3(c0), v3(c1), t0, t1, r, b->mat[(size_t)mat])); B_LEAVE return (int)b->obj.size() - 1;
}
int orc_xy_rect(OrcBuilder* b, double x0, double x1, double y0, double y1, double k, int mat) {
    CHK(b->mat, mat) B_ENTER push_obj(b, XYRect(x0, x1, y0, y1, k, b->mat[(size_t)mat])); B_LEAVE return (int)b->obj.size() - 1;
}
int orc_xz_rect(OrcBuilder* b, double x0, double x1, double z0, double z1, double k, int mat) {
    CHK(b->mat, mat) B_ENTER push_obj(b, XZRect(x0, x1, z0, z1, k, b->mat[(size_t)mat])); B_LEAVE return (int)b->obj.size() - 1;
}
int orc_yz_rect(OrcBuilder* b, double y0, double y1, double z0, double z1, double k, int mat) {
    CHK(b->mat, mat) B_ENTER push_obj(b, YZRect(y0, y1, z0, z1, k, b->mat[(size_t)mat])); B_LEAVE return (int)b->obj.size() - 1;
}
int orc_box6(OrcBuilder* b, const double p0[3], const double p1[3], int mat) {
    CHK(b->mat, mat) B_ENTER push_obj(b, mk<Box6>(v3(p0), v3(p1), b->mat[(size_t)mat])); B_LEAVE return (int)b->obj.size() - 1;
}
int orc_triangle(OrcBuilder* b, const double p1[3], const double p2[3], const double p0[3], int mat) {
    CHK(b->mat, mat) B_ENTER push_obj(b, mk<Triangle>(v3(p1), v3(p2), v3(p0), b->mat[(size_t)mat])); B_LEAVE return (int)b->obj.size() - 1;
}
int orc_translate(OrcBuilder* b, int obj, const double offset[3]) {
    CHK(b->obj, obj) B_ENTER push_obj(b, mk<Translate>(b->obj[(size_t)obj], v3(offset))); B_LEAVE return (int)b->obj.size() - 1;
}
int orc_rotate_y(OrcBuilder* b, int obj, double angle) {
    CHK(b->obj, obj) B_ENTER push_obj(b, mk<RotateY>(b->obj[(size_t)obj], angle)); B_LEAVE return (int)b->obj.size() - 1;
}
int orc_constant_medium(OrcBuilder* b, int boundary, double density, int tex) {
    CHK(b->obj, boundary) CHK(b->tex, tex)
    B_ENTER
    push_obj(b, mk<ConstantMedium>(b->obj[(size_t)boundary], density, b->tex[(size_t)tex]));
    B_LEAVE return (int)b->obj.size() - 1;
}
int orc_list_new(OrcBuilder* b) { b->lists.push_back(std::make_shared<HittableList>()); return (int)b->lists.size() - 1; }
int orc_list_add(OrcBuilder* b, int list, int obj) {
    CHK(b->lists, list) CHK(b->obj, obj) b->lists[(size_t)list]->add(b->obj[(size_t)obj]); return 0;
}
int orc_bvh(OrcBuilder* b, int list, double time0, double time1) {
    CHK(b->lists, list) B_ENTER push_obj(b, BVHNode_new(*b->lists[(size_t)list], time0, time1)); B_LEAVE return (int)b->obj.size() - 1;
}
int orc_bvh_lbvh(OrcBuilder* b, int list, double time0, double time1) {  // the linear BVH of lbvh.hpp over the list
    CHK(b->lists, list)
    B_ENTER
    if (b->lists[(size_t)list]->objects.size() < 2) throw std::runtime_error("a linear BVH needs at least two objects");
    push_obj(b, lbvh_from_list(b->lists[(size_t)list]->objects, time0, time1));
    B_LEAVE return (int)b->obj.size() - 1;
}
// the checker for rt_bvh_build: same arguments, same tokens
int orc_lbvh_tokens(const double* boxes, uint64_t n, RtNode* tokens_out) {
    if (!boxes || !tokens_out) return fail(RT_E_INVALID, "null argument");
    try {
        const std::vector<RtNode> t = orc_lbvh::build_tokens(boxes, (size_t)n);
        std::memcpy(tokens_out, t.data(), t.size() * sizeof(RtNode));
    } catch (const std::exception& e) { return fail(RT_E_INVALID, e.what()); }
    return 0;
}
int orc_builder_finish(OrcBuilder* b, int root, OrcScene** out) {
    CHK(b->obj, root)
    if (!dynamic_cast<BVHNode*>(b->obj[(size_t)root].get())) return fail(RT_E_INVALID, "root must be a BVHNode");
    auto* s = new OrcScene();
    s->world = std::make_shared<World>(b->obj[(size_t)root]); s->ctx = b->ctx; s->seed = b->seed;
    *out = s;
    return 0;
}

}  // extern "C"
