// Closest-hit query over the flattened scene: the device replacement of
// `world.hit(r, t_min, t_max, &mut rec)` (src/color.rs:41), i.e. of BVHNode::hit
// (src/bvh.rs:101-111) and every Hittable::hit it dispatches to.
//
// The entry stream (RtNode[]) is the left-then-right depth-first order of the
// reference's object tree, so one front-to-back walk with skip links visits
// objects exactly in the reference's order; the running closest t plays the role
// of the t_max the reference threads through its recursion. Only (t, entry, side,
// innermost wrapper) of the winning candidate are kept; the full HitRecord is
// rebuilt once per ray in hitrecord.h.
#pragma once
#include "../../../include/rt/abi.h"
#include "../rng.h"
#include "vec.h"

namespace rtd {

constexpr double D_INF = __builtin_huge_val();

struct DevScene {
    const RtNode* nodes;
    const RtSphere* spheres;
    const RtMovingSphere* msphere;
    const RtRect* rects;
    const RtBox6* boxes;
    const RtTriangle* triangles;
    const RtXform* xforms;
    const RtMedium* media;
    const RtMaterial* materials;
    const RtTexture* textures;
    const RtPerlin* perlins;
    const RtImage* images;
    const uint8_t* texels;
    uint32_t n_nodes;
};

struct RayKey {  // identifies the path for the counter-based RNG
    uint64_t seed;
    uint32_t pixel, sample, bounce;
};

// Winning candidate of a query. ref = meta word of the stream entry that produced it
// (kind | table index << 4) or RT_MISS; aux = box side | (xform table index + 1) << 8.
struct Cand {
    double t;
    uint32_t ref;
    uint32_t aux;
};

// Work counters of one thread (only in the instrumented kernels).
struct Work {
    uint32_t entries;   // stream entries fetched (32 B each)
    uint32_t prim[9];   // primitive tests by RtNodeKind (RT_N_INNER slot counts box tests)
};

// ---- wrappers: ray into object space -------------------------------------------
// Translate::hit src/hittable.rs:278-282; RotateY::hit :355-368.
RT_D void xf_apply(const RtXform* x, V3& o, V3& d) {
    if (x->kind == 0u) {
        o = o - v3p(x->offset);
    } else {
        const double c = x->cos_theta, s = x->sin_theta;
        const double ox = c * o.x - s * o.z;
        const double oz = s * o.x + c * o.z;
        const double dx = c * d.x - s * d.z;
        const double dz = s * d.x + c * d.z;
        o.x = ox; o.z = oz; d.x = dx; d.z = dz;
    }
}

RT_D V3 inv_dir(V3 d) { return v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z); }  // src/aabb.rs:17

// ---- box test -------------------------------------------------------------------
// The expressions of src/aabb.rs:17-25 on the f32 box, with tmin/tmax carried from
// axis to axis (the reference restarts from the caller's tmin/tmax on every axis,
// which only makes its test looser). tmin never decreases and tmax never increases
// and neither can become NaN, so one comparison at the end equals the per-axis exit.
RT_D bool box_hit(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 o, V3 invd, double tmin, double tmax) {
    double t0 = ((double)lox - o.x) * invd.x;
    double t1 = ((double)hix - o.x) * invd.x;
    if (invd.x < 0.0) { const double s = t0; t0 = t1; t1 = s; }
    tmin = t0 > tmin ? t0 : tmin;
    tmax = t1 < tmax ? t1 : tmax;
    t0 = ((double)loy - o.y) * invd.y;
    t1 = ((double)hiy - o.y) * invd.y;
    if (invd.y < 0.0) { const double s = t0; t0 = t1; t1 = s; }
    tmin = t0 > tmin ? t0 : tmin;
    tmax = t1 < tmax ? t1 : tmax;
    t0 = ((double)loz - o.z) * invd.z;
    t1 = ((double)hiz - o.z) * invd.z;
    if (invd.z < 0.0) { const double s = t0; t0 = t1; t1 = s; }
    tmin = t0 > tmin ? t0 : tmin;
    tmax = t1 < tmax ? t1 : tmax;
    return !(tmax <= tmin);
}

// ---- primitives -----------------------------------------------------------------
// Sphere::hit src/hittable.rs:72-113 (strict t_min < t < t_max).
RT_D bool sphere_roots(V3 oc, V3 d, double r, double tmin, double tmax, double& t_out) {
    const double a = squared_length(d);
    const double half_b = dot(oc, d);
    const double c = squared_length(oc) - r * r;
    const double discriminant = half_b * half_b - a * c;
    if (discriminant > 0.0) {
        const double root = sqrt(discriminant);
        double temp = (-half_b - root) / a;
        if (temp < tmax && temp > tmin) { t_out = temp; return true; }
        temp = (-half_b + root) / a;
        if (temp < tmax && temp > tmin) { t_out = temp; return true; }
    }
    return false;
}

// MovingSphere::center src/hittable.rs:208-211.
RT_D V3 msphere_center(const RtMovingSphere* s, double time) {
    const V3 c0 = v3p(s->c0), c1 = v3p(s->c1);
    return c0 + (c1 - c0) * ((time - s->t0) / (s->t1 - s->t0));
}

// XYRect/XZRect/YZRect::hit src/aarect.rs:30-39, 81-90, 132-141: plane, then inclusive bounds.
// (ok, dk) = ray origin/direction along the plane normal, (oa, da), (ob, db) in the plane.
RT_D bool rect_t(double k, double a0, double a1, double b0, double b1, double ok, double dk, double oa, double da, double ob,
                 double db, double tmin, double tmax, double& t_out) {
    const double t = (k - ok) / dk;
    if (t < tmin || t > tmax) return false;
    const double a = oa + t * da;
    const double b = ob + t * db;
    if (a < a0 || a > a1 || b < b0 || b > b1) return false;
    t_out = t;
    return true;
}

RT_D bool rect_hit(const RtRect* r, V3 o, V3 d, double tmin, double tmax, double& t_out) {
    const uint32_t axis = r->axis;
    double ok, dk, oa, da, ob, db;
    if (axis == 0u) { ok = o.z; dk = d.z; oa = o.x; da = d.x; ob = o.y; db = d.y; }
    else if (axis == 1u) { ok = o.y; dk = d.y; oa = o.x; da = d.x; ob = o.z; db = d.z; }
    else { ok = o.x; dk = d.x; oa = o.y; da = d.y; ob = o.z; db = d.z; }
    return rect_t(r->k, r->a0, r->a1, r->b0, r->b1, ok, dk, oa, da, ob, db, tmin, tmax, t_out);
}

// Box6::hit -> HittableList::hit over the six sides in src/box6.rs:22-71 order, each against the
// running closest t (src/hittable.rs:151-164). Returns the side index of the last accepted side.
RT_D bool box6_hit(const RtBox6* b, V3 o, V3 d, double tmin, double& tmax, uint32_t& side_out) {
    const V3 p0 = v3p(b->p0), p1 = v3p(b->p1);
    bool any = false;
    double t;
    if (rect_t(p1.z, p0.x, p1.x, p0.y, p1.y, o.z, d.z, o.x, d.x, o.y, d.y, tmin, tmax, t)) { tmax = t; side_out = 0u; any = true; }
    if (rect_t(p0.z, p0.x, p1.x, p0.y, p1.y, o.z, d.z, o.x, d.x, o.y, d.y, tmin, tmax, t)) { tmax = t; side_out = 1u; any = true; }
    if (rect_t(p1.y, p0.x, p1.x, p0.z, p1.z, o.y, d.y, o.x, d.x, o.z, d.z, tmin, tmax, t)) { tmax = t; side_out = 2u; any = true; }
    if (rect_t(p0.y, p0.x, p1.x, p0.z, p1.z, o.y, d.y, o.x, d.x, o.z, d.z, tmin, tmax, t)) { tmax = t; side_out = 3u; any = true; }
    if (rect_t(p1.x, p0.y, p1.y, p0.z, p1.z, o.x, d.x, o.y, d.y, o.z, d.z, tmin, tmax, t)) { tmax = t; side_out = 4u; any = true; }
    if (rect_t(p0.x, p0.y, p1.y, p0.z, p1.z, o.x, d.x, o.y, d.y, o.z, d.z, tmin, tmax, t)) { tmax = t; side_out = 5u; any = true; }
    return any;
}

// Triangle::hit src/triangle.rs:26-59 (Moeller-Trumbore, two-sided, t range inclusive).
RT_D bool triangle_hit(const RtTriangle* tr, V3 o, V3 d, double tmin, double tmax, double& t_out) {
    const V3 point0 = v3p(tr->p0), edge1 = v3p(tr->e1), edge2 = v3p(tr->e2);
    const V3 pvec = cross(d, edge2);
    const double det = dot(pvec, edge1);
    const double EPSILON = 1e-8;
    if (det < EPSILON && det > -EPSILON) return false;
    const double inv_det = 1.0 / det;
    const V3 tvec = o - point0;
    const double u = dot(tvec, pvec) * inv_det;
    if (u < 0.0 || u > 1.0) return false;
    const V3 qvec = cross(tvec, edge1);
    const double v = dot(d, qvec) * inv_det;
    if (v < 0.0 || u + v > 1.0) return false;
    const double t = dot(edge2, qvec) * inv_det;
    if (t < tmin || t > tmax) return false;
    t_out = t;
    return true;
}

// ---- the walk ---------------------------------------------------------------------
// TOP = false is the boundary query of a ConstantMedium: same walk over the boundary's entries,
// without media inside (the flattener rejects nested media).
//   (bo, bd)   ray in the space the walk starts in, base_depth wrappers deep, innermost base_xf
//   [begin,end) entries to walk
template <bool TOP, bool COUNT>
RT_D void closest_hit(const DevScene& S, V3 bo, V3 bd, double tm, double tmin, double tmax, uint32_t begin, uint32_t end,
                      uint32_t base_depth, uint32_t base_xf, const RayKey& key, Cand& best, Work* work) {
    V3 o = bo, d = bd;
    V3 invd = inv_dir(d);
    uint32_t cur_xf = base_xf;  // xform table index + 1, 0 = none
    best.t = tmax;
    best.ref = RT_MISS;
    best.aux = 0u;
    uint32_t i = begin;
    while (i < end) {
        const uint4* np = reinterpret_cast<const uint4*>(S.nodes + i);
        const uint4 na = np[0];
        const uint4 nb = np[1];
        const uint32_t meta = nb.z;
        const uint32_t kind = meta & 15u;
        const uint32_t idx = meta >> 4;
        if (COUNT) { work->entries++; work->prim[kind]++; }
        if (kind == RT_N_INNER) {
            const bool pass = box_hit(__uint_as_float(na.x), __uint_as_float(na.y), __uint_as_float(na.z), __uint_as_float(na.w),
                                      __uint_as_float(nb.x), __uint_as_float(nb.y), o, invd, tmin, best.t);
            i = pass ? i + 1u : nb.w;
            continue;
        }
        switch (kind) {
            case RT_N_SPHERE: {
                const RtSphere* s = S.spheres + idx;
                double t;
                if (sphere_roots(o - v3p(s->c), d, s->r, tmin, best.t, t)) { best.t = t; best.ref = meta; best.aux = cur_xf << 8; }
                break;
            }
            case RT_N_MSPHERE: {
                const RtMovingSphere* s = S.msphere + idx;
                double t;
                if (sphere_roots(o - msphere_center(s, tm), d, s->r, tmin, best.t, t)) { best.t = t; best.ref = meta; best.aux = cur_xf << 8; }
                break;
            }
            case RT_N_RECT: {
                double t;
                if (rect_hit(S.rects + idx, o, d, tmin, best.t, t)) { best.t = t; best.ref = meta; best.aux = cur_xf << 8; }
                break;
            }
            case RT_N_BOX6: {
                uint32_t side = 0u;
                double t = best.t;
                if (box6_hit(S.boxes + idx, o, d, tmin, t, side)) { best.t = t; best.ref = meta; best.aux = side | (cur_xf << 8); }
                break;
            }
            case RT_N_TRIANGLE: {
                double t;
                if (triangle_hit(S.triangles + idx, o, d, tmin, best.t, t)) { best.t = t; best.ref = meta; best.aux = cur_xf << 8; }
                break;
            }
            case RT_N_XF_ENTER: {
                xf_apply(S.xforms + idx, o, d);
                invd = inv_dir(d);
                cur_xf = idx + 1u;
                break;
            }
            case RT_N_XF_EXIT: {
                // back to the parent's space: replay the parent's chain on the base ray (same operations in
                // the same order as on the way in, hence the same bits)
                const RtXform* x = S.xforms + idx;
                const uint32_t depth = x->depth;
                o = bo; d = bd;
                for (uint32_t k = base_depth; k + 1u < depth; ++k) xf_apply(S.xforms + x->anc[k], o, d);
                invd = inv_dir(d);
                cur_xf = (depth - 1u > base_depth) ? x->anc[depth - 2u] + 1u : base_xf;
                break;
            }
            case RT_N_MEDIUM: {
                if (TOP) {
                    // ConstantMedium::hit src/constant_medium.rs:34-89
                    const RtMedium* m = S.media + idx;
                    const uint32_t depth_here = cur_xf ? S.xforms[cur_xf - 1u].depth : 0u;
                    Cand c1, c2;
                    closest_hit<false, COUNT>(S, o, d, tm, -D_INF, D_INF, i + 1u, nb.w, depth_here, cur_xf, key, c1, work);
                    if (c1.ref != RT_MISS) {
                        closest_hit<false, COUNT>(S, o, d, tm, c1.t + 0.0001, D_INF, i + 1u, nb.w, depth_here, cur_xf, key, c2, work);
                        if (c2.ref != RT_MISS) {
                            double t1 = c1.t, t2 = c2.t;
                            if (t1 < tmin) t1 = tmin;
                            if (t2 > best.t) t2 = best.t;
                            if (!(t1 > t2)) {
                                if (t1 < 0.0) t1 = 0.0;
                                const double ray_length = length(d);
                                const double distance_inside_boundary = (t2 - t1) * ray_length;
                                rt::Rng rng;
                                rng.set_path(key.seed, key.pixel, key.sample, key.bounce, rt::RNG_MEDIUM0 + m->rng_id);
                                const double hit_distance = m->neg_inv_density * log10(rng.uniform(0.0, 1.0));  // :69
                                if (!(hit_distance > distance_inside_boundary)) {
                                    best.t = t1 + hit_distance / ray_length;
                                    best.ref = meta;
                                    best.aux = cur_xf << 8;
                                }
                            }
                        }
                    }
                }
                i = nb.w;
                continue;
            }
            default: break;
        }
        ++i;
    }
}

}  // namespace rtd
