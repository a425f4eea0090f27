// Rebuilds the reference's HitRecord (src/hittable.rs:11-19) for the winning
// candidate of a closest-hit query: the point, normal, front_face, u, v and
// material that ray_color (src/color.rs:41-54) reads after world.hit returns.
// The steps are those the reference performs on its way out of the recursion:
// the primitive fills the record in its own space, then every enclosing
// Translate / RotateY, innermost first, moves p and the normal back and calls
// set_face_normal again with ITS ray (src/hittable.rs:287-289, 374-384).
#pragma once
#include "traverse.h"

namespace rtd {

constexpr double D_PI = 3.14159265358979323846264338327950288;  // src/rtweekend.rs:4

struct HitRec {
    V3 p, normal;
    double t, u, v;
    uint32_t mat;
    bool front_face;
};

// HitRecord::set_face_normal src/hittable.rs:33-40
RT_D void set_face_normal(HitRec& rec, V3 ray_dire, V3 outward_normal) {
    rec.front_face = dot(ray_dire, outward_normal) < 0.0;
    rec.normal = rec.front_face ? outward_normal : -outward_normal;
}

// get_sphere_uv src/hittable.rs:64-69
RT_D void get_sphere_uv(V3 p, double& u, double& v) {
    const double phi = atan2(p.z, p.x);
    const double theta = asin(p.y);
    u = 1.0 - (phi + D_PI) / (2.0 * D_PI);
    v = (theta + D_PI / 2.0) / D_PI;
}

// Ray of the candidate's own space: the world ray pushed through `depth` wrappers, outermost first.
template <class SC, class PX>
RT_D void ray_to_depth(const SC& S, PX inner, uint32_t depth, V3& o, V3& d) {
    for (uint32_t k = 0; k < depth; ++k) xf_apply(S.xforms + inner->anc[k], o, d);
}

RT_D void fill_rect_record(HitRec& rec, V3 o, V3 d, double t, uint32_t axis, double a0, double a1, double b0, double b1) {
    // src/aarect.rs:35-46 (XY), 86-97 (XZ), 137-148 (YZ)
    double oa, da, ob, db;
    V3 outward_normal;
    if (axis == 0u) { oa = o.x; da = d.x; ob = o.y; db = d.y; outward_normal = v3(0.0, 0.0, 1.0); }
    else if (axis == 1u) { oa = o.x; da = d.x; ob = o.z; db = d.z; outward_normal = v3(0.0, 1.0, 0.0); }
    else { oa = o.y; da = d.y; ob = o.z; db = d.z; outward_normal = v3(1.0, 0.0, 0.0); }
    const double a = oa + t * da;
    const double b = ob + t * db;
    rec.u = (a - a0) / (a1 - a0);
    rec.v = (b - b0) / (b1 - b0);
    set_face_normal(rec, d, outward_normal);
    rec.p = at(o, d, t);
}

// ow, dw, tm: the ray handed to world.hit. `best` must be a hit.
// want_uv = false skips the spherical u, v (atan2, asin), which only an image texture reads (src/texture.rs:93-98).
template <class SC>
RT_D void build_hit_record(const SC& S, V3 ow, V3 dw, double tm, const Cand& best, HitRec& rec, bool want_uv = true) {
    const uint32_t kind = best.ref & 15u;
    const uint32_t idx = best.ref >> 4;
    const uint32_t xf1 = best.aux >> 8;
    const auto inner = S.xforms + (xf1 ? xf1 - 1u : 0u);  // only dereferenced when xf1 != 0
    const uint32_t depth = xf1 ? inner->depth : 0u;
    V3 o = ow, d = dw;
    if (xf1) ray_to_depth(S, inner, depth, o, d);
    const double t = best.t;
    rec.t = t;
    rec.u = 0.0;
    rec.v = 0.0;
    switch (kind) {
        case RT_N_SPHERE: {  // src/hittable.rs:84-93
            const auto s = S.spheres + idx;
            const V3 c = v3p(s->c);
            rec.p = at(o, d, t);
            const V3 outward_normal = (rec.p - c) / s->r;
            set_face_normal(rec, d, outward_normal);
            if (want_uv) get_sphere_uv(outward_normal, rec.u, rec.v);
            rec.mat = s->mat;
            break;
        }
        case RT_N_MSPHERE: {  // src/hittable.rs:227-231 (u, v not written there; 0 here)
            const auto s = S.msphere + idx;
            rec.p = at(o, d, t);
            const V3 outward_normal = (rec.p - msphere_center(s, tm)) / s->r;
            set_face_normal(rec, d, outward_normal);
            rec.mat = s->mat;
            break;
        }
        case RT_N_RECT: {
            const auto r = S.rects + idx;
            fill_rect_record(rec, o, d, t, r->axis, r->a0, r->a1, r->b0, r->b1);
            rec.mat = r->mat;
            break;
        }
        case RT_N_BOX6: {  // the side that won, src/box6.rs:22-71
            const auto b = S.boxes + idx;
            const V3 p0 = v3p(b->p0), p1 = v3p(b->p1);
            const uint32_t side = best.aux & 255u;
            if (side < 2u) fill_rect_record(rec, o, d, t, 0u, p0.x, p1.x, p0.y, p1.y);
            else if (side < 4u) fill_rect_record(rec, o, d, t, 1u, p0.x, p1.x, p0.z, p1.z);
            else fill_rect_record(rec, o, d, t, 2u, p0.y, p1.y, p0.z, p1.z);
            rec.mat = b->mat;
            break;
        }
        case RT_N_TRIANGLE: {  // src/triangle.rs:30-67; u, v recomputed with the same expressions as the test
            const auto tr = S.triangles + idx;
            const V3 point0 = v3p(tr->p0), edge1 = v3p(tr->e1), edge2 = v3p(tr->e2);
            const V3 pvec = cross(d, edge2);
            const double det = dot(pvec, edge1);
            const double inv_det = 1.0 / det;
            const V3 tvec = o - point0;
            rec.u = dot(tvec, pvec) * inv_det;
            const V3 qvec = cross(tvec, edge1);
            rec.v = dot(d, qvec) * inv_det;
            const V3 n = cross(edge1, edge2);
            const double len = length(n);
            const V3 outward_normal = v3(n.x / len, n.y / len, n.z / len);  // Vec3::unit src/vec3.rs:283-294
            set_face_normal(rec, d, outward_normal);
            rec.p = at(o, d, t);
            rec.mat = tr->mat;
            break;
        }
        default: {  // RT_N_MEDIUM, src/constant_medium.rs:75-87
            const auto m = S.media + idx;
            rec.p = at(o, d, t);
            rec.normal = v3(1.0, 0.0, 0.0);
            rec.front_face = true;
            rec.mat = m->mat;
            break;
        }
    }
    // out through the wrappers, innermost first
    for (uint32_t lvl = depth; lvl > 0u; --lvl) {
        const auto x = S.xforms + inner->anc[lvl - 1u];
        // direction of the ray this wrapper handed to its child (Translate keeps it, RotateY rotates it)
        V3 ol = ow, dl = dw;
        for (uint32_t k = 0; k < lvl; ++k) xf_apply(S.xforms + inner->anc[k], ol, dl);
        if (x->kind == 0u) {  // Translate::hit src/hittable.rs:287-289
            rec.p = rec.p + v3p(x->offset);
            set_face_normal(rec, dl, rec.normal);
        } else {  // RotateY::hit src/hittable.rs:374-384
            const double c = x->cos_theta, s = x->sin_theta;
            V3 p = rec.p, normal = rec.normal;
            p.x = c * rec.p.x + s * rec.p.z;
            p.z = -s * rec.p.x + c * rec.p.z;
            normal.x = c * rec.normal.x + s * rec.normal.z;
            normal.z = -s * rec.normal.x + c * rec.normal.z;
            rec.p = p;
            set_face_normal(rec, dl, normal);
        }
    }
}

// Material of the winning candidate without building the record.
template <class SC>
RT_D uint32_t hit_material(const SC& S, const Cand& best) {
    const uint32_t idx = best.ref >> 4;
    switch (best.ref & 15u) {
        case RT_N_SPHERE: return S.spheres[idx].mat;
        case RT_N_MSPHERE: return S.msphere[idx].mat;
        case RT_N_RECT: return S.rects[idx].mat;
        case RT_N_BOX6: return S.boxes[idx].mat;
        case RT_N_TRIANGLE: return S.triangles[idx].mat;
        default: return S.media[idx].mat;
    }
}

// prim_id / sub as RtHit reports them.
template <class SC>
RT_D void hit_ids(const SC& S, const Cand& best, uint32_t& prim_id, uint32_t& sub) {
    const uint32_t kind = best.ref & 15u;
    const uint32_t idx = best.ref >> 4;
    switch (kind) {
        case RT_N_SPHERE: prim_id = S.spheres[idx].prim_id; break;
        case RT_N_MSPHERE: prim_id = S.msphere[idx].prim_id; break;
        case RT_N_RECT: prim_id = S.rects[idx].prim_id; break;
        case RT_N_BOX6: prim_id = S.boxes[idx].prim_id; break;
        case RT_N_TRIANGLE: prim_id = S.triangles[idx].prim_id; break;
        default: prim_id = S.media[idx].prim_id; break;
    }
    const uint32_t xf1 = best.aux >> 8;
    const uint32_t inst1 = xf1 ? S.xforms[xf1 - 1u].inst_id + 1u : 0u;
    sub = (best.aux & 255u) | (inst1 << 8);
}

}  // namespace rtd
