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
//
// Shape of the walk ("while-while"): every lane first runs through box tests until
// it stands on a non-box entry, then the wave handles those entries; lanes of a
// wave therefore execute the box test together instead of interleaving it with
// primitive tests. The ray a wrapper must return to on exit is not kept in
// registers: it is re-derived from the stored world ray by replaying the parent's
// wrapper chain (same operations, same order, hence the same bits).
#pragma once
#include "../../../include/rt/abi.h"
#include "../rng.h"
#include "vec.h"

namespace rtd {

constexpr double D_INF = __builtin_huge_val();

// Pointers to the scene tables. IN_LDS = true: the staged copies in LDS (address space 3); images and texels
// always stay in global memory.
template <bool IN_LDS> struct TablePtr { template <class T> using type = const T*; };
template <> struct TablePtr<true> { template <class T> using type = const RT_LDS T*; };

// Media in world space whose boundary is one sphere (the fog and the ball of smoke of final_scene, src/main.rs:642-662)
// are not walked as stream entries: the library takes them out of its copy of the stream and they are put to the ray
// before the walk (deferred_media below).
constexpr uint32_t MAX_DEFERRED_MEDIA = 2u;
struct DeferredMedia {
    uint32_t n;
    uint32_t medium[MAX_DEFERRED_MEDIA];  // index into media
    uint32_t sphere[MAX_DEFERRED_MEDIA];  // index into spheres: the boundary
};
// boundary queries made by deferred_media, as the reference would have made them (for the work counters)
struct FirstWork {
    uint32_t spheres;  // Sphere::hit calls
};

template <bool IN_LDS>
struct SceneT {
    template <class T> using P = typename TablePtr<IN_LDS>::template type<T>;
    P<RtNode> nodes;
    P<RtSphere> spheres;
    P<RtMovingSphere> msphere;
    P<RtRect> rects;
    P<RtBox6> boxes;
    P<RtTriangle> triangles;
    P<RtXform> xforms;
    P<RtMedium> media;
    P<RtMaterial> materials;
    P<RtTexture> textures;
    P<RtPerlin> perlins;
    const RtImage* images;
    const uint8_t* texels;
    uint32_t n_nodes;
    double carried_t0, carried_t1;  // ray times for which boxes marked RT_INNER_CARRIED / RT_INNER_COVER hold (PER_AXIS scenes)
    DeferredMedia deferred;         // media taken out of the entry stream, see deferred_media()
};
using DevScene = SceneT<false>;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
RT_D void load_node(const RtNode* p, uint4& a, uint4& b) {
    const u32x4* q = reinterpret_cast<const u32x4*>(p);
    u32x4 x = q[0], y = q[1];
    // Both halves are requested before either is used: one trip to L2 / HBM per entry instead of two dependent
    // ones (left alone, the compiler postpones the box half until it has looked at the kind).
    asm volatile("" : "+v"(x), "+v"(y));
    a = make_uint4(x.x, x.y, x.z, x.w);
    b = make_uint4(y.x, y.y, y.z, y.w);
}
RT_D void load_node(const RT_LDS RtNode* p, uint4& a, uint4& b) {
    const RT_LDS u32x4* q = reinterpret_cast<const RT_LDS u32x4*>(p);
    u32x4 x = q[0], y = q[1];
#ifdef RT_LDS_BOTH_HALVES
    asm volatile("" : "+v"(x), "+v"(y));
#endif
    a = make_uint4(x.x, x.y, x.z, x.w);
    b = make_uint4(y.x, y.y, y.z, y.w);
}

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
    uint32_t prim[9];   // entries visited by RtNodeKind (the RT_N_INNER slot counts box tests)
    uint32_t first_media;  // of those: media asked before the walk ...
    FirstWork first;       // ... and the queries of their boundaries
};

// Where the world-space ray of a query lives in memory.
struct RaySoA {  // path state columns
    const double *ox, *oy, *oz, *dx, *dy, *dz;
    uint32_t slot;
    RT_D void load(V3& o, V3& d) const { o = v3(ox[slot], oy[slot], oz[slot]); d = v3(dx[slot], dy[slot], dz[slot]); }
};
struct RayAoS {  // RtRay
    const RtRay* r;
    RT_D void load(V3& o, V3& d) const { o = v3p(r->o); d = v3p(r->d); }
};

// ---- wrappers: ray into object space -------------------------------------------
// Translate::hit src/hittable.rs:278-282; RotateY::hit :355-368.
template <class PX>
RT_D void xf_apply(PX x, V3& o, V3& d) {
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

// The ray as seen inside wrapper xf1 - 1 (xf1 == 0: the world ray).
template <class SC, class RS>
RT_D void ray_at(const SC& S, const RS& rs, uint32_t xf1, V3& o, V3& d) {
    rs.load(o, d);
    if (xf1) {
        const auto x = S.xforms + (xf1 - 1u);
        const uint32_t depth = x->depth;
        for (uint32_t k = 0; k < depth; ++k) xf_apply(S.xforms + x->anc[k], o, d);
    }
}

// Per-ray-segment constants of the box and sphere tests.
struct Seg {
    V3 o, d, invd;
    V3 noi;     // -(o * invd), the constant term of the box test
    double dd;  // d . d, the `a` of the sphere quadratic (src/hittable.rs:74)
    RT_D void set(V3 o_, V3 d_) {
        o = o_; d = d_;
        invd = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);  // src/aabb.rs:17
        noi = v3(-(o.x * invd.x), -(o.y * invd.y), -(o.z * invd.z));
        dd = squared_length(d);
    }
    // Translate::hit moves the origin only (src/hittable.rs:278-282): direction-derived terms stay.
    RT_D void set_origin(V3 o_) {
        o = o_;
        noi = v3(-(o.x * invd.x), -(o.y * invd.y), -(o.z * invd.z));
    }
    // RotateY::hit changes x and z of origin and direction (src/hittable.rs:358-362): the y terms stay.
    RT_D void set_xz(double ox, double oz, double dx, double dz) {
        o.x = ox; o.z = oz; d.x = dx; d.z = dz;
        invd.x = 1.0 / dx; invd.z = 1.0 / dz;
        noi.x = -(ox * invd.x); noi.z = -(oz * invd.z);
        dd = squared_length(d);
    }
};

// ---- box test -------------------------------------------------------------------
// Carried-interval slab test on the f32 box of a BVHNode. The reference's AABB::hit (src/aabb.rs:15-30)
// restarts from the caller's tmin/tmax on every axis, which only makes it looser; any conservative test gives
// the same closest hit as long as boxes bound their contents. The rule, shared bit for bit with the CPU
// oracle's "tight" mode:
//     near/far = the box planes picked by the sign of 1/d (the reference's swap, src/aabb.rs:20-22)
//     t0 = fma(near, 1/d, -(o * 1/d)),  t1 = fma(far, 1/d, -(o * 1/d))      per axis
//     lo = max(t0x, t0y, t0z, tmin),  hi = min(t1x, t1y, t1z, tmax),  NaN terms ignored (fmax/fmin)
//     pass unless hi <= lo; a NaN tmin or tmax passes (as src/aabb.rs:23-25 would)
RT_D bool box_hit(const uint4 na, const uint4 nb, const Seg& g, double tmin, double tmax) {
    const float lox = __uint_as_float(na.x), loy = __uint_as_float(na.y), loz = __uint_as_float(na.z);
    const float hix = __uint_as_float(na.w), hiy = __uint_as_float(nb.x), hiz = __uint_as_float(nb.y);
    const bool sx = g.invd.x < 0.0, sy = g.invd.y < 0.0, sz = g.invd.z < 0.0;
    const double t0x = fma((double)(sx ? hix : lox), g.invd.x, g.noi.x);
    const double t1x = fma((double)(sx ? lox : hix), g.invd.x, g.noi.x);
    const double t0y = fma((double)(sy ? hiy : loy), g.invd.y, g.noi.y);
    const double t1y = fma((double)(sy ? loy : hiy), g.invd.y, g.noi.y);
    const double t0z = fma((double)(sz ? hiz : loz), g.invd.z, g.noi.z);
    const double t1z = fma((double)(sz ? loz : hiz), g.invd.z, g.noi.z);
    // The last max / min take the interval bounds, values the compiler cannot prove free of signalling NaNs; written
    // as the bare instructions they cost one operation each instead of two (a quieting max(x, x) first). No value
    // here is ever a signalling NaN, for which alone the two forms differ.
    double lo = fmax(fmax(t0x, t0y), t0z), hi = fmin(fmin(t1x, t1y), t1z);
    asm("v_max_f64 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(tmin));
    asm("v_min_f64 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(tmax));
    return !(hi <= lo) || (tmin != tmin) || (tmax != tmax);
}

// The reference's own rule (src/aabb.rs:15-30): tmin / tmax are not carried from axis to axis, the box passes when
// on every axis by itself the slab interval clipped to [tmin, tmax] is not empty. Looser than the carried test; the
// two find the same closest hit when boxes bound their contents, and differ when they do not (a MovingSphere in a
// tree built over a shorter time interval than its motion, src/main.rs:315). Used for scenes flagged
// RT_DESC_BOXES_MAY_NOT_BOUND; the planes are the f32 outward-rounded ones of the stream.
RT_D bool box_hit_per_axis(const uint4 na, const uint4 nb, const Seg& g, double tmin, double tmax) {
    const float lox = __uint_as_float(na.x), loy = __uint_as_float(na.y), loz = __uint_as_float(na.z);
    const float hix = __uint_as_float(na.w), hiy = __uint_as_float(nb.x), hiz = __uint_as_float(nb.y);
    const bool sx = g.invd.x < 0.0, sy = g.invd.y < 0.0, sz = g.invd.z < 0.0;
    // (plane - origin) * (1 / d), the reference's own form (:18-19): unlike the fused form of box_hit it stays +-inf,
    // not NaN, for a ray that runs parallel to the slab outside it, and such a ray must miss a box that is relied on
    // to clip
    const double t0x = ((double)(sx ? hix : lox) - g.o.x) * g.invd.x;
    const double t1x = ((double)(sx ? lox : hix) - g.o.x) * g.invd.x;
    const double t0y = ((double)(sy ? hiy : loy) - g.o.y) * g.invd.y;
    const double t1y = ((double)(sy ? loy : hiy) - g.o.y) * g.invd.y;
    const double t0z = ((double)(sz ? hiz : loz) - g.o.z) * g.invd.z;
    const double t1z = ((double)(sz ? loz : hiz) - g.o.z) * g.invd.z;
    // t0 = t0 > tmin ? t0 : tmin; t1 = t1 < tmax ? t1 : tmax; fail when t1 <= t0 (:23-25); a NaN t0 / t1 gives way
    // to tmin / tmax, as the comparisons of the reference do
    const bool fx = fmin(t1x, tmax) <= fmax(t0x, tmin);
    const bool fy = fmin(t1y, tmax) <= fmax(t0y, tmin);
    const bool fz = fmin(t1z, tmax) <= fmax(t0z, tmin);
    return !(fx || fy || fz) || (tmin != tmin) || (tmax != tmax);
}

// ---- primitives -----------------------------------------------------------------
// Every primitive test below evaluates the reference's expressions in the reference's order.

// Sphere::hit src/hittable.rs:72-113 (strict t_min < t < t_max). `a` = d . d > 0.
RT_D bool sphere_roots(V3 oc, V3 d, double a, double r, double tmin, double tmax, double& t_out) {
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

// MovingSphere::center src/hittable.rs:208-211. x / 1.0 == x exactly, so the common shutter (0, 1) needs no division.
template <class PM>
RT_D V3 msphere_center(PM s, double time) {
    const V3 c0 = v3p(s->c0), c1 = v3p(s->c1);
    const double den = s->t1 - s->t0;
    const double num = time - s->t0;
    const double f = den == 1.0 ? num : num / den;
    return c0 + (c1 - c0) * f;
}

// XYRect/XZRect/YZRect::hit src/aarect.rs:30-39, 81-90, 132-141: plane, then inclusive bounds.
// (ok, dk) = ray origin/direction along the plane normal, (oa, da), (ob, db) in the plane.
RT_D bool rect_t(double k, double a0, double a1, double b0, double b1, double ok, double dk, double oa, double da,
                 double ob, double db, double tmin, double tmax, double& t_out) {
    const double t = (k - ok) / dk;
    if (t < tmin || t > tmax) return false;
    const double a = oa + t * da;
    const double b = ob + t * db;
    if (a < a0 || a > a1 || b < b0 || b > b1) return false;
    t_out = t;
    return true;
}

template <class PR>
RT_D bool rect_hit(PR r, const Seg& g, double tmin, double tmax, double& t_out) {
    const uint32_t axis = r->axis;
    const V3 o = g.o, d = g.d;
    double ok, dk, oa, da, ob, db;
    if (axis == 0u) { ok = o.z; dk = d.z; oa = o.x; da = d.x; ob = o.y; db = d.y; }
    else if (axis == 1u) { ok = o.y; dk = d.y; oa = o.x; da = d.x; ob = o.z; db = d.z; }
    else { ok = o.x; dk = d.x; oa = o.y; da = d.y; ob = o.z; db = d.z; }
    return rect_t(r->k, r->a0, r->a1, r->b0, r->b1, ok, dk, oa, da, ob, db, tmin, tmax, t_out);
}

// Box6::hit -> HittableList::hit over the six sides in src/box6.rs:22-71 order, each against the
// running closest t (src/hittable.rs:151-164). Returns the side index of the last accepted side.
template <class PB>
RT_D bool box6_hit(PB b, V3 o, V3 d, double tmin, double& tmax, uint32_t& side_out) {
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

// The same answer as box6_hit for one division instead of six, for the rays whose case is clear-cut; every other ray
// takes box6_hit. The list of six sides returns the accepted side with the smallest t (an equal t goes to the later
// side), and a side is accepted when its t = (k - o) / d lies in [tmin, tmax] and the point at t lies within the side's
// bounds. With the plane distances estimated as (k - o) * (1 / d) (within two ulp of the quotient) and a margin m far
// above every rounding error involved:
//   entry: the ray reaches the last of its three near planes at te > tmin + m, more than m after the other two and
//          more than m before the first far plane. At that plane the ray is strictly inside the other two slabs (its
//          side is in bounds), at the other two near planes it has not yet entered this slab (those sides are out of
//          bounds), and the far sides lie beyond. The answer is that side, if its exact t is not above tmax.
//   exit:  all three near planes lie more than m before tmin (those sides fail the range check) and the first far
//          plane is reached at tx > tmin + m, more than m before the other two: that far side is in bounds and the
//          other two lie beyond. The answer is that side, if its exact t is not above tmax.
//   behind: the first far plane lies more than 2 m before tmin (a ray leaving the box it starts on): a side whose t
//          is not below tmin is reached more than m after the ray has left that slab, out of bounds. No side is hit.
// The margin is relative to the coordinates involved, not to t: a point o + t d is compared with bounds whose
// rounding error scales with |o| and the plane positions.
template <class PB>
RT_D bool box6_hit_seg(PB b, const Seg& g, double tmin, double& tmax, uint32_t& side_out) {
    const V3 p0 = v3p(b->p0), p1 = v3p(b->p1);
    const double ex0 = p0.x - g.o.x, ex1 = p1.x - g.o.x, ey0 = p0.y - g.o.y, ey1 = p1.y - g.o.y, ez0 = p0.z - g.o.z, ez1 = p1.z - g.o.z;
    const double x0 = ex0 * g.invd.x, x1 = ex1 * g.invd.x, y0 = ey0 * g.invd.y, y1 = ey1 * g.invd.y, z0 = ez0 * g.invd.z, z1 = ez1 * g.invd.z;
    const double nx = fmin(x0, x1), fx = fmax(x0, x1), ny = fmin(y0, y1), fy = fmax(y0, y1), nz = fmin(z0, z1), fz = fmax(z0, z1);
    const double te = fmax(fmax(nx, ny), nz), tx = fmin(fmin(fx, fy), fz);
    const double te2 = fmax(fmin(nx, ny), fmin(fmax(nx, ny), nz));  // second largest near distance
    const double tx2 = fmin(fmax(fx, fy), fmax(fmin(fx, fy), fz));  // second smallest far distance
    const double scale = (fabs(g.o.x) + fabs(p0.x) + fabs(p1.x)) * fabs(g.invd.x) + (fabs(g.o.y) + fabs(p0.y) + fabs(p1.y)) * fabs(g.invd.y) +
                         (fabs(g.o.z) + fabs(p0.z) + fabs(p1.z)) * fabs(g.invd.z);
    const double m = scale * 0x1p-40;
    // finite everywhere (no direction component is zero, nothing is NaN) and a box with volume
    const bool regular = scale < 1e300 && p0.x < p1.x && p0.y < p1.y && p0.z < p1.z;
    const bool entry = regular && te > tmin + m && te - te2 > m && tx - te > m;
    const bool leave = regular && te < tmin - m && tx > tmin + m && tx2 - tx > m;
    if (regular && tx < tmin - (m + m)) return false;
    if (!(entry || leave)) return box6_hit(b, g.o, g.d, tmin, tmax, side_out);
    const double ref = entry ? te : tx;  // exactly one of the six estimates equals it
    double num = ez1, den = g.d.z;
    uint32_t side = 0u;
    if (z0 == ref) { num = ez0; side = 1u; }
    if (y1 == ref) { num = ey1; den = g.d.y; side = 2u; }
    if (y0 == ref) { num = ey0; den = g.d.y; side = 3u; }
    if (x1 == ref) { num = ex1; den = g.d.x; side = 4u; }
    if (x0 == ref) { num = ex0; den = g.d.x; side = 5u; }
    const double t = num / den;  // the reference's (k - o) / d of that side
    if (t < tmin || t > tmax) return false;
    tmax = t;
    side_out = side;
    return true;
}

// Triangle::hit src/triangle.rs:26-59 (Moeller-Trumbore, two-sided, t range inclusive).
template <class PT>
RT_D bool triangle_hit(PT tr, V3 o, V3 d, double tmin, double tmax, double& t_out) {
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

// The two boundary queries of ConstantMedium::hit (src/constant_medium.rs:41-47) when the boundary is a
// single Sphere: boundary.hit(r, -inf, inf) and boundary.hit(r, t1 + 0.0001, inf) evaluate the same quadratic,
// so the roots are computed once and run through Sphere::hit's two range checks twice. `queries` = how many of
// the two Sphere::hit calls the reference would have made (the second only after the first hit).
template <class PS>
RT_D bool sphere_boundary(PS s, V3 o, V3 d, double dd, double& t1, double& t2, uint32_t& queries) {
    queries = 1u;
    const V3 oc = o - v3p(s->c);
    const double a = dd;
    const double half_b = dot(oc, d);
    const double c = squared_length(oc) - s->r * s->r;
    const double discriminant = half_b * half_b - a * c;
    if (!(discriminant > 0.0)) return false;
    const double root = sqrt(discriminant);
    const double ra = (-half_b - root) / a;
    const double rb = (-half_b + root) / a;
    if (ra < D_INF && ra > -D_INF) t1 = ra;
    else if (rb < D_INF && rb > -D_INF) t1 = rb;
    else return false;
    queries = 2u;
    const double lo = t1 + 0.0001;
    if (ra < D_INF && ra > lo) t2 = ra;
    else if (rb < D_INF && rb > lo) t2 = rb;
    else return false;
    return true;
}

// ConstantMedium::hit after its two boundary queries, src/constant_medium.rs:52-88: [t1, t2] is where the ray is inside
// the boundary, [tmin, best.t] the range of the query; dd = d . d; wrapper = innermost wrapper around the medium, table
// index + 1, 0 = none. The one random number is keyed by the medium, so the outcome does not depend on what was searched before.
template <class SC>
RT_D void medium_resolve(const SC& S, const RayKey& key, uint32_t meta, double t1, double t2, double tmin, double dd, uint32_t wrapper,
                         Cand& best) {
    const auto m = S.media + (meta >> 4);
    if (t1 < tmin) t1 = tmin;
    if (t2 > best.t) t2 = best.t;
    if (t1 > t2) return;
    if (t1 < 0.0) t1 = 0.0;
    const double ray_length = sqrt(dd);  // r.dire.length()
    const double distance_inside_boundary = (t2 - t1) * ray_length;
    rt::Rng rng;
    rng.set_path(key.seed, key.pixel, key.sample, key.bounce, rt::RNG_MEDIUM0 + m->rng_id);
    const double hit_distance = m->neg_inv_density * log10(rng.uniform(0.0, 1.0));  // :69
    if (hit_distance > distance_inside_boundary) return;
    best.t = t1 + hit_distance / ray_length;
    best.ref = meta;
    best.aux = wrapper << 8;
}

// The media that were taken out of the entry stream, put to the world ray (o, d): ConstantMedium::hit of each, in stream
// order, with the closest hit so far as t_max. The kernels do this BEFORE the walk (first_candidate in kernels.h), as if
// these media stood first in the world's list. In the reference's order such a medium is asked somewhere in the middle,
// with the closest hit so far as t_max, and what comes after it with the medium's hit as t_max; the closest of all
// candidates wins either way, because the medium's candidate t does not depend on t_max (only whether it is accepted
// does) and its random number is keyed by the medium. The orders can differ only when the medium's t and another
// object's t agree to the last bits (the medium's acceptance compares distances, the other objects compare t).
template <class SC>
RT_D void deferred_media(const SC& S, const RayKey& key, V3 o, V3 d, double tmin, Cand& best, FirstWork& fw) {
    fw.spheres = 0u;
    const double dd = squared_length(d);
#pragma unroll  // constant indices: the lists live in registers, not in scratch memory
    for (uint32_t k = 0u; k < MAX_DEFERRED_MEDIA; ++k) {
        if (k >= S.deferred.n) break;
        double t1 = 0.0, t2 = 0.0;
        uint32_t queries;
        const bool inside = sphere_boundary(S.spheres + S.deferred.sphere[k], o, d, dd, t1, t2, queries);
        fw.spheres += queries;
        if (inside) medium_resolve(S, key, RT_N_MEDIUM | (S.deferred.medium[k] << 4), t1, t2, tmin, dd, 0u, best);
    }
}

#ifdef RT_DIAG_WAVE
// Diagnostic build (tools/diag_util.py): the work counters count wave-level executions of each entry kind (one
// lane per wave counts) instead of lanes, which gives the lane utilisation of every phase of the loop.
RT_D bool diag_first_lane() { return (int)__lane_id() == __ffsll((unsigned long long)__ballot(1)) - 1; }
#endif

// ---- the walk ---------------------------------------------------------------------
// State of one closest-hit query, advanced in two kinds of steps so that a wave can keep its lanes together:
// box_run() consumes box entries until the lane stands on a non-box entry (or the query is finished), leaf()
// handles that one entry. One walker serves the world query and, re-entered in place, the two boundary queries
// of a ConstantMedium (src/constant_medium.rs:41-47): on a medium entry the running result is parked, the walk
// covers the boundary's entries [i+1, link) with the boundary query's own t range (mode 1, then mode 2), and at
// the end of that range the medium is resolved against the parked result and the world walk resumes at `link`.
// Wrappers inside a boundary are entered and left like anywhere else, so at the end of the range the ray is back
// in the medium's own space. The flattener rejects media inside a boundary: one level of parking is enough.
// BOXMEDIA: the scene has a box of six sides alone inside wrappers, on its own or as the boundary of a ConstantMedium;
// only then is the code compiled in that handles such a box in one step, in its own space, without entering and
// leaving the wrappers (it costs registers everywhere else).
// GENERAL: the scene has a ConstantMedium whose boundary is anything else than one sphere or one (wrapped) box, so the
// re-entrant boundary walk (modes 1 and 2 and the parked world result) is needed. Without it the walker fits 128
// registers and the kernel runs four waves per SIMD instead of three.
// PER_AXIS: box entries not marked RT_INNER_CARRIED are tested with the reference's per-axis rule (box_hit_per_axis)
// instead of the carried interval; for scenes whose boxes may not bound their contents.
template <bool COUNT, bool BOXMEDIA = false, bool GENERAL = true, bool PER_AXIS = false>
struct Walker {
    Seg g;
    double tm, tmin0, tmin;
    Cand best;
    uint32_t i, end, cur_xf;  // cur_xf: innermost wrapper around the current entry, table index + 1, 0 = none
    uint32_t mode;            // 0 world, 1 first boundary query, 2 second boundary query
    double out_t, t_enter;    // parked world result, entry distance of the boundary
    uint32_t out_ref, out_aux, med_meta, med_begin;
    uint4 na, nb;             // the entry box_run() stopped on
    bool done;

    template <class SC>
    RT_D void init(const SC& S, V3 o, V3 d, double tm_, double tmin_, double tmax_) {
        g.set(o, d);
        tm = tm_; tmin0 = tmin_; tmin = tmin_;
        best.t = tmax_; best.ref = RT_MISS; best.aux = 0u;
        i = 0u; end = S.n_nodes; cur_xf = 0u; mode = 0u;
        out_t = 0.0; t_enter = 0.0; out_ref = 0u; out_aux = 0u; med_meta = 0u; med_begin = 0u;
        done = false;
    }

    // Entries (i, link) of the medium entry the lane stands on are: k wrapper enters, one box of six sides, k exits.
    template <class SC>
    RT_D bool wrapped_box_boundary(const SC& S, uint32_t first) const {
        uint32_t j = i + 1u, m = first;
        while ((m & 15u) == RT_N_XF_ENTER && j + 1u < nb.w) m = S.nodes[++j].meta;
        const uint32_t wrappers = j - (i + 1u);
        return (m & 15u) == RT_N_BOX6 && nb.w == i + 2u + 2u * wrappers;
    }

    // ConstantMedium::hit after its two boundary queries, src/constant_medium.rs:52-88
    template <class SC>
    RT_D void resolve_medium(const SC& S, const RayKey& key, uint32_t meta, double t1, double t2) {
        medium_resolve(S, key, meta, t1, t2, tmin, g.dd, cur_xf, best);
    }

    // Returns true when the lane stands on a non-box entry. With VOTE_DEN != 0 a lane also leaves,
    // returning false and still on a box entry, once fewer than 1 / VOTE_DEN of the lanes that entered are still
    // testing boxes: in deep trees the last few lanes of a box phase otherwise run it nearly alone (measured: final
    // scene and Weekend +8 ... 12 %), in scenes of a handful of objects the extra rounds cost more than they save
    // (Cornell smoke -16 %), so the host picks the kernel variant by the size of the entry stream.
    template <uint32_t VOTE_DEN = 0u, class SC>
    RT_D bool box_run(const SC& S, const RayKey& key, Work* work) {
        const uint32_t n_in = VOTE_DEN ? (uint32_t)__popcll(__ballot(1)) : 0u;
        for (;;) {
            // the hot loop: nothing but the entry index changes in it
            {
                uint32_t j = i;
                const uint32_t stop = end;
                const double lo = tmin, hi = best.t;
                bool leaf_found = false;
                while (j < stop) {
                    load_node(S.nodes + j, na, nb);
#ifdef RT_DIAG_WAVE
                    if (COUNT && (nb.z & 15u) == RT_N_INNER && diag_first_lane()) { work->entries++; work->prim[RT_N_INNER]++; }
#else
                    if (COUNT) { work->entries++; work->prim[nb.z & 15u]++; }
#endif
                    if ((nb.z & 15u) != RT_N_INNER) { leaf_found = true; break; }
                    bool pass;
                    if constexpr (!PER_AXIS) {
                        pass = box_hit(na, nb, g, lo, hi);
                    } else {
                        // PER_AXIS scenes: boxes marked RT_INNER_CARRIED / RT_INNER_COVER hold their content while the moving
                        // spheres are within their motion intervals and are tested with the carried interval for rays of
                        // such times; for other rays a covering box is no obstacle and a node gets the reference's test.
                        // (Whether the ray's time lies within the motion intervals is looked up from tm: a flag kept per
                        // lane costs a register the kernel does not have.)
                        uint32_t mark = nb.z >> 4;
                        const bool in_motion = tm >= S.carried_t0 && tm <= S.carried_t1;
                        pass = true;
                        if (mark != 0u && in_motion) pass = box_hit(na, nb, g, lo, hi);
                        // A covering box that lets the ray pass is followed by the node it covers, an unmarked box entry,
                        // which is tested in the same round: lanes on covering boxes and lanes on nodes would otherwise
                        // drift apart and every round would run both tests for half of the lanes each.
                        if (mark == RT_INNER_COVER && pass) {
                            ++j;
                            load_node(S.nodes + j, na, nb);
#ifdef RT_DIAG_WAVE
                            if (COUNT && diag_first_lane()) { work->entries++; work->prim[RT_N_INNER]++; }
#else
                            if (COUNT) { work->entries++; work->prim[RT_N_INNER]++; }
#endif
                            mark = 0u;
                        }
                        if (mark == 0u || (mark == RT_INNER_CARRIED && !in_motion)) pass = box_hit_per_axis(na, nb, g, lo, hi);
                    }
                    j = pass ? j + 1u : nb.w;
                    if (VOTE_DEN != 0u && (uint32_t)__popcll(__ballot(1)) * VOTE_DEN < n_in) { i = j; return false; }
                }
                i = j;
                if (leaf_found) return true;
            }
            // i >= end: the end of the world, or of a boundary query (`end` is then the medium's link)
            if (!GENERAL || mode == 0u) { done = true; return false; }
            if (mode == 1u && best.ref != RT_MISS) {
                t_enter = best.t;
                tmin = best.t + 0.0001;  // src/constant_medium.rs:45
                best.t = D_INF;
                best.ref = RT_MISS;
                mode = 2u;
                i = med_begin;
                continue;
            }
            const bool inside = best.ref != RT_MISS;  // mode 2 and the second query hit
            const double t2 = best.t;
            best.t = out_t; best.ref = out_ref; best.aux = out_aux;
            tmin = tmin0;
            i = end;
            end = S.n_nodes;
            mode = 0u;
            if (inside) resolve_medium(S, key, med_meta, t_enter, t2);
        }
    }

    template <class SC, class RS>
    RT_D void leaf(const SC& S, const RS& rs, const RayKey& key, Work* work) {
        const uint32_t meta = nb.z;
        const uint32_t kind = meta & 15u;
        const uint32_t idx = meta >> 4;
#ifdef RT_DIAG_WAVE
#define RT_DIAG_LEAF(k) if (COUNT && diag_first_lane()) { work->entries++; work->prim[k]++; }
#else
#define RT_DIAG_LEAF(k)
#endif
        switch (kind) {
            case RT_N_SPHERE:
            case RT_N_MSPHERE: {
                RT_DIAG_LEAF(RT_N_SPHERE)  // one code path: a moving sphere is a sphere about center(r.tm) (src/hittable.rs:216)
                V3 c;
                double r;
                if (kind == RT_N_SPHERE) { const auto s = S.spheres + idx; c = v3p(s->c); r = s->r; }
                else { const auto s = S.msphere + idx; c = msphere_center(s, tm); r = s->r; }
                double t;
                if (sphere_roots(g.o - c, g.d, g.dd, r, tmin, best.t, t)) { best.t = t; best.ref = meta; best.aux = cur_xf << 8; }
                break;
            }
            case RT_N_RECT: {
                RT_DIAG_LEAF(RT_N_RECT)
                double t;
                if (rect_hit(S.rects + idx, g, tmin, best.t, t)) { best.t = t; best.ref = meta; best.aux = cur_xf << 8; }
                break;
            }
            case RT_N_BOX6: {
                RT_DIAG_LEAF(RT_N_BOX6)
                uint32_t side = 0u;
                double t = best.t;
                if (box6_hit_seg(S.boxes + idx, g, tmin, t, side)) { best.t = t; best.ref = meta; best.aux = side | (cur_xf << 8); }
                break;
            }
            case RT_N_TRIANGLE: {
                RT_DIAG_LEAF(RT_N_TRIANGLE)
                double t;
                if (triangle_hit(S.triangles + idx, g.o, g.d, tmin, best.t, t)) { best.t = t; best.ref = meta; best.aux = cur_xf << 8; }
                break;
            }
            case RT_N_XF_ENTER: {
                RT_DIAG_LEAF(RT_N_XF_ENTER)
                if constexpr (BOXMEDIA) {
                    // wrappers around nothing but one box of six sides (src/main.rs:497-515): the ray is taken into
                    // the box's space for this one test and the walker's own ray is left alone
                    uint32_t j = i, m = meta, wrappers = 0u;
                    while ((m & 15u) == RT_N_XF_ENTER && j + 1u < end) { m = S.nodes[++j].meta; ++wrappers; }
                    bool alone = (m & 15u) == RT_N_BOX6 && j + wrappers < end;
                    for (uint32_t k = 1u; alone && k <= wrappers; ++k) alone = (S.nodes[j + k].meta & 15u) == RT_N_XF_EXIT;
                    if (alone) {
                        V3 o = g.o, d = g.d;
                        uint32_t inner = idx;
                        for (uint32_t k = 0u; k < wrappers; ++k) {
                            inner = S.nodes[i + k].meta >> 4;
                            xf_apply(S.xforms + inner, o, d);
                        }
                        uint32_t side = 0u;
                        double t = best.t;
                        if (box6_hit(S.boxes + (m >> 4), o, d, tmin, t, side)) { best.t = t; best.ref = m; best.aux = side | ((inner + 1u) << 8); }
#ifndef RT_DIAG_WAVE
                        if (COUNT) {
                            work->entries += 2u * wrappers;
                            work->prim[RT_N_XF_ENTER] += wrappers - 1u;
                            work->prim[RT_N_BOX6] += 1u;
                            work->prim[RT_N_XF_EXIT] += wrappers;
                        }
#endif
                        i = j + wrappers;  // the last exit; the common ++i below steps past it
                        break;
                    }
                }
                // a run of wrappers (Translate around RotateY, src/main.rs:690-696) is entered in one step
                uint32_t xi = idx;
                for (;;) {
                    const auto x = S.xforms + xi;
                    V3 o = g.o, d = g.d;
                    xf_apply(x, o, d);
                    if (x->kind == 0u) g.set_origin(o);
                    else g.set_xz(o.x, o.z, d.x, d.z);
                    cur_xf = xi + 1u;
                    const uint32_t next = S.nodes[i + 1u].meta;
                    if ((next & 15u) != RT_N_XF_ENTER) break;
                    if (COUNT) { work->entries++; work->prim[RT_N_XF_ENTER]++; }
                    xi = next >> 4;
                    ++i;
                }
                break;
            }
            case RT_N_XF_EXIT: {
                RT_DIAG_LEAF(RT_N_XF_EXIT)
                // back in the space of the first enclosing wrapper that is not left here as well (or the world):
                // its ray is the world ray pushed through that wrapper's chain
                uint32_t xi = idx;
                bool rotated = false;
                for (;;) {
                    const auto x = S.xforms + xi;
                    rotated = rotated || x->kind != 0u;
                    const uint32_t depth = x->depth;
                    cur_xf = depth > 1u ? x->anc[depth - 2u] + 1u : 0u;
                    if (i + 1u >= end) break;
                    const uint32_t next = S.nodes[i + 1u].meta;
                    if ((next & 15u) != RT_N_XF_EXIT) break;
                    if (COUNT) { work->entries++; work->prim[RT_N_XF_EXIT]++; }
                    xi = next >> 4;
                    ++i;
                }
                V3 o, d;
                ray_at(S, rs, cur_xf, o, d);
                if (!rotated) g.set_origin(o);        // only Translates were left: the direction was never changed
                else g.set_xz(o.x, o.z, d.x, d.z);    // a RotateY was left: y components were never changed ...
                if (rotated) g.set_origin(o);         // ... but a Translate left with it may have moved the origin's y
                break;
            }
            default: {  // RT_N_MEDIUM, ConstantMedium::hit src/constant_medium.rs:34-89 (mode is 0 here)
                RT_DIAG_LEAF(RT_N_MEDIUM)
                const uint32_t first = S.nodes[i + 1u].meta;
                if (nb.w == i + 2u && (first & 15u) == RT_N_SPHERE) {
                    // boundary is one sphere: both queries from one quadratic
                    double t1 = 0.0, t2 = 0.0;
                    uint32_t queries;
                    const bool inside = sphere_boundary(S.spheres + (first >> 4), g.o, g.d, g.dd, t1, t2, queries);
#ifndef RT_DIAG_WAVE
                    if (COUNT) { work->entries += queries; work->prim[RT_N_SPHERE] += queries; }
#endif
                    if (inside) resolve_medium(S, key, meta, t1, t2);
                    i = nb.w;
                } else if (BOXMEDIA && wrapped_box_boundary(S, first)) {
                    // boundary is one box of six sides inside zero or more wrappers (src/main.rs:558-582): the ray is
                    // taken into the box's space once and both queries (src/constant_medium.rs:41-47) are made there
                    V3 o = g.o, d = g.d;
                    uint32_t j = i + 1u;
                    for (uint32_t m = S.nodes[j].meta; (m & 15u) == RT_N_XF_ENTER; m = S.nodes[++j].meta) xf_apply(S.xforms + (m >> 4), o, d);
                    const auto b = S.boxes + (S.nodes[j].meta >> 4);
                    uint32_t side = 0u, queries = 1u;
                    double t1 = D_INF, t2 = D_INF;
                    bool inside = box6_hit(b, o, d, -D_INF, t1, side);
                    if (inside) {
                        queries = 2u;
                        inside = box6_hit(b, o, d, t1 + 0.0001, t2, side);
                    }
#ifndef RT_DIAG_WAVE
                    if (COUNT) {
                        const uint32_t wrappers = j - (i + 1u);
                        work->entries += queries * (2u * wrappers + 1u);
                        work->prim[RT_N_BOX6] += queries;
                        work->prim[RT_N_XF_ENTER] += queries * wrappers;
                        work->prim[RT_N_XF_EXIT] += queries * wrappers;
                    }
#endif
                    if (inside) resolve_medium(S, key, meta, t1, t2);
                    i = nb.w;
                } else if (!GENERAL) {
                    i = nb.w;  // not reached: the host picks this variant only for scenes without such media
                } else {
                    out_t = best.t; out_ref = best.ref; out_aux = best.aux;
                    med_meta = meta;
                    med_begin = i + 1u;
                    best.t = D_INF;  // boundary.hit(r, -inf, inf), src/constant_medium.rs:41
                    best.ref = RT_MISS;
                    tmin = -D_INF;
                    mode = 1u;
                    end = nb.w;
                    i = i + 1u;
                }
                return;
            }
        }
        ++i;
    }
};

}  // namespace rtd
