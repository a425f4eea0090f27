//! `src/flatten.rs` for Mighty-A/RayTracer: writes the `Arc<dyn Hittable>` tree into the flat tables of
//! `librt_hip.so` (include/rt/abi.h) and renders through it. Drop-in replacement for the thread closure of
//! src/main.rs:193-230. NEVER COMPILED in this repository (the build image has no Rust toolchain); it restates,
//! type by type, what raytracer_amd/csrc/host/scene.hpp does and is tested there.
//!
//! Changes the reference needs besides adding this file (fields are private and `dyn Hittable` cannot be
//! downcast, so each trait gains one method and the fields read below become `pub(crate)`):
//!   src/main.rs      `mod flatten;` and the call in `render_with_hip` below instead of :193-230
//!   src/hittable.rs  trait Hittable (:43-46): `fn flatten(&self, f: &mut crate::flatten::Flat);`
//!                    Translate { ptr, offset } (:263-264) -> pub(crate)
//!   src/material.rs  trait Material (:8-20): `fn describe(&self, f: &mut crate::flatten::Flat) -> u32;`
//!                    Dielectric.ref_idx (:87), DiffuseLight.emit (:151) -> pub(crate)
//!   src/texture.rs   trait Texture (:8-10): `fn describe(&self, f: &mut crate::flatten::Flat) -> u32;`
//!                    SolidColor.color_value (:13), CheckerTexture.{odd, even} (:29-30), ImageTexture.img (:78) -> pub(crate)
//!   src/perlin.rs    Perlin.{ran_vec, perm_x, perm_y, perm_z} (:7-10) -> pub(crate)
//!   src/aarect.rs    XYRect / XZRect / YZRect fields (:8-13, 59-64, 110-115) -> pub(crate)
//!   src/bvh.rs       BVHNode.{left, right, _box} (:12-14) -> pub(crate)
//!   src/constant_medium.rs  ConstantMedium.{boundary, phase_function, neg_inv_density} (:11-13) -> pub(crate)
//!   src/camera.rs    Camera fields (:7-16) -> pub(crate)
//!   Cargo.toml       `raytracer-hip-sys = { path = ".../rust/raytracer-hip-sys" }`
//! and, for results comparable with the CPU path, the counter-based generator in src/rtweekend.rs:11-20
//! (raytracer_amd/csrc/rng.h).
//!
//! Identification: the reference's objects carry no ids, so `prim_id`, `inst_id` and a medium's `rng_id` are
//! handed out in the order objects are first met by the depth-first walk (the C++ host hands them out in
//! creation order; they only name hits in `RtHit` and pick a medium's random stream).
use std::collections::HashMap;
use std::sync::Arc;

use image::GenericImageView;
use raytracer_hip_sys::*;

use crate::aarect::{XYRect, XZRect, YZRect};
use crate::box6::Box6;
use crate::bvh::BVHNode;
use crate::camera::Camera;
use crate::constant_medium::ConstantMedium;
use crate::hittable::{Hittable, HittableList, MovingSphere, RotateY, Sphere, Translate};
use crate::material::{Dielectric, DiffuseLight, Isotropic, Lambertian, Material, Metal};
use crate::perlin::Perlin;
use crate::texture::{CheckerTexture, ImageTexture, NoiseTexture, SolidColor, Texture};
use crate::triangle::Triangle;
use crate::vec3::Vec3;

const RT_N_INNER: u32 = 0;
const RT_INNER_CARRIED: u32 = 1;                                           // include/rt/abi.h
const RT_INNER_COVER: u32 = 2;
const RT_N_SPHERE: u32 = 1;
const RT_N_MSPHERE: u32 = 2;
const RT_N_RECT: u32 = 3;
const RT_N_BOX6: u32 = 4;
const RT_N_TRIANGLE: u32 = 5;
const RT_N_XF_ENTER: u32 = 6;
const RT_N_XF_EXIT: u32 = 7;
const RT_N_MEDIUM: u32 = 8;
const RT_NONE: u32 = 0xFFFF_FFFF;
const RT_ABI_VERSION: u32 = 2;

fn a3(v: &Vec3) -> [f64; 3] { [v.x, v.y, v.z] }
fn key<T: ?Sized>(p: &T) -> usize { p as *const T as *const u8 as usize }

/// Largest f32 <= x / smallest f32 >= x: boxes are stored rounded outward.
fn f32_down(x: f64) -> f32 { let f = x as f32; if (f as f64) > x { f32::from_bits(if f > 0.0 { f.to_bits() - 1 } else if f < 0.0 { f.to_bits() + 1 } else { 0x8000_0001 }) } else { f } }
fn f32_up(x: f64) -> f32 { let f = x as f32; if (f as f64) < x { f32::from_bits(if f > 0.0 { f.to_bits() + 1 } else if f < 0.0 { f.to_bits() - 1 } else { 0x0000_0001 }) } else { f } }

#[derive(Default)]
pub struct Flat {
    pub nodes: Vec<RtNode>, pub spheres: Vec<RtSphere>, pub msphere: Vec<RtMovingSphere>, pub rects: Vec<RtRect>,
    pub boxes: Vec<RtBox6>, pub triangles: Vec<RtTriangle>, pub xforms: Vec<RtXform>, pub media: Vec<RtMedium>,
    pub materials: Vec<RtMaterial>, pub textures: Vec<RtTexture>, pub perlins: Vec<RtPerlin>, pub images: Vec<RtImage>,
    pub texels: Vec<u8>,
    seen: HashMap<usize, u32>,   // object address -> table index: shared objects are stored once
    xf_stack: Vec<u32>,
    in_medium: bool,
    next_prim: u32, next_inst: u32, next_medium: u32,
    /// box of the innermost BVHNode around the object being flattened, and whether some MovingSphere reaches
    /// outside it during its motion (random_scene: tree over t in (0, 0.1), src/main.rs:315): the scene is then
    /// flagged RT_DESC_BOXES_MAY_NOT_BOUND and the kernels run the per-axis box rule of src/aabb.rs:15-30
    enclosing: Option<([f64; 3], [f64; 3])>,
    pub boxes_may_not_bound: bool,
    world_box: Option<([f64; 3], [f64; 3])>,   // box of the first BVHNode flattened, the world's root
}

impl Flat {
    pub fn push_node(&mut self, kind: u32, index: u32) -> usize {
        assert!(index < (1 << 28), "table index exceeds 28 bits");
        self.nodes.push(RtNode { lo: [0.0; 3], hi: [0.0; 3], meta: kind | (index << 4), link: 0 });
        self.nodes.len() - 1
    }
    fn prim_id(&mut self) -> u32 { self.next_prim += 1; self.next_prim - 1 }

    /// Entry for a primitive stored in `table` under `addr`, building the row on first sight.
    fn leaf<R>(&mut self, addr: usize, kind: u32, table: fn(&mut Flat) -> &mut Vec<R>, make: impl FnOnce(&mut Flat) -> R) {
        let idx = match self.seen.get(&addr) {
            Some(&i) => i,
            None => { let row = make(self); let t = table(self); t.push(row); let i = (t.len() - 1) as u32; self.seen.insert(addr, i); i }
        };
        self.push_node(kind, idx);
    }

    fn push_xform(&mut self, mut x: RtXform, child: &dyn Hittable) {
        assert!(self.xf_stack.len() < 4, "more than 4 nested Translate/RotateY wrappers");
        let me = self.xforms.len() as u32;
        x.depth = self.xf_stack.len() as u32 + 1;
        x.anc = [RT_NONE; 4];
        for (i, a) in self.xf_stack.iter().enumerate() { x.anc[i] = *a; }
        x.anc[x.depth as usize - 1] = me;
        x.inst_id = self.next_inst; self.next_inst += 1;
        self.xforms.push(x);
        self.push_node(RT_N_XF_ENTER, me);
        self.xf_stack.push(me);
        child.flatten(self);
        self.xf_stack.pop();
        self.push_node(RT_N_XF_EXIT, me);
    }

    pub fn desc(&self) -> RtSceneDesc {
        RtSceneDesc {
            abi_version: RT_ABI_VERSION, flags: if self.boxes_may_not_bound { RT_DESC_BOXES_MAY_NOT_BOUND } else { 0 },
            nodes: self.nodes.as_ptr(), n_nodes: self.nodes.len() as u64,
            spheres: self.spheres.as_ptr(), n_spheres: self.spheres.len() as u64,
            msphere: self.msphere.as_ptr(), n_msphere: self.msphere.len() as u64,
            rects: self.rects.as_ptr(), n_rects: self.rects.len() as u64,
            boxes: self.boxes.as_ptr(), n_boxes: self.boxes.len() as u64,
            triangles: self.triangles.as_ptr(), n_triangles: self.triangles.len() as u64,
            xforms: self.xforms.as_ptr(), n_xforms: self.xforms.len() as u64,
            media: self.media.as_ptr(), n_media: self.media.len() as u64,
            materials: self.materials.as_ptr(), n_materials: self.materials.len() as u64,
            textures: self.textures.as_ptr(), n_textures: self.textures.len() as u64,
            perlins: self.perlins.as_ptr(), n_perlins: self.perlins.len() as u64,
            images: self.images.as_ptr(), n_images: self.images.len() as u64,
            texels: self.texels.as_ptr(), n_texel_bytes: self.texels.len() as u64,
        }
    }
}

// ---- Hittable::flatten ------------------------------------------------------------------------------------
pub fn flatten_sphere(s: &Sphere, f: &mut Flat) {                           // src/hittable.rs:48-52
    f.leaf(key(s), RT_N_SPHERE, |f| &mut f.spheres, |f| RtSphere { c: a3(&s.center), r: s.radius, mat: s.mat_ptr.describe(f), prim_id: f.prim_id() });
}
pub fn flatten_moving_sphere(s: &MovingSphere, f: &mut Flat) {              // :187-194
    if let Some((mn, mx)) = f.enclosing {
        for c in [&s.center0, &s.center1].iter() {                           // the ends of the motion bound all of it
            let c = a3(c);
            if (0..3).any(|a| c[a] - s.radius < mn[a] || c[a] + s.radius > mx[a]) { f.boxes_may_not_bound = true; }
        }
    }
    f.leaf(key(s), RT_N_MSPHERE, |f| &mut f.msphere, |f| RtMovingSphere {
        c0: a3(&s.center0), c1: a3(&s.center1), t0: s.time0, t1: s.time1, r: s.radius, mat: s.mat_ptr.describe(f), prim_id: f.prim_id() });
}
fn rect(f: &mut Flat, addr: usize, axis: u32, a0: f64, a1: f64, b0: f64, b1: f64, k: f64, mp: &Arc<dyn Material>) {
    f.leaf(addr, RT_N_RECT, |f| &mut f.rects, |f| RtRect { a0, a1, b0, b1, k, axis, mat: mp.describe(f), prim_id: f.prim_id(), _pad: 0 });
}
pub fn flatten_xy(r: &XYRect, f: &mut Flat) { rect(f, key(r), 0, r.x0, r.x1, r.y0, r.y1, r.k, &r.mp) }   // src/aarect.rs:7-14
pub fn flatten_xz(r: &XZRect, f: &mut Flat) { rect(f, key(r), 1, r.x0, r.x1, r.z0, r.z1, r.k, &r.mp) }   // :58-65
pub fn flatten_yz(r: &YZRect, f: &mut Flat) { rect(f, key(r), 2, r.y0, r.y1, r.z0, r.z1, r.k, &r.mp) }   // :109-116
pub fn flatten_box6(b: &Box6, mat: &Arc<dyn Material>, f: &mut Flat) {      // src/box6.rs:9-13; the six sides (:22-71) are implied by the corners
    f.leaf(key(b), RT_N_BOX6, |f| &mut f.boxes, |f| RtBox6 { p0: a3(&b.box_min), p1: a3(&b.box_max), mat: mat.describe(f), prim_id: f.prim_id() });
}
pub fn flatten_triangle(t: &Triangle, f: &mut Flat) {                       // src/triangle.rs:7-12; edges as :27-28 compute them
    f.leaf(key(t), RT_N_TRIANGLE, |f| &mut f.triangles, |f| RtTriangle {
        p0: a3(&t.point0), e1: a3(&(t.point1 - t.point0)), e2: a3(&(t.point2 - t.point0)), mat: t.mp.describe(f), prim_id: f.prim_id() });
}
pub fn flatten_list(l: &HittableList, f: &mut Flat) { for o in &l.objects { o.flatten(f); } }   // visit order of :151-164
/// 2^-30 of a box's size and position: no face of the content may coincide with a face of a box put in front of it.
fn cull_pad(mn: f64, mx: f64) -> f64 { mn.abs().max(mx.abs()).max(mx - mn).max(1e-30) / 1073741824.0 }
/// `full_bounds`: one more method to add to `trait Hittable` (default `None`): the box that holds the object over the
/// whole motion of every MovingSphere inside it (`bounding_box(time0, time1)` of each sphere's own interval, the plain
/// `bounding_box` of everything static, the union for lists and BVHNodes; `None` behind a Translate / RotateY). A node
/// whose stored box (built over the interval BVHNode::new was given, src/bvh.rs:83-88) contains it is marked
/// RT_INNER_CARRIED: the kernels may test it with the carried interval. One whose box does not (random_scene,
/// src/main.rs:315) keeps the reference's per-axis test and gets a box with the full bounds, marked RT_INNER_COVER, in front of it.
/// Restates `BVHNode::flatten` of raytracer_amd/csrc/host/scene.hpp, which decides the same from the time intervals.
pub fn flatten_bvh(n: &BVHNode, f: &mut Flat) {                             // src/bvh.rs:11-15, 101-111
    let (mn, mx) = (a3(&n._box._min), a3(&n._box._max));
    let full = n.full_bounds();
    let holds = match full { Some((lo, hi)) => (0..3).all(|a| lo[a] >= mn[a] && hi[a] <= mx[a]), None => false };
    // a node whose box contains the world's (the root first of all) is passed by every ray from inside the scene: no entry,
    // its children take its place (world space only, not for a box flat on an axis, which the reference never hits)
    let world = *f.world_box.get_or_insert((mn, mx));
    if holds && f.xf_stack.is_empty() && !f.in_medium && (0..3).all(|a| mn[a] != mx[a] && mn[a] <= world.0[a] && mx[a] >= world.1[a]) {
        n.left.flatten(f);
        if !Arc::ptr_eq(&n.left, &n.right) { n.right.flatten(f); }
        return;
    }
    let mut cover = None;
    if !holds {
        if let Some((lo, hi)) = full {
            let c = f.push_node(RT_N_INNER, RT_INNER_COVER);
            for a in 0..3 {                                                  // padded, then rounded outward
                let pad = cull_pad(lo[a], hi[a]);
                f.nodes[c].lo[a] = f32_down(lo[a] - pad);
                f.nodes[c].hi[a] = f32_up(hi[a] + pad);
            }
            cover = Some(c);
        }
    }
    let at = f.push_node(RT_N_INNER, if holds { RT_INNER_CARRIED } else { 0 });
    // a box flat on an axis is never hit (src/aabb.rs:25): stored inverted instead of rounded outward
    let flat = (0..3).any(|a| mn[a] == mx[a]);
    for a in 0..3 {
        f.nodes[at].lo[a] = if flat { f32::INFINITY } else { f32_down(mn[a]) };
        f.nodes[at].hi[a] = if flat { f32::NEG_INFINITY } else { f32_up(mx[a]) };
    }
    let outer = f.enclosing.replace((mn, mx));
    n.left.flatten(f);
    if !Arc::ptr_eq(&n.left, &n.right) { n.right.flatten(f); }              // span-1 node holds one object twice (:39-44)
    f.enclosing = outer;
    f.nodes[at].link = f.nodes.len() as u32;
    if let Some(c) = cover { f.nodes[c].link = f.nodes.len() as u32; }
}
/// A tree built by `rt_bvh_build` on the GPU instead of `BVHNode::new` (src/bvh.rs:17-98): the object list and the
/// tree as pre-order tokens. It implements `Hittable` only to be flattened and bounded; `hit` is never called on
/// the host. Restates `DeviceBvh` of raytracer_amd/csrc/host/scene.hpp.
pub struct DeviceBvh { pub objects: Vec<Arc<dyn Hittable>>, pub tokens: Vec<RtNode>, pub _box: AABB }
impl DeviceBvh {
    pub fn new(objects: Vec<Arc<dyn Hittable>>, time0: f64, time1: f64) -> Result<DeviceBvh, String> {
        let mut boxes = Vec::with_capacity(objects.len() * 6);
        let mut whole = AABB::default();
        for (i, o) in objects.iter().enumerate() {
            let mut b = AABB::default();
            if !o.bounding_box(time0, time1, &mut b) { return Err("object without a bounding box".into()); }   // src/bvh.rs:83-87
            boxes.extend_from_slice(&[b._min.x, b._min.y, b._min.z, b._max.x, b._max.y, b._max.z]);
            whole = if i == 0 { b } else { surrounding_box(&whole, &b) };                                       // src/aabb.rs:33-50
        }
        let mut tokens = vec![RtNode { lo: [0.0; 3], hi: [0.0; 3], meta: 0, link: 0 }; 2 * objects.len() - 1];
        check(unsafe { rt_bvh_build(boxes.as_ptr(), objects.len() as u64, tokens.as_mut_ptr(), std::ptr::null_mut()) })?;
        Ok(DeviceBvh { objects, tokens, _box: whole })
    }
}
pub fn flatten_device_bvh(n: &DeviceBvh, f: &mut Flat) {
    let mut open: Vec<(usize, u32)> = Vec::new();                           // (entry, token after its subtree)
    for (k, t) in n.tokens.iter().enumerate() {
        if t.meta == RT_BVH_TOKEN_LEAF {
            n.objects[t.link as usize].flatten(f);
        } else {
            let at = f.push_node(RT_N_INNER, 0);
            f.nodes[at].lo = t.lo;
            f.nodes[at].hi = t.hi;
            open.push((at, t.link));
        }
        while let Some(&(at, end)) = open.last() {
            if end as usize != k + 1 { break; }
            f.nodes[at].link = f.nodes.len() as u32;
            open.pop();
        }
    }
}
pub fn flatten_translate(t: &Translate, f: &mut Flat) {                     // src/hittable.rs:262-265
    let x = RtXform { kind: 0, depth: 0, anc: [RT_NONE; 4], inst_id: 0, _pad: 0, sin_theta: 0.0, cos_theta: 0.0, offset: a3(&t.offset) };
    f.push_xform(x, &*t.ptr);
}
pub fn flatten_rotate_y(r: &RotateY, f: &mut Flat) {                        // :306-312
    let x = RtXform { kind: 1, depth: 0, anc: [RT_NONE; 4], inst_id: 0, _pad: 0, sin_theta: r.sin_theta, cos_theta: r.cos_theta, offset: [0.0; 3] };
    f.push_xform(x, &*r.ptr);
}
pub fn flatten_medium(m: &ConstantMedium, f: &mut Flat) {                   // src/constant_medium.rs:10-14
    assert!(!f.in_medium, "a ConstantMedium inside the boundary of another is not supported");
    let idx = match f.seen.get(&key(m)) {
        Some(&i) => i,
        None => {
            let row = RtMedium { neg_inv_density: m.neg_inv_density, mat: m.phase_function.describe(f), rng_id: f.next_medium, prim_id: f.prim_id(), _pad: 0 };
            f.next_medium += 1;
            f.media.push(row);
            let i = (f.media.len() - 1) as u32;
            f.seen.insert(key(m), i);
            i
        }
    };
    let at = f.push_node(RT_N_MEDIUM, idx);
    f.in_medium = true;
    m.boundary.flatten(f);
    f.in_medium = false;
    f.nodes[at].link = f.nodes.len() as u32;
}
// In each `impl Hittable for X` add:  fn flatten(&self, f: &mut Flat) { crate::flatten::flatten_x(self, f) }
// (Box6 passes the material of any of its sides: `flatten_box6(self, &side_material, f)`; keeping the
// `Arc<dyn Material>` handed to Box6::new (src/box6.rs:16) in a new field is the simplest way to have it.)

// ---- Material::describe / Texture::describe -------------------------------------------------------------------
fn material(f: &mut Flat, addr: usize, make: impl FnOnce(&mut Flat) -> RtMaterial) -> u32 {
    if let Some(&i) = f.seen.get(&addr) { return i; }
    let row = make(f);
    f.materials.push(row);
    let i = (f.materials.len() - 1) as u32;
    f.seen.insert(addr, i);
    i
}
pub fn describe_lambertian(m: &Lambertian, f: &mut Flat) -> u32 { material(f, key(m), |f| RtMaterial { kind: 0, tex: m.albedo.describe(f), albedo: [0.0; 3], param: 0.0 }) }
pub fn describe_metal(m: &Metal, f: &mut Flat) -> u32 { material(f, key(m), |_| RtMaterial { kind: 1, tex: RT_NONE, albedo: a3(&m.albedo), param: m.fuzz }) }
pub fn describe_dielectric(m: &Dielectric, f: &mut Flat) -> u32 { material(f, key(m), |_| RtMaterial { kind: 2, tex: RT_NONE, albedo: [0.0; 3], param: m.ref_idx }) }
pub fn describe_diffuse_light(m: &DiffuseLight, f: &mut Flat) -> u32 { material(f, key(m), |f| RtMaterial { kind: 3, tex: m.emit.describe(f), albedo: [0.0; 3], param: 0.0 }) }
pub fn describe_isotropic(m: &Isotropic, f: &mut Flat) -> u32 { material(f, key(m), |f| RtMaterial { kind: 4, tex: m.albedo.describe(f), albedo: [0.0; 3], param: 0.0 }) }

fn texture(f: &mut Flat, addr: usize, make: impl FnOnce(&mut Flat) -> RtTexture) -> u32 {
    if let Some(&i) = f.seen.get(&addr) { return i; }
    let row = make(f);
    f.textures.push(row);
    let i = (f.textures.len() - 1) as u32;
    f.seen.insert(addr, i);
    i
}
pub fn describe_solid(t: &SolidColor, f: &mut Flat) -> u32 { texture(f, key(t), |_| RtTexture { kind: 0, aux: 0, a: a3(&t.color_value), b: [0.0; 3] }) }
/// CheckerTexture holds two textures (src/texture.rs:28-31); the scenes only ever put solid colours there
/// (src/main.rs:246-249), which is what the table row can express: a = even, b = odd.
pub fn describe_checker(t: &CheckerTexture, f: &mut Flat) -> u32 {
    let zero = crate::vec3::Point::zero();
    texture(f, key(t), |_| RtTexture { kind: 1, aux: 0, a: a3(&t.even.value(0.0, 0.0, &zero)), b: a3(&t.odd.value(0.0, 0.0, &zero)) })
}
pub fn describe_noise(t: &NoiseTexture, f: &mut Flat) -> u32 {
    texture(f, key(t), |f| {
        let p: &Perlin = &t.noise;
        let mut tab = RtPerlin { ran_vec: [[0.0; 3]; 256], perm_x: p.perm_x, perm_y: p.perm_y, perm_z: p.perm_z };
        for i in 0..256 { tab.ran_vec[i] = a3(&p.ran_vec[i]); }
        f.perlins.push(tab);
        RtTexture { kind: 2, aux: (f.perlins.len() - 1) as u32, a: [t.scale, 0.0, 0.0], b: [0.0; 3] }
    })
}
pub fn describe_image(t: &ImageTexture, f: &mut Flat) -> u32 {
    texture(f, key(t), |f| {
        let offset = f.texels.len() as u64;
        for y in 0..t.height { for x in 0..t.width { let px = t.img.get_pixel(x, y); f.texels.extend_from_slice(&[px[0], px[1], px[2]]); } }   // src/texture.rs:107
        f.images.push(RtImage { width: t.width, height: t.height, offset });
        RtTexture { kind: 3, aux: (f.images.len() - 1) as u32, a: [0.0; 3], b: [0.0; 3] }
    })
}

// ---- the render call that replaces src/main.rs:193-230 --------------------------------------------------------
pub fn camera_desc(c: &Camera) -> RtCamera {                                // src/camera.rs:6-17
    RtCamera { orig: a3(&c.orig), lower_left_corner: a3(&c.lower_left_corner), horizontal: a3(&c.horizontal), vertical: a3(&c.vertical),
               u: a3(&c.u), v: a3(&c.v), lens_radius: c.lens_radius, time0: c.time0, time1: c.time1 }
}

fn check(rc: i32) -> Result<(), String> {
    if rc == 0 { Ok(()) } else { Err(unsafe { std::ffi::CStr::from_ptr(rt_last_error()) }.to_string_lossy().into_owned()) }
}

/// Returns width * height * 3 bytes, row 0 first, ready for `img.put_pixel(x, y, ...)` (src/main.rs:226).
pub fn render_with_hip(world: &dyn Hittable, cam: &Camera, background: &Vec3, width: u32, height: u32,
                       samples_per_pixel: u32, max_depth: u32, seed: u64) -> Result<Vec<u8>, String> {
    let mut flat = Flat::default();
    world.flatten(&mut flat);
    let desc = flat.desc();
    let mut scene: *mut RtScene = std::ptr::null_mut();
    check(unsafe { rt_scene_create(&desc, &mut scene) })?;
    let params = RtRenderParams { width, height, sample_begin: 0, sample_end: samples_per_pixel, max_depth, samples_per_batch: 0,
                                  seed, background: a3(background) };
    let mut accum = vec![0f32; (width as usize) * (height as usize) * 3];
    let mut rgb = vec![0u8; accum.len()];
    let rendered = check(unsafe { rt_render(scene, &camera_desc(cam), &params, accum.as_mut_ptr(), std::ptr::null_mut()) })
        .and_then(|_| check(unsafe { rt_resolve_u8(accum.as_ptr(), width, height, samples_per_pixel as u64, rgb.as_mut_ptr()) }));
    unsafe { rt_scene_destroy(scene) };
    rendered.map(|_| rgb)
}
