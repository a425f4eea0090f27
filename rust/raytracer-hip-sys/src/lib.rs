//! Declarations of `librt_hip.so` (include/rt/abi.h) for the reference's Rust host.
//! Documentation only in this repository: the build image has no Rust toolchain, so this
//! crate has never been compiled. Layouts mirror include/rt/abi.h field for field.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] #[derive(Clone, Copy)] pub struct RtNode { pub lo: [f32; 3], pub hi: [f32; 3], pub meta: u32, pub link: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtSphere { pub c: [f64; 3], pub r: f64, pub mat: u32, pub prim_id: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtMovingSphere { pub c0: [f64; 3], pub c1: [f64; 3], pub t0: f64, pub t1: f64, pub r: f64, pub mat: u32, pub prim_id: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtRect { pub a0: f64, pub a1: f64, pub b0: f64, pub b1: f64, pub k: f64, pub axis: u32, pub mat: u32, pub prim_id: u32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtBox6 { pub p0: [f64; 3], pub p1: [f64; 3], pub mat: u32, pub prim_id: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtTriangle { pub p0: [f64; 3], pub e1: [f64; 3], pub e2: [f64; 3], pub mat: u32, pub prim_id: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtXform { pub kind: u32, pub depth: u32, pub anc: [u32; 4], pub inst_id: u32, pub _pad: u32, pub sin_theta: f64, pub cos_theta: f64, pub offset: [f64; 3] }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtMedium { pub neg_inv_density: f64, pub mat: u32, pub rng_id: u32, pub prim_id: u32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtMaterial { pub kind: u32, pub tex: u32, pub albedo: [f64; 3], pub param: f64 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtTexture { pub kind: u32, pub aux: u32, pub a: [f64; 3], pub b: [f64; 3] }
#[repr(C)] pub struct RtPerlin { pub ran_vec: [[f64; 3]; 256], pub perm_x: [i32; 256], pub perm_y: [i32; 256], pub perm_z: [i32; 256] }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtImage { pub width: u32, pub height: u32, pub offset: u64 }

#[repr(C)]
pub struct RtSceneDesc {
    pub abi_version: u32, pub flags: u32,
    pub nodes: *const RtNode, pub n_nodes: u64,
    pub spheres: *const RtSphere, pub n_spheres: u64,
    pub msphere: *const RtMovingSphere, pub n_msphere: u64,
    pub rects: *const RtRect, pub n_rects: u64,
    pub boxes: *const RtBox6, pub n_boxes: u64,
    pub triangles: *const RtTriangle, pub n_triangles: u64,
    pub xforms: *const RtXform, pub n_xforms: u64,
    pub media: *const RtMedium, pub n_media: u64,
    pub materials: *const RtMaterial, pub n_materials: u64,
    pub textures: *const RtTexture, pub n_textures: u64,
    pub perlins: *const RtPerlin, pub n_perlins: u64,
    pub images: *const RtImage, pub n_images: u64,
    pub texels: *const u8, pub n_texel_bytes: u64,
}

#[repr(C)] #[derive(Clone, Copy)] pub struct RtRay { pub o: [f64; 3], pub d: [f64; 3], pub tm: f64, pub tmin: f64, pub tmax: f64, pub pixel: u32, pub sample: u32, pub bounce: u32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtHit { pub prim_id: u32, pub sub: u32, pub t: f64, pub u: f64, pub v: f64 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtHitRecord { pub p: [f64; 3], pub normal: [f64; 3], pub t: f64, pub u: f64, pub v: f64, pub prim_id: u32, pub sub: u32, pub mat: u32, pub front_face: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtPathStep { pub emitted: [f64; 3], pub attenuation: [f64; 3], pub o: [f64; 3], pub d: [f64; 3], pub t: f64, pub prim_id: u32, pub sub: u32, pub scattered: u32, pub _pad: u32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtCamera { pub orig: [f64; 3], pub lower_left_corner: [f64; 3], pub horizontal: [f64; 3], pub vertical: [f64; 3], pub u: [f64; 3], pub v: [f64; 3], pub lens_radius: f64, pub time0: f64, pub time1: f64 }
#[repr(C)] #[derive(Clone, Copy)] pub struct RtRenderParams { pub width: u32, pub height: u32, pub sample_begin: u32, pub sample_end: u32, pub max_depth: u32, pub samples_per_batch: u32, pub seed: u64, pub background: [f64; 3] }
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct RtTraceStats { pub ms_kernel: f64 }
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct RtRenderStats { pub paths: u64, pub rays: u64, pub ms_total: f64, pub ms_extend: f64, pub ms_shade: f64, pub batches: u32, pub _pad: u32 }
pub enum RtScene {}
pub enum RtShardedScene {}

#[link(name = "rt_hip")]
extern "C" {
    pub fn rt_abi_version() -> u32;
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_device_count() -> c_int;
    pub fn rt_scene_create(desc: *const RtSceneDesc, out: *mut *mut RtScene) -> c_int;
    pub fn rt_scene_destroy(scene: *mut RtScene);
    pub fn rt_trace_closest(scene: *mut RtScene, rays: *const RtRay, n: u64, seed: u64, out: *mut RtHit, stats: *mut RtTraceStats) -> c_int;
    pub fn rt_trace_records(scene: *mut RtScene, rays: *const RtRay, n: u64, seed: u64, out: *mut RtHitRecord) -> c_int;
    pub fn rt_trace_closest_device(scene: *mut RtScene, rays_dev: *const RtRay, n: u64, seed: u64, out_dev: *mut RtHit, stream: *mut c_void) -> c_int;
    pub fn rt_trace_work(scene: *mut RtScene, rays: *const RtRay, n: u64, seed: u64, out: *mut RtHit, work: *mut u64) -> c_int;
    pub fn rt_render(scene: *mut RtScene, cam: *const RtCamera, params: *const RtRenderParams, accum_rgb: *mut f32, stats: *mut RtRenderStats) -> c_int;
    pub fn rt_render_device(scene: *mut RtScene, cam: *const RtCamera, params: *const RtRenderParams, accum_rgb_dev: *mut f32, stream: *mut c_void, stats: *mut RtRenderStats) -> c_int;
    pub fn rt_render_work(scene: *mut RtScene, cam: *const RtCamera, params: *const RtRenderParams, accum_rgb_dev: *mut f32, stream: *mut c_void, stats: *mut RtRenderStats, work: *mut u64) -> c_int;
    pub fn rt_path_step(scene: *mut RtScene, params: *const RtRenderParams, profile: c_int, rays: *const RtRay, n: u64, out: *mut RtPathStep, work: *mut u64) -> c_int;
    pub fn rt_scene_profile(scene: *const RtScene) -> c_int;
    pub fn rt_resolve_u8(accum_rgb: *const f32, width: u32, height: u32, spp: u64, rgb_out: *mut u8) -> c_int;
    pub fn rt_resolve_u8_device(accum_rgb_dev: *const f32, width: u32, height: u32, spp: u64, rgb_out_dev: *mut u8, stream: *mut c_void) -> c_int;
    /// boxes: n x 6 f64 (min xyz, max xyz); tokens_out: 2 n - 1 records in pre-order (leaf: meta == RT_BVH_TOKEN_LEAF, link = object)
    pub fn rt_bvh_build(boxes: *const f64, n: u64, tokens_out: *mut RtNode, ms_device: *mut f64) -> c_int;
    pub fn rt_bvh_build_device(boxes_dev: *const f64, n: u64, tokens_dev: *mut RtNode, flag_dev: *mut u32, stream: *mut c_void, ms_device: *mut f64) -> c_int;
    pub fn rt_bvh_expand_device(tokens_dev: *const RtNode, n: u64, obj_entries_dev: *const RtNode, obj_offsets_dev: *const u32, n_entries_out: u64, out_dev: *mut RtNode, stream: *mut c_void) -> c_int;
    pub fn rt_bvh_build_stream(boxes: *const f64, n: u64, obj_entries: *const RtNode, obj_offsets: *const u32, out: *mut RtNode, ms_device: *mut f64) -> c_int;
    pub fn rt_bvh_refit_device(tokens_dev: *mut RtNode, n: u64, boxes_dev: *const f64, flag_dev: *mut u32, stream: *mut c_void) -> c_int;
    /// the same without any allocation: scratch memory in a static device array, at most RT_BVH_NOALLOC_MAX_OBJECTS objects
    pub fn rt_bvh_build_device_noalloc(boxes_dev: *const f64, n: u64, tokens_dev: *mut RtNode, flag_dev: *mut u32, stream: *mut c_void) -> c_int;
    /// one replica of the scene per GPU of this process; device i renders rt_shard_samples(.., i, G) of every pixel, one RCCL all-reduce
    pub fn rt_sharded_create(desc: *const RtSceneDesc, devices: *const c_int, n_devices: c_int, out: *mut *mut RtShardedScene) -> c_int;
    pub fn rt_sharded_destroy(scene: *mut RtShardedScene);
    pub fn rt_sharded_device_count(scene: *const RtShardedScene) -> c_int;
    pub fn rt_sharded_render(scene: *mut RtShardedScene, cam: *const RtCamera, params: *const RtRenderParams, accum_rgb: *mut f32, stats: *mut RtRenderStats) -> c_int;
    pub fn rt_shard_samples(sample_begin: u32, sample_end: u32, rank: u32, world: u32, lo: *mut u32, hi: *mut u32);
}
pub const RT_BVH_TOKEN_LEAF: u32 = 0xF;
pub const RT_BVH_NOALLOC_MAX_OBJECTS: u32 = 65536;
pub const RT_DESC_BOXES_MAY_NOT_BOUND: u32 = 1;
pub const RT_INNER_CARRIED: u32 = 1;
pub const RT_INNER_COVER: u32 = 2;
pub const RT_PROFILE_OWN: c_int = -1;
