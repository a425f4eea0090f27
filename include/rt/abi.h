/*
 * C ABI of the MI355X (gfx950) path-tracing library `librt_hip.so`.
 *
 * The reference (Mighty-A/RayTracer, Rust) has no FFI boundary: its hot path is
 * the thread closure of src/main.rs:193-230, which calls Camera::get_ray
 * (src/camera.rs:61-71) and ray_color (src/color.rs:34-63), which in turn calls
 * Hittable::hit (src/hittable.rs:43-46), Material::scatter/emitted
 * (src/material.rs:8-20) and Texture::value (src/texture.rs:8-10). The entry
 * points below are what a `raytracer-hip-sys` crate would bind to replace that
 * closure: the host keeps building scenes with the reference's constructors,
 * flattens the Arc<dyn Hittable> tree into the tables of RtSceneDesc, and calls
 * rt_render instead of looping over pixels.
 *
 * Conventions: plain pointers and sizes; inputs are borrowed for the duration of
 * the call; outputs are caller-allocated; RtScene is library-owned and immutable.
 * Every function returns 0 on success and a negative RT_E* code otherwise (the
 * reference panics instead: src/bvh.rs:86,124, src/vec3.rs:285-286); nothing
 * unwinds or aborts across the boundary. rt_last_error() returns a thread-local
 * message for the last failure on the calling thread. A scene lives on the HIP
 * device that was current when it was created; calls on one scene are
 * serialised by the caller, calls on different scenes may run concurrently.
 *
 * All arithmetic is IEEE binary64 without fused multiply-add contraction, as in
 * the reference; BVH boxes are binary32 rounded outward.
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 2u   /* 2: RtSceneDesc.flags (was padding) and the box markers in RtNode.meta are validated */
#define RT_MISS 0xFFFFFFFFu
#define RT_NONE 0xFFFFFFFFu

enum RtError {
    RT_OK = 0,
    RT_E_INVALID = -1,   /* bad argument / malformed scene description */
    RT_E_DEVICE = -2,    /* HIP runtime error (no device, launch failure, ...) */
    RT_E_NOMEM = -3,     /* host or device allocation failed */
    RT_E_UNSUPPORTED = -4
};

/* ---- flattened scene ---------------------------------------------------- */

/* One 32-byte entry of the traversal stream. The stream is the depth-first,
 * left-then-right linearisation of the reference's object tree, so walking it
 * front to back visits objects in the order BVHNode::hit (src/bvh.rs:101-111)
 * and HittableList::hit (src/hittable.rs:151-164) do.
 *   meta = kind | index << 4
 *   RT_N_INNER   : lo/hi = box of a BVHNode (src/bvh.rs:14), link = entry to
 *                  continue at when the box is missed (one past its subtree);
 *                  index = 0; or RT_INNER_CARRIED (1): the box of a BVHNode that
 *                  holds everything in its subtree while the moving spheres are
 *                  within their motion intervals; or RT_INNER_COVER (2): a box
 *                  that is not the reference's, put in front of a node whose own
 *                  box does not hold its subtree, and that does. Only looked at
 *                  in a scene flagged RT_DESC_BOXES_MAY_NOT_BOUND: for a ray whose
 *                  time lies within the motion interval of every moving sphere of
 *                  the scene, boxes 1 and 2 are tested with the carried interval
 *                  like the boxes of any other scene; box 0, and box 1 for any
 *                  other ray, with the reference's per-axis rule; box 2 lets any
 *                  other ray pass
 *   RT_N_SPHERE .. RT_N_TRIANGLE : index into the matching primitive table
 *   RT_N_XF_ENTER: index into xforms; the ray is moved into the wrapper's
 *                  object space (src/hittable.rs:278-282, 355-368)
 *   RT_N_XF_EXIT : index of the wrapper being left
 *   RT_N_MEDIUM  : index into media; entries [i+1, link) are the boundary
 *                  object (src/constant_medium.rs:11), queried twice
 * lo/hi are unused (0) except for RT_N_INNER.
 * The ranges [i, link) of RT_N_INNER and RT_N_MEDIUM entries nest, with each other
 * and with the wrappers: a wrapper entered inside a range is left inside it, and a
 * range begun inside a wrapper ends before the wrapper is left (a depth-first
 * linearisation gives this by construction; rt_scene_create refuses a stream that
 * does not have it). */
typedef struct RtNode {
    float lo[3];
    float hi[3];
    uint32_t meta;
    uint32_t link;
} RtNode;

#define RT_INNER_CARRIED 1u
#define RT_INNER_COVER 2u

enum RtNodeKind {
    RT_N_INNER = 0, RT_N_SPHERE = 1, RT_N_MSPHERE = 2, RT_N_RECT = 3, RT_N_BOX6 = 4,
    RT_N_TRIANGLE = 5, RT_N_XF_ENTER = 6, RT_N_XF_EXIT = 7, RT_N_MEDIUM = 8
};

typedef struct RtSphere {        /* src/hittable.rs:48-52 */
    double c[3]; double r; uint32_t mat; uint32_t prim_id;
} RtSphere;
typedef struct RtMovingSphere {  /* src/hittable.rs:187-194 */
    double c0[3]; double c1[3]; double t0, t1, r; uint32_t mat; uint32_t prim_id;
} RtMovingSphere;
typedef struct RtRect {          /* src/aarect.rs:7-14, 58-65, 109-116; axis 0 = XY (z=k), 1 = XZ (y=k), 2 = YZ (x=k) */
    double a0, a1, b0, b1, k; uint32_t axis; uint32_t mat; uint32_t prim_id; uint32_t _pad;
} RtRect;
typedef struct RtBox6 {          /* src/box6.rs:9-13; the six sides are implied, in src/box6.rs:22-71 order */
    double p0[3]; double p1[3]; uint32_t mat; uint32_t prim_id;
} RtBox6;
typedef struct RtTriangle {      /* src/triangle.rs:7-12 with e1 = point1 - point0, e2 = point2 - point0 (:27-28) */
    double p0[3]; double e1[3]; double e2[3]; uint32_t mat; uint32_t prim_id;
} RtTriangle;
typedef struct RtXform {         /* src/hittable.rs:262-265 (kind 0, Translate), 306-312 (kind 1, RotateY) */
    uint32_t kind; uint32_t depth;   /* depth = number of wrappers from the outermost down to this one */
    uint32_t anc[4];                 /* xform indices, outermost first; anc[depth-1] is this entry */
    uint32_t inst_id; uint32_t _pad;
    double sin_theta, cos_theta; double offset[3];
} RtXform;
typedef struct RtMedium {        /* src/constant_medium.rs:10-14 */
    double neg_inv_density; uint32_t mat; uint32_t rng_id; uint32_t prim_id; uint32_t _pad;
} RtMedium;

enum RtMaterialKind { RT_M_LAMBERTIAN = 0, RT_M_METAL = 1, RT_M_DIELECTRIC = 2, RT_M_DIFFUSE_LIGHT = 3, RT_M_ISOTROPIC = 4 };
typedef struct RtMaterial {      /* src/material.rs:22-24, 53-56, 86-88, 150-152, 181-183 */
    uint32_t kind; uint32_t tex;     /* tex: albedo / emit texture (Lambertian, DiffuseLight, Isotropic) */
    double albedo[3];                /* Metal */
    double param;                    /* Metal: fuzz (already clamped to <= 1), Dielectric: ref_idx */
} RtMaterial;

enum RtTextureKind { RT_T_SOLID = 0, RT_T_CHECKER = 1, RT_T_NOISE = 2, RT_T_IMAGE = 3 };
typedef struct RtTexture {       /* src/texture.rs:12-14, 28-31, 53-56, 75-79 */
    uint32_t kind; uint32_t aux;     /* aux: Noise -> perlin index, Image -> image index */
    double a[3];                     /* Solid: colour; Checker: even colour; Noise: a[0] = scale */
    double b[3];                     /* Checker: odd colour */
} RtTexture;
typedef struct RtPerlin {        /* src/perlin.rs:6-11 */
    double ran_vec[256][3]; int32_t perm_x[256]; int32_t perm_y[256]; int32_t perm_z[256];
} RtPerlin;
typedef struct RtImage {         /* src/texture.rs:75-79; texels are RGB u8 rows, top row first */
    uint32_t width, height; uint64_t offset;   /* byte offset into RtSceneDesc.texels */
} RtImage;

/* RtSceneDesc.flags. BOXES_MAY_NOT_BOUND: some BVHNode box does not bound what is under it (the reference's
 * random_scene builds its tree over t in (0, 0.1) while its spheres move over (0, 1): src/main.rs:315, 267-279).
 * The closest hit then depends on the box rule, and the kernels run the reference's own per-axis rule
 * (src/aabb.rs:15-30) on such a scene instead of the tighter carried-interval rule they use everywhere else
 * (the two agree whenever boxes bound their contents). Boxes of such a scene that do hold their content say so
 * (RT_INNER_CARRIED) and keep the carried-interval rule; the flattener also puts a box of that kind, built over the
 * whole motion of what is below, in front of every box that does not: a ray that misses it cannot hit anything
 * below, whichever way the reference's test on the node's own box would have gone.
 * A scene without the flag is walked with the carried interval throughout, which takes every ray's time to lie within
 * the motion interval of every MovingSphere (true of the reference's cameras and scenes: both run over the shutter):
 * MovingSphere::center extrapolates outside that interval (src/hittable.rs:208-211), out of the boxes built for it,
 * where the reference's per-axis rule may still find the sphere and the carried interval does not. */
#define RT_DESC_BOXES_MAY_NOT_BOUND 1u

typedef struct RtSceneDesc {
    uint32_t abi_version; uint32_t flags;
    const RtNode* nodes; uint64_t n_nodes;
    const RtSphere* spheres; uint64_t n_spheres;
    const RtMovingSphere* msphere; uint64_t n_msphere;
    const RtRect* rects; uint64_t n_rects;
    const RtBox6* boxes; uint64_t n_boxes;
    const RtTriangle* triangles; uint64_t n_triangles;
    const RtXform* xforms; uint64_t n_xforms;
    const RtMedium* media; uint64_t n_media;
    const RtMaterial* materials; uint64_t n_materials;
    const RtTexture* textures; uint64_t n_textures;
    const RtPerlin* perlins; uint64_t n_perlins;
    const RtImage* images; uint64_t n_images;
    const uint8_t* texels; uint64_t n_texel_bytes;
} RtSceneDesc;

/* ---- rays, hits, camera, parameters ------------------------------------- */

typedef struct RtRay {           /* src/ray.rs:3-7 + the t range of Hittable::hit + the RNG key of the path */
    double o[3]; double d[3]; double tm; double tmin; double tmax;
    uint32_t pixel, sample, bounce, _pad;   /* keys the ConstantMedium draw (src/constant_medium.rs:69) */
} RtRay;

/* sub = box side (bits 0-7, src/box6.rs:22-71 order) | (inst_id + 1) << 8, where
 * inst_id is the creation-order id of the innermost Translate/RotateY around the
 * primitive (0 in those bits when there is none). prim_id = RT_MISS on a miss. */
typedef struct RtHit {
    uint32_t prim_id; uint32_t sub; double t; double u; double v;
} RtHit;

typedef struct RtHitRecord {     /* src/hittable.rs:11-19 as ray_color sees it after world.hit */
    double p[3]; double normal[3]; double t, u, v;
    uint32_t prim_id, sub, mat, front_face;
} RtHitRecord;

typedef struct RtCamera {        /* src/camera.rs:6-17 (without the unused _w) */
    double orig[3]; double lower_left_corner[3]; double horizontal[3]; double vertical[3];
    double u[3]; double v[3]; double lens_radius, time0, time1;
} RtCamera;

typedef struct RtRenderParams {
    uint32_t width, height;          /* src/main.rs:164-165 */
    uint32_t sample_begin, sample_end;   /* samples [begin, end) of every pixel (src/main.rs:205) */
    uint32_t max_depth;              /* src/main.rs:49 */
    uint32_t samples_per_batch;      /* samples of every pixel traced per wavefront batch; 0 = as many as fit
                                        640 Mi path slots or half of the free device memory (232 bytes per slot) */
    uint64_t seed;
    double background[3];            /* src/main.rs:62-162 */
} RtRenderParams;

typedef struct RtTraceStats {
    double ms_kernel;                /* device time of the traversal kernel */
} RtTraceStats;

typedef struct RtRenderStats {
    uint64_t paths;                  /* path samples started */
    uint64_t rays;                   /* closest-hit queries issued (all bounces) */
    double ms_total;                 /* device time of the whole call, all kernels */
    double ms_extend;                /* device time inside the extend (closest-hit) kernel */
    double ms_shade;                 /* device time inside the shade kernel */
    uint32_t batches; uint32_t _pad;
} RtRenderStats;

typedef struct RtScene RtScene;

/* ---- entry points -------------------------------------------------------- */

uint32_t rt_abi_version(void);
const char* rt_last_error(void);
int rt_device_count(void);

/* Replaces the shallow Arc clone of the world into each render thread
 * (src/main.rs:198): copies the tables to the HBM of the current HIP device. */
int rt_scene_create(const RtSceneDesc* desc, RtScene** out);
void rt_scene_destroy(RtScene* scene);

/* Replaces `world.hit(r, t_min, t_max, &mut rec)` (src/color.rs:41 ->
 * src/bvh.rs:101-111) for n independent rays. `seed` keys the ConstantMedium
 * draws together with each ray's (pixel, sample, bounce). Host buffers. */
int rt_trace_closest(RtScene* scene, const RtRay* rays, uint64_t n, uint64_t seed, RtHit* out, RtTraceStats* stats);
/* Same, also returning p / normal / front_face / material as ray_color uses them. */
int rt_trace_records(RtScene* scene, const RtRay* rays, uint64_t n, uint64_t seed, RtHitRecord* out);
/* Same as rt_trace_closest with rays and hits already resident in device memory;
 * `stream` is a hipStream_t (NULL = default stream). Asynchronous. */
int rt_trace_closest_device(RtScene* scene, const RtRay* rays_dev, uint64_t n, uint64_t seed, RtHit* out_dev, void* stream);
/* Instrumented rt_trace_closest: also counts the traversal work of the whole call.
 * work[0] = stream entries fetched, work[1 + k] = entries of RtNodeKind k visited
 * (k = RT_N_INNER counts box tests). 10 words. */
int rt_trace_work(RtScene* scene, const RtRay* rays, uint64_t n, uint64_t seed, RtHit* out, uint64_t* work);

/* Replaces the per-pixel sample loop (src/main.rs:201-211): for every pixel adds
 * the sum of ray_color over samples [sample_begin, sample_end) to
 * accum_rgb[(y*width + x)*3 + c]. rt_render takes a host buffer (W*H*3 floats,
 * read-modify-write); rt_render_device takes a device buffer and a hipStream_t,
 * returns once the work is enqueued unless `stats` is non-NULL (then it waits). */
int rt_render(RtScene* scene, const RtCamera* cam, const RtRenderParams* params, float* accum_rgb, RtRenderStats* stats);
int rt_render_device(RtScene* scene, const RtCamera* cam, const RtRenderParams* params, float* accum_rgb_dev, void* stream,
                     RtRenderStats* stats);

/* Instrumented rt_render_device: same image, and the traversal work of all extend launches of the call summed
 * into work[0 .. 9] (layout as in rt_trace_work); work[10] = path slots that gathered light (whose radiance
 * columns the shade launches read, added to and wrote back); work[11], work[12] = the part of work[1 + RT_N_MEDIUM] and
 * work[1 + RT_N_SPHERE] that the closest-hit launches did not do themselves: media that are asked before the walk by
 * the kernel that makes the ray, and the Sphere::hit calls of their boundaries; work[13] = 0. 14 words. Waits for
 * completion. Used to derive the algorithmic bytes and operations of the closest-hit and shade launches. */
int rt_render_work(RtScene* scene, const RtCamera* cam, const RtRenderParams* params, float* accum_rgb_dev, void* stream,
                   RtRenderStats* stats, uint64_t* work);

/* What one iteration of ray_color (src/color.rs:41-62) does with one ray. */
typedef struct RtPathStep {
    double emitted[3];           /* what the iteration adds to the pixel at throughput 1: the background on a
                                    miss (:41-43), Material::emitted at a hit (:50) */
    double attenuation[3];       /* Material::scatter's attenuation (:52-54); 0 when not scattered */
    double o[3], d[3];           /* the scattered ray (its time is the incoming ray's); 0 when not scattered */
    double t;                    /* closest hit, as RtHit; 0 on a miss */
    uint32_t prim_id, sub;       /* as RtHit */
    uint32_t scattered;          /* 1: scatter returned true and the path goes on */
    uint32_t _pad;
} RtPathStep;

#define RT_PROFILE_OWN (-1)

/* One iteration of the bounce loop of ray_color (src/color.rs:41-62) for n caller-supplied rays, run by the very
 * launches rt_render makes for a bounce (the closest-hit kernel, then the shade kernel) instead of the stand-alone
 * kernels behind rt_trace_*: rays are laid out as the render path's ray records with throughput 1, and the
 * outcome of each is read back from the hit slots, the radiance columns and the survivors' records. Each ray's
 * (pixel, sample, bounce) keys its random draws as in a render with these `params` (width, height, sample range,
 * seed, background, max_depth); rays must have tmin = 0.0001 and tmax = +inf (src/color.rs:41), pixel < width *
 * height, sample in [sample_begin, sample_end), bounce < max_depth, and no two rays of one bounce may share
 * (pixel, sample). `profile` selects the variant of the closest-hit kernel: RT_PROFILE_OWN = the one rt_render
 * runs for this scene; otherwise a bit set (1: boxes alone inside wrappers handled in one step, 2: general walk of
 * medium boundaries, 4: lanes leave the box phase by vote) which must be one of 0, 1, 3, 4, 5, 7 and able to
 * handle the scene's media (RT_E_INVALID otherwise); on a scene flagged RT_DESC_BOXES_MAY_NOT_BOUND bit 8 (the
 * per-axis box rule) is implied and only 0 and 4 may be asked for. `work`, when not NULL, receives the traversal work of the
 * closest-hit launches (10 words as in rt_trace_work). Host buffers. */
int rt_path_step(RtScene* scene, const RtRenderParams* params, int profile, const RtRay* rays, uint64_t n, RtPathStep* out,
                 uint64_t* work);
/* The profile rt_render runs on this scene (the bit set described above), or a negative RtError. */
int rt_scene_profile(const RtScene* scene);

/* Replaces the gamma / clamp / u8 conversion (src/main.rs:212-225): out =
 * clamp(sqrt(accum / spp), 0, 0.999) * 255 as u8. Host buffers / device buffers. The host-buffer form is called once
 * per picture and has no scene to keep staging memory in: it allocates and frees its two device buffers per call; the
 * device form allocates nothing and is asynchronous on `stream`. */
int rt_resolve_u8(const float* accum_rgb, uint32_t width, uint32_t height, uint64_t spp, uint8_t* rgb_out);
int rt_resolve_u8_device(const float* accum_rgb_dev, uint32_t width, uint32_t height, uint64_t spp, uint8_t* rgb_out_dev,
                         void* stream);

/* ---- several GPUs in one process -------------------------------------------- */

/* Stands in for the worker threads of the reference's driver (src/main.rs:189-230: 8 threads, each owning a band of
 * image columns, merged by the caller): the scene is replicated in the HBM of every listed device, device i of G
 * renders samples rt_shard_samples(sample_begin, sample_end, i, G) of EVERY pixel on its own stream from its own
 * host thread, and one RCCL all-reduce (sum, float32, over xGMI) of the G accumulation buffers ends the render.
 * The picture does not depend on G except for the order of the f32 additions. */
typedef struct RtShardedScene RtShardedScene;
/* devices: n_devices HIP device ordinals, distinct; NULL = the first n_devices devices (n_devices <= 0: all).
 * RCCL is loaded at run time (librccl.so); with one device it is not needed. */
int rt_sharded_create(const RtSceneDesc* desc, const int* devices, int n_devices, RtShardedScene** out);
void rt_sharded_destroy(RtShardedScene* scene);
int rt_sharded_device_count(const RtShardedScene* scene);
/* As rt_render (host buffer, read-modify-write) over all devices of the scene. stats: paths and rays summed over
 * the devices, times = the slowest device's. */
int rt_sharded_render(RtShardedScene* scene, const RtCamera* cam, const RtRenderParams* params, float* accum_rgb, RtRenderStats* stats);
/* The partition used: samples [*lo, *hi) of [sample_begin, sample_end) belong to `rank` of `world`; contiguous,
 * disjoint, covering, sizes differing by at most one (the first (total mod world) ranks hold the larger share). */
void rt_shard_samples(uint32_t sample_begin, uint32_t sample_end, uint32_t rank, uint32_t world, uint32_t* lo, uint32_t* hi);

/* ---- tree construction on the device ------------------------------------- */

#define RT_BVH_TOKEN_LEAF 0xFu          /* meta of a leaf token (not an RtNodeKind) */
#define RT_BVH_MAX_OBJECTS (1u << 28)

/* Stands in for BVHNode::new (src/bvh.rs:17-98, a recursive stable sort per
 * node on the host: O(n log^2 n)) where its tree need not be reproduced: builds
 * a binary tree over n >= 2 objects on the current HIP device from their
 * bounding boxes alone. boxes = n x 6 doubles (min x y z, max x y z; what
 * Hittable::bounding_box, src/hittable.rs:45, returns). The objects are
 * ordered by the 63-bit Morton code of their box centres (ties: input order)
 * and the tree is the binary radix tree over the ordered codes; node boxes are
 * surrounding_box (src/aabb.rs:33-50) of the children's. The result is the
 * tree in pre-order, 2 n - 1 tokens in `tokens_out` (host memory):
 *   inner node: lo/hi = box rounded outward to f32 (inverted when flat on an
 *               axis), meta = RT_N_INNER, link = index of the first token after
 *               its subtree;
 *   leaf:       meta = RT_BVH_TOKEN_LEAF, link = position of the object in
 *               `boxes`.
 * The caller turns tokens into stream entries by replacing every leaf token
 * with the entries of that object and rebasing the links. One object per leaf,
 * left subtree first, as in BVHNode::hit (src/bvh.rs:101-111). Boxes with NaN
 * or without a finite centre are refused (RT_E_INVALID). `ms_device`, when not
 * NULL, receives the device time of the build without the copies. */
int rt_bvh_build(const double* boxes, uint64_t n, RtNode* tokens_out, double* ms_device);
/* The same tree (src/bvh.rs:17-98 stand-in) with boxes and tokens resident on the current HIP device: boxes_dev =
 * n x 6 doubles, tokens_dev = room for 2 n - 1 tokens; nothing crosses PCIe. The work is enqueued on `stream`
 * (a hipStream_t, NULL = default stream); the call returns once it has completed, because the scratch memory it
 * allocates for the build (about 200 bytes per object) is released on return. `flag_dev`, when not NULL, is a
 * device word that receives 1 if some box holds NaN or has no finite centre (the tokens are then well-formed but
 * meaningless) and 0 otherwise. */
int rt_bvh_build_device(const double* boxes_dev, uint64_t n, RtNode* tokens_dev, uint32_t* flag_dev, void* stream, double* ms_device);

/* Turns the tokens of rt_bvh_build_device into the entry stream of RtSceneDesc.nodes on the device: every leaf
 * token is replaced by the stream entries of its object and the links are translated (what the caller of
 * rt_bvh_build does on the host). obj_entries_dev: the entries of all objects, object after object, as a flattener
 * emits them when each object is flattened on its own (links of RT_N_INNER / RT_N_MEDIUM entries index into this
 * array); obj_offsets_dev: n + 1 offsets into it (object i owns [offsets[i], offsets[i + 1])); out_dev: room for
 * n_entries_out = (n - 1) + offsets[n] entries (checked). Waits for completion. */
int rt_bvh_expand_device(const RtNode* tokens_dev, uint64_t n, const RtNode* obj_entries_dev, const uint32_t* obj_offsets_dev,
                         uint64_t n_entries_out, RtNode* out_dev, void* stream);
/* rt_bvh_build + rt_bvh_expand_device with host buffers in and out; the tokens never leave the device.
 * out: room for (n - 1) + obj_offsets[n] entries. */
int rt_bvh_build_stream(const double* boxes, uint64_t n, const RtNode* obj_entries, const uint32_t* obj_offsets, RtNode* out, double* ms_device);
/* Refits a built tree to moved objects (BVHNode::new, src/bvh.rs:80-88, computes a node's box from its children's;
 * this recomputes those boxes over the topology the tokens hold): every inner token of tokens_dev (2 n - 1 tokens as
 * rt_bvh_build_device wrote them) gets the box of the leaves under it, taken from boxes_dev (n x 6 doubles, indexed by
 * the leaf tokens' object numbers), rounded outward to f32; links and leaf tokens stay. The result is bit for bit
 * what rt_bvh_build_device writes when the boxes still sort into the same order. flag_dev as there. Waits. */
int rt_bvh_refit_device(RtNode* tokens_dev, uint64_t n, const double* boxes_dev, uint32_t* flag_dev, void* stream);

#define RT_BVH_NOALLOC_MAX_OBJECTS 65536u
#define RT_BVH_NOALLOC_SCRATCH_BYTES (16u << 20)
/* rt_bvh_build_device for callers that cannot allocate: tight per-frame rebuild loops, serving loops, harnesses that
 * forbid hipMalloc. It has no counterpart in the reference. Same arguments and same tokens as rt_bvh_build_device,
 * but on the success path it allocates nothing, neither device nor host memory (a refused call builds its message
 * in a std::string), and does not wait: the work is enqueued on `stream` and the call returns. The scratch memory of
 * the build (sort keys, tree links, range unions) lives in a static device array of RT_BVH_NOALLOC_SCRATCH_BYTES that
 * each GPU the library runs on holds until the process ends; n is limited to RT_BVH_NOALLOC_MAX_OBJECTS (RT_E_INVALID
 * with a message beyond it). The first call on a device looks the array's address up (hipGetSymbolAddress, which may
 * load the library's code on that device): make one call per device before entering a region that must not allocate.
 * `stream` must belong to the current device (RT_E_INVALID otherwise): each GPU has its own array.
 * All calls on one GPU share that array: two calls whose work overlaps in time on one GPU (different streams,
 * different threads) race; order them on one stream or with events. */
int rt_bvh_build_device_noalloc(const double* boxes_dev, uint64_t n, RtNode* tokens_dev, uint32_t* flag_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
