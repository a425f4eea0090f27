// gfx950 kernels of the wavefront path tracer: generate / extend / shade / accumulate /
// resolve, plus the stand-alone closest-hit kernels behind rt_trace_*.
//
// One path sample per lane. The live paths of a bounce are a dense array of 96-byte ray records
// (origin, direction, time, slot, throughput); every bounce is two launches over it:
//   extend: closest hit of record i            -> hit[i]                        (src/color.rs:41)
//   shade : rebuild the HitRecord, add emitted light, scatter; survivors are written, compacted
//           with one ballot per wave into record ranges the wave reserves 256 at a time, into the
//           other record array                                                   (src/color.rs:50-62)
// so both kernels stream consecutive records instead of chasing an index queue, and a path costs one
// record read per kernel. Radiance gathered by a path lives in per-slot columns (slot = local_sample *
// n_pixels + pixel) that are touched only when light is added (miss or emitter) and summed per pixel by
// accumulate. No kernel needs the array length on the host: launches are fixed-size loops that read the
// length from device memory.
#pragma once
#include "shade.h"

namespace rtd {

struct alignas(32) RayRec {     // 96 bytes; extend reads the first 64, and the last 16 in scenes with media put to the ray first
    double o[3], d[3], tm;      // the ray handed to world.hit (src/ray.rs:3-7)
    uint32_t slot, first_ref;   // first_ref / first_t: the hit the walk starts from (see first_candidate), RT_MISS / +inf: none
    double T[3];                // throughput: product of attenuations so far
    double first_t;
};
struct alignas(16) HitSlot {    // extend -> shade, indexed like the records
    double t;
    uint32_t ref, aux;
};

struct PathState {
    RayRec* rec[2];             // records of the current / next bounce
    HitSlot* hit;
    double *lr, *lg, *lb;       // radiance gathered so far, per slot
};

// slot value of a padding record: the shade kernel reserves output records in chunks and pads what a wave did
// not use; extend and shade skip such records
constexpr uint32_t DEAD_SLOT = 0xFFFFFFFFu;
constexpr uint32_t DEAD_REF = 0xFFFFFFFEu;   // HitSlot::ref of a padding record (no entry has kind 14)

// 16-byte loads/stores of a record (global_load_dwordx4 / global_store_dwordx4)
RT_D void load_ray64(const RayRec* r, V3& o, V3& d, double& tm, uint32_t& slot, uint32_t& first_ref) {
    const double2* p = reinterpret_cast<const double2*>(r);
    const double2 a = p[0], b = p[1], c = p[2];
    const uint4 e = reinterpret_cast<const uint4*>(r)[3];
    o = v3(a.x, a.y, b.x);
    d = v3(b.y, c.x, c.y);
    tm = __hiloint2double((int)e.y, (int)e.x);
    slot = e.z;
    first_ref = e.w;
}
RT_D void load_ray64(const RayRec* r, V3& o, V3& d, double& tm, uint32_t& slot) {
    uint32_t first_ref;
    load_ray64(r, o, d, tm, slot, first_ref);
}
RT_D void load_throughput(const RayRec* r, V3& T) {
    const double2* p = reinterpret_cast<const double2*>(r);
    const double2 a = p[4], b = p[5];
    T = v3(a.x, a.y, b.x);
}
RT_D void store_ray(RayRec* r, V3 o, V3 d, double tm, uint32_t slot, V3 T, const Cand& first) {
    double2* p = reinterpret_cast<double2*>(r);
    p[0] = make_double2(o.x, o.y);
    p[1] = make_double2(o.z, d.x);
    p[2] = make_double2(d.y, d.z);
    uint4 e;
    e.x = (uint32_t)__double2loint(tm); e.y = (uint32_t)__double2hiint(tm); e.z = slot; e.w = first.ref;
    reinterpret_cast<uint4*>(r)[3] = e;
    p[4] = make_double2(T.x, T.y);
    p[5] = make_double2(T.z, first.t);
}

// The hit a world query starts from: in the reference's order a medium that every ray reaches is asked somewhere in
// the middle of the walk; here such media (taken out of the entry stream, SceneT::deferred) are asked first, as if they
// stood first in the world's list, and the walk then starts with their hit as the closest so far, which also spares
// it everything behind that hit. The candidate is computed by whoever makes the ray (k_generate, the shade launch of the
// previous bounce, k_step_pack: kernels with issue slots to spare) and travels in the ray's record. Returns the boundary
// queries made (work counters).
template <class SC>
RT_D FirstWork first_candidate(const SC& S, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, V3 o, V3 d, double tmin, double tmax,
                               Cand& first) {
    first.t = tmax; first.ref = RT_MISS; first.aux = 0u;
    FirstWork fw;
    fw.spheres = 0u;
    if (S.deferred.n == 0u) return fw;
    RayKey key;
    key.seed = seed; key.pixel = pixel; key.sample = sample; key.bounce = bounce;
    deferred_media(S, key, o, d, tmin, first, fw);
    return fw;
}

struct RenderConst {
    RtCamera cam;
    uint64_t seed;
    double bg[3];
    uint32_t width, height, npix;
    uint32_t sample0;      // first sample index of this batch
    uint32_t n_slots;      // npix * samples in this batch
    uint32_t max_depth;
};

constexpr int BLOCK = 256;
#ifndef RT_EXT_BLOCK_GLOBAL
#define RT_EXT_BLOCK_GLOBAL 768
#endif
#ifndef RT_EXTEND_WAVES_GLOBAL
#define RT_EXTEND_WAVES_GLOBAL 3
#endif
constexpr int EXT_BLOCK_GLOBAL = RT_EXT_BLOCK_GLOBAL;  // extend when the tables stay in global memory
constexpr int EXT_BLOCK = 768;    // extend: one 12-wave workgroup per CU sharing one LDS copy of the scene tables ...
constexpr int EXT_BLOCK_LEAN = 1024;  // ... or 16 waves when the walker needs no more than 128 registers (see Walker)
constexpr int TRACE_BLOCK = 512;  // rt_trace_*: also rebuilds the hit record, needs more registers per lane
constexpr int SHADE_BLOCK = 512;  // shade: 8 waves per workgroup share the LDS copy of primitive / material / texture tables

// Which scene tables the closest-hit kernels copy into LDS at launch (bytes, 0 = read from global memory).
// The small reference scenes fit entirely (final scene: 37 KB of entries + 27 KB of primitives), so every
// entry and primitive fetch of the walk is an LDS read instead of a trip through the vector cache.
struct LdsPlan {
    uint32_t nodes, spheres, msphere, rects, boxes, triangles, xforms, media, materials, textures, perlins;
};

extern __shared__ uint4 rt_lds[];

template <class T>
RT_D const RT_LDS T* stage_table(const T* src, uint32_t bytes, uint32_t& off) {
    const uint32_t chunks = (bytes + 15u) >> 4;  // tables are allocated with 16 bytes of slack
    uint4* dst = rt_lds + (off >> 4);
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    for (uint32_t k = threadIdx.x; k < chunks; k += blockDim.x) dst[k] = s4[k];
    off += chunks << 4;
    return (const RT_LDS T*)dst;
}

// Copies the planned tables into LDS and returns the scene as the kernels of the LDS variant see it. Tables with
// 0 bytes in the plan are not used by that kernel and keep a dangling (never dereferenced) LDS pointer.
RT_D SceneT<true> stage_scene(const DevScene& S, const LdsPlan& plan) {
    SceneT<true> L;
    uint32_t off = 0u;
    L.nodes = stage_table(S.nodes, plan.nodes, off);
    L.boxes = stage_table(S.boxes, plan.boxes, off);
    L.spheres = stage_table(S.spheres, plan.spheres, off);
    L.msphere = stage_table(S.msphere, plan.msphere, off);
    L.rects = stage_table(S.rects, plan.rects, off);
    L.triangles = stage_table(S.triangles, plan.triangles, off);
    L.xforms = stage_table(S.xforms, plan.xforms, off);
    L.media = stage_table(S.media, plan.media, off);
    L.materials = stage_table(S.materials, plan.materials, off);
    L.textures = stage_table(S.textures, plan.textures, off);
    L.perlins = stage_table(S.perlins, plan.perlins, off);
    L.images = S.images;
    L.texels = S.texels;
    L.n_nodes = S.n_nodes;
    L.carried_t0 = S.carried_t0; L.carried_t1 = S.carried_t1;
    L.deferred = S.deferred;
    __syncthreads();
    return L;
}

// minimum waves per SIMD the register allocator must leave room for (second __launch_bounds__ argument)
#ifndef RT_EXTEND_WAVES
#define RT_EXTEND_WAVES 3
#endif
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 2
#endif

// The same for a scene that does not fit as a whole: the tables the plan names are copied into LDS, the others
// stay where they are, and the kernel reaches both through generic pointers (flat loads).
template <class T>
RT_D const T* stage_table_generic(const T* src, uint32_t bytes, uint32_t& off) {
    if (bytes == 0u) return src;
    return (const T*)stage_table(src, bytes, off);
}
RT_D DevScene stage_scene_partial(const DevScene& S, const LdsPlan& plan) {
    DevScene L = S;
    uint32_t off = 0u;
    L.nodes = stage_table_generic(S.nodes, plan.nodes, off);
    L.boxes = stage_table_generic(S.boxes, plan.boxes, off);
    L.spheres = stage_table_generic(S.spheres, plan.spheres, off);
    L.msphere = stage_table_generic(S.msphere, plan.msphere, off);
    L.rects = stage_table_generic(S.rects, plan.rects, off);
    L.triangles = stage_table_generic(S.triangles, plan.triangles, off);
    L.xforms = stage_table_generic(S.xforms, plan.xforms, off);
    L.media = stage_table_generic(S.media, plan.media, off);
    L.materials = stage_table_generic(S.materials, plan.materials, off);
    L.textures = stage_table_generic(S.textures, plan.textures, off);
    L.perlins = stage_table_generic(S.perlins, plan.perlins, off);
    __syncthreads();
    return L;
}
RT_D bool plan_is_empty(const LdsPlan& p) {
    return (p.nodes | p.boxes | p.spheres | p.msphere | p.rects | p.triangles | p.xforms | p.media | p.materials | p.textures | p.perlins) == 0u;
}

// ---- persistent closest-hit loop ----------------------------------------------------------
// Rays of one launch form a pool; every wave pulls rays from it (large pools: through one shared cursor, 256
// rays per atomic; small pools: a fixed share per wave) and keeps its lanes busy: a lane whose query has finished
// stores its result and, as soon as enough lanes of the wave are idle, the idle lanes take new rays together
// while the other lanes keep their half-finished queries. Per step the wave runs box tests until every busy lane stands on a primitive /
// wrapper / medium entry, then handles those entries. This removes the two big sources of idle lanes of a
// one-ray-per-thread loop: waiting for the longest query of the wave, and box tests interleaved with
// primitive tests.
//
// IO supplies the pool: n(), fetch(index, ...) and store(index, ...).
#ifndef RT_REFILL_LANES
#define RT_REFILL_LANES 32
#endif
constexpr uint32_t REFILL_LANES = RT_REFILL_LANES;  // refill when at least this many lanes are idle
#ifndef RT_VOTE_DEN
#define RT_VOTE_DEN 4u
#endif
#ifndef RT_VOTE_DEN_GLOBAL
#define RT_VOTE_DEN_GLOBAL 2u  // the same when no table is staged in LDS
#endif
#ifndef RT_RAY_CHUNK
#define RT_RAY_CHUNK 256
#endif
constexpr uint32_t RAY_CHUNK = RT_RAY_CHUNK;  // rays a wave reserves per atomic while the pool is large
#ifndef RT_RAY_MIN_CHUNKS
#define RT_RAY_MIN_CHUNKS 4
#endif
constexpr uint32_t RAY_MIN_CHUNKS = RT_RAY_MIN_CHUNKS;  // ... "large" = at least this many chunks per wave
#ifndef RT_OUT_CHUNK
#define RT_OUT_CHUNK 256
#endif
constexpr uint32_t OUT_CHUNK = RT_OUT_CHUNK;  // output records a shading wave reserves per atomic while there are many rays
#ifndef RT_OUT_MIN_CHUNKS
#define RT_OUT_MIN_CHUNKS 2
#endif
constexpr uint32_t OUT_MIN_CHUNKS = RT_OUT_MIN_CHUNKS;  // ... "many" = at least this many chunks of input per wave

template <bool COUNT, bool BOXMEDIA, bool GENERAL, uint32_t VOTE_DEN, bool PER_AXIS, class SC, class IO>
RT_D void extend_pool(const SC& S, IO& io, uint32_t* __restrict__ cursor, unsigned long long* __restrict__ work_out) {
    Work work;
    if (COUNT) { work.entries = 0; for (int k = 0; k < 9; ++k) work.prim[k] = 0; work.first_media = 0; work.first.spheres = 0; }
    const uint32_t n = io.n();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total_waves = gridDim.x * (blockDim.x / 64u);
    // with few rays, spread them over the waves of the grid instead of filling a few waves completely
    uint32_t max_active = (n + total_waves - 1u) / total_waves;
    max_active = max_active < 8u ? 8u : (max_active > 64u ? 64u : max_active);
    const uint32_t refill_at = REFILL_LANES < max_active ? REFILL_LANES : max_active;
    // Where the wave's rays come from. Large pools: the shared cursor, RAY_CHUNK rays per atomic (dynamic balance).
    // Small pools: a fixed share per wave and no atomics at all; same-address atomics cost about 12 ns each however
    // few rays there are, which would dominate the late bounces.
    const bool dynamic = n / total_waves >= RAY_MIN_CHUNKS * RAY_CHUNK;
    uint32_t c_next = 0u, c_end = 0u;  // the wave's reserved range (wave-uniform)
    if (!dynamic) {
        const uint32_t wave_id = blockIdx.x * (blockDim.x / 64u) + (threadIdx.x >> 6);
        const uint32_t share = (n + total_waves - 1u) / total_waves;
        c_next = wave_id * share < n ? wave_id * share : n;
        c_end = c_next + share < n ? c_next + share : n;
    }
    Walker<COUNT, BOXMEDIA, GENERAL, PER_AXIS> w;
    w.done = true;
    typename IO::Ray ray;
    uint32_t index = 0u;
    bool busy = false;      // the lane holds an unfinished query
    bool pending = false;   // the lane holds a finished query whose result is not stored yet
    bool exhausted = false; // wave-uniform: the pool has no rays left
    for (;;) {
        const unsigned long long act = __ballot(busy);
        const uint32_t n_act = (uint32_t)__popcll(act);
        const uint32_t room = max_active > n_act ? max_active - n_act : 0u;
        // Results are stored and new rays fetched at the same moment, so that the wave meets global-memory
        // latency once per refill instead of at every step in which some lane finishes.
        if ((!exhausted && (room >= refill_at || n_act == 0u)) || (exhausted && n_act == 0u)) {
            if (pending) {
                io.store(S, index, ray, w.best);
                pending = false;
            }
            if (!exhausted && room > 0u) {
                const uint32_t my_rank = (uint32_t)__popcll(~act & ((1ull << lane) - 1ull));
                // rays come from the wave's reserved range [c_next, c_end); when it runs short and the pool is
                // dynamic, one atomic reserves the missing rays plus a whole chunk for later refills
                const uint32_t avail = c_end - c_next;
                uint32_t take = room, base = 0u;
                const bool more = dynamic && avail < room;
                if (more) {
                    const uint32_t want = room - avail > RAY_CHUNK ? room - avail : RAY_CHUNK;
                    if (lane == 0u) base = atomicAdd(cursor, want);
                    base = __shfl(base, 0, 64);
                    c_end = base + want;
                } else if (avail < room) {
                    take = avail;
                }
                const uint32_t mine = my_rank < avail ? c_next + my_rank : base + (my_rank - avail);
                if (!busy && my_rank < take && mine < n) {
                    index = mine;
                    Cand first;
                    if (io.fetch(index, ray, first)) {
                        // the media that are not in the stream are asked first: the stand-alone queries do it here, the
                        // render path gets the answer with the ray; instrumented launches ask again so that the work is counted
                        if ((IO::FIRST_CANDIDATE_HERE || COUNT) && S.deferred.n != 0u) {
                            const FirstWork fw = first_candidate(S, ray.key.seed, ray.key.pixel, ray.key.sample, ray.key.bounce, ray.o, ray.d,
                                                                 ray.tmin, ray.tmax, first);
#ifdef RT_DIAG_WAVE
                            (void)fw;  // the diagnostic build counts the executions of the walk's own phases only
#else
                            if (COUNT) {  // what the walk would have counted had these media stood in the stream
                                work.entries += S.deferred.n + fw.spheres;
                                work.prim[RT_N_MEDIUM] += S.deferred.n;
                                work.prim[RT_N_SPHERE] += fw.spheres;
                                work.first_media += S.deferred.n;
                                work.first.spheres += fw.spheres;
                            }
#endif
                        }
                        w.init(S, ray.o, ray.d, ray.tm, ray.tmin, first.t);
                        w.best.ref = first.ref;
                        busy = true;
                    } else {
                        io.skip(index);
                    }
                }
                c_next = more ? base + (room - avail) : c_next + take;
                exhausted = dynamic ? c_next >= n : c_next >= c_end;
            }
        }
        if (__ballot(busy) == 0ull) {
            if (exhausted) break;
            continue;
        }
        if (busy) {
            if (w.template box_run<VOTE_DEN>(S, ray.key, &work)) w.leaf(S, ray.rs, ray.key, &work);
            if (w.done) {
                busy = false;
                pending = true;
            }
        }
    }
    if (COUNT) {
        atomicAdd(work_out + 0, (unsigned long long)work.entries);
        for (int k = 0; k < 9; ++k) atomicAdd(work_out + 1 + k, (unsigned long long)work.prim[k]);
        // words 11 and 12 of the counter block (word 10 belongs to k_accumulate): the part of the above that was not walked
        if (work.first_media) {
            atomicAdd(work_out + 11, (unsigned long long)work.first_media);
            atomicAdd(work_out + 12, (unsigned long long)work.first.spheres);
        }
    }
}

// ---- stand-alone closest hit (rt_trace_closest / rt_trace_records) ---------------------
struct TraceIO {
    const RtRay* rays;
    uint64_t n_rays, seed;
    RtHit* hits;
    RtHitRecord* recs;
    struct Ray {
        V3 o, d;
        double tm, tmin, tmax;
        RayKey key;
        RayAoS rs;
    };
    static constexpr bool FIRST_CANDIDATE_HERE = true;
    RT_D uint32_t n() const { return (uint32_t)n_rays; }
    RT_D bool fetch(uint32_t i, Ray& r, Cand& first) const {
        const RtRay* q = rays + i;
        r.o = v3p(q->o); r.d = v3p(q->d); r.tm = q->tm; r.tmin = q->tmin; r.tmax = q->tmax;
        r.key.seed = seed; r.key.pixel = q->pixel; r.key.sample = q->sample; r.key.bounce = q->bounce;
        r.rs.r = q;
        first.t = q->tmax; first.ref = RT_MISS; first.aux = 0u;
        return true;
    }
    RT_D void skip(uint32_t) const {}
    template <class SC>
    RT_D void store(const SC& S, uint32_t i, const Ray& r, const Cand& best) const {
        V3 o, d;
        r.rs.load(o, d);
        if (recs) {
            RtHitRecord out;
            out.prim_id = RT_MISS; out.sub = 0u; out.mat = RT_NONE; out.front_face = 0u;
            out.t = 0.0; out.u = 0.0; out.v = 0.0;
            for (int k = 0; k < 3; ++k) { out.p[k] = 0.0; out.normal[k] = 0.0; }
            if (best.ref != RT_MISS) {
                HitRec rec;
                build_hit_record(S, o, d, r.tm, best, rec);
                hit_ids(S, best, out.prim_id, out.sub);
                out.p[0] = rec.p.x; out.p[1] = rec.p.y; out.p[2] = rec.p.z;
                out.normal[0] = rec.normal.x; out.normal[1] = rec.normal.y; out.normal[2] = rec.normal.z;
                out.t = rec.t; out.u = rec.u; out.v = rec.v; out.mat = rec.mat; out.front_face = rec.front_face ? 1u : 0u;
            }
            recs[i] = out;
        } else {
            RtHit h;
            h.prim_id = RT_MISS; h.sub = 0u; h.t = 0.0; h.u = 0.0; h.v = 0.0;
            if (best.ref != RT_MISS) {
                HitRec rec;
                build_hit_record(S, o, d, r.tm, best, rec);
                hit_ids(S, best, h.prim_id, h.sub);
                h.t = rec.t; h.u = rec.u; h.v = rec.v;
            }
            hits[i] = h;
        }
    }
};

// LDS = true: every table the kernel reads (all but images / texels) is staged in LDS first. PER_AXIS: the
// reference's own box rule (see Walker). Writes RtHitRecords when `recs` is not null, else RtHits.
template <bool COUNT, bool LDS, bool PER_AXIS>
__global__ void __launch_bounds__(TRACE_BLOCK, 2) k_trace(DevScene S0, LdsPlan plan, const RtRay* __restrict__ rays, uint64_t n, uint64_t seed,
                                                        RtHit* __restrict__ hits, RtHitRecord* __restrict__ recs,
                                                        uint32_t* __restrict__ cursor, unsigned long long* __restrict__ work_out) {
    TraceIO io;
    io.rays = rays; io.n_rays = n; io.seed = seed; io.hits = hits; io.recs = recs;
    if constexpr (LDS) {
        const SceneT<true> S = stage_scene(S0, plan);
        extend_pool<COUNT, false, true, 0u, PER_AXIS>(S, io, cursor, work_out);
    } else if (plan_is_empty(plan)) {
        extend_pool<COUNT, false, true, 0u, PER_AXIS>(S0, io, cursor, work_out);
    } else {
        const DevScene S = stage_scene_partial(S0, plan);
        extend_pool<COUNT, false, true, 0u, PER_AXIS>(S, io, cursor, work_out);
    }
}

// ---- generate: camera rays of one batch (src/main.rs:206-209, src/camera.rs:61-71) ---------
__global__ void __launch_bounds__(BLOCK) k_generate(DevScene S, RenderConst C, PathState P) {
    // A lane's record is 96 bytes, so lanes storing their own records would touch 48 cache lines per store
    // instruction; the wave lays its 64 records out in LDS instead and writes the 6 KiB out in address order.
    __shared__ uint4 s_rec[BLOCK / 64][64 * 6];
    const uint32_t stride = gridDim.x * BLOCK;
    const uint32_t lane = threadIdx.x & 63u;
    uint4* mine = s_rec[threadIdx.x >> 6];
    const uint32_t n_round = (C.n_slots + 63u) & ~63u;  // whole waves, so that every lane takes part in the copy
    for (uint32_t slot = blockIdx.x * BLOCK + threadIdx.x; slot < n_round; slot += stride) {
        if (slot < C.n_slots) {
            const uint32_t pix = slot % C.npix;
            const uint32_t sample = C.sample0 + slot / C.npix;
            const uint32_t x = pix % C.width, y = pix / C.width;
            V3 o, d;
            double tm;
            camera_ray(C.cam, C.seed, C.width, C.height, x, y, pix, sample, o, d, tm);
            Cand first;
            (void)first_candidate(S, C.seed, pix, sample, 0u, o, d, 0.0001, D_INF, first);
            store_ray(reinterpret_cast<RayRec*>(mine + lane * 6u), o, d, tm, slot, v3(1.0, 1.0, 1.0), first);
            P.lr[slot] = 0.0; P.lg[slot] = 0.0; P.lb[slot] = 0.0;
        }
        // lanes read what other lanes of the wave wrote: the wave runs in lockstep and LDS keeps its order; the fence
        // and the scheduling barrier keep the compiler from moving accesses across this point
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // the records of this wave's 64 slots are contiguous: slot - lane is the first
        const uint32_t first = slot - lane;
        const uint32_t n_chunks = (C.n_slots - first < 64u ? C.n_slots - first : 64u) * 6u;
        uint4* out = reinterpret_cast<uint4*>(P.rec[0] + first);
        for (uint32_t q = lane; q < n_chunks; q += 64u) out[q] = mine[q];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- extend ------------------------------------------------------------------------------
struct RayRecRef {  // the query's world ray, kept in registers: needed again whenever a wrapper is left
    V3 wo, wd;
    RT_D void load(V3& o, V3& d) const { o = wo; d = wd; }
};
struct ExtendIO {
    const RayRec* recs;
    HitSlot* hits;
    uint64_t seed;
    uint32_t npix, sample0, count, bounce;
    struct Ray {
        V3 o, d;
        double tm;
        static constexpr double tmin = 0.0001;  // src/color.rs:41
        static constexpr double tmax = D_INF;
        RayKey key;
        RayRecRef rs;
    };
    static constexpr bool FIRST_CANDIDATE_HERE = false;  // it comes with the record
    bool has_first;  // the scene has media that are asked first (wave-uniform)
    RT_D uint32_t n() const { return count; }
    RT_D bool fetch(uint32_t i, Ray& r, Cand& first) const {  // false: a padding record, no ray
        uint32_t slot;
        first.t = D_INF; first.ref = RT_MISS; first.aux = 0u;
        load_ray64(recs + i, r.o, r.d, r.tm, slot, first.ref);
        if (has_first) first.t = reinterpret_cast<const double2*>(recs + i)[5].y;
        r.rs.wo = r.o; r.rs.wd = r.d;
        r.key.seed = seed; r.key.pixel = slot % npix; r.key.sample = sample0 + slot / npix; r.key.bounce = bounce;
        return slot != DEAD_SLOT;
    }
    // a padding record: marked in the hit array, so that the shade kernel's sorting pass can tell without touching
    // the record
    RT_D void skip(uint32_t i) const { hits[i].ref = DEAD_REF; }
    template <class SC>
    RT_D void store(const SC&, uint32_t i, const Ray&, const Cand& best) const {
        uint4 h;
        h.x = (uint32_t)__double2loint(best.t); h.y = (uint32_t)__double2hiint(best.t); h.z = best.ref; h.w = best.aux;
        *reinterpret_cast<uint4*>(hits + i) = h;
    }
};

// counts[0], counts[1]: lengths of the two record arrays; counts[2]: the ray cursor of this launch (zero on
// entry, reset by the shade launch that follows).
// PROFILE bits: 1 = the scene has boxes alone inside wrappers, 2 = it has media that need the general boundary walk
// (set together with 1), 4 = lanes leave the box phase by vote (entry streams of 128 entries and more), 8 = the
// reference's per-axis box rule (scenes whose boxes may not bound their contents; built without bits 1 and 2); see Walker.
// Without bit 2 the kernel runs 16 waves per CU, with it 12.
constexpr int extend_block(bool lds, int profile) { return (profile & 2) ? (lds ? EXT_BLOCK : EXT_BLOCK_GLOBAL) : EXT_BLOCK_LEAN; }
constexpr int extend_waves(bool lds, int profile) { return (profile & 2) ? (lds ? RT_EXTEND_WAVES : RT_EXTEND_WAVES_GLOBAL) : 4; }

template <bool COUNT, bool LDS, int PROFILE>
__global__ void __launch_bounds__(extend_block(LDS, PROFILE), extend_waves(LDS, PROFILE)) k_extend(DevScene S0, LdsPlan plan, RenderConst C, PathState P,
                                                                       uint32_t* __restrict__ counts, uint32_t cur, uint32_t bounce,
                                                                       unsigned long long* __restrict__ work_out) {
    ExtendIO io;
    io.recs = P.rec[cur]; io.hits = P.hit; io.seed = C.seed; io.npix = C.npix; io.sample0 = C.sample0; io.bounce = bounce;
    io.count = bounce == 0u ? C.n_slots : counts[cur];
    io.has_first = S0.deferred.n != 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[cur ^ 1u] = 0u;  // the array the following shade launch appends to
    if constexpr (LDS) {
        const SceneT<true> S = stage_scene(S0, plan);
        extend_pool<COUNT, (PROFILE & 1) != 0, (PROFILE & 2) != 0, (PROFILE & 4) ? RT_VOTE_DEN : 0u, (PROFILE & 8) != 0>(S, io, counts + 2, work_out);
    } else if (plan_is_empty(plan)) {
        // every entry fetch is a trip to L2 / HBM here: lanes leave the box phase earlier (10 M-triangle mesh: +2.3 % at 1/2, -9 % at 1/8)
        extend_pool<COUNT, (PROFILE & 1) != 0, (PROFILE & 2) != 0, (PROFILE & 4) ? RT_VOTE_DEN_GLOBAL : 0u, (PROFILE & 8) != 0>(S0, io, counts + 2, work_out);
    } else {
        const DevScene S = stage_scene_partial(S0, plan);
        extend_pool<COUNT, (PROFILE & 1) != 0, (PROFILE & 2) != 0, (PROFILE & 4) ? RT_VOTE_DEN : 0u, (PROFILE & 8) != 0>(S, io, counts + 2, work_out);
    }
}

// ---- shade ---------------------------------------------------------------------------------
// The body of ray_color after world.hit (src/color.rs:42-62), unrolled from recursion into
// L += T * emitted; T *= attenuation. Survivors go to the other record array.
// L += add. The columns start at +0.0 and x + (+-0.0) == x bit for bit for every x but -0.0, which a sum of
// products of non-negative colours never is: an all-zero term (black background, black emitter) is skipped.
RT_D void add_radiance(const PathState& P, uint32_t slot, V3 add) {
    if (add.x == 0.0 && add.y == 0.0 && add.z == 0.0) return;
    P.lr[slot] += add.x; P.lg[slot] += add.y; P.lb[slot] += add.z;
}

// What shading a hit involves, as a sort key: 0 = miss, else 1 + material kind * 32 + texture kind * 8 + primitive
// kind. The last bin is reserved for "no ray".
constexpr uint32_t SORT_BINS = 192u;                      // 3 per lane of the wave that scans them
// Records per thread and sorting window of the sorting shade kernel: two sizes are built. Measured on the final scene
// (seven kinds of shading), shade ms per 1024-spp step by records per thread: 2 -> 271, 4 -> 201, 8 -> 175, 12 -> 168,
// 16 -> 168 before the kernel also asked the media of the next ray; with that work in: 8 -> 210, 12 -> 200, 18 -> 195,
// 20 -> 192.5, 24 -> 192.8, 28 -> 191.5. Cornell smoke (three kinds): 12 -> 138, 18 -> 140, 20 -> 141, 24 -> 148. The host
// picks the larger window for scenes with many kinds.
#ifndef RT_SORT_ROUNDS
#define RT_SORT_ROUNDS 12
#endif
#ifndef RT_SORT_ROUNDS_WIDE
#define RT_SORT_ROUNDS_WIDE 20
#endif
constexpr uint32_t SORT_ROUNDS = RT_SORT_ROUNDS, SORT_ROUNDS_WIDE = RT_SORT_ROUNDS_WIDE;
constexpr uint32_t sort_window(uint32_t rounds) { return SHADE_BLOCK * rounds; }
static_assert(sort_window(SORT_ROUNDS_WIDE) <= 65536u, "positions within a window are kept in 16 bits");
template <class SC>
RT_D uint32_t shade_key(const SC& S, uint32_t ref) {
    if (ref == RT_MISS) return 0u;
    Cand c;
    c.t = 0.0; c.ref = ref; c.aux = 0u;
    const auto M = S.materials + hit_material(S, c);
    const uint32_t mkind = M->kind;
    const uint32_t tkind = material_has_texture(mkind) ? S.textures[M->tex].kind : 0u;
    return 1u + mkind * 32u + tkind * 8u + ((ref & 15u) - 1u);
}

// One path at its hit (or miss): adds what it gathers to its radiance columns; returns true with the next ray in
// (no, nd) and the updated throughput T when the path goes on.
template <class SC>
RT_D bool shade_one(const SC& S, const RenderConst& C, const PathState& P, uint32_t bounce, V3 o, V3 d, double tm, uint32_t slot,
                    uint4 h, V3& T, V3& no, V3& nd, Cand& first) {
    Cand best;
    best.t = __hiloint2double((int)h.y, (int)h.x); best.ref = h.z; best.aux = h.w;
    if (best.ref == RT_MISS) {
        add_radiance(P, slot, elemul(T, v3p(C.bg)));  // src/color.rs:41-43
        return false;
    }
    const auto M = S.materials + hit_material(S, best);
    const uint32_t mkind = M->kind;
    const bool textured = material_has_texture(mkind);
    const bool want_uv = textured && S.textures[M->tex].kind == RT_T_IMAGE;
    HitRec rec;
    build_hit_record(S, o, d, tm, best, rec, want_uv);
    V3 texv = v3(0.0, 0.0, 0.0);
    if (textured) texv = texture_value(S, M->tex, rec.u, rec.v, rec.p);
    if (mkind == RT_M_DIFFUSE_LIGHT) add_radiance(P, slot, elemul(T, texv));  // emitted, :50 (zero for every other material)
    rt::Rng rng;
    rng.set_path(C.seed, slot % C.npix, C.sample0 + slot / C.npix, bounce, rt::RNG_SCATTER);
    V3 attenuation = v3(0.0, 0.0, 0.0), dout = v3(1.0, 1.0, 1.0);
    if (!material_scatter(M, d, rec, texv, rng, attenuation, dout)) return false;  // :52-57
    T = elemul(T, attenuation);
    no = rec.p;
    nd = dout;
    if (bounce + 1u >= C.max_depth) return false;  // ray_color(depth - 1) returns black at depth 0 (:37-39)
    // what the next world.hit starts from (media that are asked first)
    (void)first_candidate(S, C.seed, slot % C.npix, C.sample0 + slot / C.npix, bounce + 1u, no, nd, 0.0001, D_INF, first);
    return true;
}

// Where a shading wave writes its survivors: consecutive records of a range it reserves `chunk` records at a time
// (one atomic per reservation; chunk = 0: exactly what each round needs).
struct SurvivorOut {
    RayRec* next;
    uint32_t* next_count;
    uint4* stage;  // 64 records of LDS owned by this wave
    uint32_t chunk, o_next, o_end, survivors;
    // A lane storing its own 96-byte record makes the wave touch 48 cache lines per store instruction, each only in
    // part; the survivors are laid out in LDS in output order instead and copied out in address order.
    RT_D void emit(bool alive, uint32_t lane, V3 no, V3 nd, double tm, uint32_t slot, V3 T, const Cand& first) {
        const unsigned long long mask = __ballot(alive);
        if (!mask) return;
        const uint32_t cnt = (uint32_t)__popcll(mask);
        const uint32_t avail = o_end - o_next;
        uint32_t base = 0u;
        if (avail < cnt) {
            const uint32_t want = cnt - avail > chunk ? cnt - avail : chunk;
            if (lane == 0u) base = atomicAdd(next_count, want);
            base = __shfl(base, 0, 64);
            o_end = base + want;
        }
        const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (alive) store_ray(reinterpret_cast<RayRec*>(stage + rank * 6u), no, nd, tm, slot, T, first);
        // lanes read what other lanes of the wave wrote: the wave runs in lockstep and LDS keeps its order; the fence
        // and the scheduling barrier keep the compiler from moving accesses across this point
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (uint32_t q = lane; q < cnt * 6u; q += 64u) {
            const uint32_t r = q / 6u, piece = q - r * 6u;
            const uint32_t dest = r < avail ? o_next + r : base + (r - avail);
            reinterpret_cast<uint4*>(next + dest)[piece] = stage[q];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        o_next = avail < cnt ? base + (cnt - avail) : o_next + cnt;
        survivors += cnt;
    }
    // pads the unused part of the last reservation and reports the survivors
    RT_D void finish(uint32_t lane, unsigned long long* ray_total) {
        for (uint32_t k = o_next + lane; k < o_end; k += 64u) next[k].slot = DEAD_SLOT;
        if (lane == 0u && survivors) atomicAdd(ray_total, (unsigned long long)survivors);
    }
};

template <uint32_t ROUNDS, class SC>  // ROUNDS: records per thread of a sorting window, 0 = no sorting
RT_D void shade_all(const SC& S, const RenderConst& C, const PathState& P, uint32_t* __restrict__ counts, uint32_t cur, uint32_t bounce,
                    unsigned long long* __restrict__ ray_total) {
    const uint32_t n = bounce == 0u ? C.n_slots : counts[cur];
    const RayRec* recs = P.rec[cur];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (bounce == 0u) atomicAdd(ray_total, (unsigned long long)n);  // later bounces: counted as survivors below
        counts[2] = 0u;  // ray cursor of the next extend launch
    }
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total_waves = gridDim.x * (SHADE_BLOCK / 64u);
    __shared__ uint4 s_stage[SHADE_BLOCK / 64][64 * 6];
    SurvivorOut out;
    out.stage = s_stage[threadIdx.x >> 6];
    out.next = P.rec[cur ^ 1u];
    out.next_count = counts + (cur ^ 1u);
    out.chunk = n / total_waves >= OUT_MIN_CHUNKS * OUT_CHUNK ? OUT_CHUNK : 0u;
    out.o_next = 0u; out.o_end = 0u; out.survivors = 0u;
  constexpr uint32_t SORT_WINDOW = sort_window(ROUNDS);
  if constexpr (ROUNDS != 0u) {
    // Scenes with many kinds of material: every workgroup walks windows of SORT_WINDOW records. The rays of a window are sorted by what shading them
    // involves (counting sort in LDS) and handed out 64 at a time to whichever wave is free, so that a wave mostly
    // runs one branch of shade_one instead of all of them.
    __shared__ uint32_t s_hist[2][SORT_BINS];
    __shared__ uint16_t s_perm[2][SORT_WINDOW];
    __shared__ uint32_t s_next_chunk[2];
    if (threadIdx.x < SORT_BINS) { s_hist[0][threadIdx.x] = 0u; s_hist[1][threadIdx.x] = 0u; }
    if (threadIdx.x < 2u) s_next_chunk[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t round = 0u;
    for (uint32_t base = blockIdx.x * SORT_WINDOW; base < n; base += gridDim.x * SORT_WINDOW, ++round) {
        uint32_t* hist = s_hist[round & 1u];
        uint16_t* perm = s_perm[round & 1u];
        uint32_t keys[ROUNDS], ranks[ROUNDS];
#pragma unroll
        for (uint32_t k = 0; k < ROUNDS; ++k) {
            const uint32_t i = base + k * SHADE_BLOCK + threadIdx.x;
            uint32_t key = SORT_BINS - 1u;  // no ray: beyond n, or a padding record
            if (i < n) {
                const uint32_t ref = P.hit[i].ref;
                if (ref != DEAD_REF) key = shade_key(S, ref);
            }
            keys[k] = key;
            ranks[k] = atomicAdd(&hist[key], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64u) {  // exclusive prefix sum over the bins, three bins per lane of the first wave
            const uint32_t a = hist[3u * threadIdx.x], b = hist[3u * threadIdx.x + 1u], c = hist[3u * threadIdx.x + 2u];
            uint32_t incl = a + b + c;
            for (uint32_t dlt = 1u; dlt < 64u; dlt <<= 1) {
                const uint32_t t = __shfl_up(incl, dlt, 64);
                if (threadIdx.x >= dlt) incl += t;
            }
            const uint32_t excl = incl - (a + b + c);
            hist[3u * threadIdx.x] = excl; hist[3u * threadIdx.x + 1u] = excl + a; hist[3u * threadIdx.x + 2u] = excl + a + b;
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < ROUNDS; ++k) perm[hist[keys[k]] + ranks[k]] = (uint16_t)(k * SHADE_BLOCK + threadIdx.x);
        // the other set of bins and the other chunk counter are free again: every wave is past their last use
        if (threadIdx.x < SORT_BINS) s_hist[(round & 1u) ^ 1u][threadIdx.x] = 0u;
        if (threadIdx.x == 0u) s_next_chunk[(round & 1u) ^ 1u] = 0u;
        __syncthreads();
        const uint32_t n_valid = hist[SORT_BINS - 1u];  // lanes at or past the start of the last bin have no ray
        // chunks of 64 sorted rays, taken from a shared counter; the records of the next chunk are requested before
        // the current one is shaded
        auto grab = [&]() {
            uint32_t c = 0u;
            if (lane == 0u) c = atomicAdd(&s_next_chunk[round & 1u], 1u);
            return (uint32_t)__shfl(c, 0, 64);
        };
        V3 o_n = v3(0.0, 0.0, 0.0), d_n = o_n, T_n = o_n;
        double tm_n = 0.0;
        uint32_t slot_n = DEAD_SLOT;
        uint4 h_n = make_uint4(0u, 0u, RT_MISS, 0u);
        auto request = [&](uint32_t c) {
            const uint32_t j = c * 64u + lane;
            slot_n = DEAD_SLOT;
            if (j < n_valid) {
                const uint32_t i = base + perm[j];
                load_ray64(recs + i, o_n, d_n, tm_n, slot_n);
                load_throughput(recs + i, T_n);
                h_n = *reinterpret_cast<const uint4*>(P.hit + i);
            }
        };
        uint32_t c = grab();
        request(c);
        while (c * 64u < n_valid) {
            V3 no = v3(0.0, 0.0, 0.0), nd = no;
            const V3 o = o_n, d = d_n;
            V3 T = T_n;
            const double tm = tm_n;
            const uint32_t slot = slot_n;
            const uint4 h = h_n;
            c = grab();
            request(c);
            Cand first;
            first.t = D_INF; first.ref = RT_MISS; first.aux = 0u;
            const bool alive = slot != DEAD_SLOT && shade_one(S, C, P, bounce, o, d, tm, slot, h, T, no, nd, first);
            out.emit(alive, lane, no, nd, tm, slot, T, first);
        }
    }
  } else {
    const uint32_t stride = gridDim.x * SHADE_BLOCK;
    // every lane of a wave runs the same number of iterations so that the ballot of emit() sees whole waves
    const uint32_t n_round = (n + 63u) & ~63u;
    // The record and hit of the next round are requested before the current round is shaded, so that their trip to
    // HBM overlaps the arithmetic (two waves per SIMD cannot hide it otherwise).
    V3 o_n = v3(0.0, 0.0, 0.0), d_n = o_n, T_n = o_n;
    double tm_n = 0.0;
    uint32_t slot_n = DEAD_SLOT;
    uint4 h_n = make_uint4(0u, 0u, RT_MISS, 0u);
    {
        const uint32_t i0 = blockIdx.x * SHADE_BLOCK + threadIdx.x;
        if (i0 < n) {
            load_ray64(recs + i0, o_n, d_n, tm_n, slot_n);
            load_throughput(recs + i0, T_n);
            h_n = *reinterpret_cast<const uint4*>(P.hit + i0);
        }
    }
    for (uint32_t i = blockIdx.x * SHADE_BLOCK + threadIdx.x; i < n_round; i += stride) {
        V3 no = v3(0.0, 0.0, 0.0), nd = no;
        const V3 o = o_n, d = d_n;
        V3 T = T_n;
        const double tm = tm_n;
        const uint32_t slot = slot_n;
        const uint4 h = h_n;
        slot_n = DEAD_SLOT;
        if (i + stride < n) {
            load_ray64(recs + i + stride, o_n, d_n, tm_n, slot_n);
            load_throughput(recs + i + stride, T_n);
            h_n = *reinterpret_cast<const uint4*>(P.hit + i + stride);
        }
        Cand first;
        first.t = D_INF; first.ref = RT_MISS; first.aux = 0u;
        const bool alive = slot != DEAD_SLOT && shade_one(S, C, P, bounce, o, d, tm, slot, h, T, no, nd, first);
        out.emit(alive, lane, no, nd, tm, slot, T, first);
    }
  }
    out.finish(lane, ray_total);
}

// LDS bytes the sorting variant of the shade kernel declares for itself (bins, permutations, chunk counters)
constexpr uint32_t shade_sort_lds(uint32_t rounds) { return rounds ? 2u * SORT_BINS * 4u + 2u * sort_window(rounds) * 2u + 16u : 0u; }
// LDS bytes both variants declare for laying out the survivors' records
constexpr uint32_t SHADE_STAGE_LDS = (SHADE_BLOCK / 64u) * 64u * 6u * 16u;

template <bool LDS, uint32_t ROUNDS>
__global__ void __launch_bounds__(SHADE_BLOCK, RT_SHADE_WAVES) k_shade(DevScene S0, LdsPlan plan, RenderConst C, PathState P,
                                                                       uint32_t* __restrict__ counts, uint32_t cur, uint32_t bounce,
                                                                       unsigned long long* __restrict__ ray_total) {
    if constexpr (LDS) {
        const SceneT<true> S = stage_scene(S0, plan);
        shade_all<ROUNDS>(S, C, P, counts, cur, bounce, ray_total);
    } else if (plan_is_empty(plan)) {
        shade_all<ROUNDS>(S0, C, P, counts, cur, bounce, ray_total);
    } else {
        const DevScene S = stage_scene_partial(S0, plan);
        shade_all<ROUNDS>(S, C, P, counts, cur, bounce, ray_total);
    }
}

// ---- one bounce on caller-supplied rays (rt_path_step) ------------------------------------------------
// pack: ray idx[k] of the caller's array becomes record k of the current record array, with throughput 1 and the
// slot its (pixel, sample) has in a render of these parameters; its radiance columns start at zero.
__global__ void __launch_bounds__(BLOCK) k_step_pack(DevScene S, uint64_t seed, const RtRay* __restrict__ rays, const uint32_t* __restrict__ idx, uint32_t n,
                                                     uint32_t npix, uint32_t sample_begin, PathState P, uint32_t cur,
                                                     uint32_t* __restrict__ slot_to_ray) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n) return;
    const uint32_t i = idx[k];
    const RtRay* q = rays + i;
    const uint32_t slot = (q->sample - sample_begin) * npix + q->pixel;
    Cand first;
    (void)first_candidate(S, seed, q->pixel, q->sample, q->bounce, v3p(q->o), v3p(q->d), 0.0001, D_INF, first);
    store_ray(P.rec[cur] + k, v3p(q->o), v3p(q->d), q->tm, slot, v3(1.0, 1.0, 1.0), first);
    P.lr[slot] = 0.0; P.lg[slot] = 0.0; P.lb[slot] = 0.0;
    slot_to_ray[slot] = i;
}
// unpack, first half: the closest hit the extend launch stored for record k and what the shade launch added to the
// ray's radiance columns
__global__ void __launch_bounds__(BLOCK) k_step_unpack_hits(DevScene S, const uint32_t* __restrict__ idx, uint32_t n, PathState P,
                                                            uint32_t cur, RtPathStep* __restrict__ out) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n) return;
    const uint4 h = *reinterpret_cast<const uint4*>(P.hit + k);
    Cand best;
    best.t = __hiloint2double((int)h.y, (int)h.x); best.ref = h.z; best.aux = h.w;
    const uint32_t slot = P.rec[cur][k].slot;
    RtPathStep o;
    o.emitted[0] = P.lr[slot]; o.emitted[1] = P.lg[slot]; o.emitted[2] = P.lb[slot];
    for (int c = 0; c < 3; ++c) { o.attenuation[c] = 0.0; o.o[c] = 0.0; o.d[c] = 0.0; }
    o.t = 0.0; o.prim_id = RT_MISS; o.sub = 0u; o.scattered = 0u; o._pad = 0u;
    if (best.ref != RT_MISS) {
        o.t = best.t;
        hit_ids(S, best, o.prim_id, o.sub);
    }
    out[idx[k]] = o;
}
// unpack, second half: the records the shade launch appended to the other array are the scattered rays; with an
// incoming throughput of 1 the stored throughput is the attenuation itself
__global__ void __launch_bounds__(BLOCK) k_step_unpack_survivors(PathState P, uint32_t next, const uint32_t* __restrict__ counts,
                                                                 const uint32_t* __restrict__ slot_to_ray, RtPathStep* __restrict__ out) {
    const uint32_t n = counts[next];
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t k = blockIdx.x * BLOCK + threadIdx.x; k < n; k += stride) {
        V3 o, d, T;
        double tm;
        uint32_t slot;
        load_ray64(P.rec[next] + k, o, d, tm, slot);
        if (slot == DEAD_SLOT) continue;
        load_throughput(P.rec[next] + k, T);
        RtPathStep* r = out + slot_to_ray[slot];
        r->o[0] = o.x; r->o[1] = o.y; r->o[2] = o.z;
        r->d[0] = d.x; r->d[1] = d.y; r->d[2] = d.z;
        r->attenuation[0] = T.x; r->attenuation[1] = T.y; r->attenuation[2] = T.z;
        r->scattered = 1u;
    }
}

// ---- accumulate: sum the batch's samples of each pixel in sample order ---------------------------
// lit_out, when not null (instrumented renders only): receives the number of slots whose radiance columns hold light,
// i.e. that the shade launches read, added to and wrote back (a path gathers light at most once: a miss and an emitter
// both end it, src/color.rs:41-43, src/material.rs:167-178)
__global__ void __launch_bounds__(BLOCK) k_accumulate(RenderConst C, PathState P, float* __restrict__ accum, unsigned long long* __restrict__ lit_out) {
    const uint32_t stride = gridDim.x * BLOCK;
    const uint32_t n_samples = C.n_slots / C.npix;
    uint32_t lit = 0u;
    for (uint32_t pix = blockIdx.x * BLOCK + threadIdx.x; pix < C.npix; pix += stride) {
        double r = 0.0, g = 0.0, b = 0.0;
        for (uint32_t s = 0; s < n_samples; ++s) {
            const uint32_t slot = s * C.npix + pix;
            const double sr = P.lr[slot], sg = P.lg[slot], sb = P.lb[slot];
            if (lit_out && (sr != 0.0 || sg != 0.0 || sb != 0.0)) ++lit;
            r += sr; g += sg; b += sb;
        }
        float* a = accum + (size_t)pix * 3u;
        a[0] = (float)((double)a[0] + r);
        a[1] = (float)((double)a[1] + g);
        a[2] = (float)((double)a[2] + b);
    }
    if (lit_out && lit) atomicAdd(lit_out, (unsigned long long)lit);
}

// ---- resolve (src/main.rs:212-225) -----------------------------------------------------------------
RT_D uint8_t f64_to_u8_sat(double f) {  // Rust `as u8`
    if (!(f > 0.0)) return 0;
    if (f >= 255.0) return 255;
    return (uint8_t)f;
}
__global__ void __launch_bounds__(BLOCK) k_resolve(const float* __restrict__ accum, uint64_t n_values, double scale,
                                                   uint8_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n_values; i += stride) {
        double c = sqrt(scale * (double)accum[i]);
        c = c < 0.0 ? 0.0 : (c > 0.999 ? 0.999 : c);  // clamp(c, 0.0, 0.999), src/rtweekend.rs:22-30
        out[i] = f64_to_u8_sat(c * 255.0);
    }
}

}  // namespace rtd
