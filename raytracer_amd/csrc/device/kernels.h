// gfx950 kernels of the wavefront path tracer: generate / extend / shade / accumulate /
// resolve, plus the stand-alone closest-hit kernels behind rt_trace_*.
//
// One path sample per lane. Path state lives in HBM as structure-of-arrays f64
// columns indexed by path slot (slot = local_sample * n_pixels + pixel), so lanes
// of a wave touch consecutive addresses while the queue is still dense. Every
// bounce is two launches over the queue of live slots:
//   extend: closest hit of the slot's ray  -> hit columns            (src/color.rs:41)
//   shade : rebuild the HitRecord, add emitted light, scatter, write the next
//           ray and throughput, append survivors to the next queue    (src/color.rs:50-62)
// Survivors are compacted with one ballot and one atomic per wave. No kernel needs
// the queue length on the host: launches are fixed-size grid-stride loops that read
// the length from device memory.
#pragma once
#include "shade.h"

namespace rtd {

struct PathState {
    double *ox, *oy, *oz, *dx, *dy, *dz, *tm;  // current ray (src/ray.rs:3-7)
    double *tr, *tg, *tb;                      // throughput: product of attenuations so far
    double *lr, *lg, *lb;                      // radiance gathered so far
    double* hit_t;                             // extend -> shade
    uint32_t *hit_ref, *hit_aux;
};

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

// minimum waves per SIMD the register allocator must leave room for (second __launch_bounds__ argument)
#ifndef RT_EXTEND_WAVES
#define RT_EXTEND_WAVES 3
#endif
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 2
#endif

// ---- persistent closest-hit loop ----------------------------------------------------------
// Rays of one launch form a pool; every wave pulls rays from it through one shared cursor and keeps its
// lanes busy: a lane whose query has finished stores its result and, as soon as enough lanes of the wave are
// idle, the idle lanes pull new rays together (one atomic per refill) while the other lanes keep their
// half-finished queries. Per step the wave runs box tests until every busy lane stands on a primitive /
// wrapper / medium entry, then handles those entries. This removes the two big sources of idle lanes of a
// one-ray-per-thread loop: waiting for the longest query of the wave, and box tests interleaved with
// primitive tests.
//
// IO supplies the pool: n(), fetch(index, ...) and store(index, ...).
constexpr uint32_t REFILL_LANES = 16;  // refill when at least this many lanes are idle

template <bool COUNT, class IO>
RT_D void extend_pool(const DevScene& S, IO& io, uint32_t* __restrict__ cursor, unsigned long long* __restrict__ work_out) {
    Work work;
    if (COUNT) { work.entries = 0; for (int k = 0; k < 9; ++k) work.prim[k] = 0; }
    const uint32_t n = io.n();
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t total_waves = gridDim.x * (BLOCK / 64);
    // with few rays, spread them over the waves of the grid instead of filling a few waves completely
    uint32_t max_active = (n + total_waves - 1u) / total_waves;
    max_active = max_active < 8u ? 8u : (max_active > 64u ? 64u : max_active);
    const uint32_t refill_at = REFILL_LANES < max_active ? REFILL_LANES : max_active;
    Walker<COUNT> w;
    w.done = true;
    typename IO::Ray ray;
    uint32_t index = 0u;
    bool busy = false;
    bool exhausted = false;  // wave-uniform
    for (;;) {
        const unsigned long long act = __ballot(busy);
        const uint32_t n_act = (uint32_t)__popcll(act);
        if (!exhausted) {
            const uint32_t room = max_active > n_act ? max_active - n_act : 0u;
            if (room >= refill_at || (n_act == 0u && room > 0u)) {
                const uint32_t my_rank = (uint32_t)__popcll(~act & ((1ull << lane) - 1ull));
                uint32_t base = 0u;
                if (lane == 0u) base = atomicAdd(cursor, room);
                base = __shfl(base, 0, 64);
                if (!busy && my_rank < room && base + my_rank < n) {
                    index = base + my_rank;
                    io.fetch(index, ray);
                    w.init(S, ray.o, ray.d, ray.tm, ray.tmin, ray.tmax);
                    busy = true;
                }
                exhausted = base + room >= n;
            }
        }
        if (__ballot(busy) == 0ull) {
            if (exhausted) break;
            continue;
        }
        if (busy) {
            w.box_run(S, ray.key, &work);
            if (!w.done) w.leaf(S, ray.rs, ray.key, &work);
            if (w.done) {
                io.store(S, index, ray, w.best);
                busy = false;
            }
        }
    }
    if (COUNT) {
        atomicAdd(work_out + 0, (unsigned long long)work.entries);
        for (int k = 0; k < 9; ++k) atomicAdd(work_out + 1 + k, (unsigned long long)work.prim[k]);
    }
}

// ---- stand-alone closest hit (rt_trace_closest / rt_trace_records) ---------------------
template <bool RECORDS>
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
    RT_D uint32_t n() const { return (uint32_t)n_rays; }
    RT_D void fetch(uint32_t i, Ray& r) const {
        const RtRay* q = rays + i;
        r.o = v3p(q->o); r.d = v3p(q->d); r.tm = q->tm; r.tmin = q->tmin; r.tmax = q->tmax;
        r.key.seed = seed; r.key.pixel = q->pixel; r.key.sample = q->sample; r.key.bounce = q->bounce;
        r.rs.r = q;
    }
    RT_D void store(const DevScene& S, uint32_t i, const Ray& r, const Cand& best) const {
        V3 o, d;
        r.rs.load(o, d);
        if (RECORDS) {
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

template <bool RECORDS, bool COUNT>
__global__ void __launch_bounds__(BLOCK) k_trace(DevScene S, const RtRay* __restrict__ rays, uint64_t n, uint64_t seed,
                                                 RtHit* __restrict__ hits, RtHitRecord* __restrict__ recs,
                                                 uint32_t* __restrict__ cursor, unsigned long long* __restrict__ work_out) {
    TraceIO<RECORDS> io;
    io.rays = rays; io.n_rays = n; io.seed = seed; io.hits = hits; io.recs = recs;
    extend_pool<COUNT>(S, io, cursor, work_out);
}

// ---- generate: camera rays of one batch (src/main.rs:206-209, src/camera.rs:61-71) ---------
__global__ void __launch_bounds__(BLOCK) k_generate(RenderConst C, PathState P) {
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t slot = blockIdx.x * BLOCK + threadIdx.x; slot < C.n_slots; slot += stride) {
        const uint32_t pix = slot % C.npix;
        const uint32_t sample = C.sample0 + slot / C.npix;
        const uint32_t x = pix % C.width, y = pix / C.width;
        V3 o, d;
        double tm;
        camera_ray(C.cam, C.seed, C.width, C.height, x, y, pix, sample, o, d, tm);
        P.ox[slot] = o.x; P.oy[slot] = o.y; P.oz[slot] = o.z;
        P.dx[slot] = d.x; P.dy[slot] = d.y; P.dz[slot] = d.z;
        P.tm[slot] = tm;
        P.tr[slot] = 1.0; P.tg[slot] = 1.0; P.tb[slot] = 1.0;
        P.lr[slot] = 0.0; P.lg[slot] = 0.0; P.lb[slot] = 0.0;
    }
}

// ---- extend ------------------------------------------------------------------------------
// queue == nullptr: the identity queue of bounce 0 (every slot of the batch is live).
struct ExtendIO {
    RenderConst C;
    PathState P;
    const uint32_t* queue;
    uint32_t count, bounce;
    struct Ray {
        V3 o, d;
        double tm;
        static constexpr double tmin = 0.0001;  // src/color.rs:41
        static constexpr double tmax = D_INF;
        RayKey key;
        RaySoA rs;
    };
    RT_D uint32_t n() const { return count; }
    RT_D void fetch(uint32_t i, Ray& r) const {
        const uint32_t slot = queue ? queue[i] : i;
        r.rs.ox = P.ox; r.rs.oy = P.oy; r.rs.oz = P.oz; r.rs.dx = P.dx; r.rs.dy = P.dy; r.rs.dz = P.dz; r.rs.slot = slot;
        r.rs.load(r.o, r.d);
        r.tm = P.tm[slot];
        r.key.seed = C.seed; r.key.pixel = slot % C.npix; r.key.sample = C.sample0 + slot / C.npix; r.key.bounce = bounce;
    }
    RT_D void store(const DevScene&, uint32_t, const Ray& r, const Cand& best) const {
        const uint32_t slot = r.rs.slot;
        P.hit_t[slot] = best.t;
        P.hit_ref[slot] = best.ref;
        P.hit_aux[slot] = best.aux;
    }
};

// counts[0], counts[1]: lengths of the two queues; counts[2]: the ray cursor of this launch (zero on entry,
// reset by the shade launch that follows).
template <bool COUNT>
__global__ void __launch_bounds__(BLOCK, RT_EXTEND_WAVES) k_extend(DevScene S, RenderConst C, PathState P, const uint32_t* __restrict__ queue,
                                                  uint32_t* __restrict__ counts, uint32_t cur, uint32_t bounce,
                                                  unsigned long long* __restrict__ work_out) {
    ExtendIO io;
    io.C = C; io.P = P; io.queue = queue; io.bounce = bounce;
    io.count = queue ? counts[cur] : C.n_slots;
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[cur ^ 1u] = 0u;  // the queue the following shade launch appends to
    extend_pool<COUNT>(S, io, counts + 2, work_out);
}

// ---- shade ---------------------------------------------------------------------------------
// The body of ray_color after world.hit (src/color.rs:42-62), unrolled from recursion into
// L += T * emitted; T *= attenuation. Survivors go to next_queue.
__global__ void __launch_bounds__(BLOCK, RT_SHADE_WAVES) k_shade(DevScene S, RenderConst C, PathState P, const uint32_t* __restrict__ queue,
                                                 uint32_t* __restrict__ counts, uint32_t cur, uint32_t* __restrict__ next_queue,
                                                 uint32_t bounce, unsigned long long* __restrict__ ray_total) {
    const uint32_t n = queue ? counts[cur] : C.n_slots;
    uint32_t* next_count = counts + (cur ^ 1u);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(ray_total, (unsigned long long)n);
        counts[2] = 0u;  // ray cursor of the next extend launch
    }
    const uint32_t stride = gridDim.x * BLOCK;
    // every lane of a wave runs the same number of iterations so that the ballot below sees whole waves
    const uint32_t n_round = (n + 63u) & ~63u;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n_round; i += stride) {
        bool alive = false;
        uint32_t slot = 0u;
        if (i < n) {
            slot = queue ? queue[i] : i;
            const uint32_t ref = P.hit_ref[slot];
            V3 T = v3(P.tr[slot], P.tg[slot], P.tb[slot]);
            V3 L = v3(P.lr[slot], P.lg[slot], P.lb[slot]);
            if (ref == RT_MISS) {
                L = L + elemul(T, v3p(C.bg));  // src/color.rs:41-43
            } else {
                Cand best;
                best.t = P.hit_t[slot]; best.ref = ref; best.aux = P.hit_aux[slot];
                const V3 o = v3(P.ox[slot], P.oy[slot], P.oz[slot]);
                const V3 d = v3(P.dx[slot], P.dy[slot], P.dz[slot]);
                const double tm = P.tm[slot];
                const RtMaterial* M = S.materials + hit_material(S, best);
                const uint32_t mkind = M->kind;
                const bool textured = material_has_texture(mkind);
                const bool want_uv = textured && S.textures[M->tex].kind == RT_T_IMAGE;
                HitRec rec;
                build_hit_record(S, o, d, tm, best, rec, want_uv);
                V3 texv = v3(0.0, 0.0, 0.0);
                if (textured) texv = texture_value(S, M->tex, rec.u, rec.v, rec.p);
                if (mkind == RT_M_DIFFUSE_LIGHT) L = L + elemul(T, texv);  // emitted, :50 (zero for every other material)
                rt::Rng rng;
                rng.set_path(C.seed, slot % C.npix, C.sample0 + slot / C.npix, bounce, rt::RNG_SCATTER);
                V3 attenuation = v3(0.0, 0.0, 0.0), dout = v3(1.0, 1.0, 1.0);
                if (material_scatter(M, d, rec, texv, rng, attenuation, dout)) {  // :52-57
                    T = elemul(T, attenuation);
                    alive = bounce + 1u < C.max_depth;  // ray_color(depth - 1) returns black at depth 0 (:37-39)
                    if (alive) {
                        P.ox[slot] = rec.p.x; P.oy[slot] = rec.p.y; P.oz[slot] = rec.p.z;
                        P.dx[slot] = dout.x; P.dy[slot] = dout.y; P.dz[slot] = dout.z;
                        P.tr[slot] = T.x; P.tg[slot] = T.y; P.tb[slot] = T.z;
                    }
                }
            }
            P.lr[slot] = L.x; P.lg[slot] = L.y; P.lb[slot] = L.z;
        }
        // compaction: one atomic per wave
        const unsigned long long mask = __ballot(alive);
        if (mask) {
            const uint32_t lane = threadIdx.x & 63u;
            uint32_t base = 0u;
            if (lane == (uint32_t)__ffsll((long long)mask) - 1u) base = atomicAdd(next_count, (uint32_t)__popcll(mask));
            base = __shfl(base, __ffsll((long long)mask) - 1, 64);
            if (alive) next_queue[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = slot;
        }
    }
}

// ---- accumulate: sum the batch's samples of each pixel in sample order ---------------------------
__global__ void __launch_bounds__(BLOCK) k_accumulate(RenderConst C, PathState P, float* __restrict__ accum) {
    const uint32_t stride = gridDim.x * BLOCK;
    const uint32_t n_samples = C.n_slots / C.npix;
    for (uint32_t pix = blockIdx.x * BLOCK + threadIdx.x; pix < C.npix; pix += stride) {
        double r = 0.0, g = 0.0, b = 0.0;
        for (uint32_t s = 0; s < n_samples; ++s) {
            const uint32_t slot = s * C.npix + pix;
            r += P.lr[slot]; g += P.lg[slot]; b += P.lb[slot];
        }
        float* a = accum + (size_t)pix * 3u;
        a[0] = (float)((double)a[0] + r);
        a[1] = (float)((double)a[1] + g);
        a[2] = (float)((double)a[2] + b);
    }
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
