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

// ---- stand-alone closest hit (rt_trace_closest / rt_trace_records) ---------------------
template <bool RECORDS, bool COUNT>
__global__ void __launch_bounds__(BLOCK) k_trace(DevScene S, const RtRay* __restrict__ rays, uint64_t n, uint64_t seed,
                                                 RtHit* __restrict__ hits, RtHitRecord* __restrict__ recs,
                                                 unsigned long long* __restrict__ work_out) {
    Work work;
    if (COUNT) { work.entries = 0; for (int k = 0; k < 9; ++k) work.prim[k] = 0; }
    const uint64_t stride = (uint64_t)gridDim.x * BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const RtRay* q = rays + i;
        const V3 o = v3p(q->o), d = v3p(q->d);
        const double tm = q->tm;
        RayKey key;
        key.seed = seed; key.pixel = q->pixel; key.sample = q->sample; key.bounce = q->bounce;
        Cand best;
        closest_hit<true, COUNT>(S, o, d, tm, q->tmin, q->tmax, 0u, S.n_nodes, 0u, 0u, key, best, &work);
        if (RECORDS) {
            RtHitRecord r;
            r.prim_id = RT_MISS; r.sub = 0u; r.mat = RT_NONE; r.front_face = 0u;
            r.t = 0.0; r.u = 0.0; r.v = 0.0;
            for (int k = 0; k < 3; ++k) { r.p[k] = 0.0; r.normal[k] = 0.0; }
            if (best.ref != RT_MISS) {
                HitRec rec;
                build_hit_record(S, o, d, tm, best, rec);
                hit_ids(S, best, r.prim_id, r.sub);
                r.p[0] = rec.p.x; r.p[1] = rec.p.y; r.p[2] = rec.p.z;
                r.normal[0] = rec.normal.x; r.normal[1] = rec.normal.y; r.normal[2] = rec.normal.z;
                r.t = rec.t; r.u = rec.u; r.v = rec.v; r.mat = rec.mat; r.front_face = rec.front_face ? 1u : 0u;
            }
            recs[i] = r;
        } else {
            RtHit h;
            h.prim_id = RT_MISS; h.sub = 0u; h.t = 0.0; h.u = 0.0; h.v = 0.0;
            if (best.ref != RT_MISS) {
                HitRec rec;
                build_hit_record(S, o, d, tm, best, rec);
                hit_ids(S, best, h.prim_id, h.sub);
                h.t = rec.t; h.u = rec.u; h.v = rec.v;
            }
            hits[i] = h;
        }
    }
    if (COUNT) {
        atomicAdd(work_out + 0, (unsigned long long)work.entries);
        for (int k = 0; k < 9; ++k) atomicAdd(work_out + 1 + k, (unsigned long long)work.prim[k]);
    }
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
template <bool COUNT>
__global__ void __launch_bounds__(BLOCK) k_extend(DevScene S, RenderConst C, PathState P, const uint32_t* __restrict__ queue,
                                                  const uint32_t* __restrict__ count, uint32_t bounce,
                                                  unsigned long long* __restrict__ work_out) {
    Work work;
    if (COUNT) { work.entries = 0; for (int k = 0; k < 9; ++k) work.prim[k] = 0; }
    const uint32_t n = queue ? *count : C.n_slots;
    const uint32_t stride = gridDim.x * BLOCK;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const uint32_t slot = queue ? queue[i] : i;
        const V3 o = v3(P.ox[slot], P.oy[slot], P.oz[slot]);
        const V3 d = v3(P.dx[slot], P.dy[slot], P.dz[slot]);
        const double tm = P.tm[slot];
        RayKey key;
        key.seed = C.seed; key.pixel = slot % C.npix; key.sample = C.sample0 + slot / C.npix; key.bounce = bounce;
        Cand best;
        closest_hit<true, COUNT>(S, o, d, tm, 0.0001, D_INF, 0u, S.n_nodes, 0u, 0u, key, best, &work);  // src/color.rs:41
        P.hit_t[slot] = best.t;
        P.hit_ref[slot] = best.ref;
        P.hit_aux[slot] = best.aux;
    }
    if (COUNT) {
        atomicAdd(work_out + 0, (unsigned long long)work.entries);
        for (int k = 0; k < 9; ++k) atomicAdd(work_out + 1 + k, (unsigned long long)work.prim[k]);
    }
}

// ---- shade ---------------------------------------------------------------------------------
// The body of ray_color after world.hit (src/color.rs:42-62), unrolled from recursion into
// L += T * emitted; T *= attenuation. Survivors go to next_queue.
__global__ void __launch_bounds__(BLOCK) k_shade(DevScene S, RenderConst C, PathState P, const uint32_t* __restrict__ queue,
                                                 const uint32_t* __restrict__ count, uint32_t* __restrict__ next_queue,
                                                 uint32_t* __restrict__ next_count, uint32_t bounce,
                                                 unsigned long long* __restrict__ ray_total) {
    const uint32_t n = queue ? *count : C.n_slots;
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ray_total, (unsigned long long)n);
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
                HitRec rec;
                build_hit_record(S, o, d, tm, best, rec);
                const RtMaterial* M = S.materials + rec.mat;
                const V3 emitted = material_emitted(S, M, rec);  // :50
                L = L + elemul(T, emitted);
                rt::Rng rng;
                rng.set_path(C.seed, slot % C.npix, C.sample0 + slot / C.npix, bounce, rt::RNG_SCATTER);
                V3 attenuation = v3(0.0, 0.0, 0.0), dout = v3(1.0, 1.0, 1.0);
                if (material_scatter(S, M, d, rec, rng, attenuation, dout)) {  // :52-57
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
