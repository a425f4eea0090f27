// Counter-based random numbers shared bit-for-bit by host C++ and gfx950 device
// code. Replaces the reference's only RNG entry points, random_double and
// random_int (src/rtweekend.rs:11-20), whose thread_rng is seeded from the OS and
// cannot be reproduced.
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel Random Numbers: As Easy as
// 1, 2, 3", SC'11): 32-bit multiplies and xors only, so x86 and gfx950 agree.
//
//   draw k of stream (a, b, c) under `seed`:
//     block = philox4x32_10({a, b, c, k >> 1}, {seed_lo, seed_hi})
//     u64   = block[2*(k&1)+1] << 32 | block[2*(k&1)]
//     f64   = (u64 >> 11) * 2^-53   in [0, 1)
//   path sampling: a = pixel index (y*W + x), b = sample index, c = bounce << 16 | stream
//   scene construction: a = 0xFFFFFFFF, b = 0, c = 0, k running
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __attribute__((host)) __attribute__((device)) inline __attribute__((always_inline))
#else
#define RT_HD inline
#endif

namespace rt {

enum RngStream : uint32_t {
    RNG_JITTER = 0,   // src/main.rs:206-208: draw 0 -> u jitter, draw 1 -> v jitter
    RNG_LENS = 1,     // src/camera.rs:62 random_in_unit_disk rejection loop
    RNG_TIME = 2,     // src/camera.rs:69
    RNG_SCATTER = 3,  // Material::scatter, src/material.rs:42,78,125,205
    RNG_MEDIUM0 = 16  // + RtMedium.rng_id: src/constant_medium.rs:69
};

struct Philox4 { uint32_t v[4]; };

RT_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#if defined(__HIPCC__)
#pragma unroll
#endif
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 r;
    r.v[0] = c0; r.v[1] = c1; r.v[2] = c2; r.v[3] = c3;
    return r;
}

RT_HD double u64_to_unit(uint64_t x) { return (double)(x >> 11) * 0x1.0p-53; }

// Sequential view of one stream: next() returns draw 0, 1, 2, ... A block yields two draws.
struct Rng {
    uint32_t a, b, c, k0, k1;
    uint32_t k;        // next draw index (path streams never exceed 2^32 draws)
    uint64_t spare;    // second half of the current block
    uint32_t k_hi;     // high word of the draw index, only used by the scene stream

    RT_HD void set(uint64_t seed, uint32_t a_, uint32_t b_, uint32_t c_) {
        a = a_; b = b_; c = c_; k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32); k = 0; spare = 0; k_hi = 0;
    }
    RT_HD void set_path(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stream) {
        set(seed, pixel, sample, (bounce << 16) | stream);
    }
    RT_HD void set_scene(uint64_t seed) { set(seed, 0xFFFFFFFFu, 0u, 0u); }

    RT_HD uint64_t next_u64() {
        uint64_t r;
        if ((k & 1u) == 0u) {
            const uint32_t blk = (k >> 1) | (k_hi << 31);
            const Philox4 o = philox4x32_10(a, b, c, blk, k0, k1);
            r = ((uint64_t)o.v[1] << 32) | o.v[0];
            spare = ((uint64_t)o.v[3] << 32) | o.v[2];
        } else {
            r = spare;
        }
        ++k;
        if (k == 0u) ++k_hi;
        return r;
    }
    RT_HD double next_unit() { return u64_to_unit(next_u64()); }
    // random_double, src/rtweekend.rs:11-14
    RT_HD double uniform(double min, double max) { return min + (max - min) * next_unit(); }
};

}  // namespace rt
