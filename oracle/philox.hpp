// TEST INFRASTRUCTURE ONLY — part of the CPU oracle. Nothing under raytracer_amd/
// includes or links this file.
//
// Counter-based RNG that replaces the reference's single RNG choke point
// (src/rtweekend.rs:11-20: `rand::thread_rng()`, OS-entropy seeded and therefore
// irreproducible). Philox4x32-10 (Salmon et al., "Parallel Random Numbers: As
// Easy as 1, 2, 3", SC'11) — pure 32-bit integer arithmetic, so the x86 host and
// the gfx950 kernels produce identical bits.
//
// Draw k of stream (a, b, c) under `seed`:
//     block = philox4x32_10(counter = {a, b, c, k >> 1}, key = {seed_lo, seed_hi})
//     u64   = block[2*(k&1)+1] << 32 | block[2*(k&1)]
//     f64   = (u64 >> 11) * 2^-53          in [0, 1)
// For path sampling a = pixel index, b = sample index, c = bounce << 16 | stream.
// Scene construction uses a = 0xFFFFFFFF, b = 0, c = 0 and a running k.
#pragma once
#include <cstdint>

namespace orc {

inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
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
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum Stream : uint32_t {
    STREAM_JITTER = 0,   // pixel jitter, src/main.rs:206-208 (draw 0 = u, draw 1 = v)
    STREAM_LENS = 1,     // lens disk rejection loop, src/camera.rs:62
    STREAM_TIME = 2,     // shutter time, src/camera.rs:69
    STREAM_SCATTER = 3,  // Material::scatter draws, src/material.rs
    STREAM_MEDIUM0 = 16  // + medium rng id: ConstantMedium::hit, src/constant_medium.rs:69
};

struct RngCtx {
    uint64_t seed = 0x5EED;
    uint32_t a = 0xFFFFFFFFu, b = 0, c = 0;
    uint64_t k = 0;

    void set(uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t stream) {
        a = pixel; b = sample; c = (bounce << 16) | stream; k = 0;
    }
    void set_scene(uint64_t s) { seed = s; a = 0xFFFFFFFFu; b = 0; c = 0; k = 0; }

    uint64_t next_u64() {
        const uint32_t ctr[4] = {a, b, c, (uint32_t)(k >> 1)};
        const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        uint32_t o[4];
        philox4x32_10(ctr, key, o);
        const int j = (int)(k & 1);
        ++k;
        return ((uint64_t)o[2 * j + 1] << 32) | o[2 * j];
    }
    double next_unit() { return (double)(next_u64() >> 11) * 0x1.0p-53; }
};

}  // namespace orc
