// Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the shade and extend kernels:
// the counters are known to under-report wide coalesced streaming reads by one half, other widths are uncalibrated.
// Every kernel below moves a known number of bytes over a buffer far larger than the 256 MiB Infinity Cache.
//
//   stream16      lane i reads 16 B at 16 * i                         (the calibrated case: FETCH_SIZE = bytes / 2)
//   rec_seq96     lane i reads the six 16-byte pieces of record i     (96-byte records, read in address order)
//   rec_seq64     lane i reads the first four pieces of record i      (what the closest-hit kernel reads)
//   rec_gather96  lane i reads the six pieces of record perm[i]       (perm shuffles records within 6144-record windows:
//                                                                      what the sorting shade kernel reads)
//   hit_gather16  lane i reads 16 B at 16 * perm[i]                   (the hit slots behind the same permutation)
//   node_gather32 lane i reads the 32 bytes at 32 * hash(i) of a 1.5 GB table, both halves requested together
//                                                                     (entry fetches of a tree that lives in HBM)
//   rec_write96   64 lanes write 64 records in address order, 16 B per lane per store (shade's survivor store)
//
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/bin/fetch_calibrate tools/fetch_calibrate.hip
// Run:   rocprofv3 --pmc FETCH_SIZE -d out -- tools/bin/fetch_calibrate     (and once more with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t WINDOW = 6144;

__global__ void stream16(const uint4* __restrict__ in, uint64_t n16, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int PIECES>
__global__ void rec_seq(const uint4* __restrict__ in, uint32_t n_rec, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rec; i += gridDim.x * blockDim.x) {
        const uint4* r = in + (uint64_t)i * 6u;
#pragma unroll
        for (int p = 0; p < PIECES; ++p) { const uint4 v = r[p]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void rec_gather96(const uint4* __restrict__ in, const uint16_t* __restrict__ perm, uint32_t n_rec, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    // one workgroup per window at a time, as the shade kernel walks them
    for (uint32_t base = blockIdx.x * WINDOW; base < n_rec; base += gridDim.x * WINDOW) {
        for (uint32_t j = threadIdx.x; j < WINDOW; j += blockDim.x) {
            const uint4* r = in + (uint64_t)(base + perm[base + j]) * 6u;
#pragma unroll
            for (int p = 0; p < 6; ++p) { const uint4 v = r[p]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void hit_gather16(const uint4* __restrict__ in, const uint16_t* __restrict__ perm, uint32_t n_rec, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (uint32_t base = blockIdx.x * WINDOW; base < n_rec; base += gridDim.x * WINDOW) {
        for (uint32_t j = threadIdx.x; j < WINDOW; j += blockDim.x) {
            const uint4 v = in[base + perm[base + j]];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void node_gather32(const uint4* __restrict__ in, uint32_t n_nodes, uint32_t n_reads, uint32_t* __restrict__ out) {
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_reads; i += gridDim.x * blockDim.x) {
        uint32_t h = i * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const uint4* r = in + (uint64_t)(h % n_nodes) * 2u;
        const uint4 a = r[0], b = r[1];
        acc ^= a.x ^ a.w ^ b.x ^ b.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

__global__ void rec_write96(uint4* __restrict__ outp, uint32_t n_rec) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t waves = gridDim.x * blockDim.x >> 6;
    for (uint32_t c = wave; c * 64u < n_rec; c += waves) {
        uint4* dst = outp + (uint64_t)c * 64u * 6u;
        for (uint32_t q = lane; q < 64u * 6u; q += 64u) dst[q] = make_uint4(q, c, lane, 1u);
    }
}

int main() {
    const uint32_t n_rec = WINDOW * 5120u;  // 31.5 M records: 3.0 GB of records, 0.5 GB of 16-byte slots
    const uint64_t bytes = (uint64_t)n_rec * 96u;
    uint4 *recs, *hits;
    uint16_t* perm;
    uint32_t* out;
    CHECK(hipMalloc(&recs, bytes));
    CHECK(hipMalloc(&hits, (uint64_t)n_rec * 16u));
    CHECK(hipMalloc(&perm, (uint64_t)n_rec * 2u));
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMemset(recs, 1, bytes));
    CHECK(hipMemset(hits, 1, (uint64_t)n_rec * 16u));
    {
        std::vector<uint16_t> h(n_rec);
        uint64_t s = 0x9E3779B97F4A7C15ull;
        for (uint32_t b = 0; b < n_rec; b += WINDOW) {
            for (uint32_t j = 0; j < WINDOW; ++j) h[b + j] = (uint16_t)j;
            for (uint32_t j = WINDOW - 1u; j > 0u; --j) {  // Fisher-Yates with an xorshift generator
                s ^= s << 13; s ^= s >> 7; s ^= s << 17;
                const uint32_t k = (uint32_t)(s % (j + 1u));
                const uint16_t t = h[b + j]; h[b + j] = h[b + k]; h[b + k] = t;
            }
        }
        CHECK(hipMemcpy(perm, h.data(), (uint64_t)n_rec * 2u, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timed = [&](const char* name, double moved, auto launch) {
        CHECK(hipEventRecord(e0));
        launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipGetLastError());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-14s %8.3f GB moved by the program, %8.3f ms, %7.1f GB/s\n", name, moved / 1e9, ms, moved / ms / 1e6);
    };
    const dim3 grid(2048), block(512);
    for (int rep = 0; rep < 2; ++rep) {
        timed("stream16", (double)bytes, [&] { stream16<<<grid, block>>>(recs, bytes / 16u, out); });
        timed("rec_seq96", (double)bytes, [&] { rec_seq<6><<<grid, block>>>(recs, n_rec, out); });
        timed("rec_seq64", (double)n_rec * 64.0, [&] { rec_seq<4><<<grid, block>>>(recs, n_rec, out); });
        timed("rec_gather96", (double)bytes + 2.0 * n_rec, [&] { rec_gather96<<<dim3(256), block>>>(recs, perm, n_rec, out); });
        timed("hit_gather16", 18.0 * n_rec, [&] { hit_gather16<<<dim3(256), block>>>(hits, perm, n_rec, out); });
        // 1.5 GB of 32-byte nodes, 64 M reads of 32 bytes = 2.0 GB asked for (repeats are rare: 64 M draws of 47 M nodes)
        timed("node_gather32", 32.0 * (1u << 26), [&] { node_gather32<<<grid, block>>>(recs, (uint32_t)(bytes / 64u), 1u << 26, out); });
        timed("rec_write96", (double)bytes, [&] { rec_write96<<<grid, block>>>(recs, n_rec); });
    }
    CHECK(hipDeviceSynchronize());
    return 0;
}
