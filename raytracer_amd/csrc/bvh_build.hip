// rt_bvh_build: a binary tree over n bounding boxes, built on the device. It stands in for the host recursion of
// BVHNode::new (src/bvh.rs:17-98) where that tree need not be reproduced: objects ordered by the Morton code of their
// box centres (Lauterbach et al. 2009), the binary radix tree over the ordered codes built with one thread per inner
// node (Karras 2012), and the tree written out in pre-order. The box of a node is the union (src/aabb.rs:33-50) of the
// boxes of a contiguous range of sorted leaves, so no thread waits for another: unions over every 64-leaf block from
// its start and to its end, a sparse table over whole blocks, and each node combines at most four of those. min and max
// are exact and associative: the result is the box that uniting the children's boxes gives, bit for bit.
//
// Everything that decides the shape of the tree is exact: centres and their quantisation use + - * / in f64 without
// contraction, the sort is stable, ties between equal codes are broken by sorted position. A CPU builder following the
// same definition therefore gives the same tokens bit for bit.
//
// Device memory per object: 48 B box in, 2 x 12 B sort keys and values, 20 B tree links, 2 x 48 B block unions, 64 B tokens,
// and 48 B x log2(n / 64) per 64 objects for the table.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "../../include/rt/abi.h"
#include "host/errors.hpp"

namespace bvh {

constexpr int BLOCK = 256;
constexpr int MAX_DEPTH = 128;           // more than the 96 bits of a key (code, sorted position)
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr uint32_t RIGHT = 0x80000000u;  // in a parent link: the node is its parent's right child

constexpr uint32_t RANGE = 64u;          // leaves per block of the range unions: one wave scans a block

struct Tree {
    const uint64_t* code;   // n sorted codes
    const uint32_t* object; // n: object of sorted position
    uint32_t* parent_inner; // n - 1 (RIGHT flag); NONE for the root
    uint32_t* parent_leaf;  // n
    uint32_t* first;        // n - 1: first sorted position under inner node i
    uint32_t* last;         // n - 1
    uint32_t* split;        // n - 1: last sorted position of the left subtree
    double* from_start;     // n x 6: union of the leaf boxes from the start of the leaf's block up to the leaf
    double* to_end;         // n x 6: union from the leaf to the end of its block
    double* table;          // levels x blocks x 6: table[k][b] = union of blocks b .. b + 2^k - 1 (clipped at the end)
    uint32_t n, blocks;
};

__device__ inline uint64_t spread21(uint64_t v) {  // bit i -> bit 3 i
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x001F00000000FFFFull;
    v = (v | (v << 16)) & 0x001F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__device__ inline double wave_min(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ inline double wave_max(double v) {
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// least and greatest box centre per axis of this block's share -> partial[block][6]; flags boxes with NaN or
// without a finite centre
__global__ __launch_bounds__(BLOCK) void k_centre_bounds(const double* __restrict__ boxes, uint32_t n, double* __restrict__ partial,
                                                         uint32_t* __restrict__ bad) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool wrong = false;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK)
        for (int a = 0; a < 3; ++a) {
            const double mn = boxes[6 * i + a], mx = boxes[6 * i + 3 + a];
            const double c = (mn + mx) * 0.5;
            wrong = wrong || mn != mn || mx != mx || c != c || isinf(c);
            lo[a] = fmin(lo[a], c);
            hi[a] = fmax(hi[a], c);
        }
    __shared__ double s[BLOCK / 64][6];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int a = 0; a < 3; ++a) {
        const double l = wave_min(lo[a]), h = wave_max(hi[a]);
        if (lane == 0) { s[wave][a] = l; s[wave][3 + a] = h; }
    }
    if (wrong) *bad = 1u;
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = s[0][threadIdx.x];
        for (int w = 1; w < BLOCK / 64; ++w) v = threadIdx.x < 3 ? fmin(v, s[w][threadIdx.x]) : fmax(v, s[w][threadIdx.x]);
        partial[(uint64_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(BLOCK) void k_bounds_final(const double* __restrict__ partial, uint32_t blocks, double* __restrict__ bounds) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t b = threadIdx.x; b < blocks; b += BLOCK)
        for (int a = 0; a < 3; ++a) {
            lo[a] = fmin(lo[a], partial[(uint64_t)b * 6 + a]);
            hi[a] = fmax(hi[a], partial[(uint64_t)b * 6 + 3 + a]);
        }
    __shared__ double s[BLOCK / 64][6];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int a = 0; a < 3; ++a) {
        const double l = wave_min(lo[a]), h = wave_max(hi[a]);
        if (lane == 0) { s[wave][a] = l; s[wave][3 + a] = h; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double v = s[0][threadIdx.x];
        for (int w = 1; w < BLOCK / 64; ++w) v = threadIdx.x < 3 ? fmin(v, s[w][threadIdx.x]) : fmax(v, s[w][threadIdx.x]);
        bounds[threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(BLOCK) void k_morton(const double* __restrict__ boxes, uint32_t n, const double* __restrict__ bounds,
                                                  uint64_t* __restrict__ code, uint32_t* __restrict__ object) {
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint64_t q[3];
    for (int a = 0; a < 3; ++a) {
        const double c = (boxes[6 * i + a] + boxes[6 * i + 3 + a]) * 0.5;
        const double lo = bounds[a], ext = bounds[3 + a] - lo;
        double s = ext > 0.0 ? (c - lo) / ext * 2097152.0 : 0.0;
        if (!(s < 2097151.0)) s = 2097151.0;
        if (!(s >= 0.0)) s = 0.0;
        q[a] = (uint64_t)s;
    }
    code[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
    object[i] = (uint32_t)i;
}

// leading bits shared by the keys (code, position) of sorted positions i and j; -1 when j lies outside
__device__ inline int prefix(const uint64_t* __restrict__ code, uint32_t n, int64_t i, int64_t j) {
    if (j < 0 || j >= (int64_t)n) return -1;
    const uint64_t x = code[i] ^ code[j];
    if (x) return __clzll((long long)x);
    return 64 + __clz((int)((uint32_t)i ^ (uint32_t)j));
}

// one thread per inner node: its range of sorted positions, its split and its children (Karras 2012, algorithm 1)
__global__ __launch_bounds__(BLOCK) void k_radix_tree(Tree t) {
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t n = t.n;
    if (i >= (int64_t)n - 1) return;
    const int64_t d = prefix(t.code, n, i, i + 1) > prefix(t.code, n, i, i - 1) ? 1 : -1;
    const int below = prefix(t.code, n, i, i - d);
    int64_t reach = 2;
    while (prefix(t.code, n, i, i + reach * d) > below) reach *= 2;
    int64_t len = 0;
    for (int64_t s = reach / 2; s >= 1; s /= 2)
        if (prefix(t.code, n, i, i + (len + s) * d) > below) len += s;
    const int64_t j = i + len * d;
    const int whole = prefix(t.code, n, i, j);
    int64_t off = 0, s = len;
    do {
        s = (s + 1) / 2;
        if (prefix(t.code, n, i, i + (off + s) * d) > whole) off += s;
    } while (s > 1);
    const int64_t gamma = i + off * d + (d < 0 ? -1 : 0);
    const int64_t lo = i < j ? i : j, hi = i < j ? j : i;
    t.first[i] = (uint32_t)lo;
    t.last[i] = (uint32_t)hi;
    t.split[i] = (uint32_t)gamma;
    // the left child is leaf gamma when the range starts there, else inner node gamma; likewise gamma + 1 on the right
    if (lo == gamma) t.parent_leaf[gamma] = (uint32_t)i; else t.parent_inner[gamma] = (uint32_t)i;
    if (hi == gamma + 1) t.parent_leaf[gamma + 1] = (uint32_t)i | RIGHT; else t.parent_inner[gamma + 1] = (uint32_t)i | RIGHT;
    if (i == 0) t.parent_inner[0] = NONE;
}

struct Box6d {
    double v[6];  // min x y z, max x y z
};
__device__ inline Box6d empty_box() { return Box6d{{INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY}}; }
__device__ inline Box6d unite(Box6d a, const Box6d& b) {
    for (int k = 0; k < 3; ++k) { a.v[k] = fmin(a.v[k], b.v[k]); a.v[3 + k] = fmax(a.v[3 + k], b.v[3 + k]); }
    return a;
}
__device__ inline Box6d load_box(const double* __restrict__ p, uint64_t i) {
    Box6d b;
    for (int k = 0; k < 6; ++k) b.v[k] = p[6 * i + k];
    return b;
}
__device__ inline void store_box(double* __restrict__ p, uint64_t i, const Box6d& b) {
    for (int k = 0; k < 6; ++k) p[6 * i + k] = b.v[k];
}

// One wave per block of 64 sorted leaves: the unions from the block's start up to each leaf and from each leaf to the
// block's end (two wave scans), and the union of the whole block as level 0 of the table.
__global__ __launch_bounds__(BLOCK) void k_block_unions(Tree t, const double* __restrict__ boxes) {
    const uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const bool live = i < t.n;
    Box6d fwd = live ? load_box(boxes, t.object[i]) : empty_box();
    Box6d bwd = fwd;
    for (uint32_t o = 1; o < RANGE; o <<= 1) {
        Box6d a, b;
        for (int k = 0; k < 6; ++k) { a.v[k] = __shfl_up(fwd.v[k], o, 64); b.v[k] = __shfl_down(bwd.v[k], o, 64); }
        if (lane >= o) fwd = unite(fwd, a);
        if (lane + o < RANGE) bwd = unite(bwd, b);
    }
    if (live) {
        store_box(t.from_start, i, fwd);
        store_box(t.to_end, i, bwd);
        if (lane == 0u) store_box(t.table, i / RANGE, bwd);
    }
}

// level k of the table from level k - 1
__global__ __launch_bounds__(BLOCK) void k_table_level(Tree t, uint32_t k) {
    const uint64_t b = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (b >= t.blocks) return;
    const double* below = t.table + 6ull * t.blocks * (k - 1);
    Box6d x = load_box(below, b);
    const uint64_t right = b + (1ull << (k - 1));
    if (right < t.blocks) x = unite(x, load_box(below, right));
    store_box(t.table + 6ull * t.blocks * k, b, x);
}

// the union of the leaf boxes of sorted positions [l, r], l < r
__device__ inline Box6d range_box(const Tree& t, const double* __restrict__ boxes, uint32_t l, uint32_t r) {
    const uint32_t bl = l / RANGE, br = r / RANGE;
    if (bl == br) {
        if (l % RANGE == 0u) return load_box(t.from_start, r);
        if (r % RANGE == RANGE - 1u || r == t.n - 1u) return load_box(t.to_end, l);
        Box6d x = load_box(boxes, t.object[l]);
        for (uint32_t i = l + 1u; i <= r; ++i) x = unite(x, load_box(boxes, t.object[i]));
        return x;
    }
    Box6d x = unite(load_box(t.to_end, l), load_box(t.from_start, r));
    if (br > bl + 1u) {  // whole blocks bl + 1 .. br - 1: two overlapping table entries
        const uint32_t len = br - bl - 1u;
        const uint32_t k = 31u - (uint32_t)__clz((int)len);
        const double* level = t.table + 6ull * t.blocks * k;
        x = unite(x, load_box(level, bl + 1u));
        x = unite(x, load_box(level, br - (1u << k)));
    }
    return x;
}

__device__ inline float f32_down(double v) {
    float f = (float)v;
    if ((double)f > v) f = nextafterf(f, -INFINITY);
    return f;
}
__device__ inline float f32_up(double v) {
    float f = (float)v;
    if ((double)f < v) f = nextafterf(f, INFINITY);
    return f;
}

// One thread per node. Pre-order position: the root is token 0; a left child follows its parent; a right child
// follows the 2 L - 1 tokens of the left subtree (L leaves).
__global__ __launch_bounds__(BLOCK) void k_tokens(Tree t, const double* __restrict__ boxes, RtNode* __restrict__ tokens) {
    const uint64_t id = (uint64_t)blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t n = t.n;
    if (id >= 2ull * n - 1) return;
    const bool inner = id < n - 1;
    const uint64_t self = inner ? id : id - (n - 1);
    uint64_t pos = 0;
    // keys are 96 bits wide, so no node lies deeper than that; the bound keeps a damaged link from looping for ever
    int depth = 0;
    for (uint32_t up = inner ? t.parent_inner[self] : t.parent_leaf[self]; up != NONE && depth < MAX_DEPTH; ++depth) {
        const uint32_t p = up & ~RIGHT;
        pos += (up & RIGHT) ? 2ull * (t.split[p] - t.first[p] + 1u) : 1ull;
        up = t.parent_inner[p];
    }
    RtNode out;
    if (inner) {
        const Box6d box = range_box(t, boxes, t.first[self], t.last[self]);
        const double* b = box.v;
        const bool flat = b[0] == b[3] || b[1] == b[4] || b[2] == b[5];
        for (int a = 0; a < 3; ++a) {
            out.lo[a] = flat ? INFINITY : f32_down(b[a]);
            out.hi[a] = flat ? -INFINITY : f32_up(b[3 + a]);
        }
        out.meta = RT_N_INNER;
        out.link = (uint32_t)(pos + 2ull * (t.last[self] - t.first[self] + 1u) - 1ull);
    } else {
        for (int a = 0; a < 3; ++a) { out.lo[a] = 0.0f; out.hi[a] = 0.0f; }
        out.meta = RT_BVH_TOKEN_LEAF;
        out.link = t.object[self];
    }
    tokens[pos] = out;
}

// Scratch memory of one build, carved out of one block: every array starts on a 256-byte boundary.
struct Scratch {
    double *partial, *bounds, *from_start, *to_end, *table;
    uint64_t *code_in, *code;
    uint32_t *obj_in, *obj, *parent_inner, *parent_leaf, *first, *last, *split, *bad;
    char* sort;
    size_t sort_bytes, total;
    uint32_t reduce_blocks, blocks, levels;
};

inline size_t align256(size_t b) { return (b + 255u) & ~(size_t)255u; }

// Lays the arrays of a build over n objects out from `base` (which may be null: then only `total` is of use).
inline Scratch carve(char* base, uint32_t n, size_t sort_bytes) {
    Scratch s{};
    s.reduce_blocks = (uint32_t)std::min<uint64_t>(1024, ((uint64_t)n + BLOCK - 1) / BLOCK);
    s.blocks = (n + RANGE - 1u) / RANGE;
    s.levels = 1;
    while ((2u << (s.levels - 1)) <= s.blocks) ++s.levels;  // levels - 1 = floor(log2(blocks))
    s.sort_bytes = sort_bytes;
    size_t off = 0;
    auto take = [&](auto** out, size_t count) {
        using T = std::remove_pointer_t<std::remove_pointer_t<decltype(out)>>;
        *out = reinterpret_cast<T*>(base + off);
        off += align256(count * sizeof(T) + 16);
    };
    const size_t inner = n - 1;
    take(&s.partial, 6ull * s.reduce_blocks);
    take(&s.bounds, 6);
    take(&s.bad, 1);
    take(&s.code_in, n);
    take(&s.code, n);
    take(&s.obj_in, n);
    take(&s.obj, n);
    take(&s.parent_inner, inner);
    take(&s.parent_leaf, n);
    take(&s.first, inner);
    take(&s.last, inner);
    take(&s.split, inner);
    take(&s.from_start, 6ull * n);
    take(&s.to_end, 6ull * n);
    take(&s.table, 6ull * s.blocks * s.levels);
    take(&s.sort, sort_bytes);
    s.total = off;
    return s;
}

}  // namespace bvh

#define BVH_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) {                                                                            \
            (void)hipGetLastError();                                                                       \
            return rt::set_error(e_ == hipErrorOutOfMemory ? RT_E_NOMEM : RT_E_DEVICE,                     \
                                 std::string(#expr) + ": " + hipGetErrorString(e_));                       \
        }                                                                                                  \
    } while (0)

namespace bvh {

int check_count(uint64_t n64) {
    if (n64 < 2) return rt::set_error(RT_E_INVALID, "a tree needs at least two objects");
    if (n64 > RT_BVH_MAX_OBJECTS) return rt::set_error(RT_E_INVALID, "too many objects for one tree");
    return RT_OK;
}

int sort_temp_bytes(uint32_t n, size_t* bytes) {
    *bytes = 0;
    BVH_TRY(rocprim::radix_sort_pairs(nullptr, *bytes, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                      (size_t)n, 0u, 63u, (hipStream_t) nullptr));
    return RT_OK;
}

// The build proper: everything is enqueued on `stream`, nothing is allocated and nothing waits. d_flag (may be
// null) receives 1 when a box holds NaN or has no finite centre.
int enqueue_build(const double* d_boxes, uint32_t n, RtNode* d_tokens, uint32_t* d_flag, const Scratch& m, hipStream_t stream) {
    const uint32_t inner = n - 1;
    BVH_TRY(hipMemsetAsync(m.bad, 0, sizeof(uint32_t), stream));
    const uint32_t per_leaf = (n + BLOCK - 1) / BLOCK, per_inner = (inner + BLOCK - 1) / BLOCK;
    const uint32_t per_node = (uint32_t)((2ull * n - 1 + BLOCK - 1) / BLOCK);
    hipLaunchKernelGGL(k_centre_bounds, dim3(m.reduce_blocks), dim3(BLOCK), 0, stream, d_boxes, n, m.partial, m.bad);
    hipLaunchKernelGGL(k_bounds_final, dim3(1), dim3(BLOCK), 0, stream, m.partial, m.reduce_blocks, m.bounds);
    hipLaunchKernelGGL(k_morton, dim3(per_leaf), dim3(BLOCK), 0, stream, d_boxes, n, m.bounds, m.code_in, m.obj_in);
    BVH_TRY(hipGetLastError());
    size_t sort_bytes = m.sort_bytes;
    BVH_TRY(rocprim::radix_sort_pairs(m.sort, sort_bytes, m.code_in, m.code, m.obj_in, m.obj, (size_t)n, 0u, 63u, stream));
    Tree t{m.code, m.obj, m.parent_inner, m.parent_leaf, m.first, m.last, m.split, m.from_start, m.to_end, m.table, n, m.blocks};
    hipLaunchKernelGGL(k_radix_tree, dim3(per_inner), dim3(BLOCK), 0, stream, t);
    hipLaunchKernelGGL(k_block_unions, dim3(per_leaf), dim3(BLOCK), 0, stream, t, d_boxes);
    for (uint32_t k = 1; k < m.levels; ++k)
        hipLaunchKernelGGL(k_table_level, dim3((m.blocks + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, t, k);
    hipLaunchKernelGGL(k_tokens, dim3(per_node), dim3(BLOCK), 0, stream, t, d_boxes, d_tokens);
    BVH_TRY(hipGetLastError());
    if (d_flag) BVH_TRY(hipMemcpyAsync(d_flag, m.bad, sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
    return RT_OK;
}

struct DeviceBuffers {  // freed on every way out
    void* p[8] = {};
    int used = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ~DeviceBuffers() {
        for (int i = 0; i < used; ++i) (void)hipFree(p[i]);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    hipError_t get(void** out, size_t bytes) {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, bytes + 16);
        if (e == hipSuccess) { p[used++] = q; *out = q; }
        return e;
    }
};

int need_device() {
    int device_count = 0;
    if (hipGetDeviceCount(&device_count) != hipSuccess || device_count < 1) {
        (void)hipGetLastError();
        return rt::set_error(RT_E_DEVICE, "building a tree on the device needs a HIP device; there is no CPU path");
    }
    return RT_OK;
}

}  // namespace bvh

extern "C" int rt_bvh_build_device(const double* boxes_dev, uint64_t n64, RtNode* tokens_dev, uint32_t* flag_dev, void* stream,
                                   double* ms_device) {
    using namespace bvh;
    if (!boxes_dev || !tokens_dev) return rt::set_error(RT_E_INVALID, "null argument");
    if (int rc = check_count(n64)) return rc;
    if (int rc = need_device()) return rc;
    const uint32_t n = (uint32_t)n64;
    size_t sort_bytes = 0;
    if (int rc = sort_temp_bytes(n, &sort_bytes)) return rc;
    Scratch m = carve(nullptr, n, sort_bytes);
    DeviceBuffers mem;
    void* base = nullptr;
    BVH_TRY(mem.get(&base, m.total));
    m = carve(static_cast<char*>(base), n, sort_bytes);
    BVH_TRY(hipEventCreate(&mem.e0));
    BVH_TRY(hipEventCreate(&mem.e1));
    hipStream_t st = (hipStream_t)stream;
    BVH_TRY(hipEventRecord(mem.e0, st));
    if (int rc = enqueue_build(boxes_dev, n, tokens_dev, flag_dev, m, st)) return rc;
    BVH_TRY(hipEventRecord(mem.e1, st));
    BVH_TRY(hipStreamSynchronize(st));  // the scratch memory is released on return
    if (ms_device) {
        float ms = 0.0f;
        BVH_TRY(hipEventElapsedTime(&ms, mem.e0, mem.e1));
        *ms_device = ms;
    }
    return RT_OK;
}

// ---- tokens -> entry stream, and refitting the boxes of a built tree ------------------------------------------
namespace bvh {

// how many stream entries token k turns into: an inner token is one entry, a leaf token the entries of its object
__global__ __launch_bounds__(BLOCK) void k_token_sizes(const RtNode* __restrict__ tokens, uint32_t n_tokens, const uint32_t* __restrict__ obj_offsets,
                                                       uint32_t* __restrict__ sizes) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k > n_tokens) return;
    uint32_t sz = 0u;  // sizes[n_tokens] = 0: the scan then leaves the total there
    if (k < n_tokens) {
        const RtNode t = tokens[k];
        sz = t.meta == RT_BVH_TOKEN_LEAF ? obj_offsets[t.link + 1u] - obj_offsets[t.link] : 1u;
    }
    sizes[k] = sz;
}

// pos = exclusive prefix sum of the sizes. Inner token: the entry with its link translated; leaf token: the object's
// entries, links of box and medium entries (the only kinds that carry one) moved from the object's own range to where
// the entries now stand.
__global__ __launch_bounds__(BLOCK) void k_expand(const RtNode* __restrict__ tokens, uint32_t n_tokens, const uint32_t* __restrict__ pos,
                                                  const RtNode* __restrict__ obj_entries, const uint32_t* __restrict__ obj_offsets,
                                                  RtNode* __restrict__ out) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n_tokens) return;
    RtNode t = tokens[k];
    const uint32_t at = pos[k];
    if (t.meta != RT_BVH_TOKEN_LEAF) {
        t.link = pos[t.link];
        out[at] = t;
        return;
    }
    const uint32_t first = obj_offsets[t.link], last = obj_offsets[t.link + 1u];
    for (uint32_t e = first; e < last; ++e) {
        RtNode x = obj_entries[e];
        const uint32_t kind = x.meta & 15u;
        if (kind == RT_N_INNER || kind == RT_N_MEDIUM) x.link = x.link - first + at;
        out[at + (e - first)] = x;
    }
}

// refit: leaf rank (position among the leaves, which is the sorted position the tree was built over) per token
__global__ __launch_bounds__(BLOCK) void k_leaf_flags(const RtNode* __restrict__ tokens, uint32_t n_tokens, uint32_t* __restrict__ flags) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k > n_tokens) return;
    flags[k] = (k < n_tokens && tokens[k].meta == RT_BVH_TOKEN_LEAF) ? 1u : 0u;
}
__global__ __launch_bounds__(BLOCK) void k_rank_objects(const RtNode* __restrict__ tokens, uint32_t n_tokens, const uint32_t* __restrict__ rank,
                                                        uint32_t* __restrict__ object, uint32_t n_objects, uint32_t* __restrict__ bad) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n_tokens) return;
    const RtNode t = tokens[k];
    if (t.meta != RT_BVH_TOKEN_LEAF) return;
    if (t.link >= n_objects) { *bad = 1u; return; }
    object[rank[k]] = t.link;
}
__global__ __launch_bounds__(BLOCK) void k_refit(Tree t, const double* __restrict__ boxes, RtNode* __restrict__ tokens, uint32_t n_tokens,
                                                 const uint32_t* __restrict__ rank) {
    const uint32_t k = blockIdx.x * BLOCK + threadIdx.x;
    if (k >= n_tokens) return;
    RtNode x = tokens[k];
    if (x.meta == RT_BVH_TOKEN_LEAF || x.link > n_tokens || x.link <= k) return;
    const uint32_t first = rank[k], past = rank[x.link];  // leaves under an inner token: those between it and its link
    if (past <= first || past > t.n) return;
    const Box6d box = past - first == 1u ? load_box(boxes, t.object[first]) : range_box(t, boxes, first, past - 1u);
    const double* b = box.v;
    const bool flat = b[0] == b[3] || b[1] == b[4] || b[2] == b[5];
    for (int a = 0; a < 3; ++a) {
        x.lo[a] = flat ? INFINITY : f32_down(b[a]);
        x.hi[a] = flat ? -INFINITY : f32_up(b[3 + a]);
    }
    tokens[k] = x;
}

int scan_temp_bytes(uint32_t n, size_t* bytes) {
    *bytes = 0;
    BVH_TRY(rocprim::exclusive_scan(nullptr, *bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)n, rocprim::plus<uint32_t>(), (hipStream_t) nullptr));
    return RT_OK;
}

}  // namespace bvh

extern "C" int rt_bvh_expand_device(const RtNode* tokens_dev, uint64_t n_objects, const RtNode* obj_entries_dev, const uint32_t* obj_offsets_dev,
                                    uint64_t n_entries_out, RtNode* out_dev, void* stream) {
    using namespace bvh;
    if (!tokens_dev || !obj_entries_dev || !obj_offsets_dev || !out_dev) return rt::set_error(RT_E_INVALID, "null argument");
    if (int rc = check_count(n_objects)) return rc;
    if (n_entries_out >= 0xFFFFFFFFull) return rt::set_error(RT_E_INVALID, "too many entries");
    const uint32_t n_tokens = (uint32_t)(2 * n_objects - 1);
    size_t scan_bytes = 0;
    if (int rc = scan_temp_bytes(n_tokens + 1u, &scan_bytes)) return rc;
    DeviceBuffers mem;
    void *d_sizes = nullptr, *d_pos = nullptr, *d_tmp = nullptr;
    BVH_TRY(mem.get(&d_sizes, (size_t)(n_tokens + 1u) * sizeof(uint32_t)));
    BVH_TRY(mem.get(&d_pos, (size_t)(n_tokens + 1u) * sizeof(uint32_t)));
    BVH_TRY(mem.get(&d_tmp, scan_bytes));
    hipStream_t st = (hipStream_t)stream;
    const uint32_t grid = (n_tokens + 1u + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(k_token_sizes, dim3(grid), dim3(BLOCK), 0, st, tokens_dev, n_tokens, obj_offsets_dev, (uint32_t*)d_sizes);
    BVH_TRY(hipGetLastError());
    BVH_TRY(rocprim::exclusive_scan(d_tmp, scan_bytes, (uint32_t*)d_sizes, (uint32_t*)d_pos, 0u, (size_t)n_tokens + 1u, rocprim::plus<uint32_t>(), st));
    uint32_t total = 0;
    BVH_TRY(hipMemcpyAsync(&total, (uint32_t*)d_pos + n_tokens, sizeof total, hipMemcpyDeviceToHost, st));
    BVH_TRY(hipStreamSynchronize(st));
    if (total != n_entries_out)
        return rt::set_error(RT_E_INVALID, "the tokens expand to " + std::to_string(total) + " entries, the caller expects " + std::to_string(n_entries_out));
    hipLaunchKernelGGL(k_expand, dim3(grid), dim3(BLOCK), 0, st, tokens_dev, n_tokens, (const uint32_t*)d_pos, obj_entries_dev, obj_offsets_dev, out_dev);
    BVH_TRY(hipGetLastError());
    BVH_TRY(hipStreamSynchronize(st));  // the scratch memory is released on return
    return RT_OK;
}

extern "C" int rt_bvh_refit_device(RtNode* tokens_dev, uint64_t n_objects, const double* boxes_dev, uint32_t* flag_dev, void* stream) {
    using namespace bvh;
    if (!tokens_dev || !boxes_dev) return rt::set_error(RT_E_INVALID, "null argument");
    if (int rc = check_count(n_objects)) return rc;
    const uint32_t n = (uint32_t)n_objects, n_tokens = 2u * n - 1u;
    size_t scan_bytes = 0;
    if (int rc = scan_temp_bytes(n_tokens + 1u, &scan_bytes)) return rc;
    Scratch m = carve(nullptr, n, scan_bytes);  // the same arrays as a build (the sort area holds the scan's temporary storage)
    DeviceBuffers mem;
    void *base = nullptr, *d_flags = nullptr, *d_rank = nullptr;
    BVH_TRY(mem.get(&base, m.total));
    BVH_TRY(mem.get(&d_flags, (size_t)(n_tokens + 1u) * sizeof(uint32_t)));
    BVH_TRY(mem.get(&d_rank, (size_t)(n_tokens + 1u) * sizeof(uint32_t)));
    m = carve(static_cast<char*>(base), n, scan_bytes);
    hipStream_t st = (hipStream_t)stream;
    const uint32_t per_token = (n_tokens + 1u + BLOCK - 1) / BLOCK, per_leaf = (n + BLOCK - 1) / BLOCK;
    BVH_TRY(hipMemsetAsync(m.bad, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(k_leaf_flags, dim3(per_token), dim3(BLOCK), 0, st, tokens_dev, n_tokens, (uint32_t*)d_flags);
    BVH_TRY(hipGetLastError());
    size_t sb = scan_bytes;
    BVH_TRY(rocprim::exclusive_scan(m.sort, sb, (uint32_t*)d_flags, (uint32_t*)d_rank, 0u, (size_t)n_tokens + 1u, rocprim::plus<uint32_t>(), st));
    hipLaunchKernelGGL(k_rank_objects, dim3(per_token), dim3(BLOCK), 0, st, tokens_dev, n_tokens, (const uint32_t*)d_rank, m.obj, n, m.bad);
    hipLaunchKernelGGL(k_centre_bounds, dim3(m.reduce_blocks), dim3(BLOCK), 0, st, boxes_dev, n, m.partial, m.bad);  // only for its NaN check
    Tree t{m.code, m.obj, m.parent_inner, m.parent_leaf, m.first, m.last, m.split, m.from_start, m.to_end, m.table, n, m.blocks};
    hipLaunchKernelGGL(k_block_unions, dim3(per_leaf), dim3(BLOCK), 0, st, t, boxes_dev);
    for (uint32_t k = 1; k < m.levels; ++k)
        hipLaunchKernelGGL(k_table_level, dim3((m.blocks + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, st, t, k);
    hipLaunchKernelGGL(k_refit, dim3(per_token), dim3(BLOCK), 0, st, t, boxes_dev, tokens_dev, n_tokens, (const uint32_t*)d_rank);
    BVH_TRY(hipGetLastError());
    if (flag_dev) BVH_TRY(hipMemcpyAsync(flag_dev, m.bad, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    BVH_TRY(hipStreamSynchronize(st));  // the scratch memory is released on return
    return RT_OK;
}

// Host buffers in and out: the tree over `boxes` and its tokens turned into stream entries, without the tokens ever
// leaving the device.
extern "C" int rt_bvh_build_stream(const double* boxes, uint64_t n_objects, const RtNode* obj_entries, const uint32_t* obj_offsets,
                                   RtNode* out, double* ms_device) {
    using namespace bvh;
    if (!boxes || !obj_entries || !obj_offsets || !out) return rt::set_error(RT_E_INVALID, "null argument");
    if (int rc = check_count(n_objects)) return rc;
    if (int rc = need_device()) return rc;
    const uint32_t n = (uint32_t)n_objects;
    const uint64_t n_obj_entries = obj_offsets[n];
    const uint64_t n_out = (uint64_t)(n - 1u) + n_obj_entries;
    if (obj_offsets[0] != 0u || n_out >= 0xFFFFFFFFull) return rt::set_error(RT_E_INVALID, "bad object offsets");
    DeviceBuffers mem;
    void *d_boxes = nullptr, *d_tokens = nullptr, *d_flag = nullptr, *d_entries = nullptr;
    BVH_TRY(mem.get(&d_boxes, 6ull * n * sizeof(double)));
    BVH_TRY(mem.get(&d_tokens, (2ull * n - 1) * sizeof(RtNode)));
    BVH_TRY(mem.get(&d_flag, sizeof(uint32_t)));
    BVH_TRY(mem.get(&d_entries, n_obj_entries * sizeof(RtNode) + (n + 1ull) * sizeof(uint32_t)));
    uint32_t* d_offsets = reinterpret_cast<uint32_t*>(static_cast<RtNode*>(d_entries) + n_obj_entries);
    BVH_TRY(hipMemcpy(d_boxes, boxes, 6ull * n * sizeof(double), hipMemcpyHostToDevice));
    BVH_TRY(hipMemcpy(d_entries, obj_entries, n_obj_entries * sizeof(RtNode), hipMemcpyHostToDevice));
    BVH_TRY(hipMemcpy(d_offsets, obj_offsets, (n + 1ull) * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (int rc = rt_bvh_build_device(static_cast<const double*>(d_boxes), n_objects, static_cast<RtNode*>(d_tokens), static_cast<uint32_t*>(d_flag),
                                     nullptr, ms_device))
        return rc;
    uint32_t bad = 0;
    BVH_TRY(hipMemcpy(&bad, d_flag, sizeof(bad), hipMemcpyDeviceToHost));
    if (bad) return rt::set_error(RT_E_INVALID, "a bounding box holds NaN or has no finite centre");
    void* d_out = nullptr;
    BVH_TRY(mem.get(&d_out, n_out * sizeof(RtNode)));
    if (int rc = rt_bvh_expand_device(static_cast<const RtNode*>(d_tokens), n_objects, static_cast<const RtNode*>(d_entries), d_offsets, n_out,
                                      static_cast<RtNode*>(d_out), nullptr))
        return rc;
    BVH_TRY(hipMemcpy(out, d_out, n_out * sizeof(RtNode), hipMemcpyDeviceToHost));
    return RT_OK;
}

// ---- the build without any allocation -----------------------------------------------------------------------
// Scratch memory of rt_bvh_build_device_noalloc: one array per GPU the library runs on, held until the process ends.
// In host code the array's name is only a stand-in for the symbol; its device address on the current GPU comes from
// hipGetSymbolAddress, once per device, and is what every kernel argument and memset below uses.
static __device__ __align__(256) unsigned char scratch[RT_BVH_NOALLOC_SCRATCH_BYTES];

namespace bvh {
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
char* g_scratch_ptr[MAX_DEVICES] = {};  // device address of `scratch` on each GPU, looked up on first use

// The scratch array of the device `stream` belongs to. Kernels enqueued on a stream run on the stream's device, so that
// device's copy of the array is the one they must be given; a stream of another device than the current one is refused
// (the launches below go to the current device).
int scratch_for_stream(hipStream_t stream, char** out) {
    int dev = -1;
    BVH_TRY(hipGetDevice(&dev));
    if (stream) {
        int stream_dev = -1;
        BVH_TRY(hipStreamGetDevice(stream, &stream_dev));
        if (stream_dev != dev) return rt::set_error(RT_E_INVALID, "the stream belongs to another device than the current one");
    }
    if (dev < 0 || dev >= MAX_DEVICES) return rt::set_error(RT_E_UNSUPPORTED, "device ordinal beyond the scratch table");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        BVH_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return rt::set_error(RT_E_DEVICE, "hipGetSymbolAddress returned a null address for the scratch array");
        g_scratch_ptr[dev] = static_cast<char*>(p);
    }
    *out = g_scratch_ptr[dev];
    return RT_OK;
}
}  // namespace bvh

extern "C" int rt_bvh_build_device_noalloc(const double* boxes_dev, uint64_t n64, RtNode* tokens_dev, uint32_t* flag_dev, void* stream) {
    using namespace bvh;
    if (!boxes_dev || !tokens_dev) return rt::set_error(RT_E_INVALID, "null argument");
    if (n64 < 2) return rt::set_error(RT_E_INVALID, "a tree needs at least two objects");
    if (n64 > RT_BVH_NOALLOC_MAX_OBJECTS)
        return rt::set_error(RT_E_INVALID, "rt_bvh_build_device_noalloc handles at most " + std::to_string(RT_BVH_NOALLOC_MAX_OBJECTS) +
                                               " objects (" + std::to_string(n64) + " given); use rt_bvh_build_device");
    const uint32_t n = (uint32_t)n64;
    size_t sort_bytes = 0;
    if (int rc = sort_temp_bytes(n, &sort_bytes)) return rc;  // a size query: nothing is allocated or launched
    const size_t need = carve(nullptr, n, sort_bytes).total;
    if (need > (size_t)RT_BVH_NOALLOC_SCRATCH_BYTES)
        return rt::set_error(RT_E_INVALID, "the build needs " + std::to_string(need) + " bytes of scratch memory, the static area holds " +
                                               std::to_string((size_t)RT_BVH_NOALLOC_SCRATCH_BYTES));
    char* base = nullptr;
    if (int rc = scratch_for_stream((hipStream_t)stream, &base)) return rc;
    // every array of the build is written in full before it is read, except the flag word, which enqueue_build clears;
    // what an earlier call left behind beyond this call's n is never read
    return enqueue_build(boxes_dev, n, tokens_dev, flag_dev, carve(base, n, sort_bytes), (hipStream_t)stream);
}

extern "C" int rt_bvh_build(const double* boxes, uint64_t n64, RtNode* tokens_out, double* ms_device) {
    using namespace bvh;
    if (!boxes || !tokens_out) return rt::set_error(RT_E_INVALID, "null argument");
    if (int rc = check_count(n64)) return rc;
    if (int rc = need_device()) return rc;
    const uint32_t n = (uint32_t)n64;
    DeviceBuffers mem;
    void *d_boxes = nullptr, *d_tokens = nullptr, *d_flag = nullptr;
    BVH_TRY(mem.get(&d_boxes, 6ull * n * sizeof(double)));
    BVH_TRY(mem.get(&d_tokens, (2ull * n - 1) * sizeof(RtNode)));
    BVH_TRY(mem.get(&d_flag, sizeof(uint32_t)));
    BVH_TRY(hipMemcpy(d_boxes, boxes, 6ull * n * sizeof(double), hipMemcpyHostToDevice));
    if (int rc = rt_bvh_build_device(static_cast<const double*>(d_boxes), n64, static_cast<RtNode*>(d_tokens), static_cast<uint32_t*>(d_flag),
                                     nullptr, ms_device))
        return rc;
    uint32_t bad = 0;
    BVH_TRY(hipMemcpy(&bad, d_flag, sizeof(bad), hipMemcpyDeviceToHost));
    if (bad) return rt::set_error(RT_E_INVALID, "a bounding box holds NaN or has no finite centre");
    BVH_TRY(hipMemcpy(tokens_out, d_tokens, (2ull * n - 1) * sizeof(RtNode), hipMemcpyDeviceToHost));
    return RT_OK;
}
