// Test infrastructure: CPU statement of the linear BVH that the HIP library can build in place of BVHNode::new
// (src/bvh.rs:17-98). The reference has no such builder, so there is nothing in it to follow here; this file fixes
// the tree by definition and the GPU builder is checked against it bit for bit. The published method: objects are
// ordered by the Morton code of their box centres (Lauterbach et al. 2009) and the tree is the binary radix tree
// over the ordered codes (Karras 2012). This statement builds the radix tree top-down by recursive splitting; the
// GPU builds it bottom-up, one thread per inner node.
//
// Definition, in f64 with + - * / rounded once each:
//   centre c = (min + max) * 0.5 per axis; lo / hi = least / greatest centre per axis over all objects;
//   q = (c - lo) / (hi - lo) * 2^21 truncated to an integer and capped at 2^21 - 1 (0 on an axis where hi == lo);
//   code = bits of qx, qy, qz interleaved, x highest (63 bits);
//   objects sorted by (code, position in the input list); the key of sorted position i is (code_i, i), 96 bits, all
//   keys distinct; a node over sorted positions [first, last] splits after the last position whose key shares more
//   leading bits with key(first) than key(last) does; one object per leaf; the box of a node is
//   surrounding_box (src/aabb.rs:33-50) of its children's boxes.
// Tokens: the tree in pre-order as 32-byte RtNode records. Inner node: lo/hi = its box rounded outward to f32
// (inverted when flat on an axis, like every inner entry), meta = RT_N_INNER, link = index of the first token
// after its subtree. Leaf: meta = RT_BVH_TOKEN_LEAF, link = position of the object in the input list.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <vector>

#include "../include/rt/abi.h"

namespace orc_lbvh {

inline uint64_t spread21(uint64_t v) {  // bit i of v -> bit 3 i
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x001F00000000FFFFull;
    v = (v | (v << 16)) & 0x001F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

struct Key {
    uint64_t code;
    uint32_t object;
};

// boxes: n x 6 doubles (min x y z, max x y z). Returns the keys in tree order.
inline std::vector<Key> sorted_keys(const double* boxes, size_t n) {
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = std::numeric_limits<double>::infinity(); hi[a] = -lo[a]; }
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            const double mn = boxes[6 * i + a], mx = boxes[6 * i + 3 + a];
            if (mn != mn || mx != mx) throw std::invalid_argument("NaN in a bounding box");
            const double c = (mn + mx) * 0.5;
            if (std::isinf(c) || c != c) throw std::invalid_argument("bounding box without a finite centre");
            lo[a] = std::fmin(lo[a], c);
            hi[a] = std::fmax(hi[a], c);
        }
    std::vector<Key> keys(n);
    for (size_t i = 0; i < n; ++i) {
        uint64_t q[3];
        for (int a = 0; a < 3; ++a) {
            const double c = (boxes[6 * i + a] + boxes[6 * i + 3 + a]) * 0.5;
            const double ext = hi[a] - lo[a];
            double s = ext > 0.0 ? (c - lo[a]) / ext * 2097152.0 : 0.0;
            if (!(s < 2097151.0)) s = 2097151.0;  // also an infinite extent (NaN quotient)
            if (!(s >= 0.0)) s = 0.0;
            q[a] = (uint64_t)s;
        }
        keys[i].code = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
        keys[i].object = (uint32_t)i;
    }
    std::stable_sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.code < b.code; });
    return keys;
}

// leading bits shared by the keys of sorted positions i and j
inline int common_prefix(const std::vector<Key>& k, size_t i, size_t j) {
    const uint64_t x = k[i].code ^ k[j].code;
    if (x) return __builtin_clzll(x);
    return 64 + __builtin_clz((uint32_t)i ^ (uint32_t)j);  // i != j
}

struct Box { double mn[3], mx[3]; };

inline float f32_down(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
    return f;
}
inline float f32_up(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
    return f;
}

struct Builder {
    const double* boxes;
    const std::vector<Key>& keys;
    std::vector<RtNode>& tokens;

    Box build(size_t first, size_t last) {
        if (first == last) {
            RtNode t{};
            t.meta = RT_BVH_TOKEN_LEAF;
            t.link = keys[first].object;
            tokens.push_back(t);
            Box b;
            for (int a = 0; a < 3; ++a) { b.mn[a] = boxes[6 * (size_t)t.link + a]; b.mx[a] = boxes[6 * (size_t)t.link + 3 + a]; }
            return b;
        }
        const int whole = common_prefix(keys, first, last);
        size_t split = first;  // last position that shares more than `whole` bits with `first`
        size_t step = last - first;
        do {
            step = (step + 1) >> 1;
            const size_t probe = split + step;
            if (probe < last && common_prefix(keys, first, probe) > whole) split = probe;
        } while (step > 1);
        const size_t at = tokens.size();
        tokens.push_back(RtNode{});
        const Box l = build(first, split), r = build(split + 1, last);
        Box b;
        bool flat = false;
        for (int a = 0; a < 3; ++a) {
            b.mn[a] = std::fmin(l.mn[a], r.mn[a]);
            b.mx[a] = std::fmax(l.mx[a], r.mx[a]);
            flat = flat || b.mn[a] == b.mx[a];
        }
        RtNode& t = tokens[at];
        for (int a = 0; a < 3; ++a) {
            t.lo[a] = flat ? std::numeric_limits<float>::infinity() : f32_down(b.mn[a]);
            t.hi[a] = flat ? -std::numeric_limits<float>::infinity() : f32_up(b.mx[a]);
        }
        t.meta = RT_N_INNER;
        t.link = (uint32_t)tokens.size();
        return b;
    }
};

// 2 n - 1 tokens for n >= 2 boxes
inline std::vector<RtNode> build_tokens(const double* boxes, size_t n) {
    if (n < 2) throw std::invalid_argument("a linear BVH needs at least two objects");
    if (n > RT_BVH_MAX_OBJECTS) throw std::invalid_argument("too many objects for a linear BVH");
    const std::vector<Key> keys = sorted_keys(boxes, n);
    std::vector<RtNode> tokens;
    tokens.reserve(2 * n - 1);
    Builder{boxes, keys, tokens}.build(0, n - 1);
    return tokens;
}

}  // namespace orc_lbvh
