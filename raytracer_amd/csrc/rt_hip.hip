// Implementation of the device-facing entry points of include/rt/abi.h: scene
// upload, closest-hit queries, the wavefront render loop and resolve.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt/abi.h"
#include "device/kernels.h"
#include "host/errors.hpp"

namespace {

using rtd::DevScene;
using rtd::PathState;
using rtd::RenderConst;

constexpr uint64_t kDefaultSlotTarget = 640ull << 20;  // path slots per wavefront batch when the caller does not say (145 GiB)
constexpr uint64_t kBytesPerSlot = 2 * sizeof(rtd::RayRec) + sizeof(rtd::HitSlot) + 3 * sizeof(double);
constexpr int kWorkWords = 10;                      // entries + 9 kinds
constexpr int kRenderWorkWords = 14;                // rt_render_work: the same + slots that gathered light + media asked before the walk, their sphere queries + a spare word
constexpr int kCounterWords = 16;                   // size of every device counter block the kernels write into

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return rt::set_error(e_ == hipErrorOutOfMemory ? RT_E_NOMEM : RT_E_DEVICE,                     \
                                 std::string(#expr) + ": " + hipGetErrorString(e_));                       \
    } while (0)

struct Workspace {
    uint64_t cap_slots = 0;
    char* base = nullptr;  // one allocation: two record arrays, hit slots, three radiance columns
    PathState P{};
    uint32_t* counts = nullptr;              // [0], [1] queue lengths, [2] ray cursor of the extend launch, [3] cursor of rt_trace_*
    unsigned long long* ray_total = nullptr; // [1] + work counters [kRenderWorkWords]
    std::vector<hipEvent_t> events;
};

}  // namespace

struct RtScene {
    int device = 0;
    DevScene S{};
    std::vector<void*> allocs;
    Workspace ws;
    uint32_t n_cus = 256;
    uint32_t grid_extend = 256;  // resident blocks of the persistent closest-hit kernels
    rtd::LdsPlan plan{};         // tables the closest-hit kernels stage in LDS
    uint32_t lds_bytes = 0;
    rtd::LdsPlan plan_shade{};   // tables the shade kernel stages in LDS
    uint32_t lds_bytes_shade = 0;
    uint32_t grid_shade = 256;
    bool extend_lds = false, shade_lds = false;  // which variant of each kernel this scene runs
    uint32_t shade_rounds = 0;  // records per thread of the sorting shade kernel's window; 0: the plain kernel
    bool box_media = false;   // some box of six sides sits alone inside wrappers (on its own or as a medium's boundary)
    bool general_media = false;  // some medium's boundary is neither one sphere nor one box alone inside wrappers
    bool boxed_media = false;    // some medium's boundary is one box alone inside zero or more wrappers
    bool per_axis = false;       // RT_DESC_BOXES_MAY_NOT_BOUND: the kernels run the reference's per-axis box rule
    int profile = 3;          // extend kernel profile (bits as in kernels.h): 1 box_media, 2 general boundary walk, 4 vote
    // Device staging areas of the entry points that take host buffers, kept between calls and grown on demand
    // (rt_trace_*, rt_render, rt_path_step): [0] input, [1] output, [2] small extras.
    void* stage[3] = {nullptr, nullptr, nullptr};
    size_t stage_bytes[3] = {0, 0, 0};
    hipEvent_t ev_trace[2] = {nullptr, nullptr};
};

namespace {

// ---- validation: nothing a kernel indexes with may point outside its table ------------------
int validate(const RtSceneDesc* d) {
    auto bad = [](const char* what) { return rt::set_error(RT_E_INVALID, std::string("scene description: ") + what); };
    if (!d) return bad("null");
    if (d->abi_version != RT_ABI_VERSION) return bad("abi_version mismatch");
    if (d->flags & ~RT_DESC_BOXES_MAY_NOT_BOUND) return bad("unknown flag");
    if (d->n_nodes >= 0xFFFFFFFFull) return bad("too many entries");
    const uint64_t counts[] = {d->n_nodes, d->n_spheres, d->n_msphere, d->n_rects, d->n_boxes, d->n_triangles, d->n_xforms,
                               d->n_media, d->n_materials, d->n_textures, d->n_perlins, d->n_images, d->n_texel_bytes};
    const void* ptrs[] = {d->nodes, d->spheres, d->msphere, d->rects, d->boxes, d->triangles, d->xforms,
                          d->media, d->materials, d->textures, d->perlins, d->images, d->texels};
    for (int i = 0; i < 13; ++i)
        if (counts[i] && !ptrs[i]) return bad("non-zero count with a null table");
    for (uint64_t i = 0; i < d->n_images; ++i) {
        const RtImage& im = d->images[i];
        if (im.width == 0 || im.height == 0) return bad("empty image");
        if (im.offset > d->n_texel_bytes || (uint64_t)im.width * im.height * 3u > d->n_texel_bytes - im.offset) return bad("image outside texel blob");
    }
    for (uint64_t i = 0; i < d->n_textures; ++i) {
        const RtTexture& t = d->textures[i];
        if (t.kind > RT_T_IMAGE) return bad("texture kind");
        if (t.kind == RT_T_NOISE && t.aux >= d->n_perlins) return bad("texture -> perlin index");
        if (t.kind == RT_T_IMAGE && t.aux >= d->n_images) return bad("texture -> image index");
    }
    for (uint64_t i = 0; i < d->n_perlins; ++i)
        for (int k = 0; k < 256; ++k) {
            const RtPerlin& p = d->perlins[i];
            if ((uint32_t)p.perm_x[k] > 255u || (uint32_t)p.perm_y[k] > 255u || (uint32_t)p.perm_z[k] > 255u) return bad("perlin permutation entry");
        }
    for (uint64_t i = 0; i < d->n_materials; ++i) {
        const RtMaterial& m = d->materials[i];
        if (m.kind > RT_M_ISOTROPIC) return bad("material kind");
        const bool textured = m.kind == RT_M_LAMBERTIAN || m.kind == RT_M_DIFFUSE_LIGHT || m.kind == RT_M_ISOTROPIC;
        if (textured && m.tex >= d->n_textures) return bad("material -> texture index");
    }
    auto mat_ok = [&](uint32_t m) { return m < d->n_materials; };
    for (uint64_t i = 0; i < d->n_spheres; ++i) if (!mat_ok(d->spheres[i].mat)) return bad("sphere material");
    for (uint64_t i = 0; i < d->n_msphere; ++i) if (!mat_ok(d->msphere[i].mat)) return bad("moving sphere material");
    for (uint64_t i = 0; i < d->n_rects; ++i) if (!mat_ok(d->rects[i].mat) || d->rects[i].axis > 2u) return bad("rect material/axis");
    for (uint64_t i = 0; i < d->n_boxes; ++i) if (!mat_ok(d->boxes[i].mat)) return bad("box material");
    for (uint64_t i = 0; i < d->n_triangles; ++i) if (!mat_ok(d->triangles[i].mat)) return bad("triangle material");
    for (uint64_t i = 0; i < d->n_media; ++i) if (!mat_ok(d->media[i].mat) || d->media[i].rng_id >= 0xFFF0u) return bad("medium material/rng id");
    for (uint64_t i = 0; i < d->n_xforms; ++i) {
        const RtXform& x = d->xforms[i];
        if (x.kind > 1u || x.depth < 1u || x.depth > 4u) return bad("xform kind/depth");
        for (uint32_t k = 0; k < x.depth; ++k) if (x.anc[k] >= d->n_xforms) return bad("xform ancestor index");
        if (x.anc[x.depth - 1u] != i) return bad("xform ancestor chain does not end at itself");
        for (uint32_t k = 0; k + 1u < x.depth; ++k) if (d->xforms[x.anc[k]].depth != k + 1u) return bad("xform ancestor depth");
    }
    // entry stream: links forward and inside the stream, indices inside their tables, wrappers balanced
    const uint64_t table_n[9] = {0, d->n_spheres, d->n_msphere, d->n_rects, d->n_boxes, d->n_triangles, d->n_xforms, d->n_xforms, d->n_media};
    // Ranges nest, among themselves and with the wrappers: the walk skips from a box or medium entry to its link and
    // must arrive in the space it left, and a boundary walk must end in the space of its medium. (A stream that breaks
    // this is walked without a fault, but with the wrong ray.)
    struct Range { uint64_t end; size_t depth; };
    std::vector<uint32_t> open;        // xform stack
    std::vector<Range> ranges;         // box / medium entries we are inside, innermost last, and the wrapper depth they began at
    uint64_t medium_end = 0;           // end of the boundary range we are inside (0 = none)
    for (uint64_t i = 0; i < d->n_nodes; ++i) {
        const RtNode& n = d->nodes[i];
        const uint32_t kind = n.meta & 15u, idx = n.meta >> 4;
        while (!ranges.empty() && ranges.back().end == i) {
            if (open.size() != ranges.back().depth) return bad("the range of a box or medium entry ends inside a wrapper entered within it");
            ranges.pop_back();
        }
        if (kind > RT_N_MEDIUM) return bad("entry kind");
        if (kind != RT_N_INNER && idx >= table_n[kind]) return bad("entry -> table index");
        if (kind == RT_N_INNER && idx > RT_INNER_COVER) return bad("box entry with an unknown marker");
        if (kind == RT_N_INNER && idx == RT_INNER_COVER) {  // the walk steps from a covering box straight onto the node it covers
            if (i + 1 >= d->n_nodes || d->nodes[i + 1].meta != RT_N_INNER || d->nodes[i + 1].link != n.link)
                return bad("covering box not followed by the unmarked box entry it covers");
        }
        if (medium_end && i >= medium_end) medium_end = 0;
        if (kind == RT_N_INNER || kind == RT_N_MEDIUM) {
            if (n.link <= i || n.link > d->n_nodes) return bad("entry link");
            if (medium_end && n.link > medium_end) return bad("entry link leaves the medium boundary");
            if (!ranges.empty() && n.link > ranges.back().end) return bad("entry link leaves the range of the entry around it");
            ranges.push_back({n.link, open.size()});
        }
        if (kind == RT_N_MEDIUM) {
            if (medium_end) return bad("medium inside a medium boundary");
            if (n.link == i + 1u) return bad("medium without a boundary");  // the walk looks at the entry after it
            medium_end = n.link;
        }
        if (kind == RT_N_XF_ENTER) {
            const RtXform& x = d->xforms[idx];
            if (x.depth != open.size() + 1u) return bad("wrapper depth does not match nesting");
            for (size_t k = 0; k < open.size(); ++k) if (x.anc[k] != open[k]) return bad("wrapper ancestors do not match nesting");
            open.push_back(idx);
        } else if (kind == RT_N_XF_EXIT) {
            if (open.empty() || open.back() != idx) return bad("unbalanced wrapper exit");
            if (!ranges.empty() && open.size() - 1u < ranges.back().depth) return bad("wrapper left inside the range of an entry that began within it");
            open.pop_back();
        }
    }
    if (!open.empty()) return bad("unbalanced wrapper enter");
    for (const Range& r : ranges)
        if (r.depth != 0u) return bad("the range of a box or medium entry ends inside a wrapper entered within it");
    return RT_OK;
}

// Media that are asked before the walk instead of inside it (rtd::deferred_media, rtd::first_candidate): RT_N_MEDIUM
// entries in world space (inside no wrapper) whose boundary is one sphere, the first rtd::MAX_DEFERRED_MEDIA of them in
// stream order (box-bounded media such as cornell_smoke's were tried too: their twelve divisions per medium cost the
// shade launch as much as the walk saves). What a medium answers does not depend on when it is asked (its random
// number is keyed by the medium), and a walk spends a whole round of a thinly filled wave on such an entry (measured on
// final_scene: 11 % of the lanes for the medium inside the glass sphere), while the kernels that make rays ask with full
// waves. They leave the
// library's copy of the stream (`out` = the stream without them, links moved), and box entries that are left with
// nothing below go with them. In a scene whose boxes may not bound their contents only media that every ray reaches
// (under no box entry) are taken: its covering boxes come in pairs that must stay whole.
void strip_first_asked_media(const RtSceneDesc* d, std::vector<RtNode>& out, rtd::DeferredMedia& df) {
    df = rtd::DeferredMedia{};
    const uint64_t n = d->n_nodes;
    const bool pairs = (d->flags & RT_DESC_BOXES_MAY_NOT_BOUND) != 0;
    std::vector<uint8_t> keep((size_t)n, 1);
    uint64_t covered_until = 0;  // entries before this index lie under some box entry
    uint32_t wrappers = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const RtNode& e = d->nodes[i];
        const uint32_t kind = e.meta & 15u;
        if (kind == RT_N_XF_ENTER) ++wrappers;
        if (kind == RT_N_XF_EXIT) --wrappers;
        if (kind == RT_N_INNER) covered_until = std::max<uint64_t>(covered_until, e.link);
        if (kind != RT_N_MEDIUM) continue;
        const bool one_sphere = e.link == i + 2 && (d->nodes[i + 1].meta & 15u) == RT_N_SPHERE;
        if (wrappers == 0 && one_sphere && df.n < rtd::MAX_DEFERRED_MEDIA && (!pairs || i >= covered_until)) {
            df.medium[df.n] = e.meta >> 4;
            df.sphere[df.n] = d->nodes[i + 1].meta >> 4;
            ++df.n;
            keep[(size_t)i] = keep[(size_t)i + 1] = 0;
        }
        i = e.link - 1;  // nothing inside a boundary is looked at (validate(): no media there, links stay inside)
    }
    if (df.n == 0) return;
    // box entries with nothing left below (back to front, so that a box over emptied boxes goes too)
    if (!pairs) {
        std::vector<uint32_t> kept_after((size_t)n + 1, 0);  // kept entries at or after position i
        for (uint64_t i = n; i-- > 0;) {
            const RtNode& e = d->nodes[i];
            if (keep[(size_t)i] && (e.meta & 15u) == RT_N_INNER && kept_after[(size_t)i + 1] == kept_after[(size_t)e.link]) keep[(size_t)i] = 0;
            kept_after[(size_t)i] = kept_after[(size_t)i + 1] + keep[(size_t)i];
        }
    }
    std::vector<uint32_t> pos((size_t)n + 1, 0);  // where entry i (or the end of the stream) stands afterwards
    for (uint64_t i = 0; i < n; ++i) pos[(size_t)i + 1] = pos[(size_t)i] + keep[(size_t)i];
    out.reserve(pos[(size_t)n]);
    for (uint64_t i = 0; i < n; ++i) {
        if (!keep[(size_t)i]) continue;
        RtNode e = d->nodes[i];
        const uint32_t kind = e.meta & 15u;
        if (kind == RT_N_INNER || kind == RT_N_MEDIUM) e.link = pos[(size_t)e.link];
        out.push_back(e);
    }
}

template <class T>
int upload(RtScene* s, const T* src, uint64_t n, const T** dst) {
    *dst = nullptr;
    if (n == 0) return RT_OK;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, n * sizeof(T) + 16));  // 16 bytes of slack: the LDS staging copy reads whole 16-byte chunks
    s->allocs.push_back(p);
    HIP_TRY(hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(p);
    return RT_OK;
}

int bind_device(RtScene* s) {
    HIP_TRY(hipSetDevice(s->device));
    return RT_OK;
}

uint32_t grid_for(const RtScene* s, uint64_t n) {
    const uint64_t want = (n + rtd::BLOCK - 1) / rtd::BLOCK;
    const uint64_t cap = (uint64_t)s->n_cus * 8u;  // 8 blocks of 256 threads fill a CU's 32 wave slots
    return (uint32_t)std::max<uint64_t>(1, std::min(want, cap));
}

// The trace kernels are persistent: never more blocks than can be resident, fewer when there are few rays.
uint32_t trace_grid(const RtScene* s, uint64_t n) {
    const uint64_t want = (n + 8 * (rtd::TRACE_BLOCK / 64) - 1) / (8 * (rtd::TRACE_BLOCK / 64));
    return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(want, s->grid_extend));
}

void free_workspace(Workspace& w) {
    if (w.base) (void)hipFree(w.base);
    if (w.counts) (void)hipFree(w.counts);
    if (w.ray_total) (void)hipFree(w.ray_total);
    for (auto e : w.events) (void)hipEventDestroy(e);
    w = Workspace();
}

uint64_t shade_padding(const RtScene* s) {
    return (uint64_t)s->grid_shade * (rtd::SHADE_BLOCK / 64u) * rtd::OUT_CHUNK;
}

int ensure_workspace(RtScene* s, uint64_t n_slots) {
    Workspace& w = s->ws;
    if (!w.counts) {
        HIP_TRY(hipMalloc((void**)&w.counts, 4 * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void**)&w.ray_total, (1 + kCounterWords) * sizeof(unsigned long long)));
    }
    if (w.cap_slots >= n_slots) return RT_OK;
    if (w.base) { HIP_TRY(hipFree(w.base)); w.base = nullptr; w.cap_slots = 0; }
    // The shade kernel reserves output records in chunks and pads what a wave leaves unused, so the record and
    // hit arrays can grow past the number of paths by one chunk per shading wave.
    const uint64_t n_recs = n_slots + shade_padding(s);
    const uint64_t bytes = n_recs * (2 * sizeof(rtd::RayRec) + sizeof(rtd::HitSlot)) + n_slots * 3 * sizeof(double);
    HIP_TRY(hipMalloc((void**)&w.base, bytes));
    char* p = w.base;
    w.P.rec[0] = reinterpret_cast<rtd::RayRec*>(p); p += n_recs * sizeof(rtd::RayRec);
    w.P.rec[1] = reinterpret_cast<rtd::RayRec*>(p); p += n_recs * sizeof(rtd::RayRec);
    w.P.hit = reinterpret_cast<rtd::HitSlot*>(p); p += n_recs * sizeof(rtd::HitSlot);
    w.P.lr = reinterpret_cast<double*>(p); p += n_slots * sizeof(double);
    w.P.lg = reinterpret_cast<double*>(p); p += n_slots * sizeof(double);
    w.P.lb = reinterpret_cast<double*>(p);
    w.cap_slots = n_slots;
    return RT_OK;
}

int check_render_args(const RtScene* s, const RtCamera* cam, const RtRenderParams* p, const float* accum) {
    if (!s || !cam || !p || !accum) return rt::set_error(RT_E_INVALID, "null argument");
    if (p->width < 2 || p->height < 2) return rt::set_error(RT_E_INVALID, "width and height must be at least 2");
    if ((uint64_t)p->width * p->height >= (1ull << 31)) return rt::set_error(RT_E_INVALID, "image too large");
    if (p->sample_end < p->sample_begin) return rt::set_error(RT_E_INVALID, "sample_end < sample_begin");
    if (p->max_depth > 0xFFFFu) return rt::set_error(RT_E_INVALID, "max_depth too large");
    return RT_OK;
}

// The two launches of one bounce, shared by the render loop and rt_path_step. `profile`: bits as in kernels.h.
void launch_extend(RtScene* s, int profile, bool count_work, const RenderConst& C, uint32_t cur, uint32_t bounce, hipStream_t stream) {
    Workspace& w = s->ws;
    const dim3 eg(s->grid_extend);
    unsigned long long* wk = count_work ? w.ray_total + 1 : nullptr;
#define RT_LAUNCH_EXTEND(W, L, PR) hipLaunchKernelGGL((rtd::k_extend<W, L, PR>), eg, dim3(rtd::extend_block(L, PR)), s->lds_bytes, stream, s->S, s->plan, C, w.P, w.counts, cur, bounce, wk)
#define RT_LAUNCH_EXTEND_P(PR) do { if (count_work) { if (s->extend_lds) RT_LAUNCH_EXTEND(true, true, PR); else RT_LAUNCH_EXTEND(true, false, PR); } \
                                    else { if (s->extend_lds) RT_LAUNCH_EXTEND(false, true, PR); else RT_LAUNCH_EXTEND(false, false, PR); } } while (0)
    switch (profile) {
        case 0: RT_LAUNCH_EXTEND_P(0); break;
        case 1: RT_LAUNCH_EXTEND_P(1); break;
        case 3: RT_LAUNCH_EXTEND_P(3); break;
        case 4: RT_LAUNCH_EXTEND_P(4); break;
        case 5: RT_LAUNCH_EXTEND_P(5); break;
        case 8: RT_LAUNCH_EXTEND_P(8); break;
        case 12: RT_LAUNCH_EXTEND_P(12); break;
        default: RT_LAUNCH_EXTEND_P(7); break;
    }
#undef RT_LAUNCH_EXTEND_P
#undef RT_LAUNCH_EXTEND
}

void launch_shade(RtScene* s, const RenderConst& C, uint32_t cur, uint32_t bounce, hipStream_t stream) {
    Workspace& w = s->ws;
    const dim3 sg(std::min(s->grid_shade, (C.n_slots + rtd::SHADE_BLOCK - 1) / rtd::SHADE_BLOCK)), sb(rtd::SHADE_BLOCK);
#define RT_LAUNCH_SHADE(L, Q) hipLaunchKernelGGL((rtd::k_shade<L, Q>), sg, sb, s->lds_bytes_shade, stream, s->S, s->plan_shade, C, w.P, w.counts, cur, bounce, w.ray_total)
#define RT_LAUNCH_SHADE_L(L) do { if (s->shade_rounds == rtd::SORT_ROUNDS_WIDE) RT_LAUNCH_SHADE(L, rtd::SORT_ROUNDS_WIDE); \
                                  else if (s->shade_rounds) RT_LAUNCH_SHADE(L, rtd::SORT_ROUNDS); else RT_LAUNCH_SHADE(L, 0u); } while (0)
    if (s->shade_lds) RT_LAUNCH_SHADE_L(true); else RT_LAUNCH_SHADE_L(false);
#undef RT_LAUNCH_SHADE_L
#undef RT_LAUNCH_SHADE
}

// A device staging area of the scene, at least `bytes` long (contents are not kept when it grows).
int stage_buffer(RtScene* s, int which, size_t bytes, void** out) {
    if (s->stage_bytes[which] < bytes) {
        if (s->stage[which]) { HIP_TRY(hipFree(s->stage[which])); s->stage[which] = nullptr; s->stage_bytes[which] = 0; }
        const size_t want = std::max(bytes, (size_t)1 << 16);
        HIP_TRY(hipMalloc(&s->stage[which], want));
        s->stage_bytes[which] = want;
    }
    *out = s->stage[which];
    return RT_OK;
}

int render_on_stream(RtScene* s, const RtCamera* cam, const RtRenderParams* p, float* accum_dev, hipStream_t stream,
                     RtRenderStats* stats, uint64_t* work = nullptr) {
    if (int rc = bind_device(s)) return rc;
    const uint64_t npix = (uint64_t)p->width * p->height;
    const uint32_t total = p->sample_end - p->sample_begin;
    if (stats) std::memset(stats, 0, sizeof *stats);
    if (total == 0) return RT_OK;
    uint32_t spb = p->samples_per_batch;
    if (spb == 0) {
        // Large batches amortise the late bounces, whose launches take as long as their longest ray however few rays
        // are left: as many slots as half of the free device memory holds, up to the default target.
        uint64_t target = kDefaultSlotTarget;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess)
            target = std::min<uint64_t>(target, (free_b + s->ws.cap_slots * kBytesPerSlot) / 2 / kBytesPerSlot);
        spb = (uint32_t)std::max<uint64_t>(1, target / npix);
    }
    spb = std::min(spb, total);
    const uint64_t max_slots = (1ull << 32) - 2 - shade_padding(s);  // record indices and counts are 32-bit
    if (npix * spb > max_slots) spb = (uint32_t)(max_slots / npix);
    if (int rc = ensure_workspace(s, npix * spb)) return rc;
    Workspace& w = s->ws;

    RenderConst C{};
    C.cam = *cam;
    C.seed = p->seed;
    for (int k = 0; k < 3; ++k) C.bg[k] = p->background[k];
    C.width = p->width; C.height = p->height; C.npix = (uint32_t)npix;
    C.max_depth = p->max_depth;

    const uint32_t n_batches = (total + spb - 1) / spb;
    size_t ev_needed = stats ? 2 + (size_t)n_batches * p->max_depth * 4 : 0;
    while (w.events.size() < ev_needed) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        w.events.push_back(e);
    }
    size_t ev = 0;
    HIP_TRY(hipMemsetAsync(w.ray_total, 0, (1 + kCounterWords) * sizeof(unsigned long long), stream));
    HIP_TRY(hipMemsetAsync(w.counts, 0, 3 * sizeof(uint32_t), stream));
    if (stats) HIP_TRY(hipEventRecord(w.events[ev++], stream));
    for (uint32_t b = 0; b < n_batches; ++b) {
        const uint32_t s0 = p->sample_begin + b * spb;
        const uint32_t ns = std::min(spb, p->sample_end - s0);
        C.sample0 = s0;
        C.n_slots = (uint32_t)(npix * ns);
        const uint32_t grid = grid_for(s, C.n_slots);
        hipLaunchKernelGGL(rtd::k_generate, dim3(grid), dim3(rtd::BLOCK), 0, stream, s->S, C, w.P);
        for (uint32_t bounce = 0; bounce < p->max_depth; ++bounce) {
            const uint32_t cur = bounce & 1u;
            if (stats) HIP_TRY(hipEventRecord(w.events[ev++], stream));
            launch_extend(s, s->profile, work != nullptr, C, cur, bounce, stream);
            if (stats) HIP_TRY(hipEventRecord(w.events[ev++], stream));
            launch_shade(s, C, cur, bounce, stream);
            if (stats) HIP_TRY(hipEventRecord(w.events[ev++], stream));
        }
        hipLaunchKernelGGL(rtd::k_accumulate, dim3(grid_for(s, npix)), dim3(rtd::BLOCK), 0, stream, C, w.P, accum_dev,
                           work ? w.ray_total + 1 + kWorkWords : nullptr);
    }
    HIP_TRY(hipGetLastError());
    if (stats) {
        HIP_TRY(hipEventRecord(w.events[ev++], stream));
        HIP_TRY(hipStreamSynchronize(stream));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, w.events[0], w.events[ev - 1]));
        stats->ms_total = ms;
        size_t k = 1;
        const bool log_bounces = std::getenv("RT_BOUNCE_LOG") != nullptr;  // developer knob: the launches of every bounce on stderr
        for (uint32_t b = 0; b < n_batches; ++b)
            for (uint32_t bounce = 0; bounce < p->max_depth; ++bounce) {
                float me = 0.f, msh = 0.f;
                HIP_TRY(hipEventElapsedTime(&me, w.events[k], w.events[k + 1])); stats->ms_extend += me;
                HIP_TRY(hipEventElapsedTime(&msh, w.events[k + 1], w.events[k + 2])); stats->ms_shade += msh;
                if (log_bounces) std::fprintf(stderr, "batch %u bounce %2u extend %9.4f ms shade %9.4f ms\n", b, bounce, me, msh);
                k += 3;
            }
        unsigned long long rays = 0;
        HIP_TRY(hipMemcpy(&rays, w.ray_total, sizeof rays, hipMemcpyDeviceToHost));
        stats->rays = rays;
        stats->paths = npix * total;
        stats->batches = n_batches;
    }
    if (work) {
        HIP_TRY(hipStreamSynchronize(stream));
        unsigned long long tmp[kRenderWorkWords];
        HIP_TRY(hipMemcpy(tmp, w.ray_total + 1, sizeof tmp, hipMemcpyDeviceToHost));
        for (int k = 0; k < kRenderWorkWords; ++k) work[k] = tmp[k];
    }
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rt_scene_create(const RtSceneDesc* d, RtScene** out) {
    if (!out) return rt::set_error(RT_E_INVALID, "null output");
    *out = nullptr;
    if (int rc = validate(d)) return rc;
    // the stream the kernels walk: the caller's without the media that are asked before the walk
    RtSceneDesc own = *d;
    std::vector<RtNode> own_nodes;
    rtd::DeferredMedia deferred{};
    if (!std::getenv("RT_NO_DEFERRED_MEDIA")) strip_first_asked_media(d, own_nodes, deferred);  // developer knob: walk them in place
    if (deferred.n) { own.nodes = own_nodes.data(); own.n_nodes = own_nodes.size(); }
    d = &own;
    RtScene* s = new (std::nothrow) RtScene();
    if (!s) return rt::set_error(RT_E_NOMEM, "out of host memory");
    int rc = RT_OK;
    do {
        if (hipGetDevice(&s->device) != hipSuccess) { rc = rt::set_error(RT_E_DEVICE, "no HIP device"); break; }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, s->device) != hipSuccess) { rc = rt::set_error(RT_E_DEVICE, "hipGetDeviceProperties failed"); break; }
        s->n_cus = (uint32_t)std::max(1, prop.multiProcessorCount);
        for (uint64_t i = 0; i + 1 < d->n_nodes && !s->box_media; ++i) {  // a box of six sides alone inside wrappers?
            const uint32_t kind = d->nodes[i].meta & 15u;
            if (kind != RT_N_XF_ENTER || (i > 0 && (d->nodes[i - 1].meta & 15u) == RT_N_XF_ENTER)) continue;
            uint64_t j = i, wrappers = 0;
            while (j + 1 < d->n_nodes && (d->nodes[j].meta & 15u) == RT_N_XF_ENTER) { ++j; ++wrappers; }
            bool alone = (d->nodes[j].meta & 15u) == RT_N_BOX6 && j + wrappers < d->n_nodes;
            for (uint64_t k = 1; alone && k <= wrappers; ++k) alone = (d->nodes[j + k].meta & 15u) == RT_N_XF_EXIT;
            s->box_media = alone;
        }
        bool general_media = false;  // a medium whose boundary is neither one sphere nor one box alone inside wrappers
        for (uint64_t i = 0; i < d->n_nodes; ++i) {
            if ((d->nodes[i].meta & 15u) != RT_N_MEDIUM) continue;
            const uint64_t link = d->nodes[i].link;
            if (link == i + 2 && (d->nodes[i + 1].meta & 15u) == RT_N_SPHERE) continue;
            uint64_t j = i + 1;
            while (j + 1 < link && (d->nodes[j].meta & 15u) == RT_N_XF_ENTER) ++j;
            const bool boxed = (d->nodes[j].meta & 15u) == RT_N_BOX6 && link == i + 2 + 2 * (j - (i + 1));
            if (!boxed) general_media = true; else s->boxed_media = true;
        }
        s->general_media = general_media;
        s->profile = (general_media ? 3 : (s->box_media ? 1 : 0)) | (d->n_nodes >= 128 ? 4 : 0);
        if (general_media) s->box_media = true;  // profile 3 compiles both in
        s->per_axis = (d->flags & RT_DESC_BOXES_MAY_NOT_BOUND) != 0;
        if (s->per_axis) {
            // the per-axis variants are built without the one-step box code and the general boundary walk
            if (general_media || s->boxed_media) {
                rc = rt::set_error(RT_E_UNSUPPORTED, "RT_DESC_BOXES_MAY_NOT_BOUND on a scene with media bounded by anything but one sphere");
                break;
            }
            s->box_media = false;
            s->profile = 8 | (d->n_nodes >= 128 ? 4 : 0);
        }
        // LDS plans: a kernel runs its LDS variant when every table it reads (all but images / texels) fits
        {
            const uint32_t kBudget = 144u * 1024u;  // of the 160 KiB of a CU; one workgroup per CU
            // the shade kernel's own LDS arrays (sorting windows, when built with them) come out of its share
            // Sorting the rays of a window by what shading them involves pays when a scene has many kinds of material
            // (measured, shade ms per 1024-spp step sorted / unsorted: Weekend with 4 kinds 113 / 108, Cornell with 2
            // 118 / 109, Cornell smoke with 3 142 / 147; final scene with 7: 12 % less sorted); the sorting variant's own LDS
            // arrays come out of the kernel's share.
            {
                bool seen[5][4] = {};
                uint32_t classes = 0;
                for (uint64_t i = 0; i < d->n_materials; ++i) {
                    const RtMaterial& m = d->materials[i];
                    const bool textured = m.kind == RT_M_LAMBERTIAN || m.kind == RT_M_DIFFUSE_LIGHT || m.kind == RT_M_ISOTROPIC;
                    const uint32_t tk = textured ? d->textures[m.tex].kind : 0u;
                    if (!seen[m.kind][tk]) { seen[m.kind][tk] = true; ++classes; }
                }
#ifndef RT_SORT_MIN_CLASSES
#define RT_SORT_MIN_CLASSES 6
#endif
                // a scene with media and three kinds also gains (Cornell smoke: 886 -> 913 M samples/s)
                // with many kinds the larger window pays as well (final scene: shade 200 -> 192.5 ms per step), with few it
                // costs (Cornell smoke: 138 -> 141)
                s->shade_rounds = classes >= RT_SORT_MIN_CLASSES ? rtd::SORT_ROUNDS_WIDE : ((d->n_media > 0 && classes >= 3) ? rtd::SORT_ROUNDS : 0u);
            }
            const uint32_t kPlainBudget = kBudget - rtd::SHADE_STAGE_LDS;
            auto sort_budget = [&](uint32_t rounds) { return kPlainBudget - ((rtd::shade_sort_lds(rounds) + 1023u) & ~1023u); };
            const uint32_t kShadeBudget = sort_budget(s->shade_rounds);
            auto pad = [](uint64_t bytes) { return (bytes + 15u) & ~15ull; };
            const uint64_t prims = pad(d->n_spheres * sizeof(RtSphere)) + pad(d->n_msphere * sizeof(RtMovingSphere)) + pad(d->n_rects * sizeof(RtRect)) +
                                   pad(d->n_boxes * sizeof(RtBox6)) + pad(d->n_triangles * sizeof(RtTriangle)) + pad(d->n_xforms * sizeof(RtXform)) +
                                   pad(d->n_media * sizeof(RtMedium));
            const uint64_t ext = pad(d->n_nodes * sizeof(RtNode)) + prims;
            const uint64_t shd = prims + pad(d->n_materials * sizeof(RtMaterial)) + pad(d->n_textures * sizeof(RtTexture)) + pad(d->n_perlins * sizeof(RtPerlin));
            rtd::LdsPlan all{};
            all.spheres = (uint32_t)(d->n_spheres * sizeof(RtSphere)); all.msphere = (uint32_t)(d->n_msphere * sizeof(RtMovingSphere));
            all.rects = (uint32_t)(d->n_rects * sizeof(RtRect)); all.boxes = (uint32_t)(d->n_boxes * sizeof(RtBox6));
            all.triangles = (uint32_t)(d->n_triangles * sizeof(RtTriangle)); all.xforms = (uint32_t)(d->n_xforms * sizeof(RtXform));
            all.media = (uint32_t)(d->n_media * sizeof(RtMedium));
            if (ext <= kBudget) {
                s->plan = all;
                s->plan.nodes = (uint32_t)(d->n_nodes * sizeof(RtNode));
                s->lds_bytes = (uint32_t)ext;
            }
            if (shd <= kShadeBudget) {
                s->plan_shade = all;
                s->plan_shade.materials = (uint32_t)(d->n_materials * sizeof(RtMaterial));
                s->plan_shade.textures = (uint32_t)(d->n_textures * sizeof(RtTexture));
                s->plan_shade.perlins = (uint32_t)(d->n_perlins * sizeof(RtPerlin));
                s->lds_bytes_shade = (uint32_t)shd;
            }
            s->extend_lds = ext <= kBudget && d->n_nodes > 0;
            s->shade_lds = shd <= kShadeBudget && d->n_nodes > 0;
            // A scene that does not fit as a whole: the tables read most often go first, as many as fit
            uint32_t limit = kBudget;
            auto take = [&](uint32_t& field, uint64_t bytes, uint32_t& used) {
                if (bytes && used + pad(bytes) <= limit) { field = (uint32_t)bytes; used += (uint32_t)pad(bytes); }
            };
            if (!s->extend_lds && d->n_nodes > 0 && pad(d->n_nodes * sizeof(RtNode)) <= kBudget) {  // not without the entries
                uint32_t used = 0;
                take(s->plan.nodes, d->n_nodes * sizeof(RtNode), used);
                take(s->plan.xforms, d->n_xforms * sizeof(RtXform), used);
                take(s->plan.media, d->n_media * sizeof(RtMedium), used);
                take(s->plan.spheres, d->n_spheres * sizeof(RtSphere), used);
                take(s->plan.msphere, d->n_msphere * sizeof(RtMovingSphere), used);
                take(s->plan.rects, d->n_rects * sizeof(RtRect), used);
                take(s->plan.boxes, d->n_boxes * sizeof(RtBox6), used);
                take(s->plan.triangles, d->n_triangles * sizeof(RtTriangle), used);
                s->lds_bytes = used;
            }
            if (!s->shade_lds && d->n_nodes > 0) {
                uint32_t used = 0;
                limit = kShadeBudget;
                take(s->plan_shade.materials, d->n_materials * sizeof(RtMaterial), used);
                take(s->plan_shade.textures, d->n_textures * sizeof(RtTexture), used);
                take(s->plan_shade.perlins, d->n_perlins * sizeof(RtPerlin), used);
                take(s->plan_shade.xforms, d->n_xforms * sizeof(RtXform), used);
                take(s->plan_shade.media, d->n_media * sizeof(RtMedium), used);
                take(s->plan_shade.spheres, d->n_spheres * sizeof(RtSphere), used);
                take(s->plan_shade.msphere, d->n_msphere * sizeof(RtMovingSphere), used);
                take(s->plan_shade.rects, d->n_rects * sizeof(RtRect), used);
                take(s->plan_shade.boxes, d->n_boxes * sizeof(RtBox6), used);
                take(s->plan_shade.triangles, d->n_triangles * sizeof(RtTriangle), used);
                s->lds_bytes_shade = used;
            }
            const int big = (int)kBudget;
#define RT_BIG_LDS(K) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, big)
#define RT_BIG_LDS_P(PR) RT_BIG_LDS((rtd::k_extend<false, false, PR>)); RT_BIG_LDS((rtd::k_extend<false, true, PR>)); \
                         RT_BIG_LDS((rtd::k_extend<true, false, PR>)); RT_BIG_LDS((rtd::k_extend<true, true, PR>))
            RT_BIG_LDS_P(0); RT_BIG_LDS_P(1); RT_BIG_LDS_P(3); RT_BIG_LDS_P(4); RT_BIG_LDS_P(5); RT_BIG_LDS_P(7); RT_BIG_LDS_P(8); RT_BIG_LDS_P(12);
#undef RT_BIG_LDS_P
#define RT_SHADE_LDS(L, Q) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rtd::k_shade<L, Q>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sort_budget(Q))
            RT_SHADE_LDS(false, 0u); RT_SHADE_LDS(false, rtd::SORT_ROUNDS); RT_SHADE_LDS(false, rtd::SORT_ROUNDS_WIDE);
            RT_SHADE_LDS(true, 0u); RT_SHADE_LDS(true, rtd::SORT_ROUNDS); RT_SHADE_LDS(true, rtd::SORT_ROUNDS_WIDE);
#undef RT_SHADE_LDS
#define RT_BIG_LDS_T(W, L) RT_BIG_LDS((rtd::k_trace<W, L, false>)); RT_BIG_LDS((rtd::k_trace<W, L, true>))
            RT_BIG_LDS_T(false, false); RT_BIG_LDS_T(false, true); RT_BIG_LDS_T(true, false); RT_BIG_LDS_T(true, true);
#undef RT_BIG_LDS_T
#undef RT_BIG_LDS
            (void)hipGetLastError();  // a refused attribute must not surface as the error of a later launch
        }
        int per_cu = 0;
        hipError_t oe = hipErrorUnknown;
#define RT_OCC(L, PR) oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, rtd::k_extend<false, L, PR>, rtd::extend_block(L, PR), s->lds_bytes)
#define RT_OCC_P(PR) do { if (s->extend_lds) RT_OCC(true, PR); else RT_OCC(false, PR); } while (0)
        switch (s->profile) {
            case 0: RT_OCC_P(0); break;
            case 1: RT_OCC_P(1); break;
            case 3: RT_OCC_P(3); break;
            case 4: RT_OCC_P(4); break;
            case 5: RT_OCC_P(5); break;
            case 8: RT_OCC_P(8); break;
            case 12: RT_OCC_P(12); break;
            default: RT_OCC_P(7); break;
        }
#undef RT_OCC_P
#undef RT_OCC
        if (oe != hipSuccess || per_cu < 1) per_cu = 1;
        s->grid_extend = s->n_cus * (uint32_t)per_cu;
        if (const char* e = std::getenv("RT_EXTEND_CUS")) s->grid_extend = (uint32_t)std::max(1, std::atoi(e)) * (uint32_t)per_cu;  // developer knob
        int shade_per_cu = 0;
#define RT_SHADE_OCC(L, Q) oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&shade_per_cu, rtd::k_shade<L, Q>, rtd::SHADE_BLOCK, s->lds_bytes_shade)
#define RT_SHADE_OCC_L(L) do { if (s->shade_rounds == rtd::SORT_ROUNDS_WIDE) RT_SHADE_OCC(L, rtd::SORT_ROUNDS_WIDE); \
                               else if (s->shade_rounds) RT_SHADE_OCC(L, rtd::SORT_ROUNDS); else RT_SHADE_OCC(L, 0u); } while (0)
        if (s->shade_lds) RT_SHADE_OCC_L(true); else RT_SHADE_OCC_L(false);
#undef RT_SHADE_OCC_L
#undef RT_SHADE_OCC
        if (oe != hipSuccess || shade_per_cu < 1) shade_per_cu = 1;
#ifndef RT_SHADE_GRID_MULT
#define RT_SHADE_GRID_MULT 1
#endif
        s->grid_shade = s->n_cus * (uint32_t)shade_per_cu * RT_SHADE_GRID_MULT;
        if (const char* e = std::getenv("RT_SHADE_CUS")) s->grid_shade = (uint32_t)std::max(1, std::atoi(e)) * (uint32_t)shade_per_cu;  // developer knob
        if ((rc = upload(s, d->nodes, d->n_nodes, &s->S.nodes))) break;
        if ((rc = upload(s, d->spheres, d->n_spheres, &s->S.spheres))) break;
        if ((rc = upload(s, d->msphere, d->n_msphere, &s->S.msphere))) break;
        if ((rc = upload(s, d->rects, d->n_rects, &s->S.rects))) break;
        if ((rc = upload(s, d->boxes, d->n_boxes, &s->S.boxes))) break;
        if ((rc = upload(s, d->triangles, d->n_triangles, &s->S.triangles))) break;
        if ((rc = upload(s, d->xforms, d->n_xforms, &s->S.xforms))) break;
        if ((rc = upload(s, d->media, d->n_media, &s->S.media))) break;
        if ((rc = upload(s, d->materials, d->n_materials, &s->S.materials))) break;
        if ((rc = upload(s, d->textures, d->n_textures, &s->S.textures))) break;
        if ((rc = upload(s, d->perlins, d->n_perlins, &s->S.perlins))) break;
        if ((rc = upload(s, d->images, d->n_images, &s->S.images))) break;
        if ((rc = upload(s, d->texels, d->n_texel_bytes, &s->S.texels))) break;
        s->S.n_nodes = (uint32_t)d->n_nodes;
        s->S.deferred = deferred;
        // the ray times for which every moving sphere is within its own motion interval (RT_INNER_CARRIED / _COVER)
        s->S.carried_t0 = -rtd::D_INF; s->S.carried_t1 = rtd::D_INF;
        for (uint64_t i = 0; i < d->n_msphere; ++i) {
            const RtMovingSphere& m = d->msphere[i];
            if (m.c0[0] == m.c1[0] && m.c0[1] == m.c1[1] && m.c0[2] == m.c1[2]) continue;
            s->S.carried_t0 = std::fmax(s->S.carried_t0, std::fmin(m.t0, m.t1));
            s->S.carried_t1 = std::fmin(s->S.carried_t1, std::fmax(m.t0, m.t1));
        }
    } while (0);
    if (rc) { rt_scene_destroy(s); return rc; }
    *out = s;
    return RT_OK;
}

void rt_scene_destroy(RtScene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (void* p : s->allocs) (void)hipFree(p);
    for (void* p : s->stage) if (p) (void)hipFree(p);
    for (hipEvent_t e : s->ev_trace) if (e) (void)hipEventDestroy(e);
    free_workspace(s->ws);
    delete s;
}

int rt_trace_closest_device(RtScene* s, const RtRay* rays_dev, uint64_t n, uint64_t seed, RtHit* out_dev, void* stream) {
    if (!s || (n && (!rays_dev || !out_dev))) return rt::set_error(RT_E_INVALID, "null argument");
    if (n == 0) return RT_OK;
    if (int rc = bind_device(s)) return rc;
    if (n >= (1ull << 32)) return rt::set_error(RT_E_INVALID, "too many rays in one call");
    if (int rc = ensure_workspace(s, 0)) return rc;
    HIP_TRY(hipMemsetAsync(s->ws.counts + 3, 0, sizeof(uint32_t), (hipStream_t)stream));
#define RT_LAUNCH_TRACE_DEV(L, A) hipLaunchKernelGGL((rtd::k_trace<false, L, A>), dim3(trace_grid(s, n)), dim3(rtd::TRACE_BLOCK), s->lds_bytes, (hipStream_t)stream, s->S, s->plan, \
                                                     rays_dev, n, seed, out_dev, (RtHitRecord*)nullptr, s->ws.counts + 3, (unsigned long long*)nullptr)
    if (s->extend_lds) { if (s->per_axis) RT_LAUNCH_TRACE_DEV(true, true); else RT_LAUNCH_TRACE_DEV(true, false); }
    else { if (s->per_axis) RT_LAUNCH_TRACE_DEV(false, true); else RT_LAUNCH_TRACE_DEV(false, false); }
#undef RT_LAUNCH_TRACE_DEV
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

}  // extern "C"

namespace {

// Shared body of rt_trace_closest / rt_trace_records / rt_trace_work: host buffers in and out, staged through the
// scene's device staging areas (allocated once, grown on demand).
int trace_host(RtScene* s, const RtRay* rays, uint64_t n, uint64_t seed, RtHit* hits, RtHitRecord* recs, uint64_t* work,
               RtTraceStats* stats) {
    if (!s || (n && !rays) || (n && !hits && !recs)) return rt::set_error(RT_E_INVALID, "null argument");
    if (stats) stats->ms_kernel = 0.0;
    if (work) std::memset(work, 0, kWorkWords * sizeof(uint64_t));
    if (n == 0) return RT_OK;
    if (n >= (1ull << 32)) return rt::set_error(RT_E_INVALID, "too many rays in one call");
    if (int rc = bind_device(s)) return rc;
    const size_t out_bytes = n * (recs ? sizeof(RtHitRecord) : sizeof(RtHit));
    void *d_rays_v = nullptr, *d_out = nullptr, *d_work_v = nullptr;
    if (int rc = stage_buffer(s, 0, n * sizeof(RtRay), &d_rays_v)) return rc;
    if (int rc = stage_buffer(s, 1, out_bytes, &d_out)) return rc;
    if (int rc = stage_buffer(s, 2, kCounterWords * sizeof(unsigned long long), &d_work_v)) return rc;
    RtRay* d_rays = static_cast<RtRay*>(d_rays_v);
    unsigned long long* d_work = work ? static_cast<unsigned long long*>(d_work_v) : nullptr;
    for (hipEvent_t& e : s->ev_trace)
        if (!e) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipMemcpy(d_rays, rays, n * sizeof(RtRay), hipMemcpyHostToDevice));
    if (work) HIP_TRY(hipMemset(d_work, 0, kCounterWords * sizeof(unsigned long long)));
    if (int rc = ensure_workspace(s, 0)) return rc;
    uint32_t* cursor = s->ws.counts + 3;
    HIP_TRY(hipMemsetAsync(cursor, 0, sizeof(uint32_t), nullptr));
    const dim3 grid(trace_grid(s, n)), block(rtd::TRACE_BLOCK);
    HIP_TRY(hipEventRecord(s->ev_trace[0], nullptr));
    RtHit* oh = recs ? nullptr : (RtHit*)d_out;
    RtHitRecord* orc = recs ? (RtHitRecord*)d_out : nullptr;
#define RT_LAUNCH_TRACE(W, L, A) hipLaunchKernelGGL((rtd::k_trace<W, L, A>), grid, block, s->lds_bytes, nullptr, s->S, s->plan, d_rays, n, seed, oh, orc, cursor, d_work)
    const int variant = (s->per_axis ? 4 : 0) | (work ? 2 : 0) | (s->extend_lds ? 1 : 0);
    switch (variant) {
        case 0: RT_LAUNCH_TRACE(false, false, false); break;
        case 1: RT_LAUNCH_TRACE(false, true, false); break;
        case 2: RT_LAUNCH_TRACE(true, false, false); break;
        case 3: RT_LAUNCH_TRACE(true, true, false); break;
        case 4: RT_LAUNCH_TRACE(false, false, true); break;
        case 5: RT_LAUNCH_TRACE(false, true, true); break;
        case 6: RT_LAUNCH_TRACE(true, false, true); break;
        default: RT_LAUNCH_TRACE(true, true, true); break;
    }
#undef RT_LAUNCH_TRACE
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev_trace[1], nullptr));
    HIP_TRY(hipDeviceSynchronize());
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev_trace[0], s->ev_trace[1]));
    if (stats) stats->ms_kernel = ms;
    HIP_TRY(hipMemcpy(recs ? (void*)recs : (void*)hits, d_out, out_bytes, hipMemcpyDeviceToHost));
    if (work) {
        unsigned long long tmp[kWorkWords];
        HIP_TRY(hipMemcpy(tmp, d_work, sizeof tmp, hipMemcpyDeviceToHost));
        for (int k = 0; k < kWorkWords; ++k) work[k] = tmp[k];
    }
    return RT_OK;
}

}  // namespace

extern "C" {

int rt_trace_closest(RtScene* s, const RtRay* rays, uint64_t n, uint64_t seed, RtHit* out, RtTraceStats* stats) {
    return trace_host(s, rays, n, seed, out, nullptr, nullptr, stats);
}

int rt_trace_records(RtScene* s, const RtRay* rays, uint64_t n, uint64_t seed, RtHitRecord* out) {
    return trace_host(s, rays, n, seed, nullptr, out, nullptr, nullptr);
}

int rt_trace_work(RtScene* s, const RtRay* rays, uint64_t n, uint64_t seed, RtHit* out, uint64_t* work) {
    if (!work) return rt::set_error(RT_E_INVALID, "null argument");
    return trace_host(s, rays, n, seed, out, nullptr, work, nullptr);
}

int rt_scene_profile(const RtScene* s) {
    if (!s) return rt::set_error(RT_E_INVALID, "null argument");
    return s->profile;
}

int rt_path_step(RtScene* s, const RtRenderParams* p, int profile, const RtRay* rays, uint64_t n, RtPathStep* out, uint64_t* work) {
    if (!s || !p || (n && (!rays || !out))) return rt::set_error(RT_E_INVALID, "null argument");
    if (work) std::memset(work, 0, kWorkWords * sizeof(uint64_t));
    if (profile == RT_PROFILE_OWN) profile = s->profile & ~8;
    if (profile != 0 && profile != 1 && profile != 3 && profile != 4 && profile != 5 && profile != 7)
        return rt::set_error(RT_E_INVALID, "profile must be one of 0, 1, 3, 4, 5, 7");
    if (s->per_axis) {
        if (profile & 3) return rt::set_error(RT_E_INVALID, "a scene whose boxes may not bound their contents runs profile 0 or 4 only");
        profile |= 8;
    }
    if (s->general_media && !(profile & 2)) return rt::set_error(RT_E_INVALID, "the scene's media need the general boundary walk (profile bit 2)");
    if (s->boxed_media && !(profile & 3)) return rt::set_error(RT_E_INVALID, "the scene's box-bounded media need profile bit 1 or 2");
    if (p->width < 2 || p->height < 2 || p->sample_end < p->sample_begin || p->max_depth > 0xFFFFu)
        return rt::set_error(RT_E_INVALID, "bad render parameters");
    if (n == 0) return RT_OK;
    const uint64_t npix = (uint64_t)p->width * p->height;
    const uint64_t n_slots = npix * (p->sample_end - p->sample_begin);
    if (n_slots == 0 || n_slots + shade_padding(s) >= (1ull << 32) - 2) return rt::set_error(RT_E_INVALID, "sample range empty or too large");
    // group the rays by bounce; within a group (pixel, sample) must be unique: a slot holds one path
    std::vector<uint32_t> order;
    order.reserve(n);
    std::vector<std::pair<uint32_t, uint64_t>> groups;  // (bounce, rays)
    {
        std::vector<std::pair<uint64_t, uint32_t>> keyed((size_t)n);  // (bounce << 32 | slot, ray)
        for (uint64_t i = 0; i < n; ++i) {
            const RtRay& q = rays[i];
            if (q.tmin != 0.0001 || q.tmax != rtd::D_INF) return rt::set_error(RT_E_INVALID, "rays must have tmin = 0.0001 and tmax = +inf");
            if (q.pixel >= npix || q.sample < p->sample_begin || q.sample >= p->sample_end || q.bounce >= p->max_depth)
                return rt::set_error(RT_E_INVALID, "ray key outside the render parameters");
            keyed[(size_t)i] = {((uint64_t)q.bounce << 32) | ((uint64_t)(q.sample - p->sample_begin) * npix + q.pixel), (uint32_t)i};
        }
        std::sort(keyed.begin(), keyed.end());
        for (size_t i = 0; i < keyed.size(); ++i) {
            if (i && keyed[i].first == keyed[i - 1].first) return rt::set_error(RT_E_INVALID, "two rays share (pixel, sample, bounce)");
            const uint32_t bounce = (uint32_t)(keyed[i].first >> 32);
            if (groups.empty() || groups.back().first != bounce) groups.push_back({bounce, 0});
            ++groups.back().second;
            order.push_back(keyed[i].second);
        }
    }
    if (int rc = bind_device(s)) return rc;
    if (int rc = ensure_workspace(s, n_slots)) return rc;
    void *d_rays_v = nullptr, *d_out_v = nullptr, *d_idx_v = nullptr;
    if (int rc = stage_buffer(s, 0, n * sizeof(RtRay), &d_rays_v)) return rc;
    if (int rc = stage_buffer(s, 1, n * sizeof(RtPathStep), &d_out_v)) return rc;
    if (int rc = stage_buffer(s, 2, (n + n_slots) * sizeof(uint32_t), &d_idx_v)) return rc;
    const RtRay* d_rays = static_cast<const RtRay*>(d_rays_v);
    RtPathStep* d_out = static_cast<RtPathStep*>(d_out_v);
    uint32_t* d_idx = static_cast<uint32_t*>(d_idx_v);
    uint32_t* d_slot_to_ray = d_idx + n;
    Workspace& w = s->ws;
    hipStream_t stream = nullptr;
    HIP_TRY(hipMemcpy(d_rays_v, rays, n * sizeof(RtRay), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_idx, order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(w.ray_total, 0, (1 + kCounterWords) * sizeof(unsigned long long), stream));

    RenderConst C{};
    C.seed = p->seed;
    for (int k = 0; k < 3; ++k) C.bg[k] = p->background[k];
    C.width = p->width; C.height = p->height; C.npix = (uint32_t)npix;
    C.max_depth = p->max_depth;
    C.sample0 = p->sample_begin;
    uint64_t at = 0;
    for (const auto& g : groups) {
        const uint32_t bounce = g.first, ng = (uint32_t)g.second;
        const uint32_t cur = bounce & 1u;
        C.n_slots = ng;  // the length of the record array of bounce 0; later bounces read it from counts[cur]
        const uint32_t counts_host[3] = {cur == 0u ? ng : 0u, cur == 1u ? ng : 0u, 0u};
        HIP_TRY(hipMemcpyAsync(w.counts, counts_host, sizeof counts_host, hipMemcpyHostToDevice, stream));
        const uint32_t grid = grid_for(s, ng);
        hipLaunchKernelGGL(rtd::k_step_pack, dim3(grid), dim3(rtd::BLOCK), 0, stream, s->S, p->seed, d_rays, d_idx + at, ng, (uint32_t)npix, p->sample_begin, w.P, cur, d_slot_to_ray);
        launch_extend(s, profile, work != nullptr, C, cur, bounce, stream);
        launch_shade(s, C, cur, bounce, stream);
        hipLaunchKernelGGL(rtd::k_step_unpack_hits, dim3(grid), dim3(rtd::BLOCK), 0, stream, s->S, d_idx + at, ng, w.P, cur, d_out);
        hipLaunchKernelGGL(rtd::k_step_unpack_survivors, dim3(grid), dim3(rtd::BLOCK), 0, stream, w.P, cur ^ 1u, w.counts, d_slot_to_ray, d_out);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));  // counts_host leaves scope; groups are few
        at += ng;
    }
    HIP_TRY(hipMemcpy(out, d_out, n * sizeof(RtPathStep), hipMemcpyDeviceToHost));
    if (work) {
        unsigned long long tmp[kWorkWords];
        HIP_TRY(hipMemcpy(tmp, w.ray_total + 1, sizeof tmp, hipMemcpyDeviceToHost));
        for (int k = 0; k < kWorkWords; ++k) work[k] = tmp[k];
    }
    return RT_OK;
}

int rt_render_device(RtScene* s, const RtCamera* cam, const RtRenderParams* p, float* accum_dev, void* stream, RtRenderStats* stats) {
    if (int rc = check_render_args(s, cam, p, accum_dev)) return rc;
    return render_on_stream(s, cam, p, accum_dev, (hipStream_t)stream, stats);
}

int rt_render_work(RtScene* s, const RtCamera* cam, const RtRenderParams* p, float* accum_dev, void* stream, RtRenderStats* stats,
                   uint64_t* work) {
    if (!work) return rt::set_error(RT_E_INVALID, "null argument");
    if (int rc = check_render_args(s, cam, p, accum_dev)) return rc;
    return render_on_stream(s, cam, p, accum_dev, (hipStream_t)stream, stats, work);
}

int rt_render(RtScene* s, const RtCamera* cam, const RtRenderParams* p, float* accum_rgb, RtRenderStats* stats) {
    if (int rc = check_render_args(s, cam, p, accum_rgb)) return rc;
    if (int rc = bind_device(s)) return rc;
    const size_t bytes = (size_t)p->width * p->height * 3 * sizeof(float);
    void* d = nullptr;
    if (int rc = stage_buffer(s, 0, bytes, &d)) return rc;
    HIP_TRY(hipMemcpy(d, accum_rgb, bytes, hipMemcpyHostToDevice));
    if (int r = render_on_stream(s, cam, p, static_cast<float*>(d), nullptr, stats)) return r;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(accum_rgb, d, bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_resolve_u8_device(const float* accum_dev, uint32_t width, uint32_t height, uint64_t spp, uint8_t* out_dev, void* stream) {
    if (!accum_dev || !out_dev || spp == 0) return rt::set_error(RT_E_INVALID, "null argument or spp == 0");
    const uint64_t n = (uint64_t)width * height * 3u;
    if (n == 0) return RT_OK;
    const double scale = 1.0 / (double)spp;  // src/main.rs:216
    const uint32_t grid = (uint32_t)std::min<uint64_t>((n + rtd::BLOCK - 1) / rtd::BLOCK, 2048);
    hipLaunchKernelGGL(rtd::k_resolve, dim3(grid), dim3(rtd::BLOCK), 0, (hipStream_t)stream, accum_dev, n, scale, out_dev);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

int rt_resolve_u8(const float* accum_rgb, uint32_t width, uint32_t height, uint64_t spp, uint8_t* rgb_out) {
    if (!accum_rgb || !rgb_out || spp == 0) return rt::set_error(RT_E_INVALID, "null argument or spp == 0");
    const size_t n = (size_t)width * height * 3u;
    if (n == 0) return RT_OK;
    float* d_in = nullptr;
    uint8_t* d_out = nullptr;
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc((void**)&d_in, n * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&d_out, n));
        HIP_TRY(hipMemcpy(d_in, accum_rgb, n * sizeof(float), hipMemcpyHostToDevice));
        if (int r = rt_resolve_u8_device(d_in, width, height, spp, d_out, nullptr)) return r;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(rgb_out, d_out, n, hipMemcpyDeviceToHost));
        return RT_OK;
    };
    const int rc = body();
    if (d_in) (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

}  // extern "C"
