// srt_hip.hip -- HIP kernels (gfx950) and the extern "C" ABI of include/srt.h.
//
// Path replaced: sendRaysAndIntersectPointsColors -> rayIntersection -> boundingBoxIntersection /
// intersectRayAabbNoOrigin -> rayTriangleIntersection -> softShadow -> shadowIntersection +
// phongIllumination (/root/reference/simple_raytracer.cpp:505-525, 405-457, 296-317, 252-293, 42-75,
// 348-401, 321-342, 144-200).  No CPU fallback exists in this library.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <map>
#include <memory>
#include <new>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/srt.h"
#include "srt_device.h"

using namespace srt;

#include "srt_kernels.h"
#include "srt_packet.h"

// =================================================================================================
// Host side of the ABI
// =================================================================================================
static thread_local int g_last_hip = 0;
#define HIP_TRY(expr)                                                  \
    do {                                                               \
        hipError_t e_ = (expr);                                        \
        if (e_ != hipSuccess) { g_last_hip = (int)e_; return SRT_ERR_DEVICE; } \
    } while (0)

// No C++ exception may cross the extern "C" boundary ("never throws"): every entry point that allocates
// or spawns threads runs its body through guarded().
template <typename F>
static int guarded(F&& body) noexcept {
    try { return body(); }
    catch (const std::bad_alloc&) { return SRT_ERR_OOM; }
    catch (...) { return SRT_ERR_DEVICE; }
}
// Test hook (srt_debug_fail_host_allocs): the next n guarded host allocations throw std::bad_alloc.
static std::atomic<int> g_fail_allocs{0};
static inline void alloc_gate() {
    int n = g_fail_allocs.load(std::memory_order_relaxed);
    while (n > 0 && !g_fail_allocs.compare_exchange_weak(n, n - 1)) {}
    if (n > 0) throw std::bad_alloc();
}

constexpr double PACKET_OVERLAP_THRESHOLD = 150.;     // expected slab tests per ray from which primary rays use the packet walk (measured: DESIGN.md s5)
constexpr int RING = 64;         // HIP-event triples kept for per-kernel timing between two srt_sync calls

// The device records of a scene: owned by the handle that uploaded them and by every handle made from it with srt_scene_share.
struct SceneRecords {
    int device = 0;
    std::vector<void*> allocs;
    std::vector<uint64_t> tex_off; std::vector<uint32_t> tex_w, tex_h;      // the uploaded texture table (host copy) ...
    uint64_t tex_bytes = 0, tex_hash = 0;                                    // ... and the size and content hash of the uploaded images
    double overlap = 0.;             // expected slab tests per ray (surface-area estimate, see scene_create_impl)
    bool prefer_packet = false;      // hierarchy of heavily overlapping boxes: primary rays take the packet walk too
    bool int_shin = false;           // every object's shininess is an integer in [1, 64]: the shading kernel without the general pow
    // host copies of the static topology (srt_scene_update_frame checks counts against them and re-estimates the overlap from a frame's boxes)
    std::vector<int2> h_ranges; std::vector<int32_t> h_tri_first, h_leaf;
    // f1, device half: node -> DevWide index (static), this frame's inputs, the per-triangle attributes in source order
    int32_t* d_widx = nullptr; float4* d_src_points = nullptr; uint32_t* d_order = nullptr; float* d_box_min = nullptr; float* d_box_max = nullptr;
    float* d_src_tc = nullptr; float* d_src_nrm = nullptr; int32_t* d_src_tex = nullptr; bool have_source = false;
    ~SceneRecords() { (void)hipSetDevice(device); for (void* d : allocs) (void)hipFree(d); }
};

struct srt_scene {
    int device = 0;
    DevScene dev{};
    std::shared_ptr<SceneRecords> rec;
    uint64_t bytes = 0;
    // workspace
    int32_t* ws_hit = nullptr; float* ws_t = nullptr; size_t ws_pixels = 0;
    float* ws_lin = nullptr; uint8_t* ws_rgb8 = nullptr; size_t ws_out_pixels = 0;
    float* d_lights = nullptr; float* h_lights = nullptr; uint32_t lights_cap = 0, lights_valid = 0;
    unsigned long long* d_counters = nullptr; unsigned long long* h_counters = nullptr;   // device: two sets used alternately
    unsigned long long* d_ctr_last = nullptr;     // set written by the most recent render
    uint64_t render_seq = 0;
    bool ctr_dirty = false;                       // a render returned an error after its first launch
    char pipeline[96] = "";                       // kernels of the last render, in launch order
    uint32_t n_textures = 0; bool has_tex = false;
    void* stage = nullptr; size_t stage_bytes = 0; hipEvent_t staged = nullptr;      // pinned staging of srt_scene_update
    hipStream_t stream = nullptr;                 // the scene's own stream (srt_render, srt_render_async, srt_scene_update with stream NULL)
    unsigned long long* ws_shadow = nullptr; size_t ws_shadow_words = 0;
    uint32_t* ws_qlist = nullptr; uint32_t* d_qcount = nullptr; uint32_t qcap = 0;      // quadrants with hits: 64 shard lists of qcap entries, their counters
    float* ws_acc = nullptr; float* ws_sub = nullptr; int32_t* ws_sub_hit = nullptr; float* ws_sub_t = nullptr; size_t ws_acc_pixels = 0;
    int n_cu = 256;
    hipEvent_t ev[RING][4] = {};     // start, closest-hit done, shadow done, shade done
    uint32_t ring_count = 0;         // renders since the last srt_sync
    hipEvent_t last_done = nullptr;  // ev[..][2] of the most recent render
    hipStream_t last_stream = nullptr;
    bool pending = false;
    srt_stats last{};
};

template <typename T>
static int upload(srt_scene* s, const T* host, size_t n, const T** out) {
    void* d = nullptr;
    size_t bytes = sizeof(T) * (n ? n : 1);
    HIP_TRY(hipMalloc(&d, bytes));
    s->rec->allocs.push_back(d);
    if (n) HIP_TRY(hipMemcpy(d, host, sizeof(T) * n, hipMemcpyHostToDevice));
    s->bytes += bytes;
    *out = (const T*)d;
    return SRT_OK;
}

extern "C" {

uint32_t srt_abi_version(void) { return SRT_ABI_VERSION; }
int srt_last_hip_error(void) { return g_last_hip; }

const char* srt_strerror(int code) {
    switch (code) {
    case SRT_OK: return "ok";
    case SRT_ERR_ARG: return "invalid argument";
    case SRT_ERR_LAYOUT: return "scene arrays violate the layout contract of include/srt.h";
    case SRT_ERR_DEVICE: return "HIP runtime error (see srt_last_hip_error)";
    case SRT_ERR_NO_GPU: return "no HIP device: this library has no CPU fallback";
    case SRT_ERR_TEXTURE: return "triangle references a texture that does not exist";
    case SRT_ERR_LIMIT: return "size exceeds an implementation limit";
    case SRT_ERR_OOM: return "out of host memory";
    default: return "unknown error";
    }
}

void srt_params_default(srt_params* p, uint32_t width, uint32_t height) {
    std::memset(p, 0, sizeof(*p));
    p->width = width; p->height = height;
    p->block_rows = height; p->block_first = 0; p->block_stride = 1; p->block_cols = 0;
    p->focal = 400.0f;                 // simple_raytracer.cpp:506
    p->n_lights = 1;                   // :445
    p->light_pos = nullptr;
    p->ray_matrix = nullptr;           // the reference's frame: scene in camera space, rays from the origin
    p->shadow_div = 5.0f;              // :369
    p->reinhard = 0.5f;                // :391
    p->gamma = 1.1f;                   // :396
    p->background[0] = 173; p->background[1] = 216; p->background[2] = 230;   // :476
    p->spp = 1; p->flags = 0;
}

void srt_light_staircase(const float base[3], uint32_t n, float* out) {
    float L[3] = { base[0], base[1], base[2] };          // softShadow:363
    for (uint32_t i = 0; i < n; i++) {
        out[i * 3] = L[0]; out[i * 3 + 1] = L[1]; out[i * 3 + 2] = L[2];
        L[i % 3] += 3.0f;                                 // :372-382
    }
}

uint32_t srt_cols_owned(const srt_params* p) {
    if (!p || !p->block_stride) return 0;
    if (!p->block_cols) return p->width;
    const uint32_t n_bx = (p->width + p->block_cols - 1) / p->block_cols;
    return (n_bx + p->block_stride - 1) / p->block_stride * p->block_cols;      // padded: every block row has the same local width
}

// pixels of the image a call with these params renders (padding of a tile deal excluded)
static uint64_t pixels_owned(const srt_params* p) {
    if (!p->block_cols) return (uint64_t)p->width * srt_rows_owned(p);
    const uint32_t n_bx = (p->width + p->block_cols - 1) / p->block_cols, n_by = (p->height + p->block_rows - 1) / p->block_rows;
    uint64_t n = 0;
    for (uint32_t by = 0; by < n_by; by++) {
        const uint32_t h = (by + 1) * p->block_rows <= p->height ? p->block_rows : p->height - by * p->block_rows;
        for (uint32_t bx = (p->block_first + p->block_stride - by % p->block_stride) % p->block_stride; bx < n_bx; bx += p->block_stride) {
            const uint32_t w = (bx + 1) * p->block_cols <= p->width ? p->block_cols : p->width - bx * p->block_cols;
            n += (uint64_t)w * h;
        }
    }
    return n;
}

uint32_t srt_rows_owned(const srt_params* p) {
    if (!p || !p->block_rows || !p->block_stride) return 0;
    if (p->block_cols) return p->block_first < p->block_stride ? p->height : 0;      // tiles dealt in two dimensions: tiles in every row
    const uint32_t nblocks = (p->height + p->block_rows - 1) / p->block_rows;
    uint32_t rows = 0;
    for (uint32_t b = p->block_first; b < nblocks; b += p->block_stride) {
        const uint32_t y0 = b * p->block_rows;
        uint32_t y1 = y0 + p->block_rows;
        if (y1 > p->height) y1 = p->height;
        rows += y1 - y0;
    }
    return rows;
}

// Wait for the work of earlier renders before their buffers are reused or freed.
static hipError_t wait_idle(srt_scene* s) {
    if (!s->pending) return hipSuccess;
    return s->last_done ? hipEventSynchronize(s->last_done) : hipStreamSynchronize(s->last_stream);
}

int srt_scene_destroy(srt_scene* s) {
    if (!s) return SRT_ERR_ARG;
    (void)hipSetDevice(s->device);
    (void)wait_idle(s);
    s->rec.reset();                      // frees the records with their last handle
    if (s->ws_hit) (void)hipFree(s->ws_hit);
    if (s->ws_t) (void)hipFree(s->ws_t);
    if (s->ws_lin) (void)hipFree(s->ws_lin);
    if (s->ws_rgb8) (void)hipFree(s->ws_rgb8);
    if (s->ws_shadow) (void)hipFree(s->ws_shadow);
    if (s->ws_qlist) (void)hipFree(s->ws_qlist);
    if (s->d_qcount) (void)hipFree(s->d_qcount);
    if (s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
    if (s->stage) (void)hipHostFree(s->stage);
    if (s->staged) (void)hipEventDestroy(s->staged);
    if (s->ws_acc) (void)hipFree(s->ws_acc);
    if (s->ws_sub) (void)hipFree(s->ws_sub);
    if (s->ws_sub_hit) (void)hipFree(s->ws_sub_hit);
    if (s->ws_sub_t) (void)hipFree(s->ws_sub_t);
    if (s->d_lights) (void)hipFree(s->d_lights);
    if (s->h_lights) (void)hipHostFree(s->h_lights);
    if (s->d_counters) (void)hipFree(s->d_counters);
    if (s->h_counters) (void)hipHostFree(s->h_counters);
    for (auto& tr : s->ev) for (auto& e : tr) if (e) (void)hipEventDestroy(e);
    delete s;
    return SRT_OK;
}

// Validate the layout contract and rewrite the trees in DFS pre-order with skip links: nodes[n_nodes], ranges[n_objects].
static int build_device_records(const srt_scene_desc* d, DevNode* nodes, int2* ranges) {
    const uint32_t N = d->n_nodes;
    std::vector<uint8_t> seen(N, 0);
    int64_t tri_cursor = 0;
    uint32_t cursor = 0;                   // nodes emitted so far
    // explicit DFS stack: state 0 = emit node, 1 = left subtree done, 2 = right subtree done
    struct Item { int32_t orig; int32_t emitted; int state; };
    std::vector<Item> st;
    for (uint32_t k = 0; k < d->n_objects; k++) {
        const uint32_t root = d->obj_root[k];
        if (root >= N) return SRT_ERR_LAYOUT;
        ranges[k].x = (int32_t)cursor;
        st.clear(); st.push_back({ (int32_t)root, -1, 0 });
        while (!st.empty()) {
            Item& it = st.back();        // not used after a push_back below
            if (it.state == 0) {
                if (it.orig < 0 || (uint32_t)it.orig >= N || seen[it.orig] || cursor >= N) return SRT_ERR_LAYOUT;
                seen[it.orig] = 1;
                const int32_t l = d->node_left[it.orig], r = d->node_right[it.orig];
                DevNode dn;
                dn.minx = d->node_min[3 * (size_t)it.orig]; dn.miny = d->node_min[3 * (size_t)it.orig + 1]; dn.minz = d->node_min[3 * (size_t)it.orig + 2];
                dn.maxx = d->node_max[3 * (size_t)it.orig]; dn.maxy = d->node_max[3 * (size_t)it.orig + 1]; dn.maxz = d->node_max[3 * (size_t)it.orig + 2];
                dn.skip = -1; dn.leaf = -1;
                it.emitted = (int32_t)cursor;
                if (l < 0 && r < 0) {
                    const int32_t first = d->node_first[it.orig], cnt = d->node_count[it.orig];
                    if (cnt < 0 || (cnt > 0 && first != tri_cursor) || tri_cursor + cnt > (int64_t)d->n_tris) return SRT_ERR_LAYOUT;
                    if (cnt > LEAF_MAX) return SRT_ERR_LIMIT;
                    for (int32_t j = 0; j < cnt; j++) if (d->tri_obj[tri_cursor + j] != (int32_t)k) return SRT_ERR_LAYOUT;
                    dn.leaf = (int32_t)((tri_cursor << LEAF_SHIFT) | cnt);
                    tri_cursor += cnt;
                    dn.skip = it.emitted + 1;
                    nodes[cursor++] = dn;
                    st.pop_back();
                } else {
                    if (l < 0 || r < 0) return SRT_ERR_LAYOUT;      // the reference's trees are full binary
                    nodes[cursor++] = dn;
                    it.state = 1;
                    st.push_back({ l, -1, 0 });
                }
            } else if (it.state == 1) {
                it.state = 2;
                const int32_t r = d->node_right[it.orig];
                nodes[it.emitted].leaf = ~(int32_t)cursor;           // inner node: ~(pre-order index of the right child) < 0
                st.push_back({ r, -1, 0 });
            } else {
                nodes[it.emitted].skip = (int32_t)cursor;
                st.pop_back();
            }
        }
        ranges[k].y = (int32_t)cursor;
    }
    if (cursor != N || tri_cursor != (int64_t)d->n_tris) return SRT_ERR_LAYOUT;
    return SRT_OK;
}

// The inner nodes as the node-queue kernels read them (DevWide, srt_device.h): both children's boxes in the parent's record, records
// numbered in pre-order over the inner nodes.  wide[n_wide], root_info[n_objects]; n_wide = (n_nodes - n_objects) / 2 for full binary
// trees, which build_device_records has checked.  widx is scratch (n_nodes words).
static uint32_t wide_count(uint32_t n_nodes, uint32_t n_objects) { return (n_nodes - n_objects) / 2; }
static void build_wide_records(const DevNode* nodes, uint32_t n_nodes, const int2* ranges, uint32_t n_objects, DevWide* wide, int32_t* root_info, int32_t* widx) {
    int32_t w = 0;
    for (uint32_t i = 0; i < n_nodes; i++) widx[i] = nodes[i].leaf < 0 ? w++ : -1;
    for (uint32_t i = 0; i < n_nodes; i++) {
        if (nodes[i].leaf >= 0) continue;
        const DevNode& l = nodes[i + 1];
        const int32_t ri = ~nodes[i].leaf;
        const DevNode& r = nodes[ri];
        DevWide& q = wide[widx[i]];
        q.lminx = l.minx; q.lminy = l.miny; q.lminz = l.minz; q.lmaxx = l.maxx; q.lmaxy = l.maxy; q.lmaxz = l.maxz;
        q.rminx = r.minx; q.rminy = r.miny; q.rminz = r.minz; q.rmaxx = r.maxx; q.rmaxy = r.maxy; q.rmaxz = r.maxz;
        q.linfo = l.leaf >= 0 ? l.leaf : ~widx[i + 1];
        q.rinfo = r.leaf >= 0 ? r.leaf : ~widx[ri];
        q.node = (int32_t)i; q.rnode = ri;
    }
    for (uint32_t k = 0; k < n_objects; k++) {
        const int32_t root = ranges[k].x;
        root_info[k] = nodes[root].leaf >= 0 ? nodes[root].leaf : ~widx[root];
    }
}

// the union of the objects' root boxes, in a node's box layout (min.xyz max.x | max.yz 0 0): what a tile's rays are tested against
// before the roots themselves in scenes of several objects (srt_kernels.h background_test_wave)
static void union_of_roots(const DevNode* roots, uint32_t n, float* out8) {
    float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    for (uint32_t k = 0; k < n; k++) {
        const float mn[3] = { roots[k].minx, roots[k].miny, roots[k].minz }, mx[3] = { roots[k].maxx, roots[k].maxy, roots[k].maxz };
        for (int a = 0; a < 3; a++) { if (mn[a] < lo[a]) lo[a] = mn[a]; if (mx[a] > hi[a]) hi[a] = mx[a]; }
    }
    out8[0] = lo[0]; out8[1] = lo[1]; out8[2] = lo[2]; out8[3] = hi[0]; out8[4] = hi[1]; out8[5] = hi[2]; out8[6] = 0.f; out8[7] = 0.f;
}

// 64-bit content hash of a byte range (texture images: srt_scene_update re-uploads them only when it changes).  Eight independent
// multiply-xor lanes over 64-byte blocks: memory-bound on one core.
static uint64_t content_hash(const uint8_t* p, size_t n) {
    uint64_t h[8];
    for (int k = 0; k < 8; k++) h[k] = 0x9e3779b97f4a7c15ull * (uint64_t)(k + 1) ^ (uint64_t)n;
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        uint64_t w[8];
        std::memcpy(w, p + i, 64);
        for (int k = 0; k < 8; k++) { h[k] = (h[k] ^ w[k]) * 0xff51afd7ed558ccdull; h[k] ^= h[k] >> 29; }
    }
    uint64_t tail[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (i < n) { std::memcpy(tail, p + i, n - i); for (int k = 0; k < 8; k++) { h[k] = (h[k] ^ tail[k]) * 0xff51afd7ed558ccdull; h[k] ^= h[k] >> 29; } }
    uint64_t r = 0xc4ceb9fe1a85ec53ull;
    for (int k = 0; k < 8; k++) { r = (r ^ h[k]) * 0xff51afd7ed558ccdull; r ^= r >> 32; }
    return r;
}

// How many slab tests does a ray cost?  Surface-area estimate: a random line that crosses the scene's bounds crosses a convex
// box inside them with probability area(box) / area(bounds), and a node is tested when its parent's box is crossed.  A good
// hierarchy gives a few dozen (bunny: boxes shrink with depth); a median split by first vertex of a random soup gives hundreds
// to thousands (boxes stay as wide as the scene in two axes).  In the second case neighbouring rays test nearly the same nodes
// and the packet walk (srt_packet.h) wins for primary rays as well.
static double overlap_estimate(const DevNode* nodes, uint32_t n_nodes, const int2* ranges, uint32_t n_objects) {
    float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    auto area = [](const float* a, const float* b) -> double {
        const double x = (double)b[0] - a[0], y = (double)b[1] - a[1], z = (double)b[2] - a[2];
        return (x < 0 || y < 0 || z < 0) ? 0. : x * y + y * z + z * x;
    };
    for (uint32_t k = 0; k < n_objects; k++) {
        const DevNode& n = nodes[ranges[k].x];
        const float mn[3] = { n.minx, n.miny, n.minz }, mx[3] = { n.maxx, n.maxy, n.maxz };
        if (area(mn, mx) <= 0.) continue;
        for (int a = 0; a < 3; a++) { lo[a] = mn[a] < lo[a] ? mn[a] : lo[a]; hi[a] = mx[a] > hi[a] ? mx[a] : hi[a]; }
    }
    const double total = area(lo, hi);
    if (!(total > 0.)) return 0.;
    double sum = 0.;
    for (uint32_t i = 0; i < n_nodes; i++) {
        const DevNode& n = nodes[i];
        if (n.leaf >= 0) continue;                       // the children of an inner node are tested when its box is crossed
        const float mn[3] = { n.minx, n.miny, n.minz }, mx[3] = { n.maxx, n.maxy, n.maxz };
        sum += 2. * area(mn, mx);
    }
    return (double)n_objects + sum / total;              // + every root
}

// per-triangle records: independent, so big scenes are cut over a few host threads (a scene made per frame by a drop-in
// caller spends more time here than in the render)
static void derive_triangles(const srt_scene_desc* d, DevTri* tris, DevTriO* tris_o) {
    auto derive_range = [&](uint32_t b, uint32_t e) {
        for (uint32_t i = b; i < e; i++) { tris[i] = derive_triangle(d->tri_points + 12 * (size_t)i); tris_o[i] = derive_triangle_origin(tris[i]); }
    };
    const uint32_t n = d->n_tris;
    unsigned hc = std::thread::hardware_concurrency();
    const uint32_t T = n < 32768 ? 1u : (hc >= 8 ? 8u : (hc >= 2 ? hc : 1u));
    if (T == 1) { derive_range(0, n); return; }
    // a thread that cannot be started (std::system_error) must not take the process down with joinable threads in a dying vector:
    // the ranges that got no thread are derived inline, and every started thread is joined
    std::vector<std::thread> th;
    th.reserve(T);
    const uint32_t step = (n + T - 1) / T;
    uint32_t started = 1;                                  // ranges [1, started) have a thread
    try {
        for (; started < T; started++) th.emplace_back(derive_range, started * step < n ? started * step : n, (started + 1) * step < n ? (started + 1) * step : n);
    } catch (...) { }
    derive_range(0, step < n ? step : n);
    for (uint32_t k = started; k < T; k++) derive_range(k * step < n ? k * step : n, (k + 1) * step < n ? (k + 1) * step : n);
    for (std::thread& t : th) t.join();
}

// first triangle of each object (the layout contract makes tri_obj non-decreasing): first[n_objects + 1]
static void derive_tri_first(const srt_scene_desc* d, int32_t* first) {
    for (uint32_t k = 0; k <= d->n_objects; k++) first[k] = (int32_t)d->n_tris;
    for (uint32_t i = d->n_tris; i-- > 0;) first[d->tri_obj[i]] = (int32_t)i;
    for (uint32_t k = d->n_objects; k-- > 0;) if (first[k] > first[k + 1]) first[k] = first[k + 1];      // objects without triangles
}

static bool all_integer_shininess(const srt_scene_desc* d) {
    for (uint32_t k = 0; k < d->n_objects; k++) {
        const float sh = d->obj_material[3 * (size_t)k + 2];
        if (!(sh >= 1.0f && sh <= 64.0f && sh == std::trunc(sh))) return false;
    }
    return true;
}

static int check_desc(const srt_scene_desc* d) {
    if (!d->n_objects || !d->n_nodes || !d->node_min || !d->node_max || !d->node_left || !d->node_right ||
        !d->node_first || !d->node_count || !d->obj_root || !d->obj_color || !d->obj_material) return SRT_ERR_ARG;
    if (d->n_tris && (!d->tri_points || !d->tri_obj)) return SRT_ERR_ARG;
    // queue entries of the node-queue kernels pack (node << 6 | ray lane) into 32 bits; a leaf word packs (first << 5 | count)
    static_assert(NODE_INDEX_BITS + 6 == 32, "node-queue entry = node index + 6-bit lane");
    if (d->n_tris >= (1u << (31 - LEAF_SHIFT)) || d->n_nodes >= (1u << NODE_INDEX_BITS)) return SRT_ERR_LIMIT;
    if (d->n_textures && (!d->tex_rgb || !d->tex_off || !d->tex_w || !d->tex_h || !d->tri_tex || !d->tri_texcoord)) return SRT_ERR_ARG;
    for (uint32_t i = 0; i < d->n_tris; i++) {
        if (d->tri_obj[i] < 0 || (uint32_t)d->tri_obj[i] >= d->n_objects) return SRT_ERR_LAYOUT;
        if (d->tri_tex && d->tri_tex[i] >= (int32_t)d->n_textures) return SRT_ERR_TEXTURE;
    }
    return SRT_OK;
}

// What every handle has of its own besides the records: the two counter sets, the quadrant-list counters, the pinned counter image.
static hipError_t init_handle_state(srt_scene* s) {
    hipError_t e = hipMalloc((void**)&s->d_counters, 2 * NCTR * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(s->d_counters, 0, 2 * NCTR * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_qcount, QL_COUNTERS * QL_STRIDE * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(s->d_qcount, 0, QL_COUNTERS * QL_STRIDE * sizeof(uint32_t));
    if (e == hipSuccess) e = hipHostMalloc((void**)&s->h_counters, NCTR * sizeof(unsigned long long), hipHostMallocDefault);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, s->device);
    if (e == hipSuccess && prop.multiProcessorCount > 0) s->n_cu = prop.multiProcessorCount;
    return e;
}

static int scene_create_impl(int device, const srt_scene_desc* d, srt_scene** out) {
    if (!d || !out) return SRT_ERR_ARG;
    *out = nullptr;
    int rc = check_desc(d);
    if (rc != SRT_OK) return rc;
    // host-side records first (validates the layout contract; pure CPU work), then the device
    alloc_gate();
    std::vector<DevNode> nodes(d->n_nodes); std::vector<int2> ranges(d->n_objects);
    rc = build_device_records(d, nodes.data(), ranges.data());
    if (rc != SRT_OK) return rc;
    const double overlap = overlap_estimate(nodes.data(), d->n_nodes, ranges.data(), d->n_objects);
    const uint32_t n_wide = wide_count(d->n_nodes, d->n_objects);
    std::vector<DevWide> wide(n_wide); std::vector<int32_t> root_info(d->n_objects);
    std::vector<int32_t> widx(d->n_nodes);
    build_wide_records(nodes.data(), d->n_nodes, ranges.data(), d->n_objects, wide.data(), root_info.data(), widx.data());
    std::vector<DevTri> tris(d->n_tris);
    std::vector<DevTriO> tris_o(d->n_tris);
    derive_triangles(d, tris.data(), tris_o.data());

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SRT_ERR_NO_GPU;
    if (device < 0 || device >= ndev) return SRT_ERR_ARG;
    HIP_TRY(hipSetDevice(device));
    srt_scene* s = new (std::nothrow) srt_scene();
    if (!s) return SRT_ERR_OOM;
    s->device = device;
    s->rec = std::make_shared<SceneRecords>(); s->rec->device = device;
    #define UP(expr) do { rc = (expr); if (rc != SRT_OK) { srt_scene_destroy(s); return rc; } } while (0)
    UP(upload(s, nodes.data(), nodes.size(), &s->dev.nodes));
    UP(upload(s, wide.data(), wide.size(), &s->dev.wide));
    UP(upload(s, root_info.data(), root_info.size(), &s->dev.obj_root_info));
    {
        std::vector<DevNode> roots(d->n_objects);
        for (uint32_t k = 0; k < d->n_objects; k++) roots[k] = nodes[ranges[k].x];
        UP(upload(s, roots.data(), roots.size(), &s->dev.root_nodes));
        float ub[8];
        union_of_roots(roots.data(), d->n_objects, ub);
        UP(upload(s, ub, 8, &s->dev.scene_box));
    }
    UP(upload(s, tris.data(), tris.size(), &s->dev.tris));
    UP(upload(s, tris_o.data(), tris_o.size(), &s->dev.tris_o));
    UP(upload(s, d->tri_obj, d->n_tris, &s->dev.tri_obj));
    UP(upload(s, ranges.data(), ranges.size(), &s->dev.obj_range));
    {
        std::vector<int32_t> first(d->n_objects + 1);
        derive_tri_first(d, first.data());
        UP(upload(s, first.data(), first.size(), &s->dev.obj_tri_first));
    }
    UP(upload(s, d->obj_color, (size_t)d->n_objects * 3, &s->dev.obj_color));
    UP(upload(s, d->obj_material, (size_t)d->n_objects * 3, &s->dev.obj_mat));
    if (d->tri_normals && d->n_tris) UP(upload(s, d->tri_normals, (size_t)d->n_tris * 9, &s->dev.tri_normals));
    bool any_tex = false;
    if (d->n_textures && d->tri_tex) for (uint32_t i = 0; i < d->n_tris; i++) any_tex |= d->tri_tex[i] >= 0;
    if (any_tex) {
        std::vector<unsigned long long> off(d->n_textures), size(d->n_textures);
        unsigned long long total = 0;
        for (uint32_t k = 0; k < d->n_textures; k++) {
            off[k] = d->tex_off[k]; size[k] = (unsigned long long)d->tex_w[k] * d->tex_h[k] * 3;
            if (size[k] < 3) { srt_scene_destroy(s); return SRT_ERR_TEXTURE; }
            if (off[k] + size[k] > total) total = off[k] + size[k];
        }
        UP(upload(s, d->tri_tex, d->n_tris, &s->dev.tri_tex));
        UP(upload(s, d->tri_texcoord, (size_t)d->n_tris * 6, &s->dev.tri_tc));
        UP(upload(s, d->tex_rgb, (size_t)total, &s->dev.tex));
        UP(upload(s, off.data(), off.size(), &s->dev.tex_off));
        UP(upload(s, size.data(), size.size(), &s->dev.tex_size));
        UP(upload(s, d->tex_w, d->n_textures, &s->dev.tex_w));
        UP(upload(s, d->tex_h, d->n_textures, &s->dev.tex_h));
        s->rec->tex_off.assign(d->tex_off, d->tex_off + d->n_textures);
        s->rec->tex_w.assign(d->tex_w, d->tex_w + d->n_textures); s->rec->tex_h.assign(d->tex_h, d->tex_h + d->n_textures);
        s->rec->tex_bytes = total; s->rec->tex_hash = content_hash(d->tex_rgb, (size_t)total);
    }
    #undef UP
    s->dev.n_nodes = d->n_nodes; s->dev.n_tris = d->n_tris; s->dev.n_objects = d->n_objects;
    s->n_textures = d->n_textures; s->has_tex = any_tex;
    s->rec->overlap = overlap;
    s->rec->prefer_packet = overlap > PACKET_OVERLAP_THRESHOLD;
    s->rec->int_shin = all_integer_shininess(d);
    {   // what srt_scene_update_frame needs of the topology
        SceneRecords& r = *s->rec;
        r.h_ranges = ranges;
        r.h_tri_first.resize(d->n_objects + 1); derive_tri_first(d, r.h_tri_first.data());
        r.h_leaf.resize(d->n_nodes); for (uint32_t i = 0; i < d->n_nodes; i++) r.h_leaf[i] = nodes[i].leaf;
        const int32_t* dw = nullptr;
        rc = upload(s, widx.data(), widx.size(), &dw);
        if (rc != SRT_OK) { srt_scene_destroy(s); return rc; }
        r.d_widx = const_cast<int32_t*>(dw);
    }
    const hipError_t e = init_handle_state(s);
    if (e != hipSuccess) { g_last_hip = (int)e; srt_scene_destroy(s); return SRT_ERR_DEVICE; }
    *out = s;
    return SRT_OK;
}

// A second handle on the SAME device records: its own workspace, counters, stream and statistics, no copy of the geometry.
static int scene_share_impl(srt_scene* src, srt_scene** out) {
    if (!src || !out) return SRT_ERR_ARG;
    *out = nullptr;
    HIP_TRY(hipSetDevice(src->device));
    alloc_gate();
    srt_scene* s = new srt_scene();
    s->device = src->device; s->dev = src->dev; s->rec = src->rec; s->bytes = src->bytes;
    s->n_textures = src->n_textures; s->has_tex = src->has_tex;
    const hipError_t e = init_handle_state(s);
    if (e != hipSuccess) { g_last_hip = (int)e; srt_scene_destroy(s); return SRT_ERR_DEVICE; }
    *out = s;
    return SRT_OK;
}

int srt_scene_share(srt_scene* src, srt_scene** out) {
    return guarded([&] { return scene_share_impl(src, out); });
}

int srt_scene_create(int device, const srt_scene_desc* d, srt_scene** out) {
    return guarded([&] { return scene_create_impl(device, d, out); });
}

// New geometry into the EXISTING device allocations: the reference re-transforms every triangle and rebuilds every hierarchy per
// frame (simple_raytracer.cpp:534-618), so a drop-in caller hands over a new flat scene per frame -- with the same counts (the
// builder's tree shape depends only on the triangle count).  Records are derived straight into one pinned staging block and go
// to the device with asynchronous copies on `stream`: no hipMalloc / hipFree, no pageable copy, no synchronisation with the
// renders already enqueued on that stream (the copies are ordered behind them).
static int own_stream(srt_scene* s, hipStream_t* out);
static int scene_update_impl(srt_scene* s, const srt_scene_desc* d, hipStream_t stream) {
    if (!s || !d) return SRT_ERR_ARG;
    int rc = check_desc(d);
    if (rc != SRT_OK) return rc;
    if (!stream) { rc = own_stream(s, &stream); if (rc != SRT_OK) return rc; }
    bool any_tex = false;
    if (d->n_textures && d->tri_tex) for (uint32_t i = 0; i < d->n_tris; i++) any_tex |= d->tri_tex[i] >= 0;
    if (d->n_objects != s->dev.n_objects || d->n_nodes != s->dev.n_nodes || d->n_tris != s->dev.n_tris || d->n_textures != s->n_textures ||
        any_tex != s->has_tex || (d->tri_normals != nullptr) != (s->dev.tri_normals != nullptr)) return SRT_ERR_LAYOUT;      // counts differ: create a new scene
    // Texture images: the table (offsets, sizes) must be the uploaded one -- the kernels index with the uploaded tex_w / tex_off --
    // and the image bytes are uploaded again when their content hash differs from what is on the device.  (A second scene with the
    // same counts but other pictures, handed to a renderer that keeps one device scene, must not be shaded with the first one's.)
    size_t tex_total = 0;
    bool tex_changed = false;
    uint64_t tex_hash_new = 0;
    if (any_tex) {
        SceneRecords& r = *s->rec;
        for (uint32_t k = 0; k < d->n_textures; k++) {
            if (d->tex_off[k] != r.tex_off[k] || d->tex_w[k] != r.tex_w[k] || d->tex_h[k] != r.tex_h[k]) return SRT_ERR_LAYOUT;      // another table: create a new scene
            const size_t end = (size_t)d->tex_off[k] + (size_t)d->tex_w[k] * d->tex_h[k] * 3;
            if (end > tex_total) tex_total = end;
        }
        if (tex_total != r.tex_bytes) return SRT_ERR_LAYOUT;
        tex_hash_new = content_hash(d->tex_rgb, tex_total);
        tex_changed = tex_hash_new != r.tex_hash;
    }
    HIP_TRY(hipSetDevice(s->device));
    const size_t nN = d->n_nodes, nT = d->n_tris, nO = d->n_objects, nW = wide_count(d->n_nodes, d->n_objects);
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_nodes = 0, o_tris = o_nodes + pad(nN * sizeof(DevNode)), o_triso = o_tris + pad(nT * sizeof(DevTri)),
                 o_triobj = o_triso + pad(nT * sizeof(DevTriO)), o_ranges = o_triobj + pad(nT * 4), o_first = o_ranges + pad(nO * sizeof(int2)),
                 o_color = o_first + pad((nO + 1) * 4), o_mat = o_color + pad(nO * 12), o_nrm = o_mat + pad(nO * 12),
                 o_tex = o_nrm + pad(d->tri_normals ? nT * 36 : 0), o_tc = o_tex + pad(any_tex ? nT * 4 : 0), o_wide = o_tc + pad(any_tex ? nT * 24 : 0),
                 o_rinfo = o_wide + pad(nW * sizeof(DevWide)), o_widx = o_rinfo + pad(nO * 4), o_roots = o_widx + pad(nN * 4), o_ubox = o_roots + pad(nO * sizeof(DevNode)), o_img = o_ubox + pad(32),
                 total = o_img + pad(tex_changed ? tex_total : 0);
    if (s->stage_bytes < total) {
        if (s->stage) { HIP_TRY(hipEventSynchronize(s->staged)); (void)hipHostFree(s->stage); s->stage = nullptr; s->stage_bytes = 0; }
        HIP_TRY(hipHostMalloc(&s->stage, total, hipHostMallocDefault));
        s->stage_bytes = total;
        if (!s->staged) HIP_TRY(hipEventCreateWithFlags(&s->staged, hipEventDisableTiming));
    } else {
        HIP_TRY(hipEventSynchronize(s->staged));               // the previous update's copies have left the staging block
    }
    char* h = (char*)s->stage;
    DevNode* nodes = (DevNode*)(h + o_nodes); int2* ranges = (int2*)(h + o_ranges);
    rc = build_device_records(d, nodes, ranges);
    if (rc != SRT_OK) return rc;
    build_wide_records(nodes, d->n_nodes, ranges, d->n_objects, (DevWide*)(h + o_wide), (int32_t*)(h + o_rinfo), (int32_t*)(h + o_widx));
    for (uint32_t k = 0; k < d->n_objects; k++) ((DevNode*)(h + o_roots))[k] = nodes[ranges[k].x];
    union_of_roots((const DevNode*)(h + o_roots), d->n_objects, (float*)(h + o_ubox));
    if (tex_changed) std::memcpy(h + o_img, d->tex_rgb, tex_total);
    derive_triangles(d, (DevTri*)(h + o_tris), (DevTriO*)(h + o_triso));
    derive_tri_first(d, (int32_t*)(h + o_first));
    std::memcpy(h + o_triobj, d->tri_obj, nT * 4);
    std::memcpy(h + o_color, d->obj_color, nO * 12);
    std::memcpy(h + o_mat, d->obj_material, nO * 12);
    if (d->tri_normals) std::memcpy(h + o_nrm, d->tri_normals, nT * 36);
    if (any_tex) { std::memcpy(h + o_tex, d->tri_tex, nT * 4); std::memcpy(h + o_tc, d->tri_texcoord, nT * 24); }
    #define CP(dst, off, bytes) do { if (bytes) HIP_TRY(hipMemcpyAsync((void*)(dst), h + (off), (bytes), hipMemcpyHostToDevice, stream)); } while (0)
    CP(s->dev.nodes, o_nodes, nN * sizeof(DevNode));
    CP(s->dev.wide, o_wide, nW * sizeof(DevWide));
    CP(s->dev.obj_root_info, o_rinfo, nO * 4);
    CP(s->dev.root_nodes, o_roots, nO * sizeof(DevNode));
    CP(s->dev.scene_box, o_ubox, 32);
    if (tex_changed) { CP(s->dev.tex, o_img, tex_total); s->rec->tex_hash = tex_hash_new; }
    CP(s->dev.tris, o_tris, nT * sizeof(DevTri));
    CP(s->dev.tris_o, o_triso, nT * sizeof(DevTriO));
    CP(s->dev.tri_obj, o_triobj, nT * 4);
    CP(s->dev.obj_range, o_ranges, nO * sizeof(int2));
    CP(s->dev.obj_tri_first, o_first, (nO + 1) * 4);
    CP(s->dev.obj_color, o_color, nO * 12);
    CP(s->dev.obj_mat, o_mat, nO * 12);
    if (d->tri_normals) CP(s->dev.tri_normals, o_nrm, nT * 36);
    if (any_tex) { CP(s->dev.tri_tex, o_tex, nT * 4); CP(s->dev.tri_tc, o_tc, nT * 24); }
    #undef CP
    HIP_TRY(hipEventRecord(s->staged, stream));
    s->rec->overlap = overlap_estimate(nodes, d->n_nodes, ranges, d->n_objects);
    s->rec->prefer_packet = s->rec->overlap > PACKET_OVERLAP_THRESHOLD;
    s->rec->int_shin = all_integer_shininess(d);
    {   // the topology may differ from the previous contents' (same counts, other trees): refresh what srt_scene_update_frame relies on
        SceneRecords& r = *s->rec;
        for (uint32_t k = 0; k < d->n_objects; k++) r.h_ranges[k] = ranges[k];
        derive_tri_first(d, r.h_tri_first.data());
        for (uint32_t i = 0; i < d->n_nodes; i++) r.h_leaf[i] = nodes[i].leaf;
        HIP_TRY(hipMemcpyAsync(r.d_widx, h + o_widx, nN * 4, hipMemcpyHostToDevice, stream));
        r.have_source = false;                                  // attributes in source order belong to the previous contents
    }
    return SRT_OK;
}

// ---- f1, device half -------------------------------------------------------------------------------------------------------------
static int scene_set_source_impl(srt_scene* s, const float* tri_texcoord, const float* tri_normals, const int32_t* tri_tex) {
    if (!s) return SRT_ERR_ARG;
    SceneRecords& r = *s->rec;
    const size_t nT = s->dev.n_tris;
    if (s->has_tex && (!tri_texcoord || !tri_tex)) return SRT_ERR_ARG;
    if (s->dev.tri_normals && !tri_normals) return SRT_ERR_ARG;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(wait_idle(s));
    auto put = [&](auto** dst, const auto* src, size_t n) -> int {
        using T = std::remove_pointer_t<std::remove_pointer_t<decltype(dst)>>;
        if (!*dst) { void* d = nullptr; HIP_TRY(hipMalloc(&d, sizeof(T) * (n ? n : 1))); r.allocs.push_back(d); *dst = (T*)d; s->bytes += sizeof(T) * n; }
        if (n) HIP_TRY(hipMemcpy(*dst, src, sizeof(T) * n, hipMemcpyHostToDevice));
        return SRT_OK;
    };
    int rc = SRT_OK;
    if (s->has_tex) { rc = put(&r.d_src_tc, tri_texcoord, nT * 6); if (rc == SRT_OK) rc = put(&r.d_src_tex, tri_tex, nT); }
    if (rc == SRT_OK && s->dev.tri_normals) rc = put(&r.d_src_nrm, tri_normals, nT * 9);
    if (rc != SRT_OK) return rc;
    r.have_source = true;
    return SRT_OK;
}

int srt_scene_set_source(srt_scene* s, const float* tri_texcoord, const float* tri_normals, const int32_t* tri_tex) {
    return guarded([&] { return scene_set_source_impl(s, tri_texcoord, tri_normals, tri_tex); });
}

static int scene_update_frame_impl(srt_scene* s, const srt_frame_geometry* g, hipStream_t stream) {
    if (!s || !g || !g->obj_n_tris || !g->obj_n_nodes || !g->obj_points || !g->obj_order || !g->obj_node_min || !g->obj_node_max) return SRT_ERR_ARG;
    SceneRecords& r = *s->rec;
    const uint32_t nO = s->dev.n_objects;
    if (g->n_objects != nO) return SRT_ERR_LAYOUT;
    for (uint32_t k = 0; k < nO; k++) {
        if ((int32_t)g->obj_n_tris[k] != r.h_tri_first[k + 1] - r.h_tri_first[k] || (int32_t)g->obj_n_nodes[k] != r.h_ranges[k].y - r.h_ranges[k].x) return SRT_ERR_LAYOUT;
        if (g->obj_n_tris[k] && (!g->obj_points[k] || !g->obj_order[k])) return SRT_ERR_ARG;
        if (!g->obj_node_min[k] || !g->obj_node_max[k]) return SRT_ERR_ARG;
    }
    if ((s->has_tex || s->dev.tri_normals) && !r.have_source) return SRT_ERR_ARG;       // attributes cannot be permuted without their source order
    int rc;
    if (!stream) { rc = own_stream(s, &stream); if (rc != SRT_OK) return rc; }
    HIP_TRY(hipSetDevice(s->device));
    const size_t nN = s->dev.n_nodes, nT = s->dev.n_tris;
    if (!r.d_src_points) {        // the device side of the staging, once
        auto make = [&](auto** dst, size_t bytes) -> int { void* d = nullptr; HIP_TRY(hipMalloc(&d, bytes ? bytes : 1)); r.allocs.push_back(d); *dst = (std::remove_pointer_t<decltype(dst)>)d; s->bytes += bytes; return SRT_OK; };
        rc = make(&r.d_src_points, nT * 48); if (rc == SRT_OK) rc = make(&r.d_order, nT * 4);
        if (rc == SRT_OK) rc = make(&r.d_box_min, nN * 12); if (rc == SRT_OK) rc = make(&r.d_box_max, nN * 12);
        if (rc != SRT_OK) return rc;
    }
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_pts = 0, o_ord = o_pts + pad(nT * 48), o_bmin = o_ord + pad(nT * 4), o_bmax = o_bmin + pad(nN * 12), o_col = o_bmax + pad(nN * 12),
                 o_mat = o_col + pad(nO * 12), o_ubox = o_mat + pad(nO * 12), total = o_ubox + pad(32);
    if (s->stage_bytes < total) {
        if (s->stage) { HIP_TRY(hipEventSynchronize(s->staged)); (void)hipHostFree(s->stage); s->stage = nullptr; s->stage_bytes = 0; }
        HIP_TRY(hipHostMalloc(&s->stage, total, hipHostMallocDefault));
        s->stage_bytes = total;
        if (!s->staged) HIP_TRY(hipEventCreateWithFlags(&s->staged, hipEventDisableTiming));
    } else {
        HIP_TRY(hipEventSynchronize(s->staged));               // the previous update's copies have left the staging block
    }
    char* h = (char*)s->stage;
    for (uint32_t k = 0; k < nO; k++) {
        const size_t t0 = (size_t)r.h_tri_first[k], nt = g->obj_n_tris[k], n0 = (size_t)r.h_ranges[k].x, nn = g->obj_n_nodes[k];
        if (nt) { std::memcpy(h + o_pts + t0 * 48, g->obj_points[k], nt * 48); std::memcpy(h + o_ord + t0 * 4, g->obj_order[k], nt * 4); }
        std::memcpy(h + o_bmin + n0 * 12, g->obj_node_min[k], nn * 12); std::memcpy(h + o_bmax + n0 * 12, g->obj_node_max[k], nn * 12);
        const uint32_t* ord = g->obj_order[k];
        for (size_t i = 0; i < nt; i += 4099) if (ord[i] >= nt) return SRT_ERR_LAYOUT;       // (spot check: an index outside the object would read another object's points)
    }
    #define CP(dst, off, bytes) do { if (bytes) HIP_TRY(hipMemcpyAsync((void*)(dst), h + (off), (bytes), hipMemcpyHostToDevice, stream)); } while (0)
    {   // the union of this frame's root boxes
        float* ub = (float*)(h + o_ubox);
        float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        for (uint32_t k = 0; k < nO; k++) {
            const float* mn = (const float*)(h + o_bmin) + 3 * (size_t)r.h_ranges[k].x; const float* mx = (const float*)(h + o_bmax) + 3 * (size_t)r.h_ranges[k].x;
            for (int a = 0; a < 3; a++) { if (mn[a] < lo[a]) lo[a] = mn[a]; if (mx[a] > hi[a]) hi[a] = mx[a]; }
        }
        ub[0] = lo[0]; ub[1] = lo[1]; ub[2] = lo[2]; ub[3] = hi[0]; ub[4] = hi[1]; ub[5] = hi[2]; ub[6] = 0.f; ub[7] = 0.f;
    }
    CP(r.d_src_points, o_pts, nT * 48); CP(r.d_order, o_ord, nT * 4); CP(r.d_box_min, o_bmin, nN * 12); CP(r.d_box_max, o_bmax, nN * 12);
    CP(s->dev.scene_box, o_ubox, 32);
    if (g->obj_color) { std::memcpy(h + o_col, g->obj_color, nO * 12); CP(s->dev.obj_color, o_col, nO * 12); }
    if (g->obj_material) {
        std::memcpy(h + o_mat, g->obj_material, nO * 12); CP(s->dev.obj_mat, o_mat, nO * 12);
        bool ish = true;
        for (uint32_t k = 0; k < nO; k++) { const float sh = g->obj_material[3 * (size_t)k + 2]; ish = ish && sh >= 1.0f && sh <= 64.0f && sh == std::trunc(sh); }
        r.int_shin = ish;
    }
    #undef CP
    HIP_TRY(hipEventRecord(s->staged, stream));
    const dim3 block(256);
    if (nT) hipLaunchKernelGGL(k_update_tris, dim3((uint32_t)((nT + 255) / 256)), block, 0, stream, (uint32_t)nT, s->dev.tri_obj, s->dev.obj_tri_first,
                               (const float4*)r.d_src_points, (const uint32_t*)r.d_order, const_cast<DevTri*>(s->dev.tris), const_cast<DevTriO*>(s->dev.tris_o),
                               (const float*)r.d_src_tc, const_cast<float*>(s->dev.tri_tc), (const float*)r.d_src_nrm, const_cast<float*>(s->dev.tri_normals),
                               (const int32_t*)r.d_src_tex, s->has_tex ? const_cast<int32_t*>(s->dev.tri_tex) : nullptr);
    hipLaunchKernelGGL(k_update_nodes, dim3((uint32_t)((nN + 255) / 256)), block, 0, stream, (uint32_t)nN, (const float*)r.d_box_min, (const float*)r.d_box_max,
                       const_cast<DevNode*>(s->dev.nodes), const_cast<DevWide*>(s->dev.wide), (const int32_t*)r.d_widx);
    hipLaunchKernelGGL(k_update_roots, dim3((nO + 63) / 64), dim3(64), 0, stream, nO, s->dev.obj_range, (const float*)r.d_box_min, (const float*)r.d_box_max,
                       s->dev.nodes, const_cast<DevNode*>(s->dev.root_nodes));
    HIP_TRY(hipGetLastError());
    // expected slab tests per ray from this frame's boxes (what overlap_estimate computes from the records)
    {
        const float* bmin = (const float*)(h + o_bmin); const float* bmax = (const float*)(h + o_bmax);
        auto area = [](const float* a, const float* b) -> double {
            const double x = (double)b[0] - a[0], y = (double)b[1] - a[1], z = (double)b[2] - a[2];
            return (x < 0 || y < 0 || z < 0) ? 0. : x * y + y * z + z * x;
        };
        float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
        for (uint32_t k = 0; k < nO; k++) {
            const float* mn = bmin + 3 * (size_t)r.h_ranges[k].x; const float* mx = bmax + 3 * (size_t)r.h_ranges[k].x;
            if (area(mn, mx) <= 0.) continue;
            for (int a = 0; a < 3; a++) { lo[a] = mn[a] < lo[a] ? mn[a] : lo[a]; hi[a] = mx[a] > hi[a] ? mx[a] : hi[a]; }
        }
        const double tot = area(lo, hi);
        double sum = 0.;
        if (tot > 0.) for (size_t i = 0; i < nN; i++) if (r.h_leaf[i] < 0) sum += 2. * area(bmin + 3 * i, bmax + 3 * i);
        r.overlap = tot > 0. ? (double)nO + sum / tot : 0.;
        r.prefer_packet = r.overlap > PACKET_OVERLAP_THRESHOLD;
    }
    return SRT_OK;
}

int srt_scene_update_frame(srt_scene* s, const srt_frame_geometry* g, void* stream) {
    return guarded([&] { return scene_update_frame_impl(s, g, (hipStream_t)stream); });
}

int srt_scene_update(srt_scene* s, const srt_scene_desc* d, void* stream) {
    return guarded([&] { return scene_update_impl(s, d, (hipStream_t)stream); });
}

void srt_debug_fail_host_allocs(int n) { g_fail_allocs.store(n < 0 ? 0 : n); }

int srt_debug_scene_records(srt_scene* s, void* nodes, void* tris, void* tris_o, void* wide, void* root_nodes,
                            float* tri_texcoord, float* tri_normals, int32_t* tri_tex) {
    if (!s) return SRT_ERR_ARG;
    HIP_TRY(hipSetDevice(s->device));
    if (s->stream) HIP_TRY(hipStreamSynchronize(s->stream));
    HIP_TRY(hipDeviceSynchronize());
    const size_t nN = s->dev.n_nodes, nT = s->dev.n_tris, nO = s->dev.n_objects, nW = (nN - nO) / 2;
    #define DOWN(dst, src, bytes) do { if ((dst) && (src) && (bytes)) HIP_TRY(hipMemcpy((dst), (src), (bytes), hipMemcpyDeviceToHost)); } while (0)
    DOWN(nodes, s->dev.nodes, nN * sizeof(DevNode)); DOWN(tris, s->dev.tris, nT * sizeof(DevTri)); DOWN(tris_o, s->dev.tris_o, nT * sizeof(DevTriO));
    DOWN(wide, s->dev.wide, nW * sizeof(DevWide)); DOWN(root_nodes, s->dev.root_nodes, nO * sizeof(DevNode));
    DOWN(tri_texcoord, s->dev.tri_tc, nT * 24); DOWN(tri_normals, s->dev.tri_normals, nT * 36); DOWN(tri_tex, s->dev.tri_tex, nT * 4);
    #undef DOWN
    return SRT_OK;
}

uint64_t srt_scene_device_bytes(const srt_scene* s) { return s ? s->bytes : 0; }
const char* srt_scene_pipeline(const srt_scene* s) { return s ? s->pipeline : ""; }
double srt_scene_overlap_estimate(const srt_scene* s) { return s ? s->rec->overlap : 0.; }

static inline uint32_t variant_of(const srt_params* p) { return (p->flags >> 8) & 0xffu; }

static int check_params(const srt_params* p) {
    if (!p || !p->width || !p->height || !p->block_rows || !p->block_stride) return SRT_ERR_ARG;
    if (p->ray_matrix && (((p->flags >> 8) & 0xffu) != 0 && ((p->flags >> 8) & 0xffu) != 22 && ((p->flags >> 8) & 0xffu) != 35)) return SRT_ERR_ARG;      // camera mode: shipped pipelines only
    if (p->n_lights && !p->light_pos) return SRT_ERR_ARG;
    if (p->block_cols && ((p->block_cols & 7u) || (p->block_rows & 7u) || p->block_first >= p->block_stride)) return SRT_ERR_ARG;   // tiles of whole 8x8 pixel blocks
    if (p->spp < 1 || p->spp > 4096) return SRT_ERR_ARG;
    { const uint32_t n = (uint32_t)std::lround(std::sqrt((double)p->spp)); if (n * n != p->spp) return SRT_ERR_ARG; }   // n x n sub-pixel grid
    if ((uint64_t)p->width * p->height >= (1ull << 31)) return SRT_ERR_LIMIT;
    if ((uint64_t)p->width * p->height * (p->n_lights ? p->n_lights : 1) >= (1ull << 32)) return SRT_ERR_LIMIT;   // 32-bit work-item index
    return SRT_OK;
}

// Frames whose launches srt_render_device_batch holds back to issue them as one grid (k_trace_nq_batch + k_shade_tile_batch).
struct BatchCollector {
    std::vector<FrameItem> items;        // frames of the fused pipeline (1..7 light samples): k_trace_nq_batch + k_shade_tile_batch
    std::vector<FrameItem> items_pk;     // frames of the 8+-sample pipeline: k_closest_hit_nq_batch + k_shadow_pk_batch + k_shade_tile_batch
    uint32_t wl = 0, rows = 0;           // every held frame writes the same local width and row count: one grid fits all
    uint32_t max_lights = 0;
    bool accepts(uint32_t w, uint32_t r) { if (items.empty() && items_pk.empty()) { wl = w; rows = r; } return w == wl && r == rows; }
};

static int render_device_impl(srt_scene* s, const srt_params* p, void* stream_, int32_t* d_hit_id, float* d_t,
                              float* d_rgb_linear, uint8_t* d_rgb8, BatchCollector* bc = nullptr) {
    if (!s) return SRT_ERR_ARG;
    int rc = check_params(p);
    if (rc != SRT_OK) return rc;
    // every argument check comes before any state of the handle changes (counter sets, pending work)
    if ((p->flags & SRT_FLAG_SMOOTH_NORMALS) && (!s->dev.tri_normals || variant_of(p) == 1)) return SRT_ERR_ARG;   // needs vertex normals
    hipStream_t stream = (hipStream_t)stream_;
    HIP_TRY(hipSetDevice(s->device));
    const uint32_t rows = srt_rows_owned(p);
    const uint32_t wl = srt_cols_owned(p);    // width of the rows this call writes
    if (!rows) {                              // nothing to launch; work of an earlier render stays pending
        if (!s->pending) std::memset(&s->last, 0, sizeof(s->last));
        return SRT_OK;
    }
    std::memset(&s->last, 0, sizeof(s->last));
    s->last.rows = rows;
    s->last.primary_rays = pixels_owned(p);
    const size_t pixels = (size_t)wl * rows;
    // workspace for hit ids / t when the caller does not want them (the shade kernel does)
    if ((!d_hit_id || !d_t) && s->ws_pixels < pixels) {
        HIP_TRY(wait_idle(s));
        if (s->ws_hit) (void)hipFree(s->ws_hit);
        if (s->ws_t) (void)hipFree(s->ws_t);
        s->ws_hit = nullptr; s->ws_t = nullptr; s->ws_pixels = 0;
        HIP_TRY(hipMalloc((void**)&s->ws_hit, pixels * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void**)&s->ws_t, pixels * sizeof(float)));
        s->ws_pixels = pixels;
    }
    if (!d_hit_id) d_hit_id = s->ws_hit;
    if (!d_t) d_t = s->ws_t;
    if (p->n_lights > s->lights_cap) {
        HIP_TRY(wait_idle(s));
        if (s->d_lights) (void)hipFree(s->d_lights);
        if (s->h_lights) (void)hipHostFree(s->h_lights);
        s->d_lights = nullptr; s->h_lights = nullptr; s->lights_cap = 0;
        HIP_TRY(hipMalloc((void**)&s->d_lights, (size_t)p->n_lights * 3 * sizeof(float)));
        HIP_TRY(hipHostMalloc((void**)&s->h_lights, (size_t)p->n_lights * 3 * sizeof(float), hipHostMallocDefault));
        s->lights_cap = p->n_lights; s->lights_valid = 0;
    }
    const size_t light_bytes = (size_t)p->n_lights * 3 * sizeof(float);
    if (p->n_lights && !(s->lights_valid == p->n_lights && std::memcmp(s->h_lights, p->light_pos, light_bytes) == 0)) {
        HIP_TRY(wait_idle(s));     // staging buffer still in flight
        std::memcpy(s->h_lights, p->light_pos, light_bytes);
        HIP_TRY(hipMemcpyAsync(s->d_lights, s->h_lights, light_bytes, hipMemcpyHostToDevice, stream));
        s->lights_valid = p->n_lights;
    }
    // Work / hit counters: two sets used alternately.  The last kernel of a render zeroes the set the NEXT
    // render will use, so no memset or copy is enqueued per frame; srt_sync reads the last set.
    unsigned long long* ctr = s->d_counters + (s->render_seq & 1) * NCTR;
    unsigned long long* ctr_next = s->d_counters + ((s->render_seq + 1) & 1) * NCTR;
    // a render that failed half-way may have left either set dirty: clear both before the next one
    if (s->ctr_dirty) {
        HIP_TRY(hipMemsetAsync(s->d_counters, 0, 2 * NCTR * sizeof(unsigned long long), stream));
        HIP_TRY(hipMemsetAsync(s->d_qcount, 0, QL_COUNTERS * QL_STRIDE * sizeof(uint32_t), stream));
    }
    s->ctr_dirty = true;
    // a counting run must not inherit whatever replayed graphs left in the set (their frames use fixed sets)
    if (p->flags & SRT_FLAG_COUNT_WORK) HIP_TRY(hipMemsetAsync(ctr, 0, NCTR * sizeof(unsigned long long), stream));

    DevParams dp;
    dp.smooth = (p->flags & SRT_FLAG_SMOOTH_NORMALS) ? 1u : 0u;
    dp.shadow_px_major = 0u;
    dp.cam = p->ray_matrix ? 1u : 0u;
    for (int c = 0; c < 4; c++) for (int r3 = 0; r3 < 3; r3++) dp.cm[c * 3 + r3] = p->ray_matrix ? p->ray_matrix[c * 4 + r3] : 0.0f;
    dp.xcd_rows = (s->bytes > (32ull << 20) || variant_of(p) == 18) ? 1u : 0u;       // records far beyond one XCD's 4 MiB L2 (variant 18: forced, for the tests)
    dp.W = wl; dp.Wimg = p->width; dp.col_block = p->block_cols; dp.H = p->height; dp.rows = rows;
    dp.block_rows = p->block_rows; dp.block_first = p->block_first; dp.block_stride = p->block_stride;
    dp.i0 = (int)(-(float)p->width / 2); dp.j0 = (int)(-(float)p->height / 2);       // :511,513
    dp.sub_x = 0.0f; dp.sub_y = 0.0f;                                                 // rayXY = (0, 0), :507,514-515
    dp.focal = p->focal; dp.n_lights = p->n_lights; dp.lights = s->d_lights;
    dp.shadow_div = p->shadow_div; dp.reinhard = p->reinhard; dp.gamma = p->gamma;
    dp.bg = (uint32_t)p->background[0] | ((uint32_t)p->background[1] << 8) | ((uint32_t)p->background[2] << 16);

    const dim3 block(256), grid((wl + 15) / 16, (rows + 15) / 16);
    const dim3 grid8((wl + 7) / 8, (rows + 7) / 8);                // 8x8 pixels per workgroup: 4 waves x (4x4 pixels)
    const bool count = (p->flags & SRT_FLAG_COUNT_WORK) != 0;
    uint32_t variant = (p->flags >> 8) & 0xffu;            // experimental kernel selector (0 = shipped pipeline)
    const bool force_nq = variant == 24;                   // 24: what variant 0 does for a scene WITHOUT the packet preference (A/B on soups)
    const bool coarse_grid = variant == 27;                // 27: variant 0 with 2 x 2 tiles per workgroup in the unfused closest-hit launch (A/B, not shipped)
    // 40: the round-2 form everywhere (32 B node records, queue pushes in lane order); 41 / 42: the 64 B / the 32 B records in every
    // node-queue kernel (node-major order); 43: the shipped kernels with pushes in lane order.  Shipped (0): node-major order; 32 B records
    // in the fused and the closest-hit kernel, 64 B records in the stand-alone shadow kernel
    const bool all_narrow = variant == 40 || variant == 42, all_wide = variant == 41;
    dp.exp = ((variant == 40 || variant == 43) ? 1u : 0u) | (variant == 45 ? 2u : 0u) | (variant == 47 ? 4u : 0u) | (variant == 46 ? 8u : 0u) | ((variant == 28 || variant == 2) ? 32u : 0u);      // (28: k_trace_shade_nq has no shading launch behind it, 2: k_closest_hit_q counts itself: the hit statistic is not the shading kernel's)      // (47: diagnostic counters of the wide shadow kernel's steps)      // 45: the tile's root tests by the round-2 loop of dependent loads (A/B)
    dp.heavy_steps = 0u;                                   // (set below: frames a batch call holds back, and single frames that have the device to themselves)
    static const uint32_t heavy_default = [] { const char* e = std::getenv("SRT_HEAVY_STEPS"); return e ? (uint32_t)std::strtoul(e, nullptr, 10) : 64u; }();      // walks of this many node steps make a quadrant a heavy one (0 = off)
    static const uint32_t pk_units_default = [] { const char* e = std::getenv("SRT_PK_UNITS"); return e ? (uint32_t)std::strtoul(e, nullptr, 10) : 64u; }();
    static const uint32_t pk_take_default = [] { const char* e = std::getenv("SRT_PK_TAKE"); return e ? (uint32_t)std::strtoul(e, nullptr, 10) : 4u; }();
    const bool in_flight = (p->flags & SRT_FLAG_FRAMES_IN_FLIGHT) != 0;
    dp.pk_units = in_flight ? pk_units_default : 0u; dp.pk_take = in_flight && pk_take_default ? pk_take_default : 1u;      // (batch frames: set where they are held back)
    if (force_nq || variant == 25 || variant == 29 || variant == 35 || coarse_grid || (variant >= 40 && variant <= 62)) variant = 0;      // (44: the general shading kernel forced)            // 25: variant 0 with the packet shadow kernel reading records through LDS windows (A/B)
    const uint32_t spp = p->spp;
    // workspace of the tile pipeline: per 8x8 tile and light sample one 64-bit word of shadow bits
    // shadow bits: tile-major (one word per tile and light sample, node-queue kernels) or pixel-major (one word per pixel and 64 light
    // samples, packet shadow kernel): room for either
    const size_t words_tile = (size_t)grid8.x * grid8.y * (p->n_lights ? p->n_lights : 1), words_px = pixels * ((p->n_lights + 63) / 64);
    const size_t shadow_words = words_tile > words_px ? words_tile : words_px;
    if (variant != 1 && s->ws_shadow_words < shadow_words) {
        HIP_TRY(wait_idle(s));
        if (s->ws_shadow) (void)hipFree(s->ws_shadow);
        s->ws_shadow = nullptr; s->ws_shadow_words = 0;
        HIP_TRY(hipMalloc((void**)&s->ws_shadow, shadow_words * sizeof(unsigned long long)));
        s->ws_shadow_words = shadow_words;
    }
    const size_t n_tiles = (size_t)grid8.x * grid8.y;
    const uint32_t qcap_need = (uint32_t)((n_tiles + QL_SHARDS - 1) / QL_SHARDS) * 4u;      // a shard gets every 64th tile, four quadrants each
    if (variant != 1 && s->qcap < qcap_need) {
        HIP_TRY(wait_idle(s));
        if (s->ws_qlist) (void)hipFree(s->ws_qlist);
        s->ws_qlist = nullptr; s->qcap = 0;
        // two words per entry, and behind the lists one word per quadrant: the cost map the shadow kernel leaves for the next frame's list
        HIP_TRY(hipMalloc((void**)&s->ws_qlist, (size_t)QL_SHARDS * qcap_need * 3 * sizeof(uint32_t)));
        HIP_TRY(hipMemset(s->ws_qlist + (size_t)QL_SHARDS * qcap_need * 2, 0, (size_t)QL_SHARDS * qcap_need * sizeof(uint32_t)));
        s->qcap = qcap_need;
    }
    if (spp > 1 && s->ws_acc_pixels < pixels) {                    // supersampling extension: accumulation buffers
        HIP_TRY(wait_idle(s));
        if (s->ws_acc) (void)hipFree(s->ws_acc);
        if (s->ws_sub) (void)hipFree(s->ws_sub);
        if (s->ws_sub_hit) (void)hipFree(s->ws_sub_hit);
        if (s->ws_sub_t) (void)hipFree(s->ws_sub_t);
        s->ws_acc = nullptr; s->ws_sub = nullptr; s->ws_sub_hit = nullptr; s->ws_sub_t = nullptr; s->ws_acc_pixels = 0;
        HIP_TRY(hipMalloc((void**)&s->ws_acc, pixels * 3 * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&s->ws_sub, pixels * 3 * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&s->ws_sub_hit, pixels * sizeof(int32_t)));
        HIP_TRY(hipMalloc((void**)&s->ws_sub_t, pixels * sizeof(float)));
        s->ws_acc_pixels = pixels;
    }

    // One pass of the path over this call's pixels: closest hit (+ shadow rays) and shading.
    auto launch_frame = [&](const DevParams& fp_in, int32_t* o_hit, float* o_t, float* o_lin, uint8_t* o_rgb8,
                            unsigned long long* zero_next, hipEvent_t* ev) -> int {
        DevParams fp = fp_in;
        if (variant == 1) {                // v0 reference kernels: per-lane walk with inline triangle loop, per-pixel shade
            if (count) hipLaunchKernelGGL(k_closest_hit<true>, grid, block, 0, stream, s->dev, fp, o_hit, o_t, ctr);
            else       hipLaunchKernelGGL(k_closest_hit<false>, grid, block, 0, stream, s->dev, fp, o_hit, o_t, ctr);
            HIP_TRY(hipGetLastError());
            if (ev) { HIP_TRY(hipEventRecord(ev[1], stream)); HIP_TRY(hipEventRecord(ev[2], stream)); }
            if (count) hipLaunchKernelGGL(k_shade<true>, grid, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ctr, zero_next);
            else       hipLaunchKernelGGL(k_shade<false>, grid, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ctr, zero_next);
            HIP_TRY(hipGetLastError());
            std::snprintf(s->pipeline, sizeof(s->pipeline), "k_closest_hit+k_shade");
            return SRT_OK;
        }
        // closest-hit kernel: CAP = node queue entries, TWL/THL = log2 tile size per wave, FILTER = filtered slab test
        #define LAUNCH_NQ(CAP, TWL, THL, FILTER) do { \
            const dim3 g_((wl + (2u << TWL) - 1) / (2u << TWL), (rows + (2u << THL) - 1) / (2u << THL)); \
            if (count) hipLaunchKernelGGL((k_closest_hit_nq<true, CAP, TWL, THL, FILTER>), g_, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ctr, ql_cnt, ql, s->qcap); \
            else       hipLaunchKernelGGL((k_closest_hit_nq<false, CAP, TWL, THL, FILTER>), g_, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ctr, ql_cnt, ql, s->qcap); } while (0)
        // Pipelines (variant 0 picks per scene and light count; the numbered variants force one):
        //   fused        k_trace_nq (node-queue closest hit + shadow rays in one launch): 1..7 light samples
        //   nq + pk      node-queue closest hit, then the packet shadow kernel over the list of quadrants with hits: 8+ light samples
        //                (the samples of a pixel walk the other objects' trees in lock step), or variant 21 at any count
        //   pk + nq      packet closest hit, then the node-queue shadow kernel: hierarchies of heavily overlapping boxes (the 1 M soup:
        //                neighbouring primary rays test the same ~2,500 nodes, while the shadow rays of a tile start all over the
        //                scene), 1..7 light samples, or variant 23
        //   pk + pk      both packet kernels: such scenes with 8+ light samples, or variant 22
        //   nq chunked   the round-1 form for 8+ samples (k_shadow_nq, 64 rays in flight, samples cut over blockIdx.z): variant 20
        // camera mode (rays that do not start at the origin): closest hit on the packet kernel, which takes a general ray; the shadow
        // kernels start from the hit point either way
        // camera mode: the fused node-queue kernel has a build for rays with an origin (1..7 samples, variant 0); everything else
        // in camera mode goes through the packet closest-hit kernel, which takes a general ray
        const bool cam_nq = fp.cam && variant == 0 && !count && !s->rec->prefer_packet && p->n_lights >= 1 && p->n_lights < 8 && !fp.xcd_rows && (p->flags >> 8 & 0xffu) != 35;
        const bool pk_closest = (fp.cam && !cam_nq) || variant == 22 || variant == 23 || (variant == 0 && s->rec->prefer_packet && !force_nq);
        // 8 .. 15 samples on a scene whose rays test few nodes (expected slab tests per ray < 14: cube scenes 3-6, bunny over a slab 12):
        // the node-queue shadow kernel with the samples cut over blockIdx.z beats the packet walk since round 3's queue order (K3 with
        // 8 / 12 samples 0.212 / 0.281 ms against 0.254 / 0.321, cube over ground with 8: 0.107 against 0.197); the composite scene
        // (seven objects, tree crowns; estimate 17+) and 16+ samples stay with the packet walk (K4 with 8 / 12: 0.390 / 0.473 against 0.416 / 0.621)
        const bool few_nodes_mid = variant == 0 && !count && !fp.cam && p->n_lights >= 8 && p->n_lights < 16 && s->rec->overlap < 14.0 && !s->rec->prefer_packet &&
                                   (p->flags >> 8 & 0xffu) == 0;
        const bool pk_shadow = p->n_lights && (variant == 21 || variant == 22 || (variant == 0 && p->n_lights >= 8 && !few_nodes_mid));
        // A frame that has the device to itself (no in-flight hint, not part of a batch call): quadrants whose walks were long in this
        // handle's previous frame are dealt early (srt_kernels.h) -- nothing else fills the slots the launch's tail frees.  Same box:
        // K3 with 16 samples on one stream 4.04 -> 3.70 ms per 8 frames, K4 11.89 -> 11.72; with frames on four streams the lists LOSE
        // (K4 9.41 -> 9.64, the reference's main() scene 6.55 -> 6.91), so the hint turns them off.
        if (pk_shadow && variant == 0 && (p->flags >> 8 & 0xffu) == 0 && !count && !in_flight && !bc) fp.heavy_steps = heavy_default;
        uint32_t* const ql = pk_shadow ? s->ws_qlist : nullptr;      // the closest-hit kernel fills the quadrant list only for a consumer
        uint32_t* const ql_cnt = pk_shadow ? s->d_qcount : nullptr;
        const uint32_t L_CHUNK = p->n_lights / 4 > 4 ? (p->n_lights + 3) / 4 : 4;
        const bool chunked = p->n_lights >= 8 && !count && (variant == 20 || few_nodes_mid);
        const bool fused = (variant == 0 || variant > 10) && p->n_lights && !chunked && !pk_shadow && !pk_closest && variant != 20;     // (variants 11, 17, 18 are configurations of the fused kernel)
        const dim3 grid8x(grid8.x, fp.xcd_rows ? (grid8.y + 7) / 8 * 8 : grid8.y);      // whole tile rows per XCD: y padded to 8 rows
        bool shaded = false;               // the trace launch shaded its tiles itself
        switch (variant) {
        case 2:                        // one ray per lane + triangle queue
            if (count) hipLaunchKernelGGL((k_closest_hit_q<true>), grid, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ctr);
            else       hipLaunchKernelGGL((k_closest_hit_q<false>), grid, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ctr);
            break;
        case 3: LAUNCH_NQ(160, 2, 2, false); break;      // tiny node queue: exercises the stackless overflow path
        case 4: LAUNCH_NQ(512, 2, 2, false); break;      // shipped geometry with exact divides only
        case 6: LAUNCH_NQ(160, 2, 2, true); break;       // tiny node queue + filtered slab test: the overflow walk as shipped
        case 5: LAUNCH_NQ(1024, 3, 2, false); break;     // 8x4 pixels per wave, 1024-entry queue (tile-size experiment)
        case 10: LAUNCH_NQ(512, 2, 2, true); break;      // shipped kernels, unfused (closest hit, then shadow)
        default:
            if (pk_closest) {          // one wavefront per 8x8 tile walks the trees in lock step; a workgroup = 2 x 2 tiles
                const dim3 gp((grid8.x + 1) / 2, fp.xcd_rows ? ((grid8.y + 1) / 2 + 7) / 8 * 8 : (grid8.y + 1) / 2);
                if (fp.cam && count)  hipLaunchKernelGGL((k_closest_hit_pk<true, true, false, true>), dim3((grid8.x + 1) / 2, (grid8.y + 1) / 2), block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ql_cnt, ql, s->qcap, ctr);
                else if (fp.cam)      hipLaunchKernelGGL((k_closest_hit_pk<false, true, false, true>), dim3((grid8.x + 1) / 2, (grid8.y + 1) / 2), block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ql_cnt, ql, s->qcap, ctr);
                else if (count)       hipLaunchKernelGGL((k_closest_hit_pk<true, true, false>), dim3((grid8.x + 1) / 2, (grid8.y + 1) / 2), block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ql_cnt, ql, s->qcap, ctr);
                else if (fp.xcd_rows) hipLaunchKernelGGL((k_closest_hit_pk<false, true, true>), gp, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ql_cnt, ql, s->qcap, ctr);
                else                  hipLaunchKernelGGL((k_closest_hit_pk<false, true, false>), gp, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ql_cnt, ql, s->qcap, ctr);
            } else if (fused && bc && !count && (p->flags >> 8 & 0xffu) == 0 && !fp.xcd_rows && !fp.cam && spp == 1 && bc->accepts(wl, rows)) {
                // held back: the batch call launches this frame together with the others (same kernels, same arguments)
                bc->items.push_back(FrameItem{s->dev, fp, o_hit, o_t, o_lin, o_rgb8, s->ws_shadow, ctr, zero_next, s->d_qcount});
                std::snprintf(s->pipeline, sizeof(s->pipeline), "k_trace_nq+k_shade_tile (batched)");
                return SRT_OK;
            } else if (fused && variant == 28 && !count && !fp.xcd_rows && p->n_lights < 64) {      // 28 (A/B): the whole frame in one launch
                hipLaunchKernelGGL((k_trace_shade_nq<512, true, 6, 16>), grid8, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ctr, zero_next, s->d_qcount);
                shaded = true;
            } else if (fused && cam_nq) {      // camera mode on the node queues
                hipLaunchKernelGGL((k_trace_nq<false, 512, true, 5, 16, false, false, true>), grid8, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, s->ws_shadow, ctr);
            } else if (fused) {        // closest hit + shadow rays in one launch
                if (count)              hipLaunchKernelGGL((k_trace_nq<true, 512, true, 5, 16>), grid8, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, s->ws_shadow, ctr);
                else if (variant == 11) hipLaunchKernelGGL((k_trace_nq<false, 512, true, 5, 16>), grid8, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, s->ws_shadow, ctr);
                else if (variant == 12) hipLaunchKernelGGL((k_trace_nq<false, 512, true, 6, 16, false, true>), grid8, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, s->ws_shadow, ctr);      // round-1 form: roots re-tested per wave (A/B)
                else if (variant == 17) hipLaunchKernelGGL((k_trace_nq<false, 512, true, 5, 64>), grid8, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, s->ws_shadow, ctr);      // 64 shadow rays in flight per wave
                else if ((p->flags >> 8 & 0xffu) == 54) hipLaunchKernelGGL((k_trace_nq<false, 512, true, 6, 16>), grid8, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, s->ws_shadow, ctr);      // (A/B) the build for 6 waves per SIMD (rounds 1-2)
                else if (all_wide)      hipLaunchKernelGGL((k_trace_nq<false, 512, true, 6, 16, false, false, false, true>), grid8, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, s->ws_shadow, ctr);
                else if (fp.xcd_rows)   hipLaunchKernelGGL((k_trace_nq<false, 512, true, 6, 16, true>), grid8x, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, s->ws_shadow, ctr);
                else                    hipLaunchKernelGGL((k_trace_nq<false, 512, true, 7, 16>), grid8, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, s->ws_shadow, ctr);      // 72 VGPRs (14 spills), 7 waves per SIMD: since the node-major order 3 % ahead of the 6-wave build (80 VGPRs, 5 spills); round 2's kernel lost 2.5 % that way
            } else if (bc && !count && (p->flags >> 8 & 0xffu) == 0 && pk_shadow && !pk_closest && spp == 1 && bc->accepts(wl, rows)) {
                // 8+ light samples, held back: node-queue closest hit, packet shadow kernel and shading of the batch's frames in three launches
                FrameItem it{s->dev, fp, o_hit, o_t, o_lin, o_rgb8, s->ws_shadow, ctr, zero_next, s->d_qcount, s->ws_qlist, s->qcap, 0u};
                it.p.shadow_px_major = 1u;
                // packet shadow walks of this many node steps put their quadrant on the heavy list of the next frame (srt_kernels.h).  Only
                // for the frames of a batch: their shadow rays are ONE launch and its tail is idle machine (a K4 step of eight share-frames
                // 2.24 -> 1.73 ms).  Frames launched one by one on several streams pipeline -- the next frame's closest-hit launch fills the
                // slots a tail frees -- and lose that when the launch ends abruptly (K4 on four streams 10.1 -> 10.5-11.1 ms per 8 frames,
                // K3 with 16 samples 3.38 -> 3.6-4.0; one stream: no difference): k_shadow_pk is built without the heavy lists.
                it.p.heavy_steps = heavy_default;
                it.p.pk_take = pk_take_default ? pk_take_default : 1u;      // unit numbers four at a time: fewer same-address atomics (srt_packet.h)
                it.p.pk_units = 0u;                   // the frames of a batch share the machine: every frame keeps its part of the grid (K3 with 16 samples, an eighth: 0.45 ms per step against 0.68 with surplus waves leaving)
                bc->items_pk.push_back(it);
                if (p->n_lights > bc->max_lights) bc->max_lights = p->n_lights;
                std::snprintf(s->pipeline, sizeof(s->pipeline), "k_closest_hit_nq+k_shadow_pk+k_shade_tile (batched)");
                return SRT_OK;
            } else if (!count && coarse_grid) {
                // 2 x 2 tiles per workgroup (a quarter of the workgroups for frames that are mostly background).  Measured and NOT
                // shipped: K4 closest hit 0.44 ms against 0.29 with one tile per workgroup, K3 0.22 against 0.10 -- the launch is not
                // dispatch-bound, and a workgroup that walks its live tiles one after the other is a longer tail
                hipLaunchKernelGGL((k_closest_hit_nq<false, 512, 2, 2, true, true>), dim3((grid8.x + 1) / 2, (grid8.y + 1) / 2), block, 0, stream,
                                   s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ctr, ql_cnt, ql, s->qcap);
            } else if (all_wide && !count) {
                hipLaunchKernelGGL((k_closest_hit_nq<false, 512, 2, 2, true, false, true>), grid8, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ctr, ql_cnt, ql, s->qcap);
            } else if (!count && (p->flags >> 8 & 0xffu) != 53) {
                // built for 7 waves per SIMD (72 VGPRs, 5 spills; 76 without the bound: 6 waves): a frame of mostly background tiles is a stream
                // of short workgroups, and one more resident per SIMD is worth the spills -- K4 closest hit 0.234 -> 0.223 ms, frame 1.285 -> 1.236
                // (variant 53 = the unbounded build, A/B)
                hipLaunchKernelGGL((k_closest_hit_nq<false, 512, 2, 2, true, false, false, 7>), grid8, block, 0, stream, s->dev, fp, o_hit, o_t, o_lin, o_rgb8, ctr, ql_cnt, ql, s->qcap);
            } else {
                LAUNCH_NQ(512, 2, 2, true);
            }
            break;
        }
        #undef LAUNCH_NQ
        HIP_TRY(hipGetLastError());
        if (ev) HIP_TRY(hipEventRecord(ev[1], stream));
        if (pk_shadow) {
            // a fixed number of waves pull units (the shadow rays of 64 / n_lights pixels) from the quadrant list: no grid over the image
            const uint64_t max_units = (uint64_t)n_tiles * 4u * 2u * ((p->n_lights + 7) / 8);
            const uint32_t wgs = (uint32_t)(max_units / 4 + 1 < (uint64_t)s->n_cu * 8 ? max_units / 4 + 1 : (uint64_t)s->n_cu * 8);      // (8 per CU: every wave slot; fewer was measured)
            if (count)                                hipLaunchKernelGGL((k_shadow_pk<true, true, false>), dim3(wgs), block, 0, stream, s->dev, fp, o_hit, o_t, s->d_qcount, s->ws_qlist, s->qcap, s->ws_shadow, ctr);
            else if ((p->flags >> 8 & 0xffu) == 29)   hipLaunchKernelGGL((k_shadow_pk<false, true, false, true>), dim3(wgs), block, 0, stream, s->dev, fp, o_hit, o_t, s->d_qcount, s->ws_qlist, s->qcap, s->ws_shadow, ctr);    // units in entry order (A/B)
            else if ((p->flags >> 8 & 0xffu) == 55)   hipLaunchKernelGGL((k_shadow_pk<false, true, false, false, 1>), dim3(wgs), block, 0, stream, s->dev, fp, o_hit, o_t, s->d_qcount, s->ws_qlist, s->qcap, s->ws_shadow, ctr);    // record of i + 1 requested ahead (A/B)
            else if ((p->flags >> 8 & 0xffu) == 56)   hipLaunchKernelGGL((k_shadow_pk<false, true, false, false, 2>), dim3(wgs), block, 0, stream, s->dev, fp, o_hit, o_t, s->d_qcount, s->ws_qlist, s->qcap, s->ws_shadow, ctr);    // + skip[i] / first triangle (A/B)
            else if (fp.heavy_steps)                  hipLaunchKernelGGL((k_shadow_pk<false, true, false, false, 0, true>), dim3(wgs), block, 0, stream, s->dev, fp, o_hit, o_t, s->d_qcount, s->ws_qlist, s->qcap, s->ws_shadow, ctr);    // heavy quadrants dealt early (a frame alone on the device)
            else if ((p->flags >> 8 & 0xffu) == 58)   hipLaunchKernelGGL((k_shadow_pk<false, true, false, false, -1>), dim3(wgs), block, 0, stream, s->dev, fp, o_hit, o_t, s->d_qcount, s->ws_qlist, s->qcap, s->ws_shadow, ctr);    // the shipped walk with wave clocks (SRT_DIAG_COUNTERS)
            else if ((p->flags >> 8 & 0xffu) == 57)   hipLaunchKernelGGL((k_shadow_pk<false, true, false, false, 0>), dim3(wgs), block, 0, stream, s->dev, fp, o_hit, o_t, s->d_qcount, s->ws_qlist, s->qcap, s->ws_shadow, ctr);    // the plain walk (A/B)
            else if ((p->flags >> 8 & 0xffu) == 25)   hipLaunchKernelGGL((k_shadow_pk<false, true, true>), dim3(wgs), block, 0, stream, s->dev, fp, o_hit, o_t, s->d_qcount, s->ws_qlist, s->qcap, s->ws_shadow, ctr);    // records through LDS windows (A/B)
            else                                      hipLaunchKernelGGL((k_shadow_pk<false, true, false>), dim3(wgs), block, 0, stream, s->dev, fp, o_hit, o_t, s->d_qcount, s->ws_qlist, s->qcap, s->ws_shadow, ctr);
            HIP_TRY(hipGetLastError());
        } else if (p->n_lights && !fused) {
            // scenes far bigger than an L2 (fp.xcd_rows): whole tile rows per XCD for the shadow rays too (variant 62 = the plain order, A/B)
            DevParams sq = fp;
            if ((p->flags >> 8 & 0xffu) == 62) sq.xcd_rows = 0u;
            const dim3 gq(grid8.x, sq.xcd_rows ? (grid8.y + 7) / 8 * 8 : grid8.y);
            if (chunked)           hipLaunchKernelGGL((k_shadow_nq<false, 512, true, 64, 6>), dim3(gq.x, gq.y, (p->n_lights + L_CHUNK - 1) / L_CHUNK), block, 0, stream,
                                                      s->dev, sq, o_hit, o_t, s->ws_shadow, ctr, L_CHUNK);
            else if (count)        hipLaunchKernelGGL((k_shadow_nq<true, 512, false>), gq, block, 0, stream, s->dev, sq, o_hit, o_t, s->ws_shadow, ctr);
            else if (variant == 3) hipLaunchKernelGGL((k_shadow_nq<false, 160, false>), gq, block, 0, stream, s->dev, sq, o_hit, o_t, s->ws_shadow, ctr);
            else if (variant == 6) hipLaunchKernelGGL((k_shadow_nq<false, 160, true>), gq, block, 0, stream, s->dev, sq, o_hit, o_t, s->ws_shadow, ctr);
            else if (variant == 4) hipLaunchKernelGGL((k_shadow_nq<false, 512, false>), gq, block, 0, stream, s->dev, sq, o_hit, o_t, s->ws_shadow, ctr);
            else if (all_narrow)   hipLaunchKernelGGL((k_shadow_nq<false, 512, true, 16, 6, false>), gq, block, 0, stream, s->dev, sq, o_hit, o_t, s->ws_shadow, ctr);
            else                   hipLaunchKernelGGL((k_shadow_nq<false, 512, true, 16, 6>), gq, block, 0, stream, s->dev, sq, o_hit, o_t, s->ws_shadow, ctr);      // 80 VGPRs: six waves per SIMD (86 without the bound: five)
            HIP_TRY(hipGetLastError());
        }
        if (ev) HIP_TRY(hipEventRecord(ev[2], stream));
        std::snprintf(s->pipeline, sizeof(s->pipeline), "%s%s+k_shade_tile", fused ? "k_trace_nq" : (pk_closest ? "k_closest_hit_pk" : "k_closest_hit_nq"),
                      fused || !p->n_lights ? "" : (pk_shadow ? "+k_shadow_pk" : "+k_shadow_nq"));
        if (shaded) { std::snprintf(s->pipeline, sizeof(s->pipeline), "k_trace_shade_nq"); return SRT_OK; }
        DevParams sp = fp;
        sp.shadow_px_major = pk_shadow ? 1u : 0u;
        if (s->rec->int_shin && variant_of(p) != 44) hipLaunchKernelGGL(k_shade_tile<1>, grid, block, 0, stream, s->dev, sp, o_hit, o_t, s->ws_shadow, o_lin, o_rgb8, zero_next, s->d_qcount, ctr);
        else                                         hipLaunchKernelGGL(k_shade_tile<0>, grid, block, 0, stream, s->dev, sp, o_hit, o_t, s->ws_shadow, o_lin, o_rgb8, zero_next, s->d_qcount, ctr);      // (44: the general form forced, A/B)
        HIP_TRY(hipGetLastError());
        return SRT_OK;
    };

    // SRT_FLAG_NO_TIMING: no event records (a caller capturing the launches into a hipGraph)
    hipEvent_t* ev = ((p->flags & SRT_FLAG_NO_TIMING) || bc) ? nullptr : s->ev[s->ring_count % RING];      // (a batch has no per-frame times)
    if (ev && !ev[0]) for (int i = 0; i < 4; i++) HIP_TRY(hipEventCreate(&ev[i]));      // a ring slot's events are made on first use
    if (ev) HIP_TRY(hipEventRecord(ev[0], stream));
    if (spp == 1) {
        rc = launch_frame(dp, d_hit_id, d_t, d_rgb_linear, d_rgb8, ctr_next, ev);
        if (rc != SRT_OK) return rc;
    } else {
        // Supersampling (extension): n x n regular sub-pixel grid, offsets (k+0.5)/n - 0.5 added to
        // dir.xy; the pre-tone-map sums are added in sub-sample order, divided by spp, then tone-mapped once.
        // hit_id / t report sub-sample 0.  Per-kernel times are those of the last sub-frame.
        const uint32_t n = (uint32_t)std::lround(std::sqrt((double)spp));
        const uint32_t gq = (uint32_t)((pixels * 3 + 255) / 256);
        for (uint32_t k = 0; k < spp; k++) {
            DevParams fp = dp;
            fp.sub_x = ((float)(k % n) + 0.5f) / (float)n - 0.5f;
            fp.sub_y = ((float)(k / n) + 0.5f) / (float)n - 0.5f;
            rc = launch_frame(fp, k == 0 ? d_hit_id : s->ws_sub_hit, k == 0 ? d_t : s->ws_sub_t, s->ws_sub, nullptr, nullptr,
                              (k == spp - 1) ? ev : nullptr);
            if (rc != SRT_OK) return rc;
            hipLaunchKernelGGL(k_accumulate, dim3(gq), block, 0, stream, s->ws_acc, s->ws_sub, (uint32_t)(pixels * 3), k == 0 ? 1 : 0);
            HIP_TRY(hipGetLastError());
        }
        hipLaunchKernelGGL(k_resolve, dim3((uint32_t)((pixels + 255) / 256)), block, 0, stream, dp, s->ws_acc, (float)spp, (uint32_t)pixels,
                           d_rgb_linear, d_rgb8, ctr_next);
        HIP_TRY(hipGetLastError());
    }
    if (ev) {
        HIP_TRY(hipEventRecord(ev[3], stream));
        s->last_done = ev[3];
        s->ring_count++;
    } else {
        s->last_done = nullptr;
    }
    s->render_seq++;                          // the sets only swap once every launch of this render is enqueued
    s->d_ctr_last = ctr;
    s->ctr_dirty = false;
    s->last_stream = stream;
    s->pending = true;
    s->last.primary_rays = pixels_owned(p) * spp;
    s->last.shadow_rays = p->n_lights;     // multiplied by hit count in srt_sync
    return SRT_OK;
}

int srt_render_device(srt_scene* s, const srt_params* p, void* stream, int32_t* d_hit_id, float* d_t, float* d_rgb_linear, uint8_t* d_rgb8) {
    return guarded([&] { return render_device_impl(s, p, stream, d_hit_id, d_t, d_rgb_linear, d_rgb8); });
}

// The frames of a step in as few launches as their arguments fit (srt.h).  Every frame goes through render_device_impl -- the same
// checks, workspaces, light upload and counter sets as a single render; frames that take the fused pipeline at one size are held back
// and launched together, the others (another pipeline, counting build, supersampling, a different size) are launched as they come.
static int render_device_batch_impl(uint32_t n, srt_scene* const* scenes, const srt_params* params, void* stream_,
                                    int32_t* const* d_hit_id, float* const* d_t, float* const* d_rgb_linear, uint8_t* const* d_rgb8) {
    if (!n) return SRT_OK;
    if (!scenes || !params) return SRT_ERR_ARG;
    if (n > 65535u) return SRT_ERR_LIMIT;                     // the frame is a grid dimension
    for (uint32_t i = 0; i < n; i++) {
        if (!scenes[i] || scenes[i]->device != scenes[0]->device) return SRT_ERR_ARG;
        for (uint32_t k = 0; k < i; k++) if (scenes[k] == scenes[i]) return SRT_ERR_ARG;      // a handle's workspace serves one frame at a time
        const int rc = check_params(&params[i]);                                               // nothing is enqueued if any frame is malformed
        if (rc != SRT_OK) return rc;
        if ((params[i].flags & SRT_FLAG_SMOOTH_NORMALS) && (!scenes[i]->dev.tri_normals || variant_of(&params[i]) == 1)) return SRT_ERR_ARG;
    }
    hipStream_t stream = (hipStream_t)stream_;
    bool batch_int_shin = true;                               // the specialised shading kernel only if every scene of the batch qualifies
    for (uint32_t i = 0; i < n; i++) batch_int_shin = batch_int_shin && scenes[i]->rec->int_shin;
    BatchCollector bc;
    for (uint32_t i = 0; i < n; i++) {
        const int rc = render_device_impl(scenes[i], &params[i], stream_, d_hit_id ? d_hit_id[i] : nullptr, d_t ? d_t[i] : nullptr,
                                          d_rgb_linear ? d_rgb_linear[i] : nullptr, d_rgb8 ? d_rgb8[i] : nullptr, &bc);
        if (rc != SRT_OK) { bc.items.clear(); bc.items_pk.clear(); for (uint32_t k = 0; k <= i; k++) scenes[k]->ctr_dirty = true; return rc; }   // held frames are dropped: their counter sets may be half-used
    }
    // the held frames, FRAME_TAB_MAX at a time: their arguments by value in the launch (srt_kernels.h: FrameTab)
    int rc = SRT_OK;
    const dim3 block(256);
    for (size_t first = 0; first < bc.items_pk.size() && rc == SRT_OK; first += FRAME_TAB_MAX) {
        const uint32_t held_pk = (uint32_t)std::min<size_t>(FRAME_TAB_MAX, bc.items_pk.size() - first);
        FrameTab tab;
        std::memset(&tab, 0, sizeof(tab));
        std::memcpy(tab.it, bc.items_pk.data() + first, held_pk * sizeof(FrameItem));
        const dim3 g8((bc.wl + 7) / 8, (bc.rows + 7) / 8, held_pk), g16((bc.wl + 15) / 16, (bc.rows + 15) / 16, held_pk);
        const uint64_t n_tiles = (uint64_t)g8.x * g8.y, max_units = n_tiles * 4u * 2u * ((bc.max_lights + 7) / 8);
        const uint64_t wgs_all = (uint64_t)scenes[0]->n_cu * 8;         // the chip's worth of waves, shared by the frames
        uint32_t wgs = (uint32_t)((wgs_all + held_pk - 1) / held_pk);
        if ((uint64_t)wgs > max_units / 4 + 1) wgs = (uint32_t)(max_units / 4 + 1);
        // grid = (tiles per row, frames, tile rows): the same tile row of all the frames is in flight together (srt_kernels.h)
        hipLaunchKernelGGL((k_closest_hit_nq_batch<512, true, true>), dim3(g8.x, held_pk, g8.y), block, 0, stream, tab);
        hipLaunchKernelGGL((k_shadow_pk_batch<true>), dim3(wgs, held_pk), block, 0, stream, tab);
        if (batch_int_shin) hipLaunchKernelGGL(k_shade_tile_batch<1>, g16, block, 0, stream, tab);
        else                hipLaunchKernelGGL(k_shade_tile_batch<0>, g16, block, 0, stream, tab);
        if (hipGetLastError() != hipSuccess) rc = SRT_ERR_DEVICE;
    }
    for (size_t first = 0; first < bc.items.size() && rc == SRT_OK; first += FRAME_TAB_MAX) {
        const uint32_t held = (uint32_t)std::min<size_t>(FRAME_TAB_MAX, bc.items.size() - first);
        FrameTab tab;
        std::memset(&tab, 0, sizeof(tab));
        std::memcpy(tab.it, bc.items.data() + first, held * sizeof(FrameItem));
        const dim3 g_trace((bc.wl + 7) / 8, (bc.rows + 7) / 8, held), g_shade((bc.wl + 15) / 16, (bc.rows + 15) / 16, held);
        hipLaunchKernelGGL((k_trace_nq_batch<512, true, 7, 16, true>), dim3(g_trace.x, held, g_trace.y), block, 0, stream, tab);
        if (batch_int_shin) hipLaunchKernelGGL(k_shade_tile_batch<1>, g_shade, block, 0, stream, tab);
        else                hipLaunchKernelGGL(k_shade_tile_batch<0>, g_shade, block, 0, stream, tab);
        if (hipGetLastError() != hipSuccess) rc = SRT_ERR_DEVICE;
    }
    if (rc != SRT_OK) for (uint32_t k = 0; k < n; k++) scenes[k]->ctr_dirty = true;       // the set the shading would have zeroed
    return rc;
}

int srt_render_device_batch(uint32_t n, srt_scene* const* scenes, const srt_params* params, void* stream,
                            int32_t* const* d_hit_id, float* const* d_t, float* const* d_rgb_linear, uint8_t* const* d_rgb8) {
    return guarded([&] { return render_device_batch_impl(n, scenes, params, stream, d_hit_id, d_t, d_rgb_linear, d_rgb8); });
}

int srt_sync(srt_scene* s, srt_stats* stats) {
    if (!s) return SRT_ERR_ARG;
    if (s->pending) {
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipStreamSynchronize(s->last_stream));
        const uint32_t n = s->ring_count < RING ? s->ring_count : RING;
        double a = 0., b = 0., c = 0., sh = 0.;
        if (n == 0) { s->last.ms_primary = s->last.ms_shadow = s->last.ms_shade = s->last.ms_total = 0.f; }
        for (uint32_t k = 0; k < n; k++) {
            hipEvent_t* ev = s->ev[(s->ring_count - 1 - k) % RING];
            float x = 0.f, y = 0.f, z = 0.f, w = 0.f;
            HIP_TRY(hipEventElapsedTime(&x, ev[0], ev[1]));
            HIP_TRY(hipEventElapsedTime(&y, ev[1], ev[2]));
            HIP_TRY(hipEventElapsedTime(&w, ev[2], ev[3]));
            HIP_TRY(hipEventElapsedTime(&z, ev[0], ev[3]));
            a += x; b += y; c += z; sh += w;
        }
        if (n) {
            s->last.ms_primary = (float)(a / n); s->last.ms_shadow = (float)(b / n); s->last.ms_shade = (float)(sh / n);
            s->last.ms_total = (float)(c / n);
        }
        s->last.launches = n;
        s->ring_count = 0;
        HIP_TRY(hipMemcpy(s->h_counters, s->d_ctr_last, NCTR * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long hits = 0;
        for (int k = 0; k < 64; k++) hits += s->h_counters[8 + 8 * k];
        s->last.hit_rays = hits;
        s->last.shadow_rays = s->last.shadow_rays * hits;
        s->last.node_tests_primary = s->h_counters[1];
        s->last.tri_tests_primary = s->h_counters[2];
        s->last.node_tests_shadow = s->h_counters[3];
        s->last.tri_tests_shadow = s->h_counters[4];
        if (std::getenv("SRT_DIAG_COUNTERS"))      // counting build of the packet shadow kernel: shape of its walks
            std::fprintf(stderr, "srt diag: walks %llu steps %llu node-window loads %llu triangle iterations %llu | lane tests: nodes %llu tris %llu\n",
                         s->h_counters[0], s->h_counters[5], s->h_counters[6], s->h_counters[7], s->h_counters[3], s->h_counters[4]);
        if (std::getenv("SRT_DIAG_COUNTERS") && s->h_counters[17]) {
            const double span = (double)(s->h_counters[10] - ((1ull << 62) - s->h_counters[11])) * 0.01;      // us
            std::fprintf(stderr, "srt diag: packet shadow kernel: %llu waves, first start to last end %.1f us, mean wave busy %.1f us (%.0f %%), longest walk %.1f us, "
                                 "most steps in a walk %llu, walks of > 256 steps %llu (mean %.1f us)\n",
                         s->h_counters[17], span, (double)s->h_counters[9] * 0.01 / (double)s->h_counters[17],
                         span > 0 ? 100.0 * (double)s->h_counters[9] * 0.01 / (double)s->h_counters[17] / span : 0.0, (double)s->h_counters[12] * 0.01,
                         s->h_counters[13], s->h_counters[14], s->h_counters[14] ? (double)s->h_counters[15] * 0.01 / (double)s->h_counters[14] : 0.0);
            std::fprintf(stderr, "srt diag: walks of 100 us and more: %llu\n", s->h_counters[18]);
        }
        s->pending = false;
    }
    if (stats) *stats = s->last;
    return SRT_OK;
}

// The scene's own stream: srt_render / srt_render_async / srt_scene_update(stream = NULL) are ordered on it.
static int own_stream(srt_scene* s, hipStream_t* out) {
    if (!s->stream) HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    *out = s->stream;
    return SRT_OK;
}

static int render_async_impl(srt_scene* s, const srt_params* p, int32_t* hit_id, float* t, float* rgb_linear, uint8_t* rgb8, bool wait, srt_stats* stats) {
    if (!s) return SRT_ERR_ARG;
    int rc = check_params(p);
    if (rc != SRT_OK) return rc;
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t st;
    rc = own_stream(s, &st);
    if (rc != SRT_OK) return rc;
    const uint32_t rows = srt_rows_owned(p);
    const size_t pixels = (size_t)srt_cols_owned(p) * rows;
    if (pixels > s->ws_out_pixels) {
        HIP_TRY(wait_idle(s));
        if (s->ws_lin) (void)hipFree(s->ws_lin);
        if (s->ws_rgb8) (void)hipFree(s->ws_rgb8);
        s->ws_lin = nullptr; s->ws_rgb8 = nullptr; s->ws_out_pixels = 0;
        HIP_TRY(hipMalloc((void**)&s->ws_lin, pixels * 3 * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&s->ws_rgb8, pixels * 3));
        s->ws_out_pixels = pixels;
    }
    rc = render_device_impl(s, p, st, nullptr, nullptr, rgb_linear ? s->ws_lin : nullptr, rgb8 ? s->ws_rgb8 : nullptr);
    if (rc != SRT_OK) return rc;
    if (wait) {        // srt_render: the caller's buffers are ordinary (pageable) memory as a rule, where a synchronous copy is the fast one
        rc = srt_sync(s, stats);
        if (rc != SRT_OK) return rc;
        if (pixels) {
            if (hit_id) HIP_TRY(hipMemcpy(hit_id, s->ws_hit, pixels * sizeof(int32_t), hipMemcpyDeviceToHost));
            if (t) HIP_TRY(hipMemcpy(t, s->ws_t, pixels * sizeof(float), hipMemcpyDeviceToHost));
            if (rgb_linear) HIP_TRY(hipMemcpy(rgb_linear, s->ws_lin, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost));
            if (rgb8) HIP_TRY(hipMemcpy(rgb8, s->ws_rgb8, pixels * 3, hipMemcpyDeviceToHost));
        }
        return SRT_OK;
    }
    if (pixels) {      // device -> host behind the kernels; truly asynchronous into pinned memory (srt_host_alloc)
        if (hit_id) HIP_TRY(hipMemcpyAsync(hit_id, s->ws_hit, pixels * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (t) HIP_TRY(hipMemcpyAsync(t, s->ws_t, pixels * sizeof(float), hipMemcpyDeviceToHost, st));
        if (rgb_linear) HIP_TRY(hipMemcpyAsync(rgb_linear, s->ws_lin, pixels * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
        if (rgb8) HIP_TRY(hipMemcpyAsync(rgb8, s->ws_rgb8, pixels * 3, hipMemcpyDeviceToHost, st));
    }
    return SRT_OK;
}

int srt_render_async(srt_scene* s, const srt_params* p, int32_t* hit_id, float* t, float* rgb_linear, uint8_t* rgb8) {
    return guarded([&] { return render_async_impl(s, p, hit_id, t, rgb_linear, rgb8, false, nullptr); });
}

int srt_render(srt_scene* s, const srt_params* p, int32_t* hit_id, float* t, float* rgb_linear, uint8_t* rgb8, srt_stats* stats) {
    return guarded([&] { return render_async_impl(s, p, hit_id, t, rgb_linear, rgb8, true, stats); });
}

void* srt_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void srt_host_free(void* p) { if (p) (void)hipHostFree(p); }

// ---- known-answer entry points (device leaf functions on caller vectors; host pointers in and out) ----
namespace {
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { hipError_t e = hipMalloc(&p, bytes ? bytes : 1); if (e != hipSuccess) { g_last_hip = (int)e; return SRT_ERR_DEVICE; } return SRT_OK; }
    int up(const void* h, size_t bytes) { int rc = alloc(bytes); if (rc) return rc; HIP_TRY(hipMemcpy(p, h, bytes, hipMemcpyHostToDevice)); return SRT_OK; }
    int down(void* h, size_t bytes) { HIP_TRY(hipMemcpy(h, p, bytes, hipMemcpyDeviceToHost)); return SRT_OK; }
};
}
#define KAT_TRY(expr) do { int rc_ = (expr); if (rc_ != SRT_OK) return rc_; } while (0)

int srt_kat_ray_aabb(int device, uint32_t n, const float* ray_od, const float* box, uint8_t* exact, uint8_t* branchless,
                     uint8_t* filtered, uint8_t* ambiguous) {
    if (!n || !ray_od || !box || !exact || !branchless || !filtered || !ambiguous) return SRT_ERR_ARG;
    HIP_TRY(hipSetDevice(device));
    DevBuf r, b, o0, o1, o2, o3;
    KAT_TRY(r.up(ray_od, (size_t)n * 24)); KAT_TRY(b.up(box, (size_t)n * 24));
    KAT_TRY(o0.alloc(n)); KAT_TRY(o1.alloc(n)); KAT_TRY(o2.alloc(n)); KAT_TRY(o3.alloc(n));
    hipLaunchKernelGGL(k_kat_ray_aabb, dim3((n + 255) / 256), dim3(256), 0, 0, n, (const float*)r.p, (const float*)b.p,
                       (uint8_t*)o0.p, (uint8_t*)o1.p, (uint8_t*)o2.p, (uint8_t*)o3.p);
    HIP_TRY(hipGetLastError()); HIP_TRY(hipDeviceSynchronize());
    KAT_TRY(o0.down(exact, n)); KAT_TRY(o1.down(branchless, n)); KAT_TRY(o2.down(filtered, n)); KAT_TRY(o3.down(ambiguous, n));
    return SRT_OK;
}

int srt_kat_ray_triangle(int device, uint32_t n, const float* ray_od, const float* tri_points, float* t) {
    if (!n || !ray_od || !tri_points || !t) return SRT_ERR_ARG;
    return guarded([&]() -> int {
    HIP_TRY(hipSetDevice(device));
    std::vector<DevTri> tris(n);
    for (uint32_t i = 0; i < n; i++) tris[i] = derive_triangle(tri_points + 12 * (size_t)i);
    DevBuf r, q, o;
    KAT_TRY(r.up(ray_od, (size_t)n * 24)); KAT_TRY(q.up(tris.data(), (size_t)n * sizeof(DevTri))); KAT_TRY(o.alloc((size_t)n * 4));
    hipLaunchKernelGGL(k_kat_ray_triangle, dim3((n + 255) / 256), dim3(256), 0, 0, n, (const float*)r.p, (const DevTri*)q.p, (float*)o.p);
    HIP_TRY(hipGetLastError()); HIP_TRY(hipDeviceSynchronize());
    return o.down(t, (size_t)n * 4);
    });
}

int srt_kat_phong(int device, uint32_t n, const float* in28, float* rgb) {
    if (!n || !in28 || !rgb) return SRT_ERR_ARG;
    return guarded([&]() -> int {
    HIP_TRY(hipSetDevice(device));
    std::vector<DevTri> tris(n);
    for (uint32_t i = 0; i < n; i++) tris[i] = derive_triangle(in28 + 28 * (size_t)i + 6);
    DevBuf a, q, o;
    KAT_TRY(a.up(in28, (size_t)n * 28 * 4)); KAT_TRY(q.up(tris.data(), (size_t)n * sizeof(DevTri))); KAT_TRY(o.alloc((size_t)n * 12));
    hipLaunchKernelGGL(k_kat_phong, dim3((n + 255) / 256), dim3(256), 0, 0, n, (const float*)a.p, (const DevTri*)q.p, (float*)o.p);
    HIP_TRY(hipGetLastError()); HIP_TRY(hipDeviceSynchronize());
    return o.down(rgb, (size_t)n * 12);
    });
}

int srt_kat_interp_normal(int device, uint32_t n, const float* in12, float* out3) {
    if (!n || !in12 || !out3) return SRT_ERR_ARG;
    HIP_TRY(hipSetDevice(device));
    DevBuf a, o;
    KAT_TRY(a.up(in12, (size_t)n * 48)); KAT_TRY(o.alloc((size_t)n * 12));
    hipLaunchKernelGGL(k_kat_interp_normal, dim3((n + 255) / 256), dim3(256), 0, 0, n, (const float*)a.p, (float*)o.p);
    HIP_TRY(hipGetLastError()); HIP_TRY(hipDeviceSynchronize());
    return o.down(out3, (size_t)n * 12);
}

int srt_kat_pow(int device, uint32_t n, const float* x, const float* y, float* fast, float* lib) {
    if (!n || !x || !y || !fast || !lib) return SRT_ERR_ARG;
    HIP_TRY(hipSetDevice(device));
    DevBuf a, b, o, o2;
    KAT_TRY(a.up(x, (size_t)n * 4)); KAT_TRY(b.up(y, (size_t)n * 4)); KAT_TRY(o.alloc((size_t)n * 4)); KAT_TRY(o2.alloc((size_t)n * 4));
    hipLaunchKernelGGL(k_kat_pow, dim3((n + 255) / 256), dim3(256), 0, 0, n, (const float*)a.p, (const float*)b.p, (float*)o.p, (float*)o2.p);
    HIP_TRY(hipGetLastError()); HIP_TRY(hipDeviceSynchronize());
    KAT_TRY(o.down(fast, (size_t)n * 4));
    return o2.down(lib, (size_t)n * 4);
}

// out[0] = VALU wave-instructions one SIMD issues per cycle: per SIMD (XCC / SE / SH / CU / SIMD of HW_ID), the instructions of the
// waves that ran on it over the cycles from its first wave's start to its last wave's end (s_memtime), median over the SIMDs;
// out[1] = shader clock in GHz during the run (s_memtime against the 100 MHz s_memrealtime); out[2] = the same rate from the
// chip-wide span (all waves' instructions / (SIMDs x clock x (last end - first start))), which includes launch ramp and tail;
// out[3] = waves per SIMD (median) that shared a SIMD during the run.
int srt_debug_valu_rate(int device, uint32_t iters, double* out) {
    if (!iters || !out) return SRT_ERR_ARG;
    return guarded([&]() -> int {
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const uint32_t n_cu = prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256u;
    const uint32_t wgs = n_cu * 8u;                            // 8 workgroups of 4 waves per CU = 8 waves per SIMD, all resident
    DevBuf sink, st;
    KAT_TRY(sink.alloc((size_t)wgs * 256 * 4)); KAT_TRY(st.alloc((size_t)wgs * 4 * 8 * 8));
    for (int rep = 0; rep < 2; rep++) {                        // the first launch warms the clock
        static const bool packed = std::getenv("SRT_VALU_PACKED") != nullptr;        // measure v_pk_fma_f32 instead (same instruction count)
        if (packed) hipLaunchKernelGGL(k_valu_rate<true>, dim3(wgs), dim3(256), 0, 0, iters, (float*)sink.p, (unsigned long long*)st.p);
        else        hipLaunchKernelGGL(k_valu_rate<false>, dim3(wgs), dim3(256), 0, 0, iters, (float*)sink.p, (unsigned long long*)st.p);
        HIP_TRY(hipGetLastError()); HIP_TRY(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h((size_t)wgs * 32);
    KAT_TRY(st.down(h.data(), h.size() * 8));
    const size_t nw = (size_t)wgs * 4;
    std::vector<double> clk(nw);
    unsigned long long rmin = ~0ull, rmax = 0;
    struct Simd { unsigned long long t0 = ~0ull, t1 = 0; uint32_t waves = 0; };
    std::map<unsigned long long, Simd> simds;
    for (size_t w = 0; w < nw; w++) {
        const unsigned long long t0 = h[8 * w], t1 = h[8 * w + 1], r0 = h[8 * w + 2], r1 = h[8 * w + 3], where = h[8 * w + 4];
        clk[w] = (double)(t1 - t0) / (double)(r1 - r0) * 0.1;      // GHz: ticks per 10 ns
        if (r0 < rmin) rmin = r0;
        if (r1 > rmax) rmax = r1;
        Simd& sd = simds[((where >> 32) & 0xfull) << 16 | (where & 0x7f30ull)];      // XCC | SE, SH, CU, SIMD bits of HW_ID
        if (t0 < sd.t0) sd.t0 = t0;
        if (t1 > sd.t1) sd.t1 = t1;
        sd.waves++;
    }
    std::vector<double> rate, share;
    for (const auto& kv : simds) { rate.push_back(64.0 * iters * kv.second.waves / (double)(kv.second.t1 - kv.second.t0)); share.push_back((double)kv.second.waves); }
    std::nth_element(rate.begin(), rate.begin() + rate.size() / 2, rate.end());
    std::nth_element(share.begin(), share.begin() + share.size() / 2, share.end());
    std::nth_element(clk.begin(), clk.begin() + nw / 2, clk.end());
    out[0] = rate[rate.size() / 2];
    out[1] = clk[nw / 2];
    out[2] = (64.0 * iters * (double)nw) / ((double)simds.size() * out[1] * 1e9 * ((double)(rmax - rmin) * 1e-8));
    out[3] = share[share.size() / 2];
    return SRT_OK;
    });
}

int srt_kat_tonemap(int device, uint32_t n, const float* lin, float reinhard, float gamma, float* tone, int32_t* q) {
    if (!n || !lin || !tone || !q) return SRT_ERR_ARG;
    HIP_TRY(hipSetDevice(device));
    DevBuf a, o, o2;
    KAT_TRY(a.up(lin, (size_t)n * 12)); KAT_TRY(o.alloc((size_t)n * 12)); KAT_TRY(o2.alloc((size_t)n * 12));
    hipLaunchKernelGGL(k_kat_tonemap, dim3((3 * n + 255) / 256), dim3(256), 0, 0, n, (const float*)a.p, reinhard, gamma, (float*)o.p, (int32_t*)o2.p);
    HIP_TRY(hipGetLastError()); HIP_TRY(hipDeviceSynchronize());
    KAT_TRY(o.down(tone, (size_t)n * 12));
    return o2.down(q, (size_t)n * 12);
}

} // extern "C"
