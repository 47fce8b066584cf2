// srt_kernels.h -- the HIP kernels of the ray-trace path (gfx950).  Included by srt_hip.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "srt_device.h"

using namespace srt;

// =================================================================================================
// Device-side scene / params views (passed by value in kernarg SGPRs)
// =================================================================================================
struct DevScene {
    const DevNode* nodes;
    const DevWide* wide;          // inner nodes with both children's boxes (node-queue kernels), pre-order over inner nodes
    const int32_t* obj_root_info; // n_objects: what an object's ROOT is -- ~(its DevWide index) < 0, or a leaf's (first << 5 | count)
    const DevNode* root_nodes;    // n_objects: a copy of every object's root record, contiguous -- the root tests of a tile read them with
                                  // INDEPENDENT loads (index = object) instead of obj_range[ob] -> nodes[root], one dependent pair per object
    const float* scene_box;       // 8 floats (min.xyz max.x | max.yz - -): the union of the objects' root boxes, laid out like a node's box
    const DevTri* tris;
    const DevTriO* tris_o;        // the same triangles as rays from the origin test them (closest-hit phase)
    const int32_t* tri_obj;
    const int32_t* tri_tex;       // may be null (no textures)
    const float* tri_tc;          // n_tris x 6, may be null
    const float* tri_normals;     // n_tris x 9 vertex normals, may be null (SRT_FLAG_SMOOTH_NORMALS only)
    const float* obj_color;       // n_objects x 3
    const float* obj_mat;         // n_objects x 3
    const int2* obj_range;        // n_objects: [first node, end node) in pre-order
    const int32_t* obj_tri_first; // n_objects + 1: first triangle id of each object (triangles are numbered object by object)
    const uint8_t* tex;
    const unsigned long long* tex_off;
    const uint32_t* tex_w;
    const uint32_t* tex_h;
    const unsigned long long* tex_size;
    uint32_t n_nodes, n_tris, n_objects;
    uint32_t pad_;                // explicit: argument tables are compared bytewise (frame_table), so no implicit padding anywhere
};
static_assert(sizeof(DevScene) == 20 * 8 + 16, "DevScene has no implicit padding");

struct DevParams {
    uint32_t W, H, rows;          // W = width of the rows this call writes (local width); rows of them
    uint32_t Wimg;                // image width (== W unless the frame is dealt in tiles: col_block > 0)
    uint32_t col_block;           // srt_params.block_cols: 0 = full-width scanline blocks
    uint32_t block_rows, block_first, block_stride;
    int32_t i0, j0;
    float sub_x, sub_y;           // sub-pixel offset added to dir.xy (0 for the reference's one ray per pixel)
    float focal;
    uint32_t n_lights;
    const float* lights;          // device, n_lights x 3
    float shadow_div, reinhard, gamma;
    uint32_t bg;                  // r | g << 8 | b << 16
    uint32_t smooth;              // interpolateNormal mode (simple_raytracer.cpp:132-140,162)
    uint32_t xcd_rows;            // host-side choice of the k_trace_nq build that deals whole tile rows to XCDs
    uint32_t shadow_px_major;     // shadow bits as the packet shadow kernel writes them: per pixel one u64 per 64 light samples
    uint32_t cam;                 // camera mode (srt_params.ray_matrix, an extension): rays are taken into the scene's space
    float cm[12];                 // columns 0..2 (direction) and 3 (origin) of that matrix, xyz each
    uint32_t exp, heavy_steps;    // heavy_steps: packet shadow walks of that many steps make their quadrant a HEAVY one in the next frame's list (0 = off).  exp: experiment switches (A/B variants, wave-uniform branches): bit 0 = queue pushes in LANE order (the round-2 form) instead of node-major
    uint32_t pk_units, pk_take;   // packet shadow kernel: units a wave should have to walk -- surplus waves of the fixed-size grid leave at once (0 = all stay) --, and how many unit numbers one atomic takes
};
static_assert(sizeof(DevParams) == 160, "DevParams has no implicit padding");

// counters[0] hit pixels, [1]/[2] node/triangle tests of the closest-hit kernel, [3]/[4] of the shade kernel
__device__ __forceinline__ void wave_add(unsigned long long* ctr, unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(ctr, v);
}

// Hit-pixel statistic without a hot atomic: one word takes ~12 ns per atomic, so 32 k waves adding to one
// address would serialise for ~0.4 ms.  64 shards, 64 B apart, summed on the host.
constexpr int HIT_SHARDS = 64;
constexpr int CTR_HIT_BASE = 8;                  // counters[8 + 8*shard]
__device__ __forceinline__ void count_hits(unsigned long long* counters, bool is_hit, uint32_t tile_index) {
    const unsigned long long m = __ballot(is_hit);
    if ((threadIdx.x & 63) == 0 && m) {
        const uint32_t shard = tile_index & (HIT_SHARDS - 1);
        atomicAdd(counters + CTR_HIT_BASE + 8 * shard, (unsigned long long)__popcll(m));
    }
}

constexpr int NCTR = 8 + 8 * HIT_SHARDS;        // [1..4] node / triangle tests, [8 + 8*shard] hit counts
// The last kernel of a render clears the counter set of the next render (two sets alternate).
__device__ __forceinline__ void zero_next_counters(unsigned long long* next) {
    if (blockIdx.x == 0 && blockIdx.y == 0)
        for (int i = threadIdx.x; i < NCTR; i += 256) next[i] = 0ull;
}

// local output row -> image row under block-cyclic scanline ownership (include/srt.h srt_params)
__device__ __forceinline__ uint32_t image_row(const DevParams& p, uint32_t r) {
    if (p.col_block) return r;                                             // tiles dealt in two dimensions: every call owns tiles in every row
    if (p.block_stride == 1) return p.block_first * p.block_rows + r;      // consecutive blocks (whole frame on one device): no division
    if ((p.block_rows & (p.block_rows - 1u)) == 0u) {                      // 8, 16, 32 ... rows a block (the usual deal): a shift instead of the
        const uint32_t sh = 31u - (uint32_t)__builtin_clz(p.block_rows);   // ~35-instruction integer division every lane would run
        return (((r >> sh) * p.block_stride + p.block_first) << sh) + (r & (p.block_rows - 1u));
    }
    return ((r / p.block_rows) * p.block_stride + p.block_first) * p.block_rows + (r % p.block_rows);
}
// local output column -> image column (include/srt.h srt_params.block_cols); y = image row
__device__ __forceinline__ uint32_t image_col(const DevParams& p, uint32_t px, uint32_t y) {
    if (!p.col_block) return px;
    const uint32_t off = (p.block_first + p.block_stride - (y / p.block_rows) % p.block_stride) % p.block_stride;     // owned bx = first - by (mod stride)
    return ((px / p.col_block) * p.block_stride + off) * p.col_block + px % p.col_block;
}
// is local pixel (px, r) a pixel of the image (not beyond the buffer, not padding of a tile deal)?
__device__ __forceinline__ bool pixel_live(const DevParams& p, uint32_t px, uint32_t r) {
    return px < p.W && r < p.rows && (!p.col_block || image_col(p, px, r) < p.Wimg);
}
// sendRaysAndIntersectPointsColors:511-517: dir = (i, j, focal), i = x + int(-W/2); px = LOCAL column, y = image row.
// Camera mode (extension): the same ray in the scene's space, (M[0] * dx + M[1] * dy) + M[2] * dz, leaving M[3].xyz.
__device__ __forceinline__ V3 primary_dir(const DevParams& p, uint32_t px, uint32_t y) {
    const V3 d = mk((float)(p.i0 + (int)image_col(p, px, y)) + p.sub_x, (float)(p.j0 + (int)y) + p.sub_y, p.focal);
    if (!p.cam) return d;
    return mk((p.cm[0] * d.x + p.cm[3] * d.y) + p.cm[6] * d.z, (p.cm[1] * d.x + p.cm[4] * d.y) + p.cm[7] * d.z,
              (p.cm[2] * d.x + p.cm[5] * d.y) + p.cm[8] * d.z);
}
__device__ __forceinline__ V3 ray_origin(const DevParams& p) {
    return p.cam ? mk(p.cm[9], p.cm[10], p.cm[11]) : mk(0.0f, 0.0f, 0.0f);
}

// 16x16 pixel tile per 256-thread workgroup, one 8x8 sub-tile per wavefront: the 64 primary rays of
// a wave are neighbours, so they walk the same top-of-tree nodes (loads of one node by many lanes
// coalesce into one 32 B fetch) and diverge only deep in the tree.
__device__ __forceinline__ bool tile_pixel(const DevParams& p, uint32_t& px, uint32_t& r) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    px = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    r  = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);
    return pixel_live(p, px, r);
}
// =================================================================================================
// Kernel 1: closest hit.  rayIntersection:405-431 with boundingBoxIntersection:296-317 fused in:
// walk ALL slab-passing nodes of ALL objects in pre-order (== reference visit order), test leaf
// triangles in stored order, strict '<' keeps the first (lowest id) of equal t.  No t-pruning: the
// reference has none and its slab test may cull what Moller-Trumbore would hit, so the candidate
// set must be reproduced exactly.
// =================================================================================================
template <bool COUNT>
__global__ __launch_bounds__(256) void k_closest_hit(DevScene s, DevParams p, int32_t* __restrict__ hit_id,
                                                     float* __restrict__ t_out, unsigned long long* __restrict__ counters) {
    uint32_t px, r;
    const bool live = tile_pixel(p, px, r);
    unsigned long long n_node = 0, n_tri = 0;
    float best = __builtin_inff();
    int32_t best_id = -1;
    if (live) {
        const V3 o = mk(0.0f, 0.0f, 0.0f);
        const V3 d = primary_dir(p, px, image_row(p, r));
        const float4* nodes4 = reinterpret_cast<const float4*>(s.nodes);
        const float4* tris4 = reinterpret_cast<const float4*>(s.tris);
        int32_t i = 0;
        const int32_t n = (int32_t)s.n_nodes;
        while (i < n) {
            const float4 a = nodes4[2 * (size_t)i], b = nodes4[2 * (size_t)i + 1];
            const int32_t skip = __float_as_int(b.z), leaf = __float_as_int(b.w);
            if (COUNT) n_node++;
            if (ray_aabb(o, d, a.x, a.y, a.z, a.w, b.x, b.y)) {
                if (leaf >= 0) {
                    const int32_t first = leaf >> LEAF_SHIFT, cnt = leaf & LEAF_MAX;
                    for (int32_t k = 0; k < cnt; k++) {
                        const size_t ti = (size_t)(first + k) * 3;
                        const float4 t0 = tris4[ti], t1 = tris4[ti + 1];
                        const float e2z = reinterpret_cast<const float*>(tris4 + ti + 2)[0];
                        if (COUNT) n_tri++;
                        const float t = ray_triangle(o, d, mk(t0.x, t0.y, t0.z), mk(t0.w, t1.x, t1.y), mk(t1.z, t1.w, e2z));
                        if (t != SRT_NEG_INF && t < best) { best = t; best_id = first + k; }
                    }
                }
                i = i + 1;
            } else {
                i = skip;
            }
        }
        const size_t pix = (size_t)r * p.W + px;
        hit_id[pix] = best_id;
        t_out[pix] = best;
    }
    if (COUNT) { wave_add(counters + 1, n_node); wave_add(counters + 2, n_tri); }
}

// =================================================================================================
// Kernel 1, wave-queue form.  Same traversal, but the Moller-Trumbore tests are taken out of the
// divergent per-lane loop: a lane that passes a leaf's box pushes (triangle, lane) pairs into a
// per-wavefront LDS queue; whenever 64 pairs are queued the whole wave tests them at once (one pair per
// lane, 100 % SIMD utilisation) and merges with a 64-bit LDS atomicMin on (t bits << 32 | triangle id).
// t >= 0 orders as its bit pattern, so the minimum key is the reference's closest hit and, on equal t,
// the lowest id, i.e. the first in visit order (strict '<', simple_raytracer.cpp:429).
// =================================================================================================
constexpr int QCAP = 64 + 64 * 8 + 64;      // leftover (< 64) + one push round (64 lanes x <= 8) + slack
constexpr int PUSH_MAX = 8;                 // per lane per round; larger leaves are pushed in slices

__device__ __forceinline__ uint32_t lane_prefix(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

template <bool COUNT>
__global__ __launch_bounds__(256) void k_closest_hit_q(DevScene s, DevParams p, int32_t* __restrict__ hit_id,
                                                       float* __restrict__ t_out, float* __restrict__ rgb_linear,
                                                       uint8_t* __restrict__ rgb8, unsigned long long* __restrict__ counters) {
    __shared__ uint32_t q_all[4][QCAP];
    __shared__ unsigned long long best_all[256];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* q = q_all[wave];
    unsigned long long* best = best_all + wave * 64;
    uint32_t px, r;
    const bool live = tile_pixel(p, px, r);
    const uint32_t tile_x = blockIdx.x * 16 + (wave & 1) * 8, tile_r = blockIdx.y * 16 + (wave >> 1) * 8;
    unsigned long long n_node = 0, n_tri = 0;
    best[lane] = ~0ull;
    const V3 o = mk(0.0f, 0.0f, 0.0f);
    const V3 d = primary_dir(p, px, image_row(p, live ? r : 0));
    const float4* nodes4 = reinterpret_cast<const float4*>(s.nodes);
    const float4* tris4 = reinterpret_cast<const float4*>(s.tris);
    const int32_t n = (int32_t)s.n_nodes;
    int32_t i = live ? 0 : n;
    int32_t leaf_off = 0;
    uint32_t qn = 0;                                    // wave-uniform queue length
    float4 na = nodes4[0], nb = nodes4[1];              // node i (valid while i < n)
    for (;;) {
        const bool active = i < n;
        const unsigned long long act = __ballot(active);
        if (act) {
            uint32_t cnt = 0, first = 0;
            if (active) {
                const float4 a = na, b = nb;
                const int32_t skip = __float_as_int(b.z), leaf = __float_as_int(b.w);
                if (COUNT && leaf_off == 0) n_node++;
                int32_t next;
                bool stay = false;
                if (ray_aabb_nb(o, d, a.x, a.y, a.z, a.w, b.x, b.y)) {
                    next = i + 1;
                    if (leaf >= 0) {
                        const int32_t c = (leaf & LEAF_MAX) - leaf_off;
                        first = (uint32_t)((leaf >> LEAF_SHIFT) + leaf_off);
                        cnt = (uint32_t)(c < PUSH_MAX ? c : PUSH_MAX);
                        if (c > PUSH_MAX) { leaf_off += PUSH_MAX; stay = true; } else leaf_off = 0;
                    }
                } else {
                    next = skip;
                }
                if (!stay) {
                    if (next < n) { na = nodes4[2 * (size_t)next]; nb = nodes4[2 * (size_t)next + 1]; }
                    i = next;
                }
            }
            // wave-wide exclusive prefix sum of cnt (0..8) by bit planes
            uint32_t pre = 0, tot = 0;
            #pragma unroll
            for (int bit = 0; bit < 4; bit++) {
                const unsigned long long m = __ballot((cnt >> bit) & 1u);
                pre += lane_prefix(m) << bit;
                tot += (uint32_t)__popcll(m) << bit;
            }
            for (uint32_t k = 0; k < cnt; k++) q[qn + pre + k] = ((first + k) << 6) | lane;
            qn += tot;
        } else if (qn == 0) {
            break;
        }
        __builtin_amdgcn_wave_barrier();
        while (qn >= 64 || (!act && qn)) {
            const uint32_t m = qn < 64 ? qn : 64;
            qn -= m;
            if (lane < m) {
                const uint32_t e = q[qn + lane];
                const uint32_t src = e & 63u, tri = e >> 6;
                const V3 ds = primary_dir(p, tile_x + (src & 7), image_row(p, tile_r + (src >> 3)));
                const size_t ti = (size_t)tri * 3;
                const float4 t0 = tris4[ti], t1 = tris4[ti + 1];
                const float e2z = reinterpret_cast<const float*>(tris4 + ti + 2)[0];
                if (COUNT) n_tri++;
                const float t = ray_triangle(o, ds, mk(t0.x, t0.y, t0.z), mk(t0.w, t1.x, t1.y), mk(t1.z, t1.w, e2z));
                // candidate iff t != -inf && t < +inf (the initial distanceComparison, :408); NaN fails '<'
                if (t != SRT_NEG_INF && t < __builtin_inff()) {
                    const uint32_t tb = (t == 0.0f) ? 0u : __float_as_uint(t);     // -0.0 ties with +0.0
                    atomicMin(&best[src], ((unsigned long long)tb << 32) | tri);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __builtin_amdgcn_wave_barrier();
    bool is_hit = false;
    if (live) {
        const unsigned long long key = best[lane];
        int32_t id = -1;
        float t = __builtin_inff();
        if (key != ~0ull) {
            id = (int32_t)(uint32_t)key;
            // the winner's t with its own bits (incl. the sign of a zero): same function, same inputs
            const size_t ti = (size_t)id * 3;
            const float4 t0 = tris4[ti], t1 = tris4[ti + 1];
            const float e2z = reinterpret_cast<const float*>(tris4 + ti + 2)[0];
            t = ray_triangle(o, d, mk(t0.x, t0.y, t0.z), mk(t0.w, t1.x, t1.y), mk(t1.z, t1.w, e2z));
        }
        const size_t pix = (size_t)r * p.W + px;
        hit_id[pix] = id;
        t_out[pix] = t;
        if (id < 0) {      // a miss is final here: zero light sum, background pixel (:518, drawImage:476-487)
            if (rgb_linear) { rgb_linear[pix * 3] = 0.0f; rgb_linear[pix * 3 + 1] = 0.0f; rgb_linear[pix * 3 + 2] = 0.0f; }
            if (rgb8) { rgb8[pix * 3] = (uint8_t)(p.bg & 255); rgb8[pix * 3 + 1] = (uint8_t)((p.bg >> 8) & 255); rgb8[pix * 3 + 2] = (uint8_t)((p.bg >> 16) & 255); }
        }
        is_hit = id >= 0;
    }
    count_hits(counters, is_hit, blockIdx.y * gridDim.x + blockIdx.x);
    if (COUNT) { wave_add(counters + 1, n_node); wave_add(counters + 2, n_tri); }
}

// =================================================================================================
// Kernel 1, node-queue form (shipped).  Lanes are decoupled from pixels: a wavefront owns a 4x4 pixel
// tile and keeps TWO LDS queues -- (node, pixel) pairs still to be slab-tested and (triangle, pixel)
// pairs still to be Moller-Trumbore-tested.  Every step pops 64 node pairs (LIFO, so the frontier stays
// depth-first small), tests them one per lane, pushes both children of a passing inner node and the
// triangles of a passing leaf.  All 64 lanes work as long as the tile has work, whatever the spread
// between its rays, and a heavy tile's work is cut into 4x more (and 4x shorter) waves than with one
// ray per lane.  The set of tested pairs is exactly "every node whose ancestors all pass"
// (boundingBoxIntersection:296-317) and the merge is the order-independent (t, id) minimum, so hits,
// t bits and work counts are those of the reference walk.
// If the node queue cannot take a step's children the wave finishes those subtrees with the stackless
// pre-order walk (skip links) instead -- any tree shape is handled with bounded LDS.
// =================================================================================================
// ---- list of quadrants (4x4 pixels) with hits: what the packet shadow kernel (srt_packet.h) works through ----------------
constexpr int QL_SHARDS = 64;            // shard lists; a quadrant goes to shard (tile index & 63)
constexpr int QL_STRIDE = 16;            // counters 64 B apart
// entry = two words: tile index << 2 | quadrant, and the quadrant's 16-bit hit mask (bit = y * 4 + x).  Called by one lane of a
// wave whose quadrant has a hit.
//
// HEAVY quadrants first (round 3).  The packet shadow kernel's launch ends when its longest walk does: a walk through a tree crown
// is 300-500 dependent steps (~0.3 ms), most walks are a dozen, and the quadrants arrive in the order the closest-hit waves happen to
// finish -- so some long walks start when the launch is nearly over (SQ_WAVE_CYCLES: the kernel's 8,192 wave slots are 70 % busy on the
// K4 frame, 45 % on an eighth of it).  A quadrant's walks are as long in the next frame as in this one (an orbit moves the picture by
// a few pixels, a benchmark not at all), so the shadow kernel leaves the step count of a quadrant's longest walk in a cost map
// (one word per quadrant, behind the list: qlist + 2 * QL_SHARDS * qcap words), and the NEXT frame's append reads it (and clears it):
// quadrants at or above heavy_steps grow from the END of their shard's array (own counters, [QL_SHARDS, 2 * QL_SHARDS)), and the
// shadow kernel deals all units of the heavy entries before any other.  Order only: every unit still owns its bytes of the result.
constexpr int QL_FETCH = 64;                     // counters the shadow kernel's waves take their unit numbers from: unit u belongs to counter u % 64.
                                                 // 256 of them (32 waves per counter instead of 128) were measured at the end of round 3 and are much WORSE
                                                 // (K4 9.37 -> 10.2 ms per 8 frames, K3 with 64 samples 7.37 -> 10.3; one number per atomic 14.3 / 14.8): the
                                                 // cost of these atomics is not the waves per address -- fewer atomics (pk_take) is what helps
constexpr int QL_COUNTERS = 2 * QL_SHARDS + QL_FETCH;      // list lengths | heavy-list lengths | units handed out
__device__ __forceinline__ uint32_t* quadrant_cost_map(uint32_t* qlist, uint32_t qcap) { return qlist + 2u * (size_t)QL_SHARDS * qcap; }
__device__ __forceinline__ void quadrant_list_append(uint32_t* __restrict__ qcount, uint32_t* __restrict__ qlist, uint32_t qcap,
                                                     uint32_t tile_index, uint32_t quadrant, uint32_t hit_mask, uint32_t heavy_steps) {
    const uint32_t shard = tile_index & (QL_SHARDS - 1), key = (tile_index << 2) | quadrant;
    bool heavy = false;
    if (heavy_steps) {
        uint32_t* const cost = quadrant_cost_map(qlist, qcap) + key;       // key < 4 * tiles <= QL_SHARDS * qcap
        const uint32_t c = *cost;
        heavy = c >= heavy_steps;
        if (c) *cost = 0u;
    }
    if (heavy) {
        const uint32_t slot = atomicAdd(qcount + (QL_SHARDS + shard) * QL_STRIDE, 1u);
        if (slot < qcap) reinterpret_cast<uint2*>(qlist)[(size_t)shard * qcap + (qcap - 1u - slot)] = make_uint2(key, hit_mask);
    } else {
        const uint32_t slot = atomicAdd(qcount + shard * QL_STRIDE, 1u);
        if (slot < qcap) reinterpret_cast<uint2*>(qlist)[(size_t)shard * qcap + slot] = make_uint2(key, hit_mask);
    }
}

constexpr int NQ_P = 16;                    // rays per wavefront of the shadow kernel (4x4 pixel quadrant)
constexpr int LQ_WORDS = 2 * (64 + 128);    // (leaf, ray) pair queue of the node-queue kernels: < 64 left over + <= 128 per push (two children per lane), 2 words each

// TWL / THL: log2 of the tile width / height a wavefront owns (shipped: 4x4); FILTER: filtered slab predicate.
#ifdef SRT_DIAG
// Diagnostic build only (python -m simple_raytracer_amd.build --diag -> libsrt_hip_diag.so, never shipped): per-wave
// cycle stamps of the closest-hit phase, written where rgb_linear would go (8 x u64 per wave).
#define SRT_STAMP(v) do { v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); } while (0)
// whole-tile record of the fused kernels: slot 5 = HW_ID | XCC_ID << 32, slots 6 / 7 = absolute start / end stamps
__device__ __forceinline__ void diag_tile_record(float* rgb_linear, uint32_t bx, uint32_t by, uint32_t gx, unsigned long long k0, unsigned long long k1,
                                                 unsigned long long ka = 0, unsigned long long kb = 0) {      // ka / kb: after the launch-time barrier / after the closest-hit phase
    if ((threadIdx.x & 63) == 0 && rgb_linear) {
        unsigned long long* dgp = reinterpret_cast<unsigned long long*>(rgb_linear) + (((size_t)by * gx + bx) * 4 + (threadIdx.x >> 6)) * 8;
        const unsigned long long hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
        dgp[5] = hw | (xcc << 32); dgp[6] = k0; dgp[7] = k1;
        if (ka) { dgp[3] = ka - k0; dgp[4] = kb - ka; }
    }
}
#else
#define SRT_STAMP(v) do { } while (0)
#endif

// The phase runs per wavefront on LDS the caller provides: nq[NQCAP], tq[QCAP], best[P], dir[P].  On return the
// lanes < P hold their pixel's hit id and t (also written to hit_id / t_out, with the final pixel for misses).
// CAM: camera mode (srt_params.ray_matrix): the rays start at the camera's position in the scene's space and their directions have
// three varying components, so the phase keeps (dx, dy, dz) per ray, takes the reciprocals per test and tests triangles in the general
// form (P1, e1, e2 and the ray's origin) instead of the origin form -- the arithmetic of the oracle's camera mode.
// WIDE: the queue holds inner nodes that are known to pass, as indices of their DevWide records (srt_device.h): a pop tests BOTH
// children.  !WIDE is the round-1/2 form (a queue entry is a node still to be tested, 32 B records), kept for A/B (variant 40).
template <bool COUNT, int NQCAP, int TWL, int THL, bool FILTER, bool CAM = false, bool WIDE = false>
__device__ __forceinline__ void closest_hit_phase(const DevScene& s, const DevParams& p, uint32_t* nq, uint32_t* tq,
                                                  unsigned long long* best, float4* dir,
                                                  int32_t* __restrict__ hit_id, float* __restrict__ t_out,
                                                  float* __restrict__ rgb_linear, uint8_t* __restrict__ rgb8,
                                                  unsigned long long* __restrict__ counters, int32_t& out_id, float& out_t, V3& out_d,
                                                  const uint32_t bx, const uint32_t by, const uint32_t gx,      // workgroup tile coordinates, tiles per row
                                                  const uint32_t wave,                                          // quadrant of the tile this wave owns
                                                  uint32_t* __restrict__ qcount = nullptr, uint32_t* __restrict__ qlist = nullptr, uint32_t qcap = 0,   // list of quadrants with hits (4x4 waves only)
                                                  const uint32_t* root_pass = nullptr) {      // per ray of this quadrant: which objects' ROOT boxes it passes (bit = object), from
                                                                                              // finish_background_tile, which has put all 64 rays of the tile through every root already
    constexpr int P = 1 << (TWL + THL);           // rays per wavefront
    const uint32_t lane = threadIdx.x & 63;
    const float4* nodes4 = reinterpret_cast<const float4*>(s.nodes);
    const float4* tris4 = reinterpret_cast<const float4*>(s.tris);
    const uint32_t tile_x = (bx * 2 + (wave & 1)) << TWL, tile_r = (by * 2 + (wave >> 1)) << THL;
    const uint32_t px = tile_x + (lane & ((1u << TWL) - 1)), r = tile_r + ((lane >> TWL) & ((1u << THL) - 1));
    const bool live = lane < P && pixel_live(p, px, r);
    const V3 o = CAM ? ray_origin(p) : mk(0.0f, 0.0f, 0.0f);
    V3 dmine = mk(0.f, 0.f, p.focal);
    if (lane < P) {
        best[lane] = ~0ull;
        if (live) dmine = primary_dir(p, px, image_row(p, r));
        if (CAM) dir[lane] = make_float4(dmine.x, dmine.y, dmine.z, 0.f);
        else dir[lane] = make_float4(dmine.x, dmine.y, __builtin_amdgcn_rcpf(dmine.x), __builtin_amdgcn_rcpf(dmine.y));   // + reciprocals for the filtered slab test
    }
    const float rcp_focal = __builtin_amdgcn_rcpf(p.focal);
    const unsigned long long livem = __ballot(live);
    unsigned long long n_node = 0, n_tri = 0;
    uint32_t nqn = 0, tqn = 0;                       // wave-uniform queue lengths
    __builtin_amdgcn_wave_barrier();
    unsigned long long dg_pop = 0, dg_load = 0; (void)dg_pop; (void)dg_load;
    unsigned long long dg_t0 = 0, dg_a = 0, dg_b = 0, dg_test = 0, dg_commit = 0, dg_tri = 0, dg_steps = 0, dg_batches = 0, dg_items = 0, dg_titems = 0;
    (void)dg_t0; (void)dg_a; (void)dg_b; (void)dg_test; (void)dg_commit; (void)dg_tri; (void)dg_steps; (void)dg_batches; (void)dg_items; (void)dg_titems;
    SRT_STAMP(dg_t0);

    // One batch of <= 64 queued (leaf, pixel) pairs, one per lane: the lane runs Moller-Trumbore over the leaf's
    // triangles in stored order (next triangle's loads issued before the current test), keeps the first minimum
    // (strict '<', :429) and merges once with the LDS atomicMin.  A queue entry is two words: the leaf's
    // (first << 5 | count) and the pixel lane.
    auto tri_batch = [&]() {
        const uint32_t m = tqn < 64 ? tqn : 64;
        tqn -= m;
#ifdef SRT_DIAG
        unsigned long long c0, c1; SRT_STAMP(c0); dg_batches++; dg_titems += m;
#endif
        // a batch that does not fill the wave (the flush at the end of a tile: 16 rays x a leaf or two) is spread over
        // sub-lanes: S lanes share one (leaf, ray) pair and take every S-th triangle; the merge below does not care who
        // found the minimum.  A full batch has S = 1: one pair per lane.
        const uint32_t sh = m <= 8 ? 3u : (m <= 16 ? 2u : (m <= 32 ? 1u : 0u));
        const uint32_t S = 1u << sh, pi = lane >> sh, sub = lane & (S - 1u);
        if (pi < m) {
            const uint32_t info = tq[2 * (tqn + pi)], pl = tq[2 * (tqn + pi) + 1];
            const uint32_t first = info >> LEAF_SHIFT, cnt = info & LEAF_MAX;
            const float4 dxy = dir[pl];
            const V3 d = mk(dxy.x, dxy.y, CAM ? dxy.z : p.focal);
            float bt = __builtin_inff();
            uint32_t bi = 0;
            // one triangle per iteration, the next one's three dwordx4 loads issued before the current test: a wave spends
            // most of its life behind s_waitcnt, and the all-lanes-rejected exits of a single test skip more code than two
            // interleaved tests could (measured: 5 % of the launch); ids rise inside a lane, so strict '<' keeps the first minimum
            // The GENERAL record (P1, e1, e2) also for rays from the origin: 12 operations more a test than the origin record (tvec, qvec
            // precomputed; still what the packet kernel reads, whose operands are scalar), but ONE record array in the caches instead of two
            // -- this phase and the shadow phase then read the same 48 bytes of a triangle.  o = 0 gives tvec = 0 - P1 and qvec =
            // cross(tvec, e1) with the host's operations: same bits.  Same box: K3 4.30 -> 4.25 ms per 36 frames, cube over ground
            // 3.12 -> 3.06, K4's closest-hit launch 0.239 -> 0.232 ms.
            const float4* tp = reinterpret_cast<const float4*>(s.tris) + ((size_t)first + sub) * 3;
            float4 n0, n1; float n2;                     // (the record's last three words are the face normal: not read here)
            if (sub < cnt) { n0 = tp[0]; n1 = tp[1]; n2 = tp[2].x; }
            for (uint32_t k = sub; k < cnt; k += S) {
                const float4 a0 = n0, a1 = n1; const float a2 = n2;
                if (k + S < cnt) { tp += 3 * S; n0 = tp[0]; n1 = tp[1]; n2 = tp[2].x; }      // the next triangle's record is in flight during this test
                if (COUNT) n_tri++;
                const float ta = ray_triangle(o, d, mk(a0.x, a0.y, a0.z), mk(a0.w, a1.x, a1.y), mk(a1.z, a1.w, a2));
                // candidate iff t != -inf && t < best (initially +inf, :408); NaN fails '<'; -0.0 == +0.0 keeps the first
                if (ta != SRT_NEG_INF && ta < bt) { bt = ta; bi = first + k; }
            }
            if (bt < __builtin_inff()) {
                const uint32_t tb = (bt == 0.0f) ? 0u : __float_as_uint(bt);     // -0.0 ties with +0.0
                atomicMin(&best[pl], ((unsigned long long)tb << 32) | bi);
            }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef SRT_DIAG
        SRT_STAMP(c1); dg_tri += c1 - c0;
#endif
    };
    // queue each lane's passing leaf (is_leaf = false: nothing)
    auto push_tris = [&](uint32_t info, bool is_leaf, uint32_t pl) {
        const unsigned long long lm = __ballot(is_leaf);
        if (lm) {
            if (is_leaf) {
                const uint32_t pos = tqn + lane_prefix(lm);
                tq[2 * pos] = info; tq[2 * pos + 1] = pl;
            }
            tqn += (uint32_t)__popcll(lm);
            __builtin_amdgcn_wave_barrier();
            while (tqn >= 64) tri_batch();
        }
    };

    // the same for up to two passing leaves per lane (the two children of a DevWide record)
    auto push_tris2 = [&](uint32_t info_a, bool leaf_a, uint32_t info_b, bool leaf_b, uint32_t pl) {
        const unsigned long long ma = __ballot(leaf_a), mb = __ballot(leaf_b);
        if (ma | mb) {
            uint32_t pos = tqn + lane_prefix(ma) + lane_prefix(mb), q = pos + (leaf_a ? 1u : 0u);
            if (!(p.exp & 1u)) { pos = tqn + lane_prefix(ma); q = tqn + (uint32_t)__popcll(ma) + lane_prefix(mb); }      // node-major
            if (leaf_a) { tq[2 * pos] = info_a; tq[2 * pos + 1] = pl; }
            if (leaf_b) { tq[2 * q] = info_b; tq[2 * q + 1] = pl; }
            tqn += (uint32_t)__popcll(ma) + (uint32_t)__popcll(mb);
            __builtin_amdgcn_wave_barrier();
            while (tqn >= 64) tri_batch();
        }
    };

    const uint32_t n_obj = s.n_objects;
    const uint32_t nlive = (uint32_t)__popcll(livem);
    constexpr uint32_t OBJ_G = (NQCAP / (2 * P)) < 16 ? (NQCAP / (2 * P)) : 16;      // roots pushed at once: P * OBJ_G <= NQCAP / 2
    if constexpr (WIDE) {
    const float4* wide4 = reinterpret_cast<const float4*>(s.wide);
    for (uint32_t obj0 = 0; obj0 < n_obj && nlive; obj0 += OBJ_G) {
        const uint32_t g = (n_obj - obj0) < OBJ_G ? (n_obj - obj0) : OBJ_G;
        // the roots: either tested already (root_pass: wave 0 ran them for the whole tile with full lanes) or tested here, one (ray,
        // object) pair per lane; a passing root that is an inner node is queued, a passing leaf's triangles are
        for (uint32_t base = 0; base < P * g; base += 64) {
            const uint32_t k = base + lane;
            const uint32_t pl = k & (P - 1), ob = k >> (TWL + THL);
            bool ok = k < P * g && ((livem >> pl) & 1ull);
            if (!COUNT && root_pass) ok = ok && ((root_pass[pl] >> (obj0 + ob)) & 1u);
            else if (ok) {
                const int32_t root = s.obj_range[obj0 + ob].x;
                const float4 a = nodes4[2 * (size_t)root], b = nodes4[2 * (size_t)root + 1];
                const float4 dxy = dir[pl];
                const V3 d = mk(dxy.x, dxy.y, CAM ? dxy.z : p.focal);
                RayRcp rc;
                if (CAM) rc = ray_rcp(d);
                else { rc.x = dxy.z; rc.y = dxy.w; rc.z = rcp_focal; }
                if (COUNT) n_node++;
                ok = slab_pass<FILTER>(o, d, rc, a.x, a.y, a.z, a.w, b.x, b.y);
            }
            int32_t info = 0;
            if (ok) info = s.obj_root_info[obj0 + ob];
            const bool inner = ok && info < 0, leafp = ok && info >= 0 && (info & LEAF_MAX) != 0;
            const unsigned long long im = __ballot(inner);
            if (inner) nq[nqn + lane_prefix(im)] = ((uint32_t)(~info) << 6) | pl;
            nqn += (uint32_t)__popcll(im);
            push_tris2((uint32_t)info, leafp, 0u, false, pl);
        }
        __builtin_amdgcn_wave_barrier();
        while (nqn) {
            const uint32_t m = nqn < 64 ? nqn : 64;
            nqn -= m;
#ifdef SRT_DIAG
            SRT_STAMP(dg_a); dg_steps++; dg_items += m;
            const unsigned long long tri_before = dg_tri;
#endif
            const bool have = lane < m;
            uint32_t pl = 0;
            int32_t linfo = 0, rinfo = 0, node = 0, rnode = 0;
            bool in_l = false, in_r = false, lf_l = false, lf_r = false;
            V3 d = mk(0.f, 0.f, p.focal);
            if (have) {
                const uint32_t e = nq[nqn + lane];
                pl = e & 63u;
#ifdef SRT_DIAG
                { unsigned long long c_; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SRT_STAMP(c_); dg_pop += c_ - dg_a; dg_a = c_; }
#endif
                const float4* wp = wide4 + 4 * (size_t)(e >> 6);
                const float4 r0 = wp[0], r1 = wp[1], r2 = wp[2];
                const int4 r3 = reinterpret_cast<const int4*>(wp)[3];
                const float4 dxy = dir[pl];
#ifdef SRT_DIAG
                { unsigned long long c_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); SRT_STAMP(c_); dg_load += c_ - dg_a; dg_a = c_; }
#endif
                d = mk(dxy.x, dxy.y, CAM ? dxy.z : p.focal);
                RayRcp rc;
                if (CAM) rc = ray_rcp(d);
                else { rc.x = dxy.z; rc.y = dxy.w; rc.z = rcp_focal; }
                linfo = r3.x; rinfo = r3.y; node = r3.z; rnode = r3.w;
                if (COUNT) n_node += 2;
                bool pass_l, pass_r;
                slab_pass2<FILTER>(o, d, rc, r0, r1, r2, pass_l, pass_r);
                in_l = pass_l && linfo < 0; lf_l = pass_l && linfo >= 0 && (linfo & LEAF_MAX) != 0;
                in_r = pass_r && rinfo < 0; lf_r = pass_r && rinfo >= 0 && (rinfo & LEAF_MAX) != 0;
            }
            __builtin_amdgcn_wave_barrier();
#ifdef SRT_DIAG
            SRT_STAMP(dg_b); dg_test += dg_b - dg_a;
#endif
            const unsigned long long ml = __ballot(in_l), mr = __ballot(in_r);
            const uint32_t n_in = (uint32_t)__popcll(ml) + (uint32_t)__popcll(mr);
            if (nqn + n_in <= (uint32_t)NQCAP) {
                if (!(p.exp & 1u)) {      // node-major: all right children, then all left children -- rays that visit one node stay neighbours
                    if (in_r) nq[nqn + lane_prefix(mr)] = ((uint32_t)(~rinfo) << 6) | pl;
                    if (in_l) nq[nqn + (uint32_t)__popcll(mr) + lane_prefix(ml)] = ((uint32_t)(~linfo) << 6) | pl;
                } else {
                const uint32_t pos = nqn + lane_prefix(ml) + lane_prefix(mr);
                if (in_r) nq[pos] = ((uint32_t)(~rinfo) << 6) | pl;
                if (in_l) nq[pos + (in_r ? 1u : 0u)] = ((uint32_t)(~linfo) << 6) | pl;      // left child on top: popped first
                }
                nqn += n_in;
                push_tris2((uint32_t)linfo, lf_l, (uint32_t)rinfo, lf_r, pl);
            } else {
                // queue full: finish the passing children's subtrees with the stackless pre-order walk over the 32 B records
                // (i = pass ? i + 1 : skip[i]).  Left subtree = nodes (node + 2 .. rnode), right = (rnode + 1 .. skip[rnode]); with both
                // the walk runs through and hops over the right child itself, which has been tested above.
                push_tris2((uint32_t)linfo, lf_l, (uint32_t)rinfo, lf_r, pl);
                int32_t i = 0, end = 0, hop = -1;
                if (in_l | in_r) {
                    i = in_l ? node + 2 : rnode + 1;
                    end = in_r ? s.nodes[rnode].skip : rnode;
                    hop = (in_l && in_r) ? rnode : -1;
                }
                while (__ballot(i < end)) {
                    int32_t inf2 = -1;
                    bool lp2 = false;
                    if (i < end) {
                        if (i == hop) i = hop + 1;
                        else {
                            const float4 a = nodes4[2 * (size_t)i], b = nodes4[2 * (size_t)i + 1];
                            const int32_t sk = __float_as_int(b.z);
                            inf2 = __float_as_int(b.w);
                            if (COUNT) n_node++;
                            if (ray_aabb_nb(o, d, a.x, a.y, a.z, a.w, b.x, b.y)) {
                                lp2 = inf2 >= 0 && (inf2 & LEAF_MAX) != 0;
                                i = i + 1;
                            } else {
                                i = sk;
                            }
                        }
                    }
                    push_tris2((uint32_t)inf2, lp2, 0u, false, pl);
                }
            }
            __builtin_amdgcn_wave_barrier();
#ifdef SRT_DIAG
            { unsigned long long c2_; SRT_STAMP(c2_); dg_commit += (c2_ - dg_b) - (dg_tri - tri_before); }
#endif
        }
    }
    } else
    for (uint32_t obj0 = 0; obj0 < n_obj && nlive; obj0 += OBJ_G) {
        // roots of up to OBJ_G objects for every live pixel, in chunks of 64 (node, pixel) pairs
        const uint32_t g = (n_obj - obj0) < OBJ_G ? (n_obj - obj0) : OBJ_G;
        if (!COUNT && root_pass) {
            // the root tests are done (wave 0 ran them for the whole tile with full lanes): queue what a passing root queues -- its two
            // children, or its triangles -- and skip the first node step, which would run at 16 rays x 2 roots = 32 of 64 lanes
            for (uint32_t base = 0; base < P * g; base += 64) {
                const uint32_t k = base + lane;
                const uint32_t pl = k & (P - 1), ob = k >> (TWL + THL);
                const bool ok = k < P * g && ((livem >> pl) & 1ull) && ((root_pass[pl] >> (obj0 + ob)) & 1u);
                int32_t root = 0, info = -1;
                if (ok) { root = s.obj_range[obj0 + ob].x; info = s.root_nodes[obj0 + ob].leaf; }      // two independent loads
                const bool inner = ok && info < 0, leafp = ok && info >= 0 && (info & LEAF_MAX) != 0;
                const unsigned long long im = __ballot(inner);
                if (inner) {
                    uint32_t pos = nqn + 2 * lane_prefix(im), pos1 = pos + 1;
                    if (!(p.exp & 1u)) { pos = nqn + lane_prefix(im); pos1 = pos + (uint32_t)__popcll(im); }      // node-major
                    nq[pos] = ((uint32_t)(~info) << 6) | pl;          // right child
                    nq[pos1] = ((uint32_t)(root + 1) << 6) | pl;      // left child on top: popped first
                }
                nqn += 2 * (uint32_t)__popcll(im);
                push_tris((uint32_t)info, leafp, pl);
            }
        } else
        for (uint32_t base = 0; base < P * g; base += 64) {
            const uint32_t k = base + lane;
            const uint32_t pl = k & (P - 1), ob = k >> (TWL + THL);
            const bool ok = k < P * g && ((livem >> pl) & 1ull);
            const unsigned long long m = __ballot(ok);
            if (ok) nq[nqn + lane_prefix(m)] = ((uint32_t)s.obj_range[obj0 + ob].x << 6) | pl;
            nqn += (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        while (nqn) {
            const uint32_t m = nqn < 64 ? nqn : 64;
            nqn -= m;
#ifdef SRT_DIAG
            SRT_STAMP(dg_a); dg_steps++; dg_items += m;
            const unsigned long long tri_before = dg_tri;
#endif
            const bool have = lane < m;
            uint32_t pl = 0;
            int32_t node = 0, info = -1, skip = 0;
            bool inner = false, leafp = false;
            V3 d = mk(0.f, 0.f, p.focal);
            if (have) {
                const uint32_t e = nq[nqn + lane];
                pl = e & 63u; node = (int32_t)(e >> 6);
#ifdef SRT_DIAG
                { unsigned long long c_; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); SRT_STAMP(c_); dg_pop += c_ - dg_a; dg_a = c_; }
#endif
                const float4 a = nodes4[2 * (size_t)node], b = nodes4[2 * (size_t)node + 1];
                const float4 dxy = dir[pl];
#ifdef SRT_DIAG
                { unsigned long long c_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); SRT_STAMP(c_); dg_load += c_ - dg_a; dg_a = c_; }
#endif
                d = mk(dxy.x, dxy.y, CAM ? dxy.z : p.focal);
                skip = __float_as_int(b.z); info = __float_as_int(b.w);
                if (COUNT) n_node++;
                bool pass;
                if (FILTER) {
                    bool amb;
                    RayRcp rc;
                    if (CAM) rc = ray_rcp(d);
                    else { rc.x = dxy.z; rc.y = dxy.w; rc.z = rcp_focal; }
                    pass = ray_aabb_filtered(o, rc, a.x, a.y, a.z, a.w, b.x, b.y, amb);
                    if (amb) pass = ray_aabb_nb(o, d, a.x, a.y, a.z, a.w, b.x, b.y);      // rare: exact divides decide
                } else {
                    pass = ray_aabb_nb(o, d, a.x, a.y, a.z, a.w, b.x, b.y);
                }
                if (pass) {
                    if (info < 0) inner = true;
                    else leafp = (info & LEAF_MAX) != 0;
                }
            }
            __builtin_amdgcn_wave_barrier();
#ifdef SRT_DIAG
            SRT_STAMP(dg_b); dg_test += dg_b - dg_a;
#endif
            const unsigned long long im = __ballot(inner);
            const uint32_t n_in = (uint32_t)__popcll(im);
            if (nqn + 2 * n_in <= (uint32_t)NQCAP) {
                if (inner) {
                    uint32_t pos = nqn + 2 * lane_prefix(im), pos1 = pos + 1;
                    if (!(p.exp & 1u)) { pos = nqn + lane_prefix(im); pos1 = pos + n_in; }      // node-major
                    nq[pos] = ((uint32_t)(~info) << 6) | pl;          // right child
                    nq[pos1] = ((uint32_t)(node + 1) << 6) | pl;      // left child on top: popped first
                }
                nqn += 2 * n_in;
                push_tris((uint32_t)info, leafp, pl);
            } else {
                // queue full: finish these subtrees with the stackless pre-order walk (i = pass ? i+1 : skip[i])
                push_tris((uint32_t)info, leafp, pl);
                int32_t i = inner ? node + 1 : 0, end = inner ? skip : 0;
                while (__ballot(i < end)) {
                    int32_t inf2 = -1;
                    bool lp2 = false;
                    if (i < end) {
                        const float4 a = nodes4[2 * (size_t)i], b = nodes4[2 * (size_t)i + 1];
                        const int32_t sk = __float_as_int(b.z);
                        inf2 = __float_as_int(b.w);
                        if (COUNT) n_node++;
                        if (ray_aabb_nb(o, d, a.x, a.y, a.z, a.w, b.x, b.y)) {
                            lp2 = inf2 >= 0 && (inf2 & LEAF_MAX) != 0;
                            i = i + 1;
                        } else {
                            i = sk;
                        }
                    }
                    push_tris((uint32_t)inf2, lp2, pl);
                }
            }
            __builtin_amdgcn_wave_barrier();
#ifdef SRT_DIAG
            { unsigned long long c2_; SRT_STAMP(c2_); dg_commit += (c2_ - dg_b) - (dg_tri - tri_before); }
#endif
        }
    }
    while (tqn) tri_batch();
    __builtin_amdgcn_wave_barrier();
#ifdef SRT_DIAG
    {
        unsigned long long c3; SRT_STAMP(c3);
        if (lane == 0 && rgb_linear) {
            unsigned long long* dgp = reinterpret_cast<unsigned long long*>(rgb_linear) +
                                      (((size_t)by * gx + bx) * 4 + wave) * 8;
            dgp[0] = c3 - dg_t0; dgp[1] = dg_steps; dgp[2] = dg_batches; dgp[3] = dg_test; dgp[4] = dg_commit; dgp[5] = dg_tri;
            dgp[6] = dg_pop; dgp[7] = dg_load;
        }
        rgb_linear = nullptr;
    }
#endif

    bool is_hit = false;
    out_id = -1; out_t = __builtin_inff(); out_d = dmine;
    if (live) {
        const unsigned long long key = best[lane];
        int32_t id = -1;
        float t = __builtin_inff();
        if (key != ~0ull) {
            id = (int32_t)(uint32_t)key;
            // the key carries the winner's t bit for bit, except that -0.0 was merged as +0.0 (they tie, :429): only a
            // zero is evaluated again, to get its sign (same function, same inputs)
            t = __uint_as_float((uint32_t)(key >> 32));
            if (t == 0.0f) {
                const size_t ti = (size_t)id * 3;
                const float4 t0 = tris4[ti], t1 = tris4[ti + 1];
                const float e2z = reinterpret_cast<const float*>(tris4 + ti + 2)[0];
                t = ray_triangle(o, dmine, mk(t0.x, t0.y, t0.z), mk(t0.w, t1.x, t1.y), mk(t1.z, t1.w, e2z));
            }
        }
        const size_t pix = (size_t)r * p.W + px;
        hit_id[pix] = id;
        t_out[pix] = t;
        if (id < 0) {      // a miss is final here: zero light sum, background pixel (:518, drawImage:476-487)
            if (rgb_linear) { rgb_linear[pix * 3] = 0.0f; rgb_linear[pix * 3 + 1] = 0.0f; rgb_linear[pix * 3 + 2] = 0.0f; }
            if (rgb8) { rgb8[pix * 3] = (uint8_t)(p.bg & 255); rgb8[pix * 3 + 1] = (uint8_t)((p.bg >> 8) & 255); rgb8[pix * 3 + 2] = (uint8_t)((p.bg >> 16) & 255); }
        }
        is_hit = id >= 0;
        out_id = id; out_t = t;
    }
    // The hit statistic is taken by the SHADING kernel since the end of round 3 (one atomic per 8x8 tile with hits there instead of one per
    // quadrant wave here: these fire-and-forget atomics cost the K3 trace launch 4 % -- 0.1157 -> 0.111 ms without them); exp bit 5: the
    // pipeline has no shading launch behind this phase (k_trace_shade_nq), count here.
    if (p.exp & 32u) count_hits(counters, is_hit, by * gx + bx);
    if (TWL == 2 && THL == 2 && qlist) {
        const unsigned long long hm = __ballot(is_hit);          // all lanes vote: not inside the lane-0 branch
        if (lane == 0 && hm) quadrant_list_append(qcount, qlist, qcap, by * gx + bx, wave, (uint32_t)hm, p.heavy_steps);
    }
    if (COUNT) { wave_add(counters + 1, n_node); wave_add(counters + 2, n_tri); }
}

// Background tiles: most of a typical frame, and a wave that only learns that its 16 rays miss every object costs ~150 VALU
// instructions of set-up, queueing and write-out.  So wave 0 of a workgroup first puts the tile's 64 rays (one per lane)
// through the slab test of every object's root box -- the very test the first node step of each ray would run -- while the
// other three waves wait at a launch-time barrier (nobody has work to wait behind yet); if no ray passes any root it writes
// the 64 background pixels (and, when shadow_bits is given, the tile's all-clear shadow words) with full lanes.  Returns true
// to every thread of the workgroup when the tile is finished.
// One WAVE puts the 64 rays of tile (bx, by) through every object's root box; if none passes any it writes the tile's background
// pixels (and all-clear shadow words).  Returns (wave-uniform) whether the tile is live; root_pass (LDS, 64 words, quadrant * 16 +
// pixel) receives, per ray, the objects whose root box it passes.
template <bool FILTER, bool CAM = false>
__device__ __forceinline__ bool background_test_wave(const DevScene& s, const DevParams& p, int32_t* __restrict__ hit_id, float* __restrict__ t_out,
                                                     float* __restrict__ rgb_linear, uint8_t* __restrict__ rgb8,
                                                     unsigned long long* __restrict__ shadow_bits,
                                                     const uint32_t bx, const uint32_t by, const uint32_t gx, uint32_t* root_pass) {
    const uint32_t lane = threadIdx.x & 63, quad = lane >> 4, ql = lane & 15u;
    const uint32_t px = bx * 8 + (quad & 1) * 4 + (ql & 3), r = by * 8 + (quad >> 1) * 4 + (ql >> 2);
    const bool live = pixel_live(p, px, r);
    const V3 o = CAM ? ray_origin(p) : mk(0.f, 0.f, 0.f);
    const V3 dd = live ? primary_dir(p, px, image_row(p, r)) : mk(0.f, 0.f, p.focal);
    const RayRcp rc = ray_rcp(dd);
    const float4* nodes4 = reinterpret_cast<const float4*>(s.nodes);
    bool any = false;
    uint32_t pmask = 0;
    // four objects at a time: their root records come from the contiguous table by wave-uniform (scalar) loads that depend on nothing
    // but the object's number, so all eight requests of a group are in flight together (a scene of seven objects used to be a chain of
    // fourteen dependent round trips before the workgroup's other three waves could start)
    const float4* rn4 = reinterpret_cast<const float4*>(s.root_nodes);
    const uint32_t n_obj = s.n_objects;
    // Scenes of several objects: the union of their root boxes first.  A ray that CERTAINLY fails it -- filtered test decided, not
    // ambiguous: tfar_U < tnear_U by more than the margin -- fails every root box inside it under the reference's own comparisons: per
    // axis an inner box's t-interval nests in the union's, so tnear_A >= tnear_U and tfar_A <= tfar_U, and tnear_A - tfar_A exceeds both
    // the union's margin (2e-6 of |tnear_U| + |tfar_U|) and, when A's values are much larger than the union's, half of |tnear_A| +
    // |tfar_A| itself -- either way far more than the half-ulp the correctly rounded quotients of A can move.  Most tiles of a 4K frame
    // of the reference's scenes are sky: one test instead of one per object on the path every tile waits behind.
    bool skip_roots = false;
    if (FILTER && n_obj >= 3u && !(p.exp & 10u)) {      // (variants 45 / 46 switch it off, A/B)
        const float4 ua = reinterpret_cast<const float4*>(s.scene_box)[0], ub = reinterpret_cast<const float4*>(s.scene_box)[1];
        bool amb_u;
        const bool pass_u = ray_aabb_filtered(o, rc, ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, amb_u);
        skip_roots = !__ballot(live && (pass_u || amb_u));
    }
    if (skip_roots) { /* no ray can pass any root: pmask stays 0 */ } else
    if (p.exp & 2u) {        // A/B: the round-2 loop (obj_range[ob] -> nodes[root], one object after the other)
        for (uint32_t ob = 0; ob < n_obj; ob++) {
            const int32_t root = s.obj_range[ob].x;
            const float4 a = nodes4[2 * (size_t)root], b = nodes4[2 * (size_t)root + 1];
            const bool pass = slab_pass<FILTER>(o, dd, rc, a.x, a.y, a.z, a.w, b.x, b.y);
            any |= pass;
            if (pass && ob < 32u) pmask |= 1u << ob;
        }
    } else {
        // G objects at a time (2 for scenes of one or two objects, else 4)
        auto group = [&](const uint32_t ob0, auto G_) {
            constexpr uint32_t G = decltype(G_)::value;
            float4 a[G], b[G];
#pragma unroll
            for (uint32_t k = 0; k < G; k++) {
                const uint32_t ob = ob0 + k < n_obj ? ob0 + k : n_obj - 1u;
                a[k] = rn4[2 * (size_t)ob]; b[k] = rn4[2 * (size_t)ob + 1];
            }
            // (a slot beyond the last object repeats the last object: same bit.)  The filtered tests are one straight line of code, so
            // the loads above stay together at its head; what the filter could not decide is settled afterwards by the exact form
            bool pass[G], amb[G];
            bool any_amb = false;
#pragma unroll
            for (uint32_t k = 0; k < G; k++) {
                if (FILTER) pass[k] = ray_aabb_filtered(o, rc, a[k].x, a[k].y, a[k].z, a[k].w, b[k].x, b[k].y, amb[k]);
                else { pass[k] = ray_aabb_nb(o, dd, a[k].x, a[k].y, a[k].z, a[k].w, b[k].x, b[k].y); amb[k] = false; }
                any_amb |= amb[k];
            }
            if (FILTER && any_amb) {
#pragma unroll
                for (uint32_t k = 0; k < G; k++) if (amb[k]) pass[k] = ray_aabb_nb(o, dd, a[k].x, a[k].y, a[k].z, a[k].w, b[k].x, b[k].y);
            }
#pragma unroll
            for (uint32_t k = 0; k < G; k++) {
                const uint32_t ob = ob0 + k < n_obj ? ob0 + k : n_obj - 1u;
                any |= pass[k];
                if (pass[k] && ob < 32u) pmask |= 1u << ob;
            }
        };
        if (n_obj <= 2u) group(0u, std::integral_constant<uint32_t, 2>());
        else for (uint32_t ob0 = 0; ob0 < n_obj; ob0 += 4u) group(ob0, std::integral_constant<uint32_t, 4>());
    }
    if (root_pass) root_pass[lane] = live ? pmask : 0u;
    const unsigned long long m = __ballot(live && any);
    if (m == 0ull) {
        if (live) {      // what closest_hit_phase writes for a miss (:518, drawImage:476-487)
            const size_t pix = (size_t)r * p.W + px;
            hit_id[pix] = -1;
            t_out[pix] = __builtin_inff();
            if (rgb_linear) { rgb_linear[pix * 3] = 0.0f; rgb_linear[pix * 3 + 1] = 0.0f; rgb_linear[pix * 3 + 2] = 0.0f; }
            if (rgb8) { rgb8[pix * 3] = (uint8_t)(p.bg & 255); rgb8[pix * 3 + 1] = (uint8_t)((p.bg >> 8) & 255); rgb8[pix * 3 + 2] = (uint8_t)((p.bg >> 16) & 255); }
        }
        if (shadow_bits) {
            const size_t tile_index = (size_t)by * gx + bx;
            for (uint32_t l = lane; l < p.n_lights; l += 64) shadow_bits[tile_index * p.n_lights + l] = 0ull;
        }
    }
    return m != 0ull;
}

template <bool FILTER, bool CAM = false>
__device__ __forceinline__ bool finish_background_tile(const DevScene& s, const DevParams& p, int32_t* __restrict__ hit_id, float* __restrict__ t_out,
                                                       float* __restrict__ rgb_linear, uint8_t* __restrict__ rgb8,
                                                       unsigned long long* __restrict__ shadow_bits,
                                                       const uint32_t bx, const uint32_t by, const uint32_t gx,
                                                       uint32_t* root_pass = nullptr) {      // LDS, 64 words (quadrant * 16 + pixel): out, per ray the objects whose root box it passes
    __shared__ uint32_t tile_live;
    if ((threadIdx.x >> 6) == 0) {
        const bool live = background_test_wave<FILTER, CAM>(s, p, hit_id, t_out, rgb_linear, rgb8, shadow_bits, bx, by, gx, root_pass);
        if ((threadIdx.x & 63) == 0) tile_live = live ? 1u : 0u;
    }
    __syncthreads();
    return tile_live == 0u;
}

// COARSE: a workgroup owns 2 x 2 tiles (16 x 16 pixels): its four waves put one tile's 64 rays each through the root boxes at once,
// then work through the live tiles one after the other (four waves per tile, as ever).  For frames that are mostly background
// (3840x2160 of the reference's scenes: 94 % of 129,600 tiles) the launch is bound by workgroup dispatch, and this is a quarter
// of the workgroups; the host picks it for big frames only (a frame full of geometry keeps the finer grid's balance).
template <bool COUNT, int NQCAP, int TWL, int THL, bool FILTER, bool COARSE = false, bool WIDE = false, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void k_closest_hit_nq(DevScene s, DevParams p, int32_t* __restrict__ hit_id,
                                                        float* __restrict__ t_out, float* __restrict__ rgb_linear,
                                                        uint8_t* __restrict__ rgb8, unsigned long long* __restrict__ counters,
                                                        uint32_t* __restrict__ qcount, uint32_t* __restrict__ qlist, uint32_t qcap) {
    constexpr int P = 1 << (TWL + THL);
    __shared__ uint32_t nq_all[4][NQCAP];
    __shared__ uint32_t tq_all[4][LQ_WORDS];
    __shared__ unsigned long long best_all[4][P];
    __shared__ float4 dir_all[4][P];
    const uint32_t wave = threadIdx.x >> 6;
    int32_t id; float t; V3 d;
    if (COARSE && !COUNT && TWL == 2 && THL == 2) {
        __shared__ uint32_t root_pass4[4][64];
        __shared__ uint32_t live4[4];
        const uint32_t gx = (p.W + 7u) / 8u, gy = (p.rows + 7u) / 8u;
        {
            const uint32_t bx = blockIdx.x * 2u + (wave & 1u), by = blockIdx.y * 2u + (wave >> 1);
            bool live = false;
            if (bx < gx && by < gy) live = background_test_wave<FILTER>(s, p, hit_id, t_out, rgb_linear, rgb8, nullptr, bx, by, gx, root_pass4[wave]);
            if ((threadIdx.x & 63) == 0) live4[wave] = live ? 1u : 0u;
        }
        __syncthreads();
        const bool roots_done = s.n_objects <= 32u;
        for (uint32_t k = 0; k < 4u; k++) {
            if (!live4[k]) continue;                                  // workgroup-uniform
            closest_hit_phase<COUNT, NQCAP, TWL, THL, FILTER, false, WIDE>(s, p, nq_all[wave], tq_all[wave], best_all[wave], dir_all[wave],
                                                              hit_id, t_out, rgb_linear, rgb8, counters, id, t, d,
                                                              blockIdx.x * 2u + (k & 1u), blockIdx.y * 2u + (k >> 1), gx, wave, qcount, qlist, qcap,
                                                              roots_done ? root_pass4[k] + wave * 16 : nullptr);
        }
        return;
    }
    __shared__ uint32_t root_pass[64];
    const bool roots_done = !COUNT && TWL == 2 && THL == 2 && s.n_objects <= 32u;      // wave 0 tests every root for the tile's 64 rays first
    if (!COUNT && TWL == 2 && THL == 2 && finish_background_tile<FILTER>(s, p, hit_id, t_out, rgb_linear, rgb8, nullptr, blockIdx.x, blockIdx.y, gridDim.x, root_pass)) return;
    closest_hit_phase<COUNT, NQCAP, TWL, THL, FILTER, false, WIDE>(s, p, nq_all[wave], tq_all[wave], best_all[wave], dir_all[wave],
                                                      hit_id, t_out, rgb_linear, rgb8, counters, id, t, d, blockIdx.x, blockIdx.y, gridDim.x, wave, qcount, qlist, qcap,
                                                      roots_done ? root_pass + wave * 16 : nullptr);
}

// =================================================================================================
// Kernel 2: shadow rays + shading + tone map.  softShadow:348-401 -> shadowIntersection:321-342 +
// phongIllumination:144-200, then the quantiser (:447-449) and the black -> background rule
// (:518, drawImage:476-487).  Shading runs once, for the closest hit (the reference re-shades every
// improving hit and keeps the last: same value).  The hit object's own tree is skipped (the reference
// walks it and discards the result, :328/:331) and the any-hit walk exits at the first hit.
// =================================================================================================
template <bool COUNT>
__device__ __forceinline__ bool any_hit_range(const DevScene& s, int2 self, V3 so, V3 sd,
                                              unsigned long long& n_node, unsigned long long& n_tri) {
    const float4* nodes4 = reinterpret_cast<const float4*>(s.nodes);
    const float4* tris4 = reinterpret_cast<const float4*>(s.tris);
    int32_t i = 0;
    const int32_t n = (int32_t)s.n_nodes;
    while (i < n) {
        if (i == self.x) { i = self.y; continue; }
        const float4 a = nodes4[2 * (size_t)i], b = nodes4[2 * (size_t)i + 1];
        const int32_t skip = __float_as_int(b.z), leaf = __float_as_int(b.w);
        if (COUNT) n_node++;
        if (ray_aabb(so, sd, a.x, a.y, a.z, a.w, b.x, b.y)) {
            if (leaf >= 0) {
                const int32_t first = leaf >> LEAF_SHIFT, cnt = leaf & LEAF_MAX;
                for (int32_t k = 0; k < cnt; k++) {
                    const size_t ti = (size_t)(first + k) * 3;
                    const float4 t0 = tris4[ti], t1 = tris4[ti + 1];
                    const float e2z = reinterpret_cast<const float*>(tris4 + ti + 2)[0];
                    if (COUNT) n_tri++;
                    const float t = ray_triangle(so, sd, mk(t0.x, t0.y, t0.z), mk(t0.w, t1.x, t1.y), mk(t1.z, t1.w, e2z));
                    if (t != SRT_NEG_INF) return true;       // any t >= 0, NaN included (:335)
                }
            }
            i = i + 1;
        } else {
            i = skip;
        }
    }
    return false;
}
template <bool COUNT>
__device__ __forceinline__ bool any_hit(const DevScene& s, int32_t self_obj, V3 so, V3 sd,
                                        unsigned long long& n_node, unsigned long long& n_tri) {
    return any_hit_range<COUNT>(s, s.obj_range[self_obj], so, sd, n_node, n_tri);
}

template <bool COUNT>
__global__ __launch_bounds__(256) void k_shade(DevScene s, DevParams p, const int32_t* __restrict__ hit_id,
                                               const float* __restrict__ t_in, float* __restrict__ rgb_linear,
                                               uint8_t* __restrict__ rgb8, unsigned long long* __restrict__ counters,
                                               unsigned long long* __restrict__ counters_next) {
    if (counters_next) zero_next_counters(counters_next);
    uint32_t px, r;
    const bool live = tile_pixel(p, px, r);
    unsigned long long n_node = 0, n_tri = 0;
    bool is_hit = false;
    if (live) {
        const size_t pix = (size_t)r * p.W + px;
        const int32_t id = hit_id[pix];
        V3 sum = mk(0.0f, 0.0f, 0.0f);
        int q0 = 0, q1 = 0, q2 = 0;
        if (id >= 0) {
            is_hit = true;
            const float t = t_in[pix];
            const V3 o = mk(0.0f, 0.0f, 0.0f);
            const V3 d = primary_dir(p, px, image_row(p, r));
            const int32_t obj = s.tri_obj[id];
            const float4* tp = reinterpret_cast<const float4*>(s.tris) + (size_t)id * 3;
            const float4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
            const V3 nrm = mk(t2.y, t2.z, t2.w);
            V3 color = mk(s.obj_color[obj * 3], s.obj_color[obj * 3 + 1], s.obj_color[obj * 3 + 2]);     // :437-440
            const int32_t tex = s.tri_tex ? s.tri_tex[id] : -1;
            if (tex >= 0) {                                                                             // :350-361
                const V3 P = o + d * t;
                const V3 bc = barycentric(mk(t0.x, t0.y, t0.z), mk(t0.w, t1.x, t1.y), mk(t1.z, t1.w, t2.x), P);
                const float* tc = s.tri_tc + (size_t)id * 6;
                const float tx = (bc.x * tc[0] + bc.y * tc[2]) + bc.z * tc[4];                          // :123-125
                const float ty = (bc.x * tc[1] + bc.y * tc[3]) + bc.z * tc[5];
                long long idx = ((long long)((int)ty * (int)s.tex_w[tex] + (int)tx)) * 3;               // :357
                // the reference reads out of bounds here if the texel index leaves the image (UB);
                // this kernel clamps into the texture instead of faulting
                const long long last = (long long)s.tex_size[tex] - 3;
                idx = idx < 0 ? 0 : (idx > last ? last : idx);
                const uint8_t* td = s.tex + s.tex_off[tex] + idx;
                color = mk(td[0] / 255.0f, td[1] / 255.0f, td[2] / 255.0f);
            }
            const float ka = s.obj_mat[obj * 3], ks = s.obj_mat[obj * 3 + 1], sh = s.obj_mat[obj * 3 + 2];
            const V3 dt = d * t;                      // shadowIntersection:325-326: origin d*t, dir L - d*t
            for (uint32_t l = 0; l < p.n_lights; l++) {                                                 // :366-383
                const V3 L = mk(p.lights[l * 3], p.lights[l * 3 + 1], p.lights[l * 3 + 2]);
                const bool shadowed = any_hit<COUNT>(s, obj, dt, L - dt, n_node, n_tri);
                V3 c = phong(nrm, o, d, L, color, ka, ks, sh, t);
                if (shadowed) c = mk(c.x / p.shadow_div, c.y / p.shadow_div, c.z / p.shadow_div);       // :369
                sum = sum + c;                                                                          // :370
            }
            q0 = quant1(tone1(sum.x, p.reinhard, p.gamma));                                             // :391-398,447-449
            q1 = quant1(tone1(sum.y, p.reinhard, p.gamma));
            q2 = quant1(tone1(sum.z, p.reinhard, p.gamma));
        }
        if (rgb_linear) { rgb_linear[pix * 3] = sum.x; rgb_linear[pix * 3 + 1] = sum.y; rgb_linear[pix * 3 + 2] = sum.z; }
        if (rgb8) {
            if ((q0 | q1 | q2) == 0) { q0 = p.bg & 255; q1 = (p.bg >> 8) & 255; q2 = (p.bg >> 16) & 255; }   // :518, :476-487
            rgb8[pix * 3] = (uint8_t)q0; rgb8[pix * 3 + 1] = (uint8_t)q1; rgb8[pix * 3 + 2] = (uint8_t)q2;
        }
    }
    count_hits(counters, is_hit, blockIdx.y * gridDim.x + blockIdx.x);
    if (COUNT) { wave_add(counters + 3, n_node); wave_add(counters + 4, n_tri); }
}


// =================================================================================================
// Kernel 2a: shadow rays, node-queue form.  A 256-thread workgroup owns an 8x8 pixel tile, each of its
// four wavefronts a 4x4 quadrant.  A wave compacts its hit pixels (ballot + rank) and walks work items
// (hit pixel, light sample) 16 rays at a time with the same two LDS queues as the closest-hit kernel:
// 64 lanes serve 16 rays, so lanes stay full whatever the hit pattern, and with many light samples the
// rays of a step share their origin.  shadowIntersection:321-342: origin d*t, direction L - d*t
// (unnormalised, no epsilon); any candidate of ANOTHER object with Moller-Trumbore != -inf (NaN included)
// shadows.  A hit raises the ray's flag; queued pairs of a flagged ray are dropped when popped.
// Result: per 8x8 tile and light sample one 64-bit word of four 16-bit fields, field = quadrant (wave), bit = y*4 + x inside it.
// SEQ = true is the counting build: per-ray sequential pre-order walk with exit at the first hit, whose
// slab / triangle test counts are the algorithmic counts the CPU oracle mirrors.
// =================================================================================================
// Per-wave LDS of the shadow phase (beside the two queues)
template <int RS>                  // RS = shadow rays in flight per round: 16, or 64 when there are many light samples
struct ShadowLds {
    float4 ray[2 * RS];            // per ray slot: origin, direction
    float4 pixd[NQ_P];             // per hit rank: t, pixel lane, own object's node range
    float4 pso[NQ_P];              // per hit rank: the shadow rays' origin d * t (:326)
    int2 selfr[RS];                // per ray slot: node range of the hit object
    uint32_t flag[RS];
    uint32_t mask[64];             // per light sample of the current group: shadowed pixels of this wave's 4x4 quadrant
};

// Runs per wavefront, with no workgroup-level synchronisation: `id` / `t_hit` are the hit id and t of this lane's pixel
// (lanes < 16; -1 = miss).  The wave writes its own 16-bit field of the tile's word (field = quadrant, bit = pixel lane
// y * 4 + x inside the quadrant), so a wave that is done leaves the CU without waiting for its three neighbours.
template <bool SEQ, int NQCAP, bool FILTER, int RS, bool EARLY = true, bool WIDE = false>
__device__ __forceinline__ void shadow_phase(const DevScene& s, const DevParams& p, uint32_t* nq, uint32_t* tq, ShadowLds<RS>& L,
                                             int32_t id, float t_hit, V3 d_hit,
                                             unsigned long long* __restrict__ shadow_bits, unsigned long long* __restrict__ counters,
                                             const uint32_t bx, const uint32_t by, const uint32_t gx, const uint32_t wave,
                                             const uint32_t l_begin = 0u, const uint32_t l_end_ = 0xffffffffu) {   // light samples [l_begin, l_end) of p.n_lights
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t l_end = l_end_ < p.n_lights ? l_end_ : p.n_lights;
    float4* ray = L.ray;
    int2* selfr = L.selfr;
    uint32_t* flag = L.flag;
    float4* pixd = L.pixd;
    const float4* nodes4 = reinterpret_cast<const float4*>(s.nodes);
    const float4* tris4 = reinterpret_cast<const float4*>(s.tris);
    const uint32_t qx = wave & 1, qy = wave >> 1;
    const uint32_t tile_x = bx * 8 + qx * 4, tile_r = by * 8 + qy * 4;
    const uint32_t px = tile_x + (lane & 3), r = tile_r + ((lane >> 2) & 3);
    const bool live = lane < NQ_P && pixel_live(p, px, r);
    const size_t tile_index = (size_t)by * gx + bx;
    unsigned long long n_node = 0, n_tri = 0;
    if (!live) id = -1;
    const uint32_t hm = (uint32_t)__ballot(id >= 0);
    const uint32_t nh = (uint32_t)__popc(hm);
    int32_t self_root = -2;
    V3 so_mine = mk(0.f, 0.f, 0.f);
    if (id >= 0) {
        const int2 self = s.obj_range[s.tri_obj[id]];
        const uint32_t rank = __popc(hm & ((1u << lane) - 1u));
        pixd[rank] = make_float4(t_hit, __uint_as_float(lane), __int_as_float(self.x), __int_as_float(self.y));
        V3 so_px = d_hit * t_hit;                                      // :326
        if (p.cam) so_px = ray_origin(p) + so_px;                      // camera mode: the primary ray did not start at 0
        L.pso[rank] = make_float4(so_px.x, so_px.y, so_px.z, 0.f);
        self_root = self.x; so_mine = so_px;
    }
    // Quick reject (EARLY): on most quadrants of the ground no shadow ray passes the ROOT box of any other object, and the queue
    // machinery below is a chain of dependent steps to find that out (profiles/diag_timeline_parts.py: the waves of slab tiles hold 22
    // of a CU's 24 wave slots, so a wave's lifetime IS the throughput).  So the wave first puts every (ray, light sample, other
    // object) through that root's slab test -- the test the first node step would run -- at 16 rays x 4 combinations per pass; if
    // nothing passes, the quadrant's shadow words are zero.  K3 with 4 light samples: 0.210 -> 0.187 ms per frame.
    const uint32_t n_combo = (l_end - l_begin) * s.n_objects;
    if (!SEQ && EARLY && l_end > l_begin && (nh == 0u || n_combo <= 64u)) {
        const uint32_t rl = lane & 15u, slot = lane >> 4;
        const float sx = __shfl(so_mine.x, (int)rl, 64), sy = __shfl(so_mine.y, (int)rl, 64), sz = __shfl(so_mine.z, (int)rl, 64);
        const int32_t sroot = __shfl(self_root, (int)rl, 64);
        const V3 so = mk(sx, sy, sz);
        bool pass_any = false;
        if (sroot != -2 && nh) {
            for (uint32_t c = slot; c < n_combo; c += 4u) {
                const uint32_t lq = c / s.n_objects, ob = c - lq * s.n_objects, l = l_begin + lq;
                const int32_t root = s.obj_range[ob].x;
                const float4 a = reinterpret_cast<const float4*>(s.root_nodes)[2 * (size_t)ob], b = reinterpret_cast<const float4*>(s.root_nodes)[2 * (size_t)ob + 1];      // (independent of `root`)
                if (root == sroot) continue;                          // never the hit object's own tree (:331)
                const V3 sd = mk(p.lights[l * 3], p.lights[l * 3 + 1], p.lights[l * 3 + 2]) - so;      // :325
                bool pass;
                if (FILTER) {
                    bool amb;
                    pass = ray_aabb_filtered(so, ray_rcp(sd), a.x, a.y, a.z, a.w, b.x, b.y, amb);
                    if (amb) pass = ray_aabb_nb(so, sd, a.x, a.y, a.z, a.w, b.x, b.y);
                } else pass = ray_aabb_nb(so, sd, a.x, a.y, a.z, a.w, b.x, b.y);
                pass_any |= pass;
            }
        }
        if (!__ballot(pass_any)) {
            L.mask[lane] = 0u;                                        // (read by the wave that shades the tile in the one-launch build)
            if (shadow_bits) for (uint32_t l = l_begin + lane; l < l_end; l += 64u) reinterpret_cast<uint16_t*>(shadow_bits)[(tile_index * p.n_lights + l) * 4 + wave] = 0;
            return;
        }
    }
    uint32_t nqn = 0, tqn = 0;
    unsigned long long dg_k = 0, dg_m = 0, dg_steps = 0;      // (diagnostic, variant 47)

    // one batch of <= 64 queued (leaf, ray) pairs: the lane walks the leaf's triangles until one hits
    auto tri_batch = [&]() {
        const uint32_t m = tqn < 64 ? tqn : 64;
        tqn -= m;
        if (lane < m) {
            const uint32_t info = tq[2 * (tqn + lane)], rs = tq[2 * (tqn + lane) + 1];
            if (!flag[rs]) {
                const uint32_t first = info >> LEAF_SHIFT, cnt = info & LEAF_MAX;
                const float4 ro4 = ray[rs], rd4 = ray[RS + rs];
                const V3 ro = mk(ro4.x, ro4.y, ro4.z), rd = mk(rd4.x, rd4.y, rd4.z);
                const float4* tp = tris4 + (size_t)first * 3;
                float4 t0 = tp[0], t1 = tp[1];
                float e2z = reinterpret_cast<const float*>(tp + 2)[0];
                bool hit = false;
                for (uint32_t k = 0; k < cnt && !hit; k++) {
                    const float4 c0_ = t0, c1_ = t1; const float cz = e2z;
                    if (k + 1 < cnt) { tp += 3; t0 = tp[0]; t1 = tp[1]; e2z = reinterpret_cast<const float*>(tp + 2)[0]; }
                    const float t = ray_triangle(ro, rd, mk(c0_.x, c0_.y, c0_.z), mk(c0_.w, c1_.x, c1_.y), mk(c1_.z, c1_.w, cz));
                    hit = t != SRT_NEG_INF;                          // any t >= 0, NaN included (:335)
                }
                if (hit) flag[rs] = 1u;
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto push_tris = [&](uint32_t info, bool is_leaf, uint32_t rs) {
        const unsigned long long lm = __ballot(is_leaf);
        if (lm) {
            if (is_leaf) {
                const uint32_t pos = tqn + lane_prefix(lm);
                tq[2 * pos] = info; tq[2 * pos + 1] = rs;
            }
            tqn += (uint32_t)__popcll(lm);
            __builtin_amdgcn_wave_barrier();
            while (tqn >= 64) tri_batch();
        }
    };

    auto push_tris2 = [&](uint32_t info_a, bool leaf_a, uint32_t info_b, bool leaf_b, uint32_t rs) {
        const unsigned long long ma = __ballot(leaf_a), mb = __ballot(leaf_b);
        if (ma | mb) {
            uint32_t pos = tqn + lane_prefix(ma) + lane_prefix(mb), q = pos + (leaf_a ? 1u : 0u);
            if (!(p.exp & 1u)) { pos = tqn + lane_prefix(ma); q = tqn + (uint32_t)__popcll(ma) + lane_prefix(mb); }      // node-major
            if (leaf_a) { tq[2 * pos] = info_a; tq[2 * pos + 1] = rs; }
            if (leaf_b) { tq[2 * q] = info_b; tq[2 * q + 1] = rs; }
            tqn += (uint32_t)__popcll(ma) + (uint32_t)__popcll(mb);
            __builtin_amdgcn_wave_barrier();
            while (tqn >= 64) tri_batch();
        }
    };
    const float4* wide4 = reinterpret_cast<const float4*>(s.wide);

    constexpr uint32_t OBJ_G = (NQCAP / (2 * RS)) < 16 ? (NQCAP / (2 * RS)) : 16;     // RS * OBJ_G <= NQCAP / 2
    const uint32_t n_obj = s.n_objects;
    for (uint32_t l0 = l_begin; l0 < l_end; l0 += 64) {              // light samples in groups of 64
        const uint32_t Lg = (l_end - l0) < 64u ? (l_end - l0) : 64u;
        L.mask[lane] = 0u;
        __builtin_amdgcn_wave_barrier();
        const uint32_t n_items = nh * Lg;
        for (uint32_t base = 0; base < n_items; base += RS) {        // RS rays per round
            const uint32_t item = base + lane;
            const bool valid = lane < RS && item < n_items;
            uint32_t pl = 0, lg = 0;
            V3 so = mk(0.f, 0.f, 0.f), sd = mk(0.f, 0.f, 1.f);
            int2 self = make_int2(-1, -1);
            if (valid) {
                const uint32_t hr = (Lg == 1u) ? item : item / Lg;
                lg = item - hr * Lg;
                const float4 pd = pixd[hr];
                pl = __float_as_uint(pd.y);
                self = make_int2(__float_as_int(pd.z), __float_as_int(pd.w));
                const float4 so4 = L.pso[hr];
                const uint32_t l = l0 + lg;
                const V3 L = mk(p.lights[l * 3], p.lights[l * 3 + 1], p.lights[l * 3 + 2]);
                so = mk(so4.x, so4.y, so4.z);
                sd = L - so;                                      // :325
            }
            bool shadowed = false;
            if (SEQ) {
                if (valid) shadowed = any_hit_range<true>(s, self, so, sd, n_node, n_tri);
            } else {
                if (lane < RS) {
                    ray[lane] = make_float4(so.x, so.y, so.z, 0.f);
                    ray[RS + lane] = make_float4(sd.x, sd.y, sd.z, 0.f);
                    selfr[lane] = self;
                    flag[lane] = valid ? 0u : 1u;
                }
                const unsigned long long validm = __ballot(valid);
                __builtin_amdgcn_wave_barrier();
                if constexpr (WIDE) {
                for (uint32_t obj0 = 0; obj0 < n_obj; obj0 += OBJ_G) {
                    const uint32_t g = (n_obj - obj0) < OBJ_G ? (n_obj - obj0) : OBJ_G;
                    // the other objects' roots, one (ray, object) pair per lane: a passing inner root is queued, a passing leaf's triangles are
                    for (uint32_t kb = 0; kb < RS * g; kb += 64) {
                        const uint32_t k = kb + lane;
                        const uint32_t rs = k & (RS - 1), ob = k / RS;
                        bool ok = k < RS * g && ((validm >> rs) & 1ull);
                        int32_t info = 0;
                        if (ok) {
                            const int32_t root = s.obj_range[obj0 + ob].x;
                            ok = root != selfr[rs].x && !flag[rs];                 // never the hit object's own tree (:331)
                            if (ok) {
                                const float4* rn = reinterpret_cast<const float4*>(s.root_nodes) + 2 * (size_t)(obj0 + ob);
                                const float4 a = rn[0], b = rn[1];
                                const float4 o4 = ray[rs], d4 = ray[RS + rs];
                                const V3 ro = mk(o4.x, o4.y, o4.z), rd = mk(d4.x, d4.y, d4.z);
                                ok = slab_pass<FILTER>(ro, rd, ray_rcp(rd), a.x, a.y, a.z, a.w, b.x, b.y);
                                if (ok) info = s.obj_root_info[obj0 + ob];
                            }
                        }
                        const bool inner = ok && info < 0, leafp = ok && info >= 0 && (info & LEAF_MAX) != 0;
                        const unsigned long long im = __ballot(inner);
                        if (inner) nq[nqn + lane_prefix(im)] = ((uint32_t)(~info) << 6) | rs;
                        nqn += (uint32_t)__popcll(im);
                        push_tris2((uint32_t)info, leafp, 0u, false, rs);
                    }
                    __builtin_amdgcn_wave_barrier();
                    while (nqn) {
                        const uint32_t m = nqn < 64 ? nqn : 64;
                        nqn -= m;
                        uint32_t rs = 0;
                        int32_t linfo = 0, rinfo = 0, node = 0, rnode = 0;
                        bool in_l = false, in_r = false, lf_l = false, lf_r = false;
                        V3 ro = mk(0.f, 0.f, 0.f), rd = mk(0.f, 0.f, 1.f);
                        if (p.exp & 4u) {        // diagnostic (variant 47): how many DISTINCT records does a step read?
                            const uint32_t e_ = lane < m ? nq[nqn + lane] : 0xffffffffu;
                            const uint32_t w_ = e_ >> 6, prev_ = (uint32_t)__shfl_up((int)w_, 1, 64);
                            const bool live_ = lane < m && !flag[e_ & 63u];
                            dg_k += (unsigned long long)__popcll(__ballot(live_ && (lane == 0 || w_ != prev_)));
                            dg_m += (unsigned long long)__popcll(__ballot(live_));
                            dg_steps++;
                        }
                        if (lane < m) {
                            const uint32_t e = nq[nqn + lane];
                            rs = e & 63u;
                            if (!flag[rs]) {                          // already shadowed rays drop their queued pairs
                                const float4* wp = wide4 + 4 * (size_t)(e >> 6);
                                const float4 r0 = wp[0], r1 = wp[1], r2 = wp[2];
                                const int4 r3 = reinterpret_cast<const int4*>(wp)[3];
                                const float4 o4 = ray[rs], d4 = ray[RS + rs];
                                ro = mk(o4.x, o4.y, o4.z); rd = mk(d4.x, d4.y, d4.z);
                                linfo = r3.x; rinfo = r3.y; node = r3.z; rnode = r3.w;
                                bool pass_l, pass_r;
                                slab_pass2<FILTER>(ro, rd, ray_rcp(rd), r0, r1, r2, pass_l, pass_r);
                                in_l = pass_l && linfo < 0; lf_l = pass_l && linfo >= 0 && (linfo & LEAF_MAX) != 0;
                                in_r = pass_r && rinfo < 0; lf_r = pass_r && rinfo >= 0 && (rinfo & LEAF_MAX) != 0;
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        const unsigned long long ml = __ballot(in_l), mr = __ballot(in_r);
                        const uint32_t n_in = (uint32_t)__popcll(ml) + (uint32_t)__popcll(mr);
                        if (nqn + n_in <= (uint32_t)NQCAP) {
                            if (!(p.exp & 1u)) {
                                if (in_r) nq[nqn + lane_prefix(mr)] = ((uint32_t)(~rinfo) << 6) | rs;
                                if (in_l) nq[nqn + (uint32_t)__popcll(mr) + lane_prefix(ml)] = ((uint32_t)(~linfo) << 6) | rs;
                            } else {
                            const uint32_t pos = nqn + lane_prefix(ml) + lane_prefix(mr);
                            if (in_r) nq[pos] = ((uint32_t)(~rinfo) << 6) | rs;
                            if (in_l) nq[pos + (in_r ? 1u : 0u)] = ((uint32_t)(~linfo) << 6) | rs;
                            }
                            nqn += n_in;
                            push_tris2((uint32_t)linfo, lf_l, (uint32_t)rinfo, lf_r, rs);
                        } else {
                            // queue full: the passing children's subtrees by the stackless walk (see closest_hit_phase)
                            push_tris2((uint32_t)linfo, lf_l, (uint32_t)rinfo, lf_r, rs);
                            int32_t i = 0, end = 0, hop = -1;
                            if (in_l | in_r) {
                                i = in_l ? node + 2 : rnode + 1;
                                end = in_r ? s.nodes[rnode].skip : rnode;
                                hop = (in_l && in_r) ? rnode : -1;
                            }
                            while (__ballot(i < end)) {
                                int32_t inf2 = -1;
                                bool lp2 = false;
                                if (i < end) {
                                    if (flag[rs]) i = end;
                                    else if (i == hop) i = hop + 1;
                                    else {
                                        const float4 a = nodes4[2 * (size_t)i], b = nodes4[2 * (size_t)i + 1];
                                        const int32_t sk = __float_as_int(b.z);
                                        inf2 = __float_as_int(b.w);
                                        if (ray_aabb_nb(ro, rd, a.x, a.y, a.z, a.w, b.x, b.y)) {
                                            lp2 = inf2 >= 0 && (inf2 & LEAF_MAX) != 0;
                                            i = i + 1;
                                        } else {
                                            i = sk;
                                        }
                                    }
                                }
                                push_tris2((uint32_t)inf2, lp2, 0u, false, rs);
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                } else
                for (uint32_t obj0 = 0; obj0 < n_obj; obj0 += OBJ_G) {
                    const uint32_t g = (n_obj - obj0) < OBJ_G ? (n_obj - obj0) : OBJ_G;
                    for (uint32_t kb = 0; kb < RS * g; kb += 64) {
                        const uint32_t k = kb + lane;
                        const uint32_t rs = k & (RS - 1), ob = k / RS;
                        bool ok = k < RS * g && ((validm >> rs) & 1ull);
                        int32_t root = 0;
                        if (ok) { root = s.obj_range[obj0 + ob].x; ok = root != selfr[rs].x; }   // never the hit object's own tree (:331)
                        const unsigned long long m = __ballot(ok);
                        if (ok) nq[nqn + lane_prefix(m)] = ((uint32_t)root << 6) | rs;
                        nqn += (uint32_t)__popcll(m);
                    }
                    __builtin_amdgcn_wave_barrier();
                    while (nqn) {
                        const uint32_t m = nqn < 64 ? nqn : 64;
                        nqn -= m;
                        uint32_t rs = 0;
                        int32_t node = 0, info = -1, skip = 0;
                        bool inner = false, leafp = false;
                        V3 ro = mk(0.f, 0.f, 0.f), rd = mk(0.f, 0.f, 1.f);
                        if (lane < m) {
                            const uint32_t e = nq[nqn + lane];
                            rs = e & 63u; node = (int32_t)(e >> 6);
                            if (!flag[rs]) {                          // already shadowed rays drop their queued pairs
                                const float4 a = nodes4[2 * (size_t)node], b = nodes4[2 * (size_t)node + 1];
                                const float4 o4 = ray[rs], d4 = ray[RS + rs];
                                ro = mk(o4.x, o4.y, o4.z); rd = mk(d4.x, d4.y, d4.z);
                                skip = __float_as_int(b.z); info = __float_as_int(b.w);
                                bool pass;
                                if (FILTER) {
                                    bool amb;
                                    pass = ray_aabb_filtered(ro, ray_rcp(rd), a.x, a.y, a.z, a.w, b.x, b.y, amb);
                                    if (amb) pass = ray_aabb_nb(ro, rd, a.x, a.y, a.z, a.w, b.x, b.y);
                                } else {
                                    pass = ray_aabb_nb(ro, rd, a.x, a.y, a.z, a.w, b.x, b.y);
                                }
                                if (pass) {
                                    if (info < 0) inner = true;
                                    else leafp = (info & LEAF_MAX) != 0;
                                }
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                        const unsigned long long im = __ballot(inner);
                        const uint32_t n_in = (uint32_t)__popcll(im);
                        if (nqn + 2 * n_in <= (uint32_t)NQCAP) {
                            if (inner) {
                                uint32_t pos = nqn + 2 * lane_prefix(im), pos1 = pos + 1;
                                if (!(p.exp & 1u)) { pos = nqn + lane_prefix(im); pos1 = pos + n_in; }
                                nq[pos] = ((uint32_t)(~info) << 6) | rs;
                                nq[pos1] = ((uint32_t)(node + 1) << 6) | rs;
                            }
                            nqn += 2 * n_in;
                            push_tris((uint32_t)info, leafp, rs);
                        } else {
                            push_tris((uint32_t)info, leafp, rs);
                            int32_t i = inner ? node + 1 : 0, end = inner ? skip : 0;
                            while (__ballot(i < end)) {
                                int32_t inf2 = -1;
                                bool lp2 = false;
                                if (i < end) {
                                    if (flag[rs]) { i = end; }
                                    else {
                                        const float4 a = nodes4[2 * (size_t)i], b = nodes4[2 * (size_t)i + 1];
                                        const int32_t sk = __float_as_int(b.z);
                                        inf2 = __float_as_int(b.w);
                                        if (ray_aabb_nb(ro, rd, a.x, a.y, a.z, a.w, b.x, b.y)) {
                                            lp2 = inf2 >= 0 && (inf2 & LEAF_MAX) != 0;
                                            i = i + 1;
                                        } else {
                                            i = sk;
                                        }
                                    }
                                }
                                push_tris((uint32_t)inf2, lp2, rs);
                            }
                        }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                while (tqn) tri_batch();
                __builtin_amdgcn_wave_barrier();
                shadowed = valid && flag[lane] != 0u;
                __builtin_amdgcn_wave_barrier();
            }
            if (shadowed && valid) atomicOr(&L.mask[lg], 1u << pl);
        }
        __builtin_amdgcn_wave_barrier();
        if (shadow_bits && lane < Lg) reinterpret_cast<uint16_t*>(shadow_bits)[(tile_index * p.n_lights + l0 + lane) * 4 + wave] = (uint16_t)L.mask[lane];
        __builtin_amdgcn_wave_barrier();
    }
    if (SEQ) { wave_add(counters + 3, n_node); wave_add(counters + 4, n_tri); }
    if (!SEQ && (p.exp & 4u) && lane == 0) { atomicAdd(counters + 5, dg_k); atomicAdd(counters + 6, dg_m); atomicAdd(counters + 7, dg_steps); }
}

// blockIdx.z = chunk of `l_chunk` light samples: with many samples a tile's shadow rays are cut over several workgroups (a
// wave that walks 16 pixels x 64 samples through dense geometry alone can outlast the rest of the launch)
template <bool SEQ, int NQCAP, bool FILTER, int RS = 16, int MINW = 1, bool WIDE = true>
__global__ __launch_bounds__(256, MINW) void k_shadow_nq(DevScene s, DevParams p, const int32_t* __restrict__ hit_id,
                                                   const float* __restrict__ t_in, unsigned long long* __restrict__ shadow_bits,
                                                   unsigned long long* __restrict__ counters, uint32_t l_chunk = 0xffffffffu) {
    __shared__ uint32_t nq_all[4][NQCAP];
    __shared__ uint32_t tq_all[4][LQ_WORDS];
    __shared__ ShadowLds<RS> lds_all[4];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // p.xcd_rows (scenes far bigger than an L2; the host pads gridDim.y to a multiple of 8): XCD j walks tile rows j, j + 8, ... left to
    // right, as the closest-hit kernels of such scenes do -- neighbouring tiles' shadow rays read the same records in the same L2
    const uint32_t gx = gridDim.x;
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_rows) {
        const uint32_t w = blockIdx.y * gx + blockIdx.x, idx = w >> 3;
        by = (idx / gx) * 8u + (w & 7u); bx = idx % gx;
        if (by >= (p.rows + 7u) / 8u) return;
    }
    const uint32_t px = bx * 8 + (wave & 1) * 4 + (lane & 3), r = by * 8 + (wave >> 1) * 4 + ((lane >> 2) & 3);
    int32_t id = -1; float t = 0.f;
    V3 d = mk(0.f, 0.f, p.focal);
    if (lane < NQ_P && pixel_live(p, px, r)) { id = hit_id[(size_t)r * p.W + px]; t = t_in[(size_t)r * p.W + px]; d = primary_dir(p, px, image_row(p, r)); }
    const uint32_t l_begin = l_chunk == 0xffffffffu ? 0u : blockIdx.z * l_chunk;
    shadow_phase<SEQ, NQCAP, FILTER, RS, true, WIDE>(s, p, nq_all[wave], tq_all[wave], lds_all[wave], id, t, d, shadow_bits, counters, bx, by, gx, wave,
                                         l_begin, l_chunk == 0xffffffffu ? 0xffffffffu : l_begin + l_chunk);
}

// One hit pixel: `shadowed(l)` says whether light sample l's shadow ray was blocked.
template <bool INT_SHIN = false, typename SH>
__device__ __forceinline__ void shade_hit_pixel(const DevScene& s, const DevParams& p, const int32_t id, const float t, const uint32_t px, const uint32_t r,
                                                SH shadowed, float* __restrict__ rgb_linear, uint8_t* __restrict__ rgb8) {
    const size_t pix = (size_t)r * p.W + px;
    const V3 o = ray_origin(p);
    const V3 d = primary_dir(p, px, image_row(p, r));
    const int32_t obj = s.tri_obj[id];
    const float4* tp = reinterpret_cast<const float4*>(s.tris) + (size_t)id * 3;
    const float4 t2 = tp[2];
    const V3 nrm = mk(t2.y, t2.z, t2.w);
    V3 color = mk(s.obj_color[obj * 3], s.obj_color[obj * 3 + 1], s.obj_color[obj * 3 + 2]);     // :437-440
    const int32_t tex = s.tri_tex ? s.tri_tex[id] : -1;
    if (tex >= 0) {                                                                             // :350-361
        const float4 t0 = tp[0], t1 = tp[1];
        const V3 P = o + d * t;
        const V3 bc = barycentric(mk(t0.x, t0.y, t0.z), mk(t0.w, t1.x, t1.y), mk(t1.z, t1.w, t2.x), P);
        const float* tc = s.tri_tc + (size_t)id * 6;
        const float tx = (bc.x * tc[0] + bc.y * tc[2]) + bc.z * tc[4];                          // :123-125
        const float ty = (bc.x * tc[1] + bc.y * tc[3]) + bc.z * tc[5];
        long long idx = ((long long)((int)ty * (int)s.tex_w[tex] + (int)tx)) * 3;               // :357
        // the reference reads out of bounds if the texel index leaves the image (UB); clamp instead
        const long long last = (long long)s.tex_size[tex] - 3;
        idx = idx < 0 ? 0 : (idx > last ? last : idx);
        const uint8_t* td = s.tex + s.tex_off[tex] + idx;
        color = mk(td[0] / 255.0f, td[1] / 255.0f, td[2] / 255.0f);
    }
    const float ka = s.obj_mat[obj * 3], ks = s.obj_mat[obj * 3 + 1], sh = s.obj_mat[obj * 3 + 2];
    V3 nrm_use = nrm;
    if (p.smooth) {        // phongIllumination:159,162 with the interpolateNormal line enabled (opt-in mode)
        const float4 t0 = tp[0], t1 = tp[1];
        const V3 bc = barycentric(mk(t0.x, t0.y, t0.z), mk(t0.w, t1.x, t1.y), mk(t1.z, t1.w, t2.x), o + d * t);
        const float* n9 = s.tri_normals + (size_t)id * 9;
        nrm_use = normalize3(mk((bc.x * n9[0] + bc.y * n9[3]) + bc.z * n9[6], (bc.x * n9[1] + bc.y * n9[4]) + bc.z * n9[7],
                                (bc.x * n9[2] + bc.y * n9[5]) + bc.z * n9[8]));
    }
    V3 sum = mk(0.0f, 0.0f, 0.0f);
    for (uint32_t l = 0; l < p.n_lights; l++) {                                                 // :366-383
        const V3 L = mk(p.lights[l * 3], p.lights[l * 3 + 1], p.lights[l * 3 + 2]);
        const bool sd = shadowed(l);
        V3 c = phong<INT_SHIN>(nrm_use, o, d, L, color, ka, ks, sh, t);
        if (sd) c = mk(c.x / p.shadow_div, c.y / p.shadow_div, c.z / p.shadow_div);             // :369
        sum = sum + c;                                                                          // :370
    }
    int q0 = quant1(tone1(sum.x, p.reinhard, p.gamma));                                         // :391-398,447-449
    int q1 = quant1(tone1(sum.y, p.reinhard, p.gamma));
    int q2 = quant1(tone1(sum.z, p.reinhard, p.gamma));
    if (rgb_linear) { rgb_linear[pix * 3] = sum.x; rgb_linear[pix * 3 + 1] = sum.y; rgb_linear[pix * 3 + 2] = sum.z; }
    if (rgb8) {
        if ((q0 | q1 | q2) == 0) { q0 = p.bg & 255; q1 = (p.bg >> 8) & 255; q2 = (p.bg >> 16) & 255; }   // :518, :476-487
        rgb8[pix * 3] = (uint8_t)q0; rgb8[pix * 3 + 1] = (uint8_t)q1; rgb8[pix * 3 + 2] = (uint8_t)q2;
    }
}

// =================================================================================================
// Kernel 1+2a fused (shipped): the wavefront that found its 4x4 tile's closest hits goes straight on to the
// tile's shadow rays -- hit ids, t and the hit object are still in registers, the queues are reused, and
// a frame is two launches (this + shading).  Workgroup = 8x8 pixel tile, 4 waves.
// =================================================================================================
// ROOTS_AGAIN: the round-1 form (every wave re-tests the roots), for A/B
// SHADE: the frame in ONE launch.  The tile's four waves finish at different times and none waits: each leaves its 16 hit ids / t
// in LDS (its shadow masks are there already) and bumps an LDS counter, and the wave that finds the other three done shades the
// tile's 64 pixels, one per lane -- what k_shade_tile does in a second launch from hit ids, t and shadow words re-read from
// memory.  Up to 63 light samples (one group of masks); `shadow_bits` may be null then (nobody reads the words).
// WIDE (the 64 B inner-node records): measured per kernel -- the stand-alone shadow kernel, whose rays cross a
// soup's overlapping boxes hundreds of nodes deep, gains 10 % from it; the fused kernel and the closest-hit kernel, whose frames are
// mostly short waves, lose 5 % (16 VGPRs of record per lane instead of 8: spills at six waves per SIMD), so they keep the 32 B records.
template <bool COUNT, int NQCAP, bool FILTER, int RS, bool XCD_ROWS, bool ROOTS_AGAIN, bool SHADE = false, bool CAM = false, bool WIDE = false, bool ROW_Z = false>
__device__ __forceinline__ void trace_nq_body(const DevScene& s, const DevParams& p, int32_t* __restrict__ hit_id, float* __restrict__ t_out,
                                              float* __restrict__ rgb_linear, uint8_t* __restrict__ rgb8,
                                              unsigned long long* __restrict__ shadow_bits, unsigned long long* __restrict__ counters,
                                              unsigned long long* __restrict__ counters_next = nullptr, uint32_t* __restrict__ qcount = nullptr) {
    __shared__ uint32_t fin_count;
    __shared__ int32_t fin_id[64];
    __shared__ float fin_t[64];
    if (SHADE) {
        if (threadIdx.x == 0) fin_count = 0u;              // (the barrier in finish_background_tile orders this before the first bump)
        if (counters_next) zero_next_counters(counters_next);
        if (qcount && blockIdx.x == 0 && blockIdx.y == 0) for (uint32_t i = threadIdx.x; i < (uint32_t)QL_COUNTERS; i += 256u) qcount[i * QL_STRIDE] = 0u;
    }
    __shared__ uint32_t nq_all[4][NQCAP];
    __shared__ uint32_t tq_all[4][LQ_WORDS];
    __shared__ unsigned long long best_all[4][NQ_P];
    __shared__ float4 dir_all[4][NQ_P];
    __shared__ ShadowLds<RS> lds_all[4];
    const uint32_t wave = threadIdx.x >> 6;
    int32_t id; float t; V3 d;
    unsigned long long k0 = 0, k1 = 0; (void)k0; (void)k1;
    SRT_STAMP(k0);
    // Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one, each XCD has its own 4 MiB L2).  When the
    // scene's records are far bigger than an L2 (XCD_ROWS; the host pads the grid's y extent to a multiple of 8 rows), every
    // XCD gets whole rows of tiles -- XCD j walks tile rows j, j + 8, ... left to right -- so that neighbouring tiles, which
    // read the same nodes and triangles, hit in the same L2: 1 M-triangle soup -15 %.  For a scene of a few MB the plain
    // order is as good or slightly better (K3: +1 % with the row deal, +2 % with a run-time switch), so it has its own build.
    const uint32_t gx = gridDim.x;
    uint32_t bx = blockIdx.x, by = ROW_Z ? blockIdx.z : blockIdx.y;      // ROW_Z: a batch launched with the tile row in z and the frame in y
    if (XCD_ROWS) {
        const uint32_t w = blockIdx.y * gx + blockIdx.x, idx = w >> 3;
        by = (idx / gx) * 8u + (w & 7u); bx = idx % gx;
        if (by >= (p.rows + 7u) / 8u) return;
    }
    __shared__ uint32_t root_pass[64];
    const bool roots_done = !COUNT && !ROOTS_AGAIN && s.n_objects <= 32u;              // wave 0 tests every root for the tile's 64 rays first
    if (!COUNT && finish_background_tile<FILTER, CAM>(s, p, hit_id, t_out, rgb_linear, rgb8, shadow_bits, bx, by, gx, root_pass)) return;
    unsigned long long ka = 0, kb = 0; (void)ka; (void)kb;
    SRT_STAMP(ka);
    closest_hit_phase<COUNT, NQCAP, 2, 2, FILTER, CAM, WIDE>(s, p, nq_all[wave], tq_all[wave], best_all[wave], dir_all[wave],
                                                       hit_id, t_out, rgb_linear, rgb8, counters, id, t, d, bx, by, gx, wave, nullptr, nullptr, 0,
                                                       roots_done ? root_pass + wave * 16 : nullptr);
    __builtin_amdgcn_wave_barrier();
    SRT_STAMP(kb);
    shadow_phase<COUNT, NQCAP, FILTER, RS, true, WIDE>(s, p, nq_all[wave], tq_all[wave], lds_all[wave], id, t, d, shadow_bits, counters, bx, by, gx, wave);
#ifdef SRT_DIAG
    SRT_STAMP(k1); diag_tile_record(rgb_linear, blockIdx.x, blockIdx.y, gridDim.x, k0, k1, ka, kb);
#endif
    if (SHADE) {
        const uint32_t lane = threadIdx.x & 63;
        if (lane < NQ_P) { fin_id[wave * NQ_P + lane] = id; fin_t[wave * NQ_P + lane] = t; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        uint32_t before = 0;
        if (lane == 0) before = atomicAdd(&fin_count, 1u);
        before = __builtin_amdgcn_readfirstlane(before);
        if (before != 3u) return;                          // a neighbour is still walking: it will shade the tile
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t q = lane >> 4, pl = lane & 15u;       // lane -> (quadrant = the wave that traced it, pixel of the quadrant)
        const uint32_t px = bx * 8 + (q & 1u) * 4 + (pl & 3u), r = by * 8 + (q >> 1) * 4 + (pl >> 2);
        const int32_t hid = fin_id[lane];
        if (hid >= 0 && pixel_live(p, px, r)) {
            const uint32_t* mask = lds_all[q].mask;
            shade_hit_pixel(s, p, hid, fin_t[lane], px, r, [&](uint32_t l) -> bool { return (mask[l] >> pl) & 1u; }, rgb_linear, rgb8);
        }
    }
}

template <bool COUNT, int NQCAP, bool FILTER, int MINW, int RS, bool XCD_ROWS = false, bool ROOTS_AGAIN = false, bool CAM = false, bool WIDE = false>
__global__ __launch_bounds__(256, MINW) void k_trace_nq(DevScene s, DevParams p, int32_t* __restrict__ hit_id, float* __restrict__ t_out,
                                                  float* __restrict__ rgb_linear, uint8_t* __restrict__ rgb8,
                                                  unsigned long long* __restrict__ shadow_bits, unsigned long long* __restrict__ counters) {
    trace_nq_body<COUNT, NQCAP, FILTER, RS, XCD_ROWS, ROOTS_AGAIN, false, CAM, WIDE>(s, p, hit_id, t_out, rgb_linear, rgb8, shadow_bits, counters);
}
// closest hit, shadow rays and shading of a frame in one launch (SHADE)
template <int NQCAP, bool FILTER, int MINW, int RS, bool XCD_ROWS = false>
__global__ __launch_bounds__(256, MINW) void k_trace_shade_nq(DevScene s, DevParams p, int32_t* __restrict__ hit_id, float* __restrict__ t_out,
                                                        float* __restrict__ rgb_linear, uint8_t* __restrict__ rgb8,
                                                        unsigned long long* __restrict__ counters, unsigned long long* __restrict__ counters_next,
                                                        uint32_t* __restrict__ qcount) {
    trace_nq_body<false, NQCAP, FILTER, RS, XCD_ROWS, false, true>(s, p, hit_id, t_out, rgb_linear, rgb8, nullptr, counters, counters_next, qcount);
}

// ---- several frames in ONE launch (srt_render_device_batch): blockIdx.z selects the frame ------------------------------------
// The frames of an orbit are independent renders (own scene, own parameters, own outputs).  A tile on the bunny's silhouette keeps
// its workgroup busy for ~60 us (~80 with the CU fully occupied), and the chip holds 1,536 workgroups of this kernel at a time: a
// launch lasts a whole number of such rounds.  The 1080p frame has ~2,500 heavy tiles (1.6 rounds in 0.125 ms); the eighth of it
// that one of eight GPUs owns has ~310 and still takes 0.072 ms as its own launch, nine eighths launched together take 0.199 ms
// (two rounds) where 0.142 would be their share -- measured with tools/strip_probe.py, whatever the height of the scanline blocks.
// With ALL frames of a step in one grid the rounds are filled.  A frame's arguments come from a table in device memory (368 bytes
// a frame; the kernarg segment's 4 KiB would hold ten), read with scalar loads: the index is blockIdx.z, the table is read-only.
struct FrameItem {
    DevScene s; DevParams p;
    int32_t* hit_id; float* t_out; float* rgb_linear; uint8_t* rgb8;
    unsigned long long* shadow_bits; unsigned long long* counters; unsigned long long* counters_next; uint32_t* qcount;
    uint32_t* qlist; uint32_t qcap, pad_;      // the frame's quadrant list (8+-sample pipeline)
};
static_assert(sizeof(FrameItem) == sizeof(DevScene) + sizeof(DevParams) + 9 * 8 + 8, "FrameItem has no implicit padding");
// The frames' arguments travel BY VALUE, in the kernel-argument segment (round 3).  Rounds 1-2 kept the table in device memory; a pointer
// that comes out of memory is a GENERIC pointer to the compiler, every access through it a flat_load / flat_store (both wait counters,
// no scalar form), where the same body with its arguments by value reads wave-uniform records with s_load and the rest with
// global_load -- measured on K4 with one whole frame per "batch": k_shadow_pk_batch 1.57 ms against k_shadow_pk 1.10, closest hit 0.257
// against 0.219, shading 0.135 against 0.122.  Pointer members of a by-value kernel argument are global pointers to the compiler
// (also behind a dynamic index), nothing has to be allocated, copied or kept alive for a captured graph.  The segment is ordinary
// memory on this stack (a 26 KB argument was launched and read back on the MI355X; there is no 4 KB ceiling as on other runtimes):
// 36 frames (the reference's orbit) = 15 KB a launch, a batch of more is issued as several launches -- and every launch boundary costs: the 36 share-frames of a
// K3 step at N = 8 in five launches of 8 took 0.81 ms a step, in one launch 0.64.
constexpr uint32_t FRAME_TAB_MAX = 36;
struct FrameTab { FrameItem it[FRAME_TAB_MAX]; };
static_assert(sizeof(FrameTab) <= 16384, "the frame table is copied into the kernel-argument segment at every launch");
// the unfused closest-hit launch of the 8+-sample pipeline over the frames of a batch (k_closest_hit_nq<false, NQCAP, 2, 2, FILTER>)
// ROW_Z: grid (tiles per row, frames, tile rows) instead of (tiles per row, tile rows, frames): workgroups are dispatched x, then y, then
// z, so the SAME tile row of all the batch's frames is in flight together -- frames of one scene read the same records there, where a
// share's consecutive tile rows (64 image rows apart at N = 8) read different ones.  Same box, shares of 8: K4 2.03 -> 1.86 ms per step,
// K3 0.625 -> 0.599, a quarter of K3 1.179 -> 1.128.  (The bench's frames of a step are one picture, which flatters this; the frames of
// an orbit are 10 degrees apart and still look at the same part of the scene in the same rows.)  Shipped as the only order.
template <int NQCAP, bool FILTER, bool ROW_Z = false>
__global__ __launch_bounds__(256, 7) void k_closest_hit_nq_batch(const FrameTab tab) {
    const FrameItem& it = tab.it[ROW_Z ? blockIdx.y : blockIdx.z];
    const uint32_t tile_row = ROW_Z ? blockIdx.z : blockIdx.y;
    __shared__ uint32_t nq_all[4][NQCAP];
    __shared__ uint32_t tq_all[4][LQ_WORDS];
    __shared__ unsigned long long best_all[4][NQ_P];
    __shared__ float4 dir_all[4][NQ_P];
    __shared__ uint32_t root_pass[64];
    const uint32_t wave = threadIdx.x >> 6;
    int32_t id; float t; V3 d;
    const bool roots_done = it.s.n_objects <= 32u;
    if (finish_background_tile<FILTER>(it.s, it.p, it.hit_id, it.t_out, it.rgb_linear, it.rgb8, nullptr, blockIdx.x, tile_row, gridDim.x, root_pass)) return;
    closest_hit_phase<false, NQCAP, 2, 2, FILTER, false, false>(it.s, it.p, nq_all[wave], tq_all[wave], best_all[wave], dir_all[wave],
                                                  it.hit_id, it.t_out, it.rgb_linear, it.rgb8, it.counters, id, t, d, blockIdx.x, tile_row, gridDim.x, wave,
                                                  it.qcount, it.qlist, it.qcap, roots_done ? root_pass + wave * 16 : nullptr);
}

template <int NQCAP, bool FILTER, int MINW, int RS, bool ROW_Z = false>
__global__ __launch_bounds__(256, MINW) void k_trace_nq_batch(const FrameTab tab) {
    const FrameItem& it = tab.it[ROW_Z ? blockIdx.y : blockIdx.z];
    trace_nq_body<false, NQCAP, FILTER, RS, false, false, false, false, false, ROW_Z>(it.s, it.p, it.hit_id, it.t_out, it.rgb_linear, it.rgb8, it.shadow_bits, it.counters);
}

// =================================================================================================
// Kernel 2b: shading, one thread per hit pixel (same tiles).  softShadow:348-401 (texture fetch,
// light-sample loop with in-order f32 sum, /5 for shadowed samples, Reinhard + gamma),
// phongIllumination:144-200, quantiser :447-449, black -> background (:518, drawImage:476-487).
// Pure ALU, no traversal; misses were finished by the closest-hit kernel.
// =================================================================================================
template <bool INT_SHIN = false>
__device__ __forceinline__ void shade_tile_body(const DevScene& s, const DevParams& p, const int32_t* __restrict__ hit_id,
                                                const float* __restrict__ t_in,
                                                const unsigned long long* __restrict__ shadow_bits,
                                                float* __restrict__ rgb_linear, uint8_t* __restrict__ rgb8,
                                                unsigned long long* __restrict__ counters_next, uint32_t* __restrict__ qcount,
                                                unsigned long long* __restrict__ counters) {
    __shared__ uint32_t wg_hits, wg_done;
    if (threadIdx.x == 0) { wg_hits = 0u; wg_done = 0u; }
    __syncthreads();
    if (counters_next) zero_next_counters(counters_next);
    // the quadrant list of this frame has been consumed by the shadow kernel before this launch: empty it for the next one
    if (qcount && blockIdx.x == 0 && blockIdx.y == 0) for (uint32_t i = threadIdx.x; i < (uint32_t)QL_COUNTERS; i += 256u) qcount[i * QL_STRIDE] = 0u;      // list lengths + units handed out
#ifdef SRT_DIAG
    rgb_linear = nullptr;       // holds the trace kernel's stamps in the diagnostic build
#endif
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t px, r;
    const bool live = tile_pixel(p, px, r);
    const size_t pix = (size_t)r * p.W + px;
    const int32_t id = live ? hit_id[pix] : -1;
    const size_t tile_index = (size_t)(blockIdx.y * 2 + (wave >> 1)) * ((p.W + 7) / 8) + blockIdx.x * 2 + (wave & 1);   // 8x8 tile
    // The frame's hit statistic, here, where the launch overlaps the next frame's tracing.  The closest-hit phase used to add one global
    // atomic per 4x4 quadrant with hits (56 k a K3 frame): 4 % of the trace launch; one per 8x8 tile here (14 k) still made this launch
    // 3.5 us longer.  So the workgroup's four tiles are summed in LDS and the wave that arrives last adds the sum (4 k atomics a frame).
    if (counters && !(p.exp & 32u)) {
        const unsigned long long hm = __ballot(id >= 0);
        if (lane == 0) {
            if (hm) atomicAdd(&wg_hits, (uint32_t)__popcll(hm));
            if (atomicAdd(&wg_done, 1u) == 3u) {
                const uint32_t total = atomicAdd(&wg_hits, 0u);
                if (total) atomicAdd(counters + CTR_HIT_BASE + 8 * ((blockIdx.y * gridDim.x + blockIdx.x) & (HIT_SHARDS - 1)), (unsigned long long)total);
            }
        }
    }
    if (id < 0) return;
    const float t = t_in[pix];
    // shadow bits, tile-major (node-queue kernels): per tile and light sample one word, one 16-bit field per 4x4 quadrant (the wave
    // that traced it), bit = y * 4 + x inside the quadrant; pixel-major (packet shadow kernel): per pixel one word per 64 samples
    const uint32_t n_lch = (p.n_lights + 63u) >> 6;
    const unsigned long long* sb = p.shadow_px_major ? shadow_bits + pix * n_lch : shadow_bits + tile_index * p.n_lights;
    const uint32_t sbit = ((((lane >> 5) & 1u) * 2u + ((lane >> 2) & 1u)) << 4) + ((lane >> 3) & 3u) * 4u + (lane & 3u);
    unsigned long long word = 0ull;                        // pixel-major: the pixel's shadow bits of the current 64 samples, loaded once per 64
    const bool px_major = p.shadow_px_major != 0u;
    shade_hit_pixel<INT_SHIN>(s, p, id, t, px, r, [&](uint32_t l) -> bool {
        if (px_major) {
            if ((l & 63u) == 0u) word = sb[l >> 6];
            return (word >> (l & 63u)) & 1ull;
        }
        return (sb[l] >> sbit) & 1ull;
    }, rgb_linear, rgb8);
}

// INT_SHIN: every object of the scene has an integer shininess in [1, 64] (the reference's default is 15): the per-sample pow is the
// square-and-multiply form alone, and the kernel is built for eight waves per SIMD (the general pow's f64 temporaries cost 100 VGPRs)
template <int INT_SHIN>      // (an int: rocprofv3 consumers take "<true" in a kernel's name for a counting build)
__global__ __launch_bounds__(256, INT_SHIN ? 8 : 1) void k_shade_tile(DevScene s, DevParams p, const int32_t* __restrict__ hit_id,
                                                    const float* __restrict__ t_in,
                                                    const unsigned long long* __restrict__ shadow_bits,
                                                    float* __restrict__ rgb_linear, uint8_t* __restrict__ rgb8,
                                                    unsigned long long* __restrict__ counters_next, uint32_t* __restrict__ qcount,
                                                    unsigned long long* __restrict__ counters) {
    shade_tile_body<INT_SHIN != 0>(s, p, hit_id, t_in, shadow_bits, rgb_linear, rgb8, counters_next, qcount, counters);
}
template <int INT_SHIN>
__global__ __launch_bounds__(256, INT_SHIN ? 8 : 1) void k_shade_tile_batch(const FrameTab tab) {      // the shading of the frames k_trace_nq_batch traced
    const FrameItem& it = tab.it[blockIdx.z];
    shade_tile_body<INT_SHIN != 0>(it.s, it.p, it.hit_id, it.t_out, it.shadow_bits, it.rgb_linear, it.rgb8, it.counters_next, it.qcount, it.counters);
}

// =================================================================================================
// Supersampling extension (spp = n x n > 1; not in the reference): the sub-frames' pre-tone-map
// sums are added in sub-sample order, divided by spp, then tone-mapped / quantised once.
// =================================================================================================
__global__ __launch_bounds__(256) void k_accumulate(float* __restrict__ acc, const float* __restrict__ sub, uint32_t n, int first) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) acc[i] = first ? sub[i] : acc[i] + sub[i];
}
__global__ __launch_bounds__(256) void k_resolve(DevParams p, const float* __restrict__ acc, float spp, uint32_t n_pixels,
                                                 float* __restrict__ rgb_linear, uint8_t* __restrict__ rgb8,
                                                 unsigned long long* __restrict__ counters_next) {
    if (blockIdx.x == 0) for (int i = threadIdx.x; i < NCTR; i += 256) counters_next[i] = 0ull;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pixels) return;
    const float a0 = acc[(size_t)i * 3] / spp, a1 = acc[(size_t)i * 3 + 1] / spp, a2 = acc[(size_t)i * 3 + 2] / spp;
    if (rgb_linear) { rgb_linear[(size_t)i * 3] = a0; rgb_linear[(size_t)i * 3 + 1] = a1; rgb_linear[(size_t)i * 3 + 2] = a2; }
    if (rgb8) {
        int q0 = quant1(tone1(a0, p.reinhard, p.gamma)), q1 = quant1(tone1(a1, p.reinhard, p.gamma)), q2 = quant1(tone1(a2, p.reinhard, p.gamma));
        if ((q0 | q1 | q2) == 0) { q0 = p.bg & 255; q1 = (p.bg >> 8) & 255; q2 = (p.bg >> 16) & 255; }
        rgb8[(size_t)i * 3] = (uint8_t)q0; rgb8[(size_t)i * 3 + 1] = (uint8_t)q1; rgb8[(size_t)i * 3 + 2] = (uint8_t)q2;
    }
}

// =================================================================================================
// f1, the device half of the per-frame rebuild (srt_scene_update_frame): the records of a frame from the transformed points in
// source order, the build's permutation and the node boxes.
// =================================================================================================
// Ray-independent prefix of rayTriangleIntersection:45-51 and calculateTriangleNormal:32-37 -- ONE definition for the host
// (srt_scene_create) and the device (srt_scene_update_frame); both are compiled with -ffp-contract=off and correctly rounded divide /
// sqrt, so the records are the same bits either way.
__host__ __device__ inline DevTri derive_triangle(const float* p) {
    DevTri t;
    const float P1x = p[0] / p[3], P1y = p[1] / p[3], P1z = p[2] / p[3];
    const float P2x = p[4] / p[7], P2y = p[5] / p[7], P2z = p[6] / p[7];
    const float P3x = p[8] / p[11], P3y = p[9] / p[11], P3z = p[10] / p[11];
    t.p1x = P1x; t.p1y = P1y; t.p1z = P1z;
    t.e1x = P2x - P1x; t.e1y = P2y - P1y; t.e1z = P2z - P1z;
    t.e2x = P3x - P1x; t.e2y = P3y - P1y; t.e2z = P3z - P1z;
    const float ax = p[4] - p[0], ay = p[5] - p[1], az = p[6] - p[2];       // raw xyz, no w-divide (:33-34)
    const float bx = p[8] - p[0], by = p[9] - p[1], bz = p[10] - p[2];
    const float cx = ay * bz - by * az, cy = az * bx - bz * ax, cz = ax * by - bx * ay;
    const float s = 1.0f / sqrtf((cx * cx + cy * cy) + cz * cz);
    t.nx = cx * s; t.ny = cy * s; t.nz = cz * s;
    return t;
}
// tvec = o - P1 with o = 0 (:53) and qvec = cross(tvec, e1) (:58, glm::cross) for rays from the origin
__host__ __device__ inline DevTriO derive_triangle_origin(const DevTri& t) {
    DevTriO r;
    r.tx = 0.0f - t.p1x; r.ty = 0.0f - t.p1y; r.tz = 0.0f - t.p1z;
    r.e1x = t.e1x; r.e1y = t.e1y; r.e1z = t.e1z; r.e2x = t.e2x; r.e2y = t.e2y; r.e2z = t.e2z;
    r.qx = r.ty * t.e1z - t.e1y * r.tz;
    r.qy = r.tz * t.e1x - t.e1z * r.tx;
    r.qz = r.tx * t.e1y - t.e1x * r.ty;
    return r;
}

// visit-order triangle g <- source triangle first[obj] + order[g]: records derived, attributes permuted
__global__ __launch_bounds__(256) void k_update_tris(uint32_t n_tris, const int32_t* __restrict__ tri_obj, const int32_t* __restrict__ obj_tri_first,
                                                     const float4* __restrict__ src_points, const uint32_t* __restrict__ order,
                                                     DevTri* __restrict__ tris, DevTriO* __restrict__ tris_o,
                                                     const float* __restrict__ src_tc, float* __restrict__ tri_tc,
                                                     const float* __restrict__ src_nrm, float* __restrict__ tri_nrm,
                                                     const int32_t* __restrict__ src_tex, int32_t* __restrict__ tri_tex) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g >= n_tris) return;
    const int32_t ob = tri_obj[g], f0 = obj_tri_first[ob], cnt = obj_tri_first[ob + 1] - f0;
    const uint32_t oi = order[g];
    const size_t src = (size_t)f0 + (oi < (uint32_t)cnt ? oi : (uint32_t)(cnt - 1));      // (an index outside the object -- a caller's error -- must not become an access outside the buffers)
    const float4 a = src_points[3 * src], b = src_points[3 * src + 1], c = src_points[3 * src + 2];
    const float p[12] = { a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w };
    const DevTri t = derive_triangle(p);
    tris[g] = t;
    tris_o[g] = derive_triangle_origin(t);
    if (tri_tc && src_tc) for (int k = 0; k < 6; k++) tri_tc[(size_t)g * 6 + k] = src_tc[src * 6 + k];
    if (tri_nrm && src_nrm) for (int k = 0; k < 9; k++) tri_nrm[(size_t)g * 9 + k] = src_nrm[src * 9 + k];
    if (tri_tex) tri_tex[g] = src_tex ? src_tex[src] : -1;
}
// node i: its box into the 32 B record (skip / leaf stay), into its parent's 64 B record (the record of inner node i holds its
// CHILDREN's boxes: written by the parent's thread from the box array), and the roots into the contiguous root table
__global__ __launch_bounds__(256) void k_update_nodes(uint32_t n_nodes, const float* __restrict__ box_min, const float* __restrict__ box_max,
                                                      DevNode* __restrict__ nodes, DevWide* __restrict__ wide, const int32_t* __restrict__ widx) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_nodes) return;
    DevNode n = nodes[i];
    n.minx = box_min[3 * (size_t)i]; n.miny = box_min[3 * (size_t)i + 1]; n.minz = box_min[3 * (size_t)i + 2];
    n.maxx = box_max[3 * (size_t)i]; n.maxy = box_max[3 * (size_t)i + 1]; n.maxz = box_max[3 * (size_t)i + 2];
    nodes[i] = n;
    if (n.leaf < 0) {
        const size_t l = (size_t)i + 1, r = (size_t)(~n.leaf);
        DevWide& q = wide[widx[i]];
        q.lminx = box_min[3 * l]; q.lminy = box_min[3 * l + 1]; q.lminz = box_min[3 * l + 2];
        q.lmaxx = box_max[3 * l]; q.lmaxy = box_max[3 * l + 1]; q.lmaxz = box_max[3 * l + 2];
        q.rminx = box_min[3 * r]; q.rminy = box_min[3 * r + 1]; q.rminz = box_min[3 * r + 2];
        q.rmaxx = box_max[3 * r]; q.rmaxy = box_max[3 * r + 1]; q.rmaxz = box_max[3 * r + 2];
    }
}
__global__ void k_update_roots(uint32_t n_objects, const int2* __restrict__ obj_range, const float* __restrict__ box_min, const float* __restrict__ box_max,
                               const DevNode* __restrict__ nodes, DevNode* __restrict__ root_nodes) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_objects) return;
    const int32_t r = obj_range[k].x;
    DevNode n = nodes[r];                  // (skip / leaf: static; the box from the array, whatever k_update_nodes has written so far)
    n.minx = box_min[3 * (size_t)r]; n.miny = box_min[3 * (size_t)r + 1]; n.minz = box_min[3 * (size_t)r + 2];
    n.maxx = box_max[3 * (size_t)r]; n.maxy = box_max[3 * (size_t)r + 1]; n.maxz = box_max[3 * (size_t)r + 2];
    root_nodes[k] = n;
}

// =================================================================================================
// The chip's VALU issue rate, measured (srt_debug_valu_rate): every wave runs `iters` x 64 v_fma_f32 over 16 independent accumulators
// between two pairs of stamps (s_memtime = shader cycles, s_memrealtime = the constant 100 MHz counter all CUs share) and notes the
// SIMD it ran on.  With several such waves per SIMD the SIMDs issue back to back, so a SIMD's instructions over the cycles between its
// first wave's start and its last wave's end is what ONE SIMD issues per cycle -- the yardstick bench.py's roofline prices VALU work
// against (MI355X_MICROARCH.md: SIMD-32, a wave64 VALU instruction over 2 cycles -> 0.5; measured: 0.45).
// =================================================================================================
// PACKED: the same count of v_pk_fma_f32 (two f32 fused multiply-adds per lane and instruction): do they issue at the rate of v_fma_f32?
template <bool PACKED>
__global__ __launch_bounds__(256) void k_valu_rate(uint32_t iters, float* __restrict__ sink, unsigned long long* __restrict__ stamps) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    float a[16];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = (float)(threadIdx.x + k) * 1.0e-3f;
    const float b = 0.99999f, c = 1.0e-7f;
    v2f pa[16]; const v2f pb = { b, b }, pc = { c, c };
#pragma unroll
    for (int k = 0; k < 16; k++) pa[k] = v2f{ a[k], a[k] + 1.0f };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < 16; k++) {
                if (PACKED) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pa[k]) : "v"(pb), "v"(pc));
                else        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
            }
    }
    if (PACKED) {
#pragma unroll
        for (int k = 0; k < 16; k++) a[k] = pa[k].x + pa[k].y;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) sum += a[k];
    sink[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) {      // + where the wave ran: HW_ID (wave / SIMD / CU / SH / SE) and XCC_ID
        unsigned long long* o = stamps + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
        o[0] = t0; o[1] = t1; o[2] = r0; o[3] = r1;
        o[4] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32);
    }
}

// =================================================================================================
// Known-answer kernels: the device leaf functions on caller-supplied vectors (srt_kat_* in the ABI),
// so that the reference's KAT fixtures pin the DEVICE code paths directly.
// =================================================================================================
__global__ void k_kat_ray_aabb(uint32_t n, const float* __restrict__ ray_od, const float* __restrict__ box,
                               uint8_t* __restrict__ exact, uint8_t* __restrict__ branchless,
                               uint8_t* __restrict__ filtered, uint8_t* __restrict__ ambiguous) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = ray_od + 6 * (size_t)i; const float* b = box + 6 * (size_t)i;
    const V3 o = mk(r[0], r[1], r[2]), d = mk(r[3], r[4], r[5]);
    exact[i] = ray_aabb(o, d, b[0], b[1], b[2], b[3], b[4], b[5]) ? 1 : 0;
    branchless[i] = ray_aabb_nb(o, d, b[0], b[1], b[2], b[3], b[4], b[5]) ? 1 : 0;
    bool amb;
    filtered[i] = ray_aabb_filtered(o, ray_rcp(d), b[0], b[1], b[2], b[3], b[4], b[5], amb) ? 1 : 0;
    ambiguous[i] = amb ? 1 : 0;
}
__global__ void k_kat_ray_triangle(uint32_t n, const float* __restrict__ ray_od, const DevTri* __restrict__ tris, float* __restrict__ t) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* r = ray_od + 6 * (size_t)i; const DevTri q = tris[i];
    t[i] = ray_triangle(mk(r[0], r[1], r[2]), mk(r[3], r[4], r[5]), mk(q.p1x, q.p1y, q.p1z), mk(q.e1x, q.e1y, q.e1z), mk(q.e2x, q.e2y, q.e2z));
}
// in: ray_od(6) normal-from-DevTri(3 via tris) light(3) color(3) props(3) t(1)
__global__ void k_kat_phong(uint32_t n, const float* __restrict__ in, const DevTri* __restrict__ tris, float* __restrict__ rgb) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* q = in + 28 * (size_t)i; const DevTri tr = tris[i];
    const V3 c = phong(mk(tr.nx, tr.ny, tr.nz), mk(q[0], q[1], q[2]), mk(q[3], q[4], q[5]), mk(q[18], q[19], q[20]),
                       mk(q[21], q[22], q[23]), q[24], q[25], q[26], q[27]);
    rgb[i * 3] = c.x; rgb[i * 3 + 1] = c.y; rgb[i * 3 + 2] = c.z;
}
__global__ void k_kat_interp_normal(uint32_t n, const float* __restrict__ in12, float* __restrict__ out3) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* q = in12 + 12 * (size_t)i;
    const V3 r = normalize3(mk((q[9] * q[0] + q[10] * q[3]) + q[11] * q[6], (q[9] * q[1] + q[10] * q[4]) + q[11] * q[7],
                               (q[9] * q[2] + q[10] * q[5]) + q[11] * q[8]));
    out3[i * 3] = r.x; out3[i * 3 + 1] = r.y; out3[i * 3 + 2] = r.z;
}
// pow_like_host against the f64 library pow on the same inputs: out[0] = fast path result, out[1] = (float)pow(double)
__global__ void k_kat_pow(uint32_t n, const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ fast, float* __restrict__ lib) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fast[i] = pow_like_host(x[i], y[i]);
    lib[i] = (float)pow((double)x[i], (double)y[i]);
}
__global__ void k_kat_tonemap(uint32_t n, const float* __restrict__ lin, float reinhard, float gamma, float* __restrict__ tone, int32_t* __restrict__ q) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 3 * n) return;
    const float c = tone1(lin[i], reinhard, gamma);
    tone[i] = c; q[i] = quant1(c);
}
