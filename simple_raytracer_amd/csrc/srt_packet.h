// srt_packet.h -- wave-coherent "packet" traversal kernels (gfx950).  Included by srt_hip.hip after srt_kernels.h.
//
// The reference's traversal (boundingBoxIntersection, simple_raytracer.cpp:296-317) tests, for one ray, every node whose
// ancestors all pass the slab test.  On the pre-order node array that is the stackless walk  i = pass ? i + 1 : skip[i].
// Here the 64 rays of a wavefront run that walk IN LOCK STEP: every lane keeps the index `n` of the next node its own walk
// would visit, the wave visits i = min n, and the lanes with n == i test node i.  A lane's n only ever grows, so every
// (node, ray) pair of the reference's candidate set is tested exactly once and in the reference's own visit order per ray
// -- closest hit needs no merge (a lane meets its triangles in increasing id order: strict '<' keeps the first, :429) and
// the any-hit walk of a shadow ray stops exactly where the sequential walk stops, so the work counts ARE the algorithmic
// counts the CPU oracle mirrors.
//
// What the lock step buys: the node index is wave-uniform, so node and triangle records come through the SCALAR cache
// (s_load_dwordx8 / x4: once per wave instead of once per lane, no VGPRs, no LDS queues, no ballot-prefix pushes) and
// the slab / Moller-Trumbore tests take their box / triangle operands from SGPRs.  It pays when the rays of a wave
// visit mostly the same nodes: the light samples of one pixel (same origin, directions a 3-unit staircase apart on a
// ~600-unit lever, softShadow:363-383), and primary rays through a hierarchy of heavily overlapping boxes (the 1 M
// triangle soup: ~2,500 slab tests per ray, nearly all shared by the 8x8 pixels of a tile).  Where rays of a tile part
// ways deep in a good hierarchy (bunny at 1080p) the node-queue kernels of srt_kernels.h keep lanes fuller.
#pragma once
#include "srt_kernels.h"

// wave-uniform minimum of a per-lane index (rare path of the walk: every lane that could have continued at the expected
// successor dropped out -- shadowed, or jumping over its own object)
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64); v = o < v ? o : v; }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

// Slab test of one node for the active lanes: filtered form first, the reference's comparisons with exact divides where the
// filter cannot decide (srt_device.h).  Box operands are wave-uniform.
template <bool FILTER>
__device__ __forceinline__ bool packet_slab(V3 o, V3 d, RayRcp rc, const DevNode& nd) {
    if (FILTER) {
        bool amb;
        bool pass = ray_aabb_filtered(o, rc, nd.minx, nd.miny, nd.minz, nd.maxx, nd.maxy, nd.maxz, amb);
        if (amb) pass = ray_aabb_nb(o, d, nd.minx, nd.miny, nd.minz, nd.maxx, nd.maxy, nd.maxz);
        return pass;
    }
    return ray_aabb_nb(o, d, nd.minx, nd.miny, nd.minz, nd.maxx, nd.maxy, nd.maxz);
}

// ---- any-hit packet walk (shadowIntersection:321-342) -----------------------------------------------------------------
// valid lanes carry a shadow ray (origin ro = d*t, direction rd = L - d*t, :325-326) and the node range `self` of the hit
// object, which the walk jumps over (the reference walks it and discards the result, :328/:331).  Returns "shadowed".
//
// PF > 0: the successors' records requested AHEAD.  A step of the plain form is a chain  record load -> slab test -> ballot -> next
// record load (-> triangle load -> test -> next triangle load ...): ~860 cycles, most of it the two scalar round trips, and a walk
// through a tree crown is 300-500 such steps in a row -- the launch ends when the last of them does (counting build: wave slots 71 %
// busy on K4, 44 % on an eighth of it).  The walk only ever continues at i + 1 (some lane passed, or a leaf), at skip[i] (no lane
// passed) or, rarely, further on (every lane at the successor dropped out); so the record of i + 1 (PF >= 1), the record of skip[i] of
// an inner node and the first triangle of a leaf (PF >= 2) are requested at the TOP of the step, while the slab test of node i runs,
// and the next triangle of a leaf while the current one is tested.  Same visits, same tests, same order: only loads that may turn out
// unused (clamped into the arrays).
template <bool COUNT, bool FILTER, int PF = 0, bool STEPS = false>
__device__ __forceinline__ bool packet_any_hit(const DevScene& s, bool valid, V3 ro, V3 rd, int2 self, uint32_t& steps,
                                               unsigned long long& n_node, unsigned long long& n_tri, unsigned long long* diag = nullptr) {
    steps = 0u;                                              // node steps of this walk (wave-uniform): what makes a quadrant a heavy one
    const uint32_t N = s.n_nodes;
    const RayRcp rc = ray_rcp(rd);
    uint32_t n = valid ? 0u : N;
    if (n == (uint32_t)self.x) n = (uint32_t)self.y;
    bool flag = false;
    uint32_t i = wave_min_u32(n);
    if (PF > 0) {
        // records as float4 pairs / triples (plain vector types: the compiler keeps them in SGPRs; copies of the structs did not)
        const float4* const n4 = reinterpret_cast<const float4*>(s.nodes);
        const float4* const t4 = reinterpret_cast<const float4*>(s.tris);
        const uint32_t last = N - 1u;
        const uint32_t i0 = i < N ? i : last;
        float4 na = n4[2 * (size_t)i0], nb = n4[2 * (size_t)i0 + 1];
        // scalar loads return out of order, so the only wait there is is "all of them": every way into a step must arrive with nothing
        // outstanding, or the compiler has to put that wait right behind the requests at the top of the step (it did) and nothing overlaps
        // (an empty asm the record passes THROUGH makes the compiler wait there; the s_waitcnt builtin and a volatile asm count as memory
        // clobbers and turn every record load of the walk into a vector load)
#define SRT_ARRIVED(a, b) asm("" : "+s"((a).x), "+s"((a).y), "+s"((a).z), "+s"((a).w), "+s"((b).x), "+s"((b).y), "+s"((b).z), "+s"((b).w))
        SRT_ARRIVED(na, nb);
        while (i < N) {
            if (STEPS) steps++;
            const uint32_t i1 = i < last ? i + 1u : last;
            const int32_t skip = __float_as_int(nb.z), leaf = __float_as_int(nb.w);
            const bool is_leaf = leaf >= 0;
            const uint32_t first = (uint32_t)leaf >> LEAF_SHIFT, cnt = is_leaf ? (uint32_t)leaf & LEAF_MAX : 0u;
            float4 ya = make_float4(0.f, 0.f, 0.f, 0.f), yb = ya;            // inner node: record of skip[i]; leaf: first triangle (ya, yb, yc)
            float yc = 0.f;
            if (PF >= 2) {
                if (is_leaf) { if (cnt) { ya = t4[3 * (size_t)first]; yb = t4[3 * (size_t)first + 1]; yc = t4[3 * (size_t)first + 2].x; } }
                else { const uint32_t sk = (uint32_t)skip < N ? (uint32_t)skip : last; ya = n4[2 * (size_t)sk]; yb = n4[2 * (size_t)sk + 1]; }
            }
            const float4 xa = n4[2 * (size_t)i1], xb = n4[2 * (size_t)i1 + 1];           // record of i + 1
            const bool act = n == i;
            bool pass = false;
            if (act) {
                if (FILTER) {
                    bool amb;
                    pass = ray_aabb_filtered(ro, rc, na.x, na.y, na.z, na.w, nb.x, nb.y, amb);
                    if (amb) pass = ray_aabb_nb(ro, rd, na.x, na.y, na.z, na.w, nb.x, nb.y);
                } else pass = ray_aabb_nb(ro, rd, na.x, na.y, na.z, na.w, nb.x, nb.y);
            }
            uint32_t cand;
            if (is_leaf) {
                bool todo = pass;
                if (PF < 2 && cnt && __ballot(todo)) { ya = t4[3 * (size_t)first]; yb = t4[3 * (size_t)first + 1]; yc = t4[3 * (size_t)first + 2].x; }
                for (uint32_t k = 0; k < cnt && __ballot(todo); k++) {
                    float4 za = ya, zb = yb; float zc = yc;
                    if (k + 1u < cnt) { const size_t j = 3 * (size_t)(first + k + 1u); za = t4[j]; zb = t4[j + 1]; zc = t4[j + 2].x; }      // in flight during this triangle's test
                    if (todo) {
                        const float t = ray_triangle(ro, rd, mk(ya.x, ya.y, ya.z), mk(ya.w, yb.x, yb.y), mk(yb.z, yb.w, yc));
                        if (t != SRT_NEG_INF) { flag = true; todo = false; }      // any t >= 0, NaN included (:335)
                    }
                    ya = za; yb = zb; yc = zc;
                }
                if (act) n = flag ? N : i + 1u;
                cand = i + 1u;
            } else {
                if (act) n = pass ? i + 1u : (uint32_t)skip;
                cand = __ballot(pass) ? i + 1u : (uint32_t)skip;
            }
            if (n == (uint32_t)self.x) n = (uint32_t)self.y;        // never the hit object's own tree
            const uint32_t inext = __ballot(n == cand) ? cand : wave_min_u32(n);
            if (inext == i + 1u) { na = xa; nb = xb; }
            else if (PF >= 2 && !is_leaf && inext == (uint32_t)skip) { na = ya; nb = yb; }
            else if (inext < N) { na = n4[2 * (size_t)inext]; nb = n4[2 * (size_t)inext + 1]; SRT_ARRIVED(na, nb); }
            i = inext;
        }
#undef SRT_ARRIVED
        return flag;
    }
    while (i < N) {
        const DevNode nd = s.nodes[i];                          // wave-uniform index: one request per wave
        if (STEPS) steps++;
        if (COUNT && diag) diag[0]++;
        const bool act = n == i;
        bool pass = false;
        if (act) {
            if (COUNT) n_node++;
            pass = packet_slab<FILTER>(ro, rd, rc, nd);
        }
        uint32_t cand;
        if (nd.leaf >= 0) {
            const uint32_t first = (uint32_t)nd.leaf >> LEAF_SHIFT, cnt = (uint32_t)nd.leaf & LEAF_MAX;
            bool todo = pass;
            for (uint32_t k = 0; k < cnt && __ballot(todo); k++) {
                if (COUNT && diag) diag[2]++;
                const DevTri tr = s.tris[first + k];            // wave-uniform
                if (todo) {
                    if (COUNT) n_tri++;
                    const float t = ray_triangle(ro, rd, mk(tr.p1x, tr.p1y, tr.p1z), mk(tr.e1x, tr.e1y, tr.e1z), mk(tr.e2x, tr.e2y, tr.e2z));
                    if (t != SRT_NEG_INF) { flag = true; todo = false; }      // any t >= 0, NaN included (:335)
                }
            }
            if (act) n = flag ? N : i + 1u;
            cand = i + 1u;
        } else {
            if (act) n = pass ? i + 1u : (uint32_t)nd.skip;
            cand = __ballot(pass) ? i + 1u : (uint32_t)nd.skip;
        }
        if (n == (uint32_t)self.x) n = (uint32_t)self.y;        // never the hit object's own tree
        i = __ballot(n == cand) ? cand : wave_min_u32(n);       // no lane is behind cand (see header): usually one is AT it
    }
    return flag;
}

// ---- any-hit packet walk through LDS windows ------------------------------------------------------------------------------------
// Same walk, same results and counts as packet_any_hit, but the records come from two per-wave LDS windows instead of one scalar
// load per step: a step of the scalar form costs a dependent scalar-cache / L2 round trip (measured on the K3 scene with 16 light
// samples: ~860 cycles per step, 69 % of the wave-cycles waiting), and the walk only ever moves FORWARD through the pre-order node
// array (i + 1 or skip[i] > i) and, with it, through the triangle array (leaves lie in visit order).  So the wave loads the next
// 64 nodes (2 KB, one coalesced 2 x dwordx4 per lane) when the walk leaves its window and reads node i from LDS with a
// wave-uniform address (a broadcast); likewise the next 64 triangles (3 KB) when a leaf reaches past the window.  Deep in a tree
// most steps stay inside the window (a subtree of depth 6 is 63 consecutive records).
struct PkWindows {
    float4 na[64], nb[64];               // nodes base .. base + 63: min.xyz max.x | max.yz skip leaf
    float4 t0[64], t1[64], t2[64];       // triangles base .. base + 63 (DevTri)
};
template <bool COUNT, bool FILTER>
__device__ __forceinline__ bool packet_any_hit_win(const DevScene& s, bool valid, V3 ro, V3 rd, int2 self, uint32_t& steps, PkWindows& w,
                                                   uint32_t& nbase, uint32_t& tbase, unsigned long long& n_node, unsigned long long& n_tri,
                                                   unsigned long long* diag = nullptr) {      // counting build: [0] steps [1] node-window loads [2] triangle iterations [3] triangle-window loads
    const uint32_t N = s.n_nodes, NT = s.n_tris;
    steps = 0u;
    const uint32_t lane = threadIdx.x & 63;
    const float4* nodes4 = reinterpret_cast<const float4*>(s.nodes);
    const float4* tris4 = reinterpret_cast<const float4*>(s.tris);
    const RayRcp rc = ray_rcp(rd);
    uint32_t n = valid ? 0u : N;
    if (n == (uint32_t)self.x) n = (uint32_t)self.y;
    bool flag = false;
    uint32_t i = wave_min_u32(n);
    while (i < N) {
        if (i - nbase >= 64u) {                                  // the walk left the node window (which outlives the walk): load nodes i .. i + 63
            nbase = i;
            const uint32_t j = nbase + lane;
            if (j < N) { w.na[lane] = nodes4[2 * (size_t)j]; w.nb[lane] = nodes4[2 * (size_t)j + 1]; }
            __builtin_amdgcn_wave_barrier();
            if (COUNT && diag) diag[1]++;
        }
        const float4 a = w.na[i - nbase], b = w.nb[i - nbase];   // wave-uniform address: broadcast
        steps++;
        if (COUNT && diag) diag[0]++;
        const int32_t skip = __builtin_amdgcn_readfirstlane(__float_as_int(b.z)), leaf = __builtin_amdgcn_readfirstlane(__float_as_int(b.w));
        const bool act = n == i;
        bool pass = false;
        if (act) {
            if (COUNT) n_node++;
            if (FILTER) {
                bool amb;
                pass = ray_aabb_filtered(ro, rc, a.x, a.y, a.z, a.w, b.x, b.y, amb);
                if (amb) pass = ray_aabb_nb(ro, rd, a.x, a.y, a.z, a.w, b.x, b.y);
            } else pass = ray_aabb_nb(ro, rd, a.x, a.y, a.z, a.w, b.x, b.y);
        }
        uint32_t cand;
        if (leaf >= 0) {
            const uint32_t first = (uint32_t)leaf >> LEAF_SHIFT, cnt = (uint32_t)leaf & LEAF_MAX;
            bool todo = pass;
            if (cnt && __ballot(todo)) {
                if (first - tbase > 64u - cnt) {                 // the leaf reaches past the triangle window (cnt <= 31): load first .. first + 63
                    tbase = first;
                    const uint32_t j = tbase + lane;
                    if (j < NT) { w.t0[lane] = tris4[3 * (size_t)j]; w.t1[lane] = tris4[3 * (size_t)j + 1]; w.t2[lane] = tris4[3 * (size_t)j + 2]; }
                    __builtin_amdgcn_wave_barrier();
                    if (COUNT && diag) diag[3]++;
                }
                const uint32_t tw = first - tbase;
                for (uint32_t k = 0; k < cnt && __ballot(todo); k++) {
                    if (COUNT && diag) diag[2]++;
                    const float4 q0 = w.t0[tw + k], q1 = w.t1[tw + k];
                    const float e2z = w.t2[tw + k].x;
                    if (todo) {
                        if (COUNT) n_tri++;
                        const float t = ray_triangle(ro, rd, mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(q1.z, q1.w, e2z));
                        if (t != SRT_NEG_INF) { flag = true; todo = false; }      // any t >= 0, NaN included (:335)
                    }
                }
            }
            if (act) n = flag ? N : i + 1u;
            cand = i + 1u;
        } else {
            if (act) n = pass ? i + 1u : (uint32_t)skip;
            cand = __ballot(pass) ? i + 1u : (uint32_t)skip;
        }
        if (n == (uint32_t)self.x) n = (uint32_t)self.y;        // never the hit object's own tree
        i = __ballot(n == cand) ? cand : wave_min_u32(n);
    }
    return flag;
}

// ---- closest-hit packet walk for rays from the origin (rayIntersection:405-431) ------------------------------------------
// WORLD: camera mode -- the rays leave `o` instead of the origin, so the test reads the general triangle record (P1, e1, e2)
template <bool COUNT, bool FILTER, bool WORLD>
__device__ __forceinline__ void packet_closest_hit(const DevScene& s, bool valid, V3 o_in, V3 d, float& best, int32_t& best_id,
                                                   unsigned long long& n_node, unsigned long long& n_tri) {
    const uint32_t N = s.n_nodes;
    const V3 o = WORLD ? o_in : mk(0.f, 0.f, 0.f);
    const RayRcp rc = ray_rcp(d);
    uint32_t n = valid ? 0u : N;
    best = __builtin_inff(); best_id = -1;
    uint32_t i = valid ? 0u : N;
    i = wave_min_u32(i);
    while (i < N) {
        const DevNode nd = s.nodes[i];
        const bool act = n == i;
        bool pass = false;
        if (act) {
            if (COUNT) n_node++;
            pass = packet_slab<FILTER>(o, d, rc, nd);
        }
        uint32_t cand;
        if (nd.leaf >= 0) {
            const uint32_t first = (uint32_t)nd.leaf >> LEAF_SHIFT, cnt = (uint32_t)nd.leaf & LEAF_MAX;
            if (__ballot(pass)) {
                for (uint32_t k = 0; k < cnt; k++) {
                    float t;
                    if (WORLD) {
                        const DevTri tr = s.tris[first + k];    // wave-uniform
                        t = pass ? ray_triangle(o, d, mk(tr.p1x, tr.p1y, tr.p1z), mk(tr.e1x, tr.e1y, tr.e1z), mk(tr.e2x, tr.e2y, tr.e2z)) : SRT_NEG_INF;
                    } else {
                        const DevTriO tr = s.tris_o[first + k];     // wave-uniform
                        t = pass ? ray_triangle_origin(d, mk(tr.tx, tr.ty, tr.tz), mk(tr.e1x, tr.e1y, tr.e1z), mk(tr.e2x, tr.e2y, tr.e2z), mk(tr.qx, tr.qy, tr.qz))
                                 : SRT_NEG_INF;
                    }
                    if (pass) {
                        if (COUNT) n_tri++;
                        if (t != SRT_NEG_INF && t < best) { best = t; best_id = (int32_t)(first + k); }      // strict '<', ids rise (:429)
                    }
                }
            }
            if (act) n = i + 1u;
            cand = i + 1u;
        } else {
            if (act) n = pass ? i + 1u : (uint32_t)nd.skip;
            cand = __ballot(pass) ? i + 1u : (uint32_t)nd.skip;
        }
        i = __ballot(n == cand) ? cand : wave_min_u32(n);
    }
}

// Pixel of lane l in an 8x8 tile, "quadrant layout": bits 5:4 = 4x4 quadrant, 3:2 = row, 1:0 = column inside it.  A ballot
// over the lanes of a tile IS a word of the tile's shadow-bit / hit-mask format (srt_kernels.h).
__device__ __forceinline__ void tile_lane_pixel(uint32_t bx, uint32_t by, uint32_t lane, uint32_t& px, uint32_t& r) {
    const uint32_t q = lane >> 4, ql = lane & 15u;
    px = bx * 8u + (q & 1u) * 4u + (ql & 3u);
    r = by * 8u + (q >> 1) * 4u + (ql >> 2);
}

// workgroup (4 waves) -> 8x8 tiles.  XCD_ROWS: whole tile rows per XCD (workgroups are dealt round-robin over the 8 XCDs),
// for scenes far bigger than one XCD's L2; the host pads gridDim.y to a multiple of 8.
template <bool XCD_ROWS>
__device__ __forceinline__ bool packet_tile_of_wave(const DevParams& p, uint32_t& bx, uint32_t& by) {
    const uint32_t gx = gridDim.x;                              // workgroups per row: each covers 2 x 2 tiles
    uint32_t wx = blockIdx.x, wy = blockIdx.y;
    if (XCD_ROWS) {
        const uint32_t w = blockIdx.y * gx + blockIdx.x, idx = w >> 3;
        wy = (idx / gx) * 8u + (w & 7u); wx = idx % gx;
    }
    const uint32_t wave = threadIdx.x >> 6;
    bx = wx * 2u + (wave & 1u); by = wy * 2u + (wave >> 1);
    return bx * 8u < p.W && by * 8u < p.rows;
}

// =================================================================================================
// Closest hit, packet form: one wavefront per 8x8 pixel tile.  Writes hit ids, t, the final pixel of a miss, the hit-pixel
// statistic and (qlist != null) the tile's quadrants with hits into the quadrant list.
// =================================================================================================
template <bool COUNT, bool FILTER, bool XCD_ROWS, bool WORLD = false>
__global__ __launch_bounds__(256) void k_closest_hit_pk(DevScene s, DevParams p, int32_t* __restrict__ hit_id, float* __restrict__ t_out,
                                                        float* __restrict__ rgb_linear, uint8_t* __restrict__ rgb8,
                                                        uint32_t* __restrict__ qcount, uint32_t* __restrict__ qlist, uint32_t qcap,
                                                        unsigned long long* __restrict__ counters) {
    uint32_t bx, by;
    if (!packet_tile_of_wave<XCD_ROWS>(p, bx, by)) return;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t px, r;
    tile_lane_pixel(bx, by, lane, px, r);
    const bool live = pixel_live(p, px, r);
    const V3 d = live ? primary_dir(p, px, image_row(p, r)) : mk(0.f, 0.f, p.focal);
    unsigned long long n_node = 0, n_tri = 0;
    float best; int32_t id;
    packet_closest_hit<COUNT, FILTER, WORLD>(s, live, ray_origin(p), d, best, id, n_node, n_tri);
    const uint32_t tiles_x = (p.W + 7u) / 8u;
    if (live) {
        const size_t pix = (size_t)r * p.W + px;
        hit_id[pix] = id;
        t_out[pix] = best;
        if (id < 0) {      // a miss is final here: zero light sum, background pixel (:518, drawImage:476-487)
            if (rgb_linear) { rgb_linear[pix * 3] = 0.0f; rgb_linear[pix * 3 + 1] = 0.0f; rgb_linear[pix * 3 + 2] = 0.0f; }
            if (rgb8) { rgb8[pix * 3] = (uint8_t)(p.bg & 255); rgb8[pix * 3 + 1] = (uint8_t)((p.bg >> 8) & 255); rgb8[pix * 3 + 2] = (uint8_t)((p.bg >> 16) & 255); }
        }
    }
    const unsigned long long hm = __ballot(live && id >= 0);
    if (lane == 0 && hm) {
        const uint32_t tile_index = by * tiles_x + bx;
        if (p.exp & 32u) atomicAdd(counters + CTR_HIT_BASE + 8 * (tile_index & (HIT_SHARDS - 1)), (unsigned long long)__popcll(hm));      // (the shading kernel takes the statistic otherwise)
        if (qlist)
            for (uint32_t q = 0; q < 4; q++) if ((hm >> (16 * q)) & 0xffffull) quadrant_list_append(qcount, qlist, qcap, tile_index, q, (uint32_t)(hm >> (16 * q)) & 0xffffu, p.heavy_steps);
    }
    if (COUNT) { wave_add(counters + 1, n_node); wave_add(counters + 2, n_tri); }
}

// =================================================================================================
// Shadow rays, packet form, fed from the list of 4x4-pixel quadrants that contain hits.
//
// A 3840x2160 frame of the reference's scenes is mostly sky: 129,600 tiles of which a tenth contain a hit.  A grid over all
// tiles (x light-sample chunks) is bound by workgroup dispatch, not by work (measured: 2.07 M waves launched, 23 % of the wave
// slots occupied), and a tile in a tree crown is a long workgroup.  So the closest-hit kernels append every quadrant with a
// hit to a list -- 64 shard lists with their own counters, one atomic per wave with hits -- and this kernel is a fixed-size
// grid whose WAVES pull work units from it, every wave on its own (no workgroup-level synchronisation: a quadrant in a tree
// crown is 16 long walks, and they spread over 16 waves anywhere on the chip).
//
// A unit is ONE packet walk: (quadrant, group of 8 hit pixels, 8 consecutive light samples) = at most 64 shadow rays that stay
// close together all the way: 8 samples move the light by at most 9 units per axis on a ~600-unit lever, 8 neighbouring pixels
// start within a unit of each other.  Measured alternatives (K4 shape): the 64 samples of ONE pixel fan out over
// ~5 units at the far side of the scene and part ways in the tree crowns; 16 pixels x 4 samples start on too many different
// leaves.  The counting build reports the shape of the walks: 7 steps on average (6 of them the roots of the other objects),
// 49 of 64 lanes active per step.
// Units are dealt dynamically: unit u belongs to fetch shard u % 64; a wave takes the next unit of its home shard with one
// atomic (issued one unit ahead, so its round trip hides behind the walk) and moves on to shards that still have work when
// its own is empty -- with static striding the launch lasted as long as the wave that happened to own a tree crown.
// Result, pixel-major: per hit pixel one u64 per 64 light samples, bit = sample; a walk owns one BYTE of a pixel's word (the rays
// of a pixel are 8 consecutive lanes: a byte of the ballot), so there are no atomics on the result.
// =================================================================================================
// qcount: [0, 64) entries per shard list (filled by the closest-hit kernel), [64, 128) units handed out per fetch shard; counters
// QL_STRIDE words apart.  Both are zeroed by the shading kernel that follows.
// HEAVY: the kernel of a batch call -- two lists per shard (heavy quadrants first, srt_kernels.h) and the cost map for the next call.  The
// single-frame kernel is built without (its closest-hit launch gets heavy_steps = 0 and fills the ordinary lists only): the second
// prefix sum, the step counter and two more spilled registers cost it 2-6 % (same box: K4 9.7-10.0 -> 10.05-10.1 ms per 8 frames, K3
// with 16 samples 3.33 -> 3.51-3.61) and its frames pipeline on their streams anyway.
template <bool COUNT, bool FILTER, bool WINDOWS, bool ENTRY_MAJOR, int PF = 0, bool HEAVY = false>
__device__ __forceinline__ void shadow_pk_body(const DevScene& s, const DevParams& p, const int32_t* __restrict__ hit_id, const float* __restrict__ t_in,
                                               uint32_t* __restrict__ qcount, const uint32_t* __restrict__ qlist, uint32_t qcap,
                                               unsigned long long* __restrict__ shadow_px, unsigned long long* __restrict__ counters) {
    __shared__ PkWindows win_all[WINDOWS ? 4 : 1];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t nbase = 0x80000000u, tbase = 0x80000000u;                   // window bases (sentinel: nothing loaded); the windows outlive a walk
    const uint32_t tiles_x = (p.W + 7u) / 8u;
    const uint32_t n_lch = (p.n_lights + 63u) >> 6;                       // u64 words per pixel
    const uint32_t n_lc8 = (p.n_lights + 7u) >> 3;                        // chunks of 8 light samples = bytes per pixel
    const uint32_t upe = 2u * n_lc8;                                      // units per entry: two groups of 8 hit ranks x the chunks
    const uint32_t pr = lane >> 3, lg = lane & 7u;                        // this lane's pixel slot and light sample within a walk
    // prefix sums over the shard list lengths (HEAVY: heavy lists and the others): lane k holds shard k
    uint32_t cnt = qcount[lane * QL_STRIDE], cnt_h = HEAVY ? qcount[(QL_SHARDS + lane) * QL_STRIDE] : 0u;
    cnt = cnt < qcap ? cnt : qcap; cnt_h = cnt_h < qcap ? cnt_h : qcap;
    uint32_t incl = cnt, incl_h = cnt_h;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= (uint32_t)off) incl += o;
        if (HEAVY) { const uint32_t oh = (uint32_t)__shfl_up((int)incl_h, off, 64); if (lane >= (uint32_t)off) incl_h += oh; }
    }
    const uint32_t n_rest = (uint32_t)__shfl((int)incl, 63, 64), n_heavy = HEAVY ? (uint32_t)__shfl((int)incl_h, 63, 64) : 0u;
    const uint32_t n_units = (n_rest + n_heavy) * upe, units_h = n_heavy * upe;
    uint32_t* const cost_map = quadrant_cost_map(const_cast<uint32_t*>(qlist), qcap);
    // (odd: unit u is fetched through counter u % 64, and an even spacing would put every heavy unit into a few of the counters)
    const uint32_t every = HEAVY && units_h && n_units / (2u * units_h) > 1u ? (n_units / (2u * units_h)) | 1u : 1u;
    uint32_t* const fetch = qcount + 2 * QL_SHARDS * QL_STRIDE;
    unsigned long long n_node = 0, n_tri = 0;
    unsigned long long diag[5] = { 0, 0, 0, 0, 0 };          // counting build only: steps, node-window loads, triangle iterations, triangle-window loads, walks
    // The grid is sized for the worst case (every tile full of hits); a frame with few units keeps only as many waves as have ~p.pk_units
    // units each to walk (at least one per fetch shard): the others leave before their first atomic.  8,192 waves on the 222 k units of a
    // 1080p frame with 16 samples are 27 units a wave, most of whose life is start-up and contention on the 64 fetch counters -- K3 with
    // 16 samples 3.40 -> 2.9 ms per 8 frames with the grid cut by hand, K4 (61 units a wave) best with the full grid.
    if (!COUNT && p.pk_units) {
        const uint32_t keep = n_units / p.pk_units > (uint32_t)QL_FETCH ? n_units / p.pk_units : (uint32_t)QL_FETCH;
        if (blockIdx.x * 4u + wave >= keep) return;
    }
    uint32_t home = (blockIdx.x * 4u + wave) & (QL_FETCH - 1);
    uint32_t k_next = 0;
    // counting build: when this wave started, and what its longest walk was (10 ns ticks of the constant clock)
    constexpr bool TIMING = COUNT || PF < 0;          // PF < 0: the plain walk of the shipped build with the wave clocks of the counting build
    const unsigned long long t_begin = TIMING ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long w_max = 0, w_max_steps = 0, w_long = 0, w_long_ticks = 0;
    uint32_t w_slow = 0;                                          // walks of 100 us and more
    // units are taken `take` at a time (p.pk_take, >= 1): one atomic per `take` walks on the 64 fetch counters that every wave of the launch hits
    // -- 8,192 waves on 64 addresses are same-address atomics the L2 serialises: K3 with 64 samples 9.19 -> 7.41 ms per 8 frames with four
    // at a time, K4 10.05 -> 9.36, an eighth of K4 1.93 -> 1.69.  A frame alone on the device loses with it where walks are long (K4 on
    // one stream 11.8 -> 12.4: a wave that holds a long walk holds three more units back), so the host asks for four only when other
    // frames are in flight (the hint, batch calls); one at a time near the end of a shard was tried on top and changed nothing for it.
    const uint32_t take = p.pk_take ? p.pk_take : 1u;
    uint32_t kc = 0, ke = 0;                                  // the numbers in hand: [kc, ke) of shard `home`
    if (lane == 0) k_next = atomicAdd(fetch + home * QL_STRIDE, take);
    for (;;) {
        const bool fresh = kc == ke;
        if (fresh) { kc = (uint32_t)__builtin_amdgcn_readfirstlane((int)k_next); ke = kc + take; }
        const uint32_t u = kc * QL_FETCH + home;
        if (u >= n_units) {
            kc = ke = 0;
            // home counter used up: look (agent-scope loads, the counters only grow) for one that still has units, next after home
            // (lane l looks at counter home + 1 + l (+ 64 j, were there more than 64 counters))
            uint32_t found = 0xffffffffu;
#pragma unroll
            for (uint32_t j = 0; j < (uint32_t)QL_FETCH / 64u; j++) {
                const uint32_t c = (home + 1u + lane + 64u * j) & (QL_FETCH - 1);
                const uint32_t seen = __hip_atomic_load(fetch + c * QL_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long open = __ballot((unsigned long long)seen * QL_FETCH + c < (unsigned long long)n_units);
                if (open && found == 0xffffffffu) found = (home + 1u + (uint32_t)__builtin_ctzll(open) + 64u * j) & (QL_FETCH - 1);
            }
            if (found == 0xffffffffu) break;
            home = found;
            if (lane == 0) k_next = atomicAdd(fetch + home * QL_STRIDE, take);
            continue;
        }
        if (fresh && lane == 0) k_next = atomicAdd(fetch + home * QL_STRIDE, take);        // the next numbers: in flight during these walks
        kc++;
        // unit -> (chunk of 8 samples, entry, pixel group): consecutive units are different ENTRIES, so that the waves that
        // start together do not all read the same quadrant
        uint32_t lc, g, e;
        // heavy units are dealt EARLY but not all at once: one in every `every` units until they are used up, by about the middle of
        // the launch (all of them first was measured too: a share of a frame gained as much, but whole frames on four streams LOST 7-15 % --
        // a launch that opens with nothing but long latency-bound walks and closes with nothing but short issue-bound ones uses
        // the machine worse than the mix)
        const uint32_t hslot = HEAVY ? u / every : 0u;
        const bool hv = HEAVY && hslot < units_h && hslot * every == u;      // wave-uniform
        const uint32_t uu = !HEAVY ? u : hv ? hslot : u - (hslot < units_h ? hslot + 1u : units_h), n_entries = hv ? n_heavy : n_rest;
        if (ENTRY_MAJOR) { e = uu / upe; const uint32_t w_ = uu - e * upe; lc = w_ >> 1; g = w_ & 1u; }
        else { lc = uu / (2u * n_entries); const uint32_t ue = uu - lc * 2u * n_entries; g = ue / n_entries; e = ue - g * n_entries; }
        const uint32_t incl_c = hv ? incl_h : incl, cnt_c = hv ? cnt_h : cnt;
        const unsigned long long above = __ballot(incl_c > e);               // first shard list whose inclusive prefix exceeds e
        const uint32_t shard = (uint32_t)__builtin_ctzll(above);
        const uint32_t at = e - (uint32_t)__shfl((int)(incl_c - cnt_c), (int)shard, 64);
        const uint2 ent = reinterpret_cast<const uint2*>(qlist)[(size_t)shard * qcap + (hv ? qcap - 1u - at : at)];      // wave-uniform
        const uint32_t hm = ent.y & 0xffffu, nh = (uint32_t)__popc(hm);
        if (g * 8u >= nh) continue;                                          // at most 8 hit pixels in this quadrant: no second group
        const uint32_t tile_index = ent.x >> 2, q = ent.x & 3u;
        const uint32_t by = tile_index / tiles_x, bx = tile_index - by * tiles_x;
        const uint32_t rank = g * 8u + pr, l = lc * 8u + lg;
        const bool has_px = rank < nh;
        const bool valid = has_px && l < p.n_lights;
        V3 so = mk(0.f, 0.f, 0.f), sd = mk(0.f, 0.f, 1.f);
        int2 self = make_int2(-1, -1);
        uint32_t pix = 0;
        if (has_px) {
            // the rank-th set bit of the hit mask = this lane's pixel inside the quadrant (bit = y * 4 + x)
            uint32_t m = hm;
            for (uint32_t j = 0; j < rank; j++) m &= m - 1u;
            const uint32_t bit = (uint32_t)__builtin_ctz(m);
            const uint32_t px = bx * 8u + (q & 1u) * 4u + (bit & 3u), r = by * 8u + (q >> 1) * 4u + (bit >> 2);
            pix = r * p.W + px;
            const int32_t id = hit_id[pix];
            const float t = t_in[pix];
            self = s.obj_range[s.tri_obj[id]];
            so = primary_dir(p, px, image_row(p, r)) * t;                     // :326
            if (p.cam) so = ray_origin(p) + so;                               // camera mode: the primary ray did not start at 0
        }
        if (valid) sd = mk(p.lights[l * 3], p.lights[l * 3 + 1], p.lights[l * 3 + 2]) - so;      // :325
        if (COUNT) diag[4]++;
        const unsigned long long w0 = COUNT ? __builtin_amdgcn_s_memrealtime() : 0ull, st0 = diag[0];
        uint32_t steps;
        const bool shadowed = WINDOWS ? packet_any_hit_win<COUNT, FILTER>(s, valid, so, sd, self, steps, win_all[wave], nbase, tbase, n_node, n_tri, diag)
                                      : packet_any_hit<COUNT, FILTER, PF, HEAVY>(s, valid, so, sd, self, steps, n_node, n_tri, diag);
        if (HEAVY && p.heavy_steps && steps >= p.heavy_steps && lane == 0) atomicMax(cost_map + ent.x, steps);      // for the next frame's list (srt_kernels.h)
        const unsigned long long sm = __ballot(shadowed);
        if (valid && lg == 0) reinterpret_cast<uint8_t*>(shadow_px)[(size_t)pix * n_lch * 8u + lc] = (uint8_t)(sm >> (pr * 8u));
        if (COUNT) {
            const unsigned long long w = __builtin_amdgcn_s_memrealtime() - w0, st = COUNT ? diag[0] - st0 : 0ull;
            if (w > w_max) w_max = w;
            if (COUNT && st > w_max_steps) w_max_steps = st;
            if (COUNT && st > 256) { w_long++; w_long_ticks += w; }
            w_slow += w >= 10000ull;
        }
    }
    if (TIMING && lane == 0) {      // slots 9..15 and 17..19 are free between the hit shards
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        atomicAdd(counters + 9, t_end - t_begin); atomicMax(counters + 10, t_end); atomicMax(counters + 11, (1ull << 62) - t_begin);
        atomicMax(counters + 12, w_max); atomicMax(counters + 13, w_max_steps); atomicAdd(counters + 14, w_long); atomicAdd(counters + 15, w_long_ticks);
        atomicAdd(counters + 17, 1ull);
        atomicAdd(counters + 18, (unsigned long long)w_slow);
    }
    if (COUNT) {
        wave_add(counters + 3, n_node); wave_add(counters + 4, n_tri);
        // shape of the packet walks (read by srt_sync into the SRT_DIAG_COUNTERS dump): uniform values, lane 0 adds
        if (lane == 0) { atomicAdd(counters + 5, diag[0]); atomicAdd(counters + 6, diag[1]); atomicAdd(counters + 7, diag[2]); atomicAdd(counters + 0, diag[4]); }
    }
}

template <bool COUNT, bool FILTER, bool WINDOWS, bool ENTRY_MAJOR = false, int PF = 0, bool HEAVY = false>
__global__ __launch_bounds__(256, WINDOWS ? 1 : 8) void k_shadow_pk(DevScene s, DevParams p, const int32_t* __restrict__ hit_id, const float* __restrict__ t_in,
                                                   uint32_t* __restrict__ qcount, const uint32_t* __restrict__ qlist, uint32_t qcap,
                                                   unsigned long long* __restrict__ shadow_px, unsigned long long* __restrict__ counters) {
    shadow_pk_body<COUNT, FILTER, WINDOWS, ENTRY_MAJOR, PF, HEAVY>(s, p, hit_id, t_in, qcount, qlist, qcap, shadow_px, counters);
}
// the frames of a batch (srt_render_device_batch): blockIdx.y = frame, gridDim.x waves-of-four pull the units of THAT frame's list.
// (Waves going on to the next frame's list when theirs is empty: measured twice, 25-35 % SLOWER -- 8,192 waves arriving at a drained
// frame's 64 fetch counters are tens of thousands of same-address atomics, which the L2 serialises.)
template <bool FILTER>
__global__ __launch_bounds__(256, 8) void k_shadow_pk_batch(const FrameTab tab) {
    const FrameItem& it = tab.it[blockIdx.y];
    shadow_pk_body<false, FILTER, false, false, 0, true>(it.s, it.p, it.hit_id, it.t_out, it.qcount, it.qlist, it.qcap, it.shadow_bits, it.counters);
}
