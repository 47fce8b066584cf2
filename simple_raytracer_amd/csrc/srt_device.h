// srt_device.h -- device-side leaf math of the ray-trace path, hand-written for gfx950.
//
// Every function reproduces the float expression tree of one reference function exactly (IEEE f32,
// no FMA contraction: the build passes -ffp-contract=off; correctly rounded divide / sqrt are hipcc's
// default), because hit indices must be bit-exact against the reference.
// Reference lines are in /root/reference/simple_raytracer.cpp; GLM expression trees in
// library/glm-master/glm/detail/func_geometric.inl:48-55,73-83,98-105,120-125.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace srt {

// ---- device-resident records -------------------------------------------------------------------
// Node: 32 B = two dwordx4.  Nodes are stored in DFS pre-order (left child of node i is i+1), all
// objects concatenated in object order, so that "every node whose ancestors all pass the slab test"
// (boundingBoxIntersection:296-317) is walked stacklessly:  i = pass ? i+1 : skip[i].
struct __attribute__((aligned(16))) DevNode {
    float minx, miny, minz, maxx;
    float maxy, maxz;
    int32_t skip;      // index of the first node after this node's subtree
    int32_t leaf;      // inner: ~(index of the right child) < 0 (the left child is i+1); leaf: (first_triangle << 5) | count
};
static_assert(sizeof(DevNode) == 32, "node record is 32 B");

// Triangle: 48 B = three dwordx4: ray-independent prefix of rayTriangleIntersection:45-51
// (P1 = p1/w, e1 = P2-P1, e2 = P3-P1: 36 B, what a Moller-Trumbore test reads) followed by the flat
// face normal of calculateTriangleNormal:32-37 (12 B, read once per shaded hit).
struct __attribute__((aligned(16))) DevTri {
    float p1x, p1y, p1z, e1x;
    float e1y, e1z, e2x, e2y;
    float e2z, nx, ny, nz;
};
static_assert(sizeof(DevTri) == 48, "triangle record is 48 B");

// Triangle as a ray FROM THE ORIGIN tests it (every primary ray, sendRaysAndIntersectPointsColors:507): with o = 0,
// tvec = o - P1 (:53) and qvec = cross(tvec, e1) (:58) do not depend on the ray either, so they are evaluated once per
// triangle with the same IEEE operations (the host TU is built with -ffp-contract=off) and the test reads them instead
// of P1: 48 B = three dwordx4, 12 VALU operations fewer per test, same bits.
struct __attribute__((aligned(16))) DevTriO {
    float tx, ty, tz, e1x;
    float e1y, e1z, e2x, e2y;
    float e2z, qx, qy, qz;
};
static_assert(sizeof(DevTriO) == 48, "origin-ray triangle record is 48 B");

// Inner node as the node-queue kernels read it: 64 B = four dwordx4 -- BOTH children's boxes and what each child is.  A queue
// entry then names an inner node that is already KNOWN to pass (its box was tested from its parent's record), one pop loads one
// record and runs the two slab tests of boundingBoxIntersection:305-313 side by side (independent: the second hides the first's
// latencies), and only passing children are queued.  Same boxes, same predicate, every (node, ray) pair still tested exactly once,
// when its parent passes -- but half the pops, LDS pushes and dependent record loads per ray of the 32 B form.
// Records are numbered in pre-order over the INNER nodes only (leaves have none); `linfo` / `rinfo`: a leaf child's
// (first_triangle << 5 | count) >= 0, or ~(record index of an inner child) < 0; `node` / `rnode`: the pre-order node indices of this
// node and of its right child (the left child is node + 1), for the stackless walk that takes over when the queue is full.
struct __attribute__((aligned(16))) DevWide {
    float lminx, lminy, lminz, lmaxx;
    float lmaxy, lmaxz, rminx, rminy;
    float rminz, rmaxx, rmaxy, rmaxz;
    int32_t linfo, rinfo, node, rnode;
};
static_assert(sizeof(DevWide) == 64, "wide inner-node record is 64 B");

constexpr int LEAF_SHIFT = 5;
constexpr int NODE_INDEX_BITS = 26;      // a node-queue entry is (node index << 6 | ray lane) in 32 bits
constexpr int LEAF_MAX = (1 << LEAF_SHIFT) - 1;

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
// glm::dot(vec3): (x + y) + z
__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// glm::cross
__device__ __forceinline__ V3 cross3(V3 x, V3 y) {
    return mk(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
// glm::normalize = v * (1 / sqrt(dot(v, v)))
__device__ __forceinline__ V3 normalize3(V3 v) { float s = 1.0f / sqrtf(dot3(v, v)); return v * s; }
// glm::max(x, y) = (x < y) ? y : x   (NaN-asymmetric on purpose)
__device__ __forceinline__ float glm_max(float x, float y) { return (x < y) ? y : x; }

#define SRT_NEG_INF (-__builtin_inff())

// ---- a4: intersectRayAabbNoOrigin, simple_raytracer.cpp:252-293 ---------------------------------
// Comparison for comparison (x/0 = +-inf, 0/0 = NaN must fall through the same way): no fmin/fmax.
__device__ __forceinline__ bool ray_aabb(V3 o, V3 d, float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
    float minX = (mnx - o.x) / d.x, maxX = (mxx - o.x) / d.x;
    if (minX > maxX) { float s = minX; minX = maxX; maxX = s; }
    float minY = (mny - o.y) / d.y, maxY = (mxy - o.y) / d.y;
    if (minY > maxY) { float s = minY; minY = maxY; maxY = s; }
    if (maxX < minY || maxY < minX) return false;
    if (minY > minX) minX = minY;
    if (maxY < maxX) maxX = maxY;
    float minZ = (mnz - o.z) / d.z, maxZ = (mxz - o.z) / d.z;
    if (minZ > maxZ) { float s = minZ; minZ = maxZ; maxZ = s; }
    if ((minX > maxZ) || (minZ > maxX)) return false;
    return true;
}

// Same predicate without the early return: all six quotients first (so that a node is fetched as two
// dwordx4 and divergent lanes do not serialise on the xy-reject branch).  The z terms are only
// consulted when the xy test did not reject, exactly as the reference's control flow does.
__device__ __forceinline__ bool ray_aabb_nb(V3 o, V3 d, float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
    float minX = (mnx - o.x) / d.x, maxX = (mxx - o.x) / d.x;
    float minY = (mny - o.y) / d.y, maxY = (mxy - o.y) / d.y;
    float minZ = (mnz - o.z) / d.z, maxZ = (mxz - o.z) / d.z;
    if (minX > maxX) { float s = minX; minX = maxX; maxX = s; }
    if (minY > maxY) { float s = minY; minY = maxY; maxY = s; }
    if (minZ > maxZ) { float s = minZ; minZ = maxZ; maxZ = s; }
    const bool rej_xy = (maxX < minY) | (maxY < minX);
    if (minY > minX) minX = minY;
    if (maxY < maxX) maxX = maxY;
    const bool rej_z = (minX > maxZ) | (minZ > maxX);
    return !(rej_xy | rej_z);
}

// Filtered form of the same predicate.  The six quotients are approximated as (box - o) * rcp(d) -- the
// subtraction is the reference's own, so only the division is approximate (<= ~3 ulp against the
// reference's correctly rounded quotient).  For FINITE quotients the reference's comparison chain is the
// classic slab test: after the per-axis swaps, "not (maxX < minY or maxY < minX)" is lo <= hi with
// lo = max(minX, minY), hi = min(maxX, maxY), and "not (lo > maxZ or minZ > hi)" adds lo <= maxZ and
// minZ <= hi, so the box passes iff  tnear = max3(minX, minY, minZ) <= tfar = min3(maxX, maxY, maxZ).
// min / max selection is 1-Lipschitz, so tnear and tfar carry the quotients' relative error and ONE
// comparison with a margin that covers both decides; if it falls inside the margin `ambiguous` is set and
// the caller must evaluate ray_aabb_nb (the reference's comparisons one by one).  Non-finite quotients
// (d = 0, 0 * inf, the FLT_MAX boxes of empty leaves) end up as an infinity or NaN in tnear / tfar -- a
// NaN quotient always has an infinite partner from the same rcp, and v_min / v_max return the other
// operand -- which makes the margin infinite or the difference NaN: ambiguous as well.
// When `ambiguous` is false the returned bool equals the reference's.
struct RayRcp { float x, y, z; };
__device__ __forceinline__ RayRcp ray_rcp(V3 d) {
    RayRcp r; r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y); r.z = __builtin_amdgcn_rcpf(d.z); return r;
}
__device__ __forceinline__ bool ray_aabb_filtered(V3 o, RayRcp rc, float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                                                  bool& ambiguous) {
    const float E = 2.0e-6f, TINY = 1.0e-37f;
    const float x0 = (mnx - o.x) * rc.x, x1 = (mxx - o.x) * rc.x;
    const float y0 = (mny - o.y) * rc.y, y1 = (mxy - o.y) * rc.y;
    const float z0 = (mnz - o.z) * rc.z, z1 = (mxz - o.z) * rc.z;
    const float tnear = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fminf(z0, z1));
    const float tfar = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fmaxf(z0, z1));
    const float m = __builtin_fmaf(E, __builtin_fabsf(tnear) + __builtin_fabsf(tfar), TINY);
    const float df = tfar - tnear;
    ambiguous = !(__builtin_fabsf(df) > m);            // also true when df is NaN or m is inf
    return df > 0.0f;
}

// The predicate as the kernels call it: the filter where it can decide, the reference's comparisons with exact divides where not.
template <bool FILTER>
__device__ __forceinline__ bool slab_pass(V3 o, V3 d, RayRcp rc, float mnx, float mny, float mnz, float mxx, float mxy, float mxz) {
    if (FILTER) {
        bool amb;
        bool pass = ray_aabb_filtered(o, rc, mnx, mny, mnz, mxx, mxy, mxz, amb);
        if (amb) pass = ray_aabb_nb(o, d, mnx, mny, mnz, mxx, mxy, mxz);
        return pass;
    }
    return ray_aabb_nb(o, d, mnx, mny, mnz, mxx, mxy, mxz);
}
// Both children of a DevWide record (r0..r2 = its first three dwordx4) for one ray.  The two filtered tests are independent
// instruction streams; the exact form behind them is ONE block of code that runs once per ambiguous box (rare).
template <bool FILTER>
__device__ __forceinline__ void slab_pass2(V3 o, V3 d, RayRcp rc, float4 r0, float4 r1, float4 r2, bool& passL, bool& passR) {
    if (FILTER) {
        bool ambL, ambR;
        passL = ray_aabb_filtered(o, rc, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, ambL);
        passR = ray_aabb_filtered(o, rc, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w, ambR);
        if (ambL | ambR) {
#pragma nounroll
            for (int c = 0; c < 2; c++) {
                if (c ? ambR : ambL) {
                    const bool e = ray_aabb_nb(o, d, c ? r1.z : r0.x, c ? r1.w : r0.y, c ? r2.x : r0.z, c ? r2.y : r0.w, c ? r2.z : r1.x, c ? r2.w : r1.y);
                    if (c) passR = e; else passL = e;
                }
            }
        }
    } else {
        passL = ray_aabb_nb(o, d, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y);
        passR = ray_aabb_nb(o, d, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w);
    }
}

// ---- a5: rayTriangleIntersection, simple_raytracer.cpp:42-75 (Moller-Trumbore) ------------------
// Returns -inf on a miss, t >= 0 on a hit, NaN when NaN falls through every test (as the reference).
__device__ __forceinline__ float ray_triangle(V3 o, V3 d, V3 p1, V3 e1, V3 e2) {
    V3 pvec = cross3(d, e2);
    float det = dot3(e1, pvec);
    if (__builtin_fabsf(det) < 1e-12f) return SRT_NEG_INF;
    float inv = 1.0f / det;
    V3 tvec = o - p1;
    float u = dot3(tvec, pvec) * inv;
    if (u < 0.0f || u > 1.0f) return SRT_NEG_INF;
    V3 qvec = cross3(tvec, e1);
    float v = dot3(d, qvec) * inv;
    if (v < 0.0f || u + v > 1.0f) return SRT_NEG_INF;
    float t = dot3(e2, qvec) * inv;
    if (t < 0.0f) return SRT_NEG_INF;
    return t;
}

// The same test for a ray from the origin, on the record that carries tvec and qvec (DevTriO)
__device__ __forceinline__ float ray_triangle_origin(V3 d, V3 tvec, V3 e1, V3 e2, V3 qvec) {
    V3 pvec = cross3(d, e2);
    float det = dot3(e1, pvec);
    if (__builtin_fabsf(det) < 1e-12f) return SRT_NEG_INF;
    float inv = 1.0f / det;
    float u = dot3(tvec, pvec) * inv;
    if (u < 0.0f || u > 1.0f) return SRT_NEG_INF;
    float v = dot3(d, qvec) * inv;
    if (v < 0.0f || u + v > 1.0f) return SRT_NEG_INF;
    float t = dot3(e2, qvec) * inv;
    if (t < 0.0f) return SRT_NEG_INF;
    return t;
}

// ---- a8b: calculateBarycentricCoords :79-117 ------------------------------------------------------
__device__ __forceinline__ V3 barycentric(V3 p1, V3 e1, V3 e2, V3 point) {
    V3 v2 = point - p1;
    float d00 = dot3(e1, e1), d01 = dot3(e1, e2), d11 = dot3(e2, e2);
    float d20 = dot3(v2, e1), d21 = dot3(v2, e2);
    float denom = d00 * d11 - d01 * d01;
    float v = (d11 * d20 - d01 * d21) / denom;
    float w = (d00 * d21 - d01 * d20) / denom;
    float u = 1.0f - v - w;
    return mk(u, v, w);
}

// powf as the host libm computes it.  glibc's powf is within a hair of correctly rounded (it differs from the
// correctly rounded result on ~0.08 % of inputs); evaluating in f64 and rounding once lands on the correctly rounded
// float, which is as close to glibc as any device routine can get without replicating its tables.
// The only host<->device op that is not bit-pinned.  Fast path for positive normal x: x = m * 2^e with m in
// [sqrt(1/2), sqrt(2)), log(m) = 2 atanh(s), s = (m-1)/(m+1) (|s| <= 0.172, series to s^19), 2^z by n = rint(z) and a
// degree-12 Taylor polynomial of exp((z-n) ln 2): relative error < 1e-13, i.e. the same float as the f64 library
// pow on all but ~1e-6 of inputs (checked against it in tests), at a quarter of its instruction count.
__device__ __forceinline__ float pow_like_host(float xf, float yf) {
    if (xf > 1.0e-30f && xf < 1.0e30f && __builtin_fabsf(yf) < 1.0e4f) {
        // small integer exponents (the reference's default shininess is 15, Object.cpp:33): square-and-multiply in f64, at most
        // 11 multiplications, relative error < 2e-15 before the one rounding to float -- the correctly rounded x^y on all but
        // ~3e-8 of inputs, at a sixth of the general path's instructions
        if (yf >= 1.0f && yf <= 64.0f && yf == __builtin_truncf(yf)) {
            double r = 1.0, b = (double)xf;
            for (uint32_t e = (uint32_t)yf; e; e >>= 1) { if (e & 1u) r *= b; b *= b; }
            return (float)r;
        }
        const double x = (double)xf, y = (double)yf;
        double m = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1)
        int e = __builtin_amdgcn_frexp_exp(x);
        if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
        const double den = m + 1.0;
        double r = __builtin_amdgcn_rcp(den);                 // ~1e-8; two Newton steps -> ~1e-16
        r = r * (2.0 - den * r);
        r = r * (2.0 - den * r);
        const double s = (m - 1.0) * r, s2 = s * s;
        double p = 1.0 / 19.0;
        p = p * s2 + 1.0 / 17.0; p = p * s2 + 1.0 / 15.0; p = p * s2 + 1.0 / 13.0; p = p * s2 + 1.0 / 11.0;
        p = p * s2 + 1.0 / 9.0;  p = p * s2 + 1.0 / 7.0;  p = p * s2 + 1.0 / 5.0;  p = p * s2 + 1.0 / 3.0;
        p = p * s2 + 1.0;
        const double log2x = (double)e + (2.0 * s * p) * 1.4426950408889634074;
        const double z = y * log2x;
        if (__builtin_fabs(z) < 1000.0) {
            const double n = __builtin_rint(z), f = (z - n) * 0.69314718055994530942;
            double q = 1.0 / 479001600.0;
            q = q * f + 1.0 / 39916800.0; q = q * f + 1.0 / 3628800.0; q = q * f + 1.0 / 362880.0; q = q * f + 1.0 / 40320.0;
            q = q * f + 1.0 / 5040.0;     q = q * f + 1.0 / 720.0;     q = q * f + 1.0 / 120.0;    q = q * f + 1.0 / 24.0;
            q = q * f + 1.0 / 6.0;        q = q * f + 0.5;             q = q * f + 1.0;            q = q * f + 1.0;
            return (float)__builtin_ldexp(q, (int)n);
        }
    }
    return (float)pow((double)xf, (double)yf);                // zeros, denormals, infinities, NaN, negative bases, huge exponents
}

// x^e for a small integer exponent, e in [1, 64]: the square-and-multiply branch of pow_like_host on its own.  For such an exponent it IS
// pow_like_host on every x >= 0 (and NaN, -0): below 1e-30, where pow_like_host calls the library, x^e is x itself (e = 1) or underflows
// to a zero in binary32 either way, and the f64 products cannot overflow before the cast does.  A kernel that only ever shades with
// such exponents (the reference's default shininess is 15, Object.cpp:33) carries none of the general path's registers.
__device__ __forceinline__ float pow_small_int(float xf, uint32_t e) {
    double r = 1.0, b = (double)xf;
    for (; e; e >>= 1) { if (e & 1u) r *= b; b *= b; }
    return (float)r;
}

// ---- a8: phongIllumination :144-200, lightColor = (1,1,1) (:433) ---------------------------------
// INT_SHIN: the caller guarantees an integer shininess in [1, 64] (checked on the host for every object of the scene)
template <bool INT_SHIN = false>
__device__ __forceinline__ V3 phong(V3 n, V3 o, V3 d, V3 L, V3 objColor, float ka, float ks, float shin, float t) {
    const float rView = 1.0f / 3.14159265358979323846264338327950288f;
    const float lc = 1.0f;
    V3 P = o + d * t;
    V3 l = normalize3(L - P);
    float dp = dot3(n, l);
    if (dp < 0.0f) dp = -dp;
    float m = glm_max(dp, 0.0f);
    V3 diffuse = mk(((rView * objColor.x) * lc) * m, ((rView * objColor.y) * lc) * m, ((rView * objColor.z) * lc) * m);
    float ak = rView * ka;
    V3 ambient = mk((ak * objColor.x) * lc, (ak * objColor.y) * lc, (ak * objColor.z) * lc);
    V3 v = normalize3(neg(d));
    V3 I = neg(l);
    float ndi = dot3(n, I);
    V3 r = mk(I.x - (n.x * ndi) * 2.0f, I.y - (n.y * ndi) * 2.0f, I.z - (n.z * ndi) * 2.0f);
    const float sx = glm_max(dot3(r, v), 0.0f);
    float sp = INT_SHIN ? pow_small_int(sx, (uint32_t)shin) : pow_like_host(sx, shin);
    float s1 = ((lc * ks) * m) * sp;
    return mk((diffuse.x + s1) + ambient.x, (diffuse.y + s1) + ambient.y, (diffuse.z + s1) + ambient.z);
}

// ---- a9: Reinhard + gamma (:391-398), quantiser (:447-449) ----------------------------------------
__device__ __forceinline__ float tone1(float c, float reinhard, float gamma) {
    c = c / (c + reinhard);
    return pow_like_host(c, gamma);
}
// int(c*255), truncation.  NaN / out-of-range is UB in the reference; defined as clamp, NaN -> 0
// (identical to the oracle's quant1).
__device__ __forceinline__ int quant1(float c) {
    float s = c * 255.0f;
    if (!(s > 0.0f)) return 0;
    if (s >= 255.0f) return 255;
    return (int)s;
}

} // namespace srt
