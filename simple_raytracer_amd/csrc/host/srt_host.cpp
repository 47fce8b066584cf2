// srt_host.cpp -- host-side mirror of the reference's scene interface (see srt_host.h).
//
// Everything here runs once per frame on the host, as in the reference: transforms (Object.cpp:183-190),
// the median-split hierarchy builder (Object.cpp:205-284), the flattener that writes the flat scene of
// include/srt.h, and the drop-in sendRaysAndIntersectPointsColors that calls the HIP path through the
// C ABI.  Built with -ffp-contract=off: transformed points, boxes and the builder's comparisons must be
// bit-identical to what the reference computes, because leaf order decides tie-breaks.
#include "srt_host.h"

#include <algorithm>
#include <array>
#include <thread>
#include <memory>
#include <sys/stat.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>

#include <zlib.h>

namespace srt_host {

namespace {
// Worker threads for the per-frame host stages (transform, subtree builds, flatten, ImageData).  Created once and
// kept (a stage is a few milliseconds: threads started per call spend most of it being created and migrated to a
// free core); idle workers spin briefly, then sleep.
// Leaked on purpose: workers blocked at process exit die with the process, no static-destructor joins.
class BuildPool {
public:
    static BuildPool& get() { static BuildPool* p = new BuildPool(); return *p; }
    void submit(std::function<void()> f) {
        { std::lock_guard<std::mutex> g(m_); q_.push_back(std::move(f)); }
        queued_.fetch_add(1, std::memory_order_release);
        cv_.notify_one();
    }
    bool run_one() {                       // run one queued task on the calling thread
        if (queued_.load(std::memory_order_acquire) == 0) return false;
        std::function<void()> f;
        {
            std::lock_guard<std::mutex> g(m_);
            if (q_.empty()) return false;
            f = std::move(q_.front()); q_.pop_front();
            queued_.fetch_sub(1, std::memory_order_relaxed);
        }
        f();
        return true;
    }
    unsigned workers() const { return n_; }
private:
    BuildPool() {
        unsigned hc = std::thread::hardware_concurrency();
        n_ = hc > 1 ? (hc > 16 ? 15 : hc - 1) : 0;
        for (unsigned i = 0; i < n_; i++) std::thread([this] { loop(); }).detach();
    }
    void loop() {
        for (;;) {
            bool ran = false;
            for (int spin = 0; spin < 400 && !ran; spin++) { ran = run_one(); if (!ran) std::this_thread::yield(); }
            if (ran) continue;
            std::unique_lock<std::mutex> g(m_);
            cv_.wait(g, [this] { return !q_.empty(); });
        }
    }
    std::mutex m_; std::condition_variable cv_; std::deque<std::function<void()>> q_;
    std::atomic<int> queued_{0};
    unsigned n_ = 0;
};

// One hierarchy build cut into pool tasks (the default: lowest latency for ONE frame), or every build on its caller's thread alone
// (setHierarchyBuildTasks(false)): a host that builds the hierarchies of SEVERAL frames at once on threads of its own -- the orbit
// pipeline -- gets more frames per second out of the same cores that way, because a build's tasks spend most of their life waiting
// for each other (the top-level sorts are serial stretches) and the waiting threads take the cores the other frames' builds need.
std::atomic<bool> g_build_tasks{true};
inline bool build_tasks_allowed() { return g_build_tasks.load(std::memory_order_relaxed) && BuildPool::get().workers(); }

// fn(begin, end) over [0, n) in chunks of at least `grain`, on the pool plus the calling thread
void parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)>& fn) {
    BuildPool& pool = BuildPool::get();
    size_t chunks = grain ? (n + grain - 1) / grain : 1;
    if (chunks > pool.workers() + 1) chunks = pool.workers() + 1;
    if (chunks <= 1) { if (n) fn(0, n); return; }
    const size_t step = (n + chunks - 1) / chunks;
    std::atomic<int> pending{0};
    for (size_t c = 1; c < chunks; c++) {
        const size_t b = c * step, e = (b + step < n) ? b + step : n;
        if (b >= e) continue;
        pending.fetch_add(1, std::memory_order_relaxed);
        pool.submit([&fn, &pending, b, e] { fn(b, e); pending.fetch_sub(1, std::memory_order_release); });
    }
    fn(0, step < n ? step : n);
    while (pending.load(std::memory_order_acquire) > 0)
        if (!pool.run_one()) std::this_thread::yield();
}
} // namespace

// ------------------------------------------------------------------------------------------------
// glm operations the reference's host code applies (exact op order)
// ------------------------------------------------------------------------------------------------
vec4 operator*(const mat4& m, const vec4& v) {
    // type_mat4x4.inl:562-573: Add0 = m0*v0 + m1*v1; Add1 = m2*v2 + m3*v3; Add0 + Add1
    vec4 r;
    for (int i = 0; i < 4; i++) r[i] = (m[0][i] * v[0] + m[1][i] * v[1]) + (m[2][i] * v[2] + m[3][i] * v[3]);
    return r;
}

mat4 operator*(const mat4& a, const mat4& b) {
    // type_mat4x4.inl:681-700 (unaligned path): tmp = A0*b.x; tmp += A1*b.y; tmp += A2*b.z; tmp += A3*b.w
    mat4 r;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) {
            float t = a[0][i] * b[j][0];
            t += a[1][i] * b[j][1];
            t += a[2][i] * b[j][2];
            t += a[3][i] * b[j][3];
            r[j][i] = t;
        }
    return r;
}

float radians(float degrees) { return degrees * static_cast<float>(0.01745329251994329576923690768489); }

mat4 inverse(const mat4& m) {
    // func_matrix.inl:388-446
    float Coef00 = m[2][2] * m[3][3] - m[3][2] * m[2][3];
    float Coef02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
    float Coef03 = m[1][2] * m[2][3] - m[2][2] * m[1][3];
    float Coef04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
    float Coef06 = m[1][1] * m[3][3] - m[3][1] * m[1][3];
    float Coef07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
    float Coef08 = m[2][1] * m[3][2] - m[3][1] * m[2][2];
    float Coef10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
    float Coef11 = m[1][1] * m[2][2] - m[2][1] * m[1][2];
    float Coef12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
    float Coef14 = m[1][0] * m[3][3] - m[3][0] * m[1][3];
    float Coef15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
    float Coef16 = m[2][0] * m[3][2] - m[3][0] * m[2][2];
    float Coef18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
    float Coef19 = m[1][0] * m[2][2] - m[2][0] * m[1][2];
    float Coef20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
    float Coef22 = m[1][0] * m[3][1] - m[3][0] * m[1][1];
    float Coef23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
    vec4 Fac0(Coef00, Coef00, Coef02, Coef03), Fac1(Coef04, Coef04, Coef06, Coef07), Fac2(Coef08, Coef08, Coef10, Coef11);
    vec4 Fac3(Coef12, Coef12, Coef14, Coef15), Fac4(Coef16, Coef16, Coef18, Coef19), Fac5(Coef20, Coef20, Coef22, Coef23);
    vec4 Vec0(m[1][0], m[0][0], m[0][0], m[0][0]), Vec1(m[1][1], m[0][1], m[0][1], m[0][1]);
    vec4 Vec2(m[1][2], m[0][2], m[0][2], m[0][2]), Vec3(m[1][3], m[0][3], m[0][3], m[0][3]);
    const float SignA[4] = { +1, -1, +1, -1 }, SignB[4] = { -1, +1, -1, +1 };
    mat4 Inverse;
    for (int i = 0; i < 4; i++) {
        float Inv0 = (Vec1[i] * Fac0[i] - Vec2[i] * Fac1[i]) + Vec3[i] * Fac2[i];
        float Inv1 = (Vec0[i] * Fac0[i] - Vec2[i] * Fac3[i]) + Vec3[i] * Fac4[i];
        float Inv2 = (Vec0[i] * Fac1[i] - Vec1[i] * Fac3[i]) + Vec3[i] * Fac5[i];
        float Inv3 = (Vec0[i] * Fac2[i] - Vec1[i] * Fac4[i]) + Vec2[i] * Fac5[i];
        Inverse[0][i] = Inv0 * SignA[i]; Inverse[1][i] = Inv1 * SignB[i];
        Inverse[2][i] = Inv2 * SignA[i]; Inverse[3][i] = Inv3 * SignB[i];
    }
    vec4 Row0(Inverse[0][0], Inverse[1][0], Inverse[2][0], Inverse[3][0]);
    vec4 Dot0(m[0][0] * Row0[0], m[0][1] * Row0[1], m[0][2] * Row0[2], m[0][3] * Row0[3]);
    float Dot1 = (Dot0.x + Dot0.y) + (Dot0.z + Dot0.w);
    float OneOverDeterminant = 1.0f / Dot1;
    for (int j = 0; j < 4; j++) for (int i = 0; i < 4; i++) Inverse[j][i] = Inverse[j][i] * OneOverDeterminant;
    return Inverse;
}

// ------------------------------------------------------------------------------------------------
// Transformation.cpp:6-90
// ------------------------------------------------------------------------------------------------
mat4 Transformation::scaleObj(float sx, float sy, float sz) {
    mat4 m(0.0f); m[0][0] = sx; m[1][1] = sy; m[2][2] = sz; m[3][3] = 1.0f; return m;
}
mat4 Transformation::rotateObjX(float degree) {
    mat4 m(0.0f);
    m[0][0] = 1; m[1][1] = std::cos(degree); m[1][2] = -std::sin(degree);
    m[2][1] = std::sin(degree); m[2][2] = std::cos(degree); m[3][3] = 1.0f;
    return m;
}
mat4 Transformation::rotateObjY(float degree) {
    mat4 m(0.0f);
    m[0][0] = std::cos(degree); m[0][2] = std::sin(degree); m[1][1] = 1.0f;
    m[2][0] = -std::sin(degree); m[2][2] = std::cos(degree); m[3][3] = 1.0f;
    return m;
}
mat4 Transformation::rotateObjZ(float degree) {
    mat4 m(0.0f);
    m[0][0] = std::cos(degree); m[0][1] = -std::sin(degree);
    m[1][0] = std::sin(degree); m[1][1] = std::cos(degree); m[2][2] = 1.0f; m[3][3] = 1.0f;
    return m;
}
mat4 Transformation::mirrorObj(bool mirrorX, bool mirrorY, bool mirrorZ) {
    mat4 m(1.0f);
    if (mirrorX) m[0][0] = -1.0f;
    if (mirrorY) m[1][1] = -1.0f;
    if (mirrorZ) m[2][2] = -1.0f;
    return m;
}
mat4 Transformation::shearObj(float shearXY, float shearXZ, float shearYX, float shearYZ, float shearZX, float shearZY) {
    mat4 m(1.0f);
    m[1][0] = shearXY; m[2][0] = shearXZ; m[0][1] = shearYX; m[2][1] = shearYZ; m[0][2] = shearZX; m[1][2] = shearZY;
    return m;
}
mat4 Transformation::changeObjPosition(vec3 position) {
    mat4 m(1.0f); m[3] = vec4(position, 1.0f); return m;
}
mat4 Transformation::createViewMatrix(vec3 position, vec3 rotation) {
    mat4 m = changeObjPosition(position);
    m = m * rotateObjZ(rotation.z);
    m = m * rotateObjY(rotation.y);
    m = m * rotateObjX(rotation.x);
    return m;
}

// ------------------------------------------------------------------------------------------------
// Texture decoding for the loader (the reference uses stbi_load(path, ..., 3), Object.cpp:57).
// Supported here: PNG (8/16-bit, grey / RGB / palette / alpha, non-interlaced), JPEG (baseline and progressive
// Huffman, srt_jpeg.cpp), binary PPM, 24/32-bit BMP.  Anything else fails to load, which the reference also
// tolerates (:63-65).
// ------------------------------------------------------------------------------------------------
static bool read_file(const std::string& path, std::vector<unsigned char>& out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    out.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
    return true;
}
static uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static bool decode_png(const std::vector<unsigned char>& d, Texture& t) {
    static const unsigned char sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (d.size() < 33 || std::memcmp(d.data(), sig, 8) != 0) return false;
    uint32_t W = 0, H = 0; int depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte;
    size_t pos = 8;
    while (pos + 12 <= d.size()) {
        uint32_t len = be32(&d[pos]); const unsigned char* ty = &d[pos + 4];
        if (pos + 12 + (size_t)len > d.size()) return false;
        const unsigned char* body = &d[pos + 8];
        if (!std::memcmp(ty, "IHDR", 4)) { W = be32(body); H = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (!std::memcmp(ty, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(ty, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(ty, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (!W || !H || interlace || (depth != 8 && depth != 16 && ctype != 3)) return false;
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!ch || (ctype == 3 && depth != 8)) return false;
    const size_t bpp = (size_t)ch * depth / 8, stride = (size_t)W * bpp;
    std::vector<unsigned char> raw((stride + 1) * H);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return false;
    std::vector<unsigned char> img(stride * H);
    for (uint32_t y = 0; y < H; y++) {
        const unsigned char* in = &raw[(stride + 1) * y]; unsigned char* out = &img[stride * y];
        const unsigned char* up = y ? &img[stride * (y - 1)] : nullptr;
        const int ft = in[0]; in++;
        for (size_t i = 0; i < stride; i++) {
            int a = i >= bpp ? out[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0, x = in[i];
            switch (ft) {
            case 0: break;
            case 1: x += a; break;
            case 2: x += b; break;
            case 3: x += (a + b) >> 1; break;
            case 4: { int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                      x += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
            default: return false;
            }
            out[i] = (unsigned char)x;
        }
    }
    t.dim.x = (int)W; t.dim.y = (int)H; t.rgb.resize((size_t)W * H * 3);
    const size_t step = depth / 8;      // 16-bit samples: keep the high byte (stb_image does the same)
    for (size_t i = 0; i < (size_t)W * H; i++) {
        const unsigned char* px = &img[i * bpp]; unsigned char* o = &t.rgb[i * 3];
        if (ctype == 3) { size_t k = (size_t)px[0] * 3; if (k + 2 >= plte.size()) return false; o[0] = plte[k]; o[1] = plte[k + 1]; o[2] = plte[k + 2]; }
        else if (ch <= 2) { o[0] = o[1] = o[2] = px[0]; }
        else { o[0] = px[0]; o[1] = px[step]; o[2] = px[2 * step]; }
    }
    return true;
}

static bool decode_ppm(const std::vector<unsigned char>& d, Texture& t) {
    if (d.size() < 11 || d[0] != 'P' || d[1] != '6') return false;
    size_t pos = 2; int vals[3], n = 0;
    while (n < 3 && pos < d.size()) {
        while (pos < d.size() && (std::isspace(d[pos]) || d[pos] == '#')) { if (d[pos] == '#') while (pos < d.size() && d[pos] != '\n') pos++; else pos++; }
        int v = 0; bool any = false;
        while (pos < d.size() && std::isdigit(d[pos])) { v = v * 10 + (d[pos] - '0'); pos++; any = true; }
        if (!any) return false;
        vals[n++] = v;
    }
    pos++;
    if (n < 3 || vals[2] != 255 || pos + (size_t)vals[0] * vals[1] * 3 > d.size()) return false;
    t.dim.x = vals[0]; t.dim.y = vals[1]; t.rgb.assign(d.begin() + pos, d.begin() + pos + (size_t)vals[0] * vals[1] * 3);
    return true;
}

static bool decode_bmp(const std::vector<unsigned char>& d, Texture& t) {
    if (d.size() < 54 || d[0] != 'B' || d[1] != 'M') return false;
    auto le32 = [&](size_t o) { return (int32_t)((uint32_t)d[o] | ((uint32_t)d[o + 1] << 8) | ((uint32_t)d[o + 2] << 16) | ((uint32_t)d[o + 3] << 24)); };
    const int32_t off = le32(10), W = le32(18), Hs = le32(22); const int bpp = d[28] | (d[29] << 8), comp = le32(30);
    if (W <= 0 || Hs == 0 || (bpp != 24 && bpp != 32) || comp != 0) return false;
    const int32_t H = std::abs(Hs); const size_t stride = (((size_t)W * bpp / 8) + 3) & ~(size_t)3;
    if ((size_t)off + stride * H > d.size()) return false;
    t.dim.x = W; t.dim.y = H; t.rgb.resize((size_t)W * H * 3);
    for (int32_t y = 0; y < H; y++) {
        const unsigned char* row = &d[off + stride * (Hs > 0 ? (H - 1 - y) : y)];
        for (int32_t x = 0; x < W; x++) { const unsigned char* p = row + (size_t)x * bpp / 8; unsigned char* o = &t.rgb[((size_t)y * W + x) * 3]; o[0] = p[2]; o[1] = p[1]; o[2] = p[0]; }
    }
    return true;
}

bool load_texture(const std::string& path, Texture& t) {
    std::vector<unsigned char> d;
    if (!read_file(path, d)) return false;
    return decode_png(d, t) || decode_jpeg(d, t) || decode_ppm(d, t) || decode_bmp(d, t);
}

// ------------------------------------------------------------------------------------------------
// loadObjFile, Object.cpp:25-170.  OBJ / MTL parsing restates what tinyobjloader's ObjReader does for
// the statements the reference's assets use (v, vt, vn, f, usemtl, mtllib; newmtl, map_Kd):
// triangulation on (quads split along the shorter diagonal, tiny_obj_loader.h:1562-1605; larger
// polygons by earcut, :1618-1740), faces in file order.
// ------------------------------------------------------------------------------------------------
namespace {
struct ObjIndex { int v = -1, vt = -1, vn = -1; };
struct ObjFace { ObjIndex i[3]; int material = -1; };

bool fix_index(int idx, int n, int& out) {          // tinyobj fixIndex: 1-based, negative = relative
    if (idx > 0) { out = idx - 1; return true; }
    if (idx < 0) { out = n + idx; return out >= 0; }
    return false;
}

bool parse_triple(const char*& p, int nv, int nvt, int nvn, ObjIndex& o) {
    char* e = nullptr;
    long a = std::strtol(p, &e, 10);
    if (e == p || !fix_index((int)a, nv, o.v)) return false;
    p = e;
    if (*p != '/') return true;
    p++;
    if (*p == '/') { p++; long c = std::strtol(p, &e, 10); if (e != p) { fix_index((int)c, nvn, o.vn); p = e; } return true; }
    long b = std::strtol(p, &e, 10);
    if (e != p) { fix_index((int)b, nvt, o.vt); p = e; }
    if (*p == '/') { p++; long c = std::strtol(p, &e, 10); if (e != p) { fix_index((int)c, nvn, o.vn); p = e; } }
    return true;
}

// Faces with more than four corners.  The reference builds tinyobjloader with TINYOBJLOADER_USE_MAPBOX_EARCUT
// (simple_raytracer.cpp:16), so such a face is projected onto the plane of its Newell normal in f32
// (tiny_obj_loader.h:1618-1701) and cut by mapbox earcut in f64 (mapbox/earcut.hpp).  Which triangles come out, and in
// which order and rotation, is the algorithm's: ear slicing over a doubly linked ring that skips a vertex after every
// cut, with the fall-backs filterPoints -> cureLocalIntersections -> splitEarcut.  Restated here for one ring without
// holes (tinyobj never passes holes); the z-order hash earcut switches on above 80 vertices only narrows the set of
// points an ear test visits and does not change its answer, so the plain test is used for every size.
class Earcut {
public:
    std::vector<uint32_t> indices;
    explicit Earcut(const std::vector<std::array<float, 2>>& ring) {
        const size_t len = ring.size();
        n_.reserve(len * 3 / 2 + 4);
        double sum = 0;                                                   // winding (earcut.hpp linkedList)
        for (size_t i = 0, j = len ? len - 1 : 0; i < len; j = i++) {
            const double p10 = ring[i][0], p11 = ring[i][1], p20 = ring[j][0], p21 = ring[j][1];
            sum += (p20 - p10) * (p11 + p21);
        }
        int last = -1;
        if (sum > 0) for (size_t i = 0; i < len; i++) last = insert((uint32_t)i, ring[i][0], ring[i][1], last);
        else         for (size_t i = len; i-- > 0;)   last = insert((uint32_t)i, ring[i][0], ring[i][1], last);
        if (last >= 0 && equals(last, n_[last].next)) { remove(last); last = n_[last].next; }
        if (last < 0 || n_[last].prev == n_[last].next) return;
        slice(last, 0);
    }

private:
    struct Node { uint32_t i; double x, y; int prev, next; };
    std::vector<Node> n_;
    int insert(uint32_t i, double x, double y, int last) {
        const int p = (int)n_.size();
        n_.push_back({ i, x, y, p, p });
        if (last >= 0) { n_[p].next = n_[last].next; n_[p].prev = last; n_[n_[last].next].prev = p; n_[last].next = p; }
        return p;
    }
    void remove(int p) { n_[n_[p].next].prev = n_[p].prev; n_[n_[p].prev].next = n_[p].next; }
    double area(int p, int q, int r) const { return (n_[q].y - n_[p].y) * (n_[r].x - n_[q].x) - (n_[q].x - n_[p].x) * (n_[r].y - n_[q].y); }
    bool equals(int a, int b) const { return n_[a].x == n_[b].x && n_[a].y == n_[b].y; }
    static bool in_triangle(double ax, double ay, double bx, double by, double cx, double cy, double px, double py) {
        return (cx - px) * (ay - py) - (ax - px) * (cy - py) >= 0 && (ax - px) * (by - py) - (bx - px) * (ay - py) >= 0 &&
               (bx - px) * (cy - py) - (cx - px) * (by - py) >= 0;
    }
    bool is_ear(int ear) const {
        const int a = n_[ear].prev, b = ear, c = n_[ear].next;
        if (area(a, b, c) >= 0) return false;                             // reflex
        for (int p = n_[c].next; p != a; p = n_[p].next)
            if (in_triangle(n_[a].x, n_[a].y, n_[b].x, n_[b].y, n_[c].x, n_[c].y, n_[p].x, n_[p].y) && area(n_[p].prev, p, n_[p].next) >= 0) return false;
        return true;
    }
    int filter(int start, int end = -1) {                                 // drop duplicate / collinear points
        if (end < 0) end = start;
        int p = start;
        bool again;
        do {
            again = false;
            if (equals(p, n_[p].next) || area(n_[p].prev, p, n_[p].next) == 0) {
                remove(p);
                p = end = n_[p].prev;
                if (p == n_[p].next) break;
                again = true;
            } else p = n_[p].next;
        } while (again || p != end);
        return end;
    }
    void slice(int ear, int pass) {
        int stop = ear;
        while (n_[ear].prev != n_[ear].next) {
            const int prev = n_[ear].prev, next = n_[ear].next;
            if (is_ear(ear)) {
                indices.push_back(n_[prev].i); indices.push_back(n_[ear].i); indices.push_back(n_[next].i);
                remove(ear);
                ear = n_[next].next; stop = n_[next].next;                // skipping a vertex gives fewer slivers
                continue;
            }
            ear = next;
            if (ear == stop) {                                            // went round without finding an ear
                if (pass == 0) slice(filter(ear), 1);
                else if (pass == 1) { ear = cure(filter(ear)); slice(ear, 2); }
                else split(ear);
                break;
            }
        }
    }
    static int sign(double v) { return (0.0 < v) - (v < 0.0); }
    bool on_segment(int p, int q, int r) const {
        return n_[q].x <= std::max(n_[p].x, n_[r].x) && n_[q].x >= std::min(n_[p].x, n_[r].x) &&
               n_[q].y <= std::max(n_[p].y, n_[r].y) && n_[q].y >= std::min(n_[p].y, n_[r].y);
    }
    bool intersects(int p1, int q1, int p2, int q2) const {
        const int o1 = sign(area(p1, q1, p2)), o2 = sign(area(p1, q1, q2)), o3 = sign(area(p2, q2, p1)), o4 = sign(area(p2, q2, q1));
        if (o1 != o2 && o3 != o4) return true;
        if (o1 == 0 && on_segment(p1, p2, q1)) return true;
        if (o2 == 0 && on_segment(p1, q2, q1)) return true;
        if (o3 == 0 && on_segment(p2, p1, q2)) return true;
        if (o4 == 0 && on_segment(p2, q1, q2)) return true;
        return false;
    }
    bool locally_inside(int a, int b) const {
        return area(n_[a].prev, a, n_[a].next) < 0 ? area(a, b, n_[a].next) >= 0 && area(a, n_[a].prev, b) >= 0
                                                   : area(a, b, n_[a].prev) < 0 || area(a, n_[a].next, b) < 0;
    }
    int cure(int start) {                                                 // small local self-intersections
        int p = start;
        do {
            const int a = n_[p].prev, b = n_[n_[p].next].next;
            if (!equals(a, b) && intersects(a, p, n_[p].next, b) && locally_inside(a, b) && locally_inside(b, a)) {
                indices.push_back(n_[a].i); indices.push_back(n_[p].i); indices.push_back(n_[b].i);
                remove(p); remove(n_[p].next);
                p = start = b;
            }
            p = n_[p].next;
        } while (p != start);
        return filter(p);
    }
    bool intersects_polygon(int a, int b) const {
        int p = a;
        do {
            const int q = n_[p].next;
            if (n_[p].i != n_[a].i && n_[q].i != n_[a].i && n_[p].i != n_[b].i && n_[q].i != n_[b].i && intersects(p, q, a, b)) return true;
            p = q;
        } while (p != a);
        return false;
    }
    bool middle_inside(int a, int b) const {
        int p = a;
        bool inside = false;
        const double px = (n_[a].x + n_[b].x) / 2, py = (n_[a].y + n_[b].y) / 2;
        do {
            const int q = n_[p].next;
            if (((n_[p].y > py) != (n_[q].y > py)) && n_[q].y != n_[p].y &&
                (px < (n_[q].x - n_[p].x) * (py - n_[p].y) / (n_[q].y - n_[p].y) + n_[p].x)) inside = !inside;
            p = q;
        } while (p != a);
        return inside;
    }
    bool valid_diagonal(int a, int b) const {
        return n_[n_[a].next].i != n_[b].i && n_[n_[a].prev].i != n_[b].i && !intersects_polygon(a, b) &&
               ((locally_inside(a, b) && locally_inside(b, a) && middle_inside(a, b) &&
                 (area(n_[a].prev, a, n_[b].prev) != 0.0 || area(a, n_[b].prev, b) != 0.0)) ||
                (equals(a, b) && area(n_[a].prev, a, n_[a].next) > 0 && area(n_[b].prev, b, n_[b].next) > 0));
    }
    int split_polygon(int a, int b) {
        const int a2 = (int)n_.size(); n_.push_back({ n_[a].i, n_[a].x, n_[a].y, 0, 0 });
        const int b2 = (int)n_.size(); n_.push_back({ n_[b].i, n_[b].x, n_[b].y, 0, 0 });
        const int an = n_[a].next, bp = n_[b].prev;
        n_[a].next = b; n_[b].prev = a;
        n_[a2].next = an; n_[an].prev = a2;
        n_[b2].next = a2; n_[a2].prev = b2;
        n_[bp].next = b2; n_[b2].prev = bp;
        return b2;
    }
    void split(int start) {                                               // last resort: cut along a valid diagonal
        int a = start;
        do {
            int b = n_[n_[a].next].next;
            while (b != n_[a].prev) {
                if (n_[a].i != n_[b].i && valid_diagonal(a, b)) {
                    int c = split_polygon(a, b);
                    a = filter(a, n_[a].next);
                    c = filter(c, n_[c].next);
                    slice(a, 0); slice(c, 0);
                    return;
                }
                b = n_[b].next;
            }
            a = n_[a].next;
        } while (a != start);
    }
};

// corner indices (into the face) of the triangles tinyobjloader makes of a face with more than four corners
std::vector<uint32_t> triangulate_polygon(const std::vector<ObjIndex>& poly, const std::vector<float>& V) {
    const size_t n = poly.size();
    float nx = 0.f, ny = 0.f, nz = 0.f;                                    // Newell normal, f32 (:1624-1652)
    for (size_t k = 0; k < n; k++) {
        const float* p1 = &V[3 * (size_t)poly[k].v]; const float* p2 = &V[3 * (size_t)poly[(k + 1) % n].v];
        const float ax = p1[0] - p2[0], ay = p1[1] - p2[1], az = p1[2] - p2[2];
        const float bx = p1[0] + p2[0], by = p1[1] + p2[1], bz = p1[2] + p2[2];
        nx += ay * bz; ny += az * bx; nz += ax * by;
    }
    const float len = std::sqrt(nx * nx + ny * ny + nz * nz);
    if (len <= 0) return {};                                               // zero normal: the face is dropped (:1655-1657)
    const float inv = -1.0f / len;
    const float wx = nx * inv, wy = ny * inv, wz = nz * inv;
    const float ax = std::fabs(wx) > 0.9999999f ? 0.f : 1.f, ay = std::fabs(wx) > 0.9999999f ? 1.f : 0.f, az = 0.f;
    float vx = wy * az - wz * ay, vy = wz * ax - wx * az, vz = wx * ay - wy * ax;      // cross(axis_w, a)
    const float il = 1.0f / std::sqrt(vx * vx + vy * vy + vz * vz);
    vx *= il; vy *= il; vz *= il;
    const float ux = wy * vz - wz * vy, uy = wz * vx - wx * vz, uz = wx * vy - wy * vx;  // cross(axis_w, axis_v)
    std::vector<std::array<float, 2>> ring(n);
    for (size_t k = 0; k < n; k++) {
        const float* q = &V[3 * (size_t)poly[k].v];
        ring[k] = { q[0] * ux + q[1] * uy + q[2] * uz, q[0] * vx + q[1] * vy + q[2] * vz };
    }
    return Earcut(ring).indices;
}

std::string dirname_of(const std::string& path) {
    size_t k = path.find_last_of("/\\");
    return k == std::string::npos ? std::string() : path.substr(0, k + 1);
}
std::string trim(const std::string& s) {
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
} // namespace

// The reference's main() loads every OBJ again for every frame of the orbit (simple_raytracer.cpp:534-618); parsing the
// bunny and decoding a 1024x1024 JPEG cost more than building the hierarchies and rendering the frame.  A parsed asset
// is therefore kept per process, keyed by path and valid while the OBJ file's size and modification time are unchanged
// (its MTL and texture files are assumed to change with it); SRT_HOST_NO_ASSET_CACHE=1 turns the cache off.
namespace {
struct CachedAsset {
    long long mtime_ns = 0, size = 0;
    std::vector<Triangle> triangles;
    std::vector<std::pair<std::string, Texture>> textures;     // diffuse maps this OBJ's materials name, decoded
    std::vector<std::string> failed;                            // ... and the ones that did not decode
};
std::mutex g_asset_mu;
std::unordered_map<std::string, std::shared_ptr<const CachedAsset>> g_assets;
bool file_stamp(const std::string& path, long long& mtime_ns, long long& size) {
    struct stat st;
    if (::stat(path.c_str(), &st) != 0) return false;
    mtime_ns = (long long)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec; size = (long long)st.st_size;
    return true;
}
bool asset_cache_enabled() { const char* e = std::getenv("SRT_HOST_NO_ASSET_CACHE"); return !(e && *e && *e != '0'); }
} // namespace

void ObjectManager::loadObjFile(const std::string& objFilename) {
    objColors[objFilename] = vec3(1.f, 0.f, 0.f);                        // :29
    objProperties[objFilename] = vec3(0.2f, 0.5f, 15.0f);                // :31-34
    long long stamp_mtime = 0, stamp_size = 0;
    const bool cacheable = asset_cache_enabled() && file_stamp(objFilename, stamp_mtime, stamp_size);
    if (cacheable) {
        std::shared_ptr<const CachedAsset> hit;
        { std::lock_guard<std::mutex> g(g_asset_mu); auto it = g_assets.find(objFilename); if (it != g_assets.end()) hit = it->second; }
        if (hit && hit->mtime_ns == stamp_mtime && hit->size == stamp_size) {
            for (const auto& t : hit->textures) if (!textureData.count(t.first)) textureData[t.first] = t.second;
            for (const std::string& f : hit->failed) if (!textureData.count(f)) std::cerr << "Failed to load texture: " << f << std::endl;
            objTriangles[objFilename] = hit->triangles;
            return;
        }
    }
    std::vector<float> V, VT, VN;
    std::vector<ObjFace> faces;
    std::vector<std::string> mat_names, mat_tex;
    std::ifstream in(objFilename);
    if (!in) {
        std::cerr << "TinyObjReader: Cannot open file [" << objFilename << "]\n";      // :35-39 prints and carries on
        objTriangles[objFilename] = {};
        return;
    }
    int cur_mat = -1;
    std::string line;
    while (std::getline(in, line)) {
        const char* p = line.c_str();
        while (*p == ' ' || *p == '\t') p++;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            char* e; float x = std::strtof(p + 2, &e), y = std::strtof(e, &e), z = std::strtof(e, &e);
            V.push_back(x); V.push_back(y); V.push_back(z);
        } else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) {
            char* e; float u = std::strtof(p + 3, &e), v = std::strtof(e, &e);
            VT.push_back(u); VT.push_back(v);
        } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
            char* e; float x = std::strtof(p + 3, &e), y = std::strtof(e, &e), z = std::strtof(e, &e);
            VN.push_back(x); VN.push_back(y); VN.push_back(z);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            std::vector<ObjIndex> poly;
            const int nv = (int)V.size() / 3, nvt = (int)VT.size() / 2, nvn = (int)VN.size() / 3;
            while (true) {
                while (*p == ' ' || *p == '\t') p++;
                if (!*p || *p == '\r' || *p == '\n' || *p == '#') break;
                ObjIndex o;
                if (!parse_triple(p, nv, nvt, nvn, o)) break;
                poly.push_back(o);
            }
            auto emit = [&](int a, int b, int c) { ObjFace f; f.i[0] = poly[a]; f.i[1] = poly[b]; f.i[2] = poly[c]; f.material = cur_mat; faces.push_back(f); };
            if (poly.size() == 3) emit(0, 1, 2);
            else if (poly.size() == 4) {
                bool ok = true;
                for (auto& q : poly) ok &= (q.v >= 0 && 3 * (size_t)q.v + 2 < V.size());
                if (!ok) continue;
                const float* v0 = &V[3 * poly[0].v], *v1 = &V[3 * poly[1].v], *v2 = &V[3 * poly[2].v], *v3 = &V[3 * poly[3].v];
                float e02x = v2[0] - v0[0], e02y = v2[1] - v0[1], e02z = v2[2] - v0[2];
                float e13x = v3[0] - v1[0], e13y = v3[1] - v1[1], e13z = v3[2] - v1[2];
                float sqr02 = e02x * e02x + e02y * e02y + e02z * e02z, sqr13 = e13x * e13x + e13y * e13y + e13z * e13z;
                if (sqr02 < sqr13) { emit(0, 1, 2); emit(0, 2, 3); } else { emit(0, 1, 3); emit(1, 2, 3); }
            } else if (poly.size() > 4) {
                bool ok = true;
                for (auto& q : poly) ok &= (q.v >= 0 && 3 * (size_t)q.v + 2 < V.size());
                if (!ok) continue;
                const std::vector<uint32_t> tri = triangulate_polygon(poly, V);
                for (size_t k = 0; k + 2 < tri.size(); k += 3) emit((int)tri[k], (int)tri[k + 1], (int)tri[k + 2]);
            }
        } else if (!std::strncmp(p, "usemtl", 6)) {
            std::string name = trim(p + 6);
            cur_mat = -1;
            for (size_t k = 0; k < mat_names.size(); k++) if (mat_names[k] == name) cur_mat = (int)k;
        } else if (!std::strncmp(p, "mtllib", 6)) {
            std::string mtl = dirname_of(objFilename) + trim(p + 6);
            std::ifstream mf(mtl);
            std::string ml;
            while (mf && std::getline(mf, ml)) {
                std::string s = trim(ml);
                if (!s.compare(0, 6, "newmtl")) { mat_names.push_back(trim(s.substr(6))); mat_tex.emplace_back(); }
                else if (!s.compare(0, 6, "map_Kd") && !mat_names.empty()) {
                    std::istringstream ss(s.substr(6)); std::string tok, last;
                    while (ss >> tok) last = tok;          // options (-s, -o ...) precede the file name
                    mat_tex.back() = last;
                }
            }
        }
    }

    std::shared_ptr<CachedAsset> fresh = cacheable ? std::make_shared<CachedAsset>() : nullptr;
    for (const std::string& texturePath : mat_tex) {                      // :52-68
        if (texturePath.empty()) continue;
        if (textureData.count(texturePath)) {
            if (fresh) fresh->textures.emplace_back(texturePath, textureData[texturePath]);
            continue;
        }
        Texture t;
        if (load_texture(texturePath, t)) {
            if (fresh) fresh->textures.emplace_back(texturePath, t);
            textureData[texturePath] = std::move(t);
        } else {
            if (fresh) fresh->failed.push_back(texturePath);
            std::cerr << "Failed to load texture: " << texturePath << std::endl;
        }
    }

    std::vector<Triangle> triangles;
    triangles.reserve(faces.size());
    for (const ObjFace& f : faces) {                                      // :70-167
        Triangle tria;
        for (int v = 0; v < 3; v++) {
            std::string texturePath;
            vec2 texCoordinate(0, 0);
            vec4 vertex(0.0f, 0.0f, 0.0f, 1.0f);
            vec3 vnormal(0.0f, 0.0f, 0.0f), vcolor(1.0f, 1.0f, 1.0f);
            const ObjIndex& idx = f.i[v];
            if (idx.v >= 0 && 3 * (size_t)idx.v + 2 < V.size()) { vertex.x = V[3 * idx.v]; vertex.y = V[3 * idx.v + 1]; vertex.z = V[3 * idx.v + 2]; }
            if (idx.vn >= 0 && 3 * (size_t)idx.vn + 2 < VN.size()) { vnormal = vec3(VN[3 * idx.vn], VN[3 * idx.vn + 1], VN[3 * idx.vn + 2]); }
            if (idx.vt >= 0 && 2 * (size_t)idx.vt + 1 < VT.size()) {
                const float tx = VT[2 * idx.vt], ty = VT[2 * idx.vt + 1];
                if (f.material >= 0 && f.material < (int)mat_names.size()) {
                    texturePath = mat_tex[f.material];
                    auto it = texturePath.empty() ? textureData.end() : textureData.find(texturePath);
                    if (it != textureData.end() && it->second.dim.x > 0 && it->second.dim.y > 0) {
                        const ivec2 texDim = it->second.dim;
                        int u = static_cast<int>(std::floor(tx * texDim.x)) % texDim.x;              // :113
                        int w = static_cast<int>(std::floor((1.0f - ty) * texDim.y)) % texDim.y;     // :114
                        u = (u + texDim.x) % texDim.x; w = (w + texDim.y) % texDim.y;                // :116-117
                        texCoordinate = vec2((float)u, (float)w);
                        const size_t texIndex = ((size_t)w * texDim.x + u) * 3;
                        vcolor = vec3(it->second.rgb[texIndex] / 255.0f, it->second.rgb[texIndex + 1] / 255.0f, it->second.rgb[texIndex + 2] / 255.0f);
                    }
                }
            }
            if (v == 0) { tria.pointOne = vertex; tria.normalOne = vnormal; tria.colorOneCoordinate = texCoordinate; tria.color = vcolor; tria.textureName = texturePath; }
            if (v == 1) { tria.pointTwo = vertex; tria.normalTwo = vnormal; tria.colorTwoCoordinate = texCoordinate; }
            if (v == 2) { tria.pointThree = vertex; tria.normalThree = vnormal; tria.colorThreeCoordinate = texCoordinate; }
        }
        triangles.push_back(tria);
    }
    if (fresh) {
        fresh->mtime_ns = stamp_mtime; fresh->size = stamp_size; fresh->triangles = triangles;
        std::lock_guard<std::mutex> g(g_asset_mu);
        g_assets[objFilename] = fresh;
    }
    objTriangles[objFilename] = std::move(triangles);
}

const std::vector<Triangle>& ObjectManager::getTriangles(const std::string& objFilename) const { return objTriangles.at(objFilename); }
void ObjectManager::setTriangles(const std::string& objFilename, const std::vector<Triangle>& triangles) { objTriangles[objFilename] = triangles; }
void ObjectManager::setColor(const std::string& objFilename, const vec3& color) { objColors[objFilename] = color; }
vec3 ObjectManager::getColor(const std::string& objFilename) const { return objColors.at(objFilename); }

void ObjectManager::transformTriangles(const std::string& objFilename, const mat4& matrix) {
    std::vector<Triangle>& triangles = objTriangles[objFilename];      // operator[]: creates an empty object, like the reference
    parallel_for(triangles.size(), 8192, [&](size_t b, size_t e) {
        for (size_t i = b; i < e; i++) {
            Triangle& t = triangles[i];
            t.pointOne = matrix * t.pointOne;
            t.pointTwo = matrix * t.pointTwo;
            t.pointThree = matrix * t.pointThree;
        }
    });
}

// ------------------------------------------------------------------------------------------------
// createBoundingHierarchy, Object.cpp:205-284.  Same tree, same leaf order as the reference builds:
// std::sort (libstdc++ introsort) over the node's triangles with the reference's three comparators.
// The permutation std::sort produces depends only on the comparison outcomes (first vertices are shared
// between triangles, so equal keys are the rule, and their order is introsort's), so the node's
// (key, index) pairs are sorted with the same '<' on the key: the reference's order without moving
// 152-byte Triangles.  The tree's shape depends only on the triangle count (halve while > 8), so every
// node's pre-order slot is known before it is built and disjoint subtrees are built by parallel threads.
// Per-frame cost is what matters here: the reference re-transforms and re-builds every frame
// (simple_raytracer.cpp:534-618), and at HIP render times the build is the frame.
// ------------------------------------------------------------------------------------------------
namespace {
struct KeyIdx { float key; uint32_t idx; };

// std::sort as libstdc++ runs it, with its independent halves on different threads.  The order std::sort leaves equal keys
// in is the reference's leaf order, so the result has to be libstdc++'s to the element: introsort = quicksort (pivot =
// median of first + 1, middle, last - 1 moved to the front; Hoare-style unguarded partition around it), recursing into the
// right part and looping on the left, 2 * floor(log2 n) levels deep before it falls back to heapsort, ranges of <= 16 left
// for one final insertion-sort pass.  The right part and the left part never touch each other's elements, so the recursion
// runs as a pool task and the final pass runs when all tasks are done -- the same comparisons on the same ranges, in another
// order in time.  tests/test_host_mirror.py compares it with std::sort on arrays full of equal keys.
struct ExactSort {
    std::atomic<int> pending{0};
    bool parallel = true;                    // the right parts of big partitions as pool tasks
    static bool less(const KeyIdx& a, const KeyIdx& b) { return a.key < b.key; }
    static void median_to_first(KeyIdx* result, KeyIdx* a, KeyIdx* b, KeyIdx* c) {
        if (less(*a, *b)) {
            if (less(*b, *c)) std::iter_swap(result, b);
            else if (less(*a, *c)) std::iter_swap(result, c);
            else std::iter_swap(result, a);
        } else if (less(*a, *c)) std::iter_swap(result, a);
        else if (less(*b, *c)) std::iter_swap(result, c);
        else std::iter_swap(result, b);
    }
    // libstdc++'s __unguarded_partition: the k-th element from the left that is not < pivot is swapped with the k-th from the right
    // that is not > pivot, for as long as the former lies left of the latter; the scan that finds them mispredicts on every other
    // element of a random array.  While the two cursors are more than two blocks apart the same pairs are found WITHOUT branches:
    // the positions of the stops of a block of 64 elements are collected into a small array (offs[n] = i; n += stop -- no jump), the
    // pairs are swapped in order, the block whose stops are used up is left behind.  Blocks from the left never overlap blocks from
    // the right there, so every pair is one the scalar scan would have made, in the scan's order; the scalar loop then takes over
    // in the state it would have been in at that point (cursor on the first stop not yet swapped) and ends the partition -- same
    // swaps, same return value.  Alone the sort of 69,451 keys goes from 5.3 to 3.5 ms; inside the build (children inherit their parent.s
    // order, so half the partitions scan sorted input and predict well anyway) the sorts of a bunny frame go from 35 to 30 ms of one core.
    static KeyIdx* partition(KeyIdx* first, KeyIdx* last, KeyIdx* pivot) {
        constexpr ptrdiff_t B = 64;
        const float pv = pivot->key;
        uint8_t off_l[B], off_r[B];
        ptrdiff_t n_l = 0, n_r = 0, s_l = 0, s_r = 0;
        KeyIdx* l = first; KeyIdx* r = last;                                  // [l, l + B) / [r - B, r): the active blocks
        for (;;) {
            const ptrdiff_t room = (r - (n_r ? B : 0)) - (l + (n_l ? B : 0));     // elements no block covers yet
            if (room < ((n_l ? 0 : B) + (n_r ? 0 : B))) break;
            if (!n_l) { s_l = 0; for (ptrdiff_t i = 0; i < B; i++) { off_l[n_l] = (uint8_t)i; n_l += !(l[i].key < pv); } }
            if (!n_r) { s_r = 0; for (ptrdiff_t i = 0; i < B; i++) { off_r[n_r] = (uint8_t)i; n_r += !(pv < (r - 1 - i)->key); } }
            const ptrdiff_t m = n_l < n_r ? n_l : n_r;
            for (ptrdiff_t k = 0; k < m; k++) std::iter_swap(l + off_l[s_l + k], r - 1 - off_r[s_r + k]);
            n_l -= m; n_r -= m; s_l += m; s_r += m;
            if (!n_l) l += B;
            if (!n_r) r -= B;
        }
        first = n_l ? l + off_l[s_l] : l;                                     // the left scan would stand on the first stop not yet swapped
        last = n_r ? r - off_r[s_r] : r;                                      // the right scan's next --last lands on its first such stop
        for (;;) {
            while (less(*first, *pivot)) ++first;
            --last;
            while (less(*pivot, *last)) --last;
            if (!(first < last)) return first;
            std::iter_swap(first, last);
            ++first;
        }
    }
    void loop(KeyIdx* first, KeyIdx* last, int depth_limit) {
        while (last - first > 16) {
            if (depth_limit == 0) { std::partial_sort(first, last, last, less); return; }     // the library's own heapsort fallback
            --depth_limit;
            KeyIdx* mid = first + (last - first) / 2;
            median_to_first(first, first + 1, mid, last - 1);
            KeyIdx* cut = partition(first + 1, last, first);
            if (parallel && last - cut >= PAR_MIN && build_tasks_allowed()) {
                pending.fetch_add(1, std::memory_order_relaxed);
                BuildPool::get().submit([this, cut, last, depth_limit] { loop(cut, last, depth_limit); pending.fetch_sub(1, std::memory_order_release); });
            } else loop(cut, last, depth_limit);
            last = cut;
        }
    }
    static void linear_insert(KeyIdx* last) {
        KeyIdx val = *last; KeyIdx* next = last - 1;
        while (less(val, *next)) { *last = *next; last = next; --next; }
        *last = val;
    }
    static void insertion(KeyIdx* first, KeyIdx* last) {
        if (first == last) return;
        for (KeyIdx* i = first + 1; i != last; ++i) {
            if (less(*i, *first)) { KeyIdx val = *i; std::move_backward(first, i, i + 1); *first = val; }
            else linear_insert(i);
        }
    }
    void sort(KeyIdx* first, KeyIdx* last) {
        if (first == last) return;
        int lg = 0; for (size_t n = (size_t)(last - first); n > 1; n >>= 1) lg++;
        loop(first, last, 2 * lg);
        while (pending.load(std::memory_order_acquire) > 0)
            if (!BuildPool::get().run_one()) std::this_thread::yield();
        if (last - first > 16) { insertion(first, first + 16); for (KeyIdx* i = first + 16; i != last; ++i) linear_insert(i); }
        else insertion(first, last);
    }
    static constexpr ptrdiff_t PAR_MIN = 4096;
};

struct Builder {
    ObjectManager::Hierarchy& h;
    std::vector<float> p1;          // pointOne xyz per triangle (the sort keys)
    std::vector<float> lo, hi;      // per triangle: first minimum / maximum over its three points, per axis
    std::vector<KeyIdx> scratch;    // one slot per triangle; a node sorts its own [first, first + count) range
    std::atomic<int> pending{0};    // subtree tasks handed to the pool and not finished yet

    // nodes of the subtree of a CHILD holding c triangles (a child is split only while it holds > 8, :261-267)
    static uint32_t subtree(uint32_t c) { return c > 8 ? 1 + subtree(c / 2) + subtree(c - c / 2) : 1; }

    void bounds(uint32_t first, uint32_t count, vec3& mn, vec3& mx) const {       // calculateBoundingBoxes :205-221
        float a[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, b[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
        for (uint32_t k = 0; k < count; k++) {
            const uint32_t t = h.order[first + k];
            // glm::min(x,y) = y<x ? y : x keeps the earlier of equal values (+0 / -0): reducing a triangle's three
            // points first and then folding keeps the same element as the reference's point-by-point fold
            for (int c = 0; c < 3; c++) { const float v = lo[3 * t + c]; if (v < a[c]) a[c] = v; }
            for (int c = 0; c < 3; c++) { const float v = hi[3 * t + c]; if (b[c] < v) b[c] = v; }
        }
        mn = vec3(a[0], a[1], a[2]); mx = vec3(b[0], b[1], b[2]);
    }

    void split(int32_t node, int depth) {                                          // splitTrianglesForBox :225-272
        const vec3 mn = h.nodes[node].minBox, mx = h.nodes[node].maxBox;
        const uint32_t first = h.nodes[node].first, n = h.nodes[node].count;
        const float sx = std::fabs(mx.x - mn.x), sy = std::fabs(mx.y - mn.y), sz = std::fabs(mx.z - mn.z);
        const int axis = (sx > sy && sx > sz) ? 0 : ((sy > sx && sy > sz) ? 1 : 2);
        KeyIdx* b = scratch.data() + first;
        for (uint32_t k = 0; k < n; k++) { const uint32_t t = h.order[first + k]; b[k].key = p1[3 * t + axis]; b[k].idx = t; }
        if (n >= 16384) { ExactSort es; es.sort(b, b + n); }                       // std::sort's result, its partitions in parallel
        else if (n >= 256) { ExactSort es; es.parallel = false; es.sort(b, b + n); }    // the same, on this thread: the partition without branches
        else std::sort(b, b + n, [](const KeyIdx& x, const KeyIdx& y) { return x.key < y.key; });
        for (uint32_t k = 0; k < n; k++) h.order[first + k] = b[k].idx;
        const uint32_t nl = n / 2, nr = n - nl;
        const int32_t li = node + 1, ri = li + (int32_t)subtree(nl);               // DFS pre-order slots
        h.nodes[node].left = li; h.nodes[node].right = ri;
        Node& L = h.nodes[li]; Node& R = h.nodes[ri];
        L.first = first; L.count = nl; bounds(L.first, L.count, L.minBox, L.maxBox);
        R.first = first + nl; R.count = nr; bounds(R.first, R.count, R.minBox, R.maxBox);
        // the two halves touch disjoint ranges of order / scratch and disjoint node slots
        if (nl > 8 && nr > 8 && depth < PAR_DEPTH && n >= PAR_MIN && build_tasks_allowed()) {
            pending.fetch_add(1, std::memory_order_relaxed);
            BuildPool::get().submit([this, li, depth] { split(li, depth + 1); pending.fetch_sub(1, std::memory_order_release); });
            split(ri, depth + 1);
        } else {
            if (nl > 8) split(li, depth + 1);                                      // triangleSizeStop = 8 (:261)
            if (nr > 8) split(ri, depth + 1);
        }
    }
    static constexpr int PAR_DEPTH = 5;          // up to 32 concurrent subtrees
    static constexpr uint32_t PAR_MIN = 2048;    // not worth a task below this
};
} // namespace

void setHierarchyBuildTasks(bool on) { g_build_tasks.store(on, std::memory_order_relaxed); }

// test hook: the permutation the builder's parallel sort gives and the one std::sort gives (they must be the same)
void sort_keys_both_ways(const float* keys, uint32_t n, uint32_t* order_parallel, uint32_t* order_std) {
    std::vector<KeyIdx> a(n), b(n);
    for (uint32_t i = 0; i < n; i++) { a[i].key = b[i].key = keys[i]; a[i].idx = b[i].idx = i; }
    { ExactSort es; es.sort(a.data(), a.data() + n); }
    std::sort(b.begin(), b.end(), [](const KeyIdx& x, const KeyIdx& y) { return x.key < y.key; });
    for (uint32_t i = 0; i < n; i++) { order_parallel[i] = a[i].idx; order_std[i] = b[i].idx; }
}

void ObjectManager::createBoundingHierarchy(const std::string& objFilename) {
    std::vector<Triangle>& triangles = objTriangles[objFilename];
    const uint32_t n = (uint32_t)triangles.size();
    Hierarchy h;
    h.order.resize(n);
    Builder b{ h, {}, {}, {}, {}, {} };
    b.p1.resize(3 * (size_t)n); b.lo.resize(3 * (size_t)n); b.hi.resize(3 * (size_t)n); b.scratch.resize(n);
    h.points.resize(12 * (size_t)n);
    parallel_for(n, 8192, [&](size_t i0, size_t i1) {
        for (size_t i = i0; i < i1; i++) {
            h.order[i] = (uint32_t)i;
            const Triangle& t = triangles[i];
            const vec4* p[3] = { &t.pointOne, &t.pointTwo, &t.pointThree };
            for (int j = 0; j < 3; j++) for (int c = 0; c < 4; c++) h.points[12 * i + 4 * j + c] = (*p[j])[c];
            for (int c = 0; c < 3; c++) {
                b.p1[3 * i + c] = t.pointOne[c];
                float lo = (*p[0])[c], hi = (*p[0])[c];
                for (int j = 1; j < 3; j++) { const float v = (*p[j])[c]; if (v < lo) lo = v; if (hi < v) hi = v; }
                b.lo[3 * i + c] = lo; b.hi[3 * i + c] = hi;
            }
        }
    });
    const uint32_t nl = n / 2;
    h.nodes.resize(1 + (size_t)Builder::subtree(nl) + Builder::subtree(n - nl));   // the root is always split (:282)
    Node& root = h.nodes[0]; root.first = 0; root.count = n;
    b.bounds(0, n, root.minBox, root.maxBox);
    minBox[objFilename] = root.minBox; maxBox[objFilename] = root.maxBox;          // :280
    b.split(0, 0);
    while (b.pending.load(std::memory_order_acquire) > 0)                        // help with the queued subtrees
        if (!BuildPool::get().run_one()) std::this_thread::yield();
    h.triangles.resize(n);                                                       // the nodes' by-value triangles (Object.h:49)
    parallel_for(n, 4096, [&](size_t lo, size_t hi) { for (size_t k = lo; k < hi; k++) h.triangles[k] = triangles[h.order[k]]; });
    h.node_min.resize(3 * h.nodes.size()); h.node_max.resize(3 * h.nodes.size());
    for (size_t i = 0; i < h.nodes.size(); i++) for (int c = 0; c < 3; c++) { h.node_min[3 * i + c] = h.nodes[i].minBox[c]; h.node_max[3 * i + c] = h.nodes[i].maxBox[c]; }
    boundingVolumeHierarchy[objFilename] = std::move(h);
}

// ------------------------------------------------------------------------------------------------
// Flattener: ObjectManager -> flat scene of include/srt.h
// ------------------------------------------------------------------------------------------------
srt_scene_desc FlatScene::desc() const {
    srt_scene_desc d;
    std::memset(&d, 0, sizeof(d));
    d.n_objects = (uint32_t)obj_root.size(); d.n_nodes = (uint32_t)node_left.size();
    d.n_tris = (uint32_t)tri_obj.size(); d.n_textures = (uint32_t)tex_w.size();
    d.node_min = node_min.data(); d.node_max = node_max.data();
    d.node_left = node_left.data(); d.node_right = node_right.data(); d.node_first = node_first.data(); d.node_count = node_count.data();
    d.obj_root = obj_root.data();
    d.tri_points = tri_points.data(); d.tri_obj = tri_obj.data(); d.tri_tex = tri_tex.data();
    d.tri_texcoord = tri_texcoord.data(); d.tri_normals = tri_normals.data();
    d.obj_color = obj_color.data(); d.obj_material = obj_material.data();
    d.tex_rgb = tex_rgb.data(); d.tex_off = tex_off.data(); d.tex_w = tex_w.data(); d.tex_h = tex_h.data();
    return d;
}

FlatScene flattenScene(ObjectManager* om) {
    FlatScene f;
    std::unordered_map<std::string, int32_t> tex_id;
    // pass 1 (serial, per object): slots.  Hierarchy::nodes is already DFS pre-order and its leaves cover
    // Hierarchy::order left to right, so a triangle's flat index is its object's base + its position in order.
    struct Obj { const std::vector<Triangle>* tris; const ObjectManager::Hierarchy* h; int32_t node_base; int32_t tri_base; int32_t id; };
    std::vector<Obj> objs;
    size_t nt = 0, nn = 0;
    for (const auto& pair : om->objTriangles) {                 // the iteration order rayIntersection:409 uses
        const std::string& name = pair.first;
        auto hit = om->boundingVolumeHierarchy.find(name);
        if (hit == om->boundingVolumeHierarchy.end())
            throw std::runtime_error("object '" + name + "' has no bounding hierarchy (createBoundingHierarchy not called)");
        objs.push_back({ &hit->second.triangles, &hit->second, (int32_t)nn, (int32_t)nt, (int32_t)objs.size() });
        f.names.push_back(name);
        f.obj_root.push_back((uint32_t)nn);
        const vec3 c = om->objColors[name], m = om->objProperties[name];    // operator[]: default-inserts (0,0,0) like :368,439
        f.obj_color.insert(f.obj_color.end(), { c.x, c.y, c.z });
        f.obj_material.insert(f.obj_material.end(), { m.x, m.y, m.z });
        nn += hit->second.nodes.size(); nt += hit->second.triangles.size();
    }
    f.node_min.resize(3 * nn); f.node_max.resize(3 * nn); f.node_left.resize(nn); f.node_right.resize(nn);
    f.node_first.resize(nn); f.node_count.resize(nn);
    f.tri_points.resize(12 * nt); f.tri_texcoord.resize(6 * nt); f.tri_normals.resize(9 * nt); f.tri_obj.resize(nt); f.tri_tex.resize(nt);
    for (const Obj& o : objs) {
        const ObjectManager::Hierarchy& h = *o.h;
        const std::vector<Triangle>& tris = *o.tris;
        // pass 2 (serial): texture ids in first-use order over the flat triangle sequence
        for (size_t k = 0; k < tris.size(); k++) {
            const Triangle& t = tris[k];
            int32_t tid = -1;
            if (!t.textureName.empty()) {
                auto ti = om->textureData.find(t.textureName);
                if (ti != om->textureData.end()) {
                    auto known = tex_id.find(t.textureName);
                    if (known == tex_id.end()) {
                        tid = (int32_t)f.tex_w.size();
                        tex_id[t.textureName] = tid;
                        f.tex_names.push_back(t.textureName);
                        f.tex_off.push_back((uint64_t)f.tex_rgb.size());
                        f.tex_w.push_back((uint32_t)ti->second.dim.x); f.tex_h.push_back((uint32_t)ti->second.dim.y);
                        f.tex_rgb.insert(f.tex_rgb.end(), ti->second.rgb.begin(), ti->second.rgb.end());
                    } else tid = known->second;
                }
            }
            f.tri_tex[o.tri_base + k] = tid;
        }
        // pass 3 (parallel): node and triangle records
        parallel_for(h.nodes.size(), 4096, [&](size_t b, size_t e) {
            for (size_t i = b; i < e; i++) {
                const Node& n = h.nodes[i];
                const size_t g = (size_t)o.node_base + i;
                for (int a = 0; a < 3; a++) { f.node_min[3 * g + a] = n.minBox[a]; f.node_max[3 * g + a] = n.maxBox[a]; }
                const bool leaf = n.left < 0 && n.right < 0;
                f.node_left[g] = leaf ? -1 : o.node_base + n.left;
                f.node_right[g] = leaf ? -1 : o.node_base + n.right;
                f.node_first[g] = leaf ? o.tri_base + (int32_t)n.first : -1;
                f.node_count[g] = leaf ? (int32_t)n.count : 0;
            }
        });
        parallel_for(tris.size(), 4096, [&](size_t b, size_t e) {
            for (size_t k = b; k < e; k++) {
                const Triangle& t = tris[k];
                const size_t g = (size_t)o.tri_base + k;
                const vec4* p[3] = { &t.pointOne, &t.pointTwo, &t.pointThree };
                for (int j = 0; j < 3; j++) for (int c = 0; c < 4; c++) f.tri_points[12 * g + 4 * j + c] = (*p[j])[c];
                const vec2* tc[3] = { &t.colorOneCoordinate, &t.colorTwoCoordinate, &t.colorThreeCoordinate };
                for (int j = 0; j < 3; j++) { f.tri_texcoord[6 * g + 2 * j] = tc[j]->x; f.tri_texcoord[6 * g + 2 * j + 1] = tc[j]->y; }
                const vec3* nm[3] = { &t.normalOne, &t.normalTwo, &t.normalThree };
                for (int j = 0; j < 3; j++) for (int c = 0; c < 3; c++) f.tri_normals[9 * g + 3 * j + c] = (*nm[j])[c];
                f.tri_obj[g] = o.id;
            }
        });
    }
    return f;
}

// ------------------------------------------------------------------------------------------------
// Drop-in entry point and image output
// ------------------------------------------------------------------------------------------------
// hit pixels of an 8-bit frame in the reference's emission order: x outer, y inner (:511-513).  Count per column, then fill the
// slots in parallel.  Black = "not emitted" (:518).
// `out` may come pre-sized (Renderer::collect sizes it by the previous frame's count while the GPU is still rendering: resize
// value-initialises, 20 bytes per emitted pixel of zero fill that then costs nothing on the critical path).
static void resize_both(ImageData& out, size_t n) {
    parallel_for(2, 1, [&](size_t b, size_t e) { for (size_t k = b; k < e; k++) { if (k == 0) out.imagePoints.resize(n); else out.imageColors.resize(n); } });
}
static ImageData image_data_from_rgb8(const uint8_t* rgb8, uint32_t W, uint32_t H, ImageData&& pre = ImageData()) {
    ImageData out = std::move(pre);
    std::vector<size_t> col(W + 1, 0);
    parallel_for(W, 64, [&](size_t x0, size_t x1) {
        for (size_t x = x0; x < x1; x++) {
            size_t n = 0;
            for (uint32_t y = 0; y < H; y++) { const uint8_t* c = &rgb8[((size_t)y * W + x) * 3]; n += (c[0] | c[1] | c[2]) != 0; }
            col[x + 1] = n;
        }
    });
    for (uint32_t x = 0; x < W; x++) col[x + 1] += col[x];
    if (out.imagePoints.size() < col[W] || out.imageColors.size() < col[W]) resize_both(out, col[W]);      // first frame, or more pixels than guessed
    parallel_for(W, 64, [&](size_t x0, size_t x1) {
        for (size_t x = x0; x < x1; x++) {
            size_t k = col[x];
            for (uint32_t y = 0; y < H; y++) {
                const uint8_t* c = &rgb8[((size_t)y * W + x) * 3];
                if (c[0] | c[1] | c[2]) {
                    out.imagePoints[k] = vec2((float)x, (float)y);
                    out.imageColors[k] = vec3((float)c[0], (float)c[1], (float)c[2]);
                    k++;
                }
            }
        }
    });
    if (out.imagePoints.size() != col[W]) { out.imagePoints.resize(col[W]); out.imageColors.resize(col[W]); }      // shrinking: no fill
    return out;
}

Renderer::Renderer(int device) : device_(device) {}
Renderer::~Renderer() {
    if (scene_) srt_scene_destroy(scene_);
    if (rgb8_) srt_host_free(rgb8_);
}

// A sample of the per-triangle attributes the device keeps in source order (texture, texel coordinates, normals): every 61st triangle
// of every hierarchy, and the identity of the texture images.  Catches another mesh or another material under the same name and counts;
// an edit of single triangles between two frames of otherwise the same scene is NOT seen -- setFastPath(false) for such callers.
static uint64_t attribute_sample(ObjectManager* om) {
    uint64_t h = 0xcbf29ce484222325ull;
    auto mix = [&](const void* p, size_t n) { const unsigned char* c = (const unsigned char*)p; for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 0x100000001b3ull; } };
    for (const auto& pair : om->objTriangles) {
        auto hit = om->boundingVolumeHierarchy.find(pair.first);
        if (hit == om->boundingVolumeHierarchy.end()) continue;
        const std::vector<Triangle>& tr = hit->second.triangles;
        for (size_t i = 0; i < tr.size(); i += 61) {
            const Triangle& t = tr[i];
            mix(&t.normalOne, 3 * sizeof(vec3)); mix(&t.colorOneCoordinate, 3 * sizeof(vec2)); mix(t.textureName.data(), t.textureName.size());
        }
    }
    for (const auto& tx : om->textureData) {
        mix(tx.first.data(), tx.first.size()); mix(&tx.second.dim, sizeof(tx.second.dim));
        const std::vector<unsigned char>& px = tx.second.rgb;
        const size_t n = px.size();
        mix(&n, sizeof(n));
        for (size_t i = 0; i < n; i += 4099) mix(&px[i], 1);
    }
    return h;
}

// The short way (include/srt.h, f1 device half): the hierarchies' compact arrays straight to srt_scene_update_frame.  false = the
// resident scene is not this ObjectManager's (first frame, other objects or counts, other attributes): the caller takes the long way.
bool Renderer::upload_frame(ObjectManager* om) {
    if (!scene_ || !fast_path_ || sig_names_.size() != om->objTriangles.size()) return false;
    const size_t nO = sig_names_.size();
    std::vector<uint32_t> n_tris(nO), n_nodes(nO);
    std::vector<const float*> pts(nO), bmin(nO), bmax(nO);
    std::vector<const uint32_t*> ord(nO);
    std::vector<float> color(3 * nO), mat(3 * nO);
    size_t k = 0;
    for (const auto& pair : om->objTriangles) {                 // the iteration order rayIntersection:409 uses (and flattenScene)
        if (pair.first != sig_names_[k]) return false;
        auto hit = om->boundingVolumeHierarchy.find(pair.first);
        if (hit == om->boundingVolumeHierarchy.end()) return false;      // (the long way reports it)
        const ObjectManager::Hierarchy& h = hit->second;
        if (h.order.size() != sig_tris_[k] || h.nodes.size() != sig_nodes_[k] || h.points.size() != 12 * h.order.size()) return false;
        n_tris[k] = sig_tris_[k]; n_nodes[k] = sig_nodes_[k];
        pts[k] = h.points.data(); ord[k] = h.order.data(); bmin[k] = h.node_min.data(); bmax[k] = h.node_max.data();
        const vec3 c = om->objColors[pair.first], m = om->objProperties[pair.first];
        for (int a = 0; a < 3; a++) { color[3 * k + a] = c[a]; mat[3 * k + a] = m[a]; }
        k++;
    }
    if (attribute_sample(om) != sig_attr_) return false;
    srt_frame_geometry g;
    std::memset(&g, 0, sizeof(g));
    g.n_objects = (uint32_t)nO; g.obj_n_tris = n_tris.data(); g.obj_n_nodes = n_nodes.data();
    g.obj_points = pts.data(); g.obj_order = ord.data(); g.obj_node_min = bmin.data(); g.obj_node_max = bmax.data();
    g.obj_color = color.data(); g.obj_material = mat.data();
    const int rc = srt_scene_update_frame(scene_, &g, nullptr);
    if (rc == SRT_ERR_LAYOUT) return false;
    if (rc != SRT_OK) throw std::runtime_error(std::string("srt_scene_update_frame: ") + srt_strerror(rc));
    fast_frames_++;
    return true;
}

// the ObjectManager's current state into the device scene: the device half of the rebuild when the resident scene is this one with
// other positions (upload_frame), else flattened -- in place when the counts allow it, else a new scene
void Renderer::upload(ObjectManager* objManager) {
    if (upload_frame(objManager)) return;
    FlatScene flat = flattenScene(objManager);
    srt_scene_desc d = flat.desc();
    int rc = scene_ ? srt_scene_update(scene_, &d, nullptr) : SRT_ERR_LAYOUT;
    if (rc == SRT_ERR_LAYOUT) {
        if (scene_) { srt_scene_destroy(scene_); scene_ = nullptr; }
        rc = srt_scene_create(device_, &d, &scene_);
        if (rc != SRT_OK) { scene_ = nullptr; throw std::runtime_error(std::string("srt_scene_create: ") + srt_strerror(rc)); }
    } else if (rc != SRT_OK) throw std::runtime_error(std::string("srt_scene_update: ") + srt_strerror(rc));
    // what is resident now, and its per-triangle attributes in SOURCE order for the frames that follow (the hierarchies' by-value
    // copies are in visit order: position i holds source triangle order[i])
    sig_names_ = flat.names; sig_tris_.clear(); sig_nodes_.clear();
    const size_t nt = flat.tri_obj.size();
    std::vector<float> tc(6 * nt), nrm(9 * nt);
    std::vector<int32_t> tex(nt, -1);
    bool any_tex = false;
    size_t base = 0;
    for (const std::string& name : flat.names) {
        const ObjectManager::Hierarchy& h = objManager->boundingVolumeHierarchy.at(name);
        sig_tris_.push_back((uint32_t)h.order.size()); sig_nodes_.push_back((uint32_t)h.nodes.size());
        for (size_t i = 0; i < h.order.size(); i++) {
            const size_t src = base + h.order[i], vis = base + i;
            for (int c = 0; c < 6; c++) tc[6 * src + c] = flat.tri_texcoord[6 * vis + c];
            for (int c = 0; c < 9; c++) nrm[9 * src + c] = flat.tri_normals[9 * vis + c];
            tex[src] = flat.tri_tex[vis];
            any_tex = any_tex || tex[src] >= 0;
        }
        base += h.order.size();
    }
    sig_attr_ = attribute_sample(objManager);
    rc = srt_scene_set_source(scene_, tc.data(), nrm.data(), any_tex ? tex.data() : nullptr);
    if (rc != SRT_OK) { sig_names_.clear(); }              // no short way for this scene (it stays correct: the long way every frame)
}

void Renderer::enqueue(const vec2& imageSize, const vec4& lightPos, int lightAmount, const mat4* viewMatrix) {
    const uint32_t W = (uint32_t)imageSize.x, H = (uint32_t)imageSize.y;
    const size_t bytes = (size_t)W * H * 3;
    if (bytes > rgb8_bytes_) {
        if (rgb8_) srt_host_free(rgb8_);
        rgb8_ = (uint8_t*)srt_host_alloc(bytes);
        rgb8_bytes_ = rgb8_ ? bytes : 0;
        if (!rgb8_) throw std::runtime_error("srt_host_alloc failed");
    }
    srt_params p;
    srt_params_default(&p, W, H);
    std::vector<float> lights((size_t)lightAmount * 3);
    const float base[3] = { lightPos.x, lightPos.y, lightPos.z };       // vec4 -> const vec3& drops w (:517 -> :405)
    srt_light_staircase(base, (uint32_t)lightAmount, lights.data());
    p.n_lights = (uint32_t)lightAmount; p.light_pos = lights.data();     // copied by the call
    p.background[0] = p.background[1] = p.background[2] = 0;             // ask for black so that ImageData can skip it (:518)
    float m16[16];
    if (viewMatrix) { for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) m16[c * 4 + r] = (*viewMatrix)[c][r]; p.ray_matrix = m16; }
    const int rc = srt_render_async(scene_, &p, nullptr, nullptr, nullptr, rgb8_);
    if (rc != SRT_OK) throw std::runtime_error(std::string("srt_render_async: ") + srt_strerror(rc));
    W_ = W; H_ = H; pending_ = true;
}

void Renderer::submit(const vec2& imageSize, const vec4& lightPos, ObjectManager* objManager, int lightAmount) {
    if (pending_) throw std::runtime_error("Renderer::submit: collect() the previous frame first");
    upload(objManager);
    enqueue(imageSize, lightPos, lightAmount, nullptr);
}

ImageData Renderer::collect() {
    if (!pending_) throw std::runtime_error("Renderer::collect: nothing submitted");
    pending_ = false;
    ImageData pre;
    if (last_count_) resize_both(pre, last_count_ + last_count_ / 16 + 64);      // while the GPU renders: the zero fill of the result vectors
    const int rc = srt_sync(scene_, nullptr);
    if (rc != SRT_OK) throw std::runtime_error(std::string("srt_sync: ") + srt_strerror(rc));
    ImageData out = image_data_from_rgb8(rgb8_, W_, H_, std::move(pre));
    last_count_ = out.imagePoints.size();
    return out;
}

ImageData Renderer::render(const vec2& imageSize, const vec4& lightPos, ObjectManager* objManager, int lightAmount) {
    submit(imageSize, lightPos, objManager, lightAmount);
    return collect();
}

ImageData Renderer::renderFromCamera(const vec2& imageSize, const vec4& lightPosWorld, const mat4& viewMatrix, ObjectManager* objManager,
                                     int lightAmount, bool sceneChanged) {
    if (pending_) throw std::runtime_error("Renderer::renderFromCamera: collect() the previous frame first");
    if (!scene_ || sceneChanged) upload(objManager);
    enqueue(imageSize, lightPosWorld, lightAmount, &viewMatrix);
    return collect();
}

MultiRenderer::MultiRenderer(const std::vector<int>& devices, uint32_t blockRows) : blockRows_(blockRows ? blockRows : 8) {
    if (devices.empty()) throw std::runtime_error("MultiRenderer: no devices");
    for (int d : devices) { Part p; p.device = d; parts_.push_back(p); }
}
MultiRenderer::~MultiRenderer() {
    for (Part& p : parts_) { if (p.scene) srt_scene_destroy(p.scene); if (p.rgb8) srt_host_free(p.rgb8); }
}

ImageData MultiRenderer::render(const vec2& imageSize, const vec4& lightPos, ObjectManager* objManager, int lightAmount) {
    const uint32_t W = (uint32_t)imageSize.x, H = (uint32_t)imageSize.y, N = (uint32_t)parts_.size();
    FlatScene flat = flattenScene(objManager);                          // once; every device gets the same records
    srt_scene_desc d = flat.desc();
    std::vector<float> lights((size_t)lightAmount * 3);
    const float base[3] = { lightPos.x, lightPos.y, lightPos.z };       // vec4 -> const vec3& drops w (:517 -> :405)
    srt_light_staircase(base, (uint32_t)lightAmount, lights.data());
    auto params_of = [&](uint32_t k) {
        srt_params p;
        srt_params_default(&p, W, H);
        p.n_lights = (uint32_t)lightAmount; p.light_pos = lights.data();
        p.background[0] = p.background[1] = p.background[2] = 0;         // black = not emitted (:518)
        p.block_rows = blockRows_; p.block_first = k; p.block_stride = N;
        return p;
    };
    // enqueue everywhere first (upload + render + read-back on each scene's own stream), then wait
    for (uint32_t k = 0; k < N; k++) {
        Part& pt = parts_[k];
        int rc = pt.scene ? srt_scene_update(pt.scene, &d, nullptr) : SRT_ERR_LAYOUT;
        if (rc == SRT_ERR_LAYOUT) {
            if (pt.scene) { srt_scene_destroy(pt.scene); pt.scene = nullptr; }
            rc = srt_scene_create(pt.device, &d, &pt.scene);
            if (rc != SRT_OK) { pt.scene = nullptr; throw std::runtime_error(std::string("srt_scene_create: ") + srt_strerror(rc)); }
        } else if (rc != SRT_OK) throw std::runtime_error(std::string("srt_scene_update: ") + srt_strerror(rc));
        const srt_params p = params_of(k);
        pt.rows = srt_rows_owned(&p);
        const size_t bytes = (size_t)pt.rows * W * 3;
        if (bytes > pt.bytes) {
            if (pt.rgb8) srt_host_free(pt.rgb8);
            pt.rgb8 = (uint8_t*)srt_host_alloc(bytes ? bytes : 1);
            pt.bytes = pt.rgb8 ? bytes : 0;
            if (!pt.rgb8) throw std::runtime_error("srt_host_alloc failed");
        }
        rc = srt_render_async(pt.scene, &p, nullptr, nullptr, nullptr, pt.rgb8);
        if (rc != SRT_OK) throw std::runtime_error(std::string("srt_render_async: ") + srt_strerror(rc));
    }
    frame_.assign((size_t)W * H * 3, 0);
    for (uint32_t k = 0; k < N; k++) {
        Part& pt = parts_[k];
        const int rc = srt_sync(pt.scene, nullptr);
        if (rc != SRT_OK) throw std::runtime_error(std::string("srt_sync: ") + srt_strerror(rc));
        for (uint32_t r = 0; r < pt.rows; r++) {                        // local row -> image row (include/srt.h srt_params)
            const uint32_t y = ((r / blockRows_) * N + k) * blockRows_ + r % blockRows_;
            std::memcpy(&frame_[(size_t)y * W * 3], pt.rgb8 + (size_t)r * W * 3, (size_t)W * 3);
        }
    }
    return image_data_from_rgb8(frame_.data(), W, H);
}

ImageData sendRaysAndIntersectPointsColors(const vec2& imageSize, const vec4& lightPos, ObjectManager* objManager,
                                           int lightAmount, int device) {
    // one Renderer per thread and device, kept for the life of the thread (leaked on purpose: no HIP calls from thread-exit
    // destructors): the second frame of an orbit re-uses the first frame's device scene and pinned buffers
    static thread_local std::unordered_map<int, Renderer*> renderers;
    Renderer*& r = renderers[device];
    if (!r) r = new Renderer(device);
    return r->render(imageSize, lightPos, objManager, lightAmount);
}

void writeBmp(const std::string& path, uint32_t W, uint32_t H, const uint8_t* rgb) {
    const uint32_t stride = (W * 3 + 3) & ~3u, size = 54 + stride * H;
    std::vector<uint8_t> f(size, 0);
    auto le32 = [&](size_t o, uint32_t v) { f[o] = v & 255; f[o + 1] = (v >> 8) & 255; f[o + 2] = (v >> 16) & 255; f[o + 3] = (v >> 24) & 255; };
    f[0] = 'B'; f[1] = 'M'; le32(2, size); le32(10, 54); le32(14, 40); le32(18, W); le32(22, H); f[26] = 1; f[28] = 24; le32(34, stride * H);
    for (uint32_t y = 0; y < H; y++) {
        uint8_t* row = &f[54 + (size_t)stride * (H - 1 - y)];
        for (uint32_t x = 0; x < W; x++) { const uint8_t* c = &rgb[((size_t)y * W + x) * 3]; row[x * 3] = c[2]; row[x * 3 + 1] = c[1]; row[x * 3 + 2] = c[0]; }
    }
    std::ofstream o(path, std::ios::binary);
    if (!o) throw std::runtime_error("cannot write " + path);
    o.write((const char*)f.data(), (std::streamsize)f.size());
}

void drawImage(const vec2& imgSize, const std::vector<vec2>& imagePoints, const std::vector<vec3>& imageColors,
               const int& angleDegree, const bool& saveImage, const std::string& directory) {
    const uint32_t W = (uint32_t)imgSize.x, H = (uint32_t)imgSize.y;
    std::vector<uint8_t> img((size_t)W * H * 3, 0);                                     // :463-464
    for (size_t i = 0; i < imagePoints.size(); i++) {                                   // :468-474
        const uint32_t x = (uint32_t)imagePoints[i].x, y = (uint32_t)imagePoints[i].y;
        if (x >= W || y >= H) continue;
        uint8_t* c = &img[((size_t)y * W + x) * 3];
        c[0] = (uint8_t)imageColors[i].x; c[1] = (uint8_t)imageColors[i].y; c[2] = (uint8_t)imageColors[i].z;
    }
    for (size_t i = 0; i < (size_t)W * H; i++)                                          // :476-487
        if (!img[i * 3] && !img[i * 3 + 1] && !img[i * 3 + 2]) { img[i * 3] = 173; img[i * 3 + 1] = 216; img[i * 3 + 2] = 230; }
    if (saveImage) writeBmp(directory + "/output" + std::to_string(static_cast<int>(angleDegree)) + ".bmp", W, H, img.data());   // :488-494
}

} // namespace srt_host
