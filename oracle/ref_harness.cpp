// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin extern "C" shim around the *unmodified* reference sources, compiled from where they lie
// under /root/reference (never copied into this repo) by oracle/Makefile into
// oracle/_ref/libsrt_ref.so.  It exists so that
//   * tools/make_golden.py can generate the committed fixtures under tests/golden/ from the
//     reference's own functions, and
//   * tests can cross-check the C restatement (oracle/srt_oracle.c) against the real thing on
//     random inputs when oracle/_ref is present (this container only; the GPU box has the prebuilt
//     .so but no /root/reference assets, so only array-fed entry points work there).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// The reference has main() in the same translation unit as the hot path
// (simple_raytracer.cpp:530); it is renamed so that the TU can be included.
#define main srt_reference_main_unused
#include "simple_raytracer.cpp"   // found via -I/root/reference (see oracle/Makefile)
#undef main

#include <cstdint>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>

namespace {

struct FlatCounts { uint32_t n_objects, n_nodes, n_tris; };

void count_nodes(const Node* n, uint32_t& nodes, uint32_t& tris) {
    nodes++;
    if (!n->left && !n->right) { tris += (uint32_t)n->triangles.size(); return; }
    count_nodes(n->left, nodes, tris);
    count_nodes(n->right, nodes, tris);
}

// DFS left-first, exactly the order boundingBoxIntersection (simple_raytracer.cpp:296-317) visits.
struct FlatOut {
    float* node_min; float* node_max; int32_t* node_left; int32_t* node_right;
    int32_t* node_first; int32_t* node_count;
    float* tri_points;      // n_tris x 3 x 4 (raw homogeneous points, leaf order)
    float* tri_texcoord;    // n_tris x 6
    float* tri_color;       // n_tris x 3
    int32_t* tri_obj;       // n_tris
    int32_t* tri_has_tex;   // n_tris
    uint32_t node_cursor, tri_cursor;
};

int32_t flatten(Node* n, FlatOut& o, int32_t obj, bool tag_ids) {
    int32_t me = (int32_t)o.node_cursor++;
    for (int a = 0; a < 3; a++) { o.node_min[me*3+a] = n->minBox[a]; o.node_max[me*3+a] = n->maxBox[a]; }
    if (!n->left && !n->right) {
        o.node_left[me] = -1; o.node_right[me] = -1;
        o.node_first[me] = (int32_t)o.tri_cursor; o.node_count[me] = (int32_t)n->triangles.size();
        for (Triangle& t : n->triangles) {
            uint32_t id = o.tri_cursor++;
            const glm::vec4* p[3] = { &t.pointOne, &t.pointTwo, &t.pointThree };
            for (int k = 0; k < 3; k++) for (int c = 0; c < 4; c++) o.tri_points[(id*3+k)*4+c] = (*p[k])[c];
            const glm::vec2* tc[3] = { &t.colorOneCoordinate, &t.colorTwoCoordinate, &t.colorThreeCoordinate };
            for (int k = 0; k < 3; k++) { o.tri_texcoord[id*6+k*2] = tc[k]->x; o.tri_texcoord[id*6+k*2+1] = tc[k]->y; }
            for (int c = 0; c < 3; c++) o.tri_color[id*3+c] = t.color[c];
            o.tri_obj[id] = obj;
            o.tri_has_tex[id] = t.textureName.empty() ? 0 : 1;
            // Tag the leaf copy with its canonical id in a field the live hot path never reads
            // (vertex normals: the interpolateNormal call is commented out, simple_raytracer.cpp:162).
            if (tag_ids) t.normalOne = glm::vec3((float)(id & 0xFFFF), (float)(id >> 16), -7.0f);
        }
        return me;
    }
    o.node_first[me] = -1; o.node_count[me] = 0;
    o.node_left[me]  = flatten(n->left, o, obj, tag_ids);
    o.node_right[me] = flatten(n->right, o, obj, tag_ids);
    return me;
}

inline int32_t tagged_id(const Triangle& t) {
    if (t.normalOne.z != -7.0f) return -2;
    return (int32_t)((uint32_t)t.normalOne.x | ((uint32_t)t.normalOne.y << 16));
}

} // namespace

extern "C" {

// ---- scene construction: straight calls into the reference's ObjectManager (Object.h:59-89) ----
void* ref_om_new() { return new ObjectManager(); }
void  ref_om_free(void* om) { delete (ObjectManager*)om; }   // Node trees leak, as in the reference

void ref_om_load_obj(void* om, const char* name) { ((ObjectManager*)om)->loadObjFile(name); }

// Object fed from arrays: default colour/material exactly as loadObjFile sets them (Object.cpp:29-34).
void ref_om_add_object(void* om_, const char* name, uint32_t n, const float* points /* n x 3 x 4 */) {
    ObjectManager* om = (ObjectManager*)om_;
    om->objColors[name] = glm::vec3(1.f, 0.f, 0.f);
    om->objProperties[name] = glm::vec3(0.2f, 0.5f, 15.0f);
    std::vector<Triangle> tris(n);
    for (uint32_t i = 0; i < n; i++) {
        const float* p = points + (size_t)i * 12;
        tris[i].pointOne   = glm::vec4(p[0], p[1], p[2],  p[3]);
        tris[i].pointTwo   = glm::vec4(p[4], p[5], p[6],  p[7]);
        tris[i].pointThree = glm::vec4(p[8], p[9], p[10], p[11]);
        tris[i].color = glm::vec3(1.f, 1.f, 1.f);              // loader defaults, Object.cpp:81-84
        tris[i].colorOneCoordinate = tris[i].colorTwoCoordinate = tris[i].colorThreeCoordinate = glm::vec2(0.f, 0.f);
    }
    om->objTriangles[name] = tris;
}

// Array-fed textured object: what loadObjFile leaves behind for a textured mesh (Object.cpp:98-161): integer texel
// coordinates per vertex, textureName on every triangle, the decoded texture in textureData / textureDimensions.
void ref_om_add_texture(void* om_, const char* texname, int32_t w, int32_t h, const uint8_t* rgb) {
    ObjectManager* om = (ObjectManager*)om_;
    unsigned char* data = (unsigned char*)malloc((size_t)w * h * 3);      // stbi_load mallocs too; never freed, as in the reference
    std::memcpy(data, rgb, (size_t)w * h * 3);
    om->textureData[texname] = data;
    om->textureDimensions[texname] = glm::ivec2(w, h);
}
void ref_om_add_textured_object(void* om_, const char* name, uint32_t n, const float* points, const float* texcoord /* n x 6 */, const char* texname) {
    ObjectManager* om = (ObjectManager*)om_;
    ref_om_add_object(om_, name, n, points);
    std::vector<Triangle>& tris = om->objTriangles[name];
    for (uint32_t i = 0; i < n; i++) {
        tris[i].colorOneCoordinate = glm::vec2(texcoord[i*6], texcoord[i*6+1]);
        tris[i].colorTwoCoordinate = glm::vec2(texcoord[i*6+2], texcoord[i*6+3]);
        tris[i].colorThreeCoordinate = glm::vec2(texcoord[i*6+4], texcoord[i*6+5]);
        tris[i].textureName = texname;
    }
}

// Array-fed object with several textures (house.obj: six materials with their own map_Kd): per triangle an index into
// `names` (newline-separated; -1 = untextured triangle, textureName stays empty as the loader leaves it when the material has no
// diffuse map, Object.cpp:98-102,149).  The textures themselves are added with ref_om_add_texture.
void ref_om_add_multi_textured_object(void* om_, const char* name, uint32_t n, const float* points, const float* texcoord /* n x 6 */,
                                      const int32_t* tri_tex /* n */, const char* names) {
    ObjectManager* om = (ObjectManager*)om_;
    ref_om_add_object(om_, name, n, points);
    std::vector<std::string> list;
    for (const char* p = names; *p;) { const char* e = std::strchr(p, '\n'); if (!e) e = p + std::strlen(p); list.emplace_back(p, e); p = *e ? e + 1 : e; }
    std::vector<Triangle>& tris = om->objTriangles[name];
    for (uint32_t i = 0; i < n; i++) {
        if (tri_tex[i] < 0) continue;
        tris[i].colorOneCoordinate = glm::vec2(texcoord[i*6], texcoord[i*6+1]);
        tris[i].colorTwoCoordinate = glm::vec2(texcoord[i*6+2], texcoord[i*6+3]);
        tris[i].colorThreeCoordinate = glm::vec2(texcoord[i*6+4], texcoord[i*6+5]);
        tris[i].textureName = list.at((size_t)tri_tex[i]);
    }
}

// Clone as main() does it (simple_raytracer.cpp:565,597,644): triangles only; colour/material are
// whatever unordered_map::operator[] default-inserts later unless set explicitly.
void ref_om_clone(void* om_, const char* src, const char* dst) {
    ObjectManager* om = (ObjectManager*)om_;
    om->objTriangles[dst] = om->getTriangles(src);
}
void ref_om_set_color(void* om, const char* name, float r, float g, float b) { ((ObjectManager*)om)->setColor(name, glm::vec3(r, g, b)); }
void ref_om_set_props(void* om, const char* name, float ka, float ks, float sh) { ((ObjectManager*)om)->objProperties[name] = glm::vec3(ka, ks, sh); }
void ref_om_transform(void* om, const char* name, const float* m /* column-major 16 */) {
    glm::mat4 M; std::memcpy(&M[0][0], m, 64);
    ((ObjectManager*)om)->transformTriangles(name, M);
}
void ref_om_build_bvh(void* om, const char* name) { ((ObjectManager*)om)->createBoundingHierarchy(name); }
uint32_t ref_om_num_tris(void* om, const char* name) { return (uint32_t)((ObjectManager*)om)->getTriangles(name).size(); }
void ref_om_get_points(void* om, const char* name, float* out /* n x 3 x 4 */) {
    const std::vector<Triangle>& v = ((ObjectManager*)om)->getTriangles(name);
    for (size_t i = 0; i < v.size(); i++) {
        const glm::vec4* p[3] = { &v[i].pointOne, &v[i].pointTwo, &v[i].pointThree };
        for (int k = 0; k < 3; k++) for (int c = 0; c < 4; c++) out[(i*3+k)*4+c] = (*p[k])[c];
    }
}
// per-triangle loader outputs (Object.cpp:113-161): integer texel coords, vertex-0 colour, texture flag
void ref_om_get_tri_attrs(void* om, const char* name, float* texcoord /* n x 6 */, float* color /* n x 3 */, int32_t* has_tex, float* normals /* n x 9 */) {
    const std::vector<Triangle>& v = ((ObjectManager*)om)->getTriangles(name);
    for (size_t i = 0; i < v.size(); i++) {
        const glm::vec2* tc[3] = { &v[i].colorOneCoordinate, &v[i].colorTwoCoordinate, &v[i].colorThreeCoordinate };
        for (int k = 0; k < 3; k++) { texcoord[i*6+k*2] = tc[k]->x; texcoord[i*6+k*2+1] = tc[k]->y; }
        for (int c = 0; c < 3; c++) color[i*3+c] = v[i].color[c];
        has_tex[i] = v[i].textureName.empty() ? 0 : 1;
        const glm::vec3* nn[3] = { &v[i].normalOne, &v[i].normalTwo, &v[i].normalThree };
        for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) normals[i*9+k*3+c] = (*nn[k])[c];
    }
}
// texture name of triangle i of an object ("" if none); returns length
uint32_t ref_om_tri_texture_name(void* om, const char* name, uint32_t i, char* buf, uint32_t cap) {
    const std::string& s = ((ObjectManager*)om)->getTriangles(name)[i].textureName;
    if (cap) { std::strncpy(buf, s.c_str(), cap - 1); buf[cap - 1] = 0; }
    return (uint32_t)s.size();
}
// loaded texture lookup: returns 1 and fills w,h and (if rgb != null) copies w*h*3 bytes
int ref_om_texture(void* om_, const char* texname, int32_t* w, int32_t* h, uint8_t* rgb) {
    ObjectManager* om = (ObjectManager*)om_;
    auto it = om->textureData.find(texname);
    if (it == om->textureData.end()) return 0;
    glm::ivec2 d = om->textureDimensions[texname];
    *w = d.x; *h = d.y;
    if (rgb) std::memcpy(rgb, it->second, (size_t)d.x * d.y * 3);
    return 1;
}

// The reference's image decoder as its loader calls it (Object.cpp:57: stbi_load(path, &w, &h, &ch, 3); the
// implementation is compiled into the reference's Object.cpp).  rgb == null: dimensions only.  Returns 1 on success.
extern "C" unsigned char* stbi_load(char const* filename, int* x, int* y, int* comp, int req_comp);
extern "C" void stbi_image_free(void* p);
int ref_stbi_load(const char* path, int32_t* w, int32_t* h, uint8_t* rgb) {
    int x = 0, y = 0, ch = 0;
    unsigned char* d = stbi_load(path, &x, &y, &ch, 3);
    if (!d) return 0;
    *w = x; *h = y;
    if (rgb) std::memcpy(rgb, d, (size_t)x * y * 3);
    stbi_image_free(d);
    return 1;
}

// Object names in objTriangles iteration order (the order rayIntersection:409 visits them).
uint32_t ref_om_object_order(void* om_, char* buf, uint32_t cap) {
    ObjectManager* om = (ObjectManager*)om_;
    std::string s; uint32_t n = 0;
    for (const auto& p : om->objTriangles) { s += p.first; s += '\n'; n++; }
    if (cap) { std::strncpy(buf, s.c_str(), cap - 1); buf[cap - 1] = 0; }
    return n;
}

void ref_om_counts(void* om_, uint32_t* n_objects, uint32_t* n_nodes, uint32_t* n_tris) {
    ObjectManager* om = (ObjectManager*)om_;
    uint32_t o = 0, nn = 0, nt = 0;
    for (const auto& p : om->objTriangles) { o++; count_nodes(om->boundingVolumeHierarchy[p.first], nn, nt); }
    *n_objects = o; *n_nodes = nn; *n_tris = nt;
}

// Export the reference's own Node* trees as flat arrays (and tag leaf triangles with canonical ids).
void ref_om_export(void* om_, uint32_t* obj_root /* n_objects */, float* obj_color, float* obj_props,
                   float* node_min, float* node_max, int32_t* node_left, int32_t* node_right,
                   int32_t* node_first, int32_t* node_count,
                   float* tri_points, float* tri_texcoord, float* tri_color, int32_t* tri_obj, int32_t* tri_has_tex) {
    ObjectManager* om = (ObjectManager*)om_;
    FlatOut o{ node_min, node_max, node_left, node_right, node_first, node_count,
               tri_points, tri_texcoord, tri_color, tri_obj, tri_has_tex, 0, 0 };
    int32_t k = 0;
    for (const auto& p : om->objTriangles) {
        obj_root[k] = o.node_cursor;
        // same accessors the hot path uses: operator[] (default-inserting) for colour and material
        glm::vec3 c = om->objColors[p.first], m = om->objProperties[p.first];
        for (int a = 0; a < 3; a++) { obj_color[k*3+a] = c[a]; obj_props[k*3+a] = m[a]; }
        flatten(om->boundingVolumeHierarchy[p.first], o, k, true);
        k++;
    }
}

// Texture ids of the exported triangles (same walk as ref_om_export): index into the list of distinct LOADED texture names in
// order of first use over the flat triangle sequence, -1 for untextured triangles (and for a textureName whose image failed to
// load -- the reference would null-deref on such a hit, simple_raytracer.cpp:354-358; scenes used for goldens contain none).
// Writes the names newline-separated; returns how many.
namespace {
void tex_walk(const Node* n, ObjectManager* om, std::vector<std::string>& names, int32_t* tri_tex, uint32_t& cursor) {
    if (!n->left && !n->right) {
        for (const Triangle& t : n->triangles) {
            int32_t id = -1;
            if (!t.textureName.empty() && om->textureData.count(t.textureName)) {
                size_t k = 0;
                while (k < names.size() && names[k] != t.textureName) k++;
                if (k == names.size()) names.push_back(t.textureName);
                id = (int32_t)k;
            }
            tri_tex[cursor++] = id;
        }
        return;
    }
    tex_walk(n->left, om, names, tri_tex, cursor);
    tex_walk(n->right, om, names, tri_tex, cursor);
}
}
uint32_t ref_om_export_textures(void* om_, int32_t* tri_tex /* n_tris */, char* buf, uint32_t cap) {
    ObjectManager* om = (ObjectManager*)om_;
    std::vector<std::string> names; uint32_t cursor = 0;
    for (const auto& p : om->objTriangles) tex_walk(om->boundingVolumeHierarchy[p.first], om, names, tri_tex, cursor);
    std::string s;
    for (const std::string& nm : names) { s += nm; s += '\n'; }
    if (cap) { std::strncpy(buf, s.c_str(), cap - 1); buf[cap - 1] = 0; }
    return (uint32_t)names.size();
}

// ---- the hot path itself --------------------------------------------------------------------
// sendRaysAndIntersectPointsColors (simple_raytracer.cpp:505-525), result scattered to a dense
// W x H x 3 float image (row-major, y down), zero where the reference emitted nothing.
uint32_t ref_render(void* om, uint32_t W, uint32_t H, const float* light4, float* rgb /* H x W x 3 */) {
    ImageData d = sendRaysAndIntersectPointsColors(glm::vec2((float)W, (float)H),
                                                   glm::vec4(light4[0], light4[1], light4[2], light4[3]), (ObjectManager*)om);
    std::memset(rgb, 0, (size_t)W * H * 3 * sizeof(float));
    for (size_t i = 0; i < d.imagePoints.size(); i++) {
        int x = (int)d.imagePoints[i].x, y = (int)d.imagePoints[i].y;
        for (int c = 0; c < 3; c++) rgb[((size_t)y * W + x) * 3 + c] = d.imageColors[i][c];
    }
    return (uint32_t)d.imagePoints.size();
}

// Closest-hit ids and per-hit shading, evaluated with the reference's own leaf functions in the
// reference's loop structure (rayIntersection:405-457: object order, candidate order, strict <),
// for an arbitrary lightAmount.  Requires ref_om_export() to have tagged the leaves.
//   hit_id[H*W] canonical id or -1;  t[H*W];  tone[H*W*3] = softShadow(lightAmount,...) result;
//   lin[H*W*3] = pre-tone-map sum rebuilt from shadowIntersection + phongIllumination (:366-383).
//   rows [y0, y1) of the W x H frame only (outputs stay full-frame arrays; other rows are not written): a 3840x2160 frame with
//   64 light samples is hours of reference time, a band of scanlines is minutes.
void ref_trace_rows(void* om_, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1, const float* light3, int lightAmount,
                    int32_t* hit_id, float* t_out, float* tone, float* lin);
void ref_trace(void* om_, uint32_t W, uint32_t H, const float* light3, int lightAmount,
               int32_t* hit_id, float* t_out, float* tone, float* lin) {
    ref_trace_rows(om_, W, H, 0, H, light3, lightAmount, hit_id, t_out, tone, lin);
}
void ref_trace_rows(void* om_, uint32_t W, uint32_t H, uint32_t y0, uint32_t y1, const float* light3, int lightAmount,
                    int32_t* hit_id, float* t_out, float* tone, float* lin) {
    ObjectManager* om = (ObjectManager*)om_;
    glm::vec3 lightPos(light3[0], light3[1], light3[2]);
    glm::vec3 lightColor(1.0f, 1.0f, 1.0f);
    glm::vec2 imageSize((float)W, (float)H);
    Ray ray(glm::vec3(0.0f, 0.0f, 400.0f));
    for (int i = -imageSize.x / 2; i < imageSize.x / 2; ++i) {
        for (int j = -imageSize.y / 2; j < imageSize.y / 2; ++j) {
            ray.direction.x = i + 0.0f; ray.direction.y = j + 0.0f;
            int px = i + imageSize.x / 2, py = j + imageSize.y / 2;
            if ((uint32_t)py < y0 || (uint32_t)py >= y1) continue;
            size_t pix = (size_t)py * W + px;
            float best = INFINITY; int32_t best_id = -1; Triangle best_tri; std::string best_obj;
            for (const auto& pair : om->objTriangles) {
                const std::vector<Triangle>& cand = boundingBoxIntersection(om->boundingVolumeHierarchy[pair.first], ray);
                for (size_t k = 0; k < cand.size(); k++) {
                    float d = rayTriangleIntersection(&ray, &cand[k]);
                    if (d != -INFINITY && d < best) { best = d; best_id = tagged_id(cand[k]); best_tri = cand[k]; best_obj = pair.first; }
                }
            }
            hit_id[pix] = best_id; t_out[pix] = best;
            for (int c = 0; c < 3; c++) { tone[pix*3+c] = 0.f; lin[pix*3+c] = 0.f; }
            if (best_id == -1) continue;
            glm::vec3 objColor = best_tri.textureName.empty() ? om->objColors[best_obj] : best_tri.color;
            glm::vec3 col = softShadow(lightAmount, om, best_obj, &best_tri, &ray, lightPos, lightColor, objColor, best);
            for (int c = 0; c < 3; c++) tone[pix*3+c] = col[c];
            // pre-tone-map sum: the loop of softShadow:362-383 re-run with the reference's functions
            glm::vec3 testColor = objColor;
            if (!best_tri.textureName.empty()) {
                glm::vec3 P = ray.origin + best * ray.direction;
                glm::vec2 tc = getTextureCoordinate(calculateBarycentricCoords(&best_tri, P), best_tri.colorOneCoordinate, best_tri.colorTwoCoordinate, best_tri.colorThreeCoordinate);
                unsigned char* texData = om->textureData[best_tri.textureName];
                glm::ivec2 texDim = om->textureDimensions[best_tri.textureName];
                size_t texIndex = (static_cast<int>(tc.y) * texDim.x + static_cast<int>(tc.x)) * 3;
                testColor = glm::vec3(texData[texIndex] / 255.0f, texData[texIndex+1] / 255.0f, texData[texIndex+2] / 255.0f);
            }
            glm::vec3 sum(0.f), L = lightPos;
            for (int s = 0; s < lightAmount; s++) {
                bool sh = shadowIntersection(om, best_obj, L, best, ray);
                glm::vec3 c = phongIllumination(&best_tri, &ray, L, lightColor, testColor, om->objProperties[best_obj][0], om->objProperties[best_obj][1], om->objProperties[best_obj][2], best);
                if (sh) c /= 5;
                sum += c;
                switch (s % 3) { case 0: L.x += 3.0f; break; case 1: L.y += 3.0f; break; case 2: L.z += 3.0f; break; }
            }
            for (int c = 0; c < 3; c++) lin[pix*3+c] = sum[c];
        }
    }
}

// ---- leaf-function known-answer entry points ---------------------------------------------------
// rayTriangleIntersection (simple_raytracer.cpp:42-75); tri = 3 homogeneous points
void ref_kat_ray_triangle(uint32_t n, const float* ray_od /* n x 6 */, const float* tri /* n x 12 */, float* t) {
    for (uint32_t i = 0; i < n; i++) {
        Ray r(glm::vec3(ray_od[i*6+3], ray_od[i*6+4], ray_od[i*6+5]));
        r.origin = glm::vec3(ray_od[i*6], ray_od[i*6+1], ray_od[i*6+2]);
        const float* p = tri + (size_t)i * 12;
        Triangle T(glm::vec4(p[0],p[1],p[2],p[3]), glm::vec4(p[4],p[5],p[6],p[7]), glm::vec4(p[8],p[9],p[10],p[11]), glm::vec3(0), glm::vec3(0), glm::vec3(0));
        t[i] = rayTriangleIntersection(&r, &T);
    }
}
// intersectRayAabbNoOrigin (:252-293) and the origin-0 form intersectRayAabb (:204-248)
void ref_kat_ray_aabb(uint32_t n, const float* ray_od, const float* box /* n x 6: min,max */, uint8_t* hit, uint8_t* hit_origin0) {
    for (uint32_t i = 0; i < n; i++) {
        Ray r(glm::vec3(ray_od[i*6+3], ray_od[i*6+4], ray_od[i*6+5]));
        r.origin = glm::vec3(ray_od[i*6], ray_od[i*6+1], ray_od[i*6+2]);
        glm::vec3 mn(box[i*6], box[i*6+1], box[i*6+2]), mx(box[i*6+3], box[i*6+4], box[i*6+5]);
        hit[i] = intersectRayAabbNoOrigin(r, mn, mx) ? 1 : 0;
        hit_origin0[i] = intersectRayAabb(r.direction, mn, mx) ? 1 : 0;
    }
}
// phongIllumination (:144-200): in = ray_od(6), tri(12), light(3), objcolor(3), props(3), t(1) = 28 floats
void ref_kat_phong(uint32_t n, const float* in, float* rgb) {
    for (uint32_t i = 0; i < n; i++) {
        const float* q = in + (size_t)i * 28;
        Ray r(glm::vec3(q[3], q[4], q[5])); r.origin = glm::vec3(q[0], q[1], q[2]);
        const float* p = q + 6;
        Triangle T(glm::vec4(p[0],p[1],p[2],p[3]), glm::vec4(p[4],p[5],p[6],p[7]), glm::vec4(p[8],p[9],p[10],p[11]), glm::vec3(0), glm::vec3(0), glm::vec3(0));
        glm::vec3 c = phongIllumination(&T, &r, glm::vec3(q[18], q[19], q[20]), glm::vec3(1.f, 1.f, 1.f), glm::vec3(q[21], q[22], q[23]), q[24], q[25], q[26], q[27]);
        rgb[i*3] = c.x; rgb[i*3+1] = c.y; rgb[i*3+2] = c.z;
    }
}
// calculateBarycentricCoords (:79-117): tri(12) + point(3) -> (u,v,w)
void ref_kat_barycentric(uint32_t n, const float* in /* n x 15 */, float* uvw) {
    for (uint32_t i = 0; i < n; i++) {
        const float* p = in + (size_t)i * 15;
        Triangle T(glm::vec4(p[0],p[1],p[2],p[3]), glm::vec4(p[4],p[5],p[6],p[7]), glm::vec4(p[8],p[9],p[10],p[11]), glm::vec3(0), glm::vec3(0), glm::vec3(0));
        glm::vec3 b = calculateBarycentricCoords(&T, glm::vec3(p[12], p[13], p[14]));
        uvw[i*3] = b.x; uvw[i*3+1] = b.y; uvw[i*3+2] = b.z;
    }
}
// interpolateNormal (:132-140; its call at :162 is commented out in the reference): in = 3 vertex normals (9) + barycentrics (3)
void ref_kat_interp_normal(uint32_t n, const float* in /* n x 12 */, float* out /* n x 3 */) {
    for (uint32_t i = 0; i < n; i++) {
        const float* p = in + (size_t)i * 12;
        Triangle T(glm::vec4(0), glm::vec4(0), glm::vec4(0), glm::vec3(p[0], p[1], p[2]), glm::vec3(p[3], p[4], p[5]), glm::vec3(p[6], p[7], p[8]));
        glm::vec3 r = interpolateNormal(T, glm::vec3(p[9], p[10], p[11]));
        out[i*3] = r.x; out[i*3+1] = r.y; out[i*3+2] = r.z;
    }
}
// Reinhard + gamma tail of softShadow (:391-398) and the quantiser of rayIntersection (:447-449)
void ref_kat_tonemap(uint32_t n, const float* lin /* n x 3 */, float* tone, int32_t* q) {
    for (uint32_t i = 0; i < n; i++) {
        glm::vec3 color(lin[i*3], lin[i*3+1], lin[i*3+2]);
        color = color / (color + 0.5f);
        color = glm::pow(color, glm::vec3(1.1f, 1.1f, 1.1f));
        for (int c = 0; c < 3; c++) { tone[i*3+c] = color[c]; q[i*3+c] = int((color[c] * 255)); }
    }
}

// ---- Transformation.h:10-20 factories and the glm ops main() applies to them -------------------
void ref_mat_scale(float x, float y, float z, float* m)   { glm::mat4 M = Transformation::scaleObj(x, y, z); std::memcpy(m, &M[0][0], 64); }
void ref_mat_rotx(float a, float* m)  { glm::mat4 M = Transformation::rotateObjX(a); std::memcpy(m, &M[0][0], 64); }
void ref_mat_roty(float a, float* m)  { glm::mat4 M = Transformation::rotateObjY(a); std::memcpy(m, &M[0][0], 64); }
void ref_mat_rotz(float a, float* m)  { glm::mat4 M = Transformation::rotateObjZ(a); std::memcpy(m, &M[0][0], 64); }
void ref_mat_mirror(int x, int y, int z, float* m) { glm::mat4 M = Transformation::mirrorObj(x, y, z); std::memcpy(m, &M[0][0], 64); }
void ref_mat_shear(float xy, float xz, float yx, float yz, float zx, float zy, float* m) { glm::mat4 M = Transformation::shearObj(xy, xz, yx, yz, zx, zy); std::memcpy(m, &M[0][0], 64); }
void ref_mat_translate(float x, float y, float z, float* m) { glm::mat4 M = Transformation::changeObjPosition(glm::vec3(x, y, z)); std::memcpy(m, &M[0][0], 64); }
void ref_mat_view(const float* pos, const float* rot, float* m) { glm::mat4 M = Transformation::createViewMatrix(glm::vec3(pos[0], pos[1], pos[2]), glm::vec3(rot[0], rot[1], rot[2])); std::memcpy(m, &M[0][0], 64); }
void ref_mat_inverse(const float* a, float* m) { glm::mat4 A; std::memcpy(&A[0][0], a, 64); glm::mat4 M = glm::inverse(A); std::memcpy(m, &M[0][0], 64); }
void ref_mat_mul(const float* a, const float* b, float* m) { glm::mat4 A, B; std::memcpy(&A[0][0], a, 64); std::memcpy(&B[0][0], b, 64); glm::mat4 M = A * B; std::memcpy(m, &M[0][0], 64); }
void ref_mat_mul_vec4(const float* a, const float* v, float* out) { glm::mat4 A; std::memcpy(&A[0][0], a, 64); glm::vec4 r = A * glm::vec4(v[0], v[1], v[2], v[3]); std::memcpy(out, &r[0], 16); }
float ref_radians(float deg) { return glm::radians(deg); }

} // extern "C"
