// srt_host_c.cpp -- plain-C handles over srt_host.h so that Python (ctypes) tests, bench.py and other
// FFI users can drive the host-side mirror.  No compute here; exceptions never cross the boundary.
#include <cstring>
#include <string>

#include "srt_host.h"

using namespace srt_host;

static thread_local std::string g_err;
#define GUARD(...) try { __VA_ARGS__; return 0; } catch (const std::exception& e) { g_err = e.what(); return -1; } catch (...) { g_err = "unknown"; return -1; }

static mat4 to_mat(const float* m) { mat4 M; std::memcpy(&M[0][0], m, 64); return M; }
static void from_mat(const mat4& M, float* m) { std::memcpy(m, &M[0][0], 64); }

extern "C" {

const char* srth_last_error() { return g_err.c_str(); }

void* srth_om_new() { return new ObjectManager(); }
void srth_om_free(void* om) { delete (ObjectManager*)om; }
int srth_om_load_obj(void* om, const char* name) { GUARD(((ObjectManager*)om)->loadObjFile(name)) }

// Object fed from arrays with the loader's defaults (Object.cpp:29-34, 81-84)
int srth_om_add_object(void* om_, const char* name, uint32_t n, const float* points) {
    GUARD({
        ObjectManager* om = (ObjectManager*)om_;
        om->objColors[name] = vec3(1.f, 0.f, 0.f);
        om->objProperties[name] = vec3(0.2f, 0.5f, 15.0f);
        std::vector<Triangle>& tris = om->objTriangles[name];      // filled in place: a 69 k-triangle object is 10 MB of Triangle
        tris.clear(); tris.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            const float* p = points + (size_t)i * 12;
            tris[i].pointOne = vec4(p[0], p[1], p[2], p[3]); tris[i].pointTwo = vec4(p[4], p[5], p[6], p[7]); tris[i].pointThree = vec4(p[8], p[9], p[10], p[11]);
            tris[i].color = vec3(1.f, 1.f, 1.f);
        }
    })
}
// Array-fed textured object: the state loadObjFile leaves for a textured mesh (Object.cpp:98-161)
// decode an image file as the loader does; rgb == NULL: dimensions only.  Returns 0, or 1 if the file does not decode.
int srth_decode_image(const char* path, int32_t* w, int32_t* h, uint8_t* rgb) {
    try {
        Texture t;
        if (!load_texture(path, t)) return 1;
        *w = t.dim.x; *h = t.dim.y;
        if (rgb) std::memcpy(rgb, t.rgb.data(), t.rgb.size());
        return 0;
    } catch (const std::exception& e) { g_err = e.what(); return 1; }
}
int srth_om_add_texture(void* om_, const char* texname, int32_t w, int32_t h, const uint8_t* rgb) {
    GUARD({
        ObjectManager* om = (ObjectManager*)om_;
        Texture t; t.dim.x = w; t.dim.y = h; t.rgb.assign(rgb, rgb + (size_t)w * h * 3);
        om->textureData[texname] = std::move(t);
    })
}
int srth_om_add_textured_object(void* om_, const char* name, uint32_t n, const float* points, const float* texcoord, const char* texname) {
    int rc = srth_om_add_object(om_, name, n, points);
    if (rc) return rc;
    GUARD({
        std::vector<Triangle>& tris = ((ObjectManager*)om_)->objTriangles[name];
        for (uint32_t i = 0; i < n; i++) {
            tris[i].colorOneCoordinate = vec2(texcoord[i * 6], texcoord[i * 6 + 1]);
            tris[i].colorTwoCoordinate = vec2(texcoord[i * 6 + 2], texcoord[i * 6 + 3]);
            tris[i].colorThreeCoordinate = vec2(texcoord[i * 6 + 4], texcoord[i * 6 + 5]);
            tris[i].textureName = texname;
        }
    })
}
// object with several textures (house.obj): per triangle an index into the newline-separated `names`, -1 = untextured
int srth_om_add_multi_textured_object(void* om_, const char* name, uint32_t n, const float* points, const float* texcoord,
                                      const int32_t* tri_tex, const char* names) {
    int rc = srth_om_add_object(om_, name, n, points);
    if (rc) return rc;
    GUARD({
        std::vector<std::string> list;
        for (const char* p = names; *p;) { const char* e = std::strchr(p, '\n'); if (!e) e = p + std::strlen(p); list.emplace_back(p, e); p = *e ? e + 1 : e; }
        std::vector<Triangle>& tris = ((ObjectManager*)om_)->objTriangles[name];
        for (uint32_t i = 0; i < n; i++) {
            if (tri_tex[i] < 0) continue;
            tris[i].colorOneCoordinate = vec2(texcoord[i * 6], texcoord[i * 6 + 1]);
            tris[i].colorTwoCoordinate = vec2(texcoord[i * 6 + 2], texcoord[i * 6 + 3]);
            tris[i].colorThreeCoordinate = vec2(texcoord[i * 6 + 4], texcoord[i * 6 + 5]);
            tris[i].textureName = list.at((size_t)tri_tex[i]);
        }
    })
}
// objTriangles[dst] = getTriangles(src), as main() clones objects (simple_raytracer.cpp:565,597,644)
int srth_om_clone(void* om_, const char* src, const char* dst) {
    GUARD({ ObjectManager* om = (ObjectManager*)om_; std::vector<Triangle> t = om->getTriangles(src); om->objTriangles[dst] = t; })
}
int srth_om_set_color(void* om, const char* name, float r, float g, float b) { GUARD(((ObjectManager*)om)->setColor(name, vec3(r, g, b))) }
int srth_om_set_props(void* om, const char* name, float ka, float ks, float sh) { GUARD(((ObjectManager*)om)->objProperties[name] = vec3(ka, ks, sh)) }
int srth_om_transform(void* om, const char* name, const float* m) { GUARD(((ObjectManager*)om)->transformTriangles(name, to_mat(m))) }
void srth_set_build_tasks(int on) { setHierarchyBuildTasks(on != 0); }
int srth_sort_keys_both_ways(const float* keys, uint32_t n, uint32_t* order_parallel, uint32_t* order_std) {
    GUARD(sort_keys_both_ways(keys, n, order_parallel, order_std))
}
int srth_om_build_bvh(void* om, const char* name) { GUARD(((ObjectManager*)om)->createBoundingHierarchy(name)) }
int64_t srth_om_num_tris(void* om, const char* name) {
    try { return (int64_t)((ObjectManager*)om)->getTriangles(name).size(); } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int srth_om_get_points(void* om, const char* name, float* out) {
    GUARD({
        const std::vector<Triangle>& v = ((ObjectManager*)om)->getTriangles(name);
        for (size_t i = 0; i < v.size(); i++) {
            const vec4* p[3] = { &v[i].pointOne, &v[i].pointTwo, &v[i].pointThree };
            for (int k = 0; k < 3; k++) for (int c = 0; c < 4; c++) out[(i * 3 + k) * 4 + c] = (*p[k])[c];
        }
    })
}
// the hierarchy of an object as srt_scene_update_frame takes it: node count, then points (n x 12, source order, at build time), the
// build's permutation (n), node boxes (m x 3 each, pre-order)
int64_t srth_om_hierarchy_nodes(void* om, const char* name) {
    try { return (int64_t)((ObjectManager*)om)->boundingVolumeHierarchy.at(name).nodes.size(); } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int srth_om_hierarchy(void* om, const char* name, float* points, uint32_t* order, float* node_min, float* node_max) {
    GUARD({
        const ObjectManager::Hierarchy& h = ((ObjectManager*)om)->boundingVolumeHierarchy.at(name);
        std::memcpy(points, h.points.data(), h.points.size() * sizeof(float));
        std::memcpy(order, h.order.data(), h.order.size() * sizeof(uint32_t));
        std::memcpy(node_min, h.node_min.data(), h.node_min.size() * sizeof(float));
        std::memcpy(node_max, h.node_max.data(), h.node_max.size() * sizeof(float));
    })
}
uint64_t srth_renderer_fast_frames(void* r) { return ((Renderer*)r)->fastFrames(); }
void srth_renderer_set_fast_path(void* r, int on) { ((Renderer*)r)->setFastPath(on != 0); }

int srth_om_get_tri_attrs(void* om, const char* name, float* texcoord, float* color, int32_t* has_tex, float* normals) {
    GUARD({
        const std::vector<Triangle>& v = ((ObjectManager*)om)->getTriangles(name);
        for (size_t i = 0; i < v.size(); i++) {
            const vec2* tc[3] = { &v[i].colorOneCoordinate, &v[i].colorTwoCoordinate, &v[i].colorThreeCoordinate };
            for (int k = 0; k < 3; k++) { texcoord[i * 6 + k * 2] = tc[k]->x; texcoord[i * 6 + k * 2 + 1] = tc[k]->y; }
            color[i * 3] = v[i].color.x; color[i * 3 + 1] = v[i].color.y; color[i * 3 + 2] = v[i].color.z;
            has_tex[i] = v[i].textureName.empty() ? 0 : 1;
            const vec3* nn[3] = { &v[i].normalOne, &v[i].normalTwo, &v[i].normalThree };
            for (int k = 0; k < 3; k++) for (int c = 0; c < 3; c++) normals[i * 9 + k * 3 + c] = (*nn[k])[c];
        }
    })
}

// ---- flattener ---------------------------------------------------------------------------------
void* srth_flatten(void* om) {
    try { return new FlatScene(flattenScene((ObjectManager*)om)); } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void srth_flat_free(void* f) { delete (FlatScene*)f; }
void srth_flat_desc(void* f, srt_scene_desc* out) { *out = ((FlatScene*)f)->desc(); }
uint32_t srth_flat_names(void* f_, char* buf, uint32_t cap) {
    FlatScene* f = (FlatScene*)f_;
    std::string s;
    for (const auto& n : f->names) { s += n; s += '\n'; }
    if (cap) { std::strncpy(buf, s.c_str(), cap - 1); buf[cap - 1] = 0; }
    return (uint32_t)f->names.size();
}

// ---- drop-in entry point (GPU) + writer -----------------------------------------------------------
// Dense H x W x 3 float image like oracle/ref_harness.cpp's ref_render: 0 where nothing was emitted.
int64_t srth_render(void* om, uint32_t W, uint32_t H, const float* light4, int light_amount, int device, float* rgb) {
    try {
        ImageData d = sendRaysAndIntersectPointsColors(vec2((float)W, (float)H), vec4(light4[0], light4[1], light4[2], light4[3]),
                                                       (ObjectManager*)om, light_amount, device);
        std::memset(rgb, 0, (size_t)W * H * 3 * sizeof(float));
        for (size_t i = 0; i < d.imagePoints.size(); i++) {
            const size_t x = (size_t)d.imagePoints[i].x, y = (size_t)d.imagePoints[i].y;
            rgb[(y * W + x) * 3] = d.imageColors[i].x; rgb[(y * W + x) * 3 + 1] = d.imageColors[i].y; rgb[(y * W + x) * 3 + 2] = d.imageColors[i].z;
        }
        return (int64_t)d.imagePoints.size();
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int srth_write_bmp(const char* path, uint32_t W, uint32_t H, const uint8_t* rgb) { GUARD(writeBmp(path, W, H, rgb)) }

// ---- Transformation.h factories + the glm ops main() applies --------------------------------------
float srth_radians(float d) { return radians(d); }
void srth_mat_scale(float x, float y, float z, float* m) { from_mat(Transformation::scaleObj(x, y, z), m); }
void srth_mat_rotx(float a, float* m) { from_mat(Transformation::rotateObjX(a), m); }
void srth_mat_roty(float a, float* m) { from_mat(Transformation::rotateObjY(a), m); }
void srth_mat_rotz(float a, float* m) { from_mat(Transformation::rotateObjZ(a), m); }
void srth_mat_mirror(int x, int y, int z, float* m) { from_mat(Transformation::mirrorObj(x, y, z), m); }
void srth_mat_shear(float xy, float xz, float yx, float yz, float zx, float zy, float* m) { from_mat(Transformation::shearObj(xy, xz, yx, yz, zx, zy), m); }
void srth_mat_translate(float x, float y, float z, float* m) { from_mat(Transformation::changeObjPosition(vec3(x, y, z)), m); }
void srth_mat_view(const float* pos, const float* rot, float* m) { from_mat(Transformation::createViewMatrix(vec3(pos[0], pos[1], pos[2]), vec3(rot[0], rot[1], rot[2])), m); }
void srth_mat_inverse(const float* a, float* m) { from_mat(inverse(to_mat(a)), m); }
void srth_mat_mul(const float* a, const float* b, float* m) { from_mat(to_mat(a) * to_mat(b), m); }
void srth_mat_mul_vec4(const float* a, const float* v, float* out) { vec4 r = to_mat(a) * vec4(v[0], v[1], v[2], v[3]); std::memcpy(out, &r.x, 16); }

// ---- Renderer: a device scene kept across frames (srt_host.h) ------------------------------------------------------------
void* srth_renderer_new(int device) { try { return new Renderer(device); } catch (...) { return nullptr; } }
void srth_renderer_free(void* r) { delete (Renderer*)r; }
static int64_t image_to_dense(const ImageData& d, uint32_t W, uint32_t H, float* rgb) {
    std::memset(rgb, 0, (size_t)W * H * 3 * sizeof(float));
    for (size_t i = 0; i < d.imagePoints.size(); i++) {
        const size_t x = (size_t)d.imagePoints[i].x, y = (size_t)d.imagePoints[i].y;
        float* c = rgb + (y * W + x) * 3;
        c[0] = d.imageColors[i].x; c[1] = d.imageColors[i].y; c[2] = d.imageColors[i].z;
    }
    return (int64_t)d.imagePoints.size();
}
// rgb = NULL: the frame is rendered and collected but not written out densely (timing runs)
int64_t srth_renderer_render(void* r, void* om, uint32_t W, uint32_t H, const float* light4, int light_amount, float* rgb) {
    try {
        ImageData d = ((Renderer*)r)->render(vec2((float)W, (float)H), vec4(light4[0], light4[1], light4[2], light4[3]), (ObjectManager*)om, light_amount);
        return rgb ? image_to_dense(d, W, H, rgb) : (int64_t)d.imagePoints.size();
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int srth_renderer_submit(void* r, void* om, uint32_t W, uint32_t H, const float* light4, int light_amount) {
    GUARD(((Renderer*)r)->submit(vec2((float)W, (float)H), vec4(light4[0], light4[1], light4[2], light4[3]), (ObjectManager*)om, light_amount))
}
int64_t srth_renderer_collect(void* r, uint32_t W, uint32_t H, float* rgb) {
    try {
        ImageData d = ((Renderer*)r)->collect();
        return rgb ? image_to_dense(d, W, H, rgb) : (int64_t)d.imagePoints.size();
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}
int64_t srth_renderer_render_from_camera(void* r, void* om, uint32_t W, uint32_t H, const float* light4_world, const float* view16, int light_amount,
                                         int scene_changed, float* rgb) {
    try {
        ImageData d = ((Renderer*)r)->renderFromCamera(vec2((float)W, (float)H), vec4(light4_world[0], light4_world[1], light4_world[2], light4_world[3]),
                                                      to_mat(view16), (ObjectManager*)om, light_amount, scene_changed != 0);
        return rgb ? image_to_dense(d, W, H, rgb) : (int64_t)d.imagePoints.size();
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

// MultiRenderer: several devices, one process (the framebuffer split on the C++ host)
void* srth_multi_new(const int* devices, uint32_t n, uint32_t block_rows) {
    try { return new MultiRenderer(std::vector<int>(devices, devices + n), block_rows); } catch (const std::exception& e) { g_err = e.what(); return nullptr; }
}
void srth_multi_free(void* r) { delete (MultiRenderer*)r; }
int64_t srth_multi_render(void* r, void* om, uint32_t W, uint32_t H, const float* light4, int light_amount, float* rgb) {
    try {
        ImageData d = ((MultiRenderer*)r)->render(vec2((float)W, (float)H), vec4(light4[0], light4[1], light4[2], light4[3]), (ObjectManager*)om, light_amount);
        return rgb ? image_to_dense(d, W, H, rgb) : (int64_t)d.imagePoints.size();
    } catch (const std::exception& e) { g_err = e.what(); return -1; }
}

} // extern "C"
