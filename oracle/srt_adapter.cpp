// oracle/srt_adapter.cpp -- TEST INFRASTRUCTURE (reference-side binding, kept verbatim in INTEGRATION.md).
//
// The translation unit a maintainer of the reference would add to call the HIP path through include/srt.h from
// the reference's OWN ObjectManager / Node / Triangle types.  oracle/Makefile compiles it against the reference's
// headers (build container only) into oracle/_ref/libsrt_ref_adapter.so so that the GPU tests can run the
// reference's data structures end to end through the C ABI.  tests/test_host_mirror.py asserts that the code
// below and the block in INTEGRATION.md are the same text.
// srt_adapter.cpp  (reference side; uses the reference's Object.h types)
#include "Object.h"
#include "srt.h"
#include <stdexcept>

struct ImageData { std::vector<glm::vec2> imagePoints; std::vector<glm::vec3> imageColors; };

namespace {
struct Flat {
    std::vector<float> nmin, nmax, pts, tc, color, mat;
    std::vector<int32_t> left, right, first, count, tobj, ttex;
    std::vector<uint32_t> root, tw, th; std::vector<uint64_t> toff; std::vector<uint8_t> trgb;
    std::unordered_map<std::string, int32_t> tex_id;
};
int32_t walk(Node* n, Flat& f, int32_t obj, ObjectManager* om) {
    int32_t me = (int32_t)f.left.size();
    for (int a = 0; a < 3; a++) { f.nmin.push_back(n->minBox[a]); }
    for (int a = 0; a < 3; a++) { f.nmax.push_back(n->maxBox[a]); }
    f.left.push_back(-1); f.right.push_back(-1); f.first.push_back(-1); f.count.push_back(0);
    if (!n->left && !n->right) {
        f.first[me] = (int32_t)f.tobj.size(); f.count[me] = (int32_t)n->triangles.size();
        for (const Triangle& t : n->triangles) {
            for (const glm::vec4* p : { &t.pointOne, &t.pointTwo, &t.pointThree }) for (int c = 0; c < 4; c++) f.pts.push_back((*p)[c]);
            for (const glm::vec2* q : { &t.colorOneCoordinate, &t.colorTwoCoordinate, &t.colorThreeCoordinate }) { f.tc.push_back(q->x); f.tc.push_back(q->y); }
            f.tobj.push_back(obj);
            int32_t tid = -1;
            auto it = t.textureName.empty() ? om->textureData.end() : om->textureData.find(t.textureName);
            if (it != om->textureData.end()) {
                auto k = f.tex_id.find(t.textureName);
                if (k == f.tex_id.end()) {
                    glm::ivec2 d = om->textureDimensions[t.textureName];
                    tid = (int32_t)f.tw.size(); f.tex_id[t.textureName] = tid;
                    f.toff.push_back(f.trgb.size()); f.tw.push_back(d.x); f.th.push_back(d.y);
                    f.trgb.insert(f.trgb.end(), it->second, it->second + (size_t)d.x * d.y * 3);
                } else tid = k->second;
            }
            f.ttex.push_back(tid);
        }
        return me;
    }
    int32_t l = walk(n->left, f, obj, om);  f.left[me] = l;
    int32_t r = walk(n->right, f, obj, om); f.right[me] = r;
    return me;
}
} // namespace

ImageData sendRaysAndIntersectPointsColors(const glm::vec2& imageSize, const glm::vec4& lightPos, ObjectManager* om) {
    Flat f; int32_t obj = 0;
    for (const auto& pair : om->objTriangles) {                       // same order as rayIntersection:409
        f.root.push_back((uint32_t)f.left.size());
        glm::vec3 c = om->objColors[pair.first], m = om->objProperties[pair.first];
        for (int a = 0; a < 3; a++) { f.color.push_back(c[a]); f.mat.push_back(m[a]); }
        walk(om->boundingVolumeHierarchy[pair.first], f, obj++, om);
    }
    srt_scene_desc d = {};
    d.n_objects = (uint32_t)f.root.size(); d.n_nodes = (uint32_t)f.left.size(); d.n_tris = (uint32_t)f.tobj.size(); d.n_textures = (uint32_t)f.tw.size();
    d.node_min = f.nmin.data(); d.node_max = f.nmax.data(); d.node_left = f.left.data(); d.node_right = f.right.data();
    d.node_first = f.first.data(); d.node_count = f.count.data(); d.obj_root = f.root.data();
    d.tri_points = f.pts.data(); d.tri_obj = f.tobj.data(); d.tri_tex = f.ttex.data(); d.tri_texcoord = f.tc.data();
    d.obj_color = f.color.data(); d.obj_material = f.mat.data();
    d.tex_rgb = f.trgb.data(); d.tex_off = f.toff.data(); d.tex_w = f.tw.data(); d.tex_h = f.th.data();

    srt_scene* scene = nullptr;
    int rc = srt_scene_create(/*device*/0, &d, &scene);
    if (rc) throw std::runtime_error(srt_strerror(rc));
    const uint32_t W = (uint32_t)imageSize.x, H = (uint32_t)imageSize.y;
    srt_params p; srt_params_default(&p, W, H);
    float light[3]; const float base[3] = { lightPos.x, lightPos.y, lightPos.z };
    srt_light_staircase(base, 1, light);                               // lightAmount = 1 (:445)
    p.light_pos = light; p.background[0] = p.background[1] = p.background[2] = 0;   // black = "not emitted" (:518)
    std::vector<uint8_t> rgb8((size_t)W * H * 3);
    rc = srt_render(scene, &p, nullptr, nullptr, nullptr, rgb8.data(), nullptr);
    srt_scene_destroy(scene);
    if (rc) throw std::runtime_error(srt_strerror(rc));
    ImageData out;
    for (uint32_t x = 0; x < W; x++) for (uint32_t y = 0; y < H; y++) {     // reference emission order (:511-513)
        const uint8_t* c = &rgb8[((size_t)y * W + x) * 3];
        if (c[0] | c[1] | c[2]) { out.imagePoints.emplace_back(x, y); out.imageColors.emplace_back(c[0], c[1], c[2]); }
    }
    return out;
}

// C entry for the tests: dense H x W x 3 float image of the emitted (px, py, rgb) list (0 where nothing was emitted),
// same shape as ref_render() of oracle/ref_harness.cpp.  Returns the number of emitted pixels, -1 on error.
extern "C" long long srt_adapter_render(void* om, uint32_t W, uint32_t H, const float* light4, float* rgb, char* err, uint32_t err_cap) {
    try {
        ImageData d = sendRaysAndIntersectPointsColors(glm::vec2((float)W, (float)H), glm::vec4(light4[0], light4[1], light4[2], light4[3]), (ObjectManager*)om);
        for (size_t i = 0; i < (size_t)W * H * 3; i++) rgb[i] = 0.0f;
        for (size_t i = 0; i < d.imagePoints.size(); i++) {
            const size_t x = (size_t)d.imagePoints[i].x, y = (size_t)d.imagePoints[i].y;
            for (int c = 0; c < 3; c++) rgb[(y * W + x) * 3 + c] = d.imageColors[i][c];
        }
        return (long long)d.imagePoints.size();
    } catch (const std::exception& e) {
        if (err && err_cap) { size_t n = 0; for (const char* p = e.what(); *p && n + 1 < err_cap; p++) err[n++] = *p; err[n] = 0; }
        return -1;
    }
}
