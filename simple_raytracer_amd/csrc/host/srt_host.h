// srt_host.h -- host-side C++ mirror of the reference's scene interface for the ray-trace path.
//
// The reference is compiled C++ with no plugin layer; the code either side of its hot path is
// ObjectManager (Object.h:59-89, Object.cpp) and Transformation (Transformation.h:10-20).  This header
// offers the same names, argument meanings and error behaviour on top of the C ABI (include/srt.h),
// without glm / tinyobjloader / CImg, so that a scene script written against the reference's main()
// (simple_raytracer.cpp:530-796) reads the same here:
//
//     srt_host::ObjectManager om;
//     om.loadObjFile("cube.obj");
//     om.transformTriangles("cube.obj", srt_host::Transformation::scaleObj(20, 20, 20));
//     om.createBoundingHierarchy("cube.obj");
//     srt_host::ImageData img = srt_host::sendRaysAndIntersectPointsColors({600, 400}, light, &om);
//
// Only sendRaysAndIntersectPointsColors touches the GPU (through srt_scene_create / srt_render).
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/srt.h"

namespace srt_host {

struct vec2 { float x = 0, y = 0; vec2() = default; vec2(float a, float b) : x(a), y(b) {} };
struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default; vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float& operator[](int i) { return (&x)[i]; } const float& operator[](int i) const { return (&x)[i]; }
};
struct vec4 {
    float x = 0, y = 0, z = 0, w = 0;
    vec4() = default; vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    float& operator[](int i) { return (&x)[i]; } const float& operator[](int i) const { return (&x)[i]; }
};
struct ivec2 { int x = 0, y = 0; };
// column-major 4x4, m[c][r] like glm::mat4
struct mat4 {
    vec4 c[4];
    mat4() = default;
    explicit mat4(float diag) { c[0].x = diag; c[1].y = diag; c[2].z = diag; c[3].w = diag; }
    vec4& operator[](int i) { return c[i]; } const vec4& operator[](int i) const { return c[i]; }
};
vec4 operator*(const mat4& m, const vec4& v);      // glm type_mat4x4.inl:537-573 op order
mat4 operator*(const mat4& a, const mat4& b);      // glm type_mat4x4.inl:660-705 op order
mat4 inverse(const mat4& m);                       // glm func_matrix.inl:388-446 op order
float radians(float degrees);                      // glm func_trigonometric.inl:9-14

// Transformation.h:10-20 -- same eight factories, same matrices (incl. the reference's sign convention)
class Transformation {
public:
    static mat4 scaleObj(float sx, float sy, float sz);
    static mat4 rotateObjX(float degree);
    static mat4 rotateObjY(float degree);
    static mat4 rotateObjZ(float degree);
    static mat4 mirrorObj(bool mirrorX, bool mirrorY, bool mirrorZ);
    static mat4 shearObj(float shearXY, float shearXZ, float shearYX, float shearYZ, float shearZX, float shearZY);
    static mat4 changeObjPosition(vec3 position);
    static mat4 createViewMatrix(vec3 position, vec3 rotation);
};

// Object.h:15-36
class Triangle {
public:
    vec4 pointOne, pointTwo, pointThree;
    vec3 normalOne, normalTwo, normalThree;
    vec2 colorOneCoordinate, colorTwoCoordinate, colorThreeCoordinate;
    vec3 color;
    std::string textureName;
};

// Object.h:38-44
class Ray {
public:
    vec3 origin, direction;
    explicit Ray(vec3 d) : origin(0.f, 0.f, 0.f), direction(d) {}
};

// Object.h:46-57, with the by-value triangle vectors replaced by a range into the object's
// leaf-ordered index list (every inner node's triangles are a contiguous range of it).
struct Node {
    vec3 minBox, maxBox;
    uint32_t first = 0, count = 0;     // range in ObjectManager::Hierarchy::order
    int32_t left = -1, right = -1;     // indices into Hierarchy::nodes, -1 = none
};

struct Texture { std::vector<unsigned char> rgb; ivec2 dim; };
// baseline / progressive JPEG -> 8-bit RGB with stb_image's arithmetic (srt_jpeg.cpp)
bool decode_jpeg(const std::vector<unsigned char>& file_bytes, Texture& out);
// createBoundingHierarchy as pool tasks (default) or on the calling thread alone (false): for hosts that build several frames'
// hierarchies concurrently on their own threads (examples/frame_pipeline.py).  Same trees either way.
void setHierarchyBuildTasks(bool on);
// test hook for the hierarchy builder's sort (srt_host.cpp, ExactSort): both permutations of 0..n-1 by keys[]
void sort_keys_both_ways(const float* keys, uint32_t n, uint32_t* order_parallel, uint32_t* order_std);
// what stbi_load(path, &w, &h, &ch, 3) gives the reference (Object.cpp:57): PNG / JPEG / PPM / BMP by content
bool load_texture(const std::string& path, Texture& out);

// Object.h:59-89
class ObjectManager {
public:
    // nodes[0] = root, DFS pre-order; order = built (leaf, left-to-right) order as indices into the object's triangles AT
    // BUILD TIME; triangles = those triangles, copied in that order.  The reference's Node keeps its triangles by value
    // (Object.h:46-57), so what is rendered is the geometry as it was when createBoundingHierarchy ran, whatever
    // transformTriangles / setTriangles do to the object afterwards; the copy keeps that behaviour.
    // points / node_min / node_max: what srt_scene_update_frame takes (include/srt.h, f1): the object's transformed points AT BUILD TIME
    // in source order (12 floats a triangle) and the node boxes in the nodes' (pre-)order (3 floats each).
    struct Hierarchy {
        std::vector<Node> nodes; std::vector<uint32_t> order; std::vector<Triangle> triangles;
        std::vector<float> points, node_min, node_max;
    };
    std::unordered_map<std::string, vec3> minBox, maxBox;
    std::unordered_map<std::string, vec3> objProperties;       // ambientStrength, specularStrength, shininess
    std::unordered_map<std::string, std::vector<Triangle>> objTriangles;
    std::unordered_map<std::string, Hierarchy> boundingVolumeHierarchy;
    std::unordered_map<std::string, vec3> objColors;
    std::unordered_map<std::string, Texture> textureData;      // + textureDimensions (Object.h:70-71)

    void loadObjFile(const std::string& objFilename);                               // Object.cpp:25-170
    const std::vector<Triangle>& getTriangles(const std::string& objFilename) const; // throws std::out_of_range (:174)
    void setTriangles(const std::string& objFilename, const std::vector<Triangle>& triangles);
    void transformTriangles(const std::string& objFilename, const mat4& matrix);    // Object.cpp:183-190
    void createBoundingHierarchy(const std::string& objFilename);                   // Object.cpp:275-284
    void setColor(const std::string& objFilename, const vec3& color);
    vec3 getColor(const std::string& objFilename) const;                            // throws std::out_of_range (:292)
};

// The flat scene of include/srt.h built from an ObjectManager: objects in objTriangles iteration
// order (what rayIntersection:409 walks), trees DFS left-first, triangles in visit order.
struct FlatScene {
    std::vector<float> node_min, node_max;
    std::vector<int32_t> node_left, node_right, node_first, node_count;
    std::vector<uint32_t> obj_root;
    std::vector<float> tri_points, tri_texcoord, tri_normals;
    std::vector<int32_t> tri_obj, tri_tex;
    std::vector<float> obj_color, obj_material;
    std::vector<uint8_t> tex_rgb; std::vector<uint64_t> tex_off; std::vector<uint32_t> tex_w, tex_h;
    std::vector<std::string> names, tex_names;
    srt_scene_desc desc() const;       // pointers into this object
};
// Throws std::runtime_error if an object has no hierarchy (the reference null-derefs there, :422).
// A triangle whose textureName did not load (the reference null-derefs at :354-358) falls back to the
// object colour (tri_tex = -1); SURVEY.md s8 f3.
FlatScene flattenScene(ObjectManager* objManager);

// simple_raytracer.cpp:28
struct ImageData { std::vector<vec2> imagePoints; std::vector<vec3> imageColors; };

// Drop-in for simple_raytracer.cpp:505-525: same signature shape, same result (hit pixels in the
// reference's column-major emission order, colours as integer-valued floats), computed on HIP
// device `device` through the C ABI.  lightAmount is the reference's hard-wired 1 (:445) by default.
// Throws std::runtime_error carrying srt_strerror() on failure (the reference signals nothing).
ImageData sendRaysAndIntersectPointsColors(const vec2& imageSize, const vec4& lightPos, ObjectManager* objManager,
                                           int lightAmount = 1, int device = 0);

// A device scene that outlives the frame.  The reference builds and drops everything per frame; its drop-in call above does the
// same through a Renderer it keeps per thread and device, so that frame n + 1 re-uses frame n's device allocations
// (srt_scene_update) and pinned buffers instead of paying hipMalloc / hipFree / pageable copies every frame.
//   render()            = the drop-in call on this handle;
//   submit() / collect() = the same in two halves: submit() flattens, uploads and enqueues the frame and returns, collect() waits
//                         and builds the ImageData -- whatever the caller does in between (building the next frame's hierarchies)
//                         overlaps with the GPU;
//   renderFromCamera()  = CAMERA MODE (an extension, include/srt.h srt_params.ray_matrix): the scene stays in world space and is
//                         uploaded only when `sceneChanged`; each frame passes the viewMatrix whose inverse the reference would
//                         have applied to every triangle, and the light's WORLD position.
class Renderer {
public:
    explicit Renderer(int device = 0);
    ~Renderer();
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;
    ImageData render(const vec2& imageSize, const vec4& lightPos, ObjectManager* objManager, int lightAmount = 1);
    void submit(const vec2& imageSize, const vec4& lightPos, ObjectManager* objManager, int lightAmount = 1);
    ImageData collect();
    ImageData renderFromCamera(const vec2& imageSize, const vec4& lightPosWorld, const mat4& viewMatrix, ObjectManager* objManager,
                               int lightAmount = 1, bool sceneChanged = false);
    // upload() takes the device half of the per-frame rebuild (srt_scene_update_frame: 52 bytes a triangle, records derived on the
    // device) whenever the ObjectManager has the objects (names, order, triangle and node counts), textures and a sample of the
    // per-triangle attributes of the scene that is resident; anything else goes through flattenScene + srt_scene_update / create.
    // fastFrames() = how many uploads took the short way (tests, frame_pipeline).
    uint64_t fastFrames() const { return fast_frames_; }
    void setFastPath(bool on) { fast_path_ = on; }
private:
    void upload(ObjectManager* objManager);
    bool upload_frame(ObjectManager* objManager);
    void enqueue(const vec2& imageSize, const vec4& lightPos, int lightAmount, const mat4* viewMatrix);
    int device_;
    srt_scene* scene_ = nullptr;
    std::vector<std::string> sig_names_; std::vector<uint32_t> sig_tris_, sig_nodes_; uint64_t sig_attr_ = 0;      // what is resident
    bool fast_path_ = true; uint64_t fast_frames_ = 0;
    uint8_t* rgb8_ = nullptr; size_t rgb8_bytes_ = 0;      // pinned
    uint32_t W_ = 0, H_ = 0;
    bool pending_ = false;
    size_t last_count_ = 0;          // pixels the previous frame emitted: the next frame's ImageData is sized by it while the GPU renders
};

// The framebuffer split of several GPUs driven from ONE C++ host process: the scene is replicated (one device scene per entry of
// `devices`), every frame is dealt in scanline blocks of `blockRows` rows block-cyclically (srt_params.block_rows / block_first /
// block_stride), each device renders its rows into its own pinned buffer on its own stream -- the devices run concurrently -- and
// the host puts the rows together.  Results are the single-device results bit for bit (a pixel does not depend on who renders
// it).  (bench.py's N-GPU path is one process per GPU with one RCCL gather per step; this is the same split for a host program
// that owns all the GPUs of a node.)
class MultiRenderer {
public:
    explicit MultiRenderer(const std::vector<int>& devices, uint32_t blockRows = 8);
    ~MultiRenderer();
    MultiRenderer(const MultiRenderer&) = delete;
    MultiRenderer& operator=(const MultiRenderer&) = delete;
    ImageData render(const vec2& imageSize, const vec4& lightPos, ObjectManager* objManager, int lightAmount = 1);
    const std::vector<uint8_t>& frame() const { return frame_; }      // H x W x 3 of the last render (black = nothing emitted)
private:
    struct Part { int device = 0; srt_scene* scene = nullptr; uint8_t* rgb8 = nullptr; size_t bytes = 0; uint32_t rows = 0; };
    std::vector<Part> parts_;
    uint32_t blockRows_;
    std::vector<uint8_t> frame_;
};

// drawImage (:461-498) pixel contract: 8-bit RGB, every all-black pixel -> (173,216,230); written as
// a 24-bit BMP like CImg::save_bmp.  displayImage is not supported (headless).
void drawImage(const vec2& imgSize, const std::vector<vec2>& imagePoints, const std::vector<vec3>& imageColors,
               const int& angleDegree, const bool& saveImage, const std::string& directory = "images/generation");
void writeBmp(const std::string& path, uint32_t W, uint32_t H, const uint8_t* rgb /* H x W x 3, top-down */);

} // namespace srt_host
