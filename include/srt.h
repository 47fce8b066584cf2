/* include/srt.h -- C ABI of the MI355X-native ray-trace core for simple_raytracer scenes.
 *
 * This is the drop-in boundary for ONE path of leonlang/simple_raytracer: the per-pixel loop
 *
 *     ImageData sendRaysAndIntersectPointsColors(const glm::vec2& imageSize,
 *                                                const glm::vec4& lightPos,
 *                                                ObjectManager* objManager);   // simple_raytracer.cpp:505
 *
 * called once per frame from main() (simple_raytracer.cpp:784) and consumed by drawImage (:793).
 * The reference has no FFI of its own; the seam is that one call.  Everything here is plain C:
 * PODs, caller-owned buffers, int return codes, no exceptions, no glm / STL / torch types.
 * All compute behind these entry points is hand-written HIP for gfx950; there is NO CPU fallback
 * (the CPU restatement used to check results lives in oracle/ and is test infrastructure only).
 *
 * Thread-safety: one host thread per srt_scene handle.  One handle lives on one HIP device.
 */
#ifndef SRT_H
#define SRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRT_ABI_VERSION 3

/* ---- error codes (the reference signals nothing: it prints, throws or crashes) -- */
enum {
    SRT_OK              = 0,
    SRT_ERR_ARG         = 1,   /* null pointer / zero size / inconsistent counts                    */
    SRT_ERR_LAYOUT      = 2,   /* scene arrays violate the layout contract below                    */
    SRT_ERR_DEVICE      = 3,   /* HIP runtime error (srt_last_hip_error() has the hipError_t)       */
    SRT_ERR_NO_GPU      = 4,   /* no HIP device visible: the product path never falls back to CPU   */
    SRT_ERR_TEXTURE     = 5,   /* triangle references a texture id >= n_textures                    */
    SRT_ERR_LIMIT       = 6,   /* size exceeds an implementation limit (n_tris < 2^26, n_nodes < 2^26) */
    SRT_ERR_OOM         = 7    /* host allocation failed (std::bad_alloc caught at the boundary)    */
};
/* No entry point lets a C++ exception escape: bad_alloc -> SRT_ERR_OOM, anything else -> SRT_ERR_DEVICE. */

/* ---- flat scene: what the host hands over once per frame ---------------------------------------
 * It is the reference's ObjectManager state (Object.h:59-89) with the string-keyed maps and Node*
 * trees written out as arrays:
 *   objects    in objTriangles iteration order (the order rayIntersection:409 visits them),
 *   nodes      each object's Node tree (Object.h:46-57) in any order, children by global index,
 *   triangles  in VISIT order: object order -> DFS left-first leaf order (boundingBoxIntersection
 *              :296-317) -> in-leaf order.  The index in this order is the canonical triangle id
 *              reported in hit_id, and ties in t resolve to the lowest id exactly as the reference's
 *              strict '<' (:429) does.
 * Layout contract (checked, SRT_ERR_LAYOUT): every node is reachable from exactly one root; a leaf
 * has left == right == -1 and owns triangles [first, first+count), 0 <= count; walking all objects'
 * trees DFS left-first meets the leaves' ranges contiguously in increasing order, covering
 * [0, n_tris) exactly once; tri_obj[i] equals the object whose tree owns triangle i.
 */
typedef struct srt_scene_desc {
    uint32_t n_objects, n_nodes, n_tris, n_textures;
    /* nodes: Node::minBox / maxBox (Object.h:47-48), left / right (:50-51) */
    const float*    node_min;      /* n_nodes x 3                                                   */
    const float*    node_max;      /* n_nodes x 3                                                   */
    const int32_t*  node_left;     /* n_nodes, global child index or -1                             */
    const int32_t*  node_right;    /* n_nodes                                                       */
    const int32_t*  node_first;    /* n_nodes, first triangle of a leaf (ignored for inner nodes)   */
    const int32_t*  node_count;    /* n_nodes, triangle count of a leaf (ignored for inner nodes)   */
    const uint32_t* obj_root;      /* n_objects, root node of each object's tree                    */
    /* triangles: Triangle::pointOne/Two/Three (Object.h:17-19) as raw homogeneous xyzw             */
    const float*    tri_points;    /* n_tris x 3 x 4                                                */
    const int32_t*  tri_obj;       /* n_tris, owning object                                         */
    const int32_t*  tri_tex;       /* n_tris, texture id or -1 (Triangle::textureName empty)        */
    const float*    tri_texcoord;  /* n_tris x 6, colorOne/Two/ThreeCoordinate (Object.h:23-25):    */
                                   /* integer texel coords stored as floats; may be NULL if no tex  */
    const float*    tri_normals;   /* n_tris x 9 vertex normals (Object.h:20-22) or NULL; only read */
                                   /* with SRT_FLAG_SMOOTH_NORMALS                                   */
    /* objects: objColors / objProperties (Object.h:64,68) */
    const float*    obj_color;     /* n_objects x 3                                                 */
    const float*    obj_material;  /* n_objects x 3: ambient, specularStrength, shininess           */
    /* textures: textureData / textureDimensions (Object.h:70-71), 8-bit RGB rows top-down          */
    const uint8_t*  tex_rgb;       /* concatenated                                                  */
    const uint64_t* tex_off;       /* n_textures byte offsets into tex_rgb                          */
    const uint32_t* tex_w;         /* n_textures                                                    */
    const uint32_t* tex_h;         /* n_textures                                                    */
} srt_scene_desc;

/* ---- render parameters: every compile-time literal of the reference path, as data -------------- */
enum {
    SRT_FLAG_NONE           = 0,
    SRT_FLAG_SMOOTH_NORMALS = 1u << 0,  /* interpolateNormal (simple_raytracer.cpp:132-140) instead of the flat face
                                         * normal: the line the reference keeps commented out at :162; needs
                                         * srt_scene_desc.tri_normals.  The function is pinned by a reference KAT,
                                         * the images only by the oracle (the reference cannot render this mode) */
    SRT_FLAG_COUNT_WORK     = 1u << 1,  /* run the counting build: fills node/tri test counters      */
    SRT_FLAG_NO_TIMING      = 1u << 2,  /* record no HIP events: for launches captured into a hipGraph (an even number
                                         * of renders per graph keeps the two alternating counter sets in step)     */
    SRT_FLAG_FRAMES_IN_FLIGHT = 1u << 3 /* a HINT, results do not depend on it: this frame is one of several the caller keeps in
                                         * flight on the device (other streams, other handles).  Launches of a fixed number of
                                         * waves that pull work (the shadow rays of 16+ light samples) then keep only as many waves
                                         * as the frame's work can feed and leave the rest of the machine to the other frames; a
                                         * frame that has the device to itself wants every wave (1080p, 16 samples: 12-16 % faster
                                         * with the hint on four streams, 11-23 % slower with it on one)                  */
};

typedef struct srt_params {
    uint32_t width, height;        /* imageSize (simple_raytracer.cpp:773)                          */
    /* scanline blocks rendered by THIS call: blocks of block_rows rows; this call owns blocks
     * block_first, block_first+block_stride, ...  Output row r_local = k*block_rows + (y % block_rows)
     * for the k-th owned block.  Whole frame on one device: block_rows = height, first 0, stride 1. */
    uint32_t block_rows, block_first, block_stride;
    /* block_cols = 0: full-width scanline blocks, as above.  block_cols = C > 0 (C and block_rows multiples of 8): the frame is cut
     * into tiles of block_rows x C pixels and tile (bx, by) belongs to the call with (bx + by) % block_stride == block_first -- a
     * diagonal deal, so that every call owns tiles in every block row and every block column (expensive pixels cluster: tree crowns,
     * a bunny; whole-width rows spread them over 8 devices only coarsely).  Output: `height` rows x srt_cols_owned(p) columns; local
     * column xl of image row y is image column ((xl / C) * block_stride + (block_first + block_stride - (y / block_rows) %
     * block_stride) % block_stride) * C + xl % C, and local columns whose image column is >= width are padding (not written). */
    uint32_t block_cols;
    float    focal;                /* 400 (:506)                                                    */
    uint32_t n_lights;             /* lightAmount (:348,445)                                        */
    const float* light_pos;        /* n_lights x 3, host-accumulated staircase (:372-382)           */
    /* NULL = the reference's frame: the scene has been transformed into camera space (main() applies inverse(viewMatrix) to every
     * triangle each frame, simple_raytracer.cpp:558 etc.) and rays leave the origin.  Non-NULL = CAMERA MODE, an opt-in EXTENSION
     * (SURVEY.md s8 f1): the scene and the light stay where they are, the hierarchy is built once, and the camera moves instead --
     * 16 floats, column-major like glm::mat4, the matrix M that takes a camera-space ray into the scene's space (for the
     * reference's scenes: M = the viewMatrix whose inverse main() applies to the triangles).  Ray origin = M[3].xyz, direction =
     * (M[0] * dx + M[1] * dy) + M[2] * dz with (dx, dy, dz) the reference's (i, j, focal).  Same geometry, different rounding:
     * results are pinned by the oracle run in the same mode, not by the reference. */
    const float* ray_matrix;
    float    shadow_div;           /* 5   (:369)                                                    */
    float    reinhard;             /* 0.5 (:391)                                                    */
    float    gamma;                /* 1.1 (:396)                                                    */
    uint8_t  background[4];        /* 173,216,230 (:476); [3] unused                                */
    uint32_t spp;                  /* 1 = reference.  n*n > 1 = EXTENSION (the reference has no supersampling):
                                    * regular n x n sub-pixel grid, offsets (k+0.5)/n - 0.5 added to dir.xy, the
                                    * sub-frames' pre-tone-map sums added in order, divided by spp, tone-mapped
                                    * once; hit_id / t report sub-sample 0                            */
    uint32_t flags;                /* SRT_FLAG_* in bits 0..7; bits 8..15: kernel-variant selector for A/B measurements and
                                    * the parity tests (0 = the shipped pipeline; the others compute the same results with
                                    * older or differently configured kernels, DESIGN.md s5 lists them)                  */
} srt_params;

typedef struct srt_stats {
    uint64_t primary_rays;         /* width x owned rows x spp (of the last render)                 */
    uint64_t hit_rays;             /* primary rays that hit                                         */
    uint64_t shadow_rays;          /* hit_rays x n_lights (algorithmic count)         */
    /* work of the kernels' own traversal; SRT_FLAG_COUNT_WORK only, else 0 */
    uint64_t node_tests_primary;   /* slab tests in the closest-hit kernel                          */
    uint64_t tri_tests_primary;    /* Moller-Trumbore tests in the closest-hit kernel               */
    uint64_t node_tests_shadow;    /* slab tests in the shadow/shade kernel                         */
    uint64_t tri_tests_shadow;     /* Moller-Trumbore tests in the shadow/shade kernel              */
    /* HIP-event times on the launch stream, AVERAGED over the `launches` renders since the previous
     * srt_sync (at most 64 are kept; older ones are dropped from the average) */
    float    ms_primary;           /* closest-hit kernel                                            */
    float    ms_shadow;            /* shadow-ray kernel                                             */
    float    ms_shade;             /* shading kernel                                                */
    float    ms_total;             /* first launch -> last kernel done                              */
    uint32_t launches;
    uint32_t rows;                 /* rows written by the last render                               */
} srt_stats;

typedef struct srt_scene srt_scene;     /* opaque: device-resident flat scene + workspace           */

/* Fill p with the reference's literals for a WxH frame and one light at (lx,ly,lz) rendered
 * whole on one device.  light_pos is left NULL: point it at a table (srt_light_staircase). */
void srt_params_default(srt_params* p, uint32_t width, uint32_t height);

/* The soft-shadow light table of softShadow (simple_raytracer.cpp:363-383): sample 0 = base, then
 * x, y, z, x, ... += 3.0f accumulated in f32 exactly as the reference does.  out = n x 3. */
void srt_light_staircase(const float base[3], uint32_t n, float* out);

/* Number of rows a call with these params writes (<= height) and the width of those rows (width unless block_cols > 0). */
uint32_t srt_rows_owned(const srt_params* p);
uint32_t srt_cols_owned(const srt_params* p);

/* Validate + upload a flat scene to HIP device `device`.  The descriptor's arrays are only read
 * during the call. */
int srt_scene_create(int device, const srt_scene_desc* desc, srt_scene** out);
int srt_scene_destroy(srt_scene* s);

/* Another handle on the SAME device records (no copy of the geometry; they are freed with the last handle): its own workspace,
 * counters, statistics and stream, so that several frames of one static scene -- other lights, another camera (ray_matrix),
 * another share of the frame -- can be in flight at once (streams, srt_render_device_batch) while the records stay hot in L2 /
 * Infinity Cache once.  srt_scene_update through any of the handles rewrites the records all of them read; it is ordered on THAT
 * call's stream only, so the other handles must not have renders in flight and must not render before it has completed (e.g. a
 * render enqueued behind it on the same stream has been waited for). */
int srt_scene_share(srt_scene* src, srt_scene** out);

/* The next frame's geometry into the SAME device allocations (the reference re-transforms and rebuilds everything per frame,
 * simple_raytracer.cpp:534-618): `desc` must have the counts of the scene's current contents (n_objects, n_nodes, n_tris,
 * n_textures, the same triangles textured / with normals) and the texture table of the uploaded scene (tex_off, tex_w, tex_h), else
 * SRT_ERR_LAYOUT -- create a new scene then.  Texture images are uploaded again only when their bytes differ from what the device
 * holds (a 64-bit content hash is compared; they rarely change between frames); tri_tex / tri_texcoord always are.  Asynchronous on `stream`: the copies are
 * ordered behind the renders already enqueued there (NULL = the scene's own stream, the one srt_render / srt_render_async use);
 * the descriptor's arrays are only read during the call. */
int srt_scene_update(srt_scene* s, const srt_scene_desc* desc, void* stream);

/* ---- f1, the device half of the per-frame rebuild (Object.cpp:183-190 transform, :205-221 boxes, :225-284 build) ----
 * The reference re-transforms every triangle and rebuilds every hierarchy per frame.  What the HOST must keep doing for an exact
 * result is the build's std::sort (its order of equal keys is libstdc++'s; tests/test_host_mirror.py) and, on the way, the node boxes
 * that choose each split axis.  Everything else of a frame's scene is derived ON THE DEVICE from what that build leaves behind:
 *   the transformed points in SOURCE order (as the host holds them: object by object, inside an object in load order),
 *   the permutation the build leaves (visit order -> source index inside the object), and the node boxes in DFS pre-order.
 * The device gathers the triangles into visit order, derives the ray-independent triangle records (P1 = p1 / w, e1, e2, the face
 * normal, tvec and qvec of rays from the origin -- the same IEEE operations srt_scene_create runs on the host, bit for bit), permutes
 * the per-triangle attributes, and writes the boxes into the node records.  Per frame the host sends 52 bytes a triangle and 24 a node
 * instead of flattening, deriving and copying ~130 bytes a triangle.
 * Contract: the scene was created (srt_scene_create) from hierarchies of the same SHAPE -- same objects in the same order, same
 * triangle count per object (the reference's builder halves while a node holds more than 8 triangles, so the shape depends on the
 * count alone) -- and only positions, order and boxes change.  Counts are checked (SRT_ERR_LAYOUT), the shape cannot be.          */
typedef struct srt_frame_geometry {
    uint32_t n_objects;
    const uint32_t*        obj_n_tris;     /* n_objects: triangles per object (checked against the scene)                      */
    const uint32_t*        obj_n_nodes;    /* n_objects: nodes per object (checked against the scene)                          */
    const float* const*    obj_points;     /* per object: n_tris x 3 x 4 transformed points (Triangle::pointOne/Two/Three), SOURCE order */
    const uint32_t* const* obj_order;      /* per object: n_tris; visit-order triangle i of the object is its source triangle obj_order[k][i] */
    const float* const*    obj_node_min;   /* per object: n_nodes x 3, the object's nodes in DFS pre-order (root first, left subtree, right subtree) */
    const float* const*    obj_node_max;
    const float*           obj_color;      /* n_objects x 3 or NULL (unchanged)                                                */
    const float*           obj_material;   /* n_objects x 3 or NULL (unchanged)                                                */
} srt_frame_geometry;

/* Per-triangle attributes in SOURCE order (objects concatenated in the scene's order), set once: srt_scene_update_frame permutes
 * them into each frame's visit order.  tri_texcoord: n_tris x 6 (needed if the scene has textured triangles), tri_normals: n_tris x 9
 * or NULL (needed if the scene was created with normals), tri_tex: n_tris texture ids or NULL (= none textured).  A later
 * srt_scene_update (a whole new flat scene) discards them: set them again before the next srt_scene_update_frame.                 */
int srt_scene_set_source(srt_scene* s, const float* tri_texcoord, const float* tri_normals, const int32_t* tri_tex);

/* The next frame's geometry, derived on the device (see above).  Asynchronous on `stream` (NULL = the scene's own stream), ordered
 * behind the renders already enqueued there; the arrays are only read during the call. */
int srt_scene_update_frame(srt_scene* s, const srt_frame_geometry* g, void* stream);

/* Render into DEVICE buffers (rows = srt_rows_owned(p)); any output pointer may be NULL.
 *   d_hit_id     rows x W   int32   canonical triangle id, -1 = miss
 *   d_t          rows x W   f32     closest-hit distance (+inf on miss)
 *   d_rgb_linear rows x W x 3 f32   pre-tone-map light-sample sum (softShadow:362-383); 0 on miss
 *   d_rgb8       rows x W x 3 u8    tone-mapped, quantised (:391-398,447-449), black -> background
 *                                   (:518, drawImage:476-487)
 * Asynchronous on `stream` (a hipStream_t, NULL = default stream); stats->ms_* are valid after
 * srt_sync().  */
int srt_render_device(srt_scene* s, const srt_params* p, void* stream,
                      int32_t* d_hit_id, float* d_t, float* d_rgb_linear, uint8_t* d_rgb8);

/* The frames of a step (the reference's main() renders a 36-frame orbit, simple_raytracer.cpp:534) in ONE set of launches:
 * frame i = srt_render_device(scenes[i], &params[i], stream, d_hit_id[i], ...), with bitwise the same outputs.  The handles must
 * be n DISTINCT handles on one device (a handle's workspace serves one frame at a time; srt_scene_share gives n handles on one
 * copy of a scene); each output table may be NULL, and so may its entries.  Frames that take the default pipeline at one common
 * size share the launches -- those with 1..7 light samples one pair of launches, those with 8 and more three -- which fills the
 * chip where one frame, or the eighth of it one of eight GPUs owns, does not (a silhouette tile occupies its workgroup for the
 * better part of such a launch); any other frame is launched on its own as srt_render_device would.  The frames' arguments travel
 * by value with the launches (up to 36 frames a launch, more frames = more launches): nothing is allocated or copied, and the call
 * may be captured into a hipGraph like any other (ABI version 3; earlier versions kept argument tables in device memory and could not
 * make one while capturing).  Frames with 16 and more light samples: the shadow-ray launch of a call remembers which 4x4-pixel
 * quadrants had long walks and the next call on the same handles deals those early -- order only, results do not depend on it
 * (SRT_HEAVY_STEPS=0 in the environment turns it off; single frames without SRT_FLAG_FRAMES_IN_FLIGHT do the same).  No per-frame times: srt_sync()'s ms_* keep the values of the last timed
 * render of each handle. */
int srt_render_device_batch(uint32_t n, srt_scene* const* scenes, const srt_params* params, void* stream,
                            int32_t* const* d_hit_id, float* const* d_t, float* const* d_rgb_linear, uint8_t* const* d_rgb8);

/* Same, into HOST buffers: allocates nothing per call beyond the scene's workspace, copies back,
 * synchronises.  stats may be NULL. */
int srt_render(srt_scene* s, const srt_params* p,
               int32_t* hit_id, float* t, float* rgb_linear, uint8_t* rgb8, srt_stats* stats);

/* srt_render without the wait: the kernels and the copies into the host buffers are enqueued on the scene's own stream and the
 * call returns; srt_sync() waits.  With buffers from srt_host_alloc (pinned memory) the copies are asynchronous and run at full
 * PCIe rate; with ordinary memory they still work.  srt_scene_update(scene, desc, NULL) is ordered on the same stream, so
 * "update, render_async, (build the next frame on the CPU), sync" keeps the GPU work off the host's critical path. */
int srt_render_async(srt_scene* s, const srt_params* p, int32_t* hit_id, float* t, float* rgb_linear, uint8_t* rgb8);
void* srt_host_alloc(size_t bytes);        /* pinned host memory, NULL on failure */
void  srt_host_free(void* p);

/* Wait for the last srt_render_device / srt_render_async on this scene and collect its stats. */
int srt_sync(srt_scene* s, srt_stats* stats);

/* Device-resident size of the scene records and the per-record algorithmic byte sizes used by
 * the bytes model: 32 B per node test, 36 B per triangle test. */
uint64_t srt_scene_device_bytes(const srt_scene* s);

/* Which kernels the last srt_render* on this scene launched, in order, e.g. "k_trace_nq+k_shade_tile" (DESIGN.md s5 explains
 * how the pipeline is chosen per scene and light-sample count).  The string lives as long as the scene. */
const char* srt_scene_pipeline(const srt_scene* s);

/* Expected slab tests per ray from the surface areas of the scene's node boxes (what decides whether primary rays take the
 * packet walk: hierarchies of heavily overlapping boxes do). */
double srt_scene_overlap_estimate(const srt_scene* s);

/* ---- known-answer entry points: the DEVICE leaf functions on caller vectors (host pointers), so that the
 * reference's known-answer fixtures pin the device code directly.  Layouts as in tests/golden/kat.npz:
 * ray_od = n x (origin xyz, direction xyz); box = n x (min xyz, max xyz); tri_points = n x 3 x xyzw.
 *   srt_kat_ray_aabb      intersectRayAabbNoOrigin (simple_raytracer.cpp:252-293): the literal form, the
 *                         branch-free form and the filtered form (+ its "ambiguous, use the exact form" flag)
 *   srt_kat_ray_triangle  rayTriangleIntersection (:42-75): t, -inf = miss
 *   srt_kat_phong         phongIllumination (:144-200): in28 = ray_od(6) tri(12) light(3) colour(3) ka ks shin t
 *   srt_kat_tonemap       Reinhard + gamma (:391-398) and the quantiser (:447-449)                         */
int srt_kat_ray_aabb(int device, uint32_t n, const float* ray_od, const float* box, uint8_t* exact, uint8_t* branchless,
                     uint8_t* filtered, uint8_t* ambiguous);
int srt_kat_ray_triangle(int device, uint32_t n, const float* ray_od, const float* tri_points, float* t);
int srt_kat_phong(int device, uint32_t n, const float* in28, float* rgb);
int srt_kat_interp_normal(int device, uint32_t n, const float* in12 /* 3 normals + barycentrics */, float* out3);   /* interpolateNormal :132-140 */
int srt_kat_pow(int device, uint32_t n, const float* x, const float* y, float* fast, float* lib);   /* the device powf: shipped form vs (float)pow(double) */
int srt_kat_tonemap(int device, uint32_t n, const float* lin, float reinhard, float gamma, float* tone, int32_t* q);

/* Measurement hook: the chip's VALU issue rate from independent v_fma_f32 streams at 8 waves per SIMD (the yardstick bench.py's
 * roofline prices the kernels' VALU work against).  out[0] = wave-instructions one SIMD issues per cycle (MI355X: SIMD-32, a wave64
 * instruction over 2 cycles -> 0.5), out[1] = shader clock in GHz during the run, out[2] = the same rate over the whole launch span. */
int srt_debug_valu_rate(int device, uint32_t iters, double* out4);      /* out[3] = waves that shared a SIMD (median) */

/* Test hook: the device records of a scene copied back (any pointer may be NULL): nodes n_nodes x 32 B, tris / tris_o n_tris x 48 B, wide
 * (n_nodes - n_objects) / 2 x 64 B, root_nodes n_objects x 32 B, tri_texcoord n_tris x 6 floats, tri_normals n_tris x 9 floats, tri_tex
 * n_tris ids (the last three only where the scene has them).  Waits for the scene's pending work first.  What srt_scene_update_frame
 * derives on the device is compared, byte for byte, with what srt_scene_create derives on the host. */
int srt_debug_scene_records(srt_scene* s, void* nodes, void* tris, void* tris_o, void* wide, void* root_nodes,
                            float* tri_texcoord, float* tri_normals, int32_t* tri_tex);

/* Test hook: the next n host allocations made on behalf of a caller fail (std::bad_alloc inside the library), so that
 * the SRT_ERR_OOM path can be exercised without exhausting memory.  Not for production use. */
void srt_debug_fail_host_allocs(int n);

const char* srt_strerror(int code);
int         srt_last_hip_error(void);
uint32_t    srt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SRT_H */
