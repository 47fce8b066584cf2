/* oracle/srt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's per-pixel ray-trace path, written from the algorithm
 * (not from the reference's text): every function cites the reference lines it follows
 * (paths relative to /root/reference).  Float expression trees follow GLM 1.0.2's scalar code
 * paths (library/glm-master/glm/detail/func_geometric.inl:48-55,73-83,98-105,120-125).
 * Must be compiled with -ffp-contract=off (oracle/Makefile): FMA contraction changes hit results.
 *
 * Pinned by tests/test_oracle_golden.py against tests/golden/ (vectors produced by the compiled
 * reference through oracle/ref_harness.cpp).
 */
#include "srt_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static inline v3 v3make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 v3sub(v3 a, v3 b) { return v3make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3add(v3 a, v3 b) { return v3make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3scale(v3 a, float s) { return v3make(a.x * s, a.y * s, a.z * s); }
static inline v3 v3neg(v3 a) { return v3make(-a.x, -a.y, -a.z); }
/* glm dot(vec3): tmp = a*b; (tmp.x + tmp.y) + tmp.z        (func_geometric.inl:48-55) */
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/* glm cross                                                 (func_geometric.inl:73-83) */
static inline v3 cross3(v3 x, v3 y) {
    return v3make(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
/* glm normalize = v * inversesqrt(dot(v,v)), inversesqrt = 1/sqrt  (func_geometric.inl:98-105,
 * func_exponential.inl:134-139) */
static inline v3 normalize3(v3 v) { float s = 1.0f / sqrtf(dot3(v, v)); return v3scale(v, s); }
/* glm max(x,y) = (x < y) ? y : x */
static inline float glm_max(float x, float y) { return (x < y) ? y : x; }

/* ---- a5: rayTriangleIntersection, simple_raytracer.cpp:42-75 -------------------------------- */
typedef struct { v3 p1, e1, e2; } tri_geom;

/* Ray-independent prefix of :45-51 (w-divide, edges) */
static inline tri_geom tri_geom_from_points(const float* p /* 3 x xyzw */) {
    tri_geom g;
    v3 P1 = v3make(p[0] / p[3], p[1] / p[3], p[2]  / p[3]);
    v3 P2 = v3make(p[4] / p[7], p[5] / p[7], p[6]  / p[7]);
    v3 P3 = v3make(p[8] / p[11], p[9] / p[11], p[10] / p[11]);
    g.p1 = P1; g.e1 = v3sub(P2, P1); g.e2 = v3sub(P3, P1);
    return g;
}

static inline float ray_triangle(v3 o, v3 d, const tri_geom* g) {
    v3 pvec = cross3(d, g->e2);                       /* :54 */
    float det = dot3(g->e1, pvec);                    /* :55 */
    if (fabsf(det) < 1e-12f) return -INFINITY;        /* :57 */
    float inv = 1.0f / det;                           /* :60 */
    v3 tvec = v3sub(o, g->p1);                        /* :62 */
    float u = dot3(tvec, pvec) * inv;                 /* :64 */
    if (u < 0.0f || u > 1.0f) return -INFINITY;       /* :65 */
    v3 qvec = cross3(tvec, g->e1);                    /* :66 */
    float v = dot3(d, qvec) * inv;                    /* :67 */
    if (v < 0.0f || u + v > 1.0f) return -INFINITY;   /* :68 */
    float t = dot3(g->e2, qvec) * inv;                /* :70 */
    if (t < 0.0f) return -INFINITY;                   /* :73 */
    return t;                                         /* t == 0 is a hit; NaN falls through */
}

/* ---- a4: intersectRayAabbNoOrigin, simple_raytracer.cpp:252-293 ------------------------------ */
static inline int ray_aabb(v3 o, v3 d, const float* mn, const float* mx) {
    float minX = (mn[0] - o.x) / d.x, maxX = (mx[0] - o.x) / d.x;     /* :256-257 */
    if (minX > maxX) { float s = minX; minX = maxX; maxX = s; }       /* :258-260 */
    float minY = (mn[1] - o.y) / d.y, maxY = (mx[1] - o.y) / d.y;     /* :262-263 */
    if (minY > maxY) { float s = minY; minY = maxY; maxY = s; }
    if (maxX < minY || maxY < minX) return 0;                         /* :269 */
    if (minY > minX) minX = minY;                                     /* :274 */
    if (maxY < maxX) maxX = maxY;                                     /* :277 */
    float minZ = (mn[2] - o.z) / d.z, maxZ = (mx[2] - o.z) / d.z;     /* :281-282 */
    if (minZ > maxZ) { float s = minZ; minZ = maxZ; maxZ = s; }
    if ((minX > maxZ) || (minZ > maxX)) return 0;                     /* :288 */
    return 1;
}

/* ---- a8b: calculateBarycentricCoords :79-117 -------------------------------------------------- */
static inline v3 barycentric(const tri_geom* g, v3 point) {
    v3 v0 = g->e1, v1 = g->e2, v2 = v3sub(point, g->p1);   /* :89-91 */
    float d00 = dot3(v0, v0), d01 = dot3(v0, v1), d11 = dot3(v1, v1);
    float d20 = dot3(v2, v0), d21 = dot3(v2, v1);
    float denom = d00 * d11 - d01 * d01;                   /* :105 */
    float v = (d11 * d20 - d01 * d21) / denom;             /* :109 */
    float w = (d00 * d21 - d01 * d20) / denom;             /* :110 */
    float u = 1.0f - v - w;                                /* :112 */
    return v3make(u, v, w);
}

/* ---- calculateTriangleNormal :32-37: raw xyz differences (no w-divide) ------------------------ */
static inline v3 face_normal(const float* p) {
    v3 a = v3make(p[4] - p[0], p[5] - p[1], p[6] - p[2]);
    v3 b = v3make(p[8] - p[0], p[9] - p[1], p[10] - p[2]);
    return normalize3(cross3(a, b));
}

/* ---- interpolateNormal :132-140 (smooth shading; the reference's call at :162 is commented out, so this is
 * the opt-in SRT_FLAG_SMOOTH_NORMALS mode): normalize(bc.x*n1 + bc.y*n2 + bc.z*n3) */
static inline v3 interp_normal(const float* n9, v3 bc) {
    v3 n = v3make((bc.x * n9[0] + bc.y * n9[3]) + bc.z * n9[6], (bc.x * n9[1] + bc.y * n9[4]) + bc.z * n9[7],
                  (bc.x * n9[2] + bc.y * n9[5]) + bc.z * n9[8]);
    return normalize3(n);
}

/* ---- a8: phongIllumination :144-200 (lightColor = (1,1,1), :433) ------------------------------ */
static inline v3 phong(v3 n, v3 o, v3 d, v3 L, v3 objColor, float ka, float ks, float shin, float t) {
    const float rView = 1.0f / 3.14159265358979323846264338327950288f;   /* :153, glm::pi<float>() */
    const v3 lightColor = { 1.0f, 1.0f, 1.0f };
    v3 P = v3add(o, v3scale(d, t));                          /* :156  origin + distance*direction */
    v3 l = normalize3(v3sub(L, P));                          /* :166 */
    float dp = dot3(n, l);                                   /* :174 */
    if (dp < 0.0f) dp = -dp;                                 /* :175-177 */
    float m = glm_max(dp, 0.0f);
    v3 diffuse = v3make(((rView * objColor.x) * lightColor.x) * m,
                        ((rView * objColor.y) * lightColor.y) * m,
                        ((rView * objColor.z) * lightColor.z) * m);          /* :178 */
    float ak = rView * ka;
    v3 ambient = v3make((ak * objColor.x) * lightColor.x, (ak * objColor.y) * lightColor.y,
                        (ak * objColor.z) * lightColor.z);                   /* :184 */
    v3 v = normalize3(v3neg(d));                             /* :190 */
    v3 I = v3neg(l);                                         /* reflect(-l, n) = I - n*dot(n,I)*2  :191 */
    float ndi = dot3(n, I);
    v3 r = v3make(I.x - (n.x * ndi) * 2.0f, I.y - (n.y * ndi) * 2.0f, I.z - (n.z * ndi) * 2.0f);
    float sp = powf(glm_max(dot3(r, v), 0.0f), shin);        /* :196 */
    v3 specular = v3make(((lightColor.x * ks) * m) * sp, ((lightColor.y * ks) * m) * sp,
                         ((lightColor.z * ks) * m) * sp);
    return v3make((diffuse.x + specular.x) + ambient.x, (diffuse.y + specular.y) + ambient.y,
                  (diffuse.z + specular.z) + ambient.z);     /* :199 */
}

/* ---- a9: Reinhard + gamma (:391-398) and quantiser (:447-449) --------------------------------- */
static inline float tone1(float c, float reinhard, float gamma) {
    c = c / (c + reinhard);
    return powf(c, gamma);
}
/* int(c*255): truncation; c in [0,1) for every finite non-negative sum.  NaN / out of range is
 * undefined behaviour in the reference (x86 yields 0 in the low byte): defined here as clamp, NaN->0. */
static inline int32_t quant1(float c) {
    float s = c * 255.0f;
    if (!(s > 0.0f)) return 0;
    if (s >= 255.0f) return 255;
    return (int32_t)s;
}

void oracle_light_staircase(const float base[3], uint32_t n, float* out) {
    /* softShadow:363-383: sample i uses the current position, then axis (i%3) += 3.0f */
    float L[3] = { base[0], base[1], base[2] };
    for (uint32_t i = 0; i < n; i++) {
        out[i * 3] = L[0]; out[i * 3 + 1] = L[1]; out[i * 3 + 2] = L[2];
        L[i % 3] += 3.0f;
    }
}

/* include/srt.h srt_params.block_cols: tiles of block_rows x block_cols pixels, tile (bx, by) owned iff (bx + by) % stride == first */
uint32_t oracle_cols_owned(const srt_params* p) {
    if (!p->block_stride) return 0;
    if (!p->block_cols) return p->width;
    uint32_t n_bx = (p->width + p->block_cols - 1) / p->block_cols;
    return (n_bx + p->block_stride - 1) / p->block_stride * p->block_cols;
}
static uint32_t image_col(const srt_params* p, uint32_t xl, uint32_t y) {
    if (!p->block_cols) return xl;
    uint32_t off = (p->block_first + p->block_stride - (y / p->block_rows) % p->block_stride) % p->block_stride;
    return ((xl / p->block_cols) * p->block_stride + off) * p->block_cols + xl % p->block_cols;
}

uint32_t oracle_rows_owned(const srt_params* p) {
    if (!p->block_rows || !p->block_stride) return 0;
    if (p->block_cols) return p->block_first < p->block_stride ? p->height : 0;
    uint32_t nblocks = (p->height + p->block_rows - 1) / p->block_rows, rows = 0;
    for (uint32_t b = p->block_first; b < nblocks; b += p->block_stride) {
        uint32_t y0 = b * p->block_rows, y1 = y0 + p->block_rows;
        if (y1 > p->height) y1 = p->height;
        rows += y1 - y0;
    }
    return rows;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- scene view with ray-independent per-triangle prefixes precomputed ------------------------- */
typedef struct {
    const srt_scene_desc* d;
    tri_geom* geom;     /* n_tris */
    v3* normal;         /* n_tris */
} scene_view;

typedef struct { uint64_t node_tests, tri_tests; } work_ctr;   /* one per kernel: primary, shadow */

/* a3: boundingBoxIntersection :296-317 fused with the closest-hit loop of rayIntersection :424-431.
 * DFS left-first; every leaf whose ancestors all pass the slab test contributes its triangles in
 * stored order; strict '<' keeps the first of equal t. */
static void closest_in_tree(const scene_view* s, int32_t node, v3 o, v3 d, float* best, int32_t* best_id, work_ctr* w) {
    const srt_scene_desc* sd = s->d;
    w->node_tests++;
    if (!ray_aabb(o, d, sd->node_min + 3 * (size_t)node, sd->node_max + 3 * (size_t)node)) return;
    int32_t l = sd->node_left[node], r = sd->node_right[node];
    if (l < 0 && r < 0) {
        int32_t first = sd->node_first[node], cnt = sd->node_count[node];
        for (int32_t k = 0; k < cnt; k++) {
            w->tri_tests++;
            float t = ray_triangle(o, d, &s->geom[first + k]);
            if (t != -INFINITY && t < *best) { *best = t; *best_id = first + k; }   /* :428-431 */
        }
        return;
    }
    closest_in_tree(s, l, o, d, best, best_id, w);
    closest_in_tree(s, r, o, d, best, best_id, w);
}

/* a6: shadowIntersection :321-342 for one object's tree: any candidate with MT != -inf (NaN too). */
static int anyhit_in_tree(const scene_view* s, int32_t node, v3 o, v3 d, work_ctr* w) {
    const srt_scene_desc* sd = s->d;
    w->node_tests++;
    if (!ray_aabb(o, d, sd->node_min + 3 * (size_t)node, sd->node_max + 3 * (size_t)node)) return 0;
    int32_t l = sd->node_left[node], r = sd->node_right[node];
    if (l < 0 && r < 0) {
        int32_t first = sd->node_first[node], cnt = sd->node_count[node];
        for (int32_t k = 0; k < cnt; k++) {
            w->tri_tests++;
            float t = ray_triangle(o, d, &s->geom[first + k]);
            if (t != -INFINITY) return 1;                                           /* :335 */
        }
        return 0;
    }
    if (anyhit_in_tree(s, l, o, d, w)) return 1;
    return anyhit_in_tree(s, r, o, d, w);
}

/* shadowIntersection :321-342.  The reference also walks the hit object's own tree and discards
 * the result (:328 before :331); skipping it is result-neutral and is what is counted. */
static int in_shadow(const scene_view* s, int32_t self_obj, v3 L, float t, v3 d, work_ctr* w, int cam, v3 o) {
    v3 dt = v3scale(d, t);               /* ray.direction * fDistance            :325-326 */
    if (cam) dt = v3add(o, dt);          /* camera mode (extension): the hit point of a ray that does not start at 0 */
    v3 sd = v3sub(L, dt), so = dt;
    for (uint32_t k = 0; k < s->d->n_objects; k++) {
        if ((int32_t)k == self_obj) continue;
        if (anyhit_in_tree(s, (int32_t)s->d->obj_root[k], so, sd, w)) return 1;
    }
    return 0;
}

int oracle_render(const srt_scene_desc* d, const srt_params* p,
                  int32_t* hit_id, float* t_out, float* rgb_linear, float* rgb_tone, uint8_t* rgb8,
                  srt_stats* stats, int n_threads) {
    if (!d || !p || !p->width || !p->height || !p->block_rows || !p->block_stride) return SRT_ERR_ARG;
    if (p->n_lights && !p->light_pos) return SRT_ERR_ARG;
    const uint32_t spp = p->spp, spp_n = (uint32_t)lroundf(sqrtf((float)p->spp));
    if (spp < 1 || spp_n * spp_n != spp) return SRT_ERR_ARG;      /* supersampling extension: n x n sub-pixel grid */
    const int smooth = (p->flags & SRT_FLAG_SMOOTH_NORMALS) != 0;
    if (smooth && !d->tri_normals) return SRT_ERR_ARG;
    scene_view s; s.d = d;
    s.geom = (tri_geom*)malloc(sizeof(tri_geom) * (d->n_tris ? d->n_tris : 1));
    s.normal = (v3*)malloc(sizeof(v3) * (d->n_tris ? d->n_tris : 1));
    for (uint32_t i = 0; i < d->n_tris; i++) {
        s.geom[i] = tri_geom_from_points(d->tri_points + 12 * (size_t)i);
        s.normal[i] = face_normal(d->tri_points + 12 * (size_t)i);
    }
    const uint32_t W = p->width, H = p->height;
    const uint32_t rows = oracle_rows_owned(p), WL = oracle_cols_owned(p);      /* WL: width of the rows this call writes */
    /* row table: local row -> image y */
    uint32_t* row_y = (uint32_t*)malloc(sizeof(uint32_t) * (rows ? rows : 1));
    if (p->block_cols) { for (uint32_t r = 0; r < rows; r++) row_y[r] = r; }
    else {
        uint32_t nblocks = (H + p->block_rows - 1) / p->block_rows, r = 0;
        for (uint32_t b = p->block_first; b < nblocks; b += p->block_stride)
            for (uint32_t y = b * p->block_rows; y < (b + 1) * p->block_rows && y < H; y++) row_y[r++] = y;
    }
    uint64_t n_pixels = 0;
    uint64_t hits = 0, node_tests = 0, tri_tests = 0, snode_tests = 0, stri_tests = 0;
    /* sendRaysAndIntersectPointsColors:511-517: i = px + int(-W/2), dir = (i, j, focal), origin 0 */
    const int i0 = (int)(-(float)W / 2), j0 = (int)(-(float)H / 2);
    const int cam = p->ray_matrix != NULL;       /* camera mode (extension, include/srt.h): rays go into the scene's space */
    const float* M = p->ray_matrix;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads) reduction(+ : hits, node_tests, tri_tests, snode_tests, stri_tests, n_pixels)
#endif
    for (uint32_t r = 0; r < rows; r++) {
        const uint32_t y = row_y[r];
        work_ctr w = { 0, 0 }, ws = { 0, 0 };
        for (uint32_t xl = 0; xl < WL; xl++) {
            const uint32_t x = image_col(p, xl, y);                 /* local column -> image column */
            if (x >= W) continue;                                    /* padding of a tile deal: not written */
            n_pixels++;
            const size_t pix = (size_t)r * WL + xl;
            v3 o = cam ? v3make(M[12], M[13], M[14]) : v3make(0.0f, 0.0f, 0.0f);
            v3 sum = v3make(0.f, 0.f, 0.f), tone = v3make(0.f, 0.f, 0.f);
            int32_t q[3] = { 0, 0, 0 };
            for (uint32_t ss = 0; ss < spp; ss++) {
            /* spp = 1: offset 0, the reference's ray (i + rayXY.x with rayXY = 0, :507,514-515); spp > 1 (extension,
             * SURVEY.md R4): regular n x n sub-pixel grid, offsets (k + 0.5)/n - 0.5 */
            const float sub_x = spp == 1 ? 0.0f : ((float)(ss % spp_n) + 0.5f) / (float)spp_n - 0.5f;
            const float sub_y = spp == 1 ? 0.0f : ((float)(ss / spp_n) + 0.5f) / (float)spp_n - 0.5f;
            v3 dir = v3make((float)(i0 + (int)x) + sub_x, (float)(j0 + (int)y) + sub_y, p->focal);
            if (cam) dir = v3make((M[0] * dir.x + M[4] * dir.y) + M[8] * dir.z, (M[1] * dir.x + M[5] * dir.y) + M[9] * dir.z,
                                  (M[2] * dir.x + M[6] * dir.y) + M[10] * dir.z);
            float best = INFINITY; int32_t best_id = -1;
            for (uint32_t k = 0; k < d->n_objects; k++)                      /* rayIntersection:409 */
                closest_in_tree(&s, (int32_t)d->obj_root[k], o, dir, &best, &best_id, &w);
            if (ss == 0) {
                if (hit_id) hit_id[pix] = best_id;
                if (t_out) t_out[pix] = best;
            }
            v3 ssum = v3make(0.f, 0.f, 0.f);
            if (best_id >= 0) {
                hits++;
                const int32_t obj = d->tri_obj[best_id];
                v3 color = v3make(d->obj_color[obj * 3], d->obj_color[obj * 3 + 1], d->obj_color[obj * 3 + 2]);   /* :437-440 */
                const int32_t tex = d->tri_tex ? d->tri_tex[best_id] : -1;
                if (tex >= 0) {                                                 /* softShadow:350-361 */
                    v3 P = v3add(o, v3scale(dir, best));
                    v3 bc = barycentric(&s.geom[best_id], P);
                    const float* tc = d->tri_texcoord + 6 * (size_t)best_id;
                    float tx = (bc.x * tc[0] + bc.y * tc[2]) + bc.z * tc[4];   /* getTextureCoordinate:123-125 */
                    float ty = (bc.x * tc[1] + bc.y * tc[3]) + bc.z * tc[5];
                    long long texIndex = ((long long)((int)ty * (int)d->tex_w[tex] + (int)tx)) * 3;   /* :357 */
                    /* outside the image the reference reads out of bounds (UB); clamped like the HIP path */
                    const long long last = (long long)d->tex_w[tex] * d->tex_h[tex] * 3 - 3;
                    if (texIndex < 0) texIndex = 0;
                    if (texIndex > last) texIndex = last;
                    const uint8_t* td = d->tex_rgb + d->tex_off[tex];
                    color = v3make(td[texIndex] / 255.0f, td[texIndex + 1] / 255.0f, td[texIndex + 2] / 255.0f);
                }
                const float ka = d->obj_material[obj * 3], ks = d->obj_material[obj * 3 + 1], sh = d->obj_material[obj * 3 + 2];
                v3 nrm = s.normal[best_id];
                if (smooth) {       /* phongIllumination:159,162 with the interpolateNormal line enabled */
                    v3 P = v3add(o, v3scale(dir, best));
                    nrm = interp_normal(d->tri_normals + 9 * (size_t)best_id, barycentric(&s.geom[best_id], P));
                }
                for (uint32_t l = 0; l < p->n_lights; l++) {                    /* softShadow:366-383 */
                    v3 L = v3make(p->light_pos[l * 3], p->light_pos[l * 3 + 1], p->light_pos[l * 3 + 2]);
                    int sh_hit = in_shadow(&s, obj, L, best, dir, &ws, cam, o);
                    v3 c = phong(nrm, o, dir, L, color, ka, ks, sh, best);
                    if (sh_hit) c = v3make(c.x / p->shadow_div, c.y / p->shadow_div, c.z / p->shadow_div);   /* :369 */
                    ssum = v3add(ssum, c);                                       /* :370 */
                }
            }
            sum = ss == 0 ? ssum : v3add(sum, ssum);
            }
            if (spp > 1) sum = v3make(sum.x / (float)spp, sum.y / (float)spp, sum.z / (float)spp);
            tone = v3make(tone1(sum.x, p->reinhard, p->gamma), tone1(sum.y, p->reinhard, p->gamma), tone1(sum.z, p->reinhard, p->gamma));
            q[0] = quant1(tone.x); q[1] = quant1(tone.y); q[2] = quant1(tone.z);
            if (rgb_linear) { rgb_linear[pix * 3] = sum.x; rgb_linear[pix * 3 + 1] = sum.y; rgb_linear[pix * 3 + 2] = sum.z; }
            if (rgb_tone) { rgb_tone[pix * 3] = tone.x; rgb_tone[pix * 3 + 1] = tone.y; rgb_tone[pix * 3 + 2] = tone.z; }
            if (rgb8) {
                /* :518 + drawImage:476-487: all-black (hit or not) -> background */
                if (q[0] == 0 && q[1] == 0 && q[2] == 0) { q[0] = p->background[0]; q[1] = p->background[1]; q[2] = p->background[2]; }
                rgb8[pix * 3] = (uint8_t)q[0]; rgb8[pix * 3 + 1] = (uint8_t)q[1]; rgb8[pix * 3 + 2] = (uint8_t)q[2];
            }
        }
        node_tests += w.node_tests; tri_tests += w.tri_tests;
        snode_tests += ws.node_tests; stri_tests += ws.tri_tests;
    }
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->primary_rays = n_pixels * spp;
        stats->hit_rays = hits;
        stats->shadow_rays = hits * p->n_lights;
        stats->node_tests_primary = node_tests;
        stats->tri_tests_primary = tri_tests;
        stats->node_tests_shadow = snode_tests;
        stats->tri_tests_shadow = stri_tests;
        stats->rows = rows;
    }
    free(row_y); free(s.geom); free(s.normal);
    return SRT_OK;
}

/* ---- batched leaf functions (KAT entry points) ------------------------------------------------- */
void oracle_ray_triangle(uint32_t n, const float* ray_od, const float* tri, float* t) {
    for (uint32_t i = 0; i < n; i++) {
        tri_geom g = tri_geom_from_points(tri + 12 * (size_t)i);
        const float* r = ray_od + 6 * (size_t)i;
        t[i] = ray_triangle(v3make(r[0], r[1], r[2]), v3make(r[3], r[4], r[5]), &g);
    }
}
void oracle_ray_aabb(uint32_t n, const float* ray_od, const float* box, uint8_t* hit) {
    for (uint32_t i = 0; i < n; i++) {
        const float* r = ray_od + 6 * (size_t)i;
        hit[i] = (uint8_t)ray_aabb(v3make(r[0], r[1], r[2]), v3make(r[3], r[4], r[5]), box + 6 * (size_t)i, box + 6 * (size_t)i + 3);
    }
}
void oracle_phong(uint32_t n, const float* in, float* rgb) {
    for (uint32_t i = 0; i < n; i++) {
        const float* q = in + 28 * (size_t)i;
        v3 c = phong(face_normal(q + 6), v3make(q[0], q[1], q[2]), v3make(q[3], q[4], q[5]), v3make(q[18], q[19], q[20]),
                     v3make(q[21], q[22], q[23]), q[24], q[25], q[26], q[27]);
        rgb[i * 3] = c.x; rgb[i * 3 + 1] = c.y; rgb[i * 3 + 2] = c.z;
    }
}
void oracle_barycentric(uint32_t n, const float* in, float* uvw) {
    for (uint32_t i = 0; i < n; i++) {
        const float* p = in + 15 * (size_t)i;
        tri_geom g = tri_geom_from_points(p);
        v3 b = barycentric(&g, v3make(p[12], p[13], p[14]));
        uvw[i * 3] = b.x; uvw[i * 3 + 1] = b.y; uvw[i * 3 + 2] = b.z;
    }
}
void oracle_interp_normal(uint32_t n, const float* in12, float* out3) {
    for (uint32_t i = 0; i < n; i++) {
        const float* q = in12 + 12 * (size_t)i;
        v3 r = interp_normal(q, v3make(q[9], q[10], q[11]));
        out3[i * 3] = r.x; out3[i * 3 + 1] = r.y; out3[i * 3 + 2] = r.z;
    }
}
void oracle_tonemap(uint32_t n, const float* lin, float reinhard, float gamma, float* tone, int32_t* q) {
    for (uint32_t i = 0; i < 3 * n; i++) { tone[i] = tone1(lin[i], reinhard, gamma); q[i] = quant1(tone[i]); }
}
