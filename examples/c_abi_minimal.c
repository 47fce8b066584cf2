/* examples/c_abi_minimal.c -- the C ABI of include/srt.h from plain C, no host mirror: a hand-written flat scene
 * (two objects, each a root with two leaves, the shape the reference's createBoundingHierarchy gives even a tiny object),
 * one render into host buffers, a PPM on stdout or an ASCII preview.
 *
 *   c_abi_minimal [width height [out.ppm]]
 *
 * Object 0: a quad (two triangles) floating at z = 300.  Object 1: a large ground quad below it (y = +120, y grows
 * downward as in the reference's image convention), so that the first quad's shadow falls on it.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/srt.h"

static void box_of(const float* pts, int n_tris, float* mn, float* mx) {
    for (int a = 0; a < 3; a++) { mn[a] = 3.4e38f; mx[a] = -3.4e38f; }
    for (int i = 0; i < n_tris * 3; i++)
        for (int a = 0; a < 3; a++) {
            const float v = pts[i * 4 + a];
            if (v < mn[a]) mn[a] = v;
            if (v > mx[a]) mx[a] = v;
        }
}

int main(int argc, char** argv) {
    const uint32_t W = argc > 2 ? (uint32_t)atoi(argv[1]) : 64, H = argc > 2 ? (uint32_t)atoi(argv[2]) : 32;
    /* triangles in visit order: object 0 (leaf, leaf), then object 1 (leaf, leaf); points are xyzw */
    static const float tri_points[4 * 12] = {
        /* object 0: quad x in [-60, 60], y in [-40, 20], z = 300 */
        -60, -40, 300, 1,   60, -40, 300, 1,   60, 20, 300, 1,
        -60, -40, 300, 1,   60, 20, 300, 1,   -60, 20, 300, 1,
        /* object 1: ground y = 120, x in [-400, 400], z in [100, 900] */
        -400, 120, 100, 1,   400, 120, 100, 1,   400, 120, 900, 1,
        -400, 120, 100, 1,   400, 120, 900, 1,   -400, 120, 900, 1,
    };
    static const int32_t tri_obj[4] = { 0, 0, 1, 1 };
    /* nodes: per object a root (inner) and two leaves of one triangle each */
    float node_min[6 * 3], node_max[6 * 3];
    static const int32_t node_left[6] = { 1, -1, -1, 4, -1, -1 }, node_right[6] = { 2, -1, -1, 5, -1, -1 };
    static const int32_t node_first[6] = { -1, 0, 1, -1, 2, 3 }, node_count[6] = { 0, 1, 1, 0, 1, 1 };
    static const uint32_t obj_root[2] = { 0, 3 };
    for (int ob = 0; ob < 2; ob++) {
        box_of(tri_points + ob * 24, 2, node_min + (ob * 3) * 3, node_max + (ob * 3) * 3);
        box_of(tri_points + ob * 24, 1, node_min + (ob * 3 + 1) * 3, node_max + (ob * 3 + 1) * 3);
        box_of(tri_points + ob * 24 + 12, 1, node_min + (ob * 3 + 2) * 3, node_max + (ob * 3 + 2) * 3);
    }
    static const float obj_color[6] = { 0.9f, 0.3f, 0.2f, 0.3f, 0.7f, 0.4f };
    static const float obj_material[6] = { 0.2f, 0.5f, 15.0f, 0.2f, 0.5f, 15.0f };   /* Object.cpp:31-34 defaults */

    srt_scene_desc d;
    memset(&d, 0, sizeof d);
    d.n_objects = 2; d.n_nodes = 6; d.n_tris = 4; d.n_textures = 0;
    d.node_min = node_min; d.node_max = node_max; d.node_left = node_left; d.node_right = node_right;
    d.node_first = node_first; d.node_count = node_count; d.obj_root = obj_root;
    d.tri_points = tri_points; d.tri_obj = tri_obj;
    d.obj_color = obj_color; d.obj_material = obj_material;

    srt_scene* scene = NULL;
    int rc = srt_scene_create(0, &d, &scene);
    if (rc != SRT_OK) { fprintf(stderr, "srt_scene_create: %s (hip error %d)\n", srt_strerror(rc), srt_last_hip_error()); return 1; }

    srt_params p;
    srt_params_default(&p, W, H);
    p.focal = 0.5f * (float)W;                       /* the reference's literal is 400 for a 600 x 400 image */
    const float light_base[3] = { 150.0f, -500.0f, 100.0f };
    float lights[3 * 4];
    srt_light_staircase(light_base, 4, lights);      /* 4 light samples of the reference's soft-shadow staircase */
    p.n_lights = 4; p.light_pos = lights;

    int32_t* hit = (int32_t*)malloc(sizeof(int32_t) * W * H);
    float* t = (float*)malloc(sizeof(float) * W * H);
    uint8_t* rgb8 = (uint8_t*)malloc((size_t)3 * W * H);
    srt_stats st;
    rc = srt_render(scene, &p, hit, t, NULL, rgb8, &st);
    if (rc != SRT_OK) { fprintf(stderr, "srt_render: %s\n", srt_strerror(rc)); return 1; }
    fprintf(stderr, "%ux%u: %llu primary rays, %llu hit, %llu shadow rays, %.3f ms on the GPU\n", W, H,
            (unsigned long long)st.primary_rays, (unsigned long long)st.hit_rays, (unsigned long long)st.shadow_rays, st.ms_total);

    if (argc > 3) {
        FILE* f = fopen(argv[3], "wb");
        if (!f) { perror(argv[3]); return 1; }
        fprintf(f, "P6\n%u %u\n255\n", W, H);
        fwrite(rgb8, 3, (size_t)W * H, f);
        fclose(f);
    } else {
        for (uint32_t y = 0; y < H; y++) {           /* hit ids as characters: '.' background, '0'..'3' triangles */
            for (uint32_t x = 0; x < W; x++) putchar(hit[y * W + x] < 0 ? '.' : (char)('0' + hit[y * W + x]));
            putchar('\n');
        }
    }
    free(hit); free(t); free(rgb8);
    srt_scene_destroy(scene);
    return 0;
}
