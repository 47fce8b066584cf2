// examples/orbit.cpp -- a scene script in the shape of the reference's main() (simple_raytracer.cpp:530-796),
// written against the host-side mirror (simple_raytracer_amd/csrc/host/srt_host.h): the "Scene with 4 Cubes in
// different colors" block (:726-769) rendered over the camera orbit (:534-551) with the HIP path.
//
//   orbit <cube.obj> <out_dir> [frames=36] [width=600] [height=400] [lightAmount=1]
//
// Writes <out_dir>/output<angle>.bmp like drawImage (:488-494) and prints the reference's timing line (:791).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "../simple_raytracer_amd/csrc/host/srt_host.h"

using namespace srt_host;

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: orbit <cube.obj> <out_dir> [frames] [width] [height] [lightAmount]\n"); return 2; }
    const std::string cube = argv[1], out_dir = argv[2];
    const int frames = argc > 3 ? std::atoi(argv[3]) : 36;
    const float W = argc > 4 ? (float)std::atoi(argv[4]) : 600.f, H = argc > 5 ? (float)std::atoi(argv[5]) : 400.f;
    const int lightAmount = argc > 6 ? std::atoi(argv[6]) : 1;
    try {
        for (int f = 0; f < frames; f++) {
            const float angleDegree = (float)f * 10.0f;                              // :534
            ObjectManager objManager;
            const float radius = 100.0f;                                              // :730-735
            const float rad = radians(angleDegree);
            const float circleX = radius * std::cos(rad), circleZ = radius * std::sin(rad);
            const mat4 viewMatrix = Transformation::createViewMatrix(vec3(circleX, 0.f, circleZ),
                                                                     vec3(radians(0.f), radians(angleDegree + 90), radians(0.f)));
            objManager.loadObjFile(cube);                                             // :738-740
            objManager.setColor(cube, vec3(1.f, 1.f, 0.f));
            objManager.transformTriangles(cube, Transformation::scaleObj(10.0f, 10.0f, 10.0f));
            const char* names[3] = { "cube1.obj", "cube2.obj", "cube3.obj" };      // :743-753
            const vec3 colors[3] = { vec3(1.f, 0.f, 1.f), vec3(1.f, 0.f, 0.f), vec3(0.f, 1.f, 0.f) };
            const vec3 pos[3] = { vec3(0.f, -15.f, -15.f), vec3(0.f, -15.f, 15.f), vec3(0.f, 15.f, 15.f) };
            for (int k = 0; k < 3; k++) {
                objManager.objTriangles[names[k]] = objManager.getTriangles(cube);
                objManager.objColors[names[k]] = colors[k];
                objManager.transformTriangles(names[k], Transformation::changeObjPosition(pos[k]));
            }
            objManager.transformTriangles(cube, Transformation::changeObjPosition(vec3(0.f, 15.f, -15.f)));   // :756
            const std::string all[4] = { cube, names[0], names[1], names[2] };
            for (const std::string& n : all) objManager.transformTriangles(n, inverse(viewMatrix));           // :759-762
            for (const std::string& n : all) objManager.createBoundingHierarchy(n);                          // :765-768

            const vec2 imageSize(W, H);                                               // :773
            vec4 lightPos(500.0f, -300.0f, -200.f, 1.0f);                             // :776
            lightPos = inverse(viewMatrix) * lightPos;                                // :778
            auto start = std::chrono::high_resolution_clock::now();
            ImageData points = sendRaysAndIntersectPointsColors(imageSize, lightPos, &objManager, lightAmount);   // :784
            auto end = std::chrono::high_resolution_clock::now();
            std::chrono::duration<double> elapsed = end - start;
            std::cout << "Time taken for Intersection: " << elapsed.count() << " seconds " << std::endl;     // :791
            drawImage(imageSize, points.imagePoints, points.imageColors, (int)angleDegree, true, out_dir);  // :793
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "orbit: %s\n", e.what());
        return 1;
    }
    return 0;
}
