// replay_main.cpp -- headless counterpart of the reference's main loop (main.cpp:47-96, 175-229):
//   pfslam_replay <scene.txt> <lidar.f32|.mat> [frames] [grid]      ("grid": the 2-D occupancy-grid stages)
// iteration 0: Free + Init; then particleFilter(pbo=NULL, ++iteration, lidar) until the scans run out.
// Prints one line per frame (pose, map size) and the mean step time.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "kernel.h"

int main(int argc, char **argv)
{
    if (argc < 3) {
        printf("Usage: %s SCENEFILE.txt LIDARFILE.f32|.mat [frames] [grid]\n", argv[0]);
        return 1;
    }
    Scene *scene = new Scene(argv[1]);
    Lidar *lidar = new Lidar(argv[2]);
    size_t last = lidar->scans.size() - 1;
    if (argc > 3 && atoi(argv[3]) > 0) last = std::min(last, (size_t)atoi(argv[3]));
    if (argc > 4 && strcmp(argv[4], "grid") == 0) pfslamUseGridMap(true);
    particleFilterFree();
    particleFilterInit(scene);
    double total_ms = 0;
    size_t iteration = 0;
    while (iteration < last) {
        iteration++;
        auto t0 = std::chrono::steady_clock::now();
        particleFilter(nullptr, (int)iteration, lidar);
        Particle *p; MAP_TYPE *map; KDTree::Node *kd; int np, nkd; glm::vec3 pos;
        getPCData(&p, &map, &kd, &np, &nkd, pos);
        total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        unsigned int bx, by, bt;
        memcpy(&bx, &pos.x, 4); memcpy(&by, &pos.y, 4); memcpy(&bt, &pos.z, 4);
        printf("frame %zu pose %.6f %.6f %.6f bits %08x %08x %08x particles %d kd %d\n", iteration, pos.x, pos.y, pos.z, bx, by, bt, np, nkd);
    }
    printf("mean step+readback %.3f ms over %zu frames\n", iteration ? total_ms / iteration : 0.0, iteration);
    particleFilterFree();
    delete lidar;
    delete scene;
    return 0;
}
