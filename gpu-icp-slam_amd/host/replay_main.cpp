// replay_main.cpp -- headless counterpart of the reference's main loop (main.cpp:47-96, 175-229):
//   pfslam_replay <scene.txt> <lidar.f32|.mat> [frames] [grid] [loop] [export=PREFIX]
//     grid           the 2-D occupancy-grid stages instead of the point-cloud ones
//     loop           UpdateTopology + CheckLoopClosure at the end of every frame (kernel.cu:1750-1751, commented out in
//                    the reference's shipped step); loop-closure proposals are printed per frame
//     export=PREFIX  after the last frame: the map as the reference's viewer filters it (KD nodes with w > -100,
//                    main.cpp:269-284) and the occupancy grid -> PREFIX.kd.bin / .kd.csv / .grid.i8 / .grid.pgm
// iteration 0: Free + Init; then particleFilter(pbo=NULL, ++iteration, lidar) until the scans run out.
// Prints one line per frame (pose, map size) and the mean step time.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "kernel.h"

int main(int argc, char **argv)
{
    if (argc < 3) {
        printf("Usage: %s SCENEFILE.txt LIDARFILE.f32|.mat [frames] [grid] [loop] [export=PREFIX]\n", argv[0]);
        return 1;
    }
    Scene *scene = new Scene(argv[1]);
    Lidar *lidar = new Lidar(argv[2]);
    size_t last = lidar->scans.size() - 1;
    bool loop = false;
    std::string export_prefix;
    for (int i = 3; i < argc; i++) {
        if (strcmp(argv[i], "grid") == 0) pfslamUseGridMap(true);
        else if (strcmp(argv[i], "loop") == 0) loop = true;
        else if (strncmp(argv[i], "export=", 7) == 0) export_prefix = argv[i] + 7;
        else if (atoi(argv[i]) > 0) last = std::min(last, (size_t)atoi(argv[i]));
    }
    particleFilterFree();
    particleFilterInit(scene);
    pfslamUseTopology(loop);
    double total_ms = 0;
    size_t iteration = 0, proposals = 0;
    while (iteration < last) {
        iteration++;
        auto t0 = std::chrono::steady_clock::now();
        particleFilter(nullptr, (int)iteration, lidar);
        Particle *p; MAP_TYPE *map; KDTree::Node *kd; int np, nkd; glm::vec3 pos;
        getPCData(&p, &map, &kd, &np, &nkd, pos);
        total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        unsigned int bx, by, bt;
        memcpy(&bx, &pos.x, 4); memcpy(&by, &pos.y, 4); memcpy(&bt, &pos.z, 4);
        printf("frame %zu pose %.6f %.6f %.6f bits %08x %08x %08x particles %d kd %d", iteration, pos.x, pos.y, pos.z, bx, by, bt, np, nkd);
        if (loop) {
            const auto pairs = pfslamLoopClosures();
            proposals += pairs.size();
            printf(" closures %zu", pairs.size());
            for (size_t k = 0; k < pairs.size() && k < 8; k++) printf(" (%d,%d)", pairs[k].first, pairs[k].second);
        }
        printf("\n");
    }
    printf("mean step+readback %.3f ms over %zu frames\n", iteration ? total_ms / iteration : 0.0, iteration);
    if (loop) printf("loop-closure proposals %zu\n", proposals);
    if (!export_prefix.empty()) printf("exported %d map points to %s.*\n", pfslamExportMap(export_prefix.c_str()), export_prefix.c_str());
    particleFilterFree();
    delete lidar;
    delete scene;
    return 0;
}
