// bench_main.cpp -- C++ counterpart of bench.py's timed loop, straight on the C-ABI (include/pfslam.h):
//   pfslam_bench <map.nodes> <scans.f32> <particles> [steps] [warmup] [first_frame]
// map.nodes : KDTree::Node array (32 bytes per node, the layout of kdtree.hpp:16-27), e.g. written by
//             numpy's tree.tofile() from gpu-icp-slam_amd.kd_create(points)
// scans.f32 : frames x 1081 float32 (the flat format Lidar also reads); frame k of the run uses scans[k % frames]
// Prints one JSON line: particle-scan evaluations/s of whole pfslam_step frames, timed with std::chrono around a
// pfslam_synchronize on both sides (the same bracket bench.py uses).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>
#include "../../include/pfslam.h"

static std::vector<char> slurp(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) { fprintf(stderr, "cannot open %s\n", path); exit(EXIT_FAILURE); }
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
#define OK(call) do { if (call) { fprintf(stderr, "%s: %s\n", #call, pfslam_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 4) {
        printf("Usage: %s MAP.nodes SCANS.f32 PARTICLES [steps=20] [warmup=5] [first_frame=6]\n", argv[0]);
        return 1;
    }
    const std::vector<char> map = slurp(argv[1]), scans = slurp(argv[2]);
    const int n_nodes = (int)(map.size() / sizeof(pfslam_node)), beams = 1081;
    const int n_frames = (int)(scans.size() / (beams * sizeof(float)));
    const int particles = atoi(argv[3]);
    const int steps = argc > 4 ? atoi(argv[4]) : 20, warmup = argc > 5 ? atoi(argv[5]) : 5;
    int frame = argc > 6 ? atoi(argv[6]) : 6;
    if (n_nodes <= 0 || n_frames <= 0 || particles <= 0) { fprintf(stderr, "empty map / scans / particle count\n"); return 1; }
    pfslam_config cfg;
    pfslam_default_config(&cfg);
    cfg.n_particles = particles;
    cfg.kd_capacity = n_nodes + (1 << 17);
    pfslam_handle *h = nullptr;
    OK(pfslam_create(&cfg, &h));
    OK(pfslam_set_map(h, reinterpret_cast<const pfslam_node *>(map.data()), n_nodes));
    const float *s = reinterpret_cast<const float *>(scans.data());
    for (int f = 1; f <= 5; f++) OK(pfslam_motion_update(h, f)); // same dispersed starting cloud as bench.py
    int k = 0;
    for (int i = 0; i < warmup; i++, k++) OK(pfslam_step(h, frame++, s + (size_t)(k % n_frames) * beams));
    OK(pfslam_synchronize(h));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; i++, k++) OK(pfslam_step(h, frame++, s + (size_t)(k % n_frames) * beams));
    OK(pfslam_synchronize(h));
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    float pose[3];
    OK(pfslam_get_pose(h, pose));
    printf("{\"metric\": \"particle-scan evals/sec, full particleFilter step (C++ driver)\", \"value\": %.6e, \"ms_per_step\": %.6f, "
           "\"particles\": %d, \"map_points\": %d, \"kd_size_end\": %d, \"steps\": %d, \"warmup\": %d, \"pose\": [%.6f, %.6f, %.6f]}\n",
           (double)particles * steps / sec, sec / steps * 1e3, particles, n_nodes, pfslam_kd_size(h), steps, warmup, pose[0], pose[1], pose[2]);
    OK(pfslam_destroy(h));
    return 0;
}
