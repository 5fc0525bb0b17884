// pfslam_mgpu.cpp -- multi-GPU driver in C++ on librccl directly: one process per GPU, particles sharded over the ranks,
// map / scan / ICP / map update replicated, three all-gathers per frame over xGMI on a fixed schedule, no host wait per frame.
// The frame loop is libpfslam_mgpu.so's (include/pfslam_mgpu.h, host/pfslam_mgpu_lib.cpp: the sharded frame of include/pfslam.h with
// its all-gathers launched straight into the frame's own streams); bench.py --gpus N steps its ranks through the same library.
//
//   pfslam_mgpu --gpus N MAP.nodes SCANS.f32 PARTICLES_PER_GPU [--steps K] [--warmup W] [--first-frame F] [--dump PREFIX]
//               [--global-particles G] [--topology 1|2]
//
// Launcher mode (no rank in the environment): starts N copies of itself, one per GPU (fork + exec, so no process inherits
// an initialised HIP / RCCL runtime), and waits for them.  Rank mode: PFSLAM_RANK / PFSLAM_WORLD / PFSLAM_LOCAL_RANK, or the
// torchrun names RANK / WORLD_SIZE / LOCAL_RANK; the ncclUniqueId travels through the file PFSLAM_ID_FILE (rank 0 writes
// it, the others wait for it).
// MAP.nodes : KDTree::Node array (kdtree.hpp:16-27, 32 bytes per node);  SCANS.f32 : frames x 1081 float32.
// Rank 0 prints ONE JSON line of the same form as bench.py's: whole-job particle-scan evaluations per second over exactly K
// steps, barrier + stream synchronisation on both sides, MAX over the ranks.  --dump writes every rank's final particles and
// map (PREFIX.rank<r>.particles / .nodes) for bit-comparison with a single-handle pfslam_step run.
// --global-particles G: total particle count when it is not N x PARTICLES_PER_GPU (ragged last shard).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "../../include/pfslam_mgpu.h"

static int g_rank = 0;
#define PF(call)                                                                                          \
    do {                                                                                                  \
        if (call) {                                                                                       \
            fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, pfslam_last_error());                    \
            return 1;                                                                                     \
        }                                                                                                 \
    } while (0)
#define HIP(call)                                                                                         \
    do {                                                                                                  \
        hipError_t e__ = (call);                                                                          \
        if (e__ != hipSuccess) {                                                                          \
            fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, hipGetErrorString(e__));                 \
            return 1;                                                                                     \
        }                                                                                                 \
    } while (0)
static std::vector<char> slurp(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) {
        fprintf(stderr, "cannot open %s\n", path);
        exit(EXIT_FAILURE);
    }
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static const char *env_any(const char *a, const char *b)
{
    const char *v = getenv(a);
    return v ? v : getenv(b);
}

struct Args {
    int gpus = 1, steps = 20, warmup = 5, first_frame = 6;
    int topology = 0; // --topology 1|2: UpdateTopology + CheckLoopClosure inside the sharded frame (pfslam_set_topology; BASELINE configs[4])
    long global_particles = 0;
    std::string map, scans, dump;
    int particles = 0;
};

static bool parse(int argc, char **argv, Args &a)
{
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
        const std::string s = argv[i];
        auto next = [&](int &dst) { if (i + 1 < argc) dst = atoi(argv[++i]); };
        if (s == "--gpus") next(a.gpus);
        else if (s == "--steps") next(a.steps);
        else if (s == "--warmup") next(a.warmup);
        else if (s == "--first-frame") next(a.first_frame);
        else if (s == "--topology") next(a.topology);
        else if (s == "--global-particles") { if (i + 1 < argc) a.global_particles = atol(argv[++i]); }
        else if (s == "--dump") { if (i + 1 < argc) a.dump = argv[++i]; }
        else pos.push_back(s);
    }
    if (pos.size() < 3) return false;
    a.map = pos[0];
    a.scans = pos[1];
    a.particles = atoi(pos[2].c_str());
    return a.gpus >= 1 && a.particles > 0 && a.steps > 0 && a.warmup >= 0;
}

// ---- launcher: N copies of this binary, one per GPU ---------------------------------------------------------------
static int launch(int argc, char **argv, const Args &a)
{
    // a rank without a GPU of its own would leave the others hanging in ncclCommInitRank: refuse up front (the children exec a fresh
    // image, so looking at the device count here initialises nothing they inherit)
    const int ndev = pfslam_device_count();
    if (a.gpus > ndev) {
        fprintf(stderr, "pfslam_mgpu: --gpus %d but only %d device(s) are visible: refusing to launch ranks\n", a.gpus, ndev);
        return 2;
    }
    char tmpl[] = "/tmp/pfslam_mgpu_XXXXXX";
    if (!mkdtemp(tmpl)) {
        perror("mkdtemp");
        return 1;
    }
    const std::string id_file = std::string(tmpl) + "/nccl_id";
    std::vector<pid_t> kids;
    for (int r = 0; r < a.gpus; r++) {
        const pid_t pid = fork();
        if (pid < 0) {
            perror("fork");
            return 1;
        }
        if (pid == 0) {
            setenv("PFSLAM_RANK", std::to_string(r).c_str(), 1);
            setenv("PFSLAM_WORLD", std::to_string(a.gpus).c_str(), 1);
            setenv("PFSLAM_LOCAL_RANK", std::to_string(r).c_str(), 1);
            setenv("PFSLAM_ID_FILE", id_file.c_str(), 1);
            setenv("LOCAL_WORLD_SIZE", std::to_string(a.gpus).c_str(), 0); // the host kd build shares the node's cores with the other ranks (csrc/kd_host.cpp)
            execv("/proc/self/exe", argv);
            perror("execv");
            _exit(127);
        }
        kids.push_back(pid);
    }
    int rc = 0;
    for (pid_t pid : kids) {
        int st = 0;
        waitpid(pid, &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
    }
    unlink(id_file.c_str());
    rmdir(tmpl);
    (void)argc;
    return rc;
}

// ---- one rank ---------------------------------------------------------------------------------------------------------
static int get_id(int rank, const std::string &path, unsigned char id[PFSLAM_MGPU_ID_BYTES])
{
    if (rank == 0) {
        if (pfslam_mgpu_make_id(id)) {
            fprintf(stderr, "[rank 0] %s\n", pfslam_mgpu_last_error());
            return 1;
        }
        unlink(path.c_str()); // an id left behind by a run that died (a clean run removes its own, see run_rank)
        const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
        FILE *f = fd >= 0 ? fdopen(fd, "wb") : nullptr;
        if (!f || fwrite(id, PFSLAM_MGPU_ID_BYTES, 1, f) != 1) {
            fprintf(stderr, "cannot write %s\n", tmp.c_str());
            return 1;
        }
        fclose(f);
        if (rename(tmp.c_str(), path.c_str())) { // atomic: readers never see a partial id
            perror("rename");
            return 1;
        }
        return 0;
    }
    for (int tries = 0; tries < 6000; tries++) { // up to 60 s
        struct stat st;
        if (stat(path.c_str(), &st) == 0 && st.st_size == (off_t)PFSLAM_MGPU_ID_BYTES && st.st_uid == getuid()) {
            FILE *f = fopen(path.c_str(), "rb");
            if (f && fread(id, PFSLAM_MGPU_ID_BYTES, 1, f) == 1) {
                fclose(f);
                return 0;
            }
            if (f) fclose(f);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    fprintf(stderr, "[rank %d] no job id in %s after 60 s\n", rank, path.c_str());
    return 1;
}

#define MG(call)                                                                                          \
    do {                                                                                                  \
        if (call) {                                                                                       \
            fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #call, pfslam_mgpu_last_error());               \
            return 1;                                                                                     \
        }                                                                                                 \
    } while (0)

static int run_rank(const Args &a, int rank, int world, int local_rank, const std::string &id_file)
{
    g_rank = rank;
    const std::vector<char> map = slurp(a.map.c_str()), scans = slurp(a.scans.c_str());
    const int n_nodes = (int)(map.size() / sizeof(pfslam_node)), beams = 1081;
    const int n_frames = (int)(scans.size() / (beams * sizeof(float)));
    if (n_nodes <= 0 || n_frames <= 0) {
        fprintf(stderr, "empty map or scan file\n");
        return 1;
    }
    // shard layout: rank r owns [r * stride, min((r + 1) * stride, G))
    const long G = a.global_particles > 0 ? a.global_particles : (long)a.particles * world;
    const int stride = (int)((G + world - 1) / world);
    const long off = (long)rank * stride;
    const int count = (int)std::min<long>(stride, G - off);
    if (count <= 0) {
        fprintf(stderr, "[rank %d] empty shard: %ld particles over %d ranks\n", rank, G, world);
        return 1;
    }
    {
        int ndev = 0;
        HIP(hipGetDeviceCount(&ndev));
        if (local_rank >= ndev) { // (started by torchrun or by hand with more ranks than devices)
            fprintf(stderr, "[rank %d] local rank %d but only %d device(s) are visible: refusing to join (the other ranks would hang in ncclCommInitRank)\n", rank, local_rank, ndev);
            return 2;
        }
    }
    HIP(hipSetDevice(local_rank));
    unsigned char id[PFSLAM_MGPU_ID_BYTES];
    memset(id, 0, sizeof(id));
    if (world > 1 && get_id(rank, id_file, id)) return 1;
    pfslam_config cfg;
    pfslam_default_config(&cfg);
    cfg.n_particles = count;
    cfg.kd_capacity = n_nodes + (1 << 18);
    cfg.device = local_rank;
    cfg.global_offset = (int)off;
    cfg.global_n = (int)G;
    cfg.shard_stride = stride;
    pfslam_handle *h = nullptr;
    PF(pfslam_create(&cfg, &h));
    pfslam_mgpu *M = nullptr;
    MG(pfslam_mgpu_create(world > 1 ? id : nullptr, world, rank, h, &M));
    if (rank == 0 && world > 1) unlink(id_file.c_str()); // every rank has joined: a later run must never pick this id up
    if (a.topology) PF(pfslam_set_topology(h, a.topology)); // replicated: every rank keeps the same graph, nothing is exchanged for it
    PF(pfslam_set_map(h, reinterpret_cast<const pfslam_node *>(map.data()), n_nodes));
    const float *s = reinterpret_cast<const float *>(scans.data());
    for (int f = 1; f <= 5; f++) PF(pfslam_motion_update(h, f)); // same dispersed starting cloud as bench.py
    int frame = a.first_frame, k = 0;
    for (int i = 0; i < a.warmup; i++, k++) MG(pfslam_mgpu_step(M, frame++, s + (size_t)(k % n_frames) * beams));
    MG(pfslam_mgpu_barrier_max(M, nullptr));
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < a.steps; i++, k++) MG(pfslam_mgpu_step(M, frame++, s + (size_t)(k % n_frames) * beams));
    MG(pfslam_mgpu_barrier_max(M, nullptr));
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    MG(pfslam_mgpu_barrier_max(M, &sec)); // MAX over the ranks
    float pose[3];
    PF(pfslam_get_pose(h, pose));
    int mode[4] = {0, 0, 0, 0};
    PF(pfslam_frame_mode(h, mode));
    // topology: the loop-closure proposals of the last frame and the graph, which must be the same on every rank (checked: MAX == MIN of a digest)
    int n_closures = 0, n_topo = 0, topo_idx = 0;
    if (a.topology) {
        PF(pfslam_get_closures(h, nullptr, 0, &n_closures));
        std::vector<float> nodes(3 * 4096);
        PF(pfslam_get_topology(h, nodes.data(), 4096, &n_topo, &topo_idx));
        double digest = 0.0;
        for (int k = 0; k < 3 * std::min(n_topo, 4096); k++) digest = digest * 1.000001 + (double)nodes[k];
        digest += 1e6 * n_closures + 1e3 * topo_idx + n_topo;
        double hi = digest, lo = -digest;
        MG(pfslam_mgpu_barrier_max(M, &hi));
        MG(pfslam_mgpu_barrier_max(M, &lo));
        if (hi != -lo) {
            fprintf(stderr, "[rank %d] the ranks' topology graphs differ\n", rank);
            return 3;
        }
    }
    if (!a.dump.empty()) {
        const pfslam_particle *p;
        const pfslam_node *nd;
        int np = 0, nn = 0;
        PF(pfslam_get_particles(h, &p, &np));
        std::ofstream(a.dump + ".rank" + std::to_string(rank) + ".particles", std::ios::binary).write((const char *)p, (size_t)np * sizeof(*p));
        PF(pfslam_get_map(h, &nd, &nn));
        std::ofstream(a.dump + ".rank" + std::to_string(rank) + ".nodes", std::ios::binary).write((const char *)nd, (size_t)nn * sizeof(*nd));
    }
    int st[4] = {0, 0, 0, 0};
    MG(pfslam_mgpu_stats(M, st));
    if (rank == 0) {
        printf("{\"metric\": \"particle-scan evals/sec (1081 beams x N particles), full particleFilter step, KD path\", \"value\": %.6e, "
               "\"unit\": \"particle-scan evals/s\", \"n_gpus\": %d, \"steps\": %d, \"warmup\": %d, \"ms_per_step\": %.6f, "
               "\"higher_is_better\": true, \"scaling\": \"weak\", \"vs_baseline\": null, \"dtype\": \"f32\", \"data\": \"synthetic\", "
               "\"config\": {\"workload\": \"1081-beam scans, %d particles/GPU, %d-point KD map, full SLAM step\", \"particles_global\": %ld, "
               "\"parallelism\": \"particles sharded x%d, map replicated\", \"kd_size_end\": %d, \"driver\": \"C++ / librccl (host/pfslam_mgpu.cpp)\", "
               "\"collectives\": %d, \"frame\": {\"round5\": %d, \"edges\": \"%s\", \"one_stream\": %d}, "
               "\"topology\": {\"mode\": %d, \"nodes\": %d, \"node\": %d, \"closures_last_frame\": %d}, \"pose\": [%.9g, %.9g, %.9g]}}\n",
               (double)G * a.steps / sec, world, a.steps, a.warmup, sec / a.steps * 1e3, stride, n_nodes, G, world, pfslam_kd_size(h),
               st[0], mode[0], mode[1] ? "gates" : "events", mode[2], a.topology, n_topo, topo_idx, n_closures, pose[0], pose[1], pose[2]);
        fflush(stdout);
    }
    MG(pfslam_mgpu_destroy(M));
    PF(pfslam_destroy(h));
    return 0;
}

int main(int argc, char **argv)
{
    // multi-process GPU work on this host driver needs dmabuf IPC (RCCL's hipIpcGetMemHandle fails otherwise); the HSA runtime reads the
    // variable when it initialises -- before this process's, and its children's, first HIP call
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
    Args a;
    if (!parse(argc, argv, a)) {
        printf("Usage: %s --gpus N MAP.nodes SCANS.f32 PARTICLES_PER_GPU [--steps K] [--warmup W] [--first-frame F] [--dump PREFIX] "
               "[--global-particles G] [--topology 1|2]\n", argv[0]);
        return 1;
    }
    const char *rk = env_any("PFSLAM_RANK", "RANK");
    if (!rk) {
        if (a.gpus == 1) { // one rank: no children, no job id
            setenv("PFSLAM_RANK", "0", 1);
            return run_rank(a, 0, 1, 0, "");
        }
        return launch(argc, argv, a);
    }
    const char *ws = env_any("PFSLAM_WORLD", "WORLD_SIZE"), *lr = env_any("PFSLAM_LOCAL_RANK", "LOCAL_RANK");
    const int rank = atoi(rk), world = ws ? atoi(ws) : 1, local_rank = lr ? atoi(lr) : rank;
    std::string id_file = getenv("PFSLAM_ID_FILE") ? getenv("PFSLAM_ID_FILE") : "";
    if (id_file.empty()) {
        const char *port = getenv("MASTER_PORT");
        id_file = std::string("/tmp/pfslam_nccl_id.") + std::to_string((long)getuid()) + "." + (port ? port : "0");
    }
    return run_rank(a, rank, world, local_rank, id_file);
}
