// host_impl.cpp -- C++ host layer above the C-ABI: the reference's kernel.h / Lidar / Scene / Pointcloud /
// KDTree surface.  Module-global state and print-and-exit errors mirror the reference (kernel.cu:55-83,
// kernel.h:42-60); everything else is a thin forwarding layer onto include/pfslam.h.
#include "kernel.h"
#include "pointcloud.h"
#include "mat5_reader.h"
#include "../../include/pfslam.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

// ---------------------------------------------------------------- KDTree
static_assert(sizeof(KDTree::Node) == sizeof(pfslam_node), "Node layout");
void KDTree::Create(std::vector<glm::vec4> input, Node *list)
{
    pfslam_kd_create(reinterpret_cast<const float *>(input.data()), (int)input.size(), reinterpret_cast<pfslam_node *>(list));
}
void KDTree::InsertList(std::vector<glm::vec4> input, Node *list, int idx, int parent)
{
    pfslam_kd_insert_list(reinterpret_cast<const float *>(input.data()), (int)input.size(), reinterpret_cast<pfslam_node *>(list), idx, parent);
}
void KDTree::InsertNode(glm::vec4 point, Node *list, int listSize)
{
    const float p[4] = {point.x, point.y, point.z, point.w};
    pfslam_kd_insert_node(p, reinterpret_cast<pfslam_node *>(list), listSize);
}
void KDTree::Balance(Node *list, int listSize) { pfslam_kd_balance(reinterpret_cast<pfslam_node *>(list), listSize); }

// ---------------------------------------------------------------- loaders
static std::vector<std::string> tokens_of(const std::string &line)
{
    std::istringstream ss(line);
    std::vector<std::string> out;
    std::string t;
    while (ss >> t) out.push_back(t);
    return out;
}
static bool next_line(std::istream &f, std::string &line)
{
    if (!std::getline(f, line)) { line.clear(); return false; }
    if (!line.empty() && line.back() == '\r') line.pop_back();
    return true;
}

Lidar::Lidar(std::string filename)
{
    std::cout << "Reading lidar data from " << filename << " ..." << std::endl;
    const bool is_mat = filename.size() > 4 && filename.compare(filename.size() - 4, 4, ".mat") == 0;
    if (is_mat) { // the reference's format: MAT-v5, cell array `lidar` of structs with a `scan` field (lidar.cpp:17-49)
        std::string err;
        if (!mat5_read_lidar_scans(filename, scans, err)) {
            std::cout << "Error reading from file - aborting! (" << err << ")" << std::endl;
            throw std::runtime_error("Lidar: " + err);
        }
        return;
    }
    std::ifstream f(filename, std::ios::binary);
    if (!f.is_open()) {
        std::cout << "Error reading from file - aborting!" << std::endl;
        throw std::runtime_error("Lidar: cannot open " + filename);
    }
    f.seekg(0, std::ios::end);
    const std::streamoff bytes = f.tellg();
    f.seekg(0);
    const int beams = 1081;
    if (bytes % (beams * 4) != 0) throw std::runtime_error("Lidar: file is neither .mat nor frames x 1081 float32");
    const size_t frames = (size_t)(bytes / (beams * 4));
    scans.resize(frames, std::vector<float>(beams));
    for (size_t i = 0; i < frames; i++) f.read(reinterpret_cast<char *>(scans[i].data()), beams * 4);
}

Scene::Scene(std::string filename)
{
    std::cout << "Reading scene from " << filename << " ..." << std::endl;
    std::ifstream in(filename);
    if (!in.is_open()) {
        std::cout << "Error reading from file - aborting!" << std::endl;
        throw std::runtime_error("Scene: cannot open " + filename);
    }
    std::string line;
    while (next_line(in, line)) {
        const auto t = tokens_of(line);
        if (t.empty()) continue;
        if (t[0] == "MAP") parse_map_block(in);
        else if (t[0] == "CAMERA") parse_camera_block(in);
    }
}
void Scene::parse_map_block(std::istream &in)
{
    glm::vec3 gridSize;
    float res = 1.0f;
    std::string line;
    while (next_line(in, line) && !line.empty()) {
        const auto t = tokens_of(line);
        if (t.empty()) break;
        if (t[0] == "SIZE" && t.size() >= 3) gridSize = glm::vec3((float)atof(t[1].c_str()), (float)atof(t[2].c_str()), 0);
        else if (t[0] == "RES" && t.size() >= 2) res = (float)atof(t[1].c_str());
    }
    Patch m;
    m.scale = gridSize;
    m.resolution = glm::vec3(res, res, 1.0f);
    m.grid = nullptr;
    m.uid = 0;
    maps.push_back(m);
}
static glm::vec3 v_normalize(const glm::vec3 &v) // glm::normalize: v * inversesqrt(dot(v, v)); the zero vector gives NaN
{
    const float inv = 1.0f / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return glm::vec3(v.x * inv, v.y * inv, v.z * inv);
}
void Scene::parse_camera_block(std::istream &in)
{
    Camera &c = state.camera;
    std::string line;
    while (next_line(in, line) && !line.empty()) {
        const auto t = tokens_of(line);
        if (t.empty()) break;
        if (t[0] == "RES" && t.size() >= 3) c.resolution = glm::ivec2(atoi(t[1].c_str()), atoi(t[2].c_str()));
        else if (t[0] == "FOVY" && t.size() >= 2) c.fov.y = (float)atof(t[1].c_str());
        else if (t[0] == "FILE" && t.size() >= 2) state.imageName = t[1];
        else if (t[0] == "EYE" && t.size() >= 4) c.position = glm::vec3((float)atof(t[1].c_str()), (float)atof(t[2].c_str()), (float)atof(t[3].c_str()));
        else if (t[0] == "LOOKAT" && t.size() >= 4) c.lookAt = glm::vec3((float)atof(t[1].c_str()), (float)atof(t[2].c_str()), (float)atof(t[3].c_str()));
        else if (t[0] == "UP" && t.size() >= 4) c.up = glm::vec3((float)atof(t[1].c_str()), (float)atof(t[2].c_str()), (float)atof(t[3].c_str()));
    }
    // derived fields, in the reference's order (scene.cpp:107-121): fov.x from the aspect ratio; `right` is computed from
    // `view` BEFORE view is assigned (so from the zero vector here: NaN components, as normalize(0) gives) and is not used by
    // the SLAM path; pixelLength; view; the image buffer sized to the resolution
    const float PI_F = 3.1415926535897932384626422832795028841971f; // utilities.h:12
    const float fovy = c.fov.y;
    const float yscaled = tanf(fovy * (PI_F / 180));
    const float xscaled = (yscaled * c.resolution.x) / c.resolution.y;
    const float fovx = (atanf(xscaled) * 180) / PI_F;
    c.fov = glm::vec2(fovx, fovy);
    const glm::vec3 cr(c.view.y * c.up.z - c.view.z * c.up.y, c.view.z * c.up.x - c.view.x * c.up.z, c.view.x * c.up.y - c.view.y * c.up.x);
    c.right = v_normalize(cr);
    c.pixelLength = glm::vec2(2 * xscaled / (float)c.resolution.x, 2 * yscaled / (float)c.resolution.y);
    c.view = v_normalize(glm::vec3(c.lookAt.x - c.position.x, c.lookAt.y - c.position.y, c.lookAt.z - c.position.z));
    const int arraylen = c.resolution.x * c.resolution.y;
    state.image.assign(arraylen > 0 ? (size_t)arraylen : 0, glm::vec3());
}

Pointcloud::Pointcloud(std::string filename)
{
    std::ifstream f(filename);
    if (!f.is_open()) throw std::runtime_error("Pointcloud: cannot open " + filename);
    std::string line;
    int i = 0;
    while (next_line(f, line)) {
        const auto t = tokens_of(line);
        if (t.size() < 3) continue;
        points.push_back(glm::vec4((float)atoi(t[2].c_str()), (float)atoi(t[0].c_str()), (float)atoi(t[1].c_str()), (float)i++));
    }
}

// ---------------------------------------------------------------- kernel.h
static pfslam_handle *g_handle = nullptr;
static bool g_grid_map = false;
static Scene *g_scene = nullptr;
static int g_particles = 1000; // PARTICLE_COUNT, kernel.cu:30
static glm::vec3 g_robotPos;
static bool g_topology = false;

void checkPfslamErrorFn(int rc, const char *msg, const char *file, int line)
{
    if (rc == 0) return;
    fprintf(stderr, "pfslam error (%s:%d): %s: %s\n", file, line, msg, pfslam_last_error());
    exit(EXIT_FAILURE);
}
#define PFCHK(call, msg) checkPfslamErrorFn((call), msg, __FILE__, __LINE__)
void checkCUDAErrorFn(const char *msg, const char *file, int line)
{
    if (g_handle) checkPfslamErrorFn(pfslam_synchronize(g_handle), msg, file, line); // cudaDeviceSynchronize + cudaGetLastError
}

void pfslamSetParticleCount(int n) { if (n > 0) g_particles = n; }

void particleFilterInit(Scene *scene)
{
    g_scene = scene;
    if (const char *e = getenv("PFSLAM_PARTICLES")) pfslamSetParticleCount(atoi(e));
    if (const char *e = getenv("PFSLAM_MAP")) g_grid_map = strcmp(e, "grid") == 0;
    pfslam_config cfg;
    pfslam_default_config(&cfg);
    cfg.n_particles = g_particles;
    if (scene && !scene->maps.empty()) { // map_params = scene->maps[0] (kernel.cu:119)
        cfg.map_scale_x = scene->maps[0].scale.x;
        cfg.map_scale_y = scene->maps[0].scale.y;
        cfg.map_res_x = scene->maps[0].resolution.x;
        cfg.map_res_y = scene->maps[0].resolution.y;
    }
    if (const char *e = getenv("PFSLAM_KD_CAPACITY")) cfg.kd_capacity = atoi(e);
    PFCHK(pfslam_create(&cfg, &g_handle), "particleFilterInit");
    if (const char *e = getenv("PFSLAM_TOPOLOGY")) g_topology = atoi(e) != 0;
    PFCHK(pfslam_set_topology(g_handle, g_topology ? 1 : 0), "particleFilterInit (topology)");
    g_robotPos = glm::vec3(0.0f);
    particleFilterInitPC();
}
void particleFilterFree()
{
    if (g_handle) {
        PFCHK(pfslam_synchronize(g_handle), "particleFilterFree (a frame still in flight failed)"); // deferred errors of the last frame
        PFCHK(pfslam_destroy(g_handle), "particleFilterFree");
    }
    g_handle = nullptr;
    particleFilterFreePC();
}
void particleFilterInitPC() {} // folded into pfslam_create
void particleFilterFreePC() {}

void particleFilter(uchar4 *, int frame, Lidar *lidar)
{
    if (!g_handle || !lidar || frame < 0 || (size_t)frame >= lidar->scans.size()) {
        fprintf(stderr, "particleFilter: not initialised or frame out of range\n");
        exit(EXIT_FAILURE);
    }
    if (g_grid_map)
        PFCHK(pfslam_step_grid(g_handle, frame, lidar->scans[frame].data()), "particleFilter (grid)");
    else
        PFCHK(pfslam_step(g_handle, frame, lidar->scans[frame].data()), "particleFilter");
}
void particleFilterPC(int frame, Lidar *lidar)
{
    const bool grid = g_grid_map;
    g_grid_map = false;
    particleFilter(nullptr, frame, lidar);
    g_grid_map = grid;
}
void pfslamUseGridMap(bool on) { g_grid_map = on; }
void pfslamUseTopology(bool on)
{
    g_topology = on;
    if (g_handle) PFCHK(pfslam_set_topology(g_handle, on ? 1 : 0), "pfslamUseTopology");
}
std::vector<std::pair<int, int>> pfslamLoopClosures()
{
    std::vector<std::pair<int, int>> out;
    if (!g_handle) return out;
    int n = 0;
    PFCHK(pfslam_get_closures(g_handle, nullptr, 0, &n), "pfslamLoopClosures");
    std::vector<int32_t> buf((size_t)2 * (n > 0 ? n : 1));
    PFCHK(pfslam_get_closures(g_handle, buf.data(), n, &n), "pfslamLoopClosures");
    for (int k = 0; k < n; k++) out.push_back(std::make_pair((int)buf[2 * k], (int)buf[2 * k + 1]));
    return out;
}
void drawMap(uchar4 *) {}

int pfslamExportMap(const char *prefix)
{
    if (!g_handle || !prefix) return -1;
    const std::string base(prefix);
    const pfslam_node *kd = nullptr;
    const int8_t *grid = nullptr;
    int nk = 0, dx = 0, dy = 0;
    PFCHK(pfslam_get_map(g_handle, &kd, &nk), "pfslamExportMap map");
    PFCHK(pfslam_get_grid(g_handle, &grid, &dx, &dy), "pfslamExportMap grid");
    std::ofstream bin(base + ".kd.bin", std::ios::binary), csv(base + ".kd.csv");
    csv << "x y z w\n";
    int kept = 0;
    char line[160];
    for (int i = 0; i < nk; i++) {
        if (!(kd[i].w > -100)) continue; // the viewer's filter, main.cpp:269
        const float p[4] = {kd[i].x, kd[i].y, kd[i].z, kd[i].w};
        bin.write(reinterpret_cast<const char *>(p), 16);
        snprintf(line, sizeof(line), "%.9g %.9g %.9g %.9g\n", p[0], p[1], p[2], p[3]);
        csv << line;
        kept++;
    }
    std::ofstream(base + ".grid.i8", std::ios::binary).write(reinterpret_cast<const char *>(grid), (std::streamsize)dx * dy);
    std::ofstream pgm(base + ".grid.pgm", std::ios::binary);
    pgm << "P5\n" << dy << " " << dx << "\n255\n";
    std::vector<unsigned char> row((size_t)dy);
    for (int x = 0; x < dx; x++) {
        for (int y = 0; y < dy; y++) row[y] = (unsigned char)((int)grid[(size_t)x * dx + y] + 128);
        pgm.write(reinterpret_cast<const char *>(row.data()), dy);
    }
    return kept;
}

void getPCData(Particle **ptrParticles, MAP_TYPE **ptrMap, KDTree::Node **ptrKD, int *nParticles, int *nKD, glm::vec3 &pos)
{
    const pfslam_particle *p = nullptr;
    const pfslam_node *kd = nullptr;
    const int8_t *grid = nullptr;
    int np = 0, nk = 0, dx = 0, dy = 0;
    PFCHK(pfslam_get_particles(g_handle, &p, &np), "getPCData particles");
    PFCHK(pfslam_get_map(g_handle, &kd, &nk), "getPCData map");
    PFCHK(pfslam_get_grid(g_handle, &grid, &dx, &dy), "getPCData grid");
    float pose[3];
    PFCHK(pfslam_get_pose(g_handle, pose), "getPCData pose");
    // non-owning pointers into the library's host mirrors, valid until the next particleFilter / Free (kernel.cu:803-813)
    *ptrParticles = reinterpret_cast<Particle *>(const_cast<pfslam_particle *>(p));
    *ptrMap = reinterpret_cast<MAP_TYPE *>(const_cast<int8_t *>(grid));
    *ptrKD = reinterpret_cast<KDTree::Node *>(const_cast<pfslam_node *>(kd));
    *nParticles = np;
    *nKD = nk;
    pos = glm::vec3(pose[0], pose[1], pose[2]);
}
