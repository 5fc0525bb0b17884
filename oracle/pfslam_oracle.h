/*
 * pfslam_oracle.h -- CPU oracle for the particle-filter SLAM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the algorithm in
 * the reference's src/kernel.cu (+ kdtree.cpp, svd3.h, utilities.cpp); it is the
 * checker the parity tests compare the HIP path against.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (gpu-icp-slam_amd/) never includes, links or calls anything in oracle/.
 *
 * Parity pinning status (see DESIGN.md "Oracle"):
 *   - KD-tree Create/Balance/InsertNode: PINNED against the reference's own
 *     src/kdtree.cpp compiled unmodified into oracle/_ref/libkdtree_ref.so.
 *   - RNG (minstd_rand / uniform_real / normal via erfcinv): PINNED against the
 *     image's rocThrust headers compiled for the host (oracle/_ref/thrust_probe);
 *     thrust itself is a dependency absent from /root/reference (CUDA 7.5 toolkit).
 *   - 3x3 SVD (svd3.h): PINNED on the GPU box against the reference's own svd3.h compiled unmodified for
 *     the device (oracle/svd_ref_kernel.cpp -> oracle/_ref/svd_ref.hsaco); tests/test_gpu_svd_ref.py.
 *   - The device kernels of kernel.cu (traversal: findCorrespondenceIndexKD / findCorrespondenceKD / the loop inside
 *     EvaluateParticleKD; kernEvaluateParticlesKD as a whole; traceRay / kernGetWalls / kernGetWallsKD; getHyperplaneDist;
 *     utilhash / makeSeededRandomEngine / uniform_real; kernWeightedSample; kernAddNoise; kernUpdateWeights /
 *     kernCopyWeights; kernUpdateMapKD / kernTestCorrespondance; kernEvaluateParticles / kernUpdateMap): PINNED on the GPU
 *     box against the reference's own kernel.cu compiled for gfx950, device code only, from a scratch copy made at build
 *     time by sed (byte-order mark; the blank inside `<< <` / `>> >`) and hipify-perl -- no hand edit, nothing committed
 *     (oracle/kernel_ref_wrap.cpp, oracle/Makefile -> oracle/_ref/kernel_ref.hsaco; tests/test_gpu_ref_kernels.py).
 *     Bit for bit wherever no transcendental function is involved, and bit for bit BEHIND CleanLidarScan's cos / sin (the
 *     reference's own end points fed to the restated traversal / ray code).  BOUND of that pin: the transcendentals of that
 *     build are ROCm's device library, not CUDA's libdevice (the fp64 specification here agrees with ROCm's cosf / sinf on
 *     70 % of the end points, always within 2 ulp; scores change only at nearest-node ties: 7 of 12 000 particles of the
 *     bench workload), and nvcc's default fma contraction is a property of ITS code generation (the same text built with
 *     clang's contraction on is reported beside it).  What never ran anywhere here: the reference's HOST code of
 *     kernel.cu (thrust reductions' order, the cudaMemcpy choreography, PFUpdateMapKD's host loops) -- restated from the
 *     cited lines; the undefined behaviours H1..H11 are given the definitions in DESIGN.md.
 *
 * Arithmetic contract: IEEE-754 binary32, one rounding per source-level
 * operation, no FMA contraction (built with -ffp-contract=off), sqrt and divide
 * correctly rounded.  Transcendentals (cos/sin, erfcinv, log, asin) follow the
 * "pf_math" specification below: fixed sequences of IEEE double operations, so
 * that the CPU oracle and the gfx950 kernels produce identical bits.
 */
#ifndef PFSLAM_ORACLE_H
#define PFSLAM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* KDTree::Node, reference src/kdtree.hpp:16-27 (32 bytes, align 4) */
typedef struct {
    int32_t axis, left, right, parent;
    float x, y, z, w;
} orc_node;

/* Particle, reference src/sceneStructs.h:33-38 (32 bytes: vec3 pos@0, w@12, cluster@16, ptr@24) */
typedef struct {
    float x, y, theta, w;
    uint8_t cluster;
    uint8_t pad_[7];
    uint64_t map;
} orc_particle;

/* the parts of Patch (sceneStructs.h:40-45) the hot path reads */
typedef struct {
    float scale_x, scale_y;
    float res_x, res_y;
} orc_patch;

#define ORC_LIDAR_SIZE 1081 /* kernel.cu:43 */

/* ---- A2: RNG (kernel.cu:89-102 + thrust::minstd_rand) ---- */
uint32_t orc_utilhash(uint32_t a);
uint32_t orc_engine_seed(int iter, int index, int depth);
uint32_t orc_minstd_next(uint32_t *state);
float orc_uniform_real(uint32_t *state, float a, float b);
float orc_normal(uint32_t *state, float mean, float stddev);

/* ---- pf_math specification (bit-reproducible transcendentals) ---- */
void orc_sincosf(float x, float *s, float *c);
void orc_sincos_d(float x, double *s, double *c); /* the same before the rounding to float */
double orc_log(double x);
double orc_ndtri(double y0);
float orc_erfcinvf(float y);
float orc_asinf(float x);
float orc_rsqrtf(float x);
/* canonical reduction order standing in for thrust::reduce (order unspecified) */
float orc_sum_f32(const float *v, int n, int stride);
/* canonical inclusive scan standing in for thrust::inclusive_scan */
void orc_inclusive_scan_f32(const float *w, int n, float *cdf);

/* ---- A3: motion / dispersion (kernel.cu:375-397) ---- */
void orc_add_noise(orc_particle *p, int n, int frame, int global_idx0);

/* ---- A4/A5: scan match against the KD map (kernel.cu:182-187, 1198-1308) ---- */
void orc_clean_lidar_scan(int n, float scan, float theta, float *x, float *y);
int orc_kd_traverse(const orc_node *tree, float px, float py, float pz, int *visits);
void orc_score_kd(const orc_node *tree, const orc_particle *p, int n, const float *scan,
                  int n_beams, float *fit, uint64_t *node_visits, uint64_t *valid_beams);
void orc_score_kd_mt(const orc_node *tree, const orc_particle *p, int n, const float *scan,
                     int n_beams, float *fit, int n_threads);
void orc_traverse_batch(const orc_node *tree, const float *xyz, int n, int32_t *best, int32_t *visits);

/* ---- A6: min/max/argmax + weight update (kernel.cu:297-304, 1327-1338) ---- */
void orc_minmax_first_f32(const float *v, int n, int *imin, int *imax);
void orc_minmax_first_i32(const int32_t *v, int n, int *imin, int *imax);
void orc_update_weights_f32(orc_particle *p, int n, const float *fit, float c, int min_trunc);
void orc_update_weights_i32(orc_particle *p, int n, const int32_t *fit, float c, int min_v);

/* ---- A7-A9: single-step ICP (kernel.cu:974-1093, svd3.h:354-401) ---- */
void orc_svd3(const float a[9], float u[9], float s[9], float v[9]);
/* dbg (optional, 31 floats): W[9] (a11..a33 as passed to svd), mu_tar[3], mu_cor[3], R[9] (glm col-major), t[3], theta, n_valid(as float),pad */
void orc_icp(const orc_node *tree, const float robot[3], const float start[3], const float *scan,
             int n_beams, float out_pose[3], float *dbg);

/* ---- A10: Bresenham free-cell raycast (kernel.cu:190-240, 524-549) ---- */
void orc_trace_ray(int sx, int sy, int ex, int ey, int dimx, int dimy, uint8_t *out);
void orc_get_walls(const float *scan, int n_beams, int cx, int cy, float theta, uint8_t *free_mask,
                   uint8_t *wall_mask, int dimx, int dimy, float res_x, float res_y);

/* ---- A11-A15: point-cloud map update (kernel.cu:1350-1540, kdtree.cpp:25-105) ---- */
/* masks -> world-coordinate point lists, x-major order (kernel.cu:1435-1461).  Returns counts. */
void orc_masks_to_points(const uint8_t *free_mask, const uint8_t *wall_mask, int dimx, int dimy,
                         const orc_patch *patch, const float robot[3], float *wall_xyzw,
                         int *n_wall, float *free_xyzw, int *n_free);
void orc_update_map_kd(orc_node *tree, const float *pts_xyzw, const int32_t *idx, int n, int val,
                       const orc_patch *patch);
void orc_test_correspondence(const orc_node *tree, const float *pts_xyzw, const int32_t *idx, int n,
                             uint8_t *create, const orc_patch *patch);
void orc_kd_insert_node(const float p[4], orc_node *list, int list_size);
/* implemented in kdtree_oracle.cpp (needs std::sort) */
void orc_kd_create(const float *pts_xyzw, int n, orc_node *list);
void orc_kd_balance(orc_node *list, int n);

/* ---- A16: resample (kernel.cu:420-511) ---- */
/* returns 1 if resampled; src_idx (optional, n ints) receives the chosen source per slot */
int orc_resample(orc_particle *p, int n, int frame, float *neff_out, int32_t *src_idx);
void orc_weighted_sample_indices(const float *cdf, int n, float neff, int frame, int i0, int count,
                                 int32_t *src_idx);

/* ---- A17/A18: 2-D occupancy grid path (kernel.cu:243-372, 513-621) ---- */
void orc_score_grid(const int8_t *grid, int dimx, int dimy, const orc_patch *patch,
                    const orc_particle *p, int n, const float *scan, int n_beams, int32_t *fit);
void orc_update_map_grid(int8_t *grid, int dimx, int dimy, const orc_patch *patch,
                         const float robot[3], const float *scan, int n_beams);

/* ---- topology graph / loop-closure proposal (kernel.cu:623-795; commented out of the shipped step at 1750-1751) ---- */
#define ORC_TOPO_MAX_NODES 4096
typedef struct {
    int n_nodes, node_idx;               /* clusters[0].nodes.size(), clusters[0].nodeIdx */
    float pos[ORC_TOPO_MAX_NODES][2];    /* Node.pos  */
    float dist[ORC_TOPO_MAX_NODES];      /* Node.dist */
    int n_edges[ORC_TOPO_MAX_NODES];
    int edges[ORC_TOPO_MAX_NODES][8];
} orc_topology;
void orc_topology_init(orc_topology *t);
/* UpdateTopology (kernel.cu:695-726): returns 1 if a node was created */
int orc_topology_update(orc_topology *t, const float robot[3]);
/* FindWalls (kernel.cu:661-693) between two world points on the 2-D grid: cells on the Bresenham ray with map > 30 */
int orc_find_walls(const int8_t *grid, int dimx, int dimy, const orc_patch *patch, const float a[2], const float b[2]);
/* CheckLoopClosure (kernel.cu:738-795): pairs (closure node j, visible node k); returns the number of pairs written */
int orc_check_loop_closure(const orc_topology *t, const int8_t *grid, int dimx, int dimy, const orc_patch *patch,
                           const float robot[3], int32_t *pairs, int cap);

/* ---- whole SLAM step (kernel.cu:1702-1762), KD / point-cloud path ---- */
typedef struct orc_slam orc_slam;
typedef struct {
    int n_particles;
    int n_beams;         /* 1081 */
    orc_patch patch;     /* 40 x 40 m, 0.025 m */
    int kd_capacity;
    int strict_host_mirror; /* H11: reproduce the half-array D2H at kernel.cu:1341 */
    int free_upload_bug;    /* H6: 0 = upload full free list, 1 = zero tail past wallPC.size() */
    int balance_period;     /* 100 (kernel.cu:1707); 0 disables */
} orc_slam_config;

orc_slam *orc_slam_create(const orc_slam_config *cfg);
void orc_slam_destroy(orc_slam *s);
void orc_slam_set_map(orc_slam *s, const orc_node *tree, int n);
void orc_slam_step(orc_slam *s, int frame, const float *scan);
/* 2-D grid variant (stages kernel.cu:400-418, 307-339, 551-577, 447-511 in the frame loop of 1702-1762) */
void orc_slam_step_grid(orc_slam *s, int frame, const float *scan);
/* run UpdateTopology + CheckLoopClosure at the end of every frame (the two calls commented out at kernel.cu:1750-1751) */
void orc_slam_set_topology(orc_slam *s, int enable);
int orc_slam_last_closures(const orc_slam *s, int32_t *pairs, int cap); /* returns the number of pairs of the last frame */
const orc_topology *orc_slam_topology(const orc_slam *s);
/* the reference's CPU branches of the same loop (GPU_* == 0; H7 semantics; timing baseline for BASELINE configs[0]) */
void orc_slam_step_grid_cpu(orc_slam *s, int frame, const float *scan);
void orc_slam_set_grid(orc_slam *s, const int8_t *grid);
const int8_t *orc_slam_grid(orc_slam *s);
void orc_slam_set_particles(orc_slam *s, const orc_particle *p); /* device array and host mirror */
void orc_slam_shift_particles(orc_slam *s, const float d[3]);       /* odometry hook: particles and robotPos += d */
void orc_slam_get_pose(const orc_slam *s, float pose[3]);
int orc_slam_kd_size(const orc_slam *s);
const orc_node *orc_slam_tree(const orc_slam *s);
const orc_particle *orc_slam_particles(const orc_slam *s);
/* per-step trace of the last orc_slam_step: best idx, resampled flag, n_wall, n_free, n_insert, neff(float bits) */
void orc_slam_last_trace(const orc_slam *s, int32_t out[8]);
/* wall/free cell index lists (x*dimx+y, ascending) of the last step */
int orc_slam_last_cells(const orc_slam *s, int which /*0 wall,1 free*/, int32_t *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
