/*
 * pfslam.h -- C-ABI of libpfslam_hip.so: the MI355X-native particle-filter SLAM
 * inner loop (disperse -> KD scan-match score -> min/max/argmax + weight
 * normalise -> single-step ICP/SVD pose -> point-cloud map update -> weighted
 * resample), the drop-in for the path the reference implements in
 * src/kernel.cu behind src/kernel.h.
 *
 * Plain pointers and sizes only: no C++ types, no torch types.  Every entry
 * point returns 0 on success, non-zero on failure (pfslam_last_error() gives
 * the message); nothing in the library calls exit() (the reference does,
 * kernel.h:42-60).  One handle = one GPU = one host thread.
 *
 * Reference interface each entry replaces (file:line under the reference repo):
 *   pfslam_create / pfslam_destroy     particleFilterInit(Scene*) / particleFilterFree()
 *                                      + particleFilterInitPC / FreePC    kernel.h:14-15,22-23 (kernel.cu:107-178,1096-1122)
 *   pfslam_step                        particleFilter(uchar4*, int frame, Lidar*)   kernel.h:16 (kernel.cu:1702-1762)
 *   pfslam_get_pose/particles/map      getPCData(...)                     kernel.h:19 (kernel.cu:803-813)
 *   pfslam_motion_update               PFMotionUpdate / kernAddNoise      kernel.cu:375-418
 *   pfslam_score_kd                    kernEvaluateParticlesKD            kernel.cu:1198-1308
 *   pfslam_measurement_update          PFMeasurementUpdateKD host logic   kernel.cu:1311-1348
 *   pfslam_icp                         transformPointICP                  kernel.cu:993-1093
 *   pfslam_update_map_kd               PFUpdateMapKD                      kernel.cu:1406-1540
 *   pfslam_resample                    PFResample / kernWeightedSample    kernel.cu:420-511
 *   pfslam_score_grid/update_map_grid  kernEvaluateParticles / PFUpdateMap kernel.cu:243-372,513-621
 *   pfslam_kd_create/insert_list/insert_node/balance   KDTree::Create/InsertList/InsertNode/Balance  kdtree.cpp:25-105
 *   pfslam_step_grid                   the frame loop of kernel.cu:1702-1762 with the 2-D stages
 *                                      PFMeasurementUpdate / PFUpdateMap  kernel.cu:307-339, 551-577
 *   pfslam_traverse                    findCorrespondenceIndexKD          kernel.cu:924-972
 *   pfslam_topology_update, find_walls,
 *   check_loop_closure, get_topology   UpdateTopology / FindWalls / CheckLoopClosure   kernel.cu:623-795
 *   pfslam_shard_disperse / score / weights / finish   particleFilter split where a multi-GPU caller places its three
 *                                      all-gathers (no reference counterpart: the reference is single-GPU)
 */
#ifndef PFSLAM_H
#define PFSLAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* KDTree::Node (kdtree.hpp:16-27): 32 bytes, children/parent are array indices, -1 = none */
typedef struct pfslam_node {
    int32_t axis, left, right, parent;
    float x, y, z, w;
} pfslam_node;

/* Particle (sceneStructs.h:33-38): 32 bytes, pos@0 (x, y, heading), w@12, cluster@16, map ptr@24 */
typedef struct pfslam_particle {
    float x, y, theta, w;
    uint8_t cluster;
    uint8_t pad_[7];
    void *map;
} pfslam_particle;

typedef struct pfslam_config {
    int32_t n_particles;   /* PARTICLE_COUNT (kernel.cu:30), runtime here; particles owned by THIS handle */
    int32_t n_beams;       /* LIDAR_SIZE (kernel.cu:43) = 1081 */
    float map_scale_x, map_scale_y; /* Patch.scale (data/map_settings.txt: 40 40) */
    float map_res_x, map_res_y;     /* Patch.resolution (0.025).  int(scale / res) must be the same in x and y: the
                                     * reference indexes cell (x, y) as x * dim.x + y (kernel.cu:120, 539, 1438), which is
                                     * only well defined on a square grid; pfslam_create refuses anything else */
    int32_t kd_capacity;   /* KD_MAX_SIZE (kernel.cu:77); nodes, at most 2^27 - 1.  A frame whose new walls do not fit inserts
                            * none of them and fails ("kd_capacity exhausted"; reported by the call that books the frame, see
                            * pfslam_step) -- the reference has no bound check at all */
    int32_t device;        /* HIP device ordinal */
    int32_t strict_host_mirror; /* 1 = reproduce the half-array weight read-back of kernel.cu:1341 (H11) */
    int32_t free_upload_bug;    /* 1 = reproduce kernel.cu:1475 (free list tail zero) (H6); 0 = full list */
    int32_t balance_period;     /* 100 = KDTree::Balance at frame%100==5 (kernel.cu:1707); 0 = never */
    /* multi-GPU particle sharding: this handle holds global particles [global_offset, global_offset+n_particles)
     * of global_n; RNG streams are keyed by the GLOBAL index so results do not depend on the sharding.
     * Layout: rank r owns [r * shard_stride, min((r + 1) * shard_stride, global_n)) -- every rank but the last is full,
     * the last may be shorter (not empty); the exchange buffers are padded to shard_stride so that all-gather counts are
     * equal on every rank. */
    int32_t global_offset;
    int32_t global_n;      /* 0 -> n_particles */
    int32_t shard_stride;  /* 0 -> n_particles (equal shards) */
    int32_t reserved_[2];
} pfslam_config;

typedef struct pfslam_handle pfslam_handle;

/* defaults of the reference: 1000 particles, 1081 beams, 40x40 m @ 0.025 m, kd_capacity = KD_MAX_SIZE = 10 M nodes,
 * balance_period 100, strict_host_mirror = 1 (H11 reproduced), free_upload_bug = 0 (H6 NOT reproduced: the full free list is
 * applied -- the reference's truncated upload reads uninitialised device memory, which has no defined result to match) */
void pfslam_default_config(pfslam_config *cfg);
int pfslam_create(const pfslam_config *cfg, pfslam_handle **out);
int pfslam_destroy(pfslam_handle *h);
const char *pfslam_last_error(void);
/* number of visible HIP devices, or <0 with the HIP error negated */
int pfslam_device_count(void);
/* launch on this HIP stream (hipStream_t as void*) instead of the handle's own */
int pfslam_set_stream(pfslam_handle *h, void *hip_stream);
/* books every frame in flight (see pfslam_step), then waits for the stream */
int pfslam_synchronize(pfslam_handle *h);

/* ---- whole step (kernel.h:16) ----
 * pfslam_step ENQUEUES the frame -- dispersion, scan-match, weights, ICP, map update including the insert of the new walls
 * (KDTree::InsertNode on the device), resample decided on the device -- copies the scan into a pinned slot of its own, and
 * then books the frame `lag` steps back (default 1): trace, pose, map size and deferred errors come from a 128-byte header the
 * device writes into pinned memory.  Successive calls therefore overlap the host with the device and leave no idle gap between
 * frames.  EVERY other entry point first books all frames in flight, so pfslam_step followed by any getter behaves like the
 * reference's synchronous particleFilter.  A deferred error (kd_capacity exhausted, cell list overflow) is returned by the call
 * that books the frame: the next pfslam_step, a getter, or pfslam_synchronize.  The first scan (it seeds the map on the host),
 * re-balance frames and handles with pfslam_set_topology(h, 1) are booked at once.
 * One deferred error is fatal for the handle: "a stream gate ... gave up waiting" (a cross-stream edge of the frame was not served
 * within 1 s: streams sharing a hardware queue, a multi-GPU peer that never arrived).  The launches behind that gate have run without
 * what they waited for, so map, weights and particles are undefined from that frame on; the error is sticky -- destroy the handle.
 * pfslam_set_lag: 0 = every pfslam_step books its own frame before it returns, up to 2 frames in flight. */
int pfslam_step(pfslam_handle *h, int frame, const float *scan_host);
int pfslam_set_lag(pfslam_handle *h, int frames);
/* The same frame loop with the reference's 2-D occupancy-grid stages (PFMotionUpdate kernel.cu:400-418,
 * PFMeasurementUpdate 307-339, PFUpdateMap 551-577, PFResample 447-511): pose = best particle, grid updated
 * in place.  The grid starts at -100 everywhere (kernel.cu:124) unless pfslam_set_grid replaced it. */
int pfslam_step_grid(pfslam_handle *h, int frame, const float *scan_host);

/* ---- read-back (kernel.h:19): non-owning pointers into the handle's host mirrors, valid until the next call */
int pfslam_get_pose(pfslam_handle *h, float pose[3]);
int pfslam_get_particles(pfslam_handle *h, const pfslam_particle **out, int *n);
int pfslam_get_map(pfslam_handle *h, const pfslam_node **out, int *n);
int pfslam_get_grid(pfslam_handle *h, const int8_t **grid, int *dimx, int *dimy);
/* per-step trace of the last pfslam_step: [best, resampled, n_wall, n_free, n_insert, neff(float bits), kd_size, 0] */
int pfslam_get_trace(pfslam_handle *h, int32_t out[8]);
/* ascending cell indices (x*dimx+y) of the last map update; which: 0 = wall, 1 = free. Returns count via *n. */
int pfslam_get_cells(pfslam_handle *h, int which, int32_t *out, int cap, int *n);

/* ---- state upload ---- */
int pfslam_set_map(pfslam_handle *h, const pfslam_node *nodes, int n);
int pfslam_set_particles(pfslam_handle *h, const pfslam_particle *p, int n);
int pfslam_set_scan(pfslam_handle *h, const float *scan_host, int n_beams);
int pfslam_set_pose(pfslam_handle *h, const float pose[3]);
int pfslam_set_grid(pfslam_handle *h, const int8_t *grid, int dimx, int dimy);

/* ---- stage entry points (operate on the handle's device-resident state) ---- */
int pfslam_motion_update(pfslam_handle *h, int frame);
/* odometry hook (no reference counterpart: the reference's filter has no motion model besides the diffusion of kernel.cu:375-397):
 * every pose the filter holds -- all particles and robotPos -- moves by delta = (dx, dy, dtheta), one float addition per
 * component.  Enqueued behind the frames in flight, no host wait.  Sharded handles: call it on every rank. */
int pfslam_shift_particles(pfslam_handle *h, const float delta[3]);
/* fit_host may be NULL (result stays on the device; no synchronisation) */
int pfslam_score_kd(pfslam_handle *h, float *fit_host);
/* min/max/first-argmax of fit + weight update; outputs may be NULL */
int pfslam_measurement_update(pfslam_handle *h, int *best, float *fmin, float *fmax);
/* ICP around the handle's current pose (the PREVIOUS robotPos, kernel.cu:1013) from `start`; dbg29 optional:
 * A[9], mu_tar[3], mu_cor[3], R[9], t[3], theta, n_valid */
int pfslam_icp(pfslam_handle *h, const float start[3], float pose_out[3], float *dbg29);
int pfslam_update_map_kd(pfslam_handle *h);
int pfslam_resample(pfslam_handle *h, int frame, int *resampled, float *neff);
/* the two halves of pfslam_resample for callers that drive the STAGES of a sharded handle themselves (the sharded frame
 * pfslam_shard_* needs neither): plan = Neff + cdf + source indices on the GLOBAL weights (device buffer 10, filled by the
 * caller's all-gather); gather = pull the chosen particles out of the GLOBAL pose blocks (device buffer 17, all-gathered by the
 * caller from buffer 16 when plan reports resampled = 1) */
int pfslam_resample_plan(pfslam_handle *h, int frame, int *resampled, float *neff);
int pfslam_resample_gather(pfslam_handle *h);
int pfslam_score_grid(pfslam_handle *h, int32_t *fit_host);
int pfslam_update_map_grid(pfslam_handle *h);

/* batch KD "nearest neighbour" with the reference traversal (findCorrespondenceIndexKD, kernel.cu:924-972);
 * xyz_host: n*3 floats; best_host: n ints */
int pfslam_traverse(pfslam_handle *h, const float *xyz_host, int n, int32_t *best_host);

/* ---- topology graph / loop-closure proposal (UpdateTopology, FindWalls, CheckLoopClosure: kernel.cu:623-795).
 * The reference leaves both calls commented out of its step (kernel.cu:1750-1751) and discards the clusters it builds
 * (kernel.cu:776), so they are explicit entry points, and part of the frame loops only after pfslam_set_topology(h, 1).
 * They act on the current robot pose
 * and on the 2-D occupancy grid (pfslam_set_grid / pfslam_update_map_grid).
 * topology_update: adds a graph node when the pose is > 2.5 m from every node; *n_nodes receives the node count.
 * find_walls: cells with occupancy > 30 on the Bresenham ray between two world points (exact count; the reference's
 *   CheckVisibility accumulates non-atomically).
 * check_loop_closure: pairs (candidate node j, visible node k) for every node j closer than 6 m on the map and
 *   farther than 20 m along the graph; returns the pair count in *n (pairs beyond cap are counted, not written).
 * get_topology: nodes as (x, y, dist) triples; *node_idx = index of the current node. */
/* pfslam_set_topology(h, 2): the same calls, made when the frame is BOOKED (`lag` steps after it was enqueued, see pfslam_step)
 *   instead of at once: KD frames stay in flight (their visibility test reads the 2-D grid, which KD frames never write; the counts
 *   run on a stream of their own), 2-D frames are still booked at once.  The graph and every frame's proposals are unchanged;
 *   pfslam_get_closures returns those of the last booked frame (any getter books everything first).
 * pfslam_set_topology(h, 1): pfslam_step and pfslam_step_grid then run UpdateTopology and CheckLoopClosure at the end of every
 *   frame, exactly where the reference has the two calls commented out (kernel.cu:1750-1751); pfslam_get_closures returns the
 *   pairs the LAST frame proposed (count in *n; pairs beyond cap are counted, not written).  In the KD frame loop FindWalls
 *   reads the 2-D grid the KD path never updates (dev_occupancyGrid, kernel.cu:680: all -100, every node visible); in the 2-D
 *   frame loop it reads the live map. */
int pfslam_set_topology(pfslam_handle *h, int enable);
int pfslam_get_closures(pfslam_handle *h, int32_t *pairs, int cap, int *n);
int pfslam_topology_update(pfslam_handle *h, int *n_nodes);
int pfslam_find_walls(pfslam_handle *h, const float a_xy[2], const float b_xy[2], int *n_walls);
int pfslam_check_loop_closure(pfslam_handle *h, int32_t *pairs, int cap, int *n);
int pfslam_get_topology(pfslam_handle *h, float *nodes_xyd, int cap, int *n, int *node_idx);

/* the frame % balance_period == 5 re-balance that pfslam_step performs first (kernel.cu:1707-1711), as its own
 * entry for callers that drive the stages themselves; pfslam_kd_size = number of map nodes (kdSize) */
int pfslam_maybe_balance(pfslam_handle *h, int frame);
int pfslam_kd_size(pfslam_handle *h);

/* ---- multi-GPU (particles sharded over ranks; the collectives are the caller's, e.g. RCCL over xGMI) ----
 * The sharded frame.  Like pfslam_step it is only ENQUEUED: no call waits for the device, the frame is booked one step later
 * from its pinned header, and the schedule of the caller's collectives is FIXED -- three all-gathers in every frame, none of
 * them data-dependent (the resample is decided on the device, on the gathered weights, and reads its sources from the gathered
 * pose blocks).  The four calls are the four parts of the very frame pfslam_step enqueues, and every collective goes straight into
 * one of that frame's streams -- pfslam_shard_stream names it; stream order is all the ordering the caller has to provide (no event,
 * no stream of the caller's own: launch the collective with that stream as its stream argument):
 *   pfslam_shard_disperse   scan upload, re-balance if due; dispersion and lane order of this shard; the cells' passes; the ICP solve.
 *                           *seeded = 1 when the frame only seeded the map (first scan): nothing else to do for this frame.
 *   [all-gather buffer 16 -> buffer 17, stream 0]   pose blocks [x | y | theta] in ONE piece of 3 * shard_stride floats per rank.  They
 *                           are final right after the dispersion; the stream has nothing else to do until the reduce, so the
 *                           collective runs UNDER the scan-match kernel
 *   pfslam_shard_score      scan-match, reduce -> this rank's 16-byte record in buffer 14: its packed {int64 max key, int64 negated-min
 *                           key}; key = (orderable_u32(fit) << 32) | (0xFFFFFFFF - global_index)
 *   [all-gather buffer 14 -> buffer 15, stream 1]   16 bytes per rank: the one collective on the frame's critical chain
 *   pfslam_shard_weights    global min / max / first argmax from the gathered keys; the best particle's pose is read out of the gathered
 *                           pose blocks; pose = best particle + increment of the ICP solve; walls at that pose, insert, cell rows;
 *                           weight update of this shard
 *   [all-gather buffer 5 -> buffer 10, stream 0]    weights, shard_stride floats per rank
 *   pfslam_shard_finish     sums + Neff on the gathered weights, frame header, gated resample (sources from buffer 17); the free
 *                           cells' chain of the replicated map update; booking of the frame `lag` steps back
 * Results are bit-identical for any number of ranks.  Buffers 14 and 16 alternate between two allocations from frame to frame: query
 * 16 after pfslam_shard_disperse, 14 after pfslam_shard_score of the same frame.
 * A handle that holds ALL particles (global_n == n_particles) may be driven through the same calls (world 1): buffers 10 / 17
 * then alias 5 / 16, nothing reads buffer 15, and there is no collective to issue. */
int pfslam_shard_disperse(pfslam_handle *h, int frame, const float *scan_host, int *seeded);
int pfslam_shard_score(pfslam_handle *h);
int pfslam_shard_weights(pfslam_handle *h);
int pfslam_shard_finish(pfslam_handle *h);
/* the stream collective `which` (0 pose blocks, 1 keys, 2 weights) of the frame being enqueued is to be issued on: a hipStream_t */
int pfslam_shard_stream(pfslam_handle *h, int which, void **hip_stream);
/* Multi-GPU re-balance, ONE host build per node (the ranks hold identical maps): after pfslam_set_shard_balance(h, 1) the sharded
 * frame does not re-balance by itself.  In front of pfslam_shard_disperse every rank calls pfslam_shard_balance_due (books the frames
 * in flight when the period hits; *n_nodes = map size).  If due: the root rank calls pfslam_shard_balance_build (KDTree::Balance,
 * kdtree.cpp:31-40, on the host with all usable cores), every rank broadcasts pfslam_device_ptr buffers 20 (hot records, 16 B per
 * node), 21 (parents), 22 (z / z-level links), 23 (weights; 4 B per node each: the first n_nodes entries) and 24 (16 bytes of
 * state) from the root on the handle's stream, and the other ranks call pfslam_shard_balance_adopt. */
int pfslam_set_shard_balance(pfslam_handle *h, int external);
int pfslam_shard_balance_due(pfslam_handle *h, int frame, int *due, int *n_nodes);
int pfslam_shard_balance_build(pfslam_handle *h, int frame);
int pfslam_shard_balance_adopt(pfslam_handle *h);
/* stage-level merge hooks (the sharded frame above does not need them): local packed keys into the stats buffer 0
 * ([0] max key, [1] negated-min key: MAX-reduce across ranks), then weights + this rank's share of the best pose in buffer 8
 * (zero on non-owners: SUM-reduce across ranks) */
int pfslam_measurement_local(pfslam_handle *h);  /* score must have run; fills the stats buffer */
int pfslam_measurement_apply(pfslam_handle *h, int *best_global, float *fmin, float *fmax); /* after the MAX merge */
/* device pointers of the handle's buffers, for zero-copy wrapping by the harness.
 * which: 0 stats (8 x i64), 1 fit (n x f32), 2 x, 3 y, 4 theta (n x f32 each; inside buffer 16), 5 w (shard_stride x f32, the
 *        first n valid), 6 weight tile sums, 7 scan (n_beams x f32), 8 best-particle pose (4 x f32), 9 robot pose (4 x f32),
 *        10 global w (world * shard_stride x f32, rank-major, the first global_n valid; aliases 5 when unsharded),
 *        14 this rank's 16-byte measurement record (its packed max / negated-min keys), 15 the gathered records (world x 16 B),
 *        16 local pose block [x | y | theta] (3 * shard_stride x f32; moves when a resample swaps the double buffer),
 *        17 global pose blocks (world x 3 * shard_stride x f32, rank-major; aliases 16 when unsharded) */
int pfslam_device_ptr(pfslam_handle *h, int which, void **ptr, size_t *bytes);

/* ---- measurement support ----
 * pfslam_time_score_kd: `iters` back-to-back scoring passes (lane order + scan-match kernel + partial-sum reduce, i.e. all of
 *   pfslam_score_kd's device work) between two HIP events on the handle's stream; average milliseconds per pass.
 * pfslam_set_timing: 0 = off; 1 = HIP events on the handle's stream bracket every launch of the scan-match kernel (only that
 *   kernel) inside pfslam_step / pfslam_shard_begin; 2 = additionally the four phases the reference times per frame
 *   (kernel.cu:1727-1760: motion, measurement incl. ICP, map update incl. its host part, resample).  Resets the accumulators.
 * pfslam_get_timers -> out[2k] = total ms, out[2k+1] = count, k = 0 scan-match kernel, 1 motion, 2 measurement, 3 map,
 *   4 resample, 5 the planning launches that precede the scan-match kernel (pose boxes + shared-prefix plan).
 * pfslam_score_census: what one scoring launch on the handle's CURRENT particles, scan and map issues, counted by a counting
 *   instantiation of the same kernel with the same launch shape and lane order: out[0] wave-level trips of the descent loop
 *   (= wave-level 16-byte gathers of node records), out[1] active lanes in them (= node visits), out[2] wave-level
 *   parent-hyperplane tests (one 4-byte + one 16-byte wave gather each), out[3] lanes in them, out[4] trips in which every
 *   active lane stood on the same node, out[5] those of them on the common path of all 64 lanes from the root.
 * pfslam_set_variant: how the scoring pass is organised (results are bit-identical; A/B measurements and tests): 0 = default
 *   (lanes along a Hilbert curve: counting sort over cells of the cloud; from ~4.6 k particles on a planar map: lattice-cell rows
 *   when every map point lies on the lattice k * resolution -- true of every map the SLAM step builds --, the round-2
 *   shared-prefix plan otherwise), 1 = identity lane order, 2 = the plain per-lane traversal, 3 = cell rows / plan at any
 *   particle count, 4 = like 3 but always the shared-prefix plan.  The environment variable PFSLAM_VARIANT sets the initial value. */
int pfslam_time_score_kd(pfslam_handle *h, int iters, float *ms_per_launch);
int pfslam_set_timing(pfslam_handle *h, int enable);
int pfslam_get_timers(pfslam_handle *h, double out[12]);
int pfslam_score_census(pfslam_handle *h, unsigned long long out[8]);
/* census log: while enabled, every scoring pass of pfslam_step / pfslam_shard_score / pfslam_score_kd is followed by the counting
 * instantiation of the scan-match kernel on the very SAME inputs (particles, scan, map, lane order, plan) -- one record of eight
 * counters (the layout of pfslam_score_census) per pass, up to 1024 passes.  bench.py replays its timed frames on a second
 * handle with the log on: the frame loop is deterministic, so the replay's launches are the timed launches.
 * pfslam_get_census_log: *n = passes logged, out[k * 8 + c] for k < min(*n, cap). */
int pfslam_set_census(pfslam_handle *h, int enable);
int pfslam_get_census_log(pfslam_handle *h, unsigned long long *out, int cap, int *n);
/* the chip's wave-level 16-byte gather rate measured live by a micro-benchmark (cache-resident 2 MB table, 8 waves per SIMD):
 * out[0] = wave gathers per second (whole chip), out[1] = compute units, out[2] = nominal clock in GHz, out[3] = cycles per
 * wave gather per CU at the nominal clock.  The scan-match kernel issues one such gather per node visit of a wave. */
int pfslam_ubench_gather(pfslam_handle *h, double out[4]);
/* the shared-prefix plan of the LAST scoring pass (kd_device.h): out[0] rows (waves x beams), [1] mean length of the root path
 * common to a wave's 64 queries, [2] mean candidates kept of it, [3] fraction of rows whose first descent is complete,
 * [4] fraction without a plan, [5] fraction with a full candidate list, [6..8] mean extent of a wave's pose box in x, y (m) and
 * heading (rad), [9] waves.  All zero when no plan was made (non-planar map, few particles, variant 2). */
int pfslam_plan_stats(pfslam_handle *h, double out[10]);
/* the persistent lattice-cell rows (csrc/kd_cells.hip.inc): out[0] lattice cells claimed since the last wipe, [1] live rows (one per
 * sub-cell: up to four per cell), [2] mean first-descent candidates per row, [3] mean re-descent candidates per row, [4] sub-cells
 * without a row, [5] 16-byte pool slots used, [6] [7] lattice index of the window's corner cell, [8 .. 11] since the last wipe:
 * cells walked from the root / extensions (one of the cell's links had gained a node) / looks that found a cell unchanged / cells
 * claimed, [12] device flags (1 list full, 2 pool full, 8 cloud far from the window centre), [13] publishing updates since the
 * last wipe (divide [9] and [10] by it for per-frame figures), [14] wipes so far, [15] 1 = suspended: the list / pool overflowed
 * twice within 16 frames (a cloud too wide for the table), the round-2 plan scores until the map is replaced or re-balanced.
 * [0 .. 13] are zero when the last scoring pass did not use cell rows. */
int pfslam_cell_stats(pfslam_handle *h, double out[16]);
/* mean duration (ms) of the 2-D scan-match kernel alone and of the whole 2-D scoring pass (kernel + partial sums + min / max / argmax +
 * weights) over `iters` launches each, HIP events on the handle's stream */
int pfslam_time_score_grid(pfslam_handle *h, int iters, float *ms_kernel, float *ms_pass);
int pfslam_set_variant(pfslam_handle *h, int variant);
/* Transcendentals.  0 (default) = the specification of csrc/pf_math.h: fixed sequences of IEEE double operations rounded once, which the
 * CPU oracle follows bit for bit.  1 = the device library's cosf / sinf in CleanLidarScan (kernel.cu:182-187) and erfcinvf in the
 * dispersion (kernel.cu:375-397) -- what the reference's own text compiles to on this platform: with it the product's kernels equal the
 * reference's kernels built for gfx950 with zero mismatches (tests/test_gpu_ref_kernels.py); results then differ from the oracle's in
 * the last place of an end point now and then. */
int pfslam_set_trig(pfslam_handle *h, int devlib);
/* ---- round-5 frame loop: test and measurement support (no reference counterpart) ----
 * pfslam_set_serial(h, 1): every launch of every frame on ONE stream, in the order the four chains of a frame are enqueued (what the
 * environment variable PFSLAM_SERIAL=1 sets at creation).  Results and the cell rows' bookkeeping are the same as with the chains on their
 * own streams; tests/test_gpu_frame.py steps the same cases both ways and compares.
 * pfslam_debug_check_cells: the invariants of the persistent cell rows checked on the device, with no frame in flight -- out[0] records,
 * [1] never walked, [2] walked and not yet published, [3] published, [4] row words in the table, [5] pending words, [6] fallback words,
 * [7] records without rows; VIOLATIONS (all zero or the bookkeeping is broken): [8] claimed cell neither walked nor pending, [9] unpublished
 * record whose claim word is not pending (or a malformed word), [10] row outside its record's pool allocation, [11] row slot that is not a
 * candidate of its record (or out of order), [12] a watched link that has gained a node and was not extended (a stale row waiting to
 * happen), [13] malformed link word, [14] record without rows that has one, [15] the table does not hold what the counters say.
 * pfslam_set_probe(h, frames) / pfslam_get_probe: the first thread of every launch of a round-5 frame stores the 100 MHz wall clock;
 * out[f][32] for the last n tickets (0 = that launch did not run), slot names from pfslam_probe_name. */
int pfslam_set_serial(pfslam_handle *h, int serial);
int pfslam_debug_check_cells(pfslam_handle *h, long long out[16]);
int pfslam_set_probe(pfslam_handle *h, int frames);
int pfslam_get_probe(pfslam_handle *h, unsigned long long *out, int cap_frames, int *n_frames, int *last_ticket);
const char *pfslam_probe_name(int slot);
/* out[0] 1 = the last frame ran as a round-5 frame, [1] 1 = its cross-stream edges are gates (0 = events), [2] 1 = one-stream mode, [3] publication lag */
int pfslam_frame_mode(pfslam_handle *h, int out[4]);

/* ---- host-side map structure (kdtree.cpp counterpart; no GPU needed) ---- */
int pfslam_kd_create(const float *pts_xyzw, int n, pfslam_node *out);
/* KDTree::InsertList (kdtree.cpp:46-67): sub-tree of n points in pre-order at list[idx ...] below `parent` (-1: root) */
int pfslam_kd_insert_list(const float *pts_xyzw, int n, pfslam_node *list, int idx, int parent);
int pfslam_kd_insert_node(const float p[4], pfslam_node *list, int list_size);
int pfslam_kd_balance(pfslam_node *list, int n);
/* 1 = the host build sorts on several threads (libstdc++ only; a start-up self-check against std::sort on a heavily tied array
 * must have passed), 0 = plain std::sort.  Either way the tree is std::sort's. */
int pfslam_kd_parallel_sort(void);
/* threads a host-side build may use: usable cores (scheduler affinity, cgroup quota) / LOCAL_WORLD_SIZE; PFSLAM_SORT_THREADS overrides */
int pfslam_kd_sort_threads(void);
/* 1: lift the LOCAL_WORLD_SIZE split (the caller is the node's only builder while the other ranks wait: pfslam_shard_balance_build
 * does this around its build), 0: back to the per-rank share.  No reference counterpart (the reference is single-process). */
void pfslam_kd_whole_node(int on);

#ifdef __cplusplus
}
#endif
#endif
