// pfslam_hip.hip -- libpfslam_hip.so: gfx950 kernels + the C-ABI of include/pfslam.h.
//
// MI355X-native implementation of the particle-filter SLAM inner loop the reference implements
// in src/kernel.cu.  Written for CDNA4 only (wave64, 256 CUs / 8 XCDs, 160 KB LDS per CU):
// no CUDA compatibility layer, no CPU fallback -- every entry point that needs the GPU fails
// loudly without one.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see build.py).
#include "../../include/pfslam.h"
#include "kd_device.h"
#include "pf_math.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <deque>
#include <thread>
#include <cmath>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

// ------------------------------------------------------------------------------------------
// constants of the reference (kernel.cu:30-52)
// ------------------------------------------------------------------------------------------
#define PF_LIDAR_RANGE 20.0f
#define PF_RANGE_NEVER 28.4f /* > 20 * sqrt(2) with room for the 1-ulp sincos and the product's rounding */
#define PF_FREE_WEIGHT (-1)
#define PF_OCCUPIED_WEIGHT 4
#define PF_EFFECTIVE_PARTICLES .7
#define PF_CLAMP_VAL 113.0f /* (1 << 7) - 15, kernel.cu:518,1355 */
#define PF_SVD_EPSILON 0.00001f
#define PF_SUM_TILE 4096
#define PF_SCAN_TILE 1024
#define PF_SCAN_CHUNK 16
enum { PF_T_SCORE = 0, PF_T_MOTION, PF_T_MEASURE, PF_T_MAP, PF_T_RESAMPLE, PF_T_PLAN, PF_TIMER_SLOTS };
#define PF_CENSUS_LOG 1024 /* scoring passes pfslam_set_census keeps a record of */
#define PF_KD_MAX_NODES ((1 << 27) - 1) /* idx << 4 must fit the 0x7ffffff0-byte buffer descriptor; links are 30-bit */

// Environment switches.  What the library reads with getenv() is the documented set (tools/README.md): test switches (one-stream mode,
// events instead of gates, fault injection, canonical lane order, capacities of the overflow tests, the variant, the host build's threads).
// The A/B knobs of past experiments (tools/experiments/*: priorities, workgroup counts, chains kept for comparison) go through ab_env()
// and exist only in a -DPF_EXPERIMENTS build (PFSLAM_EXTRA_FLAGS=-DPF_EXPERIMENTS python gpu-icp-slam_amd/build.py): the product takes
// the defaults they lost their A/B to, and the branches behind them fold away.
#ifdef PF_EXPERIMENTS
static inline const char *ab_env(const char *name) { return getenv(name); }
#else
static inline const char *ab_env(const char *) { return nullptr; }
#endif

// How long a stream gate (pfslam_frame.hip.inc) spins before it gives up, in ticks of the 100 MHz wall clock: 1 s for a handle on its own
// (nothing it waits for takes a millisecond); 60 s once the process holds a shard of a multi-GPU job -- there a gate may sit behind a
// collective, i.e. behind the slowest peer (a rank that prints its report, a host re-balance, a page fault).
__device__ unsigned long long g_gate_ticks = 100000000ull;
#define PF_GATE_TICKS_SHARDED 6000000000ull

static thread_local std::string g_err;
static int fail(const std::string &m)
{
    g_err = m;
    return 1;
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e__ = (expr);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e__) + " (" + __FILE__ + ":" +   \
                        std::to_string(__LINE__) + ")");                                               \
    } while (0)
#define CHK(expr)                                                                                      \
    do {                                                                                               \
        int r__ = (expr);                                                                              \
        if (r__) return r__;                                                                           \
    } while (0)

// ------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------
// Everything the host needs from a frame: written by the device at the end of the frame's map-update chain straight into
// pinned host memory (`seq` last, after a system-scope fence), read by the host one frame later -- or at once by any getter.
struct HostHeader {
    int32_t n_wall, n_free, n_new, kd_size;
    float r, r2, neff;
    int32_t flags; // PF_HDR_*
    float pose[4];
    int64_t stats[2];
    int32_t seq; // ticket of the frame that wrote this slot
    float cloud_sigma; // spread of the particle cloud as the frame's lane order saw it (m; max of x, y, reach x heading)
    float theta_max;   // largest |heading| among the frame's particles (inf: a NaN among them); 0 when the pass made no pose boxes
    int32_t seq2;      // round-5 frame loop: ticket of the frame whose PARTICLE chain wrote r, r2, neff, stats, cloud_sigma, theta_max
                       // (`seq` then covers the map update's fields only; the host waits for both)
    int32_t pad[12];
};
static_assert(sizeof(HostHeader) == 128, "HostHeader is two 64-byte lines");
#define PF_HDR_SLOTS 4 /* header / scan staging slots: frames in flight + 1 (PF_MAX_LAG + 2) */
#define PF_MAX_LAG 2

struct pfslam_handle {
    pfslam_config cfg;
    int n = 0, nb = 0, dimx = 0, dimy = 0;
    int gn = 0, goff = 0;
    int stride = 0, world = 1, rank = 0; // shard layout: rank r owns global particles [r * stride, min((r + 1) * stride, gn))
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int variant = 0;
    // particles, SoA: one block [x | y | theta] of 3 * stride floats (a single all-gather moves all three), double-buffered
    // for the resample gather; w has `stride` slots too (equal all-gather counts on every rank; the pad stays zero)
    float *pblk = nullptr, *pblk2 = nullptr;
    float *x = nullptr, *y = nullptr, *th = nullptr, *w = nullptr, *wm = nullptr;
    float *x2 = nullptr, *y2 = nullptr, *th2 = nullptr;
    // global views for the resample (all ranks' particles, rank-major): weights [world * stride], pose blocks
    // [world][x | y | theta]; they alias the local arrays when not sharded
    float *gw = nullptr, *gpose = nullptr;
    bool own_global = false;
    // sharded measurement merge: a rank's record is its packed {max key, negated-min key} (16 bytes: h->stats / the frame ring's
    // fstats by ticket parity); gkeys = the all-gathered records of every rank (world x 16 bytes), mkeys = their merge (-> header),
    // gtheta = largest |heading| over every rank's pose block, ring of 4 by ticket (k_theta_gmax)
    long long *gkeys = nullptr, *mkeys = nullptr;
    float *gtheta = nullptr;
    float *scan = nullptr;      // the current frame's scan on the device: one of the PF_HDR_SLOTS slots of scan_base
    float *scan_base = nullptr; // (the map update of frame t may still read its scan while frame t + 1's is uploaded)
    // map
    int kd_size = 0, kd_cap = 0, planar = 1;
    uint4 *hot = nullptr;
    int *parent = nullptr;
    float *kz = nullptr, *kw = nullptr;
    std::vector<pfslam_node> h_nodes; // host mirror (topology + positions; w refreshed on demand)
    bool integral_w = true; // every map weight is an integer (true for every map the SLAM step itself produces)
    float w_absmax = 113.0f; // largest |weight| the map can hold: of the uploaded map, or the clamp of the map update (113) if that is larger
    // scoring
    float *fit = nullptr, *partial = nullptr;
    size_t partial_elems = 0;
    // space-filling-curve processing order of the particles (performance only; results do not depend on it)
    unsigned *mkey = nullptr; // Hilbert cell of every particle
    int *order2 = nullptr;    // lane -> particle
    int *cells = nullptr; // counting sort of the lane order: [2^18 cell counts | 2^18 cursors | 256 tile totals]
    int64_t *stats = nullptr;
    float *pose = nullptr;  // device robotPos[4]
    float *start = nullptr; // device best-particle pose [4]
    float h_pose[3] = {0, 0, 0};
    // ICP scratch
    float *icp_tar = nullptr, *icp_cor = nullptr, *icp_dbg = nullptr;
    // map update scratch
    uint8_t *free_mask = nullptr, *wall_mask = nullptr;
    int *blk_cnt = nullptr; // per block wall/free counts then offsets
    int *wall_cell = nullptr, *free_cell = nullptr;
    float4 *wall_pts = nullptr, *free_pts = nullptr;
    int *wall_c = nullptr, *free_c = nullptr;
    int *wall_leaf = nullptr; // link of the tree every wall point would be inserted on (node * 2 + right)
    int *counts = nullptr;    // [0] n_wall [1] n_free [2] n_new [3] header flags of the last k_test_new
    int *kd_state = nullptr;  // [0] map size, device side: the insert happens there (k_test_new)
    int max_free = 0, max_wall = 0;
    // resample scratch
    float *tile_r = nullptr, *tile_r2 = nullptr, *sums = nullptr; // sums: [r, r2, neff]
    float *cdf = nullptr, *chunk_max = nullptr, *tile_tot = nullptr, *tile_off = nullptr, *tile_pmax = nullptr;
    int *src = nullptr;
    // grid path
    int8_t *grid = nullptr;
    int32_t *fit_i = nullptr;
    std::vector<int8_t> h_grid;
    // topology graph of cluster 0 (kernel.cu:147-157): node (x, y, dist), adjacency, current node
    struct TopoNode { float x, y, dist; };
    std::vector<TopoNode> topo_nodes{TopoNode{0.0f, 0.0f, 0.0f}};
    std::vector<std::vector<unsigned>> topo_edges{std::vector<unsigned>()};
    unsigned topo_idx = 0;
    int *d_count = nullptr;
    int topo_in_step = 0;               // pfslam_set_topology: UpdateTopology + CheckLoopClosure at the end of every frame (1) / with its booking (2)
    hipStream_t topo_stream = nullptr;  // mode 2: the visibility counts of a frame being booked do not wait for the frames behind it
    std::vector<int32_t> frame_closures; // (candidate node, visible node) pairs proposed by the last frame
    // Frame pipeline.  A frame's kernels need nothing from the host (the map insert and the resample decision are taken on
    // the device), so pfslam_step only ENQUEUES frame t and then reads the header of frame t - lag: the host runs ahead of the
    // device, there is no idle gap between frames.  Any other entry point settles the frames in flight first (settle()).
    HostHeader *h_hdr = nullptr, *hdr_dev = nullptr; // PF_HDR_SLOTS pinned headers and their device view
    float *h_scan = nullptr;                         // PF_HDR_SLOTS pinned scan staging buffers
    struct Frame { int seq, frame, kind; bool boxes = false; bool v2 = false; bool gtheta = false; }; // kind 0 = KD step, 1 = 2-D step, 2 = seed / sharded (settled at once); boxes: its scoring pass made pose boxes (theta_max of its header is this frame's)
    std::deque<Frame> in_flight;
    int seq = 0; // tickets handed out
    int cur_seq = 0, cur_frame = 0; // the frame being enqueued (frame_front .. frame_tail)
    int shard_stage = 0;            // sharded frame: 1 dispersed, 2 scored, 3 weights done (call order check)
    bool shard_v2 = false;          // the sharded frame being enqueued is a cut of the round-5 frame (else: the staged round-4 chain, every collective on the handle's stream)
    bool shard_call = false;        // the frame being enqueued came in through pfslam_shard_*
    bool theta_global_known = true; // sharded jobs: a header with the job-wide |heading| maximum has been booked since the particles were last set
    struct FrameV2 { int used = 0, bpc = 0, frame = 0; bool sync_cells = false, serial = false, sharded = false, gates = false, ftail_gate = false; } fv; // the round-5 frame being enqueued, between its parts
    bool gates_tested = false;      // the self-test of the four streams has run since they last changed (pfslam_set_stream)
    bool gates_ok = false;          // ... and they make progress independently of one another
    bool gates_live = false;        // the frames in flight use gates (a change of mode drains the pipeline first)
    hipEvent_t ev_poseg = nullptr;  // events mode, sharded: the gathered pose blocks are there (P -> C)
    int lag = 1; // frames the host may run ahead (PFSLAM_LAG; 0 = every step settles itself)
    float scan_reach = 8.0f; // mean in-range beam length of the current scan (m): the lever arm of a heading difference
    std::vector<pfslam_particle> h_particles;
    std::vector<float> h_tmp;
    int32_t trace[8] = {0};
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // ICP needs only the scan, the previous pose and the map (kernel.cu:974-1075): it runs on `aux` under the score kernel
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_mapfork = nullptr, ev_map = nullptr; // map update of a frame on the aux stream (join_map)
    bool map_forked = false;
    bool scan_front_done = false; // tile totals and offsets of the resample scan were produced with the weight sums of this frame
    bool header_packed = false;   // k_test_new already filled the frame's HostHeader
    bool lds_attr_set = false;    // k_test_new's dynamic LDS limit raised (scans of more than 1536 beams)
    bool mirror_stale = false;    // the device has inserted nodes since h_nodes was last made current
    int mirror_n = 0;             // nodes [0, mirror_n) of h_nodes are current except for links and weights (see pfslam_get_map)
    bool icp_forked = false;      // aux work in flight (between fork_icp and join_icp)
    bool icp_delta_ready = false; // joined: icp_dbg[24..27] holds this frame's increment, not yet added to the best particle
    bool masks_cleared = false;   // the aux stream already zeroed the free / wall masks for this frame
    bool stats_clean = false;     // ... and reset the min/max keys
    // live timing of the dominant kernel inside pfslam_step (bench.py roofline leg)
    // timing == 1: HIP events bracket every k_score_kd launch (bench.py roofline leg).  timing == 2: also the four phases the
    // reference times per frame (kernel.cu:1727-1760): motion / measurement / map / resample.
    int timing = 0;
    struct TimedSpan { hipEvent_t a, b; int slot; bool keep_b = false; }; // keep_b: b is also the start of the next span
    std::vector<hipEvent_t> ev_pool;
    std::vector<TimedSpan> ev_pending;
    hipEvent_t phase_ev = nullptr; // start of the phase being timed
    double timer_ms[PF_TIMER_SLOTS] = {0};
    long timer_count[PF_TIMER_SLOTS] = {0};
    pf::KdCensus *d_census = nullptr;
    pf::KdCensus *census_store = nullptr, *census_log = nullptr; // pfslam_set_census: one record per scoring pass
    int census_n = 0;
    // shared-prefix plan of the score kernel: one row per (wave of 64 lanes, beam), pose box per wave
    pf::KdPlanRow *plan = nullptr;
    size_t plan_rows = 0;
    bool plan_valid = false; // the last scoring pass made a plan
    pf::KdGroupBox *group_box = nullptr;
    pf::AngleParts *group_parts = nullptr; // cos / sin of every group's centre heading (k_group_box -> k_cells_mark)
    // lattice-cell rows (kd_cells.hip.inc)
    bool lattice_ok = false;   // planar map, every node on the lattice k * res of the config (all maps the SLAM step builds are)
    bool cells_valid = false;  // the last scoring pass used cell rows
    unsigned *cell_tab = nullptr;
    int *cell_list = nullptr, *cell_state = nullptr, *cell_rec = nullptr;
    int *cell_touched = nullptr; // [0] count, then the links that gained a node in the last k_test_new
    uint4 *cell_pool = nullptr;
    // the rows persist across frames (kd_cells.hip.inc): wiped when the map is replaced (set_map, re-balance) or the device asks for
    // it in a frame's header (list / pool exhausted, cloud far from the window centre)
    bool cells_wipe_pending = false;
    // spread of the cloud (k_cell_count -> frame header -> here, one frame late; set_particles estimates it on the host): the cell rows'
    // marking pass costs the AREA of the waves' beam-end boxes in lattice cells, so a wide cloud is scored with the round-2 plan
    float cloud_sigma = 0.0f;
    float *d_sigma = nullptr; // [0] spread  [2 .. 65] partial maxima of |heading| (k_group_box)
    // upper bound of |heading| over the particles, for the choice of the scan-match kernel's instantiation (sincos_sum_spec<GUARD>): exact
    // from set_particles, + 0.1 per dispersion (three draws of at most 6 sigma x 0.01 rad ... generously), + |d theta| per odometry
    // shift, replaced by the device's own maximum + 1 whenever a frame's header comes in
    float theta_bound = 0.0f, theta_shift = 0.0f;
    int theta_shift_seq = 0;
    bool cells_suspended = false;  // the list / pool overflowed twice in a row: round-2 plan until the next upload_tree
    int cells_full_frame = -1;
    bool balance_external = false; // multi-GPU: ONE rank of the node re-balances, the others adopt its arrays (pfslam_set_shard_balance)
    int cells_wipe_seq = 0;       // header flags of frames with an older ticket predate the last wipe
    bool score_on_aux = false;    // pfslam_step asks for it; launch_score grants it (scored_on_aux) in a frame whose cell passes are asynchronous
    bool scored_on_aux = false;
    hipEvent_t ev_scored = nullptr; // scores + min / max keys + pose of the frame are there (aux -> main)
    bool cells_async = false;     // frame loops: new cells are found and walked on the aux stream, under the scan-match kernel
    long cells_wipes = 0;         // statistics
    long cells_passes = 0;        // publishing updates since the last wipe
    int cells_gen = 0;            // wipe generation: a record written in an earlier one counts as not written
    hipEvent_t ev_boxes = nullptr; // the pose boxes of the pass are ready (main stream -> aux stream)
    hipEvent_t ev_marked = nullptr; // ... and the cell stream's marking pass has read them (-> main, before the next pass's boxes)
    bool mark_on_aux = false;
    hipStream_t istream = nullptr;  // the ICP solve: behind the previous frame's insert (ev_tree), beside its k_cells_update, in front of the scan-match kernel
    hipEvent_t ev_tree = nullptr;
    void *pin_tree = nullptr;      // pinned staging of the map's device arrays (upload_tree)
    bool tree_event = false;
    hipStream_t cstream = nullptr;  // marking + walks of the new cells in the frame loops: beside the ICP solve, under the scan-match kernel
    hipEvent_t ev_walked = nullptr; // ... finished: the frame's insert (k_test_new changes the tree they read) waits for it
    bool walk_pending = false;
    bool update_unsnapped = false; // the last post-insert k_cells_update read the raw list counter: the next marking pass starts behind it
    bool cells_snap = false;       // this frame's k_cells_update<true> may run beside the next frame's marking pass: records below the walk pass's snapshot only
    pf::BeamParts *beam_angle = nullptr; // LIDAR_ANGLE(j) and its cos / sin as doubles, nb entries
    float *fit_acc = nullptr;    // per-lane score accumulators of the cell-row kernel (zero between passes)
    // ---- round-5 frame loop (pfslam_frame.hip.inc): four in-order chains, a fixed set of events between them ----
    int trig = 0;                 // pfslam_set_trig: 1 = the device library's cosf / sinf / erfcinvf instead of the pf_math.h specification (see sincos_sum_spec)
    int serial = 0;               // PFSLAM_SERIAL=1: every frame's launches on ONE stream, in enqueue order (same results, same bookkeeping)
    int frame_v2 = 1;             // PFSLAM_FRAME_V2=0: the round-4 frame (A/B runs)
    bool pipe_live = false;       // the last frame was a round-5 frame: its events and ring slots are what the next one waits on
    bool cloud_valid = false;     // the cloud statistics k_motion_count starts from describe the current particles
    int publish_lag = 2;          // a publishing pass takes the records walked `publish_lag` frames ago (ordered through ev_join)
    int *fs = nullptr;            // frame state words (PF_FS_*)
    int *wcounts = nullptr;       // k_walls: wall cells, new walls, header flags, map size
    float2 *wall_xy2 = nullptr;   // k_walls<1> -> k_walls_traverse -> k_walls<3>: the frame's wall points, the link each falls off, new-wall flags
    int *wall_leaf2 = nullptr, *wall_new2 = nullptr;
    int *wall_c2 = nullptr;       // nearest index of every wall of the frame (k_walls -> k_wall_weights)
    uint32_t *wall_runs = nullptr, *wall_keys_s = nullptr; // k_wall_runs -> k_walls_rank_traverse -> k_walls<3>: the beams' wall cells sorted inside runs of 64; in
    double2 *tparts = nullptr, *tparts_cur = nullptr; // cos / sin of every particle's heading as doubles (k_motion_count -> the frame's scan-match kernel)
    int *wall_c2s = nullptr;      // rank order with the duplicates flagged (the other per-wall arrays of that chain are indexed by rank too)
    long long *fstats = nullptr;  // packed min / max keys, [2][4] by ticket parity (a frame's last reduce workgroup resets the other one)
    float *cloud = nullptr;       // cloud statistics {mean x, y, heading, spread}, [2][4] by ticket parity
    float *sigr = nullptr;        // [PF_FRAME_RING][80]: {spread, -, 64 partial maxima of |heading|} of a frame's cloud -> header part A
    int *order_ring = nullptr;    // [PF_FRAME_RING][n]: lane order by ticket (the cells' stream reads a frame's order while the next is made)
    short *pgroup = nullptr;      // group-major 16-bit beam-chunk partials
    size_t pgroup_elems = 0;
    hipStream_t fstream = nullptr; // F: the free cells' chain (the round-4 ICP stream's place: a FIFTH stream halves the frame rate on this runtime)
    hipEvent_t ev_reduced = nullptr, ev_tree2 = nullptr, ev_order = nullptr, ev_icp = nullptr, ev_ftail = nullptr, ev_shift = nullptr;
    hipEvent_t ev_marked_r[4] = {nullptr, nullptr, nullptr, nullptr}, ev_walked_r[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_tail[3] = {nullptr, nullptr, nullptr}; // settle(): the tails of C, F, K -> P
    int fault = 0;                // PFSLAM_FAULT (tests): see CellPass
    int stable_order = 0;         // PFSLAM_STABLE_ORDER (tests): a canonical lane order (k_order_stable)
    int *order_tmp = nullptr;
    int mark_early = 1;           // the cells' passes start with the lane order (0: behind the frame's reduce)
    int gates = 1;                // cross-stream edges of a round-5 frame through device words + one-wave gate kernels instead of events (k_gate)
    bool ftail_recorded = false;  // ev_ftail holds the tail of the last frame's free-cell chain (not with the FTAIL gate: recorded by whoever needs it)
    int *flags = nullptr;         // PF_FL_*: the ticket of the last frame whose reduce / insert / lane order / ICP solve is done
    int *gate_err = nullptr, *gate_err_dev = nullptr; // pinned: ticket of a frame one of whose gates gave up (reported by the call that books it)
    const float *x_frame = nullptr, *y_frame = nullptr, *th_frame = nullptr;
    bool shift_pending = false;   // an odometry shift was enqueued behind the last frame: the next ICP solve waits for it
    bool balance_done = false;    // pfslam_step has already re-balanced for the frame frame_front is about to enqueue
    bool walls_attr_set = false;
    // frame probe (pfslam_set_probe): wall-clock stamps by the first thread of a frame's launches
    unsigned long long *probe = nullptr;
    int probe_frames = 0;
};
#define PF_PROBE_SLOTS 32
enum { PB_ICP = 0, PB_MOTION, PB_SCATTER, PB_BOX, PB_MARK, PB_WALK, PB_SCORE, PB_REDUCE, PB_WALLS, PB_UPDATE, PB_WEIGHTS, PB_APPLY, PB_GATHER, PB_RAYS, PB_COUNT,
       PB_LISTS, PB_FREE, PB_WALLW, PB_REDUCE_LAST, PB_WALLS_KEYS, PB_WALLS_SORTED, PB_WALLS_TRAV, PB_WALLS_END, PB_WALLS_REP, PB_ICP_SOLVE, PB_ICP_END, PB_UPDATE_END, PB_REDUCE_END, PB_END };
static const char *const pb_names[PB_END] = {"K icp", "P motion+cells", "P scatter", "K boxes", "K mark", "K walk", "C scan-match", "C reduce", "C walls+insert",
                                             "C cells update", "P weights", "P scan apply", "P sample+gather", "F rays", "F count", "F lists", "F free pass", "F wall weights",
                                             "C reduce: last wg", "C walls: keys", "C walls: sorted", "C walls: traversed", "C walls: end", "C walls: traversal launch", "K icp: solve", "K icp: end", "C cells update: last wg ends", "C reduce: last wg ends"};
static_assert(PB_END <= PF_PROBE_SLOTS, "probe slots");

// ==========================================================================================
// kernels
// ==========================================================================================

// ---- A3: dispersion (ParticleAddNoise / kernAddNoise, kernel.cu:375-397) ------------------
// One thread per particle, coalesced SoA.  `wm` is the reference's host-side particle array:
// PFMotionUpdate starts with an H2D of it (kernel.cu:408), which -- because the measurement
// update only reads back the first half of the array (kernel.cu:1341, H11) -- resets the
// weights of the second half to their pre-measurement values.
__global__ __launch_bounds__(256) void k_motion(float *__restrict__ x, float *__restrict__ y,
                                                float *__restrict__ th, float *__restrict__ w,
                                                const float *__restrict__ wm, int n, int frame, int goff, int trig = 0)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    w[i] = wm[i];
    uint32_t e2 = pf::engine_seed(frame, goff + i, 0);
    const float sx = 0.015, sy = 0.015, st = .01; // COV, kernel.cu:45 (used as std-dev, H10)
    float nx = pf::normal(e2, 0.0f, sx, trig != 0);
    float ny = pf::normal(e2, 0.0f, sy, trig != 0);
    float nt = pf::normal(e2, 0.0f, st, trig != 0);
    x[i] += nx;
    y[i] += ny;
    th[i] += nt;
}

// LIDAR_ANGLE(j) of every beam (kernel.cu:42) and its cos / sin as doubles, once per handle: the score kernels read them as
// wave-uniform scalars (sincos_sum_spec, pf_math.h)
__global__ void k_beam_angles(pf::BeamParts *__restrict__ beams, int nb)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nb) return;
    const float angle = pf::lidar_angle(j);
    const pf::AngleParts p = pf::angle_parts(angle);
    beams[j] = pf::BeamParts{p.c, p.s, p.a, angle, 0.0f};
}
template <int GUARD = 1>
__device__ __forceinline__ void beam_end_point(const pf::BeamParts *__restrict__ beams, int j, float range, float theta, const pf::AngleParts &T,
                                               float &x, float &y)
{
    const pf::BeamParts bp = beams[j]; // wave-uniform: scalar loads
    pf::clean_lidar_scan_parts<GUARD>(bp.angle, pf::AngleParts{bp.c, bp.s, bp.a}, range, theta, T, x, y);
}
// ... with the trigonometry chosen at run time (pfslam_set_trig; every kernel but the cell-row scan-match kernel, which is instantiated for it)
__device__ __forceinline__ void beam_end_point_rt(const pf::BeamParts *__restrict__ beams, int j, float range, float theta, const pf::AngleParts &T,
                                                  float &x, float &y, int trig)
{
    if (trig) beam_end_point<PF_TRIG_DEVLIB>(beams, j, range, theta, T, x, y);
    else beam_end_point(beams, j, range, theta, T, x, y);
}

// ---- A5: scan-match score (EvaluateParticleKD / kernEvaluateParticlesKD, kernel.cu:1198-1308)
// Lane = particle, the wave walks a chunk of beams: all 64 lanes query the same beam from
// near-identical poses, so their descents touch the same nodes until the last levels (one
// cache line per step instead of 64) and the per-lane sum keeps the reference's beam order.
// blockIdx.y selects the beam chunk; partial sums are combined by k_reduce_partials.
// CENSUS: the same kernel counting its own loop trips (pfslam_score_census); never the timed instantiation.
template <bool PLANAR, bool CENSUS = false>
__global__ __launch_bounds__(256) void k_score_kd(const float *__restrict__ px, const float *__restrict__ py,
                                                  const float *__restrict__ pth, int n,
                                                  const float *__restrict__ scan, const pf::BeamParts *__restrict__ beams, int nb,
                                                  int beams_per_chunk, pf::KdView tree, const int *__restrict__ order, int direct,
                                                  float *__restrict__ out, pf::KdCensus *__restrict__ census = nullptr, int trig = 0)
{
    const int slot = blockIdx.x * blockDim.x + threadIdx.x;
    const int j0 = blockIdx.y * beams_per_chunk;
    const int j1 = min(nb, j0 + beams_per_chunk);
    if (slot >= n) return;
    // lane -> particle through the Hilbert order: the 64 lanes of a wave hold neighbouring poses
    const int i = order ? order[slot] : slot;
    const float x = px[i], y = py[i], th = pth[i];
    const pf::AngleParts T = pf::angle_parts(th); // the heading's part of every end point of this lane
    float acc = 0.0f;
    pf::KdCensusLocal cl = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = j0; j < j1; j++) {
        float wx, wy;
        beam_end_point_rt(beams, j, scan[j], th, T, wx, wy, trig);
        if (fabsf(wx) < PF_LIDAR_RANGE && fabsf(wy) < PF_LIDAR_RANGE) {
            wx += x;
            wy += y;
            const int b = pf::kd_nearest_ref<PLANAR, CENSUS>(tree, wx, wy, 0.0f, &cl);
            acc += tree.w[b];
        }
    }
    if (CENSUS) pf::census_flush(cl, census);
    out[(size_t)blockIdx.y * n + (direct ? i : slot)] = acc;
}

#include "kd_cells.hip.inc" // lattice-cell rows (k_cells_mark, k_cells_update, k_score_kd_cells); beam_box is shared with k_plan

// ---- shared-prefix plan (kd_device.h "Shared-prefix plan"): round 2; still used for planar maps that are not on the lattice -----
// pose bounding box of every group of 64 lanes (= one wave of the score kernel)
__global__ __launch_bounds__(64) void k_group_box(const float *__restrict__ px, const float *__restrict__ py, const float *__restrict__ pth,
                                                  int n, const int *__restrict__ order, pf::KdGroupBox *__restrict__ box,
                                                  pf::AngleParts *__restrict__ parts, float *__restrict__ sigma)
{
    const int slot = blockIdx.x * 64 + threadIdx.x;
    const bool in = slot < n;
    const int i = in ? (order ? order[slot] : slot) : 0;
    const float x = in ? px[i] : 0.0f, y = in ? py[i] : 0.0f, t = in ? pth[i] : 0.0f;
    float xlo = in ? x : INFINITY, xhi = in ? x : -INFINITY, ylo = in ? y : INFINITY, yhi = in ? y : -INFINITY;
    float tlo = in ? t : INFINITY, thi = in ? t : -INFINITY;
    for (int off = 32; off >= 1; off >>= 1) {
        xlo = fminf(xlo, __shfl_xor(xlo, off, 64)); xhi = fmaxf(xhi, __shfl_xor(xhi, off, 64));
        ylo = fminf(ylo, __shfl_xor(ylo, off, 64)); yhi = fmaxf(yhi, __shfl_xor(yhi, off, 64));
        tlo = fminf(tlo, __shfl_xor(tlo, off, 64)); thi = fmaxf(thi, __shfl_xor(thi, off, 64));
    }
    if (threadIdx.x == 0) {
        // a NaN pose poisons min / max silently (fminf ignores NaN): such a group gets no plan
        const unsigned long long bad = __builtin_amdgcn_ballot_w64(in && !(x == x && y == y && t == t));
        pf::KdGroupBox b{xlo, xhi, ylo, yhi, tlo, thi, min(64, n - (int)blockIdx.x * 64), 0};
        if (bad != 0ull) b.xlo = b.xhi = NAN;
        box[blockIdx.x] = b;
        // largest |heading| of the cloud (non-negative floats order like their bit patterns; a NaN pose counts as infinity)
        // (spread over 64 words: 1563 atomics on ONE word took this kernel from 6 to 21 us; the header's writer takes their maximum)
        const float tmax = bad != 0ull ? INFINITY : fmaxf(fabsf(tlo), fabsf(thi));
        atomicMax((int *)&sigma[2 + (blockIdx.x & 63)], __float_as_int(tmax));
        if (parts) parts[blockIdx.x] = pf::angle_parts(0.5f * (b.tlo + b.thi)); // of beam_box's centre heading: once per group here

    }
}

// One lane per (group, beam): the common root path of the wave's 64 queries, pruned to the possible nearest nodes.
// Every bound is conservative: W contains every lane's float-computed beam end point with >= 1e-4 m to spare.
// The 64 lanes of a planning wave are 64 CONSECUTIVE GROUPS looking along the SAME beam: Hilbert neighbours, so their boxes
// nearly coincide and the lanes walk the same nodes (coherent gathers, no loop divergence) -- with lanes = beams of one group
// the same kernel took 0.45 ms instead of ~0.03.  The running candidate list lives in LDS (node index + lower bound).
__global__ __launch_bounds__(64) void k_plan(const pf::KdGroupBox *__restrict__ box, int groups, const float *__restrict__ scan, int nb,
                                             pf::KdView tree, pf::KdPlanRow *__restrict__ plan)
{
    __shared__ int s_idx[PF_PLAN_CAND][64];
    __shared__ float s_lb[PF_PLAN_CAND][64];
    const int lane = threadIdx.x, g = blockIdx.x * 64 + lane, j = blockIdx.y;
    if (g >= groups) return;
    pf::KdPlanRow *row = plan + ((size_t)g * nb + j);
    const pf::KdGroupBox b = box[g];
    const float r = scan[j];
    float wxlo, wxhi, wylo, wyhi;
    // no plan (every lane walks from the root): non-finite or absurd geometry, NaN poses, and beams no lane can accept
    // (|r| >= PF_RANGE_NEVER puts |r cos| or |r sin| beyond the 20 m reject of kernel.cu:1213 for every heading: the score kernel
    // skips such beams outright)
    const bool usable = beam_box(b, j, r, wxlo, wxhi, wylo, wyhi);
    int n_cand = 0, resume = 0, path_len = 0;
    float U = INFINITY; // upper bound (with margin) of the final minimum for every point of W
    if (usable) {
        int head = 0;
        while (head >= 0) {
            const uint4 nd = tree.hot[head];
            const float nx = __uint_as_float(nd.x), ny = __uint_as_float(nd.y);
            const uint32_t axis = nd.z >> 30;
            // does W lie entirely on one side of the split plane?  lanes go left iff q < node (strictly)
            const float lo = axis == 0 ? wxlo : wylo, hi = axis == 0 ? wxhi : wyhi, v = axis == 0 ? nx : ny;
            const bool all_left = hi < v, all_right = lo >= v;
            if (axis < 2 && !(all_left || all_right)) break; // W straddles: lanes part here -> resume at this node
            // squared distance range of W to the node
            const float dxn = fmaxf(fmaxf(wxlo - nx, nx - wxhi), 0.0f), dyn = fmaxf(fmaxf(wylo - ny, ny - wyhi), 0.0f);
            const float dxf = fmaxf(fabsf(nx - wxlo), fabsf(nx - wxhi)), dyf = fmaxf(fabsf(ny - wylo), fabsf(ny - wyhi));
            const float lb = (dxn * dxn + dyn * dyn) * 0.99999f, ub = (dxf * dxf + dyf * dyf) * 1.00001f;
            U = fminf(U, ub);
            if (lb <= U) { // may be the nearest for some lane
                if (n_cand == PF_PLAN_CAND) { // full: drop what the tighter U has ruled out meanwhile
                    int m = 0;
                    for (int k = 0; k < PF_PLAN_CAND; k++) {
                        const int ci = s_idx[k][lane];
                        const float cl = s_lb[k][lane];
                        if (cl <= U) {
                            s_idx[m][lane] = ci;
                            s_lb[m][lane] = cl;
                            m++;
                        }
                    }
                    n_cand = m;
                    if (n_cand == PF_PLAN_CAND) break; // still full: the lanes take over at this node
                }
                s_idx[n_cand][lane] = head;
                s_lb[n_cand][lane] = lb;
                n_cand++;
            }
            path_len++;
            head = (axis < 2 && all_left) ? pf::hot_left(nd.z) : (int)nd.w; // planar z levels: both links hold the right child
        }
        resume = head;
    }
    int m = 0; // final pruning with the final U; the survivors' coordinates come from the map again (a few per row)
    for (int k = 0; k < n_cand; k++) {
        const int ci = s_idx[k][lane];
        if (s_lb[k][lane] <= U) {
            const uint4 nd = tree.hot[ci];
            row->cand[m++] = make_float4(__uint_as_float(nd.x), __uint_as_float(nd.y), __int_as_float(ci), 0.0f);
        }
    }
    row->n_cand = m;
    row->resume = resume;
    row->path_len = path_len;
    row->range = r;
}

// The score kernel on a plan: per beam, every lane evaluates the wave's few candidate nodes out of scalar registers (the row
// is wave-uniform), with the per-visit arithmetic of kd_resume, then continues per lane from the row's resume node.
// The row of the NEXT beam (and its range) is requested before the current beam is worked on: rows are streamed once from
// HBM / L2, and a wave that waited for its row on demand spent more time waiting than computing.
__device__ __forceinline__ void plan_visit(const float4 cd, float wx, float wy, float &sBest, int &bestIdx)
{
    const float dx = cd.x - wx, dy = cd.y - wy;
    const float s = dx * dx + dy * dy;
    const float sGuard = sBest * PF_GUARD_K;
    bool take = s < sGuard;
    const bool inBand = (s < sBest) != take;
    if (__builtin_amdgcn_ballot_w64(inBand) != 0ull) {
        float sb = sBest;
        asm volatile("" : "+v"(sb));
        take = take | (inBand && pf::fsqrt(s) < pf::fsqrt(sb));
    }
    sBest = take ? s : sBest;
    bestIdx = take ? __float_as_int(cd.z) : bestIdx;
}

template <bool CENSUS = false>
__global__ __launch_bounds__(64) void k_score_kd_plan(const float *__restrict__ px, const float *__restrict__ py,
                                                      const float *__restrict__ pth, int n, const float *__restrict__ scan,
                                                      const pf::BeamParts *__restrict__ beams, int nb,
                                                      int beams_per_chunk, pf::KdView tree, const pf::KdPlanRow *__restrict__ plan,
                                                      const int *__restrict__ order, int direct, float *__restrict__ out,
                                                      pf::KdCensus *__restrict__ census = nullptr, int trig = 0)
{
    const int g = blockIdx.x, slot = g * 64 + threadIdx.x;
    const int j0 = blockIdx.y * beams_per_chunk;
    const int j1 = min(nb, j0 + beams_per_chunk);
    if (slot >= n) return;
    const int i = order ? order[slot] : slot;
    const float x = px[i], y = py[i], th = pth[i];
    const pf::AngleParts T = pf::angle_parts(th);
    float acc = 0.0f;
    pf::KdCensusLocal cl = {0, 0, 0, 0, 0, 0, 0, 0};
    const pf::KdPlanRow *rows = plan + (size_t)g * nb; // wave-uniform addresses: scalar loads
    pf::KdPlanRow cur = rows[j0];
    for (int j = j0; j < j1; j++) {
        const pf::KdPlanRow nxt = rows[min(j + 1, j1 - 1)]; // in flight while this beam is scored
        float wx, wy;
        // wave-uniform: a beam of 28.4 m or more fails the +-20 m test below for every heading (20 * sqrt(2) = 28.28) -- and so
        // does a NaN range; skipping it here saves the lanes the end-point arithmetic (synthetic scans clip at 30 m)
        if (!(fabsf(cur.range) < PF_RANGE_NEVER)) {
            cur = nxt;
            continue;
        }
        beam_end_point_rt(beams, j, cur.range, th, T, wx, wy, trig);
        if (fabsf(wx) < PF_LIDAR_RANGE && fabsf(wy) < PF_LIDAR_RANGE) {
            wx += x;
            wy += y;
            float sBest = INFINITY;
            int bestIdx = 0;
            const int nc = cur.n_cand;
#pragma unroll
            for (int c = 0; c < PF_PLAN_CAND; c++)
                if (c < nc) plan_visit(cur.cand[c], wx, wy, sBest, bestIdx); // wave-uniform branch
            if (CENSUS) cl.prefix += (unsigned)nc; // candidate evaluations (per lane)
            const int b = pf::kd_resume<true, CENSUS>(tree, wx, wy, 0.0f, sBest, bestIdx, cur.resume, &cl);
            acc += tree.w[b];
        }
        cur = nxt;
    }
    if (CENSUS) pf::census_flush(cl, census);
    out[(size_t)blockIdx.y * n + (direct ? i : slot)] = acc;
}

// beam-chunk partials -> fit, undoing the processing order.  Map weights are integers by construction
// (0 / -100 initial, -1 / +4 steps, clamp +-113), so the chunked sum equals the reference's sequential one.
__global__ __launch_bounds__(256) void k_reduce_partials(const float *__restrict__ partial, int n, int chunks,
                                                         const int *__restrict__ order, float *__restrict__ fit)
{
    int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= n) return;
    // sequential chunk order (deterministic for any weights; with one beam per chunk -- small N -- it is the reference's own
    // beam order); loads are independent, so keep 64, then 16, of them in flight
    float a = 0.0f;
    int c = 0;
    for (; c + 64 <= chunks; c += 64) {
        float v[64];
#pragma unroll
        for (int k = 0; k < 64; k++) v[k] = partial[(size_t)(c + k) * n + slot];
#pragma unroll
        for (int k = 0; k < 64; k++) a += v[k];
    }
    for (; c + 16 <= chunks; c += 16) {
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = partial[(size_t)(c + k) * n + slot];
#pragma unroll
        for (int k = 0; k < 16; k++) a += v[k];
    }
    for (; c < chunks; c++) a += partial[(size_t)c * n + slot];
    fit[order ? order[slot] : slot] = a;
}

// ------------------------------------------------------------------------------------------
// Lane order in 3 launches: a counting sort over Hilbert cells laid over the particle cloud itself -- 2^18 cells (64 per
// dimension) up to 400 k particles, 2^21 (128 per dimension) above -- across +-3.2 sigma around the mean, both estimated from
// the first 1024 slots (slots are exchangeable).  There are more cells than particles, so almost every particle has a cell of
// its own and the order inside a cell (atomic arrival order) does not matter.
// The order only decides which lane scores which particle; results do not depend on it.
//   k_cell_count:   cell of every particle (kept in `cell[]`), histogram with global atomics
//   k_cell_scan:    exclusive scan inside 1024-cell tiles + tile totals; the counts are zeroed for the next frame
//   k_cell_scatter: slot = tile offset + atomicAdd(cursor[cell]) -> order[slot] = particle
// ------------------------------------------------------------------------------------------
#define PF_CELL_BITS_MAX 7
#define PF_CELLS_MAX (1 << (3 * PF_CELL_BITS_MAX))
__device__ __forceinline__ unsigned spread3(unsigned v, int bits) // bit b of v -> bit 3 b
{
    unsigned r = 0;
    for (int b = 0; b < bits; b++) r |= ((v >> b) & 1u) << (3 * b);
    return r;
}
__device__ __forceinline__ float block_sum_256(float v, float *red)
{
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
// reach: metres per radian -- a heading difference d moves a beam end point by ~ reach * d, so one cell is equally wide in x, y
// and reach * theta (what makes the 64 queries of a wave a small box, see the shared-prefix plan)
// cs != nullptr (lattice-cell rows, kd_cells.hip.inc): block 0 also puts the window of the cell table around the cloud's mean when the
// table has just been wiped, and reports a cloud that has drifted away from the window's middle
__global__ __launch_bounds__(256) void k_cell_count(const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ th, int n, float reach, int bits,
                                                    unsigned *__restrict__ cell, int *__restrict__ hist, int *__restrict__ cs, CellGeom geo, float *__restrict__ sigma_out)
{
    __shared__ float red[4];
    const int ns = min(n, 1024); // cloud statistics, identical in every block
    float sx = 0, sy = 0, st = 0;
    for (int k = threadIdx.x; k < ns; k += 256) { sx += x[k]; sy += y[k]; st += th[k]; }
    const float inv = 1.0f / (float)ns;
    const float mx = block_sum_256(sx, red) * inv, my = block_sum_256(sy, red) * inv, mt = block_sum_256(st, red) * inv;
    if (cs && blockIdx.x == 0 && threadIdx.x == 0) {
        const bool fin = fabsf(mx) < 1e6f && fabsf(my) < 1e6f; // NaN poses among the first slots: any window will do
        const int kx = fin ? lattice_floor(mx, geo.resx, geo.invx) : 0, ky = fin ? lattice_floor(my, geo.resy, geo.invy) : 0;
        if (cs[PF_CS_FLAGS] & PF_CF_NEED_ORIGIN) { // first pass after a wipe: in sub-cell units, on a whole cell (even)
            cs[PF_CS_OX] = PF_CELL_SUB * kx - PF_CELL_WIN / 2;
            cs[PF_CS_OY] = PF_CELL_SUB * ky - PF_CELL_WIN / 2;
            atomicAnd(&cs[PF_CS_FLAGS], ~PF_CF_NEED_ORIGIN);
        } else if (fin) { // the window stays where it is while the rows persist; the host is told when the cloud leaves its middle
            const int dx = kx - (cs[PF_CS_OX] + PF_CELL_WIN / 2) / PF_CELL_SUB, dy = ky - (cs[PF_CS_OY] + PF_CELL_WIN / 2) / PF_CELL_SUB;
            if (abs(dx) > PF_CELL_DRIFT_MAX || abs(dy) > PF_CELL_DRIFT_MAX) atomicOr(&cs[PF_CS_FLAGS], PF_CF_FAR);
        }
    }
    float vx = 0, vy = 0, vt = 0;
    for (int k = threadIdx.x; k < ns; k += 256) {
        const float a = x[k] - mx, b = y[k] - my, c = th[k] - mt;
        vx += a * a; vy += b * b; vt += c * c;
    }
    // D = 2^bits cells over +-3.2 sigma of the widest dimension in the common metric (x, y, reach * theta): cell = 6.4 sigma / D
    // of it, never finer than 0.25 mm
    const float dev_x = sqrtf(block_sum_256(vx, red) * inv), dev_y = sqrtf(block_sum_256(vy, red) * inv);
    const float dev_t = sqrtf(block_sum_256(vt, red) * inv) * reach;
    const float D = (float)(1 << bits), half = 0.5f * D, top = D - 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sigma_out[0] = fmaxf(fmaxf(dev_x, dev_y), dev_t); // -> frame header -> host (organisation of the next pass)
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) sigma_out[2 + threadIdx.x] = 0.0f; // max |heading|, 64 partial maxima: k_group_box, behind this kernel
    const float e = fmaxf(6.4f / D * fmaxf(fmaxf(dev_x, dev_y), dev_t), 2.5e-4f);
    const float cx = 1.0f / e, cy = 1.0f / e, ct = reach / e;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned X[3] = {(unsigned)fminf(fmaxf((th[i] - mt) * ct + half, 0.0f), top),
                     (unsigned)fminf(fmaxf((x[i] - mx) * cx + half, 0.0f), top),
                     (unsigned)fminf(fmaxf((y[i] - my) * cy + half, 0.0f), top)};
    const unsigned M = 1u << (bits - 1); // Hilbert index, Skilling's axes-to-transpose
    for (unsigned Q = M; Q > 1; Q >>= 1) {
        const unsigned P = Q - 1;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if (X[a] & Q) X[0] ^= P;
            else { const unsigned t = (X[0] ^ X[a]) & P; X[0] ^= t; X[a] ^= t; }
        }
    }
    X[1] ^= X[0]; X[2] ^= X[1];
    unsigned t = 0;
    for (unsigned Q = M; Q > 1; Q >>= 1) if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t; X[1] ^= t; X[2] ^= t;
    const unsigned c = (spread3(X[0], bits) << 2) | (spread3(X[1], bits) << 1) | spread3(X[2], bits);
    cell[i] = c;
    atomicAdd(&hist[c], 1);
}
// one block per 1024-cell tile: exclusive scan inside the tile + the tile total; the counts are zeroed for the next frame
__global__ __launch_bounds__(256) void k_cell_scan(int *__restrict__ hist, int *__restrict__ cursor, int *__restrict__ tile_tot)
{
    __shared__ int wtot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 1024 + threadIdx.x * 4;
    const int4 v = *(const int4 *)(hist + c0);
    *(int4 *)(hist + c0) = make_int4(0, 0, 0, 0);
    const int s = v.x + v.y + v.z + v.w;
    int inc = s;
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(inc, off, 64);
        if (lane >= off) inc += u;
    }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int base = inc - s;
    for (int k = 0; k < wave; k++) base += wtot[k];
    *(int4 *)(cursor + c0) = make_int4(base, base + v.x, base + v.x + v.y, base + v.x + v.y + v.z);
    if (threadIdx.x == 255) tile_tot[blockIdx.x] = base + s;
}
__global__ __launch_bounds__(256) void k_cell_scatter(const unsigned *__restrict__ cell, int n, int *__restrict__ cursor,
                                                      const int *__restrict__ tile_tot, int ntiles, int *__restrict__ order)
{
    // offsets of the tiles: every block scans the tile totals itself (256 or 2048 values, 256 at a time with a carry)
    __shared__ int tile_off[PF_CELLS_MAX / 1024];
    __shared__ int wtot[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int carry = 0;
    for (int t0 = 0; t0 < ntiles; t0 += 256) {
        const int t = tile_tot[t0 + threadIdx.x];
        int inc = t;
        for (int off = 1; off < 64; off <<= 1) {
            const int u = __shfl_up(inc, off, 64);
            if (lane >= off) inc += u;
        }
        __syncthreads(); // wtot of the previous chunk has been read by everyone
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int base = carry + inc - t;
        for (int k = 0; k < wave; k++) base += wtot[k];
        tile_off[t0 + threadIdx.x] = base;
        carry += wtot[0] + wtot[1] + wtot[2] + wtot[3];
    }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const unsigned c = cell[i];
        order[tile_off[c >> 10] + atomicAdd(&cursor[c], 1)] = i;
    }
}

// ---- findCorrespondenceIndexKD (kernel.cu:924-972) over an arbitrary xyz batch --------------
template <bool PLANAR>
__global__ __launch_bounds__(256) void k_traverse(const float *__restrict__ xyz, int n, pf::KdView tree,
                                                  int *__restrict__ best)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    best[i] = pf::kd_nearest_ref<PLANAR>(tree, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
}

// ---- debug: evaluate the pf_math specification on the device (parity tests) -----------------
__global__ void k_debug_math(int which, const float *__restrict__ in, int n, float *__restrict__ out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (which == 0) {
        float s, c;
        pf::sincosf_spec(in[i], s, c);
        out[2 * i] = s;
        out[2 * i + 1] = c;
    } else if (which == 1) {
        out[i] = pf::erfcinvf_spec(in[i]);
    } else if (which == 2) {
        out[i] = pf::asinf_spec(in[i]);
    } else if (which == 3) {
        out[i] = pf::rsqrtf_spec(in[i]);
    } else if (which == 4) {
        out[i] = pf::fsqrt(in[i]);
    } else if (which == 5) {
        out[i] = pf::fdiv(in[i], 0.025f);
    } else if (which == 6) { // the score kernel's sub-cell index: exact, and the shortcut (INT_MIN where it does not apply)
        const float res = 0.025f, inv = pf::fdiv(1.0f, res);
        float clear;
        const int fast = sub_index_fast(in[i], 2.0f * inv, clear);
        out[2 * i] = __int_as_float(sub_index(in[i], res, inv));
        out[2 * i + 1] = __int_as_float(clear > 0.0f ? fast : (int)0x80000000);
    } else if (which == 7) { // CleanLidarScan as the hot loops compute it (per-beam table x heading parts): beam i % 1081, range 1, heading in[i]
        float x, y;
        pf::clean_lidar_scan(i % 1081, 1.0f, in[i], x, y);
        out[2 * i] = x;
        out[2 * i + 1] = y;
    }
}

// ==========================================================================================
// C-ABI
// ==========================================================================================
extern "C" const char *pfslam_last_error(void) { return g_err.c_str(); }

extern "C" int pfslam_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return -(int)e;
    return n;
}

extern "C" void pfslam_default_config(pfslam_config *cfg)
{
    memset(cfg, 0, sizeof(*cfg));
    cfg->n_particles = 1000; // PARTICLE_COUNT, kernel.cu:30
    cfg->n_beams = 1081;
    cfg->map_scale_x = cfg->map_scale_y = 40.0f;
    cfg->map_res_x = cfg->map_res_y = 0.025f;
    cfg->kd_capacity = 10000000; // KD_MAX_SIZE, kernel.cu:77
    cfg->device = 0;
    cfg->strict_host_mirror = 1;
    cfg->free_upload_bug = 0;
    cfg->balance_period = 100;
}

// Every event of a handle orders work of ONE device (streams of the handle) or times it; none of them hands memory to the host (frame
// headers go through explicit system-scope stores, host copies through stream synchronisation).  The default hipEventRecord ends in a
// SYSTEM-scope release -- an L2 write-back and invalidate that costs the recording stream ~10 us and leaves the kernels behind it cold
// caches (14 us between two 6 us kernels of the frame's critical chain): a device-scope release is what these events need.
static hipError_t pf_event_create(hipEvent_t *e, unsigned flags)
{
    static const bool dev_scope = !(ab_env("PFSLAM_EVENT_SYSTEM") && atoi(ab_env("PFSLAM_EVENT_SYSTEM")) != 0); // A/B: 1 = the default (system) release
    hipError_t rc = hipEventCreateWithFlags(e, flags | (dev_scope ? hipEventReleaseToDevice : 0u));
    if (rc != hipSuccess && dev_scope) { // (a runtime that does not know the flag)
        (void)hipGetLastError();
        rc = hipEventCreateWithFlags(e, flags);
    }
    return rc;
}
template <typename T>
static int dalloc(T **p, size_t count)
{
    HIPCHK(hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
    return 0;
}

static pf::KdView kd_view(const pfslam_handle *h) { return pf::KdView{h->hot, h->kz, h->parent, h->kw, h->planar}; }
static int settle(pfslam_handle *h);   // finish and book the frames in flight (pfslam_stages.hip.inc)
static int settle_staged(pfslam_handle *h); // settle + leave the round-5 frame loop (a staged call reads or writes what its frames keep on four streams)
static int join_all(pfslam_handle *h); // round-5 frames: the tails of the chain / free-cell / cell streams -> the handle's stream
static void frame_free(pfslam_handle *h);
static int join_map(pfslam_handle *h); // main stream waits for the map update a frame left on the aux stream

extern "C" int pfslam_destroy(pfslam_handle *h);

// Handles alive in this process.  The stream gates of the round-5 frame (pfslam_frame.hip.inc) spin inside kernels and are only safe
// when the streams involved sit on hardware queues of their own: that is self-tested for ONE handle's four streams, but the runtime
// pools a handful of hardware queues per priority, so the streams of two handles may share one -- and two handles with frames in flight
// could then wait on each other (h1's signal queued behind h2's gate and vice versa) until the gates give up.  So: gates only while a
// handle is the process's only one; with a second handle alive every handle's edges are events (PFSLAM_GATES=0 forces that too).
static std::atomic<int> g_live_handles{0};

// allocations and initial state of a handle; on failure the caller destroys the partially built handle
static int create_impl(pfslam_handle *h)
{
    const pfslam_config *cfg = &h->cfg;
    h->n = cfg->n_particles;
    h->nb = cfg->n_beams;
    h->gn = cfg->global_n > 0 ? cfg->global_n : cfg->n_particles;
    h->goff = cfg->global_offset;
    h->stride = cfg->shard_stride > 0 ? cfg->shard_stride : cfg->n_particles;
    h->world = (h->gn + h->stride - 1) / h->stride;
    h->rank = h->goff / h->stride;
    h->dimx = (int)(cfg->map_scale_x / cfg->map_res_x); // map_dim, kernel.cu:120
    h->dimy = (int)(cfg->map_scale_y / cfg->map_res_y);
    h->kd_cap = cfg->kd_capacity;
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
    HIPCHK(pf_event_create(&h->ev0, 0));
    HIPCHK(pf_event_create(&h->ev1, 0));
    {
        // the aux streams carry short dependent chains that run BESIDE the scan-match kernel's 131 k single-wave workgroups: at normal
        // priority their workgroups queue behind that flood (the one-workgroup ICP solve: 33 -> 284 us)
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        static const bool prio = !(ab_env("PFSLAM_AUX_PRIO") && atoi(ab_env("PFSLAM_AUX_PRIO")) == 0);
        HIPCHK(hipStreamCreateWithPriority(&h->aux, hipStreamNonBlocking, prio ? hi : lo));
        static const int kprio = ab_env("PFSLAM_K_PRIO") ? atoi(ab_env("PFSLAM_K_PRIO")) : 1; // 0 low, 1 normal (default), 2 high
        HIPCHK(hipStreamCreateWithPriority(&h->cstream, hipStreamNonBlocking, kprio == 2 ? hi : kprio == 1 ? (lo + hi) / 2 : lo)); // (beside the scan-match kernel: it must not get in its way)
        static const int fprio = ab_env("PFSLAM_F_PRIO") ? atoi(ab_env("PFSLAM_F_PRIO")) : 2; // A/B: the free cells' stream 0 low, 1 normal, 2 high
        HIPCHK(hipStreamCreateWithPriority(&h->istream, hipStreamNonBlocking, !prio ? lo : fprio == 2 ? hi : fprio == 1 ? (lo + hi) / 2 : lo));
        HIPCHK(pf_event_create(&h->ev_tree, hipEventDisableTiming));
        HIPCHK(pf_event_create(&h->ev_scored, hipEventDisableTiming));
        h->fstream = h->istream; // round-5 frames: the free cells' chain (their ICP solve rides on the cells' stream)
        if (const char *e = getenv("PFSLAM_SERIAL")) h->serial = atoi(e) != 0;
        if (const char *e = ab_env("PFSLAM_FRAME_V2")) h->frame_v2 = atoi(e) != 0;
        if (const char *e = getenv("PFSLAM_GATES")) h->gates = atoi(e) != 0;
        if (const char *e = getenv("PFSLAM_FAULT")) h->fault = atoi(e);
        if (const char *e = getenv("PFSLAM_STABLE_ORDER")) h->stable_order = atoi(e) != 0;
        if (const char *e = getenv("PFSLAM_MARK_EARLY")) h->mark_early = atoi(e) != 0;
        if (const char *e = getenv("PFSLAM_PUBLISH_LAG")) h->publish_lag = std::min(std::max(atoi(e), 2), 3);
    }
    HIPCHK(pf_event_create(&h->ev_fork, hipEventDisableTiming));
    HIPCHK(pf_event_create(&h->ev_join, hipEventDisableTiming));
    HIPCHK(pf_event_create(&h->ev_mapfork, hipEventDisableTiming));
    HIPCHK(pf_event_create(&h->ev_map, hipEventDisableTiming));
    const size_t n = h->n, M = (size_t)h->dimx * h->dimy, S = (size_t)h->stride;
    CHK(dalloc(&h->pblk, 3 * S)); CHK(dalloc(&h->pblk2, 3 * S)); CHK(dalloc(&h->w, S)); CHK(dalloc(&h->wm, S));
    HIPCHK(hipMemsetAsync(h->pblk, 0, 3 * S * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->pblk2, 0, 3 * S * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->w, 0, S * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->wm, 0, S * 4, h->stream));
    h->x = h->pblk; h->y = h->pblk + S; h->th = h->pblk + 2 * S;
    h->x2 = h->pblk2; h->y2 = h->pblk2 + S; h->th2 = h->pblk2 + 2 * S;
    CHK(dalloc(&h->gkeys, (size_t)2 * h->world)); CHK(dalloc(&h->mkeys, 2)); CHK(dalloc(&h->gtheta, 4));
    HIPCHK(hipMemsetAsync(h->gtheta, 0, 16, h->stream));
    if (h->gn != h->n) {
        // rank r owns [r * stride, min((r + 1) * stride, gn)): every rank but the last is full, the last one is not empty
        if (h->goff < 0 || h->goff % h->stride != 0 || h->n > h->stride || h->goff + h->n != std::min(h->goff + h->stride, h->gn))
            return fail("pfslam_create: shard [global_offset, +n_particles) does not fit the layout rank r = [r * shard_stride, min((r + 1) * shard_stride, global_n))");
        CHK(dalloc(&h->gw, (size_t)h->world * S)); CHK(dalloc(&h->gpose, (size_t)h->world * 3 * S));
        h->own_global = true;
        const unsigned long long ticks = PF_GATE_TICKS_SHARDED;
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_gate_ticks), &ticks, sizeof(ticks)));
    } else {
        h->gw = h->w;
        h->gpose = h->pblk;
    }
    CHK(dalloc(&h->scan_base, (size_t)h->nb * PF_HDR_SLOTS));
    CHK(dalloc(&h->beam_angle, (size_t)h->nb));
    hipLaunchKernelGGL(k_beam_angles, dim3((h->nb + 255) / 256), dim3(256), 0, h->stream, h->beam_angle, h->nb);
    HIPCHK(hipGetLastError());
    h->scan = h->scan_base;
    CHK(dalloc(&h->hot, (size_t)h->kd_cap)); CHK(dalloc(&h->parent, (size_t)h->kd_cap));
    CHK(dalloc(&h->kz, (size_t)h->kd_cap)); CHK(dalloc(&h->kw, (size_t)h->kd_cap));
    CHK(dalloc(&h->fit, n)); CHK(dalloc(&h->fit_i, n));
    CHK(dalloc(&h->mkey, n)); CHK(dalloc(&h->order2, n));
    CHK(dalloc(&h->cells, (size_t)2 * PF_CELLS_MAX + PF_CELLS_MAX / 1024));
    HIPCHK(hipMemsetAsync(h->cells, 0, ((size_t)2 * PF_CELLS_MAX + PF_CELLS_MAX / 1024) * sizeof(int), h->stream));
    CHK(dalloc(&h->stats, 8)); CHK(dalloc(&h->pose, 4)); CHK(dalloc(&h->start, 4));
    CHK(dalloc(&h->icp_tar, (size_t)h->nb * 4)); CHK(dalloc(&h->icp_cor, (size_t)h->nb * 4)); CHK(dalloc(&h->icp_dbg, 64)); // (round-5 frames: two of them, by ticket parity)
    CHK(dalloc(&h->free_mask, 2 * M)); // the two masks are contiguous: one memset per frame
    h->wall_mask = h->free_mask + M;
    h->max_wall = h->nb;
    h->max_free = (int)std::min<size_t>(M, (size_t)h->nb * (size_t)std::max(h->dimx, h->dimy));
    CHK(dalloc(&h->blk_cnt, 2 * ((M + 4095) / 4096) + 8));
    CHK(dalloc(&h->wall_cell, (size_t)h->max_wall)); CHK(dalloc(&h->free_cell, (size_t)h->max_free));
    CHK(dalloc(&h->wall_pts, (size_t)h->max_wall)); CHK(dalloc(&h->free_pts, (size_t)h->max_free));
    CHK(dalloc(&h->wall_c, (size_t)h->max_wall)); CHK(dalloc(&h->free_c, (size_t)h->max_free));
    CHK(dalloc(&h->counts, 8));
    const size_t G = (size_t)h->gn;
    const size_t nt_sum = (G + PF_SUM_TILE - 1) / PF_SUM_TILE, nt_scan = (G + PF_SCAN_TILE - 1) / PF_SCAN_TILE;
    CHK(dalloc(&h->tile_r, nt_sum)); CHK(dalloc(&h->tile_r2, nt_sum)); CHK(dalloc(&h->sums, 4));
    CHK(dalloc(&h->cdf, G)); CHK(dalloc(&h->chunk_max, (G + PF_SCAN_CHUNK - 1) / PF_SCAN_CHUNK));
    CHK(dalloc(&h->tile_tot, nt_scan)); CHK(dalloc(&h->tile_off, nt_scan)); CHK(dalloc(&h->tile_pmax, nt_scan));
    CHK(dalloc(&h->src, n));
    CHK(dalloc(&h->grid, M));
    CHK(dalloc(&h->d_count, 4));
    CHK(dalloc(&h->wall_leaf, (size_t)h->max_wall));
    CHK(dalloc(&h->kd_state, 4));
    CHK(dalloc(&h->d_sigma, 2 + 64));
    HIPCHK(hipMemsetAsync(h->d_sigma, 0, (2 + 64) * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->kd_state, 0, 16, h->stream));
    // headers and scan staging: pinned, coherent (the device writes a header while the stream keeps running, the host may poll it)
    HIPCHK(hipHostMalloc((void **)&h->h_hdr, PF_HDR_SLOTS * sizeof(HostHeader), hipHostMallocCoherent | hipHostMallocMapped));
    memset(h->h_hdr, 0, PF_HDR_SLOTS * sizeof(HostHeader));
    {
        void *dp = nullptr;
        HIPCHK(hipHostGetDevicePointer(&dp, h->h_hdr, 0));
        h->hdr_dev = (HostHeader *)dp;
    }
    HIPCHK(hipHostMalloc((void **)&h->h_scan, (size_t)PF_HDR_SLOTS * h->nb * 4));
    if (const char *e = ab_env("PFSLAM_LAG")) h->lag = std::min(std::max(atoi(e), 0), PF_MAX_LAG);
    if (const char *e = getenv("PFSLAM_VARIANT")) h->variant = atoi(e); // initial pfslam_set_variant (A/B runs, the fuzz)
    h->h_nodes.reserve(1024);
    // particleFilterInit (kernel.cu:122-132): grid = -100, particles at the origin with w = 1, robotPos = 0
    std::vector<float> ones(n, 1.0f);
    HIPCHK(hipMemsetAsync(h->x, 0, n * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->y, 0, n * 4, h->stream));
    HIPCHK(hipMemsetAsync(h->th, 0, n * 4, h->stream));
    HIPCHK(hipMemcpyAsync(h->w, ones.data(), n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->wm, ones.data(), n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemsetAsync(h->pose, 0, 16, h->stream));
    HIPCHK(hipMemsetAsync(h->start, 0, 16, h->stream));
    HIPCHK(hipMemsetAsync(h->grid, 0x9c /* -100 */, M, h->stream));
    HIPCHK(hipMemsetAsync(h->stats, 0, 64, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int pfslam_create(const pfslam_config *cfg, pfslam_handle **out)
{
    if (!cfg || !out) return fail("pfslam_create: null argument");
    if (cfg->n_particles <= 0 || cfg->n_beams <= 0 || cfg->kd_capacity <= 0)
        return fail("pfslam_create: n_particles, n_beams and kd_capacity must be positive");
    if (cfg->n_particles > (1 << 24)) return fail("pfslam_create: at most 2^24 particles per handle");
    if (cfg->n_beams > 4096) return fail("pfslam_create: at most 4096 beams per scan (the map insert keeps a frame's new walls in LDS)");
    if (cfg->kd_capacity > PF_KD_MAX_NODES) return fail("pfslam_create: kd_capacity above 2^27 - 1 nodes (32-bit byte offsets of the map records)");
    if (!(cfg->map_res_x > 0.0f && cfg->map_res_y > 0.0f && cfg->map_scale_x > 0.0f && cfg->map_scale_y > 0.0f))
        return fail("pfslam_create: map scale and resolution must be positive");
    {
        // map_dim = int(scale / resolution) per axis (kernel.cu:120).  The reference addresses cell (x, y) as x * dim.x + y
        // everywhere, which is only a bijection onto the dim.x * dim.y grid when the map is square (its own map is 40 x 40 m):
        // a wider map reads out of bounds, a taller one aliases rows.  Refuse instead of guessing a meaning.
        const int dx = (int)(cfg->map_scale_x / cfg->map_res_x), dy = (int)(cfg->map_scale_y / cfg->map_res_y);
        if (dx != dy) return fail("pfslam_create: the map must have as many cells in x as in y (reference cell index is x * dim.x + y)");
        // 16384 cells per side keeps every product of the closed-form Bresenham (k * deltay, cell index x * dim + y) inside int32
        if (dx < 2 || dx > 16384) return fail("pfslam_create: map dimensions out of range (2 .. 16384 cells per side)");
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(std::string("pfslam_create: no HIP device available (") + hipGetErrorString(e) +
                    "); this library has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail("pfslam_create: bad device ordinal");
    HIPCHK(hipSetDevice(cfg->device));
    pfslam_handle *h = new pfslam_handle();
    g_live_handles.fetch_add(1);
    h->cfg = *cfg;
    const int rc = create_impl(h);
    if (rc) { // release whatever was allocated before the failure; g_err keeps the message of the failure
        const std::string msg = g_err;
        pfslam_destroy(h);
        g_err = msg;
        return rc;
    }
    *out = h;
    return 0;
}

extern "C" int pfslam_destroy(pfslam_handle *h)
{
    if (!h) return 0;
    (void)hipSetDevice(h->cfg.device);
    (void)settle(h); // frames in flight finish first (their deferred errors are the caller's to collect: pfslam_synchronize)
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *bufs[] = {h->pblk, h->pblk2, h->w, h->wm, h->gkeys, h->mkeys, h->gtheta, h->scan_base, h->hot, h->parent, h->kz, h->kw,
                    h->fit, h->fit_i, h->partial, h->mkey, h->order2, h->cells, h->stats, h->pose, h->start, h->icp_tar, h->icp_cor, h->icp_dbg,
                    h->free_mask, h->blk_cnt, h->wall_cell, h->free_cell, h->wall_pts, h->free_pts,
                    h->wall_c, h->free_c, h->counts, h->tile_r, h->tile_r2, h->sums, h->cdf,
                    h->chunk_max, h->tile_tot, h->tile_off, h->tile_pmax, h->src, h->grid, h->d_count, h->wall_leaf, h->kd_state, h->d_sigma};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (h->own_global) {
        (void)hipFree(h->gw); (void)hipFree(h->gpose);
    }
    if (h->pin_tree) (void)hipHostFree(h->pin_tree);
    if (h->h_hdr) (void)hipHostFree(h->h_hdr);
    if (h->h_scan) (void)hipHostFree(h->h_scan);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->aux) { (void)hipStreamSynchronize(h->aux); (void)hipStreamDestroy(h->aux); }
    if (h->cstream) { (void)hipStreamSynchronize(h->cstream); (void)hipStreamDestroy(h->cstream); }
    if (h->istream) { (void)hipStreamSynchronize(h->istream); (void)hipStreamDestroy(h->istream); }
    if (h->ev_tree) (void)hipEventDestroy(h->ev_tree);
    if (h->ev_scored) (void)hipEventDestroy(h->ev_scored);
    if (h->ev_walked) (void)hipEventDestroy(h->ev_walked);
    if (h->topo_stream) { (void)hipStreamSynchronize(h->topo_stream); (void)hipStreamDestroy(h->topo_stream); }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    if (h->ev_mapfork) (void)hipEventDestroy(h->ev_mapfork);
    if (h->ev_map) (void)hipEventDestroy(h->ev_map);
    for (auto &e : h->ev_pool) (void)hipEventDestroy(e);
    for (auto &e : h->ev_pending) { (void)hipEventDestroy(e.a); if (!e.keep_b) (void)hipEventDestroy(e.b); }
    if (h->phase_ev) (void)hipEventDestroy(h->phase_ev);
    if (h->d_census) (void)hipFree(h->d_census);
    if (h->census_store) (void)hipFree(h->census_store);
    if (h->plan) (void)hipFree(h->plan);
    if (h->group_box) (void)hipFree(h->group_box);
    if (h->group_parts) (void)hipFree(h->group_parts);
    if (h->cell_tab) (void)hipFree(h->cell_tab);
    if (h->cell_list) (void)hipFree(h->cell_list);
    if (h->cell_state) (void)hipFree(h->cell_state);
    if (h->cell_pool) (void)hipFree(h->cell_pool);
    if (h->cell_rec) (void)hipFree(h->cell_rec);
    if (h->cell_touched) (void)hipFree(h->cell_touched);
    if (h->ev_boxes) (void)hipEventDestroy(h->ev_boxes);
    if (h->ev_marked) (void)hipEventDestroy(h->ev_marked);
    if (h->beam_angle) (void)hipFree(h->beam_angle);
    if (h->fit_acc) (void)hipFree(h->fit_acc);
    frame_free(h);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    g_live_handles.fetch_sub(1);
    return 0;
}

extern "C" int pfslam_set_stream(pfslam_handle *h, void *hip_stream)
{
    if (!h) return fail("null handle");
    CHK(settle(h));
    if (h->stream) HIPCHK(hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipStreamDestroy(h->stream));
    }
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    h->gates_tested = false; // the gates' self-test saw the old stream: it runs again in front of the next round-5 frame
    return 0;
}

extern "C" int pfslam_synchronize(pfslam_handle *h)
{
    if (!h) return fail("null handle");
    CHK(settle(h)); // books the frames in flight: a deferred error of one of them is reported here
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// Timed spans: pairs of HIP events on the stream the kernels are launched on, collected lazily.
static int flush_timers(pfslam_handle *h)
{
    for (auto &e : h->ev_pending) {
        HIPCHK(hipEventSynchronize(e.b));
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, e.a, e.b));
        h->timer_ms[e.slot] += ms;
        h->timer_count[e.slot] += 1;
        h->ev_pool.push_back(e.a);
        if (!e.keep_b) h->ev_pool.push_back(e.b);
    }
    h->ev_pending.clear();
    return 0;
}
static int timer_event(pfslam_handle *h, hipEvent_t *e)
{
    if (h->ev_pending.size() >= 512) CHK(flush_timers(h));
    if (h->ev_pool.empty()) {
        HIPCHK(pf_event_create(e, 0));
    } else {
        *e = h->ev_pool.back();
        h->ev_pool.pop_back();
    }
    return 0;
}
// phases of a frame (timing == 2): phase_begin records the start, phase_end closes the span into `slot` and starts the next
static int phase_begin(pfslam_handle *h)
{
    if (h->timing < 2) return 0;
    if (h->phase_ev) h->ev_pool.push_back(h->phase_ev);
    CHK(timer_event(h, &h->phase_ev));
    HIPCHK(hipEventRecord(h->phase_ev, h->stream));
    return 0;
}
static int phase_end(pfslam_handle *h, int slot)
{
    if (h->timing < 2 || !h->phase_ev) return 0;
    hipEvent_t e = nullptr;
    CHK(timer_event(h, &e));
    HIPCHK(hipEventRecord(e, h->stream));
    h->ev_pending.push_back(pfslam_handle::TimedSpan{h->phase_ev, e, slot});
    h->phase_ev = nullptr;
    CHK(timer_event(h, &h->phase_ev)); // the next phase starts where this one ended
    HIPCHK(hipEventRecord(h->phase_ev, h->stream));
    return 0;
}
extern "C" int pfslam_set_timing(pfslam_handle *h, int enable)
{
    if (!h) return fail("null handle");
    CHK(flush_timers(h));
    h->timing = enable;
    for (int k = 0; k < PF_TIMER_SLOTS; k++) {
        h->timer_ms[k] = 0.0;
        h->timer_count[k] = 0;
    }
    return 0;
}
extern "C" int pfslam_get_timers(pfslam_handle *h, double out[12])
{
    if (!h || !out) return fail("pfslam_get_timers: bad argument");
    CHK(flush_timers(h));
    for (int k = 0; k < PF_TIMER_SLOTS; k++) {
        out[2 * k] = h->timer_ms[k];
        out[2 * k + 1] = (double)h->timer_count[k];
    }
    return 0;
}

extern "C" int pfslam_set_variant(pfslam_handle *h, int variant)
{
    if (!h) return fail("null handle");
    h->variant = variant;
    return 0;
}

// [0, n) in chunks on up to 16 threads (host-side passes over the whole map: the re-balance is a stall of the frame pipeline)
template <typename F>
static void parallel_chunks(int n, F fn)
{
    const int nt = (int)std::min<unsigned>(16u, std::max(1u, std::min(std::thread::hardware_concurrency(), (unsigned)(n / 32768 + 1))));
    if (nt <= 1) {
        fn(0, n, 0);
        return;
    }
    std::vector<std::thread> th;
    const int per = (n + nt - 1) / nt;
    for (int t = 0; t < nt; t++) th.emplace_back([=] { fn(std::min(n, t * per), std::min(n, (t + 1) * per), t); });
    for (auto &t : th) t.join();
}

// upload a tree: host mirror + split device layout.  Everything is validated and packed BEFORE the handle changes, so a
// refused map leaves the previous one intact (host mirror, size and device arrays stay consistent).
// own: the caller's vector that `nodes` points into -- it becomes the host mirror (a swap instead of a 32-byte-per-node copy)
static int upload_tree(pfslam_handle *h, const pfslam_node *nodes, int n, std::vector<pfslam_node> *own = nullptr)
{
    if (n > h->kd_cap) return fail("map larger than kd_capacity");
    if (n == 0) {
        h->h_nodes.clear();
        h->kd_size = 0;
        h->mirror_n = 0;
        h->mirror_stale = false;
        h->cells_wipe_pending = true;
        HIPCHK(hipMemsetAsync(h->kd_state, 0, 16, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        return 0;
    }
    // the four device arrays are staged in pinned memory (28 B per node, allocated once for the map's capacity): a 500 k-node map goes
    // up in 0.5 ms instead of 2.3 from pageable vectors -- this runs inside the frame loop's re-balance
    std::vector<uint4> hot_v;
    std::vector<int> par_v;
    std::vector<float> z_v, w_v;
    if (!h->pin_tree) {
        void *p = nullptr;
        if (hipHostMalloc(&p, (size_t)h->kd_cap * 28, hipHostMallocDefault) == hipSuccess) h->pin_tree = p;
        else (void)hipGetLastError();
    }
    uint4 *hot;
    int *par;
    float *z, *w;
    if (h->pin_tree) {
        hot = (uint4 *)h->pin_tree;
        par = (int *)(hot + h->kd_cap);
        z = (float *)(par + h->kd_cap);
        w = z + h->kd_cap;
    } else {
        hot_v.resize(n); par_v.resize(n); z_v.resize(n); w_v.resize(n);
        hot = hot_v.data(); par = par_v.data(); z = z_v.data(); w = w_v.data();
    }
    // pass 1 (parallel): links in range, planar, integer weights, on the lattice of the config -- x == fl(k * res) bit for bit, the way
    // cell_to_point (ROUND_FRAC, kernel.cu:52) makes map points
    std::atomic<int> bad{-1}, nonplanar{0}, nonintegral{0}, offlattice{0}, wmax_bits{0};
    const float rx = h->cfg.map_res_x, ry = h->cfg.map_res_y, ix = 1.0f / rx, iy = 1.0f / ry;
    const float xmax = rx * (float)PF_LATTICE_KMAX, ymax = ry * (float)PF_LATTICE_KMAX;
    parallel_chunks(n, [&](int lo, int hi, int) {
        bool np = false, ni = false, ol = false;
        float wm = 0.0f;
        for (int i = lo; i < hi; i++) {
            const pfslam_node &nd = nodes[i];
            if (nd.axis < 0 || nd.axis > 2 || nd.left < -1 || nd.left >= n || nd.right < -1 || nd.right >= n || nd.parent < -1 || nd.parent >= n) {
                int none = -1;
                bad.compare_exchange_strong(none, i);
                return;
            }
            par[i] = nd.parent;
            z[i] = nd.z;
            w[i] = nd.w;
            np |= nd.z != 0.0f;
            // |w| <= 2^13 keeps every partial sum of 1081 weights below 2^24, i.e. exact in any order
            ni |= !(nd.w == (float)(int)nd.w && fabsf(nd.w) <= 8192.0f);
            wm = nd.w == nd.w ? std::max(wm, fabsf(nd.w)) : INFINITY;
            if (!ol) {
                if (!(fabsf(nd.x) < xmax && fabsf(nd.y) < ymax)) ol = true; // |k| < 2^20 cells per axis (PF_LATTICE_KMAX); NaN fails
                else {
                    const float kx = roundf(nd.x * ix), ky = roundf(nd.y * iy);
                    ol = !((kx * rx == nd.x || (kx + 1.0f) * rx == nd.x || (kx - 1.0f) * rx == nd.x) &&
                           (ky * ry == nd.y || (ky + 1.0f) * ry == nd.y || (ky - 1.0f) * ry == nd.y));
                }
            }
        }
        {
            int cur = wmax_bits.load(), mine;
            memcpy(&mine, &wm, 4);
            while (mine > cur && !wmax_bits.compare_exchange_weak(cur, mine)) {}
        }
        if (np) nonplanar = 1;
        if (ni) nonintegral = 1;
        if (ol) offlattice = 1;
    });
    if (bad.load() >= 0) return fail("pfslam_set_map: node " + std::to_string(bad.load()) + " has out-of-range links or axis");
    const int planar = nonplanar.load() ? 0 : 1;
    const bool integral = nonintegral.load() == 0, lattice = planar && offlattice.load() == 0;
    parallel_chunks(n, [&](int lo, int hi, int) { // pass 2: the hot records need `planar`
        for (int i = lo; i < hi; i++) {
            const pfslam_node &nd = nodes[i];
            hot[i] = pf::pack_hot(nd.x, nd.y, nd.axis, nd.left, nd.right, planar != 0);
            if (planar && nd.axis == 2) memcpy(&z[i], &nd.left, 4); // true left child of a planar z-level node
        }
    });
    HIPCHK(hipMemcpyAsync(h->hot, hot, (size_t)n * 16, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->parent, par, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->kz, z, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->kw, w, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    float wabs;
    {
        const int wb = wmax_bits.load();
        memcpy(&wabs, &wb, 4);
        wabs = std::max(wabs, PF_CLAMP_VAL); // the map update moves a weight by -1 / +4 and clamps to +-113: never beyond the larger of the two
    }
    // [1 .. 3]: for the ranks that adopt this tree (pfslam_shard_balance_adopt); [2]: 0 = weights not all integers, else their largest magnitude
    const int state[4] = {n, planar, integral ? (int)std::min(wabs, 1e9f) : 0, lattice ? 1 : 0};
    HIPCHK(hipMemcpyAsync(h->kd_state, state, 16, hipMemcpyHostToDevice, h->stream));
    if (own && own->data() == nodes && (int)own->size() == n) h->h_nodes.swap(*own); // (while the copies are in flight)
    else h->h_nodes.assign(nodes, nodes + n);
    HIPCHK(hipStreamSynchronize(h->stream));
    h->kd_size = n;
    h->mirror_n = n;
    h->mirror_stale = false;
    h->planar = planar;
    h->integral_w = integral;
    h->w_absmax = wabs;
    h->lattice_ok = lattice;
    h->cells_wipe_pending = true; // rows of the previous map
    h->pipe_live = false;         // (round-5 frames: the tree-size words of the frame ring start over)
    h->cells_suspended = false;
    h->cells_full_frame = -1;
    return 0;
}

extern "C" int pfslam_set_map(pfslam_handle *h, const pfslam_node *nodes, int n)
{
    if (!h || (n > 0 && !nodes) || n < 0) return fail("pfslam_set_map: bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    return upload_tree(h, nodes, n);
}

extern "C" int pfslam_set_particles(pfslam_handle *h, const pfslam_particle *p, int n)
{
    if (!h || !p || n != h->n) return fail("pfslam_set_particles: n must equal cfg.n_particles");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    std::vector<float> tmp(4 * (size_t)n);
    for (int i = 0; i < n; i++) {
        tmp[i] = p[i].x; tmp[n + i] = p[i].y; tmp[2 * (size_t)n + i] = p[i].theta; tmp[3 * (size_t)n + i] = p[i].w;
    }
    { // spread of the new cloud, as k_cell_count will see it (first 1024 slots): decides how the next scoring pass is organised
        const int ns = std::min(n, 1024);
        double m[3] = {0, 0, 0}, v[3] = {0, 0, 0};
        for (int i = 0; i < ns; i++) { m[0] += p[i].x; m[1] += p[i].y; m[2] += p[i].theta; }
        for (int k = 0; k < 3; k++) m[k] /= ns;
        for (int i = 0; i < ns; i++) {
            const double d0 = p[i].x - m[0], d1 = p[i].y - m[1], d2 = p[i].theta - m[2];
            v[0] += d0 * d0; v[1] += d1 * d1; v[2] += d2 * d2;
        }
        const double sg = std::max(std::max(std::sqrt(v[0] / ns), std::sqrt(v[1] / ns)), std::sqrt(v[2] / ns) * h->scan_reach);
        h->cloud_sigma = std::isfinite(sg) ? (float)sg : 0.0f;
        float tb = 0.0f;
        for (int i = 0; i < n; i++) tb = p[i].theta == p[i].theta ? std::max(tb, fabsf(p[i].theta)) : INFINITY;
        h->theta_bound = tb;
        h->theta_shift = 0.0f;
        if (h->world > 1) h->theta_global_known = false; // (this call saw one shard's headings only)
        h->cloud_valid = false; // (round-5 frames: k_motion_count's statistics describe the cloud this call replaces)
    }
    HIPCHK(hipMemcpyAsync(h->x, &tmp[0], (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->y, &tmp[n], (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->th, &tmp[2 * (size_t)n], (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->w, &tmp[3 * (size_t)n], (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->wm, &tmp[3 * (size_t)n], (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

// mean length of the beams that can score (finite, inside the +-20 m reject in at least one heading): performance hint only
static void scan_reach_of(pfslam_handle *h, const float *scan)
{
    double sum = 0.0;
    int cnt = 0;
    for (int j = 0; j < h->nb; j++) {
        const float r = fabsf(scan[j]);
        if (r < 28.3f) { // 20 * sqrt(2); NaN fails the compare
            sum += r;
            cnt++;
        }
    }
    h->scan_reach = cnt ? (float)std::min(std::max(sum / cnt, 0.5), 30.0) : 8.0f;
}

extern "C" int pfslam_set_scan(pfslam_handle *h, const float *scan_host, int n_beams)
{
    if (!h || !scan_host || n_beams != h->nb) return fail("pfslam_set_scan: n_beams must equal cfg.n_beams");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle_staged(h));
    scan_reach_of(h, scan_host);
    HIPCHK(hipMemcpyAsync(h->scan, scan_host, (size_t)n_beams * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream)); // scan_host is pageable: do not return before it is consumed
    return 0;
}

extern "C" int pfslam_set_pose(pfslam_handle *h, const float pose[3])
{
    if (!h || !pose) return fail("pfslam_set_pose: bad argument");
    CHK(settle_staged(h));
    float p4[4] = {pose[0], pose[1], pose[2], 0.0f};
    memcpy(h->h_pose, pose, 12);
    HIPCHK(hipMemcpyAsync(h->pose, p4, 16, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int pfslam_get_particles(pfslam_handle *h, const pfslam_particle **out, int *n)
{
    if (!h || !out || !n) return fail("pfslam_get_particles: bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    const size_t N = h->n;
    h->h_tmp.resize(4 * N);
    HIPCHK(hipMemcpyAsync(&h->h_tmp[0], h->x, N * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(&h->h_tmp[N], h->y, N * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(&h->h_tmp[2 * N], h->th, N * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(&h->h_tmp[3 * N], h->w, N * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->h_particles.resize(N);
    for (size_t i = 0; i < N; i++) {
        pfslam_particle &p = h->h_particles[i];
        memset(&p, 0, sizeof(p));
        p.x = h->h_tmp[i]; p.y = h->h_tmp[N + i]; p.theta = h->h_tmp[2 * N + i]; p.w = h->h_tmp[3 * N + i];
    }
    *out = h->h_particles.data();
    *n = (int)N;
    return 0;
}

extern "C" int pfslam_get_map(pfslam_handle *h, const pfslam_node **out, int *n)
{
    if (!h || !out || !n) return fail("pfslam_get_map: bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    const size_t K = h->kd_size;
    if (K) {
        // The device owns the tree: weights change every frame and the map update inserts there.  The host mirror keeps what
        // never changes (position, axis, parent of the nodes it has seen); links and weights are read back, and so are the nodes
        // appended since the last call.  (A planar z-level node never gains a LEFT child, so its mirrored left link stays valid.)
        const size_t M0 = std::min((size_t)h->mirror_n, K);
        std::vector<uint4> hot(K);
        h->h_tmp.resize(K);
        std::vector<int> par(K - M0);
        std::vector<float> z(K - M0);
        HIPCHK(hipMemcpyAsync(h->h_tmp.data(), h->kw, K * 4, hipMemcpyDeviceToHost, h->stream));
        if (h->mirror_stale) {
            HIPCHK(hipMemcpyAsync(hot.data(), h->hot, K * 16, hipMemcpyDeviceToHost, h->stream));
            if (K > M0) {
                HIPCHK(hipMemcpyAsync(par.data(), h->parent + M0, (K - M0) * 4, hipMemcpyDeviceToHost, h->stream));
                HIPCHK(hipMemcpyAsync(z.data(), h->kz + M0, (K - M0) * 4, hipMemcpyDeviceToHost, h->stream));
            }
        }
        HIPCHK(hipStreamSynchronize(h->stream));
        if (h->mirror_stale) {
            h->h_nodes.resize(K);
            const bool planar = h->planar != 0;
            for (size_t i = 0; i < K; i++) {
                pfslam_node &nd = h->h_nodes[i];
                const int axis = (int)(hot[i].z >> 30);
                if (i >= M0) {
                    memcpy(&nd.x, &hot[i].x, 4);
                    memcpy(&nd.y, &hot[i].y, 4);
                    nd.axis = axis;
                    nd.parent = par[i - M0];
                    nd.z = planar ? 0.0f : z[i - M0];
                    nd.left = -1;
                    if (planar && axis == 2) memcpy(&nd.left, &z[i - M0], 4);
                }
                if (!(planar && axis == 2)) nd.left = pf::hot_left(hot[i].z);
                nd.right = (int)hot[i].w;
            }
            h->mirror_n = (int)K;
            h->mirror_stale = false;
        }
        for (size_t i = 0; i < K; i++) h->h_nodes[i].w = h->h_tmp[i];
    }
    *out = h->h_nodes.data();
    *n = (int)K;
    return 0;
}

extern "C" int pfslam_get_pose(pfslam_handle *h, float pose[3])
{
    if (!h || !pose) return fail("pfslam_get_pose: bad argument");
    CHK(settle(h));
    float p4[4];
    HIPCHK(hipMemcpyAsync(p4, h->pose, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(pose, p4, 12);
    memcpy(h->h_pose, p4, 12);
    return 0;
}

extern "C" int pfslam_get_trace(pfslam_handle *h, int32_t out[8])
{
    if (!h || !out) return fail("pfslam_get_trace: bad argument");
    CHK(settle(h)); // the trace of the LAST frame: every frame in flight is booked first
    memcpy(out, h->trace, sizeof(h->trace));
    return 0;
}

// ---- A3 -------------------------------------------------------------------------------------
extern "C" int pfslam_motion_update(pfslam_handle *h, int frame)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle_staged(h));
    hipLaunchKernelGGL(k_motion, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, h->x, h->y, h->th, h->w, h->wm,
                       h->n, frame, h->goff, h->trig);
    HIPCHK(hipGetLastError());
    h->theta_bound += 0.1f;
    return 0;
}

// ---- odometry hook (no reference counterpart: the reference's filter has no motion model besides the diffusion) -------------
__global__ __launch_bounds__(256) void k_shift(float *__restrict__ x, float *__restrict__ y, float *__restrict__ th, int n, float dx, float dy,
                                               float dt, float *__restrict__ pose, float *__restrict__ cloud)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) { pose[0] = pose[0] + dx; pose[1] = pose[1] + dy; pose[2] = pose[2] + dt; }
    if (i < 2 && cloud) { // round-5 frames: the cloud statistics of both ticket parities move along (performance only)
        cloud[4 * i] += dx; cloud[4 * i + 1] += dy; cloud[4 * i + 2] += dt;
    }
    if (i >= n) return;
    x[i] = x[i] + dx;
    y[i] = y[i] + dy;
    th[i] = th[i] + dt;
}
// Every pose the filter holds -- all particles and robotPos -- moves by the same increment (one float addition per component).
// Enqueued behind the frames in flight; no host wait.
extern "C" int pfslam_shift_particles(pfslam_handle *h, const float delta[3])
{
    if (!h || !delta) return fail("pfslam_shift_particles: bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(join_map(h)); // the map update a frame left on the aux stream reads the pose
    if (h->pipe_live && !h->serial) { // round-5 frames: the pose's readers of the last frame are k_walls (chain) and the free-cell chain
        HIPCHK(hipEventRecord(h->ev_tail[0], h->aux));
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_tail[0], 0));
        if (!h->ftail_recorded) { // (stream gates: the free-cell chain's tail is not recorded per frame)
            HIPCHK(hipEventRecord(h->ev_ftail, h->fstream));
            h->ftail_recorded = true;
        }
        HIPCHK(hipStreamWaitEvent(h->stream, h->ev_ftail, 0));
    }
    const float ad = delta[2] == delta[2] ? fabsf(delta[2]) : INFINITY;
    h->theta_shift += ad; // (headers of frames enqueued before this do not know of it)
    h->theta_bound += ad; // ... and the choice of the scan-match kernel's instantiation must: the bound itself moves at once
    h->theta_shift_seq = h->seq;
    hipLaunchKernelGGL(k_shift, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, h->x, h->y, h->th, h->n, delta[0], delta[1], delta[2], h->pose,
                       h->cloud_valid ? h->cloud : (float *)nullptr);
    HIPCHK(hipGetLastError());
    if (h->pipe_live) {
        HIPCHK(hipEventRecord(h->ev_shift, h->stream));
        h->shift_pending = true;
    }
    return 0;
}

// ---- A5 -------------------------------------------------------------------------------------
static int score_chunks(const pfslam_handle *h)
{
    // ~16 rounds of the 8192 wave slots (256 CUs x 32): fine-grained enough that the tail of the last round is
    // small (measured: 16 k waves 3.33 ms, 128 k waves 2.9-3.0 ms at 100 k particles); down to one beam per chunk for small N
    const int groups = (h->n + 63) / 64;
    // (round 5: 65 536 -- 42 chunks of 26 beams at 100 k particles.  A wave's prologue, the fp64 sincos of its 64 headings, is ~10 % of a
    // 13-beam chunk, and the reduce reads half the partials: frame 0.530 -> 0.520 ms in an A/B on one box; 32 768: 0.538, 262 144: 0.569)
    // Below ~40 k particles fewer, longer waves win (a wave's prologue and the launch's ramp against the beams it scores): ~160 chunks per
    // group between 24 576 and 65 536 waves (profiles/r05_sweep_waves.txt: 10 k particles 0.218 -> 0.209 ms per frame, 20 k 0.2493 -> 0.2478,
    // 1 k / 4 k unchanged within noise)
    static const int target_env = ab_env("PFSLAM_TARGET_WAVES") ? atoi(ab_env("PFSLAM_TARGET_WAVES")) : 0;
    const int target = target_env > 0 ? target_env : std::max(24576, std::min(65536, groups * 160));
    int chunks = (target + groups - 1) / groups;
    chunks = std::max(1, std::min(chunks, h->nb)); // small particle counts go down to one beam per wave
    // Beam-chunk partials added afterwards equal the reference's sequential beam-order float sum only when every term is an
    // integer (any map the SLAM step builds: 0 / -100 initial, -1 / +4 steps, clamp +-113).  A map uploaded through
    // pfslam_set_map with other weights is scored in one chunk: slower, but kernEvaluateParticlesKD's own summation order.
    if (!h->integral_w) chunks = 1;
    return chunks;
}

// ---- persistent lattice-cell rows: host side ---------------------------------------------------------------------------------
#define PF_CELLS_WALK_GRID 32 /* ... of k_cells_update<false>: it runs beside the scan-match kernel, finds a handful of new cells (at the list's end:
                                 one or two waves) and otherwise only reads record headers -- 512 workgroups with 47 KB of LDS each, on a high-priority
                                 stream, cost that kernel anything between 0 and 80 us from run to run */
#define PF_CELLS_GRID 512 /* workgroups (one wave each) of k_cells_update, grid-stride over the records: two fit a CU (78 KB of LDS each), so 512
                             are resident together -- 2048 of them queued through the few free places while the first 476 worked (56 us) */
struct CellArgs { unsigned *tab; int *list, *cs; uint4 *pool; int *rec; };
__global__ void k_cells_reset(int *cs, int list_cap, int pool_cap)
{
    if (threadIdx.x < PF_CS_WORDS)
        cs[threadIdx.x] = threadIdx.x == PF_CS_FLAGS ? PF_CF_NEED_ORIGIN : threadIdx.x == PF_CS_LIST_CAP ? list_cap : threadIdx.x == PF_CS_POOL_CAP ? pool_cap : 0;
}
// table, records and pool start over (enqueued; the caller has joined the aux stream).  105 MB of memset: rare -- a new map, a
// re-balance (an 11 ms stall of its own), an exhausted list / pool, a cloud that has left the middle of the window.
static int cells_wipe(pfslam_handle *h)
{
    HIPCHK(hipMemsetAsync(h->cell_tab, 0, (size_t)PF_CELL_WIN * PF_CELL_WIN * 4, h->stream));
    // (the capacities in use: the allocations, unless a test of the overflow paths asks for less)
    static const int list_cap = getenv("PFSLAM_CELL_LIST_CAP") ? std::min(std::max(atoi(getenv("PFSLAM_CELL_LIST_CAP")), 64), PF_CELL_LIST_CAP) : PF_CELL_LIST_CAP;
    static const int pool_cap = getenv("PFSLAM_CELL_POOL_CAP") ? std::min(std::max(atoi(getenv("PFSLAM_CELL_POOL_CAP")), 64), PF_CELL_POOL_CAP) : PF_CELL_POOL_CAP;
    hipLaunchKernelGGL(k_cells_reset, dim3(1), dim3(64), 0, h->stream, h->cell_state, list_cap, pool_cap);
    HIPCHK(hipGetLastError());
    h->cells_wipe_pending = false;
    h->cells_wipe_seq = h->seq;
    h->cells_wipes++;
    h->cells_gen++;
    h->cells_passes = 0;
    return 0;
}
// behind every insert into the tree (k_test_new), on the stream that ran it: the records' links are looked at, the cells that gained
// a node are extended, their rows and those of the cells walked under the scan-match kernel are published
static int launch_cells_update(pfslam_handle *h, hipStream_t st)
{
    if (!h->cell_tab || h->cells_wipe_pending) return 0; // (a pending wipe: the records belong to a map that is gone)
    const CellGeom geo{h->cfg.map_res_x, h->cfg.map_res_y, 1.0f / h->cfg.map_res_x, 1.0f / h->cfg.map_res_y};
    // Every PF_CELLS_RECUT_EVERY-th update cuts EVERY cell's rows again, into a pool that starts over: rows that outgrow their place
    // move to the pool's end, and after ~20 frames the rows a wave gathers together no longer sit together -- the scan-match kernel,
    // with nothing running beside it, went from 0.37 to 0.40-0.42 ms between the 10th and the 20th frame after a wipe (the round-3
    // build, which cut everything every frame, stayed flat).  A cut needs no walk: the records hold the candidates.
    static const int recut_every = ab_env("PFSLAM_CELLS_RECUT_EVERY") ? atoi(ab_env("PFSLAM_CELLS_RECUT_EVERY")) : 16; // (as in the round-5 frame: tools/experiments/r05/recut_ab.sh)
    const int recut = recut_every > 0 && h->cells_passes > 0 && h->cells_passes % recut_every == 0 ? 1 : 0;
    if (recut) HIPCHK(hipMemsetAsync(h->cell_state + PF_CS_POOL, 0, 4, st));
    hipLaunchKernelGGL(k_cells_update<true>, dim3(PF_CELLS_GRID), dim3(64), 0, st, kd_view(h), geo, h->cell_tab, (const int *)h->cell_list, h->cell_state,
                       h->cell_pool, h->cell_rec, h->cells_gen, (const int *)h->cell_touched, (int)(h->cells_passes & 15), recut, h->cells_snap ? 1 : 0);
    // Below the snapshot whenever this frame's scoring pass left one (the walk pass of an asynchronous frame, the publishing pass of a
    // synchronous one: the next frame's marking pass may already be bumping the counter in front of list entries it has not written yet).
    // A frame scored WITHOUT a cell pass (the plan, the plain traversal) has none -- the last one may be frames old, or wiped: every record
    // is looked at, and the next marking pass waits for this update instead (launch_score).
    h->update_unsnapped = !h->cells_snap;
    h->cells_snap = false;
    HIPCHK(hipGetLastError());
    h->cells_passes++;
    return 0;
}

// How a scoring pass is organised (launch_score and the round-5 frame agree through this): lattice-cell rows, the round-2 plan, or the plain traversal
static int plan_min_particles()
{
    static const int v = getenv("PFSLAM_PLAN_MIN_N") ? atoi(getenv("PFSLAM_PLAN_MIN_N")) : 4608;
    return v;
}
// frame_loop: the round-5 frame of pfslam_step -- nothing on its chain waits for the marking pass or the walks of new cells, so the rows pay
// from a few waves of particles on (1000 particles: 0.160 -> 0.144 ms per frame, 3000: 0.229 -> 0.177); everywhere else (stage-level calls,
// the sharded frame: synchronous or same-frame passes) they start at ~4.6 k particles
static bool org_use_cells(const pfslam_handle *h, bool *use_plan, bool frame_loop = false)
{
    static const bool env_min = getenv("PFSLAM_PLAN_MIN_N") != nullptr;
    const int plan_min_n = (frame_loop && !env_min) ? 65 : plan_min_particles();
    const bool organised = h->planar && h->variant != 2 && h->variant != 1 && h->n > 64 && (h->n >= plan_min_n || h->variant >= 3);
    const float Dside = h->n <= 400000 ? 64.0f : 128.0f;
    const float box_cells = 2.0f * (6.4f * h->cloud_sigma / Dside) * cbrtf(64.0f * Dside * Dside * Dside / (float)h->n) / std::min(h->cfg.map_res_x, h->cfg.map_res_y);
    static const float box_max = ab_env("PFSLAM_CELLS_BOX_MAX") ? (float)atof(ab_env("PFSLAM_CELLS_BOX_MAX")) : 24.0f;
    const bool narrow = h->variant == 3 || !(box_cells > box_max);
    const bool use_cells = organised && h->lattice_ok && h->variant != 4 && !h->cells_suspended && narrow; // lattice-cell rows (kd_cells.hip.inc)
    if (use_plan) *use_plan = !use_cells && h->planar && h->variant != 2 && (h->n >= plan_min_n || h->variant >= 3);
    return use_cells;
}
__global__ void k_reduce_partials_minmax(float *partial, int n, int chunks, const int *order, float *fit, int goff, long long *stats,
                                         const float *x, const float *y, const float *th, int wipe, int p16);
__global__ void k_reduce_partials_minmax_wide(const float *partial, int n, int chunks, const int *order, float *fit, long long *stats);
template <typename T> __global__ void k_minmax(const T *fit, int n, int goff, long long *stats);
static int join_icp(pfslam_handle *h);
static int launch_stats_reset(pfslam_handle *h, hipStream_t st);

// fuse_minmax: the frame loops want the packed min/max keys of this shard right away; the reduce kernel then produces them
// too (one launch less on the chain).  Needs more than one beam chunk, which every launch below ~8 M particles has.
// census: run the counting instantiation of the score kernel instead (same launch shape, same results).
static int launch_score(pfslam_handle *h, bool fuse_minmax = false, pf::KdCensus *census = nullptr, bool sharded = false)
{
    if (h->kd_size <= 0) return fail("pfslam_score_kd: no map loaded");
    const int chunks = score_chunks(h);
    const int bpc = (h->nb + chunks - 1) / chunks;
    const int used = (h->nb + bpc - 1) / bpc;
    float *out = h->fit;
    if (used > 1) {
        const size_t need = (size_t)used * h->n;
        if (need > h->partial_elems) {
            if (h->partial) HIPCHK(hipFree(h->partial));
            h->partial = nullptr;
            CHK(dalloc(&h->partial, need));
            h->partial_elems = need;
        }
        out = h->partial;
    }
    const int *order = nullptr;
    // How the scoring pass is organised (results are bit-identical): variant 0 = default; 1 = identity lane order; 2 = plain per-lane
    // traversal; 3 = plan / cell rows at any particle count; 4 = the round-2 shared-prefix plan instead of cell rows (A/B, tests).
    // With few particles neither pays for its extra launches (~0.1 ms of marking + rows): measured scoring pass at 3 k / 5 k / 10 k /
    // 50 k particles: cell rows 0.175 / 0.185 / 0.251 / 0.502 ms, plan 0.162 / 0.204 / 0.312 / 0.825 ms, plain traversal 0.139 / 0.198 /
    // 0.351 / 1.307 ms (tools/experiments/r03/cells_threshold.py): organised from ~4.6 k particles on.
    bool use_plan_ = false;
    const bool use_cells_ = org_use_cells(h, &use_plan_);
    // The marking pass of the cell rows costs the area of a wave's beam-end box in lattice cells: a wave's 64 Hilbert neighbours span
    // ~(64 D^3 / N)^(1/3) of the D^3 cells laid over +-3.2 sigma, in position and (x reach) in heading.  Up to ~24 cells per side the
    // rows win (0.5 m of spread at 100 k particles and 2.5 cm: cell rows 0.48 / 0.84 / 1.9 ms at sigma 0.05 / 0.2 / 0.5 m against 1.2 / 1.9 / 2.4 ms
    // for the plan); at 1 m they took 60 ms against 2.7 (tools/sigma_sweep.py, profiles/r04_sigma_sweep.json).  variant 3 forces them.
    const bool use_cells = use_cells_, use_plan = use_plan_; // (org_use_cells: lattice-cell rows, else the round-2 plan, else the plain traversal)
    const CellGeom geo{h->cfg.map_res_x, h->cfg.map_res_y, 1.0f / h->cfg.map_res_x, 1.0f / h->cfg.map_res_y};
    if (use_cells && !h->cell_tab) {
        CHK(dalloc(&h->cell_tab, (size_t)PF_CELL_WIN * PF_CELL_WIN));
        CHK(dalloc(&h->cell_list, (size_t)PF_CELL_LIST_CAP));
        HIPCHK(hipMemsetAsync(h->cell_list, 0, (size_t)PF_CELL_LIST_CAP * 4, h->stream));
        CHK(dalloc(&h->cell_state, PF_CS_ALLOC)); // ([64, 128): PF_CELLS_PROFILE builds)
        HIPCHK(hipMemsetAsync(h->cell_state, 0, PF_CS_ALLOC * 4, h->stream));
        CHK(dalloc(&h->cell_rec, (size_t)PF_CELL_LIST_CAP * PF_REC_WORDS));
        HIPCHK(hipMemsetAsync(h->cell_rec, 0, (size_t)PF_CELL_LIST_CAP * PF_REC_WORDS * 4, h->stream)); // generation 0: no record is written
        CHK(dalloc(&h->cell_pool, (size_t)PF_CELL_POOL_CAP + PF_ROW_SLACK));
        CHK(dalloc(&h->cell_touched, (size_t)h->max_wall + 1));
        HIPCHK(hipMemsetAsync(h->cell_touched, 0, 4, h->stream));
        CHK(dalloc(&h->fit_acc, (size_t)h->n));
        HIPCHK(pf_event_create(&h->ev_boxes, hipEventDisableTiming));
        HIPCHK(pf_event_create(&h->ev_marked, hipEventDisableTiming));
        HIPCHK(pf_event_create(&h->ev_walked, hipEventDisableTiming));
        HIPCHK(hipMemsetAsync(h->fit_acc, 0, (size_t)h->n * 4, h->stream));
        h->cells_wipe_pending = true;
    }
    // The rows persist (kd_cells.hip.inc).  Synchronous pass: new cells are marked, walked and published in front of the scan-match
    // kernel, on this stream (stage-level calls, the first pass after a wipe, per-phase timing).  Asynchronous pass (frame loops):
    // marking and the walk of the new cells run on the aux stream UNDER the scan-match kernel, whose lanes take the generic traversal
    // in a cell that has no rows yet; k_cells_update publishes them behind the frame's insert (launch_map_update_device).
    static const int cells_mode = ab_env("PFSLAM_CELLS_MODE") ? atoi(ab_env("PFSLAM_CELLS_MODE")) : 0; // 1: always synchronous (A/B)
    const bool cells_sync = use_cells && (!h->cells_async || h->cells_wipe_pending || census != nullptr || cells_mode == 1);
    if (use_cells && h->cells_wipe_pending) {
        CHK(join_map(h)); // the previous frame's k_cells_update is the table's last writer
        CHK(cells_wipe(h));
    }
    // default: counting sort over Hilbert cells of the cloud (3 launches), 2^18 cells up to 400 k particles, 2^21 above
    static const float theta_weight = ab_env("PFSLAM_THETA_WEIGHT") ? (float)atof(ab_env("PFSLAM_THETA_WEIGHT")) : 1.0f;
    if (h->variant != 1 && h->n > 64) {
        const int bits = h->n <= 400000 ? 6 : PF_CELL_BITS_MAX, ncell = 1 << (3 * bits);
        int *hist = h->cells, *cursor = h->cells + ncell, *tile_tot = h->cells + 2 * ncell;
        hipLaunchKernelGGL(k_cell_count, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, h->x, h->y, h->th, h->n, h->scan_reach * theta_weight, bits, h->mkey, hist,
                           use_cells ? h->cell_state : (int *)nullptr, geo, h->d_sigma);
        hipLaunchKernelGGL(k_cell_scan, dim3(ncell / 1024), dim3(256), 0, h->stream, hist, cursor, tile_tot);
        hipLaunchKernelGGL(k_cell_scatter, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, h->mkey, h->n, cursor, tile_tot, ncell / 1024, h->order2);
        HIPCHK(hipGetLastError());
        order = h->order2;
    } // variant 1 = identity lane order
    // cell-row kernel on a map with integer weights, several beam chunks, frame loop: the chunks ADD their sums to one accumulator
    // per lane (exact in any order) instead of writing `used` partials per lane for the reduce kernel to read back (33 MB per frame)
    // MEASURED AND OFF BY DEFAULT (PFSLAM_ACC_OUT=1 turns it on): the reduce kernel gets 10 us shorter and 33 MB of writes go away, but
    // 8.4 M float atomics cost the scan-match kernel 30 us (0.626 -> 0.656 ms) -- plain stores retire for free next to VALU-bound work
    static const bool acc_enabled = ab_env("PFSLAM_ACC_OUT") && atoi(ab_env("PFSLAM_ACC_OUT")) != 0;
    const bool acc_out = acc_enabled && use_cells && h->integral_w && used > 1 && fuse_minmax && !census;
    const int direct = used > 1 ? 0 : 1;
    // beam-chunk partials of the cell-row kernel as 16-bit integers: integer weights, and a chunk's sum cannot leave the range
    const bool wide_reduce = used >= 256 && fuse_minmax && !sharded && h->goff == 0 && !acc_out;
    static const bool p16_ok = !(ab_env("PFSLAM_P16") && atoi(ab_env("PFSLAM_P16")) == 0);
    const bool p16 = p16_ok && use_cells && h->integral_w && used > 1 && fuse_minmax && !acc_out && !wide_reduce && (float)bpc * h->w_absmax <= 32767.0f;
    h->plan_valid = use_plan;
    h->cells_valid = use_cells;
    if (use_plan || use_cells) { // the pose boxes do not need the map: in front of the join
        const int groups = (h->n + 63) / 64;
        if (!h->group_box) CHK(dalloc(&h->group_box, (size_t)groups)); // n is fixed for the handle's lifetime
        if (!h->group_parts) CHK(dalloc(&h->group_parts, (size_t)groups));
        const size_t rows = (size_t)groups * h->nb;
        if (use_plan && rows > h->plan_rows) {
            if (h->plan) HIPCHK(hipFree(h->plan));
            h->plan = nullptr;
            CHK(dalloc(&h->plan, rows));
            h->plan_rows = rows;
        }
        if (h->mark_on_aux) { // the previous frame's marking pass (aux stream, long finished) read the boxes this launch overwrites
            HIPCHK(hipStreamWaitEvent(h->stream, h->ev_marked, 0));
            h->mark_on_aux = false;
        }
        hipLaunchKernelGGL(k_group_box, dim3(groups), dim3(64), 0, h->stream, h->x, h->y, h->th, h->n, order, h->group_box, h->group_parts, h->d_sigma);
    }
    hipEvent_t t_a = nullptr, t_b = nullptr;
    if (h->timing && !census) {
        CHK(timer_event(h, &t_a));
        CHK(timer_event(h, &t_b));
        HIPCHK(hipEventRecord(t_a, h->stream));
    }
    const CellArgs ca{h->cell_tab, h->cell_list, h->cell_state, h->cell_pool, h->cell_rec};
    if (use_cells) { // the cells the beam ends can fall into: needs the pose boxes and the scan, not the map -- still in front of the join
        hipStream_t st = h->stream;
        if (!cells_sync) {
            // On a stream of their own.  The marking pass needs the pose boxes and the scan only, and claims nothing but table words that
            // are still zero, so it starts NOW -- beside the previous frame's map update on the aux stream, while this stream waits for
            // that anyway -- and is over when the scan-match kernel starts: running beside THAT, its 1081 four-wave workgroups cost the
            // scan-match kernel 27 us (0.373 -> 0.400 ms).  The walks wait for the previous frame's k_cells_update (below).
            HIPCHK(hipEventRecord(h->ev_boxes, h->stream));
            HIPCHK(hipStreamWaitEvent(h->cstream, h->ev_boxes, 0));
            st = h->cstream;
        }
        if (h->update_unsnapped) { // (see launch_cells_update; an event that has fired -- or was never recorded -- costs nothing)
            HIPCHK(hipStreamWaitEvent(st, h->ev_map, 0));
            h->update_unsnapped = false;
        }
        const int groups = (h->n + 63) / 64, per_block = PF_MARK_THREADS * PF_MARK_GROUPS;
        hipLaunchKernelGGL(k_cells_mark, dim3(h->nb, (groups + per_block - 1) / per_block), dim3(PF_MARK_THREADS), 0, st,
                           (const pf::KdGroupBox *)h->group_box, groups, (const float *)h->scan, h->nb, geo, h->cell_tab, h->cell_list, h->cell_state,
                           (const pf::AngleParts *)h->group_parts, (const pf::BeamParts *)h->beam_angle);
        if (!cells_sync) { // the new cells' walks from the root: records only, published behind this frame's insert
            HIPCHK(hipEventRecord(h->ev_marked, st));
            h->mark_on_aux = true;
            // The tree as the previous frame's insert left it, and BEHIND that frame's k_cells_update -- always, not only while the host
            // still has that map update down as unjoined: a getter between two frames (pfslam_get_trace books the frame as soon as its
            // header is there, which k_test_new writes BEFORE k_cells_update runs) clears that flag, and the walk pass then ran beside the
            // previous k_cells_update, which could pick up a record whose header the walk had written and whose candidates it had
            // not (one diverging case in 12 000 of the differential fuzz with a look every third frame).  Waiting for an event that
            // has fired -- or was never recorded -- costs nothing.
            HIPCHK(hipStreamWaitEvent(st, h->ev_map, 0));
            hipLaunchKernelGGL(k_cells_update<false>, dim3(PF_CELLS_WALK_GRID), dim3(64), 0, st, kd_view(h), geo, ca.tab, (const int *)ca.list, ca.cs, ca.pool, ca.rec, h->cells_gen, (const int *)nullptr, 0);
            HIPCHK(hipEventRecord(h->ev_walked, st));
            h->walk_pending = true;
            h->cells_snap = true;
        }
        HIPCHK(hipGetLastError());
    }
    // The frame's critical chain is scan-match -> reduce -> best pose -> map update (rays, lists, traversal, insert, k_cells_update) -> the
    // next frame's scan-match.  pfslam_step runs ALL of it on the aux stream: the scan-match kernel only needs this stream's lane order
    // (an event that has long fired), and what this stream goes on with -- weights, sums, resample, the next dispersion and lane
    // order -- is shorter than the map update and off the chain.  With the scan-match kernel on this stream, the chain crossed
    // streams twice per frame (fork behind the best pose, join in front of the next scan-match), 12-17 us each.
    static const bool aux_ok = !(ab_env("PFSLAM_SCORE_ON_AUX") && atoi(ab_env("PFSLAM_SCORE_ON_AUX")) == 0);
    const bool on_aux = aux_ok && h->score_on_aux && use_cells && !cells_sync && fuse_minmax && !census && !sharded;
    h->scored_on_aux = on_aux;
    struct StreamSwap { // the rest of this function launches on h->stream: point it at the aux stream for that long
        pfslam_handle *h;
        hipStream_t keep;
        ~StreamSwap() { h->stream = keep; }
    } swap_guard{h, h->stream};
    if (on_aux) {
        HIPCHK(hipStreamWaitEvent(h->aux, h->ev_boxes, 0)); // recorded above, behind the pose boxes
        h->stream = h->aux;
    }
    CHK(join_map(h)); // from here on the scoring pass reads the map (lane order, pose boxes and cell marking above did not)
    // one wave per workgroup: a finished wave's slot is refilled at once instead of waiting for the slowest of four
    // (2.387 vs 2.400 ms with 256-thread groups)
    const dim3 grid64((h->n + 63) / 64, used), grid256((h->n + 255) / 256, used);
    // the scan-match kernel itself; cen != nullptr: its counting instantiation (same launch shape, lane order, plan and results)
    auto scan_match = [&](pf::KdCensus *cen) {
        if (use_cells) {
#define PF_CELLS_ARGS grid64, dim3(64), 0, h->stream, h->x, h->y, h->th, h->n, h->scan, (const pf::BeamParts *)h->beam_angle, h->nb, bpc, kd_view(h), geo, \
                      (const unsigned *)h->cell_tab, (const uint4 *)h->cell_pool, (const int *)h->cell_state, order, direct
            // a census replay BEHIND an accumulating pass must not add its (identical) sums a second time
            // no heading anywhere near the bound of the angle-addition sincos (the host's running bound, refreshed from every frame's
            // header): the instantiation without the direct form -- 60 instead of 79 VGPRs, 8 instead of 6 waves per SIMD
            const bool guard = !(h->theta_bound < 0.5f * PF_SUM_THETA_MAX) || h->own_global; // (a shard imports particles at every resample: always guarded)
            if (h->trig && cen) hipLaunchKernelGGL((k_score_kd_cells<true, PF_TRIG_DEVLIB>), PF_CELLS_ARGS, acc_out ? h->fit_acc : out, acc_out ? 2 : p16 ? 3 : 0, cen);
            else if (h->trig) hipLaunchKernelGGL((k_score_kd_cells<false, PF_TRIG_DEVLIB>), PF_CELLS_ARGS, acc_out ? h->fit_acc : out, acc_out ? 1 : p16 ? 3 : 0, (pf::KdCensus *)nullptr);
            else if (cen) hipLaunchKernelGGL((k_score_kd_cells<true, true>), PF_CELLS_ARGS, acc_out ? h->fit_acc : out, acc_out ? 2 : p16 ? 3 : 0, cen);
            else if (guard) hipLaunchKernelGGL((k_score_kd_cells<false, true>), PF_CELLS_ARGS, acc_out ? h->fit_acc : out, acc_out ? 1 : p16 ? 3 : 0, (pf::KdCensus *)nullptr);
            else hipLaunchKernelGGL((k_score_kd_cells<false, false>), PF_CELLS_ARGS, acc_out ? h->fit_acc : out, acc_out ? 1 : p16 ? 3 : 0, (pf::KdCensus *)nullptr);
#undef PF_CELLS_ARGS
        } else if (use_plan) {
            if (cen)
                hipLaunchKernelGGL((k_score_kd_plan<true>), grid64, dim3(64), 0, h->stream, h->x, h->y, h->th, h->n, h->scan, (const pf::BeamParts *)h->beam_angle, h->nb, bpc,
                                   kd_view(h), (const pf::KdPlanRow *)h->plan, order, direct, out, cen, h->trig);
            else
                hipLaunchKernelGGL((k_score_kd_plan<false>), grid64, dim3(64), 0, h->stream, h->x, h->y, h->th, h->n, h->scan, (const pf::BeamParts *)h->beam_angle, h->nb, bpc,
                                   kd_view(h), (const pf::KdPlanRow *)h->plan, order, direct, out, (pf::KdCensus *)nullptr, h->trig);
        } else if (h->planar && cen)
            hipLaunchKernelGGL((k_score_kd<true, true>), grid64, dim3(64), 0, h->stream, h->x, h->y, h->th, h->n, h->scan, (const pf::BeamParts *)h->beam_angle, h->nb, bpc,
                               kd_view(h), order, direct, out, cen, h->trig);
        else if (cen)
            hipLaunchKernelGGL((k_score_kd<false, true>), grid256, dim3(256), 0, h->stream, h->x, h->y, h->th, h->n, h->scan, (const pf::BeamParts *)h->beam_angle, h->nb, bpc,
                               kd_view(h), order, direct, out, cen, h->trig);
        else if (h->planar)
            hipLaunchKernelGGL((k_score_kd<true, false>), grid64, dim3(64), 0, h->stream, h->x, h->y, h->th, h->n, h->scan, (const pf::BeamParts *)h->beam_angle, h->nb, bpc,
                               kd_view(h), order, direct, out, (pf::KdCensus *)nullptr, h->trig);
        else
            hipLaunchKernelGGL((k_score_kd<false, false>), grid256, dim3(256), 0, h->stream, h->x, h->y, h->th, h->n, h->scan, (const pf::BeamParts *)h->beam_angle, h->nb, bpc,
                               kd_view(h), order, direct, out, (pf::KdCensus *)nullptr, h->trig);
    };
    if (use_plan || use_cells) {
        const int groups = (h->n + 63) / 64;
        if (use_cells) { // rows of the new cells (and a look at every record's links)
            if (cells_sync) h->cells_passes++;
            if (cells_sync)
                hipLaunchKernelGGL(k_cells_update<true>, dim3(PF_CELLS_GRID), dim3(64), 0, h->stream, kd_view(h), geo, ca.tab, (const int *)ca.list, ca.cs, ca.pool, ca.rec, h->cells_gen, (const int *)nullptr, 0, 0, 2);
            if (cells_sync) h->cells_snap = true; // (use_snap 2: the pass has left the snapshot)
        } else
            hipLaunchKernelGGL(k_plan, dim3((groups + 63) / 64, h->nb), dim3(64), 0, h->stream, (const pf::KdGroupBox *)h->group_box, groups,
                               (const float *)h->scan, h->nb, kd_view(h), h->plan);
        if (t_a) { // the planning launches are timed on their own: t_a .. t_p; t_p .. t_b brackets the scan-match kernel only
            hipEvent_t t_p = nullptr;
            CHK(timer_event(h, &t_p));
            HIPCHK(hipEventRecord(t_p, h->stream));
            h->ev_pending.push_back(pfslam_handle::TimedSpan{t_a, t_p, PF_T_PLAN, /*keep_b=*/true});
            t_a = t_p;
        }
    }
    scan_match(census);
    HIPCHK(hipGetLastError());
    if (t_a) {
        HIPCHK(hipEventRecord(t_b, h->stream));
        h->ev_pending.push_back(pfslam_handle::TimedSpan{t_a, t_b, PF_T_SCORE});
    }
    // census log (pfslam_set_census): the counting instantiation once more on the very same inputs, one record per scoring pass
    if (h->census_log && !census && h->census_n < PF_CENSUS_LOG) {
        scan_match(h->census_log + h->census_n++);
        HIPCHK(hipGetLastError());
    }
    if (fuse_minmax) {
        if (h->icp_forked) CHK(join_icp(h)); // the aux stream reset the keys
        if (!h->stats_clean) CHK(launch_stats_reset(h, h->stream));
        h->stats_clean = false;
    }
    if (wide_reduce) { // few particles, one beam per wave: 16 threads per particle
        hipLaunchKernelGGL(k_reduce_partials_minmax_wide, dim3((h->n + 63) / 64), dim3(1024), 0, h->stream, h->partial, h->n, used, order,
                           h->fit, (long long *)h->stats);
        HIPCHK(hipGetLastError());
    } else if (used > 1 && fuse_minmax) {
#define PF_REDUCE4_WGS 128 /* workgroups of k_reduce_partials_minmax (grid-stride over the slots; see the kernel) */
        static const int reduce4_wgs = ab_env("PFSLAM_REDUCE4_WGS") ? std::max(1, atoi(ab_env("PFSLAM_REDUCE4_WGS"))) : PF_REDUCE4_WGS; // (A/B: 1000000 = one workgroup per 256 slots)
        hipLaunchKernelGGL(k_reduce_partials_minmax, dim3(std::min((h->n + 255) / 256, reduce4_wgs)), dim3(256), 0, h->stream, acc_out ? h->fit_acc : h->partial, h->n,
                           acc_out ? 1 : used, order, h->fit, h->goff, (long long *)h->stats, h->x, h->y, h->th, acc_out ? 1 : 0, p16 ? 1 : 0);
        HIPCHK(hipGetLastError());
    } else {
        if (used > 1) {
            hipLaunchKernelGGL(k_reduce_partials, dim3((h->n + 255) / 256), dim3(256), 0, h->stream, h->partial, h->n,
                               used, order, h->fit);
            HIPCHK(hipGetLastError());
        }
        if (fuse_minmax) { // single chunk: the score kernel wrote fit directly
            const int blocks = std::min(1024, (h->n + 255) / 256);
            hipLaunchKernelGGL(k_minmax<float>, dim3(blocks), dim3(256), 0, h->stream, (const float *)h->fit, h->n, h->goff, (long long *)h->stats);
            HIPCHK(hipGetLastError());
        }
    }
    return 0;
}

// What one launch of the score kernel issues on the handle's current particles, scan and map: the counting instantiation
// of the same kernel (same launch shape and lane order) -> out[0] wave-level trips of the descent loop (= wave-level 16-byte
// gathers of node records), out[1] active lanes in them (= node visits), out[2] wave-level parent-hyperplane tests (each one
// 4-byte and one 16-byte wave gather), out[3] lanes in them, out[4] trips in which every active lane stood on the same node,
// out[5] those of them on the common path of all 64 lanes from the root.  fit[] is recomputed (identical values).
extern "C" int pfslam_score_census(pfslam_handle *h, unsigned long long out[8])
{
    if (!h || !out) return fail("pfslam_score_census: bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle_staged(h));
    if (!h->d_census) CHK(dalloc(&h->d_census, 1));
    HIPCHK(hipMemsetAsync(h->d_census, 0, sizeof(pf::KdCensus), h->stream));
    CHK(launch_score(h, false, h->d_census));
    pf::KdCensus c;
    HIPCHK(hipMemcpyAsync(&c, h->d_census, sizeof(c), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    out[0] = c.trips; out[1] = c.lanes; out[2] = c.tests; out[3] = c.test_lanes; out[4] = c.uniform; out[5] = c.prefix; out[6] = c.redesc; out[7] = c.redesc_noop;
    return 0;
}

// Census log: while enabled, every scoring pass (pfslam_step, pfslam_shard_score, pfslam_score_kd) is followed by the counting
// instantiation of the scan-match kernel on the very same inputs (particles, scan, map, lane order, plan); one record per pass.
// This is how bench.py counts what its TIMED launches issue: a second handle replays the same frames with the log on.
extern "C" int pfslam_set_census(pfslam_handle *h, int enable)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (enable) {
        if (!h->census_store) CHK(dalloc(&h->census_store, (size_t)PF_CENSUS_LOG));
        HIPCHK(hipMemsetAsync(h->census_store, 0, sizeof(pf::KdCensus) * PF_CENSUS_LOG, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        h->census_n = 0;
    }
    h->census_log = enable ? h->census_store : nullptr;
    return 0;
}
// out[k][0..7] = {trips, lanes, tests, test_lanes, uniform, prefix, redescents, redescents_noop} of the k-th logged pass
extern "C" int pfslam_get_census_log(pfslam_handle *h, unsigned long long *out, int cap, int *n)
{
    if (!h || !n || (cap > 0 && !out)) return fail("pfslam_get_census_log: bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    *n = h->census_n;
    const int m = std::min(h->census_n, cap);
    if (m > 0 && h->census_store) {
        static_assert(sizeof(pf::KdCensus) == 64, "eight counters per record");
        HIPCHK(hipMemcpyAsync(out, h->census_store, (size_t)m * sizeof(pf::KdCensus), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

extern "C" int pfslam_score_kd(pfslam_handle *h, float *fit_host)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle_staged(h));
    CHK(launch_score(h));
    if (fit_host) {
        HIPCHK(hipMemcpyAsync(fit_host, h->fit, (size_t)h->n * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
    }
    return 0;
}

extern "C" int pfslam_time_score_kd(pfslam_handle *h, int iters, float *ms_per_launch)
{
    if (!h || iters <= 0 || !ms_per_launch) return fail("pfslam_time_score_kd: bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle_staged(h));
    CHK(launch_score(h)); // warm
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    for (int k = 0; k < iters; k++) CHK(launch_score(h));
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    HIPCHK(hipEventSynchronize(h->ev1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    *ms_per_launch = ms / iters;
    return 0;
}

// ---- measurement support: what the last scoring pass's shared-prefix plan looks like ---------------------------------
__global__ __launch_bounds__(256) void k_plan_stats(const pf::KdPlanRow *__restrict__ plan, size_t rows, const pf::KdGroupBox *__restrict__ box,
                                                    int groups, double *__restrict__ out)
{
    double path = 0, cand = 0, done = 0, none = 0, full = 0, dx = 0, dy = 0, dt = 0;
    for (size_t r = (size_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (size_t)gridDim.x * 256) {
        const pf::KdPlanRow *p = plan + r;
        path += p->path_len;
        cand += p->n_cand;
        done += p->resume < 0;
        none += p->path_len == 0;
        full += p->n_cand == PF_PLAN_CAND;
    }
    for (int g = blockIdx.x * 256 + threadIdx.x; g < groups; g += gridDim.x * 256) {
        dx += box[g].xhi - box[g].xlo;
        dy += box[g].yhi - box[g].ylo;
        dt += box[g].thi - box[g].tlo;
    }
    double v[8] = {path, cand, done, none, full, dx, dy, dt};
    for (int k = 0; k < 8; k++) {
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&out[k], v[k]);
    }
}
// out[0] rows (waves x beams), [1] mean length of the common root path, [2] mean candidates kept of it, [3] fraction of rows whose
// first descent is complete (no per-lane tail), [4] fraction without a plan (root straddled / unusable), [5] fraction with a full
// candidate list, [6..8] mean extent of a wave's pose box in x, y (m) and heading (rad), [9] waves
extern "C" int pfslam_plan_stats(pfslam_handle *h, double out[10])
{
    if (!h || !out) return fail("pfslam_plan_stats: bad argument");
    for (int k = 0; k < 10; k++) out[k] = 0.0;
    if (!h->plan || !h->plan_valid) return 0;
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    const int groups = (h->n + 63) / 64;
    const size_t rows = (size_t)groups * h->nb;
    double *d = nullptr;
    CHK(dalloc(&d, 8));
    HIPCHK(hipMemsetAsync(d, 0, 64, h->stream));
    hipLaunchKernelGGL(k_plan_stats, dim3(1024), dim3(256), 0, h->stream, (const pf::KdPlanRow *)h->plan, rows, (const pf::KdGroupBox *)h->group_box, groups, d);
    HIPCHK(hipGetLastError());
    double v[8];
    HIPCHK(hipMemcpyAsync(v, d, 64, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipFree(d));
    out[0] = (double)rows;
    for (int k = 0; k < 5; k++) out[1 + k] = v[k] / (double)rows;
    for (int k = 0; k < 3; k++) out[6 + k] = v[5 + k] / groups;
    out[9] = groups;
    return 0;
}

// the persistent lattice-cell rows: out[0] cells claimed since the last wipe (one record each), [1] live rows (one per sub-cell: up to four
// per cell), [2] mean first-descent candidates per row, [3] mean re-descent candidates per row, [4] sub-cells without a row (too many
// candidates / pool exhausted: generic lanes), [5] 16-byte pool slots used, [6] [7] lattice index of the window's corner cell,
// [8 .. 11] since the last wipe: cells walked from the root / extensions (a link of the cell had gained a node) / looks that found a
// cell unchanged (reused as it was) / cells claimed by the marking passes, [12] device flags (PF_CF_*: 1 list full, 2 pool full,
// 8 cloud far from the window centre), [13] publishing updates since the last wipe (the divisor of [9] [10]), [14] wipes so far,
// [15] 1 = suspended (list / pool overflowed twice in a row: the round-2 plan scores until the map is replaced or re-balanced).
// [0 .. 13] are zero when the last scoring pass did not use cell rows.
extern "C" int pfslam_cell_stats(pfslam_handle *h, double out[16])
{
    if (!h || !out) return fail("pfslam_cell_stats: bad argument");
    for (int k = 0; k < 16; k++) out[k] = 0.0;
    out[14] = (double)h->cells_wipes;
    out[15] = h->cells_suspended ? 1.0 : 0.0;
    if (!h->cell_state || !h->cells_valid) return 0;
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    int cs[PF_CS_ALLOC];
    HIPCHK(hipMemcpyAsync(cs, h->cell_state, sizeof(cs), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int rows = cs[PF_CS_ROWS];
    out[0] = std::min(cs[PF_CS_COUNT], cs[PF_CS_LIST_CAP]);
    out[1] = rows;
    out[2] = rows ? (double)cs[PF_CS_N1] / rows : 0.0;
    out[3] = rows ? (double)cs[PF_CS_N2] / rows : 0.0;
    out[4] = cs[PF_CS_NONE];
    out[5] = cs[PF_CS_POOL];
    out[6] = cs[PF_CS_OX];
    out[7] = cs[PF_CS_OY];
    for (int k = 0; k < 4; k++) out[8 + k] = cs[PF_CS_CUR_FRESH + k];
    out[12] = cs[PF_CS_FLAGS];
    out[13] = (double)h->cells_passes;
    out[14] = (double)h->cells_wipes;
    out[15] = h->cells_suspended ? 1.0 : 0.0;
#ifdef PF_CELLS_PROFILE
    {
        int prof[64];
        HIPCHK(hipMemcpy(prof, h->cell_state + 64, sizeof(prof), hipMemcpyDeviceToHost));
        int waves = 0;
        HIPCHK(hipMemcpy(&waves, h->cell_state + 63, 4, hipMemcpyDeviceToHost));
        int ext[2];
        HIPCHK(hipMemcpy(ext, h->cell_state + 60, 8, hipMemcpyDeviceToHost));
        fprintf(stderr, "extensions %d, of them without a new candidate or re-descent candidate %d\n", ext[0], ext[1]);
        fprintf(stderr, "k_cells_update<true> phases over %d wave-passes (mean / max us):", waves);
        for (int p = 0; p < 9; p++) fprintf(stderr, "  [%d] %.2f / %.2f", p, waves ? prof[2 * p] * 0.01 / waves : 0.0, prof[2 * p + 1] * 0.01);
        fprintf(stderr, "\n  slowest wave of the last 16 passes (us) / waves above 20 us:");
        int wmax[32];
        HIPCHK(hipMemcpy(wmax, h->cell_state + 96, sizeof(wmax), hipMemcpyDeviceToHost));
        for (int p = 0; p < 16; p++) fprintf(stderr, " %.1f/%d", wmax[p] * 0.01, wmax[16 + p]);
        fprintf(stderr, "\n");
    }
#endif
    return 0;
}

// ---- the cell rows' invariants, checked on the device (tests; PFSLAM_CHECK_CELLS=1 makes pfslam_synchronize run it) -------------------
// An oracle comparison sees wrong SCORES; it cannot see rows that are right today and that nobody watches (stale at the next insert
// there), records read half-written, or a table word nobody accounts for -- the round-4 races were of that kind.  What must hold
// whenever no frame is in flight:
//   * a record of the current generation that is published (neither FRESH nor DEAD) owns its four table words: each is FALLBACK or a row
//     whose slots lie inside the record's pool allocation and list, in order, nodes the record holds as candidates;
//   * none of its watched links -- the first descent's, every candidate's re-descent walk's -- has a child in the tree (a link that
//     gained a node and was not extended is a stale row waiting to happen);
//   * a claimed cell that is not published yet (never walked, or walked and FRESH) still has its claim word PENDING;
//   * the table holds exactly the rows / fallback words the counters say, and one PENDING word per unpublished record.
// out: [0] records  [1] never walked  [2] fresh  [3] published  [4] row words in the table  [5] pending words  [6] fallback words
//      [7] dead records  [8..15] violations: unwalked-not-pending, fresh-not-pending, row outside its allocation (or a table word that is not the one
//      the record published last), row slot not a candidate
//      (or out of order), watched link has a child, malformed link word, dead record with a row, counter mismatch (host side)
__global__ __launch_bounds__(256) void k_cells_check(pf::KdView tree, const unsigned *__restrict__ tab, const int *__restrict__ list, const int *__restrict__ cs,
                                                     const uint4 *__restrict__ pool, const int *__restrict__ rec_base, int gen, unsigned long long *__restrict__ out)
{
    const int count = min(cs[PF_CS_COUNT], cs[PF_CS_LIST_CAP]);
    unsigned long long c[16] = {0};
    for (int r = blockIdx.x * 256 + threadIdx.x; r < count; r += gridDim.x * 256) {
        const int *rec = rec_base + (size_t)r * PF_REC_WORDS;
        const int cell = list[r];
        c[0]++;
        const unsigned w0 = tab[cell];
        if (rec[3] != gen) { c[1]++; if (w0 != PF_CELL_PENDING) c[8]++; continue; }
        const int m = rec[2] & 0xff, fl = rec[2] >> 16;
        if (fl & PF_RF_FRESH) { c[2]++; if (w0 != PF_CELL_PENDING) c[9]++; continue; }
        const unsigned wd[4] = {w0, tab[cell + 1], tab[cell + PF_CELL_WIN], tab[cell + PF_CELL_WIN + 1]};
        if (fl & PF_RF_DEAD) {
            c[7]++;
            for (int b = 0; b < 4; b++) if (wd[b] != PF_CELL_FALLBACK || (unsigned)rec[PF_REC_LAST + b] != wd[b]) c[14]++;
            continue;
        }
        c[3]++;
        const int p0 = rec[PF_REC_POOL], plen = rec[PF_REC_POOL + 1];
        for (int b = 0; b < 4; b++) {
            if ((unsigned)rec[PF_REC_LAST + b] != wd[b]) { c[10]++; continue; } // the table word is not what the record says it published last (its accounts would close wrongly)
            if (wd[b] == PF_CELL_FALLBACK) continue;
            if (wd[b] == 0u || wd[b] >= 0x40000000u) { c[9]++; continue; }
            const int n1 = (int)(wd[b] & 15u), n2 = (int)((wd[b] >> 4) & 15u), off = (int)(wd[b] >> 8);
            if (off < p0 || off + n1 + (n2 == 15 ? 0 : n2) > p0 + plen || n1 < 1) { c[10]++; continue; }
            int pos = 0;
            for (int k = 0; k < n1; k++) { // the row's candidates are a sub-sequence of the record's
                const int node = (int)(pool[off + k].z & 0x3fffffffu);
                while (pos < m && rec[PF_REC_CAND + pos] != node) pos++;
                if (pos == m) { c[11]++; break; }
                pos++;
            }
        }
        if (rec[1] < 0 || watched_child(tree, rec[1]) >= 0) c[12]++;
        if (!(fl & PF_RF_ROVER))
            for (int k = 0; k < m; k++) {
                const int ct = rec[PF_REC_TERM + k];
                if (ct < -1) c[13]++;
                else if (ct >= 0 && watched_child(tree, ct) >= 0) c[12]++;
            }
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)PF_CELL_WIN * PF_CELL_WIN; i += (size_t)gridDim.x * 256) {
        const unsigned w = tab[i];
        if (w == 0u) continue;
        if (w == PF_CELL_PENDING) c[5]++;
        else if (w == PF_CELL_FALLBACK) c[6]++;
        else c[4]++;
    }
    for (int k = 0; k < 15; k++) {
        unsigned long long v = c[k];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v) atomicAdd(&out[k], v);
    }
}
extern "C" int pfslam_debug_check_cells(pfslam_handle *h, long long out[16])
{
    if (!h || !out) return fail("pfslam_debug_check_cells: bad argument");
    for (int k = 0; k < 16; k++) out[k] = 0;
    if (!h->cell_tab || !h->cells_valid || h->cells_wipe_pending) return 0;
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    unsigned long long *d = nullptr;
    CHK(dalloc(&d, 16));
    HIPCHK(hipMemsetAsync(d, 0, 128, h->stream));
    hipLaunchKernelGGL(k_cells_check, dim3(2048), dim3(256), 0, h->stream, kd_view(h), (const unsigned *)h->cell_tab, (const int *)h->cell_list,
                       (const int *)h->cell_state, (const uint4 *)h->cell_pool, (const int *)h->cell_rec, h->cells_gen, d);
    HIPCHK(hipGetLastError());
    int cs[PF_CS_WORDS];
    HIPCHK(hipMemcpyAsync(out, d, 128, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(cs, h->cell_state, sizeof(cs), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipFree(d));
    // the counters against the table itself (exact while neither the list nor the pool has overflowed)
    if (!(cs[PF_CS_FLAGS] & (PF_CF_LIST_FULL | PF_CF_POOL_FULL)))
        out[15] = (out[4] != cs[PF_CS_ROWS]) + (out[6] != cs[PF_CS_NONE]) + (out[5] != out[1] + out[2]);
    return 0;
}

// ---- frame probe: where a round-5 frame's time goes, from the frame's own kernels ---------------------------------------------------
// frames > 0: keep the stamps of the last `frames` tickets (the first thread of each launch of a round-5 frame stores the 100 MHz wall
// clock); 0: off.  pfslam_get_probe copies them out: out[f][PF_PROBE_SLOTS] for tickets last - n + 1 .. last (0 = launch did not run).
extern "C" int pfslam_set_probe(pfslam_handle *h, int frames)
{
    if (!h || frames < 0 || frames > 4096) return fail("pfslam_set_probe: 0 .. 4096 frames");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (h->probe) HIPCHK(hipFree(h->probe));
    h->probe = nullptr;
    h->probe_frames = 0;
    if (frames > 0) {
        CHK(dalloc(&h->probe, (size_t)frames * PF_PROBE_SLOTS));
        HIPCHK(hipMemset(h->probe, 0, (size_t)frames * PF_PROBE_SLOTS * 8));
        h->probe_frames = frames;
    }
    return 0;
}
extern "C" int pfslam_get_probe(pfslam_handle *h, unsigned long long *out, int cap_frames, int *n_frames, int *last_ticket)
{
    if (!h || !n_frames || (cap_frames > 0 && !out)) return fail("pfslam_get_probe: bad argument");
    *n_frames = 0;
    if (last_ticket) *last_ticket = h->seq;
    if (!h->probe) return 0;
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int n = std::min(std::min(cap_frames, h->probe_frames), h->seq);
    std::vector<unsigned long long> all((size_t)h->probe_frames * PF_PROBE_SLOTS);
    HIPCHK(hipMemcpy(all.data(), h->probe, all.size() * 8, hipMemcpyDeviceToHost));
    for (int k = 0; k < n; k++) {
        const int ticket = h->seq - n + 1 + k;
        memcpy(out + (size_t)k * PF_PROBE_SLOTS, &all[(size_t)(ticket % h->probe_frames) * PF_PROBE_SLOTS], PF_PROBE_SLOTS * 8);
    }
    *n_frames = n;
    return 0;
}
// how the handle's round-5 frames run: out[0] 1 = the last frame was one, [1] 1 = cross-stream edges are gates (0: events -- asked for, or the
// start-up self-test found two streams on one hardware queue), [2] one-stream mode, [3] publication lag in frames
extern "C" int pfslam_frame_mode(pfslam_handle *h, int out[4])
{
    if (!h || !out) return fail("pfslam_frame_mode: bad argument");
    out[0] = h->pipe_live ? 1 : 0;
    out[1] = h->pipe_live ? (h->gates_live ? 1 : 0) : h->gates; // (frames in flight: what they use -- gates are withdrawn while the process holds a second handle)
    out[2] = h->serial;
    out[3] = h->publish_lag;
    return 0;
}
extern "C" const char *pfslam_probe_name(int slot) { return slot >= 0 && slot < PB_END ? pb_names[slot] : ""; }
// 0 (default): the transcendentals of pf_math.h (fixed sequences of double operations, what the CPU oracle follows bit for bit).  1: the
// device library's cosf / sinf in CleanLidarScan and erfcinvf in the dispersion -- what the reference's text compiles to on this platform;
// with it the product's kernels equal the reference's own kernels built for gfx950 with zero mismatches (tests/test_gpu_ref_kernels.py).
extern "C" int pfslam_set_trig(pfslam_handle *h, int devlib)
{
    if (!h || devlib < 0 || devlib > 1) return fail("pfslam_set_trig: 0 (specification) or 1 (device library)");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    h->trig = devlib;
    return 0;
}
// 1: every frame's launches on one stream, in enqueue order (what PFSLAM_SERIAL=1 sets at creation); same results, same bookkeeping
extern "C" int pfslam_set_serial(pfslam_handle *h, int serial)
{
    if (!h) return fail("null handle");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->serial = serial != 0;
    h->pipe_live = false;
    return 0;
}

// ---- measurement support: the chip's wave-gather rate, measured live (the roofline the score kernel is priced against) ----
// Dependent-free wave-level 16-byte gathers from a 2 MB table (cache resident, like the hot map records), 8 waves per SIMD, 8
// gathers in flight per lane; every lane of a wave reads the same pseudo-random record (the cheapest case for the L1: what is
// measured is the address / data path of the texture addresser, 4 lanes per clock for 64-bit and wider loads).
__global__ __launch_bounds__(256) void k_ubench_gather(const unsigned *__restrict__ table, int n_rec, int iters, unsigned *__restrict__ out)
{
    const pf::kd_rsrc_t r = pf::kd_rsrc(table);
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    unsigned state = (gid >> 6) * 2654435761u + 12345u, acc = 0;
    for (int it = 0; it < iters; it++) {
        int idx[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            state = state * 1664525u + 1013904223u;
            idx[k] = (int)((state >> 8) & (unsigned)(n_rec - 1));
        }
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint4 v = pf::kd_load_hot(r, idx[k]);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    out[gid] = acc;
}
// out[0] = wave-level 16-byte gathers per second over the whole chip, out[1] = compute units, out[2] = nominal clock (GHz),
// out[3] = cycles per wave gather per CU at the nominal clock
extern "C" int pfslam_ubench_gather(pfslam_handle *h, double out[4])
{
    if (!h || !out) return fail("pfslam_ubench_gather: bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, h->cfg.device));
    const int cus = prop.multiProcessorCount, n_rec = 131072, iters = 400;
    const int blocks = cus * 8 * 4; // 8 blocks of 4 waves per CU = 8 waves per SIMD, 4 rounds
    unsigned *table = nullptr, *sink = nullptr;
    CHK(dalloc(&table, (size_t)n_rec * 4));
    CHK(dalloc(&sink, (size_t)blocks * 256));
    std::vector<unsigned> init((size_t)n_rec * 4);
    for (size_t i = 0; i < init.size(); i++) init[i] = (unsigned)(i * 2654435761u);
    HIPCHK(hipMemcpyAsync(table, init.data(), init.size() * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_ubench_gather, dim3(blocks), dim3(256), 0, h->stream, table, n_rec, 10, sink);
    float best = 0.0f;
    for (int rep = 0; rep < 3; rep++) { // best of three
        HIPCHK(hipEventRecord(h->ev0, h->stream));
        hipLaunchKernelGGL(k_ubench_gather, dim3(blocks), dim3(256), 0, h->stream, table, n_rec, iters, sink);
        HIPCHK(hipEventRecord(h->ev1, h->stream));
        HIPCHK(hipEventSynchronize(h->ev1));
        float ms = 0.0f;
        HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
        if (rep == 0 || ms < best) best = ms;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipFree(table));
    HIPCHK(hipFree(sink));
    const double wave_gathers = (double)blocks * 4 * iters * 8;
    const double ghz = prop.clockRate * 1e-6;
    out[0] = wave_gathers / (best * 1e-3);
    out[1] = cus;
    out[2] = ghz;
    out[3] = best * 1e-3 * ghz * 1e9 * cus / wave_gathers;
    return 0;
}

extern "C" int pfslam_traverse(pfslam_handle *h, const float *xyz_host, int n, int32_t *best_host)
{
    if (!h || !xyz_host || !best_host || n < 0) return fail("pfslam_traverse: bad argument");
    if (h->kd_size <= 0) return fail("pfslam_traverse: no map loaded");
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    float *d_xyz = nullptr;
    int *d_best = nullptr;
    CHK(dalloc(&d_xyz, (size_t)n * 3));
    CHK(dalloc(&d_best, (size_t)n));
    HIPCHK(hipMemcpyAsync(d_xyz, xyz_host, (size_t)n * 12, hipMemcpyHostToDevice, h->stream));
    bool planar = h->planar;
    if (planar)
        for (int i = 0; i < n; i++)
            if (xyz_host[3 * i + 2] != 0.0f) { planar = false; break; }
    if (planar)
        hipLaunchKernelGGL(k_traverse<true>, dim3((n + 255) / 256), dim3(256), 0, h->stream, d_xyz, n, kd_view(h), d_best);
    else
        hipLaunchKernelGGL(k_traverse<false>, dim3((n + 255) / 256), dim3(256), 0, h->stream, d_xyz, n, kd_view(h), d_best);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(best_host, d_best, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipFree(d_xyz));
    HIPCHK(hipFree(d_best));
    return 0;
}

extern "C" int pfslam_debug_math(pfslam_handle *h, int which, const float *in_host, int n, float *out_host)
{
    if (!h || !in_host || !out_host || n <= 0) return fail("pfslam_debug_math: bad argument");
    HIPCHK(hipSetDevice(h->cfg.device));
    CHK(settle(h));
    const int per = which == 0 || which == 6 || which == 7 ? 2 : 1;
    float *d_in = nullptr, *d_out = nullptr;
    CHK(dalloc(&d_in, (size_t)n));
    CHK(dalloc(&d_out, (size_t)n * per));
    HIPCHK(hipMemcpyAsync(d_in, in_host, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, h->stream, which, d_in, n, d_out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_host, d_out, (size_t)n * per * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipFree(d_in));
    HIPCHK(hipFree(d_out));
    return 0;
}

extern "C" int pfslam_device_ptr(pfslam_handle *h, int which, void **ptr, size_t *bytes)
{
    if (!h || !ptr || !bytes) return fail("pfslam_device_ptr: bad argument");
    // no settle: the buffers are fixed for the handle's lifetime, except the pose blocks 2-4 / 16 (and 17 on an unsharded handle),
    // which alternate between two allocations with every enqueued frame -- the pointer returned is the one the NEXT enqueued
    // kernel will use, which is what a caller placing collectives between the pfslam_shard_* calls needs
    const size_t n = h->n;
    switch (which) {
    case 0: *ptr = h->stats; *bytes = 64; break;
    case 1: *ptr = h->fit; *bytes = n * 4; break;
    case 2: *ptr = h->x; *bytes = n * 4; break;
    case 3: *ptr = h->y; *bytes = n * 4; break;
    case 4: *ptr = h->th; *bytes = n * 4; break;
    case 5: *ptr = h->w; *bytes = (size_t)h->stride * 4; break; // padded to the shard stride (equal all-gather counts)
    case 6: *ptr = h->tile_r; *bytes = ((n + PF_SUM_TILE - 1) / PF_SUM_TILE) * 4; break;
    case 7: *ptr = h->scan; *bytes = (size_t)h->nb * 4; break;
    case 8: *ptr = h->start; *bytes = 16; break;
    case 9: *ptr = h->pose; *bytes = 16; break;
    case 10: *ptr = h->gw; *bytes = (size_t)h->world * h->stride * 4; break;
    // this rank's record of the frame being enqueued = its packed {max key, negated-min key}, straight from the reduce: query it after
    // pfslam_shard_score of the same frame (round-5 frames keep two of them, by ticket parity); 15 = every rank's record
    case 14: *ptr = (h->shard_v2 && h->fstats) ? (void *)(h->fstats + 4 * (h->cur_seq & 1)) : (void *)h->stats; *bytes = 16; break;
    case 15: *ptr = h->gkeys; *bytes = (size_t)16 * h->world; break;
    case 16: *ptr = h->pblk; *bytes = (size_t)3 * h->stride * 4; break;
    case 17: *ptr = h->gpose; *bytes = (size_t)h->world * 3 * h->stride * 4; break;
    // the map as the device holds it (pfslam_shard_balance_*: broadcast from the rank that re-balanced): kd_capacity entries each
    case 20: *ptr = h->hot; *bytes = (size_t)h->kd_cap * 16; break;
    case 21: *ptr = h->parent; *bytes = (size_t)h->kd_cap * 4; break;
    case 22: *ptr = h->kz; *bytes = (size_t)h->kd_cap * 4; break;
    case 23: *ptr = h->kw; *bytes = (size_t)h->kd_cap * 4; break;
    case 24: *ptr = h->kd_state; *bytes = 16; break;
    default: return fail("pfslam_device_ptr: unknown buffer");
    }
    return 0;
}

static bool frame_v2_ok(pfslam_handle *h, int *chunks_out, int *bpc_out);                  // pfslam_frame.hip.inc
static int frame_v2(pfslam_handle *h, int frame, const float *scan_host, int used, int bpc); // (the round-5 frame of pfslam_step)
static int frame_v2_begin(pfslam_handle *h, int frame, const float *scan_host, int used, int bpc, bool sharded); // ... in four parts: the sharded frame's cuts
static int frame_v2_score(pfslam_handle *h);
static int frame_v2_chain(pfslam_handle *h);
static int frame_v2_finish(pfslam_handle *h);
#include "pfslam_stages.hip.inc"
#include "pfslam_frame.hip.inc"
