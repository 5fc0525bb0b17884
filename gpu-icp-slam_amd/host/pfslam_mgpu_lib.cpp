// pfslam_mgpu_lib.cpp -- libpfslam_mgpu.so: the sharded frame of include/pfslam.h with its three all-gathers on librccl, launched
// straight into the frame's own streams (include/pfslam_mgpu.h).  host/pfslam_mgpu.cpp (the C++ driver) and bench.py --gpus N (through
// ctypes) both step their rank through pfslam_mgpu_step: one frame loop, whoever drives it.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/pfslam_mgpu.h"

static thread_local std::string g_merr;
static int mfail(const std::string &m)
{
    g_merr = m;
    return 1;
}
extern "C" const char *pfslam_mgpu_last_error(void) { return g_merr.c_str(); }

#define PF(call)                                                                  \
    do {                                                                          \
        if (call) return mfail(std::string(#call ": ") + pfslam_last_error());    \
    } while (0)
#define HIP(call)                                                                 \
    do {                                                                          \
        hipError_t e__ = (call);                                                  \
        if (e__ != hipSuccess) return mfail(std::string(#call ": ") + hipGetErrorString(e__)); \
    } while (0)
#define NCCL(call)                                                                \
    do {                                                                          \
        ncclResult_t r__ = (call);                                                \
        if (r__ != ncclSuccess) return mfail(std::string(#call ": ") + ncclGetErrorString(r__)); \
    } while (0)

struct pfslam_mgpu {
    pfslam_handle *h = nullptr;
    // one communicator per stream the frame's collectives go into: RCCL orders a communicator's operations, and would have to do so
    // ACROSS streams (events) if the keys' all-gather on the chain stream shared one with the particle stream's two
    ncclComm_t comm_p = nullptr, comm_c = nullptr;
    int world = 1, rank = 0, stride = 0;
    int device = 0; // the handle's GPU: ncclCommInitRank binds a communicator to the CURRENT device, so it is made current first
    int collectives = 0, balance_builds = 0, balance_broadcasts = 0;
    double *scratch = nullptr;
};

static_assert(sizeof(ncclUniqueId) * 2 == PFSLAM_MGPU_ID_BYTES, "two ncclUniqueId");

extern "C" int pfslam_mgpu_make_id(unsigned char id[PFSLAM_MGPU_ID_BYTES])
{
    if (!id) return mfail("pfslam_mgpu_make_id: null argument");
    ncclUniqueId a, b;
    NCCL(ncclGetUniqueId(&a));
    NCCL(ncclGetUniqueId(&b));
    memcpy(id, &a, sizeof(a));
    memcpy(id + sizeof(a), &b, sizeof(b));
    return 0;
}

static int buf(pfslam_handle *h, int which, void **p)
{
    size_t bytes = 0;
    PF(pfslam_device_ptr(h, which, p, &bytes));
    return 0;
}
static int stream_of(pfslam_handle *h, int which, hipStream_t *st)
{
    void *s = nullptr;
    PF(pfslam_shard_stream(h, which, &s));
    *st = (hipStream_t)s;
    return 0;
}

// RCCL connects a communicator's peers at its first collective, and sets up each protocol at the first message of its size class:
// hundreds of milliseconds, seconds on a cold node.  Inside a frame that wait would sit in front of stream gates that give up after a
// second (pfslam_frame.hip.inc) -- so everything the frame loop issues runs once here, at the frame's sizes, on a scratch buffer,
// before the first frame: the three all-gathers, the re-balance's broadcast, the barrier's all-reduce.
static int warm_up_on(pfslam_mgpu *m, char *tmp, size_t big, hipStream_t st)
{
    char *dst = tmp, *src = tmp + big * (size_t)m->world;
    NCCL(ncclAllGather(src, dst, big, ncclChar, m->comm_p, st));                        // pose blocks
    NCCL(ncclAllGather(src, dst, (size_t)m->stride * 4, ncclChar, m->comm_p, st));      // weights
    NCCL(ncclAllGather(src, dst, 16, ncclChar, m->comm_c, st));                         // a shard's packed keys
    NCCL(ncclBroadcast(src, src, big, ncclChar, 0, m->comm_p, st));                     // KDTree::Balance
    NCCL(ncclBroadcast(src, src, 16, ncclChar, 0, m->comm_p, st));
    NCCL(ncclAllReduce(m->scratch, m->scratch, 1, ncclDouble, ncclMax, m->comm_p, st)); // pfslam_mgpu_barrier_max
    HIP(hipStreamSynchronize(st));
    return 0;
}
static int warm_up(pfslam_mgpu *m)
{
    const size_t big = (size_t)3 * m->stride * 4;
    char *tmp = nullptr;
    hipStream_t st = nullptr;
    if (stream_of(m->h, 0, &st)) return 1;
    HIP(hipMalloc((void **)&tmp, big * (size_t)(m->world + 1)));
    int rc = 0;
    if (hipMemsetAsync(tmp, 0, big * (size_t)(m->world + 1), st) != hipSuccess || hipMemsetAsync(m->scratch, 0, 8, st) != hipSuccess)
        rc = mfail("hipMemsetAsync failed");
    if (!rc) rc = warm_up_on(m, tmp, big, st);
    (void)hipFree(tmp);
    return rc;
}

extern "C" int pfslam_mgpu_create(const unsigned char *id, int world, int rank, pfslam_handle *h, pfslam_mgpu **out)
{
    if (!h || !out || world < 1 || rank < 0 || rank >= world) return mfail("pfslam_mgpu_create: bad argument");
    if (world > 1 && !id) return mfail("pfslam_mgpu_create: a job of more than one rank needs the id rank 0 made (pfslam_mgpu_make_id)");
    pfslam_mgpu *m = new pfslam_mgpu();
    m->h = h;
    m->world = world;
    m->rank = rank;
    {
        void *p = nullptr;
        size_t bytes = 0;
        if (pfslam_device_ptr(h, 5, &p, &bytes)) { // (buffer 5 = the weights, padded to the shard stride)
            delete m;
            return mfail(std::string("pfslam_device_ptr: ") + pfslam_last_error());
        }
        m->stride = (int)(bytes / 4);
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, p) != hipSuccess || hipSetDevice(attr.device) != hipSuccess) {
            delete m;
            return mfail("pfslam_mgpu_create: cannot make the handle's device current");
        }
        m->device = attr.device;
    }
    if (world > 1) {
        ncclUniqueId a, b;
        memcpy(&a, id, sizeof(a));
        memcpy(&b, id + sizeof(a), sizeof(b));
        ncclResult_t r = ncclCommInitRank(&m->comm_p, world, a, rank);
        if (r == ncclSuccess) r = ncclCommInitRank(&m->comm_c, world, b, rank);
        if (r != ncclSuccess) {
            const std::string msg = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
            if (m->comm_p) (void)ncclCommDestroy(m->comm_p);
            delete m;
            return mfail(msg);
        }
        if (pfslam_set_shard_balance(h, 1)) { // ONE KDTree::Balance per node: rank 0 builds, the others adopt the broadcast arrays
            delete m;
            return mfail(std::string("pfslam_set_shard_balance: ") + pfslam_last_error());
        }
    }
    if (hipMalloc((void **)&m->scratch, 8) != hipSuccess) {
        delete m;
        return mfail("hipMalloc failed");
    }
    if (world > 1) {
        const int wrc = warm_up(m);
        if (wrc) {
            (void)pfslam_mgpu_destroy(m);
            return wrc;
        }
    }
    *out = m;
    return 0;
}

extern "C" int pfslam_mgpu_destroy(pfslam_mgpu *m)
{
    if (!m) return 0;
    if (m->scratch) (void)hipFree(m->scratch);
    if (m->comm_p) (void)ncclCommDestroy(m->comm_p);
    if (m->comm_c) (void)ncclCommDestroy(m->comm_c);
    delete m;
    return 0;
}

// the sharded frame of include/pfslam.h: three all-gathers on a fixed schedule, NO host wait (the frame is booked one step
// later from its pinned header, like pfslam_step's)
extern "C" int pfslam_mgpu_step(pfslam_mgpu *m, int frame, const float *scan)
{
    if (!m || !scan) return mfail("pfslam_mgpu_step: bad argument");
    HIP(hipSetDevice(m->device));
    pfslam_handle *h = m->h;
    const bool comm = m->world > 1; // world 1: buffers 10 / 17 alias 5 / 16, nothing reads buffer 15, nothing to move
    if (comm) { // KDTree::Balance (frame % 100 == 5) ONCE per node: rank 0 builds, the others take its device arrays (28 B per node)
        int due = 0, n_nodes = 0;
        PF(pfslam_shard_balance_due(h, frame, &due, &n_nodes));
        if (due) {
            if (m->rank == 0) {
                PF(pfslam_shard_balance_build(h, frame));
                m->balance_builds++;
            }
            hipStream_t st = nullptr; // (nothing of a frame is in flight: pfslam_shard_balance_due has settled the handle)
            if (stream_of(h, 0, &st)) return 1;
            static const int ids[5] = {20, 21, 22, 23, 24};
            const size_t per_node[5] = {16, 4, 4, 4, 0};
            for (int k = 0; k < 5; k++) {
                void *p = nullptr;
                if (buf(h, ids[k], &p)) return 1;
                const size_t count = k < 4 ? per_node[k] * (size_t)n_nodes : 16;
                NCCL(ncclBroadcast(p, p, count, ncclChar, 0, m->comm_p, st));
            }
            if (m->rank != 0) PF(pfslam_shard_balance_adopt(h));
            m->balance_broadcasts++;
        }
    }
    int seeded = 0;
    PF(pfslam_shard_disperse(h, frame, scan, &seeded));
    if (seeded) return 0; // the first scan only seeds the (replicated) map
    hipStream_t st = nullptr;
    void *src = nullptr, *dst = nullptr;
    if (comm) { // the poses are final right after the dispersion: the particle stream has nothing else to do until the reduce
        if (stream_of(h, 0, &st) || buf(h, 16, &src) || buf(h, 17, &dst)) return 1;
        NCCL(ncclAllGather(src, dst, (size_t)3 * m->stride, ncclFloat, m->comm_p, st));
    }
    PF(pfslam_shard_score(h));
    if (comm) { // 16 bytes per rank, between the reduce and the walls on the chain stream
        if (stream_of(h, 1, &st) || buf(h, 14, &src) || buf(h, 15, &dst)) return 1;
        NCCL(ncclAllGather(src, dst, 16, ncclChar, m->comm_c, st));
    }
    PF(pfslam_shard_weights(h));
    if (comm) {
        if (stream_of(h, 2, &st) || buf(h, 5, &src) || buf(h, 10, &dst)) return 1;
        NCCL(ncclAllGather(src, dst, (size_t)m->stride, ncclFloat, m->comm_p, st));
    }
    PF(pfslam_shard_finish(h));
    m->collectives += 3;
    return 0;
}

extern "C" int pfslam_mgpu_barrier_max(pfslam_mgpu *m, double *value)
{
    if (!m) return mfail("pfslam_mgpu_barrier_max: null argument");
    HIP(hipSetDevice(m->device));
    PF(pfslam_synchronize(m->h)); // books the frames in flight, joins the frame's streams into the handle's, waits for it
    HIP(hipDeviceSynchronize());
    if (m->world == 1) return 0;
    double v = value ? *value : 0.0;
    hipStream_t st = nullptr; // (the handle's own stream: between frames nothing else is on it)
    if (stream_of(m->h, 0, &st)) return 1;
    HIP(hipMemcpyAsync(m->scratch, &v, 8, hipMemcpyHostToDevice, st));
    NCCL(ncclAllReduce(m->scratch, m->scratch, 1, ncclDouble, ncclMax, m->comm_p, st));
    HIP(hipMemcpyAsync(&v, m->scratch, 8, hipMemcpyDeviceToHost, st));
    HIP(hipStreamSynchronize(st));
    if (value) *value = v;
    return 0;
}

extern "C" int pfslam_mgpu_stats(pfslam_mgpu *m, int out[4])
{
    if (!m || !out) return mfail("pfslam_mgpu_stats: null argument");
    out[0] = m->collectives;
    out[1] = m->balance_builds;
    out[2] = m->balance_broadcasts;
    out[3] = m->world;
    return 0;
}

extern "C" int pfslam_mgpu_time_collectives(pfslam_mgpu *m, int reps, float ms[3])
{
    if (!m || !ms || reps <= 0) return mfail("pfslam_mgpu_time_collectives: bad argument");
    ms[0] = ms[1] = ms[2] = 0.0f;
    if (m->world == 1) return 0;
    HIP(hipSetDevice(m->device));
    PF(pfslam_synchronize(m->h));
    HIP(hipDeviceSynchronize());
    hipEvent_t a, b;
    HIP(hipEventCreate(&a));
    HIP(hipEventCreate(&b));
    // (between frames every collective's stream is the handle's own; the sizes and buffers are the frame's)
    const int srcs[3] = {16, 14, 5}, dsts[3] = {17, 15, 10};
    const size_t bytes[3] = {(size_t)3 * m->stride * 4, 16, (size_t)m->stride * 4};
    ncclComm_t comms[3] = {m->comm_p, m->comm_c, m->comm_p};
    hipStream_t st = nullptr;
    if (stream_of(m->h, 0, &st)) return 1;
    for (int k = 0; k < 3; k++) {
        void *src = nullptr, *dst = nullptr;
        if (buf(m->h, srcs[k], &src) || buf(m->h, dsts[k], &dst)) return 1;
        NCCL(ncclAllGather(src, dst, bytes[k], ncclChar, comms[k], st)); // warm
        HIP(hipStreamSynchronize(st));
        HIP(hipEventRecord(a, st));
        for (int r = 0; r < reps; r++) NCCL(ncclAllGather(src, dst, bytes[k], ncclChar, comms[k], st));
        HIP(hipEventRecord(b, st));
        HIP(hipEventSynchronize(b));
        float t = 0.0f;
        HIP(hipEventElapsedTime(&t, a, b));
        ms[k] = t / reps;
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return 0;
}
