/* pfslam_mgpu.h -- the sharded frame of pfslam.h with its collectives on librccl (RCCL over xGMI), as a C ABI of its own
 * (libpfslam_mgpu.so: links libpfslam_hip.so and librccl.so; libpfslam_hip.so itself stays free of any collective library).
 *
 * One process per GPU, one pfslam handle per process (created by the caller with global_offset / global_n / shard_stride set:
 * pfslam.h "multi-GPU").  The reference is single-GPU (kernel.cu:1702-1762 is the frame; kernel.h:14-24 its boundary), so nothing
 * here replaces a reference interface: this is the multi-GPU driver north_star asks for around the same frame.
 *
 * A rank's three all-gathers per frame are launched STRAIGHT INTO THE FRAME'S OWN STREAMS (pfslam_shard_stream): ncclAllGather with the
 * particle stream (pose blocks under the scan-match kernel; weights) or the chain stream (the 16-byte keys, between the reduce and the
 * walls) as its stream argument -- no event on either side, no stream of the collectives' own (a fifth busy stream halves the frame rate
 * on this runtime).  Two communicators, one per stream, so that RCCL never has to order a communicator's operations across streams.
 * The re-balance (frame % balance_period == 5) runs ONCE per node: rank 0 builds, 28 bytes per node are broadcast.
 *
 * Every function returns 0 on success; pfslam_mgpu_last_error() has the message otherwise (thread-local). */
#ifndef PFSLAM_MGPU_H
#define PFSLAM_MGPU_H

#include "pfslam.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pfslam_mgpu pfslam_mgpu;

#define PFSLAM_MGPU_ID_BYTES 256 /* two ncclUniqueId (128 bytes each): particle-stream and chain-stream communicator */

/* rank 0 makes the job's id; the caller carries it to the other ranks (a file, torch.distributed's store, MPI ...) */
int pfslam_mgpu_make_id(unsigned char id[PFSLAM_MGPU_ID_BYTES]);
/* every rank, collectively (ncclCommInitRank x 2).  `h` is borrowed: it must outlive the pfslam_mgpu and is stepped only through it.
 * world == 1: no communicator is made (id may be NULL), pfslam_mgpu_step is the four pfslam_shard_* calls back to back. */
int pfslam_mgpu_create(const unsigned char *id, int world, int rank, pfslam_handle *h, pfslam_mgpu **out);
int pfslam_mgpu_destroy(pfslam_mgpu *m);
/* one frame of the sharded job, ENQUEUED (no call waits for the device except in front of a re-balance): balance protocol,
 * pfslam_shard_disperse / score / weights / finish with the three all-gathers between them */
int pfslam_mgpu_step(pfslam_mgpu *m, int frame, const float *scan_host);
/* barrier across the ranks; *value (may be NULL) becomes the MAX over the ranks.  Settles the handle (frames in flight are booked)
 * and synchronises its streams first. */
int pfslam_mgpu_barrier_max(pfslam_mgpu *m, double *value);
/* out[0] all-gathers issued, [1] host re-balances this rank ran, [2] re-balance broadcasts it took part in, [3] world size */
int pfslam_mgpu_stats(pfslam_mgpu *m, int out[4]);
/* the frame's three all-gathers on their own, `reps` times each, on the streams the frame issues them on: average milliseconds per
 * call (this rank), ms[0] pose blocks, [1] keys, [2] weights.  Collective: every rank calls it with the same reps. */
int pfslam_mgpu_time_collectives(pfslam_mgpu *m, int reps, float ms[3]);
const char *pfslam_mgpu_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
