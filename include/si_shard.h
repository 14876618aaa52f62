/*
 * si_shard.h -- C-ABI of the multi-GPU part of the path: one process per GPU of ONE node, the batch sharded by
 * contiguous slabs, and ONE exchange per step -- the all-gather of the output slabs [B/G, rows, 85] into
 * [B, rows, 85] (BASELINE.json north_star).  The reference has no counterpart
 * (SimpleInfer is single-process): these entry points are what a C++ / cgo / JNI host of
 * include/engine.h binds to shard a batch without torch.
 *
 * Two pieces:
 *   SiNodeGroup    host-only rendezvous of the node's ranks through POSIX shared memory: barrier and small
 *                  all-gather of host bytes (handles, checksums, timings).  No GPU, no sockets.
 *   SiDirectGather the direct (non-ring) all-gather: every rank owns `slots` gathered buffers
 *                  [world][slab_bytes] in its HBM, exports them with hipIpcGetMemHandle, opens every peer's, and
 *                  after each step PUSHES its slab into the same slot of every peer with one device-to-device
 *                  copy per peer on that peer's own stream (xGMI is point to point: 7 concurrent copies use 7
 *                  links; a ring would move 7 slabs over one link per GPU).  The producer writes its slab in
 *                  place (si_gather_slab is where Engine::Output binds), so the local copy costs nothing.
 *
 * Step protocol (slot = step % slots, slots >= 3 for the overlapped form, 4 by default):
 *     bind the engine's output to si_gather_slab(g, slot); Forward()
 *     si_gather_push(g, slot, engine_stream)        asynchronous fan-out behind the producer's work
 *     si_gather_complete(g, previous slot)          waits for MY pushes of the previous step, then the node
 *                                                    barrier: every rank's slab of that step is now in
 *                                                    si_gather_buffer(g, previous slot)
 * How long a completed slot may be read.  Call C(t) the si_gather_complete of step t (made during step t + 1).  Step
 * s + slots is pushed into the slot of step s, and a peer may start that step as soon as it has left the barrier of
 * C(s + slots - 2), i.e. as soon as THIS rank has ENTERED that call.  So after C(s) returns, the slot of step s is
 * stable until this rank makes its (slots - 2)-th FURTHER si_gather_complete call: with 3 slots only until the next
 * call (a consumer must have finished reading before the next step completes), with 4 slots -- the default of
 * ShardedEngine / ShardedForward -- through one more whole step, the room an asynchronous consumer (device NMS on the
 * gathered tensor while the next Forward runs) needs.  With
 * slots == 1 call push and complete back to back (no overlap).
 *
 * Return value: 0 on success; positive hipError_t / negative SI_E_* from the HIP layer (include/si_hip.h); or
 * SI_SHARD_E_* below.
 */
#ifndef SI_SHARD_H_
#define SI_SHARD_H_

#include <stddef.h>

#include "si_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SI_SHARD_E_BADARG (-101)
#define SI_SHARD_E_SYS (-102)      /* shm_open / mmap / ftruncate failed */
#define SI_SHARD_E_TIMEOUT (-103)  /* a peer did not arrive within the group's timeout */
#define SI_SHARD_E_TOOBIG (-104)   /* more than SI_GROUP_MAX_BYTES per rank in si_group_allgather */
#define SI_SHARD_E_PEER (-105)     /* another rank reported a failure in a collective setup step */
#define SI_SHARD_E_RCCL (-106)     /* an RCCL call failed (the ncclResult_t is logged); librccl.so missing is SI_SHARD_E_SYS */

#define SI_GROUP_MAX_BYTES 4096
#define SI_GROUP_MAX_WORLD 64

typedef struct SiNodeGroup SiNodeGroup;
typedef struct SiDirectGather SiDirectGather;
typedef struct SiRcclComm SiRcclComm;

/* Every rank of the node calls this with the same `name` (a POSIX shm name: "/something", unique per job -- e.g.
 * derived from the launcher's pid and port) and `world`; rank 0 creates the segment, the others attach, and the
 * name is unlinked as soon as all have (nothing stale survives a crash).  `timeout_s` bounds every wait. */
int si_group_create(const char* name, int rank, int world, double timeout_s, SiNodeGroup** group);
int si_group_destroy(SiNodeGroup* group);
int si_group_rank(const SiNodeGroup* group);
int si_group_world(const SiNodeGroup* group);
int si_group_barrier(SiNodeGroup* group);
/* all[r * bytes .. ] = rank r's `mine`; bytes <= SI_GROUP_MAX_BYTES, the same on every rank */
int si_group_allgather(SiNodeGroup* group, const void* mine, size_t bytes, void* all);

/* Collective over the group.  `device`: this rank's HIP device.  Allocates slots x world x slab_bytes of HBM. */
int si_gather_create(SiNodeGroup* group, int device, size_t slab_bytes, int slots, SiDirectGather** gather);
/* The same object with the transport chosen (round 4; north_star: "RCCL all-gather of outputs over xGMI"):
 *   SI_GATHER_DIRECT  the IPC fan-out above (si_gather_create)
 *   SI_GATHER_RCCL    ncclAllGather (in place, on the gather's own stream behind the producer) into the same slot buffers; no IPC
 *                     mapping, no peer access: what works wherever RCCL does
 *   SI_GATHER_AUTO    direct; if its setup fails on ANY rank (hipIpc / peer mapping refused on the real node), every rank tears it
 *                     down and all fall back to RCCL together
 * push / complete / buffer / slab / stats / destroy are the same calls for every mode (for RCCL `complete` waits for this rank's
 * collective of that slot; the collective itself orders the ranks, so there is no node barrier in it). */
#define SI_GATHER_DIRECT 0
#define SI_GATHER_RCCL 1
#define SI_GATHER_AUTO 2
int si_gather_create_mode(SiNodeGroup* group, int device, size_t slab_bytes, int slots, int mode, SiDirectGather** gather);
int si_gather_mode(const SiDirectGather* gather); /* SI_GATHER_DIRECT or SI_GATHER_RCCL: what it ended up as */
int si_gather_destroy(SiDirectGather* gather); /* collective */
int si_gather_slots(const SiDirectGather* gather);
size_t si_gather_slab_bytes(const SiDirectGather* gather);
void* si_gather_buffer(SiDirectGather* gather, int slot); /* device pointer, world x slab_bytes */
void* si_gather_slab(SiDirectGather* gather, int slot);   /* = buffer + rank x slab_bytes: the producer writes here */
int si_gather_push(SiDirectGather* gather, int slot, si_stream_t producer_stream);
int si_gather_complete(SiDirectGather* gather, int slot); /* collective */

/* Where a step's time went, accumulated since the last reset -- what tells a reader of an N > 1 benchmark line whether the
 * steps were compute- or gather-bound:
 *   wait_copies_ms_total   host time si_gather_complete spent waiting for THIS rank's peer copies of the slot (0 when the
 *                          fan-out finished behind the next step's compute: the overlap worked)
 *   wait_barrier_ms_total  ... and then in the node barrier (waiting for the slowest rank)
 *   copy_ms_total / copies device time of a peer copy itself (an event recorded on the copy stream right before it -> landed):
 *                          slab_bytes / that is the achieved per-link rate (xGMI: ~153 GB/s per link peak)
 *   landed_ms_total        device time from "slab ready" to "landed in the peer", summed over the same copies: the latency
 *                          behind the producer, queueing behind earlier copies on that peer's stream included */
typedef struct SiGatherStats {
    double wait_copies_ms_total;
    double wait_barrier_ms_total;
    double copy_ms_total;
    double copy_ms_max;
    long long completes;
    long long copies;
    double landed_ms_total;
} SiGatherStats;
int si_gather_stats(SiDirectGather* gather, SiGatherStats* out, int reset);

/* ---- RCCL behind the C-ABI (SURVEY.md 8b `si_rccl_allgather`).  librccl.so is dlopen()ed at the first call (SI_RCCL_LIB names
 * another file); nothing links against it, so the library loads -- and every other entry point works -- where RCCL is absent:
 * these then return SI_SHARD_E_SYS.  No torch, no MPI: the ncclUniqueId travels through si_group_allgather. */
int si_rccl_available(void); /* 1 when librccl.so can be loaded and has the five entry points used here */
/* collective over the group: rank 0 draws the unique id, all ranks ncclCommInitRank on `device` */
int si_rccl_init(SiNodeGroup* group, int device, SiRcclComm** comm);
/* recv[r * bytes_per_rank ..] = rank r's send[0 .. bytes_per_rank), enqueued on `stream` (hipStream_t); in place when
 * send == recv + rank * bytes_per_rank */
int si_rccl_allgather(SiRcclComm* comm, const void* send, void* recv, size_t bytes_per_rank, si_stream_t stream);
int si_rccl_destroy(SiRcclComm* comm);

#ifdef __cplusplus
}
#endif

#endif /* SI_SHARD_H_ */
