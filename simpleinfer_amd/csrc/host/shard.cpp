// shard.cpp -- implementation of include/si_shard.h: the node-local rank group (POSIX shared memory) and the
// direct output all-gather over IPC-shared HBM buffers.  Host code only; the GPU is reached through si_hip.h.
// No reference counterpart (SimpleInfer is single-process); the partitioning it serves is SURVEY.md 8(e).
#include "si_shard.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "logger.h"

namespace {

constexpr uint32_t kMagic = 0x53494752u;  // "SIGR"

struct alignas(64) ShmHeader {
    std::atomic<uint32_t> magic;
    uint32_t world;
    std::atomic<uint32_t> attached;
    std::atomic<uint32_t> count;
    std::atomic<uint32_t> gen;
    std::atomic<uint32_t> failed;
};
static_assert(std::atomic<uint32_t>::is_always_lock_free, "cross-process atomics must be lock free");

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

// bounded wait: spin, then yield, then sleep 20 us at a time
template <typename Pred>
bool wait_until(Pred done, double timeout_s) {
    for (int i = 0; i < 4000; ++i) {
        if (done()) return true;
        if (i > 1000) sched_yield();
    }
    const double t0 = now_s();
    const timespec nap = {0, 20000};
    while (!done()) {
        if (now_s() - t0 > timeout_s) return false;
        nanosleep(&nap, nullptr);
    }
    return true;
}

size_t segment_bytes(int world) { return sizeof(ShmHeader) + (size_t)world * SI_GROUP_MAX_BYTES; }

}  // namespace

struct SiNodeGroup {
    int rank = 0, world = 1;
    double timeout_s = 60.0;
    ShmHeader* hdr = nullptr;
    unsigned char* slots = nullptr;
    size_t bytes = 0;
    std::string name;
};

extern "C" {

int si_group_create(const char* name, int rank, int world, double timeout_s, SiNodeGroup** group) {
    if (!group) return SI_SHARD_E_BADARG;
    *group = nullptr;
    if (!name || name[0] != '/' || world < 1 || world > SI_GROUP_MAX_WORLD || rank < 0 || rank >= world) return SI_SHARD_E_BADARG;
    if (timeout_s <= 0) timeout_s = 60.0;
    const size_t bytes = segment_bytes(world);
    int fd = -1;
    if (rank == 0) {
        shm_unlink(name);  // a leftover of a crashed job with the same name
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) { LOG(ERROR) << "si_group_create: shm_open(" << name << "): " << strerror(errno); return SI_SHARD_E_SYS; }
        if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); return SI_SHARD_E_SYS; }
    } else {
        // the segment appears (full size) once rank 0 has created it
        const bool ok = wait_until([&] {
            fd = shm_open(name, O_RDWR, 0600);
            if (fd < 0) return false;
            struct stat st;
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) return true;
            close(fd);
            fd = -1;
            return false;
        }, timeout_s);
        if (!ok) { LOG(ERROR) << "si_group_create: rank " << rank << " timed out waiting for " << name; return SI_SHARD_E_TIMEOUT; }
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { if (rank == 0) shm_unlink(name); return SI_SHARD_E_SYS; }
    ShmHeader* h = static_cast<ShmHeader*>(p);
    if (rank == 0) {
        // a fresh segment is zero filled; publish it with the magic word last
        h->world = (uint32_t)world;
        h->attached.store(0, std::memory_order_relaxed);
        h->count.store(0, std::memory_order_relaxed);
        h->gen.store(0, std::memory_order_relaxed);
        h->failed.store(0, std::memory_order_relaxed);
        h->magic.store(kMagic, std::memory_order_release);
    } else {
        const bool ok = wait_until([&] { return h->magic.load(std::memory_order_acquire) == kMagic; }, timeout_s);
        if (!ok || h->world != (uint32_t)world) { munmap(p, bytes); return ok ? SI_SHARD_E_BADARG : SI_SHARD_E_TIMEOUT; }
    }
    h->attached.fetch_add(1, std::memory_order_acq_rel);
    const bool all = wait_until([&] { return h->attached.load(std::memory_order_acquire) >= (uint32_t)world; }, timeout_s);
    if (rank == 0) shm_unlink(name);  // mapped segments live on; the name cannot go stale
    if (!all) {
        LOG(ERROR) << "si_group_create: only " << h->attached.load() << " of " << world << " ranks attached";
        munmap(p, bytes);
        return SI_SHARD_E_TIMEOUT;
    }
    SiNodeGroup* g = new (std::nothrow) SiNodeGroup;
    if (!g) { munmap(p, bytes); return SI_SHARD_E_SYS; }
    g->rank = rank; g->world = world; g->timeout_s = timeout_s; g->hdr = h;
    g->slots = reinterpret_cast<unsigned char*>(h + 1); g->bytes = bytes; g->name = name;
    *group = g;
    return 0;
}

int si_group_destroy(SiNodeGroup* g) {
    if (!g) return 0;
    if (g->hdr) munmap(g->hdr, g->bytes);
    delete g;
    return 0;
}

int si_group_rank(const SiNodeGroup* g) { return g ? g->rank : -1; }
int si_group_world(const SiNodeGroup* g) { return g ? g->world : -1; }

int si_group_barrier(SiNodeGroup* g) {
    if (!g) return SI_SHARD_E_BADARG;
    if (g->world == 1) return 0;
    ShmHeader* h = g->hdr;
    // a rank that timed out leaves its arrival in `count` behind: the group is dead from then on (every later barrier on
    // every rank fails instead of releasing with fewer than `world` arrivals)
    if (h->failed.load(std::memory_order_acquire)) return SI_SHARD_E_PEER;
    const uint32_t gen = h->gen.load(std::memory_order_acquire);
    if (h->count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)g->world) {
        h->count.store(0, std::memory_order_relaxed);
        h->gen.fetch_add(1, std::memory_order_acq_rel);
        return h->failed.load(std::memory_order_acquire) ? SI_SHARD_E_PEER : 0;
    }
    const bool ok = wait_until([&] { return h->gen.load(std::memory_order_acquire) != gen || h->failed.load(std::memory_order_acquire) != 0; },
                               g->timeout_s);
    if (!ok) {
        h->failed.store(1, std::memory_order_release);
        LOG(ERROR) << "si_group_barrier: rank " << g->rank << " timed out after " << g->timeout_s << " s";
        return SI_SHARD_E_TIMEOUT;
    }
    if (h->failed.load(std::memory_order_acquire)) {
        LOG(ERROR) << "si_group_barrier: rank " << g->rank << ": another rank timed out, the group is dead";
        return SI_SHARD_E_PEER;
    }
    return 0;
}

int si_group_allgather(SiNodeGroup* g, const void* mine, size_t bytes, void* all) {
    if (!g || (!mine && bytes) || !all) return SI_SHARD_E_BADARG;
    if (bytes > SI_GROUP_MAX_BYTES) return SI_SHARD_E_TOOBIG;
    memcpy(g->slots + (size_t)g->rank * SI_GROUP_MAX_BYTES, mine, bytes);
    int rc = si_group_barrier(g);
    if (rc != 0) return rc;
    for (int r = 0; r < g->world; ++r) memcpy(static_cast<unsigned char*>(all) + (size_t)r * bytes, g->slots + (size_t)r * SI_GROUP_MAX_BYTES, bytes);
    return si_group_barrier(g);  // nobody overwrites a slot before everybody has read it
}

}  // extern "C"

// ---- RCCL behind the C-ABI -----------------------------------------------------------------------------------------------
// librccl.so is loaded with dlopen at the first use: nothing links against it, and the five entry points are declared here with
// the (stable, NCCL-compatible) C signatures of rccl.h -- ncclResult_t is an int, ncclComm_t an opaque pointer, ncclUniqueId 128
// opaque bytes passed BY VALUE, ncclChar = 0.
namespace {

struct RcclUniqueId { char internal[128]; };
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

RcclApi& rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
        // SI_RCCL_LIB, when set, is the ONLY file tried (a test can name one that does not exist)
        const char* forced = getenv("SI_RCCL_LIB");
        const char* defaults[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        if (forced && forced[0]) {
            a.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char* n : defaults) {
                a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
                if (a.lib) break;
            }
        }
        if (!a.lib) return a;
        a.GetUniqueId = reinterpret_cast<int (*)(RcclUniqueId*)>(dlsym(a.lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<int (*)(void**, int, RcclUniqueId, int)>(dlsym(a.lib, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(a.lib, "ncclCommDestroy"));
        a.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, void*)>(dlsym(a.lib, "ncclAllGather"));
        a.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(a.lib, "ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.GetErrorString;
        return a;
    }();
    return api;
}

int rccl_fail(const char* what, int res) {
    LOG(ERROR) << what << ": RCCL error " << res << " (" << (rccl_api().GetErrorString ? rccl_api().GetErrorString(res) : "?") << ")";
    return SI_SHARD_E_RCCL;
}

}  // namespace

struct SiRcclComm {
    void* comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" {

int si_rccl_available(void) { return rccl_api().ok ? 1 : 0; }

int si_rccl_init(SiNodeGroup* g, int device, SiRcclComm** out) {
    if (!out) return SI_SHARD_E_BADARG;
    *out = nullptr;
    if (!g) return SI_SHARD_E_BADARG;
    RcclApi& api = rccl_api();
    // every rank learns whether every rank has the library before anybody enters a collective that the others would hang in
    int32_t have = api.ok ? 1 : 0;
    std::vector<int32_t> all((size_t)g->world);
    int rc = si_group_allgather(g, &have, sizeof(have), all.data());
    if (rc != 0) return rc;
    for (int32_t v : all)
        if (!v) {
            if (!api.ok) LOG(ERROR) << "si_rccl_init: librccl.so could not be loaded (" << (dlerror() ? "dlopen failed" : "symbols missing") << "); SI_RCCL_LIB names another file";
            return api.ok ? SI_SHARD_E_PEER : SI_SHARD_E_SYS;
        }
    RcclUniqueId id;
    memset(&id, 0, sizeof(id));
    int res = 0;
    if (g->rank == 0) res = api.GetUniqueId(&id);
    std::vector<RcclUniqueId> ids((size_t)g->world);
    rc = si_group_allgather(g, &id, sizeof(id), ids.data());
    if (rc != 0) return rc;
    int mine = res != 0 ? rccl_fail("ncclGetUniqueId", res) : si_hip_set_device(device);
    void* comm = nullptr;
    if (mine == 0) {
        res = api.CommInitRank(&comm, g->world, ids[0], g->rank);
        if (res != 0) mine = rccl_fail("ncclCommInitRank", res);
    }
    // (ncclCommInitRank is itself collective: a rank that failed BEFORE it leaves the others waiting inside RCCL until its own
    // timeout; what can be agreed on here is the outcome)
    std::vector<int32_t> rcs((size_t)g->world);
    const int32_t m32 = mine;
    rc = si_group_allgather(g, &m32, sizeof(m32), rcs.data());
    if (rc == 0 && mine == 0)
        for (int32_t v : rcs) if (v != 0) rc = SI_SHARD_E_PEER;
    if (mine != 0) rc = mine;
    if (rc != 0) {
        if (comm) api.CommDestroy(comm);
        return rc;
    }
    SiRcclComm* c = new (std::nothrow) SiRcclComm;
    if (!c) { api.CommDestroy(comm); return SI_SHARD_E_SYS; }
    c->comm = comm; c->rank = g->rank; c->world = g->world; c->device = device;
    *out = c;
    return 0;
}

int si_rccl_allgather(SiRcclComm* c, const void* send, void* recv, size_t bytes_per_rank, si_stream_t stream) {
    if (!c || !c->comm || !send || !recv) return SI_SHARD_E_BADARG;
    const int res = rccl_api().AllGather(send, recv, bytes_per_rank, /*ncclChar*/ 0, c->comm, stream);
    return res == 0 ? 0 : rccl_fail("ncclAllGather", res);
}

int si_rccl_destroy(SiRcclComm* c) {
    if (!c) return 0;
    int rc = 0;
    if (c->comm) {
        si_hip_set_device(c->device);
        const int res = rccl_api().CommDestroy(c->comm);
        if (res != 0) rc = rccl_fail("ncclCommDestroy", res);
    }
    delete c;
    return rc;
}

}  // extern "C"

// ---- direct all-gather -----------------------------------------------------------------------------------------------
struct SiDirectGather {
    SiNodeGroup* group = nullptr;
    int device = 0, slots = 0;
    size_t slab = 0;
    std::vector<void*> mine;                 // [slot] my gathered buffer, world x slab
    std::vector<std::vector<void*>> peer;    // [rank][slot] that rank's gathered buffer as mapped here (nullptr for me)
    std::vector<si_stream_t> stream;         // [rank] copy stream towards that peer
    std::vector<si_event_t> ready;           // [slot] producer finished writing the slab
    std::vector<std::vector<si_event_t>> sent;  // [slot][rank] my slab of that slot has landed in that peer
    std::vector<std::vector<si_event_t>> start; // [slot][rank] recorded on the copy stream right before that copy (bandwidth = slab / (sent - start))
    std::vector<char> pushed;                // [slot] a push is outstanding
    SiGatherStats stats;                     // since the last reset
    // SI_GATHER_RCCL: the same slot buffers, filled by an in-place ncclAllGather on `coll_stream` behind the producer; sent[slot][rank]
    // of THIS rank is the event that collective records
    int mode = SI_GATHER_DIRECT;
    SiRcclComm* rccl = nullptr;
    si_stream_t coll_stream = nullptr;
};

namespace {

struct Advert {  // what a rank tells the others about itself
    int32_t device;   // its index of the device in ITS process (diagnostics only: indices are per process)
    int32_t ok;
    char bus_id[32];  // the device's PCI bus id: the same GPU for every process, whatever HIP_VISIBLE_DEVICES each one has
    unsigned char handle[8][SI_IPC_HANDLE_BYTES];
};
static_assert(sizeof(Advert) <= SI_GROUP_MAX_BYTES, "advert must fit one group slot");

// every rank reports rc; all return the first non-zero one (so a setup failure on one rank fails every rank the same way)
int agree(SiNodeGroup* g, int rc) {
    std::vector<int32_t> all((size_t)g->world);
    const int32_t mine = rc;
    const int e = si_group_allgather(g, &mine, sizeof(mine), all.data());
    if (e != 0) return e;
    if (rc != 0) return rc;
    for (int32_t v : all) if (v != 0) return SI_SHARD_E_PEER;
    return 0;
}

void release(SiDirectGather* d) {
    for (auto& slots : d->peer) for (void* p : slots) if (p) si_hip_ipc_close_mem_handle(p);
    for (si_stream_t s : d->stream) if (s) si_hip_stream_destroy(s);
    if (d->coll_stream) si_hip_stream_destroy(d->coll_stream);
    if (d->rccl) si_rccl_destroy(d->rccl);
    for (si_event_t e : d->ready) if (e) si_hip_event_destroy(e);
    for (auto& evs : d->sent) for (si_event_t e : evs) if (e) si_hip_event_destroy(e);
    for (auto& evs : d->start) for (si_event_t e : evs) if (e) si_hip_event_destroy(e);
    for (void* p : d->mine) if (p) si_hip_free(p);
    delete d;
}

}  // namespace

extern "C" {

int si_gather_create(SiNodeGroup* g, int device, size_t slab_bytes, int slots, SiDirectGather** out) {
    if (!out) return SI_SHARD_E_BADARG;
    *out = nullptr;
    if (!g || slab_bytes == 0 || slots < 1 || slots > 8) return SI_SHARD_E_BADARG;
    SiDirectGather* d = new (std::nothrow) SiDirectGather;
    if (!d) return SI_SHARD_E_SYS;
    d->group = g; d->device = device; d->slots = slots; d->slab = slab_bytes;
    const int world = g->world, rank = g->rank;
    d->mine.assign((size_t)slots, nullptr);
    d->peer.assign((size_t)world, std::vector<void*>((size_t)slots, nullptr));
    d->stream.assign((size_t)world, nullptr);
    d->ready.assign((size_t)slots, nullptr);
    d->sent.assign((size_t)slots, std::vector<si_event_t>((size_t)world, nullptr));
    d->start.assign((size_t)slots, std::vector<si_event_t>((size_t)world, nullptr));
    d->pushed.assign((size_t)slots, 0);

    Advert me;
    memset(&me, 0, sizeof(me));
    me.device = device;
    memset(&d->stats, 0, sizeof(d->stats));
    int rc = si_hip_set_device(device);
    if (rc == 0) rc = si_hip_device_pci_bus_id(device, me.bus_id, (int)sizeof(me.bus_id));
    for (int s = 0; rc == 0 && s < slots; ++s) {
        rc = si_hip_malloc(&d->mine[(size_t)s], slab_bytes * (size_t)world);
        if (rc == 0) rc = si_hip_memset_async(d->mine[(size_t)s], 0, slab_bytes * (size_t)world, nullptr);
        if (rc == 0 && world > 1) rc = si_hip_ipc_get_mem_handle(d->mine[(size_t)s], me.handle[s]);
        if (rc == 0) rc = si_hip_event_create(&d->ready[(size_t)s]);
    }
    if (rc == 0) rc = si_hip_device_sync();
    me.ok = rc == 0;
    std::vector<Advert> all((size_t)world);
    int e = si_group_allgather(g, &me, sizeof(me), all.data());
    if (e == 0) for (const Advert& a : all) if (!a.ok && rc == 0) rc = SI_SHARD_E_PEER;
    if (e != 0) rc = e;
    if (rc != 0) { LOG(ERROR) << "si_gather_create: rank " << rank << " setup failed (" << rc << ")"; release(d); return rc; }

    for (int p = 0; rc == 0 && p < world; ++p) {
        if (p == rank) continue;
        // the peer's GPU, as THIS process numbers it (the peer's own index means nothing here when HIP_VISIBLE_DEVICES differs
        // between ranks); a GPU hidden from this process cannot be named for hipDeviceEnablePeerAccess -- the IPC mapping below
        // then has to carry the access by itself
        char peer_bus[32];
        memcpy(peer_bus, all[(size_t)p].bus_id, sizeof(peer_bus));
        peer_bus[sizeof(peer_bus) - 1] = 0;
        if (strcmp(peer_bus, me.bus_id) != 0) {
            const int local = si_hip_device_by_pci_bus_id(peer_bus);
            const int pe = local >= 0 ? si_hip_enable_peer_access(local) : -1;
            if (pe != 0) LOG(INFO) << "si_gather_create: peer access " << me.bus_id << " -> " << peer_bus << " not enabled (" << pe << "); relying on the IPC mapping";
        }
        for (int s = 0; rc == 0 && s < slots; ++s) rc = si_hip_ipc_open_mem_handle(all[(size_t)p].handle[s], &d->peer[(size_t)p][(size_t)s]);
        if (rc == 0) rc = si_hip_stream_create(&d->stream[(size_t)p]);
        for (int s = 0; rc == 0 && s < slots; ++s) rc = si_hip_event_create(&d->sent[(size_t)s][(size_t)p]);
        for (int s = 0; rc == 0 && s < slots; ++s) rc = si_hip_event_create(&d->start[(size_t)s][(size_t)p]);
    }
    rc = agree(g, rc);
    if (rc != 0) { LOG(ERROR) << "si_gather_create: opening the peers' buffers failed (" << rc << ")"; release(d); return rc; }
    *out = d;
    return 0;
}

// the RCCL-backed form of the same object: own slot buffers, one communicator, one stream for the collectives
static int gather_create_rccl(SiNodeGroup* g, int device, size_t slab_bytes, int slots, SiDirectGather** out) {
    SiDirectGather* d = new (std::nothrow) SiDirectGather;
    if (!d) return SI_SHARD_E_SYS;
    d->group = g; d->device = device; d->slots = slots; d->slab = slab_bytes; d->mode = SI_GATHER_RCCL;
    const int world = g->world, rank = g->rank;
    d->mine.assign((size_t)slots, nullptr);
    d->ready.assign((size_t)slots, nullptr);
    d->sent.assign((size_t)slots, std::vector<si_event_t>((size_t)world, nullptr));
    d->start.assign((size_t)slots, std::vector<si_event_t>((size_t)world, nullptr));
    d->pushed.assign((size_t)slots, 0);
    memset(&d->stats, 0, sizeof(d->stats));
    int rc = si_hip_set_device(device);
    for (int s = 0; rc == 0 && s < slots; ++s) {
        rc = si_hip_malloc(&d->mine[(size_t)s], slab_bytes * (size_t)world);
        if (rc == 0) rc = si_hip_memset_async(d->mine[(size_t)s], 0, slab_bytes * (size_t)world, nullptr);
        if (rc == 0) rc = si_hip_event_create(&d->ready[(size_t)s]);
        if (rc == 0) rc = si_hip_event_create(&d->sent[(size_t)s][(size_t)rank]);
        if (rc == 0) rc = si_hip_event_create(&d->start[(size_t)s][(size_t)rank]);
    }
    if (rc == 0) rc = si_hip_stream_create(&d->coll_stream);
    if (rc == 0) rc = si_hip_device_sync();
    rc = agree(g, rc);
    if (rc == 0) rc = si_rccl_init(g, device, &d->rccl);   // collective; agrees on its own outcome
    if (rc != 0) { LOG(ERROR) << "si_gather_create_mode: RCCL gather setup failed on rank " << rank << " (" << rc << ")"; release(d); return rc; }
    *out = d;
    return 0;
}

int si_gather_create_mode(SiNodeGroup* g, int device, size_t slab_bytes, int slots, int mode, SiDirectGather** out) {
    if (!out) return SI_SHARD_E_BADARG;
    *out = nullptr;
    if (!g || slab_bytes == 0 || slots < 1 || slots > 8) return SI_SHARD_E_BADARG;
    if (mode == SI_GATHER_DIRECT) return si_gather_create(g, device, slab_bytes, slots, out);
    if (mode == SI_GATHER_RCCL) return gather_create_rccl(g, device, slab_bytes, slots, out);
    if (mode != SI_GATHER_AUTO) return SI_SHARD_E_BADARG;
    // si_gather_create fails on every rank when it fails on one (its setup steps are agreed on), so the fallback is collective
    const int rc = si_gather_create(g, device, slab_bytes, slots, out);
    if (rc == 0) return 0;
    if (rc == SI_SHARD_E_TIMEOUT) return rc;   // a dead group cannot agree on anything
    LOG(INFO) << "si_gather_create_mode: the direct gather could not be set up (" << rc << "); falling back to RCCL on every rank";
    return gather_create_rccl(g, device, slab_bytes, slots, out);
}

int si_gather_mode(const SiDirectGather* d) { return d ? d->mode : -1; }

int si_gather_destroy(SiDirectGather* d) {
    if (!d) return 0;
    si_hip_set_device(d->device);
    for (si_stream_t s : d->stream) if (s) si_hip_stream_sync(s);
    if (d->coll_stream) si_hip_stream_sync(d->coll_stream);
    // nobody frees a buffer a peer may still be copying into
    const int rc = si_group_barrier(d->group);
    SiNodeGroup* g = d->group;
    for (auto& slots : d->peer) for (void*& p : slots) if (p) { si_hip_ipc_close_mem_handle(p); p = nullptr; }
    const int rc2 = si_group_barrier(g);  // every mapping is closed before the owner frees
    release(d);
    return rc != 0 ? rc : rc2;
}

int si_gather_slots(const SiDirectGather* d) { return d ? d->slots : -1; }
size_t si_gather_slab_bytes(const SiDirectGather* d) { return d ? d->slab : 0; }
void* si_gather_buffer(SiDirectGather* d, int slot) { return (d && slot >= 0 && slot < d->slots) ? d->mine[(size_t)slot] : nullptr; }
void* si_gather_slab(SiDirectGather* d, int slot) {
    void* b = si_gather_buffer(d, slot);
    return b ? static_cast<unsigned char*>(b) + (size_t)d->group->rank * d->slab : nullptr;
}

int si_gather_push(SiDirectGather* d, int slot, si_stream_t producer) {
    if (!d || slot < 0 || slot >= d->slots) return SI_SHARD_E_BADARG;
    const int world = d->group->world, rank = d->group->rank;
    if (world == 1 && d->mode != SI_GATHER_RCCL) return 0;   // (a one-rank RCCL gather still runs its collective: tests)
    int rc = si_hip_event_record(d->ready[(size_t)slot], producer);
    if (d->mode == SI_GATHER_RCCL) {
        // in place: this rank's slab already sits at its offset of the slot buffer
        unsigned char* buf = static_cast<unsigned char*>(d->mine[(size_t)slot]);
        if (rc == 0) rc = si_hip_stream_wait_event(d->coll_stream, d->ready[(size_t)slot]);
        if (rc == 0) rc = si_hip_event_record(d->start[(size_t)slot][(size_t)rank], d->coll_stream);
        if (rc == 0) rc = si_rccl_allgather(d->rccl, buf + (size_t)rank * d->slab, buf, d->slab, d->coll_stream);
        if (rc == 0) rc = si_hip_event_record(d->sent[(size_t)slot][(size_t)rank], d->coll_stream);
        d->pushed[(size_t)slot] = 1;
        return rc;
    }
    const size_t off = (size_t)rank * d->slab;
    const unsigned char* src = static_cast<const unsigned char*>(d->mine[(size_t)slot]) + off;
    // one copy per peer, each on its own stream: point-to-point links, no ring; start at rank + 1 so that at any moment the
    // ranks are not all writing into the same destination
    for (int i = 1; rc == 0 && i < world; ++i) {
        const int p = (rank + i) % world;
        rc = si_hip_stream_wait_event(d->stream[(size_t)p], d->ready[(size_t)slot]);
        if (rc == 0) rc = si_hip_event_record(d->start[(size_t)slot][(size_t)p], d->stream[(size_t)p]);
        if (rc == 0) rc = si_hip_memcpy_d2d(static_cast<unsigned char*>(d->peer[(size_t)p][(size_t)slot]) + off, src, d->slab, d->stream[(size_t)p]);
        if (rc == 0) rc = si_hip_event_record(d->sent[(size_t)slot][(size_t)p], d->stream[(size_t)p]);
    }
    d->pushed[(size_t)slot] = 1;
    return rc;
}

int si_gather_complete(SiDirectGather* d, int slot) {
    if (!d || slot < 0 || slot >= d->slots) return SI_SHARD_E_BADARG;
    int rc = 0;
    const double t0 = now_s();
    // a copy command completes with its bytes visible at system scope; the barrier then tells every rank that all peers'
    // pushes into ITS buffer are done; kernels launched afterwards see them
    // (only THIS slot's copies are waited for: a later slot's fan-out keeps running behind the next step's compute)
    if (d->pushed[(size_t)slot]) {
        for (si_event_t e : d->sent[(size_t)slot]) if (e && rc == 0) rc = si_hip_event_sync(e);
        d->pushed[(size_t)slot] = 0;
        // per peer copy: the copy itself (start event on the copy stream right before it -> landed: slab / that is the link
        // rate) and the latency behind the producer (slab ready -> landed, which includes queueing behind earlier copies)
        for (size_t p = 0; p < d->sent[(size_t)slot].size(); ++p) {
            si_event_t e = d->sent[(size_t)slot][p];
            float ms = 0.f, lat = 0.f;
            if (e && rc == 0 && si_hip_event_elapsed_ms(d->start[(size_t)slot][p], e, &ms) == 0 && ms > 0.f) {
                d->stats.copy_ms_total += ms;
                d->stats.copies += 1;
                if (ms > d->stats.copy_ms_max) d->stats.copy_ms_max = ms;
                if (si_hip_event_elapsed_ms(d->ready[(size_t)slot], e, &lat) == 0 && lat > 0.f) d->stats.landed_ms_total += lat;
            }
        }
    }
    const double t1 = now_s();
    // (RCCL: the collective itself orders the ranks -- every rank's slab is in this buffer once THIS rank's collective is done)
    const int b = d->mode == SI_GATHER_RCCL ? 0 : si_group_barrier(d->group);
    const double t2 = now_s();
    d->stats.completes += 1;
    d->stats.wait_copies_ms_total += (t1 - t0) * 1e3;
    d->stats.wait_barrier_ms_total += (t2 - t1) * 1e3;
    return rc != 0 ? rc : b;
}

int si_gather_stats(SiDirectGather* d, SiGatherStats* out, int reset) {
    if (!d || !out) return SI_SHARD_E_BADARG;
    *out = d->stats;
    if (reset) memset(&d->stats, 0, sizeof(d->stats));
    return 0;
}

}  // extern "C"
