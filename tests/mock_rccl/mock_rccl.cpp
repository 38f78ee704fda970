// mock_rccl -- a TEST DOUBLE for librccl (test infrastructure; never shipped, never loaded by the product unless GSR_RCCL_LIB names it).
//
// Why: csrc/comm.hip calls RCCL itself (ncclAllReduce / ncclAllGather / grouped ncclSend + ncclRecv), but RCCL refuses two ranks on
// one device and the boxes of this pool have ONE GPU, so the RCCL transport of the library had only ever run with world = 1.  This
// library implements the ten entry points comm.hip binds, for several PROCESSES THAT SHARE ONE GPU, over POSIX shared memory, so that
// the library's RCCL code path -- which calls it makes, in which order, with which counts, types, peers and pointers -- executes with
// world > 1 under the real kernels (tests/test_distributed_gpu.py, the *_mock_rccl tests).  It is STRICTER than RCCL where that
// catches a bug early: every collective checks that all ranks issued the same operation with the same count and type (RCCL would
// hang or corrupt), a send must meet a receive of exactly its size, an in-place all-gather must have its send buffer at
// recvbuff + rank * count.  It is NOT RCCL: operations run synchronously at the call (stream-ordered only in that the stream is
// synchronised first and the result is in place when the call returns), nothing about xGMI is exercised, no performance meaning.
//
// build: hipcc -shared -fPIC -O1 tests/mock_rccl/mock_rccl.cpp -o tests/mock_rccl/libmock_rccl.so -lrt   (the test does it)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <vector>

namespace {

constexpr int MAX_RANKS = 8;
// per rank: what one rank publishes in one operation.  96 MB by default (the tests stay far below); GSR_MOCK_RCCL_SLOT_MB raises it for
// the full-size runs of BASELINE configs[4] (40 M splats over 8 ranks: a rank's SH halo rows are a few hundred MB).  Every rank reads the
// same environment; the segment is sparse (POSIX shm: only touched pages exist).
size_t slot_bytes() {
    static const size_t v = [] {
        const char* e = getenv("GSR_MOCK_RCCL_SLOT_MB");
        const long mb = e ? atol(e) : 0;
        return (size_t)(mb > 0 ? mb : 96) << 20;
    }();
    return v;
}
#define SLOT_BYTES (slot_bytes())
constexpr int MAX_DIR = 64;

struct DirEntry { int peer; unsigned long long off, bytes; };
struct RankHeader {
    unsigned long long seq;                 // operations issued so far
    int op, dtype, redop;
    unsigned long long count;
    int ndir;
    DirEntry dir[MAX_DIR];                  // the sends of a group: peer, offset in the slot, bytes
};
struct Shared {
    std::atomic<int> attached;
    std::atomic<int> barrier_count;
    std::atomic<unsigned> barrier_gen;
    std::atomic<int> failed;                // a rank saw an inconsistency: everybody fails from then on instead of waiting
    RankHeader hdr[MAX_RANKS];
};

struct PendingOp { bool send; void* buf; size_t bytes; int peer; hipStream_t stream; };

struct Comm {
    int rank = 0, world = 1;
    Shared* sh = nullptr;
    unsigned char* slots = nullptr;          // world x SLOT_BYTES behind the header
    size_t map_bytes = 0;
    char name[64];
    std::vector<unsigned char> host;         // staging for the result
};

thread_local int g_group_depth = 0;
thread_local std::vector<PendingOp> g_pending;
thread_local Comm* g_group_comm = nullptr;

const char* g_err = "mock_rccl: ok";

size_t dtype_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

bool barrier(Comm* c) {
    Shared* s = c->sh;
    const unsigned gen = s->barrier_gen.load();
    if (s->barrier_count.fetch_add(1) + 1 == c->world) {
        s->barrier_count.store(0);
        s->barrier_gen.fetch_add(1);
        return s->failed.load() == 0;
    }
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0; s->barrier_gen.load() == gen; ++spins) {
        if (s->failed.load()) return false;
        if ((spins & 1023u) == 1023u) {
            sched_yield();
            timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (t1.tv_sec - t0.tv_sec > 120) { s->failed.store(1); g_err = "mock_rccl: a rank did not arrive within 120 s (collectives out of step?)"; return false; }
        }
    }
    return s->failed.load() == 0;
}

ncclResult_t fail(Comm* c, const char* msg) {
    g_err = msg;
    if (c && c->sh) c->sh->failed.store(1);
    fprintf(stderr, "[mock_rccl rank %d] %s\n", c ? c->rank : -1, msg);
    return ncclInvalidUsage;
}

// every rank publishes what it is about to do; after the barrier everyone checks that all did the same
ncclResult_t announce(Comm* c, int op, ncclDataType_t dt, int redop, size_t count) {
    RankHeader& h = c->sh->hdr[c->rank];
    h.seq += 1; h.op = op; h.dtype = (int)dt; h.redop = redop; h.count = count;
    if (!barrier(c)) return fail(c, g_err);
    for (int r = 0; r < c->world; ++r) {
        const RankHeader& o = c->sh->hdr[r];
        if (o.seq != h.seq || o.op != op || (op != 3 && (o.dtype != (int)dt || o.redop != redop || o.count != count)))
            return fail(c, "mock_rccl: the ranks issued different operations (order, count, type or reduction differ) -- RCCL would hang here");
    }
    return ncclSuccess;
}

template <typename T>
void reduce_into(T* acc, const T* src, size_t n, bool sum) {
    for (size_t i = 0; i < n; ++i) acc[i] = sum ? (T)(acc[i] + src[i]) : (src[i] > acc[i] ? src[i] : acc[i]);
}

ncclResult_t run_group(Comm* c, std::vector<PendingOp>& ops) {
    // the sends go into this rank's slot, with a directory; then everyone reads what is addressed to it
    RankHeader& h = c->sh->hdr[c->rank];
    h.ndir = 0;
    unsigned long long off = 0;
    unsigned char* mine = c->slots + (size_t)c->rank * SLOT_BYTES;
    for (const PendingOp& p : ops) {
        if (!p.send) continue;
        if (p.peer < 0 || p.peer >= c->world || p.peer == c->rank) return fail(c, "mock_rccl: ncclSend to an invalid peer");
        if (h.ndir >= MAX_DIR || off + p.bytes > SLOT_BYTES) return fail(c, "mock_rccl: a group exceeds the mock's slot (raise SLOT_BYTES)");
        if (hipStreamSynchronize(p.stream) != hipSuccess) return fail(c, "mock_rccl: hipStreamSynchronize failed");
        if (p.bytes && hipMemcpy(mine + off, p.buf, p.bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(c, "mock_rccl: D2H copy of a send failed");
        h.dir[h.ndir++] = {p.peer, off, (unsigned long long)p.bytes};
        off += p.bytes;
    }
    const ncclResult_t a = announce(c, 3, ncclUint8, 0, 0);
    if (a != ncclSuccess) return a;
    // pairing: every receive must meet exactly one send of its size, and every send addressed to me must be received
    int n_recv_from[MAX_RANKS] = {0};
    for (const PendingOp& p : ops) {
        if (p.send) continue;
        if (p.peer < 0 || p.peer >= c->world || p.peer == c->rank) return fail(c, "mock_rccl: ncclRecv from an invalid peer");
        const RankHeader& o = c->sh->hdr[p.peer];
        int seen = 0;
        bool done = false;
        for (int k = 0; k < o.ndir && !done; ++k) {
            if (o.dir[k].peer != c->rank) continue;
            if (seen++ != n_recv_from[p.peer]) continue;            // the k-th receive from a peer takes its k-th send to me
            if (o.dir[k].bytes != p.bytes) return fail(c, "mock_rccl: a receive does not match the size of the peer's send");
            if (hipStreamSynchronize(p.stream) != hipSuccess) return fail(c, "mock_rccl: hipStreamSynchronize failed");
            if (p.bytes && hipMemcpy(p.buf, c->slots + (size_t)p.peer * SLOT_BYTES + o.dir[k].off, p.bytes, hipMemcpyHostToDevice) != hipSuccess)
                return fail(c, "mock_rccl: H2D copy of a receive failed");
            done = true;
        }
        if (!done) return fail(c, "mock_rccl: ncclRecv without a matching ncclSend on the peer -- RCCL would hang here");
        n_recv_from[p.peer] += 1;
    }
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        int to_me = 0;
        for (int k = 0; k < c->sh->hdr[r].ndir; ++k) to_me += c->sh->hdr[r].dir[k].peer == c->rank ? 1 : 0;
        if (to_me != n_recv_from[r]) return fail(c, "mock_rccl: a peer's ncclSend has no matching ncclRecv here -- RCCL would hang here");
    }
    if (!barrier(c)) return fail(c, g_err);                        // nobody overwrites its slot before everyone has read
    return ncclSuccess;
}

}  // namespace

extern "C" {

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : g_err; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    memset(id, 0, sizeof(*id));
    timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof(id->internal), "/gsr_mock_rccl_%d_%ld", (int)getpid(), (long)t.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) { g_err = "mock_rccl: bad rank / world"; return ncclInvalidArgument; }
    Comm* c = new Comm();
    c->rank = rank; c->world = nranks;
    strncpy(c->name, id.internal, sizeof(c->name) - 1);
    c->map_bytes = sizeof(Shared) + (size_t)nranks * SLOT_BYTES;
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { g_err = "mock_rccl: shm_open / ftruncate failed"; delete c; return ncclSystemError; }
    } else {
        for (int tries = 0; tries < 20000 && fd < 0; ++tries) { fd = shm_open(c->name, O_RDWR, 0600); if (fd < 0) usleep(1000); }
        if (fd < 0) { g_err = "mock_rccl: the shared segment did not appear"; delete c; return ncclSystemError; }
        for (int tries = 0; tries < 20000; ++tries) {              // wait until rank 0 has sized it
            off_t sz = lseek(fd, 0, SEEK_END);
            if (sz >= (off_t)c->map_bytes) break;
            usleep(1000);
        }
    }
    void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { g_err = "mock_rccl: mmap failed"; delete c; return ncclSystemError; }
    c->sh = reinterpret_cast<Shared*>(p);
    c->slots = reinterpret_cast<unsigned char*>(p) + sizeof(Shared);
    c->sh->attached.fetch_add(1);
    for (int tries = 0; c->sh->attached.load() < nranks; ++tries) { if (tries > 120000) { g_err = "mock_rccl: not every rank attached"; return ncclSystemError; } usleep(1000); }
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    if (c->sh) {
        const int left = c->sh->attached.fetch_sub(1) - 1;
        munmap(c->sh, c->map_bytes);
        if (left == 0) shm_unlink(c->name);
    }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { g_group_depth += 1; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
    if (g_group_depth <= 0) { g_err = "mock_rccl: ncclGroupEnd without ncclGroupStart"; return ncclInvalidUsage; }
    if (--g_group_depth > 0) return ncclSuccess;
    ncclResult_t r = ncclSuccess;
    if (g_group_comm) r = run_group(g_group_comm, g_pending);      // (an empty group on every rank: still one synchronised step)
    g_pending.clear();
    g_group_comm = nullptr;
    return r;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t sz = dtype_size(dt);
    if (!c || !sz) { g_err = "mock_rccl: ncclSend: bad communicator or type"; return ncclInvalidArgument; }
    if (g_group_comm && g_group_comm != c) return fail(c, "mock_rccl: one group, two communicators");
    g_group_comm = c;
    g_pending.push_back({true, const_cast<void*>(buf), count * sz, peer, stream});
    if (g_group_depth == 0) { std::vector<PendingOp> one; one.swap(g_pending); g_group_comm = nullptr; return run_group(c, one); }
    return ncclSuccess;
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t sz = dtype_size(dt);
    if (!c || !sz) { g_err = "mock_rccl: ncclRecv: bad communicator or type"; return ncclInvalidArgument; }
    if (g_group_comm && g_group_comm != c) return fail(c, "mock_rccl: one group, two communicators");
    g_group_comm = c;
    g_pending.push_back({false, buf, count * sz, peer, stream});
    if (g_group_depth == 0) { std::vector<PendingOp> one; one.swap(g_pending); g_group_comm = nullptr; return run_group(c, one); }
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t dt, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t sz = dtype_size(dt);
    if (!c || !sz) { g_err = "mock_rccl: ncclAllReduce: bad communicator or type"; return ncclInvalidArgument; }
    if (g_group_depth > 0) return fail(c, "mock_rccl: collectives inside a group are not part of this mock");
    if (op != ncclSum && op != ncclMax) return fail(c, "mock_rccl: only ncclSum and ncclMax");
    const size_t bytes = count * sz;
    if (bytes > SLOT_BYTES) return fail(c, "mock_rccl: all-reduce larger than the mock's slot");
    if (hipStreamSynchronize(stream) != hipSuccess) return fail(c, "mock_rccl: hipStreamSynchronize failed");
    if (bytes && hipMemcpy(c->slots + (size_t)c->rank * SLOT_BYTES, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(c, "mock_rccl: D2H copy failed");
    const ncclResult_t a = announce(c, 1, dt, (int)op, count);
    if (a != ncclSuccess) return a;
    c->host.resize(bytes);
    memcpy(c->host.data(), c->slots, bytes);                       // rank 0's contribution first, then 1, 2, ...: the same order on every rank
    for (int r = 1; r < c->world; ++r) {
        const unsigned char* src = c->slots + (size_t)r * SLOT_BYTES;
        const bool sum = op == ncclSum;
        switch (dt) {
            case ncclFloat64: reduce_into((double*)c->host.data(), (const double*)src, count, sum); break;
            case ncclFloat32: reduce_into((float*)c->host.data(), (const float*)src, count, sum); break;
            case ncclInt32: reduce_into((int32_t*)c->host.data(), (const int32_t*)src, count, sum); break;
            case ncclUint32: reduce_into((uint32_t*)c->host.data(), (const uint32_t*)src, count, sum); break;
            case ncclInt64: reduce_into((int64_t*)c->host.data(), (const int64_t*)src, count, sum); break;
            case ncclUint64: reduce_into((uint64_t*)c->host.data(), (const uint64_t*)src, count, sum); break;
            case ncclUint8: reduce_into((uint8_t*)c->host.data(), (const uint8_t*)src, count, sum); break;
            default: return fail(c, "mock_rccl: all-reduce of this type is not part of the mock");
        }
    }
    if (bytes && hipMemcpy(recv, c->host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(c, "mock_rccl: H2D copy failed");
    if (!barrier(c)) return fail(c, g_err);
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t sendcount, ncclDataType_t dt, ncclComm_t comm, hipStream_t stream) {
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t sz = dtype_size(dt);
    if (!c || !sz) { g_err = "mock_rccl: ncclAllGather: bad communicator or type"; return ncclInvalidArgument; }
    if (g_group_depth > 0) return fail(c, "mock_rccl: collectives inside a group are not part of this mock");
    const size_t bytes = sendcount * sz;
    if (bytes > SLOT_BYTES) return fail(c, "mock_rccl: all-gather larger than the mock's slot");
    // NCCL's in-place rule: a send buffer inside the receive buffer must be the rank's own chunk
    const unsigned char* s8 = (const unsigned char*)send;
    unsigned char* r8 = (unsigned char*)recv;
    if (s8 + bytes > r8 && s8 < r8 + bytes * c->world && s8 != r8 + (size_t)c->rank * bytes)
        return fail(c, "mock_rccl: in-place all-gather with sendbuff != recvbuff + rank * sendcount");
    if (hipStreamSynchronize(stream) != hipSuccess) return fail(c, "mock_rccl: hipStreamSynchronize failed");
    if (bytes && hipMemcpy(c->slots + (size_t)c->rank * SLOT_BYTES, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(c, "mock_rccl: D2H copy failed");
    const ncclResult_t a = announce(c, 2, dt, 0, sendcount);
    if (a != ncclSuccess) return a;
    for (int r = 0; r < c->world; ++r)
        if (bytes && hipMemcpy(r8 + (size_t)r * bytes, c->slots + (size_t)r * SLOT_BYTES, bytes, hipMemcpyHostToDevice) != hipSuccess)
            return fail(c, "mock_rccl: H2D copy failed");
    if (!barrier(c)) return fail(c, g_err);
    return ncclSuccess;
}

}  // extern "C"
