// fake_rccl.cpp -- TEST INFRASTRUCTURE ONLY: a loopback stand-in for the eight RCCL entry points bf_comm.cpp binds
// (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclGetErrorString, ncclGroupStart, ncclGroupEnd, ncclSend, ncclRecv),
// so that SEVERAL ranks can time-share ONE GPU: RCCL itself refuses two ranks on one device, and the build pool has
// 1-GPU boxes only.  Messages travel through single-producer / single-consumer byte rings in a POSIX shared-memory
// segment (one ring per ordered pair of ranks), staged through host memory; the point-to-point semantics the product
// relies on are kept: messages between one (sender, receiver) pair match in issue order, the operations of a group
// progress concurrently (a rank may send and receive big messages in one group without deadlock), sizes must agree.
// Unlike RCCL everything completes inside ncclGroupEnd (the stream is synchronised first) -- fine for checking WHERE
// every float lands, useless for timing.  Selected with DSABF_RCCL_LIB=<this .so>; never shipped, never linked.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr int kMaxRanks = 8;
struct ring_hdr {
    std::atomic<uint64_t> head;  // bytes written
    std::atomic<uint64_t> tail;  // bytes read
    char pad[48];
};
struct seg_hdr {
    std::atomic<int> joined;
    std::atomic<int> left;
    uint64_t ring_bytes;
    int nranks;
    char pad[40];
};

struct fake_comm {
    int rank = 0, nranks = 1;
    std::string name;
    size_t map_bytes = 0;
    char* base = nullptr;
    seg_hdr* hdr() const { return reinterpret_cast<seg_hdr*>(base); }
    ring_hdr* ring(int src, int dst) const
    {
        return reinterpret_cast<ring_hdr*>(base + sizeof(seg_hdr) + (size_t)(src * nranks + dst) * (sizeof(ring_hdr) + hdr()->ring_bytes));
    }
    char* ring_data(int src, int dst) const { return reinterpret_cast<char*>(ring(src, dst)) + sizeof(ring_hdr); }
};

struct op {
    bool send;
    void* dev;
    size_t bytes;
    int peer;
    fake_comm* comm;
    hipStream_t stream;
    std::vector<char> host;
    size_t done = 0;
};
thread_local int g_depth = 0;
thread_local std::vector<op> g_ops;

size_t dtype_bytes(int dt) { return (dt == 0 || dt == 1) ? 1 : (dt == 6 || dt == 9) ? 2 : (dt == 4 || dt == 5 || dt == 8) ? 8 : 4; }

int run_ops()
{
    if (g_ops.empty()) return 0;
    // everything the stream was asked to do before the group must be visible: synchronise, then stage the sends
    for (op& o : g_ops)
        if (hipStreamSynchronize(o.stream) != hipSuccess) return 1;
    for (op& o : g_ops) {
        o.host.resize(o.bytes);
        if (o.send && hipMemcpy(o.host.data(), o.dev, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    }
    const auto t0 = std::chrono::steady_clock::now();
    size_t remaining = g_ops.size();
    while (remaining) {
        bool progress = false;
        bool send_busy[kMaxRanks] = {false}, recv_busy[kMaxRanks] = {false};  // one message at a time per direction and peer, in order
        for (op& o : g_ops) {
            if (o.done == o.bytes) continue;
            bool* busy = o.send ? send_busy : recv_busy;
            if (busy[o.peer]) continue;
            busy[o.peer] = true;
            fake_comm* c = o.comm;
            const uint64_t cap = c->hdr()->ring_bytes;
            ring_hdr* r = o.send ? c->ring(c->rank, o.peer) : c->ring(o.peer, c->rank);
            char* data = o.send ? c->ring_data(c->rank, o.peer) : c->ring_data(o.peer, c->rank);
            // every message is framed by its 8-byte length so that a size mismatch is an error, not silent corruption
            if (o.send) {
                uint64_t head = r->head.load(std::memory_order_relaxed), tail = r->tail.load(std::memory_order_acquire);
                uint64_t free_b = cap - (head - tail);
                if (o.done == 0 && o.host.size() == o.bytes) {   // frame header first (once)
                    if (free_b < 8) continue;
                    uint64_t len = o.bytes;
                    for (int i = 0; i < 8; i++) data[(head + i) % cap] = reinterpret_cast<char*>(&len)[i];
                    head += 8;
                    free_b -= 8;
                    o.host.push_back(0);   // marks "header written" (size != bytes from now on)
                    r->head.store(head, std::memory_order_release);
                    progress = true;
                }
                const size_t n = (size_t)std::min<uint64_t>(free_b, o.bytes - o.done);
                for (size_t i = 0; i < n;) {
                    const size_t pos = (size_t)((head + i) % cap), run = std::min(n - i, (size_t)cap - pos);
                    memcpy(data + pos, o.host.data() + o.done + i, run);
                    i += run;
                }
                if (n) {
                    o.done += n;
                    r->head.store(head + n, std::memory_order_release);
                    progress = true;
                }
            } else {
                uint64_t tail = r->tail.load(std::memory_order_relaxed), head = r->head.load(std::memory_order_acquire);
                uint64_t avail = head - tail;
                if (o.done == 0 && o.host.size() == o.bytes) {
                    if (avail < 8) continue;
                    uint64_t len = 0;
                    for (int i = 0; i < 8; i++) reinterpret_cast<char*>(&len)[i] = data[(tail + i) % cap];
                    if (len != o.bytes) {
                        fprintf(stderr, "fake_rccl: rank %d expected %zu bytes from rank %d, the sender posted %llu\n", c->rank,
                                o.bytes, o.peer, (unsigned long long)len);
                        return 2;
                    }
                    tail += 8;
                    avail -= 8;
                    o.host.push_back(0);
                    r->tail.store(tail, std::memory_order_release);
                    progress = true;
                }
                const size_t n = (size_t)std::min<uint64_t>(avail, o.bytes - o.done);
                for (size_t i = 0; i < n;) {
                    const size_t pos = (size_t)((tail + i) % cap), run = std::min(n - i, (size_t)cap - pos);
                    memcpy(o.host.data() + o.done + i, data + pos, run);
                    i += run;
                }
                if (n) {
                    o.done += n;
                    r->tail.store(tail + n, std::memory_order_release);
                    progress = true;
                }
            }
            if (o.done == o.bytes) remaining--;
        }
        if (!progress) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                fprintf(stderr, "fake_rccl: no progress for 120 s (a send without its receive?)\n");
                return 3;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
    }
    // FAKERCCL_CORRUPT=1: a fabric that delivers wrong bits -- one bit of the LAST message this group received is flipped
    // (tests of bench.py's checksum verification: the gather must be caught, not believed)
    if (const char* e = getenv("FAKERCCL_CORRUPT"))
        if (e[0] == '1')
            for (size_t k = g_ops.size(); k-- > 0;)
                if (!g_ops[k].send && g_ops[k].bytes >= 8) {
                    g_ops[k].host[g_ops[k].bytes / 2] ^= 0x10;
                    break;
                }
    for (op& o : g_ops)
        if (!o.send && hipMemcpy(o.dev, o.host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
    g_ops.clear();
    return 0;
}

}  // namespace

extern "C" {

struct fake_id {
    char internal[128];
};

int ncclGetUniqueId(fake_id* id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/dsabf_fakerccl_%d_%lld", (int)getpid(),
             (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return 0;
}

int ncclCommInitRank(void** comm, int nranks, fake_id id, int rank)
{
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;
    fake_comm* c = new fake_comm();
    c->rank = rank;
    c->nranks = nranks;
    c->name = id.internal;
    const char* mb = getenv("FAKERCCL_MAILBOX_MB");
    const uint64_t ring_bytes = (uint64_t)(mb ? atoi(mb) : 16) << 20;
    c->map_bytes = sizeof(seg_hdr) + (size_t)nranks * nranks * (sizeof(ring_hdr) + ring_bytes);
    int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) return 5;   // fresh segments are zero-filled: counters start at 0
    c->base = static_cast<char*>(mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
    close(fd);
    if (c->base == MAP_FAILED) return 5;
    c->hdr()->ring_bytes = ring_bytes;   // every rank writes the same values
    c->hdr()->nranks = nranks;
    c->hdr()->joined.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->hdr()->joined.load() < nranks) {   // ncclCommInitRank is a collective
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return 6;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    *comm = c;
    return 0;
}

// reports only (bf_comm_info): the ranks that actually JOINED the segment, and version 0 = "not RCCL"
int ncclCommCount(void* comm, int* count)
{
    fake_comm* c = static_cast<fake_comm*>(comm);
    if (!c || !count) return 4;
    *count = (int)c->hdr()->joined.load();
    return 0;
}

int ncclGetVersion(int* v)
{
    if (v) *v = 0;
    return 0;
}

int ncclCommDestroy(void* comm)
{
    fake_comm* c = static_cast<fake_comm*>(comm);
    if (!c) return 0;
    const bool last = c->hdr()->left.fetch_add(1) + 1 == c->nranks;
    munmap(c->base, c->map_bytes);
    if (last) shm_unlink(c->name.c_str());
    delete c;
    return 0;
}

const char* ncclGetErrorString(int code)
{
    static const char* names[] = {"ok", "HIP error", "message size mismatch", "timeout", "bad rank", "shared memory", "init timeout"};
    return (code >= 0 && code <= 6) ? names[code] : "unknown";
}

int ncclGroupStart()
{
    g_depth++;
    return 0;
}

int ncclGroupEnd()
{
    if (--g_depth > 0) return 0;
    g_depth = 0;
    return run_ops();
}

static int post(bool send, void* buf, size_t count, int dt, int peer, void* comm, hipStream_t s)
{
    fake_comm* c = static_cast<fake_comm*>(comm);
    if (!c || peer < 0 || peer >= c->nranks) return 4;
    op o;
    o.send = send;
    o.dev = buf;
    o.bytes = count * dtype_bytes(dt);
    o.peer = peer;
    o.comm = c;
    o.stream = s;
    g_ops.push_back(std::move(o));
    return g_depth > 0 ? 0 : run_ops();
}

int ncclSend(const void* buf, size_t count, int dt, int peer, void* comm, hipStream_t s)
{
    // FAKERCCL_HANG_SEND=1: a fabric on which the first send never returns (tests of bench.py's deadline)
    if (const char* e = getenv("FAKERCCL_HANG_SEND"))
        if (e[0] == '1')
            for (;;) std::this_thread::sleep_for(std::chrono::seconds(1));
    return post(true, const_cast<void*>(buf), count, dt, peer, comm, s);
}
int ncclRecv(void* buf, size_t count, int dt, int peer, void* comm, hipStream_t s) { return post(false, buf, count, dt, peer, comm, s); }

}  // extern "C"
