// bf_comm.cpp -- the one collective of the path: gathering the detected powers of the frequency shards.
// RCCL point-to-point over xGMI, behind the C-ABI (include/dsabf.h, "Multi-GPU").  One process per GPU; a bf_comm is one
// RCCL communicator + the rank's place in the frequency partition.
//
// The layout arithmetic (which floats of which rank land where) is a pure host function, bf_gather_plan /
// bf_gather_offset, so that it is tested on the CPU for any world size; the device part only walks that plan with
// grouped ncclSend / ncclRecv -- every message is received straight at its final position, there is no staging buffer
// and no re-layout pass.
//
// RCCL is bound at run time (dlopen): libdsabf.so must not drag a second HIP runtime into a process that already has
// one (torch bundles its own librccl.so + libamdhip64.so; a C++ application links ROCm's) -- the same rule as for the HIP
// runtime itself.  Resolution order: $DSABF_RCCL_LIB, a librccl already loaded in the process,
// librccl.so.1, librccl.so.
#include "../../include/dsabf.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "bf_host_internal.h"
#include "bf_kernels.h"

namespace {

// The few RCCL declarations this file needs (rccl/rccl.h:40-43,187,220,260,339,466,700,722,923,929), bound with dlsym.
struct rccl_unique_id {
    char internal[128];
};
typedef void* rccl_comm_t;
constexpr int kNcclFloat32 = 7;
struct rccl_api {
    void* lib = nullptr;
    int (*GetUniqueId)(rccl_unique_id*) = nullptr;
    int (*CommInitRank)(rccl_comm_t*, int, rccl_unique_id, int) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*GetVersion)(int*) = nullptr;                 // optional (reports only)
    int (*CommCount)(rccl_comm_t, int*) = nullptr;     // optional: how many ranks the LIBRARY says the communicator has
    std::string error;
};

rccl_api& rccl()
{
    static rccl_api api;
    if (api.lib || !api.error.empty()) return api;
    // $DSABF_RCCL_LIB wins outright (Python callers get it set to the librccl next to the HIP runtime they preloaded,
    // dsabeamformer_amd/_lib.py; tests point it at a loopback stand-in); otherwise a copy that is already in the process,
    // then the system's.
    if (const char* e = getenv("DSABF_RCCL_LIB")) {
        if (e[0]) api.lib = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
        if (e[0] && !api.lib) {
            const char* why = dlerror();   // once: the call clears the error, a second one returns NULL
            api.error = std::string("DSABF_RCCL_LIB=") + e + " could not be loaded: " + (why ? why : "?");
            return api;
        }
    }
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (int pass = 0; pass < 2 && !api.lib; pass++)      // pass 0: only a copy that is already in the process
        for (const char* n : names) {
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (api.lib) break;
        }
    if (!api.lib) {
        const char* why = dlerror();
        api.error = std::string("librccl could not be loaded: ") + (why ? why : "not found");
        return api;
    }
#define BIND(field, sym)                                                            \
    *reinterpret_cast<void**>(&api.field) = dlsym(api.lib, sym);                     \
    if (!api.field) api.error = std::string("librccl lacks ") + sym;
    BIND(GetUniqueId, "ncclGetUniqueId")
    BIND(CommInitRank, "ncclCommInitRank")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(GetErrorString, "ncclGetErrorString")
    BIND(GroupStart, "ncclGroupStart")
    BIND(GroupEnd, "ncclGroupEnd")
    BIND(Send, "ncclSend")
    BIND(Recv, "ncclRecv")
#undef BIND
    *reinterpret_cast<void**>(&api.GetVersion) = dlsym(api.lib, "ncclGetVersion");
    *reinterpret_cast<void**>(&api.CommCount) = dlsym(api.lib, "ncclCommCount");
    return api;
}

int comm_fail(int code, const std::string& msg) { return dsabf::set_error(code, msg.c_str()); }

}  // namespace

struct bf_comm {
    int rank = 0, world = 1, device = 0;
    int n_cus = 0;   // of `device`, asked for when the staged transport first needs it
    rccl_comm_t comm = nullptr;
};

extern "C" {

size_t bf_gather_offset(int layout, size_t n_rows, size_t row_floats, int world, int rank, size_t row)
{
    if (layout == BF_GATHER_LAYOUT_RANK_MAJOR) return ((size_t)rank * n_rows + row) * row_floats;
    return (row * (size_t)world + (size_t)rank) * row_floats;   // [row][rank][row_floats] == [o][f][b], f = rank * F/R + f_local
}

// owner of gathered row `row`: the root, everybody (-1: returned as -1), or -- distributed -- rank row / (n_rows / world)
static int row_owner(size_t n_rows, int world, int root, size_t row)
{
    if (root == BF_GATHER_ROOT_DISTRIBUTED) return (int)(row / (n_rows / (size_t)world));
    return root;
}

size_t bf_gather_rows_held(size_t n_rows, int world, int rank, int root)
{
    if (world <= 0 || rank < 0 || rank >= world) return 0;
    if (root == BF_GATHER_ROOT_DISTRIBUTED) return n_rows % (size_t)world ? 0 : n_rows / (size_t)world;
    return (root < 0 || root == rank) ? n_rows : 0;
}

size_t bf_gather_plan(int layout, size_t n_rows, size_t row_floats, int world, int rank, int root, bf_gather_msg* msgs,
                      size_t capacity)
{
    // Messages this rank takes part in, in issue order (ascending row, then ascending receiver, then ascending sender:
    // the same order on every rank, which is what matches sends with receives between a pair of ranks).
    // A receiver holds `held` rows [first, first + held) of every sender; its array is [held][world][row_floats]
    // (freq-major) or [world][held][row_floats] (rank-major).  Rows that are contiguous on both sides travel as one
    // message: rank-major = one message per (sender, receiver); freq-major = one message per (row, sender).
    if (world <= 0 || rank < 0 || rank >= world || root >= world || root < BF_GATHER_ROOT_DISTRIBUTED) return 0;
    if (root == BF_GATHER_ROOT_DISTRIBUTED && n_rows % (size_t)world) return 0;
    const size_t held = root == BF_GATHER_ROOT_DISTRIBUTED ? n_rows / (size_t)world : n_rows;   // rows per receiver
    const size_t run_rows = layout == BF_GATHER_LAYOUT_RANK_MAJOR ? held : 1;
    size_t n = 0;
    auto emit = [&](int kind, int peer, size_t local_off, size_t full_off, size_t count) {
        if (msgs && n < capacity) msgs[n] = bf_gather_msg{kind, peer, local_off, full_off, count};
        n++;
    };
    for (size_t row0 = 0; row0 < n_rows; row0 += run_rows) {
        const size_t count = run_rows * row_floats;
        const int owner = row_owner(n_rows, world, root, row0);
        const size_t first = root == BF_GATHER_ROOT_DISTRIBUTED ? (size_t)owner * held : 0;   // first row the owner holds
        for (int dst = 0; dst < world; dst++) {
            if (owner >= 0 && dst != owner) continue;
            if (rank == dst) {                                // this rank receives from every rank, itself included
                for (int src = 0; src < world; src++) {
                    const size_t off = bf_gather_offset(layout, held, row_floats, world, src, row0 - first);
                    emit(src == rank ? BF_GATHER_COPY : BF_GATHER_RECV, src, row0 * row_floats, off, count);
                }
            } else {
                emit(BF_GATHER_SEND, dst, row0 * row_floats, 0, count);
            }
        }
    }
    return n;
}

int bf_comm_unique_id(void* id128)
{
    if (!id128) return comm_fail(BF_ERR_INVALID, "id buffer is NULL");
    rccl_api& r = rccl();
    if (!r.error.empty()) return comm_fail(BF_ERR_DEVICE, r.error);
    rccl_unique_id id;
    const int rc = r.GetUniqueId(&id);
    if (rc != 0) return comm_fail(BF_ERR_DEVICE, std::string("ncclGetUniqueId: ") + r.GetErrorString(rc));
    memcpy(id128, &id, sizeof id);
    return BF_OK;
}

int bf_comm_create(int rank, int world, const void* id128, int device, bf_comm** out)
{
    if (!out) return comm_fail(BF_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return comm_fail(BF_ERR_INVALID, "need 0 <= rank < world");
    bf_comm* c = new (std::nothrow) bf_comm();
    if (!c) return comm_fail(BF_ERR_DEVICE, "out of host memory");
    c->rank = rank;
    c->world = world;
    c->device = device;
    if (world > 1 || id128) {   // a unique id with world == 1 still builds a real (one-rank) RCCL communicator
        if (!id128) {
            delete c;
            return comm_fail(BF_ERR_INVALID, "a unique id (bf_comm_unique_id of rank 0) is needed when world > 1");
        }
        rccl_api& r = rccl();
        if (!r.error.empty()) {
            delete c;
            return comm_fail(BF_ERR_DEVICE, r.error);
        }
        int prev = -1;
        (void)hipGetDevice(&prev);
        if (hipSetDevice(device) != hipSuccess) {
            delete c;
            return comm_fail(BF_ERR_DEVICE, "hipSetDevice failed");
        }
        rccl_unique_id id;
        memcpy(&id, id128, sizeof id);
        const int rc = r.CommInitRank(&c->comm, world, id, rank);
        if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
        if (rc != 0) {
            delete c;
            return comm_fail(BF_ERR_DEVICE, std::string("ncclCommInitRank: ") + r.GetErrorString(rc));
        }
    }
    *out = c;
    return BF_OK;
}

int bf_comm_destroy(bf_comm* c)
{
    if (!c) return BF_OK;
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
    return BF_OK;
}

// What a scaling record must be able to prove: which library carried the gather, its version, and how many ranks the
// LIBRARY (not the caller) says the communicator spans.  lib_path = the file ncclSend was resolved from (dladdr).
int bf_comm_info(const bf_comm* c, int* lib_ranks, int* version, char* lib_path, size_t n)
{
    if (!c) return comm_fail(BF_ERR_INVALID, "comm is NULL");
    if (lib_ranks) *lib_ranks = c->comm ? -1 : 0;      // 0: no RCCL communicator behind this bf_comm (world 1 without an id)
    if (version) *version = 0;
    if (lib_path && n) lib_path[0] = 0;
    if (!c->comm) return BF_OK;
    rccl_api& r = rccl();
    if (!r.error.empty()) return comm_fail(BF_ERR_DEVICE, r.error);
    if (lib_ranks && r.CommCount) {
        int cnt = -1;
        if (r.CommCount(c->comm, &cnt) == 0) *lib_ranks = cnt;
    }
    if (version && r.GetVersion) (void)r.GetVersion(version);
    Dl_info di{};
    if (lib_path && n && dladdr(reinterpret_cast<void*>(r.Send), &di) && di.dli_fname) snprintf(lib_path, n, "%s", di.dli_fname);
    return BF_OK;
}

// The same evidence BEFORE a communicator exists (a diagnostic line must be able to name the library whose ncclCommInitRank never
// came back): binds librccl as bf_comm_create would -- dlopen only, no device, no network -- and reports its version and file.
int bf_comm_library_info(int* version, char* lib_path, size_t n)
{
    if (version) *version = 0;
    if (lib_path && n) lib_path[0] = 0;
    rccl_api& r = rccl();
    if (!r.error.empty()) return comm_fail(BF_ERR_DEVICE, r.error);
    if (version && r.GetVersion) (void)r.GetVersion(version);
    Dl_info di{};
    if (lib_path && n && dladdr(reinterpret_cast<void*>(r.Send), &di) && di.dli_fname) snprintf(lib_path, n, "%s", di.dli_fname);
    return BF_OK;
}

int bf_comm_rank(const bf_comm* c) { return c ? c->rank : BF_ERR_INVALID; }
int bf_comm_world(const bf_comm* c) { return c ? c->world : BF_ERR_INVALID; }

// The exchange itself.  Point-to-point messages follow the plan of `wire_layout` and are received into d_wire; the rank's own
// rows are copied (no RCCL) to their place in d_self, which is laid out as `self_layout`.  In-place transport: both are the
// caller's d_full in the layout it asked for.  Staged transport: the wire is rank-major into the staging area, the own rows go
// straight to their freq-major place in d_full.
static int gather_exchange(bf_comm* c, const float* d_local, size_t n_rows, size_t row_floats, int root, int wire_layout,
                           float* d_wire, int self_layout, float* d_self, hipStream_t s)
{
    const size_t held = bf_gather_rows_held(n_rows, c->world, c->rank, root);
    const bool receives = held > 0;
    const size_t n = bf_gather_plan(wire_layout, n_rows, row_floats, c->world, c->rank, root, nullptr, 0);
    std::vector<bf_gather_msg> plan(n);
    bf_gather_plan(wire_layout, n_rows, row_floats, c->world, c->rank, root, plan.data(), n);
    int result = BF_OK;
    // own rows: a strided device-to-device copy (one call for the freq-major layout), not through RCCL -- unless
    // DSABF_GATHER_SELF_RCCL=1 asks for it (test switch: lets ONE GPU exercise the grouped ncclSend / ncclRecv path)
    const char* self_env = dsabf::lab_getenv("DSABF_GATHER_SELF_RCCL");
    const bool self_rccl = c->comm && self_env && self_env[0] == '1' && d_wire == d_self && wire_layout == self_layout;
    if (!self_rccl && receives) {
        const size_t first = root == BF_GATHER_ROOT_DISTRIBUTED ? (size_t)c->rank * held : 0;   // my own rows that I keep
        const size_t off0 = bf_gather_offset(self_layout, held, row_floats, c->world, c->rank, 0);
        const float* src = d_local + first * row_floats;
        hipError_t e;
        if (self_layout == BF_GATHER_LAYOUT_RANK_MAJOR)
            e = hipMemcpyAsync(d_self + off0, src, held * row_floats * sizeof(float), hipMemcpyDeviceToDevice, s);
        else
            e = hipMemcpy2DAsync(d_self + off0, (size_t)c->world * row_floats * sizeof(float), src, row_floats * sizeof(float),
                                 row_floats * sizeof(float), held, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) result = comm_fail(BF_ERR_DEVICE, std::string("gather self-copy: ") + hipGetErrorString(e));
    }
    if (result == BF_OK && (c->world > 1 || self_rccl)) {
        rccl_api& r = rccl();
        // One ncclGroup per range of rows, the SAME ranges on every rank (symmetric groups: each group is a complete
        // exchange of its rows), sized so that a group stays below ~2048 messages on the busiest rank.
        const size_t rows_per_group = wire_layout == BF_GATHER_LAYOUT_RANK_MAJOR ? n_rows : (2048 / (size_t)c->world ? 2048 / (size_t)c->world : 1);
        size_t group_end_row = 0;
        bool open = false;
        auto close_group = [&]() {
            if (!open) return 0;
            open = false;
            return r.GroupEnd();
        };
        for (const bf_gather_msg& m : plan) {
            if (m.kind == BF_GATHER_COPY && !self_rccl) continue;
            const size_t row = m.local_offset / row_floats;
            if (open && row >= group_end_row) {
                const int rc2 = close_group();
                if (rc2 != 0) {
                    result = comm_fail(BF_ERR_DEVICE, std::string("ncclGroupEnd: ") + r.GetErrorString(rc2));
                    break;
                }
            }
            if (!open) {
                int rc = r.GroupStart();
                if (rc != 0) {
                    result = comm_fail(BF_ERR_DEVICE, std::string("ncclGroupStart: ") + r.GetErrorString(rc));
                    break;
                }
                open = true;
                group_end_row = (row / rows_per_group + 1) * rows_per_group;
            }
            int rc = 0;
            if (m.kind == BF_GATHER_SEND || m.kind == BF_GATHER_COPY)
                rc = r.Send(d_local + m.local_offset, m.count, kNcclFloat32, m.peer, c->comm, s);
            if (rc == 0 && (m.kind == BF_GATHER_RECV || m.kind == BF_GATHER_COPY))
                rc = r.Recv(d_wire + m.full_offset, m.count, kNcclFloat32, m.peer, c->comm, s);
            if (rc != 0) {
                (void)close_group();
                result = comm_fail(BF_ERR_DEVICE, std::string("ncclSend/Recv: ") + r.GetErrorString(rc));
                break;
            }
        }
        if (result == BF_OK) {
            const int rc = close_group();
            if (rc != 0) result = comm_fail(BF_ERR_DEVICE, std::string("ncclGroupEnd: ") + r.GetErrorString(rc));
        }
    }
    return result;
}

static int gather_check(bf_comm* c, const float* d_local, size_t n_rows, int root)
{
    if (!c || !d_local) return comm_fail(BF_ERR_INVALID, "NULL argument");
    if (root >= c->world || root < BF_GATHER_ROOT_DISTRIBUTED) return comm_fail(BF_ERR_INVALID, "root out of range");
    if (root == BF_GATHER_ROOT_DISTRIBUTED && n_rows % (size_t)c->world)
        return comm_fail(BF_ERR_INVALID, "distributed owners need n_rows divisible by the number of ranks");
    return BF_OK;
}

int bf_gather_detected(bf_comm* c, const float* d_local, size_t n_rows, size_t row_floats, int root, int layout,
                       float* d_full, void* hip_stream)
{
    if (int rc = gather_check(c, d_local, n_rows, root)) return rc;
    if (layout != BF_GATHER_LAYOUT_FREQ_MAJOR && layout != BF_GATHER_LAYOUT_RANK_MAJOR)
        return comm_fail(BF_ERR_INVALID, "unknown gather layout");
    if (bf_gather_rows_held(n_rows, c->world, c->rank, root) > 0 && !d_full)
        return comm_fail(BF_ERR_INVALID, "this rank receives: d_full must not be NULL");
    if (n_rows == 0 || row_floats == 0) return BF_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != c->device && hipSetDevice(c->device) != hipSuccess) return comm_fail(BF_ERR_DEVICE, "hipSetDevice failed");
    const int result = gather_exchange(c, d_local, n_rows, row_floats, root, layout, d_full, layout, d_full, static_cast<hipStream_t>(hip_stream));
    if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
    return result;
}

int bf_gather_detected_staged(bf_comm* c, const float* d_local, size_t n_rows, size_t row_floats, int root, float* d_full,
                              float* d_stage, void* hip_stream)
{
    if (int rc = gather_check(c, d_local, n_rows, root)) return rc;
    const size_t held = bf_gather_rows_held(n_rows, c->world, c->rank, root);
    if (held > 0 && (!d_full || (!d_stage && c->world > 1)))
        return comm_fail(BF_ERR_INVALID, "this rank receives: d_full and d_stage must not be NULL");
    if (held > 0 && (row_floats % 4 || ((uintptr_t)d_full & 15) || ((uintptr_t)d_stage & 15)))
        return comm_fail(BF_ERR_INVALID, "the staged transport moves 16-byte pieces: row_floats must be a multiple of 4, d_full and "
                                         "d_stage 16-byte aligned");
    if (n_rows == 0 || row_floats == 0) return BF_OK;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != c->device && hipSetDevice(c->device) != hipSuccess) return comm_fail(BF_ERR_DEVICE, "hipSetDevice failed");
    int result = gather_exchange(c, d_local, n_rows, row_floats, root, BF_GATHER_LAYOUT_RANK_MAJOR, d_stage, BF_GATHER_LAYOUT_FREQ_MAJOR,
                                 d_full, s);
    if (result == BF_OK && held > 0 && c->world > 1) {
        if (!c->n_cus) {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || n <= 0) n = 256;
            c->n_cus = n;
        }
        const hipError_t e = dsabf::launch_gather_relayout(d_stage, d_full, held, c->world, row_floats, c->rank, c->n_cus, s);
        if (e != hipSuccess) result = comm_fail(BF_ERR_DEVICE, std::string("gather re-layout launch: ") + hipGetErrorString(e));
    }
    if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
    return result;
}

}  // extern "C"

namespace dsabf {
int comm_all_ok(bf_comm* c, bool ok, bool* all)
{
    *all = ok;
    if (!c || c->world == 1) return BF_OK;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != c->device && hipSetDevice(c->device) != hipSuccess) return comm_fail(BF_ERR_DEVICE, "hipSetDevice failed");
    float* d = nullptr;   // [0, 4): this rank's flag (one 16-byte row), [4, 4 + 4 world): everybody's
    int rc = BF_OK;
    std::vector<float> got((size_t)4 * c->world, 0.0f);
    const float mine[4] = {ok ? 1.0f : 0.0f, 0.0f, 0.0f, 0.0f};
    hipError_t e = hipMalloc((void**)&d, sizeof(float) * (4 + got.size()));
    if (e == hipSuccess) e = hipMemcpy(d, mine, sizeof mine, hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = comm_fail(BF_ERR_DEVICE, std::string("comm_all_ok: ") + hipGetErrorString(e));
    // (a rank whose device calls failed still takes part if it can: its flag is false either way)
    if (d) {
        const int g = bf_gather_detected(c, d, 1, 4, BF_GATHER_ROOT_ALL, BF_GATHER_LAYOUT_RANK_MAJOR, d + 4, nullptr);
        if (g != BF_OK) rc = g;
        if (rc == BF_OK && (e = hipMemcpy(got.data(), d + 4, sizeof(float) * got.size(), hipMemcpyDeviceToHost)) != hipSuccess)
            rc = comm_fail(BF_ERR_DEVICE, std::string("comm_all_ok: ") + hipGetErrorString(e));
        (void)hipFree(d);
    }
    bool every = rc == BF_OK;
    for (int r = 0; every && r < c->world; r++) every = got[(size_t)4 * r] == 1.0f;
    *all = every;
    if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
    return rc;
}
}  // namespace dsabf
