// bf_runtime.cpp -- the C-ABI of libdsabf.so (include/dsabf.h): handle, device memory, HIP queues and events.
//
// This is the thin host layer the reference keeps inline in main() (src/beamformer.cu:159-320,560-618):
// one transfer queue + n_streams compute queues, a device ring of n_blocks_on_gpu PSRDADA-sized blocks, one
// detected-power buffer per compute queue.  There is no CPU fallback: without a gfx950 device every compute
// entry point fails with BF_ERR_NO_DEVICE / BF_ERR_DEVICE.
#include "../../include/dsabf.h"
#include "../../include/dsabf_bench.h"

#include <cstdlib>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "bf_kernels.h"
#include "bf_host_internal.h"

struct bf_event {
    hipEvent_t ev = nullptr;
    bool recorded = false;
};

struct bf_handle {
    bf_config cfg{};
    dsabf::Geometry geom{};
    int device = 0;
    int n_cus = 256;
    bool weights_set = false;
    void* d_wimage = nullptr;     // MFMA fragment image of the weights
    void* d_wimage_p = nullptr;   // conjugate-pair image (geometries with a paired kernel)
    int* d_flag = nullptr;        // relayout flags: [0] weight out of range, [1] weights are not conjugate-paired
    uint8_t* d_data = nullptr;    // ring: n_blocks_on_gpu x bytes_per_block
    float* d_out = nullptr;       // n_streams x floats_per_detect
    float* d_ded = nullptr;       // n_streams x n_beams
    // Scratch of the DM-trial dedispersion, ONE PER STREAM the caller has used (kDmScratchBytes each: which trial groups the
    // wide kernel takes + a 512-byte row of zeros).  The wide kernel writes the flags and the per-thread-window kernel reads
    // them later on the same stream: calls on one stream are ordered by the stream, calls on different streams must not share.
    std::vector<std::pair<hipStream_t, int*>> dm_scratch;
    bool force_general = false;   // bf_set_switch("paired", 0): never select the conjugate-pair kernel
    bool dm_ring = true;          // bf_set_switch("dm_ring", 0): the next bf_dm_stream_create takes the linear buffer (test switch)
    // bf_enqueue_gemm_unit coalesces (see flush_units): the caller keeps the reference's one-unit-per-call loop
    // (src/beamformer.cu:454-519), the device sees one launch per run of consecutive gemm-units.
    struct pending_unit {
        int stream_idx, slot, time_slice;
        float* host_out;      // a4: D2H destination of the unit's detected powers (NULL: none)
        float* ded_row;       // a8: D2H destination of its DM-0 row (bf_enqueue_dedisperse after the unit), NULL: none
        bool ded;
    };
    std::vector<pending_unit> pending;
    bool coalesce = true;         // DSABF_COALESCE=0 / bf_set_switch("coalesce", 0): one launch per call, the literal pattern
    uint64_t flush_seq = 0;       // flushes alternate between the first two compute queues
    hipEvent_t flush_done = nullptr;   // end of the previous flush's host copies: the next flush's copies queue behind it
    bool flush_recorded = false;
    uint64_t n_fused_launches = 0;        // fused-kernel launches this handle has issued (bf_get_counter)
    std::vector<const float*> last_out;   // per caller-visible queue: where its most recent gemm-unit's powers are on the device ...
    std::vector<int> last_q;              // ... and the queue that wrote them
    std::vector<float*> d_out_blk;  // per compute queue, n_gemms_per_block x floats_per_detect: bf_enqueue_block (lazy)
    std::vector<char> blk_ran;      // per compute queue: bf_enqueue_block has launched into d_out_blk[q]
    std::vector<float*> d_ded_blk;  // per compute queue, n_gemms_per_block x n_beams: bf_enqueue_block_dedisperse (lazy)
    std::vector<float*> d_full_blk; // per compute queue, the gathered block (world x as large): bf_block_gather_device (lazy)
    std::vector<float*> d_stage_blk; // per compute queue, the staged transport's landing area: bf_block_gather_stage_device (lazy)
    int full_world = 0;
    std::vector<struct bf_dm_stream*> dm_streams;   // DM stages created on this handle: bf_destroy releases their device memory
    hipStream_t h2d = nullptr;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> join;  // one per compute queue, for bf_record_analysis_event
    hipEvent_t t0 = nullptr, t1 = nullptr;
};

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(BF_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int check_cfg(const bf_config* c)
{
    if (!c) return fail(BF_ERR_INVALID, "config is NULL");
    if (c->n_beams <= 0 || c->n_ant <= 0 || c->n_freq <= 0 || c->n_pol <= 0 || c->n_avg <= 0 ||
        c->n_out_per_gemm <= 0 || c->n_gemms_per_block <= 0 || c->n_blocks_on_gpu <= 0 || c->n_streams <= 0)
        return fail(BF_ERR_INVALID, "every geometry field must be positive");
    if (c->n_beams % 4) return fail(BF_ERR_INVALID, "N_BEAMS must be divisible by 4");       // src/beamformer.hh:155
    if (c->n_ant % 4) return fail(BF_ERR_INVALID, "N_ANTENNAS must be divisible by 4");      // src/beamformer.hh:156
    if (c->detect_mode != BF_DETECT_CANONICAL && c->detect_mode != BF_DETECT_FAST && c->detect_mode != BF_DETECT_CONTRACTED)
        return fail(BF_ERR_INVALID, "detect_mode must be BF_DETECT_CANONICAL, BF_DETECT_CONTRACTED or BF_DETECT_FAST");
    return BF_OK;
}

dsabf::Geometry make_geom(const bf_config& c)
{
    dsabf::Geometry g{};
    g.n_beams = c.n_beams;
    g.n_ant = c.n_ant;
    g.n_freq = c.n_freq;
    g.n_ipo = c.n_pol * c.n_avg;
    g.n_out = c.n_out_per_gemm;
    g.n_time = g.n_out * g.n_ipo;
    g.n_ctiles = (c.n_beams + 15) / 16;
    g.fast_detect = c.detect_mode == BF_DETECT_FAST;
    g.contracted_detect = c.detect_mode == BF_DETECT_CONTRACTED;
    dsabf::read_env_switches(g);
    return g;
}

hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

// Makes `device` current for the duration of one entry point and puts the caller's device back afterwards: a library
// call must not change the current device of a multi-device host process (torch's included).
struct DeviceScope {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) err = hipSetDevice(device);
    }
    ~DeviceScope()
    {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};
#define ON_DEVICE(h_)                                                                                       \
    DeviceScope dev_scope_((h_)->device);                                                                   \
    if (dev_scope_.err != hipSuccess)                                                                       \
        return fail(BF_ERR_DEVICE, "hipSetDevice(%d) failed: %s", (h_)->device, hipGetErrorString(dev_scope_.err))

}  // namespace

namespace dsabf {
int set_error(int code, const char* msg)
{
    g_err = msg ? msg : "";
    return code;
}
}  // namespace dsabf

extern "C" {

const char* bf_last_error(void) { return g_err.c_str(); }
#ifndef DSABF_KERNEL_BUILD_ID
#define DSABF_KERNEL_BUILD_ID "unknown"   // build.py: a hash over the kernel sources and their flags (build.kernel_build_id())
#endif
const char* bf_version(void) { return "dsabf 0.3 (gfx950, fused expand+int8 MFMA+detect; kernels " DSABF_KERNEL_BUILD_ID ")"; }

int bf_config_default(bf_config* cfg, int debug)
{
    if (!cfg) return fail(BF_ERR_INVALID, "config is NULL");
    cfg->n_beams = 256;
    cfg->n_ant = 64;
    cfg->n_freq = 256;
    cfg->n_pol = 2;
    cfg->n_avg = debug ? 1 : 16;
    cfg->n_out_per_gemm = 8;
    cfg->n_gemms_per_block = 32;
    cfg->n_blocks_on_gpu = 8;
    cfg->n_streams = 8;
    cfg->verbose = 0;
    cfg->detect_mode = BF_DETECT_CANONICAL;
    return BF_OK;
}

int bf_n_inputs_per_output(const bf_config* c) { return c ? c->n_pol * c->n_avg : 0; }
int bf_n_timesteps_per_gemm(const bf_config* c) { return c ? c->n_out_per_gemm * c->n_pol * c->n_avg : 0; }
size_t bf_bytes_per_gemm(const bf_config* c)
{
    return c ? (size_t)c->n_ant * c->n_freq * (size_t)bf_n_timesteps_per_gemm(c) : 0;
}
size_t bf_bytes_per_block(const bf_config* c) { return c ? bf_bytes_per_gemm(c) * (size_t)c->n_gemms_per_block : 0; }
size_t bf_floats_per_detect(const bf_config* c)
{
    return c ? (size_t)c->n_out_per_gemm * c->n_freq * (size_t)c->n_beams : 0;
}

int bf_device_count(int* count)
{
    if (!count) return fail(BF_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(BF_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return BF_OK;
}

int bf_device_name(int device, char* buf, size_t buflen)
{
    if (!buf || !buflen) return fail(BF_ERR_INVALID, "buffer is NULL");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return BF_OK;
}

int bf_create(const bf_config* cfg, int device, bf_handle** out)
{
    if (!out) return fail(BF_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (int rc = check_cfg(cfg)) return rc;
    dsabf::Geometry g = make_geom(*cfg);
    const char* why = nullptr;
    if (!dsabf::fused_supported(g, &why)) return fail(BF_ERR_INVALID, "unsupported geometry: %s", why);

    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(BF_ERR_NO_DEVICE, "no HIP device visible (libdsabf has no CPU fallback)");
    if (device < 0 || device >= n) return fail(BF_ERR_INVALID, "device %d out of range (0..%d)", device, n - 1);
    DeviceScope dev_scope_(device);
    if (dev_scope_.err != hipSuccess) return fail(BF_ERR_DEVICE, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(dev_scope_.err));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(BF_ERR_NO_DEVICE, "device %d is %s; libdsabf is built for gfx950 only", device, prop.gcnArchName);

    bf_handle* h = new (std::nothrow) bf_handle();
    if (!h) return fail(BF_ERR_DEVICE, "out of host memory");
    h->cfg = *cfg;
    h->geom = g;
    h->device = device;
    h->n_cus = prop.multiProcessorCount;
    *out = h;  // from here on bf_destroy cleans up partial state

#define CREATE_TRY(expr)                                                                                     \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) {                                                                              \
            int rc_ = fail(BF_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));                    \
            std::string keep = g_err;                                                                        \
            bf_destroy(h);                                                                                   \
            *out = nullptr;                                                                                  \
            g_err = keep;                                                                                    \
            return rc_;                                                                                      \
        }                                                                                                    \
    } while (0)

    CREATE_TRY(hipMalloc(&h->d_wimage, dsabf::weight_image_bytes(g)));
    if (dsabf::weight_pair_image_bytes(g)) CREATE_TRY(hipMalloc(&h->d_wimage_p, dsabf::weight_pair_image_bytes(g)));
    CREATE_TRY(hipMalloc((void**)&h->d_flag, 2 * sizeof(int)));
    CREATE_TRY(hipMalloc((void**)&h->d_data, bf_bytes_per_block(cfg) * (size_t)cfg->n_blocks_on_gpu));
    CREATE_TRY(hipMalloc((void**)&h->d_out, bf_floats_per_detect(cfg) * sizeof(float) * (size_t)cfg->n_streams));
    CREATE_TRY(hipMalloc((void**)&h->d_ded, (size_t)cfg->n_beams * sizeof(float) * (size_t)cfg->n_streams));
    // src/beamformer.cu:291-298: the reference zeroes its buffers
    CREATE_TRY(hipMemset(h->d_data, 0, bf_bytes_per_block(cfg) * (size_t)cfg->n_blocks_on_gpu));
    CREATE_TRY(hipMemset(h->d_out, 0, bf_floats_per_detect(cfg) * sizeof(float) * (size_t)cfg->n_streams));
    CREATE_TRY(hipMemset(h->d_ded, 0, (size_t)cfg->n_beams * sizeof(float) * (size_t)cfg->n_streams));
    CREATE_TRY(hipStreamCreateWithFlags(&h->h2d, hipStreamNonBlocking));
    h->streams.resize(cfg->n_streams, nullptr);
    h->join.resize(cfg->n_streams, nullptr);
    h->last_out.resize(cfg->n_streams, nullptr);
    h->last_q.resize(cfg->n_streams, 0);
    for (int i = 0; i < cfg->n_streams; i++) {
        h->last_out[i] = h->d_out + bf_floats_per_detect(cfg) * (size_t)i;
        h->last_q[i] = i;
    }
    {
        const char* env = getenv("DSABF_COALESCE");
        h->coalesce = !(env && env[0] == '0');
    }
    CREATE_TRY(hipEventCreateWithFlags(&h->flush_done, hipEventDisableTiming));
    for (int i = 0; i < cfg->n_streams; i++) {
        CREATE_TRY(hipStreamCreateWithFlags(&h->streams[i], hipStreamNonBlocking));
        CREATE_TRY(hipEventCreateWithFlags(&h->join[i], hipEventDisableTiming));
    }
#undef CREATE_TRY
    return BF_OK;
}

static int flush_units(bf_handle* h);
static void dm_stream_release(struct bf_dm_stream* s);

int bf_destroy(bf_handle* h)
{
    if (!h) return BF_OK;
    DeviceScope dev_scope_(h->device);
    // gemm-units queued but never joined by an event or a sync: launch them (as the literal pattern would have) so that their
    // host copies land before the queues are drained below -- a destroy must not silently drop work that was accepted
    if (!h->streams.empty() && h->streams[0] && h->flush_done) (void)flush_units(h);
    h->pending.clear();
    for (auto s : h->streams)
        if (s) (void)hipStreamSynchronize(s);
    if (h->h2d) (void)hipStreamSynchronize(h->h2d);
    for (auto e : h->join)
        if (e) (void)hipEventDestroy(e);
    for (auto s : h->streams)
        if (s) (void)hipStreamDestroy(s);
    if (h->h2d) (void)hipStreamDestroy(h->h2d);
    if (h->flush_done) (void)hipEventDestroy(h->flush_done);
    if (h->t0) (void)hipEventDestroy(h->t0);
    if (h->t1) (void)hipEventDestroy(h->t1);
    (void)hipFree(h->d_wimage);
    (void)hipFree(h->d_wimage_p);
    (void)hipFree(h->d_flag);
    (void)hipFree(h->d_data);
    (void)hipFree(h->d_out);
    (void)hipFree(h->d_ded);
    for (auto* ds : h->dm_streams) dm_stream_release(ds);   // a DM stage that outlives its handle is left empty, not dangling
    for (auto& sc : h->dm_scratch) (void)hipFree(sc.second);
    for (float* p : h->d_out_blk) (void)hipFree(p);
    for (float* p : h->d_full_blk) (void)hipFree(p);
    for (float* p : h->d_stage_blk) (void)hipFree(p);
    for (float* p : h->d_ded_blk) (void)hipFree(p);
    delete h;
    return BF_OK;
}

int bf_get_config(const bf_handle* h, bf_config* cfg)
{
    if (!h || !cfg) return fail(BF_ERR_INVALID, "NULL argument");
    *cfg = h->cfg;
    return BF_OK;
}

static int finish_weights(bf_handle* h, const int8_t* d_w, hipStream_t s)
{
    if (int rc = flush_units(h)) return rc;   // gemm-units still queued were enqueued under the OLD weights: launch them first
    for (auto q : h->streams) HIP_TRY(hipStreamSynchronize(q));   // ... and let them finish before the images change
    HIP_TRY(hipMemsetAsync(h->d_flag, 0, 2 * sizeof(int), s));
    HIP_TRY(dsabf::launch_weight_relayout(h->geom, d_w, h->d_wimage, h->d_wimage_p, h->d_flag, s));
    int bad[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(bad, h->d_flag, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    h->geom.paired = false;
    if (bad[0]) {
        h->weights_set = false;
        return fail(BF_ERR_INVALID, "weights contain an imaginary part of -128 (must be >= -127)");
    }
    // Beam sets symmetric about the boresight (the reference's linear fan and grid) have W[B-1-b] = conj(W[b]) exactly;
    // the device check above decides per weight set; DSABF_PAIRED=0 in the environment, or bf_set_switch(h, "paired", 0) before
    // the weights are set, forces the general kernel.
    const char* env = getenv("DSABF_PAIRED");   // read per weight set (not per launch): the choice is part of setting weights
    h->geom.paired = h->d_wimage_p != nullptr && bad[1] == 0 && !h->force_general && !(env && env[0] == '0');
    h->weights_set = true;
    return BF_OK;
}

int bf_set_weights(bf_handle* h, const int8_t* w)
{
    if (!h || !w) return fail(BF_ERR_INVALID, "NULL argument");
    ON_DEVICE(h);
    const size_t n = (size_t)h->cfg.n_freq * h->cfg.n_ant * h->cfg.n_beams * 2;
    int8_t* d_w = nullptr;
    HIP_TRY(hipMalloc((void**)&d_w, n));
    hipError_t e = hipMemcpy(d_w, w, n, hipMemcpyHostToDevice);
    int rc = BF_OK;
    if (e != hipSuccess)
        rc = fail(BF_ERR_DEVICE, "weight upload failed: %s", hipGetErrorString(e));
    else
        rc = finish_weights(h, d_w, h->streams[0]);
    (void)hipFree(d_w);
    return rc;
}

int bf_set_weights_device(bf_handle* h, const int8_t* d_w, void* hip_stream)
{
    if (!h || !d_w) return fail(BF_ERR_INVALID, "NULL argument");
    ON_DEVICE(h);
    return finish_weights(h, d_w, as_stream(hip_stream));
}

int bf_alloc_pinned(void** ptr, size_t nbytes)
{
    if (!ptr) return fail(BF_ERR_INVALID, "ptr is NULL");
    *ptr = nullptr;
    HIP_TRY(hipHostMalloc(ptr, nbytes, hipHostMallocDefault));
    return BF_OK;
}

int bf_free_pinned(void* ptr)
{
    if (!ptr) return BF_OK;
    HIP_TRY(hipHostFree(ptr));
    return BF_OK;
}

int bf_host_register(void* ptr, size_t nbytes)
{
    if (!ptr || !nbytes) return fail(BF_ERR_INVALID, "NULL argument");
    HIP_TRY(hipHostRegister(ptr, nbytes, hipHostRegisterDefault));
    return BF_OK;
}

int bf_host_unregister(void* ptr)
{
    if (!ptr) return BF_OK;
    HIP_TRY(hipHostUnregister(ptr));
    return BF_OK;
}

static int event_create_here(bf_event** ev);

// HIP binds an event to the device that is current when it is created, and it can only be recorded on that device's
// streams: events a handle's queues will record are created under the HANDLE's device, whatever the caller's is.
int bf_event_create_on(bf_handle* h, bf_event** ev)
{
    if (!h) return fail(BF_ERR_INVALID, "handle is NULL");
    ON_DEVICE(h);
    return event_create_here(ev);
}

int bf_event_create(bf_event** ev) { return event_create_here(ev); }

static int event_create_here(bf_event** ev)
{
    if (!ev) return fail(BF_ERR_INVALID, "ev is NULL");
    *ev = nullptr;
    bf_event* e = new (std::nothrow) bf_event();
    if (!e) return fail(BF_ERR_DEVICE, "out of host memory");
    hipError_t rc = hipEventCreateWithFlags(&e->ev, hipEventDisableTiming);  // src/observation_loop.hh:59-60
    if (rc != hipSuccess) {
        delete e;
        return fail(BF_ERR_DEVICE, "hipEventCreateWithFlags: %s", hipGetErrorString(rc));
    }
    *ev = e;
    return BF_OK;
}

int bf_event_destroy(bf_event* ev)
{
    if (!ev) return BF_OK;
    hipError_t rc = hipEventDestroy(ev->ev);
    delete ev;
    if (rc != hipSuccess) return fail(BF_ERR_DEVICE, "hipEventDestroy: %s", hipGetErrorString(rc));
    return BF_OK;
}

int bf_event_query(bf_event* ev)
{
    if (!ev) return fail(BF_ERR_INVALID, "ev is NULL");
    hipError_t rc = hipEventQuery(ev->ev);  // an event never recorded reports "done", as in CUDA
    if (rc == hipSuccess) return BF_OK;
    if (rc == hipErrorNotReady) return BF_NOT_READY;
    return fail(BF_ERR_DEVICE, "hipEventQuery: %s", hipGetErrorString(rc));
}

int bf_event_synchronize(bf_event* ev)
{
    if (!ev) return fail(BF_ERR_INVALID, "ev is NULL");
    HIP_TRY(hipEventSynchronize(ev->ev));
    return BF_OK;
}

int bf_submit_block(bf_handle* h, int slot, const void* host, size_t nbytes, bf_event* ev)
{
    if (!h || !host) return fail(BF_ERR_INVALID, "NULL argument");
    if (slot < 0 || slot >= h->cfg.n_blocks_on_gpu) return fail(BF_ERR_INVALID, "slot %d out of range", slot);
    const size_t block = bf_bytes_per_block(&h->cfg);
    if (nbytes > block) return fail(BF_ERR_INVALID, "nbytes %zu exceeds the block size %zu", nbytes, block);
    ON_DEVICE(h);
    HIP_TRY(hipMemcpyAsync(h->d_data + block * (size_t)slot, host, nbytes, hipMemcpyHostToDevice, h->h2d));
    if (ev) {
        HIP_TRY(hipEventRecord(ev->ev, h->h2d));
        ev->recorded = true;
    }
    return BF_OK;
}

int bf_record_transfer_event(bf_handle* h, bf_event* ev)
{
    if (!h || !ev) return fail(BF_ERR_INVALID, "NULL argument");
    ON_DEVICE(h);
    HIP_TRY(hipEventRecord(ev->ev, h->h2d));
    ev->recorded = true;
    return BF_OK;
}

static int ensure_block_buffers(bf_handle* h, int q, bool ded)
{
    const size_t per_det = bf_floats_per_detect(&h->cfg);
    if (h->d_out_blk.empty()) h->d_out_blk.assign((size_t)h->cfg.n_streams, nullptr);
    if (!h->d_out_blk[q])
        HIP_TRY(hipMalloc((void**)&h->d_out_blk[q], per_det * sizeof(float) * (size_t)h->cfg.n_gemms_per_block));
    if (ded) {
        if (h->d_ded_blk.empty()) h->d_ded_blk.assign((size_t)h->cfg.n_streams, nullptr);
        if (!h->d_ded_blk[q])
            HIP_TRY(hipMalloc((void**)&h->d_ded_blk[q], (size_t)h->cfg.n_beams * sizeof(float) * (size_t)h->cfg.n_gemms_per_block));
    }
    return BF_OK;
}

// Launches what bf_enqueue_gemm_unit / bf_enqueue_dedisperse have queued: per run of consecutive gemm-units of one ring
// slot ONE fused launch (the reference's loop enqueues time slices 0, 1, 2, ... of a block: one run = the block), one
// DM-0 launch per run of units that asked for it, and the host copies -- every unit's, in the order they were enqueued,
// neighbours in device AND host memory as one copy.  All of it on ONE compute queue (they rotate per flush); the host copies
// wait for the previous flush's, so that two units copied to the same host buffer land in enqueue order as they do on
// the reference's per-queue streams (src/beamformer.cu:485-488 overwrites beam_out[stream] unit after unit), while this
// flush's kernel already overlaps the previous flush's copies.
// A launch on compute queue q is about to overwrite gemm-units [ts0, ts1) of that queue's block buffer.  A caller-visible queue
// whose MOST RECENT gemm-unit's powers still live there -- bf_enqueue_dedisperse may yet be called for it: "the unit last
// enqueued on stream_idx", include/dsabf.h -- and that is not given a newer unit by this very launch (`reassigned`) gets them
// moved to its own slot of d_out first: behind whatever queue s still has in flight on that slot (a literal-pattern unit's
// host copy), in front of the launch.  Never happens in the reference's loop (a time slice belongs to one queue there).
static int preserve_last_units(bf_handle* h, int q, size_t ts0, size_t ts1, const std::vector<char>* reassigned)
{
    if (h->d_out_blk.empty() || !h->d_out_blk[q]) return BF_OK;
    const size_t per_det = bf_floats_per_detect(&h->cfg);
    const float* lo = h->d_out_blk[q] + per_det * ts0;
    const float* hi = h->d_out_blk[q] + per_det * ts1;
    for (int s = 0; s < h->cfg.n_streams; s++) {
        if (reassigned && (*reassigned)[(size_t)s]) continue;
        const float* p = h->last_out[s];
        if (p < lo || p >= hi) continue;
        float* keep = h->d_out + per_det * (size_t)s;
        if (s != q) {
            HIP_TRY(hipEventRecord(h->join[s], h->streams[s]));
            HIP_TRY(hipStreamWaitEvent(h->streams[q], h->join[s], 0));
        }
        HIP_TRY(hipMemcpyAsync(keep, p, per_det * sizeof(float), hipMemcpyDeviceToDevice, h->streams[q]));
        h->last_out[s] = keep;
        h->last_q[s] = q;
    }
    return BF_OK;
}

static int flush_units(bf_handle* h)
{
    if (h->pending.empty()) return BF_OK;
    std::vector<bf_handle::pending_unit> units;
    units.swap(h->pending);              // (whatever happens below, nothing stays queued)
    // two queues take turns (each owns a block-sized device buffer, allocated at first use): flush i + 1's kernel runs under
    // flush i's host copies; more queues would only hold more buffers
    const int q = (int)(h->flush_seq++ % (uint64_t)(h->cfg.n_streams < 2 ? 1 : 2));
    hipStream_t s = h->streams[q];
    const size_t per_gemm = bf_bytes_per_gemm(&h->cfg), per_det = bf_floats_per_detect(&h->cfg);
    const size_t n_beams = (size_t)h->cfg.n_beams;
    bool any_ded = false;
    for (const auto& u : units) any_ded |= u.ded;
    if (int rc = ensure_block_buffers(h, q, any_ded)) return rc;
    float* blk = h->d_out_blk[q];
    const size_t n = units.size();
    auto follows = [&](size_t k) {       // unit k continues the run of unit k - 1
        return units[k].slot == units[k - 1].slot && units[k].time_slice == units[k - 1].time_slice + 1;
    };
    std::vector<char> reassigned((size_t)h->cfg.n_streams, 0);   // queues that get a newer "most recent unit" from this flush
    for (const auto& u : units) reassigned[(size_t)u.stream_idx] = 1;
    for (size_t i = 0; i < n;) {
        size_t j = i + 1;
        while (j < n && follows(j)) j++;
        const uint8_t* in = h->d_data + per_gemm * ((size_t)h->cfg.n_gemms_per_block * units[i].slot + units[i].time_slice);
        if (int rc = preserve_last_units(h, q, (size_t)units[i].time_slice, (size_t)units[i].time_slice + (j - i), &reassigned)) return rc;
        h->n_fused_launches++;
    HIP_TRY(dsabf::launch_fused(h->geom, h->d_wimage, h->d_wimage_p, in, (int)(j - i), blk + per_det * (size_t)units[i].time_slice,
                                    h->n_cus, s));
        for (size_t a = i; a < j;) {     // DM-0 rows of the run: one launch per stretch of units that asked for one
            if (!units[a].ded) {
                a++;
                continue;
            }
            size_t b = a + 1;
            while (b < j && units[b].ded) b++;
            HIP_TRY(dsabf::launch_dedisperse_units(h->geom, blk + per_det * (size_t)units[a].time_slice, per_det, (int)(b - a),
                                                   h->d_ded_blk[q] + n_beams * (size_t)units[a].time_slice, s));
            a = b;
        }
        i = j;
    }
    if (h->flush_recorded) HIP_TRY(hipStreamWaitEvent(s, h->flush_done, 0));
    for (size_t i = 0; i < n;) {         // a4: the detected powers
        if (!units[i].host_out) {
            i++;
            continue;
        }
        size_t j = i + 1;
        while (j < n && follows(j) && units[j].host_out == units[j - 1].host_out + per_det) j++;
        HIP_TRY(hipMemcpyAsync(units[i].host_out, blk + per_det * (size_t)units[i].time_slice, per_det * sizeof(float) * (j - i),
                               hipMemcpyDeviceToHost, s));
        i = j;
    }
    for (size_t i = 0; i < n;) {         // a8: the DM-0 rows
        if (!units[i].ded || !units[i].ded_row) {
            i++;
            continue;
        }
        size_t j = i + 1;
        while (j < n && follows(j) && units[j].ded && units[j].ded_row == units[j - 1].ded_row + n_beams) j++;
        HIP_TRY(hipMemcpyAsync(units[i].ded_row, h->d_ded_blk[q] + n_beams * (size_t)units[i].time_slice, n_beams * sizeof(float) * (j - i),
                               hipMemcpyDeviceToHost, s));
        i = j;
    }
    HIP_TRY(hipEventRecord(h->flush_done, s));
    h->flush_recorded = true;
    // The per-queue ordering guarantee of the literal pattern, kept: every caller-visible queue that had a unit in this flush
    // waits for the flush's end.  Whatever the caller orders on streams[stream_idx] afterwards -- a raw hipStreamSynchronize on
    // the stream bf_queue_stream handed out earlier, its own event, a bf_enqueue_d2h, RCCL chained on it -- is behind the
    // unit's launch AND its host copy, exactly as when the unit itself ran there.
    for (int st = 0; st < h->cfg.n_streams; st++)
        if (reassigned[(size_t)st] && st != q) HIP_TRY(hipStreamWaitEvent(h->streams[st], h->flush_done, 0));
    for (const auto& u : units) {
        h->last_out[u.stream_idx] = blk + per_det * (size_t)u.time_slice;
        h->last_q[u.stream_idx] = q;
    }
    return BF_OK;
}
#define FLUSH_UNITS(h_)                        \
    do {                                       \
        if (int rc_ = flush_units(h_)) return rc_; \
    } while (0)

int bf_enqueue_gemm_unit(bf_handle* h, int stream_idx, int slot, int time_slice, float* host_out)
{
    if (!h) return fail(BF_ERR_INVALID, "handle is NULL");
    if (!h->weights_set) return fail(BF_ERR_STATE, "bf_set_weights has not been called");
    if (stream_idx < 0 || stream_idx >= h->cfg.n_streams) return fail(BF_ERR_INVALID, "stream %d out of range", stream_idx);
    if (slot < 0 || slot >= h->cfg.n_blocks_on_gpu) return fail(BF_ERR_INVALID, "slot %d out of range", slot);
    if (time_slice < 0 || time_slice >= h->cfg.n_gemms_per_block)
        return fail(BF_ERR_INVALID, "time_slice %d out of range", time_slice);
    ON_DEVICE(h);
    if (h->coalesce) {
        // a whole block is queued, or this time slice's place in the block buffer is taken: launch what is there first
        bool clash = h->pending.size() >= (size_t)h->cfg.n_gemms_per_block;
        for (const auto& u : h->pending) clash |= u.time_slice == time_slice;
        if (clash) FLUSH_UNITS(h);
        h->pending.push_back({stream_idx, slot, time_slice, host_out, nullptr, false});
        return BF_OK;
    }
    const size_t per_gemm = bf_bytes_per_gemm(&h->cfg);
    const size_t per_det = bf_floats_per_detect(&h->cfg);
    // src/beamformer.cu:464: &d_data[N_BYTES_PRE_EXPANSION_PER_GEMM*(N_GEMMS_PER_BLOCK*block + timeSlice)]
    const uint8_t* in = h->d_data + per_gemm * ((size_t)h->cfg.n_gemms_per_block * slot + time_slice);
    float* out = h->d_out + per_det * (size_t)stream_idx;
    hipStream_t s = h->streams[stream_idx];
    if (h->last_out[stream_idx] == out && h->last_q[stream_idx] != stream_idx) {
        // this queue's slot holds powers that preserve_last_units moved here on ANOTHER queue (and a DM-0 request may be reading
        // them there): overwrite it behind that queue's work
        const int lq = h->last_q[stream_idx];
        HIP_TRY(hipEventRecord(h->join[lq], h->streams[lq]));
        HIP_TRY(hipStreamWaitEvent(s, h->join[lq], 0));
    }
    h->n_fused_launches++;
    HIP_TRY(dsabf::launch_fused(h->geom, h->d_wimage, h->d_wimage_p, in, 1, out, h->n_cus, s));
    if (host_out) HIP_TRY(hipMemcpyAsync(host_out, out, per_det * sizeof(float), hipMemcpyDeviceToHost, s));
    h->last_out[stream_idx] = out;
    h->last_q[stream_idx] = stream_idx;
    return BF_OK;
}

// d_dst: where the launch's powers go ([unit][o][f][b] of its n_units gemm-units); NULL: this queue's block buffer
static int enqueue_block_impl(bf_handle* h, int stream_idx, int slot, int first_unit, int n_units, float* d_dst, float* const* host_out)
{
    if (!h) return fail(BF_ERR_INVALID, "handle is NULL");
    if (!h->weights_set) return fail(BF_ERR_STATE, "bf_set_weights has not been called");
    if (stream_idx < 0 || stream_idx >= h->cfg.n_streams) return fail(BF_ERR_INVALID, "stream %d out of range", stream_idx);
    if (slot < 0 || slot >= h->cfg.n_blocks_on_gpu) return fail(BF_ERR_INVALID, "slot %d out of range", slot);
    if (first_unit < 0 || n_units <= 0 || first_unit + n_units > h->cfg.n_gemms_per_block)
        return fail(BF_ERR_INVALID, "gemm-units [%d, %d) are not inside a block of %d", first_unit, first_unit + n_units,
                    h->cfg.n_gemms_per_block);
    ON_DEVICE(h);
    FLUSH_UNITS(h);
    const size_t per_gemm = bf_bytes_per_gemm(&h->cfg);
    const size_t per_det = bf_floats_per_detect(&h->cfg);
    float* out = d_dst;
    if (!d_dst) {
        if (h->d_out_blk.empty()) h->d_out_blk.assign((size_t)h->cfg.n_streams, nullptr);
        if (h->blk_ran.empty()) h->blk_ran.assign((size_t)h->cfg.n_streams, 0);
        // This queue's block buffer, at first use.  A hipMalloc in the middle of a stream of blocks stalls the device (measured: 9.4 ->
        // 10.9 us per beam-block), so a caller that rotates over queues reserves them BEFORE its loop with bf_block_output_device
        // (run_observation does, for the queues it will use; include/dsabf.h says so at bf_enqueue_block) -- the library does not
        // guess and allocate all n_streams of them (8 x 128 MiB at the production geometry, six of them dead for a two-queue loop).
        if (!h->d_out_blk[stream_idx])
            HIP_TRY(hipMalloc((void**)&h->d_out_blk[stream_idx], per_det * sizeof(float) * (size_t)h->cfg.n_gemms_per_block));
        h->blk_ran[stream_idx] = 1;
        if (int rc = preserve_last_units(h, stream_idx, (size_t)first_unit, (size_t)first_unit + (size_t)n_units, nullptr)) return rc;
        out = h->d_out_blk[stream_idx] + per_det * (size_t)first_unit;
    }
    const uint8_t* in = h->d_data + per_gemm * ((size_t)h->cfg.n_gemms_per_block * slot + first_unit);
    hipStream_t s = h->streams[stream_idx];
    h->n_fused_launches++;
    HIP_TRY(dsabf::launch_fused(h->geom, h->d_wimage, h->d_wimage_p, in, n_units, out, h->n_cus, s));
    if (host_out)
        for (int u = 0; u < n_units;) {   // destinations that follow each other in host memory travel as ONE copy
            if (!host_out[u]) {
                u++;
                continue;
            }
            int run = 1;
            while (u + run < n_units && host_out[u + run] == host_out[u] + per_det * (size_t)run) run++;
            HIP_TRY(hipMemcpyAsync(host_out[u], out + per_det * (size_t)u, per_det * sizeof(float) * (size_t)run,
                                   hipMemcpyDeviceToHost, s));
            u += run;
        }
    return BF_OK;
}

int bf_enqueue_block(bf_handle* h, int stream_idx, int slot, int first_unit, int n_units, float* const* host_out)
{
    return enqueue_block_impl(h, stream_idx, slot, first_unit, n_units, nullptr, host_out);
}

int bf_enqueue_block_to(bf_handle* h, int stream_idx, int slot, int first_unit, int n_units, float* d_dst, float* const* host_out)
{
    if (!d_dst) return fail(BF_ERR_INVALID, "d_dst is NULL");
    return enqueue_block_impl(h, stream_idx, slot, first_unit, n_units, d_dst, host_out);
}

int bf_enqueue_block_dedisperse(bf_handle* h, int stream_idx, int first_unit, int n_units, float* host_rows)
{
    if (!h) return fail(BF_ERR_INVALID, "handle is NULL");
    if (stream_idx < 0 || stream_idx >= h->cfg.n_streams) return fail(BF_ERR_INVALID, "stream %d out of range", stream_idx);
    if (first_unit < 0 || n_units <= 0 || first_unit + n_units > h->cfg.n_gemms_per_block)
        return fail(BF_ERR_INVALID, "gemm-units [%d, %d) are not inside a block of %d", first_unit, first_unit + n_units,
                    h->cfg.n_gemms_per_block);
    if (h->blk_ran.empty() || !h->blk_ran[stream_idx])
        return fail(BF_ERR_STATE, "bf_enqueue_block has not run on queue %d", stream_idx);
    ON_DEVICE(h);
    FLUSH_UNITS(h);
    const size_t per_det = bf_floats_per_detect(&h->cfg);
    if (h->d_ded_blk.empty()) h->d_ded_blk.assign((size_t)h->cfg.n_streams, nullptr);
    if (!h->d_ded_blk[stream_idx])   // (n_gemms_per_block x n_beams floats: 32 KiB)
        HIP_TRY(hipMalloc((void**)&h->d_ded_blk[stream_idx], (size_t)h->cfg.n_beams * sizeof(float) * (size_t)h->cfg.n_gemms_per_block));
    hipStream_t s = h->streams[stream_idx];
    float* ded = h->d_ded_blk[stream_idx] + (size_t)h->cfg.n_beams * first_unit;
    HIP_TRY(dsabf::launch_dedisperse_units(h->geom, h->d_out_blk[stream_idx] + per_det * (size_t)first_unit, per_det, n_units, ded, s));
    if (host_rows)
        HIP_TRY(hipMemcpyAsync(host_rows, ded, (size_t)h->cfg.n_beams * sizeof(float) * (size_t)n_units, hipMemcpyDeviceToHost, s));
    return BF_OK;
}

int bf_block_output_device(bf_handle* h, int stream_idx, float** d_out)
{
    if (!h || !d_out) return fail(BF_ERR_INVALID, "NULL argument");
    if (stream_idx < 0 || stream_idx >= h->cfg.n_streams) return fail(BF_ERR_INVALID, "stream %d out of range", stream_idx);
    ON_DEVICE(h);
    if (h->d_out_blk.empty()) h->d_out_blk.assign((size_t)h->cfg.n_streams, nullptr);
    if (!h->d_out_blk[stream_idx])
        HIP_TRY(hipMalloc((void**)&h->d_out_blk[stream_idx],
                          bf_floats_per_detect(&h->cfg) * sizeof(float) * (size_t)h->cfg.n_gemms_per_block));
    if (h->blk_ran.empty()) h->blk_ran.assign((size_t)h->cfg.n_streams, 0);
    h->blk_ran[stream_idx] = 1;   // (the caller may fill the buffer itself and ask for its DM-0 rows)
    *d_out = h->d_out_blk[stream_idx];
    return BF_OK;
}

int bf_block_gather_device(bf_handle* h, int stream_idx, int world, float** d_full)
{
    if (!h || !d_full) return fail(BF_ERR_INVALID, "NULL argument");
    if (stream_idx < 0 || stream_idx >= h->cfg.n_streams) return fail(BF_ERR_INVALID, "stream %d out of range", stream_idx);
    if (world < 1) return fail(BF_ERR_INVALID, "world must be positive");
    if (h->full_world && h->full_world != world) return fail(BF_ERR_STATE, "the gather buffers were sized for world %d", h->full_world);
    ON_DEVICE(h);
    h->full_world = world;
    if (h->d_full_blk.empty()) h->d_full_blk.assign((size_t)h->cfg.n_streams, nullptr);
    if (!h->d_full_blk[stream_idx])
        HIP_TRY(hipMalloc((void**)&h->d_full_blk[stream_idx],
                          bf_floats_per_detect(&h->cfg) * sizeof(float) * (size_t)h->cfg.n_gemms_per_block * (size_t)world));
    *d_full = h->d_full_blk[stream_idx];
    return BF_OK;
}

int bf_block_gather_stage_device(bf_handle* h, int stream_idx, int world, float** d_stage)
{
    if (!h || !d_stage) return fail(BF_ERR_INVALID, "NULL argument");
    if (stream_idx < 0 || stream_idx >= h->cfg.n_streams) return fail(BF_ERR_INVALID, "stream %d out of range", stream_idx);
    if (world < 1) return fail(BF_ERR_INVALID, "world must be positive");
    if (h->full_world && h->full_world != world) return fail(BF_ERR_STATE, "the gather buffers were sized for world %d", h->full_world);
    ON_DEVICE(h);
    h->full_world = world;
    if (h->d_stage_blk.empty()) h->d_stage_blk.assign((size_t)h->cfg.n_streams, nullptr);
    if (!h->d_stage_blk[stream_idx])
        HIP_TRY(hipMalloc((void**)&h->d_stage_blk[stream_idx],
                          bf_floats_per_detect(&h->cfg) * sizeof(float) * (size_t)h->cfg.n_gemms_per_block * (size_t)world));
    *d_stage = h->d_stage_blk[stream_idx];
    return BF_OK;
}

int bf_enqueue_d2h(bf_handle* h, int stream_idx, const float* d_src, float* host_dst, size_t n_floats)
{
    if (!h || !d_src || !host_dst) return fail(BF_ERR_INVALID, "NULL argument");
    if (stream_idx < 0 || stream_idx >= h->cfg.n_streams) return fail(BF_ERR_INVALID, "stream %d out of range", stream_idx);
    ON_DEVICE(h);
    FLUSH_UNITS(h);
    HIP_TRY(hipMemcpyAsync(host_dst, d_src, n_floats * sizeof(float), hipMemcpyDeviceToHost, h->streams[stream_idx]));
    return BF_OK;
}

int bf_queue_stream(bf_handle* h, int stream_idx, void** hip_stream)
{
    if (!h || !hip_stream) return fail(BF_ERR_INVALID, "NULL argument");
    if (stream_idx < 0 || stream_idx >= h->cfg.n_streams) return fail(BF_ERR_INVALID, "stream %d out of range", stream_idx);
    ON_DEVICE(h);
    FLUSH_UNITS(h);   // the caller is about to order its own work against this queue: nothing of ours may still be only queued
    *hip_stream = h->streams[stream_idx];
    return BF_OK;
}

int bf_enqueue_dedisperse(bf_handle* h, int stream_idx, float* host_out_row)
{
    if (!h) return fail(BF_ERR_INVALID, "handle is NULL");
    if (stream_idx < 0 || stream_idx >= h->cfg.n_streams) return fail(BF_ERR_INVALID, "stream %d out of range", stream_idx);
    ON_DEVICE(h);
    // the gemm-unit this call refers to -- the most recent one of queue stream_idx -- may still be queued: its DM-0 row is then
    // part of the same flush (one launch for all the rows of a run)
    for (size_t k = h->pending.size(); k-- > 0;)
        if (h->pending[k].stream_idx == stream_idx) {
            if (h->pending[k].ded) break;   // a second collapse of the same unit: run it directly below
            h->pending[k].ded = true;
            h->pending[k].ded_row = host_out_row;
            return BF_OK;
        }
    FLUSH_UNITS(h);
    // d_ded[stream_idx] and the host row belong to queue stream_idx: every direct request runs THERE, in call order, behind the
    // queue that produced (or moved) the unit's powers if that was another one -- two successive requests can then neither
    // overwrite d_ded under a copy in flight nor land their rows out of order
    hipStream_t s = h->streams[stream_idx];
    const int lq = h->last_q[stream_idx];
    if (lq != stream_idx) {
        HIP_TRY(hipEventRecord(h->join[lq], h->streams[lq]));
        HIP_TRY(hipStreamWaitEvent(s, h->join[lq], 0));
    }
    float* ded = h->d_ded + (size_t)h->cfg.n_beams * stream_idx;
    HIP_TRY(dsabf::launch_dedisperse(h->geom, h->last_out[stream_idx], ded, s));
    if (lq != stream_idx) {
        // ... and the producer queue waits for this read: the next launch that overwrites the unit's place in ITS block buffer (a later
        // flush on queue lq, for a queue whose latest unit is being replaced: preserve_last_units skips those) must not start under
        // it.  (Found by tools/fuzz_calls.py, seed 2118: unit, flush on queue A, late DM-0 on its own queue, next flush on A.)
        HIP_TRY(hipEventRecord(h->join[stream_idx], s));
        HIP_TRY(hipStreamWaitEvent(h->streams[lq], h->join[stream_idx], 0));
    }
    if (host_out_row)
        HIP_TRY(hipMemcpyAsync(host_out_row, ded, (size_t)h->cfg.n_beams * sizeof(float), hipMemcpyDeviceToHost, s));
    return BF_OK;
}

int bf_record_analysis_event(bf_handle* h, bf_event* ev)
{
    if (!h || !ev) return fail(BF_ERR_INVALID, "NULL argument");
    ON_DEVICE(h);
    FLUSH_UNITS(h);
    const int last = h->cfg.n_streams - 1;
    for (int i = 0; i < last; i++) {
        HIP_TRY(hipEventRecord(h->join[i], h->streams[i]));
        HIP_TRY(hipStreamWaitEvent(h->streams[last], h->join[i], 0));
    }
    HIP_TRY(hipEventRecord(ev->ev, h->streams[last]));
    ev->recorded = true;
    return BF_OK;
}

int bf_stream_sync(bf_handle* h, int stream_idx)
{
    if (!h) return fail(BF_ERR_INVALID, "handle is NULL");
    ON_DEVICE(h);
    FLUSH_UNITS(h);
    if (stream_idx < 0) {
        HIP_TRY(hipStreamSynchronize(h->h2d));
        for (auto s : h->streams) HIP_TRY(hipStreamSynchronize(s));
        return BF_OK;
    }
    if (stream_idx >= h->cfg.n_streams) return fail(BF_ERR_INVALID, "stream %d out of range", stream_idx);
    HIP_TRY(hipStreamSynchronize(h->streams[stream_idx]));
    if (h->last_q[stream_idx] != stream_idx)   // its most recent gemm-unit was coalesced into a launch on another queue
        HIP_TRY(hipStreamSynchronize(h->streams[h->last_q[stream_idx]]));
    return BF_OK;
}

int bf_timer_start(bf_handle* h)
{
    if (!h) return fail(BF_ERR_INVALID, "handle is NULL");
    ON_DEVICE(h);
    if (!h->t0) HIP_TRY(hipEventCreate(&h->t0));
    if (!h->t1) HIP_TRY(hipEventCreate(&h->t1));
    HIP_TRY(hipEventRecord(h->t0, nullptr));
    return BF_OK;
}

int bf_timer_stop(bf_handle* h, float* ms)
{
    if (!h || !ms) return fail(BF_ERR_INVALID, "NULL argument");
    if (!h->t0) return fail(BF_ERR_STATE, "bf_timer_start has not been called");
    ON_DEVICE(h);   // t1 goes onto the handle's device's null stream, where t0 is
    FLUSH_UNITS(h);
    HIP_TRY(hipEventRecord(h->t1, nullptr));
    HIP_TRY(hipEventSynchronize(h->t1));
    HIP_TRY(hipEventElapsedTime(ms, h->t0, h->t1));
    return BF_OK;
}

int bf_beamform_device(bf_handle* h, const void* d_packed, int n_units, float* d_out, void* hip_stream)
{
    if (!h || !d_packed || !d_out) return fail(BF_ERR_INVALID, "NULL argument");
    if (n_units <= 0) return fail(BF_ERR_INVALID, "n_units must be positive");
    if (!h->weights_set) return fail(BF_ERR_STATE, "bf_set_weights has not been called");
    if (((uintptr_t)d_packed & 15) || ((uintptr_t)d_out & 15))
        return fail(BF_ERR_INVALID, "misaligned device pointer: d_packed and d_out must be 16-byte aligned (the kernel loads "
                                    "16-byte pieces and stores 16-byte groups of beams)");
    ON_DEVICE(h);
    h->n_fused_launches++;
    HIP_TRY(dsabf::launch_fused(h->geom, h->d_wimage, h->d_wimage_p, d_packed, n_units, d_out, h->n_cus, as_stream(hip_stream)));
    return BF_OK;
}

int bf_expand_device(bf_handle* h, const void* d_in, size_t nbytes, void* d_out, void* hip_stream)
{
    if (!h || !d_in || !d_out) return fail(BF_ERR_INVALID, "NULL argument");
    if (nbytes % 16 || ((uintptr_t)d_in & 15) || ((uintptr_t)d_out & 15))
        return fail(BF_ERR_INVALID, "expand needs 16-byte aligned pointers and a multiple of 16 bytes");
    ON_DEVICE(h);
    HIP_TRY(dsabf::launch_expand(d_in, nbytes, d_out, as_stream(hip_stream)));
    return BF_OK;
}

int bf_mfma_peak_device(bf_handle* h, const void* d_operands, size_t operand_bytes, void* d_scratch, size_t scratch_bytes, int iters,
                        double* ops, void* hip_stream)
{
    if (!h || !d_operands || !d_scratch) return fail(BF_ERR_INVALID, "NULL argument");
    if (operand_bytes < dsabf::kMfmaPeakSrcBytes || scratch_bytes < dsabf::kMfmaPeakSinkBytes || ((uintptr_t)d_operands & 15) || iters <= 0)
        return fail(BF_ERR_INVALID, "need >= %zu operand bytes (16-byte aligned), >= %zu scratch bytes, iters > 0",
                    dsabf::kMfmaPeakSrcBytes, dsabf::kMfmaPeakSinkBytes);
    ON_DEVICE(h);
    HIP_TRY(dsabf::launch_mfma_peak(d_operands, d_scratch, iters, h->n_cus, ops, as_stream(hip_stream)));
    return BF_OK;
}

int bf_gather_relayout_device(bf_handle* h, const float* d_stage, float* d_full, size_t rows_held, int world, size_t row_floats,
                              int skip_rank, void* hip_stream)
{
    if (!h || !d_stage || !d_full) return fail(BF_ERR_INVALID, "NULL argument");
    if (world < 1 || row_floats % 4 || ((uintptr_t)d_stage & 15) || ((uintptr_t)d_full & 15))
        return fail(BF_ERR_INVALID, "need world >= 1, row_floats a multiple of 4 and 16-byte aligned pointers");
    ON_DEVICE(h);
    HIP_TRY(dsabf::launch_gather_relayout(d_stage, d_full, rows_held, world, row_floats, skip_rank, h->n_cus, as_stream(hip_stream)));
    return BF_OK;
}

int bf_gemm_device(bf_handle* h, const void* d_packed_unit, float* d_c, void* hip_stream)
{
    if (!h || !d_packed_unit || !d_c) return fail(BF_ERR_INVALID, "NULL argument");
    if (!h->weights_set) return fail(BF_ERR_STATE, "bf_set_weights has not been called");
    ON_DEVICE(h);
    HIP_TRY(dsabf::launch_gemm_only(h->geom, h->d_wimage, d_packed_unit, d_c, h->n_cus, as_stream(hip_stream)));
    return BF_OK;
}

int bf_dedisperse_device(bf_handle* h, const float* d_out_unit, float* d_ded, void* hip_stream)
{
    if (!h || !d_out_unit || !d_ded) return fail(BF_ERR_INVALID, "NULL argument");
    ON_DEVICE(h);
    HIP_TRY(dsabf::launch_dedisperse(h->geom, d_out_unit, d_ded, as_stream(hip_stream)));
    return BF_OK;
}

// Scratch of the DM-trial dedispersion for calls on stream `s`: kDwMaxGroups flag ints + one 512-byte row of zeros
// (dsabf::kDmScratchBytes), zeroed ON THAT STREAM when it is first used -- ordered before the kernels that read it, also on a
// non-blocking stream (a memset on the null stream would not be).
static hipError_t dm_scratch(bf_handle* h, hipStream_t s, int** out)
{
    for (auto& sc : h->dm_scratch)
        if (sc.first == s) {
            *out = sc.second;
            return hipSuccess;
        }
    if (h->dm_scratch.size() >= 64) {   // a caller that keeps creating streams: nothing of ours may still be in flight
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) return e;
        for (auto& sc : h->dm_scratch) (void)hipFree(sc.second);   // (the handle's bf_dm_streams own their scratch: untouched)
        h->dm_scratch.clear();
    }
    int* p = nullptr;
    hipError_t e = hipMalloc((void**)&p, dsabf::kDmScratchBytes);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(p, 0, dsabf::kDmScratchBytes, s);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return e;
    }
    h->dm_scratch.emplace_back(s, p);
    *out = p;
    return hipSuccess;
}

int bf_dedisperse_dm_device(bf_handle* h, const float* d_series, int n_t, const int32_t* d_delays, int n_dm, int n_t_out,
                            float* d_out, void* hip_stream)
{
    if (!h || !d_series || !d_delays || !d_out) return fail(BF_ERR_INVALID, "NULL argument");
    if (n_t <= 0 || n_dm < 0 || n_t_out < 0 || n_t_out > n_t) return fail(BF_ERR_INVALID, "need 0 <= n_t_out <= n_t, n_dm >= 0");
    ON_DEVICE(h);
    int* flags = nullptr;
    HIP_TRY(dm_scratch(h, as_stream(hip_stream), &flags));
    HIP_TRY(dsabf::launch_dedisperse_dm(h->geom, d_series, n_t, d_delays, n_dm, n_t_out, d_out, flags, as_stream(hip_stream)));
    return BF_OK;
}

int bf_dedisperse_band_device(bf_handle* h, const float* d_out_unit, int n_freq_total, float* d_ded, void* hip_stream)
{
    if (!h || !d_out_unit || !d_ded) return fail(BF_ERR_INVALID, "NULL argument");
    if (n_freq_total <= 0) return fail(BF_ERR_INVALID, "n_freq_total must be positive");
    ON_DEVICE(h);
    dsabf::Geometry g = h->geom;
    g.n_freq = n_freq_total;
    HIP_TRY(dsabf::launch_dedisperse(g, d_out_unit, d_ded, as_stream(hip_stream)));
    return BF_OK;
}

int bf_dedisperse_dm_band_device(bf_handle* h, const float* d_series, int n_t, int n_freq_total, const int32_t* d_delays,
                                 int n_dm, int n_t_out, float* d_out, void* hip_stream)
{
    if (!h || !d_series || !d_delays || !d_out) return fail(BF_ERR_INVALID, "NULL argument");
    if (n_freq_total <= 0 || n_t <= 0 || n_dm < 0 || n_t_out < 0 || n_t_out > n_t)
        return fail(BF_ERR_INVALID, "need n_freq_total > 0, 0 <= n_t_out <= n_t, n_dm >= 0");
    ON_DEVICE(h);
    dsabf::Geometry g = h->geom;
    g.n_freq = n_freq_total;
    int* flags = nullptr;
    HIP_TRY(dm_scratch(h, as_stream(hip_stream), &flags));
    HIP_TRY(dsabf::launch_dedisperse_dm(g, d_series, n_t, d_delays, n_dm, n_t_out, d_out, flags, as_stream(hip_stream)));
    return BF_OK;
}

// ---- DM-trial dedispersion as a stage of the observation loop (include/dsabf.h) ---------------------------
// The detected stream arrives block by block; out[dm][t][b] needs rows t .. t + max_delay.  The stream keeps the last
// max_delay rows of what it has seen in front of the rows of the next push (one device buffer, slid back to its start when its
// end is reached), so every push runs the SAME kernels over [carry | new rows] that bf_dedisperse_dm_device runs over a
// whole series -- and emits exactly the output times that became complete.  Every (trial, time, beam) sum still runs over
// ascending f in one register from +0: the concatenated chunks are bit-identical to one call over the whole series.
struct bf_dm_stream {
    bf_handle* h = nullptr;
    int n_dm = 0, n_freq = 0, max_delay = 0, max_rows = 0;
    size_t row_floats = 0;
    // The rows live in a RING of cap_rows rows whose physical memory is mapped TWICE, back to back, into one virtual range (HIP's
    // virtual-memory API): row i is also row i + cap_rows, so every window of <= cap_rows consecutive rows -- the carried-over
    // delay window in front of a push's rows -- is contiguous for the kernels wherever it starts, and nothing ever moves.  (Rounds
    // 5's linear buffer slid the carry back to its start every few pushes: at the production block, 31 MiB read and written again
    // every 2.5 blocks.)  ring == false: that linear buffer -- for a device without VMM support, and bf_set_switch("dm_ring", 0).
    bool ring = false;
    size_t cap_rows = 0;          // ring: rows of physical memory (>= max_delay + 3 max_rows); linear: 2 (max_delay + max_rows)
    size_t wpos = 0;              // ring: physical row the next pushed row goes to (< cap_rows)
    size_t fill = 0;              // linear: rows of d_buf in use, [fill - carry, fill) are the newest rows of the series
    hipMemGenericAllocationHandle_t phys{};
    size_t phys_bytes = 0;
    bool phys_created = false, mapped0 = false, mapped1 = false;
    uint64_t pushed = 0;          // rows the stream has been given
    uint64_t n_push = 0;          // pushes so far
    float* d_buf = nullptr;       // ring: the double mapping (2 x phys_bytes of address space); linear: cap_rows x [freq][beam]
    int32_t* d_delays = nullptr;  // [n_dm][freq]
    // Three pushes may be in flight at once (a caller that alternates queues, as run_observation does: a production block is 64
    // tiles of the shared-window kernel, a quarter of the chip -- the tiles of consecutive blocks run side by side).  Push j works
    // in set j % 3: its chunk [n_dm][max_rows][beam] and the wide kernel's scratch (dsabf::kDmScratchBytes).
    //   rows_ready[j % 3]: the rows of push j -- and of every push before it -- are in the buffer (recorded on push j's queue behind
    //                      its producer and behind rows_ready of push j - 1): what push j + 1's kernels wait for, not push j's END;
    //   done[j % 3]:       push j and every push before it are complete, host copy included (recorded behind done of push j - 1).
    // Push j waits for done of push j - 3 (its set's previous user); so does the producer of push j's rows, which overwrites what
    // only pushes <= j - 3 can still be reading (cap_rows >= max_delay + 3 max_rows).  The linear buffer keeps one push at a time.
    float* d_out[3] = {nullptr, nullptr, nullptr};
    int* d_flags[3] = {nullptr, nullptr, nullptr};
    bool flags_zeroed[3] = {false, false, false};
    hipEvent_t done[3] = {nullptr, nullptr, nullptr}, rows_ready[3] = {nullptr, nullptr, nullptr};
    bool done_recorded[3] = {false, false, false}, rows_recorded[3] = {false, false, false};
    float* reserved = nullptr;    // bf_dm_stream_reserve: where the NEXT push's rows are being written by their producer ...
    int reserved_rows = 0;        // ... and how many (0: no reservation outstanding)
};

// device side of a DM stage (its handle's device must be current); the object itself stays, detached from the handle
static void dm_stream_release(bf_dm_stream* s)
{
    for (int k = 0; k < 3; k++) {
        if (s->done[k]) {
            if (s->done_recorded[k]) (void)hipEventSynchronize(s->done[k]);
            (void)hipEventDestroy(s->done[k]);
        }
        if (s->rows_ready[k]) (void)hipEventDestroy(s->rows_ready[k]);
        s->done[k] = s->rows_ready[k] = nullptr;
        s->done_recorded[k] = s->rows_recorded[k] = false;
    }
    if (s->ring || s->phys_created) {
        if (s->mapped0) (void)hipMemUnmap(s->d_buf, s->phys_bytes);
        if (s->mapped1) (void)hipMemUnmap(reinterpret_cast<char*>(s->d_buf) + s->phys_bytes, s->phys_bytes);
        if (s->phys_created) (void)hipMemRelease(s->phys);   // (the addresses go back to nobody: ring_address_space)
        s->mapped0 = s->mapped1 = s->phys_created = false;
    } else {
        (void)hipFree(s->d_buf);
    }
    for (int k = 0; k < 3; k++) {
        (void)hipFree(s->d_out[k]);
        (void)hipFree(s->d_flags[k]);
        s->d_out[k] = nullptr;
        s->d_flags[k] = nullptr;
    }
    (void)hipFree(s->d_delays);
    s->d_buf = nullptr;
    s->d_delays = nullptr;
    s->h = nullptr;
}

// Address space for the rings: taken from arenas that are reserved once per process and NEVER given back or handed out twice.
// On this stack (ROCm 7.2, gfx950) a virtual range that is unmapped and mapped again to other physical memory keeps stale
// translations: kernels and copies then disagree about where the rows are (tools/vmm_probe.cpp modes 0-4: wrong from the second
// ring on, whatever is freed, synchronised or allocated in between; modes 5-6, fresh addresses every time: always right --
// profiles/r06_vmm_probe.txt).  Addresses cost nothing (47 bits of them); a stage takes 2 x its ring's bytes.
static void* ring_address_space(size_t bytes, size_t gran)
{
    static std::mutex mu;
    static char* base = nullptr;
    static size_t size = 0, used = 0;
    std::lock_guard<std::mutex> lock(mu);
    used = (used + gran - 1) / gran * gran;
    if (!base || used + bytes > size) {
        void* va = nullptr;
        for (size_t want : {(size_t)256 << 30, (size_t)32 << 30, (size_t)4 << 30, bytes}) {
            if (want < bytes) continue;
            if (hipMemAddressReserve(&va, want, gran, nullptr, 0) == hipSuccess && va) {
                base = static_cast<char*>(va);
                size = want;
                used = 0;
                break;
            }
            (void)hipGetLastError();
            va = nullptr;
        }
        if (!va) return nullptr;
    }
    void* out = base + used;
    used += bytes;
    return out;
}

// The ring: one physical allocation, mapped at va and at va + phys_bytes.  False (and nothing left behind): no VMM here.
static bool dm_ring_create(bf_dm_stream* s, int device, size_t want_rows)
{
    int vmm = 0;
    if (hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, device) != hipSuccess || !vmm) return false;
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || !gran) return false;
    const size_t row_bytes = s->row_floats * sizeof(float);
    size_t g = gran, r = row_bytes;          // rows per granule-aligned stretch: gran / gcd(row_bytes, gran)
    while (r) {
        const size_t t = g % r;
        g = r;
        r = t;
    }
    const size_t step = gran / g;
    const size_t rows = (want_rows + step - 1) / step * step;
    s->phys_bytes = rows * row_bytes;
    void* va = ring_address_space(2 * s->phys_bytes, gran);
    bool ok = va != nullptr && hipMemCreate(&s->phys, s->phys_bytes, &prop, 0) == hipSuccess;
    s->phys_created = ok;
    s->d_buf = static_cast<float*>(va);
    ok = ok && (s->mapped0 = hipMemMap(va, s->phys_bytes, 0, s->phys, 0) == hipSuccess);
    ok = ok && (s->mapped1 = hipMemMap(static_cast<char*>(va) + s->phys_bytes, s->phys_bytes, 0, s->phys, 0) == hipSuccess);
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    ok = ok && hipMemSetAccess(va, 2 * s->phys_bytes, &acc, 1) == hipSuccess;
    if (!ok) {
        if (s->mapped0) (void)hipMemUnmap(va, s->phys_bytes);
        if (s->mapped1) (void)hipMemUnmap(static_cast<char*>(va) + s->phys_bytes, s->phys_bytes);
        if (s->phys_created) (void)hipMemRelease(s->phys);
        s->mapped0 = s->mapped1 = s->phys_created = false;
        s->d_buf = nullptr;
        (void)hipGetLastError();
        return false;
    }
    s->ring = true;
    s->cap_rows = rows;
    return true;
}

int bf_dm_stream_create(bf_handle* h, const int32_t* delays, int n_dm, int n_freq_total, int max_rows_per_push, bf_dm_stream** out)
{
    if (!out) return fail(BF_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!h || !delays) return fail(BF_ERR_INVALID, "NULL argument");
    if (n_dm <= 0 || n_freq_total <= 0 || max_rows_per_push <= 0) return fail(BF_ERR_INVALID, "need n_dm, n_freq_total, max_rows_per_push > 0");
    int dmax = 0;
    for (size_t i = 0; i < (size_t)n_dm * n_freq_total; i++) {
        if (delays[i] < 0) return fail(BF_ERR_INVALID, "a streamed dedispersion needs delays >= 0 (delay[%zu] = %d)", i, delays[i]);
        if (delays[i] > dmax) dmax = delays[i];
    }
    ON_DEVICE(h);
    bf_dm_stream* s = new (std::nothrow) bf_dm_stream();
    if (!s) return fail(BF_ERR_DEVICE, "out of host memory");
    s->h = h;
    s->n_dm = n_dm;
    s->n_freq = n_freq_total;
    s->max_delay = dmax;
    s->max_rows = max_rows_per_push;
    s->row_floats = (size_t)n_freq_total * h->cfg.n_beams;
    hipError_t e = hipSuccess;
    // the ring: the window of a push (<= max_delay + max_rows rows) + two more pushes' rows that may be written while it is read
    if (!h->dm_ring || !dm_ring_create(s, h->device, (size_t)dmax + 3 * (size_t)max_rows_per_push)) {
        // linear: room for the carry and a push twice over -- when the end is reached the carry moves to the start without overlapping itself
        s->cap_rows = 2 * ((size_t)dmax + (size_t)max_rows_per_push);
        e = hipMalloc((void**)&s->d_buf, s->cap_rows * s->row_floats * sizeof(float));
    }
    const int n_sets = s->ring ? 3 : 1;   // (the linear buffer keeps one push at a time: one chunk, one scratch)
    for (int k = 0; k < n_sets && e == hipSuccess; k++) {
        e = hipMalloc((void**)&s->d_out[k], (size_t)n_dm * max_rows_per_push * h->cfg.n_beams * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&s->d_flags[k], dsabf::kDmScratchBytes);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&s->d_delays, (size_t)n_dm * n_freq_total * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(s->d_delays, delays, (size_t)n_dm * n_freq_total * sizeof(int32_t), hipMemcpyHostToDevice);
    for (int k = 0; k < 3 && e == hipSuccess; k++) {
        e = hipEventCreateWithFlags(&s->done[k], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->rows_ready[k], hipEventDisableTiming);
    }
    h->dm_streams.push_back(s);
    if (e != hipSuccess) {
        const int rc = fail(BF_ERR_DEVICE, "bf_dm_stream_create: %s", hipGetErrorString(e));
        std::string keep = g_err;
        bf_dm_stream_destroy(s);
        g_err = keep;
        return rc;
    }
    *out = s;
    return BF_OK;
}

int bf_dm_stream_destroy(bf_dm_stream* s)
{
    if (!s) return BF_OK;
    if (s->h) {   // (NULL: the handle went first and took the device memory with it)
        bf_handle* h = s->h;
        DeviceScope dev_scope_(h->device);
        for (size_t i = 0; i < h->dm_streams.size(); i++)
            if (h->dm_streams[i] == s) {
                h->dm_streams.erase(h->dm_streams.begin() + (long)i);
                break;
            }
        dm_stream_release(s);
    }
    delete s;
    return BF_OK;
}

int bf_dm_stream_max_delay(const bf_dm_stream* s) { return s ? s->max_delay : BF_ERR_INVALID; }

int bf_dm_stream_output_device(bf_dm_stream* s, float** d_out)
{
    if (!s || !d_out) return fail(BF_ERR_INVALID, "NULL argument");
    if (!s->h) return fail(BF_ERR_STATE, "the handle of this DM stage has been destroyed");
    *d_out = s->d_out[s->ring && s->n_push ? (s->n_push - 1) % 3 : 0];   // the most recent push's chunk
    return BF_OK;
}

// Where the next n_rows rows go, with everything the writer of those rows must wait for queued on q first.
//   ring: behind the previous rows, wherever that is -- they overwrite rows that only pushes <= j - 3 can still be reading;
//   linear: behind the previous push, then behind the previous rows -- unless the buffer's end is reached: then the carry slides
//           back to the start first.
static int dm_place_rows(bf_dm_stream* s, int n_rows, hipStream_t q, float** dst)
{
    const size_t D = (size_t)s->max_delay;
    const size_t carry = s->pushed < D ? (size_t)s->pushed : D;
    if (s->ring) {
        const int old = (int)(s->n_push % 3);        // the slot push j will record into: last recorded by push j - 3
        if (s->done_recorded[old]) HIP_TRY(hipStreamWaitEvent(q, s->done[old], 0));
        *dst = s->d_buf + s->wpos * s->row_floats;   // (wpos + n_rows may pass cap_rows: the second mapping continues the first)
        return BF_OK;
    }
    // linear: the writer of the new rows runs behind the previous push, always -- after a slide the new rows walk into the area the
    // pushes before it read (and an earlier, still pending slide copies from), and only the chain of pushes orders those
    // (tools/fuzz_dm_stream.py without synchronisation between pushes found the version that waited only when sliding)
    const int prev = (int)((s->n_push + 2) % 3);
    if (s->n_push && s->done_recorded[prev]) HIP_TRY(hipStreamWaitEvent(q, s->done[prev], 0));
    if (s->fill + (size_t)n_rows > s->cap_rows) {    // slide: fill - carry >= carry here (cap = 2 (D + max_rows))
        if (carry)
            HIP_TRY(hipMemcpyAsync(s->d_buf, s->d_buf + (s->fill - carry) * s->row_floats, carry * s->row_floats * sizeof(float),
                                   hipMemcpyDeviceToDevice, q));
        s->fill = carry;                              // (the carry HAS moved: committed here, not at the push)
    }
    *dst = s->d_buf + s->fill * s->row_floats;
    return BF_OK;
}

// Zero-copy feed (round 6): the place of the next n_rows rows in the stage's own buffer, directly behind the carried-over window.
// The producer -- bf_enqueue_block_to, bf_gather_detected -- writes them there, ordered on (or behind) hip_stream; the push that
// follows finds them in place and only launches.  The reference's collapse sits directly behind detect, no copy in between
// (src/beamformer.cu:492-511); round 5's push copied every row device-to-device first (64 MiB read + 64 MiB written per production
// block).
int bf_dm_stream_reserve(bf_dm_stream* s, int n_rows, float** d_dst, void* hip_stream)
{
    if (!s || !d_dst) return fail(BF_ERR_INVALID, "NULL argument");
    *d_dst = nullptr;
    if (n_rows <= 0 || n_rows > s->max_rows) return fail(BF_ERR_INVALID, "n_rows must be 1 .. %d (max_rows_per_push)", s->max_rows);
    if (!s->h) return fail(BF_ERR_STATE, "the handle of this DM stage has been destroyed");
    if (s->reserved_rows) return fail(BF_ERR_STATE, "bf_dm_stream_reserve: the previous reservation has not been pushed");
    bf_handle* h = s->h;
    ON_DEVICE(h);
    float* dst = nullptr;
    if (int rc = dm_place_rows(s, n_rows, as_stream(hip_stream), &dst)) return rc;
    s->reserved = dst;
    s->reserved_rows = n_rows;
    *d_dst = dst;
    return BF_OK;
}

int bf_dm_stream_push(bf_dm_stream* s, const float* d_rows, int n_rows, float* host_out, uint64_t* first_t, int* n_t_out,
                      void* hip_stream)
{
    if (!s || !d_rows) return fail(BF_ERR_INVALID, "NULL argument");
    if (n_rows <= 0 || n_rows > s->max_rows) return fail(BF_ERR_INVALID, "n_rows must be 1 .. %d (max_rows_per_push)", s->max_rows);
    if (!s->h) return fail(BF_ERR_STATE, "the handle of this DM stage has been destroyed");
    const bool in_place = s->reserved_rows != 0;
    if (in_place && (d_rows != s->reserved || n_rows != s->reserved_rows))
        return fail(BF_ERR_STATE, "bf_dm_stream_push: %d rows are reserved at %p (bf_dm_stream_reserve); push exactly those", s->reserved_rows,
                    (void*)s->reserved);
    bf_handle* h = s->h;
    ON_DEVICE(h);
    hipStream_t q = as_stream(hip_stream);
    const int prev = (int)((s->n_push + 2) % 3), mine = (int)(s->n_push % 3);
    const int set = s->ring ? mine : 0;                                   // chunk + scratch this push works in
    if (s->ring) {
        // this set's previous user is push j - 3 (the producer of in-place rows waited for it too, on the stream it was given)
        if (s->done_recorded[mine]) HIP_TRY(hipStreamWaitEvent(q, s->done[mine], 0));
    } else if (s->n_push && s->done_recorded[prev]) {
        HIP_TRY(hipStreamWaitEvent(q, s->done[prev], 0));                 // linear: behind the previous push, whatever queue that ran on
    }
    if (!s->flags_zeroed[set]) {
        HIP_TRY(hipMemsetAsync(s->d_flags[set], 0, dsabf::kDmScratchBytes, q));
        s->flags_zeroed[set] = true;
    }
    // (the stream's bookkeeping -- wpos / fill, pushed -- is committed at the end: a call that fails on the way leaves it as it found it)
    const size_t D = (size_t)s->max_delay;
    const size_t carry = s->pushed < D ? (size_t)s->pushed : D;          // the rows in front of the new ones = series rows [pushed - carry, pushed)
    if (!in_place) {                                                      // rows that live elsewhere: brought behind the carry first
        float* dst = nullptr;
        if (int rc = dm_place_rows(s, n_rows, q, &dst)) return rc;
        HIP_TRY(hipMemcpyAsync(dst, d_rows, (size_t)n_rows * s->row_floats * sizeof(float), hipMemcpyDeviceToDevice, q));
    }
    if (s->ring) {
        // the kernels read [carry | new rows]: the carry was written by the producers of the pushes before this one, possibly on
        // other queues -- wait until THEIR rows are in place (not for their dedispersion), then say that ours are
        if (s->n_push && s->rows_recorded[prev]) HIP_TRY(hipStreamWaitEvent(q, s->rows_ready[prev], 0));
        HIP_TRY(hipEventRecord(s->rows_ready[mine], q));
        s->rows_recorded[mine] = true;
    }
    const uint64_t emitted = s->pushed > D ? s->pushed - D : 0;          // output times [0, emitted) have been produced
    const uint64_t after = s->pushed + (uint64_t)n_rows;
    const uint64_t complete = after > D ? after - D : 0;                   // ... and [0, complete) can be now
    const int n_out = (int)(complete - emitted);
    const size_t n_t = carry + (size_t)n_rows;                            // the series the kernels see: starts at output time `emitted`
    // first row of [carry | new rows]: linear: fill - carry; ring: wpos - carry, through the second mapping when that is negative
    const size_t start = s->ring ? (s->wpos >= carry ? s->wpos - carry : s->wpos + s->cap_rows - carry) : s->fill - carry;
    if (n_out > 0) {
        dsabf::Geometry g = h->geom;
        g.n_freq = s->n_freq;
        HIP_TRY(dsabf::launch_dedisperse_dm(g, s->d_buf + start * s->row_floats, (int)n_t, s->d_delays, s->n_dm, n_out, s->d_out[set],
                                            s->d_flags[set], q));
        if (host_out)
            HIP_TRY(hipMemcpyAsync(host_out, s->d_out[set], (size_t)s->n_dm * n_out * h->cfg.n_beams * sizeof(float), hipMemcpyDeviceToHost, q));
    }
    // the end of push j implies the end of every push before it (chunks leave in order; a producer that waits for push j - 3 knows
    // that nothing older reads the rows it overwrites)
    if (s->ring && s->n_push && s->done_recorded[prev]) HIP_TRY(hipStreamWaitEvent(q, s->done[prev], 0));
    HIP_TRY(hipEventRecord(s->done[mine], q));
    s->done_recorded[mine] = true;
    if (s->ring)
        s->wpos = (s->wpos + (size_t)n_rows) % s->cap_rows;
    else
        s->fill += (size_t)n_rows;
    s->pushed = after;
    s->n_push++;
    s->reserved = nullptr;
    s->reserved_rows = 0;
    if (first_t) *first_t = emitted;
    if (n_t_out) *n_t_out = n_out;
    return BF_OK;
}

int bf_set_switch(bf_handle* h, const char* name, int value)
{
    if (!h || !name) return fail(BF_ERR_INVALID, "NULL argument");
    ON_DEVICE(h);
    if (!strcmp(name, "tsplit")) {
        if (value < 0) return fail(BF_ERR_INVALID, "tsplit must be >= 0 (0: the library decides)");
        h->geom.tsplit = value;
    } else if (!strcmp(name, "rtw_kout")) {
        if (value < 0 || value > 32) return fail(BF_ERR_INVALID, "rtw_kout must be 0 .. 32 (0: the library decides)");
        h->geom.rtw_kout = value;
    } else if (!strcmp(name, "lds_pad")) {
        if (value < 0 || value > dsabf::kLdsPerCuBytes) return fail(BF_ERR_INVALID, "lds_pad must be 0 .. %d bytes", dsabf::kLdsPerCuBytes);
        h->geom.lds_pad = value;
    } else if (!strcmp(name, "dm_wide")) {
        h->geom.dm_wide = value != 0;
    } else if (!strcmp(name, "coalesce")) {
        FLUSH_UNITS(h);
        h->coalesce = value != 0;
    } else if (!strcmp(name, "paired")) {
        h->force_general = value == 0;   // takes effect at the next bf_set_weights (the kernel is chosen per weight set)
    } else if (!strcmp(name, "dm_ring")) {
        h->dm_ring = value != 0;         // takes effect at the next bf_dm_stream_create
    } else {
        return fail(BF_ERR_INVALID, "unknown switch \"%s\" (tsplit, rtw_kout, lds_pad, dm_wide, dm_ring, paired, coalesce)", name);
    }
    return BF_OK;
}

int bf_get_counter(const bf_handle* h, const char* name, uint64_t* value)
{
    if (!h || !name || !value) return fail(BF_ERR_INVALID, "NULL argument");
    if (!strcmp(name, "fused_launches"))
        *value = h->n_fused_launches;
    else if (!strcmp(name, "queued_units"))
        *value = h->pending.size();
    else if (!strcmp(name, "dm_ring_stages")) {   // live DM stages of this handle whose buffer is the twice-mapped ring (the rest: linear)
        uint64_t n = 0;
        for (const bf_dm_stream* s : h->dm_streams) n += s->ring ? 1 : 0;
        *value = n;
    } else
        return fail(BF_ERR_INVALID, "unknown counter \"%s\" (fused_launches, queued_units, dm_ring_stages)", name);
    return BF_OK;
}

int bf_kernel_info(const bf_handle* h, int n_units, int* grid, int* block, int* lds_bytes, int* vgprs)
{
    if (!h) return fail(BF_ERR_INVALID, "handle is NULL");
    const dsabf::LaunchShape ls = dsabf::fused_launch_shape(h->geom, n_units > 0 ? n_units : 1, h->n_cus);
    if (grid) *grid = ls.grid;
    if (block) *block = ls.block;
    if (lds_bytes) *lds_bytes = ls.lds_bytes;
    if (vgprs) *vgprs = dsabf::fused_vgprs(h->geom);
    return BF_OK;
}

int bf_launch_plan(const bf_config* cfg, int paired, int n_units, int n_cus, int* grid, int* block, int* lds_bytes, char* name,
                   size_t name_len)
{
    if (int rc = check_cfg(cfg)) return rc;
    dsabf::Geometry g = make_geom(*cfg);
    const char* why = nullptr;
    if (!dsabf::fused_supported(g, &why)) return fail(BF_ERR_INVALID, "unsupported geometry: %s", why);
    if (n_units <= 0 || n_cus <= 0) return fail(BF_ERR_INVALID, "need n_units > 0 and n_cus > 0");
    g.paired = paired && dsabf::pairing_supported(g);   // what bf_set_weights decides for a conjugate-symmetric weight set
    const dsabf::LaunchShape ls = dsabf::fused_launch_shape(g, n_units, n_cus);
    if (grid) *grid = ls.grid;
    if (block) *block = ls.block;
    if (lds_bytes) *lds_bytes = ls.lds_bytes;
    if (name && name_len) dsabf::fused_kernel_name(g, name, name_len);
    return BF_OK;
}

int bf_rtw_plan(const bf_config* cfg, int n_units, int n_cus, int* windows_per_stream, int* chunks_total)
{
    if (int rc = check_cfg(cfg)) return rc;
    const dsabf::Geometry g = make_geom(*cfg);
    const char* why = nullptr;
    if (!dsabf::fused_supported(g, &why)) return fail(BF_ERR_INVALID, "unsupported geometry: %s", why);
    if (n_units <= 0 || n_cus <= 0) return fail(BF_ERR_INVALID, "need n_units > 0 and n_cus > 0");
    const dsabf::LaunchShape ls = dsabf::fused_launch_shape(g, n_units, n_cus);
    if (windows_per_stream) *windows_per_stream = ls.rt_kout;
    if (chunks_total) *chunks_total = ls.chunks_total;
    return BF_OK;
}

int bf_kernel_name(const bf_handle* h, char* buf, size_t buflen)
{
    if (!h || !buf || !buflen) return fail(BF_ERR_INVALID, "NULL argument");
    dsabf::fused_kernel_name(h->geom, buf, buflen);
    return BF_OK;
}

int bf_variant_key(const bf_config* cfg, int paired, int write_c, char* buf, size_t buflen)
{
    if (!buf || !buflen) return fail(BF_ERR_INVALID, "NULL argument");
    if (int rc = check_cfg(cfg)) return rc;
    dsabf::Geometry g = make_geom(*cfg);
    const char* why = nullptr;
    if (!dsabf::fused_supported(g, &why)) return fail(BF_ERR_INVALID, "unsupported geometry: %s", why);
    g.paired = paired && dsabf::pairing_supported(g);
    dsabf::fused_variant_key(g, write_c != 0, buf, buflen);
    return BF_OK;
}

int bf_handle_variant_key(const bf_handle* h, int write_c, char* buf, size_t buflen)
{
    if (!h || !buf || !buflen) return fail(BF_ERR_INVALID, "NULL argument");
    dsabf::fused_variant_key(h->geom, write_c != 0, buf, buflen);
    return BF_OK;
}

}  // extern "C"
