// bf_host_c.cpp -- C wrappers (include/dsabf_host.h) over the C++ host mirror (include/dsabf_host.hpp): what ctypes / cgo /
// a C program binds.
#include "../../include/dsabf_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <thread>
#include <unistd.h>

#include "../../include/dsabf_host.h"
#include "bf_host_internal.h"

using namespace dsabf;

struct bfh_generator {
    test_data_generator* g;
    bf_config cfg;
};
struct bfh_obs {
    observation_loop_state* o;
    event_backend* backend;
};

namespace {
// event_backend over a table of C callbacks (bfh_obs_create_custom)
struct callback_backend : event_backend {
    bfh_event_ops ops;
    explicit callback_backend(const bfh_event_ops& o) : ops(o) {}
    void* create() override { return ops.create(ops.user); }
    void destroy(void* ev) override { ops.destroy(ops.user, ev); }
    int record_transfer(void* ev) override { return ops.record_transfer(ops.user, ev); }
    int record_analysis(void* ev) override { return ops.record_analysis(ops.user, ev); }
    int query(void* ev) override { return ops.query(ops.user, ev); }
};
}  // namespace

static std::vector<antenna> to_antennas(const float* pos, int n)
{
    std::vector<antenna> a((size_t)n);
    for (int i = 0; i < n; i++) {
        a[i].x = pos[3 * i];
        a[i].y = pos[3 * i + 1];
        a[i].z = pos[3 * i + 2];
    }
    return a;
}
static std::vector<beam_direction> to_dirs(const float* d, int n)
{
    std::vector<beam_direction> v((size_t)n);
    for (int i = 0; i < n; i++) v[i] = beam_direction(d[2 * i], d[2 * i + 1]);
    return v;
}

extern "C" {

int bfh_default_positions(int n_ant, float* pos)
{
    if (!pos || n_ant <= 0) return BF_ERR_INVALID;
    std::vector<antenna> a((size_t)n_ant);
    default_positions(n_ant, a.data());
    for (int i = 0; i < n_ant; i++) {
        pos[3 * i] = a[i].x;
        pos[3 * i + 1] = a[i].y;
        pos[3 * i + 2] = a[i].z;
    }
    return BF_OK;
}

int bfh_default_directions(int n_beams, float* dir)
{
    if (!dir || n_beams <= 0) return BF_ERR_INVALID;
    std::vector<beam_direction> d((size_t)n_beams);
    default_directions(n_beams, d.data());
    for (int i = 0; i < n_beams; i++) {
        dir[2 * i] = d[i].theta;
        dir[2 * i + 1] = d[i].phi;
    }
    return BF_OK;
}

int bfh_read_positions(const char* path, int n_ant, float* pos)
{
    if (!path || !pos) return BF_ERR_INVALID;
    std::vector<antenna> a((size_t)n_ant);
    if (read_in_position_locations(path, n_ant, a.data()) != 0) return BF_ERR_INVALID;
    for (int i = 0; i < n_ant; i++) {
        pos[3 * i] = a[i].x;
        pos[3 * i + 1] = a[i].y;
        pos[3 * i + 2] = a[i].z;
    }
    return BF_OK;
}

int bfh_read_directions(const char* path, int expected, float* dir)
{
    if (!path || !dir) return BF_ERR_INVALID;
    std::vector<beam_direction> d((size_t)expected);
    if (read_in_beam_directions(path, expected, d.data()) != 0) return BF_ERR_INVALID;
    for (int i = 0; i < expected; i++) {
        dir[2 * i] = d[i].theta;
        dir[2 * i + 1] = d[i].phi;
    }
    return BF_OK;
}

int bfh_count_entries(const char* path)
{
    std::ifstream f(path);
    if (!f.is_open()) return BF_ERR_INVALID;
    int n = 0;
    f >> n;
    return n;
}

int bfh_write_python_file(const float* data, int rows, int cols, const char* path)
{
    if (!data || !path) return BF_ERR_INVALID;
    return write_array_to_disk_as_python_file(data, rows, cols, path) == 0 ? BF_OK : BF_ERR_INVALID;
}

float bfh_channel_frequency(int generator_variant, int gpu, int chan)
{
    return generator_variant ? channel_frequency_generator(gpu, chan) : channel_frequency_weights(gpu, chan);
}

int bfh_make_weights(int n_beams, int n_ant, int n_freq, int chan0, int gpu, const float* pos, const float* dir,
                     int8_t* out)
{
    if (!pos || !dir || !out || n_beams <= 0 || n_ant <= 0 || n_freq <= 0) return BF_ERR_INVALID;
    auto a = to_antennas(pos, n_ant);
    auto d = to_dirs(dir, n_beams);
    generate_fourier_coefficients(n_beams, n_ant, n_freq, chan0, gpu, a.data(), d.data(), out);
    return BF_OK;
}

int bfh_gen_create(const bf_config* cfg, int per_batch, int pin, bfh_generator** out)
{
    if (!cfg || !out || per_batch <= 0) return BF_ERR_INVALID;
    bfh_generator* g = new bfh_generator{new test_data_generator(*cfg, per_batch, pin != 0), *cfg};
    if (!g->g->get_data()) {
        delete g->g;
        delete g;
        return BF_ERR_DEVICE;
    }
    *out = g;
    return BF_OK;
}
int bfh_gen_destroy(bfh_generator* g)
{
    if (g) {
        delete g->g;
        delete g;
    }
    return BF_OK;
}
int bfh_gen_read_sources(bfh_generator* g, const char* path)
{
    return (g && path && g->g->read_in_source_directions(path) == 0) ? BF_OK : BF_ERR_INVALID;
}
int bfh_gen_set_sources(bfh_generator* g, const float* src, int n)
{
    if (!g || !src || n < 0) return BF_ERR_INVALID;
    auto d = to_dirs(src, n);
    g->g->set_source_directions(d.data(), n);
    return BF_OK;
}
int bfh_gen_generate(bfh_generator* g, const float* pos, int gpu)
{
    if (!g || !pos) return BF_ERR_INVALID;
    auto a = to_antennas(pos, g->cfg.n_ant);
    g->g->generate_test_data(a.data(), gpu);
    return BF_OK;
}
void* bfh_gen_data(bfh_generator* g) { return g ? g->g->get_data() : nullptr; }
size_t bfh_gen_size(bfh_generator* g) { return g ? g->g->input_data_size() : 0; }
int bfh_gen_n_pt_sources(bfh_generator* g) { return g ? g->g->get_n_pt_sources() : BF_ERR_INVALID; }
int bfh_gen_need_more(bfh_generator* g, int bt) { return g ? g->g->check_need_to_generate_more_input_data(bt) : BF_ERR_INVALID; }
int bfh_gen_ready(bfh_generator* g, int tq) { return g ? g->g->check_data_ready_for_transfer(tq) : BF_ERR_INVALID; }

int bfh_obs_create(uint64_t mts, uint64_t mtot, const bf_config* cfg, bf_handle* h, int debug_mode, bfh_obs** out)
{
    if (!cfg || !out || !h) return BF_ERR_INVALID;
    bfh_obs* o = new bfh_obs{nullptr, make_hip_event_backend(h)};
    o->o = new observation_loop_state(mts, mtot, *cfg, o->backend, debug_mode != 0);
    *out = o;
    return BF_OK;
}
int bfh_obs_create_custom(uint64_t mts, uint64_t mtot, const bf_config* cfg, const bfh_event_ops* ops, int debug_mode,
                          bfh_obs** out)
{
    if (!cfg || !out || !ops || !ops->create || !ops->destroy || !ops->record_transfer || !ops->record_analysis || !ops->query)
        return BF_ERR_INVALID;
    bfh_obs* o = new bfh_obs{nullptr, new callback_backend(*ops)};
    o->o = new observation_loop_state(mts, mtot, *cfg, o->backend, debug_mode != 0);
    *out = o;
    return BF_OK;
}
int bfh_obs_status(bfh_obs* o) { return o ? o->o->status() : BF_ERR_INVALID; }
int bfh_obs_destroy(bfh_obs* o)
{
    if (o) {
        delete o->o;
        delete o->backend;
        delete o;
    }
    return BF_OK;
}
int bfh_obs_generate_transfer_event(bfh_obs* o) { return o ? (o->o->generate_transfer_event(), o->o->status()) : BF_ERR_INVALID; }
int bfh_obs_generate_analysis_event(bfh_obs* o) { return o ? (o->o->generate_analysis_event(), o->o->status()) : BF_ERR_INVALID; }
int bfh_obs_check_transfer_events(bfh_obs* o) { return o ? (o->o->check_transfer_events(), o->o->status()) : BF_ERR_INVALID; }
int bfh_obs_check_analysis_events(bfh_obs* o) { return o ? (o->o->check_analysis_events(), o->o->status()) : BF_ERR_INVALID; }
int bfh_obs_counters(bfh_obs* o, uint64_t* A, uint64_t* AQ, uint64_t* T, uint64_t* TQ)
{
    if (!o) return BF_ERR_INVALID;
    if (A) *A = o->o->get_blocks_analyzed();
    if (AQ) *AQ = o->o->get_blocks_analysis_queue();
    if (T) *T = o->o->get_blocks_transferred();
    if (TQ) *TQ = o->o->get_blocks_transfer_queue();
    return BF_OK;
}
int bfh_obs_check_ready_for_transfer(bfh_obs* o) { return o ? o->o->check_ready_for_transfer() : BF_ERR_INVALID; }
int bfh_obs_check_ready_for_analysis(bfh_obs* o) { return o ? o->o->check_ready_for_analysis() : BF_ERR_INVALID; }
int bfh_obs_check_ready_for_dh2_transfer(bfh_obs* o, int ts) { return o ? o->o->check_ready_for_dh2_transfer(ts) : BF_ERR_INVALID; }
int bfh_obs_check_observations_complete(bfh_obs* o)
{
    if (!o) return BF_ERR_INVALID;
    std::streambuf* keep = std::cout.rdbuf();
    std::ostringstream sink;  // the reference prints "obs Complete"; keep wrapper callers' stdout clean
    std::cout.rdbuf(sink.rdbuf());
    const bool r = o->o->check_observations_complete();
    std::cout.rdbuf(keep);
    return r;
}
int bfh_obs_check_transfers_complete(bfh_obs* o) { return o ? o->o->check_transfers_complete() : BF_ERR_INVALID; }
int bfh_obs_set_transfers_complete(bfh_obs* o, int v) { return o ? (o->o->set_transfers_complete(v != 0), BF_OK) : BF_ERR_INVALID; }
int bfh_obs_set_n_pt_sources(bfh_obs* o, int n) { return o ? (o->o->set_n_pt_sources(n), BF_OK) : BF_ERR_INVALID; }
uint64_t bfh_obs_get_current_analysis_gemm(bfh_obs* o, int ts) { return o ? o->o->get_current_analysis_gemm(ts) : 0; }
uint64_t bfh_obs_get_current_transfer_gemm(bfh_obs* o) { return o ? o->o->get_current_transfer_gemm() : 0; }
uint64_t bfh_obs_get_next_gpu_analysis_block(bfh_obs* o) { return o ? o->o->get_next_gpu_analysis_block() : 0; }
uint64_t bfh_obs_get_next_gpu_transfer_block(bfh_obs* o) { return o ? o->o->get_next_gpu_transfer_block() : 0; }
int bfh_obs_describe(bfh_obs* o, char* buf, size_t buflen)
{
    if (!o || !buf || !buflen) return BF_ERR_INVALID;
    std::ostringstream ss;
    ss << *o->o;
    snprintf(buf, buflen, "%s", ss.str().c_str());
    return BF_OK;
}
int bfh_run_debug_observation(const bf_config* cfg, int gpu, const char* positions, const char* directions,
                              const char* sources, const char* output, int device, int verbose, float* ded_out,
                              size_t ded_capacity, int* n_pt_sources, float* observation_ms)
{
    return bfh_run_debug_observation2(cfg, gpu, positions, directions, sources, output, device, verbose, ded_out, ded_capacity,
                                      n_pt_sources, observation_ms, 0);
}
int bfh_run_debug_observation2(const bf_config* cfg, int gpu, const char* positions, const char* directions,
                               const char* sources, const char* output, int device, int verbose, float* ded_out,
                               size_t ded_capacity, int* n_pt_sources, float* observation_ms, int per_unit_launches)
{
    if (!cfg) return BF_ERR_INVALID;
    debug_run_options opt;
    opt.block_launch = per_unit_launches == 0;
    opt.gpu = gpu;
    opt.positions = positions;
    opt.directions = directions;
    opt.sources = sources;
    opt.output = output;
    opt.device = device;
    opt.verbose = verbose != 0;
    debug_run_result res;
    std::vector<float> ded;
    std::ostringstream quiet;
    int rc = run_debug_observation(*cfg, opt, &res, &ded, verbose ? static_cast<std::ostream&>(std::cout) : quiet);
    if (rc != BF_OK) return rc;
    if (n_pt_sources) *n_pt_sources = res.n_pt_sources;
    if (observation_ms) *observation_ms = res.observation_time_ms;
    if (ded_out) {
        if (ded.size() > ded_capacity) return BF_ERR_INVALID;
        std::memcpy(ded_out, ded.data(), ded.size() * sizeof(float));
    }
    return BF_OK;
}

static int run_junk(const bf_config* cfg, uint64_t n_blocks, int ring_blocks, uint64_t seed, int gpu, int device,
                    int burn_in, int verbose, detected_sink* sink, observation_result* res, void* ring_copy,
                    const int32_t* dm_delays = nullptr, int n_dm = 0, dm_chunk_sink* dm_sink = nullptr)
{
    junk_block_source src(*cfg, n_blocks, ring_blocks, seed);
    if (!src.ok()) return BF_ERR_DEVICE;
    std::vector<antenna> pos((size_t)cfg->n_ant);
    std::vector<beam_direction> dir((size_t)cfg->n_beams);
    default_positions(cfg->n_ant, pos.data());
    default_directions(cfg->n_beams, dir.data());
    observation_options opt;
    opt.gpu = gpu;
    opt.device = device;
    opt.burn_in = burn_in;
    opt.verbose = verbose != 0;
    opt.sink = sink;
    opt.dm_delays = dm_delays;
    opt.n_dm = n_dm;
    opt.dm_sink = dm_sink;
    std::ostringstream quiet;
    std::streambuf* keep = std::cout.rdbuf();
    if (!verbose) std::cout.rdbuf(quiet.rdbuf());  // "obs Complete" etc.
    int rc = run_observation(*cfg, opt, src, pos.data(), dir.data(), res, verbose ? static_cast<std::ostream&>(std::cout) : quiet);
    std::cout.rdbuf(keep);
    if (rc == BF_OK && ring_copy) std::memcpy(ring_copy, src.ring_data(), (size_t)src.get_block_size() * src.get_ring_blocks());
    return rc;
}

int bfh_run_observation_junk(const bf_config* cfg, uint64_t n_blocks, int ring_blocks, uint64_t seed, int gpu, int device,
                             int burn_in, int verbose, float* observation_ms, float* beam_out, long long* last_gemm,
                             void* ring_copy)
{
    if (!cfg) return BF_ERR_INVALID;
    observation_result res;
    int rc = run_junk(cfg, n_blocks, ring_blocks, seed, gpu, device, burn_in, verbose, nullptr, &res, ring_copy);
    if (rc != BF_OK) return rc;
    if (observation_ms) *observation_ms = res.observation_time_ms;
    if (beam_out) std::memcpy(beam_out, res.beam_out.data(), res.beam_out.size() * sizeof(float));
    if (last_gemm) std::memcpy(last_gemm, res.last_gemm.data(), res.last_gemm.size() * sizeof(long long));
    return BF_OK;
}

int bfh_run_observation_junk_to_file(const bf_config* cfg, uint64_t n_blocks, int ring_blocks, uint64_t seed, int gpu,
                                     int device, int burn_in, int verbose, const char* path, float* observation_ms,
                                     uint64_t* gemms_written, void* ring_copy)
{
    if (!cfg || !path) return BF_ERR_INVALID;
    file_sink sink(*cfg, path, gpu);
    if (!sink.ok() || !sink.is_open()) return BF_ERR_INVALID;
    observation_result res;
    int rc = run_junk(cfg, n_blocks, ring_blocks, seed, gpu, device, burn_in, verbose, &sink, &res, ring_copy);
    if (rc != BF_OK) return rc;
    if (observation_ms) *observation_ms = res.observation_time_ms;
    if (gemms_written) *gemms_written = sink.get_delivered();
    return BF_OK;
}

int bfh_run_observation_junk_dm(const bf_config* cfg, uint64_t n_blocks, int ring_blocks, uint64_t seed, int gpu, int device,
                                int burn_in, int verbose, const int32_t* delays, int n_dm, const char* dm_path,
                                const char* detected_path, float* observation_ms, uint64_t* dm_times, void* ring_copy)
{
    if (!cfg || !delays || n_dm <= 0) return BF_ERR_INVALID;
    int dmax = 0;
    for (size_t i = 0; i < (size_t)n_dm * cfg->n_freq; i++) dmax = delays[i] > dmax ? delays[i] : dmax;
    std::unique_ptr<dm_chunk_sink> dms;
    if (dm_path && !std::strncmp(dm_path, "ring:", 5)) {   // "ring:<name>[:<blocks>]": the chunks to another process (dm_ring_sink)
        std::string nm(dm_path + 5);
        uint64_t blocks = 4;
        const size_t colon = nm.find(':');
        if (colon != std::string::npos) {
            blocks = std::strtoull(nm.c_str() + colon + 1, nullptr, 10);
            nm.resize(colon);
        }
        dm_ring_sink* rs = new dm_ring_sink(*cfg, cfg->n_freq, n_dm, dmax, cfg->n_gemms_per_block * cfg->n_out_per_gemm, nm.c_str(), blocks, gpu);
        dms.reset(rs);
        if (!rs->is_open()) return BF_ERR_INVALID;
    } else if (dm_path) {
        dm_file_sink* fs2 = new dm_file_sink(*cfg, cfg->n_freq, n_dm, dmax, dm_path, gpu);
        dms.reset(fs2);
        if (!fs2->is_open()) return BF_ERR_INVALID;
    }
    std::unique_ptr<file_sink> fs;
    if (detected_path) {
        fs.reset(new file_sink(*cfg, detected_path, gpu));
        if (!fs->ok() || !fs->is_open()) return BF_ERR_INVALID;
    }
    observation_result res;
    int rc = run_junk(cfg, n_blocks, ring_blocks, seed, gpu, device, burn_in, verbose, fs.get(), &res, ring_copy, delays, n_dm, dms.get());
    if (rc != BF_OK) return rc;
    if (observation_ms) *observation_ms = res.observation_time_ms;
    if (dm_times) *dm_times = res.dm_times;
    return BF_OK;
}

// One frequency shard of a sharded observation (what `beam -R world -r rank` runs), with any geometry: communicator from the id,
// the junk source (every shard reads the same bytes with its local geometry), the gather after every block in either transport,
// to one root or to every rank, the DM stage on the root(s) or split by trials, file sinks where this rank holds the band.
int bfh_run_observation_junk_sharded(const bf_config* cfg, uint64_t n_blocks, int ring_blocks, uint64_t seed, int gpu, int device, int rank,
                                     int world, const void* id128, int gather_root, int staged, const int32_t* delays, int n_dm,
                                     int split_trials, const char* detected_path, const char* dm_path, float* observation_ms,
                                     uint64_t* dm_times, void* ring_copy)
{
    if (!cfg || world < 1 || rank < 0 || rank >= world) return BF_ERR_INVALID;
    bf_comm* comm = nullptr;
    int rc = bf_comm_create(rank, world, id128, device, &comm);
    if (rc != BF_OK) return rc;
    struct comm_guard {
        bf_comm* c;
        ~comm_guard() { bf_comm_destroy(c); }
    } cg{comm};
    const bool holds_band = gather_root == BF_GATHER_ROOT_ALL || gather_root == rank;
    bf_config full = *cfg;
    full.n_freq = cfg->n_freq * world;
    // (a failed preparation of THIS rank does not return here: the other ranks would wait for it in the first gather.  It is
    //  reported to run_observation, which lets every shard know before anything starts -- observation_options::local_setup_ok)
    bool setup_ok = true;
    std::unique_ptr<file_sink> fs;
    if (detected_path && holds_band) {
        fs.reset(new file_sink(full, detected_path, gpu));
        if (!fs->ok() || !fs->is_open()) {
            setup_ok = false;
            fs.reset();
        }
    }
    int first = 0, count = n_dm;
    if (split_trials) dm_trial_share(n_dm, world, rank, &first, &count);
    std::unique_ptr<dm_file_sink> dms;
    if (delays && dm_path && holds_band && count > 0) {
        int dmax = 0;
        for (size_t i = (size_t)first * full.n_freq; i < (size_t)(first + count) * full.n_freq; i++) dmax = delays[i] > dmax ? delays[i] : dmax;
        dms.reset(new dm_file_sink(*cfg, full.n_freq, count, dmax, dm_path, gpu, first));
        if (!dms->is_open()) {
            setup_ok = false;
            dms.reset();
        }
    }
    junk_block_source src(*cfg, n_blocks, ring_blocks, seed);
    if (!src.ok()) setup_ok = false;
    std::vector<antenna> pos((size_t)cfg->n_ant);
    std::vector<beam_direction> dir((size_t)cfg->n_beams);
    default_positions(cfg->n_ant, pos.data());
    default_directions(cfg->n_beams, dir.data());
    observation_options opt;
    opt.gpu = gpu;
    opt.device = device;
    opt.world = world;
    opt.rank = rank;
    opt.comm = comm;
    opt.gather_root = gather_root;
    opt.gather_staged = staged != 0;
    opt.sink = fs.get();
    opt.dm_delays = delays;
    opt.n_dm = delays ? n_dm : 0;
    opt.dm_split_trials = split_trials != 0;
    opt.dm_sink = dms.get();
    opt.local_setup_ok = setup_ok;
    observation_result res;
    std::ostringstream quiet;
    std::streambuf* keep = std::cout.rdbuf();
    std::cout.rdbuf(quiet.rdbuf());
    rc = run_observation(*cfg, opt, src, pos.data(), dir.data(), &res, quiet);
    std::cout.rdbuf(keep);
    if (rc != BF_OK) return rc;
    if (ring_copy) std::memcpy(ring_copy, src.ring_data(), (size_t)src.get_block_size() * src.get_ring_blocks());
    if (observation_ms) *observation_ms = res.observation_time_ms;
    if (dm_times) *dm_times = res.dm_times;
    return BF_OK;
}

int bfh_dm_trials(double dm0, double dm_max, int nchan, double epsilon, double nu_ghz, double chan_bw_mhz, double ti_us,
                  double tscat_us, double tsamp_us, double* out, int cap)
{
    if (!out || cap < 1) return BF_ERR_INVALID;
    const std::vector<double> v = dm_trials(dm0, dm_max, nchan, epsilon, nu_ghz, chan_bw_mhz, ti_us, tscat_us, tsamp_us);
    const int n = (int)v.size() < cap ? (int)v.size() : cap;
    std::memcpy(out, v.data(), (size_t)n * sizeof(double));
    return n;
}

int bfh_dm_delays(const double* dms, int n_dm, const float* freq_ghz, int n_freq, double f_ref_ghz, double tsamp_ms,
                  int32_t* out)
{
    if (!dms || !freq_ghz || !out || n_dm < 0 || n_freq < 0) return BF_ERR_INVALID;
    dm_delays(dms, n_dm, freq_ghz, n_freq, f_ref_ghz, tsamp_ms, out);
    return BF_OK;
}

int bfh_dm_trial_share(int n_dm, int world, int rank, int* first, int* count)
{
    if (n_dm < 0 || world < 1 || rank < 0 || rank >= world) return BF_ERR_INVALID;
    dm_trial_share(n_dm, world, rank, first, count);
    return BF_OK;
}

int bfh_junk_fill(const bf_config* cfg, int ring_blocks, uint64_t seed, void* out)
{
    if (!cfg || !out || ring_blocks < 1) return BF_ERR_INVALID;
    junk_fill(*cfg, ring_blocks, seed, static_cast<char*>(out));
    return BF_OK;
}

struct bfh_shm_ring {
    shm_ring* r;
};

int bfh_shm_ring_create(const char* name, uint64_t n_blocks, uint64_t block_size, const char* header_text,
                        bfh_shm_ring** out)
{
    if (!name || !out) return BF_ERR_INVALID;
    shm_ring* r = shm_ring::create(name, n_blocks, block_size, header_text);
    if (!r) return BF_ERR_INVALID;
    *out = new bfh_shm_ring{r};
    return BF_OK;
}
int bfh_shm_ring_attach(const char* name, int timeout_ms, bfh_shm_ring** out)
{
    if (!name || !out) return BF_ERR_INVALID;
    shm_ring* r = shm_ring::attach(name, timeout_ms);
    if (!r) return BF_ERR_STATE;
    *out = new bfh_shm_ring{r};
    return BF_OK;
}
int bfh_shm_ring_detach(bfh_shm_ring* r)
{
    if (!r) return BF_OK;
    delete r->r;
    delete r;
    return BF_OK;
}
int bfh_shm_ring_unlink(const char* name) { return name && shm_ring::unlink(name) == 0 ? BF_OK : BF_ERR_INVALID; }
int bfh_shm_ring_info(bfh_shm_ring* r, uint64_t* n_blocks, uint64_t* block_size, char* header, size_t header_cap)
{
    if (!r) return BF_ERR_INVALID;
    if (n_blocks) *n_blocks = r->r->get_n_blocks();
    if (block_size) *block_size = r->r->get_block_size();
    if (header && header_cap) {
        std::strncpy(header, r->r->get_header(), header_cap - 1);
        header[header_cap - 1] = 0;
    }
    return BF_OK;
}
int bfh_shm_ring_write(bfh_shm_ring* r, const void* data, uint64_t bytes)
{
    if (!r || bytes > r->r->get_block_size() || (bytes && !data)) return BF_ERR_INVALID;
    char* b = r->r->open_block_write();
    if (!b) return BF_ERR_STATE;
    if (bytes) std::memcpy(b, data, bytes);
    r->r->close_block_write(bytes);
    return BF_OK;
}
int bfh_shm_ring_read(bfh_shm_ring* r, void* out, uint64_t cap, uint64_t* bytes, uint64_t* block_id)
{
    if (!r) return BF_ERR_INVALID;
    uint64_t n = 0, id = 0;
    char* b = r->r->open_block_read(&n, &id);
    if (!b) return BF_ERR_STATE;
    if (out) std::memcpy(out, b, n < cap ? n : cap);
    r->r->close_block_read();
    if (bytes) *bytes = n;
    if (block_id) *block_id = id;
    return BF_OK;
}

int bfh_run_observation_shm(const bf_config* cfg, const char* name, int core, int gpu, int device, int verbose,
                            const char* path, float* observation_ms, uint64_t* gemms_written, int* pinned)
{
    return bfh_run_observation_shm_dm(cfg, name, core, gpu, device, verbose, path, nullptr, 0, nullptr, observation_ms, gemms_written,
                                      nullptr, pinned);
}

int bfh_run_observation_shm_dm(const bf_config* cfg, const char* name, int core, int gpu, int device, int verbose, const char* path,
                               const int32_t* delays, int n_dm, const char* dm_path, float* observation_ms, uint64_t* gemms_written,
                               uint64_t* dm_times, int* pinned)
{
    if (!cfg || !name) return BF_ERR_INVALID;
    std::unique_ptr<dm_file_sink> dms;
    if (delays && dm_path) {
        int dmax = 0;
        for (size_t i = 0; i < (size_t)n_dm * cfg->n_freq; i++) dmax = delays[i] > dmax ? delays[i] : dmax;
        dms.reset(new dm_file_sink(*cfg, cfg->n_freq, n_dm, dmax, dm_path, gpu));
        if (!dms->is_open()) return BF_ERR_INVALID;
    }
    std::ostringstream quiet;
    std::ostream& log = verbose ? static_cast<std::ostream&>(std::cout) : quiet;
    shm_block_source src(name, core, /*pin=*/true, log);
    if (!src.ok()) return BF_ERR_STATE;
    src.expect_block_bytes(bf_bytes_per_block(cfg));
    if (pinned) *pinned = src.is_pinned() ? 1 : 0;
    std::unique_ptr<file_sink> sink;
    if (path) {
        sink.reset(new file_sink(*cfg, path, gpu));
        if (!sink->ok() || !sink->is_open()) return BF_ERR_INVALID;
    }
    std::vector<antenna> pos((size_t)cfg->n_ant);
    std::vector<beam_direction> dir((size_t)cfg->n_beams);
    default_positions(cfg->n_ant, pos.data());
    default_directions(cfg->n_beams, dir.data());
    observation_options opt;
    opt.gpu = gpu;
    opt.device = device;
    opt.verbose = verbose != 0;
    opt.sink = sink.get();
    opt.dm_delays = delays;
    opt.n_dm = delays ? n_dm : 0;
    opt.dm_sink = dms.get();
    observation_result res;
    std::streambuf* keep = std::cout.rdbuf();
    if (!verbose) std::cout.rdbuf(quiet.rdbuf());
    int rc = run_observation(*cfg, opt, src, pos.data(), dir.data(), &res, log);
    std::cout.rdbuf(keep);
    if (rc != BF_OK) return rc;
    if (dm_times) *dm_times = res.dm_times;
    if (observation_ms) *observation_ms = res.observation_time_ms;
    if (gemms_written) *gemms_written = sink ? sink->get_delivered() : res.blocks * cfg->n_gemms_per_block;
    return BF_OK;
}

int bfh_run_observation_junk_to_ring(const bf_config* cfg, uint64_t n_blocks, int ring_blocks, uint64_t seed, int gpu,
                                     int device, const char* out_ring, uint64_t out_ring_blocks, float* observation_ms,
                                     uint64_t* gemms_written, void* ring_copy)
{
    if (!cfg || !out_ring) return BF_ERR_INVALID;
    ring_sink sink(*cfg, out_ring, out_ring_blocks, gpu);
    if (!sink.ok() || !sink.is_open()) return BF_ERR_INVALID;
    observation_result res;
    int rc = run_junk(cfg, n_blocks, ring_blocks, seed, gpu, device, 0, 0, &sink, &res, ring_copy);
    if (rc != BF_OK) return rc;
    if (observation_ms) *observation_ms = res.observation_time_ms;
    if (gemms_written) *gemms_written = sink.get_delivered();
    return BF_OK;
}

struct bfh_sink {
    file_sink* s;
};

int bfh_file_sink_create(const bf_config* cfg, const char* path, int gpu, uint64_t slots, bfh_sink** out)
{
    if (!cfg || !path || !out) return BF_ERR_INVALID;
    file_sink* s = new (std::nothrow) file_sink(*cfg, path, gpu, slots);
    if (!s || !s->ok() || !s->is_open()) {
        delete s;
        return BF_ERR_INVALID;
    }
    *out = new bfh_sink{s};
    return BF_OK;
}
int bfh_sink_acquire(bfh_sink* s, uint64_t gemm_index, float** slot)
{
    if (!s || !slot) return BF_ERR_INVALID;
    *slot = s->s->acquire(gemm_index);
    return *slot ? BF_OK : BF_ERR_STATE;
}
int bfh_sink_commit(bfh_sink* s, uint64_t gemm_index)
{
    if (!s) return BF_ERR_INVALID;
    return s->s->commit(gemm_index) ? BF_OK : BF_ERR_STATE;
}
int bfh_sink_close(bfh_sink* s)
{
    if (!s) return BF_ERR_INVALID;
    s->s->close();
    return BF_OK;
}
int bfh_sink_destroy(bfh_sink* s)
{
    if (!s) return BF_OK;
    delete s->s;
    delete s;
    return BF_OK;
}

// the DM chunk sinks on their own (tests; no device needed)
struct bfh_dm_sink {
    dm_chunk_sink* s;
};
int bfh_dm_sink_create(const bf_config* cfg, const char* target, int n_freq_total, int n_dm, int max_delay, int max_rows, int first_trial,
                       bfh_dm_sink** out)
{
    if (!cfg || !target || !out || n_dm <= 0 || max_rows <= 0) return BF_ERR_INVALID;
    *out = nullptr;
    if (!std::strncmp(target, "ring:", 5)) {
        std::string nm(target + 5);
        uint64_t blocks = 4;
        const size_t colon = nm.find(':');
        if (colon != std::string::npos) {
            blocks = std::strtoull(nm.c_str() + colon + 1, nullptr, 10);
            nm.resize(colon);
        }
        dm_ring_sink* r = new dm_ring_sink(*cfg, n_freq_total, n_dm, max_delay, max_rows, nm.c_str(), blocks, 0, first_trial);
        if (!r->is_open()) {
            delete r;
            return BF_ERR_INVALID;
        }
        *out = new bfh_dm_sink{r};
        return BF_OK;
    }
    dm_file_sink* f = new dm_file_sink(*cfg, n_freq_total, n_dm, max_delay, target, 0, first_trial);
    if (!f->is_open()) {
        delete f;
        return BF_ERR_INVALID;
    }
    *out = new bfh_dm_sink{f};
    return BF_OK;
}
int bfh_dm_sink_deliver(bfh_dm_sink* s, uint64_t first_t, int n_t, int n_dm, int n_beams, const float* data)
{
    if (!s || !data) return BF_ERR_INVALID;
    return s->s->deliver(first_t, n_t, n_dm, n_beams, data) ? BF_OK : BF_ERR_STATE;
}
int bfh_dm_sink_destroy(bfh_dm_sink* s)
{
    if (!s) return BF_OK;
    s->s->close();
    delete s->s;
    delete s;
    return BF_OK;
}

}  // extern "C"
