// bf_host.cpp -- C++ host mirror of the reference's host-side classes (include/dsabf_host.hpp) and their C wrappers
// (include/dsabf_host.h).  Plain host C++: no device code here, compute goes through the C-ABI of dsabf.h.
//
// Floating-point fidelity: the reference's trig expressions are written with unqualified sin/cos/round on float
// operands, which under g++ resolve to the double C functions.  Every promotion is spelled out
// below so the bytes match the reference's CPU path exactly; this file is compiled with -ffp-contract=off.
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

namespace dsabf {

// ---- antenna / beam_direction (src/beamformer.hh:183-213) -------------------------------------------------
std::istream& operator>>(std::istream& in, antenna& a) { return in >> a.x >> a.y >> a.z; }
std::ostream& operator<<(std::ostream& out, const antenna& a)
{
    return out << "(" << a.x << ", " << a.y << ", " << a.z << ")" << std::endl;
}
std::istream& operator>>(std::istream& in, beam_direction& a) { return in >> a.theta >> a.phi; }
std::ostream& operator<<(std::ostream& out, const beam_direction& a)
{
    return out << "(" << a.theta << ", " << a.phi << ")" << std::endl;
}

int read_in_beam_directions(const char* file_name, int expected_beams, beam_direction* dir)
{
    std::ifstream input_file;
    input_file.open(file_name);
    if (!input_file.is_open()) return -1;
    int nbeam = 0;
    input_file >> nbeam;
    if (nbeam != expected_beams) {
        std::cout << "Number of beams in file (" << nbeam << ") does not match expected (" << expected_beams << ")"
                  << std::endl;
        std::cout << "Excess beams will be ignored, missing beams will be set to 0." << std::endl;
    }
    for (int beam_idx = 0; beam_idx < expected_beams; beam_idx++) {
        beam_direction d;  // a failed extraction leaves this and later entries at their zero defaults
        input_file >> d;
        dir[beam_idx] = d;
    }
    return 0;
}

int read_in_position_locations(const char* file_name, int n_antennas, antenna* pos)
{
    std::ifstream input_file;
    input_file.open(file_name);
    if (!input_file.is_open()) return -1;
    int nant = 0;
    input_file >> nant;
    if (nant != n_antennas) {
        std::cout << "Number of antennas in file (" << nant << ") does not match N_ANTENNAS (" << n_antennas << ")"
                  << std::endl;
        std::cout << "Excess antennas will be ignored, missing antennas will be set to 0." << std::endl;
    }
    for (int ant = 0; ant < n_antennas; ant++) {
        antenna a;
        input_file >> a;
        pos[ant] = a;
    }
    return 0;
}

int write_array_to_disk_as_python_file(const float* data_out, int rows, int cols, const char* output_filename)
{
    std::ofstream f;
    f.open(output_filename);
    if (!f.is_open()) return -1;
    f << "A = [[";
    for (int jj = 0; jj < rows; jj++) {
        for (int ii = 0; ii < cols; ii++) {
            f << data_out[(size_t)jj * cols + ii];
            if (ii != cols - 1) f << ",";
        }
        if (jj != rows - 1)
            f << "],\n[";
        else
            f << "]]" << std::endl;
    }
    f.close();
    return 0;
}

void print_all_defines(const bf_config& c, std::ostream& out)
{
    const long long n_ipo = bf_n_inputs_per_output(&c), n_time = bf_n_timesteps_per_gemm(&c);
    const long long n_cx_in = (long long)c.n_ant * c.n_freq * n_time, n_cx_out = (long long)c.n_beams * c.n_freq * n_time;
    out << "N_BEAMS: " << c.n_beams << "\n";
    out << "N_ANTENNAS: " << c.n_ant << "\n";
    out << "N_FREQUENCIES: " << c.n_freq << "\n";
    out << "N_AVERAGING: " << c.n_avg << "\n";
    out << "N_POL: " << c.n_pol << "\n";
    out << "N_CX: " << 2 << "\n";
    out << "N_GEMMS_PER_GPU: " << c.n_gemms_per_block * c.n_blocks_on_gpu << "\n";
    out << "N_OUTPUTS_PER_GEMM: " << c.n_out_per_gemm << "\n";
    out << "N_GEMMS_PER_BLOCK: " << c.n_gemms_per_block << "\n";
    out << "N_INPUTS_PER_OUTPUT: " << n_ipo << "\n";
    out << "N_TIMESTEPS_PER_GEMM: " << n_time << "\n";
    out << "N_BLOCKS_ON_GPU: " << c.n_blocks_on_gpu << "\n";
    out << "N_CX_IN_PER_GEMM: " << n_cx_in << "\n";
    out << "N_CX_OUT_PER_GEMM: " << n_cx_out << "\n";
    out << "N_BYTES_POST_EXPANSION_PER_GEMM: " << n_cx_in * 2 << "\n";
    out << "N_BYTES_PRE_EXPANSION_PER_GEMM: " << bf_bytes_per_gemm(&c) << "\n";
    out << "N_BYTES_PRE_EXPANSION_PER_BLOCK: " << bf_bytes_per_block(&c) << "\n";
    out << "N_GPUS: " << kNGpus << "\n";
    out << "TOT_CHANNELS: " << kTotChannels << "\n";
    out << "START_F: " << kStartF << "\n";
    out << "END_F: " << kEndF << "\n";
    out << "ZERO_PT: " << kZeroPt << "\n";
    out << "BW_PER_CHANNEL: " << ((kEndF - kStartF) / kTotChannels) << "\n";
    out << "C_SPEED: " << kCSpeed << "\n";
    out << "PI: " << kPi << "\n";
    out << "N_BITS: " << 8 << "\n";
    out << "MAX_VAL: " << kMaxVal << "\n";
    out << "SIG_BITS: " << 4 << "\n";
    out << "SIG_MAX_VAL: " << kSigMaxVal << "\n";
    out << "N_STREAMS: " << c.n_streams << "\n";
    out << "N_SOURCES_PER_BATCH: " << kSourcesPerBatch << "\n";
    out << std::endl;
}

void usage(bool debug_mode, std::ostream& out)
{
    if (debug_mode) {
        out << "dsaX_beamformer_DEBUG_MODE [options]\n"
               " -g gpu                  select a predefined frequency range\n"
               " -p position_filename    file where the antenna positions are stored\n"
               " -d direction_filename   file where the beam directions are stored\n"
               " -s source_filename      file where the source directions are stored\n"
               " -h                      print usage\n";
    } else {
        out << "dsaX_beamformer [options]\n"
               " -c core                 bind process to CPU core\n"
               " -k key                  [default dada]\n"
               " -g gpu                  select a predefined frequency range\n"
               " -p position_filename    file where the antenna positions are stored\n"
               " -d direction_filename   file where the beam directions are stored\n"
               " -h                      print usage\n";
    }
}

void default_positions(int n_antennas, antenna* pos)
{
    for (int i = 0; i < n_antennas; i++) {
        pos[i] = antenna();
        pos[i].x = i * 500.0 / (n_antennas - 1) - 250.0;  // src/beamformer.cu:138
    }
}

void default_directions(int n_beams, beam_direction* dir)
{
    const double deg2rad_2fov = (2 * kHalfFov) * kPi / 180.0;  // DEG2RAD(2*HALF_FOV)
    const double deg2rad_fov = (kHalfFov)*kPi / 180.0;         // DEG2RAD(HALF_FOV)
    for (int i = 0; i < n_beams; i++) {
        dir[i] = beam_direction();
        dir[i].theta = i * deg2rad_2fov / (n_beams - 1) - deg2rad_fov;  // src/beamformer.cu:145
    }
}

float channel_frequency_weights(int gpu, int chan)
{
    float bw_per_channel = (kEndF - kStartF) / kTotChannels;  // src/beamformer.cu:173
    float freq = kEndF - (kZeroPt + gpu * kTotChannels / (kNGpus - 1) + chan) * bw_per_channel;  // :233
    return freq;
}

float channel_frequency_generator(int gpu, int chan)
{
    float freq = kEndF - (kZeroPt + gpu * kTotChannels / (kNGpus - 1) + chan) * ((kEndF - kStartF) / kTotChannels);
    return freq;  // src/test_data_generator.hh:72
}

static void parallel_for(long n, const std::function<void(long, long)>& body)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if ((long)nt > n) nt = (unsigned)std::max<long>(1, n);
    if (nt <= 1) {
        body(0, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back(body, n * t / nt, n * (t + 1) / nt);
    for (auto& x : th) x.join();
}

void generate_fourier_coefficients(int n_beams, int n_antennas, int n_freq, int chan0, int gpu, const antenna* pos,
                                   const beam_direction* dir, int8_t* out)
{
    parallel_for(n_freq, [&](long lo, long hi) {
        for (long i = lo; i < hi; i++) {
            float freq = channel_frequency_weights(gpu, chan0 + (int)i);
            float wavelength = kCSpeed / (1E9 * freq);  // src/beamformer.cu:234
            for (int j = 0; j < n_antennas; j++) {
                for (int k = 0; k < n_beams; k++) {
                    const double proj = (double)pos[j].x * ::sin((double)dir[k].theta) +
                                        (double)pos[j].y * ::sin((double)dir[k].phi);
                    int8_t* o = out + 2 * ((size_t)i * n_antennas * n_beams + (size_t)j * n_beams + k);
                    o[0] = (int8_t)::round(kMaxVal * ::cos(-2 * kPi * proj / (double)wavelength));  // :237
                    o[1] = (int8_t)::round(kMaxVal * ::sin(-2 * kPi * proj / (double)wavelength));  // :238
                }
            }
        }
    });
}

// ---- test_data_generator -------------------------------------------------------------------------------------
test_data_generator::test_data_generator(const bf_config& c, int per_batch, bool pin)
    : cfg(c), n_sources_per_batch(per_batch)
{
    const size_t n = input_data_size();
    void* p = nullptr;
    if (pin && bf_alloc_pinned(&p, n) == BF_OK) {  // cudaHostAlloc, src/test_data_generator.hh:35
        pinned = true;
    } else {
        p = ::malloc(n);
        pinned = false;
    }
    data = static_cast<char*>(p);
    if (data) ::memset(data, kBogusData, n);  // :36
}

test_data_generator::~test_data_generator()
{
    if (!data) return;
    if (pinned)
        bf_free_pinned(data);
    else
        ::free(data);
}

size_t test_data_generator::input_data_size() const { return bf_bytes_per_gemm(&cfg) * (size_t)n_sources_per_batch; }

void test_data_generator::set_source_directions(const beam_direction* src, int n)
{
    if (!use_source_catalog) {
        n_pt_sources = n;
        sources.assign(src, src + n);
        use_source_catalog = true;
        n_source_batches = (n_pt_sources + n_sources_per_batch - 1) / n_sources_per_batch;  // CEILING, :54
    }
}

int test_data_generator::read_in_source_directions(const char* file_name)
{
    if (use_source_catalog) return 0;
    std::ifstream input_file;
    input_file.open(file_name);
    if (!input_file.is_open()) return -1;
    int n = 0;
    input_file >> n;
    if (n < 0) n = 0;
    std::vector<beam_direction> s((size_t)n);
    for (int beam_idx = 0; beam_idx < n; beam_idx++) input_file >> s[beam_idx];
    set_source_directions(s.data(), n);
    if (cfg.verbose) std::cout << "Read in " << n_pt_sources << " source directions" << std::endl;
    return 0;
}

void test_data_generator::generate_test_data(const antenna pos[], int gpu)
{
    const int na = cfg.n_ant, nf = cfg.n_freq, nt = bf_n_timesteps_per_gemm(&cfg);
    const size_t per_gemm = bf_bytes_per_gemm(&cfg);
    parallel_for(n_sources_per_batch, [&](long lo, long hi) {
        for (long direction = lo; direction < hi; direction++) {
            const int source_look_up = (int)direction + source_batch_counter * n_sources_per_batch;  // :77
            for (int i = 0; i < nf; i++) {
                float freq = channel_frequency_generator(gpu, i);  // :72
                float wavelength = kCSpeed / (1E9 * freq);         // :74
                char* slab = data + (size_t)direction * per_gemm + (size_t)i * nt * na;
                for (int k = 0; k < na; k++) {
                    char byte = 0;  // :85
                    if (source_look_up < n_pt_sources) {
                        const double proj = (double)pos[k].x * ::sin((double)sources[source_look_up].theta) +
                                            (double)pos[k].y * ::sin((double)sources[source_look_up].phi);
                        const char high = (char)::round(kSigMaxVal * ::cos(2 * kPi * proj / (double)wavelength));  // :80
                        const char low = (char)::round(kSigMaxVal * ::sin(2 * kPi * proj / (double)wavelength));   // :81
                        byte = (char)(((int)high * 16) | (0x0F & (int)low));                                      // :83
                    }
                    slab[k] = byte;
                }
                // the reference evaluates the same expression for every time column j (it has no j in it):
                // replicate column 0 -- identical bytes, 1/n_time of the trig calls
                for (int j = 1; j < nt; j++) ::memcpy(slab + (size_t)j * na, slab, (size_t)na);
            }
        }
    });
    source_batch_counter++;  // :94
}

bool test_data_generator::check_need_to_generate_more_input_data(int blocks_transfered)
{
    return (use_source_catalog &&
            (blocks_transfered == (source_batch_counter * n_sources_per_batch) / cfg.n_gemms_per_block));  // :100
}

bool test_data_generator::check_data_ready_for_transfer(int blocks_transfer_queue)
{
    if (!use_source_catalog && (source_batch_counter == 0)) source_batch_counter = 1;             // :104-106
    return (blocks_transfer_queue < (source_batch_counter * n_sources_per_batch) / cfg.n_gemms_per_block);  // :107
}

// ---- event backends ---------------------------------------------------------------------------------------------
namespace {

struct hip_backend : event_backend {
    bf_handle* h;
    explicit hip_backend(bf_handle* hh) : h(hh) {}
    void* create() override
    {
        bf_event* e = nullptr;
        if (bf_event_create(&e) != BF_OK) return nullptr;
        return e;
    }
    void destroy(void* ev) override { bf_event_destroy(static_cast<bf_event*>(ev)); }
    void record_transfer(void* ev) override { bf_record_transfer_event(h, static_cast<bf_event*>(ev)); }
    void record_analysis(void* ev) override { bf_record_analysis_event(h, static_cast<bf_event*>(ev)); }
    int query(void* ev) override { return bf_event_query(static_cast<bf_event*>(ev)); }
};

// Test double: events complete only when the test says so (bfh_obs_fake_complete).
struct fake_event {
    int state = 0;  // 0 = never recorded (queries "done", like a fresh CUDA event), 1 = pending, 2 = done
};
struct fake_backend : event_backend {
    std::vector<fake_event*> pending_transfers, pending_analyses;
    void* create() override { return new fake_event(); }
    void destroy(void* ev) override
    {
        fake_event* e = static_cast<fake_event*>(ev);
        pending_transfers.erase(std::remove(pending_transfers.begin(), pending_transfers.end(), e), pending_transfers.end());
        pending_analyses.erase(std::remove(pending_analyses.begin(), pending_analyses.end(), e), pending_analyses.end());
        delete e;
    }
    void record_transfer(void* ev) override
    {
        fake_event* e = static_cast<fake_event*>(ev);
        e->state = 1;
        pending_transfers.push_back(e);
    }
    void record_analysis(void* ev) override
    {
        fake_event* e = static_cast<fake_event*>(ev);
        e->state = 1;
        pending_analyses.push_back(e);
    }
    int query(void* ev) override { return static_cast<fake_event*>(ev)->state == 1 ? BF_NOT_READY : BF_OK; }
    void complete(int nt, int na)
    {
        for (; nt > 0 && !pending_transfers.empty(); nt--) {
            pending_transfers.front()->state = 2;
            pending_transfers.erase(pending_transfers.begin());
        }
        for (; na > 0 && !pending_analyses.empty(); na--) {
            pending_analyses.front()->state = 2;
            pending_analyses.erase(pending_analyses.begin());
        }
    }
};

}  // namespace

event_backend* make_hip_event_backend(bf_handle* h) { return new hip_backend(h); }

// ---- observation_loop_state (src/observation_loop.hh:54-176) -------------------------------------------------------
observation_loop_state::observation_loop_state(uint64_t max_tsep, uint64_t max_totsep, const bf_config& cfg,
                                               event_backend* backend, bool debug)
    : maximum_transfer_seperation(max_tsep), maximum_total_seperation(max_totsep), debug_mode(debug),
      verbose(cfg.verbose != 0), n_gemms_per_block(cfg.n_gemms_per_block), n_blocks_on_gpu(cfg.n_blocks_on_gpu),
      n_events(5 * cfg.n_blocks_on_gpu), ev(backend)
{
    BlockTransferredSync.resize(n_events);
    BlockAnalyzedSync.resize(n_events);
    for (int i = 0; i < n_events; i++) {  // :58-61
        BlockTransferredSync[i] = ev->create();
        BlockAnalyzedSync[i] = ev->create();
    }
}

observation_loop_state::~observation_loop_state()
{
    for (int event = 0; event < n_events; event++) {  // :65-68
        ev->destroy(BlockAnalyzedSync[event]);
        ev->destroy(BlockTransferredSync[event]);
    }
}

void observation_loop_state::generate_transfer_event()
{
    ev->record_transfer(BlockTransferredSync[blocks_transfer_queue % n_events]);  // :73
    blocks_transfer_queue++;
}

void observation_loop_state::generate_analysis_event()
{
    ev->record_analysis(BlockAnalyzedSync[blocks_analysis_queue % n_events]);  // :79
    blocks_analysis_queue++;
}

void observation_loop_state::check_transfer_events()
{
    for (uint64_t event = blocks_transferred; event < blocks_transfer_queue; event++) {  // :85
        if (ev->query(BlockTransferredSync[event % n_events]) == BF_OK) {
            if (verbose) std::cout << "Block " << event << " transfered to GPU" << std::endl;
            blocks_transferred++;
            ev->destroy(BlockTransferredSync[event % n_events]);  // :95-96 destroy and recreate
            BlockTransferredSync[event % n_events] = ev->create();
        } else {
            break;  // :98
        }
    }
}

void observation_loop_state::check_analysis_events()
{
    for (uint64_t event = blocks_analyzed; event < blocks_analysis_queue; event++) {  // :104
        if (ev->query(BlockAnalyzedSync[event % n_events]) == BF_OK) {
            blocks_analyzed++;
            if (verbose) std::cout << "Block " << event << " Analyzed" << std::endl;
            ev->destroy(BlockAnalyzedSync[event % n_events]);  // :112-113
            BlockAnalyzedSync[event % n_events] = ev->create();
        } else {
            break;  // :116
        }
    }
}

uint64_t observation_loop_state::get_current_analysis_gemm(int time_slice)
{
    most_recent_gemm = (int)(blocks_analysis_queue * n_gemms_per_block + time_slice);  // :122
    return most_recent_gemm;
}

uint64_t observation_loop_state::get_current_transfer_gemm() const { return blocks_transfer_queue * n_gemms_per_block; }

bool observation_loop_state::check_ready_for_transfer() const
{
    return ((blocks_transfer_queue - blocks_analyzed < maximum_total_seperation) &&
            (blocks_transfer_queue - blocks_transferred < maximum_transfer_seperation) && !transfers_complete);  // :131-133
}

bool observation_loop_state::check_ready_for_dh2_transfer(int time_slice)
{
    int current_gemm = (int)get_current_analysis_gemm(time_slice);  // :137
    return (current_gemm < n_pt_sources);
}

bool observation_loop_state::check_ready_for_analysis() const { return (blocks_analysis_queue < blocks_transferred); }

bool observation_loop_state::check_observations_complete()
{
    if (debug_mode) {  // :146-151
        if ((most_recent_gemm >= n_pt_sources - 1) && (blocks_analyzed == blocks_transfer_queue) && transfers_complete) {
            std::cout << "obs Complete" << std::endl;
            return true;
        }
        return false;
    }
    if ((blocks_analyzed == blocks_transfer_queue) && transfers_complete) {  // :153-157
        std::cout << "obs Complete" << std::endl;
        return true;
    }
    return false;
}

bool observation_loop_state::check_transfers_complete()
{
    if (blocks_transfer_queue * n_gemms_per_block >= (uint64_t)std::max(n_pt_sources, 0)) {  // :163
        transfers_complete = 1;
        return true;
    }
    return false;
}

std::ostream& operator<<(std::ostream& out, const observation_loop_state& a)
{
    return out << "A: " << a.blocks_analyzed << ", AQ: " << a.blocks_analysis_queue << ", T: " << a.blocks_transferred
               << ", TQ: " << a.blocks_transfer_queue << "\n"
               << "current_gemm: " << a.most_recent_gemm << ", transfers_complete: " << a.transfers_complete;
}

// ---- the DEBUG main() flow (src/beamformer.cu:12-621 with -DDEBUG) ----------------------------------------------------
int run_debug_observation(const bf_config& cfg, const debug_run_options& opt, debug_run_result* res,
                          std::vector<float>* dedispersed_result, std::ostream& log)
{
    const int n_streams = cfg.n_streams;
    if (cfg.n_gemms_per_block % n_streams) {
        log << "N_GEMMS_PER_BLOCK must be divisible by N_STREAMS" << std::endl;
        return BF_ERR_INVALID;
    }
    if (kSourcesPerBatch % cfg.n_gemms_per_block) {  // static_assert src/beamformer.hh:151
        log << "N_SOURCES_PER_BATCH must be divisible by N_GEMMS_PER_BLOCK" << std::endl;
        return BF_ERR_INVALID;
    }
    std::vector<antenna> pos((size_t)cfg.n_ant);
    std::vector<beam_direction> dir((size_t)cfg.n_beams);
    bool pos_set = false, dir_set = false;

    test_data_generator input_data_generator(cfg);
    if (!input_data_generator.get_data()) return BF_ERR_DEVICE;
    if (opt.sources && input_data_generator.read_in_source_directions(opt.sources) != 0) {
        log << "beam: could not read source direction file " << opt.sources << std::endl;
        return BF_ERR_INVALID;
    }
    if (opt.positions) {
        if (read_in_position_locations(opt.positions, cfg.n_ant, pos.data()) != 0) return BF_ERR_INVALID;
        pos_set = true;
    }
    if (opt.directions) {
        if (read_in_beam_directions(opt.directions, cfg.n_beams, dir.data()) != 0) return BF_ERR_INVALID;
        dir_set = true;
    }
    if (!pos_set) default_positions(cfg.n_ant, pos.data());    // :135-140
    if (!dir_set) default_directions(cfg.n_beams, dir.data());  // :142-147
    if (opt.verbose) print_all_defines(cfg, log);

    bf_handle* h = nullptr;
    int rc = bf_create(&cfg, opt.device, &h);
    if (rc != BF_OK) {
        log << "GPUassert: " << bf_last_error() << std::endl;
        return rc;
    }
    struct guard {
        bf_handle* h;
        std::vector<void*> pinned;
        ~guard()
        {
            if (h) bf_stream_sync(h, -1);
            for (void* p : pinned) bf_free_pinned(p);
            bf_destroy(h);
        }
    } g{h, {}};

    const int n_src = input_data_generator.get_n_pt_sources();
    const size_t n_f_per_detect = bf_floats_per_detect(&cfg);
    float *beam_out = nullptr, *dedispersed_out = nullptr;
    void* p = nullptr;
    if ((rc = bf_alloc_pinned(&p, n_f_per_detect * n_streams * sizeof(float))) != BF_OK) return rc;  // :249
    g.pinned.push_back(p);
    beam_out = static_cast<float*>(p);
    if ((rc = bf_alloc_pinned(&p, (size_t)cfg.n_beams * std::max(n_src, 1) * sizeof(float))) != BF_OK) return rc;  // :212
    g.pinned.push_back(p);
    dedispersed_out = static_cast<float*>(p);
    ::memset(dedispersed_out, 0, (size_t)cfg.n_beams * std::max(n_src, 1) * sizeof(float));

    {  // :230-241, :251, :272
        std::vector<int8_t> fourier_coefficients((size_t)cfg.n_freq * cfg.n_ant * cfg.n_beams * 2);
        generate_fourier_coefficients(cfg.n_beams, cfg.n_ant, cfg.n_freq, 0, opt.gpu, pos.data(), dir.data(),
                                      fourier_coefficients.data());
        if ((rc = bf_set_weights(h, fourier_coefficients.data())) != BF_OK) {
            log << "GPUassert: " << bf_last_error() << std::endl;
            return rc;
        }
    }

    std::vector<int> timeSlice((size_t)n_streams);
    for (int i = 0; i < n_streams; i++) timeSlice[i] = i;  // :319

    hip_backend backend(h);
    observation_loop_state obs_state(kMaxTransferSep, kMaxTotalSep, cfg, &backend, /*debug_mode=*/true);  // :322
    obs_state.set_n_pt_sources(n_src);                                                                  // :325

    if (opt.verbose) {
        log << "Executing beamformer.cu" << "\n";
        log << "MAX_TOTAL_SEP: " << kMaxTotalSep << "\n";
        log << "MAX_TRANSFER_SEP: " << kMaxTransferSep << std::endl;
    }

    float time_accumulator_ms = 0, observation_time_ms = 0;
    bf_timer_start(h);  // :358
    const size_t block_bytes = bf_bytes_per_block(&cfg);
    const size_t input_data_size = input_data_generator.input_data_size();

    while (!obs_state.check_observations_complete()) {  // :364
        if (opt.verbose) {
            log << "##########################################" << std::endl;
            log << obs_state << std::endl;
        }
        if (obs_state.check_ready_for_transfer()) {  // :378
            if (input_data_generator.check_need_to_generate_more_input_data((int)obs_state.get_blocks_transferred())) {
                log << "Generating new source data" << std::endl;
                bf_timer_stop(h, &time_accumulator_ms);  // :408-409
                observation_time_ms += time_accumulator_ms;
                input_data_generator.generate_test_data(pos.data(), opt.gpu);
                bf_timer_start(h);
                log << "done generating test data" << std::endl;
            }
            if (input_data_generator.check_data_ready_for_transfer((int)obs_state.get_blocks_transfer_queue())) {  // :421
                char* input_data = input_data_generator.get_data();
                rc = bf_submit_block(h, (int)obs_state.get_next_gpu_transfer_block(),
                                     &input_data[(block_bytes * obs_state.get_blocks_transfer_queue()) % input_data_size],
                                     block_bytes, nullptr);  // :425-429
                if (rc != BF_OK) {
                    log << "GPUassert: " << bf_last_error() << std::endl;
                    return rc;
                }
                obs_state.generate_transfer_event();  // :431
            }
            obs_state.check_transfers_complete();  // :438
        }
        obs_state.check_transfer_events();  // :446

        if (obs_state.check_ready_for_analysis()) {  // :452
            for (int part = 0; part < cfg.n_gemms_per_block / n_streams; part++) {
                if (opt.verbose)
                    log << "Queueing Beamforming. Start Dir = " << obs_state.get_current_analysis_gemm(timeSlice[0])
                        << std::endl;
                for (int st = 0; st < n_streams; st++) {
                    rc = bf_enqueue_gemm_unit(h, st, (int)obs_state.get_next_gpu_analysis_block(), timeSlice[st],
                                              &beam_out[(size_t)st * n_f_per_detect]);  // :464-488
                    if (rc != BF_OK) {
                        log << "GPUassert: " << bf_last_error() << std::endl;
                        return rc;
                    }
                    if (obs_state.check_ready_for_dh2_transfer(timeSlice[st])) {  // :492
                        int current_gemm = (int)obs_state.get_current_analysis_gemm(timeSlice[st]);
                        if (opt.verbose) log << "Current GEMM: " << current_gemm << std::endl;
                        rc = bf_enqueue_dedisperse(h, st, &dedispersed_out[(size_t)current_gemm * cfg.n_beams]);  // :498-510
                        if (rc != BF_OK) {
                            log << "GPUassert: " << bf_last_error() << std::endl;
                            return rc;
                        }
                    }
                    timeSlice[st] += n_streams;  // :515
                    if (timeSlice[st] >= cfg.n_gemms_per_block) timeSlice[st] -= cfg.n_gemms_per_block;
                }
            }
            obs_state.generate_analysis_event();  // :525
        }
        obs_state.check_analysis_events();  // :532
    }

    bf_timer_stop(h, &time_accumulator_ms);  // :540
    observation_time_ms += time_accumulator_ms;
    const long long chunks = (long long)n_src * cfg.n_out_per_gemm;
    log << "Observation ran in " << observation_time_ms << "milliseconds.\n";
    log << "Code produced outputs for " << chunks << " data chunks.\n";
    log << "Time per data chunk: " << observation_time_ms / chunks << " milliseconds.\n";
    log << "Approximate datarate: " << bf_bytes_per_gemm(&cfg) * (double)n_src / observation_time_ms / 1e6 << "GB/s"
        << std::endl;

    bf_stream_sync(h, -1);  // :560-562
    log << "Synchronized" << std::endl;

    if (opt.output && opt.output[0])
        if (write_array_to_disk_as_python_file(dedispersed_out, n_src, cfg.n_beams, opt.output) != 0)  // :568-571
            log << "could not write " << opt.output << std::endl;
    if (dedispersed_result) dedispersed_result->assign(dedispersed_out, dedispersed_out + (size_t)n_src * cfg.n_beams);
    if (res) {
        res->observation_time_ms = observation_time_ms;
        res->n_pt_sources = n_src;
        res->data_chunks = chunks;
    }
    return BF_OK;
}

// ---- junk_block_source -------------------------------------------------------------------------------------------
junk_block_source::junk_block_source(const bf_config& c, uint64_t nb, int rb, uint64_t seed)
    : block_size(bf_bytes_per_block(&c)), n_blocks(nb), ring_blocks(rb < 1 ? 1 : rb)
{
    const size_t total = (size_t)block_size * ring_blocks;
    void* p = nullptr;
    if (bf_alloc_pinned(&p, total) == BF_OK) {  // dada_cuda_dbregister pins the shm blocks, src/dada_handler.hh:127-177
        pinned = true;
    } else {
        p = ::malloc(total);
    }
    ring = static_cast<char*>(p);
    if (!ring) return;
    junk_fill(c, ring_blocks, seed, ring);
}

void junk_fill(const bf_config& c, int ring_blocks, uint64_t seed, char* ring)
{
    // every byte value (all 16 nibble codes in both halves), distinct blocks: 64-bit xorshift* per 8 bytes
    const size_t total = (size_t)bf_bytes_per_block(&c) * ring_blocks;
    parallel_for((long)ring_blocks * 64, [&](long lo, long hi) {
        for (long part = lo; part < hi; part++) {
            const size_t n8 = total / 8 / ((size_t)ring_blocks * 64);
            uint64_t x = seed * 0x9E3779B97F4A7C15ULL + (uint64_t)(part + 1) * 0xBF58476D1CE4E5B9ULL;
            uint64_t* q = reinterpret_cast<uint64_t*>(ring) + (size_t)part * n8;
            for (size_t i = 0; i < n8; i++) {
                x ^= x >> 12;
                x ^= x << 25;
                x ^= x >> 27;
                q[i] = x * 0x2545F4914F6CDD1DULL;
            }
        }
    });
}

junk_block_source::~junk_block_source()
{
    if (!ring) return;
    if (pinned)
        bf_free_pinned(ring);
    else
        ::free(ring);
}

char* junk_block_source::read()
{
    if (served < n_blocks) {
        bytes_read = block_size;
        return ring + (size_t)(served++ % (uint64_t)ring_blocks) * block_size;
    }
    bytes_read = 0;  // short block: end of data
    return ring;
}

bool junk_block_source::check_transfers_complete() { return bytes_read < block_size; }  // src/dada_handler.hh:105-113

// ---- dedispersion trial ladder and delays (sandbox/Dispersion Theory.ipynb) ------------------------------------------
std::vector<double> dm_trials(double dm0, double dm_max, int nchan, double epsilon, double nu_ghz, double chan_bw_mhz,
                              double ti_us, double tscat_us, double tsamp_us)
{
    const double n2 = (double)nchan * (double)nchan;                               // cell 1
    const double alpha = 1.0 / (16 + n2);
    const double beta = ti_us * ti_us + tscat_us * tscat_us + tsamp_us * tsamp_us;
    const double k = (nu_ghz * nu_ghz * nu_ghz) / (8.3 * chan_bw_mhz);
    std::vector<double> dms{dm0};
    double dm_prev = dm0;
    while (dm_prev < dm_max) {                                                      // cell 2
        dm_prev = n2 * alpha * dm_prev + 4 * std::sqrt(alpha * (epsilon * epsilon - n2 * alpha) * dm_prev * dm_prev +
                                                       alpha * beta * (epsilon * epsilon - 1) * (k * k));
        dms.push_back(dm_prev);
    }
    return dms;
}

void dm_delays(const double* dms, int n_dm, const float* freq_ghz, int n_freq, double f_ref_ghz, double tsamp_ms,
               int32_t* out)
{
    for (int d = 0; d < n_dm; d++)
        for (int f = 0; f < n_freq; f++) {
            const double fr = (double)freq_ghz[f];                                  // cell 5
            out[(size_t)d * n_freq + f] =
                (int32_t)(4.15 * dms[d] * (-1.0 / (f_ref_ghz * f_ref_ghz) + 1.0 / (fr * fr)) / tsamp_ms);
        }
}

// ---- detected-stream sinks ------------------------------------------------------------------------------------------
detected_sink::detected_sink(const bf_config& cfg, uint64_t slots)
    : floats_per_gemm(bf_floats_per_detect(&cfg)), n_slots(slots ? slots : slots_for(cfg))
{
    const size_t bytes = floats_per_gemm * n_slots * sizeof(float);
    void* p = nullptr;
    if (bf_alloc_pinned(&p, bytes) == BF_OK)
        pinned = true;
    else
        p = ::malloc(bytes);  // no device (CPU tests of the ring logic)
    ring = static_cast<float*>(p);
}

detected_sink::~detected_sink()
{
    if (!ring) return;
    if (pinned)
        bf_free_pinned(ring);
    else
        ::free(ring);
}

float* detected_sink::acquire(uint64_t gemm_index)
{
    if (!ring || gemm_index < next_commit || gemm_index >= next_commit + n_slots) return nullptr;
    return ring + (size_t)(gemm_index % n_slots) * floats_per_gemm;
}

bool detected_sink::commit(uint64_t gemm_index)
{
    if (!ring || gemm_index != next_commit) return false;
    if (!deliver(gemm_index, ring + (size_t)(gemm_index % n_slots) * floats_per_gemm, floats_per_gemm)) failed = true;
    next_commit++;
    delivered++;
    return !failed;
}

file_sink::file_sink(const bf_config& cfg, const char* path, int gpu, uint64_t slots) : detected_sink(cfg, slots)
{
    fp = ::fopen(path, "wb");
    if (!fp) return;
    char header[kHeaderBytes];
    ::memset(header, 0, sizeof(header));
    ::snprintf(header, sizeof(header),
               "HDR_VERSION 1.0\nHDR_SIZE %zu\nINSTRUMENT DSA\nCONTENT detected_power\nDTYPE float32\nENDIAN little\n"
               "ORDER gemm,output,frequency,beam\nN_BEAMS %d\nN_FREQUENCIES %d\nN_OUTPUTS_PER_GEMM %d\nN_ANTENNAS %d\n"
               "N_POL %d\nN_AVERAGING %d\nN_GEMMS_PER_BLOCK %d\nGPU %d\nFLOATS_PER_GEMM %zu\n",
               kHeaderBytes, cfg.n_beams, cfg.n_freq, cfg.n_out_per_gemm, cfg.n_ant, cfg.n_pol, cfg.n_avg,
               cfg.n_gemms_per_block, gpu, get_floats_per_gemm());
    if (::fwrite(header, 1, sizeof(header), fp) != sizeof(header)) {
        ::fclose(fp);
        fp = nullptr;
    }
}

file_sink::~file_sink() { finish(); }

bool file_sink::deliver(uint64_t, const float* data, size_t n_floats)
{
    return fp && ::fwrite(data, sizeof(float), n_floats, fp) == n_floats;
}

void file_sink::finish()
{
    if (fp) ::fclose(fp);
    fp = nullptr;
}

ring_sink::ring_sink(const bf_config& cfg, const char* ring_name, uint64_t ring_blocks, int gpu, uint64_t slots)
    : detected_sink(cfg, slots), name(ring_name ? ring_name : "")
{
    char header[kRingHeaderBytes];
    ::snprintf(header, sizeof(header),
               "HDR_VERSION 1.0\nHDR_SIZE %zu\nINSTRUMENT DSA\nCONTENT detected_power\nDTYPE float32\nENDIAN little\n"
               "ORDER output,frequency,beam\nN_BEAMS %d\nN_FREQUENCIES %d\nN_OUTPUTS_PER_GEMM %d\nGPU %d\n",
               kRingHeaderBytes, cfg.n_beams, cfg.n_freq, cfg.n_out_per_gemm, gpu);
    out = shm_ring::create(name.c_str(), ring_blocks, get_floats_per_gemm() * sizeof(float), header);
}

ring_sink::~ring_sink()
{
    finish();
}

bool ring_sink::deliver(uint64_t, const float* data, size_t n_floats)
{
    if (!out) return false;
    char* b = out->open_block_write();  // blocks while the consumer is behind by a whole ring
    if (!b) return false;
    ::memcpy(b, data, n_floats * sizeof(float));
    out->close_block_write(n_floats * sizeof(float));
    return true;
}

void ring_sink::finish()
{
    if (!out) return;
    if (out->open_block_write()) out->close_block_write(0);  // short block: end of data
    // wait for the consumer to drain, then remove the ring (dada_db -d)
    for (int waited = 0; out->get_blocks_read() < out->get_blocks_written() && waited < 10000; waited += 5) ::usleep(5000);
    delete out;
    out = nullptr;
    shm_ring::unlink(name.c_str());
}

// ---- production observation loop (src/beamformer.cu:364-534, #ifndef DEBUG branches) -------------------------------------
int run_observation(const bf_config& cfg, const observation_options& opt, block_source& source, const antenna* pos,
                    const beam_direction* dir, observation_result* res, std::ostream& log)
{
    const int n_streams = cfg.n_streams;
    if (cfg.n_gemms_per_block % n_streams) return BF_ERR_INVALID;
    bf_handle* h = nullptr;
    int rc = bf_create(&cfg, opt.device, &h);
    if (rc != BF_OK) {
        log << "GPUassert: " << bf_last_error() << std::endl;
        return rc;
    }
    struct guard {
        bf_handle* h;
        void* pinned;
        ~guard()
        {
            if (h) bf_stream_sync(h, -1);
            bf_free_pinned(pinned);
            bf_destroy(h);
        }
    } g{h, nullptr};

    const size_t n_f_per_detect = bf_floats_per_detect(&cfg);
    if ((rc = bf_alloc_pinned(&g.pinned, n_f_per_detect * n_streams * sizeof(float))) != BF_OK) return rc;  // :249
    float* beam_out = static_cast<float*>(g.pinned);
    ::memset(beam_out, 0, n_f_per_detect * n_streams * sizeof(float));
    {
        std::vector<int8_t> fourier_coefficients((size_t)cfg.n_freq * cfg.n_ant * cfg.n_beams * 2);
        generate_fourier_coefficients(cfg.n_beams, cfg.n_ant, cfg.n_freq, 0, opt.gpu, pos, dir, fourier_coefficients.data());
        if ((rc = bf_set_weights(h, fourier_coefficients.data())) != BF_OK) return rc;
    }
    std::vector<int> timeSlice((size_t)n_streams);
    for (int i = 0; i < n_streams; i++) timeSlice[i] = i;  // :319
    std::vector<long long> last_gemm((size_t)n_streams, -1);

    hip_backend backend(h);
    observation_loop_state obs_state(kMaxTransferSep, kMaxTotalSep, cfg, &backend, /*debug_mode=*/false);  // :322
    source.read_headers();  // :334
    if (opt.burn_in > 0) {  // :348-355
        log << "Burning IN" << std::endl;
        for (int i = 0; i < opt.burn_in; i++) {
            source.read();
            source.close();
        }
        log << "Done Burn in" << std::endl;
    }
    const size_t block_bytes = bf_bytes_per_block(&cfg);
    if (source.get_block_size() != block_bytes)
        log << "ERROR: block size " << source.get_block_size() << ", Should also be " << block_bytes << std::endl;

    uint64_t sink_committed = 0;
    bf_timer_start(h);  // :358
    while (!obs_state.check_observations_complete()) {  // :364
        if (opt.verbose) {
            log << "##########################################" << std::endl;
            log << obs_state << std::endl;
        }
        if (obs_state.check_ready_for_transfer()) {  // :378
            char* block = source.read();             // :384
            if (!source.check_transfers_complete()) {  // :386
                rc = bf_submit_block(h, (int)obs_state.get_next_gpu_transfer_block(), block, block_bytes, nullptr);  // :389-393
                if (rc != BF_OK) {
                    log << "GPUassert: " << bf_last_error() << std::endl;
                    return rc;
                }
                obs_state.generate_transfer_event();  // :396
                if (source.close_releases_block())      // not in the reference: see block_source::close_releases_block
                    while (obs_state.get_blocks_transferred() < obs_state.get_blocks_transfer_queue())
                        obs_state.check_transfer_events();
            } else {
                obs_state.set_transfers_complete(true);  // :398
            }
            source.close();  // :401
        }
        obs_state.check_transfer_events();  // :446
        if (obs_state.check_ready_for_analysis()) {  // :452
            const long long block_index = (long long)obs_state.get_blocks_analysis_queue();
            for (int part = 0; part < cfg.n_gemms_per_block / n_streams; part++) {
                for (int st = 0; st < n_streams; st++) {
                    float* dst = &beam_out[(size_t)st * n_f_per_detect];  // the reference's destination, :485-488
                    if (opt.sink) {
                        dst = opt.sink->acquire((uint64_t)block_index * cfg.n_gemms_per_block + timeSlice[st]);
                        if (!dst) {
                            log << "ERROR: detected sink has no free slot" << std::endl;
                            return BF_ERR_STATE;
                        }
                    }
                    rc = bf_enqueue_gemm_unit(h, st, (int)obs_state.get_next_gpu_analysis_block(), timeSlice[st], dst);  // :464-488
                    if (rc != BF_OK) {
                        log << "GPUassert: " << bf_last_error() << std::endl;
                        return rc;
                    }
                    last_gemm[st] = block_index * cfg.n_gemms_per_block + timeSlice[st];
                    timeSlice[st] += n_streams;  // :515-519
                    if (timeSlice[st] >= cfg.n_gemms_per_block) timeSlice[st] -= cfg.n_gemms_per_block;
                }
            }
            obs_state.generate_analysis_event();  // :525
        }
        obs_state.check_analysis_events();  // :532
        if (opt.sink) {  // every D2H copy of an analysed block has landed: hand its gemm-units over, in order
            for (; sink_committed < obs_state.get_blocks_analyzed() * (uint64_t)cfg.n_gemms_per_block; sink_committed++)
                if (!opt.sink->commit(sink_committed)) {
                    log << "ERROR: detected sink failed at gemm-unit " << sink_committed << std::endl;
                    return BF_ERR_STATE;
                }
        }
    }
    float ms = 0;
    bf_timer_stop(h, &ms);
    bf_stream_sync(h, -1);  // :560-562
    if (opt.sink) opt.sink->close();
    const uint64_t blocks = obs_state.get_blocks_analyzed();
    const uint64_t chunks = obs_state.get_current_transfer_gemm() * cfg.n_out_per_gemm;  // :552
    const double rate = (double)source.get_block_size() * obs_state.get_blocks_transfer_queue() / ms / 1e6;  // :554
    log << "Observation ran in " << ms << "milliseconds.\n";
    log << "Code produced outputs for " << chunks << " data chunks.\n";
    log << "Time per data chunk: " << ms / (chunks ? chunks : 1) << " milliseconds.\n";
    log << "Approximate datarate: " << rate << "GB/s" << std::endl;
    log << "Synchronized" << std::endl;
    if (res) {
        res->observation_time_ms = ms;
        res->blocks = blocks;
        res->data_chunks = chunks;
        res->gbytes_per_s = rate;
        res->beam_out.assign(beam_out, beam_out + n_f_per_detect * n_streams);
        res->last_gemm = last_gemm;
    }
    return BF_OK;
}

}  // namespace dsabf

// ======================================== C wrappers (include/dsabf_host.h) =========================================
using namespace dsabf;

struct bfh_generator {
    test_data_generator* g;
    bf_config cfg;
};
struct bfh_obs {
    observation_loop_state* o;
    event_backend* backend;
    fake_backend* fake;
};

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
    if (!cfg || !out) return BF_ERR_INVALID;
    bfh_obs* o = new bfh_obs{nullptr, nullptr, nullptr};
    if (h) {
        o->backend = make_hip_event_backend(h);
    } else {
        o->fake = new fake_backend();
        o->backend = o->fake;
    }
    o->o = new observation_loop_state(mts, mtot, *cfg, o->backend, debug_mode != 0);
    *out = o;
    return BF_OK;
}
int bfh_obs_destroy(bfh_obs* o)
{
    if (o) {
        delete o->o;
        delete o->backend;
        delete o;
    }
    return BF_OK;
}
int bfh_obs_generate_transfer_event(bfh_obs* o) { return o ? (o->o->generate_transfer_event(), BF_OK) : BF_ERR_INVALID; }
int bfh_obs_generate_analysis_event(bfh_obs* o) { return o ? (o->o->generate_analysis_event(), BF_OK) : BF_ERR_INVALID; }
int bfh_obs_check_transfer_events(bfh_obs* o) { return o ? (o->o->check_transfer_events(), BF_OK) : BF_ERR_INVALID; }
int bfh_obs_check_analysis_events(bfh_obs* o) { return o ? (o->o->check_analysis_events(), BF_OK) : BF_ERR_INVALID; }
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
int bfh_obs_fake_complete(bfh_obs* o, int nt, int na)
{
    if (!o || !o->fake) return BF_ERR_INVALID;
    o->fake->complete(nt, na);
    return BF_OK;
}

int bfh_run_debug_observation(const bf_config* cfg, int gpu, const char* positions, const char* directions,
                              const char* sources, const char* output, int device, int verbose, float* ded_out,
                              size_t ded_capacity, int* n_pt_sources, float* observation_ms)
{
    if (!cfg) return BF_ERR_INVALID;
    debug_run_options opt;
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
                    int burn_in, int verbose, detected_sink* sink, observation_result* res, void* ring_copy)
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
    if (!cfg || !name) return BF_ERR_INVALID;
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
    observation_result res;
    std::streambuf* keep = std::cout.rdbuf();
    if (!verbose) std::cout.rdbuf(quiet.rdbuf());
    int rc = run_observation(*cfg, opt, src, pos.data(), dir.data(), &res, log);
    std::cout.rdbuf(keep);
    if (rc != BF_OK) return rc;
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

}  // extern "C"
