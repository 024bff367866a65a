// bf_scheduler.cpp -- host mirror, part 3: observation_loop_state (src/observation_loop.hh:1-177) on an
// event backend, and the two branches of the reference's main loop (src/beamformer.cu:364-534) on the C-ABI of dsabf.h:
// run_debug_observation (DEBUG) and run_observation (production).
#include "../../include/dsabf_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <iostream>
#include <memory>
#include <sstream>
#include <thread>
#include <unistd.h>

#include "../../include/dsabf_host.h"
#include "bf_host_internal.h"

namespace dsabf {

// ---- event backends ---------------------------------------------------------------------------------------------
namespace {

struct hip_backend : event_backend {
    bf_handle* h;
    explicit hip_backend(bf_handle* hh) : h(hh) {}
    void* create() override
    {
        bf_event* e = nullptr;
        if (bf_event_create_on(h, &e) != BF_OK) return nullptr;   // on the handle's device, not the caller's current one
        return e;
    }
    void destroy(void* ev) override { bf_event_destroy(static_cast<bf_event*>(ev)); }
    int record_transfer(void* ev) override { return ev ? bf_record_transfer_event(h, static_cast<bf_event*>(ev)) : BF_ERR_INVALID; }
    int record_analysis(void* ev) override { return ev ? bf_record_analysis_event(h, static_cast<bf_event*>(ev)) : BF_ERR_INVALID; }
    int query(void* ev) override { return ev ? bf_event_query(static_cast<bf_event*>(ev)) : BF_ERR_INVALID; }
};

}  // namespace

event_backend* make_hip_event_backend(bf_handle* h) { return new hip_backend(h); }

// ---- observation_loop_state (src/observation_loop.hh:54-176) -------------------------------------------------------
observation_loop_state::observation_loop_state(uint64_t max_tsep, uint64_t max_totsep, const bf_config& cfg,
                                               event_backend* backend, bool debug)
    // The device ring has n_blocks_on_gpu slots and block j lives in slot j % n_blocks_on_gpu until its analysis has COMPLETED: a
    // transfer may run at most n_blocks_on_gpu blocks ahead of blocks_analyzed, or it overwrites voltages a kernel is still reading.
    // The reference never meets the case (N_BLOCKS_ON_GPU 8 > MAX_TOTAL_SEP 4, both compile-time: src/beamformer.hh:85,124); with
    // a run-time ring the separations are held to its size (found by the random DEBUG-flow fuzz, round 5: 2 slots, 1 run in 400).
    : maximum_transfer_seperation(std::min<uint64_t>(max_tsep, (uint64_t)std::max(cfg.n_blocks_on_gpu, 1))),
      maximum_total_seperation(std::min<uint64_t>(max_totsep, (uint64_t)std::max(cfg.n_blocks_on_gpu, 1))), debug_mode(debug),
      verbose(cfg.verbose != 0), n_gemms_per_block(cfg.n_gemms_per_block), n_blocks_on_gpu(cfg.n_blocks_on_gpu),
      n_events(5 * cfg.n_blocks_on_gpu), ev(backend)
{
    BlockTransferredSync.resize(n_events);
    BlockAnalyzedSync.resize(n_events);
    for (int i = 0; i < n_events; i++) {  // :58-61
        BlockTransferredSync[i] = ev->create();
        BlockAnalyzedSync[i] = ev->create();
        if (!BlockTransferredSync[i] || !BlockAnalyzedSync[i]) fail(BF_ERR_DEVICE);
    }
}

void observation_loop_state::fail(int code)
{
    if (error == BF_OK) {
        error = code < 0 ? code : BF_ERR_DEVICE;
        std::cerr << "observation_loop_state: event backend failed (" << error << "): " << bf_last_error() << std::endl;
    }
}

observation_loop_state::~observation_loop_state()
{
    for (int event = 0; event < n_events; event++) {  // :65-68
        if (BlockAnalyzedSync[event]) ev->destroy(BlockAnalyzedSync[event]);
        if (BlockTransferredSync[event]) ev->destroy(BlockTransferredSync[event]);
    }
}

void observation_loop_state::generate_transfer_event()
{
    if (error != BF_OK) return;
    void* e = BlockTransferredSync[blocks_transfer_queue % n_events];
    const int rc = e ? ev->record_transfer(e) : BF_ERR_DEVICE;  // :73
    if (rc != BF_OK) return fail(rc);
    blocks_transfer_queue++;
}

void observation_loop_state::generate_analysis_event()
{
    if (error != BF_OK) return;
    void* e = BlockAnalyzedSync[blocks_analysis_queue % n_events];
    const int rc = e ? ev->record_analysis(e) : BF_ERR_DEVICE;  // :79
    if (rc != BF_OK) return fail(rc);
    blocks_analysis_queue++;
}

void observation_loop_state::check_transfer_events()
{
    if (error != BF_OK) return;
    for (uint64_t event = blocks_transferred; event < blocks_transfer_queue; event++) {  // :85
        const int rc = ev->query(BlockTransferredSync[event % n_events]);
        if (rc == BF_OK) {
            if (verbose) std::cout << "Block " << event << " transfered to GPU" << std::endl;
            blocks_transferred++;
            ev->destroy(BlockTransferredSync[event % n_events]);  // :95-96 destroy and recreate
            BlockTransferredSync[event % n_events] = ev->create();
            if (!BlockTransferredSync[event % n_events]) return fail(BF_ERR_DEVICE);
        } else if (rc < 0) {
            return fail(rc);  // the reference dies in gpuErrchk here; never treat an error as "not ready yet"
        } else {
            break;  // :98
        }
    }
}

void observation_loop_state::check_analysis_events()
{
    if (error != BF_OK) return;
    for (uint64_t event = blocks_analyzed; event < blocks_analysis_queue; event++) {  // :104
        const int rc = ev->query(BlockAnalyzedSync[event % n_events]);
        if (rc == BF_OK) {
            blocks_analyzed++;
            if (verbose) std::cout << "Block " << event << " Analyzed" << std::endl;
            ev->destroy(BlockAnalyzedSync[event % n_events]);  // :112-113
            BlockAnalyzedSync[event % n_events] = ev->create();
            if (!BlockAnalyzedSync[event % n_events]) return fail(BF_ERR_DEVICE);
        } else if (rc < 0) {
            return fail(rc);
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

void dm_trial_share(int n_dm, int world, int rank, int* first, int* count)
{
    const int base = world > 0 ? n_dm / world : 0, extra = world > 0 ? n_dm % world : 0;
    if (count) *count = (rank >= 0 && rank < world) ? base + (rank < extra ? 1 : 0) : 0;
    if (first) *first = rank * base + std::min(std::max(rank, 0), extra);
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
    float* beam_out_blk[2] = {nullptr, nullptr};   // block_launch: a block's detected powers on the host, per alternating queue
    uint64_t block_launches = 0;
    for (int q = 0; q < 2 && opt.block_launch; q++) {   // (page-locking 2 x N_GEMMS_PER_BLOCK x 2 MiB is setup, like beam_out)
        void* pb = nullptr;
        if ((rc = bf_alloc_pinned(&pb, n_f_per_detect * (size_t)cfg.n_gemms_per_block * sizeof(float))) != BF_OK) return rc;
        g.pinned.push_back(pb);
        beam_out_blk[q] = static_cast<float*>(pb);
        float* d_blk = nullptr;                        // and the queue's device-side block buffer (allocated at first use otherwise)
        if ((rc = bf_block_output_device(h, q % n_streams, &d_blk)) != BF_OK) return rc;
    }

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

        if (obs_state.check_ready_for_analysis() && opt.block_launch) {
            // Block-granular analysis: one fused launch over the block's gemm-units, their detected powers to the host in
            // ONE copy (a4: every unit still travels, as src/beamformer.cu:485-488 has it), one DM-0 launch for all of
            // them (a8).  Two queues alternate so that block i+1's kernel overlaps block i's copy.
            const int q = (int)(block_launches++ % 2) % n_streams;
            const int first_gemm = (int)obs_state.get_current_analysis_gemm(0);
            if (opt.verbose) log << "Queueing Beamforming. Start Dir = " << first_gemm << std::endl;
            std::vector<float*> dst((size_t)cfg.n_gemms_per_block);
            for (int u = 0; u < cfg.n_gemms_per_block; u++) dst[(size_t)u] = beam_out_blk[q] + n_f_per_detect * (size_t)u;
            rc = bf_enqueue_block(h, q, (int)obs_state.get_next_gpu_analysis_block(), 0, cfg.n_gemms_per_block, dst.data());
            const int n_valid = std::min(cfg.n_gemms_per_block, n_src - first_gemm);   // check_ready_for_dh2_transfer, :492
            if (rc == BF_OK && n_valid > 0)
                rc = bf_enqueue_block_dedisperse(h, q, 0, n_valid, &dedispersed_out[(size_t)first_gemm * cfg.n_beams]);
            if (rc != BF_OK) {
                log << "GPUassert: " << bf_last_error() << std::endl;
                return rc;
            }
            (void)obs_state.get_current_analysis_gemm(cfg.n_gemms_per_block - 1);   // most_recent_gemm = the block's last unit
            obs_state.generate_analysis_event();  // :525
        } else if (obs_state.check_ready_for_analysis()) {  // :452
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
        if (obs_state.status() != BF_OK) {  // a failed record / query: the reference exits in gpuErrchk, never keep polling
            log << "GPUassert: event backend failed: " << bf_last_error() << std::endl;
            return obs_state.status();
        }
    }

    bf_timer_stop(h, &time_accumulator_ms);  // :540
    observation_time_ms += time_accumulator_ms;
    const long long chunks = (long long)n_src * cfg.n_out_per_gemm;
    log << "Observation ran in " << observation_time_ms << "milliseconds.\n";
    log << "Code produced outputs for " << chunks << " data chunks.\n";
    log << "Time per data chunk: " << observation_time_ms / chunks << " milliseconds.\n";
    log << "Approximate datarate: " << bf_bytes_per_gemm(&cfg) * (double)n_src / observation_time_ms / 1e6 << "GB/s"
        << std::endl;
    // (not a line of the reference: which caller-side loop the two timing lines above were measured with)
    log << "Launch pattern: " << (opt.block_launch ? "one bf_enqueue_block per PSRDADA block"
                                                    : "the reference's loop, one bf_enqueue_gemm_unit per gemm-unit (src/beamformer.cu:454-519)")
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

// ---- production observation loop (src/beamformer.cu:364-534, #ifndef DEBUG branches) -------------------------------------
int run_observation(const bf_config& cfg, const observation_options& opt, block_source& source, const antenna* pos,
                    const beam_direction* dir, observation_result* res, std::ostream& log)
{
    // A sharded run enters its first collective -- the gather behind block 0 -- only if EVERY shard got through its setup: each rank
    // takes part in ONE exchange of "ready" flags before the loop, whichever way its setup went (a rank that bailed out alone would
    // leave the others waiting in that gather for ever: ADVICE r05).  The caller's own preparations count too (local_setup_ok).
    struct setup_gate {
        bf_comm* comm;
        bool done = false;
        int fail(int rc)
        {
            if (comm && !done) {
                done = true;
                bool all = false;
                (void)comm_all_ok(comm, false, &all);
            }
            return rc;
        }
    } gate{opt.comm};
    const int n_streams = cfg.n_streams;
    if (cfg.n_gemms_per_block % n_streams) return gate.fail(BF_ERR_INVALID);
    if (opt.world < 1 || opt.rank < 0 || opt.rank >= opt.world || opt.gather_root < BF_GATHER_ROOT_ALL || opt.gather_root >= opt.world)
        return gate.fail(set_error(BF_ERR_INVALID, "run_observation: need 0 <= rank < world and gather_root a rank or BF_GATHER_ROOT_ALL"));
    // who holds the gathered band after every block: one rank, or (BF_GATHER_ROOT_ALL) every rank
    const bool is_root = !opt.comm || opt.gather_root == BF_GATHER_ROOT_ALL || opt.rank == opt.gather_root;
    if (opt.comm && (bf_comm_rank(opt.comm) != opt.rank || bf_comm_world(opt.comm) != opt.world))
        return gate.fail(set_error(BF_ERR_INVALID, "run_observation: the communicator's rank / world differ from the options'"));
    if (opt.comm && !opt.block_launch)
        return gate.fail(set_error(BF_ERR_INVALID, "run_observation: the sharded gather needs block-granular launches"));
    if (opt.comm && opt.sink && !is_root)
        return gate.fail(set_error(BF_ERR_INVALID, "run_observation: only the gather root may have a sink"));
    if (opt.dm_delays && opt.n_dm <= 0) return gate.fail(set_error(BF_ERR_INVALID, "run_observation: dm_delays without n_dm"));
    if (opt.dm_sink && !opt.dm_delays) return gate.fail(set_error(BF_ERR_INVALID, "run_observation: a dm_sink needs dm_delays"));
    if (opt.comm && opt.dm_sink && !is_root)
        return gate.fail(set_error(BF_ERR_INVALID, "run_observation: only the gather root may have a dm_sink"));
    if (opt.dm_split_trials && (!opt.comm || opt.gather_root != BF_GATHER_ROOT_ALL))
        return gate.fail(set_error(BF_ERR_INVALID, "run_observation: dm_split_trials needs a sharded run gathered to every rank (BF_GATHER_ROOT_ALL)"));
    source.read_headers();  // :334 (before the device exists here: the block-size check below needs no GPU)
    const size_t block_bytes = bf_bytes_per_block(&cfg);
    if (source.get_block_size() != block_bytes) {
        // The reference prints this (src/beamformer.cu:336-339) and carries on; every ring block is then copied with the
        // GEOMETRY's size: a smaller ring block is read past its end (past the mapping on the last one), a larger one is
        // silently truncated.  Neither can produce a meaningful beam: refuse.
        log << "ERROR: block size " << source.get_block_size() << ", Should also be " << block_bytes << std::endl;
        return gate.fail(set_error(BF_ERR_INVALID, "run_observation: the source's block size does not match bf_bytes_per_block(cfg)"));
    }

    bf_handle* h = nullptr;
    int rc = bf_create(&cfg, opt.device, &h);
    if (rc != BF_OK) {
        log << "GPUassert: " << bf_last_error() << std::endl;
        return gate.fail(rc);
    }
    struct guard {
        bf_handle* h;
        void* pinned;
        bf_dm_stream* dm = nullptr;
        std::vector<void*> more_pinned;
        ~guard()
        {
            if (h) bf_stream_sync(h, -1);
            bf_dm_stream_destroy(dm);   // (before its handle)
            bf_free_pinned(pinned);
            for (void* p : more_pinned) bf_free_pinned(p);
            bf_destroy(h);
        }
    } g{h, nullptr};

    const size_t n_f_per_detect = bf_floats_per_detect(&cfg);
    // beam_out: the reference's pinned D2H destination (:249); the root of a sharded run receives the whole band there
    const size_t beam_out_stride = n_f_per_detect * (size_t)(opt.comm ? opt.world : 1);
    if ((rc = bf_alloc_pinned(&g.pinned, beam_out_stride * n_streams * sizeof(float))) != BF_OK) return gate.fail(rc);
    float* beam_out = static_cast<float*>(g.pinned);
    ::memset(beam_out, 0, beam_out_stride * n_streams * sizeof(float));
    {
        std::vector<int8_t> fourier_coefficients((size_t)cfg.n_freq * cfg.n_ant * cfg.n_beams * 2);
        generate_fourier_coefficients(cfg.n_beams, cfg.n_ant, cfg.n_freq, opt.rank * cfg.n_freq, opt.gpu, pos, dir,
                                      fourier_coefficients.data());
        if ((rc = bf_set_weights(h, fourier_coefficients.data())) != BF_OK) return gate.fail(rc);
    }
    std::vector<int> timeSlice((size_t)n_streams);
    for (int i = 0; i < n_streams; i++) timeSlice[i] = i;  // :319
    std::vector<long long> last_gemm((size_t)n_streams, -1);
    std::vector<float*> unit_dst;

    hip_backend backend(h);
    observation_loop_state obs_state(kMaxTransferSep, kMaxTotalSep, cfg, &backend, /*debug_mode=*/false);  // :322
    if (opt.burn_in > 0) {  // :348-355
        log << "Burning IN" << std::endl;
        for (int i = 0; i < opt.burn_in; i++) {
            source.read();
            source.close();
        }
        log << "Done Burn in" << std::endl;
    }
    uint64_t sink_committed = 0;
    const char* unit_env = lab_getenv("DSABF_UNIT_LAUNCH");   // measurement / test switch (DSABF_LAB=1): the reference's per-gemm-unit launches
    const bool block_launch = opt.block_launch && !(unit_env && unit_env[0] == '1');
    if (opt.comm && !block_launch) return gate.fail(set_error(BF_ERR_INVALID, "run_observation: the sharded gather needs block-granular launches"));
    int upl = opt.units_per_launch;                                // gemm-units per launch of the block path
    if (const char* e = lab_getenv("DSABF_UNITS_PER_LAUNCH")) upl = atoi(e);
    if (upl <= 0 || upl > cfg.n_gemms_per_block || opt.comm) upl = cfg.n_gemms_per_block;   // 0 / sharded: the whole block
    while (cfg.n_gemms_per_block % upl) upl--;                    // whole launches only
    uint64_t launch_seq = 0;
    const int n_queues_used = upl == cfg.n_gemms_per_block ? std::min(n_streams, 2) : n_streams;   // (the rotation below)
    bool staged = opt.gather_staged;
    if (const char* e = lab_getenv("DSABF_GATHER_STAGED")) staged = e[0] == '1';
    // ---- the DM stage: where the reference's loop has its frequency collapse, src/beamformer.cu:492-511 ----
    const bool dm_here = opt.dm_delays && is_root;   // a sharded run dedisperses the gathered band
    // dm_split_trials: every rank holds the whole band (one all-gather instead of a gather) and takes ITS share of the trial ladder --
    // the DM work scales with the GPUs, the path still has ONE collective; rank r: trials [r n / R, (r + 1) n / R), the first n % R
    // ranks one more
    int dm_first = 0, dm_count = opt.n_dm;
    if (opt.dm_split_trials) dm_trial_share(opt.n_dm, opt.world, opt.rank, &dm_first, &dm_count);
    const int n_freq_band = cfg.n_freq * (opt.comm ? opt.world : 1);
    const int dm_rows = upl * cfg.n_out_per_gemm;                                       // beam-blocks per launch = rows per push
    struct dm_chunk {
        uint64_t block, first_t;
        int n_t;
        float* host;
    };
    std::deque<dm_chunk> dm_pending;            // pushed, not yet delivered (their block's analysis event has not fired)
    std::vector<float*> dm_host;                // pinned chunk buffers, used round robin
    uint64_t dm_seq = 0, dm_times = 0, dm_chunks = 0;
    if (opt.dm_delays && !block_launch) return gate.fail(set_error(BF_ERR_INVALID, "run_observation: the DM stage needs block-granular launches"));
    const bool dm_run = dm_here && dm_count > 0;      // (more ranks than trials: the surplus ranks only beamform)
    if (dm_run) {
        if ((rc = bf_dm_stream_create(h, opt.dm_delays + (size_t)dm_first * n_freq_band, dm_count, n_freq_band, dm_rows, &g.dm)) != BF_OK) {
            log << "GPUassert: " << bf_last_error() << std::endl;
            return gate.fail(rc);
        }
        // chunks in flight: the launches of MAX_TOTAL_SEP blocks queued + the one being delivered
        const size_t n_buf = (size_t)(kMaxTotalSep + 2) * (size_t)(cfg.n_gemms_per_block / upl);
        for (size_t i = 0; i < n_buf; i++) {
            void* pb = nullptr;
            if ((rc = bf_alloc_pinned(&pb, (size_t)dm_count * dm_rows * cfg.n_beams * sizeof(float))) != BF_OK) return gate.fail(rc);
            g.more_pinned.push_back(pb);
            dm_host.push_back(static_cast<float*>(pb));
        }
    }

    if (block_launch)   // the per-queue block buffers are allocated on first use: do that here, not inside the timed loop
        for (int q = 0; q < n_queues_used; q++) {
            float* unused = nullptr;
            if ((!dm_run || opt.comm) && (rc = bf_block_output_device(h, q, &unused)) != BF_OK) return gate.fail(rc);   // (a DM stage takes the rows itself)
            if (opt.comm && is_root && !dm_run && (rc = bf_block_gather_device(h, q, opt.world, &unused)) != BF_OK) return gate.fail(rc);
            if (opt.comm && staged && is_root && (rc = bf_block_gather_stage_device(h, q, opt.world, &unused)) != BF_OK) return gate.fail(rc);
        }
    if (opt.comm) {
        gate.done = true;
        bool all = false;
        if ((rc = comm_all_ok(opt.comm, opt.local_setup_ok, &all)) != BF_OK) return rc;
        if (!all)
            return set_error(BF_ERR_STATE, opt.local_setup_ok ? "run_observation: another shard of this run failed its setup; nothing was started"
                                                              : "run_observation: this shard's caller reported a failed setup (local_setup_ok = false)");
    } else if (!opt.local_setup_ok) {
        return set_error(BF_ERR_STATE, "run_observation: the caller reported a failed setup (local_setup_ok = false)");
    }
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
                    while (obs_state.status() == BF_OK &&
                           obs_state.get_blocks_transferred() < obs_state.get_blocks_transfer_queue())
                        obs_state.check_transfer_events();
            } else {
                obs_state.set_transfers_complete(true);  // :398
            }
            source.close();  // :401
        }
        obs_state.check_transfer_events();  // :446
        if (obs_state.check_ready_for_analysis()) {  // :452
            const long long block_index = (long long)obs_state.get_blocks_analysis_queue();
            if (block_launch) {
                // unit u's powers go to its sink slot, or to beam_out[queue of its launch] (later units on that queue
                // overwrite earlier ones, in order: beam_out[q] ends up holding gemm-unit last_gemm[q])
                const int n_units = cfg.n_gemms_per_block;
                unit_dst.assign((size_t)n_units, nullptr);
                for (int first = 0; first < n_units; first += upl) {
                    // whole blocks alternate between TWO queues (block i + 1's kernel under block i's copies, as in the DEBUG flow):
                    // more queues only put more concurrent host copies beside the H2D stream (9.6 -> 9.2 us per beam-block, and
                    // 11.6 -> 9.x with other streams alive in the process: tools/stream_queues.py); sub-block launches rotate over all
                    const int q = (int)(launch_seq++ % (uint64_t)n_queues_used);
                    for (int u = first; u < first + upl; u++) {
                        unit_dst[u] = &beam_out[(size_t)q * beam_out_stride];
                        if (opt.sink) {
                            unit_dst[u] = opt.sink->acquire((uint64_t)block_index * n_units + u);
                            if (!unit_dst[u]) {
                                log << "ERROR: detected sink has no free slot" << std::endl;
                                return BF_ERR_STATE;
                            }
                        }
                        last_gemm[q] = block_index * n_units + u;
                    }
                    // The DM stage owns the place its rows belong in -- directly behind its carried-over delay window: the beamformer
                    // (one GPU) or the gather (a holder of a sharded run) writes them THERE, and the push below only launches.  The
                    // reference's collapse reads what detect wrote, no copy in between (src/beamformer.cu:481,492-511).
                    float* d_rows = nullptr;     // where this launch's beam-blocks are on the device, [row][freq (over the band)][beam]
                    void* dm_qs = nullptr;
                    if (dm_run && (rc = bf_queue_stream(h, q, &dm_qs)) == BF_OK) rc = bf_dm_stream_reserve(g.dm, dm_rows, &d_rows, dm_qs);
                    if (rc == BF_OK)
                        rc = (dm_run && !opt.comm) ? bf_enqueue_block_to(h, q, (int)obs_state.get_next_gpu_analysis_block(), first, upl, d_rows,
                                                                         &unit_dst[first])
                                                   : bf_enqueue_block(h, q, (int)obs_state.get_next_gpu_analysis_block(), first, upl,
                                                                      opt.comm ? nullptr : &unit_dst[first]);
                    if (rc == BF_OK && opt.comm) {
                        // sharded (upl == n_units): bring the shards' powers together on the root, in [unit][o][f over the
                        // band][b], behind the launch on the same queue; only the root copies to the host
                        float *d_blk = nullptr, *d_full = nullptr, *d_stage = nullptr;
                        void* qs = nullptr;
                        const size_t full_det = n_f_per_detect * (size_t)opt.world;
                        const bool root = is_root;
                        if (dm_run) d_full = d_rows;     // (a holder with a DM stage receives straight into the stage's buffer)
                        if ((rc = bf_block_output_device(h, q, &d_blk)) == BF_OK && (rc = bf_queue_stream(h, q, &qs)) == BF_OK &&
                            (!root || dm_run || (rc = bf_block_gather_device(h, q, opt.world, &d_full)) == BF_OK) &&
                            (!root || !staged || (rc = bf_block_gather_stage_device(h, q, opt.world, &d_stage)) == BF_OK))
                            rc = staged ? bf_gather_detected_staged(opt.comm, d_blk, (size_t)n_units * cfg.n_out_per_gemm,
                                                                    (size_t)cfg.n_freq * cfg.n_beams, opt.gather_root, d_full, d_stage, qs)
                                        : bf_gather_detected(opt.comm, d_blk, (size_t)n_units * cfg.n_out_per_gemm,
                                                             (size_t)cfg.n_freq * cfg.n_beams, opt.gather_root,
                                                             BF_GATHER_LAYOUT_FREQ_MAJOR, d_full, qs);
                        for (int u = 0; rc == BF_OK && root && u < n_units; u++)
                            rc = bf_enqueue_d2h(h, q, d_full + full_det * (size_t)u, unit_dst[u], full_det);
                    }
                    if (rc == BF_OK && dm_run) {
                        // the DM stage, where the reference's loop collapses frequency (src/beamformer.cu:492-511): this launch's rows
                        // are in the stream's buffer already; its kernels run on the launch's own queue, the chunk that becomes complete
                        // travels to a pinned buffer
                        uint64_t first_t = 0;
                        int n_t = 0;
                        float* chunk = dm_host[(size_t)(dm_seq++ % dm_host.size())];
                        rc = bf_dm_stream_push(g.dm, d_rows, dm_rows, chunk, &first_t, &n_t, dm_qs);
                        if (rc == BF_OK && n_t > 0) dm_pending.push_back({(uint64_t)block_index, first_t, n_t, chunk});
                    }
                    if (rc != BF_OK) {
                        log << "GPUassert: " << bf_last_error() << std::endl;
                        return rc;
                    }
                }
            } else {  // the reference's launch pattern: one gemm-unit per launch, round-robin over the compute queues
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
            }
            obs_state.generate_analysis_event();  // :525
        }
        obs_state.check_analysis_events();  // :532
        if (obs_state.status() != BF_OK) {  // a failed record / query: the reference exits in gpuErrchk, never keep polling
            log << "GPUassert: event backend failed: " << bf_last_error() << std::endl;
            return obs_state.status();
        }
        // the DM chunks of analysed blocks have landed too (their copies sit on the queues the analysis event joins)
        while (!dm_pending.empty() && dm_pending.front().block < obs_state.get_blocks_analyzed()) {
            const dm_chunk c = dm_pending.front();
            dm_pending.pop_front();
            dm_times += (uint64_t)c.n_t;
            if (opt.dm_sink) {
                dm_chunks++;
                if (!opt.dm_sink->deliver(c.first_t, c.n_t, dm_count, cfg.n_beams, c.host)) {
                    log << "ERROR: DM sink failed at output time " << c.first_t << std::endl;
                    return BF_ERR_STATE;
                }
            }
        }
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
    if (opt.dm_sink) opt.dm_sink->close();
    const uint64_t blocks = obs_state.get_blocks_analyzed();
    const uint64_t chunks = obs_state.get_current_transfer_gemm() * cfg.n_out_per_gemm;  // :552
    const double rate = (double)source.get_block_size() * obs_state.get_blocks_transfer_queue() / ms / 1e6;  // :554
    log << "Observation ran in " << ms << "milliseconds.\n";
    log << "Code produced outputs for " << chunks << " data chunks.\n";
    log << "Time per data chunk: " << ms / (chunks ? chunks : 1) << " milliseconds.\n";
    log << "Approximate datarate: " << rate << "GB/s" << std::endl;
    log << "Launch pattern: " << (block_launch ? "bf_enqueue_block" : "the reference's loop, one bf_enqueue_gemm_unit per gemm-unit (src/beamformer.cu:454-519)")
        << std::endl;
    log << "Synchronized" << std::endl;
    if (res) {
        res->observation_time_ms = ms;
        res->blocks = blocks;
        res->data_chunks = chunks;
        res->gbytes_per_s = rate;
        res->beam_out.assign(beam_out, beam_out + beam_out_stride * n_streams);
        res->last_gemm = last_gemm;
        res->dm_times = dm_times;
        res->dm_chunks = dm_chunks;
    }
    if (dm_run)
        log << "DM stage: trials " << dm_first << " .. " << dm_first + dm_count - 1 << " of " << opt.n_dm << ", " << dm_times
            << " output times (largest delay " << bf_dm_stream_max_delay(g.dm) << " samples carried over on the device)" << std::endl;
    return BF_OK;
}

}  // namespace dsabf
