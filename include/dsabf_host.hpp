// dsabf_host.hpp -- C++ host-side mirror of the reference's host classes and helpers, on top of the C-ABI
// (include/dsabf.h).  A reference-style main() can include this header, link libdsabf.so and keep its structure:
// same class names, method names, argument meaning and loop semantics as
//   observation_loop_state   src/observation_loop.hh:1-177
//   test_data_generator      src/test_data_generator.hh:11-108
//   antenna / beam_direction src/beamformer.hh:170-213, readers :250-284, python writer :287-311
//   steering weights         src/beamformer.cu:230-241 (inline in the reference's main)
// Differences, on purpose: geometry is runtime (bf_config) instead of #defines; errors are return codes /
// exceptions instead of exit(); the analysis-complete event is recorded behind ALL compute queues (the
// reference records on stream[7] only -- SURVEY.md section 5, latent race 2).
#pragma once

#include <condition_variable>
#include <cstdint>
#include <iosfwd>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "dsabf.h"

namespace dsabf {

// ---- reference constants (src/beamformer.hh:45-85) -----------------------------------------------------
constexpr int kBurnIn = 25;            // BURNIN
constexpr double kHalfFov = 3.5;       // HALF_FOV, degrees
constexpr int kNGpus = 8;              // N_GPUS
constexpr int kTotChannels = 2048;     // TOT_CHANNELS
constexpr double kStartF = 1.28;       // START_F (GHz)
constexpr double kEndF = 1.53;         // END_F
constexpr int kZeroPt = 0;             // ZERO_PT
constexpr double kCSpeed = 299792458.0;
constexpr double kPi = 3.14159265358979;
constexpr int kMaxVal = 127;           // MAX_VAL
constexpr int kSigMaxVal = 7;          // SIG_MAX_VAL
constexpr int kMaxTransferSep = 2;     // MAX_TRANSFER_SEP
constexpr int kMaxTotalSep = 4;        // MAX_TOTAL_SEP
constexpr int kSourcesPerBatch = 1024; // N_SOURCES_PER_BATCH
constexpr unsigned char kBogusData = 0x70;  // BOGUS_DATA, src/test_data_generator.hh:8

// src/beamformer.hh:170-213
class antenna {
public:
    float x = 0, y = 0, z = 0;
};
std::istream& operator>>(std::istream& in, antenna& a);
std::ostream& operator<<(std::ostream& out, const antenna& a);

class beam_direction {
public:
    float theta = 0, phi = 0;
    beam_direction() {}
    beam_direction(float th, float ph) : theta(th), phi(ph) {}
};
std::istream& operator>>(std::istream& in, beam_direction& a);
std::ostream& operator<<(std::ostream& out, const beam_direction& a);

// src/beamformer.hh:250-284.  Return 0, or -1 if the file cannot be opened (the reference does not check).
int read_in_beam_directions(const char* file_name, int expected_beams, beam_direction* dir);
int read_in_position_locations(const char* file_name, int n_antennas, antenna* pos);
// src/beamformer.hh:287-311
int write_array_to_disk_as_python_file(const float* data_out, int rows, int cols, const char* output_filename);
// src/beamformer.hh:314-350 (runtime values)
void print_all_defines(const bf_config& cfg, std::ostream& out);
// src/beamformer.hh:222-243
void usage(bool debug_mode, std::ostream& out);

// src/beamformer.cu:135-147: default linear array / beam fan when no -p / -d file is given
void default_positions(int n_antennas, antenna* pos);
void default_directions(int n_beams, beam_direction* dir);

// Channel centre frequencies (GHz), both of the reference's variants (float bw in main(), double macro in the
// generator); integer division gpu*2048/7 kept verbatim.
float channel_frequency_weights(int gpu, int chan);    // src/beamformer.cu:173,233
float channel_frequency_generator(int gpu, int chan);  // src/test_data_generator.hh:72

// src/beamformer.cu:230-241: int8 steering weights, layout [freq][ant][beam]{re,im}; channels chan0 ..
// chan0+n_freq-1 of sub-band `gpu` (chan0 > 0 is how a frequency shard of a multi-GPU job gets its slice).
void generate_fourier_coefficients(int n_beams, int n_antennas, int n_freq, int chan0, int gpu, const antenna* pos,
                                   const beam_direction* dir, int8_t* out);

// ---- test_data_generator (src/test_data_generator.hh:11-108) --------------------------------------------
class test_data_generator {
private:
    bf_config cfg;
    int n_pt_sources = 1024;
    int n_source_batches = 1;
    int source_batch_counter = 0;
    std::vector<beam_direction> sources;
    bool use_source_catalog = false;
    char* data = nullptr;      // pinned (bf_alloc_pinned) when a GPU runtime is present, else plain host memory
    bool pinned = false;
    int n_sources_per_batch;

public:
    explicit test_data_generator(const bf_config& cfg, int n_sources_per_batch = kSourcesPerBatch, bool pin = true);
    ~test_data_generator();
    test_data_generator(const test_data_generator&) = delete;
    test_data_generator& operator=(const test_data_generator&) = delete;

    int get_n_pt_sources() const { return n_pt_sources; }
    char* get_data() const { return data; }
    size_t input_data_size() const;  // INPUT_DATA_SIZE, src/beamformer.hh:150
    int get_source_batch_counter() const { return source_batch_counter; }

    void generate_test_data(const antenna pos[], int gpu);
    int read_in_source_directions(const char* file_name);
    void set_source_directions(const beam_direction* src, int n);  // same effect as the file reader
    bool check_need_to_generate_more_input_data(int blocks_transfered);
    bool check_data_ready_for_transfer(int blocks_transfer_queue);
};

// ---- observation_loop_state (src/observation_loop.hh:1-177) ----------------------------------------------
// Event backend: the real one records/queries bf_event (HIP events); bfh_obs_create_custom (dsabf_host.h) plugs in any
// other implementation through a table of C callbacks (the CPU tests drive the scheduler that way, with no GPU).
// Every operation reports failure: the reference exits inside gpuErrchk (src/beamformer.cuh:19-29); here the error
// becomes the state's sticky status() and the loops return it instead of polling a dead device forever.
struct event_backend {
    virtual ~event_backend() {}
    virtual void* create() = 0;                  // nullptr on failure
    virtual void destroy(void* ev) = 0;
    virtual int record_transfer(void* ev) = 0;   // behind the last bf_submit_block; BF_OK or < 0
    virtual int record_analysis(void* ev) = 0;   // behind all compute queues; BF_OK or < 0
    virtual int query(void* ev) = 0;             // BF_OK done, BF_NOT_READY, < 0 error
};
event_backend* make_hip_event_backend(bf_handle* h);  // caller deletes

// (One difference from the reference, where both are compile-time and N_BLOCKS_ON_GPU 8 > MAX_TOTAL_SEP 4: the two separations
// are held to the ring's size, cfg.n_blocks_on_gpu -- a transfer must never overwrite a slot whose block is still being analysed.)
class observation_loop_state {
private:
    uint64_t blocks_analyzed = 0;
    uint64_t blocks_transferred = 0;
    uint64_t blocks_analysis_queue = 0;
    uint64_t blocks_transfer_queue = 0;
    uint64_t maximum_transfer_seperation;
    uint64_t maximum_total_seperation;
    bool transfers_complete = false;
    std::vector<void*> BlockTransferredSync;  // N_EVENTS_ON_GPU = 5 * N_BLOCKS_ON_GPU, src/beamformer.hh:128
    std::vector<void*> BlockAnalyzedSync;
    int most_recent_gemm = 0;
    int n_pt_sources = 0;
    bool debug_mode;
    bool verbose;
    int n_gemms_per_block, n_blocks_on_gpu, n_events;
    event_backend* ev;
    int error = 0;  // first backend failure (BF_ERR_*), sticky; 0 = BF_OK
    void fail(int code);

public:
    observation_loop_state(uint64_t maximum_transfer_seperation, uint64_t maximum_total_seperation,
                           const bf_config& cfg, event_backend* backend, bool debug_mode);
    ~observation_loop_state();
    observation_loop_state(const observation_loop_state&) = delete;
    observation_loop_state& operator=(const observation_loop_state&) = delete;

    void generate_transfer_event();  // the reference passes the stream; the backend knows the queue
    void generate_analysis_event();
    void check_transfer_events();
    void check_analysis_events();

    uint64_t get_blocks_analyzed() const { return blocks_analyzed; }
    uint64_t get_blocks_transferred() const { return blocks_transferred; }
    uint64_t get_blocks_analysis_queue() const { return blocks_analysis_queue; }
    uint64_t get_blocks_transfer_queue() const { return blocks_transfer_queue; }

    uint64_t get_current_analysis_gemm(int time_slice);
    uint64_t get_current_transfer_gemm() const;
    uint64_t get_next_gpu_analysis_block() const { return blocks_analysis_queue % n_blocks_on_gpu; }
    uint64_t get_next_gpu_transfer_block() const { return blocks_transfer_queue % n_blocks_on_gpu; }

    void set_transfers_complete(bool value) { transfers_complete = value; }
    bool get_transfers_complete() const { return transfers_complete; }

    bool check_ready_for_transfer() const;
    bool check_ready_for_analysis() const;
    bool check_ready_for_dh2_transfer(int time_slice);

    bool check_observations_complete();
    // BF_OK, or the first error any event operation reported (a failed record / create, a query that returned < 0).
    // Once set, no counter advances any more: callers must stop polling and return it.
    int status() const { return error; }
    void set_n_pt_sources(int val) { n_pt_sources = val; }  // DEBUG only in the reference
    bool check_transfers_complete();                        // DEBUG only in the reference

    friend std::ostream& operator<<(std::ostream& out, const observation_loop_state& a);
};

// ---- the reference's DEBUG main() as a callable (src/beamformer.cu:12-621, `make debug` flow) -------------
struct debug_run_options {
    int gpu = 0;                 // -g
    const char* positions = nullptr;   // -p
    const char* directions = nullptr;  // -d
    const char* sources = nullptr;     // -s
    const char* output = "bin/data.py";
    int device = 0;
    bool verbose = false;
    // launch granularity (as observation_options::block_launch): true = one fused launch, ONE D2H copy of the block's
    // detected powers and one dedisperse launch (bf_enqueue_block + bf_enqueue_block_dedisperse) per block of
    // N_GEMMS_PER_BLOCK sources, alternating between two compute queues; false = the reference's own pattern, per
    // gemm-unit launches + copies round-robin over the N_STREAMS queues (src/beamformer.cu:454-519).  Same data.py.
    bool block_launch = true;
};
struct debug_run_result {
    float observation_time_ms = 0;
    int n_pt_sources = 0;
    long long data_chunks = 0;  // n_pt_sources * N_OUTPUTS_PER_GEMM, src/beamformer.cu:544
};
// Runs generate -> H2D -> fused beamform -> dedisperse -> data.py for every source; dedispersed_out (optional)
// receives the [n_pt_sources][n_beams] table that is also written to opt.output.
int run_debug_observation(const bf_config& cfg, const debug_run_options& opt, debug_run_result* res,
                          std::vector<float>* dedispersed_out, std::ostream& log);

// ---- observation (production) mode: src/beamformer.cu:364-534 without -DDEBUG ------------------------------------
// The surface of dada_handler that the observation loop uses (src/dada_handler.hh:15-23).  The real PSRDADA
// implementation needs libpsrdada (not in this build); anything that hands out pinned blocks
// can stand behind this interface.
struct block_source {
    virtual ~block_source() {}
    virtual void read_headers() {}                 // src/dada_handler.hh:66-90
    virtual char* read() = 0;                      // :92-94  next block (the reference blocks on the shm semaphore)
    virtual void close() = 0;                      // :96-98
    virtual bool check_transfers_complete() = 0;   // :100-116 short block => the observation ends
    virtual uint64_t get_block_size() const = 0;
    virtual uint64_t get_bytes_read() const = 0;
    // true if close() lets a producer overwrite the block.  The reference closes the PSRDADA block right after
    // ENQUEUEING the asynchronous H2D copy (src/beamformer.cu:389-401), i.e. while the DMA may still be reading it -- a
    // latent race that a fast writer turns into corrupted voltages.  For such sources the loop waits for the block's
    // transfer event before close() (PSRDADA allows one open block, so the wait cannot be deferred past the next read).
    virtual bool close_releases_block() const { return false; }
};

// In-memory stand-in for `dada_junkdb` (makefile:28-29, README.md:173): a pinned ring of distinct pseudo-random
// blocks, served n_blocks times, then one short (empty) block.
class junk_block_source : public block_source {
    uint64_t block_size, bytes_read = 0, served = 0, n_blocks;
    int ring_blocks;
    char* ring = nullptr;
    bool pinned = false;

public:
    junk_block_source(const bf_config& cfg, uint64_t n_blocks, int ring_blocks = 4, uint64_t seed = 0xD5A);
    ~junk_block_source() override;
    junk_block_source(const junk_block_source&) = delete;
    junk_block_source& operator=(const junk_block_source&) = delete;
    char* read() override;
    void close() override {}
    bool check_transfers_complete() override;
    uint64_t get_block_size() const override { return block_size; }
    uint64_t get_bytes_read() const override { return bytes_read; }
    const char* ring_data() const { return ring; }           // tests: the bytes block i was served from are
    int get_ring_blocks() const { return ring_blocks; }      // ring_data() + (i % ring_blocks) * block_size
    bool ok() const { return ring != nullptr; }
};

// ---- dedispersion beyond DM 0: the formulas of sandbox/Dispersion Theory.ipynb ------------
// Trial ladder of cells 1-2 (double arithmetic, as numpy): appends trials until the first one >= dm_max.
std::vector<double> dm_trials(double dm0 = 0.0, double dm_max = 2000.0, int nchan = 2048, double epsilon = 1.25,
                              double nu_ghz = (1.28 + 1.53) / 2, double chan_bw_mhz = (1.53 - 1.28) / 2048 * 1000,
                              double ti_us = 40.0, double tscat_us = 0.0, double tsamp_us = 131.0);
// Cell 5: delay[dm][f] = (int)(4.15 * dm * (freq_f^-2 - f_ref^-2) / tsamp_ms); freq in GHz (channel_frequency()).
void dm_delays(const double* dms, int n_dm, const float* freq_ghz, int n_freq, double f_ref_ghz, double tsamp_ms,
               int32_t* out);

// dm_split_trials (observation_options): rank r of R takes trials [first, first + count) of an n_dm-trial ladder -- n_dm / R each,
// the first n_dm % R ranks one more; count may be 0 (more ranks than trials).
void dm_trial_share(int n_dm, int world, int rank, int* first, int* count);

// Fills `ring_blocks` consecutive blocks of cfg's block size at `ring` with the junk source's bytes (64-bit xorshift*,
// every nibble code in both halves; depends only on seed, ring_blocks and the block size).
void junk_fill(const bf_config& cfg, int ring_blocks, uint64_t seed, char* ring);

// ---- shared-memory input ring: the PSRDADA stand-in, csrc/bf_shmring.cpp -------------------
constexpr int kMaxRingBlocks = 64;
constexpr size_t kRingHeaderBytes = 4096;   // HDR_SIZE of config/correlator_header_dsaX.txt

class shm_ring {
    struct control;
    control* ctl = nullptr;
    size_t map_bytes = 0;
    shm_ring();

public:
    ~shm_ring();
    shm_ring(const shm_ring&) = delete;
    shm_ring& operator=(const shm_ring&) = delete;
    // `dada_db -k name -n n_blocks -b block_size` (README.md:151-160); header_text: the ASCII header block
    static shm_ring* create(const char* name, uint64_t n_blocks, uint64_t block_size, const char* header_text);
    static shm_ring* attach(const char* name, int timeout_ms = 10000);
    static int unlink(const char* name);    // `dada_db -k name -d`
    char* open_block_write();               // blocks while the ring is full
    void close_block_write(uint64_t bytes); // bytes < block_size: end of data
    char* open_block_read(uint64_t* bytes, uint64_t* block_id);  // blocks while the ring is empty
    void close_block_read();
    uint64_t get_n_blocks() const;
    uint64_t get_block_size() const;
    uint64_t get_header_size() const;
    const char* get_header() const;
    uint64_t get_blocks_written() const;
    uint64_t get_blocks_read() const;
    char* block(uint64_t slot) const;
};

// dada_handler (src/dada_handler.hh:1-177) on the shared-memory ring: same constructor arguments and messages.
class shm_block_source : public block_source {
    shm_ring* ring = nullptr;
    std::ostream& log;
    uint64_t header_size = 0, block_size = 0, bytes_read = 0, block_id = 0, expected_bytes = 0;
    bool registered = false;

public:
    shm_block_source(const char* name, int core, bool pin, std::ostream& log);
    ~shm_block_source() override;
    shm_block_source(const shm_block_source&) = delete;
    shm_block_source& operator=(const shm_block_source&) = delete;
    bool ok() const { return ring != nullptr; }
    bool is_pinned() const { return registered; }
    void expect_block_bytes(uint64_t n) { expected_bytes = n; }  // N_BYTES_PRE_EXPANSION_PER_BLOCK check, :101-103
    bool close_releases_block() const override { return true; }  // the writer reuses the slot as soon as it is closed
    void read_headers() override;
    char* read() override;
    void close() override;
    bool check_transfers_complete() override;
    uint64_t get_block_size() const override { return block_size; }
    uint64_t get_bytes_read() const override { return bytes_read; }
    const char* get_header() const { return ring ? ring->get_header() : ""; }
};

#ifdef DSABF_WITH_PSRDADA
// dada_handler itself (src/dada_handler.hh:1-177) on a real PSRDADA ring -- only in builds made with -DDSABF_WITH_PSRDADA
// against libpsrdada (csrc/bf_dada.cpp).  Same constructor arguments as the reference (name, core, hex key), same
// messages; errors that make the reference exit(-1) make ok() false / the loop's source fail instead.  No psrdada type
// appears here: the hdu and the multilog are kept as opaque pointers.
class dada_block_source : public block_source {
    void* log = nullptr;      // multilog_t*
    void* hdu_in = nullptr;   // dada_hdu_t*
    std::ostream& out;
    uint64_t header_size = 0, block_size = 0, bytes_read = 0, block_id = 0, expected_bytes = 0;
    bool registered = false, failed = false;
    void cleanup();           // dsaX_dbgpu_cleanup, :118-124
    int dbregister();         // dada_cuda_dbregister, :127-158
    int dbunregister();       // dada_cuda_dbunregister, :160-177

public:
    dada_block_source(const char* name, int core, unsigned in_key, std::ostream& log);   // :25-60
    ~dada_block_source() override;                                                        // :62-64
    dada_block_source(const dada_block_source&) = delete;
    dada_block_source& operator=(const dada_block_source&) = delete;
    bool ok() const { return hdu_in != nullptr && !failed; }
    bool is_pinned() const { return registered; }
    void expect_block_bytes(uint64_t n) { expected_bytes = n; }   // N_BYTES_PRE_EXPANSION_PER_BLOCK check, :101-103
    bool close_releases_block() const override { return true; }   // ipcio_close_block_read hands the block back to the writer
    void read_headers() override;                                  // :66-90
    char* read() override;                                         // :92-94
    void close() override;                                         // :96-98
    bool check_transfers_complete() override;                      // :100-116
    uint64_t get_block_size() const override { return block_size; }
    uint64_t get_bytes_read() const override { return bytes_read; }
};
#endif

// ---- detected-stream sink ---------------------------------------------------------------
// The reference copies each gemm-unit's detected powers into beam_out[stream] and the next gemm-unit of that stream
// overwrites them (src/beamformer.cu:485-488); "writing out ... has not yet been implemented" (README.md:149).  A sink
// gives the stream a consumer without changing the loop: the D2H copy of gemm-unit g lands in acquire(g) -- a pinned
// slot of a ring -- and once the analysis event of g's block has been observed (src/observation_loop.hh:100-117) the
// loop commits the block's gemm-units in index order; commit hands the slot to deliver() and frees it.
// Asynchronous sinks (file, shared-memory ring) deliver on a thread of their own, in order: the loop's commit() only
// marks the slot ready, so a slow consumer costs the loop nothing until the slot ring (five PSRDADA blocks of gemm-units)
// is full -- then acquire() waits for the oldest slot, which is the back-pressure a PSRDADA writer applies.
class detected_sink {
    size_t floats_per_gemm;
    uint64_t n_slots;
    float* ring = nullptr;
    bool pinned = false;
    bool async = false;
    uint64_t next_commit = 0, delivered = 0;   // committed by the loop / handed to deliver(); guarded by `m` when async
    bool failed = false, stop = false;
    std::mutex m;
    std::condition_variable work, done;
    std::thread worker;
    void run();

protected:
    virtual bool deliver(uint64_t gemm_index, const float* data, size_t n_floats) = 0;  // in gemm order
    // `count` consecutive gemm-units that are contiguous in the slot ring (an asynchronous sink that has fallen behind is
    // handed everything that is ready at once); the default hands them to deliver() one by one
    virtual bool deliver_many(uint64_t first_gemm, const float* data, size_t n_floats_each, uint64_t count);
    virtual void finish() {}
    void drain_and_stop();   // every derived destructor calls this first: deliver() must not run on a half-destroyed object

public:
    // slots: gemm-units that can be in flight; the loop needs (MAX_TOTAL_SEP + 1) * N_GEMMS_PER_BLOCK
    detected_sink(const bf_config& cfg, uint64_t slots, bool asynchronous = false);
    virtual ~detected_sink();
    detected_sink(const detected_sink&) = delete;
    detected_sink& operator=(const detected_sink&) = delete;
    static uint64_t slots_for(const bf_config& cfg) { return (uint64_t)(kMaxTotalSep + 1) * cfg.n_gemms_per_block; }
    bool ok();
    float* acquire(uint64_t gemm_index);   // nullptr if the slot still holds an uncommitted gemm-unit (or was committed
                                           // already); waits while an asynchronous delivery of its last occupant is pending
    bool commit(uint64_t gemm_index);      // gemm units must be committed in increasing order, each exactly once
    void close()                           // everything committed has been delivered when this returns
    {
        drain_and_stop();
        finish();
    }
    uint64_t get_delivered();
    size_t get_floats_per_gemm() const { return floats_per_gemm; }
};

// Raw file: a 4096-byte ASCII header (PSRDADA style `KEY value` lines, NUL padded) followed by the gemm-units in
// order, each [N_OUTPUTS_PER_GEMM][N_FREQUENCIES][N_BEAMS] little-endian float32 -- i.e. one long [o][f][b] series.
class file_sink : public detected_sink {
    int fd = -1;
    int write_threads = 4;   // a backlog of several gemm-units is written with positional writes from this many threads

protected:
    bool deliver(uint64_t gemm_index, const float* data, size_t n_floats) override;
    bool deliver_many(uint64_t first_gemm, const float* data, size_t n_floats_each, uint64_t count) override;
    void finish() override;

public:
    static constexpr size_t kHeaderBytes = 4096;
    file_sink(const bf_config& cfg, const char* path, int gpu, uint64_t slots = 0);
    ~file_sink() override;
    bool is_open() const { return fd >= 0; }
};

// Hands every gemm-unit to another process through a shared-memory ring (one ring block per gemm-unit, then a short
// block): the output side of the PSRDADA picture the reference left open ("writing out to the PSRDADA buffer has not yet
// been implemented", README.md:149).  The sink creates the ring; a consumer attaches by name and reads until the short
// block; like any PSRDADA writer the loop blocks while the ring is full.
class ring_sink : public detected_sink {
    shm_ring* out = nullptr;
    std::string name;

protected:
    bool deliver(uint64_t gemm_index, const float* data, size_t n_floats) override;
    void finish() override;

public:
    ring_sink(const bf_config& cfg, const char* ring_name, uint64_t ring_blocks, int gpu, uint64_t slots = 0);
    ~ring_sink() override;
    bool is_open() const { return out != nullptr; }
};

// Keeps everything in host memory (tests, small runs).
class memory_sink : public detected_sink {
protected:
    bool deliver(uint64_t, const float* data, size_t n_floats) override
    {
        data_.insert(data_.end(), data, data + n_floats);
        return true;
    }

public:
    std::vector<float> data_;
    memory_sink(const bf_config& cfg, uint64_t slots = 0) : detected_sink(cfg, slots ? slots : slots_for(cfg)) {}
};

// ---- DM-trial dedispersion of the detected stream, as a stage of the loop ------------------------
// The reference collapses frequency inside its loop behind every gemm-unit (src/beamformer.cu:492-511, DM 0 only).  With
// observation_options::dm_delays set, run_observation pushes every analysed block's beam-blocks into a bf_dm_stream
// (include/dsabf.h) on the block's compute queue -- on the gather root, the gathered whole-band rows -- and hands the chunks
// [n_dm][n_t][n_beams] that become complete to a dm_chunk_sink, in time order, once the block's analysis event has fired.
struct dm_chunk_sink {
    virtual ~dm_chunk_sink() {}
    // out[dm][t][b] for t = first_t .. first_t + n_t - 1 (counted from the first beam-block of the observation)
    virtual bool deliver(uint64_t first_t, int n_t, int n_dm, int n_beams, const float* data) = 0;
    virtual void close() {}
};

// File: a 4096-byte ASCII header (`KEY value` lines, NUL padded; N_DM, N_BEAMS, N_FREQUENCIES, MAX_DELAY, the ladder's
// delays are not repeated), then one record per chunk: 32 bytes {uint64 first_t; uint32 n_t, n_dm, n_beams; 12 bytes of
// zeros} followed by n_dm * n_t * n_beams little-endian float32 in [dm][t][beam] order.
class dm_file_sink : public dm_chunk_sink {
    int fd = -1;
    uint64_t written_t = 0, chunks = 0;

public:
    static constexpr size_t kHeaderBytes = 4096;
    static constexpr size_t kRecordBytes = 32;
    dm_file_sink(const bf_config& cfg, int n_freq_total, int n_dm, int max_delay, const char* path, int gpu, int first_trial = 0);
    ~dm_file_sink() override;
    dm_file_sink(const dm_file_sink&) = delete;
    dm_file_sink& operator=(const dm_file_sink&) = delete;
    bool is_open() const { return fd >= 0; }
    bool deliver(uint64_t first_t, int n_t, int n_dm, int n_beams, const float* data) override;
    void close() override;
    uint64_t get_times_written() const { return written_t; }
    uint64_t get_chunks_written() const { return chunks; }
};

// Hands every chunk to another process through a shared-memory ring (the downstream search's input): one ring block per chunk --
// the 32-byte record {uint64 first_t; uint32 n_t, n_dm, n_beams; zeros} followed by [dm][n_t][beam] float32, padded to the block
// size (32 + n_dm * max_rows * n_beams * 4 bytes: chunks emitted before the delay window has filled hold fewer times) -- then a
// short block.  The sink creates the ring; like any PSRDADA writer the loop blocks while the consumer is a whole ring behind.
class dm_ring_sink : public dm_chunk_sink {
    shm_ring* out = nullptr;
    std::string name;
    size_t block_bytes = 0;
    uint64_t chunks = 0;

public:
    dm_ring_sink(const bf_config& cfg, int n_freq_total, int n_dm, int max_delay, int max_rows, const char* ring_name, uint64_t ring_blocks,
                 int gpu, int first_trial = 0);
    ~dm_ring_sink() override;
    dm_ring_sink(const dm_ring_sink&) = delete;
    dm_ring_sink& operator=(const dm_ring_sink&) = delete;
    bool is_open() const { return out != nullptr; }
    bool deliver(uint64_t first_t, int n_t, int n_dm, int n_beams, const float* data) override;
    void close() override;
    uint64_t get_chunks_written() const { return chunks; }
};

// Keeps the chunks in host memory, assembled as one [n_dm][T][n_beams] series on request (tests, small runs).
class dm_memory_sink : public dm_chunk_sink {
public:
    struct chunk {
        uint64_t first_t;
        int n_t, n_dm, n_beams;
        std::vector<float> data;
    };
    std::vector<chunk> chunks;
    bool deliver(uint64_t first_t, int n_t, int n_dm, int n_beams, const float* data) override
    {
        chunks.push_back({first_t, n_t, n_dm, n_beams, std::vector<float>(data, data + (size_t)n_dm * n_t * n_beams)});
        return true;
    }
};

struct observation_options {
    int gpu = 0;          // -g
    int device = 0;
    int burn_in = 0;      // BURNIN read/close cycles before the loop (src/beamformer.cu:348-355)
    bool verbose = false;
    detected_sink* sink = nullptr;  // optional consumer of every gemm-unit's detected powers; replaces beam_out as
                                    // the D2H destination (beam_out then stays zero)
    // launch granularity: true = bf_enqueue_block launches over `units_per_launch` consecutive gemm-units of the block
    // (0 = the whole PSRDADA block in one launch, the default), consecutive launches on consecutive compute queues, each
    // followed on its queue by the D2H copies of its units (into the sink's slots, or beam_out[queue]); false = the
    // reference's own pattern, one launch per gemm-unit round-robin over the queues (src/beamformer.cu:454-519).
    // Identical detected powers either way.  End to end the loop is PCIe-bound and the granularity does not matter
    // (9.7-10.5 us per beam-block for 1 ... 32 units per launch, profiles/r02_streaming.txt); the whole block per launch
    // is what keeps the kernel at its best rate once the input is resident (profiles/r02_launch_size.txt).
    bool block_launch = true;
    int units_per_launch = 0;
    // -R / -r: this process beamforms frequencies [rank * n_freq, (rank + 1) * n_freq) of a world x n_freq sub-band (cfg.n_freq
    // is the LOCAL count).  The weights are generated for those channels; the input blocks are the rank's own slice.
    int world = 1, rank = 0;
    // Sharded run: the communicator of the frequency partition (bf_comm_create; rank / world above must agree with it).
    // After every block the detected powers of all shards are gathered to rank `gather_root` (BF_GATHER_ROOT_ALL: to every rank) in the reference's
    // [unit][output][freq over the whole band][beam] layout; only that rank copies to the host / feeds `sink` (which
    // must then be built for the whole band: cfg.n_freq * world) -- the others pass sink = nullptr.  Needs block_launch.
    bf_comm* comm = nullptr;
    int gather_root = 0;
    // Transport of that gather (both give the same [unit][o][f over the band][b] on the root): false = every (row, sender) run
    // received in place (bf_gather_detected), true = one message per sender + one device re-layout pass
    // (bf_gather_detected_staged).
    bool gather_staged = false;
    // Sharded run: false if THIS rank's caller failed its own preparations (a sink that did not open, a source that is not there).
    // Every shard exchanges one "ready" flag before the loop; if any is false, all of them return BF_ERR_STATE and nothing is started
    // -- a rank that bails out alone would leave the others waiting in the first gather for ever.  A caller that detects such a
    // failure therefore still CALLS run_observation (with this flag cleared) instead of returning on its own.
    bool local_setup_ok = true;
    // DM-trial dedispersion of the detected stream (needs block_launch): int32 [n_dm][cfg.n_freq * world] sample delays, all >= 0
    // (dm_delays()).  Every analysed block is pushed into a bf_dm_stream behind its launch -- on a sharded run by the gather
    // root, on the gathered band -- and the chunks go to dm_sink (may be NULL: the stage still runs, e.g. for timing).
    const int32_t* dm_delays = nullptr;
    int n_dm = 0;
    dm_chunk_sink* dm_sink = nullptr;
    // Sharded runs only, with gather_root = BF_GATHER_ROOT_ALL (every rank receives the whole band: the ONE collective of the path
    // becomes an all-gather): rank r dedisperses ITS share of the ladder -- trials [r n / R, (r + 1) n / R), the first n % R ranks
    // one more -- so the DM work scales with the GPUs.  Every rank may then have a dm_sink of its own; a chunk carries the rank's
    // trials only (dm_file_sink records the first one as DM_FIRST_TRIAL).
    bool dm_split_trials = false;
};
struct observation_result {
    float observation_time_ms = 0;
    uint64_t blocks = 0;            // blocks analysed
    uint64_t data_chunks = 0;       // blocks * N_GEMMS_PER_BLOCK * N_OUTPUTS_PER_GEMM
    double gbytes_per_s = 0;        // "Approximate datarate", src/beamformer.cu:554
    std::vector<float> beam_out;    // final contents of beam_out: [stream][N_F_PER_DETECT] (src/beamformer.cu:249,485-488)
    std::vector<long long> last_gemm;  // global gemm-unit index (block * N_GEMMS_PER_BLOCK + time_slice) behind each stream
    uint64_t dm_times = 0;          // output times the DM stage produced (dm_delays set): rows analysed - the largest delay
    uint64_t dm_chunks = 0;         // chunks handed to dm_sink
};
// The reference's production main() loop on top of the C-ABI.  pos/dir: antenna positions and beam directions.
int run_observation(const bf_config& cfg, const observation_options& opt, block_source& source, const antenna* pos,
                    const beam_direction* dir, observation_result* res, std::ostream& log);

}  // namespace dsabf
