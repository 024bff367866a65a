/* dsabf_host.h -- C-ABI wrappers over the C++ host mirror (include/dsabf_host.hpp) so that non-C++ callers
 * (ctypes tests, other FFIs) can drive the reference's host-side entry points:
 *   weights           src/beamformer.cu:230-241          config readers  src/beamformer.hh:250-284
 *   python writer     src/beamformer.hh:287-311          generator       src/test_data_generator.hh:11-108
 *   scheduler         src/observation_loop.hh:1-177      DEBUG main()    src/beamformer.cu:12-621
 * Same conventions as dsabf.h: int return, 0 = ok, negative = error, bf_last_error() for the message.
 */
#ifndef DSABF_HOST_H
#define DSABF_HOST_H

#include "dsabf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* pos: n_ant x {x,y,z} float32; dir: n x {theta,phi} float32 (radians) */
int bfh_default_positions(int n_ant, float *pos);
int bfh_default_directions(int n_beams, float *dir);
int bfh_read_positions(const char *path, int n_ant, float *pos);
int bfh_read_directions(const char *path, int expected, float *dir);
int bfh_count_entries(const char *path); /* first token of a config file; < 0 if unreadable */
int bfh_write_python_file(const float *data, int rows, int cols, const char *path);
float bfh_channel_frequency(int generator_variant, int gpu, int chan);
int bfh_make_weights(int n_beams, int n_ant, int n_freq, int chan0, int gpu, const float *pos, const float *dir,
                     int8_t *out);

/* test_data_generator */
typedef struct bfh_generator bfh_generator;
int bfh_gen_create(const bf_config *cfg, int n_sources_per_batch, int pin, bfh_generator **out);
int bfh_gen_destroy(bfh_generator *g);
int bfh_gen_read_sources(bfh_generator *g, const char *path);
int bfh_gen_set_sources(bfh_generator *g, const float *src, int n);
int bfh_gen_generate(bfh_generator *g, const float *pos, int gpu);
void *bfh_gen_data(bfh_generator *g);
size_t bfh_gen_size(bfh_generator *g);
int bfh_gen_n_pt_sources(bfh_generator *g);
int bfh_gen_need_more(bfh_generator *g, int blocks_transferred);
int bfh_gen_ready(bfh_generator *g, int blocks_transfer_queue);

/* observation_loop_state on HIP events of handle `h`'s queues (src/observation_loop.hh:54-176). */
typedef struct bfh_obs bfh_obs;
int bfh_obs_create(uint64_t max_transfer_sep, uint64_t max_total_sep, const bf_config *cfg, bf_handle *h, int debug_mode,
                   bfh_obs **out);
/* The same scheduler on caller-supplied events (another runtime's, or a test's: the CPU tests drive the state machine
 * with events they complete -- or fail -- by hand).  create returns NULL on failure; record_* return BF_OK or < 0;
 * query returns BF_OK (done), BF_NOT_READY or < 0.  The table is copied; `user` must outlive the bfh_obs. */
typedef struct bfh_event_ops {
    void *user;
    void *(*create)(void *user);
    void (*destroy)(void *user, void *ev);
    int (*record_transfer)(void *user, void *ev);
    int (*record_analysis)(void *user, void *ev);
    int (*query)(void *user, void *ev);
} bfh_event_ops;
int bfh_obs_create_custom(uint64_t max_transfer_sep, uint64_t max_total_sep, const bf_config *cfg, const bfh_event_ops *ops,
                          int debug_mode, bfh_obs **out);
/* BF_OK, or the first error an event operation reported (sticky): stop polling and return it. */
int bfh_obs_status(bfh_obs *o);
int bfh_obs_destroy(bfh_obs *o);
int bfh_obs_generate_transfer_event(bfh_obs *o);
int bfh_obs_generate_analysis_event(bfh_obs *o);
int bfh_obs_check_transfer_events(bfh_obs *o);
int bfh_obs_check_analysis_events(bfh_obs *o);
int bfh_obs_counters(bfh_obs *o, uint64_t *A, uint64_t *AQ, uint64_t *T, uint64_t *TQ);
int bfh_obs_check_ready_for_transfer(bfh_obs *o);
int bfh_obs_check_ready_for_analysis(bfh_obs *o);
int bfh_obs_check_ready_for_dh2_transfer(bfh_obs *o, int time_slice);
int bfh_obs_check_observations_complete(bfh_obs *o);
int bfh_obs_check_transfers_complete(bfh_obs *o);
int bfh_obs_set_transfers_complete(bfh_obs *o, int v);
int bfh_obs_set_n_pt_sources(bfh_obs *o, int n);
uint64_t bfh_obs_get_current_analysis_gemm(bfh_obs *o, int time_slice);
uint64_t bfh_obs_get_current_transfer_gemm(bfh_obs *o);
uint64_t bfh_obs_get_next_gpu_analysis_block(bfh_obs *o);
uint64_t bfh_obs_get_next_gpu_transfer_block(bfh_obs *o);
int bfh_obs_describe(bfh_obs *o, char *buf, size_t buflen); /* operator<<, src/observation_loop.hh:172-176 */

/* The reference's `make debug` run end to end (generate -> H2D -> beamform -> dedisperse -> data.py).
 * Paths may be NULL (defaults src/beamformer.cu:135-147 and BOGUS_DATA).  ded_out (optional) receives
 * [n_pt_sources][n_beams] floats (capacity in floats given by ded_capacity). */
int bfh_run_debug_observation(const bf_config *cfg, int gpu, const char *positions, const char *directions,
                              const char *sources, const char *output, int device, int verbose, float *ded_out,
                              size_t ded_capacity, int *n_pt_sources, float *observation_ms);
/* The same with the caller-side loop chosen (a NEW entry point: the signature above is the round-2 one and stays binary
 * compatible).  per_unit_launches != 0: the reference's own loop -- bf_enqueue_gemm_unit + bf_enqueue_dedisperse per
 * gemm-unit, round-robin over the queues, src/beamformer.cu:454-519 (the library coalesces those calls into one launch per
 * block, include/dsabf.h) -- instead of bf_enqueue_block + bf_enqueue_block_dedisperse; the table is the same bit for bit. */
int bfh_run_debug_observation2(const bf_config *cfg, int gpu, const char *positions, const char *directions,
                               const char *sources, const char *output, int device, int verbose, float *ded_out,
                               size_t ded_capacity, int *n_pt_sources, float *observation_ms, int per_unit_launches);

/* Production observation loop (src/beamformer.cu:364-534 without -DDEBUG) fed by the in-memory dada_junkdb stand-in:
 * n_blocks pseudo-random PSRDADA-sized blocks from a pinned ring of ring_blocks distinct blocks, default linear
 * geometry (src/beamformer.cu:135-147).  Outputs: elapsed ms; beam_out [n_streams][N_F_PER_DETECT] floats (final
 * contents of the reference's beam_out); last_gemm[n_streams] = global gemm-unit index behind each stream's slot;
 * ring_copy (optional, ring_blocks * block bytes) receives the ring so a caller can recompute any block. */
int bfh_run_observation_junk(const bf_config *cfg, uint64_t n_blocks, int ring_blocks, uint64_t seed, int gpu, int device,
                             int burn_in, int verbose, float *observation_ms, float *beam_out, long long *last_gemm,
                             void *ring_copy);

/* Same loop with every gemm-unit's detected powers written to `path` (dsabf::file_sink: 4096-byte ASCII header, then
 * [gemm][o][f][b] float32).  gemms_written (optional) receives the number of gemm-units delivered. */
int bfh_run_observation_junk_to_file(const bf_config *cfg, uint64_t n_blocks, int ring_blocks, uint64_t seed, int gpu,
                                     int device, int burn_in, int verbose, const char *path, float *observation_ms,
                                     uint64_t *gemms_written, void *ring_copy);

/* Same loop with the detected stream handed to another process through a shared-memory ring `out_ring` (dsabf::ring_sink:
 * one block per gemm-unit, then a short block; the call creates the ring, blocks while it is full, and removes it once
 * the consumer has drained it). */
int bfh_run_observation_junk_to_ring(const bf_config *cfg, uint64_t n_blocks, int ring_blocks, uint64_t seed, int gpu,
                                     int device, const char *out_ring, uint64_t out_ring_blocks, float *observation_ms,
                                     uint64_t *gemms_written, void *ring_copy);

/* Same loop with the DM stage on (SURVEY.md 8f-4 as a stage of the loop; where the reference collapses frequency,
 * src/beamformer.cu:492-511): delays int32 [n_dm][cfg->n_freq] (bfh_dm_delays), every analysed block pushed into a bf_dm_stream
 * behind its launch, the chunks written to dm_path (dsabf::dm_file_sink: 4096-byte header, then per chunk a 32-byte record
 * {u64 first_t, u32 n_t, u32 n_dm, u32 n_beams} + float32 [dm][t][beam]; "ring:<name>[:<blocks>]": the same records, one per block of a
 * shared-memory ring the call creates, to another process (dsabf::dm_ring_sink); NULL: the stage runs, nothing is kept) and,
 * optionally, the detected stream itself to detected_path.  dm_times (optional): output times produced. */
int bfh_run_observation_junk_dm(const bf_config *cfg, uint64_t n_blocks, int ring_blocks, uint64_t seed, int gpu, int device,
                                int burn_in, int verbose, const int32_t *delays, int n_dm, const char *dm_path,
                                const char *detected_path, float *observation_ms, uint64_t *dm_times, void *ring_copy);

/* One frequency SHARD of a sharded observation with any geometry (`beam -R world -r rank` is the production-geometry form): cfg is
 * the shard's (n_freq = the local count), id128 the bytes of bf_comm_unique_id from rank 0; after every block the shards' powers are
 * gathered to gather_root (a rank, or BF_GATHER_ROOT_ALL) in the reference's [o][f][b] over the band -- staged != 0: by
 * bf_gather_detected_staged.  delays (optional): int32 [n_dm][world * cfg->n_freq], the DM stage on the rank(s) that hold the band;
 * split_trials != 0 (needs BF_GATHER_ROOT_ALL): each rank takes its dm_trial_share.  detected_path / dm_path: files, written only
 * by ranks that hold the band.  Every shard must be given the same n_blocks. */
int bfh_run_observation_junk_sharded(const bf_config *cfg, uint64_t n_blocks, int ring_blocks, uint64_t seed, int gpu, int device,
                                     int rank, int world, const void *id128, int gather_root, int staged, const int32_t *delays,
                                     int n_dm, int split_trials, const char *detected_path, const char *dm_path,
                                     float *observation_ms, uint64_t *dm_times, void *ring_copy);

/* The sink's ring on its own (tests; works without a device, the ring is then plain memory). */
typedef struct bfh_sink bfh_sink;
int bfh_file_sink_create(const bf_config *cfg, const char *path, int gpu, uint64_t slots, bfh_sink **out);
int bfh_sink_acquire(bfh_sink *s, uint64_t gemm_index, float **slot); /* BF_ERR_STATE if the slot is still occupied */
int bfh_sink_commit(bfh_sink *s, uint64_t gemm_index);                 /* in order, each once */
int bfh_sink_close(bfh_sink *s);
int bfh_sink_destroy(bfh_sink *s);

/* The DM-chunk sinks on their own (tests; no device): target = a file path (dsabf::dm_file_sink) or "ring:<name>[:<blocks>]"
 * (dsabf::dm_ring_sink, which creates the ring; destroy waits until a consumer has drained it).  Chunks must follow each other
 * without gaps (first_t = the times delivered so far), else BF_ERR_STATE. */
typedef struct bfh_dm_sink bfh_dm_sink;
int bfh_dm_sink_create(const bf_config *cfg, const char *target, int n_freq_total, int n_dm, int max_delay, int max_rows,
                       int first_trial, bfh_dm_sink **out);
int bfh_dm_sink_deliver(bfh_dm_sink *s, uint64_t first_t, int n_t, int n_dm, int n_beams, const float *data);
int bfh_dm_sink_destroy(bfh_dm_sink *s);

/* DM trial ladder and per-channel sample delays (sandbox/Dispersion Theory.ipynb cells 1-2 and 5; dsabf::dm_trials,
 * dsabf::dm_delays).  bfh_dm_trials returns the number of trials written (<= cap). */
int bfh_dm_trials(double dm0, double dm_max, int nchan, double epsilon, double nu_ghz, double chan_bw_mhz, double ti_us,
                  double tscat_us, double tsamp_us, double *out, int cap);
int bfh_dm_delays(const double *dms, int n_dm, const float *freq_ghz, int n_freq, double f_ref_ghz, double tsamp_ms,
                  int32_t *out);

/* dsabf::dm_trial_share: which trials rank `rank` of `world` dedisperses when a sharded run splits the ladder. */
int bfh_dm_trial_share(int n_dm, int world, int rank, int *first, int *count);

/* The junk source's bytes: ring_blocks blocks of cfg's block size into `out` (dsabf::junk_fill). */
int bfh_junk_fill(const bf_config *cfg, int ring_blocks, uint64_t seed, void *out);

/* Shared-memory input ring (dsabf::shm_ring, the PSRDADA stand-in: csrc/bf_shmring.cpp). */
typedef struct bfh_shm_ring bfh_shm_ring;
int bfh_shm_ring_create(const char *name, uint64_t n_blocks, uint64_t block_size, const char *header_text,
                        bfh_shm_ring **out);
int bfh_shm_ring_attach(const char *name, int timeout_ms, bfh_shm_ring **out);
int bfh_shm_ring_detach(bfh_shm_ring *r);
int bfh_shm_ring_unlink(const char *name);
int bfh_shm_ring_info(bfh_shm_ring *r, uint64_t *n_blocks, uint64_t *block_size, char *header, size_t header_cap);
/* copy `bytes` (<= block size; fewer = end of data) into the next free block; blocks while the ring is full */
int bfh_shm_ring_write(bfh_shm_ring *r, const void *data, uint64_t bytes);
/* copy the next filled block out (at most cap bytes); blocks while the ring is empty */
int bfh_shm_ring_read(bfh_shm_ring *r, void *out, uint64_t cap, uint64_t *bytes, uint64_t *block_id);

/* Production observation loop fed from the shared-memory ring `name` (what `beam -k name` runs): detected stream to
 * `path` if not NULL.  pinned (optional) receives whether the ring blocks could be page-locked. */
int bfh_run_observation_shm(const bf_config *cfg, const char *name, int core, int gpu, int device, int verbose,
                            const char *path, float *observation_ms, uint64_t *gemms_written, int *pinned);

/* The same with the DM stage on (as bfh_run_observation_junk_dm): delays int32 [n_dm][cfg->n_freq] or NULL, chunks to dm_path. */
int bfh_run_observation_shm_dm(const bf_config *cfg, const char *name, int core, int gpu, int device, int verbose,
                               const char *path, const int32_t *delays, int n_dm, const char *dm_path, float *observation_ms,
                               uint64_t *gemms_written, uint64_t *dm_times, int *pinned);

#ifdef __cplusplus
}
#endif
#endif
