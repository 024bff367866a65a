/* dsabf.h -- C-ABI of libdsabf.so, the MI355X-native (gfx950) DSA beamformer hot path.
 *
 * The reference (devincody/DSAbeamformer) has no plugin/FFI layer: its device code is inlined into main()
 * of one CUDA translation unit.  This header is the drop-in boundary for that implicit interface -- one entry
 * point per cluster of CUDA/cuBLAS calls made by main(), observation_loop_state and test_data_generator
 * (SURVEY.md section 8b).  Every entry point cites the reference call sites it replaces (file:line under
 * the reference repository).  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - every function returns int: 0 = BF_OK, negative = error; bf_last_error() gives the message of the
 *     calling thread's most recent failure.  The library never calls exit() (the reference's gpuErrchk does,
 *     src/beamformer.cuh:19-29; a driver may keep that policy on top of the return codes).
 *   - one host thread drives one handle (as in the reference, src/beamformer.cu:364-534); the library is
 *     re-entrant per handle but a handle is not thread-safe.
 *   - "gemm-unit": the work of one cublasGemmStridedBatchedEx call in the reference = n_freq batches of
 *     [n_beams x n_ant] . [n_ant x n_time], n_time = n_out_per_gemm * n_pol * n_avg, producing n_out_per_gemm
 *     detected [n_freq][n_beams] float32 "beam-blocks".
 *
 * Data layouts (identical to the reference's)
 *   packed voltages  uint8  [unit][freq][time][ant]      one byte per complex sample, high nibble = real,
 *                                                         low nibble = imaginary, two's complement 4-bit
 *   weights          int8   [freq][ant][beam]{re,im}     src/beamformer.cu:230-241
 *   detected power   float  [unit][output][freq][beam]   src/beamformer.cuh:147
 */
#ifndef DSABF_H
#define DSABF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BF_OK 0
#define BF_NOT_READY 1          /* bf_event_query: work still in flight (cudaErrorNotReady) */
#define BF_ERR_INVALID (-1)     /* bad argument / unsupported geometry */
#define BF_ERR_DEVICE (-2)      /* HIP runtime error (message in bf_last_error) */
#define BF_ERR_NO_DEVICE (-3)   /* no gfx950 device visible */
#define BF_ERR_STATE (-4)       /* call out of order (e.g. beamform before bf_set_weights) */

/* Geometry.  The reference fixes these at compile time (src/beamformer.hh:45-152); they are runtime here
 * because BASELINE config 5 needs 100 ant / 512 beams / 1024 freq.  bf_config_default fills the reference's
 * values (debug != 0: `make debug`, n_avg = 1; debug == 0: production, n_avg = 16). */
typedef struct bf_config {
    int n_beams;           /* N_BEAMS            src/beamformer.hh:47   multiple of 4 (:155) */
    int n_ant;             /* N_ANTENNAS         src/beamformer.hh:48   multiple of 4 (:156), at most 2048 */
    int n_freq;            /* N_FREQUENCIES      src/beamformer.hh:49   frequencies owned by THIS handle/GPU */
    int n_pol;             /* N_POL              src/beamformer.hh:52 */
    int n_avg;             /* N_AVERAGING        src/beamformer.hh:55-60  any; n_pol*n_avg in {2,...,64} powers of 2: specialised kernels */
    int n_out_per_gemm;    /* N_OUTPUTS_PER_GEMM src/beamformer.hh:111 */
    int n_gemms_per_block; /* N_GEMMS_PER_BLOCK  src/beamformer.hh:114 */
    int n_blocks_on_gpu;   /* N_BLOCKS_ON_GPU    src/beamformer.hh:124 */
    int n_streams;         /* N_STREAMS          src/beamformer.hh:83 */
    int verbose;           /* -DVERBOSE as a runtime flag */
    int detect_mode;       /* BF_DETECT_CANONICAL (0, default), BF_DETECT_CONTRACTED or BF_DETECT_FAST */
} bf_config;

/* detect_mode: how the power term of detect_sum, `acc += x*x + y*y` (src/beamformer.cuh:150-152), is evaluated.  x, y are
 * the float32 real / imaginary parts fl(n * (float)(1/127)) of one beam sample, n the exact integer dot product.
 *   BF_DETECT_CANONICAL   xx = x*x; yy = y*y; p = xx + yy; acc = acc + p -- two multiplies, an add and the accumulate add,
 *                         ascending time order: what the reference's source says when no contraction happens (g++ on
 *                         x86-64, the CPU restatement in oracle/).  Bit-identical to that restatement for every geometry.
 *   BF_DETECT_CONTRACTED  yy = y*y; p = fma(x, x, yy); acc = acc + p -- what nvcc makes of the same source with its
 *                         default -fmad=true (the reference's makefile:13-16 never turns it off), i.e. most likely the
 *                         bits the reference's GPU build produces.  5 instead of 6 VALU ops per sample.  Bit-identical to
 *                         the restatement's ORC_CONTRACT_NVCC reading.
 *   BF_DETECT_FAST        (n_pol*n_avg >= 16; canonical elsewhere) d = 16 n exactly; acc = fma(d, d, acc) for re then im;
 *                         one (alpha/16)^2 scale per output: 4 VALU ops per sample.
 * Which of the first two the reference's device code really computes cannot be decided without NVIDIA hardware; the
 * stated tolerance covers all three: with E = alpha^2 * sum over the n_ipo samples of |n|^2 evaluated exactly,
 *   |canonical - E|, |contracted - E| <= (n_ipo + 4) * 2^-24 * E        |fast - E| <= (n_ipo + 1) * 2^-23 * E
 * (at most 4 roundings per term -- x, x^2 or the fma, y^2, the pair sum -- plus n_ipo - 1 accumulate roundings of
 * non-negative terms; fast: 2 n_ipo fma roundings, the scale and its constant).  tests/test_oracle.py and
 * tests/test_gpu_parity.py hold all modes to these bounds; measured errors are a few units of 2^-24. */
#define BF_DETECT_CANONICAL 0
#define BF_DETECT_FAST 1
#define BF_DETECT_CONTRACTED 2

typedef struct bf_handle bf_handle; /* owns device memory, 1 transfer queue + n_streams compute queues */
typedef struct bf_event bf_event;

/* Replace gpuErrchk / gpuBLASchk, src/beamformer.cuh:19-36 (print `GPUassert: <msg> <file> <line>` and exit): the message of
 * the calling thread's most recent failure, and the library's identification string. */
const char *bf_last_error(void);
const char *bf_version(void);

/* The reference's compile-time geometry as a value, src/beamformer.hh:45-152 (debug != 0: N_AVERAGING 1, :55-57). */
int bf_config_default(bf_config *cfg, int debug);
/* Derived sizes (src/beamformer.hh:117,120,144,147,137): */
int bf_n_inputs_per_output(const bf_config *cfg);  /* N_INPUTS_PER_OUTPUT */
int bf_n_timesteps_per_gemm(const bf_config *cfg); /* N_TIMESTEPS_PER_GEMM */
size_t bf_bytes_per_gemm(const bf_config *cfg);    /* N_BYTES_PRE_EXPANSION_PER_GEMM */
size_t bf_bytes_per_block(const bf_config *cfg);   /* N_BYTES_PRE_EXPANSION_PER_BLOCK */
size_t bf_floats_per_detect(const bf_config *cfg); /* N_F_PER_DETECT */

/* Device selection.  Replaces CUDA_select_GPU (src/beamformer.cuh:171-192): device index instead of a name
 * match; bf_device_name reports what was found. */
int bf_device_count(int *count);
int bf_device_name(int device, char *buf, size_t buflen);

/* Replaces: allocations src/beamformer.cu:249-266, memsets :291-298, streams/handles :305-320 and the
 * teardown :560-605.  d_B and d_C of the reference do not exist here (expand, GEMM and detect are fused).
 * bf_destroy drains the handle's queues first, as the reference's teardown does (cudaStreamSynchronize of every stream,
 * src/beamformer.cu:560-562): gemm-units that bf_enqueue_gemm_unit accepted and nothing has launched yet are launched, and their
 * host copies land -- so the host_out / host_out_row buffers of every enqueued unit must stay valid until bf_destroy has
 * returned (free pinned memory AFTER the handle, as the reference frees beam_out after its streams, :605,613). */
int bf_create(const bf_config *cfg, int device, bf_handle **out);
int bf_destroy(bf_handle *h);
int bf_get_config(const bf_handle *h, bf_config *cfg); /* the geometry the handle was built for (the #defines of src/beamformer.hh:45-152) */

/* Replaces the weight upload src/beamformer.cu:251,272.  `w` is a HOST array in the reference layout
 * [freq][ant][beam]{re,im} int8; the library re-lays it out once for the MFMA operand fragments.
 * Imaginary parts must be >= -127 (the reference's round(127*sin) never produces -128).
 * If W[f][a][n_beams-1-b] == conj(W[f][a][b]) for every f, a, b -- any beam set symmetric about the boresight, e.g. the
 * reference's -- the library notices (checked on the device, exactly) and runs a kernel that forms each such beam pair
 * from shared products: half the matrix-core work, identical results.  Nothing to configure. */
int bf_set_weights(bf_handle *h, const int8_t *w);
/* Same (the cudaMemcpy of src/beamformer.cu:272 becomes a device-side read), from a DEVICE array: caller-owned HBM, e.g.
 * weights computed on the GPU or a sharded slice. */
int bf_set_weights_device(bf_handle *h, const int8_t *d_w, void *hip_stream);

/* Replaces cudaHostAlloc/cudaFreeHost in src/test_data_generator.hh:35,40 and src/beamformer.cu:212,249,
 * 605,613 (pinned input batch, beam_out, dedispersed_out). */
int bf_alloc_pinned(void **ptr, size_t nbytes);
int bf_free_pinned(void *ptr);
/* Replace cudaHostRegister / cudaHostUnregister on the PSRDADA data blocks (dada_cuda_dbregister /
 * dada_cuda_dbunregister, src/dada_handler.hh:127-177): page-lock caller-owned host memory for DMA. */
int bf_host_register(void *ptr, size_t nbytes);
int bf_host_unregister(void *ptr);

/* Events.  Replace the cudaEvent ring of observation_loop_state (src/observation_loop.hh:58-61,65-68,73,
 * 79,86,95-96,105,112-113).  bf_event_query returns BF_OK (done), BF_NOT_READY, or a negative error. */
int bf_event_create(bf_event **ev);                    /* on the caller's current device (as cudaEventCreate) */
int bf_event_create_on(bf_handle *h, bf_event **ev);   /* on the handle's device: use this for events that bf_submit_block /
                                                        * bf_record_*_event of `h` will record when h's device may not be
                                                        * the caller's current one (`beam -D i`, one thread driving several GPUs) */
int bf_event_destroy(bf_event *ev);
int bf_event_query(bf_event *ev);
int bf_event_synchronize(bf_event *ev);

/* Replaces cudaMemcpyAsync H2D of one PSRDADA block + generate_transfer_event, src/beamformer.cu:389-396,
 * 425-431 and src/observation_loop.hh:71-75.  Copies `nbytes` (<= bf_bytes_per_block) from host memory into
 * ring slot `slot` (0 .. n_blocks_on_gpu-1) on the transfer queue and records `ev` (may be NULL) behind it.
 * The host buffer must stay valid until the event fires. */
int bf_submit_block(bf_handle *h, int slot, const void *host, size_t nbytes, bf_event *ev);

/* Replaces generate_transfer_event's cudaEventRecord(..., HtoDstream), src/observation_loop.hh:73, when the
 * copy and the event are issued separately (bf_submit_block with ev == NULL, then this). */
int bf_record_transfer_event(bf_handle *h, bf_event *ev);

/* Replaces K1-K4 for one gemm-unit, src/beamformer.cu:464-488: expand_input, cublasGemmStridedBatchedEx,
 * detect_sum and the D2H copy of the detected powers.  Reads gemm-unit `time_slice` (0 .. n_gemms_per_block-1)
 * of ring slot `slot`, produces [n_out_per_gemm][n_freq][n_beams] float32 on the device and, if host_out != NULL,
 * copies it there asynchronously (host_out should be pinned).
 * The caller keeps the reference's loop -- one call per gemm-unit, round-robin over the N_STREAMS queues,
 * src/beamformer.cu:454-519 -- but the calls are COALESCED: units are queued and, per run of consecutive time slices of
 * one slot (the reference's loop enqueues 0, 1, 2, ... of a block: one run), launched as ONE kernel, with one DM-0 launch for
 * the units that bf_enqueue_dedisperse was called for and the host copies behind it -- every unit's, in call order, so a
 * host buffer written by several units ends up holding the last one, as on the reference's per-queue streams.  One-unit
 * launches cannot fill the chip (0.28 of the int8 peak alone, 0.385 with 8 in flight; a block launch 0.49); same results.
 * The queued work is launched when a block's worth (n_gemms_per_block units) is queued and at every call that orders or
 * observes device work: bf_record_analysis_event (the reference calls it after each block, :525), bf_stream_sync,
 * bf_enqueue_block*, bf_enqueue_d2h, bf_queue_stream, bf_timer_stop, bf_destroy.  Ordering contract -- the literal pattern's:
 * once launched, a unit's kernel and host copy are ORDERED ON QUEUE stream_idx (that queue waits for the coalesced launch
 * wherever it ran), so anything the caller puts on that queue afterwards -- bf_enqueue_d2h, a hipStreamSynchronize or an event
 * of its own on the hipStream_t from bf_queue_stream, RCCL chained on it -- sees the unit complete; and a unit's results are
 * complete when an event of bf_record_analysis_event recorded after the call fires, or after bf_stream_sync.  What is NOT the
 * literal pattern: a hipStream_t obtained earlier says nothing about units that are still only queued (nothing has been
 * launched for them yet) -- one of the calls above launches them.  DSABF_COALESCE=0 at bf_create (or the "coalesce" switch of
 * dsabf_bench.h): the literal pattern itself, one launch per call on queue stream_idx. */
int bf_enqueue_gemm_unit(bf_handle *h, int stream_idx, int slot, int time_slice, float *host_out);

/* Block-granular form of the same: ONE kernel launch over the n_units consecutive gemm-units [first_unit, first_unit +
 * n_units) of ring slot `slot` on compute queue `stream_idx` (the reference launches K1-K3 once per gemm-unit,
 * src/beamformer.cu:454-519; a whole PSRDADA block of 32 units keeps the chip filled where one unit cannot).  The detected
 * powers land in a per-queue device buffer of n_gemms_per_block units; host_out, if not NULL, is an array of n_units host
 * pointers (pinned; NULL entries are skipped): unit first_unit + i is copied to host_out[i] asynchronously, behind the
 * launch, on the same queue.  Results are identical to n_units calls of bf_enqueue_gemm_unit.
 * The per-queue buffer is allocated when a queue is first used: a caller that rotates over queues reserves them BEFORE its loop
 * with bf_block_output_device(h, q, &p) for every queue it will use (a hipMalloc in the middle of a stream of blocks stalls the
 * device: 9.4 -> 10.9 us per beam-block measured; run_observation and examples/ do).
 * bf_enqueue_block_to: the same launch with the powers written to d_dst (device, room for n_units * bf_floats_per_detect floats,
 * [unit][o][f][b]) instead of the queue's buffer -- for a consumer that owns the place the rows belong in (bf_dm_stream_reserve:
 * the frequency collapse directly behind detect, src/beamformer.cu:481,492-511, with no copy in between). */
int bf_enqueue_block(bf_handle *h, int stream_idx, int slot, int first_unit, int n_units, float *const *host_out);
int bf_enqueue_block_to(bf_handle *h, int stream_idx, int slot, int first_unit, int n_units, float *d_dst, float *const *host_out);

/* Replaces K5, the DEBUG dedisperse, src/beamformer.cu:498-510: sums output 0 of the unit last enqueued on
 * `stream_idx` over frequency (ascending f, fp32) and copies the n_beams floats to host_out_row. */
int bf_enqueue_dedisperse(bf_handle *h, int stream_idx, float *host_out_row);
/* The same K5 (src/beamformer.cu:498-510) for the gemm-units [first_unit, first_unit + n_units) of the block bf_enqueue_block last put on
 * `stream_idx`, in ONE launch (units x beams threads; every beam's sum still runs over ascending f in one thread, so the
 * bits are those of n_units bf_enqueue_dedisperse calls).  host_rows, if not NULL, receives [n_units][n_beams] floats. */
int bf_enqueue_block_dedisperse(bf_handle *h, int stream_idx, int first_unit, int n_units, float *host_rows);

/* Replaces generate_analysis_event, src/beamformer.cu:525 / src/observation_loop.hh:77-81.  The reference
 * records on stream[N_STREAMS-1] only (a latent race); this records `ev` behind ALL
 * compute queues. */
int bf_record_analysis_event(bf_handle *h, bf_event *ev);

/* Replaces cudaStreamSynchronize x N_STREAMS, src/beamformer.cu:560-562 (stream_idx < 0: all queues incl.
 * the transfer queue). */
int bf_stream_sync(bf_handle *h, int stream_idx);

/* Replaces START_TIMER / STOP_RECORD_TIMER, src/beamformer.cuh:43-58 (events on the default queue order). */
int bf_timer_start(bf_handle *h);
int bf_timer_stop(bf_handle *h, float *ms);

/* ---- Device-pointer entry points: operands already resident in HBM (caller-owned memory and queue). ----
 * These are what bench.py and the multi-GPU path call; `hip_stream` is a hipStream_t (NULL = default). */

/* K1-K3 of src/beamformer.cu:464-481 (expand_input, cublasGemmStridedBatchedEx, detect_sum) without the copies around them.
 * Fused a1+a2+a3 over n_units gemm-units: d_packed [n_units][freq][time][ant] -> d_out
 * [n_units][output][freq][beam].  One kernel launch.  Both pointers must be 16-byte aligned (16-byte loads; groups of
 * four beams are stored with one 16-byte store).  Like every entry point, the call leaves the caller's current HIP
 * device as it found it. */
int bf_beamform_device(bf_handle *h, const void *d_packed, int n_units, float *d_out, void *hip_stream);

/* a1 alone (expand_input, src/beamformer.cuh:66-109): nbytes packed bytes -> 2*nbytes int8 (re, im pairs in
 * order).  nbytes must be a multiple of 16, pointers 16-byte aligned. */
int bf_expand_device(bf_handle *h, const void *d_in, size_t nbytes, void *d_out, void *hip_stream);

/* a2 alone, for stage-level parity checks (the reference's d_C, src/beamformer.cu:470-477): one gemm-unit of
 * packed voltages -> complex float32 [freq][time][beam]{re,im} = (1/127) * W * V. */
int bf_gemm_device(bf_handle *h, const void *d_packed_unit, float *d_c, void *hip_stream);

/* a8 alone (the cublasSgemv of src/beamformer.cu:498-504): d_out_unit [output][freq][beam] -> d_ded [beam] = sum over freq
 * of output 0. */
int bf_dedisperse_device(bf_handle *h, const float *d_out_unit, float *d_ded, void *hip_stream);

/* Incoherent dedispersion beyond DM 0 (the reference stops at the DM-0 sum above,
 * src/beamformer.cu:498-504, and sketches the delay law in sandbox/Dispersion Theory.ipynb).  d_series: n_t consecutive beam-blocks [t][freq][beam] (the
 * detected stream is exactly that); d_delays: int32 [n_dm][freq] sample delays (dsabf::dm_delays / bfh_dm_delays);
 * d_out [n_dm][n_t_out][beam] = sum over freq, ascending, fp32, of d_series[t + delay][freq][beam]; rows outside [0, n_t)
 * contribute nothing, so size n_t_out = n_t - (largest delay) for complete sums.  Groups of 32 consecutive trials whose
 * delays, at every channel, span no more than ~100 samples (any fine DM ladder) run a kernel that fetches each window of
 * input rows once for the whole group (csrc/bf_dm_wide.hip); other groups -- coarse or non-monotonic ladders -- a kernel with
 * a window per thread; the result does not depend on which (one thread, one ascending-f sum either way).  Delays are read
 * on the device: nothing about them has to be known to the host. */
int bf_dedisperse_dm_device(bf_handle *h, const float *d_series, int n_t, const int32_t *d_delays, int n_dm, int n_t_out,
                            float *d_out, void *hip_stream);

/* ---- DM-trial dedispersion as a STAGE of the observation loop ---------------------------------
 * The reference collapses frequency INSIDE its loop, right behind each gemm-unit's detect (src/beamformer.cu:492-511: the
 * cublasSgemv of :498-504 and the copy of :506-510) -- at DM 0 only.  A bf_dm_stream does the same for a ladder of DM trials on
 * the detected stream as it is produced: the caller pushes the beam-blocks of every block it has beamformed ([row][freq][beam],
 * rows in time order: exactly what bf_enqueue_block leaves in bf_block_output_device, or bf_gather_detected in the freq-major
 * layout on the gather root), the stream keeps the last max_delay rows on the device in front of the next push, and every push
 * emits the output times that have just become complete:
 *   chunk [n_dm][n_t_out][beam], out[dm][t][b] = sum over f (ascending, fp32) of series[t + delay[dm][f]][f][b],
 *   t = first_t .. first_t + n_t_out - 1 counted from the first row ever pushed.
 * Chunks follow each other without gaps or overlap (first_t of a push = first_t + n_t_out of the one before); concatenated
 * along t they are BIT-IDENTICAL to one bf_dedisperse_dm_device call over the whole series (same kernels, same ascending-f
 * sum per output; the last max_delay times of a series are never complete, there as here).  n_t_out is 0 until more than
 * max_delay rows have been seen; the first chunk holds the times beyond them, every later one n_rows.
 *   delays: HOST int32 [n_dm][n_freq_total], all >= 0 (dsabf::dm_delays / bfh_dm_delays); n_freq_total = the channels of one
 *   pushed row (cfg.n_freq, or world * cfg.n_freq on a gather root); max_rows_per_push: the largest n_rows of a push
 *   (n_gemms_per_block * n_out_per_gemm for block launches).
 * bf_dm_stream_push is asynchronous on `hip_stream` (the queue that produced d_rows: bf_queue_stream); successive pushes are
 * ordered by the stream itself, whichever queues they are issued on: their chunks complete in push order -- but up to three
 * pushes issued on different queues RUN side by side (each in a chunk buffer and scratch of its own; a push waits for the rows
 * of the one before it, not for its end), so every push in flight needs a host_out of its own.  host_out (optional, pinned, room for n_dm * n_rows *
 * n_beams floats) receives the chunk; first_t / n_t_out are known to the host at once (pure arithmetic).  The device copy of
 * the most recent chunk: bf_dm_stream_output_device.  Destroy the stream before its handle (a handle that goes first releases the
 * stage's device memory; the stage then only answers BF_ERR_STATE and can still be destroyed).
 * Zero-copy feed (what run_observation uses): bf_dm_stream_reserve(s, n_rows, &d_dst, hip_stream) hands out the place of the next
 * n_rows rows INSIDE the stage's buffer, directly behind the carried-over window; the producer writes them there ON hip_stream (or
 * ordered behind it) -- bf_enqueue_block_to(h, q, ..., d_dst, ...), or bf_gather_detected(..., d_full = d_dst, stream) on a gather
 * root -- and bf_dm_stream_push(s, d_dst, n_rows, ...) then launches the kernels without moving a row (a push of rows that live
 * elsewhere copies them in first: one more read and write of every row).  One reservation at a time; it must be pushed as
 * reserved (same pointer, same n_rows).  The stage's buffer is a ring of max_delay + 3 max_rows_per_push rows whose memory is
 * mapped twice, back to back (hipMemCreate / hipMemMap): the delay window in front of a push is contiguous wherever it starts, and
 * no row is ever moved or read twice by anything but the dedispersion itself. */
typedef struct bf_dm_stream bf_dm_stream;
int bf_dm_stream_create(bf_handle *h, const int32_t *delays, int n_dm, int n_freq_total, int max_rows_per_push,
                        bf_dm_stream **out);
int bf_dm_stream_destroy(bf_dm_stream *s);
int bf_dm_stream_max_delay(const bf_dm_stream *s);
int bf_dm_stream_reserve(bf_dm_stream *s, int n_rows, float **d_dst, void *hip_stream);
int bf_dm_stream_push(bf_dm_stream *s, const float *d_rows, int n_rows, float *host_out, uint64_t *first_t, int *n_t_out,
                      void *hip_stream);
int bf_dm_stream_output_device(bf_dm_stream *s, float **d_out);

/* ---- Multi-GPU: frequency shards and the gather of their detected powers -------------------------------
 * The reference runs 8 independent processes, one sub-band per GPU (`-g`, src/beamformer.cu:92-100,233; README.md:168)
 * and never brings their outputs together.  Here a handle may own any contiguous range of frequencies (bf_config.n_freq
 * = the LOCAL count; weights generated for those channels), one process per GPU, and the ONE collective of the path is
 * the gather of the detected powers -- RCCL point-to-point over xGMI, received straight at the final position.
 * The input voltages are never exchanged: [freq][time][ant] is frequency-major, every rank reads its own slice.
 *
 * Row = one detected beam-block of one rank = row_floats = n_freq_local * n_beams consecutive floats; a launch over
 * n_units gemm-units produces n_rows = n_units * n_out_per_gemm rows per rank.  Layout of the gathered array:
 *   BF_GATHER_LAYOUT_FREQ_MAJOR  [row][world * n_freq_local][beam] -- the reference's [o][f][b] over the whole band: rank r's
 *                                row o sits at ((o * world + r) * row_floats): one message per (row, sender)
 *   BF_GATHER_LAYOUT_RANK_MAJOR  [rank][row][n_freq_local][beam]   -- sub-band-major (the reference's own 8-stream shape):
 *                                one contiguous message per sender
 * Who receives (root): a rank >= 0: only that rank (d_full may be NULL elsewhere); BF_GATHER_ROOT_ALL: every rank receives
 * everything; BF_GATHER_ROOT_DISTRIBUTED: rank j becomes the owner of rows [j * n_rows/world, (j+1) * n_rows/world) of
 * the WHOLE band (n_rows % world == 0) -- an all-to-all that loads every xGMI link of every GPU in both directions
 * instead of funnelling world-1 shards into one GPU's links, and leaves all frequencies of a time range on one device,
 * which is what a dedispersion search wants.  In the layouts above `row` then counts from the owner's first row and
 * n_rows is the number of rows the receiver holds (bf_gather_rows_held). */
#define BF_GATHER_ROOT_ALL (-1)
#define BF_GATHER_ROOT_DISTRIBUTED (-2)
typedef struct bf_comm bf_comm;
#define BF_COMM_ID_BYTES 128
#define BF_GATHER_LAYOUT_FREQ_MAJOR 0
#define BF_GATHER_LAYOUT_RANK_MAJOR 1
int bf_comm_unique_id(void *id128);                 /* rank 0: ncclGetUniqueId; hand the 128 bytes to the other ranks */
int bf_comm_create(int rank, int world, const void *id128, int device, bf_comm **out); /* world == 1 needs no id / RCCL */
int bf_comm_destroy(bf_comm *c);
int bf_comm_rank(const bf_comm *c);
int bf_comm_world(const bf_comm *c);
/* Evidence for a scaling record (the reference's 8 processes never meet, src/beamformer.cu:92-100): how many ranks the LIBRARY reports for the communicator (ncclCommCount; 0 = this bf_comm
 * has no RCCL communicator, -1 = the library cannot say), its version (ncclGetVersion, e.g. 22203; 0 = unknown) and the
 * file the point-to-point calls were resolved from.  Any pointer may be NULL. */
int bf_comm_info(const bf_comm *c, int *lib_ranks, int *version, char *lib_path, size_t n);
int bf_comm_library_info(int *version, char *lib_path, size_t n); /* the same two answers BEFORE a communicator exists (dlopen only) */
int bf_gather_detected(bf_comm *c, const float *d_local, size_t n_rows, size_t row_floats, int root, int layout,
                       float *d_full, void *hip_stream);
/* The same gather into BF_GATHER_LAYOUT_FREQ_MAJOR -- the reference's [o][f][b] over the whole band, src/beamformer.cuh:147 --
 * by a second TRANSPORT.  bf_gather_detected receives every (row, sender) run at its final position: no extra pass over the
 * data, but one point-to-point message per row and sender (14,336 messages of 32 KiB per rank for a 128-gemm-unit step of
 * BASELINE configs[3] on 8 GPUs).  This entry point puts ONE message per sender on the wire (the rank-major plan, received
 * into d_stage) and then moves the rows to their freq-major places with one device pass (every float read and written once,
 * whole 128-byte lines, nontemporal); the receiver's own rows go straight from d_local to d_full.  d_stage: caller-owned,
 * as large as d_full (rows held x world x row_floats floats; may be NULL on ranks that receive nothing, and when world == 1);
 * row_floats a multiple of 4, both buffers 16-byte aligned.  Bit-identical d_full either way; which is faster is a property
 * of the fabric and the message count (bench.py times both: gather_modes.*_freq_major[_staged]). */
int bf_gather_detected_staged(bf_comm *c, const float *d_local, size_t n_rows, size_t row_floats, int root, float *d_full,
                              float *d_stage, void *hip_stream);
/* The layout arithmetic as plain host functions (no device, no RCCL): float offset of (rank, row) in the gathered array --
 * [o][f][b] of src/beamformer.cuh:147 with f running over the shards -- and the list of messages one rank issues (kind: send
 * to peer, receive from peer, or copy its own rows). */
size_t bf_gather_offset(int layout, size_t n_rows_held, size_t row_floats, int world, int rank, size_t row);
size_t bf_gather_rows_held(size_t n_rows, int world, int rank, int root); /* rows `rank` ends up holding (0: it only sends) */
#define BF_GATHER_SEND 0
#define BF_GATHER_RECV 1
#define BF_GATHER_COPY 2
typedef struct bf_gather_msg {
    int kind, peer;
    size_t local_offset; /* floats into d_local (send / copy source; for a receive: the sender's offset) */
    size_t full_offset;  /* floats into d_full (receive / copy destination) */
    size_t count;        /* floats */
} bf_gather_msg;
size_t bf_gather_plan(int layout, size_t n_rows, size_t row_floats, int world, int rank, int root, bf_gather_msg *msgs,
                      size_t capacity); /* returns the number of messages (call with capacity 0 to size the array) */

/* Device pointer of the per-queue block buffer bf_enqueue_block fills ([n_gemms_per_block][output][freq][beam]: the
 * reference's d_dedispersed per stream, src/beamformer.cu:258,481) and the hipStream_t of compute queue `stream_idx`
 * (stream[i], src/beamformer.cu:305-312): what a caller needs to chain bf_gather_detected behind a block launch. */
int bf_block_output_device(bf_handle *h, int stream_idx, float **d_out);
int bf_queue_stream(bf_handle *h, int stream_idx, void **hip_stream);
/* A handle-owned device buffer (allocated like src/beamformer.cu:249-266 allocates the per-stream ones) for the gathered block of compute queue `stream_idx`: n_gemms_per_block * world *
 * bf_floats_per_detect floats ([unit][output][world * n_freq][beam] with the freq-major layout); allocated on first use. */
int bf_block_gather_device(bf_handle *h, int stream_idx, int world, float **d_full);
int bf_block_gather_stage_device(bf_handle *h, int stream_idx, int world, float **d_stage); /* same size: d_stage of bf_gather_detected_staged */
/* Asynchronous device-to-host copy of n_floats on compute queue `stream_idx` (the cudaMemcpyAsync of src/beamformer.cu:485-488
 * for a caller-chosen source): behind everything enqueued on that queue before the call, gemm-units included. */
int bf_enqueue_d2h(bf_handle *h, int stream_idx, const float *d_src, float *host_dst, size_t n_floats);

/* The two dedispersion entry points above (src/beamformer.cu:498-504) on a gathered band: the same kernels over n_freq_total = world * n_freq
 * channels of a [row][n_freq_total][beam] array (BF_GATHER_LAYOUT_FREQ_MAJOR on the receiving rank) -- ascending f over the
 * WHOLE band, i.e. the bits one GPU holding all channels would produce. */
int bf_dedisperse_band_device(bf_handle *h, const float *d_out_unit, int n_freq_total, float *d_ded, void *hip_stream);
int bf_dedisperse_dm_band_device(bf_handle *h, const float *d_series, int n_t, int n_freq_total, const int32_t *d_delays,
                                 int n_dm, int n_t_out, float *d_out, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* DSABF_H */
