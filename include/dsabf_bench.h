/* dsabf_bench.h -- measurement and test instrumentation of libdsabf.so.  NOT part of the drop-in boundary.
 *
 * include/dsabf.h is what a maintainer of the reference binds: one entry point per cluster of CUDA / cuBLAS calls, each citing
 * the call sites it replaces.  Everything here exists for benchmarks, roofline reports, A/B runs and tests: switches that pick
 * among launches producing the same bits, counters, launch-shape introspection, the matrix pipe's micro-benchmark.  Nothing
 * in this header is needed to run an observation, and none of it has a counterpart in the reference.
 * (bench.py, tools/ and tests/ use it through dsabeamformer_amd/_lib.py.) */
#ifndef DSABF_BENCH_H
#define DSABF_BENCH_H

#include "dsabf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The matrix pipe by itself, for roofline reports ("a back-to-back v_mfma micro-benchmark; report utilisation
 * against both nominal and measured peak"): one launch of `iters` x 16 independent v_mfma_i32_16x16x64_i8 per wave, 4 waves per
 * SIMD, nothing else in the loop.  Operands are read from the caller's buffer (>= 3 MiB; A = bytes & 0xF0 as the fused kernel
 * sees voltages, B = bytes as it sees weights): the clock the chip holds depends on the operand bits.  d_scratch: >= 4 MiB.
 * *ops = int8 ops the launch executes; the caller times it (HIP events on hip_stream). */
int bf_mfma_peak_device(bf_handle *h, const void *d_operands, size_t operand_bytes, void *d_scratch, size_t scratch_bytes,
                        int iters, double *ops, void *hip_stream);

/* The device pass of the staged gather transport (bf_gather_detected_staged) by itself, for its roofline on ONE GPU:
 * d_full[row][rank][row_floats] = d_stage[rank][row][row_floats] for every rank but skip_rank (-1: none).  HBM-bound: every
 * float read once and written once.  16-byte aligned pointers, row_floats a multiple of 4. */
int bf_gather_relayout_device(bf_handle *h, const float *d_stage, float *d_full, size_t rows_held, int world, size_t row_floats,
                              int skip_rank, void *hip_stream);

/* Introspection for benchmarks/roofline reports. */
int bf_kernel_info(const bf_handle *h, int n_units, int *grid, int *block, int *lds_bytes, int *vgprs);
/* Accumulation windows without a compile-time instantiation (n_pol * n_avg not 2 .. 64 a power of two): how the launch of n_units
 * gemm-units lays a frequency's samples out -- windows_per_stream whole windows per lane-group stream (0: this geometry does not
 * run the run-time-window class), chunks_total 128-row chunks per frequency, padding included.  No GPU needed. */
int bf_rtw_plan(const bf_config *cfg, int n_units, int n_cus, int *windows_per_stream, int *chunks_total);
/* Measurement / test switches of ONE handle (A/B runs inside one process).  They select among launches and kernels that
 * produce the same bits; none of them is needed in production.  The environment variables of the same meaning are read ONCE,
 * at bf_create (DSABF_TSPLIT, DSABF_LDS_PAD, DSABF_DM_WIDE) -- never in a launch path.
 *   "tsplit"   n >= 0   time splits per frequency of the fused launch (0: the library decides)
 *   "rtw_kout" 0 .. 32  whole windows per lane-group stream of a run-time-window launch (accumulation windows without a
 *                       compile-time instantiation; 0: the library picks the one with the least padding that still fills the chip)
 *   "lds_pad"  bytes    extra dynamic LDS per workgroup (fewer resident workgroups per CU); clamped to what a CU has
 *   "dm_wide"  0 / 1    0: bf_dedisperse_dm*_device runs the per-thread-window kernel alone
 *   "dm_ring"  0 / 1    0: the next bf_dm_stream_create keeps its rows in a linear buffer whose carried-over window slides back to the
 *                       start when the end is reached (what a device without virtual-memory management gets), instead of the ring
 *                       that is mapped twice back to back (nothing ever moves); same chunks
 *   "paired"   0 / 1    0: the next bf_set_weights selects the general kernel even for conjugate-symmetric weights
 *   "coalesce" 0 / 1    0: bf_enqueue_gemm_unit launches one kernel per call (the reference's literal launch pattern) */
int bf_set_switch(bf_handle *h, const char *name, int value);
/* Counters of one handle: "fused_launches" = fused-kernel launches issued so far (what coalescing saves),
 * "queued_units" = gemm-units bf_enqueue_gemm_unit has queued and not launched yet, "dm_ring_stages" = live DM stages of the handle
 * whose buffer is the twice-mapped ring (a device without virtual-memory management gives them the linear buffer instead). */
int bf_get_counter(const bf_handle *h, const char *name, uint64_t *value);
int bf_kernel_name(const bf_handle *h, char *buf, size_t buflen); /* which fused kernel this geometry runs */
/* The same answers WITHOUT a handle or a device: which kernel and launch shape a configuration would run for n_units gemm-units
 * on a chip of n_cus compute units (MI355X: 256); paired != 0: as for a conjugate-symmetric weight set (honoured where a
 * conjugate-pair kernel exists).  Pure host arithmetic -- for planning, and so that the launch logic is testable anywhere. */
int bf_launch_plan(const bf_config *cfg, int paired, int n_units, int n_cus, int *grid, int *block, int *lds_bytes, char *name,
                   size_t name_len);
/* The instantiation census (tests/test_census_cpu.py, tests/test_gpu_census.py, profiles/r06_instantiations.txt): which COMPILED kernel
 * a configuration selects, spelled as its demangled symbol spells the template arguments -- "fused16_kernel<-1, 32, false, 0, true,
 * 4, 4>" = <antenna class, window (0: run-time), stage-parity store, detect mode, conjugate-pair, waves per workgroup, output slots
 * per wave>, "fusedg_kernel<true, 0, false>" = <16-byte rows, detect mode, stage-parity store>.  write_c != 0: the kernel of
 * bf_gemm_device (the stage-parity launch).  bf_variant_key is host arithmetic (no device); bf_handle_variant_key answers for a
 * live handle -- after bf_set_weights, which decides `paired` -- so a test can assert that the launch it checked against the oracle
 * WAS the instantiation it meant to cover. */
int bf_variant_key(const bf_config *cfg, int paired, int write_c, char *buf, size_t buflen);
int bf_handle_variant_key(const bf_handle *h, int write_c, char *buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* DSABF_BENCH_H */
